mkdir -p gpurun_out/r03
timeout 600 python tools/p2g_time.py C2 C3 2>&1 | grep -v amdgpu.ids | tail -2
HOT_AMD_AB=1 HOT_FORCE_CELLS1=1 timeout 600 python tools/p2g_time.py C2 C3 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1200 python -m pytest tests/test_gpu_transfer.py tests/test_gpu_golden.py tests/test_gpu_force.py tests/test_gpu_solver.py -q -x 2>&1 | grep -v amdgpu.ids | tail -4
