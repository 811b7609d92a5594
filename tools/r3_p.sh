mkdir -p gpurun_out/r03
timeout 600 python tools/hess_time.py C2 C3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/hess_time.log
timeout 1200 python -m pytest tests/test_gpu_force.py tests/test_gpu_solver.py tests/test_gpu_variants.py tests/test_gpu_golden.py -q 2>&1 | grep -v amdgpu.ids | tail -8
