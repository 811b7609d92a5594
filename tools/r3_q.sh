mkdir -p gpurun_out/r03
HOT_LIB=hot_amd/csrc/libhotmi355x_clk.so timeout 600 python tools/hess_time.py C2 C3 2>&1 | grep -v amdgpu.ids | awk '!seen[$0]++' | tee gpurun_out/r03/hess_exp.log
timeout 600 python tools/hess_time.py C2 C3 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1200 python -m pytest tests/test_gpu_force.py tests/test_gpu_solver.py tests/test_gpu_variants.py tests/test_gpu_golden.py -q -x 2>&1 | grep -v amdgpu.ids | tail -4
