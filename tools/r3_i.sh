mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_gpu_solver.py -q -m gpu -x -k "composite or collision or fp32" > gpurun_out/r03/t_i.log 2>&1; echo "rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03/t_i.log | tail -15
timeout 1200 python -m pytest tests/test_gpu_multirank.py -q -m gpu -x > gpurun_out/r03/t_i3.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r03/t_i3.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "fixed_iterations" > gpurun_out/r03/t_i2.log 2>&1; echo "rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03/t_i2.log | grep -E "rel dv|passed|failed|FAILED|assert" | head
