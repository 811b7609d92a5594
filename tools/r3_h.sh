mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_gpu_solver.py tests/test_gpu_force.py tests/test_gpu_transfer.py tests/test_gpu_golden.py -q -m gpu -s -k "fp32 or objective_pieces or transfer or golden" > gpurun_out/r03/t_h.log 2>&1; echo "rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03/t_h.log | grep -E "fp32 step|passed|failed|FAILED|assert|Error" | head -40
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "fixed_iterations" > gpurun_out/r03/t_h2.log 2>&1; echo "rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03/t_h2.log | grep -E "rel dv|passed|failed|FAILED|assert" | head
timeout 1200 python -m pytest tests/test_gpu_multirank.py -q -m gpu -x > gpurun_out/r03/t_h3.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r03/t_h3.log
