mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests/test_gpu_multirank.py -q -m gpu -s --durations=8 > gpurun_out/r03/mr_e.log 2>&1; echo "mr rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03/mr_e.log | grep -E "passed|failed|FAILED|Error|comm bytes|C2-size|assert" | head -40
timeout 600 python tools/spmv_ab.py C2 C3 > gpurun_out/r03/spmv_ab.log 2>&1; grep -v amdgpu.ids gpurun_out/r03/spmv_ab.log | tail -12
timeout 900 python tools/fp32_truth_probe.py > gpurun_out/r03/fp32_truth.log 2>&1; grep -v amdgpu.ids gpurun_out/r03/fp32_truth.log | tail -30
