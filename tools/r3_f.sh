mkdir -p gpurun_out/r03
timeout 900 python tools/fp32_truth_probe.py > gpurun_out/r03/fp32_truth2.log 2>&1; grep -v amdgpu.ids gpurun_out/r03/fp32_truth2.log | tail -30
timeout 2400 python -m pytest tests/test_gpu_multirank.py -q -m gpu -x > gpurun_out/r03/mr_f.log 2>&1; echo "mr rc=$?"; tail -5 gpurun_out/r03/mr_f.log
