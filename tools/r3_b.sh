mkdir -p gpurun_out/r03
timeout 1500 python tools/fp32_probe.py > gpurun_out/r03/fp32_probe.log 2>&1; echo "probe rc=$?"; cat gpurun_out/r03/fp32_probe.log | grep -v amdgpu.ids
timeout 2400 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "C4_full or C5_full" --durations=10 > gpurun_out/r03/full_tests.log 2>&1; echo "full rc=$?"; tail -30 gpurun_out/r03/full_tests.log
timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "c2_size" --durations=5 > gpurun_out/r03/mr_c2.log 2>&1; echo "mr rc=$?"; tail -30 gpurun_out/r03/mr_c2.log
