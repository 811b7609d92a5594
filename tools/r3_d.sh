mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -s --durations=8 > gpurun_out/r03/mr_d.log 2>&1; echo "mr rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03/mr_d.log | tail -60
