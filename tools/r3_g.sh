mkdir -p gpurun_out/r03
timeout 900 python tools/fp32_truth_probe.py > gpurun_out/r03/fp32_truth3.log 2>&1; grep -v amdgpu.ids gpurun_out/r03/fp32_truth3.log | tail -30
timeout 600 python tools/fp32_probe.py > gpurun_out/r03/fp32_probe3.log 2>&1; grep -v amdgpu.ids gpurun_out/r03/fp32_probe3.log | tail -30
timeout 3000 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r03/all_g.log 2>&1; echo "all rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03/all_g.log | tail -40
