#!/bin/bash
# The whole GPU suite + smoke on a GPU box, logs under gpurun_out/: gpurun -- 'bash tools/run_gpu_suite.sh'
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu --durations=5 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "amdgpu.ids" gpurun_out/pytest_gpu.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -5
