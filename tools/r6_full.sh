#!/bin/bash
# whole GPU suite + smoke + the driver-shaped bench line
mkdir -p gpurun_out/r6full
O=gpurun_out/r6full
timeout 3000 python -m pytest tests -q -m gpu -x --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "amdgpu.ids" $O/pytest_gpu.log | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -5
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6full/bench.json'))
print('ms/iter', d['value'], 'ms/step', d['ms_per_step'], 'it/step', d.get('iterations_per_step'), 'build', d.get('hessian_mg_build_ms_per_step'))
print('top', d.get('kernel_ms_per_step_top'))
r=d['roofline']; print('roof', r['kernel'], r['achieved'], r['frac'], r.get('traffic'), r['avg_launch_ms'], r['algorithmic_bytes_per_launch'], r['peak_measured'])
print('xfer', d['transfers'])
print('cpu', d['cpu_baseline'])
PY
