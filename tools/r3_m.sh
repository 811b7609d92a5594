mkdir -p gpurun_out/r03
timeout 3000 python -m pytest tests -q -m gpu -x --durations=8 > gpurun_out/r03/all_m.log 2>&1; echo "all rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03/all_m.log | tail -25
timeout 600 python bench.py --no-cpu > gpurun_out/r03/bench_C2_m.json 2> gpurun_out/r03/bench_C2_m.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r03/bench_C2_m.json')); print(d['value'], d['ms_per_step'], d['iterations_per_step'], d['transfers']); print(d['kernel_ms_per_step_top'])"
