mkdir -p gpurun_out/r03
timeout 900 python bench.py > gpurun_out/r03/bench_final.json 2> gpurun_out/r03/bench_final.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r03/bench_final.json')); print(d['value'], d['ms_per_step'], d['iterations_per_step'], d['transfers']['frac_of_hbm_peak'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['fair_value'])"
timeout 3000 python -m pytest tests -q -m gpu --durations=5 > gpurun_out/r03/all_final.log 2>&1; echo "all rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03/all_final.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -5
