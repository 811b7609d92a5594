#!/bin/bash
# copies the batch tools/r6_final.sh left under gpurun_out/r06 into profiles/r06_* (the files the documents cite)
O=gpurun_out/r06
cp $O/bench.json profiles/r06_bench_c2.json; cp $O/bench_under_rocprof.json profiles/r06_bench_c2_under_rocprof.json
cp $O/kernel_stats.csv profiles/r06_kernel_stats.csv; cp $O/pmc_summary.json profiles/r06_pmc_summary.json
for c in C1 C3 C4 C5; do cp $O/bench_$c.json profiles/r06_bench_$c.json; cp $O/kernel_stats_$c.csv profiles/r06_kernel_stats_$c.csv; done
cp $O/sq_C2_summary.json profiles/r06_sq_counters_C2.json; cp $O/gpu_idle_C2.txt profiles/r06_gpu_idle_C2.txt
cp $O/bench_eight_ranks_one_gpu_functional.json profiles/r06_bench_eight_ranks_one_gpu_functional.json
cp $O/bench_two_ranks_one_gpu_functional.json profiles/r06_bench_two_ranks_one_gpu_functional.json
grep -v amdgpu.ids $O/pytest_gpu.log > profiles/r06_pytest_gpu.log; grep -v amdgpu.ids $O/smoke.log > profiles/r06_smoke.log
[ -f $O/soak_C2.txt ] && grep -v amdgpu.ids $O/soak_C2.txt > profiles/r06_soak_C2.txt
[ -f $O/fuzz_parity.txt ] && grep -v amdgpu.ids $O/fuzz_parity.txt > profiles/r06_fuzz_parity_seed6.txt
[ -f $O/fuzz_parity_seed7.txt ] && grep -v amdgpu.ids $O/fuzz_parity_seed7.txt > profiles/r06_fuzz_parity_seed7.txt
python - <<'P'
import json,csv
d=json.load(open('profiles/r06_bench_c2.json')); r=d['roofline']
print('C2 value %.3f ms/iter, %.2f ms/step, %.1f its, build %.2f' % (d['value'], d['ms_per_step'], d['iterations_per_step'], d['hessian_mg_build_ms_per_step']))
print('launch %.2f us frac %.3f traffic %.1f MB peak_measured %.2f torch %.2f frac_measured %.3f' % (1e3*r['avg_launch_ms'], r['frac'], r['traffic']/1e6, r['peak_measured']/1e3, r['peak_measured_torch']/1e3, r['frac_of_measured']))
rows=list(csv.DictReader(open('profiles/r06_kernel_stats.csv'))); tot=sum(float(x['TotalDurationNs']) for x in rows)
c=[x for x in rows if 'k_gs_colour' in x['Name']]; n=sum(int(x['Calls']) for x in c); t=sum(float(x['TotalDurationNs']) for x in c)
print('rocprofv3 colour: %d launches avg %.2f us share %.3f frac %.3f' % (n, t/n/1e3, t/tot, r['algorithmic_bytes_per_launch']/(t/n)/8000))
for x in rows:
    if 'k_gs_residual<double, true>' in x['Name']: print('residual', x['Calls'], float(x['AverageNs'])/1e3, float(x['TotalDurationNs'])/tot)
t=d['transfers']; print('p2g %.3f g2p %.3f Mp/s %.0f frac %.3f' % (t['p2g_ms'], t['g2p_ms'], t['mparticles_per_s'], t['frac_of_hbm_peak']))
c=d['cpu_baseline']; print('cpu %.0f fair %.0f gpu %.2f' % (c['value'], c['fair_value'], c['gpu_same_step_ms_per_iter']))
print(d['kernel_ms_per_step_top'])
for k in ['C1','C3','C4','C5']:
    e=json.load(open(f'profiles/r06_bench_{k}.json')); print(k, round(e['value'],3), round(e['ms_per_step'],1), e['iterations_per_step'], round(e['roofline']['frac'],3), e['roofline']['kernel'])
P
grep "^span" profiles/r06_gpu_idle_C2.txt; tail -1 profiles/r06_pytest_gpu.log
