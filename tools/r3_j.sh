mkdir -p gpurun_out/r03
timeout 900 python bench.py > gpurun_out/r03/bench_C2_j.json 2> gpurun_out/r03/bench_C2_j.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r03/bench_C2_j.json')); print(d['value'], d['ms_per_step'], d['iterations_per_step']); c=d['cpu_baseline']; print({k:c[k] for k in ('value','fair_value','cores','omp_threads','oracle_rebuilt_on_this_host','gpu_same_step_ms_per_iter','speedup_per_iteration_vs_faithful','cpu_build_ms','cpu_fair_build_ms','sample')})"
tail -3 gpurun_out/r03/bench_C2_j.err
timeout 900 python bench.py --gpus 2 --share-gpu --backend gloo --comm torch --cells 40 --steps 2 --no-cpu > gpurun_out/r03/bench_2r.json 2> gpurun_out/r03/bench_2r.err; echo "2r rc=$?"; tail -c 1500 gpurun_out/r03/bench_2r.json; tail -3 gpurun_out/r03/bench_2r.err
timeout 900 python bench.py --gpus 2 --share-gpu --backend gloo --comm torch --cells 50 --steps 2 --no-cpu --shard-gs 0 > gpurun_out/r03/bench_2r_cs.json 2> gpurun_out/r03/bench_2r_cs.err; echo "2r rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r03/bench_2r_cs.json')); print(d['ms_per_step'], d['iterations_per_step'], d['comm_per_step_rank0'], d['comm_calls_per_step'])"
