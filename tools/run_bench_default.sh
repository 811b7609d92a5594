#!/bin/bash
# The driver's default bench line (C2, one GPU, with the CPU baseline) and a functional two-rank run on the same GPU, into gpurun_out/
mkdir -p gpurun_out
if [ -z "$SKIP_DEFAULT" ]; then
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['ms_per_step'], d['iterations_per_step'], d['transfers']['frac_of_hbm_peak'], d['transfers']['per_step_ms'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['fair_value'])"
fi
timeout 900 python bench.py --share-gpu --gpus 2 --steps 2 --warmup 1 --no-cpu --cells 40 2> gpurun_out/bench_2r.err | tail -1 > gpurun_out/bench_2r.json
python -c "
import json; d=json.load(open('gpurun_out/bench_2r.json')); print('2 ranks on one GPU:', d['value'], d['ms_per_step'], d['n_gpus'], d['config']['parallelism'][:80], d['comm_per_step_rank0'])"
