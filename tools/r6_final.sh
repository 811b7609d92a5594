#!/bin/bash
# round 6: the evidence set of the final tree — GPU suite + smoke, driver-shaped bench with kernel stats and PMC traffic, the other configurations, SQ counters,
# GPU idle time, functional multi-rank lines.  Everything lands under gpurun_out/r06/ (copied into profiles/ afterwards).
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 3000 python -m pytest tests -q -m gpu --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "amdgpu.ids" $O/pytest_gpu.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; grep -v amdgpu $O/smoke.log | tail -4
bash profiles/run_profiles.sh r06 > $O/run_profiles.log 2>&1; tail -3 $O/run_profiles.log
bash profiles/run_config_profiles.sh r06 C1 C3 C4 C5 > $O/run_config.log 2>&1; grep "bench rc" $O/run_config.log
bash profiles/run_sq_counters.sh r06 C2 > $O/sq.log 2>&1; tail -3 $O/sq.log
R=$(pwd); (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1; python $R/tools/gpu_idle.py /tmp/kt > $R/$O/gpu_idle_C2.txt 2>&1); tail -4 $O/gpu_idle_C2.txt
timeout 900 python bench.py --share-gpu --gpus 8 --steps 2 --warmup 1 --no-cpu --cells 32 --backend gloo 2> $O/bench_8r.err | tail -1 > $O/bench_eight_ranks_one_gpu_functional.json; echo "8 ranks rc=$?"
timeout 900 python bench.py --share-gpu --gpus 2 --steps 2 --warmup 1 --no-cpu --cells 40 --backend gloo 2> $O/bench_2r.err | tail -1 > $O/bench_two_ranks_one_gpu_functional.json; echo "2 ranks rc=$?"
timeout 600 python tools/soak.py C2 40 > $O/soak_C2.txt 2>&1; echo "soak rc=$?"; tail -1 $O/soak_C2.txt | cut -c1-300
timeout 900 python tools/fuzz_parity.py 24 6 > $O/fuzz_parity.txt 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz_parity.txt
du -sh $O
