#!/bin/bash
# round 6: persistent PCG without a host round trip per V-cycle — parity (solver, variants subset, fullsize subset), then C1 / C2 step tables
mkdir -p gpurun_out/cgasync
O=gpurun_out/cgasync
timeout 1500 python -m pytest tests/test_gpu_solver.py tests/test_gpu_golden.py -x -q -m gpu > $O/t1.log 2>&1; echo "solver/golden rc=$?"; tail -2 $O/t1.log
timeout 2400 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "CG or cg or TIMEOUT or timeout" > $O/t2.log 2>&1; echo "variants rc=$?"; tail -2 $O/t2.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "not C5 and not C4" > $O/t3.log 2>&1; echo "fullsize rc=$?"; tail -2 $O/t3.log
HOT_PROF_TOP=3 timeout 300 python tools/prof_table.py C1 > $O/prof_C1.txt 2>&1; head -3 $O/prof_C1.txt
HOT_PROF_TOP=3 timeout 300 python tools/prof_table.py C2 > $O/prof_C2.txt 2>&1; head -2 $O/prof_C2.txt
timeout 300 python bench.py --config C1 --steps 4 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C1 bench', d['value'], d['ms_per_step'], d['iterations_per_step'])"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2 bench', d['value'], d['ms_per_step'], d['iterations_per_step'])"
