#!/bin/bash
mkdir -p gpurun_out/r6b
O=gpurun_out/r6b
timeout 600 python tools/r6_dbg.py > $O/dbg.txt 2>&1; grep -v amdgpu.ids $O/dbg.txt | tail -40
timeout 600 python -m pytest tests/test_gpu_transfer.py tests/test_gpu_golden.py -x -q -m gpu > $O/t_transfer.log 2>&1; echo "transfer rc=$?"; tail -5 $O/t_transfer.log
timeout 1500 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "PAIR or SUBST_D or P2G" > $O/t_variants.log 2>&1; echo "variants rc=$?"; tail -5 $O/t_variants.log
export HOT_PROF_TOP=40
timeout 300 python tools/prof_table.py C2 > $O/prof_prod.txt 2>&1; grep -E "wall|p2g|g2p" $O/prof_prod.txt
timeout 300 python tools/p2g_time.py > $O/p2g_time.txt 2>&1; tail -12 $O/p2g_time.txt
