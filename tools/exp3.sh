timeout 900 python -m pytest tests/test_gpu_solver.py -m gpu -x -q -k "smoothers or vcycle or iterates or convergence or knobs" 2>&1 | tail -4
HOT_PROF_TOP=30 timeout 300 python tools/prof_table.py C2 2>&1 | grep -E "wall|cg_|spmv_L2|copy|dot|diag_scale"
for cfg in ""; do
HOT_SOAK_CFG=$cfg timeout 600 python tools/soak.py C2 16 2>&1 | grep "^step" | awk '{it+=$4; ms+=$NF; if (NR>4) {it2+=$4; ms2+=$NF}} END {printf "all: %.2f ms/iter  steps 4..: %.3f ms/iter %.1f ms/step\n", ms/it, ms2/it2, ms2/(NR-4)}'
done
timeout 300 python tools/soak.py C1 4 2>&1 | tail -1
