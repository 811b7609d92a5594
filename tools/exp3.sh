for a in "8 1" "8 0" "17 1"; do
echo "== prod $a"; timeout 120 python tools/dbg_gs2.py $a 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_solver.py -m gpu -x -q -k "smoothers or vcycle or iterates" 2>&1 | tail -2
HOT_TEST_CFG=gs_chain=1,gs_sub_block=32 timeout 900 python -m pytest tests/test_gpu_solver.py -m gpu -x -q -k "smoothers or vcycle or iterates" 2>&1 | tail -2
HOT_PROF_TOP=16 timeout 300 python tools/prof_table.py C2 2>&1 | grep -v "^$" | head -17
HOT_SOAK_CFG= timeout 600 python tools/soak.py C2 16 2>&1 | grep "^step" | awk '{it+=$4; ms+=$NF; if (NR>4) {it2+=$4; ms2+=$NF}} END {printf "all: %.2f ms/iter  steps 4..: %.3f ms/iter %.1f ms/step\n", ms/it, ms2/it2, ms2/(NR-4)}'
