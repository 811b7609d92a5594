for cfg in "" "gs_chain=1,gs_sub_block=32" "gs_sub_block=64"; do
echo "== cfg '$cfg'"; HOT_SOAK_CFG=$cfg timeout 600 python tools/soak.py C2 16 2>&1 | grep "^step" | awk '{it+=$4; ms+=$NF; if (NR>4) {it2+=$4; ms2+=$NF}} END {printf "all: %.2f ms/iter  steps 4..: %.3f ms/iter %.1f ms/step\n", ms/it, ms2/it2, ms2/(NR-4)}'
done
