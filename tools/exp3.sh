for a in "8 1" "17 1"; do
echo "== prod $a"; timeout 120 python tools/dbg_gs2.py $a 2>&1 | tail -1
done
HOT_PROF_TOP=9 timeout 300 python tools/prof_table.py C2 2>&1 | grep -v "^$" | tail -10
timeout 300 python tools/soak.py C2 8 2>&1 | tail -3
