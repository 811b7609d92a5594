HOT_PROF_TOP=8 timeout 300 python tools/prof_table.py C2 2>&1 | grep -v "^$" | head -7
timeout 120 python tools/dbg_gs2.py 17 1 2>&1 | tail -1
