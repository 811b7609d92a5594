bash tools/r3_j.sh
bash profiles/run_sq_counters.sh r03 C2 2>&1 | tail -30
