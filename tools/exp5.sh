HOT_FUZZ_CFG=ls_energy_only=1 timeout 1200 python tools/fuzz_parity.py 14 11 2>&1 | grep -v "^ok" | cut -c1-250
