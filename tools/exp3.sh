HOT_SOAK_CFG=gs_one_stream=0 timeout 300 python tools/soak.py C2 8 2>&1 | tail -3
HOT_SOAK_CFG=gs_one_stream=1 timeout 300 python tools/soak.py C2 8 2>&1 | tail -3
HOT_AMD_AB=1 HOT_GS_V1=1 timeout 300 python tools/soak.py C2 8 2>&1 | tail -3
