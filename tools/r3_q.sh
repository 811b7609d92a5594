HOT_COLD=1 timeout 600 python tools/p2g_time.py C2 C3 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/p2g_time.py C2 2>&1 | grep -v amdgpu.ids | tail -1
