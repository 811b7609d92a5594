mkdir -p gpurun_out/r03
timeout 900 python tools/fuzz_parity.py 120 7 > gpurun_out/r03/fuzz.log 2>&1; echo "fuzz rc=$?"; grep -v amdgpu.ids gpurun_out/r03/fuzz.log | tail -12
timeout 600 python tools/fuzz_parity.py 40 11 fp32_one > gpurun_out/r03/fuzz32.log 2>&1; echo "fuzz32 rc=$?"; grep -v amdgpu.ids gpurun_out/r03/fuzz32.log | tail -6
timeout 900 python tools/soak.py C2 60 > gpurun_out/r03/soak_C2.log 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r03/soak_C2.log
timeout 900 python tools/soak.py C3 15 > gpurun_out/r03/soak_C3.log 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r03/soak_C3.log
HOT_SOAK_CFG=coarseSolver=7 timeout 600 python tools/soak.py C2 6 > gpurun_out/r03/soak_C2_ic.log 2>&1; echo "soak ic rc=$?"; tail -2 gpurun_out/r03/soak_C2_ic.log
