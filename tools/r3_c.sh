mkdir -p gpurun_out/r03
timeout 900 python tools/fp32_probe.py > gpurun_out/r03/fp32_probe2.log 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids gpurun_out/r03/fp32_probe2.log
timeout 1500 python -m pytest tests/test_gpu_solver.py tests/test_gpu_force.py tests/test_gpu_transfer.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/r03/t_c.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r03/t_c.log
timeout 600 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "LBFGS or FORCE or P2G" > gpurun_out/r03/t_c2.log 2>&1; echo "variants rc=$?"; tail -5 gpurun_out/r03/t_c2.log
timeout 600 python bench.py --no-cpu > gpurun_out/r03/bench_C2_c.json 2> gpurun_out/r03/bench_C2_c.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r03/bench_C2_c.json')); print(d['value'], d['ms_per_step'], d['iterations_per_step'], d['transfers']); print(d['kernel_ms_per_step_top'])"
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/r03/counters.txt); wc -l gpurun_out/r03/counters.txt
