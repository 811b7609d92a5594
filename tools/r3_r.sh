rm -rf gpurun_out/r03b
bash profiles/run_profiles.sh r03b > gpurun_out/run_profiles.log 2>&1; tail -3 gpurun_out/run_profiles.log | cut -c1-300
bash profiles/run_sq_counters.sh r03b C2 > gpurun_out/run_sq.log 2>&1; tail -12 gpurun_out/run_sq.log | cut -c1-400
bash profiles/run_config_profiles.sh r03b C1 C3 C4 C5 > gpurun_out/run_cfg.log 2>&1; grep "bench rc" gpurun_out/run_cfg.log
du -sh gpurun_out/r03b
