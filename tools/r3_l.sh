bash profiles/run_profiles.sh r03 2>&1 | tail -5
bash profiles/run_config_profiles.sh r03 C1 C3 C4 C5 2>&1 | tail -50
