cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_env.py -m gpu -x -q -k "physics_substep_matches_oracle" 2>&1 | tail -60 | cut -c1-250 > gpurun_out/r04_fail.txt; cat gpurun_out/r04_fail.txt
