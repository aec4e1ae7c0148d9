cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_env.py -m gpu -x -q 2>&1 | tail -3 | cut -c1-300 &&
for cfg in "4 0" "41 41" "4 0" "41 41"; do set -- $cfg; echo "NW=$1 PWSEL=$2"; LG_SUBSTEPS_NW=$1 LG_SUBSTEPS_PWSEL=$2 timeout -k 10 120 python tools/env_step_time.py 2>&1 | grep lg_step; done > gpurun_out/r04_nw41.txt 2>&1; cat gpurun_out/r04_nw41.txt
