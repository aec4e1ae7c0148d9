cd $GRAFT_REPO_ROOT
for cfg in "4 0" "41 0" "41 32" "41 40" "41 41" "41 312" "41 313" "41 568" "41 569" "41 824" "41 825" "41 1080" "41 1081" "4 0"; do set -- $cfg; echo "NW=$1 PWSEL=$2"; LG_SUBSTEPS_NW=$1 LG_SUBSTEPS_PWSEL=$2 timeout -k 10 120 python tools/env_step_time.py anymal_c_flat 2>&1 | grep lg_step; done > gpurun_out/r04_nw41.txt 2>&1; cat gpurun_out/r04_nw41.txt
