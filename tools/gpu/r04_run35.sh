cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tools/diag_faults.py 300 2>&1 | grep "^it " | awk 'NR%4==0' > gpurun_out/r04_diag_faults_final.txt; cat gpurun_out/r04_diag_faults_final.txt | cut -c1-250
