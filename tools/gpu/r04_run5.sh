set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests_a.log 2>&1; tail -15 gpurun_out/r04_gpu_tests_a.log
timeout -k 10 200 python tools/diag_faults.py 300 2>&1 | grep -v "amdgpu.ids\|Setting seed\|Warning\|warn" > gpurun_out/r04_diag_faults.txt; cat gpurun_out/r04_diag_faults.txt
