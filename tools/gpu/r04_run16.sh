set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests_g.log 2>&1; tail -3 gpurun_out/r04_gpu_tests_g.log | cut -c1-200
