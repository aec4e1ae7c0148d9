cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r04_gpu_tests.log 2>&1; tail -2 gpurun_out/r04_gpu_tests.log | cut -c1-200
