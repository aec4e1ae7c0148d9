cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests_h.log 2>&1; tail -3 gpurun_out/r04_gpu_tests_h.log | cut -c1-300
timeout -k 10 600 python bench.py > gpurun_out/r04_bench_h.json 2> gpurun_out/r04_bench_h.err; tail -c 3000 gpurun_out/r04_bench_h.json
