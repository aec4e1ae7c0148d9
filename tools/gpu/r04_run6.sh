set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests_b.log 2>&1; tail -8 gpurun_out/r04_gpu_tests_b.log | cut -c1-200
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_alt --no_other --sustained 0 > gpurun_out/r04_bench_quick.json 2> gpurun_out/r04_bench_quick.err; cat gpurun_out/r04_bench_quick.json | cut -c1-1500
LG_GEMM_GLDS=0 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_alt --no_other --sustained 0 > gpurun_out/r04_bench_quick_noglds.json 2>> gpurun_out/r04_bench_quick.err; cat gpurun_out/r04_bench_quick_noglds.json | cut -c1-900
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_alt --no_other --sustained 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py gpurun_out/r04_prof1 > gpurun_out/r04_timeline1.txt; cat gpurun_out/r04_timeline1.txt
find gpurun_out/r04_prof1 -type f ! -name "*.csv" -delete
