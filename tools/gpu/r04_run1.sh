set -x
cd $GRAFT_REPO_ROOT
B=tools/micro/_bin/gemm_glds
timeout -k 10 120 $B 24576 256 512 2 > gpurun_out/r04_glds_a.txt 2>&1
timeout -k 10 120 $B 32768 256 512 1 >> gpurun_out/r04_glds_a.txt 2>&1
timeout -k 10 120 $B 65536 256 512 1 >> gpurun_out/r04_glds_a.txt 2>&1
cat gpurun_out/r04_glds_a.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_prof0 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_alt --no_other --sustained 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py gpurun_out/r04_prof0 > gpurun_out/r04_timeline0.txt; cat gpurun_out/r04_timeline0.txt
find gpurun_out/r04_prof0 -type f ! -name "*.csv" -delete
