set -x
cd $GRAFT_REPO_ROOT
B=tools/micro/_bin/gemm_glds
timeout -k 10 200 $B 24576 256 512 2 > gpurun_out/r04_glds_b.txt 2>&1
timeout -k 10 200 $B 32768 256 512 1 >> gpurun_out/r04_glds_b.txt 2>&1
cat gpurun_out/r04_glds_b.txt
