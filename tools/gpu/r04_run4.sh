set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_glds_d.txt
B=tools/micro/_bin/gemm_glds
timeout -k 5 90 $B 24576 256 512 2 > $O 2>&1
timeout -k 5 90 $B 32768 256 512 1 >> $O 2>&1
timeout -k 5 90 $B 24576 256 512 2 -1 1 >> $O 2>&1
timeout -k 5 90 $B 24576 128 256 2 >> $O 2>&1
timeout -k 5 90 $B 24576 512 64 2 >> $O 2>&1
cat $O
