set -x
cd $GRAFT_REPO_ROOT
for B in gemm_glds gemm_glds_noslp; do
  echo "== $B" >> gpurun_out/r04_glds_c.txt
  timeout -k 10 200 tools/micro/_bin/$B 24576 256 512 2 >> gpurun_out/r04_glds_c.txt 2>&1
  timeout -k 10 200 tools/micro/_bin/$B 32768 256 512 1 >> gpurun_out/r04_glds_c.txt 2>&1
done
cat gpurun_out/r04_glds_c.txt
