set -x
cd $GRAFT_REPO_ROOT
export LG_HIP_LIB=$GRAFT_REPO_ROOT/legged_gym_dev_amd/lib/liblegged_hip_exp.so
echo "# k_gemm_glds (LDS-DMA forward)" > gpurun_out/r04_gemm_clock.txt
timeout -k 10 100 python tools/gemm_inkernel_clock.py 2>&1 | grep operands >> gpurun_out/r04_gemm_clock.txt
echo "# k_gemm (register-staged, LG_GEMM_GLDS=0)" >> gpurun_out/r04_gemm_clock.txt
LG_GEMM_GLDS=0 timeout -k 10 100 python tools/gemm_inkernel_clock.py 2>&1 | grep operands >> gpurun_out/r04_gemm_clock.txt
cat gpurun_out/r04_gemm_clock.txt
