set -x
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_quant_glds -- python3 $GRAFT_REPO_ROOT/tools/gemm_quant.py run > /dev/null 2>&1
export LG_GEMM_GLDS=0
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_quant_reg -- python3 $GRAFT_REPO_ROOT/tools/gemm_quant.py run > /dev/null 2>&1
unset LG_GEMM_GLDS
cd $GRAFT_REPO_ROOT
python tools/gemm_quant.py digest gpurun_out/r04_quant_glds > gpurun_out/r04_gemm_quant.txt
python tools/gemm_quant.py digest gpurun_out/r04_quant_reg >> gpurun_out/r04_gemm_quant.txt
cat gpurun_out/r04_gemm_quant.txt
find gpurun_out/r04_quant_glds gpurun_out/r04_quant_reg -type f ! -name "*kernel_trace.csv" -delete
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests_d.log 2>&1; tail -3 gpurun_out/r04_gpu_tests_d.log | cut -c1-200
