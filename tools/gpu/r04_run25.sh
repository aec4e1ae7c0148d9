cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_ppo.py tests/test_hip_runner.py -m gpu -x -q 2>&1 | tail -2 | cut -c1-300 &&
bash tools/kernel_avg.sh k_head "A=1" > gpurun_out/r04_head.txt 2>&1; cat gpurun_out/r04_head.txt
LG_HIP_LIB=$GRAFT_REPO_ROOT/legged_gym_dev_amd/lib/liblegged_hip_exp.so timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_alt --no_other --sustained 0 2>&1 | grep "head wg" | tail -8 > gpurun_out/r04_head_stamps.txt; cat gpurun_out/r04_head_stamps.txt
