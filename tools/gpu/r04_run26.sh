cd $GRAFT_REPO_ROOT
LG_HIP_LIB=$GRAFT_REPO_ROOT/legged_gym_dev_amd/lib/liblegged_hip_exp.so timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_alt --no_other --sustained 0 2>&1 | grep "head wg" | tail -24 > gpurun_out/r04_head_stamps.txt; cat gpurun_out/r04_head_stamps.txt
