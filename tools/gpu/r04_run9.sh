set -x
cd $GRAFT_REPO_ROOT
LG_CLOCK_JSON=$GRAFT_REPO_ROOT/gpurun_out/r04_substeps_clock.json LG_HIP_LIB=$GRAFT_REPO_ROOT/legged_gym_dev_amd/lib/liblegged_hip_prof.so python tools/substeps_sections.py anymal_c_flat 2>&1 | grep -v "amdgpu.ids\|Setting seed\|Warning\|warn" > gpurun_out/r04_substeps_sections.txt; cat gpurun_out/r04_substeps_sections.txt
python tools/env_step_time.py 2>&1 | grep -v "amdgpu.ids\|Setting seed\|Warning\|warn" > gpurun_out/r04_env_step_time.txt; cat gpurun_out/r04_env_step_time.txt
timeout -k 10 600 python bench.py > gpurun_out/r04_bench_n1_a.json 2> gpurun_out/r04_bench_n1_a.err; cat gpurun_out/r04_bench_n1_a.json | cut -c1-3000
