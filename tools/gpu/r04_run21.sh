cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/legged_gym_dev_amd/lib
timeout -k 10 600 python -m pytest tests/test_hip_env.py -m gpu -x -q 2>&1 | tail -3 | cut -c1-400 &&
{ LG_HIP_LIB=$L/liblegged_hip_prof.so python tools/substeps_sections_spread.py anymal_c_flat anymal_c_rough cassie 2>&1 | grep -v "Setting\|Warn\|self.setup\|amdgpu.ids"
LG_HIP_LIB=$L/liblegged_hip_prof_span.so python tools/substeps_span.py anymal_c_flat anymal_c_rough cassie 2>&1 | grep workgroups
python tools/env_step_time.py 2>&1 | grep lg_step; } > gpurun_out/r04_spread2.txt 2>&1; cat gpurun_out/r04_spread2.txt
