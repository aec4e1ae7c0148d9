cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/legged_gym_dev_amd/lib
LG_HIP_LIB=$L/liblegged_hip_prof.so python tools/substeps_sections_spread.py anymal_c_flat anymal_c_rough 2>&1 | grep -v "Setting\|Warn\|self.setup\|amdgpu.ids" > gpurun_out/r04_spread.txt; cat gpurun_out/r04_spread.txt
