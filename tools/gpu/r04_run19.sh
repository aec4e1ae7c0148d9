cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/legged_gym_dev_amd/lib
{ LG_HIP_LIB=$L/liblegged_hip_prof_span.so python tools/substeps_span.py anymal_c_flat anymal_c_rough cassie 2>&1 | grep workgroups
LG_SUBSTEPS_NW=41 LG_SUBSTEPS_PWSEL=41 LG_HIP_LIB=$L/liblegged_hip_prof_span.so python tools/substeps_span.py anymal_c_flat 2>&1 | grep workgroups
LG_SPAN_ENVS=1024 LG_HIP_LIB=$L/liblegged_hip_prof_span.so python tools/substeps_span.py anymal_c_flat 2>&1 | grep workgroups
LG_HIP_LIB=$L/liblegged_hip_prof.so python tools/substeps_sections.py anymal_c_flat 2>&1 | grep -v Setting ; } > gpurun_out/r04_span.txt 2>&1; cat gpurun_out/r04_span.txt
