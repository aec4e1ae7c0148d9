set -x
cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|Setting seed\|Warning\|warn\|EnvSetup('
echo "# control loop with the W phase computing 1 of its 3 test impulses (make prof PROFFLAGS=-DLG_EXP_W_DIRS=1: wrong contact law, timing only)" > gpurun_out/r04_substeps_w_bound.txt
LG_HIP_LIB=$GRAFT_REPO_ROOT/legged_gym_dev_amd/lib/liblegged_hip_prof_w1.so python tools/substeps_sections.py anymal_c_flat 2>&1 | grep -v "$F" >> gpurun_out/r04_substeps_w_bound.txt
echo "# the same build with all three (make prof)" >> gpurun_out/r04_substeps_w_bound.txt
LG_HIP_LIB=$GRAFT_REPO_ROOT/legged_gym_dev_amd/lib/liblegged_hip_prof.so python tools/substeps_sections.py anymal_c_flat 2>&1 | grep -v "$F" >> gpurun_out/r04_substeps_w_bound.txt
cat gpurun_out/r04_substeps_w_bound.txt
