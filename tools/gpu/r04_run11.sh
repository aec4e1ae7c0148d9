set -x
cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|Setting seed\|Warning\|warn\|EnvSetup('
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests_e.log 2>&1; tail -3 gpurun_out/r04_gpu_tests_e.log | cut -c1-200
timeout -k 10 200 python tools/diag_faults.py 300 2>&1 | grep -v "$F" > gpurun_out/r04_diag_faults_b.txt; tail -2 gpurun_out/r04_diag_faults_b.txt | cut -c1-250
LG_CLOCK_JSON=$GRAFT_REPO_ROOT/gpurun_out/r04_substeps_clock.json LG_HIP_LIB=$GRAFT_REPO_ROOT/legged_gym_dev_amd/lib/liblegged_hip_prof.so python tools/substeps_sections.py anymal_c_flat 2>&1 | grep -v "$F" > gpurun_out/r04_substeps_sections.txt; cat gpurun_out/r04_substeps_sections.txt
timeout -k 10 200 python tools/train_sanity.py 300 512,256,128 walk anymal_c 2>&1 | grep -v "$F" > gpurun_out/r04_learn_flat_walk.log; tail -2 gpurun_out/r04_learn_flat_walk.log
python tools/env_step_time.py 2>&1 | grep -v "$F" > gpurun_out/r04_env_step_time.txt; cat gpurun_out/r04_env_step_time.txt
