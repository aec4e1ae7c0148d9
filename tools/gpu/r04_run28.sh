cd $GRAFT_REPO_ROOT
F="amdgpu.ids\|Setting seed\|UserWarning\|self.setup"
timeout -k 10 200 python tools/train_sanity.py 300 512,256,128 walk anymal_c 2>&1 | grep -v "$F" > gpurun_out/r04_learn_flat_walk.log; tail -2 gpurun_out/r04_learn_flat_walk.log
timeout -k 10 300 python tools/train_sanity.py 600 512,256,128 walk anymal_c_rough 2>&1 | grep -v "$F" > gpurun_out/r04_learn_rough.log; tail -2 gpurun_out/r04_learn_rough.log
timeout -k 10 200 python tools/train_sanity.py 200 512,256,128 curriculum anymal_c_trajectory 2>&1 | grep -v "$F" > gpurun_out/r04_learn_traj_curriculum.log; tail -3 gpurun_out/r04_learn_traj_curriculum.log
timeout -k 10 300 python legged_gym_dev_amd/scripts/train.py --task=cassie --headless --max_iterations 400 2>&1 | grep "Learning iteration\|Mean reward\|Mean episode length\|terrain_level\|fault\|steps/s" | paste - - - - - | awk 'NR%50==1' > gpurun_out/r04_learn_cassie.log; tail -3 gpurun_out/r04_learn_cassie.log | cut -c1-300
