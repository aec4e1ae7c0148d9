"""A few ppo.act calls (fused rollout forward) -- target for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
env, runner = bench.make_runner(4096, [512, 256, 128], "cuda:0", 0, 1)
obs = env.get_observations()
for _ in range(6):
    runner.ppo._call("end_update"); runner.ppo.act(obs, None)
torch.cuda.synchronize()
print("done")
