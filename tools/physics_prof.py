"""Target of the physics PMC passes: 4096 ANYmal-C envs standing on the plane; 10 x lg_simulate (physics stage only),
10 x lg_compute_torques (actuator-net stage only), 10 x lg_step.  Run under rocprofv3 --pmc ... (see profiles/r02_README.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

task = sys.argv[1] if len(sys.argv) > 1 else "flat"
if task == "flat":
    env, runner = bench.make_runner(4096, [128, 64, 32], "cuda:0", 0, 1)
else:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from tests.test_hip_env import _product_env
    env = _product_env(task, 4096, terrain=None)
    env.reset()
a = torch.zeros(4096, env.num_actions, device="cuda")
for _ in range(3):
    env.step(a)
torch.cuda.synchronize()
for name in ("simulate", "compute_torques"):
    for _ in range(10):
        env.core.call(name)
    torch.cuda.synchronize()
for _ in range(10):
    env.core.step(a)
torch.cuda.synchronize()
