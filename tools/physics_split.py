"""How the physics substep's time splits: articulated-body pass alone (robots lifted clear of the
ground: every contact phase is skipped wave-uniformly) vs. with feet on the ground."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

env, runner = bench.make_runner(4096, [128, 64, 32], "cuda:0", 0, 1)
a = torch.zeros(4096, 12, device="cuda")
def run(tag, lift):
    env.reset()
    env.root_states[:, 2] += lift
    env.core.call("compute_torques")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        env.core.call("simulate")
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag:34s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per physics substep", flush=True)
for _ in range(2):
    run("standing (feet in contact)", 0.0)
    run("lifted 10 m (no contact phases)", 10.0)
