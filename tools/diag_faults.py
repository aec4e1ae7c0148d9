"""What happens to runaway robots during a long run of the bench's own loop (anymal_c_flat as committed in the fork: its reward is
identically 0 after the positive clip, so PPO's entropy bonus is the only gradient on the policy's std)?
    python tools/diag_faults.py [ITERS] [TASK]
Every 25 iterations: non-finite solves stopped by the fault guard (fault_total), base-velocity clamps at asset.max_*_velocity
(vel_clamp_total), policy steps that END above round 3's guard threshold (|v_base|^2 + |w_base|^2 >= 2e4, i.e. ~141 rad/s: until
round 3 such an env was reset instead of clamped), resets, the policy's mean std, |action| statistics, base-velocity and joint-rate
maxima over the iteration's rollout."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
task = sys.argv[2] if len(sys.argv) > 2 else "anymal_c_flat"
env, runner = bench.make_runner(4096, [512, 256, 128], "cuda:0", 0, 1, task=task)
t = env.core.t
prev = 0
over = torch.zeros((), dtype=torch.int64, device="cuda")          # policy steps ending above the old threshold
wmax = torch.zeros((), device="cuda")
_step = env.step


def step(actions):                                                # per-step hook around the product env's step (device-side, no sync)
    out = _step(actions)
    tw = t["root_states"][:, 7:13]
    over.add_((tw.pow(2).sum(1) >= 2.0e4).sum())
    torch.maximum(wmax, t["root_states"][:, 10:13].norm(dim=1).max(), out=wmax)
    return out


env.step = step
for it in range(1, iters + 1):
    runner.rollout()
    runner.ppo.update()
    if it % 25 == 0 or it == 1:
        torch.cuda.synchronize()
        ft = int(t["fault_total"][0])
        a = runner.ppo.t["actions"]
        print(f"it {it:4d} faults {ft:7d} (+{ft - prev:5d}) clamps {int(t['vel_clamp_total'][0]):6d} over_old_thr {int(over):7d} |w| max so far {float(wmax):6.1f} ep_done {int(runner.ppo.t['ep_ring_count'].cpu()) & 0xFFFFFFFF:8d} "
              f"std {float(runner.ppo.param_views['std'].mean()):6.3f} |a| mean {float(a.abs().mean()):6.2f} max {float(a.abs().max()):7.1f} "
              f"|v_base| max {float(t['root_states'][:, 7:10].norm(dim=1).max()):6.1f} |w_base| max {float(t['root_states'][:, 10:13].norm(dim=1).max()):6.1f} "
              f"|qd| max {float(t['dof_state'][..., 1].abs().max()):5.1f} z min {float(t['root_states'][:, 2].min()):6.2f} lr {runner.ppo.learning_rate:.2e}", flush=True)
        prev = ft
