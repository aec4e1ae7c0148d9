"""env-steps/s of the other BASELINE.json configurations on one MI355X (not the bench line: those are parity
cases; the numbers are kept in profiles/ for orientation).

    python tools/bench_configs.py [anymal_c_rough cassie anymal_c_flat] [--hidden 512,256,128] [--num_envs 4096]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from legged_gym_dev_amd.envs import task_registry
from legged_gym_dev_amd.utils import get_args

ap = argparse.ArgumentParser()
ap.add_argument("tasks", nargs="*", default=["anymal_c_rough", "cassie", "anymal_c_flat"])
ap.add_argument("--hidden", default="512,256,128")
ap.add_argument("--num_envs", type=int, default=4096)
ap.add_argument("--iters", type=int, default=6)
a = ap.parse_args()
hidden = [int(v) for v in a.hidden.split(",")]
for task in a.tasks:
    args = get_args(["--task", task, "--num_envs", str(a.num_envs), "--headless"])
    args.sim_device = args.rl_device = "cuda:0"
    env_cfg, train_cfg = task_registry.get_cfgs(task)
    train_cfg.policy.actor_hidden_dims = list(hidden)
    train_cfg.policy.critic_hidden_dims = list(hidden)
    train_cfg.runner.resume = False
    torch.manual_seed(1)
    np.random.seed(1)
    env, _ = task_registry.make_env(name=task, args=args, env_cfg=env_cfg)
    runner, _ = task_registry.make_alg_runner(env=env, name=task, args=args, train_cfg=train_cfg, log_root=None)
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
    for _ in range(3):
        runner.rollout(); runner.ppo.update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr = 0.0
    for _ in range(a.iters):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        runner.rollout()
        torch.cuda.synchronize()
        tr += time.perf_counter() - t1
        runner.ppo.update()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    steps = runner.num_steps_per_env * env.num_envs * a.iters
    bad = int((~torch.isfinite(env.obs_buf).all(1)).sum())
    print(f"{task:16s} envs {env.num_envs} obs {env.num_obs} hidden {hidden}: {steps / el:12.0f} env-steps/s  "
          f"{1e3 * el / a.iters:7.2f} ms/iter (rollout {1e3 * tr / a.iters:6.2f})  mean terrain level "
          f"{float(env.terrain_levels.float().mean()) if hasattr(env, 'terrain_levels') else 0:.2f}  non-finite obs rows {bad}", flush=True)
    env.close(); runner.ppo.close()
