"""Roll a trained (or freshly initialised) policy on a trajectory-tracking task and record, at the reduced-order model's rate,
what the tube-learning stage consumes: ROM state z, ROM input v, projection of the robot state pz_x, termination flags
[and the full robot state x].  Same loop and the same record layout as the reference's
deep_tube_learning/data_collection_trajectory.py:97-183 (one ``epoch_<k>.pickle`` per epoch with numpy arrays
z (N, T+1, n), v (N, T, m), pz_x (N, T+1, n), done (N, T) [, x (N, T+1, 7 + A + 6 + A)]), without the hydra / wandb plumbing:

    python legged_gym_dev_amd/scripts/collect_trajectory_data.py --task anymal_c_flat_trajectory --num_envs 4096 \\
        --epochs 2 --out rom_tracking_data/run0 [--load_run -1 --checkpoint -1] [--save_debugging_data]

The env steps entirely on the GPU; the host only watches the ROM step counters (one small device->host read per env step).
"""
import argparse
import json
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from legged_gym_dev_amd.envs import *  # noqa: E402,F401,F403
from legged_gym_dev_amd.utils import get_args, task_registry  # noqa: E402


def collect(env, policy, epochs, episode_length_s=None, save_debugging_data=False, out_dir=None, progress=False):
    """The reference's epoch loop (data_collection_trajectory.py:97-183).  Returns the list of epoch dicts."""
    rom, tg = env.rom, env.traj_gen
    N = env.num_envs
    T = int((episode_length_s if episode_length_s is not None else env.max_episode_length_s) / rom.dt)
    x_n = env.get_state().shape[1]
    out = []
    for epoch in range(epochs):
        x = torch.zeros((N, T + 1, x_n), device=env.device)
        z = torch.zeros((N, T + 1, rom.n), device=env.device)
        pz_x = torch.zeros((N, T + 1, rom.n), device=env.device)
        v = torch.zeros((N, T, rom.m), device=env.device)
        done = torch.zeros((N, T), dtype=torch.bool, device=env.device)
        obs, _ = env.reset()
        x[:, 0] = env.get_state()
        pz_x[:, 0] = rom.proj_z(env.root_states)
        z[:, 0] = tg.trajectory[:, 0, :]
        for t in range(T):
            k = tg.k.clone()
            dones = None
            while bool(torch.any(tg.k == k)):                 # step the env until every ROM has stepped
                obs, _, _, dones, _ = env.step(policy(obs.detach()).detach())
            d = dones.clone()
            proj = rom.proj_z(env.root_states)
            done[:, t] = d
            v[:, t] = tg.v
            x[:, t + 1] = env.get_state()
            z[:, t + 1] = tg.get_trajectory()[:, 0, :]
            z[d, t + 1] = proj[d]                             # terminated envs restart with zero tracking error
            pz_x[:, t + 1] = proj
            if progress and t % 20 == 0:
                print(f"epoch {epoch} rom step {t}/{T}", flush=True)
        rec = {"z": z.cpu().numpy(), "v": v.cpu().numpy(), "pz_x": pz_x.cpu().numpy(), "done": done.cpu().numpy()}
        if save_debugging_data:
            rec["x"] = x.cpu().numpy()
        if out_dir is not None:
            with open(os.path.join(out_dir, f"epoch_{epoch}.pickle"), "wb") as f:
                pickle.dump(rec, f)
        out.append(rec)
    return out


def main():
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--out", type=str, default="rom_tracking_data/run")
    ap.add_argument("--save_debugging_data", action="store_true")
    ap.add_argument("--untrained", action="store_true", help="do not resume a checkpoint (random policy)")
    own, rest = ap.parse_known_args()
    args = get_args(rest)
    env_cfg, train_cfg = task_registry.get_cfgs(args.task)
    if not hasattr(env_cfg, "trajectory_generator"):
        raise SystemExit(f"{args.task} is not a trajectory-tracking task")
    env, env_cfg = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    train_cfg.runner.resume = not own.untrained
    runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg, log_root=None if own.untrained else "default")
    policy = runner.get_inference_policy(device=env.device)
    os.makedirs(own.out, exist_ok=True)
    with open(os.path.join(own.out, "config.json"), "w") as f:
        json.dump({"task": args.task, "num_envs": env.num_envs, "epochs": own.epochs, "rom_dt": env.rom.dt,
                   "episode_length_s": env.max_episode_length_s}, f)
    recs = collect(env, policy, own.epochs, save_debugging_data=own.save_debugging_data, out_dir=own.out, progress=True)
    err = np.linalg.norm(recs[-1]["z"] - recs[-1]["pz_x"], axis=-1)
    print(f"wrote {own.epochs} epoch(s) to {own.out}: mean tracking error {err.mean():.3f} m, done rate {recs[-1]['done'].mean():.4f}")


if __name__ == "__main__":
    main()
