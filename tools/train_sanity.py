"""End-to-end learning sanity on the GPU.

    python tools/train_sanity.py ITERS [HIDDEN] [MODE] [ROBOT]

ROBOT "anymal_c" (default, actuator net, plane), "a1" (Unitree A1, PD law, plane), "anymal_c_rough" (mixed terrain
with the height scan and the terrain curriculum) or "anymal_c_trajectory" (the trajectory-tracking task anymal_c_flat_trajectory
with the reward table the fork's authors launch it with, deep_tube_learning/configs/rl/default.yaml:29-39: tracking_rom 6, ...).

MODE "ref"  : the fork's anymal_c_flat config as committed (its reward is identically 0 after the
              positive clip: commands x,y are 0 so feet_air_time never pays).
MODE "walk" : same env with ETH legged_gym's default velocity-tracking reward table and command
              ranges -- a task with a learning signal, to show that rollout + physics + PPO learn.
Prints the learning curve (mean episode return / length of the episodes finished in each window).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
hidden = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "128,64,32").split(",")]
mode = sys.argv[3] if len(sys.argv) > 3 else "walk"
robot = sys.argv[4] if len(sys.argv) > 4 else "anymal_c"

from legged_gym_dev_amd.envs.anymal_c.anymal import Anymal
from legged_gym_dev_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg, AnymalCFlatCfgPPO
from legged_gym_dev_amd.rl.runner import OnPolicyRunner
from legged_gym_dev_amd.utils.helpers import class_to_dict, get_args, parse_sim_params

env_cfg, train_cfg = AnymalCFlatCfg(), AnymalCFlatCfgPPO()
EnvCls = Anymal
if robot == "anymal_c_rough":
    from legged_gym_dev_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg, AnymalCRoughCfgPPO
    env_cfg, train_cfg = AnymalCRoughCfg(), AnymalCRoughCfgPPO()
if robot == "anymal_c_trajectory":
    from legged_gym_dev_amd.envs.anymal_c.anymal_trajectory import AnymalTrajectory as EnvCls
    from legged_gym_dev_amd.envs.anymal_c.flat_trajectory.anymal_c_flat_trajectory_config import AnymalCFlatTrajectoryCfg, AnymalCFlatTrajectoryCfgPPO
    env_cfg, train_cfg = AnymalCFlatTrajectoryCfg(), AnymalCFlatTrajectoryCfgPPO()
    for k, v in dict(termination=-0.5, tracking_rom=6.0, ang_vel_xy=-0.05, orientation=-1.0, torques=-1e-5, dof_acc=-2.5e-7, collision=-1.0,
                     action_rate=-0.1, feet_air_time=0.0).items():
        setattr(env_cfg.rewards.scales, k, v)
    if mode == "curriculum":
        # the authors' staged curriculum (deep_tube_learning/configs/rl/default.yaml:77-109: three stages -- pushes 0.1 / 0.5 / 1 x,
        # hold times 3 / 2 / 1 x, ROM speed 0.5 / 0.75 / 1 x, tracking sigma and every reward scale 1 / 0.8 / 0.6 x), with the stage
        # changes at 60 and 120 iterations (24 policy steps each) so that a 200-iteration run shows all three
        env_cfg.curriculum.use_curriculum = True
        env_cfg.curriculum.curriculum_steps = [24 * 60, 24 * 120]
        env_cfg.rewards.scales.feet_air_time = 0.0
    else:
        mode = "as launched by the authors"
if robot == "a1":
    from legged_gym_dev_amd.envs.a1.a1_config import A1RoughCfg, A1RoughCfgPPO
    from legged_gym_dev_amd.envs.base.legged_robot import LeggedRobot as EnvCls
    env_cfg, train_cfg = A1RoughCfg(), A1RoughCfgPPO()
    env_cfg.terrain.mesh_type, env_cfg.terrain.measure_heights, env_cfg.terrain.curriculum = "plane", False, False
    env_cfg.env.num_observations = 48
env_cfg.env.num_envs = 4096
env_cfg.seed = 1
if mode == "walk":
    r = env_cfg.commands.ranges
    r.lin_vel_x, r.lin_vel_y, r.ang_vel_yaw = [-1.0, 1.0], [-1.0, 1.0], [-1.5, 1.5]
    sc = env_cfg.rewards.scales
    for k, v in dict(tracking_lin_vel=1.0, tracking_ang_vel=0.5, lin_vel_z=-2.0, ang_vel_xy=-0.05, dof_acc=-2.5e-7,
                     collision=-1.0, action_rate=-0.01, orientation=-5.0 if robot == "anymal_c" else -0.0,   # ETH: -5 only in the flat ANYmal task, 0 on rough terrain / A1
                     torques=-0.000025 if robot.startswith("anymal_c") else -0.0002,
                     feet_air_time=2.0 if robot.startswith("anymal_c") else 1.0).items():
        setattr(sc, k, v)
train_cfg.policy.actor_hidden_dims = list(hidden)
train_cfg.policy.critic_hidden_dims = list(hidden)
args = get_args([])
args.sim_device = args.rl_device = "cuda:0"
torch.manual_seed(1)
np.random.seed(1)
env = EnvCls(env_cfg, parse_sim_params(args, {"sim": class_to_dict(env_cfg.sim)}), args.physics_engine, "cuda:0", True)
runner = OnPolicyRunner(env, class_to_dict(train_cfg), None, device="cuda:0")
ppo = runner.ppo
env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
t0 = time.time()
for it in range(iters):
    runner.rollout()
    vl, sl = ppo.update()
    if it % 25 == 24 or it == 0:
        es = ppo.t["ep_stats"].cpu().tolist()
        ppo.t["ep_stats"].zero_()
        n = max(es[2], 1)
        bad = int((~torch.isfinite(env.root_states).all(1)).sum()) + int((~torch.isfinite(env.obs_buf).all(1)).sum())
        lvl = float(env.terrain_levels.float().mean()) if hasattr(env, "terrain_levels") else 0.0
        trk_key = "rew_tracking_rom" if robot == "anymal_c_trajectory" else "rew_tracking_lin_vel"
        trk = float(env.extras["episode"].get(trk_key, torch.zeros(()))) if "episode" in env.extras else 0.0
        if robot == "anymal_c_trajectory":
            terr = float((env.trajectory[:, 0] - env.root_states[:, :2]).norm(dim=1).mean())
            print(f"        mean distance to the reference trajectory {terr:.3f} m, physics faults {int(env.fault_total[0])}, "
                  f"curriculum stage {env.curriculum_state} (sigma {env.tracking_sigma:.3f}, v_max {float(env.rom.v_max[0]):.3f})", flush=True)
        print(f"it {it + 1:4d}  mean_return {es[0] / n:8.3f}  mean_ep_len {es[1] / n:7.1f}  episodes {int(es[2]):6d}  "
              f"std {float(ppo.param_views['std'].mean()):.3f} lr {ppo.learning_rate:.2e} vloss {float(vl):.4f}  "
              f"base_z {float(env.root_states[:, 2].mean()):.3f} rew_tracking {trk:.4f} terrain_level {lvl:.2f} nonfinite_envs {bad}", flush=True)
dt = time.time() - t0
print(f"{iters} iterations in {dt:.1f}s -> {iters * 24 * 4096 / dt:.0f} env-steps/s incl. logging")
