"""python legged_gym_dev_amd/scripts/play.py --task=anymal_c_flat [--load_run R --checkpoint K --num_envs N]
Evaluate a trained policy (reference scripts/play.py:53-212): resume the latest checkpoint, export the actor
as TorchScript (policy_1.pt), run the inference policy for one episode length, log robot 0 and the episode
rewards, and write the per-step arrays to play_data.mat.  Headless: there is no viewer in this stack."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from legged_gym_dev_amd import LEGGED_GYM_ROOT_DIR  # noqa: E402
from legged_gym_dev_amd.envs import *  # noqa: E402,F401,F403
from legged_gym_dev_amd.utils import Logger, export_policy_as_jit, get_args, task_registry  # noqa: E402

EXPORT_POLICY = True


def play(args, num_steps=None, out_mat="play_data.mat"):
    env_cfg, train_cfg = task_registry.get_cfgs(name=args.task)
    # the reference plays 1 env on a 1x1 terrain without noise / friction randomisation / pushes (play.py:56-62)
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, args.num_envs if args.num_envs else 1)
    env_cfg.terrain.num_rows = 1
    env_cfg.terrain.num_cols = 1
    env_cfg.terrain.curriculum = False
    env_cfg.noise.add_noise = False
    env_cfg.domain_rand.randomize_friction = False
    env_cfg.domain_rand.push_robots = False
    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    obs = env.get_observations()
    train_cfg.runner.resume = True
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg)
    policy = ppo_runner.get_inference_policy(device=env.device)
    if EXPORT_POLICY:
        path = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name, "exported", "policies")
        print("Exported policy as jit script to:", export_policy_as_jit(ppo_runner.alg.actor_critic, path))
    logger = Logger(env.dt)
    robot_index, joint_index, stop_state_log = 0, 1, 100
    N = int(num_steps if num_steps else env.max_episode_length)
    A = env.num_actions
    rec = {k: np.zeros((N, w)) for k, w in (("cmd", env.commands.shape[1]), ("action", A), ("torque", A), ("pos", 3), ("quat", 4),
                                             ("dof", A), ("vel", 3), ("omega", 3), ("ddof", A))}
    for i in range(N):
        rs = env.root_states[robot_index].cpu().numpy()
        rec["cmd"][i] = env.commands[robot_index].cpu().numpy()
        rec["pos"][i], rec["quat"][i], rec["vel"][i], rec["omega"][i] = rs[:3], rs[3:7], rs[7:10], rs[10:13]
        rec["dof"][i] = env.dof_pos[robot_index].cpu().numpy()
        rec["ddof"][i] = env.dof_vel[robot_index].cpu().numpy()
        actions = policy(obs.detach())
        obs, _, rews, dones, infos = env.step(actions.detach())
        rec["action"][i] = actions[robot_index].cpu().numpy()
        rec["torque"][i] = env.torques[robot_index].cpu().numpy()
        if i < stop_state_log:
            logger.log_states({
                "dof_pos_target": actions[robot_index, joint_index].item() * env.cfg.control.action_scale,
                "dof_pos": env.dof_pos[robot_index, joint_index].item(),
                "dof_vel": env.dof_vel[robot_index, joint_index].item(),
                "dof_torque": env.torques[robot_index, joint_index].item(),
                "command_x": env.commands[robot_index, 0].item(),
                "command_y": env.commands[robot_index, 1].item(),
                "command_yaw": env.commands[robot_index, 2].item(),
                "base_vel_x": env.base_lin_vel[robot_index, 0].item(),
                "base_vel_y": env.base_lin_vel[robot_index, 1].item(),
                "base_vel_z": env.base_lin_vel[robot_index, 2].item(),
                "base_vel_yaw": env.base_ang_vel[robot_index, 2].item(),
                "contact_forces_z": env.contact_forces[robot_index, env.feet_indices, 2].cpu().numpy()})
        elif i == stop_state_log:
            logger.plot_states()
        if i > 0 and infos.get("episode"):
            num_episodes = int(torch.sum(env.reset_buf).item())
            if num_episodes > 0:
                logger.log_rewards({k: float(v) for k, v in infos["episode"].items()}, num_episodes)
    logger.print_rewards()
    if out_mat:
        import scipy.io
        scipy.io.savemat(out_mat, rec)
    return rec


if __name__ == "__main__":
    play(get_args())
