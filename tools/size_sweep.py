"""Boundary sizes through the product path: envs not a multiple of any tile (control loop, post-step, one-launch act, minibatch rows
off the GEMM / head tile grid), two PPO iterations each; reports finiteness, physics faults and parameter change."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from legged_gym_dev_amd.envs import task_registry
from legged_gym_dev_amd.rl.runner import OnPolicyRunner
from legged_gym_dev_amd.utils import get_args
from legged_gym_dev_amd.utils.helpers import class_to_dict

CASES = (("anymal_c_flat", 1000, [512, 256, 128]), ("anymal_c_flat", 4097, [128, 64, 32]), ("cassie", 777, [512, 256, 128]),
         ("anymal_c_flat_trajectory", 1234, [512, 256, 128]), ("anymal_c_flat", 8192, [512, 256, 128]), ("anymal_c_flat", 7, [64, 32]))


def run(task, n, hidden):
    args = get_args(["--task", task, "--num_envs", str(n), "--headless"])
    args.sim_device = args.rl_device = "cuda:0"
    env_cfg, train_cfg = (copy.deepcopy(c) for c in task_registry.get_cfgs(task))
    env_cfg.env.num_envs = n
    if getattr(env_cfg.terrain, "mesh_type", None) in ("heightfield", "trimesh"):
        env_cfg.terrain.num_rows, env_cfg.terrain.num_cols, env_cfg.terrain.border_size = 3, 4, 5
        env_cfg.terrain.max_init_terrain_level = 2
    train_cfg.policy.actor_hidden_dims = train_cfg.policy.critic_hidden_dims = hidden
    if (24 * n) % train_cfg.algorithm.num_mini_batches:
        train_cfg.algorithm.num_mini_batches = 1 if n == 7 else 3 if (24 * n) % 3 == 0 else 2
    env, _ = task_registry.make_env(name=task, args=args, env_cfg=env_cfg)
    runner = OnPolicyRunner(env, class_to_dict(train_cfg), None, device="cuda:0")
    p0 = runner.ppo.t["params"][: runner.ppo.num_params].clone()
    runner.learn(2, init_at_random_ep_len=True)
    torch.cuda.synchronize()
    p1 = runner.ppo.t["params"][: runner.ppo.num_params]
    ok = bool(torch.isfinite(p1).all()) and bool(torch.isfinite(env.core.t["obs"]).all()) and bool(torch.isfinite(env.core.t["root_states"]).all())
    print(f"{task:26s} envs {n:5d} hidden {hidden} minibatches {train_cfg.algorithm.num_mini_batches}: finite {ok}  faults {int(env.fault_total)}  "
          f"|dparams| {float((p1 - p0).abs().max()):.2e}  lr {runner.ppo.learning_rate:.2e}", flush=True)
    assert ok and int(env.fault_total) == 0 and float((p1 - p0).abs().max()) > 0
    env.close(); runner.ppo.close()


if __name__ == "__main__":
    for case in CASES:
        run(*case)
    print("size sweep ok")
