"""How far apart two runs of the product runner from ONE seed land, with float atomics (default) and in deterministic mode
(lg_ppo_set_deterministic): relative distance of the parameter vectors after 1..3 PPO iterations."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from legged_gym_dev_amd.envs import task_registry
from legged_gym_dev_amd.rl.runner import OnPolicyRunner
from legged_gym_dev_amd.utils import get_args
from legged_gym_dev_amd.utils.helpers import class_to_dict


def run(task, n, hidden, det, iters):
    args = get_args(["--task", task, "--num_envs", str(n), "--headless"])
    args.sim_device = args.rl_device = "cuda:0"
    env_cfg, train_cfg = (copy.deepcopy(c) for c in task_registry.get_cfgs(task))
    env_cfg.env.num_envs = n
    train_cfg.policy.actor_hidden_dims = train_cfg.policy.critic_hidden_dims = list(hidden)
    env, _ = task_registry.make_env(name=task, args=args, env_cfg=env_cfg)
    torch.manual_seed(11)
    runner = OnPolicyRunner(env, class_to_dict(train_cfg), None, device="cuda:0")
    runner.ppo.set_deterministic(det)
    snaps = []
    for _ in range(iters):
        runner.learn(1, init_at_random_ep_len=False)
        torch.cuda.synchronize()
        snaps.append(runner.ppo.t["params"][: runner.ppo.num_params].clone())
    env.close(); runner.ppo.close()
    return snaps


for task, n, hidden in (("anymal_c_flat", 256, [512, 256, 128]), ("anymal_c_flat", 4096, [512, 256, 128])):
    for det in (False, True):
        a, b = run(task, n, hidden, det, 3), run(task, n, hidden, det, 3)
        d = [float((x - y).norm() / x.norm()) for x, y in zip(a, b)]
        print(f"{task} {n} envs {hidden} {'deterministic' if det else 'float atomics'}: |run1 - run2| / |run1| after 1, 2, 3 iterations = "
              + ", ".join(f"{v:.2e}" for v in d), flush=True)
