"""python legged_gym_dev_amd/scripts/train.py --task=anymal_c_flat [--num_envs N --max_iterations K --headless ...]
Same flow as the reference's scripts/train.py:40-44; multi-GPU: launch with torchrun (one rank per GPU)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from legged_gym_dev_amd.envs import *  # noqa: E402,F401,F403  (registers the tasks)
from legged_gym_dev_amd.utils import get_args, task_registry  # noqa: E402


def train(args):
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1 and not torch.distributed.is_initialized():
        # LG_COMM_BACKEND=gloo + LG_SHARE_GPU=1: every rank on cuda:0 over gloo (rehearsal on a one-GPU box)
        local = 0 if os.environ.get("LG_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        args.sim_device = args.rl_device = f"cuda:{local}"
        backend = os.environ.get("LG_COMM_BACKEND", "nccl")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device(args.sim_device))
        else:
            torch.distributed.init_process_group(backend)
    # rank r owns the global envs [r * num_envs, (r + 1) * num_envs): own constants, own Philox streams
    env, env_cfg = task_registry.make_env(name=args.task, args=args, rank=rank, world_size=world)
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args)
    ppo_runner.learn(num_learning_iterations=train_cfg.runner.max_iterations, init_at_random_ep_len=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    train(get_args())
