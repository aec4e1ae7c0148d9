"""name -> (env class, env cfg, train cfg); ``make_env`` / ``make_alg_runner`` with the reference's
signatures (legged_gym/utils/task_registry.py:45-159).  The runner is this repo's HIP-backed
``OnPolicyRunner`` (legged_gym_dev_amd/rl/runner.py)."""
import os
from datetime import datetime

from legged_gym_dev_amd import LEGGED_GYM_ROOT_DIR
from .helpers import class_to_dict, get_args, get_load_path, parse_sim_params, set_seed, update_cfg_from_args


class TaskRegistry:
    def __init__(self):
        self.task_classes, self.env_cfgs, self.train_cfgs = {}, {}, {}

    def register(self, name, task_class, env_cfg, train_cfg):
        self.task_classes[name], self.env_cfgs[name], self.train_cfgs[name] = task_class, env_cfg, train_cfg

    def get_task_class(self, name):
        return self.task_classes[name]

    def get_cfgs(self, name):
        train_cfg, env_cfg = self.train_cfgs[name], self.env_cfgs[name]
        env_cfg.seed = train_cfg.seed
        return env_cfg, train_cfg

    def make_env(self, name, args=None, env_cfg=None, rank=0, world_size=1):
        if args is None:
            args = get_args()
        if name not in self.task_classes:
            raise ValueError(f"Task with name: {name} was not registered")
        task_class = self.get_task_class(name)
        if env_cfg is None:
            env_cfg, _ = self.get_cfgs(name)
        env_cfg, _ = update_cfg_from_args(env_cfg, None, args)
        set_seed(env_cfg.seed)
        sim_params = parse_sim_params(args, {"sim": class_to_dict(env_cfg.sim)})
        kw = {"rank": rank, "world_size": world_size} if world_size > 1 else {}
        env = task_class(cfg=env_cfg, sim_params=sim_params, physics_engine=args.physics_engine,
                         sim_device=args.sim_device, headless=args.headless, **kw)
        return env, env_cfg

    def make_alg_runner(self, env, name=None, args=None, train_cfg=None, log_root="default", wandb_callback=None):
        from legged_gym_dev_amd.rl.runner import OnPolicyRunner
        if args is None:
            args = get_args()
        if train_cfg is None:
            if name is None:
                raise ValueError("Either 'name' or 'train_cfg' must be not None")
            _, train_cfg = self.get_cfgs(name)
        elif name is not None:
            print(f"'train_cfg' provided -> Ignoring 'name={name}'")
        _, train_cfg = update_cfg_from_args(None, train_cfg, args)
        stamp = datetime.now().strftime("%b%d_%H-%M-%S") + "_" + train_cfg.runner.run_name
        if log_root == "default":
            log_root = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name)
            log_dir = os.path.join(log_root, stamp)
        elif log_root is None:
            log_dir = None
        else:
            log_dir = os.path.join(log_root, stamp)
        runner = OnPolicyRunner(env, class_to_dict(train_cfg), log_dir, device=args.rl_device, wandb_callback=wandb_callback)
        if train_cfg.runner.resume:
            resume_path = get_load_path(log_root, load_run=train_cfg.runner.load_run, checkpoint=train_cfg.runner.checkpoint)
            print(f"Loading model from: {resume_path}")
            runner.load(resume_path)
        return runner, train_cfg


task_registry = TaskRegistry()
