"""Driver helpers with the reference's names and behaviour (legged_gym/utils/helpers.py:111-271):
``class_to_dict``, ``set_seed``, ``parse_sim_params``, ``get_load_path``, ``update_cfg_from_args``,
``get_args``.  ``gymutil.parse_arguments`` (Isaac Gym) is replaced by argparse with the same flags.
"""
import argparse
import os
import random
from types import SimpleNamespace

import numpy as np
import torch


def class_to_dict(obj) -> dict:
    """Recursively turn a config object into a dict.  Keys come from ``dir()`` and are therefore
    alphabetical -- this is what fixes the reward summation order (helpers.py:111-126)."""
    if not hasattr(obj, "__dict__"):
        return obj
    out = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        out[key] = [class_to_dict(v) for v in val] if isinstance(val, list) else class_to_dict(val)
    return out


def update_class_from_dict(obj, dct):
    for key, val in dct.items():
        attr = getattr(obj, key, None)
        if isinstance(attr, type):
            update_class_from_dict(attr, val)
        else:
            setattr(obj, key, val)


def set_seed(seed):
    if seed == -1:
        seed = np.random.randint(0, 10000)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    return seed


class SimParams(SimpleNamespace):
    """Stand-in for gymapi.SimParams: ``dt`` is stored as a C float (see env_setup.sim_dt_float)."""

    def __init__(self, **kw):
        super().__init__(**kw)

    def __setattr__(self, k, v):
        if k == "dt":
            v = float(np.float32(v))
        super().__setattr__(k, v)


def parse_sim_params(args, cfg):
    """cfg = {"sim": class_to_dict(env_cfg.sim)} as in the reference (task_registry.py:96-97)."""
    sp = SimParams(dt=1.0 / 60.0, substeps=2, up_axis=1, gravity=[0.0, 0.0, -9.81],
                   use_gpu_pipeline=getattr(args, "use_gpu_pipeline", True), physx=SimpleNamespace())
    for k, v in cfg.get("sim", {}).items():
        if k == "physx":
            for pk, pv in v.items():
                setattr(sp.physx, pk, pv)
        else:
            setattr(sp, k, v)
    if getattr(args, "num_threads", 0) > 0:
        sp.physx.num_threads = args.num_threads
    return sp


def get_load_path(root, load_run=-1, checkpoint=-1):
    try:
        runs = sorted(os.listdir(root))
        if "exported" in runs:
            runs.remove("exported")
        last_run = os.path.join(root, runs[-1])
    except Exception:
        raise ValueError("No runs in this directory: " + root)
    load_run = last_run if load_run == -1 else os.path.join(root, load_run)
    if checkpoint == -1:
        models = [f for f in os.listdir(load_run) if "model" in f]
        models.sort(key=lambda m: "{0:0>15}".format(m))
        model = models[-1]
    else:
        model = "model_{}.pt".format(checkpoint)
    return os.path.join(load_run, model)


def update_cfg_from_args(env_cfg, cfg_train, args):
    if env_cfg is not None and getattr(args, "num_envs", None) is not None:
        env_cfg.env.num_envs = args.num_envs
    if cfg_train is not None:
        if getattr(args, "seed", None) is not None:
            cfg_train.seed = args.seed
        r = cfg_train.runner
        if getattr(args, "max_iterations", None) is not None:
            r.max_iterations = args.max_iterations
        if getattr(args, "resume", False):
            r.resume = args.resume
        for name in ("experiment_name", "run_name", "load_run", "checkpoint"):
            if getattr(args, name, None) is not None:
                setattr(r, name, getattr(args, name))
    return env_cfg, cfg_train


def get_args(argv=None):
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--task", type=str, default="anymal_c_flat")
    p.add_argument("--resume", action="store_true", default=False)
    p.add_argument("--experiment_name", type=str)
    p.add_argument("--run_name", type=str)
    p.add_argument("--load_run", type=str)
    p.add_argument("--checkpoint", type=int)
    p.add_argument("--headless", action="store_true", default=False)
    p.add_argument("--horovod", action="store_true", default=False)
    p.add_argument("--rl_device", type=str, default="cuda:0")
    p.add_argument("--num_envs", type=int)
    p.add_argument("--seed", type=int)
    p.add_argument("--max_iterations", type=int)
    p.add_argument("--sim_device", type=str, default="cuda:0")
    p.add_argument("--pipeline", type=str, default="gpu")
    p.add_argument("--num_threads", type=int, default=0)
    args, _ = p.parse_known_args(argv)
    args.physics_engine = 1          # SIM_PHYSX placeholder: the HIP physics is the only engine
    args.use_gpu_pipeline = args.pipeline.lower() in ("gpu", "cuda")
    args.sim_device_type = args.sim_device.split(":")[0]
    args.compute_device_id = int(args.sim_device.split(":")[1]) if ":" in args.sim_device else 0
    args.sim_device_id = args.compute_device_id
    return args
