"""Same public names as the reference's legged_gym/utils/__init__.py."""
from .helpers import class_to_dict, get_args, get_load_path, set_seed, update_class_from_dict  # noqa: F401
from .math import *  # noqa: F401,F403
from .terrain import Terrain  # noqa: F401
from .task_registry import task_registry  # noqa: F401
from .logger import Logger  # noqa: F401,E402
from ..rl.checkpoint import export_policy_as_jit  # noqa: F401,E402
