"""us per lg_step (k_substeps + k_post_step + k_finalize) and per stage, 4096 envs: flat ANYmal-C (actuator net), rough ANYmal-C, Cassie."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_hip_env import _product_env


def t(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for task in (sys.argv[1:] or ["anymal_c_flat", "anymal_c_rough", "cassie"]):
    env = _product_env(task, 4096, terrain=None)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(20):
        env.step(torch.randn(4096, env.num_actions, device="cuda", generator=g) * 0.3)
    a = torch.zeros(4096, env.num_actions, device="cuda")
    print(f"{task:16s} lg_step {t(lambda: env.core.step(a)):7.1f} us | torques {t(lambda: env.core.call('compute_torques')):6.1f} | "
          f"simulate {t(lambda: env.core.call('simulate')):6.1f} | post_step {t(lambda: env.core.call('post_physics_step')):6.1f}", flush=True)
    env.close()
