"""Where the control loop's time goes: s_memtime deltas per section of k_substeps (pair-lane physics), summed over the
decimation substeps of one launch, mean over the two physics waves of the first 16 workgroups.

Needs the section-timing build:  make -C legged_gym_dev_amd/csrc prof
    LG_HIP_LIB=legged_gym_dev_amd/lib/liblegged_hip_prof.so python tools/substeps_sections.py [task ...]
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.test_hip_env import _product_env

NAMES = ["prologue", "torques", "kinematics", "inward", "base+inv", "outward", "detect", "W", "limits", "sweeps", "forces",
         "integrate", "store", "barrier", "epilogue"]
for task in (sys.argv[1:] or ["anymal_c_flat"]):
    env = _product_env(task, 4096, terrain=None)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = np.zeros(16)
    for it in range(40):
        a = torch.randn(4096, env.num_actions, device="cuda", generator=g) * 0.3
        env.step(a)
        if it >= 20:
            env.core.lib.lg_debug_control_loop(env.core.ctx, ctypes.c_void_p(a.data_ptr()))
            buf = (ctypes.c_ulonglong * 512)()
            env.core.lib.lg_debug_post_step_cycles(env.core.ctx, buf)
            tot += np.array(buf[:], dtype=np.float64).reshape(32, 16).mean(0)
    tot /= 20
    print(f"{task}: {tot.sum():.0f} ticks per launch")
    for n, v in zip(NAMES, tot):
        print(f"  {n:10s} {v:9.0f}  {100 * v / tot.sum():5.1f}%")
    env.close()
