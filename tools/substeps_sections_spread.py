"""Sections of the control loop for the fastest and the slowest of the 16 stamped workgroups (make prof build): which sections make
the launch's tail.  LG_HIP_LIB=.../liblegged_hip_prof.so python tools/substeps_sections_spread.py [task]"""
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
    for it in range(40):
        a = torch.randn(4096, env.num_actions, device="cuda", generator=g) * 0.3
        env.step(a)
    env.core.lib.lg_debug_control_loop(env.core.ctx, ctypes.c_void_p(a.data_ptr()))
    buf = (ctypes.c_ulonglong * 512)()
    env.core.lib.lg_debug_post_step_cycles(env.core.ctx, buf)
    t = np.array(buf[:], dtype=np.float64).reshape(16, 2, 16)[:, 0, :]          # physics wave 0 of each stamped workgroup
    tot = t[:, :15].sum(1)
    lo, hi = int(tot.argmin()), int(tot.argmax())
    print(f"{task}: workgroup totals (k ticks): " + " ".join(f"{v / 1e3:.0f}" for v in tot))
    print(f"  {'section':10s} {'fastest':>9s} {'slowest':>9s} {'diff':>8s}")
    for i, n in enumerate(NAMES):
        print(f"  {n:10s} {t[lo, i]:9.0f} {t[hi, i]:9.0f} {t[hi, i] - t[lo, i]:8.0f}")
    cf = env.contact_forces.norm(dim=-1)
    print("  bodies in contact per env (first 256 envs, by workgroup of 16):",
          " ".join(str(int((cf[w * 16:(w + 1) * 16] > 0).sum(1).max())) for w in range(16)))
    env.close()
