"""Where k_post_step's time goes: s_memtime stamps of the first 64 workgroups at the phase boundaries
(H height scan | A per-env logic + rewards | R resets | O observations), mean cycles per phase."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.test_hip_env import _product_env

for task in (sys.argv[1:] or ["anymal_c_flat", "anymal_c_rough", "cassie"]):
    env = _product_env(task, 4096, terrain=None)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = np.zeros(4)
    sub = np.zeros(4)                       # phase O: frame + joint entries | height entries | noise pass | bookkeeping
    nres = 0
    for it in range(30):
        env.step(torch.randn(4096, env.num_actions, device="cuda", generator=g) * 0.3)
        buf = (ctypes.c_ulonglong * 512)()
        env.core.lib.lg_debug_post_step_cycles(env.core.ctx, buf)
        c = np.array(buf[:], dtype=np.float64).reshape(64, 8)
        if it >= 10:
            tot += np.diff(c[:, :5], axis=1).mean(0)
            o = c[:, [3, 5, 6, 7, 4]]
            sub += np.diff(o, axis=1).mean(0)
            nres += int(env.core.t["n_reset"][0])
    tot /= 20
    sub /= 20
    print(f"{task:16s} cycles/phase  H {tot[0]:8.0f}  A {tot[1]:8.0f}  R {tot[2]:8.0f}  O {tot[3]:8.0f}   (resets/step {nres / 20:.1f})   O = entries {sub[0]:.0f} + heights {sub[1]:.0f} + noise pass {sub[2]:.0f} + bookkeeping {sub[3]:.0f}", flush=True)
    env.close()
