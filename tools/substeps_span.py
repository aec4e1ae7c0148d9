"""Per-workgroup life of one k_substeps launch on the chip-wide 100 MHz clock (s_memrealtime): when each workgroup's physics wave
started and ended.  Separates the launch's duration (last end - first start, what rocprofv3 reports) into dispatch skew, the
workgroups' own lives and the tail of the slowest.

    make -C legged_gym_dev_amd/csrc prof PROFFLAGS=-DLG_PROF_SPAN PROF_OUT=../lib/liblegged_hip_prof_span.so
    LG_HIP_LIB=legged_gym_dev_amd/lib/liblegged_hip_prof_span.so python tools/substeps_span.py [task ...]
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.test_hip_env import _product_env

for task in (sys.argv[1:] or ["anymal_c_flat"]):
    n = int(os.environ.get("LG_SPAN_ENVS", "4096"))
    env = _product_env(task, n, terrain=None)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    rows = []
    for it in range(40):
        a = torch.randn(n, env.num_actions, device="cuda", generator=g) * 0.3
        env.step(a)
        if it >= 20:
            env.core.lib.lg_debug_control_loop(env.core.ctx, ctypes.c_void_p(a.data_ptr()))
            buf = (ctypes.c_ulonglong * 512)()
            env.core.lib.lg_debug_post_step_cycles(env.core.ctx, buf)
            t = np.array(buf[:], dtype=np.float64).reshape(256, 2)
            t = t[t[:, 1] > 0]
            t0 = t[:, 0].min()
            st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0            # us
            rows.append([len(t), st.max(), np.median(st), (en - st).min(), np.median(en - st), (en - st).max(), en.max(), np.median(en)])
    r = np.array(rows).mean(0)
    print(f"{task}: {r[0]:.0f} workgroups | start skew median {r[2]:.1f} max {r[1]:.1f} us | life min {r[3]:.1f} median {r[4]:.1f} max {r[5]:.1f} us | "
          f"last end - first start {r[6]:.1f} us (median end {r[7]:.1f})", flush=True)
    env.close()
