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
    ticks, real = tot[:15].sum(), tot[15]                 # s_memtime ticks of the sections; s_memrealtime (100 MHz) over the same span
    print(f"{task}: {ticks:.0f} ticks per launch over {real / 100.0:.1f} us: in-kernel clock {ticks / real * 0.1:.2f} GHz")
    for n, v in zip(NAMES, tot[:15]):
        print(f"  {n:10s} {v:9.0f}  {100 * v / ticks:5.1f}%")
    if os.environ.get("LG_CLOCK_JSON") and task == "anymal_c_flat":       # what bench.py's roofline_env_step prices the VALU issue rate with
        import json
        json.dump({"workload": "anymal_c_flat, 4096 envs, k_substeps physics waves of the first 16 workgroups (make prof build)",
                   "critical_wave_ticks": round(float(ticks)), "critical_wave_us": round(float(real) / 100.0, 2),
                   "clock_ghz_in_kernel": round(float(ticks / real * 0.1), 3),
                   "sections_percent": {n: round(100 * float(v) / float(ticks), 1) for n, v in zip(NAMES, tot[:15])},
                   "note": "s_memtime ticks between the kernel's first and last stamp over s_memrealtime (100 MHz) of the same span"},
                  open(os.environ["LG_CLOCK_JSON"], "w"), indent=1)
    env.close()
