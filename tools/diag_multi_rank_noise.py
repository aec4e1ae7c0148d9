"""Run-to-run spread of the two-rank product run (two processes over gloo sharing cuda:0) and of its in-process thread
emulation, and the distance between the two kinds (tests/test_hip_multi_rank.py compares them):
    python tools/diag_multi_rank_noise.py [task] [n] [iters]"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_multi_rank as t

task = sys.argv[1] if len(sys.argv) > 1 else "anymal_c_rough"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
runs = {}
for k in range(3):
    with tempfile.TemporaryDirectory() as d:
        snap = t._run_ranks(d, task, n, iters)[0]
        runs[f"proc{k}"] = snap["params"]
        print(f"proc{k} lr {float(snap['lr']):.6g} adv_mean {float(snap['adv_mean']):.9g}", flush=True)
for k in range(3):
    snap = t._emulate(task, n, iters)[0]
    runs[f"emu{k}"] = snap["params"]
    print(f"emu{k} lr {float(snap['lr']):.6g} adv_mean {float(snap['adv_mean']):.9g}", flush=True)
names = list(runs)
ref = np.linalg.norm(runs["proc0"])
for i, a in enumerate(names):
    print(a, " ".join(f"{np.linalg.norm(runs[a] - runs[b]) / ref:9.2e}" for b in names), flush=True)
