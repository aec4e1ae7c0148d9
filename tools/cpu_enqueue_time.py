"""Host time to ENQUEUE one rollout / one update (no synchronisation until the end) against the GPU time of the same work:
the iteration is GPU-bound only while the host stays ahead.  Also times OnPolicyRunner.learn(), whose per-iteration log
line synchronises."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

env, runner = bench.make_runner(4096, [512, 256, 128], "cuda:0", 0, 1)
ar = runner._grad_reduce
for _ in range(3):
    runner.rollout(); runner.ppo.update(ar)
torch.cuda.synchronize()
res = {}
for name, fn in (("rollout", runner.rollout), ("update", lambda: runner.ppo.update(ar))):
    enq, tot = [], []
    for _ in range(5):
        if name == "update":
            runner.rollout()
        torch.cuda.synchronize()
        a = time.perf_counter()
        fn()
        b = time.perf_counter()
        torch.cuda.synchronize()
        c = time.perf_counter()
        enq.append(b - a); tot.append(c - a)
        if name == "rollout":
            runner.ppo.update(ar)
    res[name] = (min(enq) * 1e3, min(tot) * 1e3)
    print(f"{name:8s} host enqueue {min(enq) * 1e3:7.3f} ms   enqueue + GPU drain {min(tot) * 1e3:7.3f} ms", flush=True)
torch.cuda.synchronize()
a = time.perf_counter()
n = 20
for _ in range(n):
    runner.rollout(); runner.ppo.update(ar)
torch.cuda.synchronize()
print(f"bench loop (no sync inside)      {(time.perf_counter() - a) / n * 1e3:7.3f} ms per iteration", flush=True)
import io, contextlib
buf = io.StringIO()
torch.cuda.synchronize()
a = time.perf_counter()
with contextlib.redirect_stdout(buf):
    runner.learn(n, init_at_random_ep_len=False)
torch.cuda.synchronize()
print(f"OnPolicyRunner.learn (log line per iteration) {(time.perf_counter() - a) / n * 1e3:7.3f} ms per iteration", flush=True)
