"""Phase timing of the rollout (host-issue time vs device time per phase)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

hidden = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512,256,128").split(",")]
env, runner = bench.make_runner(4096, hidden, "cuda:0", 0, 1)
ppo = runner.ppo
obs = env.get_observations()
def timeit(name, fn, n=24, reps=5):
    fn(); torch.cuda.synchronize()
    best_host, best_tot = 1e9, 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(n): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best_host, best_tot = min(best_host, t1 - t0), min(best_tot, t2 - t0)
    print(f"{name:28s} host-issue {1e6*best_host/n:8.1f} us/call   total {1e6*best_tot/n:8.1f} us/call", flush=True)
a = torch.zeros(4096, 12, device="cuda")
timeit("env.core.step", lambda: env.core.step(a))
timeit("env.step (python shim)", lambda: env.step(a))
def act():
    ppo._call("end_update"); ppo.act(obs, None)
timeit("ppo.act (fused forward)", act)
ppo.lib.lg_ppo_debug_set_fused_act(ppo.ctx, 0)
timeit("ppo.act (per-layer GEMMs)", act)
ppo.lib.lg_ppo_debug_set_fused_act(ppo.ctx, 1)
def proc():
    ppo._call("end_update"); ppo.process_env_step(env.rew_buf, env.core.t["reset"], {"time_outs": env.core.t["extras_time_outs"]})
timeit("ppo.process_env_step", proc)
timeit("set_actions only", lambda: env.core.call("set_actions", __import__("ctypes").c_void_p(a.data_ptr())))
timeit("compute_torques only", lambda: env.core.call("compute_torques"))
timeit("simulate only", lambda: env.core.call("simulate"))
timeit("post_physics_step only", lambda: env.core.call("post_physics_step"))
timeit("rollout() whole", lambda: (ppo._call("end_update"), runner.rollout()), n=1)
ppo._call("begin_update")
timeit("minibatch_backward", lambda: ppo._call("minibatch_backward", 0, 0), n=4)
timeit("minibatch_step", lambda: ppo._call("minibatch_step"), n=4)
