"""Does a captured hipGraph shorten the minibatch period?  {minibatch_backward, minibatch_step} of the bench workload launched back to back
directly and as one replayed graph (lg_ppo_debug_graph_period): us per minibatch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

env, runner = bench.make_runner(4096, [512, 256, 128], "cuda:0", 0, 1)
ppo = runner.ppo
for _ in range(2):
    runner.rollout(); ppo.update(None)
runner.rollout()
ppo._call("begin_update")
torch.cuda.synchronize()
ppo.lib.lg_ppo_debug_graph_period.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
out = ctypes.c_float()
for rep in range(3):
    for g in (0, 1):
        rc = ppo.lib.lg_ppo_debug_graph_period(ppo.ctx, 40, g, ctypes.byref(out))
        print(f"{'hipGraph replay ' if g else 'direct launches '}: rc {rc}  {out.value:7.1f} us per minibatch (backward + optimiser step, gather included)"
              + ("" if rc == 0 else "  " + ppo.lib.lg_last_error().decode()), flush=True)
