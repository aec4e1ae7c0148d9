"""Run a few PPO minibatch forward/backward passes (the MFMA GEMM group) for profiling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from legged_gym_dev_amd.rl.ppo import HipPPO
hidden = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512,256,128").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pol = {"actor_hidden_dims": hidden, "critic_hidden_dims": hidden, "activation": "elu", "init_noise_std": 1.0}
alg = dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=5,
           num_mini_batches=4, learning_rate=1e-3, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0)
ppo = HipPPO(4096, 48, None, 12, pol, alg, 24, device="cuda:0")
ppo.t["obs"].normal_(); ppo.t["actions"].normal_(); ppo.t["advantages"].normal_(); ppo.t["returns"].normal_()
ppo.t["values"].normal_(); ppo.t["mu"].normal_(); ppo.t["sigma"].fill_(1.0); ppo.t["log_prob"].fill_(-17.0)
ppo._call("begin_update")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ppo._call("minibatch_backward", 0, 0)
e0.record()
for k in range(reps):
    ppo._call("minibatch_backward", 0, k % 4)
e1.record()
torch.cuda.synchronize()
print("ms per minibatch fwd+bwd:", e0.elapsed_time(e1) / reps)
if len(sys.argv) > 3:
    import ctypes
    for v in (0, 1, 0, 1):
        ppo.lib.ppok_debug_set_t96(ctypes.c_int(v))
        ppo._call("minibatch_backward", 0, 0)
        torch.cuda.synchronize()
        e0.record()
        for k in range(reps):
            ppo._call("minibatch_backward", 0, k % 4)
        e1.record()
        torch.cuda.synchronize()
        print("96-row tiles", v, "ms:", e0.elapsed_time(e1) / reps)
