"""Section timing of the one-launch PPO.act (k_mlp_fwd): s_memtime deltas of wave 0 of the first 32 workgroups of each net:
input staging | layer 1 | layer 2 | layer 3 | head | sampling epilogue.
    make -C legged_gym_dev_amd/csrc prof && LG_HIP_LIB=legged_gym_dev_amd/lib/liblegged_hip_prof.so python tools/mlp_sections.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_hip_ppo import _make

N, O, A, T = 4096, 48, 12, 24
hip, _, _ = _make(N, O, A, T)
obs = torch.randn(N, O, device="cuda")
for t in range(8):
    hip.act(obs)
    hip.process_env_step(torch.zeros(N, device="cuda"), torch.zeros(N, dtype=torch.uint8, device="cuda"), {})
torch.cuda.synchronize()
d = hip.t["noise"].flatten()[: 2 * 32 * 8].reshape(2, 32, 8).cpu()
names = ["stage in", "layer 1", "layer 2", "layer 3", "head", "sample"]
for z, nm in enumerate(("actor", "critic")):
    m = d[z].mean(0)
    print(nm, "  ".join(f"{n} {v:7.0f}" for n, v in zip(names, m[:6].tolist())), f"  total {m[:6].sum():.0f} ticks")
