"""Checkpoint / export formats on CPU: the optimiser entry of model_<it>.pt must be a
state_dict torch.optim.Adam itself accepts (and round-trips through), parameter order must be
ActorCritic.parameters() order, and the exported TorchScript actor must compute the oracle's actor."""
import os

import numpy as np
import torch

from legged_gym_dev_amd.rl import checkpoint as ck
from oracle import ppo_torch


def _ac(O=48, A=12, hidden=(32, 16, 8)):
    torch.manual_seed(0)
    return ppo_torch.ActorCritic(O, O, A, list(hidden), list(hidden), "elu", 1.0)


def test_parameter_order_is_module_order():
    ac = _ac()
    sd = ac.state_dict()
    assert ck.parameter_order({k: sd[k] for k in reversed(list(sd))}) == [n for n, _ in ac.named_parameters()]


def test_adam_state_round_trips_through_torch_adam():
    ac = _ac()
    names = [n for n, _ in ac.named_parameters()]
    shapes = {n: tuple(p.shape) for n, p in ac.named_parameters()}
    total = sum(p.numel() for p in ac.parameters())
    g = torch.Generator().manual_seed(1)
    m, v = torch.randn(total, generator=g), torch.rand(total, generator=g)
    sd = ck.adam_state_to_torch(names, shapes, m, v, step=37, lr=3e-4)
    opt = torch.optim.Adam(ac.parameters(), lr=1e-3)
    opt.load_state_dict(sd)                                   # torch accepts the layout
    assert opt.param_groups[0]["lr"] == 3e-4
    p0 = list(ac.parameters())[1]
    assert torch.equal(opt.state[p0]["exp_avg"], m[12:12 + p0.numel()].reshape(p0.shape))
    # ... and what torch writes back is read to the same flat buffers
    m2, v2, step, lr = ck.adam_state_from_torch(opt.state_dict(), names, shapes)
    assert torch.equal(m2, m) and torch.equal(v2, v) and step == 37 and lr == 3e-4
    # one real torch step from that state == one hand-written Adam step on the flat buffers
    for p in ac.parameters():
        p.grad = torch.full_like(p, 0.01)
    before = torch.cat([p.detach().flatten().clone() for p in ac.parameters()])
    opt.step()
    gflat = torch.full((total,), 0.01)
    mm, vv, t = 0.9 * m + 0.1 * gflat, 0.999 * v + 0.001 * gflat * gflat, 38
    want = before - (3e-4 / (1 - 0.9 ** t)) * mm / (vv.sqrt() / np.sqrt(1 - 0.999 ** t) + 1e-8)
    after = torch.cat([p.detach().flatten() for p in ac.parameters()])
    assert torch.allclose(after, want, rtol=1e-5, atol=1e-7)


def test_fresh_torch_optimizer_state_loads_as_zeros():
    ac = _ac()
    names = [n for n, _ in ac.named_parameters()]
    shapes = {n: tuple(p.shape) for n, p in ac.named_parameters()}
    m, v, step, lr = ck.adam_state_from_torch(torch.optim.Adam(ac.parameters(), lr=1e-3).state_dict(), names, shapes)
    assert float(m.abs().sum()) == 0.0 and float(v.abs().sum()) == 0.0 and step == 0.0 and lr == 1e-3


def test_legacy_layout_still_loads():
    m, v, step, lr = ck.adam_state_from_torch({"adam_m": torch.ones(5), "adam_v": torch.zeros(5), "step": 3.0, "lr": 1e-4}, [], {})
    assert m.numel() == 5 and step == 3.0 and lr == 1e-4


def test_exported_jit_actor_matches_module(tmp_path):
    ac = _ac()
    path = ck.export_policy_as_jit(ac, str(tmp_path))
    assert os.path.basename(path) == "policy_1.pt"
    mod = torch.jit.load(path)                               # a file this test wrote itself
    x = torch.randn(7, 48)
    assert torch.allclose(mod(x), ac.actor(x), rtol=0, atol=0)
    assert [n for n, _ in mod.named_parameters()] == [n[len("actor."):] for n, _ in ac.named_parameters() if n.startswith("actor.")]


def test_checkpoint_file_round_trip(tmp_path):
    ac = _ac()
    names = [n for n, _ in ac.named_parameters()]
    shapes = {n: tuple(p.shape) for n, p in ac.named_parameters()}
    total = sum(p.numel() for p in ac.parameters())
    d = {"model_state_dict": ac.state_dict(), "optimizer_state_dict": ck.adam_state_to_torch(names, shapes, torch.ones(total), torch.ones(total), 5, 1e-3),
         "iter": 50, "infos": None}
    f = tmp_path / "model_50.pt"
    torch.save(d, f)
    back = torch.load(f, map_location="cpu", weights_only=True)          # loadable without executing anything
    assert back["iter"] == 50 and set(back["model_state_dict"]) == set(ac.state_dict())
    ac2 = _ac()
    ac2.load_state_dict(back["model_state_dict"])                        # the reference side: ActorCritic.load_state_dict
    torch.optim.Adam(ac2.parameters()).load_state_dict(back["optimizer_state_dict"])
