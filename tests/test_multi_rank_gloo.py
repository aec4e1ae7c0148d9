"""world_size-2 `gloo` rehearsal of the multi-GPU protocol on CPU, with the torch
oracle standing in for the kernels: envs are sharded by rank; per optimiser step ONE all-reduce(sum)
of the fused [gradients | KL-sum] buffer followed by / world; once per iteration an all-reduce of
[sum adv, sum adv^2, count].  Sharded results must equal the single-process results on the union."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ppo_torch as pt


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch(n, O=6, A=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    return dict(obs=torch.randn(n, O, generator=g), acts=torch.randn(n, A, generator=g), v=torch.randn(n, 1, generator=g),
                adv_raw=torch.randn(n, generator=g) * 2 + 0.5, ret=torch.randn(n, 1, generator=g),
                lp=torch.randn(n, 1, generator=g) - 4, mu=torch.randn(n, A, generator=g) * 0.1)


def _grad_and_kl(ac, b, adv):
    algo = pt.PPO(ac)
    ac.zero_grad()
    loss, kl, _, _ = algo.minibatch_loss(b["obs"], b["obs"], b["acts"], b["v"], adv.unsqueeze(-1), b["ret"], b["lp"], b["mu"],
                                         torch.ones_like(b["mu"]))
    loss.backward()
    return torch.cat([p.grad.reshape(-1) for p in ac.parameters()]), kl * b["obs"].shape[0]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 64
    full = _make_batch(n)
    lo, hi = rank * n // world, (rank + 1) * n // world
    shard = {k: v[lo:hi] for k, v in full.items()}
    # --- advantage normalisation from all-reduced moments
    part = torch.tensor([shard["adv_raw"].sum(), (shard["adv_raw"] ** 2).sum(), float(hi - lo)])
    dist.all_reduce(part)
    cnt, mean = part[2], part[0] / part[2]
    var = (part[1] - cnt * mean * mean) / (cnt - 1)
    adv = (shard["adv_raw"] - mean) / (var.sqrt() + 1e-8)
    # --- fused [grads | kl_sum] all-reduce, then / world  (what k_pre_step / k_grad_norm / k_adam consume)
    torch.manual_seed(1)
    ac = pt.ActorCritic(6, 6, 3, [16, 16, 16], [16, 16, 16])
    for p in ac.parameters():
        dist.broadcast(p.data, src=0)
    grad, kl_sum = _grad_and_kl(ac, shard, adv)
    buf = torch.cat([grad, kl_sum.reshape(1), torch.zeros(1)])
    dist.all_reduce(buf)
    q.put((rank, adv.numpy().copy(), (buf[:-2] / world).detach().numpy().copy(), float(buf[-2] / (world * (hi - lo)))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_protocol_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = _make_batch(64)
    adv_ref = pt.normalize_advantages(full["adv_raw"])
    outs = [(r, torch.from_numpy(a), torch.from_numpy(g), torch.tensor(k)) for r, a, g, k in outs]
    torch.testing.assert_close(torch.cat([outs[0][1], outs[1][1]]), adv_ref, rtol=1e-5, atol=1e-5)
    torch.manual_seed(1)
    ac = pt.ActorCritic(6, 6, 3, [16, 16, 16], [16, 16, 16])
    g_ref, kl_ref = _grad_and_kl(ac, full, adv_ref)
    for r in range(world):
        torch.testing.assert_close(outs[r][2], g_ref, rtol=1e-4, atol=1e-6)     # identical on every rank
        torch.testing.assert_close(outs[r][3], (kl_ref / 64).detach(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(outs[0][2], outs[1][2], rtol=0, atol=0)
