"""GPU: the multi-GPU path of the PRODUCT code on a one-GPU box.

Two fresh processes (tests/multi_rank_worker.py), rank 0 and 1 of world 2, share cuda:0 and reduce over gloo: each builds
its env shard with task_registry.make_env(rank=, world_size=) and trains the real OnPolicyRunner.  Checked: the ranks hold
different env shards (constants, origins, observations), identical parameters bit for bit after training, and the same
parameters as an in-process emulation in which the two ranks are threads and the all-reduce is an explicit sum
(tolerance: the kernels accumulate with float atomics, so two runs of one rank already differ in the last bits).
bench.py's own launcher (`python bench.py --gpus 2`, no torchrun) is exercised the same way.
"""
import json
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(outdir, task, n, iters, world=2):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multi_rank_worker.py"), str(outdir), task,
                                       str(n), str(iters)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    return [dict(np.load(os.path.join(outdir, f"rank{r}.npz"))) for r in range(world)]


class _ThreadComm:
    """Ranks as threads of this process; all_reduce = explicit sum of the ranks' buffers.  While the runners are being
    constructed (one after the other, on the main thread: construction draws from the process-wide CPU generators) the
    initial broadcast is a direct copy from rank 0's parameters."""

    def __init__(self, rank, shared):
        self.rank, self.world_size, self.sh = rank, shared["world"], shared

    def all_reduce(self, t):
        torch.cuda.synchronize()
        self.sh["bufs"][self.rank] = t
        self.sh["barrier"].wait()
        if self.rank == 0:
            bufs = self.sh["bufs"]
            tot = bufs[0].clone()
            for b in bufs[1:]:
                tot += b
            for b in bufs:
                b.copy_(tot)
            torch.cuda.synchronize()
        self.sh["barrier"].wait()

    def broadcast(self, t, src=0):
        assert self.sh["phase"] == "build" and src == 0
        if self.rank == 0:
            self.sh["src"] = t
        else:
            t.copy_(self.sh["src"])
        torch.cuda.synchronize()


def _emulate(task, n, iters, world=2):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import multi_rank_worker as w
    shared = {"world": world, "bufs": [None] * world, "barrier": threading.Barrier(world), "phase": "build"}
    built = [w.build(task, n, r, world, comm=_ThreadComm(r, shared)) for r in range(world)]
    shared["phase"] = "learn"
    errs = []
    firsts = [None] * world

    def learn(r):
        try:
            torch.cuda.set_device(0)
            firsts[r] = w.learn_with_first_snapshot(built[r][1], iters)
        except BaseException as e:          # noqa: BLE001
            errs.append(e)
            shared["barrier"].abort()
    ths = [threading.Thread(target=learn, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(600)
    if errs:
        raise errs[0]
    snaps = [dict(w.snapshot(*built[r]), params_it1=firsts[r]) for r in range(world)]
    for env, runner in built:
        env.close()
        runner.ppo.close()
    return snaps


@pytest.mark.parametrize("task", ["anymal_c_flat", "anymal_c_rough"])
def test_two_rank_runner_over_gloo(task, tmp_path):
    n, iters = 64, 2
    ranks = _run_ranks(tmp_path, task, n, iters)
    a, b = ranks
    # disjoint shards: env offsets, constants, Philox streams
    assert int(a["env_offset"]) == 0 and int(b["env_offset"]) == n
    assert not np.array_equal(a["friction"], b["friction"])
    assert not np.array_equal(a["base_mass_delta"], b["base_mass_delta"]) or task == "anymal_c_rough"
    assert not np.array_equal(a["env_origins"], b["env_origins"])
    assert not np.array_equal(a["obs"], b["obs"]) and not np.array_equal(a["commands"], b["commands"])
    # one policy: identical on both ranks, bit for bit (parameters, Adam state, KL-adapted learning rate, advantage mean)
    np.testing.assert_array_equal(a["params"], b["params"])
    np.testing.assert_array_equal(a["adam_m"], b["adam_m"])
    assert float(a["lr"]) == float(b["lr"]) and float(a["adv_mean"]) == float(b["adv_mean"])
    assert np.isfinite(a["params"]).all() and int(a["fault_total"][0]) == 0 and int(b["fault_total"][0]) == 0
    # same result as the in-process emulation (ranks = threads, all_reduce = explicit sum)
    emu = _emulate(task, n, iters)
    np.testing.assert_array_equal(emu[0]["params"], emu[1]["params"])
    for r in range(2):                                    # the shards themselves are reproduced exactly
        for key in ("friction", "base_mass_delta", "env_origins"):
            np.testing.assert_array_equal(emu[r][key], ranks[r][key], err_msg=f"rank {r} {key}")
    # The kernels accumulate gradients, bias sums and loss statistics with float atomics, so two runs of the SAME code differ
    # in the last bits of every gradient.  After ONE iteration (24 policy steps on identical parameters, 20 optimiser steps)
    # that is all there is: process run and emulation agree to 1e-7 of the parameter norm (tools/diag_multi_rank_noise.py:
    # 8e-8 .. 9e-8 between any two of three process runs and three emulations).
    rel1 = np.linalg.norm(emu[0]["params_it1"] - a["params_it1"]) / np.linalg.norm(a["params_it1"])
    assert rel1 < 1e-5, rel1
    np.testing.assert_array_equal(a["params_it1"], b["params_it1"])
    # The second rollout runs on parameters that differ by those 1e-7, and contact dynamics turn that into discrete events
    # (a foot that touches down one substep later, a reset one step later): runs of the same kind then fall on branches
    # 1.5e-4 .. 8.5e-4 apart and agree to 2e-6 .. 1e-5 inside a branch (same tool: process runs {0, 2} and {1}, emulations
    # {0, 2} and {1}, process run 1 with emulation 1).  Emulation-vs-emulation spread is therefore no noise floor for this
    # comparison; after two iterations only the order of magnitude is asserted, and element-wise that no parameter is further
    # off than a few sign steps of Adam (a step on a near-zero gradient is lr * sign(g)).
    # (observed up to 7e-3 on rough terrain; the exact comparison of this path is the deterministic-mode test below, where the two runs
    # must agree bit for bit -- here only a gross error of the collective, e.g. a dropped or doubled bucket, is excluded)
    rel2 = np.linalg.norm(emu[0]["params"] - a["params"]) / np.linalg.norm(a["params"])
    assert rel2 < 3e-2, rel2
    d = np.abs(emu[0]["params"] - a["params"])
    bad = d > 2e-4 * np.abs(a["params"]) + 4.0 * float(a["lr"])
    assert bad.mean() < 3e-2, bad.mean()


@pytest.mark.parametrize("task", ["anymal_c_flat", "anymal_c_rough"])
def test_two_ranks_in_deterministic_mode_equal_the_emulation_bit_for_bit(task, tmp_path, monkeypatch):
    """With LG_DETERMINISTIC=1 (lg_ppo_set_deterministic: fixed-point accumulation instead of float atomics) the comparison the
    test above can only make to an order of magnitude becomes exact: two processes over gloo and the in-process emulation hold
    the same parameters, Adam moments and learning rate after two iterations, bit for bit -- on the flat task (BASELINE configs[1]:
    actuator net, 48 observations) and on rough terrain.  A wrong `/world` on one layer or a dropped bucket cannot pass here."""
    monkeypatch.setenv("LG_DETERMINISTIC", "1")
    n, iters = 64, 2
    a, b = _run_ranks(tmp_path, task, n, iters)
    emu = _emulate(task, n, iters)
    np.testing.assert_array_equal(a["params"], b["params"])
    for key in ("params_it1", "params", "adam_m"):
        np.testing.assert_array_equal(emu[0][key], a[key], err_msg=key)
    assert float(emu[0]["lr"]) == float(a["lr"]) and float(emu[0]["adv_mean"]) == float(a["adv_mean"])
    for r in range(2):
        for key in ("obs", "root_states", "episode_length"):
            np.testing.assert_array_equal(emu[r][key], (a, b)[r][key], err_msg=f"rank {r} {key}")


def test_default_collective_over_rccl_with_one_rank(tmp_path):
    """The product's default collective is torch.distributed on the "nccl" backend (= RCCL).  A one-GPU box allows one RCCL
    rank: that rank runs the real runner with every all-reduce / broadcast going through the RCCL group on the learner's
    stream, between the library's kernels, and must end where the same run without the group ends.  (More than one RCCL
    rank has never run: DESIGN.md section 6.)"""
    port = _free_port()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0", LG_TEST_MODE="rccl_one_rank")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multi_rank_worker.py"), str(tmp_path), "anymal_c_flat", "64", "1"],
                       env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    r = dict(np.load(os.path.join(tmp_path, "rccl_one_rank.npz")))
    a, b = r["params_it1_group"], r["params_it1_plain"]
    assert int(r["calls"]) >= 21                         # one per optimiser step + the advantage moments
    assert np.isfinite(a).all() and int(r["fault_group"][0]) == 0 and int(r["fault_plain"][0]) == 0
    rel = np.linalg.norm(a - b) / np.linalg.norm(b)      # float-atomic noise of one iteration (see the gloo test below)
    assert rel < 1e-5, rel


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` as the driver calls it (no torchrun): the parent starts one child per rank before it has
    made any GPU call; rank 0 prints the JSON line with n_gpus = 2.  Both ranks on cuda:0 over gloo here."""
    env = dict(os.environ, LG_BENCH_BACKEND="gloo", LG_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--num_envs", "256", "--hidden", "64,32"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["roofline"]["bound"] == "mfma" and 0.0 < out["roofline"]["frac"] < 1.0     # the GEMM-group probe also runs with N > 1
    assert out["config"]["num_envs_per_gpu"] == 256 and "LG_COMM=torch" in out["config"]["parallelism"]
    # per-rank communication time and iteration time, so that a scaling loss on a real node can be attributed
    cm = out["comm"]
    assert len(cm["comm_ms_per_step_by_rank"]) == 2 and len(cm["iter_ms_by_rank"]) == 2 and cm["collectives_per_step"] == 20.0
    assert all(0.0 < v < out["ms_per_step"] for v in cm["comm_ms_per_step_by_rank"])
    assert cm["iter_ms_min"] <= cm["iter_ms_max"] <= out["ms_per_step"] * 1.05
    # a rank that dies takes the job down with a non-zero code
    env["LG_BENCH_FAIL_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--num_envs", "64", "--hidden", "64,32"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
