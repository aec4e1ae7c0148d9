#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the full PPO iteration (24-step rollout of the HIP env +
HIP PPO update), ANYmal-C flat, 4096 envs per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one PPO iteration: num_steps_per_env(24) x num_envs env steps (each = 4 physics
substeps + actuator net + post-physics logic) followed by 5 epochs x 4 minibatches of PPO.
value = 24 * num_envs * n_gpus / mean iteration time  (rsl_rl's fps definition).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# host threads for the cpu_baseline leg: the GPU box gives a 1-GPU job a 16-CPU share
CPU_THREADS = max(1, min(16, os.cpu_count() or 1))
os.environ.setdefault("OMP_NUM_THREADS", str(CPU_THREADS))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # fp32-input MFMA dense peak (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TF = 2500.0     # bf16 MFMA dense peak (MI355X_MICROARCH.md); the split-bf16 GEMM issues 6 bf16 MFMAs per
MFMA_X6_PEAK_TF = MFMA_BF16_PEAK_TF / 6.0   # fp32 product block, so its fp32-equivalent ceiling is 2500 / 6 = 416.7 TFLOP/s
BYTES_PER_ENV_STEP = 4200.0    # flat ANYmal, fused-step algorithmic bytes
PMC_FILE = next((f"r{n:02d}_pmc_traffic.json" for n in range(9, 0, -1)        # the latest round's committed PMC digest
                 if os.path.isfile(os.path.join(ROOT, "profiles", f"r{n:02d}_pmc_traffic.json"))), "r01_pmc_traffic.json")


def macs_per_sample(obs, hidden, actions):
    dims_a = [obs] + hidden + [actions]
    dims_c = [obs] + hidden + [1]
    return sum(a * b for a, b in zip(dims_a[:-1], dims_a[1:])) + sum(a * b for a, b in zip(dims_c[:-1], dims_c[1:]))


def make_runner(num_envs, hidden, device, rank, world, task="anymal_c_flat"):
    """env + runner of a registered task as a user builds them (task_registry.get_cfgs: the reference's cfg values; rough-terrain
    tasks on the full 10 x 20 tile terrain), with the policy widths of the bench line."""
    import copy
    import numpy as np
    from legged_gym_dev_amd.envs import task_registry
    from legged_gym_dev_amd.utils.helpers import class_to_dict, get_args, parse_sim_params
    from legged_gym_dev_amd.rl.runner import OnPolicyRunner
    env_cfg, train_cfg = (copy.deepcopy(c) for c in task_registry.get_cfgs(task))
    env_cfg.env.num_envs = num_envs
    env_cfg.seed = 1
    train_cfg.policy.actor_hidden_dims = list(hidden)
    train_cfg.policy.critic_hidden_dims = list(hidden)
    args = get_args([])
    args.sim_device = args.rl_device = device
    torch.manual_seed(1)
    np.random.seed(1)
    sim_params = parse_sim_params(args, {"sim": class_to_dict(env_cfg.sim)})
    kw = {"rank": rank, "world_size": world} if world > 1 else {}
    env = task_registry.task_classes[task](env_cfg, sim_params, args.physics_engine, device, True, **kw)
    runner = OnPolicyRunner(env, class_to_dict(train_cfg), None, device=device)
    return env, runner


class TimedReduce:
    """HIP events around every all-reduce the update issues between backward and optimiser step (TorchDistComm: the collective is
    serial on the learner's stream, so event-to-event time IS the communication time of the minibatch)."""

    def __init__(self, fn):
        self.fn, self.pairs = fn, []

    def __call__(self, t):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.fn(t)
        e1.record()
        self.pairs.append((e0, e1))

    def total_ms(self):
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.pairs)
        n, self.pairs = len(self.pairs), []
        return ms, n


def time_iterations(runner, steps, warmup, world, comm_out=None):
    """(wall seconds of the `steps` timed iterations = MAX over ranks, rollout seconds of as many iterations).  With world > 1 and
    `comm_out` a dict: per-rank communication time of the timed iterations (ms per iteration on every rank: the all-reduces of the
    default collective, or what the learner's stream waited for the overlapped buckets of LG_COMM=native) and every rank's own
    iteration time, so that a scaling loss can be attributed."""
    ar = runner._grad_reduce
    for _ in range(warmup):
        runner.rollout()
        runner.ppo.update(ar)
    timed = None
    native = world > 1 and getattr(runner.comm, "overlapped", False)
    if world > 1 and comm_out is not None:
        if native:
            runner.ppo.comm_wait_ms()                # clear
            runner.ppo.comm_timing(True)
        elif ar is not None:
            timed = ar = TimedReduce(ar)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):                       # timed region: no host synchronisation inside
        runner.rollout()
        runner.ppo.update(ar)
    torch.cuda.synchronize()
    el_own = time.perf_counter() - t0            # this rank's own time, before it waits for the others
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1 and comm_out is not None:
        if native:
            ms, n = runner.ppo.comm_wait_ms()
            runner.ppo.comm_timing(False)
        else:
            ms, n = timed.total_ms() if timed is not None else (0.0, 0)
        ar = runner._grad_reduce
        mine = torch.tensor([ms / steps, 1e3 * el_own / steps, float(n) / steps], device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        comm_out.update({"backend": "native (RCCL through lg_comm_*, per-layer buckets inside the backward pass; ms = the learner stream's wait "
                                    "for the last bucket)" if native else
                                    f"torch.distributed {torch.distributed.get_backend()} (one all-reduce of [gradients | KL] per optimiser step; "
                                    "ms = HIP events around the collective)",
                         "comm_ms_per_step_by_rank": [round(float(v[0]), 4) for v in allr],
                         "iter_ms_by_rank": [round(float(v[1]), 3) for v in allr],
                         "collectives_per_step": round(float(allr[0][2]), 1),
                         "iter_ms_min": round(min(float(v[1]) for v in allr), 3), "iter_ms_max": round(max(float(v[1]) for v in allr), 3)})
    # informative rollout/update split, measured separately with a sync between the phases
    t_roll = 0.0
    for _ in range(2):
        torch.cuda.synchronize()
        a = time.perf_counter()
        runner.rollout()
        torch.cuda.synchronize()
        t_roll += (time.perf_counter() - a) / 2
        runner.ppo.update(ar)
    torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([el, t_roll], device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el, t_roll = float(t[0]), float(t[1])
    return el, t_roll * steps


def gemm_roofline(runner, hidden, with_largest=True, task="anymal_c_flat"):
    """GEMM group of one minibatch (forward + head + backward of actor and critic: everything lg_ppo_minibatch_backward
    launches), timed live with HIP events on the launch stream INSIDE a real update -- begin_update, then epochs x minibatches
    of {backward, optimiser step} as HipPPO.update() runs them, the events bracketing each backward (the optimiser step, which
    also gathers the next minibatch, is outside the brackets; the first minibatch of the update, which gathers for itself, is
    left out of the mean).  Algorithmic (fp32) FLOPs = 2 * MACs * rows * 3.  The kernels compute
    fp32-accurate products as 6 bf16 MFMAs per 32x32x16 block, so the ceiling is the bf16 dense peak / 6; the fp32-input MFMA
    peak (what a v_mfma_f32_32x32x2_f32 kernel is bounded by) is reported beside it."""
    ppo = runner.ppo
    R = ppo.T * ppo.N // ppo.cfg.num_mini_batches
    ppo._call("begin_update")
    pairs = []
    for epoch in range(ppo.cfg.num_epochs):
        for mb in range(ppo.cfg.num_mini_batches):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ppo._call("minibatch_backward", epoch, mb)
            e1.record()
            ppo._call("minibatch_step")
            pairs.append((e0, e1))
    ppo._call("end_update")
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in pairs[1:]) / (len(pairs) - 1)
    flops = 2.0 * macs_per_sample(ppo.O, list(hidden), ppo.A) * R * 3.0
    n_launch = (len(hidden) + 1) * 3 - 1          # fwd + dW per layer, dX for all but the first (actor+critic batched on grid.z)
    ach = flops / (ms * 1e-3) / 1e12
    traffic = None                                  # HBM bytes per minibatch group from the committed PMC passes
    try:
        name = PMC_FILE if task == "anymal_c_flat" else PMC_FILE.replace("_pmc_traffic", f"_{task}_pmc_traffic")
        with open(os.path.join(ROOT, "profiles", name)) as f:
            pm = json.load(f)
        if pm.get("policy_hidden") == list(hidden) and pm.get("task", "anymal_c_flat") == task:
            traffic = pm["gemm_group_bytes_per_minibatch"]
    except (OSError, ValueError, KeyError):
        pass
    big = single_gemm_roofline(R, hidden) if with_largest else None
    out = {"bound": "mfma", "kernel": "k_gemm<split-bf16 x6, mfma_f32_32x32x16_bf16> (ActorCritic fwd+bwd of one minibatch)",
           "achieved": round(ach, 3), "peak": round(MFMA_X6_PEAK_TF, 1), "unit": "TFLOP/s", "frac": round(ach / MFMA_X6_PEAK_TF, 4),
           "traffic": traffic, "flops_per_minibatch": flops, "ms_per_minibatch": round(ms, 4), "gemm_launches": n_launch,
           "largest_launch": big, "mfma_executed_tflops": round(6.0 * ach, 2), "bf16_mfma_peak": MFMA_BF16_PEAK_TF,
           "fp32_input_mfma_peak": MFMA_F32_PEAK_TF, "frac_of_fp32_input_mfma_peak": round(ach / MFMA_F32_PEAK_TF, 4)}
    try:        # `frac` is priced at the nominal 2.4 GHz; on real data the chip runs these kernels slower (committed in-kernel measurement)
        with open(os.path.join(ROOT, "profiles", PMC_FILE.replace("pmc_traffic", "gemm_clock"))) as f:
            ck = float(json.load(f)["clock_ghz_in_kernel_random_operands"])
        out["clock_ghz_in_kernel"] = ck
        out["frac_at_that_clock"] = round(ach / (MFMA_X6_PEAK_TF * ck / 2.4), 4)
    except (OSError, ValueError, KeyError):
        pass
    return out


def single_gemm_roofline(rows, hidden, reps=20):
    """The largest single GEMM launch of the update (forward of the widest hidden layer, one net), alone,
    through the debug entry of the same kernel: flops of that launch / its duration (HIP events)."""
    import ctypes
    from legged_gym_dev_amd.lib import load
    lib = load()
    lib.ppok_debug_gemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    dims = list(hidden)
    l = max(range(len(dims) - 1), key=lambda i: dims[i] * dims[i + 1]) if len(dims) > 1 else 0
    K, N = (dims[l], dims[l + 1]) if len(dims) > 1 else (48, dims[0])
    A = torch.randn(rows, K, device="cuda")
    B = torch.randn(N, K, device="cuda")
    C = torch.empty(rows, N, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.ppok_debug_gemm(ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(B.data_ptr()), ctypes.c_void_p(C.data_ptr()),
                                       rows, N, K, 0, 1, st)
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    tf = 2.0 * rows * N * K / (us * 1e-6) / 1e12
    return {"shape_MNK": [rows, N, K], "us": round(us, 2), "achieved": round(tf, 2), "frac": round(tf / MFMA_X6_PEAK_TF, 4)}


def env_roofline(env, reps=50):
    """lg_step against the two bounds that could apply.  HBM: 4.20 KB algorithmic per env-step -- nowhere near (the step is
    latency-bound at 4096 envs: one workgroup per CU).  VALU issue: the control-loop kernel's critical wave (physics of 16 envs +
    its share of the actuator net) executes `critical_wave_valu_per_lg_step` vector instructions (SQ_INSTS_VALU, committed PMC
    pass profiles/*_substeps_pmc.json) and a wave64 issues at most one per 4 cycles; the clock is the wave's cycle count of the
    same pass over the live duration."""
    import ctypes
    a = torch.zeros(env.num_envs, env.num_actions, device=env.device)
    env.step(a)
    torch.cuda.synchronize()

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms = timed(lambda: env.core.step(a))
    ms_loop = timed(lambda: env.core.lib.lg_debug_control_loop(env.core.ctx, ctypes.c_void_p(a.data_ptr())))
    gbs = BYTES_PER_ENV_STEP * env.num_envs / (ms * 1e-3) / 1e9
    traffic, pmc = None, None
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
            traffic = json.load(f)["env_step_bytes_per_call"]
        with open(os.path.join(ROOT, "profiles", PMC_FILE.replace("pmc_traffic", "substeps_pmc"))) as f:
            pmc = json.load(f)
    except (OSError, ValueError, KeyError):
        pass
    out = {"bound": "valu-issue", "kernel": "k_substeps<4,3,true,true> (clip + 4 x {actuator net, pair-lane ABA + contact}): one launch per policy step",
           "us_per_launch": round(ms_loop * 1e3, 2), "us_per_lg_step": round(ms * 1e3, 2),
           "hbm": {"achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                   "algorithmic_bytes_per_env_step": BYTES_PER_ENV_STEP}}
    if pmc:
        # VALU issue rate of the critical wave (a physics wave: it executes the most instructions and lives for the whole launch) against one
        # instruction per 4 cycles of its SIMD.  The clock is the IN-KERNEL one, measured with s_memtime / s_memrealtime stamps in the section-
        # timing build (profiles/*_substeps_clock.json): 2.36 GHz -- SQ_WAVE_CYCLES x 4 / duration, used until late in round 3, reads 1.63 GHz for
        # the same launch and overstated the fraction (0.63 instead of 0.44).
        valu = pmc["critical_wave_valu_per_lg_step"]
        clock_ghz, clock_src = 2.4, "nominal 2.4 GHz (no committed in-kernel measurement found)"
        try:
            with open(os.path.join(ROOT, "profiles", PMC_FILE.replace("pmc_traffic", "substeps_clock"))) as f:
                clock_ghz = float(json.load(f)["clock_ghz_in_kernel"])
            clock_src = "in-kernel clock: committed s_memtime / s_memrealtime stamps (profiles/" + PMC_FILE.replace("pmc_traffic", "substeps_clock") + ")"
        except (OSError, ValueError, KeyError):
            pass
        cycles = clock_ghz * 1e9 * ms_loop * 1e-3
        out.update({"achieved": round(valu / (ms_loop * 1e-3) / 1e6, 1), "peak": round(clock_ghz * 1e3 / 4.0, 1), "unit": "M VALU instr/s per SIMD",
                    "frac": round(4.0 * valu / cycles, 4), "critical_wave_valu": valu, "launch_cycles": round(cycles), "clock_ghz": clock_ghz,
                    "source": "instruction counts: committed SQ counter pass (profiles/" + PMC_FILE.replace("pmc_traffic", "substeps_pmc")
                              + "); " + clock_src + "; duration: HIP events in this run"})
    return out


def cpu_baseline(num_envs, hidden):
    """Own CPU restatement (oracle/: C++ env step with OpenMP + torch PPO) timed on ONE WHOLE PPO iteration of the same
    workload: the 24-step rollout on `num_envs` envs (policy forward + env step) and the full update, 5 epochs x 4
    minibatches of forward + loss + backward + clip + Adam.  kind = "port": PhysX / rsl_rl are absent, this is not the
    reference's CPU path."""
    import numpy as np
    from oracle import oracle_lib, ppo_torch
    from legged_gym_dev_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
    from legged_gym_dev_amd.envs.base.env_setup import EnvSetup, sim_dt_float
    from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model
    cores = CPU_THREADS
    torch.set_num_threads(cores)
    cfg = AnymalCFlatCfg()
    cfg.env.num_envs = num_envs
    cm = compile_model(resolve_model("", "anymal_c"))
    env = oracle_lib.OracleEnv(EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt), seed=1))
    env.call("reset_all")
    T, A, O, E, MB = 24, 12, 48, 5, 4
    ac = ppo_torch.ActorCritic(O, O, A, hidden, hidden)
    algo = ppo_torch.PPO(ac)
    st_obs, st_act, st_mu = torch.zeros(T, num_envs, O), torch.zeros(T, num_envs, A), torch.zeros(T, num_envs, A)
    t0 = time.perf_counter()
    with torch.no_grad():
        for t in range(T):
            obs = torch.from_numpy(env.buf["obs"]).clone()
            act = ac.act(obs)
            ac.evaluate(obs)
            st_obs[t], st_act[t], st_mu[t] = obs, act, ac.action_mean
            env.step(act.numpy())
    t_roll = time.perf_counter() - t0
    R = T * num_envs // MB
    o, a, m = st_obs.reshape(T * num_envs, O), st_act.reshape(T * num_envs, A), st_mu.reshape(T * num_envs, A)
    perm = torch.randperm(T * num_envs)
    t0 = time.perf_counter()
    for _ in range(E):
        for mb in range(MB):
            idx = perm[mb * R:(mb + 1) * R]
            algo.step_minibatch(o[idx], o[idx], a[idx], torch.zeros(R, 1), torch.randn(R, 1), torch.randn(R, 1),
                                torch.zeros(R, 1), m[idx], torch.ones(R, A))
    t_upd = time.perf_counter() - t0
    env.close()
    total = t_roll + t_upd
    return {"value": round(T * num_envs / total, 1), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"one whole PPO iteration: {T} rollout steps x {num_envs} envs (oracle C++/OpenMP env + torch policy: {t_roll:.2f}s) "
                      f"+ {E * MB} minibatch updates of {R} rows ({t_upd:.2f}s); {cores} threads"}


def lg_step_us(env, reps=50):
    """us per lg_step (control loop + post-step + finalize) on the env's current state, HIP events on the launch stream."""
    a = torch.zeros(env.num_envs, env.num_actions, device=env.device)
    env.core.step(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        env.core.step(a)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def other_config(task, num_envs, hidden, device, steps=6, warmup=3, sustained=0):
    """BASELINE.json configs[2] / configs[4] beside the bench line: the same whole-iteration measurement on another task."""
    env, runner = make_runner(num_envs, hidden, device, 0, 1, task=task)
    try:
        env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        el, t_roll = time_iterations(runner, steps, warmup, 1)
        rf = gemm_roofline(runner, hidden, with_largest=False, task=task)
        ep_done = int(runner.ppo.t["ep_ring_count"].cpu()) & 0xFFFFFFFF
        iters = steps + warmup + 2 + 1                     # timed + warm-up + the two split iterations + the roofline's update
        out = {"task": task, "num_envs": num_envs, "num_obs": env.num_obs, "policy_hidden": list(hidden),
               "terrain": (f"{env.cfg.terrain.mesh_type} {env.cfg.terrain.num_rows}x{env.cfg.terrain.num_cols} tiles, "
                           f"{env.terrain.tot_rows}x{env.terrain.tot_cols} height samples") if env.terrain is not None else "plane",
               "value": round(runner.num_steps_per_env * num_envs * steps / el, 1), "unit": "env-steps/s",
               "ms_per_step": round(1e3 * el / steps, 3), "rollout_ms": round(1e3 * t_roll / steps, 3),
               "update_ms": round(1e3 * (el - t_roll) / steps, 3),
               "roofline": {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "ms_per_minibatch", "flops_per_minibatch")},
               "us_per_lg_step": round(lg_step_us(env), 2),
               "resets_per_env_step": round(ep_done / max(runner.num_steps_per_env * num_envs * (iters - 1), 1), 5),
               "physics_fault_resets": int(env.fault_total.cpu()), "base_velocity_clamps": int(env.vel_clamp_total.cpu())}
        if sustained > 0:
            out["sustained"] = sustained_pass(env, runner, hidden, sustained, runner.num_steps_per_env * num_envs, ep_done, task)
        return out
    finally:
        env.close()
        runner.ppo.close()


def sustained_pass(env, runner, hidden, iters, steps_per_iter, ep_done, task):
    """`iters` more iterations of the same loop with no host synchronisation inside: the timed K steps are a burst of a few hundred
    ms, a training run settles at the power / thermal state of minutes of load.  Two probes separate the chip's state from the
    workload's drift: the largest GEMM of the update on FIXED synthetic operands before and after the pass (same data, same launch:
    its change is clocks alone), and the GEMM group of a real update on the storage the pass leaves behind."""
    R = runner.ppo.T * runner.ppo.N // runner.ppo.cfg.num_mini_batches
    faults0, clamps0 = int(env.fault_total.cpu()), int(env.vel_clamp_total.cpu())
    fixed0 = single_gemm_roofline(R, hidden)["us"]
    ar = runner._grad_reduce
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        runner.rollout()
        runner.ppo.update(ar)
    torch.cuda.synchronize()
    s_el = time.perf_counter() - t0
    fixed1 = single_gemm_roofline(R, hidden)["us"]
    ep2 = int(runner.ppo.t["ep_ring_count"].cpu()) & 0xFFFFFFFF
    return {"iterations": iters, "value": round(steps_per_iter * iters / s_el, 1), "unit": "env-steps/s",
            "ms_per_step": round(1e3 * s_el / iters, 3), "seconds": round(s_el, 2),
            "resets_per_env_step": round((ep2 - ep_done) / (steps_per_iter * iters), 5),
            "physics_fault_resets": int(env.fault_total.cpu()) - faults0,
            "base_velocity_clamps": int(env.vel_clamp_total.cpu()) - clamps0,
            "policy_std_at_end": round(float(runner.ppo.param_views["std"].mean()), 3),
            "gemm_fixed_operands_us_before": fixed0, "gemm_fixed_operands_us_after": fixed1,
            "power_state_slowdown": round(fixed1 / fixed0, 4),
            "gemm_ms_per_minibatch_at_end": gemm_roofline(runner, hidden, with_largest=False, task=task)["ms_per_minibatch"]}


def rank_environments(n, port, base=None, comm_port=None):
    """Environment of each of the n child ranks (what torch.distributed.run would export)."""
    base = dict(os.environ if base is None else base)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this driver
    if comm_port is not None:
        base["LG_COMM_PORT"] = str(comm_port)                   # NativeComm's id exchange (rl/comm.py): a port probed free as well
    return [dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)) for r in range(n)]


def visible_gpu_count():
    """GPUs this process would see, WITHOUT a HIP / torch.cuda call (torch.cuda.device_count() can initialise the HIP runtime
    in the launcher parent, whose children must start from a GPU-clean process): KFD topology nodes with SIMDs, narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None if the topology is not readable (then a child rank fails instead)."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for p in nodes:
        try:
            with open(p) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        except OSError:
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n, argv):
    """One process per GPU, started from a parent that has made no GPU call; rank 0 prints the JSON line.
    Returns the exit code: 0 only if every rank ended cleanly; the others are stopped when one fails."""
    import socket
    import subprocess
    have = visible_gpu_count()
    if os.environ.get("LG_BENCH_SHARE_GPU") != "1" and have is not None and have < n:
        print(f"bench.py: --gpus {n} but {have} visible", file=sys.stderr)
        return 2
    with socket.socket() as s, socket.socket() as s2:
        s.bind(("127.0.0.1", 0))
        s2.bind(("127.0.0.1", 0))
        port, comm_port = s.getsockname()[1], s2.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e)
             for e in rank_environments(n, port, comm_port=comm_port)]
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            r = p.poll()
            if r is None:
                continue
            pending.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in pending:
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--num_envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--hidden", type=str, default="512,256,128")
    ap.add_argument("--task", type=str, default="anymal_c_flat", help="registered task of the main line (profiling other configs)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_alt", action="store_true", help="skip the [128,64,32] pass (profiling runs)")
    ap.add_argument("--no_other", action="store_true", help="skip the anymal_c_rough / cassie passes (BASELINE configs[2], [4])")
    ap.add_argument("--sustained", type=int, default=300, help="iterations of the sustained-speed pass after the timed steps (0: skip)")
    args = ap.parse_args()
    hidden = [int(v) for v in args.hidden.split(",")]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process has not touched the GPU yet (importing torch does not);
        # start one fresh child per device and leave with their exit code
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # rehearsal knobs (1-GPU box): LG_BENCH_BACKEND=gloo LG_BENCH_SHARE_GPU=1 run all ranks on cuda:0 over gloo
    backend = os.environ.get("LG_BENCH_BACKEND", "nccl")
    if os.environ.get("LG_BENCH_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if world > 1:        # the process group also serves the barrier / MAX of the timing contract when LG_COMM=native reduces the gradients
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device(device))
        else:
            torch.distributed.init_process_group(backend)
    if os.environ.get("LG_BENCH_FAIL_RANK") == str(rank) and world > 1:      # test hook: a rank that dies must fail the job
        raise SystemExit(3)
    env, runner = make_runner(args.num_envs, hidden, device, rank, world, task=args.task)
    if args.task != "anymal_c_flat":
        env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        args.no_other = args.no_alt = args.no_cpu_baseline = True
    comm_info = {} if world > 1 else None
    el, t_roll = time_iterations(runner, args.steps, args.warmup, world, comm_out=comm_info)
    steps_per_iter = runner.num_steps_per_env * args.num_envs * world
    sustained = None
    # what the timed steps were measured on: how often the random-init policy makes robots fall, and how often the
    # physics fault guard fired (a counted reset of an env whose substep diverged: 0 for ANYmal under the random-init policy, one or
    # two per 2.4 M env-steps for Cassie, whose fixed-gain PD law meets its joint limits in falls)
    ep_done = int(runner.ppo.t["ep_ring_count"].cpu()) & 0xFFFFFFFF
    steps_done = runner.num_steps_per_env * args.num_envs * (args.steps + args.warmup + 2)
    faults = int(env.fault_total.cpu())
    label = "ANYmal-C flat" if args.task == "anymal_c_flat" else args.task
    out = {"metric": f"env-steps/sec (whole node), {label} {args.num_envs} envs per GPU", "value": round(steps_per_iter * args.steps / el, 1),
           "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * el / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{args.task}, {args.num_envs} envs/GPU, 24 env steps/iter (decimation 4, "
                                  f"{'actuator LSTM' if env.setup.use_actuator_net else 'PD law'}, "
                                  f"ABA+contact), PPO 5 epochs x 4 minibatches, ActorCritic {hidden}",
                      "num_envs_per_gpu": args.num_envs, "policy_hidden": hidden,
                      "parallelism": f"env-shard x{world}" + (f", LG_COMM={os.environ.get('LG_COMM', 'torch')}" if world > 1 else ""),
                      "rollout_ms": round(1e3 * t_roll / args.steps, 3),
                      "update_ms": round(1e3 * (el - t_roll) / args.steps, 3)}}
    out["config"]["resets_per_env_step"] = round(ep_done / max(steps_done, 1), 5)
    out["config"]["physics_fault_resets"] = faults
    out["config"]["base_velocity_clamps"] = int(env.vel_clamp_total.cpu())
    if comm_info:
        out["comm"] = comm_info
    if world > 1:
        # the GEMM-group probe on every rank (a communicator that reduces inside the backward pass needs all of them in it); rank 0 reports
        rf = gemm_roofline(runner, hidden, with_largest=False, task=args.task)
        if rank == 0:
            rf["traffic_source"] = "committed single-GPU PMC passes (profiles/" + PMC_FILE + "); not collected during this run"
            out["roofline"] = rf
    if rank == 0 and world == 1:
        out["roofline"] = gemm_roofline(runner, hidden, task=args.task)
        out["roofline"]["traffic_source"] = ("committed PMC passes of this command (profiles/" + PMC_FILE + ": rocprofv3 --pmc FETCH_SIZE / "
                                             "WRITE_SIZE in separate runs); not collected during this run")
        out["roofline_env_step"] = env_roofline(env) if args.task == "anymal_c_flat" else {"us_per_lg_step": round(lg_step_us(env), 2)}
        if args.sustained > 0:   # after the probes above: they are taken in the state the timed steps ran in
            sustained = sustained_pass(env, runner, hidden, args.sustained, steps_per_iter, ep_done, args.task)
            sustained["note"] = ("the fork's anymal_c_flat reward is identically 0 after its positive clip (commands x, y = 0: SURVEY.md 0.8), so over "
                                 "hundreds of iterations PPO's entropy bonus alone inflates the policy's std and its actions (|a| up to the clip of "
                                 "100): robots thrash and resets per env-step rise.  power_state_slowdown (one GEMM launch on fixed synthetic operands "
                                 "before / after the pass) is the chip's clocks alone; what remains of the gap to the burst figure is that drift of "
                                 "the workload.  other_configs[anymal_c_rough].sustained is the same pass on a task whose reward has a signal")
            out["sustained"] = sustained
        env.close()
        runner.ppo.close()
        if not args.no_other:
            out["other_configs"] = [other_config(t, args.num_envs, hidden, device, sustained=(args.sustained // 2 if t == "anymal_c_rough" else 0))
                                    for t in ("anymal_c_rough", "cassie")]
        if tuple(hidden) != (128, 64, 32) and not args.no_alt:
            env2, runner2 = make_runner(args.num_envs, [128, 64, 32], device, rank, world)
            el2, tr2 = time_iterations(runner2, max(3, args.steps // 2), 2, world)
            n2 = max(3, args.steps // 2)
            out["alt_reference_policy_dims"] = {"policy_hidden": [128, 64, 32], "value": round(steps_per_iter * n2 / el2, 1),
                                                "ms_per_step": round(1e3 * el2 / n2, 3), "rollout_ms": round(1e3 * tr2 / n2, 3)}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.num_envs, hidden)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
