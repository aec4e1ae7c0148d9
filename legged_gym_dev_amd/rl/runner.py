"""OnPolicyRunner with rsl_rl's surface (constructor call site legged_gym/utils/task_registry.py:148,
``learn`` scripts/train.py:44, ``load/get_inference_policy/alg.actor_critic`` scripts/play.py:70-86),
driving the HIP env step and the HIP PPO learner.  Semantics per SURVEY.md Appendix B.

Multi-GPU: one process per GPU (torchrun); envs are sharded, the fused [gradients | KL] buffer is
all-reduced over RCCL once per optimiser step and the advantage moments once per iteration.
The rollout loop issues only asynchronous C-ABI calls; the host synchronises once per iteration
(for the timers that define the env-steps/s metric).
"""
import os
import statistics
import time
from collections import deque

import torch

from legged_gym_dev_amd.capi import NUM_TERMS as _NUM_TERMS
from .comm import TorchDistComm, default_comm  # noqa: F401  (TorchDistComm re-exported)
from .ppo import HipPPO


class _ActorFacade(torch.nn.Module):
    """`alg.actor_critic.actor(obs)` -> action means (play.py:70-86)."""

    def __init__(self, ppo):
        super().__init__()
        self._ppo = ppo

    def forward(self, obs):
        return self._ppo.act_inference(obs)


class ActorCriticFacade:
    def __init__(self, ppo):
        self._ppo = ppo
        self.actor = _ActorFacade(ppo)
        self.activation = ppo.activation

    @property
    def std(self):
        return self._ppo.param_views["std"]

    def state_dict(self):
        return self._ppo.state_dict()

    def load_state_dict(self, sd):
        self._ppo.load_state_dict(sd)

    def act_inference(self, obs):
        return self._ppo.act_inference(obs)

    def eval(self):
        return self

    def train(self):
        return self

    def to(self, device):
        return self


class _Alg:
    def __init__(self, ppo):
        self.ppo = ppo
        self.actor_critic = ActorCriticFacade(ppo)

    @property
    def learning_rate(self):
        return self.ppo.learning_rate


class OnPolicyRunner:
    def __init__(self, env, train_cfg, log_dir=None, device="cuda:0", wandb_callback=None, comm=None):
        self.cfg = train_cfg["runner"]
        self.alg_cfg, self.policy_cfg = train_cfg["algorithm"], train_cfg["policy"]
        self.device = str(device).replace("hip", "cuda")
        self.env = env
        self.wandb_callback = wandb_callback
        if comm is None:
            comm = default_comm()
        self.comm = comm if comm is not None and comm.world_size > 1 else None
        self.world_size = self.comm.world_size if self.comm else 1
        self.rank = self.comm.rank if self.comm else 0
        if getattr(env, "world_size", 1) != self.world_size or getattr(env, "rank", 0) != self.rank:
            raise RuntimeError(f"env was built for rank {getattr(env, 'rank', 0)} of {getattr(env, 'world_size', 1)} but the process "
                               f"group says rank {self.rank} of {self.world_size}: pass rank/world_size to make_env so that every "
                               "rank owns its own env shard (env_offset, constants, Philox streams)")
        if str(torch.device(self.device)) != str(torch.device(env.device)):
            raise RuntimeError(f"rl_device {self.device} != sim_device {env.device}: the HIP learner consumes the env's "
                               "HBM buffers in place")
        self.num_steps_per_env = self.cfg["num_steps_per_env"]
        self.save_interval = self.cfg["save_interval"]
        ncrit = env.num_privileged_obs if env.num_privileged_obs is not None else env.num_obs
        self.ppo = HipPPO(env.num_envs, env.num_obs, ncrit, env.num_actions, self.policy_cfg, self.alg_cfg,
                          self.num_steps_per_env, device=self.device, seed=train_cfg.get("seed", 1),
                          world_size=self.world_size, rank=self.rank)
        self.alg = _Alg(self.ppo)
        if self.comm:                                 # identical initial policy on every rank
            torch.cuda.synchronize()
            self.comm.broadcast(self.ppo.t["params"], src=0)
            self.ppo.params_changed()
            if hasattr(self.comm, "attach"):
                self.comm.attach(self.ppo)            # NativeComm: gradients reduced inside the backward pass from here on
        self.fuse_epilogue = os.environ.get("LG_FUSE_EPILOGUE", "1") != "0"
        self.log_dir = log_dir
        self.tot_timesteps, self.tot_time, self.current_learning_iteration = 0, 0.0, 0
        self.rewbuffer, self.lenbuffer = deque(maxlen=100), deque(maxlen=100)
        self.last_fps = 0.0
        _, _ = self.env.reset()

    def _all_reduce(self, t):
        self.comm.all_reduce(t)

    @property
    def _grad_reduce(self):
        """What ppo.update calls between backward and step: nothing when the communicator reduces inside the backward pass."""
        if self.world_size == 1 or getattr(self.comm, "overlapped", False):
            return None
        return self._all_reduce

    # ------------------------------------------------------------------
    def rollout(self):
        """24 x {act, env.step, process_env_step} + compute_returns.  For the duration of the rollout the env is attached to the
        learner (lg_ppo_attach_env): the step's single-workgroup epilogue and process_env_step ride on the next act's launch --
        3 launches per policy step instead of 5, same results (LG_FUSE_EPILOGUE=0 keeps them as launches of their own)."""
        env, ppo = self.env, self.ppo
        obs = env.get_observations()
        fuse = self.fuse_epilogue
        if fuse:
            ppo.attach_env(env.core)
        try:
            for _ in range(self.num_steps_per_env):
                actions = ppo.act(obs, None)
                obs, priv, rewards, dones, infos = env.step(actions)
                ppo.process_env_step(rewards, env.core.t["reset"], {"time_outs": env.core.t["extras_time_outs"]}
                                     if "time_outs" in infos else {})
            ppo.compute_returns(obs, self._all_reduce if self.world_size > 1 else None)
        finally:
            if fuse:
                ppo.attach_env(None)

    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        """rsl_rl's OnPolicyRunner.learn: rollout, update, one log block per iteration, checkpoints every save_interval.

        The log block needs a handful of device values (episode ring, losses, lr, std, per-term episode sums).  Fetching and
        printing them between iterations leaves the GPU idle for ~1 ms of every 14 (tools/cpu_enqueue_time.py), so by default the
        iteration's values go out as ONE packed asynchronous copy at the end of its update, its phase times come from HIP events
        on the learner's stream, and its block is printed while the GPU is already working on the next iteration: the host never
        waits for an idle GPU.  LG_LOG_SYNC=1 (and any wandb_callback, which is handed live parameters) keeps rsl_rl's
        synchronise-measure-print order."""
        env, ppo = self.env, self.ppo
        if init_at_random_ep_len:
            env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        tot_iter = self.current_learning_iteration + num_learning_iterations
        deferred = self.wandb_callback is None and os.environ.get("LG_LOG_SYNC", "0") in ("", "0")
        pending = []

        def flush():
            while pending:
                self._log(*pending.pop(0))

        for it in range(self.current_learning_iteration, tot_iter):
            if deferred:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
                if pending:                              # the previous iteration's learning phase ends where this iteration begins: the
                    pending[-1][2][2] = ev[0]            # log snapshot, its copy and any host gap count, and the times add up to wall time
                self.rollout()
                ev[1].record()
                losses = ppo.update(self._grad_reduce)
                snap = self._log_snapshot(*losses)
                ev[2].record()                           # (last iteration: after the snapshot's copy and the accumulator clears)
                pending.append((it, tot_iter, ev, snap))
                while len(pending) > 1:                  # the previous iteration's block, beside this iteration's GPU work
                    self._log(*pending.pop(0))
            else:
                torch.cuda.synchronize()
                start = time.time()
                self.rollout()
                torch.cuda.synchronize()
                stop = time.time()
                losses = ppo.update(self._grad_reduce)
                torch.cuda.synchronize()
                self._log(it, tot_iter, (stop - start, time.time() - stop), self._log_snapshot(*losses))
            if self.log_dir is not None and self.rank == 0 and it % self.save_interval == 0:
                flush()
                self.save(os.path.join(self.log_dir, f"model_{it}.pt"))
        flush()
        torch.cuda.synchronize()
        self.current_learning_iteration += num_learning_iterations
        if self.log_dir is not None and self.rank == 0:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))

    def _log_snapshot(self, vloss, sloss):
        """Everything one log block reads from the device, as one float64 vector copied to pinned host memory without blocking
        (layout: episode ring 200 | episodes finished | mean std | learner stats 8 | losses 2 | per-term episode sums + step count |
        physics faults); the per-iteration accumulators are cleared behind the copy, in stream order.  Returns (host vector, event)."""
        ppo, t = self.ppo, self.env.core.t
        parts = [ppo.t["ep_ring"].reshape(-1), ppo.t["ep_ring_count"].reshape(-1)[:1], ppo.param_views["std"].mean().reshape(1),
                 ppo.t["stats"].reshape(-1)[:8], vloss.reshape(1), sloss.reshape(1), t["extras_episode_acc"].reshape(-1),
                 t["fault_total"].reshape(-1)[:1]]
        dev = torch.cat([p.to(torch.float64) for p in parts])
        if getattr(self, "_log_host", None) is None or self._log_host[0].numel() != dev.numel():
            self._log_host = [torch.empty(dev.numel(), dtype=torch.float64).pin_memory() for _ in range(2)]
            self._log_flip = 0
        host = self._log_host[self._log_flip]
        self._log_flip ^= 1
        host.copy_(dev, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        ppo.t["ep_stats"].zero_()
        t["extras_episode_acc"].zero_()
        return host, done, int(t["extras_episode_acc"].numel())

    def _log(self, it, tot_iter, times, snap, width=80, pad=35):
        ppo = self.ppo
        host, done, nacc = snap
        done.synchronize()
        if isinstance(times, list):                       # HIP events on the learner's stream
            times[2].synchronize()                        # (the next iteration's start event: recorded right behind the snapshot)
            collection_time, learn_time = times[0].elapsed_time(times[1]) * 1e-3, times[1].elapsed_time(times[2]) * 1e-3
        else:
            collection_time, learn_time = times
        v = host.tolist()
        ring, cnt, mean_std, stats = v[:200], int(v[200]) & 0xFFFFFFFF, v[201], v[202:210]
        vloss, sloss, acc, faults = v[210], v[211], v[212:212 + nacc], int(v[212 + nacc])
        lr = stats[0]
        steps = self.num_steps_per_env * self.env.num_envs * self.world_size
        self.tot_timesteps += steps
        self.tot_time += collection_time + learn_time
        fps = int(steps / (collection_time + learn_time))
        self.last_fps = fps
        # rsl_rl bookkeeping, kept on the device by k_process_step / k_finalize and fetched once per iteration:
        # rewbuffer / lenbuffer = the last 100 finished episodes; ep_infos = infos["episode"] of every step, averaged
        # (the device counter is a free-running uint32: slot = count % 100; once 100 episodes have finished the ring stays full)
        self._ring_full = getattr(self, "_ring_full", False) or cnt >= 100
        k = 100 if self._ring_full else cnt
        self.rewbuffer, self.lenbuffer = deque(ring[:k], maxlen=100), deque(ring[100:100 + k], maxlen=100)
        nsteps = max(float(acc[-1]), 1.0)
        ep = {}
        rows = self.env.setup.term_row
        for name in self.env.extras.get("episode", {}):
            if name == "terrain_level":
                ep[name] = float(acc[_NUM_TERMS]) / nsteps
            elif name.startswith("rew_") and name[4:] in rows:
                ep[name] = float(acc[rows[name[4:]]]) / nsteps
        if self.rank != 0:
            return
        lines = [f" Learning iteration {it}/{tot_iter} ".center(width, " "), "",
                 f"{'Computation:':>{pad}} {fps:.0f} steps/s (collection: {collection_time:.3f}s, learning {learn_time:.3f}s)",
                 f"{'Value function loss:':>{pad}} {vloss:.4f}", f"{'Surrogate loss:':>{pad}} {sloss:.4f}",
                 f"{'Mean action noise std:':>{pad}} {mean_std:.2f}", f"{'Learning rate:':>{pad}} {lr:.2e}"]
        if self.rewbuffer:
            lines += [f"{'Mean reward:':>{pad}} {statistics.mean(self.rewbuffer):.2f}",
                      f"{'Mean episode length:':>{pad}} {statistics.mean(self.lenbuffer):.2f}"]
        lines += [f"{'Mean episode ' + k + ':':>{pad}} {v:.4f}" for k, v in ep.items()]
        if faults:
            lines += [f"{'Physics fault resets (total):':>{pad}} {faults}"]
        lines += ["-" * width, f"{'Total timesteps:':>{pad}} {self.tot_timesteps}",
                  f"{'Iteration time:':>{pad}} {collection_time + learn_time:.2f}s", f"{'Total time:':>{pad}} {self.tot_time:.2f}s"]
        print("#" * width + "\n" + "\n".join(lines) + "\n", flush=True)
        if self.wandb_callback is not None:
            locs = {"mean_value_loss": vloss, "mean_surrogate_loss": sloss, "rewbuffer": self.rewbuffer,
                    "lenbuffer": self.lenbuffer, "ep_infos": [ep] if ep else [], "it": it,
                    "collection_time": collection_time, "learn_time": learn_time, "tot_iter": tot_iter}
            self.wandb_callback(locs, lr, mean_std, self.alg.actor_critic.state_dict(),
                                ppo.optimizer_state_dict(), self.device, 0)

    # ------------------------------------------------------------------ checkpoints
    def save(self, path, infos=None):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        # rsl_rl OnPolicyRunner.save layout; the optimiser entry is torch.optim.Adam's own state_dict
        torch.save({"model_state_dict": {k: v.cpu() for k, v in self.ppo.state_dict().items()},
                    "optimizer_state_dict": self.ppo.optimizer_state_dict(),
                    "iter": self.current_learning_iteration, "infos": infos}, path)

    def load(self, path, load_optimizer=True):
        d = torch.load(path, map_location="cpu", weights_only=True)
        self.ppo.load_state_dict(d["model_state_dict"])
        if load_optimizer and d.get("optimizer_state_dict"):
            self.ppo.load_optimizer_state_dict(d["optimizer_state_dict"])
        self.current_learning_iteration = d["iter"]
        return d.get("infos")

    def get_inference_policy(self, device=None):
        return self.ppo.act_inference
