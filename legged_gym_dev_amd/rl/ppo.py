"""Python face of the HIP PPO learner (lg_ppo_* in include/legged_hip.h).

Mirrors the roles of rsl_rl's ``ActorCritic`` / ``PPO`` / ``RolloutStorage`` (third-party, not in the
reference tree; semantics per SURVEY.md Appendix B).  All tensors are zero-copy torch views of
library-owned HBM; every compute call is a C-ABI call -- there is no torch fallback.
"""
import ctypes as C
import os

import torch

from legged_gym_dev_amd import capi
from legged_gym_dev_amd.lib import LeggedHipError, device_tensor, load

_STATS = ["lr", "kl", "value_loss_sum", "surrogate_loss_sum", "adam_t", "n_updates", "adv_mean", "adv_std"]


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# rsl_rl v1.0.2 get_activation (rsl_rl/modules/actor_critic.py, third-party -- recalled, not in the reference tree): its "crelu"
# entry returns nn.ReLU(), so the name listed at legged_robot_config.py:244 is plain ReLU with unchanged layer widths
_ACTIVATIONS = {"elu": 0, "selu": 1, "relu": 2, "lrelu": 3, "tanh": 4, "sigmoid": 5, "crelu": 2}


class HipPPO:
    def __init__(self, num_envs, num_obs, num_critic_obs, num_actions, policy_cfg, alg_cfg, num_steps,
                 device="cuda:0", seed=1, world_size=1, rank=0):
        self.lib = load()
        if not hasattr(self.lib, "lg_ppo_create"):
            raise LeggedHipError("liblegged_hip.so was built without the PPO kernels")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        ah, ch = list(policy_cfg["actor_hidden_dims"]), list(policy_cfg["critic_hidden_dims"])
        if len(ah) != len(ch) or len(ah) > capi.MAX_HIDDEN:
            raise ValueError("actor/critic need the same number (<=4) of hidden layers")
        self.activation = policy_cfg.get("activation", "elu")
        if self.activation not in _ACTIVATIONS:
            raise NotImplementedError(f"activation {self.activation!r}: HIP epilogues exist for {sorted(_ACTIVATIONS)}")
        c = capi.lg_ppo_cfg()
        c.num_envs, c.num_obs, c.num_actions = num_envs, num_obs, num_actions
        self.privileged = num_critic_obs is not None and num_critic_obs != num_obs
        c.num_critic_obs = num_critic_obs if self.privileged else 0
        c.num_hidden = len(ah)
        for i, (a, b) in enumerate(zip(ah, ch)):
            c.actor_hidden[i], c.critic_hidden[i] = a, b
        c.activation = _ACTIVATIONS[self.activation]
        c.num_steps = num_steps
        c.num_epochs, c.num_mini_batches = alg_cfg["num_learning_epochs"], alg_cfg["num_mini_batches"]
        c.adaptive_schedule = int(alg_cfg.get("schedule", "adaptive") == "adaptive" and alg_cfg.get("desired_kl") is not None)
        c.use_clipped_value_loss = int(alg_cfg.get("use_clipped_value_loss", True))
        c.world_size = world_size
        c.seed = (int(seed) * 1000003 + rank) & 0xFFFFFFFFFFFFFFFF
        c.init_noise_std = policy_cfg.get("init_noise_std", 1.0)
        c.value_loss_coef, c.clip_param = alg_cfg["value_loss_coef"], alg_cfg["clip_param"]
        c.entropy_coef, c.learning_rate = alg_cfg["entropy_coef"], alg_cfg["learning_rate"]
        c.gamma, c.lam = alg_cfg["gamma"], alg_cfg["lam"]
        c.desired_kl = alg_cfg.get("desired_kl") or 0.0
        c.max_grad_norm = alg_cfg["max_grad_norm"]
        self.cfg = c
        self.N, self.O, self.A, self.T = num_envs, num_obs, num_actions, num_steps
        self.OC = num_critic_obs if self.privileged else num_obs
        self.world_size = world_size
        self.ctx = C.c_void_p()
        rc = self.lib.lg_ppo_create(C.byref(c), C.byref(self.ctx))
        if rc != 0:
            raise LeggedHipError(f"lg_ppo_create failed ({rc}): {self.lib.lg_last_error().decode()}")
        b = capi.lg_ppo_buffers()
        self.lib.lg_ppo_get_buffers(self.ctx, C.byref(b))
        self.num_params, self.num_reduce = int(b.num_params), int(b.num_reduce)
        T, N, O, A, OC, P = self.T, self.N, self.O, self.A, self.OC, self.num_params
        shapes = {"params": (P + 2,), "grads": (P + 2,), "adam_m": (P + 2,), "adam_v": (P + 2,),
                  "obs": (T, N, O), "critic_obs": (T, N, OC), "actions": (T, N, A), "rewards": (T, N),
                  "values": (T, N), "returns": (T, N), "advantages": (T, N), "log_prob": (T, N), "mu": (T, N, A),
                  "sigma": (A,), "act_actions": (N, A), "act_values": (N,), "act_log_prob": (N,), "act_mu": (N, A),
                  "stats": (8,), "noise": (N, A), "adv_partial": (4,), "cur_reward_sum": (N,), "cur_episode_len": (N,),
                  "ep_stats": (4,), "ep_ring": (2, 100)}
        self.t = {}
        for name, shape in shapes.items():
            ptr = C.cast(getattr(b, name), C.c_void_p).value
            self.t[name] = device_tensor(ptr, shape, "f4", self, self.device)
        self.t["dones"] = device_tensor(C.cast(b.dones, C.c_void_p).value, (T, N), "u1", self, self.device)
        self.t["perm"] = device_tensor(C.cast(b.perm, C.c_void_p).value, (T * N,), "i4", self, self.device)
        self.t["ep_ring_count"] = device_tensor(C.cast(b.ep_ring_count, C.c_void_p).value, (1,), "i4", self, self.device)
        # named parameter views, ActorCritic.parameters() order
        offs = (C.c_int64 * 32)()
        shp = (C.c_int64 * 64)()
        n = self.lib.lg_ppo_param_layout(self.ctx, offs, shp, 32)
        names = ["std"]
        for net, nl in (("actor", len(ah) + 1), ("critic", len(ch) + 1)):
            for l in range(nl):
                names += [f"{net}.{2 * l}.weight", f"{net}.{2 * l}.bias"]
        assert n == len(names), (n, names)
        self.param_views, self.grad_views = {}, {}
        for k, name in enumerate(names):
            r, cdim = int(shp[2 * k]), int(shp[2 * k + 1])
            size = r * cdim if cdim else r
            view_shape = (r, cdim) if cdim else (r,)
            self.param_views[name] = self.t["params"][int(offs[k]):int(offs[k]) + size].view(view_shape)
            self.grad_views[name] = self.t["grads"][int(offs[k]):int(offs[k]) + size].view(view_shape)
        self._init_parameters(ah, ch)
        self.params_changed()
        self.use_current_stream()
        if os.environ.get("LG_DETERMINISTIC", "0") not in ("", "0"):
            self.set_deterministic(True)

    # nn.Linear default initialisation, drawn in the order rsl_rl's ActorCritic constructs its layers
    def _init_parameters(self, ah, ch):
        def mlp(i, hidden, o):
            dims = [i] + list(hidden) + [o]
            return [torch.nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]
        for net, layers in (("actor", mlp(self.O, ah, self.A)), ("critic", mlp(self.OC, ch, 1))):
            for l, lin in enumerate(layers):
                self.param_views[f"{net}.{2 * l}.weight"].copy_(lin.weight.detach())
                self.param_views[f"{net}.{2 * l}.bias"].copy_(lin.bias.detach())

    def use_current_stream(self):
        self.lib.lg_ppo_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))

    def _call(self, fn, *args):
        rc = getattr(self.lib, "lg_ppo_" + fn)(self.ctx, *args)
        if rc != 0:
            raise LeggedHipError(f"lg_ppo_{fn} failed ({rc}): {self.lib.lg_last_error().decode()}")

    # ------------------------------------------------------------------ rsl_rl PPO surface
    def act(self, obs, critic_obs=None):
        self._call("act", _vp(obs), _vp(critic_obs if self.privileged else None))
        return self.t["act_actions"]

    def process_env_step(self, rewards, dones, infos):
        to = infos.get("time_outs") if isinstance(infos, dict) else None
        self._call("process_env_step", _vp(rewards), _vp(dones), _vp(to))

    def compute_returns(self, last_critic_obs, all_reduce=None):
        self._call("compute_returns", _vp(last_critic_obs))
        if all_reduce is not None and self.world_size > 1:
            all_reduce(self.t["adv_partial"])
        self._call("normalize_advantages")

    def update(self, all_reduce=None):
        """5 epochs x 4 minibatches (cfg); returns (mean_value_loss, mean_surrogate_loss) lazily as tensors."""
        self._call("begin_update")
        reduce_view = self.t["grads"][: self.num_reduce]
        for epoch in range(self.cfg.num_epochs):
            for mb in range(self.cfg.num_mini_batches):
                self._call("minibatch_backward", epoch, mb)
                if all_reduce is not None and self.world_size > 1:
                    all_reduce(reduce_view)
                self._call("minibatch_step")
        self._call("end_update")
        st = self.t["stats"]
        return st[2] / st[5], st[3] / st[5]

    def act_inference(self, obs):
        out = torch.empty(obs.shape[0], self.A, device=self.device)
        self._call("act_inference", _vp(obs.contiguous()), _vp(out), C.c_int64(obs.shape[0]))
        return out

    def inject_noise(self, enable):
        self.lib.lg_ppo_inject_noise(self.ctx, int(enable))

    @property
    def learning_rate(self):
        return float(self.t["stats"][0])

    def stats(self):
        v = self.t["stats"].cpu().tolist()
        return dict(zip(_STATS, v))

    # ------------------------------------------------------------------ checkpoints (rsl_rl key names)
    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.param_views.items()}

    def load_state_dict(self, sd):
        for k, v in sd.items():
            self.param_views[k].copy_(v.to(self.device))
        self.params_changed()

    def comm_timing(self, enable):
        """lg_ppo_comm_timing: record the learner stream's wait for the gradient buckets (NativeComm) of every minibatch."""
        self.lib.lg_ppo_comm_timing.argtypes = [C.c_void_p, C.c_int]
        self.lib.lg_ppo_comm_timing(self.ctx, int(bool(enable)))

    def comm_wait_ms(self):
        """(total ms the learner's stream waited for reduced buckets, minibatches recorded) since the last call; synchronises."""
        ms, n = C.c_double(), C.c_int64()
        self.lib.lg_ppo_comm_wait_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        rc = self.lib.lg_ppo_comm_wait_ms(self.ctx, C.byref(ms), C.byref(n))
        if rc != 0:
            raise LeggedHipError(f"lg_ppo_comm_wait_ms failed ({rc}): {self.lib.lg_last_error().decode()}")
        return float(ms.value), int(n.value)

    def attach_env(self, core):
        """lg_ppo_attach_env: ``core`` = the env's HipEnvCore (or None to detach, which runs whatever is still pending)."""
        self._call("attach_env", core.ctx if core is not None else None)

    def set_deterministic(self, on=True):
        """lg_ppo_set_deterministic: order-independent (fixed-point) accumulation of every cross-workgroup sum of the update, so
        that two runs from one seed agree bit for bit (debugging aid: three extra small launches per optimiser step).  Also
        switched on by LG_DETERMINISTIC=1 in the environment."""
        self._call("set_deterministic", 1 if on else 0)

    def params_changed(self):
        """Call after writing parameters through ``param_views`` / ``t["params"]``: the rollout's weight images are re-derived
        by the next act() (also in the middle of a rollout)."""
        self.lib.lg_ppo_params_changed(self.ctx)

    def optimizer_state_dict(self):
        """torch.optim.Adam.state_dict() layout over ActorCritic.parameters() order (rl/checkpoint.py)."""
        from .checkpoint import adam_state_to_torch
        names = list(self.param_views.keys())          # flat-buffer order == parameters() order
        shapes = {k: tuple(v.shape) for k, v in self.param_views.items()}
        return adam_state_to_torch(names, shapes, self.t["adam_m"][: self.num_params], self.t["adam_v"][: self.num_params],
                                   float(self.t["stats"][4]), float(self.t["stats"][0]))

    def load_optimizer_state_dict(self, sd):
        from .checkpoint import adam_state_from_torch
        names = list(self.param_views.keys())
        shapes = {k: tuple(v.shape) for k, v in self.param_views.items()}
        m, v, step, lr = adam_state_from_torch(sd, names, shapes)
        if m.numel() != self.num_params:
            raise ValueError(f"optimizer state holds {m.numel()} values, the policy has {self.num_params}")
        self.t["adam_m"][: self.num_params].copy_(m.to(self.device))
        self.t["adam_v"][: self.num_params].copy_(v.to(self.device))
        self.t["stats"][4] = step
        self.t["stats"][0] = lr

    def close(self):
        if getattr(self, "ctx", None):
            self.t, self.param_views, self.grad_views = {}, {}, {}
            self.lib.lg_ppo_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
