"""VecEnv base (reference: legged_gym/envs/base/base_task.py:40-148): device selection, the
attribute contract rsl_rl relies on, ``reset()`` = reset_idx(all) + zero-action step, twice.
The viewer / keyboard part of the reference is out of scope (headless only)."""
import torch


class BaseTask:
    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.sim_device = sim_device
        dev_type = str(sim_device).split(":")[0]
        if dev_type not in ("cuda", "hip"):
            raise RuntimeError(
                f"sim_device={sim_device!r}: this build runs the env on an MI355X only (no CPU pipeline). "
                "Use sim_device='cuda:<id>'.")
        self.device = str(sim_device).replace("hip", "cuda")
        self.headless = True                     # no viewer in this build
        self.num_envs = cfg.env.num_envs
        self.num_obs = cfg.env.num_observations
        self.num_privileged_obs = cfg.env.num_privileged_obs
        self.num_actions = cfg.env.num_actions
        if self.num_privileged_obs is not None:
            raise NotImplementedError("privileged observations are not produced by the reference LeggedRobot either")
        self.privileged_obs_buf = None
        self.extras = {}
        self.viewer = None
        self.enable_viewer_sync = True

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def reset_idx(self, env_ids):
        raise NotImplementedError

    def reset(self):
        ids = torch.arange(self.num_envs, device=self.device)
        zeros = torch.zeros(self.num_envs, self.num_actions, device=self.device, requires_grad=False)
        self.reset_idx(ids)
        obs, priv, _, _, _ = self.step(zeros)
        self.reset_idx(ids)                       # "For some reason, need an additional reset" (base_task.py:115)
        obs, priv, _, _, _ = self.step(zeros)
        return obs, priv

    def step(self, actions):
        raise NotImplementedError

    def render(self, sync_frame_time=True):
        return None
