"""Host-side setup: LeggedRobotCfg (+ collapsed model) -> the POD ``lg_cfg`` the kernels run on.

Restates the setup-time logic of the reference env (it runs once, in Python, there too):
``_parse_cfg`` (legged_robot.py:819-837), the body-index sets and ``base_init_state`` of
``_create_envs`` (:718-724,757-770), ``_process_dof_props`` soft limits (:313-327),
``_get_noise_scale_vec`` (:507-530), default joint angles / PD gains by substring match
(:588-602), ``_init_height_points`` (:861-875) and ``_prepare_reward_function`` (:605-629).
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os

import numpy as np

from legged_gym_dev_amd import capi
from legged_gym_dev_amd.envs.base import cfg_contract
from legged_gym_dev_amd.utils.helpers import class_to_dict

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "assets")
_CONTROL_TYPES = {"P": 0, "V": 1, "T": 2}


def _fill(arr, values):
    for i, v in enumerate(values):
        arr[i] = v


def sim_dt_float(dt) -> float:
    """``gymapi.SimParams.dt`` is a C float: Python sees float32(dt) widened to double, so
    4 * dt = 0.0199999996 and ceil(20 / dt) = 1001."""
    return float(np.float32(dt))


def load_actuator_weights(path_hint: str = "") -> np.ndarray:
    """Actuator-net weights in lg_cfg.lstm_w order.  ``cfg.control.actuator_net_file`` names a
    TorchScript archive in the reference tree; this build never executes such a file -- the
    numbers were extracted once by tools/compile_assets.py into assets/<stem>.json."""
    stem = os.path.splitext(os.path.basename(path_hint))[0] if path_hint else "anydrive_v3_lstm"
    p = os.path.join(_ASSETS, f"{stem}.json")
    if not os.path.isfile(p):
        raise FileNotFoundError(f"no compiled actuator-net weights for {path_hint!r} (expected {p})")
    with open(p) as f:
        w = json.load(f)
    order = ["in_scale", "out_scale", "weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0",
             "weight_ih_l1", "weight_hh_l1", "bias_ih_l1", "bias_hh_l1", "linear_weight", "linear_bias"]
    flat = np.concatenate([np.asarray(w[k], dtype=np.float32).reshape(-1) for k in order])
    assert flat.size == capi.LSTM_NW, flat.size
    return flat


class CurriculumClock:
    """The reference's curriculum stage machine (legged_robot.py:360-363 / legged_robot_trajectory.py:414-417): at the end of
    the step callback, ``if use_curriculum and state < len(curriculum_steps) and common_step_counter % curriculum_steps[state]
    == 0: state += 1``.  Host logic there and here; the constants of a stage travel through lg_set_curriculum_stage."""

    def __init__(self, setup):
        self.enabled, self.steps, self.state = bool(setup.use_curriculum), list(setup.curriculum_steps), 0

    def tick(self, common_step_counter):
        """True when the step that carries this (already incremented) counter moves to the next stage."""
        if self.enabled and self.state < len(self.steps) and common_step_counter % self.steps[self.state] == 0:
            self.state += 1
            return True
        return False


class EnvSetup:
    """Everything derived from (cfg, model) before any device work."""

    def __init__(self, cfg, cm: dict, sim_dt: float, terrain=None, env_offset=0, total_envs=None, seed=1, extra_terms=None):
        self.cfg, self.cm = cfg, cm
        cfg_contract.enforce(cfg)                           # no silently ignored field: refusals and one-time warnings (cfg_contract.py)
        self.extra_terms = dict(extra_terms or {})          # name -> reward_terms._Term (declared by the env class)
        A, B = cm["num_dofs"], cm["num_bodies"]
        self.num_envs = N = cfg.env.num_envs
        self.num_dof = self.num_actions = A
        if cfg.env.num_actions != A:
            raise ValueError(f"cfg.env.num_actions={cfg.env.num_actions} but the asset has {A} DOFs")
        self.num_bodies = B
        self.body_names, self.dof_names = list(cm["body_names"]), list(cm["dof_names"])

        # ---- _parse_cfg
        self.sim_dt = sim_dt
        self.dt = cfg.control.decimation * sim_dt
        self.obs_scales = cfg.normalization.obs_scales
        self.reward_scales = class_to_dict(cfg.rewards.scales)
        self.traj = self._parse_trajectory(cfg)             # None for the velocity-command env
        if hasattr(cfg, "commands"):
            self.command_ranges = class_to_dict(cfg.commands.ranges)
        else:                                               # the trajectory env has no commands section
            self.command_ranges = {k: [0.0, 0.0] for k in ("lin_vel_x", "lin_vel_y", "ang_vel_yaw", "heading")}
        self.nominal_command_ranges = {k: list(v) for k, v in self.command_ranges.items()}
        self.push_time = math.ceil(cfg.domain_rand.push_interval_s / self.dt)
        self.max_push_vel = getattr(cfg.domain_rand, "max_push_vel", cfg.domain_rand.max_push_vel_xy)
        if self.traj is not None:                            # a 6-vector there, and unused: pushes draw from max_push_vel_xy (:483-486)
            self.max_push_vel = cfg.domain_rand.max_push_vel_xy
        if cfg.terrain.mesh_type not in ("heightfield", "trimesh"):
            cfg.terrain.curriculum = False
        self.max_episode_length_s = cfg.env.episode_length_s
        self.max_episode_length = math.ceil(self.max_episode_length_s / self.dt)
        # staged curriculum (legged_robot.py:360-363,488-505,828-829; legged_robot_trajectory.py:78-79,414-417,519-553)
        cur = getattr(cfg, "curriculum", None)
        self.use_curriculum = bool(getattr(cur, "use_curriculum", False))
        self.curriculum_steps = [int(v) for v in getattr(cur, "curriculum_steps", [])] if self.use_curriculum else []
        self.nominal_push_time = float(math.ceil(cfg.domain_rand.push_interval_s / self.dt))
        # domain_rand.dof_properties.* (stiffness / damping of the hopper's joints) are read by the hopper env only
        # (legged_gym/envs/hopper); LeggedRobot._process_dof_props (legged_robot.py:301-328) never looks at them

        # ---- body index sets (substring match on names)
        feet = [i for i, s in enumerate(self.body_names) if cfg.asset.foot_name in s]
        pen = []
        for name in cfg.asset.penalize_contacts_on:
            pen.extend(i for i, s in enumerate(self.body_names) if name in s)
        term = []
        for name in cfg.asset.terminate_after_contacts_on:
            term.extend(i for i, s in enumerate(self.body_names) if name in s)
        self.feet_indices, self.penalised_contact_indices, self.termination_contact_indices = feet, pen, term
        if len(feet) > capi.MAX_FEET or len(pen) > capi.MAX_PEN or len(term) > capi.MAX_TERM:
            raise ValueError("too many feet / penalised / termination bodies for LG_MAX_*")

        # ---- dof properties + soft limits
        lo, hi = cm["q_lower"].astype(np.float32), cm["q_upper"].astype(np.float32)
        mid = (lo + hi) / np.float32(2)
        rng = hi - lo
        soft = cfg.rewards.soft_dof_pos_limit
        self.dof_pos_limits = np.stack([mid - np.float32(0.5) * rng * np.float32(soft),
                                        mid + np.float32(0.5) * rng * np.float32(soft)], 1).astype(np.float32)
        self.dof_vel_limits = cm["vel_limit"].astype(np.float32)
        self.torque_limits = cm["effort"].astype(np.float32)

        # ---- default joint angles and PD gains
        self.default_dof_pos = np.zeros(A, np.float32)
        self.p_gains = np.zeros(A, np.float32)
        self.d_gains = np.zeros(A, np.float32)
        for i, name in enumerate(self.dof_names):
            self.default_dof_pos[i] = cfg.init_state.default_joint_angles[name]
            found = False
            for key in cfg.control.stiffness.keys():
                if key in name:
                    self.p_gains[i] = cfg.control.stiffness[key]
                    self.d_gains[i] = cfg.control.damping[key]
                    found = True
            if not found and cfg.control.control_type in ("P", "V"):
                print(f"PD gain of joint {name} were not defined, setting them to zero")
        if cfg.control.control_type not in _CONTROL_TYPES:
            raise NameError(f"Unknown controller type: {cfg.control.control_type}")

        # ---- height scan points: meshgrid(x, y) 'ij' -> x-major
        self.measure_heights = bool(cfg.terrain.measure_heights)
        if self.measure_heights:
            xs = np.asarray(cfg.terrain.measured_points_x, np.float32)
            ys = np.asarray(cfg.terrain.measured_points_y, np.float32)
            gx, gy = np.meshgrid(xs, ys, indexing="ij")
            self.height_points = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
        else:
            self.height_points = np.zeros((0, 2), np.float32)
        self.num_height_points = H = self.height_points.shape[0]
        O = cfg.env.num_observations
        ncmd = 3 if self.traj is None else self.traj["N"] * 2        # commands[:3] | the N trajectory points (xy)
        if O != 9 + ncmd + 3 * A + H:
            raise ValueError(f"num_observations={O} does not match 9 + {ncmd} + 3*{A} + {H} height points")

        # ---- noise scale vector (legged_robot.py:507-530 / legged_robot_trajectory.py:563-587)
        ns, lvl, osc = cfg.noise.noise_scales, cfg.noise.noise_level, self.obs_scales
        nv = np.zeros(O, np.float32)
        nv[0:3] = ns.lin_vel * lvl * osc.lin_vel
        nv[3:6] = ns.ang_vel * lvl * osc.ang_vel
        nv[6:9] = ns.gravity * lvl
        j0 = 9 + ncmd
        nv[j0:j0 + A] = ns.dof_pos * lvl * osc.dof_pos
        nv[j0 + A:j0 + 2 * A] = ns.dof_vel * lvl * osc.dof_vel
        if self.measure_heights:
            nv[j0 + 3 * A:] = ns.height_measurements * lvl * osc.height_measurements
        self.noise_scale_vec = nv

        # ---- reward scales: drop zeros, x dt, alphabetical order (class_to_dict walks dir())
        for k in list(self.reward_scales.keys()):
            if self.reward_scales[k] == 0:
                self.reward_scales.pop(k)
            else:
                self.reward_scales[k] *= self.dt
        unknown = [k for k in self.reward_scales if k not in capi.REWARD_NAMES and k not in self.extra_terms]
        if unknown:
            raise AttributeError(f"no reward term named {unknown}: builtin terms are {capi.REWARD_NAMES}; a subclass adds terms by "
                                 "declaring them in extra_reward_terms() (envs/base/reward_terms.py), the counterpart of the "
                                 "reference's _reward_<name> methods")
        # active extra terms keep the index they are declared with; the sum runs over ALL active names alphabetically
        # (class_to_dict walks dir(), helpers.py:111-126), termination excepted: it is applied last (legged_robot.py:199-206)
        self.xterm_names = [k for k in self.extra_terms if k in self.reward_scales]
        if len(self.xterm_names) > capi.MAX_XTERMS:
            raise ValueError(f"at most {capi.MAX_XTERMS} extra reward terms")
        if self.traj is not None and "stand_still" in self.reward_scales:
            raise AttributeError("stand_still reads self.commands, which the trajectory env does not have "
                                 "(legged_robot_trajectory.py:1088-1090 would raise as well)")
        self.term_row = {k: (capi.REWARD_NAMES.index(k) if k in capi.REWARD_NAMES else capi.NUM_REWARDS + self.xterm_names.index(k))
                         for k in self.reward_scales}
        self.term_order = [self.term_row[k] for k in sorted(self.reward_scales) if k != "termination"]

        self.base_init_state = np.asarray(
            list(cfg.init_state.pos) + list(cfg.init_state.rot) + list(cfg.init_state.lin_vel)
            + list(cfg.init_state.ang_vel), np.float32)
        self.terrain = terrain
        self.custom_origins = cfg.terrain.mesh_type in ("heightfield", "trimesh")
        self.use_actuator_net = bool(getattr(cfg.control, "use_actuator_network", False))
        self.env_offset = env_offset
        self.total_envs = total_envs if total_envs is not None else N
        self.seed = seed

    # ------------------------------------------------------------------------------
    @staticmethod
    def _parse_trajectory(cfg):
        """cfg.rom / cfg.trajectory_generator of the trajectory-tracking env (legged_robot_trajectory.py:87-122), or None."""
        if not hasattr(cfg, "trajectory_generator"):
            return None
        rom, tg, dr = cfg.rom, cfg.trajectory_generator, cfg.domain_rand
        if rom.cls != "SingleInt2D":
            raise NotImplementedError(f"reduced-order model {rom.cls!r}: the HIP step implements SingleInt2D (the fork's ANYmal "
                                      "trajectory tasks); DoubleInt2D / unicycle models are not implemented")
        if tg.cls != "TrajectoryGenerator" or tg.t_samp_cls != "UniformSampleHoldDT":
            raise NotImplementedError(f"trajectory generator {tg.cls!r} / time sampler {tg.t_samp_cls!r} are not implemented")
        if tg.weight_samp_cls != "UniformWeightSampler":
            raise NotImplementedError(f"weight sampler {tg.weight_samp_cls!r}: only UniformWeightSampler is implemented")
        dN = int(getattr(tg, "dN", 1))
        if int(tg.N) * dN + 1 > capi.TRAJ_MAX_PTS:
            raise ValueError("trajectory window longer than LG_TRAJ_MAX_PTS")
        if dN != 1:
            raise NotImplementedError("trajectory_generator.dN != 1 is not implemented")
        return {"N": int(tg.N), "dN": dN, "rom_dt": float(rom.dt), "t_low": float(tg.t_low), "t_high": float(tg.t_high),
                "freq_low": float(tg.freq_low), "freq_high": float(tg.freq_high), "prob_stationary": float(tg.prob_stationary),
                "v_min": [float(v) for v in rom.v_min], "v_max": [float(v) for v in rom.v_max],
                "obs_scale": [float(v) for v in cfg.normalization.obs_scales.trajectory],
                "randomize_rom_distance": bool(getattr(dr, "randomize_rom_distance", False)),
                "max_rom_dist": [float(v) for v in (getattr(dr, "max_rom_dist", None) or [0.0, 0.0])],
                "zero_rom_dist_llh": float(getattr(dr, "zero_rom_distance_likelihood", 0.0)),
                "max_push_vel_xy": float(dr.max_push_vel_xy), "push_t": [float(v) for v in dr.time_between_pushes]}

    # ------------------------------------------------------------------------------ staged curriculum
    def _nominal_push_vel(self):
        """``_push_robots`` draws from +-self.max_push_vel as a scalar (legged_robot.py:459), ``update_command_curriculum`` iterates
        over it as a list (:503): with the curriculum on, the reference needs the list form to get through _parse_cfg and then
        raises at its first push.  Both forms are accepted here; the push magnitude is the (first) value."""
        v = self.max_push_vel
        return float(v[0]) if isinstance(v, (list, tuple)) else float(v)

    def stage_values(self, ind):
        """What ``update_command_curriculum`` leaves in the env at curriculum state ``ind`` (None: no curriculum, the nominal
        values), as a dict of Python-side values; ``stage_struct`` packs it for lg_set_curriculum_stage.
        Base env (legged_robot.py:488-505): command ranges x commands[ind] (heading excepted), push magnitude x
        push.magnitude[ind], push period x push.time[ind].  Trajectory env (legged_robot_trajectory.py:519-553): reward scales
        x rewards.<name>[ind], tracking sigma x sigma.tracking_rom[ind], ROM input bounds x rom.v[ind], hold-time sampler bounds
        x trajectory_generator.t_low / t_high[ind], start-offset range x max_rom_distance[ind].  Reproduced as the reference
        has them: that update also scales max_push_vel / push_time, which the trajectory env's per-env push timers never read
        (:150-160,486-492), writes ``zero_rom_distance_likelihood`` while reset_traj reads ``zero_rom_dist_llh`` (:73,251), and
        ignores the freq_low / freq_high / weight_sampler rows of the launch file."""
        cfg, cur = self.cfg, getattr(self.cfg, "curriculum", None)
        out = {"command_ranges": {k: list(v) for k, v in self.nominal_command_ranges.items()},
               "max_push_vel": self._nominal_push_vel(), "push_time": self.nominal_push_time,
               "reward_scales": dict(self.reward_scales), "tracking_sigma": float(getattr(cfg.rewards, "tracking_sigma", 0.25))}
        tj = self.traj
        if tj is not None:
            out.update(v_min=list(tj["v_min"]), v_max=list(tj["v_max"]), t_low=tj["t_low"], t_high=tj["t_high"],
                       max_rom_dist=list(tj["max_rom_dist"]))
        if ind is None:
            return out
        out["max_push_vel"] = self._nominal_push_vel() * cur.push.magnitude[ind]
        out["push_time"] = self.nominal_push_time * cur.push.time[ind]
        if tj is None:
            out["command_ranges"] = {k: [v * cur.commands[ind] if k != "heading" else v for v in val]
                                     for k, val in self.nominal_command_ranges.items()}
            return out
        # torch.tensor(nominal) * multiplier (:530): a float32 product, unlike the ROM bounds below (double products, then cast)
        out["max_rom_dist"] = [float(np.float32(v) * np.float32(cur.max_rom_distance[ind])) for v in tj["max_rom_dist"]]
        out["v_max"] = [v * cur.rom.v[ind] for v in tj["v_max"]]
        out["v_min"] = [v * cur.rom.v[ind] for v in tj["v_min"]]
        out["t_low"] = tj["t_low"] * cur.trajectory_generator.t_low[ind]
        out["t_high"] = tj["t_high"] * cur.trajectory_generator.t_high[ind]
        out["tracking_sigma"] = out["tracking_sigma"] * cur.sigma.tracking_rom[ind]
        for key in list(self.reward_scales.keys()):        # an active term without a row in curriculum.rewards raises, as there
            out["reward_scales"][key] = getattr(cfg.rewards.scales, key) * getattr(cur.rewards, key)[ind] * self.dt
        return out

    def stage_struct(self, ind):
        v = self.stage_values(ind)
        s = capi.lg_stage()
        for k, name in enumerate(("lin_vel_x", "lin_vel_y", "ang_vel_yaw", "heading")):
            s.cmd_lo[k], s.cmd_hi[k] = float(v["command_ranges"][name][0]), float(v["command_ranges"][name][1])
        s.max_push_vel, s.push_time = float(v["max_push_vel"]), float(v["push_time"])
        for k, name in enumerate(capi.REWARD_NAMES):
            s.rew_scale[k] = float(v["reward_scales"].get(name, 0.0))
        for k, name in enumerate(self.xterm_names):
            s.xterm_scale[k] = float(v["reward_scales"][name])
            st = self.extra_terms[name].to_struct(v["reward_scales"][name])
            s.xterm_p0[k] = st.p[0]
            if name == "tracking_rom":                     # its sigma is env.tracking_sigma (legged_robot_trajectory.py:1069)
                s.xterm_p0[k] = float(v["tracking_sigma"])
        if self.traj is not None:
            for k in range(2):
                s.traj_v_min[k], s.traj_v_max[k], s.traj_max_rom_dist[k] = float(v["v_min"][k]), float(v["v_max"][k]), float(v["max_rom_dist"][k])
            s.traj_t_low, s.traj_t_high = float(v["t_low"]), float(v["t_high"])
        return s

    def to_structs(self):
        """(lg_cfg, lg_model, keepalive list).  Pointers in lg_cfg reference numpy arrays that
        must stay alive until lg_create returns; they are in the keepalive list."""
        cfg, cm = self.cfg, self.cm
        A = self.num_dof
        m = capi.lg_model()
        m.num_bodies, m.num_dofs = cm["num_bodies"], A
        m.num_legs, m.joints_per_leg, m.num_spheres = cm["num_legs"], cm["joints_per_leg"], cm["num_spheres"]
        for l in range(A + 1):
            m.mass[l] = float(cm["mass"][l])
            _fill(m.com[l], cm["com"][l].tolist())
            _fill(m.inertia[l], cm["inertia"][l].reshape(-1).tolist())
        for d in range(A):
            _fill(m.R_pj[d], cm["R_pj"][d].reshape(-1).tolist())
            _fill(m.p_pj[d], cm["p_pj"][d].tolist())
            _fill(m.axis[d], cm["axis"][d].tolist())
        for name in ("q_lower", "q_upper", "effort", "vel_limit", "joint_damping"):
            _fill(getattr(m, name), cm[name].tolist())
        _fill(m.body_dyn, cm["body_dyn"].tolist())
        _fill(m.sph_link, cm["sph_link"].tolist())
        _fill(m.sph_body, cm["sph_body"].tolist())
        for k in range(cm["num_spheres"]):
            _fill(m.sph_center[k], cm["sph_center"][k].tolist())
        _fill(m.sph_radius, cm["sph_radius"].tolist())

        c = capi.lg_cfg()
        c.num_envs, c.num_obs, c.num_actions, c.num_bodies = self.num_envs, cfg.env.num_observations, A, self.num_bodies
        c.num_feet, c.num_pen, c.num_term = (len(self.feet_indices), len(self.penalised_contact_indices),
                                              len(self.termination_contact_indices))
        c.num_height_points = self.num_height_points
        _fill(c.feet_idx, self.feet_indices)
        _fill(c.pen_idx, self.penalised_contact_indices)
        _fill(c.term_idx, self.termination_contact_indices)
        c.decimation = cfg.control.decimation
        c.control_type = _CONTROL_TYPES[cfg.control.control_type]
        c.use_actuator_net = int(self.use_actuator_net)
        c.heading_command = int(cfg.commands.heading_command) if hasattr(cfg, "commands") else 0
        c.max_episode_length = int(self.max_episode_length)
        c.resample_steps = int(cfg.commands.resampling_time / self.dt) if hasattr(cfg, "commands") else 1 << 30
        c.push_interval = int(self.push_time)
        c.push_robots = int(cfg.domain_rand.push_robots)
        c.add_noise = int(cfg.noise.add_noise)
        c.measure_heights = int(self.measure_heights)
        c.only_positive_rewards = int(cfg.rewards.only_positive_rewards)
        c.send_timeouts = int(cfg.env.send_timeouts)
        t = self.terrain
        keep = []
        if self.custom_origins:
            if t is None:
                raise ValueError("heightfield/trimesh terrain requested but no Terrain object given")
            c.terrain_type = 1
            c.hf_rows, c.hf_cols = int(t.tot_rows), int(t.tot_cols)
            c.terrain_num_cols = int(cfg.terrain.num_cols)
            c.max_terrain_level = int(cfg.terrain.num_rows)
            c.terrain_env_length = float(t.env_length)
            to = np.ascontiguousarray(t.env_origins, dtype=np.float32)
            keep.append(to)
            c.terrain_origins = to.ctypes.data_as(capi.PF)
        c.curriculum = int(bool(cfg.terrain.curriculum))
        c.custom_origins = int(self.custom_origins)
        c.phys_substeps = int(getattr(cfg.sim, "substeps", 1))
        c.env_offset, c.total_envs = int(self.env_offset), int(self.total_envs)
        physx = cfg.sim.physx
        c.solver_iterations = int(getattr(physx, "num_position_iterations", 4))
        c.seed = int(self.seed) & 0xFFFFFFFFFFFFFFFF
        c.sim_dt, c.dt = self.sim_dt, self.dt
        c.action_scale = cfg.control.action_scale
        c.clip_actions = cfg.normalization.clip_actions
        c.clip_obs = cfg.normalization.clip_observations
        c.max_push_vel = self._nominal_push_vel()
        c.episode_length_s = float(self.max_episode_length_s)
        r = self.command_ranges
        for k, name in enumerate(("lin_vel_x", "lin_vel_y", "ang_vel_yaw", "heading")):
            c.cmd_lo[k], c.cmd_hi[k] = float(r[name][0]), float(r[name][1])
        osc = self.obs_scales
        c.obs_scale_lin_vel, c.obs_scale_ang_vel = osc.lin_vel, osc.ang_vel
        c.obs_scale_dof_pos, c.obs_scale_dof_vel, c.obs_scale_height = osc.dof_pos, osc.dof_vel, osc.height_measurements
        rw = cfg.rewards
        c.tracking_sigma = float(getattr(rw, "tracking_sigma", 0.25))
        c.soft_dof_vel_limit, c.soft_torque_limit = rw.soft_dof_vel_limit, rw.soft_torque_limit
        c.base_height_target, c.max_contact_force = rw.base_height_target, rw.max_contact_force
        c.hf_hscale, c.hf_vscale, c.border_size = cfg.terrain.horizontal_scale, cfg.terrain.vertical_scale, cfg.terrain.border_size
        for k, name in enumerate(capi.REWARD_NAMES):
            c.rew_scale[k] = float(self.reward_scales.get(name, 0.0))
        c.num_xterms = len(self.xterm_names)
        for k, name in enumerate(self.xterm_names):
            c.xterms[k] = self.extra_terms[name].to_struct(self.reward_scales[name])
        c.num_terms = len(self.term_order)
        _fill(c.term_order, self.term_order)
        if self.traj is not None:
            tj, t = self.traj, c.traj
            t.enabled, t.N, t.dN, t.randomize_rom_distance = 1, tj["N"], tj["dN"], int(tj["randomize_rom_distance"])
            for k in ("rom_dt", "t_low", "t_high", "freq_low", "freq_high", "prob_stationary", "zero_rom_dist_llh", "max_push_vel_xy"):
                setattr(t, k, tj[k])
            for k in ("v_min", "v_max", "obs_scale", "max_rom_dist"):
                _fill(getattr(t, k), tj[k])
            t.push_t_lo, t.push_t_hi = tj["push_t"]
            c.feet_air_time_ungated = 1
        _fill(c.base_init_state, self.base_init_state.tolist())
        _fill(c.default_dof_pos, self.default_dof_pos.tolist())
        _fill(c.p_gains, self.p_gains.tolist())
        _fill(c.d_gains, self.d_gains.tolist())
        for d in range(A):
            c.dof_pos_limits[d][0], c.dof_pos_limits[d][1] = float(self.dof_pos_limits[d, 0]), float(self.dof_pos_limits[d, 1])
        _fill(c.dof_vel_limits, self.dof_vel_limits.tolist())
        _fill(c.torque_limits, self.torque_limits.tolist())
        # asset options the reference hands to the simulator (legged_robot.py:692-705)
        _fill(c.gravity, [0.0, 0.0, 0.0] if cfg.asset.disable_gravity else [float(g) for g in cfg.sim.gravity])
        c.max_linear_velocity, c.max_angular_velocity = float(cfg.asset.max_linear_velocity), float(cfg.asset.max_angular_velocity)
        c.armature = float(cfg.asset.armature)
        c.rest_offset = float(cfg.asset.thickness) + float(getattr(physx, "rest_offset", 0.0))
        c.ground_friction = float(cfg.terrain.static_friction)
        c.contact_offset = float(getattr(physx, "contact_offset", 0.01))
        c.max_depenetration_velocity = float(getattr(physx, "max_depenetration_velocity", 1.0))
        c.contact_erp = 0.2
        c.bounce_threshold = float(getattr(physx, "bounce_threshold_velocity", 0.5))
        c.ground_restitution = float(getattr(cfg.terrain, "restitution", 0.0))
        rsp = getattr(cfg.domain_rand, "rigid_shape_properties", None)
        c.material_rand = int(bool(cfg.domain_rand.randomize_friction) and rsp is not None and any(
            getattr(rsp, "randomize_" + k, False) for k in ("restitution", "compliance", "thickness")))
        if self.use_actuator_net:
            w = load_actuator_weights(getattr(cfg.control, "actuator_net_file", ""))
            C.memmove(c.lstm_w, w.ctypes.data, w.nbytes)
        nv = np.ascontiguousarray(self.noise_scale_vec, np.float32)
        hp = np.ascontiguousarray(self.height_points, np.float32)
        keep += [nv, hp]
        c.noise_vec = nv.ctypes.data_as(capi.PF)
        c.height_points = hp.ctypes.data_as(capi.PF) if hp.size else None
        return c, m, keep
