"""Play-time logger (reference legged_gym/utils/logger.py:36-137): per-step state logs, episode reward
statistics; plot_states() writes files instead of opening a window (there is no viewer in this stack)."""
import os
from collections import defaultdict

import numpy as np


class Logger:
    def __init__(self, dt):
        self.state_log = defaultdict(list)
        self.rew_log = defaultdict(list)
        self.dt = dt
        self.num_episodes = 0

    def log_state(self, key, value):
        self.state_log[key].append(value)

    def log_states(self, dict):
        for key, value in dict.items():
            self.log_state(key, value)

    def log_rewards(self, dict, num_episodes):
        for key, value in dict.items():
            if "rew" in key:
                self.rew_log[key].append(float(value) * num_episodes)
        self.num_episodes += num_episodes

    def reset(self):
        self.state_log.clear()
        self.rew_log.clear()

    def plot_states(self, path="play_states.npz"):
        """Headless stand-in for the reference's multiprocess matplotlib window (logger.py:63-126): the logs
        go to `path` (.npz); with matplotlib importable the same nine panels are drawn into a .png next to it."""
        log = {k: np.asarray(v) for k, v in self.state_log.items()}
        np.savez(path, dt=self.dt, **log)
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except ImportError:
            print(f"state logs written to {path}")
            return path
        n = max((len(v) for v in log.values()), default=0)
        t = np.arange(n) * self.dt
        panels = [("dof position [rad]", ("dof_pos", "dof_pos_target")), ("dof velocity [rad/s]", ("dof_vel", "dof_vel_target")),
                  ("base velocity x [m/s]", ("base_vel_x", "command_x")), ("base velocity y [m/s]", ("base_vel_y", "command_y")),
                  ("base velocity yaw [rad/s]", ("base_vel_yaw", "command_yaw")), ("base velocity z [m/s]", ("base_vel_z",)),
                  ("vertical contact forces [N]", ("contact_forces_z",)), ("torque/velocity", ()), ("torque [Nm]", ("dof_torque",))]
        fig, axs = plt.subplots(3, 3, figsize=(14, 9))
        for ax, (title, keys) in zip(axs.flatten(), panels):
            for k in keys:
                if k in log and len(log[k]):
                    ax.plot(t[: len(log[k])], log[k], label=k)
            if title == "torque/velocity" and "dof_vel" in log and "dof_torque" in log:
                ax.plot(log["dof_vel"], log["dof_torque"], "x", label="measured")
            ax.set_title(title)
            if ax.get_legend_handles_labels()[0]:
                ax.legend(fontsize=7)
        png = os.path.splitext(path)[0] + ".png"
        fig.tight_layout()
        fig.savefig(png, dpi=80)
        plt.close(fig)
        print(f"state logs written to {path} and {png}")
        return path

    def print_rewards(self):
        print("Average rewards per second:")
        for key, values in self.rew_log.items():
            mean = np.sum(np.array(values)) / max(self.num_episodes, 1)
            print(f" - {key}: {mean}")
        print(f"Total number of episodes: {self.num_episodes}")
