"""Extra reward terms as data.

The reference binds any method ``_reward_<name>`` of the env class to a non-zero ``cfg.rewards.scales.<name>``
(legged_gym/envs/base/legged_robot.py:605-629); subclasses add terms that way (``Cassie._reward_no_fly`` cassie.py:43-46,
``LeggedRobotTrajectory._reward_tracking_rom`` / ``_reward_differential_error`` legged_robot_trajectory.py:1060-1110).
Here the whole post-step runs inside one HIP kernel, so a subclass declares an extra term as one of a few generic kinds over
named per-env signals (include/legged_hip.h: lg_xterm) instead of writing tensor code::

    class MyRobot(LeggedRobot):
        def extra_reward_terms(self):
            return {"tracking_xy": ExpNegWeightedSqErr("commands", "base_lin_vel", weights=[1.0, 1.0], sigma=0.25)}

with ``cfg.rewards.scales.tracking_xy = 1.0``.  The term is summed at its alphabetical position among all active terms, has
its own ``episode_sums[name]`` and ``extras["episode"]["rew_" + name]`` entry, and follows the reference's rules (scale x dt,
dropped when its scale is zero).  At most ``capi.MAX_XTERMS`` extra terms per env.
"""
from legged_gym_dev_amd import capi


def _sig(name):
    if name not in capi.SIGNALS:
        raise KeyError(f"unknown signal {name!r}; available: {sorted(capi.SIGNALS)}")
    return capi.SIGNALS[name]


class _Term:
    kind = 0

    def _fill(self, t):
        raise NotImplementedError

    def to_struct(self, scale):
        t = capi.lg_xterm()
        t.kind, t.scale = self.kind, float(scale)
        self._fill(t)
        return t


class ExpNegWeightedSqErr(_Term):
    """exp(-sum_k w[k] (a[k] - b[k])^2 / sigma)  -- the form of tracking_lin_vel / tracking_ang_vel / tracking_rom."""
    kind = capi.XT_EXP_NEG_WSQ_ERR

    def __init__(self, a, b, weights, sigma, a_off=0, b_off=0):
        self.a, self.b, self.w, self.sigma, self.a_off, self.b_off = a, b, list(weights), float(sigma), a_off, b_off
        (_, la), (_, lb) = _sig(a), _sig(b)
        if len(self.w) > 8 or a_off + len(self.w) > la or b_off + len(self.w) > lb:
            raise ValueError("weights longer than the signals (or than 8)")

    def _fill(self, t):
        t.n, t.sig_a, t.off_a, t.sig_b, t.off_b = len(self.w), _sig(self.a)[0], self.a_off, _sig(self.b)[0], self.b_off
        t.p[0] = self.sigma
        for k, w in enumerate(self.w):
            t.w[k] = float(w)


class WeightedSq(_Term):
    """sum_k w[k] a[k]^2  -- the form of orientation / ang_vel_xy / lin_vel_z."""
    kind = capi.XT_WSQ

    def __init__(self, a, weights, a_off=0):
        self.a, self.w, self.a_off = a, list(weights), a_off
        if len(self.w) > 8 or a_off + len(self.w) > _sig(a)[1]:
            raise ValueError("weights longer than the signal (or than 8)")

    def _fill(self, t):
        t.n, t.sig_a, t.off_a = len(self.w), _sig(self.a)[0], self.a_off
        for k, w in enumerate(self.w):
            t.w[k] = float(w)


class SlopedErrChange(_Term):
    """d = |(a - b)^2|_2 - |c|_2 ;  (neg_slope if d < 0 else pos_slope) * d  -- differential_error
    (legged_robot_trajectory.py:1100-1110; c = the squared error stored at the last reset)."""
    kind = capi.XT_SLOPED_ERR_CHANGE

    def __init__(self, a, b, c, n, neg_slope, pos_slope):
        self.a, self.b, self.c, self.n, self.neg, self.pos = a, b, c, int(n), float(neg_slope), float(pos_slope)
        if self.n > 8 or any(self.n > _sig(s)[1] for s in (a, b, c)):
            raise ValueError("n longer than the signals (or than 8)")

    def _fill(self, t):
        t.n, t.sig_a, t.sig_b, t.sig_c = self.n, _sig(self.a)[0], _sig(self.b)[0], _sig(self.c)[0]
        t.p[0], t.p[1] = self.neg, self.pos
