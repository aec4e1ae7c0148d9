"""Known-answer tests for the torch PPO restatement (oracle/ppo_torch.py).  rsl_rl is not available
(parity with it is unpinned, see the oracle header); these pin the restatement to hand-computed values."""
import math

import torch

from oracle import ppo_torch as pt


def test_gae_hand_computed_with_done_in_the_middle():
    g, l = 0.9, 0.5
    rewards = torch.tensor([[1.0, 2.0], [0.5, -1.0], [2.0, 0.0]])
    values = torch.tensor([[0.5, 1.0], [1.5, 0.0], [-0.5, 2.0]])
    dones = torch.tensor([[0, 0], [1, 0], [0, 0]])
    last = torch.tensor([1.0, -1.0])
    ret, adv = pt.compute_returns(rewards, dones, values, last, g, l)
    # env 0
    d2 = 2.0 + g * 1.0 - (-0.5); a2 = d2
    d1 = 0.5 + 0.0 - 1.5; a1 = d1                      # done at t=1 cuts the bootstrap and the trace
    d0 = 1.0 + g * 1.5 - 0.5; a0 = d0 + g * l * a1
    # env 1
    e2 = 0.0 + g * (-1.0) - 2.0; b2 = e2
    e1 = -1.0 + g * 2.0 - 0.0; b1 = e1 + g * l * b2
    e0 = 2.0 + g * 0.0 - 1.0; b0 = e0 + g * l * b1
    exp_adv = torch.tensor([[a0, b0], [a1, b1], [a2, b2]])
    torch.testing.assert_close(adv, exp_adv)
    torch.testing.assert_close(ret, exp_adv + values)
    n = pt.normalize_advantages(adv)
    assert abs(float(n.mean())) < 1e-6 and abs(float(n.std()) - 1.0) < 1e-5


def _algo(A=3, O=4):
    torch.manual_seed(0)
    ac = pt.ActorCritic(O, O, A, [8, 8, 8], [8, 8, 8])
    return ac, pt.PPO(ac)


def test_kl_of_identical_gaussians_and_unit_ratio():
    ac, algo = _algo()
    obs = torch.randn(16, 4)
    with torch.no_grad():
        mu = ac.actor(obs)
        acts = mu + 0.3
        lp = torch.distributions.Normal(mu, ac.std.expand_as(mu)).log_prob(acts).sum(-1, keepdim=True)
        v = ac.critic(obs)
    adv = torch.randn(16, 1)
    loss, kl, vl, sl = algo.minibatch_loss(obs, obs, acts, v, adv, v + 0.1, lp, mu, ac.std.detach().expand_as(mu))
    assert abs(float(kl) - 3 * math.log(1 + 1e-5)) < 1e-6            # sum_a ln(1 + 1e-5) with mu, sigma unchanged
    torch.testing.assert_close(sl, (-adv.squeeze()).mean())           # ratio == 1
    torch.testing.assert_close(vl, torch.tensor(0.01), rtol=1e-4, atol=1e-6)
    ent = 3 * (0.5 + 0.5 * math.log(2 * math.pi))
    torch.testing.assert_close(loss, sl + vl - 0.01 * ent, rtol=1e-5, atol=1e-6)


def test_clipped_surrogate_at_known_ratios():
    ac, algo = _algo(A=1)
    obs = torch.zeros(3, 4)
    with torch.no_grad():
        mu = ac.actor(obs)
        acts = mu.clone()
        lp_new = torch.distributions.Normal(mu, ac.std.expand_as(mu)).log_prob(acts).sum(-1, keepdim=True)
        v = ac.critic(obs)
    ratios = torch.tensor([[0.7], [1.0], [1.3]])
    lp_old = lp_new - torch.log(ratios)
    adv = torch.tensor([[1.0], [1.0], [1.0]])
    _, _, _, sl = algo.minibatch_loss(obs, obs, acts, v, adv, v, lp_old, mu, ac.std.detach().expand_as(mu))
    # max(-A r, -A clip(r, .8, 1.2)) with A = 1: ratios .7 -> -.7 ; 1 -> -1 ; 1.3 -> -1.2
    torch.testing.assert_close(sl, torch.tensor((-0.7 - 1.0 - 1.2) / 3), rtol=1e-5, atol=1e-6)
    _, _, _, sl = algo.minibatch_loss(obs, obs, acts, v, -adv, v, lp_old, mu, ac.std.detach().expand_as(mu))
    # A = -1: max(r, clip(r)) -> .8, 1, 1.3
    torch.testing.assert_close(sl, torch.tensor((0.8 + 1.0 + 1.3) / 3), rtol=1e-5, atol=1e-6)


def test_adaptive_lr_schedule_and_param_order():
    ac, algo = _algo()
    names = [n for n, _ in ac.named_parameters()]
    assert names[0] == "std" and names[1] == "actor.0.weight" and names[-1] == "critic.6.bias"
    assert pt.flat_params(ac).numel() == sum(p.numel() for p in ac.parameters())
    obs = torch.randn(32, 4)
    with torch.no_grad():
        mu = ac.actor(obs)
    args = (obs, obs, mu + 0.1, torch.zeros(32, 1), torch.randn(32, 1), torch.randn(32, 1), torch.zeros(32, 1))
    algo.step_minibatch(*args, mu, torch.ones(32, 3))                       # kl ~ 0 < 0.005 -> lr * 1.5
    assert abs(algo.learning_rate - 1.5e-3) < 1e-12
    algo.step_minibatch(*args, mu + 1.0, torch.ones(32, 3))                 # kl >> 0.02 -> lr / 1.5
    assert abs(algo.learning_rate - 1.0e-3) < 1e-12
