"""PyTorch fp32 restatement of rsl_rl v1.0.2's ActorCritic / RolloutStorage / PPO.

TEST INFRASTRUCTURE (imported only by tests/, smoke() and bench.py's cpu_baseline leg).

rsl_rl is a third-party dependency of the reference (README.md:33-35 pins tag v1.0.2; call sites
legged_gym/utils/task_registry.py:37-38,148-155, scripts/train.py:44) that is NOT in the reference
tree and not installable here, and the reference holds no tests or golden vectors for it:
**PARITY WITH rsl_rl IS UNPINNED**.  This file restates its published algorithm (SURVEY.md
Appendix B) with plain torch ops + autograd; the HIP kernels are checked against it within fp32
tolerance, and it against hand-computed known answers (tests/test_ppo_oracle.py).
"""
import torch
import torch.nn as nn


class ActorCritic(nn.Module):
    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=(256, 256, 256),
                 critic_hidden_dims=(256, 256, 256), activation="elu", init_noise_std=1.0):
        super().__init__()
        act = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "lrelu": nn.LeakyReLU, "tanh": nn.Tanh,
               "sigmoid": nn.Sigmoid, "crelu": nn.ReLU}[activation]   # rsl_rl get_activation ("crelu" returns nn.ReLU() there)

        def mlp(i, hidden, o):
            layers, d = [], i
            for h in hidden:
                layers += [nn.Linear(d, h), act()]
                d = h
            layers.append(nn.Linear(d, o))
            return nn.Sequential(*layers)
        self.actor = mlp(num_actor_obs, list(actor_hidden_dims), num_actions)
        self.critic = mlp(num_critic_obs, list(critic_hidden_dims), 1)
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        self.distribution = None

    def update_distribution(self, obs):
        mean = self.actor(obs)
        self.distribution = torch.distributions.Normal(mean, mean * 0.0 + self.std)

    def act(self, obs):
        self.update_distribution(obs)
        return self.distribution.sample()

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def act_inference(self, obs):
        return self.actor(obs)

    def evaluate(self, critic_obs):
        return self.critic(critic_obs)


def flat_params(ac: ActorCritic):
    """Parameters in ActorCritic.parameters() order (std, actor W/b..., critic W/b...) -- the
    layout of lg_ppo_buffers.params."""
    return torch.cat([p.detach().reshape(-1) for p in ac.parameters()])


def compute_returns(rewards, dones, values, last_values, gamma, lam):
    """RolloutStorage.compute_returns (time-major (T, N) tensors) -> returns, raw advantages."""
    T = rewards.shape[0]
    returns = torch.zeros_like(rewards)
    adv = torch.zeros_like(last_values)
    for t in reversed(range(T)):
        next_values = last_values if t == T - 1 else values[t + 1]
        nnt = 1.0 - dones[t].float()
        delta = rewards[t] + nnt * gamma * next_values - values[t]
        adv = delta + nnt * gamma * lam * adv
        returns[t] = adv + values[t]
    advantages = returns - values
    return returns, advantages


def normalize_advantages(adv):
    return (adv - adv.mean()) / (adv.std() + 1e-8)


class PPO:
    """PPO.update over pre-filled flattened storage; one call = epochs x minibatches."""

    def __init__(self, ac, clip_param=0.2, value_loss_coef=1.0, entropy_coef=0.01, learning_rate=1e-3,
                 max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01):
        self.ac = ac
        self.clip_param, self.value_loss_coef, self.entropy_coef = clip_param, value_loss_coef, entropy_coef
        self.learning_rate, self.max_grad_norm = learning_rate, max_grad_norm
        self.use_clipped_value_loss, self.schedule, self.desired_kl = use_clipped_value_loss, schedule, desired_kl
        self.optimizer = torch.optim.Adam(ac.parameters(), lr=learning_rate)

    def minibatch_loss(self, obs, critic_obs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma):
        ac = self.ac
        ac.act(obs)
        logp = ac.get_actions_log_prob(actions)
        value = ac.evaluate(critic_obs)
        mu, sigma, entropy = ac.action_mean, ac.action_std, ac.entropy
        with torch.inference_mode():
            kl = torch.sum(torch.log(sigma / old_sigma + 1.e-5)
                           + (torch.square(old_sigma) + torch.square(old_mu - mu)) / (2.0 * torch.square(sigma)) - 0.5, axis=-1)
            kl_mean = torch.mean(kl)
        ratio = torch.exp(logp - torch.squeeze(old_logp))
        surrogate = -torch.squeeze(advantages) * ratio
        surrogate_clipped = -torch.squeeze(advantages) * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)
        surrogate_loss = torch.max(surrogate, surrogate_clipped).mean()
        if self.use_clipped_value_loss:
            value_clipped = target_values + (value - target_values).clamp(-self.clip_param, self.clip_param)
            value_loss = torch.max((value - returns).pow(2), (value_clipped - returns).pow(2)).mean()
        else:
            value_loss = (returns - value).pow(2).mean()
        loss = surrogate_loss + self.value_loss_coef * value_loss - self.entropy_coef * entropy.mean()
        return loss, kl_mean, value_loss, surrogate_loss

    def step_minibatch(self, *batch):
        loss, kl_mean, vl, sl = self.minibatch_loss(*batch)
        if self.desired_kl is not None and self.schedule == "adaptive":
            if kl_mean > self.desired_kl * 2.0:
                self.learning_rate = max(1e-5, self.learning_rate / 1.5)
            elif kl_mean < self.desired_kl / 2.0 and kl_mean > 0.0:
                self.learning_rate = min(1e-2, self.learning_rate * 1.5)
            for g in self.optimizer.param_groups:
                g["lr"] = self.learning_rate
        self.optimizer.zero_grad()
        loss.backward()
        nn.utils.clip_grad_norm_(self.ac.parameters(), self.max_grad_norm)
        self.optimizer.step()
        return float(kl_mean), float(vl), float(sl)
