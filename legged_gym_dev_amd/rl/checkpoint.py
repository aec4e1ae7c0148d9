"""Checkpoint and export formats of the reference stack, produced from / consumed into the flat HBM
parameter buffers of the HIP learner.

* ``model_<it>.pt`` (rsl_rl ``OnPolicyRunner.save``; read by scripts/play.py:69-76 through
  ``task_registry.make_alg_runner(resume=True)``, legged_gym/utils/task_registry.py:150-155):
  ``{"model_state_dict": ActorCritic.state_dict(), "optimizer_state_dict": torch.optim.Adam.state_dict(),
  "iter": int, "infos": ...}``.  The optimiser entry is written in torch's own Adam layout (per-parameter
  ``step / exp_avg / exp_avg_sq`` in ``ActorCritic.parameters()`` order) so either stack can resume the other's run.
* ``policy_1.pt``: TorchScript of the actor MLP (legged_gym/utils/helpers.py:274-285, play.py:78-82).

Pure torch/CPU code: nothing here is on the training path.
"""
import copy
import os
from collections import OrderedDict

import torch

ACTIVATIONS = {"elu": torch.nn.ELU, "selu": torch.nn.SELU, "relu": torch.nn.ReLU, "lrelu": torch.nn.LeakyReLU,
               "tanh": torch.nn.Tanh, "sigmoid": torch.nn.Sigmoid}


def parameter_order(state_dict):
    """Names in ``ActorCritic.parameters()`` order: std, actor.*, critic.* (registration order in rsl_rl)."""
    names = list(state_dict.keys())
    rank = lambda n: (0 if n == "std" else 1 if n.startswith("actor.") else 2, int(n.split(".")[1]) if "." in n else 0,
                      0 if n.endswith("weight") else 1)
    return sorted(names, key=rank)


def adam_state_to_torch(names, shapes, adam_m, adam_v, step, lr):
    """Flat first/second moments -> ``torch.optim.Adam.state_dict()`` (betas 0.9/0.999, eps 1e-8)."""
    state, off = {}, 0
    for i, n in enumerate(names):
        cnt = 1
        for d in shapes[n]:
            cnt *= d
        state[i] = {"step": torch.tensor(float(step)), "exp_avg": adam_m[off:off + cnt].reshape(shapes[n]).clone().cpu(),
                    "exp_avg_sq": adam_v[off:off + cnt].reshape(shapes[n]).clone().cpu()}
        off += cnt
    group = {"lr": float(lr), "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(names)))}
    return {"state": state, "param_groups": [group]}


def adam_state_from_torch(sd, names, shapes):
    """Inverse of :func:`adam_state_to_torch`; also accepts this package's first-round layout
    (``adam_m / adam_v / step / lr``).  Returns (m_flat, v_flat, step, lr); a fresh optimiser (empty state) gives zeros."""
    if "adam_m" in sd:
        return sd["adam_m"].float().flatten(), sd["adam_v"].float().flatten(), float(sd["step"]), float(sd["lr"])
    ms, vs, step = [], [], 0.0
    for i, n in enumerate(names):
        st = sd["state"].get(i, sd["state"].get(str(i)))
        if st is None:
            z = torch.zeros(shapes[n]).flatten()
            ms.append(z); vs.append(z.clone())
            continue
        if tuple(st["exp_avg"].shape) != tuple(shapes[n]):
            raise ValueError(f"optimizer state of parameter {i} ({n}) has shape {tuple(st['exp_avg'].shape)}, expected {tuple(shapes[n])}")
        ms.append(st["exp_avg"].float().flatten()); vs.append(st["exp_avg_sq"].float().flatten())
        step = max(step, float(st["step"]))
    return torch.cat(ms), torch.cat(vs), step, float(sd["param_groups"][0]["lr"])


def build_mlp(state_dict, prefix="actor", activation="elu"):
    """``nn.Sequential`` [Linear, act, ..., Linear] with rsl_rl's module indices (actor.0, actor.2, ...)."""
    act = ACTIVATIONS[activation]
    idx = sorted({int(k.split(".")[1]) for k in state_dict if k.startswith(prefix + ".")})
    layers = OrderedDict()
    for n, i in enumerate(idx):
        w = state_dict[f"{prefix}.{i}.weight"]
        lin = torch.nn.Linear(w.shape[1], w.shape[0])
        with torch.no_grad():
            lin.weight.copy_(w.detach().cpu())
            lin.bias.copy_(state_dict[f"{prefix}.{i}.bias"].detach().cpu())
        layers[str(i)] = lin
        if n + 1 < len(idx):
            layers[str(i + 1)] = act()
    return torch.nn.Sequential(layers)


def export_policy_as_jit(actor_critic, path, activation=None):
    """helpers.py:274-285: ``<path>/policy_1.pt`` = torch.jit.script(actor MLP on CPU)."""
    os.makedirs(path, exist_ok=True)
    sd = actor_critic.state_dict()
    activation = activation or getattr(actor_critic, "activation", "elu")
    model = copy.deepcopy(build_mlp(sd, "actor", activation)).to("cpu")
    scripted = torch.jit.script(model)
    out = os.path.join(path, "policy_1.pt")
    scripted.save(out)
    return out
