"""Flatten a collapsed URDF model (model/urdf.py) into the constant tables the physics
kernels consume (the ``lg_model`` struct of include/legged_hip.h).

Dynamics topology supported by the lane-parallel HIP kernel: a floating base carrying L
serial chains ("legs") of J revolute joints each (ANYmal-C/B, A1: 4x3; Cassie: 2x6).
Bodies attached by kept fixed joints (ANYmal feet, anymal_c.urdf:700) are folded into their
parent link for the dynamics but keep their own row in the net-contact-force tensor.
"""
from __future__ import annotations

import os
from typing import Dict, List

import numpy as np

from . import urdf as _urdf

MAX_DOF = 16
MAX_BODIES = 24
MAX_SPHERES = 48

_ASSET_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def resolve_model(asset_file: str, asset_name: str, collapse_fixed_joints=True,
                  replace_cylinder_with_capsule=True) -> dict:
    """Return the collapsed model for ``cfg.asset.file``.

    A URDF that exists on disk is parsed (this is the drop-in for ``gym.load_asset``).
    When it does not (the path is formatted with the reference's LEGGED_GYM_ROOT_DIR and the
    reference tree is absent, e.g. on the GPU box) the pre-compiled model of the same stem
    shipped under legged_gym_dev_amd/assets/ is used.
    """
    if asset_file and os.path.isfile(asset_file):
        return _urdf.load_urdf(asset_file, collapse_fixed_joints, replace_cylinder_with_capsule)
    stem = os.path.splitext(os.path.basename(asset_file))[0] if asset_file else asset_name
    for cand in (stem, asset_name):
        p = os.path.join(_ASSET_DIR, f"{cand}.json")
        if os.path.isfile(p):
            with open(p) as f:
                return _urdf.model_from_json(f.read())
    raise FileNotFoundError(
        f"robot asset {asset_file!r} not found and no compiled model {stem!r} in {_ASSET_DIR}")


def compile_model(model: dict) -> Dict[str, np.ndarray]:
    bodies: List[dict] = model["bodies"]
    B = len(bodies)
    if B > MAX_BODIES:
        raise ValueError(f"{B} bodies > MAX_BODIES={MAX_BODIES}")
    dof_of_body = [-1] * B       # dof index of a revolute body
    dyn_of_body = [0] * B        # dynamics link: -1 base, else dof index
    R_in_dyn = [np.eye(3)] * B   # pose of the body frame in its dynamics link frame
    p_in_dyn = [np.zeros(3)] * B
    dofs = []
    for i, b in enumerate(bodies):
        if b["joint_type"] == "floating":
            dyn_of_body[i] = -1
        elif b["joint_type"] == "revolute":
            dof_of_body[i] = len(dofs)
            dyn_of_body[i] = len(dofs)
            dofs.append(i)
        else:                    # kept fixed joint
            par = b["parent"]
            dyn_of_body[i] = dyn_of_body[par]
            R_in_dyn[i] = R_in_dyn[par] @ b["R_pj"]
            p_in_dyn[i] = p_in_dyn[par] + R_in_dyn[par] @ b["p_pj"]
    A = len(dofs)
    if A > MAX_DOF:
        raise ValueError(f"{A} dofs > MAX_DOF={MAX_DOF}")

    # dynamics links: index 0 = base, 1+d = link of dof d ; merge fixed bodies
    mass = np.zeros(A + 1)
    com = np.zeros((A + 1, 3))
    inertia = np.zeros((A + 1, 3, 3))
    for i, b in enumerate(bodies):
        k = dyn_of_body[i] + 1
        R, p = R_in_dyn[i], p_in_dyn[i]
        mass[k], com[k], inertia[k] = _urdf._add_mass(
            mass[k], com[k], inertia[k], b["mass"], p + R @ b["com"], R @ b["inertia"] @ R.T)

    parent_dof = np.full(A, -1, dtype=np.int32)     # parent dynamics link of each dof (-1 base)
    R_pj = np.zeros((A, 3, 3))
    p_pj = np.zeros((A, 3))
    axis = np.zeros((A, 3))
    for d, bi in enumerate(dofs):
        b = bodies[bi]
        par = b["parent"]
        parent_dof[d] = dyn_of_body[par]
        # joint frame pose in the parent's *dynamics* link frame
        R_pj[d] = R_in_dyn[par] @ b["R_pj"]
        p_pj[d] = p_in_dyn[par] + R_in_dyn[par] @ b["p_pj"]
        axis[d] = b["axis"]

    # serial-chain decomposition
    roots = [d for d in range(A) if parent_dof[d] == -1]
    L = len(roots)
    J = A // L if L else 0
    chains_ok = L > 0 and L * J == A
    if chains_ok:
        for l, r in enumerate(roots):
            chains_ok &= (r == l * J)
            for j in range(1, J):
                chains_ok &= (parent_dof[l * J + j] == l * J + j - 1)
    if not chains_ok:
        raise ValueError("unsupported topology: need a floating base with L equal serial chains "
                         f"in DOF order (parents={parent_dof.tolist()})")

    sph = model["spheres"]
    S = len(sph)
    if S > MAX_SPHERES:
        raise ValueError(f"{S} collision spheres > MAX_SPHERES={MAX_SPHERES}")
    sph_link = np.zeros(S, dtype=np.int32)
    sph_body = np.zeros(S, dtype=np.int32)
    sph_center = np.zeros((S, 3))
    sph_radius = np.zeros(S)
    for k, s in enumerate(sph):
        bi = s["body"]
        sph_body[k] = bi
        sph_link[k] = dyn_of_body[bi]
        sph_center[k] = p_in_dyn[bi] + R_in_dyn[bi] @ s["center"]
        sph_radius[k] = s["radius"]

    f32 = np.float32
    return {
        "num_bodies": B, "num_dofs": A, "num_legs": L, "joints_per_leg": J, "num_spheres": S,
        "mass": mass.astype(f32), "com": com.astype(f32), "inertia": inertia.astype(f32),
        "parent_dof": parent_dof, "R_pj": R_pj.astype(f32), "p_pj": p_pj.astype(f32),
        "axis": axis.astype(f32),
        "q_lower": np.array([bodies[i]["lower"] for i in dofs], f32),
        "q_upper": np.array([bodies[i]["upper"] for i in dofs], f32),
        "effort": np.array([bodies[i]["effort"] for i in dofs], f32),
        "vel_limit": np.array([bodies[i]["velocity"] for i in dofs], f32),
        "joint_damping": np.array([bodies[i]["damping"] for i in dofs], f32),
        "dof_body": np.array(dofs, dtype=np.int32),
        "body_dyn": np.array(dyn_of_body, dtype=np.int32),
        "sph_link": sph_link, "sph_body": sph_body,
        "sph_center": sph_center.astype(f32), "sph_radius": sph_radius.astype(f32),
        "body_names": list(model["body_names"]), "dof_names": list(model["dof_names"]),
        "num_shapes": int(model.get("num_shapes", 0)),
    }
