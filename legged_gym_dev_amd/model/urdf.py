"""URDF -> collapsed articulated model ("asset loading" for the HIP physics).

Replaces what the reference obtains from Isaac Gym's ``gym.load_asset`` and the
``get_asset_*`` queries (legged_gym/envs/base/legged_robot.py:693-724) under the asset options
of legged_robot_config.py:104-124:

* ``collapse_fixed_joints``: links joined by fixed joints are merged into their parent
  (mass, centre of mass, inertia, collision shapes) unless the joint carries
  ``dont_collapse="true"`` (anymal_c.urdf foot joints) -- those stay separate *bodies*
  (they report their own net contact force) but are rigidly attached for the dynamics.
* ``replace_cylinder_with_capsule``: cylinders become capsules of the same radius whose
  segment spans the cylinder length; for ground contact a capsule is represented by
  spheres on its segment (ends, plus the middle for long ones).
* Body / DOF order: depth first, siblings in alphabetical order of the child link name
  (gives LF, LH, RF, RH for ANYmal-C -- the order the reference configs assume,
  anymal_c_rough_config.py:43-58; Isaac Gym's own ordering cannot be checked offline,
  SURVEY.md §8(c), so it is explicit in the compiled model file).
"""
from __future__ import annotations

import json
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional

import numpy as np


def rpy_to_matrix(rpy) -> np.ndarray:
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return rz @ ry @ rx


def _vec(text: Optional[str], n=3) -> np.ndarray:
    if text is None:
        return np.zeros(n)
    return np.array([float(t) for t in text.split()], dtype=np.float64)


def _origin(elem) -> tuple:
    """(R, p) of an <origin> child, identity when absent."""
    o = elem.find("origin") if elem is not None else None
    if o is None:
        return np.eye(3), np.zeros(3)
    return rpy_to_matrix(_vec(o.get("rpy"))), _vec(o.get("xyz"))


class _Link:
    def __init__(self, name):
        self.name = name
        self.mass = 0.0
        self.com = np.zeros(3)
        self.inertia = np.zeros((3, 3))      # about com, link axes
        self.shapes: List[dict] = []          # in link frame
        self.children: List["_Joint"] = []


class _Joint:
    def __init__(self):
        self.name = ""
        self.type = "fixed"
        self.parent = ""
        self.child = ""
        self.R = np.eye(3)
        self.p = np.zeros(3)
        self.axis = np.array([1.0, 0.0, 0.0])
        self.lower = 0.0
        self.upper = 0.0
        self.effort = 0.0
        self.velocity = 0.0
        self.damping = 0.0
        self.friction = 0.0
        self.keep = False                     # dont_collapse


def _add_mass(m1, c1, i1, m2, c2, i2):
    """Combine two rigid bodies given (mass, com, inertia-about-com) in one frame."""
    m = m1 + m2
    if m <= 0.0:
        return 0.0, np.zeros(3), np.zeros((3, 3))
    c = (m1 * c1 + m2 * c2) / m

    def shift(mass, ic, d):
        return ic + mass * (np.dot(d, d) * np.eye(3) - np.outer(d, d))

    return m, c, shift(m1, i1, c1 - c) + shift(m2, i2, c2 - c)


def _parse_shape(col) -> Optional[dict]:
    geom = col.find("geometry")
    if geom is None or len(geom) == 0:
        return None
    g = list(geom)[0]
    R, p = _origin(col)
    if g.tag == "sphere":
        return {"type": "sphere", "R": R, "p": p, "radius": float(g.get("radius"))}
    if g.tag == "cylinder":
        return {"type": "cylinder", "R": R, "p": p, "radius": float(g.get("radius")),
                "length": float(g.get("length"))}
    if g.tag == "box":
        return {"type": "box", "R": R, "p": p, "size": _vec(g.get("size"))}
    return None                               # meshes carry no analytic collision here


def load_urdf(path: str, collapse_fixed_joints: bool = True,
              replace_cylinder_with_capsule: bool = True) -> dict:
    root = ET.parse(path).getroot()
    links: Dict[str, _Link] = {}
    num_shapes = 0                            # every <collision> geometry is one rigid shape of the asset (meshes: one convex hull
    for le in root.findall("link"):           # each with vhacd off, legged_robot_config.py:104-124) -- len(rigid_shape_props)
        lk = _Link(le.get("name"))
        ine = le.find("inertial")
        if ine is not None:
            R, p = _origin(ine)
            lk.mass = float(ine.find("mass").get("value"))
            it = ine.find("inertia")
            if it is not None:
                ixx, ixy, ixz = float(it.get("ixx")), float(it.get("ixy")), float(it.get("ixz"))
                iyy, iyz, izz = float(it.get("iyy")), float(it.get("iyz")), float(it.get("izz"))
                I = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
                lk.inertia = R @ I @ R.T
            lk.com = p
        for col in le.findall("collision"):
            geom = col.find("geometry")
            num_shapes += int(geom is not None and len(geom) > 0)
            s = _parse_shape(col)
            if s is not None:
                lk.shapes.append(s)
        if ine is None and lk.shapes:
            # Isaac Gym would give such a link the mass of its shapes at asset.density (legged_robot.py:698); not implemented
            raise NotImplementedError(f"link {lk.name!r} has collision geometry but no <inertial>: deriving its mass from "
                                      "cfg.asset.density is not implemented")
        links[lk.name] = lk

    has_parent = set()
    for je in root.findall("joint"):
        j = _Joint()
        j.name, j.type = je.get("name"), je.get("type")
        j.parent, j.child = je.find("parent").get("link"), je.find("child").get("link")
        j.R, j.p = _origin(je)
        ax = je.find("axis")
        if ax is not None:
            a = _vec(ax.get("xyz"))
            j.axis = a / max(np.linalg.norm(a), 1e-12)
        lim = je.find("limit")
        if lim is not None:
            j.lower = float(lim.get("lower", 0.0))
            j.upper = float(lim.get("upper", 0.0))
            j.effort = float(lim.get("effort", 0.0))
            j.velocity = float(lim.get("velocity", 0.0))
        dyn = je.find("dynamics")
        if dyn is not None:
            j.damping = float(dyn.get("damping", 0.0))
            j.friction = float(dyn.get("friction", 0.0))
        j.keep = je.get("dont_collapse", "false").lower() == "true"
        if j.type not in ("fixed", "revolute", "continuous"):
            raise ValueError(f"unsupported joint type {j.type!r} on joint {j.name}")
        links[j.parent].children.append(j)
        has_parent.add(j.child)
    roots = [n for n in links if n not in has_parent]
    if len(roots) != 1:
        raise ValueError(f"URDF must have exactly one root link, found {roots}")

    bodies: List[dict] = []

    def absorb(body: dict, link: _Link, R: np.ndarray, p: np.ndarray):
        """Merge `link` (pose (R,p) in the body frame) into `body`, recursing through
        collapsible fixed joints; returns the kinematic children as (joint, R, p)."""
        body["mass"], body["com"], body["inertia"] = _add_mass(
            body["mass"], body["com"], body["inertia"],
            link.mass, p + R @ link.com, R @ link.inertia @ R.T)
        for s in link.shapes:
            body["shapes"].append({**s, "R": R @ s["R"], "p": p + R @ s["p"]})
        kids = []
        for j in link.children:
            Rc, pc = R @ j.R, p + R @ j.p
            if j.type == "fixed" and collapse_fixed_joints and not j.keep:
                kids.extend(absorb(body, links[j.child], Rc, pc))
            else:
                kids.append((j, Rc, pc))
        return kids

    def build(link: _Link, parent: int, joint: Optional[_Joint], R, p):
        body = {"name": link.name, "parent": parent, "mass": 0.0, "com": np.zeros(3),
                "inertia": np.zeros((3, 3)), "shapes": [],
                "joint_name": joint.name if joint else "", "R_pj": R, "p_pj": p,
                "joint_type": "floating" if joint is None else
                ("fixed" if joint.type == "fixed" else "revolute"),
                "axis": joint.axis if joint else np.array([1.0, 0, 0]),
                "lower": joint.lower if joint else 0.0, "upper": joint.upper if joint else 0.0,
                "effort": joint.effort if joint else 0.0,
                "velocity": joint.velocity if joint else 0.0,
                "damping": joint.damping if joint else 0.0,
                "friction": joint.friction if joint else 0.0}
        idx = len(bodies)
        bodies.append(body)
        kids = absorb(body, link, np.eye(3), np.zeros(3))
        for j, Rc, pc in sorted(kids, key=lambda t: t[0].child):
            build(links[j.child], idx, j, Rc, pc)

    build(links[roots[0]], -1, None, np.eye(3), np.zeros(3))

    spheres = []
    for bi, b in enumerate(bodies):
        for s in b.pop("shapes"):
            if s["type"] == "sphere":
                spheres.append({"body": bi, "center": s["p"], "radius": s["radius"]})
            elif s["type"] == "cylinder":
                half = 0.5 * s["length"]
                zaxis = s["R"][:, 2]
                offs = [-half, half] if s["length"] < 0.3 else [-half, 0.0, half]
                if not replace_cylinder_with_capsule:
                    offs = [-half, half]
                for o in offs:
                    spheres.append({"body": bi, "center": s["p"] + o * zaxis, "radius": s["radius"]})
            elif s["type"] == "box":
                hx, hy, hz = 0.5 * s["size"]
                h = np.array([hx, hy, hz])
                order = np.argsort(h)
                if h[order[2]] >= 2.5 * h[order[1]]:
                    # a bar (A1 thigh / calf): spheres along its long axis, like a capsule of the bar's mean half-width
                    rad = 0.5 * (h[order[0]] + h[order[1]])
                    axis = np.zeros(3)
                    axis[order[2]] = 1.0
                    reach = h[order[2]] - rad
                    offs = [-reach, reach] if 2 * reach < 0.3 else [-reach, 0.0, reach]
                    for o in offs:
                        spheres.append({"body": bi, "center": s["p"] + s["R"] @ (o * axis), "radius": float(rad)})
                    continue
                rad = 0.25 * min(hx, hy, hz) + 1e-3
                for sx in (-1, 1):
                    for sy in (-1, 1):
                        for sz in (-1, 1):
                            c = np.array([sx * (hx - rad), sy * (hy - rad), sz * (hz - rad)])
                            spheres.append({"body": bi, "center": s["p"] + s["R"] @ c, "radius": rad})

    dof_names = [b["joint_name"] for b in bodies if b["joint_type"] == "revolute"]
    return {"name": root.get("name", "robot"), "bodies": bodies, "spheres": spheres, "num_shapes": num_shapes,
            "dof_names": dof_names, "body_names": [b["name"] for b in bodies]}


# ---------------------------------------------------------------- (de)serialisation helpers
def model_to_json(model: dict) -> str:
    def conv(o):
        if isinstance(o, np.ndarray):
            return o.tolist()
        if isinstance(o, (np.floating, np.integer)):
            return o.item()
        raise TypeError(type(o))
    return json.dumps(model, default=conv, indent=1)


def model_from_json(text: str) -> dict:
    m = json.loads(text)
    for b in m["bodies"]:
        for k in ("com", "inertia", "R_pj", "p_pj", "axis"):
            b[k] = np.array(b[k], dtype=np.float64)
    for s in m["spheres"]:
        s["center"] = np.array(s["center"], dtype=np.float64)
    return m
