#!/usr/bin/env python3
"""Compile robot assets into the numeric model files shipped under legged_gym_dev_amd/assets/.

Run in the build container (the reference tree is mounted at /root/reference there):

    python tools/compile_assets.py [--reference /root/reference]

* URDFs (resources/robots/<robot>/urdf/*.urdf) -> <robot>.json: collapsed bodies, inertias,
  joint frames, limits, collision spheres (legged_gym_dev_amd/model/urdf.py).
* resources/actuator_nets/anydrive_v3_lstm.pt -> anydrive_v3_lstm.json: the 1 313 weights of the
  ANYdrive actuator network.  The archive is read as a plain zip: the
  raw little-endian float32 storages ``data/0..11`` are copied out and shaped according to the
  opcode listing of ``data.pkl`` (pickletools.dis -- nothing from the file is executed or
  unpickled).  Storage order: in_scale, out_scale, weight_ih_l0, weight_hh_l0, bias_ih_l0,
  bias_hh_l0, weight_ih_l1, weight_hh_l1, bias_ih_l1, bias_hh_l1, linear.weight, linear.bias.
"""
import argparse
import json
import os
import sys
import zipfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from legged_gym_dev_amd.model import urdf  # noqa: E402

ROBOTS = {
    "anymal_c": "resources/robots/anymal_c/urdf/anymal_c.urdf",
    "cassie": "resources/robots/cassie/urdf/cassie.urdf",
    "a1": "resources/robots/a1/urdf/a1.urdf",
    "anymal_b": "resources/robots/anymal_b/urdf/anymal_b.urdf",
}
LSTM_LAYOUT = [("in_scale", (2,)), ("out_scale", (1,)),
               ("weight_ih_l0", (32, 2)), ("weight_hh_l0", (32, 8)),
               ("bias_ih_l0", (32,)), ("bias_hh_l0", (32,)),
               ("weight_ih_l1", (32, 8)), ("weight_hh_l1", (32, 8)),
               ("bias_ih_l1", (32,)), ("bias_hh_l1", (32,)),
               ("linear_weight", (1, 8)), ("linear_bias", (1,))]


def extract_actuator_net(path: str) -> dict:
    out = {}
    with zipfile.ZipFile(path) as z:
        prefix = [n for n in z.namelist() if n.endswith("/version")][0].rsplit("/", 1)[0]
        for k, (name, shape) in enumerate(LSTM_LAYOUT):
            raw = z.read(f"{prefix}/data/{k}")
            arr = np.frombuffer(raw, dtype="<f4")
            if arr.size != int(np.prod(shape)):
                raise ValueError(f"storage {k} ({name}): {arr.size} floats, expected {shape}")
            out[name] = arr.reshape(shape).tolist()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    dst = os.path.join(ROOT, "legged_gym_dev_amd", "assets")
    os.makedirs(dst, exist_ok=True)
    for name, rel in ROBOTS.items():
        model = urdf.load_urdf(os.path.join(args.reference, rel))
        total = sum(b["mass"] for b in model["bodies"])
        with open(os.path.join(dst, f"{name}.json"), "w") as f:
            f.write(urdf.model_to_json(model))
        print(f"{name}: {len(model['bodies'])} bodies, {len(model['dof_names'])} dofs, "
              f"{len(model['spheres'])} spheres, mass {total:.3f} kg")
        print("  bodies:", model["body_names"])
        print("  dofs:  ", model["dof_names"])
    net = extract_actuator_net(os.path.join(args.reference, "resources/actuator_nets/anydrive_v3_lstm.pt"))
    with open(os.path.join(dst, "anydrive_v3_lstm.json"), "w") as f:
        json.dump(net, f)
    print("actuator net: in_scale", net["in_scale"], "out_scale", net["out_scale"])


if __name__ == "__main__":
    main()
