"""ctypes binding of liblegged_hip.so + zero-copy torch views of library-owned HBM.

There is no CPU fallback: if the library is missing or no HIP device is present the calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import capi

_DIR = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("LG_HIP_LIB") or os.path.join(_DIR, "lib", "liblegged_hip.so")   # LG_HIP_LIB: A/B builds of the same tree
_lib = None


class LeggedHipError(RuntimeError):
    pass


def build(force=False):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_DIR, "csrc")
    if force and os.path.isfile(SO_PATH):
        os.remove(SO_PATH)
    subprocess.check_call(["make", "-C", src, "-s", "-j4"])
    return SO_PATH


def load():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  -- first, so that liblegged_hip binds to the HIP runtime instance torch uses
        if not os.path.isfile(SO_PATH):
            raise LeggedHipError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the product path)")
        _lib = C.CDLL(SO_PATH)
        capi.declare_env_api(_lib, prefix="lg_")
        _declare_ppo(_lib)
    return _lib


def _declare_ppo(lib):
    vp = C.c_void_p
    if not hasattr(lib, "lg_ppo_create"):
        return
    lib.lg_ppo_create.argtypes = [C.POINTER(capi.lg_ppo_cfg), C.POINTER(vp)]
    lib.lg_ppo_destroy.argtypes = [vp]
    lib.lg_ppo_get_buffers.argtypes = [vp, C.POINTER(capi.lg_ppo_buffers)]
    lib.lg_ppo_set_stream.argtypes = [vp, vp]
    lib.lg_ppo_param_layout.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]
    lib.lg_ppo_inject_noise.argtypes = [vp, C.c_int]
    lib.lg_ppo_act.argtypes = [vp, vp, vp]
    lib.lg_ppo_process_env_step.argtypes = [vp, vp, vp, vp]
    lib.lg_ppo_compute_returns.argtypes = [vp, vp]
    lib.lg_ppo_normalize_advantages.argtypes = [vp]
    lib.lg_ppo_begin_update.argtypes = [vp]
    lib.lg_ppo_minibatch_backward.argtypes = [vp, C.c_int, C.c_int]
    lib.lg_ppo_minibatch_step.argtypes = [vp]
    lib.lg_ppo_end_update.argtypes = [vp]
    lib.lg_ppo_act_inference.argtypes = [vp, vp, vp, C.c_int64]
    lib.lg_ppo_params_changed.argtypes = [vp]
    lib.lg_ppo_set_deterministic.argtypes = [vp, C.c_int]
    lib.lg_ppo_attach_env.argtypes = [vp, vp]
    lib.lg_ppo_debug_bucket_extents.argtypes = [vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]


_TORCH_DT = {"f4": "float32", "u1": "uint8", "i8": "int64", "i4": "int32"}


class _DevArray:
    """Minimal __cuda_array_interface__ carrier (PyTorch-ROCm consumes it like CUDA)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(int(s) for s in shape),
                                         "typestr": typestr, "version": 2, "strides": None}
        self._owner = owner


def device_tensor(ptr, shape, dt, owner, device):
    """torch tensor aliasing `ptr` (no copy).  dt in {'f4','u1','i8','i4'}."""
    import torch
    typestr = {"f4": "<f4", "u1": "|u1", "i8": "<i8", "i4": "<i4"}[dt]
    t = torch.as_tensor(_DevArray(ptr, shape, typestr, owner), device=device)
    if t.data_ptr() != int(ptr):
        raise LeggedHipError("torch copied the buffer instead of aliasing it")
    return t


class HipEnvCore:
    """One lg_ctx + torch views of its buffers."""

    def __init__(self, setup, height_samples=None, device="cuda:0"):
        import torch
        self.lib = load()
        self.setup = setup
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise LeggedHipError("the HIP env needs a GPU device (no CPU fallback); got " + str(device))
        torch.cuda.set_device(self.device)
        cfg, model, keep = setup.to_structs()
        hs = None
        if height_samples is not None:
            hs = np.ascontiguousarray(height_samples, dtype=np.int16)
        self.ctx = C.c_void_p()
        rc = self.lib.lg_create(C.byref(cfg), C.byref(model), hs.ctypes.data if hs is not None else None,
                                C.byref(self.ctx))
        if rc != 0:
            raise LeggedHipError(f"lg_create failed ({rc}): {self.lib.lg_last_error().decode()}")
        self.cfg_struct = cfg
        bufs = capi.lg_buffers()
        self.lib.lg_get_buffers(self.ctx, C.byref(bufs))
        shapes = capi.buffer_shapes(setup.num_envs, setup.num_dof, setup.num_bodies, cfg.num_obs,
                                    len(setup.feet_indices), setup.num_height_points,
                                    traj_N=setup.traj["N"] if setup.traj else 0, traj_dN=setup.traj["dN"] if setup.traj else 1)
        self.t = {}
        for name, (shape, dt) in shapes.items():
            ptr = C.cast(getattr(bufs, name), C.c_void_p).value
            self.t[name] = device_tensor(ptr, shape, dt, self, self.device)
        self.use_current_stream()

    def use_current_stream(self):
        import torch
        self.lib.lg_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))

    def call(self, fn, *args):
        rc = getattr(self.lib, "lg_" + fn)(self.ctx, *args)
        if rc != 0:
            raise LeggedHipError(f"lg_{fn} failed ({rc}): {self.lib.lg_last_error().decode()}")

    def step(self, actions):
        self.call("step", C.c_void_p(actions.data_ptr()))

    def close(self):
        if getattr(self, "ctx", None):
            self.t = {}
            rc = self.lib.lg_destroy(self.ctx)
            if rc != 0:                                   # e.g. a learner is still attached (lg_ppo_attach_env): the context stays alive
                raise LeggedHipError(f"lg_destroy failed ({rc}): {self.lib.lg_last_error().decode()}")
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
