"""Tile quantisation of the forward GEMM (128 x 128 tiles, two 8-wave workgroups per CU = 512 slots): time against the number
of workgroups, one net, W [256][512] (the update's second layer), weight planes.  us, us per 512-workgroup round, TF."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from legged_gym_dev_amd.lib import load
lib = load()
lib.ppok_debug_gemm_planes.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
vp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


rows, cols = 256, 512
W = torch.randn(rows, cols, device="cuda")
stride = (rows * cols + 7) // 8 * 8
planes = torch.zeros(3 * stride + 8, dtype=torch.int16, device="cuda")
base = None
for mode in (0, 1):
    for wgs in (128, 256, 384, 512, 640, 768, 1024, 1536, 2048):
        K, N = (cols, rows) if mode == 0 else (rows, cols)
        M = wgs * 128 // (N // 128)
        A = torch.randn(M, K, device="cuda")
        C = torch.ones(M, N, device="cuda")
        t_sp = timeit(lambda: lib.ppok_debug_gemm_planes(vp(A), vp(W), vp(C), vp(planes), stride, 64, rows, cols, mode, st))
        t = timeit(lambda: lib.ppok_debug_gemm_planes(vp(A), vp(W), vp(C), vp(planes), stride, M, rows, cols, mode, st)) - t_sp
        print(f"mode {mode} {wgs:5d} workgroups (M {M:6d}): {t:7.1f} us  {t / wgs * 512:7.1f} us per 512 workgroups  {2.0 * M * N * K / t / 1e6:6.1f} TF", flush=True)
