"""Per-k-tile cost of the 64x64 rollout configuration of k_gemm on the weight planes: forward GEMM of M rows, K swept
(times are net of the plane split + a one-tile launch, i.e. relative; the two-stage variant LG_GEMM_LDB lives in the exp build: make exp EXPFLAGS=-DLG_EXP_KERNELS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from legged_gym_dev_amd.lib import load
lib = load()
lib.ppok_debug_gemm_planes.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
vp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M in (2048, 4096, 8192):
    for N in (128, 256, 512):
        row = []
        for K in (32, 64, 128, 256, 512, 1024):
            A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); C = torch.ones(M, N, device="cuda")
            stride = (N * K + 7) // 8 * 8
            planes = torch.zeros(3 * stride + 8, dtype=torch.int16, device="cuda")
            t = timeit(lambda: lib.ppok_debug_gemm_planes(vp(A), vp(W), vp(C), vp(planes), stride, M, N, K, 0, st))
            t0 = timeit(lambda: lib.ppok_debug_gemm_planes(vp(A), vp(W), vp(C), vp(planes), stride, 64, N, K, 0, st))   # split kernel + a one-tile gemm
            row.append(t - t0)
        print(f"M {M} N {N}: " + "  ".join(f"K{k}: {t:6.1f}" for k, t in zip((32, 64, 128, 256, 512, 1024), row)) +
              f"   us; per k-tile {(row[-1] - row[-2]) / 16:5.2f} us", flush=True)
