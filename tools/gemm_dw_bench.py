"""Weight-gradient GEMM (A^T . B over the minibatch rows, split-K with atomics) against the split count, update shapes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from legged_gym_dev_amd.lib import load
lib = load()
lib.ppok_debug_gemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
vp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


K = 24576
for M, N in ((256, 512), (128, 256), (512, 48), (12, 128)):
    A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda"); C = torch.zeros(M, N, device="cuda")
    for splits in (4, 8, 16, 32, 64, 96):
        t = timeit(lambda: lib.ppok_debug_gemm(vp(A), vp(B), vp(C), M, N, K, 2, splits, st))
        print(f"dW M{M} N{N} K{K} splits {splits:3d}: {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.1f} TF", flush=True)
for M, N, Kk in ((24576, 512, 256), (24576, 256, 512)):
    A = torch.randn(M, Kk, device="cuda"); B = torch.randn(N, Kk, device="cuda"); C = torch.zeros(M, N, device="cuda")
    t = timeit(lambda: lib.ppok_debug_gemm(vp(A), vp(B), vp(C), M, N, Kk, 0, 1, st))
    print(f"fwd M{M} N{N} K{Kk}: {t:7.1f} us {2.0 * M * N * Kk / t / 1e6:6.1f} TF", flush=True)
