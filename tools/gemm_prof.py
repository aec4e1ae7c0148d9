"""One GEMM shape through ppok_debug_gemm, a few launches -- the target of rocprofv3 --pmc runs.

    python tools/gemm_prof.py M N K MODE X6 [REPS]
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from legged_gym_dev_amd.lib import load
M, N, K, mode, x6 = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
lib = load()
lib.ppok_debug_gemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
lib.ppok_debug_set_x6(ctypes.c_int(x6))
vp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
A = torch.randn(M, K, device="cuda") if mode < 2 else torch.randn(K, M, device="cuda")
B = torch.randn(N, K, device="cuda") if mode == 0 else torch.randn(K, N, device="cuda")
C = torch.ones(M, N, device="cuda")
splits = 1 if mode < 2 else max(1, min(K // 256, 256 // (((M + 127) // 128) * ((N + 127) // 128))))
for _ in range(reps):
    lib.ppok_debug_gemm(vp(A), vp(B), vp(C), M, N, K, mode, splits, st)
torch.cuda.synchronize()
print("done")
