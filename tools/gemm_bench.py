"""fp32-MFMA GEMM microbenchmark through ppok_debug_gemm (correctness vs torch + TFLOP/s)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from legged_gym_dev_amd.lib import load
lib = load()
lib.ppok_debug_gemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
vp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [(4096, 4096, 4096), (24576, 256, 512), (24576, 512, 48), (24576, 128, 256), (24576, 512, 256), (4096, 512, 48), (4096, 256, 512)]
for dbuf in (1, 0):
    lib.ppok_debug_set_dbuf(ctypes.c_int(dbuf))
    for (M, N, K) in shapes:
        for mode in (0, 1):
            A = torch.randn(M, K, device="cuda")
            B = torch.randn(N, K, device="cuda") if mode == 0 else torch.randn(K, N, device="cuda")
            C = torch.ones(M, N, device="cuda")        # mode 1 multiplies by ELU'(aux=C=1) = 1
            lib.ppok_debug_gemm(vp(A), vp(B), vp(C), M, N, K, mode, 1, st)
            ref = A @ (B.t() if mode == 0 else B)
            err = float((C - ref).abs().max() / ref.abs().max())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            C.fill_(1.0)
            e0.record()
            for _ in range(reps):
                lib.ppok_debug_gemm(vp(A), vp(B), vp(C), M, N, K, mode, 1, st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print(f"dbuf {dbuf} mode {mode} M{M} N{N} K{K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF  relerr {err:.1e}", flush=True)
