"""GEMM microbenchmark through ppok_debug_gemm: split-bf16 (x6=1) vs fp32-input MFMA (x6=0),
error against a float64 product and TFLOP/s (fp32-equivalent flops)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from legged_gym_dev_amd.lib import load
lib = load()
lib.ppok_debug_gemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
vp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [(4096, 4096, 4096), (24576, 256, 512), (24576, 512, 48), (24576, 128, 256), (24576, 512, 256), (4096, 512, 48), (4096, 256, 512)]
for dbuf in (3, 1, 0):
    lib.ppok_debug_set_x6(ctypes.c_int(dbuf))
    for (M, N, K) in shapes:
        for mode in (0, 1, 2):
            if mode == 2 and M > 4096:
                M, N, K = N, K, M                      # weight-gradient shape: reduction over the batch
            splits = 1 if mode < 2 else max(1, min(K // 256, 256 // (((M + 127) // 128) * ((N + 127) // 128))))
            A = torch.randn(M, K, device="cuda") if mode < 2 else torch.randn(K, M, device="cuda")
            B = torch.randn(N, K, device="cuda") if mode == 0 else torch.randn(K, N, device="cuda")
            C = torch.ones(M, N, device="cuda") if mode < 2 else torch.zeros(M, N, device="cuda")   # mode 1 multiplies by ELU'(aux=C=1) = 1
            lib.ppok_debug_gemm(vp(A), vp(B), vp(C), M, N, K, mode, splits, st)
            Ad = A.double() if mode < 2 else A.double().t()
            ref = Ad @ (B.double().t() if mode == 0 else B.double())
            err = float((C.double() - ref).abs().max() / ref.abs().max())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            C.fill_(1.0)
            e0.record()
            for _ in range(reps):
                lib.ppok_debug_gemm(vp(A), vp(B), vp(C), M, N, K, mode, splits, st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print(f"x6 {dbuf} mode {mode} M{M} N{N} K{K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF  relerr {err:.1e}", flush=True)
