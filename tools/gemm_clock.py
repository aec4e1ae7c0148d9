"""Is the split-bf16 GEMM held back by its own structure or by the clock the chip sustains under load?
Same kernel, same shape: random operands vs all-zero operands (zeros draw far less power, so the chip keeps
its clock: MI355X_MICROARCH.md, DVFS give-back).  A large gap = power/clock-limited, not issue-limited."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from legged_gym_dev_amd.lib import load
lib = load()
lib.ppok_debug_gemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
vp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in ((4096, 4096, 4096), (24576, 512, 256)):
    for x6 in (3, 0):
        lib.ppok_debug_set_x6(ctypes.c_int(x6))
        for name, gen in (("random", torch.randn), ("zeros ", torch.zeros)):
            A, B, C = gen(M, K, device="cuda"), gen(N, K, device="cuda"), torch.empty(M, N, device="cuda")
            for _ in range(3):
                lib.ppok_debug_gemm(vp(A), vp(B), vp(C), M, N, K, 0, 1, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 200 if M * N * K > 1e10 else 1000          # >= 0.15 s of back-to-back launches
            e0.record()
            for _ in range(reps):
                lib.ppok_debug_gemm(vp(A), vp(B), vp(C), M, N, K, 0, 1, st)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            print(f"x6={x6} {M}x{N}x{K} {name}: {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TF", flush=True)
