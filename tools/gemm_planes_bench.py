"""Forward / input-gradient GEMMs on the weight planes (ppok_debug_gemm_planes) beside the fp32-operand split-bf16 kernels
(ppok_debug_gemm), update shapes.  us and fp32-equivalent TFLOP/s."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from legged_gym_dev_amd.lib import load
lib = load()
lib.ppok_debug_gemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
lib.ppok_debug_gemm_planes.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
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


for M, rows, cols in ((24576, 256, 512), (24576, 128, 256), (24576, 512, 48), (4096, 256, 512), (4096, 128, 256)):
    W = torch.randn(rows, cols, device="cuda")
    stride = (rows * cols + 7) // 8 * 8
    planes = torch.zeros(3 * stride + 8, dtype=torch.int16, device="cuda")
    for mode in (0, 1):
        K, N = (cols, rows) if mode == 0 else (rows, cols)
        A = torch.randn(M, K, device="cuda")
        C = torch.ones(M, N, device="cuda")
        t_pl = timeit(lambda: lib.ppok_debug_gemm_planes(vp(A), vp(W), vp(C), vp(planes), stride, M, rows, cols, mode, st))
        t_sp = timeit(lambda: lib.ppok_debug_gemm_planes(vp(A), vp(W), vp(C), vp(planes), stride, 64, rows, cols, mode, st))   # split kernel + tiny gemm
        t_f = timeit(lambda: lib.ppok_debug_gemm(vp(A), vp(W), vp(C), M, N, K, mode, 1, st))
        fl = 2.0 * M * N * K
        print(f"M{M} W[{rows}][{cols}] mode {mode} ({'fwd' if mode == 0 else 'dX '}): planes {t_pl:7.1f} us (incl. ~{t_sp:5.1f} us split+launch) "
              f"{fl / t_pl / 1e6:6.1f} TF | fp32 operand {t_f:7.1f} us {fl / t_f / 1e6:6.1f} TF", flush=True)
