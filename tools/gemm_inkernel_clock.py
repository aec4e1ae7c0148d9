"""In-kernel clock of the forward GEMM's workgroups under sustained load (MI355X_MICROARCH.md, DVFS give-back item 6): needs the diagnostic build
    make -C legged_gym_dev_amd/csrc exp EXPFLAGS=-DLG_EXP_GEMM_CLOCK ;  LG_HIP_LIB=.../liblegged_hip_exp.so python tools/gemm_inkernel_clock.py
Back-to-back launches of the update's second-layer forward (W [256][512], weight planes) for >= 2 s on random and on zero operands, then the
median over workgroups of d s_memtime / d s_memrealtime x 100 MHz of the last launch."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from legged_gym_dev_amd.lib import load
lib = load()
lib.ppok_debug_gemm_planes.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
vp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
rows, cols, M = 256, 512, 49152
stride = (rows * cols + 7) // 8 * 8
for name, gen in (("random", torch.randn), ("zeros ", torch.zeros)):
    W, A = gen(rows, cols, device="cuda"), gen(M, cols, device="cuda")
    C = torch.ones(M, rows, device="cuda")
    planes = torch.zeros(3 * stride + 8, dtype=torch.int16, device="cuda")
    lib.ppok_debug_gemm_planes(vp(A), vp(W), vp(C), vp(planes), stride, M, rows, cols, 0, st)
    torch.cuda.synchronize()
    t0, n = time.perf_counter(), 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while time.perf_counter() - t0 < 2.5:
        e0.record()
        for _ in range(200):
            lib.ppok_debug_gemm_planes(vp(A), vp(W), vp(C), vp(planes), stride, M, rows, cols, 0, st)
        e1.record(); torch.cuda.synchronize(); n += 200
    us = e0.elapsed_time(e1) / 200 * 1e3
    buf = (ctypes.c_ulonglong * 4096)()
    lib.ppok_debug_read_gemm_clock(buf)
    v = np.array(buf[:], dtype=np.float64).reshape(2048, 2)[: M // 128 * 2]
    clk = v[:, 0] / v[:, 1] * 0.1
    print(f"{name} operands: {us:7.1f} us per launch incl. the plane split ({2.0 * M * rows * cols / us / 1e6:6.1f} TF), after {n} launches: in-kernel clock "
          f"median {np.median(clk):.2f} GHz (10 % {np.percentile(clk, 10):.2f}, 90 % {np.percentile(clk, 90):.2f}); workgroup life {np.median(v[:, 1]) / 100:.1f} us", flush=True)
