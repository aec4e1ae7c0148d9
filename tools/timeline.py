"""Timelines of one PPO minibatch and one rollout step out of a rocprofv3 kernel trace (csv):
    python tools/timeline.py gpurun_out/<dir>"""
import csv, glob, os, sys

d = sys.argv[1]
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)   # gpurun merges: older collections may lie beside
trace = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in trace]


def timeline(lo, hi):
    t0 = int(trace[lo]["Start_Timestamp"])
    out = []
    for r in trace[lo:hi]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        out.append(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:86]}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
    return "\n".join(out)


def pick(starts, must):
    """the last window [starts[k], starts[k+1]) that contains every kernel of `must` (bench.py also times bare sequences)"""
    for k in range(len(starts) - 2, -1, -1):
        w = names[starts[k]:starts[k + 1]]
        if all(any(m in n for n in w) for m in must):
            return starts[k], starts[k + 1]
    raise SystemExit(f"no window with {must}")


gi = [i for i, n in enumerate(names) if n.startswith("k_opt_adam")]     # a minibatch period = optimiser step .. next optimiser step
si = [i for i, n in enumerate(names) if "k_substeps" in n]
lo, hi = pick(gi, ["k_opt_prepare", "k_gemm_dw"])
print("== one PPO minibatch of the update (optimiser step .. next optimiser step) ==")
print(timeline(lo, hi))
print(f"   minibatch period: {(int(trace[hi]['Start_Timestamp']) - int(trace[lo]['Start_Timestamp'])) / 1e3:.1f} us")
print("\n== one policy step of the rollout (k_substeps .. next k_substeps) ==")
lo, hi = pick(si, ["k_post_step", "k_act_sample" if any("k_act_sample" in n for n in names) else "k_mlp_fwd"])
print(timeline(lo, hi))
print(f"   step period: {(int(trace[hi]['Start_Timestamp']) - int(trace[lo]['Start_Timestamp'])) / 1e3:.1f} us")
