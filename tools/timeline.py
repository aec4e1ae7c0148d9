"""Timelines of one PPO minibatch and one rollout step out of a rocprofv3 kernel trace (csv):
    python tools/timeline.py gpurun_out/<dir>"""
import csv, glob, os, sys

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
trace = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in trace]


def timeline(lo, hi):
    t0 = int(trace[lo]["Start_Timestamp"])
    out = []
    for r in trace[lo:hi]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        out.append(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:86]}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
    return "\n".join(out)


gi = [i for i, n in enumerate(names) if n.startswith("k_gather")]
si = [i for i, n in enumerate(names) if "k_substeps" in n]
k = -6 if len(gi) > 8 else -2
print("== one PPO minibatch (gather .. next gather) ==")
print(timeline(gi[k], gi[k + 1]))
print(f"   minibatch period: {(int(trace[gi[k + 1]]['Start_Timestamp']) - int(trace[gi[k]]['Start_Timestamp'])) / 1e3:.1f} us")
print("\n== one policy step of the rollout (k_substeps .. next k_substeps) ==")
k = -6 if len(si) > 8 else -2
print(timeline(si[k], si[k + 1]))
print(f"   step period: {(int(trace[si[k + 1]]['Start_Timestamp']) - int(trace[si[k]]['Start_Timestamp'])) / 1e3:.1f} us")
