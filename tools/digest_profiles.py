"""Turn the rocprofv3 outputs of a round (gpurun_out/rNN_prof, rNN_pmc_fetch, rNN_pmc_write) into the
committed summaries under profiles/: kernel-stats CSV + markdown table, per-kernel HBM traffic
(FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for 16-B/lane streams, WRITE_SIZE as read;
both in KiB) and rNN_pmc_traffic.json (what bench.py reports as roofline.traffic).

    python tools/digest_profiles.py r01
"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"          # "r03" or "r03_anymal_c_rough" (rocprofv3 passes of bench.py --task ...)
task = tag.split("_", 1)[1] if "_" in tag else "anymal_c_flat"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def one(pattern):
    f = glob.glob(os.path.join(go, pattern))
    if not f:
        raise SystemExit(f"missing {pattern}")
    return max(f, key=os.path.getmtime)            # gpurun merges into gpurun_out/: an earlier collection's files may still lie there


stats = list(csv.DictReader(open(one(f"{tag}_prof/*/*kernel_stats.csv"))))
shutil.copy(one(f"{tag}_prof/*/*kernel_stats.csv"), os.path.join(pr, f"{tag}_bench_kernel_stats.csv"))
tot = sum(float(r["TotalDurationNs"]) for r in stats)


def pmc(name, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(one(f"{tag}_pmc_{name}/*/*counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
lines = ["| kernel | calls | total ms | avg us | % | HBM read KiB/launch (2 x FETCH_SIZE) | HBM write KiB/launch (WRITE_SIZE) |",
         "|---|---|---|---|---|---|---|"]
for r in stats[:24]:
    n = r["Name"]
    f = 2.0 * sum(fetch[n]) / len(fetch[n]) if n in fetch else float("nan")
    w = sum(write[n]) / len(write[n]) if n in write else float("nan")
    lines.append(f"| `{n[:78]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
                 f"{100 * float(r['TotalDurationNs']) / tot:.2f} | {f:.0f} | {w:.0f} |")

# traffic of one minibatch GEMM group / one env step, from the per-launch means x launches per unit
trace = list(csv.DictReader(open(one(f"{tag}_prof/*/*kernel_trace.csv"))))
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in trace]
def window(starts, must):
    """the last window [starts[k], starts[k+1]) holding every kernel of `must` (bench.py also launches bare timing loops)"""
    for k in range(len(starts) - 2, -1, -1):
        w = names[starts[k]:starts[k + 1]]
        if all(any(m in n for n in w) for m in must):
            return starts[k], starts[k + 1]
    raise SystemExit(f"no window with {must}")


# one minibatch of the update = optimiser step .. next optimiser step (the gather of the next minibatch rides in k_opt_prepare);
# the GEMM group bench.py times = gather + forward + head + backward of one lg_ppo_minibatch_backward
gi = window([i for i, n in enumerate(names) if n.startswith("k_opt_adam")], ["k_opt_prepare", "k_gemm_dw"])
mb = names[gi[0]:gi[1]]
si = window([i for i, n in enumerate(names) if "k_substeps" in n], ["k_post_step", "k_mlp_fwd"])
st = names[si[0]:si[1]]                         # one policy step: substeps .. act of the next
kib = lambda n: (2.0 * sum(fetch[n]) / len(fetch[n]) if n in fetch else 0.0) + (sum(write[n]) / len(write[n]) if n in write else 0.0)
group = [n for n in mb if "k_gemm" in n or n.startswith("k_loss") or "k_head" in n]      # what lg_ppo_minibatch_backward launches inside an update
envk = [n for n in st if "k_substeps" in n or "k_post_step" in n or "k_finalize" in n]
out = {"policy_hidden": [512, 256, 128], "task": task,
       "gemm_group_bytes_per_minibatch": round(1024.0 * sum(kib(n) for n in group)),
       "gemm_group_launches": len(group),
       "env_step_bytes_per_call": round(1024.0 * sum(kib(n) for n in envk)),
       "env_step_launches": len(envk),
       "note": "HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate --pmc passes, per-launch means x launches of one unit"}
json.dump(out, open(os.path.join(pr, f"{tag}_pmc_traffic.json"), "w"), indent=1)
open(os.path.join(pr, f"{tag}_kernel_table.md"), "w").write("\n".join(lines) + "\n")
# one minibatch / one env step timelines from the trace
def timeline(lo, hi):
    t0 = int(trace[lo]["Start_Timestamp"])
    rows = []
    for r in trace[lo:hi]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        rows.append(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:70]}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
    return "\n".join(rows)
open(os.path.join(pr, f"{tag}_timelines.txt"), "w").write(
    "== one PPO minibatch of the update (optimiser step .. next optimiser step), profiled ==\n" + timeline(gi[0], gi[1]) +
    "\n\n== one policy step of the rollout (k_substeps .. next k_substeps), profiled ==\n" + timeline(si[0], si[1]) + "\n")
for f in (f"{tag}_gpu_tests.log", f"{tag}_bench_n1.json"):
    if os.path.exists(os.path.join(go, f)):
        shutil.copy(os.path.join(go, f), os.path.join(pr, f))
print("\n".join(lines))
print(json.dumps(out))
