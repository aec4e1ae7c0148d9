"""Per-kernel effective clock and matrix-core / LDS busy fractions from rocprofv3 --pmc passes of bench.py (gpurun_out/<tag>_pmc_clock*):
  effective clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X_MICROARCH.md, DVFS give-back; reads high on short dispatches),
  MFMA busy       = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES,  LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
    python tools/digest_clock_pmc.py r03  ->  profiles/r03_kernel_clocks.txt"""
import collections, csv, glob, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", f"{tag}_pmc_clock*"))):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:64]
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (d, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[(k, d)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
out = [f"# {tag}: per-kernel means over the dispatches of `bench.py --steps 1 --warmup 1` under rocprofv3 --pmc (kernels run one at a time there)",
       f"{'kernel':66s} {'n':>5s} {'us':>8s} {'eff. clock GHz':>15s} {'MFMA busy':>10s} {'LDS conflict / active':>22s}"]
mean = lambda v: sum(v) / max(len(v), 1)
for k, c in sorted(rows.items(), key=lambda kv: -sum(sum(v) for (kk, _), v in dur.items() if kk == kv[0])):
    ds = [x for (kk, d), v in dur.items() if kk == k and "GRBM_GUI_ACTIVE" in c and d.endswith("clock1") for x in v] or \
         [x for (kk, _), v in dur.items() if kk == k for x in v]
    us = mean(ds)
    if us < 4.0:
        continue
    clk = mean(c["GRBM_GUI_ACTIVE"]) / 8.0 / (us * 1e-6) / 1e9 if "GRBM_GUI_ACTIVE" in c else float("nan")
    mf = mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / max(mean(c["SQ_BUSY_CU_CYCLES"]), 1.0) if "SQ_VALU_MFMA_BUSY_CYCLES" in c else float("nan")
    lc = mean(c["SQ_LDS_BANK_CONFLICT"]) / max(mean(c["SQ_LDS_IDX_ACTIVE"]), 1.0) if "SQ_LDS_BANK_CONFLICT" in c else float("nan")
    out.append(f"{k:66s} {len(ds):5d} {us:8.1f} {clk:15.2f} {mf:10.3f} {lc:22.3f}")
open(os.path.join(root, "profiles", f"{tag}_kernel_clocks.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
