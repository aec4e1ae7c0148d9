"""profiles/<tag>_substeps_pmc.json from the SQ counter pass over tools/physics_prof.py (10 x lg_simulate, 10 x lg_compute_torques,
10 x lg_step on 4096 flat ANYmal-C envs): instruction counts of the control-loop kernel's critical wave.

Counters are sums over the 1024 waves of a launch (256 workgroups x 4 waves).  In the physics-only launches the 512 physics waves
(waves 0 and 1 of a workgroup: pair-lane physics, lg_physics_pair.h) execute nearly all VALU work (the other waves only stage and wait
at barriers); in the torque-only launches all 1024 waves share it.
"""
import collections, csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = max(glob.glob(os.path.join(root, "gpurun_out", f"{tag}_pmc_phys", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if "k_substeps" in r["Kernel_Name"]:
        agg[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(agg)
steps, sims, torqs = ids[-10:], ids[-30:-20], ids[-20:-10]          # launch order of tools/physics_prof.py
mean = lambda sel, c: sum(agg[i][c] for i in sel) / len(sel)
W, PW = 1024.0, 512.0
out = {
    "workload": "anymal_c_flat, 4096 envs: k_substeps<4,3,true,true>, 256 workgroups x 4 waves (2 physics waves each)",
    "physics_only": {"valu_per_physics_wave": round(mean(sims, "SQ_INSTS_VALU") / PW), "salu_per_physics_wave": round(mean(sims, "SQ_INSTS_SALU") / PW),
                     "lds_per_physics_wave": round(mean(sims, "SQ_INSTS_LDS") / PW), "wave_quad_cycles": round(mean(sims, "SQ_WAVE_CYCLES") / W),
                     "active_quad_cycles_per_physics_wave": round(mean(sims, "SQ_ACTIVE_INST_ANY") / PW)},
    "torques_only": {"valu_per_wave": round(mean(torqs, "SQ_INSTS_VALU") / W), "lds_per_wave": round(mean(torqs, "SQ_INSTS_LDS") / W),
                     "wave_quad_cycles": round(mean(torqs, "SQ_WAVE_CYCLES") / W)},
    "lg_step": {"valu_total": round(mean(steps, "SQ_INSTS_VALU")), "wave_quad_cycles": round(mean(steps, "SQ_WAVE_CYCLES") / W),
                "wait_any_quad_cycles": round(mean(steps, "SQ_WAIT_ANY") / W), "active_quad_cycles": round(mean(steps, "SQ_ACTIVE_INST_ANY") / W)},
}
dec = 4
out["critical_wave_valu_per_lg_step"] = dec * (out["physics_only"]["valu_per_physics_wave"] + out["torques_only"]["valu_per_wave"])
out["note"] = ("critical wave = a physics wave of a workgroup (wave 0 or 1): decimation x (one half-spatial-vector share of the physics of 8 envs + its quarter of the actuator-net rows); a wave64 "
               "issues at most one VALU instruction per 4 cycles on its SIMD, so T >= 4 x critical_wave_valu / clock")
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_substeps_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
