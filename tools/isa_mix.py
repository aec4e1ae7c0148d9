"""Static instruction mix of a kernel from `hipcc -S` output (loops counted once).
    python tools/isa_mix.py /tmp/envk.s k_substeps
"""
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
cur, out = None, {}
for ln in txt:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1) if pat in m.group(1) else None
        if cur:
            out[cur] = []
        continue
    if cur is None:
        continue
    t = ln.strip()
    if t.startswith("s_endpgm"):
        cur = None
        continue
    if not t or t[0] in ".;" or t.endswith(":"):
        continue
    out[cur].append(t.split(";")[0].strip())
for name, ls in out.items():
    def n(f):
        return sum(1 for l in ls if f(l))
    print(name)
    print("  total", len(ls), "valu", n(lambda l: l.startswith("v_")), "dpp", n(lambda l: "quad_perm" in l or "row_" in l),
          "s_nop", n(lambda l: l.startswith("s_nop")), "nop_cycles", sum(int(l.split()[1]) + 1 for l in ls if l.startswith("s_nop")),
          "ds", n(lambda l: l.startswith("ds_")), "waitcnt", n(lambda l: l.startswith("s_waitcnt")),
          "cndmask", n(lambda l: l.startswith("v_cndmask")), "mov", n(lambda l: l.startswith(("v_mov", "v_accvgpr"))),
          "salu", n(lambda l: l.startswith("s_") and not l.startswith(("s_nop", "s_waitcnt"))),
          "trans", n(lambda l: l.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log"))))
