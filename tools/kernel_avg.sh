# average duration of kernels matching $1 under env settings:  tools/kernel_avg.sh k_head_net "LG_X=1" "LG_X=2" ...
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do
  rm -rf /tmp/kavg
  env $kv rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kavg -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_alt --no_other --sustained 0 > /dev/null 2>&1
  python3 - "$pat" "$kv" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/kavg/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[1] in r["Name"]:
        print(sys.argv[2], r["Name"][:60], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
PY
done
