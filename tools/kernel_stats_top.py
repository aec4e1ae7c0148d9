import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:8.1f} pct {100*float(r["TotalDurationNs"])/tot:5.1f}')
