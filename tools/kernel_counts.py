"""tools/kernel_counts.py STATS_DIR STEPS -- launches per step and time per kernel from a rocprofv3 --stats run."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["Calls"]) for r in rows)
tns = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"{tot} launches, {tot / steps:.1f} per step, {tns / steps / 1e3:.1f} us of kernel time per step")
for r in rows[:45]:
    print(f'{int(r["Calls"]) / steps:7.2f}/step {float(r["AverageNs"]) / 1e3:8.2f} us  {r["Name"][:120]}')
