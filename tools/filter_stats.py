"""tools/filter_stats.py IN.csv OUT.csv -- keep our kernels (whvi::) and anything above 1 % from a rocprofv3 kernel_stats.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "whvi::" in r["Name"] or float(r["Percentage"]) >= 1.0]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    w.writerows(keep)
print(f"{len(keep)} of {len(rows)} rows kept")
