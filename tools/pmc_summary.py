"""tools/pmc_summary.py DIR OUT.csv -- rows of our kernels (whvi::) from a rocprofv3 --pmc counter_collection.csv,
reduced to kernel, grid, block, VGPRs, counter, value."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "whvi::" in r["Kernel_Name"]]
with open(sys.argv[2], "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"])
    for r in rows:
        w.writerow([r["Kernel_Name"][:120], r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"],
                    r["Counter_Name"], r["Counter_Value"]])
print(len(rows), "rows")
