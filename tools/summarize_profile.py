"""tools/summarize_profile.py OUT.csv ALG_BYTES TRACE_DIR [FETCH_DIR WRITE_DIR [SQ_DIR]] -- one line per whvi:: kernel from the
rocprofv3 passes of ONE script: `--kernel-trace` (launches, average / median / min duration, average of the last 8: the
clocks take a few launches to ramp), `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (HBM bytes per launch with the gfx950
correction of MI355X_MICROARCH.md: FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane streams as 64 -> doubled;
WRITE_SIZE exact; both in KiB) and optionally SQ counters (VALU / LDS instructions per wave, wave lifetime, share of it
waiting).  ALG_BYTES: algorithmic bytes per launch (one number for every kernel of the script, or 0 = leave the rate
columns empty)."""
import collections
import csv
import glob
import statistics
import sys

out_path, alg = sys.argv[1], int(sys.argv[2])
dirs = sys.argv[3:]


def rows_of(d, pattern):
    f = glob.glob(d + "/**/*" + pattern, recursive=True)
    return [r for r in csv.DictReader(open(f[0])) if "whvi::" in r["Kernel_Name"]] if f else []


def short(name):
    name = name[5:] if name.startswith("void ") else name
    return name[:name.index("(")] if "(" in name else name


dur = collections.OrderedDict()
for r in rows_of(dirs[0], "kernel_trace.csv"):
    dur.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for d in dirs[1:]:
    for r in rows_of(d, "counter_collection.csv"):
        pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[short(r["Kernel_Name"])] = (r["Grid_Size"], r["Workgroup_Size"], r.get("LDS_Block_Size", ""))
avg = lambda v: sum(v) / len(v) if v else float("nan")   # noqa: E731
with open(out_path, "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["kernel", "launches", "avg_ns_all", "median_ns", "min_ns", "avg_ns_last_8", "algorithmic_bytes_per_launch",
                "GB_per_s_last_8", "frac_of_8TBs_last_8", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg", "hbm_bytes(FETCHx2+WRITE)",
                "traffic_over_algorithmic", "grid_threads", "workgroup", "VALU_insts_per_wave", "LDS_insts_per_wave",
                "wave_quad_cycles_per_wave", "share_of_wave_cycles_waiting(SQ_WAIT_ANY)"])
    for k, v in dur.items():
        last = v[-8:]
        c = pmc.get(k, {})
        fk, wk = avg(c.get("FETCH_SIZE", [])), avg(c.get("WRITE_SIZE", []))
        hbm = (2 * fk + wk) * 1024
        waves = avg(c.get("SQ_WAVES", []))
        row = [k, len(v), round(avg(v)), round(statistics.median(v)), min(v), round(avg(last)), alg or "",
               round(alg / avg(last), 1) if alg else "", round(alg / avg(last) / 8000, 4) if alg else "",
               round(fk, 1), round(wk, 1), round(hbm) if hbm == hbm else "", round(hbm / alg, 5) if alg and hbm == hbm else "",
               *meta.get(k, ("", ""))[:2]]
        if waves == waves and waves > 0:
            row += [round(avg(c.get("SQ_INSTS_VALU", [])) / waves, 1), round(avg(c.get("SQ_INSTS_LDS", [0])) / waves, 1),
                    round(avg(c.get("SQ_WAVE_CYCLES", [])) / waves, 1),
                    round(avg(c.get("SQ_WAIT_ANY", [])) / avg(c.get("SQ_WAVE_CYCLES", [1])), 3)]
        w.writerow(row)
print(open(out_path).read())
