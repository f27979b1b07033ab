"""tools/summarize_stream_profile.py TRACE_DIR FETCH_DIR WRITE_DIR BYTES_PER_LAUNCH OUT.csv -- one line per whvi:: kernel
from three rocprofv3 passes of the same script (--kernel-trace; --pmc FETCH_SIZE; --pmc WRITE_SIZE): launches, average
duration (all launches and the last 8: the clocks take a few launches to ramp), GB/s of algorithmic bytes, and the HBM
bytes per launch with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts the 128-byte requests of 16-byte-
per-lane streams as 64: doubled; WRITE_SIZE exact; both in KiB).  Behind profiles/r02/f16_config5_rocprof_summary.csv."""
import collections
import csv
import glob
import sys

trace_dir, fetch_dir, write_dir, alg_bytes, out_path = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]


def rows_of(d, pattern):
    f = glob.glob(d + "/**/*" + pattern, recursive=True)[0]
    return [r for r in csv.DictReader(open(f)) if "whvi::" in r["Kernel_Name"]]


def short(name):
    name = name[5:] if name.startswith("void ") else name
    return name[:name.index("(")] if "(" in name else name


dur = collections.defaultdict(list)
for r in rows_of(trace_dir, "kernel_trace.csv"):
    dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
counters = {}
for key, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
    acc = collections.defaultdict(list)
    for r in rows_of(d, "counter_collection.csv"):
        if r["Counter_Name"] == key:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    counters[key] = acc
with open(out_path, "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["kernel", "launches", "avg_ns_all", "avg_ns_last_8", "algorithmic_bytes_per_launch", "GB_per_s_last_8",
                "frac_of_8TBs", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg", "hbm_bytes(FETCHx2+WRITE)", "traffic_over_algorithmic"])
    for k, v in dur.items():
        last = v[-8:]
        avg_last = sum(last) / len(last)
        f = counters["FETCH_SIZE"].get(k, [])
        wr = counters["WRITE_SIZE"].get(k, [])
        fk = sum(f) / len(f) if f else float("nan")
        wk = sum(wr) / len(wr) if wr else float("nan")
        hbm = (2 * fk + wk) * 1024
        w.writerow([k, len(v), round(sum(v) / len(v)), round(avg_last), alg_bytes, round(alg_bytes / avg_last, 1),
                    round(alg_bytes / avg_last / 8000, 4), round(fk, 1), round(wk, 1), round(hbm), round(hbm / alg_bytes, 5)])
print(open(out_path).read())
