"""tools/update_hbm_traffic.py KEY FETCH_DIR WRITE_DIR [KERNEL_SUBSTRING [ALGORITHMIC_BYTES]] -- turn two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; tools/pmc_cmd.sh) of `bench.py --no-extras --no-cpu-baseline` into the record bench.py reports
as `roofline.traffic`: profiles/hbm_traffic.json[KEY], stamped with the kernel symbol and the hash of the kernel
sources (bench.kernel_source_hash) so a later build cannot silently inherit it.

Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE counts the 128-B requests of
16-B/lane coalesced streams as 64 B -> doubled; WRITE_SIZE is exact; both are in KiB."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def counter_rows(directory, counter, needle):
    f = glob.glob(directory + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and needle in r["Kernel_Name"]]
    if not rows:
        sys.exit(f"no {counter} rows for {needle!r} in {f}")
    return rows


def main():
    key, fetch_dir, write_dir = sys.argv[1:4]
    needle = sys.argv[4] if len(sys.argv) > 4 else "fwht_rows_kernel<float, 12"
    fetch = counter_rows(fetch_dir, "FETCH_SIZE", needle)
    write = counter_rows(write_dir, "WRITE_SIZE", needle)
    # one instantiation only: the launches of the most frequent symbol (a script may end with a few launches of another form)
    names = [r["Kernel_Name"] for r in fetch]
    name = max(set(names), key=names.count)
    fetch = [r for r in fetch if r["Kernel_Name"] == name]
    write = [r for r in write if r["Kernel_Name"] == name]
    symbol = "whvi::" + re.search(r"whvi::(\w+<[^>]*>)", name).group(1)
    f_kib = sum(float(r["Counter_Value"]) for r in fetch) / len(fetch)
    w_kib = sum(float(r["Counter_Value"]) for r in write) / len(write)
    m = re.match(r"fwht_f32_D(\d+)_rows(\d+)", key)
    alg = int(sys.argv[5]) if len(sys.argv) > 5 else (2 * int(m.group(1)) * 4 * int(m.group(2)) if m else None)
    rec = {"hbm_bytes_per_launch": int(round((2 * f_kib + w_kib) * 1024)),
           "read_bytes_corrected": int(round(2 * f_kib * 1024)), "write_bytes": int(round(w_kib * 1024)),
           "FETCH_SIZE_raw_KB": f_kib, "WRITE_SIZE_raw_KB": w_kib, "launches_averaged": [len(fetch), len(write)],
           "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane coalesced streams -> doubled "
                         "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact for 16-B/lane stores; units are KiB",
           "algorithmic_bytes_per_launch": alg, "kernel_symbol": symbol, "source_sha256": bench.kernel_source_hash(),
           "grid": fetch[0]["Grid_Size"], "workgroup": fetch[0]["Workgroup_Size"], "vgprs": fetch[0]["VGPR_Count"],
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes ({os.path.basename(fetch_dir)}, "
                     f"{os.path.basename(write_dir)})"}
    if alg:
        rec["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / alg
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    data = json.load(open(path))
    data[key] = rec
    json.dump(data, open(path, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
