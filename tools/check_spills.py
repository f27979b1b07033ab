#!/usr/bin/env python3
"""tools/check_spills.py -- build-time guard: compile every kernel translation unit with
-Rpass-analysis=kernel-resource-usage and fail if any shipped kernel uses scratch (a spill turns a
bandwidth-bound kernel into a scratch-bound one without any functional symptom).
Usage: python tools/check_spills.py [-j N]      (cross-compiles; no GPU needed)"""
import concurrent.futures as cf
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "whvi_amd", "csrc")
UNITS = ["fwht_f32", "fwht_f64", "fwht_f16", "fwht_bf16", "fwht_i32", "fused_f32", "fused_f64", "wbar_bwd_f32",
         "wbar_bwd_f64", "wbar_fwd_f32", "wbar_fwd_f64", "train_aux", "abi", "fwht_wide", "diag_apply", "stream_probe", "layer_apply"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]


EXTRA = {"fused_f32": ["-fno-slp-vectorize"], "fwht_wide": ["-fno-slp-vectorize"]}      # per-unit flags, as in whvi_amd/csrc/Makefile


def scan(unit):
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *EXTRA.get(unit, []), os.path.join(CSRC, unit + ".hip")],
                         capture_output=True, text=True).stderr
    bad, name, n = [], None, 0
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name, n = m.group(1), n + 1
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and int(m.group(1)) > 0:
            bad.append((unit, name, int(m.group(1))))
    return n, bad


def main():
    jobs = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else min(8, os.cpu_count() or 1)
    total, bad = 0, []
    with cf.ThreadPoolExecutor(jobs) as ex:
        for n, b in ex.map(scan, UNITS):
            total += n
            bad += b
    for unit, name, sz in bad:
        print(f"SPILL {unit}: {name} scratch={sz} B/lane")
    print(f"check_spills: {total} kernels, {len(bad)} with scratch")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
