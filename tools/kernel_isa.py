#!/usr/bin/env python3
"""tools/kernel_isa.py UNIT REGEX [--dump] -- resource usage (VGPRs, SGPRs, LDS, occupancy) and an instruction-class
histogram of the kernels of one translation unit whose demangled name matches REGEX (cross-compiles; no GPU).
    python tools/kernel_isa.py wbar_bwd_f32 'wbar_bwd_kernel<float, 11, 16, true, false, 2>'
Used for the ISA comparisons in DESIGN.md (VALU instructions per tile, s_nop counts, DPP / permlane / LDS mix)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "whvi_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
EXTRA = {"fused_f32": ["-fno-slp-vectorize"]}


def classify(op):
    if op.startswith("v_pk_"):
        return "VALU packed"
    if op.endswith("_dpp") or "dpp" in op:
        return "VALU dpp"
    if op.startswith("v_permlane"):
        return "VALU permlane"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "VALU mov"
    if op.startswith("v_"):
        return "VALU other"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "VMEM"
    return "other"


def main():
    unit, pattern = sys.argv[1], re.compile(sys.argv[2])
    dump = "--dump" in sys.argv
    defs = [a for a in sys.argv[3:] if a.startswith("-D")]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-fvisibility=hidden", "--cuda-device-only", *EXTRA.get(unit, []), *defs, "-S",
               os.path.join(CSRC, unit + ".hip"), "-o", asm]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode:
            sys.exit(res.stderr[-3000:])
        text = open(asm).read()
    demangle = lambda s: subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()  # noqa: E731
    # a kernel's code runs from its label to .Lfunc_endN; the resource comments ("; NumVgprs: ...") follow it
    lines = text.splitlines()
    idx = 0
    while idx < len(lines):
        m = re.match(r"^(_Z\w+):", lines[idx])
        idx += 1
        if not m:
            continue
        mangled, body = m.group(1), []
        while idx < len(lines) and not lines[idx].startswith(".Lfunc_end"):
            body.append(lines[idx])
            idx += 1
        usage = {}
        while idx < len(lines) and not re.match(r"^(_Z\w+):", lines[idx]):
            u = re.match(r";\s*(NumVgprs|NumAgprs|TotalNumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize):\s*(\d+)", lines[idx])
            if u:
                usage[u.group(1)] = int(u.group(2))
            idx += 1
        pretty = demangle(mangled)
        if not pattern.search(pretty):
            continue
        ops = [ln.split()[0] for ln in body if re.match(r"\s+[a-z_0-9]+(\s|$)", ln) and not ln.strip().startswith((";", "."))]
        hist = collections.Counter(classify(o) for o in ops)
        top = collections.Counter(ops).most_common(16)
        print(pretty[:160])
        print("  usage:", usage)
        print("  instructions:", len(ops), dict(sorted(hist.items())))
        print("  VALU total:", sum(v for k, v in hist.items() if k.startswith("VALU")))
        print("  top:", top)
        if dump:
            print("\n".join(body))


if __name__ == "__main__":
    main()
