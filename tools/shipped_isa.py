#!/usr/bin/env python3
"""tools/shipped_isa.py [REGEX] -- what is INSIDE a built libwhvi_hip.so: per kernel, the register / scratch / LDS figures
of the gfx950 code objects' metadata, and (for kernels whose demangled name matches REGEX) the disassembly with the issue
pattern of its 16-byte stores.  Needs no GPU and no recompilation: it reads the shipped binary (llvm-objdump --offloading,
llvm-readelf --notes, llvm-objdump -d).  tests/test_build.py uses it to pin the occupancy budgets and the store issue forms
the streaming kernels were tuned for (9 % of the headline rate hangs on how 16 stores are issued).

    python tools/shipped_isa.py 'fwht_rows_kernel<float, 12, 16, 0, false, true, 256, 1, false>'
    python tools/shipped_isa.py --lib whvi_amd/_exp/libwhvi_hip_x.so 'wbar_fwd_kernel<float, 11'"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT_LIB = os.path.join(ROOT, "whvi_amd", "libwhvi_hip.so")


class ShippedLibrary:
    """Kernels of one built library.  ``kernels``: demangled name -> dict(mangled, vgprs, agprs, sgprs, scratch, lds, code_object)."""

    def __init__(self, lib=DEFAULT_LIB):
        self.tmp = tempfile.mkdtemp(prefix="whvi_isa_")
        copy = os.path.join(self.tmp, "lib.so")
        shutil.copy(lib, copy)                      # the extractor writes its bundles next to the file it reads
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", copy], check=True, capture_output=True, cwd=self.tmp)
        self.kernels = {}
        mangled_of = {}
        for name in sorted(os.listdir(self.tmp)):
            if "gfx950" not in name:
                continue
            path = os.path.join(self.tmp, name)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", path], check=True, capture_output=True, text=True).stdout
            # one YAML block per kernel under amdhsa.kernels: take the scalar fields we need from each "- .args:" item
            for block in re.split(r"\n  - \.", notes):
                m = re.search(r"\.name:\s+(\S+)", block)
                v = re.search(r"\.vgpr_count:\s+(\d+)", block)
                if not m or not v:
                    continue
                field = lambda key, d=0: int((re.search(r"\.%s:\s+(\d+)" % key, block) or [None, d])[1])   # noqa: E731
                mangled_of[m.group(1)] = dict(mangled=m.group(1), vgprs=int(v.group(1)), agprs=field("agpr_count"),
                                              sgprs=field("sgpr_count"), scratch=field("private_segment_fixed_size"),
                                              lds=field("group_segment_fixed_size"), code_object=path)
        names = list(mangled_of)
        pretty = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        for mangled, p in zip(names, pretty):
            p = re.sub(r"\(.*$", "", p.strip())     # "void whvi::k<...>(args)" -> "whvi::k<...>"
            p = re.sub(r"^void ", "", p)
            self.kernels[p] = mangled_of[mangled]

    def close(self):
        shutil.rmtree(self.tmp, ignore_errors=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def find(self, name):
        """The kernel whose demangled name is exactly ``name`` (KeyError with near misses otherwise)."""
        if name in self.kernels:
            return self.kernels[name]
        family = name.split("<")[0]
        near = [k for k in self.kernels if k.startswith(family + "<")][:8]
        raise KeyError(f"{name} is not in the shipped library; same family: {near}")

    def ops(self, name):
        """[(mnemonic, operand text)] of the kernel's instructions, in program order."""
        k = self.find(name)
        out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f"--disassemble-symbols={k['mangled']}",
                              k["code_object"]], check=True, capture_output=True, text=True).stdout
        ops = []
        for line in out.splitlines():
            m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*(//.*)?$", line)
            if m and not line.strip().startswith(("//", ";")):
                ops.append((m.group(1), m.group(2)))
        return ops


def store_runs(ops, mnemonic):
    """Lengths of the maximal runs of consecutive ``mnemonic`` instructions (16-byte stores issued back to back)."""
    runs, cur = [], 0
    for op, _ in ops:
        if op == mnemonic:
            cur += 1
        else:
            if cur:
                runs.append(cur)
            cur = 0
    if cur:
        runs.append(cur)
    return runs


def main():
    args = [a for a in sys.argv[1:]]
    lib = DEFAULT_LIB
    if "--lib" in args:
        i = args.index("--lib")
        lib = args[i + 1]
        del args[i:i + 2]
    pattern = re.compile(args[0]) if args else None
    with ShippedLibrary(lib) as shipped:
        print(f"{len(shipped.kernels)} kernels in {lib}; with scratch: {sum(1 for k in shipped.kernels.values() if k['scratch'])}")
        for name, k in sorted(shipped.kernels.items()):
            if pattern is None or not pattern.search(name):
                continue
            ops = shipped.ops(name)
            print(name)
            print("   ", {f: k[f] for f in ("vgprs", "agprs", "sgprs", "scratch", "lds")}, f"{len(ops)} instructions")
            for mnemonic in ("buffer_store_dwordx4", "global_store_dwordx4"):
                runs = store_runs(ops, mnemonic)
                if runs:
                    print(f"    {mnemonic}: {sum(runs)} stores in runs of {runs}")


if __name__ == "__main__":
    main()
