"""tools/probe_f16_convert.py -- which half <-> float conversion costs the fp16 stream its 3 % against bf16?  Interleaved
A/B of the production library against builds of kernels.hpp with -DWHVI_F16_UNPACK=0 (the compiler's form: SDWA operand
for the high half; production = 1: explicit shift), =2 (bf16's bit moves: WRONG values, timing only), =3 (shift form as
volatile asm), -DWHVI_F16_PACK_EXP=1 (bf16's pack: WRONG values, timing only), -DWHVI_BF16_PACK=0 (bf16: the compiler's
two converts + or per dword) -- whvi_amd/_exp/libexp_*.so, built with make ... DEFS=... as in whvi_amd/csrc/Makefile.  D = 4096, 2^20 rows (config 5); every timed launch starts
from the same finite data (an untimed copy from a master buffer), so the variants with wrong values cannot drift into
inf / NaN, which run faster (tools/probe_f16_data.py)."""
import ctypes
import glob
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

rows, log2d = 1 << 20, 12
sfx = sys.argv[1] if len(sys.argv) > 1 else "f16"       # f16 | bf16 (bf16: -DWHVI_BF16_PACK=0 builds = the compiler's pack)
master = (torch.randn(4096, 4096, device="cuda") * 2.0 ** -8).to(torch.float16 if sfx == "f16" else torch.bfloat16).repeat(rows // 4096, 1)
x = torch.empty_like(master)
libs = {"prod": _hip.lib()}
for path in sorted(glob.glob(os.path.join(os.path.dirname(_hip.LIB_PATH), "_exp", "libexp_*.so"))):
    L = ctypes.CDLL(path)
    getattr(L, "whvi_fwht_" + sfx).restype = ctypes.c_int
    getattr(L, "whvi_fwht_" + sfx).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p]
    libs[os.path.basename(path)[7:-3]] = L
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {k: [] for k in libs}
for rnd in range(9):
    for k, L in libs.items():
        x.copy_(master)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = getattr(L, "whvi_fwht_" + sfx)(x.data_ptr(), x.data_ptr(), rows, log2d, st)
        e.record()
        torch.cuda.synchronize()
        assert rc == 0
        if rnd >= 2:
            res[k].append(s.elapsed_time(e))
for k, v in res.items():
    v.sort()
    med = v[len(v) // 2]
    print(f"{sfx} {k:14s} median {med:.3f} ms  min {v[0]:.3f}  max {v[-1]:.3f}  -> {rows * 2 * 4096 * 2 / med / 1e9:.2f} TB/s", flush=True)
