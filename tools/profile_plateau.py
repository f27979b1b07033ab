"""tools/profile_plateau.py [pmc] -- why does the f32 stream at D <= 2048 sit ~2 % under the i32 instance of the same
shape and under f32 D = 4096?  All of these launches move the SAME tiles in the SAME order (16 KiB
per wave, one tile per wave, 4 GiB in place): only the instructions between the loads and the stores differ.
Interleaved HIP-event timing of
    f32 D=2048 / f32 D=4096 / i32 D=2048 / i32 D=4096   on random float bits, and on all-zero data
(data-dependent power -> clock is one candidate cause).  With the argument `pmc` it only makes a few launches of each
for a rocprofv3 --pmc pass (GRBM_GUI_ACTIVE gives the effective clock, SQ_* the issue picture)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

pmc = len(sys.argv) > 1 and sys.argv[1] == "pmc"
dev = torch.device("cuda", 0)
n = 1 << 30                                   # 4 GiB of 32-bit words
buf = torch.empty(n, dtype=torch.float32, device=dev)
cases = [("f32", 11), ("f32", 12), ("i32", 11), ("i32", 12), ("f32", 9), ("i32", 9)]


def view(kind, log2d):
    t = buf if kind == "f32" else buf.view(torch.int32)
    return t.view(-1, 1 << log2d)


def fill(mode):
    if mode == "zeros":
        buf.zero_()
    else:
        torch.manual_seed(0)
        buf.normal_()
        buf.mul_(2.0 ** -100)                 # 12 in-place transforms of D = 4096 stay far inside fp32's range


def timed(x, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _hip.fwht_rows(x, out=x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for mode in ("random", "zeros"):
    results = {c: [] for c in cases}
    for rnd in range(1 if pmc else 4):
        for kind, log2d in cases:
            fill(mode)                        # same starting bits for every case (i32 launches leave arbitrary float bits)
            x = view(kind, log2d)
            for _ in range(2):
                _hip.fwht_rows(x, out=x)
            results[(kind, log2d)].append(timed(x, 3 if pmc else 8))
    for (kind, log2d), ms in results.items():
        ms.sort()
        med = ms[len(ms) // 2]
        print(f"{mode:6s} {kind} D={1 << log2d:5d}: median {med:.4f} ms = {8 * n / med / 1e9:.3f} TB/s  (min {ms[0]:.4f})  "
              f"{_hip.last_kernel() if False else ''}", flush=True)
