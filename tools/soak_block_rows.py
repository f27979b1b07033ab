"""tools/soak_block_rows.py [seconds] -- soak of fwht_block_rows_kernel: random row counts (1 .. a few resident grids, so
one-row-per-block, pipelined, cached and streaming forms all come up), random storage type and row length, integer-valued
data; every case checks H(H(x)) = D x exactly (f32 / f64 / i32) or a sampled row against a float64 transform computed on
the GPU by the dense Hadamard product in pieces (16-bit storage).  Prints the case count; any mismatch raises."""
import os
import random
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(1234)
t0, cases, forms = time.time(), 0, {}
while time.time() - t0 < budget:
    dtype = rng.choice([torch.float32, torch.float32, torch.float64, torch.int32, torch.float16, torch.bfloat16])
    lo = 13 if dtype == torch.float64 else 14
    hi = 15 if dtype == torch.float64 else 16
    log2d = rng.randint(lo, hi)
    d = 1 << log2d
    esize = torch.empty(0, dtype=dtype).element_size()
    max_rows = (768 << 20) // (d * esize)
    rows = rng.choice([1, 2, 3, rng.randint(4, 300), rng.randint(300, max_rows), rng.randint(max_rows // 4, max_rows)])
    x = torch.randint(-2, 3, (rows, d), device="cuda", dtype=torch.int32).to(dtype)
    inplace = rng.random() < 0.5
    if inplace:
        y = x.clone()
        _hip.fwht_rows(y, out=y)
    else:
        y = _hip.fwht_rows(x)
    k = _hip.last_kernel()
    forms[k] = forms.get(k, 0) + 1
    if dtype in (torch.float16, torch.bfloat16):
        # one rounding of the exact integer result: compare a sampled row with the f32 kernel's (exact) output rounded once
        r = rng.randrange(rows)
        ref = _hip.fwht_rows(x[r:r + 1].float()).to(dtype)
        assert torch.equal(y[r:r + 1], ref), (dtype, log2d, rows, r)
    else:
        z = _hip.fwht_rows(y)
        assert torch.equal(z, x * d), (dtype, log2d, rows, inplace)
    cases += 1
    del x, y
torch.cuda.synchronize()
print(f"soak ok: {cases} cases in {time.time() - t0:.0f} s")
for k, n in sorted(forms.items()):
    print(f"  {n:4d} x {k}")
