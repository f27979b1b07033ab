"""tools/profile_fused_inst.py [launches] -- the fused-kernel instantiations round 3 changed, one after the other, for rocprofv3
passes (`--kernel-trace --stats`, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_*`): 4 GiB in place, 64 MC samples, the
norm-preserving +/- D^-1/2 sign vectors of bench.py so the data stays O(1) over the launches.

    f64 D = 4096 shared a / c                 (round 2: one wave per SIMD, 2.9 TB/s)
    f32 D = 4096 / 2048 per-sample a / c, rows in (sample, batch, D) order  -> the block's sample's vectors staged in LDS
    f32 D = 4096 / 2048 per-sample a / c, rows in (batch, sample, D) order  -> three vectors from L2, split request schedule
    f64 D = 2048 per-sample a / c, both orders
Every launch's kernel symbol is printed (it is what the summaries are keyed by)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
S = 64
CASES = [(torch.float64, 4096, False, "batch"), (torch.float64, 4096, True, "sample"),
         (torch.float32, 4096, True, "sample"), (torch.float32, 2048, True, "sample"),
         (torch.float32, 4096, True, "batch"), (torch.float32, 2048, True, "batch"),
         (torch.float64, 2048, True, "sample"), (torch.float64, 2048, True, "batch")]
for dtype, d, per_sample, order in CASES:
    rows = (1 << 32) // (d * (4 if dtype == torch.float32 else 8))
    stride = 1 if order == "batch" else rows // S
    x = torch.randn(rows, d, device=dev, dtype=dtype)
    sgn = lambda k: ((torch.randint(0, 2, (k, d), device=dev) * 2 - 1).to(dtype) * d ** -0.5)   # noqa: E731
    g = torch.randn(S, d, device=dev, dtype=dtype)
    a, c = (sgn(S), sgn(S)) if per_sample else (sgn(1)[0], sgn(1)[0])
    for _ in range(n):
        _hip.fused_shs(x, a, g, c, axis="col", n_samples=S, sample_stride=stride, out=x, a_per_sample=per_sample,
                       c_per_sample=per_sample)
    torch.cuda.synchronize()
    print(f"{str(dtype)[6:]} D={d} per_sample={int(per_sample)} order={order}: {_hip.last_kernel()} finite={bool(torch.isfinite(x[::4099]).all())}",
          flush=True)
    del x
