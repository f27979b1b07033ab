"""Hardware probe (not a pytest file): is the FWHT kernel limited by its memory access structure or by
its compute?  Compares, on one resident buffer, torch's in-place elementwise kernel, the FWHT kernel
with LOG2D = 0 (same loads/stores, no butterflies) and the real transform, over variants."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import torch
from whvi_amd import _hip


def bench(fn, iters=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    gib = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    n = gib * (1 << 30) // 4
    x = torch.randn(n, device="cuda") * 2.0 ** -100
    tb = lambda ms: 2 * n * 4 / (ms * 1e-3) / 1e12
    ms = bench(lambda: x.mul_(1.0)); print(f"{gib} GiB  torch mul_ in place : {ms:.3f} ms {tb(ms):.2f} TB/s", flush=True)
    ms = bench(lambda: x.add_(0.0)); print(f"{gib} GiB  torch add_ in place : {ms:.3f} ms {tb(ms):.2f} TB/s", flush=True)
    for log2d in (12, 11, 9):
        v2 = x.view(-1, 1 << log2d)
        for var, blks in ((6, (2,)), (3, (0, 1, 2)), (7, (0, 1, 2))):
            for blk in blks:
                for bpc in (0,):
                    v = var | (blk << 4) | (bpc << 8)
                    ms = bench(lambda: _hip.fwht_rows(v2, out=v2, variant=v))
                    print(f"   log2d={log2d:2d} variant {var} block {256 << blk:4d} bpc {bpc:4d}: {ms:.3f} ms {tb(ms):.2f} TB/s", flush=True)
        x.mul_(0).add_(1e-30)


if __name__ == "__main__":
    main()
