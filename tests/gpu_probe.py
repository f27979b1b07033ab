"""Quick hardware probe (not a pytest file): variant sweep of the FWHT kernel with HIP events.
Usage on the GPU box:  python tests/gpu_probe.py [log2rows]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip


def bench(fn, iters=10, warm=2):
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
    log2rows = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count, "CUs")
    for log2d in (12, 11, 10, 9):
        d = 1 << log2d
        rows = 1 << (log2rows + 12 - log2d)
        x = torch.randn(rows, d, device="cuda")
        gb = 2 * x.numel() * 4 / 1e9
        ms = bench(lambda: x.mul_(1.0))   # in-place elementwise = same algorithmic bytes
        print(f"D={d} rows={rows} ({gb / 2:.2f} GB)  torch in-place mul: {ms:.3f} ms {gb / ms:.1f} GB/s... wait", flush=True)
        y = torch.empty_like(x)
        ms = bench(lambda: y.copy_(x))
        print(f"   torch copy: {ms:.3f} ms {gb / ms * 1e0:.1f} GB/ms -> {gb / (ms / 1e3) / 1e3:.2f} TB/s", flush=True)
        for var in (0, 1, 2, 4, 6):
            for bpc in (2, 3, 4, 8, 64):
                v = var | (bpc << 8)
                try:
                    ms = bench(lambda: _hip.fwht_rows(x, out=x, variant=v))
                except RuntimeError as err:
                    print("   variant", var, "bpc", bpc, "ERR", err)
                    continue
                print(f"   variant {var} bpc {bpc:3d}: {ms:.3f} ms  {gb / (ms / 1e3) / 1e3:.2f} TB/s  "
                      f"{rows / (ms / 1e3) / 1e9:.4f} Gtransforms/s", flush=True)
        del x, y


if __name__ == "__main__":
    main()
