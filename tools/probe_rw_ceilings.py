"""tools/probe_rw_ceilings.py -- what torch's own streaming kernels reach on this GPU over 4 GiB: write-only (fill_, zero_)
and read-only (sum).  The yardsticks quoted beside the write-dominated launches in DESIGN.md section 5.3 / 6.2b."""
import torch

x = torch.empty(1 << 30, device="cuda")
x.normal_()


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


gb = x.numel() * 4 / 1e6
print("fill_ (write-only) GB/s:", round(gb / timed(lambda: x.fill_(1.5)), 1))
print("zero_ (memset)     GB/s:", round(gb / timed(lambda: x.zero_()), 1))
print("sum   (read-only)  GB/s:", round(gb / timed(lambda: x.sum()), 1))
