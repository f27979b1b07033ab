import sys; sys.path.insert(0, "/root/repo")
import torch
from whvi_amd import _hip
dev = torch.device("cuda", 0)
def timed(fn, iters=20, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for d, B in ((2048, 8192), (4096, 4096), (512, 32768)):
    S = 64
    x = torch.randn(B, d, device=dev)
    a = (torch.randint(0, 2, (d,), device=dev).float() * 2 - 1) * d ** -0.5
    c = (torch.randint(0, 2, (d,), device=dev).float() * 2 - 1) * d ** -0.5
    g = torch.randn(S, d, device=dev)
    out = torch.empty(S * B, d, device=dev)
    ms = timed(lambda: _hip.fused_shs(x, a, g, c, axis="col", n_samples=S, sample_stride=B, src_shared=True, out=out))
    k = _hip.last_kernel()
    t = _hip.fused_shs(x, None, c.reshape(1, -1), None, axis="col", n_samples=1, one_transform=True)
    ms1 = timed(lambda: _hip.fused_shs(t, a, g, None, axis="col", n_samples=S, sample_stride=B, src_shared=True, one_transform=True, out=out))
    k1 = _hip.last_kernel()
    ms0 = timed(lambda: _hip.fused_shs(x, None, c.reshape(1, -1), None, axis="col", n_samples=1, one_transform=True))
    print(f"D={d} B={B}: one transform per sample on the shared first half {ms1:.3f} ms = {out.numel() * 4 / 1e6 / ms1:7.1f} GB/s written "
          f"(+ the first half once: {ms0 * 1e3:.1f} us)   {k1[6:]}")
    xe = x.repeat(S, 1)
    ms2 = timed(lambda: _hip.fused_shs(xe, a, g, c, axis="col", n_samples=S, sample_stride=B, out=out))
    wr = out.numel() * 4 / 1e6
    print(f"D={d} B={B}: shared source {ms:.3f} ms = {wr/ms:7.1f} GB/s written | expanded out of place {ms2:.3f} ms = {2*wr/ms2:7.1f} GB/s r+w   {k[6:]}")
