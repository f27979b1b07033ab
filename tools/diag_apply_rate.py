"""tools/diag_apply_rate.py -- HIP-event rates of whvi_diag_apply / whvi_diag_apply_bwd against the matrix route
(weight construction + GEMM) on the layer shapes of BASELINE configs 2 and 4.  GB/s = algorithmic bytes (x read unless shared,
out written; backward: g + x read, grad_x written) / time."""
import sys; sys.path.insert(0, "/root/repo")
import torch
from whvi_amd import _hip
from whvi_amd.weights import WBarFunction

dev = torch.device("cuda", 0)


def timed(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


shapes = [(512, 32, 4096, True), (512, 32, 4096, False), (512, 16, 4096, True), (512, 64, 4096, True), (1024, 16, 45730, False), (1024, 16, 8192, True),
          (2048, 16, 8192, False), (4096, 16, 4096, False), (256, 32, 16384, False), (64, 32, 65536, False)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")[:3]) + (a.split(",")[3] == "1",) for a in sys.argv[1:]]
for D, S, B, shared in shapes:
    g = torch.Generator(device=dev).manual_seed(D)
    s1, s2 = torch.randn(D, device=dev, generator=g), torch.randn(D, device=dev, generator=g)
    u = torch.randn(1 + S, D, device=dev, generator=g)
    bias = torch.randn(D, device=dev, generator=g)
    x = torch.randn((B, D) if shared else (S, B, D), device=dev, generator=g)
    out = torch.empty(S, B, D, device=dev)
    ms = timed(lambda: _hip.diag_apply(x, s1, s2, u, bias, n_samples=S, out=out))
    k = _hip.last_kernel()
    nbytes = out.numel() * 4 + (0 if shared else x.numel() * 4)
    line = f"D={D} S={S} B={B} shared={int(shared)}: fwd {ms*1e3:8.1f} us {nbytes/ms/1e6:7.0f} GB/s"
    for name, tune in (("nt", 16), ("cached", 32), ("cached big tiles", 32 | 128), ("nt+plain", 16 | 64), ("cached+plain", 32 | 64)):
        if not shared and "plain" in name:
            continue
        ms = timed(lambda: _hip.diag_apply(x, s1, s2, u, bias, n_samples=S, out=out, tune=tune))
        line += f" [{name} {ms*1e3:.1f}]"
    gout = torch.randn(S, B, D, device=dev, generator=g)
    for need in (True, False):
        ms = timed(lambda: _hip.diag_apply_bwd(gout, x, s1, s2, u, n_samples=S, need_grad_x=need))
        nb = gout.numel() * 4 + x.numel() * 4 + (gout.numel() * 4 if need else 0)
        line += f" | bwd gx={int(need)} {ms*1e3:8.1f} us {nb/ms/1e6:7.0f} GB/s"
        for name, tune in (("nt", 16), ("cached", 32), ("nt plain order+stores", 16 | 64)):
            ms = timed(lambda: _hip.diag_apply_bwd(gout, x, s1, s2, u, n_samples=S, need_grad_x=need, tune=tune))
            line += f" [{name} {ms*1e3:.1f}]"
    if S * D * D * 4 <= (4 << 30):
        def matrix():
            W = WBarFunction.apply(s1.unsqueeze(0), u.unsqueeze(0), s2.unsqueeze(0), None, True).squeeze(0)
            return torch.matmul(x, W.transpose(1, 2)) + bias
        line += f" | matrix route fwd {timed(matrix, iters=5, warm=2)*1e3:9.1f} us"
    print(line + "   " + k[6:], flush=True)
    del x, out, gout
