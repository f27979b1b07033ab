"""Time WBarFunction's backward: one-launch kernel (whvi_wbar_bwd) vs the differentiable op chain, plus the
WHVILinear(512, 512) forward_mc + backward step of BASELINE config 2 and the toy training step under both."""
import sys, time
sys.path.insert(0, ".")
import torch
from whvi_amd import _hip
from whvi_amd.layers import WHVILinear
from whvi_amd.weights import WBarFunction

dev = "cuda"


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


supported = _hip.wbar_bwd_supported
for (J, S, D) in ((1, 32, 512), (1, 16, 1024), (1, 8, 4096), (3, 16, 64), (1, 64, 2048)):
    s1, s2 = (torch.randn(J, D, device=dev, requires_grad=True) for _ in range(2))
    u = torch.randn(J, S, D, device=dev, requires_grad=True)
    gw = torch.randn(J, S, D, D, device=dev)

    def step():
        W = WBarFunction.apply(s1, u, s2, None)
        torch.autograd.grad(W, (s1, u, s2), gw)

    def fwd():
        WBarFunction.apply(s1, u, s2, None)

    res = {}
    for mode in ("kernel", "chain"):
        _hip.wbar_bwd_supported = supported if mode == "kernel" else (lambda *a: False)
        res[mode] = timed(step)
    _hip.wbar_bwd_supported = supported
    f = timed(fwd)
    gb = gw.numel() * 4 / 1e9
    print(f"J={J} S={S} D={D}: fwd {f:.3f} ms | fwd+bwd kernel {res['kernel']:.3f} ms, chain {res['chain']:.3f} ms "
          f"| bwd-only kernel {res['kernel'] - f:.3f} ms = {gb / max(res['kernel'] - f, 1e-9) * 1e3:.0f} GB/s of grad_W", flush=True)

layer = WHVILinear(512, 512).to(dev)
x = torch.randn(4096, 512, device=dev)


def layer_step():
    layer.zero_grad(set_to_none=True)
    y = layer.forward_mc(x, 32)
    (y.square().mean() + layer.kl).backward()


for mode in ("kernel", "chain"):
    _hip.wbar_bwd_supported = supported if mode == "kernel" else (lambda *a: False)
    print(f"WHVILinear(512,512) forward_mc(32) + backward, batch 4096 [{mode}]: {timed(layer_step, 20):.3f} ms", flush=True)
_hip.wbar_bwd_supported = supported

big = WHVILinear(4096, 4096).to(dev)
xs = torch.randn(64, 4096, device=dev)


def big_fwd():
    with torch.no_grad():
        big.forward_mc(xs, 8)


def big_step():
    big.zero_grad(set_to_none=True)
    (big.forward_mc(xs, 8).square().mean() + big.kl).backward()


print(f"WHVILinear(4096,4096) forward_mc(8), batch 64: forward {timed(big_fwd, 20):.3f} ms, forward+backward {timed(big_step, 20):.3f} ms", flush=True)
