"""tools/probe_fused_inst.py -- TB/s and kernel symbol of the fused S.H.diag(g).H.S launch per instantiation family:
f32 / f64, shared vs per-sample outer scale vectors (WHVI_FUSED_{A,C}_PER_SAMPLE), D = 512 .. 4096, 4 GiB in place,
plus the stacked-layer weight construction (J = 256 sub-matrices of D = 4, whvi_wbar_fwd).  Run under `python tools/_tuning.py --run <lib.so> tools/probe_fused_inst.py ...` to A/B
another build of the library."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

dev = torch.device("cuda", 0)


def timed(fn, iters=20, warm=25):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


print("library:", _hip.LIB_PATH)
LAYOUTS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["batch-major"]      # batch-major: (batch, sample, D); sample-major
for dtype in (torch.float32, torch.float64):
  for layout in LAYOUTS:
    for d in (512, 2048, 4096):
        S = 64
        rows = (1 << 32) // (d * (4 if dtype == torch.float32 else 8))
        stride = 1 if layout == "batch-major" else rows // S
        x = torch.randn(rows, d, device=dev, dtype=dtype)
        sgn = lambda n: ((torch.randint(0, 2, (n, d), device=dev) * 2 - 1).to(dtype) * d ** -0.5)   # noqa: E731
        g = torch.randn(S, d, device=dev, dtype=dtype)
        for per_sample in (False, True):
            a, c = (sgn(S), sgn(S)) if per_sample else (sgn(1)[0], sgn(1)[0])
            ms = timed(lambda: _hip.fused_shs(x, a, g, c, axis="col", n_samples=S, sample_stride=stride, out=x,
                                              a_per_sample=per_sample, c_per_sample=per_sample))
            gbs = 2 * x.numel() * x.element_size() / ms / 1e6
            print(f"{str(dtype)[6:]:8s} {layout:12s} D={d:5d} per-sample a/c={int(per_sample)}: {ms:.4f} ms {gbs:7.1f} GB/s = {gbs / 8000:.3f} "
                  f"finite={bool(torch.isfinite(x[::4099]).all())}  {_hip.last_kernel()}", flush=True)
        del x
# stacked layer of BASELINE config 4: WHVILinear(3, 1024) = 256 sub-matrices of D = 4, 16 MC samples (+ mean)
J, S, D = 256, 16, 4
s1, s2, u = torch.randn(J, D, device=dev), torch.randn(J, D, device=dev), torch.randn(J, 1 + S, D, device=dev)
ms = timed(lambda: _hip.wbar_fwd(s1, u, s2, D, first=1), iters=200, warm=50)
print(f"wbar_fwd J={J} S={S} D={D}: {ms * 1e3:.2f} us per launch ({J * S * D * D * 4 / 1e3:.0f} KB written: launch-bound)")
J, S, D = 1, 64, 2048
s1, s2, u = torch.randn(J, D, device=dev), torch.randn(J, D, device=dev), torch.randn(J, 1 + S, D, device=dev)
ms = timed(lambda: _hip.wbar_fwd(s1, u, s2, D, first=1), iters=50, warm=30)
print(f"wbar_fwd J={J} S={S} D={D}: {ms:.4f} ms {J * S * D * D * 4 / ms / 1e6:.0f} GB/s written")
