"""tools/probe_long_rows.py [GiB] -- rows longer than one wavefront tile, in place on a 4 GiB buffer, HIP-event timed.
Run twice for the A/B:   python tools/probe_long_rows.py 4            (production: one block per row up to D = 65536)
                         python tools/probe_long_rows.py 4 passes     (round 1: 4096-element pieces + high-bit passes; also
                         "ungrouped": the same with every pass over the whole buffer) -- selected through bits 20..22 of
                         whvi_fwht_ex's variant word, not through the environment.
Data stay finite: every launch is followed by an untimed rescale."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whvi_amd import _hip

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
mode = sys.argv[2] if len(sys.argv) > 2 else "block"
variant = {"block": None, "passes": 1 << 20, "ungrouped": 4 << 20}[mode]
cases = [(torch.float32, l) for l in (12, 13, 14, 15, 16, 17, 20)] + [(torch.float64, l) for l in (12, 13, 14, 15, 16)] + \
        [(torch.float16, l) for l in (12, 13, 14, 15, 16)] + [(torch.bfloat16, 16), (torch.int32, 16)]
for dtype, log2d in cases:
    if mode != "block" and dtype in (torch.float16, torch.bfloat16) and log2d > 13:
        continue                                    # no multi-pass form for 16-bit storage
    d = 1 << log2d
    esize = torch.empty(0, dtype=dtype).element_size()
    rows = int(gib * (1 << 30)) // (d * esize)
    if dtype == torch.int32:
        x = torch.randint(-3, 4, (rows, d), device="cuda", dtype=torch.int32)
    else:
        x = (torch.randn(rows, d, device="cuda", dtype=torch.float32) * 0.25).to(dtype)
    times = []
    for it in range(8):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        _hip.fwht_rows(x, out=x, variant=variant)
        e.record()
        torch.cuda.synchronize()
        times.append(s.elapsed_time(e))
        if dtype != torch.int32:
            x.mul_(2.0 ** (-(log2d // 2) - (it & 1) * (log2d & 1)))   # |H x| ~ sqrt(D) |x|: back to where the values started
    finite = bool(torch.isfinite(x.float()).all()) if dtype != torch.int32 else True
    ms = sum(times[3:]) / len(times[3:])
    print(f"{mode:6s} {str(dtype)[6:]:8s} D=2^{log2d}: {ms:.3f} ms for {gib:g} GiB -> {2 * x.numel() * esize / ms / 1e9:.2f} TB/s "
          f"algorithmic, finite={finite}  {_hip.last_kernel()[6:]}", flush=True)
    del x
