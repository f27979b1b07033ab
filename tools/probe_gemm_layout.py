"""tools/probe_gemm_layout.py -- forward_mc's GEMM for a shared 2-D input: torch.matmul(x (B, D), W^T (S, D, D)) as a batched
GEMM (what forward_mc does) vs ONE GEMM against the concatenated weights (B, D) x (D, S*D); config 2 and config 4 shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def timed(fn, iters=20, warm=5):
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


for (B, D, S) in ((4096, 512, 32), (45730, 1024, 16), (256, 1024, 16), (4096, 2048, 16)):
    x = torch.randn(B, D, device="cuda")
    W = torch.randn(S, D, D, device="cuda")
    a = timed(lambda: torch.matmul(x, W.transpose(1, 2)))
    Wc = W.reshape(S * D, D)
    b = timed(lambda: (x @ Wc.t()))
    c = timed(lambda: torch.nn.functional.linear(x, Wc))
    y1 = torch.matmul(x, W.transpose(1, 2))
    y2 = (x @ Wc.t()).view(B, S, D).permute(1, 0, 2)
    err = float((y1 - y2).abs().max() / y1.abs().max())
    fl = 2 * B * D * D * S / 1e12
    print(f"B={B} D={D} S={S}: batched {a:.3f} ms ({fl / a * 1e3:.0f} TF/s) | one GEMM {b:.3f} ms ({fl / b * 1e3:.0f} TF/s) | F.linear {c:.3f} ms | max rel diff {err:.1e}", flush=True)
