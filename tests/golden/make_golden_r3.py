"""tests/golden/make_golden_r3.py -- round-3 golden fixtures from the LIVE reference (build container only).

    python tests/golden/make_golden_r3.py

Same rules as make_golden.py / make_golden_r2.py: the reference's package is imported from where it lies
(/root/reference), ``fwht_cpp`` is the reference's own compiled C++ FWHT (oracle/_ref), only DATA is written.

pipeline_golden.npz -- the COLUMN-scaling pipeline of BASELINE config 3, ``x -> x . S2 . H . diag(g_k) . H . S1`` per
  Monte-Carlo sample, composed from the reference's OWN primitives -- ``matmul_diag_right`` (src/utils.py:15-23) and
  the FWHT autograd function over its C++ extension (src/fwht/cpp/fwht.py:7-18) -- for rows in (batch, sample, D) and
  (sample, batch, D) order, float32 (D = 64 / 512 / 2048) and float64 (D = 64 / 512); also with per-sample outer vectors (every
  "sample" an independent matrix with its own s1 / s2, the WHVIStackedMatrix situation of src/weights.py:130-132);
  plus the one-transform half ``matmul_diag_right(fwht(matmul_diag_right(x, b)), a)``.  Until now the column axis was
  pinned only through the oracle's own composition of the (reference-pinned) primitives; these vectors pin it to the
  reference's code directly.
  Also int32 / int64 rows whose sums overflow (inputs near +/-2^30, +/-2^62; D = 64 and 4096): what the reference's
  integer tensors do with them (wrap), recorded from ``fwht_cpp.forward``.
"""
import os
import sys

sys.dont_write_bytecode = True      # importing the reference must not leave __pycache__ files in its checkout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


def main():
    ref_so_dir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.isdir(REFERENCE) or not os.path.isdir(ref_so_dir):
        sys.exit("needs the reference tree and oracle/_ref (python oracle/build_ref.py)")
    sys.path[:0] = [ref_so_dir, REFERENCE, ROOT]
    import torch
    import fwht_cpp
    assert os.path.dirname(fwht_cpp.__file__) == ref_so_dir, fwht_cpp.__file__
    sys.path.insert(0, HERE)
    from make_golden import bind_reference_src
    bind_reference_src(REFERENCE)
    from src.utils import matmul_diag_right
    from src.fwht.cpp.fwht import FWHTFunction

    fwht = FWHTFunction.apply
    out = {}
    g = torch.Generator().manual_seed(2026)
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        for D in ((64, 512, 2048) if dtype == torch.float32 else (64, 512)):
            S, B = 2, 3
            x = torch.randn(S * B, D, generator=g, dtype=dtype)
            s1, s2 = torch.randn(S, D, generator=g, dtype=dtype), torch.randn(S, D, generator=g, dtype=dtype)
            gk = torch.randn(S, D, generator=g, dtype=dtype)
            key = f"{tag}_D{D}"
            out[key + "/x"], out[key + "/s1"], out[key + "/s2"], out[key + "/g"] = (t.numpy() for t in (x, s1, s2, gk))
            for order, sample_of in (("batch", lambda r: r % S), ("sample", lambda r: r // B)):
                shared = torch.stack([matmul_diag_right(fwht(matmul_diag_right(fwht(matmul_diag_right(x[r:r + 1], s2[0])),
                                                                               gk[sample_of(r)])), s1[0])[0]
                                      for r in range(S * B)])
                per = torch.stack([matmul_diag_right(fwht(matmul_diag_right(fwht(matmul_diag_right(x[r:r + 1], s2[sample_of(r)])),
                                                                            gk[sample_of(r)])), s1[sample_of(r)])[0]
                                   for r in range(S * B)])
                half = torch.stack([matmul_diag_right(fwht(matmul_diag_right(x[r:r + 1], gk[sample_of(r)])), s1[0])[0]
                                    for r in range(S * B)])
                out[f"{key}/{order}/shared"], out[f"{key}/{order}/per_sample"], out[f"{key}/{order}/one_transform"] = (
                    shared.numpy(), per.numpy(), half.numpy())
    rng = np.random.default_rng(30)
    for D in (64, 4096):
        mag = rng.integers((1 << 30) - 4096, (1 << 30) + 4096, (2, D))
        sign = rng.choice([-1, 1], (2, D))
        x32 = (mag * sign).astype(np.int32)
        x64 = ((mag.astype(np.int64) << 32) * sign).astype(np.int64)
        out[f"wrap_i32_D{D}/in"], out[f"wrap_i32_D{D}/out"] = x32, fwht_cpp.forward(torch.from_numpy(x32)).numpy()
        out[f"wrap_i64_D{D}/in"], out[f"wrap_i64_D{D}/out"] = x64, fwht_cpp.forward(torch.from_numpy(x64)).numpy()
    path = os.path.join(HERE, "pipeline_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
