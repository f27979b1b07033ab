"""tests/golden/make_golden.py -- regenerate the committed golden fixtures from the LIVE reference.

Runs only in the build container (needs /root/reference and oracle/_ref, see oracle/build_ref.py):

    python tests/golden/make_golden.py

The reference's Python package is imported from where it lies.  ``import fwht_cpp`` inside it
resolves to the reference's OWN compiled C++ FWHT (oracle/_ref, first on sys.path) and
``import fwht_cuda`` to this repo's drop-in module (never called: everything here is on the CPU,
src/weights.py:35-41).  Only DATA is written: inputs, recorded random draws and the reference's
outputs, as small .npz files next to this script.  No reference source is copied.

The random vectors the reference draws inside forward passes (``torch.randn(self.D, ...)``,
src/weights.py:82,92) are recorded by wrapping ``torch.randn`` and stored, so tests can replay
them on any device without depending on a generator's stream.
"""
import argparse
import zlib
import os
import sys

sys.dont_write_bytecode = True      # importing the reference must not leave __pycache__ files in its checkout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def bind_reference_src(reference):
    """Make ``import src.*`` resolve to the REFERENCE tree only.  The reference's ``src`` has no ``__init__.py``
    (a namespace package), so this repo's ``src/`` alias package -- a regular package -- would win the import no
    matter where it sits on sys.path; pin the package object by hand and check where the modules come from."""
    import types
    pkg = types.ModuleType("src")
    pkg.__path__ = [os.path.join(reference, "src")]
    sys.modules["src"] = pkg
    import src.layers
    assert os.path.abspath(src.layers.__file__).startswith(os.path.abspath(reference) + os.sep), src.layers.__file__


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    ref_so_dir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.isdir(args.reference) or not os.path.isdir(ref_so_dir):
        sys.exit("needs the reference tree and oracle/_ref (python oracle/build_ref.py)")
    sys.path[:0] = [ref_so_dir, args.reference, ROOT]

    import torch
    import fwht_cpp  # the reference's compiled module
    assert os.path.dirname(fwht_cpp.__file__) == ref_so_dir, fwht_cpp.__file__
    bind_reference_src(args.reference)
    import src.fwht.cpp.fwht as ref_cpp
    import src.fwht.python.fwht as ref_py
    from src.utils import build_H, kl_diag_normal, matmul_diag_left, matmul_diag_right
    from src.layers import WHVILinear
    from src.networks import WHVIRegression
    from src.likelihoods import GaussianLikelihood
    import torch.nn as nn

    torch.set_num_threads(1)   # make the dense-H matmul path reproducible

    # ------------------------------------------------------------------ FWHT vectors
    out = {}
    gen = torch.Generator().manual_seed(20260203)
    for d in (1, 2, 4, 8, 32, 64, 512, 1024, 4096):
        rows = 19 if d <= 64 else (5 if d <= 1024 else 3)
        xi = torch.randint(-8, 8, (rows, d), generator=gen, dtype=torch.int32)
        xf = torch.randn(rows, d, generator=gen, dtype=torch.float32)
        xd = torch.randn(rows, d, generator=gen, dtype=torch.float64)
        out[f"i32_in_{d}"] = xi.numpy()
        out[f"i32_out_{d}"] = ref_cpp.FWHTFunction.apply(xi).numpy()
        out[f"intf32_out_{d}"] = ref_cpp.FWHTFunction.apply(xi.float()).numpy()
        out[f"f32_in_{d}"] = xf.numpy()
        out[f"f32_out_{d}"] = ref_cpp.FWHTFunction.apply(xf).numpy()
        out[f"f32_pyout_{d}"] = ref_py.FWHTFunction.apply(xf).numpy()
        out[f"f64_in_{d}"] = xd.numpy()
        out[f"f64_out_{d}"] = ref_cpp.FWHTFunction.apply(xd).numpy()
    # the reference's own known-answer vectors (test/walsh.py:12-13,17-18), re-derived live
    for name, vec in (("kat1", [1.0, 2.0, 3.0, 4.0]), ("kat2", [0.0, 1.0, 2.0, 3.0])):
        a = torch.tensor([vec])
        out[name + "_in"] = a.numpy()
        out[name + "_out"] = ref_cpp.FWHTFunction.apply(a).numpy()
    # 3-D input as fed by benchmarks/walsh.py:21 (transform along dim 1)
    x3 = torch.randn(2, 16, 5, generator=gen)
    out["f32_in_3d"] = x3.numpy()
    out["f32_out_3d"] = ref_cpp.FWHTFunction.apply(x3).numpy()
    # dense H and the diag helpers
    out["H_8"] = build_H(8, torch.device("cpu")).numpy()
    A = torch.randn(32, 32, generator=gen)
    dvec = torch.randn(32, generator=gen)
    out["diag_A"], out["diag_d"] = A.numpy(), dvec.numpy()
    out["diag_left"] = matmul_diag_left(dvec, A).numpy()
    out["diag_right"] = matmul_diag_right(A, dvec).numpy()
    mu1, sd1 = torch.randn(10, generator=gen), torch.exp(torch.randn(10, generator=gen))
    mu2, sd2 = torch.randn(10, generator=gen), torch.exp(torch.randn(10, generator=gen))
    out["kl_args"] = torch.stack([mu1, sd1, mu2, sd2]).numpy()
    out["kl_value"] = kl_diag_normal(mu1, sd1, mu2, sd2).numpy()
    np.savez_compressed(os.path.join(HERE, "fwht_golden.npz"), **out)
    print("fwht_golden.npz:", len(out), "arrays")

    # ------------------------------------------------------------------ WHVILinear bundles
    real_randn = torch.randn
    recorded = []

    def recording_randn(*a, **k):
        t = real_randn(*a, **k)
        recorded.append(t.detach().clone())
        return t

    layers = {}
    configs = [
        # name, n_in, n_out, bias, batch
        ("sq8", 8, 8, False, 6), ("sq64b", 64, 64, True, 5), ("sq512", 512, 512, False, 4),
        ("sq4096", 4096, 4096, False, 2),
        ("st3x16", 3, 16, False, 7), ("st5x7b", 5, 7, True, 6), ("st13x128", 13, 128, False, 3),
        ("col1x10b", 1, 10, True, 9), ("col16x1", 16, 1, False, 8),
    ]
    for name, n_in, n_out, bias, batch in configs:
        torch.manual_seed(zlib.crc32(name.encode()) % 10007)
        layer = WHVILinear(n_in, n_out, lambda_=0.7, bias=bias)
        # make g_mu and bias non-trivial so every term is exercised
        with torch.no_grad():
            for pname, p in layer.named_parameters():
                if pname.endswith("g_mu") or pname.endswith("bias"):
                    p.copy_(real_randn(p.shape) * 0.3)
                if pname.endswith("s1") or pname.endswith("s2"):
                    p.mul_(10.0)
        x = real_randn(batch, n_in, requires_grad=True)
        weight = real_randn(batch, n_out)
        recorded.clear()
        torch.randn = recording_randn
        try:
            y = layer(x)
        finally:
            torch.randn = real_randn
        kl = layer.kl
        loss = (y * weight).sum() + kl
        loss.backward()
        b = {"n_in": n_in, "n_out": n_out, "bias": int(bias), "lambda_": 0.7,
             "x": x.detach().numpy(), "weight": weight.numpy(), "y": y.detach().numpy(),
             "kl": kl.detach().numpy(), "grad_x": x.grad.numpy(), "n_eps": len(recorded)}
        for i, e in enumerate(recorded):
            b[f"eps{i}"] = e.numpy()
        for pname, p in layer.named_parameters():
            b["param." + pname] = p.detach().numpy()
            b["grad." + pname] = p.grad.numpy()
        for k, v in b.items():
            layers[f"{name}/{k}"] = np.asarray(v)
        print(name, "eps draws:", len(recorded), "y", tuple(y.shape), "kl", float(kl))

    # direct weight sampling (use_lrt=False) and w_bar on a square matrix
    torch.manual_seed(5)
    sq = WHVILinear(16, 16, lambda_=0.7).weight_submodule
    with torch.no_grad():
        sq.g_mu.copy_(real_randn(16) * 0.3)
    recorded.clear()
    torch.randn = recording_randn
    try:
        W = sq.sample()
    finally:
        torch.randn = real_randn
    u = real_randn(16)
    for k, v in {"s1": sq.s1, "s2": sq.s2, "g_mu": sq.g_mu, "g_rho": sq.g_rho, "eps0": recorded[0], "W": W,
                 "u": u, "w_bar": sq.w_bar(u)}.items():
        layers[f"sample16/{k}"] = v.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "whvi_golden.npz"), **layers)
    print("whvi_golden.npz:", len(layers), "arrays")

    # ------------------------------------------------------------------ WHVIRegression + likelihood
    net_out = {}
    torch.manual_seed(11)
    net = WHVIRegression([nn.Linear(1, 8), nn.ReLU(), WHVILinear(8, 8), nn.ReLU(), nn.Linear(8, 2)],
                         train_samples=3, eval_samples=4)
    xb, yb = real_randn(10, 1), real_randn(10, 2)
    recorded.clear()
    torch.randn = recording_randn
    try:
        net.train()
        pred = net(xb)
    finally:
        torch.randn = real_randn
    mnll = net.likelihood.mnll_batch_estimate(yb, pred, 100)
    net_out.update({"x": xb.numpy(), "y": yb.numpy(), "pred": pred.detach().numpy(),
                    "mnll": mnll.detach().numpy(), "kl": net.kl.detach().numpy(), "n_eps": len(recorded)})
    for i, e in enumerate(recorded):
        net_out[f"eps{i}"] = e.numpy()
    for k, v in net.state_dict().items():
        net_out["state." + k] = v.numpy()
    # the reference's fixed likelihood example (test/likelihoods.py:9-33)
    y = torch.reshape(torch.tensor([0., 1., 2., -1.]), (-1, 1))
    y_hat = torch.tensor([[0.2, 1.1, 2.2, -1.3], [-0.1, 1.05, 2, -1.1]]).T.unsqueeze(1)
    net_out["lik_value"] = GaussianLikelihood(sigma=1.0).mnll_batch_estimate(y, y_hat, 12).detach().numpy()
    np.savez_compressed(os.path.join(HERE, "network_golden.npz"), **net_out)
    print("network_golden.npz:", len(net_out), "arrays")


if __name__ == "__main__":
    main()
