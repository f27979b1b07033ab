"""tests/golden/make_golden_r4.py -- round-4 golden fixture from the LIVE reference (build container only).

    python tests/golden/make_golden_r4.py

Same rules as make_golden.py / make_golden_r2.py (whose import set-up it reuses): the reference's Python package is imported
from where it lies (/root/reference), ``fwht_cpp`` is the reference's own compiled C++ FWHT (oracle/_ref), only DATA is written.

config4_real_size_golden.npz -- BASELINE config 4's network, ``WHVIRegression`` 3 -> 1024 -> 1024 -> 1 (src/networks.py:36-54
over src/layers.py:31-38: a stacked layer of 256 D = 4 sub-matrices, a square D = 1024 layer, a transposed column layer), at a
REAL batch: 512 rows x 16 Monte-Carlo samples (one GPU's share of the 128) in eval mode -- the predictions
``net(x)`` of shape (512, 1, 16) with every eps the reference drew recorded in draw order, plus ``eval_model``'s RMSE / MNLL
of them (src/networks.py:101-115,130-133).  The round-2 fixture of the same network holds batch 6 x 3 samples."""
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
    import torch.nn as nn
    import fwht_cpp
    assert os.path.dirname(fwht_cpp.__file__) == ref_so_dir, fwht_cpp.__file__
    sys.path.insert(0, HERE)
    from make_golden import bind_reference_src
    bind_reference_src(REFERENCE)
    from src.layers import WHVILinear
    from src.networks import WHVIRegression

    torch.set_num_threads(1)   # the dense-H matmul path (D < 4096 on the host, src/weights.py:38-39) reproducible
    real_randn = torch.randn
    recorded = []

    def recording_randn(*a, **k):
        t = real_randn(*a, **k)
        recorded.append(t.detach().clone())
        return t

    S, batch = 16, 512
    torch.manual_seed(4096)
    net = WHVIRegression([WHVILinear(3, 1024, lambda_=2.0), nn.ReLU(), WHVILinear(1024, 1024, lambda_=2.0), nn.ReLU(),
                          WHVILinear(1024, 1, lambda_=2.0)], train_samples=1, eval_samples=S)
    with torch.no_grad():      # non-trivial g_mu / bias and O(0.1) sign vectors, so every term of the pipeline carries signal
        for pname, p in net.named_parameters():
            if pname.endswith("g_mu") or pname.endswith("bias"):
                p.copy_(real_randn(p.shape) * 0.3)
            if pname.endswith("s1") or pname.endswith("s2"):
                p.mul_(10.0)
    x = real_randn(batch, 3)
    y = torch.sin(x.sum(dim=1, keepdim=True)) + 0.1 * real_randn(batch, 1)
    net.eval()
    torch.randn = recording_randn
    try:
        with torch.no_grad():
            pred = net(x)                                            # (batch, 1, S), src/networks.py:36-54
    finally:
        torch.randn = real_randn
    draws = [e.numpy() for e in recorded]
    assert pred.shape == (batch, 1, S) and len(draws) == S * 258, (pred.shape, len(draws))
    per_sample = [draws[k * 258:(k + 1) * 258] for k in range(S)]   # per sample: 256 sub-matrices of D = 4, then two D = 1024 draws
    out = {"x": x.numpy(), "y": y.numpy(), "pred": pred.numpy(),
           "eps_layer0": np.stack([np.stack(d[:256]) for d in per_sample]),            # (S, 256, 4)
           "eps_layer1": np.stack([d[256] for d in per_sample])[:, None],              # (S, 1, 1024)
           "eps_layer2": np.stack([d[257] for d in per_sample])[:, None],              # (S, 1, 1024)
           "param_names": np.array("\n".join(n for n, _ in net.named_parameters())),
           "params": np.concatenate([p.detach().numpy().reshape(-1) for p in net.parameters()])}
    # eval_model on the SAME predictions (the eps replayed into the reference): RMSE of the predictive mean and test MNLL
    it = iter(draws)
    torch.randn = lambda *a, **k: torch.from_numpy(next(it))
    try:
        with torch.no_grad():
            rmse, mnll = net.eval_model(x, y)
    finally:
        torch.randn = real_randn
    out["rmse"], out["mnll"] = np.array(rmse, dtype=np.float64), np.array(mnll, dtype=np.float64)
    # The same pass with the reference's DEVICE dispatch: on a GPU every square matrix transforms through the butterfly FWHT
    # (src/weights.py:34-36), on the host the D < 4096 ones go through a dense H matmul instead (:38-39), whose rounding
    # leaves ~1e-7 off-diagonals in the as-written (exactly diagonal) weight -- summed over 1024 hidden units that is ~1e-4 of
    # these predictions.  The CUDA kernel cannot run here; the reference's own vectorised butterfly FWHT (src/fwht/python,
    # what the reference itself selects on the host for D >= 4096; bit-equal to its C++ FWHT, test/walsh.py:51-59) stands in
    # for it on every square matrix.  This is the record a GPU implementation is compared with.
    import src.weights as ref_weights
    squares = [m for m in net.modules() if isinstance(m, ref_weights.WHVISquarePow2Matrix)]
    assert len(squares) == 256 + 1 + 1
    for m in squares:
        m.fwht = ref_weights.fwht_python.apply
    it = iter(draws)
    torch.randn = lambda *a, **k: torch.from_numpy(next(it))
    try:
        with torch.no_grad():
            pred_b = net(x)
            it = iter(draws)
            rmse_b, mnll_b = net.eval_model(x, y)
    finally:
        torch.randn = real_randn
    out["pred_butterfly"] = pred_b.numpy()
    out["rmse_butterfly"], out["mnll_butterfly"] = np.array(rmse_b, dtype=np.float64), np.array(mnll_b, dtype=np.float64)
    print("butterfly dispatch vs dense-H dispatch of the reference itself: max |diff| / max |pred| =",
          float(np.abs(out["pred_butterfly"] - out["pred"]).max() / np.abs(out["pred"]).max()))
    np.savez_compressed(os.path.join(HERE, "config4_real_size_golden.npz"), **out)
    print("config4_real_size_golden.npz: pred", out["pred"].shape, "max |pred|", float(np.abs(out["pred"]).max()),
          "std over samples", float(out["pred"].std(axis=2).mean()), "rmse", rmse, "mnll", mnll,
          "bytes", os.path.getsize(os.path.join(HERE, "config4_real_size_golden.npz")))


if __name__ == "__main__":
    main()
