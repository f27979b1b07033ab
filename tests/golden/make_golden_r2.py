"""tests/golden/make_golden_r2.py -- round-2 golden fixtures from the LIVE reference (build container only).

    python tests/golden/make_golden_r2.py

Same rules as make_golden.py (which it reuses for the import set-up): the reference's Python package is imported
from where it lies (/root/reference), ``fwht_cpp`` is the reference's own compiled C++ FWHT (oracle/_ref), only
DATA is written.  Two files:

config4_golden.npz -- BASELINE config 4's real layer shapes (src/layers.py:31-38 dispatch):
    ``WHVILinear(3, 1024)``   stacked, 256 sub-matrices of D = 4          (src/weights.py:135-160, :179-180)
    ``WHVILinear(1024, 1024)`` square                                      (src/weights.py:87-93)
    ``WHVILinear(1024, 1)``   column, transposed                           (src/weights.py:239-248)
  each as a forward / KL / backward bundle with the drawn eps recorded, plus one ``WHVIRegression``
  3 -> 1024 -> 1024 -> 1 training-mode pass (3 MC samples, batch 6): predictions, MNLL, KL, every gradient.

train_golden.npz -- the reference's ``train_model`` (src/networks.py:71-99) driven by the reference's
  ``make_optimizer`` (src/evaluation.py:15-27) on a toy set, eps recorded: per-step loss / MNLL / KL / learning rate,
  the state after training, and the ``epoch-0.pth`` checkpoint the loop writes (src/networks.py:95-96).  Two runs:
  "default" = ``make_optimizer(net)`` (lr = lambda0^2 * (1 + gamma t)^-p = 1e-6 * ..., the quirk of
  src/evaluation.py:25-26) and "fast" = ``make_optimizer(net, lambda0=0.05)`` so that parameters move visibly.
"""
import os
import sys

sys.dont_write_bytecode = True      # importing the reference must not leave __pycache__ files in its checkout
import tempfile
import zlib

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
    from src.evaluation import make_optimizer
    from torch.utils.data import DataLoader, TensorDataset

    torch.set_num_threads(1)   # the dense-H matmul path (D < 4096 on the host, src/weights.py:38-39) reproducible
    real_randn = torch.randn
    recorded = []

    def recording_randn(*a, **k):
        t = real_randn(*a, **k)
        recorded.append(t.detach().clone())
        return t

    def record(fn):
        recorded.clear()
        torch.randn = recording_randn
        try:
            return fn()
        finally:
            torch.randn = real_randn

    def perturb(module):
        """non-trivial g_mu / bias and O(0.1) sign vectors, so every term of the pipeline carries signal"""
        with torch.no_grad():
            for pname, p in module.named_parameters():
                if pname.endswith("g_mu") or pname.endswith("bias"):
                    p.copy_(real_randn(p.shape) * 0.3)
                if pname.endswith("s1") or pname.endswith("s2"):
                    p.mul_(10.0)

    # ------------------------------------------------------------------ config 4: layer bundles
    out = {}
    for name, n_in, n_out, batch in (("st3x1024", 3, 1024, 4), ("sq1024", 1024, 1024, 3), ("col1024x1", 1024, 1, 5)):
        torch.manual_seed(zlib.crc32(name.encode()) % 10007)
        layer = WHVILinear(n_in, n_out, lambda_=0.7)
        perturb(layer)
        x = real_randn(batch, n_in, requires_grad=True)
        weight = real_randn(batch, n_out)
        y = record(lambda: layer(x))
        eps = [e.numpy() for e in recorded]
        kl = layer.kl
        ((y * weight).sum() + kl).backward()
        b = {"n_in": n_in, "n_out": n_out, "bias": 0, "lambda_": 0.7, "x": x.detach().numpy(), "weight": weight.numpy(),
             "y": y.detach().numpy(), "kl": kl.detach().numpy(), "grad_x": x.grad.numpy(),
             "eps": np.stack(eps),                                               # (draws, D) in draw order
             "param_names": np.array("\n".join(n for n, _ in layer.named_parameters())),
             "params": np.concatenate([p.detach().numpy().reshape(-1) for p in layer.parameters()]),
             "grads": np.concatenate([p.grad.numpy().reshape(-1) for p in layer.parameters()])}
        for k, v in b.items():
            out[f"layer/{name}/{k}"] = np.asarray(v)
        print(name, "eps draws:", len(eps), "y", tuple(y.shape), "kl", float(kl))

    # ------------------------------------------------------------------ config 4: the network
    torch.manual_seed(41)
    S, batch = 3, 6
    net = WHVIRegression([WHVILinear(3, 1024, lambda_=2.0), nn.ReLU(), WHVILinear(1024, 1024, lambda_=2.0), nn.ReLU(),
                          WHVILinear(1024, 1, lambda_=2.0)], train_samples=S, eval_samples=2)
    perturb(net)
    xb, yb = real_randn(batch, 3), real_randn(batch, 1)
    net.train()
    loss = record(lambda: net.loss(xb, yb, n=100))
    draws = [e.numpy() for e in recorded]
    assert len(draws) == S * 258, len(draws)      # per sample: 256 sub-matrices of D = 4, then two D = 1024 draws
    per_sample = [draws[k * 258:(k + 1) * 258] for k in range(S)]
    loss.backward()
    out["net/x"], out["net/y"] = xb.numpy(), yb.numpy()
    out["net/loss"] = loss.detach().numpy()
    out["net/mnll"], out["net/kl"] = net.current_mnll.detach().numpy(), net.current_kl.detach().numpy()
    out["net/eps_layer0"] = np.stack([np.stack(d[:256]) for d in per_sample])          # (S, 256, 4)
    out["net/eps_layer1"] = np.stack([d[256] for d in per_sample])[:, None]            # (S, 1, 1024)
    out["net/eps_layer2"] = np.stack([d[257] for d in per_sample])[:, None]            # (S, 1, 1024)
    out["net/param_names"] = np.array("\n".join(n for n, _ in net.named_parameters()))
    out["net/params"] = np.concatenate([p.detach().numpy().reshape(-1) for p in net.parameters()])
    out["net/grads"] = np.concatenate([p.grad.numpy().reshape(-1) for p in net.parameters()])
    # the same pass again for the predictions themselves (same eps replayed into the reference)
    it = iter(draws)
    torch.randn = lambda *a, **k: torch.from_numpy(next(it))
    try:
        pred = net(xb)
    finally:
        torch.randn = real_randn
    out["net/pred"] = pred.detach().numpy()                                           # (batch, 1, S)
    print("network 3->1024->1024->1: pred", tuple(pred.shape), "loss", float(loss), "params", len(out["net/params"]))
    np.savez_compressed(os.path.join(HERE, "config4_golden.npz"), **out)
    print("config4_golden.npz:", len(out), "arrays")

    # ------------------------------------------------------------------ training trajectory
    out = {}
    for run, opt_kwargs, seed in (("default", {}, 101), ("fast", {"lambda0": 0.05}, 202)):
        torch.manual_seed(seed)
        S = 2
        net = WHVIRegression([WHVILinear(3, 16, lambda_=3.0), nn.ReLU(), WHVILinear(16, 16, lambda_=3.0), nn.ReLU(),
                              WHVILinear(16, 1, lambda_=3.0)], train_samples=S, eval_samples=4)
        perturb(net)
        X = real_randn(24, 3)
        Y = torch.sin(X.sum(dim=1, keepdim=True)) + 0.1 * real_randn(24, 1)
        loader = DataLoader(TensorDataset(X, Y), batch_size=8)            # no shuffling, like src/evaluation.py:76
        optimizer, scheduler = make_optimizer(net, **opt_kwargs)
        init = {k: v.clone() for k, v in net.state_dict().items()}
        trace = {"loss": [], "mnll": [], "kl": [], "lr": []}
        inner_loss = net.loss

        def traced_loss(*a, **k):
            trace["lr"].append(optimizer.param_groups[0]["lr"])
            value = inner_loss(*a, **k)
            trace["loss"].append(float(value))
            trace["mnll"].append(float(net.current_mnll))
            trace["kl"].append(float(net.current_kl))
            return value
        net.loss = traced_loss
        epochs1, epochs2 = 4, 3
        with tempfile.TemporaryDirectory() as ckpt:
            record(lambda: net.train_model(loader, optimizer, scheduler, epochs1=epochs1, epochs2=epochs2,
                                           checkpoint_dir=ckpt))
            saved = sorted(os.listdir(ckpt))
            checkpoint = torch.load(os.path.join(ckpt, "epoch-0.pth"))
        assert saved == ["epoch-0.pth"], saved
        steps = (epochs1 + epochs2) * 3
        draws = [e.numpy() for e in recorded]
        assert len(draws) == steps * S * 6 and len(trace["loss"]) == steps
        per = [draws[i * 6:(i + 1) * 6] for i in range(steps * S)]        # one forward of one sample: 4 + 1 + 1 draws
        out[f"{run}/eps_layer0"] = np.stack([np.stack(d[:4]) for d in per]).reshape(steps, S, 4, 4)
        out[f"{run}/eps_layer1"] = np.stack([d[4] for d in per]).reshape(steps, S, 1, 16)
        out[f"{run}/eps_layer2"] = np.stack([d[5] for d in per]).reshape(steps, S, 1, 16)
        out[f"{run}/X"], out[f"{run}/Y"] = X.numpy(), Y.numpy()
        out[f"{run}/epochs"] = np.array([epochs1, epochs2])
        out[f"{run}/optimizer_kwargs"] = np.array(repr(opt_kwargs))
        for k, v in trace.items():
            out[f"{run}/{k}"] = np.array(v, dtype=np.float64)
        for k, v in init.items():
            out[f"{run}/init.{k}"] = v.numpy()
        for k, v in net.state_dict().items():
            out[f"{run}/final.{k}"] = v.numpy()
        for k, v in checkpoint.items():
            out[f"{run}/ckpt_epoch0.{k}"] = v.numpy()
        out[f"{run}/training_flag_after"] = np.array(int(net.training))
        print(run, "steps", steps, "loss", trace["loss"][0], "->", trace["loss"][-1], "lr", trace["lr"][0], trace["lr"][-1],
              "sigma", float(init["likelihood.sigma"]), "->", float(net.state_dict()["likelihood.sigma"]))
    np.savez_compressed(os.path.join(HERE, "train_golden.npz"), **out)
    print("train_golden.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
