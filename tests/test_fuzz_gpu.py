"""Randomised GPU parity sweep: shapes the fixed tests do not enumerate (prime row counts, every launch
geometry boundary: < 32 tiles per CU, the 1024/512-thread path, grids that are / are not multiples of 8 for
the XCD-contiguous order, the >= 256 MiB non-temporal path) -- bit-exact against the oracle on sampled rows
and through H.H = D.I on the whole tensor.  WHVI_FUZZ_CASES raises the case count for soak runs."""
import os

import numpy as np
import pytest
import torch

import oracle
from whvi_amd import _hip

pytestmark = pytest.mark.gpu
DEV = "cuda"
N_CASES = int(os.environ.get("WHVI_FUZZ_CASES", "40"))


def _cases():
    rng = np.random.default_rng(20260203)
    dtypes = [torch.float32, torch.float32, torch.int32, torch.float16, torch.float64, torch.bfloat16]
    out = []
    for i in range(N_CASES):
        dt = dtypes[i % len(dtypes)]
        log2d = int(rng.integers(0, 17 if dt not in (torch.float16, torch.bfloat16) else 14))   # incl. multi-pass rows
        d = 1 << log2d
        budget = int(rng.choice([1 << 14, 1 << 20, 1 << 25, 1 << 27]))          # elements
        rows = max(1, int(budget // d * rng.uniform(0.5, 1.5)))
        if rng.random() < 0.5:
            rows |= 1                                                            # odd / ragged
        out.append((i, dt, log2d, rows))
    # one case per dtype that crosses the 256 MiB non-temporal threshold with a ragged tail
    out += [(1000, torch.float32, 11, (1 << 15) + 13), (1001, torch.float16, 12, (1 << 15) + 7),
            (1002, torch.int32, 3, (1 << 23) + 5)]
    return out


@pytest.mark.parametrize("case,dtype,log2d,rows", _cases())
def test_random_shape(case, dtype, log2d, rows, hip_lib):
    d = 1 << log2d
    g = torch.Generator(device=DEV).manual_seed(case)
    lim = 2 if dtype in (torch.float16, torch.bfloat16) else 4
    xi = torch.randint(-lim, lim + 1, (rows, d), generator=g, device=DEV, dtype=torch.int32)
    x = xi.to(dtype)
    fx = _hip.fwht_rows(x)
    # whole tensor: involution on small integers (exact in every dtype here: |fx| <= lim * D)
    if dtype == torch.bfloat16 and lim * d > 256:
        back = None                       # bf16 cannot hold lim*D exactly beyond 8 bits
    else:
        back = _hip.fwht_rows(fx)
        assert torch.equal(back.to(torch.float64), xi.to(torch.float64) * d), (case, dtype, log2d, rows)
    # sampled rows against the oracle, bit for bit
    pick = torch.unique(torch.tensor([0, rows // 3, rows // 2, rows - 1]).clamp_(0, rows - 1)).to(DEV)
    sub = x[pick].cpu()
    if dtype in (torch.float16, torch.bfloat16):
        want = torch.from_numpy(oracle.fwht(sub.float().numpy())).to(dtype)
    else:
        want = torch.from_numpy(oracle.fwht(sub.numpy()))
    assert torch.equal(fx[pick].cpu().view(torch.uint8), want.view(torch.uint8)), (case, dtype, log2d, rows)
    # in place == out of place
    y = x.clone()
    _hip.fwht_rows(y, out=y)
    assert torch.equal(y.view(torch.uint8), fx.view(torch.uint8))


def _fused_cases():
    rng = np.random.default_rng(77)
    out = []
    for i in range(max(12, N_CASES // 2)):
        log2d = int(rng.integers(2, 13))
        S = int(rng.integers(1, 9))
        axis = "col" if rng.random() < 0.6 else "row"
        layout = int(rng.integers(0, 2))            # 0: (batch, sample, D)  1: (sample, batch, D)
        B = int(rng.integers(1, 40))
        big = rng.random() < 0.25
        out.append((i, log2d, S, axis, layout, B * (400 if big else 1), np.float32, 0))
    # float64 and the per-sample outer vectors (flags bit 0: a, bit 1: c), every third case >= 32 tiles per CU (the tuned
    # instantiations of dispatch.hpp: launch_fused); drawn from a second stream so the cases above stay what they were
    rng = np.random.default_rng(78)
    for i in range(max(18, N_CASES // 2)):
        dtype = np.float64 if i % 2 else np.float32
        log2d = int(rng.integers(1 if dtype == np.float64 else 2, 13))
        S = int(rng.integers(1, 9))
        axis = "col" if rng.random() < 0.6 else "row"
        layout = int(rng.integers(0, 2))
        B = int(rng.integers(1, 40))
        out.append((100 + i, log2d, S, axis, layout, B * (400 if i % 3 == 0 else 1), dtype, int(rng.integers(0, 4))))
    return out


@pytest.mark.parametrize("case,log2d,S,axis,layout,B,dtype,flags", _fused_cases())
def test_random_fused(case, log2d, S, axis, layout, B, dtype, flags, hip_lib):
    d = 1 << log2d
    if B * S * d > (1 << 26):
        B = max(1, (1 << 26) // (S * d))
    rng = np.random.default_rng(1000 + case)
    rows = B * S
    x = rng.standard_normal((rows, d)).astype(dtype)
    stride = 1 if layout == 0 else B
    a_ps, c_ps = bool(flags & 1), bool(flags & 2)
    if axis == "col":
        unit = d
        kw = dict(axis="col", n_samples=S, sample_stride=stride)
    else:
        G = int(rng.choice([1, 3, d, 2 * d + 1]))
        rows = (rows // G) * G or G
        x = rng.standard_normal((rows, d)).astype(dtype)
        unit = G
        stride = G if layout == 1 else 1
        kw = dict(axis="row", n_samples=S, sample_stride=stride, group_rows=G)
    a = rng.standard_normal((S, unit) if a_ps else unit).astype(dtype)
    c = rng.standard_normal((S, unit) if c_ps else unit).astype(dtype)
    b = rng.standard_normal((S, unit)).astype(dtype)
    want = oracle.pipeline(x, a, b, c, a_per_sample=a_ps, c_per_sample=c_ps, **kw)
    t = lambda v: torch.from_numpy(v).to(DEV)   # noqa: E731
    got = _hip.fused_shs(t(x), t(a), t(b), t(c), a_per_sample=a_ps, c_per_sample=c_ps, **kw).cpu().numpy()
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (case, log2d, S, axis, layout, rows, dtype, flags)


def _wbar_cases():
    rng = np.random.default_rng(4242)
    out = []
    for i in range(max(16, N_CASES // 2)):
        dt = torch.float64 if i % 4 == 3 else torch.float32
        log2d = int(rng.integers(1 if dt == torch.float64 else 2, 12))
        d = 1 << log2d
        J = int(rng.integers(1, 5))
        S = int(rng.integers(1, 5))
        R = d if rng.random() < 0.5 else int(rng.integers(1, d + 1))
        if J * S * R * d > (1 << 24):
            J, S = 1, 1
        out.append((i, dt, J, S, d, R, bool(i % 2)))
    return out


@pytest.mark.parametrize("case,dtype,J,S,D,R,mean", _wbar_cases())
def test_random_weight_construction(case, dtype, J, S, D, R, mean, hip_lib):
    """whvi_wbar_fwd / whvi_wbar_bwd on random (J, S, D, R): the forward bit for bit against the CPU oracle's
    two-transform pipeline on every matrix (and the in-epilogue mean add against a separate add), the backward
    against the float64 closed form the as-written matrix implies (W = D diag(s1 u s2))."""
    g = torch.Generator().manual_seed(1000 + case)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    U = S + 1 if mean else S
    s1, s2 = (torch.randn(J, D, generator=g, dtype=dtype) for _ in range(2))
    u = torch.randn(J, U, D, generator=g, dtype=dtype)
    every = _hip.wbar_fwd(s1.to(DEV), u.to(DEV), s2.to(DEV), R).cpu().numpy()          # (J, U, R, D)
    eye = np.eye(D, dtype=npdt)[:R]
    for j in range(J):
        want = oracle.pipeline(np.tile(eye, (U, 1)), s1[j, :R].numpy(), u[j, :, :R].numpy(), s2[j, :R].numpy(),
                               n_samples=U, sample_stride=R, group_rows=R, axis="row").reshape(U, R, D)
        assert np.array_equal(every[j].view(np.uint8), want.view(np.uint8)), (case, j)
    from whvi_amd.weights import WBarFunction
    p = [t.to(DEV).requires_grad_() for t in (s1, u, s2)]
    W = WBarFunction.apply(p[0], p[1], p[2], None if R == D else R, mean)
    if mean:
        assert np.array_equal(W.detach().cpu().numpy().view(np.uint8), (every[:, :1] + every[:, 1:]).view(np.uint8))
    gw = torch.randn(W.shape, generator=g, dtype=dtype)
    got = torch.autograd.grad(W, p, gw.to(DEV))
    # closed form in float64: W[j,k] = D diag(s1 (u_k [+ u_0]) s2)
    s1d, s2d, ud, gd = s1.double()[:, None, :R], s2.double()[:, None, :R], u.double()[:, :, :R], gw.double()
    diag = torch.diagonal(gd, dim1=2, dim2=3)                                                 # (J, S, R)
    g_u_samples = D * s1d * s2d * diag
    usum = ud[:, 1:] + ud[:, :1] if mean else ud
    want_u = torch.cat((g_u_samples.sum(1, keepdim=True), g_u_samples), dim=1) if mean else g_u_samples
    want = {"s1": (D * usum * s2d * diag).sum(1), "u": want_u, "s2": (D * s1d * usum * diag).sum(1)}
    noise = (2e-6 if dtype == torch.float32 else 1e-14) * D * float(gd.abs().max()) * 8 * U
    bound = noise * float(max(s1d.abs().max(), 1) * max(s2d.abs().max(), 1) * max(ud.abs().max(), 1))
    for name, a in zip(("s1", "u", "s2"), got):
        a = a.double().cpu()
        assert float((a[..., :R] - want[name]).abs().max()) <= bound, (case, name)
        assert R == D or float(a[..., R:].abs().max()) == 0.0
    # every launch form of the backward kernel gives the same values: quarter-size tiles (what these sizes take by
    # default), 16 KiB tiles with the LDS-staged network (f32 rows of 256 .. 4096) and with the DPP network
    if _hip.wbar_bwd_supported(dtype, D):
        dev = [t.detach() for t in p]
        forms = [_hip.wbar_bwd(gw.to(DEV), dev[0], dev[1], dev[2], mean=mean, **kw)
                 for kw in ({}, {"tiles": "big"}, {"tiles": "big", "no_lds": True}, {"tiles": "small"})]
        first = 1 if mean else 0
        for other in forms[1:]:
            assert torch.equal(other[:, :, first:, :R], forms[0][:, :, first:, :R]), case


def _shared_cases():
    rng = np.random.default_rng(404)
    out = []
    for i in range(max(16, N_CASES // 2)):
        dtype = np.float64 if i % 3 == 2 else np.float32
        log2d = int(rng.integers(8 if dtype == np.float32 else 7, 13))
        S = int(rng.integers(1, 9))
        B = int(rng.integers(1, 70)) * (32 if i % 4 == 0 else 1)            # some batches whole blocks, most ragged
        out.append((i, dtype, log2d, S, B, bool(i % 2)))
    return out


@pytest.mark.parametrize("case,dtype,log2d,S,B,one", _shared_cases())
def test_random_shared_source(case, dtype, log2d, S, B, one, hip_lib):
    """WHVI_FUSED_SRC_SHARED (a (batch, D) source read by every sample) and WHVI_FUSED_ONE_TRANSFORM (the second half of
    the pipeline alone) on random shapes, bit for bit against ``oracle.pipeline`` on the expanded input."""
    d = 1 << log2d
    if S * B * d > (1 << 25):
        B = max(1, (1 << 25) // (S * d))
    rng = np.random.default_rng(3000 + case)
    x = rng.standard_normal((B, d)).astype(dtype)
    a, c = rng.standard_normal(d).astype(dtype), rng.standard_normal(d).astype(dtype)
    b = rng.standard_normal((S, d)).astype(dtype)
    t = lambda v: torch.from_numpy(v).to(DEV)   # noqa: E731
    want = oracle.pipeline(np.tile(x, (S, 1)), a, b, c, n_samples=S, sample_stride=B, axis="col")
    if one:
        first = _hip.fused_shs(t(x), None, t(c).reshape(1, -1), None, axis="col", n_samples=1, one_transform=True)
        got = _hip.fused_shs(first, t(a), t(b), None, axis="col", n_samples=S, sample_stride=B, src_shared=True, one_transform=True)
    else:
        got = _hip.fused_shs(t(x), t(a), t(b), t(c), axis="col", n_samples=S, sample_stride=B, src_shared=True)
    assert np.array_equal(got.cpu().numpy().view(np.uint8), want.view(np.uint8)), (case, dtype, log2d, S, B, one)


def _diag_cases():
    rng = np.random.default_rng(20261005)
    out = []
    for i in range(max(12, N_CASES // 2)):
        dtype = torch.float64 if i % 4 == 3 else torch.float32
        log2d = int(rng.integers(1 if dtype == torch.float64 else 2, 12 if dtype == torch.float64 else 13))
        S = int(rng.integers(1, 9))
        budget = int(rng.choice([1 << 12, 1 << 17, 1 << 21]))
        B = max(1, int(budget // (S << log2d) * rng.uniform(0.5, 1.5)))
        out.append((i, dtype, log2d, S, B, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)),
                    bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("case,dtype,log2d,S,B,shared,mean_plus,relu_in,relu_out,poison", _diag_cases())
def test_random_diag_apply(case, dtype, log2d, S, B, shared, mean_plus, relu_in, relu_out, poison, hip_lib):
    """whvi_diag_apply on random shapes / flag combinations (rows shorter and longer than a wave's 64 chunks, ragged batches,
    blocks straddling samples, partial tiles), with and without non-finite activations, value-identical to the matrix route:
    weight construction through the butterfly kernels + dense product (src/weights.py:87-93) with torch.relu around it."""
    from whvi_amd.weights import WBarFunction
    d = 1 << log2d
    g = torch.Generator(device=DEV).manual_seed(7000 + case)
    kw = dict(device=DEV, dtype=dtype, generator=g)
    s1, s2, bias = torch.randn(d, **kw), torch.randn(d, **kw), torch.randn(1, d, **kw)
    u = torch.randn(S + (1 if mean_plus else 0), d, **kw)
    x = torch.randn((B, d) if shared else (S, B, d), **kw)
    if poison:
        flat = x.view(-1, d)
        for r in torch.randint(0, flat.shape[0], (max(1, flat.shape[0] // 7),), device=DEV, generator=g).tolist():
            flat[r, int(torch.randint(0, d, (1,), device=DEV, generator=g))] = (float("inf"), float("-inf"), float("nan"))[r % 3]
    use_bias = case % 3 != 0
    got = _hip.diag_apply(x, s1, s2, u, bias if use_bias else None, n_samples=S, mean_plus=mean_plus, relu_in=relu_in, relu_out=relu_out)
    W = WBarFunction.apply(s1.unsqueeze(0), u.unsqueeze(0), s2.unsqueeze(0), None, mean_plus).squeeze(0)      # (S, D, D)
    want = torch.matmul(torch.relu(x) if relu_in else x, W.transpose(1, 2))
    want = want + bias if use_bias else want
    want = torch.relu(want) if relu_out else want
    na, nb = torch.isnan(got), torch.isnan(want)
    assert got.shape == (S, B, d) and bool((na == nb).all()) and bool((got[~na] == want[~nb]).all()), (case, dtype, log2d, S, B)
