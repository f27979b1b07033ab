"""Sentinel check of the round-4 launches: every output buffer sits in the middle of a sentinel-filled allocation, random shapes
(ragged batches, partial tiles, blocks straddling samples); not one byte beyond either end may change, and every byte inside must
(whvi_diag_apply, whvi_diag_apply_bwd's grad_x, whvi_small_k_apply_f32, whvi_row_dot_f32 -- straight through the C ABI)."""
import numpy as np
import pytest
import torch

from whvi_amd import _hip

pytestmark = pytest.mark.gpu
DEV = "cuda"
PAD = 4096          # floats of sentinel on each side
SENT = -7.25e33


def _guarded(numel):
    buf = torch.full((numel + 2 * PAD,), SENT, device=DEV, dtype=torch.float32)
    return buf, buf[PAD:PAD + numel]


def _intact(buf, numel):
    return bool((buf[:PAD] == SENT).all()) and bool((buf[PAD + numel:] == SENT).all())


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    return [(i, int(rng.integers(2, 13)), int(rng.integers(1, 7)), int(rng.integers(1, 400)), bool(rng.integers(0, 2))) for i in range(n)]


@pytest.mark.parametrize("case,log2d,S,B,shared", _cases(24, 1))
def test_diag_apply_and_backward_stay_inside_their_buffers(case, log2d, S, B, shared, hip_lib):
    D = 1 << log2d
    g = torch.Generator(device=DEV).manual_seed(case)
    s1, s2, bias = (torch.randn(D, device=DEV, generator=g) for _ in range(3))
    u = torch.randn(1 + S, D, device=DEV, generator=g)
    x = torch.randn((B, D) if shared else (S, B, D), device=DEV, generator=g)
    buf, out = _guarded(S * B * D)
    fn = _hip.lib().whvi_diag_apply_f32
    rc = fn(out.data_ptr(), x.data_ptr(), s1.data_ptr(), s2.data_ptr(), u.data_ptr(), bias.data_ptr(), S, B, log2d,
            (1 if shared else 0) | 2, None)
    torch.cuda.synchronize()
    assert rc == 0 and _intact(buf, S * B * D) and bool((out != SENT).all())
    want = _hip.diag_apply(x, s1, s2, u, bias, n_samples=S)
    assert torch.equal(out.view(S, B, D), want)
    gbuf, gx = _guarded(S * B * D)
    gout = torch.randn(S, B, D, device=DEV, generator=g)
    n_slabs = int(_hip.lib().whvi_diag_apply_bwd_slabs(0, S, B, log2d))
    pbuf, part = _guarded(S * n_slabs * 2 * D)
    obuf, res = _guarded(4 * (1 + S) * D)
    rc = _hip.lib().whvi_diag_apply_bwd_f32(gx.data_ptr(), res.data_ptr(), part.data_ptr(), gout.data_ptr(), x.data_ptr(), s1.data_ptr(),
                                            s2.data_ptr(), u.data_ptr(), None, S, B, log2d, n_slabs, (1 if shared else 0) | 2, None)
    torch.cuda.synchronize()
    assert rc == 0 and _intact(gbuf, S * B * D) and _intact(pbuf, S * n_slabs * 2 * D) and _intact(obuf, 4 * (1 + S) * D)
    assert bool((gx != SENT).all()) and bool((part != SENT).all())
    assert bool((res.view(4, 1 + S, D)[:, 1:] != SENT).all())                 # row 0 of every slot is the caller's (left untouched)


@pytest.mark.parametrize("case,log2d,S,B,shared", _cases(16, 2))
def test_layer_apply_launches_stay_inside_their_buffers(case, log2d, S, B, shared, hip_lib):
    g = torch.Generator(device=DEV).manual_seed(100 + case)
    K = 4 if shared else 8
    N = [4, 16, 48, 64, 256, 1024, 3072, 4096][case % 8]
    x = torch.randn(B, K, device=DEV, generator=g)
    w = torch.randn(S, N, K, device=DEV, generator=g)
    buf, out = _guarded(S * B * N)
    rc = _hip.lib().whvi_small_k_apply_f32(out.data_ptr(), x.data_ptr(), w.data_ptr(), None, S, B, N, K.bit_length() - 1, 0, None)
    torch.cuda.synchronize()
    assert rc == 0 and _intact(buf, S * B * N) and bool((out != SENT).all())
    ref = torch.matmul(x.double(), w.double().transpose(1, 2))
    assert float((out.view(S, B, N).double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    D = 1 << log2d
    h = torch.randn(S, B, D, device=DEV, generator=g)
    wv = torch.randn(S, D, device=DEV, generator=g)
    ybuf, y = _guarded(S * B)
    rc = _hip.lib().whvi_row_dot_f32(y.data_ptr(), h.data_ptr(), wv.data_ptr(), None, S, B, log2d, 0, None)
    torch.cuda.synchronize()
    assert rc == 0 and _intact(ybuf, S * B) and bool((y != SENT).all())
    ref = torch.matmul(h.double(), wv.double().unsqueeze(-1)).view(-1)
    scale = float((h.double().abs() * wv.double().abs().unsqueeze(1)).sum(dim=-1).max())
    assert float((y.double() - ref).abs().max()) <= 2e-6 * scale
