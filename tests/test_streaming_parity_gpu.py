"""Oracle parity of every fused / weight-kernel instantiation a number is published for, AT THE SIZE IT IS TIMED AT.

`launch_fused` (whvi_amd/csrc/dispatch.hpp) takes its tuned paths only for problems of >= 32 tiles per CU and its
streaming (non-temporal) form beyond the 256 MiB Infinity Cache; the small-shape tests of tests/test_fused_gpu.py run
15-33 rows and therefore never see those instantiations.  Here every family of DESIGN.md section 5.2's table --
f32 / f64, shared and per-sample outer scale vectors, D = 512 .. 4096 -- runs on >= 320 MiB in place AND out of place,
in both row orders ((batch, sample, D): sample_stride = 1; (sample, batch, D): sample_stride = batch), with a ragged
tail, asserts through ``whvi_last_kernel`` that the launch it checks is the instantiation production dispatches, and
compares >= 64 sampled rows bit for bit with ``oracle.pipeline`` (the CPU restatement of
matmul_diag . fwht . matmul_diag . fwht, src/weights.py:73,84 / src/utils.py:4-23 / src/fwht/cpp/fwht.cpp:7-18).
Same for the f64 weight construction and its backward (whvi_wbar_fwd_f64 / whvi_wbar_bwd_f64) and for int32
wrap-around (reference: integer tensors wrap, src/fwht/cpp/fwht.cpp:11-13 on at::kInt)."""
import numpy as np
import pytest
import torch

import oracle
from whvi_amd import _hip

pytestmark = pytest.mark.gpu
DEV = "cuda"
MIB = 1 << 20


def _bits(a):
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def expected_fused_kernel(dtype, log2d, per_sample, layout):
    """The symbol production dispatches for a column-axis launch of >= 32 tiles per CU beyond the Infinity Cache
    (dispatch.hpp: launch_fused).  Last template argument = which scale vectors the block stages in LDS (kernels.hpp):
    1 = a and c (shared ones, or those of the block's one sample), 0 = none (every vector from L2)."""
    name, esize = ("float", 4) if dtype == torch.float32 else ("double", 8)
    K = 32 if (dtype == torch.float64 and log2d == 12) else 16      # one f64 row of 4096 per wave: 128 data registers
    # staged: shared a / c always; per-sample ones when whole blocks lie inside one sample -- rows in (sample, batch, D)
    # order with an aligned batch, or (batch, sample, D) order with one 128-register row per tile (f64 D = 4096), where
    # the block takes four rows of the same sample, S rows apart
    stage = 1 if (not per_sample or layout == "sample" or (layout == "batch" and K == 32)) else 0
    return f"whvi::fused_shs_kernel<{name}, {log2d}, {K}, 1, false, true, 256, 0, {stage}, false, false>"    # as rocprofv3 prints it


CASES = [(torch.float32, 9), (torch.float32, 10), (torch.float32, 11), (torch.float32, 12),
         (torch.float64, 9), (torch.float64, 11), (torch.float64, 12)]


@pytest.mark.parametrize("in_place", [True, False])
@pytest.mark.parametrize("layout", ["batch", "sample", "sample_ragged"])
@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("dtype,log2d", CASES)
def test_fused_streaming_instantiation_vs_oracle(dtype, log2d, per_sample, layout, in_place, hip_lib):
    """layout: "batch" = rows in (batch, sample, D) order (sample_stride = 1, BASELINE config 3's order); "sample" =
    (sample, batch, D) with the batch a multiple of the rows per block (fastfood's order: every block inside one sample);
    "sample_ragged" = the same with an odd batch (blocks straddle samples: no per-block staging of per-sample vectors)."""
    d = 1 << log2d
    esize = 4 if dtype == torch.float32 else 8
    S = 48                                               # not a power of two: the FastDiv row -> sample map
    B = (320 * MIB) // (d * esize * S) + 1               # >= 320 MiB; B * S rows
    B = (B + 31) // 32 * 32 + (1 if layout == "sample_ragged" else 0)      # 32 = the most rows a block holds (D = 512 f32)
    rows = B * S
    assert rows * d * esize >= 320 * MIB
    npdt = np.float32 if dtype == torch.float32 else np.float64
    g = torch.Generator(device=DEV).manual_seed(1000 * log2d + 10 * esize + 2 * per_sample + len(layout))
    x = torch.randn(rows, d, device=DEV, dtype=dtype, generator=g)
    nv = S if per_sample else 1
    a = torch.randn(nv, d, device=DEV, dtype=dtype, generator=g) * 0.1
    c = torch.randn(nv, d, device=DEV, dtype=dtype, generator=g) * 0.1
    b = torch.randn(S, d, device=DEV, dtype=dtype, generator=g)
    stride = 1 if layout == "batch" else B
    rng = np.random.default_rng(7 + log2d)
    rpt = max(1, (16384 // esize) // d)                  # rows per 16 KiB tile
    edge = [0, 1, rpt - 1, rpt, 4 * rpt - 1, 4 * rpt, S - 1, S, B - 1, B, B + 1, 2 * B - 1, 2 * B, rows // 2,
            rows - B - 1, rows - B, rows - rpt - 1, rows - 2, rows - 1]
    idx = np.unique(np.clip(np.concatenate([edge, rng.integers(0, rows, 72)]), 0, rows - 1))
    assert len(idx) >= 64
    tidx = torch.from_numpy(idx).to(DEV)
    src_rows = x[tidx].cpu().numpy()
    out = _hip.fused_shs(x, a if per_sample else a[0], b, c if per_sample else c[0], axis="col", n_samples=S,
                         sample_stride=stride, out=x if in_place else None, a_per_sample=per_sample,
                         c_per_sample=per_sample)
    assert _hip.last_kernel() == expected_fused_kernel(dtype, log2d, per_sample, layout), _hip.last_kernel()
    assert (out.data_ptr() == x.data_ptr()) == in_place
    got = out[tidx].cpu().numpy()
    # oracle on the gathered rows: every gathered row is its own "sample" carrying its row's b (and a, c when per-sample)
    an, bn, cn = a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy()
    smp = (idx // stride) % S
    want = oracle.pipeline(src_rows, an[smp] if per_sample else an[0], bn[smp], cn[smp] if per_sample else cn[0],
                           n_samples=len(idx), sample_stride=1, axis="col", a_per_sample=per_sample,
                           c_per_sample=per_sample)
    assert want.dtype == npdt and np.array_equal(_bits(got), _bits(want))
    assert bool((out != 0).any(dim=1).all()) and bool(torch.isfinite(out[::1031]).all())      # every row written
    if not in_place:
        assert np.array_equal(x[tidx].cpu().numpy(), src_rows), "out-of-place must leave the source untouched"


@pytest.mark.parametrize("dtype,D,S", [(torch.float64, 2048, 12), (torch.float64, 512, 170), (torch.float64, 4096, 3),
                                       (torch.float32, 2048, 20), (torch.float32, 512, 340), (torch.float32, 4096, 5)])
def test_weight_construction_at_streaming_size(dtype, D, S, hip_lib):
    """whvi_wbar_fwd_f32 / _f64 with the mean matrix added, > 256 MiB of matrices (the write-only streaming launch:
    non-temporal global stores, XCD-sliced order -- the instantiations ``extras.wbar_fwd`` times): sampled rows bit for bit
    against ``oracle.pipeline`` on one-hot rows (row i of diag(s2): the dataflow of src/weights.py:73) plus ONE add of the
    mean row (src/weights.py:93); off-diagonals exactly zero; every row of every matrix written."""
    g = torch.Generator(device=DEV).manual_seed(D + S)
    s1, s2 = (torch.randn(1, D, device=DEV, dtype=dtype, generator=g) for _ in range(2))
    u = torch.randn(1, 1 + S, D, device=DEV, dtype=dtype, generator=g)
    mean = _hip.wbar_fwd(s1, u, s2, D, first=0, count=1).view(1, D, D)
    full = _hip.wbar_fwd(s1, u, s2, D, base=mean, first=1)                                   # (1, S, D, D)
    log2d = D.bit_length() - 1
    assert full.numel() * full.element_size() > 256 * MIB
    name = "double" if dtype == torch.float64 else "float"
    assert _hip.last_kernel() == f"whvi::wbar_fwd_kernel<{name}, {log2d}, {32 if (D == 4096 and dtype == torch.float64) else 16}, true>", \
        _hip.last_kernel()
    assert bool((full[0].diagonal(dim1=1, dim2=2) != 0).all())                               # every row of every matrix
    rng = np.random.default_rng(D)
    ks = np.unique(np.concatenate([[0, S - 1], rng.integers(0, S, 6)]))
    iis = np.unique(np.concatenate([[0, 1, 63, 64, D - 1], rng.integers(0, D, 11)]))
    s1n, s2n, un = s1[0].cpu().numpy(), s2[0].cpu().numpy(), u[0].cpu().numpy()
    onehot = np.eye(D, dtype=np.float64 if dtype == torch.float64 else np.float32)[iis]
    n = len(iis)
    kw = dict(n_samples=1, sample_stride=n, group_rows=n, axis="row")
    want_mean = oracle.pipeline(onehot, s1n[iis], un[0][iis], s2n[iis], **kw)
    assert np.array_equal(_bits(mean[0][torch.from_numpy(iis).to(DEV)].cpu().numpy()), _bits(want_mean))
    for k in ks:
        want = want_mean + oracle.pipeline(onehot, s1n[iis], un[1 + k][iis], s2n[iis], **kw)
        got = full[0, int(k)][torch.from_numpy(iis).to(DEV)].cpu().numpy()
        assert np.array_equal(_bits(got), _bits(want)), k
    off = full[0, S - 1].clone()
    off.diagonal().zero_()
    assert float(off.abs().max()) == 0.0                                                    # SURVEY finding 1


@pytest.mark.parametrize("mean", [False, True])
@pytest.mark.parametrize("D,S", [(2048, 12), (4096, 3)])
def test_wbar_backward_f64_at_streaming_size(D, S, mean, hip_lib):
    """whvi_wbar_bwd_f64 on > 256 MiB of dL/dW (non-temporal loads): every output against the closed form the
    as-written matrix implies (W = D diag(s1 u s2): dL/du_i = D s1_i s2_i gW_ii, ...; float64, 1e-13 of the largest
    term), and bit-identical to the cache-resident launch of the same kernel family run on two of the matrices alone."""
    g = torch.Generator(device=DEV).manual_seed(7 * D + S)
    f64 = dict(device=DEV, dtype=torch.float64, generator=g)
    s1, s2 = torch.randn(1, D, **f64), torch.randn(1, D, **f64)
    u = torch.randn(1, S + (1 if mean else 0), D, **f64)
    gw = torch.randn(1, S, D, D, **f64)
    assert gw.numel() * 8 > 256 * MIB
    out = _hip.wbar_bwd(gw, s1, u, s2, mean=mean)
    log2d = D.bit_length() - 1
    K = 32 if D == 4096 else 16
    assert _hip.last_kernel() == f"whvi::wbar_bwd_kernel<double, {log2d}, {K}, true, {'true' if mean else 'false'}, 0>", _hip.last_kernel()
    first = 1 if mean else 0
    diag = torch.diagonal(gw, dim1=2, dim2=3)                                              # (1, S, D)
    uk = u[:, first:]
    u_tot = uk + (u[:, :1] if mean else 0.0)
    want_u = D * s1.unsqueeze(1) * s2.unsqueeze(1) * diag
    want_s2 = D * s1.unsqueeze(1) * u_tot * diag
    want_s1 = D * u_tot * s2.unsqueeze(1) * diag
    # the kernel's c = (H g1)[i] is a D^2-term signed sum of O(1) values that cancels down to D * gW_ii * s1_i
    noise = 1e-13 * D * float(gw.abs().max()) * float(s1.abs().max() * s2.abs().max() * u.abs().max()) * 8
    for got, want, name in ((out[0], want_u, "u"), (out[1], want_s1, "s1"), (out[2], want_s2, "s2")):
        assert float((got[:, first:] - want).abs().max()) <= noise, name
    k0 = S // 2
    sub_u = torch.cat((u[:, :1], u[:, first + k0:first + k0 + 2]), dim=1) if mean else u[:, k0:k0 + 2]
    two = _hip.wbar_bwd(gw[:, k0:k0 + 2].contiguous(), s1, sub_u.contiguous(), s2, mean=mean)   # <= 256 MiB: cached launch
    assert ", false, " in _hip.last_kernel()
    assert torch.equal(two[:, :, first:], out[:, :, first + k0:first + k0 + 2])


@pytest.mark.parametrize("in_place", [True, False])
def test_int32_wraps_like_the_reference(in_place, hip_lib):
    """int32 rows whose partial sums overflow: inputs near +/-2^30, D = 4096 -- twelve doublings wrap many times.  The
    reference's integer tensors wrap (two's complement ATen adds, src/fwht/cpp/fwht.cpp:11-13); the oracle's int32 leg is
    built with -fwrapv and pinned to the reference on wrapping inputs in tests/test_oracle.py.  Streaming size (320 MiB)
    so that the launch checked is the one the int32 stream is timed with; sampled rows bit for bit, the rest through
    H.H = D.I modulo 2^32."""
    d, rows = 4096, (320 * MIB) // (4096 * 4) + 3
    g = torch.Generator(device=DEV).manual_seed(31)
    big = torch.randint((1 << 30) - 4096, (1 << 30) + 4096, (rows, d), device=DEV, dtype=torch.int32, generator=g)
    sign = torch.randint(0, 2, (rows, d), device=DEV, dtype=torch.int32, generator=g) * 2 - 1
    x = big * sign
    rng = np.random.default_rng(4)
    idx = np.unique(np.concatenate([[0, 1, 3, 4, rows - 2, rows - 1], rng.integers(0, rows, 64)]))
    tidx = torch.from_numpy(idx).to(DEV)
    src = x[tidx].cpu().numpy()
    keep = x.clone() if in_place else x
    y = _hip.fwht_rows(x, out=x if in_place else None)
    assert _hip.last_kernel() == "whvi::fwht_rows_kernel<int, 12, 16, 0, false, true, 256, 1, false>", _hip.last_kernel()
    want = oracle.fwht(src)
    assert want.dtype == np.int32 and np.array_equal(y[tidx].cpu().numpy(), want)
    # the check really wrapped: exact integer results do not fit 32 bits
    exact = oracle.fwht(src.astype(np.int64))
    assert np.abs(exact).max() > 2 ** 31 and np.array_equal(exact.astype(np.int32), want)
    ref = oracle.load_reference_cpp()
    if ref is not None:                                      # the reference's own C++ FWHT (oracle/_ref), int32 tensors
        assert np.array_equal(ref.forward(torch.from_numpy(src)).numpy(), want)
    # H.H = 4096.I modulo 2^32 on the whole buffer
    back = _hip.fwht_rows(y)
    assert torch.equal(back, keep * 4096)


@pytest.mark.parametrize("dtype,log2d,B", [(torch.float32, 11, 8192), (torch.float32, 12, 1028), (torch.float32, 8, 4096 * 5 + 3),
                                           (torch.float64, 11, 1024), (torch.float64, 12, 516), (torch.float32, 9, 33)])
def test_shared_source_equals_the_expanded_launch(dtype, log2d, B, hip_lib):
    """WHVI_FUSED_SRC_SHARED: a (batch, D) input shared by all MC samples (a fastfood layer's first pass) is read by every
    sample from the caches instead of being expanded to (S, batch, D) first.  Bit-equal to the launch on the expanded
    input (whose instantiations the test above pins to the oracle) and, on sampled rows, to ``oracle.pipeline`` directly;
    BASELINE config 3's shape (D = 2048, batch 8192, 64 samples: 4 GiB written) included, batches that are and are not
    multiples of the rows per block, a launch below the streaming threshold, the source left untouched."""
    d, S = 1 << log2d, 64 if B >= 512 else 5
    g = torch.Generator(device=DEV).manual_seed(log2d * 131 + B)
    x = torch.randn(B, d, device=DEV, dtype=dtype, generator=g)
    a, c = (torch.randn(d, device=DEV, dtype=dtype, generator=g) * 0.1 for _ in range(2))
    b = torch.randn(S, d, device=DEV, dtype=dtype, generator=g)
    keep = x.clone()
    got = _hip.fused_shs(x, a, b, c, axis="col", n_samples=S, sample_stride=B, src_shared=True)
    symbol = _hip.last_kernel()
    assert symbol.endswith(", true, false>") and symbol.count(",") == 10, symbol      # the SHARED_SRC instantiation
    assert got.shape == (S * B, d) and torch.equal(x, keep)
    rows = S * B
    rng = np.random.default_rng(B)
    idx = np.unique(np.clip(np.concatenate([[0, 1, B - 1, B, B + 1, 2 * B - 1, rows - B, rows - 1], rng.integers(0, rows, 70)]), 0, rows - 1))
    want = oracle.pipeline(x[torch.from_numpy(idx % B).to(DEV)].cpu().numpy(), a.cpu().numpy(), b.cpu().numpy()[idx // B],
                           c.cpu().numpy(), n_samples=len(idx), sample_stride=1, axis="col")
    assert np.array_equal(_bits(got[torch.from_numpy(idx).to(DEV)].cpu().numpy()), _bits(want))
    expanded = _hip.fused_shs(x.repeat(S, 1), a, b, c, axis="col", n_samples=S, sample_stride=B)
    assert _hip.last_kernel().endswith(", false, false>")                               # the expanded input: neither flag
    assert torch.equal(got.view(torch.uint8), expanded.view(torch.uint8))
    # WHVI_FUSED_ONE_TRANSFORM: fwht(c * x) ONCE, then one transform per sample on the shared result -- the same bits
    t = _hip.fused_shs(x, None, c.reshape(1, -1), None, axis="col", n_samples=1, one_transform=True)
    assert _hip.last_kernel().endswith(", false, true>"), _hip.last_kernel()
    assert torch.equal(t, _hip.fwht_rows(c * x)), "scale -> FWHT alone == a multiply and the plain transform"
    halves = _hip.fused_shs(t, a, b, None, axis="col", n_samples=S, sample_stride=B, src_shared=True, one_transform=True)
    assert _hip.last_kernel().endswith(", true, true>"), _hip.last_kernel()
    assert torch.equal(halves.view(torch.uint8), got.view(torch.uint8))
    with pytest.raises(RuntimeError, match="one_transform needs"):
        _hip.fused_shs(x, a, b, c, axis="col", n_samples=S, sample_stride=B, one_transform=True)
    with pytest.raises(RuntimeError, match="src_shared needs"):
        _hip.fused_shs(torch.zeros(4, 64, device=DEV), a[:64], b[:, :64], c[:64], axis="col", n_samples=S, sample_stride=4, src_shared=True)


@pytest.mark.parametrize("one", [False, True])
@pytest.mark.parametrize("dtype,log2d,S,B", [(torch.float32, 11, 64, 256),     # S > the 32 blocks per sample
                                             (torch.float32, 12, 10, 1024),    # S does not divide the 256 blocks per sample
                                             (torch.float32, 11, 80, 256),     # not a power of two, > the 32 blocks per sample
                                             (torch.float64, 12, 300, 32),     # one 128-register row per tile
                                             (torch.float32, 9, 36, 2048)])    # 32 rows per block
def test_shared_source_with_per_sample_outer_vectors(dtype, log2d, S, B, one, hip_lib):
    """WHVI_FUSED_SRC_SHARED together with per-sample a / c on launches big enough for the tuned dispatch (>= 32 tiles per
    CU; shared-source block order "sample fastest within an XCD", a / c of the block's ONE sample staged in LDS).  Round 3
    shipped a block map that put the four waves of such a block into different samples whenever S did not divide the
    blocks per sample while staging wave 0's vectors only (kernels.hpp) -- no test had this flag combination.
    Sampled rows bit for bit against ``oracle.pipeline(a_per_sample, c_per_sample)`` (matmul_diag_right . fwht chains of
    src/utils.py:15-23 + src/fwht/cpp/fwht.cpp:7-18) and the whole result against the expanded-input launch."""
    d = 1 << log2d
    g = torch.Generator(device=DEV).manual_seed(log2d * 977 + S * 13 + B)
    x = torch.randn(B, d, device=DEV, dtype=dtype, generator=g)
    a = torch.randn(S, d, device=DEV, dtype=dtype, generator=g) * 0.1
    c = torch.randn(S, d, device=DEV, dtype=dtype, generator=g) * 0.1
    b = torch.randn(S, d, device=DEV, dtype=dtype, generator=g)
    rows = S * B
    props = torch.cuda.get_device_properties(0)
    k_rows = 1 if (dtype == torch.float64 and log2d == 12) else max(1, (16384 // x.element_size()) // d)   # rows per tile
    n_tiles = rows // k_rows
    assert n_tiles >= (16 if k_rows == 1 and dtype == torch.float64 else 32) * props.multi_processor_count, "not the tuned dispatch"
    if one:
        got = _hip.fused_shs(x, a, b, None, axis="col", n_samples=S, sample_stride=B, src_shared=True, one_transform=True,
                             a_per_sample=True)
        assert _hip.last_kernel().endswith(", 1, true, true>"), _hip.last_kernel()          # a staged per block
    else:
        got = _hip.fused_shs(x, a, b, c, axis="col", n_samples=S, sample_stride=B, src_shared=True, a_per_sample=True,
                             c_per_sample=True)
        assert _hip.last_kernel().endswith(", 1, true, false>"), _hip.last_kernel()
    rng = np.random.default_rng(S * B)
    idx = np.unique(np.clip(np.concatenate([[0, 1, B - 1, B, B + 1, 2 * B - 1, rows - B, rows - 1],
                                            rng.integers(0, rows, 150)]), 0, rows - 1))
    smp = idx // B
    an, bn, cn = a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy()
    xs = x[torch.from_numpy(idx % B).to(DEV)].cpu().numpy()
    if one:      # a_s (.) FWHT(b_s (.) x): the pipeline with c = 1 and an identity first transform has no oracle form;
        want = an[smp] * oracle.fwht(bn[smp] * xs)                                         # compose it from the parts
    else:
        want = oracle.pipeline(xs, an[smp], bn[smp], cn[smp], n_samples=len(idx), sample_stride=1, axis="col",
                               a_per_sample=True, c_per_sample=True)
    assert np.array_equal(_bits(got[torch.from_numpy(idx).to(DEV)].cpu().numpy()), _bits(want))
    if one:
        expanded = _hip.fused_shs(x.repeat(S, 1), a, b, None, axis="col", n_samples=S, sample_stride=B, one_transform=True,
                                  a_per_sample=True)
    else:
        expanded = _hip.fused_shs(x.repeat(S, 1), a, b, c, axis="col", n_samples=S, sample_stride=B, a_per_sample=True,
                                  c_per_sample=True)
    assert torch.equal(got.view(torch.uint8), expanded.view(torch.uint8))
