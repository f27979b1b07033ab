"""GPU parity of the batched row FWHT (C ABI ``whvi_fwht_*``) against the CPU oracle.

Bar: bit-exact for int32, integer-valued and random float32/float64 (the kernel applies the
butterfly stages in the oracle's ascending-stride order, src/fwht/cpp/fwht.cpp:7-18); fp16/bf16
follow the build's contract "f32 arithmetic, one rounding on store" and are compared bit-exactly
with ``oracle(x.float()).to(half)``.
"""
import numpy as np
import pytest
import torch

import oracle
import fwht_cuda
from whvi_amd import _hip
from whvi_amd.fwht.cuda import FWHTFunction

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _rand(rows, d, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    if dtype == torch.int32:
        return torch.randint(-8, 8, (rows, d), generator=g, dtype=torch.int32)
    return torch.randn(rows, d, generator=g, dtype=torch.float32).to(dtype)


def _oracle(x: torch.Tensor) -> torch.Tensor:
    if x.dtype in (torch.float16, torch.bfloat16):
        return torch.from_numpy(oracle.fwht(x.float().numpy())).to(x.dtype)
    return torch.from_numpy(oracle.fwht(x.numpy()))


@pytest.mark.parametrize("log2d", list(range(0, 14)))
@pytest.mark.parametrize("dtype", [torch.float32, torch.int32, torch.float64, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("variant", [0, 1])
def test_bit_exact_all_sizes(log2d, dtype, variant, hip_lib):
    if log2d > _hip.max_log2d(dtype):
        pytest.skip("beyond the supported row length")
    if log2d > (12 if dtype == torch.float64 else 13) and variant != 0:
        pytest.skip("beyond the single-wave row limit: only the production (block-per-row) launch applies")
    d = 1 << log2d
    for rows in (1, 19, 67):   # odd batches like test/walsh.py:73; 67 rows -> a partial last tile
        x = _rand(rows, d, dtype, seed=1000 * log2d + rows)
        got = _hip.fwht_rows(x.to(DEV), variant=variant).cpu()
        want = _oracle(x)
        assert got.dtype == x.dtype and got.shape == x.shape
        assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), \
            f"log2d={log2d} rows={rows} dtype={dtype} variant={variant}: " \
            f"max|diff|={float((got.double() - want.double()).abs().max())}"


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("log2d", [9, 10, 11, 12])
def test_tuning_variants_agree(log2d, variant, hip_lib):
    d = 1 << log2d
    x = _rand(2051, d, torch.float32, seed=7 + log2d)   # several tiles per wave + partial tile
    got = _hip.fwht_rows(x.to(DEV), variant=variant | (1 << 8)).cpu()   # 1 block/CU: forces the tile loop
    assert torch.equal(got, _oracle(x))


def test_reference_known_answers(hip_lib):
    # test/walsh.py:12-13,17-18
    a = torch.tensor([[1.0, 2.0, 3.0, 4.0], [0.0, 1.0, 2.0, 3.0]], device=DEV)
    out = fwht_cuda.fwht(a).cpu()
    assert torch.equal(out, torch.tensor([[10.0, -2.0, -4.0, 0.0], [6.0, -2.0, -4.0, 0.0]]))


def test_cuda_simple_and_large_like_reference(hip_lib):
    # test/walsh.py:61-79 (D=4 batch 2, default allclose; D=1024 batch 19, atol 1e-4) vs dense H
    from whvi_amd.utils import build_H
    g = torch.Generator().manual_seed(0)
    for d, batch, kw in ((4, 2, {}), (1024, 19, {"atol": 1e-4})):
        H = build_H(d, DEV)
        A = torch.randn(batch, d, generator=g).to(DEV)
        reference = (H @ A.T).T
        output = FWHTFunction.apply(A)
        assert torch.allclose(output, reference, **kw)


def test_out_of_place_and_in_place(hip_lib):
    x = _rand(33, 512, torch.float32, 3).to(DEV)
    keep = x.clone()
    y = fwht_cuda.fwht(x)
    assert torch.equal(x, keep), "input must be left untouched (fwht_cuda.cpp:11)"
    _hip.fwht_rows(x, out=x)
    assert torch.equal(x, y)


def test_noncontiguous_and_misaligned_inputs(hip_lib):
    base = _rand(64, 256, torch.float32, 5)
    xt = base.to(DEV).T                       # strides (1, 256)
    assert torch.equal(fwht_cuda.fwht(xt).cpu(), _oracle(base.T.contiguous()))
    flat = torch.zeros(64 * 4 + 1, device=DEV)
    flat[1:] = base[:, :4].reshape(-1).to(DEV)
    view = flat[1:].view(64, 4)               # data_ptr is 4 bytes off a 16-byte boundary
    assert view.data_ptr() % 16 != 0
    assert torch.equal(fwht_cuda.fwht(view).cpu(), _oracle(base[:, :4].contiguous()))


def test_error_behaviour(hip_lib):
    with pytest.raises(RuntimeError, match="X must be a CUDA tensor"):
        fwht_cuda.fwht(torch.randn(2, 4))
    with pytest.raises(RuntimeError, match="X must be two-dimensional"):
        fwht_cuda.fwht(torch.randn(2, 4, 4, device=DEV))
    with pytest.raises(RuntimeError, match="n must be a power of 2"):
        fwht_cuda.fwht(torch.randn(2, 6, device=DEV))
    with pytest.raises(RuntimeError, match="outside the supported range"):
        fwht_cuda.fwht(torch.randn(1, 1 << 25, device=DEV))
    assert fwht_cuda.fwht(torch.empty(0, 8, device=DEV)).shape == (0, 8)


def test_gradcheck_float64(hip_lib):
    # src/fwht/grad_check.py:26-34
    x = torch.randn(3, 32, dtype=torch.float64, device=DEV, requires_grad=True)
    assert torch.autograd.gradcheck(FWHTFunction.apply, (x,))
    assert torch.autograd.gradgradcheck(FWHTFunction.apply, (x,))


def test_involution_and_linearity_full_size(hip_lib):
    """Size-independent properties at a BASELINE-scale shape (D=4096, 2^15 rows, 512 MiB):
    H.H = D.I exactly on small integers, and FWHT(x + y) == FWHT(x) + FWHT(y) on integers."""
    d, rows = 4096, 1 << 15
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randint(-4, 5, (rows, d), generator=g, device=DEV, dtype=torch.int32).float()
    y = torch.randint(-4, 5, (rows, d), generator=g, device=DEV, dtype=torch.int32).float()
    fx, fy = _hip.fwht_rows(x), _hip.fwht_rows(y)
    assert torch.equal(_hip.fwht_rows(fx), x * d)
    assert torch.equal(_hip.fwht_rows(x + y), fx + fy)
    # spot-check 64 rows against the oracle
    idx = torch.arange(0, rows, rows // 64)
    assert torch.equal(fx[idx.to(DEV)].cpu(), _oracle(x[idx.to(DEV)].cpu()))


@pytest.mark.parametrize("dtype,log2d,rows", [(torch.float32, 10, 40000), (torch.float16, 12, 12000),
                                              (torch.float64, 11, 9000), (torch.int32, 9, 70001)])
def test_production_launch_geometries(dtype, log2d, rows, hip_lib):
    """Mid-size problems (>= 32 tiles per CU but inside the Infinity Cache) take cached accesses -- 256-thread
    blocks, f64 1024 -- incl. a partial last tile / partial last block; spot-check rows against the
    oracle and the whole tensor through H.H = D.I on small integers."""
    d = 1 << log2d
    g = torch.Generator(device=DEV).manual_seed(5)
    xi = torch.randint(-3, 4, (rows, d), generator=g, device=DEV, dtype=torch.int32)
    x = xi.to(dtype)
    fx = _hip.fwht_rows(x)
    back = _hip.fwht_rows(fx)
    if dtype == torch.float16:      # |FWHT| can exceed fp16 range only for huge rows; here max 3*4096 < 65504
        assert torch.equal(fx.float(), _hip.fwht_rows(x.float()))
    else:
        assert torch.equal(back, x * d)
    idx = torch.tensor([0, 1, rows // 2, rows - 2, rows - 1], device=DEV)
    assert torch.equal(fx[idx].cpu().view(torch.uint8), _oracle(x[idx].cpu()).view(torch.uint8))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("log2d,rows", [(12, 40000), (10, 150001), (7, 1200003), (3, 17000001)])
def test_streaming_launch_of_16bit_types(dtype, log2d, rows, hip_lib):
    """Streams >= 256 MiB of fp16 / bf16 take the LDS-staged butterfly network (256-thread blocks, block barrier
    before the non-temporal stores), incl. a partial last tile: every row bit-identical to the ds_bpermute
    cross-check network (whvi_fwht_ex variant 1) and to f32 arithmetic + one rounding; in place == out of place."""
    d = 1 << log2d
    g = torch.Generator(device=DEV).manual_seed(log2d)
    x = (torch.randn(rows, d, generator=g, device=DEV) * 0.25).to(dtype)
    assert x.numel() * x.element_size() >= 256 << 20
    got = _hip.fwht_rows(x)
    assert torch.equal(got.view(torch.int16), _hip.fwht_rows(x, variant=1).view(torch.int16))
    idx = torch.tensor([0, 1, rows // 3, rows - 2, rows - 1], device=DEV)
    assert torch.equal(got[idx].view(torch.int16), _hip.fwht_rows(x[idx].float()).to(dtype).view(torch.int16))
    assert torch.equal(got[idx].cpu().view(torch.int16), _oracle(x[idx].cpu()).view(torch.int16))
    keep = got.clone()
    _hip.fwht_rows(x, out=x)
    assert torch.equal(x.view(torch.int16), keep.view(torch.int16))


def test_headline_size_in_place_involution(hip_lib):
    """The bench workload itself (D = 4096, 2^20 rows = 16 GiB, > 2^32 bytes of offsets, in place,
    streaming launch: non-temporal, 256-thread blocks + store barrier): H.H = 4096.I exactly on small integers, plus
    oracle rows."""
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2 ** 30:
        pytest.skip("needs ~34 GiB of free HBM")
    d, rows = 4096, 1 << 20
    x = torch.empty(rows, d, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    step = 1 << 16
    for r in range(0, rows, step):      # fill in slabs: randint has no >2^31-element fast path guarantees
        x[r:r + step] = torch.randint(-2, 3, (step, d), generator=g, device=DEV, dtype=torch.int32).float()
    keep = x.clone()
    idx = torch.tensor([0, 1, 4095, rows // 2 + 7, rows - 1], device=DEV)
    _hip.fwht_rows(x, out=x)
    assert torch.equal(x[idx].cpu(), _oracle(keep[idx].cpu()))
    _hip.fwht_rows(x, out=x)
    keep.mul_(float(d))
    assert torch.equal(x, keep)


def test_side_stream_and_autograd_thread(hip_lib):
    """Launches go to torch's CURRENT stream (the reference uses the legacy default stream,
    fwht_cuda_kernel.cu:171,177) and backward runs on the autograd engine's thread."""
    side = torch.cuda.Stream()
    x = _rand(257, 1024, torch.int32, 9).float().to(DEV)     # integers: H.H = D.I holds exactly
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        y = fwht_cuda.fwht(x)
        z = fwht_cuda.fwht(y)
    side.synchronize()
    assert torch.equal(z, x * 1024)
    xr = x.clone().requires_grad_(True)
    w = torch.randn_like(x)
    (FWHTFunction.apply(xr) * w).sum().backward()
    assert torch.equal(xr.grad, fwht_cuda.fwht(w))       # d/dx <w, xH> = wH (H symmetric)


@pytest.mark.parametrize("dtype,log2d", [(torch.float32, 14), (torch.float32, 15), (torch.float32, 16), (torch.float32, 17),
                                         (torch.float32, 20), (torch.float64, 13), (torch.float64, 14), (torch.float64, 15),
                                         (torch.float64, 16), (torch.int32, 14), (torch.int32, 15), (torch.int32, 16),
                                         (torch.int32, 19), (torch.float16, 14), (torch.float16, 15), (torch.float16, 16),
                                         (torch.bfloat16, 14), (torch.bfloat16, 15), (torch.bfloat16, 16)])
def test_rows_longer_than_one_wave(dtype, log2d, hip_lib):
    """D beyond the register-resident limit (the reference's fwht_batch2 territory, dead code there): one block of
    4 / 8 / 16 waves per row up to D = 65536 (f64: 32768), the bits above the wave tile through LDS; beyond that the
    block kernel on 65536-element pieces + ascending high-bit passes.  Bit-exact vs the oracle, out of place and in
    place; 16-bit storage keeps its single rounding (the oracle rounds once, from the f32 result)."""
    d = 1 << log2d
    for rows in (1, 3, 9):
        if rows * d > (1 << 21):
            continue
        x = _rand(rows, d, dtype, seed=77 + log2d + rows)
        want = _oracle(x)
        xd = x.to(DEV)
        got = _hip.fwht_rows(xd)
        assert torch.equal(got.cpu().view(torch.uint8), want.view(torch.uint8)), (dtype, log2d, rows)
        assert torch.equal(xd.cpu().view(torch.uint8), x.view(torch.uint8)), "input must stay untouched"
        _hip.fwht_rows(xd, out=xd)
        assert torch.equal(xd.cpu().view(torch.uint8), want.view(torch.uint8))
    if log2d <= 16:
        assert "fwht_block_rows_kernel" in _hip.last_kernel()


@pytest.mark.parametrize("dtype,log2d", [(torch.float32, 14), (torch.float32, 16), (torch.float64, 15), (torch.float16, 16),
                                         (torch.bfloat16, 15), (torch.int32, 15)])
def test_block_rows_at_streaming_size(dtype, log2d, hip_lib):
    """The same kernel in its streaming instantiation (non-temporal loads, write-through stores, XCD-contiguous block
    order): 320 MiB in place, a row count that is not a multiple of 8 as well; sampled rows against the oracle, the
    rest through the involution H(H(x)) = D x on exactly representable data."""
    d = 1 << log2d
    esize = torch.empty(0, dtype=dtype).element_size()
    for rows in ((320 << 20) // (d * esize), (320 << 20) // (d * esize) + 3):
        g = torch.Generator(device=DEV).manual_seed(5 + log2d)
        x = torch.randint(-3, 4, (rows, d), device=DEV, generator=g, dtype=torch.int32).to(dtype)
        keep = [0, 1, rows // 2, rows - 1]
        want = _oracle(x[keep].cpu())
        y = x.clone()
        _hip.fwht_rows(y, out=y)
        assert "fwht_block_rows_kernel" in _hip.last_kernel() and "true" in _hip.last_kernel()
        assert torch.equal(y[keep].cpu().view(torch.uint8), want.view(torch.uint8)), (dtype, log2d, rows)
        if dtype in (torch.float16, torch.bfloat16):
            continue            # H x of +-3 integers needs more than 8 / 11 bits: no exact involution for 16-bit storage
        _hip.fwht_rows(y, out=y)
        assert torch.equal(y, x * d), (dtype, log2d, rows)
        del y, x


@pytest.mark.parametrize("dtype,log2d", [(torch.float32, 16), (torch.float32, 14), (torch.float64, 13), (torch.int32, 16)])
def test_block_rows_pipelined_cached_form(dtype, log2d, hip_lib):
    """Exactly 256 MiB in place is the one size at which the persistent pipelined grid runs with CACHED accesses (enough
    rows per resident block, not yet a stream): its own instantiation, checked like the streaming one."""
    d = 1 << log2d
    rows = (256 << 20) // (d * torch.empty(0, dtype=dtype).element_size())
    g = torch.Generator(device=DEV).manual_seed(11 + log2d)
    x = torch.randint(-3, 4, (rows, d), device=DEV, generator=g, dtype=torch.int32).to(dtype)
    keep = [0, 1, 255, 256, 257, rows // 2 + 1, rows - 2, rows - 1]     # first, second and last trip of several blocks
    want = _oracle(x[keep].cpu())
    y = x.clone()
    _hip.fwht_rows(y, out=y)
    assert _hip.last_kernel().endswith("false, true>"), _hip.last_kernel()
    assert torch.equal(y[keep].cpu().view(torch.uint8), want.view(torch.uint8)), (dtype, log2d)
    _hip.fwht_rows(y, out=y)
    assert torch.equal(y, x * d), (dtype, log2d)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_16bit_special_values(dtype, hip_lib):
    """Subnormals, signed zeros, the largest finite values (sums overflow to inf when the row is stored), inf and NaN in
    16-bit storage: converts in, f32 butterflies, ONE rounding out -- the same bits as the oracle's f32 transform
    rounded once by torch (NaN compared by position).  Covers the DPP network (small problems), the LDS-staged network
    (streams) and the block-per-row kernel; the fp16 unpack is hand-written (explicit shift form), so its subnormal
    handling is pinned here."""
    info = torch.finfo(dtype)
    specials = torch.tensor([0.0, -0.0, info.tiny, -info.tiny, info.tiny / 4, -info.tiny / 8, info.smallest_normal * 1.5,
                             info.max, -info.max, info.max / 2, 1.0, -1.0, float("inf"), -float("inf"), float("nan"), 3.0],
                            dtype=torch.float32).to(dtype)
    g = torch.Generator().manual_seed(3)

    def check(x, keep=None):
        rows = slice(None) if keep is None else keep
        want = _oracle(x[rows].cpu())
        got = _hip.fwht_rows(x.to(DEV))[rows].cpu()
        nan_w, nan_g = torch.isnan(want.float()), torch.isnan(got.float())
        assert torch.equal(nan_w, nan_g)
        assert torch.equal(got.view(torch.int16)[~nan_g], want.view(torch.int16)[~nan_w])

    for d in (8, 64, 4096, 1 << 14):
        x = (torch.randn(33, d, generator=g) * 0.5).to(dtype)
        x[1, :16 if d >= 16 else d] = specials[:16 if d >= 16 else d]         # everything at once: NaN / inf rows
        x[2] = specials[torch.randint(0, 12, (d,), generator=g)]                # finite specials only: overflow + subnormal sums
        x[3] = specials[torch.randint(2, 7, (d,), generator=g)]                 # subnormal-sized values only
        x[4, ::7] = specials[7]
        check(x)
    d, rows = 4096, (320 << 20) // (4096 * 2) + 5                                # the streaming (LDS-staged) launch
    x = (torch.randn(64, d, generator=g) * 0.5).to(dtype).repeat(rows // 64 + 1, 1)[:rows].contiguous()
    x[5] = specials[torch.randint(0, 12, (d,), generator=g)]
    x[rows - 2] = specials[torch.randint(2, 7, (d,), generator=g)]
    x[rows // 2, :16] = specials
    check(x, keep=[0, 5, rows // 2, rows - 2, rows - 1])
    assert "2, false, true, 256, 1" in _hip.last_kernel(), _hip.last_kernel()   # POLICY_LDS, streaming


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_float_special_values(dtype, hip_lib):
    """f32 / f64 rows holding subnormals, signed zeros, huge values (sums overflow), inf and NaN, through the cached
    and the streaming launches and the block-per-row kernel: NaN in the same places as the oracle, every other value
    equal, and bit-identical wherever the oracle's value is not a zero."""
    info = torch.finfo(dtype)
    specials = torch.tensor([0.0, -0.0, info.tiny, -info.tiny, info.tiny / 4, -info.tiny / 8, info.smallest_normal * 1.5,
                             info.max, -info.max, info.max / 2, 1.0, -1.0, float("inf"), -float("inf"), float("nan"), 3.0],
                            dtype=dtype)
    g = torch.Generator().manual_seed(5)
    ibits = torch.int32 if dtype == torch.float32 else torch.int64

    def check(x, keep=None):
        rows = slice(None) if keep is None else keep
        want = _oracle(x[rows].cpu())
        got = _hip.fwht_rows(x.to(DEV))[rows].cpu()
        nan_w, nan_g = torch.isnan(want), torch.isnan(got)
        assert torch.equal(nan_w, nan_g)
        ok = ~nan_w
        assert torch.equal(got[ok], want[ok])                                   # values (-0 == +0 here)
        nz = ok & (want != 0)
        assert torch.equal(got.view(ibits)[nz], want.view(ibits)[nz])           # bits wherever the value is not a zero

    for d in (8, 64, 2048, 4096, 1 << 14):
        x = torch.randn(33, d, generator=g, dtype=dtype)
        x[1, :16 if d >= 16 else d] = specials[:16 if d >= 16 else d]
        x[2] = specials[torch.randint(0, 12, (d,), generator=g)]
        x[3] = specials[torch.randint(0, 7, (d,), generator=g)]                 # zeros of both signs and subnormal-sized values
        x[4] = -0.0                                                             # a row of negative zeros: the result is -0 everywhere
        check(x)
    d = 2048
    rows = (320 << 20) // (d * specials.element_size()) + 3                     # the streaming launch
    x = torch.randn(64, d, generator=g, dtype=dtype).repeat(rows // 64 + 1, 1)[:rows].contiguous()
    x[5] = specials[torch.randint(0, 12, (d,), generator=g)]
    x[6] = -0.0
    x[rows - 2] = specials[torch.randint(0, 7, (d,), generator=g)]
    x[rows // 2, :16] = specials
    check(x, keep=[0, 5, 6, rows // 2, rows - 2, rows - 1])
    assert "true, 256, 1" in _hip.last_kernel() or "1024" in _hip.last_kernel(), _hip.last_kernel()


def test_long_row_launch_forms_agree(hip_lib):
    """Every launch form of rows longer than one wave tile returns the production launch's bits: round 1's pieces +
    high-bit passes, one row per block, the pipelined grid, and the multi-pass form without row groups -- selected through
    bits 20..22 of ``whvi_fwht_ex``'s variant word (include/whvi_hip.h; the library reads no environment)."""
    forms = {"passes": 1 << 20, "one row per block": 2 << 20, "pipelined": 3 << 20, "ungrouped passes": 4 << 20}
    for dt, l, rows in ((torch.float32, 14, 4100), (torch.float32, 16, 1030), (torch.float64, 13, 4100), (torch.int32, 15, 70),
                        (torch.float32, 18, 5), (torch.float64, 17, 3)):
        g = torch.Generator(device=DEV).manual_seed(l)
        x = torch.randint(-99, 100, (rows, 1 << l), device=DEV, generator=g, dtype=torch.int32).to(dt)
        if dt != torch.int32:
            x = x * 0.37
        want = _hip.fwht_rows(x)
        production = _hip.last_kernel()
        seen = {production}
        for name, variant in forms.items():
            got = _hip.fwht_rows(x, variant=variant)
            seen.add(_hip.last_kernel())
            assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), (dt, l, name)
        if dt == torch.float32 and l <= 16:                 # more rows than resident blocks: both grid forms were really launched
            assert {k[-6:] for k in seen if "block_rows" in k} >= {"false>", " true>"}, seen


def test_half_types_reject_multi_pass_lengths(hip_lib):
    """fp16 / bf16 promise one rounding of the f32 result; a second pass would round the intermediate, so rows
    beyond the one-block limit (D = 65536) are refused instead of silently losing bits."""
    for dt in (torch.float16, torch.bfloat16):
        with pytest.raises(RuntimeError, match="outside the supported range"):
            _hip.fwht_rows(torch.zeros(1, 1 << 17, dtype=dt, device=DEV))


def test_concurrent_host_threads(hip_lib):
    """The C ABI keeps no global mutable state besides a thread-local error string: four host threads, each on
    its own stream, transform their own buffers concurrently."""
    import threading
    d, rows = 1024, 513
    inputs = [_rand(rows, d, torch.int32, 100 + i).float().to(DEV) for i in range(4)]
    outs, errs = [None] * 4, []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                y = inputs[i]
                for _ in range(6):          # three H.H = D.I round trips
                    y = fwht_cuda.fwht(y)
            st.synchronize()
            outs[i] = y
        except Exception as err:            # surfaced below
            errs.append(err)
    torch.cuda.synchronize()
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for i in range(4):
        assert torch.equal(outs[i], inputs[i] * float(d) ** 3)


def test_last_kernel_names_the_launched_instantiation(hip_lib):
    """whvi_last_kernel() reports the demangled symbol the dispatch selected (what bench.py prints as roofline.kernel):
    the cached 256-thread launch for a small problem, the streaming one beyond the Infinity Cache, the fused kernel."""
    x = torch.randn(64, 4096, device=DEV)
    _hip.fwht_rows(x)
    assert _hip.last_kernel() == "whvi::fwht_rows_kernel<float, 12, 16, 0, false, false, 256, 0, false>"
    big = torch.zeros((1 << 29) // 2048 // 4 * 4, 2048, device=DEV)          # 512 MiB in place: streaming launch
    _hip.fwht_rows(big, out=big)
    assert _hip.last_kernel() == "whvi::fwht_rows_kernel<float, 11, 16, 0, false, true, 256, 1, false>"
    _hip.fwht_rows(big, out=big, signed_lanes=True)                         # opt-in: WHVI_FWHT_SIGNED_LANES
    assert _hip.last_kernel() == "whvi::fwht_rows_kernel<float, 11, 16, 0, false, true, 256, 1, true>"
    h = torch.zeros(1 << 14, 4096, device=DEV, dtype=torch.float16)
    _hip.fwht_rows(h, out=h)
    assert _hip.last_kernel().startswith("whvi::fwht_rows_kernel<__half, 12, 8, ")
    _hip.fused_shs(x, torch.ones(4096, device=DEV), None, None, axis="col")
    assert _hip.last_kernel().startswith("whvi::fused_shs_kernel<float, 12, 16, 1, false, ")


def test_bench_line_on_the_gpu(hip_lib):
    """bench.py end to end at a small row count: one JSON line with the contract fields, the roofline object naming
    the kernel that was launched, and a traffic value or an explicit reason why there is none for this shape."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rows", "65536", "--steps", "3", "--warmup", "1",
                          "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["dtype"] == "f32" and rec["value"] > 0
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["kernel"].startswith("whvi::fwht_rows_kernel<float, 12, 16, ")
    assert roof["traffic"] is None and "no PMC record" in roof["traffic_note"]          # only the headline shape has one
    assert rec["config"]["values_finite_after_run"] is True


@pytest.mark.parametrize("dtype,log2d", [(torch.float32, 9), (torch.float32, 10), (torch.float32, 11),
                                         (torch.float64, 6), (torch.float64, 9), (torch.float64, 11)])
def test_signed_streaming_launch_of_f32_and_f64_rows(dtype, log2d, hip_lib):
    """WHVI_FWHT_SIGNED_LANES (opt-in through whvi_fwht_ex): f32 streams of D = 512 .. 2048 and f64 streams of D = 64 .. 2048
    beyond the Infinity Cache take the SIGNED DPP network (one fma per lane-stage element, the tile carrying
    (-1)^popcount(lane & 15) until one repair multiply at the end): 320 MiB in place, random floats and small integers,
    sampled rows bit-identical to the oracle -- and to the unsigned network (a cached out-of-place launch of the same
    rows) -- and H.H = D.I exactly on the integers.  The DEFAULT launch of the same buffer (whvi_fwht_<dtype>) stays on
    the unsigned network, and returns the same bits."""
    d = 1 << log2d
    esize = 4 if dtype == torch.float32 else 8
    name = "float" if dtype == torch.float32 else "double"
    bits = torch.int32 if dtype == torch.float32 else torch.int64
    rows = (320 << 20) // (esize * d) + 3      # + 3: a partial last tile
    g = torch.Generator(device=DEV).manual_seed(log2d)
    idx = torch.cat((torch.tensor([0, 1, 2, 3, rows // 2 + 1, rows - 2, rows - 1]), torch.randint(0, rows, (121,)))).to(DEV)
    for kind in ("randn", "ints"):
        if kind == "randn":
            x = torch.randn(rows, d, device=DEV, generator=g, dtype=dtype)
        else:
            x = torch.randint(-3, 4, (rows, d), device=DEV, generator=g, dtype=torch.int32).to(dtype)
        keep = x[idx].clone()
        plain = _hip.fwht_rows(x)[idx].cpu()                      # the drop-in entry point: unsigned network at every size
        assert _hip.last_kernel().endswith(", false>") and ", false, true, " in _hip.last_kernel(), _hip.last_kernel()
        _hip.fwht_rows(x, out=x, signed_lanes=True)
        assert _hip.last_kernel() == f"whvi::fwht_rows_kernel<{name}, {log2d}, 16, 0, false, true, 256, 1, true>", _hip.last_kernel()
        got = x[idx].cpu()
        assert torch.equal(plain.view(bits), got.view(bits)), kind
        assert torch.equal(got.view(bits), _oracle(keep.cpu()).view(bits)), kind
        small = _hip.fwht_rows(keep)                              # 128 rows: the cached, unsigned launch
        assert _hip.last_kernel().endswith(", 0, false>") and ", false, false, " in _hip.last_kernel()
        assert torch.equal(small.view(bits), got.to(DEV).view(bits))
        if kind == "ints":
            _hip.fwht_rows(x, out=x)
            assert torch.equal(x[idx], keep * d) and torch.equal(x[::4099], x[::4099].round())


def test_negative_zero_contract_of_both_launch_forms(hip_lib):
    """include/whvi_hip.h, "Sign of zero": a row made of negative zeros only has the result [-0, +0, +0, ...] in the
    reference's arithmetic (element 0 is a sum of negative zeros, -0 + -0 = -0; every difference -0 - -0 is +0).  The
    drop-in entry point returns exactly that at EVERY size -- cache-resident launch and stream alike (round 4: one contract
    for zero).  The opt-in WHVI_FWHT_SIGNED_LANES stream of f32 rows of D = 512 .. 2048 returns +0 for
    those rows -- and ONLY that differs: every other row of the same buffer is bit-identical between the two forms."""
    d = 1024
    rows = (320 << 20) // (4 * d)
    x = torch.randn(rows, d, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    x[7] = -0.0
    x[rows - 3] = -0.0
    x[11] = 0.0
    small = _hip.fwht_rows(x[:16])                                  # cache-resident launch, unsigned network
    assert _hip.last_kernel().endswith("256, 0, false>")
    assert bool((small[7] == 0).all()) and bool(torch.signbit(small[7, 0])) and not bool(torch.signbit(small[7, 1:]).any())
    assert not bool(torch.signbit(small[11]).any())
    assert torch.equal(small[7].cpu().view(torch.int32), _oracle(x[7:8].cpu())[0].view(torch.int32)), "reference arithmetic"
    plain = _hip.fwht_rows(x)                                       # 320 MiB out of place: the streaming launch, unsigned network
    assert _hip.last_kernel() == "whvi::fwht_rows_kernel<float, 10, 16, 0, false, true, 256, 1, false>"
    for r in (7, rows - 3):
        assert torch.equal(plain[r].view(torch.int32), small[7].view(torch.int32)), "the reference's bits at every size"
    assert torch.equal(plain[:16].view(torch.int32), small.view(torch.int32))
    del plain
    big = _hip.fwht_rows(x, signed_lanes=True)                      # opt-in: signed streaming launch
    assert _hip.last_kernel() == "whvi::fwht_rows_kernel<float, 10, 16, 0, false, true, 256, 1, true>"
    for r in (7, rows - 3):
        assert bool((big[r] == 0).all()) and not bool(torch.signbit(big[r]).any()), "documented exception: +0"
    other = torch.ones(16, dtype=torch.bool)
    other[7] = False
    assert torch.equal(big[:16][other.to(DEV)].view(torch.int32), small[other.to(DEV)].view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,log2d,K", [(torch.float32, 13, 32), (torch.int32, 13, 32), (torch.float64, 12, 32),
                                           (torch.float16, 13, 16), (torch.bfloat16, 13, 16)])
def test_one_row_tiles_of_128_registers_at_streaming_size(dtype, log2d, K, hip_lib):
    """The widest single-wave rows (f32 / i32 / fp16 / bf16 D = 8192, f64 D = 4096: 128 accumulator registers per lane) as
    STREAMS of 320 MiB with a partial last block -- for f32 the three-waves-per-SIMD kernel of fwht_wide.hip, whose stores
    carry the chunk offset as the scalar offset of a bounds-checked buffer instruction: sampled
    rows bit for bit against the oracle, in place == out of place, H.H = D.I on small integers, the launched symbol, and
    not one byte touched beyond either end of the buffer (the tensor sits in the middle of a sentinel-filled allocation)."""
    d = 1 << log2d
    esize = torch.empty(0, dtype=dtype).element_size()
    rows = ((320 << 20) // (esize * d)) | 1                     # odd: one tile = one row, so the last BLOCK has idle waves
    pad = 4 * d
    g = torch.Generator(device=DEV).manual_seed(log2d + esize)
    ints = torch.randint(-3, 4, (rows, d), device=DEV, generator=g, dtype=torch.int32)
    big = torch.full((rows * d + 2 * pad,), 7, device=DEV, dtype=dtype)
    x = big[pad:pad + rows * d].view(rows, d)
    x.copy_(ints.to(dtype) if dtype == torch.int32 else (torch.randn(rows, d, device=DEV, generator=g) * 0.25).to(dtype))
    idx = torch.cat((torch.tensor([0, 1, 2, 3, rows // 2, rows - 2, rows - 1]), torch.randint(0, rows, (57,)))).to(DEV)
    keep = x[idx].clone()
    out = _hip.fwht_rows(x)                                      # out of place
    name = {torch.float32: "float", torch.int32: "int", torch.float64: "double", torch.float16: "__half",
            torch.bfloat16: "__hip_bfloat16"}[dtype]
    assert _hip.last_kernel() == f"whvi::fwht_rows_kernel<{name}, {log2d}, {K}, 0, false, true, 256, 1, false>", _hip.last_kernel()
    _hip.fwht_rows(x, out=x)                                     # in place
    assert torch.equal(x.view(torch.uint8), out.view(torch.uint8))
    assert torch.equal(x[idx].cpu().view(torch.uint8), _oracle(keep.cpu()).view(torch.uint8))
    sentinel = torch.full((pad,), 7, device=DEV, dtype=dtype)
    assert torch.equal(big[:pad].view(torch.uint8), sentinel.view(torch.uint8))
    assert torch.equal(big[pad + rows * d:].view(torch.uint8), sentinel.view(torch.uint8))
    if dtype in (torch.int32, torch.float64):                    # exact there at any magnitude the sums reach: an involution up to D
        x.copy_(ints.to(dtype))
        _hip.fwht_rows(x, out=x)
        _hip.fwht_rows(x, out=x)
        assert torch.equal(x[idx].to(torch.int64), ints[idx].to(torch.int64) * d)
        assert torch.equal(x[-1].to(torch.int64), ints[-1].to(torch.int64) * d)


def test_config1_shape_on_the_gpu_box(hip_lib):
    """BASELINE config 1 at exactly its shape -- (1024, 512) fp32, benchmarks/walsh.py:14-21 -- on the GPU box: the device
    path (``fwht_cuda.fwht`` -> whvi_fwht_f32) AND the host library (``fwht_cpp.forward`` -> libwhvi_cpu.so, checked on THIS
    box's CPU) both bit-equal to the C oracle (restatement of src/fwht/cpp/fwht.cpp:3-21) and, where its binary travelled,
    to the reference's own compiled FWHT (oracle/_ref)."""
    import fwht_cpp
    import fwht_cuda
    g = torch.Generator().manual_seed(1024 * 512)
    x = torch.randn(1024, 512, generator=g)
    want = oracle.fwht(x.numpy())
    dev = fwht_cuda.fwht(x.to(DEV))
    assert dev.shape == (1024, 512) and dev.dtype == torch.float32
    assert np.array_equal(dev.cpu().numpy().view(np.uint32), want.view(np.uint32))
    host = fwht_cpp.forward(x)
    assert np.array_equal(host.numpy().view(np.uint32), want.view(np.uint32))
    ref = oracle.load_reference_cpp()
    if ref is not None:
        assert np.array_equal(ref.forward(x).numpy().view(np.uint32), want.view(np.uint32))
    # integer-valued input: exact on every path (north_star: bit-exact on integer inputs)
    xi = torch.randint(-8, 8, (1024, 512), generator=g)
    wi = oracle.fwht(xi.to(torch.int32).numpy())
    assert np.array_equal(fwht_cuda.fwht(xi.float().to(DEV)).cpu().numpy(), wi.astype(np.float32))
    assert np.array_equal(fwht_cpp.forward(xi.float()).numpy(), wi.astype(np.float32))
