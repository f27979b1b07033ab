"""WHVI weight matrices (mirror of the reference's src/weights.py).

Public surface, parameter names/shapes/initialisation and numerics follow the reference so that
``state_dict``s interchange and ``WHVILinear`` / ``WHVIRegression`` consume these classes
unchanged:

    WHVISquarePow2Matrix  src/weights.py:13-108
    WHVIStackedMatrix     src/weights.py:111-208
    WHVIColumnMatrix      src/weights.py:211-251

Dataflow note (SURVEY.md finding 1).  ``fwht`` transforms ROWS and ``matmul_diag_left`` scales
ROWS, so ``w_bar(u) = S1 . fwht(diag(u) . fwht(diag(s2)))`` is, as written in the reference,
``D * diag(s1 * u * s2)``.  Parity is judged against the reference as written, therefore this
module reproduces that row-scaling dataflow and does not "fix" it.

Device dispatch (src/weights.py:34-41): GPU tensors run on the MI355X HIP kernels -- the whole
scale -> FWHT -> scale -> FWHT -> scale chain of ``w_bar`` is ONE kernel launch that synthesises
``diag(s2)`` in registers (``whvi_fused_shs_f32``, include/whvi_hip.h) -- and host tensors run
the same torch ops as the reference (dense H for D < 4096, vectorised butterflies otherwise).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from whvi_amd.utils import matmul_diag_left, kl_diag_normal
from whvi_amd.fwht.cuda import FWHTFunction as fwht_cuda
from whvi_amd.fwht.python import FWHTFunction as fwht_python
from whvi_amd.fwht.python import WHT_matmul as wht_matmul

__all__ = ["WHVISquarePow2Matrix", "WHVIStackedMatrix", "WHVIColumnMatrix", "WBarFunction", "ReparamKLFunction",
           "ReparamKLPhiloxFunction", "DiagApplyFunction", "SmallKApplyFunction", "RowDotFunction"]


class WBarFunction(torch.autograd.Function):
    """Batched ``w_bar``: for J independent square matrices with S ``u`` vectors each,

        W[j, k] = diag(s1[j]) . fwht(diag(u[j, k]) . fwht(diag(s2[j])))[:rows]

    as ONE HIP launch producing ``(J, S, rows, D)``.  ``s1, s2``: (J, D); ``u``: (J, S, D).
    ``rows < D`` keeps only the first rows of every matrix (all that WHVIColumnMatrix uses,
    src/weights.py:245); row i depends on ``s1[j,i], u[j,k,i], s2[j,i]`` alone.
    ``mean_plus``: ``u`` is ``(J, 1 + S, D)`` and the result is ``W[j, 0] + W[j, 1 + k]`` for k < S -- the
    ``w_bar(g_mu) + w_bar(g_sigma * eps_k)`` of src/weights.py:93 -- built as two launches (the mean matrix once,
    then every sample's matrix with the mean added in the epilogue) instead of 1 + S matrices and an add pass.

    Forward: ``whvi_wbar_fwd`` (no HBM read; multiply / butterfly order and roundings exactly those of
    src/weights.py:73, bit-identical to ``whvi_fused_shs_ex`` with the identity input).  Backward: the adjoint chain (H is symmetric, so every FWHT's adjoint is
    the same FWHT, src/fwht/cuda/fwht.py:14-16): one fused launch (``whvi_wbar_bwd``) for first-order
    gradients, the chain written with differentiable ops when a graph of the backward is requested."""

    @staticmethod
    def forward(ctx, s1, u, s2, rows, mean_plus=False):
        from whvi_amd import _hip
        J, U, D = u.shape
        R = D if rows is None else int(rows)
        ctx.save_for_backward(s1, u, s2)
        ctx.rows, ctx.mean_plus = R, bool(mean_plus)
        if _hip.wbar_bwd_supported(u.dtype, D):
            # dedicated launch (whvi_wbar_fwd): one transform per row, no HBM read
            if not mean_plus:
                return _hip.wbar_fwd(s1, u, s2, R)
            # w_bar(g_mu) + w_bar(g_sigma * eps_k): one launch computing both terms while the result is cache-resident,
            # the mean matrix once + every sample with the mean added in its epilogue beyond (same bits)
            return _hip.wbar_fwd_mean(s1, u, s2, R)
        # rows shorter than one 16-byte chunk (D = 1, 2): the generic fused launch with the identity input.  Rows
        # of each (j, k) matrix form one group of R rows; with several matrices every group carries its own s1 / s2
        if J == 1:
            out = _hip.fused_shs(None, a=s1[0, :R], b=u[0, :, :R], c=s2[0, :R], axis="row", n_samples=U,
                                 sample_stride=R, group_rows=R, rows=U * R, d=D, dtype=u.dtype, device=u.device)
        else:
            out = _hip.fused_shs(None, a=s1[:, :R].repeat_interleave(U, dim=0), b=u[:, :, :R],
                                 c=s2[:, :R].repeat_interleave(U, dim=0), axis="row", n_samples=J * U,
                                 sample_stride=R, group_rows=R, rows=J * U * R, d=D, dtype=u.dtype,
                                 device=u.device, a_per_sample=True, c_per_sample=True)
        out = out.view(J, U, R, D)
        return out[:, :1] + out[:, 1:] if mean_plus else out

    @staticmethod
    def backward(ctx, grad_W):
        s1, u, s2 = ctx.saved_tensors
        J, U, D = u.shape
        R = ctx.rows
        from whvi_amd import _hip
        if not torch.is_grad_enabled() and _hip.wbar_bwd_supported(u.dtype, D):
            # first-order backward (the training loop): one launch, grad_W read once (whvi_wbar_bwd)
            out = _hip.wbar_bwd(grad_W, s1, u, s2, mean=ctx.mean_plus)   # (3, J, U, D): grad_u, per-sample s1 / s2 parts
            if ctx.mean_plus:
                # slot 0 <- sum of the sample slots: dL/du_mean and the per-matrix totals of s1 / s2, one reduction
                torch.sum(out[:, :, 1:], dim=2, out=out[:, :, 0])
                return out[1, :, 0], out[0], out[2, :, 0], None, None
            parts = out[1:, :, 0] if U == 1 else out[1:].sum(dim=2)     # one reduction for both scale vectors
            return parts[0], out[0], parts[1], None, None
        # create_graph=True (or a shape outside the fused kernel): the same chain as differentiable ops
        if ctx.mean_plus:                        # W[j,k] = w_bar(u_0) + w_bar(u_{1+k}): the mean takes the summed gradient
            grad_W = torch.cat((grad_W.sum(dim=1, keepdim=True), grad_W), dim=1)
        S = U
        fw = fwht_cuda.apply
        with torch.enable_grad():
            s1r, s2r, ur = s1[:, :R], s2[:, :R], u[:, :, :R]
            eye = torch.eye(D, dtype=u.dtype, device=u.device)[:R]                     # rows of I
            t1 = fw((s2r.unsqueeze(-1) * eye).reshape(J * R, D)).view(J, 1, R, D)      # fwht(diag(s2))
            g2 = s1r.view(J, 1, R, 1) * grad_W                                         # adjoint of s1 scaling
            g1 = fw(g2.reshape(J * S * R, D)).view(J, S, R, D)                         # adjoint of outer FWHT
            grad_u_r = (g1 * t1).sum(dim=3)                                            # (J, S, R)
            gx = fw((ur.unsqueeze(-1) * g1).reshape(J * S * R, D)).view(J, S, R, D)
            grad_s2_r = torch.diagonal(gx, dim1=2, dim2=3).sum(dim=1)                  # input was diag(s2)
            t2 = fw((ur.unsqueeze(-1) * t1).reshape(J * S * R, D)).view(J, S, R, D)    # pre-s1 tensor
            grad_s1_r = (grad_W * t2).sum(dim=(1, 3))
            if R == D:
                grad_s1, grad_u, grad_s2 = grad_s1_r, grad_u_r, grad_s2_r
            else:
                pad = (0, D - R)
                grad_s1, grad_u, grad_s2 = (F.pad(grad_s1_r, pad), F.pad(grad_u_r, pad), F.pad(grad_s2_r, pad))
        return grad_s1, grad_u, grad_s2, None, None


class DiagApplyFunction(torch.autograd.Function):
    """``out[k] = x[(k)] @ (w_bar(u[0]) + w_bar(u[1 + k])).T (+ bias)`` for every MC sample k in ONE HIP launch that never
    builds the matrices (``whvi_diag_apply``, whvi_amd/csrc/diag_apply.hpp).

    As written in the reference ``w_bar(u)`` is exactly ``D * diag(s1 * u * s2)`` (SURVEY.md finding 1), so the dense
    product of src/weights.py:93 adds exact zeros to one product per output.  The kernel computes that product with the
    roundings of the as-written chain in the same order -- ``u * s2``, ``D * .`` (exact), ``s1 * .``, mean + sample,
    ``h * w``, ``+ bias`` -- hence the same values as weight construction + GEMM for every input, non-finite ones
    included (a row of ``x`` with an inf / NaN turns its other outputs into NaN like the dot products with W's exact
    zeros do) and zeros too (the product is added to the +0 a GEMM's accumulator holds): bit-identical to the matrix route
    on MI355X / rocBLAS (tests/test_diag_apply_gpu.py).

    ``x``: (S, B, D) or a shared (B, D); ``u``: (1 + S, D) with ``mean_plus`` (the buffer the reparameterisation kernel
    writes) else (S, D); ``s1, s2``: (D,); ``bias``: D elements or None.  Backward: one call (``whvi_diag_apply_bwd``:
    grad_x = g * w and the batch reduction sum_b g * x in the same pass, a tiny finishing launch) plus one reduction
    over the samples; with ``create_graph`` the same expression as differentiable torch ops."""

    @staticmethod
    def forward(ctx, x, s1, s2, u, bias, n_samples, mean_plus, relu_in=False, relu_out=False):
        from whvi_amd import _hip
        ctx.n_samples, ctx.mean_plus = int(n_samples), bool(mean_plus)
        ctx.relu_in, ctx.relu_out = bool(relu_in), bool(relu_out)
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        # (the fused activations' backward needs no saved output: their masks are recomputed from x, the diagonal and the bias)
        ctx.save_for_backward(x, s1, s2, u, bias if (bias is not None and relu_out) else None)
        return _hip.diag_apply(x, s1, s2, u, bias, n_samples=n_samples, mean_plus=mean_plus, relu_in=relu_in, relu_out=relu_out)

    @staticmethod
    def _reference_ops(x, s1, s2, u, bias, mean_plus, relu_in=False, relu_out=False):
        """The same expression as differentiable torch ops (finite operands): double backward, shapes the kernel lacks."""
        D = float(s1.shape[0])
        w = s1 * (D * (u * s2))                                   # (U, D): diag(w_bar(u_r)), the reference's roundings
        if mean_plus:
            w = _mean_plus_rest(w, 0)                             # (S, D)
        if relu_in:
            x = torch.relu(x)
        out = (x if x.dim() == 3 else x.unsqueeze(0)) * w.unsqueeze(1)
        out = out + bias.reshape(-1) if bias is not None else out
        return torch.relu(out) if relu_out else out

    @staticmethod
    def backward(ctx, grad_out):
        from whvi_amd import _hip
        x, s1, s2, u, bias = ctx.saved_tensors
        S, mean_plus = ctx.n_samples, ctx.mean_plus
        need = ctx.needs_input_grad
        if torch.is_grad_enabled():
            # create_graph=True: the first-order gradients as differentiable functions of (grad_out, x, s1, s2, u)
            D = float(s1.shape[0])
            w = s1 * (D * (u * s2))
            w_k = _mean_plus_rest(w, 0) if mean_plus else w
            xa = torch.relu(x) if ctx.relu_in else x
            x3 = xa if xa.dim() == 3 else xa.unsqueeze(0)
            if ctx.relu_out:
                z = x3 * w_k.unsqueeze(1)
                if bias is not None:
                    z = z + bias.reshape(-1)
                grad_out = grad_out * (z > 0).to(grad_out.dtype)
            gx = grad_out * w_k.unsqueeze(1)
            if ctx.relu_in:
                gx = gx * (x3 > 0).to(gx.dtype)
            if x.dim() == 2:
                gx = gx.sum(dim=0)
            gw = (grad_out * x3).sum(dim=1)                                                  # (S, D)
            gw_u = torch.cat((gw.sum(dim=0, keepdim=True), gw), dim=0) if mean_plus else gw      # d w_k / d w_r
            grad_u = gw_u * (s1 * D * s2)
            grad_s1 = (gw_u * (D * (u * s2))).sum(dim=0)
            grad_s2 = (gw_u * (s1 * D * u)).sum(dim=0)
            grad_bias = grad_out.sum(dim=(0, 1)).reshape(ctx.bias_shape) if ctx.bias_shape is not None else None
            return gx, grad_s1, grad_s2, grad_u, grad_bias, None, None, None, None
        gx, out = _hip.diag_apply_bwd(grad_out, x, s1, s2, u, n_samples=S, mean_plus=mean_plus, need_grad_x=need[0],
                                      bias=bias, relu_in=ctx.relu_in, relu_out=ctx.relu_out)
        if mean_plus:
            # row 0 <- sum of the sample rows: dL/du_mean and the totals of s1 / s2 / bias, one reduction for all four
            torch.sum(out[:, 1:], dim=1, out=out[:, 0])
            grad_u, tot = out[0], out[1:, 0]
        else:
            grad_u, tot = out[0], (out[1:, 0] if S == 1 else out[1:].sum(dim=1))
        if gx is not None and x.dim() == 2:
            gx = gx.sum(dim=0)
        grad_bias = tot[2].reshape(ctx.bias_shape) if ctx.bias_shape is not None else None
        return gx, tot[0], tot[1], grad_u, grad_bias, None, None, None, None


class SmallKApplyFunction(torch.autograd.Function):
    """``out[s] = x @ W[s].T`` for a narrow input ``x`` (B, K), K in {4, 8}, shared by all samples, ``W`` (S, N, K): the dense
    product of a stacked layer's batched pass (src/weights.py:179-180,195-206) as ONE write-only launch
    (``whvi_small_k_apply_f32``) instead of a batched GEMM with K = 4.  Backward: the two matrix products of the product rule
    as torch ops (``grad_x = sum_s g[s] @ W[s]``, ``grad_W[s] = g[s].T @ x``)."""

    @staticmethod
    def forward(ctx, x, W, bias=None, relu_out=False):
        from whvi_amd import _hip
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        out = _hip.small_k_apply(x, W, bias, relu_out=relu_out)
        ctx.relu_out = bool(relu_out)
        ctx.save_for_backward(x, W, out if relu_out else None)          # (the activation's mask comes from its own output)
        return out

    @staticmethod
    def backward(ctx, g):
        x, W, out = ctx.saved_tensors
        if ctx.relu_out:
            g = g * (out > 0).to(g.dtype)
        grad_x = torch.matmul(g, W).sum(dim=0) if ctx.needs_input_grad[0] else None
        grad_W = torch.matmul(g.transpose(1, 2), x) if ctx.needs_input_grad[1] else None
        grad_bias = g.sum(dim=(0, 1)).reshape(ctx.bias_shape) if (ctx.bias_shape is not None and ctx.needs_input_grad[2]) else None
        return grad_x, grad_W, grad_bias, None


class RowDotFunction(torch.autograd.Function):
    """``y[s, b] = x[s, b, :] . w[s]``: ``F.linear(x, w[None])`` of the transposed column layer for all samples
    (src/weights.py:239-251) as ONE read-only launch (``whvi_row_dot_f32``).  Backward as torch ops."""

    @staticmethod
    def forward(ctx, x, w, relu_in=False):
        from whvi_amd import _hip
        ctx.save_for_backward(x, w)
        ctx.relu_in = bool(relu_in)
        return _hip.row_dot(x, w, relu_in=relu_in)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        grad_x = grad_w = None
        if ctx.needs_input_grad[0]:
            grad_x = g * w.unsqueeze(1)
            if ctx.relu_in:
                grad_x = grad_x * (x > 0).to(g.dtype)
        if ctx.needs_input_grad[1]:
            grad_w = torch.matmul((torch.relu(x) if ctx.relu_in else x).transpose(1, 2), g).squeeze(-1)
        return grad_x, grad_w, None


class ReparamKLFunction(torch.autograd.Function):
    """``(u, kl) = ReparamKL(g_mu (J, D), g_rho (J, D), eps (J, S, D), lambda)`` in one HIP launch:
    ``u[:, 0] = g_mu``, ``u[:, 1 + k] = softplus(g_rho) * eps[:, k]`` -- the ``b`` operand of the fused weight
    kernel -- and ``kl[j] = kl_diag_normal(g_mu[j], softplus(g_rho[j]), 0, lambda)`` with the reference's argument
    convention (src/weights.py:52-64, src/utils.py:49-71).  Backward: closed form, one launch (``whvi_reparam_kl_bwd``)."""

    @staticmethod
    def forward(ctx, g_mu, g_rho, eps, lambda_):
        from whvi_amd import _hip
        u, sigma, kl = _hip.reparam_kl(g_mu, g_rho, eps, lambda_)
        ctx.save_for_backward(g_mu, g_rho, eps, sigma)
        ctx.lambda_ = float(lambda_)
        return u, kl

    @staticmethod
    def backward(ctx, grad_u, grad_kl):
        from whvi_amd import _hip
        g_mu, g_rho, eps, sigma = ctx.saved_tensors
        lam = ctx.lambda_
        grad_eps = grad_u[:, 1:] * sigma.unsqueeze(1) if ctx.needs_input_grad[2] and grad_u is not None else None
        if not torch.is_grad_enabled():
            grad_mu, grad_rho = _hip.reparam_kl_bwd(grad_u, grad_kl, g_mu, g_rho, eps, sigma, lam)
            return grad_mu, grad_rho, grad_eps, None
        # create_graph=True: the same closed form as differentiable ops.  sigma is recomputed from g_rho: the saved
        # one is a kernel output without a grad_fn, and both 1 / sigma below and grad_eps = grad_u * sigma must
        # stay differentiable with respect to g_rho for second derivatives (Hessians, gradient penalties)
        sigma = F.softplus(g_rho)
        if ctx.needs_input_grad[2] and grad_u is not None:
            grad_eps = grad_u[:, 1:] * sigma.unsqueeze(1)
        gk = (torch.zeros_like(g_mu[:, 0]) if grad_kl is None else grad_kl).unsqueeze(-1)
        gu = torch.zeros_like(g_mu).unsqueeze(1).expand(-1, eps.shape[1] + 1, -1) if grad_u is None else grad_u
        grad_mu = gu[:, 0] + gk * (g_mu / lam)
        grad_sigma = (gu[:, 1:] * eps).sum(dim=1) + gk * (0.5 * (1.0 / lam - 1.0 / sigma))
        return grad_mu, grad_sigma * torch.sigmoid(g_rho), grad_eps, None


class ReparamKLPhiloxFunction(torch.autograd.Function):
    """``ReparamKLFunction`` with eps drawn inside the kernel (``whvi_reparam_kl_philox_f32``: Philox4x32-10 +
    Box-Muller, generator state in device memory, hipGraph-safe).  Same outputs, same one-launch backward."""

    @staticmethod
    def forward(ctx, g_mu, g_rho, state, n_samples, lambda_):
        from whvi_amd import _hip
        u, sigma, kl, eps = _hip.reparam_kl_philox(g_mu, g_rho, n_samples, lambda_, state)
        ctx.save_for_backward(g_mu, g_rho, eps, sigma)
        ctx.lambda_ = float(lambda_)
        return u, kl

    @staticmethod
    def backward(ctx, grad_u, grad_kl):
        grad_mu, grad_rho, _, _ = ReparamKLFunction.backward(ctx, grad_u, grad_kl)   # (input 2 = the state: no grad)
        return grad_mu, grad_rho, None, None, None


def _mean_plus_rest(t, dim):
    """``t[0:1] + t[1:]`` along ``dim`` (the ``w_bar(g_mu) + w_bar(g_sigma * eps_k)`` sum of src/weights.py:93), written
    with ``split``: the same additions, but its backward is one ``cat`` instead of two zero-filled scatter copies
    and an add over tensors the size of all weight matrices."""
    mean, rest = t.split((1, t.size(dim) - 1), dim=dim)
    return mean + rest


def _reparam(g_mu, g_rho, eps, lambda_):
    """(u (J, 1+S, D), kl (J,) or None): one fused launch on the GPU when a loss is being built (autograd
    on: the KL of this very pass comes for free and its backward is closed-form); the reference's op chain on
    the host and for pure inference (three tiny launches: less host overhead than the custom Function, and -- measured
    again in round 4 -- than a direct call of the kernel's ctypes wrapper: toy network's eager 64-sample pass 0.195 vs 0.236 ms;
    a hipGraph replay of the same pass gains from the single launch, 0.075 -> 0.062 ms: taken while a capture is recording)."""
    if g_mu.device.type == "cuda" and g_mu.dtype == torch.float32:
        if torch.is_grad_enabled():
            return ReparamKLFunction.apply(g_mu, g_rho, eps, lambda_)
        if torch.cuda.is_current_stream_capturing():
            # inference pass being recorded into a hipGraph (GraphedPredictor): host overhead is paid once, launches on every
            # replay -- the one-launch kernel called directly
            from whvi_amd import _hip
            u, _, kl = _hip.reparam_kl(g_mu, g_rho, eps, lambda_)
            return u, kl
    sigma = F.softplus(g_rho)
    return torch.cat((g_mu.unsqueeze(1), sigma.unsqueeze(1) * eps), dim=1), None


def _posterior_kl(g_mu, g_rho, lambda_):
    """``sum_j kl_diag_normal(g_mu[j], softplus(g_rho[j]), 0, lambda)`` for ``(J, D)`` parameters -- the ``kl``
    property of every weight flavour (src/weights.py:52-64, :169).  float32 on the GPU: ONE launch of the
    reparameterisation + KL kernel with zero samples (differentiable through its closed-form backward) instead of the
    formula's ten tiny launches; otherwise the reference's formula (src/utils.py:49-71) as torch ops.

    Tolerance contract of the GPU float32 value: the same D terms as the reference's formula, added per term and then
    over 256-element blocks instead of as five separate D-term sums -- so it agrees with ``kl_diag_normal`` to float32
    summation noise (<= 1e-5 relative; pinned at 2e-6 by tests/test_fused_gpu.py::test_kl_property_tolerance_contract),
    not bit for bit; ``lambda_ <= 0`` raises where the formula would return NaN.  float64 parameters and host tensors
    evaluate the formula itself."""
    if g_mu.device.type == "cuda" and g_mu.dtype == torch.float32:
        eps = g_mu.new_empty((g_mu.shape[0], 0, g_mu.shape[1]))
        kl = ReparamKLFunction.apply(g_mu, g_rho, eps, lambda_)[1]
        return kl[0] if kl.numel() == 1 else kl.sum()
    mu, sd = g_mu.reshape(-1), F.softplus(g_rho).reshape(-1)
    return kl_diag_normal(mu, sd, torch.zeros_like(mu), torch.ones_like(mu) * lambda_)


def _fresh_philox_seed():
    """Seed of a layer's in-kernel generator: drawn from torch's default CPU generator (so ``torch.manual_seed`` makes
    runs repeatable) and, inside a ``torch.distributed`` job, mixed with the RANK -- ranks usually share one
    ``manual_seed`` so that their initial parameters agree, and would otherwise all draw the same eps, turning the
    gathered Monte-Carlo samples into ``world`` copies of one shard."""
    import torch.distributed as dist
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    if dist.is_available() and dist.is_initialized():
        seed = (seed + 0x9E3779B97F4A7C15 * (dist.get_rank() + 1)) % (2 ** 62)
    return seed


def _draw_and_reparam(module, g_mu, g_rho, n_samples, lambda_):
    """(u (J, 1+S, D), kl) for a batched MC pass.  Default: one ``torch.randn(J, S, D)`` draw (the stream the
    parity tests replay) followed by ``_reparam``.  With ``module.inkernel_rng`` on a GPU the draw happens inside
    the reparameterisation kernel: one launch, graph-safe, this library's own Philox stream."""
    J, D = g_mu.shape
    if module.inkernel_rng and g_mu.device.type == "cuda" and g_mu.dtype == torch.float32:
        from whvi_amd import _hip
        state = getattr(module, "_rng_state", None)
        if state is None or state.device != g_mu.device:
            state = module._rng_state = _hip.new_rng_state(g_mu.device, seed=_fresh_philox_seed())
        return ReparamKLPhiloxFunction.apply(g_mu, g_rho, state, n_samples, lambda_)
    static = getattr(module, "_eps_static", None)
    if static is None:
        eps = torch.randn(J, n_samples, D, device=g_mu.device)
    else:
        # GraphedTrainStep(static_eps=True): the layer's draws come from a static buffer the caller refills between
        # replays (recorded trajectories through a captured step); allocated -- with a real draw -- on first use
        if static is True or tuple(static.shape) != (J, n_samples, D) or static.device != g_mu.device:
            static = module._eps_static = torch.randn(J, n_samples, D, device=g_mu.device)
        eps = static
    return _reparam(g_mu, g_rho, eps, lambda_)


class WHVISquarePow2Matrix(nn.Module):
    inkernel_rng = False      # opt-in: draw eps inside the reparameterisation kernel (batched MC passes on the GPU)
    default_exploit_diagonal = "auto"      # see __init__; tests flip it to run every golden check on both routes

    def __init__(self, D, lambda_=1e-5, bias=False):
        """Square (D, D) WHVI matrix, D a power of two (src/weights.py:14-32).

        :param int D: number of rows/columns.
        :param float lambda_: prior variance.
        :param boolean bias: add a (non-variational) bias row vector in ``forward``.
        """
        super().__init__()
        self.D = D
        self.lambda_ = lambda_
        self.padding = 0  # interface parity with the stacked matrix
        # How `h @ W.T` is evaluated.  As written in the reference, w_bar(u) is EXACTLY D * diag(s1 * u * s2)
        # (SURVEY.md finding 1; with butterflies the off-diagonals are exact zeros), so the dense product equals the
        # elementwise h * diag(W) with the same roundings.
        #   "auto" (default): GPU tensors take ONE launch per layer call for all MC samples that applies the diagonal
        #       (DiagApplyFunction -- same values as the matrix route, non-finite inputs included, no D x D matrices, no
        #       GEMM); host tensors keep the reference's dataflow;
        #   False (= ``faithful_dataflow = True``): the as-written route everywhere -- weight construction through the
        #       FWHT kernels + dense GEMM;
        #   True: the diagonal everywhere (torch ops on the host).
        self.exploit_diagonal = None      # None = the class default below
        self.wht_slow = wht_matmul()  # dense-H transform for small host matrices; H built lazily

        # creation order = the reference's RNG consumption order (bias, s1, s2, g_mu, g_rho)
        self.bias = nn.Parameter(torch.zeros(1, D)) if bias else None
        self.s1 = nn.Parameter(torch.randn(D) * 0.01)
        self.s2 = nn.Parameter(torch.randn(D) * 0.01)
        self.g_mu = nn.Parameter(torch.zeros(D))
        self.g_rho = nn.Parameter(torch.rand(D) - 3)

    def fwht(self, x):
        """Row FWHT with the reference's dispatch rule (src/weights.py:34-41)."""
        if x.device.type == "cuda":
            return fwht_cuda.apply(x)
        if self.D < 2 ** 12:
            return self.wht_slow.apply(x)
        return fwht_python.apply(x)

    @property
    def g_sigma(self):
        """Standard deviations of the variational posterior over g, softplus(g_rho)
        (src/weights.py:43-50)."""
        return F.softplus(self.g_rho)

    @property
    def kl(self):
        """KL from the N(0, lambda I) prior to the posterior, via the reference's formula and
        argument convention (src/weights.py:52-64)."""
        return _posterior_kl(self.g_mu.unsqueeze(0), self.g_rho.unsqueeze(0), self.lambda_)

    def _w_bar_stack(self, u, rows=None, mean_plus=False):
        """``w_bar`` for every row of ``u`` (S, D) -> (S, D, D) (first ``rows`` rows on the GPU)."""
        if u.device.type == "cuda":
            return WBarFunction.apply(self.s1.unsqueeze(0), u.unsqueeze(0), self.s2.unsqueeze(0), rows,
                                      mean_plus).squeeze(0)
        base = self.fwht(torch.diag(self.s2))
        return torch.stack([matmul_diag_left(self.s1, self.fwht(matmul_diag_left(row, base)))
                            for row in u])

    def w_bar(self, u):
        """``S1 . fwht(diag(u) . fwht(diag(s2)))`` (src/weights.py:66-73)."""
        if u.device.type == "cuda":
            return self._w_bar_stack(u.unsqueeze(0)).squeeze(0)
        return matmul_diag_left(self.s1, self.fwht(matmul_diag_left(u, self.fwht(torch.diag(self.s2)))))

    def sample(self, rows=None):
        """Draw W with g ~ N(g_mu, g_sigma^2) (src/weights.py:75-85).  ``rows`` (GPU only) asks
        for the first rows of W alone -- what the column layer consumes."""
        epsilon = torch.randn(self.D, device=self.g_mu.device)
        g_tilde = self.g_mu + self.g_sigma * epsilon
        if rows is not None and g_tilde.device.type == "cuda":
            return self._w_bar_stack(g_tilde.unsqueeze(0), rows=rows).squeeze(0)
        return self.w_bar(g_tilde)

    def _w_bar_diagonal(self, u):
        """diag(w_bar(u)) with the reference's roundings: s2 * 1 (exact), u * ., the two transforms of a
        one-hot row (exact: D * .), s1 * ."""
        return self.s1 * (float(self.D) * (u * self.s2))

    @property
    def faithful_dataflow(self):
        """True = the as-written route (weight construction + dense GEMM) on every device."""
        return self._diag_mode() is False

    @faithful_dataflow.setter
    def faithful_dataflow(self, value):
        self.exploit_diagonal = False if value else "auto"

    def _diag_mode(self):
        mode = self.exploit_diagonal
        return type(self).default_exploit_diagonal if mode is None else mode

    def _diag_route(self, x):
        """"kernel" (one HIP launch), "ops" (the diagonal as torch ops) or None (the as-written matrix route)."""
        mode = self._diag_mode()
        if mode is False:
            return None
        if x.device.type == "cuda" and x.dtype == self.g_mu.dtype:
            from whvi_amd import _hip
            if _hip.diag_apply_supported(x.dtype, self.D):
                return "kernel"
        return "ops" if mode is True else None

    def _diag_kernel(self, x, u, bias, n_samples, mean_plus, relu_in=False, relu_out=False):
        """x: (..., D) shared by all samples when ``n_samples`` is None (one sample), else (S, B, D) / (B, D)."""
        if not torch.is_grad_enabled():          # inference: the launch itself, without an autograd Function around it
            from whvi_amd import _hip
            if n_samples is None:
                return _hip.diag_apply(x.reshape(1, -1, self.D), self.s1, self.s2, u, bias, n_samples=1, mean_plus=mean_plus).view(x.shape)
            return _hip.diag_apply(x, self.s1, self.s2, u, bias, n_samples=n_samples, mean_plus=mean_plus, relu_in=relu_in,
                                   relu_out=relu_out)
        if n_samples is None:
            out = DiagApplyFunction.apply(x.reshape(1, -1, self.D), self.s1, self.s2, u, bias, 1, mean_plus)
            return out.view(x.shape)
        return DiagApplyFunction.apply(x, self.s1, self.s2, u, bias, n_samples, mean_plus, relu_in, relu_out)

    def fuses_relu(self, x):
        """True when ``forward_mc(x, ..., relu_in=, relu_out=)`` folds the activations into its one launch (the GPU's diagonal
        route); otherwise they are applied as separate ``torch.relu`` passes -- same values either way."""
        return self._diag_route(x) == "kernel"

    def sample_lrt(self, h, _bias=None):
        """``h @ (w_bar(g_mu) + w_bar(g_sigma * eps)).T`` with one eps per call
        (src/weights.py:87-93).  On the GPU: one launch that applies the diagonal both matrices have, or (faithful
        dataflow) both ``w_bar`` matrices from a single launch and the dense product."""
        epsilon = torch.randn(self.D, device=self.g_mu.device)
        route = self._diag_route(h)
        if route == "kernel":
            return self._diag_kernel(h, torch.stack((self.g_mu, self.g_sigma * epsilon)), _bias, None, True)
        if route == "ops":
            out = h * (self._w_bar_diagonal(self.g_mu) + self._w_bar_diagonal(self.g_sigma * epsilon))
        elif self.g_mu.device.type == "cuda":
            W = self._w_bar_stack(torch.stack((self.g_mu, self.g_sigma * epsilon)), mean_plus=True)   # (1, D, D)
            out = h @ W.squeeze(0).T
        else:
            out = h @ (self.w_bar(self.g_mu) + self.w_bar(self.g_sigma * epsilon)).T
        return out + _bias if _bias is not None else out

    def forward(self, x, use_lrt=True):
        """(src/weights.py:95-108) ``x``: (batch, D)."""
        if use_lrt:
            return self.sample_lrt(x, self.bias)
        if self._diag_route(x) == "kernel":
            epsilon = torch.randn(self.D, device=self.g_mu.device)          # sample()'s draw
            g_tilde = self.g_mu + self.g_sigma * epsilon
            return self._diag_kernel(x, g_tilde.unsqueeze(0), self.bias, None, False)
        return F.linear(x, self.sample(), self.bias)

    def forward_mc(self, x, n_samples, relu_in=False, relu_out=False):
        """``n_samples`` independent forward passes in one go: ``x`` is ``(batch, D)``
        (shared input) or ``(n_samples, batch, D)``; returns ``(n_samples, batch, D)``.  Sample k is
        what ``forward`` computes with the k-th row of one ``randn(n_samples, D)`` draw: one launch applies every
        sample's (diagonal) weight on the GPU, or -- faithful dataflow -- one fused launch builds every sample's weight
        matrix and one batched GEMM applies them.  ``relu_in`` / ``relu_out``: ``relu(layer(relu(x)))`` -- an ``nn.ReLU``
        in front of / behind this layer in the caller's module list, folded into the launch where ``fuses_relu(x)``."""
        self._mc_kl = None
        if (relu_in or relu_out) and not self.fuses_relu(x):
            out = self.forward_mc(torch.relu(x) if relu_in else x, n_samples)
            return torch.relu(out) if relu_out else out
        # one randn(S, D) draw (or the in-kernel generator), softplus, g_sigma * eps and the KL terms
        u, kl = _draw_and_reparam(self, self.g_mu.unsqueeze(0), self.g_rho.unsqueeze(0), n_samples, self.lambda_)
        u = u.squeeze(0)                                                          # (1 + S, D)
        self._mc_kl = None if kl is None else kl.squeeze(0)   # KL of this pass, for WHVINetwork.loss
        route = self._diag_route(x)
        if route == "kernel":
            return self._diag_kernel(x, u, self.bias, n_samples, True, relu_in, relu_out)   # (S, batch, D), bias included
        if route == "ops":
            w = _mean_plus_rest(self._w_bar_diagonal(u), 0)                       # (S, D)
            out = (x if x.dim() == 3 else x.unsqueeze(0)) * w.unsqueeze(1)
            return out + self.bias if self.bias is not None else out
        if u.device.type == "cuda":
            W = self._w_bar_stack(u, mean_plus=True)                             # (S, D, D), the sum in-kernel
        else:
            W = torch.stack([self.w_bar(row) for row in u])
            W = _mean_plus_rest(W, 0)                                            # (S, D, D)
        out = torch.matmul(x, W.transpose(1, 2))                                 # broadcasts a 2-D x
        return out + self.bias if self.bias is not None else out


_PACKED_NAMES = ("s1", "s2", "g_mu", "g_rho")


class _PackedSubMatrix(WHVISquarePow2Matrix):
    """A sub-matrix of a packed ``WHVIStackedMatrix``: same methods, but its four vectors are row ``_index`` of the
    parent's packed parameters (views made on access, so autograd and ``.to()`` go through the parent)."""

    def _row(self, name):
        return getattr(self._packed_parent(), "packed_" + name)[self._index]

    def __getstate__(self):                    # the back-reference is a weakref: rebuilt by the parent's __setstate__
        state = self.__dict__.copy()
        state.pop("_packed_parent", None)
        return state

    s1 = property(lambda self: self._row("s1"))
    s2 = property(lambda self: self._row("s2"))
    g_mu = property(lambda self: self._row("g_mu"))
    g_rho = property(lambda self: self._row("g_rho"))


class WHVIStackedMatrix(nn.Module):
    inkernel_rng = False      # see WHVISquarePow2Matrix
    hip_apply = True          # batched GPU pass: the narrow-input product as one HIP launch (False: torch.matmul / rocBLAS)

    def __init__(self, n_in, n_out, lambda_=1e-5, bias=False):
        """Arbitrary (n_out, n_in) matrix as a vertical stack of square power-of-two blocks
        (src/weights.py:112-133)."""
        super().__init__()
        self.n_in = n_in
        self.n_out = n_out
        self.lambda_ = lambda_
        self.D_in, self.D_out, self.padding, self.stack = self.setup_dimensions(n_in, n_out)
        self.weight_matrices = nn.ModuleList(
            [WHVISquarePow2Matrix(self.D_in, lambda_=lambda_) for _ in range(self.stack)])
        self.bias = nn.Parameter(torch.zeros(1, self.D_out)) if bias else None
        self._packed = False

    # ---- opt-in packed parameter layout -------------------------------------------------------------------
    def pack_parameters(self):
        """Replace the ``4 * stack`` per-sub-matrix parameter vectors (the reference's layout, src/weights.py:130-132)
        by four ``(stack, D_in)`` parameters ``packed_s1 / packed_s2 / packed_g_mu / packed_g_rho``.

        Same values, same forward / backward arithmetic, and ``state_dict()`` / ``load_state_dict()`` keep the
        reference's per-sub-matrix keys (``weight_matrices.<j>.s1`` ...), so checkpoints interchange in both
        directions.  What changes is ``parameters()`` / ``named_parameters()``: 4 tensors instead of ``4 * stack``,
        which is what an eager training step of e.g. ``WHVILinear(3, 1024)`` (1024 parameter tensors) is bound by.
        Call it before creating the optimizer."""
        if self._packed:
            return self
        for name in _PACKED_NAMES:
            with torch.no_grad():
                packed = torch.stack([getattr(m, name) for m in self.weight_matrices]).contiguous()
            self.register_parameter("packed_" + name, nn.Parameter(packed))
        for m in self.weight_matrices:
            for name in _PACKED_NAMES:
                del m._parameters[name]
        self._packed = True
        self._bind_sub_matrices()
        return self

    def _bind_sub_matrices(self):
        import weakref
        ref = weakref.ref(self)
        for j, m in enumerate(self.weight_matrices):
            object.__setattr__(m, "_packed_parent", ref)
            object.__setattr__(m, "_index", j)
            m.__class__ = _PackedSubMatrix

    def __setstate__(self, state):             # unpickling and copy.deepcopy: point the sub-matrix views at THIS object
        super().__setstate__(state)
        if getattr(self, "_packed", False):
            self._bind_sub_matrices()

    def _stacked(self, name):
        """(stack, D_in) tensor of one per-sub-matrix vector: the packed parameter itself, or a stack of the leaves."""
        if self._packed:
            return getattr(self, "packed_" + name)
        return torch.stack([getattr(m, name) for m in self.weight_matrices])

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if not self._packed:
            return super()._save_to_state_dict(destination, prefix, keep_vars)
        if self.bias is not None:
            destination[prefix + "bias"] = self.bias if keep_vars else self.bias.detach()
        for j in range(self.stack):                       # the reference's keys, sub-matrix by sub-matrix
            for name in _PACKED_NAMES:
                row = getattr(self, "packed_" + name)[j]
                destination[f"{prefix}weight_matrices.{j}.{name}"] = row if keep_vars else row.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if not self._packed:
            return super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                                 unexpected_keys, error_msgs)
        for j in range(self.stack):
            for name in _PACKED_NAMES:
                key = f"{prefix}weight_matrices.{j}.{name}"
                if key not in state_dict:
                    missing_keys.append(key)
                    continue
                value = state_dict.pop(key)               # consumed here: the (parameter-less) sub-modules never see it
                target = getattr(self, "packed_" + name)
                if tuple(value.shape) != tuple(target.shape[1:]):
                    error_msgs.append(f"size mismatch for {key}: {tuple(value.shape)} vs {tuple(target.shape[1:])}")
                    continue
                with torch.no_grad():
                    target[j].copy_(value)
        packed = {n: self._parameters.pop("packed_" + n) for n in _PACKED_NAMES}    # keep the base class off them
        try:
            super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                          error_msgs)
        finally:
            for n, p_ in packed.items():
                self._parameters["packed_" + n] = p_

    @staticmethod
    def setup_dimensions(D_in, D_out):
        """(D_in_adjusted, D_out_adjusted, padding, stack) -- src/weights.py:135-160, including its
        float-log guard: when ``2 ** ceil(log2(D_in))`` comes out as ``2 * D_in`` the input is
        already a power of two and is left alone."""
        next_power = 2 ** math.ceil(math.log(D_in, 2))
        if next_power == 2 * D_in:
            padding = 0
        else:
            padding = next_power - D_in
            D_in = next_power
        stack = -(-D_out // D_in)
        if D_out % D_in != 0:
            D_out = D_in * stack
        return D_in, D_out, padding, stack

    @property
    def kl(self):
        if self._on_gpu():
            # one evaluation over all sub-matrices (they share lambda): the same sum of terms as the reference's
            # per-matrix loop, in one pass instead of stack x ~12 tiny launches
            return _posterior_kl(self._stacked("g_mu"), self._stacked("g_rho"), self.lambda_)
        return sum(weight.kl for weight in self.weight_matrices)

    def _stacked_w_bar(self, parts, mean_plus=False):
        """All sub-matrices in ONE fused launch (the reference loops over them,
        src/weights.py:177-180: 4 FWHT launches each).  ``parts(m, eps)`` returns the list of ``u``
        vectors of sub-matrix m; result (stack, len(parts), D, D)."""
        dev = self._stacked_device()
        # one draw for every sub-matrix, in sub-matrix order like the reference's sequential draws
        eps = torch.randn(self.stack, self.D_in, device=dev)
        s1, s2, g_mu = self._stacked("s1"), self._stacked("s2"), self._stacked("g_mu")
        g_sigma = F.softplus(self._stacked("g_rho"))
        u = torch.stack(parts(g_mu, g_sigma, eps), dim=1)          # (stack, n_parts, D)
        return WBarFunction.apply(s1, u, s2, None, mean_plus)

    def _stacked_device(self):
        return (self.packed_g_mu if self._packed else self.weight_matrices[0].g_mu).device

    def _on_gpu(self):
        return self._stacked_device().type == "cuda"

    def sample(self):
        if self._on_gpu():
            W = self._stacked_w_bar(lambda mu, sg, eps: [mu + sg * eps])
            return W.reshape(self.stack * self.D_in, self.D_in)
        return torch.cat([weight.sample() for weight in self.weight_matrices])

    def sample_lrt(self, h):
        if self._on_gpu():
            W = self._stacked_w_bar(lambda mu, sg, eps: [mu, sg * eps], mean_plus=True)   # (stack, 1, D, D)
            W = W.reshape(self.stack * self.D_in, self.D_in)                      # cat over sub-matrices
            return h @ W.T
        return torch.cat([weight.sample_lrt(h) for weight in self.weight_matrices], dim=1)

    def fuses_relu(self, x):
        """True when the batched pass folds an ``nn.ReLU`` behind this layer into its product's launch (a ReLU in front of it
        acts on the narrow input and is a negligible op of its own)."""
        from whvi_amd import _hip
        return bool(self.hip_apply and x.dim() == 2 and x.device.type == "cuda" and x.dtype == torch.float32
                    and self.D_in in (4, 8) and _hip.small_k_apply_supported(x.new_empty((1, self.D_in)), self.stack * self.D_in))

    def forward_mc(self, x, n_samples, relu_in=False, relu_out=False):
        """Batched MC forward, see WHVISquarePow2Matrix.forward_mc; ``x``: (batch, n_in) or
        (n_samples, batch, n_in) -> (n_samples, batch, n_out).  Sample k of sub-matrix j uses row
        ``[j, k]`` of one ``randn(stack, n_samples, D_in)`` draw.  ``relu_in`` / ``relu_out``: ``relu(layer(relu(x)))``."""
        if relu_in:
            x = torch.relu(x)
        if relu_out and not self.fuses_relu(x):
            return torch.relu(self.forward_mc(x, n_samples))
        S, J, D = n_samples, self.stack, self.D_in
        dev = self._stacked_device()
        s1, s2 = self._stacked("s1"), self._stacked("s2")
        g_mu, g_rho = self._stacked("g_mu"), self._stacked("g_rho")
        u, kl = _draw_and_reparam(self, g_mu, g_rho, S, self.lambda_)               # (J, 1 + S, D)
        self._mc_kl = None if kl is None else kl.sum()
        if dev.type == "cuda":
            W = WBarFunction.apply(s1, u, s2, None, True)                           # (J, S, D, D), the sum in-kernel
        else:
            W = torch.stack([torch.stack([m.w_bar(row) for row in u[j]]) for j, m in enumerate(self.weight_matrices)])
            W = _mean_plus_rest(W, 1)
        W = W.transpose(0, 1).reshape(S, J * D, D)                                  # (S, stack*D, D)
        x_padded = torch.zeros((*x.size()[:-1], D), device=x.device)
        x_padded[..., :self.n_in] = x
        from whvi_amd import _hip
        if self.hip_apply and _hip.small_k_apply_supported(x_padded, J * D):
            out = SmallKApplyFunction.apply(x_padded, W, self.bias, relu_out)       # one write-only launch (K = D_in = 4 or 8), bias (and ReLU) included
        else:
            out = torch.matmul(x_padded, W.transpose(1, 2))
            if self.bias is not None:
                out = out + self.bias
        return out[..., :self.n_out]

    def forward(self, x, use_lrt=True):
        """Zero-pad the features to D_in, multiply, drop the surplus outputs
        (src/weights.py:182-208)."""
        x_padded = torch.zeros((*x.size()[:-1], self.D_in), device=x.device)
        x_padded[..., :self.n_in] = x
        if use_lrt:
            output = self.sample_lrt(x_padded)
            if self.bias is not None:
                output = output + self.bias
        else:
            output = F.linear(x_padded, self.sample(), self.bias)
        return output[..., :self.n_out]


class WHVIColumnMatrix(nn.Module):
    inkernel_rng = False      # see WHVISquarePow2Matrix
    hip_apply = True          # batched GPU pass of the transposed layer: the row dot as one HIP launch (False: torch.matmul)

    def __init__(self, n_out, lambda_=1e-5, bias=False, transposed=False):
        """Column (n_out, 1) matrix -- or row (1, n) when ``transposed`` -- cut from a square
        sample (src/weights.py:212-228)."""
        super().__init__()
        self.D = n_out
        self.D_adjusted = 2 ** math.ceil(math.log(n_out, 2))
        self.weight_submodule = WHVISquarePow2Matrix(self.D_adjusted, lambda_=lambda_)
        self.transposed = transposed
        self.bias = nn.Parameter(torch.zeros(1, 1 if transposed else n_out)) if bias else None

    @property
    def kl(self):
        return self.weight_submodule.kl

    def sample(self):
        """First ``D`` entries of the row-major flattening of a square sample
        (src/weights.py:239-248)."""
        if self.weight_submodule.g_mu.device.type == "cuda":
            # the first D entries of the row-major flattening lie in row 0 (D <= D_adjusted)
            matrix = self.weight_submodule.sample(rows=1).reshape(-1, 1)[:self.D]
        else:
            matrix = torch.reshape(self.weight_submodule.sample(), (-1, 1))[:self.D]
        return matrix.T if self.transposed else matrix

    def forward(self, x):
        return F.linear(x, self.sample(), self.bias)

    def fuses_relu(self, x):
        """True when the batched pass folds an ``nn.ReLU`` in front of this (transposed) layer into its row-dot launch."""
        from whvi_amd import _hip
        return bool(self.hip_apply and self.transposed and x.dim() == 3 and self.D == self.weight_submodule.D and _hip.row_dot_supported(x))

    def forward_mc(self, x, n_samples, relu_in=False, relu_out=False):
        """Batched MC forward (direct weight sampling like ``forward``); ``x``: (batch, n_in) or
        (n_samples, batch, n_in) -> (n_samples, batch, n_out).  Only row 0 of every sampled square
        matrix is ever built.  ``relu_in`` / ``relu_out``: ``relu(layer(relu(x)))``."""
        if relu_out:
            return torch.relu(self.forward_mc(x, n_samples, relu_in=relu_in))
        if relu_in and not (self.fuses_relu(x) and x.shape[0] == n_samples):
            x, relu_in = torch.relu(x), False
        sq = self.weight_submodule
        u, kl = _draw_and_reparam(self, sq.g_mu.unsqueeze(0), sq.g_rho.unsqueeze(0), n_samples, sq.lambda_)
        self._mc_kl = None if kl is None else kl.squeeze(0)
        g_tilde = _mean_plus_rest(u.squeeze(0), 0)                       # (S, D_adj): g_mu + g_sigma * eps
        if g_tilde.device.type == "cuda":
            rows0 = sq._w_bar_stack(g_tilde, rows=1).squeeze(1)          # (S, D_adj)
        else:
            rows0 = torch.stack([sq.w_bar(g)[0] for g in g_tilde])
        w = rows0 if self.D == sq.D else rows0[:, :self.D]               # (S, D)
        if self.transposed:                       # weight (1, D): out = F.linear(x, w[None]) per sample = x @ w: one read of x
            from whvi_amd import _hip
            if self.hip_apply and x.dim() == 3 and x.shape[0] == n_samples and _hip.row_dot_supported(x):
                out = RowDotFunction.apply(x, w.contiguous(), relu_in)   # one read-only launch for all samples (ReLU on load)
            else:
                out = torch.matmul(x, w.unsqueeze(-1))   # batched GEMV (a product + a sum pass moved 2.5x the bytes: 1.6 vs 0.62 ms at config 4)
        else:                                     # weight (D, 1): out = x[..., :1] * w
            out = x * w.unsqueeze(1)
        return out + self.bias if self.bias is not None else out
