"""Multi-GPU execution of the WHVI hot path: one process per GPU, ``torch.distributed`` over RCCL
(backend ``"nccl"`` on ROCm) across the xGMI mesh; ``gloo`` on CPUs for tests.

The reference is single-process; this module is additive.  What shards:

* rows of a batched FWHT and Monte-Carlo samples of a WHVI network are INDEPENDENT units, so they
  are partitioned over ranks with NO collective on the data path (``shard_bounds``,
  ``fwht_row_shard``);
* the only exchange is the predictive-sample reduction: every rank holds predictions for its own
  MC samples, ``(batch, n_out, S_local)``, and one all-gather assembles ``(batch, n_out, S)`` for
  the mean / MNLL (src/networks.py:51,112-114; src/likelihoods.py:26-28).  The blocks are small
  (MBs at most), i.e. latency-bound: a single all-gather per forward, never one per layer.
* training additionally averages gradients of the O(D) parameter vectors (``all_reduce_grads``).
"""
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

__all__ = ["init_from_env", "shard_bounds", "fwht_row_shard", "gather_predictions",
           "mc_sharded_forward", "mc_sharded_loss", "all_reduce_grads", "sample_seed", "seed_inkernel_rng"]


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, torch.device]:
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun's contract).
    Returns (rank, world_size, device).  Single-process runs need no environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if use_gpu and (backend or "nccl") == "nccl":
            kwargs["device_id"] = device
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world, **kwargs)
    return rank, world, device


def _world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _in_group() -> bool:
    """True inside an initialised process group -- of ANY size: a one-rank group still runs its collectives (through
    RCCL on a GPU), which is how the single-GPU box exercises the N > 1 code path (tests/test_rccl_gpu.py)."""
    return dist.is_available() and dist.is_initialized()


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of a balanced contiguous partition of n units (the first n % world ranks get
    one extra unit).  Units are rows or MC samples."""
    base, extra = divmod(n, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def fwht_row_shard(x_full_rows: int, rank: int = None, world: int = None) -> Tuple[int, int]:
    """Row range of the ``(rows, D)`` FWHT problem owned by this rank: rows are independent, so the
    sharded transform is each rank calling the single-GPU kernel on its slice -- no collective."""
    r, w = _world()
    return shard_bounds(x_full_rows, r if rank is None else rank, w if world is None else world)


def sample_seed(base_seed: int, rank: int) -> int:
    """Distinct, reproducible generator seed per rank so that ranks draw different eps."""
    return (int(base_seed) * 1_000_003 + 7919 * (rank + 1)) % (2 ** 63 - 1)


def _inkernel_rng_modules(net: torch.nn.Module):
    """(index, module) of every layer that draws eps inside the reparameterisation kernel (``net.set_inkernel_rng()``)."""
    for index, module in enumerate(net.modules()):
        if getattr(module, "inkernel_rng", False) and hasattr(type(module), "inkernel_rng"):
            yield index, module


def seed_inkernel_rng(net: torch.nn.Module, base_seed: int, rank: int = None) -> int:
    """Re-seed every layer that draws eps inside the reparameterisation kernel (``net.set_inkernel_rng()``): Philox seed
    from (base_seed, RANK, layer index), offset 0.  torch's generators do not reach that stream, so without this all
    ranks of a job started from one ``torch.manual_seed`` would draw identical eps.

    The state is re-seeded IN PLACE whenever the layer already has one: a captured hipGraph (``GraphedPredictor``,
    ``GraphedTrainStep``) has the state tensor's device address baked into its launches, so replacing the tensor would
    leave the graph reading the seed and writing the offset through memory the caching allocator has since handed to
    someone else.  Returns the number of layers seeded."""
    from whvi_amd import _hip
    r = _world()[0] if rank is None else rank
    count = 0
    for index, module in _inkernel_rng_modules(net):
        device = next(module.parameters()).device
        seed = (sample_seed(base_seed, r) + 104_729 * (index + 1)) % (2 ** 62)
        state = getattr(module, "_rng_state", None)
        if state is None or state.device != device:
            module._rng_state = _hip.new_rng_state(device, seed=seed)
        else:
            state.copy_(torch.tensor([seed, 0, 0], dtype=torch.int64))
        count += 1
    return count


class _ForkedInkernelRng:
    """``with _ForkedInkernelRng(net):`` -- what ``torch.random.fork_rng`` does for torch's generators, for the in-kernel
    Philox states: the values are saved on entry and copied back (into the SAME tensors) on exit, so an evaluation pass
    that re-seeds them neither disturbs the training stream (which would otherwise restart from the same seed at offset
    0 after every evaluation and repeat its eps) nor moves a tensor a captured graph points at."""

    def __init__(self, net):
        self.net = net

    def __enter__(self):
        self.saved = []
        for _, module in _inkernel_rng_modules(self.net):
            state = getattr(module, "_rng_state", None)
            self.saved.append((module, state, None if state is None else state.clone()))
        return self

    def __exit__(self, *exc):
        for module, state, values in self.saved:
            if state is None:
                module._rng_state = None       # had none: the training stream still draws its own seed on first use
            else:
                state.copy_(values)
                module._rng_state = state
        return False


def gather_predictions(local: torch.Tensor, counts=None) -> torch.Tensor:
    """All-gather per-rank predictions ``(batch, n_out, S_local)`` into ``(batch, n_out, S)`` with
    rank-major sample order.  ``counts`` (samples per rank) allows ragged shards; ranks with fewer
    samples are padded for the collective and trimmed afterwards."""
    rank, world = _world()
    if not _in_group():
        return local
    s_local = local.size(2)
    if counts is None:
        counts = [s_local] * world
    s_max = max(counts)
    send = local
    if s_local < s_max:
        pad = torch.zeros(*local.shape[:2], s_max - s_local, dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], dim=2)
    # the gathered tensor is for the predictive reduction (mean / MNLL reporting); gradients flow
    # through each rank's LOCAL samples only and are combined by all_reduce_grads
    send = send.detach().contiguous()
    # rank-major concatenation along dim 0 (the layout both RCCL and gloo accept)
    flat = torch.empty((world * send.size(0),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(flat, send)
    buf = flat.view((world,) + tuple(send.shape))
    parts = [buf[r, :, :, :counts[r]] for r in range(world)]
    return torch.cat(parts, dim=2)


def mc_sharded_forward(net, x: torch.Tensor, n_samples: int, base_seed: int = 0) -> torch.Tensor:
    """Predictive samples of a ``WHVINetwork`` with the MC-sample axis sharded over ranks.

    Every rank runs the full batch through the network for its share of the ``n_samples`` draws
    (parameters are replicated: 4 vectors of D floats per square matrix), seeded per rank, then one
    all-gather returns ``(batch, n_out, n_samples)`` on every rank -- the layout of
    src/networks.py:41,50-51."""
    rank, world = _world()
    begin, end = shard_bounds(n_samples, rank, world)
    counts = [shard_bounds(n_samples, r, world)[1] - shard_bounds(n_samples, r, world)[0] for r in range(world)]
    devices = [x.device] if x.device.type == "cuda" else []
    n_local = end - begin
    with torch.random.fork_rng(devices=devices), _ForkedInkernelRng(net):
        torch.manual_seed(sample_seed(base_seed, rank))
        seed_inkernel_rng(net, base_seed, rank)      # the opt-in Philox stream: per rank too, same determinism
        if n_local == 0:
            n_out = net.sequential(x).size(-1)
            local = torch.zeros(x.size(0), n_out, 0, dtype=x.dtype, device=x.device)
        elif x.device.type == "cuda" and hasattr(net, "forward_batched"):
            local = net.forward_batched(x, n_local)          # all local samples in one batched pass
        else:
            draws = []
            for _ in range(n_local):
                out = net.sequential(x)
                draws.append(out.reshape(x.size(0), out.size(-1)))
            local = torch.stack(draws, dim=2)
    return gather_predictions(local, counts)


def _local_predictions(net, x: torch.Tensor, n_local: int) -> torch.Tensor:
    """``(batch, n_out, n_local)`` predictions of this rank's Monte-Carlo samples, with autograd."""
    if n_local == 0:
        n_out = net.sequential(x).size(-1)
        return torch.zeros(x.size(0), n_out, 0, dtype=x.dtype, device=x.device)
    mode = getattr(net, "mc_mode", "auto")
    if mode == "auto":
        mode = "batched" if x.device.type == "cuda" else "loop"
    if mode == "batched" and hasattr(net, "forward_batched"):
        return net.forward_batched(x, n_local)               # all local samples in one batched pass
    draws = []
    for _ in range(n_local):
        out = net.sequential(x)
        draws.append(out.reshape(x.size(0), out.size(-1)))
    return torch.stack(draws, dim=2)


def mc_sharded_loss(net, x: torch.Tensor, y: torch.Tensor, n: int, n_samples: int, base_seed: int = None,
                    ignore_kl: bool = False, backward: bool = True) -> torch.Tensor:
    """One training step's negative ELBO with the Monte-Carlo samples sharded over the ranks ("training
    would add an all-reduce(sum) of 4.D-float gradient vectors per layer + likelihood.sigma").

    The single-process loss of src/networks.py:56-69 with S samples is ``MNLL_S + KL`` where the MNLL estimate averages
    over all S samples (src/likelihoods.py:26-28).  Rank r runs the FULL batch through the network for its S_r samples
    and forms

        loss_r = (S_r / S) . MNLL_{S_r}  +  KL / world

    whose sum over ranks is exactly the single-process loss over the union of the ranks' samples; ``backward()`` then
    yields each rank's share of the gradient and ONE flattened all-reduce (sum) of the O(D) parameter gradients makes
    every rank hold the full gradient -- so identical optimizer steps keep the replicated parameters bit-equal across
    ranks.  No collective on the data path.

    ``x`` / ``y`` are the same batch on every rank.  ``base_seed``: when given, rank r draws from generators seeded
    with (base_seed, r) for this call (reproducible; pass the step index); otherwise the current generators are used
    (seed them per rank once).  Returns the detached global loss (one scalar all-reduce, for monitoring); with
    ``backward=False`` only the local share is built and returned with its graph."""
    import contextlib
    rank, world = _world()
    begin, end = shard_bounds(n_samples, rank, world)
    n_local = end - begin
    if base_seed is None:
        context = contextlib.nullcontext()
    else:
        devices = [x.device] if x.device.type == "cuda" else []
        context = contextlib.ExitStack()
        context.enter_context(torch.random.fork_rng(devices=devices))
        context.enter_context(_ForkedInkernelRng(net))
    with context:
        if base_seed is not None:
            torch.manual_seed(sample_seed(base_seed, rank))
            seed_inkernel_rng(net, base_seed, rank)
        net._pass_kl = None
        if n_local > 0:
            pred = _local_predictions(net, x, n_local)
            mnll = net.likelihood.mnll_batch_estimate(y, pred, n)
        else:
            # a rank without samples (n_samples < world): no network pass at all; its share of the loss is KL / world
            pred, mnll = None, torch.zeros((), dtype=x.dtype, device=x.device)
        pass_kl, net._pass_kl = getattr(net, "_pass_kl", None), None
        local = mnll * (n_local / float(n_samples))
        kl = None
        if not ignore_kl:
            kl = pass_kl if pass_kl is not None else net.kl
            local = local + kl / world
    net.current_mnll = mnll.detach() if torch.is_tensor(mnll) else mnll       # this rank's estimate (monitoring)
    if kl is not None:
        net.current_kl = kl.detach() if torch.is_tensor(kl) else kl
    if not backward:
        return local
    if local.requires_grad:
        local.backward()
    # (else: no samples here and ignore_kl -- nothing to differentiate; this rank still JOINS the all-reduce below with
    # zero gradients: a rank that skipped it would leave the others blocked inside the collective)
    total = local.detach().clone()
    del local, mnll, pred                                    # no graph of this pass outlives it
    all_reduce_grads(net, average=False)
    if _in_group():
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    return total


def all_reduce_grads(module: torch.nn.Module, average: bool = True) -> None:
    """Sum (or average) parameter gradients over ranks in ONE flattened all-reduce -- the WHVI
    parameters are a handful of length-D vectors, so bucketing per tensor would be latency-bound."""
    rank, world = _world()
    if not _in_group():
        return
    # every rank sends the SAME layout: all parameters that require a gradient, zeros where this rank has none
    # (a shard with no samples, a parameter its local pass did not touch) -- never a rank-dependent subset, never a
    # skipped collective, or the other ranks would block or mismatch sizes
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= world
    offset = 0
    for p in params:
        n = p.numel()
        piece = flat[offset:offset + n].view_as(p)
        if p.grad is None:
            p.grad = piece.clone()
        else:
            p.grad.copy_(piece)
        offset += n
