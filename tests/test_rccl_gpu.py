"""The additive multi-GPU layer (whvi_amd/parallel.py, bench.py's N > 1 phases) through a real RCCL communicator on
the GPU box.  The box has ONE GPU and RCCL cannot put two ranks on one device, so the group has one rank -- created
inside this test process (no re-exec, no launcher).  Every collective of the N > 1 path still goes through
``ncclAllGather`` / ``ncclAllReduce``: a one-rank group does not short-circuit (parallel._in_group).  The
world-size-2 semantics of the same functions are covered on CPUs with gloo (tests/test_parallel.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.nn as nn

from whvi_amd import parallel
from whvi_amd.layers import WHVILinear
from whvi_amd.networks import WHVIRegression

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_group():
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        yield dev
    finally:
        dist.destroy_process_group()


def test_backend_is_rccl(rccl_group):
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    assert torch.version.hip is not None, "backend 'nccl' must be RCCL: a ROCm build of torch"
    t = torch.tensor([1.5, -2.0], dtype=torch.float64, device=rccl_group)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    assert t.tolist() == [1.5, -2.0]


def test_gather_predictions_runs_the_collective(rccl_group, monkeypatch):
    calls = []
    real = dist.all_gather_into_tensor

    def spy(out, inp, *a, **k):
        calls.append((tuple(out.shape), tuple(inp.shape), inp.device.type))
        return real(out, inp, *a, **k)
    monkeypatch.setattr(dist, "all_gather_into_tensor", spy)
    local = torch.randn(45, 2, 3, device=rccl_group)
    got = parallel.gather_predictions(local)
    assert calls == [((45, 2, 3), (45, 2, 3), "cuda")], "a one-rank group must still all-gather"
    assert torch.equal(got, local)
    # ragged counts: the pad-and-trim path
    got = parallel.gather_predictions(local, counts=[3])
    assert torch.equal(got, local)


def test_mc_sharded_forward_and_grad_all_reduce(rccl_group):
    dev = rccl_group
    torch.manual_seed(2)
    net = WHVIRegression([WHVILinear(3, 16), nn.ReLU(), WHVILinear(16, 16), nn.ReLU(), WHVILinear(16, 1)]).to(dev)
    x, y = torch.randn(7, 3, device=dev), torch.randn(7, 1, device=dev)
    a = parallel.mc_sharded_forward(net, x, n_samples=6, base_seed=1)
    b = parallel.mc_sharded_forward(net, x, n_samples=6, base_seed=1)
    c = parallel.mc_sharded_forward(net, x, n_samples=6, base_seed=2)
    assert a.shape == (7, 1, 6) and torch.equal(a, b) and not torch.equal(a, c)
    # the opt-in in-kernel generator follows the same (base_seed, rank) determinism
    net.set_inkernel_rng()
    a = parallel.mc_sharded_forward(net, x, n_samples=6, base_seed=1)
    b = parallel.mc_sharded_forward(net, x, n_samples=6, base_seed=1)
    c = parallel.mc_sharded_forward(net, x, n_samples=6, base_seed=2)
    assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a[:, :, 0], a[:, :, 1])
    net.set_inkernel_rng(False)
    net.train()
    net.loss(x, y, 70).backward()
    before = [p.grad.clone() for p in net.parameters()]
    net.likelihood.sigma.grad = None                      # a parameter without a local gradient: zeros are sent
    parallel.all_reduce_grads(net)
    for p, g in zip(net.parameters(), before):
        if p is net.likelihood.sigma:
            assert float(p.grad) == 0.0
        else:
            assert torch.equal(p.grad, g)                 # average over one rank


def test_sharded_training_step_through_rccl(rccl_group, monkeypatch):
    """``parallel.mc_sharded_loss`` on the GPU inside a one-rank RCCL group: the batched pass through the HIP kernels,
    backward, the flattened gradient all-reduce and the scalar loss all-reduce really go through ``nccl``; with one
    rank the step must equal the plain single-process loss with the same seeding, bit for bit; ``train_model`` takes
    the sharded step by itself inside a group."""
    from torch.utils.data import DataLoader, TensorDataset
    from whvi_amd.evaluation import make_optimizer
    dev = rccl_group
    torch.manual_seed(6)
    net = WHVIRegression([WHVILinear(3, 16, lambda_=2.0), nn.ReLU(), WHVILinear(16, 16, lambda_=2.0), nn.ReLU(),
                          WHVILinear(16, 1, lambda_=2.0)], train_samples=4).to(dev).train()
    x, y = torch.randn(10, 3, device=dev), torch.randn(10, 1, device=dev)
    seen = []
    real = dist.all_reduce
    monkeypatch.setattr(dist, "all_reduce", lambda t, *a, **k: (seen.append((tuple(t.shape), t.device.type)), real(t, *a, **k))[1])
    total = parallel.mc_sharded_loss(net, x, y, n=100, n_samples=4, base_seed=3)
    n_params = sum(p.numel() for p in net.parameters())
    assert seen == [((n_params,), "cuda"), ((), "cuda")], seen          # gradients (one flat buffer), then the loss scalar
    grads = [p.grad.clone() for p in net.parameters()]
    net.zero_grad(set_to_none=True)
    with torch.random.fork_rng(devices=[dev]):
        torch.manual_seed(parallel.sample_seed(3, 0))
        want = net.loss(x, y, n=100)
        want.backward()
    assert float(total) == float(want)
    for p, g in zip(net.parameters(), grads):
        assert torch.equal(p.grad, g)
    del want
    net.zero_grad(set_to_none=True)
    seen.clear()
    loader = DataLoader(TensorDataset(x, y), batch_size=5)
    optimizer, scheduler = make_optimizer(net, lambda0=0.05)
    before = [p.detach().clone() for p in net.parameters()]
    net.train_model(loader, optimizer, scheduler, epochs1=1, epochs2=1)      # a process group exists: sharded by itself
    assert len(seen) == 8 and all(dev_type == "cuda" for _, dev_type in seen)   # 4 steps x (gradients + loss)
    assert any(not torch.equal(p.detach(), b) for p, b in zip(net.parameters(), before))
    assert all(bool(torch.isfinite(p).all()) for p in net.parameters())


def test_graph_survives_an_evaluation_pass_with_the_inkernel_rng(rccl_group):
    """``mc_sharded_forward`` used to REPLACE every layer's Philox state tensor; a hipGraph captured
    before it kept launching with the old tensor's address (freed by then).  Now the state is re-seeded in place and
    restored: the graph's tensor is still the layer's tensor, its offset keeps advancing across replays, and the
    training stream continues where it was instead of restarting at offset 0 after every evaluation."""
    from whvi_amd.graphs import GraphedPredictor
    dev = rccl_group
    torch.manual_seed(4)
    net = WHVIRegression([WHVILinear(3, 16), nn.ReLU(), WHVILinear(16, 16), nn.ReLU(), WHVILinear(16, 1)]).to(dev)
    net.set_inkernel_rng()
    x = torch.randn(9, 3, device=dev)
    graphed = GraphedPredictor(net, x, n_samples=4)
    mods = [m for m in net.modules() if getattr(m, "_rng_state", None) is not None]
    assert len(mods) >= 3
    ptrs = [m._rng_state.data_ptr() for m in mods]
    first = graphed(x).clone()
    torch.cuda.synchronize()
    offsets = [int(m._rng_state[1]) for m in mods]
    assert all(o > 0 for o in offsets)
    a = parallel.mc_sharded_forward(net, x, n_samples=6, base_seed=1)
    b = parallel.mc_sharded_forward(net, x, n_samples=6, base_seed=1)
    assert torch.equal(a, b)                                             # evaluation is still reproducible
    assert [m._rng_state.data_ptr() for m in mods] == ptrs, "the state tensors a graph points at must stay"
    assert [int(m._rng_state[1]) for m in mods] == offsets, "the training stream continues where it was"
    second = graphed(x).clone()
    torch.cuda.synchronize()
    assert all(int(m._rng_state[1]) > o for m, o in zip(mods, offsets)), "the replay advanced the live state"
    assert not torch.equal(first, second) and bool(torch.isfinite(second).all())
    net.set_inkernel_rng(False)


def test_bench_multi_gpu_phases_small(rccl_group):
    """bench.py's N > 1 phases (config 5 row shards, config 4 MC-sharded pass with its all-gather) at test sizes."""
    import bench
    out = bench.multi_gpu_extras(rccl_group, 0, 1, small=True)
    f16 = out["fwht_f16_D4096_2^20rows_row_sharded"]
    assert f16["rows_total"] == 4096 and f16["ms"] > 0
    net = out["whviregression_3_1024_1024_1_mc128_sharded"]
    assert net["prediction_shape"] == [33, 1, 8] and net["ms"] > 0
    train = out["whviregression_3_1024_1024_1_mc128_sharded_train_step"]
    assert train["ms"] > 0 and train["values_finite"] and train["mc_samples_per_gpu"] == 8
