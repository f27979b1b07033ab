"""world_size-2 ``gloo`` tests of the sharding helpers on the CPU (the N > 1 path of bench.py and
``whvi_amd.parallel``)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from whvi_amd import parallel


def test_shard_bounds_partition():
    for n in (0, 1, 7, 128, 1000):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    assert len({parallel.sample_seed(0, r) for r in range(8)}) == 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.nn as nn
    import fwht_cpp
    from whvi_amd.layers import WHVILinear
    from whvi_amd.networks import WHVIRegression
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1. row-sharded FWHT: every rank transforms its slice, results concatenate to the full answer
        g = torch.Generator().manual_seed(0)
        full = torch.randn(38, 64, generator=g)   # even split: plain all_gather needs equal blocks
        b, e = parallel.fwht_row_shard(38)
        mine = fwht_cpp.forward(full[b:e])
        gathered = [torch.zeros(parallel.shard_bounds(38, r, world)[1] - parallel.shard_bounds(38, r, world)[0], 64)
                    for r in range(world)]
        dist.all_gather(gathered, mine)
        assert torch.equal(torch.cat(gathered), fwht_cpp.forward(full))

        # 2. MC-sample sharding + predictive all-gather (ragged: 5 samples over 2 ranks)
        torch.manual_seed(1)   # identical replicated parameters on every rank
        net = WHVIRegression([nn.Linear(2, 8), nn.ReLU(), WHVILinear(8, 8), nn.ReLU(), nn.Linear(8, 3)])
        x = torch.randn(6, 2, generator=g)
        pred = parallel.mc_sharded_forward(net, x, n_samples=5, base_seed=42)
        assert pred.shape == (6, 3, 5)
        # every rank holds the same gathered tensor
        ref = pred.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, pred)
        # and rank r's block equals what r computes alone with its seed
        lo, hi = parallel.shard_bounds(5, rank, world)
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(parallel.sample_seed(42, rank))
            own = torch.stack([net.sequential(x) for _ in range(hi - lo)], dim=2)
        assert torch.equal(pred[:, :, lo:hi], own)
        # different ranks drew different eps
        assert not torch.equal(pred[:, :, 0], pred[:, :, 3])

        # 3. gradient all-reduce == gradient of the mean loss over ranks
        y = torch.randn(6, 3, generator=g)
        torch.manual_seed(100 + rank)
        loss = net.loss(x, y, n=60)
        loss.backward()
        before = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
        parallel.all_reduce_grads(net)
        after = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        both = [torch.zeros_like(before) for _ in range(world)]
        dist.all_gather(both, before)
        assert torch.allclose(after, sum(both) / world, rtol=1e-6, atol=1e-7)

        # 4. a rank with missing gradients sends zeros in the same layout (never a shorter buffer, never a skip)
        for p in net.parameters():
            p.grad = None
        if rank == 0:
            net.loss(x, y, n=60).backward()
            mine = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
        parallel.all_reduce_grads(net)
        after = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        ref = mine if rank == 0 else torch.zeros_like(after)
        dist.broadcast(ref, src=0)
        assert torch.allclose(after, ref / world, rtol=1e-6, atol=1e-7)

        # 5. the opt-in in-kernel Philox stream is seeded per rank (ranks share torch.manual_seed for their parameters)
        torch.manual_seed(7)
        from whvi_amd.weights import _fresh_philox_seed
        seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(seeds, torch.tensor([_fresh_philox_seed()], dtype=torch.int64))
        assert len({int(t) for t in seeds}) == world, "same torch seed, different ranks -> different Philox seeds"
        wnet = WHVIRegression([WHVILinear(3, 8), nn.ReLU(), WHVILinear(8, 8), nn.ReLU(), WHVILinear(8, 1)])
        wnet.set_inkernel_rng()
        n_rng = sum(1 for m in wnet.modules() if getattr(m, "inkernel_rng", False))    # weight modules + their squares
        assert n_rng >= 3 and parallel.seed_inkernel_rng(wnet, base_seed=5) == n_rng
        mine = torch.stack([m._rng_state for m in wnet.modules() if getattr(m, "_rng_state", None) is not None])
        assert mine.shape == (n_rng, 3) and len({int(v) for v in mine[:, 0]}) == n_rng and int(mine[:, 1].abs().sum()) == 0
        states = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(states, mine)
        assert len({int(v) for st in states for v in st[:, 0]}) == n_rng * world, "every (rank, layer) has its own seed"
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_inkernel_rng_is_reseeded_in_place_and_restored():
    """a captured hipGraph has the Philox state tensor's address baked in, so re-seeding must write INTO the
    existing tensor (never replace it), and an evaluation pass must hand the training stream back where it was --
    otherwise every training step after an evaluation with the same base_seed would repeat the same eps."""
    import torch.nn as nn
    from whvi_amd.layers import WHVILinear
    from whvi_amd.networks import WHVIRegression
    net = WHVIRegression([WHVILinear(3, 8), nn.ReLU(), WHVILinear(8, 8), nn.ReLU(), WHVILinear(8, 1)])
    net.set_inkernel_rng()
    assert parallel.seed_inkernel_rng(net, base_seed=3) >= 3
    mods = [m for m in net.modules() if getattr(m, "_rng_state", None) is not None]
    ptrs = [m._rng_state.data_ptr() for m in mods]
    first = [m._rng_state.clone() for m in mods]
    for m in mods:
        m._rng_state[1] += 17                                  # "training has advanced the offset"
    parallel.seed_inkernel_rng(net, base_seed=4)
    assert [m._rng_state.data_ptr() for m in mods] == ptrs, "re-seeding must not replace the state tensor"
    assert all(int(m._rng_state[1]) == 0 and int(m._rng_state[0]) != int(f[0]) for m, f in zip(mods, first))
    # an evaluation pass re-seeds inside and restores on exit: same tensors, same values as before the call
    for m in mods:
        m._rng_state[1] += 5
    before = [m._rng_state.clone() for m in mods]
    pred = parallel.mc_sharded_forward(net, torch.randn(4, 3), n_samples=3, base_seed=9)
    assert pred.shape == (4, 1, 3)
    assert [m._rng_state.data_ptr() for m in mods] == ptrs
    assert all(torch.equal(m._rng_state, b) for m, b in zip(mods, before))
    # a layer that had no state yet has none afterwards (the training stream draws its own seed on first use)
    net.set_inkernel_rng()
    parallel.mc_sharded_forward(net, torch.randn(4, 3), n_samples=2, base_seed=9)
    assert all(getattr(m, "_rng_state", None) is None for m in net.modules())


def test_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _train_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import copy
        import torch.nn as nn
        from torch.utils.data import DataLoader, TensorDataset
        from whvi_amd.evaluation import make_optimizer
        from whvi_amd.layers import WHVILinear
        from whvi_amd.networks import WHVIRegression
        S = 5                                                 # ragged: 3 samples on rank 0, 2 on rank 1
        torch.manual_seed(11)                                 # replicated parameters and data: same seed on every rank
        net = WHVIRegression([WHVILinear(3, 8, lambda_=2.0), nn.ReLU(), WHVILinear(8, 8, lambda_=2.0), nn.ReLU(),
                              WHVILinear(8, 1, lambda_=2.0)], train_samples=S)
        x, y = torch.randn(12, 3), torch.randn(12, 1)
        single = copy.deepcopy(net)                           # the single-process reference, stepped beside the job
        opt = torch.optim.Adam(net.parameters(), lr=0.05)
        opt1 = torch.optim.Adam(single.parameters(), lr=0.05)
        for step in range(1, 4):
            # ---- one sharded step: local samples, backward, ONE all-reduce of the gradients
            opt.zero_grad(set_to_none=True)
            total = parallel.mc_sharded_loss(net, x, y, n=120, n_samples=S, base_seed=step)
            # ---- the same step in ONE process fed the union of both ranks' eps (rank r's draws come from the generator
            # seeded with (step, r), in rank order), through the plain single-process loss of src/networks.py:56-69
            draws = []
            for r in range(world):
                lo, hi = parallel.shard_bounds(S, r, world)
                with torch.random.fork_rng(devices=[]):
                    torch.manual_seed(parallel.sample_seed(step, r))
                    for _ in range(hi - lo):
                        out = single.sequential(x)
                        draws.append(out.reshape(x.size(0), out.size(-1)))
            pred = torch.stack(draws, dim=2)
            assert pred.shape == (12, 1, S)
            loss1 = single.likelihood.mnll_batch_estimate(y, pred, 120) + single.kl
            opt1.zero_grad(set_to_none=True)
            loss1.backward()
            assert abs(float(total) - float(loss1)) <= 1e-6 * abs(float(loss1)), (step, float(total), float(loss1))
            for (name, p), q in zip(net.named_parameters(), single.parameters()):
                assert p.grad is not None and torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-6 * float(q.grad.abs().max()) + 1e-12), (step, name)
            opt.step()
            opt1.step()
            for (name, p), q in zip(net.named_parameters(), single.parameters()):
                assert torch.allclose(p, q, rtol=0.0, atol=1e-6), (step, name)
        # replicated parameters are BIT-equal across ranks after three steps
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1])
        # ---- train_model picks the sharded step up by itself inside a process group (rank 0 writes the checkpoint)
        calls = []
        real = parallel.all_reduce_grads
        parallel.all_reduce_grads = lambda module, average=True: (calls.append(average), real(module, average))[1]
        try:
            loader = DataLoader(TensorDataset(x, y), batch_size=6)
            optimizer, scheduler = make_optimizer(net, lambda0=0.05)
            ckpt = os.path.join(tmp, f"ckpt{rank}")
            os.makedirs(ckpt)
            net.train_model(loader, optimizer, scheduler, epochs1=1, epochs2=1, checkpoint_dir=ckpt)
        finally:
            parallel.all_reduce_grads = real
        assert calls == [False] * 4, calls                    # 2 epochs x 2 batches, gradients SUMMED
        assert (os.listdir(ckpt) == ["epoch-0.pth"]) == (rank == 0)
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1])
        with pytest.raises(RuntimeError, match="cannot be combined"):
            net.train_model(loader, optimizer, scheduler, epochs1=0, epochs2=0, graphed=True, sharded=True)
        # ---- fewer samples than ranks.  Rank 1 holds NO sample; with ignore_kl its loss share has no graph --
        # it must skip backward() and still join the gradient all-reduce (zeros), not raise while rank 0 waits inside it
        for ignore_kl in (True, False):
            opt.zero_grad(set_to_none=True)
            total = parallel.mc_sharded_loss(net, x, y, n=120, n_samples=1, base_seed=9, ignore_kl=ignore_kl)
            with torch.random.fork_rng(devices=[]):
                torch.manual_seed(parallel.sample_seed(9, 0))
                out = net.sequential(x)
            want = net.likelihood.mnll_batch_estimate(y, out.reshape(12, 1, 1), 120) + (0.0 if ignore_kl else net.kl)
            assert abs(float(total) - float(want)) <= 1e-6 * abs(float(want)), (ignore_kl, float(total), float(want))
            grads = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
            both = [torch.zeros_like(grads) for _ in range(world)]
            dist.all_gather(both, grads)
            assert torch.equal(both[0], both[1]) and float(grads.abs().max()) > 0
        # ... and train_model does not switch to sample sharding by itself when there is less than one sample per rank (the
        # default train_samples = 1: a data-parallel caller feeding each rank its own batches keeps its behaviour)
        net.train_samples = 1
        calls.clear()
        parallel.all_reduce_grads = lambda module, average=True: (calls.append(average), real(module, average))[1]
        try:
            net.train_model(loader, optimizer, scheduler, epochs1=1, epochs2=0)
        finally:
            parallel.all_reduce_grads = real
        assert calls == [], calls
        open(os.path.join(tmp, f"train_ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_sharded_training_step_world_size_2_gloo(tmp_path):
    """``parallel.mc_sharded_loss`` -- MC samples of a TRAINING step sharded over two ranks, one
    all-reduce (sum) of the parameter gradients.  Loss, gradients and the parameters after each of three Adam steps
    equal (1e-6) a single-process step over the union of both ranks' samples; the replicated parameters are bit-equal
    across ranks; ``train_model`` uses the sharded step inside a process group."""
    port = _free_port()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "train_ok0").exists() and (tmp_path / "train_ok1").exists()


def test_bench_two_ranks_gloo_cpu_plumbing(tmp_path):
    """bench.py's N > 1 control flow (barrier, max-over-ranks, single JSON line) on 2 CPU ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WHVI_BENCH_CPU_PLUMBING="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    # the N > 1 extras: MC-sharded network pass with one all-gather (tiny sizes in plumbing mode)
    sharded = rec["extras_multi_gpu"]["whviregression_3_1024_1024_1_mc128_sharded"]
    assert sharded["prediction_shape"] == [16, 1, 4] and sharded["mc_samples_per_gpu"] == 2 and sharded["ms"] > 0
    # the line says which backend and how many ranks the collectives saw
    assert rec["distributed"]["backend"] == "gloo" and rec["distributed"]["world_size"] == 2


def test_bench_plain_command_starts_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` as typed -- no launcher, no WORLD_SIZE: bench.py starts the two ranks itself as child
    processes (torch.distributed.run), passes rank 0's ONE JSON line through and exits with the children's code.  The
    line proves what ran: one entry per rank in ``ranks_seen`` (distinct devices / processes), every rank's own time,
    and a checksum of the all-gathered predictions that all ranks computed identically (CPU plumbing mode: gloo)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WHVI_BENCH_CPU_PLUMBING"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["value"] > 0
    assert len(rec["ranks_seen"]) == 2 and rec["distinct_devices"] == 2
    assert len(rec["per_rank"]["ms_per_step_wall"]) == 2 and all(v > 0 for v in rec["per_rank"]["ms_per_step_wall"])
    assert abs(max(rec["per_rank"]["ms_per_step_wall"]) - rec["ms_per_step"]) < 1e-3
    check = rec["sharded_prediction_check"]
    assert check["identical_on_all_ranks"] is True and check["prediction_shape"] == [33, 1, 8] and check["values_finite"]
    assert check["mean_std_over_samples"] > 0 and len(check["sha256_rank0"]) == 64
    assert rec["extras_multi_gpu"]["whviregression_3_1024_1024_1_mc128_sharded"]["gathered_predictions_identical_on_all_ranks"] is True
    # a failing rank is an exit code, not a silent success: an impossible row shape makes the ranks raise
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--log2d", "-3"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]
