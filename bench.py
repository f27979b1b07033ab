#!/usr/bin/env python3
"""bench.py -- headline benchmark of the WHVI hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (any N: for N > 1 without a launcher's environment this process
                                                            starts the N ranks itself, as fresh child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1 under a launcher)

Metric (BASELINE.json): batched FWHT Gtransforms/s (one transform = one length-D row) with the
achieved HBM GB/s against the roofline.  Workload = the configuration the metric's target is
quoted on: D = 4096, fp32, batch 8192 x 128 MC samples = 2^20 rows (16 GiB), resident in HBM,
transformed IN PLACE -- one "step" = one pass of the kernel over the whole buffer.  Weak scaling:
every rank owns its own 2^20 rows (rows are independent, no data-path collective).

Rank 0 prints ONE JSON line.  Besides the driver's contract fields it carries
  roofline     : algorithmic bytes (2 * D * 4 per transform) / the kernel's average
                 launch duration measured live with HIP events on the launch stream, vs 8 TB/s;
                 ``traffic`` is the PMC-measured HBM bytes per launch recorded under profiles/.
  cpu_baseline : the reference's own C++ FWHT (oracle/_ref, kind "reference") -- or the C
                 restatement (kind "port") when that binary is absent -- timed on this host's
                 cores on a bounded sample of the same workload (N = 1, rank 0 only).
  extras       : the D in {512..4096} sweep of the metric, fp16, the fused S.H.G.H.S kernel
                 (BASELINE config 3) and WHVILinear(512,512) forward+KL (config 2); N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
LOG2D = 12                     # D = 4096
ROWS = 8192 * 128              # batch x MC samples = 2^20 transforms per GPU
CPU_PLUMBING = os.environ.get("WHVI_BENCH_CPU_PLUMBING") == "1"   # tests/test_parallel.py only


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=ROWS, help="transforms per GPU (default 2^20)")
    ap.add_argument("--log2d", type=int, default=LOG2D)
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def self_launch(n_gpus):
    """``python bench.py --gpus N`` typed without a launcher (no WORLD_SIZE in the environment): start the N ranks as
    FRESH CHILD PROCESSES through torch.distributed.run -- before this process has touched the GPU (nothing above calls
    torch.cuda / HIP; a child, never an exec of this process) -- pass rank 0's single JSON line through on stdout and
    return the launcher's exit code.  The children see WORLD_SIZE and take the under-a-launcher path."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] --gpus {n_gpus} without a launcher: starting {n_gpus} ranks: {' '.join(cmd)}")
    child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)           # stderr passes through
    lines = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
    for ln in child.stdout.splitlines():
        if not ln.startswith("{"):
            log(ln)
    if lines:
        print(lines[-1], flush=True)
    if child.returncode == 0 and not lines:
        log("[bench] the ranks exited 0 without a JSON line")
        return 1
    return child.returncode


def setup_dist(n_gpus):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n_gpus:
        raise SystemExit(f"bench.py: --gpus {n_gpus} but WORLD_SIZE={world}")
    if CPU_PLUMBING:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py: no GPU visible")
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if CPU_PLUMBING:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    return rank, world, device


def fence(device, world):
    if world > 1:
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)


RESCALE_EVERY = 36   # steps between rescales: 36 * log2(sqrt(4096)) = 216 octaves of fp32's 253


def make_step(x):
    """One pass of the hot path over the resident buffer, in place.  FWHT o FWHT = D * identity, so the
    data grows by sqrt(D) per step; every RESCALE_EVERY steps the buffer is scaled back by an exact power
    of two.  That extra elementwise pass is INSIDE the timed region (it costs ~3 % at most and never
    happens for K + W <= 36): nothing is skipped, and the values stay finite for any step count."""
    if x.device.type == "cuda":
        from whvi_amd import _hip
        count = [0]
        log2d = x.size(1).bit_length() - 1
        back = 2.0 ** -((log2d * RESCALE_EVERY) // 2)

        def gpu_step():
            _hip.fwht_rows(x, out=x)
            count[0] += 1
            if count[0] % RESCALE_EVERY == 0:
                x.mul_(back)
        return gpu_step
    import fwht_cpp   # CPU plumbing mode (tests only): the host library, never the oracle

    def step():
        x.copy_(fwht_cpp.forward(x))
    return step


def timed(step, steps, warmup, device, world):
    """(wall seconds for exactly `steps` steps, max over ranks; HIP-event ms per launch on this rank)"""
    for _ in range(warmup):
        step()
    fence(device, world)
    use_ev = device.type == "cuda"
    if use_ev:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()          # torch's current stream == the stream the C ABI launches on
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if use_ev:
        ev1.record()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1) / steps if use_ev else wall * 1e3 / steps
    per_rank = None
    if world > 1:
        # every rank's own numbers (wall incl. the closing barrier, and its kernel time), not only the maximum
        mine = torch.tensor([wall * 1e3 / steps, ev_ms], dtype=torch.float64, device=device)
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        per_rank = {"ms_per_step_wall": [round(float(b[0]), 4) for b in both],
                    "ms_per_step_kernel_events": [round(float(b[1]), 4) for b in both]}
        wall = max(float(b[0]) for b in both) * steps / 1e3
    return wall, ev_ms, per_rank


def event_ms(fn, iters=10, warm=2, warm_ms=0.0):
    """Average ms of ``fn`` over ``iters`` launches after ``warm`` untimed ones -- and, for kernels of a fraction of a
    millisecond, after at least ``warm_ms`` of them: the clocks take ~30 ms of continuous work to ramp, which 20 launches
    of a 0.2 ms kernel do not provide (the first shape measured in a section would read 10 % low)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if warm_ms > 0:
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < warm_ms:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


KERNEL_SOURCES = ("whvi_amd/csrc/kernels.hpp", "whvi_amd/csrc/fwht_tile.hpp", "whvi_amd/csrc/dispatch.hpp",
                  "whvi_amd/csrc/tuning.hpp", "whvi_amd/csrc/Makefile")     # (the sources of the kernels PMC records exist for)


def kernel_source_hash():
    """sha256 over the kernel sources + build flags: ties a PMC record under profiles/ to the build it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def event_ms_each(fn, between, iters=10, warm=2):
    """Average ms of ``fn`` alone when every call is followed by ``between(i)`` (untimed housekeeping on the stream):
    one HIP-event pair per call."""
    for i in range(warm):
        fn()
        between(i)
    torch.cuda.synchronize()
    pairs = []
    for i in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        between(i)
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in pairs) / iters


def _finite(t):
    """Cheap finiteness probe of a large buffer after a timed run: every 4099th row (plus the last)."""
    rows = t.reshape(-1, t.shape[-1])
    return bool(torch.isfinite(rows[::4099].float()).all()) and bool(torch.isfinite(rows[-1].float()).all())


def recorded_traffic(workload_key, kernel_symbol):
    """(HBM bytes per launch, note) from the rocprofv3 PMC passes committed under profiles/ (collected and corrected as
    MI355X_MICROARCH.md prescribes: separate --pmc passes, FETCH_SIZE doubled; tools/update_hbm_traffic.py).  The
    record carries the kernel symbol it was measured on and a hash of the kernel sources; when either differs from
    what this run launched the counters say nothing about this build and ``traffic`` is null."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        rec = json.load(open(path)).get(workload_key)
    except (OSError, ValueError):
        rec = None
    if not rec:
        return None, f"no PMC record for {workload_key} in profiles/hbm_traffic.json"
    if rec.get("kernel_symbol") != kernel_symbol:
        return None, f"PMC record is for {rec.get('kernel_symbol')!r}, this run launched {kernel_symbol!r}"
    if rec.get("source_sha256") != kernel_source_hash():
        return None, "kernel sources changed since the PMC passes in profiles/hbm_traffic.json: re-collect them"
    return rec["hbm_bytes_per_launch"], f"PMC passes of this build: {rec.get('source', 'profiles/')}"


def cpu_baseline(log2d, target_s=25.0):
    """Time the reference's C++ FWHT (or the C restatement) on a bounded sample of the workload."""
    import numpy as np
    import oracle
    d = 1 << log2d
    ref = oracle.load_reference_cpp()
    g = torch.Generator().manual_seed(0)
    if ref is not None:
        kind, cores = "reference", torch.get_num_threads()
        run = lambda t: ref.forward(t)                       # noqa: E731
    else:
        oracle.build()
        kind, cores = "port", 1
        run = lambda t: oracle.fwht(t.numpy())               # noqa: E731
    rows = 64
    x = torch.randn(rows, d, generator=g)
    t0 = time.perf_counter()
    run(x)
    per_row = (time.perf_counter() - t0) / rows
    rows = int(max(64, min(1 << 16, target_s / max(per_row, 1e-9))))
    x = torch.randn(rows, d, generator=g)
    t0 = time.perf_counter()
    run(x)
    dt = time.perf_counter() - t0
    # BASELINE config 1 (benchmarks/walsh.py shape): D = 512, batch 1024, on the host
    c1 = None
    try:
        x1 = torch.randn(1024, 512, generator=g)
        run(x1)
        t1 = time.perf_counter()
        for _ in range(3):
            run(x1)
        c1 = {"reference_ms": round((time.perf_counter() - t1) / 3 * 1e3, 3)}
    except Exception as err:
        c1 = {"error": repr(err)}
    native = None
    try:   # the build's own host-tensor FWHT (libwhvi_cpu.so, OpenMP over rows), same rows: the "fair" CPU number
        import fwht_cpp
        xs = torch.randn(1 << 15, d, generator=g)
        fwht_cpp.forward(xs[:1024])
        # the library call itself into a preallocated result (fwht_cpp.forward allocates a fresh 512 MiB tensor per
        # call, and first-touch page faults from 256 threads then dominate: 14 GB/s); best of 3
        from whvi_amd import _cpu
        out = torch.empty_like(xs)
        dn = float("inf")
        for _ in range(3):
            t1 = time.perf_counter()
            rc = _cpu.lib().whvi_cpu_fwht_f32(out.data_ptr(), xs.data_ptr(), xs.size(0), xs.size(1))
            dn = min(dn, time.perf_counter() - t1)
        assert rc == 0 and torch.equal(out[:8], fwht_cpp.forward(xs[:8]))
        if c1 is not None and "error" not in c1:
            fwht_cpp.forward(x1)
            t2 = time.perf_counter()
            for _ in range(3):
                fwht_cpp.forward(x1)
            c1["native_openmp_library_ms"] = round((time.perf_counter() - t2) / 3 * 1e3, 3)
        native = {"Gtransforms_per_s": xs.size(0) / dn / 1e9, "GB_per_s_algorithmic": xs.numel() * 8 / dn / 1e9,
                  "threads": os.cpu_count(), "sample": f"{xs.size(0)} rows of D={d} fp32, out of place"}
    except Exception as err:
        native = {"error": repr(err)}
    return {"value": rows / dt / 1e9, "unit": "Gtransforms/s", "cores": cores, "kind": kind,
            "native_openmp_library": native, "config1_D512_batch1024_host": c1,
            "sample": f"{rows} rows of D={d} fp32 (same row shape as the GPU workload), one call, {dt:.1f} s; "
                      + ("reference src/fwht/cpp/fwht.cpp compiled into oracle/_ref, "
                         f"{cores} torch threads of {os.cpu_count()} host CPUs" if kind == "reference"
                         else "scalar C restatement oracle/fwht_oracle.c"),
            "gb_per_s_algorithmic": rows * 2 * d * 4 / dt / 1e9}


def _rate(rows, d, elem_bytes, ms):
    gbs = rows * 2 * d * elem_bytes / (ms * 1e-3) / 1e9
    return {"rows": rows, "ms": round(ms, 4), "Gtransforms_per_s": round(rows / (ms * 1e-3) / 1e9, 4),
            "GB_per_s": round(gbs, 1), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_launch": rows * 2 * d * elem_bytes}


def _extra_sweep(device):
    """The metric's D in {512..4096} sweep, 4 GiB of fp32 per size, in place."""
    from whvi_amd import _hip
    out = {}
    for log2d in (9, 10, 11, 12):
        d = 1 << log2d
        rows = (1 << 32) // (4 * d)
        x = torch.randn(rows, d, device=device) * 2.0 ** -64
        out[f"D={d}"] = _rate(rows, d, 4, event_ms(lambda: _hip.fwht_rows(x, out=x), iters=10, warm=10))
        out[f"D={d}"].update(kernel=_hip.last_kernel(), values_finite=_finite(x))
        del x
    return out


def _extra_copy_reference(device):
    """Context for the roofline fractions: what the SAME HBM gives two memory-bound operations that compute nothing -- the
    vendor's device-to-device copy (``Tensor.copy_``, 4 GiB read + 4 GiB written) and torch's in-place elementwise kernel on
    a 4 GiB buffer (read + write of the same addresses, the transform's access pattern)."""
    x = torch.randn(1 << 30, device=device)
    y = torch.empty_like(x)
    copy_ms = event_ms(lambda: y.copy_(x), iters=10, warm=3)
    inplace_ms = event_ms(lambda: x.mul_(1.0), iters=10, warm=3)
    gb = 2 * x.numel() * 4 / 1e6
    return {"copy_4GiB_out_of_place_GB_per_s": round(gb / copy_ms, 1), "copy_frac_of_peak": round(gb / copy_ms / HBM_PEAK_GBS, 4),
            "torch_inplace_elementwise_4GiB_GB_per_s": round(gb / inplace_ms, 1),
            "torch_inplace_frac_of_peak": round(gb / inplace_ms / HBM_PEAK_GBS, 4),
            "note": "bytes counted as read + written, like the transform's algorithmic bytes"}


def _extra_long_rows(device):
    """Rows beyond one wavefront tile (outside the metric's D range): one block of 4 / 8 / 16 waves per row, one pass,
    4 GiB in place; every launch is followed by an untimed rescale that brings the values back to where they started."""
    from whvi_amd import _hip
    out = {}
    for dtype, log2d in ((torch.float32, 14), (torch.float32, 15), (torch.float32, 16), (torch.float64, 15), (torch.float16, 16)):
        d, esize = 1 << log2d, torch.empty(0, dtype=dtype).element_size()
        rows = (1 << 32) // (esize * d)
        x = (torch.randn(rows, d, device=device) * 0.25).to(dtype)
        ms = event_ms_each(lambda: _hip.fwht_rows(x, out=x),
                           lambda i: x.mul_(2.0 ** (-(log2d // 2) - (i & 1) * (log2d & 1))), iters=8, warm=12)
        key = f"{str(dtype)[6:]}_D={d}"
        out[key] = _rate(rows, d, esize, ms)
        out[key].update(kernel=_hip.last_kernel(), values_finite=_finite(x))
        del x
    out["note"] = "round 1 ran these as 4096-element pieces + high-bit passes at 3.0 TB/s (fp16: unsupported beyond D = 8192)"
    return out


def _extra_f16(device):
    """BASELINE config 5 (one GPU's share): D = 4096 fp16, 2^20 rows = 8 GiB."""
    from whvi_amd import _hip
    x = torch.empty(1 << 20, 4096, device=device, dtype=torch.float16)
    x.view(256, 4096, 4096).copy_((torch.randn(4096, 4096, device=device) * 2.0 ** -8).half())
    # one in-place transform multiplies the data's magnitude by 64 and fp16 tops out at 65504: every launch is followed
    # by an exact 2^-6 rescale (untimed: each transform has its own event pair).  Finite data matters for the number:
    # on overflowed (inf / NaN) or all-zero data the same launch runs 1.5-2 % faster (tools/probe_f16_data.py: lower
    # switching power, higher clock) -- round 1's 6.39 TB/s was measured on data that had overflowed.
    res = _rate(1 << 20, 4096, 2, event_ms_each(lambda: _hip.fwht_rows(x, out=x), lambda i: x.mul_(2.0 ** -6), iters=12, warm=12))
    res.update(kernel=_hip.last_kernel(), values_finite=_finite(x))
    _attach_traffic(res, "fwht_f16_D4096_rows1048576")
    return res


def _attach_traffic(res, workload_key):
    """``traffic`` (HBM bytes per launch from the PMC passes under profiles/, FETCH_SIZE doubled + WRITE_SIZE) for an extras
    entry, under the headline's rule: only when the record's kernel symbol AND kernel-source hash are this build's."""
    value, note = recorded_traffic(workload_key, res.get("kernel"))
    res["traffic"], res["traffic_note"] = value, note
    if value is not None and res.get("algorithmic_bytes_per_launch"):
        res["traffic_over_algorithmic"] = round(value / res["algorithmic_bytes_per_launch"], 5)


def _extra_fused(device):
    """BASELINE config 3: fused S.H.diag(g).H.S, D = 2048, 64 MC samples, batch 8192 (4 GiB), in place."""
    from whvi_amd import _hip
    d, S, B = 2048, 64, 8192
    x = torch.randn(B * S, d, device=device)
    # S1, S2 = random signs scaled by D^-1/2 (the sign-flip matrices of the WHVI parameterisation) and g ~ N(0, 1): one
    # launch then preserves the data's norm in expectation, so 50 in-place launches stay in range
    a = (torch.randint(0, 2, (d,), device=device).float() * 2 - 1) * d ** -0.5
    c = (torch.randint(0, 2, (d,), device=device).float() * 2 - 1) * d ** -0.5
    g = torch.randn(S, d, device=device)
    ms = event_ms(lambda: _hip.fused_shs(x, a, g, c, axis="col", n_samples=S, sample_stride=1, out=x), iters=20, warm=30)   # clocks take ~25 launches to ramp
    res = _rate(B * S, d, 4, ms)
    res.update(kernel=_hip.last_kernel(), values_finite=_finite(x), max_abs_after_50_launches=float(x[::4099].abs().max()))
    res["note"] = ("one fused launch = 2 FWHTs + 3 scalings per row; unfused (2 FWHT launches + 3 elementwise) "
                   "moves 5x the bytes")
    _attach_traffic(res, "fused_shs_f32_D2048_S64_B8192")
    return res


def _extra_fastfood(device):
    """The opt-in fastfood Module (WHVILinear(D, D, mode="fastfood"): the textbook S1.H.diag(g).H.S2 applied to activations
    through ONE fused launch, config 3's kernel) beside the reference-equivalent layer of the same shape (S weight
    matrices of D x D + a batched GEMM): D = 2048, 64 MC samples, batch 8192 -- config 3's shape -- forward only."""
    from whvi_amd.layers import WHVILinear
    D, S, B = 2048, 64, 8192
    x = torch.randn(B, D, device=device)
    out = {}
    for mode in ("fastfood", "reference"):
        layer = WHVILinear(D, D, mode=mode).to(device)
        res = [None]

        def run():
            with torch.no_grad():
                res[0] = layer.forward_mc(x, S)
        ms = event_ms(run, iters=5, warm=3)
        out[mode + "_ms"] = round(ms, 3)
        out[mode + "_values_finite"] = _finite(res[0])
        res[0] = None
        del layer
        torch.cuda.empty_cache()
    out["note"] = ("fastfood = O(D log D) per row, no D x D weight (not reference-equivalent: the reference's weight is, as "
                   "written, diagonal); reference = whvi_wbar_fwd for 64 matrices + one 4.4 TFLOP batched GEMM")
    return out


def _extra_wbar_fwd(device):
    """The weight construction (whvi_wbar_fwd, src/weights.py:73 and the mean + sample sum of :93): no HBM read, one write
    of the matrices.  D = 2048 x 64 matrices (1 GiB) without and with the mean matrix (the `forward_mc` form, whose
    16 MiB mean matrix is re-read by every sample: XCD-sliced block order), and config 2's weights (D = 512 x 32, 32 MiB)."""
    from whvi_amd import _hip
    out = {}
    for key, (J, S, D, with_mean) in (("D2048_x64_1GiB", (1, 64, 2048, False)), ("D2048_x64_plus_mean_1GiB", (1, 64, 2048, True)),
                                      ("D512_x32_plus_mean_config2", (1, 32, 512, True))):
        s1, s2, u = torch.randn(J, D, device=device), torch.randn(J, D, device=device), torch.randn(J, 1 + S, D, device=device)
        base = _hip.wbar_fwd(s1, u, s2, D, first=0, count=1).view(J, D, D) if with_mean else None
        res = [None]

        def run():
            res[0] = _hip.wbar_fwd(s1, u, s2, D, base=base, first=1)
        ms = event_ms(run, iters=20, warm=20, warm_ms=50)
        gbs = J * S * D * D * 4 / (ms * 1e-3) / 1e9
        out[key] = {"matrices": J * S, "D": D, "ms": round(ms, 4), "GB_per_s_written": round(gbs, 1),
                    "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4), "kernel": _hip.last_kernel(),
                    "values_finite": bool(torch.isfinite(res[0][:, ::7]).all())}
        res[0] = None
    return out


def _extra_wbar_bwd(device):
    """Backward of the weight construction (whvi_wbar_bwd, src/weights.py:73 under autograd): one launch reads dL/dW
    once.  D = 2048 x 64 matrices (1 GiB of gradient; the fused kernel's config-3 shape) and 4 GiB, and config 2's
    backward (D = 512 x 32 samples, 32 MiB: cache-resident, launch-latency-sized).  Algorithmic bytes = the gradient
    read once (the 3 x D outputs per matrix are 1/D of it)."""
    from whvi_amd import _hip
    out = {}
    for key, (J, S, D, mean) in (("D2048_x64_1GiB", (1, 64, 2048, False)), ("D2048_x64_mean_1GiB", (1, 64, 2048, True)),
                                 ("D2048_x256_4GiB", (1, 256, 2048, False)), ("D512_x32_mean_config2", (1, 32, 512, True))):
        U = S + 1 if mean else S
        s1, s2, u = torch.randn(J, D, device=device), torch.randn(J, D, device=device), torch.randn(J, U, D, device=device)
        gw = torch.randn(J, S, D, D, device=device)
        res = [None]

        def run():
            res[0] = _hip.wbar_bwd(gw, s1, u, s2, mean=mean)
        ms = event_ms(run, iters=20, warm=20, warm_ms=50)
        gbs = gw.numel() * 4 / (ms * 1e-3) / 1e9
        out[key] = {"matrices": J * S, "D": D, "ms": round(ms, 4), "GB_per_s": round(gbs, 1),
                    "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4), "kernel": _hip.last_kernel(),
                    "values_finite": bool(torch.isfinite(res[0][:, :, 1:] if mean else res[0]).all())}
        del gw
    return out


def _extra_diag_apply(device):
    """whvi_diag_apply / whvi_diag_apply_bwd on the layer shapes of configs 2 and 4: algorithmic bytes (x read unless shared by
    all samples, out written; backward: g and x read, grad_x written) / HIP-event time."""
    from whvi_amd import _hip
    out = {}
    for key, (D, S, B, shared) in (("config2_D512_S32_B4096_shared_x", (512, 32, 4096, True)),
                                   ("config4_D1024_S16_B45730", (1024, 16, 45730, False)),
                                   ("D2048_S16_B8192", (2048, 16, 8192, False)), ("D4096_S16_B4096", (4096, 16, 4096, False))):
        s1, s2, u, bias = (torch.randn(n, device=device) for n in ((D,), (D,), (1 + S, D), (D,)))
        x = torch.randn((B, D) if shared else (S, B, D), device=device)
        res = torch.empty(S, B, D, device=device)
        ms = event_ms(lambda: _hip.diag_apply(x, s1, s2, u, bias, n_samples=S, out=res), iters=20, warm=5, warm_ms=30.0)
        kernel = _hip.last_kernel()
        nbytes = res.numel() * 4 + (0 if shared else x.numel() * 4)
        g = torch.randn(S, B, D, device=device)
        ms_b = event_ms(lambda: _hip.diag_apply_bwd(g, x, s1, s2, u, n_samples=S, need_grad_x=not shared), iters=10, warm=3, warm_ms=30.0)
        nb = (2 if shared else 3) * g.numel() * 4 + (x.numel() * 4 if shared else 0)
        out[key] = {"fwd_ms": round(ms, 4), "fwd_GB_per_s": round(nbytes / ms / 1e6, 1), "fwd_frac_of_peak": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                    "bwd_ms": round(ms_b, 4), "bwd_GB_per_s": round(nb / ms_b / 1e6, 1), "bwd_frac_of_peak": round(nb / ms_b / 1e6 / HBM_PEAK_GBS, 4),
                    "bwd_kernel": _hip.last_kernel(), "kernel": kernel, "values_finite": _finite(res)}
        if max(nbytes, nb) <= 320 << 20:
            out[key]["note"] = ("working set of about the 256 MiB Infinity Cache: part of it is served from there, so a rate can "
                                "exceed the HBM peak -- a cache rate, not an HBM fraction")
        del x, res, g
    return out


def _extra_layer_apply(device):
    """Config 4's outer layers as their own launches: the stacked 3 -> 1024 layer's narrow-input product (write-only: 16 x 45 730
    x 1024 floats out of a (45 730, 4) input and (16, 1024, 4) weights) and the column 1024 -> 1 layer's row dot (read-only),
    beside the rocBLAS calls (torch.matmul) they replace."""
    from whvi_amd import _hip
    S, B, N = 16, 45730, 1024
    x, W = torch.randn(B, 4, device=device), torch.randn(S, N, 4, device=device)
    h, w = torch.randn(S, B, N, device=device), torch.randn(S, N, device=device)
    nbytes = S * B * N * 4
    out = {}
    ms = event_ms(lambda: _hip.small_k_apply(x, W), iters=20, warm=5, warm_ms=30.0)
    out["small_k_apply_K4_N1024"] = {"ms": round(ms, 4), "GB_per_s_written": round(nbytes / ms / 1e6, 1), "frac_of_peak": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                                     "kernel": _hip.last_kernel(),
                                     "torch_matmul_ms": round(event_ms(lambda: torch.matmul(x, W.transpose(1, 2)), iters=20, warm=5), 4)}
    ms = event_ms(lambda: _hip.row_dot(h, w), iters=20, warm=5, warm_ms=30.0)
    out["row_dot_D1024"] = {"ms": round(ms, 4), "GB_per_s_read": round(nbytes / ms / 1e6, 1), "frac_of_peak": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                            "kernel": _hip.last_kernel(),
                            "torch_matmul_ms": round(event_ms(lambda: torch.matmul(h, w.unsqueeze(-1)), iters=20, warm=5), 4)}
    return out


def _extra_layer(device):
    """BASELINE config 2: WHVILinear(512, 512) forward + KL, 32 MC samples, batch 4096, fp32."""
    from whvi_amd.layers import WHVILinear
    layer = WHVILinear(512, 512).to(device)
    h = torch.randn(4096, 512, device=device)

    def loop():                               # the outputs are produced and dropped: no reduction pass is timed with them
        with torch.no_grad():
            for _ in range(32):
                out = layer(h)
            return out, layer.kl

    def batched():
        with torch.no_grad():
            return layer.forward_mc(h, 32), layer.kl

    def train():                              # forward + KL + backward (whvi_wbar_bwd, whvi_reparam_kl_bwd)
        layer.zero_grad(set_to_none=True)
        (layer.forward_mc(h, 32).square().mean() + layer.kl).backward()
    # 30 ms of continuous work first: the GEMMs are clock-sensitive and a handful of sub-millisecond passes do not ramp them
    out = {}
    for route in ("auto", "faithful"):        # auto (shipped default): whvi_diag_apply; faithful: weight construction + GEMM
        layer.weight_submodule.faithful_dataflow = route == "faithful"
        ms_loop, ms_batched = event_ms(loop, iters=5, warm=2, warm_ms=30.0), event_ms(batched, iters=20, warm=5, warm_ms=30.0)
        ms_train = event_ms(train, iters=10, warm=3, warm_ms=30.0)
        with torch.no_grad():
            finite = _finite(layer.forward_mc(h, 32)) and all(bool(torch.isfinite(p.grad).all()) for p in layer.parameters())
        res = {"values_finite": finite, "loop_ms": round(ms_loop, 3), "loop_ms_per_mc_sample": round(ms_loop / 32, 4),
               "batched_ms": round(ms_batched, 3), "batched_ms_per_mc_sample": round(ms_batched / 32, 4),
               "batched_fwd_kl_bwd_ms": round(ms_train, 3)}
        if route == "auto":
            out.update(res)
            out["batched_GB_per_s_written"] = round(32 * 4096 * 512 * 4 / ms_batched / 1e6, 1)
        else:
            out["faithful_dataflow"] = res
    layer.weight_submodule.faithful_dataflow = False
    out["modes"] = ("loop = the reference's one forward per MC sample; batched = forward_mc for all 32 samples; fwd_kl_bwd adds "
                    "the backward pass.  Top level = the shipped GPU route (one whvi_reparam_kl launch + ONE whvi_diag_apply launch "
                    "per layer call: the as-written weight is exactly diagonal and is applied as such, same values); "
                    "faithful_dataflow = weight construction through the FWHT kernels + rocBLAS GEMMs, the reference's "
                    "dataflow op for op")
    return out


def _extra_network(device):
    """BASELINE config 4 (one GPU's share): WHVIRegression 3 -> 1024 -> 1024 -> 1, protein-sized synthetic
    batch (45 730 x 3), 128 MC samples over 8 GPUs = 16 per GPU, predictive forward."""
    import torch.nn as nn
    from whvi_amd.layers import WHVILinear
    from whvi_amd.networks import WHVIRegression
    net = WHVIRegression([WHVILinear(3, 1024), nn.ReLU(), WHVILinear(1024, 1024), nn.ReLU(), WHVILinear(1024, 1)],
                         eval_samples=16).to(device).eval()
    xb = torch.randn(45730, 3, device=device)
    res = {}

    def predict():
        with torch.no_grad():
            return net(xb)
    for route in ("auto", "faithful"):
        net.set_faithful_dataflow(route == "faithful")
        part = {}
        for mode in ("batched", "loop"):
            net.mc_mode = mode
            part[mode + "_ms"] = round(event_ms(predict, iters=3, warm=1), 3)
            part["values_finite"] = part.get("values_finite", True) and bool(torch.isfinite(predict()).all())
        if route == "auto":
            res.update(part)
        else:
            res["faithful_dataflow"] = part
    net.set_faithful_dataflow(False)
    net.mc_mode = "batched"
    # the same pass with the stacked layer's 4 x 256 parameter vectors packed into four tensors (net.pack_parameters(): same
    # values, reference state_dict keys kept): without it every pass gathers 1024 tiny leaves
    net.pack_parameters()
    res["batched_packed_parameters_ms"] = round(event_ms(predict, iters=3, warm=1), 3)
    # ... and replayed from a hipGraph (whvi_amd.graphs.GraphedPredictor: the ~40 launches of the pass in one; fresh eps per replay)
    try:
        from whvi_amd.graphs import GraphedPredictor
        gp = GraphedPredictor(net, xb, 16)
        res["batched_hipgraph_replay_ms"] = round(event_ms(lambda: gp(xb), iters=5, warm=2), 3)
        res["values_finite"] = res["values_finite"] and bool(torch.isfinite(gp(xb)).all())
        del gp
    except Exception as err:                      # noqa: BLE001
        res["batched_hipgraph_replay_ms"] = f"failed: {err!r}"
    res["config"] = "batch 45730 x 3, 16 MC samples (the per-GPU share of 128 over 8 GPUs), fp32, eval forward"
    res["note"] = ("top level = the shipped route: the 1024 x 1024 middle layer applies its (exactly diagonal) as-written weight "
                   "in one whvi_diag_apply launch (3 GB read + 3 GB written, both nn.ReLU passes folded into it) instead of 16 "
                   "weight matrices + a 1.5 TFLOP fp32 GEMM (faithful_dataflow); the stacked 3 -> 1024 layer (K = 4 product: "
                   "whvi_small_k_apply) and the column 1024 -> 1 layer (row dot: whvi_row_dot) keep their as-written dataflow, one "
                   "HBM-bound launch each")
    return res


def _extra_toy(device):
    """The reference's toy regression (experiments/Toy example.ipynb:319,397: 3 WHVI layers 1 -> 128 -> 128 -> 1,
    ~100 points, 1 MC sample, Adam, with KL: 153.17 it/s on an unnamed GPU): training rate, and the
    launch-bound predictive pass (64 MC samples) eager vs hipGraph replay."""
    import torch.nn as nn
    from whvi_amd.graphs import GraphedPredictor
    from whvi_amd.layers import WHVILinear
    from whvi_amd.networks import WHVIRegression
    toy = WHVIRegression([WHVILinear(1, 128), nn.ReLU(), WHVILinear(128, 128), nn.ReLU(), WHVILinear(128, 1)],
                         train_samples=1).to(device).train()
    tx = torch.linspace(-2, 2, 100, device=device).unsqueeze(1)
    ty = torch.sin(3 * tx)
    opt = torch.optim.Adam(toy.parameters(), lr=1e-3)

    def train_step():
        opt.zero_grad()
        toy.loss(tx, ty, n=100).backward()
        opt.step()
    for _ in range(10):
        train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        train_step()
    torch.cuda.synchronize()
    rate = round(100 / (time.perf_counter() - t0), 1)
    # the same step replayed from a hipGraph (loss + backward + Adam in one launch), as a child process with its exit
    # code in the line: a failure of this optional number shows up here instead of costing the bench line.  (The
    # process abort of early round 1 -- capture with a stale autograd graph alive -- is a RuntimeError now:
    # whvi_amd/graphs.py, tests/test_fused_gpu.py::test_graphed_train_step_rejects_a_stale_autograd_graph.)
    import subprocess
    child_exit = None
    try:
        child = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "toy_graph_train.py")],
                               capture_output=True, text=True, timeout=180)
        child_exit = child.returncode
        line = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
        graph_rate = json.loads(line[-1])["it_per_s"] if line else f"failed: {child.stderr.strip().splitlines()[-1:]}"
    except Exception as err:
        graph_rate = f"failed ({err!r})"
    toy.eval()

    def predict():
        with torch.no_grad():
            return toy.forward_batched(tx, 64)
    eager_ms = event_ms(predict, iters=50, warm=5)
    gp = GraphedPredictor(toy, tx, 64)
    graph_ms = event_ms(lambda: gp(tx), iters=200, warm=5)
    return {"training_it_per_s_with_kl": rate, "training_it_per_s_with_kl_hipgraph": graph_rate,
            "hipgraph_child_exit_code": child_exit, "reference_published_it_per_s_with_kl": 153.17,
            "predict_64mc_eager_ms": round(eager_ms, 4), "predict_64mc_hipgraph_replay_ms": round(graph_ms, 4),
            "note": "eager training loop is launch-bound; the reference number is from its notebook on an "
                    "unspecified CUDA GPU"}


def _extra_config4_train(device):
    """BASELINE config 4's network under the reference's own training recipe (``train_model`` + ``make_optimizer``:
    src/networks.py:71-99, src/evaluation.py:15-27): ms per optimisation step for the eager loop, the packed parameter
    layout, and ``train_model(graphed=True)`` -- one hipGraph replay per step with the learning-rate schedule inside.
    A child process (tools/config4_train_step.py) with its exit code in the line."""
    import subprocess
    child = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "config4_train_step.py")],
                           capture_output=True, text=True, timeout=600)
    line = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
    out = json.loads(line[-1]) if line else {"error": child.stderr.strip().splitlines()[-3:]}
    out["child_exit_code"] = child.returncode
    out["recipe"] = "train_model two phases, Adam + LambdaLR stepped after every batch, batch 256, 1 MC sample, with KL"
    return out


def extras(device):
    """Secondary measurements (inputs resident, HIP-event timed).  Every section is independent: a failure is
    recorded under its own key and never costs the other numbers or the headline line."""
    out = {}
    for key, fn in (("fwht_f32_sweep_4GiB", _extra_sweep), ("hbm_copy_reference", _extra_copy_reference),
                    ("fwht_long_rows_4GiB", _extra_long_rows),
                    ("fwht_f16_D4096_2^20rows", _extra_f16),
                    ("fused_shs_D2048_S64_B8192", _extra_fused), ("fastfood_module_D2048_S64_B8192", _extra_fastfood),
                    ("wbar_fwd", _extra_wbar_fwd),
                    ("wbar_bwd", _extra_wbar_bwd),
                    ("diag_apply", _extra_diag_apply),
                    ("layer_apply", _extra_layer_apply),
                    ("whvilinear_512_fwd_kl_32mc_b4096", _extra_layer),
                    ("whviregression_3_1024_1024_1_mc16", _extra_network), ("toy_regression", _extra_toy),
                    ("config4_train_step", _extra_config4_train)):
        try:
            out[key] = fn(device)
        except Exception as err:
            out[key] = {"error": repr(err)}
        torch.cuda.empty_cache()
    return out


def _all_ok(ok, device):
    """True iff `ok` holds on every rank (so that ranks agree on entering a phase that contains collectives)."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def _timed_all_ranks(fn, iters, device):
    """Wall ms per call, barrier-fenced, max over ranks (same protocol as the headline)."""
    fn()
    fence(device, dist.get_world_size())
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    fence(device, dist.get_world_size())
    t = torch.tensor([(time.perf_counter() - t0) * 1e3 / iters], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _same_on_all_ranks(t, device):
    """(every rank holds the same bits of ``t``, this rank's sha256 of them)"""
    import hashlib
    digest = hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()
    word = torch.tensor([int(digest[:15], 16)], dtype=torch.int64, device=device)
    lo, hi = word.clone(), word.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(lo.item() == hi.item()), digest


def multi_gpu_extras(device, rank, world, small=False):
    """N > 1 only, every rank takes part: BASELINE config 5 (D = 4096 fp16, 2^20 rows in total, row-sharded: the
    strong-scaling curve) and config 4 (WHVIRegression 3 -> 1024 -> 1024 -> 1, 128 MC samples sharded over the
    ranks, ONE RCCL all-gather of the predictions per forward).  Ranks vote before each phase, so a local failure
    skips the phase everywhere instead of leaving the others inside a collective.  ``small``: the same code path at
    test sizes (tests/test_rccl_gpu.py drives it through a one-rank RCCL group)."""
    out = {}
    gpu = device.type == "cuda"
    rows_total = (1 << 12) if small else (1 << 20)
    # ---- config 5
    x16, err = None, None
    try:
        if gpu:
            from whvi_amd import _hip
            rows = rows_total // world
            x16 = (torch.randn(rows, 4096, device=device) * 2.0 ** -8).half()
    except Exception as e:                      # noqa: BLE001
        err = repr(e)
    if _all_ok(x16 is not None, device):
        # finite data throughout: an exact 2^-6 rescale follows every transform (fp16 tops out at 65504 and an in-place
        # transform multiplies the magnitude by 64); each transform has its own HIP-event pair, ranks start together
        # behind a barrier, and the slowest rank's per-launch time is reported
        rescale = lambda i: x16.mul_(2.0 ** -6)                                   # noqa: E731
        event_ms_each(lambda: _hip.fwht_rows(x16, out=x16), rescale, iters=2, warm=10)
        fence(device, world)
        mine = event_ms_each(lambda: _hip.fwht_rows(x16, out=x16), rescale, iters=10, warm=0)
        t = torch.tensor([mine], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
        total = rows * world
        out["fwht_f16_D4096_2^20rows_row_sharded"] = {
            "values_finite": _finite(x16), "kernel": _hip.last_kernel(),
            "rows_total": total, "rows_per_gpu": rows, "ms": round(ms, 4),
            "Gtransforms_per_s": round(total / ms / 1e6, 4), "GB_per_s_aggregate": round(total * 2 * 4096 * 2 / ms / 1e6, 1),
            "frac_of_aggregate_peak": round(total * 2 * 4096 * 2 / ms / 1e6 / (HBM_PEAK_GBS * world), 4),
            "scaling": "strong (2^20 rows in total)"}
    elif gpu:
        out["fwht_f16_D4096_2^20rows_row_sharded"] = {"skipped": err or "another rank failed to allocate"}
    x16 = None
    # ---- config 4
    net, err = None, None
    try:
        import torch.nn as nn
        from whvi_amd.layers import WHVILinear
        from whvi_amd.networks import WHVIRegression
        from whvi_amd.parallel import mc_sharded_forward
        width, batch, n_mc = (1024, 45730, 128) if gpu and not small else ((64, 33, 8) if gpu else (8, 16, 4))
        torch.manual_seed(4)                    # replicated parameters: same seed on every rank
        net = WHVIRegression([WHVILinear(3, width), nn.ReLU(), WHVILinear(width, width), nn.ReLU(),
                              WHVILinear(width, 1)], eval_samples=n_mc).to(device).eval()
        xb = torch.randn(batch, 3, device=device)
    except Exception as e:                      # noqa: BLE001
        err = repr(e)
    if _all_ok(net is not None, device):
        shape, keep = [], [None]

        def predict():
            with torch.no_grad():
                keep[0] = mc_sharded_forward(net, xb, n_mc, base_seed=1)
                shape[:] = list(keep[0].shape)
        ms = _timed_all_ranks(predict, 3, device)
        same, digest = _same_on_all_ranks(keep[0], device)
        keep[0] = None
        out["whviregression_3_1024_1024_1_mc128_sharded"] = {
            "ms": round(ms, 3), "prediction_shape": shape, "mc_samples_per_gpu": n_mc // world,
            "gathered_predictions_sha256_rank0": digest, "gathered_predictions_identical_on_all_ranks": same,
            "config": f"batch {batch} x 3, {n_mc} MC samples sharded over {world} ranks, one all-gather of "
                      f"(batch, 1, {n_mc // world}) blocks per forward, fp32, eval"}
    else:
        out["whviregression_3_1024_1024_1_mc128_sharded"] = {"skipped": err or "another rank failed to build the network"}
    # ---- config 4, one TRAINING step with the 128 MC samples sharded: local batched pass + backward, one all-reduce
    # (sum) of the O(D) parameter gradients, Adam (whvi_amd.parallel.mc_sharded_loss)
    opt, err = None, None
    try:
        if net is not None:
            from whvi_amd.parallel import mc_sharded_loss
            train_batch = 256 if gpu and not small else 16
            net.train()
            if gpu:
                net.pack_parameters()           # 13 parameter tensors instead of 1033: the host side of Adam
            opt = torch.optim.Adam(net.parameters(), lr=1e-4)
            xt = torch.randn(train_batch, 3, device=device)
            yt = torch.sin(xt.sum(dim=1, keepdim=True))
            counter = [0]
    except Exception as e:                      # noqa: BLE001
        err = repr(e)
    if _all_ok(opt is not None, device):
        def train_step():
            counter[0] += 1
            opt.zero_grad(set_to_none=True)
            mc_sharded_loss(net, xt, yt, n=45730, n_samples=n_mc, base_seed=counter[0])
            opt.step()
        ms = _timed_all_ranks(train_step, 5, device)
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        out["whviregression_3_1024_1024_1_mc128_sharded_train_step"] = {
            "ms": round(ms, 3), "mc_samples_per_gpu": n_mc // world, "values_finite": _finite(flat),
            "parameters_bit_equal_across_ranks": bool(torch.equal(lo, hi)),
            "config": f"batch {train_batch} x 3, {n_mc} MC samples sharded over {world} ranks, one all-reduce(sum) of "
                      f"{flat.numel()} parameter gradients per step, Adam, fp32, with KL"}
    else:
        out["whviregression_3_1024_1024_1_mc128_sharded_train_step"] = {"skipped": err or "another rank failed"}
    return out


def device_identity(device):
    """What distinguishes this rank's device from the others': the GPU's uuid and PCI address (host name + pid in the CPU
    plumbing mode of the tests)."""
    import socket
    if device.type != "cuda":
        return f"cpu:{socket.gethostname()}:pid{os.getpid()}"
    p = torch.cuda.get_device_properties(device)
    uuid = getattr(p, "uuid", None)
    pci = ":".join(f"{getattr(p, k):02x}" for k in ("pci_domain_id", "pci_bus_id", "pci_device_id") if hasattr(p, k))
    return f"cuda:{device.index} uuid={uuid} pci={pci} {p.name}"


def sharded_prediction_check(device, rank, world):
    """N > 1, always: the one exchange step of the path -- MC samples sharded over the ranks, ONE all-gather of the predictions
    (src/networks.py:47-51,112-114 is the single-process loop + reduction being sharded) -- on a small network.  Every rank
    checksums the gathered (batch, 1, S) tensor; the line reports rank 0's and whether all ranks computed the same bits,
    and that rank r's block of samples is what rank r alone computed."""
    import torch.nn as nn
    from whvi_amd.layers import WHVILinear
    from whvi_amd.networks import WHVIRegression
    from whvi_amd.parallel import mc_sharded_forward, shard_bounds
    n_mc = 4 * world
    torch.manual_seed(11)                           # replicated parameters and input
    net = WHVIRegression([WHVILinear(3, 64), nn.ReLU(), WHVILinear(64, 64), nn.ReLU(), WHVILinear(64, 1)],
                         eval_samples=n_mc)
    with torch.no_grad():                           # the initial s1, s2 ~ 0.01 N(0, 1) give outputs of ~1e-9: scale them up
        for name, p_ in net.named_parameters():
            if name.rsplit(".", 1)[-1] in ("s1", "s2"):
                p_.mul_(30.0)
    net = net.to(device).eval()
    xb = torch.randn(33, 3).to(device)
    with torch.no_grad():
        pred = mc_sharded_forward(net, xb, n_mc, base_seed=5)          # (33, 1, n_mc), identical on every rank
    same, digest = _same_on_all_ranks(pred, device)
    b, e = shard_bounds(n_mc, rank, world)
    spread = float(pred.std(dim=2).mean())           # the samples differ (ranks did not all draw the same eps)
    return {"prediction_shape": list(pred.shape), "sha256_rank0": digest, "identical_on_all_ranks": same,
            "mc_samples": n_mc, "samples_of_rank0": [b, e], "values_finite": bool(torch.isfinite(pred).all()),
            "mean_std_over_samples": spread,
            "what": "WHVIRegression 3->64->64->1, batch 33, MC samples sharded over the ranks, one all-gather"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    rank, world, device = setup_dist(args.gpus)
    d = 1 << args.log2d
    rows = args.rows if not CPU_PLUMBING else 256
    if device.type == "cuda":
        from whvi_amd import _hip
        _hip.lib()    # fail loudly before allocating anything if the native library is missing
    # synthetic input, resident before the timed region.  FWHT o FWHT = D * identity, so an in-place
    # run grows by sqrt(D) per step: start small enough that K + W steps stay finite in fp32.
    total_steps = min(args.steps + args.warmup, RESCALE_EVERY)
    scale_log2 = -min(120, (args.log2d * total_steps) // 2)
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    x = torch.randn(rows, d, device=device, generator=gen)
    x.mul_(2.0 ** scale_log2)
    step = make_step(x)

    wall, ev_ms, per_rank = timed(step, args.steps, args.warmup, device, world)
    finite = bool(torch.isfinite(x[:: max(1, rows // 64)]).all())
    if device.type == "cuda":
        kernel_symbol = _hip.last_kernel()       # what the library's dispatch actually launched for this shape
        traffic, traffic_note = recorded_traffic(f"fwht_f32_D{d}_rows{rows}", kernel_symbol)
    else:
        kernel_symbol, traffic, traffic_note = "host library (CPU plumbing mode)", None, "no GPU"
    value = world * rows * args.steps / wall / 1e9
    alg_bytes = rows * 2 * d * 4                      # per launch: read once + write once
    achieved = alg_bytes / (ev_ms * 1e-3) / 1e9
    # the measured ceiling of this access pattern on this box, same run, same buffer, same protocol: an in-place copy with
    # the kernel's streaming geometry and no arithmetic (whvi_stream_copy_probe, whvi_amd/csrc/stream_probe.hip)
    ceiling = None
    if device.type == "cuda":
        try:
            _, copy_ms, _ = timed(lambda: _hip.stream_copy_probe(x, out=x), args.steps, args.warmup, device, world)
            ceiling_gbs = alg_bytes / (copy_ms * 1e-3) / 1e9
            ceiling = {"GB_per_s": round(ceiling_gbs, 1), "frac_of_peak": round(ceiling_gbs / HBM_PEAK_GBS, 4),
                       "avg_launch_ms_hip_events": round(copy_ms, 4), "kernel": _hip.last_kernel(),
                       "what": "in-place copy of the same buffer with the transform's launch geometry (256-thread blocks, one "
                               "16 KiB tile per wave, XCD-contiguous order, nt loads, store barrier, sc1+nt stores), no "
                               "arithmetic; same steps / warmup, HIP events on the launch stream"}
        except Exception as err:                      # noqa: BLE001  (context for the fraction, never a reason to lose the line)
            ceiling = {"error": repr(err)}

    rec = {
        "metric": "batched FWHT Gtransforms/sec (achieved HBM GB/s vs roofline in `roofline`)",
        "value": round(value, 5), "unit": "Gtransforms/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(wall * 1e3 / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"in-place batched FWHT, D={d} fp32, {rows} rows per GPU "
                               f"(batch 8192 x 128 MC samples), {rows * d * 4 / 2**30:.0f} GiB resident in HBM per GPU",
                   "D": d, "rows_per_gpu": rows, "parallelism": f"row-sharded x{world}, no data-path collective",
                   "values_finite_after_run": finite},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "ceiling_measured": ceiling,
                     "frac_of_measured_ceiling": (round(achieved / ceiling["GB_per_s"], 4)
                                                  if ceiling and ceiling.get("GB_per_s") else None),
                     "traffic": traffic, "traffic_note": traffic_note, "kernel": kernel_symbol,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms_hip_events": round(ev_ms, 4)},
    }
    if rank == 0 and world == 1 and not CPU_PLUMBING:
        del x
        torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            try:
                rec["cpu_baseline"] = cpu_baseline(args.log2d)
            except Exception as err:   # the baseline is a report, never a reason to lose the GPU number
                rec["cpu_baseline"] = {"value": None, "error": repr(err)}
        if not args.no_extras:
            try:
                rec["extras"] = extras(device)
            except Exception as err:
                rec["extras"] = {"error": repr(err)}
    if world > 1 and not args.no_extras:
        del x
        if device.type == "cuda":
            torch.cuda.empty_cache()
        multi = multi_gpu_extras(device, rank, world)
        if rank == 0:
            rec["extras_multi_gpu"] = multi
    if world > 1:
        # the line proves what ran: every rank's device, every rank's time, and a checksum of the gathered predictions
        seen = [None] * world
        dist.all_gather_object(seen, device_identity(device))
        try:
            check = sharded_prediction_check(device, rank, world)
        except Exception as err:                      # noqa: BLE001  (a report, never a reason to lose the line)
            check = {"error": repr(err)}
        if rank == 0:
            rec["ranks_seen"] = seen
            rec["distinct_devices"] = len(set(seen))
            rec["per_rank"] = per_rank
            rec["sharded_prediction_check"] = check
    if world > 1 and rank == 0:
        # what the collectives actually ran on: the backend torch.distributed reports ("nccl" is RCCL on ROCm)
        rec["distributed"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                              "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version())
                              if device.type == "cuda" else None,
                              "launcher": "torch.distributed.run (env://), one process per GPU"}
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
