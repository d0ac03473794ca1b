#!/usr/bin/env python3
"""Headline benchmark: Jacobian columns/s of the coloured sparse-Jacobian path on MI355X.

Workload (BASELINE.json `metric`, config "N=10^7 tridiagonal"): forward-difference Jacobian of the
second-difference f! (test/coloring_tests.jl:5-13) at N = 10^7 states, SparseMatrixCSC pattern
(3N-2 stored values), colorvec[i] = mod1(i,3), x ~ U(0,1) (numpy PCG64 seed 4).  A "step" is one
complete `finite_difference_jacobian!`: step-size reduction, perturbation, 1 + 3 f! evaluations,
fused difference + decompression into nzval; plan (pattern, colours) reused, x / nzval resident in
HBM.  With --gpus P the SAME problem is split into P contiguous column ranges (strong scaling),
one process per GPU, and the step ends with the RCCL all-gather that assembles nzval.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (schema in the task contract) with `roofline` (fused diff+decompress
kernel, algorithmic bytes / HIP-event time on the launch stream) and `cpu_baseline` (the CPU
restatement of the reference path, 1 core, same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
BYTES_PER_COL_DS = 89.0     # fused diff+scatter, tridiagonal CSC, f64/int32, C = 3
BYTES_PER_COL_MIN = 71.0    # what this implementation must move at minimum (fx read once, 1-B colours per entry)
BYTES_PER_COL_CALL = 210.0  # whole forward Jacobian with an opaque streaming f!


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", dest="n", type=int, default=0, help="override the number of states (columns)")
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c4",
                    help="BASELINE.json configs: c4 = headline (N=10^7 tridiagonal forward; --gpus N shards it), "
                         "c2 = N=10^6 tridiagonal forward, c3 = N=10^7 5-point Laplacian central, "
                         "c5 = 10^4 dense 32x32 blocks block-banded complex step (c3/c5: 1 GPU, parity/side lines)")
    ap.add_argument("--f-mode", choices=["lazy", "materialized"], default="lazy",
                    help="lazy: f! perturbs while loading (fd_f_launch_lazy); materialized: perturbed points written to HBM")
    ap.add_argument("--no-gather", action="store_true", help="leave nzval sharded (compute-only scaling)")
    ap.add_argument("--shard", choices=["columns", "colors"], default="columns",
                    help="N>1 decomposition: contiguous column ranges + all-gather (default; needs a row-window-capable f!), "
                         "or colour ownership + all-reduce (any f!, at most C ranks; c4/c2 only)")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64",
                    help="element type of x / f! / J: f64 is the reference's default and the headline; f32 runs the fd32_* "
                         "instantiation (tridiagonal configs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=0, help="columns for the CPU baseline sample (0 = same as --n)")
    ap.add_argument("--cpu-reps", type=int, default=64, help="upper bound; the CPU sample stops after --cpu-seconds")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample: about this much CPU work")
    return ap.parse_args()


def cpu_baseline(n, reps, seconds=12.0):
    """The oracle (pass-for-pass restatement of src/jacobians.jl:504-653 + ext/SparseArrays:38-47,
    Int64 indices, one thread) timed on this host on the same workload."""
    from oracle import oracle
    x = np.random.default_rng(4).random(n)
    colors = ((np.arange(n, dtype=np.int64) % 3) + 1)
    colptr, rowval = oracle.tridiag_csc(n)
    fx = oracle.Fixture("tridiag", n)
    best = float("inf")
    times = []
    t_all = time.perf_counter()
    for _ in range(max(reps, 1)):
        t0 = time.perf_counter()
        oracle.jacobian("forward", fx, x, colors, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
        times.append(time.perf_counter() - t0)
        best = min(best, times[-1])
        if time.perf_counter() - t_all > seconds:
            break
    # oracle.jacobian allocates its cache arrays per call (like the reference's cache-less wrapper):
    # that is inside the timed region, as it is for FiniteDiff.finite_difference_jacobian!(J,f,x;colorvec).
    return {"value": n / best, "unit": "Jacobian columns/s", "cores": 1, "kind": "port",
            "sample": "N=%d tridiagonal forward, %d full Jacobians in %.1f s of CPU work, best one reported, gcc -O3 single thread"
                      % (n, len(times), sum(times)),
            "seconds_per_jacobian": best, "median_seconds_per_jacobian": float(np.median(times))}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import finitediff_jl_amd as fd
    from finitediff_jl_amd import patterns as P
    from finitediff_jl_amd import sharded as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    # FDJAC_BENCH_BACKEND=gloo is a functional dry run of the N>1 path on fewer GPUs than ranks (ranks
    # share devices, the gather is staged through host memory); the measured configuration is "nccl" (RCCL).
    backend = os.environ.get("FDJAC_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    local_rank = dev_index
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # one dedicated (non-default) stream for everything: the library enqueues on torch's current stream, so torch ops,
    # the RCCL collective and torch events are all ordered with the library's kernels
    bench_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(bench_stream)
    cfg = args.config
    np_dt = np.float32 if args.dtype == "f32" else np.float64
    t_dt = torch.float32 if args.dtype == "f32" else torch.float64
    if args.dtype == "f32" and cfg not in ("c2", "c4"):
        raise SystemExit("--dtype f32 is wired for the tridiagonal configs (c2, c4)")
    if cfg in ("c3", "c5") and world > 1:
        raise SystemExit("--config %s is a single-GPU line" % cfg)
    ctx = fd.Context(local_rank)
    lazy_ok = False
    if cfg in ("c2", "c4"):
        N = args.n or (10 ** 6 if cfg == "c2" else 10 ** 7)
        seed, fdtype, C = (2 if cfg == "c2" else 4), "forward", 3
        x_host = np.random.default_rng(seed).random(N)
        colors = P.cyclic_colors(N, 3)
        colptr, rowval = P.tridiag_csc(N)
        nnz = rowval.size
        pattern = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
        if args.shard == "colors" and world > 1:
            ccuts = S.partition_colors(colors, world)
            counts, c0, c1 = [nnz], 0, N
            plan = fd.make_plan(pattern, pattern, colors, fdtype, ctx=ctx, color_range=(ccuts[rank], ccuts[rank + 1]),
                                dtype=np_dt)
        else:
            cuts = S.partition_columns(colptr, world)
            ranges = S.entry_ranges(colptr, cuts)
            counts = [b - a for a, b in ranges]
            c0, c1 = int(cuts[rank]), int(cuts[rank + 1])
            xw = S.x_window(cuts, rank, N, 1, 1, 1)
            plan = fd.make_plan(pattern, pattern, colors, fdtype, ctx=ctx, col_window=(c0, c1) if world > 1 else None,
                                x_window=xw if world > 1 else None, dtype=np_dt)
        f = fd.BuiltinF("tridiag", N, ctx=ctx, dtype=np_dt)
        lazy_ok = True
        # per column: SURVEY 8(d) algorithmic bytes; what this implementation must move at minimum (fx and the three
        # f! arrays read once = 32 B, 3 values written = 24 B, index = 3 x 2-B packed (row,colour) codes with the
        # row-window kernel, 3 x (4-B row + 1-B colour) with the gather kernel); whole call with a streaming f!
        # (regular tiles of the row-window kernel compute their entry codes from a per-tile head: no index traffic)
        idx = 0.0 if plan.info(fd.lib.INFO_WIN_PERIOD) else 6.0
        bytes_ds, bytes_min, bytes_call = 89.0, (56.0 + idx if plan.info(fd.lib.INFO_WINDOW) else 71.0), 210.0
        if args.dtype == "f32":   # the same formulas with 4-byte values: 2*C*M*4 + nnz*4 + nnz*4 + (N+1)*4 + N
            bytes_ds, bytes_min, bytes_call = 53.0, (28.0 + idx if plan.info(fd.lib.INFO_WINDOW) else 43.0), 114.0
        wl = "N=%d tridiagonal CSC (nnz=3N-2), colorvec=mod1(i,3), forward, f!=second difference, x~U(0,1) seed %d" % (N, seed)
        kern = "k_decompress_window<forward>" if plan.info(fd.lib.INFO_WINDOW) else "k_decompress_list<u8,forward>"
        exact = (-2.0, 1.0)
        del rowval
        pattern.rowval = None
    elif cfg == "c3":
        nx, ny = (4000, 2500) if not args.n else (int(args.n ** 0.5), int(args.n ** 0.5))
        N = nx * ny
        fdtype, C = "central", 5
        x_host = np.random.default_rng(3).random(N)
        colors = P.lap5_colors(nx, ny)
        colptr, rowval = P.lap5_csc(nx, ny)
        nnz = rowval.size
        counts, c0, c1 = [nnz], 0, N
        pattern = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
        plan = fd.make_plan(pattern, pattern, colors, fdtype, ctx=ctx)
        f = fd.BuiltinF("lap5", nx, ny, ctx=ctx)
        lazy_ok = True
        bytes_ds = (2 * C * 8 * N + nnz * 12 + 4 * (N + 1) + N) / N      # 145 B/col
        idx_b = 2 if plan.info(fd.lib.INFO_WINDOW) else (7 if plan.info(fd.lib.INFO_SORTED_GATHER) else 5)   # index bytes per stored entry
        bytes_min = (2 * C * 8 * N + nnz * (8 + idx_b)) / N
        bytes_call = (2 * C * 16 * N + 9 * N + 2 * C * 8 * N + 9 * N) / N + bytes_ds
        wl = "N=%d (%dx%d) 5-point Laplacian CSC (nnz=%d), colours (i+2j)%%5+1, central, x~U(0,1) seed 3" % (N, nx, ny, nnz)
        kern = ("k_decompress_window2d<central>" if plan.info(fd.lib.INFO_WINDOW2D) else
                "k_decompress_window<central>" if plan.info(fd.lib.INFO_WINDOW) else
                "k_decompress_sorted<u8,central>" if plan.info(fd.lib.INFO_SORTED_GATHER) else "k_decompress_list<u8,central>")
        exact = (-4.0, 1.0)
        del rowval
        pattern.rowval = None
    else:  # c5
        nb, bs = (args.n // 32 if args.n else 10 ** 4), 32
        N = nb * bs
        fdtype = "complex"
        x_host = np.random.default_rng(5).random(N)
        lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
        colors = lay.colors()
        C = int(colors.max())
        nnz = lay.data_len
        counts, c0, c1 = [nnz], 0, N
        Jbb = fd.BlockBandedMatrix(None, lay)
        plan = fd.make_plan(Jbb, Jbb, colors, fdtype, ctx=ctx)
        f = fd.BuiltinF("blockcoupled", nb, bs, ctx=ctx)
        lazy_ok = True
        bytes_ds = (C * N * 16 + nnz * 8) / N
        bytes_min = (nnz * 16 + nnz * 8 + N) / N                          # every stored value reads one complex f value
        bytes_call = bytes_ds + (C * N * 16 * 3 + 9 * N) / N
        wl = "%d dense %dx%d blocks, block-tridiagonal BlockBandedMatrix (N=%d, %d stored values), %d colours, complex step, x~U(0,1) seed 5" % (nb, bs, bs, N, nnz, C)
        kern = ("k_decompress_colrange_wg<u8,complex>" if os.environ.get("FDJAC_COLRANGE_WG", "1") != "0"
                else "k_decompress_colrange<u8,complex>")
        exact = None
    x = torch.as_tensor(x_host.astype(np_dt), device=dev)
    if args.f_mode == "lazy" and lazy_ok:
        plan.set_lazy(f)
    f_mode = "lazy" if (args.f_mode == "lazy" and lazy_ok) else "materialized"
    gather = world > 1 and not args.no_gather
    by_color = args.shard == "colors" and world > 1 and cfg in ("c2", "c4")
    bufs = S.AllGatherBuffers(counts, dev, t_dt)
    out = bufs.local_view(rank)[: counts[rank]] if (world > 1 and not by_color) else bufs.buf
    own = torch.zeros_like(out) if by_color else None   # colour ownership: every rank holds the whole nzval, zero
                                                        # except for the columns of its colours; assembly = SUM

    host_bufs = S.AllGatherBuffers(counts, torch.device("cpu"), t_dt) if (gather and backend != "nccl") else None

    def do_gather():
        if by_color:
            bufs.buf.copy_(own)
            S.all_reduce_owned(bufs.buf, dist) if backend == "nccl" else bufs.buf.copy_(S.all_reduce_owned(bufs.buf.cpu(), dist))
            return
        if backend == "nccl":
            bufs.gather(rank, dist)          # one ncclAllGather, in place in the padded buffer
        else:                                # dry run: stage through host memory
            host_bufs.local_view(rank).copy_(bufs.local_view(rank))
            host_bufs.gather(rank, dist)
            bufs.buf.copy_(host_bufs.buf)

    if by_color:
        out = own

    enqueue = plan.bind(f, x, [out])   # pointers resolved once: one foreign call per Jacobian, as from compiled code

    def step():
        enqueue()
        if gather:
            do_gather()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    plan.enable_timing(1)   # HIP events around the graded kernel, on the launch stream (2 per step; nothing waits on them)
    # gather time measured on its own with HIP events on torch's current stream (= the plan's stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * args.steps)] if gather else []
    t0 = time.perf_counter()
    for k in range(args.steps):
        enqueue()
        if gather:
            ev[2 * k].record()
            do_gather()
            ev[2 * k + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    tm = plan.timings()
    # per-stage breakdown from a separate, untimed pass (more events => more marker packets on the stream)
    plan.enable_timing(2)
    for _ in range(min(args.steps, 10)):
        enqueue()
    torch.cuda.synchronize()
    tm_all = plan.timings()
    plan.enable_timing(0)
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    ms_step = elapsed / args.steps * 1e3
    ms_gather = (sum(ev[2 * k].elapsed_time(ev[2 * k + 1]) for k in range(args.steps)) / args.steps) if gather else 0.0

    # sanity on the result of the last step (linear fixture => exact stencil), not timed
    full = (bufs.buf if by_color else bufs.compact()) if gather or world == 1 else None
    check = None
    if full is not None:
        v = full if world > 1 else out
        sample = v[:: max(1, v.numel() // 1000003)].cpu().numpy()
        if exact is not None:  # linear fixture => the stored values are exactly the stencil weights
            check = float(np.max(np.minimum(np.abs(sample - exact[0]), np.abs(sample - exact[1]))))
        else:
            check = float(np.isfinite(sample).all()) - 1.0

    if rank == 0:
        n_local = c1 - c0
        dec = tm["decompress"]
        dec_ms = dec["ms_sum"] / max(dec["launches"], 1)
        achieved = bytes_ds * n_local / (dec_ms * 1e-3) / 1e9 if dec_ms > 0 else 0.0
        tot_ms = tm_all["total"]["ms_sum"] / max(tm_all["total"]["launches"], 1)
        pmc = None
        # HBM bytes per launch of the graded kernel from the committed rocprofv3 PMC passes of this same command
        # (scripts/profile.sh + scripts/make_pmc_json.py; counters cannot be read from inside the process)
        pmc_path = os.path.join(ROOT, "profiles", "pmc_%s.json" % cfg)
        if os.path.exists(pmc_path):
            try:
                j = json.load(open(pmc_path))
                if (int(j.get("n", -1)) == N and int(j.get("gpus", 1)) == world and j.get("kernel", "") in kern
                        and args.dtype == "f64"):
                    pmc = j.get("decompress_hbm_bytes_per_launch")
            except Exception:
                pmc = None
        res = {
            "metric": "Jacobian columns/s (coloured sparse finite-difference Jacobian; headline config N=10^7 tridiagonal forward)",
            "value": N / (ms_step * 1e-3),
            "unit": "columns/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": wl, "name": cfg, "fdtype": fdtype, "colors": C,
                       "parallelism": ("colours x%d%s" % (world, "+allreduce" if gather else "")) if by_color else
                                      ("columns x%d%s" % (world, "+allgather" if gather else "")),
                       "f_mode": ("built-in device f! behind fd_f_launch_lazy (1 launch: base + lazily perturbed points)"
                                  if f_mode == "lazy" else
                                  "built-in device f! behind fd_f_launch (materialised points, one batched launch)"),
                       "gather_in_step": bool(gather), "collective_backend": backend if world > 1 else None},
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc,
                "kernel": kern + " (fused difference + decompression)",
                "avg_launch_ms": dec_ms, "launches_timed": dec["launches"],
                "algorithmic_bytes_per_launch": bytes_ds * n_local,
                "min_traffic_bytes_per_launch": bytes_min * n_local,
                "achieved_on_min_traffic": bytes_min * n_local / (dec_ms * 1e-3) / 1e9 if dec_ms > 0 else 0.0,
            },
            "stages_ms": {k: (v["ms_sum"] / max(v["launches"], 1)) for k, v in tm_all.items()},
            "whole_call": {"gpu_ms": tot_ms, "algorithmic_bytes": bytes_call * n_local,
                           "gbps": bytes_call * n_local / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0},
            "ms_gather": ms_gather,
            "value_compute_only": N / ((ms_step - ms_gather) * 1e-3) if ms_step > ms_gather else None,
            "result_check_max_dev": check,
        }
        try:
            res["stream_copy_gbps"] = ctx.stream_copy_gbps(1 << 30, 10)
        except Exception as e:  # pragma: no cover
            res["stream_copy_gbps"] = None
            res["stream_copy_error"] = str(e)
        if not args.no_cpu_baseline and world == 1 and cfg in ("c2", "c4") and args.dtype == "f64":
            try:
                res["cpu_baseline"] = cpu_baseline(args.cpu_n or N, args.cpu_reps, args.cpu_seconds)
            except Exception as e:  # pragma: no cover
                res["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
