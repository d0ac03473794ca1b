#!/usr/bin/env python3
"""Headline benchmark: Jacobian columns/s of the coloured sparse-Jacobian path on MI355X.

Workload (BASELINE.json `metric`, config "N=10^7 tridiagonal"): forward-difference Jacobian of the
second-difference f! (test/coloring_tests.jl:5-13) at N = 10^7 states, SparseMatrixCSC pattern
(3N-2 stored values), colorvec[i] = mod1(i,3), x ~ U(0,1) (numpy PCG64 seed 4).  A "step" is one
complete `finite_difference_jacobian!`: step-size reduction, 1 + 3 f! evaluations at the lazily perturbed
points, difference, division and decompression into nzval -- since round 3 the last three run as ONE launch
(f!'s launch stores the Jacobian itself: include/fdjac_device.h, k_f_tridiag_store_wave); plan (pattern,
colours) reused, x / nzval resident in HBM.  `value` comes from EXACTLY --steps calls bracketed by
barrier + synchronize; `median_ms_per_step` / `value_median` from >= 20 individually timed calls (HIP events
on the launch stream); --sweep DIR additionally writes the c2 / c3 / c5 lines into DIR.

With --gpus P the SAME problem is split into P contiguous column ranges (strong scaling), one process per GPU.
Every rank fills its contiguous slice of nzval; the timed step ends there, with nzval device-resident and sharded by
column range -- the layout the sharded tridiagonal solve consumes (`fd_tridiag_solve`).  The
assembly of nzval on one rank ("a single RCCL gather over xGMI", fd_comm_gatherv behind the C ABI) is measured right
after the timed region and reported next to it (`gather`, `value_with_gather`); `--gather-in-step` puts it inside the
timed step instead.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line on stdout (schema in the task contract) with `roofline` (fused diff+decompress kernel: HBM
bytes actually moved / HIP-event time on the launch stream) and `cpu_baseline` (the CPU restatement of the reference
path on this host, 1 core -- plus an all-cores OpenMP variant as an upper bound).  Every rank prints one diagnostic
JSON line on stderr (device, RCCL library / version, per-stage times).
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
JSON_FD = None              # N>1: the descriptor the ONE result line goes to (stdout proper; fd 1 is pointed at stderr)


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
    ap.add_argument("--gather-in-step", action="store_true",
                    help="N>1: assemble nzval on rank 0 (fd_comm_gatherv) inside every timed step")
    ap.add_argument("--gather", choices=["root", "all"], default="root",
                    help="N>1 assembly: root = grouped point-to-point gather to rank 0 (default), all = in-place all-gather")
    ap.add_argument("--eps", choices=["replicated", "sharded"], default="replicated",
                    help="N>1 step-size reduction: every rank reduces all of x (no collective on the critical path), or "
                         "each rank reduces its blocks and the partial sums are all-gathered (fd_plan_set_comm); same bits")
    ap.add_argument("--x-layout", choices=["replicated", "sharded"], default="replicated",
                    help="N>1: replicated = every rank holds all of x (the Jacobian of a given x: no exchange in the step); sharded = "
                         "the time-stepping layout -- rank r holds its own part of x, every step starts with the neighbour halo "
                         "exchange (fd_comm_halo_exchange) and the step-size reduction is sharded over contiguous ranges "
                         "(FD_PLAN_EPS_CONTIGUOUS + fd_plan_set_comm): per step 2(l+u) values per link + one all-gather of the partial sums")
    ap.add_argument("--small-messages", choices=["rccl", "p2p"], default="rccl",
                    help="N>1: how the per-step small messages travel (halo of x, partial sums of the sharded step-size reduction, solve "
                         "interface): RCCL collectives, or direct peer-to-peer stores into the ranks' mailboxes (fd_comm_enable_p2p / "
                         "fd_p2p_*: hipIpc-mapped HBM, one kernel per exchange, no proxy); the bulk nzval gather is RCCL either way")
    ap.add_argument("--weak", action="store_true",
                    help="N>1: weak scaling -- N = gpus x 10^7 columns (each rank keeps the single-GPU problem size) instead of "
                         "splitting the fixed N = 10^7 problem; the line says scaling = weak")
    ap.add_argument("--shard", choices=["columns", "colors"], default="columns",
                    help="N>1 decomposition: contiguous column ranges (default; needs a row-window-capable f!), "
                         "or colour ownership + all-reduce (any f!, at most C ranks; c4/c2 only)")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64",
                    help="element type of x / f! / J: f64 is the reference's default and the headline; f32 runs the fd32_* "
                         "instantiation (tridiagonal configs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=0, help="columns for the CPU baseline sample (0 = same as --n)")
    ap.add_argument("--cpu-reps", type=int, default=64, help="upper bound; the CPU sample stops after --cpu-seconds")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline sample: about this much CPU work (1 core)")
    ap.add_argument("--soak-seconds", type=float, default=5.5,
                    help="untimed steady-state loop AFTER the measurement so that an external utilisation sampler sees "
                         "the GPU busy (the timed region itself lasts a few milliseconds); 0 = off")
    ap.add_argument("--no-plain-handover", "--no-side-runs", dest="no_plain_handover", action="store_true",
                    help="skip the untimed side measurement of the hand-over path (profiling runs: keeps the kernels' PMC averages clean)")
    ap.add_argument("--sides", action="store_true",
                    help="also run the untimed side measurements (scripts/bench_sides.py: buffer placements, hand-over path, opaque f!, runtime-"
                         "compiled functor, drop-in call); --sweep implies it.  The default run carries rotating_x and rank_share only")
    ap.add_argument("--variant", choices=["auto", "flags"], default="auto",
                    help="N>1 (c4 / c2): auto = time all four combinations of scaling (strong N = 10^7 / weak N = gpus x 10^7) and layout "
                         "(x and step sizes replicated: no exchange in the step / x sharded: ONE exchange of halo + group sums per step, "
                         "through the peer-to-peer mailboxes with automatic fall-back to RCCL) in this one run, report the fastest as "
                         "`value` (named in config.variant) and all of them in `variants`; flags = exactly what --weak / --x-layout / "
                         "--eps / --small-messages say (giving any of those flags implies it)")
    ap.add_argument("--spawn-timeout", type=float, default=1500.0, help="--gpus N without a launcher: give up on the spawned ranks after this many seconds")
    ap.add_argument("--sweep", default="", metavar="DIR",
                    help="after the headline line, run the other BASELINE configs (c2, c3, c5; plus c4 in Float32) as child "
                         "processes and write their JSON lines to DIR/bench_<config>.json (single GPU)")
    args = ap.parse_args()
    if args.sweep:
        args.sides = True
    explicit = any(a.split("=")[0] in ("--weak", "--x-layout", "--eps", "--small-messages", "--shard", "--gather-in-step") for a in sys.argv[1:])
    if explicit or args.config not in ("c2", "c4"):
        args.variant = "flags"
    return args


def emit_failure(args, what, json_fd=None):
    """The contract's ONE JSON line even when nothing could be measured: value null, the reason in `error` / `comm_error`."""
    line = json.dumps({"metric": "Jacobian columns/s (coloured sparse finite-difference Jacobian; headline config N=10^7 tridiagonal forward)",
                       "value": None, "unit": "columns/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                       "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": args.dtype,
                       "data": "synthetic", "config": {"workload": "not measured", "name": args.config}, "error": what, "comm_error": what}) + "\n"
    if json_fd is not None:
        os.write(json_fd, line.encode())
    else:
        sys.stdout.write(line)
        sys.stdout.flush()


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run, one process per GPU,
    rendezvous on 127.0.0.1 at a free port) and pass rank 0's ONE JSON line through.  Whatever happens to the ranks -- a crash, a hang
    in a collective -- this process still prints one line (value null, the reason in `error`) instead of running into the caller's timeout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    try:
        cp = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=args.spawn_timeout)
        lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
        if lines:
            sys.stdout.write(lines[-1] + "\n")
            sys.stdout.flush()
            return cp.returncode
        emit_failure(args, "the %d spawned ranks ended (rc %d) without a result line" % (args.gpus, cp.returncode))
        return cp.returncode or 1
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        if lines:
            sys.stdout.write(lines[-1] + "\n")
            sys.stdout.flush()
            return 0
        emit_failure(args, "the %d spawned ranks did not finish within %.0f s (a rank hung in a collective?)" % (args.gpus, args.spawn_timeout))
        return 1


def tridiag_device_pattern(torch, N, dev):
    """The tridiagonal SparseMatrixCSC pattern and colorvec = mod1(i, 3) built ON the device (1-based Int32, as a device sparse
    matrix carries them): colptr[j] = 3j (j >= 1), rowval[p] = (p + 1) div 3 + (p + 1) mod 3.  For the weak-scaling problem sizes
    (N = gpus x 10^7) nothing of size nnz is built on the host or crosses PCIe."""
    colptr = torch.arange(N + 1, device=dev, dtype=torch.int64) * 3
    colptr[0] = 1
    colptr[N] = 3 * N - 1
    q = torch.arange(1, 3 * N - 1, device=dev, dtype=torch.int64)            # p + 1 for p = 0 .. 3N - 3
    rowval = (q // 3 + q % 3).to(torch.int32)
    del q
    colors = (torch.arange(N, device=dev, dtype=torch.int32) % 3 + 1).to(torch.int32)
    return colptr.to(torch.int32), rowval, colors


def tridiag_entry_begin(c, N):
    """0-based index of column c's first stored value in the tridiagonal CSC nzval (c = N: nnz)."""
    return 0 if c <= 0 else (3 * N - 2 if c >= N else 3 * c - 1)


def cpu_baseline(n, reps, seconds=10.0):
    """The oracle (pass-for-pass restatement of src/jacobians.jl:504-653 + ext/SparseArrays:38-47, Int64 indices) timed
    on this host on the same workload with a reused cache (arrays allocated once, outside the timed call, like a
    JacobianCache): one core -- what the single-threaded reference does -- and, labelled as an upper bound that is NOT
    the reference, the same passes split over all host cores with OpenMP."""
    from oracle import oracle
    x = np.random.default_rng(4).random(n)
    colors = ((np.arange(n, dtype=np.int64) % 3) + 1)
    colptr, rowval = oracle.tridiag_csc(n)

    def sample(omp, budget, max_reps):
        fx = oracle.Fixture("tridiag", n, omp=omp)
        call, _out = oracle.jacobian("forward", fx, x, colors, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval,
                                     omp=omp, runner=True)
        call()                                   # first touch of the cache arrays
        times, t_all = [], time.perf_counter()
        for _ in range(max(max_reps, 1)):
            t0 = time.perf_counter()
            call()
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all > budget:
                break
        return times

    t1 = sample(False, seconds, reps)
    med = float(np.median(t1))
    host_cores = os.cpu_count() or 1
    res = {"value": n / med, "unit": "Jacobian columns/s", "cores": 1, "host_cores": host_cores, "kind": "port",
           "sample": "N=%d tridiagonal forward, %d full Jacobians in %.1f s of CPU work on one core, MEDIAN reported; "
                     "gcc -O3 -march=x86-64-v3, cache arrays allocated once outside the timed call"
                     % (n, len(t1), sum(t1)),
           "seconds_per_jacobian": med, "best_seconds_per_jacobian": float(min(t1))}
    try:
        res["omp"] = []
        for want in (8, 32):     # an 8-core variant; 32 threads to show where the memory system saturates
            nt = oracle.set_omp_threads(min(want, host_cores))
            tm = sample(True, max(seconds / 5.0, 1.5), reps)
            res["omp"].append({"value": n / float(np.median(tm)), "threads": nt, "seconds_per_jacobian": float(np.median(tm)),
                               "jacobians": len(tm)})
            if nt < want:
                break
        res["omp_note"] = ("upper bound, NOT the reference: FiniteDiff.jl is single-threaded; these are the same passes "
                           "with their loops split over host cores with OpenMP")
    except Exception as e:  # pragma: no cover
        res["omp"] = {"error": str(e)}
    return res


def live_pmc(cfg, dtype, kernel_like, timeout_s=150.0):
    """HBM bytes per launch of the graded kernel, measured NOW: two bounded rocprofv3 passes (--pmc FETCH_SIZE / WRITE_SIZE with
    --kernel-trace only) of a short run of this same command, corrected as MI355X_MICROARCH.md prescribes -- both counters calibrated on
    the 1 GiB stream copy the run itself contains (FETCH_SIZE comes out x2 on gfx950).  None when rocprofv3 is not on the box, when this
    process is itself being profiled, or when a pass fails / times out: the caller then falls back to the committed PMC file."""
    import shutil, sqlite3, subprocess, tempfile
    if os.environ.get("FDJAC_BENCH_LIVE_PMC", "1") == "0" or os.environ.get("FDJAC_BENCH_PMC_CHILD"):
        return None, "live PMC switched off"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTX")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is being profiled"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "no rocprofv3 on this box"
    out = {}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(prefix="fdjac_pmc_", dir="/tmp") as tmp:
        env = dict(os.environ, FDJAC_BENCH_PMC_CHILD="1", TMPDIR="/tmp")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--dtype", dtype, "--steps", "5", "--warmup", "2",
               "--no-cpu-baseline", "--soak-seconds", "0", "--no-plain-handover"]
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            try:
                subprocess.run(["timeout", "-k", "5", str(int(timeout_s)), exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "run", "--"] + cmd,
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s + 15, check=False)
            except Exception as e:
                return None, "rocprofv3 pass failed: %s" % type(e).__name__
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            if not dbs:
                return None, "rocprofv3 pass wrote no database"
            try:
                cur = sqlite3.connect(dbs[0]).cursor()
                def mean(like):
                    cur.execute("select avg(value), count(*) from counters_collection where kernel_name like ? and counter_name = ?", ("%" + like + "%", counter))
                    v, n = cur.fetchone()
                    return (v or 0.0), (n or 0)
                cal, ncal = mean("k_stream_copy")
                val, nval = mean(kernel_like)
            except Exception as e:
                return None, "counter database unreadable: %s" % type(e).__name__
            if not (ncal > 0 and nval > 0 and cal > 0):
                return None, "kernel or calibration copy not found in the %s pass" % counter
            out[counter] = {"raw_kb": val, "factor": float(1 << 30) / (cal * 1024.0), "dispatches": nval}
    rd = out["FETCH_SIZE"]["raw_kb"] * 1024.0 * out["FETCH_SIZE"]["factor"]
    wr = out["WRITE_SIZE"]["raw_kb"] * 1024.0 * out["WRITE_SIZE"]["factor"]
    return {"bytes": rd + wr, "read_bytes": rd, "write_bytes": wr, "fetch_factor": out["FETCH_SIZE"]["factor"], "write_factor": out["WRITE_SIZE"]["factor"],
            "dispatches": [out["FETCH_SIZE"]["dispatches"], out["WRITE_SIZE"]["dispatches"]], "seconds": time.perf_counter() - t0}, None


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_spawn(args))
    import torch
    import torch.distributed as dist
    import finitediff_jl_amd as fd
    from finitediff_jl_amd import patterns as P
    from finitediff_jl_amd import sharded as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly ONE line, the result JSON: RCCL prints a version banner on stdout when a communicator is
    # created, so with several ranks everything else this process (and the libraries under it) writes to fd 1 goes to stderr
    # and the JSON line is written to the saved descriptor
    global JSON_FD
    json_fd = None
    if world > 1:
        sys.stdout.flush()
        json_fd = JSON_FD = os.dup(1)
        os.dup2(2, 1)
    if not torch.cuda.is_available():
        if rank == 0:
            emit_failure(args, "no GPU visible: bench.py needs an MI355X; there is no CPU fallback for the product path", json_fd)
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    # FDJAC_BENCH_BACKEND=gloo is a functional dry run of the N>1 path on fewer GPUs than ranks (ranks share devices,
    # the assembly is staged through host memory by torch.distributed); the measured configuration is "nccl": RCCL,
    # called by libfdjac itself (fd_comm_*), torch.distributed only for the barrier / max-over-ranks / id broadcast.
    backend = os.environ.get("FDJAC_BENCH_BACKEND", "nccl")
    shared_devices = None
    if world > 1:
        os.environ.setdefault("FDJAC_P2P_TIMEOUT_MS", "500")      # (a mailbox wait that does not complete gives up after this long, loudly)
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        # fewer GPUs than ranks: RCCL refuses two ranks on one device (and may hang finding out) -- run the dry-run transport and say so
        backend = "gloo"
        shared_devices = "%d ranks on %d GPU(s): ranks share devices, RCCL is impossible -- gloo dry run, NOT a scaling measurement" % (world, torch.cuda.device_count())
        sys.stderr.write("[bench rank %d] %s\n" % (rank, shared_devices))
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=300))

    # one dedicated (non-default) stream for everything: the library enqueues on torch's current stream, so torch ops,
    # the RCCL collectives and torch events are all ordered with the library's kernels
    bench_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(bench_stream)
    cfg = args.config
    np_dt = np.float32 if args.dtype == "f32" else np.float64
    t_dt = torch.float32 if args.dtype == "f32" else torch.float64
    if cfg in ("c3", "c5") and world > 1:
        raise SystemExit("--config %s is a single-GPU line" % cfg)
    ctx = fd.Context(dev_index)
    comm = None
    comm_error = shared_devices
    if world > 1 and backend == "nccl":
        try:
            comm = fd.Comm.from_torch_distributed(ctx, dist)
        except Exception as e:   # the timed step has no collective in it: measure it anyway, report the failure loudly
            comm_error = "%s: %s" % (type(e).__name__, e)
            sys.stderr.write("[bench rank %d] fd_comm_create failed (%s): the nzval assembly and the sharded solve are skipped\n" % (rank, comm_error))
    p2p = None
    p2p_note = None
    if world > 1 and (args.small_messages == "p2p" or args.variant == "auto"):
        try:
            if comm is not None:
                comm.enable_p2p(1 << 17)       # the communicator's small messages go through the mailboxes from here on
                p2p_note = "fd_comm_enable_p2p: mailboxes mapped over hipIpc, handles exchanged over RCCL"
            else:
                p2p = fd.P2P.from_torch_distributed(ctx, dist, slot_bytes=1 << 17)
                p2p_note = "fd_p2p_* (handles exchanged through torch.distributed; %s mailbox)" % ("uncached" if p2p.info()["uncached"] else "device")
        except Exception as e:
            p2p_note = "FAILED, small messages stay on %s: %s: %s" % ("RCCL" if comm is not None else "the host", type(e).__name__, e)
            sys.stderr.write("[bench rank %d] peer-to-peer mailboxes unavailable (%s)\n" % (rank, p2p_note))
    by_color = args.shard == "colors" and world > 1 and cfg in ("c2", "c4")
    x_sharded = False
    lazy_ok = False
    vs = 4 if args.dtype == "f32" else 8          # bytes per value

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def p2p_status():
        """1 + r once a mailbox wait for rank r timed out on ANY rank (sticky), else 0."""
        mine = 0
        try:
            mine = comm.p2p_status() if comm is not None else (p2p.status() if p2p is not None else 0)
        except Exception:
            mine = 0
        t = torch.tensor([float(mine)], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())

    def build_tridiag(N, seed, sharded, on_device):
        """This rank's share of the N-column tridiagonal forward problem: the plan of its column range, x, its output slice.
        sharded = the time-stepping layout: the rank holds ONLY x[c0 - 2, c1 + 2) (everything else is NaN here, to prove it),
        columns cut where the step-size reduction's groups are cut; every call exchanges halo + group sums itself (ONE launch through
        the mailboxes, else RCCL / the host).  on_device = the pattern is built on the device (weak-scaling sizes)."""
        pb = {"N": N, "sharded": bool(sharded and world > 1)}
        x_host = np.random.default_rng(seed).random(N)
        cuts = S.eps_shard_cuts(N, world) if world > 1 else np.array([0, N], dtype=np.int64)
        if world > 1 and not sharded and not on_device and args.variant == "flags":
            cuts = None          # (flags mode, replicated layout: the cuts balanced by stored values, as in rounds 1-4)
        t0 = time.perf_counter()
        if on_device:
            cp_d, rv_d, cv_d = tridiag_device_pattern(torch, N, dev)
            pattern = fd.DevicePatternCSC(N, N, cp_d, rv_d, None)
            colors = cv_d
            nnz = 3 * N - 2
        else:
            colors = P.cyclic_colors(N, 3)
            colptr, rowval = P.tridiag_csc(N)
            nnz = rowval.size
            pattern = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
            if cuts is None:
                cuts = S.partition_columns(colptr, world)
        counts = [tridiag_entry_begin(int(b), N) - tridiag_entry_begin(int(a), N) for a, b in zip(cuts[:-1], cuts[1:])]
        c0, c1 = int(cuts[rank]), int(cuts[rank + 1])
        xw = S.x_window(cuts, rank, N, 1, 1, 1)
        t_plan = time.perf_counter()
        plan = fd.make_plan(pattern, pattern, colors, "forward", ctx=ctx, col_window=(c0, c1) if world > 1 else None,
                            x_window=xw if world > 1 else None, dtype=np_dt)
        pb["plan_build_ms"] = (time.perf_counter() - t_plan) * 1e3
        pb["setup_ms"] = (time.perf_counter() - t0) * 1e3
        f = fd.BuiltinF("tridiag", N, ctx=ctx, dtype=np_dt)
        x = torch.as_tensor(x_host.astype(np_dt), device=dev)
        if pb["sharded"]:
            lo, hi = max(c0 - 2, 0), min(c1 + 2, N)
            x_full = x
            x = torch.full_like(x_full, float("nan"))
            x[c0:c1] = x_full[c0:c1]                 # the halo cells arrive with the first call
            pb["x_full_window"] = (lo, hi, x_full[lo:hi].clone())
            del x_full
        if not on_device:
            pattern.rowval = None
        pb.update(plan=plan, f=f, x=x, cuts=cuts, counts=counts, c0=c0, c1=c1, nnz=nnz, colors=colors, pattern=pattern)
        return pb

    def attach_exchange(pb):
        """Sharded layout: hand the per-step exchange to the library (returns a description), or None if it has to be staged through the host."""
        plan = pb["plan"]
        if comm is not None:
            plan.set_comm(comm)
            plan.set_halo(pb["c0"], pb["c1"], 2)
            return "inside the call: " + ("ONE launch through the peer-to-peer mailboxes (halo + group sums + step sizes)" if comm.has_p2p() else
                                          "RCCL (grouped send / recv of the halo, all-gather of the group sums)")
        if p2p is not None:
            plan.set_p2p(p2p)
            plan.set_halo(pb["c0"], pb["c1"], 2)
            return "inside the call: ONE launch through the peer-to-peer mailboxes (halo + group sums + step sizes)"
        return None

    variants = None
    if world > 1 and args.variant == "auto" and cfg in ("c2", "c4"):
        # ---- every combination of scaling and layout timed in this one run; the fastest becomes the reported step ----
        base_n = args.n or (10 ** 6 if cfg == "c2" else 10 ** 7)
        variants = []
        for weak in (False, True):
            for sharded in (False, True):
                name = "%s-%s" % ("weak" if weak else "strong", "sharded" if sharded else "replicated")
                v = {"name": name, "scaling": "weak" if weak else "strong", "N": base_n * (world if weak else 1),
                     "layout": ("x sharded: every rank holds its own part + halo; ONE exchange per step" if sharded else
                                "x replicated: every rank reduces all of x, no exchange in the step")}
                try:
                    pb = build_tridiag(v["N"], 2 if cfg == "c2" else 4, sharded, weak)
                    v["plan_build_ms"], v["setup_ms"] = pb["plan_build_ms"], pb["setup_ms"]
                    if sharded:
                        v["exchange"] = attach_exchange(pb)
                        if v["exchange"] is None:
                            raise RuntimeError("no device-side transport for the per-step exchange (neither RCCL nor mailboxes)")
                    pb["plan"].set_lazy(pb["f"])
                    outv = torch.empty(pb["counts"][rank], dtype=t_dt, device=dev)
                    call = pb["plan"].bind(pb["f"], pb["x"], [outv])
                    call()                                   # first contact: one step, then ask every rank how its waits went
                    fence()
                    st = p2p_status() if sharded else 0
                    if st and comm is not None:
                        # stores into a peer's mailbox did not arrive (in time): every rank drops the mailboxes and the SAME layout is
                        # timed through RCCL instead (send / recv of the halo, all-gather of the group sums) -- not a timeout per step
                        v["p2p_error"] = "a mailbox wait for rank %d timed out on the first step: small messages back on RCCL" % (st - 1)
                        sys.stderr.write("[bench rank %d] %s\n" % (rank, v["p2p_error"]))
                        comm.disable_p2p()
                        v["exchange"] = attach_exchange(pb)
                        call()
                        fence()
                    elif st:
                        raise RuntimeError("a mailbox wait for rank %d timed out on the first step" % (st - 1))
                    for _ in range(max(args.warmup, 2)):
                        call()
                    fence()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        call()
                    fence()
                    tv = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
                    dist.all_reduce(tv, op=dist.ReduceOp.MAX)
                    v["ms_per_step"] = float(tv.item()) / args.steps * 1e3
                    v["value"] = v["N"] / (v["ms_per_step"] * 1e-3)
                    st = p2p_status()
                    if st:
                        raise RuntimeError("a mailbox wait for rank %d timed out" % (st - 1))
                    del call, outv, pb
                except Exception as e:
                    v["error"] = "%s: %s" % (type(e).__name__, e)
                    sys.stderr.write("[bench rank %d] variant %s failed: %s\n" % (rank, name, v["error"]))
                # every rank must agree on which variants count (a failure on one rank only disqualifies it everywhere)
                bad = torch.tensor([1.0 if "error" in v else 0.0], dtype=torch.float64, device=dev)
                dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                if bad.item() and "error" not in v:
                    v["error"] = "failed on another rank"
                variants.append(v)
                torch.cuda.empty_cache()
        good = [v for v in variants if "error" not in v]
        best = max(good, key=lambda v: v["value"]) if good else variants[0]
        args.weak = best["scaling"] == "weak"
        args.x_layout = "sharded" if best["name"].endswith("sharded") else "replicated"
        if rank == 0:
            sys.stderr.write("[bench] variants: %s -> %s\n" % (json.dumps(variants), best["name"]))
    t_plan = time.perf_counter()
    if cfg in ("c2", "c4"):
        N = (args.n or (10 ** 6 if cfg == "c2" else 10 ** 7)) * (world if args.weak else 1)
        seed, fdtype, C = (2 if cfg == "c2" else 4), "forward", 3
        if by_color:
            x_host = np.random.default_rng(seed).random(N)
            colors = P.cyclic_colors(N, 3)
            colptr, rowval = P.tridiag_csc(N)
            nnz = rowval.size
            pattern = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
            t_plan = time.perf_counter()
            ccuts = S.partition_colors(colors, world)
            counts, c0, c1 = [nnz], 0, N
            plan = fd.make_plan(pattern, pattern, colors, fdtype, ctx=ctx, color_range=(ccuts[rank], ccuts[rank + 1]),
                                dtype=np_dt)
            plan_build_ms = (time.perf_counter() - t_plan) * 1e3
            f = fd.BuiltinF("tridiag", N, ctx=ctx, dtype=np_dt)
            x = torch.as_tensor(x_host.astype(np_dt), device=dev)
            del rowval
            pattern.rowval = None
        else:
            x_sharded = world > 1 and args.x_layout == "sharded"
            pb = build_tridiag(N, seed, x_sharded, world > 1 and args.weak)
            plan, f, x, cuts, counts, c0, c1, nnz, colors, pattern = (pb[k] for k in ("plan", "f", "x", "cuts", "counts", "c0", "c1", "nnz", "colors", "pattern"))
            plan_build_ms = pb["plan_build_ms"]
        lazy_ok = True
        win, per = plan.info(fd.lib.INFO_WINDOW), plan.info(fd.lib.INFO_WIN_PERIOD)
        idx_b = (0.0 if per else 2.0) if win else 5.0          # index bytes per stored value this plan's kernel reads
        # per column.  SURVEY 8(d)'s algorithmic bytes of the fused diff+scatter kernel: C*2*M*s + nnz*s + nnz*4 + (N+1)*4 + N.
        bytes_ds = 2 * C * vs + 3 * vs + 3 * 4 + 4 + 1          # 89 (f64) / 53 (f32)
        # what this implementation has to move: fx and the C batched f! arrays read once, every value written once,
        # plus the index stream of the chosen kernel (periodic entry codes: none)
        bytes_min = (C + 1) * vs + 3 * vs + 3 * idx_b           # 56 (f64, periodic codes)
        # whole call: eps pass (x, + 1-B colours unless cyclic) + lazy f! (x, colours, C+1 outputs) + decompression
        cyc = plan.info(fd.lib.INFO_EPS_CYCLIC)
        bytes_call_model = (vs + (0 if cyc else 1)) + (vs + 1 + (C + 1) * vs) + bytes_min
        bytes_call_survey = 210.0 if args.dtype == "f64" else 114.0
        wl = "N=%d tridiagonal CSC (nnz=3N-2), colorvec=mod1(i,3), forward, f!=second difference, x~U(0,1) seed %d" % (N, seed)
        kern = "k_decompress_window<forward>" if win else "k_decompress_list<u8,forward>"
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
        t_plan = time.perf_counter()
        plan = fd.make_plan(pattern, pattern, colors, fdtype, ctx=ctx, dtype=np_dt)
        plan_build_ms = (time.perf_counter() - t_plan) * 1e3
        f = fd.BuiltinF("lap5", nx, ny, ctx=ctx, dtype=np_dt)
        lazy_ok = True
        bytes_ds = (2 * C * vs * N + nnz * (vs + 4) + 4 * (N + 1) + N) / N      # 145 B/col (f64)
        idx_b = 2 if plan.info(fd.lib.INFO_WINDOW) else (7 if plan.info(fd.lib.INFO_SORTED_GATHER) else 5)
        bytes_min = (2 * C * vs * N + nnz * (vs + idx_b)) / N
        bytes_call_model = (vs + 1.0) + ((vs + 1.0) + 2 * C * vs) + bytes_min
        bytes_call_survey = (2 * C * 2 * vs * N + (vs + 1) * N + 2 * C * vs * N + (vs + 1) * N) / N + bytes_ds
        wl = "N=%d (%dx%d) 5-point Laplacian CSC (nnz=%d), colours (i+2j)%%5+1, central, x~U(0,1) seed 3" % (N, nx, ny, nnz)
        kern = ("k_decompress_window2d<central>" if plan.info(fd.lib.INFO_WINDOW2D) else
                "k_decompress_window<central>" if plan.info(fd.lib.INFO_WINDOW) else
                "k_decompress_sorted<u8,central>" if plan.info(fd.lib.INFO_SORTED_GATHER) else "k_decompress_list<u8,central>")
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
        t_plan = time.perf_counter()
        plan = fd.make_plan(Jbb, Jbb, colors, fdtype, ctx=ctx, dtype=np_dt)
        plan_build_ms = (time.perf_counter() - t_plan) * 1e3
        f = fd.BuiltinF("blockcoupled", nb, bs, ctx=ctx, dtype=np_dt)
        lazy_ok = True
        bytes_ds = (C * N * 2 * vs + nnz * vs) / N
        bytes_min = (nnz * vs + nnz * vs + N) / N                          # imag-only lazy f!: one real value read per stored value
        bytes_call_model = bytes_min + ((vs + 1) * N + nnz * vs) / N
        bytes_call_survey = bytes_ds + (C * N * 2 * vs * 3 + (vs + 1) * N) / N
        wl = "%d dense %dx%d blocks, block-tridiagonal BlockBandedMatrix (N=%d, %d stored values), %d colours, complex step, x~U(0,1) seed 5" % (nb, bs, bs, N, nnz, C)
        kern = "k_decompress_colrange_wg<u8,complex>" if plan.info(fd.lib.INFO_COLRANGE_WG) else "k_decompress_colrange<u8,complex>"
    if cfg in ("c3", "c5"):
        x = torch.as_tensor(x_host.astype(np_dt), device=dev)
    if args.f_mode == "lazy" and lazy_ok:
        plan.set_lazy(f)
    f_mode = "lazy" if (args.f_mode == "lazy" and lazy_ok) else "materialized"
    # FD_LAZY_CAP_DIFF: the lazy launcher hands over f(x+d) - f(x) / f(x+d) - f(x-d) (the subtraction of
    # src/jacobians.jl:565,607 moves into f!'s launch): C arrays instead of C+1 / 2C, no f(x) pass -- the byte models follow
    lazy_diff = int(plan.info(fd.lib.INFO_LAZY_DIFF)) if f_mode == "lazy" else 0
    # FD_LAZY_CAP_STORE (the default for a verified exact band since round 3): f!'s launch also divides and stores -- the graded
    # "diff + scatter" kernel is that launch; it reads x once and writes every stored value once
    lazy_store = int(plan.info(fd.lib.INFO_LAZY_STORE)) if f_mode == "lazy" else 0
    if lazy_store and cfg in ("c2", "c4"):
        bytes_min = vs + 3 * vs                                 # 32 (f64): x in, nzval out
        bytes_call_model = (vs + (0 if cyc else 1)) + bytes_min
        kern = "k_f_tridiag_store_wave<0, false>" if vs == 8 else "k_f_tridiag_store_wave4<0, false>"      # (Float32: four columns per lane)
    elif lazy_diff and cfg in ("c2", "c4"):
        bytes_min = C * vs + 3 * vs + 3 * idx_b                 # 48 (f64, periodic codes)
        bytes_call_model = (vs + (0 if cyc else 1)) + (vs + 1 + C * vs) + bytes_min
    elif lazy_store and cfg == "c3":
        bytes_min = (vs * N + N + nnz * vs) / N                 # x in, one colour byte per column, nzval out
        bytes_call_model = (vs + 1.0) + bytes_min
        kern = "k_f_stencil5_store_wave<unsigned char, 1, 0>" if vs == 8 else "k_f_stencil5_store_wave4<unsigned char, 1, 0>"   # (Float32: four columns per lane)
    elif lazy_store and cfg == "c5":
        bytes_min = (vs * N + N + 16 * N + nnz * vs) / N        # x in, one colour byte + the (row range, destination) of a column, data out
        bytes_call_model = (vs + 1.0) + bytes_min
        kern = "k_f_blockcoupled_store<unsigned char>"
    elif lazy_diff and cfg == "c3":
        bytes_min = (C * vs * N + nnz * (vs + idx_b)) / N
        bytes_call_model = (vs + 1.0) + ((vs + 1.0) + C * vs) + bytes_min
    eps_sharded = world > 1 and (args.eps == "sharded" or x_sharded) and not by_color
    # the per-step exchange of the sharded layouts is the library's business when a device-side transport exists: fd_plan_set_comm /
    # fd_plan_set_p2p (+ fd_plan_set_halo for a sharded x) -- the call then reduces its own groups of x, exchanges halo + group sums
    # (ONE launch through the mailboxes, else RCCL) and finishes the step sizes.  Only the gloo dry run stages it through the host.
    in_call = None
    if x_sharded:
        in_call = attach_exchange(pb)
    elif eps_sharded and comm is not None:
        plan.set_comm(comm)
        in_call = "inside the call: group sums " + ("through the peer-to-peer mailboxes" if comm.has_p2p() else "all-gathered by RCCL")
    elif eps_sharded and p2p is not None:
        plan.set_p2p(p2p)
        in_call = "inside the call: group sums through the peer-to-peer mailboxes"
    gather_in_step = world > 1 and args.gather_in_step

    # output buffers: rank r fills out = its slice; rank 0 also owns the assembled nzval (root gather) / every rank the
    # padded slots (all-gather)
    bufs = S.AllGatherBuffers(counts, dev, t_dt)
    asm_owned = None
    if by_color:
        out = torch.zeros(nnz, dtype=t_dt, device=dev)      # every rank: whole nzval, zero except its colours' columns
        asm_owned = torch.full((nnz,), float("nan"), dtype=t_dt, device=dev)      # ... and the assembled Jacobian (do_gather)
    elif world > 1:
        out = bufs.local_view(rank)[: counts[rank]]
    else:
        out = bufs.buf
    full = torch.empty(sum(counts), dtype=t_dt, device=dev) if (world > 1 and rank == 0 and not by_color) else None
    host_bufs = S.AllGatherBuffers(counts, torch.device("cpu"), t_dt) if (world > 1 and backend != "nccl") else None

    def do_gather():
        """Assemble nzval.  nccl: libfdjac's own RCCL calls (fd_comm_gatherv / fd_comm_allgather / fd_comm_allreduce_sum)."""
        if by_color:
            # colour ownership: the assembly is the library's (fd_jacobian_owned_async: zero-fill, this rank's colours, ONE all-reduce) --
            # summing the outputs of plain calls is right only on a fresh buffer (entries this rank does not own keep the previous sum)
            plan.jacobian_owned(f, x, [asm_owned], comm=comm)
            if comm is None:
                asm_owned.copy_(S.all_reduce_owned(asm_owned.cpu(), dist))
            return asm_owned
        if comm is not None:
            if args.gather == "root":
                comm.gatherv(out, full, counts, root=0)
                return full
            comm.allgather(bufs.buf, bufs.maxlen)
            return bufs.buf
        host_bufs.local_view(rank).copy_(bufs.local_view(rank))     # dry run: staged through host memory (gloo)
        host_bufs.gather(rank, dist)
        bufs.buf.copy_(host_bufs.buf)
        return bufs.buf

    enqueue = plan.bind(f, x, [out])   # pointers resolved once: one foreign call per Jacobian, as from compiled code

    # dry run (FDJAC_BENCH_BACKEND=gloo, no mailboxes) of the sharded step-size reduction: the explicit pieces of the C ABI with the
    # exchange staged through host memory -- fd_plan_eps_partials, all-gather of the slots, fd_plan_eps_finalize, FD_EPS_PRECOMPUTED
    eps_host = None
    if eps_sharded and in_call is None:
        pptr, slot = plan.eps_partials(x, rank, world)

        class _Raw:   # the library's group-sum buffer as a torch view (zero copy)
            __cuda_array_interface__ = {"shape": (world * slot,), "typestr": "<f8", "data": (pptr, False), "version": 2}
        eps_host = {"dev": torch.as_tensor(_Raw(), device=dev), "slot": slot, "all": torch.empty(world * slot, dtype=torch.float64)}
        plan.set_eps_mode(True)

    def pre_step():
        """Dry run only: what the library does inside the call when it has a transport -- the neighbour halo of a sharded x, then
        the exchange of the sharded reduction's group sums -- staged through the host."""
        if x_sharded:
            xh = x.cpu()
            S.halo_exchange_host(xh, cuts, rank, 2, dist)
            x.copy_(xh)
        if eps_host is not None:
            plan.eps_partials(x, rank, world)
            sl = eps_host["slot"]
            mine = eps_host["dev"][rank * sl:(rank + 1) * sl].cpu()
            dist.all_gather_into_tensor(eps_host["all"], mine)
            eps_host["dev"].copy_(eps_host["all"])
            plan.eps_finalize()

    def step():
        if in_call is None and (x_sharded or eps_host is not None):
            pre_step()
        enqueue()
        if gather_in_step:
            do_gather()

    for _ in range(args.warmup):
        step()
    fence()
    # HIP events around the graded kernel, on the launch stream, on every 4th timed step (the 1st, 5th, ...): a pair of events costs ~8 us
    # of stream time per call (scripts/ubench/ext_launch_probe.hip) -- on every step it was 7 % of the headline step, 40 % of c2's
    plan.enable_timing(1)
    plan.set_timing_stride(4 if args.steps >= 8 else 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    tm = plan.timings()
    dec_samples = plan.timing_samples("decompress")          # the graded kernel, one HIP-event span per timed step
    timed_result = out.clone()
    plan.set_timing_stride(1)
    # per-stage breakdown and the MEDIAN of individually timed calls from a separate, untimed pass (every stage and the
    # whole call bracketed by HIP events on the launch stream: more marker packets than the timed region carries)
    plan.enable_timing(2)
    for _ in range(max(args.steps, 20)):
        enqueue()
    torch.cuda.synchronize()
    tm_all = plan.timings()
    plan.enable_timing(3)   # the whole call only (2 events per call): the individually timed runs the median is taken from
    for _ in range(max(args.steps, 20)):
        enqueue()
    torch.cuda.synchronize()
    call_samples = plan.timing_samples("total")
    plan.enable_timing(0)
    # the FUSED step (round 6; one GPU: N <= 2^21): no launch of its own for the step sizes -- the storing launch carries the reduction
    # (its first workgroups) and reads x twice
    fused_step = bool(lazy_store and cfg in ("c2", "c4") and tm_all["eps"]["launches"] == 0 and tm_all["decompress"]["launches"] > 0)
    if fused_step:
        kern = "k_f_tridiag_fused<0, false, 4, false>" if vs == 8 else "k_f_tridiag_fused4<0, false, 4, false>"
        bytes_min = 2 * vs + 3 * vs
        bytes_call_model = bytes_min
    # ---- side measurement (single GPU, untimed): a DIFFERENT x every call.  The timed steps re-use one x, which then sits in the
    # 256 MiB Infinity Cache when the next call starts (the state a time-stepping loop is in, too: x was just written); here four
    # copies of x (more than the cache holds at the headline size) are walked round-robin, so every call finds its x cold
    rotating = None
    if world == 1 and not args.no_plain_handover:
        try:
            xs = [x.clone() for _ in range(4)]
            calls_r = [plan.bind(f, xi, [out]) for xi in xs]
            for c in calls_r:
                c()
            torch.cuda.synchronize()
            plan.enable_timing(3)
            for i in range(max(args.steps, 20)):
                calls_r[i % 4]()
            torch.cuda.synchronize()
            rot = plan.timing_samples("total")
            plan.enable_timing(0)
            rotating = {"what": "the same call on four copies of x walked round-robin (4 x %.0f MB): x is never cache-resident from the call before; "
                                "median of %d individually timed calls" % (x.numel() * x.element_size() / 1e6, len(rot)),
                        "median_ms_per_step": float(np.median(rot)) if rot else None,
                        "bit_identical_to_timed_result": bool(torch.equal(out, timed_result))}
            del xs, calls_r
        except Exception as e:      # a side measurement never fails the bench
            rotating = {"error": repr(e)}
    # ---- rank share (single GPU, untimed; scripts/bench_sides.py): ONE rank's share of the 8-rank sharded step of this problem, alone on this
    # GPU through a loop-back mailbox -- its groups of the reduction, the exchange (stores into a local sink, the peers' values pre-filled),
    # the storing launch on N / 8 columns: as three launches and as the fused ONE-launch step; checked bit for bit against the unsharded
    # slice.  A measured per-rank floor for the 8-GPU job nobody could run (no xGMI hop in it, no waiting for a slower peer).
    rank_share = None
    if world == 1 and cfg in ("c2", "c4") and not args.no_plain_handover:
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import bench_sides
            rank_share = bench_sides.rank_share(fd, torch, ctx, N, seed, 8, steps=100, np_dt=np_dt)
        except Exception as e:      # a side measurement never fails the bench
            rank_share = {"error": "%s: %s" % (type(e).__name__, e)}
    # ---- side measurements (single GPU, untimed; scripts/bench_sides.py): buffer placements, the hand-over path, the opaque-f! call, the
    # runtime-compiled functor, the drop-in call -- run with --sides / --sweep, not in the driver's default run
    placements = handover = opaque = jit_path = dropin = None
    if world == 1 and args.sides and not args.no_plain_handover:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import bench_sides
        sd = bench_sides.side_runs(dict(locals()))
        placements, handover, opaque, jit_path, dropin = (sd.get(k) for k in ("placements", "handover", "opaque", "jit_path", "dropin"))

    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    ms_step = elapsed / args.steps * 1e3

    # ---- the assembly of nzval, measured on its own right after the timed region (HIP events on the launch stream) ----
    gather_info = None
    assembled = None
    if world > 1:
        try:
            gs = max(3, min(args.steps, 10))
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * gs)]
            assembled = do_gather()                     # warm-up (RCCL connects its channels on first use)
            fence()
            for k in range(gs):
                ev[2 * k].record()
                assembled = do_gather()
                ev[2 * k + 1].record()
            fence()
            ms_g = torch.tensor([sum(ev[2 * k].elapsed_time(ev[2 * k + 1]) for k in range(gs)) / gs], dtype=torch.float64, device=dev)
            dist.all_reduce(ms_g, op=dist.ReduceOp.MAX)
            gbytes = float(sum(counts) - counts[0]) * vs if (args.gather == "root" and not by_color) else float(sum(counts)) * vs
            gather_info = {"kind": ("fd_jacobian_owned_async: zero-fill + this rank's colours + all-reduce(sum)" if by_color else
                                    "fd_comm_gatherv to rank 0 (grouped ncclSend/ncclRecv)" if args.gather == "root" else
                                    "fd_comm_allgather (in-place ncclAllGather)") if comm is not None else "gloo dry run",
                           "ms": float(ms_g.item()), "bytes_over_links": gbytes, "in_timed_step": bool(gather_in_step)}
        except Exception as e:  # the timed result above stands even if the epilogue fails; say so loudly
            gather_info = {"error": "%s: %s" % (type(e).__name__, e)}
            sys.stderr.write("[bench rank %d] gather failed: %s\n" % (rank, e))

    # ---- the consumer (untimed side measurement): one implicit-Euler / Rosenbrock stage (I - gamma*J) y = b solved on the
    # nzval the Jacobian just wrote, where it lies -- sharded across the ranks when N>1 (8 numbers per rank exchanged) --
    consumer = None
    if cfg in ("c2", "c4") and not by_color and args.dtype == "f64" and (world == 1 or comm is not None):
        try:
            solver = fd.TridiagSolver(N, "csc", rows=(c0, c1) if world > 1 else None, ctx=ctx)
            rhs = torch.ones(c1 - c0, dtype=t_dt, device=dev)
            ysol = torch.empty(c1 - c0, dtype=t_dt, device=dev)
            gamma = 0.05
            reps = max(3, min(args.steps, 10))
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
            solver.solve([out], rhs, ysol, 1.0, -gamma, comm=comm)      # warm-up
            fence()
            for k in range(reps):
                evs[2 * k].record()
                solver.solve([out], rhs, ysol, 1.0, -gamma, comm=comm)
                evs[2 * k + 1].record()
            fence()
            ms_s = torch.tensor([sum(evs[2 * k].elapsed_time(evs[2 * k + 1]) for k in range(reps)) / reps], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(ms_s, op=dist.ReduceOp.MAX)
            # residual of the local rows that do not touch a neighbour (exact stencil: J = tridiag(1, -2, 1))
            yy = ysol.double()
            r_in = (1.0 + 2.0 * gamma) * yy[1:-1] - gamma * (yy[:-2] + yy[2:]) - 1.0
            consumer = {"what": "fd_tridiag_solve_async: (I - gamma*J) y = b on the sharded nzval (CSC layout), gamma = 0.05",
                        "ms": float(ms_s.item()), "max_interior_residual": float(r_in.abs().max().item()),
                        "exchange": "one all-gather of 8 doubles per rank" if world > 1 else None}
        except Exception as e:
            consumer = {"error": "%s: %s" % (type(e).__name__, e)}
            sys.stderr.write("[bench rank %d] consumer failed: %s\n" % (rank, e))

    # ---- verification (untimed): the timed steps produced the right thing -----------------------------------------
    # (1) recompute into a NaN-filled buffer: same bits as the timed result (a step that wrote nothing would leave NaN);
    # (2) every stored value of the linear fixture equals the exact stencil weight; (3) the same plan with the NONLINEAR
    # fixture (J depends on x: a stale buffer or a wrong step size cannot pass) against the analytic Jacobian.
    check = {}
    out.fill_(float("nan"))
    enqueue()
    torch.cuda.synchronize()
    if by_color:
        # colour ownership: this rank writes only the stored values of its colours' columns and leaves the rest untouched
        pall = torch.arange(out.numel(), device=dev, dtype=torch.int64)
        ccol = ((pall + 1) // 3) % 3                                # tridiagonal CSC: column of entry p, its 0-based colour
        mine = (ccol >= int(ccuts[rank])) & (ccol < int(ccuts[rank + 1]))
        check["recomputed_from_nan_bit_identical"] = (bool(torch.equal(out[mine], timed_result[mine])) and not bool(torch.isnan(out[mine]).any())
                                                      and bool(torch.isnan(out[~mine]).all()))
    else:
        check["recomputed_from_nan_bit_identical"] = bool(torch.equal(out, timed_result)) and not bool(torch.isnan(out).any())
    if cfg in ("c2", "c4") and not by_color:
        e0 = sum(counts[:rank]) if world > 1 else 0                  # global index of this rank's first stored value
        pidx = torch.arange(out.numel(), device=dev, dtype=torch.int64) + e0
        isdiag = (pidx % 3) == 0                                     # CSC order: column 0 = (d, dl), column j = (du, d, dl)
        exact = torch.where(isdiag, torch.full_like(out, -2.0), torch.ones_like(out))
        check["max_dev_from_exact_stencil_all_entries"] = float((out - exact).abs().max().item())
        f_nl = fd.BuiltinF("tridiag_nl", N, ctx=ctx, dtype=np_dt)
        plan.set_lazy(f_nl if f_mode == "lazy" else None)
        nl = torch.full_like(out, float("nan"))
        plan.jacobian(f_nl, x, [nl])
        xd = x.double()
        xp = torch.cat([xd[1:], xd.new_zeros(1)])
        col = (pidx + 1) // 3                                        # column of stored value p
        want = torch.where(isdiag, -2.0 + 2.0 * xd[col] * xp[col],                       # df_j/dx_j
                           torch.where((pidx % 3) == 2, 1.0 + xd[torch.clamp(col - 1, min=0)] ** 2,   # df_{j-1}/dx_j
                                       torch.ones_like(xd[col])))                       # df_{j+1}/dx_j
        check["nonlinear_fixture_max_abs_err_vs_analytic"] = float((nl.double() - want).abs().max().item())
        check["nonlinear_fixture_tolerance"] = 2e-6 if args.dtype == "f64" else 2e-2
        plan.set_lazy(f if f_mode == "lazy" else None)
    if assembled is not None and by_color and "error" not in (gather_info or {}):
        # the assembly has run gs + 1 >= 4 times on the same buffer by now: every stored value must still be the exact stencil weight
        pa = torch.arange(assembled.numel(), device=dev, dtype=torch.int64)
        exact_a = torch.where((pa % 3) == 0, torch.full_like(assembled, -2.0), torch.ones_like(assembled))
        check["assembled_after_repeated_gathers_max_dev_from_exact"] = float((assembled - exact_a).abs().max().item())
        check["assembled_ok"] = check["assembled_after_repeated_gathers_max_dev_from_exact"] <= (1e-6 if args.dtype == "f64" else 1e-2)
    if assembled is not None and rank == 0 and not by_color and "error" not in (gather_info or {}):
        a = assembled if args.gather == "root" or comm is None else bufs.compact()
        check["assembled_slice_matches_local"] = bool(torch.equal(a[: counts[0]], timed_result))
    ok = check.get("recomputed_from_nan_bit_identical", False)
    if "nonlinear_fixture_max_abs_err_vs_analytic" in check:
        ok = ok and check["nonlinear_fixture_max_abs_err_vs_analytic"] <= check["nonlinear_fixture_tolerance"]
        ok = ok and check["max_dev_from_exact_stencil_all_entries"] <= (1e-6 if args.dtype == "f64" else 1e-2)
    if "assembled_ok" in check:
        ok = ok and check["assembled_ok"]
    check["ok"] = bool(ok)

    p2p_st = p2p_status() if world > 1 else 0
    if p2p_st:
        check["ok"] = False
        check["p2p_timeout_waiting_for_rank"] = p2p_st - 1
    if x_sharded and cfg in ("c2", "c4") and not by_color:
        # the sharded layout: this rank's x holds its own part and the halo the calls brought in -- and nothing else
        lo, hi, xw_true = pb["x_full_window"]
        check["sharded_x_holds_own_part_and_halo_only"] = bool(torch.equal(x[lo:hi], xw_true) and torch.isnan(x[:lo]).all() and torch.isnan(x[hi:]).all())
        check["ok"] = bool(check["ok"] and check["sharded_x_holds_own_part_and_halo_only"])
    # ---- per-rank diagnostics (stderr): which device / RCCL each rank saw and where its time went ---------------------
    stages = {k: (v["ms_sum"] / max(v["launches"], 1)) for k, v in tm_all.items()}
    diag = {"rank": rank, "world": world, "device": dev_index, "device_name": torch.cuda.get_device_name(dev_index),
            "columns": [c0, c1], "stored_values": int(counts[rank] if world > 1 and not by_color else sum(counts)),
            "stages_ms": stages, "plan_build_ms": plan_build_ms, "check": check,
            "rccl": comm.info() if comm is not None else None, "backend": backend if world > 1 else None,
            "eps": ("sharded" if eps_sharded else "replicated"), "x_layout": "sharded (halo exchange per step)" if x_sharded else "replicated",
            "gather": gather_info, "consumer": consumer}
    sys.stderr.write("[bench rank %d] %s\n" % (rank, json.dumps(diag)))
    sys.stderr.flush()

    if rank == 0:
        n_local = c1 - c0
        dec = tm["decompress"]
        dec_ms = dec["ms_sum"] / max(dec["launches"], 1)
        dec_med = float(np.median(dec_samples)) if dec_samples else dec_ms
        call_med = float(np.median(call_samples)) if call_samples else None
        tot_ms = tm_all["total"]["ms_sum"] / max(tm_all["total"]["launches"], 1)
        pmc, pmc_src = None, None
        # HBM bytes per launch of the graded kernel from the committed rocprofv3 PMC passes of this same command
        # (scripts/profile.sh + scripts/make_pmc_json.py; counters cannot be read from inside the process)
        pmc_path = os.path.join(ROOT, "profiles", "pmc_%s.json" % cfg)
        if os.path.exists(pmc_path):
            try:
                j = json.load(open(pmc_path))
                if (int(j.get("n", -1)) == N and int(j.get("gpus", 1)) == world and j.get("kernel", "") in kern
                        and args.dtype == "f64" and int(j.get("lazy_diff", 0)) == lazy_diff
                        and int(j.get("lazy_store", 0)) == lazy_store):
                    pmc = j.get("decompress_hbm_bytes_per_launch")
                    pmc_src = ("builder PMC (committed): rocprofv3 --pmc passes of this command, profiles/pmc_%s.json "
                               "(FETCH_SIZE x2 + WRITE_SIZE, calibrated on the 1 GiB stream copy of the same run); not measured in this process" % cfg)
            except Exception:
                pmc = None
        # ... or measured now, when rocprofv3 is on the box (one GPU, the default run: two short passes of this command)
        pmc_live = None
        if world == 1 and not args.no_plain_handover and not os.environ.get("FDJAC_BENCH_PMC_CHILD"):
            klike = kern.split("<")[0]
            pmc_live, why = live_pmc(cfg, args.dtype, klike)
            if pmc_live:
                pmc = pmc_live["bytes"]
                pmc_src = ("live: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of a 5-step run of this command, started by "
                           "this bench.py; calibrated on the 1 GiB stream copy of the same run (fetch x%.3f, write x%.3f)" % (pmc_live["fetch_factor"], pmc_live["write_factor"]))
            elif pmc_src:
                pmc_src += "; live pass: " + str(why)
            why_live = why
        else:
            why_live = "not attempted (side runs off / several GPUs)"
        traffic = pmc if pmc else bytes_min * n_local
        traffic_src = pmc_src if pmc else "floor: the bytes this kernel must move (no committed PMC pass matches this run; live pass: %s)" % why_live
        achieved = traffic / (dec_ms * 1e-3) / 1e9 if dec_ms > 0 else 0.0
        survey_gbps = bytes_ds * n_local / (dec_ms * 1e-3) / 1e9 if dec_ms > 0 else 0.0
        if dropin and "ms" in dropin and call_med:
            dropin["median_ms_per_step"] = call_med
            dropin["ratio_to_median_ms_per_step"] = dropin["ms"] / call_med
            for v in dropin.get("variants", {}).values():
                if isinstance(v, dict) and "ms" in v:
                    v["ratio_to_median_ms_per_step"] = v["ms"] / call_med
        if placements and "median_ms" in placements:
            for key in ("min_ms", "median_ms", "max_ms"):
                placements["frac_at_" + key] = traffic / (placements[key] * 1e-3) / 1e9 / HBM_PEAK_GBPS
        res = {
            "metric": "Jacobian columns/s (coloured sparse finite-difference Jacobian; headline config N=10^7 tridiagonal forward)",
            "value": N / (ms_step * 1e-3),
            "unit": "columns/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak" if (args.weak and world > 1) else "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": wl, "name": cfg, "fdtype": fdtype, "colors": C,
                       "parallelism": ("colours x%d" % world) if by_color else ("columns x%d" % world),
                       "output": ("nzval in HBM" if world == 1 else
                                  "nzval assembled on rank 0 inside the step" if gather_in_step else
                                  "nzval device-resident, sharded by column range (rank r holds its contiguous slice)"),
                       "f_mode": ("built-in device f! behind fd_f_launch_lazy with FD_LAZY_CAP_STORE (1 launch: lazily perturbed points, "
                                  "difference quotients stored into nzval by the same launch)" if (f_mode == "lazy" and lazy_store) else
                                  "built-in device f! behind fd_f_launch_lazy with FD_LAZY_CAP_DIFF (1 launch: lazily perturbed points, "
                                  "written as differences from f(x) / from the minus point)" if (f_mode == "lazy" and lazy_diff) else
                                  "built-in device f! behind fd_f_launch_lazy (1 launch: base + lazily perturbed points)"
                                  if f_mode == "lazy" else
                                  "built-in device f! behind fd_f_launch (materialised points, one batched launch)"),
                       "eps_reduction": diag["eps"], "x_layout": diag["x_layout"], "gather_in_step": bool(gather_in_step),
                       "small_messages": (p2p_note or "rccl") if world > 1 else None,
                       "variant": ((("weak" if args.weak else "strong") + "-" + ("sharded" if x_sharded else "replicated")) if (world > 1 and not by_color) else None),
                       "variant_selection": ("fastest of the four timed in this run (see `variants`)" if variants else "as the flags say") if world > 1 else None,
                       "step_exchange": in_call if world > 1 else None,
                       "problem": ("N = %d = %d x 10^7 columns (weak scaling)" % (N, world)) if (args.weak and world > 1) else None,
                       "collective_backend": (("rccl via libfdjac fd_comm_* (%s)" % comm.info()["library"]) if comm is not None
                                              else backend) if world > 1 else None},
            "value_cold_x": (N / (rotating["median_ms_per_step"] * 1e-3)) if (rotating and rotating.get("median_ms_per_step")) else None,
            "value_cold_x_note": "the same call on an x that is NOT cache-resident from the call before (rotating_x): the rate a caller "
                                 "who produces a fresh x elsewhere sees; quote the lower of value / value_cold_x",
            "median_ms_per_step": call_med,
            "value_median": (N / (call_med * 1e-3)) if call_med else None,
            "median_note": "median GPU time of %d individually timed calls (HIP events on the launch stream around the whole call, "
                           "separate pass); `value` / `ms_per_step` are the contract's K bracketed steps" % len(call_samples),
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc, "traffic_source": traffic_src, "traffic_live": pmc_live,
                "kernel": kern,
                "kernel_role": ("f! at the lazily perturbed points + difference + division + store into nzval in ONE launch "
                                "(fd_lazy_points.store, include/fdjac_device.h): src/jacobians.jl:563-568 for all colours" if lazy_store else
                                "division + decompression of the differences handed over by the lazy f! launcher" if lazy_diff
                                else "fused difference + decompression"),
                "lazy_diff": lazy_diff, "lazy_store": lazy_store, "fused_step": fused_step,
                "avg_launch_ms": dec_ms, "median_launch_ms": dec_med, "launches_timed": dec["launches"],
                "achieved_on_median": (traffic / (dec_med * 1e-3) / 1e9) if dec_med > 0 else None,
                "hbm_bytes_per_launch_used": traffic,
                "min_traffic_bytes_per_launch": bytes_min * n_local,
                "algorithmic_bytes_per_launch": bytes_ds * n_local,
                "survey_equivalent": {
                    "gbps": survey_gbps, "frac_survey_bytes": survey_gbps / HBM_PEAK_GBPS,
                    "note": "SURVEY 8(d) algorithmic bytes / kernel time: counts index reads and per-colour re-reads of f(x) "
                            "this kernel does not perform -- an equivalent-work rate, NOT a bandwidth (it can exceed the peak)"},
            },
            "rank_share": rank_share,
            "placements": placements,
            "handover_path": handover,
            "rotating_x": rotating,
            "opaque_f_path": opaque,
            "jit_functor_path": jit_path,
            "dropin_call": dropin,
            "stages_ms": stages,
            "whole_call": {"gpu_ms": tot_ms, "hbm_bytes_model": bytes_call_model * n_local,
                           "gbps": bytes_call_model * n_local / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0,
                           "frac_of_peak": bytes_call_model * n_local / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if tot_ms > 0 else 0.0,
                           "survey_bytes": bytes_call_survey * n_local,
                           "note": "hbm_bytes_model = bytes the stages of this implementation move (eps: x; then either the storing f! "
                                   "launch: x in, values out -- or lazy f!: x, colours, C outputs + decompression: f! outputs + values); "
                                   "survey_bytes = SURVEY 8(d)'s 210 B/column of the reference's pass structure, kept for comparison only"},
            "plan_build_ms": plan_build_ms,
            "gather": gather_info,
            "comm_error": comm_error,
            "variants": variants,
            "p2p_status": (p2p_st if world > 1 else None),
            "consumer": consumer,
            "value_with_gather": (N / ((ms_step + (0.0 if gather_in_step else gather_info["ms"])) * 1e-3)
                                  if (gather_info and "ms" in gather_info) else None),
            "result_check": check,
        }
        try:
            res["stream_copy_gbps"] = ctx.stream_copy_gbps(1 << 30, 10)
            res["roofline"]["frac_of_copy_ceiling"] = achieved / res["stream_copy_gbps"]
        except Exception as e:  # pragma: no cover
            res["stream_copy_gbps"] = None
            res["stream_copy_error"] = str(e)
        if not args.no_cpu_baseline and world == 1 and cfg in ("c2", "c4") and args.dtype == "f64":
            try:
                res["cpu_baseline"] = cpu_baseline(args.cpu_n or N, args.cpu_reps, args.cpu_seconds)
            except Exception as e:  # pragma: no cover
                res["cpu_baseline"] = {"error": str(e)}
        line = json.dumps(res) + "\n"
        if json_fd is not None:
            os.write(json_fd, line.encode())
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
        if args.sweep and world == 1:
            # the other BASELINE configs as child processes (own plans / buffers), one JSON line per file
            import subprocess
            os.makedirs(args.sweep, exist_ok=True)
            with open(os.path.join(args.sweep, "bench_%s%s.json" % (cfg, "_f32" if args.dtype == "f32" else "")), "w") as fh:
                fh.write(line)
            for c, extra in (("c2", []), ("c3", []), ("c5", []), ("c4", ["--dtype", "f32"]), ("c3", ["--dtype", "f32"]), ("c5", ["--dtype", "f32"])):
                if c == cfg and not extra:
                    continue
                cmd = [sys.executable, os.path.abspath(__file__), "--config", c, "--steps", str(args.steps), "--warmup", str(args.warmup),
                       "--no-cpu-baseline", "--soak-seconds", "0"] + extra
                name = "bench_%s%s.json" % (c, "_f32" if extra else "")
                with open(os.path.join(args.sweep, name), "w") as fh, open(os.path.join(args.sweep, name.replace(".json", ".err")), "w") as eh:
                    subprocess.run(cmd, stdout=fh, stderr=eh, timeout=600, check=False)

    # ---- untimed soak: keep the GPU in the steady-state loop long enough for an external sampler to see it ------------
    if args.soak_seconds > 0:
        t_end = time.perf_counter() + args.soak_seconds
        while True:
            for _ in range(200):
                enqueue()
            torch.cuda.synchronize()
            go = torch.tensor([1.0 if time.perf_counter() < t_end else 0.0], dtype=torch.float64, device=dev)
            if world > 1:       # (calls that exchange inside must be made the same number of times on every rank)
                dist.all_reduce(go, op=dist.ReduceOp.MIN)
            if not go.item():
                break
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
        del enqueue, plan                 # release in dependency order: plans, then the communicator, then torch's group
        if comm is not None:
            comm._fin()
        dist.destroy_process_group()
    if not check["ok"]:
        raise SystemExit("bench.py: result check FAILED: %s" % json.dumps(check))


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:      # whatever went wrong: rank 0 still prints the contract's one line, then the failure is reported as usual
        import traceback
        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            try:
                emit_failure(parse(), "%s: %s" % (type(e).__name__, e), JSON_FD)
            except Exception:
                pass
        raise SystemExit(1)
