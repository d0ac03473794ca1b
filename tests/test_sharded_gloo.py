"""N>1 path on CPU: world_size-2 gloo processes exercise the column partition, per-rank windows and
the single all-gather that assembles nzval (finitediff.jl_amd/sharded.py).  The per-rank compute is
stood in for by the CPU oracle (tests may use it): each rank fills only ITS slice, so a wrong
partition / window / gather order cannot produce the full reference vector."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import finitediff_jl_amd as fd  # noqa: F401
from finitediff_jl_amd import patterns as P
from finitediff_jl_amd import sharded as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_properties():
    for n, world in ((10, 1), (10, 3), (1000, 8), (7, 8), (100003, 4)):
        colptr, _ = P.tridiag_csc(n)
        cuts = S.partition_columns(colptr, world)
        assert cuts[0] == 0 and cuts[-1] == n and np.all(np.diff(cuts) >= 0)
        rng = S.entry_ranges(colptr, cuts)
        assert rng[0][0] == 0 and rng[-1][1] == 3 * n - 2
        assert all(a[1] == b[0] for a, b in zip(rng[:-1], rng[1:]))
        counts = np.array([b - a for a, b in rng])
        if n >= 50 * world:
            assert counts.max() - counts.min() <= 3  # balanced by stored entries
    # skewed pattern: a dense first column must not starve the other ranks
    colptr = np.array([1, 101] + list(range(102, 202)), dtype=np.int64)
    cuts = S.partition_columns(colptr, 2)
    assert cuts.tolist() == [0, 1, 101]


def test_x_window():
    cuts = np.array([0, 40, 100])
    assert S.x_window(cuts, 0, 100, 1, 1, 1) == (0, 42)
    assert S.x_window(cuts, 1, 100, 1, 1, 1) == (38, 100)


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %(root)r)
    import finitediff_jl_amd as fd
    from finitediff_jl_amd import patterns as P, sharded as S
    from oracle import oracle
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N = 1001
    x = np.random.default_rng(4).random(N)
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    cuts = S.partition_columns(colptr, world)
    ranges = S.entry_ranges(colptr, cuts)
    counts = [b - a for a, b in ranges]
    full = oracle.jacobian("forward", oracle.Fixture("tridiag_nl", N), x, colors, kind=oracle.PAT_CSC_COMMON,
                           colptr=colptr, rowval=rowval)["out"]
    # this rank "computes" only its own slice, straight into its slot of the gather buffer
    bufs = S.AllGatherBuffers(counts, torch.device("cpu"), torch.float64)
    bufs.buf.fill_(float("nan"))
    a, b = ranges[rank]
    bufs.local_view(rank)[: counts[rank]] = torch.from_numpy(full[a:b])
    bufs.gather(rank, dist)
    got = bufs.compact().numpy()
    assert got.shape == full.shape and np.array_equal(got, full), "rank %%d: gathered vector differs" %% rank
    # the allocation-per-call variant gives the same answer
    got2 = S.all_gather_slices(torch.from_numpy(full[a:b].copy()), counts, dist).numpy()
    assert np.array_equal(got2, full)
    # every rank derives the same windows
    xw = S.x_window(cuts, rank, N, 1, 1, 1)
    assert xw[0] <= max(cuts[rank] - 2, 0) and xw[1] >= min(cuts[rank + 1] + 2, N)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


@pytest.mark.timeout(300)
def test_two_rank_gather_gloo(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29517", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("ok") == 2


HALO_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %(root)r)
    from finitediff_jl_amd import sharded as S
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, halo = 1000, 3
    cuts = np.array([0, 430, 1000]) if world == 2 else S.partition_columns(np.arange(1, N + 2), world)
    truth = torch.arange(N, dtype=torch.float64) * 0.5 + 1.0
    x = torch.full((N,), float("nan"), dtype=torch.float64)
    a, b = int(cuts[rank]), int(cuts[rank + 1])
    x[a:b] = truth[a:b]                       # every rank holds its own part only
    S.halo_exchange_host(x, cuts, rank, halo, dist)
    lo, hi = max(a - halo, 0), min(b + halo, N)
    assert torch.equal(x[lo:hi], truth[lo:hi]), "rank %%d: halo values wrong" %% rank
    rest = torch.cat([x[:lo], x[hi:]])
    assert bool(torch.isnan(rest).all()), "rank %%d: wrote outside own range + halo" %% rank
    # the column cuts taken from the reduction's shard ranges
    cuts2 = S.partition_columns_at([(0, 500), (500, 900), (900, 1000)], 1000)
    assert cuts2.tolist() == [0, 500, 900, 1000]
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


@pytest.mark.timeout(300)
def test_two_rank_halo_exchange_gloo(tmp_path):
    # the host-side mirror of fd_comm_halo_exchange (the x of a time-stepping loop stays sharded: 2 x halo values per link and step)
    script = tmp_path / "hworker.py"
    script.write_text(HALO_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29525", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29525", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("ok") == 2


@pytest.mark.gpu
@pytest.mark.timeout(400)
@pytest.mark.parametrize("variant", ["x_sharded", "eps_sharded", "colors", "x_sharded_p2p", "weak"])
def test_bench_two_ranks_dry_run_variants(variant):
    """bench.py's other N>1 decompositions end to end (two ranks sharing the one GPU, gloo transport): --x-layout sharded (halo
    exchange + contiguous sharded step-size reduction before every Jacobian: the time-stepping layout), --eps sharded (the
    explicit fd_plan_eps_partials / all-gather / fd_plan_eps_finalize pieces) and --shard colors (colour ownership +
    all-reduce).  bench.py checks every stored value itself and exits non-zero if a check fails."""
    import json
    port = {"x_sharded": "29527", "eps_sharded": "29529", "colors": "29531", "x_sharded_p2p": "29533", "weak": "29535"}[variant]
    extra = {"x_sharded": ["--x-layout", "sharded"], "eps_sharded": ["--eps", "sharded"], "colors": ["--shard", "colors"],
             # the time-stepping layout with the halo and the partial sums travelling through the peer-to-peer mailboxes (fd_p2p_*, on the GPU)
             "x_sharded_p2p": ["--x-layout", "sharded", "--small-messages", "p2p"],
             "weak": ["--weak"]}[variant]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, FDJAC_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--size", "300001", "--soak-seconds", "0"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=380)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["result_check"]["ok"] and res["result_check"]["recomputed_from_nan_bit_identical"]
    if variant == "x_sharded":
        assert res["config"]["x_layout"].startswith("sharded") and res["config"]["eps_reduction"] == "sharded"
    if variant == "eps_sharded":
        assert res["config"]["eps_reduction"] == "sharded"
    if variant == "colors":
        assert res["config"]["parallelism"] == "colours x2"
    if variant == "x_sharded_p2p":
        assert res["config"]["x_layout"].startswith("sharded") and res["config"]["small_messages"].startswith("fd_p2p_")
    if variant == "weak":
        assert res["scaling"] == "weak"


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_spawns_its_own_ranks_and_times_every_variant():
    """`python bench.py --gpus 2` with NO launcher (the form the driver uses for --gpus 1): bench.py starts the ranks itself, times
    all four variants (strong / weak x replicated / sharded with the one-launch exchange through the mailboxes -- two processes
    sharing the one GPU, gloo for the barriers), reports the fastest as `value` and the rest in `variants`; ONE JSON line."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(FDJAC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--size", "300001", "--soak-seconds", "0.3"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-6000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["result_check"]["ok"], res["result_check"]
    names = [v["name"] for v in res["variants"]]
    assert names == ["strong-replicated", "strong-sharded", "weak-replicated", "weak-sharded"]
    assert all("error" not in v and v["ms_per_step"] > 0 for v in res["variants"]), res["variants"]
    assert res["config"]["variant"] in names and res["scaling"] == res["config"]["variant"].split("-")[0]
    sh = [v for v in res["variants"] if v["name"].endswith("sharded")]
    assert all("ONE launch through the peer-to-peer mailboxes" in v["exchange"] for v in sh)
    assert res["p2p_status"] == 0


@pytest.mark.timeout(120)
def test_bench_without_gpu_still_prints_one_line():
    """Whatever goes wrong, the contract's ONE JSON line is printed: here a 2-rank self-spawn on a box without a GPU (the CPU container) --
    and on a GPU box a spawn whose ranks cannot finish in time."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "300001", "--soak-seconds", "0"]
    if has_gpu:
        cmd += ["--spawn-timeout", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    res = json.loads(lines[0])
    assert res["value"] is None and res["error"] and res["n_gpus"] == 2
    assert out.returncode != 0


@pytest.mark.gpu
@pytest.mark.timeout(400)
@pytest.mark.parametrize("in_step", [False, True], ids=["sharded_output", "gather_in_step"])
def test_bench_two_ranks_share_one_gpu(tmp_path, in_step):
    """The N>1 path of bench.py end to end on the GPU box: two ranks (sharing the one GPU, gloo transport, the
    configuration FDJAC_BENCH_BACKEND=gloo exists for) each compute their column range with a windowed plan; nzval
    stays sharded in the timed step (default) or is assembled inside it (--gather-in-step).  bench.py itself checks
    every stored value (exact stencil, nonlinear fixture vs the analytic Jacobian, recompute-from-NaN) and exits
    non-zero if a check fails."""
    import json
    port = "29519" if in_step else "29523"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, FDJAC_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--size", "300001", "--soak-seconds", "0", "--variant", "flags"] + (["--gather-in-step"] if in_step else [])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=380)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                       # rank 0 prints ONE JSON line on stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["gather_in_step"] is in_step
    chk = res["result_check"]
    assert chk["ok"] and chk["recomputed_from_nan_bit_identical"] and chk["assembled_slice_matches_local"]
    assert chk["max_dev_from_exact_stencil_all_entries"] < 1e-7 and chk["nonlinear_fixture_max_abs_err_vs_analytic"] < 2e-6
    assert res["gather"]["ms"] > 0 and res["value"] > 0 and res["value_with_gather"] > 0
    assert res["roofline"]["frac"] <= 1.0 and res["whole_call"]["gbps"] <= res["roofline"]["peak"]
    diags = [ln for ln in out.stderr.splitlines() if ln.startswith("[bench rank ")]
    assert len(diags) == 2                       # every rank reports its device / stage times on stderr


def test_partition_colors():
    for C, world in ((3, 1), (3, 2), (3, 3), (5, 2), (96, 8), (96, 5)):
        colors = (np.arange(1000) % C) + 1
        cuts = S.partition_colors(colors, world)
        assert cuts[0] == 0 and cuts[-1] == C and np.all(np.diff(cuts) >= 0)
        if C >= world:
            assert np.all(np.diff(cuts) >= 1) and np.diff(cuts).max() - np.diff(cuts).min() <= 1
    # more ranks than colours: the surplus ranks own nothing
    cuts = S.partition_colors(np.array([1, 2, 3, 1, 2, 3]), 8)
    assert cuts[0] == 0 and cuts[-1] == 3 and np.diff(cuts).sum() == 3
    # weighted: a heavy first colour gets a rank of its own
    assert S.partition_colors(np.array([1, 2, 3, 4]), 2, weights=[10, 1, 1, 1]).tolist() == [0, 1, 4]


COLOR_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %(root)r)
    import finitediff_jl_amd as fd
    from finitediff_jl_amd import patterns as P, sharded as S
    from oracle import oracle
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N = 600
    x = np.random.default_rng(6).random(N)
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    cuts = S.partition_colors(colors, world)
    full = oracle.jacobian("forward", oracle.Fixture("tridiag_nl", N), x, colors, kind=oracle.PAT_CSC_COMMON,
                           colptr=colptr, rowval=rowval)["out"]
    # this rank "computes" only the stored values of the columns whose colour it owns (oracle stand-in), rest zero
    col_of = P.csc_cols(colptr) - 1
    mine = (colors[col_of] - 1 >= cuts[rank]) & (colors[col_of] - 1 < cuts[rank + 1])
    local = torch.from_numpy(np.where(mine, full, 0.0))
    got = S.all_reduce_owned(local, dist).numpy()
    assert np.array_equal(got, full), "rank %%d: assembled vector differs" %% rank
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


@pytest.mark.timeout(300)
def test_two_rank_colour_ownership_gloo(tmp_path, oracle):
    script = tmp_path / "cworker.py"
    script.write_text(COLOR_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29521", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29521", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("ok") == 2
