"""Is fd_jacobian_async capturable into a HIP graph (torch.cuda.CUDAGraph on the launch stream)?  Replay time vs stream launches."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P

def run(N, reps=200):
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    ctx = fd.Context(0)
    cp, rv = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, cp, rv, None)
    colors = P.cyclic_colors(N, 3)
    plan = fd.make_plan(J, J, colors, "forward", ctx=ctx)
    f = fd.BuiltinF("tridiag_nl", N, ctx=ctx)
    plan.set_lazy(f)
    x = torch.rand(N, dtype=torch.float64, device="cuda")
    out = torch.full((rv.size,), float("nan"), dtype=torch.float64, device="cuda")
    enq = plan.bind(f, x, [out])
    for _ in range(3):
        enq()
    torch.cuda.synchronize()
    ref = out.clone()
    out.fill_(float("nan"))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        enq()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    same = bool(torch.equal(out, ref))
    # a new x through the same graph (the pointers are baked in, the contents are not)
    x.mul_(1.5); g.replay(); torch.cuda.synchronize(); got = out.clone()
    enq(); torch.cuda.synchronize()
    same2 = bool(torch.equal(out, got))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): enq()
    e1.record(); torch.cuda.synchronize(); t_stream = e0.elapsed_time(e1) / reps
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize(); t_graph = e0.elapsed_time(e1) / reps
    t = time.perf_counter()
    for _ in range(reps): enq()
    host_stream = (time.perf_counter() - t) / reps * 1e3
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): g.replay()
    host_graph = (time.perf_counter() - t) / reps * 1e3
    torch.cuda.synchronize()
    print("N=%d graph==stream %s, new x %s | GPU ms per Jacobian: stream %.4f graph %.4f | host ms per enqueue: stream %.4f graph %.4f"
          % (N, same, same2, t_stream, t_graph, host_stream, host_graph))

for N in (10 ** 4, 10 ** 5, 10 ** 6, 10 ** 7):
    run(N)
