"""What do the colour reads cost the step-size launch?  N = 10^7, 5 cyclic colours: arithmetic colours against FDJAC_EPS_CYCLIC=0 (read)."""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    import finitediff_jl_amd as fd
    from finitediff_jl_amd import patterns as P
    N, C = 10 ** 7, int(sys.argv[2])
    colors = P.cyclic_colors(N, C)
    cp, rv = P.banded_csc(N, N, 2, 2) if C == 5 else P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, cp, rv, None)
    plan = fd.make_plan(J, J, colors, "forward")
    x = torch.as_tensor(np.random.default_rng(1).random(N), device="cuda")
    f = fd.BuiltinF("tridiag", N)
    out = torch.zeros(rv.size, dtype=torch.float64, device="cuda")
    ts = []
    for it in range(30):
        plan.enable_timing(2)
        try:
            plan.jacobian(f, x, [out])
        except Exception as e:
            if it == 0:
                sys.stderr.write("jacobian: %r\n" % (e,))
        torch.cuda.synchronize()
        s = plan.timing_samples("eps")
        if s:
            ts.append(s[-1] * 1e3)
        plan.enable_timing(0)
    print(json.dumps({"C": C, "cyclic_env": os.environ.get("FDJAC_EPS_CYCLIC", "1"), "eps_us_median": float(np.median(ts[5:])) if ts else None}))
else:
    for C in (3, 5):
        for cyc in ("1", "0"):
            env = dict(os.environ, FDJAC_TEST_SWITCHES="1", FDJAC_EPS_CYCLIC=cyc)
            r = subprocess.run([sys.executable, __file__, "child", str(C)], env=env, capture_output=True, text=True)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], r.stderr[-400:] if "jacobian:" in r.stderr or "Error" in r.stderr else "")
