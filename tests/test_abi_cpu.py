"""CPU-side checks of the drop-in boundary: libfdjac.so builds for gfx950, loads, exports every
symbol include/fdjac.h declares, and fails loudly (no fallback) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import finitediff_jl_amd as fd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    fd.lib.build()
    return fd.lib.load()


def test_header_symbols_all_exported(L):
    hdr = open(os.path.join(ROOT, "include", "fdjac.h")).read()
    declared = set(re.findall(r"^(?:int|void \*|const char \*)\s*(fd(?:32)?_[a-z0-9_]+)\(", hdr, re.M))
    assert len(declared) >= 60 and sum(n.startswith("fd32_") for n in declared) == len(fd.lib.TYPED)
    assert declared == set(fd.lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.fd_version() == 500


def test_no_gpu_fails_loudly(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = L.fd_ctx_create(0, None, C.byref(h))
    assert rc == 7 and b"no HIP device" in L.fd_last_error()  # FD_ERR_NODEVICE
    with pytest.raises(RuntimeError):
        fd.Context(0)
    with pytest.raises(RuntimeError):
        fd.BuiltinF("tridiag", 10)


def test_plan_opts_layout():
    # struct fd_plan_opts: int32 x2 then seven int64 (include/fdjac.h)
    assert C.sizeof(fd.lib.PlanOpts) == 8 + 7 * 8
    assert fd.lib.PlanOpts.col_begin.offset == 8 and fd.lib.PlanOpts.scratch_bytes.offset == 40


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under the product package may reference it
    pkg = os.path.join(ROOT, "finitediff.jl_amd")
    for dp, _dn, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".jl")):
                txt = open(os.path.join(dp, fn)).read()
                assert "oracle" not in txt.lower(), os.path.join(dp, fn)


def test_pattern_builders():
    P = fd.patterns
    colptr, rowval = P.tridiag_csc(6)
    assert colptr.tolist() == [1, 3, 6, 9, 12, 15, 17]
    assert rowval.tolist() == [1, 2, 1, 2, 3, 2, 3, 4, 3, 4, 5, 4, 5, 6, 5, 6]
    # lap5 colouring is a valid distance-2 colouring: no two columns of one colour share a row
    nx, ny = 7, 6
    colptr, rowval = P.lap5_csc(nx, ny)
    colors = P.lap5_colors(nx, ny)
    cols = P.csc_cols(colptr)
    seen = set()
    for r, c in zip(rowval, cols):
        key = (int(r), int(colors[c - 1]))
        assert key not in seen
        seen.add(key)
    lay = P.BlockBandedLayout([2, 3, 2, 4], 1, 1)
    assert lay.data_len == 2 * 5 + 3 * 7 + 2 * 9 + 4 * 6
    d = np.arange(lay.data_len, dtype=float)
    A = lay.to_dense(d)
    r, c = np.nonzero(A + (A == 0) * 0)  # positions; entry 0 of data maps to A[0,0]
    assert A[0, 0] == 0 and A[1, 0] == 1 and A[4, 0] == 4 and A[0, 2] == 10
    assert lay.index_of(np.array([4]), np.array([0]))[0] == 4


def test_default_relstep_and_fdtype_names():
    assert fd.default_relstep("forward") == np.sqrt(np.finfo(float).eps)
    assert fd.default_relstep("Val{:central}") == np.cbrt(np.finfo(float).eps)
    with pytest.raises(ValueError):
        fd.default_relstep("backward")


def _valid_coloring(colptr, rowval, colors):
    cols = fd.patterns.csc_cols(colptr)
    seen = set()
    for r, c in zip(rowval, cols):
        key = (int(r), int(colors[c - 1]))
        if key in seen:
            return False
        seen.add(key)
    return True


def test_matrix_colors_plan_time_colouring(L):
    # colourings generated from the pattern alone (host side, no GPU needed)
    P = fd.patterns
    colptr, rowval = P.tridiag_csc(101)
    c = fd.matrix_colors(fd.SparseMatrixCSC(101, 101, colptr, rowval))
    assert c.min() == 1 and c.max() == 3 and _valid_coloring(colptr, rowval, c)
    colptr, rowval = P.lap5_csc(23, 17)
    c = fd.matrix_colors(fd.SparseMatrixCSC(23 * 17, 23 * 17, colptr, rowval))
    assert 5 <= c.max() <= 8 and _valid_coloring(colptr, rowval, c)
    # random rectangular pattern
    rng = np.random.default_rng(0)
    A = (rng.random((40, 60)) < 0.08).astype(float)
    colptr, rowval = P.csc_from_dense(A)
    c = fd.matrix_colors(fd.SparseMatrixCSC(40, 60, colptr, rowval))
    assert _valid_coloring(colptr, rowval, c) and c.max() <= 60
    # a dense row forces N colours
    A = np.zeros((3, 7)); A[1, :] = 1
    colptr, rowval = P.csc_from_dense(A)
    assert fd.matrix_colors(fd.SparseMatrixCSC(3, 7, colptr, rowval)).tolist() == [1, 2, 3, 4, 5, 6, 7]
    # closed forms
    assert fd.matrix_colors(fd.Tridiagonal(np.zeros(9), np.zeros(10), np.zeros(9))).tolist() == [1, 2, 3, 1, 2, 3, 1, 2, 3, 1]
    assert fd.matrix_colors(fd.BandedMatrix(np.zeros((4, 6), order="F"), 6, 2, 1)).tolist() == [1, 2, 3, 4, 1, 2]


def test_plain_c_client_builds_and_fails_loudly_without_gpu(tmp_path):
    # the C example links against the library with nothing but gcc; without a GPU it must report FD_ERR_NODEVICE
    import subprocess
    import torch
    exe = str(tmp_path / "c_abi_tridiag")
    libdir = os.path.join(ROOT, "finitediff.jl_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_tridiag.c"),
                           "-o", exe, "-L" + libdir, "-lfdjac", "-lm", "-Wl,-rpath," + libdir])
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the run itself is a -m gpu test")
    out = subprocess.run([exe, "30"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "no HIP device" in out.stderr


def test_c_clients_build_and_fail_loudly_without_gpu(tmp_path):
    # examples/c_abi_clients.c (one client per Julia shim method, device pointers through the ABI) links with plain gcc
    # against libfdjac + the HIP runtime's C entry points; without a GPU it reports FD_ERR_NODEVICE and runs nothing
    import subprocess
    import torch
    exe = str(tmp_path / "c_abi_clients")
    libdir = os.path.join(ROOT, "finitediff.jl_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_clients.c"), "-o", exe, "-L" + libdir, "-lfdjac",
                           "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the run itself is a -m gpu test")
    out = subprocess.run([exe, "all"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 7 and "no HIP device" in out.stderr      # FD_ERR_NODEVICE


def test_device_header_band_arithmetic_matches_a_brute_force_csc(tmp_path):
    # include/fdjac_device.h (plain C, also compiled for the host): fd_band_colptr / fd_band_dest against the band's CSC built
    # by enumeration, for random shapes (rectangular, asymmetric bandwidths), cyclic colourings with C >= w and column ranges
    import subprocess
    src = tmp_path / "band.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include "fdjac_device.h"
int main(void) {
    unsigned s = 12345u;
    #define RND(n) ((int)((s = s * 1664525u + 1013904223u) >> 8) % (n))
    for (int it = 0; it < 3000; ++it) {
        fd_band_store d = {0};
        d.l = RND(5); d.u = RND(5);
        const int w = d.l + d.u + 1;
        d.N = 1 + RND(60);
        d.M = d.N + RND(9) - 4; if (d.M < 1) d.M = 1;
        d.C = w + RND(3); d.shift = RND(d.C);
        /* every column must hold at least one row of the band */
        int ok = 1;
        for (long long j = 0; j < d.N; ++j) { long long f = j - d.u > 0 ? j - d.u : 0, l = j + d.l < d.M - 1 ? j + d.l : d.M - 1; if (l < f) ok = 0; }
        if (!ok) continue;
        long long cp[64], nn = 0;
        for (long long j = 0; j < d.N; ++j) { cp[j] = nn; long long f = j - d.u > 0 ? j - d.u : 0, l = j + d.l < d.M - 1 ? j + d.l : d.M - 1; nn += l - f + 1; }
        cp[d.N] = nn;
        for (long long j = 0; j <= d.N; ++j) if (fd_band_colptr(&d, j) != cp[j]) { printf("colptr %d\n", it); return 1; }
        d.col_begin = RND((int)d.N); d.col_end = d.col_begin + 1 + RND((int)(d.N - d.col_begin)); d.entry_begin = cp[d.col_begin];
        for (long long r = 0; r < d.M; ++r)
            for (int c = 0; c < d.C; ++c) {
                long long want = -1;
                for (long long j = r - d.l > 0 ? r - d.l : 0; j <= r + d.u && j < d.N; ++j)
                    if ((j + d.shift) % d.C == c && j >= d.col_begin && j < d.col_end) { long long f = j - d.u > 0 ? j - d.u : 0; want = cp[j] - d.entry_begin + (r - f); }
                if (fd_band_dest(&d, r, c) != want) { printf("dest it=%d r=%lld c=%d\n", it, r, c); return 2; }
            }
    }
    return 0;
}
''')
    exe = str(tmp_path / "band")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe])
    assert subprocess.run([exe], capture_output=True, text=True).returncode == 0


def test_device_header_stencil5_arithmetic_matches_a_brute_force_csc(tmp_path):
    # include/fdjac_device.h: fd_stencil5_colptr / fd_stencil5_colptr_ij (the form the kernels call: no 64-bit division) against
    # the 5-point stencil's CSC built by enumeration, for every grid shape up to 12 x 12 incl. one column / one row wide
    import subprocess
    src = tmp_path / "stencil.c"
    src.write_text(r'''
#include <stdio.h>
#include "fdjac_device.h"
int main(void) {
    for (long long nx = 1; nx <= 12; ++nx)
        for (long long ny = 1; ny <= 12; ++ny) {
            fd_stencil5_store d = {0};
            d.nx = nx; d.ny = ny;
            long long nn = 0;
            for (long long k = 0; k <= nx * ny; ++k) {
                const long long j = k / nx, i = k % nx;
                if (fd_stencil5_colptr(&d, k) != nn) { printf("colptr %lldx%lld k=%lld: %lld != %lld\n", nx, ny, k, fd_stencil5_colptr(&d, k), nn); return 1; }
                if (k < nx * ny && fd_stencil5_colptr_ij(&d, i, j) != nn) { printf("colptr_ij %lldx%lld k=%lld\n", nx, ny, k); return 2; }
                if (k < nx * ny) nn += 1 + (j > 0) + (i > 0) + (i < nx - 1) + (j < ny - 1);
            }
        }
    return 0;
}
''')
    exe = str(tmp_path / "stencil")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout


def test_user_examples_cross_compile_against_the_public_headers_only(tmp_path):
    # examples/user_f_store.hip and user_bb_store.hip (a USER's kernels that store the Jacobian themselves) build for gfx950 with
    # nothing but include/fdjac.h + include/fdjac_device.h on the include path and do not link libfdjac; their plain-C drivers
    # compile with gcc (hipcc cross-compiles without a GPU; the -m gpu tests run them)
    import subprocess
    inc = os.path.join(ROOT, "include")
    for name in ("user_f_store", "user_bb_store"):
        so = str(tmp_path / ("lib%s.so" % name))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-I" + inc,
                               os.path.join(ROOT, "examples", name + ".hip"), "-o", so])
        assert "libfdjac" not in subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    for name in ("user_store_client", "user_bb_client"):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + inc, "-c", os.path.join(ROOT, "examples", name + ".c"),
                               "-o", str(tmp_path / (name + ".o"))])


def _np_fingerprint(values, base=0):
    # F(a) = sum_k mix(v_2k + (v_2k+1 << 32) + k*K) mod 2^64, mix(z) = (z ^ z >> 29) * C, ^ >> 32; a missing last partner = 0x7fffffff
    # (csc/fdjac_match.hip), restated with numpy uint64 arithmetic
    with np.errstate(over="ignore"):
        v = (np.asarray(values, dtype=np.int64) - base).astype(np.uint64)
        if v.size % 2:
            v = np.concatenate([v, np.array([0x7fffffff], dtype=np.uint64)])
        v0, v1 = v[0::2], v[1::2]
        k = np.arange(v0.size, dtype=np.uint64)
        z = v0 + (v1 << np.uint64(32)) + k * np.uint64(0xD6E8FEB86659FD93)
        z = (z ^ (z >> np.uint64(29))) * np.uint64(0xBF58476D1CE4E5B9)
        z = z ^ (z >> np.uint64(32))
        return int(np.sum(z, dtype=np.uint64))


def test_pattern_fingerprint_host_path(L):
    # the host side of fd_plan_matches (content fingerprints of index arrays): deterministic, independent of the thread split and of
    # the index width / base, sensitive to a single in-place edit -- unlike a sampled hash (Julia's hash(::AbstractArray) reads
    # O(log n) elements of a long array)
    assert C.sizeof(fd.lib.PatternArrays) == 6 * 8 + 4 * 4
    fp3 = L.fdjac_fingerprint3
    fp3.restype = C.c_int
    vp = C.c_void_p

    def run(arrs, bytes_, i0, n, base):
        a = (vp * 3)(*[x.ctypes.data if x is not None else None for x in arrs])
        out = (C.c_uint64 * 3)()
        rc = fp3(None, a, (C.c_int * 3)(*bytes_), (C.c_int64 * 3)(*i0), (C.c_int64 * 3)(*n), (C.c_int64 * 3)(*base), 0, None, out)
        assert rc == 0
        return [int(v) for v in out]

    rng = np.random.default_rng(5)
    big = rng.integers(1, 1 << 40, size=(1 << 21) + 13, dtype=np.int64)        # long enough for the threaded split
    small = rng.integers(1, 100, size=1000, dtype=np.int64)
    h = run([big, small, None], [8, 8, 8], [0, 0, 0], [big.size, small.size, 0], [1, 0, 0])
    assert h[0] == _np_fingerprint(big, 1) and h[1] == _np_fingerprint(small) and h[2] == 0
    # the same values as Int32, 0-based: the same fingerprint
    s32 = (small - 1).astype(np.int32)
    assert run([s32, None, None], [4, 8, 8], [0, 0, 0], [s32.size, 0, 0], [0, 0, 0])[0] == _np_fingerprint(small, 1)
    # a window [i0, i0 + n): positions count from the window's start
    assert run([big, None, None], [8, 8, 8], [77, 0, 0], [5000, 0, 0], [0, 0, 0])[0] == _np_fingerprint(big[77:5077])
    # one edited element in the middle of a long array changes it; swapping two elements changes it
    e = big.copy(); e[big.size // 2 + 1] += 1
    assert run([e, None, None], [8, 8, 8], [0, 0, 0], [e.size, 0, 0], [1, 0, 0])[0] != h[0]
    w = small.copy(); w[[3, 500]] = w[[500, 3]]
    assert (w[3] != small[3]) and run([w, None, None], [8, 8, 8], [0, 0, 0], [w.size, 0, 0], [0, 0, 0])[0] != h[1]


def test_julia_shim_structure_and_foreign_calls():
    # No `julia` in the image: the shim (finitediff.jl_amd/julia/FiniteDiffMI355X.jl) cannot be run or even parsed here.  What CAN be
    # checked: its blocks / brackets balance (scripts/jl_balance.py), and every ccall names a symbol include/fdjac.h declares and the
    # library exports, and passes what the C prototype takes, parameter by parameter (pointer / 32-bit / 64-bit integer / double).
    import re
    import subprocess
    import sys
    shim = os.path.join(ROOT, "finitediff.jl_amd", "julia", "FiniteDiffMI355X.jl")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "jl_balance.py"), shim], capture_output=True, text=True)
    assert out.returncode == 0 and "depth () 0 [] 0, unclosed blocks: []" in out.stdout, out.stdout + out.stderr
    src = open(shim).read()
    hdr = open(os.path.join(ROOT, "include", "fdjac.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    fn_typedefs = set(re.findall(r"typedef\s+\w+\s*\(\s*\*\s*(\w+)\s*\)", hdr))

    def split_top(s):
        parts, depth, cur = [], 0, ""
        for ch in s:
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            if ch == "," and depth == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append(cur)
        return parts

    def c_class(t):           # what the parameter is at the ABI: a pointer, a 32-bit integer, a 64-bit integer, a double
        t = t.strip()
        if "*" in t or "[" in t:
            return "ptr"
        words = t.replace("const ", "").split()
        ty = " ".join(words[:-1]) if len(words) > 1 else words[0]
        if ty in fn_typedefs:
            return "ptr"
        return {"int": "i32", "unsigned": "i32", "unsigned int": "i32", "int32_t": "i32", "int64_t": "i64", "long long": "i64", "uint64_t": "i64",
                "unsigned long long": "i64", "size_t": "i64", "double": "f64", "float": "f32"}.get(ty, "?" + ty)

    def jl_class(t):
        t = t.strip()
        if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
            return "ptr"
        return {"Cint": "i32", "Int32": "i32", "Cuint": "i32", "UInt32": "i32", "Int64": "i64", "Clonglong": "i64", "UInt64": "i64", "Culonglong": "i64",
                "Csize_t": "i64", "Cdouble": "f64", "Float64": "f64", "Cfloat": "f32", "Float32": "f32"}.get(t, "?" + t)

    protos = {}
    for m in re.finditer(r"\b(?:int|const char \*|size_t|void)\s*(fd(?:32)?_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = [] if args in ("", "void") else [c_class(a) for a in split_top(args)]
    checked = 0
    for m in re.finditer(r"ccall\(\(\s*(:fd_[a-z0-9_]+|\$\(P \* \"([a-z0-9_]+)\"\))\s*,\s*libfdjac\)\s*,\s*\w+\s*,\s*\(", src):
        names = [m.group(1)[1:]] if m.group(2) is None else ["fd_" + m.group(2), "fd32_" + m.group(2)]
        # the argument-type tuple: from the "(" that ends the match to its matching ")"
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        jl = [jl_class(p) for p in split_top(src[i:j - 1]) if p.strip()]
        for nm in names:
            assert nm in protos, "the shim calls %s, which include/fdjac.h does not declare" % nm
            assert protos[nm] == jl, "%s: the shim passes %s, the header takes %s" % (nm, jl, protos[nm])
            checked += 1
    assert checked > 60, checked
    lib = os.path.join(ROOT, "finitediff.jl_amd", "lib", "libfdjac.so")
    if os.path.exists(lib):
        syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
        exported = set(re.findall(r" T (fd(?:32)?_[a-z0-9_]+)", syms))
        for m in re.finditer(r"ccall\(\(\s*:(fd_[a-z0-9_]+)", src):
            assert m.group(1) in exported, m.group(1)
