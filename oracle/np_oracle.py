"""Second, independent restatement of the reference's coloured-Jacobian loop, in plain numpy.

TEST INFRASTRUCTURE ONLY (see oracle/fd_oracle.c): it exists to cross-check the C oracle -- two restatements
written separately from the same reference lines must agree to rounding.  It follows
src/jacobians.jl:504-653 literally (mask, norm, epsilon, in-place perturbation, f!, difference, decompression,
un-perturbation, colour by colour) and returns a dense J with the pattern's entries filled.
"""
import numpy as np

EPS = np.finfo(np.float64).eps


def default_relstep(fdtype):   # src/epsilons.jl:133-144
    return {"forward": np.sqrt(EPS), "central": np.cbrt(EPS), "complex": 1.0}[fdtype]


def compute_epsilon(fdtype, x, relstep, absstep, dir=1.0):   # src/epsilons.jl:26-29, 50-53, 104-107
    if fdtype == "forward":
        return max(relstep * abs(x), absstep) * dir
    if fdtype == "central":
        return max(relstep * abs(x), absstep)
    return EPS


def jacobian(f, x, colorvec, pattern, fdtype="forward", relstep=None, absstep=None, dir=1.0, f_in=None):
    """f(fx, x) in place (numpy, real or complex); pattern: boolean M x N (structural non-zeros);
    returns (J dense M x N, number of f calls)."""
    x = np.array(x, dtype=np.float64)
    M, N = pattern.shape
    colorvec = np.asarray(colorvec)
    assert colorvec.size == N                                   # src/jacobians.jl:516
    relstep = default_relstep(fdtype) if relstep is None else relstep
    absstep = relstep if absstep is None else absstep
    J = np.zeros((M, N))                                        # fill_matrix!(J, false), :530-532
    ncalls = 0
    x1 = x.copy()                                               # copyto!(x1, x), :519
    if fdtype == "forward":
        fx = np.zeros(M)
        if f_in is None:
            f(fx, x)                                            # :540-545
            ncalls += 1
        else:
            fx[:] = f_in
        fx1 = np.zeros(M)
        for c in range(1, int(colorvec.max()) + 1):             # :547
            mask = (colorvec == c)
            x2 = x1 * mask                                      # :559
            eps = compute_epsilon("forward", np.sqrt(np.linalg.norm(x2)), relstep, absstep, dir)   # :560-561
            x1 = x1 + eps * mask                                # :562
            f(fx1, x1)
            ncalls += 1
            d = (fx1 - fx) / eps                                # :565
            for j in np.nonzero(mask)[0]:                       # decompression: J[row, col] = vfx[row]
                rows = np.nonzero(pattern[:, j])[0]
                J[rows, j] = d[rows]
            x1 = x1 - eps * mask                                # :584
    elif fdtype == "central":
        fx, fx1 = np.zeros(M), np.zeros(M)
        xx = x                                                   # the caller's x is perturbed too (:604, :620)
        for c in range(1, int(colorvec.max()) + 1):
            mask = (colorvec == c)
            x2 = x1 * mask
            eps = compute_epsilon("central", np.sqrt(np.linalg.norm(x2)), relstep, absstep)
            x1 = x1 + eps * mask
            xx = xx - eps * mask
            f(fx1, x1)
            f(fx, xx)
            ncalls += 2
            d = (fx1 - fx) / (2 * eps)                          # :607
            for j in np.nonzero(mask)[0]:
                rows = np.nonzero(pattern[:, j])[0]
                J[rows, j] = d[rows]
            x1 = x1 - eps * mask
            xx = xx + eps * mask
    else:
        eps = EPS                                                # :624
        cx1 = x1.astype(np.complex128)
        cfx = np.zeros(M, np.complex128)
        for c in range(1, int(colorvec.max()) + 1):
            mask = (colorvec == c)
            cx1 = cx1 + 1j * eps * mask                          # :633
            f(cfx, cx1)
            ncalls += 1
            d = cfx.imag / eps                                   # :635
            for j in np.nonzero(mask)[0]:
                rows = np.nonzero(pattern[:, j])[0]
                J[rows, j] = d[rows]
            cx1 = cx1 - 1j * eps * mask                          # :646
    return J, ncalls


def jacobian_complex_x(f, x, colorvec, pattern=None, fdtype="forward", relstep=None, absstep=None, dir=1.0, f_in=None):
    """The same loop for COMPLEX-valued x (returntype <: Complex with Val(:forward) / Val(:central)): the reference's code is
    generic in eltype(x) -- src/jacobians.jl:94-128 (cache), 537-622 (loop) -- so this is the loop above with complex arrays:
    the masked norm runs over complex elements (np.linalg.norm of a complex vector = sqrt(sum |x_j|^2)), epsilon is real
    (src/epsilons.jl:26-29, 50-53: abs of a complex number), `x1 .+= epsilon * mask` moves the real parts, the quotient is a
    complex vector over a real number.  pattern=None is the dense arm (sparsity === nothing, :548-557 / :590-598: column i
    perturbs x[i] alone with compute_epsilon(fdtype, x[i], ...)).  Known answer: test/finitedifftests.jl:480-513.
    Returns (J dense complex M x N, number of f calls)."""
    assert fdtype in ("forward", "central")                     # Val(:complex) with a complex returntype: fdtype_error (:106)
    x = np.array(x, dtype=np.complex128)
    N = x.size
    colorvec = np.asarray(colorvec)
    assert colorvec.size == N
    relstep = default_relstep(fdtype) if relstep is None else relstep
    absstep = relstep if absstep is None else absstep
    probe = np.zeros(0 if pattern is None else pattern.shape[0], np.complex128)
    M = N if pattern is None and f_in is None else (len(f_in) if pattern is None else pattern.shape[0])
    del probe
    J = np.zeros((M, N), np.complex128)
    ncalls = 0
    x1 = x.copy()
    fx, fx1 = np.zeros(M, np.complex128), np.zeros(M, np.complex128)
    if fdtype == "forward":
        if f_in is None:
            f(fx, x)
            ncalls += 1
        else:
            fx[:] = f_in
    xx = x.copy()
    for c in range(1, int(colorvec.max()) + 1):
        if pattern is None:                                      # dense arm: colour index == column index
            i = c - 1
            save = x1[i]
            eps = compute_epsilon(fdtype, save, relstep, absstep, dir)
            x1[i] = save + eps
            f(fx1, x1)
            ncalls += 1
            if fdtype == "central":
                x1[i] = save - eps
                f(fx, x1)
                ncalls += 1
                J[:, i] = (fx1 - fx) / (2 * eps)
            else:
                J[:, i] = (fx1 - fx) / eps
            x1[i] = save
            continue
        mask = (colorvec == c)
        x2 = x1 * mask
        eps = compute_epsilon(fdtype, np.sqrt(np.linalg.norm(x2)), relstep, absstep, dir)
        x1 = x1 + eps * mask
        if fdtype == "central":
            xx = xx - eps * mask
            f(fx1, x1)
            f(fx, xx)
            ncalls += 2
            d = (fx1 - fx) / (2 * eps)
        else:
            f(fx1, x1)
            ncalls += 1
            d = (fx1 - fx) / eps
        for j in np.nonzero(mask)[0]:
            rows = np.nonzero(pattern[:, j])[0]
            J[rows, j] = d[rows]
        x1 = x1 - eps * mask
        if fdtype == "central":
            xx = xx + eps * mask
    return J, ncalls
