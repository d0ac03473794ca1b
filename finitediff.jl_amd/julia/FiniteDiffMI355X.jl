# FiniteDiffMI355X.jl -- the Julia side of the drop-in boundary (include/fdjac.h).
#
# NOT RUNNABLE IN THE BUILD IMAGE (no `julia`); kept 1:1 with the executable ctypes mirror
# `finitediff.jl_amd/lib.py` + `api.py`, which the GPU parity tests drive.  A maintainer adds this
# file as a package extension (Project.toml `[extensions]`, next to FiniteDiffSparseArraysExt) or
# loads it as a stand-alone module.
#
# What it does: adds ONE method to `FiniteDiff.finite_difference_jacobian!` -- the cached in-place
# form of src/jacobians.jl:504-514 -- specialised on `f::DeviceF` (a device f! launcher), so that
#
#     cache = FiniteDiff.JacobianCache(x, Val(:forward); colorvec = colors, sparsity = Jproto)
#     FiniteDiff.finite_difference_jacobian!(J, DeviceF(:tridiag, N), x, cache)
#
# runs lines :515-:652 on the MI355X.  Everything else (kwargs, defaults, return value `nothing`,
# J's own storage being filled, x left untouched) is the reference's contract.  The method is more
# specific than the reference's only in `f`, so it introduces no ambiguity
# (test/finitedifftests.jl:4-5 `detect_ambiguities`).
module FiniteDiffMI355X

using FiniteDiff
using SparseArrays, LinearAlgebra

const libfdjac = get(ENV, "LIBFDJAC", joinpath(@__DIR__, "..", "lib", "libfdjac.so"))

const FD_HOST, FD_DEVICE = Cint(0), Cint(1)
fdtype_code(::Val{:forward}) = Cint(0)
fdtype_code(::Val{:central}) = Cint(1)
fdtype_code(::Val{:complex}) = Cint(2)

# struct fd_plan_opts (include/fdjac.h)
struct PlanOpts
    fdtype::Int32
    reserved0::Int32
    col_begin::Int64
    col_end::Int64
    x_begin::Int64
    x_end::Int64
    scratch_bytes::Int64
    color_begin::Int64      # owned colours [color_begin, color_end), 0-based; 0,0 = all (multi-GPU colour sharding)
    color_end::Int64
end
PlanOpts(fd) = PlanOpts(fdtype_code(fd), 0, 0, 0, 0, 0, 0, 0, 0)

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:fd_last_error, libfdjac), Cstring, ()))
    error("libfdjac error $rc: $msg")          # no exceptions cross the ccall; raise on the Julia side
end

mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer = 0; stream::Ptr{Cvoid} = C_NULL)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:fd_ctx_create, libfdjac), Cint, (Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), device, stream, r))
        c = new(r[])
        finalizer(c -> ccall((:fd_ctx_destroy, libfdjac), Cint, (Ptr{Cvoid},), c.h), c)
    end
end
const DEFAULT_CTX = Ref{Union{Nothing,Context}}(nothing)
default_ctx() = something(DEFAULT_CTX[], (DEFAULT_CTX[] = Context(); DEFAULT_CTX[]))

"""
A device `f!`: either one of libfdjac's built-in families or a user launcher
`@cfunction(launch, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Int64, Int64, Cint, Ptr{Cvoid}))`
that enqueues an AMDGPU.jl kernel on the given stream for `nbatch` points (fd_f_launch in fdjac.h).
"""
mutable struct DeviceF
    fn::Ptr{Cvoid}
    fctx::Ptr{Cvoid}
    builtin::Bool
end
const FAMILIES = Dict(:tridiag => 0, :tridiag_nl => 1, :lap5 => 2, :clamp5 => 3, :blockcoupled => 4, :nonsquare => 5)
function DeviceF(family::Symbol, params::Integer...; ctx = default_ctx())
    prm = Int64[params...]
    fn, fc = Ref{Ptr{Cvoid}}(C_NULL), Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:fd_builtin_f_create, libfdjac), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Int64}, Cint, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}),
                ctx.h, FAMILIES[family], prm, length(prm), fn, fc))
    f = DeviceF(fn[], fc[], true)
    finalizer(f -> ccall((:fd_builtin_f_destroy, libfdjac), Cint, (Ptr{Cvoid},), f.fctx), f)
end
DeviceF(launcher::Ptr{Cvoid}, fctx::Ptr{Cvoid} = C_NULL) = DeviceF(launcher, fctx, false)

mutable struct Plan
    h::Ptr{Cvoid}
    key::Any
end
destroy!(p::Plan) = ccall((:fd_plan_destroy, libfdjac), Cint, (Ptr{Cvoid},), p.h)

# plans live next to the cache they were compiled for (pattern / colours are stored by reference in
# the reference too, src/jacobians.jl:512-513; a changed colorvec/sparsity object => new plan)
const PLANS = WeakKeyDict{Any,Plan}()

colors64(colorvec, n) = (length(colorvec) == n || throw(DimensionMismatch("length(colorvec) != length(x)"));
                         collect(Int64, colorvec))

# ---- one constructor per `_colorediteration!` overload of the reference -----------------------
# ext/FiniteDiffSparseArraysExt.jl:38-47 (gate :51-52): J and sparsity share colptr/rowval
function make_plan(ctx, J::SparseMatrixCSC{Float64,Int64}, sparsity::SparseMatrixCSC, colorvec, fd)
    m, n = size(J)
    (J.colptr == sparsity.colptr && J.rowval == sparsity.rowval) ||
        return make_plan_entries(ctx, J, sparsity, colorvec, fd)
    cv, r = colors64(colorvec, n), Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:fd_plan_create_csc, libfdjac), Cint,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Cint, Cint, Ptr{Int64}, Cint, Ref{PlanOpts}, Ptr{Ptr{Cvoid}}),
                ctx.h, m, n, J.colptr, J.rowval, 8, 1, cv, 8, PlanOpts(fd), r))
    Plan(r[], (sparsity, colorvec))
end
# ext/FiniteDiffSparseArraysExt.jl:20-28 with a dense J
function make_plan(ctx, J::Matrix{Float64}, sparsity::SparseMatrixCSC, colorvec, fd)
    m, n = size(J)
    cv, r = colors64(colorvec, n), Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:fd_plan_create_csc_dense, libfdjac), Cint,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Cint, Cint, Ptr{Int64}, Cint, Ref{PlanOpts}, Ptr{Ptr{Cvoid}}),
                ctx.h, m, n, sparsity.colptr, sparsity.rowval, 8, 1, cv, 8, PlanOpts(fd), r))
    Plan(r[], (sparsity, colorvec))
end
# src/iteration_utils.jl:25-32 with a dense-matrix pattern (src/jacobians.jl:473-488)
function make_plan(ctx, J::Matrix{Float64}, sparsity::DenseMatrix, colorvec, fd)
    m, n = size(J)
    rows, cols = FiniteDiff._findstructralnz(sparsity)
    cv, r = colors64(colorvec, n), Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:fd_plan_create_coo_dense, libfdjac), Cint,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Int64, Cint, Cint, Ptr{Int64}, Cint, Ref{PlanOpts}, Ptr{Ptr{Cvoid}}),
                ctx.h, m, n, rows, cols, length(rows), 8, 1, cv, 8, PlanOpts(fd), r))
    Plan(r[], (sparsity, colorvec))
end
# src/iteration_utils.jl:25-32 through Tridiagonal's setindex!
function make_plan(ctx, J::Tridiagonal{Float64}, sparsity, colorvec, fd)
    n = size(J, 1)
    cv, r = colors64(colorvec, n), Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:fd_plan_create_tridiagonal, libfdjac), Cint,
                (Ptr{Cvoid}, Int64, Ptr{Int64}, Cint, Ref{PlanOpts}, Ptr{Ptr{Cvoid}}), ctx.h, n, cv, 8, PlanOpts(fd), r))
    Plan(r[], (sparsity, colorvec))
end
# any other storage: enumerate (row, col, position in J's value vector) on the host once
function make_plan_entries(ctx, J::SparseMatrixCSC{Float64,Int64}, sparsity::SparseMatrixCSC, colorvec, fd)
    m, n = size(J)
    rows, cols, _ = findnz(sparsity)
    dest = Int64[]
    for (r, c) in zip(rows, cols)
        rng = nzrange(J, c)
        k = searchsortedfirst(view(J.rowval, rng), r)
        (k <= length(rng) && J.rowval[rng[k]] == r) || throw(ArgumentError("J has no stored entry ($r,$c)"))
        push!(dest, rng[k] - 1)
    end
    cv, r = colors64(colorvec, n), Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:fd_plan_create_entries, libfdjac), Cint,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Int64, Int64, Cint, Cint, Ptr{Int64}, Cint, Ref{PlanOpts}, Ptr{Ptr{Cvoid}}),
                ctx.h, m, n, rows, cols, dest, length(dest), nnz(J), 8, 1, cv, 8, PlanOpts(fd), r))
    Plan(r[], (sparsity, colorvec))
end
# BandedMatrices / BlockBandedMatrices methods go in the corresponding extension modules:
#   BandedMatrix      -> fd_plan_create_banded(ctx, m, n, l, u, colorvec, ...);  outs = [bandeddata(J)]
#   BlockBandedMatrix -> fd_plan_create_blockbanded(ctx, nblk, blocklengths(axes(J,1)), l, u,
#                            bandeddata(J.block_sizes.block_starts), J.block_sizes.block_strides, ...); outs = [J.data]
#   BandedBlockBandedMatrix -> fd_plan_create_entries with (row, col, offset) enumerated once.

# Built-in families come with a lazy-point launcher (f! perturbs while loading, fd_f_launch_lazy) that also writes only
# imag(f) in the complex-step arm (FD_LAZY_CAP_IMAG_ONLY).  A user f! written against a lazy AbstractVector wrapper
# registers its own launcher the same way.
function install_lazy!(plan::Plan, f::DeviceF)
    fn, caps = Ref{Ptr{Cvoid}}(C_NULL), Ref{Cint}(0)
    ccall((:fd_builtin_f_lazy, libfdjac), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), f.fctx, fn) == 0 || return
    check(ccall((:fd_plan_set_lazy_f, libfdjac), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), plan.h, fn[]))
    ccall((:fd_builtin_f_lazy_caps, libfdjac), Cint, (Ptr{Cvoid}, Ptr{Cint}), f.fctx, caps)
    check(ccall((:fd_plan_set_lazy_caps, libfdjac), Cint, (Ptr{Cvoid}, Cint), plan.h, caps[]))
end

outs(J::SparseMatrixCSC) = (pointer(J.nzval), C_NULL, C_NULL)
outs(J::Matrix) = (pointer(J), C_NULL, C_NULL)
outs(J::Tridiagonal) = (pointer(J.dl), pointer(J.d), pointer(J.du))

# ---- the drop-in method (src/jacobians.jl:504-514) ---------------------------------------------
function FiniteDiff.finite_difference_jacobian!(
        J, f::DeviceF, x::Vector{Float64},
        cache::FiniteDiff.JacobianCache{T1, T2, T3, T4, cType, sType, fdtype, returntype},
        f_in = nothing;
        relstep = FiniteDiff.default_relstep(fdtype, eltype(x)),
        absstep = relstep,
        colorvec = cache.colorvec,
        sparsity = cache.sparsity,
        dir = true) where {T1, T2, T3, T4, cType, sType, fdtype, returntype}
    (fdtype == Val(:complex) && !(returntype <: Real)) && FiniteDiff.fdtype_error(returntype)
    sparsity === nothing && (sparsity = J)   # structured J is its own pattern (src/jacobians.jl:455)
    ctx = default_ctx()
    plan = get(PLANS, cache, nothing)
    if plan === nothing || plan.key[1] !== sparsity || plan.key[2] !== colorvec
        plan === nothing || destroy!(plan)
        plan = make_plan(ctx, J, sparsity, colorvec, fdtype)
        PLANS[cache] = plan
    end
    f.builtin && install_lazy!(plan, f)
    o = outs(J)
    optr = Ptr{Cvoid}[o...]
    fin = (f_in === nothing || fdtype != Val(:forward)) ? C_NULL : pointer(f_in)
    GC.@preserve J x f_in optr begin
        check(ccall((:fd_jacobian, libfdjac), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cint, Ptr{Float64}, Cint,
                     Float64, Float64, Float64, Ptr{Ptr{Cvoid}}, Cint),
                    plan.h, f.fn, f.fctx, x, FD_HOST, fin, FD_HOST,
                    Float64(relstep), Float64(absstep), Float64(dir), optr, FD_HOST))
    end
    nothing
end
# With AMDGPU.jl, a method for `x::ROCVector{Float64}` / `J.nzval::ROCVector` passes device pointers
# with FD_DEVICE (zero-copy) -- identical call, `pointer(x)` and kind flags change.

# Float32 problems (`x::Vector{Float32}`, `J` with Float32 values): the same methods with every `:fd_…` symbol that
# touches values replaced by `:fd32_…` (include/fdjac.h, "Float32 instantiation"); a thin macro over `eltype(x)`
# generates both sets of `ccall`s.

# ---- finite_difference_jvp! (src/jvp.jl:238-274) ------------------------------------------------
const JVP_PLANS = WeakKeyDict{Any,Ptr{Cvoid}}()
function FiniteDiff.finite_difference_jvp!(
        jvp::Vector{Float64}, f::DeviceF, x::Vector{Float64}, v::Vector{Float64},
        cache::FiniteDiff.JVPCache{X1, FX1, fdtype}, f_in = nothing;
        relstep = FiniteDiff.default_relstep(fdtype, eltype(x)), absstep = relstep, dir = true) where {X1, FX1, fdtype}
    fdtype == Val(:complex) && error("finite_difference_jvp doesn't support :complex-mode finite diff")   # src/jvp.jl:248-250
    ctx = default_ctx()
    h = get!(JVP_PLANS, cache) do
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:fd_jvp_plan_create, libfdjac), Cint, (Ptr{Cvoid}, Int64, Int64, Cint, Ptr{Ptr{Cvoid}}),
                    ctx.h, length(jvp), length(x), fdtype_code(fdtype), r))
        r[]
    end
    if f.builtin   # lazy-point JVP launcher (fd_f_launch_lazy_jvp): f! forms x + eps*v while loading, no points pass
        lz = Ref{Ptr{Cvoid}}(C_NULL)
        ccall((:fd_builtin_f_lazy_jvp, libfdjac), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), f.fctx, lz) == 0 &&
            check(ccall((:fd_jvp_plan_set_lazy_f, libfdjac), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h, lz[]))
    end
    fin = (f_in === nothing || fdtype != Val(:forward)) ? C_NULL : pointer(f_in)
    GC.@preserve jvp x v f_in begin
        check(ccall((:fd_jvp, libfdjac), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Cint,
                     Float64, Float64, Float64, Ptr{Float64}, Cint),
                    h, f.fn, f.fctx, x, v, FD_HOST, fin, FD_HOST, Float64(relstep), Float64(absstep), Float64(dir),
                    jvp, FD_HOST))
    end
    nothing
end

# ---- multi-GPU (one Julia process per GPU, MPI.jl / RCCL for the exchange) ------------------------
# column ranges : PlanOpts(col_begin, col_end, x_begin, x_end) -> each rank fills a contiguous slice of nzval; one
#                 all-gather assembles it (needs an f! that honours the row window it is handed).
# colour ranges : PlanOpts(color_begin, color_end) -> each rank perturbs / evaluates only its colours with ANY f!;
#                 outputs start from zero and one all-reduce(SUM) assembles them (the north-star's colour sharding).

end # module
