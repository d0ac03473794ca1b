"""Host-side mirror of the reference interface for the coloured-Jacobian path.

Julia is not available in the build image, so the host layer that a Julia shim would provide
(``finitediff.jl_amd/julia/FiniteDiffMI355X.jl``) is mirrored here in Python with the reference's
names and argument meaning:

* ``JacobianCache``                   -- src/jacobians.jl:1-128
* ``finite_difference_jacobian_b``    -- ``finite_difference_jacobian!`` (PyJulia spelling of ``!``),
                                         cache-less src/jacobians.jl:446-471 and cached :504-653
* ``SparseMatrixCSC / Tridiagonal / BandedMatrix / BlockBandedMatrix`` -- thin storage holders with
  Julia's field names and 1-based Int64 index arrays; dispatch on (J, sparsity) follows the
  reference's ``_colorediteration!`` overloads (ext/*.jl).

All arithmetic happens in libfdjac's HIP kernels; arrays are torch CUDA tensors (device
pointers are handed over as-is) or numpy arrays (staged through the C ABI's host path).
"""
import ctypes as C
import threading
import weakref

import numpy as np

from . import lib as _l

_EPS = float(np.finfo(np.float64).eps)


def default_relstep(fdtype, T=np.float64):
    """src/epsilons.jl:133-144"""
    fdtype = _norm_fdtype(fdtype)
    eps = np.finfo(np.dtype(T)).eps              # eps(eltype(x)): Float32 problems take sqrt / cbrt of eps(Float32)
    if fdtype == "forward":
        return float(np.sqrt(eps))
    if fdtype == "central":
        return float(np.cbrt(eps))
    return 1.0


def _norm_fdtype(fdtype):
    """Accepts "forward" / ":forward" / "Val{:forward}" / "Val(:forward)"."""
    s = str(fdtype).replace("Val", "").strip("{}():, ").lower()
    if s not in _l.FDTYPES:
        # fdtype_error, src/epsilons.jl:159-167
        raise ValueError("Unrecognized fdtype: valid values are Val{:forward}, Val{:central} and Val{:complex}.")
    return s


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _dtype_of(a):
    """numpy dtype (float64 / float32) of a torch tensor or numpy array; None for anything else."""
    if _is_torch(a):
        import torch
        return {torch.float64: np.dtype(np.float64), torch.float32: np.dtype(np.float32)}.get(a.dtype)
    if isinstance(a, np.ndarray) and a.dtype in (np.float64, np.float32):
        return a.dtype
    return None


def _complex_base(a):
    """float64 / float32 for a complex128 / complex64 torch tensor or numpy array (Complex{T} in memory = (re, im) pairs of T);
    None for anything else."""
    if _is_torch(a):
        import torch
        return {torch.complex128: np.dtype(np.float64), torch.complex64: np.dtype(np.float32)}.get(a.dtype)
    if isinstance(a, np.ndarray) and a.dtype in (np.complex128, np.complex64):
        return np.dtype(np.float64) if a.dtype == np.complex128 else np.dtype(np.float32)
    return None


def _ptr(a, what, dtype=np.float64, cx=False):
    """(pointer, FD_HOST|FD_DEVICE, keepalive) of a vector-like of the plan's element type (cx: of complex numbers over it)."""
    dtype = np.dtype(dtype)
    if cx:
        if _complex_base(a) != dtype:
            raise TypeError("%s must be a complex numpy array or torch tensor over %s" % (what, dtype.name))
        if _is_torch(a):
            if a.is_cuda:
                if not _colmajor_contig(a):
                    raise ValueError("%s must be contiguous (column-major for matrices)" % what)
                return a.data_ptr(), _l.DEVICE, a
            a = a.numpy()
        if not (a.flags.f_contiguous or a.ndim <= 1 and a.flags.c_contiguous):
            raise ValueError("%s must be contiguous (column-major for matrices)" % what)
        return a.ctypes.data, _l.HOST, a
    if _dtype_of(a) != dtype:
        raise TypeError("%s must be a %s numpy array or torch tensor" % (what, dtype.name))
    if _is_torch(a):
        if a.is_cuda:
            if not _colmajor_contig(a):
                raise ValueError("%s must be contiguous (column-major for matrices)" % what)
            return a.data_ptr(), _l.DEVICE, a
        a = a.numpy()
    if not isinstance(a, np.ndarray) or a.dtype != dtype:
        raise TypeError("%s must be a %s numpy array or torch tensor" % (what, dtype.name))
    if not (a.flags.f_contiguous or a.ndim <= 1 and a.flags.c_contiguous):
        raise ValueError("%s must be contiguous (column-major for matrices)" % what)
    return a.ctypes.data, _l.HOST, a


def _colmajor_contig(t):
    if t.dim() <= 1:
        return t.is_contiguous()
    return t.t().is_contiguous()


def _i64(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int64)


# ----------------------------------------------------------------------------------------------
# matrix holders (Julia field names)
# ----------------------------------------------------------------------------------------------
class SparseMatrixCSC:
    """SparseArrays.SparseMatrixCSC{Float64,Int64}: m, n, colptr, rowval (1-based), nzval."""

    def __init__(self, m, n, colptr, rowval, nzval=None):
        self.m, self.n = int(m), int(n)
        self.colptr, self.rowval = _i64(colptr), _i64(rowval)
        self.nzval = nzval  # None for a pure pattern (a `sparsity` argument)

    def size(self):
        return self.m, self.n

    def similar(self, like=None):
        return SparseMatrixCSC(self.m, self.n, self.colptr, self.rowval, _similar(self.nzval if like is None else like,
                                                                                   self.rowval.size))


class DevicePatternCSC:
    """A CSC Jacobian whose PATTERN lives on the device as well (the shim's `DevicePatternCSC`: colPtr / rowVal of a
    ROCSparseMatrixCSC; int32 or int64 torch CUDA tensors, `idx_base`-based) -- the plan is compiled by kernels
    (fd_plan_create_csc_device), nothing crosses PCIe; `colorvec` must be a CUDA tensor too.  It is its own sparsity pattern."""

    def __init__(self, m, n, colptr, rowval, nzval=None, idx_base=1):
        self.m, self.n, self.colptr, self.rowval, self.nzval, self.idx_base = int(m), int(n), colptr, rowval, nzval, int(idx_base)

    def size(self):
        return (self.m, self.n)


class _Edit:
    """`with holder.edit(): ...` -- the arrays are writable inside, the holder's generation is bumped on the way out."""

    def __init__(self, owner, arrays):
        self.owner, self.arrays = owner, arrays

    def __enter__(self):
        for a in self.arrays:
            a.flags.writeable = True
        return self.arrays[0] if len(self.arrays) == 1 else self.arrays

    def __exit__(self, *exc):
        for a in self.arrays:
            a.flags.writeable = False
        self.owner.generation += 1
        return False


class TrackedCSC(SparseMatrixCSC):
    """A host SparseMatrixCSC pattern whose EDITS are visible in O(1): colptr / rowval are read-only arrays, changed only inside
    `with A.edit() as (colptr, rowval): ...`, which bumps `A.generation`.  A cached call (JacobianCache, pattern_check "auto" /
    "content") compares generations instead of re-hashing 400 MB of Int64 indices per call (3 ms at N = 10^7: 42 x the Jacobian) --
    the reference re-reads the pattern on every call (src/jacobians.jl:512-513) because it has no way to know; a wrapper that owns
    the mutation does.  The shim's `TrackedCSC` is the same thing for Julia callers."""

    def __init__(self, m, n, colptr, rowval, nzval=None):
        super().__init__(m, n, np.array(colptr, dtype=np.int64), np.array(rowval, dtype=np.int64), nzval)
        self.generation = 0
        self.colptr.flags.writeable = False
        self.rowval.flags.writeable = False

    def edit(self):
        return _Edit(self, [self.colptr, self.rowval])


class TrackedVector:
    """The same for a host colour vector: `data` is read-only outside `with v.edit() as a: ...`."""

    def __init__(self, data):
        self.data = np.array(data, dtype=np.int64)
        self.data.flags.writeable = False
        self.generation = 0

    def edit(self):
        return _Edit(self, [self.data])

    def __array__(self, dtype=None, copy=None):
        return self.data if dtype is None else self.data.astype(dtype)

    def __len__(self):
        return int(self.data.size)

    @property
    def shape(self):
        return self.data.shape

    @property
    def size(self):
        return self.data.size

    def max(self):
        return self.data.max()


class Tridiagonal:
    """LinearAlgebra.Tridiagonal: dl (n-1), d (n), du (n-1)."""

    def __init__(self, dl, d, du):
        self.dl, self.d, self.du = dl, d, du

    def size(self):
        n = int(self.d.shape[0])
        return n, n


class _ThreeDiagonals(Tridiagonal):
    """LinearAlgebra's other structured types the reference's generic loop serves (`src/jacobians.jl:524-525`,
    `src/iteration_utils.jl:25-32`: `J[rows_index[i], cols_index[i]] = vfx[rows_index[i]]` over `findstructralnz(J)`): compiled as the
    Tridiagonal plan -- the diagonals the type does not store are computed into scratch and dropped."""

    def _scratch(self, like, n):
        key = (n, str(getattr(like, "dtype", None)), str(getattr(like, "device", "")))
        cache = self.__dict__.setdefault("_sc", {})
        if key not in cache:
            cache[key] = _similar(like, n)
        return cache[key]

    def _finish(self, colorvec):
        pass


class Bidiagonal(_ThreeDiagonals):
    """LinearAlgebra.Bidiagonal(dv, ev, uplo): J[i,i] -> dv[i]; uplo = 'U': J[i,i+1] -> ev[i], 'L': J[i+1,i] -> ev[i]."""

    def __init__(self, dv, ev, uplo="U"):
        self.dv, self.ev, self.uplo = dv, ev, uplo.upper()[0]
        self.d = dv
        n = int(dv.shape[0])
        sc = self._scratch(dv, max(n - 1, 1))[: n - 1]
        self.dl, self.du = (sc, ev) if self.uplo == "U" else (ev, sc)


class Diagonal(_ThreeDiagonals):
    """LinearAlgebra.Diagonal(diag)."""

    def __init__(self, diag):
        self.diag = self.d = diag
        n = int(diag.shape[0])
        self.dl = self._scratch(diag, max(n - 1, 1))[: n - 1]
        self.du = _similar(diag, max(n - 1, 1))[: n - 1]


class SymTridiagonal(_ThreeDiagonals):
    """LinearAlgebra.SymTridiagonal(dv, ev) -- an EXTENSION, not reference behaviour: `setindex!(::SymTridiagonal, v, i, j)` throws for
    i != j, so the reference's generic loop (src/iteration_utils.jl:25-32) cannot fill such a J at all.  Here the two off-diagonal
    quotients J[i+1,i] and J[i,i+1] are both taken as assignments to ev[i] in the loop's order (colour by colour): ev[i] ends up with the
    LATER colour's -- the upper entry (column i+1) if colorvec[i+1] > colorvec[i], else the lower one (`_finish`).  For a symmetric
    Jacobian the two agree up to the error of the difference quotient."""

    def __init__(self, dv, ev):
        self.dv, self.ev = dv, ev
        self.d = dv
        n = int(dv.shape[0])
        self.dl = self._scratch(dv, max(n - 1, 1))[: n - 1]
        self.du = _similar(dv, max(n - 1, 1))[: n - 1]

    def _finish(self, colorvec):
        cv = colorvec
        if _is_torch(self.ev):
            import torch
            c = cv if _is_torch(cv) else torch.as_tensor(np.asarray(cv), device=self.ev.device)
            c = c.to(self.ev.device)
            self.ev.copy_(torch.where(c[1:] > c[:-1], self.du, self.dl))
        else:
            c = np.asarray(cv)
            self.ev[...] = np.where(c[1:] > c[:-1], self.du, self.dl)


class BandedMatrix:
    """BandedMatrices.BandedMatrix: data is (l+u+1) x n column-major, data[u+i-j, j] = A[i,j] (0-based)."""

    def __init__(self, data, m, l, u):
        self.data, self.m, self.l, self.u = data, int(m), int(l), int(u)

    def size(self):
        return self.m, int(self.data.shape[1]) if hasattr(self.data, "shape") and len(self.data.shape) == 2 else None


class BlockBandedMatrix:
    """BlockBandedMatrices.BlockBandedMatrix: flat data + (blk_sizes, bl, bu, block_starts, block_strides)
    as produced by ``patterns.BlockBandedLayout`` (the BlockSkylineSizes layout)."""

    def __init__(self, data, layout):
        self.data, self.layout = data, layout

    def size(self):
        return self.layout.N, self.layout.N


class BandedBlockBandedMatrix:
    """BlockBandedMatrices.BandedBlockBandedMatrix: flat data + a ``patterns.BandedBlockBandedLayout``.  The reference
    stores through raw offsets into each block's banded data (ext/FiniteDiffBlockBandedMatricesExt.jl:16-42); the plan is
    structural (fd_plan_create_bandedblockbanded: block sizes, bandwidths, slab starts and strides -- no entry list).
    ``as_entries=True`` keeps round 3's form: the (row, column, offset) triples enumerated once and compiled with
    fd_plan_create_entries (what complex-valued x and column windows still use; the tests compare the two)."""

    def __init__(self, data, layout, as_entries=False):
        self.data, self.layout, self.as_entries = data, layout, as_entries

    def size(self):
        return self.layout.N, self.layout.N


def _similar(a, n):
    if _is_torch(a):
        import torch
        return torch.zeros(n, dtype=a.dtype, device=a.device)
    return np.zeros(n, dtype=getattr(a, "dtype", np.float64))


# ----------------------------------------------------------------------------------------------
# context / f launchers / plans
# ----------------------------------------------------------------------------------------------
class Context:
    """fd_ctx: a device + the stream every launch is enqueued on: torch's current stream by default (so torch ops,
    NCCL collectives and torch events stay ordered with the library's kernels without extra synchronisation);
    stream=None creates a private non-blocking stream; an integer is taken as a hipStream_t."""

    _default = {}
    _lock = threading.Lock()

    def __init__(self, device=0, stream="torch"):
        L = _l.load()
        self.L = L
        h = C.c_void_p()
        sp = None
        if stream == "torch":
            import torch
            if not torch.cuda.is_available():
                raise RuntimeError("libfdjac needs an MI355X (no HIP device visible); there is no CPU fallback")
            with torch.cuda.device(device):
                sp = torch.cuda.current_stream().cuda_stream or 1   # 0 = torch's default stream -> FD_STREAM_DEFAULT
        elif stream is not None:
            sp = int(stream)
        _l.check(L.fd_ctx_create(int(device), C.c_void_p(sp) if sp else None, C.byref(h)))
        self.handle, self.device = h, int(device)
        self._fin = weakref.finalize(self, L.fd_ctx_destroy, h)

    @classmethod
    def default(cls, device=0):
        with cls._lock:
            if device not in cls._default:
                cls._default[device] = cls(device)
            return cls._default[device]

    @property
    def stream(self):
        return self.L.fd_ctx_stream(self.handle)

    def synchronize(self):
        _l.check(self.L.fd_ctx_synchronize(self.handle))

    def stream_copy_gbps(self, nbytes=1 << 30, iters=10):
        out = C.c_double()
        _l.check(self.L.fd_stream_copy_gbps(self.handle, int(nbytes), int(iters), C.byref(out)))
        return out.value


class Comm:
    """fd_comm: the RCCL communicator behind the C ABI (one process per GPU).  Every collective is enqueued on the
    context's stream.  Bootstrap: rank 0's ``Comm.unique_id()`` bytes reach the other ranks by whatever the host has;
    ``Comm.from_torch_distributed`` uses an initialised torch.distributed group for exactly that and nothing else."""

    def __init__(self, ctx, nranks, rank, uid):
        self.ctx, self.L = ctx, ctx.L
        if len(uid) != _l.COMM_ID_BYTES:
            raise ValueError("the communicator id has %d bytes" % _l.COMM_ID_BYTES)
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(uid), _l.COMM_ID_BYTES)
        _l.check(self.L.fd_comm_create(ctx.handle, int(nranks), int(rank), buf, C.byref(h)))
        self.handle, self.nranks, self.rank = h, int(nranks), int(rank)
        self._fin = weakref.finalize(self, self.L.fd_comm_destroy, h)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(_l.COMM_ID_BYTES)
        _l.check(_l.load().fd_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, ctx, dist=None, group=None):
        if dist is None:
            import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(ctx, world, rank, box[0])

    def info(self):
        n, r, v = C.c_int(), C.c_int(), C.c_int()
        _l.check(self.L.fd_comm_info(self.handle, C.byref(n), C.byref(r), C.byref(v)))
        return {"nranks": n.value, "rank": r.value, "rccl_version": v.value,
                "library": self.L.fd_comm_library().decode("utf-8", "replace")}

    @staticmethod
    def _dev(t, what):
        if not (_is_torch(t) and t.is_cuda and t.is_contiguous()):
            raise ValueError("%s must be a contiguous CUDA tensor" % what)
        return t.data_ptr(), t.element_size()

    def allgather(self, buf, slot_elems):
        """In place: rank r's slice already sits in slot r of `buf` (nranks slots of slot_elems elements)."""
        p, eb = self._dev(buf, "buf")
        if buf.numel() < self.nranks * int(slot_elems):
            raise ValueError("buf is shorter than nranks * slot_elems")
        _l.check(self.L.fd_comm_allgather(self.handle, p, int(slot_elems), eb))

    def gatherv(self, send, recv, counts, root=0):
        """Slices of lengths counts[r] to the root, back to back in `recv` (only read on the root)."""
        counts = _i64(counts)
        displs = _i64(np.concatenate([[0], np.cumsum(counts)[:-1]]))
        sp, eb = self._dev(send, "send") if send.numel() else (None, send.element_size())
        rp = self._dev(recv, "recv")[0] if recv is not None else None
        _l.check(self.L.fd_comm_gatherv(self.handle, sp, int(send.numel()), rp, counts.ctypes.data_as(C.POINTER(C.c_int64)),
                                        displs.ctypes.data_as(C.POINTER(C.c_int64)), eb, int(root)))

    def allreduce_sum(self, buf):
        p, eb = self._dev(buf, "buf")
        _l.check(self.L.fd_comm_allreduce_sum(self.handle, p, buf.numel(), eb))

    def broadcast(self, buf, root=0):
        p, eb = self._dev(buf, "buf")
        _l.check(self.L.fd_comm_broadcast(self.handle, p, buf.numel(), eb, int(root)))

    def halo_exchange(self, buf, own_begin, own_end, halo):
        """fd_comm_halo_exchange: `buf` is a device vector in global indexing of which this rank owns [own_begin, own_end);
        the `halo` values on either side of the range arrive from the neighbouring ranks (one group of <= 4 transfers)."""
        p, eb = self._dev(buf, "buf")
        _l.check(self.L.fd_comm_halo_exchange(self.handle, p, int(own_begin), int(own_end), int(halo), eb))


    def enable_p2p(self, slot_bytes=1 << 16):
        """fd_comm_enable_p2p: map every rank's mailbox (handles exchanged over RCCL) and route this communicator's small
        messages -- all-gathers / halo exchanges up to `slot_bytes`, hence the sharded step-size reduction and the sharded
        solve -- through direct peer-to-peer stores.  Collective; raises if the peers cannot be mapped."""
        _l.check(self.L.fd_comm_enable_p2p(self.handle, int(slot_bytes)))

    def disable_p2p(self):
        """fd_comm_disable_p2p: back to RCCL for the small messages (every rank calls it)."""
        _l.check(self.L.fd_comm_disable_p2p(self.handle))

    def has_p2p(self):
        en = C.c_int()
        _l.check(self.L.fd_comm_p2p_status(self.handle, C.byref(en), None))
        return bool(en.value)

    def p2p_status(self):
        """0, or 1 + r once a mailbox wait for rank r timed out (fd_comm_p2p_status)."""
        v = C.c_int()
        _l.check(self.L.fd_comm_p2p_status(self.handle, None, C.byref(v)))
        return v.value


class P2P:
    """fd_p2p: small-message exchange by direct stores into the peers' HBM (one node, one process per GPU) -- the mailbox on
    its own, bootstrapped by the host (``from_torch_distributed`` ships the 64-byte handles through torch.distributed and
    nothing else).  Exchanges are enqueued on the context's stream; ``status()`` tells whether a wait ever timed out."""

    def __init__(self, ctx, nranks, rank, slot_bytes=1 << 16):
        self.ctx, self.L = ctx, ctx.L
        h = C.c_void_p()
        _l.check(self.L.fd_p2p_create(ctx.handle, int(nranks), int(rank), int(slot_bytes), C.byref(h)))
        self.handle, self.nranks, self.rank = h, int(nranks), int(rank)
        self._fin = weakref.finalize(self, self.L.fd_p2p_destroy, h)

    @classmethod
    def loopback(cls, ctx, nranks, rank, slot_bytes=1 << 16):
        """fd_p2p_create_loopback: rank `rank` of an `nranks`-rank job alone on one device (peers = a local sink, no wait spins);
        ``fill(sender, offset, tensor)`` puts what `sender` would have delivered into its slot."""
        self = cls.__new__(cls)
        self.ctx, self.L = ctx, ctx.L
        h = C.c_void_p()
        _l.check(self.L.fd_p2p_create_loopback(ctx.handle, int(nranks), int(rank), int(slot_bytes), C.byref(h)))
        self.handle, self.nranks, self.rank = h, int(nranks), int(rank)
        self._fin = weakref.finalize(self, self.L.fd_p2p_destroy, h)
        return self

    def fill(self, sender, offset, data):
        p, eb = Comm._dev(data, "data")
        _l.check(self.L.fd_p2p_loopback_fill(self.handle, int(sender), int(offset), p, int(data.numel()) * eb))

    def fill_fused(self, gsum64x8, halo_lo=None, halo_hi=None):
        """fd_p2p_loopback_fill_fused: the 64 x 8 group sums (float64 CUDA tensor of 512) and the neighbours' halos for the fused step."""
        p, _ = Comm._dev(gsum64x8, "gsum64x8")
        if gsum64x8.numel() != 512 or gsum64x8.element_size() != 8:
            raise ValueError("gsum64x8 must hold 64 x 8 doubles")
        lo = Comm._dev(halo_lo, "halo_lo") if halo_lo is not None else (None, 0)
        hi = Comm._dev(halo_hi, "halo_hi") if halo_hi is not None else (None, 0)
        hb = max(halo_lo.numel() * lo[1] if halo_lo is not None else 0, halo_hi.numel() * hi[1] if halo_hi is not None else 0)
        _l.check(self.L.fd_p2p_loopback_fill_fused(self.handle, p, lo[0], hi[0], int(hb)))

    def local_handle(self):
        buf = C.create_string_buffer(_l.P2P_HANDLE_BYTES)
        _l.check(self.L.fd_p2p_local_handle(self.handle, buf))
        return buf.raw

    def connect(self, handles):
        blob = b"".join(handles)
        if len(blob) != self.nranks * _l.P2P_HANDLE_BYTES:
            raise ValueError("need one %d-byte handle per rank" % _l.P2P_HANDLE_BYTES)
        _l.check(self.L.fd_p2p_connect(self.handle, C.create_string_buffer(blob, len(blob))))

    @classmethod
    def from_torch_distributed(cls, ctx, dist=None, group=None, slot_bytes=1 << 16):
        if dist is None:
            import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        self = cls(ctx, world, rank, slot_bytes)
        handles = [None] * world
        dist.all_gather_object(handles, self.local_handle(), group=group)
        self.connect(handles)
        dist.barrier(group=group)          # every rank has mapped every mailbox before anybody stores into one
        return self

    def info(self):
        n, r, sb, u = C.c_int(), C.c_int(), C.c_int64(), C.c_int()
        _l.check(self.L.fd_p2p_info(self.handle, C.byref(n), C.byref(r), C.byref(sb), C.byref(u)))
        return {"nranks": n.value, "rank": r.value, "slot_bytes": sb.value, "uncached": bool(u.value)}

    def status(self):
        """0, or 1 + r once a wait for rank r timed out (FDJAC_P2P_TIMEOUT_MS, default 2000)."""
        v = C.c_int()
        _l.check(self.L.fd_p2p_status(self.handle, C.byref(v)))
        return v.value

    def allgather(self, buf, slot_elems):
        """In place, like Comm.allgather: rank r's data sits in slot r of `buf` (nranks slots of slot_elems elements)."""
        p, eb = Comm._dev(buf, "buf")
        if buf.numel() < self.nranks * int(slot_elems):
            raise ValueError("buf is shorter than nranks * slot_elems")
        _l.check(self.L.fd_p2p_allgather(self.handle, p, int(slot_elems) * eb))

    def halo_exchange(self, buf, own_begin, own_end, halo):
        p, eb = Comm._dev(buf, "buf")
        _l.check(self.L.fd_p2p_halo_exchange(self.handle, p, int(own_begin), int(own_end), int(halo), eb))


class BandedSolver:
    """fd_banded_solver: (alpha*I + beta*J) y = b on the device for a BANDED J (0 <= l, u <= 4) in the storage the banded plans fill --
    ``BandedMatrix`` data ((l+u+1) x N column-major) or the nzval of the exact band as ``SparseMatrixCSC``.  Block cyclic reduction,
    no pivoting: a system with a row that is not diagonally dominant is refused (NaN, ``status()`` bit 0) unless ``set_policy(True)``.
    The consumer of the Jacobian path for a ``BandedMatrix`` jac_prototype."""

    def __init__(self, N, l, u, layout="banded", ctx=None, dtype=np.float64):
        self.ctx = ctx or Context.default()
        self.dtype = np.dtype(dtype)
        self.Lt = _l.typed(self.ctx.L, self.dtype)
        self.layout = {"banded": 0, "csc": 1}[layout]
        h = C.c_void_p()
        _l.check(self.Lt.fd_banded_solver_create(self.ctx.handle, int(N), int(l), int(u), self.layout, C.byref(h)))
        self.handle, self.N, self.l, self.u = h, int(N), int(l), int(u)
        self._fin = weakref.finalize(self, self.Lt.fd_banded_solver_destroy, h)

    def _dev(self, a, what):
        p, k, _keep = _ptr(a, what, self.dtype)
        if k != _l.DEVICE:
            raise ValueError("the solver takes device arrays")
        return p

    def solve(self, J, b, y, alpha=1.0, beta=-1.0):
        """Enqueue y = (alpha*I + beta*J)^-1 b on the context's stream (fd_banded_solve_async)."""
        vals = J.data if isinstance(J, BandedMatrix) else (J.nzval if isinstance(J, SparseMatrixCSC) else J)
        _l.check(self.Lt.fd_banded_solve_async(self.handle, float(alpha), float(beta), self._dev(vals, "J"), self._dev(b, "b"), self._dev(y, "y")))

    def set_policy(self, trust_non_dominant):
        _l.check(self.Lt.fd_banded_solver_set_policy(self.handle, 1 if trust_non_dominant else 0))

    def status(self):
        v = C.c_int()
        _l.check(self.Lt.fd_banded_solver_status(self.handle, C.byref(v)))
        return v.value


class BlockTridiagSolver:
    """fd_blocktridiag_solver: (alpha*I + beta*J) y = b on the device for a block-tridiagonal J of ``nblk`` dense ``b x b`` blocks
    (b <= 32) in ``BlockBandedMatrix`` data (block bandwidths (1, 1), uniform block sizes) -- the storage a block-banded plan fills
    (BASELINE's config 5).  Block cyclic reduction, no pivoting: a system with a row that is not diagonally dominant is refused (NaN,
    ``status()`` bit 0) unless ``set_policy(True)``."""

    def __init__(self, nblk, block_size, ctx=None, dtype=np.float64):
        self.ctx = ctx or Context.default()
        self.dtype = np.dtype(dtype)
        self.Lt = _l.typed(self.ctx.L, self.dtype)
        h = C.c_void_p()
        _l.check(self.Lt.fd_blocktridiag_solver_create(self.ctx.handle, int(nblk), int(block_size), C.byref(h)))
        self.handle, self.nblk, self.b = h, int(nblk), int(block_size)
        self._fin = weakref.finalize(self, self.Lt.fd_blocktridiag_solver_destroy, h)

    def _dev(self, a, what):
        p, k, _keep = _ptr(a, what, self.dtype)
        if k != _l.DEVICE:
            raise ValueError("the solver takes device arrays")
        return p

    def solve(self, J, b, y, alpha=1.0, beta=-1.0):
        """Enqueue y = (alpha*I + beta*J)^-1 b on the context's stream (fd_blocktridiag_solve_async)."""
        vals = J.data if isinstance(J, BlockBandedMatrix) else J
        _l.check(self.Lt.fd_blocktridiag_solve_async(self.handle, float(alpha), float(beta), self._dev(vals, "J"), self._dev(b, "b"), self._dev(y, "y")))

    def set_policy(self, trust_non_dominant):
        _l.check(self.Lt.fd_blocktridiag_solver_set_policy(self.handle, 1 if trust_non_dominant else 0))

    def status(self):
        v = C.c_int()
        _l.check(self.Lt.fd_blocktridiag_solver_status(self.handle, C.byref(v)))
        return v.value


class TridiagSolver:
    """fd_tridiag_solver: (alpha*I + beta*J) y = b on the device for a tridiagonal J in the storage the Jacobian plans
    fill -- ``Tridiagonal`` (dl, d, du) or the nzval of a tridiagonal ``SparseMatrixCSC`` -- whole or one rank's column
    range [rows[0], rows[1]) (then pass ``comm`` to ``solve``: the ranks solve one global system, exchanging 8 numbers
    each).  The consumer of the Jacobian path: test/downstream/ordinarydiffeq_tridiagonal_solve.jl:18-30."""

    def __init__(self, N, layout="diagonals", rows=None, ctx=None, dtype=np.float64):
        self.ctx = ctx or Context.default()
        self.dtype = np.dtype(dtype)
        self.Lt = _l.typed(self.ctx.L, self.dtype)
        self.layout = {"diagonals": _l.TRI_DIAGONALS, "csc": _l.TRI_CSC}[layout]
        r0, r1 = (0, 0) if rows is None else (int(rows[0]), int(rows[1]))
        h = C.c_void_p()
        _l.check(self.Lt.fd_tridiag_solver_create(self.ctx.handle, int(N), r0, r1, self.layout, C.byref(h)))
        self.handle, self.N = h, int(N)
        self.nlocal = int(N) if rows is None else r1 - r0
        self._fin = weakref.finalize(self, self.Lt.fd_tridiag_solver_destroy, h)

    def _jptrs(self, J):
        arrs = [J.dl, J.d, J.du] if isinstance(J, Tridiagonal) else ([J.nzval] if isinstance(J, SparseMatrixCSC) else list(J))
        if len(arrs) != (3 if self.layout == _l.TRI_DIAGONALS else 1):
            raise ValueError("J does not match the solver's layout")
        ptrs = []
        for a in arrs:
            if a is None or (hasattr(a, "numel") and a.numel() == 0):
                ptrs.append(None)
                continue
            p, k, _keep = _ptr(a, "J", self.dtype)
            if k != _l.DEVICE:
                raise ValueError("the solver takes device arrays")
            ptrs.append(p)
        return (C.c_void_p * 3)(*(ptrs + [None] * (3 - len(ptrs))))

    def _vec(self, a, what):
        p, k, _keep = _ptr(a, what, self.dtype)
        if k != _l.DEVICE:
            raise ValueError("the solver takes device arrays")
        return p

    def solve(self, J, b, y, alpha=1.0, beta=-1.0, comm=None):
        """Enqueue y = (alpha*I + beta*J)^-1 b on the context's stream (fd_tridiag_solve_async)."""
        _l.check(self.Lt.fd_tridiag_solve_async(self.handle, float(alpha), float(beta), self._jptrs(J), self._vec(b, "b"),
                                                self._vec(y, "y"), comm.handle if comm is not None else None))

    def set_policy(self, trust_non_dominant):
        """fd_tridiag_solver_set_policy: False (default) = a solve that meets a row without diagonal dominance writes NaN instead of a
        solution the pivot-free elimination cannot vouch for; True = the result anyway (the status flag is raised either way)."""
        _l.check(self.Lt.fd_tridiag_solver_set_policy(self.handle, 1 if trust_non_dominant else 0))

    def status(self):
        """fd_tridiag_solver_status: bit 0 = the last solve met a row that is not diagonally dominant (no pivoting: y is NaN unless trusted)."""
        v = C.c_int()
        _l.check(self.Lt.fd_tridiag_solver_status(self.handle, C.byref(v)))
        return v.value

    def interface(self, J, b, packet, alpha=1.0, beta=-1.0):
        """Phase A (fd_tridiag_solve_interface): this rank's 8-double packet into the device tensor ``packet``."""
        _l.check(self.Lt.fd_tridiag_solve_interface(self.handle, float(alpha), float(beta), self._jptrs(J), self._vec(b, "b"),
                                                    packet.data_ptr()))

    def finish(self, J, b, packets, rank, nranks, y, alpha=1.0, beta=-1.0):
        """Phases B + C (fd_tridiag_solve_finish) from all ranks' packets (device tensor, nranks x 8 doubles)."""
        _l.check(self.Lt.fd_tridiag_solve_finish(self.handle, float(alpha), float(beta), self._jptrs(J), self._vec(b, "b"),
                                                 packets.data_ptr(), int(rank), int(nranks), self._vec(y, "y")))


class BuiltinF:
    """One of libfdjac's device f! families (fd_builtin_f_create): the reference's fixtures."""

    def __init__(self, family, *params, ctx=None, dtype=np.float64):
        self.ctx = ctx or Context.default()
        self.dtype = np.dtype(dtype)
        L = self.Lt = _l.typed(self.ctx.L, self.dtype)
        prm = (C.c_int64 * max(len(params), 1))(*[int(p) for p in params])
        self.fn = _l.F_LAUNCH()
        self.fctx = C.c_void_p()
        _l.check(L.fd_builtin_f_create(self.ctx.handle, _l.FAMILIES[family], prm, len(params), C.byref(self.fn),
                                       C.byref(self.fctx)))
        self.family, self.params = family, params
        self._fin = weakref.finalize(self, L.fd_builtin_f_destroy, self.fctx)

    @classmethod
    def sparse(cls, M, N, colptr, rowval, ctx=None, dtype=np.float64, idx_base=1):
        """fd_builtin_f_create_sparse: the residual with ANY given sparsity pattern -- f_r = sum over the pattern's entries (r, j),
        ascending j, of w(r, j) phi(x_j) (include/fdjac.h, FD_F_SPARSE); `colptr` / `rowval` = the CSC pattern of its Jacobian."""
        self = cls.__new__(cls)
        self.ctx = ctx or Context.default()
        self.dtype = np.dtype(dtype)
        L = self.Lt = _l.typed(self.ctx.L, self.dtype)
        cp, rv = _i64(colptr), _i64(rowval)
        self.fn = _l.F_LAUNCH()
        self.fctx = C.c_void_p()
        _l.check(L.fd_builtin_f_create_sparse(self.ctx.handle, int(M), int(N), _vp(cp), _vp(rv), 8, int(idx_base), C.byref(self.fn),
                                              C.byref(self.fctx)))
        self.family, self.params = "sparse", (int(M), int(N))
        self._fin = weakref.finalize(self, L.fd_builtin_f_destroy, self.fctx)
        return self

    @property
    def lazy_fn(self):
        """The family's lazy-point launcher (fd_builtin_f_lazy) or None if it has none."""
        fn = _l.F_LAUNCH_LAZY()
        rc = self.Lt.fd_builtin_f_lazy(self.fctx, C.byref(fn))
        return fn if rc == 0 else None

    @property
    def lazy_jvp_fn(self):
        """The family's lazy-point launcher for JVPs (fd_builtin_f_lazy_jvp) or None if it has none."""
        fn = _l.F_LAUNCH_LAZY_JVP()
        rc = self.Lt.fd_builtin_f_lazy_jvp(self.fctx, C.byref(fn))
        return fn if rc == 0 else None

    @property
    def lazy_jvp_caps(self):
        """FD_LAZY_JVP_CAP_* bits of the lazy JVP launcher (fd_builtin_f_lazy_jvp_caps)."""
        caps = C.c_int32()
        rc = self.Lt.fd_builtin_f_lazy_jvp_caps(self.fctx, C.byref(caps))
        return caps.value if rc == 0 else 0

    @property
    def lazy_caps(self):
        """FD_LAZY_CAP_* bits of the lazy launcher (fd_builtin_f_lazy_caps)."""
        caps = C.c_int32()
        rc = self.Lt.fd_builtin_f_lazy_caps(self.fctx, C.byref(caps))
        return caps.value if rc == 0 else 0

    def counts(self):
        a, b = C.c_int64(), C.c_int64()
        _l.check(self.Lt.fd_builtin_f_counts(self.fctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    @property
    def fcalls(self):
        return self.counts()[1]

    def row_stores(self):
        """Storing launches of the sparse family that went row by row (FD_F_INFO_ROW_STORES)."""
        v = C.c_int64()
        _l.check(self.Lt.fd_builtin_f_info(self.fctx, 1, C.byref(v)))
        return v.value


class JitF:
    """A row functor given as SOURCE and compiled at run time (fd_f_compile_rows; hiprtc against include/fdjac_device.h,
    -ffp-contract=off): the shim's `DeviceF(src::String)`.  `source` defines a functor type `name` with
        template <class P> __device__ real_t operator()(long long r, const P &X) const     // row r at the point X, X(j) = coordinate j
    `params` = the functor object's bytes (b"" for an empty struct; e.g. struct.pack("q", n) for `struct F { long long n; ... }`).
    With a plan created with store_csc=True (store_csc_always=True on banded patterns) and `plan.set_lazy(f)` the whole Jacobian is
    the step-size launch + ONE launch of the column store instantiated for the functor -- what the built-in families get."""

    def __init__(self, source, name, M, N, params=b"", ctx=None, dtype=np.float64):
        self.ctx = ctx or Context.default()
        self.dtype = np.dtype(dtype)
        L = self.L = self.ctx.L
        self.fn = _l.F_LAUNCH()
        self._lazy = _l.F_LAUNCH_LAZY()
        self.fctx = C.c_void_p()
        caps = C.c_int32()
        params = bytes(params)
        buf = C.create_string_buffer(params, len(params)) if params else None
        rc = L.fd_f_compile_rows(self.ctx.handle, source.encode(), name.encode(), buf, len(params), int(M), int(N), self.dtype.itemsize,
                                 C.byref(self.fn), C.byref(self._lazy), C.byref(caps), C.byref(self.fctx))
        self.log = (L.fd_f_compile_log() or b"").decode("utf-8", "replace")
        _l.check(rc)
        self.lazy_caps = caps.value
        self.M, self.N = int(M), int(N)
        self._fin = weakref.finalize(self, L.fd_f_compiled_destroy, self.fctx)

    @property
    def lazy_fn(self):
        return self._lazy

    @property
    def launches(self):
        n = C.c_int64()
        _l.check(self.L.fd_f_compiled_counts(self.fctx, C.byref(n)))
        return n.value


class JitTerms(JitF):
    """A SEPARABLE residual given by its TERM alone (fd_f_compile_terms; include/fdjac_device.h "SEPARABLE residuals"): row r of f is the
    left-to-right sum, over the stored entries (r, j) of the plan's own pattern in ascending j, of `term(r, j, x[j])`.  `source` defines
        struct Name { <parameters>;  template <class T> __device__ T term(long long r, long long j, T v) const { ... } };
    `plan`: made with store_rows=True -- the functor reads that plan's row lists (the plan must outlive it).  Everything a `JitF` gets,
    all from `term`, plus the ROW-WISE store on that plan: 2 L term evaluations per row of L entries instead of L^2, same bits."""

    def __init__(self, source, name, plan, params=b"", dtype=None):
        self.ctx = plan.ctx
        self.dtype = np.dtype(dtype if dtype is not None else plan.dtype)
        self.plan = plan                       # (keeps the lists alive)
        L = self.L = self.ctx.L
        rl = plan.row_lists()
        M, N = plan.info(_l.INFO_M), plan.info(_l.INFO_N)
        self.fn = _l.F_LAUNCH()
        self._lazy = _l.F_LAUNCH_LAZY()
        self.fctx = C.c_void_p()
        caps = C.c_int32()
        params = bytes(params)
        buf = C.create_string_buffer(params, len(params)) if params else None
        rc = L.fd_f_compile_terms(self.ctx.handle, source.encode(), name.encode(), buf, len(params), int(M), int(N), self.dtype.itemsize,
                                  rl["row_ptr"], rl["row_col"], rl["serial"], C.byref(self.fn), C.byref(self._lazy), C.byref(caps), C.byref(self.fctx))
        self.log = (L.fd_f_compile_log() or b"").decode("utf-8", "replace")
        _l.check(rc)
        self.lazy_caps = caps.value
        self.M, self.N = int(M), int(N)
        self._fin = weakref.finalize(self, L.fd_f_compiled_destroy, self.fctx)

    @property
    def row_stores(self):
        n = C.c_int64()
        _l.check(self.L.fd_f_compiled_row_stores(self.fctx, C.byref(n)))
        return n.value


class BitcodeF(JitF):
    """A row function given as LLVM BITCODE (fd_f_link_rows_bitcode): what AMDGPU.jl / GPUCompiler emit for a Julia closure -- the shim's
    `DeviceF(f::Function, M, N)` -- or `hipcc -fgpu-rdc -emit-llvm --offload-device-only -c`.  The bitcode defines
    `fdjac_user_row(params, r, X)` (and `fdjac_user_row_c` for the complex step) and reads the point with `fdjac_point_get(X, j)`;
    `params` are handed to it as they are.  Everything a `JitF` gets: the column store, the band store, the complex step."""

    def __init__(self, bitcode, M, N, params=b"", ctx=None, dtype=np.float64):
        self.ctx = ctx or Context.default()
        self.dtype = np.dtype(dtype)
        L = self.L = self.ctx.L
        self.fn = _l.F_LAUNCH()
        self._lazy = _l.F_LAUNCH_LAZY()
        self.fctx = C.c_void_p()
        caps = C.c_int32()
        bitcode, params = bytes(bitcode), bytes(params)
        bbuf = C.create_string_buffer(bitcode, len(bitcode))
        pbuf = C.create_string_buffer(params, len(params)) if params else None
        rc = L.fd_f_link_rows_bitcode(self.ctx.handle, bbuf, len(bitcode), pbuf, len(params), int(M), int(N), self.dtype.itemsize,
                                      C.byref(self.fn), C.byref(self._lazy), C.byref(caps), C.byref(self.fctx))
        self.log = (L.fd_f_compile_log() or b"").decode("utf-8", "replace")
        _l.check(rc)
        self.lazy_caps = caps.value
        self.M, self.N = int(M), int(N)
        self._fin = weakref.finalize(self, L.fd_f_compiled_destroy, self.fctx)


class _DevView:
    """__cuda_array_interface__ holder so torch can view library-owned device memory."""

    def __init__(self, ptr, n, complex_, f32=False):
        ts = ("<c8" if complex_ else "<f4") if f32 else ("<c16" if complex_ else "<f8")
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": ts,
                                         "data": (int(ptr), False), "version": 2, "strides": None}


class TorchF:
    """A user f!(fx, x) written with torch ops on CUDA tensors, as an fd_f_launch callback.

    The callable receives 1-D views of library-owned device memory (complex128 in the
    complex-step arm) and must write fx in place, exactly like a Julia ``f!(fx, x)`` operating on
    ROCArrays.  Work is enqueued on the context's stream; nothing synchronises.
    """

    def __init__(self, fn, M, N, ctx=None, dtype=np.float64):
        import torch
        self.ctx = ctx or Context.default()
        self.dtype = np.dtype(dtype)
        f32 = self.dtype == np.float32
        self.fn_py, self.M, self.N = fn, int(M), int(N)
        self.fcalls = 0
        self.error = None
        dev_index = self.ctx.device

        def _launch(_fctx, fx, x, nbatch, xs, fs, r0, r1, is_complex, stream):
            try:
                sz = (8 if is_complex else 4) if f32 else (16 if is_complex else 8)
                # run the user's torch ops on the stream the LIBRARY hands over (the one its own kernels are enqueued
                # on), whatever torch's current stream is at call time: the legacy default stream when `stream` is NULL
                ts = (torch.cuda.ExternalStream(int(stream), device=dev_index) if stream
                      else torch.cuda.default_stream(dev_index))
                with torch.cuda.stream(ts):
                    for b in range(nbatch):
                        xv = torch.as_tensor(_DevView(x + b * xs * sz, self.N, is_complex, f32), device="cuda")
                        fv = torch.as_tensor(_DevView(fx + b * fs * sz, self.M, is_complex, f32), device="cuda")
                        self.fcalls += 1
                        fn(fv, xv)
                return 0
            except BaseException as e:  # never let an exception cross the C ABI
                self.error = e
                return 1

        self.fn = _l.F_LAUNCH(_launch)
        self.fctx = C.c_void_p()


class Plan:
    """fd_plan handle."""

    def __init__(self, ctx, handle, fdtype, dtype=np.float64, cx=False):
        self.ctx, self.handle, self.fdtype, self.dtype = ctx, handle, fdtype, np.dtype(dtype)
        self.cx = bool(cx)      # FD_PLAN_COMPLEX_X: x / f_in / outs are complex arrays over `dtype`
        self.Lt = _l.typed(ctx.L, self.dtype)
        self._fin = weakref.finalize(self, self.Lt.fd_plan_destroy, handle)

    def info(self, key):
        v = C.c_int64()
        _l.check(self.Lt.fd_plan_info(self.handle, key, C.byref(v)))
        return v.value

    def row_lists(self):
        """The plan's pattern BY ROWS on the device (fd_plan_row_lists; plans made with store_rows=True): device addresses of row_ptr,
        row_col, row_slot, the number of stored entries and the plan's serial.  Owned by the plan."""
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n, ser = C.c_int64(), C.c_uint64()
        _l.check(self.Lt.fd_plan_row_lists(self.handle, C.byref(a), C.byref(b), C.byref(c), C.byref(n), C.byref(ser)))
        return {"row_ptr": a.value, "row_col": b.value, "row_slot": c.value, "entries": n.value, "serial": ser.value}

    def checksum(self):
        """fd_plan_checksum: FNV-1a of the compiled pattern (plans with equal checksums drive the kernels identically)."""
        v = C.c_uint64()
        _l.check(self.Lt.fd_plan_checksum(self.handle, C.byref(v)))
        return v.value

    def stale(self):
        """fd_plan_stale: has a completed deferred check (``matches(..., deferred=True)``) found a mismatch?  Non-blocking, sticky."""
        v = C.c_int(0)
        _l.check(self.Lt.fd_plan_stale(self.handle, C.byref(v)))
        return bool(v.value)

    def matches(self, idx_a=None, idx_b=None, colorvec=None, idx_base=1, deferred=False):
        """fd_plan_matches: do these arrays still hold the content the plan was compiled from?  deferred=True (device arrays):
        fd_plan_matches_async -- ONE fused kernel, the verdict raised on the device and reported by ``stale()`` / as FD_ERR_STALE by the
        next call on the plan / by ``Context.synchronize()``; returns None at once.  `idx_a` / `idx_b` are colptr /
        rowval (CSC plans) or rows_index / cols_index (index-list plans), `colorvec` the colours; numpy arrays (host threads) or
        torch CUDA tensors (kernels), int32 or int64, all on the same side; None = not compared.  Needs a plan created with
        ``fingerprint=True`` (FD_PLAN_FINGERPRINT)."""
        pa = _l.PatternArrays()
        kinds = set()
        keep = []
        for name, a, lenf, ptrf in (("idx_a", idx_a, "len_a", "idx_a"), ("idx_b", idx_b, "len_b", "idx_b"),
                                    ("colorvec", colorvec, "len_color", "colorvec")):
            if a is None:
                continue
            if _is_torch(a):
                import torch
                if not (a.is_contiguous() and a.dtype in (torch.int32, torch.int64)):
                    raise TypeError("%s must be a contiguous int32 / int64 array" % name)
                ptr, n, nb, kind = a.data_ptr(), a.numel(), a.element_size(), (_l.DEVICE if a.is_cuda else _l.HOST)
            else:
                a = np.ascontiguousarray(a)
                if a.dtype not in (np.int32, np.int64):
                    a = a.astype(np.int64)
                ptr, n, nb, kind = a.ctypes.data, a.size, a.itemsize, _l.HOST
            keep.append(a)
            kinds.add(kind)
            setattr(pa, ptrf, ptr)
            setattr(pa, lenf, n)
            if name == "colorvec":
                pa.color_bytes = nb
            else:
                if pa.idx_bytes not in (0, nb):
                    raise TypeError("idx_a and idx_b must have the same integer type")
                pa.idx_bytes = nb
        if len(kinds) > 1:
            raise ValueError("the arrays must all be host or all be device arrays")
        pa.memkind = kinds.pop() if kinds else _l.HOST
        pa.idx_base = int(idx_base)
        if pa.idx_bytes == 0:
            pa.idx_bytes = 8
        if pa.color_bytes == 0:
            pa.color_bytes = 8
        if deferred:
            _l.check(self.Lt.fd_plan_matches_async(self.handle, C.byref(pa)))
            return None
        m = C.c_int(0)
        _l.check(self.Lt.fd_plan_matches(self.handle, C.byref(pa), C.byref(m)))
        return bool(m.value)

    @property
    def ncolors(self):
        return self.info(_l.INFO_NCOLORS)

    @property
    def nouts(self):
        return self.info(_l.INFO_NOUTS)

    def out_len(self, k=0):
        return self.info(_l.INFO_OUT0_LEN + k)

    @property
    def row_window(self):
        return self.info(_l.INFO_ROW_BEGIN), self.info(_l.INFO_ROW_END)

    @property
    def fcalls_last(self):
        return self.info(_l.INFO_FCALLS_LAST)

    def epsilons(self):
        n = self.ncolors
        buf = (C.c_double * max(n, 1))()
        _l.check(self.Lt.fd_plan_get_epsilons(self.handle, buf))
        return np.array(buf[:n])

    def fused_trace(self):
        """fd_plan_fused_trace: the device-clock marks of the last fused launch, in microseconds relative to its first mark."""
        m = (C.c_longlong * 16)()
        _l.check(self.Lt.fd_plan_fused_trace(self.handle, m))
        names = ("eps_first_start", "eps_last_published", "finisher_start", "finisher_has_sums", "eps_published", "store_first_start",
                 "store_first_has_eps", "store_last_has_eps", "store_last_done", "store_last_start")
        v = [m[i] for i in range(10)]
        t0 = min(x for x in v if x > 0)
        return {n: (x - t0) / 100.0 for n, x in zip(names, v)}

    def enable_timing(self, level=2):
        """0 off; 1 = the diff+decompress kernel only (2 events per call); 2 = every stage and the whole call; 3 = the whole call
        only; 4 = every stage without the whole-call markers."""
        _l.check(self.Lt.fd_plan_enable_timing(self.handle, int(level)))

    def set_timing_stride(self, stride):
        """Level-1 timing on every `stride`-th call only (fd_plan_set_timing_stride): a pair of events costs ~8 us of stream time."""
        _l.check(self.Lt.fd_plan_set_timing_stride(self.handle, int(stride)))

    def timings(self):
        ms = (C.c_double * len(_l.STAGES))()
        cnt = (C.c_int64 * len(_l.STAGES))()
        _l.check(self.Lt.fd_plan_get_timings(self.handle, ms, cnt))
        return {s: {"ms_sum": ms[i], "launches": cnt[i]} for i, s in enumerate(_l.STAGES)}

    def timing_samples(self, stage):
        """Individual span durations (ms) of `stage` ("eps", "perturb", "f", "decompress", "total") since timing was
        enabled (fd_plan_get_timing_samples) -- medians over individually timed runs."""
        k = _l.STAGES.index(stage)
        n = C.c_int64(0)
        _l.check(self.Lt.fd_plan_get_timing_samples(self.handle, k, None, 0, C.byref(n)))
        buf = (C.c_double * max(n.value, 1))()
        _l.check(self.Lt.fd_plan_get_timing_samples(self.handle, k, buf, n.value, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def set_lazy(self, f, imag_only=True, row_window=True, diff=True, store=None, csc_base=True, fused=True):
        """Use f's lazy-point launcher (fd_plan_set_lazy_f) for the perturbed batches; f=None clears it.  imag_only /
        row_window / diff / store=False withhold the launcher's FD_LAZY_CAP_IMAG_ONLY / FD_LAZY_CAP_ROW_WINDOW /
        FD_LAZY_CAP_DIFF / FD_LAZY_CAP_STORE capability (store defaults to diff: a launcher that may not even subtract
        hands over plain values)."""
        fn = getattr(f, "lazy_fn", None) if f is not None else None
        if f is not None and fn is None:
            raise ValueError("this f! has no lazy-point launcher")
        self._lazy_keep = fn
        _l.check(self.Lt.fd_plan_set_lazy_f(self.handle, fn if fn is not None else _l.F_LAUNCH_LAZY()))
        caps = int(getattr(f, "lazy_caps", 0)) if f is not None else 0
        if not imag_only:
            caps &= ~1
        if not row_window:
            caps &= ~2
        if not diff:
            caps &= ~4
        if not (diff if store is None else store):
            caps &= ~(8 | 16 | 32 | 64 | 256)      # FD_LAZY_CAP_STORE, FD_LAZY_CAP_STORE_CSC, _BASE, _COMPLEX and FD_LAZY_CAP_STORE_COLRANGE
        if not csc_base:
            caps &= ~32                 # (the column store takes f(x) from ONE plain evaluation instead of forming it itself)
        if not fused:
            caps &= ~128                # FD_LAZY_CAP_FUSED_EPS withheld: the step sizes come from the library's own launch
        _l.check(self.Lt.fd_plan_set_lazy_caps(self.handle, caps))

    def set_comm(self, comm):
        """Shard the step-size reduction over the communicator's ranks (fd_plan_set_comm); None detaches."""
        self._comm_keep = comm
        _l.check(self.Lt.fd_plan_set_comm(self.handle, comm.handle if comm is not None else None))

    def set_p2p(self, p2p):
        """The same with a bare mailbox (fd_plan_set_p2p: callers without RCCL); None detaches."""
        self._p2p_keep = p2p
        _l.check(self.Lt.fd_plan_set_p2p(self.handle, p2p.handle if p2p is not None else None))

    def set_halo(self, own_begin, own_end, halo):
        """x is sharded (fd_plan_set_halo): every following call exchanges `halo` elements of x with the neighbour ranks itself, in
        the launch that carries the step-size reduction's group sums.  halo = 0 turns it off."""
        _l.check(self.Lt.fd_plan_set_halo(self.handle, int(own_begin), int(own_end), int(halo)))

    def eps_shard_range(self, shard, nshards):
        """fd_plan_eps_shard_range: the elements of x shard `shard` of `nshards` of the step-size reduction reads."""
        a, b = C.c_int64(), C.c_int64()
        _l.check(self.Lt.fd_plan_eps_shard_range(self.handle, int(shard), int(nshards), C.byref(a), C.byref(b)))
        return a.value, b.value

    def eps_partials(self, x, shard, nshards):
        """Enqueue shard `shard` of `nshards` of the masked sums of squares (fd_plan_eps_partials); returns
        (device address of the partial buffer, doubles per slot)."""
        xp, xk, _k = _ptr(x, "x", self.dtype)
        if xk != _l.DEVICE:
            raise ValueError("eps_partials needs a device array")
        ptr, slot = C.c_void_p(), C.c_int64()
        _l.check(self.Lt.fd_plan_eps_partials(self.handle, xp, int(shard), int(nshards), C.byref(ptr), C.byref(slot)))
        return ptr.value, slot.value

    def eps_finalize(self, relstep=None, absstep=None, dir=True):
        _l.check(self.Lt.fd_plan_eps_finalize(self.handle, -1.0 if relstep is None else float(relstep),
                                              -1.0 if absstep is None else float(absstep), float(dir)))

    def set_eps_mode(self, precomputed):
        _l.check(self.Lt.fd_plan_set_eps_mode(self.handle, _l.EPS_PRECOMPUTED if precomputed else _l.EPS_COMPUTE))

    def jacobian(self, f, x, outs, f_in=None, relstep=None, absstep=None, dir=True, sync=True):
        """fd_jacobian / fd_jacobian_async on raw arrays (torch CUDA tensors or numpy arrays)."""
        L = self.Lt
        if np.dtype(getattr(f, "dtype", self.dtype)) != self.dtype:
            raise TypeError("f! launcher is built for %s, the plan for %s" % (np.dtype(f.dtype).name, self.dtype.name))
        xp, xk, _k1 = _ptr(x, "x", self.dtype, self.cx)
        ptrs, kinds, keep = [], set(), []
        for o in outs:
            p, k, ka = _ptr(o, "output", self.dtype, self.cx)
            ptrs.append(p)
            kinds.add(k)
            keep.append(ka)
        if len(kinds) != 1:
            raise ValueError("outputs must all be host or all be device arrays")
        ok = kinds.pop()
        arr = (C.c_void_p * 3)(*(ptrs + [None] * (3 - len(ptrs))))
        fp, fk = None, _l.DEVICE
        if f_in is not None:
            fp, fk, _k2 = _ptr(f_in, "f_in", self.dtype, self.cx)
        rel = -1.0 if relstep is None else float(relstep)
        ab = -1.0 if absstep is None else float(absstep)
        if not sync:
            if xk != _l.DEVICE or ok != _l.DEVICE or (f_in is not None and fk != _l.DEVICE):
                raise ValueError("the async path needs device arrays")
            rc = L.fd_jacobian_async(self.handle, f.fn, f.fctx, xp, fp, rel, ab, float(dir), arr)
        else:
            rc = L.fd_jacobian(self.handle, f.fn, f.fctx, xp, xk, fp, fk, rel, ab, float(dir), arr, ok)
        err = getattr(f, "error", None)
        if err is not None:
            f.error = None
            raise err
        _l.check(rc)


    def jacobian_owned(self, f, x, outs, comm=None, f_in=None, relstep=None, absstep=None, dir=True):
        """fd_jacobian_owned_async: the call of a colour-owning plan (make_plan(..., color_range=...)) with its assembly -- zero-fill,
        this rank's colours, ONE in-place all-reduce per output over `comm` (None: the caller sums the ranks' outputs).  Device arrays;
        enqueues and returns.  Right call after call on the same buffers, which summing the outputs of plain calls is not."""
        L = self.Lt
        xp, xk, _k1 = _ptr(x, "x", self.dtype, self.cx)
        ptrs, keep = [], []
        for o in outs:
            p, k, ka = _ptr(o, "output", self.dtype, self.cx)
            if k != _l.DEVICE:
                raise ValueError("jacobian_owned() needs device arrays")
            ptrs.append(p)
            keep.append(ka)
        if xk != _l.DEVICE:
            raise ValueError("jacobian_owned() needs device arrays")
        arr = (C.c_void_p * 3)(*(ptrs + [None] * (3 - len(ptrs))))
        fp = None
        if f_in is not None:
            fp, fk, _k2 = _ptr(f_in, "f_in", self.dtype, self.cx)
            if fk != _l.DEVICE:
                raise ValueError("jacobian_owned() needs device arrays")
        rc = L.fd_jacobian_owned_async(self.handle, comm.handle if comm is not None else None, f.fn, f.fctx, xp, fp,
                                       -1.0 if relstep is None else float(relstep), -1.0 if absstep is None else float(absstep), float(dir), arr)
        err = getattr(f, "error", None)
        if err is not None:
            f.error = None
            raise err
        _l.check(rc)

    def bind(self, f, x, outs, f_in=None, relstep=None, absstep=None, dir=True):
        """Validate (f, x, outs) once and return a zero-argument callable that enqueues the call on the context's
        stream (fd_jacobian_async) -- what a compiled caller of the ABI does: pointers resolved once, one foreign call
        per Jacobian.  Device arrays only; the arrays are kept alive by the callable."""
        L = self.Lt
        if np.dtype(getattr(f, "dtype", self.dtype)) != self.dtype:
            raise TypeError("f! launcher is built for %s, the plan for %s" % (np.dtype(f.dtype).name, self.dtype.name))
        keep = [f, x, list(outs), f_in]
        xp, xk, _k = _ptr(x, "x", self.dtype)
        ptrs = []
        for o in outs:
            p, k, _k = _ptr(o, "output", self.dtype)
            if k != _l.DEVICE:
                raise ValueError("bind() needs device arrays")
            ptrs.append(p)
        fp = None
        if f_in is not None:
            fp, fk, _k = _ptr(f_in, "f_in", self.dtype)
            if fk != _l.DEVICE:
                raise ValueError("bind() needs device arrays")
        if xk != _l.DEVICE:
            raise ValueError("bind() needs device arrays")
        arr = (C.c_void_p * 3)(*(ptrs + [None] * (3 - len(ptrs))))
        rel = C.c_double(-1.0 if relstep is None else float(relstep))
        ab = C.c_double(-1.0 if absstep is None else float(absstep))
        dr = C.c_double(float(dir))
        fn, handle, ffn, fctx, xpp, fpp = L.fd_jacobian_async, self.handle, f.fn, f.fctx, C.c_void_p(xp), C.c_void_p(fp)

        def call():
            rc = fn(handle, ffn, fctx, xpp, fpp, rel, ab, dr, arr)
            if rc:
                err = getattr(f, "error", None)
                if err is not None:
                    f.error = None
                    raise err
                _l.check(rc)
        call.keep = keep
        return call


def _opts(fdtype, col_window=None, x_window=None, scratch_bytes=0, color_range=None, eps_contiguous=False, fingerprint=False,
          store_csc=False, store_csc_always=False, store_rows=False):
    o = _l.PlanOpts()
    o.fdtype = _l.FDTYPES[_norm_fdtype(fdtype)]
    o.flags = ((_l.PLAN_EPS_CONTIGUOUS if eps_contiguous else 0) | (_l.PLAN_FINGERPRINT if fingerprint else 0) |
               (_l.PLAN_STORE_CSC if (store_csc or store_csc_always or store_rows) else 0) | (_l.PLAN_STORE_CSC_ALWAYS if store_csc_always else 0) |
               (_l.PLAN_STORE_CSC_ROWS if store_rows else 0))
    if col_window is not None:
        o.col_begin, o.col_end = int(col_window[0]), int(col_window[1])
    if x_window is not None:
        o.x_begin, o.x_end = int(x_window[0]), int(x_window[1])
    o.scratch_bytes = int(scratch_bytes)
    if color_range is not None:   # owned colours, 0-based [begin, end)
        o.color_begin, o.color_end = int(color_range[0]), int(color_range[1])
    return o


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def make_plan(J, sparsity, colorvec, fdtype, ctx=None, col_window=None, x_window=None, scratch_bytes=0,
              color_range=None, dtype=np.float64, eps_contiguous=False, complex_x=False, fingerprint=False, store_csc=False,
              store_csc_always=False, store_rows=False):
    """Compile (J type, sparsity, colorvec) into a device plan -- the dispatch the reference performs
    per call through `_colorediteration!` / `_use_findstructralnz` / `_use_sparseCSC_common_sparsity`
    (src/jacobians.jl:524-535; ext/*.jl)."""
    ctx = ctx or Context.default()
    L = _l.typed(ctx.L, dtype)      # fd_* for Float64, fd32_* for Float32 (eltype(x) in the reference)
    fdtype = _norm_fdtype(fdtype)
    o = _opts(fdtype, col_window, x_window, scratch_bytes, color_range, eps_contiguous, fingerprint,
              (store_csc or store_csc_always or store_rows) and not complex_x and isinstance(J, SparseMatrixCSC),
              store_csc_always and not complex_x and isinstance(J, SparseMatrixCSC),
              store_rows and not complex_x and isinstance(J, SparseMatrixCSC))
    if complex_x:      # returntype <: Complex with forward / central differences: the library lowers it (FD_PLAN_COMPLEX_X)
        o.flags |= _l.PLAN_COMPLEX_X
    if isinstance(J, DevicePatternCSC):
        if not (sparsity is J or sparsity is None):
            raise ValueError("a DevicePatternCSC is its own sparsity pattern")
        if complex_x:
            raise NotImplementedError("complex-valued x with a device-resident pattern")
        return make_plan_csc_device(J.m, J.n, J.colptr, J.rowval, colorvec, fdtype, ctx, col_window, x_window, J.idx_base, dtype,
                                    fingerprint=fingerprint, store_csc=store_csc)
    h = C.c_void_p()
    cv = _i64(colorvec)
    if isinstance(J, SparseMatrixCSC) and isinstance(sparsity, SparseMatrixCSC):
        m, n = J.size()
        if cv.size != n:
            raise ValueError("DimensionMismatch: length(colorvec) != length(x)")  # src/jacobians.jl:516
        common = (J is sparsity) or (np.array_equal(J.colptr, sparsity.colptr) and np.array_equal(J.rowval, sparsity.rowval))
        if common:  # ext/FiniteDiffSparseArraysExt.jl:51-52
            _l.check(L.fd_plan_create_csc(ctx.handle, m, n, _vp(J.colptr), _vp(J.rowval), 8, 1, _vp(cv), 8,
                                          C.byref(o), C.byref(h)))
        else:  # J[row,col] = v into J's own pattern (ext/FiniteDiffSparseArraysExt.jl:20-28)
            from .patterns import csc_cols
            cols = csc_cols(sparsity.colptr)
            dest = _csc_positions(J, sparsity.rowval, cols)
            _l.check(L.fd_plan_create_entries(ctx.handle, m, n, _vp(sparsity.rowval), _vp(cols), _vp(dest), dest.size,
                                              J.rowval.size, 8, 1, _vp(cv), 8, C.byref(o), C.byref(h)))
    elif isinstance(J, Tridiagonal):
        n = J.size()[0]
        if cv.size != n:
            raise ValueError("DimensionMismatch: length(colorvec) != length(x)")
        _l.check(L.fd_plan_create_tridiagonal(ctx.handle, n, _vp(cv), 8, C.byref(o), C.byref(h)))
    elif isinstance(J, BandedMatrix):
        n = cv.size
        _l.check(L.fd_plan_create_banded(ctx.handle, J.m, n, J.l, J.u, _vp(cv), 8, C.byref(o), C.byref(h)))
    elif isinstance(J, BandedBlockBandedMatrix):
        lay = J.layout
        if cv.size != lay.N:
            raise ValueError("DimensionMismatch: length(colorvec) != length(x)")
        if J.as_entries or col_window is not None or (o.flags & 2):
            rows, cols, dest = lay.entries()
            rows, cols, dest = _i64(rows), _i64(cols), _i64(dest)
            _l.check(L.fd_plan_create_entries(ctx.handle, lay.N, lay.N, _vp(rows), _vp(cols), _vp(dest), dest.size,
                                              lay.data_len, 8, 1, _vp(cv), 8, C.byref(o), C.byref(h)))
        else:
            bs, st, sr = _i64(lay.blk_sizes), _i64(lay.block_starts), _i64(lay.block_strides)
            _l.check(L.fd_plan_create_bandedblockbanded(ctx.handle, lay.nblk, _vp(bs), lay.bl, lay.bu, lay.lam, lay.mu, _vp(st), _vp(sr), lay.data_len, 8, 1,
                                                        _vp(cv), 8, C.byref(o), C.byref(h)))
    elif isinstance(J, BlockBandedMatrix):
        lay = J.layout
        bs, st, sr = _i64(lay.blk_sizes), _i64(lay.block_starts), _i64(lay.block_strides)
        _l.check(L.fd_plan_create_blockbanded(ctx.handle, lay.nblk, _vp(bs), lay.bl, lay.bu, _vp(st), _vp(sr), 8, 1,
                                              _vp(cv), 8, C.byref(o), C.byref(h)))
    else:  # dense J
        m, n = J.shape
        if cv.size != n:
            raise ValueError("DimensionMismatch: length(colorvec) != length(x)")
        if isinstance(sparsity, SparseMatrixCSC):
            _l.check(L.fd_plan_create_csc_dense(ctx.handle, m, n, _vp(sparsity.colptr), _vp(sparsity.rowval), 8, 1,
                                                _vp(cv), 8, C.byref(o), C.byref(h)))
        elif sparsity is not None:  # dense matrix pattern: _findstructralnz, src/jacobians.jl:473-488
            rows, cols = _findstructralnz(np.asarray(sparsity))
            _l.check(L.fd_plan_create_coo_dense(ctx.handle, m, n, _vp(rows), _vp(cols), rows.size, 8, 1, _vp(cv), 8,
                                                C.byref(o), C.byref(h)))
        else:  # sparsity === nothing: colour index == column index (src/jacobians.jl:548-557)
            ncols = int(cv.max()) if cv.size else 0
            if ncols > n:
                raise IndexError("BoundsError: maximum(colorvec) > length(x)")
            _l.check(L.fd_plan_create_dense(ctx.handle, m, n, ncols, C.byref(o), C.byref(h)))
    return Plan(ctx, h, fdtype, dtype, cx=complex_x)


def make_plan_csc_device(M, N, colptr, rowval, colorvec, fdtype, ctx=None, col_window=None, x_window=None,
                         idx_base=1, dtype=np.float64, fingerprint=False, store_csc=False):
    """fd_plan_create_csc_device: the common-pattern CSC plan from a pattern that already lives on the device --
    `colptr`, `rowval`, `colorvec` are torch CUDA tensors (int32 or int64; colptr / rowval `idx_base`-based, colours 1..C).
    The plan is compiled by kernels; nothing crosses PCIe."""
    import torch
    ctx = ctx or Context.default()
    L = _l.typed(ctx.L, dtype)
    fdtype = _norm_fdtype(fdtype)
    o = _opts(fdtype, col_window, x_window, fingerprint=fingerprint, store_csc=store_csc)
    for t, what in ((colptr, "colptr"), (rowval, "rowval"), (colorvec, "colorvec")):
        if not (_is_torch(t) and t.is_cuda and t.is_contiguous() and t.dtype in (torch.int32, torch.int64)):
            raise TypeError("%s must be a contiguous int32 / int64 CUDA tensor" % what)
    if colptr.dtype != rowval.dtype:
        raise TypeError("colptr and rowval must have the same integer type")
    if colorvec.numel() != N:
        raise ValueError("DimensionMismatch: length(colorvec) != length(x)")
    h = C.c_void_p()
    _l.check(L.fd_plan_create_csc_device(ctx.handle, int(M), int(N), colptr.data_ptr(), rowval.data_ptr(), colptr.element_size(),
                                         int(idx_base), colorvec.data_ptr(), colorvec.element_size(), C.byref(o), C.byref(h)))
    return Plan(ctx, h, fdtype, dtype)


def _findstructralnz(A):
    """src/jacobians.jl:473-488: (I, J) of A != 0 in column-major order, 1-based."""
    cols, rows = np.nonzero(np.asarray(A).T)
    return _i64(rows + 1), _i64(cols + 1)


def _csc_positions(J, rows, cols):
    """0-based positions in J.nzval of entries (rows, cols) (1-based); KeyError-like if absent."""
    dest = np.empty(rows.size, np.int64)
    for k in range(rows.size):
        a, b = J.colptr[cols[k] - 1] - 1, J.colptr[cols[k]] - 1
        q = a + np.searchsorted(J.rowval[a:b], rows[k])
        if q >= b or J.rowval[q] != rows[k]:
            raise IndexError("J has no stored entry (%d,%d)" % (rows[k], cols[k]))
        dest[k] = q
    return dest


def _outs_of(J):
    if isinstance(J, (SparseMatrixCSC, DevicePatternCSC)):
        return [J.nzval]
    if isinstance(J, Tridiagonal):
        return [J.dl, J.d, J.du]
    if isinstance(J, (BandedMatrix, BlockBandedMatrix, BandedBlockBandedMatrix)):
        return [J.data]
    return [J]


# ----------------------------------------------------------------------------------------------
# the reference-facing API
# ----------------------------------------------------------------------------------------------
class JacobianCache:
    """FiniteDiff.JacobianCache (src/jacobians.jl:1-128).

    JacobianCache(x, fdtype="forward", returntype=float; colorvec=1:length(x), sparsity=None)
    JacobianCache(x, fx, fdtype, ...)            allocating, non-square
    JacobianCache(x1, fx, fx1, fdtype, ...)      non-allocating

    The device implementation keeps its perturbed points / values in the plan's scratch, so x1,
    fx, fx1 are carried only for interface fidelity (their contents never influence a result --
    the property test/cache_reuse_tests.jl pins).
    """

    def __init__(self, x1, fx=None, fx1=None, fdtype="forward", returntype=np.float64, *, colorvec=None,
                 sparsity=None):
        if isinstance(fx, str):  # JacobianCache(x, fdtype, ...)
            fdtype, fx = fx, None
        if isinstance(fx1, str):  # JacobianCache(x, fx, fdtype, ...)
            fdtype, fx1 = fx1, None
        if isinstance(fx1, (type, np.dtype)):  # JacobianCache(x, fdtype, returntype): the reference's positional order
            returntype, fx1 = fx1, None
        self.fdtype = _norm_fdtype(fdtype)
        self.returntype = returntype
        if self.fdtype == "complex" and np.dtype(returntype).kind == "c":
            # fdtype_error(returntype), src/jacobians.jl:106
            raise ValueError("Unrecognized fdtype: valid values are Val{:forward} or Val{:central}.")
        # complex-valued x / f with forward or central differences (src/jacobians.jl:94-128, 537-622; src/epsilons.jl:26-29 with
        # abs of a complex number): the reference's generic loop -- masked norm over complex elements, REAL step on the real
        # parts, complex quotient.  The library lowers it to the real problem on (re, im) pairs (FD_PLAN_COMPLEX_X).
        self.cx = _complex_base(x1) is not None
        if np.dtype(returntype).kind == "c" and not self.cx:
            raise TypeError("a complex returntype needs a complex x (eltype(fx) == returntype, src/jacobians.jl:118-119)")
        self.x1 = x1
        self.x2 = None
        self.fx = fx if fx is not None else x1
        self.fx1 = None if self.fdtype == "complex" else (fx1 if fx1 is not None else self.fx)
        n = int(np.prod(x1.shape))
        self.colorvec = np.arange(1, n + 1, dtype=np.int64) if colorvec is None else colorvec
        self.sparsity = sparsity
        self.dtype = _dtype_of(x1) or _complex_base(x1) or np.dtype(np.float64)      # (real) eltype: selects the fd_* / fd32_* instantiation
        self._plans = {}
        self._bound = {}
        self.lazy = True       # built-in f! families: install their lazy-point launcher on new plans (the shim's install_lazy!)
        # How a cached call finds out that `colorvec` / `sparsity` changed (the reference re-reads both on every call,
        # src/jacobians.jl:512-513):  "identity" -- O(1): the plan is keyed on the identity (object, data pointer,
        # length) of the arrays; a new array object / a resize makes a new plan, an IN-PLACE edit needs `invalidate()`;
        # "content" (default since round 5: the reference's semantics -- a silently stale plan is a wrong Jacobian) -- identity,
        # then fd_plan_matches compares the arrays' content with what the plan was compiled from
        # (kernels for device arrays, host threads for host arrays: the reference's semantics, at the reference's O(N + nnz));
        # "content_async" -- device arrays: the same comparison as ONE fused kernel ahead of the Jacobian, nothing copied back, the
        # stream never stopped; an edit is reported one call late (the stale plan's call is recomputed by the call that finds out)
        # or by Context.synchronize() -- FD_ERR_STALE.  Host arrays fall back to "content".
        # "auto" (default since round 6): device arrays -> "content_async", whose kernel now runs on a side stream BESIDE the Jacobian;
        # host arrays -> "content", which for TrackedCSC / TrackedVector holders is a comparison of generation counters (the arrays are
        # re-hashed only after an edit).
        self.pattern_check = "auto"

    def invalidate(self):
        """Forget the compiled plans: the next call re-reads `colorvec` / `sparsity` (after an in-place edit of either)."""
        self._plans.clear()
        self._bound.clear()

    @staticmethod
    def _ident(a):
        """O(1) identity of a pattern / colour holder: which object, where its data lies, how long it is."""
        if a is None:
            return None
        if isinstance(a, (SparseMatrixCSC, DevicePatternCSC)):
            return (id(a), JacobianCache._ident(a.colptr), JacobianCache._ident(a.rowval), a.size())
        if isinstance(a, TrackedVector):
            return (id(a), a.data.ctypes.data, a.data.size)
        if isinstance(a, np.ndarray):
            return (id(a), a.ctypes.data, a.size)
        if _is_torch(a):
            return (id(a), a.data_ptr(), a.numel())
        if isinstance(a, range):
            return ("range", a.start, a.stop, a.step)
        return (id(a),)

    def _plan_for(self, J, sparsity, colorvec, ctx, f=None):
        """The plan of (J's type and shape, sparsity, colorvec, fdtype) -- the sequence of the Julia shim's `plan_for`
        (julia/FiniteDiffMI355X.jl) and of examples/c_abi_clients.c::client_dropin: an O(1) identity lookup, an optional content
        check, a new plan (+ the built-in family's lazy launcher, as `install_lazy!` does) only when something changed."""
        jshape = tuple(J.shape) if (isinstance(J, np.ndarray) or _is_torch(J)) else tuple(J.size())
        key = (type(J).__name__, jshape, self.fdtype, self._ident(sparsity), self._ident(colorvec), id(f) if self.lazy else None)
        mode = self.pattern_check
        if mode == "auto":
            mode = "content_async" if self._all_device(sparsity, colorvec) else "content"
        content = mode in ("content", "content_async")
        deferred = mode == "content_async" and self._all_device(sparsity, colorvec)
        gen = self._generations(sparsity, colorvec)      # not None: every host array involved owns its mutation (Tracked*)
        ent = self._plans.get(key)
        if ent is not None and ent[3] == content:
            if not content:
                return ent[0]
            if deferred:
                if not ent[0].stale():           # (the verdict of the check enqueued by the call before, if it has run)
                    self._content_matches(ent[0], sparsity, colorvec, deferred=True)
                    return ent[0]
            elif gen is not None and ent[4] == gen:
                return ent[0]                    # nothing was edited since the content was last compared
            elif self._content_matches(ent[0], sparsity, colorvec):
                self._plans[key] = ent[:4] + (gen,)
                return ent[0]
        self._bound.clear()
        if len(self._plans) >= 8:
            self._plans.clear()
        want_csc = (self.lazy and isinstance(f, (BuiltinF, JitF)) and not self.cx and (f.lazy_caps & _l.LAZY_CAP_STORE_CSC) != 0
                    and (self.fdtype != "complex" or (f.lazy_caps & _l.LAZY_CAP_STORE_CSC_COMPLEX) != 0))      # (the shim: PlanOpts(...; store_csc = f can store column by column))
        plan = make_plan(J, sparsity, colorvec, self.fdtype, ctx, dtype=self.dtype, complex_x=self.cx, fingerprint=content,
                         store_csc=want_csc, store_csc_always=want_csc and isinstance(f, JitF),
                         store_rows=want_csc and getattr(f, "family", None) == "sparse")      # (the separable built-in family: row lists -> the row-wise store)
        if self.lazy and isinstance(f, (BuiltinF, JitF)) and not self.cx and f.lazy_fn is not None:
            plan.set_lazy(f)          # built-in families: f! perturbs while loading / stores the Jacobian itself (shim: install_lazy!)
        self._plans[key] = (plan, sparsity, colorvec, content, gen)     # (the arrays are kept alive: their ids stay theirs)
        return plan

    @staticmethod
    def _generations(sparsity, colorvec):
        """(generation of the pattern, generation of the colours) when every HOST array of the pair is a Tracked* holder (a device array
        or a structural pattern has none to offer: None for it), else None -- then only the content tells."""
        gs = []
        for a in (sparsity, colorvec):
            if isinstance(a, (TrackedCSC, TrackedVector)):
                gs.append(a.generation)
            elif a is None or isinstance(a, range) or isinstance(a, (Tridiagonal, BandedMatrix, BlockBandedMatrix, BandedBlockBandedMatrix)):
                gs.append(None)
            else:
                return None
        return tuple(gs)

    @staticmethod
    def _all_device(sparsity, colorvec):
        return (_is_torch(colorvec) and colorvec.is_cuda and
                (isinstance(sparsity, DevicePatternCSC) or not isinstance(sparsity, (SparseMatrixCSC, np.ndarray))))

    @staticmethod
    def _content_matches(plan, sparsity, colorvec, deferred=False):
        if isinstance(colorvec, TrackedVector):
            colorvec = colorvec.data
        cv = colorvec if (_is_torch(colorvec) or isinstance(colorvec, np.ndarray)) else _i64(colorvec)
        if isinstance(sparsity, DevicePatternCSC):
            return plan.matches(sparsity.colptr, sparsity.rowval, cv, idx_base=sparsity.idx_base, deferred=deferred)
        if isinstance(sparsity, SparseMatrixCSC):
            return plan.matches(sparsity.colptr, sparsity.rowval, cv, idx_base=1)
        return plan.matches(None, None, cv, deferred=deferred)      # structural patterns (Tridiagonal, BandedMatrix, ...): the colours


def _has_sparsestruct(J):
    """ArrayInterface.has_sparsestruct for the holders above."""
    return isinstance(J, (SparseMatrixCSC, DevicePatternCSC, Tridiagonal, BandedMatrix, BlockBandedMatrix, BandedBlockBandedMatrix))


def finite_difference_jacobian_b(J, f, x, cache_or_fdtype="forward", returntype=np.float64, f_in=None, *,
                                 relstep=None, absstep=None, colorvec=None, sparsity="default", dir=True,
                                 ctx=None):
    """``FiniteDiff.finite_difference_jacobian!`` for the coloured path.

    Cache-less form (src/jacobians.jl:446-471):
        finite_difference_jacobian_b(J, f, x, fdtype="forward", returntype, f_in; relstep, absstep,
                                     colorvec=1:length(x), sparsity=has_sparsestruct(J) ? J : nothing)
    Cached form (src/jacobians.jl:504-514):
        finite_difference_jacobian_b(J, f, x, cache, f_in; relstep, absstep, colorvec=cache.colorvec,
                                     sparsity=cache.sparsity, dir=True)
    f is a ``BuiltinF`` or ``TorchF`` launcher.  Returns None; fills J's own storage.
    """
    if isinstance(cache_or_fdtype, JacobianCache):
        cache = cache_or_fdtype
        if isinstance(sparsity, str) and sparsity == "default":
            sparsity = cache.sparsity
        if colorvec is None:
            colorvec = cache.colorvec
    else:
        fdtype = _norm_fdtype(cache_or_fdtype)
        n = int(np.prod(x.shape))
        if colorvec is None:
            colorvec = np.arange(1, n + 1, dtype=np.int64)
        if isinstance(sparsity, str) and sparsity == "default":
            sparsity = J if _has_sparsestruct(J) else None
        cache = JacobianCache(x, fdtype, returntype, colorvec=colorvec, sparsity=sparsity)
    if sparsity is None and _has_sparsestruct(J):
        sparsity = J
    # relstep / absstep None -> default_relstep(fdtype, eltype(x)) and absstep = relstep, resolved by the library
    plan = cache._plan_for(J, sparsity, colorvec, ctx or getattr(f, "ctx", None), f)
    cache.last_plan = plan
    outs = _outs_of(J)
    fin = f_in if cache.fdtype == "forward" else None
    # Device arrays: the call only ENQUEUES on the context's stream (fd_jacobian_async, as the shim's AMDGPU methods do) through a
    # callable whose pointers were resolved once per (plan, f, arrays, steps) -- one foreign call per Jacobian.
    if not plan.cx and _is_torch(x) and x.is_cuda and all(_is_torch(o) and o.is_cuda for o in outs) and \
            (fin is None or (_is_torch(fin) and fin.is_cuda)):
        bkey = (id(plan), id(f), x.data_ptr(), tuple(o.data_ptr() for o in outs), None if fin is None else fin.data_ptr(),
                relstep, absstep, dir)
        call = cache._bound.get(bkey)
        if call is None:
            try:
                call = plan.bind(f, x, outs, fin, relstep, absstep, dir)
            except (TypeError, ValueError):
                call = None                 # (non-contiguous / mismatched arrays: the general path below reports or stages them)
            if call is not None:
                if len(cache._bound) >= 16:
                    cache._bound.clear()
                cache._bound[bkey] = call
        if call is not None:
            call()
            if isinstance(J, _ThreeDiagonals):
                J._finish(colorvec)
            return None
    staged = None
    if not isinstance(J, (SparseMatrixCSC, DevicePatternCSC, Tridiagonal, BandedMatrix, BlockBandedMatrix, BandedBlockBandedMatrix)):
        # dense J must be column-major for the library; stage a C-order numpy array
        if isinstance(J, np.ndarray) and not J.flags.f_contiguous:
            staged = np.zeros(J.shape, dtype=J.dtype, order="F")
            outs = [staged]
    plan.jacobian(f, x, outs, f_in=fin, relstep=relstep, absstep=absstep, dir=dir)
    if staged is not None:
        J[...] = staged
    if isinstance(J, _ThreeDiagonals):
        J._finish(colorvec)
    return None


class JVPCache:
    """FiniteDiff.JVPCache (src/jvp.jl:1-60): JVPCache(x, fdtype="forward") / JVPCache(x, fx1, fdtype)."""

    def __init__(self, x1, fx1=None, fdtype="forward", lazy=True, quotient=True):
        if isinstance(fx1, str):
            fdtype, fx1 = fx1, None
        self.fdtype = _norm_fdtype(fdtype)
        self.x1, self.fx1 = x1, (fx1 if fx1 is not None else x1)
        self.dtype = _dtype_of(x1) or np.dtype(np.float64)
        self._plan = None
        self.lazy = bool(lazy)     # use f!'s lazy-point JVP launcher when it has one (same bits, fewer passes)
        self.quotient = bool(quotient)   # ... and let it write the finished quotient (FD_LAZY_JVP_CAP_QUOTIENT)
        self._lazy_keep = None

    def _plan_for(self, M, N, ctx):
        if self.fdtype == "complex":
            raise ValueError("finite_difference_jvp doesn't support :complex-mode finite diff")  # src/jvp.jl:248-250
        key = (M, N, id(ctx))
        if self._plan is None or self._plan[0] != key:
            h = C.c_void_p()
            Lt = _l.typed(ctx.L, self.dtype)
            _l.check(Lt.fd_jvp_plan_create(ctx.handle, M, N, _l.FDTYPES[self.fdtype], C.byref(h)))
            fin = weakref.finalize(self, Lt.fd_jvp_plan_destroy, h)
            self._plan = (key, h, fin, ctx)
        return self._plan[1]


def finite_difference_jvp_b(jvp, f, x, v, cache=None, f_in=None, *, relstep=None, absstep=None, dir=True, ctx=None,
                            sync=True):
    """``FiniteDiff.finite_difference_jvp!(jvp, f, x, v, cache, f_in; relstep, absstep, dir)`` (src/jvp.jl:238-274);
    cache may be a JVPCache or an fdtype name (cache-less form).  Fills jvp, returns None.
    sync=False (device arrays only) only enqueues on the context's stream (fd_jvp_async)."""
    if not isinstance(cache, JVPCache):
        cache = JVPCache(x, "forward" if cache is None else cache)
    ctx = ctx or getattr(f, "ctx", None) or Context.default()
    M, N = int(np.prod(jvp.shape)), int(np.prod(x.shape))
    h = cache._plan_for(M, N, ctx)
    Lt = _l.typed(ctx.L, cache.dtype)
    if np.dtype(getattr(f, "dtype", cache.dtype)) != cache.dtype:
        raise TypeError("f! launcher is built for %s, the cache for %s" % (np.dtype(f.dtype).name, cache.dtype.name))
    lz = getattr(f, "lazy_jvp_fn", None) if cache.lazy else None
    cache._lazy_keep = lz            # (the ctypes function object must outlive the call)
    _l.check(Lt.fd_jvp_plan_set_lazy_f(h, lz if lz is not None else _l.F_LAUNCH_LAZY_JVP()))
    if lz is not None:
        _l.check(Lt.fd_jvp_plan_set_lazy_caps(h, int(getattr(f, "lazy_jvp_caps", 0)) if cache.quotient else 0))
    xp, xk, _a = _ptr(x, "x", cache.dtype)
    vp_, vk, _b = _ptr(v, "v", cache.dtype)
    if xk != vk:
        raise ValueError("x and v must both be host or both be device arrays")
    op, ok, _c = _ptr(jvp, "jvp", cache.dtype)
    fp, fk = None, _l.DEVICE
    if f_in is not None and cache.fdtype == "forward":
        fp, fk, _d = _ptr(f_in, "f_in", cache.dtype)
    if not sync:
        if xk != _l.DEVICE or ok != _l.DEVICE or (fp is not None and fk != _l.DEVICE):
            raise ValueError("the async path needs device arrays")
        rc = Lt.fd_jvp_async(h, f.fn, f.fctx, xp, vp_, fp, -1.0 if relstep is None else float(relstep),
                                -1.0 if absstep is None else float(absstep), float(dir), op)
    else:
        rc = Lt.fd_jvp(h, f.fn, f.fctx, xp, vp_, xk, fp, fk, -1.0 if relstep is None else float(relstep),
                          -1.0 if absstep is None else float(absstep), float(dir), op, ok)
    err = getattr(f, "error", None)
    if err is not None:
        f.error = None
        raise err
    _l.check(rc)
    if not sync:
        return None
    e = C.c_double()
    _l.check(Lt.fd_jvp_get_epsilon(h, C.byref(e)))
    cache.last_epsilon = e.value
    return None


def matrix_colors(A):
    """Colour vector for a sparsity pattern -- the role ``ArrayInterface.matrix_colors`` plays in the
    reference's tests (test/coloring_tests.jl:112,117).  SparseMatrixCSC: greedy column colouring
    (fd_color_columns_greedy); Tridiagonal / BandedMatrix: closed form mod1(j, l+u+1);
    BlockBandedMatrix: the layout's block colouring.  Host-side; needs no GPU."""
    L = _l.load()
    nc = C.c_int64()
    if isinstance(A, SparseMatrixCSC):
        out = np.empty(A.n, np.int64)
        _l.check(L.fd_color_columns_greedy(A.m, A.n, _vp(A.colptr), _vp(A.rowval), 8, 1,
                                           out.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nc)))
        return out
    if isinstance(A, Tridiagonal):
        n = A.size()[0]
        out = np.empty(n, np.int64)
        _l.check(L.fd_color_banded(n, 1, 1, out.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nc)))
        return out
    if isinstance(A, BandedMatrix):
        n = int(A.data.shape[1])
        out = np.empty(n, np.int64)
        _l.check(L.fd_color_banded(n, A.l, A.u, out.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nc)))
        return out
    if isinstance(A, BlockBandedMatrix):
        return A.layout.colors()
    raise TypeError("matrix_colors: unsupported matrix type %r" % type(A).__name__)


def finite_difference_jacobian(f, x, cache_or_fdtype="forward", returntype=np.float64, f_in=None, *, M=None,
                               relstep=None, absstep=None, colorvec=None, sparsity=None, jac_prototype=None,
                               dir=True, ctx=None):
    """Out-of-place ``FiniteDiff.finite_difference_jacobian(f, x, ...; sparsity, jac_prototype, colorvec)``
    (src/jacobians.jl:240-259, 277-429): allocates J like ``jac_prototype`` (else dense ``size(sparsity)``,
    else dense M x maximum(colorvec)) on x's device and fills it through the in-place device path -- the
    reference builds J by summing per-colour ``_make_Ji`` matrices, which yields the same entries.
    f is a device launcher (``BuiltinF`` / ``TorchF``) with the in-place signature; M = length(f(x)) when
    it cannot be inferred from jac_prototype / sparsity (defaults to length(x))."""
    n = int(np.prod(x.shape))
    if isinstance(cache_or_fdtype, JacobianCache):
        if colorvec is None:
            colorvec = cache_or_fdtype.colorvec
        if sparsity is None:
            sparsity = cache_or_fdtype.sparsity
    if colorvec is None:
        colorvec = np.arange(1, n + 1, dtype=np.int64)
    proto = jac_prototype   # only the prototype fixes J's type; `sparsity` alone gives a dense zeros(size(sparsity))
    if isinstance(proto, SparseMatrixCSC):
        J = proto.similar(like=x)
    elif isinstance(proto, Tridiagonal):
        J = Tridiagonal(_similar(x, n - 1), _similar(x, n), _similar(x, n - 1))
    elif isinstance(proto, BandedMatrix):
        w = proto.l + proto.u + 1
        J = BandedMatrix(_similar(x, w * n).reshape(n, w).T if _is_torch(x) else np.zeros((w, n), dtype=x.dtype, order="F"),
                         proto.m, proto.l, proto.u)
    elif isinstance(proto, BlockBandedMatrix):
        J = BlockBandedMatrix(_similar(x, proto.layout.data_len), proto.layout)
    else:
        if proto is not None:
            m, ncol = np.asarray(proto).shape
        elif sparsity is not None:
            m, ncol = sparsity.size() if hasattr(sparsity, "size") and callable(sparsity.size) else np.asarray(sparsity).shape
        else:
            m, ncol = (n if M is None else int(M)), int(np.max(colorvec))
        if _is_torch(x):
            import torch
            J = torch.zeros((ncol, m), dtype=x.dtype, device=x.device).t()   # column-major
        else:
            J = np.zeros((m, ncol), dtype=x.dtype, order="F")
    sp = sparsity if sparsity is not None else (J if _has_sparsestruct(J) else None)
    if isinstance(J, SparseMatrixCSC) and isinstance(sp, SparseMatrixCSC) and sp is not J:
        sp = J if (np.array_equal(sp.colptr, J.colptr) and np.array_equal(sp.rowval, J.rowval)) else sp
    finite_difference_jacobian_b(J, f, x, cache_or_fdtype, returntype, f_in, relstep=relstep, absstep=absstep,
                                 colorvec=colorvec, sparsity=sp if sp is not None else "default", dir=dir, ctx=ctx)
    return J
