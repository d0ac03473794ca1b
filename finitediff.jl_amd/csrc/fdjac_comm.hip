// Multi-GPU exchange steps of the coloured-Jacobian path behind the C ABI (include/fdjac.h, "multi-GPU").
//
// One process per GPU.  The reference is single-process, so there is no reference counterpart: this is the
// north-star's "single RCCL gather over xGMI to assemble nzval" as plain C entry points, so that a Julia
// process-per-GPU caller (MPI.jl only to ship the 128-byte id) needs neither torch nor a binding of RCCL.
//
// RCCL is loaded at run time (dlopen), not linked: libfdjac loads on boxes without RCCL, and inside a host framework
// that already carries its own librccl (PyTorch bundles one) the SAME library instance is used (RTLD_NOLOAD first),
// never a second copy with its own state.  Every collective is enqueued on the context's stream; nothing
// synchronises.  xGMI is point-to-point (7 links per GPU): the assembly calls move each rank's slice exactly once
// per destination, with no staging copy (in-place all-gather / grouped send-recv straight out of the buffer the
// decompression kernel wrote).
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// a build box without the RCCL headers: the handful of declarations the dlsym'd entry points need (RCCL's public ABI;
// nothing here is linked -- the library is bound at run time or fd_comm_* returns FD_ERR_COMM)
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
ncclResult_t ncclGetVersion(int *version);
ncclResult_t ncclGetUniqueId(ncclUniqueId *uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
const char *ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
}
#endif

#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "fdjac_internal.h"

#ifndef FDJAC_F32   /* element-type independent: compiled once */

struct fd_comm {
    fd_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    fd_p2p *p2p = nullptr;        // fd_comm_enable_p2p: small messages go through the peer-to-peer mailbox (fdjac_p2p.hip), owned by the communicator
    char *d_small = nullptr;      // nranks x 64 bytes of device scratch (>= 64), allocated with the communicator: the collectives of the attach-time
                                  // agreements never allocate, so no rank can drop out of one between its peers' calls
};
extern "C" int64_t fdjac_p2p_slot_bytes(const fd_p2p *p);

namespace fdjac {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    char where[160] = "";
};

static Rccl g_rccl;
static std::mutex g_rccl_mutex;

// returns nullptr (and sets the error text) when RCCL cannot be found
static const Rccl *rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    const char *env = getenv("FDJAC_RCCL_LIB");
    const char *names[] = {env && *env ? env : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    // the instance the process already has (a host framework's), else load one
    for (const char *n : names)
        if (!h && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) snprintf(g_rccl.where, sizeof(g_rccl.where), "%s (already loaded)", n);
    for (const char *n : names)
        if (!h && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) snprintf(g_rccl.where, sizeof(g_rccl.where), "%s", n);
    if (!h) {
        set_error("RCCL not found (tried librccl.so.1, librccl.so, /opt/rocm/lib; set FDJAC_RCCL_LIB): %s", dlerror());
        return nullptr;
    }
    Rccl r;
    r.handle = h;
    memcpy(r.where, g_rccl.where, sizeof(r.where));
#define FD_SYM(field, name)                                                     \
    r.field = (decltype(r.field))dlsym(h, name);                                \
    if (!r.field) {                                                             \
        set_error("RCCL symbol %s missing in %s", name, r.where);              \
        return nullptr;                                                         \
    }
    FD_SYM(GetVersion, "ncclGetVersion")
    FD_SYM(GetUniqueId, "ncclGetUniqueId")
    FD_SYM(CommInitRank, "ncclCommInitRank")
    FD_SYM(CommDestroy, "ncclCommDestroy")
    FD_SYM(GetErrorString, "ncclGetErrorString")
    FD_SYM(AllGather, "ncclAllGather")
    FD_SYM(AllReduce, "ncclAllReduce")
    FD_SYM(Broadcast, "ncclBroadcast")
    FD_SYM(Send, "ncclSend")
    FD_SYM(Recv, "ncclRecv")
    FD_SYM(GroupStart, "ncclGroupStart")
    FD_SYM(GroupEnd, "ncclGroupEnd")
#undef FD_SYM
    g_rccl = r;
    return &g_rccl;
}

#define FD_NCCL_CHECK(R, expr)                                                                                  \
    do {                                                                                                        \
        ncclResult_t _r = (expr);                                                                               \
        if (_r != ncclSuccess) {                                                                                \
            set_error("%s failed: %s (%s:%d)", #expr, (R)->GetErrorString(_r), __FILE__, __LINE__);              \
            return FD_ERR_COMM;                                                                                 \
        }                                                                                                       \
    } while (0)

static bool dtype_of(int elem_bytes, ncclDataType_t *dt)
{
    if (elem_bytes == 8) *dt = ncclFloat64;
    else if (elem_bytes == 4) *dt = ncclFloat32;
    else if (elem_bytes == 1) *dt = ncclUint8;
    else return false;
    return true;
}

}  // namespace fdjac

using namespace fdjac;

extern "C" {

int fd_comm_unique_id(void *id_out)
{
    FD_REQUIRE(id_out != nullptr, FD_ERR_ARG, "id_out is NULL");
    static_assert(sizeof(ncclUniqueId) == FD_COMM_ID_BYTES, "fd_comm id size");
    const Rccl *R = rccl();
    if (!R) return FD_ERR_COMM;
    ncclUniqueId id;
    FD_NCCL_CHECK(R, R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return FD_OK;
}

int fd_comm_create(fd_ctx *ctx, int nranks, int rank, const void *id, fd_comm **out)
{
    FD_REQUIRE(ctx && id && out, FD_ERR_ARG, "NULL argument");
    *out = nullptr;
    FD_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, FD_ERR_ARG, "rank %d outside [0,%d)", rank, nranks);
    const Rccl *R = rccl();
    if (!R) return FD_ERR_COMM;
    FD_HIP_CHECK(hipSetDevice(ctx->device));
    fd_comm *c = new (std::nothrow) fd_comm();
    FD_REQUIRE(c != nullptr, FD_ERR_NOMEM, "out of host memory");
    c->ctx = ctx;
    c->nranks = nranks;
    c->rank = rank;
    {
        const hipError_t e = hipMalloc((void **)&c->d_small, (size_t)(nranks > 1 ? nranks : 1) * FD_P2P_HANDLE_BYTES);
        if (e != hipSuccess) { set_error("fd_comm_create: hipMalloc failed: %s", hipGetErrorString(e)); delete c; return FD_ERR_HIP; }
    }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = R->CommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(nranks=%d, rank=%d) failed: %s", nranks, rank, R->GetErrorString(r));
        (void)hipFree(c->d_small);
        delete c;
        return FD_ERR_COMM;
    }
    *out = c;
    return FD_OK;
}

int fd_comm_destroy(fd_comm *c)
{
    if (!c) return FD_OK;
    const Rccl *R = rccl();
    if (R && c->comm) {
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
        (void)R->CommDestroy(c->comm);
    }
    if (c->p2p) (void)fd_p2p_destroy(c->p2p);
    if (c->d_small) (void)hipFree(c->d_small);
    delete c;
    return FD_OK;
}

int fd_comm_enable_p2p(fd_comm *c, int64_t slot_bytes)
{
    FD_REQUIRE(c != nullptr, FD_ERR_ARG, "comm is NULL");
    FD_REQUIRE(c->p2p == nullptr, FD_ERR_ARG, "the communicator has a mailbox already");
    const Rccl *R = rccl();
    if (!R) return FD_ERR_COMM;      // (a communicator exists only where RCCL was found: every rank or none)
    // collective: a rank whose local set-up fails -- selecting the device, creating its mailbox -- still takes part in the two
    // exchanges below (carrying its failure in rc), so that no rank is left waiting inside RCCL and every rank falls back together;
    // nothing is allocated on the way (the communicator's own scratch carries the handles)
    fd_p2p *p = nullptr;
    int rc = FD_OK;
    if (hipSetDevice(c->ctx->device) != hipSuccess) { set_error("fd_comm_enable_p2p: hipSetDevice(%d) failed", c->ctx->device); rc = FD_ERR_HIP; }
    if (!rc) rc = fd_p2p_create(c->ctx, c->nranks, c->rank, slot_bytes, &p);
    char first_error[512];
    snprintf(first_error, sizeof first_error, "%s", rc ? fd_last_error() : "");
    // the handles travel over the communicator itself: one small in-place all-gather
    char *d_h = c->d_small;
    std::vector<char> h((size_t)c->nranks * FD_P2P_HANDLE_BYTES, 0);
    if (!rc) rc = fd_p2p_local_handle(p, h.data() + (size_t)c->rank * FD_P2P_HANDLE_BYTES);
    hipError_t e = hipMemcpyAsync(d_h, h.data(), h.size(), hipMemcpyHostToDevice, c->ctx->stream);
    {
        const ncclResult_t r = R->AllGather(d_h + (size_t)c->rank * FD_P2P_HANDLE_BYTES, d_h, FD_P2P_HANDLE_BYTES, ncclUint8, c->comm, c->ctx->stream);
        if (r != ncclSuccess && !rc) { set_error("exchanging the mailbox handles failed: %s", R->GetErrorString(r)); rc = FD_ERR_COMM; }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_h, h.size(), hipMemcpyDeviceToHost, c->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->ctx->stream);
    if (e != hipSuccess && !rc) { set_error("exchanging the mailbox handles failed: %s", hipGetErrorString(e)); rc = FD_ERR_HIP; }
    if (!rc) rc = fd_p2p_connect(p, h.data());
    if (rc && !first_error[0]) snprintf(first_error, sizeof first_error, "%s", fd_last_error());
    // every rank must have mapped every peer before anybody stores into a mailbox: agree on success (a rank that failed makes all fall back)
    double mine[4] = {rc ? 1.0 : 0.0, 0, 0, 0}, got[4] = {0, 0, 0, 0};
    const int rc2 = fdjac_comm_allreduce_max4(c, mine, got);
    if (rc || rc2 || got[0] != 0.0) {
        if (p) (void)fd_p2p_destroy(p);
        if (rc) { set_error("%s", first_error); return rc; }
        if (!rc2) { set_error("another rank could not map the mailboxes: the communicator keeps using RCCL for small messages"); return FD_ERR_COMM; }
        return rc2;
    }
    c->p2p = p;
    return FD_OK;
}

// Back to RCCL for the small messages (every rank calls it, e.g. after fd_comm_p2p_status reported a timed-out wait: peer-to-peer stores that
// do not arrive must not cost a timeout per step).  Plans attached to the communicator follow at their next call.
int fd_comm_disable_p2p(fd_comm *c)
{
    FD_REQUIRE(c != nullptr, FD_ERR_ARG, "comm is NULL");
    if (c->p2p) {
        (void)fd_p2p_destroy(c->p2p);
        c->p2p = nullptr;
    }
    return FD_OK;
}

int fd_comm_p2p_status(const fd_comm *c, int *enabled, int *timed_out_rank_plus_1)
{
    FD_REQUIRE(c != nullptr, FD_ERR_ARG, "comm is NULL");
    if (enabled) *enabled = c->p2p ? 1 : 0;
    if (timed_out_rank_plus_1) {
        *timed_out_rank_plus_1 = 0;
        if (c->p2p) return fd_p2p_status(c->p2p, timed_out_rank_plus_1);
    }
    return FD_OK;
}

int fd_comm_info(const fd_comm *c, int *nranks, int *rank, int *rccl_version)
{
    FD_REQUIRE(c != nullptr, FD_ERR_ARG, "comm is NULL");
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    if (rccl_version) {
        const Rccl *R = rccl();
        if (!R) return FD_ERR_COMM;
        FD_NCCL_CHECK(R, R->GetVersion(rccl_version));
    }
    return FD_OK;
}

const char *fd_comm_library(void)
{
    const Rccl *R = rccl();
    return R ? R->where : "";
}

int fd_comm_allgather(fd_comm *c, void *buf, int64_t slot_elems, int elem_bytes)
{
    FD_REQUIRE(c && (buf || slot_elems == 0), FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(slot_elems >= 0, FD_ERR_ARG, "slot_elems < 0");
    ncclDataType_t dt;
    FD_REQUIRE(dtype_of(elem_bytes, &dt), FD_ERR_ARG, "elem_bytes must be 1, 4 or 8");
    if (slot_elems == 0) return FD_OK;   // (a single-rank communicator still goes through RCCL: same call path)
    if (c->p2p && c->nranks > 1 && slot_elems * elem_bytes <= fdjac_p2p_slot_bytes(c->p2p) && (slot_elems * elem_bytes) % 8 == 0)
        return fd_p2p_allgather(c->p2p, buf, slot_elems * elem_bytes);      // small message: direct peer-to-peer stores
    const Rccl *R = rccl();
    if (!R) return FD_ERR_COMM;
    FD_HIP_CHECK(hipSetDevice(c->ctx->device));
    // in place: sendbuff == recvbuff + rank * sendcount (ncclAllGather's documented in-place form)
    char *base = (char *)buf;
    FD_NCCL_CHECK(R, R->AllGather(base + (size_t)c->rank * (size_t)slot_elems * (size_t)elem_bytes, base, (size_t)slot_elems, dt,
                                 c->comm, c->ctx->stream));
    return FD_OK;
}

int fd_comm_gatherv(fd_comm *c, const void *send, int64_t send_elems, void *recv, const int64_t *counts,
                    const int64_t *displs, int elem_bytes, int root)
{
    FD_REQUIRE(c && counts && displs, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(root >= 0 && root < c->nranks, FD_ERR_ARG, "root %d outside [0,%d)", root, c->nranks);
    ncclDataType_t dt;
    FD_REQUIRE(dtype_of(elem_bytes, &dt), FD_ERR_ARG, "elem_bytes must be 1, 4 or 8");
    FD_REQUIRE(send_elems == counts[c->rank], FD_ERR_ARG, "send_elems %lld != counts[rank] %lld", (long long)send_elems,
               (long long)counts[c->rank]);
    FD_REQUIRE(c->rank != root || recv != nullptr, FD_ERR_ARG, "recv is NULL on the root");
    const Rccl *R = rccl();          // (before anything is enqueued: without RCCL nothing of the call happens)
    if (!R) return FD_ERR_COMM;
    FD_HIP_CHECK(hipSetDevice(c->ctx->device));
    hipStream_t s = c->ctx->stream;
    char *rb = (char *)recv;
    if (c->rank == root && send_elems > 0 && send != (const void *)(rb + (size_t)displs[root] * (size_t)elem_bytes))
        FD_HIP_CHECK(hipMemcpyAsync(rb + (size_t)displs[root] * (size_t)elem_bytes, send, (size_t)send_elems * (size_t)elem_bytes,
                                    hipMemcpyDeviceToDevice, s));
    // one group: every slice rides its own xGMI link into the root (7 concurrent point-to-point transfers at 8 ranks).
    // The group is ALWAYS closed: a failing Send / Recv is remembered, ncclGroupEnd still runs (an open group would swallow
    // every later collective of this thread), then the first error is reported.
    FD_NCCL_CHECK(R, R->GroupStart());
    ncclResult_t first = ncclSuccess;
    if (c->rank == root) {
        for (int r = 0; r < c->nranks && first == ncclSuccess; ++r)
            if (r != root && counts[r] > 0)
                first = R->Recv(rb + (size_t)displs[r] * (size_t)elem_bytes, (size_t)counts[r], dt, r, c->comm, s);
    } else if (send_elems > 0) {
        first = R->Send(send, (size_t)send_elems, dt, root, c->comm, s);
    }
    const ncclResult_t ge = R->GroupEnd();
    if (first == ncclSuccess) first = ge;
    if (first != ncclSuccess) {
        set_error("fd_comm_gatherv: RCCL point-to-point group failed: %s", R->GetErrorString(first));
        return FD_ERR_COMM;
    }
    return FD_OK;
}

int fd_comm_halo_exchange(fd_comm *c, void *buf, int64_t own_begin, int64_t own_end, int64_t halo, int elem_bytes)
{
    FD_REQUIRE(c && buf, FD_ERR_ARG, "NULL argument");
    ncclDataType_t dt;
    FD_REQUIRE(dtype_of(elem_bytes, &dt), FD_ERR_ARG, "elem_bytes must be 1, 4 or 8");
    FD_REQUIRE(halo >= 0 && own_begin >= 0 && own_end >= own_begin, FD_ERR_ARG, "bad range [%lld,%lld) / halo %lld", (long long)own_begin,
               (long long)own_end, (long long)halo);
    FD_REQUIRE(halo == 0 || own_end - own_begin >= halo, FD_ERR_ARG, "this rank owns fewer than `halo` = %lld elements", (long long)halo);
    if (halo == 0 || c->nranks == 1) return FD_OK;
    // (every per-rank check happens BEFORE the routing decision, so that all ranks take the same path: a check that fails on one rank
    //  only must not leave the others waiting in the other transport)
    const bool lo = c->rank > 0, hi = c->rank + 1 < c->nranks;
    FD_REQUIRE(!lo || own_begin >= halo, FD_ERR_ARG, "no room for the lower halo below element %lld", (long long)own_begin);
    if (c->p2p && (elem_bytes == 4 || elem_bytes == 8) && 2 * halo * elem_bytes <= fdjac_p2p_slot_bytes(c->p2p) && (halo * elem_bytes) % 8 == 0)
        return fd_p2p_halo_exchange(c->p2p, buf, own_begin, own_end, halo, elem_bytes);
    const Rccl *R = rccl();
    if (!R) return FD_ERR_COMM;
    FD_HIP_CHECK(hipSetDevice(c->ctx->device));
    hipStream_t s = c->ctx->stream;
    char *b = (char *)buf;
    const size_t eb = (size_t)elem_bytes, h = (size_t)halo;
    // one group of <= 4 point-to-point transfers, each over the link to a neighbour; always closed (see fd_comm_gatherv)
    FD_NCCL_CHECK(R, R->GroupStart());
    ncclResult_t first = ncclSuccess;
    if (lo && first == ncclSuccess) first = R->Send(b + (size_t)own_begin * eb, h, dt, c->rank - 1, c->comm, s);
    if (hi && first == ncclSuccess) first = R->Send(b + ((size_t)own_end - h) * eb, h, dt, c->rank + 1, c->comm, s);
    if (lo && first == ncclSuccess) first = R->Recv(b + ((size_t)own_begin - h) * eb, h, dt, c->rank - 1, c->comm, s);
    if (hi && first == ncclSuccess) first = R->Recv(b + (size_t)own_end * eb, h, dt, c->rank + 1, c->comm, s);
    const ncclResult_t ge = R->GroupEnd();
    if (first == ncclSuccess) first = ge;
    if (first != ncclSuccess) {
        set_error("fd_comm_halo_exchange: RCCL point-to-point group failed: %s", R->GetErrorString(first));
        return FD_ERR_COMM;
    }
    return FD_OK;
}

int fd_comm_allreduce_sum(fd_comm *c, void *buf, int64_t n, int elem_bytes)
{
    FD_REQUIRE(c && (buf || n == 0), FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(elem_bytes == 8 || elem_bytes == 4, FD_ERR_ARG, "elem_bytes must be 4 or 8");
    if (n == 0) return FD_OK;
    const Rccl *R = rccl();
    if (!R) return FD_ERR_COMM;
    FD_HIP_CHECK(hipSetDevice(c->ctx->device));
    FD_NCCL_CHECK(R, R->AllReduce(buf, buf, (size_t)n, elem_bytes == 8 ? ncclFloat64 : ncclFloat32, ncclSum, c->comm,
                                 c->ctx->stream));
    return FD_OK;
}

int fd_comm_broadcast(fd_comm *c, void *buf, int64_t n, int elem_bytes, int root)
{
    FD_REQUIRE(c && (buf || n == 0), FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(root >= 0 && root < c->nranks, FD_ERR_ARG, "root %d outside [0,%d)", root, c->nranks);
    ncclDataType_t dt;
    FD_REQUIRE(dtype_of(elem_bytes, &dt), FD_ERR_ARG, "elem_bytes must be 1, 4 or 8");
    if (n == 0) return FD_OK;
    const Rccl *R = rccl();
    if (!R) return FD_ERR_COMM;
    FD_HIP_CHECK(hipSetDevice(c->ctx->device));
    FD_NCCL_CHECK(R, R->Broadcast(buf, buf, (size_t)n, dt, root, c->comm, c->ctx->stream));
    return FD_OK;
}

}  // extern "C"

// used by the sharded step-size reduction of both element-type builds (fdjac_api.hip); not part of the public ABI
extern "C" {
int fdjac_comm_allgather_f64(fd_comm *c, double *buf, int64_t slot_elems) { return fd_comm_allgather(c, buf, slot_elems, 8); }
int fdjac_comm_allreduce_max4(fd_comm *c, const double *mine, double *out)
{
    FD_REQUIRE(c && mine && out, FD_ERR_ARG, "NULL argument");
    const Rccl *R = rccl();
    if (!R) return FD_ERR_COMM;
    // (no early return and no allocation before the collective: every rank enters it, whatever happened to it locally)
    int rc = FD_OK;
    if (hipSetDevice(c->ctx->device) != hipSuccess) rc = FD_ERR_HIP;
    double *d = reinterpret_cast<double *>(c->d_small);
    if (hipMemcpyAsync(d, mine, 4 * sizeof(double), hipMemcpyHostToDevice, c->ctx->stream) != hipSuccess) rc = FD_ERR_HIP;
    {
        const ncclResult_t r = R->AllReduce(d, d, 4, ncclFloat64, ncclMax, c->comm, c->ctx->stream);
        if (r != ncclSuccess && !rc) { set_error("ncclAllReduce(max) failed: %s", R->GetErrorString(r)); rc = FD_ERR_COMM; }
    }
    if (!rc && (hipMemcpyAsync(out, d, 4 * sizeof(double), hipMemcpyDeviceToHost, c->ctx->stream) != hipSuccess ||
                hipStreamSynchronize(c->ctx->stream) != hipSuccess)) { set_error("fd_plan_set_comm: reading the grid check back failed"); rc = FD_ERR_HIP; }
    return rc;
}
int fdjac_comm_nranks(const fd_comm *c) { return c ? c->nranks : 1; }
int fdjac_comm_rank(const fd_comm *c) { return c ? c->rank : 0; }
const fd_ctx *fdjac_comm_ctx(const fd_comm *c) { return c ? c->ctx : nullptr; }
fd_p2p *fdjac_comm_p2p(const fd_comm *c) { return c ? c->p2p : nullptr; }
}

#endif /* FDJAC_F32 */
