// Small-message exchange by direct peer-to-peer stores (include/fdjac.h, fd_p2p_*).
//
// A Newton / Rosenbrock step on P GPUs exchanges, per Jacobian: the halo of x (2 (l + u) values per link), the partial sums of the
// step-size reduction (62 KB in total) and the interface packets of the sharded tridiagonal solve (64 B per rank).  As RCCL
// collectives these are three launches of a general-purpose machine (proxy thread, channel setup, flag protocol) for a few
// kilobytes: 10-25 us each on comparable parts, against 6.5 us of sharded compute.  xGMI is a load / store
// fabric: a kernel can write a peer's HBM directly.  So every rank owns a MAILBOX in its HBM, maps the mailboxes of all peers
// (hipIpc handles, exchanged once by whatever the host has -- or by RCCL itself, fd_comm_enable_p2p), and an exchange is
//     put    per peer, one workgroup per 8 KB share of this rank's slot copies it into the peer's mailbox and fences at system
//            scope; the last share to land raises this rank's flag in the peer's mailbox to the exchange's epoch (a release
//            store at system scope);
//     wait   the same workgroup then polls that peer's flag in the LOCAL mailbox (system-scope acquire loads, bounded by a
//            wall-clock timeout that raises an error word instead of hanging the GPU) and copies its share of the peer's slot out.
// ONE kernel on the caller's stream (k_p2p_exchange; as two launches an exchange took 2 us longer,
// profiles/r04_zzz_p2p_mailbox_latency.md), no host involvement, no proxy.  Slots are double-buffered by epoch parity: a rank can only
// reach exchange e + 2 after every peer has released e + 1, i.e. after it has finished reading e (stream order on the peer).
// RCCL stays for what it is good at: the bulk assembly of nzval (fd_comm_gatherv / fd_comm_allgather of MBs).
//
// What can be executed in the build environment: two PROCESSES sharing one GPU (tests/test_gpu_multigpu.py::test_p2p_*): the IPC
// mapping, the epoch / parity protocol, timeouts.  Ordering of remote stores over xGMI between two devices cannot (one GPU per
// box): the code uses the documented recipe -- data stores, __threadfence_system(), flag store with system-scope release; the
// reader acquires at system scope and reads the payload with system-scope loads (no stale L2 lines) -- and times out loudly.
#include "fdjac_internal.h"

#include <cstring>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#ifndef FDJAC_F32   /* element-type independent: compiled once */

constexpr int kP2PMaxRanks = 64;
constexpr int64_t kP2PFlagStride = 128;      // one line per sender
// Fine-grained ("uncached") device allocations are never handed back to the runtime's allocator: on ROCm 7.0 / MI355X memory that was
// once allocated with hipDeviceMallocUncached, freed and then recycled into ORDINARY allocations misbehaved -- kernels reading freshly
// uploaded arrays there saw other data (plan builds failing their consistency checks, wrong group sums; gone with plain hipMalloc mailboxes --
// profiles/NOTES.md, round 6).  A mailbox that is destroyed parks its block here and the next mailbox of that size takes it.
static std::mutex g_unc_mutex;
static std::vector<std::pair<size_t, void *>> g_unc_pool;
static void *unc_take(size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_unc_mutex);
    for (size_t k = 0; k < g_unc_pool.size(); ++k)
        if (g_unc_pool[k].first == bytes) {
            void *m = g_unc_pool[k].second;
            g_unc_pool.erase(g_unc_pool.begin() + (ptrdiff_t)k);
            return m;
        }
    return nullptr;
}
static void unc_park(size_t bytes, void *mem)
{
    std::lock_guard<std::mutex> lock(g_unc_mutex);
    g_unc_pool.emplace_back(bytes, mem);
}

static void fz_region_image(std::vector<unsigned long long> &img)
{
    img.assign((size_t)(kFzBufs * kFzBufBytes / 8), kFzSentinel64);
    for (int q = 0; q < kFzBufs; ++q)
        for (int64_t k = kFzGsumBytes / 8; k < kFzBufBytes / 8; ++k) img[(size_t)(q * kFzBufBytes / 8 + k)] = kFzSentinelHalo64;
}

struct fd_p2p {
    fd_ctx *ctx = nullptr;
    int nranks = 1, rank = 0;
    int64_t slot_bytes = 0;                   // capacity per rank per exchange (a multiple of 128)
    char *local = nullptr;                    // this rank's mailbox, TWO channels (0: all-gathers -- every rank waits for every rank; 1: halo
                                              // exchanges -- neighbours only; each with its own epochs, so that the parity argument holds per
                                              // channel): [flags: nranks x 128 B][parity 0: nranks slots][parity 1: nranks slots] each
    char *peer[kP2PMaxRanks] = {};            // the mailboxes as mapped here (peer[rank] == local)
    bool mapped[kP2PMaxRanks] = {};
    char **d_peer = nullptr;                  // device copy of peer[]
    double *d_scratch = nullptr;              // nranks x 8 doubles: the attach-time agreement of the plans (fd_plan_set_p2p) is exchanged through it
    unsigned *d_arrived = nullptr;            // per-peer share counters of a put split over several workgroups (zero between launches);
                                              // [kP2PMaxRanks]: the arrival ticket of the step exchange's workgroups
    int *d_err = nullptr;                     // device error word (pinned host memory mapped to the device: readable without a sync)
    int *h_err = nullptr;
    uint64_t epoch[2] = {0, 0};
    bool connected = false;
    bool uncached = false;
    char *sink = nullptr;                     // loop-back mailbox (fd_p2p_create_loopback): what every peer's mailbox is mapped to
    int64_t fz_off = 0;                       // the FUSED step's region of a mailbox (after the two channels): kFzBufs buffers of
                                              // [8 colours x 64 groups group sums][lower halo][upper halo], every cell its own flag
    uint64_t fz_epoch = 0;                    // fused steps enqueued so far (buffer = epoch mod 3)
    size_t local_bytes = 0;                   // size of `local` (an uncached block goes back to the pool, not to the allocator)
    bool shared_device = false;               // some peer's mailbox lies on THIS device (ranks sharing a GPU: tests, dry runs) -- a launch that
                                              // fills the device with wavefronts waiting for a peer would keep that peer from running
};

namespace fdjac {

// Both copies keep FOUR 8-byte system-scope accesses per lane in flight (one round trip through fine-grained memory per 8 KB of
// a workgroup's share instead of one per 2 KB: the 62 KB step-size partials took 31 dependent round trips before).
__device__ __forceinline__ void p2p_copy_out(char *dst, const char *src, int64_t bytes)
{
    // payload written by a peer: read it at system scope (never from a stale L2 line), 8 bytes per lane
    const int64_t step = (int64_t)blockDim.x * 8;
    int64_t o = (int64_t)threadIdx.x * 8;
    for (; o + 3 * step < bytes; o += 4 * step) {
        unsigned long long v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __hip_atomic_load((const unsigned long long *)(src + o + u * step), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int u = 0; u < 4; ++u) *(unsigned long long *)(dst + o + u * step) = v[u];
    }
    for (; o < bytes; o += step)
        *(unsigned long long *)(dst + o) = __hip_atomic_load((const unsigned long long *)(src + o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void p2p_copy_in(char *dst, const char *src, int64_t bytes)
{
    const int64_t step = (int64_t)blockDim.x * 8;
    int64_t o = (int64_t)threadIdx.x * 8;
    for (; o + 3 * step < bytes; o += 4 * step) {
        unsigned long long v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const unsigned long long *)(src + o + u * step);
#pragma unroll
        for (int u = 0; u < 4; ++u) __hip_atomic_store((unsigned long long *)(dst + o + u * step), v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (; o < bytes; o += step)
        __hip_atomic_store((unsigned long long *)(dst + o), *(const unsigned long long *)(src + o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// workgroup `part` of `parts` serving one peer copies [lo, hi) of the payload (8-byte granules)
__device__ __forceinline__ void p2p_part(int64_t bytes, int part, int parts, int64_t &lo, int64_t &hi)
{
    const int64_t per = ((bytes / 8 + parts - 1) / parts) * 8;
    lo = (int64_t)part * per < bytes ? (int64_t)part * per : bytes;
    hi = lo + per < bytes ? lo + per : bytes;
}

// targets[b]: the peer workgroup b writes to; src_off[b] / dst_sub[b]: where in `buf` its payload starts and at which byte of this
// rank's slot in the peer's mailbox it lands
struct P2PPut {
    int target[4];
    int64_t src_off[4], dst_sub[4], bytes[4];
    int n;
};
__device__ __forceinline__ void p2p_put_part(char *const *__restrict__ peer, const char *__restrict__ buf, P2PPut put, int nranks, int rank,
                                                    int64_t slot_bytes, uint64_t epoch, int all, int64_t chan_off, int parts, unsigned *__restrict__ arrived)
{
    // all != 0: the all-gather form -- workgroups b*parts .. b*parts + parts-1 serve peer b (b != rank), payload = this rank's slot
    // of buf, split into `parts` shares; the LAST share to land raises the flag (arrived[b]: a counter in this rank's own memory)
    const int b = (int)blockIdx.x / parts, part = (int)blockIdx.x % parts;
    int target;
    int64_t src_off, dst_sub, bytes;
    if (all) {
        target = b;
        if (target == rank) return;
        src_off = put.src_off[0];
        dst_sub = 0;
        bytes = put.bytes[0];
    } else {
        if (b >= put.n) return;
        target = put.target[b];
        src_off = put.src_off[b];
        dst_sub = put.dst_sub[b];
        bytes = put.bytes[b];
    }
    char *mb = peer[target] + chan_off;
    char *slot = mb + (int64_t)nranks * kP2PFlagStride + ((int64_t)(epoch & 1) * nranks + rank) * slot_bytes + dst_sub;
    int64_t lo, hi;
    p2p_part(bytes, part, parts, lo, hi);
    p2p_copy_in(slot + lo, buf + src_off + lo, hi - lo);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        bool last = true;
        if (parts > 1) {
            last = __hip_atomic_fetch_add(arrived + b, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(parts - 1);
            if (last) {
                __hip_atomic_store(arrived + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (launches on the stream are ordered)
                __threadfence_system();
            }
        }
        if (last)              // this rank's flag in the peer's mailbox: "my slot of exchange `epoch` is complete"
            __hip_atomic_store((unsigned long long *)(mb + (int64_t)rank * kP2PFlagStride), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

struct P2PGet {
    int sender[4];
    int64_t dst_off[4], src_sub[4], bytes[4];
    int n;
};
__device__ __forceinline__ void p2p_wait_part(char *__restrict__ local_base, char *__restrict__ buf, P2PGet get, int nranks, int rank, int64_t slot_bytes,
                                                     uint64_t epoch, int all, int64_t timeout_ticks, int *__restrict__ err, int64_t chan_off, int parts)
{
    // `parts` workgroups per sender: each polls the sender's flag itself, then copies its share of the slot
    char *local = local_base + chan_off;
    const int b = (int)blockIdx.x / parts, part = (int)blockIdx.x % parts;
    int sender;
    int64_t dst_off, src_sub, bytes;
    if (all) {
        sender = b;
        if (sender == rank) return;
        dst_off = (int64_t)sender * get.bytes[0];
        src_sub = 0;
        bytes = get.bytes[0];
    } else {
        if (b >= get.n) return;
        sender = get.sender[b];
        dst_off = get.dst_off[b];
        src_sub = get.src_sub[b];
        bytes = get.bytes[b];
    }
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        const unsigned long long *flag = (const unsigned long long *)(local + (int64_t)sender * kP2PFlagStride);
        const long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
            if (wall_clock64() - t0 > timeout_ticks) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) __hip_atomic_store(err, 1 + sender, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // "rank `sender` never arrived"
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    const char *slot = local + (int64_t)nranks * kP2PFlagStride + ((int64_t)(epoch & 1) * nranks + sender) * slot_bytes + src_sub;
    int64_t lo, hi;
    p2p_part(bytes, part, parts, lo, hi);
    p2p_copy_out(buf + dst_off + lo, slot + lo, hi - lo);
}

// One exchange = ONE launch: every workgroup first delivers its share to its peer, then waits for the same peer's share of the
// same exchange.  No workgroup waits for another workgroup of its own launch, and a peer's puts never wait for anything, so the
// order is free of cycles whatever the residency.
__global__ void __launch_bounds__(kBlock) k_p2p_exchange(char *const *__restrict__ peer, char *__restrict__ local_base, char *__restrict__ buf, P2PPut put,
                                                         P2PGet get, int nranks, int rank, int64_t slot_bytes, uint64_t epoch, int all, int64_t chan_off,
                                                         int parts, unsigned *__restrict__ arrived, int64_t timeout_ticks, int *__restrict__ err)
{
    p2p_put_part(peer, buf, put, nranks, rank, slot_bytes, epoch, all, chan_off, parts, arrived);
    p2p_wait_part(local_base, buf, get, nranks, rank, slot_bytes, epoch, all, timeout_ticks, err, chan_off, parts);
}

// ---- the per-step exchange of a sharded Jacobian call: ONE launch (fdjac_p2p_step) ---------------------------------------------
// Workgroup b serves peer b: it stores this rank's group sums of the step-size reduction -- and, if b is a neighbour, the halo of x
// b needs -- into b's mailbox (one slot, one flag), then waits for b's slot of the same exchange and copies b's group sums (and
// halo) out.  The last workgroup to finish (agent-scope ticket; the group sums were copied out with agent-scope stores and are
// read with agent-scope loads) adds the 64 group sums in group order and writes the step sizes: level 2 of the reduction as
// k_eps_partial_reg defines it -- the bits of the unsharded call.  All-to-all on channel 0, so the parity argument of the all-gather
// holds.  A rank's own group sums come from the launch before (stream order).
struct P2PStep {
    char *x;                    // NULL: no halo
    int64_t own_begin, own_end, halo;
    int elem_bytes;
    char *gsum;                 // nranks slots of gs_bytes
    int64_t gs_bytes;
    fdjac_eps_final fin;
};
__device__ __forceinline__ void p2p_copy_out_agent(char *dst, const char *src, int64_t bytes)
{
    for (int64_t o = (int64_t)threadIdx.x * 8; o < bytes; o += (int64_t)blockDim.x * 8)
        __hip_atomic_store((unsigned long long *)(dst + o),
                           __hip_atomic_load((const unsigned long long *)(src + o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void __launch_bounds__(kBlock) k_p2p_step(char *const *__restrict__ peer, char *__restrict__ local, P2PStep st, int nranks, int rank,
                                                     int64_t slot_bytes, uint64_t epoch, unsigned *__restrict__ ticket, int64_t timeout_ticks,
                                                     int *__restrict__ err)
{
    const int b = (int)blockIdx.x;
    const int64_t hb = st.x ? st.halo * st.elem_bytes : 0;
    const bool below = st.x && b == rank - 1, above = st.x && b == rank + 1;
    __shared__ int s_ok;
    if (b != rank) {
        char *mb = peer[b];
        char *slot = mb + (int64_t)nranks * kP2PFlagStride + ((int64_t)(epoch & 1) * nranks + rank) * slot_bytes;
        p2p_copy_in(slot, st.gsum + (int64_t)rank * st.gs_bytes, st.gs_bytes);
        if (below) p2p_copy_in(slot + st.gs_bytes, st.x + st.own_begin * st.elem_bytes, hb);               // my first elements: b's upper halo
        if (above) p2p_copy_in(slot + st.gs_bytes, st.x + (st.own_end - st.halo) * st.elem_bytes, hb);     // my last elements: b's lower halo
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store((unsigned long long *)(mb + (int64_t)rank * kP2PFlagStride), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long *flag = (const unsigned long long *)(local + (int64_t)b * kP2PFlagStride);
            const long long t0 = wall_clock64();
            int ok = 1;
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
                if (wall_clock64() - t0 > timeout_ticks) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!ok) __hip_atomic_store(err, 1 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // "rank b never arrived"
            s_ok = ok;
        }
        __syncthreads();
        if (s_ok) {
            const char *in = local + (int64_t)nranks * kP2PFlagStride + ((int64_t)(epoch & 1) * nranks + b) * slot_bytes;
            p2p_copy_out_agent(st.gsum + (int64_t)b * st.gs_bytes, in, st.gs_bytes);
            if (below) p2p_copy_out(st.x + (st.own_begin - st.halo) * st.elem_bytes, in + st.gs_bytes, hb);
            if (above) p2p_copy_out(st.x + st.own_end * st.elem_bytes, in + st.gs_bytes, hb);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this workgroup's copies have left the CU before its ticket is drawn
    __syncthreads();
    if (threadIdx.x >= 64) return;
    unsigned t = 0;
    if (threadIdx.x == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
    if (t != (unsigned)(nranks - 1)) return;
    const fdjac_eps_final &f = st.fin;
    const int c = threadIdx.x;
    // (all addends loaded together, a few per lane, parked in LDS; lane c then adds its colour's 64 group sums in group order)
    __shared__ double stage[kEpsGroups * kRegColors];
    const int cnt = f.ngroups * f.ldp;                         // 64 x 8
    double v[kRegColors];
#pragma unroll
    for (int u = 0; u < kRegColors; ++u) v[u] = (u * 64 + c < cnt) ? __hip_atomic_load(f.gsum + u * 64 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
    for (int u = 0; u < kRegColors; ++u) stage[u * 64 + c] = v[u];
    __builtin_amdgcn_wave_barrier();
    if (c < f.C) {
        double tot = 0.0;
        for (int g = 0; g < f.ngroups; ++g) tot += stage[g * f.ldp + c];
        if (f.elem_bytes == 4) {
            const float e = eps_rule<float>(tot, f.relstep, f.absstep, f.dir, f.is_forward);
            ((float *)f.eps)[c] = e;
            if (f.eps2) ((float *)f.eps2)[c] = 2.0f * e;
        } else {
            const double e = eps_rule<double>(tot, f.relstep, f.absstep, f.dir, f.is_forward);
            ((double *)f.eps)[c] = e;
            if (f.eps2) ((double *)f.eps2)[c] = 2.0 * e;
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (launches on the stream are ordered)
}

}  // namespace fdjac

using namespace fdjac;

extern "C" {

int fd_p2p_create(fd_ctx *ctx, int nranks, int rank, int64_t slot_bytes, fd_p2p **out)
{
    FD_REQUIRE(ctx && out, FD_ERR_ARG, "NULL argument");
    *out = nullptr;
    FD_REQUIRE(nranks >= 1 && nranks <= kP2PMaxRanks && rank >= 0 && rank < nranks, FD_ERR_ARG, "rank %d of %d (at most %d ranks)", rank, nranks, kP2PMaxRanks);
    FD_REQUIRE(slot_bytes >= 8 && slot_bytes <= ((int64_t)1 << 24), FD_ERR_ARG, "slot_bytes must be in 8 .. 16 MiB (small messages only)");
    FD_HIP_CHECK(hipSetDevice(ctx->device));
    fd_p2p *p = new (std::nothrow) fd_p2p();
    FD_REQUIRE(p != nullptr, FD_ERR_NOMEM, "out of host memory");
    p->ctx = ctx;
    p->nranks = nranks;
    p->rank = rank;
    p->slot_bytes = (slot_bytes + 127) / 128 * 128;
    p->fz_off = 2 * ((int64_t)nranks * kP2PFlagStride + 2 * (int64_t)nranks * p->slot_bytes);                     // two channels,
    const size_t bytes = (size_t)p->fz_off + (size_t)(kFzBufs * kFzBufBytes);                                           // then the fused step's cells
    // fine-grained (uncached) device memory if the runtime shares it between processes, else plain device memory (all accesses to the
    // mailbox are system-scope atomics either way)
    void *mem = nullptr;
    p->local_bytes = bytes;
    if ((mem = unc_take(bytes)) != nullptr) {
        p->uncached = true;
    } else if (hipExtMallocWithFlags(&mem, bytes, hipDeviceMallocUncached) == hipSuccess) {
        hipIpcMemHandle_t probe;
        if (hipIpcGetMemHandle(&probe, mem) == hipSuccess) p->uncached = true;
        else { unc_park(bytes, mem); mem = nullptr; }
    }
    (void)hipGetLastError();
    if (!mem) {
        hipError_t e = hipMalloc(&mem, bytes);
        if (e != hipSuccess) { set_error("hipMalloc of the mailbox failed: %s", hipGetErrorString(e)); delete p; return FD_ERR_HIP; }
    }
    p->local = (char *)mem;
    p->peer[rank] = p->local;
    hipError_t e = hipMemset(p->local, 0, bytes);       // (blocking, on the null stream: complete before the handle is published)
    if (e == hipSuccess) {
        std::vector<unsigned long long> img;
        fz_region_image(img);
        e = hipMemcpy(p->local + p->fz_off, img.data(), img.size() * 8, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipHostMalloc((void **)&p->h_err, sizeof(int), hipHostMallocMapped);
    if (e == hipSuccess) { *p->h_err = 0; e = hipHostGetDevicePointer((void **)&p->d_err, p->h_err, 0); }
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_peer, sizeof(char *) * kP2PMaxRanks);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_scratch, sizeof(double) * 8 * (size_t)nranks);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_arrived, sizeof(unsigned) * (kP2PMaxRanks + 1));
    if (e == hipSuccess) e = hipMemset(p->d_arrived, 0, sizeof(unsigned) * (kP2PMaxRanks + 1));
    if (e != hipSuccess) {
        set_error("setting up the mailbox failed: %s", hipGetErrorString(e));
        if (p->h_err) (void)hipHostFree(p->h_err);
        if (p->d_peer) (void)hipFree(p->d_peer);
        if (p->d_arrived) (void)hipFree(p->d_arrived);
        if (p->d_scratch) (void)hipFree(p->d_scratch);
        if (p->uncached) unc_park(p->local_bytes, p->local); else (void)hipFree(p->local);
        delete p;
        return FD_ERR_HIP;
    }
    if (nranks == 1) {
        (void)hipMemcpy(p->d_peer, p->peer, sizeof(char *) * kP2PMaxRanks, hipMemcpyHostToDevice);
        p->connected = true;
    }
    *out = p;
    return FD_OK;
}

int fd_p2p_local_handle(fd_p2p *p, void *handle_out)
{
    FD_REQUIRE(p && handle_out, FD_ERR_ARG, "NULL argument");
    static_assert(sizeof(hipIpcMemHandle_t) == FD_P2P_HANDLE_BYTES, "fd_p2p handle size");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    hipIpcMemHandle_t h;
    FD_HIP_CHECK(hipIpcGetMemHandle(&h, p->local));
    memcpy(handle_out, &h, sizeof h);
    return FD_OK;
}

int fd_p2p_connect(fd_p2p *p, const void *handles)
{
    FD_REQUIRE(p && handles, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(!p->connected || p->nranks == 1, FD_ERR_ARG, "already connected");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    for (int r = 0; r < p->nranks; ++r) {
        if (r == p->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + (size_t)r * FD_P2P_HANDLE_BYTES, sizeof h);
        void *ptr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            set_error("hipIpcOpenMemHandle of rank %d's mailbox failed: %s (one process per GPU on ONE node; HSA_ENABLE_IPC_MODE_LEGACY=0)", r,
                      hipGetErrorString(e));
            return FD_ERR_COMM;
        }
        p->peer[r] = (char *)ptr;
        p->mapped[r] = true;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, ptr) == hipSuccess && at.device == p->ctx->device) p->shared_device = true;
        (void)hipGetLastError();
    }
    FD_HIP_CHECK(hipMemcpy(p->d_peer, p->peer, sizeof(char *) * kP2PMaxRanks, hipMemcpyHostToDevice));
    p->connected = true;
    return FD_OK;
}

// A mailbox whose peers are all THIS device (rank-share measurements and tests on one GPU): every peer's mailbox is
// mapped to one local sink, the flags of every sender stand at the largest epoch (no wait ever spins) and the senders' slots hold what
// fd_p2p_loopback_fill put there -- rank `rank` of `nranks` then runs its step exactly as in the W-rank job: the same launches, the
// same stores (into local HBM instead of across xGMI), the same copies out of its own mailbox.
int fd_p2p_create_loopback(fd_ctx *ctx, int nranks, int rank, int64_t slot_bytes, fd_p2p **out)
{
    int rc = fd_p2p_create(ctx, nranks, rank, slot_bytes, out);
    if (rc) return rc;
    fd_p2p *p = *out;
    const size_t chan = (size_t)nranks * kP2PFlagStride + 2 * (size_t)nranks * (size_t)p->slot_bytes;
    hipError_t e = hipMalloc((void **)&p->sink, 2 * chan + (size_t)(kFzBufs * kFzBufBytes));
    if (e == hipSuccess) e = hipMemset(p->sink, 0, 2 * chan + (size_t)(kFzBufs * kFzBufBytes));
    std::vector<unsigned long long> flags((size_t)nranks * (kP2PFlagStride / 8), ~0ull);
    for (int ch = 0; ch < 2 && e == hipSuccess; ++ch)
        e = hipMemcpy(p->local + ch * chan, flags.data(), flags.size() * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("setting up the loop-back mailbox failed: %s", hipGetErrorString(e));
        (void)fd_p2p_destroy(p);
        *out = nullptr;
        return FD_ERR_HIP;
    }
    for (int r = 0; r < nranks; ++r) p->peer[r] = r == rank ? p->local : p->sink;
    FD_HIP_CHECK(hipMemcpy(p->d_peer, p->peer, sizeof(char *) * kP2PMaxRanks, hipMemcpyHostToDevice));
    FD_HIP_CHECK(hipDeviceSynchronize());
    p->connected = true;
    return FD_OK;
}

// what `sender` would have delivered: `bytes` bytes (device or host memory) at byte `offset` of its slot, both parities of the
// all-gather / step channel
int fd_p2p_loopback_fill(fd_p2p *p, int sender, int64_t offset, const void *data, int64_t bytes)
{
    FD_REQUIRE(p && p->sink, FD_ERR_ARG, "not a loop-back mailbox");
    FD_REQUIRE(sender >= 0 && sender < p->nranks && sender != p->rank, FD_ERR_ARG, "sender %d", sender);
    FD_REQUIRE(data && offset >= 0 && bytes >= 0 && offset + bytes <= p->slot_bytes, FD_ERR_ARG, "offset %lld + %lld bytes do not fit the slot (%lld)",
               (long long)offset, (long long)bytes, (long long)p->slot_bytes);
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    FD_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    for (int par = 0; par < 2; ++par) {
        char *slot = p->local + (int64_t)p->nranks * kP2PFlagStride + ((int64_t)par * p->nranks + sender) * p->slot_bytes + offset;
        FD_HIP_CHECK(hipMemcpy(slot, data, (size_t)bytes, hipMemcpyDefault));
    }
    FD_HIP_CHECK(hipDeviceSynchronize());
    return FD_OK;
}

// the same for the FUSED step's cells: all 64 x 8 group sums (the rank's own groups are ignored) and the two halos as the lower /
// upper neighbour would deliver them (halo_bytes each, <= 64; NULL: none), into all three buffers -- which a loop-back step never resets
int fd_p2p_loopback_fill_fused(fd_p2p *p, const void *gsum64x8, const void *halo_lo, const void *halo_hi, int64_t halo_bytes)
{
    FD_REQUIRE(p && p->sink && gsum64x8, FD_ERR_ARG, "not a loop-back mailbox / NULL argument");
    FD_REQUIRE(halo_bytes >= 0 && halo_bytes <= kFzHaloBytes, FD_ERR_ARG, "halo of %lld bytes (at most %lld)", (long long)halo_bytes, (long long)kFzHaloBytes);
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    FD_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    // (the cells are colour-major -- cell (c, g) at c * 64 + g: a finisher polls its colour's 64 cells with one dense load)
    std::vector<double> gm(512), cm(512);
    FD_HIP_CHECK(hipMemcpy(gm.data(), gsum64x8, (size_t)kFzGsumBytes, hipMemcpyDefault));
    for (int g = 0; g < 64; ++g)
        for (int c = 0; c < 8; ++c) cm[(size_t)(c * 64 + g)] = gm[(size_t)(g * 8 + c)];
    for (int q = 0; q < kFzBufs; ++q) {
        char *b = p->local + p->fz_off + q * kFzBufBytes;
        FD_HIP_CHECK(hipMemcpy(b, cm.data(), (size_t)kFzGsumBytes, hipMemcpyHostToDevice));
        if (halo_lo && halo_bytes) FD_HIP_CHECK(hipMemcpy(b + kFzGsumBytes, halo_lo, (size_t)halo_bytes, hipMemcpyDefault));
        if (halo_hi && halo_bytes) FD_HIP_CHECK(hipMemcpy(b + kFzGsumBytes + kFzHaloBytes, halo_hi, (size_t)halo_bytes, hipMemcpyDefault));
    }
    FD_HIP_CHECK(hipDeviceSynchronize());
    return FD_OK;
}

int fd_p2p_destroy(fd_p2p *p)
{
    if (!p) return FD_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    if (p->sink) (void)hipFree(p->sink);
    for (int r = 0; r < p->nranks; ++r)
        if (p->mapped[r] && p->peer[r]) (void)hipIpcCloseMemHandle(p->peer[r]);
    if (p->d_peer) (void)hipFree(p->d_peer);
    if (p->d_arrived) (void)hipFree(p->d_arrived);
    if (p->d_scratch) (void)hipFree(p->d_scratch);
    if (p->h_err) (void)hipHostFree(p->h_err);
    if (p->local) { if (p->uncached) unc_park(p->local_bytes, p->local); else (void)hipFree(p->local); }
    delete p;
    return FD_OK;
}

int fd_p2p_info(const fd_p2p *p, int *nranks, int *rank, int64_t *slot_bytes, int *uncached)
{
    FD_REQUIRE(p != nullptr, FD_ERR_ARG, "p2p is NULL");
    if (nranks) *nranks = p->nranks;
    if (rank) *rank = p->rank;
    if (slot_bytes) *slot_bytes = p->slot_bytes;
    if (uncached) *uncached = p->uncached ? 1 : 0;
    return FD_OK;
}

// 0 while every exchange completed; 1 + r once a wait for rank r timed out (sticky; readable at any time: the word is host memory)
int fd_p2p_status(const fd_p2p *p, int *timed_out_rank_plus_1)
{
    FD_REQUIRE(p && timed_out_rank_plus_1, FD_ERR_ARG, "NULL argument");
    *timed_out_rank_plus_1 = *(volatile int *)p->h_err;
    return FD_OK;
}

static int64_t p2p_timeout_ticks()
{
    // wall_clock64() counts at 100 MHz on gfx9; FDJAC_P2P_TIMEOUT_MS bounds every wait (default 2 s)
    const char *v = getenv("FDJAC_P2P_TIMEOUT_MS");
    const int64_t ms = (v && *v) ? atoll(v) : 2000;
    return (ms < 1 ? 1 : ms) * 100000;
}

// in place, like fd_comm_allgather: buf holds nranks slots of `bytes` bytes, this rank's data in slot `rank`
int fd_p2p_allgather(fd_p2p *p, void *buf, int64_t bytes)
{
    FD_REQUIRE(p && (buf || bytes == 0), FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(bytes >= 0 && bytes % 8 == 0 && bytes <= p->slot_bytes, FD_ERR_ARG, "bytes = %lld must be a multiple of 8, at most the mailbox slot (%lld)",
               (long long)bytes, (long long)p->slot_bytes);
    FD_REQUIRE(p->connected, FD_ERR_COMM, "fd_p2p_connect has not been called");
    if (bytes == 0 || p->nranks == 1) return FD_OK;
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    const uint64_t epoch = ++p->epoch[0];
    P2PPut put = {};
    put.src_off[0] = (int64_t)p->rank * bytes;
    put.bytes[0] = bytes;
    P2PGet get = {};
    get.bytes[0] = bytes;
    // 8 KB per workgroup (one pass of four accesses per lane), at most 16 workgroups per peer
    const int parts = (int)std::min<int64_t>(std::max<int64_t>((bytes + 8191) / 8192, 1), 16);
    hipLaunchKernelGGL(k_p2p_exchange, dim3((unsigned)(p->nranks * parts)), dim3(kBlock), 0, p->ctx->stream, p->d_peer, p->local, (char *)buf, put, get, p->nranks,
                       p->rank, p->slot_bytes, epoch, 1, (int64_t)0, parts, p->d_arrived, p2p_timeout_ticks(), p->d_err);
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

// like fd_comm_halo_exchange: buf in GLOBAL indexing, this rank owns [own_begin, own_end)
int fd_p2p_halo_exchange(fd_p2p *p, void *buf, int64_t own_begin, int64_t own_end, int64_t halo, int elem_bytes)
{
    FD_REQUIRE(p && buf, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(elem_bytes == 8 || elem_bytes == 4, FD_ERR_ARG, "elem_bytes must be 4 or 8");
    FD_REQUIRE(halo >= 0 && own_begin >= 0 && own_end >= own_begin, FD_ERR_ARG, "bad range / halo");
    FD_REQUIRE(halo == 0 || own_end - own_begin >= halo, FD_ERR_ARG, "this rank owns fewer than `halo` elements");
    const int64_t hb = halo * elem_bytes;
    FD_REQUIRE(hb % 8 == 0 && 2 * hb <= p->slot_bytes, FD_ERR_ARG, "halo of %lld bytes: must be a multiple of 8 and fit twice into the mailbox slot", (long long)hb);
    FD_REQUIRE(p->connected, FD_ERR_COMM, "fd_p2p_connect has not been called");
    if (halo == 0 || p->nranks == 1) return FD_OK;
    const bool lo = p->rank > 0, hi = p->rank + 1 < p->nranks;
    FD_REQUIRE(!lo || own_begin >= halo, FD_ERR_ARG, "no room for the lower halo");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    const uint64_t epoch = ++p->epoch[1];
    const int64_t chan1 = (int64_t)p->nranks * kP2PFlagStride + 2 * (int64_t)p->nranks * p->slot_bytes;
    // my first `halo` elements go to rank - 1 (sub-slot 1: "from above"), my last to rank + 1 (sub-slot 0: "from below")
    P2PPut put = {};
    P2PGet get = {};
    if (lo) {
        put.target[put.n] = p->rank - 1; put.src_off[put.n] = own_begin * elem_bytes; put.dst_sub[put.n] = hb; put.bytes[put.n] = hb; ++put.n;
        get.sender[get.n] = p->rank - 1; get.dst_off[get.n] = (own_begin - halo) * elem_bytes; get.src_sub[get.n] = 0; get.bytes[get.n] = hb; ++get.n;
    }
    if (hi) {
        put.target[put.n] = p->rank + 1; put.src_off[put.n] = (own_end - halo) * elem_bytes; put.dst_sub[put.n] = 0; put.bytes[put.n] = hb; ++put.n;
        get.sender[get.n] = p->rank + 1; get.dst_off[get.n] = own_end * elem_bytes; get.src_sub[get.n] = hb; get.bytes[get.n] = hb; ++get.n;
    }
    hipLaunchKernelGGL(k_p2p_exchange, dim3((unsigned)put.n), dim3(kBlock), 0, p->ctx->stream, p->d_peer, p->local, (char *)buf, put, get, p->nranks, p->rank,
                       p->slot_bytes, epoch, 0, chan1, 1, p->d_arrived, p2p_timeout_ticks(), p->d_err);
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

}  // extern "C"


extern "C" int fdjac_p2p_step(fd_p2p *p, void *x, int64_t own_begin, int64_t own_end, int64_t halo, int elem_bytes, double *gsum,
                              int64_t gs_bytes, const fdjac_eps_final *fin)
{
    FD_REQUIRE(p && gsum && fin, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(p->connected, FD_ERR_COMM, "fd_p2p_connect has not been called");
    const bool with_halo = x != nullptr && halo > 0;
    const int64_t hb = with_halo ? halo * elem_bytes : 0;
    if (gs_bytes % 8 != 0 || hb % 8 != 0 || gs_bytes + hb > p->slot_bytes) return FD_ERR_UNSUPPORTED;      // (the caller falls back to RCCL)
    if (with_halo) {
        FD_REQUIRE(own_end - own_begin >= halo, FD_ERR_ARG, "this rank owns fewer than `halo` elements");
        FD_REQUIRE(p->rank == 0 || own_begin >= halo, FD_ERR_ARG, "no room for the lower halo");
    }
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    const uint64_t epoch = ++p->epoch[0];
    P2PStep st;
    st.x = with_halo ? (char *)x : nullptr;
    st.own_begin = own_begin; st.own_end = own_end; st.halo = halo; st.elem_bytes = elem_bytes;
    st.gsum = (char *)gsum; st.gs_bytes = gs_bytes;
    st.fin = *fin;
    hipLaunchKernelGGL(k_p2p_step, dim3((unsigned)p->nranks), dim3(kBlock), 0, p->ctx->stream, p->d_peer, p->local, st, p->nranks, p->rank, p->slot_bytes,
                       epoch, p->d_arrived + kP2PMaxRanks, p2p_timeout_ticks(), p->d_err);
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

// the fused step of a sharded call (csrc/fdjac_eps_dev.h) takes its next buffer here: everything the launch needs to know of the mailbox
extern "C" int fdjac_p2p_fused_begin(fd_p2p *p, fdjac_p2p_fused *out)
{
    FD_REQUIRE(p && out, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(p->connected, FD_ERR_COMM, "fd_p2p_connect has not been called");
    const uint64_t e = p->fz_epoch;
    out->peer = p->d_peer;
    out->local = p->local;
    out->fz_off = p->fz_off;
    out->buf = (int)(e % kFzBufs);
    out->buf_reset = p->sink ? -1 : (int)((e + 2) % kFzBufs);
    out->err = p->d_err;
    out->nranks = p->nranks;
    out->rank = p->rank;
    return FD_OK;
}
extern "C" void fdjac_p2p_fused_commit(fd_p2p *p) { if (p) ++p->fz_epoch; }
extern "C" int fdjac_p2p_shared_device(const fd_p2p *p) { return (p && p->shared_device) ? 1 : 0; }
// every rank's 8 doubles to every rank (blocking; the attach-time agreement of fd_plan_set_p2p): out = nranks x 8 doubles, host.
// A loop-back mailbox has nobody to agree with: out is filled with `mine`.
extern "C" int fdjac_p2p_agree8(fd_p2p *p, const double *mine, double *out)
{
    FD_REQUIRE(p && mine && out, FD_ERR_ARG, "NULL argument");
    if (p->sink || p->nranks == 1) {
        for (int r = 0; r < p->nranks; ++r) memcpy(out + 8 * r, mine, 64);
        return FD_OK;
    }
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    FD_HIP_CHECK(hipMemcpyAsync(p->d_scratch + 8 * p->rank, mine, 64, hipMemcpyHostToDevice, p->ctx->stream));
    const int rc = fd_p2p_allgather(p, p->d_scratch, 64);
    if (rc) return rc;
    FD_HIP_CHECK(hipMemcpyAsync(out, p->d_scratch, 64 * (size_t)p->nranks, hipMemcpyDeviceToHost, p->ctx->stream));
    FD_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    const int st = *(volatile int *)p->h_err;
    FD_REQUIRE(st == 0, FD_ERR_COMM, "rank %d never took part in the agreement (is fd_plan_set_p2p called on every rank, in the same order?)", st - 1);
    return FD_OK;
}
extern "C" int *fdjac_p2p_err_word(const fd_p2p *p) { return p ? p->d_err : nullptr; }
extern "C" int fdjac_p2p_failed(const fd_p2p *p) { return (p && p->h_err) ? *(volatile int *)p->h_err : 0; }

// internals for fdjac_comm.hip (a communicator with an attached mailbox routes its small messages here)
extern "C" int64_t fdjac_p2p_slot_bytes(const fd_p2p *p) { return p ? p->slot_bytes : 0; }
extern "C" const fd_ctx *fdjac_p2p_ctx(const fd_p2p *p) { return p ? p->ctx : nullptr; }
extern "C" int fdjac_p2p_nranks(const fd_p2p *p) { return p ? p->nranks : 0; }
extern "C" int fdjac_p2p_rank(const fd_p2p *p) { return p ? p->rank : -1; }

#endif /* FDJAC_F32 */
