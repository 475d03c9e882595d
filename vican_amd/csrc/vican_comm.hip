// vican_comm.hip - the collective of the sharded solve behind the C ABI (vican_comm_*).
//
// What crosses ranks on this path is small and of one kind: sum-all-reduces of camera-side partials in f64 (3C x 3 doubles per
// operator application, two messages - 3C + 1 doubles and one scalar - per CG iteration, one set-up message).  The Python
// driver issues them through torch.distributed (its "nccl" backend IS RCCL).  These entry points let a caller without torch -
// or one who wants the collective enqueued by the same host call that enqueues the kernel in front of it - hold an RCCL
// communicator inside the library: the message is reduced in place by ncclAllReduce on the caller's stream, in stream order
// behind the kernels that produced it, no host synchronisation.  RCCL is loaded at run time (dlopen of librccl.so.1): the
// library has no link-time dependency on it, and single-GPU users never load it.
#include <dlfcn.h>
#include <mutex>
#include <vector>
#include <rccl/rccl.h>
#include "vican_common.h"
#include "vican_hip_test.h"

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
        if (r.h) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
            r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
            if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) { dlclose(r.h); r.h = nullptr; }
        }
    }
    return r.h ? &r : nullptr;
}

int rccl_err(const char* who, ncclResult_t e) {
    Rccl* r = rccl();
    snprintf(g_vican_err, sizeof(g_vican_err), "%s: RCCL error %d (%s)", who, (int)e, r && r->GetErrorString ? r->GetErrorString(e) : "?");
    return VICAN_ERR_LAUNCH;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// Peer exchange: the all-reduce as ONE launch of the library's own kernel over mailboxes that the ranks map into each other's
// address space (hipIpc: dmabuf handles; xGMI between the GPUs of a node).
//
// The messages of this path are 8 ... 72 KB of f64 (latency-bound: SURVEY.md 8(e)); a ring or tree of RCCL pays several hops
// and a host-side enqueue of its own.  Here every rank PUSHES its partial into a slot of every rank's mailbox and then sums the
// `world` slots of its OWN mailbox in rank order - one hop, and the same sum in the same order on every rank (bit-identical
// results: the replicated camera side must not drift apart).  No data-then-flag ordering is relied on: the payload travels in
// 8-byte granules {32 data bits | 32-bit tag}, each written by ONE 8-byte system-scope store (the granule RCCL's LL protocol
// uses on the same links) and valid iff its tag is the collective's epoch.  The thread that owns element i pushes it and later
// collects it: no workgroup waits for another workgroup of its own launch.
//   mailbox of a rank: granule[2 parities][world sources][2 * cap]   (uncached device memory: remote writes must not meet a
//                      stale line of the owner's L2)
//   epoch            : a DEVICE word, advanced by the launch itself (its last workgroup) - a launch that the solver's gate
//                      cancels on the device (speculative tails) advances nothing, on any rank; parities alternate with it, so
//                      a source can overwrite a slot only after the owner has collected it (it needs the owner's NEXT push first)
//   waits            : bounded (30 s; vican_comm_peer_set_timeout); a timeout raises the host-visible status word, the message
//                      is left as NaN, and the caller switches the group to its fall-back transport (vican_comm_peer_status)
// ---------------------------------------------------------------------------------------------------------------------------
#define VICAN_PEER_MAX 8
#define VICAN_PEER_WG 1024
struct PeerArgs {
    unsigned long long* mbox[VICAN_PEER_MAX];   // every rank's mailbox as mapped in THIS process ([rank] = the local one)
    int rank, world;
    long long cap;                              // doubles per message
    unsigned int* state;                        // device: [0] epoch, [1] arrival ticket, [2] a wait of this communicator has timed out
    unsigned int* status;                       // host-visible (pinned): [0] timed-out waits
    unsigned long long limit;                   // spin bound, ticks of the 100 MHz counter
};
struct vican_comm {
    ncclComm_t comm; int rank, world; int force_enqueue;
    // peer exchange
    void* mb_local = nullptr; int64_t mb_cap = 0; size_t mb_bytes = 0; bool mb_attached = false, mb_enabled = false;
    void* mb_opened[VICAN_PEER_MAX] = {};       // handles of the other ranks opened here (to be closed)
    PeerArgs pa = {};
    int device = -1;
};

namespace {
typedef unsigned long long u64;
__device__ __forceinline__ void granule_store(u64* p, unsigned int data, unsigned int tag) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ u64 granule_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// buf[0:n] <- sum over ranks (in place).  EPT elements per thread, all pushes before the first wait.
template <int EPT>
__global__ __launch_bounds__(VICAN_PEER_WG) void peer_allreduce_kernel(const int32_t* gate, PeerArgs a, double* buf, long long n) {
    GATE_RETURN(gate);
    __shared__ int s_last;
    const unsigned int epoch = __hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // a wait that timed out poisons the communicator (its results are NaN from there on, the host sees status != 0 and the group
    // falls back): later launches keep the protocol going - they push, they advance the epoch - but give a missing granule only
    // 1/256 of the bound, so that a solve in flight (a hundred exchanges) drains in seconds instead of a hundred full bounds
    const unsigned long long limit = __hip_atomic_load(a.state + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? (a.limit >> 8) + 100000ull : a.limit;
    const unsigned int tag = (epoch & 0x7FFFFFFFu) + 1u;
    const long long par = epoch & 1u;
    const long long slot = 2 * a.cap;                                    // granules per (parity, source)
    const long long base = (long long)blockIdx.x * (VICAN_PEER_WG * EPT) + threadIdx.x;
    // ---- push: this rank's slot in every mailbox (its own included: one code path for every source)
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const long long i = base + (long long)k * VICAN_PEER_WG;
        if (i < n) {
            const u64 bits = (u64)__double_as_longlong(buf[i]);
            const unsigned int lo = (unsigned int)bits, hi = (unsigned int)(bits >> 32);
            for (int d = 0; d < a.world; ++d) {
                const int dst = (a.rank + d) % a.world;                  // (every rank starts with itself: spreads the links)
                u64* q = a.mbox[dst] + (par * a.world + a.rank) * slot + 2 * i;
                granule_store(q, lo, tag);
                granule_store(q + 1, hi, tag);
            }
        }
    }
    // ---- collect: the `world` slots of the own mailbox, summed in rank order
    const u64* mine = a.mbox[a.rank] + par * a.world * slot;
    bool timed_out = false;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const long long i = base + (long long)k * VICAN_PEER_WG;
        if (i < n) {
            u64 g0[VICAN_PEER_MAX], g1[VICAN_PEER_MAX];
            unsigned int pending = a.world >= 32 ? 0xFFFFFFFFu : ((1u << a.world) - 1u);
            unsigned long long t0 = 0;
            unsigned int spins = 0;
            while (pending && !timed_out) {
#pragma unroll
                for (int s = 0; s < VICAN_PEER_MAX; ++s)
                    if (s < a.world && (pending >> s & 1u)) { g0[s] = granule_load(mine + s * slot + 2 * i); g1[s] = granule_load(mine + s * slot + 2 * i + 1); }
#pragma unroll
                for (int s = 0; s < VICAN_PEER_MAX; ++s)
                    if (s < a.world && (pending >> s & 1u) && (unsigned int)(g0[s] >> 32) == tag && (unsigned int)(g1[s] >> 32) == tag) pending &= ~(1u << s);
                if (pending) {
                    if (spins == 0) t0 = __builtin_amdgcn_s_memrealtime();
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > limit) timed_out = true;
                }
            }
            double acc = __longlong_as_double(0x7FF8000000000000LL);
            if (!timed_out) {
                acc = __longlong_as_double((long long)(((g1[0] & 0xFFFFFFFFull) << 32) | (g0[0] & 0xFFFFFFFFull)));     // (not 0 + v_0: keeps -0)
#pragma unroll
                for (int s = 1; s < VICAN_PEER_MAX; ++s)
                    if (s < a.world) acc += __longlong_as_double((long long)(((g1[s] & 0xFFFFFFFFull) << 32) | (g0[s] & 0xFFFFFFFFull)));
            }
            buf[i] = acc;
        }
    }
    if (timed_out) {
        __hip_atomic_fetch_add(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.state + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the workgroup that finishes last advances the epoch (every workgroup has read it by then)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = __hip_atomic_fetch_add(a.state + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == gridDim.x - 1u;
        if (s_last) {
            __hip_atomic_store(a.state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.state, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int peer_allreduce(vican_comm* c, double* buf, int64_t n, hipStream_t st) {
    const int64_t per_wg = VICAN_PEER_WG;
    if (n <= 64 * per_wg) {
        const unsigned grid = (unsigned)((n + per_wg - 1) / per_wg);
        hipLaunchKernelGGL(peer_allreduce_kernel<1>, dim3(grid), dim3(VICAN_PEER_WG), 0, st, g_vican_gate, c->pa, buf, (long long)n);
    } else {
        const unsigned grid = (unsigned)((n + 4 * per_wg - 1) / (4 * per_wg));
        hipLaunchKernelGGL(peer_allreduce_kernel<4>, dim3(grid), dim3(VICAN_PEER_WG), 0, st, g_vican_gate, c->pa, buf, (long long)n);
    }
    LAUNCH_CHECK("vican_comm_allreduce_sum (peer exchange)");
    return VICAN_OK;
}
}  // namespace

extern "C" int vican_comm_unique_id(void* id_out) {
    Rccl* r = rccl();
    if (!r) return set_err(VICAN_ERR_LAUNCH, "%s: librccl.so.1 could not be loaded", "vican_comm_unique_id");
    if (!id_out) return set_err(VICAN_ERR_ARG, "%s: NULL output", "vican_comm_unique_id");
    ncclUniqueId id;
    const ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return rccl_err("vican_comm_unique_id", e);
    memcpy(id_out, &id, sizeof(id));
    return VICAN_OK;
}

extern "C" int vican_comm_create(int32_t rank, int32_t world, const void* unique_id, vican_comm_t** comm_out) {
    Rccl* r = rccl();
    if (!r) return set_err(VICAN_ERR_LAUNCH, "%s: librccl.so.1 could not be loaded", "vican_comm_create");
    if (!unique_id || !comm_out || world < 1 || rank < 0 || rank >= world) return set_err(VICAN_ERR_ARG, "%s: bad argument", "vican_comm_create");
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t e = r->CommInitRank(&c, world, id, rank);          // collective: every rank of the group calls it
    if (e != ncclSuccess) return rccl_err("vican_comm_create", e);
    *comm_out = new vican_comm{c, rank, world, 0};
    return VICAN_OK;
}

// a communicator WITHOUT RCCL: its all-reduces exist only as the peer exchange below (vican_comm_peer_export / _attach) -
// ranks that share one GPU (RCCL refuses those), callers that do not want librccl loaded
extern "C" int vican_comm_create_local(int32_t rank, int32_t world, vican_comm_t** comm_out) {
    if (!comm_out || world < 1 || world > VICAN_PEER_MAX || rank < 0 || rank >= world) return set_err(VICAN_ERR_ARG, "%s: bad argument", "vican_comm_create_local");
    *comm_out = new vican_comm{nullptr, rank, world, 0};
    return VICAN_OK;
}

// ---- peer exchange: set-up ----------------------------------------------------------------------------------------------
#define PEER_HIP(call, who)                                                                                 \
    do {                                                                                                    \
        const hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                             \
            snprintf(g_vican_err, sizeof(g_vican_err), "%s: %s: %s", who, #call, hipGetErrorString(e_));    \
            (void)hipGetLastError();                                                                        \
            return VICAN_ERR_LAUNCH;                                                                        \
        }                                                                                                   \
    } while (0)

// Mailboxes are NEVER handed back to the driver while the process lives: a freed one goes to a free list keyed by its size and
// device and serves the next communicator (zeroed again).  Round 6 measured why: create -> exchange -> destroy -> create cycles
// that hipFree'd the uncached allocation corrupted LATER allocations of the process (tensors that took over the address range
// read back other bytes; now and then a memory fault) - one communicator kept for all solves never did, nor did communicators
// that were made and destroyed without a launch having touched their mailbox.  (A communicator per process group lives as
// long as the process anyway: vican_amd/solver.py keeps them; the churn comes from one-rank communicators of tests / timing.)
struct PeerMem { void* mbox; unsigned int* state; unsigned int* status; size_t bytes; int device; };
static std::vector<PeerMem>& peer_pool() { static std::vector<PeerMem> p; return p; }
static std::mutex& peer_pool_mutex() { static std::mutex m; return m; }

static void peer_release(vican_comm* c) {
    for (int r = 0; r < VICAN_PEER_MAX; ++r)
        if (c->mb_opened[r]) { (void)hipIpcCloseMemHandle(c->mb_opened[r]); c->mb_opened[r] = nullptr; }
    if (c->mb_local) {
        // (a mailbox whose handle other processes may still hold open is not recycled: it is simply kept)
        if (c->world == 1) {
            std::lock_guard<std::mutex> lk(peer_pool_mutex());
            peer_pool().push_back(PeerMem{c->mb_local, c->pa.state, c->pa.status, c->mb_bytes, c->device});
        }
        c->mb_local = nullptr; c->pa.state = nullptr; c->pa.status = nullptr;
    }
    c->mb_attached = c->mb_enabled = false;
}

extern "C" int64_t vican_comm_peer_bytes(int32_t world, int64_t max_doubles) {
    if (world < 1 || world > VICAN_PEER_MAX || max_doubles < 1) return 0;
    return 2LL * world * 2LL * max_doubles * 8LL;
}

extern "C" int vican_comm_peer_export(vican_comm_t* c, int64_t max_doubles, void* handle_out) {
    if (!c || max_doubles < 1 || !handle_out) return set_err(VICAN_ERR_ARG, "%s: bad argument", "vican_comm_peer_export");
    if (c->world > VICAN_PEER_MAX) return set_err(VICAN_ERR_CAPACITY, "%s: more ranks than the peer exchange serves", "vican_comm_peer_export");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "vican_comm_peer_export hands out 64-byte handles");
    peer_release(c);
    const char* who = "vican_comm_peer_export";
    PEER_HIP(hipGetDevice(&c->device), who);
    const size_t bytes = (size_t)vican_comm_peer_bytes(c->world, max_doubles);
    {
        std::lock_guard<std::mutex> lk(peer_pool_mutex());
        auto& pool = peer_pool();
        for (size_t i = 0; i < pool.size(); ++i)
            if (pool[i].bytes == bytes && pool[i].device == c->device) {
                c->mb_local = pool[i].mbox; c->pa.state = pool[i].state; c->pa.status = pool[i].status;
                pool.erase(pool.begin() + (long)i);
                break;
            }
    }
    if (!c->mb_local) {
        // uncached: a remote rank's stores must never meet a stale line of this GPU's L2 (and nothing of the mailbox is ever re-read)
        PEER_HIP(hipExtMallocWithFlags(&c->mb_local, bytes, hipDeviceMallocUncached), who);
        PEER_HIP(hipMalloc((void**)&c->pa.state, 256), who);
        PEER_HIP(hipHostMalloc((void**)&c->pa.status, 64, hipHostMallocMapped), who);
    }
    c->mb_bytes = bytes;
    PEER_HIP(hipMemset(c->mb_local, 0, bytes), who);
    PEER_HIP(hipMemset(c->pa.state, 0, 256), who);
    memset(c->pa.status, 0, 64);
    PEER_HIP(hipDeviceSynchronize(), who);
    c->mb_cap = max_doubles;
    memset(handle_out, 0, 64);
    if (c->world > 1) {
        hipIpcMemHandle_t h;
        PEER_HIP(hipIpcGetMemHandle(&h, c->mb_local), who);
        memcpy(handle_out, &h, 64);
    }
    return VICAN_OK;
}

extern "C" int vican_comm_peer_attach(vican_comm_t* c, const void* handles) {
    const char* who = "vican_comm_peer_attach";
    if (!c || !c->mb_local || (c->world > 1 && !handles)) return set_err(VICAN_ERR_ARG, "%s: bad argument (vican_comm_peer_export first)", who);
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) { c->pa.mbox[r] = (unsigned long long*)c->mb_local; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + 64 * r, 64);
        void* p = nullptr;
        PEER_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), who);
        c->mb_opened[r] = p;
        c->pa.mbox[r] = (unsigned long long*)p;
    }
    c->pa.rank = c->rank; c->pa.world = c->world; c->pa.cap = c->mb_cap;
    // waits of the exchange: 30 s by default - a rank may reach a collective seconds after its peers (packing, a first-call
    // module load): RCCL's kernels would wait for it indefinitely; this bound only ends waits for a rank that is gone
    // (the 2 s bound of the cooperative kernels' grid barriers guards something else: a grid that is not co-resident)
    c->pa.limit = 3000000000ull;
    c->mb_attached = true;
    c->mb_enabled = true;
    return VICAN_OK;
}

extern "C" int vican_comm_peer_set_timeout(vican_comm_t* c, int64_t microseconds) {
    if (!c || microseconds <= 0) return set_err(VICAN_ERR_ARG, "%s: bad argument", "vican_comm_peer_set_timeout");
    c->pa.limit = (unsigned long long)microseconds * 100ull;            // ticks of the 100 MHz real-time counter
    return VICAN_OK;
}

extern "C" int vican_comm_peer_enable(vican_comm_t* c, int32_t on) {
    if (!c) return set_err(VICAN_ERR_ARG, "%s: NULL communicator", "vican_comm_peer_enable");
    if (on && !c->mb_attached) return set_err(VICAN_ERR_ARG, "%s: no mailboxes attached", "vican_comm_peer_enable");
    c->mb_enabled = on != 0;
    return VICAN_OK;
}

// 0: every wait so far was served; > 0: that many element waits timed out (their messages came back as NaN) - the caller
// stops using the exchange (vican_comm_peer_enable(c, 0)) and reports; < 0: no exchange attached.  Host-visible word: no sync.
extern "C" int vican_comm_peer_status(vican_comm_t* c) {
    if (!c || !c->mb_attached) return VICAN_ERR_ARG;
    return (int)*(volatile unsigned int*)c->pa.status;
}

extern "C" int vican_comm_allreduce_sum(vican_comm_t* comm, double* buf, int64_t n, void* stream) {
    if (!comm || !buf || n < 0) return set_err(VICAN_ERR_ARG, "%s: bad argument", "vican_comm_allreduce_sum");
    if (n == 0) return VICAN_OK;
    if (comm->mb_enabled && n <= comm->mb_cap) return peer_allreduce(comm, buf, n, (hipStream_t)stream);
    if (!comm->comm && comm->world > 1)
        return set_err(VICAN_ERR_ARG, "%s: a local communicator has no collective but the peer exchange (message larger than the mailboxes, or exchange disabled)", "vican_comm_allreduce_sum");
    if (comm->world == 1) {
        // one rank: the sum is the identity and nothing is enqueued - unless vican_comm_force_enqueue asked for the collective
        // to run anyway (tests, timing on a 1-GPU box).  RCCL itself elides an in-place ncclSum on a one-rank communicator
        // (no kernel, no copy), so the forced call asks for ncclAvg = sum x 1/world = x 1.0: the same bits, and RCCL's own
        // one-rank reduce kernel on the caller's stream.
        if (!comm->force_enqueue || !comm->comm) return VICAN_OK;
        const ncclResult_t e = rccl()->AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclAvg, comm->comm, (hipStream_t)stream);
        return e == ncclSuccess ? VICAN_OK : rccl_err("vican_comm_allreduce_sum", e);
    }
    const ncclResult_t e = rccl()->AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclSum, comm->comm, (hipStream_t)stream);
    return e == ncclSuccess ? VICAN_OK : rccl_err("vican_comm_allreduce_sum", e);
}

// test hook (include/vican_hip_test.h): count one timed-out wait without there having been one - lets a test walk the recovery
// path (solver.Comm.healthy: the whole group falls back to RCCL / torch.distributed) on hardware where the exchange works
extern "C" int vican_comm_peer_inject_fault(vican_comm_t* c) {
    if (!c || !c->mb_attached) return set_err(VICAN_ERR_ARG, "%s: no exchange attached", "vican_comm_peer_inject_fault");
    *(volatile unsigned int*)c->pa.status += 1u;
    return VICAN_OK;
}

// test / timing switch (include/vican_hip_test.h): a one-rank communicator really calls ncclAllReduce
extern "C" int vican_comm_force_enqueue(vican_comm_t* comm, int32_t on) {
    if (!comm) return set_err(VICAN_ERR_ARG, "%s: NULL communicator", "vican_comm_force_enqueue");
    comm->force_enqueue = on != 0;
    return VICAN_OK;
}

extern "C" int vican_comm_destroy(vican_comm_t* comm) {
    if (!comm) return VICAN_OK;
    peer_release(comm);
    if (comm->comm) { Rccl* r = rccl(); if (r) r->CommDestroy(comm->comm); }
    delete comm;
    return VICAN_OK;
}

// z = P x of this rank's rows, summed over the ranks: the sweep, the slab fold and the all-reduce behind ONE host call
// (vican_block_op_z + vican_comm_allreduce_sum; comm == NULL: single rank)
extern "C" int vican_block_op_z_comm(const vican_graph_t* g, const double* lamT_inv, const double* x, void* zpart, double* fx, double* z,
                                     vican_comm_t* comm, void* stream) {
    const int rc = vican_block_op_z(g, lamT_inv, x, zpart, fx, z, stream);
    if (rc < 0 || !comm) return rc;
    return vican_comm_allreduce_sum(comm, z, 9LL * g->n_cam, stream);
}
