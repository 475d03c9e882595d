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

struct vican_comm { ncclComm_t comm; int rank, world; int force_enqueue; };

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

extern "C" int vican_comm_allreduce_sum(vican_comm_t* comm, double* buf, int64_t n, void* stream) {
    if (!comm || !buf || n < 0) return set_err(VICAN_ERR_ARG, "%s: bad argument", "vican_comm_allreduce_sum");
    if (n == 0) return VICAN_OK;
    if (comm->world == 1) {
        // one rank: the sum is the identity and nothing is enqueued - unless vican_comm_force_enqueue asked for the collective
        // to run anyway (tests, timing on a 1-GPU box).  RCCL itself elides an in-place ncclSum on a one-rank communicator
        // (no kernel, no copy), so the forced call asks for ncclAvg = sum x 1/world = x 1.0: the same bits, and RCCL's own
        // one-rank reduce kernel on the caller's stream.
        if (!comm->force_enqueue) return VICAN_OK;
        const ncclResult_t e = rccl()->AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclAvg, comm->comm, (hipStream_t)stream);
        return e == ncclSuccess ? VICAN_OK : rccl_err("vican_comm_allreduce_sum", e);
    }
    const ncclResult_t e = rccl()->AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclSum, comm->comm, (hipStream_t)stream);
    return e == ncclSuccess ? VICAN_OK : rccl_err("vican_comm_allreduce_sum", e);
}

// test / timing switch (include/vican_hip_test.h): a one-rank communicator really calls ncclAllReduce
extern "C" int vican_comm_force_enqueue(vican_comm_t* comm, int32_t on) {
    if (!comm) return set_err(VICAN_ERR_ARG, "%s: NULL communicator", "vican_comm_force_enqueue");
    comm->force_enqueue = on != 0;
    return VICAN_OK;
}

extern "C" int vican_comm_destroy(vican_comm_t* comm) {
    if (!comm) return VICAN_OK;
    Rccl* r = rccl();
    if (r && comm->comm) r->CommDestroy(comm->comm);
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
