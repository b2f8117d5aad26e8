// The ONE exchange step of data-parallel training: a sum-all-reduce of the flat fp32 gradient buffer over
// RCCL (xGMI inside a node), as plain C entry points so that a host without PyTorch has the multi-GPU path too.
// The reference trains on a single device (bfcnn/train_loop.py:259-321): there is no collective to mirror, the contract is
// this repo's (SURVEY.md 8b `bf_allreduce_grads`).
//
// RCCL is bound at run time (dlopen of librccl.so.1, the soname PyTorch-ROCm's own copy carries: a process that already
// uses torch.distributed gets the SAME library instance), so libbfcnn_hip.so loads on a machine without RCCL and the
// single-GPU paths do not depend on it.
#include "bf_common.h"
#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace {

typedef int nccl_result;                       // ncclResult_t (ncclSuccess == 0)
struct nccl_unique_id { char internal[128]; }; // ncclUniqueId (NCCL_UNIQUE_ID_BYTES == 128)
typedef void* nccl_comm;                       // ncclComm_t
enum { NCCL_FLOAT32 = 7, NCCL_SUM = 0 };      // ncclDataType_t / ncclRedOp_t values of rccl.h

struct Rccl {
    void* lib = nullptr;
    nccl_result (*GetUniqueId)(nccl_unique_id*) = nullptr;
    nccl_result (*CommInitRank)(nccl_comm*, int, nccl_unique_id, int) = nullptr;
    nccl_result (*CommDestroy)(nccl_comm) = nullptr;
    nccl_result (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(nccl_result) = nullptr;
    char error[256] = {0};
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            snprintf(r.error, sizeof(r.error), "RCCL is not available: %s", dlerror());
            return;
        }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) {
            snprintf(r.error, sizeof(r.error), "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
            r.lib = nullptr;
        }
    });
    return r;
}

thread_local char g_comm_error[256] = {0};

int comm_fail(const char* what, nccl_result rc)
{
    Rccl& r = rccl();
    snprintf(g_comm_error, sizeof(g_comm_error), "%s: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
    return BF_EHIP;
}

}  // namespace

extern "C" const char* bf_comm_last_error(void) { return g_comm_error[0] ? g_comm_error : rccl().error; }

// rank 0 makes the 128-byte rendezvous id and hands it to the other ranks by any host channel (file, socket, MPI, ...)
extern "C" int bf_comm_unique_id(char id_out[128])
{
    g_comm_error[0] = 0;                       // bf_comm_last_error reports THIS call
    Rccl& r = rccl();
    if (!id_out) return BF_EINVAL;
    if (!r.lib) return BF_EUNSUPPORTED;
    nccl_unique_id id;
    const nccl_result rc = r.GetUniqueId(&id);
    if (rc) return comm_fail("ncclGetUniqueId", rc);
    memcpy(id_out, id.internal, 128);
    return BF_OK;
}

// one communicator per process / GPU: call with that GPU current (hipSetDevice); collective over all `world` ranks
extern "C" int bf_comm_init_rank(void** comm_out, int world, int rank, const char id[128])
{
    g_comm_error[0] = 0;                       // bf_comm_last_error reports THIS call
    Rccl& r = rccl();
    if (!comm_out || !id || world <= 0 || rank < 0 || rank >= world) return BF_EINVAL;
    if (!r.lib) return BF_EUNSUPPORTED;
    nccl_unique_id uid;
    memcpy(uid.internal, id, 128);
    nccl_comm c = nullptr;
    const nccl_result rc = r.CommInitRank(&c, world, uid, rank);
    if (rc) return comm_fail("ncclCommInitRank", rc);
    *comm_out = c;
    return BF_OK;
}

extern "C" int bf_comm_destroy(void* comm)
{
    g_comm_error[0] = 0;                       // bf_comm_last_error reports THIS call
    Rccl& r = rccl();
    if (!comm) return BF_OK;
    if (!r.lib) return BF_EUNSUPPORTED;
    const nccl_result rc = r.CommDestroy((nccl_comm)comm);
    return rc ? comm_fail("ncclCommDestroy", rc) : BF_OK;
}

// grads <- sum over ranks of grads (in place), stream-ordered on `stream`; the 1 / world scale rides in bf_adam_step's
// grad_scale.  `comm` is an ncclComm_t: bf_comm_init_rank's, or one the host already owns.
extern "C" int bf_allreduce_grads(bf_handle h, float* grads, int64_t n, void* comm, void* stream)
{
    (void)h;
    g_comm_error[0] = 0;
    Rccl& r = rccl();
    if (!grads || n <= 0 || !comm) return BF_EINVAL;
    if (!r.lib) return BF_EUNSUPPORTED;
    const nccl_result rc = r.AllReduce(grads, grads, (size_t)n, NCCL_FLOAT32, NCCL_SUM, (nccl_comm)comm, (hipStream_t)stream);
    return rc ? comm_fail("ncclAllReduce", rc) : BF_OK;
}
