// comm.hip -- the one exchange step of the data-parallel path behind the C ABI: sum all-reduce of a flat gradient bucket
// over RCCL (xGMI inside a node), on a communication stream of its own, with event hand-off from / to the compute stream.
//
// Replaces, for the encoder's gradients, the reference's explicit helper
//   Image/segmentation/mmseg_custom/core/utils/dist_utils.py:14-35  (_allreduce_coalesced: bucket -> flatten ->
//   dist.all_reduce -> div_(world_size) -> unflatten / copy back)
// and what DistributedDataParallel does for the Video / Image pipelines (Video/run_class_finetuning.py:739-742).  Here
// "flatten / copy back" does not exist -- every .grad is a view of one flat buffer (parallel.FlatParams) -- and the
// division by the world size is folded into the fused AdamW (me_adamw_step grad_scale), so the exchange is exactly one
// ncclAllReduce(sum) per bucket.
//
// RCCL is resolved at run time (dlopen) by me_comm_unique_id / me_comm_init only: libmetaenc.so carries no link-time
// dependency on it, single-GPU use never touches it, and inside a torch process the already-loaded librccl is reused.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {

// the slice of the RCCL / NCCL C API used here (stable since NCCL 2.x)
typedef void* ncclComm_t;
struct ncclUniqueId_ { char internal[ME_COMM_ID_BYTES]; };
enum { NCCL_SUCCESS = 0, NCCL_SUM = 0, NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9 };
typedef int (*fn_GetUniqueId)(ncclUniqueId_*);
typedef int (*fn_CommInitRank)(ncclComm_t*, int, ncclUniqueId_, int);
typedef int (*fn_CommDestroy)(ncclComm_t);
typedef int (*fn_AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef const char* (*fn_GetErrorString)(int);

struct Rccl {
    void* handle = nullptr;
    fn_GetUniqueId GetUniqueId = nullptr;
    fn_CommInitRank CommInitRank = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_AllReduce AllReduce = nullptr;
    fn_GetErrorString GetErrorString = nullptr;
    char why[256] = "";
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl not found: %s", dlerror());
        return;
    }
    g_rccl.GetUniqueId = (fn_GetUniqueId)dlsym(g_rccl.handle, "ncclGetUniqueId");
    g_rccl.CommInitRank = (fn_CommInitRank)dlsym(g_rccl.handle, "ncclCommInitRank");
    g_rccl.CommDestroy = (fn_CommDestroy)dlsym(g_rccl.handle, "ncclCommDestroy");
    g_rccl.AllReduce = (fn_AllReduce)dlsym(g_rccl.handle, "ncclAllReduce");
    g_rccl.GetErrorString = (fn_GetErrorString)dlsym(g_rccl.handle, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce) {
        snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl lacks a required symbol");
        g_rccl.handle = nullptr;
    }
}
const Rccl* rccl() {
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.handle ? &g_rccl : nullptr;
}
const char* rccl_err(int rc) {
    return (g_rccl.GetErrorString) ? g_rccl.GetErrorString(rc) : "RCCL error";
}

}  // namespace

struct me_comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;        // the communication stream
    hipEvent_t ready = nullptr;          // producer stream -> comm stream
    hipEvent_t done = nullptr;           // comm stream -> consumer stream
    int64_t buckets = 0;
};

#define ME_CHECK_HIP(call, what)                                                      \
    do {                                                                              \
        hipError_t e__ = (call);                                                      \
        if (e__ != hipSuccess) {                                                      \
            me_set_error("%s: %s", what, hipGetErrorString(e__));                     \
            return ME_ERR_HIP;                                                        \
        }                                                                             \
    } while (0)

extern "C" int me_comm_unique_id(void* id_out) {
    ME_CHECK_ARG(id_out != nullptr, "me_comm_unique_id: null id");
    const Rccl* r = rccl();
    if (!r) { me_set_error("me_comm_unique_id: %s", g_rccl.why); return ME_ERR_UNSUPPORTED; }
    ncclUniqueId_ id;
    const int rc = r->GetUniqueId(&id);
    if (rc != NCCL_SUCCESS) { me_set_error("ncclGetUniqueId: %s", rccl_err(rc)); return ME_ERR_HIP; }
    memcpy(id_out, id.internal, ME_COMM_ID_BYTES);
    return ME_OK;
}

extern "C" int me_comm_init(me_comm** out, const void* unique_id, int rank, int world, int device) {
    ME_CHECK_ARG(out && unique_id, "me_comm_init: null pointer");
    ME_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "me_comm_init: bad rank %d of %d", rank, world);
    const Rccl* r = rccl();
    if (!r) { me_set_error("me_comm_init: %s", g_rccl.why); return ME_ERR_UNSUPPORTED; }
    ME_CHECK_HIP(hipSetDevice(device), "me_comm_init(hipSetDevice)");
    me_comm* c = new me_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId_ id;
    memcpy(id.internal, unique_id, ME_COMM_ID_BYTES);
    const int rc = r->CommInitRank(&c->nccl, world, id, rank);
    if (rc != NCCL_SUCCESS) {
        me_set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, rccl_err(rc));
        delete c;
        return ME_ERR_HIP;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        me_set_error("me_comm_init: stream / event creation failed");
        (void)me_comm_destroy(c);
        return ME_ERR_HIP;
    }
    *out = c;
    // (me_gemm_reserve_cus is NOT set here: in the one-GPU rehearsal a CU reservation made the weight gradients under a 16-CU hold slower,
    //  not faster -- profiles/r06_contention.txt; it stays a caller's switch until an 8-GPU box says otherwise)
    return ME_OK;
}

extern "C" int me_comm_destroy(me_comm* c) {
    if (!c) return ME_OK;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nccl);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return ME_OK;
}

extern "C" int me_comm_info(const me_comm* c, int* rank, int* world, int64_t* buckets_reduced) {
    ME_CHECK_ARG(c != nullptr, "me_comm_info: null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (buckets_reduced) *buckets_reduced = c->buckets;
    return ME_OK;
}

extern "C" int me_allreduce_bucket(me_comm* c, void* buf, int64_t count, int dtype, void* producer_stream) {
    ME_CHECK_ARG(c && c->nccl, "me_allreduce_bucket: null communicator");
    ME_CHECK_ARG(buf != nullptr && count > 0, "me_allreduce_bucket: empty bucket");
    ME_CHECK_ARG(dtype == ME_F32 || dtype == ME_BF16, "me_allreduce_bucket: bad dtype %d", dtype);
    // everything enqueued on the producer (backward) stream so far has written the bucket before the reduction reads it
    ME_CHECK_HIP(hipEventRecord(c->ready, reinterpret_cast<hipStream_t>(producer_stream)), "me_allreduce_bucket(record)");
    ME_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0), "me_allreduce_bucket(wait)");
    const int rc = g_rccl.AllReduce(buf, buf, (size_t)count, dtype == ME_F32 ? NCCL_FLOAT32 : NCCL_BFLOAT16, NCCL_SUM, c->nccl,
                                    c->stream);
    if (rc != NCCL_SUCCESS) { me_set_error("ncclAllReduce: %s", rccl_err(rc)); return ME_ERR_HIP; }
    ++c->buckets;
    return ME_OK;
}

extern "C" int me_comm_join(me_comm* c, void* consumer_stream) {
    ME_CHECK_ARG(c != nullptr, "me_comm_join: null communicator");
    ME_CHECK_HIP(hipEventRecord(c->done, c->stream), "me_comm_join(record)");
    ME_CHECK_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(consumer_stream), c->done, 0), "me_comm_join(wait)");
    return ME_OK;
}
