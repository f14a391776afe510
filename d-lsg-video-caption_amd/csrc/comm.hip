// Gradient all-reduce of the data-parallel train step over RCCL (xGMI), behind the C ABI.
//
// Replaces the reference's `torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)` gradient
// exchange over NCCL (run_gun.py:63-64, train_debug.py:20): one process per GPU, one communicator per process, sum-all-reduce
// of ranges of the flat gradient arena, 1/world folded into the Adam kernel (dlsg_adam grad_scale).
//
// The collective is enqueued on the stream the caller passes.  RCCL's kernels are capturable, so the caller (dlsg_amd.Trainer)
// issues each bucket on a side stream forked by an event INSIDE the capture of the train step: the whole step -- forward,
// loss, backward, the bucket all-reduces overlapping the rest of the backward, Adam -- is ONE hipGraph replay per iteration.
//
// librccl is resolved at run time (dlopen + dlsym) so that the kernel library itself loads on a host without RCCL and so that
// the process-wide RCCL instance is shared: inside a PyTorch process the librccl that libtorch_hip.so already mapped is
// reused (two RCCL instances in one process would each bootstrap their own shared-memory / IPC state).
//
// The only state the library holds is the communicator handle the caller owns (dlsg_comm_init .. dlsg_comm_destroy).
#include <dlfcn.h>
#include <link.h>
#include <string.h>

#include <new>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "dlsg.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
};

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* data) {
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) {
        strncpy(static_cast<char*>(data), info->dlpi_name, 1023);
        return 1;
    }
    return 0;
}

// 0 on success.  Not cached in a global: every communicator keeps its own resolved table (dlopen reference-counts).
int load_rccl(RcclApi* api) {
    char path[1024] = {0};
    void* h = nullptr;
    if (dl_iterate_phdr(find_loaded_rccl, path) && path[0]) h = dlopen(path, RTLD_NOW | RTLD_NOLOAD);   // the one already mapped
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) return DLSG_ENOCOMM;
    api->handle = h;
#define DLSG_SYM(field, name)                                                     \
    *reinterpret_cast<void**>(&api->field) = dlsym(h, name);                      \
    if (!api->field) { dlclose(h); api->handle = nullptr; return DLSG_ENOCOMM; }
    DLSG_SYM(GetVersion, "ncclGetVersion")
    DLSG_SYM(GetUniqueId, "ncclGetUniqueId")
    DLSG_SYM(CommInitRank, "ncclCommInitRank")
    DLSG_SYM(CommDestroy, "ncclCommDestroy")
    DLSG_SYM(CommAbort, "ncclCommAbort")
    DLSG_SYM(AllReduce, "ncclAllReduce")
    DLSG_SYM(GroupStart, "ncclGroupStart")
    DLSG_SYM(GroupEnd, "ncclGroupEnd")
    DLSG_SYM(CommGetAsyncError, "ncclCommGetAsyncError")
#undef DLSG_SYM
    return DLSG_OK;
}

}  // namespace

struct dlsg_comm {
    RcclApi api;
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0, version = 0;
};

extern "C" int dlsg_comm_unique_id(void* id128) {
    if (!id128) return DLSG_EINVAL;
    RcclApi api;
    int rc = load_rccl(&api);
    if (rc) return rc;
    ncclUniqueId id;
    static_assert(sizeof(id) == DLSG_COMM_ID_BYTES, "ncclUniqueId size");
    const ncclResult_t r = api.GetUniqueId(&id);
    dlclose(api.handle);
    if (r != ncclSuccess) return DLSG_ELAUNCH;
    memcpy(id128, &id, sizeof(id));
    return DLSG_OK;
}

extern "C" int dlsg_comm_init(dlsg_comm** out, const void* id128, int world, int rank) {
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return DLSG_EINVAL;
    *out = nullptr;
    dlsg_comm* c = new (std::nothrow) dlsg_comm();
    if (!c) return DLSG_ELAUNCH;
    int rc = load_rccl(&c->api);
    if (rc) { delete c; return rc; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    c->api.GetVersion(&c->version);
    // collective call: every rank of the job must be inside ncclCommInitRank with the same id (current HIP device = the rank's)
    if (c->api.CommInitRank(&c->comm, world, id, rank) != ncclSuccess) {
        dlclose(c->api.handle);
        delete c;
        return DLSG_ELAUNCH;
    }
    c->world = world;
    c->rank = rank;
    *out = c;
    return DLSG_OK;
}

extern "C" int dlsg_comm_destroy(dlsg_comm* c) {
    if (!c) return DLSG_OK;
    ncclResult_t r = ncclSuccess;
    if (c->comm) r = c->api.CommDestroy(c->comm);
    if (c->api.handle) dlclose(c->api.handle);
    delete c;
    return r == ncclSuccess ? DLSG_OK : DLSG_ELAUNCH;
}

extern "C" int dlsg_comm_info(const dlsg_comm* c, int32_t* world, int32_t* rank, int32_t* rccl_version) {
    if (!c) return DLSG_ENOCOMM;
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    if (rccl_version) *rccl_version = c->version;
    return DLSG_OK;
}

extern "C" int dlsg_allreduce_bucket(dlsg_comm* c, float* grads, int64_t count, void* stream) {
    if (!c || !c->comm) return DLSG_ENOCOMM;
    if (count < 0 || (count > 0 && !grads)) return DLSG_EINVAL;
    if (count == 0) return DLSG_OK;
    const ncclResult_t r =
        c->api.AllReduce(grads, grads, static_cast<size_t>(count), ncclFloat32, ncclSum, c->comm, reinterpret_cast<hipStream_t>(stream));
    return r == ncclSuccess ? DLSG_OK : DLSG_ELAUNCH;
}

extern "C" int dlsg_allreduce_buckets(dlsg_comm* c, float* const* grads, const int64_t* counts, int n, void* stream) {
    if (!c || !c->comm) return DLSG_ENOCOMM;
    if (n < 0 || (n > 0 && (!grads || !counts))) return DLSG_EINVAL;
    if (n == 0) return DLSG_OK;
    // several ranges of one bucket (the arena minus frozen parameters): one RCCL group = one fused launch
    if (c->api.GroupStart() != ncclSuccess) return DLSG_ELAUNCH;
    ncclResult_t r = ncclSuccess;
    for (int i = 0; i < n && r == ncclSuccess; ++i) {
        if (counts[i] <= 0) continue;
        r = c->api.AllReduce(grads[i], grads[i], static_cast<size_t>(counts[i]), ncclFloat32, ncclSum, c->comm,
                             reinterpret_cast<hipStream_t>(stream));
    }
    const ncclResult_t e = c->api.GroupEnd();
    return (r == ncclSuccess && e == ncclSuccess) ? DLSG_OK : DLSG_ELAUNCH;
}

// words[i] <- max over the ranks of words[i] (int32): how the persistent kernels' time-out word becomes the SAME guard on every
// rank before Adam (a rank that skipped its update alone would leave the replicas diverged)
extern "C" int dlsg_allreduce_max_i32(dlsg_comm* c, int32_t* words, int64_t count, void* stream) {
    if (!c || !c->comm) return DLSG_ENOCOMM;
    if (count < 0 || (count > 0 && !words)) return DLSG_EINVAL;
    if (count == 0) return DLSG_OK;
    const ncclResult_t r =
        c->api.AllReduce(words, words, static_cast<size_t>(count), ncclInt32, ncclMax, c->comm, reinterpret_cast<hipStream_t>(stream));
    return r == ncclSuccess ? DLSG_OK : DLSG_ELAUNCH;
}

// ---- single-GPU rehearsal of a collective's footprint.  With one rank an all-reduce is the identity and RCCL launches nothing
// of substance, so what a step pays for SHARING the chip with a bucket's ring all-reduce cannot be seen on a one-GPU box.  This
// kernel stands in for it: `workgroups` x 256 threads (RCCL's launch shape on gfx942 / gfx950: one workgroup per channel) that
// stream the bucket `passes` times, read-modify-write with a factor of 1.0f -- the values are unchanged, so a rehearsed step
// is still bit-identical to the plain one -- holding `workgroups` CUs' worth of wave slots and their share of HBM for about as
// long as the caller asks.  Test / bench instrument only (dlsg_amd.Trainer.rehearse_cotenant); nothing in a real run calls it.
namespace {
__global__ __launch_bounds__(256) void comm_rehearsal_kernel(float* buf, int64_t n4, int passes, float one) {
    float4* b = reinterpret_cast<float4*>(buf);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int p = 0; p < passes; ++p)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            float4 v = b[i];
            v.x *= one; v.y *= one; v.z *= one; v.w *= one;
            b[i] = v;
        }
}
}  // namespace

extern "C" int dlsg_comm_rehearsal(float* buf, int64_t count, int workgroups, int passes, void* stream) {
    if (count < 0 || workgroups < 1 || workgroups > 1024 || passes < 0 || (count > 0 && !buf)) return DLSG_EINVAL;
    if (reinterpret_cast<uintptr_t>(buf) & 15) return DLSG_EALIGN;
    if (count < 4 || passes == 0) return DLSG_OK;
    hipLaunchKernelGGL(comm_rehearsal_kernel, dim3(workgroups), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), buf, count / 4, passes, 1.0f);
    return hipGetLastError() == hipSuccess ? DLSG_OK : DLSG_ELAUNCH;
}

// *code <- the communicator's asynchronous error state (ncclCommGetAsyncError: 0 = ncclSuccess); no synchronisation
extern "C" int dlsg_comm_async_error(dlsg_comm* c, int32_t* code) {
    if (!c || !c->comm) return DLSG_ENOCOMM;
    if (!code) return DLSG_EINVAL;
    ncclResult_t st = ncclSuccess;
    const ncclResult_t r = c->api.CommGetAsyncError(c->comm, &st);
    *code = static_cast<int32_t>(st);
    return r == ncclSuccess ? DLSG_OK : DLSG_ELAUNCH;
}
