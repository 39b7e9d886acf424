// libyolo2comm.so: RCCL communicator lifecycle + the data-parallel collectives of the training step behind the C ABI of
// include/yolo2_comm.h.  Plain host code (no kernels): RCCL launches its own.  Built by csrc/build.py against /opt/rocm's librccl.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/yolo2_comm.h"

static_assert(sizeof(ncclUniqueId) == YOLO2_COMM_ID_BYTES, "ncclUniqueId size");

struct yolo2_comm {
    ncclComm_t nccl;
    int rank, world, device;
};

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char *yolo2_comm_last_error(void) { return g_err; }

#define COMM_RCCL(call, what)                                                                                        \
    do {                                                                                                             \
        ncclResult_t r_ = (call);                                                                                    \
        if (r_ != ncclSuccess) return fail(YOLO2_COMM_E_RCCL, "%s: %s: %s", __func__, what, ncclGetErrorString(r_)); \
    } while (0)
#define COMM_HIP(call, what)                                                                                       \
    do {                                                                                                           \
        hipError_t e_ = (call);                                                                                    \
        if (e_ != hipSuccess) return fail(YOLO2_COMM_E_HIP, "%s: %s: %s", __func__, what, hipGetErrorString(e_)); \
    } while (0)

extern "C" int yolo2_comm_unique_id(void *id) {
    if (!id) return fail(YOLO2_COMM_E_ARG, "yolo2_comm_unique_id: id is NULL");
    ncclUniqueId u;
    COMM_RCCL(ncclGetUniqueId(&u), "ncclGetUniqueId");
    memcpy(id, &u, sizeof(u));
    return YOLO2_COMM_OK;
}

extern "C" int yolo2_comm_init(yolo2_comm **comm, const void *id, int rank, int world, int device) {
    if (!comm || !id || world < 1 || rank < 0 || rank >= world || device < 0)
        return fail(YOLO2_COMM_E_ARG, "yolo2_comm_init: comm / id NULL, or rank %d not in [0, %d), or device %d < 0", rank, world, device);
    *comm = nullptr;
    COMM_HIP(hipSetDevice(device), "hipSetDevice");
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    yolo2_comm *c = new (std::nothrow) yolo2_comm{nullptr, rank, world, device};
    if (!c) return fail(YOLO2_COMM_E_ARG, "yolo2_comm_init: out of host memory");
    ncclResult_t r = ncclCommInitRank(&c->nccl, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(YOLO2_COMM_E_RCCL, "yolo2_comm_init: ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, device, ncclGetErrorString(r));
    }
    *comm = c;
    return YOLO2_COMM_OK;
}

extern "C" int yolo2_comm_rank(const yolo2_comm *comm) { return comm ? comm->rank : -1; }
extern "C" int yolo2_comm_world(const yolo2_comm *comm) { return comm ? comm->world : -1; }

extern "C" int yolo2_comm_allreduce_bucket(yolo2_comm *comm, void *buf, long count, int dtype, void *stream) {
    if (!comm || !buf || count <= 0 || (dtype != YOLO2_COMM_F32 && dtype != YOLO2_COMM_BF16))
        return fail(YOLO2_COMM_E_ARG, "yolo2_comm_allreduce_bucket: comm / buf NULL, count %ld <= 0, or dtype %d", count, dtype);
    COMM_RCCL(ncclAllReduce(buf, buf, (size_t)count, dtype == YOLO2_COMM_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, comm->nccl, (hipStream_t)stream),
              "ncclAllReduce");
    return YOLO2_COMM_OK;
}

// Optimizer sharding (parallel.GradReducer shard_params): buf holds world * shard_count elements; every rank keeps the sum over ranks of
// ITS shard (elements [rank * shard_count, (rank + 1) * shard_count)) in place, the other shards are left as they were
extern "C" int yolo2_comm_reduce_scatter_bucket(yolo2_comm *comm, void *buf, long shard_count, int dtype, void *stream) {
    if (!comm || !buf || shard_count <= 0 || (dtype != YOLO2_COMM_F32 && dtype != YOLO2_COMM_BF16))
        return fail(YOLO2_COMM_E_ARG, "yolo2_comm_reduce_scatter_bucket: comm / buf NULL, shard_count %ld <= 0, or dtype %d", shard_count, dtype);
    const size_t esz = dtype == YOLO2_COMM_F32 ? 4 : 2;
    char *own = (char *)buf + (size_t)comm->rank * (size_t)shard_count * esz;       // in place: recvbuff = sendbuff + rank * recvcount
    COMM_RCCL(ncclReduceScatter(buf, own, (size_t)shard_count, dtype == YOLO2_COMM_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, comm->nccl, (hipStream_t)stream),
              "ncclReduceScatter");
    return YOLO2_COMM_OK;
}

// ... and the way back: every rank's shard (updated parameters) becomes visible in every rank's buf
extern "C" int yolo2_comm_allgather(yolo2_comm *comm, void *buf, long shard_bytes, void *stream) {
    if (!comm || !buf || shard_bytes <= 0) return fail(YOLO2_COMM_E_ARG, "yolo2_comm_allgather: comm / buf NULL or shard_bytes %ld <= 0", shard_bytes);
    const char *own = (const char *)buf + (size_t)comm->rank * (size_t)shard_bytes;  // in place: sendbuff = recvbuff + rank * sendcount
    COMM_RCCL(ncclAllGather(own, buf, (size_t)shard_bytes, ncclUint8, comm->nccl, (hipStream_t)stream), "ncclAllGather");
    return YOLO2_COMM_OK;
}

extern "C" int yolo2_comm_broadcast(yolo2_comm *comm, void *buf, long bytes, int root, void *stream) {
    if (!comm || !buf || bytes <= 0 || root < 0 || root >= comm->world)
        return fail(YOLO2_COMM_E_ARG, "yolo2_comm_broadcast: comm / buf NULL, bytes %ld <= 0, or root %d", bytes, root);
    COMM_RCCL(ncclBroadcast(buf, buf, (size_t)bytes, ncclUint8, root, comm->nccl, (hipStream_t)stream), "ncclBroadcast");
    return YOLO2_COMM_OK;
}

extern "C" int yolo2_comm_agree_max(yolo2_comm *comm, int value, int *result, void *scratch_dev4, void *stream) {
    if (!comm || !result || !scratch_dev4) return fail(YOLO2_COMM_E_ARG, "yolo2_comm_agree_max: comm / result / scratch NULL");
    hipStream_t st = (hipStream_t)stream;
    COMM_HIP(hipMemcpyAsync(scratch_dev4, &value, sizeof(int), hipMemcpyHostToDevice, st), "upload");
    COMM_RCCL(ncclAllReduce(scratch_dev4, scratch_dev4, 1, ncclInt32, ncclMax, comm->nccl, st), "ncclAllReduce(max)");
    COMM_HIP(hipMemcpyAsync(result, scratch_dev4, sizeof(int), hipMemcpyDeviceToHost, st), "download");
    COMM_HIP(hipStreamSynchronize(st), "hipStreamSynchronize");
    return YOLO2_COMM_OK;
}

extern "C" int yolo2_comm_destroy(yolo2_comm *comm) {
    if (!comm) return YOLO2_COMM_OK;
    ncclResult_t r = ncclCommDestroy(comm->nccl);
    delete comm;
    if (r != ncclSuccess) return fail(YOLO2_COMM_E_RCCL, "yolo2_comm_destroy: %s", ncclGetErrorString(r));
    return YOLO2_COMM_OK;
}
