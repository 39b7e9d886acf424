/* yolo2_comm.h -- the data-parallel exchange of the YOLOv2 training step behind a C ABI: one RCCL communicator per process
 * (one process per GPU), sum-all-reduce of gradient buckets over xGMI, broadcast of the initial replica.
 *
 * What it replaces.  The reference trains on one device (train.py:127-129 builds ONE train_op; README.md:99 lists multi-GPU as
 * unchecked); SURVEY 8b asks for the communicator lifecycle "init / allreduce_bucket / destroy" as part of the boundary so that a
 * C / C++ host can run the data-parallel step of BASELINE configs[2] without Python.  The Python host of this tree
 * (yolo_tf_amd/parallel.py) keeps using torch.distributed -- the same RCCL -- and does NOT load this library: a process may map only
 * one HIP runtime and one RCCL (DESIGN.md "Packaging notes"), and torch ships its own copies.  Hence a separate shared object,
 * libyolo2comm.so, linked against /opt/rocm's librccl + libamdhip64 -- the runtime a C / C++ host links anyway.
 *
 * Protocol (mirrors parallel.GradReducer): rank 0 calls yolo2_comm_unique_id and hands the 128 bytes to every rank by any
 * out-of-band means (file, socket, MPI, environment); every rank calls yolo2_comm_init; after loading / initialising the model,
 * yolo2_comm_broadcast(params, root 0) makes the replicas identical (train.py's sync_replicas); per step, as backward passes each
 * bucket boundary of the gradient arena (yolo_tf_amd/parallel.make_buckets: >= 64 MiB, variable-aligned), the host records an event
 * on the compute stream, makes the communication stream wait for it and calls yolo2_comm_allreduce_bucket on that stream; the
 * optimizer (yolo2_adam & co. with gscale = 1 / world) waits for the bucket's event.  bf16 wire format: cast the bucket with
 * yolo2_cast_f32_bf16 (libyolo2hip.so), reduce it with dtype YOLO2_COMM_BF16, cast back with yolo2_cast_bf16_f32.
 *
 * All calls are asynchronous on the given hipStream_t; buffers are caller-owned device memory; nothing here allocates device
 * memory.  Return value: 0 = OK, else one of the codes below; yolo2_comm_last_error() (thread-local) says what failed. */
#ifndef YOLO2_COMM_H
#define YOLO2_COMM_H
#ifdef __cplusplus
extern "C" {
#endif

enum { YOLO2_COMM_OK = 0, YOLO2_COMM_E_ARG = 1, YOLO2_COMM_E_RCCL = 2, YOLO2_COMM_E_HIP = 3 };
enum { YOLO2_COMM_F32 = 0, YOLO2_COMM_BF16 = 1 };      /* same values as YOLO2_F32 / YOLO2_BF16 of yolo2_hip.h */
#define YOLO2_COMM_ID_BYTES 128

typedef struct yolo2_comm yolo2_comm;

/* rank 0: a fresh rendezvous id (ncclUniqueId), YOLO2_COMM_ID_BYTES bytes into id */
int yolo2_comm_unique_id(void *id);
/* every rank: joins the communicator described by id on HIP device `device` (hipSetDevice is called).  Collective. */
int yolo2_comm_init(yolo2_comm **comm, const void *id, int rank, int world, int device);
int yolo2_comm_rank(const yolo2_comm *comm);
int yolo2_comm_world(const yolo2_comm *comm);
/* in-place sum over all ranks of `count` elements at buf (one gradient bucket), enqueued on `stream` */
int yolo2_comm_allreduce_bucket(yolo2_comm *comm, void *buf, long count, int dtype, void *stream);
/* Optimizer sharding ([mi355x] shard_optimizer; yolo_tf_amd/parallel.py GradReducer shard_params): instead of all-reducing a bucket and
 * running the same optimizer pass on every rank, reduce-scatter it (rank r keeps the sum of elements [r * shard_count, (r+1) * shard_count)
 * of buf, in place), update that shard alone (yolo2_adam & co. over the slice, gscale = 1 / world) and all-gather the updated PARAMETER
 * shards (buf = the same range of the parameter arena, shard_bytes = shard_count * 4).  The host cuts every bucket into world equal
 * shards of a multiple of 64 elements; the remainder (< world * 64 elements) goes through yolo2_comm_allreduce_bucket and stays
 * replicated.  Same bytes on the wire as the all-reduce; the optimizer pass shrinks to 1 / world per rank. */
int yolo2_comm_reduce_scatter_bucket(yolo2_comm *comm, void *buf, long shard_count, int dtype, void *stream);
int yolo2_comm_allgather(yolo2_comm *comm, void *buf, long shard_bytes, void *stream);
/* `bytes` bytes at buf become root's on every rank (initial replica: parameters, moving statistics, optimizer slots) */
int yolo2_comm_broadcast(yolo2_comm *comm, void *buf, long bytes, int root, void *stream);
/* a host-side decision every rank must take together (non-finite loss, a corrupt shard seen by one rank): the maximum of
 * `value` over ranks, returned to all; synchronises the given stream.  scratch = 4 bytes of device memory. */
int yolo2_comm_agree_max(yolo2_comm *comm, int value, int *result, void *scratch_dev4, void *stream);
/* drains nothing, waits for nothing: the caller synchronises its streams first.  NULL is allowed. */
int yolo2_comm_destroy(yolo2_comm *comm);
const char *yolo2_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
