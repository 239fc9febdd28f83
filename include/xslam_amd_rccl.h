/* xslam_amd_rccl.h — RCCL collectives for the sharded orchestrator, in C++ (libxslam_rccl.so), so that a C++ host runs
 * shard mode with no Python in the loop (north_star: "host stays C++ ... a single RCCL all-reduce over xGMI of the
 * per-frame 6x6 / 6x1 ICP normal equations").  The reference is single-GPU, single-process and has no counterpart
 * ("Distributed communication backend: none"; section 8e).
 *
 * Use: rank 0 calls xs_rccl_get_unique_id and hands the 128 bytes to the other ranks by any means (a file, a socket, MPI,
 * torch.distributed.broadcast_object_list ...); every rank then calls xs_rccl_comm_create on its own GPU and passes
 *      xs_kf_create_sharded(yaml, rank, count, xs_rccl_collective, comm)            (include/xslam_amd_pipeline.h)
 * The orchestrator calls back with op 0 = sum of doubles (the 55 ICP sums, the 29 Gauss-Newton sums), 1 = min of int32
 * (first raycast event per pixel), 2 = sum of int32 (vertex / normal maps as bit patterns); each becomes one in-place
 * ncclAllReduce enqueued on the comm's stream — set it to the stream the orchestrator runs on (xs_kf_set_stream). */
#ifndef XSLAM_AMD_RCCL_H
#define XSLAM_AMD_RCCL_H
#ifdef __cplusplus
extern "C" {
#endif

#define XS_RCCL_UNIQUE_ID_BYTES 128

/* ncclGetUniqueId: fills id128 (XS_RCCL_UNIQUE_ID_BYTES).  0 on success. */
int xs_rccl_get_unique_id(void *id128);
/* ncclCommInitRank on the calling thread's current HIP device; stream = hipStream_t every collective is enqueued on
 * (NULL = default stream).  NULL on failure (xs_rccl_last_error). */
void *xs_rccl_comm_create(const void *id128, int rank, int count, void *stream);
void xs_rccl_set_stream(void *comm, void *stream);
int xs_rccl_comm_destroy(void *comm);
/* the collective callback xs_kf_create_sharded takes; user = the comm handle.  Errors print and exit(-1), like every
 * runtime failure of the reference (Common/include/cx.h:125-130). */
void xs_rccl_collective(void *user, int op, void *dev_ptr, long count);
/* the same with a status instead of exit: 0, or the ncclResult_t (ops 0 - 2: all-reduce) */
int xs_rccl_all_reduce(void *comm, int op, void *dev_ptr, long count);
/* op 3 (include/xslam_amd_pipeline.h): desc = host array {device buffer address, byte offsets of the count + 1 part boundaries}; every
 * rank's part is broadcast from its owner inside one ncclGroupStart / ncclGroupEnd */
int xs_rccl_gatherv(void *comm, const long long *desc, long count);
int xs_rccl_rank(void *comm);
int xs_rccl_count(void *comm);
/* ncclGetVersion */
int xs_rccl_version(void);
const char *xs_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* XSLAM_AMD_RCCL_H */
