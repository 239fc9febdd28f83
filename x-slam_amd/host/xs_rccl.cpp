// xs_rccl.cpp — the three collectives of the sharded orchestrator as RCCL calls (include/xslam_amd_rccl.h).
// One communicator per rank / GPU; every all-reduce is in place and stream-ordered, so the orchestrator's kernels and the
// collectives queue on one stream without host waits (the host waits only where it needs a value: the ICP sums).
#include "../../include/xslam_amd_rccl.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <string>

static_assert(sizeof(ncclUniqueId) == XS_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");

namespace {
thread_local std::string g_err;
struct Comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, count = 1;
};
int fail(ncclResult_t r, const char *what) {
    g_err = std::string(what) + ": " + ncclGetErrorString(r);
    return (int)r;
}
}  // namespace

extern "C" {

const char *xs_rccl_last_error(void) { return g_err.c_str(); }

int xs_rccl_version(void) {
    int v = 0;
    return ncclGetVersion(&v) == ncclSuccess ? v : -1;
}

int xs_rccl_get_unique_id(void *id128) {
    if (!id128) { g_err = "xs_rccl_get_unique_id: null pointer"; return -1; }
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(r, "ncclGetUniqueId");
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

void *xs_rccl_comm_create(const void *id128, int rank, int count, void *stream) {
    if (!id128 || count < 1 || rank < 0 || rank >= count) { g_err = "xs_rccl_comm_create: bad arguments"; return nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    Comm *c = new Comm;
    c->rank = rank; c->count = count; c->stream = (hipStream_t)stream;
    const ncclResult_t r = ncclCommInitRank(&c->comm, count, id, rank);
    if (r != ncclSuccess) { fail(r, "ncclCommInitRank"); delete c; return nullptr; }
    return c;
}

void xs_rccl_set_stream(void *comm, void *stream) { if (comm) static_cast<Comm *>(comm)->stream = (hipStream_t)stream; }
int xs_rccl_rank(void *comm) { return comm ? static_cast<Comm *>(comm)->rank : -1; }
int xs_rccl_count(void *comm) { return comm ? static_cast<Comm *>(comm)->count : -1; }

int xs_rccl_comm_destroy(void *comm) {
    if (!comm) return 0;
    Comm *c = static_cast<Comm *>(comm);
    const ncclResult_t r = ncclCommDestroy(c->comm);
    delete c;
    return r == ncclSuccess ? 0 : fail(r, "ncclCommDestroy");
}

int xs_rccl_all_reduce(void *comm, int op, void *dev_ptr, long count) {
    if (!comm || !dev_ptr || count < 0) { g_err = "xs_rccl_all_reduce: bad arguments"; return -1; }
    if (count == 0) return 0;
    Comm *c = static_cast<Comm *>(comm);
    ncclDataType_t dt; ncclRedOp_t ro;
    switch (op) {
        case 0: dt = ncclDouble; ro = ncclSum; break;   // ICP normal equations (55), Gauss-Newton sums (29), Hessian sums (4)
        case 1: dt = ncclInt32; ro = ncclMin; break;    // first raycast event per pixel
        case 2: dt = ncclInt32; ro = ncclSum; break;    // the ranks' owned-pixel counts (or, shard_composite_gather: false, the maps as bit patterns)
        default: g_err = "xs_rccl_all_reduce: bad op"; return -1;
    }
    const ncclResult_t r = ncclAllReduce(dev_ptr, dev_ptr, (size_t)count, dt, ro, c->comm, c->stream);
    return r == ncclSuccess ? 0 : fail(r, "ncclAllReduce");
}

// op 3 of the orchestrator's callback: the ranks' variable-size parts of one device buffer, each broadcast from its owner — one group
// call, so RCCL schedules the N broadcasts together.  desc: host array {buffer address, offset[0], .., offset[count]} (bytes).
int xs_rccl_gatherv(void *comm, const long long *desc, long count) {
    if (!comm || !desc) { g_err = "xs_rccl_gatherv: bad arguments"; return -1; }
    Comm *c = static_cast<Comm *>(comm);
    if (count != c->count) { g_err = "xs_rccl_gatherv: the descriptor is for another number of ranks"; return -1; }
    char *base = reinterpret_cast<char *>((size_t)desc[0]);
    ncclResult_t r = ncclGroupStart();
    if (r != ncclSuccess) return fail(r, "ncclGroupStart");
    for (int k = 0; k < c->count; ++k) {
        const long long a = desc[1 + k], b = desc[2 + k];
        if (b <= a) continue;
        r = ncclBroadcast(base + a, base + a, (size_t)(b - a), ncclChar, k, c->comm, c->stream);
        if (r != ncclSuccess) { (void)ncclGroupEnd(); return fail(r, "ncclBroadcast"); }
    }
    r = ncclGroupEnd();
    return r == ncclSuccess ? 0 : fail(r, "ncclGroupEnd");
}

void xs_rccl_collective(void *user, int op, void *dev_ptr, long count) {
    if ((op == 3 ? xs_rccl_gatherv(user, static_cast<const long long *>(dev_ptr), count) : xs_rccl_all_reduce(user, op, dev_ptr, count)) != 0) {
        printf("RCCL error(%s)\n", g_err.c_str());
        exit(-1);
    }
}

}  // extern "C"
