// RCCL (xGMI) all-gather of rendered image shards, behind the C ABI.
//
// The render path shards over sun positions and has exactly one exchange step: every rank
// contributes its [B/G, R, R] image block and receives all G blocks.  This file
// wraps the three RCCL calls that needs so that the collective can be enqueued on a caller
// chosen HIP stream (a side stream, overlapping the next render) with a few microseconds of
// host time, instead of going through a framework process-group object per call.
// One process per GPU; the 128-byte unique id is created on rank 0 and distributed by the
// host program (doodle_amd/comm.py broadcasts it with torch.distributed).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstring>
#include "helio.h"
#include "helio_comm.h"

extern "C" {

int helio_comm_unique_id(char* out, int len) {
    if (!out || len < (int)sizeof(ncclUniqueId)) return HELIO_E_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return HELIO_E_LAUNCH;
    memcpy(out, &id, sizeof(id));
    return (int)sizeof(id);
}

int helio_comm_init(void** comm, int nranks, int rank, const char* id_bytes, int len) {
    if (!comm || !id_bytes || len < (int)sizeof(ncclUniqueId) || nranks < 1 || rank < 0 || rank >= nranks)
        return HELIO_E_INVALID;
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t c;
    if (ncclCommInitRank(&c, nranks, id, rank) != ncclSuccess) return HELIO_E_LAUNCH;
    *comm = c;
    return HELIO_OK;
}

int helio_comm_allgather_f32(void* comm, const float* send_d, float* recv_d, long count, void* stream) {
    if (!comm || !send_d || !recv_d || count < 0) return HELIO_E_INVALID;
    ncclResult_t r = ncclAllGather(send_d, recv_d, (size_t)count, ncclFloat32, static_cast<ncclComm_t>(comm),
                                   static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? HELIO_OK : HELIO_E_LAUNCH;
}

int helio_comm_count(void* comm, int* nranks, int* rank) {
    if (!comm || !nranks || !rank) return HELIO_E_INVALID;
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    if (ncclCommCount(c, nranks) != ncclSuccess || ncclCommUserRank(c, rank) != ncclSuccess) return HELIO_E_LAUNCH;
    return HELIO_OK;
}

int helio_comm_destroy(void* comm) {
    if (!comm) return HELIO_E_INVALID;
    return ncclCommDestroy(static_cast<ncclComm_t>(comm)) == ncclSuccess ? HELIO_OK : HELIO_E_LAUNCH;
}

}  // extern "C"
