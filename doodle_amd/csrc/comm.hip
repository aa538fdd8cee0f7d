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
#include <cstdint>
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

// ------------------------------------------------------------------------------------------------------------
// Peer-store gather (opt-in transport "p2p"; correctness only until an 8-GPU box has run it — DESIGN.md §5).
// Every rank owns a receive buffer of `world` slots and maps the other ranks' buffers once (HIP IPC: the dmabuf
// handles of hipIpcGetMemHandle); a gather is then ONE kernel per rank that reads its shard once and stores it into
// slot `rank` of EVERY rank's buffer — seven xGMI links written at the same time, no ring, no intermediate copy —
// and publishes "rank r's shard of step e has landed" to a table of flags in host memory that all processes of
// the node share (POSIX shared memory, registered with HIP by each process): the stores, a system-scope fence, the
// last workgroup's flag stores.  The receiver's HOST polls its row of the table (doodle_amd/comm.py): no kernel
// ever spins on a flag, so a rank that never arrives is a timeout in Python, not a hung wave.
// ------------------------------------------------------------------------------------------------------------
namespace {

constexpr int P2P_MAX_RANKS = 16;
struct P2PPeers { float* buf[P2P_MAX_RANKS]; };

__global__ void __launch_bounds__(256)
p2p_scatter_kernel(const float* __restrict__ local, long n, int rank, int world, P2PPeers peers, int* __restrict__ flags,
                   int epoch, unsigned* __restrict__ arrived) {
    const long n4 = (n & 3) ? 0 : n >> 2;            // 16-byte stores where every slot starts on a 16-byte boundary, else dwords
    const long tid = blockIdx.x * 256l + threadIdx.x, stride = (long)gridDim.x * 256;
    const float4* __restrict__ src = reinterpret_cast<const float4*>(local);
    for (long i = tid; i < n4; i += stride) {
        const float4 v = src[i];
        for (int p = 0; p < world; ++p) reinterpret_cast<float4*>(peers.buf[p] + (long)rank * n)[i] = v;
    }
    for (long i = 4 * n4 + tid; i < n; i += stride) {                     // (count % 4 != 0: the whole shard)
        const float v = local[i];
        for (int p = 0; p < world; ++p) peers.buf[p][(long)rank * n + i] = v;
    }
    // this workgroup's stores are visible system-wide before it is counted; the last one to be counted publishes
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next launch (stream-ordered)
            __threadfence_system();
            for (int p = 0; p < world; ++p)
                __hip_atomic_store(flags + (long)p * world + rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace

extern "C" {

int helio_p2p_alloc(long bytes, void** ptr, char* handle_out, int len) {
    if (bytes <= 0 || !ptr || !handle_out || len < (int)sizeof(hipIpcMemHandle_t)) return HELIO_E_INVALID;
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)bytes) != hipSuccess) return HELIO_E_LAUNCH;
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, p) != hipSuccess) { (void)hipFree(p); return HELIO_E_LAUNCH; }
    memcpy(handle_out, &h, sizeof(h));
    *ptr = p;
    return (int)sizeof(h);
}

int helio_p2p_open(const char* handle, int len, void** ptr) {
    if (!handle || len < (int)sizeof(hipIpcMemHandle_t) || !ptr) return HELIO_E_INVALID;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    return hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? HELIO_OK : HELIO_E_LAUNCH;
}

int helio_p2p_close(void* ptr) { return ptr && hipIpcCloseMemHandle(ptr) == hipSuccess ? HELIO_OK : HELIO_E_LAUNCH; }
int helio_p2p_free(void* ptr) { return ptr && hipFree(ptr) == hipSuccess ? HELIO_OK : HELIO_E_LAUNCH; }

int helio_p2p_register_host(void* host, long bytes, void** dev) {
    if (!host || bytes <= 0 || !dev) return HELIO_E_INVALID;
    if (hipHostRegister(host, (size_t)bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) return HELIO_E_LAUNCH;
    if (hipHostGetDevicePointer(dev, host, 0) != hipSuccess) { (void)hipHostUnregister(host); return HELIO_E_LAUNCH; }
    return HELIO_OK;
}

int helio_p2p_unregister_host(void* host) { return host && hipHostUnregister(host) == hipSuccess ? HELIO_OK : HELIO_E_LAUNCH; }

int helio_p2p_scatter_f32(const float* local_d, long count, int rank, int world, void* const* peer_bufs, int* flags_d,
                          int epoch, unsigned* arrived_d, void* stream) {
    if (!local_d || count < 0 || world < 1 || world > P2P_MAX_RANKS || rank < 0 || rank >= world || !peer_bufs || !flags_d ||
        !arrived_d || (reinterpret_cast<uintptr_t>(local_d) & 15))
        return HELIO_E_INVALID;
    P2PPeers peers{};
    for (int p = 0; p < world; ++p) {
        if (!peer_bufs[p] || (reinterpret_cast<uintptr_t>(peer_bufs[p]) & 15)) return HELIO_E_INVALID;
        peers.buf[p] = static_cast<float*>(peer_bufs[p]);
    }
    const long n4 = (count + 3) >> 2;
    const unsigned blocks = (unsigned)(n4 <= 0 ? 1 : (n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
    hipLaunchKernelGGL(p2p_scatter_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), local_d, count, rank,
                       world, peers, flags_d, epoch, arrived_d);
    return hipGetLastError() == hipSuccess ? HELIO_OK : HELIO_E_LAUNCH;
}

}  // extern "C"
