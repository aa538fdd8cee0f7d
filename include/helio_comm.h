/*
 * helio_comm.h — C ABI of libhelio_comm.so: the one exchange step of the sharded render,
 * an RCCL all-gather of image shards over xGMI.
 *
 * The reference (l3th4l/DOODLE) has no multi-GPU code, so there is no
 * reference interface to mirror; this is the build's own boundary for
 * doodle_amd/sharded.py.  One process per GPU.  All pointers *_d are device memory owned by
 * the caller; calls enqueue on `stream` (hipStream_t as void*) and do not synchronise.
 * Return: 0 (or a positive byte count where stated) on success, negative HELIO_E_* on error.
 */
#ifndef HELIO_COMM_H
#define HELIO_COMM_H

#ifdef __cplusplus
extern "C" {
#endif

#define HELIO_COMM_ID_BYTES 128

/* Rank 0: create the communicator's unique id into out[len >= 128]; returns the id size.
 * The host program distributes the bytes to the other ranks (any channel). */
int helio_comm_unique_id(char *out, int len);
/* Every rank (collective): join the communicator. */
int helio_comm_init(void **comm, int nranks, int rank, const char *id_bytes, int len);
/* recv_d[r*count : (r+1)*count] = send_d of rank r, for every r (ncclAllGather, fp32). */
int helio_comm_allgather_f32(void *comm, const float *send_d, float *recv_d, long count, void *stream);
/* The size of the communicator and this process's rank in it, as RCCL reports them
 * (ncclCommCount / ncclCommUserRank): what a benchmark records to show how many ranks the
 * collective really spanned. */
int helio_comm_count(void *comm, int *nranks, int *rank);
int helio_comm_destroy(void *comm);

/*
 * Peer-store gather (the opt-in "p2p" transport of doodle_amd/comm.py; correctness only — no performance claim
 * until an 8-GPU node has run it).  A rank's receive buffer is device memory made by helio_p2p_alloc (hipMalloc +
 * its 64-byte IPC handle), mapped by every other rank with helio_p2p_open.  helio_p2p_scatter_f32 enqueues ONE
 * kernel on `stream` that stores send_d[0:count] into peer_bufs[p][rank*count : (rank+1)*count] for every p in
 * [0, world) — its own buffer included — and then, behind a system-scope fence, stores `epoch` into
 * flags_d[p*world + rank] for every p.  flags_d is the device view (helio_p2p_register_host) of a
 * [world][world] int table in host memory shared by all processes of the node: rank p's HOST polls row p; no kernel
 * waits on a flag.  arrived_d: one zero-initialised device word owned by the caller (the kernel's workgroup count,
 * left at zero).  world <= 16; send_d and every buffer 16-byte aligned.
 */
int helio_p2p_alloc(long bytes, void **ptr, char *handle_out, int len);     /* → handle size (64) */
int helio_p2p_open(const char *handle, int len, void **ptr);
int helio_p2p_close(void *ptr);                                              /* a pointer from helio_p2p_open  */
int helio_p2p_free(void *ptr);                                               /* a pointer from helio_p2p_alloc */
int helio_p2p_register_host(void *host, long bytes, void **dev);
int helio_p2p_unregister_host(void *host);
int helio_p2p_scatter_f32(const float *send_d, long count, int rank, int world, void *const *peer_bufs, int *flags_d,
                          int epoch, unsigned *arrived_d, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HELIO_COMM_H */
