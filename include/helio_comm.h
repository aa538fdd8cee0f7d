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

#ifdef __cplusplus
}
#endif
#endif /* HELIO_COMM_H */
