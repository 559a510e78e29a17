/*
 * dvs_comm.h — the thin C-ABI over RCCL for the data-parallel training step: one process per GPU, every rank a
 * full replica of the splat parameter block, independent views sharded over the ranks, ONE exchange per iteration.
 *
 * What this replaces in the reference (fenghuayumo/DIVSHOT): nothing — the reference trainer is single-GPU and holds no collective
 * call site (no nccl / rccl / mpi / gloo anywhere in its tree). This is the new functionality BASELINE.json's
 * north_star asks for ("views shard one-per-GPU ... RCCL all-reduce of splat gradients over xGMI; host code stays C++").
 * libgstrain.so uses it inside train_step() when WORLD_SIZE > 1 (divshot_amd/gstrain/gstrain.cpp).
 *
 * librccl is opened with dlopen() when the first communicator is created, so libdvsraster.so has no load-time dependency on it.
 * Bootstrap: rank 0 creates the RCCL unique id and serves it over TCP on the rendezvous address every launcher exports as MASTER_ADDR,
 * on a DEDICATED port (the launcher's own store already listens on MASTER_PORT): DVS_COMM_PORT if set, else MASTER_PORT + 1789. A
 * peer introduces itself with {magic, job nonce, rank}, receives the id, acknowledges it and waits for rank 0's one-byte confirmation
 * (three-way: a rank counts on rank 0 exactly when it returns "ok" itself — one that timed out on any step simply asks again and is
 * answered again; it counts once); rank 0 ignores anything else, and every step has a timeout (DVS_COMM_TIMEOUT_S, default 180 s: an
 * error, never a hang). Plain C, int status codes as dvs_raster.h.
 *
 * DVS_COMM_BACKEND=tcp selects a TEST-ONLY backend: the same calls, executed host-staged over the bootstrap sockets (star on rank 0,
 * sums formed in rank order, every call synchronous). It needs neither librccl nor distinct devices per rank, which is the point: two
 * ranks of libgstrain.so can share the one GPU of a test box (tests/test_gpu_multirank.py). It is announced on stderr when created and
 * must never be used for measurements. With device = -1 it touches no HIP at all and the collectives take HOST buffers: the logic of
 * every collective is then testable on a machine without a GPU (tests/test_comm_bootstrap.py, three ranks).
 */
#ifndef DVS_COMM_H
#define DVS_COMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dvs_comm dvs_comm;

/* Create the communicator of this process on HIP device `device`. world < 1: rank / world are read from the environment (RANK,
 * WORLD_SIZE); master_addr NULL / master_port <= 0: MASTER_ADDR (default 127.0.0.1) / MASTER_PORT (default 29500) — each falls back
 * on its own. A 1-rank communicator is valid (every collective is then the identity, executed by RCCL all the same). Returns NULL on
 * failure (dvs_last_error). Blocks until all ranks have joined or the bootstrap deadline passes. */
dvs_comm* dvs_comm_create(int device, int rank, int world, const char* master_addr, int master_port);
/* The bootstrap step alone (no RCCL, no GPU): rank 0 hands the 128 bytes at id128 to every other rank, which receive them into id128.
 * Same argument / environment rules as dvs_comm_create. DVS_OK, or DVS_ERR_STATE on a timeout / bind failure (dvs_last_error). */
int dvs_comm_bootstrap(int rank, int world, const char* master_addr, int master_port, void* id128);
void      dvs_comm_destroy(dvs_comm* comm);
int       dvs_comm_rank(const dvs_comm* comm);
int       dvs_comm_world(const dvs_comm* comm);
/* What the BACKEND says: the size the RCCL communicator reports (ncclCommCount; -1 if this librccl has no such call), or — test backend —
 * the ranks rank 0 holds a socket to (+ itself); on the OTHER ranks of the test backend it is the world size they were started with,
 * unverified (only rank 0 sees every socket). bench.py prints rank 0's value as `rccl_nranks`: "RCCL saw N ranks" readable without the source. */
int       dvs_comm_backend_ranks(const dvs_comm* comm);
const char* dvs_comm_backend_name(const dvs_comm* comm);      /* "rccl" | "tcp (host-staged TEST backend)" */

/* Collectives on DEVICE buffers, enqueued on `stream` (hipStream_t as void*), asynchronous. Counts are in elements. */
int dvs_comm_all_reduce_sum_f32(dvs_comm* comm, void* stream, float* buf, size_t count);                 /* in place */
int dvs_comm_all_reduce_max_i32(dvs_comm* comm, void* stream, int32_t* buf, size_t count);               /* in place */
int dvs_comm_reduce_scatter_sum_f32(dvs_comm* comm, void* stream, const float* send /*[world*recv_count]*/, float* recv, size_t recv_count);
int dvs_comm_all_gather_f32(dvs_comm* comm, void* stream, const float* send, float* recv /*[world*send_count]*/, size_t send_count);
int dvs_comm_broadcast(dvs_comm* comm, void* stream, void* buf, size_t bytes, int root);
/* The collectives between a group_start and its group_end are issued as ONE launch (ncclGroupStart / ncclGroupEnd): the four ranges a
 * splat chunk's geometry gradients occupy in the flat buffer leave as one operation instead of four small ones. */
int dvs_comm_group_start(dvs_comm* comm);
int dvs_comm_group_end(dvs_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* DVS_COMM_H */
