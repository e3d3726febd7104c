/*
 * irmv_comm.h -- C ABI of libirmv_comm.so: the ONE collective of the multi-GPU path.
 *
 * Frames are independent: every GPU holds a replica of the engine, frame i belongs to GPU
 * i mod N, and the only data that crosses xGMI is the weight blob, broadcast once at start-up so that only rank 0
 * reads or generates it.  The reference itself is single-device (test/yolo_test.cpp:16, src/irm_detector.cpp:35-38:
 * three engines on device 0); this library is what a multi-camera node would add:
 *
 *   single process, one thread + one engine per GPU   irmv_comm_init_all      (ncclCommInitAll)
 *   one process per GPU (launcher sets RANK / ...)    irmv_comm_unique_id + irmv_comm_init_rank (ncclCommInitRank)
 *
 * It links against /opt/rocm/lib/librccl.so and libamdhip64 only: no torch, no Python.  libirmv_hip.so does not depend
 * on it (an engine takes the broadcast blob as a device pointer: irmv_engine_cfg::weights_on_device).
 *
 * Every entry returns IRMV_OK (0) or a negative code of irmv_hip.h; irmv_comm_last_error() is thread-local.
 */
#ifndef IRMV_COMM_H
#define IRMV_COMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRMV_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */

typedef struct irmv_comm irmv_comm;

const char *irmv_comm_last_error(void);

/* One process, `ndev` GPUs (devices[i] = HIP ordinal of local rank i; NULL = 0..ndev-1).  The communicator owns one
 * RCCL rank and one stream per device. */
int irmv_comm_init_all(int ndev, const int *devices, irmv_comm **out);

/* One process per GPU: rank 0 makes the id, the launcher's side channel (a file, a socket: 128 bytes) carries it to the
 * other ranks, every rank then joins with its HIP device. */
int irmv_comm_unique_id(uint8_t id[IRMV_COMM_ID_BYTES]);
int irmv_comm_init_rank(const uint8_t id[IRMV_COMM_ID_BYTES], int nranks, int rank, int device, irmv_comm **out);

int irmv_comm_nranks(const irmv_comm *c);       /* ranks of the whole job */
int irmv_comm_local_ranks(const irmv_comm *c);  /* ranks this process owns (init_all: ndev, init_rank: 1) */

/* The weight broadcast.  `host_blob` / `bytes` matter on the root only (every local rank passes the same `bytes`; ranks
 * of other processes learn it from the 8-byte size broadcast in front of the payload).  On return dev_ptrs[i] is a device
 * buffer on local rank i's GPU holding the blob (owned by the communicator, valid until irmv_comm_destroy) and
 * *bytes_out its size: exactly what irmv_engine_cfg::{weights_blob, weights_bytes, weights_on_device = 1} take. */
int irmv_comm_broadcast_blob(irmv_comm *c, const void *host_blob, uint64_t bytes, int root, void **dev_ptrs, uint64_t *bytes_out);

/* Job-wide reductions of one double per local rank (timing: max over ranks; counts: sum), in place; they double as a
 * barrier.  op: 0 = sum, 1 = max. */
int irmv_comm_allreduce_f64(irmv_comm *c, double *values, int op);

void irmv_comm_destroy(irmv_comm *c);

#ifdef __cplusplus
}
#endif
#endif
