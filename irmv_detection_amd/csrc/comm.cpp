// libirmv_comm.so: the weight broadcast of the multi-GPU path over RCCL (include/irmv_comm.h).
//
// The reference runs on one device (test/yolo_test.cpp:16); SURVEY.md section 8e / BASELINE configs[3] shard independent
// frames over the GPUs of a node with replicas of the engine and ONE collective: the 6 MB weight blob leaves rank 0 once
// (ncclBroadcast over xGMI).  Nothing here is on the per-frame path.
#include "irmv_comm.h"
#include "irmv_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_cerr;
static int cfail(int code, const std::string &msg)
{
    g_cerr = msg;
    return code;
}
#define C_HIP(x)                                                                                           \
    do {                                                                                                   \
        hipError_t e_ = (x);                                                                               \
        if (e_ != hipSuccess) return cfail(IRMV_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));  \
    } while (0)
#define C_NCCL(x)                                                                                          \
    do {                                                                                                   \
        ncclResult_t r_ = (x);                                                                             \
        if (r_ != ncclSuccess) return cfail(IRMV_ERR_HIP, std::string(#x) + ": " + ncclGetErrorString(r_)); \
    } while (0)

struct irmv_comm {
    int nranks = 0;
    std::vector<int> devices;          // local ranks' HIP devices
    std::vector<ncclComm_t> comms;     // one per local rank
    std::vector<hipStream_t> streams;
    std::vector<void *> blobs;         // broadcast buffers (device), one per local rank
    std::vector<double *> scratch;     // one double per local rank (device) for the reductions
    std::vector<uint64_t *> sizes;     // one u64 per local rank (device) for the size broadcast
};

extern "C" const char *irmv_comm_last_error(void) { return g_cerr.c_str(); }

static int alloc_scratch(irmv_comm *c)
{
    const size_t n = c->devices.size();
    c->streams.assign(n, nullptr);
    c->scratch.assign(n, nullptr);
    c->sizes.assign(n, nullptr);
    c->blobs.assign(n, nullptr);
    for (size_t i = 0; i < n; i++) {
        C_HIP(hipSetDevice(c->devices[i]));
        C_HIP(hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking));
        C_HIP(hipMalloc((void **)&c->scratch[i], sizeof(double)));
        C_HIP(hipMalloc((void **)&c->sizes[i], sizeof(uint64_t)));
    }
    return IRMV_OK;
}

extern "C" void irmv_comm_destroy(irmv_comm *c)
{
    if (!c) return;
    for (size_t i = 0; i < c->devices.size(); i++) {
        (void)hipSetDevice(c->devices[i]);
        if (i < c->streams.size() && c->streams[i]) (void)hipStreamSynchronize(c->streams[i]);
        if (i < c->comms.size() && c->comms[i]) (void)ncclCommDestroy(c->comms[i]);
        if (i < c->blobs.size() && c->blobs[i]) (void)hipFree(c->blobs[i]);
        if (i < c->scratch.size() && c->scratch[i]) (void)hipFree(c->scratch[i]);
        if (i < c->sizes.size() && c->sizes[i]) (void)hipFree(c->sizes[i]);
        if (i < c->streams.size() && c->streams[i]) (void)hipStreamDestroy(c->streams[i]);
    }
    delete c;
}

extern "C" int irmv_comm_init_all(int ndev, const int *devices, irmv_comm **out)
{
    if (!out || ndev < 1 || ndev > 64) return cfail(IRMV_ERR_ARG, "irmv_comm_init_all: ndev must be 1..64, out non-null");
    int have = 0;
    C_HIP(hipGetDeviceCount(&have));
    irmv_comm *c = new irmv_comm;
    c->nranks = ndev;
    for (int i = 0; i < ndev; i++) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= have) { delete c; return cfail(IRMV_ERR_HIP, "irmv_comm_init_all: no such HIP device"); }
        c->devices.push_back(d);
    }
    c->comms.assign(ndev, nullptr);
    ncclResult_t r = ncclCommInitAll(c->comms.data(), ndev, c->devices.data());
    if (r != ncclSuccess) { irmv_comm_destroy(c); return cfail(IRMV_ERR_HIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(r)); }
    const int rc = alloc_scratch(c);
    if (rc) { irmv_comm_destroy(c); return rc; }
    *out = c;
    return IRMV_OK;
}

extern "C" int irmv_comm_unique_id(uint8_t id[IRMV_COMM_ID_BYTES])
{
    static_assert(IRMV_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id) return cfail(IRMV_ERR_ARG, "id is null");
    ncclUniqueId u;
    C_NCCL(ncclGetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return IRMV_OK;
}

extern "C" int irmv_comm_init_rank(const uint8_t id[IRMV_COMM_ID_BYTES], int nranks, int rank, int device, irmv_comm **out)
{
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return cfail(IRMV_ERR_ARG, "irmv_comm_init_rank: bad rank / nranks");
    int have = 0;
    C_HIP(hipGetDeviceCount(&have));
    if (device < 0 || device >= have) return cfail(IRMV_ERR_HIP, "irmv_comm_init_rank: no such HIP device");
    C_HIP(hipSetDevice(device));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    irmv_comm *c = new irmv_comm;
    c->nranks = nranks;
    c->devices.push_back(device);
    c->comms.assign(1, nullptr);
    ncclResult_t r = ncclCommInitRank(&c->comms[0], nranks, u, rank);
    if (r != ncclSuccess) { irmv_comm_destroy(c); return cfail(IRMV_ERR_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
    const int rc = alloc_scratch(c);
    if (rc) { irmv_comm_destroy(c); return rc; }
    *out = c;
    return IRMV_OK;
}

extern "C" int irmv_comm_nranks(const irmv_comm *c) { return c ? c->nranks : 0; }
extern "C" int irmv_comm_local_ranks(const irmv_comm *c) { return c ? (int)c->devices.size() : 0; }

static int sync_all(irmv_comm *c)
{
    for (size_t i = 0; i < c->devices.size(); i++) {
        C_HIP(hipSetDevice(c->devices[i]));
        C_HIP(hipStreamSynchronize(c->streams[i]));
    }
    return IRMV_OK;
}

extern "C" int irmv_comm_broadcast_blob(irmv_comm *c, const void *host_blob, uint64_t bytes, int root, void **dev_ptrs, uint64_t *bytes_out)
{
    if (!c || !dev_ptrs || root < 0 || root >= c->nranks) return cfail(IRMV_ERR_ARG, "irmv_comm_broadcast_blob: bad arguments");
    const size_t n = c->devices.size();
    // (1) the size: 8 bytes from the root, so that ranks of other processes can allocate
    for (size_t i = 0; i < n; i++) {
        C_HIP(hipSetDevice(c->devices[i]));
        C_HIP(hipMemcpyAsync(c->sizes[i], &bytes, sizeof(uint64_t), hipMemcpyHostToDevice, c->streams[i]));
    }
    C_NCCL(ncclGroupStart());
    for (size_t i = 0; i < n; i++) C_NCCL(ncclBroadcast(c->sizes[i], c->sizes[i], sizeof(uint64_t), ncclChar, root, c->comms[i], c->streams[i]));
    C_NCCL(ncclGroupEnd());
    uint64_t total = 0;
    C_HIP(hipSetDevice(c->devices[0]));
    C_HIP(hipMemcpyAsync(&total, c->sizes[0], sizeof(uint64_t), hipMemcpyDeviceToHost, c->streams[0]));
    int rc = sync_all(c);
    if (rc) return rc;
    if (total == 0 || total > (1ull << 31)) return cfail(IRMV_ERR_MODEL, "irmv_comm_broadcast_blob: the root announced an empty or absurd blob");
    // (2) the payload.  The root's host bytes go to the device buffer of the local rank whose RCCL rank is `root`: with
    // init_all local rank i IS rank i; with init_rank the one local rank is the root iff host_blob is given.
    for (size_t i = 0; i < n; i++) {
        C_HIP(hipSetDevice(c->devices[i]));
        if (c->blobs[i]) { C_HIP(hipFree(c->blobs[i])); c->blobs[i] = nullptr; }
        C_HIP(hipMalloc(&c->blobs[i], (size_t)total));
        const bool is_root = n > 1 ? ((int)i == root) : (host_blob != nullptr);
        if (is_root) {
            if (!host_blob || bytes != total) return cfail(IRMV_ERR_ARG, "irmv_comm_broadcast_blob: the root passes the blob and its size");
            C_HIP(hipMemcpyAsync(c->blobs[i], host_blob, (size_t)total, hipMemcpyHostToDevice, c->streams[i]));
        }
    }
    C_NCCL(ncclGroupStart());
    for (size_t i = 0; i < n; i++) C_NCCL(ncclBroadcast(c->blobs[i], c->blobs[i], (size_t)total, ncclChar, root, c->comms[i], c->streams[i]));
    C_NCCL(ncclGroupEnd());
    rc = sync_all(c);
    if (rc) return rc;
    for (size_t i = 0; i < n; i++) dev_ptrs[i] = c->blobs[i];
    if (bytes_out) *bytes_out = total;
    return IRMV_OK;
}

extern "C" int irmv_comm_allreduce_f64(irmv_comm *c, double *values, int op)
{
    if (!c || !values || (op != 0 && op != 1)) return cfail(IRMV_ERR_ARG, "irmv_comm_allreduce_f64: bad arguments");
    const size_t n = c->devices.size();
    for (size_t i = 0; i < n; i++) {
        C_HIP(hipSetDevice(c->devices[i]));
        C_HIP(hipMemcpyAsync(c->scratch[i], &values[i], sizeof(double), hipMemcpyHostToDevice, c->streams[i]));
    }
    C_NCCL(ncclGroupStart());
    for (size_t i = 0; i < n; i++) C_NCCL(ncclAllReduce(c->scratch[i], c->scratch[i], 1, ncclDouble, op == 1 ? ncclMax : ncclSum, c->comms[i], c->streams[i]));
    C_NCCL(ncclGroupEnd());
    for (size_t i = 0; i < n; i++) {
        C_HIP(hipSetDevice(c->devices[i]));
        C_HIP(hipMemcpyAsync(&values[i], c->scratch[i], sizeof(double), hipMemcpyDeviceToHost, c->streams[i]));
    }
    return sync_all(c);
}
