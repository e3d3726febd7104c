// libirmv_hip.so host side: engine object, execution plan, hipGraph capture, C ABI.
//
// MI355X-first counterpart of irmv_detection::YoloEngine (reference
// src/yolo_engine.cpp) and PnPSolver (src/pnp_solver.cpp):
//   * frame slots are pinned host memory (hipHostMalloc) copied to HBM by an
//     async copy on a dedicated upload stream, event-chained to the captured
//     step (results need no download: the NMS kernel stores its records straight into mapped pinned host memory;
//     only the classical-extraction mode keeps device records + one D2H copy) -- the dGPU
//     answer to the reference's cudaMallocManaged source buffer (:60-61) +
//     TripleBuffer: slot n+1 uploads while slot n computes;
//   * one set of weights per device shared by all slots (the reference builds
//     three full engines, src/irm_detector.cpp:35-38);
//   * the kernels of a step {front, fused C2f blocks, convs, pool, decode, NMS+PnP}
//     are captured once per (first_slot, count) into a hipGraph (:102-107),
//     and `count` independent frames ride through every kernel as the batch
//     dimension of its GEMM M axis, which is what fills 256 CUs;
//   * no host work between launch and results except the final struct copy
//     (parse_output's scaling, :202-220, runs in the NMS kernel).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <set>
#include <memory>
#include <string>
#include <vector>

#include "../../include/irmv_hip.h"
#include "irmv_common.hpp"
#include "numa.hpp"

using namespace irmv;

constexpr int kLightPointsCap = 4096;       // contour points per detection

static thread_local std::string g_err;
static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return fail(IRMV_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)

extern "C" const char *irmv_last_error(void) { return g_err.c_str(); }
// "irmv_hip 0.3 (gfx950; HIP runtime <hipRuntimeGetVersion>)": the runtime that actually serves this library.  It is the
// process's FIRST libamdhip64.so.7 -- /opt/rocm's 7.2 on its own, torch's bundled 7.0 (same SONAME) when `import torch`
// came first.
extern "C" const char *irmv_version(void)
{
    static char buf[96];
    static std::once_flag once;
    std::call_once(once, [] {
        int v = 0;
        if (hipRuntimeGetVersion(&v) != hipSuccess) v = 0;
        snprintf(buf, sizeof buf, "irmv_hip 0.3 (gfx950; HIP runtime %d)", v);
    });
    return buf;
}
extern "C" int irmv_device_synchronize(int device)
{
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return IRMV_OK;
}
extern "C" int irmv_device_count(int *count)
{
    if (!count) return fail(IRMV_ERR_ARG, "count is null");
    HIP_TRY(hipGetDeviceCount(count));
    return IRMV_OK;
}

// ---- .irmw blob ------------------------------------------------------------------
#pragma pack(push, 1)
struct BlobHeader { char magic[4]; uint32_t version, nc, nk, reg_max, n_layers, dtype, reserved; };
struct BlobLayer { char name[32]; uint32_t cin, cout, k, stride, act, pad; uint64_t w_off, b_off; };
#pragma pack(pop)

struct LayerW {
    std::string name;
    int cin, cout, k, stride, act;
    int groups;         // > 1: depthwise (groups == cout, cin == 1)
    const uint16_t *w;  // OHWI fp16 bits (points into the blob copy)
    const float *b;
};

static float half_bits_to_float(uint16_t h)
{
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, x;
    if (exp == 0) {
        if (man == 0) x = sign;
        else {
            int e = -1;
            do { e++; man <<= 1; } while (!(man & 0x400u));
            x = sign | ((uint32_t)(112 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) x = sign | 0x7f800000u | (man << 13);
    else x = sign | ((exp + 112u) << 23) | (man << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// float -> IEEE fp16 bits, round to nearest even (dequantised int8 weights are stored as the fp16 the kernels multiply with)
static uint16_t float_to_half_bits(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0u));   // inf / nan
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                          // rounds to inf
    if (x < 0x33000001u) return (uint16_t)sign;                                                        // rounds to zero
    if (x < 0x38800000u) {   // subnormal half
        const int shift = 126 - (int)(x >> 23);                     // 14..24
        const uint32_t man = (x & 0x7fffffu) | 0x800000u;
        uint32_t h = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((x >> 23) - 112u) << 10 | ((x >> 13) & 0x3ffu);
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

// ---- engine ----------------------------------------------------------------------
struct Tensor {
    std::string name;
    void *base = nullptr;
    size_t slot_elems = 0;
    int H = 0, W = 0, C = 0;
    bool f32 = false;
    size_t esize() const { return f32 ? 4 : 2; }
    void *slot(int s) const { return static_cast<char *>(base) + (size_t)s * slot_elems * esize(); }
};

struct SegRef { int t = -1, coff = 0, C = 0, shift = 0; };

enum OpKind { OP_PRE, OP_CONV0, OP_CONV, OP_POOL, OP_NMS, OP_LIGHT, OP_FRONT, OP_C2F2, OP_C2F32, OP_DW, OP_SHUF, OP_SCAN, OP_BNECK, OP_KPT3 };

struct Op {
    OpKind kind;
    std::string layer;
    ConvCfg cfg{};      // tile shape for full batched steps (count == num_slots)
    ConvCfg cfg_one{};  // tile shape for single-frame steps (latency mode)
    char kname_one[48] = {0};
    SegRef s0, s1;
    int Hin = 0, Win = 0, Hout = 0, Wout = 0, cin = 0, cout = 0, cout_pad = 0, ksteps = 0;
    int out_t = -1, out_coff = 0, res_t = -1, res_coff = 0;
    half_t *w_packed = nullptr;
    half_t *w_k16 = nullptr;   // 1x1 layers with Cin = 16 and fp32 output (the keypoint branch's finals): the weights in the A layout of v_mfma_f32_16x16x16_f16 [64 lanes][4]
    half_t *w_lds[4] = {nullptr, nullptr, nullptr, nullptr};   // LDS-kernel layout for nt = 1 / 2 / 4 / 8 (eligible 3x3 layers only; nt = 8: stride-2 layers with >= 128 output channels)
    float *bias = nullptr;
    double flops = 0, bytes = 0;  // per frame (bytes: activations in + out, plus the weights)
    double w_bytes = 0;           // the weights' share of `bytes`: read once per LAUNCH, not once per frame (irmv_engine_profile)
    double out_bytes = 0;         // the output's share (a conv that carries a fused 1x1 writes that layer's output instead of its own)
    bool pair = false;
    int lane = 0;      // 0 = trunk; 1..3 = Detect branch (box / cls / kpt): which grouped launch a head conv may join
    int level = -1;    // Detect level of a head op: it may start as soon as P(level) exists
    int signal = -1;   // >= 0: this op produces P(signal); side lanes wait on its event
    char kname[48] = {0};
    int sub[4] = {-1, -1, -1, -1};   // OP_C2F2 / OP_C2F32: indices of the layer ops whose weights it uses
    int mode = 0;                    // OP_C2F32: 0 whole block, 1 cv1 + first bottleneck, 2 last bottleneck + cv2
    bool shortcut = false;
    int fuse_next = -1;        // LDS 3x3 conv: index of the 1x1 op computed in its epilogue (Detect-head finals), -1 = none
    bool fused_away = false;   // preprocess / model.0 / model.1 when the fused front kernel runs them (kept for read-backs)
    int group = -1;            // single-frame steps: index into irmv_engine::head_groups of the one launch this conv rides in
    int bneck = -1;            // single-frame steps: index of the OP_BNECK launch (k_bneck.hip) that computes this conv; OP_BNECK itself: 1 = kept
    int kpt3 = -1;             // a keypoint-branch conv: index of the OP_KPT3 launch (k_kpt.hip) that computes its level's branch in every step; OP_KPT3 itself: 1
};

struct GraphKey {
    int first, count;
    uint32_t flags;
    bool operator<(const GraphKey &o) const
    {
        return std::tie(first, count, flags) < std::tie(o.first, o.count, o.flags);
    }
};

// Events of one submitted slot group [first, first + count): h2d = its frames are in HBM (async upload only);
// out = its kernels have run and its results are host-visible (so its device frames may be overwritten too).
struct SlotGroup {
    int first = 0, count = 0;
    hipEvent_t h2d = nullptr, out = nullptr;
    hipStream_t compute = nullptr;   // compute stream of the last submit
    bool in_flight = false;          // submitted and not yet known complete
    bool async_up = false;           // the last submit uploaded on the side stream (event h2d is valid)
};

struct irmv_engine {
    irmv_engine_cfg cfg{};
    int nc = 0, nk = 0, A = 0, no = 0;
    int backbone = 0;   // 0: C2f stages (YOLOv8n), 1: ShuffleNetV2 stages (blob header)
    int num_cus = 256;  // compute units of the device (persistent kernels size their grids by it)
    int numa_node = -1;     // host NUMA node closest to the device (hipDeviceAttributeHostNumaId); -1: unknown
    bool numa_placed = false;   // the pinned frame slots were allocated and first touched under that node's CPU set and memory policy
    // Single-frame engines: the independent Detect-branch convs of the three levels as one launch per stage (k_conv.hip
    // conv3x3_lds_multi / conv_mfma_multi).  family 0: LDS 3x3 with tile (mt 1, nt); 1: direct kernel with cfg.
    struct HeadGroup { std::vector<int> members; int family = 0, nt = 1; ConvCfg cfg{}; char name[48] = {0}; };
    std::vector<HeadGroup> head_groups;
    bool post_keys_only = false;   // IRMV_POST_KEYS_ONLY=1 (tests): run_post's NMS ignores scan_decode_kernel's boxes and decodes its own, as a whole step's does
    bool emit_scan = false;   // candidates are emitted by the class-branch conv epilogues (needs split_scan's counters and all three levels fused)
    int emit_level_abase[3] = {0, 0, 0};
    bool split_scan = true;   // scan + box decode as a multi-workgroup kernel in front of nms_pnp (IRMV_SPLIT_SCAN=0: inside it)
    int *cand_counts = nullptr;
    int lvl_hw[3] = {0, 0, 0}, lvl_base[3] = {0, 0, 0};
    size_t frame_bytes = 0;
    hipStream_t stream = nullptr;                 // stream 0: single-slot detect(), read-backs, profile
    hipStream_t extra_streams[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // streams 1..num_streams-1
    int num_streams = 1;
    // frame hand-off: uploads can ride a stream of their own, chained to the compute streams by the
    // events of the submitted slot group
    hipStream_t h2d_stream = nullptr;
    std::map<std::pair<int, int>, SlotGroup> groups;   // (first, count) -> events of that group's last submit
    std::vector<SlotGroup *> slot_owner;               // per slot: the group whose submit touched it last
    bool inline_copies = false;                        // IRMV_INLINE_COPIES=1: round-1 behaviour, copies on the compute stream
    bool graph_upload = true;                          // uploads that ride the compute stream are a node of the step's graph (IRMV_GRAPH_UPLOAD=0: a separate launch in front of it; measured
                                                       // again in round 5 with the upload kernel: 0.3365 - 0.3438 ms against 0.335 - 0.336 as the first node: no hiding of the graph's launch cost)
    uint8_t *src_host = nullptr;  // pinned [S][frame]
    int sync_launch = 0;               // how a synchronous single-frame step (detect()) reaches the GPU: 0 = one hipGraph replay (upload = its first node), 1 = launched
                                       // kernel by kernel behind the upload; chosen by timing at creation (choose_sync_launch), IRMV_SYNC_LAUNCH=graph|eager forces
    hipStream_t enq_stream = nullptr;  // (set around an eager step: the stream enqueue_step launches on instead of `stream`)
    uint8_t *src_host_dev = nullptr;   // the same memory through the device's mapping (the upload kernel reads it: launch_upload_frames)
    int upload_kernel_blocks = 256;    // 0: synchronous single-frame uploads ride the copy engine like every other upload (IRMV_UPLOAD_KERNEL=0)
    uint8_t *src_dev = nullptr;   // [S][frame]
    uint8_t *rot_dev = nullptr;   // [frame]
    AxisTap *tap_x = nullptr, *tap_y = nullptr;
    std::vector<Tensor> tensors;
    std::map<std::string, int> tensor_idx;
    std::vector<Op> ops;
    std::vector<void *> dev_allocs;
    int head_t[3] = {-1, -1, -1};
    float *head_all = nullptr;
    PnpConst *pnp_dev = nullptr;
    long long *dbg_dev = nullptr;
    std::set<std::string> lazy_tensors;   // tensors a step does not write because a fused kernel keeps them on chip
    bool fused_front = false;          // OP_FRONT replaces preprocess + model.0.conv + model.1.conv in a step
    int front_tiles_x = 0, front_tiles_y = 0, front_stage_bytes = 0, front_tile_y = kFrontTileY;
    int front_v[4] = {0, 0, 0, 0};     // valid (non-padding) net-input column / row ranges
    int front_fastx = 0, front_fx_i0 = 0, front_fx_step = 2;   // every x tap is (i0 + 2 k, i0 + 2 k + 1; 1/2): the front kernel's 2 : 1 column path
    bool classical = false;            // four points from the classical light extraction instead of a keypoint head
    signed char *light_labels = nullptr;   // label pool: light_pool bytes per slot
    size_t light_pool = 0;
    short *light_points = nullptr, *light_hulls = nullptr;
    float *light_boxes = nullptr;      // explicit boxes of irmv_engine_extract_armors
    DevDet *light_dets_dev = nullptr, *light_dets_host = nullptr;
    float *boxes = nullptr;
    unsigned long long *keys = nullptr;
    DevDet *dets_dev = nullptr, *dets_host = nullptr, *dets_host_dev = nullptr;       // *_host_dev: device view of the pinned buffer
    DevFrameOut *fout_dev = nullptr, *fout_host = nullptr, *fout_host_dev = nullptr;
    bool zero_copy_results = false;   // the NMS kernel writes its results straight into pinned host memory (no D2H copy)
    half_t *conv0_w = nullptr;
    float *conv0_b = nullptr;
    PostArgs post{};
    std::map<GraphKey, hipGraphExec_t> graphs;
    double last_detect_ms = 0;
    std::vector<uint8_t> blob;
    std::vector<LayerW> layers;
    std::vector<std::vector<uint16_t>> dequant;   // int8 blobs: per layer fp16(q * scale), what LayerW::w points to
    std::vector<std::vector<uint16_t>> merged_w;  // Detect first-stage convs of a level concatenated along cout (single-frame engines)
    std::vector<std::vector<float>> merged_b;
    bool bneck64 = true;       // single-frame steps run the 64-channel Bottlenecks of the C2f blocks (and their cv2) as one launch each (IRMV_BNECK64=0: off)
    bool merge_head0 = false;
    bool kpt3 = true;          // the keypoint branch of a Detect level as one launch (engines that do not merge the first-stage Detect convs; IRMV_KPT3=0: off)

    ~irmv_engine();
};

// Optional allocation log (IRMV_LOG_ALLOC=1): every device / pinned range an engine owns, so that a faulting
// address reported by the driver can be mapped to a buffer.
static bool log_alloc() { static const bool on = getenv("IRMV_LOG_ALLOC") != nullptr; return on; }
static void log_range(const irmv_engine *e, const char *what, const void *p, size_t bytes)
{
    if (log_alloc()) fprintf(stderr, "[irmv alloc] engine %p %-18s [%p, %p) %zu bytes\n", (const void *)e, what, p, (const void *)((const char *)p + bytes), bytes);
}

// Teardown order matters: nothing may be freed while any stream of this engine can still touch it.
//   1. drain EVERY stream the engine ever enqueued work on (compute, upload, download, capture side lanes);
//   2. destroy the graph executables (they hold kernel-argument copies pointing into the buffers);
//   3. free device memory, then pinned host memory;
//   4. destroy events and streams.
irmv_engine::~irmv_engine()
{
    if (cfg.device >= 0) (void)hipSetDevice(cfg.device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (int i = 0; i < 7; i++)
        if (extra_streams[i]) (void)hipStreamSynchronize(extra_streams[i]);
    if (h2d_stream) (void)hipStreamSynchronize(h2d_stream);
    if (dbg_dev) {   // diagnostic: phase cycles of the last nms_pnp launch per slot (100 MHz s_memtime-independent clock64)
        std::vector<long long> h((size_t)cfg.num_slots * 16);
        if (hipMemcpy(h.data(), dbg_dev, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
            for (int s = 0; s < cfg.num_slots && s < std::max(4, atoi(getenv("IRMV_NMS_STAMPS") ? getenv("IRMV_NMS_STAMPS") : "0")); s++)   // IRMV_NMS_STAMPS=<n>: the first n slots (at least four)
            {
                const long long *t = &h[(size_t)s * 16];
                fprintf(stderr, "[nms stamps] slot %d: keys %lld select %lld decode %lld sort %lld gather+classes %lld rows %lld walk %lld kpts %lld pnp %lld store %lld cycles; total %lld; n=%lld kept=%lld\n", s,
                        t[7] - t[0], t[8] - t[7], t[9] - t[8], t[1] - t[9], t[10] - t[1], t[2] - t[10], t[3] - t[2], t[11] - t[3], t[12] - t[11], t[4] - t[12], t[4] - t[0], t[5], t[6]);
            }
    }
    for (auto &g : graphs) (void)hipGraphExecDestroy(g.second);
    graphs.clear();
    if (log_alloc()) fprintf(stderr, "[irmv alloc] engine %p destroy: freeing %zu device ranges\n", (const void *)this, dev_allocs.size());
    for (void *p : dev_allocs) (void)hipFree(p);
    dev_allocs.clear();
    if (src_host) (void)hipHostFree(src_host);
    if (dets_host) (void)hipHostFree(dets_host);
    if (fout_host) (void)hipHostFree(fout_host);
    if (light_dets_host) (void)hipHostFree(light_dets_host);
    for (auto &kv : groups) {
        if (kv.second.h2d) (void)hipEventDestroy(kv.second.h2d);
        if (kv.second.out) (void)hipEventDestroy(kv.second.out);
    }
    for (int i = 0; i < 7; i++)
        if (extra_streams[i]) (void)hipStreamDestroy(extra_streams[i]);
    if (h2d_stream) (void)hipStreamDestroy(h2d_stream);
    if (stream) (void)hipStreamDestroy(stream);
}

// slots handled by one stream of a multi-slot submit
static int stream_share(const irmv_engine *e, int count) { return (count + e->num_streams - 1) / e->num_streams; }

static int dev_alloc(irmv_engine *e, void **p, size_t bytes)
{
    HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
    e->dev_allocs.push_back(*p);
    log_range(e, "device", *p, bytes ? bytes : 16);
    return IRMV_OK;
}

static int new_tensor(irmv_engine *e, const std::string &name, int H, int W, int C, bool f32, int *idx)
{
    Tensor t;
    t.name = name;
    t.H = H; t.W = W; t.C = C; t.f32 = f32;
    t.slot_elems = (size_t)H * W * C;
    int rc = dev_alloc(e, &t.base, t.slot_elems * t.esize() * e->cfg.num_slots);
    if (rc) return rc;
    // activations start at zero so that never-written pad channels are finite
    HIP_TRY(hipMemset(t.base, 0, t.slot_elems * t.esize() * e->cfg.num_slots));
    *idx = (int)e->tensors.size();
    e->tensor_idx[name] = *idx;
    e->tensors.push_back(t);
    return IRMV_OK;
}

static const LayerW *find_layer(const irmv_engine *e, const std::string &name)
{
    for (auto &l : e->layers)
        if (l.name == name) return &l;
    return nullptr;
}

// output channel held by row p of 16-row MFMA tile t (see k_conv.hip epilogue)
static int tile_row_cout(int t, int p, bool pair)
{
    if (!pair) return t * 16 + p;
    return (t >> 1) * 32 + (p >> 2) * 8 + (t & 1) * 4 + (p & 3);
}

static int pack_conv(irmv_engine *e, const LayerW &l, Op &op)
{
    const int taps = l.k * l.k;
    op.cout_pad = (l.cout + 15) / 16 * 16;
    const int ntiles = op.cout_pad / 16;
    const bool pair = !op.cfg.out_f32 && (op.cout_pad % 32 == 0);   // independent of the tile shape chosen later
    op.pair = pair;
    const int cpt = (l.cin + 31) / 32;
    op.ksteps = op.cfg.cin16 ? (taps + 1) / 2 : taps * cpt;
    std::vector<uint16_t> packed((size_t)ntiles * op.ksteps * 512, 0);
    for (int t = 0; t < ntiles; t++)
        for (int ks = 0; ks < op.ksteps; ks++)
            for (int lane = 0; lane < 64; lane++) {
                const int g = lane >> 4, r = lane & 15;
                const int co = tile_row_cout(t, r, pair);
                for (int j = 0; j < 8; j++) {
                    int tap, c;
                    if (op.cfg.cin16) { tap = 2 * ks + (g >> 1); c = 8 * (g & 1) + j; }
                    else { tap = ks / cpt; c = (ks % cpt) * 32 + 8 * g + j; }
                    uint16_t v = 0;
                    if (co < l.cout && tap < taps && c < l.cin) v = l.w[((size_t)co * taps + tap) * l.cin + c];
                    packed[(((size_t)t * op.ksteps + ks) * 64 + lane) * 8 + j] = v;
                }
            }
    // SiLU layers compute on log2 e-scaled activations (irmv_common.hpp, "activation scale"): weights as they are, the bias
    // scaled once here; layers without activation (the Detect finals) undo the scale in their epilogue and keep their bias
    std::vector<float> bias(op.cout_pad, 0.f);
    for (int i = 0; i < l.cout; i++) bias[i] = l.act == 1 ? (float)((double)l.b[i] * (double)kActScale) : l.b[i];
    int rc = dev_alloc(e, (void **)&op.w_packed, packed.size() * 2);
    if (rc) return rc;
    rc = dev_alloc(e, (void **)&op.bias, bias.size() * 4);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(op.w_packed, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(op.bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    if (l.k == 1 && l.cin == 16 && op.cfg.out_f32 && l.act == 0 && op.cout_pad == 16) {   // lane (g, r): output channel r, input channels 4 g .. 4 g + 3
        std::vector<uint16_t> pk(64 * 4, 0);
        for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 4; j++)
                if ((lane & 15) < l.cout) pk[lane * 4 + j] = l.w[(size_t)(lane & 15) * l.cin + 4 * (lane >> 4) + j];
        int rc3 = dev_alloc(e, (void **)&op.w_k16, pk.size() * 2);
        if (rc3) return rc3;
        HIP_TRY(hipMemcpy(op.w_k16, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
    }
    // LDS-kernel layout: [n-block][chunk of 32 ch][tap][tile in block][lane][8]
    if (l.k == 3 && l.cin % 32 == 0 && l.act == 1 && !op.cfg.out_f32) {
        const int chunks = l.cin / 32;
        for (int v = 0; v < 4; v++) {
            const int nt = 1 << v;
            if (ntiles % nt != 0 || (nt > 1 && !pair)) continue;
            if (nt == 8 && l.stride != 2) continue;      // the 128-channel workgroup exists for the stride-2 layers only (k_conv.hip)
            std::vector<uint16_t> pl((size_t)ntiles * chunks * 9 * 512, 0);
            for (int t = 0; t < ntiles; t++)
                for (int ch = 0; ch < chunks; ch++)
                    for (int tap = 0; tap < 9; tap++)
                        for (int lane = 0; lane < 64; lane++) {
                            const int g = lane >> 4, r = lane & 15;
                            const int co = tile_row_cout(t, r, pair);
                            const int nb = t / nt, ti = t % nt;
                            const size_t base = ((((size_t)nb * chunks + ch) * 9 + tap) * nt + ti) * 512 + (size_t)lane * 8;
                            for (int j = 0; j < 8; j++) {
                                const int c = ch * 32 + 8 * g + j;
                                if (co < l.cout) pl[base + j] = l.w[((size_t)co * 9 + tap) * l.cin + c];
                            }
                        }
            int rc2 = dev_alloc(e, (void **)&op.w_lds[v], pl.size() * 2);
            if (rc2) return rc2;
            HIP_TRY(hipMemcpy(op.w_lds[v], pl.data(), pl.size() * 2, hipMemcpyHostToDevice));
        }
    }
    return IRMV_OK;
}

static int add_conv(irmv_engine *e, const std::string &layer, SegRef s0, SegRef s1, int Hin, int Win, int out_t,
                    int out_coff, int res_t = -1, int res_coff = 0)
{
    const LayerW *l = find_layer(e, layer);
    if (!l) return fail(IRMV_ERR_MODEL, "weight blob has no layer " + layer);
    if (l->cin != s0.C + s1.C) return fail(IRMV_ERR_MODEL, "layer " + layer + ": cin does not match the graph");
    Op op;
    op.kind = OP_CONV;
    op.layer = layer;
    op.s0 = s0; op.s1 = s1;
    op.Hin = Hin; op.Win = Win;
    op.Hout = Hin / l->stride; op.Wout = Win / l->stride;
    op.cin = l->cin; op.cout = l->cout;
    op.out_t = out_t; op.out_coff = out_coff; op.res_t = res_t; op.res_coff = res_coff;
    const Tensor &ot = e->tensors[out_t];
    if (ot.H != op.Hout || ot.W != op.Wout) return fail(IRMV_ERR_MODEL, "layer " + layer + ": output shape mismatch");
    const int cout_pad = (l->cout + 15) / 16 * 16;
    if (out_coff + cout_pad > ot.C) return fail(IRMV_ERR_MODEL, "layer " + layer + ": output slice out of range");
    op.cfg.ks = l->k; op.cfg.stride = l->stride; op.cfg.act = l->act; op.cfg.out_f32 = ot.f32;
    op.cfg.cin16 = (l->cin == 16 && l->k == 3);
    op.cfg.lds = false;
    op.cfg.ipw = 1;
    const int nt_all = cout_pad / 16;
    op.cfg.nt = nt_all >= 4 ? 4 : nt_all;
    // enough workgroups to cover 256 CUs a few times, else halve the pixel tile
    const long m_batch = (long)e->cfg.num_slots * op.Hout * op.Wout;
    const long blocks_mt2 = ((m_batch + 127) / 128) * (cout_pad / (16 * op.cfg.nt));
    op.cfg.mt = blocks_mt2 >= 512 ? 2 : 1;
    op.cfg_one = op.cfg;
    op.cfg_one.mt = 1;
    conv_cfg_name(op.cfg, op.kname, sizeof op.kname);
    conv_cfg_name(op.cfg_one, op.kname_one, sizeof op.kname_one);
    op.flops = 2.0 * op.Hout * op.Wout * (double)l->cout * l->cin * l->k * l->k;
    // algorithmic bytes: every input element once (a half-resolution segment = the tensor that exists, not its upsampled
    // image), the output once, the weights once
    op.bytes = 2.0 * ((double)(Hin >> s0.shift) * (Win >> s0.shift) * s0.C + (double)(Hin >> s1.shift) * (Win >> s1.shift) * s1.C) +
               (double)op.Hout * op.Wout * l->cout * (ot.f32 ? 4.0 : 2.0) + 2.0 * l->cout * l->cin * l->k * l->k;
    op.w_bytes = 2.0 * l->cout * l->cin * l->k * l->k;
    op.out_bytes = (double)op.Hout * op.Wout * l->cout * (ot.f32 ? 4.0 : 2.0);
    int rc = pack_conv(e, *l, op);
    if (rc) return rc;
    e->ops.push_back(op);
    return IRMV_OK;
}

#define TRY(x)            \
    do {                  \
        int _rc = (x);    \
        if (_rc) return _rc; \
    } while (0)

// Depthwise 3x3 (ShuffleNetV2 stages): weights repacked tap-major [9][C] so that a lane's 8 channels are one 16-byte load
static int add_dw(irmv_engine *e, const std::string &layer, SegRef in, int Hin, int Win, int out_t, int out_coff)
{
    const LayerW *l = find_layer(e, layer);
    if (!l) return fail(IRMV_ERR_MODEL, "weight blob has no layer " + layer);
    if (l->groups != l->cout || l->cout != in.C) return fail(IRMV_ERR_MODEL, "layer " + layer + ": not a depthwise conv over the graph's channels");
    Op op;
    op.kind = OP_DW;
    op.layer = layer;
    op.s0 = in;
    op.Hin = Hin; op.Win = Win; op.Hout = Hin / l->stride; op.Wout = Win / l->stride;
    op.cin = op.cout = op.cout_pad = l->cout;
    op.cfg.stride = l->stride;
    op.out_t = out_t; op.out_coff = out_coff;
    const Tensor &ot = e->tensors[out_t];
    if (ot.H != op.Hout || ot.W != op.Wout || out_coff + l->cout > ot.C) return fail(IRMV_ERR_MODEL, "layer " + layer + ": output shape mismatch");
    std::vector<uint16_t> w((size_t)9 * l->cout);
    for (int c = 0; c < l->cout; c++)
        for (int t = 0; t < 9; t++) w[(size_t)t * l->cout + c] = l->w[(size_t)c * 9 + t];
    TRY(dev_alloc(e, (void **)&op.w_packed, w.size() * 2));
    TRY(dev_alloc(e, (void **)&op.bias, (size_t)l->cout * 4));
    HIP_TRY(hipMemcpy(op.w_packed, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    {   // no activation, but the output feeds further layers: it stays at the activation scale, so the bias is scaled too
        std::vector<float> bs((size_t)l->cout);
        for (int i = 0; i < l->cout; i++) bs[i] = (float)((double)l->b[i] * (double)kActScale);
        HIP_TRY(hipMemcpy(op.bias, bs.data(), bs.size() * 4, hipMemcpyHostToDevice));
    }
    snprintf(op.kname, sizeof op.kname, "dwconv3x3s%d", l->stride);
    snprintf(op.kname_one, sizeof op.kname_one, "dwconv3x3s%d", l->stride);
    op.flops = 2.0 * op.Hout * op.Wout * (double)l->cout * 9;
    op.bytes = 2.0 * ((double)Hin * Win + (double)op.Hout * op.Wout) * l->cout + 2.0 * 9 * l->cout;
    e->ops.push_back(op);
    return IRMV_OK;
}

// concat + channel shuffle (two groups) of two bc-channel slices: out[2 i] = a[i], out[2 i + 1] = b[i]
static int add_shuffle(irmv_engine *e, const std::string &name, SegRef a, SegRef b, int H, int W, int out_t)
{
    const Tensor &ot = e->tensors[out_t];
    if (a.C != b.C || a.C % 4 != 0 || ot.C != 2 * a.C || ot.H != H || ot.W != W) return fail(IRMV_ERR_MODEL, name + ": shuffle shapes do not match");
    Op op;
    op.kind = OP_SHUF;
    op.layer = name;
    op.s0 = a; op.s1 = b;
    op.Hin = op.Hout = H; op.Win = op.Wout = W;
    op.cin = op.cout = 2 * a.C;
    op.out_t = out_t;
    snprintf(op.kname, sizeof op.kname, "shuffle_cat");
    snprintf(op.kname_one, sizeof op.kname_one, "shuffle_cat");
    op.bytes = 2.0 * 2.0 * (double)H * W * 2 * a.C;
    e->ops.push_back(op);
    return IRMV_OK;
}

// ShuffleNetV2 blocks (irmv_detection_amd/arch.py _shuffle_down / _shuffle_unit; the oracle's shuffle_down / shuffle_unit)
static int add_shuffle_down(irmv_engine *e, const std::string &prefix, int in_t, int c1, int H, int W, int c2, int out_t)
{
    const int bc = c2 / 2, Ho = H / 2, Wo = W / 2;
    int d1, b1, p1, d2, b2;
    TRY(new_tensor(e, prefix + ".b1.dw", Ho, Wo, c1, false, &d1));
    TRY(new_tensor(e, prefix + ".b1", Ho, Wo, bc, false, &b1));
    TRY(new_tensor(e, prefix + ".b2.pw1", H, W, bc, false, &p1));
    TRY(new_tensor(e, prefix + ".b2.dw", Ho, Wo, bc, false, &d2));
    TRY(new_tensor(e, prefix + ".b2", Ho, Wo, bc, false, &b2));
    TRY(add_dw(e, prefix + ".b1.dw", SegRef{in_t, 0, c1, 0}, H, W, d1, 0));
    TRY(add_conv(e, prefix + ".b1.pw", SegRef{d1, 0, c1, 0}, SegRef{}, Ho, Wo, b1, 0));
    TRY(add_conv(e, prefix + ".b2.pw1", SegRef{in_t, 0, c1, 0}, SegRef{}, H, W, p1, 0));
    TRY(add_dw(e, prefix + ".b2.dw", SegRef{p1, 0, bc, 0}, H, W, d2, 0));
    TRY(add_conv(e, prefix + ".b2.pw2", SegRef{d2, 0, bc, 0}, SegRef{}, Ho, Wo, b2, 0));
    return add_shuffle(e, prefix + ".shuffle", SegRef{b1, 0, bc, 0}, SegRef{b2, 0, bc, 0}, Ho, Wo, out_t);
}

static int add_shuffle_unit(irmv_engine *e, const std::string &prefix, int in_t, int c, int H, int W, int out_t)
{
    const int bc = c / 2;
    int p1, d2, b2;
    TRY(new_tensor(e, prefix + ".b2.pw1", H, W, bc, false, &p1));
    TRY(new_tensor(e, prefix + ".b2.dw", H, W, bc, false, &d2));
    TRY(new_tensor(e, prefix + ".b2", H, W, bc, false, &b2));
    TRY(add_conv(e, prefix + ".b2.pw1", SegRef{in_t, bc, bc, 0}, SegRef{}, H, W, p1, 0));
    TRY(add_dw(e, prefix + ".b2.dw", SegRef{p1, 0, bc, 0}, H, W, d2, 0));
    TRY(add_conv(e, prefix + ".b2.pw2", SegRef{d2, 0, bc, 0}, SegRef{}, H, W, b2, 0));
    return add_shuffle(e, prefix + ".shuffle", SegRef{in_t, 0, bc, 0}, SegRef{b2, 0, bc, 0}, H, W, out_t);
}

// A C2f block with a 32-channel hidden width (model.4 / model.15 at a 640 net) runs as fused kernels (k_c2f.hip) when its
// layers have the shapes those kernels are written for: n = 1 -> one launch, n = 2 -> two.  The layer ops stay in the
// list as `fused_away` (read-backs of the block's internal tensors run them; they are also the bit-exactness reference).
static int fuse_c2f32(irmv_engine *e, const std::string &prefix, int n, bool shortcut, int cat, int tmp, int out_t)
{
    if (const char *ff = getenv("IRMV_FUSED_C2F")) if (ff[0] == '0') return IRMV_OK;
    const int last = (int)e->ops.size() - 1, first = last - (2 * n + 1);
    if (n < 1 || n > 2 || first < 0) return IRMV_OK;
    const Op &c1 = e->ops[first], &c2 = e->ops[last];
    bool ok = c1.cfg.ks == 1 && c1.cout == 64 && c1.pair && c1.cin % 32 == 0 && c1.s0.C % 32 == 0 && c1.cfg.act == 1 &&
              (c1.ksteps == 2 || c1.ksteps == 4 || c1.ksteps == 6) && (n == 1 || shortcut) &&
              c2.cfg.ks == 1 && c2.cout == 64 && c2.pair && c2.cin == (2 + n) * 32 && c2.ksteps == 2 + n && c2.cfg.act == 1 && !c2.cfg.out_f32;
    for (int i = first + 1; i < last && ok; i++) {
        const Op &m = e->ops[i];
        ok = m.cfg.ks == 3 && m.cfg.stride == 1 && m.cin == 32 && m.cout == 32 && m.pair && m.ksteps == 9 && m.cfg.act == 1 && !m.cfg.cin16;
    }
    if (!ok) return IRMV_OK;
    const int bH = c1.Hin, bW = c1.Win;          // (copies: the pushes below may move e->ops)
    const double c1_bytes = c1.bytes, c1_w = c1.w_bytes;
    auto make = [&](int mode, int i_cv1, int i_m1, int i_m2, int i_cv2, const char *nm) {
        Op op;
        op.kind = OP_C2F32;
        op.mode = mode;
        op.shortcut = shortcut;
        op.layer = prefix + (mode == 0 ? " (fused)" : (mode == 1 ? " (cv1+m.0)" : " (m.1+cv2)"));
        snprintf(op.kname, sizeof op.kname, "%s", nm);
        op.sub[0] = i_cv1; op.sub[1] = i_m1; op.sub[2] = i_m2; op.sub[3] = i_cv2;
        op.Hin = bH; op.Win = bW;
        op.out_t = out_t;
        op.res_t = cat;                              // the block's concat buffer
        const double px = (double)bH * bW;
        for (int k = 0; k < 4; k++)
            if (op.sub[k] >= 0) { op.flops += e->ops[op.sub[k]].flops; e->ops[op.sub[k]].fused_away = true; op.w_bytes += e->ops[op.sub[k]].w_bytes; }
        // algorithmic bytes: block input once (mode 0 / 1), concat slices written / read, block output, every fused layer's weights
        op.bytes = op.w_bytes;
        if (mode != 2) op.bytes += c1_bytes - c1_w - px * 64 * 2.0;                // cv1's inputs
        if (mode == 1) op.bytes += px * 96 * 2.0;                                   // y0 | y1 | y2 written
        if (mode == 2) op.bytes += px * 96 * 2.0;                                   // read back
        if (mode != 1) op.bytes += px * 64 * 2.0;                                   // block output
        e->ops.push_back(op);
    };
    if (n == 1) make(0, first, first + 1, first + 2, last, "c2f32_ab");
    else { make(1, first, first + 1, first + 2, -1, "c2f32_a"); make(2, -1, first + 3, first + 4, last, "c2f32_b"); }
    e->lazy_tensors.insert(e->tensors[cat].name);
    e->lazy_tensors.insert(e->tensors[tmp].name);
    return IRMV_OK;
}

// Single-frame steps: a 64-channel Bottleneck (model.6 / 12 / 18 at a 640 net) as ONE launch, the block's last one together
// with cv2 (k_bneck.hip).  The OP_BNECK ops stand behind the block's layer ops, which stay what batched steps run (and the
// bit-exactness reference); a step of one frame skips the layers and runs the fused launches instead.
static int fuse_bneck64(irmv_engine *e, const std::string &prefix, int n, bool shortcut, int cat, int out_t)
{
    if (!e->bneck64 || e->backbone != 0) return IRMV_OK;
    const int last = (int)e->ops.size() - 1, first = last - (2 * n + 1);
    if (n < 1 || n > 2 || first < 0) return IRMV_OK;
    for (int i = first; i <= last; i++)
        if (e->ops[i].kind != OP_CONV) return IRMV_OK;
    const Op &c2 = e->ops[last];
    bool ok = c2.cfg.ks == 1 && c2.cout == 128 && c2.cout_pad == 128 && c2.pair && c2.cin == (2 + n) * 64 && c2.ksteps == 2 * (2 + n) && c2.cfg.act == 1 &&
              !c2.cfg.out_f32 && c2.s1.C == 0 && c2.s0.shift == 0 && c2.res_t < 0;
    for (int i = first + 1; i < last && ok; i++) {
        const Op &m = e->ops[i];
        ok = m.cfg.ks == 3 && m.cfg.stride == 1 && m.cin == 64 && m.cout == 64 && m.pair && m.ksteps == 18 && m.cfg.act == 1 && !m.cfg.cin16 && m.w_lds[0] != nullptr &&
             m.s1.C == 0 && m.s0.shift == 0;
    }
    if (!ok) return IRMV_OK;
    const int bH = c2.Hin, bW = c2.Win;
    for (int i = 0; i < n; i++) {
        const int i_m1 = first + 1 + 2 * i, i_m2 = i_m1 + 1;
        const bool with_cv2 = i == n - 1;
        Op op;
        op.kind = OP_BNECK;
        op.mode = with_cv2 ? 1 : 0;
        op.shortcut = shortcut;
        op.layer = prefix + ".m." + std::to_string(i) + (with_cv2 ? " + cv2 (one launch)" : " (one launch)");
        snprintf(op.kname, sizeof op.kname, with_cv2 ? "bneck64_b" : "bneck64_a");
        snprintf(op.kname_one, sizeof op.kname_one, "%s", op.kname);
        op.sub[0] = i_m1; op.sub[1] = i_m2; op.sub[2] = with_cv2 ? last : -1;
        op.Hin = op.Hout = bH; op.Win = op.Wout = bW;
        op.out_t = with_cv2 ? out_t : cat;
        op.res_t = cat;
        const double px = (double)bH * bW;
        for (int k = 0; k < 3; k++)
            if (op.sub[k] >= 0) { op.flops += e->ops[op.sub[k]].flops; op.w_bytes += e->ops[op.sub[k]].w_bytes; }
        op.bytes = op.w_bytes + px * 64 * 2.0 + (with_cv2 ? px * (64.0 * n + 128.0) * 2.0 : px * 64 * 2.0);   // y_in once; + the other concat slices and the block output, or y_next
        op.bneck = 1;
        e->ops.push_back(op);
        const int me = (int)e->ops.size() - 1;
        e->ops[i_m1].bneck = me; e->ops[i_m2].bneck = me;
        if (with_cv2) e->ops[last].bneck = me;
    }
    e->lazy_tensors.insert(e->tensors[cat].name);   // (a single-frame step leaves the last slice of the concat buffer and the bottleneck intermediate unwritten:
    return IRMV_OK;                                 //  read-backs of them run the layer ops, like the fused 32-channel blocks')
}

// The keypoint branch of a Detect level -- the last three ops: 3x3 (Cin -> 16), 3x3 (16 -> 16) carrying the final 1x1 -- as one
// launch (k_kpt.hip).  The OP_KPT3 op stands behind the layer ops; a step runs it and skips them, read-backs of the two
// intermediate tensors run the layers (they remain the bit-exactness reference, IRMV_KPT3=0 the switch).
static int fuse_kpt3(irmv_engine *e, int level)
{
    if (!e->kpt3 || e->ops.size() < 3) return IRMV_OK;
    const int i2 = (int)e->ops.size() - 1, i1 = i2 - 1, i0 = i2 - 2;
    const Op &o0 = e->ops[i0], &o1 = e->ops[i1], &o2 = e->ops[i2];
    const bool ok = o0.kind == OP_CONV && o1.kind == OP_CONV && o2.kind == OP_CONV && o1.fuse_next == i2 && o1.cfg.cin16 && o2.w_k16 &&
                    o0.cfg.ks == 3 && o0.cfg.stride == 1 && o0.cfg.act == 1 && !o0.cfg.out_f32 && !o0.cfg.cin16 && o0.cout_pad == 16 && !o0.pair && o0.res_t < 0 &&
                    o0.s1.C == 0 && o0.s0.shift == 0 && o0.w_lds[0] != nullptr && kpt3_eligible(o0.cin) && o1.s0.t == o0.out_t && o1.cin == 16 && o1.ksteps == 5 &&
                    o0.Hin == o1.Hin && o0.Win == o1.Win;
    if (!ok) return IRMV_OK;
    {   // the kernel addresses the level's input through a buffer descriptor with 32-bit byte offsets
        const Tensor &xt = e->tensors[o0.s0.t];
        if ((double)e->cfg.num_slots * xt.H * xt.W * xt.C * 2.0 >= 2147483648.0) return IRMV_OK;
    }
    Op op;
    op.kind = OP_KPT3;
    op.layer = "model.22.cv4." + std::to_string(level) + " (one launch)";
    snprintf(op.kname, sizeof op.kname, "kpt3_c%d", o0.cin);
    snprintf(op.kname_one, sizeof op.kname_one, "%s", op.kname);
    op.sub[0] = i0; op.sub[1] = i1; op.sub[2] = i2;
    op.Hin = op.Hout = o0.Hin; op.Win = op.Wout = o0.Win;
    op.cin = o0.cin;
    op.lane = o0.lane; op.level = level;
    op.flops = o0.flops + o1.flops + o2.flops;
    op.w_bytes = o0.w_bytes + o1.w_bytes + o2.w_bytes;
    op.out_bytes = o2.out_bytes;
    op.bytes = op.w_bytes + (o0.bytes - o0.w_bytes - o0.out_bytes) + o2.out_bytes;   // the level's input once, the head's keypoint channels once
    op.kpt3 = 1;
    e->lazy_tensors.insert(e->tensors[o0.out_t].name);
    e->lazy_tensors.insert(e->tensors[o1.out_t].name);
    e->ops.push_back(op);
    const int me = (int)e->ops.size() - 1;
    e->ops[i0].kpt3 = e->ops[i1].kpt3 = e->ops[i2].kpt3 = me;
    return IRMV_OK;
}

static int add_c2f(irmv_engine *e, const std::string &prefix, SegRef s0, SegRef s1, int H, int W, int c2, int n,
                   bool shortcut, int out_t)
{
    const int c = c2 / 2;
    int cat, tmp;
    TRY(new_tensor(e, prefix + ".cat", H, W, (2 + n) * c, false, &cat));
    TRY(new_tensor(e, prefix + ".tmp", H, W, c, false, &tmp));
    TRY(add_conv(e, prefix + ".cv1", s0, s1, H, W, cat, 0));
    for (int i = 0; i < n; i++) {
        const std::string m = prefix + ".m." + std::to_string(i);
        TRY(add_conv(e, m + ".cv1", SegRef{cat, (1 + i) * c, c, 0}, SegRef{}, H, W, tmp, 0));
        TRY(add_conv(e, m + ".cv2", SegRef{tmp, 0, c, 0}, SegRef{}, H, W, cat, (2 + i) * c, shortcut ? cat : -1,
                     (1 + i) * c));
    }
    TRY(add_conv(e, prefix + ".cv2", SegRef{cat, 0, (2 + n) * c, 0}, SegRef{}, H, W, out_t, 0));
    TRY(fuse_c2f32(e, prefix, n, shortcut, cat, tmp, out_t));
    if (c == 64) {
        TRY(fuse_bneck64(e, prefix, n, shortcut, cat, out_t));
        if (!e->ops.empty() && e->ops.back().kind == OP_BNECK) e->lazy_tensors.insert(e->tensors[tmp].name);
    }
    return IRMV_OK;
}

// integer tap geometry: same arithmetic as the oracle's axis_tap, written independently
static void axis_taps(std::vector<AxisTap> &out, int dn_total, int sn, int dn, int pad, bool rotate)
{
    out.assign(dn_total, AxisTap{-1, -1, 0, 0});
    for (int d = 0; d < dn_total; d++) {
        const int r = d - pad;
        if (r < 0 || r >= dn) continue;
        const long long num = (long long)(2 * r + 1) * sn - dn, den = 2LL * dn;
        const long long fl = num >= 0 ? num / den : -((-num + den - 1) / den);
        const long long frac = num - fl * den;
        int w = (int)((frac * 2048 + dn) / den);
        int a = (int)fl, b = a + 1;
        if (a < 0) { a = 0; b = 0; w = 0; }
        if (a >= sn - 1) { a = sn - 1; b = sn - 1; w = 0; }
        if (rotate) { a = sn - 1 - a; b = sn - 1 - b; }
        out[d] = AxisTap{a, b, w, 0};
    }
}

// The fused front kernel (k_front.hip) stages each tile's source region in LDS as 4-byte pixels, read in groups of
// 4 pixels = three aligned dwords: the width must be a multiple of 4 and the largest region must fit kFrontStageMax
// bytes of LDS.  Same box arithmetic as the kernel.
static bool front_fits(const std::vector<AxisTap> &tx, const std::vector<AxisTap> &ty, int net, int sw, int *tiles_x, int *tiles_y, int *stage_bytes)
{
    const int W1 = net / 4;
    *tiles_x = (W1 + kFrontTileX - 1) / kFrontTileX;
    *tiles_y = (W1 + kFrontTileY - 1) / kFrontTileY;
    *stage_bytes = front_min_stage_bytes();
    if (sw % 4 != 0) return false;   // 4-pixel groups = 12 source bytes read as three aligned dwords
    auto span = [&](const std::vector<AxisTap> &t, int g0, int n, int *lo, int *hi) {
        *lo = 0x7fffffff; *hi = -1;
        for (int i = g0; i < g0 + n; i++) {
            if (i < 0 || i >= net || t[i].i0 < 0) continue;
            *lo = std::min({*lo, t[i].i0, t[i].i1});
            *hi = std::max({*hi, t[i].i0, t[i].i1});
        }
    };
    int max_pitch = 0, max_rows = 0;
    for (int i = 0; i < *tiles_x; i++) {
        int lo, hi;
        span(tx, 4 * i * kFrontTileX - 3, 4 * kFrontTileX + 3, &lo, &hi);
        if (hi >= 0) max_pitch = std::max(max_pitch, std::min((hi + 4) & ~3, sw) - (lo & ~3));
    }
    for (int i = 0; i < *tiles_y; i++) {
        int lo, hi;
        span(ty, 4 * i * kFrontTileY - 3, 4 * kFrontTileY + 3, &lo, &hi);
        if (hi >= 0) max_rows = std::max(max_rows, hi - lo + 1);
    }
    if (max_pitch > 1023 || max_rows > 1023) return false;   // region-relative taps are packed in 10 bits
    const size_t need = (size_t)max_pitch * max_rows * 4;
    *stage_bytes = std::max((int)std::min<size_t>((need + 255) & ~(size_t)255, 1u << 30), front_min_stage_bytes());
    return need <= (size_t)kFrontStageMax;
}

static int autotune_convs(irmv_engine *e);
static void finalize_head_fusion(irmv_engine *e);
static int build_head_groups(irmv_engine *e);

static int build_engine(irmv_engine *e)
{
    const irmv_engine_cfg &c = e->cfg;
    const int net = c.net_size, S = c.num_slots;
    HIP_TRY(hipSetDevice(c.device));
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    // default: batched engines replay concurrent sub-batches of ~64 frames, two to four of them; a stream per slot
    // for engines of TripleBuffer size, whose single-slot steps then overlap
    e->num_streams = c.num_streams > 0 ? c.num_streams : (c.num_slots <= 4 ? c.num_slots : std::min(4, std::max(2, (c.num_slots + 127) / 128)));   // batched: two graphs of up to 128 frames (round 3: with the
                                                                                                                                                   // weights-resident / multi-block kernels larger graphs win: 256 frames as 2 x 128 +6 % over 192 as 3 x 64)
    { const char *bn = getenv("IRMV_BNECK64"); e->bneck64 = !(bn && bn[0] == '0'); }   // (read before the op list is built)
    { const char *kp = getenv("IRMV_KPT3"); e->kpt3 = !(kp && kp[0] == '0'); }
    if (const char *ns = getenv("IRMV_STREAMS")) e->num_streams = atoi(ns);
    e->num_streams = std::max(1, std::min({e->num_streams, 8, c.num_slots}));
    for (int i = 1; i < e->num_streams; i++) HIP_TRY(hipStreamCreateWithFlags(&e->extra_streams[i - 1], hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&e->h2d_stream, hipStreamNonBlocking));
    { const char *ic = getenv("IRMV_INLINE_COPIES"); e->inline_copies = ic && ic[0] == '1'; }
    { const char *gu = getenv("IRMV_GRAPH_UPLOAD"); e->graph_upload = !(gu && gu[0] == '0'); }
    e->slot_owner.assign(S, nullptr);
    e->frame_bytes = (size_t)c.src_width * c.src_height * 3;
    {
        // NUMA-local frame slots (SURVEY section 7 "hard parts": on a full node the copy engines read 8 x 14 k FPS x 3.93 MB =
        // 440 GB/s of host memory): the creating thread runs on the CPUs of the GPU's own socket and prefers its memory while
        // the slots are allocated and first touched (hipHostMallocNumaUser = "follow the caller's policy"); affinity and
        // policy are restored afterwards.  IRMV_NUMA=0: plain hipHostMallocDefault wherever the thread happens to run.
        const char *nv = getenv("IRMV_NUMA");
        const bool want = e->numa_node >= 0 && !(nv && nv[0] == '0');
        numa::ScopedNode scope(want ? e->numa_node : -1);
        const bool user = want && scope.policy();
        HIP_TRY(hipHostMalloc((void **)&e->src_host, e->frame_bytes * S, user ? (hipHostMallocDefault | hipHostMallocNumaUser) : hipHostMallocDefault));
        log_range(e, "pinned src_host", e->src_host, e->frame_bytes * S);
        memset(e->src_host, 0, e->frame_bytes * S);   // first touch, by the bound thread
        if (hipHostGetDevicePointer((void **)&e->src_host_dev, e->src_host, 0) != hipSuccess) { e->src_host_dev = nullptr; (void)hipGetLastError(); }
        if (const char *uk = getenv("IRMV_UPLOAD_KERNEL")) e->upload_kernel_blocks = atoi(uk);
        e->numa_placed = user && scope.bound();
    }
    TRY(dev_alloc(e, (void **)&e->src_dev, e->frame_bytes * S));
    HIP_TRY(hipMemset(e->src_dev, 0, e->frame_bytes * S));
    TRY(dev_alloc(e, (void **)&e->rot_dev, e->frame_bytes));

    // ---- preprocess geometry (parse_output inverse mapping) ----
    int nw = net, nh = net, px = 0, py = 0;
    if (c.resize_mode == IRMV_RESIZE_LETTERBOX) {
        const double r = std::min((double)net / c.src_width, (double)net / c.src_height);
        nw = std::min(net, (int)std::floor(c.src_width * r + 0.5));
        nh = std::min(net, (int)std::floor(c.src_height * r + 0.5));
        px = (net - nw) / 2;
        py = (net - nh) / 2;
    }
    std::vector<AxisTap> tx, ty;
    axis_taps(tx, net, c.src_width, nw, px, c.rotate180 != 0);
    axis_taps(ty, net, c.src_height, nh, py, c.rotate180 != 0);
    TRY(dev_alloc(e, (void **)&e->tap_x, net * sizeof(AxisTap)));
    TRY(dev_alloc(e, (void **)&e->tap_y, net * sizeof(AxisTap)));
    HIP_TRY(hipMemcpy(e->tap_x, tx.data(), net * sizeof(AxisTap), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->tap_y, ty.data(), net * sizeof(AxisTap), hipMemcpyHostToDevice));
    e->fused_front = front_fits(tx, ty, net, c.src_width, &e->front_tiles_x, &e->front_tiles_y, &e->front_stage_bytes);
    e->front_v[0] = px; e->front_v[1] = px + nw; e->front_v[2] = py; e->front_v[3] = py + nh;
    {   // columns at exactly 2 : 1 (1280 -> 640): the taps of column px + k are the aligned source pair (m, m + 1) with
        // m = m0 + step k even, both weights 1/2; step = 2, or -2 under rotate180 (the pair is then listed as (m + 1, m):
        // with equal weights the blend does not care)
        const int step = c.rotate180 ? -2 : 2;
        const int m0 = nw > 0 ? std::min(tx[px].i0, tx[px].i1) : -1;
        bool fx = nw > 0 && m0 >= 0 && (m0 & 1) == 0;
        for (int d = px; d < px + nw && fx; d++)
            fx = std::min(tx[d].i0, tx[d].i1) == m0 + step * (d - px) && std::max(tx[d].i0, tx[d].i1) == m0 + step * (d - px) + 1 && tx[d].w1 == 1024;
        if (const char *f = getenv("IRMV_FRONT_FASTX")) if (f[0] == '0') fx = false;
        bool direct = fx;   // tiles without a padding pixel skip the LDS staging of the source (k_front.hip); bit 1 of FrontArgs::fastx
        if (const char *f = getenv("IRMV_FRONT_DIRECT")) if (f[0] == '0') direct = false;
        e->front_fastx = fx ? (direct ? 3 : 1) : 0;
        // all tiles direct: nothing is staged, and the tile can be twice as tall (k_front.hip); IRMV_FRONT_TILE8=0 keeps the 4-row tile
        bool tall = fx && direct && e->fused_front;
        if (const char *f = getenv("IRMV_FRONT_TILE8")) if (f[0] == '0') tall = false;
        if (tall) {
            e->front_tile_y = kFrontTileYDirect;
            e->front_tiles_y = (net / 4 + kFrontTileYDirect - 1) / kFrontTileYDirect;
            e->front_stage_bytes = front_min_stage_bytes(kFrontTileYDirect);
        }
        e->front_fx_i0 = fx ? m0 : 0;
        e->front_fx_step = step;
    }
    if (const char *ff = getenv("IRMV_FUSED_FRONT")) if (ff[0] == '0') e->fused_front = false;
    if (e->fused_front && !front_prepare()) e->fused_front = false;

    // ---- graph ----
    const int s2 = net / 2, s4 = net / 4, s8 = net / 8, s16 = net / 16, s32 = net / 32;
    int x0, a0, a1, a2, a3, a4, a5, a6, a7, a8, s9, a9, a12, a15, a16, a18, a19, a21;
    TRY(new_tensor(e, "input", net, net, 4, false, &x0));
    TRY(new_tensor(e, "0", s2, s2, 16, false, &a0));
    TRY(new_tensor(e, "1", s4, s4, 32, false, &a1));
    const bool shuffle = e->backbone == 1;   // ShuffleNetV2 stages: blocks 2..8, P3 / P4 / P5 = tensors "3" / "6" / "8"
    if (shuffle) TRY(new_tensor(e, "2", s8, s8, 64, false, &a2));
    else TRY(new_tensor(e, "2", s4, s4, 32, false, &a2));
    TRY(new_tensor(e, "3", s8, s8, 64, false, &a3));
    if (shuffle) TRY(new_tensor(e, "4", s16, s16, 128, false, &a4));
    else TRY(new_tensor(e, "4", s8, s8, 64, false, &a4));
    TRY(new_tensor(e, "5", s16, s16, 128, false, &a5));
    TRY(new_tensor(e, "6", s16, s16, 128, false, &a6));
    TRY(new_tensor(e, "7", s32, s32, 256, false, &a7));
    TRY(new_tensor(e, "8", s32, s32, 256, false, &a8));
    TRY(new_tensor(e, "9.cat", s32, s32, 512, false, &s9));
    TRY(new_tensor(e, "9", s32, s32, 256, false, &a9));
    TRY(new_tensor(e, "12", s16, s16, 128, false, &a12));
    TRY(new_tensor(e, "15", s8, s8, 64, false, &a15));
    TRY(new_tensor(e, "16", s16, s16, 64, false, &a16));
    TRY(new_tensor(e, "18", s16, s16, 128, false, &a18));
    TRY(new_tensor(e, "19", s32, s32, 128, false, &a19));
    TRY(new_tensor(e, "21", s32, s32, 256, false, &a21));

    { Op op; op.kind = OP_PRE; op.layer = "preprocess"; snprintf(op.kname, sizeof op.kname, "preprocess");
      op.bytes = (double)e->frame_bytes + (double)net * net * 8; e->ops.push_back(op); }
    {
        const LayerW *l = find_layer(e, "model.0.conv");
        if (!l || l->cin != 3 || l->cout != 16 || l->k != 3 || l->stride != 2)
            return fail(IRMV_ERR_MODEL, "model.0.conv missing or not 3x3 s2 3->16");
        // A fragments of the single 16-channel tile: lane (g, r) of k-step s holds channel r,
        // k = 32 s + 8 g + j  ->  kernel row kh = 2 s + (g >> 1), tap slot kw = 2 (g & 1) + (j >> 2), channel j & 3
        std::vector<uint16_t> w(2 * 64 * 8, 0);
        std::vector<float> b(16);
        for (int o = 0; o < 16; o++) b[o] = (float)((double)l->b[o] * (double)kActScale);   // (irmv_common.hpp, "activation scale")
        for (int ks = 0; ks < 2; ks++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int g = lane >> 4, o = lane & 15;
                    const int kh = 2 * ks + (g >> 1), kw = 2 * (g & 1) + (j >> 2), ci = j & 3;
                    if (kh < 3 && kw < 3 && ci < 3) w[((size_t)ks * 64 + lane) * 8 + j] = l->w[(o * 9 + kh * 3 + kw) * 3 + ci];
                }
        TRY(dev_alloc(e, (void **)&e->conv0_w, w.size() * 2));
        TRY(dev_alloc(e, (void **)&e->conv0_b, b.size() * 4));
        HIP_TRY(hipMemcpy(e->conv0_w, w.data(), w.size() * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->conv0_b, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        Op op; op.kind = OP_CONV0; op.layer = "model.0.conv"; snprintf(op.kname, sizeof op.kname, "conv0_mfma");
        op.flops = 2.0 * s2 * s2 * 16 * 27;
        op.bytes = (double)net * net * 8 + (double)s2 * s2 * 32 + 27 * 16 * 2;
        e->ops.push_back(op);
    }
    TRY(add_conv(e, "model.1.conv", SegRef{a0, 0, 16, 0}, SegRef{}, s2, s2, a1, 0));
    {
        const Op &m1 = e->ops.back();
        if (!(m1.cfg.cin16 && m1.ksteps == 5 && m1.pair && m1.cout_pad == 32 && m1.out_coff == 0)) e->fused_front = false;
        if (e->fused_front) {
            Op op; op.kind = OP_FRONT; op.layer = "preprocess+model.0+model.1"; snprintf(op.kname, sizeof op.kname, "front_fused");
            op.flops = e->ops[1].flops + m1.flops;
            op.bytes = (double)e->frame_bytes + (double)s4 * s4 * 32 * 2;
            op.w_packed = m1.w_packed; op.bias = m1.bias; op.out_t = m1.out_t;
            for (Op &o : e->ops) o.fused_away = true;   // preprocess, model.0.conv, model.1.conv
            e->lazy_tensors.insert("input"); e->lazy_tensors.insert("0");
            e->ops.push_back(op);
        }
    }
    int p3 = a4, p4 = a6, p5 = a8;   // the tensors the neck reads
    if (shuffle) {
        TRY(add_shuffle_down(e, "model.2", a1, 32, s4, s4, 64, a2));
        TRY(add_shuffle_unit(e, "model.3", a2, 64, s8, s8, a3));
        TRY(add_shuffle_down(e, "model.4", a3, 64, s8, s8, 128, a4));
        TRY(add_shuffle_unit(e, "model.5", a4, 128, s16, s16, a5));
        TRY(add_shuffle_unit(e, "model.6", a5, 128, s16, s16, a6));
        TRY(add_shuffle_down(e, "model.7", a6, 128, s16, s16, 256, a7));
        TRY(add_shuffle_unit(e, "model.8", a7, 256, s32, s32, a8));
        p3 = a3;
    } else {
    TRY(add_c2f(e, "model.2", SegRef{a1, 0, 32, 0}, SegRef{}, s4, s4, 32, 1, true, a2));
    {
        // model.2 as one kernel (k_c2f.hip) when its four layers have the shapes that kernel is written for
        const int n = (int)e->ops.size();
        const Op &c1 = e->ops[n - 4], &m1 = e->ops[n - 3], &m2 = e->ops[n - 2], &c2 = e->ops[n - 1];
        bool ok = c1.cin == 32 && c1.cout == 32 && c1.cfg.ks == 1 && c1.pair && c1.ksteps == 1 &&
                  m1.cin == 16 && m1.cout == 16 && m1.cfg.cin16 && m1.ksteps == 5 && !m1.pair &&
                  m2.cin == 16 && m2.cout == 16 && m2.cfg.cin16 && m2.ksteps == 5 && !m2.pair && m2.res_t >= 0 &&
                  c2.cin == 48 && c2.cout == 32 && c2.cfg.ks == 1 && c2.pair && c2.ksteps == 2 &&
                  c1.cfg.act == 1 && m1.cfg.act == 1 && m2.cfg.act == 1 && c2.cfg.act == 1;
        if (const char *ff = getenv("IRMV_FUSED_C2F")) if (ff[0] == '0') ok = false;
        if (ok) {
            Op op; op.kind = OP_C2F2; op.layer = "model.2 (cv1+m.0+cv2)"; snprintf(op.kname, sizeof op.kname, "c2f2_fused");
            op.flops = c1.flops + m1.flops + m2.flops + c2.flops;
            op.bytes = 2.0 * (double)s4 * s4 * 32 * 2;
            for (int i = 0; i < 4; i++) { op.sub[i] = n - 4 + i; e->ops[n - 4 + i].fused_away = true; }
            op.s0 = c1.s0; op.out_t = c2.out_t;
            e->lazy_tensors.insert("model.2.cat"); e->lazy_tensors.insert("model.2.tmp");
            e->ops.push_back(op);
        }
    }
    TRY(add_conv(e, "model.3.conv", SegRef{a2, 0, 32, 0}, SegRef{}, s4, s4, a3, 0));
    TRY(add_c2f(e, "model.4", SegRef{a3, 0, 64, 0}, SegRef{}, s8, s8, 64, 2, true, a4));
    TRY(add_conv(e, "model.5.conv", SegRef{a4, 0, 64, 0}, SegRef{}, s8, s8, a5, 0));
    TRY(add_c2f(e, "model.6", SegRef{a5, 0, 128, 0}, SegRef{}, s16, s16, 128, 2, true, a6));
    TRY(add_conv(e, "model.7.conv", SegRef{a6, 0, 128, 0}, SegRef{}, s16, s16, a7, 0));
    TRY(add_c2f(e, "model.8", SegRef{a7, 0, 256, 0}, SegRef{}, s32, s32, 256, 1, true, a8));
    }
    TRY(add_conv(e, "model.9.cv1", SegRef{p5, 0, 256, 0}, SegRef{}, s32, s32, s9, 0));
    { Op op; op.kind = OP_POOL; op.layer = "model.9.m"; snprintf(op.kname, sizeof op.kname, "sppf_pool");
      op.bytes = (double)s32 * s32 * 128 * 2 * 4; e->ops.push_back(op); }
    TRY(add_conv(e, "model.9.cv2", SegRef{s9, 0, 512, 0}, SegRef{}, s32, s32, a9, 0));
    TRY(add_c2f(e, "model.12", SegRef{a9, 0, 256, 1}, SegRef{p4, 0, 128, 0}, s16, s16, 128, 1, false, a12));
    TRY(add_c2f(e, "model.15", SegRef{a12, 0, 128, 1}, SegRef{p3, 0, 64, 0}, s8, s8, 64, 1, false, a15));
    e->ops.back().signal = 0;
    TRY(add_conv(e, "model.16.conv", SegRef{a15, 0, 64, 0}, SegRef{}, s8, s8, a16, 0));
    TRY(add_c2f(e, "model.18", SegRef{a16, 0, 64, 0}, SegRef{a12, 0, 128, 0}, s16, s16, 128, 1, false, a18));
    e->ops.back().signal = 1;
    TRY(add_conv(e, "model.19.conv", SegRef{a18, 0, 128, 0}, SegRef{}, s16, s16, a19, 0));
    TRY(add_c2f(e, "model.21", SegRef{a19, 0, 128, 0}, SegRef{a9, 0, 256, 0}, s32, s32, 256, 1, false, a21));
    e->ops.back().signal = 2;

    // Detect head: per level one fp32 record of kHeadRec per anchor: box 64 | cls 16 | kpt 16
    const int P[3] = {a15, a18, a21}, PC[3] = {64, 128, 256}, PS[3] = {s8, s16, s32};
    int base = 0;
    for (int i = 0; i < 3; i++) {
        e->lvl_hw[i] = PS[i] * PS[i];
        e->lvl_base[i] = base;
        base += e->lvl_hw[i];
    }
    e->A = base;
    TRY(dev_alloc(e, (void **)&e->head_all, (size_t)S * e->A * kHeadRec * 4));
    HIP_TRY(hipMemset(e->head_all, 0, (size_t)S * e->A * kHeadRec * 4));
    for (int i = 0; i < 3; i++) {   // per-level views [slot][H*W][kHeadRec] into the one head allocation
        Tensor t;
        t.name = "head." + std::to_string(i);
        t.H = PS[i]; t.W = PS[i]; t.C = kHeadRec; t.f32 = true;
        t.slot_elems = (size_t)PS[i] * PS[i] * kHeadRec;
        t.base = e->head_all + (size_t)e->lvl_base[i] * S * kHeadRec;
        e->head_t[i] = (int)e->tensors.size();
        e->tensor_idx[t.name] = e->head_t[i];
        e->tensors.push_back(t);
    }
    const char *br[3] = {"cv2", "cv3", "cv4"};
    const int mid[3] = {64, 64, 16}, off[3] = {0, kClsOff, kKptOff};
    const int nbr = e->nk > 0 ? 3 : 2;
    // Engines that never batch (every step is a single frame: the reference node's shape) run the first-stage 3x3 convs of a
    // level's branches -- same input, 64 + 64 (+ 16) output channels -- as ONE conv: the weights are concatenated along
    // cout (keypoint branch padded to 32 channels with zeros), the second-stage convs read channel slices of the merged
    // output.  Same K order per output channel -> same bits; two or three launches fewer per level, and the level's input
    // is staged once.  Batched engines keep the separate convs (their nt = 4 tiles do not divide 160 channels).
    {
        const char *mh = getenv("IRMV_MERGE_HEAD0");
        e->merge_head0 = stream_share(e, S) == 1 && !(mh && mh[0] == '0');
        if (mh && mh[0] == '1') e->merge_head0 = true;
        for (int i = 0; i < 3 && e->merge_head0; i++)          // every branch conv must have the shape the merge assumes
            for (int b = 0; b < nbr; b++) {
                const LayerW *l0 = find_layer(e, std::string("model.22.") + br[b] + "." + std::to_string(i) + ".0");
                if (!l0 || l0->k != 3 || l0->stride != 1 || l0->act != 1 || l0->cin != PC[i] || l0->cout != mid[b]) e->merge_head0 = false;
            }
    }
    const int coff0[3] = {0, 64, 128};
    int t_s0[3] = {-1, -1, -1};
    if (e->merge_head0) {
        const int cm = nbr == 3 ? 160 : 128;
        e->merged_w.reserve(3); e->merged_b.reserve(3);
        e->layers.reserve(e->layers.size() + 3);      // LayerW pointers handed out below stay valid
        for (int i = 0; i < 3; i++) {
            const LayerW *src[3] = {nullptr, nullptr, nullptr};
            for (int b = 0; b < nbr; b++) src[b] = find_layer(e, std::string("model.22.") + br[b] + "." + std::to_string(i) + ".0");
            const size_t per_out = (size_t)9 * PC[i];
            e->merged_w.emplace_back((size_t)cm * per_out, (uint16_t)0);
            e->merged_b.emplace_back((size_t)cm, 0.f);
            for (int b = 0; b < nbr; b++) {
                memcpy(e->merged_w.back().data() + (size_t)coff0[b] * per_out, src[b]->w, (size_t)mid[b] * per_out * 2);
                memcpy(e->merged_b.back().data() + coff0[b], src[b]->b, (size_t)mid[b] * 4);
            }
            LayerW m;
            m.name = "model.22.s0." + std::to_string(i);
            m.cin = PC[i]; m.cout = cm; m.k = 3; m.stride = 1; m.act = 1;
            m.w = e->merged_w.back().data(); m.b = e->merged_b.back().data();
            e->layers.push_back(m);
            TRY(new_tensor(e, "22.s0." + std::to_string(i), PS[i], PS[i], cm, false, &t_s0[i]));
            TRY(add_conv(e, m.name, SegRef{P[i], 0, PC[i], 0}, SegRef{}, PS[i], PS[i], t_s0[i], 0));
            Op &mo = e->ops.back();
            const double real = nbr == 3 ? 144.0 : 128.0;
            mo.flops *= real / cm;                      // algorithmic work: the zero-padded channels do not count
            mo.lane = 1; mo.level = i;
        }
    }
    for (int b = 0; b < nbr; b++)
        for (int i = 0; i < 3; i++) {
            const std::string pre = std::string("model.22.") + br[b] + "." + std::to_string(i);
            const std::string tn = std::string("22.") + br[b] + "." + std::to_string(i);
            int t1 = -1, t2;
            TRY(new_tensor(e, tn + ".1", PS[i], PS[i], mid[b], false, &t2));
            if (e->merge_head0) {
                TRY(add_conv(e, pre + ".1", SegRef{t_s0[i], coff0[b], mid[b], 0}, SegRef{}, PS[i], PS[i], t2, 0));
            } else {
                TRY(new_tensor(e, tn + ".0", PS[i], PS[i], mid[b], false, &t1));
                TRY(add_conv(e, pre + ".0", SegRef{P[i], 0, PC[i], 0}, SegRef{}, PS[i], PS[i], t1, 0));
                TRY(add_conv(e, pre + ".1", SegRef{t1, 0, mid[b], 0}, SegRef{}, PS[i], PS[i], t2, 0));
            }
            TRY(add_conv(e, pre + ".2", SegRef{t2, 0, mid[b], 0}, SegRef{}, PS[i], PS[i], e->head_t[i], off[b]));
            for (size_t k = e->ops.size() - (e->merge_head0 ? 2 : 3); k < e->ops.size(); k++) { e->ops[k].lane = 1 + b; e->ops[k].level = i; }
            {   // the branch's final 1x1 can ride in the epilogue of its second 3x3 (k_conv.hip, N2 > 0)
                const int i1 = (int)e->ops.size() - 2, i2 = i1 + 1;
                const Op &o1 = e->ops[i1], &o2 = e->ops[i2];
                const char *fh = getenv("IRMV_FUSED_HEAD");
                if (!(fh && fh[0] == '0') && o1.cout == 64 && o1.cin % 32 == 0 && o1.pair && o1.res_t < 0 && o1.cfg.stride == 1 &&
                    o2.cin == 64 && o2.ksteps == 2 && o2.cfg.ks == 1 && o2.cfg.out_f32 && o2.cfg.act == 0 && (o2.cout_pad == 16 || o2.cout_pad == 64))
                    e->ops[i1].fuse_next = i2;
                // ... and the keypoint branch's 16 -> nk final in the epilogue of the Cin = 16 direct kernel (one 16x16x16 MFMA per 16 pixels)
                if (!(fh && fh[0] == '0') && o1.cfg.cin16 && o1.cout_pad == 16 && !o1.pair && o1.res_t < 0 && o1.cfg.stride == 1 && o1.cfg.act == 1 && o2.w_k16)
                    e->ops[i1].fuse_next = i2;
            }
            if (b == 2 && !e->merge_head0) TRY(fuse_kpt3(e, i));
        }

    // ---- post-processing buffers ----
    TRY(dev_alloc(e, (void **)&e->boxes, (size_t)S * e->A * 16));
    TRY(dev_alloc(e, (void **)&e->keys, (size_t)S * e->A * e->nc * 8));
    // per-slot candidate counters of the split scan (k_post.hip scan_decode_kernel): zeroed HERE, once, with a synchronous
    // memset -- afterwards each nms_pnp launch reads its frames' counters and resets them itself (no memset node in a
    // captured step, nothing left non-zero between steps)
    { const char *sp = getenv("IRMV_SPLIT_SCAN"); e->split_scan = !(sp && sp[0] == '0'); }
    if (e->split_scan) {
        TRY(dev_alloc(e, (void **)&e->cand_counts, (size_t)S * sizeof(int)));
        HIP_TRY(hipMemset(e->cand_counts, 0, (size_t)S * sizeof(int)));
    }
    TRY(dev_alloc(e, (void **)&e->dets_dev, (size_t)S * c.max_det * sizeof(DevDet)));
    TRY(dev_alloc(e, (void **)&e->fout_dev, (size_t)S * sizeof(DevFrameOut)));
    HIP_TRY(hipMemset(e->dets_dev, 0, (size_t)S * c.max_det * sizeof(DevDet)));
    HIP_TRY(hipMemset(e->fout_dev, 0, (size_t)S * sizeof(DevFrameOut)));
    // Result records live in mapped, coherent pinned memory: in keypoint mode the NMS kernel stores them there directly
    // (~20 KB per frame over PCIe, visible to the host once the stream is synchronised), which removes the D2H copies of
    // a step -- measured 9.6 us per synchronous call on this stack (scripts/probes/stream_probe.cpp), and copies issued
    // from the compute streams also halve the upload stream's H2D rate.  The classical mode (light_extract_kernel
    // reads and rewrites the records on the device) keeps device records + a copy.
    HIP_TRY(hipHostMalloc((void **)&e->dets_host, (size_t)S * c.max_det * sizeof(DevDet), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(hipHostMalloc((void **)&e->fout_host, (size_t)S * sizeof(DevFrameOut), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(hipHostGetDevicePointer((void **)&e->dets_host_dev, e->dets_host, 0));
    HIP_TRY(hipHostGetDevicePointer((void **)&e->fout_host_dev, e->fout_host, 0));
    log_range(e, "pinned dets_host", e->dets_host, (size_t)S * c.max_det * sizeof(DevDet));
    log_range(e, "pinned fout_host", e->fout_host, (size_t)S * sizeof(DevFrameOut));
    memset(e->dets_host, 0, (size_t)S * c.max_det * sizeof(DevDet));
    memset(e->fout_host, 0, (size_t)S * sizeof(DevFrameOut));
    // class-logit scan + box decode of the candidate anchors: kScanBlocks workgroups per frame (k_post.hip)
    if (e->split_scan) {
        Op op; op.kind = OP_SCAN; op.layer = "scan_decode"; snprintf(op.kname, sizeof op.kname, "scan_decode");
        snprintf(op.kname_one, sizeof op.kname_one, "scan_decode");
        op.bytes = (double)e->A * 64.0; e->ops.push_back(op);
    }
    // [decode +] sort + NMS + keypoints + PnP: one kernel, one workgroup per frame (k_post.hip)
    { Op op; op.kind = OP_NMS; op.layer = "decode_nms_kpt_pnp"; snprintf(op.kname, sizeof op.kname, "nms_pnp");
      op.bytes = (double)e->A * 64.0; e->ops.push_back(op); }
    if (c.point_source == IRMV_POINTS_KEYPOINT_HEAD && e->nk < 8) return fail(IRMV_ERR_MODEL, "point_source = keypoint head, but the model has none");
    e->classical = c.point_source == IRMV_POINTS_CLASSICAL || (c.point_source == IRMV_POINTS_AUTO && e->nk < 8);
    { const char *z = getenv("IRMV_ZERO_COPY_RESULTS"); e->zero_copy_results = !e->classical && !(z && z[0] == '0'); }
    if (e->classical) {
        Op op; op.kind = OP_LIGHT; op.layer = "extract_armors"; snprintf(op.kname, sizeof op.kname, "light_extract");
        e->ops.push_back(op);
    }
    // scratch of the classical extraction: per detection a padded label image (ROIs up to ~510 x 510) and contour points
    const size_t SL = e->classical ? (size_t)S : 1;   // keypoint mode keeps one slot's worth for irmv_engine_extract_armors
    e->light_pool = std::max<size_t>((size_t)8 * c.src_width * c.src_height, (size_t)(c.src_width + 2) * (c.src_height + 2) + 16);
    TRY(dev_alloc(e, (void **)&e->light_labels, SL * e->light_pool));
    TRY(dev_alloc(e, (void **)&e->light_points, SL * c.max_det * kLightPointsCap * 2 * sizeof(short)));
    TRY(dev_alloc(e, (void **)&e->light_hulls, SL * c.max_det * kLightPointsCap * 4 * sizeof(short)));
    TRY(dev_alloc(e, (void **)&e->light_boxes, (size_t)c.max_det * 16));
    TRY(dev_alloc(e, (void **)&e->light_dets_dev, (size_t)c.max_det * sizeof(DevDet)));
    HIP_TRY(hipHostMalloc((void **)&e->light_dets_host, (size_t)c.max_det * sizeof(DevDet), hipHostMallocDefault));
    log_range(e, "pinned light_dets", e->light_dets_host, (size_t)c.max_det * sizeof(DevDet));

    PostArgs &p = e->post;
    p.net = net; p.A = e->A; p.nc = e->nc; p.nk = e->nk;
    p.logit_thr = (float)std::log((double)c.score_thr / (1.0 - (double)c.score_thr));
    p.iou_thr = c.iou_thr;
    p.max_det = c.max_det;
    p.pre_nms_cap = c.pre_nms_cap;
    { const char *pk = getenv("IRMV_POST_KEYS_ONLY"); e->post_keys_only = pk && pk[0] == '1'; }
    { const char *cw = getenv("IRMV_NMS_CLASSWALK"); p.classwalk = (cw && cw[0] == '0') ? 0 : 1; }
    { const char *pf = getenv("IRMV_NMS_PREFILTER"); p.prefilter = (pf && pf[0] == '0') ? 0 : 1;
      if (const char *pe = getenv("IRMV_NMS_PRE")) { int hi = 0, lo = 0; if (sscanf(pe, "%d,%d", &hi, &lo) == 2 && hi >= 64 && hi <= 512 && lo >= 32 && lo < hi) p.prefilter = hi | (lo << 16); } }   // experiment: size of the head of the list   // =0: crowded frames sort and mask every candidate (round-3 behaviour; bit-identical)
    if (c.resize_mode == IRMV_RESIZE_STRETCH) {
        p.scale_x = (float)c.src_width / (float)net;   // src/yolo_engine.cpp:155-156
        p.scale_y = (float)c.src_height / (float)net;
        p.off_x = p.off_y = 0.f;
    } else {
        p.scale_x = (float)c.src_width / (float)nw;
        p.scale_y = (float)c.src_height / (float)nh;
        p.off_x = (float)px;
        p.off_y = (float)py;
    }
    p.armor_size = c.armor_size;
    PnpConst pc;
    pc.fx = c.camera_matrix[0]; pc.fy = c.camera_matrix[4];
    pc.cx = c.camera_matrix[2]; pc.cy = c.camera_matrix[5];
    pc.k1 = c.dist_coeffs[0]; pc.k2 = c.dist_coeffs[1]; pc.p1 = c.dist_coeffs[2];
    pc.p2 = c.dist_coeffs[3]; pc.k3 = c.dist_coeffs[4];
    pc.hy[0] = 135.0 / 2.0 / 1000.0; pc.hy[1] = 225.0 / 2.0 / 1000.0;   // src/pnp_solver.cpp:18-21
    pc.hz[0] = pc.hz[1] = 55.0 / 2.0 / 1000.0;
    TRY(dev_alloc(e, (void **)&e->pnp_dev, sizeof(PnpConst)));
    HIP_TRY(hipMemcpy(e->pnp_dev, &pc, sizeof pc, hipMemcpyHostToDevice));
    p.pnp = e->pnp_dev;
    p.dbg = nullptr;
    if (getenv("IRMV_NMS_STAMPS")) {
        TRY(dev_alloc(e, (void **)&e->dbg_dev, (size_t)S * 16 * sizeof(long long)));
        HIP_TRY(hipMemset(e->dbg_dev, 0, (size_t)S * 16 * sizeof(long long)));
        p.dbg = e->dbg_dev;
    }
    HIP_TRY(hipDeviceSynchronize());
    return IRMV_OK;
}

static int load_blob(irmv_engine *e)
{
    const irmv_engine_cfg &c = e->cfg;
    if (c.weights_path) {
        std::string path = c.weights_path;
        const size_t dot = path.find_last_of('.');
        if (dot != std::string::npos && path.substr(dot) != ".irmw") path = path.substr(0, dot) + ".irmw";
        std::ifstream f(path, std::ios::binary);
        if (!f) return fail(IRMV_ERR_MODEL, "cannot open weight blob " + path + " (convert the model to .irmw first)");
        f.seekg(0, std::ios::end);
        const size_t n = (size_t)f.tellg();
        f.seekg(0, std::ios::beg);
        e->blob.resize(n);
        f.read(reinterpret_cast<char *>(e->blob.data()), (std::streamsize)n);
    } else if (c.weights_blob && c.weights_bytes) {
        e->blob.resize(c.weights_bytes);
        if (c.weights_on_device) {
            HIP_TRY(hipSetDevice(c.device));
            HIP_TRY(hipMemcpy(e->blob.data(), c.weights_blob, c.weights_bytes, hipMemcpyDeviceToHost));
        } else {
            memcpy(e->blob.data(), c.weights_blob, c.weights_bytes);
        }
    } else {
        return fail(IRMV_ERR_MODEL, "no weights: set weights_path or weights_blob");
    }
    if (e->blob.size() < sizeof(BlobHeader)) return fail(IRMV_ERR_MODEL, "weight blob truncated");
    BlobHeader h;
    memcpy(&h, e->blob.data(), sizeof h);
    // dtype 1: fp16 weights.  dtype 2 (BASELINE configs[4], "int8 weights"): int8 OHWI weights + fp32 per-output-channel
    // scales; expanded here, once, to w = fp16(q * scale) -- the fragment packing below is dtype-agnostic from there on.
    if (memcmp(h.magic, "IRMW", 4) != 0 || h.version != 1 || (h.dtype != 1 && h.dtype != 2) || h.reg_max != 16)
        return fail(IRMV_ERR_MODEL, "not an IRMW v1 blob (fp16 or int8 weights)");
    e->dequant.reserve(h.n_layers);
    if (h.nc < 1 || h.nc > 16 || (h.nk != 0 && h.nk != 8))
        return fail(IRMV_ERR_MODEL, "unsupported head: nc must be 1..16, nk 0 or 8");
    if (h.reserved > 1) return fail(IRMV_ERR_MODEL, "unknown backbone id in the weight blob");
    e->backbone = (int)h.reserved;
    e->nc = (int)h.nc;
    e->nk = (int)h.nk;
    e->no = 64 + e->nc + e->nk;
    if (sizeof h + (size_t)h.n_layers * sizeof(BlobLayer) > e->blob.size()) return fail(IRMV_ERR_MODEL, "layer table truncated");
    for (uint32_t i = 0; i < h.n_layers; i++) {
        BlobLayer bl;
        memcpy(&bl, e->blob.data() + sizeof h + (size_t)i * sizeof bl, sizeof bl);
        LayerW l;
        char nm[33];
        memcpy(nm, bl.name, 32);
        nm[32] = 0;
        l.name = nm;
        l.cin = bl.cin; l.cout = bl.cout; l.k = bl.k; l.stride = bl.stride; l.act = bl.act;
        l.groups = bl.pad > 1 ? (int)bl.pad : 1;
        if (l.groups > 1 && !(l.groups == l.cout && l.cin == 1 && l.k == 3 && l.cout % 8 == 0 && l.act == 0))
            return fail(IRMV_ERR_MODEL, "layer " + l.name + ": only depthwise 3x3 grouped convs (no activation) are supported");
        const size_t nw = (size_t)l.cout * l.k * l.k * l.cin;
        const size_t w_bytes = h.dtype == 2 ? ((nw + 3) & ~(size_t)3) + (size_t)l.cout * 4 : nw * 2;
        if (bl.w_off + w_bytes > e->blob.size() || bl.b_off + (size_t)l.cout * 4 > e->blob.size())
            return fail(IRMV_ERR_MODEL, "layer " + l.name + " data out of range");
        l.w = reinterpret_cast<const uint16_t *>(e->blob.data() + bl.w_off);
        if (h.dtype == 2) {
            const int8_t *q = reinterpret_cast<const int8_t *>(e->blob.data() + bl.w_off);
            const float *scale = reinterpret_cast<const float *>(e->blob.data() + bl.w_off + ((nw + 3) & ~(size_t)3));
            e->dequant.emplace_back(nw);
            std::vector<uint16_t> &d = e->dequant.back();
            const size_t per_out = nw / l.cout;
            for (int o = 0; o < l.cout; o++)
                for (size_t i = 0; i < per_out; i++) d[o * per_out + i] = float_to_half_bits((float)q[o * per_out + i] * scale[o]);
            l.w = d.data();   // (the vectors were reserved above: no reallocation moves them)
        }
        l.b = reinterpret_cast<const float *>(e->blob.data() + bl.b_off);
        e->layers.push_back(l);
    }
    return IRMV_OK;
}

extern "C" void irmv_engine_cfg_default(irmv_engine_cfg *cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = sizeof *cfg;
    cfg->device = 0;
    cfg->src_width = 1280;
    cfg->src_height = 1024;
    cfg->net_size = 640;
    cfg->resize_mode = IRMV_RESIZE_STRETCH;
    cfg->rotate180 = 1;
    cfg->swap_rb = 0;
    cfg->score_thr = 0.25f;
    cfg->iou_thr = 0.45f;
    cfg->max_det = 100;
    cfg->pre_nms_cap = 4096;
    cfg->num_slots = 3;
    cfg->armor_size = IRMV_ARMOR_SMALL;
    // config/camera_info.yaml:7,12
    const double K[9] = {957.669211, 0, 345.943891, 0, 969.127115, 284.057302, 0, 0, 1};
    const double D[5] = {-0.405274, 0.126058, -0.026939, -0.006503, 0.0};
    memcpy(cfg->camera_matrix, K, sizeof K);
    memcpy(cfg->dist_coeffs, D, sizeof D);
    // src/irm_detector.cpp:152-173
    cfg->point_source = IRMV_POINTS_AUTO;
    cfg->binary_threshold = 150;
    cfg->light_min_ratio = 0.1f; cfg->light_max_ratio = 0.4f; cfg->light_max_angle = 40.0f;
    cfg->armor_min_small_center_distance = 0.8; cfg->armor_max_small_center_distance = 3.2;
    cfg->armor_min_large_center_distance = 3.2; cfg->armor_max_large_center_distance = 5.5;
}

// Which launch form serves detect() on this box: both timed on slot 0 (whatever its pinned slot holds: zeros at creation), the
// eager one kept only if it is at least 1 % faster.  IRMV_SYNC_LAUNCH=graph|eager skips the timing.
extern "C" int irmv_engine_submit(irmv_engine *e, int first, int count, uint32_t flags);
extern "C" int irmv_engine_wait_slots(irmv_engine *e, int first, int count);
static int choose_sync_launch(irmv_engine *e)
{
    if (const char *v = getenv("IRMV_SYNC_LAUNCH")) {
        if (v[0] == 'e' || v[0] == 'g') { e->sync_launch = v[0] == 'e' ? 1 : 0; return IRMV_OK; }   // (anything else, e.g. "auto": time it)
    }
    double t[2] = {0.0, 0.0};
    for (int round = 0; round < 2; round++)
        for (int mode = 0; mode < 2; mode++) {
            e->sync_launch = mode;
            for (int i = 0; i < 24; i++) {
                const auto t0 = std::chrono::high_resolution_clock::now();
                TRY(irmv_engine_submit(e, 0, 1, IRMV_SUBMIT_H2D));
                TRY(irmv_engine_wait_slots(e, 0, 1));
                if (i >= 8) t[mode] += std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
            }
        }
    e->sync_launch = t[1] < 0.99 * t[0] ? 1 : 0;
    if (getenv("IRMV_AUTOTUNE_VERBOSE")) fprintf(stderr, "[irmv] synchronous single-frame step: graph replay %.1f us, eager launches %.1f us -> %s\n", t[0] / 32, t[1] / 32, e->sync_launch ? "eager" : "graph");
    return IRMV_OK;
}

extern "C" int irmv_engine_create(const irmv_engine_cfg *cfg, irmv_engine **out)
{
    if (!cfg || !out) return fail(IRMV_ERR_ARG, "cfg/out is null");
    if (cfg->struct_size != sizeof(irmv_engine_cfg)) return fail(IRMV_ERR_ARG, "irmv_engine_cfg size mismatch");
    if (cfg->net_size < 64 || cfg->net_size % 32 != 0 || cfg->net_size > 2048) return fail(IRMV_ERR_ARG, "net_size must be a multiple of 32 in [64, 2048]");
    if (cfg->src_width < 2 || cfg->src_height < 2 || cfg->src_width > 4096) return fail(IRMV_ERR_ARG, "src size out of range (width <= 4096)");
    if (cfg->num_slots < 1 || cfg->num_slots > 256) return fail(IRMV_ERR_ARG, "num_slots must be 1..256");
    if (cfg->max_det < 1 || cfg->max_det > IRMV_MAX_DET_CAP) return fail(IRMV_ERR_ARG, "max_det must be 1..256");
    if (cfg->pre_nms_cap < 1 || cfg->pre_nms_cap > IRMV_CAND_CAP) return fail(IRMV_ERR_ARG, "pre_nms_cap must be 1..8192");
    if (!(cfg->score_thr > 0.f && cfg->score_thr < 1.f)) return fail(IRMV_ERR_ARG, "score_thr must be in (0, 1)");
    if (cfg->armor_size != IRMV_ARMOR_SMALL && cfg->armor_size != IRMV_ARMOR_LARGE) return fail(IRMV_ERR_ARG, "bad armor_size");
    if (cfg->point_source < 0 || cfg->point_source > 2) return fail(IRMV_ERR_ARG, "bad point_source");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(IRMV_ERR_HIP, "no such HIP device");
    std::unique_ptr<irmv_engine> e(new irmv_engine);
    e->cfg = *cfg;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) e->num_cus = cus;
        int node = -1;
        if (hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, cfg->device) == hipSuccess) e->numa_node = node;
        else (void)hipGetLastError();   // an older runtime serving the library (torch's bundled ROCm 7.0) does not know the attribute: no node, and no sticky error for the checks behind the tuning launches
    }
    int rc = load_blob(e.get());
    e->cfg.weights_path = nullptr;  // caller-owned, not retained
    e->cfg.weights_blob = nullptr;
    if (rc) return rc;
    rc = build_engine(e.get());
    if (rc) return rc;
    rc = autotune_convs(e.get());
    if (rc) return rc;
    finalize_head_fusion(e.get());
    rc = build_head_groups(e.get());
    if (rc) return rc;
    rc = choose_sync_launch(e.get());
    if (rc) return rc;
    *out = e.release();
    return IRMV_OK;
}

extern "C" int irmv_engine_sync_launch(const irmv_engine *e) { return e ? e->sync_launch : 0; }

extern "C" void irmv_engine_destroy(irmv_engine *e) { delete e; }
extern "C" int irmv_engine_num_slots(const irmv_engine *e) { return e ? e->cfg.num_slots : 0; }
extern "C" int irmv_engine_max_det(const irmv_engine *e) { return e ? e->cfg.max_det : 0; }
extern "C" int irmv_engine_num_streams(const irmv_engine *e) { return e ? e->num_streams : 0; }
extern "C" int irmv_engine_num_anchors(const irmv_engine *e) { return e ? e->A : 0; }
extern "C" int irmv_engine_numa_node(const irmv_engine *e) { return e ? e->numa_node : -1; }
extern "C" int irmv_engine_numa_placed(const irmv_engine *e) { return e && e->numa_placed ? 1 : 0; }

// ---- NUMA helpers for the threads / ranks that feed an engine (tools/irmv_multi_gpu.cpp, bench.py ranks) ----
extern "C" int irmv_numa_device_node(int device, int *node)
{
    if (!node) return fail(IRMV_ERR_ARG, "node is null");
    int v = -1;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeHostNumaId, device) != hipSuccess) {
        (void)hipGetLastError();        // (attribute unknown to the runtime that serves the library: not an error of the caller's)
        v = -1;
    }
    *node = v;
    return IRMV_OK;
}
extern "C" int irmv_numa_bind_thread(int node) { return numa::bind_thread_to_node(node) ? IRMV_OK : fail(IRMV_ERR_ARG, "no usable CPU on that NUMA node (or none listed in sysfs)"); }
extern "C" int irmv_numa_page_node(const void *p) { return numa::page_node(p); }
extern "C" int irmv_numa_parse_cpulist(const char *s, int *cpus, int cap)
{
    const std::vector<int> v = numa::parse_cpulist(s);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) if (cpus) cpus[i] = v[i];
    return (int)v.size();
}
extern "C" int irmv_engine_head_channels(const irmv_engine *e) { return e ? e->no : 0; }

extern "C" uint8_t *irmv_engine_src_buffer(irmv_engine *e, int slot)
{
    if (!e || slot < 0 || slot >= e->cfg.num_slots) return nullptr;
    return e->src_host + (size_t)slot * e->frame_bytes;
}
extern "C" void *irmv_engine_src_device_buffer(irmv_engine *e, int slot)
{
    if (!e || slot < 0 || slot >= e->cfg.num_slots) return nullptr;
    return e->src_dev + (size_t)slot * e->frame_bytes;
}

// tile choices already measured in this process, keyed by layer shape and batch (engines are created
// repeatedly in tests and by multi-slot nodes; the kernels and the device do not change in between)
// Process-wide and shared by every engine on every thread: all accesses hold g_tune_mu.
static std::mutex g_tune_mu;
static std::map<std::string, ConvCfg> g_tune_cache;
static bool g_tune_file_loaded = false;

// IRMV_TUNE_CACHE=<file>: persist the measured choices so that a profiled run (rocprofv3 --pmc ...)
// replays exactly the tiles of the benchmarked run without the tuning launches in its trace.
static void tune_cache_load()
{
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (g_tune_file_loaded) return;
    g_tune_file_loaded = true;
    const char *path = getenv("IRMV_TUNE_CACHE");
    if (!path) return;
    std::ifstream f(path);
    std::string key;
    int mt, nt, lds, ipw;
    while (f >> key >> mt >> nt >> lds >> ipw) {
        ConvCfg c{};
        c.mt = mt; c.nt = nt; c.lds = (lds & 1) != 0; c.deep = (lds & 2) != 0; c.ct = (lds & 4) != 0; c.pw = (lds & 8) != 0; c.pf2 = (lds & 16) != 0; c.pf4 = (lds & 32) != 0; c.cm = (lds & 64) ? 2 : ((lds & 128) ? 4 : 0); c.w8 = (lds & 256) != 0; c.wr = (lds & 512) != 0; c.pp = (lds & 1024) != 0; c.ipw = ipw;   // bit 0 LDS family, 1 deep prefetch, 2 direct kernel in the LDS family's K order, 3 pointwise kernel, 4 two-step staging, 5 four-step staging, 6 / 7 LDS family chunk-major over 2 / 4 images, 8 LDS family 8-wave workgroup, 9 LDS family with resident weights (ipw = images per workgroup, any value), 10 its ping-pong form
        g_tune_cache[key] = c;
    }
}

static void tune_cache_save()
{
    const char *path = getenv("IRMV_TUNE_CACHE");
    if (!path) return;
    std::lock_guard<std::mutex> lk(g_tune_mu);
    std::ofstream f(path);
    for (auto &kv : g_tune_cache)
        f << kv.first << ' ' << kv.second.mt << ' ' << kv.second.nt << ' ' << ((kv.second.lds ? 1 : 0) | (kv.second.deep ? 2 : 0) | (kv.second.ct ? 4 : 0) | (kv.second.pw ? 8 : 0) | (kv.second.pf2 ? 16 : 0) | (kv.second.pf4 ? 32 : 0) | (kv.second.cm == 2 ? 64 : 0) | (kv.second.cm == 4 ? 128 : 0) | (kv.second.w8 ? 256 : 0) | (kv.second.wr ? 512 : 0) | (kv.second.pp ? 1024 : 0)) << ' ' << kv.second.ipw << '\n';
}

static int lds_index(int nt) { return nt == 8 ? 3 : (nt == 4 ? 2 : (nt == 2 ? 1 : 0)); }

static bool run_conv(const Op &op, const ConvCfg &c, const ConvArgs &a, int count, hipStream_t s)
{
    const int li = lds_index(c.nt);
    if (op.w_k16) return launch_conv_k16(a, s);   // the keypoint finals: one kernel, whatever the tile table says (same bits as their fused form)
    if (c.pw) return launch_conv_pw(c, a, s);
    if (c.wr) return c.lds && op.w_lds[2] && launch_conv_wres(c.ipw, a, op.w_lds[2], count, s, c.pp);
    if (c.lds) return op.w_lds[li] && launch_conv_lds(c.stride, c.mt, c.nt, c.ipw, a, op.w_lds[li], count, s, c.pf4 ? 2 : (c.pf2 ? 1 : 0), c.cm, c.w8);
    if (c.ct) {   // direct kernel in the LDS family's K order, on that family's nt = 1 weight packing
        if (!op.w_lds[0] || a.n2 > 0) return false;
        ConvArgs a2 = a;
        a2.w = op.w_lds[0];
        return launch_conv(c, a2, s);
    }
    return launch_conv(c, a, s);
}

static void cfg_name(const ConvCfg &c, char *buf, int n)
{
    if (c.pw) snprintf(buf, n, c.ipw > 1 ? "conv1x1s1_pw_n%d" : "conv1x1s1_pw", c.ipw);
    else if (c.wr) snprintf(buf, n, "conv3x3s1_wres%s_i%d", c.pp ? "_pp" : "", c.ipw);
    else if (c.lds && c.ipw > 1) snprintf(buf, n, "conv3x3s%d_lds_mt%d_nt%d_i%d%s%s", c.stride, c.mt, c.nt, c.ipw, c.cm ? "_cm" : (c.pf4 ? "_p4" : (c.pf2 ? "_p2" : "")), c.w8 ? "_w8" : "");
    else if (c.lds) snprintf(buf, n, "conv3x3s%d_lds_mt%d_nt%d%s%s", c.stride, c.mt, c.nt, c.pf4 ? "_p4" : (c.pf2 ? "_p2" : ""), c.w8 ? "_w8" : "");
    else conv_cfg_name(c, buf, n);
}

// ---- per-layer tile autotuner ---------------------------------------------------
// Every conv layer is timed at its real shape with each (MT, NT) tile the kernel
// family offers and keeps the fastest, once for full batched steps and once for
// single-frame steps.  All tiles walk K in the same order, so the choice never
// changes a single output bit (tests/test_gpu_engine.py::test_tile_choice_is_bitwise_neutral).
static void fill_conv_args(const irmv_engine *e, const Op &op, int first, int count, ConvArgs &a, bool fused = false);

// A candidate of the autotuner: the layer's own configuration with a tile shape and family, every launch-time option off
// (the candidates below switch on the one or two they are about).
static ConvCfg tile_cfg(const ConvCfg &base, int mt, int nt, bool lds, int ipw)
{
    ConvCfg c = base;
    c.mt = mt; c.nt = nt; c.lds = lds; c.ipw = ipw;
    c.deep = false; c.ct = false; c.pw = false; c.pf2 = false; c.pf4 = false; c.cm = 0; c.w8 = false; c.wr = false; c.pp = false;
    return c;
}

static int autotune_convs(irmv_engine *e)
{
    const char *env = getenv("IRMV_AUTOTUNE");
    const bool no_tuning = env && env[0] == '0';   // no timing launches: every layer takes the first legal tile of ITS family (same K order as a tuned engine: bit-identical)
    tune_cache_load();
    hipEvent_t ea, eb;
    HIP_TRY(hipEventCreate(&ea));
    HIP_TRY(hipEventCreate(&eb));
    const int counts[2] = {stream_share(e, e->cfg.num_slots), 1};   // the batch one graph actually runs; single frame
    const bool verbose = getenv("IRMV_AUTOTUNE_VERBOSE") != nullptr;
    const bool only_direct = false;   // (IRMV_CONV_FAMILY=direct, which changed the K order of the 3x3 layers, was removed in round 4)
    for (Op &op : e->ops) {
        if (op.kind != OP_CONV) continue;
        for (int pass = 0; pass < (counts[0] > 1 ? 2 : 1); pass++) {
            ConvArgs a;
            const bool fuse_k16 = op.fuse_next >= 0 && op.cfg.cin16;   // the keypoint final in the direct kernel's epilogue: any pixel tile (nt = 1 is the layer's only one)
            bool want_fuse = op.fuse_next >= 0 && !fuse_k16;
            fill_conv_args(e, op, 0, counts[pass], a, want_fuse || fuse_k16);
            if (want_fuse) {   // the fused epilogue needs an LDS-family tile that owns all 64 channels (nt = 4)
                bool f_ok = false;
                for (int mt = 1; mt <= 4 && !f_ok; mt *= 2) f_ok = !only_direct && op.w_lds[2] && conv_lds_bytes(a, op.cfg.stride, mt, 4, nullptr) > 0;
                if (!f_ok) {
                    op.fuse_next = -1;
                    want_fuse = false;
                    fill_conv_args(e, op, 0, counts[pass], a, false);
                }
            }
            float best = 1e30f;
            ConvCfg best_cfg = pass == 0 ? op.cfg : op.cfg_one;
            // Kernel family by a rule that does not depend on the batch (the two families walk K in
            // different orders): LDS-staged whenever some tile of it fits this layer, else direct.
            bool lds_ok = false;
            if (!only_direct && op.cfg.ks == 3 && op.cfg.act == 1 && !op.cfg.out_f32)
                for (int mt = 1; mt <= 4 && !lds_ok; mt *= 2)
                    for (int nt = 1; nt <= 4 && !lds_ok; nt *= 2) {
                        int pr = 0;
                        const int li = nt == 4 ? 2 : (nt == 2 ? 1 : 0);
                        lds_ok = op.w_lds[li] && conv_lds_bytes(a, op.cfg.stride, mt, nt, &pr) > 0;
                    }
            // IRMV_FORCE_S2=lds|ct|deep (parity test): the stride-2 layers of the LDS family on ONE of their three bit-identical
            // implementations -- the LDS kernel, the direct kernel walking K chunk-major, its deep-prefetch form
            const char *force_s2 = getenv("IRMV_FORCE_S2");
            char key[160];
            snprintf(key, sizeof key, "gfx950.t5|%d.%d.%d.%d.%d.%d|%dx%d>%dx%d|c%d.%d.%d.%d>%d|ld%d.%d.%d|n%d|%d", op.cfg.ks, op.cfg.stride,   // ".t5": the table format / flag set of round 5 -- entries of another round's file never match, so they are re-tuned, not reinterpreted
                     (int)op.cfg.cin16, op.cfg.act, (int)op.cfg.out_f32, (int)lds_ok, a.Hin, a.Win, a.Hout, a.Wout, a.s0.C, a.s1.C, a.s0.shift,
                     a.s1.shift, a.cout_pad, a.s0.ld, a.s1.ld, a.out_ld, counts[pass], (a.res ? 1 : 0) + 2 * a.n2);
            // A cached choice (this process, or the IRMV_TUNE_CACHE file) is replayed only if it is a legal tile of THIS
            // layer at THIS batch: tile shape offered by the family, channel split divides cout, images per workgroup within
            // the batch, LDS geometry fits, and the fused epilogue's constraints.  Anything else is re-tuned.
            bool have_hit = false;
            {
                std::lock_guard<std::mutex> lk(g_tune_mu);
                auto hit = g_tune_cache.find(key);
                if (hit != g_tune_cache.end() && !verbose && !no_tuning && !(force_s2 && lds_ok && op.cfg.stride == 2)) {
                    const ConvCfg &h = hit->second;
                    const bool pow2 = (h.mt == 1 || h.mt == 2 || h.mt == 4) && (h.nt == 1 || h.nt == 2 || h.nt == 4 || (h.nt == 8 && h.w8 && h.mt == 1)) && (h.ipw == 1 || h.ipw == 2 || h.ipw == 4 || (h.wr && h.ipw >= 1));
                    // family: LDS-staged, or (single-frame steps only) its chunk-major stand-in on the direct kernel; never both flags
                    const bool fam_ok = lds_ok ? ((h.lds && !h.ct && !h.deep) || (!h.lds && h.ct && !want_fuse && (!h.deep || counts[pass] == 1)))
                                               : (!h.lds && !h.ct && (!h.deep || counts[pass] == 1 || op.cfg.ks == 1));
                    bool ok = pow2 && op.cout_pad % (16 * h.nt) == 0 && h.ipw <= counts[pass] && fam_ok && (!want_fuse || h.nt == 4);
                    if (h.pw) ok = !h.lds && !h.ct && !h.deep && (h.ipw == 1 || ((h.ipw == 2 || h.ipw == 4) && conv_pw_lds_bytes(a, h.ipw) > 0 && !getenv("IRMV_NO_PWN"))) &&
                                   conv_pw_eligible(op.cfg, a) && !getenv("IRMV_NO_PW");   // one tile shape; ipw = output-channel blocks per workgroup
                    if (h.pw && ok && getenv("IRMV_FORCE_PWN") && h.ipw == 1 && (conv_pw_lds_bytes(a, 2) > 0 || conv_pw_lds_bytes(a, 4) > 0)) ok = false;   // (parity tests)
                    if (!h.pw && getenv("IRMV_FORCE_PW") && conv_pw_eligible(op.cfg, a)) ok = false;                                  // (parity tests)
                    if (ok && h.lds && !h.wr) {
                        const int li = lds_index(h.nt);
                        ok = op.w_lds[li] && conv_lds_bytes(a, op.cfg.stride, h.mt, h.nt, nullptr, h.w8) > 0;
                    }
                    if (ok && h.wr) ok = !getenv("IRMV_NO_WRES") && h.lds && !h.pf2 && !h.cm && !h.w8 && h.mt == 2 && h.nt == 4 && op.w_lds[2] && conv_wres_bytes(a, op.cfg.stride, h.pp) > 0;
                    if (ok && getenv("IRMV_FORCE_WRES") && lds_ok && op.w_lds[2] && counts[pass] >= 2 && conv_wres_bytes(a, op.cfg.stride, false) > 0) ok = false;   // parity tests: always the forced form
                    if (ok && !h.lds && !h.pw) ok = h.ipw == 1;
                    if (ok && h.deep) ok = (h.mt == 1 || (h.mt == 2 && h.nt == 1)) && !op.cfg.cin16 && !op.cfg.out_f32 && op.cfg.act == 1;
                    if (ok && h.ct) ok = op.w_lds[0] != nullptr;
                    if (ok && h.pf2) ok = h.lds && h.mt == 1 && !want_fuse;
                    if (ok && h.pf4) ok = !getenv("IRMV_NO_PF4") && h.lds && !h.pf2 && !h.cm && !h.w8 && !h.wr && h.mt == 1 && h.nt == 1 && h.ipw == 1 && op.cin >= 128 && !want_fuse;   // (the only form the tuner generates, times and the bitwise tests cover)
                    if (ok && h.cm && h.nt != 8) ok = !getenv("IRMV_NO_CM") && h.lds && !h.pf2 && h.cm == h.ipw && h.nt == 4 && ((h.mt == 1 && (!want_fuse || h.cm == 4)) || (h.mt == 2 && h.cm == 2));
                    if (ok && h.w8 && h.nt == 4) ok = !getenv("IRMV_NO_W8") && h.lds && op.cfg.stride == 2 && !want_fuse && !h.pf2 && (h.mt == 1 || h.mt == 2) &&
                                                     (h.cm == 0 || (h.cm == h.ipw && ((h.mt == 2 && h.cm == 2) || (h.mt == 1 && h.cm == 4))));
                    if (ok && h.w8 && h.nt == 8) ok = !getenv("IRMV_NO_W8") && !getenv("IRMV_NO_NT8") && h.lds && op.cfg.stride == 2 && !want_fuse && !h.pf2 && h.mt == 1 && op.w_lds[3] &&
                                                     (h.cm == 0 || (h.cm == 2 && h.ipw == 2));
                    if (ok && h.w8 && h.nt != 4 && h.nt != 8) ok = false;
                    if (ok && h.cm && h.nt == 8 && !h.w8) ok = false;
                    if (ok && !h.w8 && getenv("IRMV_FORCE_W8") && lds_ok && op.cfg.stride == 2) ok = false;
                    if (ok && !h.pf4 && getenv("IRMV_FORCE_PF4") && lds_ok && op.cin >= 128 && !want_fuse) ok = false;   // parity tests: re-tune so that the forced tile is tried
                    if (ok && h.nt != 8 && getenv("IRMV_FORCE_NT8") && lds_ok && op.cfg.stride == 2 && op.w_lds[3] && !want_fuse) ok = false;   // parity tests: the 128-channel workgroup wherever it exists
                    if (ok && !h.cm && getenv("IRMV_FORCE_CM") && lds_ok && counts[pass] >= 2) ok = false;   // parity tests: the chunk-major tiles wherever one exists
                    if (ok) {
                        best_cfg = op.cfg;
                        best_cfg.mt = h.mt; best_cfg.nt = h.nt; best_cfg.lds = h.lds; best_cfg.ipw = h.ipw; best_cfg.deep = h.deep; best_cfg.ct = h.ct; best_cfg.pw = h.pw; best_cfg.pf2 = h.pf2; best_cfg.pf4 = h.pf4; best_cfg.cm = h.cm; best_cfg.w8 = h.w8; best_cfg.wr = h.wr; best_cfg.pp = h.pp;
                        best = 0.f;
                        have_hit = true;
                    } else if (getenv("IRMV_AUTOTUNE_VERBOSE") || getenv("IRMV_TUNE_WARN"))
                        fprintf(stderr, "[autotune] cached tile for %s rejected (mt %d nt %d lds %d ipw %d): re-tuning\n", key, h.mt, h.nt, (int)h.lds, h.ipw);
                }
            }
            if (!have_hit)
            {
                auto time_cfg = [&](const ConvCfg &c) -> int {
                    if (no_tuning && best < 1e29f) return IRMV_OK;   // IRMV_AUTOTUNE=0: the first candidate that runs is the choice
                    if (force_s2 && lds_ok && op.cfg.stride == 2) {
                        const char *kind = c.lds ? "lds" : (c.deep ? "deep" : "ct");
                        if (strcmp(kind, force_s2) != 0) return IRMV_OK;
                    }
                    bool ok = true;
                    for (int i = 0; i < 2 && ok; i++) ok = run_conv(op, c, a, counts[pass], e->stream);
                    if (!ok) return IRMV_OK;
                    if (no_tuning) { best = 0.f; best_cfg = c; return IRMV_OK; }
                    float ms = 1e30f;   // best of 3 bursts of 4
                    for (int rep = 0; rep < 3; rep++) {
                        HIP_TRY(hipEventRecord(ea, e->stream));
                        for (int i = 0; i < 4; i++) run_conv(op, c, a, counts[pass], e->stream);
                        HIP_TRY(hipEventRecord(eb, e->stream));
                        HIP_TRY(hipEventSynchronize(eb));
                        float t = 0.f;
                        HIP_TRY(hipEventElapsedTime(&t, ea, eb));
                        ms = t < ms ? t : ms;
                    }
                    if (verbose) {
                        char nm[48];
                        cfg_name(c, nm, sizeof nm);
                        fprintf(stderr, "[autotune] %-22s count=%-2d %-28s %8.2f us\n", op.layer.c_str(), counts[pass], nm, ms / 4 * 1e3);
                    }
                    if (ms < best) { best = ms; best_cfg = c; }
                    return IRMV_OK;
                };
                const int fam = lds_ok ? 1 : 0;
                for (int mt = 1; mt <= 4; mt *= 2)
                    for (int nt = 1; nt <= 4; nt *= 2)
                        for (int ipw = 1; ipw <= (fam == 1 ? std::min(4, counts[pass]) : 1); ipw *= 2) {
                            if (op.cout_pad % (16 * nt) != 0) continue;
                            if (want_fuse && nt != 4) continue;
                            ConvCfg c = tile_cfg(op.cfg, mt, nt, fam == 1, ipw);
                            TRY(time_cfg(c));
                        }
                // LDS family, smallest pixel tile, staging two steps ahead (layers whose step is shorter than a memory round trip)
                if (fam == 1 && !want_fuse && !getenv("IRMV_NO_PF2"))
                    for (int nt = 1; nt <= 4; nt *= 2)
                        for (int ipw = 1; ipw <= std::min(4, counts[pass]); ipw *= 2) {
                            if (op.cout_pad % (16 * nt) != 0) continue;
                            ConvCfg c = tile_cfg(op.cfg, 1, nt, true, ipw);
                            c.pf2 = true;
                            TRY(time_cfg(c));
                        }
                // ... and four steps ahead: layers of four or more chunks on maps small enough for four register sets (a lone frame's
                // 20 x 20 layers: every step of a workgroup in flight at once)
                if (fam == 1 && !want_fuse && !getenv("IRMV_NO_PF4") && op.cin >= 128 && op.w_lds[0]) {
                    ConvCfg c = tile_cfg(op.cfg, 1, 1, true, 1);
                    c.pf4 = true;
                    TRY(time_cfg(c));
                    if (getenv("IRMV_FORCE_PF4") && run_conv(op, c, a, counts[pass], e->stream)) { best = 0.f; best_cfg = c; }   // parity tests: wherever it exists
                }
                // LDS family, chunk-major over the workgroup's images: a chunk's weights staged once for all of them
                if (fam == 1 && op.cout_pad % 64 == 0 && !getenv("IRMV_NO_CM"))
                    for (int mt = 1; mt <= 2; mt *= 2)
                        for (int ipw = 2; ipw <= std::min(mt == 1 ? 4 : 2, counts[pass]); ipw *= 2) {
                            if (want_fuse && mt == 1 && ipw == 2) continue;   // (no instantiation with the fused 1x1)
                            ConvCfg c = tile_cfg(op.cfg, mt, 4, true, ipw);
                            c.cm = ipw;
                            TRY(time_cfg(c));
                            if (getenv("IRMV_FORCE_CM") && run_conv(op, c, a, counts[pass], e->stream)) { best = 0.f; best_cfg = c; }   // parity tests
                        }
                // LDS family, stride 2: one 8-wave workgroup per CU on a block twice as tall (mt = 2 fits LDS, weights staged for
                // twice the pixels)
                if (fam == 1 && op.cfg.stride == 2 && !want_fuse && op.cout_pad % 64 == 0 && !getenv("IRMV_NO_W8"))
                    for (int mt = 1; mt <= 2; mt *= 2)
                        for (int ipw = 1; ipw <= std::min(4, counts[pass]); ipw *= 2)
                            for (int cmv = 0; cmv < 2; cmv++) {
                                if (cmv && !((mt == 2 && ipw == 2) || (mt == 1 && ipw == 4))) continue;
                                ConvCfg c = tile_cfg(op.cfg, mt, 4, true, ipw);
                                c.w8 = true; c.cm = cmv ? ipw : 0;
                                TRY(time_cfg(c));
                                if (getenv("IRMV_FORCE_W8") && run_conv(op, c, a, counts[pass], e->stream)) { best = 0.f; best_cfg = c; }   // parity tests
                            }
                // ... and with all of 128 output channels per workgroup (nt = 8, mt = 1): these layers are bound by what a CU can stage
                // from L2 (12 B/clk), and the stride-2 patch -- four input pixels per output pixel -- is then fetched once per 128
                // channels instead of once per 64; chunk-major over two images halves the weight staging on top
                if (fam == 1 && op.cfg.stride == 2 && !want_fuse && op.cout_pad % 128 == 0 && op.w_lds[3] && !getenv("IRMV_NO_W8") && !getenv("IRMV_NO_NT8"))
                    for (int ipw = 1; ipw <= std::min(4, counts[pass]); ipw *= 2)
                        for (int cmv = 0; cmv < 2; cmv++) {
                            if (cmv && ipw != 2) continue;
                            ConvCfg c = tile_cfg(op.cfg, 1, 8, true, ipw);
                            c.w8 = true; c.cm = cmv ? 2 : 0;
                            TRY(time_cfg(c));
                            if (getenv("IRMV_FORCE_NT8") && run_conv(op, c, a, counts[pass], e->stream)) { best = 0.f; best_cfg = c; }   // parity tests
                        }
                // LDS family, Cin = Cout = 64, stride 1: resident weights (one 8-wave workgroup per CU walks ipw images at its tile
                // position; lockstep, or as two ping-pong groups of four waves).  ipw: the smallest that lets the chip hold the
                // grid in one round, and half of it.
                if (fam == 1 && counts[pass] >= 2 && op.w_lds[2] && !getenv("IRMV_NO_WRES"))
                    for (int ppv = 0; ppv < 2; ppv++) {
                        const int wgt = conv_wres_tiles(a, op.cfg.stride, ppv != 0);
                        if (wgt <= 0) continue;
                        int ipw1 = 1;
                        while (ipw1 < counts[pass] && (long)wgt * ((counts[pass] + ipw1 - 1) / ipw1) > e->num_cus) ipw1++;
                        const int cand[3] = {ipw1, (ipw1 + 1) / 2, std::min(counts[pass], 2 * ipw1)};
                        for (int k = 0; k < 3; k++) {
                            if (k > 0 && (cand[k] == cand[0] || (k == 2 && cand[2] == cand[1]))) continue;
                            ConvCfg c = tile_cfg(op.cfg, 2, 4, true, cand[k]);
                            c.wr = true; c.pp = ppv != 0;
                            TRY(time_cfg(c));
                        }
                    }
                // parity tests: IRMV_FORCE_WRES=<n> puts every eligible layer on the resident-weights kernel with n images per
                // workgroup -- the ping-pong form where the map tiles into its blocks (n < 0: never), else the lockstep form
                if (const char *fw = getenv("IRMV_FORCE_WRES"); fw && fam == 1 && counts[pass] >= 2 && op.w_lds[2]) {
                    const int v = atoi(fw);
                    for (int ppv = v > 0 ? 1 : 0; ppv >= 0; ppv--) {
                        ConvCfg c = tile_cfg(op.cfg, 2, 4, true, std::max(1, std::min(counts[pass], v < 0 ? -v : v)));
                        c.wr = true; c.pp = ppv != 0;
                        if (conv_wres_bytes(a, op.cfg.stride, c.pp) > 0 && run_conv(op, c, a, counts[pass], e->stream)) { best = 0.f; best_cfg = c; break; }
                    }
                }
                // 1x1 layers: the persistent pointwise kernel (same operands, same k order as the direct kernel)
                if (conv_pw_eligible(op.cfg, a) && !getenv("IRMV_NO_PW")) {
                    ConvCfg c = tile_cfg(op.cfg, 2, 4, false, 1);
                    c.pw = true;
                    TRY(time_cfg(c));
                    if (getenv("IRMV_FORCE_PW") && run_conv(op, c, a, counts[pass], e->stream)) { best = 0.f; best_cfg = c; }   // parity tests
                    // ... and its multi-block form: one workgroup runs a pixel tile against 2 / 4 output-channel blocks (input read once)
                    for (int nbw = 2; nbw <= 4 && !getenv("IRMV_NO_PWN"); nbw *= 2) {
                        if (!conv_pw_lds_bytes(a, nbw)) continue;
                        c.ipw = nbw;
                        TRY(time_cfg(c));
                        if (getenv("IRMV_FORCE_PWN") && run_conv(op, c, a, counts[pass], e->stream)) { best = 0.f; best_cfg = c; }   // parity tests: the widest form offered
                    }
                }
                // A layer of the LDS family may also run on the direct kernel walking K in that family's order on its weights (ct):
                // bit-identical, so the family rule above still holds.  Offered where the direct kernel has a chance: stride 2.
                if (lds_ok && !want_fuse && op.w_lds[0] && op.cfg.stride == 2 && !op.cfg.cin16)
                    for (int mt = 1; mt <= 4; mt *= 2)
                        for (int nt = 1; nt <= 4; nt *= 2) {
                            if (op.cout_pad % (16 * nt) != 0) continue;
                            ConvCfg c = tile_cfg(op.cfg, mt, nt, false, 1);
                            c.ct = true;
                            TRY(time_cfg(c));
                        }
                // single-frame steps: the latency variants of the direct kernel (deep prefetch ring), same rule.
                static const bool no_deep = [] { const char *v = getenv("IRMV_NO_DEEP"); return v && v[0] == '1'; }();
                // Batched steps: offered to the 1x1 layers only (4..16 k-steps: the ring then holds the wave's whole K range).
                if ((counts[pass] == 1 || op.cfg.ks == 1) && !want_fuse && !no_deep && !op.cfg.cin16 && !op.cfg.out_f32 && op.cfg.act == 1 && (!lds_ok || op.w_lds[0])) {
                    const int tiles[4][2] = {{1, 1}, {2, 1}, {1, 2}, {1, 4}};
                    for (auto &t : tiles) {
                        if (op.cout_pad % (16 * t[1]) != 0) continue;
                        ConvCfg c = tile_cfg(op.cfg, t[0], t[1], false, 1);
                        c.deep = true; c.ct = lds_ok;
                        TRY(time_cfg(c));
                    }
                }
            }
            if (!no_tuning && !(force_s2 && lds_ok && op.cfg.stride == 2)) { std::lock_guard<std::mutex> lk(g_tune_mu); g_tune_cache[key] = best_cfg; }   // (forced / untuned choices are not what a later engine should replay)
            if (pass == 0) { op.cfg = best_cfg; cfg_name(op.cfg, op.kname, sizeof op.kname); }
            else { op.cfg_one = best_cfg; cfg_name(op.cfg_one, op.kname_one, sizeof op.kname_one); }
            if (counts[0] == 1) { op.cfg_one = op.cfg; cfg_name(op.cfg_one, op.kname_one, sizeof op.kname_one); }
        }
    }
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    HIP_TRY(hipGetLastError());
    tune_cache_save();
    return IRMV_OK;
}

// A 3x3 conv carries its branch's final 1x1 only if BOTH of its tile choices can (LDS family, nt = 4); then the 1x1
// op drops out of the step and the tensor between the two is no longer written.
static void finalize_head_fusion(irmv_engine *e)
{
    struct Tail {
        irmv_engine *e;
        ~Tail()
        {
            // candidate emission from the conv epilogues: only if the class branch of EVERY level ends in a fused 1x1
            int fused = 0;
            for (const Op &op : e->ops)
                if (op.kind == OP_CONV && op.fuse_next >= 0 && op.layer.rfind("model.22.cv3.", 0) == 0) fused++;
            const char *ev = getenv("IRMV_EMIT_SCAN");
            e->emit_scan = e->split_scan && fused == 3 && !(ev && ev[0] == '0');
        }
    } tail{e};
    for (Op &op : e->ops) {
        if (op.fuse_next < 0) continue;
        const bool k16 = op.cfg.cin16 && !op.cfg.lds && !op.cfg_one.lds && !op.cfg.deep && !op.cfg_one.deep && op.cfg.nt == 1 && op.cfg_one.nt == 1 && e->ops[op.fuse_next].w_k16;
        const bool ok = k16 || (op.cfg.lds && op.cfg.nt == 4 && op.cfg_one.lds && op.cfg_one.nt == 4);
        if (!ok) { op.fuse_next = -1; continue; }
        e->ops[op.fuse_next].fused_away = true;
        e->lazy_tensors.insert(e->tensors[op.out_t].name);
        const size_t l = strlen(op.kname), l1 = strlen(op.kname_one);
        snprintf(op.kname + l, sizeof op.kname - l, "+1x1");
        snprintf(op.kname_one + l1, sizeof op.kname_one - l1, "+1x1");
    }
}

// ---- grouped Detect-branch launches (single-frame engines) ---------------------------
static void scan_args_for(const irmv_engine *e, const Op &op, const PostArgs &pa, ConvArgs &a)
{
    a.scan_keys = pa.keys; a.scan_counts = pa.counts; a.scan_thr = pa.logit_thr; a.scan_nc = pa.nc;
    a.scan_key_cap = pa.key_cap;
    const int w0 = e->cfg.net_size / 8;
    a.scan_abase = op.level == 0 ? 0 : (op.level == 1 ? w0 * w0 : w0 * w0 + (w0 / 2) * (w0 / 2));
}

static bool is_cls_final_carrier(const Op &op) { return op.fuse_next >= 0 && op.level >= 0 && op.layer.rfind("model.22.cv3.", 0) == 0; }

// one launch for all members of group g on slot `first`; pa == nullptr: no candidate emission (timing, profile repeats)
static bool launch_head_group(const irmv_engine *e, const irmv_engine::HeadGroup &g, int first, const PostArgs *pa, hipStream_t s)
{
    ConvArgs a[kMultiMax];
    const half_t *wl[kMultiMax];
    const int n = (int)g.members.size();
    for (int k = 0; k < n; k++) {
        const Op &op = e->ops[g.members[k]];
        fill_conv_args(e, op, first, 1, a[k], true);
        if (pa && e->emit_scan && is_cls_final_carrier(op)) scan_args_for(e, op, *pa, a[k]);
        wl[k] = op.w_lds[g.nt == 4 ? 2 : (g.nt == 2 ? 1 : 0)];
        if (g.family == 0 && !wl[k]) return false;
    }
    return g.family == 0 ? launch_conv_lds_multi(g.nt, a, wl, n, 1, s) : launch_conv_direct_multi(g.cfg, a, n, s);
}

static int build_head_groups(irmv_engine *e)
{
    const char *gh = getenv("IRMV_GROUP_HEAD");
    if (!e->merge_head0 || (gh && gh[0] == '0')) return IRMV_OK;
    auto find_op = [&](const std::string &layer) {
        for (size_t i = 0; i < e->ops.size(); i++)
            if (e->ops[i].kind == OP_CONV && !e->ops[i].fused_away && e->ops[i].layer == layer) return (int)i;
        return -1;
    };
    auto time_of = [&](auto &&launch) {   // ms per launch, or < 0 if it cannot run
        for (int i = 0; i < 2; i++) if (!launch()) return -1.f;
        if (hipStreamSynchronize(e->stream) != hipSuccess) return -1.f;
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1.f;
        (void)hipEventRecord(a, e->stream);
        for (int i = 0; i < 8; i++) launch();
        (void)hipEventRecord(b, e->stream);
        (void)hipEventSynchronize(b);
        float ms = -1.f;
        (void)hipEventElapsedTime(&ms, a, b);
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        return ms / 8.f;
    };
    auto members_time = [&](const std::vector<int> &mem) {
        return time_of([&] {
            for (int i : mem) {
                const Op &op = e->ops[i];
                ConvArgs a;
                fill_conv_args(e, op, 0, 1, a, true);
                if (!run_conv(op, op.cfg_one, a, 1, e->stream)) return false;
            }
            return true;
        });
    };
    auto try_group = [&](std::vector<int> mem, int family, std::vector<int> nts, const char *label) {
        for (int i : mem) if (i < 0) return;
        if (mem.size() < 2 || mem.size() > (size_t)kMultiMax) return;
        irmv_engine::HeadGroup best;
        float best_ms = -1.f;
        for (int nt : nts) {
            irmv_engine::HeadGroup g;
            g.members = mem; g.family = family; g.nt = nt;
            if (family == 1) {
                g.cfg = e->ops[mem[0]].cfg_one;
                bool same = !g.cfg.lds && !g.cfg.ct && !g.cfg.pw;   // (the members' own tile shapes and prefetch depths are bitwise neutral: the group runs mt = nt = 1)
                for (int i : mem) {
                    const ConvCfg &c = e->ops[i].cfg_one;
                    same = same && c.ks == g.cfg.ks && c.stride == g.cfg.stride && c.cin16 == g.cfg.cin16 && c.act == g.cfg.act && c.out_f32 == g.cfg.out_f32 && !c.lds && !c.ct && !c.pw;
                }
                if (!same) continue;
                g.cfg.mt = g.cfg.nt = 1; g.cfg.deep = false;
            } else {
                bool fam = true;   // every member must belong to the LDS family's K order (its single-frame choice is an LDS tile or the CT stand-in)
                for (int i : mem) fam = fam && (e->ops[i].cfg_one.lds || e->ops[i].cfg_one.ct);
                if (!fam) continue;
            }
            if (getenv("IRMV_GROUP_VERBOSE")) { fprintf(stderr, "[irmv group] timing %s nt %d ...\n", label, nt); fflush(stderr); }
            const float ms = time_of([&] { return launch_head_group(e, g, 0, nullptr, e->stream); });
            if (getenv("IRMV_GROUP_VERBOSE")) { fprintf(stderr, "[irmv group] ... %.2f us\n", ms * 1e3f); fflush(stderr); }
            if (ms > 0.f && (best_ms < 0.f || ms < best_ms)) { best_ms = ms; best = g; }
        }
        if (best_ms < 0.f) return;
        const float sep = members_time(mem);
        if (getenv("IRMV_AUTOTUNE_VERBOSE") || getenv("IRMV_GROUP_VERBOSE")) fprintf(stderr, "[irmv group] %s: %zu convs, one launch %.2f us, separate %.2f us\n", label, mem.size(), best_ms * 1e3f, sep * 1e3f);
        if (sep > 0.f && best_ms >= sep && !getenv("IRMV_GROUP_FORCE")) return;   // IRMV_GROUP_FORCE=1 (parity test): group even where the one launch timed slower
        if (family == 0) snprintf(best.name, sizeof best.name, "%s_lds_mt1_nt%d_x%zu", label, best.nt, mem.size());
        else snprintf(best.name, sizeof best.name, "%s_direct_x%zu", label, mem.size());
        const int gi = (int)e->head_groups.size();
        e->head_groups.push_back(best);
        for (int i : mem) e->ops[i].group = gi;
    };
    const bool kpt = e->nk > 0;
    std::vector<int> g1, g2, g3, g4;
    bool g2_fused = true;
    for (int i = 0; i < 3; i++) {
        const std::string si = std::to_string(i);
        g1.push_back(find_op("model.22.s0." + si));
        const int c2 = find_op("model.22.cv2." + si + ".1"), c3 = find_op("model.22.cv3." + si + ".1");
        g2.push_back(c2); g2.push_back(c3);
        g2_fused = g2_fused && c2 >= 0 && c3 >= 0 && e->ops[c2].fuse_next >= 0 && e->ops[c3].fuse_next >= 0;
        if (kpt) { g3.push_back(find_op("model.22.cv4." + si + ".1")); g4.push_back(find_op("model.22.cv4." + si + ".2")); }
    }
    try_group(g1, 0, {1, 2}, "head_s0");
    if (g2_fused) try_group(g2, 0, {4}, "head_s1+1x1");
    // A group is launched where its FIRST member stands in the op list, so every member's input must exist by then.  The
    // first-stage convs precede all of these; the keypoint finals (cv4.i.2) read cv4.i.1, and cv4.1.1 / cv4.2.1 stand BEHIND
    // cv4.0.2 in the list: the finals may only be grouped when the convs before them are (one launch, at cv4.0.1's place).
    if (kpt) {
        const size_t before = e->head_groups.size();
        try_group(g3, 1, {1}, "head_kpt1");
        bool k16 = false;   // (finals on the 16x16x16 MFMA are fused into the convs in front of them, or run launch_conv_k16: the grouped direct kernel would give other bits)
        for (int i : g4) k16 = k16 || i < 0 || e->ops[i].w_k16 != nullptr;
        if (e->head_groups.size() > before && !k16) try_group(g4, 1, {1}, "head_kpt2");
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    return IRMV_OK;
}

// ---- step execution ------------------------------------------------------------
struct EvRec { hipEvent_t a = nullptr, b = nullptr; int op = -1; };
constexpr uint32_t kProfileRepeat = 4;   // launches per event bracket in irmv_engine_profile

static void fill_conv_args(const irmv_engine *e, const Op &op, int first, int count, ConvArgs &a, bool fused)
{
    a = ConvArgs{};
    auto seg = [&](const SegRef &s) {
        ConvSeg cs{nullptr, 0, 0, 0};
        if (s.t < 0 || s.C == 0) return cs;
        const Tensor &t = e->tensors[s.t];
        cs.p = static_cast<const half_t *>(t.slot(first)) + s.coff;
        cs.ld = t.C;
        cs.C = s.C;
        cs.shift = s.shift;
        return cs;
    };
    a.s0 = seg(op.s0);
    a.s1 = seg(op.s1);
    a.Hin = op.Hin; a.Win = op.Win; a.Hout = op.Hout; a.Wout = op.Wout;
    a.M = count * op.Hout * op.Wout;
    a.Cin = op.cin;
    a.w = op.w_packed;
    a.bias = op.bias;
    const Tensor &ot = e->tensors[op.out_t];
    a.out = static_cast<char *>(ot.slot(first)) + (size_t)op.out_coff * ot.esize();
    a.out_ld = ot.C;
    a.res = nullptr;
    a.res_ld = 0;
    if (op.res_t >= 0) {
        const Tensor &rt = e->tensors[op.res_t];
        a.res = static_cast<const half_t *>(rt.slot(first)) + op.res_coff;
        a.res_ld = rt.C;
    }
    a.cout_pad = op.cout_pad;
    a.ksteps = op.ksteps;
    a.pair = op.pair ? 1 : 0;
    a.w2 = nullptr; a.bias2 = nullptr; a.out2 = nullptr; a.out2_ld = 0; a.n2 = 0;
    if (fused && op.fuse_next >= 0) {
        const Op &o2 = e->ops[op.fuse_next];
        const Tensor &t2 = e->tensors[o2.out_t];
        a.w2 = op.cfg.cin16 ? o2.w_k16 : o2.w_packed;
        a.bias2 = o2.bias;
        a.out2 = static_cast<float *>(t2.slot(first)) + o2.out_coff;
        a.out2_ld = t2.C;
        a.n2 = o2.cout_pad / 16;
    }
    if (op.w_k16) a.w2 = op.w_k16;   // a Cin = 16 final as its own launch (launch_conv_k16)
}

static LightArgs light_args(const irmv_engine *e, int first)
{
    LightArgs a{};
    const irmv_engine_cfg &c = e->cfg;
    a.frames = e->src_dev + (size_t)first * e->frame_bytes;
    a.frame_bytes = e->frame_bytes;
    a.cols = c.src_width; a.rows = c.src_height; a.rotate180 = c.rotate180;
    a.dets = e->dets_dev + (size_t)first * c.max_det;
    a.max_det = c.max_det;
    a.num_dets = reinterpret_cast<const int *>(e->fout_dev + first);
    a.num_dets_stride = (int)(sizeof(DevFrameOut) / sizeof(int));
    a.n_boxes = 0;
    a.boxes = nullptr;
    a.labels = e->light_labels + (size_t)first * e->light_pool;
    a.label_pool = e->light_pool;
    a.points = e->light_points + (size_t)first * c.max_det * kLightPointsCap * 2;
    a.points_cap = kLightPointsCap;
    a.hulls = e->light_hulls + (size_t)first * c.max_det * kLightPointsCap * 4;
    a.binary_threshold = c.binary_threshold;
    a.light_min_ratio = c.light_min_ratio; a.light_max_ratio = c.light_max_ratio; a.light_max_angle = c.light_max_angle;
    a.min_small_cd = c.armor_min_small_center_distance; a.max_small_cd = c.armor_max_small_center_distance;
    a.min_large_cd = c.armor_min_large_center_distance; a.max_large_cd = c.armor_max_large_center_distance;
    a.pnp = e->pnp_dev;
    a.pnp_armor_size = c.armor_size;
    return a;
}

static PostArgs post_args(const irmv_engine *e, int first)
{
    PostArgs p = e->post;
    p.head_all = e->head_all;
    p.slots_total = e->cfg.num_slots;
    p.first = first;
    p.boxes = e->boxes + (size_t)first * e->A * 4;
    p.keys = e->keys + (size_t)first * e->A * e->nc;
    p.key_cap = e->A * e->nc;
    p.dets = (e->zero_copy_results ? e->dets_host_dev : e->dets_dev) + (size_t)first * e->cfg.max_det;
    p.fout = (e->zero_copy_results ? e->fout_host_dev : e->fout_dev) + first;
    if (p.dbg) p.dbg += (size_t)first * 16;
    p.counts = e->split_scan ? e->cand_counts + first : nullptr;
    return p;
}

// Enqueue one step on the engine stream.  ev != nullptr: bracket every kernel with events.
// one fused C2f block (OP_C2F32).  (A 16 x 16 tile on an 8-wave workgroup -- a third less halo work, one workgroup per CU --
// was built and measured in round 3: 5 - 30 % slower than the 8 x 16 tile in an eager replay, a tie in the benchmarked one;
// dropped.)
static bool launch_c2f32_op(const irmv_engine *e, const Op &op, int first, int count, hipStream_t s)
{
    C2f32Args a{};
    const Tensor &ct = e->tensors[op.res_t];
    const Tensor &ot = e->tensors[op.out_t];
    if (op.sub[0] >= 0) {
        ConvArgs ca;
        fill_conv_args(e, e->ops[op.sub[0]], first, count, ca, false);
        a.s0 = ca.s0; a.s1 = ca.s1; a.cin1 = ca.Cin;
        a.w_cv1 = e->ops[op.sub[0]].w_packed; a.b_cv1 = e->ops[op.sub[0]].bias;
    } else {
        a.cin1 = 32;
    }
    a.cat = static_cast<half_t *>(ct.slot(first)); a.cat_ld = ct.C; a.prev_coff = 64;
    a.out = static_cast<half_t *>(ot.slot(first)); a.out_ld = ot.C;
    a.H = op.Hin; a.W = op.Win;
    a.tiles_x = (op.Win + kC2f32TileW - 1) / kC2f32TileW; a.tiles_y = (op.Hin + kC2f32TileH - 1) / kC2f32TileH;
    a.w_m1 = e->ops[op.sub[1]].w_packed; a.b_m1 = e->ops[op.sub[1]].bias;
    a.w_m2 = e->ops[op.sub[2]].w_packed; a.b_m2 = e->ops[op.sub[2]].bias;
    if (op.sub[3] >= 0) { a.w_cv2 = e->ops[op.sub[3]].w_packed; a.b_cv2 = e->ops[op.sub[3]].bias; }
    return launch_c2f32(op.mode, op.shortcut, a, count, s);
}

static int enqueue_step(irmv_engine *e, int first, int count, uint32_t flags, bool post_only, std::vector<EvRec> *ev)
{
    const int net = e->cfg.net_size;
    const bool capturing = (flags & 0x40000000u) != 0;
    const bool materialize = (flags & 0x20000000u) != 0;
    const PostArgs pa = post_args(e, first);
    (void)capturing;   // a step is one line of launches (the Detect branches forked onto side streams inside the captured graph
                       // measured 11 % slower: DESIGN.md section 6a; the option was removed in round 3)
    for (const Op &op : e->ops) {
        if (post_only && op.kind != OP_NMS && op.kind != OP_LIGHT && op.kind != OP_SCAN) continue;
        if (op.kind == OP_SCAN && e->emit_scan && !post_only) continue;   // the class-branch convs have already filled the key lists
        const bool grouped = op.kind == OP_CONV && op.group >= 0 && count == 1 && !materialize && !post_only;
        if (grouped && e->head_groups[op.group].members[0] != (int)(&op - e->ops.data())) continue;   // rides in its group's launch
        // a step skips the layers a fused kernel covers; a read-back runs only those (and the unfused form of a conv that
        // normally carries a 1x1 in its epilogue)
        // single-frame steps: the 64-channel Bottlenecks ride in their OP_BNECK launch; every other step runs the layers
        const bool one_frame = count == 1 && !materialize && !post_only && e->bneck64;
        if (op.kind == OP_BNECK ? !one_frame : (op.bneck >= 0 && one_frame)) continue;
        const bool fused_kpt = !materialize && !post_only;   // every step runs a level's keypoint branch as its OP_KPT3 launch (where the engine has one)
        if (op.kind == OP_KPT3 ? !fused_kpt : (op.kpt3 >= 0 && fused_kpt)) continue;
        if (materialize ? !(op.fused_away || op.fuse_next >= 0 || ((op.bneck >= 0 || op.kpt3 >= 0) && op.kind == OP_CONV)) : op.fused_away) continue;
        hipStream_t s = e->enq_stream ? e->enq_stream : e->stream;
        EvRec r{};
        r.op = (int)(&op - e->ops.data());
        if (ev) {
            HIP_TRY(hipEventCreate(&r.a));
            HIP_TRY(hipEventCreate(&r.b));
            HIP_TRY(hipEventRecord(r.a, s));
        }
        // every kernel is idempotent and can be repeated inside its event bracket -- except the light extraction and, with the
        // split scan, the scan / NMS pair (the scan appends to the frame's candidate list, the NMS kernel consumes and resets it)
        const bool once = op.kind == OP_LIGHT || (e->split_scan && (op.kind == OP_SCAN || op.kind == OP_NMS));
        const int reps = (ev && !once) ? (int)(flags & 0xffu) : 1;
        for (int rep = 0; rep < (reps > 0 ? reps : 1); rep++)
        switch (op.kind) {
        case OP_PRE: {
            PreArgs a;
            a.src = e->src_dev + (size_t)first * e->frame_bytes;
            a.dst = static_cast<half_t *>(e->tensors[e->tensor_idx.at("input")].slot(first));
            a.tx = e->tap_x; a.ty = e->tap_y;
            a.sw = e->cfg.src_width; a.sh = e->cfg.src_height; a.net = net; a.swap_rb = e->cfg.swap_rb;
            a.src_slot_bytes = e->frame_bytes;
            launch_preprocess(a, count, s);
            break;
        }
        case OP_FRONT: {
            FrontArgs a;
            a.src = e->src_dev + (size_t)first * e->frame_bytes;
            a.src_slot_bytes = e->frame_bytes;
            a.tx = e->tap_x; a.ty = e->tap_y;
            a.vx0 = e->front_v[0]; a.vx1 = e->front_v[1]; a.vy0 = e->front_v[2]; a.vy1 = e->front_v[3];
            a.sw = e->cfg.src_width; a.sh = e->cfg.src_height; a.net = net; a.swap_rb = e->cfg.swap_rb;
            a.fastx = e->front_fastx; a.fx_i0 = e->front_fx_i0; a.fx_step = e->front_fx_step;
            a.w0 = e->conv0_w; a.b0 = e->conv0_b;
            a.w1 = op.w_packed; a.b1 = op.bias;
            const Tensor &ot = e->tensors[op.out_t];
            a.out = static_cast<half_t *>(ot.slot(first));
            a.out_ld = ot.C;
            a.tiles_x = e->front_tiles_x; a.tiles_y = e->front_tiles_y; a.tile_y = e->front_tile_y;
            a.stage_bytes = e->front_stage_bytes;
            if (!launch_front(a, count, s)) return fail(IRMV_ERR_HIP, "fused front kernel: LDS request refused");
            break;
        }
        case OP_C2F2: {
            C2fArgs a;
            const Tensor &xt = e->tensors[op.s0.t], &ot = e->tensors[op.out_t];
            a.x = static_cast<const half_t *>(xt.slot(first)); a.x_ld = xt.C;
            a.out = static_cast<half_t *>(ot.slot(first)); a.out_ld = ot.C;
            a.S = xt.H; a.tiles = (xt.H + kC2fTile - 1) / kC2fTile;
            const Op &c1 = e->ops[op.sub[0]], &m1 = e->ops[op.sub[1]], &m2 = e->ops[op.sub[2]], &c2 = e->ops[op.sub[3]];
            a.w_cv1 = c1.w_packed; a.w_m1 = m1.w_packed; a.w_m2 = m2.w_packed; a.w_cv2 = c2.w_packed;
            a.b_cv1 = c1.bias; a.b_m1 = m1.bias; a.b_m2 = m2.bias; a.b_cv2 = c2.bias;
            launch_c2f2(a, count, s);
            break;
        }
        case OP_C2F32: {
            if (!launch_c2f32_op(e, op, first, count, s)) return fail(IRMV_ERR_ARG, "no fused C2f kernel for " + op.layer);
            break;
        }
        case OP_BNECK: {
            const Op &m1 = e->ops[op.sub[0]], &m2 = e->ops[op.sub[1]];
            const Tensor &ct = e->tensors[op.res_t];
            BneckArgs a{};
            a.yin = static_cast<const half_t *>(ct.slot(first)) + m1.s0.coff; a.yin_ld = ct.C;
            a.ynext = static_cast<half_t *>(ct.slot(first)) + m2.out_coff; a.ynext_ld = ct.C;
            a.cat = static_cast<const half_t *>(ct.slot(first)); a.cat_ld = ct.C;
            a.H = op.Hin; a.W = op.Win;
            a.tiles_x = (op.Win + kBneckTile - 1) / kBneckTile; a.tiles_y = (op.Hin + kBneckTile - 1) / kBneckTile;
            a.w_m1 = m1.w_lds[0]; a.b_m1 = m1.bias; a.w_m2 = m2.w_lds[0]; a.b_m2 = m2.bias;
            int ks2 = 6;
            if (op.sub[2] >= 0) {
                const Op &c2 = e->ops[op.sub[2]];
                const Tensor &ot = e->tensors[c2.out_t];
                a.out = static_cast<half_t *>(ot.slot(first)) + c2.out_coff; a.out_ld = ot.C;
                a.w_cv2 = c2.w_packed; a.b_cv2 = c2.bias;
                ks2 = c2.ksteps;
            }
            if (!launch_bneck64(op.mode, ks2, op.shortcut, a, count, s)) return fail(IRMV_ERR_ARG, "no fused bottleneck kernel for " + op.layer);
            break;
        }
        case OP_KPT3: {
            const Op &o0 = e->ops[op.sub[0]], &o1 = e->ops[op.sub[1]], &o2 = e->ops[op.sub[2]];
            const Tensor &xt = e->tensors[o0.s0.t], &ht = e->tensors[o2.out_t];
            Kpt3Args a{};
            a.x = static_cast<const half_t *>(xt.slot(first)) + o0.s0.coff; a.x_ld = xt.C;
            a.H = op.Hin; a.W = op.Win;
            a.tiles_x = (op.Win + kKpt3Tile - 1) / kKpt3Tile; a.tiles_y = (op.Hin + kKpt3Tile - 1) / kKpt3Tile;
            a.w1 = o0.w_lds[0]; a.b1 = o0.bias;
            a.w2 = o1.w_packed; a.b2 = o1.bias;
            a.w3 = o2.w_k16; a.b3 = o2.bias;
            a.out = static_cast<float *>(ht.slot(first)) + o2.out_coff; a.out_ld = ht.C;
            if (!launch_kpt3(a, op.cin, count, s)) return fail(IRMV_ERR_ARG, "no fused keypoint-branch kernel for " + op.layer);
            break;
        }
        case OP_DW: {
            DwArgs a;
            const Tensor &xt = e->tensors[op.s0.t], &ot = e->tensors[op.out_t];
            a.x = static_cast<const half_t *>(xt.slot(first)) + op.s0.coff; a.x_ld = xt.C;
            a.y = static_cast<half_t *>(ot.slot(first)) + op.out_coff; a.y_ld = ot.C;
            a.w = op.w_packed; a.b = op.bias;
            a.Hin = op.Hin; a.Win = op.Win; a.Hout = op.Hout; a.Wout = op.Wout; a.C = op.cout; a.stride = op.cfg.stride;
            launch_dwconv3x3(a, count, s);
            break;
        }
        case OP_SHUF: {
            ShufArgs a;
            const Tensor &at = e->tensors[op.s0.t], &bt = e->tensors[op.s1.t], &ot = e->tensors[op.out_t];
            a.a = static_cast<const half_t *>(at.slot(first)) + op.s0.coff; a.a_ld = at.C;
            a.b = static_cast<const half_t *>(bt.slot(first)) + op.s1.coff; a.b_ld = bt.C;
            a.out = static_cast<half_t *>(ot.slot(first)); a.out_ld = ot.C;
            a.bc = op.s0.C;
            a.pixels = (size_t)count * op.Hin * op.Win;
            launch_shuffle_cat(a, s);
            break;
        }
        case OP_CONV0: {
            Conv0Args a;
            a.x = static_cast<const half_t *>(e->tensors[e->tensor_idx.at("input")].slot(first));
            a.y = static_cast<half_t *>(e->tensors[e->tensor_idx.at("0")].slot(first));
            a.w = e->conv0_w; a.b = e->conv0_b; a.net = net; a.batch = count;
            launch_conv0(a, s);
            break;
        }
        case OP_CONV: {
            if (grouped) {   // (a profiled launch is repeated: only its last repetition appends candidates)
                if (!launch_head_group(e, e->head_groups[op.group], first, rep == (reps > 0 ? reps : 1) - 1 ? &pa : nullptr, s))
                    return fail(IRMV_ERR_ARG, std::string("grouped launch refused: ") + e->head_groups[op.group].name);
                break;
            }
            ConvArgs a;
            fill_conv_args(e, op, first, count, a, !materialize);
            if (e->emit_scan && !materialize && !post_only && is_cls_final_carrier(op) && rep == (reps > 0 ? reps : 1) - 1)
                scan_args_for(e, op, pa, a);   // (a profiled launch is repeated: only its last repetition appends)
            const ConvCfg &cc = (count == 1 && stream_share(e, e->cfg.num_slots) > 1) ? op.cfg_one : op.cfg;
            if (!run_conv(op, cc, a, count, s)) return fail(IRMV_ERR_ARG, std::string("no conv kernel for ") + op.kname + " (" + op.layer + ")");
            break;
        }
        case OP_POOL: {
            const Tensor &t = e->tensors[e->tensor_idx.at("9.cat")];
            launch_sppf_pool(static_cast<half_t *>(t.slot(first)), count, t.H, t.W, t.C / 4, s);
            break;
        }
        case OP_SCAN: launch_scan_decode(pa, count, s); break;
        case OP_NMS: {
            PostArgs pn = pa;
            pn.keys_only = ((e->emit_scan && !post_only) || (post_only && e->post_keys_only)) ? 1 : 0;
            launch_nms_pnp(pn, count, s);
            break;
        }
        case OP_LIGHT: launch_light_extract(light_args(e, first), e->cfg.max_det, count, s); break;
        }
        HIP_TRY(hipGetLastError());
        if (ev) {
            HIP_TRY(hipEventRecord(r.b, s));
            ev->push_back(r);
        }
    }
    return IRMV_OK;
}

// Frame upload and result download: plain async copies, pinned memory both ways, on the streams submit_group() picks.
static int copy_in(irmv_engine *e, int first, int count, hipStream_t st)
{
    HIP_TRY(hipMemcpyAsync(e->src_dev + (size_t)first * e->frame_bytes, e->src_host + (size_t)first * e->frame_bytes,
                           e->frame_bytes * count, hipMemcpyHostToDevice, st));
    return IRMV_OK;
}

static int copy_out(irmv_engine *e, int first, int count, hipStream_t st = nullptr)
{
    if (!st) st = e->stream;
    if (e->zero_copy_results) return IRMV_OK;   // the kernel has already written the pinned records
    HIP_TRY(hipMemcpyAsync(e->dets_host + (size_t)first * e->cfg.max_det, e->dets_dev + (size_t)first * e->cfg.max_det,
                           (size_t)count * e->cfg.max_det * sizeof(DevDet), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(e->fout_host + first, e->fout_dev + first, (size_t)count * sizeof(DevFrameOut),
                           hipMemcpyDeviceToHost, st));
    return IRMV_OK;
}

static int check_range(const irmv_engine *e, int first, int count)
{
    if (!e) return fail(IRMV_ERR_ARG, "engine is null");
    if (first < 0 || count < 1 || first + count > e->cfg.num_slots) return fail(IRMV_ERR_ARG, "slot range out of bounds");
    return IRMV_OK;
}

static int get_graph(irmv_engine *e, int first, int count, uint32_t flags, bool post_only, hipGraphExec_t *out)
{
    const GraphKey key{first, count, flags | (post_only ? 0x80000000u : 0u)};
    auto it = e->graphs.find(key);
    if (it != e->graphs.end()) { *out = it->second; return IRMV_OK; }
    hipGraph_t g = nullptr;
    HIP_TRY(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    int rc = IRMV_OK;
    if (flags & 0x10000000u) {   // the frames' upload as the graph's first node (synchronous single-stream submits): one or two frames as a kernel
        const size_t off = (size_t)first * e->frame_bytes, bytes = e->frame_bytes * count;
        if (count <= 2 && e->upload_kernel_blocks > 0 && e->src_host_dev && off % 16 == 0 && bytes % 16 == 0)
            launch_upload_frames(e->src_host_dev + off, e->src_dev + off, bytes, e->upload_kernel_blocks, e->stream);
        else
            rc = copy_in(e, first, count, e->stream);
    }
    if (!rc) rc = enqueue_step(e, first, count, (flags & ~0x10000000u) | 0x40000000u, post_only, nullptr);
    hipError_t ce = hipStreamEndCapture(e->stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (ce != hipSuccess) return fail(IRMV_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
    hipGraphExec_t ge = nullptr;
    HIP_TRY(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    e->graphs[key] = ge;
    *out = ge;
    return IRMV_OK;
}

static int group_of(irmv_engine *e, int first, int count, SlotGroup **out)
{
    const auto key = std::make_pair(first, count);
    auto it = e->groups.find(key);
    if (it == e->groups.end()) {
        SlotGroup g;
        g.first = first; g.count = count;
        HIP_TRY(hipEventCreateWithFlags(&g.h2d, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g.out, hipEventDisableTiming));
        it = e->groups.emplace(key, g).first;
    }
    *out = &it->second;   // std::map nodes never move
    return IRMV_OK;
}

// One slot group: [upload] -> ONE hipGraph -> download of the results, in order on compute stream `st`.
//
// IRMV_SUBMIT_ASYNC_UPLOAD moves the upload to the engine's upload stream (the dGPU form of the
// reference's TripleBuffer, include/irmv_detection/triple_buffer.hpp:24-40, whose slots ARE the engines' input memory):
//
//   upload stream    [wait: this group's previous step is done with its device frames]  H2D frames  -> ev h2d
//   compute stream   [wait: ev h2d]  hipGraph, D2H results                                           -> ev out
//
// so group B's frames cross PCIe while group A's kernels run.  The price is one cross-stream event hop per group
// (measured on this stack: scripts/probes/stream_probe.cpp), which is why a lone synchronous detect() -- nothing to
// overlap with -- keeps everything on one stream, and why the (tiny) download never leaves the compute stream.
static int submit_group(irmv_engine *e, int f, int c, uint32_t flags, hipStream_t st)
{
    const bool async_up = (flags & IRMV_SUBMIT_H2D) && (flags & IRMV_SUBMIT_ASYNC_UPLOAD) && !e->inline_copies;
    // An upload that rides the compute stream anyway (the synchronous detect()) is captured INTO the step's graph: one
    // submission instead of two, and the copy -> first kernel hand-over is the graph's own (IRMV_GRAPH_UPLOAD=0: a separate
    // hipMemcpyAsync in front of the graph, as before; same bits)
    // A synchronous single-frame step (the reference's detect(), src/yolo_engine.cpp:153-177) has two launch forms with the same
    // kernels and the same bits: ONE hipGraph replay, or its 41 launches issued one by one behind the upload (round 5: a graph
    // replay spends ~10 us of host work before its first packet reaches the GPU, a direct launch ~4; with the 70 us upload in
    // front the host stays far ahead of the GPU: 0.339 -> 0.331 ms per 1280 x 1024 frame).  Every other step is a graph replay.
    const bool eager = e->sync_launch == 1 && c == 1 && (flags & IRMV_SUBMIT_H2D) && !async_up;
    const bool graph_up = (flags & IRMV_SUBMIT_H2D) && !async_up && e->graph_upload && !eager;
    hipGraphExec_t ge = nullptr;
    if (!eager) TRY(get_graph(e, f, c, graph_up ? 0x10000000u : 0u, false, &ge));
    SlotGroup *g;
    TRY(group_of(e, f, c, &g));
    hipStream_t up = async_up ? e->h2d_stream : st;
    // slots last used through a different grouping: order behind that group's completion
    SlotGroup *seen[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int s = f; s < f + c; s++) {
        SlotGroup *o = e->slot_owner[s];
        e->slot_owner[s] = g;
        if (!o || o == g || !o->in_flight || o == seen[0] || o == seen[1] || o == seen[2] || o == seen[3]) continue;
        seen[3] = seen[2]; seen[2] = seen[1]; seen[1] = seen[0]; seen[0] = o;
        if (o->compute != st) HIP_TRY(hipStreamWaitEvent(st, o->out, 0));
        if (up != st) HIP_TRY(hipStreamWaitEvent(up, o->out, 0));
    }
    if (g->in_flight && g->compute != st) HIP_TRY(hipStreamWaitEvent(st, g->out, 0));   // the group moved to another compute stream
    if (flags & IRMV_SUBMIT_H2D) {
        if (async_up) {
            if (g->in_flight) HIP_TRY(hipStreamWaitEvent(up, g->out, 0));   // previous step has consumed the device frames
            TRY(copy_in(e, f, c, up));
            HIP_TRY(hipEventRecord(g->h2d, up));
            HIP_TRY(hipStreamWaitEvent(st, g->h2d, 0));
        } else if (!graph_up) {
            // (an eager step, or IRMV_GRAPH_UPLOAD=0)  One or two frames travel as a KERNEL (k_pre.hip upload_frame_kernel), larger groups on the copy engine.
            const size_t off = (size_t)f * e->frame_bytes, bytes = e->frame_bytes * c;
            if (c <= 2 && e->upload_kernel_blocks > 0 && e->src_host_dev && off % 16 == 0 && bytes % 16 == 0)
                launch_upload_frames(e->src_host_dev + off, e->src_dev + off, bytes, e->upload_kernel_blocks, st);
            else
                TRY(copy_in(e, f, c, st));
        }
    }
    if (eager) {
        e->enq_stream = st;
        const int rc = enqueue_step(e, f, c, 0, false, nullptr);
        e->enq_stream = nullptr;
        if (rc) return rc;
    } else {
        HIP_TRY(hipGraphLaunch(ge, st));
    }
    TRY(copy_out(e, f, c, st));
    HIP_TRY(hipEventRecord(g->out, st));
    g->in_flight = true;
    g->async_up = async_up;
    g->compute = st;
    return IRMV_OK;
}

extern "C" int irmv_engine_submit(irmv_engine *e, int first, int count, uint32_t flags)
{
    TRY(check_range(e, first, count));
    HIP_TRY(hipSetDevice(e->cfg.device));
    // A multi-slot step is cut into num_streams independent sub-batches, one captured graph each, on
    // separate streams: while one sub-batch sits in a launch gap or a kernel tail the other keeps the
    // CUs busy (two sub-batches measured +30 % frames/s over one stream at 32 frames).
    int share = count > 1 ? stream_share(e, count) : count;
    // Single-slot steps ride the compute stream of their slot (slot mod num_streams): a single frame fills a fraction of
    // the chip, so the steps of two slots in flight (the TripleBuffer's depth) overlap instead of queueing behind each other.
    int si = count == 1 ? first % e->num_streams : 0;
    // A submit of exactly ONE stream's share of the engine's slots, aligned to it, is that share's sub-batch of a whole-engine
    // step: the same captured graph on the same stream.  A caller that feeds the shares separately decides itself how far
    // apart the streams run (two shares started together execute the same kernel at the same time all the way down).
    const int full_share = stream_share(e, e->cfg.num_slots);
    if (count > 1 && count == full_share && first % full_share == 0 && first / full_share < e->num_streams) { share = count; si = first / full_share; }
    for (int f = first; f < first + count; f += share, si++) {
        const int c = std::min(share, first + count - f);
        hipStream_t st = si == 0 ? e->stream : e->extra_streams[si - 1];
        TRY(submit_group(e, f, c, flags, st));
    }
    return IRMV_OK;
}

extern "C" int irmv_engine_wait(irmv_engine *e);

extern "C" int irmv_engine_debug_poke_candidate_counts(irmv_engine *e, int value)
{
    if (!e) return fail(IRMV_ERR_ARG, "engine is null");
    if (!e->cand_counts) return fail(IRMV_ERR_ARG, "this engine keeps no candidate counters");
    TRY(irmv_engine_wait(e));
    HIP_TRY(hipSetDevice(e->cfg.device));
    std::vector<int> v((size_t)e->cfg.num_slots, value);
    HIP_TRY(hipMemcpy(e->cand_counts, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
    return IRMV_OK;
}

extern "C" int irmv_engine_run_post(irmv_engine *e, int first, int count)
{
    TRY(check_range(e, first, count));
    HIP_TRY(hipSetDevice(e->cfg.device));
    TRY(irmv_engine_wait(e));
    hipGraphExec_t ge;
    TRY(get_graph(e, first, count, 0, true, &ge));
    HIP_TRY(hipGraphLaunch(ge, e->stream));
    TRY(copy_out(e, first, count));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return IRMV_OK;
}

extern "C" int irmv_engine_wait(irmv_engine *e)
{
    if (!e) return fail(IRMV_ERR_ARG, "engine is null");
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipStreamSynchronize(e->h2d_stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int i = 1; i < e->num_streams; i++) HIP_TRY(hipStreamSynchronize(e->extra_streams[i - 1]));
    for (auto &kv : e->groups) kv.second.in_flight = false;
    return IRMV_OK;
}

// Block until the pinned slots [first, first + count) have been read by their last submit's upload: from then on a producer
// may overwrite them (the moment the TripleBuffer's consumer can give the buffer back), while the kernels still run.
extern "C" int irmv_engine_wait_upload(irmv_engine *e, int first, int count)
{
    TRY(check_range(e, first, count));
    HIP_TRY(hipSetDevice(e->cfg.device));
    SlotGroup *last = nullptr;
    for (int s = first; s < first + count; s++) {
        SlotGroup *o = e->slot_owner[s];
        if (!o || o == last || !o->in_flight) continue;
        // an upload on the side stream has its own event; an inline upload is ordered in front of the kernels, so the
        // group's completion event covers it
        HIP_TRY(hipEventSynchronize(o->async_up ? o->h2d : o->out));
        last = o;
    }
    return IRMV_OK;
}

// Block until the results of slots [first, first + count) from their last submit are host-visible; other slots may
// stay in flight (the consumer side of the TripleBuffer: take the newest finished slot while the next one runs).
extern "C" int irmv_engine_wait_slots(irmv_engine *e, int first, int count)
{
    TRY(check_range(e, first, count));
    HIP_TRY(hipSetDevice(e->cfg.device));
    SlotGroup *last = nullptr;
    for (int s = first; s < first + count; s++) {
        SlotGroup *o = e->slot_owner[s];
        if (!o || o == last || !o->in_flight) continue;
        HIP_TRY(hipEventSynchronize(o->out));
        // the whole group is done only if this call covers it; otherwise it merely stays marked in flight (harmless)
        if (o->first >= first && o->first + o->count <= first + count) o->in_flight = false;
        last = o;
    }
    return IRMV_OK;
}

extern "C" int irmv_engine_results(irmv_engine *e, int slot, irmv_det *out, int cap, int *n)
{
    TRY(check_range(e, slot, 1));
    if (!n || (cap > 0 && !out)) return fail(IRMV_ERR_ARG, "out/n is null");
    const DevFrameOut &fo = e->fout_host[slot];
    const int k = std::min(fo.num_dets, cap);
    const DevDet *d = e->dets_host + (size_t)slot * e->cfg.max_det;
    for (int i = 0; i < k; i++) {
        irmv_det &o = out[i];
        memcpy(o.xyxy, d[i].xyxy, 16);
        o.score = d[i].score;
        // magic_enum::enum_cast<ArmorClass>(label).value_or(UNKNOWN), src/yolo_engine.cpp:216
        o.class_id = (d[i].cls >= 0 && d[i].cls < IRMV_NUM_CLASSES) ? d[i].cls : IRMV_NUM_CLASSES;
        o.anchor = d[i].anchor;
        o.pnp_ok = d[i].pnp_ok;
        memcpy(o.kpts, d[i].kpts, 32);
        memcpy(o.rvec, d[i].rvec, 24);
        memcpy(o.tvec, d[i].tvec, 24);
        memcpy(o.quat, d[i].quat, 32);
        o.armor_valid = d[i].armor_valid;
        o.armor_size = d[i].armor_size;
        o.n_lights = d[i].n_lights;
        o.reserved = 0;
    }
    *n = k;
    return IRMV_OK;
}

extern "C" int irmv_engine_detect(irmv_engine *e, int slot, irmv_det *out, int cap, int *n)
{
    const auto t0 = std::chrono::high_resolution_clock::now();
    // a synchronous single-slot call has nothing to overlap with: upload, graph and download ride ONE stream
    TRY(irmv_engine_submit(e, slot, 1, IRMV_SUBMIT_H2D));
    TRY(irmv_engine_wait_slots(e, slot, 1));
    const int rc = irmv_engine_results(e, slot, out, cap, n);
    e->last_detect_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    return rc;
}

extern "C" double irmv_engine_last_detect_ms(const irmv_engine *e) { return e ? e->last_detect_ms : 0.0; }

extern "C" int irmv_engine_rotated_image(irmv_engine *e, int slot, uint8_t *dst)
{
    TRY(check_range(e, slot, 1));
    if (!dst) return fail(IRMV_ERR_ARG, "dst is null");
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipMemcpyAsync(e->src_dev + (size_t)slot * e->frame_bytes, e->src_host + (size_t)slot * e->frame_bytes,
                           e->frame_bytes, hipMemcpyHostToDevice, e->stream));
    launch_rotate180(e->src_dev + (size_t)slot * e->frame_bytes, e->rot_dev, e->cfg.src_width, e->cfg.src_height, e->stream);
    HIP_TRY(hipMemcpyAsync(dst, e->rot_dev, e->frame_bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return IRMV_OK;
}

extern "C" int irmv_engine_extract_armors(irmv_engine *e, int slot, const float *xyxy, int n, irmv_det *out)
{
    TRY(check_range(e, slot, 1));
    if (n < 0 || n > e->cfg.max_det || (n > 0 && (!xyxy || !out))) return fail(IRMV_ERR_ARG, "n must be 0..max_det with xyxy/out set");
    if (n == 0) return IRMV_OK;
    HIP_TRY(hipSetDevice(e->cfg.device));
    TRY(irmv_engine_wait(e));
    hipStream_t st = e->stream;
    HIP_TRY(hipMemcpyAsync(e->src_dev + (size_t)slot * e->frame_bytes, e->src_host + (size_t)slot * e->frame_bytes, e->frame_bytes,
                           hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(e->light_boxes, xyxy, (size_t)n * 16, hipMemcpyHostToDevice, st));
    LightArgs a = light_args(e, slot);
    a.dets = e->light_dets_dev;
    a.labels = e->light_labels;
    a.points = e->light_points;
    a.hulls = e->light_hulls;
    a.num_dets = nullptr;
    a.n_boxes = n;
    a.boxes = e->light_boxes;
    launch_light_extract(a, n, 1, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(e->light_dets_host, e->light_dets_dev, (size_t)n * sizeof(DevDet), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < n; i++) {
        const DevDet &d = e->light_dets_host[i];
        irmv_det &o = out[i];
        memset(&o, 0, sizeof o);
        memcpy(o.xyxy, xyxy + 4 * i, 16);
        o.class_id = IRMV_NUM_CLASSES;
        o.pnp_ok = d.pnp_ok;
        memcpy(o.kpts, d.kpts, 32);
        memcpy(o.rvec, d.rvec, 24);
        memcpy(o.tvec, d.tvec, 24);
        memcpy(o.quat, d.quat, 32);
        o.armor_valid = d.armor_valid;
        o.armor_size = d.armor_size;
        o.n_lights = d.n_lights;
    }
    return IRMV_OK;
}

extern "C" int irmv_engine_point_source(const irmv_engine *e)
{
    if (!e) return -1;
    return e->classical ? IRMV_POINTS_CLASSICAL : IRMV_POINTS_KEYPOINT_HEAD;
}

extern "C" int irmv_engine_set_extract_params(irmv_engine *e, int binary_threshold, float light_min_ratio, float light_max_ratio,
                                              float light_max_angle, const double cd[4])
{
    if (!e || !cd) return fail(IRMV_ERR_ARG, "null argument");
    if (binary_threshold < 0 || binary_threshold > 255) return fail(IRMV_ERR_ARG, "binary_threshold must be 0..255");
    HIP_TRY(hipSetDevice(e->cfg.device));
    TRY(irmv_engine_wait(e));
    irmv_engine_cfg &c = e->cfg;
    const bool same = c.binary_threshold == binary_threshold && c.light_min_ratio == light_min_ratio && c.light_max_ratio == light_max_ratio &&
                      c.light_max_angle == light_max_angle && c.armor_min_small_center_distance == cd[0] &&
                      c.armor_max_small_center_distance == cd[1] && c.armor_min_large_center_distance == cd[2] &&
                      c.armor_max_large_center_distance == cd[3];
    if (same) return IRMV_OK;
    c.binary_threshold = binary_threshold;
    c.light_min_ratio = light_min_ratio; c.light_max_ratio = light_max_ratio; c.light_max_angle = light_max_angle;
    c.armor_min_small_center_distance = cd[0]; c.armor_max_small_center_distance = cd[1];
    c.armor_min_large_center_distance = cd[2]; c.armor_max_large_center_distance = cd[3];
    // captured steps of the classical mode carry these values as kernel arguments: re-capture on next use
    if (e->classical) {
        for (auto &g : e->graphs) (void)hipGraphExecDestroy(g.second);
        e->graphs.clear();
    }
    return IRMV_OK;
}

// ---- read-backs --------------------------------------------------------------------
static int read_tensor_f32(irmv_engine *e, const Tensor &t, int slot, std::vector<float> &out)
{
    HIP_TRY(hipSetDevice(e->cfg.device));
    TRY(irmv_engine_wait(e));
    out.resize(t.slot_elems);
    if (t.f32) {
        HIP_TRY(hipMemcpy(out.data(), t.slot(slot), t.slot_elems * 4, hipMemcpyDeviceToHost));
    } else {
        std::vector<uint16_t> h(t.slot_elems);
        HIP_TRY(hipMemcpy(h.data(), t.slot(slot), t.slot_elems * 2, hipMemcpyDeviceToHost));
        // activation tensors hold log2 e * a (irmv_common.hpp, "activation scale"); the network input does not
        const float unscale = t.name == "input" ? 1.0f : kActUnscale;
        for (size_t i = 0; i < t.slot_elems; i++) out[i] = half_bits_to_float(h[i]) * unscale;
    }
    return IRMV_OK;
}

// A step never writes the tensors inside a fused kernel ("input", "0", "model.2.cat", "model.2.tmp"): a read-back of one of
// them first runs the stand-alone layers the fused kernels cover, on the slot's current device frame.
static int materialize_fused(irmv_engine *e, int slot)
{
    if (e->lazy_tensors.empty()) return IRMV_OK;
    HIP_TRY(hipSetDevice(e->cfg.device));
    TRY(irmv_engine_wait(e));
    TRY(enqueue_step(e, slot, 1, 0x20000000u, false, nullptr));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return IRMV_OK;
}

extern "C" int irmv_engine_read_input(irmv_engine *e, int slot, float *chw)
{
    TRY(check_range(e, slot, 1));
    TRY(materialize_fused(e, slot));
    std::vector<float> v;
    TRY(read_tensor_f32(e, e->tensors[e->tensor_idx.at("input")], slot, v));
    const size_t n = (size_t)e->cfg.net_size * e->cfg.net_size;
    for (size_t p = 0; p < n; p++)
        for (int c = 0; c < 3; c++) chw[c * n + p] = v[p * 4 + c];
    return IRMV_OK;
}

extern "C" int irmv_engine_read_head(irmv_engine *e, int slot, float *head)
{
    TRY(check_range(e, slot, 1));
    for (int l = 0; l < 3; l++) {
        std::vector<float> v;
        TRY(read_tensor_f32(e, e->tensors[e->head_t[l]], slot, v));
        for (int p = 0; p < e->lvl_hw[l]; p++) {
            float *o = head + (size_t)(e->lvl_base[l] + p) * e->no;
            const float *r = v.data() + (size_t)p * kHeadRec;
            memcpy(o, r, 64 * 4);
            memcpy(o + 64, r + kClsOff, (size_t)e->nc * 4);
            if (e->nk) memcpy(o + 64 + e->nc, r + kKptOff, (size_t)e->nk * 4);
        }
    }
    return IRMV_OK;
}

extern "C" int irmv_engine_write_head(irmv_engine *e, int slot, const float *head)
{
    TRY(check_range(e, slot, 1));
    HIP_TRY(hipSetDevice(e->cfg.device));
    TRY(irmv_engine_wait(e));
    for (int l = 0; l < 3; l++) {
        std::vector<float> v((size_t)e->lvl_hw[l] * kHeadRec, 0.f);
        for (int p = 0; p < e->lvl_hw[l]; p++) {
            const float *o = head + (size_t)(e->lvl_base[l] + p) * e->no;
            float *r = v.data() + (size_t)p * kHeadRec;
            memcpy(r, o, 64 * 4);
            memcpy(r + kClsOff, o + 64, (size_t)e->nc * 4);
            if (e->nk) memcpy(r + kKptOff, o + 64 + e->nc, (size_t)e->nk * 4);
        }
        HIP_TRY(hipMemcpy(e->tensors[e->head_t[l]].slot(slot), v.data(), v.size() * 4, hipMemcpyHostToDevice));
    }
    return IRMV_OK;
}

extern "C" int irmv_engine_read_tap(irmv_engine *e, int slot, const char *name, float *nhwc, int shape[3])
{
    TRY(check_range(e, slot, 1));
    if (!name || !shape) return fail(IRMV_ERR_ARG, "name/shape is null");
    auto it = e->tensor_idx.find(name);
    if (it == e->tensor_idx.end()) return fail(IRMV_ERR_ARG, std::string("no tensor named ") + name);
    const Tensor &t = e->tensors[it->second];
    shape[0] = t.H; shape[1] = t.W; shape[2] = t.C;
    if (!nhwc) return IRMV_OK;
    if (e->lazy_tensors.count(t.name)) TRY(materialize_fused(e, slot));
    std::vector<float> v;
    TRY(read_tensor_f32(e, t, slot, v));
    memcpy(nhwc, v.data(), v.size() * 4);
    return IRMV_OK;
}

extern "C" int irmv_engine_read_raw(irmv_engine *e, int slot, irmv_raw_dets *out)
{
    TRY(check_range(e, slot, 1));
    if (!out) return fail(IRMV_ERR_ARG, "out is null");
    const DevFrameOut &fo = e->fout_host[slot];
    const DevDet *d = e->dets_host + (size_t)slot * e->cfg.max_det;
    out->num_dets = fo.num_dets;
    out->n_candidates = fo.n_candidates;
    for (int i = 0; i < e->cfg.max_det; i++) {
        if (out->det_boxes) memcpy(out->det_boxes + 4 * i, d[i].box_net, 16);
        if (out->det_scores) out->det_scores[i] = d[i].score;
        if (out->det_classes) out->det_classes[i] = d[i].cls;
        if (out->det_anchors) out->det_anchors[i] = d[i].anchor;
        if (out->det_kpts) memcpy(out->det_kpts + 8 * i, d[i].kpts_net, 32);
    }
    return IRMV_OK;
}

extern "C" int irmv_engine_profile(irmv_engine *e, int first, int count, irmv_kernel_stat *stats, int cap, int *n)
{
    TRY(check_range(e, first, count));
    if (!n) return fail(IRMV_ERR_ARG, "n is null");
    HIP_TRY(hipSetDevice(e->cfg.device));
    // Eager replay of the step's launches on the engine stream, every kernel bracketed by an event
    // pair.  The kernels are idempotent and are launched kRep times inside their bracket: an event pair around ONE launch also times ~4 us of
    // command-processor hand-over, which would read as kernel time on these 5-80 us kernels.
    // (Event-record nodes inside a captured graph cannot be read back with hipEventElapsedTime on
    // ROCm 7.2: "invalid resource handle".)
    TRY(irmv_engine_wait(e));
    std::vector<EvRec> ev;
    TRY(enqueue_step(e, first, count, kProfileRepeat, false, &ev));
    TRY(copy_out(e, first, count));
    HIP_TRY(hipStreamSynchronize(e->stream));
    int k = 0;
    for (size_t i = 0; i < ev.size(); i++) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev[i].a, ev[i].b));
        (void)hipEventDestroy(ev[i].a);
        (void)hipEventDestroy(ev[i].b);
        const Op &op = e->ops[ev[i].op];
        if (!(op.kind == OP_LIGHT || (e->split_scan && (op.kind == OP_SCAN || op.kind == OP_NMS)))) ms /= (float)kProfileRepeat;
        if (k < cap && stats) {
            irmv_kernel_stat &st = stats[k];
            memset(&st, 0, sizeof st);
            if (op.kind == OP_CONV && op.group >= 0 && count == 1) {   // one launch for the whole group
                const irmv_engine::HeadGroup &g = e->head_groups[op.group];
                snprintf(st.name, sizeof st.name, "%s", g.name);
                snprintf(st.layer, sizeof st.layer, "%s ... (%zu convs)", op.layer.c_str(), g.members.size());
                for (int mi : g.members) {
                    const Op &mo = e->ops[mi];
                    st.flops += mo.flops + (mo.fuse_next >= 0 ? e->ops[mo.fuse_next].flops : 0.0);
                    st.bytes += mo.bytes;
                }
            } else {
            snprintf(st.name, sizeof st.name, "%s", (count == 1 && stream_share(e, e->cfg.num_slots) > 1 && op.kind == OP_CONV) ? op.kname_one : op.kname);
            snprintf(st.layer, sizeof st.layer, "%s", op.layer.c_str());
            st.flops = (op.flops + (op.fuse_next >= 0 ? e->ops[op.fuse_next].flops : 0.0)) * count;
            double b = op.bytes, w = op.w_bytes;
            if (op.fuse_next >= 0) {   // the fused 1x1's output is what reaches memory, its weights ride along
                const Op &nx = e->ops[op.fuse_next];
                b += nx.out_bytes - op.out_bytes + nx.w_bytes;
                w += nx.w_bytes;
            }
            st.bytes = (b - w) * count + w;   // activations per frame, weights once per launch
            }
            st.ms = ms;
        }
        k++;
    }
    *n = k;
    return IRMV_OK;
}

// ---- PnPSolver -------------------------------------------------------------------------
struct irmv_pnp {
    int device = 0;
    PnpConst c{};
    hipStream_t stream = nullptr;
    float *pts = nullptr;
    double *rvec = nullptr, *tvec = nullptr;
    int32_t *ok = nullptr;
    int cap = 0;
};

static void pnp_free(irmv_pnp *p)
{
    if (p->pts) (void)hipFree(p->pts);
    if (p->rvec) (void)hipFree(p->rvec);
    if (p->tvec) (void)hipFree(p->tvec);
    if (p->ok) (void)hipFree(p->ok);
    p->pts = nullptr; p->rvec = p->tvec = nullptr; p->ok = nullptr; p->cap = 0;
}

extern "C" int irmv_pnp_create(int device, const double K[9], const double D[5], irmv_pnp **out)
{
    if (!K || !D || !out) return fail(IRMV_ERR_ARG, "null argument");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(IRMV_ERR_HIP, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<irmv_pnp> p(new irmv_pnp);
    p->device = device;
    p->c.fx = K[0]; p->c.fy = K[4]; p->c.cx = K[2]; p->c.cy = K[5];
    p->c.k1 = D[0]; p->c.k2 = D[1]; p->c.p1 = D[2]; p->c.p2 = D[3]; p->c.k3 = D[4];
    p->c.hy[0] = 135.0 / 2.0 / 1000.0; p->c.hy[1] = 225.0 / 2.0 / 1000.0;
    p->c.hz[0] = p->c.hz[1] = 55.0 / 2.0 / 1000.0;
    HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    *out = p.release();
    return IRMV_OK;
}

extern "C" void irmv_pnp_destroy(irmv_pnp *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    pnp_free(p);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

extern "C" int irmv_pnp_solve(irmv_pnp *p, const float *img_pts, int n, int armor_size, double *rvec, double *tvec, int32_t *ok)
{
    if (!p || !img_pts || !rvec || !tvec || !ok || n < 0) return fail(IRMV_ERR_ARG, "null argument");
    if (armor_size != IRMV_ARMOR_SMALL && armor_size != IRMV_ARMOR_LARGE) return fail(IRMV_ERR_ARG, "bad armor_size");
    if (n == 0) return IRMV_OK;
    HIP_TRY(hipSetDevice(p->device));
    if (n > p->cap) {
        pnp_free(p);
        const int cap = std::max(n, 64);
        HIP_TRY(hipMalloc((void **)&p->pts, (size_t)cap * 32));
        HIP_TRY(hipMalloc((void **)&p->rvec, (size_t)cap * 24));
        HIP_TRY(hipMalloc((void **)&p->tvec, (size_t)cap * 24));
        HIP_TRY(hipMalloc((void **)&p->ok, (size_t)cap * 4));
        p->cap = cap;
    }
    HIP_TRY(hipMemcpyAsync(p->pts, img_pts, (size_t)n * 32, hipMemcpyHostToDevice, p->stream));
    launch_pnp_only(p->c, p->pts, n, armor_size, p->rvec, p->tvec, p->ok, p->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rvec, p->rvec, (size_t)n * 24, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(tvec, p->tvec, (size_t)n * 24, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(ok, p->ok, (size_t)n * 4, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return IRMV_OK;
}
