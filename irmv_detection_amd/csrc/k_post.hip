// Detection post-processing on the GPU: DFL box decode + score filter, score
// sort, class-aware greedy NMS, keypoint decode, parse_output scaling and planar
// (IPPE) PnP -- everything the reference does after the conv body:
//   EfficientNMS_TRT plugin   (bound at src/yolo_engine.cpp:53-57,82-85)
//   YoloEngine::parse_output  (src/yolo_engine.cpp:202-220)
//   PnPSolver::solvePnP       (src/pnp_solver.cpp:36-52, cv::SOLVEPNP_IPPE)
//   Rodrigues -> quaternion   (src/irm_detector.cpp:218-226)
//
// THIS FILE IS COMPILED WITH -ffp-contract=off: the fp32 decode / IoU
// expressions are evaluated exactly as written (fmaf only where spelled out), so
// candidate sets, sort order and NMS survivors are bit-reproducible against the
// CPU oracle on identical head tensors.
//
// Latency-bound integer/compare work: one lane per anchor for the decode, one
// workgroup per frame for sort + NMS, the IoU test of a 64-candidate block
// against the kept set and against itself done wave-wide with ballots.
#include "irmv_common.hpp"
#include "pnp_device.hpp"

#include <mutex>

namespace irmv {

// exp(x) as a fixed sequence of fp32 operations (same sequence in oracle/orc_post.c)
__device__ __forceinline__ float irmv_expf(float x)
{
    x = x < -87.0f ? -87.0f : x;
    x = x > 88.0f ? 88.0f : x;
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 720.0f;
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    return __int_as_float(__float_as_int(p) + ((int)n << 23));
}

__device__ __forceinline__ uint32_t orderable(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorderable(uint32_t u)
{
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

// Anchor of a candidate key's id (= anchor * nc + class), CLAMPED to the head: keys are only ever written with valid ids, but
// a key list read to a count that held garbage hands stale bytes to every later phase, and none of
// them may turn that into an access outside a buffer.
__device__ __forceinline__ int anchor_of(uint32_t id, int nc, int A)
{
    const uint32_t an = id / (uint32_t)nc;
    return an < (uint32_t)A ? (int)an : A - 1;
}

// anchor -> grid position, stride, and the (level base, level size, index in level) that locate its head record
__device__ __forceinline__ void anchor_geom(int a, int net, int &ix, int &iy, int &stride, int &lbase, int &lhw, int &rin)
{
    int base = 0;
    ix = iy = 0;
    stride = 0;
    lbase = 0;
    lhw = 1;
    rin = 0;
#pragma unroll
    for (int l = 0; l < 3; l++) {
        const int s = 8 << l, w = net / s, cnt = w * w;
        if (stride == 0 && a < base + cnt) {
            const int r = a - base;
            iy = r / w;
            ix = r - iy * w;
            stride = s;
            lbase = base;
            lhw = cnt;
            rin = r;
        }
        base += cnt;
    }
}

__device__ __forceinline__ const float *head_rec(const float *head_all, int slots_total, int slot, int lbase, int lhw, int rin)
{
    return head_all + ((size_t)lbase * slots_total + (size_t)slot * lhw + rin) * kHeadRec;
}

__device__ __forceinline__ float dfl_side(const float *l)
{
    float m = l[0];
#pragma unroll
    for (int j = 1; j < 16; j++) m = l[j] > m ? l[j] : m;
    float se = 0.0f, sj = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const float e = irmv_expf(l[j] - m);
        se = se + e;
        sj = sj + e * (float)j;
    }
    return sj / se;
}

// Decode phase of nms_pnp_kernel, part 1 (scan): one quad of lanes per anchor, lane q owns classes 4q..4q+3 (one 16-byte
// read of the record's class logits).  Candidate keys go to the workgroup's LDS list (and to the frame's global list,
// which only the > kCandCap path reads back); anchors with at least one candidate are appended to `alist`.  List
// positions come from LDS counters: no global atomic, no counter that outlives the kernel.
__device__ __forceinline__ void scan_quad(const PostArgs &a, int an, bool live, const f32x4 cl, unsigned long long *skeys, unsigned long long *gk,
                                          int *s_ncand, unsigned short *alist, int *s_nanch)
{
    const int q = threadIdx.x & 3;
    bool hit = false;
#pragma unroll
    for (int i = 0; i < 4; i++) hit = hit || (live && 4 * q + i < a.nc && cl[i] > a.logit_thr);
    if (!__any(hit)) return;                      // the common case: nothing in these 16 anchors
    const int lane = threadIdx.x & 63, base = lane & ~3;
    const bool quad_hit = ((__ballot(hit) >> base) & 0xfull) != 0ull;
    if (alist && quad_hit && q == 0) alist[atomicAdd(s_nanch, 1)] = (unsigned short)an;
    if (!hit) return;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = 4 * q + i;
        const float logit = cl[i];
        if (c < a.nc && logit > a.logit_thr) {
            const int idx = atomicAdd(s_ncand, 1);   // LDS; <= A * nc = key_cap by construction
            const unsigned long long key = ((unsigned long long)orderable(logit) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)(an * a.nc + c));
            if (idx < kCandCap) skeys[idx] = key;
            gk[idx] = key;
        }
    }
}

// part 2: the box of every anchor on `alist`, four lanes per anchor, lane q decoding side q (16 DFL logits, one 64-byte
// read; exactly dfl_side()).  Boxes exist only where the NMS walk can read them: 5/6 of the head's bytes are never touched.
__device__ __forceinline__ void decode_boxes(const PostArgs &a, int b, const unsigned short *alist, int n_anch)
{
    const int q = threadIdx.x & 3, lane = threadIdx.x & 63, base = lane & ~3;
    for (int i0 = 0; i0 < n_anch; i0 += 256) {
        const int ai = i0 + (threadIdx.x >> 2);
        const bool live = ai < n_anch;                 // quad-uniform
        const int an = live ? (alist ? (int)alist[ai] : ai) : 0;   // no list (more anchors than it can index): every anchor
        int ix, iy, s, lbase, lhw, rin;
        anchor_geom(an, a.net, ix, iy, s, lbase, lhw, rin);
        const float *rec = head_rec(a.head_all, a.slots_total, a.first + b, lbase, lhw, rin);
        float l[16];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const f32x4 v = reinterpret_cast<const f32x4 *>(rec + 16 * q)[i];
            l[4 * i] = v[0]; l[4 * i + 1] = v[1]; l[4 * i + 2] = v[2]; l[4 * i + 3] = v[3];
        }
        const float d = dfl_side(l);
        const float dl = __shfl(d, base), dt = __shfl(d, base + 1), dr = __shfl(d, base + 2), db = __shfl(d, base + 3);
        if (live && q == 0) {
            const float ax = (float)ix + 0.5f, ay = (float)iy + 0.5f, sf = (float)s;
            f32x4 box;
            box[0] = (ax - dl) * sf;
            box[1] = (ay - dt) * sf;
            box[2] = (ax + dr) * sf;
            box[3] = (ay + db) * sf;
            reinterpret_cast<f32x4 *>(a.boxes)[(size_t)b * a.A + an] = box;
        }
    }
}

// ---------------------------------------------------------------------------
// Scan + box decode as a kernel of its own, kScanBlocks workgroups per frame.  One workgroup reading a frame's class
// logits (8400 x 64 B, strided through 384-byte records) is bound by what ONE CU can pull from memory (~30 GB/s): 18 of
// the 53 us of a single-frame nms_pnp launch, 25 - 30 of 98 us with 64 frames on 64 CUs.  Spread over 16 CUs per frame
// the same bytes take ~2 us.  Keys go to the frame's global list in whatever order the workgroups append them (the
// sort of nms_pnp_kernel makes the order irrelevant: keys are unique); the box of every anchor with a candidate is
// decoded here, four lanes per anchor, exactly as decode_boxes does.
// ---------------------------------------------------------------------------
constexpr int kScanU = 2;   // loads in flight per lane: one memory round trip per U * 64 anchors
static int scan_quads_per_block(int A) { return ((A * 4 + kScanBlocks - 1) / kScanBlocks + 256 * kScanU - 1) / (256 * kScanU) * (256 * kScanU); }

// The workgroup's keys are collected in LDS (sized for every (anchor, class) pair of its range: it cannot overflow) and
// leave with ONE global atomic per workgroup: one atomic per candidate on the frame's counter serialises in L2 (measured:
// ~40 ns each, 15 us for 380 candidates -- as long as the scan this kernel was split off to shorten).
__global__ __launch_bounds__(256) void scan_decode_kernel(PostArgs a, int per)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];
    __shared__ int s_n, s_base;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, q = tid & 3, base = lane & ~3;
    const int quads = a.A * 4;
    constexpr int U = kScanU;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int w0 = a.net >> 3, A0 = w0 * w0, A1 = A0 + (w0 >> 1) * (w0 >> 1);   // level boundaries
    unsigned long long *gk = a.keys + (size_t)b * a.key_cap;
    const int t_end = min((int)(blockIdx.x + 1) * per, quads);
    for (int t0 = blockIdx.x * per; t0 < t_end; t0 += 256 * U) {
        f32x4 cl[U];
        const float *rec[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t0 + u * 256 + tid;
            const int an = t < quads ? (t >> 2) : 0;
            const int lbase = an < A0 ? 0 : (an < A1 ? A0 : A1), lhw = an < A0 ? A0 : (an < A1 ? A1 - A0 : a.A - A1);
            rec[u] = head_rec(a.head_all, a.slots_total, a.first + b, lbase, lhw, an - lbase);
            cl[u] = reinterpret_cast<const f32x4 *>(rec[u] + kClsOff)[q];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t0 + u * 256 + tid;
            const bool live = t < quads;
            const int an = live ? (t >> 2) : 0;
            bool hit = false;
#pragma unroll
            for (int i = 0; i < 4; i++) hit = hit || (live && 4 * q + i < a.nc && cl[u][i] > a.logit_thr);
            const unsigned long long hits = __ballot(hit);
            if (hits == 0ull) continue;                            // the common case: nothing in these 16 anchors
            if (hit) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int c = 4 * q + i;
                    if (c < a.nc && cl[u][i] > a.logit_thr)
                        s_keys[atomicAdd(&s_n, 1)] = ((unsigned long long)orderable(cl[u][i]) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)(an * a.nc + c));
                }
            }
            if (((hits >> base) & 0xfull) != 0ull) {               // this quad's anchor has a candidate: its box, side q on lane q
                float l[16];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const f32x4 v = reinterpret_cast<const f32x4 *>(rec[u] + 16 * q)[i];
                    l[4 * i] = v[0]; l[4 * i + 1] = v[1]; l[4 * i + 2] = v[2]; l[4 * i + 3] = v[3];
                }
                const float d = dfl_side(l);
                // (shuffles inside a divergent region: the four lanes of a quad are active together by construction)
                const float dl = __shfl(d, base), dt = __shfl(d, base + 1), dr = __shfl(d, base + 2), db = __shfl(d, base + 3);
                if (q == 0) {
                    int ix, iy, s, lbase, lhw, rin;
                    anchor_geom(an, a.net, ix, iy, s, lbase, lhw, rin);
                    const float ax = (float)ix + 0.5f, ay = (float)iy + 0.5f, sf = (float)s;
                    f32x4 box;
                    box[0] = (ax - dl) * sf;
                    box[1] = (ay - dt) * sf;
                    box[2] = (ax + dr) * sf;
                    box[3] = (ay + db) * sf;
                    reinterpret_cast<f32x4 *>(a.boxes)[(size_t)b * a.A + an] = box;
                }
            }
        }
    }
    __syncthreads();
    const int n = s_n;
    if (n == 0) return;
    if (tid == 0) s_base = atomicAdd(&a.counts[b], n);
    __syncthreads();
    const int gb = s_base;
    for (int i = tid; i < n; i += 256) {
        const int idx = gb + i;
        if (idx >= 0 && idx < a.key_cap) gk[idx] = s_keys[i];     // (a counter that holds garbage cannot push a write outside the list)
    }
}

void launch_scan_decode(const PostArgs &a, int batch, hipStream_t s)
{
    const int per = scan_quads_per_block(a.A);
    const size_t lds = (size_t)(per / 4) * a.nc * sizeof(unsigned long long);
    static std::mutex mu;
    static size_t raised[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(mu);
        if (raised[dev & 63] < lds) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(scan_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            raised[dev & 63] = lds;
        }
    }
    hipLaunchKernelGGL(scan_decode_kernel, dim3(kScanBlocks, batch), dim3(256), lds, s, a, per);
}

// "IoU(a, b) > thr" as inter > thr * union (same expression as the oracle's iou_gt)
__device__ __forceinline__ bool iou_gt(const f32x4 a, const f32x4 b, float thr)
{
    const float ix1 = a[0] > b[0] ? a[0] : b[0];
    const float iy1 = a[1] > b[1] ? a[1] : b[1];
    const float ix2 = a[2] < b[2] ? a[2] : b[2];
    const float iy2 = a[3] < b[3] ? a[3] : b[3];
    float iw = ix2 - ix1, ih = iy2 - iy1;
    iw = iw > 0.0f ? iw : 0.0f;
    ih = ih > 0.0f ? ih : 0.0f;
    const float inter = iw * ih;
    const float aa = (a[2] - a[0]) * (a[3] - a[1]);
    const float ab = (b[2] - b[0]) * (b[3] - b[1]);
    const float uni = (aa + ab) - inter;
    return inter > thr * uni;
}

// the same test with the two areas handed in ((a[2] - a[0]) * (a[3] - a[1]), computed once per box): same operations, same bits
__device__ __forceinline__ bool iou_gt_area(const f32x4 a, float aa, const f32x4 b, float ab, float thr)
{
    const float ix1 = a[0] > b[0] ? a[0] : b[0];
    const float iy1 = a[1] > b[1] ? a[1] : b[1];
    const float ix2 = a[2] < b[2] ? a[2] : b[2];
    const float iy2 = a[3] < b[3] ? a[3] : b[3];
    float iw = ix2 - ix1, ih = iy2 - iy1;
    iw = iw > 0.0f ? iw : 0.0f;
    ih = ih > 0.0f ? ih : 0.0f;
    const float inter = iw * ih;
    const float uni = (aa + ab) - inter;
    return inter > thr * uni;
}

__global__ void pnp_only_kernel(PnpConst c, const float *pts, int n, int armor_size, double *rvec, double *tvec, int32_t *ok)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r[3] = {0, 0, 0}, t[3] = {0, 0, 0};
    const bool good = solve_pnp_ippe(c, pts + 8 * i, armor_size, r, t, nullptr);
    for (int k = 0; k < 3; k++) { rvec[3 * i + k] = r[k]; tvec[3 * i + k] = t[k]; }
    ok[i] = good ? 1 : 0;
}

void launch_pnp_only(const PnpConst &c, const float *pts, int n, int armor_size, double *rvec, double *tvec, int32_t *ok, hipStream_t s)
{
    hipLaunchKernelGGL(pnp_only_kernel, dim3((n + 63) / 64), dim3(64), 0, s, c, pts, n, armor_size, rvec, tvec, ok);
}

// ---------------------------------------------------------------------------
// One workgroup (1024 lanes = 16 waves) per frame:
//   1. sort the candidate keys in LDS (rank sort up to 2048 keys: one pass, no
//      barrier ladder; bitonic above that);
//   2. all 16 waves, one 64-candidate block each: the block's 64x64 same-class
//      IoU > thr relation as one 64-bit "who suppresses me" mask per candidate
//      (independent of the kept set, so it is computed up front, in parallel);
//   3. wave 0 walks the blocks in score order: test against the kept boxes OF THE
//      CANDIDATE'S CLASS (per-class kept lists), resolve the block with ballots,
//      append survivors;
//   4. one lane per survivor: keypoints, parse_output scaling, fp64 IPPE PnP.
// The greedy walk is exactly the oracle's: same comparisons, same order.
// ---------------------------------------------------------------------------
constexpr int kRankSortMax = 2048;   // capacity of the rank-sort destination
constexpr int kRankSortUse = 512;    // above this the O(n^2) rank sort loses to the bitonic network
constexpr int kSupCap = 4096;   // candidates whose intra-block masks are precomputed
constexpr int kLazyN_ = 1024;   // most candidates of the lazy-matrix walk
// A crowded frame's first attempt works on the best kPreLo .. kPreHi candidates: <= 512 = the full-matrix walk, the fastest
// path there is (on the benchmark's frames the hundredth survivor sits among the first ~ 300 candidates)
constexpr int kPreHi = 512, kPreLo = 320;
constexpr int kMatW_ = 8;       // 64-bit words of a suppression-matrix row (512 candidates)
constexpr int kKptLds = 6144;   // class-walk path: the candidates' keypoint logits live in skeys[kKptLds ..) (16 KiB)

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// "who of the 64 candidates of my block suppresses me": bit j set iff j < lane, same class, IoU > thr.
// sbox / scls: this wave's private LDS staging of the block.
__device__ __forceinline__ unsigned long long block_sup_mask(f32x4 box, int cls, int lane, float iou_thr, f32x4 *sbox, int *scls)
{
    sbox[lane] = box;
    scls[lane] = cls;
    wave_lds_sync();
    unsigned long long sup = 0ull;
#pragma unroll 8
    for (int j = 0; j < 64; j++) {
        const f32x4 bj = sbox[j];
        const int cj = scls[j];
        if (j < lane && cj == cls && cls >= 0 && iou_gt(bj, box, iou_thr)) sup |= 1ull << j;
    }
    wave_lds_sync();
    return sup;
}

__global__ __launch_bounds__(1024) void nms_pnp_kernel(PostArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned long long skeys[kCandCap];   // 64 KiB: candidate keys (bitonic sorts in place)
    __shared__ unsigned long long srank[kRankSortMax];      // 16 KiB: rank-sort destination
    __shared__ __attribute__((aligned(16))) unsigned long long ssup[kSupCap];   // 32 KiB: suppression masks (and, before them, the anchor list)
    __shared__ f32x4 stage_box[16][64];                     // 16 KiB: per-wave block staging
    __shared__ __attribute__((aligned(16))) int stage_cls[16][64];   // (16-byte aligned: the matrix phase reads it as int4)
    __shared__ f32x4 kept_box[kMaxDetCap];
    __shared__ int kept_cls[kMaxDetCap];
    __shared__ unsigned long long kept_key[kMaxDetCap];
    __shared__ unsigned short cls_list[16][kMaxDetCap];     // per class: indices into kept_*
    __shared__ int cls_cnt[16];
    __shared__ int s_kept;

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#define IRMV_STAMP(k) do { if (a.dbg && tid == 0) a.dbg[b * 16 + (k)] = clock64(); } while (0)
    IRMV_STAMP(0);
    __shared__ int s_ncand, s_nanch;
    unsigned long long *gk = a.keys + (size_t)b * a.key_cap;
    const unsigned long long *sorted;
    // anchors with a candidate, as 16-bit indices in ssup (free until phase 2): up to 16384 anchors (a 640 net has 8400);
    // larger nets decode every anchor's box instead
    unsigned short *alist = a.A <= 4 * kSupCap ? reinterpret_cast<unsigned short *>(ssup) : nullptr;
    if (tid < 16) cls_cnt[tid] = 0;
    if (tid == 0) { s_ncand = 0; s_nanch = 0; }
    __syncthreads();
    if (a.counts) {
        // ---- 0'. scan_decode_kernel has filled the frame's key list and decoded the candidate anchors' boxes ----
        // (a lane's first key is requested together with the count, not behind it: one memory round trip instead of two
        // for frames of up to 1024 candidates; beyond the count it is whatever the list holds, and is not used)
        // (round 4: ALL eight keys a lane can own in the LDS list are requested with the count -- a crowded frame's 2 000 .. 5 000
        // keys used to arrive in dependent round trips of 1024, one per loop iteration, ~ 1.5 us each on a single frame)
        constexpr int KPL = kCandCap / 1024;
        unsigned long long k_mine[KPL];
#pragma unroll
        for (int k = 0; k < KPL; k++) {
            const int i = tid + k * 1024;
            k_mine[k] = gk[i < a.key_cap ? i : a.key_cap - 1];
        }
        int n = a.counts[b];
        n = n < 0 ? 0 : (n > a.key_cap ? a.key_cap : n);           // whatever the counter holds, reads stay inside the list
        if (n <= kCandCap) {
#pragma unroll
            for (int k = 0; k < KPL; k++)
                if (tid + k * 1024 < n) skeys[tid + k * 1024] = k_mine[k];
        }
        __syncthreads();                                           // every lane has read the count
        if (tid == 0) { s_ncand = n; a.counts[b] = 0; }            // the next step of this slot starts from zero
    } else {
    // ---- 0. decode.  Scan: class logits of every anchor -> candidate keys + the list of anchors that have one.  Level
    // by level (records of a level are contiguous), four lanes per anchor; the loads of U rounds are issued together.
    {
        constexpr int U = 8;
        int lbase = 0;
#pragma unroll 1
        for (int l = 0; l < 3; l++) {
            const int lw = a.net / (8 << l), lhw = lw * lw;
            const float *lrec = a.head_all + ((size_t)lbase * a.slots_total + (size_t)(a.first + b) * lhw) * kHeadRec;
            const int quads = lhw * 4;
            for (int q0 = 0; q0 < quads; q0 += 1024 * U) {
                f32x4 cl[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int t = q0 + u * 1024 + tid;
                    const int rin = t < quads ? (t >> 2) : 0;
                    cl[u] = reinterpret_cast<const f32x4 *>(lrec + (size_t)rin * kHeadRec + kClsOff)[tid & 3];
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (q0 + u * 1024 >= quads) break;          // workgroup-uniform
                    const int t = q0 + u * 1024 + tid;
                    scan_quad(a, lbase + (t >> 2), t < quads, cl[u], skeys, gk, &s_ncand, alist, &s_nanch);
                }
            }
            lbase += lhw;
        }
    }
    __syncthreads();
    decode_boxes(a, b, alist, alist ? s_nanch : a.A);   // boxes of the candidate anchors, in parallel over the whole workgroup
    }
    __syncthreads();   // keys in LDS / global and boxes in global are visible to the whole workgroup from here
    IRMV_STAMP(7);
    const int n_total = s_ncand;
    __shared__ unsigned int hist[256];
    __shared__ int s_digit, s_rem, s_fill;
    // Boxes of the candidates whose keys sit in `keys[0 .. cnt)`, four lanes per candidate (keys from the class-branch conv
    // epilogues: nobody has decoded them yet; an anchor with several classes above threshold is decoded once per class:
    // same value, same address).  Two rounds of 256 candidates per trip, both rounds' DFL logits requested before either is
    // used (a store to the box list between them would otherwise order the second round's loads behind the first round's
    // arithmetic).
    auto decode_keys = [&](const unsigned long long *keys, int cnt) {
        const int q = tid & 3, base = lane & ~3;
        for (int i0 = 0; i0 < cnt; i0 += 512) {
            f32x4 v[2][4];
            int an_[2];
            bool live_[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int ci = i0 + u * 256 + (tid >> 2);
                live_[u] = ci < cnt;                               // quad-uniform
                const unsigned long long key = live_[u] ? keys[ci] : 0ull;
                const uint32_t id = 0xffffffffu - (uint32_t)(key & 0xffffffffu);
                an_[u] = live_[u] ? anchor_of(id, a.nc, a.A) : 0;
                int ix, iy, st, lbase, lhw, rin;
                anchor_geom(an_[u], a.net, ix, iy, st, lbase, lhw, rin);
                const float *rec = head_rec(a.head_all, a.slots_total, a.first + b, lbase, lhw, rin);
#pragma unroll
                for (int i = 0; i < 4; i++) v[u][i] = reinterpret_cast<const f32x4 *>(rec + 16 * q)[i];
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (i0 + u * 256 >= cnt) break;                    // workgroup-uniform
                float l[16];
#pragma unroll
                for (int i = 0; i < 4; i++) { l[4 * i] = v[u][i][0]; l[4 * i + 1] = v[u][i][1]; l[4 * i + 2] = v[u][i][2]; l[4 * i + 3] = v[u][i][3]; }
                const float d = dfl_side(l);
                const float dl = __shfl(d, base), dt = __shfl(d, base + 1), dr = __shfl(d, base + 2), db = __shfl(d, base + 3);
                if (live_[u] && q == 0) {
                    int ix, iy, st, lbase, lhw, rin;
                    anchor_geom(an_[u], a.net, ix, iy, st, lbase, lhw, rin);
                    const float ax = (float)ix + 0.5f, ay = (float)iy + 0.5f, sf = (float)st;
                    f32x4 box;
                    box[0] = (ax - dl) * sf;
                    box[1] = (ay - dt) * sf;
                    box[2] = (ax + dr) * sf;
                    box[3] = (ay + db) * sf;
                    reinterpret_cast<f32x4 *>(a.boxes)[(size_t)b * a.A + an_[u]] = box;
                }
            }
        }
    };
    // The greedy walk visits candidates in score order and stops at max_det survivors, so on a crowded frame only the head
    // of the sorted list is ever looked at.  ATTEMPT 0 therefore works on the frame's best <= kPreHi candidates only (an
    // exact radix threshold: every key >= T, and nothing else): select, decode THEIR boxes, sort, walk.  If that walk
    // fills max_det -- or has seen everything the full algorithm would (pre_nms_cap <= the selected count) -- it IS the
    // full algorithm's result: a candidate's fate depends only on the candidates before it.  Otherwise (rare: few
    // survivors among the first thousand candidates) ATTEMPT 1 runs the whole list, as before.  A frame of 4 900 candidates
    // decodes, sorts and masks a fifth of them: the NMS kernel's 0.24 ms on such frames was all tail.
    const int pre_hi = a.prefilter > 1 ? (a.prefilter & 0xffff) : kPreHi, pre_lo = a.prefilter > 1 ? (a.prefilter >> 16) : kPreLo;   // (> 1: experiment override, hi | lo << 16)
    const bool prefilter_on = a.prefilter != 0 && n_total > pre_hi;
    int n_stored = n_total, n = 0;
    constexpr int kLazyW = kLazyN_ / 64;
    __shared__ unsigned long long s_keptw[kLazyW];        // per 64-candidate block: its survivors
    __shared__ unsigned long long s_clsmask[16][kLazyW];  // per class and block: which candidates have that class
    bool kp_lds = false;                                  // the survivors' keypoint logits sit in LDS (class-walk path)
    __shared__ int s_cbase[16], s_ccnt[16], s_W;
    for (int attempt = prefilter_on ? 0 : 1; attempt < 2; attempt++) {
    n_stored = n_total;
    if (attempt == 1 && prefilter_on) {            // start over on the whole list
        if (tid < 16) cls_cnt[tid] = 0;
        if (n_total <= kCandCap)
            for (int i = tid; i < n_total; i += blockDim.x) skeys[i] = gk[i];
        __syncthreads();
    }
    if (n_total > kCandCap) {
        // More candidates than the LDS sort holds (noise frames): keep exactly the K = pre_nms_cap largest
        // keys.  Keys are unique, so an 8-pass MSB-first radix select finds the K-th largest key T exactly;
        // every key >= T is then compacted into LDS.  Same result as sorting everything and cutting at K.
        const int K = a.pre_nms_cap < kCandCap ? a.pre_nms_cap : kCandCap;
        unsigned long long prefix = 0ull, mask = 0ull;
        if (tid == 0) { s_rem = K; s_fill = 0; }
        for (int pass = 0; pass < 8; pass++) {
            const int shift = 56 - 8 * pass;
            if (tid < 256) hist[tid] = 0u;
            __syncthreads();
            for (int i = tid; i < n_total; i += blockDim.x) {
                const unsigned long long key = gk[i];
                if ((key & mask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int rem = s_rem, d = 255;
                for (; d > 0; d--) {
                    if ((int)hist[d] >= rem) break;
                    rem -= (int)hist[d];
                }
                s_digit = d;
                s_rem = rem;
            }
            __syncthreads();
            prefix |= (unsigned long long)s_digit << shift;
            mask |= 0xffull << shift;
        }
        for (int i = tid; i < n_total; i += blockDim.x) {
            const unsigned long long key = gk[i];
            if (key >= prefix) {
                const int pos = atomicAdd(&s_fill, 1);
                if (pos < kCandCap) skeys[pos] = key;
            }
        }
        __syncthreads();
        n_stored = s_fill < kCandCap ? s_fill : kCandCap;
    }
    const int n_full = n_stored < a.pre_nms_cap ? n_stored : a.pre_nms_cap;   // candidates the full algorithm walks
    bool truncated = false;
    if (attempt == 0) {
        // ---- the frame's best lo .. hi candidates: a radix threshold T with lo <= #{key >= T} <= hi.  MSB first, 8 bits
        // per pass; a pass ends the search as soon as the buckets above the one that would overflow `hi` already hold `lo`
        // keys (usually the first or second pass: any count in [lo, hi] will do, unlike the exact K-th key above) ----
        const int hi = pre_hi, lo = pre_lo;
        unsigned long long prefix = 0ull, mask = 0ull, T = 0ull;
        int above = 0, m = 0;
        for (int pass = 0; pass < 8; pass++) {
            const int shift = 56 - 8 * pass;
            // eight copies of every bucket, picked by the lane: candidates all sit above the score threshold, so their keys
            // share their leading bits and a plain histogram's atomics pile onto two or three LDS words
            unsigned int *hist8 = reinterpret_cast<unsigned int *>(ssup);          // [256][8] (the masks' buffer: free until phase 2)
            for (int i = tid; i < 256 * 8; i += blockDim.x) hist8[i] = 0u;
            __syncthreads();
            for (int i = tid; i < n_stored; i += blockDim.x) {
                const unsigned long long key = skeys[i];
                if ((key & mask) == prefix) atomicAdd(&hist8[(((unsigned)(key >> shift) & 255u) << 3) | (lane & 7)], 1u);
            }
            __syncthreads();
            if (wave == 0) {
                // lane L owns buckets 255 - 4 L .. 252 - 4 L; cumulative counts in descending bucket order
                const int d0 = 255 - 4 * lane;
                auto bucket = [&](int d) {
                    const uint4 lo = *reinterpret_cast<const uint4 *>(&hist8[d << 3]), hi4 = *reinterpret_cast<const uint4 *>(&hist8[(d << 3) + 4]);
                    return lo.x + lo.y + lo.z + lo.w + hi4.x + hi4.y + hi4.z + hi4.w;
                };
                const unsigned int h0 = bucket(d0), h1 = bucket(d0 - 1), h2 = bucket(d0 - 2), h3 = bucket(d0 - 3);
                const unsigned int tot = h0 + h1 + h2 + h3;
                unsigned int inc = tot;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned int t = __shfl_up(inc, o);
                    if (lane >= o) inc += t;
                }
                const unsigned int exc = inc - tot + (unsigned int)above;
                const unsigned int c0 = exc + h0, c1 = c0 + h1, c2 = c1 + h2, c3 = c2 + h3;
                const unsigned long long cross = __ballot(c3 > (unsigned int)hi);
                if (cross == 0ull) {
                    if (lane == 63) { s_digit = -1; s_rem = (int)c3; }             // everything under this prefix fits
                } else if (lane == __ffsll((long long)cross) - 1) {
                    const int k = c0 > (unsigned int)hi ? 0 : (c1 > (unsigned int)hi ? 1 : (c2 > (unsigned int)hi ? 2 : 3));
                    s_digit = d0 - k;                                              // the bucket that would overflow `hi`
                    s_rem = (int)(k == 0 ? exc : (k == 1 ? c0 : (k == 2 ? c1 : c2)));   // keys above it
                }
            }
            __syncthreads();
            const int d = s_digit, before = s_rem;
            if (d < 0) { T = prefix; m = before; break; }
            if (before >= lo) { T = prefix | ((unsigned long long)(d + 1) << shift); m = before; break; }   // (d = 255 cannot get here: before = above < lo)
            prefix |= (unsigned long long)d << shift;
            mask |= 0xffull << shift;
            above = before;
            m = before;     // (after the last pass buckets are single keys: the loop has ended above)
        }
        if (tid == 0) s_fill = 0;
        __syncthreads();
        for (int i = tid; i < n_stored; i += blockDim.x) {
            const unsigned long long key = skeys[i];
            if (key >= T) {
                const int pos = atomicAdd(&s_fill, 1);
                if (pos < kRankSortMax) srank[pos] = key;
            }
        }
        __syncthreads();
        m = s_fill < hi ? s_fill : hi;                  // (= the count the search ended on)
        for (int i = tid; i < m; i += blockDim.x) skeys[i] = srank[i];
        __syncthreads();
        n_stored = m;
        truncated = true;
    }
    IRMV_STAMP(8);
    kp_lds = a.keys_only && a.classwalk && n_stored <= kRankSortUse;
    if (kp_lds) {
    // ================= up to 512 candidates whose boxes nobody has decoded yet: the usual frame (round 4) =================
    // Same algorithm, reorganised around three observations.  (1) A candidate's box is needed at its RANK, not at its anchor:
    // the keys are sorted first, the boxes are decoded in score order straight into LDS and the keypoint logits come with
    // them -- no store to and gather from the global box list (two memory round trips, ~ 2 us each on a lone frame), no
    // keypoint read after the walk (a third).  (Requesting the logits BEFORE the sort, to run it under their round trip, was
    // built first: 32 more registers across the sort loop spilled eighteen values to scratch memory all over the kernel.)  (2) Suppression is class-aware, so the walk of one class never looks at
    // another: candidates are regrouped class-major (score order kept inside a class), the IoU tests of a row are then a
    // DENSE run over the class's earlier members (no per-word class filter and bit loop: rows of different classes in one
    // wave made every wave wait for its longest loop), and the sixteen waves walk the sixteen classes side by side.  (3) The
    // max_det cap cuts the survivors in SCORE order whatever order they were found in: survivors are flagged at their
    // score-order position and counted off there.  Comparisons, operands and their order are the oracle's: bit-identical
    // (tests: every parity test runs this path; IRMV_NMS_CLASSWALK=0 is the round-3 path, compared bitwise).
    {
        f32x4 *cbox = &stage_box[0][0], *mbox = cbox + kRankSortUse;   // boxes in score order / class-major order
        int *ccls = &stage_cls[0][0], *minfo = ccls + kRankSortUse;    // class in score order / (score position << 8 | class), class-major
        float *marea = reinterpret_cast<float *>(srank + kRankSortUse); // box areas, class-major (srank holds 512 sorted keys here)
        float *ckpt = reinterpret_cast<float *>(skeys + kKptLds);       // keypoint logits in score order, 8 per candidate
        unsigned short *kept_src = &cls_list[0][0];                     // survivor -> score position (the per-class lists are not used here)
        // (lane numbers behind an opaque copy: the address arithmetic of this path, hoisted out of the attempt loop as loop
        // invariants, had seven values spilled to scratch memory at the loop's head and reloaded here)
        int tid_ = threadIdx.x;
        asm volatile("" : "+v"(tid_));
        const int tid = tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int q = tid & 3, base = lane & ~3;
        // ---- F1: rank sort (keys are unique: "keys greater than mine" is a permutation), as the rank sort of the other path ----
        const int n8 = (n_stored + 7) & ~7;
        int *s_rankacc = reinterpret_cast<int *>(ssup);
        for (int i = n_stored + tid; i < n8; i += blockDim.x) skeys[i] = 0ull;
        for (int i = tid; i < n_stored; i += blockDim.x) s_rankacc[i] = 0;
        if (tid < kLazyW) s_keptw[tid] = 0ull;
        __syncthreads();
        if (n_stored > 0) {
            const int P = min(16, (int)blockDim.x / n_stored);
            const int per = ((n8 / 8 + P - 1) / P) * 8;
            const int p = tid / n_stored, i = tid - p * n_stored;
            const int j_lo = p * per, j_hi = min(j_lo + per, n8);
            if (p < P && j_lo < j_hi) {
                const unsigned long long mine = skeys[i];
                int rank = 0;
#pragma unroll 2
                for (int j = j_lo; j < j_hi; j += 8) {
                    const ulonglong2 k0 = *reinterpret_cast<const ulonglong2 *>(&skeys[j]);
                    const ulonglong2 k1 = *reinterpret_cast<const ulonglong2 *>(&skeys[j + 2]);
                    const ulonglong2 k2 = *reinterpret_cast<const ulonglong2 *>(&skeys[j + 4]);
                    const ulonglong2 k3 = *reinterpret_cast<const ulonglong2 *>(&skeys[j + 6]);
                    rank += (k0.x > mine) + (k0.y > mine) + (k1.x > mine) + (k1.y > mine) + (k2.x > mine) + (k2.y > mine) +
                            (k3.x > mine) + (k3.y > mine);
                }
                if (P == 1) s_rankacc[i] = rank; else atomicAdd(&s_rankacc[i], rank);
            }
        }
        __syncthreads();
        for (int i = tid; i < n_stored; i += blockDim.x) srank[s_rankacc[i]] = skeys[i];
        __syncthreads();
        sorted = srank;
        IRMV_STAMP(9);
        // ---- F2: boxes (exactly decode_keys' arithmetic), classes and keypoints of the candidates IN SCORE ORDER: quad r / 4
        // decodes the candidate at sorted position r.  Both rounds' logits are requested before either is used ----
        n = n_stored < a.pre_nms_cap ? n_stored : a.pre_nms_cap;
        {
            f32x4 v[2][4];
            float kq0[2], kq1[2];
            uint32_t id_[2];
            int an_[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int r = u * 256 + (tid >> 2);
                const bool live = r < n_stored;                    // quad-uniform
                const unsigned long long key = live ? srank[r] : 0ull;
                id_[u] = 0xffffffffu - (uint32_t)(key & 0xffffffffu);
                an_[u] = live ? anchor_of(id_[u], a.nc, a.A) : 0;
                int ix, iy, st, lbase, lhw, rin;
                anchor_geom(an_[u], a.net, ix, iy, st, lbase, lhw, rin);
                const float *rec = head_rec(a.head_all, a.slots_total, a.first + b, lbase, lhw, rin);
#pragma unroll
                for (int i = 0; i < 4; i++) v[u][i] = reinterpret_cast<const f32x4 *>(rec + 16 * q)[i];
                kq0[u] = a.nk >= 8 ? rec[kKptOff + 2 * q] : 0.f;
                kq1[u] = a.nk >= 8 ? rec[kKptOff + 2 * q + 1] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (u * 256 >= n_stored) break;                    // workgroup-uniform
                const int r = u * 256 + (tid >> 2);
                float l[16];
#pragma unroll
                for (int i = 0; i < 4; i++) { l[4 * i] = v[u][i][0]; l[4 * i + 1] = v[u][i][1]; l[4 * i + 2] = v[u][i][2]; l[4 * i + 3] = v[u][i][3]; }
                const float d = dfl_side(l);
                const float dl = __shfl(d, base), dt = __shfl(d, base + 1), dr = __shfl(d, base + 2), db = __shfl(d, base + 3);
                if (r < n_stored) {
                    if (q == 0) {
                        int ix, iy, st, lbase, lhw, rin;
                        anchor_geom(an_[u], a.net, ix, iy, st, lbase, lhw, rin);
                        const float ax = (float)ix + 0.5f, ay = (float)iy + 0.5f, sf = (float)st;
                        f32x4 box;
                        box[0] = (ax - dl) * sf;
                        box[1] = (ay - dt) * sf;
                        box[2] = (ax + dr) * sf;
                        box[3] = (ay + db) * sf;
                        cbox[r] = box;
                        ccls[r] = r < n ? (int)(id_[u] % (uint32_t)a.nc) : -1;   // (behind pre_nms_cap: not walked)
                    }
                    ckpt[8 * r + 2 * q] = kq0[u];
                    ckpt[8 * r + 2 * q + 1] = kq1[u];
                }
            }
        }
        for (int i = n_stored + tid; i < ((n_stored + 63) & ~63); i += blockDim.x) ccls[i] = -1;   // padding up to the block boundary: no class
        __syncthreads();
        IRMV_STAMP(1);
        // ---- F4: class-major order.  Per class and 64-candidate word of the score order: who has that class ----
        const int nw = (n_stored + 63) >> 6;
        if (wave < nw) {
            const int c = ccls[tid];
            for (int k = 0; k < a.nc; k++) {
                const unsigned long long mk = __ballot(c == k);
                if (lane == 0) s_clsmask[k][wave] = mk;
            }
        }
        __syncthreads();
        if (wave == 0) {           // class sizes -> class-major base offsets, and the widest class in 64-bit words
            int cnt = 0;
            if (lane < a.nc)
                for (int w = 0; w < nw; w++) cnt += __popcll(s_clsmask[lane][w]);
            int inc = cnt, mx = cnt;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const int t = __shfl_up(inc, o), m2 = __shfl_xor(mx, o);
                if (lane >= o) inc += t;
                mx = mx > m2 ? mx : m2;
            }
            if (lane < 16) { s_cbase[lane] = inc - cnt; s_ccnt[lane] = cnt; }
            if (lane == 0) s_W = (mx + 63) >> 6;
        }
        __syncthreads();
        if (tid < (nw << 6)) {
            const int i = tid, c = ccls[i];
            if (c >= 0) {
                int r = __popcll(s_clsmask[c][wave] & ((1ull << lane) - 1ull));
                for (int w = 0; w < wave; w++) r += __popcll(s_clsmask[c][w]);
                const int k = s_cbase[c] + r;
                const f32x4 box = cbox[i];
                mbox[k] = box;
                marea[k] = (box[2] - box[0]) * (box[3] - box[1]);
                minfo[k] = (i << 8) | c;
            }
        }
        __syncthreads();
        IRMV_STAMP(10);
        // ---- F5: suppression rows in CLASS-LOCAL coordinates: bit jj of word w of row k <=> member 64 w + jj of k's class
        // (earlier than k in score order) has IoU > thr with k.  An item is a quarter word (16 members); the quarter index is the
        // slowest-running one, so that whole waves fall idle where classes are shorter than the widest ----
        {
            const int W = s_W;
            unsigned short *rows16 = reinterpret_cast<unsigned short *>(ssup);
            for (int item = tid; item < 4 * W * n; item += blockDim.x) {
                const int rest = item / n, k = item - rest * n;
                const int q4 = rest / W, w = rest - q4 * W;
                const int c = minfo[k] & 255, cb = s_cbase[c], r = k - cb;
                const int j0 = (w << 6) + (q4 << 4);
                unsigned int mask = 0u;
                if (j0 < r) {
                    const f32x4 bi = mbox[k];
                    const float ai = marea[k];
                    const int jend = r - j0 < 16 ? r - j0 : 16;
                    // all sixteen tests, predicated, in two batches of eight whose LDS reads are issued together: a loop to jend
                    // paid the LDS latency and the test's dependent chain once per member (~ 300 cycles each).  (Members behind
                    // jend: other rows of the list, or stale slots behind it inside the 512-slot arrays -- read, never used.)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        if (8 * h >= jend) break;
                        f32x4 bj[8];
                        float aj[8];
#pragma unroll
                        for (int jj = 0; jj < 8; jj++) {
                            const int m = min(cb + j0 + 8 * h + jj, kRankSortUse - 1);
                            bj[jj] = mbox[m];
                            aj[jj] = marea[m];
                        }
#pragma unroll
                        for (int jj = 0; jj < 8; jj++)
                            if (8 * h + jj < jend && iou_gt_area(bj[jj], aj[jj], bi, ai, a.iou_thr)) mask |= 1u << (8 * h + jj);
                    }
                }
                rows16[((k * kMatW_ + w) << 2) + q4] = (unsigned short)mask;
            }
        }
        __syncthreads();
        IRMV_STAMP(2);
        // ---- F6: wave c walks class c, 64 members per step (the in-order resolve of the matrix path below) ----
        if (wave < a.nc) {
            const int cnt = s_ccnt[wave], cb = s_cbase[wave];
            unsigned long long *ck = s_clsmask[wave];          // from here on: the class's survivors, class-local positions
            for (int blk = 0; blk < ((cnt + 63) >> 6); blk++) {
                const int il = (blk << 6) + lane, k = cb + il;
                const bool valid = il < cnt;
                bool alive = valid;
                for (int w = 0; w < blk; w++)
                    if (valid && (ssup[k * kMatW_ + w] & ck[w]) != 0ull) alive = false;
                const unsigned long long sup = valid ? ssup[k * kMatW_ + blk] : 0ull;
                unsigned long long U = __ballot(alive), K = 0ull;   // undecided, kept
                while (U) {
                    const bool ready = ((U >> lane) & 1ull) && (sup & U) == 0ull;
                    const unsigned long long R = __ballot(ready);
                    K |= __ballot(ready && (sup & K) == 0ull);
                    U &= ~R;
                }
                if (lane == 0) ck[blk] = K;
                if ((K >> lane) & 1ull) {
                    const int i = minfo[k] >> 8;                 // flag the survivor at its score-order position
                    atomicOr(reinterpret_cast<unsigned int *>(s_keptw) + (i >> 5), 1u << (i & 31));
                }
                wave_lds_sync();
            }
        }
        __syncthreads();
        // ---- F7: the first max_det survivors in score order ----
        if (tid < (nw << 6)) {
            const unsigned long long word = s_keptw[wave];
            if ((word >> lane) & 1ull) {
                int pos = __popcll(word & ((1ull << lane) - 1ull));
                for (int w = 0; w < wave; w++) pos += __popcll(s_keptw[w]);
                if (pos < a.max_det) {
                    kept_box[pos] = cbox[tid];
                    kept_cls[pos] = ccls[tid];
                    kept_key[pos] = srank[tid];
                    kept_src[pos] = (unsigned short)tid;
                }
            }
        }
        if (tid == 0) {
            int total = 0;
            for (int w = 0; w < nw; w++) total += __popcll(s_keptw[w]);
            s_kept = total < a.max_det ? total : a.max_det;
        }
    }
    } else {
    if (a.keys_only) {
        decode_keys(skeys, n_stored);
        __syncthreads();                                // boxes visible to the whole workgroup
    }
    IRMV_STAMP(9);

    if (n_stored <= kRankSortUse) {
        // keys are unique, so "number of keys greater than mine" is a permutation.  A key's count is split over P lanes (all
        // 1024 of the workgroup for n >= 342: with one lane per key a 380-candidate frame kept six waves busy for 16 k cycles),
        // each counting over its own stretch of the list; the partial counts meet in an LDS counter per key (the masks'
        // buffer: free until phase 2).
        const int n8 = (n_stored + 7) & ~7;
        int *s_rankacc = reinterpret_cast<int *>(ssup);
        for (int i = n_stored + tid; i < n8; i += blockDim.x) skeys[i] = 0ull;   // pad: never greater than a real key
        for (int i = tid; i < n_stored; i += blockDim.x) s_rankacc[i] = 0;
        __syncthreads();
        if (n_stored > 0) {
            const int P = min(16, (int)blockDim.x / n_stored);                  // lanes per key (n_stored <= 512: at least two)
            const int per = ((n8 / 8 + P - 1) / P) * 8;                         // keys per stretch (a multiple of 8)
            const int p = tid / n_stored, i = tid - p * n_stored;
            const int j_lo = p * per, j_hi = min(j_lo + per, n8);
            if (p < P && j_lo < j_hi) {
                const unsigned long long mine = skeys[i];
                int rank = 0;
#pragma unroll 2
                for (int j = j_lo; j < j_hi; j += 8) {
                    const ulonglong2 k0 = *reinterpret_cast<const ulonglong2 *>(&skeys[j]);
                    const ulonglong2 k1 = *reinterpret_cast<const ulonglong2 *>(&skeys[j + 2]);
                    const ulonglong2 k2 = *reinterpret_cast<const ulonglong2 *>(&skeys[j + 4]);
                    const ulonglong2 k3 = *reinterpret_cast<const ulonglong2 *>(&skeys[j + 6]);
                    rank += (k0.x > mine) + (k0.y > mine) + (k1.x > mine) + (k1.y > mine) + (k2.x > mine) + (k2.y > mine) +
                            (k3.x > mine) + (k3.y > mine);
                }
                if (P == 1) s_rankacc[i] = rank; else atomicAdd(&s_rankacc[i], rank);
            }
        }
        __syncthreads();
        for (int i = tid; i < n_stored; i += blockDim.x) srank[s_rankacc[i]] = skeys[i];
        __syncthreads();
        sorted = srank;
    } else {
        int npow = 512;
        while (npow < n_stored) npow <<= 1;
        for (int i = n_stored + tid; i < npow; i += blockDim.x) skeys[i] = 0ull;
        __syncthreads();
        for (int k = 2; k <= npow; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < npow; i += blockDim.x) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long x = skeys[i], y = skeys[l];
                        const bool desc = (i & k) == 0;
                        if (desc ? (x < y) : (x > y)) { skeys[i] = y; skeys[l] = x; }
                    }
                }
                __syncthreads();
            }
        }
        sorted = skeys;
    }
    n = n_stored < a.pre_nms_cap ? n_stored : a.pre_nms_cap;
    const f32x4 *boxes = reinterpret_cast<const f32x4 *>(a.boxes) + (size_t)b * a.A;
    IRMV_STAMP(1);

    constexpr int kMatN = 512, kMatW = kMatW_;   // up to this many candidates the FULL suppression matrix fits ssup
    constexpr int kLazyN = kLazyN_;   // up to this many: the matrix one 64-candidate block of rows at a time
    if (n <= kMatN) {
        // ---- 2'. full suppression matrix ----
        // Every IoU test the greedy walk can need, evaluated up front by the whole workgroup; the walk itself is then
        // AND / readlane work on 64-bit words (no boxes, no per-class kept lists).  Same comparisons as the oracle's walk.
        f32x4 *cbox = &stage_box[0][0];      // [1024] -> candidate boxes / classes in sorted order
        int *ccls = &stage_cls[0][0];
        for (int i = tid; i < ((n + 63) & ~63); i += blockDim.x) {
            if (i < n) {
                const uint32_t id = 0xffffffffu - (uint32_t)(sorted[i] & 0xffffffffu);
                cbox[i] = boxes[anchor_of(id, a.nc, a.A)];
                ccls[i] = (int)(id % (uint32_t)a.nc);
            } else {
                ccls[i] = -1;      // padding up to the block boundary: no class
            }
        }
        if (tid < kMatW) s_keptw[tid] = 0ull;
        __syncthreads();
        const int nw = (n + 63) >> 6;
        // per class and 64-candidate word: which candidates have that class (one ballot per class and wave) -- the class
        // filter of the matrix rows below is then ONE 8-byte LDS read per word instead of sixteen 16-byte ones
        if (wave < kMatW) {
            const int c = tid < (nw << 6) ? ccls[tid] : -1;
            for (int k = 0; k < a.nc; k++) {
                const unsigned long long mk = __ballot(c == k);
                if (lane == 0) s_clsmask[k][wave] = mk;
            }
        }
        __syncthreads();
        IRMV_STAMP(10);
        // Row i of the matrix: bit jj of word w <=> candidate j = 64 w + jj (j < i) has i's class and IoU(j, i) > thr -- who
        // suppresses i if kept.  Rows are built for a range of 64-candidate blocks at a time (see below).
        // (an item is HALF a word: twice the items of half the length spread more evenly over the 1024 lanes -- a row's cost
        // is its same-class candidates in the word, anything from none to 64)
        uint32_t *ssup32 = reinterpret_cast<uint32_t *>(ssup);
        auto build_rows = [&](int b_lo, int b_hi) {            // rows of blocks [b_lo, b_hi): words 0 .. b_hi - 1
            const int i_lo = b_lo << 6, rows = min(n, b_hi << 6) - i_lo;
            for (int item = tid; item < 2 * rows * b_hi; item += blockDim.x) {
                const int h = item & 1, it2 = item >> 1;
                const int w = it2 / rows, i = i_lo + (it2 - w * rows);   // (nearly) consecutive lanes: consecutive i, same word -> broadcast reads of j
                uint32_t mask = 0u;
                const int j0 = (w << 6) + (h << 5);
                if (j0 < i) {
                    const f32x4 bi = cbox[i];
                    const int ci = ccls[i];
                    const int jend = i - j0 < 32 ? i - j0 : 32;
                    // class filter first (the per-class word built above), then the IoU test only for the few same-class candidates
                    uint32_t same = (uint32_t)(s_clsmask[ci][w] >> (h << 5));
                    if (jend < 32) same &= (1u << jend) - 1u;
                    while (same) {
                        const int jj = __ffs((int)same) - 1;
                        same &= same - 1u;
                        if (iou_gt(cbox[j0 + jj], bi, a.iou_thr)) mask |= 1u << jj;
                    }
                }
                ssup32[(i * kMatW + w) * 2 + h] = mask;
            }
        };
        // ---- 3'. greedy walk on wave 0, 64 candidates per step ----
        auto walk_blocks = [&](int b_lo, int b_hi, int kept) -> int {
            for (int blk = b_lo; blk < b_hi && kept < a.max_det; blk++) {
                const int idx = (blk << 6) + lane;
                const bool valid = idx < n;
                bool alive = valid;
                for (int w = 0; w < blk; w++)                      // suppressed by a kept candidate of an earlier block?
                    if (valid && (ssup[idx * kMatW + w] & s_keptw[w]) != 0ull) alive = false;
                const unsigned long long sup = valid ? ssup[idx * kMatW + blk] : 0ull;
                // The block's in-order resolve, without walking it candidate by candidate (one wave alone on its SIMD pays
                // every instruction's full latency: ~200 cycles per survivor that way).  A candidate whose possible
                // suppressors (its sup bits) are all DECIDED is decided itself: kept iff none of them was kept.  Every round
                // decides at least the first undecided candidate, usually most of them; the kept set is the sequential walk's.
                unsigned long long U = __ballot(alive), K = 0ull;   // undecided, kept
                while (U) {
                    const bool ready = ((U >> lane) & 1ull) && (sup & U) == 0ull;
                    const unsigned long long R = __ballot(ready);
                    K |= __ballot(ready && (sup & K) == 0ull);
                    U &= ~R;
                }
                // the cap: the walk stops with the max_det-th survivor (a candidate's fate depends on the ones before it only)
                const int room = a.max_det - kept;
                unsigned long long A = K;
                if (__popcll(K) > room) {
                    const bool over = ((K >> lane) & 1ull) && __popcll(K & ((1ull << lane) - 1ull)) >= room;
                    A = K & ~__ballot(over);
                }
                const bool mine = (A >> lane) & 1ull;
                const int pos = kept + __popcll(A & ((1ull << lane) - 1ull));
                if (mine) {
                    kept_box[pos] = cbox[idx];
                    kept_cls[pos] = ccls[idx];
                    kept_key[pos] = sorted[idx];
                }
                if (lane == 0) s_keptw[blk] = A;
                wave_lds_sync();
                kept += __popcll(A);
            }
            return kept;
        };
        // (Measured and dropped: rows in two stages, the second only if the walk over the first has not filled max_det -- on
        // the 380-candidate benchmark frame the hundredth survivor sits in the fifth of six blocks, and the extra barriers cost
        // more than the unbuilt rows save.)
        build_rows(0, nw);
        __syncthreads();
        IRMV_STAMP(2);
        if (wave == 0) {
            const int kept = walk_blocks(0, nw, 0);
            if (lane == 0) s_kept = kept;
        }
    } else if (n <= kLazyN) {
        // ---- 2'' / 3''. crowded frames (513 .. 1024 candidates): the same matrix rows, built one 64-candidate block at a time
        // right before the walk reaches that block, and not at all once max_det survivors are found.  (A frame with 900
        // candidates fills max_det = 100 in its first few hundred; the all-pairs matrix would run 6 x the IoU tests of a
        // 380-candidate frame, and the per-class-list walk below it took 85 us on such frames.)
        f32x4 *cbox = &stage_box[0][0];
        int *ccls = &stage_cls[0][0];
        for (int i = tid; i < ((n + 63) & ~63); i += blockDim.x) {
            if (i < n) {
                const uint32_t id = 0xffffffffu - (uint32_t)(sorted[i] & 0xffffffffu);
                cbox[i] = boxes[anchor_of(id, a.nc, a.A)];
                ccls[i] = (int)(id % (uint32_t)a.nc);
            } else {
                ccls[i] = -1;
            }
        }
        if (tid < kLazyW) s_keptw[tid] = 0ull;
        if (tid == 0) s_kept = 0;
        __syncthreads();
        const int nw = (n + 63) >> 6;
        {
            const int c = tid < (nw << 6) ? ccls[tid] : -1;        // 16 waves = 16 blocks of 64
            for (int k = 0; k < a.nc; k++) {
                const unsigned long long mk = __ballot(c == k);
                if (lane == 0) s_clsmask[k][wave] = mk;
            }
        }
        __syncthreads();
        uint32_t *rows32 = reinterpret_cast<uint32_t *>(ssup);     // [64 rows][kLazyW words][2 halves] of the current block
        IRMV_STAMP(2);
        for (int blk = 0; blk < nw; blk++) {
            if (s_kept >= a.max_det) break;                        // workgroup-uniform (read behind the barrier below)
            for (int item = tid; item < 128 * (blk + 1); item += blockDim.x) {
                const int h = item & 1, row = (item >> 1) & 63, w = item >> 7;
                const int i = (blk << 6) + row;
                uint32_t mask = 0u;
                const int j0 = (w << 6) + (h << 5);
                if (i < n && j0 < i) {
                    const f32x4 bi = cbox[i];
                    const int jend = i - j0 < 32 ? i - j0 : 32;
                    uint32_t same = (uint32_t)(s_clsmask[ccls[i]][w] >> (h << 5));
                    if (jend < 32) same &= (1u << jend) - 1u;
                    while (same) {
                        const int jj = __ffs((int)same) - 1;
                        same &= same - 1u;
                        if (iou_gt(cbox[j0 + jj], bi, a.iou_thr)) mask |= 1u << jj;
                    }
                }
                rows32[(row * kLazyW + w) * 2 + h] = mask;
            }
            __syncthreads();
            if (wave == 0) {
                const int kept = s_kept;
                const int idx = (blk << 6) + lane;
                const bool valid = idx < n;
                bool alive = valid;
                for (int w = 0; w < blk; w++)
                    if (valid && (ssup[lane * kLazyW + w] & s_keptw[w]) != 0ull) alive = false;
                const unsigned long long sup = valid ? ssup[lane * kLazyW + blk] : 0ull;
                unsigned long long U = __ballot(alive), K = 0ull;   // undecided, kept (rounds: see the matrix path above)
                while (U) {
                    const bool ready = ((U >> lane) & 1ull) && (sup & U) == 0ull;
                    const unsigned long long R = __ballot(ready);
                    K |= __ballot(ready && (sup & K) == 0ull);
                    U &= ~R;
                }
                const int room = a.max_det - kept;
                unsigned long long A = K;
                if (__popcll(K) > room) {
                    const bool over = ((K >> lane) & 1ull) && __popcll(K & ((1ull << lane) - 1ull)) >= room;
                    A = K & ~__ballot(over);
                }
                const bool mine = (A >> lane) & 1ull;
                const int pos = kept + __popcll(A & ((1ull << lane) - 1ull));
                if (mine) {
                    kept_box[pos] = cbox[idx];
                    kept_cls[pos] = ccls[idx];
                    kept_key[pos] = sorted[idx];
                }
                if (lane == 0) { s_keptw[blk] = A; s_kept = kept + __popcll(A); }
            }
            __syncthreads();
        }
    } else {
    // ---- 2. intra-block masks, one block per wave ----
    const int n_pre = n < kSupCap ? n : kSupCap;
    for (int start = wave * 64; start < n_pre; start += 16 * 64) {
        const int idx = start + lane;
        const bool valid = idx < n;
        const unsigned long long key = valid ? sorted[idx] : 0ull;
        const uint32_t id = 0xffffffffu - (uint32_t)(key & 0xffffffffu);
        const int an = valid ? anchor_of(id, a.nc, a.A) : 0;
        const int cls = valid ? (int)(id % (uint32_t)a.nc) : -1;
        const f32x4 box = valid ? boxes[an] : (f32x4){0.f, 0.f, 0.f, 0.f};
        ssup[idx] = block_sup_mask(box, cls, lane, a.iou_thr, stage_box[wave], stage_cls[wave]);
    }
    __syncthreads();
    IRMV_STAMP(2);

    // ---- 3. greedy walk on wave 0 ----
    if (wave == 0) {
        int kept = 0;
        for (int start = 0; start < n && kept < a.max_det; start += 64) {
            const int idx = start + lane;
            const bool valid = idx < n;
            const unsigned long long key = valid ? sorted[idx] : 0ull;
            const uint32_t id = 0xffffffffu - (uint32_t)(key & 0xffffffffu);
            const int an = valid ? anchor_of(id, a.nc, a.A) : 0;
            const int cls = valid ? (int)(id % (uint32_t)a.nc) : -1;
            const f32x4 box = valid ? boxes[an] : (f32x4){0.f, 0.f, 0.f, 0.f};
            // against the kept boxes of my class
            bool alive = valid;
            const int mycnt = valid ? cls_cnt[cls] : 0;
            for (int j = 0; __any(j < mycnt); j++) {
                if (j < mycnt) {
                    const int k = cls_list[cls][j];
                    if (iou_gt(kept_box[k], box, a.iou_thr)) alive = false;
                }
            }
            const unsigned long long sup = start < n_pre ? ssup[idx]
                                                        : block_sup_mask(box, cls, lane, a.iou_thr, stage_box[0], stage_cls[0]);
            // sequential resolve, earliest first
            unsigned long long A = __ballot(alive);
            int taken = 0;
            for (unsigned long long todo = A; todo;) {          // walk the still-alive candidates in order
                const int j = __ffsll((long long)todo) - 1;
                if (kept + taken >= a.max_det) {
                    A &= (1ull << j) - 1ull;                      // cap reached: drop j and everything after
                    break;
                }
                taken++;
                const unsigned long long col = __ballot((sup >> j) & 1ull);   // lanes that j suppresses (all > j)
                A &= ~col;
                todo = A & ~((2ull << j) - 1ull);
            }
            const bool mine = (A >> lane) & 1ull;
            const int pos = kept + __popcll(A & ((1ull << lane) - 1ull));
            if (mine) {
                kept_box[pos] = box;
                kept_cls[pos] = cls;
                kept_key[pos] = key;
            }
            if (mine) cls_list[cls][atomicAdd(&cls_cnt[cls], 1)] = (unsigned short)pos;   // order within a class is irrelevant
            wave_lds_sync();
            kept += __popcll(A);
        }
        if (lane == 0) s_kept = kept;
    }
    }
    }
    __syncthreads();
    // attempt 0 stands if it filled max_det or walked everything the full algorithm would walk
    if (!truncated || s_kept >= a.max_det || n >= n_full) break;
    __syncthreads();                                    // (everyone has read s_kept before the next attempt resets it)
    }
    IRMV_STAMP(3);
    const int kept = s_kept;
    if (tid == 0) {
        DevFrameOut fo;
        fo.num_dets = kept;
        fo.n_candidates = n_total;
        fo.overflow = 0;   // (the candidate list is sized for every (anchor, class) pair; kept for ABI stability)
        fo.pad = 0;
        a.fout[b] = fo;
    }
    // ---- 4. one PAIR of lanes per survivor: the record is assembled redundantly on both lanes, the fp64 PnP runs spread
    // over them (solve_pnp_ippe_pair: two undistorted points / one IPPE solution per lane), lane 0 stages the record ----
    DevDet *sdet = reinterpret_cast<DevDet *>(skeys);   // (the key list is dead: the survivors' keys sit in kept_key)
    const int j = tid >> 1;
    {
        // keypoint logits: from LDS where the class-walk path has put them, else from the head record.  (Read BEFORE the
        // staging is written: with max_det > 236 the records reach into the keypoints' part of the key list.)
        float kpv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int an = 0;
        if (j < kept) {
            const uint32_t id = 0xffffffffu - (uint32_t)(kept_key[j] & 0xffffffffu);
            an = anchor_of(id, a.nc, a.A);
            if (kp_lds) {
                const f32x4 *kl = reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(skeys + kKptLds) + 8 * cls_list[0][j]);
                const f32x4 k0 = kl[0], k1 = kl[1];
                kpv[0] = k0[0]; kpv[1] = k0[1]; kpv[2] = k0[2]; kpv[3] = k0[3];
                kpv[4] = k1[0]; kpv[5] = k1[1]; kpv[6] = k1[2]; kpv[7] = k1[3];
            } else {
                int ix, iy, s, lbase, lhw, rin;
                anchor_geom(an, a.net, ix, iy, s, lbase, lhw, rin);
                const float *kp = head_rec(a.head_all, a.slots_total, a.first + b, lbase, lhw, rin) + kKptOff;
#pragma unroll
                for (int q = 0; q < 8; q++) kpv[q] = q < a.nk ? kp[q] : 0.f;
            }
        }
        __syncthreads();
        if (kept == 0) { IRMV_STAMP(11); IRMV_STAMP(12); }   // (diagnostic stamps: a frame without survivors has no keypoint / PnP phase; without these its two slots printed garbage)
        if (j < kept) {           // pair-uniform
            // the record's fields go to the LDS staging as they are produced (lane 0 of the pair): held in registers across the
            // solver they spilled
            DevDet *d = &sdet[j];
            const bool w = (tid & 1) == 0;
            const f32x4 box = kept_box[j];
            const unsigned long long key = kept_key[j];
            const float logit = unorderable((uint32_t)(key >> 32));
            int ix, iy, s, lbase, lhw, rin;
            anchor_geom(an, a.net, ix, iy, s, lbase, lhw, rin);
            const float axm = ((float)ix + 0.5f) - 0.5f, aym = ((float)iy + 0.5f) - 0.5f, sf = (float)s;
            if (w) {
                d->score = 1.0f / (1.0f + irmv_expf(-logit));
                d->cls = kept_cls[j];
                d->anchor = an;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    d->box_net[i] = box[i];
                    const float off = (i & 1) ? a.off_y : a.off_x, sc = (i & 1) ? a.scale_y : a.scale_x;
                    d->xyxy[i] = (box[i] - off) * sc;
                }
            }
            float px0 = 0.f, py0 = 0.f, px1 = 0.f, py1 = 0.f;      // this lane's two points for the solver (q and q + 2)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float kx = 0.f, ky = 0.f;
                if (2 * q + 1 < a.nk) {
                    kx = (2.0f * kpv[2 * q] + axm) * sf;
                    ky = (2.0f * kpv[2 * q + 1] + aym) * sf;
                }
                const float sx = (kx - a.off_x) * a.scale_x, sy = (ky - a.off_y) * a.scale_y;
                if ((q & 1) == (tid & 1)) {
                    if (q < 2) { px0 = sx; py0 = sy; } else { px1 = sx; py1 = sy; }
                }
                if (w) {
                    d->kpts_net[2 * q] = kx;
                    d->kpts_net[2 * q + 1] = ky;
                    d->kpts[2 * q] = sx;
                    d->kpts[2 * q + 1] = sy;
                }
            }
            IRMV_STAMP(11);
            double rvec[3] = {0.0, 0.0, 0.0}, tvec[3] = {0.0, 0.0, 0.0}, quat[4] = {0.0, 0.0, 0.0, 1.0};
            int pnp_ok = 0;
            if (a.nk >= 8) pnp_ok = solve_pnp_ippe_pair(*a.pnp, px0, py0, px1, py1, a.armor_size, rvec, tvec, quat) ? 1 : 0;
            IRMV_STAMP(12);
            if (w) {
#pragma unroll
                for (int i = 0; i < 3; i++) { d->rvec[i] = rvec[i]; d->tvec[i] = tvec[i]; }
#pragma unroll
                for (int i = 0; i < 4; i++) d->quat[i] = quat[i];
                d->pnp_ok = pnp_ok;
                d->armor_valid = a.nk >= 8 ? 1 : 0;   // keypoint head: every detection carries its four points
                d->armor_size = a.armor_size;
                d->n_lights = 0;
                d->pad_ = 0;
            }
        }
    }
    __syncthreads();
    // The frame's records leave as ONE contiguous run of 16-byte stores (13 per record), zero-padded behind the last
    // survivor as EfficientNMS pads its outputs.  (One lane per record storing its own 208
    // bytes put 16 different cache lines under every store instruction: 4 us of a 38 us frame went into that.)
    {
        static_assert(sizeof(DevDet) % 16 == 0 && sizeof(DevDet) * kMaxDetCap <= sizeof(skeys), "record staging");
        constexpr int kVec = sizeof(DevDet) / 16;
        const u32x4_t *src = reinterpret_cast<const u32x4_t *>(sdet);
        u32x4_t *dst = reinterpret_cast<u32x4_t *>(a.dets + (size_t)b * a.max_det);
        const u32x4_t zero = {0u, 0u, 0u, 0u};
        for (int i = tid; i < a.max_det * kVec; i += blockDim.x) dst[i] = i < kept * kVec ? src[i] : zero;
    }
    IRMV_STAMP(4);
    if (a.dbg && tid == 0) { a.dbg[b * 16 + 5] = n_total; a.dbg[b * 16 + 6] = kept; }
#undef IRMV_STAMP
}

void launch_nms_pnp(const PostArgs &a, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(nms_pnp_kernel, dim3(batch), dim3(1024), 0, s, a);
}

}  // namespace irmv
