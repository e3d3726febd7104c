// A 64-channel Bottleneck of a C2f block (two 3x3 64 -> 64 convs, optional shortcut), alone or together with the block's
// closing 1x1 (cv2 over the concat), as ONE launch -- for steps of a single frame.
//
// Inside the reference's TensorRT plan (src/yolo_engine.cpp:105) these are layers of model.6 / 12 / 18 (40 x 40 at a 640
// net).  One frame at a time each of them is a launch of 25 - 100 small workgroups that lasts 3.8 - 4.9 us whatever it
// computes: a kernel boundary, a prologue, one or two dependent memory round trips, a few hundred MFMAs.
// The three C2f blocks at 40 x 40 are 14 such launches, 62 us of a 270 us step -- and the reference's product IS one
// detect() per camera frame (src/yolo_engine.cpp:153-177).  Here a workgroup of eight waves owns an 8 x 8 pixel tile and walks
//
//   mode A   m.cv1 on the tile + 1-pixel halo -> m.cv2 (+ shortcut) on the tile -> the next 64-channel slice of the block's
//            concat buffer (the first Bottleneck of an n = 2 block: its output is needed WITH halo by the next one)
//   mode B   m.cv1 -> m.cv2 (+ shortcut) -> cv2 over [y0 | .. | y_in | y_last] -> block output (the last Bottleneck)
//
// with every intermediate in LDS planes ([pixel][64 ch], 160-byte pixel pitch).  What makes it short is what a batched
// kernel could not afford: EVERY weight fragment a wave will use (18 + 18 + 6..8 fragments of 1 KiB, 168 - 176 VGPRs) is
// requested at kernel start, together with the input region -- one memory round trip for the whole launch, then LDS reads
// and MFMAs only.  A wave owns one 16-channel output tile (3x3 phases: waves w and w + 4 share it and split the pixels;
// cv2: eight tiles, eight waves), so a weight fragment is fetched by at most two waves.  The 192 - 209 KB of weights a
// workgroup pulls through its CU are why batched steps keep the per-layer kernels (section 4b: 25 tiles x 128 frames would
// move 0.6 GB per launch); a lone frame is 25 workgroups on an empty chip.
//
// Rounding points and K order are the per-layer kernels': 3x3 (chunk of 32 channels, tap) on the LDS family's nt = 1 weight
// packing, 1x1 in steps of 32 channels over the concat, accumulators starting at the log2 e-scaled bias, SiLU outputs
// through v_fma_mix (irmv_common.hpp), shortcut added to the rounded fp32 product -- bit-identical
// (tests/test_gpu_engine.py::test_single_frame_bottleneck_kernels_are_bitwise_the_layers).
#include "irmv_common.hpp"

#include <mutex>

namespace irmv {

namespace {
constexpr int BT = kBneckTile;
constexpr int B1W = BT + 4, B1N = B1W * B1W;   // input region (halo 2): 12 x 12
constexpr int B2W = BT + 2, B2N = B2W * B2W;   // m.cv1 output region (halo 1): 10 x 10
constexpr int B3N = BT * BT;                   // the tile: 64 pixels = 4 MFMA tiles
constexpr int BPS = 160;                       // bytes per pixel of a 64-channel plane: 128 + 32 of padding (16 consecutive pixels
                                               // x 2 lane groups of a ds_read_b128 hit 16 distinct 16-byte slots of the 256-byte bank window)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ld16g(const half_t *p) { return *reinterpret_cast<const u32x4 *>(p); }
__device__ __forceinline__ u32x4 keep16(bool c, u32x4 v) { return c ? v : (u32x4){0u, 0u, 0u, 0u}; }
}  // namespace

// MODE 0 = A, 1 = B.  KS2 = k-steps of cv2 (mode B: 6 for an n = 1 block, 8 for n = 2; mode A: unused).
template <int MODE, int KS2, bool SHORTCUT>
__global__ __launch_bounds__(512) void bneck64_kernel(BneckArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NEXTRA = MODE == 1 ? KS2 / 2 - 2 : 0;   // concat slices in front of y_in that cv2 reads (y0; y0 | y1)
    uint8_t *s_in = smem;                            // y_in region, zero outside the image (= the 3x3's padding)
    uint8_t *s_t = s_in + B1N * BPS;                 // m.cv1 output region
    uint8_t *s_yn = s_t + B2N * BPS;                 // m.cv2 output on the tile (mode B)
    uint8_t *s_y0 = s_yn + B3N * BPS;                // mode B: NEXTRA tile planes (y0 [, y1])
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x / tiles, tile = blockIdx.x - b * tiles;
    const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
    const int oy0 = tyi * BT, ox0 = txi * BT;
    const int H = a.H, W = a.W;
    const int n = wave & 3, ph = wave >> 2;          // 3x3 phases: output tile n (16 channels), pixel tiles ph, ph + 2, ..
    // channels of output tile t held by this lane (paired-tile packing, k_conv.hip): (t >> 1) * 32 + g * 8 + (t & 1) * 4 + [0, 4)
    const int ch3 = (n >> 1) * 32 + g * 8 + (n & 1) * 4, ch4 = (wave >> 1) * 32 + g * 8 + (wave & 1) * 4;

    // ---- every load of the launch is requested up front, in the order the phases need the data: the input region (phase 0
    // waits for it alone), then the weight fragments of m.cv1, m.cv2 and cv2.  The memory counter is in-order, so each
    // phase waits only for what was requested before ITS operands: m.cv2's and cv2's 26 fragments per wave keep streaming under
    // phase 0's LDS stores, the first barrier and m.cv1's MFMAs (the scheduling barriers pin the request order; left alone the
    // compiler issued one input load last and phase 0 then waited for every weight of the launch) ----
    constexpr int NP1 = (B1N * 8 + 511) / 512;
    u32x4 v[NP1], x[NEXTRA > 0 ? NEXTRA : 1];
#pragma unroll
    for (int i = 0; i < NP1; i++) {
        const int e = tid + i * 512, px = e >> 3, q = e & 7;
        const int ly = px / B1W, lx = px - ly * B1W;
        const int gy = oy0 - 2 + ly, gx = ox0 - 2 + lx;
        const bool in = e < B1N * 8 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        v[i] = ld16g(a.yin + ((size_t)(b * H + (in ? gy : 0)) * W + (in ? gx : 0)) * a.yin_ld + q * 8);
    }
    const int tpx = tid >> 3, tq = tid & 7;                  // (512 pieces per tile plane: one per thread)
    const int tgy = oy0 + (tpx >> 3), tgx = ox0 + (tpx & 7);
    const bool tin = tgy < H && tgx < W;
    if constexpr (NEXTRA > 0) {
#pragma unroll
        for (int j = 0; j < NEXTRA; j++)
            x[j] = ld16g(a.cat + ((size_t)(b * H + (tin ? tgy : 0)) * W + (tin ? tgx : 0)) * a.cat_ld + j * 64 + tq * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    half8 W1[18], W2[18], W3[MODE == 1 ? KS2 : 1];
    const f32x4 b1 = *reinterpret_cast<const f32x4 *>(a.b_m1 + ch3);
    {
        const half8 *w1 = reinterpret_cast<const half8 *>(a.w_m1) + (size_t)n * 18 * 64 + lane;   // nt = 1 packing: [tile][chunk][tap][lane]
#pragma unroll
        for (int s = 0; s < 18; s++) W1[s] = w1[s * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 b2 = *reinterpret_cast<const f32x4 *>(a.b_m2 + ch3);
    {
        const half8 *w2 = reinterpret_cast<const half8 *>(a.w_m2) + (size_t)n * 18 * 64 + lane;
#pragma unroll
        for (int s = 0; s < 18; s++) W2[s] = w2[s * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 b3 = b1;
    if constexpr (MODE == 1) {
        b3 = *reinterpret_cast<const f32x4 *>(a.b_cv2 + ch4);
        const half8 *w3 = reinterpret_cast<const half8 *>(a.w_cv2) + (size_t)wave * KS2 * 64 + lane;   // direct packing: [tile][k-step][lane]
#pragma unroll
        for (int ks = 0; ks < KS2; ks++) W3[ks] = w3[ks * 64];
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- 0: y_in with a 2-pixel halo (and, mode B, the tile's y0 [, y1]) -> LDS ----
    {
#pragma unroll
        for (int i = 0; i < NP1; i++) {
            const int e = tid + i * 512, px = e >> 3, q = e & 7;
            const int ly = px / B1W, lx = px - ly * B1W;
            const int gy = oy0 - 2 + ly, gx = ox0 - 2 + lx;
            if (e < B1N * 8) *reinterpret_cast<u32x4 *>(s_in + px * BPS + q * 16) = keep16((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W, v[i]);
        }
        if constexpr (NEXTRA > 0) {
#pragma unroll
            for (int j = 0; j < NEXTRA; j++) *reinterpret_cast<u32x4 *>(s_y0 + j * (B3N * BPS) + tpx * BPS + tq * 16) = keep16(tin, x[j]);
        }
    }
    __syncthreads();

    // ---- 1: m.cv1 (3x3, 64 -> 64, SiLU) on the 10 x 10 region: pixel tiles ph, ph + 2, ph + 4, ph + 6 (the region has 6.25) ----
    {
        f32x4 acc[4];
        int base[4], mpx[4];
        bool valid[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int mm = (ph + 2 * i) * 16 + r;
            valid[i] = mm < B2N;
            mpx[i] = valid[i] ? mm : 0;
            const int ly = mpx[i] / B2W, lx = mpx[i] - ly * B2W;
            base[i] = (ly * B1W + lx) * BPS + g * 16;
            acc[i] = b1;
        }
        // the fragments of k-step s + 1 are requested from LDS before the MFMAs of k-step s are issued (the scheduling barriers
        // pin that order: left alone the compiler reads each fragment right in front of its MFMA and waits for it there)
        half8 Bf[2][4];
        auto frag = [&](int s, half8 (&B)[4]) {
            const int c = s / 9, tap = s - c * 9, kh = tap / 3, kw = tap - kh * 3;
            const int off = (kh * B1W + kw) * BPS + c * 64;
#pragma unroll
            for (int i = 0; i < 4; i++) B[i] = *reinterpret_cast<const half8 *>(s_in + base[i] + off);
        };
        frag(0, Bf[0]);
#pragma unroll
        for (int s = 0; s < 18; s++) {
            if (s + 1 < 18) frag(s + 1, Bf[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W1[s], Bf[s & 1][i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (!valid[i]) continue;
            const int ly = mpx[i] / B2W, lx = mpx[i] - ly * B2W;
            const int gy = oy0 - 1 + ly, gx = ox0 - 1 + lx;
            half4 o = (half4){0, 0, 0, 0};   // outside the image: m.cv2's zero padding
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) o = silu_pack4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            *reinterpret_cast<half4 *>(s_t + mpx[i] * BPS + ch3 * 2) = o;
        }
    }
    __syncthreads();

    // ---- 2: m.cv2 (3x3, 64 -> 64, SiLU) [+ shortcut] on the tile: pixel tiles ph, ph + 2 ----
    {
        f32x4 acc[2];
        int base[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int mm = (ph + 2 * i) * 16 + r, ly = mm >> 3, lx = mm & 7;
            base[i] = (ly * B2W + lx) * BPS + g * 16;
            acc[i] = b2;
        }
        half8 Bf[3][2];   // (two pixel tiles per wave: two k-steps of fragments in flight ahead of the MFMAs)
        auto frag = [&](int s, half8 (&B)[2]) {
            const int c = s / 9, tap = s - c * 9, kh = tap / 3, kw = tap - kh * 3;
            const int off = (kh * B2W + kw) * BPS + c * 64;
#pragma unroll
            for (int i = 0; i < 2; i++) B[i] = *reinterpret_cast<const half8 *>(s_t + base[i] + off);
        };
        frag(0, Bf[0]);
        frag(1, Bf[1]);
#pragma unroll
        for (int s = 0; s < 18; s++) {
            if (s + 2 < 18) frag(s + 2, Bf[(s + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W2[s], Bf[s % 3][i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int mm = (ph + 2 * i) * 16 + r, ly = mm >> 3, lx = mm & 7;
            half4 o;
            if constexpr (SHORTCUT) {   // the per-layer epilogue adds the shortcut to the ROUNDED fp32 activation (irmv_common.hpp)
                const half4 rv = *reinterpret_cast<const half4 *>(s_in + ((ly + 2) * B1W + lx + 2) * BPS + ch3 * 2);
#pragma unroll
                for (int k = 0; k < 4; k++) o[k] = silu_add_res(acc[i][k], (float)rv[k]);
            } else {
                o = silu_pack4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            }
            if constexpr (MODE == 0) {
                const int gy = oy0 + ly, gx = ox0 + lx;
                if (gy < H && gx < W) *reinterpret_cast<half4 *>(a.ynext + ((size_t)(b * H + gy) * W + gx) * a.ynext_ld + ch3) = o;
            } else {
                *reinterpret_cast<half4 *>(s_yn + mm * BPS + ch3 * 2) = o;
            }
        }
    }
    if constexpr (MODE == 0) return;
    __syncthreads();

    // ---- 3: cv2 (1x1 over the concat -> 128, SiLU) -> block output: output tile `wave`, all four pixel tiles ----
    if constexpr (MODE == 1) {
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = b3;
        half8 Bf[2][4];
        auto frag = [&](int ks, half8 (&B)[4]) {
            const int seg = ks >> 1, hoff = (ks & 1) * 64 + g * 16;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int mm = i * 16 + r, ly = mm >> 3, lx = mm & 7;
                const uint8_t *p = seg < NEXTRA ? s_y0 + seg * (B3N * BPS) + mm * BPS
                                                : (seg == NEXTRA ? s_in + ((ly + 2) * B1W + lx + 2) * BPS : s_yn + mm * BPS);
                B[i] = *reinterpret_cast<const half8 *>(p + hoff);
            }
        };
        frag(0, Bf[0]);
#pragma unroll
        for (int ks = 0; ks < KS2; ks++) {
            if (ks + 1 < KS2) frag(ks + 1, Bf[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W3[ks], Bf[ks & 1][i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int mm = i * 16 + r, ly = mm >> 3, lx = mm & 7;
            const int gy = oy0 + ly, gx = ox0 + lx;
            const half4 o = silu_pack4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            if (gy < H && gx < W) *reinterpret_cast<half4 *>(a.out + ((size_t)(b * H + gy) * W + gx) * a.out_ld + ch4) = o;
        }
    }
}

size_t bneck64_lds_bytes(int mode, int ks2) { return (size_t)(B1N + B2N + B3N) * BPS + (mode == 1 ? (size_t)(ks2 / 2 - 2) * B3N * BPS : 0); }

static std::mutex g_bneck_attr_mu;

bool launch_bneck64(int mode, int ks2, bool shortcut, const BneckArgs &a, int batch, hipStream_t s)
{
    const dim3 grid(a.tiles_x * a.tiles_y * batch), block(512);
    const size_t lds = bneck64_lds_bytes(mode, ks2);
#define IRMV_BNECK(MODE_, KS_, SC_)                                                                                 \
    if (mode == MODE_ && (MODE_ == 0 || ks2 == KS_) && shortcut == SC_) {                                            \
        static unsigned long long attr_done = 0;                                                                     \
        int dev = 0; (void)hipGetDevice(&dev);                                                                       \
        {   /* per device, once, and nobody launches before the limit is raised */                                   \
            std::lock_guard<std::mutex> lk(g_bneck_attr_mu);                                                         \
            if (!(attr_done & (1ull << (dev & 63)))) {                                                               \
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bneck64_kernel<MODE_, KS_, SC_>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
                attr_done |= 1ull << (dev & 63);                                                                     \
            }                                                                                                        \
        }                                                                                                            \
        hipLaunchKernelGGL((bneck64_kernel<MODE_, KS_, SC_>), grid, block, lds, s, a);                               \
        return true;                                                                                                 \
    }
    IRMV_BNECK(0, 6, true) IRMV_BNECK(0, 6, false)
    IRMV_BNECK(1, 6, true) IRMV_BNECK(1, 6, false) IRMV_BNECK(1, 8, true) IRMV_BNECK(1, 8, false)
#undef IRMV_BNECK
    return false;
}

}  // namespace irmv
