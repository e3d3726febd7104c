// model.2 (the first C2f block, 160 x 160 at a 640 net) as ONE kernel for gfx950.
//
// Inside the reference's TensorRT plan (src/yolo_engine.cpp:105) this is cv1 (1x1, 32 -> 32), a
// Bottleneck of two 3x3 16 -> 16 convs with shortcut, and cv2 (1x1 over the 48-channel concat,
// -> 32).  As four kernels it moves 11.5 MB per frame for 0.37 GFLOP: the most bandwidth-starved
// block of the network.  Here a workgroup owns a 16 x 16 pixel tile, loads the block input once
// with a 2-pixel halo (20 x 20 x 32 ch) and keeps y0 / y1 / the bottleneck intermediate / y2 in
// LDS; only the block output goes back to HBM: 1.64 x 1.56 (halo) MB read + 1.64 MB written.
//
// Every intermediate is rounded to fp16 exactly where the four-kernel path stores it and each
// MFMA sees the same operands in the same K order (direct-family packing: 1x1 convs walk
// channels in steps of 32, the Cin = 16 3x3 convs walk tap pairs), so the result is
// bit-identical to the unfused layers (tests/test_gpu_engine.py::test_fused_kernels_are_bitwise_identical).
#include "irmv_common.hpp"

#include <cstdlib>
#include <mutex>

namespace irmv {

namespace {
constexpr int T = kC2fTile;               // output tile (pixels per side)
constexpr int XW = T + 4, XN = XW * XW;   // block input / y1 region (halo 2)
constexpr int TW = T + 2, TN = TW * TW;   // bottleneck intermediate region (halo 1)
constexpr int HP = 32;                    // bytes per pixel of the 16-channel planes (16 consecutive pixels x 2 halves hit
                                          // 16 distinct 16-byte slots of a 256-byte window: conflict-free ds_read_b128)

// SiLU on log2 e-scaled accumulators that START at the (scaled) bias, rounding pinned: irmv_common.hpp, "activation scale"
__device__ __forceinline__ f32x4 bias4(const float *b) { return (f32x4){b[0], b[1], b[2], b[3]}; }

// Global-memory B fragments are loaded UNCONDITIONALLY from a clamped (always valid) address and zeroed by a select
// afterwards, as four dwords: behind `if (inside) B = load` the compiler builds the zero / loaded merge per 16-bit
// element and waits for every load where it is issued (vmcnt(0) after each pair), which serialises the round trips the
// prefetch below is there to overlap.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ld16(const half_t *p) { return *reinterpret_cast<const u32x4 *>(p); }
__device__ __forceinline__ u32x4 keep_if(bool c, u32x4 v) { return c ? v : (u32x4){0u, 0u, 0u, 0u}; }
__device__ __forceinline__ half8 as_h8(u32x4 v) { return __builtin_bit_cast(half8, v); }

// MFMA tiles of an RH x RW pixel region (RW = 16 + 2 or 16 + 4): tile t < RH is columns 0 .. 15 of row t, the tiles after
// those walk the remaining RW - 16 columns top to bottom, 16 pixels each.  Same tile count as a flat walk in runs of 16, but
// a tile's row is the (wave-uniform) tile index: no division by the region width, and the 16 LDS reads of a fragment stay
// inside one row of the plane (a flat run wraps rows: that was a third of these kernels' LDS bank conflicts).
template <int RH, int RW>
__device__ __forceinline__ bool region_tile_px(int t, int r, int &ly, int &lx)
{
    constexpr int LC = RW - 16;
    static_assert(LC == 2 || LC == 4, "16 columns + a power of two");
    if (t < RH) { ly = t; lx = r; return true; }
    const int q = (t - RH) * 16 + r;
    const bool ok = q < LC * RH;
    ly = ok ? q / LC : 0;
    lx = ok ? 16 + (q & (LC - 1)) : 0;
    return ok;
}
}  // namespace

__global__ __launch_bounds__(256) void c2f2_kernel(C2fArgs a, int batch, int xcd)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_y0[T * T * HP];
    __shared__ __attribute__((aligned(16))) uint8_t s_y1[XN * HP];
    __shared__ __attribute__((aligned(16))) uint8_t s_t[TN * HP];
    __shared__ __attribute__((aligned(16))) uint8_t s_y2[T * T * HP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    int tile_id, b;
    tile_image(blockIdx.x, a.tiles * a.tiles, batch, xcd, tile_id, b);
    const int tyi = tile_id / a.tiles, txi = tile_id - tyi * a.tiles;
    const int oy0 = tyi * T, ox0 = txi * T;
    const int S = a.S;
    const half8 zero8 = (half8){0, 0, 0, 0, 0, 0, 0, 0};

    // ---- weights (16 fragments) into registers, each a phase or more before its first use: cv1 and m.0.cv1 now (the loads
    // overlap the input loads), m.0.cv2 and cv2 behind the first barrier.  All sixteen from the start cost a wave per SIMD.
    const half8 *w;
    w = reinterpret_cast<const half8 *>(a.w_cv1) + lane;
    const half8 Wc1[2] = {w[0], w[64]};
    half8 Wm1[5], Wm2[5];
    w = reinterpret_cast<const half8 *>(a.w_m1) + lane;
#pragma unroll
    for (int ks = 0; ks < 5; ks++) Wm1[ks] = w[ks * 64];
    // biases: cv1's into registers with the weights; the later layers' -> LDS (visible after the first barrier).  Fetched
    // from global memory at the start of its phase each would expose a memory round trip behind a barrier.
    float bias_c1[8];
#pragma unroll
    for (int i = 0; i < 8; i++) bias_c1[i] = a.b_cv1[g * 8 + i];
    __shared__ float s_bias[16 + 16 + 32];
    if (tid < 16) { s_bias[tid] = a.b_m1[tid]; s_bias[16 + tid] = a.b_m2[tid]; }
    else if (tid < 48) s_bias[16 + tid] = a.b_cv2[tid - 16];

    // ---- 1: cv1 (1x1, 32 -> 32, SiLU) on the 20 x 20 region -> y0 (centre only), y1 (whole region) ----
    {
        const float (&bias)[8] = bias_c1;
        // The block input is read exactly once per workgroup, so its B fragments come straight from memory -- ALL of this
        // wave's tiles at once: loaded tile by tile, each of the 6 - 7 tiles exposed its own memory round trip.
        // Round 4: tiles 0 .. XW - 1 are ROWS of the region (row = the wave-uniform tile index, column = the lane): the row
        // part of an address is scalar, the column part a per-lane constant; addresses are 32-bit byte offsets from the
        // (scalar) input pointer; a pixel outside the image reads a clamped pixel and its OUTPUT is forced to zero below
        // (y1's zero padding), so the loaded fragment needs no select.
        constexpr int NROWT = XW / 4, NCOLT = (XN / 16 - XW + 3) / 4;      // per wave: 5 row tiles, then 1 or 2 column tiles (XN / 16 - XW = 5 of them)
        static_assert(XW % 4 == 0 && XW - 16 == 4 && NCOLT == 2, "20 row tiles + 5 column tiles");
        const char *xbase = reinterpret_cast<const char *>(a.x);
        const uint32_t img = (uint32_t)b * (uint32_t)(S * S);
        auto px_off = [&](int gy, int gx) -> uint32_t {                    // byte offset of the (clamped) pixel, this lane's 8 channels
            const int gyc = gy < 0 ? 0 : (gy >= S ? S - 1 : gy), gxc = gx < 0 ? 0 : (gx >= S ? S - 1 : gx);
            return ((img + (uint32_t)(gyc * S + gxc)) * (uint32_t)a.x_ld + 8u * g) * 2u;
        };
        const int gx_r = ox0 - 2 + r;
        const bool in_x = (unsigned)gx_r < (unsigned)S;
        int ly_c[NCOLT], lx_c[NCOLT];
        bool has_c[NCOLT];
#pragma unroll
        for (int k = 0; k < NCOLT; k++) {                                   // column tile XW + wave + 4 k: pixel q of the 4-wide strip, top to bottom
            const int tc = wave + 4 * k, q = tc * 16 + r;
            has_c[k] = tc < XN / 16 - XW;                                   // (wave-uniform; XN / 16 - XW tiles of 16 cover the strip exactly)
            ly_c[k] = has_c[k] ? q >> 2 : 0;
            lx_c[k] = 16 + (q & 3);
        }
        u32x4 Bq[NROWT + NCOLT];
#pragma unroll
        for (int i = 0; i < NROWT; i++) Bq[i] = *reinterpret_cast<const u32x4 *>(xbase + px_off(oy0 - 2 + wave + 4 * i, gx_r));
#pragma unroll
        for (int k = 0; k < NCOLT; k++) Bq[NROWT + k] = *reinterpret_cast<const u32x4 *>(xbase + px_off(oy0 - 2 + ly_c[k], ox0 - 2 + lx_c[k]));
        auto run_tile = [&](u32x4 bq, int ly, int lx, bool inside) {
            const int m = ly * XW + lx;
            f32x4 acc0 = bias4(bias), acc1 = bias4(bias + 4);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc1[0], as_h8(bq), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc1[1], as_h8(bq), acc1, 0, 0, 0);
            half8 o = zero8;   // outside the image y1 is the bottleneck's zero padding
            if (inside) o = silu_pack8(acc0[0], acc0[1], acc0[2], acc0[3], acc1[0], acc1[1], acc1[2], acc1[3]);
            if (g >= 2) {
                *reinterpret_cast<half8 *>(s_y1 + m * HP + (g - 2) * 16) = o;
            } else if ((unsigned)(ly - 2) < (unsigned)T && (unsigned)(lx - 2) < (unsigned)T) {
                *reinterpret_cast<half8 *>(s_y0 + ((ly - 2) * T + (lx - 2)) * HP + g * 16) = o;
            }
        };
#pragma unroll
        for (int i = 0; i < NROWT; i++) {
            const int t = wave + 4 * i, gy = oy0 - 2 + t;                    // wave-uniform
            run_tile(Bq[i], t, r, in_x && (unsigned)gy < (unsigned)S);
        }
#pragma unroll
        for (int k = 0; k < NCOLT; k++) {
            if (!has_c[k]) break;
            const int gy = oy0 - 2 + ly_c[k], gx = ox0 - 2 + lx_c[k];
            run_tile(Bq[NROWT + k], ly_c[k], lx_c[k], (unsigned)gy < (unsigned)S && (unsigned)gx < (unsigned)S);
        }
    }
    __syncthreads();

    w = reinterpret_cast<const half8 *>(a.w_m2) + lane;
#pragma unroll
    for (int ks = 0; ks < 5; ks++) Wm2[ks] = w[ks * 64];
    w = reinterpret_cast<const half8 *>(a.w_cv2) + lane;
    const half8 Wc2[2][2] = {{w[0], w[64]}, {w[128], w[192]}};   // [tile][k-step]

    // ---- 2: m.0.cv1 (3x3, 16 -> 16, SiLU) on the 18 x 18 region; Cin = 16: one k-step spans two taps ----
    {
        float bias[4];
#pragma unroll
        for (int i = 0; i < 4; i++) bias[i] = s_bias[g * 4 + i];
        int toff[5];   // byte offset of this lane's tap / channel half inside y1, relative to the pixel (-1 = no tap)
#pragma unroll
        for (int ks = 0; ks < 5; ks++) {
            const int tap = 2 * ks + (g >> 1), kh = tap / 3, kw = tap - kh * 3;
            toff[ks] = tap < 9 ? (kh * XW + kw) * HP + 16 * (g & 1) : -1;
        }
        for (int t = wave; t < (TN + 15) / 16; t += 4) {
            int ly, lx;
            const bool mv = region_tile_px<TW, TW>(t, r, ly, lx);
            const int m = ly * TW + lx;
            const uint8_t *base = s_y1 + (ly * XW + lx) * HP;
            f32x4 acc = bias4(bias);
#pragma unroll
            for (int ks = 0; ks < 5; ks++) {
                half8 B = zero8;
                if (toff[ks] >= 0) B = *reinterpret_cast<const half8 *>(base + toff[ks]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wm1[ks], B, acc, 0, 0, 0);
            }
            if (mv) {
                const int gy = oy0 - 1 + ly, gx = ox0 - 1 + lx;
                half4 o = (half4){0, 0, 0, 0};
                if ((unsigned)gy < (unsigned)S && (unsigned)gx < (unsigned)S) o = silu_pack4(acc[0], acc[1], acc[2], acc[3]);
                *reinterpret_cast<half4 *>(s_t + m * HP + g * 8) = o;
            }
        }
    }
    __syncthreads();

    // ---- 3: m.0.cv2 (3x3, 16 -> 16, SiLU) + shortcut y1 -> y2 on the 16 x 16 tile ----
    {
        float bias[4];
#pragma unroll
        for (int i = 0; i < 4; i++) bias[i] = s_bias[16 + g * 4 + i];
        int toff[5];
#pragma unroll
        for (int ks = 0; ks < 5; ks++) {
            const int tap = 2 * ks + (g >> 1), kh = tap / 3, kw = tap - kh * 3;
            toff[ks] = tap < 9 ? (kh * TW + kw) * HP + 16 * (g & 1) : -1;
        }
        for (int t = wave; t < T * T / 16; t += 4) {
            const int m = t * 16 + r;
            const int ly = m / T, lx = m - ly * T;
            const uint8_t *base = s_t + (ly * TW + lx) * HP;
            f32x4 acc = bias4(bias);
#pragma unroll
            for (int ks = 0; ks < 5; ks++) {
                half8 B = zero8;
                if (toff[ks] >= 0) B = *reinterpret_cast<const half8 *>(base + toff[ks]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wm2[ks], B, acc, 0, 0, 0);
            }
            const half4 rv = *reinterpret_cast<const half4 *>(s_y1 + ((ly + 2) * XW + lx + 2) * HP + g * 8);
            half4 o;
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = silu_add_res(acc[i], (float)rv[i]);   // the shortcut is added to the ROUNDED product, as in the per-layer kernel
            *reinterpret_cast<half4 *>(s_y2 + m * HP + g * 8) = o;
        }
    }
    __syncthreads();

    // ---- 4: cv2 (1x1 over [y0 | y1 | y2] = 48 channels -> 32, SiLU) -> HBM ----
    {
        float bias[8];
#pragma unroll
        for (int i = 0; i < 8; i++) bias[i] = s_bias[32 + g * 8 + i];
        half_t *out = a.out + (size_t)b * S * S * a.out_ld;
        for (int t = wave; t < T * T / 16; t += 4) {
            const int m = t * 16 + r;
            const int ly = m / T, lx = m - ly * T;
            // k-step 0: channels 8g..8g+7 of the concat: y0 for g < 2, y1 for g >= 2; k-step 1: y2 for g < 2, nothing above 48
            const half8 B0 = g < 2 ? *reinterpret_cast<const half8 *>(s_y0 + m * HP + g * 16)
                                   : *reinterpret_cast<const half8 *>(s_y1 + ((ly + 2) * XW + lx + 2) * HP + (g - 2) * 16);
            half8 B1 = zero8;
            if (g < 2) B1 = *reinterpret_cast<const half8 *>(s_y2 + m * HP + g * 16);
            f32x4 acc0 = bias4(bias), acc1 = bias4(bias + 4);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc2[0][0], B0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc2[1][0], B0, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc2[0][1], B1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc2[1][1], B1, acc1, 0, 0, 0);
            const int gy = oy0 + ly, gx = ox0 + lx;
            if (gy < S && gx < S) {
                const half8 o = silu_pack8(acc0[0], acc0[1], acc0[2], acc0[3], acc1[0], acc1[1], acc1[2], acc1[3]);
                *reinterpret_cast<half8 *>(out + ((size_t)gy * S + gx) * a.out_ld + g * 8) = o;
            }
        }
    }
}

static int c2f_xcd_order() { return xcd_image_order(); }

void launch_c2f2(const C2fArgs &a, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(c2f2_kernel, dim3(a.tiles * a.tiles * batch), dim3(256), 0, s, a, batch, c2f_xcd_order());
}


// =====================================================================================================================
// C2f blocks with a 32-channel hidden width (model.4 and model.15 at a 640 net: 80 x 80) as fused kernels.
//
// Per layer these blocks are 4 (n = 1) or 6 (n = 2) launches of small-K convolutions -- K = 288 for the 3x3s, 64..192
// for the 1x1s -- whose time goes into launch gaps, LDS staging and tensor round trips, not into the matrix cores.
// Here a workgroup owns an 8 x 16 pixel tile and walks the block's layers with every intermediate in LDS:
//
//   mode AB (n = 1)          cv1 on the tile + 2-pixel halo (12 x 20), straight from the block input(s) in memory (two K
//                            segments, one optionally at half resolution = the neck's upsample + concat) -> y0 (tile), y1
//                            (halo 2) -> m.0.cv1 on 10 x 18 -> m.0.cv2 (+ y1 if shortcut) on the tile -> cv2 over
//                            [y0 | y1 | y2] -> block output.  One launch instead of four.
//   mode A + mode B (n = 2)  A: cv1 -> m.0 -> y0, y1, y2 into the block's concat buffer; B: y2 with halo from there ->
//                            m.1 -> y3 (LDS only) -> cv2 over [y0 | y1 | y2 | y3].  Two launches instead of six; one fused
//                            kernel would need a 4-pixel halo and recompute 1.7 x the MFMAs.
//
// Planes are [pixel][32 ch] with the 96-byte pixel stride of the LDS conv family (conflict-free ds_read_b128 over 16
// consecutive pixels); weight fragments of the running layer live in registers, fetched per phase from L1/L2 (direct-
// family packing).  Rounding points and K order are those of the per-layer kernels (1x1: channels in steps of 32,
// segment 0 then segment 1; 3x3 with Cin = 32: taps 0..8), so results are bit-identical to them
// (tests/test_gpu_engine.py::test_fused_kernels_are_bitwise_identical).
// =====================================================================================================================
namespace {
constexpr int FH = kC2f32TileH, FW = kC2f32TileW;       // output tile
constexpr int R1H = FH + 4, R1W = FW + 4, R1N = R1H * R1W;   // y1 / bottleneck-input region (halo 2): 12 x 20 = 240 = 15 tiles
constexpr int R2H = FH + 2, R2W = FW + 2, R2N = R2H * R2W;   // bottleneck intermediate (halo 1): 10 x 18 = 180
constexpr int R3N = FH * FW;                                 // 128 = 8 tiles
constexpr int PS = 96;                                       // bytes per pixel of a 32-channel plane
static_assert(R1N % 16 == 0 && R3N % 16 == 0, "tile regions are whole MFMA tiles");
}  // namespace

// MODE 0 = AB, 1 = A, 2 = B.  KS1 = k-steps of cv1 (Cin / 32).
template <int MODE, int KS1, bool SHORTCUT>
__global__ __launch_bounds__(256) void c2f32_kernel(C2f32Args a, int batch, int xcd)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *s_in = smem;                         // y1 region (modes AB, A) / y_prev region (mode B)
    uint8_t *s_t = s_in + R1N * PS;               // bottleneck intermediate
    uint8_t *s_yn = s_t + R2N * PS;               // bottleneck output on the tile (modes AB, B)
    uint8_t *s_y0 = s_yn + R3N * PS;              // y0 on the tile (mode AB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    int tile_id, b;
    tile_image(blockIdx.x, a.tiles_x * a.tiles_y, batch, xcd, tile_id, b);
    const int tyi = tile_id / a.tiles_x, txi = tile_id - tyi * a.tiles_x;
    const int oy0 = tyi * FH, ox0 = txi * FW;
    const int H = a.H, W = a.W;
    const half8 zero8 = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    half_t *cat = a.cat + (size_t)b * H * W * a.cat_ld;   // the block's concat buffer [H][W][cat_ld]: y0 | y1 | y2 [| y3]
    // biases of the later phases -> LDS now (visible after the first barrier): fetched from global memory at the start of
    // its phase each would expose a memory round trip right behind a barrier
    __shared__ float s_bias[32 + 32 + 64];
    if (tid < 32) { s_bias[tid] = a.b_m1[tid]; s_bias[32 + tid] = a.b_m2[tid]; }
    if (MODE != 1 && tid < 64) s_bias[64 + tid] = a.b_cv2[tid];

    u32x4 Y01[2][2];   // mode B only
    static_assert(R3N / 16 == 8, "two output tiles per wave");
    if constexpr (MODE != 2) {
        // ---- 1: cv1 (1x1, Cin -> 64 = y0 | y1, SiLU) on the 12 x 20 region, B fragments straight from memory ----
        half8 W1[4][KS1];
        {
            const half8 *w = reinterpret_cast<const half8 *>(a.w_cv1) + lane;
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
#pragma unroll
                for (int ks = 0; ks < KS1; ks++) W1[nt][ks] = w[(size_t)(nt * KS1 + ks) * 64];
        }
        float bias[16];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) bias[u * 8 + i] = a.b_cv1[u * 32 + g * 8 + i];
        const int H0 = H >> a.s0.shift, W0 = W >> a.s0.shift, H1 = H >> a.s1.shift, W1s = W >> a.s1.shift;
        // The block input is read once per workgroup, straight into B fragments; the next tile's loads are in flight under
        // this tile's MFMAs and SiLU epilogue (a wave walks ~4 tiles: without this each would expose a full memory round trip).
        // Round 4, instruction diet (this phase was 63 % of the kernel's vector instructions, 175 of its 240 per tile index
        // arithmetic and selects):
        //  * tiles 0 .. R1H - 1 are ROWS of the region (row = the wave-uniform tile index, column = the lane): the row part of
        //    every address is scalar, the column part a per-lane constant computed once; only the last tiles (the region's
        //    four extra columns, walked top to bottom) keep per-lane rows;
        //  * addresses are a 32-bit byte offset from the segment's (scalar) base pointer: one vector add per fragment instead
        //    of a 64-bit select + add; which segment a k-step reads is wave-uniform (segment widths are multiples of 32);
        //  * no select on the loaded fragments: a pixel outside the image reads a clamped (valid, finite) pixel and its
        //    OUTPUT is forced to zero below -- y1's zero padding -- whatever the MFMAs made of it.
        constexpr int NT1 = R1N / 16, NROWT = R1H / 4, NCOLT = NT1 - R1H;   // 15 tiles: 12 rows (three per wave) + 3 column tiles (waves 0 .. 2)
        static_assert(R1H % 4 == 0 && NCOLT >= 0 && NCOLT <= 4 && R1W - 16 == 4, "three row tiles per wave, at most one column tile");
        const char *base0 = reinterpret_cast<const char *>(a.s0.p), *base1 = reinterpret_cast<const char *>(a.s1.p - a.s0.C);   // (k-step ks of segment 1 sits at channel 32 ks - s0.C)
        const uint32_t img0 = (uint32_t)b * (uint32_t)(H0 * W0), img1 = (uint32_t)b * (uint32_t)(H1 * W1s);
        auto px_off = [&](int gyc, int gxc, uint32_t &o0, uint32_t &o1) {           // byte offsets of pixel (gyc, gxc), this lane's 8 channels
            o0 = ((img0 + (uint32_t)((gyc >> a.s0.shift) * W0 + (gxc >> a.s0.shift))) * (uint32_t)a.s0.ld + 8u * g) * 2u;
            o1 = ((img1 + (uint32_t)((gyc >> a.s1.shift) * W1s + (gxc >> a.s1.shift))) * (uint32_t)a.s1.ld + 8u * g) * 2u;
        };
        auto issue = [&](uint32_t o0, uint32_t o1, u32x4 (&B)[KS1]) {
#pragma unroll
            for (int ks = 0; ks < KS1; ks++) {
                const bool seg0 = ks * 32 < a.s0.C;                                 // wave-uniform
                B[ks] = *reinterpret_cast<const u32x4 *>((seg0 ? base0 : base1) + ((seg0 ? o0 : o1) + (uint32_t)(ks * 64)));
            }
        };
        // row tiles: column = lane r, clamped into the image; the row is the tile's
        const int gx_r = ox0 - 2 + r;
        const bool in_x = (unsigned)gx_r < (unsigned)W;
        const int gxc_r = gx_r < 0 ? 0 : (gx_r >= W ? W - 1 : gx_r);
        auto row_off = [&](int t, uint32_t &o0, uint32_t &o1) {
            const int gy = oy0 - 2 + t, gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);  // scalar
            px_off(gyc, gxc_r, o0, o1);
        };
        // the column tile of this wave (waves 0 .. NCOLT - 1): pixel q = 16 wave + r of the 4-wide strip, top to bottom
        const int q_c = 16 * wave + r, ly_c = q_c >> 2, lx_c = 16 + (q_c & 3);
        const bool has_col = wave < NCOLT;
        // one tile: fragments B (already loaded), region pixel m = ly * R1W + lx, image pixel (gy, gx)
        auto run_tile = [&](const u32x4 (&Bq)[KS1], int ly, int lx, bool inside, bool want_y0) {
            const int m = ly * R1W + lx;
            const int gy = oy0 - 2 + ly, gx = ox0 - 2 + lx;
            f32x4 acc[4];   // tile 2 u + h starts at bias[u * 8 + 4 h ..]
#pragma unroll
            for (int nt = 0; nt < 4; nt++) acc[nt] = bias4(bias + (nt >> 1) * 8 + (nt & 1) * 4);
            // y0 (output tiles 0, 1) is needed on the 8 x 16 tile only: the row tiles of the region's first and last two rows
            // (one per wave) compute y1 alone -- half the MFMAs and half the SiLUs of those tiles (the wave-uniform branch
            // costs nothing; nothing of y0 was stored for them anyway)
            if (want_y0) {
#pragma unroll
                for (int ks = 0; ks < KS1; ks++)
#pragma unroll
                    for (int nt = 0; nt < 4; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W1[nt][ks], as_h8(Bq[ks]), acc[nt], 0, 0, 0);
            } else {
#pragma unroll
                for (int ks = 0; ks < KS1; ks++)
#pragma unroll
                    for (int nt = 2; nt < 4; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W1[nt][ks], as_h8(Bq[ks]), acc[nt], 0, 0, 0);
            }
            half8 o[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                o[u] = zero8;   // outside the image y1 is the bottleneck's zero padding
                if (inside && (u == 1 || want_y0)) {
                    o[u] = silu_pack8(acc[2 * u][0], acc[2 * u][1], acc[2 * u][2], acc[2 * u][3], acc[2 * u + 1][0], acc[2 * u + 1][1], acc[2 * u + 1][2], acc[2 * u + 1][3]);
                }
            }
            *reinterpret_cast<half8 *>(s_in + m * PS + g * 16) = o[1];
            const int cy = ly - 2, cx = lx - 2;
            if (inside && (unsigned)cy < (unsigned)FH && (unsigned)cx < (unsigned)FW) {
                if constexpr (MODE == 0) {
                    *reinterpret_cast<half8 *>(s_y0 + (cy * FW + cx) * PS + g * 16) = o[0];
                } else {
                    half_t *qd = cat + ((size_t)gy * W + gx) * a.cat_ld + g * 8;
                    *reinterpret_cast<half8 *>(qd) = o[0];
                    *reinterpret_cast<half8 *>(qd + 32) = o[1];
                }
            }
        };
        u32x4 Bn[KS1], Bc[KS1];
        {
            uint32_t o0, o1;
            row_off(wave, o0, o1);
            issue(o0, o1, Bn);
        }
#pragma unroll
        for (int i = 0; i < NROWT; i++) {
            const int t = wave + 4 * i;                                             // wave-uniform row of the region
#pragma unroll
            for (int ks = 0; ks < KS1; ks++) Bc[ks] = Bn[ks];
            {   // the next tile's loads, before this tile's arithmetic
                uint32_t o0, o1;
                if (i + 1 < NROWT) {
                    row_off(t + 4, o0, o1);
                    issue(o0, o1, Bn);
                } else if (has_col) {
                    const int gy = oy0 - 2 + ly_c, gx = ox0 - 2 + lx_c;
                    px_off(gy < 0 ? 0 : (gy >= H ? H - 1 : gy), gx < 0 ? 0 : (gx >= W ? W - 1 : gx), o0, o1);
                    issue(o0, o1, Bn);
                }
            }
            const int gy = oy0 - 2 + t;
            run_tile(Bc, t, r, in_x && (unsigned)gy < (unsigned)H, t >= 2 && t < R1H - 2);
        }
        if (has_col) {
            const int gy = oy0 - 2 + ly_c, gx = ox0 - 2 + lx_c;
            run_tile(Bn, ly_c, lx_c, (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W, true);
        }
    } else {
        // y0, y1 of this wave's two output tiles (operands of cv2, phase 4) are fetched NOW: they depend on nothing this
        // kernel computes, and loaded where they are used each tile exposed a memory round trip
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int m = (wave + 4 * i) * 16 + r;
            const int ly = m / FW, lx = m - ly * FW;
            const int gy = oy0 + ly, gx = ox0 + lx;
            const bool in = gy < H && gx < W;
            const half_t *q = cat + ((size_t)(in ? gy : 0) * W + (in ? gx : 0)) * a.cat_ld + g * 8;
            Y01[i][0] = ld16(q);          // zeroed (keep_if) in phase 4
            Y01[i][1] = ld16(q + 32);
        }
        // ---- 1': the previous bottleneck's output (slice y_prev of the concat buffer) with a 2-pixel halo -> LDS ----
        const half_t *src = cat + a.prev_coff;
        constexpr int NP = (R1N * 4 + 255) / 256;   // 16-byte pieces per thread: all loads issued before the first LDS store
        u32x4 v[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int e = tid + i * 256;
            const int m = e >> 2, q = e & 3;
            const int ly = m / R1W, lx = m - ly * R1W;
            const int gy = oy0 - 2 + ly, gx = ox0 - 2 + lx;
            const bool in = e < R1N * 4 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            v[i] = ld16(src + ((size_t)(in ? gy : 0) * W + (in ? gx : 0)) * a.cat_ld + q * 8);
        }
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int e = tid + i * 256;
            const int m = e >> 2;
            const int ly = m / R1W, lx = m - ly * R1W;
            const int gy = oy0 - 2 + ly, gx = ox0 - 2 + lx;
            if (e < R1N * 4) *reinterpret_cast<u32x4 *>(s_in + m * PS + (e & 3) * 16) = keep_if((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W, v[i]);
        }
    }
    __syncthreads();

    // ---- 2: m.cv1 (3x3, 32 -> 32, SiLU) on the 10 x 18 region ----
    {
        half8 Wm[2][9];
        const half8 *w = reinterpret_cast<const half8 *>(a.w_m1) + lane;
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int tap = 0; tap < 9; tap++) Wm[nt][tap] = w[(size_t)(nt * 9 + tap) * 64];
        float bias[8];
#pragma unroll
        for (int i = 0; i < 8; i++) bias[i] = s_bias[g * 8 + i];
        for (int t = wave; t < (R2N + 15) / 16; t += 4) {
            int ly, lx;
            const bool mv = region_tile_px<R2H, R2W>(t, r, ly, lx);
            const int m = ly * R2W + lx;
            const uint8_t *base = s_in + (ly * R1W + lx) * PS + g * 16;
            f32x4 acc0 = bias4(bias), acc1 = bias4(bias + 4);
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const half8 B = *reinterpret_cast<const half8 *>(base + ((tap / 3) * R1W + (tap % 3)) * PS);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wm[0][tap], B, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wm[1][tap], B, acc1, 0, 0, 0);
            }
            if (mv) {
                const int gy = oy0 - 1 + ly, gx = ox0 - 1 + lx;
                half8 o = zero8;   // outside the image: the next conv's zero padding
                if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) o = silu_pack8(acc0[0], acc0[1], acc0[2], acc0[3], acc1[0], acc1[1], acc1[2], acc1[3]);
                *reinterpret_cast<half8 *>(s_t + m * PS + g * 16) = o;
            }
        }
    }
    __syncthreads();

    // ---- 3: m.cv2 (3x3, 32 -> 32, SiLU) [+ shortcut] on the tile ----
    {
        half8 Wm[2][9];
        const half8 *w = reinterpret_cast<const half8 *>(a.w_m2) + lane;
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int tap = 0; tap < 9; tap++) Wm[nt][tap] = w[(size_t)(nt * 9 + tap) * 64];
        float bias[8];
#pragma unroll
        for (int i = 0; i < 8; i++) bias[i] = s_bias[32 + g * 8 + i];
        for (int t = wave; t < R3N / 16; t += 4) {
            const int m = t * 16 + r;
            const int ly = m / FW, lx = m - ly * FW;
            const uint8_t *base = s_t + (ly * R2W + lx) * PS + g * 16;
            f32x4 acc0 = bias4(bias), acc1 = bias4(bias + 4);
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const half8 B = *reinterpret_cast<const half8 *>(base + ((tap / 3) * R2W + (tap % 3)) * PS);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wm[0][tap], B, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wm[1][tap], B, acc1, 0, 0, 0);
            }
            half8 o;
            half8 rv = zero8;
            if constexpr (SHORTCUT) rv = *reinterpret_cast<const half8 *>(s_in + ((ly + 2) * R1W + lx + 2) * PS + g * 16);
            if constexpr (SHORTCUT) {   // the per-layer epilogue adds the shortcut to the ROUNDED activation
#pragma unroll
                for (int i = 0; i < 8; i++) o[i] = silu_add_res(i < 4 ? acc0[i] : acc1[i - 4], (float)rv[i]);
            } else {
                o = silu_pack8(acc0[0], acc0[1], acc0[2], acc0[3], acc1[0], acc1[1], acc1[2], acc1[3]);
            }
            if constexpr (MODE == 1) {
                const int gy = oy0 + ly, gx = ox0 + lx;
                if (gy < H && gx < W) *reinterpret_cast<half8 *>(cat + ((size_t)gy * W + gx) * a.cat_ld + 64 + g * 8) = o;
            } else {
                *reinterpret_cast<half8 *>(s_yn + m * PS + g * 16) = o;
            }
        }
    }
    if constexpr (MODE == 1) return;
    __syncthreads();

    // ---- 4: cv2 (1x1 over the concat, -> 64, SiLU) -> block output ----
    {
        constexpr int KS2 = MODE == 0 ? 3 : 4;
        half8 W2[4][KS2];
        const half8 *w = reinterpret_cast<const half8 *>(a.w_cv2) + lane;
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int ks = 0; ks < KS2; ks++) W2[nt][ks] = w[(size_t)(nt * KS2 + ks) * 64];
        float bias[16];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) bias[u * 8 + i] = s_bias[64 + u * 32 + g * 8 + i];
        half_t *out = a.out + (size_t)b * H * W * a.out_ld;
        for (int t = wave; t < R3N / 16; t += 4) {
            const int m = t * 16 + r;
            const int ly = m / FW, lx = m - ly * FW;
            const int gy = oy0 + ly, gx = ox0 + lx;
            const bool inside = gy < H && gx < W;
            half8 B[KS2];
            if constexpr (MODE == 0) {
                B[0] = *reinterpret_cast<const half8 *>(s_y0 + m * PS + g * 16);
                B[1] = *reinterpret_cast<const half8 *>(s_in + ((ly + 2) * R1W + lx + 2) * PS + g * 16);
                B[2] = *reinterpret_cast<const half8 *>(s_yn + m * PS + g * 16);
            } else {
                // y0, y1 are read once per pixel: straight from the concat buffer (at kernel start); y2 = the staged slice; y3 = this kernel's
                B[0] = as_h8(keep_if(inside, Y01[(t - wave) >> 2][0]));
                B[1] = as_h8(keep_if(inside, Y01[(t - wave) >> 2][1]));
                B[2] = *reinterpret_cast<const half8 *>(s_in + ((ly + 2) * R1W + lx + 2) * PS + g * 16);
                B[3] = *reinterpret_cast<const half8 *>(s_yn + m * PS + g * 16);
            }
            f32x4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; nt++) acc[nt] = bias4(bias + (nt >> 1) * 8 + (nt & 1) * 4);
#pragma unroll
            for (int ks = 0; ks < KS2; ks++)
#pragma unroll
                for (int nt = 0; nt < 4; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W2[nt][ks], B[ks], acc[nt], 0, 0, 0);
            if (inside) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const half8 o = silu_pack8(acc[2 * u][0], acc[2 * u][1], acc[2 * u][2], acc[2 * u][3], acc[2 * u + 1][0], acc[2 * u + 1][1], acc[2 * u + 1][2], acc[2 * u + 1][3]);
                    *reinterpret_cast<half8 *>(out + ((size_t)gy * W + gx) * a.out_ld + u * 32 + g * 8) = o;
                }
            }
        }
    }
}

static std::mutex g_c2f_attr_mu;

size_t c2f32_lds_bytes(int mode) { return (size_t)(R1N + R2N + (mode == 1 ? 0 : R3N) + (mode == 0 ? R3N : 0)) * PS; }

bool launch_c2f32(int mode, bool shortcut, const C2f32Args &a, int batch, hipStream_t s)
{
    const int ks1 = a.cin1 / 32;
    const dim3 grid(a.tiles_x * a.tiles_y * batch), block(256);
    const int xcd = c2f_xcd_order();
    const size_t lds = c2f32_lds_bytes(mode);
#define IRMV_C2F32(MODE_, KS_, SC_)                                                                               \
    if (mode == MODE_ && (MODE_ == 2 || ks1 == KS_) && shortcut == SC_) {                                          \
        static unsigned long long attr_done = 0;                                                                   \
        int dev = 0; (void)hipGetDevice(&dev);                                                                     \
        {   /* per device, once, and nobody launches before the limit is raised */                                 \
            std::lock_guard<std::mutex> lk(g_c2f_attr_mu);                                                         \
            if (!(attr_done & (1ull << (dev & 63)))) {                                                             \
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(c2f32_kernel<MODE_, KS_, SC_>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
                attr_done |= 1ull << (dev & 63);                                                                   \
            }                                                                                                      \
        }                                                                                                          \
        hipLaunchKernelGGL((c2f32_kernel<MODE_, KS_, SC_>), grid, block, lds, s, a, batch, xcd);                   \
        return true;                                                                                               \
    }
    IRMV_C2F32(0, 2, true) IRMV_C2F32(0, 2, false) IRMV_C2F32(0, 4, false) IRMV_C2F32(0, 6, false) IRMV_C2F32(0, 6, true) IRMV_C2F32(0, 4, true)
    IRMV_C2F32(1, 2, true) IRMV_C2F32(1, 2, false) IRMV_C2F32(1, 4, true) IRMV_C2F32(1, 6, true)
    IRMV_C2F32(2, 1, true) IRMV_C2F32(2, 1, false)
#undef IRMV_C2F32
    return false;
}

}  // namespace irmv
