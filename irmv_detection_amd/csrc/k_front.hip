// Fused network front for gfx950: frame preprocess + model.0.conv + model.1.conv in ONE pass.
//
// Reference path replaced: the four NPP launches of YoloEngine::preprocess
// (src/yolo_engine.cpp:179-200) and the first two convolutions of the TensorRT plan
// (src/yolo_engine.cpp:105).  Run as separate kernels these three steps move
// 3.93 + 3.28 | 3.28 + 3.28 | 3.28 + 1.64 MB per 1280x1024 frame -- they are the
// bandwidth-bound high-resolution part of the network.  Fused, a workgroup owns a
// 4 x 16 tile of model.1's output and keeps everything in between in LDS:
//
//   A  source region of the tile (with the halo of two stride-2 3x3 convs: 19 x 67
//      net-input pixels; ~33 rows x 140 source pixels for 1280 x 1024 -> 640) -> LDS as
//      4-byte pixels; bilinear resample, /255, fp16 NHWC4 into LDS (same fixed-point
//      arithmetic as preprocess_kernel)
//   B  model.0.conv (3 -> 16, s2, SiLU) on MFMA from LDS -> 9 x 33 fp16 pixels in LDS
//      (zero outside the image: they are model.1's padding)
//   C  model.1.conv (16 -> 32, s2, SiLU) on MFMA from LDS -> 4 x 16 x 32 fp16 to HBM
//
// so a frame costs 3.93 MB read + 1.64 MB written.  Every value is rounded to fp16
// exactly where the unfused kernels store their tensors and every MFMA sees the same
// operands in the same order, so the result is bit-identical to the three-kernel path
// (tests/test_gpu_engine.py::test_fused_kernels_are_bitwise_identical).
#include "irmv_common.hpp"

#include <type_traits>

namespace irmv {

namespace {
constexpr int kCoefBits = 11;
constexpr int kCoefOne = 1 << kCoefBits;
constexpr int TX = kFrontTileX;                      // model.1 output tile: TY x TX, TY = 4 (staged source) or 8 (direct tiles)
constexpr int C0W = 2 * TX + 1;                      // model.0 output columns it needs
constexpr int C0HALF = (C0W + 1) / 2;                // columns per parity plane
constexpr int INW = 4 * TX + 3;                      // net-input columns those need
constexpr int INP = INW + 1;                         // row pitch (pixels); the extra column stays zero
}  // namespace

#if IRMV_FSTAMP
__device__ unsigned long long g_front_stamps[65536 * 9];   // probe builds: shader-clock stamps of the phase boundaries, wave 0 of every workgroup
#define FSTAMP(k) do { if (tid == 0) { fst[k] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define FSTAMP(k)
#endif

// TY = 4: 5 waves per SIMD = 96 VGPRs (the occupancy step the kernel sat on before its biases moved to LDS).  TY = 8 (round 3,
// engines whose tiles are all direct): twice the tile on the same four waves -- the halo of the two stride-2 convs costs
// 1.15 x / 1.10 x instead of 1.24 x / 1.16 x the pixels, the per-workgroup prologue is paid half as often, and the direct
// walk fills 93 % of its lane slots instead of 84 %; 38 KB of LDS -> 4 workgroups per CU.
template <int TY>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TY == 4 ? 5 : 4))) void front_kernel(FrontArgs a, int batch, int xcd)
{
    constexpr int C0H = 2 * TY + 1;                      // model.0 output rows the tile needs
    constexpr int INH = 4 * TY + 3;                      // net-input rows those need
    constexpr int MTC = TY / 4;                          // model.1 output rows per wave
    extern __shared__ __attribute__((aligned(16))) uint8_t s_stage[];   // a.stage_bytes: source region, later model.0's tile
    __shared__ __attribute__((aligned(16))) half4 s_in[INH * INP];
    __shared__ uint32_t s_tx[INW], s_ty[INH];   // packed region-relative taps
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
#if IRMV_FSTAMP
    unsigned long long fst[10];
#endif
    FSTAMP(0);
    int tile_id, b;   // all tiles of a frame on one XCD: neighbouring tiles' shared source rows then meet in that XCD's L2 (irmv_common.hpp)
    tile_image(blockIdx.x, a.tiles_x * a.tiles_y, batch, xcd, tile_id, b);
    const int tyi = tile_id / a.tiles_x, txi = tile_id - tyi * a.tiles_x;
    const int oy0 = tyi * TY, ox0 = txi * TX;
    const int net = a.net, W0 = net >> 1, W1 = net >> 2;
    const int gy0 = 4 * oy0 - 3, gx0 = 4 * ox0 - 3;   // net-input coordinates of s_in[0][0]

    // both biases -> LDS (visible after the staging barrier): fetched from global memory where they are used (start of
    // stage B, epilogue of stage C) each exposes a memory round trip of a ~10 us workgroup; held in registers from here
    // they cost the occupancy step this kernel sits on (measured +10 %).  Loaded now, stored behind the source loads:
    // nothing in front of those loads may wait for memory (phase stamps: 34 % of a workgroup's life was the time before them)
    __shared__ float s_bias[16 + 32];
    float bias_v = 0.f;
    if (tid < 16) bias_v = a.b0[tid];
    else if (tid < 48) bias_v = a.b1[tid - 16];

    // model.1 weights (5 k-steps x 2 tiles) and model.0 weights (2 k-steps) into registers early
    half8 A1[5][2];
    {
        const half8 *wp = reinterpret_cast<const half8 *>(a.w1) + lane;
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int ks = 0; ks < 5; ks++) A1[ks][nt] = wp[(size_t)(nt * 5 + ks) * 64];
    }
    const half8 *wp0 = reinterpret_cast<const half8 *>(a.w0) + lane;
    const half8 A00 = wp0[0], A01 = wp0[64];

    // ---- direct tiles (round 3): columns at exactly 2 : 1.  Every source pixel of a tile then feeds exactly one of its
    // net-input pixels in x, so staging the region in LDS first (A1: 12-byte groups -> 4-byte pixels, a barrier, then A2's
    // reads) is pure overhead: a thread loads the 12 bytes = two aligned source pairs of TWO tap rows straight into
    // registers and blends two adjacent net-input pixels from them.  Same integers as A2's 2 : 1 path (sum of the pair,
    // vertical blend, one rounding shift), so the same bits.  Tiles that contain padding or out-of-image pixels (EDGE) load
    // from clamped addresses and select the constant afterwards.
    const bool tile_inside = gx0 >= max(a.vx0, 0) && gx0 + INW <= min(a.vx1, net) && gy0 >= max(a.vy0, 0) && gy0 + INH <= min(a.vy1, net);
    const bool direct = (a.fastx & 2) != 0;
    if (direct) {
        constexpr int NPAIR = (INW + 1) / 2, NIT = INH * NPAIR;                    // 34 pairs x 19 (35) rows = 646 (1190) items
        constexpr int NTRIP = (NIT + 255) / 256;                                   // three (five) trips
        static_assert(NPAIR == 34, "the walk below steps 256 = 7 rows + 18 pairs");
        // a 12-byte group = source pixels g .. g + 3 (g a multiple of 4) = the pairs of two adjacent columns; the tile's
        // columns are paired from column -de on so that every pair of columns is one such group.  The group of a column
        // that has a source lies inside the source row (pairs are even, sw is a multiple of 4), whatever its neighbour is.
        const int mq = a.fx_i0 + a.fx_step * (gx0 - a.vx0);                       // source pair of the tile's first column
        const int de = a.fx_step > 0 ? ((mq & 3) == 0 ? 0 : 1) : ((mq & 3) == 2 ? 0 : 1);
        const int dg0 = a.fx_step > 0 ? mq - 2 * de : mq + 2 * de - 2;            // group of the first pair of columns
        const uint8_t *srcb = a.src + (size_t)b * a.src_slot_bytes;
        const uint32_t row_bytes = (uint32_t)a.sw * 3u;
        const int sgn4 = a.fx_step > 0 ? 4 : -4;
        auto run_direct = [&](auto edge_c) {
            constexpr bool EDGE = decltype(edge_c)::value;
            int row = tid / NPAIR, p = tid - row * NPAIR;
            int rowv[NTRIP], pv[NTRIP];
            AxisTap tp[NTRIP];
#pragma unroll
            for (int k = 0; k < NTRIP; k++) {
                rowv[k] = row; pv[k] = p;
                int gy = gy0 + min(row, INH - 1);                                   // (the last trip's idle lanes: the last row again)
                if (EDGE) gy = min(max(gy, a.vy0), a.vy1 - 1);                      // a row without a source: any row's tap, the result is replaced
                tp[k] = a.ty[gy];
                p += 256 - 7 * NPAIR; row += 7;
                if (p >= NPAIR) { p -= NPAIR; row++; }
            }
            uint32_t dd[NTRIP][6];
#pragma unroll
            for (int k = 0; k < NTRIP; k++) {
                int gi = dg0 + sgn4 * pv[k];
                if (EDGE) gi = min(max(gi, 0), a.sw - 4);                           // both columns without a source
                const uint32_t xo = (uint32_t)(3 * gi);
                const uint32_t *q0 = reinterpret_cast<const uint32_t *>(srcb + ((uint32_t)tp[k].i0 * row_bytes + xo));   // (one 32-bit offset: `p + a + b` is two 64-bit adds)
                const uint32_t *q1 = reinterpret_cast<const uint32_t *>(srcb + ((uint32_t)tp[k].i1 * row_bytes + xo));
                dd[k][0] = q0[0]; dd[k][1] = q0[1]; dd[k][2] = q0[2];
                dd[k][3] = q1[0]; dd[k][4] = q1[1]; dd[k][5] = q1[2];
            }
            if (tid < 48) s_bias[tid] = bias_v;
            if (tid < INH) s_in[tid * INP + INW] = (half4){0, 0, 0, 0};           // the extra column of every row stays zero
            const float inv255 = 1.0f / 255.0f;
            const uint32_t rnd = 1u << kCoefBits;
            const half_t padv = (half_t)(114.0f / 255.0f);
            // v_perm_b32 selectors (byte i of the result: 0..3 = bytes of the second operand, 4..7 = of the first, 0x0c = zero)
            const uint32_t sel00 = a.swap_rb ? 0x0c000c02u : 0x0c020c00u, sel01 = a.swap_rb ? 0x0c030c05u : 0x0c050c03u;
            const uint32_t sel10 = a.swap_rb ? 0x0c020c04u : 0x0c040c02u, sel11 = a.swap_rb ? 0x0c010c03u : 0x0c030c01u;
#pragma unroll
            for (int k = 0; k < NTRIP; k++) {
                if (k == NTRIP - 1 && tid >= NIT - 256 * (NTRIP - 1)) break;
                const uint32_t wy = (uint32_t)tp[k].w1, wy0 = kCoefOne - wy;
                // bytes of a row's group: pixel 0 = b0 b1 b2, 1 = b3 b4 b5, 2 = b6 b7 b8, 3 = b9 b10 b11; first half = pixels 0 + 1.
                // The byte selectors put the channel that becomes output channel 0 in the low half (swap_rb: uniform, scalar)
                uint32_t rb[2][2], gg[2][2];   // [tap row][half]: first and third output channel's sums side by side, channel 1 sum
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const uint32_t d0 = dd[k][3 * t], d1 = dd[k][3 * t + 1], d2 = dd[k][3 * t + 2];
                    rb[t][0] = __builtin_amdgcn_perm(d0, d0, sel00) + __builtin_amdgcn_perm(d1, d0, sel01);   // (b0, b2) + (b3, b5)
                    gg[t][0] = ((d0 >> 8) & 255u) + (d1 & 255u);                                             // b1 + b4
                    rb[t][1] = __builtin_amdgcn_perm(d2, d1, sel10) + __builtin_amdgcn_perm(d2, d2, sel11);   // (b6, b8) + (b9, b11)
                    gg[t][1] = (d1 >> 24) + ((d2 >> 16) & 255u);                                             // b7 + b10
                }
                const int lxa = 2 * pv[k] - de;
                const int gy = gy0 + rowv[k];
#pragma unroll
                for (int h = 0; h < 2; h++) {   // half h of the group: column lxa + h under step +2, lxa + 1 - h under -2
                    const uint32_t rb0 = rb[0][h], rb1 = rb[1][h], g0v = gg[0][h], g1v = gg[1][h];
                    const uint32_t c0 = (__umul24(wy0, rb0 & 0xffffu) + __umul24(wy, rb1 & 0xffffu) + rnd) >> (kCoefBits + 1);
                    const uint32_t c1 = (__umul24(wy0, g0v) + __umul24(wy, g1v) + rnd) >> (kCoefBits + 1);
                    const uint32_t c2 = (__umul24(wy0, rb0 >> 16) + __umul24(wy, rb1 >> 16) + rnd) >> (kCoefBits + 1);
                    const half_t v0 = (half_t)((float)c0 * inv255), v1 = (half_t)((float)c1 * inv255), v2 = (half_t)((float)c2 * inv255);
                    const int lx = lxa + (a.fx_step > 0 ? h : 1 - h);
                    half4 o = (half4){v0, v1, v2, (half_t)0.0f};
                    if (EDGE) {   // outside the net input: the convolution's zero padding; inside without a source: the letterbox grey
                        const int gx = gx0 + lx;
                        const bool in_net = (unsigned)gy < (unsigned)net && (unsigned)gx < (unsigned)net;
                        const bool has_src = gy >= a.vy0 && gy < a.vy1 && gx >= a.vx0 && gx < a.vx1;
                        o = !in_net ? (half4){0, 0, 0, 0} : (has_src ? o : (half4){padv, padv, padv, (half_t)0.0f});
                    }
                    if ((unsigned)lx < (unsigned)INW) s_in[rowv[k] * INP + lx] = o;
                }
            }
        };
        if (tile_inside) run_direct(std::false_type{}); else run_direct(std::true_type{});
        FSTAMP(1); FSTAMP(2); FSTAMP(3);
    } else if constexpr (TY == 4) {
    // ---- 0: bounding box of the source pixels the tile touches.  The tap tables are monotonic, so the box follows
        // from the taps of the first and last in-image row / column of the tile (uniform addresses: scalar loads), and the
        // source loads of A1 can be issued at once instead of behind a table read and a reduction.
        const int cx_lo = max(gx0, a.vx0), cx_hi = min(gx0 + INW, a.vx1) - 1;
        const int cy_lo = max(gy0, a.vy0), cy_hi = min(gy0 + INH, a.vy1) - 1;
        const bool any_src = cx_lo <= cx_hi && cy_lo <= cy_hi;
        int x0 = 0, x1 = 0, sy_min = 0, sy_max = -1;
        if (any_src) {
            const AxisTap xa = a.tx[cx_lo], xb = a.tx[cx_hi], ya = a.ty[cy_lo], yb = a.ty[cy_hi];
            // (min3 / max3 exist on the vector unit only: back to scalar registers, or everything derived from the box --
            // the region's base address, its group count, the walk's steps -- is computed per lane)
            x0 = __builtin_amdgcn_readfirstlane(min(min(xa.i0, xa.i1), min(xb.i0, xb.i1)) & ~3);   // the region starts on a 4-pixel (12-byte) group
            x1 = __builtin_amdgcn_readfirstlane(min((max(max(xa.i0, xa.i1), max(xb.i0, xb.i1)) + 4) & ~3, a.sw));
            sy_min = __builtin_amdgcn_readfirstlane(min(min(ya.i0, ya.i1), min(yb.i0, yb.i1)));
            sy_max = __builtin_amdgcn_readfirstlane(max(max(ya.i0, ya.i1), max(yb.i0, yb.i1)));
        }
        const int pitch = x1 - x0;                                       // staged pixels per row (4 bytes each)
        // this lane's tap (lanes 0 .. INW + INH - 1: one column or row of the tile each); packed and stored behind the source loads
        // (loaded by EVERY lane from a clamped index and masked where it is used: a load under a branch is copied out of the
        // branch's block behind a wait -- for every load issued so far -- in front of the source loads)
        const bool tap_x = tid < INW;
        const int tap_i = tap_x ? gx0 + tid : gy0 + (tid - INW);
        const bool tap_ok = tid < INW + INH && (unsigned)tap_i < (unsigned)net;
        AxisTap my_tap = (tap_x ? a.tx : a.ty)[min(max(tap_i, 0), net - 1)];
    
        FSTAMP(1);
        // ---- A1: source region -> LDS as 4-byte pixels (12 source bytes -> one 16-byte LDS store) ----
        uint32_t *s_px = reinterpret_cast<uint32_t *>(s_stage);
        if (any_src) {
            const size_t row_bytes = (size_t)a.sw * 3;
            const uint8_t *src = a.src + (size_t)b * a.src_slot_bytes + (size_t)sy_min * row_bytes + (size_t)x0 * 3;
            const int gpr = pitch >> 2, total = (sy_max - sy_min + 1) * gpr;
            // All loads of a pass are issued before its first LDS store: a thread owns ~5 groups of the reference geometry
            // (33 rows x 35 groups / 256 threads), and one group per loop trip meant ~5 exposed memory round trips per
            // workgroup -- of a ~10 us workgroup lifetime.  CH = 6 groups per pass: one round trip for 1280 x 1024 -> 640.
            // The kernel is bound by its vector-instruction issue (profiles/r02_mfma.json: VALU busy ~100 %), so the walk
            // costs no division per group: group i sits at LDS dword 4 i and at source byte row(i) * skip + 12 i, and
            // row(i + 256) follows from row(i) with one compare.
            constexpr int CH = 6;
            const float inv_gpr = __builtin_amdgcn_rcpf((float)gpr);   // (1 ulp: the correction steps below absorb it)
            int dq = (int)(256.0f * inv_gpr);                         // 256 = dq * gpr + dr (uniform)
            dq -= (dq * gpr > 256) ? 1 : 0;
            dq += ((dq + 1) * gpr <= 256) ? 1 : 0;
            dq = __builtin_amdgcn_readfirstlane(dq);
            const int dr = 256 - dq * gpr;
            int row = (int)((float)tid * inv_gpr);                    // tid < 2^15: one correction step makes the quotient exact
            row -= (__mul24(row, gpr) > tid) ? 1 : 0;
            row += (__mul24(row + 1, gpr) <= tid) ? 1 : 0;
            int col = tid - __mul24(row, gpr);
            const uint32_t skip = (uint32_t)(row_bytes - (size_t)gpr * 12);   // < 2^24 (the source is at most 4096 pixels wide)
            const uint32_t off_last = (uint32_t)(sy_max - sy_min) * skip + (uint32_t)(total - 1) * 12u;
            uint32_t d0[CH], d1[CH], d2[CH];
            auto load_pass = [&](int i0) {
    #pragma unroll
                for (int c = 0; c < CH; c++) {
                    const int i = i0 + c * 256 + tid;
                    // (unconditional, past the end the region's last group again: loads under a branch are waited for one by one)
                    const uint32_t off = i < total ? __umul24((uint32_t)row, skip) + __umul24((uint32_t)i, 12u) : off_last;
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(src + off);
                    d0[c] = q[0]; d1[c] = q[1]; d2[c] = q[2];
                    col += dr; row += dq;
                    if (col >= gpr) { col -= gpr; row++; }
                }
            };
            auto store_pass = [&](int i0) {
    #pragma unroll
                for (int c = 0; c < CH; c++) {
                    const int i = i0 + c * 256 + tid;
                    if (i < total) {
                        uint4 o;
                        o.x = d0[c] & 0xffffffu;
                        o.y = (d0[c] >> 24) | ((d1[c] & 0xffffu) << 8);
                        o.z = (d1[c] >> 16) | ((d2[c] & 0xffu) << 16);
                        o.w = d2[c] >> 8;
                        *reinterpret_cast<uint4 *>(s_px + 4 * i) = o;
                    }
                }
            };
            // the first pass in straight-line code: at a loop header the compiler waits for (nearly) every load in flight --
            // the weights -- before the pass's own loads are issued
            load_pass(0);
            store_pass(0);
            for (int i0 = 256 * CH; i0 < total; i0 += 256 * CH) {
                load_pass(i0);
                store_pass(i0);
            }
        }
        // taps relative to the region, one dword each: i0 | i1 << 10 | w << 20 (w <= 2048); all ones = outside / padding.
        // (The empty asm pins the first use of the loaded tap HERE: the compiler otherwise starts packing it right behind its
        // load, and the wait that takes -- for every load issued so far -- lands in front of the source loads.)
        asm volatile("" : "+v"(my_tap.i0), "+v"(my_tap.i1), "+v"(my_tap.w1));
        if (tid < INW + INH) {
            const int base = tap_x ? x0 : sy_min;
            const uint32_t pk = (!tap_ok || my_tap.i0 < 0) ? 0xffffffffu : (uint32_t)(my_tap.i0 - base) | ((uint32_t)(my_tap.i1 - base) << 10) | ((uint32_t)my_tap.w1 << 20);
            if (tap_x) s_tx[tid] = pk; else s_ty[tid - INW] = pk;
        }
        if (tid < 48) s_bias[tid] = bias_v;
        FSTAMP(2);
        __syncthreads();
        FSTAMP(3);
    
        // ---- A2: bilinear resample into the NHWC4 tile (arithmetic of preprocess_kernel) ----
        {
            const half_t padv = (half_t)(114.0f / 255.0f);
            const float inv255 = 1.0f / 255.0f;   // (half)(q * inv255) == (half)(q / 255.0f) for every q in 0..255 (tests/test_oracle_preprocess.py)
            if (tid < INH) s_in[tid * INP + INW] = (half4){0, 0, 0, 0};   // the extra column of every row stays zero
            // 19 x 67 = 1273 pixels = five trips of 256 lanes (walking the padded 19 x 68 grid would need a sixth for 12 pixels)
            if (a.fastx) {
                // Columns at exactly 2 : 1 (every x tap = (2 k, 2 k + 1) with weight 1/2: 1280 -> 640): the horizontal blend
                // of a channel is 1024 (p0 + p1), so with s = p0 + p1 of the two tap rows the fixed-point result
                // ((2048 - wy) 1024 s0 + wy 1024 s1 + 2^21) >> 22 is ((2048 - wy) s0 + wy s1 + 2^11) >> 12 -- the same integer.
                // The pair of a row is one aligned 8-byte LDS read; red / blue are summed side by side in one register.
                const int xbase = a.fx_i0 + a.fx_step * (gx0 - a.vx0) - x0;   // region-relative pair of lx = 0 (even: x0 and fx_i0 are)
                const int xstep = a.fx_step;                                   // -2 under rotate180
                auto blend = [&](int ly, int lx, uint32_t ty) -> half4 {
                    const int xa = xbase + xstep * lx;
                    const uint2 q0 = *reinterpret_cast<const uint2 *>(s_px + __umul24(ty & 1023u, pitch) + xa);
                    const uint2 q1 = *reinterpret_cast<const uint2 *>(s_px + __umul24((ty >> 10) & 1023u, pitch) + xa);
                    const uint32_t wy = ty >> 20, wy0 = kCoefOne - wy;
                    const uint32_t rb0 = (q0.x & 0x00ff00ffu) + (q0.y & 0x00ff00ffu), rb1 = (q1.x & 0x00ff00ffu) + (q1.y & 0x00ff00ffu);
                    const uint32_t g0 = ((q0.x >> 8) & 255u) + ((q0.y >> 8) & 255u), g1 = ((q1.x >> 8) & 255u) + ((q1.y >> 8) & 255u);
                    const uint32_t rnd = 1u << (kCoefBits);
                    const uint32_t c0 = (__umul24(wy0, rb0 & 0xffffu) + __umul24(wy, rb1 & 0xffffu) + rnd) >> (kCoefBits + 1);
                    const uint32_t c1 = (__umul24(wy0, g0) + __umul24(wy, g1) + rnd) >> (kCoefBits + 1);
                    const uint32_t c2 = (__umul24(wy0, rb0 >> 16) + __umul24(wy, rb1 >> 16) + rnd) >> (kCoefBits + 1);
                    half_t v0 = (half_t)((float)c0 * inv255), v1 = (half_t)((float)c1 * inv255), v2 = (half_t)((float)c2 * inv255);
                    if (a.swap_rb) { const half_t t = v0; v0 = v2; v2 = t; }
                    return (half4){v0, v1, v2, (half_t)0.0f};
                };
                // pixel tid + 256 k sits 3 rows and 55 columns (256 = 3 INW + 55) past pixel tid + 256 (k - 1)
                static_assert(INW == 67 && INH * INW <= 5 * 256, "the walk below is written for the 19 x 67 tile");
                int ly = tid / INW, lx = tid - ly * INW;
                const bool inside = gx0 >= max(a.vx0, 0) && gx0 + INW <= min(a.vx1, net) && gy0 >= max(a.vy0, 0) && gy0 + INH <= min(a.vy1, net);
                if (inside) {   // (three tiles in four) every pixel of the tile has both taps: nothing to test per pixel
    #pragma unroll
                    for (int k = 0; k < 5; k++) {
                        if (k < 4 || tid < INH * INW - 4 * 256) s_in[ly * INP + lx] = blend(ly, lx, s_ty[ly]);
                        lx += 256 - 3 * INW; ly += 3;
                        if (lx >= INW) { lx -= INW; ly++; }
                    }
                } else {
    #pragma unroll
                    for (int k = 0; k < 5; k++) {
                        if (k < 4 || tid < INH * INW - 4 * 256) {
                            const int gx = gx0 + lx;
                            half4 o = (half4){0, 0, 0, 0};
                            if ((unsigned)(gy0 + ly) < (unsigned)net && (unsigned)gx < (unsigned)net) {
                                const uint32_t ty = s_ty[ly];
                                o = (ty == 0xffffffffu || gx < a.vx0 || gx >= a.vx1) ? (half4){padv, padv, padv, (half_t)0.0f} : blend(ly, lx, ty);
                            }
                            s_in[ly * INP + lx] = o;
                        }
                        lx += 256 - 3 * INW; ly += 3;
                        if (lx >= INW) { lx -= INW; ly++; }
                    }
                }
            } else {
    #pragma unroll 5
            for (int pp = tid; pp < INH * INW; pp += 256) {
                const int ly = pp / INW, lx = pp - ly * INW;
                const int p = ly * INP + lx;
                half4 o = (half4){0, 0, 0, 0};
                if ( (unsigned)(gy0 + ly) < (unsigned)net && (unsigned)(gx0 + lx) < (unsigned)net) {
                    const uint32_t ty = s_ty[ly], tx = s_tx[lx];
                    if (ty == 0xffffffffu || tx == 0xffffffffu) {
                        o = (half4){padv, padv, padv, (half_t)0.0f};
                    } else {
                        const uint32_t *r0 = s_px + __umul24(ty & 1023u, pitch), *r1 = s_px + __umul24((ty >> 10) & 1023u, pitch);
                        const uint32_t xa = tx & 1023u, xb = (tx >> 10) & 1023u;
                        const uint32_t wx = tx >> 20, wy = ty >> 20;
                        const uint32_t q00 = r0[xa], q01 = r0[xb], q10 = r1[xa], q11 = r1[xb];
                        half_t v[3];
    #pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const uint32_t p00 = (q00 >> (8 * c)) & 255u, p01 = (q01 >> (8 * c)) & 255u;
                            const uint32_t p10 = (q10 >> (8 * c)) & 255u, p11 = (q11 >> (8 * c)) & 255u;
                            // 24-bit multiplies (full rate): coefficients <= 2^11, pixels < 2^8, top / bot < 2^19
                            const uint32_t top = __umul24(kCoefOne - wx, p00) + __umul24(wx, p01);
                            const uint32_t bot = __umul24(kCoefOne - wx, p10) + __umul24(wx, p11);
                            const uint32_t acc = __umul24(kCoefOne - wy, top) + __umul24(wy, bot);
                            v[c] = (half_t)((float)((acc + (1u << (2 * kCoefBits - 1))) >> (2 * kCoefBits)) * inv255);
                        }
                        if (a.swap_rb) { const half_t t = v[0]; v[0] = v[2]; v[2] = t; }
                        o = (half4){v[0], v[1], v[2], (half_t)0.0f};
                    }
                }
                s_in[p] = o;
            }
            }
        }
    }
    FSTAMP(4);
    __syncthreads();
    FSTAMP(5);

    // ---- B: model.0.conv on the tile: C0H x 33 output pixels, 16 channels, K = [kh][4 tap slots][4 ch] ----
    // s_c0: parity-split columns ([row][column parity][column / 2][16 ch], 32 B per pixel): the stride-2
    // fragment reads of stage C then walk consecutive 32-byte slots (conflict-free ds_read_b128).
    half_t *s_c0 = reinterpret_cast<half_t *>(s_stage);
    {
        const half4 z4 = (half4){0, 0, 0, 0};
        float bias0[4];
#pragma unroll
        for (int i = 0; i < 4; i++) bias0[i] = s_bias[g * 4 + i];
        // MFMA tiles (16 pixels each): two per row (columns 0 .. 31) + the last column top to bottom.  Tile t = wave + 4 k of a
        // wave is then row (wave >> 1) + 2 k, half (wave & 1): every LDS address is a per-lane constant plus a compile-time
        // multiple of k -- no division, no address arithmetic per tile (a flat walk over the 33-wide rows cost ~ 20 vector
        // instructions per tile for 2 MFMAs; this kernel is bound by those).  Only the last k mixes in the column tiles.
        // A tile is a chain LDS read -> two MFMAs -> SiLU -> LDS write; run one tile at a time it is the chain's LATENCY
        // that a wave spends (ablation: the reads alone were 57 of the kernel's 182 us per 64 frames), so the wave's
        // tiles go in batches: every fragment read of a batch is issued first, then its MFMAs, then its epilogues.
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const bool c0_inside = 2 * oy0 - 1 >= 0 && 2 * oy0 - 1 + C0H <= W0 && 2 * ox0 - 1 >= 0 && 2 * ox0 - 1 + C0W <= W0;   // no model.1 padding in this tile
        constexpr int NROWT = 2 * C0H, NCOLT = (C0H + 15) / 16, NTILES = NROWT + NCOLT;
        constexpr int KT = (NTILES + 3) / 4;   // tiles per wave (the last one may not exist for the upper waves)
        static_assert(C0W == 33 && NROWT % 4 == 2 && NCOLT <= 2, "row tiles end with waves 0, 1 of the last k; waves 2, 3 take the column tiles");
        constexpr int BT = 2;
        const int row0 = wave_u >> 1, lx_r = 16 * (wave_u & 1) + r;      // row tiles: first row, this lane's column
        // k slots without a tap (kernel row 3, tap slot 3) carry ZERO weights (engine.cpp packs them so), and s_in
        // holds finite values only (pixels, padding, zeros; column INW of every row is finite): such a slot may read any
        // pixel of the tile -- 0 x finite adds nothing -- so both halves of a fragment are ONE unpredicated 16-byte read
        const half4 *rd_row[2] = {s_in + (2 * row0 + (g >> 1)) * INP + 2 * lx_r + 2 * (g & 1),     // k-step 0: kernel row g >> 1
                                  s_in + (2 * row0 + 2) * INP + 2 * lx_r + 2 * (g & 1)};            // k-step 1: kernel row 2 (g < 2) / no tap (row 2 again)
        half_t *st_row = s_c0 + (size_t)((row0 * 2 + (lx_r & 1)) * C0HALF + (lx_r >> 1)) * 16 + g * 4;
        const int cx_row = 2 * ox0 - 1 + lx_r;
        // column tiles (last k, waves 2 and 3): column C0W - 1, rows 16 (wave - 2) + r
        const int ly_c = 16 * (wave_u - 2) + r;
        const bool col_tile = wave_u >= 2, col_ok = col_tile && (unsigned)ly_c < (unsigned)C0H;
        const int ly_cc = col_ok ? ly_c : 0;
        const half4 *rd_col[2] = {s_in + (2 * ly_cc + (g >> 1)) * INP + 2 * (C0W - 1) + 2 * (g & 1),
                                  s_in + (2 * ly_cc + 2) * INP + 2 * (C0W - 1) + 2 * (g & 1)};
        half_t *st_col = s_c0 + (size_t)((ly_cc * 2 + ((C0W - 1) & 1)) * C0HALF + ((C0W - 1) >> 1)) * 16 + g * 4;
#pragma unroll
        for (int k0 = 0; k0 < KT; k0 += BT) {
            half8 bf[BT][2];
            f32x4 acc[BT];
#pragma unroll
            for (int k = 0; k < BT; k++) {
                if (k0 + k >= KT) continue;
                const bool last = k0 + k == KT - 1;
#pragma unroll
                for (int s = 0; s < 2; s++)
                    bf[k][s] = *reinterpret_cast<const half8 *>(last && col_tile ? rd_col[s] : rd_row[s] + (k0 + k) * (4 * INP));
            }
#pragma unroll
            for (int k = 0; k < BT; k++) {
                if (k0 + k >= KT) continue;
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A00, bf[k][0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < BT; k++) {
                if (k0 + k >= KT) continue;
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A01, bf[k][1], acc[k], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < BT; k++) {
                if (k0 + k >= KT) continue;
                const bool last = k0 + k == KT - 1;
                const bool colt = last && col_tile;                       // (wave-uniform)
                if (colt && !col_ok) continue;                            // rows past the tile (and wave 3 of the 4-row tile: no tile at all)
                const int ly = colt ? ly_c : row0 + 2 * (k0 + k);
                const int cy = 2 * oy0 - 1 + ly, cx = colt ? 2 * ox0 - 1 + (C0W - 1) : cx_row;
                half4 o = z4;
                if (c0_inside || ((unsigned)cy < (unsigned)W0 && (unsigned)cx < (unsigned)W0)) {
                    // model.0 reads the UNSCALED image: one fma brings the accumulator to the activation scale (bias0 = log2 e * b)
                    o = silu_pack4(__builtin_fmaf(acc[k][0], kActScale, bias0[0]), __builtin_fmaf(acc[k][1], kActScale, bias0[1]),
                                   __builtin_fmaf(acc[k][2], kActScale, bias0[2]), __builtin_fmaf(acc[k][3], kActScale, bias0[3]));
                }
                *reinterpret_cast<half4 *>(colt ? st_col : st_row + (size_t)(k0 + k) * (4 * C0HALF * 16)) = o;
            }
        }
    }
    FSTAMP(6);
    __syncthreads();
    FSTAMP(7);

    // ---- C: model.1.conv: 8 x 16 outputs x 32 channels; Cin = 16, so a k-step of 32 spans two taps ----
    {
        float bias1[8];   // lane g holds channels g*8 + [0, 8): tile nt starts at bias1[4 nt ..] (accumulators start at the bias)
#pragma unroll
        for (int i = 0; i < 8; i++) bias1[i] = s_bias[16 + g * 8 + i];
        f32x4 acc[MTC][2];
#pragma unroll
        for (int mt = 0; mt < MTC; mt++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++) acc[mt][nt] = (f32x4){bias1[4 * nt], bias1[4 * nt + 1], bias1[4 * nt + 2], bias1[4 * nt + 3]};
        const half8 zero8 = (half8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 5; ks++) {
            const int tap = 2 * ks + (g >> 1);
            const int kh = tap / 3, kw = tap - kh * 3;
            half8 B[MTC];
#pragma unroll
            for (int mt = 0; mt < MTC; mt++) {
                B[mt] = zero8;
                if (tap < 9) {
                    const int ly = 2 * (MTC * wave + mt) + kh, lx = 2 * r + kw;
                    B[mt] = *reinterpret_cast<const half8 *>(s_c0 + (size_t)((ly * 2 + (lx & 1)) * C0HALF + (lx >> 1)) * 16 + 8 * (g & 1));
                }
            }
#pragma unroll
            for (int mt = 0; mt < MTC; mt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1[ks][nt], B[mt], acc[mt][nt], 0, 0, 0);
        }
        // epilogue of the direct kernel's paired-tile path: lane g holds channels g*8 + [0, 8)
        const int ox = ox0 + r;
        if (ox < W1) {
#pragma unroll
            for (int mt = 0; mt < MTC; mt++) {
                const int oy = oy0 + MTC * wave + mt;
                if (oy >= W1) continue;
                float vals[8];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    vals[i] = acc[mt][0][i];
                    vals[4 + i] = acc[mt][1][i];
                }
                const half8 o = silu_pack8(vals[0], vals[1], vals[2], vals[3], vals[4], vals[5], vals[6], vals[7]);
                *reinterpret_cast<half8 *>(a.out + ((size_t)(b * W1 + oy) * W1 + ox) * a.out_ld + g * 8) = o;
            }
        }
    }
#if IRMV_FSTAMP
    FSTAMP(8);
    if (tid == 0) {
        const unsigned wg = blockIdx.x & 65535u;
        for (int k = 0; k < 9; k++) g_front_stamps[wg * 9 + k] = fst[k];
    }
#endif
}

int front_min_stage_bytes(int tile_y) { return (2 * tile_y + 1) * 2 * C0HALF * 32; }

// Raises the kernels' dynamic-LDS limit (default 64 KiB); call once per process before the first launch / capture.
bool front_prepare()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(front_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, kFrontStageMax) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void *>(front_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, kFrontStageMax) == hipSuccess;
}

bool launch_front(const FrontArgs &a, int batch, hipStream_t s)
{
    if (a.stage_bytes < front_min_stage_bytes(a.tile_y) || a.stage_bytes > kFrontStageMax) return false;
    const dim3 grid(a.tiles_x * a.tiles_y * batch);
    const int xcd = xcd_image_order();
    if (a.tile_y == kFrontTileY) hipLaunchKernelGGL(front_kernel<kFrontTileY>, grid, dim3(256), (size_t)a.stage_bytes, s, a, batch, xcd);
    else if (a.tile_y == kFrontTileYDirect && (a.fastx & 2)) hipLaunchKernelGGL(front_kernel<kFrontTileYDirect>, grid, dim3(256), (size_t)a.stage_bytes, s, a, batch, xcd);   // the tall tile has no staged path
    else return false;
    return true;
}

}  // namespace irmv
