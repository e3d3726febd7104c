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

namespace irmv {

namespace {
constexpr int kCoefBits = 11;
constexpr int kCoefOne = 1 << kCoefBits;
constexpr int TY = kFrontTileY, TX = kFrontTileX;   // model.1 output tile
constexpr int C0H = 2 * TY + 1, C0W = 2 * TX + 1;    // model.0 outputs it needs
constexpr int C0HALF = (C0W + 1) / 2;                // columns per parity plane
constexpr int INH = 4 * TY + 3, INW = 4 * TX + 3;    // net-input pixels those need
constexpr int MTC = TY / 4;                          // model.1 output rows per wave
constexpr int INP = INW + 1;                         // row pitch (pixels); the extra column stays zero
}  // namespace

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void front_kernel(FrontArgs a)   // 5 waves per SIMD = 96 VGPRs: the occupancy step the kernel sat on before its biases moved to LDS
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_stage[];   // a.stage_bytes: source region, later model.0's tile
    __shared__ __attribute__((aligned(16))) half4 s_in[INH * INP];
    __shared__ uint32_t s_tx[INW], s_ty[INH];   // packed region-relative taps
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int b = blockIdx.y;
    const int tyi = blockIdx.x / a.tiles_x, txi = blockIdx.x - tyi * a.tiles_x;
    const int oy0 = tyi * TY, ox0 = txi * TX;
    const int net = a.net, W0 = net >> 1, W1 = net >> 2;
    const int gy0 = 4 * oy0 - 3, gx0 = 4 * ox0 - 3;   // net-input coordinates of s_in[0][0]

    // both biases -> LDS now (visible after the staging barrier): fetched from global memory where they are used (start of
    // stage B, epilogue of stage C) each exposes a memory round trip of a ~10 us workgroup; held in registers from here
    // they cost the occupancy step this kernel sits on (measured +10 %)
    __shared__ float s_bias[16 + 32];
    if (tid < 16) s_bias[tid] = a.b0[tid];
    else if (tid < 48) s_bias[tid] = a.b1[tid - 16];

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

    // ---- 0: bounding box of the source pixels the tile touches.  The tap tables are monotonic, so the box follows
    // from the taps of the first and last in-image row / column of the tile (uniform addresses: scalar loads), and the
    // source loads of A1 can be issued at once instead of behind a table read and a reduction.
    const int cx_lo = max(gx0, a.vx0), cx_hi = min(gx0 + INW, a.vx1) - 1;
    const int cy_lo = max(gy0, a.vy0), cy_hi = min(gy0 + INH, a.vy1) - 1;
    const bool any_src = cx_lo <= cx_hi && cy_lo <= cy_hi;
    int x0 = 0, x1 = 0, sy_min = 0, sy_max = -1;
    if (any_src) {
        const AxisTap xa = a.tx[cx_lo], xb = a.tx[cx_hi], ya = a.ty[cy_lo], yb = a.ty[cy_hi];
        x0 = min(min(xa.i0, xa.i1), min(xb.i0, xb.i1)) & ~3;           // the region starts on a 4-pixel (12-byte) group
        x1 = min((max(max(xa.i0, xa.i1), max(xb.i0, xb.i1)) + 4) & ~3, a.sw);
        sy_min = min(min(ya.i0, ya.i1), min(yb.i0, yb.i1));
        sy_max = max(max(ya.i0, ya.i1), max(yb.i0, yb.i1));
    }
    const int pitch = x1 - x0;                                       // staged pixels per row (4 bytes each)
    // taps relative to the region, one dword each: i0 | i1 << 10 | w << 20 (w <= 2048); all ones = outside / padding
    if (tid < INW + INH) {
        const bool is_x = tid < INW;
        const int i = is_x ? gx0 + tid : gy0 + (tid - INW);
        AxisTap t = AxisTap{-1, -1, 0, 0};
        if ((unsigned)i < (unsigned)net) t = is_x ? a.tx[i] : a.ty[i];
        const int base = is_x ? x0 : sy_min;
        const uint32_t pk = t.i0 < 0 ? 0xffffffffu : (uint32_t)(t.i0 - base) | ((uint32_t)(t.i1 - base) << 10) | ((uint32_t)t.w1 << 20);
        if (is_x) s_tx[tid] = pk; else s_ty[tid - INW] = pk;
    }

    // ---- A1: source region -> LDS as 4-byte pixels (12 source bytes -> one 16-byte LDS store) ----
    uint32_t *s_px = reinterpret_cast<uint32_t *>(s_stage);
    if (any_src) {
        const size_t row_bytes = (size_t)a.sw * 3;
        const uint8_t *src = a.src + (size_t)b * a.src_slot_bytes + (size_t)sy_min * row_bytes + (size_t)x0 * 3;
        const int gpr = pitch >> 2, total = (sy_max - sy_min + 1) * gpr;
        // All loads of a pass are issued before its first LDS store: a thread owns ~5 groups of the reference geometry
        // (33 rows x 35 groups / 256 threads), and one group per loop trip meant ~5 exposed memory round trips per
        // workgroup -- of a ~10 us workgroup lifetime.  CH = 6 groups per pass: one round trip for 1280 x 1024 -> 640.
        constexpr int CH = 6;
        const float inv_gpr = 1.0f / (float)gpr;
        for (int i0 = 0; i0 < total; i0 += 256 * CH) {
            uint32_t d0[CH], d1[CH], d2[CH];
            int dst[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const int i = i0 + c * 256 + tid;
                dst[c] = -1;
                d0[c] = d1[c] = d2[c] = 0u;
                if (i < total) {
                    int row = (int)((float)i * inv_gpr);          // i < 2^15: one correction step makes the quotient exact
                    row -= (row * gpr > i) ? 1 : 0;
                    row += ((row + 1) * gpr <= i) ? 1 : 0;
                    const int gq = i - row * gpr;
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(src + (size_t)row * row_bytes + (size_t)gq * 12);
                    d0[c] = q[0]; d1[c] = q[1]; d2[c] = q[2];
                    dst[c] = row * pitch + gq * 4;
                }
            }
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (dst[c] >= 0) {
                    uint4 o;
                    o.x = d0[c] & 0xffffffu;
                    o.y = (d0[c] >> 24) | ((d1[c] & 0xffffu) << 8);
                    o.z = (d1[c] >> 16) | ((d2[c] & 0xffu) << 16);
                    o.w = d2[c] >> 8;
                    *reinterpret_cast<uint4 *>(s_px + dst[c]) = o;
                }
            }
        }
    }
    __syncthreads();

    // ---- A2: bilinear resample into the NHWC4 tile (arithmetic of preprocess_kernel) ----
    {
        const half_t padv = (half_t)(114.0f / 255.0f);
        const float inv255 = 1.0f / 255.0f;   // (half)(q * inv255) == (half)(q / 255.0f) for every q in 0..255 (tests/test_oracle_preprocess.py)
        if (tid < INH) s_in[tid * INP + INW] = (half4){0, 0, 0, 0};   // the extra column of every row stays zero
        // 19 x 67 = 1273 pixels = five trips of 256 lanes (walking the padded 19 x 68 grid would need a sixth for 12 pixels)
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
                    const uint32_t *r0 = s_px + (ty & 1023u) * pitch, *r1 = s_px + ((ty >> 10) & 1023u) * pitch;
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
    __syncthreads();

    // ---- B: model.0.conv on the tile: 17 x 33 output pixels, 16 channels, K = [kh][4 tap slots][4 ch] ----
    // s_c0: parity-split columns ([row][column parity][column / 2][16 ch], 32 B per pixel): the stride-2
    // fragment reads of stage C then walk consecutive 32-byte slots (conflict-free ds_read_b128).
    half_t *s_c0 = reinterpret_cast<half_t *>(s_stage);
    {
        constexpr int NPX = C0H * C0W, NTILES = (NPX + 15) / 16;
        const half4 z4 = (half4){0, 0, 0, 0};
        float bias0[4];
#pragma unroll
        for (int i = 0; i < 4; i++) bias0[i] = s_bias[g * 4 + i];
        for (int t = wave; t < NTILES; t += 4) {
            const int m = t * 16 + r;
            const bool mv = m < NPX;
            const int mm = mv ? m : 0;
            const int ly = mm / C0W, lx = mm - ly * C0W;
            half8 bf[2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int kh = 2 * s + (g >> 1);
                half4 lo = z4, hi = z4;
                if (mv && kh < 3) {
                    const half4 *q = s_in + (2 * ly + kh) * INP + 2 * lx + 2 * (g & 1);
                    lo = q[0];
                    if ((g & 1) == 0) hi = q[1];   // slot 3 is padding
                }
                bf[s] = (half8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A00, bf[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A01, bf[1], acc, 0, 0, 0);
            if (mv) {
                const int cy = 2 * oy0 - 1 + ly, cx = 2 * ox0 - 1 + lx;
                half4 o = z4;
                if ((unsigned)cy < (unsigned)W0 && (unsigned)cx < (unsigned)W0) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float v = acc[i] + bias0[i];
                        o[i] = (half_t)(v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)));
                    }
                }
                *reinterpret_cast<half4 *>(s_c0 + (size_t)((ly * 2 + (lx & 1)) * C0HALF + (lx >> 1)) * 16 + g * 4) = o;
            }
        }
    }
    __syncthreads();

    // ---- C: model.1.conv: 8 x 16 outputs x 32 channels; Cin = 16, so a k-step of 32 spans two taps ----
    {
        f32x4 acc[MTC][2];
#pragma unroll
        for (int mt = 0; mt < MTC; mt++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
            float bias1[8];
#pragma unroll
            for (int i = 0; i < 8; i++) bias1[i] = s_bias[16 + g * 8 + i];
#pragma unroll
            for (int mt = 0; mt < MTC; mt++) {
                const int oy = oy0 + MTC * wave + mt;
                if (oy >= W1) continue;
                float vals[8];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    vals[i] = acc[mt][0][i] + bias1[i];
                    vals[4 + i] = acc[mt][1][i] + bias1[4 + i];
                }
#pragma unroll
                for (int i = 0; i < 8; i++) vals[i] = vals[i] * __builtin_amdgcn_rcpf(1.0f + __expf(-vals[i]));
                half8 o;
#pragma unroll
                for (int i = 0; i < 8; i++) o[i] = (half_t)vals[i];
                *reinterpret_cast<half8 *>(a.out + ((size_t)(b * W1 + oy) * W1 + ox) * a.out_ld + g * 8) = o;
            }
        }
    }
}

int front_min_stage_bytes() { return C0H * 2 * C0HALF * 32; }

// Raises the kernel's dynamic-LDS limit (default 64 KiB); call once per process before the first launch / capture.
bool front_prepare()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(front_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               kFrontStageMax) == hipSuccess;
}

bool launch_front(const FrontArgs &a, int batch, hipStream_t s)
{
    if (a.stage_bytes < front_min_stage_bytes() || a.stage_bytes > kFrontStageMax) return false;
    hipLaunchKernelGGL(front_kernel, dim3(a.tiles_x * a.tiles_y, batch), dim3(256), (size_t)a.stage_bytes, s, a);
    return true;
}

}  // namespace irmv
