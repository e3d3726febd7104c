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

namespace irmv {

namespace {
constexpr int T = kC2fTile;               // output tile (pixels per side)
constexpr int XW = T + 4, XN = XW * XW;   // block input / y1 region (halo 2)
constexpr int TW = T + 2, TN = TW * TW;   // bottleneck intermediate region (halo 1)
constexpr int HP = 32;                    // bytes per pixel of the 16-channel planes (16 consecutive pixels x 2 halves hit
                                          // 16 distinct 16-byte slots of a 256-byte window: conflict-free ds_read_b128)

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
}  // namespace

__global__ __launch_bounds__(256) void c2f2_kernel(C2fArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_y0[T * T * HP];
    __shared__ __attribute__((aligned(16))) uint8_t s_y1[XN * HP];
    __shared__ __attribute__((aligned(16))) uint8_t s_t[TN * HP];
    __shared__ __attribute__((aligned(16))) uint8_t s_y2[T * T * HP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int b = blockIdx.y;
    const int tyi = blockIdx.x / a.tiles, txi = blockIdx.x - tyi * a.tiles;
    const int oy0 = tyi * T, ox0 = txi * T;
    const int S = a.S;
    const half8 zero8 = (half8){0, 0, 0, 0, 0, 0, 0, 0};

    // ---- weights (16 fragments) into registers; the loads overlap the input staging ----
    const half8 *w;
    w = reinterpret_cast<const half8 *>(a.w_cv1) + lane;
    const half8 Wc1[2] = {w[0], w[64]};
    half8 Wm1[5], Wm2[5];
    w = reinterpret_cast<const half8 *>(a.w_m1) + lane;
#pragma unroll
    for (int ks = 0; ks < 5; ks++) Wm1[ks] = w[ks * 64];
    w = reinterpret_cast<const half8 *>(a.w_m2) + lane;
#pragma unroll
    for (int ks = 0; ks < 5; ks++) Wm2[ks] = w[ks * 64];
    w = reinterpret_cast<const half8 *>(a.w_cv2) + lane;
    const half8 Wc2[2][2] = {{w[0], w[64]}, {w[128], w[192]}};   // [tile][k-step]

    // ---- 1: cv1 (1x1, 32 -> 32, SiLU) on the 20 x 20 region -> y0 (centre only), y1 (whole region) ----
    {
        float bias[8];
#pragma unroll
        for (int i = 0; i < 8; i++) bias[i] = a.b_cv1[g * 8 + i];
        const half_t *xin = a.x + (size_t)b * S * S * a.x_ld;
        for (int t = wave; t < XN / 16; t += 4) {
            const int m = t * 16 + r;
            // the block input is read exactly once per workgroup, so its B fragments come straight from memory (zero
            // outside the image)
            half8 B = zero8;
            {
                const int ly = m / XW, lx = m - ly * XW;
                const int gy = oy0 - 2 + ly, gx = ox0 - 2 + lx;
                if ((unsigned)gy < (unsigned)S && (unsigned)gx < (unsigned)S)
                    B = *reinterpret_cast<const half8 *>(xin + ((size_t)gy * S + gx) * a.x_ld + g * 8);
            }
            f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc1[0], B, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc1[1], B, acc1, 0, 0, 0);
            const int ly = m / XW, lx = m - ly * XW;
            const int gy = oy0 - 2 + ly, gx = ox0 - 2 + lx;
            half8 o = zero8;   // outside the image y1 is the bottleneck's zero padding
            if ((unsigned)gy < (unsigned)S && (unsigned)gx < (unsigned)S) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    o[i] = (half_t)silu(acc0[i] + bias[i]);
                    o[4 + i] = (half_t)silu(acc1[i] + bias[4 + i]);
                }
            }
            if (g >= 2) {
                *reinterpret_cast<half8 *>(s_y1 + m * HP + (g - 2) * 16) = o;
            } else if ((unsigned)(ly - 2) < (unsigned)T && (unsigned)(lx - 2) < (unsigned)T) {
                *reinterpret_cast<half8 *>(s_y0 + ((ly - 2) * T + (lx - 2)) * HP + g * 16) = o;
            }
        }
    }
    __syncthreads();

    // ---- 2: m.0.cv1 (3x3, 16 -> 16, SiLU) on the 18 x 18 region; Cin = 16: one k-step spans two taps ----
    {
        float bias[4];
#pragma unroll
        for (int i = 0; i < 4; i++) bias[i] = a.b_m1[g * 4 + i];
        int toff[5];   // byte offset of this lane's tap / channel half inside y1, relative to the pixel (-1 = no tap)
#pragma unroll
        for (int ks = 0; ks < 5; ks++) {
            const int tap = 2 * ks + (g >> 1), kh = tap / 3, kw = tap - kh * 3;
            toff[ks] = tap < 9 ? (kh * XW + kw) * HP + 16 * (g & 1) : -1;
        }
        for (int t = wave; t < (TN + 15) / 16; t += 4) {
            const int m = t * 16 + r;
            const bool mv = m < TN;
            const int mm = mv ? m : 0;
            const int ly = mm / TW, lx = mm - ly * TW;
            const uint8_t *base = s_y1 + (ly * XW + lx) * HP;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 5; ks++) {
                half8 B = zero8;
                if (toff[ks] >= 0) B = *reinterpret_cast<const half8 *>(base + toff[ks]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wm1[ks], B, acc, 0, 0, 0);
            }
            if (mv) {
                const int gy = oy0 - 1 + ly, gx = ox0 - 1 + lx;
                half4 o = (half4){0, 0, 0, 0};
                if ((unsigned)gy < (unsigned)S && (unsigned)gx < (unsigned)S) {
#pragma unroll
                    for (int i = 0; i < 4; i++) o[i] = (half_t)silu(acc[i] + bias[i]);
                }
                *reinterpret_cast<half4 *>(s_t + m * HP + g * 8) = o;
            }
        }
    }
    __syncthreads();

    // ---- 3: m.0.cv2 (3x3, 16 -> 16, SiLU) + shortcut y1 -> y2 on the 16 x 16 tile ----
    {
        float bias[4];
#pragma unroll
        for (int i = 0; i < 4; i++) bias[i] = a.b_m2[g * 4 + i];
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
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 5; ks++) {
                half8 B = zero8;
                if (toff[ks] >= 0) B = *reinterpret_cast<const half8 *>(base + toff[ks]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wm2[ks], B, acc, 0, 0, 0);
            }
            const half4 rv = *reinterpret_cast<const half4 *>(s_y1 + ((ly + 2) * XW + lx + 2) * HP + g * 8);
            half4 o;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // the unfused epilogue adds the shortcut behind a run-time branch, i.e. to the ROUNDED product: no fma here
#pragma clang fp contract(off)
                const float act = silu(acc[i] + bias[i]);
                o[i] = (half_t)(act + (float)rv[i]);
            }
            *reinterpret_cast<half4 *>(s_y2 + m * HP + g * 8) = o;
        }
    }
    __syncthreads();

    // ---- 4: cv2 (1x1 over [y0 | y1 | y2] = 48 channels -> 32, SiLU) -> HBM ----
    {
        float bias[8];
#pragma unroll
        for (int i = 0; i < 8; i++) bias[i] = a.b_cv2[g * 8 + i];
        half_t *out = a.out + (size_t)b * S * S * a.out_ld;
        for (int t = wave; t < T * T / 16; t += 4) {
            const int m = t * 16 + r;
            const int ly = m / T, lx = m - ly * T;
            // k-step 0: channels 8g..8g+7 of the concat: y0 for g < 2, y1 for g >= 2; k-step 1: y2 for g < 2, nothing above 48
            const half8 B0 = g < 2 ? *reinterpret_cast<const half8 *>(s_y0 + m * HP + g * 16)
                                   : *reinterpret_cast<const half8 *>(s_y1 + ((ly + 2) * XW + lx + 2) * HP + (g - 2) * 16);
            half8 B1 = zero8;
            if (g < 2) B1 = *reinterpret_cast<const half8 *>(s_y2 + m * HP + g * 16);
            f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc2[0][0], B0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc2[1][0], B0, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc2[0][1], B1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc2[1][1], B1, acc1, 0, 0, 0);
            const int gy = oy0 + ly, gx = ox0 + lx;
            if (gy < S && gx < S) {
                half8 o;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    o[i] = (half_t)silu(acc0[i] + bias[i]);
                    o[4 + i] = (half_t)silu(acc1[i] + bias[4 + i]);
                }
                *reinterpret_cast<half8 *>(out + ((size_t)gy * S + gx) * a.out_ld + g * 8) = o;
            }
        }
    }
}

void launch_c2f2(const C2fArgs &a, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(c2f2_kernel, dim3(a.tiles * a.tiles, batch), dim3(256), 0, s, a);
}

}  // namespace irmv
