// Implicit-GEMM convolution on the gfx950 matrix cores, and the SPPF pooling chain.
//
// Replaces the convolution layers inside the reference's opaque TensorRT plan
// (src/yolo_engine.cpp:105).  Activations are fp16 NHWC, so the 8 contiguous K
// elements an MFMA lane needs are 8 contiguous channels of one pixel at one
// filter tap: a single 16-byte load, no im2col buffer.  The GEMM is computed
// transposed, D[cout][pixel] = W[cout][k] * X[k][pixel], with
// v_mfma_f32_16x16x32_f16: weights are the A operand (pre-packed on the host in
// exact fragment order, 1 KiB contiguous per wave load), pixels ride the lane
// index of the B operand, and each lane ends up with 4 (or, with the paired-tile
// channel permutation, 8) CONTIGUOUS output channels of one pixel -> 8/16-byte
// NHWC stores.  Bias, SiLU, the C2f shortcut add and the fp16 down-convert are
// fused in the epilogue; channel concat is free (convs read and write channel
// slices of wider buffers); the neck's nearest-2x upsample + concat is folded
// into the operand addressing (two K segments, one at half resolution).
#include "irmv_common.hpp"

#include <cstdio>

namespace irmv {

template <int KS, int STRIDE, int MT, int NT, bool CIN16, int ACT, bool OUT_F32>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a)
{
    constexpr int PAD = KS / 2;
    constexpr bool PAIR = !OUT_F32 && (NT % 2 == 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int tile0 = (blockIdx.x * 4 + wave) * MT;
    const int nt0 = blockIdx.y * NT;
    const int HWo = a.Hout * a.Wout;

    int iy0[MT], ix0[MT], bb[MT], mm[MT];
    bool mv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int m = (tile0 + mt) * 16 + r;
        mv[mt] = m < a.M;
        mm[mt] = mv[mt] ? m : 0;
        const int b = mm[mt] / HWo, rem = mm[mt] - b * HWo;
        const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
        iy0[mt] = oy * STRIDE - PAD;
        ix0[mt] = ox * STRIDE - PAD;
        bb[mt] = b;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const half8 *wp = reinterpret_cast<const half8 *>(a.w) + (size_t)nt0 * a.ksteps * 64 + lane;
    const int H0 = a.Hin >> a.s0.shift, W0 = a.Win >> a.s0.shift;
    const int H1 = a.Hin >> a.s1.shift, W1 = a.Win >> a.s1.shift;
    const half8 zero8 = (half8){0, 0, 0, 0, 0, 0, 0, 0};

    if constexpr (CIN16) {
        // Cin == 16: one k-step of 32 spans TWO filter taps (lanes g<2: tap 2i, g>=2: tap 2i+1)
        const int c = 8 * (g & 1);
        for (int ks = 0; ks < a.ksteps; ks++) {
            const int tap = 2 * ks + (g >> 1);
            const int kh = tap / KS, kw = tap - kh * KS;
            half8 bf[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int iy = iy0[mt] + kh, ix = ix0[mt] + kw;
                const bool v = mv[mt] && tap < KS * KS && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
                bf[mt] = zero8;
                if (v) bf[mt] = *reinterpret_cast<const half8 *>(a.s0.p + ((size_t)(bb[mt] * H0 + iy) * W0 + ix) * a.s0.ld + c);
            }
            half8 af[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) af[nt] = wp[(size_t)(nt * a.ksteps + ks) * 64];
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt], bf[mt], acc[mt][nt], 0, 0, 0);
        }
    } else {
        int ks = 0;
        for (int tap = 0; tap < KS * KS; tap++) {
            const int kh = tap / KS, kw = tap - kh * KS;
            const half_t *p0[MT], *p1[MT];
            bool v[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int iy = iy0[mt] + kh, ix = ix0[mt] + kw;
                v[mt] = mv[mt] && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
                const int iyc = v[mt] ? iy : 0, ixc = v[mt] ? ix : 0;
                p0[mt] = a.s0.p + ((size_t)(bb[mt] * H0 + (iyc >> a.s0.shift)) * W0 + (ixc >> a.s0.shift)) * a.s0.ld;
                p1[mt] = a.s1.p + ((size_t)(bb[mt] * H1 + (iyc >> a.s1.shift)) * W1 + (ixc >> a.s1.shift)) * a.s1.ld;
            }
            for (int cc = 0; cc < a.Cin; cc += 32, ks++) {
                const int c = cc + 8 * g;
                half8 bf[MT];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    bf[mt] = zero8;
                    if (v[mt] && c < a.Cin) {
                        const half_t *q = (c < a.s0.C) ? (p0[mt] + c) : (p1[mt] + (c - a.s0.C));
                        bf[mt] = *reinterpret_cast<const half8 *>(q);
                    }
                }
                half8 af[NT];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) af[nt] = wp[(size_t)(nt * a.ksteps + ks) * 64];
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt], bf[mt], acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias, SiLU, shortcut, convert, NHWC store ----
    // D layout of 16x16x32: col = lane & 15 (pixel), row = (lane >> 4) * 4 + reg (cout row of the tile)
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        if (!mv[mt]) continue;
        const size_t m = (size_t)mm[mt];
        if constexpr (PAIR) {
#pragma unroll
            for (int u = 0; u < NT / 2; u++) {
                const int c0 = (nt0 / 2 + u) * 32 + g * 8;
                float vals[8];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    vals[i] = acc[mt][2 * u][i] + a.bias[c0 + i];
                    vals[4 + i] = acc[mt][2 * u + 1][i] + a.bias[c0 + 4 + i];
                }
                if (ACT == 1) {
#pragma unroll
                    for (int i = 0; i < 8; i++) vals[i] = vals[i] * __frcp_rn(1.0f + __expf(-vals[i]));
                }
                if (a.res) {
                    const half8 rv = *reinterpret_cast<const half8 *>(a.res + m * a.res_ld + c0);
#pragma unroll
                    for (int i = 0; i < 8; i++) vals[i] += (float)rv[i];
                }
                half8 o;
#pragma unroll
                for (int i = 0; i < 8; i++) o[i] = (half_t)vals[i];
                *reinterpret_cast<half8 *>(static_cast<half_t *>(a.out) + m * a.out_ld + c0) = o;
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const int c0 = (nt0 + nt) * 16 + g * 4;
                float vals[4];
#pragma unroll
                for (int i = 0; i < 4; i++) vals[i] = acc[mt][nt][i] + a.bias[c0 + i];
                if (ACT == 1) {
#pragma unroll
                    for (int i = 0; i < 4; i++) vals[i] = vals[i] * __frcp_rn(1.0f + __expf(-vals[i]));
                }
                if constexpr (OUT_F32) {
                    *reinterpret_cast<f32x4 *>(static_cast<float *>(a.out) + m * a.out_ld + c0) =
                        (f32x4){vals[0], vals[1], vals[2], vals[3]};
                } else {
                    if (a.res) {
                        const half4 rv = *reinterpret_cast<const half4 *>(a.res + m * a.res_ld + c0);
#pragma unroll
                        for (int i = 0; i < 4; i++) vals[i] += (float)rv[i];
                    }
                    *reinterpret_cast<half4 *>(static_cast<half_t *>(a.out) + m * a.out_ld + c0) =
                        (half4){(half_t)vals[0], (half_t)vals[1], (half_t)vals[2], (half_t)vals[3]};
                }
            }
        }
    }
}

template <int KS, int STRIDE, int MT, int NT, bool CIN16, int ACT, bool OUT_F32>
static void launch_inst(const ConvArgs &a, hipStream_t s)
{
    const int tiles = (a.M + 15) / 16;
    const int bx = (tiles + 4 * MT - 1) / (4 * MT);
    const int by = a.cout_pad / (16 * NT);
    hipLaunchKernelGGL((conv_mfma_kernel<KS, STRIDE, MT, NT, CIN16, ACT, OUT_F32>), dim3(bx, by), dim3(256), 0, s, a);
}

template <int KS, int STRIDE, int NT, bool CIN16, int ACT, bool OUT_F32>
static void launch_mt(int mt, const ConvArgs &a, hipStream_t s)
{
    if (mt == 2) launch_inst<KS, STRIDE, 2, NT, CIN16, ACT, OUT_F32>(a, s);
    else launch_inst<KS, STRIDE, 1, NT, CIN16, ACT, OUT_F32>(a, s);
}

bool launch_conv(const ConvCfg &c, const ConvArgs &a, hipStream_t s)
{
    if (c.mt != 1 && c.mt != 2) return false;
    if (a.cout_pad % (16 * c.nt) != 0) return false;
#define IRMV_CASE(KS_, ST_, NT_, C16_, ACT_, F32_)                                                        \
    if (c.ks == KS_ && c.stride == ST_ && c.nt == NT_ && c.cin16 == C16_ && c.act == ACT_ && c.out_f32 == F32_) { \
        launch_mt<KS_, ST_, NT_, C16_, ACT_, F32_>(c.mt, a, s);                                           \
        return true;                                                                                      \
    }
    // 3x3 stride 1, SiLU, fp16 out
    IRMV_CASE(3, 1, 1, false, 1, false)
    IRMV_CASE(3, 1, 1, true, 1, false)
    IRMV_CASE(3, 1, 2, false, 1, false)
    IRMV_CASE(3, 1, 4, false, 1, false)
    // 3x3 stride 2
    IRMV_CASE(3, 2, 2, true, 1, false)
    IRMV_CASE(3, 2, 2, false, 1, false)
    IRMV_CASE(3, 2, 4, false, 1, false)
    // 1x1 SiLU
    IRMV_CASE(1, 1, 2, false, 1, false)
    IRMV_CASE(1, 1, 4, false, 1, false)
    // 1x1 head finals: bias only, fp32 out
    IRMV_CASE(1, 1, 1, false, 0, true)
    IRMV_CASE(1, 1, 4, false, 0, true)
#undef IRMV_CASE
    return false;
}

const char *conv_cfg_name(const ConvCfg &c, char *buf, int n)
{
    snprintf(buf, n, "conv%dx%ds%d_mt%d_nt%d%s%s", c.ks, c.ks, c.stride, c.mt, c.nt, c.cin16 ? "_c16" : "",
             c.out_f32 ? "_f32" : "");
    return buf;
}

// SPPF (SURVEY.md Appendix A "Blocks"): p1 = maxpool5(a), p2 = maxpool5(p1),
// p3 = maxpool5(p2).  Stride-1 max pools compose, so p2 / p3 are the clipped 9x9 /
// 13x13 window maxima of `a`: one kernel, one read of the 13x13 neighbourhood,
// three nested maxima.  One lane per (pixel, 8-channel chunk).
__global__ __launch_bounds__(256) void sppf_pool_kernel(half_t *buf, int batch, int H, int W, int C)
{
    const int chunks = C >> 3;
    const int total = batch * H * W * chunks;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int ch = t % chunks, p = t / chunks;
    const int b = p / (H * W), rem = p - b * H * W, oy = rem / W, ox = rem - oy * W;
    const int ld = 4 * C;
    const half_t *base = buf + (size_t)b * H * W * ld + ch * 8;
    const half_t ninf = (half_t)(-65504.0f);
    half8 m5, m9, m13;
#pragma unroll
    for (int i = 0; i < 8; i++) m5[i] = m9[i] = m13[i] = ninf;
    for (int dy = -6; dy <= 6; dy++) {
        const int iy = oy + dy;
        if ((unsigned)iy >= (unsigned)H) continue;
        const int ady = dy < 0 ? -dy : dy;
        for (int dx = -6; dx <= 6; dx++) {
            const int ix = ox + dx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const int adx = dx < 0 ? -dx : dx;
            const int rad = ady > adx ? ady : adx;
            const half8 v = *reinterpret_cast<const half8 *>(base + ((size_t)iy * W + ix) * ld);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                m13[i] = v[i] > m13[i] ? v[i] : m13[i];
                if (rad <= 4) m9[i] = v[i] > m9[i] ? v[i] : m9[i];
                if (rad <= 2) m5[i] = v[i] > m5[i] ? v[i] : m5[i];
            }
        }
    }
    half_t *o = buf + ((size_t)b * H * W + (size_t)oy * W + ox) * ld + ch * 8;
    *reinterpret_cast<half8 *>(o + C) = m5;
    *reinterpret_cast<half8 *>(o + 2 * C) = m9;
    *reinterpret_cast<half8 *>(o + 3 * C) = m13;
}

void launch_sppf_pool(half_t *buf, int batch, int H, int W, int C, hipStream_t s)
{
    const int total = batch * H * W * (C >> 3);
    hipLaunchKernelGGL(sppf_pool_kernel, dim3((total + 255) / 256), dim3(256), 0, s, buf, batch, H, W, C);
}

}  // namespace irmv
