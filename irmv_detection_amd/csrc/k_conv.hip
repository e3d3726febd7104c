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
#include <mutex>
#include <type_traits>

namespace irmv {

// Prefetch depth of the register ring: small tiles are latency-bound (few waves per
// CU on the 20x20 / 40x40 layers), so they keep more k-steps of loads in flight.
template <int MT, int NT>
struct PrefetchDepth { static constexpr int value = (MT * NT >= 8) ? 2 : ((MT * NT >= 4) ? 3 : 4); };
// Latency variant (single-frame steps: a handful of workgroups per layer, nothing to hide a load behind but the wave's own
// prefetch): the ring holds 6..12 k-steps, so a K = 1152 layer pays 3 memory round trips instead of 9.
template <int MT, int NT>
struct DeepPrefetchDepth { static constexpr int value = (MT + NT <= 2) ? 12 : ((MT + NT <= 3) ? 10 : 6); };

// CT: walk K chunk-major -- k-step s = (32-channel chunk s / 9, tap s % 9), the order of the LDS-staged family below -- with
// weights in that family's nt = 1 packing, so a 3x3 layer of the LDS family can run on this kernel bit-identically.
// (An XCD-aware workgroup order -- image i's tiles on one XCD in every layer -- measured -0.3 % in round 2, within noise:
// at 64 frames per graph a layer's tensors are 3 - 50 MB against 4 MB of L2 per XCD.  Removed in round 3.)

template <int KS, int STRIDE, int MT, int NT, bool CIN16, int ACT, bool OUT_F32, bool CT = false, bool DEEP = false>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs &a, int wg_x, int wg_y, int grid_x, int grid_y)
{
    constexpr int PAD = KS / 2;
    constexpr int PF = DEEP ? DeepPrefetchDepth<MT, NT>::value : PrefetchDepth<MT, NT>::value;
    static_assert(!CT || (KS == 3 && !CIN16), "chunk-major order is the 3x3 LDS family's");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int bx = wg_x, by = wg_y;
    (void)grid_x; (void)grid_y;
    const int tile0 = (bx * 4 + wave) * MT;
    const int nt0 = by * NT;
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

    // bias of this lane's output channels, fetched with the first operands: read inside the epilogue it costs a memory
    // round trip per output block (a load after a store waits for the store: bias and output may alias for all the
    // compiler knows).  SiLU layers: the accumulators START at the (log2 e-scaled) bias -- no add in the epilogue.
    f32x4 bs[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int t = nt0 + nt;
        bs[nt] = *reinterpret_cast<const f32x4 *>(a.bias + (a.pair ? ((t >> 1) * 32 + g * 8 + (t & 1) * 4) : (t * 16 + g * 4)));
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[mt][nt] = ACT == 1 ? bs[nt] : (f32x4){0.f, 0.f, 0.f, 0.f};
    // Cin = 16 layers with ONE 16-channel output tile (the keypoint branch's second 3x3): the branch's final 1x1 (16 -> nk, bias
    // only, fp32 out) can ride in the epilogue (a.n2 = 1) -- the lane's four activated outputs, channels 4 g .. 4 g + 3 of its
    // pixel, ARE the B fragment of v_mfma_f32_16x16x16_f16 over those 16 channels.  Its operands are fetched here, with the
    // first k-steps (fetched in the epilogue they would cost a memory round trip behind the main loop).
    constexpr bool K16_FUSE = CIN16 && NT == 1 && ACT == 1 && !OUT_F32;
    half4 w2f = (half4){0, 0, 0, 0};
    f32x4 b2f = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (K16_FUSE) {
        if (a.n2 > 0) {
            w2f = *reinterpret_cast<const half4 *>(a.w2 + lane * 4);
            b2f = *reinterpret_cast<const f32x4 *>(a.bias2 + g * 4);
        }
    }

    const half8 *wp = reinterpret_cast<const half8 *>(a.w) + (size_t)nt0 * a.ksteps * 64 + lane;
    const int H0 = a.Hin >> a.s0.shift, W0 = a.Win >> a.s0.shift;
    const int H1 = a.Hin >> a.s1.shift, W1 = a.Win >> a.s1.shift;

    // ---- loader state: walks the k-steps in order, PF steps ahead of the MFMAs ----
    int l_ks = 0, l_tap = 0, l_cc = 0;
    const half_t *p0[MT], *p1[MT];
    bool pv[MT];
    auto set_tap = [&](int tap) {
        const int kh = tap / KS, kw = tap - kh * KS;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int iy = iy0[mt] + kh, ix = ix0[mt] + kw;
            pv[mt] = mv[mt] && tap < KS * KS && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            const int iyc = pv[mt] ? iy : 0, ixc = pv[mt] ? ix : 0;
            p0[mt] = a.s0.p + ((size_t)(bb[mt] * H0 + (iyc >> a.s0.shift)) * W0 + (ixc >> a.s0.shift)) * a.s0.ld;
            if constexpr (!CIN16)
                p1[mt] = a.s1.p + ((size_t)(bb[mt] * H1 + (iyc >> a.s1.shift)) * W1 + (ixc >> a.s1.shift)) * a.s1.ld;
        }
    };
    // Every load of the ring is UNCONDITIONAL: from a clamped, always valid address (set_tap: pixel (0, 0) of the image for
    // taps in the padding and for rows past M; the last 8 channels for K positions past Cin; the last k-step again past
    // the end), as four dwords.  What must be zero (padding taps of a 3x3; K positions past Cin) is zeroed by a select
    // where the fragment is USED.  Behind `if (valid) B = load` each load sits in its own branch, the compiler can no
    // longer count the loads in flight and every wait becomes vmcnt(0): the prefetch ring below then waits, at each
    // k-step, for the loads it has just issued -- one exposed memory round trip per k-step.
    // 1x1 convs need no select at all: there is no padding, a row past M fills a B column whose results are never stored,
    // and K positions past Cin meet zero weights (pack_conv) with finite activations.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    half8 Ab[PF][NT];
    u32x4 Bb[PF][MT];
    bool Bz[PF][MT];   // fragment must read as zero
    auto load_step = [&](half8 (&A)[NT], u32x4 (&B)[MT], bool (&Z)[MT]) {
        const int ks = l_ks < a.ksteps ? l_ks : a.ksteps - 1;
        if constexpr (CIN16) {
            // Cin == 16: one k-step of 32 spans TWO filter taps (lanes g<2: tap 2i, g>=2: tap 2i+1)
            set_tap(2 * ks + (g >> 1));
            const int c = 8 * (g & 1);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                B[mt] = *reinterpret_cast<const u32x4 *>(p0[mt] + c);
                Z[mt] = !pv[mt];
            }
        } else {
            if constexpr (CT) { set_tap(ks % 9); l_cc = (ks / 9) * 32; }
            const int c = l_cc + 8 * g;
            const int cs = c < a.Cin ? c : a.Cin - 8;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const half_t *q = (cs < a.s0.C) ? (p0[mt] + cs) : (p1[mt] + (cs - a.s0.C));
                B[mt] = *reinterpret_cast<const u32x4 *>(q);
                Z[mt] = !(pv[mt] && c < a.Cin);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) A[nt] = wp[(size_t)(nt * a.ksteps + ks) * 64];
        // advance
        l_ks++;
        if constexpr (!CIN16 && !CT) {
            l_cc += 32;
            if (l_cc >= a.Cin) {
                l_cc = 0;
                l_tap++;
                set_tap(l_tap);
            }
        }
    };

    if constexpr (!CIN16 && !CT) set_tap(0);
#pragma unroll
    for (int i = 0; i < PF; i++) load_step(Ab[i], Bb[i], Bz[i]);
    auto mma_slot = [&](int i) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const u32x4 bq = (KS == 1) ? Bb[i][mt] : (Bz[i][mt] ? (u32x4){0u, 0u, 0u, 0u} : Bb[i][mt]);
            const half8 bf = __builtin_bit_cast(half8, bq);
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ab[i][nt], bf, acc[mt][nt], 0, 0, 0);
        }
    };
    // whole rounds of the ring without a branch (a branch around a load makes the wait counts conservative again), then
    // the remainder
    int s = 0;
    for (; s + PF <= a.ksteps; s += PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            mma_slot(i);
            load_step(Ab[i], Bb[i], Bz[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < PF; i++)
        if (s + i < a.ksteps) mma_slot(i);

    // ---- epilogue: bias, SiLU, shortcut, convert, NHWC store ----
    // D layout of 16x16x32: col = lane & 15 (pixel), row = (lane >> 4) * 4 + reg (cout row of the tile).
    // a.pair: the host packed the weight rows so that tiles (2u, 2u+1) interleave in groups of 4
    // channels: lane g of tile t holds channels (t>>1)*32 + g*8 + (t&1)*4 + [0,4).
    if constexpr (K16_FUSE) {
        if (a.n2 > 0) {   // (kernel-uniform) fused final 1x1: the MFMA runs with every lane -- its A rows live in all 64 -- and only the stores are masked
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                half4 o = silu_pack4(acc[mt][0][0], acc[mt][0][1], acc[mt][0][2], acc[mt][0][3]);
                mfma_operand_fence(o);   // (VALU write -> MFMA read: two wait states the compiler cannot see through the inline asm)
                const f32x4 c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(w2f, o, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const f32x4 v2 = (f32x4){c2[0] * kActUnscale + b2f[0], c2[1] * kActUnscale + b2f[1], c2[2] * kActUnscale + b2f[2], c2[3] * kActUnscale + b2f[3]};   // (the 1x1's input carries the activation scale)
                if (mv[mt]) *reinterpret_cast<f32x4 *>(a.out2 + (size_t)mm[mt] * a.out2_ld + g * 4) = v2;
            }
            return;
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        if (!mv[mt]) continue;
        const size_t m = (size_t)mm[mt];
        if constexpr (!OUT_F32 && (NT % 2 == 0)) {
            if (a.pair) {
#pragma unroll
                for (int u = 0; u < NT / 2; u++) {
                    const int c0 = (nt0 / 2 + u) * 32 + g * 8;
                    float vals[8];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        vals[i] = ACT == 1 ? acc[mt][2 * u][i] : acc[mt][2 * u][i] * kActUnscale + bs[2 * u][i];
                        vals[4 + i] = ACT == 1 ? acc[mt][2 * u + 1][i] : acc[mt][2 * u + 1][i] * kActUnscale + bs[2 * u + 1][i];
                    }
                    half8 o;   // (rounding pinned: irmv_common.hpp)
                    if (ACT == 1 && a.res) {
                        const half8 rv = *reinterpret_cast<const half8 *>(a.res + m * a.res_ld + c0);
#pragma unroll
                        for (int i = 0; i < 8; i++) o[i] = silu_add_res(vals[i], (float)rv[i]);
                    } else if (ACT == 1) {
                        o = silu_pack8(vals[0], vals[1], vals[2], vals[3], vals[4], vals[5], vals[6], vals[7]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; i++) o[i] = (half_t)vals[i];
                    }
                    *reinterpret_cast<half8 *>(static_cast<half_t *>(a.out) + m * a.out_ld + c0) = o;
                }
                continue;
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int t = nt0 + nt;
            const int c0 = a.pair ? ((t >> 1) * 32 + g * 8 + (t & 1) * 4) : (t * 16 + g * 4);
            float vals[4];
#pragma unroll
            for (int i = 0; i < 4; i++) vals[i] = ACT == 1 ? acc[mt][nt][i] : acc[mt][nt][i] * kActUnscale + bs[nt][i];   // (no activation: the Detect finals -- undo the activation scale)
            if constexpr (OUT_F32) {
                static_assert(ACT == 0, "fp32 outputs are the Detect finals: no activation");
                *reinterpret_cast<f32x4 *>(static_cast<float *>(a.out) + m * a.out_ld + c0) =
                    (f32x4){vals[0], vals[1], vals[2], vals[3]};
            } else {
                half4 o;   // (rounding pinned: irmv_common.hpp)
                if (ACT == 1 && a.res) {
                    const half4 rv = *reinterpret_cast<const half4 *>(a.res + m * a.res_ld + c0);
#pragma unroll
                    for (int i = 0; i < 4; i++) o[i] = silu_add_res(vals[i], (float)rv[i]);
                } else if (ACT == 1) {
                    o = silu_pack4(vals[0], vals[1], vals[2], vals[3]);
                } else {
                    o = (half4){(half_t)vals[0], (half_t)vals[1], (half_t)vals[2], (half_t)vals[3]};
                }
                *reinterpret_cast<half4 *>(static_cast<half_t *>(a.out) + m * a.out_ld + c0) = o;
            }
        }
    }
}

template <int KS, int STRIDE, int MT, int NT, bool CIN16, int ACT, bool OUT_F32, bool CT = false, bool DEEP = false>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a)
{
    conv_mfma_body<KS, STRIDE, MT, NT, CIN16, ACT, OUT_F32, CT, DEEP>(a, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}

// several independent layers of one kernel shape in one launch (see conv3x3_lds_multi)
template <int KS, int STRIDE, int MT, int NT, bool CIN16, int ACT, bool OUT_F32>
__global__ __launch_bounds__(256) void conv_mfma_multi(DirectMultiArgs m)
{
    int id = blockIdx.x, k = 0;
    while (k + 1 < m.n && id >= m.start[k + 1]) k++;
    id -= m.start[k];
    const int gx = m.gx[k], gy = m.gy[k];
    conv_mfma_body<KS, STRIDE, MT, NT, CIN16, ACT, OUT_F32>(m.a[k], id % gx, id / gx, gx, gy);
}

template <int KS, int STRIDE, int MT, int NT, bool CIN16, int ACT, bool OUT_F32, bool CT = false, bool DEEP = false>
static void launch_inst(const ConvArgs &a, hipStream_t s)
{
    const int tiles = (a.M + 15) / 16;
    const int bx = (tiles + 4 * MT - 1) / (4 * MT);
    const int by = a.cout_pad / (16 * NT);
    hipLaunchKernelGGL((conv_mfma_kernel<KS, STRIDE, MT, NT, CIN16, ACT, OUT_F32, CT, DEEP>), dim3(bx, by), dim3(256), 0, s, a);
}

template <int KS, int STRIDE, bool CIN16, int ACT, bool OUT_F32>
static bool launch_tile(int mt, int nt, const ConvArgs &a, hipStream_t s)
{
#define IRMV_TILE(MT_, NT_)                                                   \
    if (mt == MT_ && nt == NT_) {                                             \
        launch_inst<KS, STRIDE, MT_, NT_, CIN16, ACT, OUT_F32>(a, s);         \
        return true;                                                          \
    }
    IRMV_TILE(1, 1) IRMV_TILE(2, 1) IRMV_TILE(4, 1)
    IRMV_TILE(1, 2) IRMV_TILE(2, 2) IRMV_TILE(4, 2)
    IRMV_TILE(1, 4) IRMV_TILE(2, 4) IRMV_TILE(4, 4)
#undef IRMV_TILE
    return false;
}

// latency variants (deep prefetch): small tiles only
template <int KS, int STRIDE, int ACT, bool OUT_F32, bool CT>
static bool launch_deep(int mt, int nt, const ConvArgs &a, hipStream_t s)
{
#define IRMV_DEEP(MT_, NT_)                                                            \
    if (mt == MT_ && nt == NT_) {                                                      \
        launch_inst<KS, STRIDE, MT_, NT_, false, ACT, OUT_F32, CT, true>(a, s);        \
        return true;                                                                   \
    }
    IRMV_DEEP(1, 1) IRMV_DEEP(2, 1) IRMV_DEEP(1, 2) IRMV_DEEP(1, 4)
#undef IRMV_DEEP
    return false;
}

bool launch_conv(const ConvCfg &c, const ConvArgs &a, hipStream_t s)
{
    if (a.cout_pad % (16 * c.nt) != 0) return false;
    if (c.deep) {
        // a.w must be the packing that matches the K order: direct packing, or (chunk-major) the LDS family's nt = 1 packing
        if (c.ks == 3 && c.stride == 1 && c.act == 1 && !c.out_f32 && !c.cin16) return c.ct ? launch_deep<3, 1, 1, false, true>(c.mt, c.nt, a, s) : launch_deep<3, 1, 1, false, false>(c.mt, c.nt, a, s);
        if (c.ks == 3 && c.stride == 2 && c.act == 1 && !c.out_f32 && !c.cin16) return c.ct ? launch_deep<3, 2, 1, false, true>(c.mt, c.nt, a, s) : launch_deep<3, 2, 1, false, false>(c.mt, c.nt, a, s);
        if (c.ks == 1 && c.stride == 1 && c.act == 1 && !c.out_f32 && !c.ct) return launch_deep<1, 1, 1, false, false>(c.mt, c.nt, a, s);
        return false;
    }
    if (c.ct) {   // chunk-major order on the standard prefetch ring: stride-2 3x3 layers of the LDS family
        if (!(c.ks == 3 && c.stride == 2 && c.act == 1 && !c.out_f32 && !c.cin16)) return false;
#define IRMV_CT(MT_, NT_)                                                        \
        if (c.mt == MT_ && c.nt == NT_) {                                        \
            launch_inst<3, 2, MT_, NT_, false, 1, false, true, false>(a, s);     \
            return true;                                                         \
        }
        IRMV_CT(1, 1) IRMV_CT(2, 1) IRMV_CT(4, 1) IRMV_CT(1, 2) IRMV_CT(2, 2) IRMV_CT(4, 2) IRMV_CT(1, 4) IRMV_CT(2, 4) IRMV_CT(4, 4)
#undef IRMV_CT
        return false;
    }
#define IRMV_CASE(KS_, ST_, C16_, ACT_, F32_)                                                             \
    if (c.ks == KS_ && c.stride == ST_ && c.cin16 == C16_ && c.act == ACT_ && c.out_f32 == F32_)          \
        return launch_tile<KS_, ST_, C16_, ACT_, F32_>(c.mt, c.nt, a, s);
    IRMV_CASE(3, 1, false, 1, false)   // 3x3 stride 1, SiLU, fp16 out
    IRMV_CASE(3, 1, true, 1, false)
    IRMV_CASE(3, 2, false, 1, false)   // 3x3 stride 2
    IRMV_CASE(3, 2, true, 1, false)
    IRMV_CASE(1, 1, false, 1, false)   // 1x1 SiLU
    IRMV_CASE(1, 1, false, 0, true)    // 1x1 head finals: bias only, fp32 out
#undef IRMV_CASE
    return false;
}

// The keypoint branch's final 1x1 (16 -> nk <= 16, bias only, fp32 out) as its own launch: ONE v_mfma_f32_16x16x16_f16 per 16
// pixels -- the instruction, operands and rounding of the fused form above (conv_mfma_body, K16_FUSE), so that the fused
// and the unfused head agree bit for bit.  a.w2 = the layer's weights in that instruction's A layout (lane (g, r): output
// channel r, input channels 4 g .. 4 g + 3), a.bias = its bias; input at the layer's own resolution, one segment.
__global__ __launch_bounds__(256) void conv1x1_k16_f32_kernel(ConvArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int m = (blockIdx.x * 4 + wave) * 16 + r;
    const bool mv = m < a.M;
    const half4 w = *reinterpret_cast<const half4 *>(a.w2 + lane * 4);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(a.bias + g * 4);
    const half4 x = *reinterpret_cast<const half4 *>(a.s0.p + (size_t)(mv ? m : 0) * a.s0.ld + g * 4);
    const f32x4 c = __builtin_amdgcn_mfma_f32_16x16x16f16(w, x, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    if (mv) *reinterpret_cast<f32x4 *>(static_cast<float *>(a.out) + (size_t)m * a.out_ld + g * 4) =
        (f32x4){c[0] * kActUnscale + b[0], c[1] * kActUnscale + b[1], c[2] * kActUnscale + b[2], c[3] * kActUnscale + b[3]};
}

bool launch_conv_k16(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin != 16 || a.cout_pad != 16 || !a.w2 || a.s1.C != 0 || a.s0.shift != 0) return false;
    hipLaunchKernelGGL(conv1x1_k16_f32_kernel, dim3(((a.M + 15) / 16 + 3) / 4), dim3(256), 0, s, a);
    return true;
}

bool launch_conv_direct_multi(const ConvCfg &c, const ConvArgs *a, int n, hipStream_t s)
{
    if (n < 1 || n > kMultiMax || c.mt != 1 || c.nt != 1 || c.deep || c.ct || c.lds || c.pw) return false;
    DirectMultiArgs m{};
    m.n = n;
    int total = 0;
    for (int k = 0; k < n; k++) {
        if (a[k].cout_pad % 16 != 0) return false;
        m.a[k] = a[k];
        m.start[k] = total;
        m.gx[k] = ((a[k].M + 15) / 16 + 3) / 4;
        m.gy[k] = a[k].cout_pad / 16;
        total += m.gx[k] * m.gy[k];
    }
    m.start[n] = total;
    if (c.ks == 3 && c.stride == 1 && c.cin16 && c.act == 1 && !c.out_f32) {
        hipLaunchKernelGGL((conv_mfma_multi<3, 1, 1, 1, true, 1, false>), dim3(total), dim3(256), 0, s, m);
        return true;
    }
    if (c.ks == 1 && c.stride == 1 && !c.cin16 && c.act == 0 && c.out_f32) {
        hipLaunchKernelGGL((conv_mfma_multi<1, 1, 1, 1, false, 0, true>), dim3(total), dim3(256), 0, s, m);
        return true;
    }
    return false;
}

const char *conv_cfg_name(const ConvCfg &c, char *buf, int n)
{
    snprintf(buf, n, "conv%dx%ds%d_mt%d_nt%d%s%s%s%s", c.ks, c.ks, c.stride, c.mt, c.nt, c.cin16 ? "_c16" : "",
             c.out_f32 ? "_f32" : "", c.deep ? "_deep" : "", c.ct ? "_ct" : "");
    return buf;
}

// ---------------------------------------------------------------------------
// LDS-staged 3x3 convolution (stride 1 or 2, Cin % 32 == 0, SiLU, fp16 out).
//
// The direct kernel above re-reads every weight fragment in every wave and every
// input pixel once per filter tap, all through L1/L2; on the 20x20..80x80 layers
// that traffic, not the matrix cores, sets the time.  Here a workgroup owns a 2-D
// block of 64*MT output pixels of ONE image (RH rows x 16 / 8 / 4 columns, whichever
// divides the width: halo overhead ~1.4x instead of the 3x of full-width rows) and
// NT*16 output channels, and walks Cin in chunks of 32 channels.  Per chunk it
// stages in LDS, once:
//   * the input patch = the block's input pixels with halo, 32 channels
//     per pixel, 96-byte pixel stride (64 B data + 32 B pad: the b128 fragment
//     reads of the 16 lanes of a group then hit 16 distinct 16-byte slots);
//   * the weight slab of the chunk for all 9 taps, already in MFMA fragment order
//     (host-packed [n-block][chunk][tap][tile][lane][8], one contiguous 9*NT KiB).
// All four waves then read their A/B fragments from LDS with conflict-free
// ds_read_b128: weights are fetched from L2 once per workgroup instead of once
// per wave, input pixels once instead of nine times.  The next chunk's global
// loads are issued into registers before the current chunk's MFMAs (issue-early /
// write-late), so HBM/L2 latency hides under the matrix work.
// K order is (chunk, tap); zero padding = zeroed halo pixels.
// ---------------------------------------------------------------------------
// Bytes per staged pixel (32 ch fp16 = 64 B + pad), chosen so that the ds_read_b128 of a 16-lane group hit 16 distinct
// 16-byte slots: stride 1 -> consecutive lanes are consecutive pixels, 96 B (24 dwords: period 8 over the 64 banks, the two
// channel quarters of a group interleave); stride 2 -> consecutive lanes are two pixels apart, and 2 x 96 B = 48 dwords
// folds lanes r and r + 4 onto the same banks (measured: 29-39 % of the stride-2 kernels' LDS cycles were bank conflicts,
// profiles/r02_mfma.json of the round-1 layout), whereas 2 x 80 B = 40 dwords has period 8 again.
constexpr int pix_stride(int stride) { return stride == 2 ? 80 : 96; }

// hipFuncSetAttribute is per device: remember, per kernel, on which devices it has been applied.  Engines may be created
// and run from several threads (one per GPU): the check, the attribute call and the flag update are one critical section,
// so no thread launches before the limit is raised.  (The mutex is taken on every launch CALL -- engine creation, tuning,
// graph capture, eager profiling; a captured step is replayed without any of this code.)
static std::mutex g_attr_mu;
template <class F>
static void once_per_device(unsigned long long &mask, F set_attribute)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    std::lock_guard<std::mutex> lk(g_attr_mu);
    if (mask & bit) return;
    set_attribute();
    mask |= bit;
}

// ---------------------------------------------------------------------------
// Pointwise (1x1) convolution as a persistent, software-pipelined GEMM.
//
// The direct kernel above gives a wave ONE pixel tile per lifetime: index arithmetic, a cold first load, 4..16 k-steps,
// a SiLU epilogue as long as the matrix work (K <= 512), exit -- its waves sit in waits 40 % and issue stalls 40 % of
// their cycles (profiles/r02_mfma.json) and it reaches 0.26-0.34 of the HBM line its arithmetic intensity puts it under.
// Here a workgroup keeps the weights of its 64 output channels in LDS (NT x KS KiB, fragment order: conflict-free
// ds_read_b128) and its four waves walk pixel tiles of 32 pixels with a stride of the whole grid.  Activations go
// straight from memory into B fragments, and every fragment register is re-loaded for the wave's NEXT tile right after
// the MFMAs that consumed it: a tile's loads are in flight for a whole tile time, under the MFMAs and the SiLU epilogue
// of the tile before.  Same operands, same k order as the direct kernel -> bit-identical, so the autotuner may pick either.
// ---------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void conv1x1_pw_kernel(ConvArgs a, int tiles_total, int wg_per_nblock)
{
    constexpr int MT = 2, NT = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8 *s_w = reinterpret_cast<half8 *>(smem);   // [NT][KS][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int nblk = blockIdx.y;
    // weights -> LDS: every piece of a thread is loaded before the first is stored (NT * KS * 64 / 256 = KS pieces; a
    // load -> store loop serialises KS memory round trips at the head of every workgroup)
    half8 wreg[KS];
    {
        const half8 *wsrc = reinterpret_cast<const half8 *>(a.w) + (size_t)nblk * NT * KS * 64;
#pragma unroll
        for (int i = 0; i < KS; i++) wreg[i] = wsrc[tid + i * 256];
    }
    const int HWo = a.Hout * a.Wout;
    const int H0 = a.Hin >> a.s0.shift, W0 = a.Win >> a.s0.shift, H1 = a.Hin >> a.s1.shift, W1 = a.Win >> a.s1.shift;
    const bool flat = a.s0.shift == 0 && a.s1.shift == 0;
    float bias[2][8];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int i = 0; i < 8; i++) bias[u][i] = a.bias[(nblk * 2 + u) * 32 + g * 8 + i];

    // This wave's share of the image: a CONTIGUOUS range of 16-pixel units, the same number for every wave of the launch
    // give or take one (a fixed stride of 32-pixel tiles left a quarter of the waves with a fourth tile where the others
    // had three: the launch lasted as long as they did).  The range is walked in 32-pixel tiles; an odd unit at its end
    // is a half tile with its own copy of the tile body (MT = 1: no loads, MFMAs or epilogue for the missing half).
    const int units = (a.M + 15) >> 4, nw = wg_per_nblock * 4, wv = blockIdx.x * 4 + wave;
    const int u0 = (int)((long long)wv * units / nw), u1 = (int)((long long)(wv + 1) * units / nw);
    (void)tiles_total;
    // unit ub .. ub + MT - 1: source pointers of both K segments (a unit past this wave's range reads pixel 0: never stored)
    const half_t *p0[MT], *p1[MT];
    bool mv[MT];
    auto tile_ptrs = [&](int ub) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int m = (ub + mt) * 16 + r;
            mv[mt] = ub + mt < u1 && m < a.M;
            const int mm = mv[mt] ? m : 0;
            if (flat) {   // no half-resolution segment (every layer but the neck's upsample + concat readers): pixel m of the output
                          // is pixel m of both segments -- no (image, row, column) split: two integer divisions per unit less
                p0[mt] = a.s0.p + (size_t)mm * a.s0.ld + 8 * g;
                p1[mt] = a.s1.p + (size_t)mm * a.s1.ld + 8 * g - a.s0.C;
            } else {
                const int b = mm / HWo, rem = mm - b * HWo;
                const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
                p0[mt] = a.s0.p + ((size_t)(b * H0 + (oy >> a.s0.shift)) * W0 + (ox >> a.s0.shift)) * a.s0.ld + 8 * g;
                p1[mt] = a.s1.p + ((size_t)(b * H1 + (oy >> a.s1.shift)) * W1 + (ox >> a.s1.shift)) * a.s1.ld + 8 * g - a.s0.C;
            }
        }
    };
    // UNCONDITIONAL: a pixel past the end reads pixel 0 (tile_ptrs clamps) into a B column whose results are never stored.
    // Behind `if (!mv) return zero` every load sits in its own branch, the compiler can no longer count the loads in
    // flight, and each wait becomes vmcnt(0): the fragment just re-loaded for the NEXT tile is waited for on the spot --
    // one exposed memory round trip per k-step instead of none.
    auto load_b = [&](int mt, int ks) -> half8 {
        const int c = ks * 32;                                   // + 8 g is in the pointers
        return *reinterpret_cast<const half8 *>((c < a.s0.C ? p0[mt] : p1[mt]) + c);
    };

    half8 B[MT][KS];
    tile_ptrs(u0);
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) B[mt][ks] = load_b(mt, ks);
#pragma unroll
    for (int i = 0; i < KS; i++) s_w[tid + i * 256] = wreg[i];
    __syncthreads();                                             // weights staged
    half_t *out = static_cast<half_t *>(a.out);
    // one tile: MTA active 16-pixel units; PRE: re-load every consumed fragment for the tile that follows
    auto run_tile = [&](auto mta_c, auto pre_c, int ub) {
        constexpr int MTA = decltype(mta_c)::value;
        constexpr bool PRE = decltype(pre_c)::value;
        size_t m_cur[MTA];
        bool mv_cur[MTA];
#pragma unroll
        for (int mt = 0; mt < MTA; mt++) { m_cur[mt] = (size_t)(ub + mt) * 16 + r; mv_cur[mt] = mv[mt]; }
        if constexpr (PRE) tile_ptrs(ub + MT);                   // from here p0 / p1 / mv describe the NEXT tile
        f32x4 acc[MTA][NT];   // start at the bias: tile 2 u + h, register i <-> channel u * 32 + g * 8 + 4 h + i
#pragma unroll
        for (int mt = 0; mt < MTA; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[mt][nt] = (f32x4){bias[nt >> 1][(nt & 1) * 4 + 0], bias[nt >> 1][(nt & 1) * 4 + 1], bias[nt >> 1][(nt & 1) * 4 + 2], bias[nt >> 1][(nt & 1) * 4 + 3]};
        // A fragments from LDS, double-buffered one k-step ahead; the scheduling barrier keeps the compiler from hoisting
        // all KS x NT fragment reads to the top of the unrolled loop (256 VGPRs and spills without it)
        half8 A[2][NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) A[0][nt] = s_w[(nt * KS) * 64 + lane];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++) A[(ks + 1) & 1][nt] = s_w[(nt * KS + ks + 1) * 64 + lane];
            }
#pragma unroll
            for (int mt = 0; mt < MTA; mt++) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks & 1][nt], B[mt][ks], acc[mt][nt], 0, 0, 0);
                if constexpr (PRE) B[mt][ks] = load_b(mt, ks);   // consumed: fetch the next tile's fragment into the same registers
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue of the direct kernel's paired-tile path: lane g holds channels u * 32 + g * 8 + [0, 8) of its pixel
#pragma unroll
        for (int mt = 0; mt < MTA; mt++) {
            if (!mv_cur[mt]) continue;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                float vals[8];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    vals[i] = acc[mt][2 * u][i];
                    vals[4 + i] = acc[mt][2 * u + 1][i];
                }
                const half8 o = silu_pack8(vals[0], vals[1], vals[2], vals[3], vals[4], vals[5], vals[6], vals[7]);
                *reinterpret_cast<half8 *>(out + m_cur[mt] * a.out_ld + (nblk * 2 + u) * 32 + g * 8) = o;
            }
        }
    };
    int ub = u0;
    for (; ub + MT <= u1; ub += MT) run_tile(std::integral_constant<int, MT>{}, std::true_type{}, ub);
    if (ub < u1) run_tile(std::integral_constant<int, 1>{}, std::false_type{}, ub);
}

// Pointwise kernel, several output-channel blocks per workgroup (NBW blocks of 64 channels): the kernel above gives every
// 64-channel block its own workgroups, so a layer with 128 / 256 output channels reads its input two / four times (from L2
// the second time on, but every read is a request the CU's vector-memory path has to carry).  Here ONE 8-wave workgroup
// keeps the weights of NBW blocks in LDS (NBW x KS x 4 KiB: up to 128 KiB, one workgroup per CU) and a wave runs the B
// fragments of its pixel tile -- loaded ONCE -- against all of them, block after block; the fragments of the wave's next
// tile are requested during the last block's pass.  Per output channel the k order is the kernel's above -> bit-identical.
template <int KS, int NBW>
__global__ __launch_bounds__(512) void conv1x1_pwn_kernel(ConvArgs a, int wg_per_group)
{
    constexpr int MT = 2, NT = 4, NTH = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8 *s_w = reinterpret_cast<half8 *>(smem);                                   // [NBW][NT][KS][64 lanes]
    float *s_bias = reinterpret_cast<float *>(smem + (size_t)NBW * NT * KS * 1024); // [NBW][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int nblk0 = blockIdx.y * NBW;
    constexpr int WALL = NBW * NT * KS * 64, WPT = (WALL + NTH - 1) / NTH;          // half8 pieces, per thread (<= 16)
    half8 wreg[WPT];
    {
        const half8 *wsrc = reinterpret_cast<const half8 *>(a.w) + (size_t)nblk0 * NT * KS * 64;
#pragma unroll
        for (int i = 0; i < WPT; i++) {
            const int e = tid + i * NTH;
            wreg[i] = wsrc[e < WALL ? e : WALL - 1];
        }
    }
    if (tid < NBW * 64) s_bias[tid] = a.bias[nblk0 * 64 + tid];
    const int HWo = a.Hout * a.Wout;
    const int H0 = a.Hin >> a.s0.shift, W0 = a.Win >> a.s0.shift, H1 = a.Hin >> a.s1.shift, W1 = a.Win >> a.s1.shift;
    const bool flat = a.s0.shift == 0 && a.s1.shift == 0;
    const int units = (a.M + 15) >> 4, nw = wg_per_group * 8, wv = blockIdx.x * 8 + wave;
    const int u0 = (int)((long long)wv * units / nw), u1 = (int)((long long)(wv + 1) * units / nw);
    const half_t *p0[MT], *p1[MT];
    bool mv[MT];
    auto tile_ptrs = [&](int ub) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int m = (ub + mt) * 16 + r;
            mv[mt] = ub + mt < u1 && m < a.M;
            const int mm = mv[mt] ? m : 0;
            if (flat) {   // no half-resolution segment (every layer but the neck's upsample + concat readers): pixel m of the output
                          // is pixel m of both segments -- no (image, row, column) split: two integer divisions per unit less
                p0[mt] = a.s0.p + (size_t)mm * a.s0.ld + 8 * g;
                p1[mt] = a.s1.p + (size_t)mm * a.s1.ld + 8 * g - a.s0.C;
            } else {
                const int b = mm / HWo, rem = mm - b * HWo;
                const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
                p0[mt] = a.s0.p + ((size_t)(b * H0 + (oy >> a.s0.shift)) * W0 + (ox >> a.s0.shift)) * a.s0.ld + 8 * g;
                p1[mt] = a.s1.p + ((size_t)(b * H1 + (oy >> a.s1.shift)) * W1 + (ox >> a.s1.shift)) * a.s1.ld + 8 * g - a.s0.C;
            }
        }
    };
    auto load_b = [&](int mt, int ks) -> half8 {   // unconditional (see conv1x1_pw_kernel)
        const int c = ks * 32;
        return *reinterpret_cast<const half8 *>((c < a.s0.C ? p0[mt] : p1[mt]) + c);
    };
    half8 B[MT][KS];
    tile_ptrs(u0);
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) B[mt][ks] = load_b(mt, ks);
#pragma unroll
    for (int i = 0; i < WPT; i++) {
        const int e = tid + i * NTH;
        if (e < WALL) s_w[e] = wreg[i];
    }
    __syncthreads();                                             // weights and biases staged
    half_t *out = static_cast<half_t *>(a.out);
    auto run_tile = [&](auto mta_c, auto pre_c, int ub) {
        constexpr int MTA = decltype(mta_c)::value;
        constexpr bool PRE = decltype(pre_c)::value;
        size_t m_cur[MTA];
        bool mv_cur[MTA];
#pragma unroll
        for (int mt = 0; mt < MTA; mt++) { m_cur[mt] = (size_t)(ub + mt) * 16 + r; mv_cur[mt] = mv[mt]; }
        if constexpr (PRE) tile_ptrs(ub + MT);                   // from here p0 / p1 / mv describe the NEXT tile
#pragma unroll
        for (int nb = 0; nb < NBW; nb++) {
            const half8 *s_wb = s_w + nb * (NT * KS * 64) + lane;
            f32x4 acc[MTA][NT];   // start at the block's bias (LDS): tile 2 u + h, register i <-> channel u * 32 + g * 8 + 4 h + i
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(s_bias + nb * 64 + (nt >> 1) * 32 + g * 8 + (nt & 1) * 4);
#pragma unroll
                for (int mt = 0; mt < MTA; mt++) acc[mt][nt] = b;
            }
            half8 A[2][NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) A[0][nt] = s_wb[(nt * KS) * 64];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                if (ks + 1 < KS) {
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) A[(ks + 1) & 1][nt] = s_wb[(nt * KS + ks + 1) * 64];
                }
#pragma unroll
                for (int mt = 0; mt < MTA; mt++) {
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks & 1][nt], B[mt][ks], acc[mt][nt], 0, 0, 0);
                    if constexpr (PRE) if (nb == NBW - 1) B[mt][ks] = load_b(mt, ks);   // last pass over this tile's fragments: fetch the next tile's
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mt = 0; mt < MTA; mt++) {
                if (!mv_cur[mt]) continue;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    float vals[8];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        vals[i] = acc[mt][2 * u][i];
                        vals[4 + i] = acc[mt][2 * u + 1][i];
                    }
                    const half8 o = silu_pack8(vals[0], vals[1], vals[2], vals[3], vals[4], vals[5], vals[6], vals[7]);
                    *reinterpret_cast<half8 *>(out + m_cur[mt] * a.out_ld + ((nblk0 + nb) * 2 + u) * 32 + g * 8) = o;
                }
            }
        }
    };
    int ub = u0;
    for (; ub + MT <= u1; ub += MT) run_tile(std::integral_constant<int, MT>{}, std::true_type{}, ub);
    if (ub < u1) run_tile(std::integral_constant<int, 1>{}, std::false_type{}, ub);
}

// eligible: 1x1, SiLU, fp16 out, pair-packed, 64 | cout, K a multiple of 32 with 4..16 k-steps, first segment a multiple of 32
bool conv_pw_eligible(const ConvCfg &c, const ConvArgs &a)
{
    return c.ks == 1 && c.stride == 1 && c.act == 1 && !c.out_f32 && !c.cin16 && a.pair && a.cout_pad % 64 == 0 && a.Cin % 32 == 0 && a.s0.C % 32 == 0 &&
           (a.ksteps == 4 || a.ksteps == 6 || a.ksteps == 8 || a.ksteps == 12 || a.ksteps == 16) && a.res == nullptr && a.n2 == 0;
}

// LDS of the multi-block form: NBW weight slabs + NBW x 64 biases; 0 = this (KS, NBW) is not offered
size_t conv_pw_lds_bytes(const ConvArgs &a, int nbw)
{
    if (!(nbw == 2 || nbw == 4) || a.cout_pad % (64 * nbw) != 0) return 0;
    const int ks = a.ksteps;
    if (!(ks == 4 || ks == 6 || ks == 8 || ks == 12 || ks == 16) || nbw * ks > 32) return 0;   // <= 128 KiB of weights, <= 16 staging pieces per thread
    return (size_t)nbw * 4 * ks * 1024 + (size_t)nbw * 64 * 4;
}

bool launch_conv_pw(const ConvCfg &c, const ConvArgs &a, hipStream_t s)
{
    if (!conv_pw_eligible(c, a)) return false;
    const int tiles_total = (a.M + 31) / 32, nblocks = a.cout_pad / 64;
    if (c.ipw > 1) {   // several output-channel blocks per workgroup (c.ipw = NBW): the input is read nblocks / NBW times
        const int nbw = c.ipw;
        const size_t lds = conv_pw_lds_bytes(a, nbw);
        if (!lds) return false;
        const int groups = nblocks / nbw;
        const int units = (a.M + 15) / 16;
#define IRMV_PWN(KS_, NBW_)                                                                                          \
        if (a.ksteps == KS_ && nbw == NBW_) {                                                                          \
            static unsigned long long attr_done = 0;                                                                   \
            /* per DEVICE (engines on several GPUs are created from concurrent threads): workgroups a CU holds (registers   \
               and LDS) and the device's CU count size the grid to one round.  Element `dev` is written once, inside the    \
               critical section of once_per_device, before any launch on that device leaves it. */                        \
            static int per_cu[64], cus[64];                                                                            \
            int dev = 0;                                                                                               \
            (void)hipGetDevice(&dev);                                                                                  \
            dev &= 63;                                                                                                 \
            once_per_device(attr_done, [lds, dev] {                                                                    \
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_pwn_kernel<KS_, NBW_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                int nb = 0, nc = 0;                                                                                    \
                per_cu[dev] = 1; cus[dev] = 256;                                                                       \
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(conv1x1_pwn_kernel<KS_, NBW_>), 512, lds) == hipSuccess && nb >= 1) per_cu[dev] = nb > 2 ? 2 : nb; \
                if (hipDeviceGetAttribute(&nc, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && nc > 0) cus[dev] = nc; \
            });                                                                                                        \
            int wg = (cus[dev] * per_cu[dev] + groups - 1) / groups;     /* the chip in one round ... */               \
            if (wg > (units + 7) / 8) wg = (units + 7) / 8;              /* ... but a 16-pixel unit per wave at least (20 x 20 maps: more  \
                                                                            workgroups beat longer pipelines: 2 / 3 units per wave +7 / +25 %) */ \
            if (wg < 1) wg = 1;                                                                                        \
            hipLaunchKernelGGL((conv1x1_pwn_kernel<KS_, NBW_>), dim3(wg, groups), dim3(512), lds, s, a, wg);           \
            return true;                                                                                               \
        }
        IRMV_PWN(4, 2) IRMV_PWN(6, 2) IRMV_PWN(8, 2) IRMV_PWN(12, 2) IRMV_PWN(16, 2) IRMV_PWN(4, 4) IRMV_PWN(6, 4) IRMV_PWN(8, 4)
#undef IRMV_PWN
        return false;
    }
    // ~2 workgroups per CU in total, never more workgroups than there are 4-tile rounds
    int wg = (tiles_total + 3) / 4;
    const int cap = (512 + nblocks - 1) / nblocks;
    if (wg > cap) wg = cap;
    if (wg < 1) wg = 1;
#define IRMV_PW(KS_)                                                                                               \
    if (a.ksteps == KS_) {                                                                                          \
        static unsigned long long attr_done = 0;                                                                    \
        once_per_device(attr_done, [] {                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_pw_kernel<KS_>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
        });                                                                                                         \
        hipLaunchKernelGGL((conv1x1_pw_kernel<KS_>), dim3(wg, nblocks), dim3(256), (size_t)4 * KS_ * 1024, s, a, tiles_total, wg);            \
        return true;                                                                                                \
    }
    IRMV_PW(4) IRMV_PW(6) IRMV_PW(8) IRMV_PW(12) IRMV_PW(16)
#undef IRMV_PW
    return false;
}

// 16-byte patch pieces a thread stages per chunk (registers are reserved for all of them): a stride-1 2-D block is at
// most (16 MT + 2) x 18 pixels, the other schemes stage full-width rows or stride-2 patches.
constexpr int lds_pmax(int stride, int mt, bool tile2d, int wr = 0, int pf = 0)
{   // pf = 2 (four steps staged ahead, small maps only -- lds_geom checks the fit): four register sets have to fit
    return pf == 2 ? (stride == 1 ? 4 : 8) : (wr ? 4 : ((tile2d && stride == 1) ? (mt == 4 ? 8 : 4) : 12));
}

#ifndef IRMV_LDS_WAVES
#define IRMV_LDS_WAVES 2   // minimum waves per SIMD the register allocation aims at (A/B: scripts/gpu_stage.sh abwaves)
#endif
// The kernel's body as a device function of (workgroup index, grid size), so that one launch can serve several layers
// (conv3x3_lds_multi below); conv3x3_lds_kernel itself is the thin wrapper behind it.
template <int STRIDE, int MT, int NT, bool TILE2D, int N2, int PF = 0, int CM = 0, int NWV = 4, int WR = 0, bool PP = false>
__device__ __forceinline__ void conv3x3_lds_body(const ConvArgs &a, const half_t *wl, int tiles_x, int tiles_y, int twc_log2, int a_patch_bytes, int ipw, int batch,
                                                 int wg_x, int wg_y, int grid_x, int grid_y, int stagger = 0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    // PF: register staging of the (image, chunk) steps this many ahead of the MFMAs: 0 one step, 1 two, 2 four
    constexpr bool PF2 = PF != 0;
    constexpr int NPF = PF == 2 ? 4 : (PF == 1 ? 2 : 1);
    constexpr int NTH = 64 * NWV;   // NWV waves stacked along pixels (4; 8 for the stride-2 layers' large block, see launch_conv_lds)
    // PP (ping-pong, with WR): the workgroup's waves form TWO groups of NWP = NWV / 2 waves.  Each group owns a pixel tile
    // and a patch buffer of its own; they share the resident weights and run half a step apart -- while one group is in
    // its MFMA phase the other runs its epilogue, writes its next patch to LDS and issues the loads of the one after --
    // so that every SIMD always holds one wave that feeds the matrix pipe and one that feeds the vector / memory pipes.
    static_assert(!PP || (WR > 0 && NWV % 2 == 0), "ping-pong groups share resident weights");
    constexpr int NWP = PP ? NWV / 2 : NWV;   // waves that share one patch
    constexpr int NTP = 64 * NWP;
    const int sub = PP ? wave / NWP : 0, wv = wave - sub * NWP, tidp = tid - sub * NTP;
    const int g = lane >> 4, r = lane & 15;
    // Pixel ownership, two schemes:
    //  TILE2D  : a 2-D block.  An MFMA tile is (16 / TWc) rows x TWc columns, a wave stacks MT of them
    //            in y, the 4 waves stack in y again -> RH x TWc output pixels, halo (RH*s+2) x (TWc*s+2):
    //            ~1.4x halo overhead.  Used when the block shape tiles the image exactly.
    //  row run : 64*MT consecutive pixels in row-major order (tiles_x = runs per image): no masked
    //            lanes on 20x20 / 40x40 maps whose sides are not multiples of the block, at the price
    //            of staging full-width rows.
    // A workgroup keeps its tile position for `ipw` consecutive images: the staging plan below is computed once,
    // and the load -> LDS -> MFMA pipeline runs through all (image, chunk) steps without draining.
    const int bx = wg_x, by = wg_y;
    (void)grid_x; (void)grid_y;
    const int wg_tiles = PP ? (tiles_x * tiles_y + 1) / 2 : tiles_x * tiles_y;   // tile positions per workgroup column
    const int grp = bx / wg_tiles;
    int tile = PP ? 2 * (bx - grp * wg_tiles) + sub : bx - grp * wg_tiles;
    const bool active = tile < tiles_x * tiles_y;       // (PP, odd tile count: the last workgroup's second group only keeps the barriers' count)
    if (!active) tile = 0;
    const int img = grp * ipw, nimg = min(ipw, batch - img);
    const int nblk = by;
    const int HWo = a.Hout * a.Wout;
    int PW, PR, iy_base, ix_base;
    int poff[MT], mloc[MT];
    bool mv[MT];
    if constexpr (TILE2D) {
        const int TWc = 1 << twc_log2, trows = 16 >> twc_log2;
        const int RH = NWP * MT * trows;
        const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
        const int y0 = tyi * RH, x0 = txi * TWc;
        PW = TWc * STRIDE + 2; PR = RH * STRIDE + 2;
        iy_base = y0 * STRIDE - 1; ix_base = x0 * STRIDE - 1;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int ly = (wv * MT + mt) * trows + (r >> twc_log2), lx = r & (TWc - 1);
            const int oy = y0 + ly, ox = x0 + lx;
            mv[mt] = active && oy < a.Hout && ox < a.Wout;
            mloc[mt] = mv[mt] ? oy * a.Wout + ox : 0;
            poff[mt] = ((ly * STRIDE) * PW + lx * STRIDE) * pix_stride(STRIDE) + g * 16;
        }
    } else {
        constexpr int TPX = 16 * NWP * MT;
        const int m0 = tile * TPX, m1 = min(m0 + TPX, HWo);
        const int y0 = m0 / a.Wout, y1 = (m1 - 1) / a.Wout;
        PW = a.Win + 2; PR = (y1 - y0) * STRIDE + 3;
        iy_base = y0 * STRIDE - 1; ix_base = -1;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int m = m0 + (wv * MT + mt) * 16 + r;
            mv[mt] = active && m < m1;
            mloc[mt] = mv[mt] ? m : m0;
            const int oy = mloc[mt] / a.Wout, ox = mloc[mt] - oy * a.Wout;
            poff[mt] = (((oy - y0) * STRIDE) * PW + ox * STRIDE) * pix_stride(STRIDE) + g * 16;
        }
    }
    // WR > 0 (weights resident): the layer has exactly WR chunks; the weights of ALL of them are staged once per workgroup
    // and stay, the patch holds all WR chunk planes of an image, and a step is a whole image (one pair of barriers per
    // image instead of one per chunk, no weight traffic after the first step) -- see launch_conv_wres.
    constexpr int NPL = WR > 0 ? WR : 1;               // chunk planes of the patch / chunk slabs of the weights in LDS
    static_assert(WR == 0 || (!PF2 && CM == 0), "resident weights: image-major steps, one step of staging lead");
    constexpr int NPB = PP ? 2 * NPL : NPL;            // patch planes in LDS (one set per ping-pong group)
    unsigned char *s_patch = smem + (size_t)sub * NPL * a_patch_bytes;
    half8 *s_w = reinterpret_cast<half8 *>(smem + (size_t)NPB * a_patch_bytes);
    // epilogue constants, staged once: bias of this workgroup's 16 NT channels, then (N2 > 0) the fused 1x1's bias and its
    // A fragments.  Read from global memory inside the epilogue they cost a memory round trip per 16 x 32 output block:
    // a load after a store has to wait for the store (the compiler cannot prove bias and output apart), and every wait on
    // a load also waits for the staging loads in flight ahead of it.
    float *s_bias = reinterpret_cast<float *>(smem + (size_t)NPB * a_patch_bytes + (size_t)NPL * 9 * NT * 1024);
    float *s_bias2 = s_bias + 64;
    half8 *s_w2 = reinterpret_cast<half8 *>(s_bias + 128);
    if constexpr (NT % 2 == 0) {
        if (tid < NT * 16) s_bias[tid] = a.bias[by * NT * 16 + tid];
    } else {   // one tile: s_bias[4 g + i] = bias of the channel lane group g holds in register i
        const int t = by;
        if (tid < 16) s_bias[tid] = a.bias[a.pair ? ((t >> 1) * 32 + (tid >> 2) * 8 + (t & 1) * 4 + (tid & 3)) : (t * 16 + tid)];
    }
    if constexpr (N2 > 0) {
        if (tid < N2 * 16) s_bias2[tid] = a.bias2[tid];
        for (int e = tid; e < N2 * 2 * 64; e += NTH) s_w2[e] = reinterpret_cast<const half8 *>(a.w2)[e];
    }
    const int chunks = a.Cin >> 5;
    const half8 zero8 = (half8){0, 0, 0, 0, 0, 0, 0, 0};

    // staging plan: element e -> (patch pixel, 16-byte quarter); weights: 9*NT*64 half8 per chunk
    constexpr int PMAX = lds_pmax(STRIDE, MT, TILE2D, WR, PF);   // patch 16-B pieces per thread and chunk plane (host guarantees the fit)
    constexpr int WPT = (9 * NT * 64 + NTH - 1) / NTH; // weight half8 per thread
    const int n_pe = PR * PW * 4;
    const float inv_pw = 1.0f / (float)PW;
    const half_t *src_p[PMAX];
    int dst_p[PMAX];
    bool val_p[PMAX], use_p[PMAX];
#pragma unroll
    for (int i = 0; i < PMAX; i++) {
        const int e = tidp + i * NTP;
        use_p[i] = e < n_pe;
        val_p[i] = false;
        src_p[i] = a.s0.p;
        dst_p[i] = 0;
        if (i * NTP >= n_pe) continue;                     // wave-uniform: this piece slot is unused by the whole workgroup
        const int pix = use_p[i] ? (e >> 2) : 0, q = e & 3;
        int pr = (int)((float)pix * inv_pw);               // pix < 2^16: one correction step makes the quotient exact
        pr -= (pr * PW > pix) ? 1 : 0;
        pr += ((pr + 1) * PW <= pix) ? 1 : 0;
        const int pc = pix - pr * PW;
        const int iy = iy_base + pr, ix = ix_base + pc;
        val_p[i] = active && use_p[i] && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
        src_p[i] = a.s0.p + ((size_t)(img * a.Hin + (val_p[i] ? iy : 0)) * a.Win + (val_p[i] ? ix : 0)) * a.s0.ld + q * 8;
        dst_p[i] = pix * pix_stride(STRIDE) + q * 16;
    }
    const half8 *wsrc = reinterpret_cast<const half8 *>(wl) + (size_t)nblk * chunks * (9 * NT * 64);

    // fused trailing 1x1 (N2 > 0): its A fragments are read from LDS where they are used, in the epilogue -- held across
    // the main loop they cost 8 VGPRs per 16 output channels and pushed the N2 = 4 tiles off the large register tiles
    const half8 *w2 = s_w2 + lane;

    // register staging of the (image, chunk) steps ahead of the MFMAs: one step ahead, or two (PF = 1: small-M layers whose
    // step -- 9 * MT * NT MFMAs -- is shorter than a memory round trip, so a single step of lead exposes the latency), or
    // four (PF = 2, round 4: a LONE frame's 128-channel layers on the 20 x 20 maps are 14 .. 56 workgroups of four steps
    // each -- with two steps in flight the workgroup, and with it the layer, sat out three memory round trips: 7 - 8 us
    // where the 64-channel layers (two steps: one round trip) take 4.7)
    half8 rp[NPF][PMAX], rw[NPF][WPT];
    const size_t img_stride = (size_t)a.Hin * a.Win * a.s0.ld;
    int l_im = 0, l_chunk = 0;   // loader position
    int c_im = 0, c_chunk = 0;   // consumer position
    auto issue_loads = [&](half8 (&p)[PMAX], half8 (&w)[WPT]) {
        const size_t off = (size_t)l_im * img_stride + (size_t)l_chunk * 32;
#pragma unroll
        for (int i = 0; i < PMAX; i++) {
            // (conditional loads cost nothing here: the next thing this wave does with the ring is write ALL of it to LDS, so
            // the conservative wait counts they cause -- see conv_mfma_kernel -- wait for nothing that is not needed; made
            // unconditional, the unused piece slots of the small tiles were extra loads: measured 10 - 15 % slower at MT = 1)
            p[i] = zero8;
            if (val_p[i]) p[i] = *reinterpret_cast<const half8 *>(src_p[i] + off);
        }
#pragma unroll
        for (int i = 0; i < WPT; i++) {
            if (CM > 0 && l_im != 0) break;                // chunk-major: the chunk's weights are in LDS already
            const int e = tid + i * NTH;
            if (e < 9 * NT * 64) w[i] = wsrc[(size_t)l_chunk * (9 * NT * 64) + e];
        }
        if constexpr (CM > 0) {
            if (++l_im == nimg) { l_im = 0; l_chunk++; }
        } else {
            if (++l_chunk == chunks) { l_chunk = 0; l_im++; }
        }
    };
    auto write_lds = [&](const half8 (&p)[PMAX], const half8 (&w)[WPT]) {
#pragma unroll
        for (int i = 0; i < PMAX; i++) {
            if (use_p[i]) *reinterpret_cast<half8 *>(s_patch + dst_p[i]) = p[i];
        }
#pragma unroll
        for (int i = 0; i < WPT; i++) {
            if (CM > 0 && c_im != 0) break;
            const int e = tid + i * NTH;
            if (e < 9 * NT * 64) s_w[e] = w[i];
        }
    };

    // CM > 0 (chunk-major): the workgroup's CM images each keep their own accumulators, so that a chunk's weights are
    // staged ONCE and every image's patch of that chunk runs against them (image-major order re-stages the 9 NT KiB of
    // weights per (image, chunk) step: for the small tiles that is more LDS traffic than the patch itself)
    constexpr int NA = CM > 0 ? CM : 1;
    // Accumulators start AT the bias (log2 e-scaled, irmv_common.hpp): fetched from memory here -- s_bias is not visible
    // before the first barrier --, and put back from the epilogue's own copy at the end of every image (store_tile), where
    // the zero fill used to be: the bias add of the epilogue is gone at no cost.
    f32x4 acc[NA][MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int ch = (NT % 2 == 0) ? (by * NT * 16 + (nt >> 1) * 32 + g * 8 + (nt & 1) * 4) : (a.pair ? ((by >> 1) * 32 + g * 8 + (by & 1) * 4) : (by * 16 + g * 4));
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(a.bias + ch);
#pragma unroll
        for (int ai = 0; ai < NA; ai++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[ai][mt][nt] = b0;
    }

    // epilogue of one image (bias, SiLU, shortcut, fp16, NHWC store); clears the accumulators for the next one
    const int nt0 = nblk * NT;
    auto store_tile = [&](int im, auto ai_) {
        auto &accx = acc[decltype(ai_)::value];
        // A fragments of the fused 1x1: loaded by EVERY lane (an MFMA reads its A rows from all 64 lanes, whatever the
        // pixel mask of the B side says), live only for the epilogue
        half8 W2[N2 > 0 ? N2 : 1][2];
        if constexpr (N2 > 0) {
#pragma unroll
            for (int t2 = 0; t2 < N2; t2++) { W2[t2][0] = w2[(t2 * 2 + 0) * 64]; W2[t2][1] = w2[(t2 * 2 + 1) * 64]; }
        }
        // bias: LDS -> registers once per image; shortcut: every block's load issued before the first store
        float bs[NT % 2 == 0 ? NT / 2 : 1][8];
        if constexpr (NT % 2 == 0) {
#pragma unroll
            for (int u = 0; u < NT / 2; u++) {
                const f32x4 b0 = *reinterpret_cast<const f32x4 *>(s_bias + u * 32 + g * 8), b1 = *reinterpret_cast<const f32x4 *>(s_bias + u * 32 + g * 8 + 4);
#pragma unroll
                for (int i = 0; i < 4; i++) { bs[u][i] = b0[i]; bs[u][4 + i] = b1[i]; }
            }
        } else {
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(s_bias + g * 4);
#pragma unroll
            for (int i = 0; i < 4; i++) bs[0][i] = b0[i];
        }
        half8 rv8[NT % 2 == 0 ? MT : 1][NT % 2 == 0 ? NT / 2 : 1];
        half4 rv4[MT];
        if (a.res) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const size_t m = (size_t)im * HWo + mloc[mt];
                if constexpr (NT % 2 == 0) {
#pragma unroll
                    for (int u = 0; u < NT / 2; u++) rv8[mt][u] = *reinterpret_cast<const half8 *>(a.res + m * a.res_ld + (nt0 / 2 + u) * 32 + g * 8);
                } else {
                    const int c0 = a.pair ? ((nt0 >> 1) * 32 + g * 8 + (nt0 & 1) * 4) : (nt0 * 16 + g * 4);
                    rv4[mt] = *reinterpret_cast<const half4 *>(a.res + m * a.res_ld + c0);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            // N2 > 0: the fused 1x1's MFMAs must run with every lane (their A rows live in all 64 lanes; inside a divergent
            // region the compiler is also free to sink the A-fragment loads behind the mask): masked pixels run the
            // epilogue arithmetic on whatever their accumulators hold and only their STORES are skipped.
            if (N2 > 0 || mv[mt]) {
                const size_t m = (size_t)im * HWo + mloc[mt];
                if constexpr (NT % 2 == 0) {   // pair-packed (host guarantees): 8 contiguous channels per lane
                    half8 ov[NT / 2];
#pragma unroll
                    for (int u = 0; u < NT / 2; u++) {
                        const int c0 = (nt0 / 2 + u) * 32 + g * 8;
                        float vals[8];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            vals[i] = accx[mt][2 * u][i];
                            vals[4 + i] = accx[mt][2 * u + 1][i];
                        }
                        if (a.res) {   // (rounding pinned: irmv_common.hpp)
#pragma unroll
                            for (int i = 0; i < 8; i++) ov[u][i] = silu_add_res(vals[i], (float)rv8[mt][u][i]);
                        } else {
                            ov[u] = silu_pack8(vals[0], vals[1], vals[2], vals[3], vals[4], vals[5], vals[6], vals[7]);
                        }
                        if constexpr (N2 == 0) *reinterpret_cast<half8 *>(static_cast<half_t *>(a.out) + m * a.out_ld + c0) = ov[u];
                    }
                    if constexpr (N2 > 0) {
                        mfma_operand_fence(ov[0], ov[1]);   // (irmv_common.hpp: the MFMAs below read what inline asm has just written)
                        // With the paired-tile packing lane (g, r) now holds channels u*32 + 8g + [0, 8) of pixel r: exactly the
                        // B fragment of k-step u of a 1x1 conv over these 64 channels.  Same operands, same k order as the
                        // stand-alone 1x1 kernel reading this tensor back from memory.
#pragma unroll
                        for (int t2 = 0; t2 < N2; t2++) {
                            f32x4 c2 = (f32x4){0.f, 0.f, 0.f, 0.f};
                            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(W2[t2][0], ov[0], c2, 0, 0, 0);
                            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(W2[t2][1], ov[1], c2, 0, 0, 0);
                            const int co = t2 * 16 + g * 4;
                            const f32x4 b2 = *reinterpret_cast<const f32x4 *>(s_bias2 + co);
                            const f32x4 v2 = (f32x4){c2[0] * kActUnscale + b2[0], c2[1] * kActUnscale + b2[1], c2[2] * kActUnscale + b2[2], c2[3] * kActUnscale + b2[3]};   // (the 1x1's inputs carry the activation scale)
                            if (mv[mt]) *reinterpret_cast<f32x4 *>(a.out2 + m * a.out2_ld + co) = v2;
                            if (a.scan_keys) {
                                // class logits: lane (g, r) holds classes co .. co + 3 of its pixel -- the values the head record just
                                // received.  Candidates (rare) go to the frame's key list here, so no kernel re-reads the head for them
                                // (key format and threshold test of k_post.hip: orderable(logit) << 32 | ~(anchor * nc + class)).
                                bool hit = false;
#pragma unroll
                                for (int i = 0; i < 4; i++) hit = hit || (mv[mt] && co + i < a.scan_nc && v2[i] > a.scan_thr);
                                if (hit) {
                                    const int an = a.scan_abase + mloc[mt];
#pragma unroll
                                    for (int i = 0; i < 4; i++) {
                                        if (co + i < a.scan_nc && v2[i] > a.scan_thr) {
                                            const int idx = atomicAdd(&a.scan_counts[im], 1);
                                            const unsigned int u = __float_as_uint(v2[i]);
                                            const unsigned int ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                                            if (idx >= 0 && idx < a.scan_key_cap)
                                                a.scan_keys[(size_t)im * a.scan_key_cap + idx] =
                                                    ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - (unsigned int)(an * a.scan_nc + co + i));
                                        }
                                    }
                                }
                            }
                        }
                    }
                } else {
                    const int t = nt0;
                    const int c0 = a.pair ? ((t >> 1) * 32 + g * 8 + (t & 1) * 4) : (t * 16 + g * 4);
                    half4 o;
                    if (a.res) {
#pragma unroll
                        for (int i = 0; i < 4; i++) o[i] = silu_add_res(accx[mt][0][i], (float)rv4[mt][i]);
                    } else {
                        o = silu_pack4(accx[mt][0][0], accx[mt][0][1], accx[mt][0][2], accx[mt][0][3]);
                    }
                    *reinterpret_cast<half4 *>(static_cast<half_t *>(a.out) + m * a.out_ld + c0) = o;
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; nt++)   // the next image starts at the bias again
                accx[mt][nt] = (NT % 2 == 0) ? (f32x4){bs[nt >> 1][(nt & 1) * 4 + 0], bs[nt >> 1][(nt & 1) * 4 + 1], bs[nt >> 1][(nt & 1) * 4 + 2], bs[nt >> 1][(nt & 1) * 4 + 3]}
                                             : (f32x4){bs[0][0], bs[0][1], bs[0][2], bs[0][3]};
        }
    };

    using I0 = std::integral_constant<int, 0>;
    const int steps = nimg * chunks;
    // The staging loads of the NEXT step ride inside this step's tap loop, a few per tap (SPREAD): issued in one burst
    // after the barrier they fill the CU's vector-memory queue (17 wave-instructions of 1 KiB against 64 B/clk) and the
    // wave sits in the issue stall instead of starting its MFMAs.  The last step re-loads its own pieces (no branch).
    constexpr int NPIECE = PMAX + WPT, PER_TAP = (NPIECE + 8) / 9;
    auto taps = [&](auto spread, auto ai_, int plane = 0) {
        constexpr bool SPREAD = decltype(spread)::value;
        auto &accx = acc[decltype(ai_)::value];
        const unsigned char *s_pl = s_patch + (size_t)plane * a_patch_bytes;   // (WR: chunk plane of the patch, chunk slab of the weights)
        const half8 *s_wc = s_w + plane * (9 * NT * 64);
        size_t off = 0;
        if constexpr (SPREAD) {
            if (l_im == nimg) { l_im = nimg - 1; l_chunk = chunks - 1; }
            off = (size_t)l_im * img_stride + (size_t)l_chunk * 32;
        }
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            if constexpr (SPREAD) {
#pragma unroll
                for (int k = 0; k < PER_TAP; k++) {
                    const int j = tap * PER_TAP + k;
                    if (j < PMAX) {
                        rp[0][j] = zero8;
                        if (val_p[j]) rp[0][j] = *reinterpret_cast<const half8 *>(src_p[j] + off);
                    } else if (j < NPIECE) {
                        const int e = tid + (j - PMAX) * NTH;
                        if (e < 9 * NT * 64) rw[0][j - PMAX] = wsrc[(size_t)l_chunk * (9 * NT * 64) + e];
                    }
                }
            }
            const int kh = tap / 3, kw = tap - kh * 3;
            const int toff = (kh * PW + kw) * pix_stride(STRIDE);
            half8 A[NT], B[MT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) A[nt] = s_wc[(tap * NT + nt) * 64 + lane];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) B[mt] = *reinterpret_cast<const half8 *>(s_pl + poff[mt] + toff);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    accx[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[nt], B[mt], accx[mt][nt], 0, 0, 0);
            if constexpr (SPREAD) __builtin_amdgcn_sched_barrier(0x38f);   // everything but vector-memory instructions may cross
        }
        if constexpr (SPREAD) {
            if (++l_chunk == chunks) { l_chunk = 0; l_im++; }
        }
    };
    // WR: all WR * 9 k-steps of an image as ONE software-pipelined sequence -- the fragments of k-step s + 1 are requested
    // from LDS before the MFMAs of k-step s are issued (the scheduling barrier pins that order: left to itself the
    // compiler reads a fragment right in front of its first use and the wave -- in ping-pong mode the only one on its SIMD
    // that feeds the matrix pipe -- sits out the LDS latency 18 times per image).
    auto ksteps_wr = [&]() {
        constexpr int KS = (WR > 0 ? WR : 1) * 9;
        auto &accx = acc[0];
        half8 A[2][NT], B[2][MT];
        auto frag = [&](auto ks_, half8 (&Af)[NT], half8 (&Bf)[MT]) {
            constexpr int ks = decltype(ks_)::value, c = ks / 9, tap = ks % 9, kh = tap / 3, kw = tap % 3;
            const unsigned char *s_pl = s_patch + (size_t)c * a_patch_bytes + (kh * PW + kw) * pix_stride(STRIDE);
            const half8 *s_wc = s_w + (c * 9 + tap) * (NT * 64) + lane;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) Af[nt] = s_wc[nt * 64];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) Bf[mt] = *reinterpret_cast<const half8 *>(s_pl + poff[mt]);
        };
        frag(std::integral_constant<int, 0>{}, A[0], B[0]);
        auto step = [&](auto ks_) {
            constexpr int ks = decltype(ks_)::value;
            if constexpr (ks + 1 < KS) frag(std::integral_constant<int, ks + 1>{}, A[(ks + 1) & 1], B[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    accx[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks & 1][nt], B[ks & 1][mt], accx[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto run = [&](auto self, auto ks_) -> void {
            constexpr int ks = decltype(ks_)::value;
            if constexpr (ks < KS) { step(ks_); self(self, std::integral_constant<int, ks + 1>{}); }
        };
        run(run, std::integral_constant<int, 0>{});
    };
    auto mma_step = [&]() {
        taps(std::false_type{}, I0{});
        __syncthreads();
        if (++c_chunk == chunks) {
            store_tile(img + c_im, I0{});
            c_chunk = 0;
            c_im++;
        }
    };
    if constexpr (WR > 0) {
        // ---- weights resident: stage all WR chunk slabs once, then one step per image ----
        {
            constexpr int WALL = WR * 9 * NT * 64, WPA = (WALL + NTH - 1) / NTH;   // half8 pieces, per thread
            half8 wr_[WPA];
#pragma unroll
            for (int i = 0; i < WPA; i++) {
                const int e = tid + i * NTH;
                wr_[i] = wsrc[e < WALL ? e : WALL - 1];
            }
#pragma unroll
            for (int i = 0; i < WPA; i++) {
                const int e = tid + i * NTH;
                if (e < WALL) s_w[e] = wr_[i];
            }
        }
        half8 rq[WR][PMAX];
        auto issue_wr = [&](int im) {
            const size_t off = (size_t)im * img_stride;
#pragma unroll
            for (int c = 0; c < WR; c++)
#pragma unroll
                for (int i = 0; i < PMAX; i++) {
                    rq[c][i] = zero8;
                    if (val_p[i]) rq[c][i] = *reinterpret_cast<const half8 *>(src_p[i] + off + c * 32);
                }
        };
        auto write_wr = [&]() {
#pragma unroll
            for (int c = 0; c < WR; c++)
#pragma unroll
                for (int i = 0; i < PMAX; i++)
                    if (use_p[i]) *reinterpret_cast<half8 *>(s_patch + (size_t)c * a_patch_bytes + dst_p[i]) = rq[c][i];
        };
        if constexpr (PP) {
            // group 1 runs one barrier (= half a step) behind group 0; both execute 2 nimg + 2 barriers
            issue_wr(0);
            if (sub == 1) {
                // group 0 passes this barrier through its __syncthreads below and then reads weight / bias / fused-1x1
                // pieces that THIS group's threads wrote to LDS above: their ds_writes must have landed before the
                // barrier releases (a bare s_barrier carries no lgkmcnt wait on gfx950)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_barrier();
            }
            write_wr();
            if (nimg > 1) issue_wr(1);
            __syncthreads();
            for (int s = 0; s < nimg; s++) {
                ksteps_wr();                                                          // MFMA phase (the other group: load phase)
                __syncthreads();
                store_tile(img + s, I0{});                                            // load phase (the other group: MFMA phase)
                if (s + 1 < nimg) {
                    write_wr();
                    if (s + 2 < nimg) issue_wr(s + 2);
                }
                __syncthreads();
            }
            if (sub == 0) __builtin_amdgcn_s_barrier();
            return;
        }
        // Lockstep form (maps that do not tile into the ping-pong blocks).  Left alone, the two waves of a SIMD reach their
        // MFMA phase, their fragment reads and their SiLU epilogue together: the matrix pipe idles through both epilogues.
        // STAGGER (round 5; MI355X_MICROARCH.md "Two waves per SIMD", item 9): the second-dispatched half of the workgroup
        // (waves NWV / 2 ..) defers the epilogue of image s to the start of step s + 1, behind the barrier that releases
        // patch s + 1 -- its vector work then runs beside the first half's MFMAs, and the first half's epilogue beside the
        // tail of its MFMAs.  The sums stay in their accumulator registers across the barriers; same operations on the
        // same operands, only later: bit-identical.  Barriers per image unchanged (two).
        const bool late = stagger && wave >= NWV / 2;
        issue_wr(0);
        for (int s = 0; s < nimg; s++) {
            write_wr();
            __syncthreads();
            if (late && s > 0) store_tile(img + s - 1, I0{});
            if (s + 1 < nimg) issue_wr(s + 1);
            ksteps_wr();
            __syncthreads();
            if (!late) store_tile(img + s, I0{});
        }
        if (late) store_tile(img + nimg - 1, I0{});
    } else if constexpr (CM > 0) {
        static_assert(!PF2 && CM <= 4, "chunk-major order: one step of staging lead, at most four images");
        auto each_image = [&](auto f) {
            f(std::integral_constant<int, 0>{});
            if constexpr (CM > 1) f(std::integral_constant<int, 1>{});
            if constexpr (CM > 2) f(std::integral_constant<int, 2>{});
            if constexpr (CM > 3) f(std::integral_constant<int, 3>{});
        };
        issue_loads(rp[0], rw[0]);
        for (int c = 0; c < chunks; c++) {
            each_image([&](auto ai) {
                constexpr int AI = decltype(ai)::value;
                if (AI < nimg) {                                 // workgroup-uniform (the batch's last group may be short)
                    write_lds(rp[0], rw[0]);                     // (c_im == AI: the weights go with image 0)
                    __syncthreads();
                    if (!(c == chunks - 1 && AI == nimg - 1)) issue_loads(rp[0], rw[0]);
                    taps(std::false_type{}, ai);
                    __syncthreads();
                    if (++c_im == nimg) { c_im = 0; c_chunk++; }
                }
            });
        }
        each_image([&](auto ai) {
            if (decltype(ai)::value < nimg) store_tile(img + decltype(ai)::value, ai);
        });
    } else if constexpr (!PF2) {
        issue_loads(rp[0], rw[0]);
        for (int s = 0; s < steps; s++) {
            write_lds(rp[0], rw[0]);
            __syncthreads();
            if constexpr (MT == 4) {
                taps(std::true_type{}, I0{});                          // next step's loads spread over the taps
            } else {   // small tiles: the burst is short and the spread costs more than it saves (measured)
                if (s + 1 < steps) issue_loads(rp[0], rw[0]);
                taps(std::false_type{}, I0{});
            }
            __syncthreads();
            if (++c_chunk == chunks) {
                store_tile(img + c_im, I0{});
                c_chunk = 0;
                c_im++;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NPF; k++)
            if (k < steps) issue_loads(rp[k], rw[k]);
        for (int s = 0; s < steps; s += NPF) {
#pragma unroll
            for (int k = 0; k < NPF; k++) {
                if (s + k < steps) {
                    write_lds(rp[k], rw[k]);
                    __syncthreads();
                    if (s + k + NPF < steps) issue_loads(rp[k], rw[k]);   // NPF steps ahead
                    mma_step();
                }
            }
        }
    }
}

template <int STRIDE, int MT, int NT, bool TILE2D, int N2, int PF = 0, int CM = 0, int NWV = 4, int WR = 0, bool PP = false>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(IRMV_LDS_WAVES))) void conv3x3_lds_kernel(ConvArgs a, const half_t *wl, int tiles_x, int tiles_y, int twc_log2, int a_patch_bytes, int ipw, int batch, int nblocks, int xcd)
{
    // 1-D grid, dealt so that every workgroup of an image group -- all its tiles, all its output-channel blocks -- runs on ONE
    // XCD (irmv_common.hpp tile_image): tiles share halo pixels and channel blocks share the whole input, and with the plain
    // (tile, group) x block grid those re-reads came from memory, not from the XCD's L2 (1.3 - 1.9 x the input bytes fetched
    // by the stride-2 and 80 x 80 layers, profiles/r04_traffic.json)
    const int wg_tiles = PP ? (tiles_x * tiles_y + 1) / 2 : tiles_x * tiles_y, groups = (batch + ipw - 1) / ipw;
    int rem, grp;
    tile_image(blockIdx.x, wg_tiles * nblocks, groups, xcd & 1, rem, grp);   // xcd: bit 0 = image order, bit 1 = stagger (lockstep resident-weight form)
    const int nblk = rem / wg_tiles, tile = rem - nblk * wg_tiles;
    conv3x3_lds_body<STRIDE, MT, NT, TILE2D, N2, PF, CM, NWV, WR, PP>(a, wl, tiles_x, tiles_y, twc_log2, a_patch_bytes, ipw, batch, grp * wg_tiles + tile, nblk, wg_tiles * groups, nblocks, (xcd >> 1) & 1);
}

// Several independent 3x3 layers in ONE launch (the Detect branches of the three levels in a single-frame step: fifteen
// launches of 4 - 8 us, most of it launch and ramp, become four).  Every member runs the body above with its own
// arguments and geometry on its own range of workgroups; tile shape (MT, NT) is the group's, the block scheme (2-D block
// or row run) and the fused 1x1 (none / class branch / box branch) are the member's.
template <int STRIDE, int MT, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(IRMV_LDS_WAVES))) void conv3x3_lds_multi(LdsMultiArgs m)
{
    int id = blockIdx.x, k = 0;
    while (k + 1 < m.n && id >= m.start[k + 1]) k++;
    id -= m.start[k];
    const LdsMember &l = m.m[k];
    const int wx = id % l.gx, wy = id / l.gx;
#define IRMV_BODY(T2D_, N2_) conv3x3_lds_body<STRIDE, MT, NT, T2D_, N2_>(l.a, l.wl, l.tiles_x, l.tiles_y, l.twc_log2, l.patch_bytes, 1, l.batch, wx, wy, l.gx, l.gy)
    if constexpr (NT == 4) {
        if (l.a.n2 == 4) { if (l.tile2d) IRMV_BODY(true, 4); else IRMV_BODY(false, 4); return; }
        if (l.a.n2 == 1) { if (l.tile2d) IRMV_BODY(true, 1); else IRMV_BODY(false, 1); return; }
    }
    if (l.tile2d) IRMV_BODY(true, 0); else IRMV_BODY(false, 0);
#undef IRMV_BODY
}

// Geometry of the LDS kernel for one layer and tile shape.  The 2-D block scheme is used when the
// block tiles the image exactly (no masked lanes); otherwise the row-run scheme.  bytes == 0: not eligible.
struct LdsGeom { bool tile2d; int tiles_x, tiles_y, twc_log2, patch_bytes; size_t bytes; };

static LdsGeom lds_geom(const ConvArgs &a, int stride, int mt, int nt, int nwv = 4, int wr = 0, bool pp = false, int pf = 0)
{   // nwv: the waves that share one patch (ping-pong: half the workgroup's)
    LdsGeom g{false, 0, 0, 0, 0, 0};
    if (a.Cin % 32 != 0 || a.s1.C != 0 || a.s0.shift != 0) return g;
    if (!((nt == 1) || ((nt == 2 || nt == 4) && a.pair) || (nt == 8 && a.pair && nwv == 8 && stride == 2 && mt == 1))) return g;   // nt = 8: the stride-2 layers' 8-wave workgroup only
    if (a.cout_pad % (16 * nt) != 0) return g;
    int pr, pw;
    const int l2 = a.Wout % 16 == 0 ? 4 : (a.Wout % 8 == 0 ? 3 : (a.Wout % 4 == 0 ? 2 : -1));
    const int rh = l2 >= 0 ? nwv * mt * (16 >> l2) : 0;
    if (l2 >= 0 && a.Hout % rh == 0) {
        g.tile2d = true;
        g.twc_log2 = l2;
        g.tiles_x = a.Wout >> l2;
        g.tiles_y = a.Hout / rh;
        pr = rh * stride + 2;
        pw = (1 << l2) * stride + 2;
    } else {
        const int tpx = 16 * nwv * mt;
        g.tiles_x = (a.Hout * a.Wout + tpx - 1) / tpx;
        g.tiles_y = 1;
        const int rows = (tpx + a.Wout - 2) / a.Wout + 1;      // most output rows a run of tpx pixels can touch
        pr = (rows - 1) * stride + 3;
        pw = a.Win + 2;
    }
    if ((size_t)pr * pw * 4 > (size_t)lds_pmax(stride, mt, g.tile2d, wr, pf) * 64 * nwv) return g;   // staging plan: pieces per thread
    g.patch_bytes = pr * pw * pix_stride(stride);
    const int planes = wr ? wr : 1;                     // resident weights: every chunk's patch plane and weight slab at once
    const size_t bytes = (size_t)planes * (pp ? 2 : 1) * g.patch_bytes + (size_t)planes * 9 * nt * 1024 + 512 + (size_t)a.n2 * 2048;   // + bias, bias2, fused 1x1 fragments
    if (bytes > (size_t)(wr ? 160 : (nwv == 8 ? 150 : 80)) * 1024) return g;   // two or more 4-wave workgroups per CU, or one of 8 waves
    g.bytes = bytes;
    return g;
}

size_t conv_lds_bytes(const ConvArgs &a, int stride, int mt, int nt, int *patch_rows_max, bool w8)
{
    const LdsGeom g = lds_geom(a, stride, mt, nt, w8 ? 8 : 4);
    if (patch_rows_max) *patch_rows_max = g.patch_bytes;
    return g.bytes;
}

template <int STRIDE, int MT, int NT, bool TILE2D, int N2, int PF = 0, int CM = 0, int NWV = 4, int WR = 0, bool PP = false>
static void launch_lds_inst(const ConvArgs &a, const half_t *wl, int batch, int ipw, const LdsGeom &g, hipStream_t s)
{
    static unsigned long long attr_done = 0;   // per instantiation: devices whose dynamic-LDS limit has been raised
    once_per_device(attr_done, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_lds_kernel<STRIDE, MT, NT, TILE2D, N2, PF, CM, NWV, WR, PP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int groups = (batch + ipw - 1) / ipw;
    const int wg_tiles = PP ? (g.tiles_x * g.tiles_y + 1) / 2 : g.tiles_x * g.tiles_y;   // ping-pong: two tile positions per workgroup
    const int nblocks = a.cout_pad / (16 * NT);
    hipLaunchKernelGGL((conv3x3_lds_kernel<STRIDE, MT, NT, TILE2D, N2, PF, CM, NWV, WR, PP>), dim3(wg_tiles * groups * nblocks), dim3(64 * NWV), g.bytes, s, a,
                       wl, g.tiles_x, g.tiles_y, g.twc_log2, g.patch_bytes, ipw, batch, nblocks, xcd_image_order() | (wres_stagger() << 1));
}

// Weights-resident variant: Cin = 64 -> 64 channels, stride 1.  ONE 8-wave workgroup per CU keeps the
// layer's whole 72 KiB of weights in LDS and walks `ipw` images at its tile position (any ipw >= 1: the grid is sized so
// that the chip holds it in one round); per image it stages the 64-channel patch once and runs all 18 k-steps between
// one pair of barriers.  Same K order as the chunked kernel (chunk, tap) -> bit-identical.
static LdsGeom wres_geom(const ConvArgs &a, int stride, bool pp)
{
    if (stride != 1 || a.Cin != 64 || a.cout_pad != 64 || !a.pair || (a.n2 > 0 && a.res) || !(a.n2 == 0 || a.n2 == 1 || a.n2 == 4)) return LdsGeom{false, 0, 0, 0, 0, 0};
    LdsGeom g = pp ? lds_geom(a, 1, 2, 4, 4, 2, true) : lds_geom(a, 1, 2, 4, 8, 2);
    if (pp && !g.tile2d) g.bytes = 0;   // ping-pong groups: 2-D blocks only
    return g;
}
size_t conv_wres_bytes(const ConvArgs &a, int stride, bool pp) { return wres_geom(a, stride, pp).bytes; }
int conv_wres_tiles(const ConvArgs &a, int stride, bool pp)
{
    const LdsGeom g = wres_geom(a, stride, pp);
    if (!g.bytes) return 0;
    return pp ? (g.tiles_x * g.tiles_y + 1) / 2 : g.tiles_x * g.tiles_y;
}

bool launch_conv_wres(int ipw, const ConvArgs &a, const half_t *wl, int batch, hipStream_t s, bool pp)
{
    if (ipw < 1) ipw = 1;
    if (pp) {   // ping-pong: two groups of four waves, each on its own (4 mt rows x 16)-pixel block
        const LdsGeom gp = wres_geom(a, 1, true);
        if (!gp.bytes) return false;
#define IRMV_WRES_PP(N2_)                                                                                   \
        if (a.n2 == N2_) { launch_lds_inst<1, 2, 4, true, N2_, false, 0, 8, 2, true>(a, wl, batch, ipw, gp, s); return true; }
        IRMV_WRES_PP(0) IRMV_WRES_PP(1) IRMV_WRES_PP(4)
#undef IRMV_WRES_PP
        return false;
    }
    const LdsGeom g = wres_geom(a, 1, false);
    if (!g.bytes) return false;
#define IRMV_WRES(N2_)                                                                                      \
    if (a.n2 == N2_) {                                                                                      \
        if (g.tile2d) launch_lds_inst<1, 2, 4, true, N2_, false, 0, 8, 2>(a, wl, batch, ipw, g, s);          \
        else launch_lds_inst<1, 2, 4, false, N2_, false, 0, 8, 2>(a, wl, batch, ipw, g, s);                  \
        return true;                                                                                        \
    }
    IRMV_WRES(0) IRMV_WRES(1) IRMV_WRES(4)
#undef IRMV_WRES
    return false;
}

bool launch_conv_lds(int stride, int mt, int nt, int ipw, const ConvArgs &a, const half_t *wl, int batch, hipStream_t s, int pf2, int cm, bool w8)
{
    if (ipw < 1) ipw = 1;
    if (w8) {   // stride-2 layers: ONE 8-wave workgroup per CU on a block of 128 mt pixels -- see DESIGN section 4
        if (stride != 2 || !(nt == 4 || nt == 8) || a.n2 > 0 || pf2 || (cm && cm != ipw)) return false;
        const LdsGeom g8 = lds_geom(a, stride, mt, nt, 8);
        if (!g8.bytes) return false;
        if (nt == 8) {   // 128 output channels per workgroup: the patch is staged once per 128 channels (mt = 1: 16 pixels x 128 channels per wave)
#define IRMV_LDS_W8N8(CM_)                                                                                  \
            if (mt == 1 && cm == CM_) {                                                                     \
                if (g8.tile2d) launch_lds_inst<2, 1, 8, true, 0, false, CM_, 8>(a, wl, batch, ipw, g8, s);  \
                else launch_lds_inst<2, 1, 8, false, 0, false, CM_, 8>(a, wl, batch, ipw, g8, s);           \
                return true;                                                                                \
            }
            IRMV_LDS_W8N8(0) IRMV_LDS_W8N8(2)
#undef IRMV_LDS_W8N8
            return false;
        }
#define IRMV_LDS_W8(MT_, CM_)                                                                            \
        if (mt == MT_ && cm == CM_) {                                                                    \
            if (g8.tile2d) launch_lds_inst<2, MT_, 4, true, 0, false, CM_, 8>(a, wl, batch, ipw, g8, s); \
            else launch_lds_inst<2, MT_, 4, false, 0, false, CM_, 8>(a, wl, batch, ipw, g8, s);          \
            return true;                                                                                 \
        }
        IRMV_LDS_W8(2, 0) IRMV_LDS_W8(2, 2) IRMV_LDS_W8(1, 0) IRMV_LDS_W8(1, 4)
#undef IRMV_LDS_W8
        return false;
    }
    const LdsGeom g = lds_geom(a, stride, mt, nt);
    if (!g.bytes) return false;
    if (cm) {   // chunk-major order over the workgroup's cm = ipw images (16 cm mt nt accumulator registers)
        if (cm != ipw || pf2) return false;
        if (a.n2 > 0 && (stride != 1 || nt != 4 || a.cout_pad != 64 || !a.pair || a.res)) return false;
#define IRMV_LDS_CM(ST_, MT_, NT_, CM_, N2_)                                                               \
        if (stride == ST_ && mt == MT_ && nt == NT_ && cm == CM_ && a.n2 == N2_) {                         \
            if (g.tile2d) launch_lds_inst<ST_, MT_, NT_, true, N2_, false, CM_>(a, wl, batch, ipw, g, s);  \
            else launch_lds_inst<ST_, MT_, NT_, false, N2_, false, CM_>(a, wl, batch, ipw, g, s);          \
            return true;                                                                                   \
        }
        IRMV_LDS_CM(1, 1, 4, 4, 0) IRMV_LDS_CM(1, 1, 4, 2, 0) IRMV_LDS_CM(1, 2, 4, 2, 0)
        IRMV_LDS_CM(2, 1, 4, 4, 0) IRMV_LDS_CM(2, 1, 4, 2, 0) IRMV_LDS_CM(2, 2, 4, 2, 0)
        IRMV_LDS_CM(1, 1, 4, 4, 1) IRMV_LDS_CM(1, 2, 4, 2, 1) IRMV_LDS_CM(1, 1, 4, 4, 4) IRMV_LDS_CM(1, 2, 4, 2, 4)
#undef IRMV_LDS_CM
        return false;
    }
    if (pf2) {   // two- (1) or four- (2) steps-ahead staging: the small pixel tiles (MT = 1), plain epilogue
        if (mt != 1 || a.n2 > 0) return false;
        if (pf2 == 2) {   // four register sets: one 16-channel tile, maps small enough for lds_pmax(.., pf = 2) pieces per thread
            const LdsGeom g4 = lds_geom(a, stride, 1, nt, 4, 0, false, 2);
            if (nt != 1 || !g4.bytes) return false;
            if (stride == 1) {
                if (g4.tile2d) launch_lds_inst<1, 1, 1, true, 0, 2>(a, wl, batch, ipw, g4, s);
                else launch_lds_inst<1, 1, 1, false, 0, 2>(a, wl, batch, ipw, g4, s);
            } else {
                if (g4.tile2d) launch_lds_inst<2, 1, 1, true, 0, 2>(a, wl, batch, ipw, g4, s);
                else launch_lds_inst<2, 1, 1, false, 0, 2>(a, wl, batch, ipw, g4, s);
            }
            return true;
        }
#define IRMV_LDS_P(ST_, NT_)                                                                     \
        if (stride == ST_ && nt == NT_) {                                                        \
            if (g.tile2d) launch_lds_inst<ST_, 1, NT_, true, 0, 1>(a, wl, batch, ipw, g, s);  \
            else launch_lds_inst<ST_, 1, NT_, false, 0, 1>(a, wl, batch, ipw, g, s);          \
            return true;                                                                         \
        }
        IRMV_LDS_P(1, 1) IRMV_LDS_P(1, 2) IRMV_LDS_P(1, 4) IRMV_LDS_P(2, 1) IRMV_LDS_P(2, 2) IRMV_LDS_P(2, 4)
#undef IRMV_LDS_P
        return false;
    }
    if (a.n2 > 0) {   // fused trailing 1x1: the workgroup must own all 64 channels of every pixel
        if (stride != 1 || nt != 4 || a.cout_pad != 64 || !a.pair || a.res) return false;
#define IRMV_LDS_F(MT_, N2_)                                                                   \
        if (mt == MT_ && a.n2 == N2_) {                                                        \
            if (g.tile2d) launch_lds_inst<1, MT_, 4, true, N2_>(a, wl, batch, ipw, g, s);      \
            else launch_lds_inst<1, MT_, 4, false, N2_>(a, wl, batch, ipw, g, s);              \
            return true;                                                                       \
        }
        IRMV_LDS_F(1, 1) IRMV_LDS_F(2, 1) IRMV_LDS_F(4, 1) IRMV_LDS_F(1, 4) IRMV_LDS_F(2, 4) IRMV_LDS_F(4, 4)
#undef IRMV_LDS_F
        return false;
    }
#define IRMV_LDS(ST_, MT_, NT_)                                                                \
    if (stride == ST_ && mt == MT_ && nt == NT_) {                                             \
        if (g.tile2d) launch_lds_inst<ST_, MT_, NT_, true, 0>(a, wl, batch, ipw, g, s);                \
        else launch_lds_inst<ST_, MT_, NT_, false, 0>(a, wl, batch, ipw, g, s);                        \
        return true;                                                                           \
    }
    IRMV_LDS(1, 1, 1) IRMV_LDS(1, 2, 1) IRMV_LDS(1, 4, 1)
    IRMV_LDS(1, 1, 2) IRMV_LDS(1, 2, 2) IRMV_LDS(1, 4, 2) IRMV_LDS(1, 1, 4) IRMV_LDS(1, 2, 4) IRMV_LDS(1, 4, 4)
    IRMV_LDS(2, 1, 1) IRMV_LDS(2, 2, 1) IRMV_LDS(2, 4, 1)
    IRMV_LDS(2, 1, 2) IRMV_LDS(2, 2, 2) IRMV_LDS(2, 4, 2) IRMV_LDS(2, 1, 4) IRMV_LDS(2, 2, 4) IRMV_LDS(2, 4, 4)
#undef IRMV_LDS
    return false;
}

bool launch_conv_lds_multi(int nt, const ConvArgs *a, const half_t *const *wl, int n, int batch, hipStream_t s)
{   // (round 4: deeper staging -- two and four steps ahead -- was built for this launch and measured on the first-stage
    // group of a lone frame: 14.4 us one step ahead, 18.3 / 17.5 us two / four ahead at nt = 2.  The launch is 1.9 GFLOP on
    // 500 workgroups, not a chain of round trips; removed again.)
    if (n < 1 || n > kMultiMax || !(nt == 1 || nt == 2 || nt == 4)) return false;
    LdsMultiArgs m{};
    m.n = n;
    int total = 0;
    size_t bytes = 0;
    for (int k = 0; k < n; k++) {
        const LdsGeom g = lds_geom(a[k], 1, 1, nt);
        if (!g.bytes) return false;
        if (a[k].n2 > 0 && (nt != 4 || a[k].cout_pad != 64 || !a[k].pair || a[k].res || !(a[k].n2 == 1 || a[k].n2 == 4))) return false;
        LdsMember &l = m.m[k];
        l.a = a[k]; l.wl = wl[k];
        l.tiles_x = g.tiles_x; l.tiles_y = g.tiles_y; l.twc_log2 = g.twc_log2; l.patch_bytes = g.patch_bytes; l.batch = batch; l.tile2d = g.tile2d ? 1 : 0;
        l.gx = g.tiles_x * g.tiles_y * batch;       // one image per workgroup
        l.gy = a[k].cout_pad / (16 * nt);
        m.start[k] = total;
        total += l.gx * l.gy;
        bytes = bytes > g.bytes ? bytes : g.bytes;
    }
    m.start[n] = total;
#define IRMV_LDS_M(NT_)                                                                                                  \
    if (nt == NT_) {                                                                                                      \
        static unsigned long long attr_done = 0;                                                                          \
        once_per_device(attr_done, [] {                                                                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_lds_multi<1, 1, NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        });                                                                                                               \
        hipLaunchKernelGGL((conv3x3_lds_multi<1, 1, NT_>), dim3(total), dim3(256), bytes, s, m);                          \
        return true;                                                                                                      \
    }
    IRMV_LDS_M(1) IRMV_LDS_M(2) IRMV_LDS_M(4)
#undef IRMV_LDS_M
    return false;
}

// SPPF (SURVEY.md Appendix A "Blocks"): p1 = maxpool5(a), p2 = maxpool5(p1),
// p3 = maxpool5(p2).  Stride-1 max pools compose, so p2 / p3 are the clipped 9x9 /
// 13x13 window maxima of `a`, and every window maximum is separable.
//
// LDS version: one workgroup per (frame, CW-channel slab).  The slab [H*W][CW] is
// staged once; pass 1 writes the three horizontal maxima (radius 2, 4, 6) to LDS,
// pass 2 takes the vertical maxima of those: 13 + 27 LDS reads per (pixel, 8
// channels) instead of 169 global loads.  Max is exact, so this is bit-identical
// to the chained pools.
// (round 4: the maxima are v_pk_max_f16 on the eight halves of an item -- four instructions; the compare-and-select the
// compiler makes of `v > m ? v : m` on fp16 vectors was ~ 16, and with it the kernel issued 29 vector instructions per
// LDS read.  Inline asm: the builtin max canonicalises both operands first, three instructions instead of one.  Activations
// are never NaN; of two zeros of different sign either may come out, which no later layer can tell apart.)
__device__ __forceinline__ half8 hmax8(half8 a, half8 b)
{
    u32x4_t x = __builtin_bit_cast(u32x4_t, a);
    const u32x4_t y = __builtin_bit_cast(u32x4_t, b);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        unsigned int r;
        asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(x[i]), "v"(y[i]));
        x[i] = r;
    }
    return __builtin_bit_cast(half8, x);
}

__global__ __launch_bounds__(512) void sppf_pool_lds_kernel(half_t *buf, int H, int W, int C, int CW)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int HW = H * W, chunks = CW >> 3, cs = chunks == 1 ? 0 : (chunks == 2 ? 1 : 2);   // (slabs are 8, 16 or 32 channels wide)
    half8 *sa = reinterpret_cast<half8 *>(smem);          // [HW][chunks]
    half8 *s2 = sa + (size_t)HW * chunks;                 // horizontal max, radius 2
    half8 *s4 = s2 + (size_t)HW * chunks;
    half8 *s6 = s4 + (size_t)HW * chunks;
    const int slabs = C / CW;
    const int b = blockIdx.x / slabs, slab = blockIdx.x - b * slabs;
    const int ld = 4 * C;
    half_t *base = buf + (size_t)b * HW * ld + slab * CW;
    const int items = HW * chunks;
    for (int t = threadIdx.x; t < items; t += blockDim.x) {
        const int p = t >> cs, ch = t & (chunks - 1);
        sa[t] = *reinterpret_cast<const half8 *>(base + (size_t)p * ld + ch * 8);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < items; t += blockDim.x) {
        const int p = t >> cs, ch = t & (chunks - 1);
        const int y = p / W, x = p - y * W;
        // all twelve neighbours requested at once (an index clamped into the row reads a pixel that is in the window anyway,
        // or the pixel itself: max is idempotent), then three nested maxima
        half8 v[13];
#pragma unroll
        for (int d = -6; d <= 6; d++) {
            int xx = x + d;
            xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            v[d + 6] = sa[(y * W + xx) * chunks + ch];
        }
        half8 m2 = hmax8(hmax8(hmax8(v[4], v[5]), hmax8(v[7], v[8])), v[6]);
        half8 m4 = hmax8(hmax8(hmax8(v[2], v[3]), hmax8(v[9], v[10])), m2);
        half8 m6 = hmax8(hmax8(hmax8(v[0], v[1]), hmax8(v[11], v[12])), m4);
        s2[t] = m2; s4[t] = m4; s6[t] = m6;
    }
    __syncthreads();
    // vertical maxima, TWO vertically adjacent pixels per item: their windows share 12 / 8 / 4 of the 13 / 9 / 5 rows
    // (14 + 10 + 6 = 30 LDS reads for two pixels instead of 54)
    const int HP = (H + 1) >> 1, items2 = HP * W * chunks;
    for (int t = threadIdx.x; t < items2; t += blockDim.x) {
        const int pp = t >> cs, ch = t & (chunks - 1);
        const int yp = pp / W, x = pp - yp * W, y0 = 2 * yp;
        auto row = [&](int yy) { return ((yy < 0 ? 0 : (yy >= H ? H - 1 : yy)) * W + x) * chunks + ch; };
        half8 r6[14], r4[10], r2[6];
#pragma unroll
        for (int i = 0; i < 14; i++) r6[i] = s6[row(y0 - 6 + i)];
#pragma unroll
        for (int i = 0; i < 10; i++) r4[i] = s4[row(y0 - 4 + i)];
#pragma unroll
        for (int i = 0; i < 6; i++) r2[i] = s2[row(y0 - 2 + i)];
        // shared middles first, then the row only one of the two windows holds
        half8 c6 = r6[1], c4 = r4[1], c2 = r2[1];
#pragma unroll
        for (int i = 2; i < 13; i++) c6 = hmax8(c6, r6[i]);
#pragma unroll
        for (int i = 2; i < 9; i++) c4 = hmax8(c4, r4[i]);
#pragma unroll
        for (int i = 2; i < 5; i++) c2 = hmax8(c2, r2[i]);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int y = y0 + k;
            if (y >= H) break;
            half_t *o = base + (size_t)(y * W + x) * ld + ch * 8;
            *reinterpret_cast<half8 *>(o + C) = hmax8(c2, k ? r2[5] : r2[0]);
            *reinterpret_cast<half8 *>(o + 2 * C) = hmax8(c4, k ? r4[9] : r4[0]);
            *reinterpret_cast<half8 *>(o + 3 * C) = hmax8(c6, k ? r6[13] : r6[0]);
        }
    }
}

// Global-memory version for feature maps too large for the LDS slab: one lane per
// (pixel, 8-channel chunk), three nested maxima over the 13x13 neighbourhood.
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
    // Channel slab of a workgroup: its four [H*W][CW] fp16 images must fit in LDS.  A slab is read as CW * 2 contiguous
    // bytes per pixel out of a 4 C * 2-byte row, so wide slabs use the memory system better (8 channels = 16 of every
    // 1024 bytes); narrow slabs give more workgroups.  Widest slab that still leaves ~3 workgroups per CU, else 8.
    int cw = 0;
    auto fits = [&](int c) { return C % c == 0 && (size_t)4 * H * W * c * 2 <= 150 * 1024; };
    for (int c = 32; c >= 8 && !cw; c >>= 1)
        if (fits(c) && (long)batch * (C / c) >= 768) cw = c;
    for (int c = 8; c <= 32 && !cw; c <<= 1)   // few frames: the narrowest slab = the most workgroups
        if (fits(c)) cw = c;
    if (cw) {
        const size_t lds = (size_t)4 * H * W * cw * 2;
        static unsigned long long attr_done = 0;
        once_per_device(attr_done, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(sppf_pool_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        // 512 lanes: a slab's H * W * cw / 8 items (400 .. 1600 at a 640 net) take half the rounds of each of the three
        // barrier-separated passes, and the one workgroup a CU holds (LDS) puts two waves on every SIMD instead of one
        hipLaunchKernelGGL(sppf_pool_lds_kernel, dim3(batch * (C / cw)), dim3(512), lds, s, buf, H, W, C, cw);
        return;
    }
    const int total = batch * H * W * (C >> 3);
    hipLaunchKernelGGL(sppf_pool_kernel, dim3((total + 255) / 256), dim3(256), 0, s, buf, batch, H, W, C);
}

}  // namespace irmv
