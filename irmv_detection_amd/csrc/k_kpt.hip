// The keypoint branch of one Detect level -- 3x3 (Cin -> 16), 3x3 (16 -> 16), 1x1 (16 -> nk) -- as ONE launch.
//
// Inside the reference's TensorRT plan (src/yolo_engine.cpp:105) these are the layers model.22.cv4.<level>.{0,1,2} of the
// pose-style head the north-star variant carries.  As layers they are the worst-fed kernels of a batched step: 16 output
// channels give every B fragment exactly one MFMA, the first conv re-reads the level's whole input for 0.3 % of the network's
// arithmetic, the second is a launch of its own over a 16-channel tensor (6 launches, 115 us per 128 frames at the end of
// round 5's first half).  Here a workgroup of three waves owns a 10 x 10 pixel tile:
//
//   stage 1   the first 3x3 on the tile + 1-pixel halo: a 12 x 12 region = 144 pixels = NINE 16-pixel MFMA tiles exactly, three
//             per wave (4 x 4 pixel blocks); input region 14 x 14 staged in LDS 64 channels at a time (two swizzled 32-channel
//             planes, see below), the next slab's loads in registers under the current one's MFMAs; weight fragments straight
//             from memory through a ring of six (the same 18 - 72 KB for every workgroup: cache hits) -- no stage-1 weights in
//             LDS, four workgroups per CU.  A workgroup keeps its tile for up to eight consecutive images.
//   stage 2   the second 3x3 on the tile out of a [144 pixel][16 ch] LDS plane (zero where the region leaves the image: the
//             conv's padding), five k-steps of two taps each, fragments from LDS (staged once per workgroup)
//   stage 3   the final 1x1 as one v_mfma_f32_16x16x16_f16 per 16 pixels on the activated sums, fp32 into the head record
//
// Rounding points and K order are the per-layer kernels': stage 1 walks (32-channel chunk, tap) on the LDS family's nt = 1
// weight packing with accumulators starting at the log2 e-scaled bias; stage 2 is the Cin = 16 direct kernel's k-step (lanes
// g < 2: tap 2 i, g >= 2: tap 2 i + 1, the tenth tap zero); stage 3 is that kernel's K16_FUSE epilogue -- bit-identical
// (tests/test_gpu_engine.py::test_keypoint_branch_kernel_is_bitwise_the_layers).
#include "irmv_common.hpp"

namespace irmv {

namespace {
constexpr int KT = kKpt3Tile;
constexpr int K1W = KT + 4, K1N = K1W * K1W;   // input region (halo 2): 14 x 14
constexpr int K2W = KT + 2, K2N = K2W * K2W;   // first conv's output region (halo 1): 12 x 12 = 9 MFMA tiles
// LDS layouts, both free of bank conflicts (a ds_read_b128 is served in four groups of 16 lanes -- the first: lanes 0-3, 12-15
// and 20-27, MI355X_MICROARCH.md -- and a group wants 16 distinct 16-byte slots of the 256-byte bank window; checked by
// simulation over every (tile, k-step) of the kernel, counted on the device: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE):
//   slab   two planes of 32 channels, [plane][pixel][64 B], no padding; the 16-byte slot of lane group g inside a pixel is
//          g ^ 2 (region row & 1).  An MFMA tile is a 4 x 4 pixel BLOCK (the 12 x 12 region is 3 x 3 of them; wave w owns block
//          row w): a group's lanes are then two block rows, eight pixels 64 B apart, and the row parity moves every other row
//          onto the slots the first one left free.  (A [pixel][64 ch] slab at 144 / 160 bytes per pixel with 16 consecutive
//          pixels per tile measured 44 % of its LDS cycles as conflicts: the tile's pixels wrap around the 12-wide region.)
//   plane  [144 pixel][16 ch] at 32 bytes per pixel, 4 x 4 blocks again (nine over the 10 x 10 tile, the lanes outside it
//          compute on whatever the LDS holds and store nothing): rows of a block are 12 pixels = 384 B apart.
constexpr int KPL = K1N * 64;                  // bytes of a 32-channel slab plane
constexpr int KQS = 32;                        // bytes per pixel of the 16-channel plane
constexpr int KNT = 192;                       // three waves
constexpr int KNP = (K1N * 8 + KNT - 1) / KNT; // 16-byte pieces of a slab per thread: 9
static_assert(K2N == 9 * 16, "the halo region is nine MFMA tiles");
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
}  // namespace

// STEPS = Cin / 64 (1, 2, 4).  A workgroup keeps its tile for `ipw` consecutive images: the slab of the next (image, slab) step
// is on its way while this one's MFMAs run, stage 1's fragments are fetched once (Cin = 64: all 18 in registers for the
// workgroup's lifetime) or ride a ring that runs on across images.
template <int STEPS>
__global__ __launch_bounds__(KNT) __attribute__((amdgpu_waves_per_eu(3))) void kpt3_kernel(Kpt3Args a, int batch, int ipw, int xcd)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *s_in = smem;                  // one 64-channel slab of the input region, zero outside the image (= the 3x3's padding)
    uint8_t *s_t = smem + 2 * KPL;         // first conv's output region, 16 channels
    half8 *s_w2 = reinterpret_cast<half8 *>(smem + 2 * KPL + K2N * KQS);   // second conv's fragments: 4.5 KiB (lanes 32 .. 63 of the fifth hold the tenth tap: zeros, not stored)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int tiles = a.tiles_x * a.tiles_y, groups = (batch + ipw - 1) / ipw;
    int tile, grp;
    tile_image(blockIdx.x, tiles, groups, xcd, tile, grp);   // an image group's tiles on one XCD: they share halo pixels (irmv_common.hpp)
    const int img0 = grp * ipw, nimg = min(ipw, batch - img0);
    const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
    const int oy0 = tyi * KT, ox0 = txi * KT;
    const int H = a.H, W = a.W;
    constexpr int KSTEPS = STEPS * 18;

    // ---- staging plan of a slab: piece e -> (region pixel, 16-byte eighth) ----
    // Loads go through buffer descriptors (base in SGPRs, 32-bit per-lane offset, wave-uniform scalar offset): with flat
    // addresses the compiler kept one 64-bit VGPR pair per weight fragment and slab piece alive across the image loop
    // (36 + 9 pairs at Cin = 128: 240 VGPRs, one wave per SIMD less).
    int src[KNP];             // (byte offsets from a.x: a batch's level input is < 2^31 bytes; pieces outside the image: an offset past the
                              //  descriptor's range -- the hardware answers such a load with zeros, the 3x3's padding, and no select is needed)
    static_assert(KNT % 8 == 0 && (KNP - 1) * KNT < K1N * 8, "piece i of a thread: pixel (tid >> 3) + i * KNT / 8, eighth tid & 7; only the last i can fall off the region");
    // piece (pixel, eighth q): plane q >> 2, slot (q & 3) ^ 2 (row & 1) -- two candidate addresses per thread, one bit per piece
    const int dstA = (tid >> 3) * 64 + ((tid & 7) >> 2) * KPL + (tid & 3) * 16, dstB = dstA ^ 32;
    bool odd[KNP];
    const bool last_piece = tid + (KNP - 1) * KNT < K1N * 8;
#pragma unroll
    for (int i = 0; i < KNP; i++) {
        const int e = tid + i * KNT, px = e >> 3, q = e & 7;
        const int ly = px / K1W, lx = px - ly * K1W;
        const int gy = oy0 - 2 + ly, gx = ox0 - 2 + lx;
        const bool in = e < K1N * 8 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        odd[i] = (ly & 1) != 0;
        src[i] = in ? (((img0 * H + gy) * W + gx) * a.x_ld + q * 8) * 2 : 0x7fffffff;
    }
    const int img_stride = H * W * a.x_ld * 2;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(a.x), 0, batch * img_stride, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(a.w1), 0, KSTEPS * 1024, 0x00020000);
    auto load_w = [&](int k) { return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, k * 1024, 0)); };   // nt = 1 packing: [chunk][tap][lane]
    u32x4 v[KNP];
    int l_im = 0, l_s = 0;   // loader position: the (image, slab) step the next issue_slab fetches
    auto issue_slab = [&]() {   // unconditional (a branch around loads makes every later wait a full drain): past the end the last slab again
        const int off = __builtin_amdgcn_readfirstlane(l_im * img_stride + l_s * 128);   // (wave-uniform, but not provably so: left alone every load sits in a waterfall loop)
#pragma unroll
        for (int i = 0; i < KNP; i++) v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, src[i], off, 0));
        if (!(l_im == nimg - 1 && l_s == STEPS - 1)) { if (++l_s == STEPS) { l_s = 0; l_im++; } }
    };
    auto write_slab = [&]() {
#pragma unroll
        for (int i = 0; i < KNP; i++)
            if (i + 1 < KNP || last_piece) *reinterpret_cast<u32x4 *>(s_in + (odd[i] ? dstB : dstA) + i * (KNT / 8) * 64) = v[i];
    };

    // ---- requests in the order their data is needed (the memory counter is in-order): slab 0, stage 1's fragments, then
    // what stages 2 and 3 will want ----
    issue_slab();
    __builtin_amdgcn_sched_barrier(0);
    half8 A[6];   // (Cin = 64 with all 18 fragments in registers for the workgroup's lifetime: 205 VGPRs, two waves per SIMD, 8 % slower)
#pragma unroll
    for (int k = 0; k < 6; k++) A[k] = load_w(k);
    const f32x4 b1 = *reinterpret_cast<const f32x4 *>(a.b1 + g * 4);
    __builtin_amdgcn_sched_barrier(0);
    {
        const half8 *w2p = reinterpret_cast<const half8 *>(a.w2);   // Cin = 16 direct packing: [k-step][lane]
        const half8 t0 = w2p[tid], t1 = w2p[tid + KNT < 288 ? tid + KNT : 287];
        s_w2[tid] = t0;
        if (tid + KNT < 288) s_w2[tid + KNT] = t1;
    }
    const f32x4 b2 = *reinterpret_cast<const f32x4 *>(a.b2 + g * 4);
    const half4 w3f = *reinterpret_cast<const half4 *>(a.w3 + lane * 4);
    const f32x4 b3 = *reinterpret_cast<const f32x4 *>(a.b3 + g * 4);
    __builtin_amdgcn_sched_barrier(0);

    // this lane's pixel of tile i (both stages): block (wave, i) of the 3 x 3, row r >> 2, column r & 3
    const int bly = wave * 4 + (r >> 2), blx = r & 3;
    const int baseE = (bly * K1W + blx) * 64 + ((g ^ (2 * (bly & 1))) & 3) * 16;   // taps of filter rows 0 and 2 (region row parity = bly's)
    const int baseO = (bly * K1W + blx) * 64 + ((g ^ (2 * (~bly & 1))) & 3) * 16;  // filter row 1
    const int tap_lo = g >> 1;   // stage 2: this lane's tap of k-step ks is 2 ks + tap_lo
    int toff2[5];                // ... and its offset in the plane
#pragma unroll
    for (int ks = 0; ks < 5; ks++) {
        const int tap = 2 * ks + tap_lo, tc = tap < 9 ? tap : 0, kh = tc / 3, kw = tc - kh * 3;
        toff2[ks] = ((bly + kh) * K2W + blx + kw) * KQS + (g & 1) * 16;
    }
    half8 Bf[2][3];
    auto frag = [&](int kk, half8 (&B)[3]) {   // kk: k-step inside the slab = (chunk kk / 9, tap kk % 9)
        const int c = kk / 9, tap = kk - c * 9, kh = tap / 3, kw = tap - kh * 3;
        const int off = (kh * K1W + kw) * 64 + c * KPL;
#pragma unroll
        for (int i = 0; i < 3; i++) B[i] = *reinterpret_cast<const half8 *>(s_in + (kh == 1 ? baseO : baseE) + off + i * 256);
    };

    for (int im = 0; im < nimg; im++) {
        // ---- stage 1: 3x3, Cin -> 16, SiLU, on the 12 x 12 region: pixel tiles 3 wave .. 3 wave + 2 ----
        f32x4 acc[3] = {b1, b1, b1};
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            write_slab();
            __syncthreads();
            frag(0, Bf[0]);
#pragma unroll
            for (int kk = 0; kk < 18; kk++) {
                const int k = s * 18 + kk;
                if (kk + 1 < 18) frag(kk + 1, Bf[(kk + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 3; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[kk % 6], Bf[kk & 1][i], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // the next slab's loads go out BEHIND the fragments this slab still needs and in front of the next slab's
                // first six: a fragment requested after them waits for them (in-order counter), and those six are not
                // needed before the slab itself is.  The ring runs on across slabs and images (k wraps).
                if (kk == 12) issue_slab();
                A[kk % 6] = load_w((k + 6) % KSTEPS);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (s + 1 < STEPS) __syncthreads();   // every wave is done reading the slab (the last slab: the barrier behind the epilogue)
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int lx = i * 4 + blx;
            const int gy = oy0 - 1 + bly, gx = ox0 - 1 + lx;
            half4 o = (half4){0, 0, 0, 0};   // outside the image: the second conv's zero padding
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) o = silu_pack4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            *reinterpret_cast<half4 *>(s_t + (bly * K2W + lx) * KQS + g * 8) = o;
        }
        __syncthreads();

        // ---- stages 2 + 3: 3x3 (16 -> 16, SiLU) and the final 1x1 on the tile: the same three blocks per wave ----
        half8 A2[5];
#pragma unroll
        for (int ks = 0; ks < 5; ks++) A2[ks] = s_w2[ks * 64 + (ks == 4 ? (lane & 31) : lane)];
        if (g >= 2) A2[4] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int ly = bly, lx = i * 4 + blx;
            const bool valid = ly < KT && lx < KT;            // (lanes outside the tile read past the plane -- still inside the workgroup's LDS -- and store nothing)
            f32x4 c1 = b2;
            half8 B2[5];
#pragma unroll
            for (int ks = 0; ks < 5; ks++) {
                const u32x4 q = *reinterpret_cast<const u32x4 *>(s_t + toff2[ks] + i * 4 * KQS);
                B2[ks] = __builtin_bit_cast(half8, 2 * ks + tap_lo < 9 ? q : (u32x4){0u, 0u, 0u, 0u});
            }
#pragma unroll
            for (int ks = 0; ks < 5; ks++) c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[ks], B2[ks], c1, 0, 0, 0);
            half4 o = silu_pack4(c1[0], c1[1], c1[2], c1[3]);
            mfma_operand_fence(o);   // (VALU write inside inline asm -> MFMA read: irmv_common.hpp)
            const f32x4 c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(w3f, o, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const f32x4 v2 = (f32x4){c2[0] * kActUnscale + b3[0], c2[1] * kActUnscale + b3[1], c2[2] * kActUnscale + b3[2], c2[3] * kActUnscale + b3[3]};
            const int gy = oy0 + ly, gx = ox0 + lx;
            if (valid && gy < H && gx < W) *reinterpret_cast<f32x4 *>(a.out + ((size_t)((img0 + im) * H + gy) * W + gx) * a.out_ld + g * 4) = v2;
        }
    }
}

bool kpt3_eligible(int cin) { return cin == 64 || cin == 128 || cin == 256; }

bool launch_kpt3(const Kpt3Args &a, int cin, int batch, hipStream_t s)
{
    // images per workgroup: as many as still leave one round of workgroups (256 CUs x 4) -- a workgroup's steps pipeline, its
    // start does not.  Measured at 128 frames: 80 x 80 level 61.8 / 54.7 / 53.1 / 50.6 us at 1 / 2 / 4 / 8, 20 x 20 level 13.4 / 16.4 / 26.7.
    const int tiles = a.tiles_x * a.tiles_y;
    int ipw = 1;
    while (ipw < 8 && (long long)tiles * batch >= 2048LL * ipw) ipw *= 2;
    const dim3 grid(tiles * ((batch + ipw - 1) / ipw)), block(KNT);
    constexpr size_t lds = (size_t)2 * KPL + (size_t)K2N * KQS + 4608;   // 34 304 bytes: four workgroups per CU
    static_assert(lds <= 40 * 1024, "four workgroups per CU");
    const int xcd = xcd_image_order();
    switch (cin) {
    case 64: hipLaunchKernelGGL((kpt3_kernel<1>), grid, block, lds, s, a, batch, ipw, xcd); return true;
    case 128: hipLaunchKernelGGL((kpt3_kernel<2>), grid, block, lds, s, a, batch, ipw, xcd); return true;
    case 256: hipLaunchKernelGGL((kpt3_kernel<4>), grid, block, lds, s, a, batch, ipw, xcd); return true;
    default: return false;
    }
}

}  // namespace irmv
