// Shared declarations of the gfx950 kernels and their host launchers.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

namespace irmv {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- activation scale (round 4) ---------------------------------------------------
// Every fp16 activation tensor between the layers holds a' = log2(e) * a.  With biases pre-multiplied by log2(e) at load
// time (weights unchanged) a SiLU layer's accumulator is y = log2(e) * x, and
//     log2(e) * x * sigmoid(x) = y * rcp(1 + exp2(-y)):
// v_exp_f32 (with a free negate modifier), v_add, v_rcp, v_pk_mul, v_cvt_pk -- the v_mul in front of the exponential that
// exp(-x) = exp2(-x * log2 e) costs per output value is gone, and so is the bias add (accumulators start AT the bias).
// model.0 reads the unscaled image: its epilogue scales the accumulator (one fma, the bias rides in it); the Detect
// finals (no activation, fp32 out) undo the scale the same way: out = acc * ln 2 + bias.  Read-backs of activation
// tensors multiply by ln 2 on the host (engine.cpp read_tensor_f32).
// RANGE: a stored activation overflows fp16 where log2(e) * |a| > 65504, i.e. at |a| > 45 403 instead of 65 504 (it becomes
// +-inf, as a plain fp16 pipeline's does above 65 504; nothing saturates).  Trained detection networks keep activations
// within a few hundred; tests/test_gpu_engine.py::test_activation_range_under_the_log2e_scale pins the behaviour on both sides
// of the edge (finite and equal to the fp16-emulating oracle's at 4e4, inf past 45.4e3).  sppf_pool's v_pk_max_f16 assumes
// no NaN: an inf is ordered like any value, a NaN (inf - inf in a later layer) can only arise past that edge.
constexpr float kActScale = 1.44269504088896341f;    // log2(e)
constexpr float kActUnscale = 0.693147180559945309f; // ln 2
// Rounding is pinned by hand.  Left to the compiler, (half)(y * r) becomes v_pk_mul_f32 + v_cvt_pk_f16_f32 (two roundings)
// where a lane's values pair up in registers and v_fma_mixlo/hi_f16 (the exact product rounded ONCE) where one is left
// over -- which values those are differs from kernel to kernel, so two kernels computing the same layer disagreed in one
// output of 70 000.  Every SiLU output is therefore produced by v_fma_mix*_f16 explicitly: exp, add, rcp, fma_mix -- four
// vector instructions per output value, no separate multiply or convert -- and the shortcut variants (activation + residual,
// rounded after the add) pin their fp32 intermediates with empty asm statements.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float silu_rcp(float y) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-y)); }
// two activated outputs as one packed half2: lo = half(y0 * sigma(y0)), hi = half(y1 * sigma(y1)), each rounded once
__device__ __forceinline__ unsigned int silu_pack2(float y0, float y1)
{
    const float r0 = silu_rcp(y0), r1 = silu_rcp(y1);
    unsigned int o;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(o) : "v"(y0), "v"(r0));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(o) : "v"(y1), "v"(r1));
    return o;
}
__device__ __forceinline__ half8 silu_pack8(float y0, float y1, float y2, float y3, float y4, float y5, float y6, float y7)
{
    return __builtin_bit_cast(half8, (u32x4_t){silu_pack2(y0, y1), silu_pack2(y2, y3), silu_pack2(y4, y5), silu_pack2(y6, y7)});
}
__device__ __forceinline__ half4 silu_pack4(float y0, float y1, float y2, float y3)
{
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(half4, (u32x2_t){silu_pack2(y0, y1), silu_pack2(y2, y3)});
}
// An MFMA that takes a SiLU output as its A / B operand straight from the registers (the fused finals of k_conv.hip): the
// hardware needs two wait states between a vector-ALU write of a VGPR and an MFMA's read of it, and the compiler, which
// inserts them for instructions it knows, does not look inside the inline asm above -- with the last v_fma_mixhi_f16 one
// instruction in front of the MFMA the matrix pipe read the register's OLD value (round 5: the keypoint final came out
// wrong in 60 % of its outputs).  These ties put an `s_nop 1` between the asm's writes and whatever reads the value next.
__device__ __forceinline__ void mfma_operand_fence(half4 &x) { asm volatile("s_nop 1" : "+v"(x)); }
__device__ __forceinline__ void mfma_operand_fence(half8 &x, half8 &y) { asm volatile("s_nop 1" : "+v"(x), "+v"(y)); }
// shortcut layers: half(round32(y * sigma(y)) + res), every intermediate rounded where it is written
__device__ __forceinline__ half_t silu_add_res(float y, float res)
{
    float p = y * silu_rcp(y);
    asm("" : "+v"(p));
    float s = p + res;
    asm("" : "+v"(s));
    return (half_t)s;
}

// Workgroup id -> (tile, image) with all tiles of an image on ONE XCD.  The hardware deals consecutive workgroup ids to the 8
// XCDs in turn, and each XCD has its own L2: with the plain (tile, image) grid every neighbour of a tile runs on another XCD
// and re-reads the shared halo from memory (profiles/r04_traffic.json: 1.25 - 1.45 x the algorithmic bytes in the fused C2f kernels, 1.30 x in the front kernel).
// Here id = 8 j + x  ->  image 8 (j / tiles) + x, tile j % tiles: XCD x walks whole images, their halos meet in its L2.
// The batch's last (batch % 8) images take the plain order.  `xcd` = 0 switches the mapping off (IRMV_XCD_IMAGES=0).
__device__ __forceinline__ void tile_image(int id, int tiles, int batch, int xcd, int &tile, int &image)
{
    const int full = xcd ? (batch >> 3) * 8 * tiles : 0;
    if (id < full) {
        const int j = id >> 3;
        image = (j / tiles) * 8 + (id & 7);
        tile = j % tiles;
    } else {
        const int r = id - full;
        image = (full / tiles) + r / tiles;
        tile = r % tiles;
    }
}

// host side: the mapping is on unless IRMV_XCD_IMAGES=0 (read at every launch CALL: engine creation, tuning, graph capture)
inline int xcd_image_order()
{
    const char *v = getenv("IRMV_XCD_IMAGES");
    return (v && v[0] == '0') ? 0 : 1;
}

// lockstep resident-weight 3x3 kernel: waves 4 .. 7 defer their epilogue by one image (k_conv.hip); IRMV_WRES_STAGGER=0: off
inline int wres_stagger()
{
    const char *v = getenv("IRMV_WRES_STAGGER");
    return (v && v[0] == '0') ? 0 : 1;
}

constexpr int kHeadRec = 96;   // fp32 per anchor: box 64 | cls 16 (nc<=16) | kpt 16 (nk<=16)
constexpr int kClsOff = 64;
constexpr int kKptOff = 80;
constexpr int kCandCap = 8192; // == IRMV_CAND_CAP
constexpr int kMaxDetCap = 256;

// ---- preprocess ------------------------------------------------------------
// One bilinear tap table entry per destination coordinate (built on the host
// with the integer arithmetic of the oracle's axis_tap; rotate180 already folded
// into i0/i1).  i0 < 0 marks a letterbox pad coordinate.
struct AxisTap { int32_t i0, i1, w1, pad; };

struct PreArgs {
    const uint8_t *src;   // [B][sh][sw][3]
    half_t *dst;          // [B][net][net][4]
    const AxisTap *tx, *ty;
    int sw, sh, net, swap_rb;
    size_t src_slot_bytes;
};
void launch_preprocess(const PreArgs &a, int batch, hipStream_t s);
void launch_rotate180(const uint8_t *src, uint8_t *dst, int sw, int sh, hipStream_t s);
void launch_upload_frames(const uint8_t *src_host_mapped, uint8_t *dst, size_t bytes, int blocks, hipStream_t s);   // bytes % 16 == 0, both pointers 16-byte aligned

// model.0.conv: 3x3 s2, 3(+1 pad) -> 16, SiLU
struct Conv0Args {
    const half_t *x;      // [B][net][net][4]
    half_t *y;            // [B][net/2][net/2][16]
    const half_t *w;      // packed MFMA A fragments [2 k-steps][64 lanes][8]; k = kh*16 + slot*4 + c
    const float *b;       // [16]
    int net, batch;
};
void launch_conv0(const Conv0Args &a, hipStream_t s);

// fused front: preprocess + model.0.conv + model.1.conv (k_front.hip)
constexpr int kFrontTileY = 4, kFrontTileX = 16;   // model.1 output tile of one workgroup
constexpr int kFrontTileYDirect = 8;               // ... of engines whose tiles all read their source directly (columns at exactly 2 : 1)
constexpr int kFrontStageMax = 128 * 1024;         // most LDS the tile's source region may take (else: the three kernels)
struct FrontArgs {
    const uint8_t *src;   // [B][sh][sw][3]
    size_t src_slot_bytes;
    const AxisTap *tx, *ty;
    int vx0, vx1, vy0, vy1;   // net-input columns / rows with a source tap: [v0, v1) (the rest is letterbox padding)
    int sw, sh, net, swap_rb;
    int fastx, fx_i0, fx_step;   // columns at exactly 2 : 1: column vx0 + k blends the source pair (m, m + 1), m = fx_i0 + fx_step k (fx_step = +-2), weights 1/2; fastx bit 1: tiles without padding read the source straight into registers
    const half_t *w0;     // model.0.conv fragments (Conv0Args::w)
    const float *b0;
    const half_t *w1;     // model.1.conv, direct-family packing (Cin = 16: 5 k-steps x 2 paired tiles)
    const float *b1;
    half_t *out;          // [B][net/4][net/4][out_ld]
    int out_ld;
    int tiles_x, tiles_y;
    int tile_y;           // kFrontTileY, or kFrontTileYDirect (needs fastx bit 1)
    int stage_bytes;      // dynamic LDS: >= the largest tile's source region (staged path) and >= front_min_stage_bytes(tile_y)
};
int front_min_stage_bytes(int tile_y = kFrontTileY);
bool front_prepare();
bool launch_front(const FrontArgs &a, int batch, hipStream_t s);

// fused model.2 (C2f: cv1, one shortcut Bottleneck of two 3x3 16 -> 16 convs, cv2) (k_c2f.hip)
constexpr int kC2fTile = 16;
struct C2fArgs {
    const half_t *x;      // block input [B][S][S][x_ld], 32 channels used
    int x_ld;
    half_t *out;          // block output [B][S][S][out_ld], 32 channels
    int out_ld;
    int S, tiles;         // spatial size, tiles per side
    const half_t *w_cv1, *w_m1, *w_m2, *w_cv2;   // direct-family packings of the four layers
    const float *b_cv1, *b_m1, *b_m2, *b_cv2;
};
void launch_c2f2(const C2fArgs &a, int batch, hipStream_t s);

// ---- ShuffleNetV2 stage operators (k_shuffle.hip) ------------------------------
struct DwArgs {
    const half_t *x;      // [B][Hin][Win][x_ld], already offset to the first channel
    int x_ld;
    half_t *y;            // [B][Hout][Wout][y_ld], already offset to the first channel
    int y_ld;
    const half_t *w;      // [9][C] fp16, tap-major
    const float *b;       // [C]
    int Hin, Win, Hout, Wout, C, stride;   // C a multiple of 8
};
void launch_dwconv3x3(const DwArgs &a, int batch, hipStream_t s);
struct ShufArgs {
    const half_t *a, *b;  // [pixels][a_ld / b_ld], already offset to the first channel
    int a_ld, b_ld;
    half_t *out;          // [pixels][out_ld]: out[2 i] = a[i], out[2 i + 1] = b[i], i < bc
    int out_ld, bc;       // bc a multiple of 4
    size_t pixels;        // batch * H * W
};
void launch_shuffle_cat(const ShufArgs &a, hipStream_t s);

// ---- implicit-GEMM conv on MFMA ----------------------------------------------
struct ConvSeg {
    const half_t *p;  // base pointer, already offset to the segment's first channel
    int ld;           // channel stride of a pixel, in elements
    int C;            // channels taken from this segment (multiple of 8); 0 = unused
    int shift;        // 1 = the segment is stored at half resolution (nearest 2x upsample folded in)
};

// fused C2f blocks with a 32-channel hidden width (k_c2f.hip): mode 0 = whole block (n = 1), 1 = cv1 + first
// bottleneck, 2 = last bottleneck + cv2 (n = 2)
constexpr int kC2f32TileH = 8, kC2f32TileW = 16;
struct C2f32Args {
    ConvSeg s0, s1;       // block input (modes 0, 1): up to two K segments, s0 optionally at half resolution
    int cin1;             // s0.C + s1.C
    half_t *cat;          // the block's concat buffer [B][H][W][cat_ld]: y0 | y1 | y2 [| y3], 32 channels each
    int cat_ld, prev_coff;   // mode 2: channel offset of the previous bottleneck's output inside cat
    half_t *out;          // block output [B][H][W][out_ld], 64 channels (modes 0, 2)
    int out_ld;
    int H, W, tiles_x, tiles_y;
    const half_t *w_cv1, *w_m1, *w_m2, *w_cv2;   // direct-family packings
    const float *b_cv1, *b_m1, *b_m2, *b_cv2;
};
size_t c2f32_lds_bytes(int mode);
bool launch_c2f32(int mode, bool shortcut, const C2f32Args &a, int batch, hipStream_t s);

// one 64-channel Bottleneck [+ the block's cv2] of a C2f block in one launch, single-frame steps (k_bneck.hip)
constexpr int kBneckTile = 8;
struct BneckArgs {
    const half_t *yin;    // the Bottleneck's input: a 64-channel slice of the block's concat buffer [B][H][W][yin_ld], offset to its first channel
    int yin_ld;
    half_t *ynext;        // mode A: the next 64-channel slice of the concat buffer (same pixel indexing)
    int ynext_ld;
    const half_t *cat;    // mode B: the concat buffer from channel 0 (cv2 reads the slices in front of y_in from here)
    int cat_ld;
    half_t *out;          // mode B: block output [B][H][W][out_ld], 128 channels
    int out_ld;
    int H, W, tiles_x, tiles_y;
    const half_t *w_m1, *w_m2;   // LDS-family nt = 1 packing [4 tiles][2 chunks][9 taps][64 lanes][8]
    const float *b_m1, *b_m2;
    const half_t *w_cv2;         // direct-family packing [8 tiles][k-steps][64 lanes][8]
    const float *b_cv2;
};
size_t bneck64_lds_bytes(int mode, int ks2);
bool launch_bneck64(int mode, int ks2, bool shortcut, const BneckArgs &a, int batch, hipStream_t s);   // mode 0: A (-> concat slice), 1: B (+ cv2, ks2 = 6 / 8)

// the keypoint branch of one Detect level (3x3 Cin -> 16, 3x3 16 -> 16, 1x1 16 -> nk) in one launch (k_kpt.hip)
constexpr int kKpt3Tile = 10;
struct Kpt3Args {
    const half_t *x;      // the level's input [B][H][W][x_ld], offset to the first of its Cin channels
    int x_ld;
    int H, W, tiles_x, tiles_y;
    const half_t *w1;     // first conv: LDS-family nt = 1 packing [chunk of 32][tap][64 lanes][8]
    const float *b1;
    const half_t *w2;     // second conv: Cin = 16 direct packing [5 k-steps][64 lanes][8]
    const float *b2;
    const half_t *w3;     // final 1x1: A layout of v_mfma_f32_16x16x16_f16 [64 lanes][4]
    const float *b3;      // (not scaled: the final has no activation)
    float *out;           // head records [B][H * W][out_ld], offset to the keypoint channels
    int out_ld;
};
bool kpt3_eligible(int cin);
bool launch_kpt3(const Kpt3Args &a, int cin, int batch, hipStream_t s);

struct ConvArgs {
    ConvSeg s0, s1;
    int Hin, Win;       // input size at the conv's own resolution
    int Hout, Wout;
    int M;              // batch * Hout * Wout
    int Cin;            // s0.C + s1.C
    const half_t *w;    // packed MFMA A fragments [ntile][kstep][64 lanes][8]
    const float *bias;  // [cout_pad]
    void *out;          // half_t* or float*, already offset to the first output channel
    int out_ld;
    const half_t *res;  // optional residual (same pixel indexing as out), nullable
    int res_ld;
    int cout_pad;       // multiple of 16
    int ksteps;
    int pair;           // weight rows packed with the paired-tile channel permutation
    // optional trailing 1x1 conv fused into the epilogue of the LDS 3x3 kernel (Detect-head finals: bias only, fp32
    // out): this conv's activated fp16 output never leaves the registers.  n2 = 16-row tiles of the 1x1 (0 = none).
    // (Cin = 16 direct kernel, keypoint branch: w2 = the 16 -> nk final in the 16x16x16 MFMA's A layout [64 lanes][4], n2 = 1.)
    const half_t *w2;   // direct-family packing of the 1x1 [tile][k-step][64][8], Cin = this conv's 64 channels
    const float *bias2;
    float *out2;
    int out2_ld, n2;
    // Detect class branch only (the fused 1x1 IS the class-logit conv): candidates are thresholded where the logits are
    // computed and appended to the frame's key list -- the separate scan over the head tensor then never runs.
    unsigned long long *scan_keys;   // [B][scan_key_cap], nullptr = off
    int *scan_counts;                // [B]
    float scan_thr;
    int scan_nc, scan_abase, scan_key_cap;   // classes; first anchor index of this level; list capacity (= A * nc)
};

// several independent layers in one launch (k_conv.hip conv3x3_lds_multi / conv_mfma_multi)
constexpr int kMultiMax = 6;
struct LdsMember {
    ConvArgs a;
    const half_t *wl;
    int tiles_x, tiles_y, twc_log2, patch_bytes, batch, tile2d;
    int gx, gy;                 // this member's grid
};
struct LdsMultiArgs {
    int n;
    int start[kMultiMax + 1];   // first workgroup of member k
    LdsMember m[kMultiMax];
};
struct DirectMultiArgs {
    int n;
    int start[kMultiMax + 1], gx[kMultiMax], gy[kMultiMax];
    ConvArgs a[kMultiMax];
};

struct ConvCfg {
    int ks, stride, mt, nt; bool cin16; int act; bool out_f32; bool lds; int ipw;   // ipw: images per workgroup (LDS family)
    bool deep;   // direct kernel, latency variant: prefetch ring of 6..12 k-steps (single-frame steps)
    bool ct;     // direct kernel walking K chunk-major with the LDS family's weights: bit-identical stand-in for that family
    bool pw;     // 1x1 layers: the persistent pointwise kernel (weights in LDS, pixel tiles software-pipelined)
    bool pf2;    // LDS family, mt = 1: global -> register staging two (image, chunk) steps ahead instead of one
    bool pf4;    // LDS family, mt = 1, nt = 1: four steps ahead (a lone frame's 128-channel layers: all of a workgroup's steps in flight at once)
    bool w8;     // LDS family, stride 2: one 8-wave workgroup on a block twice as tall (weights staged for twice the pixels, mt = 2 fits LDS)
    int cm;      // LDS family: chunk-major order over the workgroup's cm = ipw images (0: image-major), weights staged once per chunk
    bool wr;     // LDS family, Cin = Cout = 64, stride 1: the layer's weights stay resident in the LDS of one 8-wave workgroup that walks ipw images
    bool pp;     // ... as two ping-pong groups of four waves (one in its MFMA phase while the other stores, stages and loads)
};
// returns false if no instantiation exists for cfg
bool launch_conv(const ConvCfg &cfg, const ConvArgs &a, hipStream_t s);
const char *conv_cfg_name(const ConvCfg &cfg, char *buf, int n);
bool conv_pw_eligible(const ConvCfg &cfg, const ConvArgs &a);
// n independent layers (n <= kMultiMax) as ONE launch with a common tile: LDS family, stride 1, mt = 1, nt in {1, 2, 4}
// (fused 1x1s allowed at nt = 4, per member); direct family: the two shapes of the keypoint branch (3x3 Cin = 16 SiLU,
// 1x1 fp32 bias-only; mt = nt = 1).  false: some member has no such tile.
bool launch_conv_lds_multi(int nt, const ConvArgs *a, const half_t *const *wl, int n, int batch, hipStream_t s);
bool launch_conv_direct_multi(const ConvCfg &cfg, const ConvArgs *a, int n, hipStream_t s);
bool launch_conv_k16(const ConvArgs &a, hipStream_t s);   // 1x1, Cin = 16 -> <= 16 channels, bias only, fp32 out: a.w2 = weights in the 16x16x16 MFMA's A layout
bool launch_conv_pw(const ConvCfg &cfg, const ConvArgs &a, hipStream_t s);   // cfg.ipw = NBW: 64-channel output blocks per workgroup (1: the single-block kernel)
size_t conv_pw_lds_bytes(const ConvArgs &a, int nbw);                        // NBW = 2 / 4: LDS of the multi-block form, 0 = not offered
// LDS-staged 3x3 family (k_conv.hip): wl = weights packed [n-block][chunk 32][tap][tile][lane][8] for this nt
size_t conv_lds_bytes(const ConvArgs &a, int stride, int mt, int nt, int *patch_rows_max, bool w8 = false);
// weights-resident variant: bytes of LDS (0 = not eligible), tile positions per image and workgroup column, launcher
size_t conv_wres_bytes(const ConvArgs &a, int stride, bool pp);
int conv_wres_tiles(const ConvArgs &a, int stride, bool pp);
bool launch_conv_wres(int ipw, const ConvArgs &a, const half_t *wl, int batch, hipStream_t s, bool pp);
bool launch_conv_lds(int stride, int mt, int nt, int ipw, const ConvArgs &a, const half_t *wl, int batch, hipStream_t s, int pf2 = 0, int cm = 0, bool w8 = false);   // a.n2 > 0: fused 1x1 (needs stride 1, nt 4, cout 64); pf2: staging two steps ahead (mt = 1)

// SPPF pooling chain: slice 0 (C ch) of [B][H][W][4C] -> slices 1..3 (5x5, 9x9, 13x13 max)
void launch_sppf_pool(half_t *buf, int batch, int H, int W, int C, hipStream_t s);

// ---- post-processing -----------------------------------------------------------
struct DevDet {           // device/pinned result record
    float box_net[4];     // xyxy, net-input pixels (EfficientNMS det_boxes)
    float xyxy[4];        // source-frame pixels (parse_output)
    float score;
    int32_t cls, anchor, pnp_ok;
    float kpts_net[8];
    float kpts[8];
    double rvec[3], tvec[3], quat[4];
    int32_t armor_valid;  // 1: kpts hold an armor's four points; 0: none; -1: ROI larger than the label scratch
    int32_t armor_size;   // 0 small / 1 large (classical path: from the light-centre distance)
    int32_t n_lights;     // classical path: lights that passed the gates in this ROI
    int32_t pad_;
};
struct DevFrameOut { int32_t num_dets, n_candidates, overflow, pad; };

struct PnpConst {
    double fx, fy, cx, cy, k1, k2, p1, p2, k3;
    double hy[2], hz[2];   // half extents of the small / large armor (metres)
};

struct PostArgs {
    const float *head_all;    // one allocation: [level][slot (all num_slots)][H*W][kHeadRec]
    int slots_total, first;   // level block offset = lvl_base * slots_total records; this step starts at slot `first`
    float *boxes;             // [B][A][4]
    unsigned long long *keys; // [B][key_cap]: every (anchor, class) pair above threshold, unsorted (read back only when a frame
                              // has more candidates than the LDS list holds)
    int key_cap;              // = A * nc: the list can never overflow
    DevDet *dets;             // [B][max_det]
    DevFrameOut *fout;        // [B]
    int net, A, nc, nk;
    float logit_thr, iou_thr;
    int max_det, pre_nms_cap;
    // parse_output mapping net -> source frame: x_src = (x - off_x) * scale_x
    float scale_x, scale_y, off_x, off_y;
    int armor_size;
    const PnpConst *pnp;      // device copy (keeps the kernel-argument struct out of scratch)
    long long *dbg;           // optional [B][8] phase stamps of nms_pnp_kernel (diagnostic builds of the engine only)
    int keys_only;            // 1: the key list comes from the class-branch conv epilogues (ConvArgs::scan_keys): nms_pnp_kernel decodes
                              // the candidates' boxes itself
    int prefilter;            // 1: frames with more than 512 candidates first walk their best 320 .. 512 only (exact: nms_pnp_kernel, "ATTEMPT 0")
    int classwalk;            // 1: frames of up to 512 candidates take nms_pnp_kernel's class-major path (round 4); 0: the round-3 matrix walk (bit-identical)
    int *counts;              // [B] candidates appended by scan_decode_kernel; nullptr: nms_pnp_kernel scans the head itself.
                              // Zero at engine creation; nms_pnp_kernel reads its frame's count and resets it (no memset node, nothing
                              // for another step to find non-zero); every reader clamps it to key_cap
};
void launch_nms_pnp(const PostArgs &a, int batch, hipStream_t s);
constexpr int kScanBlocks = 16;   // workgroups per frame of scan_decode_kernel
void launch_scan_decode(const PostArgs &a, int batch, hipStream_t s);
// ---- classical light extraction (k_light.hip) -------------
constexpr int kLightLdsPoints = 256;         // contours up to this many points are sorted / hulled in LDS
constexpr int kLightLdsImage = 40 * 1024;   // padded label images up to this many bytes live in LDS
constexpr int kLightMaxContours = 1024;   // contours per ROI; more -> armor_valid = -1 (no answer), never a truncated one

struct LightArgs {
    const uint8_t *frames;   // [B][rows][cols][3] device frames (un-rotated; rotation folded into the fetch)
    size_t frame_bytes;
    int cols, rows, rotate180;
    DevDet *dets;            // [B][max_det]: xyxy in (bbox source when boxes == nullptr), armor fields out
    int max_det;
    const int *num_dets;     // per frame, stride num_dets_stride ints; nullptr -> n_boxes for every frame
    int num_dets_stride, n_boxes;
    const float *boxes;      // optional explicit boxes [n_boxes][4] (irmv_engine_extract_armors)
    signed char *labels;     // [B][label_pool] per-frame pool the padded label images are carved from, in detection order
    size_t label_pool;       // bytes per frame (>= one full-frame ROI)
    short *points;           // [B][max_det][points_cap][2] contour points
    int points_cap;
    short *hulls;            // [B][max_det][points_cap * 2][2] hull scratch
    int binary_threshold;
    float light_min_ratio, light_max_ratio, light_max_angle;
    double min_small_cd, max_small_cd, min_large_cd, max_large_cd;
    const PnpConst *pnp;
    int pnp_armor_size;
};
void launch_light_extract(const LightArgs &a, int n_boxes_max, int batch, hipStream_t s);

void launch_pnp_only(const PnpConst &c, const float *pts, int n, int armor_size, double *rvec, double *tvec,
                     int32_t *ok, hipStream_t s);

}  // namespace irmv
