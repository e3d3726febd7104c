// Frame preprocess and the stem conv for gfx950.
//
// Replaces the reference's four NPP launches (src/yolo_engine.cpp:179-200:
// nppiMirror -> nppiResize -> nppiScale -> nppiCopy packed->planar, ~29 MB of
// device traffic per frame) with ONE pass: every source row is read once per use
// with 16-byte coalesced loads into LDS, the 180-degree rotation is folded into
// the tap indices, and the result is written as fp16 NHWC4 (8 bytes per pixel,
// the layout model.0.conv consumes).  HBM-bound: 3.93 MB read + 3.28 MB written
// per 1280x1024 frame.
#include "irmv_common.hpp"

namespace irmv {

constexpr int kCoefBits = 11;
constexpr int kCoefOne = 1 << kCoefBits;
constexpr int kMaxRowBytes = 4096 * 3;

__global__ __launch_bounds__(256) void preprocess_kernel(PreArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t rows[2][kMaxRowBytes];
    __shared__ half_t lut[256];   // (half)(q / 255.0f)
    lut[threadIdx.x] = (half_t)((float)threadIdx.x / 255.0f);
    const int dy = blockIdx.x, b = blockIdx.y;
    const AxisTap ty = a.ty[dy];
    const int row_bytes = a.sw * 3;
    const uint8_t *src = a.src + (size_t)b * a.src_slot_bytes;
    half_t *dst = a.dst + ((size_t)b * a.net + dy) * a.net * 4;
    const half_t padv = (half_t)(114.0f / 255.0f);

    if (ty.i0 >= 0) {
        const uint8_t *r0 = src + (size_t)ty.i0 * row_bytes;
        const uint8_t *r1 = src + (size_t)ty.i1 * row_bytes;
        if ((row_bytes & 15) == 0) {
            const int nvec = row_bytes >> 4;
            for (int i = threadIdx.x; i < nvec; i += blockDim.x) {
                reinterpret_cast<uint4 *>(rows[0])[i] = reinterpret_cast<const uint4 *>(r0)[i];
                reinterpret_cast<uint4 *>(rows[1])[i] = reinterpret_cast<const uint4 *>(r1)[i];
            }
        } else {
            for (int i = threadIdx.x; i < row_bytes; i += blockDim.x) {
                rows[0][i] = r0[i];
                rows[1][i] = r1[i];
            }
        }
    }
    __syncthreads();

    const uint32_t wy = (uint32_t)ty.w1;
    for (int dx = threadIdx.x; dx < a.net; dx += blockDim.x) {
        const AxisTap tx = a.tx[dx];
        half4 o;
        if (ty.i0 < 0 || tx.i0 < 0) {
            o = (half4){padv, padv, padv, (half_t)0.0f};
        } else {
            const uint32_t wx = (uint32_t)tx.w1;
            half_t v[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const uint32_t p00 = rows[0][tx.i0 * 3 + c], p01 = rows[0][tx.i1 * 3 + c];
                const uint32_t p10 = rows[1][tx.i0 * 3 + c], p11 = rows[1][tx.i1 * 3 + c];
                // 24-bit multiplies (full rate): coefficients <= 2^11, pixels < 2^8, top / bot < 2^19
                const uint32_t top = __umul24(kCoefOne - wx, p00) + __umul24(wx, p01);
                const uint32_t bot = __umul24(kCoefOne - wx, p10) + __umul24(wx, p11);
                const uint32_t acc = __umul24(kCoefOne - wy, top) + __umul24(wy, bot);
                v[c] = lut[(acc + (1u << (2 * kCoefBits - 1))) >> (2 * kCoefBits)];
            }
            if (a.swap_rb) { const half_t t = v[0]; v[0] = v[2]; v[2] = t; }
            o = (half4){v[0], v[1], v[2], (half_t)0.0f};
        }
        *reinterpret_cast<half4 *>(dst + (size_t)dx * 4) = o;
    }
}

void launch_preprocess(const PreArgs &a, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(preprocess_kernel, dim3(a.net, batch), dim3(256), 0, s, a);
}

// dst(x, y) = src(sw-1-x, sh-1-y): the frame get_rotated_image() exposes.
__global__ __launch_bounds__(256) void rotate180_kernel(const uint8_t *src, uint8_t *dst, int sw, int sh)
{
    const size_t npx = (size_t)sw * sh;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (size_t)gridDim.x * blockDim.x) {
        const size_t q = npx - 1 - p;
        dst[p * 3 + 0] = src[q * 3 + 0];
        dst[p * 3 + 1] = src[q * 3 + 1];
        dst[p * 3 + 2] = src[q * 3 + 2];
    }
}

// A frame's way from its pinned host slot into HBM as a KERNEL (one 16-byte load per lane straight out of the mapped host
// memory, 256 workgroups): for ONE 3.93 MB frame the copy engine's hipMemcpyAsync costs 77 - 103 us by box (38 - 51 GB/s: a fixed
// start-up of 10 - 30 us on top of the link time), the kernel 11 us less (scripts/probes/hostread_probe.cpp: 102.9 -> 91.5 us;
// from 60 MB up both sit at the link's 57 GB/s).  Used where the upload is the first node of a synchronous single-frame step
// (the reference's detect(), src/yolo_engine.cpp:153-177); batched and pipelined uploads stay on the copy engine, which
// runs beside the compute kernels.
__global__ __launch_bounds__(256) void upload_frame_kernel(const u32x4_t *src, u32x4_t *dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

void launch_upload_frames(const uint8_t *src_host_mapped, uint8_t *dst, size_t bytes, int blocks, hipStream_t s)
{
    hipLaunchKernelGGL(upload_frame_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const u32x4_t *>(src_host_mapped), reinterpret_cast<u32x4_t *>(dst), bytes / 16);
}

void launch_rotate180(const uint8_t *src, uint8_t *dst, int sw, int sh, hipStream_t s)
{
    hipLaunchKernelGGL(rotate180_kernel, dim3(2048), dim3(256), 0, s, src, dst, sw, sh);
}

// model.0.conv (3x3 stride 2, 3 -> 16, SiLU) on the matrix cores.
// The input is NHWC4 (rgb0), so with K laid out as [kernel row kh][4 tap slots][4
// channels] (16 k per row, the 4th slot and the 4th channel carry zero weights) the
// 8 k-values of an MFMA lane are two horizontally adjacent input pixels = 16
// contiguous bytes.  K = 48 -> two v_mfma_f32_16x16x32_f16 per 16 pixels x 16
// channels; the kernel is bound by its 6.6 MB/frame of HBM traffic, not by math.
__global__ __launch_bounds__(256) void conv0_kernel(Conv0Args a)
{
    constexpr int MT = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int Ho = a.net >> 1, HWo = Ho * Ho;
    const int M = a.batch * HWo;
    const int tile0 = (blockIdx.x * 4 + wave) * MT;
    const half8 *wp = reinterpret_cast<const half8 *>(a.w) + lane;
    const half8 w0 = wp[0], w1 = wp[64];
    const half4 z4 = (half4){0, 0, 0, 0};
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int m = (tile0 + mt) * 16 + r;
        const bool mv = m < M;
        const int mm = mv ? m : 0;
        const int b = mm / HWo, rem = mm - b * HWo, oy = rem / Ho, ox = rem - oy * Ho;
        const half_t *xb = a.x + (size_t)b * a.net * a.net * 4;
        const int ix0 = ox * 2 - 1 + 2 * (g & 1);       // first pixel of this lane's tap pair
        half8 bf[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int kh = 2 * s + (g >> 1);
            const int iy = oy * 2 - 1 + kh;
            half4 lo = z4, hi = z4;
            if (mv && kh < 3 && (unsigned)iy < (unsigned)a.net) {
                const half_t *row = xb + (size_t)iy * a.net * 4;
                if ((unsigned)ix0 < (unsigned)a.net) lo = *reinterpret_cast<const half4 *>(row + (size_t)ix0 * 4);
                if ((g & 1) == 0) hi = *reinterpret_cast<const half4 *>(row + (size_t)(ix0 + 1) * 4);   // kw = 1; slot 3 is padding
            }
            bf[s] = (half8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, bf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, bf[1], acc, 0, 0, 0);
        if (mv) {
            // the image is unscaled: one fma brings the accumulator to the activation scale (a.b = log2 e * bias; irmv_common.hpp)
            const half4 o = silu_pack4(__builtin_fmaf(acc[0], kActScale, a.b[g * 4 + 0]), __builtin_fmaf(acc[1], kActScale, a.b[g * 4 + 1]),
                                       __builtin_fmaf(acc[2], kActScale, a.b[g * 4 + 2]), __builtin_fmaf(acc[3], kActScale, a.b[g * 4 + 3]));
            *reinterpret_cast<half4 *>(a.y + (size_t)mm * 16 + g * 4) = o;
        }
    }
}

void launch_conv0(const Conv0Args &a, hipStream_t s)
{
    const int Ho = a.net >> 1;
    const int tiles = (a.batch * Ho * Ho + 15) / 16;
    hipLaunchKernelGGL(conv0_kernel, dim3((tiles + 15) / 16), dim3(256), 0, s, a);
}

}  // namespace irmv
