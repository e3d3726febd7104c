// What a CU can pull into LDS, and whether LDS-DMA changes it (round 5, VERDICT r4 item 3: "stage the stride-2 family through
// LDS-DMA, not VGPRs").  One 8-wave workgroup per CU (the stride-2 `w8` kernels' shape) stages `rounds` slabs of `slab` KiB
// into LDS, nothing else -- the staging floor of those kernels (model.5 stages 106 KB per chunk step for 2.3 k cycles
// of MFMAs and measures ~12 B/clk per CU).  Arms:
//   reg   global_load_dwordx4 -> VGPR -> ds_write_b128, all of a slab's loads issued before its first store (the kernels' scheme)
//   dma   __builtin_amdgcn_global_load_lds 16 B: no VGPRs, no ds_write; waited with vmcnt(0) per slab
//   dma2  ... two slabs in flight (the next slab requested before the current one is waited for)
// Sources: `shared` = every workgroup reads the SAME 128 KiB (weights: L2-resident after the first touch) or its OWN stream
// (activations: from the Infinity Cache / HBM).
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/stage_rate_probe.cpp -o build_probe/stage_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PIECES>   // PIECES = 16-byte pieces per thread and slab (slab = PIECES * 512 * 16 bytes)
__global__ __launch_bounds__(512) void stage_kernel(const u32x4 *src, size_t wg_stride16, int rounds, unsigned int *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *lds = reinterpret_cast<u32x4 *>(smem);
    const int tid = threadIdx.x, wave = tid >> 6;
    const u32x4 *base = src + (size_t)blockIdx.x * wg_stride16;
    constexpr int SLAB16 = PIECES * 512;
    unsigned int acc = 0;
    if constexpr (MODE == 0) {
        for (int r = 0; r < rounds; r++) {
            u32x4 v[PIECES];
#pragma unroll
            for (int i = 0; i < PIECES; i++) v[i] = base[(size_t)r * SLAB16 + i * 512 + tid];
#pragma unroll
            for (int i = 0; i < PIECES; i++) lds[(r & 1) * SLAB16 + i * 512 + tid] = v[i];
            __syncthreads();
            acc += lds[(r & 1) * SLAB16 + ((tid * 7) & (SLAB16 - 1))][0];   // (somebody reads the slab)
        }
    } else {
        auto issue = [&](int r) {
#pragma unroll
            for (int i = 0; i < PIECES; i++)   // one wave-instruction = 1 KiB contiguous in LDS (wave-uniform base + lane * 16)
                __builtin_amdgcn_global_load_lds(base + (size_t)r * SLAB16 + i * 512 + tid,
                                                 (__attribute__((address_space(3))) void *)(unsigned int)(unsigned long long)(lds + (r % 3) * SLAB16 + i * 512 + wave * 64), 16, 0, 0);
        };
        issue(0);
        if (MODE == 2 && rounds > 1) issue(1);
        for (int r = 0; r < rounds; r++) {
            if (MODE == 2) {
                if (r + 2 < rounds) { issue(r + 2); __builtin_amdgcn_s_waitcnt(0x0f70 | (2 * PIECES > 63 ? 63 : 2 * PIECES)); }   // leave two slabs in flight ... (vmcnt field: low 4 + high 2 bits; PIECES <= 7)
                else if (r + 1 < rounds) __builtin_amdgcn_s_waitcnt(0x0f70 | PIECES);
                else __builtin_amdgcn_s_waitcnt(0x0f70);
            } else {
                __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
            }
            __builtin_amdgcn_s_barrier();
            acc += lds[(r % 3) * SLAB16 + ((tid * 7) & (SLAB16 - 1))][0];
            if (MODE == 1 && r + 1 < rounds) { __builtin_amdgcn_s_barrier(); issue(r + 1); }
            else if (MODE == 2) __builtin_amdgcn_s_barrier();
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, rounds = argc > 2 ? atoi(argv[2]) : 64;
    constexpr int PIECES = 4;                     // 32 KiB slabs
    const size_t slab = (size_t)PIECES * 512 * 16, per_wg = slab * rounds;
    u32x4 *d; unsigned int *d_out;
    CK(hipMalloc(&d, per_wg * wgs)); CK(hipMalloc(&d_out, 4096 * 4));
    CK(hipMemset(d, 1, per_wg * wgs));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *mn[3] = {"reg  (loads -> VGPR -> ds_write_b128)", "dma  (global_load_lds, one slab)", "dma2 (global_load_lds, two slabs in flight)"};
    for (int shared = 1; shared >= 0; shared--)
        for (int mode = 0; mode < 3; mode++) {
            auto launch = [&]() {
                const size_t stride16 = shared ? 0 : per_wg / 16;
                const size_t lds = slab * 3;
                if (mode == 0) hipLaunchKernelGGL((stage_kernel<0, PIECES>), dim3(wgs), dim3(512), lds, 0, d, stride16, rounds, d_out);
                if (mode == 1) hipLaunchKernelGGL((stage_kernel<1, PIECES>), dim3(wgs), dim3(512), lds, 0, d, stride16, rounds, d_out);
                if (mode == 2) hipLaunchKernelGGL((stage_kernel<2, PIECES>), dim3(wgs), dim3(512), lds, 0, d, stride16, rounds, d_out);
            };
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stage_kernel<0, PIECES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stage_kernel<1, PIECES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stage_kernel<2, PIECES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            for (int i = 0; i < 3; i++) launch();
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 10; i++) launch();
                CK(hipEventRecord(e1, 0));
                CK(hipDeviceSynchronize());
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double us = best * 1e3 / 10, gbs_cu = (double)per_wg / (us * 1e-6) / 1e9;
            printf("%3d workgroups, %-14s %-44s %8.1f us  %6.1f GB/s per CU (%4.1f B/clk at 2.1 GHz)  chip %6.2f TB/s\n", wgs, shared ? "shared 2 MiB" : "own stream", mn[mode], us, gbs_cu, gbs_cu / 2.1, gbs_cu * wgs / 1e3);
        }
    return 0;
}
