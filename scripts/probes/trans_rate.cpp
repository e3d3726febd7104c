// Issue cost of the vector instructions SiLU is made of, on gfx950: one wave per SIMD runs a long unrolled run of
// independent v_mul_f32 / v_exp_f32 / v_rcp_f32 / v_cvt_f16_f32 / integer multiplies and reports shader clocks per instruction.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/probes/trans_rate.cpp -o build_probe/trans_rate && ./build_probe/trans_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float *out, unsigned long long *clk, int iters)
{
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
    const unsigned long long msk = (iters & 1) ? 0x5555555555555555ull : 0xaaaaaaaaaaaaaaaaull;   // a lane mask in an SGPR pair
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(1.0001f));
                if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
                if (OP == 3) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[i]));
                if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(1.0001f));
                if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(reinterpret_cast<double *>(v) + (i & 3))) : "v"(1.0));
                if (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(3));
                if (OP == 7) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(0x3e0f83e1));
                if (OP == 8) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(reinterpret_cast<double *>(v) + (i & 3))) : "v"(v[(i + 4) & 7]), "v"(17) : "vcc");
                if (OP == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[i]) : "v"(17));
                if (OP == 10) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(17));
                if (OP == 11) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(v[i]) : "v"(17));
                if (OP == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(17));
                if (OP == 13) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(17), "v"(0x0c020c00));
                if (OP == 22) { if (i == 0) asm volatile("s_mov_b64 vcc, %0" : : "s"(msk) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(17)); }
                if (OP == 23) { if (i == 0) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(v[7]), "v"(17) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(17)); }
                if (OP == 24) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(17)); asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(v[(i + 3) & 7]) : "v"(17)); }
                if (OP == 25) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(17), "s"(msk)); asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(v[(i + 3) & 7]) : "v"(17)); }
                if (OP == 14) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(17), "s"(msk));
                if (OP == 15) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(0x7fffffff));
                if (OP == 16) asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(v[i]) : "s"(msk));
                if (OP == 17) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(v[i]), "v"(17) : "vcc");
                if (OP == 18) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(17));
                if (OP == 19) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0" : "+v"(v[i]) : "v"(17));
                if (OP == 20) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(1.5f));
                if (OP == 21) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(reinterpret_cast<double *>(v) + (i & 3))) : "v"(1.0));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int OP> static void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 2000;   // 256-thread block = one wave per SIMD of a CU
    float *out; unsigned long long *clk;
    hipMalloc(&out, blocks * 256 * sizeof(float)); hipMalloc(&clk, blocks * sizeof(unsigned long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<OP><<<blocks, 256>>>(out, clk, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<OP><<<blocks, 256>>>(out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), clk, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += (double)c; avg /= blocks;
    const double n = (double)iters * 64;
    printf("%-14s %d wave(s)/SIMD: %.2f counter ticks per instruction per wave, %.3f ns per instruction per SIMD (event time)\n",
           name, waves_per_simd, avg / n, ms * 1e6 / (n * waves_per_simd));
    hipFree(out); hipFree(clk);
}

int main()
{
    for (int w = 1; w <= 2; w++) {
        run<0>("v_mul_f32", w); run<4>("v_fma_f32", w); run<5>("v_pk_fma_f32", w); run<1>("v_exp_f32", w); run<2>("v_rcp_f32", w); run<3>("v_cvt_f16_f32", w);
        run<6>("v_mul_lo_u32", w); run<7>("v_mul_hi_u32", w); run<8>("v_mad_u64_u32", w); run<9>("v_mad_u32_u24", w); run<10>("v_mul_u32_u24", w);
        run<11>("v_lshl_add_u32", w); run<12>("v_cndmask_b32", w); run<13>("v_perm_b32", w);
        run<14>("v_cndmask_e64 sgpr", w); run<16>("v_cndmask 0,v,sgpr", w); run<15>("v_and_b32", w); run<17>("v_cmp_gt_u32", w); run<18>("v_add_u32", w);
        run<22>("cndmask vcc<-s_mov", w); run<23>("cndmask vcc<-v_cmp", w);
        run<24>("cndmask vcc + 3 adds", w); run<25>("cndmask sgpr + 3 adds", w);
        run<19>("v_add_u32_sdwa", w); run<20>("v_cvt_pk_f16_f32", w); run<21>("v_pk_mul_f32", w);
    }
    return 0;
}
