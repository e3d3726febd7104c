// Issue cost of the vector instructions SiLU is made of, on gfx950: one wave per SIMD runs a long unrolled run of
// independent v_mul_f32 / v_exp_f32 / v_rcp_f32 / v_cvt_f16_f32 and reports shader clocks per instruction.
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
    }
    return 0;
}
