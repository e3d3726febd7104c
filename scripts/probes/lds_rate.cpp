// What a CU's LDS delivers to ds_read_b128 (the fragment reads of the LDS conv family), by access pattern and waves per CU.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/probes/lds_rate.cpp -o build_probe/lds_rate && ./build_probe/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int STRIDE_B>   // byte stride between consecutive lanes' 16-byte reads (96: the conv family's pixel stride; 16: dense)
__global__ __launch_bounds__(1024) void lds_kernel(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // lane (g, r): pixel r, 16-byte quarter g -- the B-fragment pattern; different waves start at different pixels
    const int off = (((lane & 15) + (threadIdx.x >> 6) * 16) * STRIDE_B + (lane >> 4) * 16) & 32767 & ~15;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(smem + ((off + u * 1536) & 32767));
            asm volatile("" : "+v"(acc) : "v"(v));   // keep the read, no arithmetic
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
}

template <int STRIDE_B> static void run(const char *name, int waves)
{
    const int blocks = 256, iters = 4000;
    float *out; hipMalloc(&out, blocks * 1024 * sizeof(float));
    hipFuncSetAttribute(reinterpret_cast<const void *>(lds_kernel<STRIDE_B>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = 100 * 1024;   // one workgroup per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    lds_kernel<STRIDE_B><<<blocks, 64 * waves, lds>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    lds_kernel<STRIDE_B><<<blocks, 64 * waves, lds>>>(out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double reads = (double)iters * 16 * waves;           // wave-level ds_read_b128 per CU
    const double ns_per = ms * 1e6 / reads;
    printf("%-34s %2d waves/CU: %.2f ns per wave-read = %.0f B/ns per CU (x 2.1 GHz clock: %.0f B/clk)\n", name, waves, ns_per, 1024.0 / ns_per, 1024.0 / ns_per / 2.1);
    hipFree(out);
}

int main()
{
    for (int w : {4, 8, 16}) {
        run<16>("dense (16 B between lanes)", w);
        run<96>("96 B pixel stride (conv family)", w);
        run<80>("80 B pixel stride (stride-2 layers)", w);
        run<64>("64 B pixel stride", w);
        run<32>("32 B pixel stride (model.2 planes)", w);
        run<160>("160 B (stride-2 reads, 80 B pixels)", w);
        run<192>("192 B (stride-2 reads, 96 B pixels)", w);
        run<224>("224 B (stride-2 reads, 112 B pixels)", w);
        run<48>("48 B pixel stride", w);
        run<112>("112 B pixel stride", w);
    }
    return 0;
}
