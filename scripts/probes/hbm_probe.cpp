// What a plain streaming kernel reaches on this box, by transfer size: the practical ceiling for the short HBM-bound
// kernels of a step (a 64-frame 1x1 conv moves 55 MB in 17 us).  read-only, write-only and copy (read + write) kernels,
// 16 bytes per lane, grid-stride, grid = 256 CUs x 8 workgroups of 256.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/hbm_probe.cpp -o hbm_probe ; ./hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) b[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
    const uint4 v = {(unsigned)n, 1, 2, 3};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = v;
}

int main()
{
    const size_t maxb = 512ull << 20;
    uint4 *a, *b;
    CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb));
    CK(hipMemset(a, 1, maxb)); CK(hipMemset(b, 2, maxb));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    printf("%10s %14s %14s %14s   (TB/s of bytes moved; one launch after another on one stream, 40 launches)\n", "MB moved", "read", "write", "copy");
    for (size_t mb : {8, 16, 32, 55, 64, 128, 256, 512}) {
        double res[3];
        for (int mode = 0; mode < 3; mode++) {
            const size_t bytes = mb << 20, n = (mode == 2 ? bytes / 2 : bytes) / 16;   // copy: half read, half written
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, st, a, b, n);
                else if (mode == 1) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, st, a, b, n);
                else hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st, a, b, n);
            };
            for (int i = 0; i < 10; i++) launch();
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 40; i++) launch();
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            res[mode] = (double)bytes * 40 / (ms * 1e-3) * 1e-12;
        }
        printf("%10zu %14.2f %14.2f %14.2f\n", mb, res[0], res[1], res[2]);
    }
    return 0;
}
