// Can a kernel pull a camera frame out of pinned host memory as fast as the copy engine moves it?  (Round 5, single-frame
// latency: hipMemcpyAsync of one 3.93 MB frame takes 77 - 83 us = 47 - 51 GB/s, the front kernel behind it 7 us; a front
// kernel that read the pinned slot itself would save the copy's fixed cost and its own time -- if kernel reads over PCIe
// reach the link rate.)  Reads `bytes` of pinned host memory with 16-byte (and 12-byte, the front kernel's direct tiles')
// loads per lane, coherent and non-coherent allocations, against the DMA copy of the same buffer.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/hostread_probe.cpp -o build_probe/hostread_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read16(const u32x4 *src, size_t n16, unsigned int *out)
{
    unsigned int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = src[i];
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;   // (never: keeps the loads)
}
struct __attribute__((packed)) u96 { unsigned int a, b, c; };
__global__ __launch_bounds__(256) void read12(const unsigned int *src, size_t n12, unsigned int *out)
{
    unsigned int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n12; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned int *p = src + 3 * i;
        acc += p[0] ^ p[1] ^ p[2];
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) : 3932160;
    unsigned int *d_out; void *d_dst;
    CK(hipMalloc(&d_out, 1 << 20)); CK(hipMalloc(&d_dst, bytes));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned flags[2] = {hipHostMallocDefault, hipHostMallocNonCoherent};
    const char *fn[2] = {"coherent (default)", "non-coherent"};
    for (int f = 0; f < 2; f++) {
        void *h; CK(hipHostMalloc(&h, bytes, flags[f] | hipHostMallocMapped));
        for (size_t i = 0; i < bytes / 4; i++) ((unsigned int *)h)[i] = (unsigned int)(i * 2654435761u);
        void *hd; CK(hipHostGetDevicePointer(&hd, h, 0));
        auto time_it = [&](auto launch, const char *name) -> int {
            for (int i = 0; i < 3; i++) launch();
            CK(hipStreamSynchronize(st));
            float best = 1e30f;
            for (int rep = 0; rep < 7; rep++) {
                CK(hipEventRecord(e0, st));
                launch();
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            printf("%-20s %-34s %8.1f us  %6.1f GB/s\n", fn[f], name, best * 1e3, bytes / (best * 1e-3) / 1e9);
            return 0;
        };
        if (time_it([&] { (void)hipMemcpyAsync(d_dst, h, bytes, hipMemcpyHostToDevice, st); }, "hipMemcpyAsync H2D")) return 1;
        for (int grid : {256, 1024, 4096}) {
            char nm[64];
            snprintf(nm, sizeof nm, "kernel 16 B/lane, %d blocks", grid);
            if (time_it([&] { hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, st, (const u32x4 *)hd, bytes / 16, d_out); }, nm)) return 1;
        }
        if (time_it([&] { hipLaunchKernelGGL(read12, dim3(1024), dim3(256), 0, st, (const unsigned int *)hd, bytes / 12, d_out); }, "kernel 12 B/lane, 1024 blocks")) return 1;
        CK(hipHostFree(h));
    }
    return 0;
}
