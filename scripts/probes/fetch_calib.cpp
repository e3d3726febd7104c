// What does rocprofv3's FETCH_SIZE report for the front kernel's source access pattern?  (MI355X_MICROARCH.md: on gfx950
// FETCH_SIZE reads exactly 1/2 of the bytes of a wide -- 16 B per lane -- coalesced streaming read; other widths are
// uncalibrated.)  front_kernel stages its source region as 12-byte groups, three dwords per lane at a 12-byte lane stride.
// Three kernels read the SAME 480 MiB buffer exactly once: 16 B per lane, 12 B per lane as one dwordx3, 12 B per lane as
// three dword loads; run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and compare the counter with the known bytes.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/fetch_calib.cpp -o build_probe/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
__global__ void read16(const u32x4 *p, size_t n, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const u32x4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void read12x3(const unsigned *p, size_t n, unsigned *sink)   // one 12-byte access per lane
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const u32x3 v = *reinterpret_cast<const u32x3 *>(p + 3 * i); acc ^= v.x ^ v.y ^ v.z; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void read12as3(const unsigned *p, size_t n, unsigned *sink)  // three dword loads per lane (front_kernel's q[0], q[1], q[2])
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const volatile unsigned *q = p + 3 * i;
        acc ^= q[0]; acc ^= q[1]; acc ^= q[2];
    }
    if (acc == 0x12345678u) *sink = acc;
}
int main()
{
    const size_t bytes = 480ull << 20;
    unsigned *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const u32x4 *>(buf), bytes / 16, sink);
        hipLaunchKernelGGL(read12x3, dim3(4096), dim3(256), 0, 0, buf, bytes / 12, sink);
        hipLaunchKernelGGL(read12as3, dim3(4096), dim3(256), 0, 0, buf, bytes / 12, sink);
    }
    hipDeviceSynchronize();
    printf("each kernel read %zu bytes (%.1f KiB) once\n", bytes, bytes / 1024.0);
    return 0;
}
