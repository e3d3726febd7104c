// Platform probe (run on the GPU box): what the hand-off design may assume about this ROCm stack.
//   hipcc -O2 --offload-arch=gfx950 scripts/probes/stream_probe.cpp -o gpurun_out/stream_probe && gpurun_out/stream_probe
// Prints: pinned H2D bandwidth by copy size (alone / under a bandwidth-heavy kernel on another stream), cost of a
// cross-stream event hop, small D2H latency, kernel writes straight into pinned host memory.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
using clk = std::chrono::steady_clock;
static double us_since(clk::time_point t0) { return std::chrono::duration<double, std::micro>(clk::now() - t0).count(); }

__global__ void spin_kernel(float *p, int iters)
{
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; i++) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}
__global__ void stream_kernel(const float4 *a, float4 *b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void write_host_kernel(int *host, int v) { host[threadIdx.x] = v + threadIdx.x; }

int main()
{
    CK(hipSetDevice(0));
    const size_t big = 512ull << 20;
    uint8_t *h = nullptr, *d = nullptr;
    CK(hipHostMalloc((void **)&h, big, hipHostMallocDefault));
    std::memset(h, 1, big);
    CK(hipMalloc((void **)&d, big));
    float4 *a = nullptr, *b = nullptr;
    const size_t nvec = (1ull << 30) / 16;
    CK(hipMalloc((void **)&a, nvec * 16));
    CK(hipMalloc((void **)&b, nvec * 16));
    hipStream_t s1, s2, s3;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    // ---- 1. H2D bandwidth by size, alone ----
    for (size_t mb : {4, 16, 32, 64, 256, 512}) {
        const size_t n = mb << 20;
        CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        const int reps = mb <= 16 ? 40 : 10;
        auto t0 = clk::now();
        for (int i = 0; i < reps; i++) CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        const double us = us_since(t0) / reps;
        std::printf("h2d alone   %4zu MiB: %8.1f us  %6.2f GB/s\n", mb, us, n / us / 1e3);
    }
    // same, chunks issued back to back on one stream vs alternating two streams
    {
        const size_t n = 4 << 20;   // one frame-ish
        auto t0 = clk::now();
        for (int i = 0; i < 64; i++) CK(hipMemcpyAsync(d + i * n, h + i * n, n, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        double us = us_since(t0);
        std::printf("h2d 64 x 4 MiB one stream: %.1f us  %.2f GB/s\n", us, 64.0 * n / us / 1e3);
        t0 = clk::now();
        for (int i = 0; i < 64; i++) CK(hipMemcpyAsync(d + i * n, h + i * n, n, hipMemcpyHostToDevice, (i & 1) ? s2 : s1));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        us = us_since(t0);
        std::printf("h2d 64 x 4 MiB two streams: %.1f us  %.2f GB/s\n", us, 64.0 * n / us / 1e3);
    }
    // ---- 2. H2D under a bandwidth-heavy kernel on another stream ----
    {
        const size_t n = 256ull << 20;
        hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, s2, a, b, nvec);
        CK(hipStreamSynchronize(s2));
        auto t0 = clk::now();
        for (int i = 0; i < 40; i++) hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, s2, a, b, nvec);
        auto t1 = clk::now();
        for (int i = 0; i < 4; i++) CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        const double us = std::chrono::duration<double, std::micro>(clk::now() - t1).count() / 4;
        CK(hipStreamSynchronize(s2));
        std::printf("h2d under copy kernel 256 MiB: %.1f us  %.2f GB/s (kernels total %.0f us)\n", us, n / us / 1e3, us_since(t0));
    }
    // ---- 3. cross-stream event hop ----
    {
        float *p; CK(hipMalloc((void **)&p, 4096));
        hipEvent_t ev[64];
        for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        // baseline: 64 tiny kernels back to back on one stream
        for (int rep = 0; rep < 2; rep++) {
            auto t0 = clk::now();
            for (int i = 0; i < 64; i++) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, p, 10);
            CK(hipStreamSynchronize(s1));
            if (rep) std::printf("64 tiny kernels, one stream: %.1f us (%.2f us each)\n", us_since(t0), us_since(t0) / 64);
        }
        // ping-pong between two streams through events: 64 hops
        for (int rep = 0; rep < 2; rep++) {
            auto t0 = clk::now();
            for (int i = 0; i < 64; i++) {
                hipStream_t cur = (i & 1) ? s2 : s1, nxt = (i & 1) ? s1 : s2;
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, cur, p, 10);
                CK(hipEventRecord(ev[i], cur));
                CK(hipStreamWaitEvent(nxt, ev[i], 0));
            }
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            if (rep) std::printf("64 tiny kernels ping-ponging two streams via events: %.1f us (%.2f us per hop incl. kernel)\n", us_since(t0), us_since(t0) / 64);
        }
        // the same inside a captured graph (fork/join edges)
        {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < 64; i++) {
                hipStream_t cur = (i & 1) ? s2 : s1, nxt = (i & 1) ? s1 : s2;
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, cur, p, 10);
                CK(hipEventRecord(ev[i], cur));
                CK(hipStreamWaitEvent(nxt, ev[i], 0));
            }
            CK(hipStreamEndCapture(s1, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s1)); CK(hipStreamSynchronize(s1));
            auto t0 = clk::now();
            for (int i = 0; i < 10; i++) CK(hipGraphLaunch(ge, s1));
            CK(hipStreamSynchronize(s1));
            std::printf("graph of that 64-kernel ping-pong chain: %.1f us per replay\n", us_since(t0) / 10);
            // linear chain graph for comparison
            hipGraph_t g2; hipGraphExec_t ge2;
            CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < 64; i++) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, p, 10);
            CK(hipStreamEndCapture(s1, &g2));
            CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge2, s1)); CK(hipStreamSynchronize(s1));
            t0 = clk::now();
            for (int i = 0; i < 10; i++) CK(hipGraphLaunch(ge2, s1));
            CK(hipStreamSynchronize(s1));
            std::printf("graph of a 64-kernel linear chain: %.1f us per replay\n", us_since(t0) / 10);
            // 3 parallel chains of 21 kernels forked from one and joined (the Detect branches' shape)
            hipGraph_t g3; hipGraphExec_t ge3;
            CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, p, 10);
            CK(hipEventRecord(ev[0], s1));
            CK(hipStreamWaitEvent(s2, ev[0], 0)); CK(hipStreamWaitEvent(s3, ev[0], 0));
            for (int i = 0; i < 21; i++) {
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, p, 10);
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s2, p + 64, 10);
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s3, p + 128, 10);
            }
            CK(hipEventRecord(ev[1], s2)); CK(hipEventRecord(ev[2], s3));
            CK(hipStreamWaitEvent(s1, ev[1], 0)); CK(hipStreamWaitEvent(s1, ev[2], 0));
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, p, 10);
            CK(hipStreamEndCapture(s1, &g3));
            CK(hipGraphInstantiate(&ge3, g3, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge3, s1)); CK(hipStreamSynchronize(s1));
            t0 = clk::now();
            for (int i = 0; i < 10; i++) CK(hipGraphLaunch(ge3, s1));
            CK(hipStreamSynchronize(s1));
            std::printf("graph: 1 + 3 parallel chains of 21 + 1 (65 kernels): %.1f us per replay\n", us_since(t0) / 10);
        }
    }
    // ---- 4. small D2H after a kernel, in-stream; and a kernel writing straight into pinned host memory ----
    {
        float *p; CK(hipMalloc((void **)&p, 1 << 16));
        int *hp; CK(hipHostMalloc((void **)&hp, 1 << 16, hipHostMallocDefault));
        for (int rep = 0; rep < 2; rep++) {
            auto t0 = clk::now();
            for (int i = 0; i < 50; i++) {
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, p, 10);
                CK(hipStreamSynchronize(s1));
            }
            const double base = us_since(t0) / 50;
            t0 = clk::now();
            for (int i = 0; i < 50; i++) {
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, p, 10);
                CK(hipMemcpyAsync(hp, p, 20816, hipMemcpyDeviceToHost, s1));
                CK(hipStreamSynchronize(s1));
            }
            const double with = us_since(t0) / 50;
            int *dp = nullptr;
            CK(hipHostGetDevicePointer((void **)&dp, hp, 0));
            t0 = clk::now();
            int bad = 0;
            for (int i = 0; i < 50; i++) {
                hipLaunchKernelGGL(write_host_kernel, dim3(1), dim3(64), 0, s1, dp, i);
                CK(hipStreamSynchronize(s1));
                bad += hp[63] != i + 63;
            }
            const double zc = us_since(t0) / 50;
            if (rep) std::printf("kernel+sync %.1f us; kernel+20KB D2H+sync %.1f us; kernel writing pinned host+sync %.1f us (stale reads %d)\n", base, with, zc, bad);
        }
        // H2D of one frame then a kernel, same stream vs copy on another stream + event
        hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (int rep = 0; rep < 2; rep++) {
            auto t0 = clk::now();
            for (int i = 0; i < 30; i++) {
                CK(hipMemcpyAsync(d, h, 3932160, hipMemcpyHostToDevice, s1));
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, p, 10);
                CK(hipStreamSynchronize(s1));
            }
            const double inl = us_since(t0) / 30;
            t0 = clk::now();
            for (int i = 0; i < 30; i++) {
                CK(hipMemcpyAsync(d, h, 3932160, hipMemcpyHostToDevice, s2));
                CK(hipEventRecord(e, s2));
                CK(hipStreamWaitEvent(s1, e, 0));
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, p, 10);
                CK(hipStreamSynchronize(s1));
            }
            const double hop = us_since(t0) / 30;
            if (rep) std::printf("frame H2D + kernel + sync: same stream %.1f us; copy on another stream + event %.1f us\n", inl, hop);
        }
    }
    std::printf("probe done\n");
    return 0;
}
