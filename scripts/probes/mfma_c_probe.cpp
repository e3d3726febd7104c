// Does v_mfma_f32_16x16x32_f16 give the same bits when its C operand sits in other registers than its result
// (D = A*B + C, C != D: what the compiler emits for "accumulator starts at the bias") as when it accumulates in place
// (C == D after a register copy)?  And does a 4-step chain that STARTS at a non-zero C equal the zero-started chain plus
// nothing else changed?  hipcc -O2 --offload-arch=gfx950 mfma_c_probe.cpp -o mfma_c_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const half8 *A, const half8 *B, const f32x4 *C, f32x4 *out, int ksteps)
{
    const int lane = threadIdx.x;
    const f32x4 c = C[lane];
    // (1) C != D: dedicated result registers, C read from its own registers
    f32x4 d1;
    {
        half8 a = A[lane], b = B[lane];
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3\n s_nop 7\n s_nop 7" : "=&v"(d1) : "v"(a), "v"(b), "v"(c));
        for (int k = 1; k < ksteps; k++) {
            a = A[k * 64 + lane]; b = B[k * 64 + lane];
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n s_nop 7\n s_nop 7" : "+v"(d1) : "v"(a), "v"(b));
        }
    }
    // (2) in place from the start
    f32x4 d2 = c;
    for (int k = 0; k < ksteps; k++) {
        half8 a = A[k * 64 + lane], b = B[k * 64 + lane];
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n s_nop 7\n s_nop 7" : "+v"(d2) : "v"(a), "v"(b));
    }
    // (3) the builtin, compiler's choice
    f32x4 d3 = c;
    for (int k = 0; k < ksteps; k++) d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[k * 64 + lane], B[k * 64 + lane], d3, 0, 0, 0);
    // (4) zero start, bias added behind the chain (the old epilogue)
    f32x4 d4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < ksteps; k++) d4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[k * 64 + lane], B[k * 64 + lane], d4, 0, 0, 0);
    d4 += c;
    out[lane] = d1; out[64 + lane] = d2; out[128 + lane] = d3; out[192 + lane] = d4;
}

int main()
{
    const int KS = 4;
    std::vector<_Float16> a(KS * 64 * 8), b(KS * 64 * 8);
    std::vector<float> c(64 * 4), o(4 * 64 * 4);
    srand(1);
    for (auto &v : a) v = (_Float16)((rand() % 2001 - 1000) / 4000.0f);
    for (auto &v : b) v = (_Float16)((rand() % 2001 - 1000) / 500.0f);
    for (auto &v : c) v = (rand() % 2001 - 1000) / 300.0f;
    void *da, *db, *dc, *dout;
    hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dc, c.size() * 4); hipMalloc(&dout, o.size() * 4);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, (const half8 *)da, (const half8 *)db, (const f32x4 *)dc, (f32x4 *)dout, KS);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    auto diff = [&](int x, int y) { int n = 0; for (int i = 0; i < 256; i++) n += memcmp(&o[x * 256 + i], &o[y * 256 + i], 4) != 0; return n; };
    printf("C != D vs in place: %d of 256 differ; builtin vs in place: %d; bias behind the chain vs in place: %d\n", diff(0, 1), diff(2, 1), diff(3, 1));
    return 0;
}
