// Timing probe for conv3x3_lds_kernel on one layer shape, outside the engine.  (Rounds 2 - 4 drove ablation switches inside
// the kernel from here -- IRMV_ABL / IRMV_EXP: weights or patch staged once, no MFMAs, no SiLU, no stores, late epilogue,
// staggered second workgroup, phase stamps; results in DESIGN.md sections 4c, 4f.  The switches left the product sources in round 5.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I include scripts/probes/conv_probe.cpp -o conv_probe
//   ./conv_probe [S=80] [Cin=64] [batch=64] [mt=4] [nt=4] [ipw=4]
#include "../../irmv_detection_amd/csrc/k_conv.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 80, Cin = argc > 2 ? atoi(argv[2]) : 64, B = argc > 3 ? atoi(argv[3]) : 64;
    const int mt = argc > 4 ? atoi(argv[4]) : 4, nt = argc > 5 ? atoi(argv[5]) : 4, ipw = argc > 6 ? atoi(argv[6]) : 4;
    const int Cout = 64;
    const size_t n_in = (size_t)B * S * S * Cin, n_out = (size_t)B * S * S * Cout;
    const size_t n_w = (size_t)(Cout / (16 * nt)) * (Cin / 32) * 9 * nt * 64 * 8;
    std::vector<irmv::half_t> h_in(n_in), h_w(n_w);
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (float)((x >> 11) & 0xffff) / 65536.0f - 0.5f; };
    for (auto &v : h_in) v = (irmv::half_t)rnd();
    for (auto &v : h_w) v = (irmv::half_t)(rnd() * 0.1f);
    std::vector<float> h_b(Cout, 0.1f);
    irmv::half_t *d_in, *d_w, *d_out; float *d_b;
    CK(hipMalloc(&d_in, n_in * 2)); CK(hipMalloc(&d_w, n_w * 2)); CK(hipMalloc(&d_out, n_out * 2)); CK(hipMalloc(&d_b, Cout * 4));
    CK(hipMemcpy(d_in, h_in.data(), n_in * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, h_w.data(), n_w * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, h_b.data(), Cout * 4, hipMemcpyHostToDevice));
    irmv::ConvArgs a{};
    a.s0 = {d_in, Cin, Cin, 0};
    a.Hin = a.Win = a.Hout = a.Wout = S;
    a.M = B * S * S; a.Cin = Cin; a.bias = d_b; a.out = d_out; a.out_ld = Cout; a.cout_pad = Cout; a.ksteps = 9 * Cin / 32; a.pair = 1;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++)
        if (!irmv::launch_conv_lds(1, mt, nt, ipw, a, d_w, B, st, false)) { fprintf(stderr, "not eligible\n"); return 2; }
    CK(hipStreamSynchronize(st));
    const int reps = 50;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) {
        irmv::launch_conv_lds(1, mt, nt, ipw, a, d_w, B, st, false);
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, fl = 2.0 * B * S * S * Cout * Cin * 9;
    printf("S=%d Cin=%d B=%d mt%d nt%d i%d: %.2f us  %.1f TFLOP/s\n", S, Cin, B, mt, nt, ipw, us, fl / us * 1e-6);
    return 0;
}
