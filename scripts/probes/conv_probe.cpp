// Timing probe for conv3x3_lds_kernel on one layer shape, outside the engine: compile with -DIRMV_ABL=<bits> to switch
// parts of the kernel off (k_conv.hip) and see what each costs.  Results of an ablated build are wrong by construction.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -DIRMV_ABL=0 scripts/probes/conv_probe.cpp -o conv_probe
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
#if IRMV_EXP & 2
    {
        const int sleeps = argc > 7 ? atoi(argv[7]) : 0;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(irmv::g_stagger_sleeps), &sleeps, sizeof(int)));
    }
#endif
    const int reps = 50;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) {
#if IRMV_EXP & 2
        void *cnt = nullptr;
        CK(hipGetSymbolAddress(&cnt, HIP_SYMBOL(irmv::g_cu_arrivals)));
        CK(hipMemsetAsync(cnt, 0, 4096 * 4, st));
#endif
        irmv::launch_conv_lds(1, mt, nt, ipw, a, d_w, B, st, false);
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
#if IRMV_EXP & 4
    {
        unsigned long long z[16] = {0}, h[16];
        CK(hipMemcpyToSymbol(HIP_SYMBOL(irmv::g_phase), z, sizeof(z)));
        irmv::launch_conv_lds(1, mt, nt, ipw, a, d_w, B, st, false);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(irmv::g_phase), sizeof(h)));
        const double n = (double)h[10];
        static const char *nm[7] = {"prologue+first loads", "wait loads + LDS writes", "barrier 1", "issue loads", "taps (LDS reads + MFMA)", "barrier 2", "epilogue"};
        printf("  per workgroup (wave 0), %g workgroups: total %.0f cycles, %.2f us wall -> %.2f GHz\n", n, h[8] / n, h[9] / n * 0.01, (double)h[8] / ((double)h[9] * 10.0));
        for (int k = 0; k < 7; k++) printf("    %-26s %9.0f cycles  %5.1f %%\n", nm[k], h[k] / n, 100.0 * h[k] / h[8]);
    }
#endif
    const double us = ms * 1e3 / reps, fl = 2.0 * B * S * S * Cout * Cin * 9;
    printf("EXP=%d sleeps=%s ABL=%d S=%d Cin=%d B=%d mt%d nt%d i%d: %.2f us  %.1f TFLOP/s\n", IRMV_EXP, argc > 7 ? argv[7] : "-", IRMV_ABL, S, Cin, B, mt, nt, ipw, us, fl / us * 1e-6);
    return 0;
}
