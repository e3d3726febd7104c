// Timing + bitwise probe of the weights-resident 3x3 kernel (k_conv.hip, WR = 2) against the chunked LDS kernel on one layer
// shape (Cin = 64 -> 64), outside the engine; IRMV_WRES_STAGGER=0 / 1 in the environment switches the lockstep form's stagger.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include scripts/probes/wres_probe.cpp -o build_probe/wres_probe
//   ./wres_probe [S=80] [batch=64] [ref_mt=4] [ref_ipw=4] [ref_cm=0]
#include "../../irmv_detection_amd/csrc/k_conv.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 80, B = argc > 2 ? atoi(argv[2]) : 64;
    const int rmt = argc > 3 ? atoi(argv[3]) : 4, ripw = argc > 4 ? atoi(argv[4]) : 4, rcm = argc > 5 ? atoi(argv[5]) : 0;
    const int Cin = 64, Cout = 64, nt = 4;
    const size_t n_in = (size_t)B * S * S * Cin, n_out = (size_t)B * S * S * Cout;
    const size_t n_w = (size_t)(Cout / (16 * nt)) * (Cin / 32) * 9 * nt * 64 * 8;
    std::vector<irmv::half_t> h_in(n_in), h_w(n_w);
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (float)((x >> 11) & 0xffff) / 65536.0f - 0.5f; };
    for (auto &v : h_in) v = (irmv::half_t)rnd();
    for (auto &v : h_w) v = (irmv::half_t)(rnd() * 0.1f);
    std::vector<float> h_b(Cout);
    for (auto &v : h_b) v = rnd();
    irmv::half_t *d_in, *d_w, *d_ref, *d_out; float *d_b;
    CK(hipMalloc(&d_in, n_in * 2)); CK(hipMalloc(&d_w, n_w * 2)); CK(hipMalloc(&d_ref, n_out * 2)); CK(hipMalloc(&d_out, n_out * 2)); CK(hipMalloc(&d_b, Cout * 4));
    CK(hipMemcpy(d_in, h_in.data(), n_in * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, h_w.data(), n_w * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, h_b.data(), Cout * 4, hipMemcpyHostToDevice));
    irmv::ConvArgs a{};
    a.s0 = {d_in, Cin, Cin, 0};
    a.Hin = a.Win = a.Hout = a.Wout = S;
    a.M = B * S * S; a.Cin = Cin; a.bias = d_b; a.out = d_ref; a.out_ld = Cout; a.cout_pad = Cout; a.ksteps = 9 * Cin / 32; a.pair = 1;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double fl = 2.0 * B * S * S * Cout * Cin * 9;
    auto time_it = [&](auto launch, const char *name) -> int {
        for (int i = 0; i < 3; i++) if (!launch()) { printf("%-28s not eligible\n", name); return 0; }
        CK(hipStreamSynchronize(st));
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 20; i++) launch();
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double us = best * 1e3 / 20;
        printf("S=%d B=%d %-28s %8.2f us  %7.1f TFLOP/s\n", S, B, name, us, fl / us * 1e-6);
        return 0;
    };
    char nm[64];
    snprintf(nm, sizeof nm, "ref mt%d nt4 i%d cm%d", rmt, ripw, rcm);
    if (time_it([&] { return irmv::launch_conv_lds(1, rmt, nt, ripw, a, d_w, B, st, false, rcm); }, nm)) return 1;
    std::vector<irmv::half_t> h_ref(n_out), h_out(n_out);
    CK(hipMemcpy(h_ref.data(), d_ref, n_out * 2, hipMemcpyDeviceToHost));
    a.out = d_out;
    const int ipws[] = {2, 4, 7, 8, 13};
    for (int ipw : ipws) {
        if (ipw > B) continue;
        CK(hipMemset(d_out, 0xff, n_out * 2));
        snprintf(nm, sizeof nm, "wres i%d", ipw);
        for (int pp = 0; pp < 2; pp++) {
            CK(hipMemset(d_out, 0xff, n_out * 2));
            snprintf(nm, sizeof nm, "wres%s i%d", pp ? "_pp" : "", ipw);
            if (time_it([&] { return irmv::launch_conv_wres(ipw, a, d_w, B, st, pp != 0); }, nm)) return 1;
            CK(hipMemcpy(h_out.data(), d_out, n_out * 2, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < n_out; i++) bad += memcmp(&h_ref[i], &h_out[i], 2) != 0;
            printf("    %s vs ref: %zu of %zu elements differ%s\n", nm, bad, n_out, bad ? "  <-- MISMATCH" : " (bitwise equal)");
        }
    }
    return 0;
}
