// Timing probe for c2f32_kernel<0, 6, false> (model.15 at a 640 net) outside the engine; -DIRMV_EXP=4 adds phase stamps.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include scripts/probes/c2f_probe.cpp -o c2f_probe ; ./c2f_probe [batch=64]
#include "../../irmv_detection_amd/csrc/k_c2f.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static unsigned long long rs = 88172645463325252ull;
static float rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (float)((rs >> 11) & 0xffff) / 65536.0f - 0.5f; }
static irmv::half_t *dev_half(size_t n, float scale)
{
    std::vector<irmv::half_t> h(n);
    for (auto &v : h) v = (irmv::half_t)(rnd() * scale);
    irmv::half_t *d = nullptr;
    if (hipMalloc(&d, n * 2) != hipSuccess || hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice) != hipSuccess) exit(3);
    return d;
}
static float *dev_float(size_t n)
{
    std::vector<float> h(n, 0.05f);
    float *d = nullptr;
    if (hipMalloc(&d, n * 4) != hipSuccess || hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) exit(3);
    return d;
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 64, S = 80;
    irmv::C2f32Args a{};
    a.s0 = {dev_half((size_t)B * 40 * 40 * 128, 1.f), 128, 128, 1};
    a.s1 = {dev_half((size_t)B * S * S * 64, 1.f), 64, 64, 0};
    a.cin1 = 192;
    a.cat = dev_half((size_t)B * S * S * 96, 1.f); a.cat_ld = 96; a.prev_coff = 64;
    a.out = dev_half((size_t)B * S * S * 64, 1.f); a.out_ld = 64;
    a.H = a.W = S; a.tiles_x = S / irmv::kC2f32TileW; a.tiles_y = S / irmv::kC2f32TileH;
    a.w_cv1 = dev_half(4 * 6 * 512, 0.1f); a.w_m1 = dev_half(2 * 9 * 512, 0.1f); a.w_m2 = dev_half(2 * 9 * 512, 0.1f); a.w_cv2 = dev_half(4 * 3 * 512, 0.1f);
    a.b_cv1 = dev_float(64); a.b_m1 = dev_float(32); a.b_m2 = dev_float(32); a.b_cv2 = dev_float(64);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++)
        if (!irmv::launch_c2f32(0, false, a, B, st)) { fprintf(stderr, "no kernel\n"); return 2; }
    CK(hipStreamSynchronize(st));
    const int reps = 50;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) irmv::launch_c2f32(0, false, a, B, st);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
#if IRMV_EXP & 4
    {
        unsigned long long z[16] = {0}, h[16];
        CK(hipMemcpyToSymbol(HIP_SYMBOL(irmv::g_c2f_phase), z, sizeof(z)));
        irmv::launch_c2f32(0, false, a, B, st);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(irmv::g_c2f_phase), sizeof(h)));
        const double n = (double)h[10];
        static const char *nm[7] = {"1 cv1 (weights, input, 15 tiles)", "barrier", "2 m.cv1 (weights, 12 tiles)", "barrier", "3 m.cv2 (weights, 8 tiles)", "barrier", "4 cv2 (weights, 8 tiles, stores)"};
        printf("  per workgroup (wave 0), %g workgroups: total %.0f cycles\n", n, h[8] / n);
        for (int k = 0; k < 7; k++) printf("    %-36s %9.0f cycles  %5.1f %%\n", nm[k], h[k] / n, 100.0 * h[k] / h[8]);
    }
#endif
    printf("ABL=%d EXP=%d c2f32_ab 80x80 cin 192 B=%d: %.2f us\n", IRMV_ABL, IRMV_EXP, B, ms * 1e3 / reps);
    return 0;
}
