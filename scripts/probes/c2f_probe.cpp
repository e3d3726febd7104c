// Timing probe for c2f32_kernel<0, 6, false> (model.15 at a 640 net) outside the engine.  (The ablation / phase-stamp switches these probes drove inside the kernels -- IRMV_ABL, IRMV_EXP -- were
// taken out of the product sources in round 5: their results are in DESIGN.md sections 4c, 4f, 8.)
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
    printf("c2f32_ab 80x80 cin 192 B=%d: %.2f us\n", B, ms * 1e3 / reps);
    return 0;
}
