// Timing probe for front_kernel (1280 x 1024 -> 640, stretch: the benchmark's geometry) outside the engine;
// -DIRMV_FSTAMP=1 adds per-phase shader-clock sums (wave 0 of every workgroup).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include scripts/probes/front_probe.cpp -o front_probe ; ./front_probe [batch=64]
#include "../../irmv_detection_amd/csrc/k_front.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using irmv::AxisTap;

static unsigned long long rs = 88172645463325252ull;
static unsigned rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (unsigned)(rs >> 11); }

static void taps(std::vector<AxisTap> &out, int dn, int sn)
{
    out.resize(dn);
    for (int d = 0; d < dn; d++) {
        const long long num = (long long)(2 * d + 1) * sn - dn, den = 2LL * dn;
        const long long fl = num >= 0 ? num / den : -((-num + den - 1) / den);
        int w = (int)(((num - fl * den) * 2048 + dn) / den), a = (int)fl, b = a + 1;
        if (a < 0) { a = 0; b = 0; w = 0; }
        if (a >= sn - 1) { a = sn - 1; b = sn - 1; w = 0; }
        out[d] = AxisTap{a, b, w, 0};
    }
}
template <class T> static T *to_dev(const std::vector<T> &h)
{
    T *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess || hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) exit(3);
    return d;
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 64, net = 640, sw = 1280, sh = 1024, W1 = net / 4;
    std::vector<AxisTap> tx, ty;
    taps(tx, net, sw); taps(ty, net, sh);
    irmv::FrontArgs a{};
    std::vector<uint8_t> src((size_t)B * sw * sh * 3);
    for (auto &v : src) v = (uint8_t)rnd();
    a.src = to_dev(src); a.src_slot_bytes = (size_t)sw * sh * 3;
    a.tx = to_dev(tx); a.ty = to_dev(ty);
    a.vx0 = 0; a.vx1 = net; a.vy0 = 0; a.vy1 = net;
    a.sw = sw; a.sh = sh; a.net = net; a.swap_rb = 1;
    a.fastx = getenv("FASTX") ? atoi(getenv("FASTX")) : 3; a.fx_i0 = 0; a.fx_step = 2;
    std::vector<irmv::half_t> w0(2 * 64 * 8), w1(10 * 64 * 8);
    for (auto &v : w0) v = (irmv::half_t)(((int)(rnd() & 255) - 128) / 512.0f);
    for (auto &v : w1) v = (irmv::half_t)(((int)(rnd() & 255) - 128) / 1024.0f);
    a.w0 = to_dev(w0); a.w1 = to_dev(w1);
    a.b0 = to_dev(std::vector<float>(16, 0.05f)); a.b1 = to_dev(std::vector<float>(32, 0.05f));
    irmv::half_t *out = nullptr;
    CK(hipMalloc(&out, (size_t)B * W1 * W1 * 32 * 2));
    a.out = out; a.out_ld = 32;
    a.tile_y = (getenv("TILE8") ? atoi(getenv("TILE8")) : 1) && (a.fastx & 2) ? irmv::kFrontTileYDirect : irmv::kFrontTileY;
    a.tiles_x = (W1 + irmv::kFrontTileX - 1) / irmv::kFrontTileX; a.tiles_y = (W1 + a.tile_y - 1) / a.tile_y;
    {   // largest source region of a tile (engine.cpp front_fits)
        auto span = [&](const std::vector<AxisTap> &t, int g0, int n, int *lo, int *hi) {
            *lo = 0x7fffffff; *hi = -1;
            for (int i = std::max(g0, 0); i < std::min(g0 + n, net); i++) { *lo = std::min({*lo, t[i].i0, t[i].i1}); *hi = std::max({*hi, t[i].i0, t[i].i1}); }
        };
        int mp = 0, mr = 0, lo, hi;
        for (int i = 0; i < a.tiles_x; i++) { span(tx, 4 * i * irmv::kFrontTileX - 3, 4 * irmv::kFrontTileX + 3, &lo, &hi); mp = std::max(mp, std::min((hi + 4) & ~3, sw) - (lo & ~3)); }
        for (int i = 0; i < a.tiles_y; i++) { span(ty, 4 * i * irmv::kFrontTileY - 3, 4 * irmv::kFrontTileY + 3, &lo, &hi); mr = std::max(mr, hi - lo + 1); }
        a.stage_bytes = a.tile_y == irmv::kFrontTileY ? std::max((mp * mr * 4 + 255) & ~255, irmv::front_min_stage_bytes()) : irmv::front_min_stage_bytes(a.tile_y);
        printf("tiles %d x %d, region <= %d px x %d rows, stage %d B\n", a.tiles_x, a.tiles_y, mp, mr, a.stage_bytes);
    }
    if (!irmv::front_prepare()) { fprintf(stderr, "front_prepare failed\n"); return 2; }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 30; i++)
        if (!irmv::launch_front(a, B, st)) { fprintf(stderr, "launch refused\n"); return 2; }
    CK(hipStreamSynchronize(st));
    const int reps = 50;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) irmv::launch_front(a, B, st);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = (double)B * (sw * sh * 3.0 + W1 * W1 * 64.0);
    printf("front_kernel B=%d: %.1f us  %.2f TB/s algorithmic = %.3f of 8 TB/s\n", B, us, bytes / us * 1e-6, bytes / us * 1e-6 / 8.0);
#if IRMV_FSTAMP
    {
        const int nwg = std::min(a.tiles_x * a.tiles_y * B, 65536);
        std::vector<unsigned long long> h((size_t)nwg * 9);
        irmv::launch_front(a, B, st);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(irmv::g_front_stamps), h.size() * 8));
        static const char *nm[8] = {"0 weights, box, tap table", "A1 source -> LDS", "barrier", "A2 resample", "barrier", "B model.0 (tiles)", "barrier", "C model.1 + store"};
        double ph[8] = {0}, tot = 0;
        for (int w = 0; w < nwg; w++)
            for (int k = 0; k < 8; k++) ph[k] += (double)(h[(size_t)w * 9 + k + 1] - h[(size_t)w * 9 + k]);
        for (int k = 0; k < 8; k++) tot += ph[k];
        for (int k = 0; k < 8; k++) printf("  %-28s %8.0f clk  %5.1f %%\n", nm[k], ph[k] / nwg, 100.0 * ph[k] / tot);
        printf("  workgroup lifetime %.0f s_memtime ticks, %d workgroups\n", tot / nwg, nwg);
    }
#endif
    return 0;
}
