// irmv_multi_gpu: the multi-GPU form of the hot path in ONE process, host side in C++ only (SURVEY.md section 7 step 7,
// BASELINE configs[3]): ncclCommInitAll over the node's GPUs, the weight blob read by rank 0 alone and broadcast ONCE
// over RCCL / xGMI (include/irmv_comm.h), then one host thread + one engine per GPU, independent frames sharded
// round-robin (frame i -> GPU i mod N), no per-frame collective.  The reference is single-device
// (test/yolo_test.cpp:16 `cudaSetDevice(0)`, src/irm_detector.cpp:35-38: three engines on that device).
//
//   hipcc -O2 -std=c++17 -I include tools/irmv_multi_gpu.cpp -L irmv_detection_amd/lib -lirmv_hip -lirmv_comm -lpthread -o irmv_multi_gpu
//   ./irmv_multi_gpu --weights model.irmw [--gpus N] [--slots 64] [--steps 20] [--group 32] [--src 1280x1024]
//
// Prints one JSON line: per-GPU and aggregate FPS with the frames resident in HBM, and with every frame starting in a
// pinned host slot and crossing PCIe inside the timed region (SURVEY 8(d)'s clock: the number a shared host resource --
// memory bandwidth, PCIe root complexes, NUMA -- would show up in).
#include "irmv_comm.h"
#include "irmv_hip.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {
struct Barrier {   // reusable thread barrier (C++17)
    std::mutex m; std::condition_variable cv; int n, waiting = 0; unsigned long gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        const unsigned long g = gen;
        if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// deterministic synthetic camera frame (dim background, a few bright bars): content is irrelevant to the timing
void synth_frame(uint8_t *p, int w, int h, unsigned long long idx)
{
    unsigned long long x = 0xC0FFEEull + idx * 0x9E3779B97F4A7C15ull;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (size_t i = 0; i < (size_t)w * h * 3; i++) p[i] = (uint8_t)(rnd() % 96);
    for (int b = 0; b < 6; b++) {
        const int bx = (int)(rnd() % (w - 40)), by = (int)(rnd() % (h - 140)), bw = 6 + (int)(rnd() % 8), bh = 40 + (int)(rnd() % 80);
        for (int y = by; y < by + bh; y++)
            for (int xx = bx; xx < bx + bw; xx++) memset(p + ((size_t)y * w + xx) * 3, 230, 3);
    }
}
}  // namespace

int main(int argc, char **argv)
{
    std::string weights;
    int gpus = 0, slots = 64, steps = 20, group = 32, sw = 1280, sh = 1024;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--weights") weights = val();
        else if (a == "--gpus") gpus = atoi(val());
        else if (a == "--slots") slots = atoi(val());
        else if (a == "--steps") steps = atoi(val());
        else if (a == "--group") group = atoi(val());
        else if (a == "--src") { if (sscanf(val(), "%dx%d", &sw, &sh) != 2) { fprintf(stderr, "--src WxH\n"); return 2; } }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (weights.empty()) { fprintf(stderr, "usage: irmv_multi_gpu --weights model.irmw [--gpus N] [--slots B] [--steps K] [--group G] [--src WxH]\n"); return 2; }
    int ndev = 0;
    if (irmv_device_count(&ndev) != IRMV_OK || ndev < 1) { fprintf(stderr, "no HIP device: %s\n", irmv_last_error()); return 1; }
    const int N = gpus > 0 ? std::min(gpus, ndev) : ndev;
    if (gpus > ndev) fprintf(stderr, "[irmv_multi_gpu] %d GPUs asked for, %d present: running on %d\n", gpus, ndev, N);
    slots = std::max(1, std::min(slots, 256));
    group = std::max(1, std::min(group, slots));

    // rank 0 alone reads the blob; everyone receives it by ONE broadcast
    std::vector<char> blob;
    {
        std::ifstream f(weights, std::ios::binary);
        if (!f) { fprintf(stderr, "cannot read %s\n", weights.c_str()); return 1; }
        blob.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    }
    irmv_comm *comm = nullptr;
    if (irmv_comm_init_all(N, nullptr, &comm) != IRMV_OK) { fprintf(stderr, "irmv_comm_init_all: %s\n", irmv_comm_last_error()); return 1; }
    std::vector<void *> wdev(N, nullptr);
    uint64_t wbytes = 0;
    const double tb0 = now();
    if (irmv_comm_broadcast_blob(comm, blob.data(), blob.size(), 0, wdev.data(), &wbytes) != IRMV_OK) { fprintf(stderr, "broadcast: %s\n", irmv_comm_last_error()); return 1; }
    const double t_bcast = now() - tb0;

    Barrier bar(N);
    std::vector<double> t_res(N, 0.0), t_host(N, 0.0);
    std::vector<long> dets(N, 0);
    std::atomic<int> failed{0};
    const int hsteps = std::max(4, steps / 2);
    std::vector<int> numa_node(N, -1), numa_bound(N, 0);
    auto worker = [&](int r) {
        // this thread fills GPU r's pinned slots and submits its steps: it runs on the CPUs of that GPU's own socket
        // (hipDeviceAttributeHostNumaId -> /sys/devices/system/node/nodeK/cpulist), and the engine allocates the slots there
        if (irmv_numa_device_node(r, &numa_node[r]) == IRMV_OK && numa_node[r] >= 0) numa_bound[r] = irmv_numa_bind_thread(numa_node[r]) == IRMV_OK;
        irmv_engine_cfg cfg;
        irmv_engine_cfg_default(&cfg);
        cfg.device = r; cfg.src_width = sw; cfg.src_height = sh; cfg.num_slots = slots;
        cfg.weights_blob = wdev[r]; cfg.weights_bytes = wbytes; cfg.weights_on_device = 1;
        irmv_engine *e = nullptr;
        bool ok = irmv_engine_create(&cfg, &e) == IRMV_OK;
        if (!ok) fprintf(stderr, "[gpu %d] irmv_engine_create: %s\n", r, irmv_last_error());
        if (ok) {
            for (int s = 0; s < slots; s++) synth_frame(irmv_engine_src_buffer(e, s), sw, sh, (unsigned long long)r + (unsigned long long)N * s);   // frame i -> GPU i mod N
            ok = irmv_engine_submit(e, 0, slots, IRMV_SUBMIT_H2D) == IRMV_OK && irmv_engine_wait(e) == IRMV_OK;
            for (int i = 0; i < 3 && ok; i++) ok = irmv_engine_submit(e, 0, slots, 0) == IRMV_OK;
            ok = ok && irmv_engine_wait(e) == IRMV_OK;
        }
        if (!ok) failed++;
        bar.wait();
        if (failed.load()) { if (e) irmv_engine_destroy(e); return; }
        // (i) frames resident in HBM
        double t0 = now();
        for (int k = 0; k < steps && ok; k++) ok = irmv_engine_submit(e, 0, slots, 0) == IRMV_OK;
        ok = ok && irmv_engine_wait(e) == IRMV_OK;
        t_res[r] = now() - t0;
        bar.wait();
        // (ii) every frame from its pinned host slot, upload groups on the upload stream
        for (int k = 0; k < 2 && ok; k++)
            for (int f = 0; f < slots && ok; f += group) ok = irmv_engine_submit(e, f, std::min(group, slots - f), IRMV_SUBMIT_H2D | IRMV_SUBMIT_ASYNC_UPLOAD) == IRMV_OK;
        ok = ok && irmv_engine_wait(e) == IRMV_OK;
        bar.wait();
        t0 = now();
        for (int k = 0; k < hsteps && ok; k++)
            for (int f = 0; f < slots && ok; f += group) ok = irmv_engine_submit(e, f, std::min(group, slots - f), IRMV_SUBMIT_H2D | IRMV_SUBMIT_ASYNC_UPLOAD) == IRMV_OK;
        ok = ok && irmv_engine_wait(e) == IRMV_OK;
        t_host[r] = now() - t0;
        std::vector<irmv_det> out(irmv_engine_max_det(e));
        for (int s = 0; s < slots && ok; s++) {
            int n = 0;
            ok = irmv_engine_results(e, s, out.data(), (int)out.size(), &n) == IRMV_OK;
            dets[r] += n;
        }
        if (!ok) { fprintf(stderr, "[gpu %d] %s\n", r, irmv_last_error()); failed++; }
        bar.wait();
        irmv_engine_destroy(e);
    };
    std::vector<std::thread> th;
    for (int r = 0; r < N; r++) th.emplace_back(worker, r);
    for (auto &t : th) t.join();
    irmv_comm_destroy(comm);
    if (failed.load()) return 1;

    const double tr = *std::max_element(t_res.begin(), t_res.end()), thh = *std::max_element(t_host.begin(), t_host.end());
    printf("{\"tool\": \"irmv_multi_gpu\", \"n_gpus\": %d, \"frames_per_step_per_gpu\": %d, \"steps\": %d, \"weights_bytes\": %llu, \"broadcast_ms\": %.2f, "
           "\"fps_hbm_resident\": %.1f, \"fps_host_inclusive\": %.1f, \"host_steps\": %d, \"upload_group\": %d, \"per_gpu_fps_hbm_resident\": [",
           N, slots, steps, (unsigned long long)wbytes, t_bcast * 1e3, (double)N * slots * steps / tr, (double)N * slots * hsteps / thh, hsteps, group);
    for (int r = 0; r < N; r++) printf("%s%.1f", r ? ", " : "", (double)slots * steps / t_res[r]);
    printf("], \"per_gpu_fps_host_inclusive\": [");
    for (int r = 0; r < N; r++) printf("%s%.1f", r ? ", " : "", (double)slots * hsteps / t_host[r]);
    printf("], \"detections_last_step\": [");
    for (int r = 0; r < N; r++) printf("%s%ld", r ? ", " : "", dets[r]);
    printf("], \"numa_node\": [");
    for (int r = 0; r < N; r++) printf("%s%d", r ? ", " : "", numa_node[r]);
    printf("], \"numa_thread_bound\": [");
    for (int r = 0; r < N; r++) printf("%s%d", r ? ", " : "", numa_bound[r]);
    printf("]}\n");
    return 0;
}
