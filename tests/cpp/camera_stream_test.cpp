// BASELINE configs[2]: a synthetic 1280x1024 camera stream (330 FPS paced, then
// un-paced) through the reference's hand-off contract on a discrete GPU:
//   producer thread  -> TripleBuffer slot == the engine's PINNED host frame slot
//                       (reference src/camera.cpp:24-29,40-61: the producer writes
//                       straight into engine[i]'s source buffer, then commits)
//   consumer thread  -> get_consumer_buffer() -> detect(slot): async H2D on the
//                       engine stream + captured step + results
//                       (reference src/camera.cpp:64-84, src/irm_detector.cpp:181)
// One engine with three slots and ONE copy of the weights replaces the reference's
// three full engines (src/irm_detector.cpp:35-38).  Terminating and asserting,
// unlike the reference's soak tests (test/camera_test.cpp, test/triple_buffer_test.cpp):
// paced run: consumer rate within 10 % of the producer's (triple_buffer_test.cpp:65-68)
// and capture->detections latency under the reference's 10 ms warning threshold
// (test/camera_test.cpp:37-42).
// argv: <model.onnx path> <raw frame file (N x 1280x1024x3)> <n frames in file> [seconds]
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

#include "irmv_detection/triple_buffer.hpp"
#include "irmv_hip.h"

using clk = std::chrono::steady_clock;

struct StampedImage {   // reference include/irmv_detection/camera.hpp:27-32
  uint8_t * data = nullptr;
  clk::time_point time_stamp;
  int id = 0;
  long seq = -1;
};

struct RunStats { double producer_fps, consumer_fps, lat_mean_ms, lat_p99_ms, lat_max_ms; long produced, consumed, dets; };

static RunStats run(irmv_engine * eng, const std::vector<uint8_t> & frames, int n_frames, size_t frame_bytes, double fps, double seconds, bool pipelined = false)
{
  std::array<StampedImage, 3> slots;
  for (int i = 0; i < 3; i++) { slots[i].data = irmv_engine_src_buffer(eng, i); slots[i].id = i; }
  irmv_detection::TripleBuffer<StampedImage> tb(slots);
  std::atomic<bool> stop{false};
  std::atomic<long> produced{0};
  std::vector<double> lat;
  lat.reserve(100000);
  long consumed = 0, dets = 0;
  const auto t_start = clk::now();

  std::thread producer([&] {
    const auto period = std::chrono::duration_cast<clk::duration>(std::chrono::duration<double>(fps > 0 ? 1.0 / fps : 0.0));
    auto next = clk::now();
    for (long i = 0; !stop.load(std::memory_order_relaxed); i++) {
      StampedImage * s = tb.get_producer_buffer();
      std::memcpy(s->data, frames.data() + size_t(i % n_frames) * frame_bytes, frame_bytes);   // the camera SDK's frame deposit
      s->time_stamp = clk::now();
      s->seq = i;
      tb.producer_commit();
      produced.fetch_add(1, std::memory_order_relaxed);
      if (fps > 0) { next += period; std::this_thread::sleep_until(next); }
    }
    StampedImage * s = tb.get_producer_buffer();   // release a consumer blocked in wait (reference src/camera.cpp:86-91)
    s->seq = -2;
    tb.producer_commit();
  });

  std::vector<irmv_det> out(100);
  long last_seq = -1;
  bool order_ok = true;
  // pipelined consumer: the slot's upload is awaited (then the TripleBuffer may hand the buffer back to the producer), its
  // kernels run while the NEXT buffer is fetched and uploaded; results are collected one frame late
  int pending = -1;
  clk::time_point pending_stamp;
  auto collect = [&](int slot, clk::time_point stamp) {
    int n = 0;
    if (irmv_engine_wait_slots(eng, slot, 1) != IRMV_OK || irmv_engine_results(eng, slot, out.data(), 100, &n) != IRMV_OK) {
      std::printf("collect failed: %s\n", irmv_last_error());
      std::exit(4);
    }
    lat.push_back(std::chrono::duration<double, std::milli>(clk::now() - stamp).count());
    consumed++;
    dets += n;
  };
  while (true) {
    StampedImage * s = tb.get_consumer_buffer();
    if (s->seq == -2) break;
    if (s->seq <= last_seq) order_ok = false;
    last_seq = s->seq;
    if (!pipelined) {
      int n = 0;
      const int rc = irmv_engine_detect(eng, s->id, out.data(), 100, &n);
      if (rc != IRMV_OK) { std::printf("detect failed: %s\n", irmv_last_error()); std::exit(4); }
      lat.push_back(std::chrono::duration<double, std::milli>(clk::now() - s->time_stamp).count());
      consumed++;
      dets += n;
    } else {
      if (pending == s->id) { collect(pending, pending_stamp); pending = -1; }   // the same slot again: finish its previous frame first
      if (irmv_engine_submit(eng, s->id, 1, IRMV_SUBMIT_H2D | IRMV_SUBMIT_ASYNC_UPLOAD) != IRMV_OK || irmv_engine_wait_upload(eng, s->id, 1) != IRMV_OK) {
        std::printf("submit failed: %s\n", irmv_last_error());
        std::exit(4);
      }
      const clk::time_point stamp = s->time_stamp;
      const int id = s->id;
      if (pending >= 0) collect(pending, pending_stamp);
      pending = id;
      pending_stamp = stamp;
    }
    if (std::chrono::duration<double>(clk::now() - t_start).count() > seconds) stop = true;
  }
  if (pending >= 0) collect(pending, pending_stamp);
  producer.join();
  const double T = std::chrono::duration<double>(clk::now() - t_start).count();
  std::sort(lat.begin(), lat.end());
  RunStats r{};
  r.produced = produced.load(); r.consumed = consumed; r.dets = dets;
  r.producer_fps = r.produced / T; r.consumer_fps = consumed / T;
  double sum = 0; for (double v : lat) sum += v;
  r.lat_mean_ms = lat.empty() ? 0 : sum / lat.size();
  r.lat_p99_ms = lat.empty() ? 0 : lat[size_t(0.99 * (lat.size() - 1))];
  r.lat_max_ms = lat.empty() ? 0 : lat.back();
  if (!order_ok) { std::printf("consumer saw a stale frame\n"); std::exit(5); }
  return r;
}

int main(int argc, char ** argv)
{
  if (argc < 4) return 2;
  const int n_frames = std::atoi(argv[3]);
  const double seconds = argc > 4 ? std::atof(argv[4]) : 2.0;
  const size_t frame_bytes = size_t(1280) * 1024 * 3;
  std::vector<uint8_t> frames(frame_bytes * n_frames);
  std::ifstream f(argv[2], std::ios::binary);
  f.read(reinterpret_cast<char *>(frames.data()), std::streamsize(frames.size()));
  if (!f) return 3;

  irmv_engine_cfg cfg;
  irmv_engine_cfg_default(&cfg);          // 1280x1024, rotate180, 3 slots: the reference node's configuration
  cfg.weights_path = argv[1];
  irmv_engine * eng = nullptr;
  if (irmv_engine_create(&cfg, &eng) != IRMV_OK) { std::printf("create failed: %s\n", irmv_last_error()); return 4; }
  for (int w = 0; w < 20; w++) {          // warm-up of every slot's path
    int n = 0; irmv_det d[100];
    irmv_engine_detect(eng, w % 3, d, 100, &n);
  }
  const RunStats paced = run(eng, frames, n_frames, frame_bytes, 330.0, seconds);
  std::printf("paced330 producer_fps %.1f consumer_fps %.1f lat_ms mean %.3f p99 %.3f max %.3f produced %ld consumed %ld dets %ld\n",
              paced.producer_fps, paced.consumer_fps, paced.lat_mean_ms, paced.lat_p99_ms, paced.lat_max_ms, paced.produced, paced.consumed, paced.dets);
  const RunStats freerun = run(eng, frames, n_frames, frame_bytes, 0.0, seconds);
  std::printf("unpaced producer_fps %.1f consumer_fps %.1f lat_ms mean %.3f p99 %.3f max %.3f produced %ld consumed %ld dets %ld\n",
              freerun.producer_fps, freerun.consumer_fps, freerun.lat_mean_ms, freerun.lat_p99_ms, freerun.lat_max_ms, freerun.produced, freerun.consumed, freerun.dets);
  const RunStats piped = run(eng, frames, n_frames, frame_bytes, 0.0, seconds, true);
  std::printf("unpaced_pipelined producer_fps %.1f consumer_fps %.1f lat_ms mean %.3f p99 %.3f max %.3f produced %ld consumed %ld dets %ld\n",
              piped.producer_fps, piped.consumer_fps, piped.lat_mean_ms, piped.lat_p99_ms, piped.lat_max_ms, piped.produced, piped.consumed, piped.dets);
  irmv_engine_destroy(eng);
  const bool rate_ok = paced.consumer_fps > 0.9 * paced.producer_fps && paced.producer_fps > 0.9 * 330.0;
  const bool lat_ok = paced.lat_p99_ms < 10.0;
  std::printf("rate_ok %d lat_ok %d\n", int(rate_ok), int(lat_ok));
  return (rate_ok && lat_ok) ? 0 : 1;
}
