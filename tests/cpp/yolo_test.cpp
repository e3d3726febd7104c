// C++ drop-in check: the reference's test flow (reference test/yolo_test.cpp:
// yolo_engine_demo :14-51 and yolo_engine_benchmark :53-107) written against the
// facade headers in include/irmv_detection/, i.e. against the reference's own
// class names and call pattern.  No gtest / OpenCV / ament in this image, so it is
// a plain main(): argv = <model.onnx path> <raw 1280x1024x3 frame> [runs].
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <numeric>
#include <vector>

#include "irmv_detection/pnp_solver.hpp"
#include "irmv_detection/yolo_engine.hpp"

int main(int argc, char ** argv)
{
  if (argc < 3) return 2;
  const int runs = argc > 3 ? std::atoi(argv[3]) : 30;
  const cv::Size size(1280, 1024);
  std::vector<uint8_t> frame(size_t(size.width) * size.height * 3);
  std::ifstream f(argv[2], std::ios::binary);
  f.read(reinterpret_cast<char *>(frame.data()), std::streamsize(frame.size()));
  if (!f) return 3;

  // --- demo ---
  irmv_detection::YoloEngine yolo_engine(argv[1], size, true);
  uint8_t * src_image_buffer = yolo_engine.get_src_image_buffer();
  std::memcpy(src_image_buffer, frame.data(), frame.size());
  std::vector<irmv_detection::YoloEngine::bbox> bboxes = yolo_engine.detect();
  std::printf("bboxes %zu\n", bboxes.size());
  for (size_t i = 0; i < bboxes.size() && i < 3; i++)
    std::printf("bbox %zu %.6f %.6f %.6f %.6f %.6f %s\n", i, bboxes[i].xyxy[0], bboxes[i].xyxy[1], bboxes[i].xyxy[2],
                bboxes[i].xyxy[3], bboxes[i].score, irmv_detection::armor_class_name(bboxes[i].class_id));
  cv::Mat visualized_image = yolo_engine.get_rotated_image().clone();
  yolo_engine.visualize_bboxes(visualized_image, bboxes);
  const size_t last = frame.size() - 3;
  std::printf("rotated_ok %d\n", int(yolo_engine.get_rotated_image().data[0] == frame[last] && yolo_engine.get_rotated_image().data[last + 2] == frame[2]));
  std::printf("profiling_ms %.4f\n", yolo_engine.get_profiling_time());

  // --- the node's classical armor extraction on the detector's boxes (src/irm_detector.cpp:183), on the GPU ---
  {
    std::vector<irmv_det> poses;
    const auto armors = yolo_engine.extract_armors(bboxes, &poses);
    std::printf("classical armors %zu of %zu boxes, poses %zu\n", armors.size(), bboxes.size(), poses.size());
  }

  // --- PnP on the first armor, the node's call shape (src/irm_detector.cpp:204-216) ---
  int n = 0;
  const irmv_det * dets = yolo_engine.last_detections(&n);
  if (n > 0) {
    irmv_detection::PnPSolver pnp_solver({957.669211, 0, 345.943891, 0, 969.127115, 284.057302, 0, 0, 1},
                                         {-0.405274, 0.126058, -0.026939, -0.006503, 0.0});
    irmv_detection::Armor armor(
      irmv_detection::Light(cv::Point2f(dets[0].kpts[2], dets[0].kpts[3]), cv::Point2f(dets[0].kpts[0], dets[0].kpts[1])),
      irmv_detection::Light(cv::Point2f(dets[0].kpts[4], dets[0].kpts[5]), cv::Point2f(dets[0].kpts[6], dets[0].kpts[7])));
    cv::Mat rvec, tvec;
    const bool ok = pnp_solver.solvePnP(armor, rvec, tvec);
    double worst = 0;
    for (int i = 0; i < 3; i++) {
      worst = std::max(worst, std::abs(rvec.at<double>(i) - dets[0].rvec[i]));
      worst = std::max(worst, std::abs(tvec.at<double>(i) - dets[0].tvec[i]));
    }
    std::printf("pnp ok %d fused_ok %d worst_diff %.3e tvec %.9f %.9f %.9f dist %.4f\n", int(ok), dets[0].pnp_ok, worst,
                tvec.at<double>(0), tvec.at<double>(1), tvec.at<double>(2), pnp_solver.calculateDistanceToCenter(armor.center));
  }

  // --- benchmark: 100 warm-up, `runs` x 10 iterations, each = memcpy frame + detect() ---
  for (int i = 0; i < 100; i++) {
    std::memcpy(src_image_buffer, frame.data(), frame.size());
    yolo_engine.detect();
  }
  std::vector<double> avg_times;
  for (int run = 0; run < runs; run++) {
    const auto begin = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < 10; i++) {
      std::memcpy(src_image_buffer, frame.data(), frame.size());
      yolo_engine.detect();
    }
    const auto end = std::chrono::high_resolution_clock::now();
    avg_times.push_back(double(std::chrono::duration_cast<std::chrono::microseconds>(end - begin).count()) / 10000.0);
  }
  const double avg = std::accumulate(avg_times.begin(), avg_times.end(), 0.0) / double(avg_times.size());
  double mx = avg_times[0], mn = avg_times[0];
  for (double t : avg_times) { mx = std::max(mx, t); mn = std::min(mn, t); }
  std::printf("detect_ms avg %.4f max %.4f min %.4f\n", avg, mx, mn);
  return mx < 30.0 ? 0 : 1;  // reference pass/fail ceiling, test/yolo_test.cpp:106
}
