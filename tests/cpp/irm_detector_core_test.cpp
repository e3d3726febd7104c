// IrmDetectorCore (include/irmv_detection/irm_detector_core.hpp) = the ROS-free body of the reference node
// (reference src/irm_detector.cpp:25-78 ctor, :176-245 message_callback) on the GPU, checked here against the
// CPU oracle's C entry points (oracle/irmv_oracle.h; test infrastructure only):
//   * every message entry's pose == orc_solve_pnp_ippe on that armor's four points   (<= 1e-6)
//   * its quaternion == orc_rvec_to_quat(rvec) up to sign                             (<= 1e-6)
//   * distance_to_image_center == |center - (cx, cy)|                                 (<= 1e-3 px)
//   * bbox-only model: the armors == orc_extract_armor on the ROTATED frame for each bbox, same order (:292-355)
//   * keypoint model: one armor per detection
//   * live parameter update (param_event_callback, :372-403) changes the extraction like the oracle's does.
// argv: <model.onnx path (sibling .irmw exists)> <raw frames file> <n frames> <expect: kpt|classical>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

#include "irmv_detection/irm_detector_core.hpp"

extern "C" {
#include "irmv_oracle.h"
}

using namespace irmv_detection;

static int fails = 0;
#define CHECK(cond, ...)                                   \
  do {                                                     \
    if (!(cond)) {                                         \
      std::printf("FAIL %s:%d: ", __FILE__, __LINE__);     \
      std::printf(__VA_ARGS__);                            \
      std::printf("\n");                                   \
      fails++;                                             \
    }                                                      \
  } while (0)

int main(int argc, char ** argv)
{
  if (argc < 5) return 2;
  const int n_frames = std::atoi(argv[3]);
  const bool expect_kpt = std::strcmp(argv[4], "kpt") == 0;
  const cv::Size size(1280, 1024);
  const size_t fb = size_t(size.width) * size.height * 3;
  std::vector<uint8_t> frames(fb * n_frames);
  std::ifstream f(argv[2], std::ios::binary);
  f.read(reinterpret_cast<char *>(frames.data()), std::streamsize(frames.size()));
  if (!f) return 3;

  const std::array<double, 9> K = {957.669211, 0, 345.943891, 0, 969.127115, 284.057302, 0, 0, 1};   // config/camera_info.yaml:7
  const std::vector<double> D = {-0.405274, 0.126058, -0.026939, -0.006503, 0.0};                     // :12
  IrmDetectorCore::Params prm;
  prm.profiling = true;
  IrmDetectorCore core(argv[1], K, D, prm);
  CHECK(core.engine(0).has_keypoint_head() == expect_kpt, "point source");
  const auto bufs = core.image_buffers();
  CHECK(bufs[0] && bufs[1] && bufs[2] && bufs[0] != bufs[1], "three distinct frame slots");

  orc_light_params lp;
  orc_light_params_default(&lp);
  long total_armors = 0, total_boxes = 0;
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {   // live parameter update, both sides
      CHECK(core.set_parameter("binary_threshold", 120), "set binary_threshold");
      CHECK(core.set_parameter("light.max_angle", 25.0), "set light.max_angle");
      CHECK(!core.set_parameter("debug", 1), "non hot-path parameter is refused");
      lp.binary_threshold = 120;
      lp.light_max_angle = 25.0f;
    }
    for (int i = 0; i < n_frames; i++) {
      StampedFrame img;
      img.id = i % 3;                                     // the triple buffer's slot id (src/camera.cpp:24-29)
      img.time_stamp = std::chrono::system_clock::now();
      const uint8_t * src = frames.data() + size_t(i) * fb;
      std::memcpy(bufs[size_t(img.id)], src, fb);         // the producer's deposit
      const IrmDetectorCore::FrameResult r = core.message_callback(img);
      CHECK(r.armors_msg.header.frame_id == "camera_optical_frame", "frame id");
      CHECK(r.armors_msg.header.stamp_ns > 0, "stamp");
      CHECK(r.armors_msg.armors.size() == r.armors.size() && r.armors.size() == r.poses.size(), "sizes");
      CHECK(r.inference_latency_ms > 0, "profiling time");
      total_boxes += long(r.bboxes.size());
      total_armors += long(r.armors.size());

      // expected armors from the oracle
      std::vector<std::array<float, 8>> want_pts;
      if (expect_kpt) {
        CHECK(r.armors.size() <= r.bboxes.size(), "at most one armor per detection");
      } else {
        std::vector<uint8_t> rot(fb);
        orc_rotate180(src, size.width, size.height, rot.data());
        for (const auto & b : r.bboxes) {
          int sz = 0, nl = 0;
          float pts[8], c[2];
          if (orc_extract_armor(rot.data(), size.width, size.height, b.xyxy.data(), &lp, &sz, pts, c, &nl) == 1) {
            std::array<float, 8> p;
            std::memcpy(p.data(), pts, sizeof pts);
            want_pts.push_back(p);
          }
        }
      }
      size_t wi = 0;
      for (size_t k = 0; k < r.armors.size(); k++) {
        const Armor & a = r.armors[k];
        const ArmorMsg & m = r.armors_msg.armors[k];
        const float pts[8] = {a.left_light.bottom.x, a.left_light.bottom.y, a.left_light.top.x, a.left_light.top.y,
                              a.right_light.top.x, a.right_light.top.y, a.right_light.bottom.x, a.right_light.bottom.y};
        double rv[3], tv[3], rv2[3], tv2[3], err[2], q[4];
        const int ok = orc_solve_pnp_ippe(K.data(), D.data(), pts, 0, rv, tv, rv2, tv2, err);
        CHECK(ok == 1, "oracle PnP fails where the GPU succeeded (frame %d armor %zu)", i, k);
        if (std::fabs(err[0] - err[1]) > 1e-7) {           // two equally good solutions: order is numerically arbitrary
          const double dt = std::max({std::fabs(tv[0] - m.pose.position.x), std::fabs(tv[1] - m.pose.position.y), std::fabs(tv[2] - m.pose.position.z)});
          CHECK(dt <= 1e-6, "tvec differs by %g (frame %d armor %zu)", dt, i, k);
          double dr = 0;
          for (int c = 0; c < 3; c++) dr = std::max(dr, std::fabs(rv[c] - r.poses[k].rvec[c]));
          CHECK(dr <= 1e-6, "rvec differs by %g", dr);
          orc_rvec_to_quat(rv, q);
          const double g[4] = {m.pose.orientation.x, m.pose.orientation.y, m.pose.orientation.z, m.pose.orientation.w};
          double dp = 0, dm = 0;
          for (int c = 0; c < 4; c++) { dp = std::max(dp, std::fabs(q[c] - g[c])); dm = std::max(dm, std::fabs(q[c] + g[c])); }
          CHECK(std::min(dp, dm) <= 1e-6, "quaternion differs by %g", std::min(dp, dm));
        }
        const double dc = std::hypot(double(a.center.x) - K[2], double(a.center.y) - K[5]);
        CHECK(std::fabs(dc - m.distance_to_image_center) <= 1e-3, "distance_to_image_center %g vs %g", double(m.distance_to_image_center), dc);
        if (!expect_kpt) {
          // same armors, same order as the oracle's extraction (those whose PnP succeeded)
          bool found = false;
          while (wi < want_pts.size() && !found) {
            float d = 0;
            for (int c = 0; c < 8; c++) d = std::max(d, std::fabs(want_pts[wi][size_t(c)] - pts[c]));
            wi++;
            found = d <= 1e-4f;
          }
          CHECK(found, "armor %zu of frame %d is not among the oracle's (in order)", k, i);
        }
      }
      if (!expect_kpt) CHECK(r.armors.size() <= want_pts.size(), "more armors than the oracle found");
    }
  }
  std::printf("core frames %d x 2 passes: bboxes %ld armors %ld fails %d\n", n_frames, total_boxes, total_armors, fails);
  return fails == 0 ? 0 : 1;
}
