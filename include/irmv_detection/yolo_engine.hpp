// irmv_detection::YoloEngine on MI355X -- header-only facade over the C ABI of
// libirmv_hip.so with the reference's public interface (reference
// include/irmv_detection/yolo_engine.hpp:16-73): same constructor arguments,
// detect(), visualize_bboxes(), get_profiling_time(), get_rotated_image(),
// get_src_image_buffer().  Code written against the reference class
// (src/irm_detector.cpp:35-38,181-183; test/yolo_test.cpp:19-30,58-82) compiles
// unchanged against this one.
//
// Deliberate deviations (documented in DESIGN.md):
//  * HIP failures throw std::runtime_error (the reference checks no return code);
//  * the per-instance scale factors are per instance (the reference's are
//    function-local statics, src/yolo_engine.cpp:155-156);
//  * the source slot is never modified: rotation is folded into the sampling, so
//    get_rotated_image() materialises the rotated frame on demand (GPU kernel);
//  * a missing model file prints the reference's message and exit(0)s like
//    src/yolo_engine.cpp:38-39, but the file looked for is "<stem>.irmw".
#pragma once

#include <array>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "irmv_detection/armor.hpp"
#include "irmv_detection/cv_compat.hpp"
#if IRMV_HAVE_OPENCV
#include <opencv2/imgproc.hpp>   // cv::rectangle, cv::putText (visualize_bboxes)
#endif
#include "irmv_hip.h"

namespace irmv_detection
{
class YoloEngine
{
public:
  struct bbox
  {
    std::array<float, 4> xyxy;
    float score;
    ArmorClass class_id;

    bool operator==(const bbox & o) const { return xyxy == o.xyxy && score == o.score && class_id == o.class_id; }
  };

  // HIP device used when none is given: IRMV_DEVICE (one process or thread per GPU sets it), else 0
  static int default_device()
  {
    const char * v = std::getenv("IRMV_DEVICE");
    return v ? std::atoi(v) : 0;
  }

  // Network input size when none is given: IRMV_NET_SIZE (a multiple of 32), else the reference's hard-coded 640
  // (src/yolo_engine.cpp:98-99); BASELINE configs[4] runs its model at 416.
  static int default_net_size()
  {
    const char * v = std::getenv("IRMV_NET_SIZE");
    return v ? std::atoi(v) : 640;
  }

  // The reference's constructor (yolo_engine.hpp:28-30).  Extension arguments: `device` (HIP ordinal, -1 = default_device()),
  // `warm_up_now` (false: the owner calls warm_up() itself, e.g. after building several engines) and `net_size`
  // (-1 = default_net_size()).  The model file decides the architecture (YOLOv8n or its ShuffleNetV2-backbone variant).
  YoloEngine(const std::string & onnx_file_path, cv::Size src_image_size, bool enable_profiling = false, int device = -1,
             bool warm_up_now = true, int net_size = -1)
  : src_image_size_(src_image_size), enable_profiling_(enable_profiling)
  {
    irmv_engine_cfg cfg;
    irmv_engine_cfg_default(&cfg);
    cfg.net_size = net_size > 0 ? net_size : default_net_size();
    cfg.device = device >= 0 ? device : default_device();
    cfg.src_width = src_image_size.width;
    cfg.src_height = src_image_size.height;
    cfg.num_slots = 1;  // one engine per TripleBuffer slot, like the reference node
    cfg.weights_path = onnx_file_path.c_str();
    const int rc = irmv_engine_create(&cfg, &engine_);
    if (rc == IRMV_ERR_MODEL) {
      std::cout << "Please convert the model to <stem>.irmw first (" << irmv_last_error() << ")." << std::endl;
      std::exit(0);
    }
    if (rc != IRMV_OK) throw std::runtime_error(std::string("YoloEngine: ") + irmv_last_error());
    src_image_buffer_ = irmv_engine_src_buffer(engine_, 0);
    rotated_ = cv::Mat(src_image_size.height, src_image_size.width, CV_8UC3);
    dets_.resize(static_cast<size_t>(irmv_engine_max_det(engine_)));
    if (warm_up_now) warm_up();
  }

  void warm_up()
  {
    for (int i = 0; i < 50; i++) detect();  // as src/yolo_engine.cpp:114-116
  }

  // true: the model has a keypoint head and every detection carries its four armor points and pose (irmv_det);
  // false: a bbox-only model like the reference's -- the points come from extract_armors()
  bool has_keypoint_head() const { return irmv_engine_point_source(engine_) == IRMV_POINTS_KEYPOINT_HEAD; }

  // The node's live parameters of the classical extraction (src/irm_detector.cpp:372-403)
  void set_extract_params(int binary_threshold, float light_min_ratio, float light_max_ratio, float light_max_angle, double min_small_cd,
                          double max_small_cd, double min_large_cd, double max_large_cd)
  {
    const double cd[4] = {min_small_cd, max_small_cd, min_large_cd, max_large_cd};
    if (irmv_engine_set_extract_params(engine_, binary_threshold, light_min_ratio, light_max_ratio, light_max_angle, cd) != IRMV_OK)
      throw std::runtime_error(std::string("YoloEngine::set_extract_params: ") + irmv_last_error());
  }

  ~YoloEngine() { irmv_engine_destroy(engine_); }
  YoloEngine(const YoloEngine &) = delete;
  YoloEngine & operator=(const YoloEngine &) = delete;

  std::vector<bbox> detect()
  {
    int n = 0;
    const int rc = irmv_engine_detect(engine_, 0, dets_.data(), static_cast<int>(dets_.size()), &n);
    if (rc != IRMV_OK) throw std::runtime_error(std::string("YoloEngine::detect: ") + irmv_last_error());
    rotated_valid_ = false;
    std::vector<bbox> out;
    out.reserve(static_cast<size_t>(n));
    for (int i = 0; i < n; i++) {
      const irmv_det & d = dets_[static_cast<size_t>(i)];
      out.push_back(bbox{{d.xyxy[0], d.xyxy[1], d.xyxy[2], d.xyxy[3]}, d.score, static_cast<ArmorClass>(d.class_id)});
    }
    n_last_ = n;
    return out;
  }

  // The full per-armor result of the last detect(): keypoints and pose, computed on the GPU.
  const irmv_det * last_detections(int * n) const
  {
    *n = n_last_;
    return dets_.data();
  }

  // Extension: the node's IrmDetector::extract_armors(get_rotated_image(), bboxes) (reference src/irm_detector.cpp:183,
  // 292-355) on the GPU, on the frame of the last detect().  One Armor per bbox in which two gated lights were found,
  // in bbox order, like the reference; `poses`, if given, receives the IPPE pose of each returned armor.
  std::vector<Armor> extract_armors(const std::vector<bbox> & bboxes, std::vector<irmv_det> * poses = nullptr)
  {
    std::vector<Armor> armors;
    if (bboxes.empty()) return armors;
    std::vector<float> xyxy;
    xyxy.reserve(bboxes.size() * 4);
    for (const auto & b : bboxes) xyxy.insert(xyxy.end(), b.xyxy.begin(), b.xyxy.end());
    std::vector<irmv_det> out(bboxes.size());
    if (irmv_engine_extract_armors(engine_, 0, xyxy.data(), static_cast<int>(bboxes.size()), out.data()) != IRMV_OK)
      throw std::runtime_error(std::string("YoloEngine::extract_armors: ") + irmv_last_error());
    for (size_t i = 0; i < out.size(); i++) {
      const irmv_det & d = out[i];
      if (d.armor_valid != 1) continue;   // no pair of lights (0) or scratch exhausted (-1): the reference emits nothing either
      Armor a(Light(cv::Point2f(d.kpts[2], d.kpts[3]), cv::Point2f(d.kpts[0], d.kpts[1])),
              Light(cv::Point2f(d.kpts[4], d.kpts[5]), cv::Point2f(d.kpts[6], d.kpts[7])));
      a.size = d.armor_size == IRMV_ARMOR_LARGE ? ArmorSize::LARGE : ArmorSize::SMALL;
      a.armor_class = bboxes[i].class_id;
      a.confidence = bboxes[i].score;
      armors.push_back(a);
      if (poses) poses->push_back(d);
    }
    return armors;
  }

  void visualize_bboxes(cv::Mat & image, const std::vector<bbox> & bboxes) const
  {
    if (image.cols != src_image_size_.width || image.rows != src_image_size_.height) {
      std::cerr << "[YoloEngine::visualize_bboxes] Image size mismatch" << std::endl;
      return;
    }
    for (const auto & b : bboxes) {
      const std::string name(armor_class_name(b.class_id));
      const bool blue = name[0] == 'B';
#if IRMV_HAVE_OPENCV
      // the reference's own calls (src/yolo_engine.cpp:229-241)
      const cv::Point p1(int(b.xyxy[0]), int(b.xyxy[1])), p2(int(b.xyxy[2]), int(b.xyxy[3]));
      const cv::Scalar color = blue ? cv::Scalar(0, 0, 255) : cv::Scalar(255, 0, 0);
      cv::rectangle(image, p1, p2, color, 2);
      cv::putText(image, name, p1, cv::FONT_HERSHEY_SIMPLEX, 1, color, 2);
#else
      draw_rect(image, int(b.xyxy[0]), int(b.xyxy[1]), int(b.xyxy[2]), int(b.xyxy[3]), blue ? 0 : 255, 0, blue ? 255 : 0);
      draw_label(image, name, int(b.xyxy[0]), int(b.xyxy[1]), blue ? 0 : 255, 0, blue ? 255 : 0);
#endif
    }
  }

  double get_profiling_time() const { return enable_profiling_ ? irmv_engine_last_detect_ms(engine_) : 0.0; }

  const cv::Mat & get_rotated_image() const
  {
    if (!rotated_valid_) {
      if (irmv_engine_rotated_image(engine_, 0, rotated_.data) != IRMV_OK)
        throw std::runtime_error(std::string("YoloEngine::get_rotated_image: ") + irmv_last_error());
      rotated_valid_ = true;
    }
    return rotated_;
  }

  uint8_t * get_src_image_buffer() const { return src_image_buffer_; }

private:
  static void draw_rect(cv::Mat & img, int x1, int y1, int x2, int y2, int c0, int c1, int c2)
  {
    auto put = [&](int x, int y) {
      if (x < 0 || y < 0 || x >= img.cols || y >= img.rows) return;
      uint8_t * p = img.data + (size_t(y) * img.cols + x) * 3;
      p[0] = uint8_t(c0); p[1] = uint8_t(c1); p[2] = uint8_t(c2);
    };
    for (int t = 0; t < 2; t++) {
      for (int x = x1; x <= x2; x++) { put(x, y1 + t); put(x, y2 - t); }
      for (int y = y1; y <= y2; y++) { put(x1 + t, y); put(x2 - t, y); }
    }
  }

  // Class label without OpenCV: the ArmorClass names (B1..B5, BO, BS, R1..R5, RO, RS, UNKNOWN) in a 5 x 7 bitmap font at
  // scale 3 (glyphs 15 x 21 px, 18 px advance: the size of FONT_HERSHEY_SIMPLEX at scale 1), text origin = bottom-left
  // corner at (x, y) like cv::putText (src/yolo_engine.cpp:238-241).
  static const uint8_t * glyph(char c)
  {
    static const uint8_t font[][8] = {
      {'B', 0x1e, 0x11, 0x11, 0x1e, 0x11, 0x11, 0x1e}, {'R', 0x1e, 0x11, 0x11, 0x1e, 0x14, 0x12, 0x11},
      {'O', 0x0e, 0x11, 0x11, 0x11, 0x11, 0x11, 0x0e}, {'S', 0x0f, 0x10, 0x10, 0x0e, 0x01, 0x01, 0x1e},
      {'U', 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x0e}, {'N', 0x11, 0x19, 0x15, 0x13, 0x11, 0x11, 0x11},
      {'K', 0x11, 0x12, 0x14, 0x18, 0x14, 0x12, 0x11}, {'W', 0x11, 0x11, 0x11, 0x15, 0x15, 0x1b, 0x11},
      {'1', 0x04, 0x0c, 0x04, 0x04, 0x04, 0x04, 0x0e}, {'2', 0x0e, 0x11, 0x01, 0x02, 0x04, 0x08, 0x1f},
      {'3', 0x1e, 0x01, 0x01, 0x0e, 0x01, 0x01, 0x1e}, {'4', 0x02, 0x06, 0x0a, 0x12, 0x1f, 0x02, 0x02},
      {'5', 0x1f, 0x10, 0x1e, 0x01, 0x01, 0x11, 0x0e}};
    for (const auto & g : font)
      if (g[0] == uint8_t(c)) return g + 1;
    return nullptr;
  }
  static void draw_label(cv::Mat & img, const std::string & text, int x, int y, int c0, int c1, int c2)
  {
    constexpr int S = 3;
    for (size_t k = 0; k < text.size(); k++) {
      const uint8_t * g = glyph(text[k]);
      if (!g) continue;
      for (int r = 0; r < 7; r++)
        for (int c = 0; c < 5; c++) {
          if (!((g[r] >> (4 - c)) & 1)) continue;
          for (int dy = 0; dy < S; dy++)
            for (int dx = 0; dx < S; dx++) {
              const int px = x + int(k) * 6 * S + c * S + dx, py = y - 7 * S + r * S + dy;
              if (px < 0 || py < 0 || px >= img.cols || py >= img.rows) continue;
              uint8_t * p = img.data + (size_t(py) * img.cols + px) * 3;
              p[0] = uint8_t(c0); p[1] = uint8_t(c1); p[2] = uint8_t(c2);
            }
        }
    }
  }

  irmv_engine * engine_ = nullptr;
  cv::Size src_image_size_;
  bool enable_profiling_ = false;
  uint8_t * src_image_buffer_ = nullptr;
  mutable cv::Mat rotated_;
  mutable bool rotated_valid_ = false;
  std::vector<irmv_det> dets_;
  int n_last_ = 0;
};
}  // namespace irmv_detection
