// irmv_detection::IrmDetectorCore -- the ROS-free body of the reference's detector node
// (reference src/irm_detector.cpp): what IrmDetector::IrmDetector (:25-78) builds and what
// IrmDetector::message_callback (:176-245) computes per frame, with every rclcpp / cv_bridge /
// tf2 / auto_aim_interfaces type replaced by a plain struct of the same fields, so that the
// hot path of the node compiles and is tested without a ROS2 installation.  The optional ROS2
// target (ros2/irm_detector_node.cpp, built only where ament_cmake is found) is a thin shell
// around this class: it copies ArmorsMsg into auto_aim_interfaces::msg::Armors and publishes.
//
//   frame in slot `id`  ->  YoloEngine::detect()                        (:181)
//                       ->  extract_armors(get_rotated_image(), bboxes) (:183, :292-355)  [GPU]
//                       ->  per armor: solvePnP (IPPE, small armor)      (:204-209)        [GPU, fused]
//                           Rodrigues -> rotation matrix -> quaternion   (:218-226)        [GPU, fused]
//                           distance_to_image_center                     (:229)
//                       ->  ArmorsMsg                                    (:245 publishes it)
//
// Two sources of the four armor points, like the engine (irmv_hip.h, point_source): a pose-style
// model carries them in its keypoint head (then every detection is an armor); a bbox-only model
// gets them from the reference's classical light extraction, run on the GPU.  Either way the pose
// arrives with the detection: no per-armor host round trip.
#pragma once

#include <array>
#include <chrono>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "irmv_detection/armor.hpp"
#include "irmv_detection/pnp_solver.hpp"
#include "irmv_detection/yolo_engine.hpp"

namespace irmv_detection
{
// geometry_msgs::msg::Pose
struct PoseMsg
{
  struct { double x = 0, y = 0, z = 0; } position;
  struct { double x = 0, y = 0, z = 0, w = 1; } orientation;
};

// auto_aim_interfaces::msg::Armor (string number, string type, float32 distance_to_image_center,
// geometry_msgs/Pose pose).  Like the reference (:211-230) only the pose and the distance are filled;
// `number` / `type` stay empty there too.  armor_class / size ride along for consumers that want them.
struct ArmorMsg
{
  std::string number;
  std::string type;
  float distance_to_image_center = 0.f;
  PoseMsg pose;
  ArmorClass armor_class = ArmorClass::UNKNOWN;   // extension (not a field of the ROS message)
  ArmorSize size = ArmorSize::SMALL;              // extension
};

// auto_aim_interfaces::msg::Armors: std_msgs/Header + Armor[]
struct ArmorsMsg
{
  struct
  {
    int64_t stamp_ns = 0;                          // image.time_stamp.time_since_epoch() (:194)
    std::string frame_id = "camera_optical_frame"; // :197
  } header;
  std::vector<ArmorMsg> armors;
};

// Camera::StampedImage without the cv::Mat (the pixels already sit in engine slot `id`,
// reference include/irmv_detection/camera.hpp:27-32, src/camera.cpp:24-29)
struct StampedFrame
{
  std::chrono::time_point<std::chrono::system_clock> time_stamp{};
  int id = 0;
};

class IrmDetectorCore
{
public:
  // The node's parameters that reach the hot path (src/irm_detector.cpp:122-174)
  struct Params
  {
    cv::Size image_input_size = cv::Size(1280, 1024);   // camera frame size (:140-144)
    bool profiling = false;
    int binary_threshold = 150;
    double light_min_ratio = 0.1, light_max_ratio = 0.4, light_max_angle = 40.0;
    double armor_min_small_center_distance = 0.8, armor_max_small_center_distance = 3.2;
    double armor_min_large_center_distance = 3.2, armor_max_large_center_distance = 5.5;
    int device = -1;                                     // HIP device; -1: IRMV_DEVICE or 0
  };

  // What one frame produced, beyond the message: kept for debug publishers and tests.
  struct FrameResult
  {
    ArmorsMsg armors_msg;
    std::vector<YoloEngine::bbox> bboxes;
    std::vector<Armor> armors;          // one per message entry, same order
    std::vector<irmv_det> poses;        // rvec / tvec / quaternion of each, as computed on the GPU
    double inference_latency_ms = 0;    // YoloEngine::get_profiling_time() (:251)
  };

  // Three engines, one per TripleBuffer slot, like the node (:33-38); K, D from camera_info (:52).
  IrmDetectorCore(const std::string & model_path, const std::array<double, 9> & k, const std::vector<double> & d, const Params & p)
  : params_(p)
  {
    for (auto & e : yolo_engines_) {
      e = std::make_unique<YoloEngine>(model_path, p.image_input_size, p.profiling, p.device, false);
      push_params(*e);
    }
    yolo_engines_[0]->warm_up();   // the tile choices are shared by the three engines: one warm-up tunes for all
    for (size_t i = 1; i < yolo_engines_.size(); i++) yolo_engines_[i]->warm_up();
    pnp_solver_ = std::make_unique<PnPSolver>(k, d, p.device);
  }

  // Camera::Config::image_buffers (:68-72): where the producer deposits frames
  std::array<uint8_t *, 3> image_buffers() const
  {
    return {yolo_engines_[0]->get_src_image_buffer(), yolo_engines_[1]->get_src_image_buffer(), yolo_engines_[2]->get_src_image_buffer()};
  }

  YoloEngine & engine(int id) { return *yolo_engines_[size_t(id)]; }
  const PnPSolver & pnp_solver() const { return *pnp_solver_; }

  // IrmDetector::message_callback without the publishers (:176-245)
  FrameResult message_callback(const StampedFrame & image)
  {
    FrameResult out;
    YoloEngine & eng = *yolo_engines_[size_t(image.id)];
    out.bboxes = eng.detect();
    out.armors_msg.header.stamp_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(image.time_stamp.time_since_epoch()).count();

    std::vector<Armor> armors;
    std::vector<irmv_det> poses;
    if (eng.has_keypoint_head()) {
      // pose-style model: each detection carries its four points and, already, its pose
      int n = 0;
      const irmv_det * dets = eng.last_detections(&n);
      for (int i = 0; i < n; i++) {
        const irmv_det & d = dets[i];
        if (d.armor_valid != 1) continue;
        Armor a(Light(cv::Point2f(d.kpts[2], d.kpts[3]), cv::Point2f(d.kpts[0], d.kpts[1])),
                Light(cv::Point2f(d.kpts[4], d.kpts[5]), cv::Point2f(d.kpts[6], d.kpts[7])));
        a.armor_class = static_cast<ArmorClass>(d.class_id);
        a.confidence = d.score;
        a.size = d.armor_size == IRMV_ARMOR_LARGE ? ArmorSize::LARGE : ArmorSize::SMALL;
        armors.push_back(a);
        poses.push_back(d);
      }
    } else {
      armors = eng.extract_armors(out.bboxes, &poses);   // :183, on the GPU, poses included
    }

    for (size_t i = 0; i < armors.size(); i++) {
      const irmv_det & d = poses[i];
      if (!d.pnp_ok) continue;                             // `if (!pnp_solver_->solvePnP(...)) continue;` (:207-209)
      ArmorMsg m;
      m.pose.position.x = d.tvec[0];                       // :214-216
      m.pose.position.y = d.tvec[1];
      m.pose.position.z = d.tvec[2];
      m.pose.orientation.x = d.quat[0];                    // Rodrigues -> tf2::Matrix3x3::getRotation (:218-226)
      m.pose.orientation.y = d.quat[1];
      m.pose.orientation.z = d.quat[2];
      m.pose.orientation.w = d.quat[3];
      m.distance_to_image_center = pnp_solver_->calculateDistanceToCenter(armors[i].center);   // :229
      m.armor_class = armors[i].armor_class;
      m.size = armors[i].size;
      out.armors_msg.armors.push_back(m);
      out.armors.push_back(armors[i]);
      out.poses.push_back(d);
    }
    out.inference_latency_ms = eng.get_profiling_time();
    return out;
  }

  // IrmDetector::param_event_callback (:372-403): returns false for a name that is not a hot-path parameter
  bool set_parameter(const std::string & name, double value)
  {
    if (name == "binary_threshold") params_.binary_threshold = int(value);
    else if (name == "light.min_ratio") params_.light_min_ratio = value;
    else if (name == "light.max_ratio") params_.light_max_ratio = value;
    else if (name == "light.max_angle") params_.light_max_angle = value;
    else if (name == "armor.min_small_center_distance") params_.armor_min_small_center_distance = value;
    else if (name == "armor.max_small_center_distance") params_.armor_max_small_center_distance = value;
    else if (name == "armor.min_large_center_distance") params_.armor_min_large_center_distance = value;
    else if (name == "armor.max_large_center_distance") params_.armor_max_large_center_distance = value;
    else return false;
    for (auto & e : yolo_engines_) push_params(*e);
    return true;
  }

  const Params & params() const { return params_; }

private:
  void push_params(YoloEngine & e) const
  {
    e.set_extract_params(params_.binary_threshold, float(params_.light_min_ratio), float(params_.light_max_ratio), float(params_.light_max_angle),
                         params_.armor_min_small_center_distance, params_.armor_max_small_center_distance,
                         params_.armor_min_large_center_distance, params_.armor_max_large_center_distance);
  }

  Params params_;
  std::array<std::unique_ptr<YoloEngine>, 3> yolo_engines_;
  std::unique_ptr<PnPSolver> pnp_solver_;
};
}  // namespace irmv_detection
