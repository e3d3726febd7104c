// ROS2 component shell around IrmDetectorCore: the node surface of the reference (reference src/irm_detector.cpp:25-78
// constructor, :122-174 parameters, :176-245 callback -> /detector/armors, :372-403 live parameter updates), with the
// TensorRT engines replaced by the HIP engine.  NOT compiled in the build image (no ROS2 there): see CMakeLists.txt.
// Frames arrive through `image_buffers()` + `on_frame(slot, stamp)` from whatever camera driver the deployment uses
// (the reference's VirtualCamera / MVCamera, src/camera.cpp, src/mv_camera.cpp, are out of scope: SURVEY section 2).
#include <auto_aim_interfaces/msg/armors.hpp>
#include <rclcpp/rclcpp.hpp>
#include <rclcpp_components/register_node_macro.hpp>

#include "irmv_detection/irm_detector_core.hpp"

namespace irmv_detection
{
class IrmDetector
{
public:
  explicit IrmDetector(const rclcpp::NodeOptions & options)
  {
    node_ = std::make_shared<rclcpp::Node>("irmv_detector", options);
    IrmDetectorCore::Params p;
    p.profiling = node_->declare_parameter("profiling", false);
    const auto sz = node_->declare_parameter("image_input_size", std::vector<int64_t>{1280, 1024});
    p.image_input_size = cv::Size(int(sz[0]), int(sz[1]));
    p.binary_threshold = int(node_->declare_parameter("binary_threshold", 150));
    p.light_min_ratio = node_->declare_parameter("light.min_ratio", 0.1);
    p.light_max_ratio = node_->declare_parameter("light.max_ratio", 0.4);
    p.light_max_angle = node_->declare_parameter("light.max_angle", 40.0);
    p.armor_min_small_center_distance = node_->declare_parameter("armor.min_small_center_distance", 0.8);
    p.armor_max_small_center_distance = node_->declare_parameter("armor.max_small_center_distance", 3.2);
    p.armor_min_large_center_distance = node_->declare_parameter("armor.min_large_center_distance", 3.2);
    p.armor_max_large_center_distance = node_->declare_parameter("armor.max_large_center_distance", 5.5);
    p.device = int(node_->declare_parameter("device", 0));
    const std::string model = node_->declare_parameter("model_path", std::string("models/yolov7.onnx"));
    const auto k = node_->declare_parameter("camera_matrix", std::vector<double>{957.669211, 0, 345.943891, 0, 969.127115, 284.057302, 0, 0, 1});
    const auto d = node_->declare_parameter("distortion", std::vector<double>{-0.405274, 0.126058, -0.026939, -0.006503, 0.0});
    std::array<double, 9> K{};
    for (size_t i = 0; i < 9 && i < k.size(); i++) K[i] = k[i];
    core_ = std::make_unique<IrmDetectorCore>(model, K, d, p);
    armors_pub_ = node_->create_publisher<auto_aim_interfaces::msg::Armors>("/detector/armors", rclcpp::SensorDataQoS());
    param_handle_ = node_->add_on_set_parameters_callback([this](const std::vector<rclcpp::Parameter> & ps) {
      rcl_interfaces::msg::SetParametersResult r;
      r.successful = true;
      r.reason = "success";
      for (const auto & q : ps)
        if (q.get_type() == rclcpp::ParameterType::PARAMETER_INTEGER) core_->set_parameter(q.get_name(), double(q.as_int()));
        else if (q.get_type() == rclcpp::ParameterType::PARAMETER_DOUBLE) core_->set_parameter(q.get_name(), q.as_double());
      return r;
    });
  }

  rclcpp::node_interfaces::NodeBaseInterface::SharedPtr get_node_base_interface() const { return node_->get_node_base_interface(); }
  std::array<uint8_t *, 3> image_buffers() const { return core_->image_buffers(); }

  // Camera::CameraCallback body (src/irm_detector.cpp:176-245)
  void on_frame(int slot, std::chrono::time_point<std::chrono::system_clock> stamp)
  {
    const auto r = core_->message_callback(StampedFrame{stamp, slot});
    auto_aim_interfaces::msg::Armors msg;
    msg.header.stamp = rclcpp::Time(r.armors_msg.header.stamp_ns, RCL_ROS_TIME);
    msg.header.frame_id = r.armors_msg.header.frame_id;
    for (const auto & a : r.armors_msg.armors) {
      auto_aim_interfaces::msg::Armor m;
      m.pose.position.x = a.pose.position.x; m.pose.position.y = a.pose.position.y; m.pose.position.z = a.pose.position.z;
      m.pose.orientation.x = a.pose.orientation.x; m.pose.orientation.y = a.pose.orientation.y;
      m.pose.orientation.z = a.pose.orientation.z; m.pose.orientation.w = a.pose.orientation.w;
      m.distance_to_image_center = a.distance_to_image_center;
      msg.armors.emplace_back(m);
    }
    armors_pub_->publish(msg);
  }

private:
  rclcpp::Node::SharedPtr node_;
  std::unique_ptr<IrmDetectorCore> core_;
  rclcpp::Publisher<auto_aim_interfaces::msg::Armors>::SharedPtr armors_pub_;
  rclcpp::node_interfaces::OnSetParametersCallbackHandle::SharedPtr param_handle_;
};
}  // namespace irmv_detection

RCLCPP_COMPONENTS_REGISTER_NODE(irmv_detection::IrmDetector)
