// irmv_detection value types crossing the hot-path API -- drop-in for the
// reference's include/irmv_detection/armor.hpp (same names, members and
// semantics: ArmorClass :7, ArmorSize :9, Light :11-53, Armor :55-77).
#pragma once

#include <algorithm>
#include <array>
#include <cmath>

#include "irmv_detection/cv_compat.hpp"

namespace irmv_detection
{
enum class ArmorClass { B1, B2, B3, B4, B5, BO, BS, R1, R2, R3, R4, R5, RO, RS, UNKNOWN };

enum class ArmorSize { SMALL, LARGE, UNKNOWN };

// 15-entry name table in place of the vendored magic_enum reflection header
inline const char * armor_class_name(ArmorClass c)
{
  static const char * const names[] = {"B1", "B2", "B3", "B4", "B5", "BO", "BS", "R1",
                                       "R2", "R3", "R4", "R5", "RO", "RS", "UNKNOWN"};
  const int i = static_cast<int>(c);
  return names[(i >= 0 && i < 15) ? i : 14];
}

// One light bar: a rotated rectangle plus the mid-points of its short edges.
struct Light : public cv::RotatedRect
{
  Light() = default;
  explicit Light(const cv::RotatedRect & box) : cv::RotatedRect(box)
  {
    std::array<cv::Point2f, 4> corner;
    box.points(corner.data());
    std::sort(corner.begin(), corner.end(), [](const cv::Point2f & l, const cv::Point2f & r) { return l.y < r.y; });
    top = (corner[0] + corner[1]) / 2;
    bottom = (corner[2] + corner[3]) / 2;
    length = cv::norm(top - bottom);
    width = cv::norm(corner[0] - corner[1]);
    tilt_angle = std::atan2(std::abs(top.x - bottom.x), std::abs(top.y - bottom.y)) * 180.0 / 3.14159265358979323846;
  }
  // keypoint constructor: the GPU keypoint head yields top/bottom directly
  Light(const cv::Point2f & top_pt, const cv::Point2f & bottom_pt) : top(top_pt), bottom(bottom_pt)
  {
    center = (top + bottom) / 2;
    length = cv::norm(top - bottom);
    tilt_angle = std::atan2(std::abs(top.x - bottom.x), std::abs(top.y - bottom.y)) * 180.0 / 3.14159265358979323846;
  }

  bool is_light(float min_ratio, float max_ratio, float max_angle) const
  {
    const double ratio = width / length;
    return min_ratio < ratio && ratio < max_ratio && tilt_angle < max_angle;
  }

  void offset_bbox(float min_x, float min_y)
  {
    center.x += min_x; center.y += min_y;
    top.x += min_x; top.y += min_y;
    bottom.x += min_x; bottom.y += min_y;
  }

  cv::Point2f top;
  cv::Point2f bottom;
  double length = 0;
  double width = 0;
  double tilt_angle = 0;
};

struct Armor
{
  Armor() = default;
  Armor(const Light & l1, const Light & l2)
  {
    const bool first_is_left = l1.center.x < l2.center.x;
    left_light = first_is_left ? l1 : l2;
    right_light = first_is_left ? l2 : l1;
    center = (left_light.center + right_light.center) / 2;
  }

  Light left_light;
  Light right_light;
  ArmorSize size = ArmorSize::SMALL;
  ArmorClass armor_class = ArmorClass::UNKNOWN;
  float confidence = 0;
  cv::Point2f center;
};
}  // namespace irmv_detection
