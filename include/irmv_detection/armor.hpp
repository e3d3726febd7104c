// Value types that cross the hot-path API: ArmorClass, ArmorSize, Light, Armor.
// A consumer written against the reference's include/irmv_detection/armor.hpp (ArmorClass :7,
// ArmorSize :9, Light :11-53, Armor :55-77) compiles against this header unchanged: same type,
// member and method names, same meaning.  Written for this build; only the public surface is shared.
#pragma once

#include <array>
#include <cmath>
#include <utility>

#include "irmv_detection/cv_compat.hpp"

namespace irmv_detection
{
// Class ids are the network's class indices: the order below is the label order of the model.
#define IRMV_ARMOR_CLASS_LIST(X) X(B1) X(B2) X(B3) X(B4) X(B5) X(BO) X(BS) X(R1) X(R2) X(R3) X(R4) X(R5) X(RO) X(RS)

#define IRMV_AS_ENUMERATOR(name) name,
enum class ArmorClass { IRMV_ARMOR_CLASS_LIST(IRMV_AS_ENUMERATOR) UNKNOWN };
#undef IRMV_AS_ENUMERATOR

enum class ArmorSize { SMALL, LARGE, UNKNOWN };

// Name of a class id (the reference gets this from the vendored magic_enum reflection header).
inline const char * armor_class_name(ArmorClass c)
{
#define IRMV_AS_STRING(name) #name,
  static constexpr const char * kNames[] = {IRMV_ARMOR_CLASS_LIST(IRMV_AS_STRING) "UNKNOWN"};
#undef IRMV_AS_STRING
  constexpr int kCount = static_cast<int>(sizeof(kNames) / sizeof(kNames[0]));
  const int id = static_cast<int>(c);
  return kNames[(id < 0 || id >= kCount) ? kCount - 1 : id];
}

namespace detail
{
inline double tilt_from_vertical_deg(const cv::Point2f & a, const cv::Point2f & b)
{
  return std::atan2(std::fabs(a.x - b.x), std::fabs(a.y - b.y)) * (180.0 / 3.14159265358979323846);
}
}  // namespace detail

// A light bar.  `top` / `bottom` are the mid-points of the rectangle's two short edges (the upper and the lower
// one in image coordinates), `length` their distance, `width` the upper edge's length, `tilt_angle` the bar's
// inclination from the vertical in degrees.
struct Light : public cv::RotatedRect
{
  cv::Point2f top{};
  cv::Point2f bottom{};
  double tilt_angle = 0.0;
  double length = 0.0;
  double width = 0.0;

  Light() = default;

  // from the minimum-area rectangle of a contour (the classical extraction)
  explicit Light(const cv::RotatedRect & box) : cv::RotatedRect(box)
  {
    std::array<cv::Point2f, 4> quad;
    box.points(quad.data());
    // order the four corners by height: a stable insertion sort on y (ties keep OpenCV's corner order)
    for (std::size_t i = 1; i < quad.size(); ++i)
      for (std::size_t k = i; k > 0 && quad[k].y < quad[k - 1].y; --k) std::swap(quad[k], quad[k - 1]);
    top = (quad[0] + quad[1]) / 2;
    bottom = (quad[2] + quad[3]) / 2;
    width = cv::norm(quad[0] - quad[1]);
    length = cv::norm(top - bottom);
    tilt_angle = detail::tilt_from_vertical_deg(top, bottom);
  }

  // from two keypoints (the GPU keypoint head yields the end points directly; there is no rectangle)
  Light(const cv::Point2f & upper, const cv::Point2f & lower) : top(upper), bottom(lower)
  {
    center = (upper + lower) / 2;
    length = cv::norm(upper - lower);
    tilt_angle = detail::tilt_from_vertical_deg(upper, lower);
  }

  // the reference's gate: width / length inside (min_ratio, max_ratio) and inclination below max_angle degrees
  bool is_light(float min_ratio, float max_ratio, float max_angle) const
  {
    const double aspect = width / length;
    if (!(aspect > min_ratio)) return false;
    if (!(aspect < max_ratio)) return false;
    return tilt_angle < max_angle;
  }

  // ROI coordinates -> frame coordinates
  void offset_bbox(float min_x, float min_y)
  {
    const cv::Point2f shift(min_x, min_y);
    for (cv::Point2f * pt : {&center, &top, &bottom}) *pt = *pt + shift;
  }
};

// Two lights, the left one first.
struct Armor
{
  cv::Point2f center{};
  Light left_light{};
  Light right_light{};
  ArmorClass armor_class = ArmorClass::UNKNOWN;
  ArmorSize size = ArmorSize::SMALL;
  float confidence = 0.0f;

  Armor() = default;
  Armor(const Light & l1, const Light & l2) : left_light(l1), right_light(l2)
  {
    if (!(l1.center.x < l2.center.x)) std::swap(left_light, right_light);
    center = (left_light.center + right_light.center) / 2;
  }
};
}  // namespace irmv_detection
