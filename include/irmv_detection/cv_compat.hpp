// Minimal stand-ins for the handful of OpenCV value types that appear in the
// reference's public headers (cv::Size, cv::Point2f, cv::Point3d, cv::Scalar,
// cv::RotatedRect, cv::Mat), used ONLY when OpenCV itself is not installed.
// With OpenCV present the real types are used and this header adds nothing.
#pragma once

#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define IRMV_HAVE_OPENCV 1
#else
#define IRMV_HAVE_OPENCV 0

#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8UC3 16
#define CV_64F 6

namespace cv
{
constexpr double CV_PI_D = 3.14159265358979323846;

struct Size
{
  int width = 0, height = 0;
  Size() = default;
  Size(int w, int h) : width(w), height(h) {}
};

template <typename T>
struct Point_
{
  T x{}, y{};
  Point_() = default;
  Point_(T x_, T y_) : x(x_), y(y_) {}
  Point_ operator+(const Point_ & o) const { return {T(x + o.x), T(y + o.y)}; }
  Point_ operator-(const Point_ & o) const { return {T(x - o.x), T(y - o.y)}; }
  Point_ operator/(double d) const { return {T(x / d), T(y / d)}; }
  Point_ operator*(double d) const { return {T(x * d), T(y * d)}; }
};
using Point2f = Point_<float>;
using Point = Point_<int>;

struct Point3d
{
  double x = 0, y = 0, z = 0;
  Point3d() = default;
  Point3d(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
};

inline double norm(const Point2f & p) { return std::sqrt(double(p.x) * p.x + double(p.y) * p.y); }

struct Scalar
{
  double v[4] = {0, 0, 0, 0};
  Scalar() = default;
  Scalar(double a, double b, double c, double d = 0) : v{a, b, c, d} {}
  double operator[](int i) const { return v[i]; }
};

struct Size2f
{
  float width = 0, height = 0;
};

// centre / size / angle (degrees, clockwise) rectangle
struct RotatedRect
{
  Point2f center;
  Size2f size;
  float angle = 0;
  RotatedRect() = default;
  RotatedRect(Point2f c, Size2f s, float a) : center(c), size(s), angle(a) {}
  void points(Point2f pts[]) const
  {
    const double a = angle * CV_PI_D / 180.0;
    const float b = float(std::cos(a)) * 0.5f, s = float(std::sin(a)) * 0.5f;
    pts[0] = {center.x - s * size.height - b * size.width, center.y + b * size.height - s * size.width};
    pts[1] = {center.x + s * size.height - b * size.width, center.y - b * size.height - s * size.width};
    pts[2] = {2 * center.x - pts[0].x, 2 * center.y - pts[0].y};
    pts[3] = {2 * center.x - pts[1].x, 2 * center.y - pts[1].y};
  }
};

// Dense 2-D array: 8UC3 images and small CV_64F matrices, owning or aliasing.
class Mat
{
public:
  int rows = 0, cols = 0;
  uint8_t * data = nullptr;
  Mat() = default;
  Mat(Size s, int type, void * external) : rows(s.height), cols(s.width), data(static_cast<uint8_t *>(external)), type_(type) {}
  Mat(int r, int c, int type) { create(r, c, type); }
  void create(int r, int c, int type)
  {
    rows = r; cols = c; type_ = type;
    store_ = std::shared_ptr<uint8_t>(new uint8_t[total() * elemSize()](), std::default_delete<uint8_t[]>());
    data = store_.get();
  }
  static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
  size_t total() const { return size_t(rows) * cols; }
  size_t elemSize() const { return type_ == CV_64F ? 8 : 3; }
  int type() const { return type_; }
  bool empty() const { return data == nullptr; }
  Mat clone() const
  {
    Mat m(rows, cols, type_);
    if (data) std::memcpy(m.data, data, total() * elemSize());
    return m;
  }
  template <typename T> T & at(int i) { return reinterpret_cast<T *>(data)[i]; }
  template <typename T> const T & at(int i) const { return reinterpret_cast<const T *>(data)[i]; }
  template <typename T> T & at(int r, int c) { return reinterpret_cast<T *>(data)[size_t(r) * cols + c]; }
  template <typename T> const T & at(int r, int c) const { return reinterpret_cast<const T *>(data)[size_t(r) * cols + c]; }

private:
  int type_ = CV_8UC3;
  std::shared_ptr<uint8_t> store_;
};
}  // namespace cv
#endif  // OpenCV absent
