// irmv_detection::PnPSolver on MI355X -- header-only facade over the C ABI with
// the reference's interface (reference include/irmv_detection/pnp_solver.hpp:12-38,
// src/pnp_solver.cpp): PnPSolver(K[9], D), solvePnP(armor, rvec, tvec) -> bool,
// calculateDistanceToCenter(point).  IPPE runs in a HIP kernel, one lane per armor.
#pragma once

#include <array>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "irmv_detection/armor.hpp"
#include "irmv_detection/cv_compat.hpp"
#include "irmv_hip.h"

namespace irmv_detection
{
class PnPSolver
{
public:
  // The reference's constructor (pnp_solver.hpp:15-17); `device`: HIP ordinal, -1 = IRMV_DEVICE or 0
  PnPSolver(const std::array<double, 9> & camera_matrix, const std::vector<double> & distortion_coefficients, int device = -1)
  {
    if (device < 0) {
      const char * v = std::getenv("IRMV_DEVICE");
      device = v ? std::atoi(v) : 0;
    }
    double d[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < 5 && i < distortion_coefficients.size(); i++) d[i] = distortion_coefficients[i];
    cx_ = camera_matrix[2];
    cy_ = camera_matrix[5];
    if (irmv_pnp_create(device, camera_matrix.data(), d, &pnp_) != IRMV_OK)
      throw std::runtime_error(std::string("PnPSolver: ") + irmv_last_error());
  }
  ~PnPSolver() { irmv_pnp_destroy(pnp_); }
  PnPSolver(const PnPSolver &) = delete;
  PnPSolver & operator=(const PnPSolver &) = delete;

  // Image points in the order the reference feeds cv::solvePnP (src/pnp_solver.cpp:41-44);
  // always the SMALL armor model (:47-48).  rvec / tvec come back as 3x1 CV_64F.
  bool solvePnP(const Armor & armor, cv::Mat & rvec, cv::Mat & tvec) const
  {
    const float pts[8] = {armor.left_light.bottom.x, armor.left_light.bottom.y, armor.left_light.top.x, armor.left_light.top.y,
                          armor.right_light.top.x, armor.right_light.top.y, armor.right_light.bottom.x, armor.right_light.bottom.y};
    double r[3], t[3];
    int32_t ok = 0;
    if (irmv_pnp_solve(pnp_, pts, 1, IRMV_ARMOR_SMALL, r, t, &ok) != IRMV_OK)
      throw std::runtime_error(std::string("PnPSolver::solvePnP: ") + irmv_last_error());
    rvec = cv::Mat(3, 1, CV_64F);
    tvec = cv::Mat(3, 1, CV_64F);
    for (int i = 0; i < 3; i++) {
      rvec.at<double>(i) = r[i];
      tvec.at<double>(i) = t[i];
    }
    return ok != 0;
  }

  // |p - (cx, cy)| with the true principal point.  (The reference reads K with
  // at<float>() from a CV_64F matrix, src/pnp_solver.cpp:56-57, and therefore uses
  // (0.0, 7.6e12) with the shipped YAML -- SURVEY.md Appendix E.1.)
  float calculateDistanceToCenter(const cv::Point2f & image_point) const
  {
    const float dx = image_point.x - float(cx_), dy = image_point.y - float(cy_);
    return std::sqrt(dx * dx + dy * dy);
  }

private:
  irmv_pnp * pnp_ = nullptr;
  double cx_ = 0, cy_ = 0;
};
}  // namespace irmv_detection
