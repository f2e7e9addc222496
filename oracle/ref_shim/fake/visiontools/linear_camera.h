// TEST INFRASTRUCTURE -- not part of the product.
// Stand-in for VisionTools::LinearCamera (third party, absent from /root/reference): the pinhole maps the way SURVEY.md A.5 and the
// oracle (vision.c in_frame / warp_f) state them -- map(p) = f p + c, unmap(uv) = (uv - c) / f, isInFrame(uv, b): b <= u < W - b and
// b <= v < H - b -- and the pyramid scalings zeroFromPyr_* (x * 2^level).  The reference's own StereoCamera derives from it.
#pragma once
#include <Eigen/Core>
#include <opencv2/core/core.hpp>
#include <visiontools/abstract_camera.h>
namespace VisionTools {
using namespace Eigen;
class LinearCamera : public AbstractCamera {
 public:
  LinearCamera() : focal_length_(1), principle_point_(0, 0), size_(0, 0) {}
  LinearCamera(const AbstractCamera &) : focal_length_(1), principle_point_(0, 0), size_(0, 0) {}      // (unused by the pinned paths)
  LinearCamera(const LinearCamera &o) : AbstractCamera(o), focal_length_(o.focal_length_), principle_point_(o.principle_point_), size_(o.size_) {}
  LinearCamera(const Matrix3d K, const cv::Size &size) : focal_length_(K(0, 0)), principle_point_(K(0, 2), K(1, 2)), size_(size) {}
  LinearCamera(const double &focal_length, const Vector2d &principle_point, const cv::Size &size)
      : focal_length_(focal_length), principle_point_(principle_point), size_(size) {}
  Vector2d map(const Vector2d &p) const { return Vector2d(focal_length_ * p[0] + principle_point_[0], focal_length_ * p[1] + principle_point_[1]); }
  Vector2d unmap(const Vector2d &uv) const { return Vector2d((uv[0] - principle_point_[0]) / focal_length_, (uv[1] - principle_point_[1]) / focal_length_); }
  bool isInFrame(const Vector2i &uv, int border = 0) const {
    return uv[0] >= border && uv[1] >= border && uv[0] < size_.width - border && uv[1] < size_.height - border;
  }
  int width() const { return size_.width; }
  int height() const { return size_.height; }
  const double &focal_length() const { return focal_length_; }
  const Vector2d &principal_point() const { return principle_point_; }
  const cv::Size &image_size() const { return size_; }
 protected:
  double focal_length_;
  Vector2d principle_point_;
  cv::Size size_;
};
inline int zeroFromPyr_i(int pyr, int level) { return pyr << level; }
inline int pyrFromZero_i(int zero, int level) { return zero >> level; }
inline Vector2i zeroFromPyr_2i(const Vector2i &pyr, int level) { return Vector2i(zeroFromPyr_i(pyr[0], level), zeroFromPyr_i(pyr[1], level)); }
inline Vector2d pyrFromZero_2d(const Vector2d &zero, int level) { return Vector2d(pyrFromZero_d(zero[0], level), pyrFromZero_d(zero[1], level)); }
inline double zeroFromPyr_d(double pyr, int level) { return pyr * (double)(1 << level); }
inline Vector2d zeroFromPyr_2d(const Vector2d &pyr, int level) { return Vector2d(zeroFromPyr_d(pyr[0], level), zeroFromPyr_d(pyr[1], level)); }
inline Vector3d zeroFromPyr_3d(const Vector3d &pyr, int level) {
  return Vector3d(zeroFromPyr_d(pyr[0], level), zeroFromPyr_d(pyr[1], level), zeroFromPyr_d(pyr[2], level));
}
}  // namespace VisionTools
