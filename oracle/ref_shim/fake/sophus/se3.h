// TEST INFRASTRUCTURE -- not part of the product.
// Stand-in for Sophus::SE3 as the dense-tracking slice of the reference uses it (operator* on points and poses, exp, inverse, matrix).  The
// arithmetic is HANDED to the oracle's pose helpers (oracle/svs_math.h: the oracle's statement of Sophus a621ff): what the pin compiles
// from the reference and checks is the tracker's own loop, not the Lie-group library.
#pragma once
#include <Eigen/Core>
#include "../../../svs_math.h"
namespace Sophus {
class SE3 {
 public:
  static const int DoF = 6;
  double T[12];      // 3x4 row-major
  SE3() { for (int i = 0; i < 12; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0; }
  explicit SE3(const double *t) { for (int i = 0; i < 12; ++i) T[i] = t[i]; }
  Eigen::Vector3d operator*(const Eigen::Vector3d &p) const { Eigen::Vector3d r; pose_act(T, p.v, r.v); return r; }
  SE3 operator*(const SE3 &o) const { SE3 r; pose_mul(T, o.T, r.T); return r; }
  static SE3 exp(const Eigen::Matrix<double, 6, 1> &x) { SE3 r; se3_exp(x.v, r.T); return r; }
  Eigen::Vector3d translation() const { return Eigen::Vector3d(T[3], T[7], T[11]); }
  SE3 inverse() const { SE3 r; pose_inv(T, r.T); return r; }
  Eigen::Matrix3d rotation_matrix() const { Eigen::Matrix3d R; pose_R(T, R.v); return R; }
  Eigen::Matrix<double, 6, 6> Adj() const { Eigen::Matrix<double, 6, 6> A; se3_adj(T, A.v); return A; }
  static Eigen::Matrix<double, 6, 6> d_lieBracketab_by_d_a(const Eigen::Matrix<double, 6, 1> &b) { Eigen::Matrix<double, 6, 6> M; se3_dlie(b.v, M.v); return M; }
  Eigen::Matrix<double, 6, 1> log() const { Eigen::Matrix<double, 6, 1> x; se3_log(T, x.v); return x; }
  Eigen::Matrix4d matrix() const {
    Eigen::Matrix4d m;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) m(i, j) = T[4 * i + j];
    m(3, 0) = 0; m(3, 1) = 0; m(3, 2) = 0; m(3, 3) = 1;
    return m;
  }
};
struct SO3 {
  static Eigen::Matrix3d hat(const Eigen::Vector3d &v) { Eigen::Matrix3d H; hat3(v.v, H.v); return H; }
};
}  // namespace Sophus
