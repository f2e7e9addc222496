/* svs_math.h -- small fixed-size f64 helpers for the oracle (test infrastructure only).
 * Pose layout: 3x4 row-major, T[4*i+j], translation T[4*i+3].
 * SE3 semantics follow Sophus a621ff as used by the reference:
 * tangent (upsilon, omega), translation part first, left-multiplicative update. */
#ifndef SVS_MATH_H
#define SVS_MATH_H
#include <math.h>
#include <string.h>

static inline void m3_mul(const double *A, const double *B, double *C) { /* 3x3 row-major */
  double t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, t, sizeof t);
}
static inline void m3_vec(const double *A, const double *v, double *r) {
  double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
  double t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
  double t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  r[0] = t0; r[1] = t1; r[2] = t2;
}
static inline void hat3(const double *v, double *H) {
  H[0] = 0; H[1] = -v[2]; H[2] = v[1];
  H[3] = v[2]; H[4] = 0; H[5] = -v[0];
  H[6] = -v[1]; H[7] = v[0]; H[8] = 0;
}
static inline void pose_R(const double *T, double *R) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = T[4 * i + j];
}
static inline void pose_t(const double *T, double *t) { t[0] = T[3]; t[1] = T[7]; t[2] = T[11]; }
static inline void pose_set(double *T, const double *R, const double *t) {
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j]; T[4 * i + 3] = t[i]; }
}
static inline void pose_act(const double *T, const double *x, double *y) {
  double a = T[0] * x[0] + T[1] * x[1] + T[2] * x[2] + T[3];
  double b = T[4] * x[0] + T[5] * x[1] + T[6] * x[2] + T[7];
  double c = T[8] * x[0] + T[9] * x[1] + T[10] * x[2] + T[11];
  y[0] = a; y[1] = b; y[2] = c;
}
static inline void pose_mul(const double *A, const double *B, double *C) {
  double t[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j)
      t[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j];
    t[4 * i + 3] += A[4 * i + 3];
  }
  memcpy(C, t, sizeof t);
}
static inline void pose_inv(const double *A, double *B) {
  double t[12];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[4 * i + j] = A[4 * j + i];
  for (int i = 0; i < 3; ++i)
    t[4 * i + 3] = -(t[4 * i] * A[3] + t[4 * i + 1] * A[7] + t[4 * i + 2] * A[11]);
  memcpy(B, t, sizeof t);
}

#define SVS_SMALL_EPS 1e-10

/* SO3::exp (Rodrigues); returns theta */
static inline double so3_exp(const double *w, double *R) {
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double th = sqrt(th2);
  double W[9], W2[9];
  hat3(w, W);
  m3_mul(W, W, W2);
  double a, b;
  if (th < SVS_SMALL_EPS) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
  else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
  for (int i = 0; i < 9; ++i) R[i] = a * W[i] + b * W2[i];
  R[0] += 1.0; R[4] += 1.0; R[8] += 1.0;
  return th;
}
/* SE3::exp(upsilon, omega) = (V(omega) upsilon, exp(omega^)) */
static inline void se3_exp(const double *x, double *T) {
  double R[9], W[9], W2[9], V[9], t[3];
  const double *w = x + 3;
  double th = so3_exp(w, R);
  if (th < SVS_SMALL_EPS) {
    memcpy(V, R, sizeof V);
  } else {
    hat3(w, W);
    m3_mul(W, W, W2);
    double th2 = th * th;
    double a = (1.0 - cos(th)) / th2, b = (th - sin(th)) / (th2 * th);
    for (int i = 0; i < 9; ++i) V[i] = a * W[i] + b * W2[i];
    V[0] += 1.0; V[4] += 1.0; V[8] += 1.0;
  }
  m3_vec(V, x, t);
  pose_set(T, R, t);
}
/* SO3::log via unit quaternion (Sophus logAndTheta) */
static inline double so3_log(const double *R, double *w) {
  /* rotation matrix -> quaternion (w,x,y,z), robust branch */
  double q[4];
  double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double s = sqrt(tr + 1.0) * 2; q[0] = 0.25 * s;
    q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[0] = (R[7] - R[5]) / s;
    q[1] = 0.25 * s; q[2] = (R[1] + R[3]) / s; q[3] = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[0] = (R[2] - R[6]) / s;
    q[1] = (R[1] + R[3]) / s; q[2] = 0.25 * s; q[3] = (R[5] + R[7]) / s;
  } else {
    double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[0] = (R[3] - R[1]) / s;
    q[1] = (R[2] + R[6]) / s; q[2] = (R[5] + R[7]) / s; q[3] = 0.25 * s;
  }
  double n = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  double ww = q[0];
  double two_atan;
  if (n < SVS_SMALL_EPS) {
    two_atan = 2.0 / ww - 2.0 * (n * n) / (ww * ww * ww);
  } else if (fabs(ww) < SVS_SMALL_EPS) {
    two_atan = (ww > 0 ? M_PI : -M_PI) / n;
  } else {
    two_atan = 2.0 * atan(n / ww) / n;
  }
  w[0] = two_atan * q[1]; w[1] = two_atan * q[2]; w[2] = two_atan * q[3];
  return two_atan * n;
}
static inline void se3_log(const double *T, double *x) {
  double R[9], t[3], W[9], W2[9], Vi[9];
  pose_R(T, R); pose_t(T, t);
  double th = so3_log(R, x + 3);
  hat3(x + 3, W);
  m3_mul(W, W, W2);
  double c;
  if (fabs(th) < SVS_SMALL_EPS) c = 1.0 / 12.0;
  else c = (1.0 - th / (2.0 * tan(th / 2.0))) / (th * th);
  for (int i = 0; i < 9; ++i) Vi[i] = -0.5 * W[i] + c * W2[i];
  Vi[0] += 1.0; Vi[4] += 1.0; Vi[8] += 1.0;
  m3_vec(Vi, t, x);
}
/* Adj = [[R, t^R],[0, R]] (6x6 row-major) */
static inline void se3_adj(const double *T, double *A) {
  double R[9], t[3], th[9], tR[9];
  pose_R(T, R); pose_t(T, t); hat3(t, th); m3_mul(th, R, tR);
  memset(A, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    A[6 * i + j] = R[3 * i + j]; A[6 * i + 3 + j] = tR[3 * i + j]; A[6 * (i + 3) + 3 + j] = R[3 * i + j];
  }
}
/* d_lieBracketab_by_d_a(b) = -ad_b */
static inline void se3_dlie(const double *b, double *M) {
  double hu[9], hw[9];
  hat3(b, hu); hat3(b + 3, hw);
  memset(M, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    M[6 * i + j] = -hw[3 * i + j]; M[6 * i + 3 + j] = -hu[3 * i + j]; M[6 * (i + 3) + 3 + j] = -hw[3 * i + j];
  }
}
static inline void m6_mul(const double *A, const double *B, double *C) {
  double t[36];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) {
    double s = 0; for (int k = 0; k < 6; ++k) s += A[6 * i + k] * B[6 * k + j]; t[6 * i + j] = s;
  }
  memcpy(C, t, sizeof t);
}
/* closed-form 3x3 inverse (Eigen fixed-size inverse is cofactor based) */
static inline void m3_inv(const double *A, double *B) {
  double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  double id = 1.0 / det;
  double t[9];
  t[0] = c00 * id; t[1] = (A[2] * A[7] - A[1] * A[8]) * id; t[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  t[3] = c01 * id; t[4] = (A[0] * A[8] - A[2] * A[6]) * id; t[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  t[6] = c02 * id; t[7] = (A[1] * A[6] - A[0] * A[7]) * id; t[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  memcpy(B, t, sizeof t);
}
/* dense symmetric solve via LDL^T-free Gaussian elimination with partial pivoting (n<=6) */
static inline int solve_small(int n, const double *A, const double *b, double *x) {
  double M[6 * 7];
  for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) M[i * 7 + j] = A[i * n + j]; M[i * 7 + n] = b[i]; }
  for (int k = 0; k < n; ++k) {
    int p = k; double best = fabs(M[k * 7 + k]);
    for (int i = k + 1; i < n; ++i) if (fabs(M[i * 7 + k]) > best) { best = fabs(M[i * 7 + k]); p = i; }
    if (p != k) for (int j = 0; j <= n; ++j) { double t = M[k * 7 + j]; M[k * 7 + j] = M[p * 7 + j]; M[p * 7 + j] = t; }
    double piv = M[k * 7 + k];
    for (int i = k + 1; i < n; ++i) {
      double f = M[i * 7 + k] / piv;
      for (int j = k; j <= n; ++j) M[i * 7 + j] -= f * M[k * 7 + j];
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = M[i * 7 + n];
    for (int j = i + 1; j < n; ++j) s -= M[i * 7 + j] * x[j];
    x[i] = s / M[i * 7 + i];
  }
  return 0;
}
#endif
