// TEST INFRASTRUCTURE -- not part of the product.
// The OpenCV 2.4 names the reference's quadtree.h / global.h / maths_utils.h / keyframes.h / fast_grid.{h,cpp} use, written from the public
// OpenCV 2.4 interface so that those reference files compile on the host from where they lie (oracle/Makefile):
//   Point_<T>, Rect_<T> (half-open contains(): x <= pt.x < x + width, y <= pt.y < y + height), Range, Size, KeyPoint;
//   Mat: a 2-D 8-bit view (data, step, rows, cols) with the ROI operator (rowRange, colRange), clone() and copyTo();
//   FastFeatureDetector(threshold, nonmaxSuppression).detect(image, keypoints): forwards to a hook.  The pin binds the hook to the
//   oracle's restatement of OpenCV's FAST-9/16 (svs_ref_fast9_16): what is compiled from the reference and checked is everything AROUND
//   the detector -- cell layout, the adaptive threshold state machine, corner offsets and the quadtree insertion (fast_grid.cpp:23-152).
#pragma once
#include <cstring>
#include <memory>
#include <stdint.h>
#include <vector>
namespace cv {
template <typename T> struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<double> Point2d;
typedef Point_<float> Point2f;
typedef Point_<int> Point;
template <typename T> struct Rect_ {
  T x, y, width, height;
  Rect_() : x(0), y(0), width(0), height(0) {}
  Rect_(T x_, T y_, T w_, T h_) : x(x_), y(y_), width(w_), height(h_) {}
  bool contains(const Point_<T> &pt) const { return x <= pt.x && pt.x < x + width && y <= pt.y && pt.y < y + height; }
};
struct Size;
struct Range {
  int start, end;
  Range() : start(0), end(0) {}
  Range(int s, int e) : start(s), end(e) {}
};
struct Size {
  int width, height;
  Size() : width(0), height(0) {}
  Size(int w, int h) : width(w), height(h) {}
};
struct Rect : public Rect_<int> {
  Rect() {}
  Rect(int x_, int y_, int w_, int h_) : Rect_<int>(x_, y_, w_, h_) {}
  Rect(const Point &p, const Size &s) : Rect_<int>(p.x, p.y, s.width, s.height) {}
};
struct KeyPoint {
  Point2f pt;
  KeyPoint() {}
  KeyPoint(float x, float y) : pt(x, y) {}
};
}  // namespace cv
#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_HSV2BGR 54
#define CV_32F 5
#define CV_32FC4 29
namespace cv {
struct Scalar {
  double val[4];
  Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
};
struct Vec4f {
  float v[4];
  Vec4f() { v[0] = v[1] = v[2] = v[3] = 0.f; }
  template <typename A, typename B, typename C, typename D> Vec4f(A a, B b, C c, D d) { v[0] = (float)a; v[1] = (float)b; v[2] = (float)c; v[3] = (float)d; }
  float &operator[](int i) { return v[i]; }
  const float &operator[](int i) const { return v[i]; }
};
class Mat {
 public:
  int rows, cols, type_;
  size_t step;      // bytes
  uint8_t *data;
  std::shared_ptr<std::vector<uint8_t> > own;      // storage of clones / allocations (views leave it empty or share it)
  Mat() : rows(0), cols(0), type_(CV_8U), step(0), data(0) {}
  static size_t elem(int type) { return type == CV_8U ? 1 : type == CV_8UC3 ? 3 : type == CV_32F ? 4 : 16; }
  Mat(int r, int c, int type, void *d, size_t s = 0) : rows(r), cols(c), type_(type), step(s ? s : (size_t)c * elem(type)), data(static_cast<uint8_t *>(d)) {}
  Mat(int r, int c, int type) : rows(r), cols(c), type_(type), step((size_t)c * elem(type)), data(0) { own.reset(new std::vector<uint8_t>((size_t)r * step)); data = own->data(); }
  Mat(Size sz, int type) : rows(sz.height), cols(sz.width), type_(type), step((size_t)sz.width * elem(type)), data(0) { own.reset(new std::vector<uint8_t>((size_t)rows * step)); data = own->data(); }
  Mat(Size sz, int type, int) : rows(sz.height), cols(sz.width), type_(type), step((size_t)sz.width * elem(type)), data(0) { own.reset(new std::vector<uint8_t>((size_t)rows * step)); data = own->data(); }      // (fill value: display images only)
  void create(int r, int c, int type) { rows = r; cols = c; type_ = type; step = (size_t)c * elem(type); own.reset(new std::vector<uint8_t>((size_t)r * step)); data = own->data(); }
  void create(Size sz, int type) { create(sz.height, sz.width, type); }
  void setTo(const Scalar &s) {      // float images (1 or 4 channels)
    const int ch = (int)(elem(type_) / 4);
    for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) for (int k = 0; k < ch; ++k) reinterpret_cast<float *>(data + (size_t)r * step)[c * ch + k] = (float)s.val[k];
  }
  void convertTo(Mat &, int, double = 1, double = 0) const {}      // (only the colour-coded disparity DISPLAY image of processFrame goes through it)
  template <typename T> T *ptr(int y, int x) { return &at<T>(y, x); }
  template <typename T> const T *ptr(int y, int x) const { return &at<T>(y, x); }
  Mat operator()(const Rect &r) const { return (*this)(Range(r.y, r.y + r.height), Range(r.x, r.x + r.width)); }
  int type() const { return type_; }
  Size size() const { return Size(cols, rows); }
  template <typename T> T &at(int y, int x) { return *reinterpret_cast<T *>(data + (size_t)y * step + (size_t)x * sizeof(T)); }
  template <typename T> const T &at(int y, int x) const { return *reinterpret_cast<const T *>(data + (size_t)y * step + (size_t)x * sizeof(T)); }
  Mat operator()(const Range &rr, const Range &cr) const {
    Mat m(*this);
    m.rows = rr.end - rr.start; m.cols = cr.end - cr.start;
    m.data = data + (size_t)rr.start * step + (size_t)cr.start * elem(type_);
    return m;
  }
  Mat clone() const {
    Mat m;
    m.rows = rows; m.cols = cols; m.type_ = type_; m.step = (size_t)cols * elem(type_);
    m.own.reset(new std::vector<uint8_t>((size_t)rows * m.step));
    m.data = m.own->data();
    for (int r = 0; r < rows; ++r) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, m.step);
    return m;
  }
  void copyTo(Mat &dst) const {
    if (dst.rows != rows || dst.cols != cols || !dst.data) dst = clone();
    else for (int r = 0; r < rows; ++r) std::memcpy(dst.data + (size_t)r * dst.step, data + (size_t)r * step, (size_t)cols * elem(type_));
  }
};
inline void merge(const std::vector<Mat> &, Mat &) {}      // display only
inline void cvtColor(const Mat &, Mat &, int) {}            // display only
// hook: n = fn(image, w, h, stride, threshold, xy (x, y pairs, detection order), cap)
typedef int (*svs_shim_fast_fn)(const uint8_t *, int, int, int, int, int16_t *, int);
extern svs_shim_fast_fn svs_shim_fast_hook;
class FastFeatureDetector {
 public:
  FastFeatureDetector(int threshold = 10, bool nonmax = true) : thr_(threshold), nonmax_(nonmax) {}
  void detect(const Mat &img, std::vector<KeyPoint> &kp) const {
    kp.clear();
    if (nonmax_ || !svs_shim_fast_hook || img.rows <= 0 || img.cols <= 0) return;      // the reference only ever asks for (thr, false)
    std::vector<int16_t> xy((size_t)2 * img.rows * img.cols + 2);
    const int n = svs_shim_fast_hook(img.data, img.cols, img.rows, (int)img.step, thr_, xy.data(), img.rows * img.cols);
    for (int i = 0; i < n; ++i) kp.push_back(KeyPoint((float)xy[2 * i], (float)xy[2 * i + 1]));
  }
 private:
  int thr_;
  bool nonmax_;
};
// StereoBM (OpenCV 2.4.2 calib3d: external to the reference): state as calcDisparityCpu sets it (stereo_frontend.cpp:620-641), operator() forwards to a hook.  The
// sequence pin binds the hook to the oracle's restatement of the block matcher (oracle/stereo.c: svs_ref_stereo_bm), so that what is compiled from the reference is the
// call site -- which images go in, which parameters, where the disparity lands (CV_32F, pixels) -- and the HIP branch put at its head is compared with exactly that.
struct svs_shim_stereo_state { int preFilterCap, SADWindowSize, minDisparity, numberOfDisparities, textureThreshold, uniquenessRatio, speckleWindowSize, speckleRange, disp12MaxDiff; };
typedef void (*svs_shim_stereobm_fn)(const uint8_t *left, const uint8_t *right, int w, int h, int stride, const svs_shim_stereo_state *p, float *disp, int dstride);
extern svs_shim_stereobm_fn svs_shim_stereobm_hook;
class StereoBM {
 public:
  svs_shim_stereo_state st_, *state;
  StereoBM() : state(&st_) { std::memset(&st_, 0, sizeof st_); }
  void operator()(const Mat &left, const Mat &right, Mat &disp, int type) {
    std::vector<float> buf((size_t)left.rows * left.cols, -1.f);
    if (svs_shim_stereobm_hook && type == CV_32F && left.step == right.step)
      svs_shim_stereobm_hook(left.data, right.data, left.cols, left.rows, (int)left.step, state, buf.data(), left.cols);
    disp = Mat(left.rows, left.cols, CV_32F, buf.data(), (size_t)left.cols * 4).clone();
  }
};
}  // namespace cv
