// TEST INFRASTRUCTURE: stand-in for cv::gpu::GpuMat as the CUDA branch of the reference's dense tracker uses it (device images are host
// arrays here: the kernels run in the host emulation of ref_shim/svs_cuda_emul.h): create / setTo / size / step1 / data, plus a view constructor
#pragma once
#include <opencv2/core/core.hpp>
namespace cv {
namespace gpu {
class GpuMat {
 public:
  int rows, cols, type_;
  size_t step;      // bytes
  uint8_t *data;
  std::shared_ptr<std::vector<uint8_t> > own;
  GpuMat() : rows(0), cols(0), type_(CV_32F), step(0), data(0) {}
  GpuMat(int r, int c, int type, void *d, size_t s) : rows(r), cols(c), type_(type), step(s), data(static_cast<uint8_t *>(d)) {}
  void create(Size sz, int type) {
    rows = sz.height; cols = sz.width; type_ = type; step = (size_t)cols * Mat::elem(type);
    own.reset(new std::vector<uint8_t>((size_t)rows * step)); data = own->data();
  }
  void setTo(const Scalar &s) {      // float images only
    const int ch = (int)(Mat::elem(type_) / 4);
    for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) for (int k = 0; k < ch; ++k) reinterpret_cast<float *>(data + (size_t)r * step)[c * ch + k] = (float)s.val[k];
  }
  void upload(const Mat &m) {      // host -> "device": a deep copy with a tight pitch
    create(Size(m.cols, m.rows), m.type());
    for (int r = 0; r < rows; ++r) std::memcpy(data + (size_t)r * step, m.data + (size_t)r * m.step, step);
  }
  Size size() const { return Size(cols, rows); }
  size_t step1() const { return step / (type_ == CV_8U ? 1 : 4); }      // step in units of one channel's element
};
}  // namespace gpu
}  // namespace cv
