// TEST INFRASTRUCTURE -- not part of the product.
// Stand-ins for the g2o base classes the reference's g2o_types/anchored_points.h derives from, reduced to the members its edge / vertex
// code touches (written from the public g2o interface): the estimate of a vertex; the vertex array, measurement, error vector, Jacobian
// slots and the parameter slot of an edge.  No graph, no solver: the pin calls computeError() / linearizeOplus() / oplusImpl() directly.
#pragma once
#include <Eigen/Core>
#include <iostream>
#include <vector>
namespace g2o {
struct Parameter { virtual ~Parameter() {} virtual bool read(std::istream &) { return false; } virtual bool write(std::ostream &) const { return false; } };
struct SvsVertexBase { virtual ~SvsVertexBase() {} };
template <int D, typename T>
class BaseVertex : public SvsVertexBase {
 public:
  static const int Dimension = D;
  const T &estimate() const { return _estimate; }
  T &estimate() { return _estimate; }
  void setEstimate(const T &e) { _estimate = e; }
  virtual bool read(std::istream &) { return false; }
  virtual bool write(std::ostream &) const { return false; }
  virtual void oplusImpl(const double *) {}
  virtual void setToOriginImpl() {}
  T _estimate;
};
// a Jacobian slot takes whatever fixed-size block the edge assigns to it
struct SvsJacobian {
  int rows, cols;
  double v[36];
  SvsJacobian() : rows(0), cols(0) {}
  template <int R, int C> SvsJacobian &operator=(const Eigen::Matrix<double, R, C> &m) {
    rows = R; cols = C;
    for (int i = 0; i < R * C; ++i) v[i] = m.v[i];
    return *this;
  }
};
template <int D, typename E>
class SvsEdgeBase {
 public:
  SvsEdgeBase() : _pp(0) {}
  virtual ~SvsEdgeBase() {}
  const E &measurement() const { return _measurement; }
  void setMeasurement(const E &m) { _measurement = m; }
  const Eigen::Matrix<double, D, D> &information() const { return _information; }
  void resizeParameters(int) {}
  template <class P> bool installParameter(P *&p, int) { _pp = reinterpret_cast<Parameter **>(&p); return true; }
  const Parameter *parameter(int) const { return *_pp; }
  virtual bool read(std::istream &) { return false; }
  virtual bool write(std::ostream &) const { return false; }
  std::vector<SvsVertexBase *> _vertices;
  E _measurement;
  Eigen::Matrix<double, D, 1> _error;
  Eigen::Matrix<double, D, D> _information;
 private:
  Parameter **_pp;
};
template <int D, typename E>
class BaseMultiEdge : public SvsEdgeBase<D, E> {
 public:
  BaseMultiEdge() : _jacobianOplus(3) {}
  virtual void linearizeOplus() {}
  std::vector<SvsJacobian> _jacobianOplus;
};
template <int D, typename E, typename VI, typename VJ>
class BaseBinaryEdge : public SvsEdgeBase<D, E> {
 public:
  virtual void linearizeOplus() {}
  Eigen::Matrix<double, D, VI::Dimension> _jacobianOplusXi;
  Eigen::Matrix<double, D, VJ::Dimension> _jacobianOplusXj;
};
template <int D, typename E, typename VI>
class BaseUnaryEdge : public SvsEdgeBase<D, E> {};
}  // namespace g2o
