// TEST INFRASTRUCTURE -- not part of the product.
// Stand-ins for the g2o base classes the reference's g2o_types/anchored_points.h derives from, reduced to the members its edge / vertex
// code touches (written from the public g2o interface): the estimate of a vertex; the vertex array, measurement, error vector, Jacobian
// slots and the parameter slot of an edge -- the edge pin calls computeError() / linearizeOplus() / oplusImpl() directly -- and a graph that only
// RECORDS what it is given (the SlamGraph marshalling pin).  No solver.
#pragma once
#include <Eigen/Core>
#include <iostream>
#include <map>
#include <vector>
namespace g2o {
// ---- a RECORDING graph: what SlamGraph::setupG2o / copyDataToG2o / optimize hand to g2o is kept as it arrives (vertices, edges, parameters, solver
// settings); SparseOptimizer::optimize(n) runs no solver but a hook of the wrapper (which dumps the record and may move the estimates) ----
struct Parameter {
  Parameter() : _id(-1) {}
  virtual ~Parameter() {}
  virtual bool read(std::istream &) { return false; }
  virtual bool write(std::ostream &) const { return false; }
  void setId(int i) { _id = i; }
  int id() const { return _id; }
  int _id;
};
struct RobustKernel { RobustKernel() : _delta(1.0) {} virtual ~RobustKernel() {} void setDelta(double d) { _delta = d; } double delta() const { return _delta; } double _delta; };
struct RobustKernelHuber : public RobustKernel {};
class HyperGraph {
 public:
  struct Vertex { Vertex() : _id(-1) {} virtual ~Vertex() {} int id() const { return _id; } void setId(int i) { _id = i; } int _id; };
  struct Edge {
    virtual ~Edge() {}
    std::vector<Vertex *> _vertices;
    std::vector<Vertex *> &vertices() { return _vertices; }
    const std::vector<Vertex *> &vertices() const { return _vertices; }
    void resize(size_t n) { _vertices.resize(n, 0); }
  };
  typedef std::map<int, Vertex *> VertexIDMap;
  virtual ~HyperGraph() {}
};
class OptimizableGraph : public HyperGraph {
 public:
  struct Vertex : public HyperGraph::Vertex {
    Vertex() : _fixed(false), _marginalized(false) {}
    virtual int dimension() const = 0;
    void setFixed(bool f) { _fixed = f; }
    bool fixed() const { return _fixed; }
    void setMarginalized(bool m) { _marginalized = m; }
    bool marginalized() const { return _marginalized; }
    bool _fixed, _marginalized;
  };
  struct Edge : public HyperGraph::Edge {
    Edge() : _robustKernel(0) {}
    void setRobustKernel(RobustKernel *k) { _robustKernel = k; }
    RobustKernel *robustKernel() const { return _robustKernel; }
    void resizeParameters(size_t n) { _parameterIds.resize(n, -1); }
    bool setParameterId(int arg, int id) { if (arg < 0 || (size_t)arg >= _parameterIds.size()) return false; _parameterIds[arg] = id; return true; }
    RobustKernel *_robustKernel;
    std::vector<int> _parameterIds;
  };
  bool addVertex(Vertex *v) { if (_vertices.count(v->id())) return false; _vertices[v->id()] = v; _vertexOrder.push_back(v); return true; }
  bool addEdge(Edge *e) { _edges.push_back(e); return true; }
  bool addParameter(Parameter *p) { if (_parameters.count(p->id())) return false; _parameters[p->id()] = p; return true; }
  const VertexIDMap &vertices() const { return _vertices; }
  VertexIDMap _vertices;
  std::vector<Vertex *> _vertexOrder;      // in the order of addVertex
  std::vector<Edge *> _edges;              // in the order of addEdge
  std::map<int, Parameter *> _parameters;
};
typedef OptimizableGraph::Vertex SvsVertexBase;
template <int D, typename T>
class BaseVertex : public OptimizableGraph::Vertex {
 public:
  static const int Dimension = D;
  virtual int dimension() const { return D; }
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
class SvsEdgeBase : public OptimizableGraph::Edge {
 public:
  SvsEdgeBase() : _pp(0) {}
  virtual ~SvsEdgeBase() {}
  Eigen::Matrix<double, D, D> &information() { return _information; }
  const E &measurement() const { return _measurement; }
  void setMeasurement(const E &m) { _measurement = m; }
  const Eigen::Matrix<double, D, D> &information() const { return _information; }
  template <class P> bool installParameter(P *&p, int) { _pp = reinterpret_cast<Parameter **>(&p); return true; }
  const Parameter *parameter(int) const { return *_pp; }
  virtual bool read(std::istream &) { return false; }
  virtual bool write(std::ostream &) const { return false; }
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
  BaseBinaryEdge() { this->_vertices.resize(2, 0); }      // (g2o sizes the vertex array of a fixed-arity edge itself)
  virtual void linearizeOplus() {}
  Eigen::Matrix<double, D, VI::Dimension> _jacobianOplusXi;
  Eigen::Matrix<double, D, VJ::Dimension> _jacobianOplusXj;
};
template <int D, typename E, typename VI>
class BaseUnaryEdge : public SvsEdgeBase<D, E> {
 public:
  BaseUnaryEdge() { this->_vertices.resize(1, 0); }
};
}  // namespace g2o
