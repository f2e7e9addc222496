// TEST INFRASTRUCTURE: recording stand-ins for g2o::SparseOptimizer, BlockSolver_6_3, LinearSolverCSparse and OptimizationAlgorithmLevenberg
// (see base_vertex.h): settings are kept, optimize(n) calls the wrapper's hook instead of a solver
#pragma once
#include "base_vertex.h"
namespace g2o {
template <typename M> struct LinearSolver { virtual ~LinearSolver() {} };
template <typename M> struct LinearSolverCSparse : public LinearSolver<M> {};
struct Solver { virtual ~Solver() {} };
template <int P, int L> struct BlockSolverTraits { typedef Eigen::Matrix<double, P, P> PoseMatrixType; static const int PoseDim = P, LandmarkDim = L; };
template <typename Traits> struct BlockSolver : public Solver {
  typedef typename Traits::PoseMatrixType PoseMatrixType;
  typedef LinearSolver<PoseMatrixType> LinearSolverType;
  explicit BlockSolver(LinearSolverType *ls) : linear_solver(ls) {}
  LinearSolverType *linear_solver;
  static const int PoseDim = Traits::PoseDim, LandmarkDim = Traits::LandmarkDim;
};
typedef BlockSolver<BlockSolverTraits<6, 3> > BlockSolver_6_3;
struct OptimizationAlgorithm { virtual ~OptimizationAlgorithm() {} };
struct OptimizationAlgorithmLevenberg : public OptimizationAlgorithm {
  explicit OptimizationAlgorithmLevenberg(Solver *s) : solver(s), max_trials_after_failure(10), user_lambda_init(0) {}      // g2o's defaults
  void setMaxTrialsAfterFailure(int n) { max_trials_after_failure = n; }
  void setUserLambdaInit(double l) { user_lambda_init = l; }
  Solver *solver;
  int max_trials_after_failure;
  double user_lambda_init;
};
class SparseOptimizer;
extern void (*svs_shim_g2o_optimize_hook)(SparseOptimizer *, int);
class SparseOptimizer : public OptimizableGraph {
 public:
  SparseOptimizer() : verbose(false), algorithm(0), initialized(0), iterations(-1) {}
  void setVerbose(bool v) { verbose = v; }
  void setAlgorithm(OptimizationAlgorithm *a) { algorithm = a; }
  OptimizationAlgorithm *solver() { return algorithm; }
  bool initializeOptimization() { ++initialized; return true; }
  int optimize(int n) { iterations = n; if (svs_shim_g2o_optimize_hook) svs_shim_g2o_optimize_hook(this, n); return n; }
  bool verbose;
  OptimizationAlgorithm *algorithm;
  int initialized, iterations;
};
}  // namespace g2o
