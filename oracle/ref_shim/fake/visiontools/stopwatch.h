// TEST INFRASTRUCTURE: stand-in for VisionTools::StopWatch (pose_optimizer.h includes the header; SlamGraph::optimize times the solver with it)
#pragma once
namespace VisionTools {
class StopWatch {
 public:
  void start() {}
  void stop() {}
  double get_stopped_time() const { return 0.0; }
};
}
