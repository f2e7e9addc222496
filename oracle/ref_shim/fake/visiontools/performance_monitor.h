// TEST INFRASTRUCTURE: stand-in for VisionTools::PerformanceMonitor (StereoFrontend brackets its stages with start / stop)
#pragma once
namespace VisionTools { class PerformanceMonitor { public: void start(const char *) {} void stop(const char *) {} }; }
