// TEST INFRASTRUCTURE: empty stand-in (StereoFrontend keeps a pointer to one)
#pragma once
namespace VisionTools { class PerformanceMonitor {}; }
