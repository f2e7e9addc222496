// TEST INFRASTRUCTURE: VisionTools::pyrFromZero_d as maths_utils.cpp's interpolateDisparity uses it (x at pyramid level 0 -> level l:
// x / 2^l; only called with x = 1, where every formulation is exact), and the empty base class of the camera stand-in
#pragma once
namespace VisionTools {
class AbstractCamera {};
inline double pyrFromZero_d(double x, int level) { return x / (double)(1 << level); }
}
