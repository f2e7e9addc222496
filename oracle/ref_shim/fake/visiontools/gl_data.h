// TEST INFRASTRUCTURE: stand-ins for VisionTools' GlPoint2f / 3f / 4f (draw lists of the front-end: float copies of Eigen vectors)
#pragma once
#include <Eigen/Core>
namespace VisionTools {
struct GlPoint2f { float x, y; GlPoint2f() : x(0), y(0) {} GlPoint2f(const Eigen::Vector2d &p) : x((float)p[0]), y((float)p[1]) {} GlPoint2f(float x_, float y_) : x(x_), y(y_) {} };
struct GlPoint3f { float x, y, z; GlPoint3f() : x(0), y(0), z(0) {} GlPoint3f(const Eigen::Vector3d &p) : x((float)p[0]), y((float)p[1]), z((float)p[2]) {} };
struct GlPoint4f { float x, y, z, w; GlPoint4f() : x(0), y(0), z(0), w(0) {} };
}
