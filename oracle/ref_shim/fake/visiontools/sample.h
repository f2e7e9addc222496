// TEST INFRASTRUCTURE: VisionTools::Sample::uniform(a, b) -- an integer drawn uniformly from [a, b] -- as QuadTree's EquiIter names it (quadtree.h:183,267:
// the order in which StereoFrontend::addMorePointsToOtherFrame visits the corners of a new keyframe).  VisionTools' generator is third party and absent; the
// stand-in is a 32-bit LCG (Numerical Recipes constants) whose state a wrapper may reset (svs_shim_sample_state), so that two builds of the same reference code
// draw the same numbers as long as they take the same path.
#pragma once
namespace VisionTools {
inline unsigned &svs_shim_sample_state() { static unsigned s = 2463534242u; return s; }
struct Sample {
  static int uniform(int a, int b) {
    unsigned &s = svs_shim_sample_state();
    s = s * 1664525u + 1013904223u;
    return b <= a ? a : a + (int)((s >> 8) % (unsigned)(b - a + 1));
  }
};
}
