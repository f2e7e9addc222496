// TEST INFRASTRUCTURE: VisionTools::Sample::uniform is named by QuadTree's random iterators (never instantiated by the pin)
#pragma once
namespace VisionTools {
struct Sample { static int uniform(int a, int b) { return a + (b - a) / 2; } };
}
