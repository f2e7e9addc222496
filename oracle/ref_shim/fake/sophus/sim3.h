// TEST INFRASTRUCTURE: empty stand-in (maths_utils.cpp includes it, the compiled functions use nothing of it)
#pragma once
