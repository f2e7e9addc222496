// TEST INFRASTRUCTURE: empty stand-in (global.h includes it, quadtree.h uses nothing of it)
#pragma once
