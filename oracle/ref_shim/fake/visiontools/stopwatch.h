// TEST INFRASTRUCTURE: empty stand-in (pose_optimizer.h includes it and uses nothing of it)
#pragma once
