// TEST INFRASTRUCTURE: see opencv2/core/core.hpp in this directory
#pragma once
#include "core/core.hpp"
