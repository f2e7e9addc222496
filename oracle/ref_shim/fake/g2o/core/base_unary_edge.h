// TEST INFRASTRUCTURE: see base_vertex.h in this directory
#pragma once
#include "base_vertex.h"
