// TEST INFRASTRUCTURE: see core/sparse_optimizer.h
#pragma once
#include "../../core/sparse_optimizer.h"
