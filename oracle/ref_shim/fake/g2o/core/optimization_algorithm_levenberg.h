// TEST INFRASTRUCTURE: see core/sparse_optimizer.h
#pragma once
#include "sparse_optimizer.h"
