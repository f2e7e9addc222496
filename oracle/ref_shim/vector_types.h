/* stands in for CUDA's <vector_types.h> (included by the reference's gpu/dense_tracking.cuh:21) when the two reference
   files are compiled on the host for oracle/_ref -- see svs_cuda_emul.h.  Test infrastructure only. */
#include "svs_cuda_emul.h"
