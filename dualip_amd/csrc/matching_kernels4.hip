// matching_kernels4.hip -- the 256-wide fused pass (fused4_kernel.h), fp32, for handles without K-lanes-per-column slices: the benchmark's kernel.
#define DL_FUSED4_LANES 0
#define DL_FUSED4_F64 0
#include "fused4_kernel.h"
