// matching_kernels4_lanes_f64.hip -- the second binary of the 256-wide fused pass (fused4_kernel.h), fp64 (parity runs).
#define DL_FUSED4_LANES 1
#define DL_FUSED4_F64 1
#include "fused4_kernel.h"
