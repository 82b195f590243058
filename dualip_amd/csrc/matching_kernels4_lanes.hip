// matching_kernels4_lanes.hip -- the 256-wide fused pass (fused4_kernel.h), fp32, with the loop over the slices of K = 2 .. 32 lanes per
// column (simplex columns of 25 .. 512 non-zeros, sell.h), in-place one-column slices and the dynamic deal inside a workgroup: launched for
// the handles that have such slices or many single-column tiles.
#define DL_FUSED4_LANES 1
#define DL_FUSED4_F64 0
#include "fused4_kernel.h"
