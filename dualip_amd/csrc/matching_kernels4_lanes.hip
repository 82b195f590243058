// matching_kernels4_lanes.hip -- the 256-wide fused pass (fused4_kernel.h) with the loop over the slices of K = 2 .. 16 lanes per column
// (simplex columns of 25 .. 255 non-zeros, sell.h): launched for the handles that have such slices.
#define DL_FUSED4_LANES 1
#include "fused4_kernel.h"
