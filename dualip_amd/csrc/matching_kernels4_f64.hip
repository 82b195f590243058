// matching_kernels4_f64.hip -- the 256-wide fused pass (fused4_kernel.h), fp64 (parity runs), first binary.
#define DL_FUSED4_LANES 0
#define DL_FUSED4_F64 1
#include "fused4_kernel.h"
