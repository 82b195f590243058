// simplex.h -- register-resident projection operators of the fused pass (shared with tools/test_wave.hip).
//
// Everything here is written branch-free on purpose: the fused kernel is bound by instruction issue (4 wavefronts per
// SIMD, measured), and every divergent `if` costs an s_and_saveexec / s_cbranch pair plus hazard nops.
#pragma once
#include "common.h"
#include "wave.h"

namespace dl {

constexpr int kProjLds = kProjLdsSlots;  // projection table slots in LDS; the last slot is the identity (columns in no entry)

// Kernel-side projection record.  Point-wise operators are all clamp(v, lo, hi) with infinite bounds where absent
// (box.py:15-16, cone.py:21-28); the simplex kinds additionally carry z and the feasibility threshold z + 1e-6.
template <class T>
struct alignas(16) ProjT {
    T lo, hi;  // clamp bounds (-inf / +inf when absent)
    T z;       // simplex radius
    T ztol;    // (T)(z + 1e-6): the reference's feasibility slack (simplex.py:155)
    int kind;
    int pad[3];
};

template <class T>
__device__ __forceinline__ ProjT<T> make_proj(int kind, double p0, double p1) {
    ProjT<T> p;
    p.kind = kind;
    p.lo = (T)(-INFINITY);
    p.hi = (T)INFINITY;
    p.z = (T)1;
    p.ztol = (T)1;
    p.pad[0] = p.pad[1] = p.pad[2] = 0;
    if (kind == DL_PROJ_BOX) {
        p.lo = (T)p0;
        p.hi = (T)p1;
    } else if (kind == DL_PROJ_CONE_LOWER) {
        p.lo = (T)p0;
    } else if (kind == DL_PROJ_CONE_UPPER) {
        p.hi = (T)p0;
    } else if (kind == DL_PROJ_SIMPLEX || kind == DL_PROJ_SIMPLEX_EQ) {
        p.z = (T)p0;
        p.ztol = (T)(p0 + 1e-6);
    }
    return p;
}

template <class T>
__device__ __forceinline__ T tmax(T a, T b) { return a > b ? a : b; }
template <class T>
__device__ __forceinline__ T tmin(T a, T b) { return a < b ? a : b; }

// clamp(v, lo, hi), lo <= hi (box.py:15-16, cone.py:21-28 with infinite bounds where absent): v_med3_f32 for float
__device__ __forceinline__ float clamp3(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
__device__ __forceinline__ double clamp3(double v, double lo, double hi) { return tmin(tmax(v, lo), hi); }

template <class T>
__device__ __forceinline__ T project_pointwise(T v, const ProjT<T>& p) { return clamp3(v, p.lo, p.hi); }

// theta = num / den.  double: IEEE division (parity mode).  float: reciprocal, multiply, one residual correction
// (<= 1 ulp from the correctly rounded quotient, a third of the instructions of the IEEE expansion).
__device__ __forceinline__ double div_exactish(double num, double den) { return num / den; }
__device__ __forceinline__ float div_exactish(float num, float den) {
    const float r = __builtin_amdgcn_rcpf(den);  // den is a small integer count: r is within 1 ulp of 1/den
    const float q = num * r;
    return fmaf(fmaf(-den, q, num), r, q);        // one residual correction
}

// ---- simplex_eq "padded block" compatibility (dl_matching_set_eq_padding) ----
// The reference projects a column inside a zero-padded [L x K] block, L = the longest column of the column's bucket
// (sparse_utils.py:185-209; buckets by nnz: (0,2], (2,4], (4,8], ... matching.py:87-114).  For simplex_eq the padding is
// visible exactly when the clamped column sums to less than z: the deficit is then spread over L entries instead of the
// column's own.  bucket(len) = bucketize(len, [0, 2, 4, ...]) = 1 for len <= 2, else ceil(log2(len)).
constexpr int kEqBuckets = 32;
__device__ __forceinline__ int eq_bucket(int len) { return len <= 2 ? 1 : 32 - __clz(len - 1); }

__device__ __forceinline__ bool is_simplex_kind(int k) { return k == DL_PROJ_SIMPLEX || k == DL_PROJ_SIMPLEX_EQ; }

// ---- per-lane constants of the segment machinery (computed once per kernel) ----
struct LaneConst {
    int lane;
    uint32_t le_lo, le_hi;  // bits 0..lane
    uint32_t gt_lo, gt_hi;  // bits lane+1..63
    int row_base;           // lane & ~15
};
__device__ __forceinline__ LaneConst make_lane_const(int lane) {
    LaneConst c;
    c.lane = lane;
    const uint64_t le = (2ull << lane) - 1ull;
    c.le_lo = (uint32_t)le;
    c.le_hi = (uint32_t)(le >> 32);
    c.gt_lo = ~c.le_lo;
    c.gt_hi = ~c.le_hi;
    c.row_base = lane & ~15;
    return c;
}

// (The segmented scans of the 64-wide tile -- make_seginfo_fast, simplex_batch -- went with that layout in round 5; the 256-wide tile's are in
//  simplex4.h, the column-per-lane slices' in sell.h.)

}  // namespace dl
