// simplex.h -- register-resident projection operators of the fused pass (shared with tools/test_wave.hip).
//
// Everything here is written branch-free on purpose: the fused kernel is bound by instruction issue (4 wavefronts per
// SIMD, measured), and every divergent `if` costs an s_and_saveexec / s_cbranch pair plus hazard nops.
#pragma once
#include "common.h"
#include "wave.h"

namespace dl {

constexpr int kBatch = 2;      // tiles a wavefront advances in lock-step (independent dependency chains)
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

// Column segments of a short tile from its head mask (bit k <=> lane k starts a column; bit 0 must be set).
__device__ __forceinline__ SegInfo make_seginfo_fast(uint64_t head, const LaneConst& c) {
    SegInfo s;
    const uint32_t hlo = (uint32_t)head, hhi = (uint32_t)(head >> 32);
    // start: highest head bit at or below this lane
    const uint32_t blo = hlo & c.le_lo, bhi = hhi & c.le_hi;
    const int st_lo = 31 - __clz((int)blo);  // blo != 0 whenever bhi == 0 (bit 0 is a head)
    const int st_hi = 63 - __clz((int)bhi);
    s.start = bhi ? st_hi : st_lo;
    // tail: lane before the next head strictly above this lane (63 if none)
    const uint32_t alo = hlo & c.gt_lo, ahi = hhi & c.gt_hi;
    const int nx_lo = __ffs((int)alo) - 1;  // -1 when alo == 0
    const int nx_hi = ahi ? 32 + __ffs((int)ahi) - 1 : 64;
    const int next = alo ? nx_lo : nx_hi;
    s.tail = next - 1;
    s.lane = c.lane;
    s.d = c.lane - s.start;
    // bits start..tail
    const uint64_t upto_tail = (2ull << s.tail) - 1ull;
    s.segmask = upto_tail & (~0ull << s.start);
    // scan-step predicates: source lane (lane - o) must be inside the segment and inside this 16-lane DPP row
    const int lim = s.start > c.row_base ? s.start : c.row_base;
    s.p1 = c.lane - 1 >= lim;
    s.p2 = c.lane - 2 >= lim;
    s.p4 = c.lane - 4 >= lim;
    s.p8 = c.lane - 8 >= lim;
    s.pA = (c.lane & 16) && s.start < c.row_base;  // rows 1,3: segment continues from the previous row
    s.pB = (c.lane & 32) && s.start < 32;          // rows 2,3: segment reaches back past lane 32
    return s;
}

// Simplex projection of every column segment of kBatch short tiles in lock-step, one value per lane and tile.
// Equals _duchi_proj (simplex.py:126-236) column by column: clamp at 0; (inequality) keep if sum <= z + 1e-6;
// vertex z*e_argmax when only the maximum exceeds max - z (the reference's top-2 shortcut); else
// x = max(u - theta, 0) with theta = (sum of the support - z) / |support|, found by the monotone Newton (Michelot)
// iteration started from the lower bound theta_0 = max - z.
// One segmented MAX scan serves every column; SUM scans run only while some column is neither a vertex nor done.
// (The feasibility test uses the sum over {u > max - z}, which equals the full sum whenever max < z; for max >= z
//  the two can differ by at most len * 1e-6 right at the decision boundary -- documented in DESIGN.md.)
// smp[q] == false (tile absent or not a simplex tile): x[q] is left untouched.
template <bool USE_DPP, class T>
__device__ __forceinline__ void simplex_batch(const T (&v)[kBatch], const bool (&valid)[kBatch], const uint64_t (&head)[kBatch],
                                              const ProjT<T> (&pj)[kBatch], const bool (&smp)[kBatch], const LaneConst& lc, T (&x)[kBatch],
                                              const int32_t* const (&eq_row)[kBatch]) {
    SegInfo sg[kBatch];
    T u[kBatch], th[kBatch], v1[kBatch];
    bool act[kBatch], proj[kBatch], onehot[kBatch], live[kBatch];
    int cnt_prev[kBatch];
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
        sg[q] = make_seginfo_fast(head[q] | 1ull, lc);
        live[q] = valid[q] && smp[q];
        u[q] = live[q] ? relu(v[q]) : (T)0;
    }
#pragma unroll
    for (int q = 0; q < kBatch; ++q) v1[q] = seg_allreduce<USE_DPP>(u[q], sg[q], (T)(-INFINITY), OpMax());
    bool any_act = false;
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
        th[q] = (T)(v1[q] - pj[q].z);
        const bool in = u[q] > th[q];
        const int cnt = __popcll(__ballot(in && live[q]) & sg[q].segmask);
        onehot[q] = live[q] && cnt == 1 && sg[q].tail > sg[q].start;  // only the maximum exceeds max - z: vertex (simplex.py:177-193)
        proj[q] = false;
        act[q] = live[q] && !onehot[q];
        cnt_prev[q] = 0;
        any_act = any_act || act[q];
    }
    // Newton (Michelot) passes.  Each pass costs one SUM scan per tile that still has an undecided column; whether a
    // column is done is decided from the ballot alone (the support only shrinks, so an unchanged size means an
    // unchanged set and th already is the fixed point) -- no extra scan to confirm convergence.
    if (__any(any_act)) {
        for (int it = 0; it < 2 * kTileLanes + 2; ++it) {
            any_act = false;
#pragma unroll
            for (int q = 0; q < kBatch; ++q) {
                if (!__any(act[q])) continue;  // wave-uniform: this tile has nothing left to do
                const bool in = u[q] > th[q];
                const int cnt = __popcll(__ballot(in && live[q]) & sg[q].segmask);
                const bool conv = it > 0 && (cnt == cnt_prev[q] || cnt == 0);
                act[q] = act[q] && !conv;
                if (__any(act[q])) {
                    const T sumA = seg_allreduce<USE_DPP>(in ? u[q] : (T)0, sg[q], (T)0, OpAdd());
                    T den = (T)cnt;
                    if (eq_row[q]) {  // simplex_eq compatibility mode (fused_common.h: eq_bucket): sum < z on the first pass means
                                      // theta < 0, the support is the whole column (cnt = its length) plus the padding zeros
                        const int b = cnt <= 2 ? 1 : 32 - __clz(cnt - 1);
                        const T L = (T)eq_row[q][b > 0 ? b : 1];
                        den = (it == 0 && sumA < pj[q].z) ? L : den;
                    }
                    const T th_new = div_exactish((T)(sumA - pj[q].z), den);
                    // feasible after the clamp (simplex.py:153-158): only decided on the first pass
                    const bool feas = it == 0 && pj[q].kind == DL_PROJ_SIMPLEX && !(sumA > pj[q].ztol);
                    const bool upd = act[q] && !feas && cnt != 0;
                    proj[q] = proj[q] || upd;
                    th[q] = upd ? tmax(th_new, th[q]) : th[q];  // thresholds never decrease: nested supports, guaranteed termination
                    cnt_prev[q] = upd ? cnt : cnt_prev[q];
                    act[q] = upd;
                }
                any_act = any_act || act[q];
            }
            if (!__any(any_act)) break;
        }
    }
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
        const T xg = relu((T)(u[q] - th[q]));             // general: threshold
        const T xv = (u[q] > th[q]) ? pj[q].z : (T)0;    // vertex
        T r = proj[q] ? xg : u[q];
        r = onehot[q] ? xv : r;
        x[q] = live[q] ? r : x[q];
    }
}

}  // namespace dl
