// wave.h -- 64-lane wavefront primitives for gfx950: DPP lane shifts, segmented scans over the column
// segments of a wave tile, wave-wide reductions.  Wave width is hard-coded to 64 (CDNA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dl {

// Maxima without the NaN-canonicalisation pre-ops the compiler adds to fmaxf() under the IEEE mode bit (three
// instructions instead of one), and without inline asm (which it refuses to speculate, turning selects into branches):
// non-negative floats order like their bit patterns, and max(v, 0) is a signed-integer max against 0.
__device__ __forceinline__ float relu(float v) { return __int_as_float(max(__float_as_int(v), 0)); }
__device__ __forceinline__ double relu(double v) { return v > 0.0 ? v : 0.0; }
// clamp to [0, largest finite]: one v_med3 (a NaN input gives 0 or the upper bound, never a NaN)
__device__ __forceinline__ float relu_finite(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 3.402823466e+38f); }
__device__ __forceinline__ double relu_finite(double v) { return v > 0.0 ? (v < 1.7976931348623157e+308 ? v : 1.7976931348623157e+308) : 0.0; }
__device__ __forceinline__ float max_nonneg(float a, float b) { return __uint_as_float(max(__float_as_uint(a), __float_as_uint(b))); }
__device__ __forceinline__ double max_nonneg(double a, double b) { return a > b ? a : b; }

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// ---- raw DPP moves: lanes whose source is outside the DPP pattern (or masked by ROWMASK) receive `ident` ----
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_mov(float ident, float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(x), CTRL, ROWMASK, 0xf, false));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov(double ident, double x) {
    int ilo = __double2loint(ident), ihi = __double2hiint(ident);
    int xlo = __double2loint(x), xhi = __double2hiint(x);
    int rlo = __builtin_amdgcn_update_dpp(ilo, xlo, CTRL, ROWMASK, 0xf, false);
    int rhi = __builtin_amdgcn_update_dpp(ihi, xhi, CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(rhi, rlo);
}

// zero-fill form (bound_ctrl:1, old = 0): no identity pre-load, fuses into v_add_f32_dpp / v_max_f32_dpp.
// Valid whenever 0 is the identity of the reduction -- sums, and maxima of non-negative values.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_mov0(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROWMASK, 0xf, true));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov0(double x) {
    const int rlo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWMASK, 0xf, true);
    const int rhi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWMASK, 0xf, true);
    return __hiloint2double(rhi, rlo);
}

__device__ __forceinline__ float bperm(int src_lane, float x) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(x)));
}
__device__ __forceinline__ double bperm(int src_lane, double x) {
    int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(x));
    int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(x));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ long long bperm(int src_lane, long long x) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned int)(unsigned long long)x);
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned int)((unsigned long long)x >> 32));
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo);
}

constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;  // butterflies inside a row of 16
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

// ---- column segments of a short tile ----
// head mask bit k <=> lane k is the first non-zero of a column.  Segments are contiguous lane ranges.
struct SegInfo {
    int lane;
    int start;         // first lane of this lane's segment
    int tail;          // last lane of this lane's segment
    int d;             // lane - start
    uint64_t segmask;  // bits [start, tail]
    // predicates of the segmented scan steps (constant per tile)
    bool p1, p2, p4, p8, pA, pB;
};

__device__ __forceinline__ SegInfo make_seginfo(uint64_t head_mask, int lane) {
    SegInfo s;
    s.lane = lane;
    const uint64_t upto = (2ull << lane) - 1ull;  // bits 0..lane (lane 63: wraps to all ones)
    s.start = 63 - __clzll((long long)(head_mask & upto));
    const uint64_t rest = lane == 63 ? 0ull : (head_mask >> (lane + 1));
    const int end = rest ? lane + 1 + (__ffsll((long long)rest) - 1) : 64;  // exclusive
    s.tail = end - 1;
    s.d = lane - s.start;
    const uint64_t hi = end == 64 ? ~0ull : ((1ull << end) - 1ull);
    s.segmask = hi & (~0ull << s.start);
    const int r = lane & 15;
    const int dr = s.d < r ? s.d : r;  // reach inside this 16-lane DPP row
    s.p1 = dr >= 1;
    s.p2 = dr >= 2;
    s.p4 = dr >= 4;
    s.p8 = dr >= 8;
    s.pA = (lane & 16) && s.d > r;            // rows 1,3: segment continues from the previous row
    s.pB = (lane & 32) && s.d > (lane & 31);  // rows 2,3: segment reaches back past lane 32
    return s;
}

struct OpAdd {
    template <class T>
    __device__ __forceinline__ T operator()(T a, T b) const { return a + b; }
};
struct OpMax {  // any sign (compare + select)
    template <class T>
    __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; }
};
struct OpMaxNonNeg {  // operands >= 0: one integer max on the bit patterns (float)
    template <class T>
    __device__ __forceinline__ T operator()(T a, T b) const { return max_nonneg(a, b); }
};

// Segmented inclusive scan.  DPP form: 4 in-row shifts + 2 row broadcasts (no LDS traffic);
// bpermute form: Hillis-Steele over ds_bpermute (kept as the independently-simple cross-check).
template <bool USE_DPP, class T, class Op>
__device__ __forceinline__ T seg_scan(T x, const SegInfo& s, T ident, Op op) {
    if constexpr (USE_DPP) {
        T t;
        t = dpp_mov<DPP_ROW_SHR1, 0xf>(ident, x);
        x = s.p1 ? op(x, t) : x;
        t = dpp_mov<DPP_ROW_SHR2, 0xf>(ident, x);
        x = s.p2 ? op(x, t) : x;
        t = dpp_mov<DPP_ROW_SHR4, 0xf>(ident, x);
        x = s.p4 ? op(x, t) : x;
        t = dpp_mov<DPP_ROW_SHR8, 0xf>(ident, x);
        x = s.p8 ? op(x, t) : x;
        t = dpp_mov<DPP_ROW_BCAST15, 0xa>(ident, x);
        x = s.pA ? op(x, t) : x;
        t = dpp_mov<DPP_ROW_BCAST31, 0xc>(ident, x);
        x = s.pB ? op(x, t) : x;
        return x;
    } else {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int src = s.lane - o;
            T t = bperm(src < 0 ? 0 : src, x);
            x = (s.d >= o) ? op(x, t) : x;
        }
        return x;
    }
}

// Every lane receives the reduction over its own segment.
template <bool USE_DPP, class T, class Op>
__device__ __forceinline__ T seg_allreduce(T x, const SegInfo& s, T ident, Op op) {
    x = seg_scan<USE_DPP>(x, s, ident, op);
    return bperm(s.tail, x);
}

// ---- wave-wide (unsegmented) reductions: result valid in every lane ----
template <class T, class Op>
__device__ __forceinline__ T wave_allreduce(T x, Op op) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        T t = bperm((lane_id() ^ o), x);
        x = op(x, t);
    }
    return x;
}

// The same on the DPP unit: a butterfly inside every row of 16 lanes (quad_perm xor 1, xor 2, half-row mirror, row mirror -- every
// lane has a valid source, commutative steps, so all 16 end with identical bits), then the four row results are read into scalar
// registers and combined in one fixed order.  ~12 vector instructions against six ds_bpermute round trips: the single-column
// walker's Newton passes are two such reductions each and nothing else of substance (round 3).
__device__ __forceinline__ float readlane_t(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ double readlane_t(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ uint32_t readlane_t(uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); }
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov0_u(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_any(float x) { return dpp_mov0<CTRL, 0xf>(x); }
template <int CTRL>
__device__ __forceinline__ double dpp_any(double x) { return dpp_mov0<CTRL, 0xf>(x); }
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_any(uint32_t x) { return dpp_mov0_u<CTRL>(x); }
template <class T, class Op>
__device__ __forceinline__ T wave_allreduce_dpp(T x, Op op) {
    x = op(x, dpp_any<DPP_QUAD_XOR1>(x));
    x = op(x, dpp_any<DPP_QUAD_XOR2>(x));
    x = op(x, dpp_any<DPP_ROW_HALF_MIRROR>(x));
    x = op(x, dpp_any<DPP_ROW_MIRROR>(x));
    const T r0 = readlane_t(x, 0), r1 = readlane_t(x, 16), r2 = readlane_t(x, 32), r3 = readlane_t(x, 48);
    return op(op(op(r0, r1), r2), r3);
}

}  // namespace dl
