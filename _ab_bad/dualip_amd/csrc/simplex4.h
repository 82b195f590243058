// simplex4.h -- segment machinery of the 256-wide tile: FOUR consecutive non-zeros per lane.
//
// Element e of a tile lives in lane e / 4, slot e % 4 (what a 16-byte load per lane delivers).  Column boundaries are
// four wave-uniform 64-bit masks H[j] (bit L <=> element 4L+j starts a column); every per-lane predicate the segmented
// reductions need is derived from them on the SCALAR unit (64-bit mask arithmetic) and consumed as a v_cndmask condition
// through __builtin_amdgcn_inverse_ballot_w64 -- no per-lane integer work at all.
//
// Segmented all-reduce (every element receives the reduction over its own column):
//   1. in-lane forward pass over the 4 slots                                   (3 ops + 3 selects)
//   2. cross-lane segmented scan of the lane's open tail (DPP, as in wave.h)   (6 DPP ops + 6 selects)
//   3. carry from the previous lane (DPP wave_shr:1), applied to the slots before the lane's first head
//   4. in-lane backward pass; columns that end in a later lane fetch their total with ONE ds_bpermute
// = ~38 VALU + 1 LDS op per 256 elements (the one-element-per-lane tile spends 4 x 16).
#pragma once
#include "common.h"
#include "simplex.h"
#include "wave.h"

namespace dl {

constexpr int kSlots = 4;
constexpr int kTile4 = 64 * kSlots;

__device__ __forceinline__ bool lane_bit(uint64_t uniform_mask) { return __builtin_amdgcn_inverse_ballot_w64(uniform_mask); }

constexpr int DPP_WAVE_SHR1 = 0x138;

// lanes whose position inside their 16-lane DPP row is >= o
constexpr uint64_t rows_ge(int o) {
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l)
        if ((l & 15) >= o) m |= 1ull << l;
    return m;
}
constexpr uint64_t kR1 = rows_ge(1), kR2 = rows_ge(2), kR4 = rows_ge(4), kR8 = rows_ge(8);
constexpr uint64_t kRows13 = 0xFFFF0000FFFF0000ull;  // lanes 16-31 and 48-63
constexpr uint64_t kUpper = 0xFFFFFFFF00000000ull;   // lanes 32-63

// Wave-uniform masks of one tile.
struct Seg4 {
    uint64_t H[kSlots];  // element 4L+j starts a column
    uint64_t N[kSlots];  // no head in slots 0..j of the lane: the previous lane's carry applies to slot j
    uint64_t O[kSlots];  // the column of slot j runs past the end of the lane: its total comes from a later lane
    uint64_t P1, P2, P4, P8, PA, PB;  // predicates of the cross-lane scan over lanes (segments start at lanes with a head)
    uint64_t G;          // lanes in which a column that entered from the previous lane ends
};

__device__ __forceinline__ Seg4 make_seg4(const uint64_t (&H)[kSlots]) {
    Seg4 s;
#pragma unroll
    for (int j = 0; j < kSlots; ++j) s.H[j] = H[j];
    s.N[0] = ~H[0];
    s.N[1] = s.N[0] & ~H[1];
    s.N[2] = s.N[1] & ~H[2];
    s.N[3] = s.N[2] & ~H[3];
    const uint64_t open = ~(H[0] >> 1) & ~(1ull << 63);  // lane L+1 exists and does not start with a head
    s.O[3] = open;
    s.O[2] = s.O[3] & ~H[3];
    s.O[1] = s.O[2] & ~H[2];
    s.O[0] = s.O[1] & ~H[1];
    const uint64_t F = H[0] | H[1] | H[2] | H[3] | 1ull;  // lanes containing a head
    // "a head among lanes l-o+1 .. l"
    const uint64_t S2 = F | (F << 1);
    const uint64_t S4 = S2 | (S2 << 2);
    const uint64_t S8 = S4 | (S4 << 4);
    s.P1 = ~F & kR1;
    s.P2 = ~S2 & kR2;
    s.P4 = ~S4 & kR4;
    s.P8 = ~S8 & kR8;
    // "a head among the lanes of my 16-row up to me" / "... of my 32-half up to me"
    uint64_t T = F;
    T |= (T << 1) & kR1;
    T |= (T << 2) & kR2;
    T |= (T << 4) & kR4;
    T |= (T << 8) & kR8;
    s.PA = ~T & kRows13;
    // rows 1 and 3 additionally see every head of the row before them: smear bit 15 / 47 of T over the next 16 lanes
    uint64_t Y = (T & 0x0000800000008000ull) << 1;
    Y |= Y << 1;
    Y |= Y << 2;
    Y |= Y << 4;
    Y |= Y << 8;
    const uint64_t U = T | Y;
    s.PB = ~U & kUpper;
    s.G = (H[1] | H[2] | H[3]) | (H[0] >> 1) | (1ull << 63);
    return s;
}

// cross-lane segmented inclusive scan with wave-uniform predicate masks.  All reduced quantities here are >= 0
// (clamped values, indicator counts), so 0 is the identity of both OpAdd and OpMax and the zero-fill DPP form applies.
template <class T, class Op>
__device__ __forceinline__ T lane_scan4(T x, const Seg4& s, Op op) {
    T t;
    t = dpp_mov0<DPP_ROW_SHR1, 0xf>(x);
    x = lane_bit(s.P1) ? op(x, t) : x;
    t = dpp_mov0<DPP_ROW_SHR2, 0xf>(x);
    x = lane_bit(s.P2) ? op(x, t) : x;
    t = dpp_mov0<DPP_ROW_SHR4, 0xf>(x);
    x = lane_bit(s.P4) ? op(x, t) : x;
    t = dpp_mov0<DPP_ROW_SHR8, 0xf>(x);
    x = lane_bit(s.P8) ? op(x, t) : x;
    t = dpp_mov0<DPP_ROW_BCAST15, 0xa>(x);
    x = lane_bit(s.PA) ? op(x, t) : x;
    t = dpp_mov0<DPP_ROW_BCAST31, 0xc>(x);
    x = lane_bit(s.PB) ? op(x, t) : x;
    return x;
}

// Every element receives the reduction over its column.  end_lane: lane holding the end of the column that is open
// at this lane's end (from end_lane4()).
// (values must be >= 0, see lane_scan4)
template <class T, class Op>
__device__ __forceinline__ void seg_allreduce4(const T (&u)[kSlots], const Seg4& s, int end_lane, Op op, T (&tot)[kSlots]) {
    T f[kSlots];
    f[0] = u[0];
#pragma unroll
    for (int j = 1; j < kSlots; ++j) f[j] = lane_bit(s.H[j]) ? u[j] : op(f[j - 1], u[j]);
    const T c = lane_scan4(f[kSlots - 1], s, op);
    const T carry = dpp_mov0<DPP_WAVE_SHR1, 0xf>(c);
#pragma unroll
    for (int j = 0; j < kSlots; ++j) f[j] = lane_bit(s.N[j]) ? op(carry, f[j]) : f[j];
    // backward: value at the last element of the slot's column inside this lane
    T b[kSlots];
    b[kSlots - 1] = f[kSlots - 1];
#pragma unroll
    for (int j = kSlots - 2; j >= 0; --j) b[j] = lane_bit(s.H[j + 1]) ? f[j] : b[j + 1];
    const T r = bperm(end_lane, b[0]);
#pragma unroll
    for (int j = 0; j < kSlots; ++j) tot[j] = lane_bit(s.O[j]) ? r : b[j];
}

// ---- SUM reductions without selects ----
// For sums, "take the neighbour's value if the predicate holds" is x + p * t with p in {0, 1}: ONE fused multiply-add
// (exactly round(x + t) or x) instead of an add and a v_cndmask, and the DPP move folds into it (v_fmac_f32_dpp).  The 0/1
// masks are built once per tile (13 selects) and reused by every sum / count reduction of the tile (>= 2).  Values must be
// finite (an infinity times 0 would leak a NaN into the neighbouring column): simplex_tile4 clamps at FLT_MAX.
template <class T>
struct SegMul4 {
    T nh[kSlots];  // slot j does not start a column (j >= 1)
    T nN[kSlots];  // the previous lane's carry applies to slot j
    T p1, p2, p4, p8, pa, pb;
};
template <class T>
__device__ __forceinline__ SegMul4<T> make_segmul4(const Seg4& s) {
    SegMul4<T> m;
    m.nh[0] = (T)0;
#pragma unroll
    for (int j = 1; j < kSlots; ++j) m.nh[j] = lane_bit(s.H[j]) ? (T)0 : (T)1;
#pragma unroll
    for (int j = 0; j < kSlots; ++j) m.nN[j] = lane_bit(s.N[j]) ? (T)1 : (T)0;
    m.p1 = lane_bit(s.P1) ? (T)1 : (T)0;
    m.p2 = lane_bit(s.P2) ? (T)1 : (T)0;
    m.p4 = lane_bit(s.P4) ? (T)1 : (T)0;
    m.p8 = lane_bit(s.P8) ? (T)1 : (T)0;
    m.pa = lane_bit(s.PA) ? (T)1 : (T)0;
    m.pb = lane_bit(s.PB) ? (T)1 : (T)0;
    return m;
}
__device__ __forceinline__ float fma_exact(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_exact(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <class T>
__device__ __forceinline__ T lane_scan4_sum(T x, const SegMul4<T>& m) {
    x = fma_exact(dpp_mov0<DPP_ROW_SHR1, 0xf>(x), m.p1, x);
    x = fma_exact(dpp_mov0<DPP_ROW_SHR2, 0xf>(x), m.p2, x);
    x = fma_exact(dpp_mov0<DPP_ROW_SHR4, 0xf>(x), m.p4, x);
    x = fma_exact(dpp_mov0<DPP_ROW_SHR8, 0xf>(x), m.p8, x);
    x = fma_exact(dpp_mov0<DPP_ROW_BCAST15, 0xa>(x), m.pa, x);
    x = fma_exact(dpp_mov0<DPP_ROW_BCAST31, 0xc>(x), m.pb, x);
    return x;
}
// the same scan over two independent values.  float: the compiler does not fold the DPP move into v_fmac_f32, so the
// six steps are written out -- x += dpp(x) * p is ONE instruction per value (lanes without a source add 0 * p).  The
// s_nop supplies, together with the other chain's instruction, the two wait states a DPP read needs after a VALU write.
template <class T>
__device__ __forceinline__ void lane_scan4_sum2(T& xa, T& xb, const SegMul4<T>& m) {
    xa = lane_scan4_sum(xa, m);
    xb = lane_scan4_sum(xb, m);
}
#define DL_FMAC_DPP2(ctrl, mask)                                                                            \
    asm("s_nop 0\n\tv_fmac_f32_dpp %0, %0, %2 " ctrl "\n\tv_fmac_f32_dpp %1, %1, %2 " ctrl : "+v"(xa), "+v"(xb) : "v"(mask))
template <>
__device__ __forceinline__ void lane_scan4_sum2<float>(float& xa, float& xb, const SegMul4<float>& m) {
    asm("s_nop 1" : "+v"(xa), "+v"(xb));  // (the compiler's hazard recogniser does not look inside inline assembly)
    DL_FMAC_DPP2("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", m.p1);
    DL_FMAC_DPP2("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1", m.p2);
    DL_FMAC_DPP2("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1", m.p4);
    DL_FMAC_DPP2("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1", m.p8);
    DL_FMAC_DPP2("row_bcast:15 row_mask:0xa bank_mask:0xf", m.pa);
    DL_FMAC_DPP2("row_bcast:31 row_mask:0xc bank_mask:0xf", m.pb);
    asm("s_nop 1" : "+v"(xa), "+v"(xb));
}
#undef DL_FMAC_DPP2

// Two sums at once (the simplex needs the sum and the size of the support): every element receives both totals of its
// column.  The two chains are independent, so their instructions interleave and share the LDS wait.
template <class T>
__device__ __forceinline__ void seg_allreduce4_sum2(const T (&a)[kSlots], const T (&b)[kSlots], const Seg4& s, const SegMul4<T>& m, int end_lane,
                                                    T (&ta)[kSlots], T (&tb)[kSlots]) {
    T fa[kSlots], fb[kSlots];
    fa[0] = a[0];
    fb[0] = b[0];
#pragma unroll
    for (int j = 1; j < kSlots; ++j) {
        fa[j] = fma_exact(fa[j - 1], m.nh[j], a[j]);
        fb[j] = fma_exact(fb[j - 1], m.nh[j], b[j]);
    }
    T xa = fa[kSlots - 1], xb = fb[kSlots - 1];
    lane_scan4_sum2(xa, xb, m);
    const T ca = dpp_mov0<DPP_WAVE_SHR1, 0xf>(xa);
    const T cb = dpp_mov0<DPP_WAVE_SHR1, 0xf>(xb);
#pragma unroll
    for (int j = 0; j < kSlots; ++j) {
        fa[j] = fma_exact(ca, m.nN[j], fa[j]);
        fb[j] = fma_exact(cb, m.nN[j], fb[j]);
    }
    T ba[kSlots], bb[kSlots];
    ba[kSlots - 1] = fa[kSlots - 1];
    bb[kSlots - 1] = fb[kSlots - 1];
#pragma unroll
    for (int j = kSlots - 2; j >= 0; --j) {
        ba[j] = lane_bit(s.H[j + 1]) ? fa[j] : ba[j + 1];
        bb[j] = lane_bit(s.H[j + 1]) ? fb[j] : bb[j + 1];
    }
    const T ra = bperm(end_lane, ba[0]);
    const T rb = bperm(end_lane, bb[0]);
#pragma unroll
    for (int j = 0; j < kSlots; ++j) {
        ta[j] = lane_bit(s.O[j]) ? ra : ba[j];
        tb[j] = lane_bit(s.O[j]) ? rb : bb[j];
    }
}

// first lane above `lane` in which an entering column ends (bit 63 of G is always set)
__device__ __forceinline__ int end_lane4(const Seg4& s, const LaneConst& c) {
    const uint32_t glo = (uint32_t)s.G & c.gt_lo, ghi = (uint32_t)(s.G >> 32) & c.gt_hi;
    const int lo = __ffs((int)glo) - 1;
    const int hi = 32 + __ffs((int)ghi) - 1;
    const int e = glo ? lo : hi;
    return ghi | glo ? e : 63;
}

// Simplex projection of every column of a 256-element tile (see simplex.h for the algorithm and the reference lines).
//   v must already be 0 in slots that hold no element of the tile (they form dummy segments of zeros);
//   every slot of x is written.
// Two reduction rounds serve most tiles: MAX (theta_0 = max - z), then SUM and COUNT over {u > theta_0} issued together
// (two independent dependency chains, one LDS wait).  feasible <=> sum <= z + 1e-6 (simplex.py:153-158); vertex <=>
// count == 1 (simplex.py:177-193); count == 2 => theta = (sum - z)/2 is final (the runner-up stays above it exactly
// when it is above max - z); larger supports run monotone Newton (Michelot) passes.  A pass is kept cheap -- late in a
// solve almost every tile holds a column that needs two or three: the state of a column is (S, C) = (sum of its support
// - z, size of its support), membership is tested as u * C > S (no division inside the loop), a pass is only run when a
// ballot says some member dropped out, and theta = S / C and x are formed once, after the loop.
// Few boolean masks are kept alive on purpose: every per-slot flag is an SGPR pair and the kernel is SGPR-starved.
template <class T>
__device__ __forceinline__ void simplex_tile4(const T (&v)[kSlots], const Seg4& s, const ProjT<T>& pj, const LaneConst& lc, T (&x)[kSlots],
                                              const int32_t* eq_row = nullptr) {
    const int el = end_lane4(s, lc);
    T u[kSlots], th0[kSlots], S[kSlots], C[kSlots], inu[kSlots], ind[kSlots];
#pragma unroll
    for (int j = 0; j < kSlots; ++j) u[j] = relu_finite(v[j]);  // max(v, 0), and no infinity (see SegMul4)
    {
        T v1[kSlots];
        seg_allreduce4(u, s, el, OpMaxNonNeg(), v1);
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            th0[j] = (T)(v1[j] - pj.z);
            const bool in = u[j] > th0[j];
            inu[j] = in ? u[j] : (T)0;
            ind[j] = in ? (T)1 : (T)0;
        }
    }
    const SegMul4<T> sm = make_segmul4<T>(s);
    T sumA[kSlots], cnt[kSlots];
    seg_allreduce4_sum2(inu, ind, s, sm, el, sumA, cnt);
    // Column state (S, C): theta = S / C.  "Keep the clamped values" is encoded as (0, 2) -- theta = 0 returns u itself --
    // and C == 1 marks a vertex, so no per-slot flag besides `act` has to live across the Newton loop.
    bool act[kSlots];
    const bool ineq = pj.kind == DL_PROJ_SIMPLEX;
    if (!eq_row) {
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            const bool keep = ineq && !(sumA[j] > pj.ztol);
            act[j] = !keep && cnt[j] > (T)2;
            S[j] = keep ? (T)0 : (T)(sumA[j] - pj.z);
            C[j] = keep ? (T)2 : cnt[j];
        }
    } else {
        // simplex_eq in the reference-compatibility mode (wave-uniform, cold; see eq_bucket in simplex.h): a column whose
        // clamped entries sum to less than z has theta < 0 -- its whole length is the support (count = length) and so
        // are the padding zeros of the reference's block, so the deficit is divided by the block height L
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            const int len = (int)cnt[j];
            const T L = (T)eq_row[eq_bucket(len > 0 ? len : 1)];
            const bool padded = sumA[j] < pj.z;
            act[j] = cnt[j] > (T)2 && !padded;
            S[j] = (T)(sumA[j] - pj.z);
            C[j] = padded ? L : cnt[j];
        }
    }
    // (tiles without a column whose support exceeds two elements -- most tiles early in a solve -- skip even the first test)
    if (__any(act[0] || act[1] || act[2] || act[3]))
    for (int it = 0; it < kTile4; ++it) {
        bool in[kSlots], dropped = false;
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            in[j] = ind[j] != (T)0 && (T)(u[j] * C[j]) > S[j];  // still a member: u > S / C (supports only shrink, which also
                                                               // rules out a rounding-induced leave / re-enter cycle)
            dropped = dropped || (act[j] && ind[j] != (T)0 && !in[j]);
        }
        if (!__any(dropped)) break;  // no support changed: every (S, C) is final
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            ind[j] = in[j] ? (T)1 : (T)0;
            inu[j] = in[j] ? u[j] : (T)0;
        }
        seg_allreduce4_sum2(inu, ind, s, sm, el, sumA, cnt);
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            const bool changed = act[j] && cnt[j] != C[j] && cnt[j] != (T)0;
            S[j] = changed ? (T)(sumA[j] - pj.z) : S[j];
            C[j] = changed ? cnt[j] : C[j];
            act[j] = changed;
        }
    }
#pragma unroll
    for (int j = 0; j < kSlots; ++j) {
        const T th = div_exactish(S[j], C[j]);
        const T xg = relu((T)(u[j] - th));
        const T xv = (u[j] > th0[j]) ? pj.z : (T)0;  // vertex: z at the maximum, 0 elsewhere (exact z, as the reference)
        x[j] = C[j] == (T)1 ? xv : xg;
    }
}

}  // namespace dl
