// fused_common.h -- what the fused pass (fused4_kernel.h) and its slices (sell.h) share: kernel arguments, the exact fixed-point scatter,
// SGPR-base addressing, the single-column ("long tile") walker, the prologue and the epilogue.
#pragma once
#include "agd_step.h"
#include "common.h"
#include "simplex.h"
#include "wave.h"

namespace dl {

template <class T>
struct FusedArgs {
    const uint32_t* __restrict__ tiles32;  // window descriptors (desc_words dwords each)
    const void* __restrict__ rowidx;
    const T* __restrict__ a;
    const T* __restrict__ c;
    const T* __restrict__ lambda;
    T* __restrict__ x_out;
    const ProjDev* __restrict__ projs;
    long long* __restrict__ partial;     // [n_wg][mpad] (GRAD_LDS) or [mpad] (global atomics, pre-zeroed)
    long long* __restrict__ partial_scal;  // [n_wg][2]: c.x and sum x^2 of the workgroup in 64-bit fixed point (exponent shift_out[1])
    int* __restrict__ shift_out;         // fixed-point exponents chosen for this launch: [0] gradient rows, [1] the two scalar sums
    double gamma;
    double amax, cmax;                   // max |a|, max |c|
    double xmax_bounded;                 // max |x| any bounded projection present can return (box bounds, simplex z)
    double pmax_unbounded;               // max |bound| of one-sided projections present
    double row_count_max;                // largest number of non-zeros in one row (of this shard)
    int has_unbounded;                   // some column's projection does not bound |x| (cone / none): use the |v| bound
    int64_t m;
    int64_t mpad;
    int64_t nnz;
    int slab32;                          // 1: the gradient slabs are int32 low words [n_wg][mpad] (+ high words on overflow): common.h
    double slab_abound;                  //    sum of |a| a workgroup's share of one row is expected to stay below (times the bound of x: the grid's 2^30)
    int32_t* slab_hi;                    //    [n_wg][mpad] high words
    unsigned long long* slab_ovf;        //    [n_wg + 1] epochs (common.h)
    unsigned long long slab_epoch;       //    this launch's
    const uint32_t* slab_wide_bits;      //    [1024] bit u of word t: row t + 1024 u is a WIDE row (common.h: its high word is sent in every launch and is no overflow); or null
    const int32_t* slab_wide_list;       //    the wide rows' indices ...
    int32_t n_wide;                      //    ... and their number
    int32_t n_proj;
    uint32_t n_tiles;   // layout 4: window tiles of the launch (cyclic schedule); descriptor n_tiles is all-zero
    uint32_t n_long;    // layout 4: single-column tiles, descriptors n_tiles + 1 ... n_tiles + n_long
    uint32_t n_xlong;   // layout 4: very long single-column tiles (walked by a whole workgroup), descriptors after those
    uint32_t desc_words;           // layout 4: dwords per window descriptor: 12, or 2 (compact: { W[31:0] ; W[39:32] | hi << 8 | lo << 17 | proj id << 20 })
    const uint32_t* __restrict__ long32;  // layout 4: the single-column tiles' descriptors (12 dwords each)
    const int32_t* balance;               // layout 4: the window tiles' weighted deal (Deal: rounds per workgroup + tables), or null (same for all)
    unsigned long long* bal_stamps;       // layout 4: [n_wg][4] wall-clock stamps the balance kernel reads, or null
    const int32_t* sell_bal;              // layout 4, first binary: two-phase deal of the one-lane slices (common.h), or null (one even deal)
    int ablate;  // developer-only timing ablations (DUALIP_HIP_ABLATE, libdualip_hip_dev.so only): bits 8, 32, 128, 1024, 12-13, 14-17 -- see their uses
    unsigned long long* timeline;  // developer-only: [n_wg][kTimelineSlots] wall-clock stamps (common.h) or null
    int64_t m_hot;                 // hot-rows plan: rows < m_hot (renumbered by frequency) have their GRADIENT accumulator in LDS; 0 = every row does
    int64_t m_lam;                 // hot-rows plan: rows < m_lam have their DUAL entry in LDS (m_hot <= m_lam <= m; m: no tile gathers from L2)
    long long* cold_grad;          // hot-rows plan: int64 accumulators of the rows >= m_hot (global atomics, pre-zeroed): [kColdCopies][mpad]
    int cold_per_xcd;              // 1: every XCD adds to its own copy with L2-local atomics; 0: copy 0 only, device-scope atomics
    const int32_t* eq_heights;     // simplex_eq reference-compatibility mode: [n_proj][kEqBuckets] padded block heights, or null (exact)
    // fairness pair (dl_matching_set_fairness): rows m-2 / m-1 carry +f_k / -f_k on EVERY non-zero k
    const T* fair;                 // f, in the order of a / c, or null
    const T* lambda_orig;          // the caller's dual vector (g.lambda is the renumbered copy under the hot-rows plan)
    double* partial_fair;          // [n_wg]: sum f_k x_k of the workgroup
    double fair_max;               // max |f| (0 without the pair): enters the |v| bound of unbounded projections
    // column-per-lane slices (sell.h)
    const uint32_t* __restrict__ sell_desc;
    const uint8_t* __restrict__ sell_len;
    const uint64_t* __restrict__ sell_colstart;
    const T* __restrict__ sell_a;
    const T* __restrict__ sell_c;
    const void* __restrict__ sell_r;
    const T* __restrict__ sell_f;
    uint32_t n_sell;               // one-lane-per-column slices (sell_desc)
    const uint32_t* sell_lane_desc;  // slices with K = 2 .. 32 lanes per column (sell.h: sell_lanes_loop), their own table ...
    uint32_t n_sell_lanes;           // ... and count (cold: read from the kernel arguments)
    const uint32_t* sell_lane_begin; // [workgroups + 1]: workgroup w walks table entries sell_lane_begin[w] .. sell_lane_begin[w + 1]
    // the previous iteration's optimiser step, applied in this launch's prologue (agd_step.h): do_apply != 0 => `lambda` is not read,
    // every workgroup forms the new iterate from apply.{x, g_new, y} and stages THAT; workgroup 0 also stores it (and the state / log)
    int do_apply;
    ApplyArgs<T> apply;
};

// ---- the cyclic deal of window tiles to wavefronts, with a per-WORKGROUP number of rounds ----
// Unweighted, wavefront W of the S = 16 * workgroups of a launch takes slots W, W + S, W + 2S, ...  The workgroups do not stream at
// the same speed: the eight XCDs differ (workgroup w runs on XCD w mod 8; measured at 100M entities, all-box: the XCDs' mean finish
// times spread over 8.5 %, the odd XCDs late) and inside an XCD single workgroups are persistently early or late (12.5M entities:
// +-4 % of the launch, correlation 0.9 from launch to launch) -- and the launch ends with the slowest.  So every workgroup w gets
// its own number of rounds n_w: in round k only the workgroups with n_w > k take part, ranked by workgroup.  Slot of wavefront v
// of workgroup w in round k:   16 sum_w' min(k, n_w')  +  16 #{w' < w: n_w' > k}  +  v   -- a bijection onto the slots for any
// table.  For k < min n_w that is k S + W: the common case costs nothing; for the last rounds the two terms come from tables the
// balance kernel (matching_kernels.hip) writes next to the rounds.
// Table layout (int32 words): [0] min n_w, [1] J = max n_w - min n_w (<= kBalTail), [2..3] unused, [4 .. 4 + G) n_w,
// then J offsets 16 sum_w' min(k, n_w') for k = min + j, then J x G ranks #{w' < w: n_w' > k}.
struct Deal {
    uint32_t n_mine, n_min;
};
__device__ __forceinline__ Deal make_deal(const int32_t* tab) {
    // (layout-4 handles always have a table -- an even deal is min = every n_w = 2^31 - 1 -- so there is no null case to branch on.
    //  It lives in global memory the compiler cannot prove constant: its loads are vector loads, and without the readfirstlane
    //  every slot computation downstream runs on the vector unit -- +20 VALU per window tile, measured)
    Deal d;
    d.n_min = (uint32_t)__builtin_amdgcn_readfirstlane(tab[0]);
    d.n_mine = (uint32_t)__builtin_amdgcn_readfirstlane(tab[4 + blockIdx.x]);
    return d;
}
// slot of this wavefront's k-th tile among N; >= N: none (and none after it).  32-bit arithmetic: N < 2^31 and a wavefront stops at
// its first slot >= N, so k S stays below N + 2 S.
__device__ __forceinline__ uint32_t deal_slot(const Deal& d, const int32_t* tab, uint32_t k, uint32_t N) {
    const uint32_t n_wg = gridDim.x, wg = blockIdx.x;
    const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (k < d.n_min) {  // (also the whole of an unweighted deal)
        // Round k hands slots [k S, (k + 1) S) out workgroup-major: the sixteen wavefronts of a workgroup walk sixteen neighbouring tiles.  The LAST
        // round of an even deal is partial -- N - k_last S tiles -- and workgroup-major it gives the first workgroups a whole extra tile per
        // wavefront and the others none (1M entities: ten rounds, workgroups 0 .. 140 carry 160 tiles, the rest 144, and the launch ends with
        // the former); wavefront-major -- slot k S + v G + w -- every workgroup gets its share of what is left (a few wavefronts each).
        // (recognised without any state carried through the hot loop -- the round's first slot + S runs past N -- the kernel has no register to spare)
        const uint32_t base = k * n_wg * (uint32_t)kFusedWaves;
        const uint32_t q = base + (base + n_wg * (uint32_t)kFusedWaves > N ? v * n_wg + wg : wg * (uint32_t)kFusedWaves + v);
        return q < N ? q : N;
    }
    if (k >= d.n_mine) return N;
    uint32_t j = k - d.n_min;  // (< J <= kBalTail: n_mine <= n_min + J)
    j = j < (uint32_t)kBalTail ? j : (uint32_t)kBalTail - 1u;  // (keeps the two loads below inside the table wherever the compiler places them)
    const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane(tab[4 + n_wg + j]);
    const uint32_t rank = (uint32_t)__builtin_amdgcn_readfirstlane(tab[4 + n_wg + kBalTail + j * n_wg + wg]);
    const uint32_t q = off + rank * (uint32_t)kFusedWaves + v;
    return q < N ? q : N;
}

// The cold paths re-read the kernel arguments from the kernarg segment (they sit at offset 0) instead of keeping a dozen
// pointers alive in SGPRs across the hot loop.
template <class T>
__device__ __forceinline__ const FusedArgs<T>& kernarg_args(const FusedArgs<T>& fallback) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const FusedArgs<T>*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    return fallback;
#endif
}

// a x -> 64-bit fixed point (round to nearest at 2^-shift) and integer atomic add: exact, order independent.
// float : 1.5 * 2^52 trick -- for |ax * 2^shift| < 2^51 the integer sits in the mantissa of the fma result (3 VALU);
// double: full 62-bit conversion (the 2^-50 grid of the trick would be coarser than the values themselves).
template <class T>
struct FixedBits {
    static constexpr int value = 50;
};
template <>
struct FixedBits<double> {
    static constexpr int value = 61;
};
__device__ __forceinline__ long long to_fixed(float ax, double scale) {
    const double magic = 6755399441055744.0;
    const double d = fma((double)ax, scale, magic);
    return __double_as_longlong(d) - __double_as_longlong(magic);
}
__device__ __forceinline__ long long to_fixed(double ax, double scale) { return __double2ll_rn(ax * scale); }

// ---- c.x and sum x^2 in fixed point too ----
// The gradient is summed in integers, so it does not depend on which wavefront walked which tile.  The two scalar sums used to be
// per-lane doubles; since the deal of the window tiles adapts to measured timings (Deal, above) their last bits -- and with them
// the logged dual objective -- followed the timings.  Now every tile / slice / long column contributes ONE rounded integer per
// lane: the workgroup partials, and the totals the optimiser logs, are identical run to run and for any deal.
// Exponent: |total| <= nnz * q with q = max(cmax * xmax, xmax^2) must stay below 2^62, and one contribution (at most 32 elements
// of a lane: four slots of a window, 24 steps of a slice) below the 2^51 the 1.5 * 2^52 conversion trick holds.
__device__ __forceinline__ int scalar_shift(double nnz, double cmax, double xmax) {
    const double q = (cmax * xmax > xmax * xmax) ? cmax * xmax : xmax * xmax;
    // (max |c| and the projection bounds are finite -- the handle refuses arrays with inf / NaN -- but their product with nnz may not be:
    //  an overflowing bound takes the coarsest grid instead of leaving the exponent at 0, which would put garbage into the integer sums)
    int e_tot = 0, e_one = 0;
    const double tot = nnz * q, one = 32.0 * q;
    if (tot > 0.0 && tot < 1.7e308) (void)frexp(tot, &e_tot);
    else if (tot > 0.0) e_tot = 1100;
    if (one > 0.0 && one < 1.7e308) (void)frexp(one, &e_one);
    else if (one > 0.0) e_one = 1100;
    int sh = 62 - e_tot;
    sh = sh < 50 - e_one ? sh : 50 - e_one;
    return sh > 1000 ? 1000 : (sh < -1000 ? -1000 : sh);
}
// The two accumulators of a lane.  A contribution is added as the RAW bit pattern of fma(v, 2^shift, 1.5 * 2^52) -- whose low 51 bits
// are the rounded integer (|v * 2^shift| < 2^51 by the choice above) on top of the constant's own pattern -- and the constant is
// taken off once, `n` times, when the lane hands its sums over (arithmetic mod 2^64): a conversion, a fused multiply-add and one
// 64-bit add per sum and tile.  `n` counts the add sites passed; every lane of a wavefront passes the same ones, so it is a scalar.
struct FxAcc {
    long long obj = 0, ssq = 0;
    uint32_t n = 0;
};
constexpr unsigned long long kFxMagicBits = 0x4338000000000000ull;  // bit pattern of 1.5 * 2^52
template <class T>
__device__ __forceinline__ void fx_add(FxAcc& acc, T o, T q, double scale2) {
    const double magic = 6755399441055744.0;
    acc.obj += __double_as_longlong(fma((double)o, scale2, magic));
    acc.ssq += __double_as_longlong(fma((double)q, scale2, magic));
    acc.n += 1u;
}
// a whole long column's sums of one lane (can be large: full conversion, no constant to take off)
__device__ __forceinline__ void fx_add_wide(FxAcc& acc, double o, double q, double scale2) {
    acc.obj += __double2ll_rn(o * scale2);
    acc.ssq += __double2ll_rn(q * scale2);
}
__device__ __forceinline__ void fx_finish(FxAcc& acc) {
    const unsigned long long off = (unsigned long long)acc.n * kFxMagicBits;
    acc.obj = (long long)((unsigned long long)acc.obj - off);
    acc.ssq = (long long)((unsigned long long)acc.ssq - off);
    acc.n = 0;
}

template <class T>
__device__ __forceinline__ void scatter_fixed(long long* acc, uint32_t row, T ax, double scale) {
    atomicAdd(reinterpret_cast<unsigned long long*>(acc) + row, (unsigned long long)to_fixed(ax, scale));  // ds_add_u64 / global_atomic_add_x2
}

// The same with the address space SPELLED OUT.  Under the hot-rows plan a scatter is `row < m_hot ? LDS accumulator : global accumulator`;
// written with generic pointers the compiler folds the two arms into a select of the address and ONE flat_atomic_add_x2 -- every
// scatter of every hot-rows kernel went through the flat path (aperture check, both counters, a fraction of ds_add_u64's rate): found
// in the device assembly in round 4 after the timing ablation "no scatter" gave 8 of 56 us on the MovieLens shape.
template <class T>
__device__ __forceinline__ void scatter_fixed_lds(long long* acc, uint32_t row, T ax, double scale) {
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    lds_u64* p = (lds_u64*)(reinterpret_cast<unsigned long long*>(acc) + row);
    (void)__hip_atomic_fetch_add(p, (unsigned long long)to_fixed(ax, scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_u64
}
template <class T>
__device__ __forceinline__ void scatter_fixed_global(long long* acc, uint32_t row, T ax, double scale) {
    typedef __attribute__((address_space(1))) unsigned long long glb_u64;
    glb_u64* p = (glb_u64*)(reinterpret_cast<unsigned long long*>(acc) + row);
    (void)__hip_atomic_fetch_add(p, (unsigned long long)to_fixed(ax, scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_atomic_add_x2 sc1
}
// the cold row's accumulator: this XCD's copy through its own L2 (workgroup scope: no sc1), or the shared copy at the memory side
template <class T>
__device__ __forceinline__ void scatter_fixed_cold(long long* xcd_copy, long long* shared, uint32_t row, T ax, double scale) {
    if (xcd_copy) {  // (wave-uniform)
        typedef __attribute__((address_space(1))) unsigned long long glb_u64;
        glb_u64* p = (glb_u64*)(reinterpret_cast<unsigned long long*>(xcd_copy) + row);
        (void)__hip_atomic_fetch_add(p, (unsigned long long)to_fixed(ax, scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        scatter_fixed_global(shared, row, ax, scale);
    }
}

template <class P>
__device__ __forceinline__ P byte_offset(P base, uint32_t bytes) {
    return reinterpret_cast<P>(reinterpret_cast<const char*>(base) + bytes);  // SGPR base + 32-bit VGPR offset addressing
}

// Long tile: one column too long for a window, walked by the whole wavefront.  The walk is latency bound (one wavefront,
// dependent loads), so it moves in batches of four 64-wide strides whose loads are all issued before the first is used;
// data are re-read (L2-hot) for every Newton pass.  Per-lane partial sums run over the strides in ascending order.
// WG = true: the same walk by the whole workgroup (columns of thousands of non-zeros: one wavefront walking them alone sets
// the critical path of the launch).  `lane` is then the thread index, strides are kFusedThreads wide, reductions go through
// `red` (>= kFusedWaves doubles of LDS) with workgroup barriers -- every thread of the workgroup must make the call.
template <class T, class RowT, bool LAM_LDS, bool WG = false, int KRB_WG = 8>
__device__ __forceinline__ void process_long_tile(const FusedArgs<T>& g, const ProjT<T> pj, uint64_t k0, uint64_t len, const T* lam_s, long long* gacc, T s,
                                              double scale, int lane, double& obj, double& ssq, const int32_t* eq_row = nullptr, int64_t m_hot = 0,
                                              double* red = nullptr, T sd = (T)0, double* fair_acc = nullptr, const uint32_t* desc = nullptr, int64_t m_lam = 0,
                                              long long* cold_xcd = nullptr) {
    if (m_lam < m_hot) m_lam = m_hot;  // (rows whose dual entry is in LDS; the hot-rows plan may stage more of them than gradient rows)
    constexpr int kLB = 4;
    constexpr uint32_t kStride = WG ? (uint32_t)kFusedThreads : 64u;
    // WG: two quantities through ONE exchange (a pass of the walker used to spend two barrier pairs on its sum and two on its count --
    // for a column of thousands of tied values, ten passes: the barriers, not the data, were its time); the same per-wavefront partials
    // summed in the same fixed order, so no bit changes.  `red` holds >= 2 kFusedWaves doubles.
    auto all_pair = [&](double& x, double& y, bool y_is_max) {
        x = wave_allreduce_dpp(x, OpAdd());
        y = y_is_max ? wave_allreduce_dpp(y, OpMax()) : wave_allreduce_dpp(y, OpAdd());
        if constexpr (WG) {
            __syncthreads();  // the previous exchange's readers are done
            if ((lane & 63) == 0) {
                red[lane >> 6] = x;
                red[kFusedWaves + (lane >> 6)] = y;
            }
            __syncthreads();
            x = red[0];
            y = red[kFusedWaves];
            for (int q = 1; q < kFusedWaves; ++q) {  // fixed order
                x += red[q];
                const double yq = red[kFusedWaves + q];
                y = y_is_max ? (yq > y ? yq : y) : y + yq;
            }
        }
    };
    const bool is_simplex = is_simplex_kind(pj.kind);
    // v = a * (-lambda/gamma) + (-c/gamma) for the elements o0 + lane + kStride u (ok[u]: inside the column)
    auto load_batch = [&](uint64_t o0, T (&av)[kLB], T (&cv)[kLB], uint32_t (&rv)[kLB], bool (&ok)[kLB], T (&v)[kLB]) {
#pragma unroll
        for (int u = 0; u < kLB; ++u) {
            const uint64_t o = o0 + (uint64_t)lane + (uint64_t)kStride * (uint64_t)u;
            ok[u] = o < len;
            const uint64_t k = k0 + (ok[u] ? o : len - 1);
            av[u] = g.a[k];
            cv[u] = g.c[k];
            rv[u] = (uint32_t)reinterpret_cast<const RowT*>(g.rowidx)[k];
        }
        // (hot-rows plan: the cold rows' dual entries are requested for the whole batch at once, unconditionally -- a hot lane reads
        //  lambda[0] -- and chosen by a select; a load under a per-element branch waits for its own L2 round trip at every join)
        T lg[kLB];
        const bool some_cold = LAM_LDS && m_hot > 0 && m_lam < g.m;  // (wave-uniform)
        if (some_cold) {
#pragma unroll
            for (int u = 0; u < kLB; ++u) lg[u] = g.lambda[(int64_t)rv[u] >= m_lam ? rv[u] : 0u];
        }
#pragma unroll
        for (int u = 0; u < kLB; ++u) {
            T lam;
            if (some_cold) {
                const bool cold = (int64_t)rv[u] >= m_lam;
                const T hot_val = lam_s[cold ? 0u : rv[u]];
                lam = cold ? (T)(s * lg[u]) : hot_val;
            } else {
                lam = LAM_LDS ? lam_s[rv[u]] : (T)(s * g.lambda[rv[u]]);
            }
            v[u] = (T)((T)(av[u] * lam) + (T)(s * cv[u]));
        }
        if (fair_acc) {  // fairness pair: + f_k * (-(lambda_K - lambda_{K+1}) / gamma)
#pragma unroll
            for (int u = 0; u < kLB; ++u) {
                const uint64_t o = o0 + (uint64_t)lane + (uint64_t)kStride * (uint64_t)u;
                v[u] = (T)(v[u] + (T)(sd * g.fair[k0 + (ok[u] ? o : len - 1)]));
            }
        }
    };
    T th = (T)0;
    bool projected = false, onehot = false;
    // columns of up to kRB strides keep their clamped values in registers across the Newton passes (-inf outside the column);
    // longer ones re-read the arrays (L2-hot) in every pass
    // (WG: KRB_WG strides of 1024.  The second binary asks for 16 -- at 8 the MovieLens shape's longest column, 9 254 non-zeros, re-read its
    //  arrays in every pass; the benchmark's binary stays at 8: with 16 its hot loop lost its last registers, 20 bytes of scratch, and the
    //  build refuses that)
    constexpr int kRB = WG ? KRB_WG : 16;
    const bool cached = len <= (uint64_t)kStride * kRB && !DL_ABLATE(g.ablate, 8);
    T vr[kRB];
    if (is_simplex) {
        T S = (T)0, v1 = (T)(-INFINITY);
        if (cached) {
#pragma unroll
            for (int bt = 0; bt < kRB / kLB; ++bt) {
#pragma unroll
                for (int u = 0; u < kLB; ++u) vr[bt * kLB + u] = (T)(-INFINITY);
                if ((uint64_t)bt * kStride * kLB < len) {
                    T av[kLB], cv[kLB], v[kLB];
                    uint32_t rv[kLB];
                    bool ok[kLB];
                    load_batch((uint64_t)bt * kStride * kLB, av, cv, rv, ok, v);
#pragma unroll
                    for (int u = 0; u < kLB; ++u) {
                        const T uu = tmax(v[u], (T)0);
                        S = ok[u] ? (T)(S + uu) : S;
                        v1 = ok[u] ? tmax(v1, uu) : v1;
                        vr[bt * kLB + u] = ok[u] ? uu : (T)(-INFINITY);
                    }
                }
            }
        } else {
            for (uint64_t o0 = 0; o0 < len; o0 += kStride * kLB) {
                T av[kLB], cv[kLB], v[kLB];
                uint32_t rv[kLB];
                bool ok[kLB];
                load_batch(o0, av, cv, rv, ok, v);
#pragma unroll
                for (int u = 0; u < kLB; ++u) {
                    const T uu = tmax(v[u], (T)0);
                    S = ok[u] ? (T)(S + uu) : S;
                    v1 = ok[u] ? tmax(v1, uu) : v1;
                }
            }
        }
        if constexpr (WG) {  // (wavefront partials combined in double, rounded once)
            double Sd = (double)S, v1d = (double)v1;
            all_pair(Sd, v1d, true);
            S = (T)Sd;
            v1 = (T)v1d;
        } else {
            S = wave_allreduce_dpp(S, OpAdd());
            v1 = wave_allreduce_dpp(v1, OpMax());
        }
        projected = (pj.kind == DL_PROJ_SIMPLEX_EQ) || S > pj.ztol;
        const bool padded = eq_row && pj.kind == DL_PROJ_SIMPLEX_EQ && S < pj.z;
        if (padded) {  // every entry and every padding zero is in the support: theta = (S - z) / L, final
            th = (T)((T)(S - pj.z) / (T)eq_row[eq_bucket((int)(len < 0x7fffffff ? len : 0x7fffffff))]);
        } else if (projected) {
            const T z = pj.z;
            th = tmax((T)(v1 - z), (T)((T)(S - z) / (T)len));
            long long cnt_prev = 0;
            for (int it = 0; it < 4096; ++it) {
                T sumA = (T)0;
                long long cntl = 0;
                if (cached) {
#pragma unroll
                    for (int i = 0; i < kRB; ++i) {
                        const bool in = vr[i] > th;
                        sumA = in ? (T)(sumA + vr[i]) : sumA;
                        cntl += in ? 1 : 0;
                    }
                } else {
                    for (uint64_t o0 = 0; o0 < len; o0 += kStride * kLB) {
                        T av[kLB], cv[kLB], v[kLB];
                        uint32_t rv[kLB];
                        bool ok[kLB];
                        load_batch(o0, av, cv, rv, ok, v);
#pragma unroll
                        for (int u = 0; u < kLB; ++u) {
                            const T uu = tmax(v[u], (T)0);
                            const bool in = ok[u] && uu > th;
                            sumA = in ? (T)(sumA + uu) : sumA;
                            cntl += in ? 1 : 0;
                        }
                    }
                }
                long long cntw;
                if constexpr (WG) {
                    double sd2 = (double)sumA, cd2 = (double)cntl;
                    all_pair(sd2, cd2, false);
                    sumA = (T)sd2;
                    cntw = (long long)cd2;
                } else {
                    sumA = wave_allreduce_dpp(sumA, OpAdd());
                    cntw = (long long)wave_allreduce_dpp((uint32_t)cntl, OpAdd());  // (a lane counts at most the column's strides: 32 bits)
                }
                if (it == 0 && cntw == 1) {
                    onehot = true;
                    break;
                }
                if (cntw == cnt_prev || cntw == 0) break;
                // Michelot's thresholds never decrease in exact arithmetic; enforcing it in floating point keeps the supports
                // nested, so the loop ends after at most `len` passes (without it a value within rounding of the threshold
                // can leave and re-enter the support for thousands of passes -- measured on ratings-like data with ties)
                th = tmax(th, (T)((T)(sumA - z) / (T)cntw));
                cnt_prev = cntw;
            }
        }
    }
    for (uint64_t o0 = 0; o0 < len; o0 += kStride * kLB) {
        T av[kLB], cv[kLB], v[kLB];
        uint32_t rv[kLB];
        bool ok[kLB];
        load_batch(o0, av, cv, rv, ok, v);
#pragma unroll
        for (int u = 0; u < kLB; ++u) {
            if (!ok[u]) continue;
            T x;
            if (is_simplex) {
                const T uu = tmax(v[u], (T)0);
                if (!projected) x = uu;
                else if (onehot) x = (uu > th) ? pj.z : (T)0;
                else x = tmax((T)(uu - th), (T)0);
            } else {
                x = project_pointwise(v[u], pj);
            }
            const T ax = (T)(av[u] * x);
            if (ax != (T)0) {
                if (m_hot > 0) {  // (hot-rows plan: gacc is the LDS accumulator)
                    if ((int64_t)rv[u] < m_hot) scatter_fixed_lds(gacc, rv[u], ax, scale);
                    else scatter_fixed_cold(cold_xcd, g.cold_grad, rv[u], ax, scale);
                } else {
                    scatter_fixed(gacc, rv[u], ax, scale);
                }
            }
            obj += (double)(T)(cv[u] * x);
            ssq += (double)(T)(x * x);
            if (fair_acc) *fair_acc += (double)(T)(g.fair[k0 + o0 + (uint64_t)lane + (uint64_t)kStride * (uint64_t)u] * x);
            if (g.x_out) {
                // where the primal goes: the column's place in the CALLER's order (descriptor words 4 / 5, re-read here: cold) -- the place
                // it is read from, unless dl_matching_own_inputs moved the column into the handle's pool
                const uint64_t kx = desc ? ((((uint64_t)desc[5] << 32) | desc[4]) & ((1ull << 40) - 1)) : k0;
                g.x_out[kx + o0 + (uint64_t)lane + (uint64_t)kStride * (uint64_t)u] = x;
            }
        }
    }
}


// ---- workgroup context shared by both tile layouts ----
template <class T>
struct WgCtx {
    long long* grad_s;
    T* lam_s;
    ProjT<T>* proj_s;
    double* red_s;
    long long* gacc;
    long long* cold;    // hot-rows plan: THIS XCD's array of cold-row accumulators (fused_prologue); null: the shared one, device-scope atomics
    T s;           // -1/gamma rounded once to the working precision (matching.py:136)
    double scale;  // 2^shift of the fixed-point gradient
    double scale2; // 2^shift of the fixed-point scalar sums (c.x, sum x^2)
};

// Prologue: carve LDS, stage -lambda/gamma, zero the private gradient, cache the projection table, choose the
// fixed-point exponent from max|lambda| (identical in every workgroup).
template <class T, bool LAM_LDS, bool GRAD_LDS>
__device__ __forceinline__ WgCtx<T> fused_prologue(const FusedArgs<T>& g, unsigned char* smem, int tid, int lane, int wave, int wg) {
    WgCtx<T> w;
    // layout: [gradient int64 m (GRAD_LDS)] [lambda T m (LAM_LDS)] [projection table] [scratch doubles]
    const int64_t m_lds = g.m_hot > 0 ? g.m_hot : g.m;  // rows that live in LDS (all of them unless the hot-rows plan is on)
    w.grad_s = reinterpret_cast<long long*>(smem);
    size_t off = GRAD_LDS ? (size_t)m_lds * 8 : 0;
    const int64_t m_lam = g.m_hot > 0 ? g.m_lam : g.m;   // rows whose dual entry is staged
    w.lam_s = reinterpret_cast<T*>(smem + off);
    off += LAM_LDS ? (size_t)m_lam * sizeof(T) : 0;
    off = (off + 15) / 16 * 16;
    w.proj_s = reinterpret_cast<ProjT<T>*>(smem + off);
    off += (size_t)kProjLds * sizeof(ProjT<T>);
    w.red_s = reinterpret_cast<double*>(smem + off);
    w.s = (T)(-1.0 / g.gamma);
    // The projection table's entries are requested FIRST (four wavefronts, one entry per thread): nothing depends on them until the table is
    // filled after the duals, so their round trip no longer stands behind the staging loop.  (Measured: no change of the prologue's length --
    // what follows "rows staged" there is the skew between the sixteen wavefronts, tools/timeline.py -- kept because it cannot be slower.)
    ProjDev pd_early;
    pd_early.kind = DL_PROJ_NONE;
    pd_early.p0 = 0.0;
    pd_early.p1 = 0.0;
    if (wave < kProjLds / 64 && tid < g.n_proj && tid < kProjLds - 1) {
        pd_early.kind = g.projs[tid].kind;
        pd_early.p0 = g.projs[tid].p0;
        pd_early.p1 = g.projs[tid].p1;
    }
    double lmax = 0.0;
    // The optimiser step of the previous iteration, if this launch carries it (agd_step.h; matching_kernels.hip: matching_can_fuse_apply decides):
    // every workgroup forms the new iterate itself instead of reading what a separate apply launch wrote -- one launch and one boundary less per
    // iteration, against a longer head of this launch.  Rounds 3-5: every wavefront derived the step and then requested its rows (two dependent
    // trips, +5.4 us of head): worth it only for small handles.  Requesting the rows before deriving the step IN EVERY WAVEFRONT (twelve per thread in
    // registers next to the step's 48) had made the launch 12 us longer.  Round 6, below: the two trips run side by side in different wavefronts.
    const bool applying = g.do_apply != 0;
    T a_stp = (T)0, a_bb = (T)0, a_omb = (T)0;
    int64_t applied_rows = 0;  // rows [0, applied_rows) of the new iterate are staged by the block below
    if (applying) {
        // Round 6: ONE wavefront derives the step while the other fifteen have the rows in flight.  Before, every wavefront derived it
        // (157 x 6 partial statistics each) and only then requested its rows: two dependent trips, the second queued behind sixteen copies of the
        // first -- wavefront 0 had its rows at 6.7 us, wavefront 15 at 8.9 us of a launch whose plain form has them at 2.8 / 3.4 us (tools/timeline.py).
        // Wavefront 0 takes the step (and, in workgroup 0, the state and the log row); wavefronts 1 .. 15 take the rows, 960 at a time.
        const ApplyArgs<T>& ap = kernarg_args(g).apply;
        constexpr int kRowThreads = kFusedThreads - 64;
        constexpr int kUA = 14;  // 14 x 960 = 13 440 rows in one round trip: more than a gradient that fits the LDS has
        const float bt = ap.beta[ap.iter - 1];
        T xx[kUA], gg[kUA], yy[kUA];
        bool eq[kUA];
        const int rt = tid - 64;
        if (wave != 0) {
#pragma unroll
            for (int u = 0; u < kUA; ++u) {
                const int64_t i = (int64_t)rt + (int64_t)u * kRowThreads;
                const int64_t ic = i < g.m ? i : g.m - 1;
                xx[u] = ap.x[ic];
                gg[u] = ap.g_new[ic];
                yy[u] = ap.y[ic];
                eq[u] = ap.eq_mask && ap.eq_mask[ic];
            }
        } else {
            const double st = agd_step_scalars(ap, lane, wg == 0, tid);
            if (lane == 0) w.red_s[16] = st;
            if (kernarg_args(g).timeline && tid == 0) kernarg_args(g).timeline[(size_t)kTimelineSlots * (size_t)wg + 4] = wall_clock64();
        }
        __syncthreads();
        a_stp = (T)w.red_s[16];
        a_bb = (T)bt;
        a_omb = (T)(float)(1.0f - bt);
        applied_rows = (int64_t)kUA * kRowThreads < g.m ? (int64_t)kUA * kRowThreads : g.m;
        if (wave != 0) {
#pragma unroll
            for (int u = 0; u < kUA; ++u) {
                const int64_t i = (int64_t)rt + (int64_t)u * kRowThreads;
                T yn, xn;
                agd_update_values(xx[u], gg[u], yy[u], eq[u], a_stp, a_bb, a_omb, yn, xn);
                if (i < g.m) {
                    if (wg == 0) {  // one workgroup stores the new iterate for the launches that follow
                        ap.y_new[i] = yn;
                        ap.x_next[i] = xn;
                    }
                    if constexpr (LAM_LDS) {
                        if (i < (g.m_hot > 0 ? g.m_lam : g.m)) w.lam_s[i] = (T)(w.s * xn);
                    }
                    const double al = fabs((double)xn);
                    lmax = al > lmax ? al : lmax;
                }
            }
        }
    }
    // Rows to pull: all of them when their maximum matters (projections that do not bound x take their |v| bound from max |lambda|) or
    // when this launch applies the optimiser step (workgroup 0 stores every row of the new iterate); otherwise only the rows that live
    // in LDS -- under the hot-rows plan the cold tail is read by the tiles that need it, and staging it here cost the MovieLens shape
    // (26 744 rows, 12 928 of them hot) a third dependent L2 round trip per launch.
    const int64_t m_pull = (g.has_unbounded || applying || g.m_hot <= 0) ? g.m : m_lam;
    {   // latency bound (every workgroup pulls the dual vector from L2): thirteen loads in flight per thread -- the benchmark's 10^4 duals,
        // and the most rows a hot-rows plan keeps in LDS (< 13 312), in ONE round trip
        constexpr int kU = 13;
        int64_t pulled = 0;
        if (!applying && m_pull > (int64_t)kU * kFusedThreads) {  // (wave-uniform) a long dual vector -- the MovieLens shape stages all its 26 744
            constexpr int kW = 28;                                   // rows -- still in ONE round trip: 28 loads in flight per thread (three before)
            T lw[kW];
#pragma unroll
            for (int u = 0; u < kW; ++u) {
                const int64_t i = (int64_t)tid + (int64_t)u * kFusedThreads;
                lw[u] = g.lambda[i < g.m ? i : g.m - 1];
            }
#pragma unroll
            for (int u = 0; u < kW; ++u) {
                const int64_t i = (int64_t)tid + (int64_t)u * kFusedThreads;
                if (i < m_pull) {
                    if constexpr (LAM_LDS) {
                        if (i < m_lam) w.lam_s[i] = (T)(w.s * lw[u]);
                    }
                    const double al = fabs((double)lw[u]);
                    lmax = al > lmax ? al : lmax;
                }
            }
            pulled = (int64_t)kW * kFusedThreads;
        }
        if (applying) pulled = applied_rows;  // (longer dual vectors: the rest, by every thread, with the step in hand)
        for (int64_t i0 = pulled + tid; i0 < m_pull; i0 += (int64_t)kU * kFusedThreads) {
            T l[kU];
            if (!applying) {
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int64_t i = i0 + (int64_t)u * kFusedThreads;
                    l[u] = g.lambda[i < g.m ? i : g.m - 1];
                }
            } else {
                const ApplyArgs<T>& ap = kernarg_args(g).apply;
                T xx[kU], gg[kU], yy[kU];
                bool eq[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int64_t i = i0 + (int64_t)u * kFusedThreads;
                    const int64_t ic = i < g.m ? i : g.m - 1;
                    xx[u] = ap.x[ic];
                    gg[u] = ap.g_new[ic];
                    yy[u] = ap.y[ic];
                    eq[u] = ap.eq_mask && ap.eq_mask[ic];
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int64_t i = i0 + (int64_t)u * kFusedThreads;
                    T yn, xn;
                    agd_update_values(xx[u], gg[u], yy[u], eq[u], a_stp, a_bb, a_omb, yn, xn);
                    l[u] = xn;
                    if (wg == 0 && i < g.m) {  // one workgroup stores the new iterate for the launches that follow
                        ap.y_new[i] = yn;
                        ap.x_next[i] = xn;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int64_t i = i0 + (int64_t)u * kFusedThreads;
                if (i < g.m) {
                    if constexpr (LAM_LDS) {
                        if (i < m_lam) w.lam_s[i] = (T)(w.s * l[u]);
                    }
                    const double al = fabs((double)l[u]);
                    lmax = al > lmax ? al : lmax;
                }
            }
        }
    }
    if (kernarg_args(g).timeline && tid == 0) kernarg_args(g).timeline[(size_t)kTimelineSlots * (size_t)wg + 5] = wall_clock64();  // (this wavefront's rows are in LDS)
    if constexpr (GRAD_LDS) {
        for (int64_t i = tid; i < m_lds; i += kFusedThreads) w.grad_s[i] = 0;
    }
    // (a WAVE-UNIFORM branch -- `wave` is a scalar -- not `if (tid < kProjLds)`: the table is filled by whole wavefronts, and no
    //  wavefront reaches the code after it with an empty exec mask; see DESIGN.md section 8 on the spill the compiler once placed
    //  ahead of the exec restore of exactly this join)
    static_assert(kProjLds % 64 == 0 && kProjLds <= kFusedThreads, "projection table filled by whole wavefronts");
    if (wave < kProjLds / 64) {  // slot kProjLds-1 stays the identity (columns in no entry)
        w.proj_s[tid] = make_proj<T>(pd_early.kind, pd_early.p0, pd_early.p1);  // (slots past the map, and slot kProjLds - 1: the identity)
    }
    lmax = wave_allreduce(lmax, OpMax());
    if (lane == 0) w.red_s[wave] = lmax;
    __syncthreads();
    lmax = w.red_s[0];
    for (int q = 1; q < kFusedWaves; ++q) lmax = w.red_s[q] > lmax ? w.red_s[q] : lmax;
    __syncthreads();  // red_s is reused by the epilogue
    // every row sum satisfies |sum a x| <= amax * xmax * row_count_max < 2^E -> scale = 2^(bits-E)
    int shift, shift2;
    {
        double xmax = g.xmax_bounded;
        if (g.has_unbounded) {
            const double vmax = fabs(-1.0 / g.gamma) * ((g.amax + 2.0 * g.fair_max) * lmax + g.cmax);  // (|lambda_K - lambda_{K+1}| <= 2 lmax)
            const double ub = vmax > g.pmax_unbounded ? vmax : g.pmax_unbounded;
            xmax = ub > xmax ? ub : xmax;
        }
        // 32-bit slabs: the grid is taken from what ONE WORKGROUP's share of a row is expected to stay below (slab_abound: a share of the largest
        // row L1 norm of A, times the bound of x), so that
        // the low word of its 64-bit LDS accumulator normally IS the value; a share that does not fit sends its high word too (epilogue)
        const bool s32 = sizeof(T) == 4 && g.slab32;
        const double bound = s32 ? xmax * g.slab_abound : g.amax * xmax * g.row_count_max;
        int e = 0;
        if (bound > 0.0 && bound < 1.7e308) (void)frexp(bound, &e);
        else if (bound > 0.0) e = 1100;  // (an overflowing bound: the coarsest grid, not exponent 0)
        shift = (s32 ? 30 : FixedBits<T>::value) - e;
        shift = shift > 1000 ? 1000 : (shift < -1000 ? -1000 : shift);
        shift2 = scalar_shift((double)g.nnz, g.cmax, xmax);
    }
    w.scale = ldexp(1.0, shift);
    w.scale2 = ldexp(1.0, shift2);
    if (wg == 0 && tid == 0) {
        g.shift_out[0] = shift;
        g.shift_out[1] = shift2;
    }
    w.gacc = GRAD_LDS ? w.grad_s : g.partial;
    // Cold rows of the hot-rows plan: a device-scope atomic executes at the memory side -- on this chip every one of them leaves its
    // XCD (counters of the MovieLens shape, round 4: 64 MB of WRITE_SIZE per launch for 1.6 M of them).  Each XCD gets its own array
    // instead (kColdCopies of them; the statistics / reduction kernels add them up) and the atomics are WORKGROUP-scope: they execute in
    // the XCD's own L2, which every workgroup that uses this array shares.  The kernel boundary writes the dirty lines back.
    w.cold = nullptr;
    if (g.m_hot > 0 && g.cold_per_xcd) {
        const unsigned int xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((32 - 1) << 11)) & (unsigned int)(kColdCopies - 1);
        w.cold = g.cold_grad + (int64_t)xcc * g.mpad;
    }
    return w;
}

// Epilogue: scalar partials of the workgroup (integers: any order gives the same sums), then its private gradient slab.
template <class T, bool GRAD_LDS, bool FAIR = false, bool REREAD = false>
__device__ __forceinline__ void fused_epilogue(const FusedArgs<T>& g, const WgCtx<T>& w, FxAcc acc, int tid, int lane, int wave, int wg, double fair = 0.0) {
    fx_finish(acc);
    const long long obj = wave_allreduce(acc.obj, OpAdd());
    const long long ssq = wave_allreduce(acc.ssq, OpAdd());
    if constexpr (FAIR) fair = wave_allreduce(fair, OpAdd());
    long long* red_i = reinterpret_cast<long long*>(w.red_s);
    uint32_t* slab_oflag = reinterpret_cast<uint32_t*>(w.red_s + 48);  // (a free word of the 512-byte scratch: [0, 48) the sums below, [56] the second binary's claim counters)
    if (tid == 0) *slab_oflag = 0u;
    if (lane == 0) {
        red_i[2 * wave] = obj;
        red_i[2 * wave + 1] = ssq;
        if constexpr (FAIR) w.red_s[2 * kFusedWaves + wave] = fair;
    }
    __syncthreads();
    if (tid == 0) {
        // (XCD balance: every wavefront of the workgroup has walked all its tiles by now)
        unsigned long long* bst = kernarg_args(g).bal_stamps;
        if (bst) bst[4 * (size_t)wg + 2] = wall_clock64();
        long long o = 0, q = 0;
        for (int k = 0; k < kFusedWaves; ++k) {
            o += red_i[2 * k];
            q += red_i[2 * k + 1];
        }
        g.partial_scal[2 * (int64_t)wg] = o;
        g.partial_scal[2 * (int64_t)wg + 1] = q;
        if constexpr (FAIR) {
            double fsum = 0.0;
            for (int k = 0; k < kFusedWaves; ++k) fsum += w.red_s[2 * kFusedWaves + k];
            g.partial_fair[wg] = fsum;
        }
    }
    if constexpr (GRAD_LDS) {
        // (the slab's address is formed from the kernel arguments RE-READ here, not from values carried in scalar registers since the
        //  kernel's first instructions: one build of the double-precision second binary carried the 64-bit row stride through an SGPR
        //  spill whose high half the compiler had meanwhile reused for gridDim.x -- the stride came back as 157 * 2^32 + 320 and the flush
        //  faulted; found with rocgdb's precise memory mode, tools/gdb_fault.sh)
        //  Only the second binary does (REREAD): the benchmark's kernel keeps the values it has in registers -- the dependent scalar load
        //  at the end of every launch measured +0.7 ... 1.5 % at 10M entities, all-box.
        // 32-bit slabs (fp32 handles; the branch does not exist in the fp64 kernels): the low words of the accumulators -- half the bytes out,
        // and half of them back in for the slab sums -- and, only when some share of this workgroup does not fit them, the high words too,
        // stamped with this launch's epoch (common.h: slab32).  Every 64-bit argument is (re-)read in the scope that uses it.
        bool flushed = false;
        if constexpr (sizeof(T) == 4) {
            if ((REREAD ? kernarg_args(g).slab32 : g.slab32) != 0) {  // (wave-uniform)
                flushed = true;
                int ovf = 0;
                {
                    const FusedArgs<T>& ga = REREAD ? kernarg_args(g) : g;
                    const int64_t m_lds = ga.m_hot > 0 ? ga.m_hot : ga.m;
                    int32_t* slab = reinterpret_cast<int32_t*>(ga.partial) + (int64_t)wg * ga.mpad;
                    // wide rows (common.h): ONE word per thread says which of ITS rows -- tid, tid + 1024, ... -- are wide (bit u: row tid + 1024 u); it is
                    // requested before the flush and used after it, so the flush of the low words runs while it travels.  (A flag byte per row read
                    // inside the loop cost a memory round trip at the end of every launch: +1.3 us; thirteen byte loads up front were no better.)
                    const uint32_t* wbits = kernarg_args(g).slab_wide_bits;  // (null: no wide rows)
                    const bool wide = wbits != nullptr;
                    const uint32_t wmask = wide ? wbits[tid] : 0u;
                    uint32_t mism = 0u;
                    {
                        int u = 0;
                        for (int64_t i = tid; i < m_lds; i += kFusedThreads, ++u) {  // (u < 32: a gradient that fits the LDS has < 13 312 rows)
                            const long long v = w.grad_s[i];
                            const int32_t lo = (int32_t)v;
                            slab[i] = lo;
                            mism |= ((long long)lo != v ? 1u : 0u) << (u & 31);
                        }
                    }
                    ovf = (mism & ~wmask) != 0u;  // (a wide row's high word travels anyway, below)
                    if (wide) {  // the wide rows' high words: a few dozen rows, every launch
                        const FusedArgs<T>& gw = kernarg_args(g);
                        int32_t* hi = gw.slab_hi + (int64_t)wg * gw.mpad;
                        for (int32_t j = tid; j < gw.n_wide; j += kFusedThreads) {
                            const int32_t i = gw.slab_wide_list[j];
                            const long long v = w.grad_s[i];
                            hi[i] = (int32_t)((v - (long long)(int32_t)v) / 4294967296ll);
                        }
                    }
                }
                // (workgroup-wide OR through the scratch word zeroed above -- __syncthreads_or brings 256 bytes of static LDS of its own, which the
                //  160 KB plan has no room for)
                if (__any(ovf) && lane == 0) *slab_oflag = 1u;
                __syncthreads();
                if (*slab_oflag) {  // rare
                    const FusedArgs<T>& gb = kernarg_args(g);
                    const int64_t m_lds = gb.m_hot > 0 ? gb.m_hot : gb.m;
                    int32_t* hi = gb.slab_hi + (int64_t)wg * gb.mpad;
                    for (int64_t i = tid; i < m_lds; i += kFusedThreads) {
                        const long long v = w.grad_s[i];
                        hi[i] = (int32_t)((v - (long long)(int32_t)v) / 4294967296ll);  // (exact: the difference is a multiple of 2^32)
                    }
                    if (tid == 0) {
                        gb.slab_ovf[wg] = gb.slab_epoch;
                        gb.slab_ovf[gridDim.x] = gb.slab_epoch;
                    }
                }
            }
        }
        if (!flushed) {
            const FusedArgs<T>& gk = REREAD ? kernarg_args(g) : g;
            const int64_t m_lds = gk.m_hot > 0 ? gk.m_hot : gk.m;
            long long* slab = gk.partial + (int64_t)wg * gk.mpad;
            for (int64_t i = tid; i < m_lds; i += kFusedThreads) slab[i] = w.grad_s[i];
        }
    }
}

template <class T>
__device__ __forceinline__ ProjT<T> lookup_proj(const FusedArgs<T>& g, const ProjT<T>* proj_s, uint32_t pid) {
    ProjT<T> p = proj_s[pid < (uint32_t)(kProjLds - 1) ? pid : (uint32_t)(kProjLds - 1)];
    if (pid >= (uint32_t)(kProjLds - 1) && pid != kNoProj && pid != 0xFFFFFFFFu) p = make_proj<T>(g.projs[pid].kind, g.projs[pid].p0, g.projs[pid].p1);
    return p;
}

}  // namespace dl
