// sell.h -- "column per lane" slices for short simplex columns (layout 4 handles; built by sell_build.hip).
//
// The 256-wide window tile keeps the caller's CSC order: four consecutive non-zeros per lane, a column spread over lanes,
// every per-column quantity of the simplex projection (max, sum and size of the support, each Newton pass) a SEGMENTED
// cross-lane reduction -- ~38 vector instructions each, however short the columns are.  At the benchmark's ten non-zeros
// per column that machinery makes the simplex tile instruction bound (322 VALU per 256 non-zeros early in a solve, 543 late,
// against 89 for a clamp tile; profiles/r01f_bench_10m_simplex_pmc.json).
//
// Here the DATA are re-laid instead (one-off, at handle creation; the handle owns the copy): the short columns of a simplex
// entry are sorted by length and cut into slices of 64; inside a slice element t of column L sits at base + 64 t + L.  Lane L
// then owns column L: every load of a wavefront is one contiguous 256-byte (values) or 128-byte (uint16 rows) run, and max,
// sum, count and every Newton pass are plain per-lane recurrences over the column's registers -- no cross-lane traffic at all,
// and a per-COLUMN scalar (threshold, division) costs one instruction for 64 columns.  Sorting by length makes the slice
// height equal to (almost) every column's length, so padding is a few slots per length class.
//
// Same arithmetic contract as simplex4.h (the reference's _duchi_proj, simplex.py:126-236): clamp at 0, keep if the sum of
// the first support is <= z + 1e-6 ("simplex" only), vertex z e_argmax when only the maximum exceeds max - z, otherwise
// x = max(u - theta, 0) with theta from monotone Newton (Michelot) passes started at max - z; thresholds never decrease.
#pragma once
#include "fused_common.h"
#include "simplex4.h"

namespace dl {

constexpr int kSellMaxH = 24;     // tallest slice (a column's clamped values stay in registers across the passes: 32 steps would
                                  // spill); longer columns stay on the window / single-column paths
constexpr int kSellDescWords = 4; // { base[31:0] ; base[39:32] | H << 8 | Hmin << 16 | (ncols - 1) << 24 ; projection id | log2 K << 8 ; dense0 }

__host__ __device__ inline int sell_chunks(int H) { return (H + 3) >> 2; }

// ---- K lanes per column (round 3): columns of 25 .. 512 non-zeros ----
// A column longer than the tallest slice is dealt to K = 2, 4, 8 or 16 ADJACENT lanes: element e of the column sits at lane
// K * (column in slice) + (e mod K), step e / K -- the slice is still base + 64 t + lane, every load of a wavefront still one
// contiguous run, and the per-lane recurrences are unchanged; each per-COLUMN quantity (maximum, sums and sizes of the supports) is
// combined over the K lanes by log2 K butterfly steps on the DPP unit (quad_perm, row_half_mirror, row_mirror -- no LDS traffic),
// after which all K lanes hold the same bits and take the same decisions.  K is the smallest power of two that brings the height
// to <= 16 steps (the tallest variant that keeps its registers): 25-32 -> 2, 33-64 -> 4, 65-128 -> 8, 129-255 -> 16, 256-512 -> 32 (the
// last step of that butterfly crosses two DPP rows: one ds_swizzle), so a slice is 8 .. 16 steps high.  Before, such columns of a sliced entry were single-column tiles (one wavefront per column, latency bound)
// and those of an unsliced entry went through the segmented window tile (instruction bound).
constexpr int kSellMaxLenLanes = 512;  // longest column a slice can hold (K = 32: two columns per slice, their lengths ride the descriptor;
                                       // up to 255 the per-column length record is one byte)
__host__ __device__ inline int sell_lanes_log(int len) { return len <= kSellMaxH ? 0 : (len <= 32 ? 1 : (len <= 64 ? 2 : (len <= 128 ? 3 : (len <= 255 ? 4 : 5)))); }

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov0_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true); }
// all-reduce over the 2^KLOG adjacent lanes of a column; commutative steps, so every lane of the group ends with identical bits
// value of lane ^ 16 (ds_swizzle, bit mode: and 0x1f, or 0, xor 0x10 -- inside each half of the wavefront; no address register)
__device__ __forceinline__ uint32_t swizzle_xor16(uint32_t x) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x401F); }
__device__ __forceinline__ float swizzle_xor16(float x) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F)); }
__device__ __forceinline__ double swizzle_xor16(double x) {
    return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(x), 0x401F), __builtin_amdgcn_ds_swizzle(__double2loint(x), 0x401F));
}
struct OpMaxNonNegT {
    template <class T>
    __device__ __forceinline__ T operator()(T a, T b) const { return max_nonneg(a, b); }
};
template <int KLOG, class T>
__device__ __forceinline__ T group_sum(T x) {
    if constexpr (KLOG == 6) return wave_allreduce_dpp(x, OpAdd());  // one column per wavefront
    if constexpr (KLOG >= 1) x = (T)(x + dpp_mov0<DPP_QUAD_XOR1, 0xf>(x));
    if constexpr (KLOG >= 2) x = (T)(x + dpp_mov0<DPP_QUAD_XOR2, 0xf>(x));
    if constexpr (KLOG >= 3) x = (T)(x + dpp_mov0<DPP_ROW_HALF_MIRROR, 0xf>(x));
    if constexpr (KLOG >= 4) x = (T)(x + dpp_mov0<DPP_ROW_MIRROR, 0xf>(x));
    if constexpr (KLOG == 5) x = (T)(x + swizzle_xor16(x));  // lane ^ 16: the partner row of a 32-lane column
    return x;
}
template <int KLOG>
__device__ __forceinline__ uint32_t group_sum_u32(uint32_t x) {
    if constexpr (KLOG == 6) return wave_allreduce_dpp(x, OpAdd());
    if constexpr (KLOG >= 1) x += dpp_mov0_u32<DPP_QUAD_XOR1>(x);
    if constexpr (KLOG >= 2) x += dpp_mov0_u32<DPP_QUAD_XOR2>(x);
    if constexpr (KLOG >= 3) x += dpp_mov0_u32<DPP_ROW_HALF_MIRROR>(x);
    if constexpr (KLOG >= 4) x += dpp_mov0_u32<DPP_ROW_MIRROR>(x);
    if constexpr (KLOG == 5) x += swizzle_xor16(x);
    return x;
}
template <int KLOG, class T>
__device__ __forceinline__ T group_max_nonneg(T x) {
    if constexpr (KLOG == 6) return wave_allreduce_dpp(x, OpMaxNonNegT());
    if constexpr (KLOG >= 1) x = max_nonneg(x, dpp_mov0<DPP_QUAD_XOR1, 0xf>(x));
    if constexpr (KLOG >= 2) x = max_nonneg(x, dpp_mov0<DPP_QUAD_XOR2, 0xf>(x));
    if constexpr (KLOG >= 3) x = max_nonneg(x, dpp_mov0<DPP_ROW_HALF_MIRROR, 0xf>(x));
    if constexpr (KLOG >= 4) x = max_nonneg(x, dpp_mov0<DPP_ROW_MIRROR, 0xf>(x));
    if constexpr (KLOG == 5) x = max_nonneg(x, swizzle_xor16(x));
    return x;
}

// (Round 4 tried a software prefetch of the wavefront's NEXT slice into the L2 -- three plain loads touching every 128-byte line of its
//  a / c / row runs, issued once the current slice's values are formed, results carried in three registers to the next slice's loads --
//  on the K-lane loop: the MovieLens shape went from 56.6 to 66.1 us per launch, same box, three alternations (and FETCH_SIZE doubled:
//  the touched lines were fetched again by the non-temporal loads).  Wired into the one-lane loop it cost the benchmark's instantiation
//  its last two registers (16 bytes of scratch: the build refuses that).  Removed; profiles/r04_ab_movielens_touch_negative.txt.)
// (Round 4 also measured a CHUNKED slice -- four consecutive steps of a column side by side, base + 256 (t / 4) + 4 lane + t % 4, tails
//  of two and one: the same 64 H slots, read by 16- and 8-byte loads per lane, nine instructions for a ten-step slice instead of
//  thirty -- on the guess that the memory pipeline prefers wide accesses.  It does not here: same box, 100M all-simplex 1.675 / 1.749 ms
//  step-major against 1.755 / 1.809 chunked, 10M unchanged (profiles/r04d_ab_chunked_slices_negative.txt).  The slice stays step-major.)
// (Also measured there, same box, and not kept -- profiles/r04g_ab_isolation_after_issue.txt: the request for the next descriptor and the LDS
//  read of the projection entry moved BEHIND the slice's own loads (they stand between a wavefront's slices, a round trip each, with
//  nothing of its own in flight) together with the primal pointer's scalar load: +1.2 ... 1.5 % on all-simplex maps; the Newton loop's
//  control carried as a wave-uniform ballot instead of a per-lane flag (five vector instructions per pass less) and the window tile's
//  four slots as hand-written v_pk pairs: no measurable difference.)
// (And packed arithmetic -- a * lambda', s * c, their sum, u - theta and a * x for two steps per v_pk_mul_f32 / v_pk_add_f32, register
//  pairs allocated without a single extra move: 4.3 % fewer vector instructions in the slice loop, NO difference in time on any shape,
//  three alternations, profiles/r04i_ab_packed_slice_arithmetic_neutral.txt.  After the scalar clean-up above the slices are no longer
//  bound by their vector instruction count.)
// (Round 5 built the second slice in flight that DESIGN.md section 8 item 00 asked for -- for runs of full slices of one height 5 .. 9 the 3 H loads of
//  slice k + 1 issued into a second register set BEFORE slice k is projected, descriptors two slices ahead, the loop body written twice; the device
//  assembly waits with vmcnt(27) where it waited with vmcnt(0), no scratch, 126 VGPRs -- and measured it against the same binary with the runs
//  switched off, same box, three alternations: 100M all-simplex 1.710 ms against 1.678 (+1.9 %), 100M mixed 1.577 against 1.560 (+1.1 %), config 3
//  (10M simplex, gamma decay) 0.199 against 0.184 (+8 %: a wavefront's runs are three slices long there and each ends in one redundant slice load).
//  More bytes in flight per wavefront do not raise the rate the slices stream at; removed.  profiles/r05l_ab_second_slice_in_flight_same_box.txt,
//  the patch: profiles/r05l_second_slice_in_flight.patch.)
// One slice.  HM = 4 * chunks >= H.  RELOAD: the value / row registers are not kept across the Newton passes; the slice is
// read a second time (L2 / HBM) for the scatter -- tall slices, whose columns would not fit the register file otherwise.
// KLOG: log2 of the lanes per column; `len` is the COLUMN's length, `len_lane` the number of its elements this lane holds
// (KLOG = 0: the same), `Hmin` the least len_lane of the slice.
// KLOG = 6: ONE column per wavefront, read IN PLACE from the caller's arrays -- element e of a column at lane e mod 64, step e / 64 is
// the CSC order itself, so `base` is the column's offset in g.a / g.c / g.rowidx and nothing is copied: the single-column tiles of up
// to 64 HM non-zeros walked as one slice (all loads in flight at once, values kept in registers, straight-line passes) instead of by
// process_long_tile's batched loops.
template <class T, class RowT, int HM, bool RELOAD, bool LAM_LDS, bool HOT, bool FAIR, int KLOG = 0, bool EXACT = false>
__device__ __forceinline__ void sell_slice(const FusedArgs<T>& g, const WgCtx<T>& w, const ProjT<T>& pj, uint64_t base, int H, int Hmin, int len, int len_lane,
                                           uint64_t dense, bool has_col, int lane, T sd, const int32_t* eq_row, FxAcc& acc, double& fair) {
    const T s = w.s;
    // wave-uniform bases (scalar registers) + one 32-bit lane offset per element width: step t is an immediate
    constexpr bool ORIG = KLOG == 6;
    // EXACT: the variant's step count IS the slice's height (the fp32 variants 5 .. 16), so every "does this step exist" below is
    // decided at compile time -- before, the load issue of a slice spent two scalar branches and a zero fill per step on them
    const int Hc = EXACT ? HM : H;
    const T* __restrict__ pa = byte_offset((ORIG ? g.a : g.sell_a) + base, (uint32_t)lane * (uint32_t)sizeof(T));
    const T* __restrict__ pc = byte_offset((ORIG ? g.c : g.sell_c) + base, (uint32_t)lane * (uint32_t)sizeof(T));
    const RowT* __restrict__ pr = byte_offset(reinterpret_cast<const RowT*>(ORIG ? g.rowidx : g.sell_r) + base, (uint32_t)lane * (uint32_t)sizeof(RowT));
    const T* __restrict__ pf = FAIR ? byte_offset((ORIG ? g.fair : g.sell_f) + base, (uint32_t)lane * (uint32_t)sizeof(T)) : nullptr;
    // ORIG: the column's last, partly filled step must not read past the column (it may end the arrays): its lanes beyond the end
    // re-read the column's last element, and their values are dropped below
    const int lane_t = (ORIG && (len & 63)) ? (lane < (len & 63) ? lane : (len & 63) - 1) : lane;
    const T* __restrict__ pa_t = ORIG ? byte_offset(g.a + base, (uint32_t)lane_t * (uint32_t)sizeof(T)) : pa;
    const T* __restrict__ pc_t = ORIG ? byte_offset(g.c + base, (uint32_t)lane_t * (uint32_t)sizeof(T)) : pc;
    const RowT* __restrict__ pr_t = ORIG ? byte_offset(reinterpret_cast<const RowT*>(g.rowidx) + base, (uint32_t)lane_t * (uint32_t)sizeof(RowT)) : pr;
    const T* __restrict__ pf_t = (ORIG && FAIR) ? byte_offset(g.fair + base, (uint32_t)lane_t * (uint32_t)sizeof(T)) : pf;
    auto PA = [&](int t) { return ((ORIG && t >= Hmin) ? pa_t : pa) + 64 * t; };
    auto PC = [&](int t) { return ((ORIG && t >= Hmin) ? pc_t : pc) + 64 * t; };
    auto PR = [&](int t) { return ((ORIG && t >= Hmin) ? pr_t : pr) + 64 * t; };
    auto PF = [&](int t) { return ((ORIG && t >= Hmin) ? pf_t : pf) + 64 * t; };
    constexpr int KEEP = RELOAD ? 1 : HM;
    constexpr int CH = 4;  // steps per batch of the RELOAD variants (loads of a batch are in flight together)
    T a[KEEP], c[KEEP], f[FAIR ? KEEP : 1], u[HM];
    uint32_t r[KEEP];
    // (developer-only timing ablations, K-lane / in-place slices of the second binary only -- results are wrong on purpose:
    //  DUALIP_HIP_ABLATE bit 14 = no cold-row scatter, 15 = no scatter at all, 16 = no Newton passes, 17 = no cold-row gather)
    constexpr bool DEVAB = DL_ABLATE(1, 1) && KLOG > 0;  // (false in the shipped library: the branches below are compiled out)
    const int ab = DEVAB ? kernarg_args(g).ablate : 0;
    // Hot-rows plan: a row >= m_hot has its dual entry in global memory (L2).  Round 3 read it under a per-element branch
    // (`row < m_hot ? lam_s[row] : s * lambda[row]`): every step of a slice then waits for its own L2 round trip at the branch's join
    // (the compiler's s_waitcnt insertion falls back to vmcnt(0) there), sixteen dependent round trips per slice -- 11 us per slice on
    // the MovieLens shape, four slices per wavefront.  Now ALL steps' global loads are issued together, unconditionally (a hot lane
    // reads lambda[0]: one line, broadcast), straight after the row indices arrive, and the choice is a select: two round trips per slice.
    // And when the whole dual vector fits the LDS beside a smaller gradient (g.m_lam == g.m, api.hip) no row is cold for the gather.
    const uint32_t m_lam32 = HOT ? (uint32_t)g.m_lam : 0u;
    const bool lam_all = HOT && m_lam32 >= (uint32_t)g.m;  // (wave-uniform)
    auto lam_cold_load = [&](uint32_t row) -> T {  // the value a cold row needs, requested unconditionally
        if (DEVAB && (ab & (1 << 17))) return (T)0;
        return g.lambda[row >= m_lam32 ? row : 0u];
    };
    auto lam_pick = [&](uint32_t row, T cold_val) -> T {
        const bool cold = row >= m_lam32;
        const T hot_val = w.lam_s[cold ? 0u : row];
        return cold ? (T)(s * cold_val) : hot_val;
    };
    auto lam_of = [&](uint32_t row) -> T {
        if constexpr (HOT) return row < m_lam32 ? w.lam_s[row] : (T)(s * g.lambda[row]);
        else return LAM_LDS ? w.lam_s[row] : (T)(s * g.lambda[row]);
    };
    const T NEG = (T)(-INFINITY);
    // ---- pass 1: v = a (-lambda/gamma)[row] + (-c/gamma), u = max(v, 0); steps past the slice's height are never issued ----
    if constexpr (!RELOAD) {
#pragma unroll
        for (int t = 0; t < HM; ++t) {
            if (t < Hc) {  // wave-uniform (EXACT: known at compile time)
                a[t] = __builtin_nontemporal_load(PA(t));
                c[t] = __builtin_nontemporal_load(PC(t));
                r[t] = (uint32_t)__builtin_nontemporal_load(PR(t));
                if constexpr (FAIR) f[t] = __builtin_nontemporal_load(PF(t));
            } else {
                a[t] = (T)0;
                c[t] = (T)0;
                r[t] = 0u;
                if constexpr (FAIR) f[t] = (T)0;
            }
        }
        const bool gather_cold = HOT && !lam_all;  // (wave-uniform: some rows' dual entries are in L2, not in LDS)
        if (gather_cold) {
#pragma unroll
            for (int t = 0; t < HM; ++t) u[t] = lam_cold_load(r[t]);  // (all in flight together; u[t] is free until the next loop writes it)
        }
#pragma unroll
        for (int t = 0; t < HM; ++t) {
            T lam;
            if constexpr (HOT) lam = gather_cold ? lam_pick(r[t], u[t]) : w.lam_s[r[t]];
            else lam = lam_of(r[t]);
            T v = (T)((T)(a[t] * lam) + (T)(s * c[t]));
            if constexpr (FAIR) v = (T)(v + (T)(sd * f[t]));
            u[t] = relu_finite(v);
            if constexpr (ORIG) u[t] = (t < Hmin || t < len_lane) ? u[t] : (T)0;
        }
    } else {
#pragma unroll
        for (int t0 = 0; t0 < HM; t0 += CH) {
            T a8[CH], c8[CH], f8[CH];
            uint32_t r8[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int t = t0 + q;
                if (t < Hc) {
                    a8[q] = *PA(t);  // (cached loads: the second pass re-reads them)
                    c8[q] = *PC(t);
                    r8[q] = (uint32_t)*PR(t);
                    if constexpr (FAIR) f8[q] = *PF(t);
                } else {
                    a8[q] = (T)0;
                    c8[q] = (T)0;
                    r8[q] = 0u;
                    f8[q] = (T)0;
                }
            }
            const bool gather_cold = HOT && !lam_all;
            if (gather_cold) {
#pragma unroll
                for (int q = 0; q < CH; ++q) u[t0 + q] = lam_cold_load(r8[q]);
            }
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                T lam;
                if constexpr (HOT) lam = gather_cold ? lam_pick(r8[q], u[t0 + q]) : w.lam_s[r8[q]];
                else lam = lam_of(r8[q]);
                T v = (T)((T)(a8[q] * lam) + (T)(s * c8[q]));
                if constexpr (FAIR) v = (T)(v + (T)(sd * f8[q]));
                u[t0 + q] = relu_finite(v);
                if constexpr (ORIG) u[t0 + q] = (t0 + q < Hmin || t0 + q < len_lane) ? u[t0 + q] : (T)0;
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the chunks apart: hoisting every chunk's loads would cost the registers this variant exists to save
        }
    }
    // ---- projection: per-lane recurrences ----
    T mx = (T)0, sall = (T)0;  // (padding slots hold a = c = 0, so u = 0 there: harmless for the maximum and the sum)
#pragma unroll
    for (int t = 0; t < HM; ++t) {
        mx = max_nonneg(mx, u[t]);
        sall = (T)(sall + u[t]);
    }
    mx = group_max_nonneg<KLOG>(mx);
    sall = group_sum<KLOG>(sall);
    // slots past a column's own length (the slice's padding; none below Hmin) must not count as members
    if constexpr (KLOG == 0) asm volatile("" : "+v"(mx), "+v"(sall));  // (the two reductions stay AHEAD of the branch: sunk below it they keep the unmasked values alive, one copy per step)
    if (Hmin < HM) {  // (wave-uniform; all but the slices at a length-class boundary hold columns of one length: nothing to mask)
#pragma unroll
        for (int t = 0; t < HM; ++t)
            if (t >= Hmin) u[t] = t < len_lane ? u[t] : NEG;
    }
    // first support {u > theta_0}, theta_0 = the larger of two lower bounds of the threshold: max - z (the reference's top-2
    // shortcut: only the maximum above it <=> vertex) and (sum of all - z) / length (Michelot's start).  Late in a solve, when
    // most of a column is in its support, the second one is close to the answer and saves a pass or two; a single member can
    // only happen with theta_0 = max - z (the threshold of a one-element support IS max - z, and theta_0 never exceeds the
    // threshold), so the vertex test is unchanged.
    const T th0 = (T)(mx - pj.z);
    T theta0 = th0;
    if (!DL_ABLATE(g.ablate, 32)) theta0 = tmax(th0, div_exactish((T)(sall - pj.z), (T)(len > 0 ? len : 1)));
    // (supports are counted in integers: the compiler folds two members' worth of compare masks into one add-with-carry, so a
    //  step of a pass costs four vector instructions -- compare, select, add, count -- instead of five with a float counter)
    typedef uint32_t CntT;
    T sum = (T)0;
    CntT cnt = (CntT)0;
#pragma unroll
    for (int t = 0; t < HM; ++t) {
        const bool in = u[t] > theta0;
        sum = (T)(sum + (in ? u[t] : (T)0));
        cnt += in ? (CntT)1 : (CntT)0;
    }
    sum = group_sum<KLOG>(sum);
    cnt = group_sum_u32<KLOG>(cnt);
    // column state: theta (0 = keep the clamped values), vertex flag
    const bool ineq = pj.kind == DL_PROJ_SIMPLEX;
    const bool keep = ineq && !(sall > pj.ztol);  // feasible after the clamp (simplex.py:153-158: the sum of the whole column)
    bool vertex = !keep && cnt == (CntT)1;
    T theta = (T)0;
    bool act = false;
    if (!keep) {
        theta = div_exactish((T)(sum - pj.z), (T)(cnt > (CntT)0 ? cnt : (CntT)1));
        act = cnt > (CntT)2;  // a support of two is final: the runner-up stays above (sum - z)/2 exactly when it is above max - z
    }
    if (eq_row) {  // simplex_eq reference-compatibility mode (cold): a deficit is spread over the padded block height
        if (pj.kind == DL_PROJ_SIMPLEX_EQ && sall < pj.z) {
            const T L = (T)eq_row[eq_bucket(len > 0 ? len : 1)];
            theta = (T)((T)(sall - pj.z) / L);
            vertex = L == (T)1;
            act = false;
        }
    }
    theta = tmax(theta, keep ? (T)0 : theta0);  // (sum - z)/cnt >= theta_0 in exact arithmetic: keep it so under rounding (nested supports)
    theta = vertex ? theta0 : theta;  // (the threshold the single member was counted at)
    CntT cprev = cnt;
    if (DEVAB && (ab & (1 << 16))) act = false;
    for (int it = 0; it < (kSellMaxH << KLOG) + 2 && __any(act); ++it) {
        T s2 = (T)0;
        CntT c2 = (CntT)0;
#pragma unroll
        for (int t = 0; t < HM; ++t) {
            const bool in = u[t] > theta;
            s2 = (T)(s2 + (in ? u[t] : (T)0));
            c2 += in ? (CntT)1 : (CntT)0;
        }
        s2 = group_sum<KLOG>(s2);
        c2 = group_sum_u32<KLOG>(c2);
        const bool changed = act && c2 != cprev && c2 > (CntT)0;
        const T tn = div_exactish((T)(s2 - pj.z), (T)(c2 > (CntT)0 ? c2 : (CntT)1));
        theta = changed ? tmax(theta, tn) : theta;  // thresholds never decrease: nested supports, guaranteed termination
        cprev = changed ? c2 : cprev;
        act = changed;
    }
    // ---- x, scatter, objective sums ----
    T o32 = (T)0, q32 = (T)0, f32 = (T)0;
    auto finish = [&](int t, T at, T ct, uint32_t rt, T ft) {
        const T xg = relu((T)(u[t] - theta));
        const T x = (vertex && u[t] > theta) ? pj.z : xg;  // vertex: exact z at the maximum, as the reference (xg is 0 at its other members)
        const T ax = (T)(at * x);
        if (ax != (T)0 && !(DEVAB && (ab & (1 << 15)))) {
            if constexpr (HOT) {
                if ((int64_t)rt < g.m_hot) scatter_fixed_lds(w.gacc, rt, ax, w.scale);
                else if (!(DEVAB && (ab & (1 << 14)))) scatter_fixed_cold(w.cold, g.cold_grad, rt, ax, w.scale);
            } else {
                scatter_fixed(w.gacc, rt, ax, w.scale);
            }
        }
        o32 = fma_exact(ct, x, o32);
        q32 = fma_exact(x, x, q32);
        if constexpr (FAIR) f32 = fma_exact(ft, x, f32);
        return x;
    };
    const FusedArgs<T>& gk = kernarg_args(g);
    T* xo = gk.x_out;
    uint64_t k0 = 0;
    if (xo && has_col) k0 = (ORIG ? dense : gk.sell_colstart[dense]) + (uint64_t)(lane & ((1 << KLOG) - 1));  // (ORIG: `dense` IS the column's place in the caller's order)  // this lane's first element of the column
    if constexpr (!RELOAD) {
        // The primal, when requested (the last launch of a solve), is written by a loop of its OWN ahead of the scatter: one wave-uniform
        // branch per slice instead of a mask, a test and a branch per STEP inside the scatter loop, whose steps the scheduler can now
        // overlap.  (Round 2 had split the scatter loop itself on `xo` -- two copies of it -- for 9 more registers, 12 bytes of scratch
        // and +6 % kernel time; this costs none: the values are recomputed from u and theta, same expression, same bits.)
        if (xo) {
#pragma unroll
            for (int t = 0; t < HM; ++t) {
                const T xg = relu((T)(u[t] - theta));
                const T x = (vertex && u[t] > theta) ? pj.z : xg;
                if (has_col && t < len_lane) xo[k0 + ((uint64_t)t << KLOG)] = x;
            }
        }
#pragma unroll
        for (int t = 0; t < HM; ++t) (void)finish(t, a[t], c[t], r[t], FAIR ? f[t] : (T)0);
    } else {
#pragma unroll
        for (int t0 = 0; t0 < HM; t0 += CH) {
            T a8[CH], c8[CH], f8[CH];
            uint32_t r8[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int t = t0 + q;
                if (t < Hc) {
                    a8[q] = __builtin_nontemporal_load(PA(t));
                    c8[q] = __builtin_nontemporal_load(PC(t));
                    r8[q] = (uint32_t)__builtin_nontemporal_load(PR(t));
                    if constexpr (FAIR) f8[q] = __builtin_nontemporal_load(PF(t));
                } else {
                    a8[q] = (T)0;
                    c8[q] = (T)0;
                    r8[q] = 0u;
                    f8[q] = (T)0;
                }
            }
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int t = t0 + q;
                const T x = finish(t, a8[q], c8[q], r8[q], FAIR ? f8[q] : (T)0);
                if (xo && has_col && t < len_lane) xo[k0 + ((uint64_t)t << KLOG)] = x;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    fx_add(acc, o32, q32, w.scale2);  // (one rounded integer per lane, sum and slice: the totals do not depend on who walked what)
    if constexpr (FAIR) fair += (double)f32;
}

// The slices of one wavefront: descriptors q0, q0 + S, ... (plain cyclic deal: the XCD-weighted deal of the window tiles, carried
// into this loop, measured +2.5 ... 4.6 % on all-simplex maps whatever its table -- more scalar state across fourteen variants --
// and the slices' own imbalance is small); the next descriptor travels while the current slice is processed.
template <class T, class RowT, bool LAM_LDS, bool HOT, bool FAIR>
__device__ __forceinline__ void sell_loop(const FusedArgs<T>& g, const WgCtx<T>& w, uint32_t q0, uint32_t S, uint32_t n_sell, int lane, T sd, FxAcc& acc, double& fair) {
    // (n_sell: the end of the range this deal covers -- the table's length, or the end of the first phase of a two-phase deal)
    if (q0 >= n_sell) return;
    const uint32_t dlane = (uint32_t)lane < (uint32_t)kSellDescWords ? (uint32_t)lane : (uint32_t)kSellDescWords - 1u;
    auto load_desc = [&](uint32_t q) -> uint32_t {
        const uint32_t t = q < n_sell ? q : n_sell - 1u;
        return byte_offset(g.sell_desc + (size_t)t * kSellDescWords, dlane * 4u)[0];
    };
    auto rl = [&](uint32_t dv, int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane(dv, i); };
    uint32_t dv = load_desc(q0);
    for (uint32_t q = q0; q < n_sell; q += S) {
        const uint32_t w0 = rl(dv, 0), w1 = rl(dv, 1), pid = rl(dv, 2), dense0 = rl(dv, 3);
        dv = load_desc(q + S);
        const uint64_t base = ((uint64_t)(w1 & 0xFFu) << 32) | w0;
        const int H = (int)((w1 >> 8) & 0xFFu), Hmin = (int)((w1 >> 16) & 0xFFu), ncols = (int)((w1 >> 24) & 0xFFu) + 1;
        const bool has_col = lane < ncols;
        const uint64_t dense = (uint64_t)dense0 + (uint32_t)(has_col ? lane : 0);
        // columns are sorted by length: all but the slices at a length-class boundary hold columns of ONE length -- no length bytes
        // are read for those (1 byte per column = 1 % of the slices' traffic, and a dependent load off the slice's critical path)
        int len = has_col ? H : 0;
        if (Hmin != H) len = has_col ? (int)g.sell_len[dense] : 0;  // wave-uniform
        const ProjT<T> pj = w.proj_s[pid < (uint32_t)(kProjLds - 1) ? pid : (uint32_t)(kProjLds - 1)];
        const int32_t* eq_row = nullptr;
        if (pj.kind == DL_PROJ_SIMPLEX_EQ) {  // cold: the pointer is re-read from the kernel arguments
            const int32_t* eqh = kernarg_args(g).eq_heights;
            eq_row = eqh ? eqh + (size_t)pid * kEqBuckets : nullptr;
        }
        const int hmin = ncols < 64 ? 0 : Hmin;  // a partly filled slice has empty lanes: every step needs the mask
        // registers a step keeps across the passes when nothing is re-read: a, c, [f], row, u.  Variants whose columns would
        // need more than 64 of them re-read the slice for the scatter instead (RELOAD)
        constexpr int kPer = (2 + (FAIR ? 1 : 0)) * (int)(sizeof(T) / 4) + 1 + (int)(sizeof(T) / 4);
        constexpr bool R4 = 4 * kPer > 64, R8 = 8 * kPer > 64, R12 = 12 * kPer > 64, R16 = 16 * kPer > 64;
#define DL_SELL_CASE(HM_, R_) sell_slice<T, RowT, HM_, R_, LAM_LDS, HOT, FAIR, 0, false>(g, w, pj, base, H, hmin, len, len, dense, has_col, lane, sd, eq_row, acc, fair); break
#define DL_SELL_EXACT(HM_, R_) sell_slice<T, RowT, HM_, R_, LAM_LDS, HOT, FAIR, 0, kExact>(g, w, pj, base, H, hmin, len, len, dense, has_col, lane, sd, eq_row, acc, fair); break
        // The fp32 kernels without the fairness stream (the benchmark's) have one variant per height from 5 to 16: a step past the
        // slice's height costs every pass its full instruction count (a slice of 9 in the 12-step variant: +33 %), and at ten
        // non-zeros per column that padding was ~13 % of the slices' vector instructions.  The others step by four.
        constexpr bool kExact = sizeof(T) == 4 && !FAIR;
        const int hv = (kExact && H > 4 && H <= 16) ? H : 4 * sell_chunks(H);  // (kExact: cases 5 .. 16 are entered with H == the case)
        switch (hv) {
            case 4: DL_SELL_CASE(4, R4);
            case 5: DL_SELL_EXACT(kExact ? 5 : 8, R8);
            case 6: DL_SELL_EXACT(kExact ? 6 : 8, R8);
            case 7: DL_SELL_EXACT(kExact ? 7 : 8, R8);
            case 8: DL_SELL_EXACT(8, R8);
            case 9: DL_SELL_EXACT(kExact ? 9 : 12, R12);
            case 10: DL_SELL_EXACT(kExact ? 10 : 12, R12);
            case 11: DL_SELL_EXACT(kExact ? 11 : 12, R12);
            case 12: DL_SELL_EXACT(12, R12);
            case 13: DL_SELL_EXACT(kExact ? 13 : 16, R16);
            case 14: DL_SELL_EXACT(kExact ? 14 : 16, R16);
            case 15: DL_SELL_EXACT(kExact ? 15 : 16, R16);
            case 16: DL_SELL_EXACT(16, R16);
            default: DL_SELL_CASE(24, true);
        }
#undef DL_SELL_CASE
#undef DL_SELL_EXACT
    }
}

// The slices with K = 2 .. 32 lanes per column (descriptors g.sell_lane_desc[0 .. g.n_sell_lanes): the columns of 25 .. 512 non-zeros),
// walked in their OWN loop ahead of the other phases.  Two reasons: they are the most expensive slices (9 .. 16 steps, more Newton
// passes), so they must not end a launch; and with their eight variants inside sell_loop the code of the one-lane variants changed
// enough to cost the benchmark's shapes 6 % at 10M entities (same box, DUALIP_HIP_SELL_LANES=0 on the same binary no faster: the code,
// not the slices).  Heights are 9 .. 16 by construction (sell_lanes_log): two variants per K.
template <class T, class RowT, bool LAM_LDS, bool HOT, bool FAIR>
__device__ __forceinline__ void sell_lanes_loop(const FusedArgs<T>& g, const WgCtx<T>& w, uint32_t wg, uint32_t G, uint32_t* ctr, int wave, int lane, T sd, FxAcc& acc, double& fair) {
    // workgroup wg owns the table entries sell_lane_begin[wg] .. sell_lane_begin[wg + 1] (dealt by cost on the host, sell_build.hip, in
    // descending cost); its wavefronts claim them one at a time from `ctr` (LDS)
    if (kernarg_args(g).n_sell_lanes == 0u) return;
    const uint32_t* __restrict__ table = kernarg_args(g).sell_lane_desc;
    const uint32_t first = kernarg_args(g).sell_lane_begin[wg], n_sell = kernarg_args(g).sell_lane_begin[wg + 1u];
    if (first >= n_sell) return;
    const uint32_t dlane = (uint32_t)lane < (uint32_t)kSellDescWords ? (uint32_t)lane : (uint32_t)kSellDescWords - 1u;
    auto slot = [&](uint32_t j) -> uint64_t { return (uint64_t)first + j; };
    auto load_desc = [&](uint32_t j) -> uint32_t {
        const uint64_t q = slot(j);
        return byte_offset(table + (size_t)(q < n_sell ? q : n_sell - 1u) * kSellDescWords, dlane * 4u)[0];
    };
    uint32_t static_next = (uint32_t)wave;  // (fairness stream: static deal, see fused4_kernel.h)
    auto claim = [&]() -> uint32_t {
        if constexpr (FAIR) {
            const uint32_t v = static_next;
            static_next += (uint32_t)kFusedWaves;
            return v;
        } else {
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(ctr, 1u);
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        }
    };
    auto rl = [&](uint32_t dv, int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane(dv, i); };
    uint32_t j = claim();
    uint32_t dv = load_desc(j);
    while (slot(j) < n_sell) {
        const uint32_t jn = claim();
        const uint32_t w0 = rl(dv, 0), w1 = rl(dv, 1), w2 = rl(dv, 2), dense0 = rl(dv, 3);
        dv = load_desc(jn);
        j = jn;
        const uint32_t pid = w2 & 0xFFu;
        const int klog = (int)((w2 >> 8) & 7u);  // lanes per column = 1 << klog (wave-uniform)
        const uint64_t base = ((uint64_t)(w1 & 0xFFu) << 32) | w0;
        const int H = (int)((w1 >> 8) & 0xFFu), Hmin = (int)((w1 >> 16) & 0xFFu), ncols = (int)((w1 >> 24) & 0xFFu) + 1;
        const bool has_col = (lane >> klog) < ncols;
        const uint64_t dense = (uint64_t)dense0 + (uint32_t)(has_col ? lane >> klog : 0);
        int len = has_col ? H << klog : 0;  // (one length in the slice, a multiple of K: no length bytes read)
        if (klog == 5) len = has_col ? (int)((w2 >> (11 + 9 * (lane >> 5))) & 511u) + 1 : 0;  // K = 32: two columns, their lengths - 1 in bits 11 .. 28 of the descriptor
        else if (Hmin != H) len = has_col ? (int)g.sell_len[dense] : 0;
        const int sub = lane & ((1 << klog) - 1);
        const int len_lane = has_col ? (len - sub + (1 << klog) - 1) >> klog : 0;
        const ProjT<T> pj = w.proj_s[pid < (uint32_t)(kProjLds - 1) ? pid : (uint32_t)(kProjLds - 1)];
        const int32_t* eq_row = nullptr;
        if (pj.kind == DL_PROJ_SIMPLEX_EQ) {
            const int32_t* eqh = kernarg_args(g).eq_heights;
            eq_row = eqh ? eqh + (size_t)pid * kEqBuckets : nullptr;
        }
        const int hmin = ncols < (64 >> klog) ? 0 : Hmin;
        constexpr int kPer = (2 + (FAIR ? 1 : 0)) * (int)(sizeof(T) / 4) + 1 + (int)(sizeof(T) / 4);
        constexpr bool R12 = 12 * kPer > 64, R16 = 16 * kPer > 64;
        // fp32 without the fairness stream: one variant per height 9 .. 16 (a step past the slice's height costs every pass its full
        // instruction count; at 40 non-zeros per column the mean height is 10 and the 12-step variant wasted a fifth); else 12 / 16
        constexpr bool kExact = sizeof(T) == 4 && !FAIR;
        const int hv = kExact ? (H < 9 ? 9 : (H > 16 ? 16 : H)) : (H <= 12 ? 12 : 16);
#define DL_SELL_LANES_H(K_, HM_) sell_slice<T, RowT, HM_, (HM_ * kPer > 64), LAM_LDS, HOT, FAIR, K_>(g, w, pj, base, H, hmin, len, len_lane, dense, has_col, lane, sd, eq_row, acc, fair); break
#define DL_SELL_LANES(K_) \
    switch (hv) { \
        case 9: DL_SELL_LANES_H(K_, (kExact ? 9 : 12)); \
        case 10: DL_SELL_LANES_H(K_, (kExact ? 10 : 12)); \
        case 11: DL_SELL_LANES_H(K_, (kExact ? 11 : 12)); \
        case 12: DL_SELL_LANES_H(K_, 12); \
        case 13: DL_SELL_LANES_H(K_, (kExact ? 13 : 16)); \
        case 14: DL_SELL_LANES_H(K_, (kExact ? 14 : 16)); \
        case 15: DL_SELL_LANES_H(K_, (kExact ? 15 : 16)); \
        default: DL_SELL_LANES_H(K_, 16); \
    } \
    break
        switch (klog) {
            case 1: DL_SELL_LANES(1);
            case 2: DL_SELL_LANES(2);
            case 3: DL_SELL_LANES(3);
            case 4: DL_SELL_LANES(4);
            default: DL_SELL_LANES(5);
        }
#undef DL_SELL_LANES_H
#undef DL_SELL_LANES
    }
}

}  // namespace dl
