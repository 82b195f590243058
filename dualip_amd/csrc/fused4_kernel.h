// fused4_kernel.h -- the fused pass with 256-wide tiles: FOUR consecutive non-zeros per lane.
//
// Included by four translation units: matching_kernels4[_f64].hip (DL_FUSED4_LANES 0) and matching_kernels4_lanes[_f64].hip (DL_FUSED4_LANES 1).
// The second carries, in addition, the loop over the slices with K = 2 .. 32 lanes per column (sell.h: sell_lanes_loop; columns of 25 .. 512 non-zeros) and is
// launched for the handles that have such slices.  Two binaries of one source, because the eight extra slice variants inside the
// kernel cost the handles WITHOUT such slices 2 % per launch (10M entities, all-box map included: same box, HEAD 0.1670 ms,
// with the loop compiled in 0.1713, with the call compiled out 0.1664 -- code placement, not executed work), and the benchmark's
// shapes have next to none.
//
// Same computation and LDS plan as matching_kernels.hip; what changes is the tile:
//   * a tile is a 16-byte-aligned window of 256 non-zeros [W, W+256) holding whole consecutive columns in
//     [W+lo, W+hi) -- or, for point-wise projection entries (box, cone, identity), simply the next <= 256 non-zeros of the
//     entry's run of columns, cut wherever they fall (no window then re-reads the tail of its predecessor: -4 % HBM
//     traffic and time on an all-box map); lane L owns elements 4L..4L+3, fetched with ONE 16-byte load per array (a, c) and one 8-byte
//     load of four uint16 row indices -- a quarter of the load instructions and descriptor traffic per non-zero;
//   * per-tile fixed costs (descriptor unpack, loop control) amortise over 4x the work, and the four slots of a lane
//     are independent dependency chains for the element-wise part;
//   * segment structure = four wave-uniform 64-bit head masks; all predicates are scalar mask arithmetic (simplex4.h).
// Descriptor: 12 dwords { W[39:0] | hi<<40 | lo<<49 | long<<51 ; H0 ; H1 ; H2 ; H3 ; proj id ; 0 } (long: length in H0); when every
// window of a handle is point-wise the window table is COMPACT: 2 dwords { W[39:0] | hi<<40 | lo<<49 | proj id<<52 } (0xFFF: none).
// Columns that cannot sit in a window (longer than 253, touching the array's last partial quad, or using a projection
// entry beyond the LDS table) are single-column "long" tiles: their descriptors follow the window tiles (after one all-zero
// descriptor) and are walked by process_long_tile in separate loops ahead of the hot one: by one wavefront each, or -- the
// very long ones, listed last -- by a whole workgroup.
#pragma once
#include <atomic>

#include "fused_common.h"
#include "sell.h"
#include "simplex4.h"

namespace dl {

constexpr int kDesc4Words = 12;

template <class T>
struct alignas(16) Quad {
    T v[4];
};
template <class RowT>
struct alignas(sizeof(RowT) * 4) RowQuad {
    RowT v[4];
};

template <class T>
__device__ __forceinline__ void stamp(const FusedArgs<T>& g, int wg, int tid, int k) {
    unsigned long long* tl = kernarg_args(g).timeline;
    if (tl && tid == 0) {
        unsigned long long t = wall_clock64();
        if (k == 0) {  // (developer aid: the XCD this workgroup really runs on rides in the top four bits of its first stamp)
            const unsigned int xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((32 - 1) << 11));
            t = (t & 0x0FFFFFFFFFFFFFFFull) | ((unsigned long long)(xcc & 0xFu) << 60);
        }
        tl[(size_t)kTimelineSlots * (size_t)wg + k] = t;
    }
}

// HOT: the hot-rows plan (common.h) -- rows are renumbered by frequency, rows < g.m_hot gather from / scatter to LDS, the
// cold tail reads the (renumbered) dual vector through L2 and adds to g.cold_grad with 64-bit global atomics.
// FAIR: the fairness pair of dl_matching_set_fairness -- one more streamed value f_k per non-zero, entering v_k with the
// difference of the last two duals; its sum f.x goes to g.partial_fair (rows m-2 / m-1 of the gradient are +- that sum).
template <class T, class RowT, bool LAM_LDS, bool GRAD_LDS, bool HOT, bool FAIR = false, bool LANES = false>
__global__ __launch_bounds__(kFusedThreads) void matching_fused_kernel4(FusedArgs<T> g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    stamp(g, wg, tid, 0);
    const LaneConst lc = make_lane_const(lane);
    FxAcc acc;  // c.x and sum x^2 of this lane in fixed point (fused_common.h)
    double fair = 0.0;

    // ---- tile schedule ----
    // Descriptors are stored in SCHEDULE order (api.hip: schedule_tiles4) and dealt cyclically to the S = 16 * workgroups
    // wavefronts of the launch: wavefront W takes slots W, W + S, W + 2S, ...  Every wavefront therefore sees the same
    // proportion of every projection block (no cost model, no tail), the launch sweeps the arrays front to back, and
    // the host interleaves instruction-bound (simplex) and memory-bound (point-wise) tiles so that the wavefronts
    // sharing a SIMD are in different kinds at any moment.
    const uint32_t n_tiles = g.n_tiles;
    const uint32_t S = (uint32_t)gridDim.x * (uint32_t)kFusedWaves;
    // last quad of the arrays that may be read with a vector load (the final partial quad is never part of a window)
    const uint64_t last_quad = ((uint64_t)g.nnz >> 2) - 1;  // host guarantees nnz >= 1024 for this layout

    // descriptor word `lane` of schedule slot q (lanes >= 12 re-read word 11; slots past the end read the all-zero
    // descriptor the host appends).  A plain load: nothing consumes it before the next iteration.
    // Compact table (every window point-wise -- the device packer's case unless a simplex entry is not sliced): 2 dwords per
    // window, the projection id in the top 12 bits of the second; the head masks do not exist (nothing reads them).
    const uint32_t dwords = g.desc_words;
    const bool compact = dwords != (uint32_t)kDesc4Words;
    const uint32_t dlane = (uint32_t)lane < (uint32_t)kDesc4Words ? (uint32_t)lane : (uint32_t)kDesc4Words - 1u;
    const uint32_t wlane = (uint32_t)lane < dwords ? (uint32_t)lane : dwords - 1u;
    auto load_desc = [&](uint32_t q) -> uint32_t {
        const uint32_t t = q < n_tiles ? q : n_tiles;
        return byte_offset(g.tiles32 + (size_t)t * dwords, wlane * 4u)[0];  // (cached: a descriptor shares its lines with its neighbours')
    };
    struct Tile {
        uint32_t dv;  // descriptor words, one per lane; the head masks and the projection id are only unpacked when used
        uint32_t w0lo, w0hi;
        Quad<T> a, c;
        RowQuad<RowT> r;
        Quad<T> f;  // (FAIR only)
    };
    auto rl = [&](uint32_t dv, int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane(dv, i); };
    auto window_of = [&](uint32_t w0lo, uint32_t w0hi) -> uint64_t {  // element index of the window start (0 for padding / long tiles)
        const uint32_t hi = (w0hi >> 8) & 0x1FF;
        const uint64_t W = ((uint64_t)(w0hi & 0xFFu) << 32) | w0lo;
        return hi == 0 ? 0ull : W;
    };
    auto unpack_and_issue = [&](uint32_t dv, Tile& t) {
        t.dv = dv;
        t.w0lo = rl(dv, 0);
        t.w0hi = rl(dv, 1);
        const uint64_t W = window_of(t.w0lo, t.w0hi);
        const uint64_t room = last_quad - (W >> 2);                  // quads available after the window start
        const uint32_t lim = room < 63 ? (uint32_t)room : 63u;
        const uint32_t q = (uint32_t)lane < lim ? (uint32_t)lane : lim;  // lanes past the arrays' end re-read the last quad (masked later)
        // streamed once per launch: non-temporal loads keep them from displacing the descriptors / dual vector in L2
        typedef T vec4 __attribute__((ext_vector_type(4)));
        typedef RowT rvec4 __attribute__((ext_vector_type(4)));
        const vec4 av = __builtin_nontemporal_load(byte_offset(reinterpret_cast<const vec4*>(g.a + W), q * (uint32_t)sizeof(vec4)));
        const vec4 cv = __builtin_nontemporal_load(byte_offset(reinterpret_cast<const vec4*>(g.c + W), q * (uint32_t)sizeof(vec4)));
        const rvec4 rv = __builtin_nontemporal_load(byte_offset(reinterpret_cast<const rvec4*>(reinterpret_cast<const RowT*>(g.rowidx) + W), q * (uint32_t)sizeof(rvec4)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t.a.v[j] = av[j];
            t.c.v[j] = cv[j];
            t.r.v[j] = rv[j];
        }
        if constexpr (FAIR) {
            const vec4 fv = __builtin_nontemporal_load(byte_offset(reinterpret_cast<const vec4*>(g.fair + W), q * (uint32_t)sizeof(vec4)));
#pragma unroll
            for (int j = 0; j < 4; ++j) t.f.v[j] = fv[j];
        }
    };

    // this wavefront's window tiles: slots of the (XCD-weighted) cyclic deal, fused_common.h
    // Phase order: every wavefront has window tiles and column-per-lane slices (sell.h) to walk, and half of each SIMD's wavefronts
    // take the slices FIRST.  Point-wise window tiles are memory bound; slices alternate between requesting and computing (more so
    // late in a solve, when the Newton passes multiply): a CU that always has both kinds in flight keeps its request queue fuller
    // than one whose sixteen wavefronts move through the phases together.  Same box, 100M mixed: iterations 801-900 1.603 -> 1.557
    // ms (-2.8 %), whole 1000-iteration solve 1.600 -> 1.567 s, iterations 6-35 unchanged.  (DUALIP_HIP_ABLATE=128: windows first
    // everywhere.  Wavefront 0, whose stamps feed the XCD balance, is windows-first.)
    // Every LDS plan mixes the order (round 3; same box, 10M mixed: gradient-only plan 0.4575 -> 0.4168 ms per iteration, no-LDS plan
    // 2.434 -> 2.346).  Round 2 had restricted it to the both-in-LDS plan after one build of the no-LDS plan returned wrong sums with
    // it: that was a code-generation defect -- a VGPR spill placed ahead of an exec restore, DESIGN.md section 8 -- which the build now
    // screens every object for (dualip_amd/_build.py: _spill_defects).
    const bool sell_first = !DL_ABLATE(g.ablate, 128) && ((wave >> 2) & 1);
    Deal dealw;
    uint32_t kw = 0;                            // round
    uint32_t ti = n_tiles, ti_next = n_tiles;   // schedule slots of the current / next tile (n_tiles: none)
    uint32_t dv_first = 0, dv_next = 0;
    Tile tA, tB;
    auto open_windows = [&]() __attribute__((always_inline)) {
        dealw = make_deal(kernarg_args(g).balance);
        ti = deal_slot(dealw, kernarg_args(g).balance, 0u, n_tiles);
        ti_next = deal_slot(dealw, kernarg_args(g).balance, 1u, n_tiles);
        dv_first = load_desc(ti);
        dv_next = load_desc(ti_next);
    };
    // the first two descriptors are in flight while the workgroup stages lambda and zeroes its gradient (their loads
    // are older than the prologue's, so waiting for lambda does not wait for tile data)
    if (!sell_first) open_windows();
    const WgCtx<T> w = fused_prologue<T, LAM_LDS, GRAD_LDS>(g, smem, tid, lane, wave, wg);
    // (developer aid, second binary only: DUALIP_HIP_ABLATE = 4096 / 8192 / 12288 moves stamp 1 behind the workgroup-walked columns /
    //  the single-column tiles / the K-lane slices -- with a workgroup barrier, so it is the slowest wavefront's time)
    const int stamp_at = LANES ? (DL_ABLATE(g.ablate, 3 << 12) >> 12) : 0;
    if (stamp_at == 0) stamp(g, wg, tid, 1);
    unsigned long long* bst = kernarg_args(g).bal_stamps;
    if (bst && tid == 0) bst[4 * (size_t)wg] = wall_clock64();
    const T s = w.s;
    T sd = (T)0;  // -(lambda_K - lambda_{K+1}) / gamma, K = m - 2
    if constexpr (FAIR) sd = (T)(s * (T)(g.lambda_orig[g.m - 2] - g.lambda_orig[g.m - 1]));
    // ---- single-column tiles first, in their own loop: their walker is large, latency-bound code that must not sit inside
    //      the hot loop (measured: inlined there, the extra instruction footprint cost the window tiles 6 %) ----
    {   // very long columns first: the whole workgroup walks one together (a single wavefront would set the launch's critical path)
        const uint32_t n_xlong = kernarg_args(g).n_xlong;
        for (uint32_t xt = (uint32_t)wg; xt < n_xlong; xt += (uint32_t)gridDim.x) {
            const FusedArgs<T>& gk = kernarg_args(g);
            const uint32_t dvl = byte_offset(gk.long32 + (size_t)(gk.n_long + xt) * kDesc4Words, dlane * 4u)[0];
            const uint32_t w0lo = rl(dvl, 0), w0hi = rl(dvl, 1), pidl = rl(dvl, 10);
            const ProjT<T> pl = lookup_proj(gk, w.proj_s, pidl);
            const uint64_t k0 = (((uint64_t)w0hi << 32) | w0lo) & ((1ull << 40) - 1);
            const uint64_t len = ((uint64_t)rl(dvl, 3) << 32) | rl(dvl, 2);
            const int32_t* eq_row = (gk.eq_heights && pidl != kNoProj && pidl != 0xFFFFFFFFu) ? gk.eq_heights + (size_t)pidl * kEqBuckets : nullptr;
            double ol = 0.0, ql = 0.0;  // (this column's sums of this thread: one rounded integer each)
            process_long_tile<T, RowT, LAM_LDS, true, (LANES ? 16 : 8)>(gk, pl, k0, len, w.lam_s, w.gacc, s, w.scale, tid, ol, ql, eq_row, HOT ? gk.m_hot : (int64_t)0, w.red_s, sd,
                                                      FAIR ? &fair : nullptr, gk.long32 + (size_t)(gk.n_long + xt) * kDesc4Words, HOT ? gk.m_lam : (int64_t)0, w.cold);
            fx_add_wide(acc, ol, ql, w.scale2);
        }
        if (n_xlong) __syncthreads();  // red_s is free again (the epilogue reuses it)
    }
    if constexpr (LANES) {
        if (stamp_at == 1 && kernarg_args(g).timeline) {
            __syncthreads();
            stamp(g, wg, tid, 1);
        }
    }
    // (slot -> wavefront TRANSPOSED: slot q of a round goes to wavefront q / G of workgroup q mod G.  The tiles are listed longest
    //  first: the G longest then sit on G different CUs instead of sixteen to a CU, and a handful of them no longer all land on
    //  workgroup 0)
    //  (... counted from the LAST wavefront of a workgroup down: wavefront 0's stamps feed the balance of the window tiles)
    auto walk_long = [&](uint32_t dvl, const uint32_t* desc) __attribute__((always_inline)) {
        const FusedArgs<T>& gk = kernarg_args(g);
        const uint32_t w0lo = rl(dvl, 0), w0hi = rl(dvl, 1), pidl = rl(dvl, 10);
        const ProjT<T> pl = lookup_proj(gk, w.proj_s, pidl);
        const uint64_t k0 = (((uint64_t)w0hi << 32) | w0lo) & ((1ull << 40) - 1);
        const uint64_t len = ((uint64_t)rl(dvl, 3) << 32) | rl(dvl, 2);
        const int32_t* eq_row = (gk.eq_heights && pidl != kNoProj && pidl != 0xFFFFFFFFu) ? gk.eq_heights + (size_t)pidl * kEqBuckets : nullptr;
        if constexpr (LANES) {
            // a simplex column of up to 1024 non-zeros as ONE slice of one column, read in place (sell.h, KLOG = 6): every load in
            // flight at once, the values kept in registers, straight-line passes, reductions on the DPP unit
            // (up to 1024 non-zeros the values stay in registers across the passes; 1025 .. 2048: the RELOAD variants -- only the clamped
            //  values stay, a / c / rows are re-read for the scatter -- still one wavefront, every reduction on the DPP unit, no barrier)
            if (is_simplex_kind(pl.kind) && len <= 2048 && !DL_ABLATE(gk.ablate, 1024)) {
                const int L = (int)len, H = (L + 63) >> 6, Hmin = L >> 6;
                const uint64_t kx = (((uint64_t)rl(dvl, 5) << 32) | rl(dvl, 4)) & ((1ull << 40) - 1);  // the column's place in the caller's order (primal)
                const int len_lane = (L - lane + 63) >> 6;
                constexpr int kPer = (2 + (FAIR ? 1 : 0)) * (int)(sizeof(T) / 4) + 1 + (int)(sizeof(T) / 4);
#define DL_ORIG_CASE(HM_) sell_slice<T, RowT, HM_, (HM_ * kPer > 64), LAM_LDS, HOT, FAIR, 6>(gk, w, pl, k0, H, Hmin, L, len_lane, kx, true, lane, sd, eq_row, acc, fair); break
                switch ((H + 3) >> 2) {
                    case 1: DL_ORIG_CASE(4);
                    case 2: DL_ORIG_CASE(8);
                    case 3: DL_ORIG_CASE(12);
                    case 4: DL_ORIG_CASE(16);
                    case 5: DL_ORIG_CASE(20);
                    case 6: DL_ORIG_CASE(24);
                    case 7: DL_ORIG_CASE(28);
                    default: DL_ORIG_CASE(32);
                }
#undef DL_ORIG_CASE
                return;
            }
        }
        double ol = 0.0, ql = 0.0;
        process_long_tile<T, RowT, LAM_LDS>(gk, pl, k0, len, w.lam_s, w.gacc, s, w.scale, lane, ol, ql, eq_row, HOT ? gk.m_hot : (int64_t)0, nullptr, sd,
                                            FAIR ? &fair : nullptr, desc, HOT ? gk.m_lam : (int64_t)0, w.cold);
        fx_add_wide(acc, ol, ql, w.scale2);
    };
    // The second binary deals single-column tiles and K-lane slices to WORKGROUPS statically (workgroup w owns slots w, w + G, ...
    // of a list in descending cost) and to the wavefronts of a workgroup DYNAMICALLY: a wavefront claims its workgroup's next slot
    // from a counter in LDS (one ds_add per tile, against 5-15 us of work per tile).  With two to four such tiles per wavefront a
    // static deal left one wavefront in sixteen a whole tile behind the others (MovieLens-shaped problem: workgroups busy 68 us on
    // average, the launch 101).  wg_ctr: two words at the end of the reduction scratch nothing else uses.
    uint32_t* wg_ctr = reinterpret_cast<uint32_t*>(w.red_s + 56);
    // (with the fairness stream the deal stays static -- wavefront v takes its workgroup's slots v, v + 16, ...: that stream's sum
    //  f.x is a per-wavefront double, and a deal that follows the timing would put the timing into its last bits)
    uint32_t static_next = (uint32_t)wave;
    auto claim = [&](uint32_t* ctr) -> uint32_t {
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
    if constexpr (LANES) {
        if (tid == 0) {
            wg_ctr[0] = 0u;
            wg_ctr[1] = 0u;
        }
        __syncthreads();
        const uint32_t n_long = g.n_long, G = gridDim.x;
        auto desc_at = [&](uint32_t j) -> const uint32_t* {
            const uint64_t q = (uint64_t)j * G + (uint32_t)wg;
            return kernarg_args(g).long32 + (size_t)(q < n_long ? q : (n_long ? n_long - 1u : 0u)) * kDesc4Words;
        };
        auto desc_of = [&](uint32_t j) -> uint32_t { return byte_offset(desc_at(j), dlane * 4u)[0]; };
        if (n_long) {
            uint32_t j = claim(&wg_ctr[0]);
            uint32_t dvl = desc_of(j);
            while ((uint64_t)j * G + (uint32_t)wg < n_long) {
                const uint32_t jn = claim(&wg_ctr[0]);
                const uint32_t dvn = desc_of(jn);  // (travels while the current column is walked)
                walk_long(dvl, desc_at(j));
                j = jn;
                dvl = dvn;
            }
        }
    } else {
        // (... and from the LAST workgroup down: the cyclic deal of the window tiles ends in a partial round whose tiles go to the FIRST workgroups,
        //  and workgroup 0 also writes the optimiser state when a launch carries the step -- a handle's one or two single-column tiles used to land
        //  on exactly that workgroup: config 2's launch ended 3.5 us after its other 255 workgroups, tools/timeline.py)
        for (uint32_t lt = (uint32_t)(kFusedWaves - 1 - wave) * (uint32_t)gridDim.x + ((uint32_t)gridDim.x - 1u - (uint32_t)wg); lt < g.n_long; lt += S)
            walk_long(byte_offset(kernarg_args(g).long32 + (size_t)lt * kDesc4Words, dlane * 4u)[0], kernarg_args(g).long32 + (size_t)lt * kDesc4Words);
    }
    // the slices of the long columns (K lanes per column), every wavefront, ahead of everything cheap (sell.h)
    // (dealt like the single-column tiles above: one per workgroup before any workgroup gets a second)
    if constexpr (LANES) {
        if (stamp_at == 2 && kernarg_args(g).timeline) {
            __syncthreads();
            stamp(g, wg, tid, 1);
        }
    }
    if constexpr (LANES) sell_lanes_loop<T, RowT, LAM_LDS, HOT, FAIR>(g, w, (uint32_t)wg, (uint32_t)gridDim.x, &wg_ctr[1], wave, lane, sd, acc, fair);
    if constexpr (LANES) {
        if (stamp_at == 3 && kernarg_args(g).timeline) {
            __syncthreads();
            stamp(g, wg, tid, 1);
        }
    }
    // One schedule step: `cur` holds the tile whose loads were issued a step ago; the next tile's loads go into `nxt`.
    // The loop below alternates the two register sets explicitly -- a rotating copy of freshly loaded registers would
    // force a full memory wait at the end of every step.
    auto step = [&](Tile& cur, Tile& nxt) __attribute__((always_inline)) {
        // the current tile's lambda gathers go out first: their LDS latency overlaps the descriptor unpack and the
        // issue of the next tile's loads
        uint32_t row[kSlots];
        T lam[kSlots];
        if constexpr (HOT) {  // (cold rows: four unconditional requests -- a hot lane reads lambda[0] -- then a select; see sell.h)
            const uint32_t ml = (uint32_t)g.m_lam;  // rows whose dual entry is in LDS (all of them when the whole vector fits)
            if (ml >= (uint32_t)g.m) {
#pragma unroll
                for (int j = 0; j < kSlots; ++j) {
                    row[j] = (uint32_t)cur.r.v[j];
                    lam[j] = w.lam_s[row[j]];
                }
            } else {
                T lg[kSlots];
#pragma unroll
                for (int j = 0; j < kSlots; ++j) {
                    row[j] = (uint32_t)cur.r.v[j];
                    lg[j] = g.lambda[row[j] >= ml ? row[j] : 0u];
                }
#pragma unroll
                for (int j = 0; j < kSlots; ++j) {
                    const bool cold = row[j] >= ml;
                    const T hot_val = w.lam_s[cold ? 0u : row[j]];
                    lam[j] = cold ? (T)(s * lg[j]) : hot_val;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kSlots; ++j) {
                row[j] = (uint32_t)cur.r.v[j];
                lam[j] = LAM_LDS ? w.lam_s[row[j]] : (T)(s * g.lambda[row[j]]);
            }
        }
        const uint32_t dv_cur_next = dv_next;
        const uint32_t ti_nn = deal_slot(dealw, kernarg_args(g).balance, kw + 2u, n_tiles);
        dv_next = load_desc(ti_nn);
        unpack_and_issue(dv_cur_next, nxt);

        const uint32_t hi = (cur.w0hi >> 8) & 0x1FF, lo = (cur.w0hi >> 17) & 3;
        uint32_t pid = rl(cur.dv, 10);
        if (compact) pid = (cur.w0hi >> 20) == 0xFFFu ? 0xFFFFFFFFu : (cur.w0hi >> 20);
        const ProjT<T> pj = w.proj_s[pid < (uint32_t)(kProjLds - 1) ? pid : (uint32_t)(kProjLds - 1)];
        const int kind = __builtin_amdgcn_readfirstlane(pj.kind);
        T v[kSlots], x[kSlots];
        const uint32_t e0 = 4u * (uint32_t)lane - lo, span = hi - lo;  // element j of the lane is in the tile iff e0 + j < span
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            const T t1 = (T)(cur.a.v[j] * lam[j]);     // sparse_utils.py:79
            v[j] = (T)(t1 + (T)(s * cur.c.v[j]));      // matching.py:66,142
            if constexpr (FAIR) v[j] = (T)(v[j] + (T)(sd * cur.f.v[j]));
        }
        if (is_simplex_kind(kind)) {
#pragma unroll
            for (int j = 0; j < kSlots; ++j) v[j] = (e0 + (uint32_t)j < span) ? v[j] : (T)0;  // slots outside the tile: dummy columns of zeros
            uint64_t H[kSlots];
#pragma unroll
            for (int j = 0; j < kSlots; ++j) H[j] = ((uint64_t)rl(cur.dv, 3 + 2 * j) << 32) | rl(cur.dv, 2 + 2 * j);
            const Seg4 sg = make_seg4(H);
            const int32_t* eq_row = nullptr;
            if (kind == DL_PROJ_SIMPLEX_EQ) {  // cold: the pointer is re-read from the kernel arguments
                const int32_t* eqh = kernarg_args(g).eq_heights;
                eq_row = eqh ? eqh + (size_t)pid * kEqBuckets : nullptr;
            }
            // the instruction-bound section runs at raised issue priority: measured 2-4 % on all-simplex maps, neutral on
            // mixed ones (the inverse -- loads first -- measured slower)
            __builtin_amdgcn_s_setprio(2);
            simplex_tile4(v, sg, pj, lc, x, eq_row);
            __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int j = 0; j < kSlots; ++j) x[j] = project_pointwise(v[j], pj);
        }
        T o32 = (T)0, q32 = (T)0, f32 = (T)0;
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            const T xq = (e0 + (uint32_t)j < span) ? x[j] : (T)0;  // (a clamp with lower > 0 moves the zero-filled slots)
            const T ax = (T)(cur.a.v[j] * xq);
            if (ax != (T)0) {
                if constexpr (HOT) {
                    if ((int64_t)row[j] < g.m_hot) scatter_fixed_lds(w.gacc, row[j], ax, w.scale);
                    else scatter_fixed_cold(w.cold, g.cold_grad, row[j], ax, w.scale);
                } else {
                    scatter_fixed(w.gacc, row[j], ax, w.scale);
                }
            }
            o32 = fma_exact(cur.c.v[j], xq, o32);  // (the two objective sums are not bit-specified by the reference: fused multiply-adds)
            q32 = fma_exact(xq, xq, q32);
            if constexpr (FAIR) f32 = fma_exact(cur.f.v[j], xq, f32);
            x[j] = xq;
        }
        fx_add(acc, o32, q32, w.scale2);
        if constexpr (FAIR) fair += (double)f32;
        if (g.x_out) {
            T* xw = g.x_out + window_of(cur.w0lo, cur.w0hi);
#pragma unroll
            for (int j = 0; j < kSlots; ++j)
                if (e0 + (uint32_t)j < span) __builtin_nontemporal_store(x[j], &xw[4 * (uint32_t)lane + j]);  // neighbours own the rest of the quad
        }
        kw += 1u;
        ti = ti_next;
        ti_next = ti_nn;
    };
    // (ONE copy of the slice walker -- fourteen height variants, ~100 KB of code -- and two of the much smaller window loop: with the
    //  slice walker inlined twice, all-simplex maps, whose wavefronts then ran two copies side by side, measured 2-4 % slower)
    auto windows = [&]() __attribute__((always_inline)) {
        if (ti < n_tiles) unpack_and_issue(dv_first, tA);
        while (ti < n_tiles) {
            step(tA, tB);
            if (ti >= n_tiles) break;
            step(tB, tA);
        }
    };
    if (!sell_first) {
        windows();
        bst = kernarg_args(g).bal_stamps;
        if (bst && tid == 0) bst[4 * (size_t)wg + 1] = wall_clock64();
    }
    {
        // (second binary: its handles may have few one-lane slices -- one per workgroup before any workgroup gets a second)
        // Two-phase deal (first binary): slices [0, n1) go to every wavefront of the launch as before; the rest -- a few per cent, the
        // table's tail -- only to the workgroups that have been finishing EARLY (rank rk among them, s2 wavefronts in all).  The XCDs
        // of this part do not stream at the same speed (all-simplex 100M, even deal: odd XCDs finish 5 % after even ones, the launch
        // ends 4 % after its workgroups' mean); window tiles have their weighted deal for that (Deal), the slices had nothing -- and a
        // weighted deal INSIDE their loop had cost more scalar state than it won (section 3.1).  Here the loop is the same loop, run
        // twice with other bounds; sell_balance_kernel (matching_kernels.hip) moves n1 and the membership from the launches' stamps.
        const uint32_t n_sell_all = g.n_sell;
        uint32_t n1 = n_sell_all, s2 = 0u;
        int32_t rk = -1;
        if constexpr (!LANES) {
            const int32_t* sb = kernarg_args(g).sell_bal;
            if (sb) {
                n1 = (uint32_t)__builtin_amdgcn_readfirstlane(sb[0]);
                s2 = (uint32_t)__builtin_amdgcn_readfirstlane(sb[1]);
                rk = __builtin_amdgcn_readfirstlane(sb[4 + wg]);
                n1 = n1 < n_sell_all ? n1 : n_sell_all;
            }
        }
#pragma nounroll
        for (int ph = 0; ph < 2; ++ph) {  // (ONE copy of the walker: see above)
            uint32_t q0, Sp, end;
            if (ph == 0) {
                q0 = LANES ? (uint32_t)wave * (uint32_t)gridDim.x + (uint32_t)wg : (uint32_t)wg * (uint32_t)kFusedWaves + (uint32_t)wave;
                Sp = S;
                end = n1;
            } else {
                if (rk < 0 || s2 == 0u || n1 >= n_sell_all) break;
                q0 = n1 + (uint32_t)rk * (uint32_t)kFusedWaves + (uint32_t)wave;
                Sp = s2;
                end = n_sell_all;
            }
            sell_loop<T, RowT, LAM_LDS, HOT, FAIR>(g, w, q0, Sp, end, lane, sd, acc, fair);
        }
    }
    if (sell_first) {
        open_windows();
        windows();
    }
    if (kernarg_args(g).timeline) {
        __syncthreads();
        stamp(g, wg, tid, 2);
    }
    fused_epilogue<T, GRAD_LDS, FAIR, LANES>(g, w, acc, tid, lane, wave, wg, fair);
    if (kernarg_args(g).timeline) {
        __syncthreads();
        stamp(g, wg, tid, 3);
    }
}

template <class T, class RowT, bool LAM, bool GRAD, bool HOT, bool FAIR = false>
static int launch_fused4_inst(const dl_matching* h, const FusedArgs<T>& args, hipStream_t st) {
    auto kern = matching_fused_kernel4<T, RowT, LAM, GRAD, HOT, FAIR, DL_FUSED4_LANES != 0>;
    static std::atomic<uint64_t> attr_set{0};  // per instantiation, one bit per device (the opt-in to > 64 KB of LDS is per device)
    const uint64_t bit = 1ull << (h->device & 63);
    if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
        DL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
        attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(h->n_wg), dim3(kFusedThreads), h->lds_bytes, st, args);
    DL_HIP(hipGetLastError());
    return 0;
}

template <class T, class RowT>
static int launch_fused4_rt(const dl_matching* h, const FusedArgs<T>& args, hipStream_t st) {
    if (args.fair) {  // (dl_matching_set_fairness only accepts handles whose dual vector and gradient live in LDS)
        if (h->m_hot > 0) return launch_fused4_inst<T, RowT, true, true, true, true>(h, args, st);
        return launch_fused4_inst<T, RowT, true, true, false, true>(h, args, st);
    }
    if (h->m_hot > 0) return launch_fused4_inst<T, RowT, true, true, true>(h, args, st);
    if (h->lam_lds && h->grad_lds) return launch_fused4_inst<T, RowT, true, true, false>(h, args, st);
    if (h->grad_lds) return launch_fused4_inst<T, RowT, false, true, false>(h, args, st);
    return launch_fused4_inst<T, RowT, false, false, false>(h, args, st);
}

#if DL_FUSED4_LANES
#define DL_F4_NAME(x) x##_lanes
#else
#define DL_F4_NAME(x) x
#endif
// (one value type per translation unit -- DL_FUSED4_F64 0 / 1 -- so that the four compile side by side)
#if DL_FUSED4_F64
int DL_F4_NAME(launch_fused4_f64)(const dl_matching* h, const FusedArgs<double>& args, hipStream_t st) {
    return h->row_bytes == 2 ? launch_fused4_rt<double, uint16_t>(h, args, st) : launch_fused4_rt<double, uint32_t>(h, args, st);
}
#else
int DL_F4_NAME(launch_fused4_f32)(const dl_matching* h, const FusedArgs<float>& args, hipStream_t st) {
    return h->row_bytes == 2 ? launch_fused4_rt<float, uint16_t>(h, args, st) : launch_fused4_rt<float, uint32_t>(h, args, st);
}
#endif

}  // namespace dl
