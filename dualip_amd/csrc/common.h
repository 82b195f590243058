// common.h -- shared declarations of libdualip_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "../../include/dualip_hip.h"

namespace dl {

// ---------------------------------------------------------------------------------------------------------
// status plumbing (no exception crosses the C ABI)
// ---------------------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define DL_HIP(expr)                                          \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return ::dl::hip_fail(_e, #expr); \
    } while (0)

// ---------------------------------------------------------------------------------------------------------
// environment switches
// ---------------------------------------------------------------------------------------------------------
// PLAN switches force one of a handle's kernel plans -- every plan computes the same function, and the GPU test-suite is re-run under
// each of them (INTEGRATION.md lists them).  The shipped library honours them and REPORTS them: dl_matching_info(h, 2100) is the mask of
// the ones that were set when the handle was created (bit i = dl_switch_name(i)).
// DEVELOPER switches -- ablations that skip work and give wrong results on purpose, launch-shape and tuning constants behind the
// measurements of profiles/ -- exist only in a library compiled with -DDL_DEVTOOLS (libdualip_hip_dev.so, built by
// `python -m dualip_amd._build --dev` for tools/): the shipped library never reads them and compiles their branches out.
constexpr const char* kPlanSwitches[] = {
    "DUALIP_HIP_SELL", "DUALIP_HIP_SELL_MIN_SHARE", "DUALIP_HIP_SELL_LANES", "DUALIP_HIP_SELL_LANES_MIN_SHARE", "DUALIP_HIP_SELL_MERGE_SHORT",
    "DUALIP_HIP_LANES_BINARY", "DUALIP_HIP_FLAT", "DUALIP_HIP_COMPACT", "DUALIP_HIP_HOST_PACK", "DUALIP_HIP_LDS_MODE", "DUALIP_HIP_HOT_ROWS", "DUALIP_HIP_LAM_ALL",
    "DUALIP_HIP_ROW32", "DUALIP_HIP_COLD_XCD", "DUALIP_HIP_XLONG_MIN", "DUALIP_HIP_XCD_BALANCE", "DUALIP_HIP_XCD_BALANCE_MIN_ROUNDS",
    "DUALIP_HIP_SELL_BALANCE", "DUALIP_HIP_SELL_BALANCE_PPM", "DUALIP_HIP_FUSE_APPLY", "DUALIP_HIP_TIMING", "DUALIP_HIP_TIMELINE", "DUALIP_HIP_SLAB32",
};
constexpr int kNumPlanSwitches = (int)(sizeof(kPlanSwitches) / sizeof(kPlanSwitches[0]));
const char* plan_env(const char* name);  // getenv of a name of the table above (anything else: null -- a switch is registered or it does not exist)
#ifdef DL_DEVTOOLS
inline const char* dev_env(const char* name) { return getenv(name); }
#define DL_ABLATE(word, bits) ((word) & (bits))
#else
inline const char* dev_env(const char*) { return nullptr; }
#define DL_ABLATE(word, bits) 0
#endif

// ---------------------------------------------------------------------------------------------------------
// tiles
// ---------------------------------------------------------------------------------------------------------
// A window tile is 256 non-zeros walked by one wavefront, four per lane (fused4_kernel.h: 12- or 2-dword descriptors); a single-column
// tile is one column that fits no window.  (The 64-wide tiles of rounds 1-4 -- one non-zero per lane, 16-byte TileDesc records, a second
// kernel for unaligned and tiny inputs -- are gone: such inputs are staged into aligned, padded copies, dl_matching::stage_a.)
constexpr uint32_t kNoProj = 0xFFFFu;

// device copy of dl_proj_desc in the working precision is built on the fly from this
struct ProjDev {
    int32_t kind;
    int32_t pad;
    double p0;
    double p1;
};

constexpr int kFusedThreads = 1024;            // one workgroup per CU, 16 wavefronts
constexpr int kFusedWaves = kFusedThreads / 64;
constexpr int64_t kSlabMaxRow = 65536;  // 32-bit slabs: most non-zeros of a row.  Beyond, the slabs are < 0.4 % of what a launch moves (nothing to win), and a
                                      // workgroup's share of such a row overflows 32 bits on the row-L1 grid in most launches (250M entities: all 256 did)
constexpr int kSlabMinWg = 128;      // 32-bit slabs: fewest workgroups of a handle that gets them
constexpr double kSlabHeadroom = 4.0; // ... and how many mean shares of the fullest row a workgroup's share may reach before its high words travel
constexpr int kBalMinRounds = 40;   // XCD balance: least rounds of a cyclic deal for its per-XCD table to be adapted (one round = 2.5 % then)
constexpr int kBalTail = 64;       // balance: most rounds by which two workgroups may differ (fused_common.h: Deal, table layout)
inline size_t bal_table_words(int n_wg) { return 4 + (size_t)n_wg + kBalTail + (size_t)kBalTail * (size_t)n_wg; }
constexpr int kTimelineSlots = 8;  // developer timeline (DUALIP_HIP_TIMELINE): stamps per workgroup -- 0 start, 1 prologue done, 2 loop done, 3 end, 4 step derived (a launch
                                   // that carries the optimiser step), 5 dual rows staged, 6-7 unused (a stamp inside the window loop cost the benchmark kernel 36 bytes of scratch)
constexpr int kBalLaunches = 8;     // first launches of a handle, which all adapt the table ...
constexpr int kBalEvery = 16;       // ... afterwards every kBalEvery-th launch does
constexpr size_t kLdsBudget = 160 * 1024;      // gfx950 LDS per CU
constexpr int kColdCopies = 8;                 // hot-rows plan: one array of cold-row accumulators per XCD (the chip has 8; XCC_ID & 7)
constexpr size_t kLdsScratch = 512;            // per-workgroup reduction scratch (bytes)
constexpr int kProjLdsSlots = 256;            // projection table slots in LDS (simplex.h kProjLds)
constexpr int kLogCols = 8;                    // doubles per iteration in the AGD log

// ---------------------------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------------------------
}  // namespace dl

struct dl_matching {
    int64_t m = 0, n = 0, nnz = 0;
    int val_dtype = DL_F32;
    int device = 0;
    const void* a = nullptr;  // caller-owned
    const void* c = nullptr;  // caller-owned
    void* rowidx = nullptr;   // owned, uint16 or uint32
    // dl_matching_own_inputs: the handle stops borrowing -- it keeps its own copy of the value arrays' prefix that window tiles, single-
    // column tiles and in-place slices read (everything the column-per-lane slices hold lives in the handle's transposed copies already)
    void* own_a = nullptr;    // owned, val[own_count] or null
    void* own_c = nullptr;
    // Value arrays that are not 16-byte aligned, or shorter than 1024 elements (the 256-wide tiles read 16 bytes per lane and need one full
    // round of quads): the handle reads its OWN aligned, zero-padded copies instead (round 5: such inputs used to take a second, 64-wide
    // kernel).  `a` / `c` then point at the copies, `a_src` / `c_src` at the caller's arrays (dl_matching_update_values / _costs copy again),
    // and `nnz_arr` -- the length of the arrays the tiles may read with vector loads -- is the padded length.
    void* stage_a = nullptr;  // owned, val[nnz_arr] or null
    void* stage_c = nullptr;
    const void* a_src = nullptr;
    const void* c_src = nullptr;
    int64_t nnz_arr = 0;      // elements of a / c / rowidx that exist in memory (>= nnz; == nnz unless staged)
    int64_t own_count = 0;    // elements of the owned prefix
    int64_t unsliced_end = 0; // one past the last non-zero a tile reads in place from a / c / rowidx (0: every column is sliced)
    bool owns_inputs = false;
    int row_bytes = 4;
    uint32_t* tiles = nullptr;          // owned: window descriptors (desc_words dwords each), one all-zero descriptor, single-column tiles (12 dwords each)
    int layout = 1;                     // 1 = one non-zero per lane (64-wide tiles), 4 = four per lane (256-wide tiles)
    dl::ProjDev* projs = nullptr;       // owned
    int32_t n_proj = 0;
    int64_t n_tiles = 0, n_long = 0;
    // Balance (fused_common.h: Deal): rounds of the window tiles' cyclic deal per workgroup + the tables of the last rounds, adapted
    // from per-workgroup stamps; null = every workgroup takes the same number of rounds
    int32_t* bal = nullptr;                  // owned, device, bal_table_words(n_wg) ints
    bool bal_adapts = false;                 // the WINDOWS' deal adapts (wg_balance_kernel rewrites `bal`); stamps alone do not say so: the slices' deal shares them
    unsigned long long* bal_stamps = nullptr;  // owned, device, [n_wg][4]: prologue done, wavefront 0's windows done, all done
    int32_t* sell_bal = nullptr;             // owned, device, 4 + n_wg ints: the two-phase deal of the one-lane slices (fused4_kernel.h / matching_kernels.hip:
                                             // sell_balance_kernel) -- { slices dealt to everybody, wavefronts of the second phase, share in ppm, updates ; rank[n_wg] }
    bool sell_bal_frozen = false;            // (DUALIP_HIP_SELL_BALANCE_PPM: a fixed table, for tests)
    double bal_gain = 0.3;                   // (DUALIP_HIP_BALANCE_GAIN)
    double bal_gain0 = 0.6;                  // gain of the handle's FIRST update; update k uses max(bal_gain, bal_gain0 * 0.85^k) (DUALIP_HIP_BALANCE_GAIN0): measured, profiles/r06c_balance_gain.txt
    int bal_first = dl::kBalLaunches;            // first launches of the handle, which all adapt the table (DUALIP_HIP_BALANCE_LAUNCHES)
    int bal_launches = 0;                    // launches of the handle so far
    int bal_min_rounds = dl::kBalMinRounds;      // (DUALIP_HIP_XCD_BALANCE_MIN_ROUNDS: tests adapt small problems)
    int desc_words = 12;                // layout 4: dwords per WINDOW descriptor (2: compact, every window point-wise; single-column tiles always 12)
    int64_t n_short = 0;                // layout 4: window tiles (the single-column ones follow them in the descriptor array)
    int64_t n_xlong = 0;                // layout 4: single-column tiles long enough for a whole workgroup (last in the array)
    int n_wg = 0;
    bool lam_lds = false, grad_lds = false;
    size_t lds_bytes = 0;
    int64_t mpad = 0;          // row stride of the partial slabs (elements)
    // 32-bit slabs (fp32 handles of the 256-wide layout, whole gradient in LDS, every projection bounding x): a workgroup flushes the LOW
    // words of its 64-bit LDS accumulators, and -- only when one of them does not fit 32 bits -- the high words to `slab_hi` as well, stamping
    // `slab_ovf[wg]` (and [n_wg]: "any") with the launch's epoch; the m-sized kernels add the low words of every slab and the high words of
    // the stamped ones.  The fixed-point grid is chosen so that a workgroup's share of a row NORMALLY fits (bound: slab_bcnt elements of
    // the largest row L1 norm); nothing depends on that estimate but the speed -- the integer sums are exact for any deal of the tiles.
    // Only handles that fill the chip (>= kSlabMinWg workgroups): below, a workgroup's share IS most of a row, the grid would be the whole
    // row's, and there is nothing to win.
    bool slab32 = false;
    double slab_abound = 0.0;                  // sum of |a| a workgroup's share of one row is expected to stay below: kSlabHeadroom mean shares of the
                                               // largest row L1 norm of A (a deal-invariant property of the matrix), at least max |a|
    bool slab_rows_ok = false;                 // the grid is fine enough for EVERY row (api.hip: slab_refresh_bound); else the handle keeps 64-bit slabs
    // WIDE rows (round 6): the grid is taken from the row that needs the FINEST one (so every row's rounding noise stays below 2^-20 of its own L1
    // norm), and the rows whose workgroup shares do not fit 32 bits on that grid -- the few largest -- send their high words from every
    // workgroup in every launch: a static list decided at creation, beside the dynamic overflow path (which remains for everything else).
    uint8_t* slab_wide = nullptr;              // owned, [mpad]: 1 = wide row
    int32_t* slab_wide_list = nullptr;         // owned, [mpad]: the wide rows' indices, n_wide of them
    uint32_t* slab_wide_bits = nullptr;        // owned, [1024]: bit u of word t = row t + 1024 u is wide (what the fused kernel's epilogue reads: one word per thread)
    int32_t n_wide = 0;
    int32_t* slab_hi = nullptr;                // the second half of `partial`, [n_wg][mpad]: high words (written by a workgroup only in a launch where it overflowed)
    unsigned long long* slab_ovf = nullptr;    // owned, [n_wg + 1]: epoch of the last launch in which workgroup w overflowed; [n_wg]: any workgroup
    unsigned long long slab_epoch = 0;         // fused launches of this handle so far
    void* partial = nullptr;   // owned: int64 fixed point, [n_wg][mpad] (grad_lds) or [1][mpad] (global atomics)
    unsigned long long* absmax_dev = nullptr;  // owned: scratch word of dl_matching_update_costs / _values (allocated on first use)
    int* shift_dev = nullptr;  // owned: fixed-point exponents of the latest launch ([0] gradient rows, [1] scalar sums)
    double amax = 0.0, cmax = 0.0;      // max |a|, max |c| (read once at creation: A and c must not change afterwards)
    double xmax_bounded = 0.0;          // largest |x| a bounded projection in use can return
    double pmax_unbounded = 0.0;        // largest |bound| of the one-sided projections in use
    bool has_unbounded = false;         // cone / identity columns exist: |x| is bounded through |v| per launch
    int64_t row_count_max = 0;          // most non-zeros in one row
    long long* partial_scal = nullptr;  // owned: [n_wg][2], c.x and sum x^2 per workgroup in fixed point (exponent shift_dev[1])
    size_t owned_bytes = 0;
    int ablate = 0;  // developer-only timing ablations, see FusedArgs (always 0 in the shipped library)
    uint32_t switches = 0;  // plan switches set in the environment when the handle was created (bit i = kPlanSwitches[i])
    // "hot rows" plan (dual vector / gradient too large for the LDS): rows renumbered by frequency, the m_hot most frequent
    // ones live in LDS, the cold tail goes through L2 (gathers) and 64-bit global atomics (cold_grad)
    int64_t m_hot = 0;                // 0 = plan not in use
    int64_t m_lam = 0;                // hot-rows plan: rows whose DUAL entry is staged in LDS (>= m_hot; = m when the whole dual vector fits beside a
                                      // smaller gradient -- then no tile ever gathers from L2, and only the scatter of the rows >= m_hot leaves the CU)
    double hot_fraction = 1.0;        // share of the non-zeros whose row is hot
    int32_t* row_inv = nullptr;       // owned, [m]: renumbered row -> caller's row
    int32_t* row_perm = nullptr;      // owned, [m]: caller's row -> renumbered row
    // device-resident AGD loop: its step kernels leave the renumbered dual vector in lam_perm and the cold accumulators
    // zeroed for the next launch (two launches less per iteration); valid only for the dual vector at hot_ready_lambda
    bool hot_ready = false;
    const void* hot_ready_lambda = nullptr;
    uint64_t hot_ready_owner = 0;           // uid of the dl_agd whose loop prepared them (never an address: a freed optimiser's
                                            // buffers can be handed out again at the same addresses)
    // fairness pair (dl_matching_set_fairness): rows m-2 / m-1 are dense, +f_k / -f_k on every non-zero
    const void* fair = nullptr;       // caller-owned val[nnz]
    double fair_max = 0.0;            // max |f|
    double* partial_fair = nullptr;   // owned, [n_wg]
    double* dense_ax = nullptr;       // owned, [2]: (A x) of the two rows, written after every fused launch
    void* lam_perm = nullptr;         // owned, val[m]: the dual vector in renumbered order (rebuilt every launch)
    bool cold_per_xcd = true;         // every XCD adds to its own copy of the cold accumulators through its L2 (DUALIP_HIP_COLD_XCD=0: one shared copy, device-scope atomics)
    long long* cold_grad = nullptr;   // owned, int64[kColdCopies][mpad]: accumulators of the renumbered rows >= m_hot
    // column-per-lane slices (sell.h): short columns of simplex entries, sorted by length, 64 per slice, transposed copies of
    // their values and row indices owned by the handle
    int64_t n_sell = 0, n_sell_cols = 0, n_sell_elems = 0, n_sell_nnz = 0;  // slices, their columns, slots (with padding), non-zeros
    std::vector<uint64_t> wg_preload;  // per workgroup: cost (in slice slots) of the whole-workgroup columns it walks first
    uint32_t* sell_lane_begin = nullptr;  // owned: [n_wg + 1] ranges of the K-lane slice table, one per workgroup
    int64_t long_nnz = 0;             // non-zeros in single-column tiles walked by one wavefront each
    bool lanes_binary = false;        // launches go to the fused kernel's SECOND binary; decided ONCE at creation (the order of the
                                      // single-column tiles -- descending for its dynamic deal, snake for the first binary's static one -- goes with it)
    int64_t n_sell_lane_slices = 0;   // slices with K > 1 lanes per column: the FIRST n of sell_desc (walked by their own loop)
    int64_t n_sell_lane_cols = 0;     // columns dealt to K > 1 lanes each (25 .. 512 non-zeros; sell.h)
    int64_t n_sell_mixed_cols = 0;    // columns of slices that hold more than one length (the only ones whose length bytes are read)
    uint32_t* sell_desc = nullptr;    // owned, 4 dwords per slice
    uint8_t* sell_len = nullptr;      // owned, [n_sell_cols]
    uint64_t* sell_colstart = nullptr;  // owned, [n_sell_cols]: the column's first non-zero in the caller's arrays (primal output)
    void* sell_a = nullptr;           // owned, val[n_sell_elems]
    void* sell_c = nullptr;
    void* sell_r = nullptr;           // owned, row indices (row_bytes wide)
    void* sell_f = nullptr;           // owned: fairness values in slice order (dl_matching_set_fairness)
    int32_t* eq_heights = nullptr;  // owned: simplex_eq reference-compatibility table [n_proj][32] or null (exact)
    unsigned long long* timeline = nullptr;  // developer-only (DUALIP_HIP_TIMELINE): [n_wg][kTimelineSlots] wall-clock stamps of the last launch
    // measurement hook (dl_matching_profile): event pairs around the fused-pass launches
    bool prof_on = false;
    int prof_stride = 1;      // bracket every prof_stride-th launch (the event records cost ~5 us per launch pair)
    uint64_t prof_seen = 0;
    size_t prof_used = 0;
    std::vector<hipEvent_t> prof_start, prof_stop;
};

struct dl_agd {
    uint64_t uid = 0;        // process-unique, never reused
    int64_t m = 0, max_iter = 0;
    int val_dtype = DL_F32;
    void* x = nullptr;       // owned, val[m]: point of evaluation
    void* y = nullptr;       // owned
    void* y_old = nullptr;   // owned: the dual stored with the previous history entry
    void* g = nullptr;       // owned: gradient of the latest step (A x - b)
    void* g_old = nullptr;   // owned
    const uint8_t* eq_mask = nullptr;  // caller-owned
    float* beta = nullptr;   // owned, float[max_iter]
    double* log = nullptr;   // owned, [max_iter][kLogCols]
    void* state = nullptr;   // owned, two dl::AgdDevState (double buffered)
    int state_cur = 0;
    void* x_alt = nullptr;   // owned: the buffer the next iterate is written to
    double* packed = nullptr;  // owned scratch double[m+2]: the sums the latest step used
    double* packed_blk[3] = {nullptr, nullptr, nullptr};  // owned, lazily: reduced sums of the further blocks of a split shard
    double* partial_stats = nullptr;  // owned: per-workgroup partial reductions of the step kernel
    unsigned long long* chk_partial = nullptr;  // owned, [stats blocks + 2]: hash sums of what the stats launch read from a P2P mailbox (comm.h)
    int* chk_dead = nullptr;          // the communicator's sticky error word when the latest statistics came from a mailbox, else null
};
