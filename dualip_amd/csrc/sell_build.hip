// sell_build.hip -- one-off construction of the column-per-lane slices (sell.h) on the device: per-(entry, length) histogram,
// stable radix sort of the eligible columns by (entry, length), slice table, transposed copies of the value / row arrays.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <functional>
#include <queue>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sell.h"

namespace dl {

constexpr int kSellPidSlots = 256;  // projection ids that can own slices (the kernel's LDS projection table)
constexpr int kSellBins = kSellMaxLenLanes + 2;  // histogram bins per entry: lengths 0 .. kSellMaxLenLanes, last = longer
constexpr int kSellHistLds = 16;                  // entries whose histogram is privatised in LDS (the others count in memory)

// (entry, length) histogram, privatised in LDS for the first kSellHistLds entries: a single-entry map would otherwise put every
// column on a few dozen addresses.  nnz_by_pid[2 q + 1]: non-zeros of entry q in columns longer than kSellMaxLenLanes.
template <class IdxT>
__global__ __launch_bounds__(256) void sell_hist_kernel(int64_t n, const IdxT* __restrict__ colptr, const int32_t* __restrict__ col_proj,
                                                        unsigned long long* __restrict__ hist /* [256][kSellBins] */,
                                                        unsigned long long* __restrict__ nnz_by_pid /* [256][2]: unused, longer */) {
    __shared__ unsigned int sh[kSellHistLds * kSellBins];
    __shared__ unsigned long long shl[kSellHistLds];
    for (int i = threadIdx.x; i < kSellHistLds * kSellBins; i += 256) sh[i] = 0u;
    for (int i = threadIdx.x; i < kSellHistLds; i += 256) shl[i] = 0ull;
    __syncthreads();
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t len = (int64_t)colptr[j + 1] - (int64_t)colptr[j];
        const int32_t pid = col_proj ? col_proj[j] : 0;
        if (pid < 0 || pid >= kSellPidSlots - 1 || len <= 0) continue;
        const int b = len <= kSellMaxLenLanes ? (int)len : kSellMaxLenLanes + 1;
        if (pid < kSellHistLds) {
            atomicAdd(&sh[pid * kSellBins + b], 1u);
            if (len > kSellMaxLenLanes) atomicAdd(&shl[pid], (unsigned long long)len);
        } else {
            atomicAdd(&hist[(size_t)pid * kSellBins + b], 1ull);
            if (len > kSellMaxLenLanes) atomicAdd(&nnz_by_pid[2 * pid + 1], (unsigned long long)len);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSellHistLds * kSellBins; i += 256)
        if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
    for (int i = threadIdx.x; i < kSellHistLds; i += 256)
        if (shl[i]) atomicAdd(&nnz_by_pid[2 * i + 1], shl[i]);
}

template <class IdxT>
__global__ __launch_bounds__(256) void sell_keys_kernel(int64_t n, const IdxT* __restrict__ colptr, const int32_t* __restrict__ col_proj,
                                                        const uint8_t* __restrict__ pid_sell, uint32_t* __restrict__ keys, uint32_t* __restrict__ ids, int desc) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t len = (int64_t)colptr[j + 1] - (int64_t)colptr[j];
        const int32_t pid = col_proj ? col_proj[j] : 0;
        // pid_sell[q]: 0 = entry without slices, 1 = columns of <= kSellMaxH non-zeros, 2 = of <= kSellMaxLenLanes (K lanes per column)
        const int64_t max_len = (pid >= 0 && pid < kSellPidSlots - 1) ? (pid_sell[pid] == 2 ? kSellMaxLenLanes : (pid_sell[pid] ? kSellMaxH : 0)) : 0;
        const bool ok = len >= 1 && len <= max_len;
        keys[j] = ok ? (((uint32_t)pid << 10) | (uint32_t)(desc ? 1023 - len : len)) : 0x3FFFFu;  // 18 bits (desc: longest first inside an entry; pid <= 254)
        ids[j] = (uint32_t)j;
    }
}

// one wavefront per slice
template <class IdxT>
__global__ __launch_bounds__(256) void sell_meta_kernel(uint32_t n_slices, const uint32_t* __restrict__ desc, const uint32_t* __restrict__ sorted_ids,
                                                        const IdxT* __restrict__ colptr, uint8_t* __restrict__ slen, uint64_t* __restrict__ colstart) {
    const uint32_t sl = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (sl >= n_slices) return;
    const uint32_t w1 = desc[(size_t)sl * kSellDescWords + 1], dense0 = desc[(size_t)sl * kSellDescWords + 3];
    const int ncols = (int)((w1 >> 24) & 0xFFu) + 1;
    if (lane >= ncols) return;
    const uint32_t col = sorted_ids[(size_t)dense0 + lane];
    const int64_t k0 = (int64_t)colptr[col];
    slen[(size_t)dense0 + lane] = (uint8_t)((int64_t)colptr[(size_t)col + 1] - k0);
    colstart[(size_t)dense0 + lane] = (uint64_t)k0;
}

template <class V>
__global__ __launch_bounds__(256) void sell_fill_kernel(uint32_t n_slices, const uint32_t* __restrict__ desc, const uint8_t* __restrict__ slen,
                                                        const uint64_t* __restrict__ colstart, const V* __restrict__ src, V* __restrict__ dst) {
    const uint32_t sl = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (sl >= n_slices) return;
    const uint32_t w0 = desc[(size_t)sl * kSellDescWords], w1 = desc[(size_t)sl * kSellDescWords + 1], dense0 = desc[(size_t)sl * kSellDescWords + 3];
    const uint64_t base = ((uint64_t)(w1 & 0xFFu) << 32) | w0;
    const int H = (int)((w1 >> 8) & 0xFFu), ncols = (int)((w1 >> 24) & 0xFFu) + 1;
    const int klog = (int)((desc[(size_t)sl * kSellDescWords + 2] >> 8) & 7u);  // lanes per column = 1 << klog: element e of a column at lane K col + e % K, step e / K
    const int col = lane >> klog, sub = lane & ((1 << klog) - 1);
    int len = 0;
    uint64_t k0 = 0;
    if (col < ncols) {
        len = klog == 5 ? (int)((desc[(size_t)sl * kSellDescWords + 2] >> (11 + 9 * col)) & 511u) + 1 : (int)slen[(size_t)dense0 + col];  // (K = 32: lengths in the descriptor)
        k0 = colstart[(size_t)dense0 + col];
    }
    for (int t = 0; t < H; ++t) {
        const int e = (t << klog) + sub;
        dst[base + (uint64_t)t * 64u + (uint32_t)lane] = e < len ? src[k0 + (uint64_t)e] : (V)0;
    }
}

// Decide which projection entries get slices and lay the slices out.  hist / nnz are host copies of sell_hist_kernel's output.
// An entry qualifies when it is a simplex kind and at least `min_share` of its non-zeros sit in columns of <= kSellMaxH
// non-zeros (the rest of such an entry goes to single-column tiles: windows over the leftovers would stream mostly skipped data).
// Slices run in ASCENDING length order.  Longest first (so that the last, partly filled round of the kernel's cyclic deal holds
// the cheapest slices) was measured on one box, three repetitions each (tools/ab.sh bench): 100M mixed 1.677-1.685 ms against
// 1.654-1.660 ms ascending, 12.5M 0.224-0.230 against 0.221-0.224 -- slower.  DUALIP_HIP_SELL_ORDER=desc keeps it reachable.
static bool sell_descending() {
    const char* e = dev_env("DUALIP_HIP_SELL_ORDER");
    return e && e[0] == 'd';
}

static void sell_plan(const unsigned long long* hist, const unsigned long long* nnz, const dl_proj_desc* projs, int32_t n_proj, bool single_entry, double min_share,
                      bool lanes_on, double lane_share, std::vector<uint8_t>& pid_sell, std::vector<uint32_t>& desc, uint64_t* n_cols, uint64_t* n_elems, uint64_t* n_nnz, uint64_t* n_lane_cols) {
    pid_sell.assign(kSellPidSlots, 0);
    desc.clear();
    *n_cols = 0;
    *n_elems = 0;
    *n_nnz = 0;
    *n_lane_cols = 0;
    if (n_proj <= 0 || !projs) return;
    uint64_t dense = 0, base = 0;
    // length classes by lanes per column (sell_lanes_log): K = 1 << k holds lengths lo[k] .. hi[k]
    const int lo[6] = {1, kSellMaxH + 1, 33, 65, 129, 256}, hi[6] = {kSellMaxH, 32, 64, 128, 255, kSellMaxLenLanes};
    const bool down = sell_descending();
    for (int pid = 0; pid < kSellPidSlots - 1 && pid < (single_entry ? 1 : n_proj); ++pid) {
        const int kind = projs[pid].kind;
        if (kind != DL_PROJ_SIMPLEX && kind != DL_PROJ_SIMPLEX_EQ) continue;
        if (projs[pid].flags & DL_PROJ_FLAG_NO_SLICES) continue;
        const unsigned long long* hp = hist + (size_t)pid * kSellBins;
        // An entry gets K-lane slices when its columns of 25 .. 512 non-zeros hold at least `lane_share` of its non-zeros: a handle with
        // such slices runs the second binary of the fused kernel (fused4_kernel.h), and a handful of long columns -- the benchmark's
        // Poisson(10) columns: one non-zero in 10^4 -- is not worth leaving the first.  Such an entry is sliced whatever the share of its
        // short columns (its only leftovers are columns no window could hold either); otherwise the share rule of the one-lane slices
        // applies (min_share < 0: the caller set none -- 0.9).
        double sh = 0.0, sh_lanes = 0.0, lg = (double)nnz[pid * 2 + 1];
        for (int l = 1; l <= kSellMaxLenLanes; ++l) (l <= kSellMaxH ? sh : sh_lanes) += (double)l * (double)hp[l];
        const bool lanes_entry = lanes_on && sh_lanes > 0.0 && sh_lanes >= lane_share * (sh + sh_lanes + lg);
        if (lanes_entry) sh += sh_lanes;
        else lg += sh_lanes;
        const double ms = min_share < 0.0 ? (lanes_entry ? 0.0 : 0.9) : min_share;
        if (sh <= 0.0 || sh < ms * (sh + lg)) continue;
        const int n_classes = lanes_entry ? 6 : 1;
        pid_sell[pid] = lanes_entry ? 2 : 1;
        *n_nnz += (uint64_t)sh;
        // columns of this entry in sorted order: class by class, hist[l] columns of every length l, shortest first (see sell_descending)
        // An entry with K-lane slices and only a FEW short columns (fewer one-lane slices than a launch has wavefronts) puts those into
        // the two-lane class as well: a handful of one-lane slices would form a phase of their own at the end of the launch, one slice per
        // wavefront with nothing to overlap it (MovieLens-shaped problem: 343 such slices, 15 us of an 87 us launch), while the K-lane
        // table is dealt by cost and claimed dynamically.  (Heights below 9 run the 9-step variant: a few slices' worth of padding.)
        uint64_t n_short = 0;
        for (int l = 1; l <= kSellMaxH; ++l) n_short += hp[l];
        const char* me = plan_env("DUALIP_HIP_SELL_MERGE_SHORT");  // 0: never (testing: both kinds of slices in one small handle)
        const bool merge_short = lanes_entry && n_short <= 64ull * 4096ull && !(me && me[0] == '0');
        for (int kc = 0; kc < n_classes; ++kc) {
            const int k = down ? n_classes - 1 - kc : kc;
            if (merge_short && k == 0) continue;
            const int lo_k = (merge_short && k == 1) ? 1 : lo[k];
            const uint32_t per = 64u >> k;  // columns per slice
            uint64_t cnt = 0;
            for (int l = lo_k; l <= hi[k]; ++l) cnt += hp[l];
            if (cnt == 0) continue;
            if (k > 0) *n_lane_cols += cnt;
            const int l_first = down ? hi[k] : lo_k, l_last = down ? lo_k : hi[k], dl = down ? -1 : 1;
            int l_lo = l_first;    // length of the column at the current position
            uint64_t left_lo = hp[l_first];
            auto advance = [&](int& l, uint64_t& left, uint64_t by) {  // move `by` columns forward
                while (by > 0) {
                    while (left == 0 && l != l_last) left = hp[l += dl];
                    const uint64_t step = by < left ? by : left;
                    left -= step;
                    by -= step;
                    if (step == 0) break;
                }
                while (left == 0 && l != l_last) left = hp[l += dl];
            };
            advance(l_lo, left_lo, 0);
            for (uint64_t pos = 0; pos < cnt; pos += per) {
                const uint32_t ncols = (uint32_t)(cnt - pos < per ? cnt - pos : per);
                int l_hi = l_lo;
                uint64_t left_hi = left_lo;
                advance(l_hi, left_hi, ncols - 1);  // the slice's last column
                const int len_max = down ? l_lo : l_hi, len_min = down ? l_hi : l_lo;
                const int H = (len_max + (1 << k) - 1) >> k;  // steps: the longest column's elements per lane
                const int hmin = len_min >> k;                // the fewest elements any lane of a column holds
                desc.push_back((uint32_t)base);
                desc.push_back((uint32_t)(base >> 32) | ((uint32_t)H << 8) | ((uint32_t)hmin << 16) | ((ncols - 1u) << 24));
                // (K = 32: two columns per slice, whose lengths - 1 ride in bits 11 .. 28 -- the per-column length record is one byte)
                uint32_t lens32 = 0;
                if (k == 5) {
                    const int len_first = l_lo, len_second = ncols > 1 ? l_hi : l_lo;  // in table order (ascending, or descending with DUALIP_HIP_SELL_ORDER=desc)
                    lens32 = ((uint32_t)(len_first - 1) << 11) | ((uint32_t)(len_second - 1) << 20);
                }
                desc.push_back((uint32_t)pid | ((uint32_t)k << 8) | lens32);
                desc.push_back((uint32_t)dense);
                base += (uint64_t)H * 64u;
                dense += ncols;
                advance(l_lo, left_lo, ncols);
            }
        }
    }
    *n_cols = dense;
    *n_elems = base;
}

template <class IdxT>
static int sell_prepare_typed(dl_matching* h, const IdxT* colptr, const int32_t* col_proj, const dl_proj_desc* projs, int32_t n_proj, double min_share,
                              std::vector<uint8_t>& pid_sell_out, std::vector<uint32_t>& desc, hipStream_t st) {
    pid_sell_out.assign(kSellPidSlots, 0);
    desc.clear();
    if (h->n <= 0 || h->n >= (1ll << 31) || h->nnz <= 0) return 0;
    unsigned long long* stats = nullptr;  // hist [256][kSellBins] + nnz [256][2]
    const size_t stat_words = (size_t)kSellPidSlots * (kSellBins + 2);
    DL_HIP(hipMalloc((void**)&stats, sizeof(unsigned long long) * stat_words));
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(unsigned long long) * stat_words, st);
    const int blocks = (int)std::min<int64_t>(8192, (h->n + 255) / 256);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sell_hist_kernel<IdxT>, dim3(blocks), dim3(256), 0, st, h->n, colptr, col_proj, stats, stats + (size_t)kSellPidSlots * kSellBins);
        e = hipGetLastError();
    }
    std::vector<unsigned long long> stats_h(stat_words);
    if (e == hipSuccess) e = hipMemcpyAsync(stats_h.data(), stats, sizeof(unsigned long long) * stat_words, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(stats);
    if (e != hipSuccess) return hip_fail(e, "slice statistics");
    uint64_t n_cols = 0, n_elems = 0, n_nnz = 0, n_lane_cols = 0;
    // K lanes per column for the columns of 25 .. 512 non-zeros (sell.h) unless DUALIP_HIP_SELL_LANES=0, for the entries in which they
    // hold at least DUALIP_HIP_SELL_LANES_MIN_SHARE (default 1 %) of the non-zeros (sell_plan).
    const char* le = plan_env("DUALIP_HIP_SELL_LANES");
    const bool lanes_on = !(le && le[0] == '0');
    double lane_share = 0.01;
    if (const char* ls = plan_env("DUALIP_HIP_SELL_LANES_MIN_SHARE")) lane_share = atof(ls);
    sell_plan(stats_h.data(), stats_h.data() + (size_t)kSellPidSlots * kSellBins, projs, n_proj, col_proj == nullptr, min_share, lanes_on, lane_share, pid_sell_out, desc, &n_cols,
              &n_elems, &n_nnz, &n_lane_cols);
    if (n_cols == 0 || n_cols >= (1ull << 32) || desc.size() / kSellDescWords >= (1ull << 31)) {
        pid_sell_out.assign(kSellPidSlots, 0);
        desc.clear();
        return 0;
    }
    h->n_sell = (int64_t)(desc.size() / kSellDescWords);
    h->n_sell_cols = (int64_t)n_cols;
    h->n_sell_elems = (int64_t)n_elems;
    h->n_sell_nnz = (int64_t)n_nnz;
    h->n_sell_lane_cols = (int64_t)n_lane_cols;
    h->n_sell_mixed_cols = 0;
    for (size_t t = 0; t + kSellDescWords <= desc.size(); t += kSellDescWords) {
        const uint32_t w1 = desc[t + 1];
        if (((w1 >> 8) & 0xFFu) != ((w1 >> 16) & 0xFFu)) h->n_sell_mixed_cols += (int64_t)((w1 >> 24) & 0xFFu) + 1;
    }
    return 0;
}

template <class IdxT>
static int sell_finish_typed(dl_matching* h, const IdxT* colptr, const int32_t* col_proj, const std::vector<uint8_t>& pid_sell, const std::vector<uint32_t>& desc,
                             hipStream_t st) {
    if (h->n_sell == 0) return 0;
    const uint32_t n_slices = (uint32_t)h->n_sell;
    const uint64_t n_cols = (uint64_t)h->n_sell_cols, n_elems = (uint64_t)h->n_sell_elems;
    const int blocks = (int)std::min<int64_t>(8192, (h->n + 255) / 256);
    // ---- sort the columns by (entry, length): stable, so equal keys keep the caller's column order (deterministic) ----
    uint32_t *keys = nullptr, *keys2 = nullptr;
    uint32_t *ids = nullptr, *ids2 = nullptr;
    uint8_t* flags = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    auto cleanup = [&]() {
        for (void* p : {(void*)keys, (void*)keys2, (void*)ids, (void*)ids2, (void*)flags, tmp})
            if (p) (void)hipFree(p);
    };
    const size_t n = (size_t)h->n;
    hipError_t e = hipMalloc((void**)&keys, 4 * n);
    if (e == hipSuccess) e = hipMalloc((void**)&keys2, 4 * n);
    if (e == hipSuccess) e = hipMalloc((void**)&ids, 4 * n);
    if (e == hipSuccess) e = hipMalloc((void**)&ids2, 4 * n);
    if (e == hipSuccess) e = hipMalloc((void**)&flags, kSellPidSlots);
    if (e == hipSuccess) e = hipMemcpyAsync(flags, pid_sell.data(), kSellPidSlots, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sell_keys_kernel<IdxT>, dim3(blocks), dim3(256), 0, st, h->n, colptr, col_proj, flags, keys, ids, sell_descending() ? 1 : 0);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys2, ids, ids2, (int)n, 0, 18, st);
    if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16);
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys2, ids, ids2, (int)n, 0, 18, st);
    if (e != hipSuccess) {
        cleanup();
        return hip_fail(e, "slice sort");
    }
    // ---- slice table, per-column records, transposed arrays ----
    const size_t vs = h->val_dtype == DL_F32 ? 4 : 8;
    auto own = [&](void** p, size_t bytes) -> hipError_t {
        hipError_t r = hipMalloc(p, bytes ? bytes : 16);
        if (r == hipSuccess) h->owned_bytes += bytes;
        return r;
    };
    e = own((void**)&h->sell_desc, sizeof(uint32_t) * desc.size());
    if (e == hipSuccess) e = own((void**)&h->sell_len, (size_t)n_cols);
    if (e == hipSuccess) e = own((void**)&h->sell_colstart, sizeof(uint64_t) * (size_t)n_cols);
    if (e == hipSuccess) e = own(&h->sell_a, vs * (size_t)n_elems);
    if (e == hipSuccess) e = own(&h->sell_c, vs * (size_t)n_elems);
    if (e == hipSuccess) e = own(&h->sell_r, (size_t)h->row_bytes * (size_t)n_elems);
    // Order of the table = order of the kernel's cyclic deal (wavefront W of the S takes slices W, W + S, ...).  The slices ascend in
    // height, so every full round is internally even -- but the LAST round is partly filled: N mod S wavefronts walk one more slice
    // than the others, and in ascending order it is the tallest of all (24 steps against a mean of ten at the benchmark's shape).  The
    // table is therefore rotated: the N mod S SHORTEST slices go to the end, everything else keeps its ascending order.  Matters when
    // a wavefront has few rounds (a 12.5M-entity shard of an all-simplex block: 45 rounds, the odd one worth 2.4 of them).  A
    // descriptor is self-contained (base, first column), so the build kernels below do not care about the order.
    // The slices of K > 1 lanes per column come FIRST in the table, tallest class first: the kernel walks them in their own loop
    // (sell.h: sell_lanes_loop) ahead of everything else; the one-lane slices follow, and the rotation above applies to THEIR deal.
    std::vector<uint32_t> rotated;
    const uint32_t* desc_up = desc.data();
    {
        const char* te = dev_env("DUALIP_HIP_SELL_TAIL");
        const uint64_t S = (uint64_t)(h->n_wg > 0 ? h->n_wg : 1) * (uint64_t)kFusedWaves;
        std::vector<uint32_t> lanes, plain;
        for (size_t t = 0; t + kSellDescWords <= desc.size(); t += kSellDescWords) {
            std::vector<uint32_t>& dst = ((desc[t + 2] >> 8) & 7u) ? lanes : plain;
            dst.insert(dst.end(), desc.begin() + (ptrdiff_t)t, desc.begin() + (ptrdiff_t)(t + kSellDescWords));
        }
        h->n_sell_lane_slices = (int64_t)(lanes.size() / kSellDescWords);
        const uint64_t n_plain = plain.size() / kSellDescWords;
        const uint64_t r = n_plain % S;
        const bool rotate = !(te && te[0] == '0') && !sell_descending() && r > 0 && n_plain > S;
        rotated.reserve(desc.size());
        // K-lane slices: descending cost (tallest class first), then dealt to the WORKGROUPS longest-processing-time first -- each to the
        // workgroup with the least work so far, which starts at what its whole-workgroup columns cost (h->wg_preload) -- and listed
        // workgroup by workgroup; sell_lane_begin[w .. w + 1] is workgroup w's range, whose wavefronts claim from it dynamically.
        if (!sell_descending()) {
            std::vector<uint32_t> rev;
            rev.reserve(lanes.size());
            for (size_t t = lanes.size(); t >= (size_t)kSellDescWords; t -= kSellDescWords) rev.insert(rev.end(), lanes.begin() + (ptrdiff_t)(t - kSellDescWords), lanes.begin() + (ptrdiff_t)t);
            lanes.swap(rev);
        }
        const size_t G = (size_t)(h->n_wg > 0 ? h->n_wg : 1);
        std::vector<uint32_t> lane_begin(G + 1, 0);
        if (!lanes.empty()) {
            typedef std::pair<uint64_t, uint32_t> Load;  // (work so far, workgroup)
            std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
            for (size_t wgi = 0; wgi < G; ++wgi) heap.push(Load(wgi < h->wg_preload.size() ? h->wg_preload[wgi] : 0, (uint32_t)wgi));
            const size_t n_lane = lanes.size() / kSellDescWords;
            std::vector<uint32_t> owner(n_lane);
            for (size_t t = 0; t < n_lane; ++t) {
                const uint32_t Hs = (lanes[t * kSellDescWords + 1] >> 8) & 0xFFu;
                Load l = heap.top();
                heap.pop();
                owner[t] = l.second;
                ++lane_begin[l.second + 1];
                l.first += (uint64_t)(Hs + 2u) * 64u;
                heap.push(l);
            }
            for (size_t wgi = 0; wgi < G; ++wgi) lane_begin[wgi + 1] += lane_begin[wgi];
            std::vector<uint32_t> cursor(lane_begin.begin(), lane_begin.end() - 1);
            rotated.resize(lanes.size());
            for (size_t t = 0; t < n_lane; ++t) {
                const size_t at = (size_t)cursor[owner[t]]++;
                std::copy(lanes.begin() + (ptrdiff_t)(t * kSellDescWords), lanes.begin() + (ptrdiff_t)((t + 1) * kSellDescWords), rotated.begin() + (ptrdiff_t)(at * kSellDescWords));
            }
        }
        {
            hipError_t eb = hipMalloc((void**)&h->sell_lane_begin, sizeof(uint32_t) * (G + 1));
            if (eb == hipSuccess) {
                h->owned_bytes += sizeof(uint32_t) * (G + 1);
                eb = hipMemcpy(h->sell_lane_begin, lane_begin.data(), sizeof(uint32_t) * (G + 1), hipMemcpyHostToDevice);
            }
            if (eb != hipSuccess && e == hipSuccess) e = eb;
        }
        const size_t head = rotate ? (size_t)r * kSellDescWords : 0;
        rotated.insert(rotated.end(), plain.begin() + (ptrdiff_t)head, plain.end());
        rotated.insert(rotated.end(), plain.begin(), plain.begin() + (ptrdiff_t)head);
        desc_up = rotated.data();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h->sell_desc, desc_up, sizeof(uint32_t) * desc.size(), hipMemcpyHostToDevice, st);
    const unsigned sblocks = (n_slices + 3u) / 4u;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sell_meta_kernel<IdxT>, dim3(sblocks), dim3(256), 0, st, n_slices, h->sell_desc, ids2, colptr, h->sell_len, h->sell_colstart);
        if (h->val_dtype == DL_F32) {
            hipLaunchKernelGGL(sell_fill_kernel<float>, dim3(sblocks), dim3(256), 0, st, n_slices, h->sell_desc, h->sell_len, h->sell_colstart, (const float*)h->a, (float*)h->sell_a);
            hipLaunchKernelGGL(sell_fill_kernel<float>, dim3(sblocks), dim3(256), 0, st, n_slices, h->sell_desc, h->sell_len, h->sell_colstart, (const float*)h->c, (float*)h->sell_c);
        } else {
            hipLaunchKernelGGL(sell_fill_kernel<double>, dim3(sblocks), dim3(256), 0, st, n_slices, h->sell_desc, h->sell_len, h->sell_colstart, (const double*)h->a, (double*)h->sell_a);
            hipLaunchKernelGGL(sell_fill_kernel<double>, dim3(sblocks), dim3(256), 0, st, n_slices, h->sell_desc, h->sell_len, h->sell_colstart, (const double*)h->c, (double*)h->sell_c);
        }
        if (h->row_bytes == 2)
            hipLaunchKernelGGL(sell_fill_kernel<uint16_t>, dim3(sblocks), dim3(256), 0, st, n_slices, h->sell_desc, h->sell_len, h->sell_colstart, (const uint16_t*)h->rowidx, (uint16_t*)h->sell_r);
        else
            hipLaunchKernelGGL(sell_fill_kernel<uint32_t>, dim3(sblocks), dim3(256), 0, st, n_slices, h->sell_desc, h->sell_len, h->sell_colstart, (const uint32_t*)h->rowidx, (uint32_t*)h->sell_r);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // desc is the caller's host vector; the sort buffers are freed below
    cleanup();
    if (e != hipSuccess) return hip_fail(e, "slice construction");
    return 0;
}

// Step 1 (before the window tiles are packed): which entries get slices, and the slice table.  On return pid_sell[q] != 0
// marks the entries whose columns of <= kSellMaxH non-zeros will live in slices; h->n_sell / n_sell_cols / n_sell_elems are set.
int sell_prepare(dl_matching* h, const void* colptr, int idx_dtype, const int32_t* col_proj, const dl_proj_desc* projs, int32_t n_proj, double min_share,
                 std::vector<uint8_t>& pid_sell, std::vector<uint32_t>& desc, hipStream_t st) {
    if (idx_dtype == DL_I64) return sell_prepare_typed<int64_t>(h, (const int64_t*)colptr, col_proj, projs, n_proj, min_share, pid_sell, desc, st);
    return sell_prepare_typed<int32_t>(h, (const int32_t*)colptr, col_proj, projs, n_proj, min_share, pid_sell, desc, st);
}
// Step 2 (after the row indices are final: re-encoded, renumbered under the hot-rows plan): sort, transpose, upload.
int sell_finish(dl_matching* h, const void* colptr, int idx_dtype, const int32_t* col_proj, const std::vector<uint8_t>& pid_sell, const std::vector<uint32_t>& desc,
                hipStream_t st) {
    if (idx_dtype == DL_I64) return sell_finish_typed<int64_t>(h, (const int64_t*)colptr, col_proj, pid_sell, desc, st);
    return sell_finish_typed<int32_t>(h, (const int32_t*)colptr, col_proj, pid_sell, desc, st);
}

static int sell_fill_values(dl_matching* h, const void* src, void* dst, hipStream_t st) {
    const unsigned sblocks = ((unsigned)h->n_sell + 3u) / 4u;
    if (h->val_dtype == DL_F32)
        hipLaunchKernelGGL(sell_fill_kernel<float>, dim3(sblocks), dim3(256), 0, st, (uint32_t)h->n_sell, h->sell_desc, h->sell_len, h->sell_colstart, (const float*)src, (float*)dst);
    else
        hipLaunchKernelGGL(sell_fill_kernel<double>, dim3(sblocks), dim3(256), 0, st, (uint32_t)h->n_sell, h->sell_desc, h->sell_len, h->sell_colstart, (const double*)src, (double*)dst);
    DL_HIP(hipGetLastError());
    return 0;
}

// fairness values (dl_matching_set_fairness) in slice order
int sell_fill_fair(dl_matching* h, const void* f_values, hipStream_t st) {
    if (h->n_sell == 0) return 0;
    const size_t vs = h->val_dtype == DL_F32 ? 4 : 8;
    if (!h->sell_f) {
        DL_HIP(hipMalloc(&h->sell_f, vs * (size_t)h->n_sell_elems));
        h->owned_bytes += vs * (size_t)h->n_sell_elems;
    }
    return sell_fill_values(h, f_values, h->sell_f, st);
}

// the caller rewrote a in place (dl_matching_update_values): refresh the slices' copy
int sell_refill_values(dl_matching* h, hipStream_t st) {
    if (h->n_sell == 0) return 0;
    return sell_fill_values(h, h->a, h->sell_a, st);
}

// the caller rewrote c in place (dl_matching_update_costs): refresh the slices' copy
int sell_refill_costs(dl_matching* h, hipStream_t st) {
    if (h->n_sell == 0) return 0;
    return sell_fill_values(h, h->c, h->sell_c, st);
}

}  // namespace dl
