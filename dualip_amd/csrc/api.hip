// api.hip -- extern "C" surface of libdualip_hip.so (include/dualip_hip.h) and the one-off set-up work:
// row-index re-encoding, wave-tile packing and workgroup partitioning.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <atomic>
#include <chrono>

#include "comm.h"

namespace dl {

static thread_local char g_err[512] = "";

const char* plan_env(const char* name) {
    for (int i = 0; i < kNumPlanSwitches; ++i)
        if (strcmp(name, kPlanSwitches[i]) == 0) return getenv(name);
    return nullptr;  // not registered in common.h: does not exist
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return -(int)e;
}

// implemented in matching_kernels.hip / agd_kernels.hip
int matching_calculate(dl_matching* h, const void* lambda, double gamma, double* packed_out, void* x_out, hipStream_t st);
int launch_epilogue(int64_t m, int val_dtype, const double* packed, const void* b, const void* lam, double gamma, void* grad_out,
                    double* scal_out, hipStream_t st);
size_t agd_state_bytes();
int agd_state_init(void* dev_state, double initial_step, double max_step, hipStream_t st);
int agd_state_read_max_step(void* dev_state, int cur, double* out, hipStream_t st);
size_t agd_partial_stats_bytes(int64_t m);
int launch_project_dense(int64_t L, int64_t K, int val_dtype, const void* in, void* out, const dl_proj_desc* p, hipStream_t st);
int launch_jacobi(int64_t m, int64_t nnz, const void* rowidx, int idx_dtype, void* a, void* b, void* norms, int val_dtype, hipStream_t st);
int launch_absmax(int val_dtype, int64_t n, const void* v, unsigned long long* out_bits, hipStream_t st);
int cold_xcd_selftest(hipStream_t st);
int launch_row_stats(int64_t nnz, const void* rows, int row_bytes, const float* a, int64_t m, float* out_host, int* measured, hipStream_t st);
size_t fused_lds_bytes(int64_t m, int val_dtype, bool lam, bool grad);
size_t fused_lds_bytes2(int64_t rows_grad, int64_t rows_lam, int val_dtype);
int sell_prepare(dl_matching* h, const void* colptr, int idx_dtype, const int32_t* col_proj, const dl_proj_desc* projs, int32_t n_proj, double min_share,
                 std::vector<uint8_t>& pid_sell, std::vector<uint32_t>& desc, hipStream_t st);  // sell_build.hip
int sell_finish(dl_matching* h, const void* colptr, int idx_dtype, const int32_t* col_proj, const std::vector<uint8_t>& pid_sell, const std::vector<uint32_t>& desc,
                hipStream_t st);
int pack_device(int64_t n, int64_t nnz, const void* colptr, int idx_dtype, const int32_t* col_proj, int32_t n_proj, const std::vector<uint8_t>& pid_sell,
                const std::vector<uint8_t>& pid_flat, int compact,
                uint32_t** win_dev_out, int64_t* n_win_out, std::vector<uint32_t>& long_words, std::vector<uint8_t>& used, hipStream_t st);  // pack_build.hip
int sell_fill_fair(dl_matching* h, const void* f_values, hipStream_t st);
int sell_refill_costs(dl_matching* h, hipStream_t st);
int sell_refill_values(dl_matching* h, hipStream_t st);
constexpr int kSellMaxLen = 24;        // sell.h: kSellMaxH
constexpr int kSellMaxLenLanes = 512;  // sell.h: longest column of an entry sliced with K lanes per column (pid_sell == 2)

// ---- row index re-encoding: caller's int32/int64 -> uint16 (m <= 65536) or uint32 ----
constexpr int kReencLdsRows = 16384;  // rows whose histogram a workgroup keeps in LDS (64 KB)
template <class SrcT, class DstT>
__global__ __launch_bounds__(256) void reencode_rows_kernel(int64_t nnz, const SrcT* __restrict__ src, DstT* __restrict__ dst, int64_t m, int* __restrict__ bad,
                                                            unsigned int* __restrict__ row_count) {
    // one-off row histogram (bounds the fixed-point gradient accumulators): privatised per workgroup in LDS when the rows fit --
    // 10^9 global atomics on 10^4 addresses made this kernel 67 ms of a 120 ms handle creation at 100M entities
    __shared__ unsigned int sh[kReencLdsRows];
    const bool in_lds = m <= kReencLdsRows;
    if (in_lds) {
        for (int i = threadIdx.x; i < (int)m; i += blockDim.x) sh[i] = 0u;
        __syncthreads();
    }
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = (int64_t)src[k];
        if (r < 0 || r >= m) {
            *bad = 1;
            dst[k] = 0;
        } else {
            dst[k] = (DstT)r;
            if (in_lds) atomicAdd(&sh[r], 1u);
            else atomicAdd(&row_count[r], 1u);
        }
    }
    if (in_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < (int)m; i += blockDim.x)
            if (sh[i]) atomicAdd(&row_count[i], sh[i]);
    }
}

// hot-rows plan: re-encoded row ids -> ids renumbered by frequency (in place, one-off)
template <class DstT>
__global__ void remap_rows_kernel(int64_t nnz, DstT* __restrict__ rows, const int32_t* __restrict__ perm) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) rows[k] = (DstT)perm[rows[k]];
}

template <class SrcT>
static int reencode_rows(dl_matching* h, const void* rowidx, hipStream_t st, int* bad_dev, unsigned int* row_count) {
    const int threads = 256;
    int64_t b64 = (h->nnz + threads - 1) / threads;
    const int blocks = (int)(b64 > 8192 ? 8192 : (b64 > 0 ? b64 : 1));
    if (h->row_bytes == 2)
        hipLaunchKernelGGL((reencode_rows_kernel<SrcT, uint16_t>), dim3(blocks), dim3(threads), 0, st, h->nnz, (const SrcT*)rowidx, (uint16_t*)h->rowidx, h->m, bad_dev, row_count);
    else
        hipLaunchKernelGGL((reencode_rows_kernel<SrcT, uint32_t>), dim3(blocks), dim3(threads), 0, st, h->nnz, (const SrcT*)rowidx, (uint32_t*)h->rowidx, h->m, bad_dev, row_count);
    DL_HIP(hipGetLastError());
    return 0;
}

// one past the last non-zero of a column whose ENTRY has no column-per-lane slices (window tiles and their single-column tiles read
// those in place)
template <class IdxT>
__global__ void unsliced_end_kernel(int64_t n, const IdxT* __restrict__ colptr, const int32_t* __restrict__ col_proj, const uint8_t* __restrict__ pid_sell, int32_t n_sell_flags,
                                    int32_t n_proj, unsigned long long* __restrict__ out) {
    unsigned long long best = 0ull;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k0 = (int64_t)colptr[j], k1 = (int64_t)colptr[j + 1];
        const int64_t len = k1 - k0;
        if (len <= 0) continue;
        const int32_t pid = col_proj ? col_proj[j] : (n_proj > 0 ? 0 : -1);
        const uint8_t fl = (pid >= 0 && pid < n_sell_flags) ? pid_sell[pid] : (uint8_t)0;
        // (columns of a SLICED entry that are too long for a slice are single-column tiles read in place, scattered over the entry's
        //  range: dl_matching_own_inputs moves those into a pool -- only the columns of entries without slices define the prefix)
        if (fl == 0) best = (unsigned long long)k1 > best ? (unsigned long long)k1 : best;
    }
    best = (unsigned long long)wave_allreduce((long long)best, OpMax());
    if ((threadIdx.x & 63) == 0 && best) atomicMax(out, best);
}

// copies straggler columns (single-column tiles of sliced entries) into the owned arrays' pool: one block per column
template <class T, class RowT>
__global__ void pool_copy_kernel(const uint64_t* __restrict__ src, const uint64_t* __restrict__ dst, const uint64_t* __restrict__ len, const T* __restrict__ a, const T* __restrict__ c,
                                 const RowT* __restrict__ r, T* __restrict__ oa, T* __restrict__ oc, RowT* __restrict__ orow) {
    const uint64_t s0 = src[blockIdx.x], d0 = dst[blockIdx.x], n = len[blockIdx.x];
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) {
        oa[d0 + i] = a[s0 + i];
        oc[d0 + i] = c[s0 + i];
        orow[d0 + i] = r[s0 + i];
    }
}

static int owned_malloc(dl_matching* h, void** p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    h->owned_bytes += bytes;
    return 0;
}

static void matching_free(dl_matching* h) {
    if (!h) return;
    if (h->rowidx) (void)hipFree(h->rowidx);
    if (h->own_a) (void)hipFree(h->own_a);
    if (h->own_c) (void)hipFree(h->own_c);
    if (h->stage_a) (void)hipFree(h->stage_a);
    if (h->stage_c) (void)hipFree(h->stage_c);
    if (h->tiles) (void)hipFree(h->tiles);
    if (h->projs) (void)hipFree(h->projs);
    if (h->partial) (void)hipFree(h->partial);
    if (h->slab_ovf) (void)hipFree(h->slab_ovf);
    if (h->slab_wide) (void)hipFree(h->slab_wide);
    if (h->slab_wide_list) (void)hipFree(h->slab_wide_list);
    if (h->slab_wide_bits) (void)hipFree(h->slab_wide_bits);
    if (h->partial_scal) (void)hipFree(h->partial_scal);
    if (h->shift_dev) (void)hipFree(h->shift_dev);
    if (h->absmax_dev) (void)hipFree(h->absmax_dev);
    if (h->timeline) (void)hipFree(h->timeline);
    if (h->eq_heights) (void)hipFree(h->eq_heights);
    if (h->row_inv) (void)hipFree(h->row_inv);
    if (h->row_perm) (void)hipFree(h->row_perm);
    if (h->partial_fair) (void)hipFree(h->partial_fair);
    if (h->dense_ax) (void)hipFree(h->dense_ax);
    if (h->lam_perm) (void)hipFree(h->lam_perm);
    if (h->cold_grad) (void)hipFree(h->cold_grad);
    if (h->bal) (void)hipFree(h->bal);
    if (h->bal_stamps) (void)hipFree(h->bal_stamps);
    if (h->sell_bal) (void)hipFree(h->sell_bal);
    if (h->sell_lane_begin) (void)hipFree(h->sell_lane_begin);
    for (void* p : {(void*)h->sell_desc, (void*)h->sell_len, (void*)h->sell_colstart, h->sell_a, h->sell_c, h->sell_r, h->sell_f})
        if (p) (void)hipFree(p);
    for (hipEvent_t e : h->prof_start) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->prof_stop) (void)hipEventDestroy(e);
    delete h;
}



// Layout 4 schedule.  The fused kernel deals descriptor q to wavefront q mod S (S = 16 * workgroups) as its (q / S)-th
// tile, so the ORDER of the descriptor array is the schedule (a descriptor is self-contained: window start, masks,
// projection).  Simplex tiles are instruction bound, point-wise tiles memory bound: when a problem has both, the two lists
// (each kept in memory order -- two sweep fronts) are merged so that (a) every wavefront alternates between the kinds in
// proportion to their counts and (b) the four wavefronts that share a SIMD are at different phases of that alternation,
// i.e. the SIMD's vector unit works on a simplex tile while its other wavefronts wait on point-wise loads.
// Phase = Kronecker sequence frac((4 k + rho) * golden), k = tile number of the wavefront, rho = its SIMD-mate class.
static void schedule_tiles4(std::vector<uint32_t>& words, std::vector<uint32_t>& tile_pid, const dl_proj_desc* projs, int32_t n_proj, int n_wg) {
    const size_t n = tile_pid.size();
    if (n == 0 || n_wg <= 0) return;
    std::vector<uint32_t> heavy, light;
    for (size_t t = 0; t < n; ++t) {
        const uint32_t pid = tile_pid[t];
        const bool long_tile = (words[t * 12 + 1] & (1u << 19)) != 0;
        const int kind = (pid == kNoProj || (int32_t)pid >= n_proj) ? DL_PROJ_NONE : projs[pid].kind;
        const bool hv = !long_tile && (kind == DL_PROJ_SIMPLEX || kind == DL_PROJ_SIMPLEX_EQ);
        (hv ? heavy : light).push_back((uint32_t)t);
    }
    if (heavy.empty() || light.empty()) return;  // one kind: memory order is the schedule
    const size_t S = (size_t)n_wg * kFusedWaves;
    std::vector<uint32_t> out_words(words.size(), 0u), out_pid(n);
    std::vector<uint32_t> order(n);
    {
        const double r = (double)heavy.size() / (double)n;
        size_t ih = 0, il = 0;
        for (size_t q = 0; q < n; ++q) {
            const size_t k = q / S, wave = (q % S) % kFusedWaves;
            const size_t rho = ((wave >> 2) + wave) & 3;
            double v = (double)(4 * k + rho) * 0.6180339887498949;
            v -= (double)(uint64_t)v;
            const bool want_heavy = v < r;
            const bool take_heavy = (want_heavy && ih < heavy.size()) || il == light.size();
            order[q] = take_heavy ? heavy[ih++] : light[il++];
        }
    }
    for (size_t q = 0; q < n; ++q) {
        const uint32_t t = order[q];
        memcpy(&out_words[q * 12], &words[(size_t)t * 12], 12 * sizeof(uint32_t));
        out_pid[q] = tile_pid[t];
    }
    words.swap(out_words);  // (the trailing all-zero descriptor stays all-zero)
    tile_pid.swap(out_pid);
}

// Layout 4: 16-byte-aligned 256-element windows of whole columns; 12 dwords per tile (see matching_kernels4.hip).
// pid_sell[q] != 0: the columns of entry q with <= kSellMaxLen non-zeros live in column-per-lane slices (sell.h) and are skipped
// here; the entry's longer columns become single-column tiles (windows over such leftovers would stream mostly skipped data).
static int pack_tiles4(int64_t n, int64_t nnz, const int64_t* colptr, const int32_t* col_proj, int32_t n_proj, const dl_proj_desc* projs,
                       std::vector<uint32_t>& words, std::vector<uint64_t>& cost_prefix, std::vector<uint32_t>& tile_pid, int64_t* n_long,
                       const std::vector<uint8_t>& pid_sell, const std::vector<uint8_t>& pid_flat) {
    auto weight = [&](uint32_t pj) -> uint64_t {
        if (pj == kNoProj || (int32_t)pj >= n_proj) return 10;
        const int k = projs[pj].kind;
        return (k == DL_PROJ_SIMPLEX || k == DL_PROJ_SIMPLEX_EQ) ? 26 : 10;
    };
    words.clear();
    cost_prefix.clear();
    cost_prefix.push_back(0);
    tile_pid.clear();
    *n_long = 0;
    const uint64_t nnz_al4 = (uint64_t)nnz & ~3ull;
    uint64_t running = 0;
    bool open = false;
    uint64_t W = 0, H[4] = {0, 0, 0, 0};
    uint32_t lo = 0, end = 0, cur_proj = kNoProj;
    auto set_head = [&](uint32_t e) { H[e & 3] |= 1ull << (e >> 2); };
    auto emit = [&](uint64_t w0, const uint64_t* h4, uint32_t pid) {
        words.push_back((uint32_t)w0);
        words.push_back((uint32_t)(w0 >> 32));
        for (int j = 0; j < 4; ++j) {
            words.push_back((uint32_t)h4[j]);
            words.push_back((uint32_t)(h4[j] >> 32));
        }
        words.push_back(pid == kNoProj ? 0xFFFFFFFFu : pid);
        words.push_back(0u);
        tile_pid.push_back(pid);
    };
    auto flush = [&]() {
        if (!open) return;
        if (end < 256) set_head(end);  // sentinel: elements past the last column form their own dummy segment
        const uint64_t w0 = W | ((uint64_t)end << 40) | ((uint64_t)lo << 49);
        emit(w0, H, cur_proj);
        running += 256 * weight(cur_proj);
        cost_prefix.push_back(running);
        open = false;
        H[0] = H[1] = H[2] = H[3] = 0;
    };
    for (int64_t j = 0; j < n; ++j) {
        const int64_t k0 = colptr[j], k1 = colptr[j + 1];
        const int64_t len = k1 - k0;
        if (len < 0) return fail(DL_E_LAYOUT, "ccol_indices is not monotone at column %lld", (long long)j);
        if (len == 0) continue;
        if ((uint64_t)k1 >= (1ull << 40)) return fail(DL_E_ARG, "nnz exceeds 2^40");
        int32_t pid = col_proj ? col_proj[j] : (n_proj > 0 ? 0 : -1);
        if (pid >= n_proj) return fail(DL_E_PROJ, "column %lld refers to projection %d but only %d were given", (long long)j, pid, n_proj);
        const uint32_t pj = pid < 0 ? kNoProj : (uint32_t)pid;
        const bool sliced = pj != kNoProj && pj < pid_sell.size() && pid_sell[pj];
        if (sliced && len <= (pid_sell[pj] == 2 ? kSellMaxLenLanes : kSellMaxLen)) {
            flush();  // (a window holds consecutive columns only)
            continue;
        }
        const bool tail_quad = (uint64_t)k1 > nnz_al4;  // touches the array's last partial quad: no vector loads there
        // point-wise entry (pid_flat; last element = columns with no entry): windows are cut every 256 non-zeros wherever they fall
        const bool flat = !pid_flat.empty() && (pj == kNoProj ? pid_flat.back() != 0 : (pj < 254u && (size_t)pj + 1 < pid_flat.size() && pid_flat[pj]));
        const bool flat_align = flat && (pj == kNoProj ? pid_flat.back() : pid_flat[pj]) == 2;
        if ((len > 253 && !flat) || tail_quad || sliced || (pj != kNoProj && pj >= (uint32_t)kProjLdsSlots - 1)) {
            flush();
            const uint64_t h4[4] = {(uint64_t)len, 0, 0, 0};
            emit((uint64_t)k0 | (1ull << 51), h4, pj);
            running += (uint64_t)len * weight(pj) * 3;
            cost_prefix.push_back(running);
            *n_long += 1;
            continue;
        }
        if (flat) {
            if (open && pj != cur_proj) flush();
            uint64_t k = (uint64_t)k0;
            while (k < (uint64_t)k1) {
                if (!open) {
                    open = true;
                    W = k & ~3ull;
                    lo = (uint32_t)(k - W);
                    cur_proj = pj;
                }
                // cut at absolute multiples of 256 non-zeros: every window but a run's first then covers whole 128-byte lines of
                // the three arrays (1 KB / 1 KB / 512 B spans), none shared with its neighbours
                const uint64_t cut = flat_align ? ((W + 256) & ~255ull) : W + 256;
                const uint64_t stop = (uint64_t)k1 < cut ? (uint64_t)k1 : cut;
                end = (uint32_t)(stop - W);
                k = stop;
                if (W + end == cut) flush();
            }
            continue;
        }
        if (open && ((uint64_t)k1 > W + 256 || pj != cur_proj)) flush();
        if (!open) {
            open = true;
            W = (uint64_t)k0 & ~3ull;
            lo = (uint32_t)((uint64_t)k0 - W);
            cur_proj = pj;
            set_head(0);
        }
        set_head((uint32_t)((uint64_t)k0 - W));
        end = (uint32_t)((uint64_t)k1 - W);
    }
    flush();
    return 0;
}

}  // namespace dl

using namespace dl;

extern "C" {

const char* dl_last_error_string(void) { return g_err; }
int dl_version(void) { return 302; }  // ABI version: _hip.py ABI_VERSION must agree

// 32-bit slabs: the fixed-point grid and the list of WIDE rows (common.h).
//
// Every a x is rounded to ONE grid before its integer add -- step <= slab_abound xmax 2^-29 -- and a workgroup flushes the low 32-bit words of its
// sums.  Round 5 took the grid from the LARGEST row (slab_abound = kSlabHeadroom mean workgroup shares of the largest row L1 norm, so that every
// share fits 32 bits): rows far smaller than the largest then collect rounding noise that is large against THEIR sums -- count_i roundings, RMS
// step sqrt(count_i / 12) -- where the reference's fp32 scatter_add_ (sparse_utils.py:236-243) is accurate relative to each row (round-5 review).
// Round 6 takes the grid from the row that needs the FINEST one:
//      step sqrt(count_i / 12) <= 2^-20 L1_i xmax   for EVERY row with a non-zero value
// (the rounding a row collects stays below 2^-20 = 1e-6 of its own L1 norm times the bound of x: what fp32 accumulation of a few hundred terms
// gives, 2^-24 sqrt(count)), i.e. slab_abound = min over the rows of 2^29 2^-20 sqrt(12 / count_i) L1_i, never above round 5's value.  The rows whose workgroup
// share may then NOT fit 32 bits -- min(L1_i, max(kSlabHeadroom L1_i / workgroups, max_i |a|)) > slab_abound: the few largest rows -- are WIDE:
// every workgroup sends their high words in every launch (a static list; the dynamic path -- a workgroup some OTHER share of which overflows
// sends all its high words, stamped with the launch's epoch -- remains).  The integer sums stay exact and deal-invariant either way; only
// traffic depends on the list.  The handle keeps 64-bit slabs instead when more than one row in sixteen would be wide (nothing left to win: row
// scales spread over many orders of magnitude), when an element would not fit the 2^51 of the float -> fixed conversion on that grid, or when the
// rows could not be measured.  dl_matching_info(h, 2010) = 1: the criterion holds; (h, 2011): wide rows.
// DUALIP_HIP_SLAB32=force: round 5's grid whatever the rows (A/B of the criterion, tests); =tiny: a grid on which every share overflows.
static int slab_refresh_bound(dl_matching* h, hipStream_t st) {
    const double n_wg = (double)(h->n_wg > 0 ? h->n_wg : 1);
    std::vector<float> rs(3 * (size_t)(h->m > 0 ? h->m : 1));
    int measured = 0;
    int rc = launch_row_stats(h->nnz, h->rowidx, h->row_bytes, static_cast<const float*>(h->a), h->m, rs.data(), &measured, st);
    if (rc) return rc;
    const char* se = plan_env("DUALIP_HIP_SLAB32");
    const bool forced = se && (se[0] == 'f' || se[0] == 't');
    double l1max = 0.0, need = INFINITY;
    if (measured) {
        for (int64_t i = 0; i < h->m; ++i) {
            const double l1 = rs[(size_t)i], cnt = rs[(size_t)(h->m + i)];
            l1max = std::max(l1max, l1);
            if (cnt > 0.0 && l1 > 0.0) need = std::min(need, 512.0 * std::sqrt(12.0 / cnt) * l1);  // 2^29 * 2^-20 = 2^9
        }
    } else {
        l1max = h->amax * (double)(h->row_count_max > 0 ? h->row_count_max : 1);  // (rows beyond the LDS table: the count-based bound)
    }
    const double coarse = std::max(h->amax, kSlabHeadroom * l1max / n_wg);  // round 5's grid: every share fits
    h->n_wide = 0;
    if (forced || !measured) {
        h->slab_abound = coarse;
        h->slab_rows_ok = measured && need >= coarse;
        if (!forced) h->slab32 = false;  // (not measured: no criterion, no 32-bit slabs)
    } else {
        double ab = std::min(coarse, need);
        const double floor_el = h->amax * ldexp(1.0, -19);  // an element a x <= amax xmax must stay below 2^50 grid units (fused_common.h: to_fixed)
        h->slab_rows_ok = ab >= floor_el;
        ab = std::max(ab, floor_el);
        std::vector<uint8_t> wide((size_t)(h->mpad > 0 ? h->mpad : 1), 0);
        std::vector<int32_t> list;
        for (int64_t i = 0; i < h->m; ++i) {
            const double l1 = rs[(size_t)i], mx = rs[(size_t)(2 * h->m + i)];
            if (std::min(l1, std::max(kSlabHeadroom * l1 / n_wg, mx)) > ab) {
                wide[(size_t)i] = 1;
                list.push_back((int32_t)i);
            }
        }
        if (!h->slab_rows_ok || (int64_t)list.size() * 16 > h->m || h->m > 32 * (int64_t)kFusedThreads) {
            h->slab32 = false;  // (64-bit slabs from here on: the allocation is sized for them; never switched back on)
            h->slab_rows_ok = false;
            h->slab_abound = coarse;
        } else {
            h->slab_abound = ab;
            h->n_wide = (int32_t)list.size();
            if (!h->slab_wide) {
                int rcw = owned_malloc(h, (void**)&h->slab_wide, wide.size());
                if (!rcw) rcw = owned_malloc(h, (void**)&h->slab_wide_list, sizeof(int32_t) * wide.size());
                if (!rcw) rcw = owned_malloc(h, (void**)&h->slab_wide_bits, sizeof(uint32_t) * kFusedThreads);
                if (rcw) return rcw;
            }
            std::vector<uint32_t> bits((size_t)kFusedThreads, 0u);
            for (int32_t i : list) bits[(size_t)(i % kFusedThreads)] |= 1u << ((i / kFusedThreads) & 31);
            DL_HIP(hipMemcpyAsync(h->slab_wide_bits, bits.data(), sizeof(uint32_t) * bits.size(), hipMemcpyHostToDevice, st));
            DL_HIP(hipMemcpyAsync(h->slab_wide, wide.data(), wide.size(), hipMemcpyHostToDevice, st));
            if (!list.empty()) DL_HIP(hipMemcpyAsync(h->slab_wide_list, list.data(), sizeof(int32_t) * list.size(), hipMemcpyHostToDevice, st));
            DL_HIP(hipStreamSynchronize(st));  // (host vectors)
        }
    }
    if (se && se[0] == 't') h->slab_abound = h->amax / 256.0;  // "tiny": the test hook -- every workgroup's shares overflow
    return 0;
}

int dl_matching_create(dl_matching** out, int64_t m, int64_t n, int64_t nnz, const void* colptr, const void* rowidx, int idx_dtype,
                       const void* a, const void* c, int val_dtype, const dl_proj_desc* projs_host, int32_t n_proj, const int32_t* col_proj,
                       dl_stream_t stream) {
    return dl_matching_create2(out, m, n, nnz, colptr, idx_dtype, rowidx, idx_dtype, a, c, val_dtype, projs_host, n_proj, col_proj, stream);
}

int dl_matching_create2(dl_matching** out, int64_t m, int64_t n, int64_t nnz, const void* colptr, int idx_dtype, const void* rowidx, int row_dtype,
                        const void* a, const void* c, int val_dtype, const dl_proj_desc* projs_host, int32_t n_proj, const int32_t* col_proj,
                        dl_stream_t stream) {
    if (!out) return fail(DL_E_ARG, "out is null");
    *out = nullptr;
    if (m < 0 || n < 0 || nnz < 0) return fail(DL_E_ARG, "negative size");
    if (!colptr || (nnz > 0 && (!rowidx || !a || !c))) return fail(DL_E_ARG, "null CSC array");
    if (idx_dtype != DL_I32 && idx_dtype != DL_I64) return fail(DL_E_ARG, "bad idx_dtype %d", idx_dtype);
    if (row_dtype != DL_I32 && row_dtype != DL_I64 && row_dtype != DL_U16) return fail(DL_E_ARG, "bad row_dtype %d", row_dtype);
    if (row_dtype == DL_U16 && m > 65536) return fail(DL_E_ARG, "16-bit row indices with m = %lld rows", (long long)m);
    if (val_dtype != DL_F32 && val_dtype != DL_F64) return fail(DL_E_ARG, "bad val_dtype %d", val_dtype);
    if (n_proj < 0 || n_proj >= (int32_t)kNoProj || (n_proj > 0 && !projs_host)) return fail(DL_E_ARG, "bad projection table");
    if (m >= (1ll << 32)) return fail(DL_E_ARG, "m must be < 2^32");
    for (int32_t q = 0; q < n_proj; ++q) {
        const int k = projs_host[q].kind;
        if (k < DL_PROJ_NONE || k > DL_PROJ_SIMPLEX_EQ) return fail(DL_E_PROJ, "Unknown projection operator kind %d", k);
        if ((k == DL_PROJ_SIMPLEX || k == DL_PROJ_SIMPLEX_EQ) && !(projs_host[q].p0 > 0.0))
            return fail(DL_E_PROJ, "Simplex radius z must be positive.");
        if (projs_host[q].flags & ~DL_PROJ_FLAG_NO_SLICES)
            return fail(DL_E_PROJ, "projection flags %d are not available inside the fused pass (entry %d)", (int)projs_host[q].flags, (int)q);
    }
    hipStream_t st = (hipStream_t)stream;
    dl_matching* h = new (std::nothrow) dl_matching();
    if (!h) return fail(DL_E_NOMEM, "out of host memory");
    h->m = m;
    h->n = n;
    h->nnz = nnz;
    h->val_dtype = val_dtype;
    h->a = a;
    h->c = c;
    h->n_proj = n_proj;
    const char* abl = dev_env("DUALIP_HIP_ABLATE");  // (null in the shipped library)
    h->ablate = abl ? atoi(abl) : 0;
    (void)hipGetDevice(&h->device);
    int rc = 0;
#define CK(expr)                    \
    do {                            \
        rc = (expr);                \
        if (rc) {                   \
            matching_free(h);       \
            return rc;              \
        }                           \
    } while (0)
#define CKH(expr)                                   \
    do {                                            \
        hipError_t _e = (expr);                     \
        if (_e != hipSuccess) {                     \
            matching_free(h);                       \
            return hip_fail(_e, #expr);             \
        }                                           \
    } while (0)

    // developer aid: DUALIP_HIP_TIMING=1 prints the wall time of every set-up phase to stderr
    const bool timing = plan_env("DUALIP_HIP_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_phase = now();
    auto phase = [&](const char* name) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const double t = now();
        fprintf(stderr, "[dl_matching_create] %-28s %.1f ms\n", name, (t - t_phase) * 1e3);
        t_phase = t;
    };
    // the 256-wide tiles (16-byte loads) need 16-byte aligned value arrays and at least one full round of quads
    const bool aligned = (((uintptr_t)a | (uintptr_t)c) & 15u) == 0;
    h->nnz_arr = nnz;
    if (!aligned || nnz < 1024) {
        // unaligned or tiny value arrays: aligned, zero-padded copies owned by the handle (common.h: stage_a) -- one layout, one kernel
        const size_t vs_st = val_dtype == DL_F32 ? 4 : 8;
        const int64_t npad = std::max<int64_t>(1024, (nnz + 3) & ~(int64_t)3);
        CK(owned_malloc(h, &h->stage_a, (size_t)npad * vs_st));
        CK(owned_malloc(h, &h->stage_c, (size_t)npad * vs_st));
        CKH(hipMemsetAsync(h->stage_a, 0, (size_t)npad * vs_st, st));
        CKH(hipMemsetAsync(h->stage_c, 0, (size_t)npad * vs_st, st));
        if (nnz > 0) {
            CKH(hipMemcpyAsync(h->stage_a, a, (size_t)nnz * vs_st, hipMemcpyDeviceToDevice, st));
            CKH(hipMemcpyAsync(h->stage_c, c, (size_t)nnz * vs_st, hipMemcpyDeviceToDevice, st));
        }
        h->a_src = a;
        h->c_src = c;
        a = h->stage_a;
        c = h->stage_c;
        h->a = a;
        h->c = c;
        h->nnz_arr = npad;
    }
    h->layout = 4;  // (the only one since round 5; dl_matching_info(h, 8) still reports it)
    // column-per-lane slices for the short columns of simplex entries (sell.h): decided before the windows are packed.
    // DUALIP_HIP_SELL=0 switches them off; DUALIP_HIP_SELL_MIN_SHARE = least share of an entry's non-zeros in short columns.
    std::vector<uint8_t> pid_sell;
    std::vector<uint32_t> sell_desc_h;
    {
        const char* se = plan_env("DUALIP_HIP_SELL");
        double min_share = -1.0;  // default: 0.9, or none when columns of up to 255 non-zeros can be sliced (sell_build.hip)
        if (const char* ms = plan_env("DUALIP_HIP_SELL_MIN_SHARE")) min_share = atof(ms);
        if (!(se && se[0] == '0')) CK(sell_prepare(h, colptr, idx_dtype, col_proj, projs_host, n_proj, min_share, pid_sell, sell_desc_h, st));
    }
    // point-wise entries (box, cone, identity): their windows need not hold whole columns (pack_build.hip); DUALIP_HIP_FLAT=0 keeps
    // whole-column windows.  Element n_proj: columns with no entry.
    std::vector<uint8_t> pid_flat;
    const char* flat_env = plan_env("DUALIP_HIP_FLAT");  // 0: whole-column windows everywhere; 1: cut every 256 from the run's start; default: at multiples of 256
    if (!(flat_env && flat_env[0] == '0')) {
        const uint8_t mode = (flat_env && flat_env[0] == '1') ? 1 : 2;
        pid_flat.assign((size_t)n_proj + 1, 0);
        for (int32_t q = 0; q < n_proj; ++q) {
            const int k = projs_host[q].kind;
            pid_flat[(size_t)q] = (k != DL_PROJ_SIMPLEX && k != DL_PROJ_SIMPLEX_EQ) ? mode : 0;
        }
        pid_flat[(size_t)n_proj] = mode;
    }
    phase("slice plan");
    // Window tiles are packed on the device (pack_build.hip) unless the map has BOTH instruction-bound window tiles (a simplex
    // entry that is not sliced) and memory-bound ones -- those want the host's interleaved schedule (schedule_tiles4) -- or
    // DUALIP_HIP_HOST_PACK=1 asks for the host path (kept as the independent implementation for cross-checks).
    bool dev_pack = !plan_env("DUALIP_HIP_HOST_PACK") && n < (1ll << 31);
    for (int32_t q = 0; q < n_proj && dev_pack; ++q) {
        const int k = projs_host[q].kind;
        if ((k == DL_PROJ_SIMPLEX || k == DL_PROJ_SIMPLEX_EQ) && !((size_t)q < pid_sell.size() && pid_sell[(size_t)q]) && n_proj > 1) dev_pack = false;
    }
    // ---- column pointers and per-column projection ids to the host (one-off; host packing only) ----
    std::vector<int64_t> colptr_h;
    std::vector<int32_t> col_proj_h;
    if (!dev_pack) {
        colptr_h.resize((size_t)n + 1);
        if (idx_dtype == DL_I64) {
            CKH(hipMemcpyAsync(colptr_h.data(), colptr, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyDeviceToHost, st));
            CKH(hipStreamSynchronize(st));
        } else {
            std::vector<int32_t> tmp((size_t)n + 1);
            CKH(hipMemcpyAsync(tmp.data(), colptr, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyDeviceToHost, st));
            CKH(hipStreamSynchronize(st));
            for (size_t i = 0; i <= (size_t)n; ++i) colptr_h[i] = tmp[i];
        }
        if (colptr_h[0] != 0 || colptr_h[(size_t)n] != nnz) {
            matching_free(h);
            return fail(DL_E_LAYOUT, "ccol_indices[0] must be 0 and ccol_indices[n] must equal nnz");
        }
        if (col_proj) {
            col_proj_h.resize((size_t)n);
            CKH(hipMemcpyAsync(col_proj_h.data(), col_proj, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
            CKH(hipStreamSynchronize(st));
        }
    } else {  // the two ends of the column pointers
        int64_t ends[2] = {0, 0};
        const size_t isz = idx_dtype == DL_I64 ? 8 : 4;
        int32_t e32[2] = {0, 0};
        void* dst0 = idx_dtype == DL_I64 ? (void*)&ends[0] : (void*)&e32[0];
        void* dst1 = idx_dtype == DL_I64 ? (void*)&ends[1] : (void*)&e32[1];
        CKH(hipMemcpyAsync(dst0, colptr, isz, hipMemcpyDeviceToHost, st));
        CKH(hipMemcpyAsync(dst1, (const char*)colptr + isz * (size_t)n, isz, hipMemcpyDeviceToHost, st));
        CKH(hipStreamSynchronize(st));
        if (idx_dtype != DL_I64) {
            ends[0] = e32[0];
            ends[1] = e32[1];
        }
        if (ends[0] != 0 || ends[1] != nnz) {
            matching_free(h);
            return fail(DL_E_LAYOUT, "ccol_indices[0] must be 0 and ccol_indices[n] must equal nnz");
        }
    }

    phase("colptr / col_proj to host");
    // ---- tiles ----
    std::vector<uint32_t> words4, tile_pid4;
    std::vector<uint64_t> prefix;
    uint32_t* win_dev = nullptr;  // device path: window descriptors in memory order
    int64_t n_win_dev = 0;
    std::vector<uint8_t> used_dev;
    struct WinGuard {
        uint32_t*& p;
        ~WinGuard() {
            if (p) (void)hipFree(p);
        }
    } win_guard{win_dev};
    if (dev_pack) {
        std::vector<uint32_t> long_list;
        // compact window table (2 dwords instead of 12): every entry that can have windows is point-wise (a sliced simplex entry has
        // none: its long columns are single-column tiles).  DUALIP_HIP_COMPACT=0 keeps the 12-dword table.
        bool compact = !pid_flat.empty() && pid_flat.back() != 0 && !(plan_env("DUALIP_HIP_COMPACT") && plan_env("DUALIP_HIP_COMPACT")[0] == '0');
        for (int32_t q = 0; q < n_proj && compact; ++q)
            if (!pid_flat[(size_t)q] && !((size_t)q < pid_sell.size() && pid_sell[(size_t)q])) compact = false;
        h->desc_words = compact ? 2 : 12;
        CK(pack_device(n, h->nnz_arr, colptr, idx_dtype, col_proj, n_proj, pid_sell, pid_flat, compact ? 1 : 0, &win_dev, &n_win_dev, long_list, used_dev, st));
        words4 = long_list;  // single-column tiles only; split / ordered below
        for (size_t t = 0; t < long_list.size() / 12; ++t) tile_pid4.push_back(long_list[t * 12 + 10] == 0xFFFFFFFFu ? kNoProj : long_list[t * 12 + 10]);
        h->n_long = (int64_t)(long_list.size() / 12);
        h->n_tiles = n_win_dev + h->n_long;
        prefix.assign(1, 0);
    } else {
        CK(pack_tiles4(n, h->nnz_arr, colptr_h.data(), col_proj ? col_proj_h.data() : nullptr, n_proj, projs_host, words4, prefix, tile_pid4, &h->n_long, pid_sell, pid_flat));
        h->n_tiles = (int64_t)(words4.size() / 12);
    }
    if (h->n_tiles >= (1ll << 31)) {
        matching_free(h);
        return fail(DL_E_ARG, "too many tiles");
    }

    phase(dev_pack ? "window packing (device)" : "window packing (host)");
    // ---- workgroups: one per CU; descriptors in
    //      schedule order, dealt cyclically to the wavefronts (schedule_tiles4) ----
    hipDeviceProp_t prop;
    CKH(hipGetDeviceProperties(&prop, h->device));
    int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const char* wg_env = dev_env("DUALIP_HIP_NUM_WG");
    if (wg_env && atoi(wg_env) > 0) n_cu = atoi(wg_env);
    int64_t want = (h->n_tiles + h->n_sell + kFusedWaves - 1) / kFusedWaves;  // at least one tile per wavefront
    h->n_wg = (int)(want < n_cu ? want : n_cu);
    if (h->n_wg < 1) h->n_wg = (h->n_tiles + h->n_sell) > 0 ? 1 : 0;
    {
        // descriptor array = [window tiles in schedule order | one all-zero descriptor (what slots past the end read) |
        // single-column tiles]: the single-column walker runs in its own loop, outside the hot one
        // single-column tiles of more than xlong_min non-zeros are walked by a whole workgroup (listed last): one wavefront
        // walking thousands of non-zeros alone would set the critical path of the launch
        std::vector<uint32_t> short_words, long_words, short_pid, long_pid, xlong_words, xlong_pid;
        // (the second binary walks simplex columns of up to 2048 non-zeros by ONE wavefront as an in-place slice -- 32 steps, values
        //  re-read for the scatter -- with every reduction on the DPP unit; the whole-workgroup walker pays two barriers per reduction,
        //  ~14 us per ~2000 non-zeros on the MovieLens shape, and stalls the other fifteen wavefronts meanwhile.  Which binary a handle
        //  takes depends on the split itself (>= 1 % of the non-zeros in one-wavefront tiles), so: split at 1024, decide, split again.)
        const char* xl_env = plan_env("DUALIP_HIP_XLONG_MIN");
        bool has_lane_slices = false;
        for (size_t t = 0; t + 4 <= sell_desc_h.size() && !has_lane_slices; t += 4) has_lane_slices = ((sell_desc_h[t + 2] >> 8) & 7u) != 0;
        // which binary the handle's launches take (matching_kernels.hip: launch_fused4): the second one for handles with K-lane slices (only it
        // walks them) or with >= 1 % of their non-zeros in one-wavefront single-column tiles; DUALIP_HIP_LANES_BINARY=0|1 forces either on
        // handles free to use both (testing)
        auto takes_second_binary = [&]() {
            bool lb = h->long_nnz > 0 && h->long_nnz * 100 >= nnz;
            if (const char* e = plan_env("DUALIP_HIP_LANES_BINARY")) lb = e[0] == '1';
            return lb || has_lane_slices;
        };
        uint64_t xlong_min = 1024;
        for (int pass = 0; pass < 2; ++pass) {
            if (xl_env) xlong_min = strtoull(xl_env, nullptr, 10);
            short_words.clear(); long_words.clear(); short_pid.clear(); long_pid.clear(); xlong_words.clear(); xlong_pid.clear();
            h->long_nnz = 0;
            for (size_t t = 0; t < tile_pid4.size(); ++t) {
                const bool is_long = (words4[t * 12 + 1] & (1u << 19)) != 0;
                const uint64_t len = ((uint64_t)words4[t * 12 + 3] << 32) | words4[t * 12 + 2];
                if (is_long && len <= xlong_min) h->long_nnz += (int64_t)len;  // non-zeros walked one wavefront per column
                std::vector<uint32_t>& wv = !is_long ? short_words : (len > xlong_min ? xlong_words : long_words);
                std::vector<uint32_t>& pv = !is_long ? short_pid : (len > xlong_min ? xlong_pid : long_pid);
                wv.insert(wv.end(), words4.begin() + (ptrdiff_t)(t * 12), words4.begin() + (ptrdiff_t)(t * 12 + 12));
                if (is_long) {  // words 4 / 5: where the column's PRIMAL goes (the caller's order) -- the same place the column is read from, until
                                // dl_matching_own_inputs moves a straggler of a sliced entry into the handle's pool and rewrites words 0 / 1
                    wv[wv.size() - 12 + 4] = words4[t * 12];
                    wv[wv.size() - 12 + 5] = words4[t * 12 + 1] & 0xFFu;
                }
                pv.push_back(tile_pid4[t]);
            }
            if (pass == 1 || xl_env || !takes_second_binary() || xlong_min == 2048) break;
            xlong_min = 2048;  // the second binary: its one-wavefront in-place slices reach 2048 non-zeros (fused4_kernel.h: walk_long)
        }
        // longest first, dealt in snake order (wavefront W takes slots W, W + S, ...: reversing every other round pairs the
        // longest columns with the shortest ones); the workgroup-walked ones likewise over the workgroups
        // (handles of the fused kernel's second binary -- K-lane slices, or >= 1 % of the non-zeros in single-column tiles; same rule
        //  as matching_kernels.hip: wants_lanes_binary -- deal these tiles to the wavefronts of a workgroup dynamically: plain descending order)
        const bool lanes_binary = takes_second_binary();
        h->lanes_binary = lanes_binary;  // read once, here: the binary a launch takes and the tile order below must agree (matching_kernels.hip: launch_fused4)
        auto snake = [lanes_binary](std::vector<uint32_t>& wv, std::vector<uint32_t>& pv, size_t width) {
            const size_t nt = pv.size();
            if (nt < 2 || width == 0) return;
            std::vector<size_t> order(nt);
            for (size_t i = 0; i < nt; ++i) order[i] = i;
            auto len_of = [&](size_t t) { return ((uint64_t)wv[t * 12 + 3] << 32) | wv[t * 12 + 2]; };
            std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return len_of(x) > len_of(y); });
            for (size_t r0 = width; r0 < nt && !lanes_binary; r0 += 2 * width) std::reverse(order.begin() + (ptrdiff_t)r0, order.begin() + (ptrdiff_t)std::min(r0 + width, nt));
            std::vector<uint32_t> w2(wv.size()), p2(nt);
            for (size_t i = 0; i < nt; ++i) {
                std::copy(wv.begin() + (ptrdiff_t)(order[i] * 12), wv.begin() + (ptrdiff_t)(order[i] * 12 + 12), w2.begin() + (ptrdiff_t)(i * 12));
                p2[i] = pv[order[i]];
            }
            wv.swap(w2);
            pv.swap(p2);
        };
        if (!dev_env("DUALIP_HIP_NO_SNAKE")) {
            snake(long_words, long_pid, (size_t)h->n_wg * (size_t)kFusedWaves);
            snake(xlong_words, xlong_pid, (size_t)h->n_wg);
        }
        h->n_xlong = (int64_t)xlong_pid.size();
        // What a workgroup's columns outside the slice table cost it, in slice slots: the K-lane slices are dealt around it (sell_build.hip:
        // longest-processing-time first).  Whole-workgroup columns: 16 slots per non-zero -- swept on the MovieLens-shaped problem in
        // round 4 (same box, fused kernel per launch: 0 -> 66.8 / 67.3 us, 1.5 -> 65.4, 2.5 -> 65.1 / 63.7, 4 -> 64.2 / 62.6, 9 -> 61.5 /
        // 59.9, 16 -> 57.3 / 57.5): far more than the column's own time would say (14 us of its workgroup for 9 254 non-zeros ~ 2.5 slots
        // per non-zero) -- a workgroup that starts its slices late also walks them slower (its sixteen wavefronts enter the slice loop
        // together, in step, after the walker's last barrier).  One-wavefront single-column tiles of the second binary (dealt w, w + G, ...
        // in descending order) are counted with their non-zeros + a fixed part (no measurable effect: 0 / 128 / 512 / off within noise).
        h->wg_preload.assign((size_t)(h->n_wg > 0 ? h->n_wg : 1), 0);
        uint64_t xlong_cost10 = 160, long_fixed = 128;  // (DUALIP_HIP_XLONG_COST10, DUALIP_HIP_LONG_FIXED: calibration runs)
        if (const char* e = dev_env("DUALIP_HIP_XLONG_COST10")) xlong_cost10 = strtoull(e, nullptr, 10);
        if (const char* e = dev_env("DUALIP_HIP_LONG_FIXED")) long_fixed = strtoull(e, nullptr, 10);
        for (size_t t = 0; t < xlong_pid.size() && h->n_wg > 0; ++t)
            h->wg_preload[t % (size_t)h->n_wg] += xlong_cost10 * (((uint64_t)xlong_words[t * 12 + 3] << 32) | xlong_words[t * 12 + 2]) / 10;
        if (lanes_binary && long_fixed < (1ull << 40))
            for (size_t t = 0; t < long_pid.size() && h->n_wg > 0; ++t)
                h->wg_preload[t % (size_t)h->n_wg] += (((uint64_t)long_words[t * 12 + 3] << 32) | long_words[t * 12 + 2]) + long_fixed;
        long_words.insert(long_words.end(), xlong_words.begin(), xlong_words.end());
        long_pid.insert(long_pid.end(), xlong_pid.begin(), xlong_pid.end());
        if (!dev_pack && !dev_env("DUALIP_HIP_NO_INTERLEAVE")) schedule_tiles4(short_words, short_pid, projs_host, n_proj, h->n_wg);
        h->n_short = dev_pack ? n_win_dev : (int64_t)short_pid.size();
        words4 = short_words;  // (device path: empty -- the window descriptors are already on the device)
        words4.resize(words4.size() + 12, 0u);
        words4.insert(words4.end(), long_words.begin(), long_words.end());
        tile_pid4 = short_pid;
        tile_pid4.insert(tile_pid4.end(), long_pid.begin(), long_pid.end());
    }

    phase("schedule (host)");
    // ---- LDS plan ----
    auto lds_need = [&](bool lam, bool grad) { return fused_lds_bytes(m, val_dtype, lam, grad); };
    const char* mode_env = plan_env("DUALIP_HIP_LDS_MODE");  // "both" | "grad" | "none": force a smaller plan (testing)
    int max_mode = 2;
    if (mode_env) max_mode = !strcmp(mode_env, "none") ? 0 : (!strcmp(mode_env, "grad") ? 1 : 2);
    if (max_mode >= 2 && lds_need(true, true) <= kLdsBudget) {
        h->lam_lds = true;
        h->grad_lds = true;
    } else if (max_mode >= 1 && lds_need(false, true) <= kLdsBudget) {
        h->lam_lds = false;
        h->grad_lds = true;
    } else {
        h->lam_lds = false;
        h->grad_lds = false;
    }
    h->lds_bytes = lds_need(h->lam_lds, h->grad_lds);
    // hot-rows plan (256-wide layout): when the dual vector and the gradient do not both fit the LDS, renumber the rows by
    // frequency and keep the m_hot most frequent ones in LDS; the cold tail goes through L2 gathers and global atomics.
    // DUALIP_HIP_HOT_ROWS: "0" disables the plan, "N" forces m_hot = N (testing).
    {
        const char* hot_env = plan_env("DUALIP_HIP_HOT_ROWS");
        int64_t forced = hot_env ? atoll(hot_env) : -1;
        const bool allowed = nnz > 0 && max_mode >= 2 && forced != 0;
        int64_t m_hot = 0;
        if (allowed && forced > 0 && forced < m && fused_lds_bytes(forced, val_dtype, true, true) <= kLdsBudget) {
            m_hot = forced;
        } else if (allowed && !(h->lam_lds && h->grad_lds)) {
            int64_t lo = 0, hi = m;  // largest row count whose dual vector + gradient fit
            while (lo < hi) {
                const int64_t mid = (lo + hi + 1) / 2;
                if (fused_lds_bytes(mid, val_dtype, true, true) <= kLdsBudget) lo = mid;
                else hi = mid - 1;
            }
            m_hot = lo / 64 * 64;
            if (m_hot < 1024 || m_hot >= m) m_hot = 0;
        }
        if (m_hot > 0) {
            h->m_hot = m_hot;
            h->m_lam = m_hot;
            h->lam_lds = true;
            h->grad_lds = true;
            h->lds_bytes = fused_lds_bytes(m_hot, val_dtype, true, true);
            // Which rows get which kind of LDS?  A gather of a cold row is a dependent L2 round trip in every slice that has one
            // (MovieLens shape, same box: 70.6 us per launch, 59.8 with the cold gathers ablated); a scatter to a cold row is a
            // fire-and-forget global atomic (ablated: 69.9).  So when the WHOLE dual vector fits beside a still useful number of
            // gradient rows, stage all of it and keep fewer gradient rows: no tile gathers from L2 at all.
            // DUALIP_HIP_LAM_ALL=0 keeps the symmetric plan (m_lam = m_hot).
            const size_t vs_l = val_dtype == DL_F32 ? 4 : 8;
            const char* la = plan_env("DUALIP_HIP_LAM_ALL");
            if (!(la && la[0] == '0') && forced <= 0 && (size_t)m * vs_l * 100 <= kLdsBudget * 72) {
                int64_t lo = 0, hi = m_hot;  // most gradient rows that fit beside all m dual entries
                while (lo < hi) {
                    const int64_t mid = (lo + hi + 1) / 2;
                    if (fused_lds_bytes2(mid, m, val_dtype) <= kLdsBudget) lo = mid;
                    else hi = mid - 1;
                }
                const int64_t g_rows = lo / 64 * 64;
                if (g_rows >= 1024) {
                    h->m_hot = g_rows;
                    h->m_lam = m;
                    h->lds_bytes = fused_lds_bytes2(g_rows, m, val_dtype);
                }
            }
        }
    }
    h->mpad = (m + 63) / 64 * 64;
    if (h->mpad == 0) h->mpad = 64;

    // ---- device metadata ----
    h->row_bytes = m <= 65536 ? 2 : 4;
    const char* row_env = plan_env("DUALIP_HIP_ROW32");
    if (row_env && row_env[0] == '1') h->row_bytes = 4;
    CK(owned_malloc(h, &h->rowidx, (size_t)h->nnz_arr * (size_t)h->row_bytes));
    if (h->nnz_arr > nnz) CKH(hipMemsetAsync(h->rowidx, 0, (size_t)h->nnz_arr * (size_t)h->row_bytes, st));  // (staged: the padding's rows are row 0, its values 0)
    const size_t win_bytes = dev_pack ? sizeof(uint32_t) * (size_t)h->desc_words * (size_t)n_win_dev : 0;  // device-packed windows precede the host-built part
    const size_t tile_bytes = win_bytes + sizeof(uint32_t) * words4.size();
    CK(owned_malloc(h, (void**)&h->tiles, tile_bytes));
    CK(owned_malloc(h, (void**)&h->projs, sizeof(ProjDev) * (size_t)(n_proj > 0 ? n_proj : 1)));
    const size_t vs = val_dtype == DL_F32 ? 4 : 8;
    (void)vs;
    const size_t slabs = h->grad_lds ? (size_t)(h->n_wg > 0 ? h->n_wg : 1) : 1;
    CK(owned_malloc(h, &h->partial, slabs * (size_t)h->mpad * sizeof(long long)));
    CK(owned_malloc(h, (void**)&h->shift_dev, 2 * sizeof(unsigned long long) + sizeof(int)));
    CK(owned_malloc(h, (void**)&h->partial_scal, sizeof(long long) * 2 * (size_t)(h->n_wg > 0 ? h->n_wg : 1)));
    if (h->m_hot > 0) {
        CK(owned_malloc(h, (void**)&h->row_inv, sizeof(int32_t) * (size_t)m));
        CK(owned_malloc(h, (void**)&h->row_perm, sizeof(int32_t) * (size_t)m));
        CK(owned_malloc(h, &h->lam_perm, (size_t)m * (val_dtype == DL_F32 ? 4 : 8)));
        CK(owned_malloc(h, (void**)&h->cold_grad, sizeof(long long) * (size_t)h->mpad * (size_t)kColdCopies));
        // (one array of cold-row accumulators per XCD with L2-local atomics -- only on a device that passes the self-check of that assumption,
        //  matching_kernels.hip: cold_xcd_selftest; DUALIP_HIP_COLD_XCD=0: the shared array with device-scope atomics)
        h->cold_per_xcd = !(plan_env("DUALIP_HIP_COLD_XCD") && plan_env("DUALIP_HIP_COLD_XCD")[0] == '0') && cold_xcd_selftest(st) == 1;
    }
    if (plan_env("DUALIP_HIP_TIMELINE")) {
        CK(owned_malloc(h, (void**)&h->timeline, sizeof(unsigned long long) * kTimelineSlots * (size_t)(h->n_wg > 0 ? h->n_wg : 1)));
        CKH(hipMemsetAsync(h->timeline, 0, sizeof(unsigned long long) * kTimelineSlots * (size_t)(h->n_wg > 0 ? h->n_wg : 1), st));
    }
    int* bad_dev = nullptr;
    CKH(hipMalloc(&bad_dev, sizeof(int)));
    hipError_t e = hipMemsetAsync(bad_dev, 0, sizeof(int), st);
    if (e == hipSuccess && win_bytes > 0) e = hipMemcpyAsync(h->tiles, win_dev, win_bytes, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && tile_bytes > win_bytes)
        e = hipMemcpyAsync((char*)h->tiles + win_bytes, (const void*)words4.data(), tile_bytes - win_bytes, hipMemcpyHostToDevice, st);
    std::vector<ProjDev> pd((size_t)(n_proj > 0 ? n_proj : 1));
    for (int32_t q = 0; q < n_proj; ++q) pd[(size_t)q] = ProjDev{projs_host[q].kind, 0, projs_host[q].p0, projs_host[q].p1};
    if (e == hipSuccess) e = hipMemcpyAsync(h->projs, pd.data(), sizeof(ProjDev) * pd.size(), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) {
        (void)hipFree(bad_dev);
        matching_free(h);
        return hip_fail(e, "metadata upload");
    }
    unsigned int* row_count_dev = nullptr;
    std::vector<unsigned int> row_count_h((size_t)(m > 0 ? m : 1), 0u);
    e = hipMalloc((void**)&row_count_dev, sizeof(unsigned int) * row_count_h.size());
    if (e == hipSuccess) e = hipMemsetAsync(row_count_dev, 0, sizeof(unsigned int) * row_count_h.size(), st);
    if (e != hipSuccess) {
        (void)hipFree(bad_dev);
        matching_free(h);
        return hip_fail(e, "row histogram");
    }
    if (nnz > 0) {
        rc = row_dtype == DL_I64 ? reencode_rows<int64_t>(h, rowidx, st, bad_dev, row_count_dev)
                                 : (row_dtype == DL_U16 ? reencode_rows<uint16_t>(h, rowidx, st, bad_dev, row_count_dev) : reencode_rows<int32_t>(h, rowidx, st, bad_dev, row_count_dev));
        if (rc) {
            (void)hipFree(bad_dev);
            (void)hipFree(row_count_dev);
            matching_free(h);
            return rc;
        }
    }
    e = hipMemcpyAsync(row_count_h.data(), row_count_dev, sizeof(unsigned int) * row_count_h.size(), hipMemcpyDeviceToHost, st);
    // max |a|, max |c| (bound for the fixed-point gradient accumulation) and max |projection parameter|
    unsigned long long* mx_dev = nullptr;
    unsigned long long mx_host[3] = {0, 0, 0};
    uint8_t* sell_flags_dev = nullptr;
    if (e == hipSuccess) e = hipMalloc((void**)&mx_dev, 3 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(mx_dev, 0, 3 * sizeof(unsigned long long), st);
    if (e == hipSuccess && n > 0) {  // where the in-place reads of the caller-ordered arrays end (dl_matching_own_inputs)
        if (!pid_sell.empty()) {
            e = hipMalloc((void**)&sell_flags_dev, pid_sell.size());
            if (e == hipSuccess) e = hipMemcpyAsync(sell_flags_dev, pid_sell.data(), pid_sell.size(), hipMemcpyHostToDevice, st);
        }
        if (e == hipSuccess) {
            const int blocks = (int)std::min<int64_t>(4096, (n + 255) / 256);
            if (idx_dtype == DL_I64)
                hipLaunchKernelGGL(unsliced_end_kernel<int64_t>, dim3(blocks), dim3(256), 0, st, n, (const int64_t*)colptr, col_proj, sell_flags_dev, (int32_t)pid_sell.size(), n_proj, mx_dev + 2);
            else
                hipLaunchKernelGGL(unsliced_end_kernel<int32_t>, dim3(blocks), dim3(256), 0, st, n, (const int32_t*)colptr, col_proj, sell_flags_dev, (int32_t)pid_sell.size(), n_proj, mx_dev + 2);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipMemsetAsync(h->shift_dev, 0, 2 * sizeof(int), st);
    if (e == hipSuccess && launch_absmax(val_dtype, nnz, a, mx_dev, st)) e = hipErrorUnknown;
    if (e == hipSuccess && launch_absmax(val_dtype, nnz, c, mx_dev + 1, st)) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipMemcpyAsync(mx_host, mx_dev, sizeof(mx_host), hipMemcpyDeviceToHost, st);
    int bad = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, bad_dev, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // also keeps the host vectors alive until the uploads finished
    (void)hipFree(bad_dev);
    (void)hipFree(row_count_dev);
    if (mx_dev) (void)hipFree(mx_dev);
    if (sell_flags_dev) (void)hipFree(sell_flags_dev);
    if (e != hipSuccess) {
        matching_free(h);
        return hip_fail(e, "create sync");
    }
    h->unsliced_end = (int64_t)mx_host[2];
    memcpy(&h->amax, &mx_host[0], sizeof(double));
    memcpy(&h->cmax, &mx_host[1], sizeof(double));
    if (!std::isfinite(h->amax) || !std::isfinite(h->cmax)) {  // (absmax_kernel: any inf / NaN element surfaces here)
        matching_free(h);
        return fail(DL_E_ARG, "the values of %s hold an inf or NaN: the exact fixed-point sums of the fused pass are undefined for it (the reference would return NaN)",
                    !std::isfinite(h->amax) ? "A" : "c");
    }
    for (unsigned int v : row_count_h) h->row_count_max = v > (unsigned int)h->row_count_max ? (int64_t)v : h->row_count_max;
    if (h->m_hot > 0) {
        // renumber rows by descending non-zero count (stable): new id < m_hot <=> the row lives in LDS
        std::vector<int32_t> inv((size_t)m), perm((size_t)m);
        for (int64_t i = 0; i < m; ++i) inv[(size_t)i] = (int32_t)i;
        std::stable_sort(inv.begin(), inv.end(), [&](int32_t x, int32_t y) { return row_count_h[(size_t)x] > row_count_h[(size_t)y]; });
        for (int64_t pnew = 0; pnew < m; ++pnew) perm[(size_t)inv[(size_t)pnew]] = (int32_t)pnew;
        int32_t* perm_dev = h->row_perm;  // kept: the AGD step kernel writes the next dual vector in renumbered order with it
        e = hipMemcpyAsync(perm_dev, perm.data(), sizeof(int32_t) * (size_t)m, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h->row_inv, inv.data(), sizeof(int32_t) * (size_t)m, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            const int threads = 256;
            const int64_t b64 = (nnz + threads - 1) / threads;
            const int blocks = (int)(b64 > 8192 ? 8192 : (b64 > 0 ? b64 : 1));
            if (h->row_bytes == 2) hipLaunchKernelGGL(remap_rows_kernel<uint16_t>, dim3(blocks), dim3(threads), 0, st, nnz, (uint16_t*)h->rowidx, perm_dev);
            else hipLaunchKernelGGL(remap_rows_kernel<uint32_t>, dim3(blocks), dim3(threads), 0, st, nnz, (uint32_t*)h->rowidx, perm_dev);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // perm / inv are host temporaries
        if (e != hipSuccess) {
            matching_free(h);
            return hip_fail(e, "row renumbering");
        }
        uint64_t hot_nnz = 0;
        for (int64_t pnew = 0; pnew < h->m_hot; ++pnew) hot_nnz += row_count_h[(size_t)inv[(size_t)pnew]];
        h->hot_fraction = nnz > 0 ? (double)hot_nnz / (double)nnz : 1.0;
    }
    phase("uploads, row re-encoding, maxima");
    if (h->n_sell > 0) CK(sell_finish(h, colptr, idx_dtype, col_proj, pid_sell, sell_desc_h, st));
    phase("slice sort + transpose");
    // |x| bounds per projection kind: box -> max(|lower|, |upper|); simplex -> z (+ slack); cone / identity -> via |v| per launch
    bool used_none = false;
    std::vector<char> used((size_t)(n_proj > 0 ? n_proj : 1), 0);
    for (uint32_t pid : tile_pid4) {
        if (pid == kNoProj) used_none = true;
        else used[pid] = 1;
    }
    for (size_t q = 0; q < pid_sell.size() && q < used.size(); ++q)
        if (pid_sell[q]) used[q] = 1;
    for (size_t q = 0; q < used_dev.size(); ++q) {  // device-packed tiles
        if (!used_dev[q]) continue;
        if ((int32_t)q == n_proj) used_none = true;
        else if (q < used.size()) used[q] = 1;
    }
    h->has_unbounded = used_none;
    for (int32_t q = 0; q < n_proj; ++q) {
        if (!used[(size_t)q]) continue;
        const double p0 = projs_host[q].p0 < 0 ? -projs_host[q].p0 : projs_host[q].p0;
        const double p1 = projs_host[q].p1 < 0 ? -projs_host[q].p1 : projs_host[q].p1;
        switch (projs_host[q].kind) {
            case DL_PROJ_BOX: h->xmax_bounded = std::max(h->xmax_bounded, std::max(p0, p1)); break;
            case DL_PROJ_SIMPLEX:
            case DL_PROJ_SIMPLEX_EQ: h->xmax_bounded = std::max(h->xmax_bounded, p0 + 1e-6); break;
            case DL_PROJ_CONE_LOWER:
            case DL_PROJ_CONE_UPPER:
                h->has_unbounded = true;
                h->pmax_unbounded = std::max(h->pmax_unbounded, p0);
                break;
            default: h->has_unbounded = true; break;
        }
    }
    {   // 32-bit gradient slabs (common.h: slab32).  fp32 handles only (an fp64 handle is a parity run: its 61-bit grid stays); the whole
        // gradient in LDS (under the hot-rows plan the cold rows' global accumulators share the grid).  DUALIP_HIP_SLAB32=0: 64-bit slabs.
        const char* se = plan_env("DUALIP_HIP_SLAB32");
        // Every projection must bound x itself (box, simplex): with a one-sided operator the bound is |v|'s -- (amax lmax + cmax) / gamma -- which
        // typical elements sit orders of magnitude below, and a grid taken from it would round them away.
        const bool can = val_dtype == DL_F32 && h->grad_lds && h->m_hot == 0 && !h->has_unbounded && h->n_wg >= kSlabMinWg && h->row_count_max <= kSlabMaxRow;
        h->slab32 = can && !(se && se[0] == '0');
        if (h->slab32) {
            int rc_l1 = slab_refresh_bound(h, st);  // (also the gate on the rows' scales: may clear h->slab32)
            if (rc_l1) {
                matching_free(h);
                return rc_l1;
            }
        }
        if (h->slab32) {
            // (the slab allocation was sized for int64: its first half holds the low words, its second half the high words)
            h->slab_hi = static_cast<int32_t*>(h->partial) + (size_t)h->n_wg * (size_t)h->mpad;
            CK(owned_malloc(h, (void**)&h->slab_ovf, sizeof(unsigned long long) * ((size_t)h->n_wg + 1)));
            CKH(hipMemsetAsync(h->slab_ovf, 0, sizeof(unsigned long long) * ((size_t)h->n_wg + 1), st));
        }
    }
    // ---- balance of the window tiles (fused_common.h: Deal): rounds per workgroup, even to start with, adapted from the launches' stamps.
    //      Only where a round is a small share of a wavefront's work (>= kBalMinRounds rounds).  DUALIP_HIP_XCD_BALANCE=0: even deal. ----
    {
        const char* be = plan_env("DUALIP_HIP_XCD_BALANCE");
        const int64_t S = (int64_t)h->n_wg * kFusedWaves;
        const int64_t rw = S > 0 ? (h->n_short + S - 1) / S : 0;
        if (const char* mr = plan_env("DUALIP_HIP_XCD_BALANCE_MIN_ROUNDS")) h->bal_min_rounds = atoi(mr) > 1 ? atoi(mr) : 2;
        if (const char* gn = dev_env("DUALIP_HIP_BALANCE_GAIN")) h->bal_gain = atof(gn) > 0.0 ? atof(gn) : h->bal_gain;
        if (const char* g0 = dev_env("DUALIP_HIP_BALANCE_GAIN0")) h->bal_gain0 = atof(g0) > 0.0 ? atof(g0) : h->bal_gain0;
        if (const char* bl = dev_env("DUALIP_HIP_BALANCE_LAUNCHES")) h->bal_first = atoi(bl) > 0 ? atoi(bl) : h->bal_first;
        const bool adapt = !(be && be[0] == '0') && h->n_wg >= 2 && h->n_wg <= 1024 && rw >= h->bal_min_rounds;
        {   // (every handle has a table; only those that adapt have stamps)
            const size_t words = bal_table_words(h->n_wg > 0 ? h->n_wg : 1);
            CK(owned_malloc(h, (void**)&h->bal, sizeof(int32_t) * words));
            std::vector<int32_t> tab(words, 0);
            const int32_t n0 = adapt ? (int32_t)rw : 0x7FFFFFFF;
            tab[0] = n0;  // minimum = every workgroup's rounds: no tail tables in use
            for (int w = 0; w < h->n_wg; ++w) tab[4 + (size_t)w] = n0;
            CKH(hipMemcpyAsync(h->bal, tab.data(), sizeof(int32_t) * words, hipMemcpyHostToDevice, st));
            h->bal_adapts = adapt;
            if (adapt) {
                CK(owned_malloc(h, (void**)&h->bal_stamps, sizeof(unsigned long long) * 4 * (size_t)h->n_wg));
                CKH(hipMemsetAsync(h->bal_stamps, 0, sizeof(unsigned long long) * 4 * (size_t)h->n_wg, st));
            }
            CKH(hipStreamSynchronize(st));  // (tab is a host temporary)
            // ---- two-phase deal of the one-lane slices (fused4_kernel.h): handles whose WINDOWS do not adapt (too few rounds of them, or
            //      none: all-simplex maps) but whose slices are many rounds per wavefront.  First binary only; DUALIP_HIP_SELL_BALANCE=0: off. ----
            const char* se = plan_env("DUALIP_HIP_SELL_BALANCE");
            const int64_t n_plain = h->n_sell - h->n_sell_lane_slices;
            const int64_t rs = S > 0 ? n_plain / S : 0;
            const bool sell_adapt = !(se && se[0] == '0') && !(be && be[0] == '0') && !adapt && !h->lanes_binary && h->n_sell_lane_slices == 0 && h->n_wg >= 2 && h->n_wg <= 1024 &&
                                    h->n_wg % 2 == 0 && n_plain < (1ll << 31);
            // (DUALIP_HIP_SELL_BALANCE_PPM=<share>: tests -- a fixed second phase of that share for the even workgroups, never adapted, any size)
            const char* fe = plan_env("DUALIP_HIP_SELL_BALANCE_PPM");
            const int64_t forced = fe ? atoll(fe) : -1;
            // (same threshold as the windows' deal: at 38 rounds per wavefront -- 10M entities, all-simplex -- the second phase is two rounds of half
            //  the launch, and its quantisation cost what the balance won: 0.1811 -> 0.1825 ms per launch, three alternations)
            if (sell_adapt && (forced >= 0 || rs >= (int64_t)h->bal_min_rounds)) {
                std::vector<int32_t> sb(4 + (size_t)h->n_wg, -1);
                sb[0] = (int32_t)n_plain;  // everything in the first phase to start with
                sb[1] = 0;
                sb[2] = 0;
                sb[3] = 0;
                if (forced > 0) {
                    const int64_t tail = n_plain * std::min<int64_t>(forced, 900000) / 1000000;
                    sb[0] = (int32_t)(n_plain - tail);
                    sb[1] = (h->n_wg / 2) * kFusedWaves;
                    sb[2] = (int32_t)std::min<int64_t>(forced, 900000);
                    for (int w = 0; w < h->n_wg; w += 2) sb[4 + (size_t)w] = w / 2;
                }
                h->sell_bal_frozen = forced >= 0;
                CK(owned_malloc(h, (void**)&h->sell_bal, sizeof(int32_t) * sb.size()));
                CKH(hipMemcpyAsync(h->sell_bal, sb.data(), sizeof(int32_t) * sb.size(), hipMemcpyHostToDevice, st));
                if (!h->bal_stamps) {
                    CK(owned_malloc(h, (void**)&h->bal_stamps, sizeof(unsigned long long) * 4 * (size_t)h->n_wg));
                    CKH(hipMemsetAsync(h->bal_stamps, 0, sizeof(unsigned long long) * 4 * (size_t)h->n_wg, st));
                }
                CKH(hipStreamSynchronize(st));
            }
        }
    }
#undef CK
#undef CKH
    for (int i = 0; i < kNumPlanSwitches; ++i)  // what dl_matching_info(h, 2100) reports
        if (getenv(kPlanSwitches[i])) h->switches |= 1u << i;
    *out = h;
    return 0;
}

int dl_matching_destroy(dl_matching* h) {
    matching_free(h);
    return 0;
}

const char* dl_switch_name(int bit) { return (bit >= 0 && bit < dl::kNumPlanSwitches) ? dl::kPlanSwitches[bit] : nullptr; }

int64_t dl_matching_info(const dl_matching* h, int what) {
    if (!h) return -1;
    switch (what) {
        case 2007: return h->slab32 ? 4 : 8;  // bytes per element of the gradient slabs
        case 2008: {  // workgroups whose shares left 32 bits in the LAST fused launch (their high words travelled too); synchronous device read
            if (!h->slab32) return 0;
            std::vector<unsigned long long> ep((size_t)h->n_wg + 1);
            if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(ep.data(), h->slab_ovf, sizeof(unsigned long long) * ep.size(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
            int64_t k = 0;
            if (h->slab_epoch == 0) return 0;  // (no fused launch yet: the zeroed epochs are not overflows)
            for (int w = 0; w < h->n_wg; ++w) k += ep[(size_t)w] == h->slab_epoch ? 1 : 0;
            return k;
        }
        case 2100: return (int64_t)h->switches;
        case 2011: return h->slab32 ? h->n_wide : 0;  // 32-bit slabs: rows whose high words travel in every launch (the grid is the finest row's)
        case 2010: return h->slab_rows_ok ? 1 : 0;  // 32-bit slabs: the grid is fine enough for every row (slab_refresh_bound); 0 with slab_bytes 8 = refused for that
        case 2009: return h->cold_per_xcd ? 1 : 0;  // hot-rows plan: one cold-row accumulator array per XCD (self-checked at creation) / 0: one shared array
        case 2101:
#ifdef DL_DEVTOOLS
            return 1;  // a developer build: ablation switches are live (results may be wrong on purpose)
#else
            return 0;
#endif
        case 0: return h->n_tiles;
        case 1: return h->n_wg;
        case 2: return (int64_t)h->lds_bytes;
        case 3: return h->lam_lds ? 1 : 0;
        case 4: return h->grad_lds ? 1 : 0;
        case 5: return (int64_t)h->owned_bytes;
        case 6: return h->n_long;
        case 7: return h->row_bytes;
        case 8: return h->layout;
        case 9: return h->m_hot;
        case 10: return (int64_t)(h->hot_fraction * 1e6);
        case 11: return h->n_xlong;
        case 12: return h->n_sell;
        case 13: return h->n_sell_cols;
        case 14: return h->n_sell_elems;
        case 15: return h->n_sell_nnz;
        case 16: return h->desc_words;
        case 17: return h->n_sell_mixed_cols;
        case 2003: return h->m_hot > 0 ? h->m_lam : (h->lam_lds ? h->m : 0);  // rows of the dual vector staged in LDS
        case 2004: return (h->lanes_binary || h->n_sell_lane_slices > 0) ? 1 : 0;
        case 2001: return h->owns_inputs ? 1 : 0;
        case 2002: return h->owns_inputs ? h->own_count : h->unsliced_end;  // non-zeros read in place from the (caller's / owned) CSC-ordered arrays
        case 2000: return h->n_sell_lane_cols;
        case 2005:    // share of the one-lane slices in the second phase of their deal, ppm (synchronous read; -1: even deal only)
        case 2006: {  // updates of that deal so far
            if (!h->sell_bal) return -1;
            int32_t v[4] = {0, 0, 0, 0};
            if (hipMemcpy(v, h->sell_bal, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
            return what == 2005 ? v[2] : v[3];
        }
        default:
            if (what >= 18 && what < 18 + 1024) {  // rounds of workgroup (what - 18) in the window tiles' deal (synchronous read; -1: no table)
                if (!h->bal_stamps || what - 18 >= h->n_wg) return -1;
                int32_t v = -1;
                if (hipMemcpy(&v, h->bal + 4 + (what - 18), sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
                return v;
            }
            return -1;
    }
}

int dl_matching_calculate(dl_matching* h, const void* lambda, double gamma, double* packed_out, void* x_out, dl_stream_t stream) {
    if (!h || !packed_out || (h->m > 0 && !lambda)) return fail(DL_E_ARG, "null argument");
    if (!(gamma > 0.0) && !(gamma < 0.0)) return fail(DL_E_ARG, "gamma must be non-zero");
    return matching_calculate(h, lambda, gamma, packed_out, x_out, (hipStream_t)stream);
}

int dl_matching_profile(dl_matching* h, int enable) {
    if (!h) return fail(DL_E_ARG, "null handle");
    h->prof_on = enable != 0;
    h->prof_stride = enable > 1 ? enable : 1;
    h->prof_seen = 0;
    h->prof_used = 0;
    return 0;
}

int dl_matching_set_fairness(dl_matching* h, const void* f_values, dl_stream_t stream) {
    if (!h) return fail(DL_E_ARG, "null handle");
    if (!f_values) {
        h->fair = nullptr;
        return 0;
    }
    if (h->m < 2) return fail(DL_E_ARG, "the fairness pair needs at least its own two rows");
    // (the mirror of dl_matching_own_inputs' refusal: after it the straggler tiles read at pool offsets while f stays in the caller's order)
    if (h->owns_inputs) return fail(DL_E_STATE, "the handle owns its inputs (dl_matching_own_inputs): the fairness stream is read at the caller's offsets, which its straggler tiles no longer use -- build a new handle");
    if ((h->n_tiles > 0 || h->n_sell > 0) && (!h->lam_lds || !h->grad_lds || h->stage_a))  // (a staged handle's windows read past nnz inside ITS padded copies; f has no padding)
        return fail(DL_E_STATE, "the fairness pair needs the 256-wide tile layout with the dual vector and the gradient in LDS (16-byte aligned values, nnz >= 1024)");
    if ((reinterpret_cast<uintptr_t>(f_values) & 15u) != 0) return fail(DL_E_ARG, "fairness values must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (!h->partial_fair) {
        int rc = owned_malloc(h, (void**)&h->partial_fair, sizeof(double) * (size_t)(h->n_wg > 0 ? h->n_wg : 1));
        if (!rc) rc = owned_malloc(h, (void**)&h->dense_ax, sizeof(double) * 2);
        if (rc) return rc;
        DL_HIP(hipMemsetAsync(h->dense_ax, 0, sizeof(double) * 2, st));
    }
    {   // max |f| (bounds |v| for unbounded projections, like max |a| and max |c| taken at create time)
        unsigned long long* mx_dev = nullptr;
        unsigned long long bits = 0;
        DL_HIP(hipMalloc((void**)&mx_dev, sizeof(unsigned long long)));
        hipError_t e = hipMemsetAsync(mx_dev, 0, sizeof(unsigned long long), st);
        if (e == hipSuccess && launch_absmax(h->val_dtype, h->nnz, f_values, mx_dev, st)) e = hipErrorUnknown;
        if (e == hipSuccess) e = hipMemcpyAsync(&bits, mx_dev, sizeof(bits), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(mx_dev);
        if (e != hipSuccess) return hip_fail(e, "fairness values");
        memcpy(&h->fair_max, &bits, sizeof(double));
    }
    {
        int rc = sell_fill_fair(h, f_values, st);  // the slices read their own copy, in slice order
        if (rc) return rc;
    }
    h->fair = f_values;
    h->hot_ready = false;
    return 0;
}

// max |v| of one of the caller's value arrays into *out (host), through the handle's cached scratch word (no allocation per call)
static int refresh_absmax(dl_matching* h, const void* values, double* out, hipStream_t st) {
    if (!h->absmax_dev) {
        int rc = owned_malloc(h, (void**)&h->absmax_dev, sizeof(unsigned long long));
        if (rc) return rc;
    }
    unsigned long long bits = 0;
    hipError_t e = hipMemsetAsync(h->absmax_dev, 0, sizeof(unsigned long long), st);
    if (e == hipSuccess && launch_absmax(h->val_dtype, h->nnz, values, h->absmax_dev, st)) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipMemcpyAsync(&bits, h->absmax_dev, sizeof(bits), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // (the bound travels to the kernel as an argument: the host needs it)
    if (e != hipSuccess) return hip_fail(e, "value refresh");
    double v;
    memcpy(&v, &bits, sizeof(double));
    if (!std::isfinite(v)) return fail(DL_E_ARG, "the rewritten values hold an inf or NaN: the exact fixed-point sums of the fused pass are undefined for it");
    *out = v;
    return 0;
}

int dl_matching_own_inputs(dl_matching* h, dl_stream_t stream) {
    if (!h) return fail(DL_E_ARG, "null handle");
    if (h->owns_inputs) return 0;
    if (h->fair) return fail(DL_E_STATE, "a handle with the fairness stream borrows three arrays; release is not offered for it");
    hipStream_t st = (hipStream_t)stream;
    const size_t vs = h->val_dtype == DL_F32 ? 4 : 8;
    if (h->stage_a) {
        // A staged handle (unaligned or tiny inputs) already reads ONLY memory it owns: the aligned, zero-padded copies stage_a / stage_c and its
        // re-encoded rows, all nnz_arr elements long -- which its windows' vector loads may touch up to their padded end.  Cutting new, shorter
        // arrays out of them (as below) would put those loads past the new arrays' end; there is nothing to release but the link to the caller's
        // arrays, so that is all that happens: a / c keep pointing at the staged copies, which are counted in owned_bytes since creation.
        h->a_src = nullptr;
        h->c_src = nullptr;
        h->own_count = h->nnz_arr;
        h->owns_inputs = true;
        return 0;
    }
    // the prefix window tiles (and the single-column tiles of entries without slices) read in place: [0, unsliced_end) -- plus one
    // window's width, because a window's loads cover 256 slots from its start (clamped to the arrays' end); never less than the 256
    // slots at index 0 that the padding descriptors' unconditional prefetches touch
    int64_t keep = h->unsliced_end + 256;
    keep = (keep + 255) / 256 * 256;
    if (keep > h->nnz) keep = h->nnz;
    // stragglers: single-column tiles (of sliced entries) that reach beyond the prefix -> a pool behind it; their descriptors' read
    // offset (words 0 / 1) moves, the primal's position (words 4 / 5) stays
    const int64_t n_single = h->n_tiles - h->n_short;  // one-wavefront and whole-workgroup tiles
    uint32_t* long_dev = reinterpret_cast<uint32_t*>(h->tiles) + (size_t)h->n_short * (size_t)h->desc_words + 12;
    std::vector<uint32_t> lw((size_t)n_single * 12);
    if (n_single > 0) {
        DL_HIP(hipMemcpyAsync(lw.data(), long_dev, lw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        DL_HIP(hipStreamSynchronize(st));
    }
    std::vector<uint64_t> src, dst, len;
    int64_t total = keep;
    for (int64_t t = 0; t < n_single; ++t) {
        uint32_t* w = &lw[(size_t)t * 12];
        const uint64_t k0 = (((uint64_t)w[1] << 32) | w[0]) & ((1ull << 40) - 1);
        const uint64_t L = ((uint64_t)w[3] << 32) | w[2];
        if (k0 + L <= (uint64_t)keep) continue;
        src.push_back(k0);
        dst.push_back((uint64_t)total);
        len.push_back(L);
        w[0] = (uint32_t)((uint64_t)total & 0xFFFFFFFFu);
        w[1] = (w[1] & ~0xFFu) | (uint32_t)(((uint64_t)total >> 32) & 0xFFu);
        total += ((int64_t)L + 3) / 4 * 4;
    }
    void *na = nullptr, *nc = nullptr, *nr = nullptr;
    uint64_t* lists = nullptr;
    hipError_t e = hipMalloc(&na, (size_t)total * vs);
    if (e == hipSuccess) e = hipMalloc(&nc, (size_t)total * vs);
    if (e == hipSuccess) e = hipMalloc(&nr, (size_t)total * (size_t)h->row_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(na, h->a, (size_t)keep * vs, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(nc, h->c, (size_t)keep * vs, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(nr, h->rowidx, (size_t)keep * (size_t)h->row_bytes, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && !src.empty()) {
        const size_t ns = src.size();
        e = hipMalloc((void**)&lists, 3 * ns * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMemcpyAsync(lists, src.data(), ns * sizeof(uint64_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(lists + ns, dst.data(), ns * sizeof(uint64_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(lists + 2 * ns, len.data(), ns * sizeof(uint64_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            if (vs == 4 && h->row_bytes == 2) hipLaunchKernelGGL((pool_copy_kernel<float, uint16_t>), dim3((unsigned)ns), dim3(256), 0, st, lists, lists + ns, lists + 2 * ns, (const float*)h->a, (const float*)h->c, (const uint16_t*)h->rowidx, (float*)na, (float*)nc, (uint16_t*)nr);
            else if (vs == 4) hipLaunchKernelGGL((pool_copy_kernel<float, uint32_t>), dim3((unsigned)ns), dim3(256), 0, st, lists, lists + ns, lists + 2 * ns, (const float*)h->a, (const float*)h->c, (const uint32_t*)h->rowidx, (float*)na, (float*)nc, (uint32_t*)nr);
            else if (h->row_bytes == 2) hipLaunchKernelGGL((pool_copy_kernel<double, uint16_t>), dim3((unsigned)ns), dim3(256), 0, st, lists, lists + ns, lists + 2 * ns, (const double*)h->a, (const double*)h->c, (const uint16_t*)h->rowidx, (double*)na, (double*)nc, (uint16_t*)nr);
            else hipLaunchKernelGGL((pool_copy_kernel<double, uint32_t>), dim3((unsigned)ns), dim3(256), 0, st, lists, lists + ns, lists + 2 * ns, (const double*)h->a, (const double*)h->c, (const uint32_t*)h->rowidx, (double*)na, (double*)nc, (uint32_t*)nr);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(long_dev, lw.data(), lw.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // (lw / src / dst / len are host objects)
    if (lists) (void)hipFree(lists);
    if (e != hipSuccess) {
        if (na) (void)hipFree(na);
        if (nc) (void)hipFree(nc);
        if (nr) (void)hipFree(nr);
        return hip_fail(e, "dl_matching_own_inputs");
    }
    (void)hipFree(h->rowidx);  // the re-encoded rows of the sliced columns live in the slices' transposed copy
    h->owned_bytes -= (size_t)h->nnz * (size_t)h->row_bytes;
    h->rowidx = nr;
    h->own_a = na;
    h->own_c = nc;
    h->a = na;
    h->c = nc;
    h->own_count = total;
    h->owned_bytes += (size_t)total * (2 * vs + (size_t)h->row_bytes);
    h->owns_inputs = true;
    return 0;
}

int dl_matching_update_costs(dl_matching* h, dl_stream_t stream) {
    if (!h) return fail(DL_E_ARG, "null handle");
    if (h->owns_inputs) return fail(DL_E_STATE, "the handle owns its inputs (dl_matching_own_inputs): it no longer sees the caller's arrays -- build a new handle for new values");
    hipStream_t st = (hipStream_t)stream;
    if (h->stage_c && h->nnz > 0) DL_HIP(hipMemcpyAsync(h->stage_c, h->c_src, (size_t)h->nnz * (h->val_dtype == DL_F32 ? 4 : 8), hipMemcpyDeviceToDevice, st));  // (staged handle: its copy follows the caller's array)
    if (h->nnz > 0) {  // max |c| scales the fixed-point sums c.x / sum x^2 (fused_common.h: scalar_shift) and bounds |v| for projections that do not bound x
        int rc = refresh_absmax(h, h->c, &h->cmax, st);
        if (rc) return rc;
    }
    return sell_refill_costs(h, st);
}

int dl_matching_update_values(dl_matching* h, dl_stream_t stream) {
    if (!h) return fail(DL_E_ARG, "null handle");
    if (h->owns_inputs) return fail(DL_E_STATE, "the handle owns its inputs (dl_matching_own_inputs): it no longer sees the caller's arrays -- build a new handle for new values");
    hipStream_t st = (hipStream_t)stream;
    if (h->stage_a && h->nnz > 0) {  // (staged handle: its copies follow the caller's arrays)
        const size_t vs_st = h->val_dtype == DL_F32 ? 4 : 8;
        DL_HIP(hipMemcpyAsync(h->stage_a, h->a_src, (size_t)h->nnz * vs_st, hipMemcpyDeviceToDevice, st));
        DL_HIP(hipMemcpyAsync(h->stage_c, h->c_src, (size_t)h->nnz * vs_st, hipMemcpyDeviceToDevice, st));
    }
    if (h->nnz > 0) {  // max |a| scales the fixed-point gradient; max |c| as in dl_matching_update_costs
        int rc = refresh_absmax(h, h->a, &h->amax, st);
        if (!rc) rc = refresh_absmax(h, h->c, &h->cmax, st);
        if (!rc && h->slab32) rc = slab_refresh_bound(h, st);  // (the grid of the 32-bit slabs follows the rows' L1 norms)
        if (rc) return rc;
    }
    int rc = sell_refill_values(h, st);
    if (rc) return rc;
    return sell_refill_costs(h, st);
}

int dl_matching_set_eq_padding(dl_matching* h, const int32_t* heights_host, int32_t n_rows, dl_stream_t stream) {
    if (!h) return fail(DL_E_ARG, "null handle");
    if (!heights_host) {  // back to the exact projection
        if (h->eq_heights) {
            DL_HIP(hipStreamSynchronize((hipStream_t)stream));
            (void)hipFree(h->eq_heights);
            h->eq_heights = nullptr;
        }
        return 0;
    }
    if (n_rows != h->n_proj) return fail(DL_E_ARG, "heights must have one row of 32 per projection entry (%d), got %d", (int)h->n_proj, (int)n_rows);
    for (int64_t k = 0; k < (int64_t)n_rows * 32; ++k)
        if (heights_host[k] < 0) return fail(DL_E_ARG, "negative block height");
    const size_t bytes = sizeof(int32_t) * 32 * (size_t)(n_rows > 0 ? n_rows : 1);
    if (!h->eq_heights) DL_HIP(hipMalloc((void**)&h->eq_heights, bytes));
    DL_HIP(hipMemcpyAsync(h->eq_heights, heights_host, sizeof(int32_t) * 32 * (size_t)n_rows, hipMemcpyHostToDevice, (hipStream_t)stream));
    DL_HIP(hipStreamSynchronize((hipStream_t)stream));  // heights_host may be a temporary
    return 0;
}

int dl_matching_timeline_read(dl_matching* h, uint64_t* out_host, int64_t capacity) {
    if (!h || !out_host) return fail(DL_E_ARG, "null argument");
    if (!h->timeline) return fail(DL_E_ARG, "timeline not enabled (DUALIP_HIP_TIMELINE=1 at create)");
    const int64_t n = (int64_t)h->n_wg * kTimelineSlots;
    if (capacity < n) return fail(DL_E_ARG, "timeline buffer too small");
    DL_HIP(hipDeviceSynchronize());
    DL_HIP(hipMemcpy(out_host, h->timeline, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

int dl_matching_profile_read(dl_matching* h, double* total_ms_host, int64_t* launches_host) {
    if (!h || !total_ms_host || !launches_host) return fail(DL_E_ARG, "null argument");
    double total = 0.0;
    for (size_t i = 0; i < h->prof_used; ++i) {
        DL_HIP(hipEventSynchronize(h->prof_stop[i]));
        float ms = 0.f;
        DL_HIP(hipEventElapsedTime(&ms, h->prof_start[i], h->prof_stop[i]));
        total += (double)ms;
    }
    *total_ms_host = total;
    *launches_host = (int64_t)h->prof_used;
    return 0;
}

int dl_dual_epilogue(int64_t m, int val_dtype, const double* packed, const void* b, const void* lambda, double gamma, void* grad_out,
                     double* scal_out, dl_stream_t stream) {
    if (m < 0 || !packed || !scal_out || (m > 0 && (!b || !lambda || !grad_out))) return fail(DL_E_ARG, "null argument");
    if (val_dtype != DL_F32 && val_dtype != DL_F64) return fail(DL_E_ARG, "bad val_dtype");
    return launch_epilogue(m, val_dtype, packed, b, lambda, gamma, grad_out, scal_out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------
static void agd_free(dl_agd* s) {
    if (!s) return;
    void* ptrs[] = {s->x, s->x_alt, s->y, s->y_old, s->g, s->g_old, s->beta, s->log, s->state, s->packed, s->partial_stats, s->chk_partial, s->packed_blk[0], s->packed_blk[1], s->packed_blk[2]};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete s;
}

int dl_agd_create(dl_agd** out, int64_t m, int val_dtype, int64_t max_iter, const float* beta_seq_host, double initial_step_size,
                  double max_step_size, const uint8_t* eq_mask, const void* lambda0, dl_stream_t stream) {
    if (!out) return fail(DL_E_ARG, "out is null");
    *out = nullptr;
    if (m < 0 || max_iter < 0 || (max_iter > 0 && !beta_seq_host) || (m > 0 && !lambda0)) return fail(DL_E_ARG, "bad argument");
    if (val_dtype != DL_F32 && val_dtype != DL_F64) return fail(DL_E_ARG, "bad val_dtype");
    hipStream_t st = (hipStream_t)stream;
    dl_agd* s = new (std::nothrow) dl_agd();
    if (!s) return fail(DL_E_NOMEM, "out of host memory");
    static std::atomic<uint64_t> next_uid{1};
    s->uid = next_uid.fetch_add(1);
    s->m = m;
    s->max_iter = max_iter;
    s->val_dtype = val_dtype;
    s->eq_mask = eq_mask;
    const size_t vs = val_dtype == DL_F32 ? 4 : 8;
    const size_t vb = (size_t)(m > 0 ? m : 1) * vs;
    hipError_t e = hipSuccess;
    void** vecs[] = {&s->x, &s->x_alt, &s->y, &s->y_old, &s->g, &s->g_old};
    for (void** v : vecs)
        if (e == hipSuccess) e = hipMalloc(v, vb);
    if (e == hipSuccess) e = hipMalloc((void**)&s->beta, sizeof(float) * (size_t)(max_iter > 0 ? max_iter : 1));
    if (e == hipSuccess) e = hipMalloc((void**)&s->log, sizeof(double) * kLogCols * (size_t)(max_iter > 0 ? max_iter : 1));
    if (e == hipSuccess) e = hipMalloc(&s->state, agd_state_bytes());
    if (e == hipSuccess) e = hipMalloc((void**)&s->packed, sizeof(double) * (size_t)(m + 2));
    if (e == hipSuccess) e = hipMalloc((void**)&s->partial_stats, agd_partial_stats_bytes(m));
    if (e == hipSuccess) e = hipMalloc((void**)&s->chk_partial, sizeof(unsigned long long) * (size_t)((m + 63) / 64 + 2));
    if (e == hipSuccess && m > 0) e = hipMemcpyAsync(s->x, lambda0, vb, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && m > 0) e = hipMemcpyAsync(s->y, lambda0, vb, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(s->y_old, 0, vb, st);
    if (e == hipSuccess) e = hipMemsetAsync(s->g, 0, vb, st);
    if (e == hipSuccess) e = hipMemsetAsync(s->g_old, 0, vb, st);
    if (e == hipSuccess) e = hipMemsetAsync(s->log, 0, sizeof(double) * kLogCols * (size_t)(max_iter > 0 ? max_iter : 1), st);
    if (e == hipSuccess && max_iter > 0) e = hipMemcpyAsync(s->beta, beta_seq_host, sizeof(float) * (size_t)max_iter, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) {
        agd_free(s);
        return hip_fail(e, "dl_agd_create");
    }
    int rc = agd_state_init(s->state, initial_step_size, max_step_size, st);  // synchronises (covers beta_seq_host too)
    if (rc) {
        agd_free(s);
        return rc;
    }
    *out = s;
    return 0;
}

int dl_agd_destroy(dl_agd* s) {
    agd_free(s);
    return 0;
}

const void* dl_agd_x(const dl_agd* s) { return s ? s->x : nullptr; }
const void* dl_agd_y(const dl_agd* s) { return s ? s->y : nullptr; }
const void* dl_agd_grad(const dl_agd* s) { return s ? s->g : nullptr; }

int dl_agd_get(const dl_agd* s, int which, void* dst, dl_stream_t stream) {
    if (!s || which < 0 || which > 2 || (s->m > 0 && !dst)) return fail(DL_E_ARG, "bad argument");
    const void* src = which == 0 ? s->x : (which == 1 ? s->y : s->g);
    const size_t vs = s->val_dtype == DL_F32 ? 4 : 8;
    if (s->m > 0) DL_HIP(hipMemcpyAsync(dst, src, vs * (size_t)s->m, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int dl_agd_step(dl_agd* s, const double* packed, const void* b, double gamma, int64_t iter, int decay_now, double decay_factor,
                dl_stream_t stream) {
    if (!s || !packed || (s->m > 0 && !b)) return fail(DL_E_ARG, "null argument");
    if (iter < 1 || iter > s->max_iter) return fail(DL_E_STATE, "iteration %lld outside 1..max_iter=%lld", (long long)iter, (long long)s->max_iter);
    StepSource src;
    src.packed[0] = packed;
    src.n_packed = 1;
    return launch_agd_step(s, src, b, gamma, iter, decay_now, decay_factor, (hipStream_t)stream);
}

int dl_agd_run_matching(dl_agd* s, dl_matching* f, const void* b, int64_t first_iter, int64_t n_iters, double* gamma_io_host,
                        int64_t gamma_decay_steps, double decay_factor, void* x_out, dl_stream_t stream) {
    if (!s || !f || !gamma_io_host || (s->m > 0 && !b)) return fail(DL_E_ARG, "null argument");
    if (s->m != f->m || s->val_dtype != f->val_dtype) return fail(DL_E_STATE, "optimizer state and objective disagree on m / dtype");
    if (first_iter < 1 || n_iters < 0 || first_iter + n_iters - 1 > s->max_iter) return fail(DL_E_STATE, "iteration range outside 1..max_iter");
    hipStream_t st = (hipStream_t)stream;
    double gamma = *gamma_io_host;
    if (first_iter == 1 || f->hot_ready_owner != s->uid) f->hot_ready = false;  // a new run (or another optimiser) starts from its own dual vector
    // An iteration is: fused pass at x -> stats (g, partial reductions) -> apply (step, new x and y).  With DUALIP_HIP_FUSE_APPLY=1
    // (and a handle that allows it) the apply of iteration i rides the fused launch of iteration i + 1 (agd_step.h): two launches
    // per iteration instead of three; the last iteration of the call is applied by its own launch, so the state is complete when
    // the call returns.  Opt-in: measured neutral (fused_common.h: fused_prologue).
    const bool can_fuse = matching_can_fuse_apply(f);
    PendingStep pending;
    for (int64_t it = first_iter; it < first_iter + n_iters; ++it) {
        void* xo = (it == first_iter + n_iters - 1) ? x_out : nullptr;
        const bool empty = (f->n_tiles == 0 && f->n_sell == 0) || f->n_wg == 0;
        int rc = 0;
        if (empty) {
            rc = matching_calculate(f, s->x, gamma, s->packed, xo, st);
        } else if (pending.valid) {
            rc = matching_launch_fused(f, nullptr, gamma, xo, st, s->uid, s, &pending);  // applies iteration it - 1, then runs at the new x
            if (!rc) agd_rotate(s);
            pending.valid = false;
        } else {
            rc = matching_launch_fused(f, s->x, gamma, xo, st, s->uid);
        }
        if (rc) return rc;
        const int decay_now = gamma_decay_steps > 0 && (it % gamma_decay_steps == 0);
        StepSource src;
        if (empty) {
            src.packed[0] = s->packed;
            src.n_packed = 1;
        } else {
            src.slabs = f;
            src.hot = f;
        }
        rc = launch_agd_stats(s, src, b, st);
        if (rc) return rc;
        pending.valid = true;
        pending.gamma = gamma;
        pending.iter = it;
        pending.decay_now = decay_now;
        pending.decay_factor = decay_factor;
        pending.scal = agd_step_scal(s, src);
        if (!can_fuse || empty || it == first_iter + n_iters - 1) {
            rc = launch_agd_apply(s, src, pending, st);
            if (rc) return rc;
            pending.valid = false;
        }
        if (decay_now) gamma = gamma * decay_factor;  // agd.py:105
    }
    *gamma_io_host = gamma;
    return 0;
}

// The column-sharded loop: what dl_agd_run_matching does on one device, for this rank's block(s) of columns, with the ONE
// exchange of an iteration inside (comm.h).  Per iteration and rank:
//   P2P   fused pass -> slab reduction that stores its sums into every rank's mailbox -> step kernels (the first waits for all
//         ranks' slots and adds them in rank order) : 4 launches, no collective launch
//   RCCL  fused pass -> slab reduction -> ncclAllReduce (m + 2 doubles) -> step kernels
// A shard split into n_blocks handles runs the blocks back to back; with RCCL the collective of every block but the last
// goes to the communicator's side stream, overlapping the next block's fused pass (the all-reduce is linear: the step adds
// the blocks' all-reduced buffers); with P2P the blocks accumulate locally and the last reduction pushes.
int dl_agd_run_matching_sharded(dl_agd* s, dl_matching* const* blocks, int32_t n_blocks, dl_comm* comm, const void* b, int64_t first_iter,
                                int64_t n_iters, double* gamma_io_host, int64_t gamma_decay_steps, double decay_factor, dl_stream_t stream) {
    if (!s || !blocks || n_blocks < 1 || n_blocks > 4 || !comm || !gamma_io_host || (s->m > 0 && !b)) return fail(DL_E_ARG, "bad argument");
    for (int k = 0; k < n_blocks; ++k) {
        if (!blocks[k]) return fail(DL_E_ARG, "null block handle");
        if (s->m != blocks[k]->m || s->val_dtype != blocks[k]->val_dtype) return fail(DL_E_STATE, "optimizer state and objective disagree on m / dtype");
    }
    if (comm->max_count < s->m + 2) return fail(DL_E_STATE, "communicator capacity %lld is below m + 2", (long long)comm->max_count);
    if (comm->backend == DL_COMM_P2P && !comm->connected) return fail(DL_E_STATE, "P2P communicator is not connected (dl_comm_p2p_connect)");
    if (first_iter < 1 || n_iters < 0 || first_iter + n_iters - 1 > s->max_iter) return fail(DL_E_STATE, "iteration range outside 1..max_iter");
    hipStream_t st = (hipStream_t)stream;
    const size_t pbytes = sizeof(double) * (size_t)(s->m + 2);
    for (int k = 1; k < n_blocks && comm->backend == DL_COMM_RCCL; ++k)
        if (!s->packed_blk[k - 1]) DL_HIP(hipMalloc((void**)&s->packed_blk[k - 1], pbytes));
    double gamma = *gamma_io_host;
    // (as in dl_agd_run_matching: the apply of iteration i rides the first block's fused launch of iteration i + 1 when it can)
    const bool first_empty = (blocks[0]->n_tiles == 0 && blocks[0]->n_sell == 0) || blocks[0]->n_wg == 0;
    const bool can_fuse = !first_empty && matching_can_fuse_apply(blocks[0]);
    PendingStep pending;
    for (int64_t it = first_iter; it < first_iter + n_iters; ++it) {
        StepSource src;
        src.scale = comm->emu_scale;
        MailArgs mail;
        hipEvent_t ev_stop = nullptr;
        for (int k = 0; k < n_blocks; ++k) {
            dl_matching* f = blocks[k];
            const bool last = k == n_blocks - 1;
            const bool empty = (f->n_tiles == 0 && f->n_sell == 0) || f->n_wg == 0;
            double* pk = k == 0 ? s->packed : s->packed_blk[k - 1];
            int rc = 0;
            if (k == 0 && pending.valid) {
                rc = matching_launch_fused(f, nullptr, gamma, nullptr, st, 0, s, &pending);  // applies iteration it - 1, then runs at the new x
                if (!rc) agd_rotate(s);
                pending.valid = false;
            } else if (!empty) {
                rc = matching_launch_fused(f, s->x, gamma, nullptr, st, 0);
            }
            if (rc) return rc;
            if (last && comm->prof_on && (comm->prof_seen++ % (uint64_t)comm->prof_stride) == 0) {  // measurement: end of the last fused pass -> end of the step's first kernel
                if (comm->prof_used == comm->prof_start.size() && comm->prof_start.size() < 16384) {
                    hipEvent_t e0, e1;
                    DL_HIP(hipEventCreate(&e0));
                    DL_HIP(hipEventCreate(&e1));
                    comm->prof_start.push_back(e0);
                    comm->prof_stop.push_back(e1);
                }
                if (comm->prof_used < comm->prof_start.size()) {
                    DL_HIP(hipEventRecord(comm->prof_start[comm->prof_used], st));
                    ev_stop = comm->prof_stop[comm->prof_used];
                    comm->prof_used += 1;
                }
            }
            if (comm->backend == DL_COMM_P2P) {
                if (empty) {  // nothing to add: an all-zero partial (accumulated blocks stay as they are)
                    if (k == 0) DL_HIP(hipMemsetAsync(s->packed, 0, pbytes, st));
                    if (last) {  // a rank without non-zeros still takes part in the exchange: the others wait for its slot
                        const unsigned long long seq = ++comm->seq;
                        const PushArgs push = comm_push_args(comm, seq);
                        mail = comm_mail_args(comm, seq);
                        rc = comm_push_buffer(s->packed, s->m + 2, push, st);
                        src.mail = &mail;
                    }
                } else if (!last) {
                    rc = matching_reduce(f, s->packed, k == 0 ? 0 : 1, nullptr, st, 0);
                } else {
                    const unsigned long long seq = ++comm->seq;
                    const PushArgs push = comm_push_args(comm, seq);
                    mail = comm_mail_args(comm, seq);
                    rc = matching_reduce(f, s->packed, 2, &push, st, k > 0 ? 1 : 0);
                    src.mail = &mail;
                }
                if (rc) return rc;
            } else {
                if (empty) DL_HIP(hipMemsetAsync(pk, 0, pbytes, st));
                else rc = matching_reduce(f, pk, 0, nullptr, st, 0);
                if (rc) return rc;
                if (n_blocks == 1) {
                    rc = comm_rccl_allreduce(comm, pk, s->m + 2, st);
                } else {  // split shard: every collective on the side stream, in block order (one communicator, one queue);
                          // all but the last overlap the next block's fused pass
                    DL_HIP(hipEventRecord(comm->ev_ready[k], st));
                    DL_HIP(hipStreamWaitEvent(comm->side, comm->ev_ready[k], 0));
                    rc = comm_rccl_allreduce(comm, pk, s->m + 2, comm->side);
                    if (!rc && last) {
                        DL_HIP(hipEventRecord(comm->ev_done, comm->side));
                        DL_HIP(hipStreamWaitEvent(st, comm->ev_done, 0));
                    }
                }
                if (rc) return rc;
                src.packed[k] = pk;
                src.n_packed = k + 1;
            }
        }
        const int decay_now = gamma_decay_steps > 0 && (it % gamma_decay_steps == 0);
        int rc = launch_agd_stats(s, src, b, st);
        if (rc) return rc;
        if (ev_stop) DL_HIP(hipEventRecord(ev_stop, st));
        pending.valid = true;
        pending.gamma = gamma;
        pending.iter = it;
        pending.decay_now = decay_now;
        pending.decay_factor = decay_factor;
        pending.scal = agd_step_scal(s, src);
        if (!can_fuse || it == first_iter + n_iters - 1) {
            rc = launch_agd_apply(s, src, pending, st);
            if (rc) return rc;
            pending.valid = false;
        }
        if (decay_now) gamma = gamma * decay_factor;  // agd.py:105
    }
    *gamma_io_host = gamma;
    return 0;
}

int dl_agd_read_log(dl_agd* s, int64_t first, int64_t count, double* rows_host, dl_stream_t stream) {
    if (!s || !rows_host || first < 0 || count < 0 || first + count > s->max_iter) return fail(DL_E_ARG, "bad log range");
    hipStream_t st = (hipStream_t)stream;
    if (count > 0) DL_HIP(hipMemcpyAsync(rows_host, s->log + first * kLogCols, sizeof(double) * kLogCols * (size_t)count, hipMemcpyDeviceToHost, st));
    DL_HIP(hipStreamSynchronize(st));
    return 0;
}

int dl_agd_read_max_step(dl_agd* s, double* out_host, dl_stream_t stream) {
    if (!s || !out_host) return fail(DL_E_ARG, "null argument");
    return agd_state_read_max_step(s->state, s->state_cur, out_host, (hipStream_t)stream);
}

int dl_project_dense(int64_t L, int64_t K, int val_dtype, const void* in, void* out, const dl_proj_desc* proj_host, dl_stream_t stream) {
    if (L < 0 || K < 0 || !proj_host || ((L * K) > 0 && (!in || !out))) return fail(DL_E_ARG, "bad argument");
    if (val_dtype != DL_F32 && val_dtype != DL_F64) return fail(DL_E_ARG, "bad val_dtype");
    const int k = proj_host->kind;
    if (k < DL_PROJ_NONE || k > DL_PROJ_SIMPLEX_EQ) return fail(DL_E_PROJ, "Unknown projection operator kind %d", k);
    if ((k == DL_PROJ_SIMPLEX || k == DL_PROJ_SIMPLEX_EQ) && !(proj_host->p0 > 0.0)) return fail(DL_E_PROJ, "Simplex radius z must be positive.");
    return launch_project_dense(L, K, val_dtype, in, out, proj_host, (hipStream_t)stream);
}

int dl_jacobi_precondition(int64_t m, int64_t nnz, const void* rowidx, int idx_dtype, void* a, void* b, void* row_norms_out, int val_dtype,
                           dl_stream_t stream) {
    if (m < 0 || nnz < 0 || (nnz > 0 && (!rowidx || !a)) || (m > 0 && (!b || !row_norms_out))) return fail(DL_E_ARG, "bad argument");
    if (idx_dtype != DL_I32 && idx_dtype != DL_I64) return fail(DL_E_ARG, "bad idx_dtype");
    if (val_dtype != DL_F32 && val_dtype != DL_F64) return fail(DL_E_ARG, "bad val_dtype");
    return launch_jacobi(m, nnz, rowidx, idx_dtype, a, b, row_norms_out, val_dtype, (hipStream_t)stream);
}

}  // extern "C"
