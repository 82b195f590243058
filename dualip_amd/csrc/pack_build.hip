// pack_build.hip -- the window tiles of the 256-wide layout packed ON THE DEVICE (dl_matching_create).
//
// The packing rule is the greedy one of api.hip:pack_tiles4 -- a 16-byte aligned window of <= 256 non-zeros holding whole
// consecutive columns of one projection entry (point-wise entries: simply the next 256 non-zeros of the entry's run of columns,
// pid_flat); columns that cannot sit in a window become single-column tiles; columns that live in column-per-lane slices
// (sell.h) are skipped -- but the column range is cut into fixed chunks that are packed
// independently, one thread per chunk (a window never spans a chunk boundary: +0.5 % windows at the benchmark's shape).  Two
// passes over the column pointers (count, exclusive scan, write) replace what used to be the dominant part of handle creation at
// 100M entities: 0.8 GB of column pointers / projection ids copied to the host (340 ms) and a single-threaded host loop over
// them (243 ms + 61 ms of scheduling).
#include <hipcub/hipcub.hpp>

#include <vector>

#include "common.h"

namespace dl {

constexpr int kPackChunk = 8192;   // columns per chunk (= per thread)
constexpr int kPackSellMaxLen = 24;        // sell.h: kSellMaxH
constexpr int kPackSellMaxLenLanes = 512;  // sell.h: kSellMaxLenLanes (entries sliced with K lanes per column: flag bit 3)

struct PackErr {
    int bad_colptr;      // a column pointer decreases
    int bad_proj;        // a column refers to a projection entry that was not given
    long long where;     // first offending column (smallest index seen)
};

// One chunk of columns.  WRITE = false: count only.
template <class IdxT, bool WRITE>
__device__ __forceinline__ void pack_chunk(int compact, int64_t j0, int64_t j1, int64_t nnz, const IdxT* __restrict__ colptr, const int32_t* __restrict__ col_proj, int32_t n_proj,
                                           const uint8_t* __restrict__ pid_sell, uint32_t& n_win, uint32_t& n_long, uint32_t* __restrict__ win_out,
                                           uint32_t* __restrict__ long_out, uint8_t* __restrict__ used, PackErr* __restrict__ err) {
    const uint64_t nnz_al4 = (uint64_t)nnz & ~3ull;
    bool open = false;
    uint64_t W = 0, H0 = 0, H1 = 0, H2 = 0, H3 = 0;
    uint32_t lo = 0, end = 0, cur_proj = kNoProj;
    auto set_head = [&](uint32_t e) {
        const uint64_t bit = 1ull << (e >> 2);
        switch (e & 3u) {
            case 0: H0 |= bit; break;
            case 1: H1 |= bit; break;
            case 2: H2 |= bit; break;
            default: H3 |= bit; break;
        }
    };
    auto emit12 = [&](uint32_t* dst, uint64_t w0, uint64_t h0, uint64_t h1, uint64_t h2, uint64_t h3, uint32_t pid) {
        dst[0] = (uint32_t)w0;
        dst[1] = (uint32_t)(w0 >> 32);
        dst[2] = (uint32_t)h0;
        dst[3] = (uint32_t)(h0 >> 32);
        dst[4] = (uint32_t)h1;
        dst[5] = (uint32_t)(h1 >> 32);
        dst[6] = (uint32_t)h2;
        dst[7] = (uint32_t)(h2 >> 32);
        dst[8] = (uint32_t)h3;
        dst[9] = (uint32_t)(h3 >> 32);
        dst[10] = pid == kNoProj ? 0xFFFFFFFFu : pid;
        dst[11] = 0u;
    };
    auto flush = [&]() {
        if (!open) return;
        if (end < 256) set_head(end);  // sentinel: elements past the last column form their own dummy segment
        if constexpr (WRITE) {
            const uint64_t w0 = W | ((uint64_t)end << 40) | ((uint64_t)lo << 49);
            if (compact) {  // (every window point-wise: no head masks; the projection id rides in the top 12 bits)
                win_out[(size_t)n_win * 2] = (uint32_t)w0;
                win_out[(size_t)n_win * 2 + 1] = (uint32_t)(w0 >> 32) | ((cur_proj == kNoProj ? 0xFFFu : cur_proj) << 20);
            } else {
                emit12(win_out + (size_t)n_win * 12, w0, H0, H1, H2, H3, cur_proj);
            }
        }
        n_win += 1;
        open = false;
        H0 = H1 = H2 = H3 = 0;
    };
    int64_t k0 = (int64_t)colptr[j0];
    for (int64_t j = j0; j < j1; ++j) {
        const int64_t k1 = (int64_t)colptr[j + 1];
        const int64_t len = k1 - k0;
        const int64_t kc = k0;
        k0 = k1;
        if (len < 0) {
            if constexpr (!WRITE) {
                err->bad_colptr = 1;
                atomicMin(&err->where, (long long)j);
            }
            continue;
        }
        if (len == 0) continue;
        const int32_t pid = col_proj ? col_proj[j] : (n_proj > 0 ? 0 : -1);
        if (pid >= n_proj) {
            if constexpr (!WRITE) {
                err->bad_proj = 1;
                atomicMin(&err->where, (long long)j);
            }
            continue;
        }
        const uint32_t pj = pid < 0 ? kNoProj : (uint32_t)pid;
        const uint8_t fl = pid_sell[pj == kNoProj ? 255u : (pj < 255u ? pj : 254u)];  // bit 0: sliced entry, bit 1: point-wise entry (flat windows), bit 2: cut at multiples of 256, bit 3: slices hold columns of up to 255
        const bool sliced = pj != kNoProj && pj < 255u && (fl & 1u);
        const bool flat = (pj == kNoProj || pj < 254u) && (fl & 2u);
        const bool flat_align = (fl & 4u) != 0;
        if (sliced && len <= ((fl & 8u) ? kPackSellMaxLenLanes : kPackSellMaxLen)) {
            flush();  // (a window holds consecutive columns only)
            continue;
        }
        if constexpr (!WRITE) used[pj == kNoProj ? (uint32_t)n_proj : pj] = 1;
        const bool tail_quad = (uint64_t)k1 > nnz_al4;  // touches the array's last partial quad: no vector loads there
        if ((len > 253 && !flat) || tail_quad || sliced || (pj != kNoProj && pj >= (uint32_t)kProjLdsSlots - 1)) {
            flush();
            if constexpr (WRITE) emit12(long_out + (size_t)n_long * 12, (uint64_t)kc | (1ull << 51), (uint64_t)len, 0, 0, 0, pj);
            n_long += 1;
            continue;
        }
        if (flat) {
            // point-wise entry: the projection does not see column boundaries, so windows are cut every 256 non-zeros wherever
            // they fall -- no window re-reads the tail of its predecessor, and a long column is just more of the stream
            if (open && pj != cur_proj) flush();
            uint64_t k = (uint64_t)kc;
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
            W = (uint64_t)kc & ~3ull;
            lo = (uint32_t)((uint64_t)kc - W);
            cur_proj = pj;
            set_head(0);
        }
        set_head((uint32_t)((uint64_t)kc - W));
        end = (uint32_t)((uint64_t)k1 - W);
    }
    flush();
}

template <class IdxT>
__global__ __launch_bounds__(64) void pack_count_kernel(int compact, int64_t n, int64_t nnz, const IdxT* __restrict__ colptr, const int32_t* __restrict__ col_proj, int32_t n_proj,
                                                        const uint8_t* __restrict__ pid_sell, unsigned long long* __restrict__ counts, uint8_t* __restrict__ used,
                                                        PackErr* __restrict__ err) {
    const int64_t ch = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j0 = ch * kPackChunk;
    if (j0 >= n) return;
    const int64_t j1 = j0 + kPackChunk < n ? j0 + kPackChunk : n;
    uint32_t nw = 0, nl = 0;
    pack_chunk<IdxT, false>(compact, j0, j1, nnz, colptr, col_proj, n_proj, pid_sell, nw, nl, nullptr, nullptr, used, err);
    counts[ch] = (unsigned long long)nw | ((unsigned long long)nl << 32);  // both counters ride one exclusive scan
}

template <class IdxT>
__global__ __launch_bounds__(64) void pack_write_kernel(int compact, int64_t n, int64_t nnz, const IdxT* __restrict__ colptr, const int32_t* __restrict__ col_proj, int32_t n_proj,
                                                        const uint8_t* __restrict__ pid_sell, const unsigned long long* __restrict__ offsets, uint32_t* __restrict__ win_out,
                                                        uint32_t* __restrict__ long_out) {
    const int64_t ch = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j0 = ch * kPackChunk;
    if (j0 >= n) return;
    const int64_t j1 = j0 + kPackChunk < n ? j0 + kPackChunk : n;
    uint32_t nw = (uint32_t)offsets[ch], nl = (uint32_t)(offsets[ch] >> 32);
    pack_chunk<IdxT, true>(compact, j0, j1, nnz, colptr, col_proj, n_proj, pid_sell, nw, nl, win_out, long_out, nullptr, nullptr);
}

template <class IdxT>
static int pack_device_typed(int64_t n, int64_t nnz, const IdxT* colptr, const int32_t* col_proj, int32_t n_proj, const std::vector<uint8_t>& pid_sell_h,
                             const std::vector<uint8_t>& pid_flat_h, int compact, uint32_t** win_dev_out, int64_t* n_win_out, std::vector<uint32_t>& long_words, std::vector<uint8_t>& used_h, hipStream_t st) {
    *win_dev_out = nullptr;
    *n_win_out = 0;
    long_words.clear();
    used_h.assign((size_t)n_proj + 1, 0);
    if (n <= 0) return 0;
    const int64_t n_chunks = (n + kPackChunk - 1) / kPackChunk;
    unsigned long long *counts = nullptr, *offsets = nullptr;
    uint8_t *flags = nullptr, *used = nullptr;
    PackErr* err = nullptr;
    void* tmp = nullptr;
    uint32_t *win = nullptr, *lng = nullptr;
    auto cleanup = [&](bool keep_win) {
        for (void* p : {(void*)counts, (void*)offsets, (void*)flags, (void*)used, (void*)err, tmp, (void*)lng})
            if (p) (void)hipFree(p);
        if (!keep_win && win) (void)hipFree(win);
    };
    std::vector<uint8_t> flags_h(256, 0);
    for (size_t q = 0; q < pid_sell_h.size() && q < 255; ++q) flags_h[q] = pid_sell_h[q] ? (pid_sell_h[q] == 2 ? 9 : 1) : 0;
    for (size_t q = 0; q < pid_flat_h.size() && q < 254; ++q) flags_h[q] |= pid_flat_h[q] ? (pid_flat_h[q] == 2 ? 6 : 2) : 0;
    if (!pid_flat_h.empty() && pid_flat_h.back()) flags_h[255] = pid_flat_h.back() == 2 ? 6 : 2;  // last element: columns with no projection entry
    PackErr err_h = {0, 0, (long long)n};
    hipError_t e = hipMalloc((void**)&counts, sizeof(unsigned long long) * (size_t)(n_chunks + 1));
    if (e == hipSuccess) e = hipMalloc((void**)&offsets, sizeof(unsigned long long) * (size_t)(n_chunks + 1));
    if (e == hipSuccess) e = hipMalloc((void**)&flags, 256);
    if (e == hipSuccess) e = hipMalloc((void**)&used, (size_t)n_proj + 1);
    if (e == hipSuccess) e = hipMalloc((void**)&err, sizeof(PackErr));
    if (e == hipSuccess) e = hipMemcpyAsync(flags, flags_h.data(), 256, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(used, 0, (size_t)n_proj + 1, st);
    if (e == hipSuccess) e = hipMemcpyAsync(err, &err_h, sizeof(PackErr), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(counts, 0, sizeof(unsigned long long) * (size_t)(n_chunks + 1), st);
    const unsigned blocks = (unsigned)((n_chunks + 63) / 64);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pack_count_kernel<IdxT>, dim3(blocks), dim3(64), 0, st, compact, n, nnz, colptr, col_proj, n_proj, flags, counts, used, err);
        e = hipGetLastError();
    }
    size_t tmp_bytes = 0;
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, counts, offsets, (int)(n_chunks + 1), st);
    if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16);
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, counts, offsets, (int)(n_chunks + 1), st);
    unsigned long long total = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&total, offsets + n_chunks, sizeof(total), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&err_h, err, sizeof(PackErr), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(used_h.data(), used, (size_t)n_proj + 1, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        cleanup(false);
        return hip_fail(e, "window packing (count)");
    }
    if (err_h.bad_colptr) {
        cleanup(false);
        return fail(DL_E_LAYOUT, "ccol_indices is not monotone at column %lld", err_h.where);
    }
    if (err_h.bad_proj) {
        cleanup(false);
        return fail(DL_E_PROJ, "column %lld refers to a projection entry beyond the %d given", err_h.where, (int)n_proj);
    }
    const uint64_t n_win = total & 0xFFFFFFFFull, n_long = total >> 32;
    if (n_win >= (1ull << 31) || n_long >= (1ull << 31)) {  // (also catches a carry of the packed low counter)
        cleanup(false);
        return fail(DL_E_ARG, "too many tiles");
    }
    e = hipMalloc((void**)&win, sizeof(uint32_t) * (compact ? 2 : 12) * (size_t)(n_win ? n_win : 1));
    if (e == hipSuccess) e = hipMalloc((void**)&lng, sizeof(uint32_t) * 12 * (size_t)(n_long ? n_long : 1));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pack_write_kernel<IdxT>, dim3(blocks), dim3(64), 0, st, compact, n, nnz, colptr, col_proj, n_proj, flags, offsets, win, lng);
        e = hipGetLastError();
    }
    long_words.resize((size_t)n_long * 12);
    if (e == hipSuccess && n_long) e = hipMemcpyAsync(long_words.data(), lng, sizeof(uint32_t) * 12 * (size_t)n_long, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        cleanup(false);
        return hip_fail(e, "window packing (write)");
    }
    cleanup(true);
    *win_dev_out = win;
    *n_win_out = (int64_t)n_win;
    return 0;
}

// Window descriptors stay on the device (*win_dev_out, 12 dwords each -- 2 when `compact`, which the caller may ask for when every
// entry that can have windows is point-wise -- memory order; the caller frees it); the single-column
// tiles come back to the host (they are few, and the caller orders them longest first).  pid_flat[q] != 0: entry q is point-wise
// (its last element, index n_proj: columns with no entry).  used[q] != 0: entry q has a window or
// single-column tile (used[n_proj]: a column with no entry has one).
int pack_device(int64_t n, int64_t nnz, const void* colptr, int idx_dtype, const int32_t* col_proj, int32_t n_proj, const std::vector<uint8_t>& pid_sell,
                const std::vector<uint8_t>& pid_flat, int compact, uint32_t** win_dev_out, int64_t* n_win_out, std::vector<uint32_t>& long_words, std::vector<uint8_t>& used, hipStream_t st) {
    if (idx_dtype == DL_I64) return pack_device_typed<int64_t>(n, nnz, (const int64_t*)colptr, col_proj, n_proj, pid_sell, pid_flat, compact, win_dev_out, n_win_out, long_words, used, st);
    return pack_device_typed<int32_t>(n, nnz, (const int32_t*)colptr, col_proj, n_proj, pid_sell, pid_flat, compact, win_dev_out, n_win_out, long_words, used, st);
}

}  // namespace dl
