// matching_kernels.hip -- everything around the fused dual-gradient pass of the matching LP for gfx950 that is not the pass itself: the slab
// reduction (reduce_partials_kernel: exact integer sums of the workgroups' gradient slabs; MODE 2 pushes them into the ranks' mailboxes), the
// one-off measurements of handle creation (max |v|, row L1 norms, the self-check of the per-XCD accumulators), the balance kernels of the two
// deals, and the launch glue (fused_typed).  The pass is matching_fused_kernel4 (fused4_kernel.h, four translation units); the 64-wide
// kernel that used to live here served unaligned and tiny inputs until round 5 -- they are staged into the 256-wide layout now (api.hip).
#include <cmath>
#include <atomic>

#include "comm.h"
#include "fused_common.h"

namespace dl {

// ---------------------------------------------------------------------------------------------------------
// slab reduction: packed[0..m) = 2^-shift * sum_w partial[w][i] (exact integer sum), packed[m], packed[m+1] = scalars
// ---------------------------------------------------------------------------------------------------------
constexpr int kRedThreads = 1024;
constexpr int kRedRows = 64;  // rows per block; 16 slab-slices per block

// Hot-rows plan (inv != null): slab column p belongs to the caller's row inv[p]; columns >= m_hot hold nothing -- their sums
// are in `cold` (one int64 per row, global atomics).
// MODE 0: packed[i] = sum.  MODE 1: packed[i] += sum (second and later blocks of a split shard).  MODE 2: the sum (plus
// packed[i] when `accumulate`) goes to this rank's slot in EVERY rank's mailbox (comm.h: the P2P exchange) -- the slab
// reduction is the pushing launch, no separate collective.
template <int MODE>
__global__ __launch_bounds__(kRedThreads) void reduce_partials_kernel(const long long* __restrict__ partial, const long long* __restrict__ partial_scal,
                                                                      const int* __restrict__ shift_in, int n_slabs, int n_scal, int64_t m, int64_t mpad,
                                                                      double* __restrict__ packed, const int32_t* __restrict__ inv, int64_t m_hot,
                                                                      const long long* __restrict__ cold, const double* __restrict__ dense, PushArgs push,
                                                                      int accumulate, int slab32, const int32_t* __restrict__ slab_hi,
                                                                      const unsigned long long* __restrict__ slab_ovf, unsigned long long slab_epoch,
                                                                      const uint8_t* __restrict__ slab_wide) {
    unsigned long long pushed_h = 0ull;  // hashes of what this thread pushed (comm.h: the payload checksum the flag will carry)
    auto emit = [&](int64_t i, double v) {
        if constexpr (MODE == 0) packed[i] = v;
        else if constexpr (MODE == 1) packed[i] += v;
        else push_value(push, i, accumulate ? packed[i] + v : v, pushed_h);
    };
    __shared__ long long shi[kRedThreads];
    __shared__ double sh[kRedThreads / 32];
    const int tid = threadIdx.x;
    if (blockIdx.x + 1 < gridDim.x) {  // row blocks; the LAST block only sums the scalar partials (runs beside them)
        const int rl = tid & (kRedRows - 1);
        const int ws = tid / kRedRows;
        const int64_t row = (int64_t)blockIdx.x * kRedRows + rl;
        const int64_t rc = row < m ? row : (m > 0 ? m - 1 : 0);
        long long acc = 0;
        const bool in_slabs = !inv || rc < m_hot;
        if (!in_slabs && ws == 0) {  // (one copy per XCD: dl::kColdCopies)
#pragma unroll
            for (int k = 0; k < kColdCopies; ++k) acc += cold[(int64_t)k * mpad + rc];
        }
        // latency bound: eight slabs are in flight before the first is added (slabs past the end re-read the last one)
        constexpr int kU = 8, kStride = kRedThreads / kRedRows;
        auto sum_slabs_into = [&](auto* slabs, long long& into) {  // (int32 slabs of handles with slab32, common.h; else int64)
            for (int w0 = ws; in_slabs && w0 < n_slabs; w0 += kStride * kU) {
                long long v[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int w = w0 + kStride * u;
                    v[u] = (long long)slabs[(int64_t)(w < n_slabs ? w : n_slabs - 1) * mpad + rc];
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) into += (w0 + kStride * u < n_slabs) ? v[u] : 0ll;
            }
        };
        auto sum_slabs = [&](auto* slabs) { sum_slabs_into(slabs, acc); };
        if (slab32) {
            const bool is_wide = in_slabs && slab_wide && slab_wide[rc];  // (requested first: agd_kernels.hip, the statistics kernel)
            const int32_t* lo32 = reinterpret_cast<const int32_t*>(partial);
            for (int w0 = ws; in_slabs && w0 < n_slabs; w0 += kStride * kU) {
                long long v[kU], hw[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int w = w0 + kStride * u;
                    v[u] = (long long)lo32[(int64_t)(w < n_slabs ? w : n_slabs - 1) * mpad + rc];
                }
                if (is_wide) {
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int w = w0 + kStride * u;
                        hw[u] = (long long)slab_hi[(int64_t)(w < n_slabs ? w : n_slabs - 1) * mpad + rc];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < kU; ++u) hw[u] = 0ll;
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) acc += (w0 + kStride * u < n_slabs) ? v[u] + hw[u] * 4294967296ll : 0ll;
            }
            if (!is_wide && in_slabs && slab_ovf[n_slabs] == slab_epoch) {  // (uniform, rare) some workgroup's shares left 32 bits in this launch: their high words
                for (int w0 = ws; w0 < n_slabs; w0 += kStride)
                    if (slab_ovf[w0] == slab_epoch) acc += (long long)slab_hi[(int64_t)w0 * mpad + rc] * 4294967296ll;
            }
        } else {
            sum_slabs(partial);
        }
        shi[tid] = acc;
        __syncthreads();
        if (ws == 0 && row < m) {
            long long t = shi[rl];
            for (int q = 1; q < kRedThreads / kRedRows; ++q) t += shi[q * kRedRows + rl];
            const int64_t orow = inv ? (int64_t)inv[row] : row;
            emit(orow, (dense && orow >= m - 2) ? dense[orow - (m - 2)] : ldexp((double)t, -(*shift_in)));  // (fairness pair: the two dense rows)
        }
        if constexpr (MODE == 2) push_finish(push, pushed_h);
        return;
    }
    {   // exact integer sums of the workgroups' fixed-point partials, scaled once
        long long o = 0, q = 0;
        for (int w = tid; w < n_scal; w += kRedThreads) {
            o += partial_scal[2 * w];
            q += partial_scal[2 * w + 1];
        }
        o = wave_allreduce(o, OpAdd());
        q = wave_allreduce(q, OpAdd());
        long long* shl = reinterpret_cast<long long*>(sh);
        if ((tid & 63) == 0) {
            shl[2 * (tid >> 6)] = o;
            shl[2 * (tid >> 6) + 1] = q;
        }
        __syncthreads();
        if (tid == 0) {
            long long oo = 0, qq = 0;
            for (int w = 0; w < kRedThreads / 64; ++w) {
                oo += shl[2 * w];
                qq += shl[2 * w + 1];
            }
            emit(m, ldexp((double)oo, -shift_in[1]));
            emit(m + 1, ldexp((double)qq, -shift_in[1]));
        }
        if constexpr (MODE == 2) push_finish(push, pushed_h);
    }
}

// max |v| over an array (one-off, at handle creation): non-negative floats order like their bit patterns -- and +inf, then NaN, order
// ABOVE every finite value, so a non-finite element anywhere comes back as a non-finite maximum (the handle refuses such arrays: the
// fixed-point sums of the fused pass are undefined for them, where the reference would return NaN)
template <class T>
__global__ void absmax_kernel(int64_t n, const T* __restrict__ v, unsigned long long* __restrict__ out_bits) {
    long long mx = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const long long a = __double_as_longlong(fabs((double)v[k])) & 0x7FFFFFFFFFFFFFFFll;
        mx = a > mx ? a : mx;
    }
    mx = wave_allreduce(mx, OpMax());
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, (unsigned long long)mx);
}

int launch_absmax(int val_dtype, int64_t n, const void* v, unsigned long long* out_bits, hipStream_t st) {
    if (n <= 0) return 0;
    const int threads = 256;
    int64_t b64 = (n + threads - 1) / threads;
    const int blocks = (int)(b64 > 4096 ? 4096 : b64);
    if (val_dtype == DL_F32) hipLaunchKernelGGL(absmax_kernel<float>, dim3(blocks), dim3(threads), 0, st, n, (const float*)v, out_bits);
    else hipLaunchKernelGGL(absmax_kernel<double>, dim3(blocks), dim3(threads), 0, st, n, (const double*)v, out_bits);
    DL_HIP(hipGetLastError());
    return 0;
}

// ---- self-check of the per-XCD cold-row accumulators (hot-rows plan, fused_common.h) ----
// They rest on behaviour outside the HIP / LLVM memory model: WORKGROUP-scope atomics of DIFFERENT workgroups on one address are atomic when
// all of them run on the XCD whose L2 executes them, and HW_REG_XCC_ID & 7 names that XCD.  True of this part in SPX mode; a partition mode
// with another XCD numbering, tgsplit, or a future part could break it silently.  So, once per device and process, before the first
// handle relies on it: 2048 workgroups hammer 64 words of "their" XCD's copy with workgroup-scope increments, the eight copies are added
// up, and every word must hold exactly the number of increments issued.  A lost update anywhere -> the plan falls back to ONE shared
// array with device-scope atomics (what DUALIP_HIP_COLD_XCD=0 selects).  ~60 us, exact integers.
constexpr int kXcdTestBlocks = 2048, kXcdTestThreads = 256, kXcdTestReps = 32, kXcdTestWords = 64;
__global__ void cold_xcd_selftest_kernel(unsigned long long* __restrict__ copies) {
    const unsigned int xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((32 - 1) << 11)) & (unsigned int)(kColdCopies - 1);
    typedef __attribute__((address_space(1))) unsigned long long glb_u64;
    glb_u64* p = (glb_u64*)(copies + (size_t)xcc * kXcdTestWords + (threadIdx.x & (kXcdTestWords - 1)));
    for (int r = 0; r < kXcdTestReps; ++r) (void)__hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// 1: the per-XCD accumulators are safe on the current device; 0: they lost updates (or the check could not run)
int cold_xcd_selftest(hipStream_t st) {
    static std::atomic<int> verdict[64];  // per device ordinal: 0 unknown, 1 safe, 2 unsafe
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const int slot = dev & 63;
    const int known = verdict[slot].load(std::memory_order_relaxed);
    if (known) return known == 1;
    unsigned long long* copies = nullptr;
    const size_t words = (size_t)kColdCopies * kXcdTestWords;
    if (hipMalloc((void**)&copies, sizeof(unsigned long long) * words) != hipSuccess) return 0;
    std::vector<unsigned long long> host(words, 0ull);
    hipError_t e = hipMemsetAsync(copies, 0, sizeof(unsigned long long) * words, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(cold_xcd_selftest_kernel, dim3(kXcdTestBlocks), dim3(kXcdTestThreads), 0, st, copies);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host.data(), copies, sizeof(unsigned long long) * words, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(copies);
    bool ok = e == hipSuccess;
    const unsigned long long want = (unsigned long long)kXcdTestBlocks * (kXcdTestThreads / kXcdTestWords) * kXcdTestReps;
    for (int w = 0; ok && w < kXcdTestWords; ++w) {
        unsigned long long sum = 0ull;
        for (int k = 0; k < kColdCopies; ++k) sum += host[(size_t)k * kXcdTestWords + w];
        ok = sum == want;
    }
    verdict[slot].store(ok ? 1 : 2, std::memory_order_relaxed);
    return ok ? 1 : 0;
}

// Per-row statistics of the fp32 values `a` (one-off, for handles with 32-bit slabs -- common.h: slab32 -- whose fixed-point grid and list of
// WIDE rows are decided from them, api.hip: slab_refresh_bound): L1 norm, number of non-zero values, largest |a|.  Estimates feeding bounds, so
// float sums are plenty; accumulated per workgroup in LDS (rows <= kRowL1Max: every handle whose whole gradient fits the fused kernel's LDS),
// flushed with float atomics (floats >= 0 order like their bit patterns: the maximum is an integer atomicMax).
constexpr int kRowL1Max = 16384;
template <class RowT, int WHAT>  // WHAT: 0 sum |a|, 1 count of non-zero values, 2 max |a|
__global__ __launch_bounds__(1024) void row_l1_kernel(int64_t nnz, const RowT* __restrict__ rows, const float* __restrict__ a, int m, float* __restrict__ sums) {
    __shared__ float acc[kRowL1Max];
    for (int i = threadIdx.x; i < m; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride) {
        const float v = fabsf(a[k]);
        if (WHAT == 2) atomicMax(reinterpret_cast<unsigned int*>(&acc[(uint32_t)rows[k]]), __float_as_uint(v));
        else atomicAdd(&acc[(uint32_t)rows[k]], WHAT == 1 ? (v != 0.f ? 1.f : 0.f) : v);  // (counts: exact in float up to 2^24 per row; slab32 handles have <= 65 536)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        if (acc[i] == 0.f) continue;
        if (WHAT == 2) atomicMax(reinterpret_cast<unsigned int*>(&sums[i]), __float_as_uint(acc[i]));
        else atomicAdd(&sums[i], acc[i]);
    }
}
// out_host[0 .. m) = row L1 norms, [m .. 2m) = counts of non-zero values, [2m .. 3m) = row maxima of |a| (rows: the handle's re-encoded indices).
// *measured = 0 when the rows do not fit the LDS table (nothing written).
int launch_row_stats(int64_t nnz, const void* rows, int row_bytes, const float* a, int64_t m, float* out_host, int* measured, hipStream_t st) {
    *measured = 0;
    if (m <= 0 || m > kRowL1Max || nnz <= 0) return 0;
    float* dev = nullptr;
    DL_HIP(hipMalloc((void**)&dev, sizeof(float) * 3 * (size_t)m));
    hipError_t e = hipMemsetAsync(dev, 0, sizeof(float) * 3 * (size_t)m, st);
    if (e == hipSuccess) {
        const int64_t b64 = (nnz + 1023) / 1024;
        const int blocks = (int)(b64 > 256 ? 256 : b64);
        if (row_bytes == 2) {
            hipLaunchKernelGGL((row_l1_kernel<uint16_t, 0>), dim3(blocks), dim3(1024), 0, st, nnz, (const uint16_t*)rows, a, (int)m, dev);
            hipLaunchKernelGGL((row_l1_kernel<uint16_t, 1>), dim3(blocks), dim3(1024), 0, st, nnz, (const uint16_t*)rows, a, (int)m, dev + m);
            hipLaunchKernelGGL((row_l1_kernel<uint16_t, 2>), dim3(blocks), dim3(1024), 0, st, nnz, (const uint16_t*)rows, a, (int)m, dev + 2 * m);
        } else {
            hipLaunchKernelGGL((row_l1_kernel<uint32_t, 0>), dim3(blocks), dim3(1024), 0, st, nnz, (const uint32_t*)rows, a, (int)m, dev);
            hipLaunchKernelGGL((row_l1_kernel<uint32_t, 1>), dim3(blocks), dim3(1024), 0, st, nnz, (const uint32_t*)rows, a, (int)m, dev + m);
            hipLaunchKernelGGL((row_l1_kernel<uint32_t, 2>), dim3(blocks), dim3(1024), 0, st, nnz, (const uint32_t*)rows, a, (int)m, dev + 2 * m);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, dev, sizeof(float) * 3 * (size_t)m, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dev);
    if (e != hipSuccess) return hip_fail(e, "row statistics");
    *measured = 1;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
size_t fused_lds_bytes2(int64_t rows_grad, int64_t rows_lam, int val_dtype) {
    const size_t vs = val_dtype == DL_F32 ? 4 : 8;
    size_t off = (size_t)rows_grad * 8 + (size_t)rows_lam * vs;
    off = (off + 15) / 16 * 16;
    off += (size_t)kProjLds * (val_dtype == DL_F32 ? sizeof(ProjT<float>) : sizeof(ProjT<double>));
    off += kLdsScratch;
    return off;
}
size_t fused_lds_bytes(int64_t m, int val_dtype, bool lam, bool grad) {
    const size_t vs = val_dtype == DL_F32 ? 4 : 8;
    size_t off = (grad ? (size_t)m * 8 : 0) + (lam ? (size_t)m * vs : 0);
    off = (off + 15) / 16 * 16;
    off += (size_t)kProjLds * (val_dtype == DL_F32 ? sizeof(ProjT<float>) : sizeof(ProjT<double>));
    off += kLdsScratch;
    return off;
}

int launch_fused4_f32(const dl_matching* h, const FusedArgs<float>& args, hipStream_t st);   // matching_kernels4.hip
int launch_fused4_f64(const dl_matching* h, const FusedArgs<double>& args, hipStream_t st);
int launch_fused4_f32_lanes(const dl_matching* h, const FusedArgs<float>& args, hipStream_t st);   // matching_kernels4_lanes.hip: + the K-lanes-per-column slices
int launch_fused4_f64_lanes(const dl_matching* h, const FusedArgs<double>& args, hipStream_t st);
// the second binary: handles with K-lane slices, or with >= 1 % of their non-zeros in single-column tiles (which it walks as
// one-column slices, sell.h) -- the benchmark's shapes have neither
// (decided once per handle, at creation -- api.hip -- together with the order of the single-column tiles that goes with each binary)
static bool wants_lanes_binary(const dl_matching* h) { return h->lanes_binary || h->n_sell_lane_slices > 0; }
static int launch_fused4(const dl_matching* h, const FusedArgs<float>& args, hipStream_t st) {
    return wants_lanes_binary(h) ? launch_fused4_f32_lanes(h, args, st) : launch_fused4_f32(h, args, st);
}
static int launch_fused4(const dl_matching* h, const FusedArgs<double>& args, hipStream_t st) {
    return wants_lanes_binary(h) ? launch_fused4_f64_lanes(h, args, st) : launch_fused4_f64(h, args, st);
}

template <class T>
__global__ void permute_vector_kernel(int64_t m, const T* __restrict__ src, const int32_t* __restrict__ inv, T* __restrict__ dst) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < m) dst[p] = src[inv[p]];
}

// fairness pair: (A x) of the two dense rows from the workgroups' partial sums (fixed order)
// Balance of the window tiles (fused_common.h: Deal).  st[w][0..2] = wall clock of workgroup w after its prologue, after wavefront
// 0's window tiles, and when all its wavefronts have walked everything.  The knob is the number of window rounds n_w of each
// workgroup, the target equal FINISH times (half the wavefronts walk their slices first, so the end of wavefront 0's windows is not
// the end of the workgroup): n_w <- n_w - gain (D_w - mean D) / tau, D_w = the workgroup's finish time, tau = the measured time
// of one window round; at most +-15 % / +-kBalTail/2 rounds from the even share; rounded down, the missing rounds go to the
// workgroups with the largest remainders; then the offsets and ranks of the rounds above the minimum are tabulated.
// One workgroup of 1024 threads, thread w = workgroup w (n_wg <= 1024).
__global__ __launch_bounds__(1024) void wg_balance_kernel(int32_t* __restrict__ tab, const unsigned long long* __restrict__ st, int n_wg, uint32_t n_win, int min_rounds,
                                                          double gain) {
    __shared__ double red[16];
    __shared__ long long redi[16];
    __shared__ double frac_s[1024];
    __shared__ int flag_wave[16];
    __shared__ int bcast[4];
    const int w = threadIdx.x, lane = w & 63, wave = w >> 6;
    const bool live = w < n_wg;
    auto block_sum = [&](double x) -> double {
        x = wave_allreduce(x, OpAdd());
        __syncthreads();
        if (lane == 0) red[wave] = x;
        __syncthreads();
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += red[q];
        return t;
    };
    auto block_sum_i = [&](long long x) -> long long {
        for (int o = 32; o >= 1; o >>= 1) {
            const int lo = __builtin_amdgcn_ds_bpermute((lane ^ o) << 2, (int)(x & 0xFFFFFFFFll)), hi = __builtin_amdgcn_ds_bpermute((lane ^ o) << 2, (int)(x >> 32));
            x += ((long long)hi << 32) | (unsigned int)lo;
        }
        __syncthreads();
        if (lane == 0) redi[wave] = x;
        __syncthreads();
        long long t = 0;
        for (int q = 0; q < 16; ++q) t += redi[q];
        return t;
    };
    int32_t* n = tab + 4;
    const double need = ceil((double)n_win / (double)kFusedWaves);  // rounds, summed over the workgroups
    const double even = need / (double)n_wg;
    if (even < (double)min_rounds - 1.0) return;  // (uniform: every thread)
    double dall = 0.0, dwin = 0.0;
    int n_old = 0;
    bool ok = true;
    if (live) {
        const unsigned long long a = st[4 * (size_t)w], b = st[4 * (size_t)w + 1], c = st[4 * (size_t)w + 2];
        n_old = n[w];
        ok = b > a && c > a && n_old > 0;
        dwin = ok ? (double)(b - a) : 0.0;
        dall = ok ? (double)(c - a) : 0.0;
    }
    const double bad = block_sum(ok ? 0.0 : 1.0);
    if (bad > 0.0) return;  // (a launch without window tiles in some workgroup, or no stamps: keep the table)
    const double dmean = block_sum(dall) / (double)n_wg;
    const double tau = block_sum(live ? dwin / (double)n_old : 0.0) / (double)n_wg;
    if (!(tau > 0.0)) return;
    double t = 0.0;
    if (live) {
        t = (double)n_old - gain * (dall - dmean) / tau;
        const double lo = fmax(0.85 * even, even - 0.5 * kBalTail + 2.0), hi = fmin(1.15 * even, even + 0.5 * kBalTail - 2.0);
        t = t < lo ? lo : (t > hi ? hi : t);
        t = t < 1.0 ? 1.0 : t;
    }
    int fl = live ? (int)t : 0;
    long long have = block_sum_i(fl);
    long long deficit = (long long)need - have;
    if (deficit > 0) {  // every tile must have a slot: the missing rounds go to the largest remainders
        const int all = (int)(deficit / n_wg);
        fl += live ? all : 0;
        deficit -= (long long)all * n_wg;
        frac_s[w] = live ? t - floor(t) : -1.0;
        __syncthreads();
        if (live && deficit > 0) {
            int larger = 0;
            const double mine = frac_s[w];
            for (int q = 0; q < n_wg; ++q) larger += (frac_s[q] > mine || (frac_s[q] == mine && q < w)) ? 1 : 0;
            fl += larger < (int)deficit ? 1 : 0;
        }
    }
    // minimum / maximum of the new rounds
    int mn = live ? fl : 0x7FFFFFFF, mx = live ? fl : 0;
    for (int o = 32; o >= 1; o >>= 1) {
        const int a = __builtin_amdgcn_ds_bpermute((lane ^ o) << 2, mn), b = __builtin_amdgcn_ds_bpermute((lane ^ o) << 2, mx);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    __syncthreads();
    if (lane == 0) {
        flag_wave[wave] = mn;
        redi[wave] = mx;
    }
    __syncthreads();
    if (w == 0) {
        int a = 0x7FFFFFFF, b = 0;
        for (int q = 0; q < 16; ++q) {
            a = flag_wave[q] < a ? flag_wave[q] : a;
            b = (int)redi[q] > b ? (int)redi[q] : b;
        }
        bcast[0] = a;
        bcast[1] = b;
    }
    __syncthreads();
    const int n_min = bcast[0], J = bcast[1] - bcast[0];
    if (J > kBalTail || n_min < 1) return;  // (cannot happen: the clamps bound the range; keep the old table)
    if (live) n[w] = fl;
    // tables of the rounds above the minimum: wavefront q tabulates rounds q, q + 16, ... on its own (no workgroup barriers: with
    // them this kernel took ~50 us, 0.5 % of the launches it follows)
    __shared__ int fl_s[1024];
    fl_s[w] = live ? fl : 0;
    __syncthreads();
    int32_t* off = tab + 4 + n_wg;
    int32_t* rank = off + kBalTail;
    for (int j = wave; j < J; j += 16) {
        const int k = n_min + j;
        int running = 0;
        long long sum_min = 0;
        for (int base = 0; base < n_wg; base += 64) {
            const int ww = base + lane;
            const int f = ww < n_wg ? fl_s[ww] : 0;
            const bool in = ww < n_wg && f > k;
            const unsigned long long bal = __ballot(in);
            if (ww < n_wg) rank[(size_t)j * n_wg + ww] = running + __popcll(bal & ((1ull << lane) - 1ull));
            running += __popcll(bal);
            sum_min += ww < n_wg ? (f < k ? f : k) : 0;
        }
        for (int o = 32; o >= 1; o >>= 1) {
            const int lo = __builtin_amdgcn_ds_bpermute((lane ^ o) << 2, (int)(sum_min & 0xFFFFFFFFll)), hi = __builtin_amdgcn_ds_bpermute((lane ^ o) << 2, (int)(sum_min >> 32));
            sum_min += ((long long)hi << 32) | (unsigned int)lo;
        }
        if (lane == 0) off[j] = (int32_t)(sum_min * kFusedWaves);
    }
    if (w == 0) {
        tab[0] = n_min;
        tab[1] = J;
    }
}

// Two-phase deal of the one-lane slices (fused4_kernel.h): tab = { n1, wavefronts of phase 2, share in ppm, updates ; rank[n_wg] }.
// st as above (stamp 0: prologue done, stamp 2: all done).  With the share phi of the slices in the second phase and half of the
// workgroups its members, workgroup w's part of the work is (1 - phi) / n [+ phi / (n / 2)]; its SLOWNESS s_w = time / work.  The new
// members are the half with the smallest slowness, and the share that would make both halves finish together is
// phi* = (s_slow - s_fast) / (s_slow + s_fast) (means of the halves): phi moves half of the way there.  At the fixed point both halves
// finish together whatever a slice of the tail costs relative to the rest (the tail holds the tallest slices).  One workgroup.
__global__ __launch_bounds__(1024) void sell_balance_kernel(int32_t* __restrict__ tab, const unsigned long long* __restrict__ st, int n_wg, uint32_t n_sell, double gain) {
    __shared__ double red[16];
    __shared__ double slow_s[1024];
    const int w = threadIdx.x, lane = w & 63, wave = w >> 6;
    const bool live = w < n_wg;
    auto block_sum = [&](double x) -> double {
        x = wave_allreduce(x, OpAdd());
        __syncthreads();
        if (lane == 0) red[wave] = x;
        __syncthreads();
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += red[q];
        return t;
    };
    const uint32_t n1_old = (uint32_t)tab[0];
    const int members_old = tab[1] / kFusedWaves;
    const double phi_old = (members_old > 0 && n1_old < n_sell) ? (double)(n_sell - n1_old) / (double)n_sell : 0.0;
    double d = 0.0;
    bool ok = true, member = false;
    if (live) {
        const unsigned long long a = st[4 * (size_t)w], c = st[4 * (size_t)w + 2];
        ok = c > a;
        d = ok ? (double)(c - a) : 0.0;
        member = tab[4 + w] >= 0;
    }
    if (block_sum(ok ? 0.0 : 1.0) > 0.0) return;  // (no stamps from this launch: keep the deal)
    const int half = n_wg / 2;
    const double work = (1.0 - phi_old) / (double)n_wg + ((member && phi_old > 0.0) ? phi_old / (double)half : 0.0);
    const double slow = live ? d / work : 1e300;
    slow_s[w] = slow;
    __syncthreads();
    // The members are chosen ONCE, from the first stamped launch (an even deal: every workgroup did the same work, so the times rank the
    // workgroups themselves -- in effect the XCDs, whose skew is persistent); afterwards only the share moves.  Re-ranking at every update
    // was measured to flip-flop: the tail holds the tallest slices, so by the count model above a member looks slower than it is, loses
    // its membership to a workgroup of the slow half, and the launch ends 10 % late (profiles/r04m_*).
    const bool first = tab[3] == 0 || members_old == 0;
    int smaller = 0;
    if (live && first)
        for (int q = 0; q < n_wg; ++q) smaller += (slow_s[q] < slow || (slow_s[q] == slow && q < w)) ? 1 : 0;
    const bool fast = live && (first ? smaller < half : member);  // (ranked by workgroup index below)
    const double s_fast = block_sum(fast ? slow : 0.0) / (double)half, s_slow = block_sum((live && !fast) ? slow : 0.0) / (double)(n_wg - half);
    if (!(s_fast > 0.0) || !(s_slow > 0.0)) return;
    double target = (s_slow - s_fast) / (s_slow + s_fast);
    target = target < 0.0 ? 0.0 : target;
    double phi = phi_old + gain * (target - phi_old);
    phi = phi < 0.0 ? 0.0 : (phi > 0.2 ? 0.2 : phi);
    if (phi < 0.002) phi = 0.0;  // (not worth a second phase)
    // rank among the members, by workgroup index
    __shared__ int cnt_wave[16];
    const unsigned long long bal = __ballot(fast);
    if (lane == 0) cnt_wave[wave] = __popcll(bal);
    __syncthreads();
    int before = 0;
    for (int q = 0; q < wave; ++q) before += cnt_wave[q];
    if (live) tab[4 + w] = fast ? before + __popcll(bal & ((1ull << lane) - 1ull)) : -1;  // (kept while the share is 0: the kernel tests n1)
    if (w == 0) {
        const uint32_t tail = (uint32_t)(phi * (double)n_sell);
        tab[0] = (int32_t)(n_sell - tail);
        tab[1] = half * kFusedWaves;
        tab[2] = (int32_t)(phi * 1e6);
        tab[3] += 1;
    }
}

__global__ __launch_bounds__(256) void fair_finish_kernel(const double* __restrict__ partial_fair, int n_wg, double* __restrict__ dense_ax) {
    __shared__ double sh[4];
    double v = 0.0;
    for (int w = threadIdx.x; w < n_wg; w += 256) v += partial_fair[w];
    v = wave_allreduce(v, OpAdd());
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double f = ((sh[0] + sh[1]) + sh[2]) + sh[3];
        dense_ax[0] = f;
        dense_ax[1] = -f;
    }
}

// the fused pass alone: fills the handle's integer slabs, scalar partials and the fixed-point exponent
// the step of the previous iteration CAN ride this handle's launches when every dual entry the kernel reads comes from the
// workgroup's own LDS copy (256-wide layout, whole dual vector and gradient in LDS, no fairness stream)
bool matching_can_fuse_apply(const dl_matching* h) {
    // DUALIP_HIP_FUSE_APPLY=1 / 0 forces it on / off.  Default: on for handles of fewer than kFuseApplyRounds tiles per wavefront.
    // The folded step saves one launch and its boundary (2.7 us, tools/gridsync_bench.hip) and costs the fused launch a longer head.  Round 3
    // (every wavefront derived the step, then requested its rows: +5.4 us of head): -4.5 % per iteration at 1M entities, +-1 % at 10M, neutral at
    // 12.5M -- on below 32 tiles per wavefront.  Round 6 (ONE wavefront derives the step while the others have the rows in flight, fused_common.h:
    // +1.9 us of head): same box, three alternations of this switch (profiles/r06l_fuse_apply_by_size_same_box.txt) -- 3M entities -3.2 %, 10M
    // mixed -1.5 % (whole solve +1.6 %), 10M simplex -1.4 %, the 12.5M-entity rank of eight -0.8 %, 100M neutral: on below 128 (about 19M entities).
    constexpr int64_t kFuseApplyRounds = 128;
    const char* e = plan_env("DUALIP_HIP_FUSE_APPLY");
    bool on = h->n_wg > 0 && (h->n_tiles + h->n_sell) < kFuseApplyRounds * (int64_t)h->n_wg * kFusedWaves;
    if (e && e[0] == '1') on = true;
    if (e && e[0] == '0') on = false;
    return on && h->lam_lds && h->grad_lds && h->m_hot == 0 && !h->fair && (h->n_tiles > 0 || h->n_sell > 0) && h->n_wg > 0;
}

template <class T>
static int fused_typed(dl_matching* h, const void* lambda, double gamma, void* x_out, hipStream_t st, uint64_t owner_uid, const dl_agd* agd, const PendingStep* pending) {
    FusedArgs<T> args;
    args.tiles32 = reinterpret_cast<const uint32_t*>(h->tiles);
    args.rowidx = h->rowidx;
    args.a = static_cast<const T*>(h->a);
    args.c = static_cast<const T*>(h->c);
    args.lambda = static_cast<const T*>(lambda);
    args.x_out = static_cast<T*>(x_out);
    args.projs = h->projs;
    args.partial = static_cast<long long*>(h->partial);
    args.partial_scal = h->partial_scal;
    args.shift_out = h->shift_dev;
    args.gamma = gamma;
    args.amax = h->amax;
    args.cmax = h->cmax;
    args.xmax_bounded = h->xmax_bounded;
    args.pmax_unbounded = h->pmax_unbounded;
    args.row_count_max = (double)(h->row_count_max > 0 ? h->row_count_max : 1);
    args.has_unbounded = h->has_unbounded ? 1 : 0;
    args.m = h->m;
    args.mpad = h->mpad;
    args.nnz = h->nnz_arr > h->nnz ? h->nnz_arr : h->nnz;  // (the arrays' length: what the tiles may read with vector loads; padded when staged)
    args.slab32 = h->slab32 ? 1 : 0;
    args.slab_abound = h->slab_abound;
    args.slab_hi = h->slab_hi;
    args.slab_ovf = h->slab_ovf;
    args.slab_epoch = ++h->slab_epoch;
    args.slab_wide_bits = (h->slab32 && h->n_wide > 0) ? h->slab_wide_bits : nullptr;
    args.slab_wide_list = h->slab_wide_list;
    args.n_wide = h->slab32 ? h->n_wide : 0;
    args.n_proj = h->n_proj;
    args.n_tiles = (uint32_t)h->n_short;
    args.n_long = (uint32_t)(h->n_tiles - h->n_short - h->n_xlong);
    args.n_xlong = (uint32_t)h->n_xlong;
    args.desc_words = (uint32_t)h->desc_words;
    args.long32 = args.tiles32 + (size_t)h->n_short * (size_t)h->desc_words + 12;  // (after the windows and one all-zero descriptor)
    args.ablate = h->ablate;
    args.timeline = h->timeline;
    args.eq_heights = h->eq_heights;
    args.m_hot = h->m_hot;
    args.m_lam = h->m_hot > 0 ? h->m_lam : 0;
    args.cold_grad = h->cold_grad;
    args.cold_per_xcd = h->cold_per_xcd ? 1 : 0;
    args.fair = static_cast<const T*>(h->fair);
    args.lambda_orig = static_cast<const T*>(lambda);
    args.partial_fair = h->partial_fair;
    args.fair_max = h->fair ? h->fair_max : 0.0;
    args.sell_desc = h->sell_desc ? h->sell_desc + (size_t)h->n_sell_lane_slices * 4 : nullptr;  // (the one-lane slices follow the K-lane ones in the table)
    args.sell_lane_desc = h->sell_desc;
    args.n_sell_lanes = (uint32_t)h->n_sell_lane_slices;
    args.sell_lane_begin = h->sell_lane_begin;
    args.sell_len = h->sell_len;
    args.sell_colstart = h->sell_colstart;
    args.sell_a = static_cast<const T*>(h->sell_a);
    args.sell_c = static_cast<const T*>(h->sell_c);
    args.sell_r = h->sell_r;
    args.sell_f = static_cast<const T*>(h->sell_f);
    args.n_sell = (uint32_t)(h->n_sell - h->n_sell_lane_slices);
    args.balance = h->bal;
    args.sell_bal = h->sell_bal;
    // (the first launches of a handle adapt every time, later ones every kBalEvery-th: the balance point moves during a solve -- the
    //  slices get slower as the Newton passes multiply, the windows do not)
    // (not with the fairness stream: its sum f.x is a per-workgroup double, so an adapting deal would put the measured timings into the
    //  last bits of the two dense rows; those handles keep the even deal and stay bit-reproducible like the others)
    args.bal_stamps = (h->bal_stamps && !h->fair && (h->bal_launches < h->bal_first || h->bal_launches % kBalEvery == 0)) ? h->bal_stamps : nullptr;
    args.do_apply = 0;
    args.apply = ApplyArgs<T>();
    if (pending && pending->valid) {
        if (!agd || !matching_can_fuse_apply(h) || agd->m != h->m || agd->val_dtype != h->val_dtype) return fail(DL_E_STATE, "this handle cannot apply an optimiser step in its prologue");
        args.do_apply = 1;
        args.apply = make_apply_args<T>(agd, *pending);
    }
    if (h->m_hot > 0) {  // hot-rows plan: the kernel reads the dual vector in renumbered order and adds the cold rows globally
        // (the device-resident AGD loop leaves both prepared, common.h -- only honoured for the optimiser that prepared them)
        if (!(h->hot_ready && owner_uid != 0 && h->hot_ready_owner == owner_uid && h->hot_ready_lambda == lambda)) {
            const unsigned blocks = (unsigned)((h->m + 255) / 256);
            hipLaunchKernelGGL(permute_vector_kernel<T>, dim3(blocks), dim3(256), 0, st, h->m, static_cast<const T*>(lambda), h->row_inv, static_cast<T*>(h->lam_perm));
            DL_HIP(hipGetLastError());
            DL_HIP(hipMemsetAsync(h->cold_grad, 0, sizeof(long long) * (size_t)h->mpad * (size_t)kColdCopies, st));
        }
        h->hot_ready = false;  // this launch fills the cold accumulators
        args.lambda = static_cast<const T*>(h->lam_perm);
    }
    if (!h->grad_lds) DL_HIP(hipMemsetAsync(h->partial, 0, sizeof(long long) * (size_t)h->mpad, st));
    hipEvent_t ev_stop = nullptr;
    if (h->prof_on && (h->prof_seen++ % (uint64_t)h->prof_stride) == 0) {
        if (h->prof_used == h->prof_start.size() && h->prof_start.size() < 16384) {
            hipEvent_t e0, e1;
            DL_HIP(hipEventCreate(&e0));
            DL_HIP(hipEventCreate(&e1));
            h->prof_start.push_back(e0);
            h->prof_stop.push_back(e1);
        }
        if (h->prof_used < h->prof_start.size()) {
            DL_HIP(hipEventRecord(h->prof_start[h->prof_used], st));
            ev_stop = h->prof_stop[h->prof_used];
            h->prof_used += 1;
        }
    }
    int rc;
    rc = launch_fused4(h, args, st);
    if (rc) return rc;
    if (ev_stop) DL_HIP(hipEventRecord(ev_stop, st));
    if (args.bal_stamps) {  // adapt the per-XCD rounds to what this launch's stamps say (a few microseconds)
        if (h->bal_adapts) {  // (a table declared non-adapting holds 0x7FFFFFFF rounds: never rewritten from it)
            // (the first updates move further: the even deal they start from is several per cent off -- XCDs differ by 5-8 % -- and a benchmark window,
            //  or a short solve, is over before a gain of 0.3 has closed that)
            const double gain_k = std::max(h->bal_gain, h->bal_gain0 * std::pow(0.85, (double)h->bal_launches));
            hipLaunchKernelGGL(wg_balance_kernel, dim3(1), dim3(1024), 0, st, h->bal, h->bal_stamps, h->n_wg, (uint32_t)h->n_short, h->bal_min_rounds, gain_k);
            DL_HIP(hipGetLastError());
        }
        if (h->sell_bal && !h->sell_bal_frozen) {  // (handles whose windows do not adapt: the slices' two-phase deal does)
            hipLaunchKernelGGL(sell_balance_kernel, dim3(1), dim3(1024), 0, st, h->sell_bal, h->bal_stamps, h->n_wg, (uint32_t)(h->n_sell - h->n_sell_lane_slices), 0.5);
            DL_HIP(hipGetLastError());
        }
    }
    if (h->bal_stamps) h->bal_launches += 1;
    if (h->fair) {
        hipLaunchKernelGGL(fair_finish_kernel, dim3(1), dim3(256), 0, st, h->partial_fair, h->n_wg, h->dense_ax);
        DL_HIP(hipGetLastError());
    }
    return 0;
}

int matching_launch_fused(dl_matching* h, const void* lambda, double gamma, void* x_out, hipStream_t st, uint64_t owner_uid, const dl_agd* agd, const PendingStep* pending) {
    if (h->val_dtype == DL_F32) return fused_typed<float>(h, lambda, gamma, x_out, st, owner_uid, agd, pending);
    return fused_typed<double>(h, lambda, gamma, x_out, st, owner_uid, agd, pending);
}


template <class T>
static int calculate_typed(dl_matching* h, const void* lambda, double gamma, double* packed_out, void* x_out, hipStream_t st) {
    if ((h->n_tiles == 0 && h->n_sell == 0) || h->n_wg == 0) {  // no non-zeros at all: A x = 0
        DL_HIP(hipMemsetAsync(packed_out, 0, sizeof(double) * (size_t)(h->m + 2), st));
        return 0;
    }
    int rc = fused_typed<T>(h, lambda, gamma, x_out, st, 0, nullptr, nullptr);
    if (rc) return rc;
    return matching_reduce(h, packed_out, 0, nullptr, st, 0);
}

// The slab reduction alone (after matching_launch_fused).  mode 0: packed = sums; 1: packed += sums; 2: the sums
// (+ packed when `push_accumulate`) go to the mailboxes described by *push.
int matching_reduce(dl_matching* h, double* packed, int mode, const PushArgs* push, hipStream_t st, int push_accumulate) {
    const int n_slabs = h->grad_lds ? h->n_wg : 1;
    const int blocks = (int)((h->m + kRedRows - 1) / kRedRows);
    PushArgs pa = PushArgs();
    if (push) pa = *push;
    auto kern = mode == 0 ? reduce_partials_kernel<0> : (mode == 1 ? reduce_partials_kernel<1> : reduce_partials_kernel<2>);
    hipLaunchKernelGGL(kern, dim3(blocks + 1), dim3(kRedThreads), 0, st, static_cast<const long long*>(h->partial), h->partial_scal, h->shift_dev, n_slabs,
                       h->n_wg, h->m, h->mpad, packed, h->m_hot > 0 ? h->row_inv : nullptr, h->m_hot, h->cold_grad, h->fair ? h->dense_ax : nullptr, pa,
                       push_accumulate, h->slab32 ? 1 : 0, h->slab_hi, h->slab_ovf, h->slab_epoch, h->slab32 && h->n_wide > 0 ? h->slab_wide : nullptr);
    DL_HIP(hipGetLastError());
    return 0;
}

int matching_calculate(dl_matching* h, const void* lambda, double gamma, double* packed_out, void* x_out, hipStream_t st) {
    if (h->val_dtype == DL_F32) return calculate_typed<float>(h, lambda, gamma, packed_out, x_out, st);
    return calculate_typed<double>(h, lambda, gamma, packed_out, x_out, st);
}

}  // namespace dl
