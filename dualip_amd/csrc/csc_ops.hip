// csc_ops.hip -- the reference's stand-alone CSC primitives (src/dualip/utils/sparse_utils.py) as single HIP launches.
// Inside a solve their work is fused into the matching pass (matching_kernels4.hip); these entry points exist for callers
// that use the primitives on their own (data preparation, the reference's tests/test_sparse_utils.py).
#include "common.h"
#include "simplex.h"

namespace dl {

template <class T, class IdxT>
__global__ __launch_bounds__(256) void csc_scale_rows_kernel(int64_t nnz, const IdxT* __restrict__ rowidx, const T* __restrict__ in, const T* __restrict__ v,
                                                             T* __restrict__ out) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) out[k] = (T)(in[k] * v[(int64_t)rowidx[k]]);
}

// column of non-zero k: the last j with colptr[j] <= k (binary search; colptr is L2-resident for the shapes this serves)
template <class IdxT>
__device__ __forceinline__ int64_t column_of(const IdxT* __restrict__ colptr, int64_t n, int64_t k) {
    int64_t lo = 0, hi = n;  // invariant: colptr[lo] <= k < colptr[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)colptr[mid] <= k) lo = mid;
        else hi = mid;
    }
    return lo;
}
template <class T, class IdxT>
__global__ __launch_bounds__(256) void csc_scale_cols_kernel(int64_t n, int64_t nnz, const IdxT* __restrict__ colptr, const T* __restrict__ in,
                                                             const T* __restrict__ v, T* __restrict__ out) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) out[k] = (T)(in[k] * v[column_of(colptr, n, k)]);
}

template <class T>
__global__ __launch_bounds__(256) void csc_elementwise_kernel(int64_t nnz, const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int op) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        const T x = a[k], y = b[k];
        out[k] = op == DL_OP_ADD ? (T)(x + y) : (op == DL_OP_SUB ? (T)(x - y) : (op == DL_OP_MUL ? (T)(x * y) : (T)(x / y)));
    }
}

// row sums: per-workgroup partial sums in double for the rows that fit the LDS (all of them up to 16384), global double
// atomics for the flush; the result is rounded once to the value type
template <class T, class IdxT>
__global__ __launch_bounds__(1024) void csc_row_sums_kernel(int64_t m, int64_t nnz, const IdxT* __restrict__ rowidx, const T* __restrict__ vals, double* __restrict__ acc) {
    extern __shared__ double sh[];
    const int64_t m_lds = m <= 16384 ? m : 0;
    for (int64_t i = threadIdx.x; i < m_lds; i += blockDim.x) sh[i] = 0.0;
    __syncthreads();
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = (int64_t)rowidx[k];
        if (r < m_lds) atomicAdd(&sh[r], (double)vals[k]);
        else atomicAdd(&acc[r], (double)vals[k]);
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < m_lds; i += blockDim.x)
        if (sh[i] != 0.0) atomicAdd(&acc[i], sh[i]);
}
template <class T>
__global__ void round_kernel(int64_t m, const double* __restrict__ acc, T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = (T)acc[i];
}

// apply_F_to_columns for the operators with a kernel form: one thread per selected column, the column's own entries only
// (no zero padding: box / cone are point-wise, and the simplex kinds project over the column's entries exactly as the fused
// pass does -- simplex.h / project_dense_kernel)
template <class T, class IdxT>
__global__ __launch_bounds__(256) void csc_project_columns_kernel(int64_t n_sel, const int64_t* __restrict__ cols, const IdxT* __restrict__ colptr,
                                                                  const T* __restrict__ in, T* __restrict__ out, int kind, T p0, T p1, T ztol) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_sel) return;
    const int64_t j = cols ? cols[q] : q;
    const int64_t k0 = (int64_t)colptr[j], k1 = (int64_t)colptr[j + 1];
    if (kind != DL_PROJ_SIMPLEX && kind != DL_PROJ_SIMPLEX_EQ) {
        for (int64_t k = k0; k < k1; ++k) {
            T v = in[k];
            switch (kind) {
                case DL_PROJ_BOX: v = v < p0 ? p0 : v; v = v > p1 ? p1 : v; break;
                case DL_PROJ_CONE_LOWER: v = v < p0 ? p0 : v; break;
                case DL_PROJ_CONE_UPPER: v = v > p0 ? p0 : v; break;
                default: break;
            }
            out[k] = v;
        }
        return;
    }
    const T z = p0;
    const int64_t L = k1 - k0;
    T S = (T)0, v1 = (T)(-INFINITY);
    for (int64_t k = k0; k < k1; ++k) {
        T u = in[k];
        u = u > (T)0 ? u : (T)0;
        S = (T)(S + u);
        v1 = u > v1 ? u : v1;
    }
    const bool projected = kind == DL_PROJ_SIMPLEX_EQ || S > ztol;
    T th = (T)0;
    bool onehot = false;
    if (projected && L > 0) {
        const T th_a = (T)(v1 - z), th_b = (T)((T)(S - z) / (T)L);
        th = th_a > th_b ? th_a : th_b;
        int64_t cnt_prev = 0;
        for (int64_t it = 0; it <= L + 1; ++it) {
            T sumA = (T)0;
            int64_t cnt = 0;
            for (int64_t k = k0; k < k1; ++k) {
                T u = in[k];
                u = u > (T)0 ? u : (T)0;
                if (u > th) {
                    sumA = (T)(sumA + u);
                    ++cnt;
                }
            }
            if (it == 0 && cnt == 1) {
                onehot = true;
                break;
            }
            if (cnt == cnt_prev || cnt == 0) break;
            const T tn = (T)((T)(sumA - z) / (T)cnt);
            th = tn > th ? tn : th;
            cnt_prev = cnt;
        }
    }
    for (int64_t k = k0; k < k1; ++k) {
        T u = in[k];
        u = u > (T)0 ? u : (T)0;
        T x = u;
        if (projected) {
            if (onehot) x = u > th ? z : (T)0;
            else {
                const T d = (T)(u - th);
                x = d > (T)0 ? d : (T)0;
            }
        }
        out[k] = x;
    }
}

// measurement hook: one streaming pass over a buffer with 16-byte non-temporal loads, eight in flight per lane (what the fused
// kernel's value loads look like) -- the read bandwidth this box reaches, next to the 8 TB/s of the data sheet
typedef float read_vec4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void read_stream_kernel(const read_vec4* __restrict__ p, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        read_vec4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u][0];
    }
    for (; i < n; i += stride) acc += p[i][0];
    if (acc == 12345.678f) *sink = acc;  // (never true for the buffers this is used on: keeps the loads alive)
}

// the same, shaped like the fused kernel's windows: every wavefront streams three arrays side by side (16 + 16 + 8 bytes per lane
// and step, two steps in flight), its steps dealt cyclically over all wavefronts of the launch -- on the boxes measured this reads
// faster than the single stream above (three DRAM fronts instead of one)
typedef float read_vec2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void read_stream3_kernel(const read_vec4* __restrict__ pa, const read_vec4* __restrict__ pc, const read_vec2* __restrict__ pr, size_t n,
                                                            float* __restrict__ sink) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + stride < n; i += 2 * stride) {
        const read_vec4 a0 = __builtin_nontemporal_load(pa + i), c0 = __builtin_nontemporal_load(pc + i);
        const read_vec2 r0 = __builtin_nontemporal_load(pr + i);
        const read_vec4 a1 = __builtin_nontemporal_load(pa + i + stride), c1 = __builtin_nontemporal_load(pc + i + stride);
        const read_vec2 r1 = __builtin_nontemporal_load(pr + i + stride);
        acc += a0[0] + c0[0] + r0[0] + a1[0] + c1[0] + r1[0];
    }
    if (acc == 12345.678f) *sink = acc;
}

// the three-stream shape with DEPTH steps of a lane in flight (160 bytes per lane at DEPTH 4, 320 at 8: two to four times what the fused
// kernel keeps in flight per lane) and two workgroups' worth of wavefronts per CU when the registers allow: the probe must not be the
// thing that runs out of outstanding requests before the memory does (round 4's two-deep probe was beaten by the kernel it judged)
template <int DEPTH>
__global__ __launch_bounds__(1024) void read_stream3_deep_kernel(const read_vec4* __restrict__ pa, const read_vec4* __restrict__ pc, const read_vec2* __restrict__ pr, size_t n,
                                                                 float* __restrict__ sink) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (DEPTH - 1) * stride < n; i += DEPTH * stride) {
        read_vec4 a[DEPTH], c[DEPTH];
        read_vec2 r[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            a[u] = __builtin_nontemporal_load(pa + i + u * stride);
            c[u] = __builtin_nontemporal_load(pc + i + u * stride);
            r[u] = __builtin_nontemporal_load(pr + i + u * stride);
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) acc += a[u][0] + c[u][0] + r[u][0];
    }
    if (acc == 12345.678f) *sink = acc;
}

// ... and with the work CLAIMED instead of dealt: the XCDs of this part do not stream at the same speed (the fused kernel deals its window
// tiles by measured shares for that reason, fused_common.h: Deal), so a static deal ends with the slow XCDs' tail while the fast ones idle --
// which is how a kernel could beat the probe that judges it.  Every workgroup claims the next chunk of kDynSteps x 1024 lanes from a counter
// (one atomic per 160 KB) until the arrays are exhausted.
constexpr int kDynSteps = 4;
__global__ __launch_bounds__(1024) void read_stream3_dyn_kernel(const read_vec4* __restrict__ pa, const read_vec4* __restrict__ pc, const read_vec2* __restrict__ pr, size_t n,
                                                                unsigned long long* __restrict__ counter, float* __restrict__ sink) {
    __shared__ unsigned long long chunk_s;
    float acc = 0.f;
    const size_t per = (size_t)kDynSteps * 1024;
    for (;;) {
        if (threadIdx.x == 0) chunk_s = atomicAdd(counter, 1ull);
        __syncthreads();
        const size_t base = (size_t)chunk_s * per;
        __syncthreads();
        if (base >= n) break;
        read_vec4 a[kDynSteps], c[kDynSteps];
        read_vec2 r[kDynSteps];
#pragma unroll
        for (int u = 0; u < kDynSteps; ++u) {
            size_t i = base + (size_t)u * 1024 + threadIdx.x;
            i = i < n ? i : n - 1;
            a[u] = __builtin_nontemporal_load(pa + i);
            c[u] = __builtin_nontemporal_load(pc + i);
            r[u] = __builtin_nontemporal_load(pr + i);
        }
#pragma unroll
        for (int u = 0; u < kDynSteps; ++u) acc += a[u][0] + c[u][0] + r[u][0];
    }
    if (acc == 12345.678f) *sink = acc;
}

static int grid_for(int64_t n, int threads) {
    const int64_t b = (n + threads - 1) / threads;
    return (int)(b > 8192 ? 8192 : (b > 0 ? b : 1));
}

}  // namespace dl

using namespace dl;

#define DISPATCH_T_IDX(CALL)                                                            \
    do {                                                                                \
        if (val_dtype == DL_F32) {                                                      \
            if (idx_dtype == DL_I32) { CALL(float, int32_t); } else { CALL(float, int64_t); }   \
        } else {                                                                        \
            if (idx_dtype == DL_I32) { CALL(double, int32_t); } else { CALL(double, int64_t); } \
        }                                                                               \
    } while (0)

extern "C" {

int dl_measure_read_bandwidth(const void* buf, int64_t bytes, int32_t reps, double* gbps_out_host, dl_stream_t stream) {
    if (!buf || bytes < (1 << 20) || reps < 1 || !gbps_out_host || (reinterpret_cast<uintptr_t>(buf) & 15u)) return fail(DL_E_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    float* sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    DL_HIP(hipMalloc((void**)&sink, sizeof(float)));
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    // five access shapes, the best one is reported (the last: the three streams with dynamically claimed chunks): one stream of 16-byte loads; three streams side by side (40 bytes per lane and step), two,
    // four or eight steps of a lane in flight
    const size_t n3 = ((size_t)bytes / 40) & ~(size_t)((size_t)256 * 1024 * 8 - 1);  // lanes x steps of the three-stream shapes (whole rounds at every depth)
    const char* base = (const char*)buf;
    double best_gbps = 0.0;
    unsigned long long* dyn_counter = nullptr;
    if (e == hipSuccess) e = hipMalloc((void**)&dyn_counter, sizeof(unsigned long long));
    for (int shape = 0; shape < 5 && e == hipSuccess; ++shape) {
        if (shape > 0 && n3 == 0) break;
        float best = 1e30f;
        const double moved = shape == 0 ? (double)bytes : (double)n3 * 40.0;
        for (int r = 0; r <= reps && e == hipSuccess; ++r) {  // (first pass untimed)
            if (shape == 4) e = hipMemsetAsync(dyn_counter, 0, sizeof(unsigned long long), st);  // (ahead of the timed launch)
            if (e == hipSuccess) e = hipEventRecord(e0, st);
            if (shape == 0)
                hipLaunchKernelGGL(read_stream_kernel, dim3(256), dim3(1024), 0, st, (const read_vec4*)buf, (size_t)bytes / 16, sink);
            else if (shape == 1)
                hipLaunchKernelGGL(read_stream3_kernel, dim3(256), dim3(1024), 0, st, (const read_vec4*)base, (const read_vec4*)(base + n3 * 16),
                                   (const read_vec2*)(base + n3 * 32), n3, sink);
            else if (shape == 2)
                hipLaunchKernelGGL(read_stream3_deep_kernel<4>, dim3(512), dim3(1024), 0, st, (const read_vec4*)base, (const read_vec4*)(base + n3 * 16),
                                   (const read_vec2*)(base + n3 * 32), n3, sink);
            else if (shape == 3)
                hipLaunchKernelGGL(read_stream3_deep_kernel<8>, dim3(256), dim3(1024), 0, st, (const read_vec4*)base, (const read_vec4*)(base + n3 * 16),
                                   (const read_vec2*)(base + n3 * 32), n3, sink);
            else
                hipLaunchKernelGGL(read_stream3_dyn_kernel, dim3(512), dim3(1024), 0, st, (const read_vec4*)base, (const read_vec4*)(base + n3 * 16),
                                   (const read_vec2*)(base + n3 * 32), n3, dyn_counter, sink);
            if (e == hipSuccess) e = hipEventRecord(e1, st);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && ms < best) best = ms;
        }
        const double gbps = moved / ((double)best * 1e-3) / 1e9;
        if (gbps > best_gbps) best_gbps = gbps;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (dyn_counter) (void)hipFree(dyn_counter);
    (void)hipFree(sink);
    if (e != hipSuccess) return hip_fail(e, "read bandwidth measurement");
    *gbps_out_host = best_gbps;
    return 0;
}

int dl_csc_scale_rows(int64_t nnz, const void* rowidx, int idx_dtype, const void* vals_in, const void* v, void* vals_out, int val_dtype, dl_stream_t stream) {
    if (nnz < 0 || (nnz > 0 && (!rowidx || !vals_in || !v || !vals_out))) return fail(DL_E_ARG, "bad argument");
    if ((idx_dtype != DL_I32 && idx_dtype != DL_I64) || (val_dtype != DL_F32 && val_dtype != DL_F64)) return fail(DL_E_ARG, "bad dtype");
    if (nnz == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
#define CALL(T, I) hipLaunchKernelGGL((csc_scale_rows_kernel<T, I>), dim3(grid_for(nnz, 256)), dim3(256), 0, st, nnz, (const I*)rowidx, (const T*)vals_in, (const T*)v, (T*)vals_out)
    DISPATCH_T_IDX(CALL);
#undef CALL
    DL_HIP(hipGetLastError());
    return 0;
}

int dl_csc_scale_cols(int64_t n, int64_t nnz, const void* colptr, int idx_dtype, const void* vals_in, const void* v, void* vals_out, int val_dtype,
                      dl_stream_t stream) {
    if (n < 0 || nnz < 0 || !colptr || (nnz > 0 && (!vals_in || !v || !vals_out))) return fail(DL_E_ARG, "bad argument");
    if ((idx_dtype != DL_I32 && idx_dtype != DL_I64) || (val_dtype != DL_F32 && val_dtype != DL_F64)) return fail(DL_E_ARG, "bad dtype");
    if (nnz == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
#define CALL(T, I) hipLaunchKernelGGL((csc_scale_cols_kernel<T, I>), dim3(grid_for(nnz, 256)), dim3(256), 0, st, n, nnz, (const I*)colptr, (const T*)vals_in, (const T*)v, (T*)vals_out)
    DISPATCH_T_IDX(CALL);
#undef CALL
    DL_HIP(hipGetLastError());
    return 0;
}

int dl_csc_elementwise(int64_t nnz, const void* a, const void* b, void* out, int op, int val_dtype, dl_stream_t stream) {
    if (nnz < 0 || (nnz > 0 && (!a || !b || !out))) return fail(DL_E_ARG, "bad argument");
    if (op < DL_OP_ADD || op > DL_OP_DIV) return fail(DL_E_ARG, "unknown element-wise operation %d", op);
    if (val_dtype != DL_F32 && val_dtype != DL_F64) return fail(DL_E_ARG, "bad dtype");
    if (nnz == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (val_dtype == DL_F32) hipLaunchKernelGGL(csc_elementwise_kernel<float>, dim3(grid_for(nnz, 256)), dim3(256), 0, st, nnz, (const float*)a, (const float*)b, (float*)out, op);
    else hipLaunchKernelGGL(csc_elementwise_kernel<double>, dim3(grid_for(nnz, 256)), dim3(256), 0, st, nnz, (const double*)a, (const double*)b, (double*)out, op);
    DL_HIP(hipGetLastError());
    return 0;
}

int dl_csc_row_sums(int64_t m, int64_t nnz, const void* rowidx, int idx_dtype, const void* vals, void* out, int val_dtype, dl_stream_t stream) {
    if (m < 0 || nnz < 0 || (nnz > 0 && (!rowidx || !vals)) || (m > 0 && !out)) return fail(DL_E_ARG, "bad argument");
    if ((idx_dtype != DL_I32 && idx_dtype != DL_I64) || (val_dtype != DL_F32 && val_dtype != DL_F64)) return fail(DL_E_ARG, "bad dtype");
    if (m == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    double* acc = nullptr;
    DL_HIP(hipMalloc((void**)&acc, sizeof(double) * (size_t)m));
    hipError_t e = hipMemsetAsync(acc, 0, sizeof(double) * (size_t)m, st);
    if (e == hipSuccess && nnz > 0) {
        const size_t lds = m <= 16384 ? sizeof(double) * (size_t)m : 0;
        const int blocks = grid_for(nnz, 1024 * 16);
#define CALL(T, I)                                                                                                                          \
    do {                                                                                                                                    \
        auto kern = csc_row_sums_kernel<T, I>;                                                                                              \
        if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, st, m, nnz, (const I*)rowidx, (const T*)vals, acc);      \
    } while (0)
        DISPATCH_T_IDX(CALL);
#undef CALL
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e == hipSuccess) {
        const unsigned blocks = (unsigned)((m + 255) / 256);
        if (val_dtype == DL_F32) hipLaunchKernelGGL(round_kernel<float>, dim3(blocks), dim3(256), 0, st, m, acc, (float*)out);
        else hipLaunchKernelGGL(round_kernel<double>, dim3(blocks), dim3(256), 0, st, m, acc, (double*)out);
        e = hipGetLastError();
    }
    hipError_t e2 = hipStreamSynchronize(st);  // acc is a temporary
    (void)hipFree(acc);
    if (e != hipSuccess) return hip_fail(e, "row sums");
    if (e2 != hipSuccess) return hip_fail(e2, "row sums");
    return 0;
}

int dl_csc_project_columns(int64_t n_sel, const int64_t* cols, const void* colptr, int idx_dtype, const void* vals_in, void* vals_out, const dl_proj_desc* proj_host,
                           int val_dtype, dl_stream_t stream) {
    if (n_sel < 0 || !colptr || !proj_host || (n_sel > 0 && (!vals_in || !vals_out))) return fail(DL_E_ARG, "bad argument");
    if ((idx_dtype != DL_I32 && idx_dtype != DL_I64) || (val_dtype != DL_F32 && val_dtype != DL_F64)) return fail(DL_E_ARG, "bad dtype");
    const int k = proj_host->kind;
    if (k < DL_PROJ_NONE || k > DL_PROJ_SIMPLEX_EQ) return fail(DL_E_PROJ, "Unknown projection operator kind %d", k);
    if ((k == DL_PROJ_SIMPLEX || k == DL_PROJ_SIMPLEX_EQ) && !(proj_host->p0 > 0.0)) return fail(DL_E_PROJ, "Simplex radius z must be positive.");
    if (n_sel == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((n_sel + 255) / 256);
#define CALL(T, I)                                                                                                                                      \
    hipLaunchKernelGGL((csc_project_columns_kernel<T, I>), dim3(blocks), dim3(256), 0, st, n_sel, cols, (const I*)colptr, (const T*)vals_in, (T*)vals_out, k, \
                       (T)proj_host->p0, (T)proj_host->p1, (T)(proj_host->p0 + 1e-6))
    DISPATCH_T_IDX(CALL);
#undef CALL
    DL_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
