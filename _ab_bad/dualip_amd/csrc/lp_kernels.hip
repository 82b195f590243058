// lp_kernels.hip -- the generic-LP ("miplib2017") dual objective on gfx950 (reference: src/dualip/objectives/miplib.py:60-109).
//
//   z = (-1/gamma) * (A^T lambda' + c),  x = clamp(z, lower, upper),  A x,  c.x,  sum x^2        lambda' = lambda / row_norms
//
// x has one entry per VARIABLE (column), unlike the matching objective's one per non-zero, so the pass is two sparse
// mat-vecs around a point-wise clamp: A^T lambda over the CSC arrays (one wavefront per column, lanes stride the
// column, wave reduction) and A x over the CSR arrays (one wavefront per row).  The shipped instance has 1e5 non-zeros:
// these launches are latency bound by construction -- the design goal here is the result,
// bit-reproducible (fixed summation order), behind the same packed [A x | c.x | sum x^2] interface the matching pass feeds
// to the device-resident optimiser.
#include <new>

#include "common.h"
#include "wave.h"

namespace dl {

struct dl_lp_impl {
    int64_t m = 0, n = 0, nnz = 0;
    int val_dtype = DL_F32;
    const int64_t* colptr = nullptr;
    const int32_t* rowidx = nullptr;
    const void* vals_csc = nullptr;
    const int64_t* rowptr = nullptr;
    const int32_t* colidx = nullptr;
    const void* vals_csr = nullptr;
    const void* c = nullptr;
    const void* lo = nullptr;
    const void* hi = nullptr;
    const void* inv_norm = nullptr;  // 1 / row norms (Jacobi) or null
    void* x = nullptr;               // owned scratch, val[n]
};

constexpr int kLpThreads = 256;  // 4 wavefronts = 4 columns / rows per workgroup

// x_j = clamp((-1/gamma) * (sum_k a_k * lambda'[r_k] + c_j), lo_j, hi_j)   (miplib.py:77-92; clamp = box.py:15-16 / cone.py:21-28)
template <class T>
__global__ __launch_bounds__(kLpThreads) void lp_primal_kernel(int64_t n, const int64_t* __restrict__ colptr, const int32_t* __restrict__ rowidx,
                                                                const T* __restrict__ vals, const T* __restrict__ lambda, const T* __restrict__ inv_norm,
                                                                const T* __restrict__ c, const T* __restrict__ lo, const T* __restrict__ hi, double gamma,
                                                                int apply_bounds, T* __restrict__ x_out) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * (kLpThreads / 64) + (threadIdx.x >> 6);
    if (j >= n) return;
    const int64_t k0 = colptr[j], k1 = colptr[j + 1];
    T s = (T)0;
    for (int64_t k = k0 + lane; k < k1; k += 64) {
        const int32_t r = rowidx[k];
        const T lam = inv_norm ? (T)(inv_norm[r] * lambda[r]) : lambda[r];  // miplib.py:74-75
        s = (T)(s + (T)(vals[k] * lam));
    }
    s = wave_allreduce(s, OpAdd());
    if (lane == 0) {
        const T z = (T)((T)(-1.0 / gamma) * (T)(s + c[j]));
        T x = z;
        if (apply_bounds) {
            x = x > lo[j] ? x : lo[j];  // NaN-free bounds: -inf / +inf where absent
            x = x < hi[j] ? x : hi[j];
        }
        x_out[j] = x;
    }
}

// packed[i] = (A x)_i [* inv_norm_i]; the extra last workgroup sums c.x and x.x in a fixed order
template <class T>
__global__ __launch_bounds__(kLpThreads) void lp_gradient_kernel(int64_t m, int64_t n, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                                  const T* __restrict__ vals, const T* __restrict__ x, const T* __restrict__ inv_norm,
                                                                  const T* __restrict__ c, double* __restrict__ packed) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x + 1 == gridDim.x) {
        __shared__ double sh[2 * kLpThreads / 64];
        double o = 0.0, q = 0.0;
        for (int64_t j = threadIdx.x; j < n; j += kLpThreads) {
            const T xj = x[j];
            o += (double)(T)(c[j] * xj);
            q += (double)(T)(xj * xj);
        }
        o = wave_allreduce(o, OpAdd());
        q = wave_allreduce(q, OpAdd());
        if (lane == 0) {
            sh[2 * (threadIdx.x >> 6)] = o;
            sh[2 * (threadIdx.x >> 6) + 1] = q;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double oo = 0.0, qq = 0.0;
            for (int w = 0; w < kLpThreads / 64; ++w) {
                oo += sh[2 * w];
                qq += sh[2 * w + 1];
            }
            packed[m] = oo;
            packed[m + 1] = qq;
        }
        return;
    }
    const int64_t i = (int64_t)blockIdx.x * (kLpThreads / 64) + (threadIdx.x >> 6);
    if (i >= m) return;
    const int64_t k0 = rowptr[i], k1 = rowptr[i + 1];
    T s = (T)0;
    for (int64_t k = k0 + lane; k < k1; k += 64) s = (T)(s + (T)(vals[k] * x[colidx[k]]));
    s = wave_allreduce(s, OpAdd());
    if (lane == 0) packed[i] = inv_norm ? (double)(T)(inv_norm[i] * s) : (double)s;
}

template <class T>
static int lp_primal_typed(const dl_lp_impl* h, const void* lambda, double gamma, int apply_bounds, void* x_out, hipStream_t st) {
    if (h->n == 0) return 0;
    const unsigned grid = (unsigned)((h->n + kLpThreads / 64 - 1) / (kLpThreads / 64));
    hipLaunchKernelGGL(lp_primal_kernel<T>, dim3(grid), dim3(kLpThreads), 0, st, h->n, h->colptr, h->rowidx, (const T*)h->vals_csc, (const T*)lambda,
                       (const T*)h->inv_norm, (const T*)h->c, (const T*)h->lo, (const T*)h->hi, gamma, apply_bounds, (T*)x_out);
    DL_HIP(hipGetLastError());
    return 0;
}

template <class T>
static int lp_gradient_typed(const dl_lp_impl* h, const void* x, double* packed, hipStream_t st) {
    const unsigned grid = (unsigned)((h->m + kLpThreads / 64 - 1) / (kLpThreads / 64)) + 1u;
    hipLaunchKernelGGL(lp_gradient_kernel<T>, dim3(grid), dim3(kLpThreads), 0, st, h->m, h->n, h->rowptr, h->colidx, (const T*)h->vals_csr, (const T*)x,
                       (const T*)h->inv_norm, (const T*)h->c, packed);
    DL_HIP(hipGetLastError());
    return 0;
}

int lp_primal(const dl_lp_impl* h, const void* lambda, double gamma, int apply_bounds, void* x_out, hipStream_t st) {
    return h->val_dtype == DL_F32 ? lp_primal_typed<float>(h, lambda, gamma, apply_bounds, x_out, st)
                                  : lp_primal_typed<double>(h, lambda, gamma, apply_bounds, x_out, st);
}
int lp_gradient(const dl_lp_impl* h, const void* x, double* packed, hipStream_t st) {
    return h->val_dtype == DL_F32 ? lp_gradient_typed<float>(h, x, packed, st) : lp_gradient_typed<double>(h, x, packed, st);
}

}  // namespace dl

// ---------------------------------------------------------------------------------------------------------
// extern "C" surface (include/dualip_hip.h)
// ---------------------------------------------------------------------------------------------------------
namespace dl {
int fail(int code, const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
}  // namespace dl

struct dl_lp : dl::dl_lp_impl {};

extern "C" {

int dl_lp_create(dl_lp** out, int64_t m, int64_t n, int64_t nnz, const int64_t* colptr, const int32_t* rowidx, const void* vals_csc,
                 const int64_t* rowptr, const int32_t* colidx, const void* vals_csr, const void* c, const void* lower, const void* upper,
                 const void* inv_row_norm, int val_dtype) {
    using namespace dl;
    if (!out) return fail(DL_E_ARG, "out is null");
    *out = nullptr;
    if (m < 0 || n < 0 || nnz < 0) return fail(DL_E_ARG, "negative size");
    if (val_dtype != DL_F32 && val_dtype != DL_F64) return fail(DL_E_ARG, "bad val_dtype");
    if (!colptr || !rowptr || (n > 0 && (!c || !lower || !upper)) || (nnz > 0 && (!rowidx || !colidx || !vals_csc || !vals_csr)))
        return fail(DL_E_ARG, "null array");
    dl_lp* h = new (std::nothrow) dl_lp();
    if (!h) return fail(DL_E_NOMEM, "out of host memory");
    h->m = m;
    h->n = n;
    h->nnz = nnz;
    h->val_dtype = val_dtype;
    h->colptr = colptr;
    h->rowidx = rowidx;
    h->vals_csc = vals_csc;
    h->rowptr = rowptr;
    h->colidx = colidx;
    h->vals_csr = vals_csr;
    h->c = c;
    h->lo = lower;
    h->hi = upper;
    h->inv_norm = inv_row_norm;
    const hipError_t e = hipMalloc(&h->x, (size_t)(n > 0 ? n : 1) * (val_dtype == DL_F32 ? 4 : 8));
    if (e != hipSuccess) {
        delete h;
        return hip_fail(e, "dl_lp_create");
    }
    *out = h;
    return 0;
}

int dl_lp_destroy(dl_lp* h) {
    if (!h) return 0;
    if (h->x) (void)hipFree(h->x);
    delete h;
    return 0;
}

int dl_lp_primal(dl_lp* h, const void* lambda, double gamma, int apply_bounds, void* x_out, dl_stream_t stream) {
    using namespace dl;
    if (!h || (h->m > 0 && !lambda) || (h->n > 0 && !x_out)) return fail(DL_E_ARG, "null argument");
    if (!(gamma > 0.0)) return fail(DL_E_ARG, "gamma must be positive");
    return lp_primal(h, lambda, gamma, apply_bounds, x_out, (hipStream_t)stream);
}

int dl_lp_gradient(dl_lp* h, const void* x, double* packed_out, dl_stream_t stream) {
    using namespace dl;
    if (!h || !packed_out || (h->n > 0 && !x)) return fail(DL_E_ARG, "null argument");
    return lp_gradient(h, x, packed_out, (hipStream_t)stream);
}

int dl_lp_calculate(dl_lp* h, const void* lambda, double gamma, double* packed_out, void* x_out, dl_stream_t stream) {
    using namespace dl;
    if (!h || !packed_out || (h->m > 0 && !lambda)) return fail(DL_E_ARG, "null argument");
    if (!(gamma > 0.0)) return fail(DL_E_ARG, "gamma must be positive");
    void* x = x_out ? x_out : h->x;
    int rc = lp_primal(h, lambda, gamma, 1, x, (hipStream_t)stream);
    if (rc) return rc;
    return lp_gradient(h, x, packed_out, (hipStream_t)stream);
}

}  // extern "C"
