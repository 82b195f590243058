/*
 * oracle/matching_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's matching dual objective (linkedin/DuaLip v5.0.1) used ONLY as
 * the parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
 * dualip_amd/ may import, link or call this file.
 *
 * What it restates (reference file:line, paths relative to /root/reference):
 *   - per non-zero  v_k = a_k * (-(1/gamma) * lambda[row_k]) + (-(1/gamma) * c_k)
 *         src/dualip/objectives/matching.py:136-142, src/dualip/utils/sparse_utils.py:79,46
 *   - per column    x = Proj(v)  (box / cone / simplex / simplex_eq)
 *         src/dualip/objectives/matching.py:145-150, src/dualip/utils/sparse_utils.py:179-220 (zero padded
 *         [L x K] block, L = max column length of the projection entry when batching=False)
 *         src/dualip/projections/box.py:15-16, cone.py:21-28, simplex.py:126-236 (_duchi_proj)
 *   - (A x)_i = sum_{k: row_k = i} a_k x_k       matching.py:153, sparse_utils.py:236-243
 *   - sum x^2 (reg = gamma/2 * ||x||^2) and c.x  matching.py:156-160
 * The "- b", dual objective and slack epilogue (matching.py:25-34,164-178) is restated in oracle/agd_oracle.py.
 *
 * Arithmetic follows the reference's operation ORDER in the working precision T (float or double):
 * the scalar -1/gamma is formed in double and rounded to T once; every product/sum is rounded to T.
 * The three global reductions accumulate in double (the reference reduces in T with an unspecified
 * SIMD order; that difference is below the parity tolerance and documented in DESIGN.md).
 *
 * Pinned against golden vectors produced by the reference itself: tests/golden/g1_*.npz, gp_projections.npz
 * (tests/test_oracle_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { ORACLE_PROJ_NONE = 0, ORACLE_PROJ_BOX = 1, ORACLE_PROJ_CONE_LOWER = 2, ORACLE_PROJ_CONE_UPPER = 3,
       ORACLE_PROJ_SIMPLEX = 4, ORACLE_PROJ_SIMPLEX_EQ = 5 };

typedef struct {
    int32_t kind;     /* ORACLE_PROJ_* */
    int32_t lblock;   /* height L of the reference's zero-padded block for this entry (>= every column length) */
    double p0;        /* box: lower | cone: bound | simplex: z */
    double p1;        /* box: upper */
} oracle_proj_t;

static int oracle_tid(void);

#define DEFINE_ORACLE(T, SUF)                                                                                   \
    static int cmp_desc_##SUF(const void* pa, const void* pb) {                                                 \
        T x = *(const T*)pa, y = *(const T*)pb;                                                                 \
        return (x < y) - (x > y);                                                                               \
    }                                                                                                           \
    /* _duchi_proj on ONE column padded with zeros to height L (simplex.py:126-236). v has L entries. */        \
    static void duchi_column_##SUF(T* v, int64_t L, T* sorted, double z, int inequality) {                     \
        const double tol = 1e-6;                                                                                \
        for (int64_t i = 0; i < L; ++i) v[i] = v[i] > (T)0 ? v[i] : (T)0;      /* simplex.py:149 */           \
        if (inequality) {                                                                                       \
            T s = (T)0;                                                                                         \
            for (int64_t i = 0; i < L; ++i) s = (T)(s + v[i]);                                                  \
            if (s <= (T)(z + tol)) return;                                      /* simplex.py:155-156 */       \
        }                                                                                                       \
        if (L > 1) {                                                            /* simplex.py:167-193 */       \
            T v1 = -INFINITY, v2 = -INFINITY;                                                                   \
            int64_t i1 = -1;                                                                                    \
            for (int64_t i = 0; i < L; ++i) {                                                                   \
                T q = (T)(v[i] / (T)z);                                                                         \
                if (q > v1) { v2 = v1; v1 = q; i1 = i; }                                                        \
                else if (q > v2) { v2 = q; }                                                                    \
            }                                                                                                   \
            if ((T)(v1 - v2) > (T)1.0) {                                                                        \
                for (int64_t i = 0; i < L; ++i) v[i] = (T)0;                                                    \
                v[i1] = (T)z;                                                                                   \
                return;                                                                                         \
            }                                                                                                   \
        }                                                                                                       \
        memcpy(sorted, v, (size_t)L * sizeof(T));                               /* simplex.py:209-231 */       \
        qsort(sorted, (size_t)L, sizeof(T), cmp_desc_##SUF);                                                    \
        T cum = (T)0, cum_at_rho = (T)0;                                                                        \
        int64_t rho = 0;                                                                                        \
        int found = 0;                                                                                          \
        for (int64_t i = 0; i < L; ++i) {                                                                       \
            cum = (T)(cum + sorted[i]);                                                                         \
            T t = (T)(sorted[i] - (T)((T)(cum - (T)z) / (T)(i + 1)));                                           \
            if (t > (T)0) { rho = i; cum_at_rho = cum; found = 1; }                                             \
        }                                                                                                       \
        if (!found) { rho = 0; cum_at_rho = sorted[0]; }   /* mask.max() == 0 -> rho index 0 */                 \
        T theta = (T)((T)(cum_at_rho - (T)z) / (T)((T)rho + (T)1));                                             \
        for (int64_t i = 0; i < L; ++i) {                                                                       \
            T d = (T)(v[i] - theta);                                                                            \
            v[i] = d > (T)0 ? d : (T)0;                                                                         \
        }                                                                                                       \
    }                                                                                                           \
    /* Project one column of true length len in place; buf/sorted have room for max(len, lblock). */           \
    static void project_column_##SUF(T* v, int64_t len, const oracle_proj_t* p, T* buf, T* sorted) {           \
        switch (p->kind) {                                                                                      \
            case ORACLE_PROJ_BOX:                                               /* box.py:15-16 */             \
                for (int64_t i = 0; i < len; ++i) {                                                             \
                    T x = v[i];                                                                                 \
                    x = x < (T)p->p0 ? (T)p->p0 : x;                                                            \
                    x = x > (T)p->p1 ? (T)p->p1 : x;                                                            \
                    v[i] = x;                                                                                   \
                }                                                                                               \
                break;                                                                                          \
            case ORACLE_PROJ_CONE_LOWER:                                        /* cone.py:22-23 */            \
                for (int64_t i = 0; i < len; ++i) v[i] = v[i] < (T)p->p0 ? (T)p->p0 : v[i];                     \
                break;                                                                                          \
            case ORACLE_PROJ_CONE_UPPER:                                        /* cone.py:24-25 */            \
                for (int64_t i = 0; i < len; ++i) v[i] = v[i] > (T)p->p0 ? (T)p->p0 : v[i];                     \
                break;                                                                                          \
            case ORACLE_PROJ_SIMPLEX:                                                                           \
            case ORACLE_PROJ_SIMPLEX_EQ: {                                                                      \
                int64_t L = p->lblock > len ? p->lblock : len;                                                  \
                memcpy(buf, v, (size_t)len * sizeof(T));                                                        \
                for (int64_t i = len; i < L; ++i) buf[i] = (T)0;               /* sparse_utils.py:205-206 */  \
                duchi_column_##SUF(buf, L, sorted, p->p0, p->kind == ORACLE_PROJ_SIMPLEX);                      \
                memcpy(v, buf, (size_t)len * sizeof(T));                       /* sparse_utils.py:212 */      \
                break;                                                                                          \
            }                                                                                                   \
            default: break; /* identity: cone with neither bound (cone.py:26-28) / uncovered column */          \
        }                                                                                                       \
    }                                                                                                           \
    /* grad_out[m] = A x (no "- b"); scal_out[0] = c.x ; scal_out[1] = sum x^2 ; x_out[nnz] optional. */       \
    int oracle_matching_calculate_##SUF(int64_t m, int64_t n, const int64_t* colptr, const int64_t* rowidx,    \
                                        const T* a, const T* c, const int32_t* col_proj,                       \
                                        const oracle_proj_t* projs, int32_t n_proj, const T* lam, double gamma, \
                                        T* grad_out, double* scal_out, T* x_out, int32_t n_threads) {          \
        int64_t maxlen = 1;                                                                                     \
        for (int64_t j = 0; j < n; ++j) {                                                                       \
            int64_t len = colptr[j + 1] - colptr[j];                                                            \
            if (len < 0) return 1;                                                                              \
            if (len > maxlen) maxlen = len;                                                                     \
        }                                                                                                       \
        for (int32_t q = 0; q < n_proj; ++q)                                                                    \
            if (projs[q].lblock > maxlen) maxlen = projs[q].lblock;                                             \
        const T s = (T)(-1.0 / gamma);                                          /* matching.py:136 */          \
        T* scaled = (T*)malloc((size_t)(m > 0 ? m : 1) * sizeof(T));                                            \
        for (int64_t i = 0; i < m; ++i) scaled[i] = (T)(s * lam[i]);                                            \
        int nt = n_threads > 0 ? n_threads : 1;                                                                 \
        double* gacc = (double*)calloc((size_t)nt * (size_t)(m > 0 ? m : 1), sizeof(double));                   \
        double* sacc = (double*)calloc((size_t)nt * 2, sizeof(double));                                         \
        int bad = 0;                                                                                            \
        _Pragma("omp parallel num_threads(nt)")                                                                 \
        {                                                                                                       \
            const int tid = oracle_tid();                                                                       \
            T* v = (T*)malloc((size_t)maxlen * sizeof(T));                                                      \
            T* buf = (T*)malloc((size_t)maxlen * sizeof(T));                                                    \
            T* sorted = (T*)malloc((size_t)maxlen * sizeof(T));                                                 \
            double* g = gacc + (size_t)tid * (size_t)m;                                                         \
            double obj = 0.0, ssq = 0.0;                                                                        \
            _Pragma("omp for schedule(static)")                                                                 \
            for (int64_t j = 0; j < n; ++j) {                                                                   \
                const int64_t k0 = colptr[j], len = colptr[j + 1] - colptr[j];                                  \
                if (len == 0) continue;                                                                         \
                for (int64_t t = 0; t < len; ++t) {                                                             \
                    const int64_t r = rowidx[k0 + t];                                                           \
                    if (r < 0 || r >= m) { bad = 1; v[t] = (T)0; continue; }                                    \
                    T w = (T)(a[k0 + t] * scaled[r]);                           /* sparse_utils.py:79 */       \
                    v[t] = (T)(w + (T)(s * c[k0 + t]));                         /* matching.py:66,142 */       \
                }                                                                                               \
                const int32_t pid = col_proj ? col_proj[j] : (n_proj > 0 ? 0 : -1);                             \
                if (pid >= 0 && pid < n_proj) project_column_##SUF(v, len, &projs[pid], buf, sorted);           \
                for (int64_t t = 0; t < len; ++t) {                                                             \
                    const int64_t r = rowidx[k0 + t];                                                           \
                    if (r < 0 || r >= m) continue;                                                              \
                    const T x = v[t];                                                                           \
                    g[r] += (double)(T)(a[k0 + t] * x);                         /* matching.py:153 */          \
                    obj += (double)(T)(c[k0 + t] * x);                          /* matching.py:160 */          \
                    ssq += (double)(T)(x * x);                                  /* matching.py:157 */          \
                    if (x_out) x_out[k0 + t] = x;                                                               \
                }                                                                                               \
            }                                                                                                   \
            sacc[2 * tid] = obj;                                                                                \
            sacc[2 * tid + 1] = ssq;                                                                            \
            free(v); free(buf); free(sorted);                                                                   \
        }                                                                                                       \
        for (int64_t i = 0; i < m; ++i) {                                                                       \
            double t = 0.0;                                                                                     \
            for (int q = 0; q < nt; ++q) t += gacc[(size_t)q * (size_t)m + (size_t)i];                          \
            grad_out[i] = (T)t;                                                                                 \
        }                                                                                                       \
        scal_out[0] = 0.0; scal_out[1] = 0.0;                                                                   \
        for (int q = 0; q < nt; ++q) { scal_out[0] += sacc[2 * q]; scal_out[1] += sacc[2 * q + 1]; }            \
        free(scaled); free(gacc); free(sacc);                                                                   \
        return bad ? 2 : 0;                                                                                     \
    }                                                                                                           \
    /* Dense [L x K] row-major block, one column per k: restates ProjectionOperator.__call__. */               \
    int oracle_project_dense_##SUF(int64_t L, int64_t K, const T* in, T* out, const oracle_proj_t* p) {        \
        T* v = (T*)malloc((size_t)(L > 0 ? L : 1) * sizeof(T));                                                 \
        T* buf = (T*)malloc((size_t)(L > 0 ? L : 1) * sizeof(T));                                               \
        T* sorted = (T*)malloc((size_t)(L > 0 ? L : 1) * sizeof(T));                                            \
        oracle_proj_t q = *p;                                                                                   \
        q.lblock = (int32_t)L;                                                                                  \
        for (int64_t k = 0; k < K; ++k) {                                                                       \
            for (int64_t i = 0; i < L; ++i) v[i] = in[i * K + k];                                               \
            project_column_##SUF(v, L, &q, buf, sorted);                                                        \
            for (int64_t i = 0; i < L; ++i) out[i * K + k] = v[i];                                              \
        }                                                                                                       \
        free(v); free(buf); free(sorted);                                                                       \
        return 0;                                                                                               \
    }

static int oracle_tid(void) {
#ifdef _OPENMP
    return omp_get_thread_num();
#else
    return 0;
#endif
}

DEFINE_ORACLE(float, f32)
DEFINE_ORACLE(double, f64)

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
