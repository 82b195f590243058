// agd_kernels.hip -- the m-sized side of the dual ascent on gfx950: dual epilogue, accelerated-gradient step with the
// Lipschitz-history step size (device resident, no host round trip), dense-block projection operator and
// Jacobi row scaling.
#include "agd_step.h"
#include "comm.h"
#include <algorithm>
#include <cstring>
#include "wave.h"

namespace dl {

constexpr int kAgdThreads = 1024;

template <class T>
__device__ __forceinline__ T rnd(double v) { return (T)v; }

// block-wide sum / max of doubles, result in every thread
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = wave_allreduce(v, OpAdd());
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < NT / 64; ++w) t += sh[w];
    return t;
}
template <int NT>
__device__ __forceinline__ double block_max(double v, double* sh) {
    v = wave_allreduce(v, OpMax());
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = sh[0];
    for (int w = 1; w < NT / 64; ++w) t = sh[w] > t ? sh[w] : t;
    return t;
}

// ---------------------------------------------------------------------------------------------------------
// dual epilogue (matching.py:25-34, 164-178)
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(kAgdThreads) void dual_epilogue_kernel(int64_t m, const double* __restrict__ packed, const T* __restrict__ b,
                                                                    const T* __restrict__ lam, double gamma, T* __restrict__ grad_out,
                                                                    double* __restrict__ scal_out) {
    __shared__ double sh[kAgdThreads / 64];
    double dvtg = 0.0, gmax = -INFINITY, spos = 0.0;
    for (int64_t j = threadIdx.x; j < m; j += kAgdThreads) {
        const T gj = (T)((T)packed[j] - b[j]);
        grad_out[j] = gj;
        dvtg += (double)(T)(lam[j] * gj);
        gmax = (double)gj > gmax ? (double)gj : gmax;
        spos += gj > (T)0 ? (double)gj : 0.0;
    }
    dvtg = block_sum<kAgdThreads>(dvtg, sh);
    gmax = block_max<kAgdThreads>(gmax, sh);
    spos = block_sum<kAgdThreads>(spos, sh);
    if (threadIdx.x == 0) {
        const T nrm = (T)sqrt(packed[m + 1]);
        const T reg = (T)((T)(gamma / 2.0) * (T)(nrm * nrm));  // (gamma/2) * norm(x)**2, matching.py:157
        const T obj0 = (T)packed[m];
        const T dv = (T)dvtg;
        const T obj = (T)((T)(obj0 + reg) + dv);               // matching.py:33
        scal_out[0] = (double)obj;
        scal_out[1] = (double)reg;
        scal_out[2] = (double)obj0;
        scal_out[3] = (double)dv;
        scal_out[4] = (m > 0 && gmax > 0.0) ? (double)(T)gmax : 0.0;  // builtins.max(max(grad), 0), matching.py:168
        scal_out[5] = (double)(T)spos;
    }
}

// ---------------------------------------------------------------------------------------------------------
// one AGD iteration (agd.py:163-187; agd_utils.py:12-89) as two small multi-workgroup launches:
//   stats    rows in parallel: [sum the integer gradient slabs ->] g = A x - b, per-workgroup partial reductions
//   apply    every workgroup: sums the partials in a fixed order and runs the scalar step-size logic (workgroup 0 logs and
//            writes the next state), then its rows' projected ascent step + momentum            (agd_apply_kernel below)
// (a single-workgroup kernel doing all of it took 37 us at m = 10^4: it is latency bound on one CU)
// ---------------------------------------------------------------------------------------------------------
constexpr int kStatRows = 64;      // rows per stats workgroup
constexpr int kStatSlices = 16;    // slab slices per stats workgroup (1024 threads)

template <class T>
struct StatsArgs {
    int64_t m;
    // source of A x: either the integer slabs of the fused pass ...
    const long long* __restrict__ partial;
    const long long* __restrict__ partial_scal;  // fixed point, exponent shift_in[1]
    const int* __restrict__ shift_in;
    int n_slabs, n_scal;
    int slab32;  // the slabs are int32 low words, high words of the workgroups stamped with this launch's epoch (common.h: dl_matching::slab32)
    const int32_t* __restrict__ slab_hi;
    const unsigned long long* __restrict__ slab_ovf;
    unsigned long long slab_epoch;
    const uint8_t* __restrict__ slab_wide;  // [mpad] 1 = a WIDE row: its high words come from every slab in every launch (common.h), or null
    int64_t mpad;
    // hot-rows plan of the matching handle (inv != null): slab column p is the caller's row inv[p]; columns >= m_hot are in `cold`
    const int32_t* __restrict__ inv;
    int64_t m_hot;
    const long long* cold;
    long long* cold_zero;  // == cold when the step should leave the accumulators zeroed for the next fused launch, else null
    const double* dense;   // fairness pair of the matching handle: (A x) of rows m-2, m-1, or null
    // ... or already reduced (and, when sharded over RCCL, all-reduced) packed buffers, one per block of a split shard ...
    const double* packed_in[4];
    int n_packed;
    // ... or this rank's mailbox of the P2P exchange (comm.h): the kernel waits for every rank's slot and adds them in rank order
    MailArgs mail;
    unsigned long long* chk_partial;  // [gridDim.x + 2]: hash sums of what each block read from the mailbox, of the two scalars, and the announced total
    double scale;                     // developer aid (dl_comm_set_emulation): factor on the exchanged sums
    double* packed_out;               // [m+2]: A x and the two scalars as this step used them (kept for logging / callers; may alias packed_in[0])
    const T* __restrict__ b;
    const T* __restrict__ x;
    const T* __restrict__ y;
    const T* __restrict__ y_prev;  // dual stored with the previous history entry
    const T* __restrict__ g_old;
    T* __restrict__ g_new;
    const AgdDevState* st;
    double* __restrict__ partial_stats;  // [gridDim.x][kStatCols]
};

// SRC: 0 = packed buffers, 1 = the matching handle's integer slabs, 2 = the P2P mailbox
template <class T, int SRC>
__global__ __launch_bounds__(kStatRows* kStatSlices) void agd_stats_kernel(StatsArgs<T> p) {
    constexpr bool FROM_SLABS = SRC == 1;
    __shared__ long long shi[kStatRows * kStatSlices];
    const int tid = threadIdx.x;
    unsigned long long chk_expected = 0ull, chk_read = 0ull;
    if constexpr (SRC == 2) chk_expected = mail_wait(p.mail);
    const int rl = tid & (kStatRows - 1);
    const int ws = tid / kStatRows;
    const int64_t col = (int64_t)blockIdx.x * kStatRows + rl;  // slab column
    const bool live = col < p.m;
    int64_t row = col;                                           // the caller's row
    if constexpr (FROM_SLABS) {
        if (p.inv && live) row = p.inv[col];
    }
    double ax = 0.0;
    if constexpr (FROM_SLABS) {
        long long acc = 0;
        {   // latency bound: sixteen slabs in flight before the first is added (slabs past the end re-read the last one)
            const int64_t rc = live ? col : p.m - 1;
            const bool in_slabs = !p.inv || rc < p.m_hot;
            if (!in_slabs && ws == 0) {
                long long cv[kColdCopies];  // (one copy per XCD, fused_common.h: all eight loads in flight together)
#pragma unroll
                for (int k = 0; k < kColdCopies; ++k) cv[k] = p.cold[(int64_t)k * p.mpad + rc];
#pragma unroll
                for (int k = 0; k < kColdCopies; ++k) acc += cv[k];
                if (p.cold_zero && live) {  // consumed: ready for the next fused launch (no memset launch)
#pragma unroll
                    for (int k = 0; k < kColdCopies; ++k) p.cold_zero[(int64_t)k * p.mpad + rc] = 0;
                }
            }
            constexpr int kU = 16;  // (256 slabs / 16 slices: every load of a thread in flight at once)
            auto sum_slabs_into = [&](auto* slabs, long long& into) {
                for (int w0 = ws; in_slabs && w0 < p.n_slabs; w0 += kStatSlices * kU) {
                    long long v[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int w = w0 + kStatSlices * u;
                        v[u] = (long long)slabs[(int64_t)(w < p.n_slabs ? w : p.n_slabs - 1) * p.mpad + rc];
                    }
#pragma unroll
                    for (int u = 0; u < kU; ++u) into += (w0 + kStatSlices * u < p.n_slabs) ? v[u] : 0ll;
                }
            };
            auto sum_slabs = [&](auto* slabs) { sum_slabs_into(slabs, acc); };
            if (p.slab32) {
                // (the wide flag is requested FIRST and the low words unconditionally behind it: the flag is back before they are, so a wide row's
                //  high words are on their way while the low words still travel -- read behind them they were one more dependent trip for the block)
                const bool is_wide = in_slabs && p.slab_wide && p.slab_wide[rc];
                const int32_t* lo32 = reinterpret_cast<const int32_t*>(p.partial);
                for (int w0 = ws; in_slabs && w0 < p.n_slabs; w0 += kStatSlices * kU) {
                    long long v[kU], hw[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int w = w0 + kStatSlices * u;
                        v[u] = (long long)lo32[(int64_t)(w < p.n_slabs ? w : p.n_slabs - 1) * p.mpad + rc];
                    }
                    if (is_wide) {
#pragma unroll
                        for (int u = 0; u < kU; ++u) {
                            const int w = w0 + kStatSlices * u;
                            hw[u] = (long long)p.slab_hi[(int64_t)(w < p.n_slabs ? w : p.n_slabs - 1) * p.mpad + rc];
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < kU; ++u) hw[u] = 0ll;
                    }
#pragma unroll
                    for (int u = 0; u < kU; ++u) acc += (w0 + kStatSlices * u < p.n_slabs) ? v[u] + hw[u] * 4294967296ll : 0ll;
                }
                if (!is_wide && in_slabs && p.slab_ovf[p.n_slabs] == p.slab_epoch) {  // (uniform, rare: fused_common.h, epilogue)
                    for (int w0 = ws; w0 < p.n_slabs; w0 += kStatSlices)
                        if (p.slab_ovf[w0] == p.slab_epoch) acc += (long long)p.slab_hi[(int64_t)w0 * p.mpad + rc] * 4294967296ll;
                }
            } else {
                sum_slabs(p.partial);
            }
        }
        shi[tid] = acc;
        __syncthreads();
        if (ws == 0 && live) {
            long long t = shi[rl];
            for (int q = 1; q < kStatSlices; ++q) t += shi[q * kStatRows + rl];
            ax = ldexp((double)t, -(*p.shift_in));
            if (p.dense && row >= p.m - 2) ax = p.dense[row - (p.m - 2)];  // fairness pair of the matching handle: the two dense rows
            p.packed_out[row] = ax;
        }
    } else {
        if (ws == 0 && live) {
            if constexpr (SRC == 2) {
                ax = mail_sum(p.mail, row, chk_read) * p.scale;
            } else {
                ax = p.packed_in[0][row];
                for (int k = 1; k < p.n_packed; ++k) ax += p.packed_in[k][row];
                ax *= p.scale;
            }
            p.packed_out[row] = ax;
        }
        if (blockIdx.x == 0 && tid == kStatRows) {  // the two scalars (c.x, sum x^2), by a thread without a row
            unsigned long long chk_scal = 0ull;
            for (int64_t i = p.m; i < p.m + 2; ++i) {
                double v;
                if constexpr (SRC == 2) {
                    v = mail_sum(p.mail, i, chk_scal);
                } else {
                    v = p.packed_in[0][i];
                    for (int k = 1; k < p.n_packed; ++k) v += p.packed_in[k][i];
                }
                p.packed_out[i] = v * p.scale;
            }
            if constexpr (SRC == 2) {
                p.chk_partial[gridDim.x] = chk_scal;
                p.chk_partial[gridDim.x + 1] = chk_expected;
            }
        }
    }
    const bool has_prev = p.st->steps_done > 0;
    double dvtg = 0.0, gmax = -INFINITY, spos = 0.0, g2 = 0.0, dg2 = 0.0, dy2 = 0.0;
    if (ws == 0 && live) {
        const T gj = (T)((T)ax - p.b[row]);
        dvtg = (double)(T)(p.x[row] * gj);
        gmax = (double)gj;
        spos = gj > (T)0 ? (double)gj : 0.0;
        g2 = (double)gj * (double)gj;
        if (has_prev) {
            const T dg = (T)(p.g_old[row] - gj);
            const T dy = (T)(p.y_prev[row] - p.y[row]);
            dg2 = (double)dg * (double)dg;
            dy2 = (double)dy * (double)dy;
        }
        p.g_new[row] = gj;
    }
    // wave 0 holds all 64 rows of the block (ws == 0 <=> tid < 64)
    if (tid < 64) {
        dvtg = wave_allreduce_dpp(dvtg, OpAdd());
        gmax = wave_allreduce_dpp(gmax, OpMax());
        spos = wave_allreduce_dpp(spos, OpAdd());
        g2 = wave_allreduce_dpp(g2, OpAdd());
        dg2 = wave_allreduce_dpp(dg2, OpAdd());
        dy2 = wave_allreduce_dpp(dy2, OpAdd());
        if constexpr (SRC == 2) {  // what this block read from the mailbox, hashed: the step that consumes the statistics adds the blocks up
            chk_read = chk_wave_sum(chk_read);  // and compares with the announced total (agd_step.h)
            if (tid == 0) p.chk_partial[blockIdx.x] = chk_read;
        }
        if (tid == 0) {
            double* o = p.partial_stats + (int64_t)blockIdx.x * kStatCols;
            o[0] = dvtg;
            o[1] = gmax;
            o[2] = spos;
            o[3] = g2;
            o[4] = dg2;
            o[5] = dy2;
        }
    }
    if (FROM_SLABS && blockIdx.x + 1 == gridDim.x) {  // the extra last block: scalar partial sums of the fused pass (c.x, sum x^2)
        long long o = 0, q = 0;  // exact integer sums of the workgroups' fixed-point partials
        for (int w = tid; w < p.n_scal; w += kStatRows * kStatSlices) {
            o += p.partial_scal[2 * w];
            q += p.partial_scal[2 * w + 1];
        }
        o = wave_allreduce(o, OpAdd());
        q = wave_allreduce(q, OpAdd());
        __shared__ long long so[kStatRows * kStatSlices / 64], sq[kStatRows * kStatSlices / 64];
        if ((tid & 63) == 0) {
            so[tid >> 6] = o;
            sq[tid >> 6] = q;
        }
        __syncthreads();
        if (tid == 0) {
            long long oo = 0, qq = 0;
            for (int w = 0; w < kStatRows * kStatSlices / 64; ++w) {
                oo += so[w];
                qq += sq[w];
            }
            p.packed_out[p.m] = ldexp((double)oo, -p.shift_in[1]);
            p.packed_out[p.m + 1] = ldexp((double)qq, -p.shift_in[1]);
        }
    }
}

// ---- finalize + update in one launch ----
// Every workgroup derives the SAME scalars (objective pieces, Lipschitz estimate, step) from the stats partials, in the
// same order, then updates its own 1024 rows (agd_step.h holds the arithmetic: the fused pass of the NEXT iteration can run it in
// its prologue instead, saving this launch).  The optimizer state is double buffered: all workgroups read st_in, workgroup 0
// writes st_out.
template <class T>
__global__ __launch_bounds__(kApplyThreads) void agd_apply_kernel(ApplyArgs<T> p) {
    const int tid = threadIdx.x;
    const int64_t mine = (int64_t)blockIdx.x * kApplyThreads + tid;
    // this thread's row is requested BEFORE the scalars are derived (two dependent memory latencies -> one; the row does not depend on the step)
    const int64_t ic = mine < p.m ? mine : (p.m > 0 ? p.m - 1 : 0);
    T rx = (T)0, rg = (T)0, ry = (T)0;
    bool req = false;
    if (p.m > 0) {
        rx = p.x[ic];
        rg = p.g_new[ic];
        ry = p.y[ic];
        req = p.eq_mask && p.eq_mask[ic];
    }
    const float bt = p.beta[p.iter - 1];
    const double step = agd_step_scalars(p, tid & 63, blockIdx.x == 0, tid);
    if (mine < p.m) {
        T yn, xn;
        agd_update_values(rx, rg, ry, req, (T)step, (T)bt, (T)(float)(1.0f - bt), yn, xn);
        p.y_new[mine] = yn;
        p.x_next[mine] = xn;
        if (p.x_perm) p.x_perm[p.perm[mine]] = xn;  // hot-rows plan of the matching handle: next launch's dual vector, renumbered
    }
}

// ---------------------------------------------------------------------------------------------------------
// ProjectionOperator.__call__ on a dense [L x K] block: one column per thread (coalesced across columns)
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ void project_dense_kernel(int64_t L, int64_t K, const T* __restrict__ in, T* __restrict__ out, int kind, T p0, T p1, T ztol) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    if (kind != DL_PROJ_SIMPLEX && kind != DL_PROJ_SIMPLEX_EQ) {
        for (int64_t i = 0; i < L; ++i) {
            T v = in[i * K + k];
            switch (kind) {
                case DL_PROJ_BOX: v = v < p0 ? p0 : v; v = v > p1 ? p1 : v; break;
                case DL_PROJ_CONE_LOWER: v = v < p0 ? p0 : v; break;
                case DL_PROJ_CONE_UPPER: v = v > p0 ? p0 : v; break;
                default: break;
            }
            out[i * K + k] = v;
        }
        return;
    }
    const T z = p0;
    T S = (T)0, v1 = (T)(-INFINITY);
    for (int64_t i = 0; i < L; ++i) {
        T u = in[i * K + k];
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
            for (int64_t i = 0; i < L; ++i) {
                T u = in[i * K + k];
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
            th = (T)((T)(sumA - z) / (T)cnt);
            cnt_prev = cnt;
        }
    }
    for (int64_t i = 0; i < L; ++i) {
        T u = in[i * K + k];
        u = u > (T)0 ? u : (T)0;
        T x = u;
        if (projected) {
            if (onehot) x = u > th ? z : (T)0;
            else {
                const T d = (T)(u - th);
                x = d > (T)0 ? d : (T)0;
            }
        }
        out[i * K + k] = x;
    }
}

// The reference's OTHER simplex method (method="bisection_search", simplex.py:6-123), restated step by step -- it is not
// the Euclidean projection in every case, and callers that select it get its behaviour, not a substitute:
//   * "simplex" only: a column with sum(x) <= z + 1e-6 and every x >= -1e-6 is returned AS IS (small negatives included);
//   * L > 1: if the two largest entries of x / z differ by more than 1, the result is z at the argmax, 0 elsewhere;
//   * otherwise nu is bisected on [-1, 0] for sum(max(x - max(x / z) - nu, 0)) = 1 (the shift uses the maximum of the
//     NORMALISED column, as the reference does) by 19 halvings, nu* = bracket midpoint,
//     w = max(x - max(x / z) - nu*, 0) * z.
template <class T>
__global__ void project_dense_bisect_kernel(int64_t L, int64_t K, const T* __restrict__ in, T* __restrict__ out, int kind, T z) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const T tol = (T)1e-6;
    if (kind == DL_PROJ_SIMPLEX) {
        T sum = (T)0;
        bool nonneg = true;
        for (int64_t i = 0; i < L; ++i) {
            const T x = in[i * K + k];
            sum = (T)(sum + x);
            nonneg = nonneg && x >= -tol;
        }
        if (sum <= (T)(z + tol) && nonneg) {
            for (int64_t i = 0; i < L; ++i) out[i * K + k] = in[i * K + k];
            return;
        }
    }
    T m1 = (T)(-INFINITY), m2 = (T)(-INFINITY);  // two largest entries of x / z
    int64_t arg = 0;
    for (int64_t i = 0; i < L; ++i) {
        const T xn = (T)(in[i * K + k] / z);
        if (xn > m1) {
            m2 = m1;
            m1 = xn;
            arg = i;
        } else if (xn > m2) {
            m2 = xn;
        }
    }
    if (L > 1 && (T)(m1 - m2) > (T)1) {
        for (int64_t i = 0; i < L; ++i) out[i * K + k] = i == arg ? z : (T)0;
        return;
    }
    // 19 halvings: the reference stops the whole batch when consecutive midpoints differ by less than 1e-6, i.e. at the start of
    // its 20th pass (brackets are 2^-k wide for every column, so that test is the same for all of them)
    T lo = (T)(-1), hi = (T)0;
    for (int it = 0; it < 19; ++it) {
        const T mid = (T)((T)(lo + hi) / (T)2);
        T S = (T)0;
        for (int64_t i = 0; i < L; ++i) {
            const T d = (T)((T)(in[i * K + k] - m1) - mid);
            S = (T)(S + (d > (T)0 ? d : (T)0));
        }
        if (S > (T)1) lo = mid;
        else hi = mid;
    }
    const T nu = (T)((T)(lo + hi) / (T)2);
    for (int64_t i = 0; i < L; ++i) {
        const T d = (T)((T)(in[i * K + k] - m1) - nu);
        out[i * K + k] = (T)((d > (T)0 ? d : (T)0) * z);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Jacobi row scaling (preprocessing/precondition.py:8-28, sparse_utils.py:429-450)
// ---------------------------------------------------------------------------------------------------------
template <class T, class IdxT>
__global__ void row_sumsq_kernel(int64_t nnz, const IdxT* __restrict__ rowidx, const T* __restrict__ a, double* __restrict__ acc) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        const T v = a[k];
        atomicAdd(&acc[(int64_t)rowidx[k]], (double)(T)(v * v));
    }
}
template <class T>
__global__ void row_norm_finish_kernel(int64_t m, const double* __restrict__ acc, T* __restrict__ norms, T* __restrict__ b) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const T nr = (T)sqrt((double)(T)acc[i]);
    norms[i] = nr;
    b[i] = (T)(b[i] * (T)((T)1 / nr));
}
template <class T, class IdxT>
__global__ void row_scale_kernel(int64_t nnz, const IdxT* __restrict__ rowidx, T* __restrict__ a, const T* __restrict__ norms) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        a[k] = (T)(a[k] * (T)((T)1 / norms[(int64_t)rowidx[k]]));
    }
}

// ---------------------------------------------------------------------------------------------------------
// host-side launchers used by api.hip
// ---------------------------------------------------------------------------------------------------------
int launch_epilogue(int64_t m, int val_dtype, const double* packed, const void* b, const void* lam, double gamma, void* grad_out,
                    double* scal_out, hipStream_t st) {
    if (val_dtype == DL_F32)
        hipLaunchKernelGGL(dual_epilogue_kernel<float>, dim3(1), dim3(kAgdThreads), 0, st, m, packed, (const float*)b, (const float*)lam, gamma,
                           (float*)grad_out, scal_out);
    else
        hipLaunchKernelGGL(dual_epilogue_kernel<double>, dim3(1), dim3(kAgdThreads), 0, st, m, packed, (const double*)b, (const double*)lam, gamma,
                           (double*)grad_out, scal_out);
    DL_HIP(hipGetLastError());
    return 0;
}

size_t agd_state_bytes() { return 2 * sizeof(AgdDevState); }  // double buffered

int agd_state_init(void* dev_state, double initial_step, double max_step, hipStream_t st) {
    AgdDevState h;
    memset(&h, 0, sizeof(h));
    h.max_step = max_step;
    h.initial_step = initial_step;
    h.last_step = initial_step;
    DL_HIP(hipMemcpyAsync(dev_state, &h, sizeof(h), hipMemcpyHostToDevice, st));
    DL_HIP(hipMemcpyAsync((AgdDevState*)dev_state + 1, &h, sizeof(h), hipMemcpyHostToDevice, st));
    DL_HIP(hipStreamSynchronize(st));  // h is a stack object
    return 0;
}

int agd_state_read_max_step(void* dev_state, int cur, double* out, hipStream_t st) {
    AgdDevState h;
    DL_HIP(hipMemcpyAsync(&h, (AgdDevState*)dev_state + cur, sizeof(h), hipMemcpyDeviceToHost, st));
    DL_HIP(hipStreamSynchronize(st));
    *out = h.max_step;
    return 0;
}

template <class T>
static int agd_stats_typed(dl_agd* s, const StepSource& src, const void* b, hipStream_t st) {
    const int n_blocks = (int)((s->m + kStatRows - 1) / kStatRows);
    AgdDevState* states = (AgdDevState*)s->state;
    const AgdDevState* st_in = states + s->state_cur;
    dl_matching* f = src.slabs;
    if (n_blocks > 0) {
        StatsArgs<T> sa;
        memset(&sa, 0, sizeof(sa));
        sa.m = s->m;
        sa.partial = f ? static_cast<const long long*>(f->partial) : nullptr;
        sa.partial_scal = f ? f->partial_scal : nullptr;
        sa.shift_in = f ? f->shift_dev : nullptr;
        sa.n_slabs = f ? (f->grad_lds ? f->n_wg : 1) : 0;
        sa.n_scal = f ? f->n_wg : 0;
        sa.slab32 = (f && f->slab32) ? 1 : 0;
        sa.slab_hi = f ? f->slab_hi : nullptr;
        sa.slab_ovf = f ? f->slab_ovf : nullptr;
        sa.slab_wide = (f && f->slab32 && f->n_wide > 0) ? f->slab_wide : nullptr;
        sa.slab_epoch = f ? f->slab_epoch : 0;
        sa.mpad = f ? f->mpad : 0;
        sa.inv = (f && f->m_hot > 0) ? f->row_inv : nullptr;
        sa.m_hot = f ? f->m_hot : 0;
        sa.cold = f ? f->cold_grad : nullptr;
        sa.cold_zero = (f && f->m_hot > 0) ? f->cold_grad : nullptr;
        sa.dense = (f && f->fair) ? f->dense_ax : nullptr;
        for (int k = 0; k < 4; ++k) sa.packed_in[k] = k < src.n_packed ? src.packed[k] : nullptr;
        sa.n_packed = src.n_packed;
        if (src.mail) sa.mail = *src.mail;
        sa.chk_partial = s->chk_partial;
        s->chk_dead = src.mail ? src.mail->dead : nullptr;  // (the step that consumes these statistics verifies the exchange)
        sa.scale = src.scale;
        sa.packed_out = s->packed;
        sa.b = (const T*)b;
        sa.x = (const T*)s->x;
        sa.y = (const T*)s->y;
        sa.y_prev = (const T*)s->y_old;
        sa.g_old = (const T*)s->g;
        sa.g_new = (T*)s->g_old;
        sa.st = st_in;
        sa.partial_stats = s->partial_stats;
        if (f) hipLaunchKernelGGL((agd_stats_kernel<T, 1>), dim3(n_blocks + 1), dim3(kStatRows * kStatSlices), 0, st, sa);  // + the scalar-sum block
        else if (src.mail) hipLaunchKernelGGL((agd_stats_kernel<T, 2>), dim3(n_blocks), dim3(kStatRows * kStatSlices), 0, st, sa);
        else hipLaunchKernelGGL((agd_stats_kernel<T, 0>), dim3(n_blocks), dim3(kStatRows * kStatSlices), 0, st, sa);
        DL_HIP(hipGetLastError());
    }
    return 0;
}

template <class T>
static int agd_apply_typed(dl_agd* s, const StepSource& src, const PendingStep& ps, hipStream_t st) {
    dl_matching* f = src.slabs;
    dl_matching* hot = (src.hot && src.hot->m_hot > 0) ? src.hot : nullptr;
    ApplyArgs<T> aa = make_apply_args<T>(s, ps);
    aa.x_perm = hot ? (T*)hot->lam_perm : nullptr;
    aa.perm = hot ? hot->row_perm : nullptr;
    const unsigned grid = (unsigned)std::max<int64_t>(1, (s->m + kApplyThreads - 1) / kApplyThreads);
    hipLaunchKernelGGL(agd_apply_kernel<T>, dim3(grid), dim3(kApplyThreads), 0, st, aa);
    DL_HIP(hipGetLastError());
    agd_rotate(s);
    if (hot && f == hot) {  // (the slab route also re-zeroed the cold accumulators: the next fused launch needs no preparation at all)
        hot->hot_ready = true;
        hot->hot_ready_lambda = s->x;
        hot->hot_ready_owner = s->uid;
    }
    return 0;
}

// The first half of a step: g = A x - b and the per-workgroup statistics.  Where A x comes from: the matching handle's integer
// slabs (single-device loop, no separate slab reduction), reduced packed buffers (all-reduced when sharded over RCCL; one per
// block of a split shard), or the P2P mailbox.
int launch_agd_stats(dl_agd* s, const StepSource& src, const void* b, hipStream_t st) {
    if (s->val_dtype == DL_F32) return agd_stats_typed<float>(s, src, b, st);
    return agd_stats_typed<double>(s, src, b, st);
}
// The second half as its own launch (step size, projected ascent, momentum, log row, next state) + buffer rotation.  The loops
// of api.hip hand this half to the NEXT fused launch's prologue instead whenever they can (agd_step.h).
int launch_agd_apply(dl_agd* s, const StepSource& src, const PendingStep& ps, hipStream_t st) {
    if (s->val_dtype == DL_F32) return agd_apply_typed<float>(s, src, ps, st);
    return agd_apply_typed<double>(s, src, ps, st);
}
// scalars (c.x, sum x^2) the step of `src` reads: the stats launch leaves them in s->packed
const double* agd_step_scal(const dl_agd* s, const StepSource& src) {
    const int n_blocks = (int)((s->m + kStatRows - 1) / kStatRows);
    return (n_blocks > 0 || !src.n_packed) ? s->packed + s->m : src.packed[0] + s->m;
}

int launch_agd_step(dl_agd* s, const StepSource& src, const void* b, double gamma, int64_t iter, int decay_now, double decay_factor, hipStream_t st) {
    int rc = launch_agd_stats(s, src, b, st);
    if (rc) return rc;
    PendingStep ps;
    ps.valid = true;
    ps.gamma = gamma;
    ps.iter = iter;
    ps.decay_now = decay_now;
    ps.decay_factor = decay_factor;
    ps.scal = agd_step_scal(s, src);
    return launch_agd_apply(s, src, ps, st);
}

size_t agd_partial_stats_bytes(int64_t m) { return sizeof(double) * kStatCols * (size_t)((m + kStatRows - 1) / kStatRows + 1); }

int launch_project_dense(int64_t L, int64_t K, int val_dtype, const void* in, void* out, const dl_proj_desc* p, hipStream_t st) {
    if (K <= 0 || L <= 0) return 0;
    const int threads = 256;
    const int blocks = (int)((K + threads - 1) / threads);
    if ((p->flags & DL_PROJ_FLAG_BISECTION) && (p->kind == DL_PROJ_SIMPLEX || p->kind == DL_PROJ_SIMPLEX_EQ)) {
        if (val_dtype == DL_F32) hipLaunchKernelGGL(project_dense_bisect_kernel<float>, dim3(blocks), dim3(threads), 0, st, L, K, (const float*)in, (float*)out, p->kind, (float)p->p0);
        else hipLaunchKernelGGL(project_dense_bisect_kernel<double>, dim3(blocks), dim3(threads), 0, st, L, K, (const double*)in, (double*)out, p->kind, p->p0);
        DL_HIP(hipGetLastError());
        return 0;
    }
    if (val_dtype == DL_F32)
        hipLaunchKernelGGL(project_dense_kernel<float>, dim3(blocks), dim3(threads), 0, st, L, K, (const float*)in, (float*)out, p->kind, (float)p->p0,
                           (float)p->p1, (float)(p->p0 + 1e-6));
    else
        hipLaunchKernelGGL(project_dense_kernel<double>, dim3(blocks), dim3(threads), 0, st, L, K, (const double*)in, (double*)out, p->kind, p->p0, p->p1,
                           p->p0 + 1e-6);
    DL_HIP(hipGetLastError());
    return 0;
}

template <class T, class IdxT>
static int jacobi_typed(int64_t m, int64_t nnz, const void* rowidx, void* a, void* b, void* norms, hipStream_t st) {
    double* acc = nullptr;
    DL_HIP(hipMalloc(&acc, sizeof(double) * (size_t)(m > 0 ? m : 1)));
    DL_HIP(hipMemsetAsync(acc, 0, sizeof(double) * (size_t)(m > 0 ? m : 1), st));
    const int threads = 256;
    int64_t blocks64 = (nnz + threads - 1) / threads;
    const int blocks = (int)(blocks64 > 4096 ? 4096 : (blocks64 > 0 ? blocks64 : 1));
    hipLaunchKernelGGL((row_sumsq_kernel<T, IdxT>), dim3(blocks), dim3(threads), 0, st, nnz, (const IdxT*)rowidx, (const T*)a, acc);
    hipLaunchKernelGGL(row_norm_finish_kernel<T>, dim3((int)((m + threads - 1) / threads > 0 ? (m + threads - 1) / threads : 1)), dim3(threads), 0, st, m, acc,
                       (T*)norms, (T*)b);
    hipLaunchKernelGGL((row_scale_kernel<T, IdxT>), dim3(blocks), dim3(threads), 0, st, nnz, (const IdxT*)rowidx, (T*)a, (const T*)norms);
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(acc);
    if (e != hipSuccess) return hip_fail(e, "jacobi launch");
    if (e2 != hipSuccess) return hip_fail(e2, "jacobi sync");
    return 0;
}

int launch_jacobi(int64_t m, int64_t nnz, const void* rowidx, int idx_dtype, void* a, void* b, void* norms, int val_dtype, hipStream_t st) {
    if (val_dtype == DL_F32) {
        return idx_dtype == DL_I32 ? jacobi_typed<float, int32_t>(m, nnz, rowidx, a, b, norms, st) : jacobi_typed<float, int64_t>(m, nnz, rowidx, a, b, norms, st);
    }
    return idx_dtype == DL_I32 ? jacobi_typed<double, int32_t>(m, nnz, rowidx, a, b, norms, st) : jacobi_typed<double, int64_t>(m, nnz, rowidx, a, b, norms, st);
}

}  // namespace dl
