// matching_kernels.hip -- the fused dual-gradient pass of the matching LP for gfx950.
//
// One launch streams the CSC arrays (a, c, row index) exactly once and performs, per non-zero / per column,
// what the reference does in ~10^2 ATen launches (src/dualip/objectives/matching.py:136-161):
//     gather  v = a * (-(1/g) lambda[row]) + (-(1/g) c)     left_multiply_sparse + elementwise_csc(add)
//     project x = Proj_column(v)                              apply_F_to_columns (box / cone / simplex)
//     scatter (A x)[row] += a x ;  c.x ;  sum x^2             row_sums_csc(A*x), dot, norm
//
// Mapping to the hardware
//   * one 1024-thread workgroup per CU (16 wavefronts); lambda (pre-scaled by -1/gamma) is staged in LDS and the
//     gradient is privatised in LDS (ds_add), so the only HBM traffic is the coalesced CSC stream;
//   * a wavefront owns a "tile": <= 64 non-zeros of whole consecutive columns, one non-zero per lane, described by
//     a 16-byte record (start, count, column-head bit mask, projection id) that replaces the column-pointer array;
//   * the simplex projection runs in registers: segmented DPP scans give per-column sum / max, a ballot gives the
//     support size, and a monotone Newton (Michelot) iteration on the piecewise-linear f(theta) = sum max(u-theta,0)
//     finds the exact threshold the reference obtains by sort + cumsum;
//   * columns longer than 64 non-zeros are walked by a whole wavefront in 64-wide strides (re-reading L2-hot data
//     per Newton pass);
//   * every workgroup writes its private gradient to its own slab; a second small kernel sums the slabs in double.
#include "common.h"
#include "wave.h"

namespace dl {

template <class T>
struct FusedArgs {
    const TileDesc* __restrict__ tiles;
    const uint32_t* __restrict__ wg_tile_begin;
    const void* __restrict__ rowidx;
    const T* __restrict__ a;
    const T* __restrict__ c;
    const T* __restrict__ lambda;
    T* __restrict__ x_out;
    const ProjDev* __restrict__ projs;
    T* __restrict__ partial;          // [n_wg][mpad] (GRAD_LDS) or [mpad] (global atomics, pre-zeroed)
    double* __restrict__ partial_scal;  // [n_wg][2]
    double gamma;
    int64_t m;
    int64_t mpad;
};

template <class T>
__device__ __forceinline__ T tmax(T a, T b) { return a > b ? a : b; }
template <class T>
__device__ __forceinline__ T tmin(T a, T b) { return a < b ? a : b; }

template <class T>
struct ProjT {
    int kind;
    T p0, p1;
    T ztol;  // (T)(z + 1e-6): the reference's feasibility slack (simplex.py:155)
};

template <class T>
__device__ __forceinline__ ProjT<T> load_proj(const ProjDev* __restrict__ projs, uint32_t id) {
    ProjT<T> p;
    if (id == kNoProj) {
        p.kind = DL_PROJ_NONE;
        p.p0 = p.p1 = p.ztol = (T)0;
        return p;
    }
    p.kind = projs[id].kind;
    p.p0 = (T)projs[id].p0;
    p.p1 = (T)projs[id].p1;
    p.ztol = (T)(projs[id].p0 + 1e-6);
    return p;
}

// element-wise operators (box.py:15-16, cone.py:21-28)
template <class T>
__device__ __forceinline__ T project_pointwise(T v, const ProjT<T>& p) {
    switch (p.kind) {
        case DL_PROJ_BOX: return tmin(tmax(v, p.p0), p.p1);
        case DL_PROJ_CONE_LOWER: return tmax(v, p.p0);
        case DL_PROJ_CONE_UPPER: return tmin(v, p.p0);
        default: return v;
    }
}

// Simplex projection of every column segment of a short tile, one value per lane.
// Equals _duchi_proj (simplex.py:126-236) column by column: clamp at 0; (inequality) keep if sum <= z + 1e-6;
// vertex z*e_argmax when only the maximum exceeds max - z (the reference's top-2 shortcut); else
// x = max(u - theta, 0) with theta = (sum of the support - z) / |support|.
template <bool USE_DPP, class T>
__device__ __forceinline__ T simplex_short(T v, bool valid, const SegInfo& s, T z, T ztol, bool equality) {
    const T u = valid ? tmax(v, (T)0) : (T)0;
    const T S = seg_allreduce<USE_DPP>(u, s, (T)0, OpAdd());
    bool act = valid && (equality || S > ztol);
    T x = u;
    if (__any(act)) {
        const T v1 = seg_allreduce<USE_DPP>(u, s, (T)(-INFINITY), OpMax());
        const int len = s.tail - s.start + 1;
        // two lower bounds of theta*: f(v1 - z) >= z and f((S - z)/len) >= z
        T th = tmax((T)(v1 - z), (T)((T)(S - z) / (T)len));
        int cnt_prev = 0;
        bool onehot = false;
        for (int it = 0; it < 2 * kTileLanes + 2; ++it) {
            const bool in = u > th;
            const uint64_t bal = __ballot(in && valid) & s.segmask;
            const int cnt = __popcll(bal);
            if (it == 0 && act && cnt == 1) {  // only the maximum survives max - z: vertex (simplex.py:177-193)
                onehot = true;
                act = false;
            }
            if (!__any(act)) break;
            const T sumA = seg_allreduce<USE_DPP>(in ? u : (T)0, s, (T)0, OpAdd());
            if (act) {
                if (cnt == cnt_prev || cnt == 0) {
                    act = false;  // support unchanged: th is the fixed point
                } else {
                    th = (T)((T)(sumA - z) / (T)cnt);
                    cnt_prev = cnt;
                }
            }
        }
        const bool projected = valid && (equality || S > ztol);
        if (projected) {
            if (onehot) x = (u > th) ? z : (T)0;
            else x = tmax((T)(u - th), (T)0);
        }
    }
    return x;
}

template <class T, class RowT>
__device__ __forceinline__ uint32_t load_row(const void* __restrict__ rowidx, uint64_t k) {
    return (uint32_t)((const RowT*)rowidx)[k];
}

template <class T>
__device__ __forceinline__ void lds_add(T* p, T v) {
    atomicAdd(p, v);  // ds_add_f32 / ds_add_f64 (no return)
}

// ---------------------------------------------------------------------------------------------------------
// fused kernel
// ---------------------------------------------------------------------------------------------------------
template <class T, class RowT, bool LAM_LDS, bool GRAD_LDS, bool USE_DPP>
__global__ __launch_bounds__(kFusedThreads) void matching_fused_kernel(FusedArgs<T> g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* lam_s = reinterpret_cast<T*>(smem);
    T* grad_s = reinterpret_cast<T*>(smem) + (LAM_LDS ? g.m : 0);
    double* red_s = reinterpret_cast<double*>(smem + ((size_t)((LAM_LDS ? g.m : 0) + (GRAD_LDS ? g.m : 0)) * sizeof(T) + 15) / 16 * 16);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const T s = (T)(-1.0 / g.gamma);  // matching.py:136: the scalar is formed in double, rounded once

    // ---- prologue: stage -lambda/gamma, zero the private gradient ----
    if constexpr (LAM_LDS) {
        for (int64_t i = tid; i < g.m; i += kFusedThreads) lam_s[i] = (T)(s * g.lambda[i]);
    }
    if constexpr (GRAD_LDS) {
        for (int64_t i = tid; i < g.m; i += kFusedThreads) grad_s[i] = (T)0;
    }
    __syncthreads();

    T* gacc = GRAD_LDS ? grad_s : g.partial;
    double obj = 0.0, ssq = 0.0;

    const uint32_t t_begin = g.wg_tile_begin[wg];
    const uint32_t t_end = g.wg_tile_begin[wg + 1];

    // software prefetch: the next tile's CSC values are in flight while the current tile is projected
    uint32_t t = t_begin + (uint32_t)wave;
    TileDesc d_cur = {0, 0};
    T a_cur = (T)0, c_cur = (T)0;
    uint32_t r_cur = 0;
    auto issue = [&](uint32_t tt, TileDesc& d, T& av, T& cv, uint32_t& rv) {
        d.w0 = g.tiles[tt].w0;
        d.w1 = g.tiles[tt].w1;
        av = (T)0;
        cv = (T)0;
        rv = 0;
        if (!(d.w0 & kTileLongFlag)) {
            const uint32_t cnt = tile_count(d.w0);
            if ((uint32_t)lane < cnt) {
                const uint64_t k = tile_nnz_start(d.w0) + (uint64_t)lane;
                av = g.a[k];
                cv = g.c[k];
                rv = load_row<T, RowT>(g.rowidx, k);
            }
        }
    };
    if (t < t_end) issue(t, d_cur, a_cur, c_cur, r_cur);

    while (t < t_end) {
        const uint32_t t_next = t + kFusedWaves;
        TileDesc d_nxt = {0, 0};
        T a_nxt = (T)0, c_nxt = (T)0;
        uint32_t r_nxt = 0;
        if (t_next < t_end) issue(t_next, d_nxt, a_nxt, c_nxt, r_nxt);

        const ProjT<T> pj = load_proj<T>(g.projs, tile_proj(d_cur.w0));
        if (!(d_cur.w0 & kTileLongFlag)) {
            // ------------------------------ short tile: one non-zero per lane ------------------------------
            const uint32_t cnt = tile_count(d_cur.w0);
            const bool valid = (uint32_t)lane < cnt;
            const T lam = LAM_LDS ? lam_s[r_cur] : (T)(s * g.lambda[r_cur]);
            T v = (T)(a_cur * lam);             // sparse_utils.py:79
            v = (T)(v + (T)(s * c_cur));        // matching.py:66,142
            T x;
            if (pj.kind == DL_PROJ_SIMPLEX || pj.kind == DL_PROJ_SIMPLEX_EQ) {
                const SegInfo sg = make_seginfo(d_cur.w1, lane);
                x = simplex_short<USE_DPP>(v, valid, sg, pj.p0, pj.ztol, pj.kind == DL_PROJ_SIMPLEX_EQ);
            } else {
                x = project_pointwise(v, pj);
            }
            if (valid) {
                const T ax = (T)(a_cur * x);
                if (ax != (T)0) {
                    if constexpr (GRAD_LDS) lds_add(&gacc[r_cur], ax);
                    else atomicAdd(&gacc[r_cur], ax);
                }
                obj += (double)(T)(c_cur * x);
                ssq += (double)(T)(x * x);
                if (g.x_out) g.x_out[tile_nnz_start(d_cur.w0) + (uint64_t)lane] = x;
            }
        } else {
            // ------------------------------ long tile: one column, 64-wide strides ------------------------------
            const uint64_t k0 = tile_nnz_start(d_cur.w0);
            const uint64_t len = d_cur.w1;
            const bool is_simplex = pj.kind == DL_PROJ_SIMPLEX || pj.kind == DL_PROJ_SIMPLEX_EQ;
            auto value_at = [&](uint64_t k, T& av, T& cv, uint32_t& rv) -> T {
                av = g.a[k];
                cv = g.c[k];
                rv = load_row<T, RowT>(g.rowidx, k);
                const T lam = LAM_LDS ? lam_s[rv] : (T)(s * g.lambda[rv]);
                T v = (T)(av * lam);
                return (T)(v + (T)(s * cv));
            };
            T th = (T)0;
            bool projected = false, onehot = false;
            if (is_simplex) {
                T S = (T)0, v1 = (T)(-INFINITY);
                for (uint64_t off = lane; off < len; off += 64) {
                    T av, cv;
                    uint32_t rv;
                    const T u = tmax(value_at(k0 + off, av, cv, rv), (T)0);
                    S = (T)(S + u);
                    v1 = tmax(v1, u);
                }
                S = wave_allreduce(S, OpAdd());
                v1 = wave_allreduce(v1, OpMax());
                projected = (pj.kind == DL_PROJ_SIMPLEX_EQ) || S > pj.ztol;
                if (projected) {
                    const T z = pj.p0;
                    th = tmax((T)(v1 - z), (T)((T)(S - z) / (T)len));
                    long long cnt_prev = 0;
                    for (int it = 0; it < 4096; ++it) {
                        T sumA = (T)0;
                        long long cntl = 0;
                        for (uint64_t off = lane; off < len; off += 64) {
                            T av, cv;
                            uint32_t rv;
                            const T u = tmax(value_at(k0 + off, av, cv, rv), (T)0);
                            if (u > th) {
                                sumA = (T)(sumA + u);
                                cntl += 1;
                            }
                        }
                        sumA = wave_allreduce(sumA, OpAdd());
                        double cd = wave_allreduce((double)cntl, OpAdd());
                        const long long cntw = (long long)cd;
                        if (it == 0 && cntw == 1) {
                            onehot = true;
                            break;
                        }
                        if (cntw == cnt_prev || cntw == 0) break;
                        th = (T)((T)(sumA - z) / (T)cntw);
                        cnt_prev = cntw;
                    }
                }
            }
            for (uint64_t off = lane; off < len; off += 64) {
                T av, cv;
                uint32_t rv;
                const T v = value_at(k0 + off, av, cv, rv);
                T x;
                if (is_simplex) {
                    const T u = tmax(v, (T)0);
                    if (!projected) x = u;
                    else if (onehot) x = (u > th) ? pj.p0 : (T)0;
                    else x = tmax((T)(u - th), (T)0);
                } else {
                    x = project_pointwise(v, pj);
                }
                const T ax = (T)(av * x);
                if (ax != (T)0) {
                    if constexpr (GRAD_LDS) lds_add(&gacc[rv], ax);
                    else atomicAdd(&gacc[rv], ax);
                }
                obj += (double)(T)(cv * x);
                ssq += (double)(T)(x * x);
                if (g.x_out) g.x_out[k0 + off] = x;
            }
        }

        t = t_next;
        d_cur = d_nxt;
        a_cur = a_nxt;
        c_cur = c_nxt;
        r_cur = r_nxt;
    }

    // ---- epilogue: scalar partials, then the private gradient slab ----
    obj = wave_allreduce(obj, OpAdd());
    ssq = wave_allreduce(ssq, OpAdd());
    if (lane == 0) {
        red_s[2 * wave] = obj;
        red_s[2 * wave + 1] = ssq;
    }
    __syncthreads();
    if (tid == 0) {
        double o = 0.0, q = 0.0;
        for (int w = 0; w < kFusedWaves; ++w) {
            o += red_s[2 * w];
            q += red_s[2 * w + 1];
        }
        g.partial_scal[2 * (int64_t)wg] = o;
        g.partial_scal[2 * (int64_t)wg + 1] = q;
    }
    if constexpr (GRAD_LDS) {
        T* slab = g.partial + (int64_t)wg * g.mpad;
        for (int64_t i = tid; i < g.m; i += kFusedThreads) slab[i] = grad_s[i];
    }
}

// ---------------------------------------------------------------------------------------------------------
// slab reduction: packed[0..m) = sum_w partial[w][i] (double), packed[m], packed[m+1] = scalar partial sums
// ---------------------------------------------------------------------------------------------------------
constexpr int kRedThreads = 256;
constexpr int kRedRows = 64;  // rows per block; 4 slab-slices per block

template <class T>
__global__ __launch_bounds__(kRedThreads) void reduce_partials_kernel(const T* __restrict__ partial, const double* __restrict__ partial_scal,
                                                                      int n_slabs, int n_scal, int64_t m, int64_t mpad, double* __restrict__ packed) {
    __shared__ double sh[kRedThreads];
    const int tid = threadIdx.x;
    const int rl = tid & (kRedRows - 1);
    const int ws = tid / kRedRows;
    const int64_t row = (int64_t)blockIdx.x * kRedRows + rl;
    double acc = 0.0;
    if (row < m) {
        for (int w = ws; w < n_slabs; w += kRedThreads / kRedRows) acc += (double)partial[(int64_t)w * mpad + row];
    }
    sh[tid] = acc;
    __syncthreads();
    if (ws == 0 && row < m) {
        double t = sh[rl];
        for (int q = 1; q < kRedThreads / kRedRows; ++q) t += sh[q * kRedRows + rl];
        packed[row] = t;
    }
    if (blockIdx.x == 0) {
        __syncthreads();
        double o = 0.0, q = 0.0;
        for (int w = tid; w < n_scal; w += kRedThreads) {
            o += partial_scal[2 * w];
            q += partial_scal[2 * w + 1];
        }
        o = wave_allreduce(o, OpAdd());
        q = wave_allreduce(q, OpAdd());
        if ((tid & 63) == 0) {
            sh[2 * (tid >> 6)] = o;
            sh[2 * (tid >> 6) + 1] = q;
        }
        __syncthreads();
        if (tid == 0) {
            double oo = 0.0, qq = 0.0;
            for (int w = 0; w < kRedThreads / 64; ++w) {
                oo += sh[2 * w];
                qq += sh[2 * w + 1];
            }
            packed[m] = oo;
            packed[m + 1] = qq;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
template <class T, class RowT, bool LAM, bool GRAD, bool DPP>
static int launch_fused_inst(const dl_matching* h, const FusedArgs<T>& args, hipStream_t st) {
    auto kern = matching_fused_kernel<T, RowT, LAM, GRAD, DPP>;
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        DL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(h->n_wg), dim3(kFusedThreads), h->lds_bytes, st, args);
    DL_HIP(hipGetLastError());
    return 0;
}

template <class T, class RowT>
static int launch_fused_rt(const dl_matching* h, const FusedArgs<T>& args, hipStream_t st) {
    const bool L = h->lam_lds, G = h->grad_lds, D = h->use_dpp;
    if (L && G) return D ? launch_fused_inst<T, RowT, true, true, true>(h, args, st) : launch_fused_inst<T, RowT, true, true, false>(h, args, st);
    if (!L && G) return D ? launch_fused_inst<T, RowT, false, true, true>(h, args, st) : launch_fused_inst<T, RowT, false, true, false>(h, args, st);
    return D ? launch_fused_inst<T, RowT, false, false, true>(h, args, st) : launch_fused_inst<T, RowT, false, false, false>(h, args, st);
}

template <class T>
static int calculate_typed(dl_matching* h, const void* lambda, double gamma, double* packed_out, void* x_out, hipStream_t st) {
    if (h->n_tiles == 0 || h->n_wg == 0) {  // no non-zeros at all: A x = 0
        DL_HIP(hipMemsetAsync(packed_out, 0, sizeof(double) * (size_t)(h->m + 2), st));
        return 0;
    }
    FusedArgs<T> args;
    args.tiles = h->tiles;
    args.wg_tile_begin = h->wg_tile_begin;
    args.rowidx = h->rowidx;
    args.a = static_cast<const T*>(h->a);
    args.c = static_cast<const T*>(h->c);
    args.lambda = static_cast<const T*>(lambda);
    args.x_out = static_cast<T*>(x_out);
    args.projs = h->projs;
    args.partial = static_cast<T*>(h->partial);
    args.partial_scal = h->partial_scal;
    args.gamma = gamma;
    args.m = h->m;
    args.mpad = h->mpad;
    if (!h->grad_lds) DL_HIP(hipMemsetAsync(h->partial, 0, sizeof(T) * (size_t)h->mpad, st));
    hipEvent_t ev_stop = nullptr;
    if (h->prof_on) {
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
    int rc = h->row_bytes == 2 ? launch_fused_rt<T, uint16_t>(h, args, st) : launch_fused_rt<T, uint32_t>(h, args, st);
    if (rc) return rc;
    if (ev_stop) DL_HIP(hipEventRecord(ev_stop, st));
    const int n_slabs = h->grad_lds ? h->n_wg : 1;
    const int blocks = (int)((h->m + kRedRows - 1) / kRedRows);
    hipLaunchKernelGGL(reduce_partials_kernel<T>, dim3(blocks > 0 ? blocks : 1), dim3(kRedThreads), 0, st, static_cast<const T*>(h->partial),
                       h->partial_scal, n_slabs, h->n_wg, h->m, h->mpad, packed_out);
    DL_HIP(hipGetLastError());
    return 0;
}

int matching_calculate(dl_matching* h, const void* lambda, double gamma, double* packed_out, void* x_out, hipStream_t st) {
    if (h->val_dtype == DL_F32) return calculate_typed<float>(h, lambda, gamma, packed_out, x_out, st);
    return calculate_typed<double>(h, lambda, gamma, packed_out, x_out, st);
}

}  // namespace dl
