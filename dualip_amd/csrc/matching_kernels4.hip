// matching_kernels4.hip -- the fused pass with 256-wide tiles: FOUR consecutive non-zeros per lane.
//
// Same computation and LDS plan as matching_kernels.hip; what changes is the tile:
//   * a tile is a 16-byte-aligned window of 256 non-zeros [W, W+256) holding whole consecutive columns in
//     [W+lo, W+hi); lane L owns elements 4L..4L+3, fetched with ONE 16-byte load per array (a, c) and one 8-byte
//     load of four uint16 row indices -- a quarter of the load instructions and descriptor traffic per non-zero;
//   * per-tile fixed costs (descriptor unpack, loop control) amortise over 4x the work, and the four slots of a lane
//     are independent dependency chains for the element-wise part;
//   * segment structure = four wave-uniform 64-bit head masks; all predicates are scalar mask arithmetic (simplex4.h).
// Descriptor: 12 dwords { W[39:0] | hi<<40 | lo<<49 | long<<51 ; H0 ; H1 ; H2 ; H3 ; proj id ; 0 } (long: length in H0).
// Columns that cannot sit in a window (longer than 253, touching the array's last partial quad, or using a projection
// entry beyond the LDS table) are single-column "long" tiles handled by process_long_tile.
#include "fused_common.h"
#include "simplex4.h"

namespace dl {

constexpr int kDesc4Words = 12;

template <class T>
struct alignas(16) Quad {
    T v[4];
};
template <class RowT>
struct alignas(sizeof(RowT) * 4) RowQuad {
    RowT v[4];
};

// The cold single-column path re-reads the kernel arguments from the kernarg segment (they sit at offset 0) instead of
// keeping a dozen pointers alive in SGPRs across the hot loop.
template <class T>
__device__ __forceinline__ const FusedArgs<T>& kernarg_args(const FusedArgs<T>& fallback) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const FusedArgs<T>*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    return fallback;
#endif
}

template <class T, class RowT, bool LAM_LDS, bool GRAD_LDS>
__global__ __launch_bounds__(kFusedThreads) void matching_fused_kernel4(FusedArgs<T> g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const WgCtx<T> w = fused_prologue<T, LAM_LDS, GRAD_LDS>(g, smem, tid, lane, wave, wg);
    const T s = w.s;
    const LaneConst lc = make_lane_const(lane);
    double obj = 0.0, ssq = 0.0;

    const uint32_t t_begin = g.wg_tile_begin[wg];
    const uint32_t t_end = g.wg_tile_begin[wg + 1];
    uint64_t k_base = 0;  // window start of the workgroup's first tile (multiple of 4)
    if (t_begin < t_end) {
        const uint32_t lo = g.tiles32[(size_t)t_begin * kDesc4Words], hi = g.tiles32[(size_t)t_begin * kDesc4Words + 1];
        k_base = (((uint64_t)hi << 32) | lo) & ((1ull << 40) - 1) & ~3ull;  // a long first tile starts anywhere
    }
    const uint64_t nnz_al4 = (uint64_t)g.nnz & ~3ull;                          // host guarantees nnz_al4 >= 4 for this layout
    if (k_base + 4 > nnz_al4) k_base = nnz_al4 - 4;
    const uint32_t kb_lo = __builtin_amdgcn_readfirstlane((uint32_t)k_base);
    k_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(k_base >> 32)) << 32) | kb_lo;
    const T* __restrict__ a_wg = g.a + k_base;
    const T* __restrict__ c_wg = g.c + k_base;
    const RowT* __restrict__ r_wg = reinterpret_cast<const RowT*>(g.rowidx) + k_base;
    T* __restrict__ x_wg = g.x_out ? g.x_out + k_base : nullptr;
    // last quad that may be read with a vector load (the array's final partial quad is never part of a window)
    const uint32_t max_quad = (uint32_t)((nnz_al4 - k_base) / 4) - 1u;

    const size_t last_word = (size_t)t_end * kDesc4Words - 1;
    auto load_desc = [&](uint32_t t) -> uint32_t {  // unconditional, clamped (see matching_kernels.hip)
        const uint32_t l = (uint32_t)lane < (uint32_t)kDesc4Words ? (uint32_t)lane : (uint32_t)kDesc4Words - 1u;
        size_t idx = (size_t)t * kDesc4Words + l;
        idx = idx < last_word ? idx : last_word;
        const uint32_t v = g.tiles32[idx];
        const bool ok = (uint32_t)lane < (uint32_t)kDesc4Words && t < t_end;
        return v & (0u - (uint32_t)ok);
    };
    struct Tile {
        uint32_t dv;  // descriptor words, one per lane; the head masks and the projection id are only unpacked when used
        uint32_t w0lo, w0hi;
        uint32_t q0;  // quad offset of lane 0 relative to k_base
        Quad<T> a, c;
        RowQuad<RowT> r;
    };
    auto rl = [&](uint32_t dv, int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane(dv, i); };
    auto unpack_and_issue = [&](uint32_t dv, Tile& t) {
        t.dv = dv;
        t.w0lo = rl(dv, 0);
        t.w0hi = rl(dv, 1);
        const uint32_t hi = (t.w0hi >> 8) & 0x1FF;
        const bool is_long = (t.w0hi & (1u << 19)) != 0;
        t.q0 = (hi == 0 || is_long) ? 0u : (t.w0lo - kb_lo) >> 2;  // exact: a workgroup spans < 2^32 non-zeros
        uint32_t q = t.q0 + (uint32_t)lane;
        q = q < max_quad ? q : max_quad;
        t.a = *byte_offset(reinterpret_cast<const Quad<T>*>(a_wg), q * (uint32_t)sizeof(Quad<T>));
        t.c = *byte_offset(reinterpret_cast<const Quad<T>*>(c_wg), q * (uint32_t)sizeof(Quad<T>));
        t.r = *byte_offset(reinterpret_cast<const RowQuad<RowT>*>(r_wg), q * (uint32_t)sizeof(RowQuad<RowT>));
    };

    uint32_t ti = t_begin + (uint32_t)wave;
    Tile cur;
    uint32_t dv_next = 0;
    if (ti < t_end) {
        const uint32_t dv0 = load_desc(ti);
        dv_next = load_desc(ti + kFusedWaves);
        unpack_and_issue(dv0, cur);
    }
    while (ti < t_end) {
        // the current tile's lambda gathers go out first: their LDS latency overlaps the descriptor unpack and the
        // issue of the next tile's loads
        uint32_t row[kSlots];
        T lam[kSlots];
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            row[j] = (uint32_t)cur.r.v[j];
            lam[j] = (T)1;
            if (!(g.ablate & 2)) lam[j] = LAM_LDS ? w.lam_s[row[j]] : (T)(s * g.lambda[row[j]]);
        }
        Tile nxt;
        const uint32_t ti_next = ti + kFusedWaves;
        unpack_and_issue(dv_next, nxt);
        dv_next = load_desc(ti_next + kFusedWaves);

        const uint32_t hi = (cur.w0hi >> 8) & 0x1FF, lo = (cur.w0hi >> 17) & 3;
        const bool is_long = (cur.w0hi & (1u << 19)) != 0;
        const uint32_t pid = rl(cur.dv, 10);
        if (!is_long) {
            const ProjT<T> pj = w.proj_s[pid < (uint32_t)(kProjLds - 1) ? pid : (uint32_t)(kProjLds - 1)];
            const int kind = __builtin_amdgcn_readfirstlane(pj.kind);
            T v[kSlots], x[kSlots];
            const uint32_t e0 = 4u * (uint32_t)lane - lo, span = hi - lo;  // element j of the lane is in the tile iff e0 + j < span
#pragma unroll
            for (int j = 0; j < kSlots; ++j) {
                const T t1 = (T)(cur.a.v[j] * lam[j]);     // sparse_utils.py:79
                const T vj = (T)(t1 + (T)(s * cur.c.v[j]));  // matching.py:66,142
                v[j] = (e0 + (uint32_t)j < span) ? vj : (T)0;
                x[j] = v[j];
            }
            const bool simplex_tile = is_simplex_kind(kind);
            if (!simplex_tile && !(g.ablate & 4)) {
#pragma unroll
                for (int j = 0; j < kSlots; ++j) x[j] = project_pointwise(v[j], pj);
            }
            if (simplex_tile && !(g.ablate & 4)) {
                uint64_t H[kSlots];
#pragma unroll
                for (int j = 0; j < kSlots; ++j) H[j] = ((uint64_t)rl(cur.dv, 3 + 2 * j) << 32) | rl(cur.dv, 2 + 2 * j);
                const Seg4 sg = make_seg4(H);
                simplex_tile4(v, sg, pj, lc, x, g.ablate);
            }
            T o32 = (T)0, q32 = (T)0;
#pragma unroll
            for (int j = 0; j < kSlots; ++j) {
                const T xq = (e0 + (uint32_t)j < span) ? x[j] : (T)0;  // (a clamp with lower > 0 moves the zero-filled slots)
                const T ax = (T)(cur.a.v[j] * xq);
                if (ax != (T)0 && !(g.ablate & 1)) scatter_fixed(w.gacc, row[j], ax, w.scale);
                o32 = (T)(o32 + (T)(cur.c.v[j] * xq));
                q32 = (T)(q32 + (T)(xq * xq));
                x[j] = xq;
            }
            obj += (double)o32;
            ssq += (double)q32;
            if (x_wg) {
#pragma unroll
                for (int j = 0; j < kSlots; ++j)
                    if (e0 + (uint32_t)j < span) x_wg[4 * (cur.q0 + (uint32_t)lane) + j] = x[j];  // neighbours own the rest of the quad
            }
        } else {
            const FusedArgs<T>& gk = kernarg_args(g);
            const ProjT<T> pl = lookup_proj(gk, w.proj_s, pid);
            const uint64_t k0 = (((uint64_t)cur.w0hi << 32) | cur.w0lo) & ((1ull << 40) - 1);
            const uint64_t len = ((uint64_t)rl(cur.dv, 3) << 32) | rl(cur.dv, 2);
            process_long_tile<T, RowT, LAM_LDS>(gk, pl, k0, len, w.lam_s, w.gacc, s, w.scale, lane, obj, ssq);
        }
        ti = ti_next;
        cur = nxt;
    }
    fused_epilogue<T, GRAD_LDS>(g, w, obj, ssq, tid, lane, wave, wg);
}

template <class T, class RowT, bool LAM, bool GRAD>
static int launch_fused4_inst(const dl_matching* h, const FusedArgs<T>& args, hipStream_t st) {
    auto kern = matching_fused_kernel4<T, RowT, LAM, GRAD>;
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
static int launch_fused4_rt(const dl_matching* h, const FusedArgs<T>& args, hipStream_t st) {
    if (h->lam_lds && h->grad_lds) return launch_fused4_inst<T, RowT, true, true>(h, args, st);
    if (h->grad_lds) return launch_fused4_inst<T, RowT, false, true>(h, args, st);
    return launch_fused4_inst<T, RowT, false, false>(h, args, st);
}

int launch_fused4_f32(const dl_matching* h, const FusedArgs<float>& args, hipStream_t st) {
    return h->row_bytes == 2 ? launch_fused4_rt<float, uint16_t>(h, args, st) : launch_fused4_rt<float, uint32_t>(h, args, st);
}
int launch_fused4_f64(const dl_matching* h, const FusedArgs<double>& args, hipStream_t st) {
    return h->row_bytes == 2 ? launch_fused4_rt<double, uint16_t>(h, args, st) : launch_fused4_rt<double, uint32_t>(h, args, st);
}

}  // namespace dl
