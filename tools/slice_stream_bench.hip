// developer tool: what the ACCESS PATTERN of the column-per-lane slices reaches, compute taken out.  A wavefront walks slices of H steps
// (per step 256 B of a, 256 B of c, 128 B of uint16 rows, three arrays) q = W, W + S, ...; it requests a slice's 3 H loads, waits for all
// of them, adds them up, and goes on -- DEPTH = 1: nothing of its own in flight meanwhile (the fused kernel's slice loop); DEPTH = 2: the
// next slice is requested before the current one is consumed.  WAVES wavefronts per workgroup, one workgroup per CU (LDS pinned to 120 KB
// like the fused kernel).  WIN: the point-wise windows' pattern for comparison (16 + 16 + 8 bytes per lane, two tiles in flight).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ssb tools/slice_stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
extern __shared__ char smem[];
template <int H>
struct Regs {
    float a[H], c[H];
    unsigned r[H];
};
template <int H>
__device__ __forceinline__ void issue(Regs<H>& x, const float* a, const float* c, const unsigned short* r, size_t q, int lane) {
    const size_t base = q * 64 * H + lane;
#pragma unroll
    for (int t = 0; t < H; ++t) {
        x.a[t] = __builtin_nontemporal_load(a + base + 64 * t);
        x.c[t] = __builtin_nontemporal_load(c + base + 64 * t);
        x.r[t] = __builtin_nontemporal_load(r + base + 64 * t);
    }
}
// WORK: a dependent chain of that many fused multiply-adds after the data have arrived (4 cycles each when the wavefront is alone on its
// SIMD): the slice's compute phase, during which a DEPTH = 1 wavefront has nothing in flight
template <int H, int WORK>
__device__ __forceinline__ float consume(const Regs<H>& x) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < H; ++t) s += x.a[t] * x.c[t] + (float)x.r[t];
#pragma unroll 8
    for (int k = 0; k < WORK; ++k) s = __builtin_fmaf(s, 1.0000001f, 0.5f);
    return s;
}
template <int H, int DEPTH, int WAVES, int WORK>
__global__ __launch_bounds__(WAVES * 64) void slices(const float* __restrict__ a, const float* __restrict__ c, const unsigned short* __restrict__ r, size_t n_slices, float* out) {
    const int lane = threadIdx.x & 63;
    const size_t S = (size_t)gridDim.x * WAVES;
    size_t q = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    float acc = 0.f;
    if (threadIdx.x == 0) smem[0] = 0;
    if constexpr (DEPTH == 1) {
        for (; q < n_slices; q += S) {
            Regs<H> x;
            issue<H>(x, a, c, r, q, lane);
            acc += consume<H, WORK>(x);
        }
    } else {
        Regs<H> x0, x1;
        if (q < n_slices) issue<H>(x0, a, c, r, q, lane);
        for (; q < n_slices; q += 2 * S) {
            const size_t q1 = q + S < n_slices ? q + S : q, q2 = q + 2 * S < n_slices ? q + 2 * S : q;
            issue<H>(x1, a, c, r, q1, lane);
            acc += consume<H, WORK>(x0);
            issue<H>(x0, a, c, r, q2, lane);
            if (q + S < n_slices) acc += consume<H, WORK>(x1);
        }
    }
    if (acc == 12345.678f) *out = acc;
}
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void windows(const f4* __restrict__ a, const f4* __restrict__ c, const us4* __restrict__ r, size_t n_tiles, float* out) {
    const int lane = threadIdx.x & 63;
    const size_t S = (size_t)gridDim.x * WAVES;
    size_t q = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    float acc = 0.f;
    if (threadIdx.x == 0) smem[0] = 0;
    f4 a0, c0, a1, c1;
    us4 r0, r1;
    auto ld = [&](size_t t, f4& av, f4& cv, us4& rv) {
        const size_t i = (t < n_tiles ? t : n_tiles - 1) * 64 + lane;
        av = __builtin_nontemporal_load(a + i);
        cv = __builtin_nontemporal_load(c + i);
        rv = __builtin_nontemporal_load(r + i);
    };
    ld(q, a0, c0, r0);
    for (; q < n_tiles; q += 2 * S) {
        ld(q + S, a1, c1, r1);
        acc += a0[0] * c0[1] + a0[2] * c0[3] + (float)r0[0] + (float)r0[3];
        ld(q + 2 * S, a0, c0, r0);
        if (q + S < n_tiles) acc += a1[0] * c1[1] + a1[2] * c1[3] + (float)r1[0] + (float)r1[3];
    }
    if (acc == 12345.678f) *out = acc;
}
template <class K>
static void timeit(const char* name, size_t bytes, K launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-52s %.3f ms  %.0f GB/s\n", name, best, bytes / (best * 1e-3) / 1e9);
}
int main() {
    const size_t nnz = 500ull * 1000 * 1000;  // the simplex half of the benchmark's 100M-entity map
    float *a, *c, *out;
    unsigned short* r;
    if (hipMalloc(&a, nnz * 4) != hipSuccess || hipMalloc(&c, nnz * 4) != hipSuccess || hipMalloc(&r, nnz * 2) != hipSuccess) return 1;
    hipMalloc(&out, 4);
    hipMemset(a, 0, nnz * 4);
    hipMemset(c, 0, nnz * 4);
    hipMemset(r, 0, nnz * 2);
    hipDeviceSynchronize();
    const size_t bytes = nnz * 10;
    const int lds = 120 * 1024;
#define SL(H_, D_, W_, K_) \
    hipFuncSetAttribute((const void*)slices<H_, D_, W_, K_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    timeit("slices H=" #H_ " depth " #D_ " waves/CU " #W_ " work " #K_, bytes, [&] { hipLaunchKernelGGL((slices<H_, D_, W_, K_>), dim3(256), dim3(W_ * 64), lds, 0, a, c, r, nnz / (64 * H_), out); })
    SL(10, 1, 16, 0);
    SL(10, 2, 16, 0);
    SL(10, 1, 16, 100);
    SL(10, 2, 16, 100);
    SL(10, 1, 16, 200);
    SL(10, 2, 16, 200);
    SL(10, 2, 12, 200);
    SL(10, 1, 16, 400);
    SL(10, 2, 16, 400);
    SL(10, 2, 12, 400);
    SL(10, 2, 8, 400);
    SL(5, 1, 16, 100);
    SL(5, 2, 16, 100);
    SL(16, 1, 16, 300);
    SL(16, 2, 12, 300);
    hipFuncSetAttribute((const void*)windows<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    timeit("windows (16+16+8 B per lane, 2 tiles) waves/CU 16", bytes, [&] { hipLaunchKernelGGL((windows<16>), dim3(256), dim3(1024), lds, 0, (const f4*)a, (const f4*)c, (const us4*)r, nnz / 256, out); });
    hipFuncSetAttribute((const void*)windows<12>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    timeit("windows (16+16+8 B per lane, 2 tiles) waves/CU 12", bytes, [&] { hipLaunchKernelGGL((windows<12>), dim3(256), dim3(768), lds, 0, (const f4*)a, (const f4*)c, (const us4*)r, nnz / 256, out); });
    return 0;
}
