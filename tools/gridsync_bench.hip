// developer micro-benchmark: the m-sized side of one dual-ascent iteration (everything but the streaming of the non-zeros) as
//   launches     three stream-ordered launches, the shape of dl_agd_run_matching today:
//                flush (256 workgroups store their int64[m] accumulators as slabs; its head re-stages the m duals, as the
//                fused pass's prologue does) -> stats (m / 64 workgroups sum the slabs, write g and six partial statistics each)
//                -> apply (m / 1024 workgroups sum the partials and update their rows)
//   persistent   ONE launch for all iterations: 256 resident workgroups, two grid barriers per iteration (device-scope arrival
//                counter, bounded spin); slabs, g and the partials are written through (agent-scope stores) and read past the L2
//                (agent-scope loads) because the eight XCDs' L2s are not coherent with each other inside a launch; every
//                workgroup sums its own m / 256 rows of the slabs, and after the second barrier derives the step and re-stages
//                ALL m duals itself (the redundancy the fused pass's prologue has anyway)
//   barrier      the persistent kernel with the work removed: the cost of the two grid barriers alone
//   empty        back-to-back launches of an empty kernel (256 x 1024 threads, 128 KB of LDS): the launch boundary
// hipcc --offload-arch=gfx950 -O3 tools/gridsync_bench.hip -o /tmp/gridsync_bench && /tmp/gridsync_bench [m] [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kT = 1024;
constexpr int kG = 256;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long wall() { return __builtin_readcyclecounter(); }

struct P {
    long long* slabs;   // [kG][mpad]
    float* g;           // [m]
    float* x;           // [m]
    float* y;           // [m]
    double* partials;   // [kG][8]
    unsigned int* counter;
    int* dead;
    int m, mpad, iters;
    int work;           // 0: barriers only
};

__device__ __forceinline__ void grid_barrier(const P& p, unsigned int target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(p.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long spins = 0;
        while (__hip_atomic_load(p.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 20000000ull) {  // bounded: a grid that is not fully resident must not hang the device
                __hip_atomic_store(p.dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(kT) void persistent_kernel(P p) {
    extern __shared__ long long acc[];            // [m] accumulators, then [m] floats of staged duals
    float* lam = reinterpret_cast<float*>(acc + p.mpad);
    const int tid = threadIdx.x, wg = blockIdx.x;
    const int rpw = (p.m + kG - 1) / kG;          // rows this workgroup reduces
    const int r0 = wg * rpw;
    __shared__ long long red[kT];
    unsigned int epoch = 0;
    for (int i = tid; i < p.m; i += kT) lam[i] = 0.f;
    for (int it = 0; it < p.iters; ++it) {
        if (p.work) {
            for (int i = tid; i < p.m; i += kT) acc[i] = (long long)(i + 1) * (wg + 1) + it + (long long)lam[i];
            __syncthreads();
            long long* slab = p.slabs + (size_t)wg * p.mpad;
            for (int i = tid; i < p.m; i += kT) __hip_atomic_store(slab + i, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        grid_barrier(p, ++epoch * kG);
        if (p.work) {
            // rows r0 .. r0 + rpw - 1: thread (row rl, slab group sg) sums slabs sg, sg + ng, ...
            const int ng = kT / 64;               // 16 slab groups of 64 lanes; lanes >= rpw idle (rpw <= 64)
            const int rl = tid & 63, sg = tid >> 6;
            long long a = 0;
            if (rl < rpw && r0 + rl < p.m) {
                long long v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = __hip_atomic_load(p.slabs + (size_t)(sg + ng * u) * p.mpad + r0 + rl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int u = 0; u < 16; ++u) a += v[u];
            }
            red[tid] = a;
            __syncthreads();
            if (tid < 64) {
                double st = 0.0;
                if (rl < rpw && r0 + rl < p.m) {
                    long long t = 0;
                    for (int q = 0; q < ng; ++q) t += red[q * 64 + rl];
                    const float gj = (float)t * 1e-9f;
                    __hip_atomic_store(p.g + r0 + rl, gj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    st = (double)gj * gj;
                }
                for (int o = 32; o > 0; o >>= 1) st += __shfl_xor(st, o);
                if (tid < 6) __hip_atomic_store(p.partials + wg * 8 + tid, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        grid_barrier(p, ++epoch * kG);
        if (p.work) {
            // every workgroup: the partials (same order everywhere), then all m rows
            double s = 0.0;
            if (tid < kG) s = __hip_atomic_load(p.partials + tid * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            __shared__ double sh[4];
            if (tid < kG && (tid & 63) == 0) sh[tid >> 6] = s;
            __syncthreads();
            const float step = (float)(1e-3 / (1.0 + sh[0] + sh[1] + sh[2] + sh[3]));
            for (int i = tid; i < p.m; i += kT) {
                const float gj = __hip_atomic_load(p.g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float xn = p.x[i] + step * gj + 0.5f * p.y[i];
                lam[i] = xn;
                if (wg == 0) p.y[i] = xn * 0.5f;   // (stand-in for the state the one writing workgroup stores)
            }
            __syncthreads();
        }
    }
    if (tid == 0 && wg == 0) p.g[0] += lam[0];
}

__global__ __launch_bounds__(kT) void flush_kernel(P p, int it) {
    extern __shared__ long long acc[];
    float* lam = reinterpret_cast<float*>(acc + p.mpad);
    const int tid = threadIdx.x, wg = blockIdx.x;
    for (int i = tid; i < p.m; i += kT) lam[i] = p.x[i];   // prologue: the duals
    __syncthreads();
    for (int i = tid; i < p.m; i += kT) acc[i] = (long long)(i + 1) * (wg + 1) + it + (long long)lam[i];
    __syncthreads();
    long long* slab = p.slabs + (size_t)wg * p.mpad;
    for (int i = tid; i < p.m; i += kT) slab[i] = acc[i];
}
__global__ __launch_bounds__(kT) void stats_kernel(P p) {
    __shared__ long long sh[kT];
    const int rl = threadIdx.x & 63, ws = threadIdx.x >> 6;
    const int row = blockIdx.x * 64 + rl;
    long long a = 0;
    if (row < p.m) {
        long long v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p.slabs[(size_t)(ws + 16 * u) * p.mpad + row];
#pragma unroll
        for (int u = 0; u < 16; ++u) a += v[u];
    }
    sh[threadIdx.x] = a;
    __syncthreads();
    if (ws == 0) {
        double st = 0.0;
        if (row < p.m) {
            long long t = 0;
            for (int q = 0; q < 16; ++q) t += sh[q * 64 + rl];
            const float gj = (float)t * 1e-9f;
            p.g[row] = gj;
            st = (double)gj * gj;
        }
        for (int o = 32; o > 0; o >>= 1) st += __shfl_xor(st, o);
        if (threadIdx.x < 6) p.partials[blockIdx.x * 8 + threadIdx.x] = st;
    }
}
__global__ __launch_bounds__(kT) void apply_kernel(P p, int n_stat) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n_stat; i += kT) s += p.partials[i * 8];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ double sh[kT / 64];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < kT / 64; ++w) t += sh[w];
    const float step = (float)(1e-3 / (1.0 + t));
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i < p.m) {
        const float xn = p.x[i] + step * p.g[i] + 0.5f * p.y[i];
        p.y[i] = xn * 0.5f;
        p.x[i] = xn;
    }
}
__global__ __launch_bounds__(kT) void empty_kernel(int* p) {
    extern __shared__ long long acc[];
    if (p && threadIdx.x == 5000) acc[0] = 1;
}

int main(int argc, char** argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 10000;
    const int iters = argc > 2 ? atoi(argv[2]) : 500;
    P p;
    p.m = m;
    p.mpad = (m + 15) / 16 * 16;
    p.iters = iters;
    CK(hipMalloc(&p.slabs, sizeof(long long) * (size_t)kG * p.mpad));
    CK(hipMalloc(&p.g, sizeof(float) * p.mpad));
    CK(hipMalloc(&p.x, sizeof(float) * p.mpad));
    CK(hipMalloc(&p.y, sizeof(float) * p.mpad));
    CK(hipMalloc(&p.partials, sizeof(double) * 8 * 4096));
    CK(hipMalloc(&p.counter, 64));
    CK(hipMalloc(&p.dead, 64));
    CK(hipMemset(p.x, 0, sizeof(float) * p.mpad));
    CK(hipMemset(p.y, 0, sizeof(float) * p.mpad));
    CK(hipMemset(p.partials, 0, sizeof(double) * 8 * 4096));
    const size_t lds = sizeof(long long) * p.mpad + sizeof(float) * p.mpad;
    CK(hipFuncSetAttribute((const void*)persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CK(hipFuncSetAttribute((const void*)flush_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CK(hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persistent_kernel, kT, lds));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("m=%d iterations=%d  LDS %zu B  occupancy %d workgroup(s)/CU x %d CUs\n", m, iters, lds, occ, prop.multiProcessorCount);
    if (occ * prop.multiProcessorCount < kG) { printf("grid of %d would not be resident\n", kG); return 1; }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char* what) {
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        int dead = 0;
        CK(hipMemcpy(&dead, p.dead, sizeof(int), hipMemcpyDeviceToHost));
        printf("%-12s %8.2f us per iteration%s\n", what, 1e3 * ms / iters, dead ? "   (BARRIER TIMED OUT)" : "");
    };
    const int n_stat = (m + 63) / 64, n_apply = (m + kT - 1) / kT;
    for (int rep = 0; rep < 3; ++rep) {
        // launches
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < iters; ++it) {
            hipLaunchKernelGGL(flush_kernel, dim3(kG), dim3(kT), lds, 0, p, it);
            hipLaunchKernelGGL(stats_kernel, dim3(n_stat), dim3(kT), 0, 0, p);
            hipLaunchKernelGGL(apply_kernel, dim3(n_apply), dim3(kT), 0, 0, p, n_stat);
        }
        CK(hipEventRecord(e1, 0));
        report("launches");
        // persistent, with and without the work
        for (int work = 1; work >= 0; --work) {
            CK(hipMemset(p.counter, 0, 64));
            CK(hipMemset(p.dead, 0, 64));
            p.work = work;
            void* args[] = {&p};
            CK(hipEventRecord(e0, 0));
            CK(hipLaunchCooperativeKernel((const void*)persistent_kernel, dim3(kG), dim3(kT), args, (unsigned int)lds, 0));
            CK(hipEventRecord(e1, 0));
            report(work ? "persistent" : "barrier x2");
        }
        // empty launches
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(empty_kernel, dim3(kG), dim3(kT), lds, 0, (int*)nullptr);
        CK(hipEventRecord(e1, 0));
        report("empty 256wg");
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(empty_kernel, dim3(n_apply), dim3(kT), 0, 0, (int*)nullptr);
        CK(hipEventRecord(e1, 0));
        report("empty small");
    }
    return 0;
}
