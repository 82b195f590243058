// developer micro-benchmark: three ways for the 256 workgroups of a fused launch to hand their private int64[m] gradient
// accumulators (LDS) to the m-sized side of the iteration:
//   slab    every workgroup stores its m accumulators to its own slab (256 x m x 8 B), a second kernel sums the 256 slabs
//   device  every workgroup adds its accumulators into ONE int64[m] with device-scope atomics (no second kernel)
//   xcd     every workgroup adds into the int64[m] of ITS XCD with workgroup-scope atomics (executed in that XCD's L2), a second
//           kernel sums 8 slabs
// hipcc --offload-arch=gfx950 -O3 tools/flush_bench.hip -o /tmp/flush_bench && /tmp/flush_bench [m]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kT = 1024;

template <int MODE>
__global__ __launch_bounds__(kT) void flush_kernel(long long* out, int m, int mpad) {
    extern __shared__ long long acc[];
    for (int i = threadIdx.x; i < m; i += kT) acc[i] = (long long)(i + 1) * (blockIdx.x + 1);
    __syncthreads();
    if (MODE == 0) {
        long long* slab = out + (size_t)blockIdx.x * mpad;
        for (int i = threadIdx.x; i < m; i += kT) slab[i] = acc[i];
    } else if (MODE == 1) {
        for (int i = threadIdx.x; i < m; i += kT)
            __hip_atomic_fetch_add((unsigned long long*)out + i, (unsigned long long)acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        const unsigned int xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11)) & 7u;
        unsigned long long* o = (unsigned long long*)out + (size_t)xcc * mpad;
        for (int i = threadIdx.x; i < m; i += kT) __hip_atomic_fetch_add(o + i, (unsigned long long)acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__global__ __launch_bounds__(1024) void sum_kernel(const long long* in, int n_slabs, int m, int mpad, long long* out) {
    __shared__ long long sh[1024];
    const int rl = threadIdx.x & 63, ws = threadIdx.x >> 6;
    const int row = blockIdx.x * 64 + rl;
    long long a = 0;
    if (row < m)
        for (int w = ws; w < n_slabs; w += 16) a += in[(size_t)w * mpad + row];
    sh[threadIdx.x] = a;
    __syncthreads();
    if (ws == 0 && row < m) {
        long long t = 0;
        for (int q = 0; q < 16; ++q) t += sh[q * 64 + rl];
        out[row] = t;
    }
}

int main(int argc, char** argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 10000;
    const int mpad = (m + 15) / 16 * 16, G = 256;
    long long *slabs, *res;
    hipMalloc(&slabs, sizeof(long long) * (size_t)G * mpad);
    hipMalloc(&res, sizeof(long long) * mpad);
    hipFuncSetAttribute((const void*)flush_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)flush_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)flush_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    const size_t lds = sizeof(long long) * m;
    const char* names[3] = {"slab  ", "device", "xcd   "};
    for (int mode = 0; mode < 3; ++mode) {
        double t_flush = 0, t_sum = 0;
        long long check = 0;
        const int reps = 20;
        for (int r = 0; r < reps + 2; ++r) {
            const int n_in = mode == 0 ? G : (mode == 1 ? 1 : 8);
            hipMemsetAsync(slabs, 0, sizeof(long long) * (size_t)n_in * mpad, 0);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(flush_kernel<0>, dim3(G), dim3(kT), lds, 0, slabs, m, mpad);
            if (mode == 1) hipLaunchKernelGGL(flush_kernel<1>, dim3(G), dim3(kT), lds, 0, slabs, m, mpad);
            if (mode == 2) hipLaunchKernelGGL(flush_kernel<2>, dim3(G), dim3(kT), lds, 0, slabs, m, mpad);
            hipEventRecord(e1);
            hipLaunchKernelGGL(sum_kernel, dim3((m + 63) / 64), dim3(1024), 0, 0, slabs, n_in, m, mpad, res);
            hipEventRecord(e2);
            hipDeviceSynchronize();
            float a, b;
            hipEventElapsedTime(&a, e0, e1);
            hipEventElapsedTime(&b, e1, e2);
            if (r >= 2) { t_flush += a; t_sum += b; }
            std::vector<long long> h(m);
            hipMemcpy(h.data(), res, sizeof(long long) * m, hipMemcpyDeviceToHost);
            check = h[m - 1];
        }
        // expected: sum_w (m) * (w + 1) = m * G (G + 1) / 2
        printf("%s m=%d: fill+flush kernel %.1f us, sum kernel %.1f us (%s)  last element %lld (want %lld)\n", names[mode], m, t_flush / 20 * 1e3, t_sum / 20 * 1e3,
               mode == 1 ? "not needed: the accumulators ARE the sums" : "needed", check, (long long)m * G * (G + 1) / 2);
    }
    return 0;
}
