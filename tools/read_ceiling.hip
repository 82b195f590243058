// developer tool: what a pure streaming READ reaches on this box (the fused kernel is read dominated; the guide's 6.3 TB/s is a
// copy).  One pass over a 12 GiB buffer with 16-byte non-temporal loads, 256 workgroups x 1024 threads, 4 / 8 loads in flight per
// lane; also 4-byte loads (what the column-per-lane slices issue).   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/rc tools/read_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
template <class V, int U>
__global__ __launch_bounds__(1024) void rd(const V* __restrict__ p, size_t n, float* out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += ((const float*)&v[u])[0];
    }
    for (; i < n; i += stride) acc += ((const float*)&p[i])[0];
    if (acc == 12345.678f) *out = acc;
}
template <class V, int U>
static void run(const char* name, const void* buf, size_t bytes, float* out, int wg) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((rd<V, U>), dim3(wg), dim3(1024), 0, 0, (const V*)buf, bytes / sizeof(V), out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("%-28s wg %4d  %.3f ms  %.0f GB/s\n", name, wg, best, bytes / (best * 1e-3) / 1e9);
}
int main() {
    const size_t bytes = 12ull << 30;
    void* buf;
    float* out;
    if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
    hipMalloc(&out, 4);
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    for (int wg : {256, 512, 1024}) {
        run<f4, 4>("16 B/lane, 4 in flight", buf, bytes, out, wg);
        run<f4, 8>("16 B/lane, 8 in flight", buf, bytes, out, wg);
        run<float, 8>("4 B/lane, 8 in flight", buf, bytes, out, wg);
        run<float, 16>("4 B/lane, 16 in flight", buf, bytes, out, wg);
    }
    return 0;
}
