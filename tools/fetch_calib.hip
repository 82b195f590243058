// developer tool: what rocprofv3's FETCH_SIZE reports for streaming reads of KNOWN size at the access widths the fused kernel
// uses (the guide calibrates the x2 correction for 16-byte-per-lane loads only).
//   k16: 16 B per lane, non-temporal (window tiles: value quads)      k8: 8 B per lane, nt (window tiles: uint16 row quads)
//   k4 : 4 B per lane, nt (slices: values)                            k2: 2 B per lane, nt (slices: uint16 rows)
//   k4c: 4 B per lane, cached (descriptor words)
// Every kernel streams the same 1 GiB buffer once (256 workgroups x 1024 threads, grid-stride), so FETCH_SIZE x 1 KiB / 2^30 is
// the factor to divide by.   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <class V, bool NT>
__global__ __launch_bounds__(1024) void stream(const V* __restrict__ p, size_t n, unsigned long long* out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        V v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
        acc += b[0];
    }
    if (acc == 0x123456789ull) *out = acc;
}
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned short u4 __attribute__((ext_vector_type(4)));
int main() {
    const size_t bytes = 1ull << 30;
    void* buf;
    unsigned long long* out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 8);
    hipMemset(buf, 1, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((stream<f4, true>), dim3(256), dim3(1024), 0, 0, (const f4*)buf, bytes / 16, out);
        hipLaunchKernelGGL((stream<u4, true>), dim3(256), dim3(1024), 0, 0, (const u4*)buf, bytes / 8, out);
        hipLaunchKernelGGL((stream<float, true>), dim3(256), dim3(1024), 0, 0, (const float*)buf, bytes / 4, out);
        hipLaunchKernelGGL((stream<unsigned short, true>), dim3(256), dim3(1024), 0, 0, (const unsigned short*)buf, bytes / 2, out);
        hipLaunchKernelGGL((stream<float, false>), dim3(256), dim3(1024), 0, 0, (const float*)buf, bytes / 4, out);
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
