// developer micro-benchmark: throughput and latency of device-scope same-address atomics issued by every wavefront of a
// full launch (256 workgroups x 16 wavefronts), as a dynamic tile schedule would.   hipcc --offload-arch=gfx950 -O3 tools/atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(1024) void claim_kernel(unsigned int* counter, int claims, int all_lanes, unsigned long long* cyc, unsigned int* sink) {
    const int lane = threadIdx.x & 63;
    unsigned int acc = 0;
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < claims; ++k) {
        unsigned int v = 0;
        if (all_lanes) {
            v = atomicAdd(counter + lane, lane == 0 ? 1u : 0u);  // one instruction, 64 lanes, two cache lines
        } else if (lane == 0) {
            v = atomicAdd(counter, 1u);
        }
        acc += __builtin_amdgcn_readfirstlane(v);
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0) {
        cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
        sink[blockIdx.x * 16 + (threadIdx.x >> 6)] = acc;
    }
}

int main() {
    unsigned int* counter;
    unsigned long long* cyc;
    unsigned int* sink;
    hipMalloc(&counter, 4096);
    hipMalloc(&cyc, 4096 * 8);
    hipMalloc(&sink, 4096 * 4);
    for (int all = 0; all < 2; ++all)
        for (int claims : {1, 8, 64, 256}) {
            hipMemset(counter, 0, 4096);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(claim_kernel, dim3(256), dim3(1024), 0, 0, counter, claims, all, cyc, sink);  // warm
            hipDeviceSynchronize();
            hipMemset(counter, 0, 4096);
            hipEventRecord(e0);
            hipLaunchKernelGGL(claim_kernel, dim3(256), dim3(1024), 0, 0, counter, claims, all, cyc, sink);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(4096);
            hipMemcpy(h.data(), cyc, 4096 * 8, hipMemcpyDeviceToHost);
            unsigned int total;
            hipMemcpy(&total, counter, 4, hipMemcpyDeviceToHost);
            double mean = 0;
            for (auto v : h) mean += (double)v;
            mean /= 4096;
            printf("all_lanes=%d claims/wave=%4d  kernel %.1f us  -> %.1f ns per claim (device total %u)  per-wave latency %.0f ns/claim\n", all, claims, ms * 1e3,
                   ms * 1e6 / (4096.0 * claims), total, mean * 10.0 / claims);
        }
    return 0;
}
