// tools/lds_bench.hip -- developer micro-benchmark (not part of the product): cost of the LDS operations the fused
// kernel relies on, in CU cycles per wavefront-instruction with 16 wavefronts resident per CU.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_bench.hip -o /tmp/lds_bench && /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int M = 10000;
constexpr int THREADS = 1024;
constexpr int ITERS = 2000;

template <int MODE>
__global__ __launch_bounds__(THREADS) void k(const unsigned* __restrict__ idx, float* __restrict__ out, long long* __restrict__ cyc) {
    __shared__ float lds[2 * M];
    for (int i = threadIdx.x; i < 2 * M; i += THREADS) lds[i] = 0.f;
    __syncthreads();
    const unsigned base = idx[blockIdx.x * THREADS + threadIdx.x];
    unsigned r = base;
    float acc = 0.f;
    float v = 1.0f + threadIdx.x * 1e-3f;
    long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        r = (r * 1664525u + 1013904223u);
        unsigned a = (r >> 8) % M;
        if (MODE == 0) acc += lds[a];                                   // random gather
        if (MODE == 1) atomicAdd(&lds[M + a], v);                       // random scatter-add (no return)
        if (MODE == 2) acc += __int_as_float(__builtin_amdgcn_ds_bpermute(((threadIdx.x + it) & 63) << 2, __float_as_int(v + acc)));
        if (MODE == 3) atomicAdd(&lds[M + ((a & ~63u) % M)], v);        // heavy same-address (64-aligned buckets)
        if (MODE == 4) atomicAdd(&lds[M + (threadIdx.x >> 2) + (it & 7) * 256], v);  // 4 consecutive lanes share an address
        if (MODE == 5) { atomicAdd(&lds[M + a], v); acc += lds[a]; }   // gather + scatter
        if (MODE == 6) acc += v * 1.0001f + acc * 0.5f;                 // VALU only reference
        if (MODE == 7) atomicMax((int*)&lds[M + a], __float_as_int(v));  // ds_max_i32 as stand-in for max
        if (MODE == 8) atomicAdd((unsigned*)&lds[M + a], (unsigned)it);   // ds_add_u32
        if (MODE == 9) atomicAdd((unsigned long long*)&lds[(a & ~1u)], (unsigned long long)it + ((unsigned long long)r << 20));  // ds_add_u64
        if (MODE == 10) {  // fixed-point scatter: cvt + fma magic + bfe + ds_add_u64
            double d = fma((double)(v + acc), 1048576.0, 6755399441055744.0);
            int lo = __double2loint(d), hi = __double2hiint(d);
            hi = (hi << 12) >> 12;
            unsigned long long q = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
            atomicAdd((unsigned long long*)&lds[(a & ~1u)], q);
            acc += 1e-9f;
        }
        if (MODE == 11) atomicAdd((unsigned long long*)&lds[2 * ((threadIdx.x >> 2) + (it & 7) * 256)], (unsigned long long)it);  // u64, 4-lane same address
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * THREADS + threadIdx.x] = acc + lds[threadIdx.x];
}

template <int MODE>
void run(const char* name, const unsigned* d_idx, float* d_out, long long* d_cyc, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, THREADS>>>(d_idx, d_out, d_cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, THREADS>>>(d_idx, d_out, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(blocks);
    hipMemcpy(c.data(), d_cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto x : c) avg += x;
    avg /= blocks;
    // clock64 = s_memtime ticks at 100 MHz on gfx9; use wall time instead: 16 waves per CU, ITERS wave-instr each
    double ns_per_winstr = (double)ms * 1e6 / ((double)ITERS * 16);
    printf("%-34s %8.3f ms   %7.2f ns per wave-instr per CU  (~%6.1f cycles @2.3GHz)   [memtime ticks/iter %.2f]\n", name, ms, ns_per_winstr,
           ns_per_winstr * 2.3, avg / ITERS);
}

int main() {
    int blocks = 256;
    std::vector<unsigned> h(blocks * THREADS);
    for (auto& x : h) x = (unsigned)rand();
    unsigned* d_idx;
    float* d_out;
    long long* d_cyc;
    hipMalloc(&d_idx, h.size() * 4);
    hipMalloc(&d_out, h.size() * 4);
    hipMalloc(&d_cyc, blocks * 8);
    hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<6>("valu only", d_idx, d_out, d_cyc, blocks);
    run<0>("ds_read_b32 random", d_idx, d_out, d_cyc, blocks);
    run<1>("ds_add_f32 random", d_idx, d_out, d_cyc, blocks);
    run<2>("ds_bpermute", d_idx, d_out, d_cyc, blocks);
    run<3>("ds_add_f32 64-way same address", d_idx, d_out, d_cyc, blocks);
    run<4>("ds_add_f32 4-lane groups same addr", d_idx, d_out, d_cyc, blocks);
    run<5>("ds_add_f32 + ds_read_b32 random", d_idx, d_out, d_cyc, blocks);
    run<7>("ds_max_i32 random", d_idx, d_out, d_cyc, blocks);
    run<8>("ds_add_u32 random", d_idx, d_out, d_cyc, blocks);
    run<9>("ds_add_u64 random", d_idx, d_out, d_cyc, blocks);
    run<10>("fixed-point cvt + ds_add_u64", d_idx, d_out, d_cyc, blocks);
    run<11>("ds_add_u64 4-lane same addr", d_idx, d_out, d_cyc, blocks);
    return 0;
}
