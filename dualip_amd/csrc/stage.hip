// stage.hip -- PAGEABLE host memory -> device memory at the link's speed (dl_stage_to_device).
//
// The reference's drivers keep the problem on the CPU and hand every rank CPU shards (benchmark/run_matching_benchmark_dist.py:95-110,
// run_solver.py:17-32 `transfer_tensors_to_device`: one `tensor.to(device)` per field).  From pageable memory that copy is a single host
// thread feeding one bounce buffer: 21 GB/s measured at the headline (16.8 GB in 0.79 s, profiles/r05ak_host_buffers_100m.json), and the
// int64 CSC indices -- 8 of the 16.8 GB -- cross the link only to be re-encoded to 16 bits on the other side.
// Here: worker threads claim 16 MB chunks, copy them (or NARROW them: int64 -> int32 / uint16, validated) into their own pinned buffers --
// two per thread, so a thread fills one while the other's DMA runs -- and queue each chunk's DMA on the thread's own stream as soon as it is
// filled.  The link sees pinned, back-to-back transfers from several queues; the narrowed indices cross it at a quarter of their size and
// never exist in HBM in 64-bit form.
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace dl {
namespace {

constexpr size_t kStageChunk = 16u << 20;  // bytes of DESTINATION data per chunk
constexpr int kStageMaxThreads = 16;
constexpr int kStageSlots = 2;  // pinned buffers per thread

struct StagePool {
    int device = -1;
    int threads = 0;
    void* buf[kStageMaxThreads][kStageSlots] = {};
    hipEvent_t ev[kStageMaxThreads][kStageSlots] = {};
    hipStream_t st[kStageMaxThreads] = {};
};
std::mutex g_stage_mu;           // one staging call at a time per process (the pool is shared)
std::vector<StagePool*> g_pools;  // one per device, kept for the life of the process (pinning memory is the slow part: ~0.1 s per pool)

StagePool* stage_pool(int device, int threads, hipError_t* err) {
    for (StagePool* p : g_pools)
        if (p->device == device && p->threads >= threads) return p;
    StagePool* p = new (std::nothrow) StagePool();
    if (!p) {
        *err = hipErrorOutOfMemory;
        return nullptr;
    }
    p->device = device;
    p->threads = threads;
    hipError_t e = hipSuccess;
    for (int t = 0; t < threads && e == hipSuccess; ++t) {
        e = hipStreamCreateWithFlags(&p->st[t], hipStreamNonBlocking);
        for (int s = 0; s < kStageSlots && e == hipSuccess; ++s) {
            e = hipHostMalloc(&p->buf[t][s], kStageChunk, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev[t][s], hipEventDisableTiming);
        }
    }
    if (e != hipSuccess) {  // (partial pools are dropped: a later call starts again)
        for (int t = 0; t < threads; ++t) {
            for (int s = 0; s < kStageSlots; ++s) {
                if (p->buf[t][s]) (void)hipHostFree(p->buf[t][s]);
                if (p->ev[t][s]) (void)hipEventDestroy(p->ev[t][s]);
            }
            if (p->st[t]) (void)hipStreamDestroy(p->st[t]);
        }
        delete p;
        *err = e;
        return nullptr;
    }
    g_pools.push_back(p);
    return p;
}

template <class S, class D>
int64_t narrow_chunk(const S* src, D* dst, int64_t n, bool dst_unsigned) {
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        const S v = src[i];
        const D d = (D)v;
        bad += ((S)d != v) || (dst_unsigned && v < 0) ? 1 : 0;
        dst[i] = d;
    }
    return bad;
}

}  // namespace
}  // namespace dl

using namespace dl;

extern "C" int dl_stage_to_device(void* dst_dev, const void* src_host, int64_t count, int src_bytes, int dst_bytes, int dst_unsigned, int threads,
                                  int64_t* bad_out_host, double* seconds_out_host) {
    if (bad_out_host) *bad_out_host = 0;
    if (seconds_out_host) *seconds_out_host = 0.0;
    if (count < 0 || (count > 0 && (!dst_dev || !src_host))) return fail(DL_E_ARG, "dl_stage_to_device: null array or negative count");
    const bool raw = src_bytes == dst_bytes;
    const bool narrow = (src_bytes == 8 && (dst_bytes == 4 || dst_bytes == 2)) || (src_bytes == 4 && dst_bytes == 2);
    if (!(raw && src_bytes >= 1 && src_bytes <= 16) && !narrow) return fail(DL_E_ARG, "dl_stage_to_device: %d -> %d bytes per element is not offered", src_bytes, dst_bytes);
    if (count == 0) return 0;
    int device = 0;
    DL_HIP(hipGetDevice(&device));
    int T = threads > 0 ? threads : 8;
    const int64_t per_chunk = (int64_t)(kStageChunk / (size_t)dst_bytes);
    const int64_t n_chunks = (count + per_chunk - 1) / per_chunk;
    if (T > kStageMaxThreads) T = kStageMaxThreads;
    if ((int64_t)T > n_chunks) T = (int)n_chunks;
    std::lock_guard<std::mutex> lock(g_stage_mu);
    hipError_t perr = hipSuccess;
    StagePool* pool = stage_pool(device, T, &perr);
    if (!pool) return hip_fail(perr, "dl_stage_to_device: pinned staging buffers");
    const auto t0 = std::chrono::steady_clock::now();
    std::atomic<int64_t> next{0}, bad_total{0};
    std::atomic<int> first_err{(int)hipSuccess};
    auto work = [&](int t) {
        hipError_t e = hipSetDevice(device);
        int uses = 0;
        while (e == hipSuccess) {
            const int64_t ci = next.fetch_add(1, std::memory_order_relaxed);
            if (ci >= n_chunks) break;
            const int s = uses % kStageSlots;
            if (uses >= kStageSlots) e = hipEventSynchronize(pool->ev[t][s]);  // this buffer's previous DMA has left it
            if (e != hipSuccess) break;
            const int64_t i0 = ci * per_chunk, n = (count - i0 < per_chunk) ? count - i0 : per_chunk;
            const char* src = static_cast<const char*>(src_host) + (size_t)i0 * (size_t)src_bytes;
            void* pin = pool->buf[t][s];
            if (raw) {
                std::memcpy(pin, src, (size_t)n * (size_t)src_bytes);
            } else {
                int64_t bad = 0;
                if (src_bytes == 8 && dst_bytes == 4) bad = dst_unsigned ? narrow_chunk((const int64_t*)src, (uint32_t*)pin, n, true) : narrow_chunk((const int64_t*)src, (int32_t*)pin, n, false);
                else if (src_bytes == 8) bad = dst_unsigned ? narrow_chunk((const int64_t*)src, (uint16_t*)pin, n, true) : narrow_chunk((const int64_t*)src, (int16_t*)pin, n, false);
                else bad = dst_unsigned ? narrow_chunk((const int32_t*)src, (uint16_t*)pin, n, true) : narrow_chunk((const int32_t*)src, (int16_t*)pin, n, false);
                if (bad) bad_total.fetch_add(bad, std::memory_order_relaxed);
            }
            e = hipMemcpyAsync(static_cast<char*>(dst_dev) + (size_t)i0 * (size_t)dst_bytes, pin, (size_t)n * (size_t)dst_bytes, hipMemcpyHostToDevice, pool->st[t]);
            if (e == hipSuccess) e = hipEventRecord(pool->ev[t][s], pool->st[t]);
            ++uses;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(pool->st[t]);
        if (e != hipSuccess) {
            int expect = (int)hipSuccess;
            first_err.compare_exchange_strong(expect, (int)e);
        }
    };
    std::vector<std::thread> pool_threads;
    try {
        for (int t = 1; t < T; ++t) pool_threads.emplace_back(work, t);
    } catch (...) {  // (thread creation failed: the calling thread does what is left)
    }
    work(0);
    for (auto& th : pool_threads) th.join();
    (void)hipSetDevice(device);
    if (first_err.load() != (int)hipSuccess) return hip_fail((hipError_t)first_err.load(), "dl_stage_to_device");
    if (bad_out_host) *bad_out_host = bad_total.load();
    if (seconds_out_host) *seconds_out_host = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}
