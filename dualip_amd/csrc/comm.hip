// comm.hip -- dl_comm: the one exchange of a column-sharded iteration (see comm.h for the protocol).
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <new>

#include "comm.h"

namespace dl {

// ---------------------------------------------------------------------------------------------------------
// RCCL, opened at run time: the copy PyTorch-ROCm already loaded when there is one (same soname), else the system's
// ---------------------------------------------------------------------------------------------------------
struct UniqueId128 {  // ncclUniqueId (passed by value)
    char b[128];
};
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
};
static RcclApi g_rccl;
static int load_rccl() {
    if (g_rccl.lib) return 0;
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(DL_E_STATE, "librccl.so.1 cannot be opened: %s", dlerror());
    auto sym = [&](const char* n) { return dlsym(lib, n); };
    *(void**)&g_rccl.GetUniqueId = sym("ncclGetUniqueId");
    *(void**)&g_rccl.CommInitRank = sym("ncclCommInitRank");
    *(void**)&g_rccl.CommDestroy = sym("ncclCommDestroy");
    *(void**)&g_rccl.AllReduce = sym("ncclAllReduce");
    *(void**)&g_rccl.GetErrorString = sym("ncclGetErrorString");
    *(void**)&g_rccl.CommCount = sym("ncclCommCount");
    *(void**)&g_rccl.CommUserRank = sym("ncclCommUserRank");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.CommCount || !g_rccl.CommUserRank)
        return fail(DL_E_STATE, "librccl.so.1 lacks an expected entry point");
    g_rccl.lib = lib;
    return 0;
}
static int rccl_fail(int r, const char* what) {
    return fail(DL_E_STATE, "RCCL error %d (%s) in %s", r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what);
}
constexpr int kNcclFloat64 = 8, kNcclSum = 0;  // ncclDataType_t / ncclRedOp_t values of rccl.h

int comm_rccl_allreduce(dl_comm* c, double* buf, int64_t count, hipStream_t st) {
    const int r = g_rccl.AllReduce(buf, buf, (size_t)count, kNcclFloat64, kNcclSum, c->nccl, st);
    if (r != 0) return rccl_fail(r, "ncclAllReduce");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// P2P mailboxes
// ---------------------------------------------------------------------------------------------------------
static inline unsigned long long* flags_of(void* mail, int par) {
    return reinterpret_cast<unsigned long long*>(mail) + (size_t)par * kMaxWorld * kFlagStride;
}
static inline double* slots_of(void* mail, int par, int world, int64_t stride) {
    return reinterpret_cast<double*>(reinterpret_cast<char*>(mail) + kMailHeaderBytes) + (size_t)par * (size_t)world * (size_t)stride;
}

PushArgs comm_push_args(dl_comm* c, unsigned long long seq) {
    PushArgs p;
    const int par = (int)(seq & 1ull);
    for (int r = 0; r < kMaxWorld; ++r) {
        p.dst[r] = nullptr;
        p.flag[r] = nullptr;
    }
    for (int r = 0; r < c->world; ++r) {
        p.dst[r] = slots_of(c->peer[r], par, c->world, c->stride) + (size_t)c->rank * (size_t)c->stride;
        p.flag[r] = flags_of(c->peer[r], par) + (size_t)c->rank * kFlagStride;
    }
    p.world = c->world;
    p.seq = seq;
    p.counter = c->counter;
    p.fenced = c->fenced ? 1 : 0;
    p.fault = 0;
    p.fault_rank = 0;
    if (c->fault_kind && c->fault_seq == seq) {  // test hook: this exchange only
        p.fault = c->fault_kind;
        p.fault_rank = c->fault_rank;
        c->fault_kind = 0;
    }
    return p;
}
MailArgs comm_mail_args(dl_comm* c, unsigned long long seq) {
    MailArgs a;
    const int par = (int)(seq & 1ull);
    a.slots = slots_of(c->mail, par, c->world, c->stride);
    a.flags = flags_of(c->mail, par);
    a.stride = c->stride;
    a.world = c->world;
    a.seq = seq;
    a.dead = c->dead;
    a.timeout_ticks = c->timeout_ticks;
    a.fenced = c->fenced ? 1 : 0;
    a.ticket = c->counter ? c->counter + 1 : nullptr;
    return a;
}

// stand-alone all-reduce, P2P: push the buffer, then gather the sum back into it
__global__ __launch_bounds__(256) void p2p_push_kernel(const double* __restrict__ src, int64_t count, PushArgs p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long h = 0ull;
    if (i < count) push_value(p, i, src[i], h);
    push_finish(p, h);
}
// (the gathering launch checks what it loaded against the slots' announced checksums with a ticket: its last block to arrive
//  compares -- not the hot path; the solver loop's reader does the same without an atomic, agd_kernels.hip)
__global__ __launch_bounds__(256) void p2p_gather_kernel(double* __restrict__ dst, int64_t count, MailArgs a, double scale) {
    const unsigned long long expected = mail_wait(a);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long h = 0ull;
    if (i < count) dst[i] = mail_sum(a, i, h) * scale;
    __shared__ unsigned long long gather_h;
    if (threadIdx.x == 0) gather_h = 0ull;
    __syncthreads();
    h = chk_wave_sum(h);
    if ((threadIdx.x & 63) == 0 && h) atomicAdd(&gather_h, h);
    __syncthreads();
    if (threadIdx.x == 0 && a.ticket) {
        const unsigned long long add = ((gather_h & kChkMask) << kSeqBits) | 1ull;
        const unsigned long long now = __hip_atomic_fetch_add(a.ticket, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
        if ((now & kSeqMask) == (unsigned long long)gridDim.x) {
            __hip_atomic_store(a.ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mail_judge(a.dead, now >> kSeqBits, expected);
        }
    }
}

int comm_push_buffer(const double* src, int64_t count, const PushArgs& push, hipStream_t st) {
    const unsigned blocks = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(p2p_push_kernel, dim3(blocks > 0 ? blocks : 1u), dim3(256), 0, st, src, count, push);
    DL_HIP(hipGetLastError());
    return 0;
}

// ---- soak test of the exchange (dl_comm_selftest) ----
// Round t: every rank waits a pseudo-random time of its own (so ranks arrive in every order, early ranks wait, and a rank can be
// a whole exchange ahead of the slowest reader), pushes  v_r[i] = (r + 1) * ((i + 3 t) % 1021 + 1) + t * (7 r + 3)  and gathers;
// every element of the sum is an exact small integer every rank can form by itself.
__device__ __forceinline__ double soak_value(int r, int64_t i, unsigned long long t) {
    return (double)(r + 1) * (double)((i + 3 * (int64_t)t) % 1021 + 1) + (double)t * (double)(7 * r + 3);
}
__global__ void soak_fill_kernel(double* __restrict__ v, int64_t count, int rank, unsigned long long t, unsigned long long delay_ticks) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && delay_ticks) {  // (the delay sits in the launch that precedes the push, same stream)
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(8);
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) v[i] = soak_value(rank, i, t);
}
__global__ void soak_check_kernel(const double* __restrict__ v, int64_t count, int world, unsigned long long t, unsigned long long* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double want = 0.0;
    for (int r = 0; r < world; ++r) want += soak_value(r, i, t);
    if (v[i] != want) atomicAdd(bad, 1ull);
}

static int p2p_allreduce(dl_comm* c, double* buf, int64_t count, hipStream_t st) {
    const unsigned long long seq = ++c->seq;
    const unsigned blocks = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(p2p_push_kernel, dim3(blocks), dim3(256), 0, st, (const double*)buf, count, comm_push_args(c, seq));
    hipLaunchKernelGGL(p2p_gather_kernel, dim3(blocks), dim3(256), 0, st, buf, count, comm_mail_args(c, seq), c->emu_scale);
    DL_HIP(hipGetLastError());
    return 0;
}

__global__ void scale_kernel(double* __restrict__ v, int64_t n, double s) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] *= s;
}

static void comm_free(dl_comm* c) {
    if (!c) return;
    for (int r = 0; r < kMaxWorld; ++r)
        if (c->opened[r] && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->mail) (void)hipFree(c->mail);
    if (c->counter) (void)hipFree(c->counter);
    if (c->dead) (void)hipFree(c->dead);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->nccl && c->owns_nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nccl);
    if (c->side) (void)hipStreamDestroy(c->side);
    for (hipEvent_t& e : c->ev_ready)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    for (hipEvent_t e : c->prof_start) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->prof_stop) (void)hipEventDestroy(e);
    delete c;
}

static int comm_common_init(dl_comm* c) {
    DL_HIP(hipGetDevice(&c->device));
    DL_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    for (hipEvent_t& e : c->ev_ready) DL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    DL_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    DL_HIP(hipMalloc((void**)&c->dead, sizeof(int)));
    DL_HIP(hipMemset(c->dead, 0, sizeof(int)));
    return 0;
}

}  // namespace dl

using namespace dl;

extern "C" {

int dl_comm_rccl_unique_id(void* id_out_host) {
    if (!id_out_host) return fail(DL_E_ARG, "null argument");
    int rc = load_rccl();
    if (rc) return rc;
    const int r = g_rccl.GetUniqueId(id_out_host);
    if (r != 0) return rccl_fail(r, "ncclGetUniqueId");
    return 0;
}

int dl_comm_create_rccl(dl_comm** out, int32_t world, int32_t rank, const void* unique_id_host, int64_t max_count) {
    if (!out) return fail(DL_E_ARG, "out is null");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !unique_id_host || max_count < 1) return fail(DL_E_ARG, "bad communicator arguments");
    int rc = load_rccl();
    if (rc) return rc;
    dl_comm* c = new (std::nothrow) dl_comm();
    if (!c) return fail(DL_E_NOMEM, "out of host memory");
    c->backend = DL_COMM_RCCL;
    c->world = world;
    c->rank = rank;
    c->max_count = max_count;
    c->stride = (max_count + 7) / 8 * 8;
    rc = comm_common_init(c);
    if (rc) {
        comm_free(c);
        return rc;
    }
    UniqueId128 id;
    memcpy(id.b, unique_id_host, sizeof(id.b));
    const int r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
    if (r != 0) {
        comm_free(c);
        return rccl_fail(r, "ncclCommInitRank");
    }
    c->owns_nccl = true;
    *out = c;
    return 0;
}

int dl_comm_adopt_rccl(dl_comm** out, void* nccl_comm, int64_t max_count) {
    if (!out) return fail(DL_E_ARG, "out is null");
    *out = nullptr;
    if (!nccl_comm || max_count < 1) return fail(DL_E_ARG, "bad communicator arguments");
    int rc = load_rccl();
    if (rc) return rc;
    int world = 0, rank = 0;
    int r = g_rccl.CommCount(nccl_comm, &world);
    if (r == 0) r = g_rccl.CommUserRank(nccl_comm, &rank);
    if (r != 0) return rccl_fail(r, "ncclCommCount");
    dl_comm* c = new (std::nothrow) dl_comm();
    if (!c) return fail(DL_E_NOMEM, "out of host memory");
    c->backend = DL_COMM_RCCL;
    c->world = world;
    c->rank = rank;
    c->max_count = max_count;
    c->stride = (max_count + 7) / 8 * 8;
    c->nccl = nccl_comm;
    c->owns_nccl = false;
    rc = comm_common_init(c);
    if (rc) {
        comm_free(c);
        return rc;
    }
    *out = c;
    return 0;
}

int dl_comm_p2p_begin(dl_comm** out, int32_t world, int32_t rank, int64_t max_count, void* ipc_handle_out_host) {
    if (!out) return fail(DL_E_ARG, "out is null");
    *out = nullptr;
    if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world || max_count < 1 || !ipc_handle_out_host)
        return fail(DL_E_ARG, "bad communicator arguments (the P2P exchange serves up to %d ranks of one node)", kMaxWorld);
    static_assert(sizeof(hipIpcMemHandle_t) == DL_IPC_HANDLE_BYTES, "DL_IPC_HANDLE_BYTES must match hipIpcMemHandle_t");
    dl_comm* c = new (std::nothrow) dl_comm();
    if (!c) return fail(DL_E_NOMEM, "out of host memory");
    c->backend = DL_COMM_P2P;
    c->world = world;
    c->rank = rank;
    c->max_count = max_count;
    c->stride = (max_count + 7) / 8 * 8;
    int rc = comm_common_init(c);
    if (rc) {
        comm_free(c);
        return rc;
    }
    if (const char* e = getenv("DUALIP_COMM_TIMEOUT_MS")) {
        const long long ms = atoll(e);
        if (ms > 0) c->timeout_ticks = (unsigned long long)ms * 100000ull;
    }
    c->mail_bytes = kMailHeaderBytes + sizeof(double) * 2 * (size_t)world * (size_t)c->stride;
    // fine-grained (uncached) device memory: stores from other devices and from other processes become visible to a
    // running kernel, which ordinary (coarse-grained) allocations only guarantee at kernel boundaries
    hipError_t e = hipExtMallocWithFlags(&c->mail, c->mail_bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&c->mail, c->mail_bytes, hipDeviceMallocFinegrained);
    }
    if (e == hipSuccess) e = hipMemset(c->mail, 0, c->mail_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&c->counter, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(c->counter, 0, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void**)&c->scratch, sizeof(double) * (size_t)c->stride);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t hd;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&hd, c->mail);
    if (e != hipSuccess) {
        comm_free(c);
        return hip_fail(e, "P2P mailbox allocation / hipIpcGetMemHandle");
    }
    memcpy(ipc_handle_out_host, &hd, sizeof(hd));
    c->peer[rank] = c->mail;
    *out = c;
    return 0;
}

int dl_comm_p2p_connect(dl_comm* c, const void* all_handles_host) {
    if (!c || c->backend != DL_COMM_P2P || !all_handles_host) return fail(DL_E_ARG, "bad argument");
    if (c->connected) return fail(DL_E_STATE, "already connected");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t hd;
        memcpy(&hd, reinterpret_cast<const char*>(all_handles_host) + (size_t)r * sizeof(hd), sizeof(hd));
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return hip_fail(e, "hipIpcOpenMemHandle");
        c->peer[r] = p;
        c->opened[r] = true;
    }
    c->connected = true;
    return 0;
}

int dl_comm_destroy(dl_comm* c) {
    comm_free(c);
    return 0;
}

int64_t dl_comm_info(const dl_comm* c, int what) {
    if (!c) return -1;
    switch (what) {
        case 0: return c->backend;
        case 1: return c->world;
        case 2: return c->rank;
        case 3: return (int64_t)c->seq;
        case 4: return c->max_count;
        case 5: return c->fenced ? 1 : 0;
        case 6: return c->device;  // the HIP device ordinal the communicator was created on
        case 7: {                  // ranks the RCCL communicator itself reports (-1: not an RCCL communicator)
            int n = -1;
            if (c->backend == DL_COMM_RCCL && c->nccl && g_rccl.CommCount && g_rccl.CommCount(c->nccl, &n) != 0) n = -1;
            return n;
        }
        case 8: {                  // this rank as the RCCL communicator reports it
            int r = -1;
            if (c->backend == DL_COMM_RCCL && c->nccl && g_rccl.CommUserRank && g_rccl.CommUserRank(c->nccl, &r) != 0) r = -1;
            return r;
        }
        default: return -1;
    }
}

int dl_comm_set_emulation(dl_comm* c, double scale) {
    if (!c || !(scale > 0.0)) return fail(DL_E_ARG, "bad argument");
    c->emu_scale = scale;
    return 0;
}

int dl_comm_inject_fault(dl_comm* c, int32_t kind, int32_t target_rank, uint64_t at_exchange) {
    if (!c || c->backend != DL_COMM_P2P) return fail(DL_E_ARG, "fault injection needs a P2P communicator");
    if (kind < 0 || kind > 2 || target_rank < 0 || target_rank >= c->world) return fail(DL_E_ARG, "bad fault description");
    c->fault_kind = kind;
    c->fault_rank = target_rank;
    c->fault_seq = at_exchange;
    return 0;
}

int dl_comm_set_fenced(dl_comm* c, int fenced) {
    if (!c) return fail(DL_E_ARG, "null communicator");
    c->fenced = fenced != 0;
    return 0;
}

int dl_comm_reset(dl_comm* c, dl_stream_t stream) {
    if (!c) return fail(DL_E_ARG, "null communicator");
    hipStream_t st = (hipStream_t)stream;
    DL_HIP(hipStreamSynchronize(st));
    DL_HIP(hipMemsetAsync(c->dead, 0, sizeof(int), st));
    if (c->counter) DL_HIP(hipMemsetAsync(c->counter, 0, 2 * sizeof(unsigned long long), st));
    DL_HIP(hipStreamSynchronize(st));
    return 0;
}

int dl_comm_selftest(dl_comm* c, int32_t rounds, uint64_t seed, int64_t max_delay_us, int64_t* mismatches_host, dl_stream_t stream) {
    if (!c || rounds < 1 || !mismatches_host || max_delay_us < 0) return fail(DL_E_ARG, "bad argument");
    *mismatches_host = -1;
    hipStream_t st = (hipStream_t)stream;
    const int64_t count = c->max_count;
    double* buf = nullptr;
    unsigned long long* bad = nullptr;
    DL_HIP(hipMalloc((void**)&buf, sizeof(double) * (size_t)count));
    hipError_t e = hipMalloc((void**)&bad, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(bad, 0, sizeof(unsigned long long), st);
    int rc = e == hipSuccess ? 0 : hip_fail(e, "selftest allocation");
    const unsigned blocks = (unsigned)((count + 255) / 256);
    unsigned long long x = seed * 0x9E3779B97F4A7C15ull + (unsigned long long)(c->rank + 1) * 0xBF58476D1CE4E5B9ull;
    for (int t = 0; t < rounds && !rc; ++t) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;  // xorshift: a different delay sequence on every rank
        const unsigned long long r = x * 0x2545F4914F6CDD1Dull;
        // a third of the rounds without delay (back-to-back exchanges), the rest uniformly up to max_delay_us
        const unsigned long long delay_ticks = (r % 3 == 0 || max_delay_us == 0) ? 0ull : ((r >> 8) % (unsigned long long)(max_delay_us * 100 + 1));
        hipLaunchKernelGGL(soak_fill_kernel, dim3(blocks), dim3(256), 0, st, buf, count, c->rank, (unsigned long long)t, delay_ticks);
        rc = dl_allreduce_sum(c, buf, count, stream);
        if (rc) break;
        hipLaunchKernelGGL(soak_check_kernel, dim3(blocks), dim3(256), 0, st, (const double*)buf, count, c->world, (unsigned long long)t, bad);
    }
    unsigned long long nbad = 0;
    if (!rc) {
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&nbad, bad, sizeof(nbad), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = hip_fail(e, "selftest");
    } else {
        (void)hipStreamSynchronize(st);
    }
    (void)hipFree(buf);
    if (bad) (void)hipFree(bad);
    if (rc) return rc;
    *mismatches_host = (int64_t)nbad;
    return 0;
}

int dl_comm_set_timeout_ms(dl_comm* c, int64_t ms) {
    if (!c || ms <= 0) return fail(DL_E_ARG, "bad argument");
    c->timeout_ticks = (unsigned long long)ms * 100000ull;  // 100 MHz wall clock
    return 0;
}

int dl_comm_status(dl_comm* c, int32_t* dead_out_host, dl_stream_t stream) {
    if (!c || !dead_out_host) return fail(DL_E_ARG, "null argument");
    int dead = 0;
    DL_HIP(hipMemcpyAsync(&dead, c->dead, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    DL_HIP(hipStreamSynchronize((hipStream_t)stream));
    *dead_out_host = dead;
    return 0;
}

int dl_comm_check(dl_comm* c, dl_stream_t stream) {
    if (!c) return fail(DL_E_ARG, "null communicator");
    int dead = 0;
    DL_HIP(hipMemcpyAsync(&dead, c->dead, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    DL_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (dead == kDeadChecksum)
        return fail(DL_E_STATE, "a P2P exchange delivered a slot whose payload does not match the checksum its sender announced (data behind its flag, "
                                "a torn or corrupted slot): the results of this run are invalid%s", c->fenced ? "" : " -- DUALIP_COMM=p2p-fenced or rccl orders the exchange by the book");
    if (dead) return fail(DL_E_STATE, "a P2P exchange timed out waiting for another rank's partial sums: the results of this run are invalid");
    return 0;
}

int dl_allreduce_sum(dl_comm* c, double* buf, int64_t count, dl_stream_t stream) {
    if (!c || !buf || count < 0) return fail(DL_E_ARG, "bad argument");
    if (count > c->max_count) return fail(DL_E_ARG, "count %lld exceeds the communicator's capacity %lld", (long long)count, (long long)c->max_count);
    if (count == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (c->backend == DL_COMM_RCCL) {
        int rc = comm_rccl_allreduce(c, buf, count, st);
        if (rc) return rc;
        if (c->emu_scale != 1.0) {
            hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, buf, count, c->emu_scale);
            DL_HIP(hipGetLastError());
        }
        return 0;
    }
    if (!c->connected) return fail(DL_E_STATE, "P2P communicator is not connected (dl_comm_p2p_connect)");
    return p2p_allreduce(c, buf, count, st);
}

int dl_comm_profile(dl_comm* c, int enable) {
    if (!c) return fail(DL_E_ARG, "null communicator");
    c->prof_on = enable != 0;
    c->prof_stride = enable > 1 ? enable : 1;
    c->prof_seen = 0;
    c->prof_used = 0;
    return 0;
}

int dl_comm_profile_read(dl_comm* c, double* total_ms_host, int64_t* exchanges_host) {
    if (!c || !total_ms_host || !exchanges_host) return fail(DL_E_ARG, "null argument");
    double total = 0.0;
    for (size_t i = 0; i < c->prof_used; ++i) {
        DL_HIP(hipEventSynchronize(c->prof_stop[i]));
        float ms = 0.f;
        DL_HIP(hipEventElapsedTime(&ms, c->prof_start[i], c->prof_stop[i]));
        total += (double)ms;
    }
    *total_ms_host = total;
    *exchanges_host = (int64_t)c->prof_used;
    return 0;
}

}  // extern "C"
