// comm.h -- the exchange step of the column-sharded iteration (include/dualip_hip.h: dl_comm, dl_allreduce_sum,
// dl_agd_run_matching_sharded).  Replaces the reference's per-iteration collectives: three torch.distributed.reduce calls, a
// barrier and two broadcasts (src/dualip/objectives/matching.py:272-277, src/dualip/optimizers/agd.py:204-206) by ONE
// sum-all-reduce of [A x (m) | c.x | sum x^2] doubles after which every rank applies the identical update.
//
// Two back-ends behind one handle:
//   RCCL  ncclAllReduce(double, sum) on the caller's stream (or a handle-owned side stream when the local shard is split
//         into blocks); librccl is opened at run time (dlopen of the soname PyTorch-ROCm already loaded).
//   P2P   one-shot all-to-all over hipIpc-mapped mailboxes: every rank owns a fine-grained buffer with one slot per sending
//         rank (double buffered by the parity of a sequence number).  The slab-reduction kernel of the fused pass stores its
//         sums straight into slot [parity][my rank] of EVERY rank's mailbox (remote stores over xGMI), releases them with a
//         system-scope fence and raises the flag word of that slot; the step kernel of every rank polls its OWN flags (local
//         memory), acquires, and adds the slots in rank order 0..W-1 -- so all ranks obtain bit-identical sums without a
//         second hop, and an iteration costs no collective launch at all.
#pragma once
#include "agd_step.h"
#include "common.h"

namespace dl {

constexpr int kMaxWorld = 16;          // ranks of one node (8 on an MI355X node)
constexpr int kFlagStride = 8;         // one 64-byte line per flag word (uint64 units)
constexpr size_t kMailHeaderBytes = 2 * kMaxWorld * kFlagStride * sizeof(unsigned long long);  // flags [2][kMaxWorld]

// ---- payload checksum (round 4) ----
// A flag word carries, besides the sequence number of the exchange it announces, a 40-bit checksum of the slot's payload:
//     flag = checksum << 24 | (seq mod 2^24),      checksum = sum_i hash40(bits of element i)  mod 2^40.
// One 64-bit store, so the checksum arrives WITH the flag and costs the writer no extra round trip: every pushing block adds the
// hashes of what it stored to the upper 40 bits of the same 64-bit arrival counter whose lower 24 bits count the blocks (one
// atomic per block, as before); the last arriver reads the total off the returned value.  The reader hashes what it actually
// loaded and compares (stats kernel: per-block partial sums, compared by the step that consumes them -- no atomic, no extra
// launch; stand-alone all-reduce: a ticket).  A mismatch sets the communicator's sticky `dead` word to kDeadChecksum: data that
// arrived after its flag (the ordering the unfenced variant relies on, violated), a torn or corrupted slot, a stale slot of the
// same parity (the checksum of exchange seq - 2 does not carry seq's flag: the flag's sequence field differs, and a slot that is
// stale in part hashes differently unless the stale values equal the new ones -- in which case nothing is wrong).
constexpr int kSeqBits = 24;
constexpr unsigned long long kSeqMask = (1ull << kSeqBits) - 1ull;
constexpr unsigned long long kChkMask = (1ull << 40) - 1ull;
constexpr int kDeadTimeout = 1, kDeadChecksum = 2;
__host__ __device__ inline unsigned long long chk_hash(double v) {
    unsigned long long b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = (unsigned long long)__double_as_longlong(v);
#else
    __builtin_memcpy(&b, &v, sizeof(b));
#endif
    return (b * 0x9E3779B97F4A7C15ull) >> 24;  // top 40 bits of a multiplicative hash: every input bit reaches them
}
__host__ __device__ inline unsigned long long flag_word(unsigned long long seq, unsigned long long chk) { return ((chk & kChkMask) << kSeqBits) | (seq & kSeqMask); }
// has the flag reached exchange `seq`?  (modular: ranks are never 2^23 exchanges apart; a zeroed mailbox reads as "exchange 0")
__host__ __device__ inline bool flag_arrived(unsigned long long flag, unsigned long long seq) { return ((flag - seq) & kSeqMask) < (1ull << (kSeqBits - 1)); }
__host__ __device__ inline unsigned long long flag_chk(unsigned long long flag) { return (flag >> kSeqBits) & kChkMask; }

// wave-wide sum of the lanes' hash sums, mod 2^40, on the DPP unit: the 40 bits travel as two 20-bit halves in 32-bit registers (64 lanes
// x 2^20 < 2^32), so the reduction is two 32-bit butterflies instead of twelve ds_bpermute round trips of a 64-bit one
__device__ __forceinline__ unsigned long long chk_wave_sum(unsigned long long h) {
    const uint32_t lo = wave_allreduce_dpp((uint32_t)(h & 0xFFFFFull), OpAdd());
    const uint32_t hi = wave_allreduce_dpp((uint32_t)((h >> 20) & 0xFFFFFull), OpAdd());
    return ((unsigned long long)lo + ((unsigned long long)hi << 20)) & kChkMask;
}

// Where one rank's sums go: slot (parity, my rank) of every rank's mailbox.
struct PushArgs {
    double* dst[kMaxWorld];               // slot base in rank r's mailbox
    unsigned long long* flag[kMaxWorld];  // its flag word
    int world;
    unsigned long long seq;
    unsigned long long* counter;          // arrival counter of the pushing launch (device memory, zero between launches): blocks in bits 0..23, checksum above
    int fenced;                           // 1: the by-the-book variant (system-scope release / acquire fences around the flags)
    int fault;                            // test hook (dl_comm_inject_fault), this exchange only: 0 none, 1 element 0 of the slot in rank `fault_rank`'s
    int fault_rank;                       //   mailbox is stored with a flipped bit, 2 the data stores to that rank are dropped (its slot stays stale)
};

// What a rank reads: its own mailbox of the parity.
struct MailArgs {
    const double* slots;                  // [world][stride]
    const unsigned long long* flags;      // [world] at kFlagStride
    int64_t stride;
    int world;
    unsigned long long seq;
    int* dead;                            // sticky: a wait timed out (the run's results are invalid)
    unsigned long long timeout_ticks;     // 100 MHz wall clock
    int fenced;                           // see PushArgs
    unsigned long long* ticket;           // stand-alone gather only: arrival counter + checksum of what the gathering launch loaded
};

// Memory-ordering rules of the exchange.  Two variants, chosen per communicator (dl_comm_set_fenced; the creation-time soak test
// of dualip_amd/utils/comm.py falls back from the first to the second, then to RCCL):
//
// DEFAULT (fenced = 0) -- no fence anywhere.  A system-scope release writes back every dirty L2 line of the device and an acquire
// invalidates the caches, once per workgroup: measured 61 us per exchange against 19 us for a one-rank RCCL call.  Instead:
//   * the mailbox is UNCACHED fine-grained memory, and every access to it is a system-scope (sc0 sc1) store or load, i.e.
//     written through to / fetched from memory: nothing about it ever sits dirty or stale in an L1 or L2;
//   * a store is complete -- visible to every agent -- when the issuing wavefront's vmcnt has counted it down, so
//     "s_waitcnt vmcnt(0)" after the data stores orders them before everything the wavefront does next (this is exactly the
//     wait a release fence compiles to on gfx942/gfx950; what the default leaves out is the L2 write-back, which has nothing to
//     write back for write-through stores);
//   * workgroups of the pushing launch count their arrival with a device-scope atomic only AFTER that wait; the one that
//     arrives last therefore knows all data of the launch is in place when it writes the flags;
//   * the reader polls its flags and then issues its data loads (program order + a workgroup barrier in between); the loads
//     bypass the caches, so there is nothing an acquire's invalidate would have to drop.
//   This is the hardware's behaviour, not a guarantee of the HIP / LLVM memory model (relaxed atomics carry no ordering there).
//
// FENCED (fenced = 1) -- the same protocol written as the memory model asks: every thread of the pushing launch issues a
// system-scope release fence after its data stores, the arrival counter is an acq_rel atomic, the last arriver releases again
// before the flag stores; the reader's poll is followed by a system-scope acquire fence in every thread.  Slow (the measured
// 61 us) and always correct by the book; the fallback when the soak test of the default variant fails on a machine.
// `h` accumulates the hashes of what this thread pushed (handed to push_finish).
__device__ __forceinline__ void push_value(const PushArgs& p, int64_t i, double v, unsigned long long& h) {
    h += chk_hash(v);
    for (int r = 0; r < p.world; ++r) {
        double vs = v;
        if (p.fault && r == p.fault_rank) {  // (test hook: cold)
            if (p.fault == 2) continue;
            if (i == 0) vs = __longlong_as_double(__double_as_longlong(v) ^ 0x0000000100000000ll);
        }
        __hip_atomic_store(p.dst[r] + i, vs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Called by EVERY thread of EVERY block of the pushing launch after its push_value calls, with the thread's hash sum.
__device__ __forceinline__ void push_finish(const PushArgs& p, unsigned long long h) {
    __shared__ unsigned long long push_hw[16];  // one hash sum per wavefront of the block (blocks of up to 1024 threads): no zeroing, no LDS atomic
    h = chk_wave_sum(h);
    if ((threadIdx.x & 63) == 0) push_hw[threadIdx.x >> 6] = h;
    if (p.fenced) __atomic_thread_fence(__ATOMIC_RELEASE);  // (system scope: HIP's default for __atomic_thread_fence)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's stores have landed
    __syncthreads();
    unsigned long long push_h = 0ull;
    if (threadIdx.x == 0)
        for (unsigned int q = 0; q < (blockDim.x + 63u) / 64u; ++q) push_h += push_hw[q];
    if (threadIdx.x == 0) {
        const unsigned long long add = ((push_h & kChkMask) << kSeqBits) | 1ull;  // (a carry out of the top drops: arithmetic mod 2^40 up there)
        unsigned long long old;
        if (p.fenced) old = __hip_atomic_fetch_add(p.counter, add, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        else old = __hip_atomic_fetch_add(p.counter, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long now = old + add;
        if ((now & kSeqMask) == (unsigned long long)gridDim.x) {  // last arriver: every block's data is in place
            __hip_atomic_store(p.counter, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch (stream ordered)
            const unsigned long long word = flag_word(p.seq, now >> kSeqBits);
            if (p.fenced) {
                for (int r = 0; r < p.world; ++r) __hip_atomic_store(p.flag[r], word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                for (int r = 0; r < p.world; ++r) __hip_atomic_store(p.flag[r], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Whole block: returns once every rank's slot of this exchange has arrived (or the wait timed out: *dead = kDeadTimeout).
// Returns (to every thread) the sum of the slots' announced checksums: what the hashes of everything read from them must add up to.
__device__ __forceinline__ unsigned long long mail_wait(const MailArgs& a) {
    __shared__ unsigned long long mail_chk[kMaxWorld];  // the checksum each polled flag announced (no zeroing: every polling thread writes its own)
    if ((int)threadIdx.x < a.world) {
        unsigned long long word = 0ull;
        if (__hip_atomic_load(a.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            const unsigned long long* f = a.flags + (size_t)threadIdx.x * kFlagStride;
            word = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (!flag_arrived(word, a.seq)) {
                const unsigned long long t0 = wall_clock64();
                while (!flag_arrived(word = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), a.seq)) {
                    __builtin_amdgcn_s_sleep(2);
                    if (wall_clock64() - t0 > a.timeout_ticks) {
                        __hip_atomic_store(a.dead, kDeadTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
        }
        // (the word read is exchange seq's own: a writer one exchange ahead raises the OTHER parity's flag, and none can be two ahead)
        mail_chk[threadIdx.x] = flag_chk(word);
    }
    __syncthreads();
    if (a.fenced) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    unsigned long long expected = 0ull;
    for (int r = 0; r < a.world; ++r) expected += mail_chk[r];
    return expected & kChkMask;
}

// Sum of element i over the ranks' slots, in rank order (identical on every rank); `h` accumulates the hashes of the loaded
// values.  All loads of a batch of eight ranks are in flight before the first is added: the mailbox is uncached, every load is a
// memory round trip.
__device__ __forceinline__ double mail_sum(const MailArgs& a, int64_t i, unsigned long long& h) {
    double acc = 0.0;
    for (int r0 = 0; r0 < a.world; r0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u < a.world ? r0 + u : a.world - 1;
            v[u] = __hip_atomic_load(a.slots + (int64_t)r * a.stride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc += (r0 + u < a.world) ? v[u] : 0.0;
            h += (r0 + u < a.world) ? chk_hash(v[u]) : 0ull;
        }
    }
    return acc;
}

// Verdict on one exchange: `got` = sum of the hashes of everything the reader loaded, `expected` = mail_wait's return.
__device__ __forceinline__ void mail_judge(int* dead, unsigned long long got, unsigned long long expected) {
    if (((got - expected) & kChkMask) != 0ull && __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        __hip_atomic_store(dead, kDeadChecksum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace dl

struct dl_comm {
    int backend = 0;  // DL_COMM_*
    int world = 1, rank = 0;
    int device = 0;
    int64_t max_count = 0;
    int64_t stride = 0;  // doubles per slot
    double emu_scale = 1.0;  // developer aid (dl_comm_set_emulation): the all-reduced sums are multiplied by this
    // RCCL
    void* nccl = nullptr;
    bool owns_nccl = false;
    hipStream_t side = nullptr;  // owned: collectives of all but the last block of a split shard run here
    hipEvent_t ev_ready[4] = {nullptr, nullptr, nullptr, nullptr}, ev_done = nullptr;
    // P2P
    void* mail = nullptr;                         // owned, fine-grained: header + data [2][world][stride]
    size_t mail_bytes = 0;
    void* peer[dl::kMaxWorld] = {};               // mapped mailbox of every rank (own entry = mail)
    bool opened[dl::kMaxWorld] = {};
    bool connected = false;
    bool fenced = false;                          // the by-the-book variant of the protocol (see the ordering rules above)
    unsigned long long seq = 0;                   // exchanges issued
    unsigned long long* counter = nullptr;        // owned: [0] the pushing launches' arrival counter + checksum, [1] the stand-alone gather's ticket
    int* dead = nullptr;                          // owned: 0, kDeadTimeout or kDeadChecksum (sticky)
    int fault_kind = 0, fault_rank = 0;           // test hook (dl_comm_inject_fault): applied to exchange fault_seq, once
    unsigned long long fault_seq = 0;
    unsigned long long timeout_ticks = 2000000000ull;  // 20 s at 100 MHz
    double* scratch = nullptr;                    // owned, double[stride]: result staging of the stand-alone all-reduce
    // measurement (dl_comm_profile): event pairs around the exchanges of dl_agd_run_matching_sharded
    bool prof_on = false;
    int prof_stride = 1;
    uint64_t prof_seen = 0;
    size_t prof_used = 0;
    std::vector<hipEvent_t> prof_start, prof_stop;
};

namespace dl {
// Source of A x for one optimiser step (agd_kernels.hip: launch_agd_step)
struct StepSource {
    dl_matching* slabs = nullptr;   // the handle whose integer slabs hold this launch's sums (single-device loop), or
    const double* packed[4] = {nullptr, nullptr, nullptr, nullptr};  // n_packed reduced buffers to add up, or
    int n_packed = 0;
    const MailArgs* mail = nullptr;  // this rank's mailbox of the P2P exchange
    double scale = 1.0;              // factor on exchanged sums (emulation aid)
    dl_matching* hot = nullptr;      // handle under the hot-rows plan that wants the next dual vector in renumbered order, or null
};
int launch_agd_step(dl_agd* s, const StepSource& src, const void* b, double gamma, int64_t iter, int decay_now, double decay_factor, hipStream_t st);
int launch_agd_stats(dl_agd* s, const StepSource& src, const void* b, hipStream_t st);
int launch_agd_apply(dl_agd* s, const StepSource& src, const PendingStep& ps, hipStream_t st);
const double* agd_step_scal(const dl_agd* s, const StepSource& src);
int matching_reduce(dl_matching* h, double* packed, int mode, const PushArgs* push, hipStream_t st, int push_accumulate);
// pending != null && pending->valid: the launch's prologue applies that step first (the dual vector it stages is the NEW x; agd
// must be the optimiser the step belongs to); the caller rotates the optimiser's buffers afterwards (agd_rotate)
int matching_launch_fused(dl_matching* h, const void* lambda, double gamma, void* x_out, hipStream_t st, uint64_t owner_uid, const dl_agd* agd = nullptr,
                          const PendingStep* pending = nullptr);
bool matching_can_fuse_apply(const dl_matching* h);
PushArgs comm_push_args(dl_comm* c, unsigned long long seq);
MailArgs comm_mail_args(dl_comm* c, unsigned long long seq);
int comm_rccl_allreduce(dl_comm* c, double* buf, int64_t count, hipStream_t st);
// P2P: push `count` doubles of a plain device buffer as this rank's contribution to exchange `push.seq` (a rank without
// non-zeros still has to take part: the others wait for its slot)
int comm_push_buffer(const double* src, int64_t count, const PushArgs& push, hipStream_t st);
}  // namespace dl
