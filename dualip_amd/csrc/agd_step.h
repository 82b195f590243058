// agd_step.h -- the arithmetic of one accelerated-gradient step (agd.py:163-187; agd_utils.py:12-89), shared by the stand-alone
// apply kernel (agd_kernels.hip) and by the fused matching pass, which can run it in the prologue of the NEXT iteration's launch
// (every workgroup stages the whole dual vector anyway, so it can form the new iterate itself instead of reading what a separate
// launch wrote: one launch and one boundary less per iteration -- on for handles below about 19M entities, see fused_common.h and
// matching_kernels.hip: matching_can_fuse_apply).
#pragma once
#include "common.h"
#include "wave.h"

namespace dl {

constexpr int kLipsMax = 14;   // max_history_length - 1 (optimizers/agd_utils.py:69)
constexpr int kStatCols = 6;   // per-workgroup partials of the stats launch: dvtg, gmax, spos, g2, dg2, dy2
constexpr int kApplyThreads = 1024;

struct AgdDevState {
    double max_step;
    double initial_step;
    double last_step;
    int32_t n_lips;  // valid entries in the ring, oldest first starting at head
    int32_t head;
    int64_t steps_done;
    double lips[kLipsMax];  // values rounded to the working precision
};

template <class T>
struct ApplyArgs {
    int64_t m;
    const double* __restrict__ partial_stats;  // [n_blocks][kStatCols], from the stats launch
    int n_blocks;
    const T* __restrict__ g_new;               // gradient written by the stats launch
    const double* __restrict__ scal;           // c.x, sum x^2
    const AgdDevState* st_in;
    AgdDevState* st_out;
    double* __restrict__ log_row;
    double gamma;
    int decay_now;
    double decay_factor;
    // update
    const T* __restrict__ x;
    T* __restrict__ x_next;
    const T* __restrict__ y;
    T* __restrict__ y_new;
    const uint8_t* __restrict__ eq_mask;
    const float* __restrict__ beta;
    int64_t iter;
    T* __restrict__ x_perm;              // or null
    const int32_t* __restrict__ perm;    // caller's row -> renumbered row
    // P2P exchange (comm.h): hash sums of what the stats launch read from the mailbox -- [n_blocks] per block, then the two scalars',
    // then the total the senders announced -- and the communicator's sticky error word; null when the sums came from elsewhere
    const unsigned long long* chk;
    int* chk_dead;
};

// The scalars of the step, identical in every wavefront that calls it: sums the stats partials in a fixed order (no LDS, no
// barrier; four rows per lane are loaded before the first is used), runs calculate_step_size (agd_utils.py:65-89) with the
// Lipschitz ring held one entry per lane, and returns the step.  (The six sums are reduced on the DPP unit, wave.h: 72 double-width
// ds_bpermute round trips before round 3.)  `writer` (one workgroup of the launch) stores the next
// optimiser state and the log row; tid = thread index inside that workgroup.
template <class T>
__device__ __forceinline__ double agd_step_scalars(const ApplyArgs<T>& p, int lane, bool writer, int tid) {
    const AgdDevState& si = *p.st_in;
    const bool has_prev = si.steps_done > 0;
    const double ring = si.lips[lane < kLipsMax ? lane : kLipsMax - 1];
    double dvtg = 0.0, gmax = -INFINITY, spos = 0.0, g2 = 0.0, dg2 = 0.0, dy2 = 0.0;
    {
        constexpr int kU = 4;
        for (int k0 = lane; k0 < p.n_blocks; k0 += 64 * kU) {
            double o[kU][kStatCols];
            bool in[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int k = k0 + 64 * u;
                in[u] = k < p.n_blocks;
                const double* src = p.partial_stats + (int64_t)(in[u] ? k : p.n_blocks - 1) * kStatCols;
#pragma unroll
                for (int c = 0; c < kStatCols; ++c) o[u][c] = src[c];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                dvtg += in[u] ? o[u][0] : 0.0;
                gmax = (in[u] && o[u][1] > gmax) ? o[u][1] : gmax;
                spos += in[u] ? o[u][2] : 0.0;
                g2 += in[u] ? o[u][3] : 0.0;
                dg2 += in[u] ? o[u][4] : 0.0;
                dy2 += in[u] ? o[u][5] : 0.0;
            }
        }
        dvtg = wave_allreduce_dpp(dvtg, OpAdd());
        gmax = wave_allreduce_dpp(gmax, OpMax());
        spos = wave_allreduce_dpp(spos, OpAdd());
        g2 = wave_allreduce_dpp(g2, OpAdd());
        dg2 = wave_allreduce_dpp(dg2, OpAdd());
        dy2 = wave_allreduce_dpp(dy2, OpAdd());
    }
    int n_lips = si.n_lips, head = si.head, slot = -1;
    double L = 0.0;
    if (has_prev) {
        const T num = (T)sqrt(dg2), den = (T)sqrt(dy2);
        L = (double)(T)(num / den);  // estimate_lipschitz_constant; x/0 -> inf, 0/0 -> nan as in torch
        if (n_lips == kLipsMax) {
            slot = head;  // overwrite the oldest
            head = (head + 1) % kLipsMax;
        } else {
            slot = (head + n_lips) % kLipsMax;
            n_lips += 1;
        }
    }
    const double ring_new = lane == slot ? L : ring;  // lanes >= kLipsMax carry padding
    double step;
    if (n_lips < kLipsMax) {
        step = si.initial_step;  // incomplete history (agd_utils.py:57-58)
    } else {
        // builtins.max over the list, oldest first: NaN if the oldest entry is NaN, else the maximum of the non-NaN entries
        const double first = bperm(head, ring_new);
        const double cand_l = (lane < kLipsMax && !isnan(ring_new)) ? ring_new : -INFINITY;
        const double mx = wave_allreduce_dpp(cand_l, OpMax());
        const double lmax = isnan(first) ? first : mx;
        if (isnan(lmax) || isinf(lmax)) step = si.initial_step;
        else {
            const double cand = lmax != 0.0 ? 1.0 / lmax : si.max_step;
            step = cand < si.max_step ? cand : si.max_step;
        }
    }
    if (writer && tid < 64 && p.chk) {  // (wave-uniform: one wavefront of one workgroup) did the exchange deliver what its senders announced?
        unsigned long long got = 0ull;
        for (int k = lane; k < p.n_blocks + 1; k += 64) got += p.chk[k];
        {   // (mod 2^40 as two 20-bit halves: two 32-bit DPP butterflies, comm.h: chk_wave_sum)
            const uint32_t lo = wave_allreduce_dpp((uint32_t)(got & 0xFFFFFull), OpAdd());
            const uint32_t hi = wave_allreduce_dpp((uint32_t)((got >> 20) & 0xFFFFFull), OpAdd());
            got = (unsigned long long)lo + ((unsigned long long)hi << 20);
        }
        const unsigned long long expected = p.chk[p.n_blocks + 1];
        if (lane == 0 && ((got - expected) & ((1ull << 40) - 1ull)) != 0ull && *p.chk_dead == 0) *p.chk_dead = 2;  // comm.h: kDeadChecksum
    }
    if (writer && tid < kLipsMax) p.st_out->lips[tid] = ring_new;
    if (writer && tid == 0) {
        AgdDevState& so = *p.st_out;
        so.initial_step = si.initial_step;
        so.max_step = p.decay_now ? step * p.decay_factor : si.max_step;  // agd.py:106-107
        so.last_step = step;
        so.n_lips = n_lips;
        so.head = head;
        so.steps_done = si.steps_done + 1;
        if (p.log_row) {
            const T nrm = (T)sqrt(p.scal[1]);
            const T reg = (T)((T)(p.gamma / 2.0) * (T)(nrm * nrm));  // (gamma/2) * norm(x)**2, matching.py:157
            const T obj0 = (T)p.scal[0];
            const T dv = (T)dvtg;
            const T obj = (T)((T)(obj0 + reg) + dv);                 // matching.py:33
            p.log_row[0] = (double)obj;
            p.log_row[1] = step;
            p.log_row[2] = (double)reg;
            p.log_row[3] = (double)dv;
            p.log_row[4] = (p.m > 0 && gmax > 0.0) ? (double)(T)gmax : 0.0;  // builtins.max(max(grad), 0), matching.py:168
            p.log_row[5] = (double)(T)spos;
            p.log_row[6] = (double)(T)sqrt(g2);
            p.log_row[7] = (double)obj0;
        }
    }
    return step;
}

// Projected ascent step + momentum of one row (agd.py:181-184): y+ = P(x + step g), x+ = (1 - beta) y+ + beta y.
template <class T>
__device__ __forceinline__ void agd_update_values(T x, T g, T y, bool eq, T stp, T bb, T omb, T& yn, T& xn) {
    yn = (T)(x + (T)(g * stp));                    // agd.py:181
    if (!eq) yn = yn > (T)0 ? yn : (T)0;           // project_on_nn_cone, agd.py:13-21
    xn = (T)((T)(yn * omb) + (T)(y * bb));         // agd.py:184
}
template <class T>
__device__ __forceinline__ void agd_update_row(const ApplyArgs<T>& p, double step, int64_t i, T& yn, T& xn) {
    const float bt = p.beta[p.iter - 1];
    const T bb = (T)bt;
    const T omb = (T)(float)(1.0f - bt);  // 1.0 - fp32 0-dim tensor stays fp32 (agd.py:184)
    const bool eq = p.eq_mask && p.eq_mask[i];
    agd_update_values(p.x[i], p.g_new[i], p.y[i], eq, (T)step, bb, omb, yn, xn);
}

// What the optimiser state looks like to a step that is about to be applied (host side): iteration `iter` took its gradient at
// s->x; the stats launch left g in s->g_old and the scalars in scal[0..1].
struct PendingStep {
    bool valid = false;
    double gamma = 0.0;       // of the iteration being applied (log row)
    int64_t iter = 0;
    int decay_now = 0;
    double decay_factor = 1.0;
    const double* scal = nullptr;
};

template <class T>
inline ApplyArgs<T> make_apply_args(const dl_agd* s, const PendingStep& ps) {
    const int n_blocks = (int)((s->m + 63) / 64);  // kStatRows
    AgdDevState* states = (AgdDevState*)s->state;
    ApplyArgs<T> aa;
    aa.m = s->m;
    aa.partial_stats = s->partial_stats;
    aa.n_blocks = n_blocks;
    aa.g_new = (const T*)s->g_old;
    aa.scal = ps.scal;
    aa.st_in = states + s->state_cur;
    aa.st_out = states + (s->state_cur ^ 1);
    aa.log_row = (ps.iter >= 1 && ps.iter <= s->max_iter) ? s->log + (ps.iter - 1) * kLogCols : nullptr;
    aa.gamma = ps.gamma;
    aa.decay_now = ps.decay_now;
    aa.decay_factor = ps.decay_factor;
    aa.x = (const T*)s->x;
    aa.x_next = (T*)s->x_alt;
    aa.y = (const T*)s->y;
    aa.y_new = (T*)s->y_old;
    aa.eq_mask = s->eq_mask;
    aa.beta = s->beta;
    aa.iter = ps.iter;
    aa.x_perm = nullptr;
    aa.perm = nullptr;
    aa.chk = (s->chk_dead && n_blocks > 0) ? s->chk_partial : nullptr;
    aa.chk_dead = s->chk_dead;
    return aa;
}
// after a step has been applied (by the apply kernel or by a fused launch's prologue): the buffer that received y_i becomes y,
// the old y the "previous history dual"; likewise x, g, state
inline void agd_rotate(dl_agd* s) {
    void* t = s->y; s->y = s->y_old; s->y_old = t;
    t = s->g; s->g = s->g_old; s->g_old = t;
    t = s->x; s->x = s->x_alt; s->x_alt = t;
    s->state_cur ^= 1;
}

}  // namespace dl
