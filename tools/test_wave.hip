// tools/test_wave.hip -- developer test: simplex_batch (segmented scans + Newton) against a sort-based CPU projection
// on random tiles.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I dualip_amd/csrc tools/test_wave.hip -o /tmp/test_wave
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "simplex.h"
using namespace dl;

template <bool DPP>
__global__ void k(const float* v, const uint64_t* heads, const int* counts, float z, float* x, int nb) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x;
    float vv[kBatch], xx[kBatch];
    bool valid[kBatch];
    uint64_t hd[kBatch];
    ProjT<float> pj[kBatch];
    bool smp[kBatch];
    const LaneConst lc = make_lane_const(lane);
    for (int q = 0; q < kBatch; ++q) {
        smp[q] = true;
        const int t = b * kBatch + q;
        vv[q] = v[t * 64 + lane];
        hd[q] = heads[t];
        valid[q] = lane < counts[t];
        pj[q] = make_proj<float>(DL_PROJ_SIMPLEX, z, 0.0);
        xx[q] = vv[q];
    }
    const int32_t* eq_row[kBatch] = {};
    simplex_batch<DPP>(vv, valid, hd, pj, smp, lc, xx, eq_row);
    for (int q = 0; q < kBatch; ++q) x[(b * kBatch + q) * 64 + lane] = valid[q] ? xx[q] : 0.f;
}

static void ref_proj(std::vector<float>& u, float z) {
    float S = 0;
    for (auto& e : u) { e = std::max(e, 0.f); S += e; }
    if (S <= z + 1e-6f) return;
    std::vector<float> s = u;
    std::sort(s.begin(), s.end(), std::greater<float>());
    double cum = 0, th = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        cum += s[i];
        if (s[i] - (cum - z) / (i + 1) > 0) th = (cum - z) / (i + 1);
    }
    for (auto& e : u) e = std::max((float)(e - th), 0.f);
}

int main() {
    const int NT = 4096 * kBatch;
    std::mt19937 rng(1);
    std::vector<float> v(NT * 64), want(NT * 64, 0.f);
    std::vector<uint64_t> heads(NT);
    std::vector<int> counts(NT);
    for (int t = 0; t < NT; ++t) {
        int regime = t % 3;
        int pos = 0;
        uint64_t h = 0;
        while (true) {
            int len = 1 + rng() % (t % 5 == 0 ? 40 : 14);
            if (pos + len > 64) break;
            h |= 1ull << pos;
            std::vector<float> col(len);
            for (auto& e : col) {
                float r = (float)(rng() % 100000) / 100000.f;
                e = regime == 0 ? r * 30.f - 2.f : (regime == 1 ? r * 0.5f - 0.1f : r * 1.5f - 0.3f);
            }
            for (int i = 0; i < len; ++i) v[t * 64 + pos + i] = col[i];
            ref_proj(col, 1.0f);
            for (int i = 0; i < len; ++i) want[t * 64 + pos + i] = col[i];
            pos += len;
        }
        if (pos < 64) h |= 1ull << pos;
        for (int i = pos; i < 64; ++i) v[t * 64 + i] = 123.f;
        heads[t] = h;
        counts[t] = pos;
    }
    float *dv, *dx;
    uint64_t* dh;
    int* dc;
    hipMalloc(&dv, v.size() * 4); hipMalloc(&dx, v.size() * 4); hipMalloc(&dh, NT * 8); hipMalloc(&dc, NT * 4);
    hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dh, heads.data(), NT * 8, hipMemcpyHostToDevice);
    hipMemcpy(dc, counts.data(), NT * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) k<true><<<NT / kBatch, 64>>>(dv, dh, dc, 1.0f, dx, NT);
        else k<false><<<NT / kBatch, 64>>>(dv, dh, dc, 1.0f, dx, NT);
        std::vector<float> x(v.size());
        hipMemcpy(x.data(), dx, x.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0; int nbad = 0, first = -1;
        for (size_t i = 0; i < x.size(); ++i) {
            double e = std::fabs(x[i] - want[i]);
            if (e > 1e-4) { nbad++; if (first < 0) first = (int)i; }
            worst = std::max(worst, e);
        }
        printf("mode %s: worst %.3g bad %d first tile %d lane %d\n", mode == 0 ? "dpp" : "bpermute", worst, nbad, first / 64, first % 64);
        if (first >= 0) {
            int t = first / 64;
            printf("  head %016llx count %d\n  got : ", (unsigned long long)heads[t], counts[t]);
            for (int i = 0; i < 64; ++i) printf("%.3f ", x[t * 64 + i]);
            printf("\n  want: ");
            for (int i = 0; i < 64; ++i) printf("%.3f ", want[t * 64 + i]);
            printf("\n  v   : ");
            for (int i = 0; i < 64; ++i) printf("%.3f ", v[t * 64 + i]);
            printf("\n");
        }
    }
    return 0;
}
