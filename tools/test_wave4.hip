// tools/test_wave4.hip -- developer test: 4-per-lane segmented all-reduce and simplex_tile4 against CPU references.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I dualip_amd/csrc tools/test_wave4.hip -o /tmp/test_wave4
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "simplex4.h"
using namespace dl;

__global__ void k(const float* v, const uint64_t* heads, const int* range, float z, float* x, float* sums, float* maxs) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x;
    uint64_t H[4], live[4];
    for (int j = 0; j < 4; ++j) {
        H[j] = heads[t * 4 + j];
        H[j] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(H[j] >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)H[j]);
    }
    const int lo = __builtin_amdgcn_readfirstlane(range[2 * t]), hi = __builtin_amdgcn_readfirstlane(range[2 * t + 1]);
    for (int j = 0; j < 4; ++j) {
        uint64_t m = 0;
        for (int L = 0; L < 64; ++L)
            if (4 * L + j >= lo && 4 * L + j < hi) m |= 1ull << L;
        live[j] = m;
    }
    const Seg4 s = make_seg4(H);
    const LaneConst lc = make_lane_const(lane);
    float vv[4], xx[4], tot[4], mx[4], uu[4];
    for (int j = 0; j < 4; ++j) {
        vv[j] = v[t * 256 + 4 * lane + j];
        xx[j] = -7.f;
        uu[j] = lane_bit(live[j]) ? fabsf(vv[j]) : 0.f;
    }
    const int el = end_lane4(s, lc);
    seg_allreduce4(uu, s, el, OpAdd(), tot);
    seg_allreduce4(uu, s, el, OpMaxNonNeg(), mx);
    const ProjT<float> pj = make_proj<float>(DL_PROJ_SIMPLEX, z, 0.0);
    float vz[4];
    for (int j = 0; j < 4; ++j) vz[j] = lane_bit(live[j]) ? vv[j] : 0.f;
    simplex_tile4(vz, s, pj, lc, xx);
    for (int j = 0; j < 4; ++j) {
        x[t * 256 + 4 * lane + j] = xx[j];
        sums[t * 256 + 4 * lane + j] = tot[j];
        maxs[t * 256 + 4 * lane + j] = mx[j];
    }
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
    const int NT = 8192;
    std::mt19937 rng(3);
    std::vector<float> v(NT * 256), want(NT * 256, -7.f), wsum(NT * 256, 0.f), wmax(NT * 256, 0.f);
    std::vector<uint64_t> heads(NT * 4, 0);
    std::vector<int> range(NT * 2);
    for (int t = 0; t < NT; ++t) {
        const int regime = t % 3;
        const int lo = rng() % 4;
        int pos = lo;
        auto set_head = [&](int e) { heads[t * 4 + (e & 3)] |= 1ull << (e >> 2); };
        set_head(0);
        for (int i = 0; i < 256; ++i) v[t * 256 + i] = 99.f + i;  // garbage outside the valid range
        while (true) {
            int len = 1 + rng() % (t % 7 == 0 ? 120 : (t % 5 == 0 ? 40 : 14));
            if (pos + len > 256) break;
            set_head(pos);
            std::vector<float> col(len);
            float sm = 0, mx = -1e30f;
            for (auto& e : col) {
                float r = (float)(rng() % 100000) / 100000.f;
                e = regime == 0 ? r * 30.f - 2.f : (regime == 1 ? r * 0.5f - 0.1f : r * 1.5f - 0.3f);
            }
            for (int i = 0; i < len; ++i) { v[t * 256 + pos + i] = col[i]; sm += std::fabs(col[i]); mx = std::max(mx, std::fabs(col[i])); }
            for (int i = 0; i < len; ++i) { wsum[t * 256 + pos + i] = sm; wmax[t * 256 + pos + i] = mx; }
            ref_proj(col, 1.0f);
            for (int i = 0; i < len; ++i) want[t * 256 + pos + i] = col[i];
            pos += len;
            if (t % 11 == 0 && pos > 100) break;  // short tiles
        }
        if (pos < 256) set_head(pos);
        range[2 * t] = lo;
        range[2 * t + 1] = pos;
    }
    float *dv, *dx, *ds, *dm;
    uint64_t* dh;
    int* dr;
    hipMalloc(&dv, v.size() * 4); hipMalloc(&dx, v.size() * 4); hipMalloc(&ds, v.size() * 4); hipMalloc(&dm, v.size() * 4);
    hipMalloc(&dh, heads.size() * 8); hipMalloc(&dr, range.size() * 4);
    hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dh, heads.data(), heads.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dr, range.data(), range.size() * 4, hipMemcpyHostToDevice);
    k<<<NT, 64>>>(dv, dh, dr, 1.0f, dx, ds, dm);
    std::vector<float> x(v.size()), sm(v.size()), mx(v.size());
    hipMemcpy(x.data(), dx, x.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(sm.data(), ds, x.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(mx.data(), dm, x.size() * 4, hipMemcpyDeviceToHost);
    printf("hip error: %s\n", hipGetErrorString(hipGetLastError()));
    int nbs = 0, nbm = 0, nbx = 0, first = -1;
    double worst = 0;
    for (int t = 0; t < NT; ++t)
        for (int i = range[2 * t]; i < range[2 * t + 1]; ++i) {
            const size_t e = (size_t)t * 256 + i;
            if (std::fabs(sm[e] - wsum[e]) > 1e-3 * (1 + std::fabs(wsum[e]))) { nbs++; if (first < 0) first = (int)e; }
            if (mx[e] != wmax[e]) { nbm++; if (first < 0) first = (int)e; }
            const double err = std::fabs(x[e] - want[e]);
            if (err > 1e-4) { nbx++; if (first < 0) first = (int)e; }
            worst = std::max(worst, err);
        }
    printf("bad sums %d  bad max %d  bad x %d  worst |dx| %.3g\n", nbs, nbm, nbx, worst);
    if (first >= 0) {
        int t = first / 256;
        printf("first bad: tile %d elem %d range [%d,%d) heads %016llx %016llx %016llx %016llx\n", t, first % 256, range[2 * t], range[2 * t + 1],
               (unsigned long long)heads[4 * t], (unsigned long long)heads[4 * t + 1], (unsigned long long)heads[4 * t + 2], (unsigned long long)heads[4 * t + 3]);
        for (int i = std::max(0, first % 256 - 12); i < std::min(256, first % 256 + 12); ++i)
            printf("  e%3d v %.4f sum %.4f/%.4f max %.4f/%.4f x %.4f/%.4f\n", i, v[t * 256 + i], sm[t * 256 + i], wsum[t * 256 + i], mx[t * 256 + i], wmax[t * 256 + i], x[t * 256 + i], want[t * 256 + i]);
    }
    return 0;
}
