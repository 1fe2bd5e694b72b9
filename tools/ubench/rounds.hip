// rounds.hip -- how much does a ragged last round cost the bulk update?  k_update<128,true,8> (K = 1024) alone on m x m lower
// triangles with T = r (r + 1) / 2 tiles for consecutive r: time against T / 512 slots.  (measurement tool, not product)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pyipm_amd/csrc -I include -o tools/ubench/rounds tools/ubench/rounds.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "kernels_factor.hpp"
using namespace pyipm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int r0 = argc > 1 ? atoi(argv[1]) : 40, r1 = argc > 2 ? atoi(argv[2]) : 56;
    const int bn = argc > 3 ? atoi(argv[3]) : 128;
    const int alias = argc > 4 ? atoi(argv[4]) : 0;           // 1: every column of C is the same memory (ldc = 0): no HBM traffic for the C tiles            // 128 x bn tiles; slots: 512 (two blocks per CU) or 256 (one)
    const int64_t mmax = (int64_t)r1 * 128; const int K = 1024;
    double *C, *L, *W;
    CK(hipMalloc(&C, (size_t)mmax * mmax * 8)); CK(hipMalloc(&L, (size_t)mmax * K * 8)); CK(hipMalloc(&W, (size_t)mmax * K * 8));
    CK(hipMemset(C, 0, (size_t)mmax * mmax * 8)); CK(hipMemset(L, 0, (size_t)mmax * K * 8)); CK(hipMemset(W, 0, (size_t)mmax * K * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = r0; r <= r1; ++r) {
        const int64_t m = (int64_t)r * 128;
        UpdGeo u; u.row_begin = 0; u.Npad = m; u.first_lp = 0; u.sub0 = 0; u.nb = 256; u.world = 1; u.rank = 0;
        u.nrt = r; u.nct = bn == 256 ? (r + 1) / 2 : r; u.prio = 0; u.a0 = 0; u.a1 = m; u.b0 = 0; u.b1 = 0; u.dbg = nullptr; u.tiles = nullptr; u.ks_cstride = 0;
        // compact list of the lower-triangle tiles in the XCD-aware order (what the library builds on the host)
        if (bn == 256) upd_fill_affine<256>(u); else upd_fill_affine<128>(u);
        const int64_t nsup = bn == 256 ? upd_super_count<256>(u) : upd_super_count<128>(u);
        std::vector<unsigned> seq[8];
        const int nsr = (u.nrt + 7) >> 3, nsc = (u.nct + 7) >> 3;
        for (int64_t b = 0; b < ((nsup + 7) / 8) * 8 * 64; ++b) {
            const int xcd = (int)(b & 7); const int64_t slot = b >> 3; int sidx = (int)((slot >> 6) * 8) + xcd; const int within = (int)(slot & 63);
            int sJ = 0, sI = -1;
            for (; sJ < nsc; ++sJ) { const int mn = upd_super_min_row(u, sJ); const int cnt = mn < nsr ? nsr - mn : 0; if (sidx < cnt) { sI = mn + sidx; break; } sidx -= cnt; }
            if (sI < 0) continue;
            const int64_t rt = (int64_t)sI * 8 + (within & 7), ct = (int64_t)sJ * 8 + (within >> 3);
            if (rt >= u.nrt || ct >= u.nct || (rt + 1) * 128 <= ct * bn) continue;      // strictly above the diagonal
            seq[xcd].push_back((unsigned)rt | ((unsigned)ct << 16));
        }
        size_t total = 0; for (auto& v : seq) total += v.size();
        const size_t target = (total + 7) / 8;
        std::vector<unsigned> spare;
        for (auto& v : seq) while (v.size() > target) { spare.push_back(v.back()); v.pop_back(); }
        for (auto& v : seq) while (v.size() < target && !spare.empty()) { v.push_back(spare.back()); spare.pop_back(); }
        std::vector<unsigned> list(8 * target, 0xffffffffu);
        for (int x = 0; x < 8; ++x) for (size_t j = 0; j < seq[x].size(); ++j) list[8 * j + x] = seq[x][j];
        unsigned* dl; CK(hipMalloc(&dl, list.size() * 4)); CK(hipMemcpy(dl, list.data(), list.size() * 4, hipMemcpyHostToDevice));
        u.tiles = dl;
        auto launch = [&]() {
            if (bn == 256) hipLaunchKernelGGL((k_update<256, true, 8>), dim3((unsigned)list.size()), dim3(512), 0, 0, C, alias ? 0 : m, L, m, W, m, K, u);
            else hipLaunchKernelGGL((k_update<128, true, 8>), dim3((unsigned)list.size()), dim3(512), 0, 0, C, alias ? 0 : m, L, m, W, m, K, u); };
        launch(); launch(); CK(hipDeviceSynchronize());
        std::vector<float> ts;
        for (int rep = 0; rep < 7; ++rep) { float ms; CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms); }
        std::sort(ts.begin(), ts.end());
        const double T = (double)total;
        const double slots = bn == 256 ? 256.0 : 512.0;
        // flops: entries on / below the diagonal of the covered tiles (a 128 x 256 tile on the diagonal is counted whole)
        printf("r=%3d tiles %5.0f = %6.3f x %3.0f  median %8.1f us  %6.1f us per full-slot round  %5.1f TF/s (tile flops)\n", r, T, T / slots, slots,
               ts[3] * 1e3, ts[3] * 1e3 / (T / slots), 2.0 * K * (T * 128.0 * bn) / ts[3] / 1e9);
        CK(hipFree(dl));
    }
    return 0;
}
