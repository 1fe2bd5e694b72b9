// contention.hip -- what exactly slows a latency-bound kernel down beside the bulk update?  (measurement tool, not product)
// The production bulk kernel (k_update<128,true,8> / <256,true,8>, included from the library's sources) runs in a loop on
// one stream; single-block probe kernels run on a high-priority stream, alone and beside it:
//   chase  : dependent global loads (L2-missing pointer chase)         -> memory latency under load
//   fma    : dependent DP FMA chains on 4 waves, s_setprio 3            -> VALU issue beside MFMA-saturating waves
//   mfma   : dependent fp64 MFMA chain on 4 waves                       -> matrix pipe sharing
//   lds    : dependent ds_write/ds_read round trips                     -> LDS pipe sharing
//   icache : ~96 KB of straight-line code executed once                 -> instruction fetch under load
//   empty  : an empty kernel                                            -> dispatch latency
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pyipm_amd/csrc -I include -o tools/ubench/contention tools/ubench/contention.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <numeric>
#include "kernels_factor.hpp"

using namespace pyipm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(64) void p_chase(const unsigned* __restrict__ next, int n, unsigned long long* out) {
    unsigned i = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < n; ++k) i = next[i];
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ __launch_bounds__(256) void p_fma(int n, double seed, unsigned long long* out, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    double x = seed + threadIdx.x * 1e-9, y = 1.0000001;
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < n; ++k) { x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); x = fma(x, y, 1e-9); }
    const unsigned long long t1 = wall_clock64();
    if (x == 123.456) out[2] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ __launch_bounds__(256) void p_mfma(int n, unsigned long long* out, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < n; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    const unsigned long long t1 = wall_clock64();
    if (acc[0] + acc[1] == 123.456) out[2] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ __launch_bounds__(256) void p_lds(int n, unsigned long long* out, int prio) {
    __shared__ double buf[256 + 8];
    if (prio) __builtin_amdgcn_s_setprio(3);
    double x = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < n; ++k) {
        buf[threadIdx.x] = x;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        x = buf[(threadIdx.x & 192) | ((threadIdx.x + 1) & 63)] + 1.0;      // another lane of the same wave: no barrier needed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = wall_clock64();
    if (x == 123.456) out[2] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
// ~96 KB of straight-line VALU code (12288 instructions of 8 bytes), executed once
__global__ __launch_bounds__(64) void p_icache(double seed, unsigned long long* out) {
    double x = seed + threadIdx.x;
    const unsigned long long t0 = wall_clock64();
#define F8 x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9); x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9); \
           x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9); x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9);
#define F64 F8 F8 F8 F8 F8 F8 F8 F8
#define F512 F64 F64 F64 F64 F64 F64 F64 F64
#define F4096 F512 F512 F512 F512 F512 F512 F512 F512
    F4096 F4096 F4096
    const unsigned long long t1 = wall_clock64();
    if (x == 123.456) out[2] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void p_empty() {}

struct Bulk {
    double *C, *L, *W; int64_t m, ld; int K; UpdGeo u; unsigned grid; int bn;
    void init(int64_t m_, int K_, int bn_) {
        m = m_; K = K_; ld = m; bn = bn_;
        CK(hipMalloc(&C, (size_t)m * m * 8)); CK(hipMalloc(&L, (size_t)m * K * 8)); CK(hipMalloc(&W, (size_t)m * K * 8));
        CK(hipMemset(C, 0, (size_t)m * m * 8)); CK(hipMemset(L, 0, (size_t)m * K * 8)); CK(hipMemset(W, 0, (size_t)m * K * 8));
        u.row_begin = 0; u.Npad = m; u.first_lp = 0; u.sub0 = 0; u.nb = 256; u.world = 1; u.rank = 0;
        u.nrt = (int)(m / BM); u.nct = (int)(m / bn); u.prio = 0; u.a0 = 0; u.a1 = m; u.b0 = 0; u.b1 = 0; u.dbg = nullptr; u.tiles = nullptr;
        u.ks_cstride = 0;
        if (bn == 256) { upd_fill_affine<256>(u); grid = (unsigned)(upd_super_count<256>(u) * 64); }
        else { upd_fill_affine<128>(u); grid = (unsigned)(upd_super_count<128>(u) * 64); }
    }
    void launch(hipStream_t s) {
        if (bn == 256) hipLaunchKernelGGL((k_update<256, true, 8>), dim3(grid), dim3(512), 0, s, C, ld, L, ld, W, ld, K, u);
        else hipLaunchKernelGGL((k_update<128, true, 8>), dim3(grid), dim3(512), 0, s, C, ld, L, ld, W, ld, K, u);
    }
    double flops() const { return 2.0 * K * ((double)m * (m + 1) / 2); }
};

int main(int argc, char** argv) {
    const int bn = argc > 1 ? atoi(argv[1]) : 128;
    const int64_t m = argc > 2 ? atoll(argv[2]) : 16384;
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t sb, sp; CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); CK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi));
    Bulk B; B.init(m, 1024, bn);
    // pointer chase over 256 MB: a random cycle
    const size_t nn = 64u << 20;
    std::vector<unsigned> perm(nn); std::iota(perm.begin(), perm.end(), 0u);
    srand(1); for (size_t i = nn - 1; i > 0; --i) { size_t j = ((size_t)rand() * RAND_MAX + rand()) % (i + 1); std::swap(perm[i], perm[j]); }
    std::vector<unsigned> nxt(nn); for (size_t i = 0; i < nn; ++i) nxt[perm[i]] = perm[(i + 1) % nn];
    unsigned* dn; CK(hipMalloc(&dn, nn * 4)); CK(hipMemcpy(dn, nxt.data(), nn * 4, hipMemcpyHostToDevice));
    unsigned long long* out; CK(hipMalloc(&out, 64)); CK(hipMemset(out, 0, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto probe = [&](const char* name, int which, double per) {
        float ms = 0.f; unsigned long long h[3];
        CK(hipEventRecord(e0, sp));
        switch (which) {
            case 0: hipLaunchKernelGGL(p_chase, dim3(1), dim3(64), 0, sp, dn, 2000, out); break;
            case 1: hipLaunchKernelGGL(p_fma, dim3(1), dim3(256), 0, sp, 5000, 1.0, out, 1); break;
            case 2: hipLaunchKernelGGL(p_mfma, dim3(1), dim3(256), 0, sp, 2000, out, 1); break;
            case 3: hipLaunchKernelGGL(p_lds, dim3(1), dim3(256), 0, sp, 2000, out, 1); break;
            case 4: hipLaunchKernelGGL(p_icache, dim3(1), dim3(64), 0, sp, 1.0, out); break;
            case 5: hipLaunchKernelGGL(p_empty, dim3(1), dim3(64), 0, sp); break;
            case 6: hipLaunchKernelGGL(p_fma, dim3(1), dim3(256), 0, sp, 5000, 1.0, out, 0); break;
        }
        CK(hipEventRecord(e1, sp)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
        printf("  %-22s kernel %8.1f us (events)   in-kernel %8.1f us = %7.1f ns per %s\n", name, ms * 1e3, h[0] * 0.01, h[0] * 10.0 / per,
               which == 0 ? "load" : which == 4 ? "instruction" : "step");
    };
    auto all = [&]() {
        probe("chase (2000 loads)", 0, 2000); probe("fma prio3 (20000)", 1, 20000); probe("fma prio0 (20000)", 6, 20000);
        probe("mfma chain (2000)", 2, 2000); probe("lds round trips (2000)", 3, 2000); probe("icache (12288 instr)", 4, 12288); probe("empty", 5, 1);
    };
    for (int w = 0; w < 2; ++w) { B.launch(sb); } CK(hipDeviceSynchronize());
    float ms; CK(hipEventRecord(e0, sb)); B.launch(sb); B.launch(sb); CK(hipEventRecord(e1, sb)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("bulk k_update<%d,true,8> m=%lld K=1024 alone: %.2f ms per launch = %.1f TF/s\n", bn, (long long)m, ms / 2, 2 * B.flops() / ms / 1e9);
    printf("probes alone:\n"); all(); all();
    printf("probes beside the bulk kernel:\n");
    hipEvent_t b0, b1; CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    CK(hipEventRecord(b0, sb)); for (int k = 0; k < 40; ++k) B.launch(sb); CK(hipEventRecord(b1, sb));
    all(); all();
    CK(hipEventSynchronize(b1)); CK(hipEventElapsedTime(&ms, b0, b1));
    printf("bulk beside the probes: %.1f TF/s\n", 40 * B.flops() / ms / 1e9);
    return 0;
}
