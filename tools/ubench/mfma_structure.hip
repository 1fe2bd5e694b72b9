// Standalone microbenchmark: what does the k_update block structure cost on the fp64 MFMA pipe?
// Variants: 0 pure MFMA (16 acc), 1 +barrier/64 MFMA, 2 +LDS fragment reads, 3 +LDS stage stores,
//           4 = 3 with 512-thread blocks (1 per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(256, 2) void k(double* out, int steps) {
    __shared__ double Ls[2][16][144];
    __shared__ double Ws[2][16][144];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4, wi = (wave & 1) * 64, wj = (wave >> 1) * 64;
    double4_t acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0, 0, 0, 0};
    for (int e = tid; e < 2 * 16 * 144; e += 256) { (&Ls[0][0][0])[e] = 1.0 + e * 1e-9; (&Ws[0][0][0])[e] = 1.0 - e * 1e-9; }
    __syncthreads();
    double a[4], b[4];
    for (int t = 0; t < 4; ++t) { a[t] = 1.0 + lane * 1e-9 + t; b[t] = 1.0 - lane * 1e-9 + t; }
    double2_t st = {1.0 + tid, 2.0 + tid};
    int cur = 0;
    for (int s = 0; s < steps; ++s) {
        #pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            if (V >= 2) {
                #pragma unroll
                for (int t = 0; t < 4; ++t) { a[t] = Ws[cur][kk + l4][wj + t * 16 + l15]; b[t] = Ls[cur][kk + l4][wi + t * 16 + l15]; }
            }
            if (V >= 3 && kk == 8) {
                #pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    *reinterpret_cast<double2_t*>(&Ls[cur ^ 1][(tid >> 6) + 4 * ps][(tid & 63) * 2]) = st;
                    *reinterpret_cast<double2_t*>(&Ws[cur ^ 1][(tid >> 6) + 4 * ps][(tid & 63) * 2]) = st;
                }
            }
            #pragma unroll
            for (int i = 0; i < 4; ++i)
                #pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (V >= 1) __syncthreads();
        cur ^= 1;
    }
    double sum = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 123.456) out[0] = sum;
}

// V5: fragments prefetched one sub-step ahead into an alternate register set; one DS op pinned
// behind every MFMA with sched_group_barrier; LDS stage writes spread over the second sub-step.
template <int V>
__global__ __launch_bounds__(256, 2) void k5(double* out, int steps) {
    __shared__ double Ls[2][16][144];
    __shared__ double Ws[2][16][144];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4, wi = (wave & 1) * 64, wj = (wave >> 1) * 64;
    double4_t acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0, 0, 0, 0};
    for (int e = tid; e < 2 * 16 * 144; e += 256) { (&Ls[0][0][0])[e] = 1.0 + e * 1e-9; (&Ws[0][0][0])[e] = 1.0 - e * 1e-9; }
    __syncthreads();
    double2_t st = {1.0 + tid, 2.0 + tid};
    int cur = 0;
    double a0[4], b0[4], a1[4], b1[4];
#define FR(buf, kk, a, b) { _Pragma("unroll") for (int t = 0; t < 4; ++t) { a[t] = Ws[buf][kk + l4][wj + t * 16 + l15]; b[t] = Ls[buf][kk + l4][wi + t * 16 + l15]; } }
#define MM(a, b) { _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0); }
#define ILV(n) { _Pragma("unroll") for (int q = 0; q < n; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(V == 6 ? 0x100 : 0x080, 1, 0); } }
    FR(cur, 0, a0, b0)
    for (int s = 0; s < steps; ++s) {
        FR(cur, 4, a1, b1)
        MM(a0, b0)
        ILV(8)
        FR(cur, 8, a0, b0)
        MM(a1, b1)
        ILV(8)
        FR(cur, 12, a1, b1)
        #pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            *reinterpret_cast<double2_t*>(&Ls[cur ^ 1][(tid >> 6) + 4 * ps][(tid & 63) * 2]) = st;
            *reinterpret_cast<double2_t*>(&Ws[cur ^ 1][(tid >> 6) + 4 * ps][(tid & 63) * 2]) = st;
        }
        MM(a0, b0)
        ILV(16)
        __syncthreads();
        FR(cur ^ 1, 0, a0, b0)
        MM(a1, b1)
        ILV(8)
        cur ^= 1;
    }
    double sum = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 123.456) out[0] = sum;
}
// V7: V5 + the global loads of the production kernel (two 16-byte loads per operand and stage, issued a full stage ahead of the
// LDS stores that consume them): what does the VMEM side of the staging cost the matrix pipe?
__global__ __launch_bounds__(256, 2) void k7(double* out, const double* __restrict__ Lg, const double* __restrict__ Wg, long ld, int steps) {
    __shared__ double Ls[2][16][144];
    __shared__ double Ws[2][16][144];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4, wi = (wave & 1) * 64, wj = (wave >> 1) * 64;
    double4_t acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0, 0, 0, 0};
    for (int e = tid; e < 2 * 16 * 144; e += 256) { (&Ls[0][0][0])[e] = 1.0 + e * 1e-9; (&Ws[0][0][0])[e] = 1.0 - e * 1e-9; }
    __syncthreads();
    const double* lsrc = Lg + (long)(blockIdx.x % 64) * 128 + (tid & 63) * 2 + (long)(tid >> 6) * ld;
    const double* wsrc = Wg + (long)((blockIdx.x / 64) % 64) * 128 + (tid & 63) * 2 + (long)(tid >> 6) * ld;
    double2_t lreg[4], wreg[4];
    #pragma unroll
    for (int ps = 0; ps < 4; ++ps) { lreg[ps] = *reinterpret_cast<const double2_t*>(lsrc + (long)(4 * ps) * ld); wreg[ps] = *reinterpret_cast<const double2_t*>(wsrc + (long)(4 * ps) * ld); }
    int cur = 0;
    double a0[4], b0[4], a1[4], b1[4];
#define ILV7(n, m) { _Pragma("unroll") for (int q = 0; q < n; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(m, 1, 0); } }
    FR(cur, 0, a0, b0)
    for (int s = 0; s < steps; ++s) {
        const long k2 = (long)((s + 1) & 63) * 16;
        FR(cur, 4, a1, b1)
        MM(a0, b0)
        ILV7(8, 0x100)
        FR(cur, 8, a0, b0)
        MM(a1, b1)
        ILV7(8, 0x100)
        FR(cur, 12, a1, b1)
        #pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            *reinterpret_cast<double2_t*>(&Ls[cur ^ 1][(tid >> 6) + 4 * ps][(tid & 63) * 2]) = lreg[ps];
            *reinterpret_cast<double2_t*>(&Ws[cur ^ 1][(tid >> 6) + 4 * ps][(tid & 63) * 2]) = wreg[ps];
        }
        #pragma unroll
        for (int ps = 0; ps < 4; ++ps) { lreg[ps] = *reinterpret_cast<const double2_t*>(lsrc + (k2 + 4 * ps) * ld); wreg[ps] = *reinterpret_cast<const double2_t*>(wsrc + (k2 + 4 * ps) * ld); }
        MM(a0, b0)
        ILV7(16, 0x0A0)
        __syncthreads();
        FR(cur ^ 1, 0, a0, b0)
        MM(a1, b1)
        ILV7(8, 0x100)
        cur ^= 1;
    }
    double sum = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 123.456) out[0] = sum;
}
double run7(int blocks, int steps) {
    double* d; hipMalloc(&d, 64);
    const long ld = 8192 + 0;
    double *Lg, *Wg; hipMalloc(&Lg, ld * 1040 * 8); hipMalloc(&Wg, ld * 1040 * 8);
    hipMemset(Lg, 0, ld * 1040 * 8); hipMemset(Wg, 0, ld * 1040 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k7, dim3(blocks), dim3(256), 0, 0, d, Lg, Wg, ld, 8);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k7, dim3(blocks), dim3(256), 0, 0, d, Lg, Wg, ld, steps);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d); hipFree(Lg); hipFree(Wg);
    return (double)blocks * 4 * steps * 64.0 * 2048.0 / (ms * 1e-3) / 1e12;
}
template <int V> double run5(int blocks, int steps) {
    double* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k5<V>, dim3(blocks), dim3(256), 0, 0, d, 8);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k5<V>, dim3(blocks), dim3(256), 0, 0, d, steps);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return (double)blocks * 4 * steps * 64.0 * 2048.0 / (ms * 1e-3) / 1e12;
}

template <int V> double run(int blocks, int steps) {
    double* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, 8);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, steps);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return (double)blocks * 4 * steps * 64.0 * 2048.0 / (ms * 1e-3) / 1e12;
}
int main(int argc, char** argv) {
    int steps = 4096;
    for (int blocks : {256, 512, 1024, 4096}) {
        printf("blocks=%5d  V0 pure=%.1f  V1 +barrier=%.1f  V2 +ldsread=%.1f  V3 +ldswrite=%.1f TF/s\n", blocks,
               run<0>(blocks, steps * 512 / blocks > 64 ? steps * 512 / blocks : 64), run<1>(blocks, steps * 512 / blocks),
               run<2>(blocks, steps * 512 / blocks), run<3>(blocks, steps * 512 / blocks));
    }
    printf("V5 (prefetch + 1 DS per MFMA, barrier mid-stage): blocks=512 %.1f  blocks=4096 %.1f TF/s\n", run5<5>(512, 4096), run5<5>(4096, 512));
    printf("V7 (V5 + global loads a stage ahead): blocks=512 %.1f  blocks=4096 steps 64 (one K=1024 tile each) %.1f  steps 512 %.1f TF/s\n", run7(512, 4096), run7(4096, 64), run7(4096, 512));
    printf("V5 again: blocks=4096 steps 64 %.1f TF/s\n", run5<5>(4096, 64));
    // short blocks like the real kernel: 16 steps per block, many blocks
    for (int steps2 : {16, 32, 64}) {
        int blocks = 32768;
        printf("blocks=%d steps/block=%d  V3=%.1f TF/s\n", blocks, steps2, run<3>(blocks, steps2));
    }
    return 0;
}
