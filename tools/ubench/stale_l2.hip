// Does a kernel that STARTS after a written-through (sc1) store of a still-running kernel see that store with PLAIN loads, when the
// reading XCD's L2 holds an older copy of the line?  (round 6: k_tile_chain in one launch + the rows kernels behind k_chain_wait)
// build: hipcc --offload-arch=gfx950 -O2 -o stale_l2 stale_l2.hip ; prints the number of stale words per trial
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void k_touch(const double* buf, size_t n, double* sink) {          // every workgroup reads the whole buffer: every XCD's L2 holds it
    double s = 0; for (size_t i = threadIdx.x; i < n; i += blockDim.x) s += buf[i];
    if (s == 12345.678) sink[0] = s;
}
__global__ void k_fill(double* buf, size_t n, double v) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = v; }
__global__ void k_producer(double* buf, size_t n, double v, unsigned* flag, int sc1, unsigned long long spin) {
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
        if (sc1) __hip_atomic_store(buf + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else buf[i] = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // keep running: the consumer kernel starts while this one has not ended (no end-of-kernel release yet)
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(32);
}
__global__ void k_wait(const unsigned* flag) {
    if (threadIdx.x == 0) while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_consumer(const double* buf, size_t n, double want, unsigned* stale, int sc1) {   // every workgroup (all XCDs) checks the whole buffer
    unsigned bad = 0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = sc1 ? __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : buf[i];
        bad += (x != want);
    }
    if (bad) atomicAdd(stale + (blockIdx.x & 7), bad);
}
int main() {
    const size_t n = 64 * 1024;                 // 512 KB
    double *buf, *sink; unsigned *flag, *stale;
    hipMalloc(&buf, n * 8); hipMalloc(&sink, 8); hipMalloc(&flag, 4); hipMalloc(&stale, 32);
    hipStream_t sp, sc; hipStreamCreate(&sp); hipStreamCreate(&sc);
    for (int mode = 0; mode < 4; ++mode) {       // producer sc1? consumer sc1?
        const int psc1 = mode & 1, csc1 = mode >> 1;
        unsigned tot[8] = {0};
        for (int trial = 0; trial < 20; ++trial) {
            const double oldv = 1000.0 + trial, newv = 2000.0 + trial;
            hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, buf, n, oldv);
            hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, 0, buf, n, sink);     // old values into every XCD's L2
            hipMemsetAsync(flag, 0, 4, 0); hipMemsetAsync(stale, 0, 32, 0);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(k_producer, dim3(1), dim3(256), 0, sp, buf, n, newv, flag, psc1, 200000ull);   // 2 ms of spinning after the flag
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, sc, flag);
            hipLaunchKernelGGL(k_consumer, dim3(64), dim3(256), 0, sc, buf, n, newv, stale, csc1);
            hipDeviceSynchronize();
            unsigned h[8]; hipMemcpy(h, stale, 32, hipMemcpyDeviceToHost);
            for (int k = 0; k < 8; ++k) tot[k] += h[k];
        }
        printf("producer %s stores, consumer kernel (started behind a wait kernel, producer still running) %s loads: stale words by XCD:",
               psc1 ? "sc1" : "plain", csc1 ? "sc1" : "plain");
        for (int k = 0; k < 8; ++k) printf(" %u", tot[k]);
        printf("\n");
    }
    return 0;
}
