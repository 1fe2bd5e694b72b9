// pingpong.hip -- latency of a flag hand-off between two workgroups through global memory (agent-scope atomics + fences):
// what one hop of a device-side dependency costs on MI355X, same XCD and across XCDs.  (measurement tool, not product)
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pingpong tools/ubench/pingpong.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long wall() { return __builtin_readcyclecounter(); }

// block A (blockIdx 0) and block B (blockIdx == partner) bounce a counter n times; payload: 256 doubles written before the flag
__global__ __launch_bounds__(256) void k_pingpong(unsigned* flag, double* payload, int n, int partner, unsigned long long* out, int with_payload)
{
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == (unsigned)partner ? 1 : -1);
    if (me < 0) return;
    __shared__ double sink[256];
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < n; ++k) {
        const unsigned want = 2u * k + (me == 0 ? 0u : 1u);
        // wait for my turn
        if (threadIdx.x == 0) {
            unsigned long long tw = wall_clock64();
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want)
                if (wall_clock64() - tw > 100000000ull) { out[3] = 1; break; }      // 1 s
        }
        __syncthreads();
        if (with_payload) {
            __threadfence();
            sink[threadIdx.x] = payload[threadIdx.x] + 1.0;
            payload[threadIdx.x] = sink[threadIdx.x];
            __threadfence();
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag, want + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[me] = t1 - t0;
        if (me == 1) out[2] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // XCC_ID of the partner
        else out[4] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    }
}

int main() {
    unsigned* flag; double* payload; unsigned long long* out;
    CK(hipMalloc(&flag, 64)); CK(hipMalloc(&payload, 256 * 8)); CK(hipMalloc(&out, 64));
    const int n = 2000;
    for (int with_payload = 0; with_payload < 2; ++with_payload)
        for (int partner : {8, 1, 2, 4, 9, 16, 255}) {
            CK(hipMemset(flag, 0, 64)); CK(hipMemset(payload, 0, 256 * 8)); CK(hipMemset(out, 0, 64));
            hipLaunchKernelGGL(k_pingpong, dim3(256), dim3(256), 0, 0, flag, payload, n, partner, out, with_payload);
            CK(hipDeviceSynchronize());
            unsigned long long h[5]; CK(hipMemcpy(h, out, 40, hipMemcpyDeviceToHost));
            printf("payload %d  block 0 (xcc %llu) <-> block %3d (xcc %llu): %7.3f us per hop (%d round trips)%s\n", with_payload, h[4] & 15, partner,
                   h[2] & 15, h[0] * 0.01 / (2.0 * n), n, h[3] ? "  TIMEOUT" : "");
        }
    return 0;
}
