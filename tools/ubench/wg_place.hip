// wg_place.hip -- where do the workgroups of a launch whose blocks each take a whole CU (144 KB of shared memory) land, block by
// block?  (measurement tool, not product code: is there a block index whose CU shares its instruction cache -- the CU pair --
// with another block of the SAME launch?  k_tile_chain, kernels_chain.hpp)
// hipcc --offload-arch=gfx950 -O2 -o wg_place wg_place.hip && ./wg_place [blocks]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void k_where(unsigned* out, int spin)
{
    extern __shared__ char pad[];
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // XCC_ID
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) {}
}

int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 100;
    hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    unsigned* d; hipMalloc(&d, blocks * 2 * sizeof(unsigned));
    for (int rep = 0; rep < 3; ++rep) {
        hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 144 * 1024, s, d, 30000);
        hipStreamSynchronize(s);
        std::vector<unsigned> h(blocks * 2);
        hipMemcpy(h.data(), d, blocks * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
        printf("launch %d: block -> xcc.se.cu\n", rep);
        for (int b = 0; b < blocks; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            printf(" %d:%u.%u.%u", b, xcc, (hw >> 13) & 0x7, (hw >> 8) & 0xf);
            if (b % 8 == 7) printf("\n");
        }
        printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
