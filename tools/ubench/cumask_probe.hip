// cumask_probe.hip — where does a CU-masked stream place its blocks?  (measurement tool, not product code)
// hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip && ./cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <map>
#include <set>

__global__ void k_where(unsigned* out, int spin)
{
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // XCC_ID
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) {}
}

static void run(const char* name, const std::vector<uint32_t>& mask, int blocks)
{
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    unsigned* d; hipMalloc(&d, blocks * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 0, s, d, 20000);      // ~200 us at 100 MHz: blocks overlap
    hipStreamSynchronize(s);
    std::vector<unsigned> h(blocks * 2);
    hipMemcpy(h.data(), d, blocks * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> cus;        // xcc -> set of (se, cu) ids
    for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        cus[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    int total = 0;
    printf("%s:", name);
    for (auto& kv : cus) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  -> %d distinct CUs\n", total);
    hipFree(d); hipStreamDestroy(s);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    const int W = 8;
    std::vector<uint32_t> all(W, 0xffffffffu);
    run("all", all, 2048);
    { std::vector<uint32_t> m(W, 0); m[0] = 0xff; run("bits 0-7", m, 512); }
    { std::vector<uint32_t> m(W, 0); m[0] = 0xffffffffu; run("bits 0-31", m, 1024); }
    { std::vector<uint32_t> m(W, 0); for (int i = 0; i < 256; i += 8) m[i / 32] |= 1u << (i % 32); run("every 8th bit", m, 1024); }
    { std::vector<uint32_t> m(W, 0); for (int i = 0; i < 256; i += 32) m[i / 32] |= 1u << (i % 32); run("every 32nd bit", m, 512); }
    { std::vector<uint32_t> m(W, 0xffffffffu); for (int i = 0; i < 8; ++i) m[0] &= ~(1u << i); run("all but bits 0-7", m, 2048); }
    { std::vector<uint32_t> m(W, 0xffffffffu); for (int i = 0; i < 256; i += 32) m[i / 32] &= ~(1u << (i % 32)); run("all but every 32nd", m, 2048); }
    return 0;
}
