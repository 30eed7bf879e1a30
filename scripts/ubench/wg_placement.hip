// Where does the dispatcher put the workgroups of a launch that fills 7/8 of the chip's LDS?  grid = 7 x 256 workgroups of
// 256 threads with 20 KB of LDS each (the wave sweep's shape inside the generation loop): evenly, 7 per CU -- every CU keeps
// 4 wave slots and 20 KB for the accessory chain -- or packed, 8 per CU on 224 CUs with 32 CUs left empty?
//   hipcc --offload-arch=gfx950 -O3 -o wg_placement wg_placement.hip && ./wg_placement [grid]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) probe(uint32_t *out, uint32_t spin)
{
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = threadIdx.x;
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
    const uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    if (lds[(threadIdx.x + 1) & 255] == 0xFFFFFFFFu) out[0] = 0;
}
int main(int argc, char **argv)
{
    const uint32_t grid = argc > 1 ? atoi(argv[1]) : 7 * 256;
    uint32_t *d;
    hipMalloc(&d, grid * 8);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 20480);
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 20480, 0, d, 2000000u);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(grid * 2);
    hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> per_cu;
    for (uint32_t b = 0; b < grid; b++) {
        const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 15u;
        const uint32_t cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    std::map<int, int> hist;
    for (auto &kv : per_cu) hist[kv.second]++;
    printf("grid %u: %zu distinct CUs hold workgroups;", grid, per_cu.size());
    for (auto &kv : hist) printf("  %d CUs x %d workgroups", kv.second, kv.first);
    printf("\n");
    return 0;
}
