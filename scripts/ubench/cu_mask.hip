// Can a stream keep a few CUs per XCD free for kernels that do not fit beside a resident sweep?  (Round 5, item 4: RCCL's
// rcclGenericKernel on gfx950 is 256 threads x 261-280 VGPRs + 19.7 KB LDS -- one wave per SIMD and only on a SIMD with no
// sweep wave -- so an exchange beside the window sweep waits for the sweep's END unless some CUs hold no sweep workgroup.)
//   1. hipExtStreamCreateWithCUMask with the first 8 n bits clear: which CUs run the masked stream's workgroups?  (KFD
//      deals the mask bits round-robin over the 8 XCCs: bit b -> XCC b % 8.)
//   2. while a sweep-shaped kernel (256 threads, <= 72 VGPRs, 21 KB LDS, 7 workgroups per CU of the masked set) spins for
//      `spin_ms`, a kernel with RCCL's footprint is launched on an unmasked stream: when does it START?
//   hipcc --offload-arch=gfx950 -O3 -o cu_mask cu_mask.hip && ./cu_mask [free CUs per XCD = 1] [spin ms = 20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) sweep_like(uint32_t *out, unsigned long long ticks)
{
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = threadIdx.x;
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    if (lds[(threadIdx.x + 1) & 255] == 0xFFFFFFFFu) out[0] = 0;
}

// RCCL's footprint: 256 threads, > 256 unified registers (v255 + a7 touched), 19744 bytes of LDS
__global__ void __launch_bounds__(256) rccl_like(unsigned long long *stamp, uint32_t *where)
{
    __shared__ uint32_t pad[19744 / 4];
    pad[threadIdx.x] = threadIdx.x;
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a7, 0" ::: "v255", "a7");
    if (threadIdx.x == 0) {
        stamp[blockIdx.x] = wall_clock64();
        where[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        where[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    }
    if (pad[(threadIdx.x + 1) & 255] == 0xFFFFFFFFu) stamp[0] = 0;
}

__global__ void stamp_now(unsigned long long *t) { *t = wall_clock64(); }

static uint32_t cu_key(uint32_t hw, uint32_t xcc)
{
    return ((xcc & 15u) << 12) | (((hw >> 13) & 7u) << 8) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
}

int main(int argc, char **argv)
{
    const int n_free = argc > 1 ? atoi(argv[1]) : 1;
    const double spin_ms = argc > 2 ? atof(argv[2]) : 20.0;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    int khz = 0;
    CHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0xFFFFFFFFu);
    for (int b = 0; b < 8 * n_free; b++) mask[(size_t)b / 32] &= ~(1u << (b % 32));
    hipStream_t masked, plain;
    hipError_t e = hipExtStreamCreateWithCUMask(&masked, (uint32_t)mask.size(), mask.data());
    printf("{\"cus\": %d, \"free_per_xcd\": %d, \"hipExtStreamCreateWithCUMask\": \"%s\"", n_cu, n_free, hipGetErrorString(e));
    if (e != hipSuccess) { printf("}\n"); return 0; }
    CHK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    const uint32_t grid = 7u * (uint32_t)(n_cu - 8 * n_free);
    uint32_t *d_out, *d_where;
    unsigned long long *d_stamp, *d_t0;
    CHK(hipMalloc(&d_out, 7 * 256 * 8));
    CHK(hipMalloc(&d_where, 64 * 8));
    CHK(hipMalloc(&d_stamp, 64 * 8));
    CHK(hipMalloc(&d_t0, 8));
    CHK(hipFuncSetAttribute((const void *)sweep_like, hipFuncAttributeMaxDynamicSharedMemorySize, 21504));
    const unsigned long long ticks = (unsigned long long)(spin_ms * 1e-3 * khz * 1e3);
    // warm both kernels (code object load)
    hipLaunchKernelGGL(sweep_like, dim3(8), dim3(256), 21504, masked, d_out, 1000ull);
    hipLaunchKernelGGL(rccl_like, dim3(1), dim3(256), 0, plain, d_stamp, d_where);
    CHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(sweep_like, dim3(grid), dim3(256), 21504, masked, d_out, ticks);
    hipLaunchKernelGGL(stamp_now, dim3(1), dim3(1), 0, masked, d_t0);       // (runs after the sweep-like kernel: its end)
    // give the dispatcher 2 ms to place the masked grid, then the RCCL-shaped kernel on the unmasked stream
    hipEvent_t ev;
    CHK(hipEventCreate(&ev));
    struct timespec ts = { 0, 2000000 };
    nanosleep(&ts, nullptr);
    unsigned long long *d_tl;
    CHK(hipMalloc(&d_tl, 8));
    hipLaunchKernelGGL(stamp_now, dim3(1), dim3(1), 0, plain, d_tl);
    hipLaunchKernelGGL(rccl_like, dim3(16), dim3(256), 0, plain, d_stamp, d_where);
    CHK(hipDeviceSynchronize());
    std::vector<uint32_t> h(grid * 2), w(32);
    std::vector<unsigned long long> st(16);
    unsigned long long t_end = 0, t_launch = 0;
    CHK(hipMemcpy(h.data(), d_out, grid * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(w.data(), d_where, 16 * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(st.data(), d_stamp, 16 * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(&t_end, d_t0, 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(&t_launch, d_tl, 8, hipMemcpyDeviceToHost));
    std::map<uint32_t, int> per_cu;
    std::map<uint32_t, std::set<uint32_t>> per_xcc;
    for (uint32_t b = 0; b < grid; b++) {
        per_cu[cu_key(h[2 * b], h[2 * b + 1])]++;
        per_xcc[h[2 * b + 1] & 15u].insert(cu_key(h[2 * b], h[2 * b + 1]));
    }
    std::map<int, int> hist;
    for (auto &kv : per_cu) hist[kv.second]++;
    printf(", \"masked_grid\": %u, \"distinct_cus_used\": %zu, \"cus_per_xcc\": [", grid, per_cu.size());
    bool first = true;
    for (auto &kv : per_xcc) { printf("%s%zu", first ? "" : ", ", kv.second.size()); first = false; }
    printf("], \"workgroups_per_cu_histogram\": {");
    first = true;
    for (auto &kv : hist) { printf("%s\"%d\": %d", first ? "" : ", ", kv.first, kv.second); first = false; }
    int on_free = 0;
    unsigned long long latest = 0;
    for (int b = 0; b < 16; b++) {
        if (!per_cu.count(cu_key(w[2 * b], w[2 * b + 1]))) on_free++;
        if (st[b] > latest) latest = st[b];
    }
    const double tick_us = 1e3 / khz;
    printf("}, \"rccl_like_workgroups_on_unused_cus\": %d, \"rccl_like_last_start_after_its_launch_us\": %.1f, "
           "\"rccl_like_last_start_before_sweep_end_us\": %.1f, \"spin_ms\": %.1f}\n", on_free, (double)(latest - t_launch) * tick_us,
           ((double)t_end - (double)latest) * tick_us, spin_ms);
    return 0;
}
