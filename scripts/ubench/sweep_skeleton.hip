// Where does the wave sweep stand between its two ceilings?  The sweep's skeleton -- out of place, nontemporal 16-byte row
// loads, rows staged in wave-private LDS, the 16-byte-per-lane byte gather, nontemporal stores, 3 rows per wave and trip --
// with a dial for the vector-ALU work per row: R rounds of Philox4x32 per lane and row (~7 VALU instructions each; the real
// kernel issues ~195 VALU wave-instructions per row, i.e. R ~ 28).  time(R) tells which ceiling binds at the operating point:
// flat in R = the memory skeleton, rising = the vector ALUs.
//   hipcc --offload-arch=gfx950 -O3 -o sweep_skeleton sweep_skeleton.hip && ./sweep_skeleton [blocks per CU = 7]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1)
{
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
}

template <int ROWS, int R, bool GATHER>
__global__ void __launch_bounds__(256, 8) skeleton(const uint8_t *x, uint8_t *y, const uint32_t *idx, uint32_t rows, uint32_t pitch, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[4][ROWS * 1024];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    uint32_t pidx[16];
#pragma unroll
    for (int k = 0; k < 16; k++) pidx[k] = idx[lane * 16 + k];
    for (uint32_t r0 = wave * ROWS; r0 < rows; r0 += nwaves * ROWS) {
        u32x4 v[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; k++)
            v[k] = __builtin_nontemporal_load((const u32x4 *)(x + (size_t)min(r0 + k, rows - 1u) * pitch + lane * 16u));
#pragma unroll
        for (int k = 0; k < ROWS; k++) *(u32x4 *)(buf[w] + k * 1024 + lane * 16u) = v[k];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            uint32_t o[4];
            if (GATHER) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t t = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) t |= (uint32_t)buf[w][k * 1024 + pidx[4 * j + b]] << (8 * b);
                    o[j] = t;
                }
            } else {
                const u32x4 t = *(const u32x4 *)(buf[w] + k * 1024 + lane * 16u);
                o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
            }
            // the dial: R Philox rounds keyed on (row, lane); the result touches one bit of the row so that nothing is dead
            uint32_t c0 = r0 + k, c1 = lane, c2 = seed, c3 = 1u, k0 = seed, k1 = 0x9E3779B9u;
#pragma unroll
            for (int r = 0; r < R; r++) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
            if (R) o[0] ^= (c0 ^ c1 ^ c2 ^ c3) & 1u;
            if (r0 + k < rows) {
                u32x4 s = { o[0], o[1], o[2], o[3] };
                __builtin_nontemporal_store(s, (u32x4 *)(y + (size_t)(r0 + k) * pitch + lane * 16u));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// the same with the next trip's row loads issued BEFORE this trip's gather and ALU work (register double buffer)
template <int ROWS, int R>
__global__ void __launch_bounds__(256, 8) skeleton_prefetch(const uint8_t *x, uint8_t *y, const uint32_t *idx, uint32_t rows, uint32_t pitch, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[4][ROWS * 1024];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    uint32_t pidx[16];
#pragma unroll
    for (int k = 0; k < 16; k++) pidx[k] = idx[lane * 16 + k];
    u32x4 v[ROWS], n[ROWS];
    uint32_t r0 = wave * ROWS;
    if (r0 < rows) {
#pragma unroll
        for (int k = 0; k < ROWS; k++) v[k] = __builtin_nontemporal_load((const u32x4 *)(x + (size_t)min(r0 + k, rows - 1u) * pitch + lane * 16u));
    }
    for (; r0 < rows; r0 += nwaves * ROWS) {
        const uint32_t rn = r0 + nwaves * ROWS;
#pragma unroll
        for (int k = 0; k < ROWS; k++) *(u32x4 *)(buf[w] + k * 1024 + lane * 16u) = v[k];
        if (rn < rows) {
#pragma unroll
            for (int k = 0; k < ROWS; k++) n[k] = __builtin_nontemporal_load((const u32x4 *)(x + (size_t)min(rn + k, rows - 1u) * pitch + lane * 16u));
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t t = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) t |= (uint32_t)buf[w][k * 1024 + pidx[4 * j + b]] << (8 * b);
                o[j] = t;
            }
            uint32_t c0 = r0 + k, c1 = lane, c2 = seed, c3 = 1u, k0 = seed, k1 = 0x9E3779B9u;
#pragma unroll
            for (int r = 0; r < R; r++) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
            if (R) o[0] ^= (c0 ^ c1 ^ c2 ^ c3) & 1u;
            if (r0 + k < rows) {
                u32x4 s = { o[0], o[1], o[2], o[3] };
                __builtin_nontemporal_store(s, (u32x4 *)(y + (size_t)(r0 + k) * pitch + lane * 16u));
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < ROWS; k++) v[k] = n[k];
    }
}

int main(int argc, char **argv)
{
    const int bpc = argc > 1 ? atoi(argv[1]) : 7;
    const uint32_t rows = 1200000, pitch = 1024;
    const size_t bytes = (size_t)rows * pitch;
    uint8_t *x, *y;
    uint32_t *idx;
    hipMalloc(&x, bytes);
    hipMalloc(&y, bytes);
    hipMemset(x, 1, bytes);
    hipMemset(y, 1, bytes);
    hipMalloc(&idx, 1024 * 4);
    uint32_t h[1024], s = 12345;
    for (int i = 0; i < 1024; i++) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) % 1000u; }
    hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 5; i++) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int n = 40;
        for (int i = 0; i < n; i++) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= n;
        printf("{\"variant\": \"%s\", \"blocks_per_cu\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n", name, bpc, ms, 2.0 * bytes / ms / 1e9);
    };
#define RUN(R_, G_) time((G_) ? "gather, R=" #R_ : "no gather, R=" #R_, [&] { hipLaunchKernelGGL((skeleton<3, R_, G_>), dim3(256 * bpc), dim3(256), 0, 0, x, y, idx, rows, pitch, 7u); })
    RUN(0, false);
    RUN(0, true);
    RUN(5, true);
    RUN(10, true);
    RUN(15, true);
    RUN(20, true);
    RUN(25, true);
    RUN(30, true);
    RUN(40, true);
#define RUNP(R_) time("gather + prefetch, R=" #R_, [&] { hipLaunchKernelGGL((skeleton_prefetch<3, R_>), dim3(256 * bpc), dim3(256), 0, 0, x, y, idx, rows, pitch, 7u); })
    RUNP(0);
    RUNP(10);
    RUNP(20);
    RUNP(25);
    return 0;
}
