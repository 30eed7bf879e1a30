// Micro-benchmark: practical ceiling of an in-place read-modify-write stream over the core matrix
// (1.2 M rows x 1024 B), in the access shapes the sweep uses.
// hipcc --offload-arch=gfx950 -O3 -o inplace_stream inplace_stream.hip && ./inplace_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// (0) flat grid-stride: every thread 16 B
__global__ void __launch_bounds__(256) flat(uint4 *x, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = x[i];
        v.x ^= 1u;
        x[i] = v;
    }
}

// (0b) out of place: read x, write y
template <bool NT>
__global__ void __launch_bounds__(256) flat_copy(const uint4 *x, uint4 *y, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v;
        if (NT) {
            v.x = __builtin_nontemporal_load(&x[i].x); v.y = __builtin_nontemporal_load(&x[i].y);
            v.z = __builtin_nontemporal_load(&x[i].z); v.w = __builtin_nontemporal_load(&x[i].w);
        } else v = x[i];
        v.x ^= 1u;
        if (NT) {
            __builtin_nontemporal_store(v.x, &y[i].x); __builtin_nontemporal_store(v.y, &y[i].y);
            __builtin_nontemporal_store(v.z, &y[i].z); __builtin_nontemporal_store(v.w, &y[i].w);
        } else y[i] = v;
    }
}

template <bool NT>
__global__ void __launch_bounds__(256) flat_nt(uint4 *x, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v;
        v.x = __builtin_nontemporal_load(&x[i].x); v.y = __builtin_nontemporal_load(&x[i].y);
        v.z = __builtin_nontemporal_load(&x[i].z); v.w = __builtin_nontemporal_load(&x[i].w);
        v.x ^= 1u;
        __builtin_nontemporal_store(v.x, &x[i].x); __builtin_nontemporal_store(v.y, &x[i].y);
        __builtin_nontemporal_store(v.z, &x[i].z); __builtin_nontemporal_store(v.w, &x[i].w);
    }
}

// (1) wave per ROWS consecutive rows, rows dealt round-robin to persistent waves (the sweep's shape)
template <int ROWS>
__global__ void __launch_bounds__(256) wave_rows(uint8_t *x, uint32_t rows, uint32_t pitch)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t r0 = wave * ROWS; r0 < rows; r0 += nwaves * ROWS) {
        uint4 v[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; k++)
            if (r0 + k < rows) v[k] = *(const uint4 *)(x + (size_t)(r0 + k) * pitch + lane * 16u);
#pragma unroll
        for (int k = 0; k < ROWS; k++)
            if (r0 + k < rows) { v[k].x ^= 1u; *(uint4 *)(x + (size_t)(r0 + k) * pitch + lane * 16u) = v[k]; }
    }
}

// (2) the same through LDS with a random byte gather (16 ds_read_u8 per lane), no RNG work
template <int ROWS>
__global__ void __launch_bounds__(256) wave_rows_gather(uint8_t *x, const uint32_t *idx, uint32_t rows, uint32_t pitch)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[4][ROWS * 1024];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    uint32_t pidx[16];
    for (int k = 0; k < 16; k++) pidx[k] = idx[lane * 16 + k];
    for (uint32_t r0 = wave * ROWS; r0 < rows; r0 += nwaves * ROWS) {
#pragma unroll
        for (int k = 0; k < ROWS; k++)
            if (r0 + k < rows) *(uint4 *)(buf[w] + k * 1024 + lane * 16u) = *(const uint4 *)(x + (size_t)(r0 + k) * pitch + lane * 16u);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            if (r0 + k >= rows) break;
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t t = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) t |= (uint32_t)buf[w][k * 1024 + pidx[4 * j + b]] << (8 * b);
                o[j] = t;
            }
            *(uint4 *)(x + (size_t)(r0 + k) * pitch + lane * 16u) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int main()
{
    const uint32_t rows = 1200000, pitch = 1024;
    const size_t bytes = (size_t)rows * pitch;
    uint8_t *x;
    uint32_t *idx;
    hipMalloc(&x, bytes);
    hipMemset(x, 1, bytes);
    hipMalloc(&idx, 1024 * 4);
    uint32_t h[1024];
    uint32_t s = 12345;
    for (int i = 0; i < 1024; i++) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) % 1000u; }
    hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int n = 20;
        for (int i = 0; i < n; i++) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= n;
        printf("%-34s %.3f ms  %.0f GB/s (read+write)\n", name, ms, 2.0 * bytes / ms / 1e6);
    };
    for (int bpc : { 4, 8 }) {
        char nm[64];
        snprintf(nm, sizeof nm, "flat 16B/thread, %d blocks/CU", bpc);
        time(nm, [&] { flat<<<256 * bpc, 256>>>((uint4 *)x, bytes / 16); });
        snprintf(nm, sizeof nm, "wave x 3 rows, %d blocks/CU", bpc);
        time(nm, [&] { wave_rows<3><<<256 * bpc, 256>>>(x, rows, pitch); });
        snprintf(nm, sizeof nm, "wave x 4 rows, %d blocks/CU", bpc);
        time(nm, [&] { wave_rows<4><<<256 * bpc, 256>>>(x, rows, pitch); });
        snprintf(nm, sizeof nm, "wave x 3 rows + LDS gather, %d b/CU", bpc);
        time(nm, [&] { wave_rows_gather<3><<<256 * bpc, 256>>>(x, idx, rows, pitch); });
    }
    uint8_t *y;
    hipMalloc(&y, bytes);
    hipMemset(y, 1, bytes);
    for (int bpc : { 4, 8 }) {
        char nm[64];
        snprintf(nm, sizeof nm, "copy x->y, %d blocks/CU", bpc);
        time(nm, [&] { flat_copy<false><<<256 * bpc, 256>>>((const uint4 *)x, (uint4 *)y, bytes / 16); });
        snprintf(nm, sizeof nm, "copy x->y nontemporal, %d blocks/CU", bpc);
        time(nm, [&] { flat_copy<true><<<256 * bpc, 256>>>((const uint4 *)x, (uint4 *)y, bytes / 16); });
        snprintf(nm, sizeof nm, "in place nontemporal, %d blocks/CU", bpc);
        time(nm, [&] { flat_nt<true><<<256 * bpc, 256>>>((uint4 *)x, bytes / 16); });
    }
    time("copy x->y, one pass grid", [&] { flat_copy<false><<<(unsigned)(bytes / 16 / 256), 256>>>((const uint4 *)x, (uint4 *)y, bytes / 16); });
    time("flat 16B/thread, 16 blocks/CU", [&] { flat<<<256 * 16, 256>>>((uint4 *)x, bytes / 16); });
    time("flat, one pass grid (n16/256 blocks)", [&] { flat<<<(unsigned)(bytes / 16 / 256), 256>>>((uint4 *)x, bytes / 16); });
    return 0;
}
