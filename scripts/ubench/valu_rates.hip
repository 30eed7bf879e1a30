// Micro-benchmark: issue cost of the integer ops Philox is made of, on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 4096

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed)
{
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = b + 7, d = c ^ a;
    for (int i = 0; i < ITERS; i++) {
        if (OP == 0) { a += b; b ^= c; c += d; d ^= a; }
        if (OP == 1) {
            uint64_t p0 = (uint64_t)0xD2511F53u * a, p1 = (uint64_t)0xCD9E8D57u * c;
            a = (uint32_t)(p1 >> 32) ^ b; b = (uint32_t)p1; c = (uint32_t)(p0 >> 32) ^ d; d = (uint32_t)p0;
        }
        if (OP == 2) { a = __umulhi(a, 0xD2511F53u) ^ b; b = __umulhi(c, 0xCD9E8D57u) ^ d; c += a; d += b; }
        if (OP == 3) { a = a * 0xD2511F53u ^ b; b = c * 0xCD9E8D57u ^ d; c += a; d += b; }
        if (OP == 4) { a = __umul24(a, b) ^ c; b = __umul24(c, d) ^ a; c += a; d += b; }
        if (OP == 5) {
            asm volatile("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c));
            asm volatile("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(b) : "v"(c), "v"(d), "v"(a));
            c += a; d += b;
        }
        if (OP == 6) { a = __builtin_amdgcn_perm(a, b, 0x05010400u); b = __builtin_amdgcn_perm(c, d, 0x07030602u); c += a; d += b; }
        if (OP == 7) { a = __builtin_amdgcn_perm(a, b, c); b = __builtin_amdgcn_perm(c, d, a); c += a; d += b; }
        if (OP == 8) { a = (a << 4) | b; b = (c << 4) | d; c += a; d += b; }                     // v_lshl_or_b32
        if (OP == 9) { a = __builtin_amdgcn_ubfe(a, 8, 4) + b; b = __builtin_amdgcn_ubfe(c, 16, 4) + d; c += a; d += b; }
        if (OP == 10) { a = __builtin_popcount(a ^ b) + c; b = __builtin_popcount(c ^ d) + a; c += a; d += b; } // xor + bcnt
        if (OP == 11) { a = (a & 0xFFFFu) + b; b = (c >> 16) + d; c += a; d += b; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int OP>
static void run(const char *name, int ops_per_iter, int blocks_per_cu)
{
    uint32_t *out;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks_per_cu*4 waves per CU / 4 SIMDs = blocks_per_cu waves per SIMD
    const double winstr_per_simd = (double)blocks_per_cu * ITERS * ops_per_iter;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name,
           blocks_per_cu, ms, ms * 1e6 / winstr_per_simd, ms * 1e6 / winstr_per_simd * 2.4);
    hipFree(out);
}

int main()
{
    for (int w : { 1, 2, 4, 8 }) {
        run<0>("add/xor (4 ops)", 4, w);
        run<1>("2x mad_u64_u32 + 2 xor", 4, w);
        run<2>("2x mul_hi + 2 xor + 2 add", 6, w);
        run<3>("2x mul_lo + 2 xor + 2 add", 6, w);
        run<4>("2x mul_u32_u24 + 2xor + 2add", 6, w);
        run<5>("2x mad_u32_u16 + 2 add", 4, w);
        run<6>("2x perm (const sel) + 2 add", 4, w);
        run<7>("2x perm (vgpr sel) + 2 add", 4, w);
        run<8>("2x lshl_or + 2 add", 4, w);
        run<9>("2x bfe + 4 add", 6, w);
        run<10>("2x xor + 2x bcnt + 2 add", 6, w);
        run<11>("and/shr + 4 add", 6, w);
    }
    return 0;
}
