// Micro-benchmark: issue cost of single gfx950 VALU instructions (inline asm, 8 independent
// destination registers per loop body, no memory traffic).
// hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 2048
#define REP8(X) X(r0) X(r1) X(r2) X(r3) X(r4) X(r5) X(r6) X(r7)

#define KERNEL(NAME, ASM)                                                                  \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed)             \
    {                                                                                      \
        uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 11u,  \
                 r5 = r0 * 13u, r6 = r0 * 17u, r7 = r0 * 19u;                             \
        uint32_t a = r0 ^ 0x55u, b = r1 + 99u;                                             \
        for (int i = 0; i < ITERS; i++) {                                                  \
            REP8(ASM)                                                                      \
        }                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7; \
    }

#define A_ADD(r) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(r) : "v"(a));
#define A_XOR(r) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(r) : "v"(a));
#define A_AND(r) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(r) : "v"(a));
#define A_LSHL(r) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(r));
#define A_LSHR(r) asm volatile("v_lshrrev_b32_e32 %0, 1, %0" : "+v"(r));
#define A_OR3(r) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define A_ADD3(r) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define A_PERM(r) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define A_BITOP3(r) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r) : "v"(a), "v"(b));
#define A_LSHLADD(r) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r) : "v"(a));
#define A_LSHLOR(r) asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(r) : "v"(a));
#define A_ANDOR(r) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define A_BFE(r) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(r));
#define A_MBCNT(r) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(r) : "v"(a));
#define A_FFBL(r) asm volatile("v_ffbl_b32_e32 %0, %0" : "+v"(r));
#define A_BCNT(r) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r) : "v"(a));
#define A_MULLO(r) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(a));
#define A_MULHI(r) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r) : "v"(a));
#define A_MUL24(r) asm volatile("v_mul_u32_u24_e32 %0, %1, %0" : "+v"(r) : "v"(a));
#define A_CNDMASK(r) asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(r) : "v"(a) : );
#define A_CMP32(r) asm volatile("v_cmp_lt_u32_e32 vcc, %1, %0" : "+v"(r) : "v"(a) : "vcc");
#define A_CMP64(r) asm volatile("v_cmp_lt_u32_e64 s[10:11], %1, %0" : "+v"(r) : "v"(a) : "s10", "s11");
#define A_ALIGNBIT(r) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(r) : "v"(a));
#define A_ADD_E64(r) asm volatile("v_add_u32_e64 %0, %1, %0" : "+v"(r) : "v"(a));
#define A_XAD(r) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define A_MOV(r) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(r) : "v"(a));
#define A_SDWA(r) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(r) : "v"(a));

KERNEL(k_add, A_ADD) KERNEL(k_xor, A_XOR) KERNEL(k_and, A_AND) KERNEL(k_lshl, A_LSHL) KERNEL(k_lshr, A_LSHR)
KERNEL(k_or3, A_OR3) KERNEL(k_add3, A_ADD3) KERNEL(k_perm, A_PERM) KERNEL(k_bitop3, A_BITOP3)
KERNEL(k_lshladd, A_LSHLADD) KERNEL(k_lshlor, A_LSHLOR) KERNEL(k_andor, A_ANDOR) KERNEL(k_bfe, A_BFE)
KERNEL(k_mbcnt, A_MBCNT) KERNEL(k_ffbl, A_FFBL) KERNEL(k_bcnt, A_BCNT) KERNEL(k_mullo, A_MULLO)
KERNEL(k_mulhi, A_MULHI) KERNEL(k_mul24, A_MUL24) KERNEL(k_cndmask, A_CNDMASK) KERNEL(k_cmp32, A_CMP32)
KERNEL(k_cmp64, A_CMP64) KERNEL(k_alignbit, A_ALIGNBIT) KERNEL(k_add64, A_ADD_E64) KERNEL(k_xad, A_XAD)
KERNEL(k_mov, A_MOV) KERNEL(k_sdwa, A_SDWA)

template <typename K>
static void run(const char *name, K kern, int waves_per_simd)
{
    uint32_t *out;
    const int blocks = 256 * waves_per_simd;   // 4 waves per block, 4 SIMDs per CU
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<blocks, 256>>>(out, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)waves_per_simd * ITERS * 8;
    printf("%-22s waves/SIMD=%d  %.2f cycles per wave-instruction per SIMD (@2.4 GHz)\n", name, waves_per_simd,
           ms * 1e6 / n * 2.4);
    hipFree(out);
}

int main()
{
    for (int w : { 2, 8 }) {
#define R(NAME) run(#NAME, k_##NAME, w);
        R(add) R(xor) R(and) R(lshl) R(lshr) R(mov) R(add64) R(sdwa) R(or3) R(add3) R(xad) R(perm) R(bitop3) R(lshladd) R(lshlor) R(andor)
        R(bfe) R(alignbit) R(mbcnt) R(ffbl) R(bcnt) R(mullo) R(mulhi) R(mul24) R(cndmask) R(cmp32) R(cmp64)
    }
    return 0;
}
