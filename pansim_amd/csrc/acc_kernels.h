// acc_kernels.h -- gfx950 kernels for the accessory (gene presence/absence) matrix.
//
// HBM layout (bit-packed, two coherent views):
//   G-major  accG[g][w]  (u64, W = ceil(N/64) words per gene): bit (i & 63) of word
//            i >> 6 is the presence of gene g in individual i.  One wave64 ballot
//            builds one word, so gather/mutation edit the bitset with ballots.
//   I-major  accI[i][gw] (u64, GW = ceil(G/64) words per individual): the row of an
//            individual; used for rank/select of HGT donors (population.rs:636-680),
//            per-individual fitness sums (:282-322) and Jaccard pairs (:824-830).
// Padding bits (i >= N, g >= G) are always zero.
#pragma once

#include "ps_common.h"

struct acc_dims { uint32_t N, G, W, GW; };

// clonal start (population.rs:221-229)
__global__ void acc_init_kernel(uint64_t *accG, uint64_t *accI, const uint8_t *init_vec, acc_dims d)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nG = (uint64_t)d.G * d.W, nI = (uint64_t)d.N * d.GW;
    if (t < nG) {
        const uint32_t g = (uint32_t)(t / d.W), w = (uint32_t)(t % d.W);
        const uint32_t nb = min(64u, d.N - w * 64u);
        const uint64_t m = (nb == 64u) ? ~0ull : ((1ull << nb) - 1ull);
        accG[t] = init_vec[g] ? m : 0ull;
    } else if (t < nG + nI) {
        const uint64_t u = t - nG;
        const uint32_t gw = (uint32_t)(u % d.GW);
        uint64_t word = 0;
        for (uint32_t b = 0; b < 64; b++) {
            const uint32_t g = gw * 64u + b;
            if (g < d.G && init_vec[g]) word |= 1ull << b;
        }
        accI[u] = word;
    }
}

// u8 rows[N][G] -> I-major
__global__ void acc_pack_rows_kernel(const uint8_t *rows, uint64_t *accI, acc_dims d)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)d.N * d.GW) return;
    const uint32_t i = (uint32_t)(t / d.GW), gw = (uint32_t)(t % d.GW);
    uint64_t word = 0;
    for (uint32_t b = 0; b < 64; b++) {
        const uint32_t g = gw * 64u + b;
        if (g < d.G && rows[(uint64_t)i * d.G + g] != 0) word |= 1ull << b;
    }
    accI[t] = word;
}

// I-major -> G-major (one wave per (gene block, individual word): ballot per gene)
__global__ void __launch_bounds__(64) acc_i_to_g_kernel(const uint64_t *accI, uint64_t *accG, acc_dims d)
{
    const uint32_t w = blockIdx.x, gw = blockIdx.y, lane = threadIdx.x;
    const uint32_t i = w * 64u + lane;
    const uint64_t mine = (i < d.N) ? accI[(uint64_t)i * d.GW + gw] : 0ull;
    for (uint32_t b = 0; b < 64; b++) {
        const uint32_t g = gw * 64u + b;
        if (g >= d.G) break;
        const uint64_t word = __ballot((mine >> b) & 1ull);
        if (lane == 0) accG[(uint64_t)g * d.W + w] = word;
    }
}

// I-major -> u8 rows[N][G]
__global__ void acc_unpack_rows_kernel(const uint64_t *accI, uint8_t *rows, acc_dims d)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)d.N * d.G) return;
    const uint32_t i = (uint32_t)(t / d.G), g = (uint32_t)(t % d.G);
    rows[t] = (uint8_t)((accI[(uint64_t)i * d.GW + (g >> 6)] >> (g & 63u)) & 1ull);
}

struct acc_step_args {
    const uint64_t *srcG;
    uint64_t *dstG, *dstI;
    const uint32_t *idx;      // parents; may alias host-mapped pinned memory (read once per lane)
    uint32_t *idx_out;        // if set, the gene-block-0 waves publish the parents in device memory
    acc_dims d;
    uint32_t gen, k0, k1;
    ps_acc_plan plan;
};

// Fused parent gather (population.rs:450-465) and gain/loss (population.rs:486-510):
// lane = individual, loop over the 64 genes of the block; the new bits of 64
// individuals are assembled with one ballot per gene.
template <bool DO_GATHER, bool DO_MUT>
__global__ void __launch_bounds__(64) acc_step_kernel(acc_step_args a)
{
    const uint32_t w = blockIdx.x, gw = blockIdx.y, lane = threadIdx.x;
    const acc_dims d = a.d;
    const uint32_t i = w * 64u + lane;
    const bool valid = i < d.N;
    const uint32_t p = (DO_GATHER && valid) ? a.idx[i] : (valid ? i : 0u);
    if (DO_GATHER && a.idx_out && gw == 0 && valid) a.idx_out[i] = p;
    const uint32_t pw = p >> 6, pb = p & 63u;
    uint64_t rowword = 0;
    ps_u4 rnd = { 0, 0, 0, 0 };
    for (uint32_t b = 0; b < 64; b++) {
        const uint32_t g = gw * 64u + b;
        if (g >= d.G) break;
        uint32_t bit = (uint32_t)((a.srcG[(uint64_t)g * d.W + pw] >> pb) & 1ull);
        if (DO_MUT) {
            if ((b & 3u) == 0u)
                rnd = ps_philox(g >> 2, i, a.gen, PS_STREAM_ACC_MUT, a.k0, a.k1);
            const uint32_t word = ((b & 3u) == 0u) ? rnd.x : ((b & 3u) == 1u) ? rnd.y
                                  : ((b & 3u) == 2u) ? rnd.z : rnd.w;
            uint32_t thr = 0;
#pragma unroll
            for (int c = 0; c < PS_MAX_COMP; c++)
                if (c < a.plan.n_comp && g >= a.plan.comp_begin[c] && g < a.plan.comp_end[c])
                    thr = a.plan.flip_thr[c];
            bit ^= (word < thr) ? 1u : 0u;                 // population.rs:505
        }
        bit = valid ? bit : 0u;
        const uint64_t gword = __ballot(bit);
        if (lane == 0) a.dstG[(uint64_t)g * d.W + w] = gword;
        rowword |= (uint64_t)bit << b;
    }
    if (valid) a.dstI[(uint64_t)i * d.GW + gw] = rowword;
}

// Rank/select tables for HGT donors (population.rs:636-680 builds a per-donor
// WeightedIndex over the donor's present genes; this is its bitset form).  One wave
// per (individual, compartment): lane = row word, popcount, wave prefix sum, then
// every lane writes the positions of its set bits.  list[i*G + comp_begin + j] is
// the j-th present gene of individual i inside the compartment; cnt[c*N + i] their
// number.
__global__ void __launch_bounds__(64) acc_gene_lists_kernel(const uint64_t *accI, uint16_t *list,
                                                            uint32_t *cnt, acc_dims d, ps_acc_plan plan)
{
    const uint32_t i = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
    const uint32_t gb = plan.comp_begin[c], ge = plan.comp_end[c];
    uint32_t base = 0;
    for (uint32_t gw0 = gb >> 6; gw0 * 64u < ge; gw0 += 64u) {
        const uint32_t gw = gw0 + lane;
        uint64_t word = 0;
        if (gw * 64u < ge && gw < d.GW) {
            word = accI[(uint64_t)i * d.GW + gw];
            const uint32_t lo = gw * 64u;
            if (lo < gb) word &= ~0ull << (gb - lo);
            if (lo + 64u > ge) word &= (ge - lo >= 64u) ? ~0ull : ((1ull << (ge - lo)) - 1ull);
        }
        const uint32_t pc = __popcll(word);
        uint32_t incl = pc;                      // inclusive wave prefix sum
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if ((int)lane >= off) incl += t;
        }
        uint32_t pos = base + incl - pc;
        while (word) {
            const uint32_t b = (uint32_t)__builtin_ctzll(word);
            word &= word - 1ull;
            list[(uint64_t)i * d.G + gb + pos] = (uint16_t)(gw * 64u + b);
            pos++;
        }
        base += __shfl(incl, 63, 64);
    }
    if (lane == 0) cnt[(uint64_t)c * d.N + i] = base;
}

// HGT events (population.rs:544-751 accessory path):
// event e of compartment c picks a uniform donor, a uniform other recipient and a
// uniform gene among the donor's present genes of the compartment IN THE SNAPSHOT
// (the gene lists); the recipient gains the gene (value always 1, :632) -- an
// idempotent OR, so the order of events does not matter.  Only the individual-major
// view is edited here; acc_i_to_g_kernel rebuilds the gene-major view afterwards.
struct acc_hgt_args {
    const uint16_t *list;
    const uint32_t *cnt;
    uint64_t *dstI;
    acc_dims d;
    uint32_t n_comp;
    uint32_t gb[PS_MAX_COMP];
    uint64_t K[PS_MAX_COMP];      // events per compartment (0 = skipped)
    uint32_t gen, k0, k1;
    uint32_t *scratch;            // LDS-partitioned kernel: slice images [n_slices][N][2*GW] (or null: atomics)
};

__global__ void __launch_bounds__(256) acc_hgt_kernel(acc_hgt_args a, uint32_t comp)
{
    // One compartment per launch, one event per thread and iteration, one 64-bit atomicOr per
    // event.  Deliberately light: this kernel runs beside the persistent core sweep, and a
    // heavier variant (several events in flight, test-before-set loads) measurably delayed the
    // sweep's workgroups (profiles/r01_sweep_ablation.md).
    const acc_dims d = a.d;
    const uint64_t K = a.K[comp];
    const uint32_t gb = a.gb[comp];
    const uint32_t stream = PS_STREAM_HGT | (comp << 8);
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < K;
         e += (uint64_t)gridDim.x * blockDim.x) {
        const ps_u4 r = ps_philox((uint32_t)e, (uint32_t)(e >> 32), a.gen, stream, a.k0, a.k1);
        const uint32_t dn = ps_mulhi(r.x, d.N);
        uint32_t rc = ps_mulhi(r.y, d.N - 1u);
        rc += (rc >= dn) ? 1u : 0u;                         // population.rs:618
        const uint32_t n = a.cnt[(uint64_t)comp * d.N + dn];
        if (n == 0) continue;                               // population.rs:672
        const uint32_t gene = a.list[(uint64_t)dn * d.G + gb + ps_mulhi(r.z, n)];
        atomicOr((unsigned long long *)&a.dstI[(uint64_t)rc * d.GW + (gene >> 6)], 1ull << (gene & 63u));
    }
}

// Heavy HGT (cfg3-like rates: many events per matrix cell).  Most events re-set a bit that is
// already 1, so the recipient word is tested with a plain load first and the atomic is issued
// only when the bit still looks clear (bits are only ever set: a stale 1 cannot occur, a stale 0
// costs one redundant atomic).  Four events per thread are in flight to cover the dependent
// cnt -> list -> word chain.  No LDS: co-runs with the core sweep.
__global__ void __launch_bounds__(256) acc_hgt_tbs_kernel(acc_hgt_args a, uint32_t comp)
{
    const acc_dims d = a.d;
    const uint64_t K = a.K[comp];
    const uint32_t gb = a.gb[comp];
    const uint32_t stream = PS_STREAM_HGT | (comp << 8);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e0 < K; e0 += 4ull * stride) {
        uint32_t dn[4], rc[4], n[4], gz[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t e = e0 + (uint64_t)u * stride;
            ok[u] = e < K;
            const ps_u4 r = ps_philox((uint32_t)e, (uint32_t)(e >> 32), a.gen, stream, a.k0, a.k1);
            dn[u] = ps_mulhi(r.x, d.N);
            rc[u] = ps_mulhi(r.y, d.N - 1u);
            rc[u] += (rc[u] >= dn[u]) ? 1u : 0u;                // population.rs:618
            gz[u] = r.z;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) n[u] = a.cnt[(uint64_t)comp * d.N + dn[u]];
        uint32_t gene[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            ok[u] = ok[u] && n[u] != 0u;                        // population.rs:672
            gene[u] = a.list[(uint64_t)dn[u] * d.G + gb + ps_mulhi(gz[u], max(n[u], 1u))];
        }
        uint64_t cur[4];
#pragma unroll
        for (int u = 0; u < 4; u++) cur[u] = a.dstI[(uint64_t)rc[u] * d.GW + (gene[u] >> 6)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t bit = 1ull << (gene[u] & 63u);
            if (ok[u] && !(cur[u] & bit))
                atomicOr((unsigned long long *)&a.dstI[(uint64_t)rc[u] * d.GW + (gene[u] >> 6)], bit);
        }
    }
}

// Rank/select tables in place of the gene lists, for heavy HGT beside the core sweep.  The
// N x G list (8 MB at cfg2) does not survive in L2 while the sweep streams the core matrix, so
// every event's list read went to HBM; these tables are ~0.65 KB per individual and stay resident:
//   snap[i][GW]   u64  the pre-recombination row (the snapshot donors are read from, :693-695)
//   cum[i][GW8]   u16  set bits of the row before word w (GW8 = GW rounded up to 8; padding 0xFFFF)
//   top[i][NG8]   u16  cum[i][8 g] (one entry per group of 8 words; padding 0xFFFF)
//   nb[c*N + i]   u32  (genes of i inside compartment c) | (set bits before the compartment) << 16
// "the j-th present gene of the compartment" is then the (base + j)-th set bit of the row.
struct acc_select_tabs {
    uint64_t *snap;
    uint16_t *cum, *top;
    uint32_t *nb;
    uint32_t GW8, NG8;
};

__global__ void __launch_bounds__(64) acc_rank_tables_kernel(const uint64_t *accI, acc_select_tabs t, acc_dims d,
                                                             ps_acc_plan plan)
{
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    uint32_t running = 0, cn[PS_MAX_COMP], cb[PS_MAX_COMP];
#pragma unroll
    for (int c = 0; c < PS_MAX_COMP; c++) { cn[c] = 0; cb[c] = 0; }
    for (uint32_t gw0 = 0; gw0 < t.GW8; gw0 += 64u) {
        const uint32_t gw = gw0 + lane;
        const uint64_t word = (gw < d.GW) ? accI[(uint64_t)i * d.GW + gw] : 0ull;
        if (gw < d.GW) t.snap[(uint64_t)i * d.GW + gw] = word;
        const uint32_t pc = __popcll(word);
        uint32_t incl = pc;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off, 64);
            if ((int)lane >= off) incl += u;
        }
        const uint32_t excl = running + incl - pc;
        if (gw < t.GW8) t.cum[(uint64_t)i * t.GW8 + gw] = (gw < d.GW) ? (uint16_t)excl : (uint16_t)0xFFFFu;
        if (gw < t.GW8 && (gw & 7u) == 0u) t.top[(uint64_t)i * t.NG8 + (gw >> 3)] = (gw < d.GW) ? (uint16_t)excl : (uint16_t)0xFFFFu;
        const uint32_t lo = gw * 64u;
#pragma unroll
        for (int c = 0; c < PS_MAX_COMP; c++) {
            if (c < plan.n_comp) {
                const uint32_t gb = plan.comp_begin[c], ge = plan.comp_end[c];
                // bits of this word below gb, and inside [gb, ge)
                const uint64_t below_gb = (gb <= lo) ? 0ull : (gb - lo >= 64u) ? ~0ull : ((1ull << (gb - lo)) - 1ull);
                const uint64_t below_ge = (ge <= lo) ? 0ull : (ge - lo >= 64u) ? ~0ull : ((1ull << (ge - lo)) - 1ull);
                cb[c] += __popcll(word & below_gb);
                cn[c] += __popcll(word & below_ge & ~below_gb);
            }
        }
        running += __shfl(incl, 63, 64);
    }
    // groups of `top` beyond the last word chunk (NG8 > GW8 / 8 never happens: NG8 = GW8 / 8 rounded up to 8)
    for (uint32_t g = t.GW8 / 8u + lane; g < t.NG8; g += 64u) t.top[(uint64_t)i * t.NG8 + g] = (uint16_t)0xFFFFu;
#pragma unroll
    for (int c = 0; c < PS_MAX_COMP; c++) {
        if (c < plan.n_comp) {
            uint32_t n = cn[c], b = cb[c];
            for (int off = 32; off > 0; off >>= 1) { n += __shfl_down(n, off, 64); b += __shfl_down(b, off, 64); }
            if (lane == 0) t.nb[(uint64_t)c * d.N + i] = n | (b << 16);
        }
    }
}

// entries (two u16 per dword, ascending) that are <= t: their number, and the largest of them
__device__ __forceinline__ void ps_le_count_max(uint32_t x, uint32_t t, uint32_t &count, uint32_t &maxle)
{
    const uint32_t lo = x & 0xFFFFu, hi = x >> 16;
    if (lo <= t) { count++; maxle = max(maxle, lo); }
    if (hi <= t) { count++; maxle = max(maxle, hi); }
}

// position of the t-th (0-based) set bit of a 64-bit word, t < popcount(word)
__device__ __forceinline__ uint32_t ps_select64(uint64_t word, uint32_t t)
{
    uint32_t x = (uint32_t)word, pos = 0;
    uint32_t c = __popc(x);
    if (t >= c) { t -= c; x = (uint32_t)(word >> 32); pos = 32u; }
    c = __popc(x & 0xFFFFu);
    if (t >= c) { t -= c; x >>= 16; pos += 16u; }
    c = __popc(x & 0xFFu);
    if (t >= c) { t -= c; x >>= 8; pos += 8u; }
    c = __popc(x & 0xFu);
    if (t >= c) { t -= c; x >>= 4; pos += 4u; }
    c = __popc(x & 3u);
    if (t >= c) { t -= c; x >>= 2; pos += 2u; }
    c = x & 1u;
    if (t >= c) pos += 1u;
    return pos;
}

struct acc_hgt_select_args {
    acc_select_tabs t;
    uint64_t *dstI;
    acc_dims d;
    uint64_t K[PS_MAX_COMP];
    uint32_t gen, k0, k1;
};

// Heavy HGT through the rank/select tables; the recipient word is tested before the atomic (see
// acc_hgt_tbs_kernel).  Two events per thread in flight.
__global__ void __launch_bounds__(256) acc_hgt_select_kernel(acc_hgt_select_args a, uint32_t comp)
{
    const acc_dims d = a.d;
    const uint64_t K = a.K[comp];
    const uint32_t stream = PS_STREAM_HGT | (comp << 8);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    constexpr int U = 2;
    for (uint64_t e0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e0 < K; e0 += (uint64_t)U * stride) {
        uint32_t dn[U], rc[U], gz[U], nb[U], tgt[U], grp[U], wd[U], gene[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t e = e0 + (uint64_t)u * stride;
            ok[u] = e < K;
            const ps_u4 r = ps_philox((uint32_t)e, (uint32_t)(e >> 32), a.gen, stream, a.k0, a.k1);
            dn[u] = ps_mulhi(r.x, d.N);
            rc[u] = ps_mulhi(r.y, d.N - 1u);
            rc[u] += (rc[u] >= dn[u]) ? 1u : 0u;                // population.rs:618
            gz[u] = r.z;
        }
#pragma unroll
        for (int u = 0; u < U; u++) nb[u] = a.t.nb[(uint64_t)comp * d.N + dn[u]];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t n = nb[u] & 0xFFFFu;
            ok[u] = ok[u] && n != 0u;                           // population.rs:672
            tgt[u] = (nb[u] >> 16) + ps_mulhi(gz[u], n);        // rank of the gene inside the whole row
            // group of 8 words: number of `top` entries <= target, minus one
            uint32_t count = 0, mx = 0;
            for (uint32_t g0 = 0; g0 < a.t.NG8; g0 += 8u) {
                const uint4 v = *(const uint4 *)(a.t.top + (uint64_t)dn[u] * a.t.NG8 + g0);
                ps_le_count_max(v.x, tgt[u], count, mx);
                ps_le_count_max(v.y, tgt[u], count, mx);
                ps_le_count_max(v.z, tgt[u], count, mx);
                ps_le_count_max(v.w, tgt[u], count, mx);
            }
            grp[u] = ok[u] ? count - 1u : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint4 v = *(const uint4 *)(a.t.cum + (uint64_t)dn[u] * a.t.GW8 + 8u * grp[u]);
            uint32_t count = 0, mx = 0;
            ps_le_count_max(v.x, tgt[u], count, mx);
            ps_le_count_max(v.y, tgt[u], count, mx);
            ps_le_count_max(v.z, tgt[u], count, mx);
            ps_le_count_max(v.w, tgt[u], count, mx);
            wd[u] = ok[u] ? 8u * grp[u] + count - 1u : 0u;
            tgt[u] -= ok[u] ? mx : tgt[u];                      // rank inside the word
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t word = a.t.snap[(uint64_t)dn[u] * d.GW + wd[u]];
            gene[u] = wd[u] * 64u + (ok[u] ? ps_select64(word, tgt[u]) : 0u);
        }
        uint64_t cur[U];
#pragma unroll
        for (int u = 0; u < U; u++) cur[u] = a.dstI[(uint64_t)rc[u] * d.GW + (gene[u] >> 6)];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t bit = 1ull << (gene[u] & 63u);
            if (ok[u] && !(cur[u] & bit))
                atomicOr((unsigned long long *)&a.dstI[(uint64_t)rc[u] * d.GW + (gene[u] >> 6)], bit);
        }
    }
}

// HGT with the recipients tiled through LDS (small populations, many events): workgroup
// (part, slice) keeps the row words of the recipients of partition `part` in LDS, scans event
// slice `slice`, ORs the events that land in its partition with LDS atomics and finally merges
// its non-zero words into HBM with coalesced atomics.  Every partition re-derives the slice's
// events (a Philox call is far cheaper than a scattered global atomic); same events, same keys.
__global__ void __launch_bounds__(1024) acc_hgt_lds_kernel(acc_hgt_args a, uint32_t rows_per_part, uint32_t n_slices)
{
    extern __shared__ uint32_t lrow[];     // [rows_per_part][2*GW] 32-bit words
    const acc_dims d = a.d;
    const uint32_t part = blockIdx.x / n_slices, slice = blockIdx.x % n_slices;
    const uint32_t r_lo = part * rows_per_part, r_hi = min(d.N, r_lo + rows_per_part);
    const uint32_t W32 = 2u * d.GW;
    for (uint32_t w = threadIdx.x; w < rows_per_part * W32; w += blockDim.x) lrow[w] = 0u;
    __syncthreads();
    uint64_t total = 0;
    total = a.K[0] + (a.n_comp > 1 ? a.K[1] : 0ull);
    const uint64_t K0 = a.K[0];
    const uint32_t gb0 = a.gb[0], gb1 = a.gb[1];
    const uint64_t per = (total + n_slices - 1) / n_slices;
    const uint64_t t_end = min(total, (uint64_t)(slice + 1) * per);
    for (uint64_t t = (uint64_t)slice * per + threadIdx.x; t < t_end; t += blockDim.x) {
        uint64_t e = t;
        const uint32_t comp = (a.n_comp > 1 && e >= K0) ? 1u : 0u;
        e -= comp ? K0 : 0ull;
        const ps_u4 r = ps_philox((uint32_t)e, (uint32_t)(e >> 32), a.gen, PS_STREAM_HGT | (comp << 8), a.k0, a.k1);
        const uint32_t dn = ps_mulhi(r.x, d.N);
        uint32_t rc = ps_mulhi(r.y, d.N - 1u);
        rc += (rc >= dn) ? 1u : 0u;                         // population.rs:618
        if (rc < r_lo || rc >= r_hi) continue;              // another partition's recipient
        const uint32_t n = a.cnt[(uint64_t)comp * d.N + dn];
        if (n == 0) continue;                               // population.rs:672
        const uint32_t gene = a.list[(uint64_t)dn * d.G + (comp ? gb1 : gb0) + ps_mulhi(r.z, n)];
        atomicOr(&lrow[(rc - r_lo) * W32 + (gene >> 5)], 1u << (gene & 31u));
    }
    __syncthreads();
    if (a.scratch) {
        // publish this (partition, slice) image with plain coalesced stores; acc_hgt_reduce_kernel
        // ORs the slices into the matrix (64 slices x 40 K words of atomics took longer than the events)
        uint32_t *img = a.scratch + ((uint64_t)slice * d.N + r_lo) * W32;
        for (uint32_t w = threadIdx.x; w < (r_hi - r_lo) * W32; w += blockDim.x) img[w] = lrow[w];
        return;
    }
    uint32_t *dst32 = (uint32_t *)a.dstI;
    for (uint32_t w = threadIdx.x; w < (r_hi - r_lo) * W32; w += blockDim.x) {
        const uint32_t v = lrow[w];
        if (v) {
            uint32_t *g = dst32 + (uint64_t)r_lo * W32 + w;
            if ((v & ~__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u) atomicOr(g, v);
        }
    }
}

// Heavy HGT in two passes (every event's Philox call and donor lookup happen ONCE; the
// LDS-partitioned kernel above repeats them per recipient partition):
//   bin    workgroup b derives its contiguous share of the events and appends the effective ones,
//          packed (recipient row inside its partition << 16 | gene), to its own bin of the recipient's
//          partition: bins[(b * parts + part) * cap ...], counts[b * parts + part].  Wave-aggregated
//          appends through LDS counters; no global atomics.
//   apply  workgroup (part, slice) ORs the bins of its slice's workgroups into an LDS image of the
//          partition's rows and publishes the image; acc_hgt_reduce_kernel ORs the images into the matrix.
struct acc_hgt_bin_args {
    acc_hgt_args h;
    uint32_t *bins, *counts;
    uint32_t parts, rows_per_part, cap;
    uint64_t per_block;        // events per bin workgroup
    uint32_t *overflow_flag;
};

__global__ void __launch_bounds__(256) acc_hgt_bin_kernel(acc_hgt_bin_args b)
{
    __shared__ uint32_t fill[8];
    const acc_hgt_args &a = b.h;
    const acc_dims d = a.d;
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < 8u) fill[threadIdx.x] = 0u;
    __syncthreads();
    const uint64_t total = a.K[0] + (a.n_comp > 1 ? a.K[1] : 0ull);
    const uint64_t K0 = a.K[0];
    const uint32_t gb0 = a.gb[0], gb1 = a.gb[1];
    const uint64_t t_lo = (uint64_t)blockIdx.x * b.per_block, t_hi = min(total, t_lo + b.per_block);
    uint32_t *mybins = b.bins + (uint64_t)blockIdx.x * b.parts * b.cap;
    constexpr int U = 2;       // events per thread in flight (covers the cnt -> list chain)
    for (uint64_t t0 = t_lo; t0 < t_hi; t0 += (uint64_t)U * blockDim.x) {
        bool ok[U];
        uint32_t dn[U], rc[U], gz[U], comp[U], n[U], gene[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t t = t0 + (uint64_t)u * blockDim.x + threadIdx.x;
            ok[u] = t < t_hi;
            uint64_t e = ok[u] ? t : t_lo;
            comp[u] = (a.n_comp > 1 && e >= K0) ? 1u : 0u;
            e -= comp[u] ? K0 : 0ull;
            const ps_u4 r = ps_philox((uint32_t)e, (uint32_t)(e >> 32), a.gen, PS_STREAM_HGT | (comp[u] << 8), a.k0, a.k1);
            dn[u] = ps_mulhi(r.x, d.N);
            rc[u] = ps_mulhi(r.y, d.N - 1u);
            rc[u] += (rc[u] >= dn[u]) ? 1u : 0u;                // population.rs:618
            gz[u] = r.z;
        }
#pragma unroll
        for (int u = 0; u < U; u++) n[u] = a.cnt[(uint64_t)comp[u] * d.N + dn[u]];
#pragma unroll
        for (int u = 0; u < U; u++) {
            ok[u] = ok[u] && n[u] != 0u;                        // population.rs:672
            gene[u] = a.list[(uint64_t)dn[u] * d.G + (comp[u] ? gb1 : gb0) + ps_mulhi(gz[u], max(n[u], 1u))];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint32_t part = 0;
            for (uint32_t q = 1; q < b.parts; q++) part += (rc[u] >= q * b.rows_per_part) ? 1u : 0u;
            const uint32_t packed = ((rc[u] - part * b.rows_per_part) << 16) | gene[u];
            for (uint32_t q = 0; q < b.parts; q++) {
                const uint64_t m = __ballot(ok[u] && part == q);
                if (m == 0ull) continue;
                uint32_t base = 0;
                if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&fill[q], (uint32_t)__popcll(m));
                base = __shfl(base, __builtin_ctzll(m), 64);
                if (ok[u] && part == q) {
                    const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (pos < b.cap) mybins[(uint64_t)q * b.cap + pos] = packed;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < b.parts) {
        uint32_t n = fill[threadIdx.x];
        if (n > b.cap) { atomicOr(b.overflow_flag, 8u); n = b.cap; }
        b.counts[(uint64_t)blockIdx.x * b.parts + threadIdx.x] = n;
    }
}

__global__ void __launch_bounds__(1024) acc_hgt_apply_kernel(acc_hgt_bin_args b, uint32_t bin_blocks, uint32_t n_slices)
{
    extern __shared__ uint32_t lrow[];     // [rows_per_part][2*GW] 32-bit words
    const acc_dims d = b.h.d;
    const uint32_t part = blockIdx.x / n_slices, slice = blockIdx.x % n_slices;
    const uint32_t r_lo = part * b.rows_per_part, r_hi = min(d.N, r_lo + b.rows_per_part);
    const uint32_t W32 = 2u * d.GW;
    for (uint32_t w = threadIdx.x; w < b.rows_per_part * W32; w += blockDim.x) lrow[w] = 0u;
    __syncthreads();
    const uint32_t per = (bin_blocks + n_slices - 1u) / n_slices;
    const uint32_t bb_hi = min(bin_blocks, (slice + 1u) * per);
    for (uint32_t bb = slice * per; bb < bb_hi; bb++) {
        const uint32_t n = b.counts[(uint64_t)bb * b.parts + part];
        const uint32_t *src = b.bins + ((uint64_t)bb * b.parts + part) * b.cap;
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
            const uint32_t v = src[k];
            const uint32_t gene = v & 0xFFFFu;
            atomicOr(&lrow[(v >> 16) * W32 + (gene >> 5)], 1u << (gene & 31u));
        }
    }
    __syncthreads();
    uint32_t *img = b.h.scratch + ((uint64_t)slice * d.N + r_lo) * W32;
    for (uint32_t w = threadIdx.x; w < (r_hi - r_lo) * W32; w += blockDim.x) img[w] = lrow[w];
}

// dstI |= OR over the slice images written by acc_hgt_lds_kernel (scratch[slice][N][GW] u64)
__global__ void __launch_bounds__(256) acc_hgt_reduce_kernel(const uint64_t *scratch, uint64_t *dstI, uint64_t words,
                                                             uint32_t n_slices)
{
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    uint64_t v = 0;
    for (uint32_t sl = 0; sl < n_slices; sl++) v |= scratch[(uint64_t)sl * words + w];
    if (v) dstI[w] |= v;
}

// population.rs:282-322: gene count and left-to-right f64 sum of ln(1+s_g) over
// the present genes (absent genes add ln(1+0) = +0.0, which leaves an f64 sum
// unchanged); a present gene with ln(1+s_g) == -inf resets the row to 0.0.
__global__ void acc_fitness_kernel(const uint64_t *accI, const double *log1p_s, int need_logw,
                                   int32_t *num_genes, double *logw, acc_dims d)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.N) return;
    int32_t n = 0;
    double sum = 0.0;
    bool neg_inf = false;
    for (uint32_t gw = 0; gw < d.GW; gw++) {
        uint64_t word = accI[(uint64_t)i * d.GW + gw];
        n += __popcll(word);
        if (need_logw) {
            while (word) {
                const uint32_t b = (uint32_t)__builtin_ctzll(word);
                word &= word - 1ull;
                const double c = log1p_s[gw * 64u + b];
                if (c == -INFINITY) neg_inf = true;
                sum += c;
            }
        }
    }
    num_genes[i] = n;
    logw[i] = neg_inf ? 0.0 : sum;
}

// neutral selection (every ln(1+s_g) is 0): only the gene counts are needed (population.rs:282-291,
// log_sum = 0.0 for every row); one wave per individual, lanes over the row words
__global__ void __launch_bounds__(256) acc_gene_count_rows_kernel(const uint64_t *accI, int32_t *num_genes,
                                                                  double *logw, acc_dims d)
{
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (i >= d.N) return;
    uint32_t n = 0;
    for (uint32_t gw = lane; gw < d.GW; gw += 64u) n += __popcll(accI[(uint64_t)i * d.GW + gw]);
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
    if (lane == 0) { num_genes[i] = (int32_t)n; logw[i] = 0.0; }
}

// population.rs:840-863 numerators: number of individuals carrying gene g
__global__ void acc_gene_counts_kernel(const uint64_t *accG, uint32_t *counts, acc_dims d)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    uint32_t n = 0;
    for (uint32_t w = 0; w < d.W; w++) n += __popcll(accG[(uint64_t)g * d.W + w]);
    counts[g] = n;
}

// distances.rs:55-77 numerators for sampled pairs
__global__ void acc_pair_counts_kernel(const uint64_t *accI, const uint32_t *r1, const uint32_t *r2,
                                       const uint32_t *perm, uint64_t P, uint32_t *inter, uint32_t *uni,
                                       acc_dims d)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const uint64_t *x = accI + (uint64_t)r1[k] * d.GW, *y = accI + (uint64_t)r2[k] * d.GW;
    uint32_t in = 0, un = 0;
    for (uint32_t gw = 0; gw < d.GW; gw++) {
        const uint64_t a = x[gw], b = y[gw];
        in += __popcll(a & b);
        un += __popcll(a | b);
    }
    const uint64_t o = perm ? perm[k] : k;
    inter[o] = in;
    uni[o] = un;
}

// population.rs:753-784 on the accessory matrix: mean Jaccard distance of i to all
// others, summed in ascending j like the reference's fold (:770).
__global__ void acc_average_distance_kernel(const uint64_t *accI, double *out, acc_dims d,
                                            double core_genes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.N) return;
    const uint64_t *x = accI + (uint64_t)i * d.GW;
    double sum = 0.0;
    for (uint32_t j = 0; j < d.N; j++) {
        if (j == i) continue;
        const uint64_t *y = accI + (uint64_t)j * d.GW;
        uint32_t in = 0, un = 0;
        for (uint32_t gw = 0; gw < d.GW; gw++) {
            in += __popcll(x[gw] & y[gw]);
            un += __popcll(x[gw] | y[gw]);
        }
        const double pd = 1.0 - (((double)in + 0.0 + core_genes) / ((double)un + 0.0 + core_genes));
        sum = sum + pd;
    }
    double fd = sum / (double)(d.N - 1u);
    if (fd == 0.0) fd = 2.2250738585072014e-308;   // f64::MIN_POSITIVE, population.rs:774-776
    out[i] = fd;
}

// D-avg in two steps for populations whose N x N distance matrix fits in memory:
// (1) every pair's Jaccard distance, written transposed (the distance is symmetric bit for bit),
// (2) one thread per individual sums its column in ascending j -- the reference's left-to-right
//     fold (population.rs:770) -- with coalesced reads.
__global__ void __launch_bounds__(256) acc_pair_matrix_kernel(const uint64_t *accI, double *Dt, acc_dims d,
                                                              double core_genes)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= d.N) return;
    const uint64_t *x = accI + (uint64_t)i * d.GW, *y = accI + (uint64_t)j * d.GW;
    uint32_t in = 0, un = 0;
    for (uint32_t gw = 0; gw < d.GW; gw++) {
        in += __popcll(x[gw] & y[gw]);
        un += __popcll(x[gw] | y[gw]);
    }
    Dt[(uint64_t)i * d.N + j] = 1.0 - (((double)in + 0.0 + core_genes) / ((double)un + 0.0 + core_genes));
}

__global__ void __launch_bounds__(64) acc_average_from_matrix_kernel(const double *Dt, double *out, acc_dims d)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.N) return;
    double sum = 0.0;
    for (uint32_t j = 0; j < d.N; j++) {
        if (j == i) continue;
        sum = sum + Dt[(uint64_t)j * d.N + i];      // = distance(i, j)
    }
    double fd = sum / (double)(d.N - 1u);
    if (fd == 0.0) fd = 2.2250738585072014e-308;    // f64::MIN_POSITIVE, population.rs:774-776
    out[i] = fd;
}
