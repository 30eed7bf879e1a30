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
    uint32_t *dst32 = (uint32_t *)a.dstI;
    for (uint32_t w = threadIdx.x; w < (r_hi - r_lo) * W32; w += blockDim.x) {
        const uint32_t v = lrow[w];
        if (v) {
            uint32_t *g = dst32 + (uint64_t)r_lo * W32 + w;
            if ((v & ~__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u) atomicOr(g, v);
        }
    }
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
