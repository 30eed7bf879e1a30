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
    const uint64_t *srcI;     // individual-major rows of the parents' generation
    uint64_t *dstI;           // rows of the children
    const uint32_t *idx;      // parents; may alias host-mapped pinned memory (one broadcast read per wave)
    uint32_t *idx_out;        // if set, the word-0 threads publish the parents in device memory
    uint64_t *snapI;          // if set, a second copy of the children: the pre-recombination snapshot of the light HGT form
    uint32_t *zero_word;      // if set, thread 0 zeroes this word (the HGT kernel's work counter: no memset on the chain)
    acc_dims d;
    uint32_t gen, k0, k1;
    ps_acc_plan plan;
};

// 64-bit mask of the genes of row word gw that lie in [gb, ge)
__device__ __forceinline__ uint64_t ps_word_range_mask(uint32_t gw, uint32_t gb, uint32_t ge)
{
    const uint32_t lo = gw * 64u;
    if (ge <= lo || gb >= lo + 64u || gb >= ge) return 0ull;
    uint64_t m = ~0ull;
    if (gb > lo) m &= ~0ull << (gb - lo);
    if (ge < lo + 64u) m &= (1ull << (ge - lo)) - 1ull;
    return m;
}

// Fused parent gather (population.rs:450-465) and gain/loss (population.rs:486-510) on the individual-major
// view: child row i = parent row idx[i] (a contiguous GW-word copy) XOR the flip mask of (i, word).  Thread =
// (individual, row word), words fastest: a wave reads / writes whole rows, and the 16 Philox calls of a word
// (4 genes each, DESIGN.md 3.3: cell (i, g) flips iff word g mod 4 of Philox(g / 4, i, gen, 3) is below the
// compartment's threshold) are independent.  Both compartments' "below threshold" masks are built for every
// gene and selected with the word's compartment masks, so the code has no per-gene compartment tests.
// (Round 2's form gathered one parent WORD of the gene-major view per (individual, gene) -- 8 bytes read per
// bit at N = 65536, 0.63 ms for a 33 MB matrix -- and rebuilt the gene-major view with one ballot per gene;
// that view is now rebuilt only when someone asks for gene frequencies.)
template <bool DO_GATHER, bool DO_MUT>
__global__ void __launch_bounds__(256) acc_step_rows_kernel(acc_step_args a)
{
    const acc_dims d = a.d;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && a.zero_word) *a.zero_word = 0u;
    if (t >= (uint64_t)d.N * d.GW) return;
    const uint32_t i = (uint32_t)(t / d.GW), gw = (uint32_t)(t % d.GW);
    const uint32_t p = DO_GATHER ? a.idx[i] : i;
    if (DO_GATHER && a.idx_out && gw == 0) a.idx_out[i] = p;
    uint64_t word = a.srcI[(uint64_t)p * d.GW + gw];
    if (DO_MUT) {
        const uint32_t t0 = a.plan.n_comp > 0 ? a.plan.flip_thr[0] : 0u, t1 = a.plan.n_comp > 1 ? a.plan.flip_thr[1] : 0u;
        const uint64_t m0 = a.plan.n_comp > 0 ? ps_word_range_mask(gw, a.plan.comp_begin[0], a.plan.comp_end[0]) : 0ull;
        // (a gene in both ranges takes the LAST compartment's threshold, as the per-gene loop of round 2 did)
        const uint64_t m1 = a.plan.n_comp > 1 ? ps_word_range_mask(gw, a.plan.comp_begin[1], a.plan.comp_end[1]) : 0ull;
        uint32_t lt0[2] = { 0u, 0u }, lt1[2] = { 0u, 0u };
#pragma unroll
        for (uint32_t j = 0; j < 16u; j++) {
            const ps_u4 r = ps_philox(gw * 16u + j, i, a.gen, PS_STREAM_ACC_MUT, a.k0, a.k1);
            const uint32_t w[4] = { r.x, r.y, r.z, r.w };
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++) {
                const uint32_t b = 4u * j + k;
                lt0[b >> 5] |= (w[k] < t0 ? 1u : 0u) << (b & 31u);          // population.rs:505
                lt1[b >> 5] |= (w[k] < t1 ? 1u : 0u) << (b & 31u);
            }
        }
        const uint64_t f0 = ((uint64_t)lt0[1] << 32) | lt0[0], f1 = ((uint64_t)lt1[1] << 32) | lt1[0];
        word ^= (f1 & m1) | (f0 & m0 & ~m1);
    }
    a.dstI[t] = word;
    if (a.snapI) a.snapI[t] = word;
}

// ---------------------------------------------------------------------------
// HGT (population.rs:544-751, accessory path), keyed per donor like the reference draws it
// (DESIGN.md 3.4): donor d of compartment c sends k_d ~ Poisson(lambda_c) events (:599; an integer
// threshold table built on the host, word 0 of Philox(d, 0, gen, HGT_COUNT | c << 8)); event j of
// donor d is Philox(j, d, gen, HGT | c << 8): recipient uniform over the others (word 1, :616-619),
// gene = the mulhi(word 2, n)-th present gene of the donor inside the compartment IN THE SNAPSHOT
// (:636-680; nothing if n = 0, :672); the recipient gains the gene (value always 1, :632 -- an
// idempotent OR, so the order of events does not matter).
//
// Work item = (compartment, donor).  A workgroup copies the donor's present genes of the
// compartment into an LDS list once (wave prefix sum of the row's popcounts), so the per-event
// gene lookup is an LDS read; the events of the item are spread over the workgroup's threads.
// Only the individual-major view is edited; acc_i_to_g_kernel rebuilds the gene-major view.
// ---------------------------------------------------------------------------
struct acc_hgt_args {
    const uint64_t *srcI;          // donor rows: the pre-recombination snapshot
    uint64_t *dstI;                // recipient rows (atomic form)
    acc_dims d;
    uint32_t n_comp;
    uint32_t gb[PS_MAX_COMP], ge[PS_MAX_COMP];
    const uint32_t *ptab[PS_MAX_COMP];   // Poisson thresholds of the compartment (null: no events)
    uint32_t kmin[PS_MAX_COMP], plen[PS_MAX_COMP];
    uint32_t *work_ctr;            // dynamic item counter (zeroed before the launch)
    uint32_t gen, k0, k1;
    // binned form (heavy HGT)
    uint32_t *bins, *counts;       // bins[(donor workgroup * parts + part) * bin_cap], counts[donor workgroup * parts + part]
    uint32_t bin_cap;
    uint32_t parts, rows_per_part;
    uint32_t part_magic;           // floor(2^32 / rows_per_part) + 1: rc / rows_per_part = mulhi(rc, magic) for rc < 2^16
    uint32_t *scratch;             // slice images [n_slices][N][2*GW]
    unsigned long long *ovf_img;   // binned form: one more image, all zero between launches, for the events of a full bin
    uint32_t *overflow_flag;
    uint16_t *list_scratch;        // null: donor lists in LDS; else [grid][list_stride] in global memory
    uint32_t list_stride;
    // donor shard: this launch serves the donors [dn_lo, dn_lo + dn_cnt) only (all of them: 0, N).  Events are keyed per
    // donor and the recipient's bit is ORed, so the union over any partition of the donors is the unsharded result.
    uint32_t dn_lo, dn_cnt;
    uint32_t bin_prio;             // bin pass: wave priority (s_setprio) while it runs beside a sweep (0 = leave it)
};

// k_d for every (compartment, donor): kmin + number of thresholds <= u (ps_poisson_table).
// The light form also takes its snapshot copy of the matrix here (copy_src -> copy_dst, grid-stride)
// instead of a separate blit beside the sweep.
// k_d of one (compartment, donor): the threshold table searched with word 0 of Philox(d, 0, gen, HGT_COUNT | c << 8)
__device__ __forceinline__ uint32_t ps_hgt_count(const acc_hgt_args &a, uint32_t c, uint32_t dn)
{
    if (!a.ptab[c]) return 0u;
    const ps_u4 r = ps_philox(dn, 0u, a.gen, PS_STREAM_HGT_COUNT | (c << 8), a.k0, a.k1);
    uint32_t lo = 0, hi = a.plen[c];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.ptab[c][mid] <= r.x) lo = mid + 1u; else hi = mid;
    }
    if (lo >= a.plen[c]) lo = a.plen[c] - 1u;
    return a.kmin[c] + lo;
}

// the light form's snapshot copy of the matrix when the step before it did not leave one (stand-alone ps_recombine calls)
__global__ void __launch_bounds__(256) acc_snapshot_kernel(const uint64_t *copy_src, uint64_t *copy_dst, uint64_t words)
{
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x)
        copy_dst[w] = copy_src[w];
}

// the present genes of row `row` inside [gb, ge), ascending, into list[]; returns their number
// (called by one whole wave; lane = row word, wave prefix sum of the popcounts)
__device__ __forceinline__ uint32_t ps_wave_gene_list(const uint64_t *row, uint32_t GW, uint32_t gb, uint32_t ge,
                                                      uint16_t *list, uint32_t lane)
{
    uint32_t base = 0;
    for (uint32_t gw0 = gb >> 6; gw0 * 64u < ge; gw0 += 64u) {
        const uint32_t gw = gw0 + lane;
        uint64_t word = 0;
        if (gw * 64u < ge && gw < GW) {
            word = row[gw];
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
            list[pos++] = (uint16_t)(gw * 64u + b);
        }
        base += __shfl(incl, 63, 64);
    }
    return base;
}

// Light HGT: one 64-bit atomicOr per event.  One wave per workgroup and no static LDS, so that the
// kernel fits beside whatever the core sweep leaves on a CU; the donor's gene list lives in dynamic
// LDS or -- when the co-running block sweep owns the CU's LDS (cfg4 population) -- in the
// workgroup's slice of a global scratch that stays hot in L2.  srcI must be a snapshot copy of the
// matrix, dstI the live one.  Launched narrow by ps_sim.
__global__ void __launch_bounds__(64) acc_hgt_donor_wave_kernel(acc_hgt_args a)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t lds_list[];
    uint16_t *glist = a.list_scratch ? a.list_scratch + (uint64_t)blockIdx.x * a.list_stride : lds_list;
    const acc_dims d = a.d;
    const uint32_t lane = threadIdx.x;
    const uint32_t items = a.n_comp * a.dn_cnt;
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(a.work_ctr, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= items) break;
        const uint32_t c = item / a.dn_cnt, dn = a.dn_lo + item % a.dn_cnt;
        const uint32_t k = ps_hgt_count(a, c, dn);
        if (k == 0u) continue;
        __threadfence_block();                             // the previous item's list reads are done
        const uint32_t n = ps_wave_gene_list(a.srcI + (uint64_t)dn * d.GW, d.GW, a.gb[c], a.ge[c], glist, lane);
        __threadfence_block();                             // the list is written (it may live in global memory)
        if (n == 0u) continue;                             // population.rs:672
        const uint32_t stream = PS_STREAM_HGT | (c << 8);
        // one Philox block serves two events: (x, y) = (recipient, gene) words of event 2 jp, (z, w) of event 2 jp + 1
        for (uint32_t jp = lane; 2u * jp < k; jp += 64u) {
            const ps_u4 r = ps_philox(jp, dn, a.gen, stream, a.k0, a.k1);
#pragma unroll
            for (uint32_t h = 0; h < 2u; h++) {
                if (2u * jp + h >= k) break;
                uint32_t rc = ps_mulhi(h ? r.z : r.x, d.N - 1u);
                rc += (rc >= dn) ? 1u : 0u;                             // population.rs:618
                const uint32_t gene = glist[ps_mulhi(h ? r.w : r.y, n)];
                atomicOr((unsigned long long *)&a.dstI[(uint64_t)rc * d.GW + (gene >> 6)], 1ull << (gene & 63u));
            }
        }
    }
}

// Heavy HGT (>= 1e7 expected events per generation), two passes without global atomics.  The
// recipients are split into partitions whose rows fit an LDS image (cfg3: 4 partitions, cfg4: 202,
// up to 1024).  First pass: workgroup b serves the items b, b + grid, ... and appends every event,
// packed (recipient row inside its partition << 16 | gene), to ITS bin of the recipient's partition:
// bins[(b * parts + part) * cap ...]; the position comes from a per-lane LDS atomic on the
// workgroup's fill counters (parts * 4 bytes of LDS).  (Wave-aggregated appends -- one ballot per
// partition -- were slower than one LDS atomic per lane even with 4 partitions.)
#ifndef PS_HGT_BIN_BLOCKS
#define PS_HGT_BIN_BLOCKS 2u
#endif
__global__ void __launch_bounds__(256) acc_hgt_donor_bin_kernel(acc_hgt_args a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sc_lds[];
    uint32_t *fill = sc_lds;                                           // [parts]
    // present genes of the item's donor: in LDS, or -- beside a block sweep that owns the CU's LDS -- in the
    // workgroup's slice of a global scratch (hot in L2, as in the light kernel)
    uint16_t *glist = a.list_scratch ? a.list_scratch + (uint64_t)blockIdx.x * a.list_stride
                                     : (uint16_t *)(sc_lds + ((a.parts + 3u) & ~3u));
    __shared__ uint32_t sh_n;
    const acc_dims d = a.d;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t items = a.n_comp * a.dn_cnt;
    const uint32_t cap = a.bin_cap;
    // (cfg3: the pass is the longest link of the chain that has to fit beside the sweep -- it has one wave slot per SIMD
    // against the sweep's seven, and what it loses in issue slots the whole next generation waits for)
    if (a.bin_prio >= 3u) __builtin_amdgcn_s_setprio(3);
    else if (a.bin_prio == 2u) __builtin_amdgcn_s_setprio(2);
    else if (a.bin_prio == 1u) __builtin_amdgcn_s_setprio(1);
    uint32_t *mybins = a.bins + (uint64_t)blockIdx.x * a.parts * cap;
    for (uint32_t q = tid; q < a.parts; q += blockDim.x) fill[q] = 0u;
    for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
        __syncthreads();                                   // previous list is no longer read; fill[] is zeroed
        const uint32_t c = item / a.dn_cnt, dn = a.dn_lo + item % a.dn_cnt;
        const uint32_t k = ps_hgt_count(a, c, dn);      // (every thread draws the same count: no separate counts kernel on the chain)
        if (k == 0u) continue;
        if (tid < 64u) {
            const uint32_t m = ps_wave_gene_list(a.srcI + (uint64_t)dn * d.GW, d.GW, a.gb[c], a.ge[c], glist, lane);
            if (tid == 0) sh_n = m;
            __threadfence_block();                         // (the list may live in global memory)
        }
        __syncthreads();
        const uint32_t n = sh_n;
        if (n == 0u) continue;                             // population.rs:672
        const uint32_t stream = PS_STREAM_HGT | (c << 8);
        // one Philox block serves two events: (x, y) = (recipient, gene) words of event 2 jp, (z, w) of event 2 jp + 1
        // (beside a sweep the pass has ONE wave per SIMD: nothing hides the chain Philox -> list read -> LDS atomic -> store
        // of an event but the thread's own next events, so two blocks = four events are taken per trip: all four list reads,
        // then all four appends)
        auto place = [&](uint32_t rc, uint32_t gene, uint32_t part, uint32_t pos) {
            if (pos < cap) mybins[(uint64_t)part * cap + pos] = ((rc - part * a.rows_per_part) << 16) | gene;
            // a full bin (sized for mean + 10 sigma) drops nothing: the event goes to the overflow image, which the
            // reduce pass ORs in like one more slice image and clears again
            else atomicOr(&a.ovf_img[(uint64_t)rc * d.GW + (gene >> 6)], 1ull << (gene & 63u));
        };
        constexpr uint32_t NBK = PS_HGT_BIN_BLOCKS;          // Philox blocks (= pairs of events) per trip
        for (uint32_t jp = tid; 2u * jp < k; jp += NBK * blockDim.x) {
            uint32_t wr[2u * NBK], wg[2u * NBK];
            bool ok[2u * NBK];
#pragma unroll
            for (uint32_t b = 0; b < NBK; b++) {
                const uint32_t jb = jp + b * blockDim.x;
                const ps_u4 r = ps_philox(jb, dn, a.gen, stream, a.k0, a.k1);
                wr[2u * b] = r.x; wg[2u * b] = r.y; wr[2u * b + 1u] = r.z; wg[2u * b + 1u] = r.w;
                ok[2u * b] = 2u * jb < k;
                ok[2u * b + 1u] = 2u * jb + 1u < k;
            }
            uint32_t rc[2u * NBK], gene[2u * NBK], part[2u * NBK], pos[2u * NBK];
#pragma unroll
            for (uint32_t e = 0; e < 2u * NBK; e++) {
                rc[e] = ps_mulhi(wr[e], d.N - 1u);
                rc[e] += (rc[e] >= dn) ? 1u : 0u;                       // population.rs:618
                gene[e] = glist[ps_mulhi(wg[e], n)];
                part[e] = ps_mulhi(rc[e], a.part_magic);                // rc / rows_per_part
            }
#pragma unroll
            for (uint32_t e = 0; e < 2u * NBK; e++) pos[e] = ok[e] ? atomicAdd(&fill[part[e]], 1u) : 0u;
#pragma unroll
            for (uint32_t e = 0; e < 2u * NBK; e++)
                if (ok[e]) place(rc[e], gene[e], part[e], pos[e]);
        }
    }
    __syncthreads();
    for (uint32_t q = tid; q < a.parts; q += blockDim.x) {
        a.counts[(uint64_t)blockIdx.x * a.parts + q] = min(fill[q], cap);
    }
}

// second pass: workgroup (part, slice) ORs bin `part` of the donor workgroups slice, slice +
// n_slices, ... into an LDS image of the partition's rows (the four 256-thread quarters walk
// different bins, four loads in flight per thread) and publishes the image with plain coalesced
// stores; acc_hgt_reduce_kernel then ORs the slice images into the matrix (merging with atomics
// took longer than the events themselves)
__global__ void __launch_bounds__(1024) acc_hgt_apply_kernel(acc_hgt_args a, uint32_t donor_blocks, uint32_t n_slices)
{
    extern __shared__ uint32_t lrow[];     // [rows_per_part][2*GW] 32-bit words
    const acc_dims d = a.d;
    const uint32_t part = blockIdx.x / n_slices, slice = blockIdx.x % n_slices;
    const uint32_t r_lo = part * a.rows_per_part, r_hi = min(d.N, r_lo + a.rows_per_part);
    const uint32_t W32 = 2u * d.GW;
    for (uint32_t w = threadIdx.x; w < a.rows_per_part * W32; w += blockDim.x) lrow[w] = 0u;
    __syncthreads();
    const uint32_t cap = a.bin_cap;
    // a WAVE walks its own bins (round 5; the four 256-thread quarters of round 3 walked four): a bin is a dependent pair of
    // reads (its count, then its events), and with the donors sharded over the ranks a bin holds ~120 events -- the pass was
    // a chain of 51 such pairs per quarter, 58 us per workgroup, 0.23 ms between two sweeps at one rank of 8.  Sixteen bins
    // in flight per workgroup instead of four; the next bin's count is requested before this bin's events are applied.
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, nw = blockDim.x >> 6;
    uint32_t bb = slice + wave * n_slices;
    uint32_t n = bb < donor_blocks ? a.counts[(uint64_t)bb * a.parts + part] : 0u;
    for (; bb < donor_blocks; bb += nw * n_slices) {
        const uint32_t bn = bb + nw * n_slices;
        const uint32_t n_next = bn < donor_blocks ? a.counts[(uint64_t)bn * a.parts + part] : 0u;
        const uint32_t *src = a.bins + ((uint64_t)bb * a.parts + part) * cap;
        for (uint32_t k = lane; k < n; k += 256u) {
            uint32_t v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) v[u] = (k + 64u * u < n) ? src[k + 64u * u] : 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                if (k + 64u * u < n) {
                    const uint32_t gene = v[u] & 0xFFFFu;
                    atomicOr(&lrow[(v[u] >> 16) * W32 + (gene >> 5)], 1u << (gene & 31u));
                }
            }
        }
        n = n_next;
    }
    __syncthreads();
    uint32_t *img = a.scratch + ((uint64_t)slice * d.N + r_lo) * W32;
    for (uint32_t w = threadIdx.x; w < (r_hi - r_lo) * W32; w += blockDim.x) img[w] = lrow[w];
}

// dstI |= OR over the slice images written by acc_hgt_apply_kernel (scratch[slice][N][GW] u64); `assign`: dstI = that
// OR (the delta buffer of a donor-sharded run, merged into the matrix after the exchange)
__global__ void __launch_bounds__(256) acc_hgt_reduce_kernel(const uint64_t *scratch, uint64_t *dstI, uint64_t words,
                                                             uint32_t n_slices, int assign, unsigned long long *ovf_img)
{
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    uint64_t v = ovf_img[w];            // events of full bins (all zero unless a bin overflowed); left zero for the next launch
    if (v) ovf_img[w] = 0ull;
    for (uint32_t sl = 0; sl < n_slices; sl++) v |= scratch[(uint64_t)sl * words + w];
    if (assign) dstI[w] = v;
    else if (v) dstI[w] |= v;
}

// The same reduce pass with one wave per individual (lanes over the row words), which also leaves the row's gene count --
// the device half of the NEXT generation's sample_indices under neutral selection (population.rs:282-291; log_sum = 0.0)
// -- so that the chain of a generation has one launch fewer before its host half (ps_sim: cfg3's heavy HGT)
__global__ void __launch_bounds__(256) acc_hgt_reduce_rows_kernel(const uint64_t *scratch, uint64_t *dstI, acc_dims d, uint32_t n_slices,
                                                                  unsigned long long *ovf_img, int32_t *num_genes, double *logw)
{
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (i >= d.N) return;
    const uint64_t words = (uint64_t)d.N * d.GW;
    uint32_t n = 0;
    for (uint32_t gw = lane; gw < d.GW; gw += 64u) {
        const uint64_t w = (uint64_t)i * d.GW + gw;
        uint64_t v = ovf_img[w];
        if (v) ovf_img[w] = 0ull;
        for (uint32_t sl = 0; sl < n_slices; sl++) v |= scratch[(uint64_t)sl * words + w];
        const uint64_t old = dstI[w], now = old | v;
        if (now != old) dstI[w] = now;
        n += __popcll(now);
    }
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
    if (lane == 0) { num_genes[i] = (int32_t)n; logw[i] = 0.0; }
}

// The merge of a donor-sharded HGT (matrix |= exchanged delta) with one wave per individual, which also leaves the row's gene
// count -- the device half of the NEXT generation's sample_indices under neutral selection (population.rs:282-291; log_sum =
// 0.0), like acc_hgt_reduce_rows_kernel in an unsharded run: the count kernel of the next host half (a full-chip launch that
// ran between two sweeps: 36 us at N = 65536) goes away and the host half starts that much earlier
__global__ void __launch_bounds__(256) acc_or_rows_kernel(uint64_t *dstI, const uint64_t *delta, acc_dims d, int32_t *num_genes, double *logw)
{
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (i >= d.N) return;
    uint32_t n = 0;
    for (uint32_t gw = lane; gw < d.GW; gw += 64u) {
        const uint64_t w = (uint64_t)i * d.GW + gw;
        const uint64_t old = dstI[w], now = old | delta[w];
        if (now != old) dstI[w] = now;
        n += __popcll(now);
    }
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
    if (lane == 0) { num_genes[i] = (int32_t)n; logw[i] = 0.0; }
}

// dst |= src (a donor shard's exchanged HGT delta into the matrix; a peer shard's delta into this shard's)
__global__ void __launch_bounds__(256) acc_or_kernel(uint64_t *dst, const uint64_t *src, uint64_t words)
{
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    const uint64_t v = src[w];
    if (v) dst[w] |= v;
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

// P-draw on the device (population.rs:440-443, WeightedIndex::sample): parent k = number of cumulative
// weights <= x_k over the first N-1 of them, x_k = f64_k * total with the k-th f64 of the seeded PARENTS
// stream.  The cumulative table is built by the host (sequential f64 sums, libm softmaxes); the draw itself
// is one IEEE multiplication and comparisons, so it equals the host's bit for bit.  ps_sim stores the children of a
// generation in ascending parent order, so the draws are counted per parent here (cnt zeroed by the
// caller) and laid out by idx_scan_kernel / idx_fill_kernel: a counting sort.
__global__ void __launch_bounds__(256) acc_draw_parents_kernel(const double *cum, double total, uint32_t N,
                                                               uint32_t k0, uint32_t k1, uint32_t gen, uint32_t *cnt, uint32_t *draws)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const double x = ps_hs_f64(k0, k1, PS_STREAM_PARENTS, gen, k) * total;
    uint32_t lo = 0, hi = N - 1u;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (cum[mid] <= x) lo = mid + 1u; else hi = mid;
    }
    draws[k] = lo;          // (draw order, host-mapped: what the outputs' row order is derived from)
    atomicAdd(&cnt[lo], 1u);
}

// cnt[p] (children of parent p) -> inclusive prefix sums, in place, in three small launches of 256-thread workgroups (4 waves:
// they fit the one wave slot per SIMD the sweeps leave, where a 1024-thread workgroup waited for the sweep to end; one
// workgroup walking all tiles took 300 us beside the sweep at N = 65536): the totals of the 1024-parent tiles, their
// exclusive prefix sums (one workgroup), and the scan of every tile started from its prefix.
__global__ void __launch_bounds__(256) idx_tile_sums_kernel(const uint32_t *cnt, uint32_t N, uint32_t *tsum)
{
    __shared__ uint32_t wtot[4];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t p0 = blockIdx.x * 1024u + 4u * t;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t u = 0; u < 4u; u++) v += (p0 + u < N) ? cnt[p0 + u] : 0u;
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) wtot[wave] = v;
    __syncthreads();
    if (t == 0) tsum[blockIdx.x] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
}

// tsum[0 .. n) -> exclusive prefix sums, in place; one workgroup, n <= 256 * 64
__global__ void __launch_bounds__(256) idx_tile_prefix_kernel(uint32_t *tsum, uint32_t n)
{
    __shared__ uint32_t part[256];
    const uint32_t t = threadIdx.x, E = (n + 255u) / 256u;
    const uint32_t b = t * E, e = min(n, b + E);
    uint32_t sum = 0;
    for (uint32_t k = b; k < e; k++) sum += tsum[k];
    part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 256u; off <<= 1) {
        const uint32_t v = (t >= off) ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t k = b; k < e; k++) { const uint32_t x = tsum[k]; tsum[k] = run; run += x; }
}

__global__ void __launch_bounds__(256) idx_scan_kernel(uint32_t *cnt, uint32_t N, const uint32_t *tprefix)
{
    __shared__ uint32_t wtot[4];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t p0 = blockIdx.x * 1024u + 4u * t;
    uint32_t v[4];
#pragma unroll
    for (uint32_t u = 0; u < 4u; u++) v[u] = (p0 + u < N) ? cnt[p0 + u] : 0u;
    v[1] += v[0];
    v[2] += v[1];
    v[3] += v[2];
    uint32_t incl = v[3];                     // inclusive wave prefix sum of the threads' totals
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += o;
    }
    if (lane == 63u) wtot[wave] = incl;
    __syncthreads();
    uint32_t before = tprefix[blockIdx.x] + incl - v[3];
    for (uint32_t w = 0; w < wave; w++) before += wtot[w];
#pragma unroll
    for (uint32_t u = 0; u < 4u; u++)
        if (p0 + u < N) cnt[p0 + u] = before + v[u];
}

// zeroes the counts for the next generation's draws (instead of a memset: the runtime's fill kernel did not fit beside the
// window sweep and waited 2.8 ms for it to end, with the whole chain of the next generation behind it)
__global__ void __launch_bounds__(256) idx_zero_kernel(uint32_t *cnt, uint32_t N)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < N) cnt[k] = 0u;
}

// child k belongs to the first parent whose inclusive prefix exceeds k; written to device memory (the gather kernels)
// and to host-mapped memory (ps_sim_last_parents)
__global__ void __launch_bounds__(256) idx_fill_kernel(const uint32_t *incl, uint32_t N, uint32_t *idx_dev, uint32_t *idx_host)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    uint32_t lo = 0, hi = N - 1u;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (incl[mid] <= k) lo = mid + 1u; else hi = mid;
    }
    idx_dev[k] = lo;
    idx_host[k] = lo;
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

// distances.rs:55-77 numerators for sampled pairs: 8 lanes per pair, each reading every 8th word of the two rows (the 8
// lanes of a pair read 64 contiguous bytes per load -- one line instead of eight), partial counts summed across the group
__global__ void __launch_bounds__(256) acc_pair_counts_kernel(const uint64_t *accI, const uint32_t *r1, const uint32_t *r2,
                                                              const uint32_t *perm, uint64_t P, uint32_t *inter, uint32_t *uni,
                                                              acc_dims d)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t k = t >> 3;
    const uint32_t sub = (uint32_t)t & 7u;
    uint32_t in = 0, un = 0;
    if (k < P) {
        const uint64_t *x = accI + (uint64_t)r1[k] * d.GW, *y = accI + (uint64_t)r2[k] * d.GW;
        for (uint32_t gw = sub; gw < d.GW; gw += 8u) {
            const uint64_t a = x[gw], b = y[gw];
            in += __popcll(a & b);
            un += __popcll(a | b);
        }
    }
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1) {        // (whole waves take part: k >= P lanes carry zeros)
        in += __shfl_xor(in, off);
        un += __shfl_xor(un, off);
    }
    if (k < P && sub == 0u) {
        const uint64_t o = perm ? perm[k] : k;
        inter[o] = in;
        uni[o] = un;
    }
}

// D-avg, population.rs:753-784 on the accessory matrix: mean Jaccard distance of i to all others, summed in ascending j
// like the reference's fold (:770).
// D-avg in two steps for populations whose N x N distance matrix fits in memory:
// (1) every pair's Jaccard distance, written transposed (the distance is symmetric bit for bit),
// (2) one thread per individual sums its column in ascending j -- the reference's left-to-right
//     fold (population.rs:770) -- with coalesced reads.
// The matrix comes from LDS tiles: a 256-thread workgroup owns 64 x 64 pairs (only tiles on or above the
// diagonal; the distance is symmetric bit for bit, so both (i, j) and (j, i) are written), stages 16 row words
// of the 64 + 64 individuals at a time, and every thread counts its 4 x 4 pairs from 4 + 4 LDS reads per word
// (the plain kernel re-reads a 504-byte row per pair from L2: 0.15 ms alone, 0.31 ms beside the sweep at N = 1000).
#define PS_PM_CH 16u
// T x T pairs per 256-thread workgroup, R x R per thread: 64 / 4, or 32 / 2 -- four times the workgroups -- for populations
// whose 64 x 64 tiles are fewer than the CUs (N = 1000: 136 tiles)
template <uint32_t T, uint32_t R>
__global__ void __launch_bounds__(256, 8) acc_pair_matrix_tiled_kernel(const uint64_t *accI, double *Dt, acc_dims d, double core_genes)
{
    constexpr uint32_t Q = T / R, NT = Q * Q;
    static_assert(NT == 256u, "256 threads: (T / R)^2");
    // (at most 64 VGPRs -- launch bounds -- because 7 sweep waves of 64 VGPRs leave a SIMD exactly that: with 90 the kernel
    // could not start before the sweep's waves retired, whatever its priority)
    // D-avg sits on the critical chain of a --competition_strength generation (HGT -> D-avg -> host half -> parents): its waves
    // take VALU issue before the sweep's on their SIMD
    __builtin_amdgcn_s_setprio(3);
    __shared__ uint64_t TA[T * (PS_PM_CH + 1u)], TB[T * (PS_PM_CH + 1u)];
    const uint32_t bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const uint32_t tid = threadIdx.x, tx = tid % Q, ty = tid / Q;
    const uint32_t i0 = bi * T, j0 = bj * T;
    uint32_t in[R][R], cx[R], cy[R];
#pragma unroll
    for (uint32_t a = 0; a < R; a++) {
        cx[a] = cy[a] = 0;
#pragma unroll
        for (uint32_t b = 0; b < R; b++) in[a][b] = 0;
    }
    for (uint32_t g0 = 0; g0 < d.GW; g0 += PS_PM_CH) {
        __syncthreads();
        for (uint32_t t = tid; t < T * PS_PM_CH; t += NT) {
            const uint32_t r = t / PS_PM_CH, c = t % PS_PM_CH;
            const bool okc = g0 + c < d.GW;
            TA[r * (PS_PM_CH + 1u) + c] = (okc && i0 + r < d.N) ? accI[(uint64_t)(i0 + r) * d.GW + g0 + c] : 0ull;
            TB[r * (PS_PM_CH + 1u) + c] = (okc && j0 + r < d.N) ? accI[(uint64_t)(j0 + r) * d.GW + g0 + c] : 0ull;
        }
        __syncthreads();
#pragma unroll 2
        for (uint32_t c = 0; c < PS_PM_CH; c++) {
            uint64_t x[R], y[R];
#pragma unroll
            for (uint32_t a = 0; a < R; a++) x[a] = TA[(ty + Q * a) * (PS_PM_CH + 1u) + c];
#pragma unroll
            for (uint32_t b = 0; b < R; b++) y[b] = TB[(tx + Q * b) * (PS_PM_CH + 1u) + c];
            // |x u y| = |x| + |y| - |x n y|: only the intersections are counted per pair
#pragma unroll
            for (uint32_t a = 0; a < R; a++) cx[a] += __popcll(x[a]);
#pragma unroll
            for (uint32_t b = 0; b < R; b++) cy[b] += __popcll(y[b]);
#pragma unroll
            for (uint32_t a = 0; a < R; a++)
#pragma unroll
                for (uint32_t b = 0; b < R; b++) in[a][b] += __popcll(x[a] & y[b]);
        }
    }
#pragma unroll
    for (uint32_t a = 0; a < R; a++)
#pragma unroll
        for (uint32_t b = 0; b < R; b++) {
            const uint32_t i = i0 + ty + Q * a, j = j0 + tx + Q * b;
            if (i < d.N && j < d.N) {
                const double pd = 1.0 - (((double)in[a][b] + 0.0 + core_genes) / ((double)(cx[a] + cy[b] - in[a][b]) + 0.0 + core_genes));
                Dt[(uint64_t)i * d.N + j] = pd;
                if (bi != bj) Dt[(uint64_t)j * d.N + i] = pd;
            }
        }
}

// All-pairs Jaccard numerators for P > ~N^2 / 4 sampled pairs (cfg5: one thread per sampled pair re-read two 504-byte
// rows per pair -- 34 GB of L2 traffic at P = 2^25, and slowed the core kernels beside it): the intersections of every
// pair of a 64 x 64 tile from LDS tiles (as acc_pair_matrix_tiled_kernel, upper triangle only, both orders written) and
// the row counts; the lookup takes |x u y| = |x| + |y| - |x n y| (distances.rs:55-77 counts both directly).
__global__ void __launch_bounds__(256) acc_pair_inter_tiled_kernel(const uint64_t *accI, uint32_t *In, uint32_t *rowcnt, acc_dims d)
{
    __shared__ uint64_t TA[64u * (PS_PM_CH + 1u)], TB[64u * (PS_PM_CH + 1u)];
    const uint32_t bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const uint32_t tid = threadIdx.x, tx = tid & 15u, ty = tid >> 4;
    const uint32_t i0 = bi * 64u, j0 = bj * 64u;
    uint32_t in[4][4], cx[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) in[a][b] = 0;
    for (uint32_t g0 = 0; g0 < d.GW; g0 += PS_PM_CH) {
        __syncthreads();
        for (uint32_t t = tid; t < 64u * PS_PM_CH; t += 256u) {
            const uint32_t r = t / PS_PM_CH, c = t % PS_PM_CH;
            const bool okc = g0 + c < d.GW;
            TA[r * (PS_PM_CH + 1u) + c] = (okc && i0 + r < d.N) ? accI[(uint64_t)(i0 + r) * d.GW + g0 + c] : 0ull;
            TB[r * (PS_PM_CH + 1u) + c] = (okc && j0 + r < d.N) ? accI[(uint64_t)(j0 + r) * d.GW + g0 + c] : 0ull;
        }
        __syncthreads();
#pragma unroll 4
        for (uint32_t c = 0; c < PS_PM_CH; c++) {
            uint64_t x[4], y[4];
#pragma unroll
            for (int a = 0; a < 4; a++) x[a] = TA[(ty + 16u * a) * (PS_PM_CH + 1u) + c];
#pragma unroll
            for (int b = 0; b < 4; b++) y[b] = TB[(tx + 16u * b) * (PS_PM_CH + 1u) + c];
#pragma unroll
            for (int a = 0; a < 4; a++) cx[a] += __popcll(x[a]);
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) in[a][b] += __popcll(x[a] & y[b]);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const uint32_t i = i0 + ty + 16u * a;
        if (bi == bj && tx == 0 && i < d.N) rowcnt[i] = cx[a];          // (every row block has its diagonal tile)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t j = j0 + tx + 16u * b;
            if (i < d.N && j < d.N) {
                In[(uint64_t)i * d.N + j] = in[a][b];
                if (bi != bj) In[(uint64_t)j * d.N + i] = in[a][b];
            }
        }
    }
}

__global__ void __launch_bounds__(256) acc_pair_lookup_kernel(const uint32_t *In, const uint32_t *rowcnt, uint32_t N, const uint32_t *r1,
                                                              const uint32_t *r2, const uint32_t *perm, uint64_t P, uint32_t *inter,
                                                              uint32_t *uni)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const uint32_t i = r1[k], j = r2[k];
    const uint32_t in = In[(uint64_t)i * N + j];
    const uint64_t o = perm ? perm[k] : k;
    inter[o] = in;
    uni[o] = rowcnt[i] + rowcnt[j] - in;
}

// D-avg for populations whose N x N matrix of doubles is too large to keep (N > 8192): workgroup = 64 individuals i,
// all j in ascending tiles of 64.  Per tile the 64 x 64 Jaccard distances are counted from LDS tiles as in
// acc_pair_matrix_tiled_kernel and parked in LDS; then one thread per i adds its 64 values in ascending j -- the
// reference's left-to-right f64 fold (population.rs:770) is kept, only the counting is parallel.  (Round 2's
// one-thread-per-individual kernel re-read all N rows per individual from L2.)
__global__ void __launch_bounds__(256) acc_average_distance_tiled_kernel(const uint64_t *accI, double *out, acc_dims d,
                                                                         double core_genes)
{
    __shared__ uint64_t TA[64u * (PS_PM_CH + 1u)], TB[64u * (PS_PM_CH + 1u)];
    __shared__ double D[64u * 65u];
    const uint32_t tid = threadIdx.x, tx = tid & 15u, ty = tid >> 4;
    const uint32_t i0 = blockIdx.x * 64u;
    double sum = 0.0;                                   // threads 0..63: running sum of individual i0 + tid
    for (uint32_t j0 = 0; j0 < d.N; j0 += 64u) {
        uint32_t in[4][4], cx[4] = { 0, 0, 0, 0 }, cy[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) in[a][b] = 0;
        for (uint32_t g0 = 0; g0 < d.GW; g0 += PS_PM_CH) {
            __syncthreads();
            for (uint32_t t = tid; t < 64u * PS_PM_CH; t += 256u) {
                const uint32_t r = t / PS_PM_CH, c = t % PS_PM_CH;
                const bool okc = g0 + c < d.GW;
                TA[r * (PS_PM_CH + 1u) + c] = (okc && i0 + r < d.N) ? accI[(uint64_t)(i0 + r) * d.GW + g0 + c] : 0ull;
                TB[r * (PS_PM_CH + 1u) + c] = (okc && j0 + r < d.N) ? accI[(uint64_t)(j0 + r) * d.GW + g0 + c] : 0ull;
            }
            __syncthreads();
#pragma unroll 4
            for (uint32_t c = 0; c < PS_PM_CH; c++) {
                uint64_t x[4], y[4];
#pragma unroll
                for (int a = 0; a < 4; a++) x[a] = TA[(ty + 16u * a) * (PS_PM_CH + 1u) + c];
#pragma unroll
                for (int b = 0; b < 4; b++) y[b] = TB[(tx + 16u * b) * (PS_PM_CH + 1u) + c];
                // |x u y| = |x| + |y| - |x n y|: only the intersections are counted per pair
#pragma unroll
                for (int a = 0; a < 4; a++) cx[a] += __popcll(x[a]);
#pragma unroll
                for (int b = 0; b < 4; b++) cy[b] += __popcll(y[b]);
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) in[a][b] += __popcll(x[a] & y[b]);
            }
        }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++)
                D[(ty + 16u * a) * 65u + tx + 16u * b] = 1.0 - (((double)in[a][b] + 0.0 + core_genes) / ((double)(cx[a] + cy[b] - in[a][b]) + 0.0 + core_genes));
        __syncthreads();
        if (tid < 64u) {
            const uint32_t i = i0 + tid, jn = min(64u, d.N - j0);
            for (uint32_t j = 0; j < jn; j++)
                if (j0 + j != i) sum = sum + D[tid * 65u + j];
        }
        // (the next tile's first __syncthreads() orders these reads before D is rewritten)
    }
    if (tid < 64u && i0 + tid < d.N) {
        double fd = sum / (double)(d.N - 1u);
        if (fd == 0.0) fd = 2.2250738585072014e-308;    // f64::MIN_POSITIVE, population.rs:774-776
        out[i0 + tid] = fd;
    }
}

// Workgroup = 16 individuals.  The fold of an individual is sequential (f64 addition in ascending j, population.rs:770), the
// loads are not: all 256 threads stage 64 rows x 16 columns of Dt in LDS (2 x 8.5 KB: it fits beside the sweep) (the next chunk's loads are in flight while 16
// threads add the current one), so a thread's chain of additions never waits for memory.  (One thread per individual
// reading its own column, 16 loads in flight: a thousand rows = 62 memory latencies in a row, 43 us at N = 1000 whether
// beside the sweep or not; deeper register prefetch changed nothing.)
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. waits for the counts that were
// requested a moment ago for two chunks ahead -- a memory latency per chunk (2.0 us; 2.07 ms for the 1024 chunks of a row
// shard at N = 65536, whatever the prefetch depth)
__device__ __forceinline__ void ps_acc_sync_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The ordered fold of one chunk: sum = (((sum + d[0]) + d[1]) + ...) over the chunk's JB distances of individual `tid`, in
// ascending j (population.rs:770).  The chain of dependent f64 additions is the one serial thing in D-avg -- N - 1 of them
// per individual -- so nothing else may sit on it: the terms that are skipped (j == i, population.rs:126-128; j >= N) are
// replaced by +0.0 BEFORE they meet the sum (sum + 0.0 == sum for the sums that occur: they start at +0.0 and only ever
// add distances >= +0.0 or NaN), and only in the chunks that contain such a j (workgroup-uniform).  (First form: `if (j <
// N && j != i) sum = sum + w` -- the compiler selected on the RESULT, an add, a compare and two levels of v_cndmask per
// element on the chain: 65 cycles per element, 2.07 ms for a row shard at N = 65536.)
// the same fold for a kernel that must stay within 64 VGPRs beside a sweep: 16 reads, then their 16 additions
template <uint32_t JB, uint32_t STRIDE>
__device__ __forceinline__ double ps_da_fold_chunk_lean(const double *S, uint32_t tid, double sum, uint32_t j_base, uint32_t N, uint32_t i, bool need_mask)
{
#pragma unroll 1
    for (uint32_t jj = 0; jj < JB; jj += 16u) {
        double w[16];
#pragma unroll
        for (uint32_t u = 0; u < 16u; u++) w[u] = S[(jj + u) * STRIDE + tid];
        if (need_mask) {
#pragma unroll
            for (uint32_t u = 0; u < 16u; u++) {
                const uint32_t j = j_base + jj + u;
                w[u] = (j < N && j != i) ? w[u] : 0.0;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < 16u; u++) sum = sum + w[u];
    }
    return sum;
}

template <uint32_t JB, uint32_t STRIDE, bool need_mask>
__device__ __forceinline__ double ps_da_fold_chunk_impl(const double *S, uint32_t tid, double sum, uint32_t j_base, uint32_t N, uint32_t i)
{
    // eight distances are read from LDS while the eight before them are added (two sets of 8 doubles, ping-pong; the loop
    // is kept rolled: unrolled, the compiler hoists all JB reads to the top and spills -- the kernels that call this are
    // built for 64 VGPRs, 8 waves per SIMD, so that they fit beside a sweep)
    static_assert(JB % 16u == 0u, "chunks of a multiple of 16 distances");
    double w[8], x[8];
#pragma unroll
    for (uint32_t u = 0; u < 8u; u++) w[u] = S[u * STRIDE + tid];
#pragma unroll 1
    for (uint32_t jj = 0; jj < JB; jj += 16u) {
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) x[u] = S[(jj + 8u + u) * STRIDE + tid];
        if (need_mask) {
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                const uint32_t j = j_base + jj + u;
                w[u] = (j < N && j != i) ? w[u] : 0.0;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) sum = sum + w[u];
        if (jj + 16u < JB) {
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) w[u] = S[(jj + 16u + u) * STRIDE + tid];
        }
        if (need_mask) {
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                const uint32_t j = j_base + jj + 8u + u;
                x[u] = (j < N && j != i) ? x[u] : 0.0;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) sum = sum + x[u];
    }
    return sum;
}

template <uint32_t JB, uint32_t STRIDE>
__device__ __forceinline__ double ps_da_fold_chunk(const double *S, uint32_t tid, double sum, uint32_t j_base, uint32_t N, uint32_t i, bool need_mask)
{
    return need_mask ? ps_da_fold_chunk_impl<JB, STRIDE, true>(S, tid, sum, j_base, N, i)
                     : ps_da_fold_chunk_impl<JB, STRIDE, false>(S, tid, sum, j_base, N, i);
}

#define PS_AV_IB 16u
#define PS_AV_JB 64u
__global__ void __launch_bounds__(256, 8) acc_average_from_matrix_kernel(const double *Dt, double *out, acc_dims d)
{
    __shared__ double S[2][PS_AV_JB * (PS_AV_IB + 1u)];
    __builtin_amdgcn_s_setprio(3);          // (on the critical chain of a --competition_strength generation, see the pair matrix)
    const uint32_t tid = threadIdx.x, i0 = blockIdx.x * PS_AV_IB;
    const uint32_t row = tid >> 2, quarter = tid & 3u;         // this thread stages 4 doubles of chunk row `row`
    const uint32_t nch = (d.N + PS_AV_JB - 1u) / PS_AV_JB;
    double v[4];
    auto load = [&](uint32_t c) {
        const uint32_t j = c * PS_AV_JB + row;
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            const uint32_t ii = i0 + quarter * 4u + u;
            v[u] = (j < d.N && ii < d.N) ? Dt[(uint64_t)j * d.N + ii] : 0.0;      // = distance(ii, j)
        }
    };
    auto park = [&](uint32_t buf) {
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) S[buf][row * (PS_AV_IB + 1u) + quarter * 4u + u] = v[u];
    };
    load(0u);
    park(0u);
    __syncthreads();
    double sum = 0.0;
    const uint32_t i = i0 + tid;             // (threads 0..15 fold)
    for (uint32_t c = 0; c < nch; c++) {
        const uint32_t buf = c & 1u;
        if (c + 1u < nch) load(c + 1u);
        if (tid < PS_AV_IB) {
            // (rows past N hold zeros; the j == i term is replaced by +0.0 before it meets the sum: see ps_da_fold_chunk)
            const uint32_t jb = c * PS_AV_JB;
            const bool need_mask = jb < i0 + PS_AV_IB && jb + PS_AV_JB > i0;
            sum = ps_da_fold_chunk_lean<PS_AV_JB, PS_AV_IB + 1u>(S[buf], tid, sum, jb, d.N, i, need_mask);
        }
        if (c + 1u < nch) park(buf ^ 1u);
        __syncthreads();
    }
    if (tid < PS_AV_IB && i < d.N) {
        double fd = sum / (double)(d.N - 1u);
        if (fd == 0.0) fd = 2.2250738585072014e-308;    // f64::MIN_POSITIVE, population.rs:774-776
        out[i] = fd;
    }
}

// ---------------------------------------------------------------------------
// D-avg on the matrix cores (round 4; population.rs:753-784 + :114-151 for wide populations).
// |x_i n x_j| over all pairs is the {0, 1} contraction X X^T (K = the G genes): exact on
// v_mfma_scale_f32_32x32x64_f8f6f4 with both operands E2M1 (a present gene is the FP4 value 1.0 = 0b0010, block scales
// 2^0) -- {0, 1} products, f32 sums <= G <= 65536 < 2^24.  Unions come from the row counts (|x u y| = |x| + |y| - |x n y|),
// the distance 1 - ((n + 0.0 + core_genes) / (u + 0.0 + core_genes)) is evaluated in f64 exactly as get_distance writes it,
// and every individual's distances are added in ASCENDING j by ONE lane (the reference's sequential fold, :770): only the
// counting is re-associated, and integer counts have no rounding.
//   * acc_rows_pad_kernel: the individual-major bit rows copied to rows of WP dwords (a multiple of 8: whole 256-gene
//     chunks) and Npad rows (a multiple of 128), zero padded, in the blocked order of ps_da_row_offset, plus the row
//     popcounts: the contraction kernel then has no edge cases in its loads and every load is 1 KB of consecutive bytes.
//   * acc_average_distance_mfma_kernel<NB>: a WAVE owns 32 NB individuals i (the B operand: columns of the 32 x 32
//     accumulator blocks, i = lane & 31 of block b) and sweeps all j in steps of 128 (four A fragments: rows).  Lane
//     (r, h) of a fragment takes the genes [256 c + 128 h, +128) of chunk c of individual base + r: one 16-byte load per
//     chunk; K-step t of the chunk expands dword t -- 32 gene bits -- to 32 FP4 nibbles with four reads of a
//     byte -> 8-nibble table in LDS (256 entries x 64 copies x 4 bytes = 64 KB at LDS offset 0: copy = lane, so the 64
//     lanes of a ds_read_b32 hit 64 different banks).  The same K order on both operands is all the contraction needs.
//   * fold: in the accumulator layout a lane holds column i = lane & 31 and the rows (v & 3) + 8 (v >> 2) + 4 (lane >> 5),
//     i.e. ascending j alternates between the two halves of the wave in groups of four: the running sum of i travels
//     between lane i and lane i + 32 after every four additions (both halves execute the additions; only the copy that
//     holds the live sum is ever passed on; v_permlane32_swap, no LDS round trip).
// Sharding: a rank computes rows [i_lo, i_lo + i_cnt) against all N columns; the N doubles are all-gathered.
// ---------------------------------------------------------------------------
// Blocked rows: [group of 32 rows][chunk of 8 dwords][half h][row % 32][4 dwords] -- what one operand fragment loads for a
// chunk (lane = h * 32 + r, 16 bytes each) is 1 KB of consecutive addresses.  Row-major rows gave every load instruction 32
// bytes of 32 different 128-byte lines, each crossing the L2 -> L1 path four times: the contraction ran at 0.40 of the FP4 peak
// on cache fills (round 6).  Offset of dword 0 of `row` (half 0, chunk 0); a chunk is 256 dwords further, half 1 128 dwords.
__device__ __forceinline__ size_t ps_da_row_offset(uint32_t row, uint32_t nch)
{
    return (size_t)(row >> 5) * nch * 256u + (row & 31u) * 4u;
}

__global__ void __launch_bounds__(256) acc_rows_pad_kernel(const uint64_t *accI, uint32_t *rowsP, uint32_t *rowcnt, acc_dims d,
                                                           uint32_t WP, uint32_t Npad)
{
    // one wave per padded row
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (row >= Npad) return;
    const uint32_t *src = (const uint32_t *)(accI + (uint64_t)row * d.GW);
    uint32_t cnt = 0;
    for (uint32_t w = lane; w < WP; w += 64u) {
        const uint32_t v = (row < d.N && w < 2u * d.GW) ? src[w] : 0u;
        rowsP[ps_da_row_offset(row, WP / 8u) + (size_t)(w >> 3) * 256u + ((w >> 2) & 1u) * 128u + (w & 3u)] = v;
        cnt += __popc(v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) rowcnt[row] = cnt;
}

typedef int ps_da_v8i __attribute__((ext_vector_type(8)));
typedef float ps_da_v16f __attribute__((ext_vector_type(16)));
typedef uint32_t ps_da_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t ps_u32x4_acc __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t ps_da_lut(uint32_t raw, uint32_t colofs, uint32_t b)
{
    uint32_t addr;      // (byte b of raw) << 8 | colofs  (colofs = lane * 4 < 256); the table sits at LDS offset 0
    switch (b) {
    case 0: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0400u); break;
    case 1: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0500u); break;
    case 2: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0600u); break;
    default: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0700u); break;
    }
    return *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)addr;
}

// the f64 of lane (l & 31) + 32 * upper, in lanes l and l ^ 32 (v_permlane32_swap: the upper half of its first operand
// trades places with the lower half of its second -- with both operands the same register, the first result carries the
// lower half's values in both halves and the second the upper half's; VALU, no LDS round trip)
__device__ __forceinline__ double ps_da_half_bcast(double s, bool upper)
{
    const uint32_t lo = (uint32_t)__double2loint(s), hi = (uint32_t)__double2hiint(s);
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return upper ? __hiloint2double((int)r1[1], (int)r0[1]) : __hiloint2double((int)r1[0], (int)r0[0]);
}

// one Jaccard distance from its integer counts, bit for bit the value of `1.0 - ((n + 0.0 + core_genes) / (u + 0.0 +
// core_genes))` (population.rs:144-145).  FAST = the lean form described in the epilogue of
// acc_average_distance_mfma_kernel below (one integer add + one conversion per operand; the arithmetic core of the
// compiler's own f64 division without the scale / fixup steps that only act at the ends of the exponent range).
template <bool FAST>
__device__ __forceinline__ double ps_da_distance(uint32_t in, uint32_t un, double core_genes, uint32_t cg_int)
{
    if (FAST) {
        const double num = (double)(in + cg_int), den = (double)(un + cg_int);
        double rc = __builtin_amdgcn_rcp(den);
        double er = __builtin_fma(-den, rc, 1.0);
        rc = __builtin_fma(rc, er, rc);
        er = __builtin_fma(-den, rc, 1.0);
        rc = __builtin_fma(rc, er, rc);
        const double q0 = num * rc;
        const double rem = __builtin_fma(-den, q0, num);
        return 1.0 - __builtin_fma(rem, rc, q0);
    }
    return 1.0 - (((double)in + 0.0 + core_genes) / ((double)un + 0.0 + core_genes));
}

// ---------------------------------------------------------------------------
// D-avg in TWO PHASES (round 5): the one-kernel form below keeps a wave on its 32 NB individuals for the whole ordered
// fold -- 256 waves for 1024 SIMDs when a rank of 8 folds its 8192 rows, and a serial f64 epilogue between the MFMA
// bursts (0.31 of the FP4 peak).  Split by what each part needs:
//   phase 1, acc_intersections_mfma_kernel<NB>: ONLY the {0, 1} contraction, on every SIMD.  2-D grid: x = groups of four
//     waves x 32 NB individuals i, y = segments of `jsteps` x 128 individuals j; the same operands, table and K order as
//     the one-kernel form.  The exact f32 counts (<= G <= 65535) leave as u16 into In[i - i_lo][j] (row pitch `ld`): a
//     lane holds column i and four consecutive j per accumulator group, one 8-byte store each.  2 bytes per pair instead
//     of an f64 distance: 8.6 GB for N = 65536 against 34 GB, and the division moves to phase 2 where it costs nothing.
//   phase 2, acc_average_from_counts_kernel<FAST>: the reference's fold (population.rs:770: one running f64 sum per
//     individual, ascending j).  Workgroup = 16 individuals; all 256 threads turn a chunk of 64 j x 16 i counts into f64
//     distances (the unions from the row counts, the division exactly as get_distance writes it) and park them in LDS
//     while 16 threads add the previous chunk in ascending j -- the chain of dependent additions never waits for memory
//     or for a division.
// Bit-equal to the one-kernel form (and to the CPU restatement the tests compare with): the counts are integers, the distance expression and the order of
// the additions are unchanged.
// ---------------------------------------------------------------------------
// The contraction of one 128 x 32 NB block of pairs over all chunks, software-pipelined like core_allpairs_mfma_fp4_kernel: the
// table reads of K-step t + 1 are issued before the MFMAs of step t (two operand sets, ping-pong by the parity of t; four
// steps per chunk, so the parity carries over the chunk loop), the 16-byte loads of the next chunk are in flight during the
// whole chunk.  (Left to the compiler, 26 of the loop's waits sat between its 32 MFMAs: round 6.)
// PIPE = false leaves the order to the compiler: better where the registers are short (the one-kernel form at NB = 2: 11.1 ms
// against 12.3 at N = 65536; NB = 4), worse elsewhere (phase 1 at NB = 2: 10.4 against 9.0 ms).
template <uint32_t NB, bool PIPE>
__device__ __forceinline__ void ps_da_contract(const uint32_t *const (&srcA)[4], const uint32_t *const (&srcB)[NB], uint32_t nch,
                                               uint32_t colofs, ps_da_v16f (&acc)[4][NB])
{
    const int one = 0x7f7f7f7f;          // E8M0 block scale 2^0 in every byte
    uint4 cur[4 + NB], nxt[4 + NB];
    if (!PIPE) {
#pragma unroll
        for (uint32_t f = 0; f < 4u + NB; f++) cur[f] = *(const uint4 *)((f < 4u ? srcA[f] : srcB[f - 4u]));
        for (uint32_t c = 0; c < nch; c++) {
            const uint32_t cn = min(c + 1u, nch - 1u);
#pragma unroll
            for (uint32_t f = 0; f < 4u + NB; f++) nxt[f] = *(const uint4 *)((f < 4u ? srcA[f] : srcB[f - 4u]) + (size_t)cn * 256u);
#pragma unroll
            for (uint32_t t = 0; t < 4u; t++) {
                ps_da_v8i op[4 + NB];
#pragma unroll
                for (uint32_t f = 0; f < 4u + NB; f++) {
                    const uint32_t raw = t == 0u ? cur[f].x : t == 1u ? cur[f].y : t == 2u ? cur[f].z : cur[f].w;
                    op[f] = ps_da_v8i{ (int)ps_da_lut(raw, colofs, 0u), (int)ps_da_lut(raw, colofs, 1u), (int)ps_da_lut(raw, colofs, 2u),
                                       (int)ps_da_lut(raw, colofs, 3u), 0, 0, 0, 0 };
                }
#pragma unroll
                for (uint32_t a = 0; a < 4u; a++)
#pragma unroll
                    for (uint32_t b = 0; b < NB; b++)
                        acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(op[a], op[4u + b], acc[a][b], 4, 4, 0, one, 0, one);
            }
#pragma unroll
            for (uint32_t f = 0; f < 4u + NB; f++) cur[f] = nxt[f];
        }
        return;
    }
    ps_u32x4_acc opA[4 + NB], opB[4 + NB];
    auto fetch = [&](const uint4 &w, uint32_t t, ps_u32x4_acc &op) {
        const uint32_t raw = t == 0u ? w.x : t == 1u ? w.y : t == 2u ? w.z : w.w;
        op = ps_u32x4_acc{ ps_da_lut(raw, colofs, 0u), ps_da_lut(raw, colofs, 1u), ps_da_lut(raw, colofs, 2u), ps_da_lut(raw, colofs, 3u) };
    };
#pragma unroll
    for (uint32_t f = 0; f < 4u + NB; f++) cur[f] = *(const uint4 *)((f < 4u ? srcA[f] : srcB[f - 4u]));
#pragma unroll
    for (uint32_t f = 0; f < 4u + NB; f++) fetch(cur[f], 0u, opA[f]);
    for (uint32_t c = 0; c < nch; c++) {
        const uint32_t cn = min(c + 1u, nch - 1u);
#pragma unroll
        for (uint32_t f = 0; f < 4u + NB; f++) nxt[f] = *(const uint4 *)((f < 4u ? srcA[f] : srcB[f - 4u]) + (size_t)cn * 256u);
#pragma unroll
        for (uint32_t t = 0; t < 4u; t++) {
            const uint32_t tn = (t + 1u) & 3u;
#pragma unroll
            for (uint32_t f = 0; f < 4u + NB; f++) {
                if (t & 1u) fetch((t == 3u) ? nxt[f] : cur[f], tn, opA[f]);
                else fetch((t == 3u) ? nxt[f] : cur[f], tn, opB[f]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (uint32_t a = 0; a < 4u; a++)
#pragma unroll
                for (uint32_t b = 0; b < NB; b++) {
                    const ps_u32x4_acc &xa = (t & 1u) ? opB[a] : opA[a], &xb = (t & 1u) ? opB[4u + b] : opA[4u + b];
                    const ps_da_v8i va = { (int)xa.x, (int)xa.y, (int)xa.z, (int)xa.w, 0, 0, 0, 0 };
                    const ps_da_v8i vb = { (int)xb.x, (int)xb.y, (int)xb.z, (int)xb.w, 0, 0, 0, 0 };
                    acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc[a][b], 4, 4, 0, one, 0, one);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (uint32_t f = 0; f < 4u + NB; f++) cur[f] = nxt[f];
    }
}

template <uint32_t NB>
__global__ void __launch_bounds__(256, NB == 2u ? 2 : 1) acc_intersections_mfma_kernel(const uint32_t *rowsP, uint32_t WP, uint32_t Npad,
                                                                                      uint32_t i_lo, uint32_t i_cnt, uint32_t jsteps,
                                                                                      uint16_t *In, uint32_t ld)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lut[];      // 256 entries x 64 copies x 4 bytes, at LDS offset 0
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t x = tid; x < 256u * 64u; x += 256u) {
        const uint32_t e = x >> 6;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) v |= ((e >> k) & 1u) ? (2u << (4u * k)) : 0u;       // gene k of the byte -> nibble k = E2M1 1.0
        *(uint32_t *)(lut + (size_t)x * 4u) = v;         // x = e * 64 + copy
    }
    __syncthreads();
    const uint32_t colofs = lane << 2;
    const uint32_t r = lane & 31u, h = lane >> 5;
    const uint32_t i_rel = (blockIdx.x * 4u + wave) * 32u * NB;       // first row of this wave inside the shard
    if (i_rel >= i_cnt) return;                          // (wave-uniform; no barrier follows)
    const uint32_t *srcB[NB];
    const uint32_t nch = WP / 8u;
#pragma unroll
    for (uint32_t b = 0; b < NB; b++) srcB[b] = rowsP + ps_da_row_offset(min(i_lo + i_rel + 32u * b + r, Npad - 1u), nch) + h * 128u;
    const uint32_t j_begin = blockIdx.y * jsteps * 128u, j_end = min(Npad, j_begin + jsteps * 128u);
    for (uint32_t j0 = j_begin; j0 < j_end; j0 += 128u) {
        const uint32_t *srcA[4];
#pragma unroll
        for (uint32_t a = 0; a < 4u; a++) srcA[a] = rowsP + ps_da_row_offset(j0 + 32u * a + r, nch) + h * 128u;
        ps_da_v16f acc[4][NB];
#pragma unroll
        for (uint32_t a = 0; a < 4u; a++)
#pragma unroll
            for (uint32_t b = 0; b < NB; b++)
#pragma unroll
                for (int v = 0; v < 16; v++) acc[a][b][v] = 0.0f;
        // (carrying the pipeline over the j steps -- the next step's first chunk loaded and expanded during this step's last --
        // changed nothing: 0.987 against 0.98 ms for a row shard of 8 at N = 65536)
        ps_da_contract<NB, (NB <= 2u)>(srcA, srcB, nch, colofs, acc);
        // block (a, b): column i = i_rel + 32 b + r, rows j0 + 32 a + 8 g + 4 h + q (q = 0..3): four consecutive j per group
#pragma unroll
        for (uint32_t b = 0; b < NB; b++) {
            uint16_t *row = In + (size_t)(i_rel + 32u * b + r) * ld + j0 + 4u * h;
#pragma unroll
            for (uint32_t a = 0; a < 4u; a++)
#pragma unroll
                for (uint32_t g = 0; g < 4u; g++) {
                    const uint32_t c0 = (uint32_t)acc[a][b][4 * g], c1 = (uint32_t)acc[a][b][4 * g + 1];
                    const uint32_t c2 = (uint32_t)acc[a][b][4 * g + 2], c3 = (uint32_t)acc[a][b][4 * g + 3];
                    // (plain stores: a 128-byte line of In is made of 16 such 8-byte pieces from 8 store instructions of this
                    // wave, and only the L2 merges them -- as nontemporal stores every piece went to memory on its own: 29.6 ms
                    // for the whole population against 9.5)
                    *(ps_da_u32x2 *)(row + 32u * a + 8u * g) = ps_da_u32x2{ c0 | (c1 << 16), c2 | (c3 << 16) };
                }
        }
    }
}

// phase 2 (see above).  Counts of rows [i_lo, i_lo + i_cnt) against all N columns; out[i] for those rows only.
// Workgroup = IB individuals, 8 IB staging threads (every thread turns 8 consecutive j of one individual into distances per
// chunk: one 16-byte load of counts) + one wave that does nothing but the fold (IB lanes: one running sum each), so that
// the chain of dependent additions -- the one serial thing here -- shares its wave with no division.  Two chunks of counts
// are in flight.  (Built for 128 VGPRs: this kernel runs with phase 1, ahead of the sweep, not beside it.)
#define PS_AC_JB 64u
// IB individuals per workgroup: 8 IB staging threads + the fold wave (the host chooses: 16 nearly everywhere since round 6 -- the
// fold's chain of N dependent additions sets a workgroup's duration, and with 16 four workgroups share a CU.  Four columns per
// staging thread instead of eight, i.e. twice the staging threads, was slower everywhere: 7.0 against 5.4 ms at N = 65536)
template <bool FAST, uint32_t IB>
__global__ void __launch_bounds__(IB * 8u + 64u, IB == 32u ? 2 : 4) acc_average_from_counts_kernel(const uint16_t *In, uint32_t ld, const uint32_t *rowcnt, uint32_t N,
                                                                                   uint32_t i_lo, uint32_t i_cnt, double core_genes, uint32_t cg_int,
                                                                                   double *out)
{
    __shared__ double S[2][PS_AC_JB * (IB + 1u)];
    const uint32_t tid = threadIdx.x, i0 = blockIdx.x * IB;       // (relative to the shard)
    const uint32_t nch = (N + PS_AC_JB - 1u) / PS_AC_JB;
    const bool folder = tid >= IB * 8u;                                    // (wave-uniform)
    if (!folder) {
        const uint32_t ii = tid >> 3, jq = tid & 7u;                    // 8 consecutive j of individual i0 + ii per chunk
        const bool live = i0 + ii < i_cnt;
        const uint32_t ci = live ? rowcnt[i_lo + i0 + ii] : 0u;
        const uint16_t *src = In + (size_t)(i0 + ii) * ld + 8u * jq;
        uint4 raw0, raw1, cja0, cjb0, cja1, cjb1;
        auto load = [&](uint32_t c, uint4 &raw, uint4 &cja, uint4 &cjb) {
            // (In has ld >= Npad columns and rowcnt Npad entries: no edge in the loads; rows past N are skipped by the fold)
            if (live) {
                const ps_u32x4_acc v = __builtin_nontemporal_load((const ps_u32x4_acc *)(src + c * PS_AC_JB));
                raw = make_uint4(v.x, v.y, v.z, v.w);
            } else {
                raw = make_uint4(0u, 0u, 0u, 0u);
            }
            cja = *(const uint4 *)(rowcnt + c * PS_AC_JB + 8u * jq);
            cjb = *(const uint4 *)(rowcnt + c * PS_AC_JB + 8u * jq + 4u);
        };
        auto park = [&](double *Sb, const uint4 &raw, const uint4 &cja, const uint4 &cjb) {
            const uint32_t in[8] = { raw.x & 0xffffu, raw.x >> 16, raw.y & 0xffffu, raw.y >> 16, raw.z & 0xffffu, raw.z >> 16, raw.w & 0xffffu, raw.w >> 16 };
            const uint32_t cjv[8] = { cja.x, cja.y, cja.z, cja.w, cjb.x, cjb.y, cjb.z, cjb.w };
#pragma unroll
            for (uint32_t q = 0; q < 8u; q++)
                Sb[(8u * jq + q) * (IB + 1u) + ii] = ps_da_distance<FAST>(in[q], ci + cjv[q] - in[q], core_genes, cg_int);
        };
        load(0u, raw0, cja0, cjb0);
        if (nch > 1u) load(1u, raw1, cja1, cjb1);
        park(S[0], raw0, cja0, cjb0);
        if (nch > 2u) load(2u, raw0, cja0, cjb0);                   // (set 0 is free again)
        ps_acc_sync_lds();
        // chunk c sits in S[c & 1]; the counts of chunk c + 1 are in set (c + 1) & 1, which then takes chunk c + 3
        uint32_t c = 0;
        for (; c + 1u < nch; c += 2u) {
            park(S[1], raw1, cja1, cjb1);
            if (c + 3u < nch) load(c + 3u, raw1, cja1, cjb1);
            ps_acc_sync_lds();
            if (c + 2u < nch) park(S[0], raw0, cja0, cjb0);
            if (c + 4u < nch) load(c + 4u, raw0, cja0, cjb0);
            ps_acc_sync_lds();
        }
        return;
    }
    const uint32_t lane = tid - IB * 8u;          // the fold wave: lanes 0..IB-1 hold one running sum each
    const uint32_t i = i_lo + i0 + lane, ib = i_lo + i0;
    double sum = 0.0;
    auto fold = [&](uint32_t c, const double *Sb) {
        if (lane < IB) {
            const uint32_t jb = c * PS_AC_JB;
            const bool need_mask = jb + PS_AC_JB > N || (jb < ib + IB && jb + PS_AC_JB > ib);
            sum = ps_da_fold_chunk<PS_AC_JB, IB + 1u>(Sb, lane, sum, jb, N, i, need_mask);
        }
    };
    ps_acc_sync_lds();                           // (chunk 0 is parked)
    uint32_t c = 0;
    for (; c + 1u < nch; c += 2u) {
        fold(c, S[0]);
        ps_acc_sync_lds();
        fold(c + 1u, S[1]);
        ps_acc_sync_lds();
    }
    if (c < nch) fold(c, S[0]);
    if (lane < IB && i0 + lane < i_cnt && i < N) {
        double fd = sum / (double)(N - 1u);
        if (fd == 0.0) fd = 2.2250738585072014e-308;    // f64::MIN_POSITIVE, population.rs:774-776
        out[i] = fd;
    }
}

template <uint32_t NB, bool FAST>
__global__ void __launch_bounds__(256, NB == 2u ? 2 : 1) acc_average_distance_mfma_kernel(const uint32_t *rowsP, uint32_t WP, const uint32_t *rowcnt,
                                                                        uint32_t N, uint32_t Npad, uint32_t i_lo, uint32_t i_cnt,
                                                                        double core_genes, uint32_t cg_int, double *out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lut[];      // 256 entries x 64 copies x 4 bytes, at LDS offset 0
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t x = tid; x < 256u * 64u; x += 256u) {
        const uint32_t e = x >> 6;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) v |= ((e >> k) & 1u) ? (2u << (4u * k)) : 0u;       // gene k of the byte -> nibble k = E2M1 1.0
        *(uint32_t *)(lut + (size_t)x * 4u) = v;         // x = e * 64 + copy
    }
    __syncthreads();
    const uint32_t colofs = lane << 2;
    const uint32_t r = lane & 31u, h = lane >> 5;
    const uint32_t i_base = i_lo + (blockIdx.x * 4u + wave) * 32u * NB;
    if (i_base >= i_lo + i_cnt) return;                  // (wave-uniform; no barrier follows)
    const uint32_t *srcB[NB];
    const uint32_t nch = WP / 8u;
    uint32_t ci[NB];
    double sum[NB];
#pragma unroll
    for (uint32_t b = 0; b < NB; b++) {
        const uint32_t i = min(i_base + 32u * b + r, Npad - 1u);
        srcB[b] = rowsP + ps_da_row_offset(i, nch) + h * 128u;
        ci[b] = rowcnt[i];
        sum[b] = 0.0;
    }
    for (uint32_t j0 = 0; j0 < Npad; j0 += 128u) {
        const uint32_t *srcA[4];
#pragma unroll
        for (uint32_t a = 0; a < 4u; a++) srcA[a] = rowsP + ps_da_row_offset(j0 + 32u * a + r, nch) + h * 128u;
        ps_da_v16f acc[4][NB];
#pragma unroll
        for (uint32_t a = 0; a < 4u; a++)
#pragma unroll
            for (uint32_t b = 0; b < NB; b++)
#pragma unroll
                for (int v = 0; v < 16; v++) acc[a][b][v] = 0.0f;
        ps_da_contract<NB, (NB == 1u)>(srcA, srcB, nch, colofs, acc);
        // distances and the ordered fold.  Block (a, b): column i = i_base + 32 b + r, rows j0 + 32 a + (v & 3) + 8 (v >> 2) + 4 h.
        // The epilogue is VALU work -- PMC: 30 instructions per distance in the plain form, more wave-cycles than the
        // contraction itself -- so it is written lean, without changing a bit of any distance:
        //   * (double)n + 0.0 + core_genes == (double)(n + core_genes) for integers below 2^32 (x + 0.0 == x for x >= 0; the sum
        //     of two integers below 2^53 is exact): one integer add + one conversion instead of a conversion + two f64 adds;
        //   * the quotient by the arithmetic core of the compiler's own f64 division -- v_rcp_f64, two Newton steps, the
        //     product, its residual, one correction: the very fmas of the IEEE sequence -- without v_div_scale / v_div_fmas /
        //     v_div_fixup, which only act on operands near the ends of the exponent range, zeros, infinities (here: integers in
        //     [0, 2^32), and 0 / 0 still gives NaN through rcp(0) = inf, 0 * inf);
        //   * the j == i / j >= N selects only in the steps that can contain such a j (wave-uniform).
        const bool need_mask = (j0 + 128u > N) || (j0 < i_base + 32u * NB && j0 + 128u > i_base);
        auto epilogue = [&](const bool masked) __attribute__((always_inline)) {
#pragma unroll
        for (uint32_t a = 0; a < 4u; a++) {
#pragma unroll
            for (uint32_t g = 0; g < 4u; g++) {
                const uint32_t jr = j0 + 32u * a + 8u * g + 4u * h;        // this lane's four rows of group g: jr .. jr + 3
                const uint4 cj4 = *(const uint4 *)(rowcnt + jr);
                const uint32_t cjv[4] = { cj4.x, cj4.y, cj4.z, cj4.w };
                double dv[NB][4];
#pragma unroll
                for (uint32_t b = 0; b < NB; b++) {
                    const uint32_t i = i_base + 32u * b + r;
#pragma unroll
                    for (uint32_t q = 0; q < 4u; q++) {
                        const uint32_t in = (uint32_t)acc[a][b][4 * g + q], j = jr + q;
                        const uint32_t un = ci[b] + cjv[q] - in;
                        double pd;
                        if (FAST) {
                            const double num = (double)(in + cg_int), den = (double)(un + cg_int);
                            double rc = __builtin_amdgcn_rcp(den);
                            double er = __builtin_fma(-den, rc, 1.0);
                            rc = __builtin_fma(rc, er, rc);
                            er = __builtin_fma(-den, rc, 1.0);
                            rc = __builtin_fma(rc, er, rc);
                            const double q0 = num * rc;
                            const double rem = __builtin_fma(-den, q0, num);
                            pd = 1.0 - __builtin_fma(rem, rc, q0);
                        } else {
                            pd = 1.0 - (((double)in + 0.0 + core_genes) / ((double)un + 0.0 + core_genes));
                        }
                        dv[b][q] = (masked && (j == i || j >= N)) ? 0.0 : pd;     // the j == i term is skipped (:126-128); + 0.0 leaves a sum >= 0 as it is
                    }
                }
                // rows 8 g .. 8 g + 3 sit in the lower half of the wave, 8 g + 4 .. 8 g + 7 in the upper; the live sum of column i
                // is in lane i when the group starts.  Every lane adds its four rows (the upper half on a stale copy), the
                // sums swap halves -- the live one is now in lane i + 32, which has not added its rows to IT yet -- every
                // lane adds its four rows again (now the lower half works on the stale copy), and the sums swap back.
#pragma unroll
                for (uint32_t b = 0; b < NB; b++) {
                    double s = sum[b];
#pragma unroll
                    for (uint32_t q = 0; q < 4u; q++) s = s + dv[b][q];
                    s = ps_da_half_bcast(s, false);        // the lower half's sums, in both halves
#pragma unroll
                    for (uint32_t q = 0; q < 4u; q++) s = s + dv[b][q];
                    s = ps_da_half_bcast(s, true);         // the upper half's sums, in both halves
                    sum[b] = s;
                }
            }
        }
        };
        if (need_mask) epilogue(true);
        else epilogue(false);
    }
#pragma unroll
    for (uint32_t b = 0; b < NB; b++) {
        const uint32_t i = i_base + 32u * b + r;
        if (h == 0u && i < N && i < i_lo + i_cnt) {
            double fd = sum[b] / (double)(N - 1u);
            if (fd == 0.0) fd = 2.2250738585072014e-308;    // f64::MIN_POSITIVE, population.rs:774-776
            out[i] = fd;
        }
    }
}
