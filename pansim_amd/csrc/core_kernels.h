// core_kernels.h -- gfx950 kernels for the core-genome matrix (site-major u8 in HBM).
//
// HBM layout: state[row][pitch], one row per core SITE, byte i of a row is the
// allele (1/2/4/8) of individual i at that site; pitch = N rounded up to 128 so
// that a row is a whole number of 16-byte lane chunks and starts 128-B aligned.
// Every per-generation operator is row-local in this layout:
//   gather  child[s][i] = parent[s][idx[i]]                 population.rs:450-465
//   mutate  cell (s,i) <- 2/4/8 with prob 1-exp(-lam/L)     population.rs:511-540
//   HR      cell (s,r) <- post-mutation cell (s,d)          population.rs:544-751
// so one workgroup owns a row at a time, stages it in LDS, and writes it back in
// place: algorithmic HBM traffic is one read and one write of N*L bytes.
#pragma once

#include "ps_common.h"

#include <type_traits>

struct core_sweep_args {
    uint8_t *state;        // the generation that is read
    uint8_t *out;          // where the new generation is written: == state (in place) or the second buffer (out of place)
    const uint32_t *idx;   // parents (device), DO_GATHER only
    const uint32_t *idxT;  // idxT[k * cpr + chunk] = idx[16 * chunk + k] (block sweep)
    uint32_t N, pitch, cpr, rows;
    uint32_t site_offset;  // global site index of local row 0 (Philox counter)
    uint32_t gen, k0, k1;
    ps_core_plan plan;
    uint32_t *overflow_flag;   // host-mapped sticky error word
    unsigned long long *stamps; // diagnostic builds (PS_STAMP) only
    uint32_t *work_ctr;        // wave sweep: 2 x 8 chunk counters, 128 bytes apart
    uint32_t launch_parity;    // which counter set this launch uses
    uint32_t nt;               // out-of-place form: nontemporal row loads / stores (every byte is touched once per launch)
    uint32_t qcap;             // wave / window sweeps: entries of a wave's candidate queue (sized by the host for the plan)
    uint32_t qcap_limit;       // tests: pretend the candidate queues / HR lists hold only this many entries (0 = their real size)
    uint32_t *wide_flags;      // window sweep: [(generation & 1) * 32] != 0 iff the first launch met a segment for the second one
};

typedef uint32_t ps_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ps_load_row16(const uint8_t *p, bool nt)
{
    if (nt) {
        const ps_u32x4 v = __builtin_nontemporal_load((const ps_u32x4 *)p);      // global_load_dwordx4 ... nt
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *(const uint4 *)p;
}
__device__ __forceinline__ void ps_store_row16(uint8_t *p, const uint4 &v, bool nt)
{
    if (nt) {
        ps_u32x4 w = { v.x, v.y, v.z, v.w };
        __builtin_nontemporal_store(w, (ps_u32x4 *)p);
    } else {
        *(uint4 *)p = v;
    }
}

__device__ __forceinline__ void ps_set_byte(uint32_t (&w)[4], uint32_t k, uint32_t v)
{
    const uint32_t j = k >> 2, sh = (k & 3u) * 8u;
    const uint32_t m = ~(0xFFu << sh), val = v << sh;
#pragma unroll
    for (uint32_t jj = 0; jj < 4; jj++) w[jj] = (jj == j) ? ((w[jj] & m) | val) : w[jj];
}

#ifndef PS_WAVE_DMA
#define PS_WAVE_DMA 0      // wave sweep: 1 = parent rows by LDS-DMA; measured no faster than through registers (profiles/r03_sweep_experiments.md)
#endif
// one 16-byte piece per lane straight into LDS (global_load_lds_dwordx4: no VGPR, asynchronous; the LDS destination is
// the wave-uniform base + lane * 16)
__device__ __forceinline__ void ps_dma16(const uint8_t *gsrc, uint8_t *lds_base, bool nt)
{
    if (nt) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                             (__attribute__((address_space(3))) void *)lds_base, 16, 0, 2);
    else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                          (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}

// words of a block A whose set bits rule a cell out as a candidate: planes at or above cshift (ps_common.h)
__device__ __forceinline__ uint32_t ps_noncand_word(const ps_u4 &A, uint32_t cshift)
{
    return (cshift <= 0u ? A.x : 0u) | (cshift <= 1u ? A.y : 0u) | (cshift <= 2u ? A.z : 0u) | (cshift <= 3u ? A.w : 0u);
}

// Plane-word geometry (ps_plane_pos): bit 8b + s of a word = cell 4 (s % 4) + b of the chunk at site parity s / 4.
// the bits of one site of the pair
__device__ __forceinline__ uint32_t ps_site_bits(uint32_t site) { return 0x0F0F0F0Fu << (4u * (site & 1u)); }
// cell (0..15) of a plane bit
__device__ __forceinline__ uint32_t ps_bit_cell(uint32_t p) { return ((p & 3u) << 2) | (p >> 3); }
// the bits (both sites) of the cells k < nvalid
__device__ __forceinline__ uint32_t ps_valid_word(uint32_t nvalid)
{
    if (nvalid >= 16u) return 0xFFFFFFFFu;
    uint32_t m = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16u; k++)
        if (k < nvalid) m |= 0x11u << (8u * (k & 3u) + (k >> 2));
    return m;
}

// plane-word mask of the cells of one site that may hold an event (candidates: no bit of n at or above cshift)
__device__ __forceinline__ uint32_t ps_cand_word(const ps_u4 &A, uint32_t site, uint32_t cshift)
{
    return ~ps_noncand_word(A, cshift) & ps_site_bits(site);
}

// Queue-free evaluation of one candidate cell from its two blocks: the allele it mutates to (0 = none) and whether it
// receives a donor allele (then l2y is the word the donor is drawn from).  A pure function of (seed, generation, site,
// individual): the redo paths, the inline sweep and the window sweep's donor recomputation all go through it.
__device__ __forceinline__ ps_cell ps_cell_events(const ps_u4 &A, const ps_u4 &B, uint32_t site, uint32_t ind, uint32_t gen,
                                                  uint32_t k0, uint32_t k1, const ps_core_plan &pl, uint32_t &l2y)
{
    ps_cell o = { 0u, 0u };
    const uint32_t n = ps_cell_nibble(A, site, ind);
    if (n >> pl.cshift) return o;
    const uint32_t code = ps_sym_code(4u * n + ps_cell_pair(B, site, ind), pl);
    if (code > 1u) {
        o.mut = code;
    } else if (code == 1u) {
        const ps_u4 l2 = ps_philox(site, ind, gen, PS_STREAM_CORE_L2, k0, k1);
        o = ps_classify(l2.x, pl);
        l2y = l2.y;
    }
    return o;
}

// Inline block sweep: the queue-free fallback of the block sweep below.  One 1024-thread
// workgroup per row; every candidate is handled by its owner lane (lanes diverge), HR cells are
// remembered in a 16-bit mask per chunk.  Correct for any rates (nothing can overflow); used when
// the queues of core_sweep_block_kernel cannot be sized safely.
template <bool DO_GATHER, bool DO_MUT, bool DO_HR>
__global__ void __launch_bounds__(1024) core_sweep_inline_kernel(core_sweep_args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x, lpr = blockDim.x;
    uint8_t *rowA = lds;
    uint8_t *rowS = rowA + a.pitch;
    uint16_t *hrm = (uint16_t *)(rowS + a.pitch);
    const ps_core_plan pl = a.plan;
    const bool events = pl.has_events && (DO_MUT || DO_HR);

    for (uint32_t row = blockIdx.x; row < a.rows; row += gridDim.x) {
        const uint32_t site = a.site_offset + row;
        uint8_t *grow = a.state + (size_t)row * a.pitch;
        if (DO_GATHER) {
            for (uint32_t c = lane; c < a.cpr; c += lpr)
                *(uint4 *)(rowA + 16u * c) = *(const uint4 *)(grow + 16u * c);
            __syncthreads();
        }
        for (uint32_t c = lane; c < a.cpr; c += lpr) {
            uint32_t d[4];
            if (DO_GATHER) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t w = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t i = c * 16u + 4 * j + b;
                        const uint32_t v = (i < a.N) ? (uint32_t)rowA[a.idx[i]] : 0u;
                        w |= v << (8 * b);
                    }
                    d[j] = w;
                }
            } else {
                const uint4 v = *(const uint4 *)(grow + 16u * c);
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            uint32_t hm = 0;
            if (events) {
                const ps_u4 A = ps_philox_l1a(site >> 1, c, a.gen, a.k0, a.k1), B = ps_philox_l1b(site >> 2, c, a.gen, a.k0, a.k1);
                uint32_t cm = ps_cand_word(A, site, pl.cshift);
                while (cm) {
                    const uint32_t k = ps_bit_cell(__builtin_ctz(cm));
                    cm &= cm - 1u;
                    const uint32_t i = c * 16u + k;
                    if (i >= a.N) continue;
                    uint32_t l2y;
                    const ps_cell cell = ps_cell_events(A, B, site, i, a.gen, a.k0, a.k1, pl, l2y);
                    if (DO_MUT && cell.mut) ps_set_byte(d, k, cell.mut);
                    if (DO_HR && cell.hr) hm |= 1u << k;
                }
            }
            if (DO_HR) {
                *(uint4 *)(rowS + 16u * c) = make_uint4(d[0], d[1], d[2], d[3]);
                hrm[c] = (uint16_t)hm;
            } else {
                *(uint4 *)(grow + 16u * c) = make_uint4(d[0], d[1], d[2], d[3]);
            }
        }
        if (DO_HR) {
            __syncthreads();   // rowS now holds the post-mutation snapshot of the row
            for (uint32_t c = lane; c < a.cpr; c += lpr) {
                const uint4 v = *(const uint4 *)(rowS + 16u * c);
                uint32_t d[4] = { v.x, v.y, v.z, v.w };
                uint32_t hm = hrm[c];
                while (hm) {
                    const uint32_t k = __builtin_ctz(hm);
                    hm &= hm - 1u;
                    const uint32_t i = c * 16u + k;
                    const ps_u4 l2 = ps_philox(site, i, a.gen, PS_STREAM_CORE_L2, a.k0, a.k1);
                    uint32_t donor = ps_mulhi(l2.y, a.N - 1u);
                    donor += (donor >= i) ? 1u : 0u;          // population.rs:618
                    ps_set_byte(d, k, (uint32_t)rowS[donor]); // snapshot read, :693-695
                }
                *(uint4 *)(grow + 16u * c) = make_uint4(d[0], d[1], d[2], d[3]);
            }
            if (!DO_GATHER) __syncthreads();   // rowS is rewritten by the next row
        }
    }
}

// ---------------------------------------------------------------------------
// Wave-per-row sweep (pitch <= 1024: one wavefront holds a whole site row, 16 cells
// per lane).  No workgroup barriers: the four waves of a block work on their own
// rows and synchronise only through the in-order LDS queue of their own wave.
//
// A wave takes the 4 site rows of one GLOBAL site group (sites 4g .. 4g + 3: the unit the level-1 blocks are keyed by,
// ps_common.h) per iteration; their loads are in flight together and the 16 parent indices of a lane are shared by all
// rows.  Events are sparse (about 5 % of the cells at the default rates), so
//   1. per batch, THREE Philox calls per lane give the six symbol planes of its 4 x 16 cells; per row pair three boolean
//      operations on the plane words give the DECIDED cells (the symbol names an allele) and the RESIDUAL ones (no
//      per-cell arithmetic at all: ps_classes);
//   2. the symbol-decided mutations -- three of four events -- are applied in registers to the gathered child dwords
//      before they go back to LDS (ps_apply_prepare / ps_apply_dword: byte-lane flags, two v_perm_b32 look-ups, one v_bfi);
//   3. one prefix sum over the wave pushes the residual cells of the 4 rows into a wave-private LDS queue of 16-bit
//      entries (ps_push_scan);
//   4. an exact pass over the residual cells: level-2 Philox, 32-bit thresholds, mutation bytes into the LDS rows; HR
//      donors are read from the post-mutation rows and written back after all reads.
// The host only selects this kernel when the queue cannot overflow in practice (mean + 10 sigma of the entry count
// fits); a full queue sends the batch to the queue-free method, nothing is dropped.
// ---------------------------------------------------------------------------
#define PS_BATCH_ROWS 4u     // site rows per wave iteration = sites per level-1 block group

__device__ __forceinline__ void ps_wave_sync()
{
    // LDS operations of one wave complete in issue order; this only stops the
    // compiler from moving LDS accesses across the phase boundary.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t ps_lane_prefix(uint64_t bal)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
}

// wave-wide inclusive prefix sum on the DPP path: four shifts inside the 16-lane rows, then lane 15 of a row into the next
// row (rows 1 and 3) and lane 31 into rows 2 and 3.  Lanes without a source keep the 0 of `old` (bound_ctrl off).
__device__ __forceinline__ uint32_t ps_wave_scan_incl(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);     // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);     // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);     // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);     // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);     // row_bcast:15
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);     // row_bcast:31
    return x;
}

// Class words of a row pair for the plans the queued sweeps take (k <= 1, 3k + R =: nE <= 8): every symbol that can
// hold an event has n <= 1, so the class of a cell is a function of the 3-bit number s3 = (plane 0, plane 5, plane 4):
//   ev  = s3 < nE  -- ONE v_bitop3_b32 whose truth table is the constant (1 << nE) - 1 in the order (P0, P5, P4) = (a, b, c),
//                     picked by a wave-uniform switch (a truth table is an immediate)
//   dec = s3 < 3k  -- k = 1: plane 0 clear and planes 4, 5 not both set; the allele index is then (plane 5, plane 4)
//   res = ev & ~dec
struct ps_class_words { uint32_t dec, res; };
__device__ __forceinline__ ps_class_words ps_classes(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t p4, uint32_t p5,
                                                     uint32_t k, uint32_t nE, uint32_t vm)
{
    const uint32_t c1 = __builtin_amdgcn_bitop3_b32(p1, p2, p3, 0x01) & vm;       // ~(p1 | p2 | p3): n <= 1
    // bitop3 truth table: bit (a << 2 | b << 1 | c) of the immediate is the result for inputs (a, b, c); with a = P0,
    // b = P5, c = P4 that index is s3
    uint32_t lt;
    switch (nE) {
    case 1: lt = __builtin_amdgcn_bitop3_b32(p0, p5, p4, 0x01); break;
    case 2: lt = __builtin_amdgcn_bitop3_b32(p0, p5, p4, 0x03); break;
    case 3: lt = __builtin_amdgcn_bitop3_b32(p0, p5, p4, 0x07); break;
    case 4: lt = __builtin_amdgcn_bitop3_b32(p0, p5, p4, 0x0F); break;
    case 5: lt = __builtin_amdgcn_bitop3_b32(p0, p5, p4, 0x1F); break;
    case 6: lt = __builtin_amdgcn_bitop3_b32(p0, p5, p4, 0x3F); break;
    case 7: lt = __builtin_amdgcn_bitop3_b32(p0, p5, p4, 0x7F); break;
    default: lt = 0xFFFFFFFFu; break;
    }
    ps_class_words o;
    o.dec = k ? c1 & __builtin_amdgcn_bitop3_b32(p0, p5, p4, 0x07) : 0u;
    o.res = c1 & lt & ~o.dec;
    return o;
}

// Symbol-decided mutations in registers.  dec / p4 / p5 are plane words of one row pair (p4, p5 already masked by dec):
// for the dword of cells 4j .. 4j + 3 at site parity h the three flags of a cell sit at bit s = j + 4h of its byte in
// each word.  Five merged words put (p4, p5, dec) of a slot side by side -- value 4 + allele index in the mutated bytes,
// 0 elsewhere -- so that per dword one shift and one mask give a byte-wise selector, two v_perm_b32 look up the allele
// and the byte mask, and one v_bfi_b32 writes the alleles into the child dword: no per-cell work, no queue.
struct ps_apply_words { uint32_t y0, y1, ya, yb, yc; };
__device__ __forceinline__ uint32_t ps_bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
__device__ __forceinline__ ps_apply_words ps_apply_prepare(uint32_t dec, uint32_t p4, uint32_t p5)
{
    const uint32_t r1 = p5 >> 1, r2 = p4 >> 2, l1 = p5 << 1, l2 = dec << 2;
    ps_apply_words o;
    o.y0 = ps_bfi(0x01010101u, p4, ps_bfi(0x02020202u, l1, l2));      // slot 0: bits 0-2
    o.y1 = ps_bfi(0x02020202u, p4, ps_bfi(0x04040404u, l1, l2));      // slot 1: bits 1-3
    o.ya = ps_bfi(0x09090909u, r2, ps_bfi(0x12121212u, r1, dec));     // slots 2, 5: bits 0-2, 3-5
    o.yb = ps_bfi(0x12121212u, r2, ps_bfi(0x24242424u, r1, dec));     // slots 3, 6: bits 1-3, 4-6
    o.yc = ps_bfi(0x24242424u, r2, ps_bfi(0x48484848u, r1, dec));     // slots 4, 7: bits 2-4, 5-7
    return o;
}
template <uint32_t SLOT>
__device__ __forceinline__ uint32_t ps_apply_dword(const ps_apply_words &y, uint32_t d)
{
    constexpr uint32_t sh = SLOT == 0u ? 0u : SLOT == 1u ? 1u : SLOT == 2u ? 0u : SLOT == 3u ? 1u : SLOT == 4u ? 2u : SLOT == 5u ? 3u : SLOT == 6u ? 4u : 5u;
    const uint32_t src = SLOT == 0u ? y.y0 : SLOT == 1u ? y.y1 : (SLOT == 2u || SLOT == 5u) ? y.ya : (SLOT == 3u || SLOT == 6u) ? y.yb : y.yc;
    const uint32_t f = (src >> sh) & 0x07070707u;                          // 4 + allele index in the mutated bytes, else 0
    const uint32_t allele = __builtin_amdgcn_perm(0x00080402u, 0u, f);     // selector 4, 5, 6 -> 2, 4, 8; 0 -> 0
    const uint32_t mask = __builtin_amdgcn_perm(0x00FFFFFFu, 0u, f);       // ... -> 0xFF; 0 -> 0
    return ps_bfi(mask, allele, d);
}
// all four dwords of the child row at site parity H of the pair
template <uint32_t H>
__device__ __forceinline__ void ps_apply_row(const ps_apply_words &y, uint4 &d)
{
    d.x = ps_apply_dword<4u * H + 0u>(y, d.x);
    d.y = ps_apply_dword<4u * H + 1u>(y, d.y);
    d.z = ps_apply_dword<4u * H + 2u>(y, d.z);
    d.w = ps_apply_dword<4u * H + 3u>(y, d.w);
}

// Residual push (phase 2 of the wave and window sweeps).  w0 / w1 are the residual words of the batch's row pairs (plane
// layout).  A lane counts its residual cells, ONE prefix sum over the wave gives every lane its own stretch of the queue,
// and the lane writes its entries there in a loop of its own.  Nothing is written when the batch does not fit (the caller
// redoes it queue-free).  Entry (12 bits): bit position p (0-31) | row pair << 5 | lane << 6; returns the wave-uniform
// number of entries of the batch.
__device__ __forceinline__ uint32_t ps_push_scan(uint32_t w0, uint32_t w1, uint16_t *q, uint32_t lane, uint32_t qcap)
{
    const uint32_t c0 = (uint32_t)__popc(w0), c = c0 + (uint32_t)__popc(w1);
    const uint32_t incl = ps_wave_scan_incl(c);
    const uint32_t qn = __builtin_amdgcn_readlane(incl, 63);
    if (qn <= qcap) {
        // (pre-increment and a start of its own per loop: the store takes the new address and no pointer lives across the loops,
        // i.e. no register copy per trip)
        uint16_t *qp = q + (incl - c) - 1, *qp1 = qp + c0;
        const uint32_t tag = lane << 6;
        for (uint32_t m = w0; m; m &= m - 1u) *++qp = (uint16_t)(tag | (uint32_t)__builtin_ctz(m));
        for (uint32_t m = w1; m; m &= m - 1u) *++qp1 = (uint16_t)(tag | 32u | (uint32_t)__builtin_ctz(m));
    }
    return qn;
}
// ... and the cell such an entry names: (cell of the row | row << 10), cell = 16 * lane + 4 (s % 4) + b for p = 8b + s,
// row = 2 * pair + s / 4
__device__ __forceinline__ uint32_t ps_entry_cell(uint32_t ent)
{
    const uint32_t cell = ((ent & 3u) << 2) | ((ent >> 3) & 3u), row = ((ent >> 2) & 1u) | ((ent >> 4) & 2u);
    return (row << 10) | ((ent >> 2) & 0x3F0u) | cell;
}

// four zero-extended bytes -> one dword (two v_perm + v_or; the compiler's own form masks every byte again)
__device__ __forceinline__ uint32_t ps_pack4(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3)
{
    return __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u) | __builtin_amdgcn_perm(b3, b2, 0x04000c0cu);
}

#ifdef PS_STAMP
// (asm volatile with a memory clobber: the compiler may not move the stamp across LDS / global accesses; pure
// VALU work can still drift across it, so the split between adjacent compute phases stays indicative)
#define PS_T(k) do { unsigned long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_acc[k] += t_ - st_last; st_last = t_; } while (0)
#else
#define PS_T(k) do { } while (0)
#endif
#ifndef PS_WAVE_LB
#define PS_WAVE_LB 6      // waves per SIMD the wave sweep is built for (8 = 64 VGPRs, 7 = 72, 6 = 80)
#endif
#ifndef PS_WAVE_PREFETCH
#define PS_WAVE_PREFETCH 0   // 1: the rows of a wave's next batch are requested behind the push of the current one (16 more registers)
#endif
// LDS of one wave: 4 child rows and the queue (a.qcap 16-bit entries)
__host__ __device__ constexpr uint32_t ps_queue_bytes(uint32_t qcap) { return (qcap * 2u + 15u) & ~15u; }
__host__ __device__ constexpr uint32_t ps_wave_lds(uint32_t qcap) { return PS_BATCH_ROWS * 1024u + ps_queue_bytes(qcap); }

template <bool DO_GATHER, bool DO_MUT, bool DO_HR, bool NT = false>
__global__ void __launch_bounds__(256, PS_WAVE_LB) core_sweep_wave_kernel(core_sweep_args a)
{
#ifdef PS_STAMP
    unsigned long long st_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory");
#endif
#ifdef PS_GROUP_TIMES      // (diagnostic builds: when does each of the 8 row groups start and end?  100 MHz clock; stamps 8.. / 16..)
    const unsigned long long gt_start = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr uint32_t PS_ROWS = PS_BATCH_ROWS;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // LDS rows have a fixed 1024-byte stride, so the low 12 bits of a queue entry
    // (cell | row << 10) are the byte address of the cell inside rowbuf
    uint8_t *rowbuf = lds + wave * ps_wave_lds(a.qcap);
    uint16_t *q = (uint16_t *)(rowbuf + PS_ROWS * 1024u);
    const ps_core_plan pl = a.plan;
    const uint32_t nE = 3u * pl.k + pl.R;             // symbols below nE can hold an event (<= 8 here)
    // (the host launches the mutate / HR variants only for plans that have events: the per-row code below
    // is straight-line -- no wave-uniform branches around the Philox calls, the LDS stores or the tail rows)
    constexpr bool events = DO_MUT || DO_HR;
    const bool has_chunk = lane < a.cpr;
    const uint32_t i0 = lane * 16u;
    const uint32_t ld_off = has_chunk ? i0 : 0u;      // lanes past the row load its first bytes; nothing of theirs is stored
    const uint32_t nvalid = (i0 >= a.N) ? 0u : min(16u, a.N - i0);
    const uint32_t vm = ps_valid_word(nvalid);        // the lane's cells that exist, at both sites of a plane word

    uint32_t pidx[16];
    if (DO_GATHER) {
        // cells beyond N gather the first padding byte of the row, which is always 0
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) pidx[k] = (k < nvalid) ? a.idx[i0 + k] : min(a.N, a.pitch - 1u);
    }

    // Dynamic row assignment.  The batches (global site groups: batch b = sites 4 * (site_offset / 4 + b) .. + 3, of which
    // the first and the last may lie partly outside this shard) are split into 8 contiguous ranges, one per group of
    // workgroups that share blockIdx.x % 8 (observed to share an XCD: affinity only); inside a
    // range every wave grabs chunks of PS_CHUNK row batches from the range's atomic counter.
    // A workgroup that starts late (another kernel held its CU) simply takes fewer chunks, so
    // the kernel has no static tail.  The next grab is issued before the current chunk is
    // processed; its latency is hidden behind the chunk's work.
    constexpr uint32_t PS_CHUNK = 3u;
    const uint32_t grp = blockIdx.x & 7u;
    const uint32_t off = a.site_offset & 3u, g0 = a.site_offset >> 2;
    const uint32_t batches = (a.rows + off + PS_ROWS - 1u) / PS_ROWS;
    const uint32_t b_lo = (uint32_t)((uint64_t)batches * grp / 8u), b_hi = (uint32_t)((uint64_t)batches * (grp + 1u) / 8u);
    // two counter sets alternate between launches; this launch zeroes the set of the next one
    // (the previous launch, which used that set, has completed: same stream)
    uint32_t *ctr = a.work_ctr + (a.launch_parity * 8u + grp) * 32u;     // counters 128 bytes apart
    if (blockIdx.x < 8u && threadIdx.x == 0) a.work_ctr[((a.launch_parity ^ 1u) * 8u + blockIdx.x) * 32u] = 0u;
    uint32_t next_chunk = 0;
    if (lane == 0) next_chunk = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if PS_WAVE_PREFETCH
    uint4 vn[PS_ROWS];                  // the next batch's rows, in flight
    uint32_t pf_batch = 0xFFFFFFFFu;    // ... and which batch they belong to
#pragma unroll
    for (uint32_t rr = 0; rr < PS_ROWS; rr++) vn[rr] = make_uint4(0, 0, 0, 0);
#endif
    for (;;) {
        const uint32_t chunk = __builtin_amdgcn_readfirstlane(next_chunk);
        if (b_lo + chunk * PS_CHUNK >= b_hi) break;
        if (lane == 0) next_chunk = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t cb = 0; cb < PS_CHUNK; cb++) {
        const uint32_t batch = b_lo + chunk * PS_CHUNK + cb;
        if (batch >= b_hi) break;
        // local row of the batch's first site (negative in a shard that starts inside a site group); rows outside the
        // shard are processed as copies of its first / last row and never stored
        const int lr0 = (int)(batch * PS_ROWS) - (int)off;
        const uint32_t sg = g0 + batch;                 // global site group: sites 4 sg .. 4 sg + 3
        auto lrow = [&](uint32_t rr) -> uint32_t { return (uint32_t)min(max(lr0 + (int)rr, 0), (int)a.rows - 1); };
        uint4 v[PS_ROWS];
#if PS_WAVE_PREFETCH
        // (rows of this batch were requested while the previous batch's residual cells were worked on)
        if (pf_batch == batch) {
#pragma unroll
            for (uint32_t rr = 0; rr < PS_ROWS; rr++) v[rr] = vn[rr];
        } else
#endif
#pragma unroll
        for (uint32_t rr = 0; rr < PS_ROWS; rr++)
            v[rr] = ps_load_row16(a.state + (size_t)lrow(rr) * a.pitch + ld_off, NT);
        if (DO_GATHER) {
#pragma unroll
            for (uint32_t rr = 0; rr < PS_ROWS; rr++) *(uint4 *)(rowbuf + rr * 1024u + i0) = v[rr];
            ps_wave_sync();
        }
        PS_T(0);   // global load + LDS stage

        // Phase 1: the symbol planes of the batch (three Philox calls per lane for 4 x 16 cells), then, row by row, the child
        // bytes.  (Interleaving the rows' gathers needs more registers than 8 waves per SIMD leave.)
        uint32_t wl[2] = { 0u, 0u };      // the residual words of ps_push_scan, one per row pair
        ps_apply_words y[2] = {};
        if (events) {
            const ps_u4 A0 = ps_philox_l1a(2u * sg, lane, a.gen, a.k0, a.k1);
            const ps_u4 A1 = ps_philox_l1a(2u * sg + 1u, lane, a.gen, a.k0, a.k1);
            const ps_u4 B = ps_philox_l1b(sg, lane, a.gen, a.k0, a.k1);
            const ps_class_words c0 = ps_classes(A0.x, A0.y, A0.z, A0.w, B.x, B.z, pl.k, nE, vm);
            const ps_class_words c1 = ps_classes(A1.x, A1.y, A1.z, A1.w, B.y, B.w, pl.k, nE, vm);
            wl[0] = c0.res;
            wl[1] = c1.res;
            if (DO_MUT) {
                y[0] = ps_apply_prepare(c0.dec, B.x & c0.dec, B.z & c0.dec);
                y[1] = ps_apply_prepare(c1.dec, B.y & c1.dec, B.w & c1.dec);
            }
        }
        PS_T(2);   // level-1 Philox + class words
#pragma unroll
        for (uint32_t rr = 0; rr < PS_ROWS; rr++) {
            uint8_t *row = rowbuf + rr * 1024u;
            uint4 d = v[rr];
            if (DO_GATHER) {
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    w[j] = ps_pack4(row[pidx[4 * j]], row[pidx[4 * j + 1]], row[pidx[4 * j + 2]], row[pidx[4 * j + 3]]);
                d = make_uint4(w[0], w[1], w[2], w[3]);
            }
            // the mutations the symbols decide, in registers (population.rs:511-540)
            if (events && DO_MUT) {
                if (rr & 1u) ps_apply_row<1u>(y[rr >> 1], d);
                else ps_apply_row<0u>(y[rr >> 1], d);
            }
            // the LDS row becomes the child row; every gather read precedes this store
            if (DO_GATHER) ps_wave_sync();
            if (DO_GATHER || events) *(uint4 *)(row + i0) = d;
            __builtin_amdgcn_sched_barrier(0);      // keep the rows apart (see above)
        }
        PS_T(1);   // gather + symbol-decided mutations
        const uint32_t qcap = a.qcap_limit ? min(a.qcap, a.qcap_limit) : a.qcap;
        uint32_t qn = 0;        // wave-uniform number of queued residual cells
        // Phase 2: every residual cell of the batch into the wave queue (one prefix sum, then lane-private writes)
        if (events) qn = ps_push_scan(wl[0], wl[1], q, lane, qcap);
        PS_T(3);   // queue push
        ps_wave_sync();
#if PS_WAVE_PREFETCH
        {
            // the rows of the wave's NEXT batch -- the next one of this chunk, or the first one of the chunk it has already
            // been handed -- requested now, in flight behind the exact pass and the stores of this batch
            uint32_t nb = batch + 1u;
            if (cb + 1u >= PS_CHUNK || nb >= b_hi) nb = b_lo + (uint32_t)__builtin_amdgcn_readfirstlane(next_chunk) * PS_CHUNK;
            pf_batch = nb < b_hi ? nb : 0xFFFFFFFFu;
            if (nb < b_hi) {
                const int nlr0 = (int)(nb * PS_ROWS) - (int)off;
#pragma unroll
                for (uint32_t rr = 0; rr < PS_ROWS; rr++)
                    vn[rr] = ps_load_row16(a.state + (size_t)(uint32_t)min(max(nlr0 + (int)rr, 0), (int)a.rows - 1) * a.pitch + ld_off, NT);
            }
        }
#endif
        // (HR parks a second 16-bit word per residual cell in the queue's upper half, mirrored: entry e at qcap - 1 - e)
        const bool redo = events && (qn > qcap || (DO_HR && 2u * qn > qcap));       // wave-uniform
        // Queue overflow (the host sizes the queue for mean + 10 sigma of the entry count, so this is a
        // once-in-the-age-of-the-universe event at the rates it admits -- and every batch under the test hook
        // `sweep_queue_cap`): nothing is dropped; the batch is redone by the queue-free method of
        // core_sweep_inline_kernel, every candidate handled by its owner lane (the cells whose symbol decides an allele
        // have it already and get it again).
        if (redo) {
#pragma unroll 1
            for (uint32_t rr = 0; rr < PS_ROWS; rr++) {
                uint8_t *row = rowbuf + rr * 1024u;
                const uint32_t site = 4u * sg + rr;
                const ps_u4 Ar = ps_philox_l1a(site >> 1, lane, a.gen, a.k0, a.k1);
                const ps_u4 Br = ps_philox_l1b(sg, lane, a.gen, a.k0, a.k1);
                uint32_t cmr = ps_cand_word(Ar, site, pl.cshift) & vm;
                uint32_t hm = 0;
                while (cmr) {
                    const uint32_t k = ps_bit_cell(__builtin_ctz(cmr)), cellidx = i0 + k;
                    cmr &= cmr - 1u;
                    uint32_t l2y = 0;
                    const ps_cell cell = ps_cell_events(Ar, Br, site, cellidx, a.gen, a.k0, a.k1, pl, l2y);
                    if (DO_MUT && cell.mut) row[cellidx] = (uint8_t)cell.mut;
                    if (DO_HR && cell.hr) hm |= 1u << k;
                }
                if (DO_HR) {
                    ps_wave_sync();       // the row is the post-mutation snapshot (population.rs:693-695)
                    uint32_t dv[4] = { 0u, 0u, 0u, 0u };
                    for (uint32_t tmp = hm; tmp;) {
                        const uint32_t k = __builtin_ctz(tmp), cellidx = i0 + k;
                        tmp &= tmp - 1u;
                        const ps_u4 l2 = ps_philox(site, cellidx, a.gen, PS_STREAM_CORE_L2, a.k0, a.k1);
                        uint32_t donor = ps_mulhi(l2.y, a.N - 1u);
                        donor += (donor >= cellidx) ? 1u : 0u;                       // population.rs:618
                        ps_set_byte(dv, k, (uint32_t)row[donor]);
                    }
                    ps_wave_sync();       // all donor reads are done
                    for (uint32_t tmp = hm; tmp;) {
                        const uint32_t k = __builtin_ctz(tmp);
                        tmp &= tmp - 1u;
                        row[i0 + k] = (uint8_t)((dv[k >> 2] >> (8u * (k & 3u))) & 0xFFu);
                    }
                }
            }
            ps_wave_sync();
        } else if (events) {
            // exact pass over the residual cells: level-2 Philox, 32-bit thresholds
            uint16_t *qh = q + qcap - 1u;       // HR: donor | 0x8000, later the donor's allele | 0x8000, of entry e at qh[-e]
            for (uint32_t base = 0; base < qn; base += 64u) {
                const uint32_t e = base + lane;
                if (e < qn) {
                    const uint32_t ent = ps_entry_cell(q[e]);      // cell | row << 10 = the cell's byte address in rowbuf
                    const uint32_t cellidx = ent & 1023u, rr = ent >> 10;
                    const ps_u4 l2 = ps_philox(4u * sg + rr, cellidx, a.gen, PS_STREAM_CORE_L2, a.k0, a.k1);
                    const ps_cell cell = ps_classify(l2.x, pl);
                    if (DO_MUT && cell.mut) rowbuf[ent] = (uint8_t)cell.mut;
                    if (DO_HR) {
                        uint32_t out = 0;
                        if (cell.hr) {
                            uint32_t donor = ps_mulhi(l2.y, a.N - 1u);
                            donor += (donor >= cellidx) ? 1u : 0u;       // population.rs:618
                            out = donor | 0x8000u;
                        }
                        *(qh - e) = (uint16_t)out;
                    }
                }
            }
            if (DO_HR) {
                ps_wave_sync();   // the LDS rows are now the post-mutation snapshot (population.rs:693-695)
                for (uint32_t base = 0; base < qn; base += 64u) {
                    const uint32_t e = base + lane;
                    if (e < qn) {
                        const uint32_t h = *(qh - e);
                        if (h >> 15) *(qh - e) = (uint16_t)((uint32_t)rowbuf[(ps_entry_cell(q[e]) & 3072u) | (h & 1023u)] | 0x8000u);
                    }
                }
                ps_wave_sync();   // all donor reads are done; now apply the copies
                for (uint32_t base = 0; base < qn; base += 64u) {
                    const uint32_t e = base + lane;
                    if (e < qn) {
                        const uint32_t h = *(qh - e);
                        if (h >> 15) rowbuf[ps_entry_cell(q[e])] = (uint8_t)(h & 0xFFu);
                    }
                }
            }
            ps_wave_sync();
        }
        PS_T(5);   // exact pass + HR

        if (has_chunk) {
#pragma unroll
            for (uint32_t rr = 0; rr < PS_ROWS; rr++) {
                const int lr = lr0 + (int)rr;
                if (lr >= 0 && lr < (int)a.rows) {
                    const uint4 o = (DO_GATHER || events) ? *(const uint4 *)(rowbuf + rr * 1024u + i0) : v[rr];
                    ps_store_row16(a.out + (size_t)lr * a.pitch + i0, o, NT);
                }
            }
        }
        ps_wave_sync();   // the next iteration overwrites the LDS rows and the queue
        PS_T(6);   // LDS read-back + global store
    }
    }
#ifdef PS_STAMP
    if (lane == 0)
        for (int k = 0; k < 8; k++) atomicAdd((unsigned long long *)a.stamps + k, st_acc[k]);
#endif
#ifdef PS_GROUP_TIMES
    if (lane == 0) {
        atomicMax((unsigned long long *)a.stamps + 8 + grp, (unsigned long long)__builtin_amdgcn_s_memrealtime());
        atomicMax((unsigned long long *)a.stamps + 16 + grp, gt_start);
        if (blockIdx.x < 8 && wave == 0) ((unsigned long long *)a.stamps)[24 + grp] = gt_start;     // the group's first workgroup
    }
#endif
}

// ---------------------------------------------------------------------------
// Window sweep: the wave-per-row design for populations WIDER than one wavefront (N > 1024), available when the
// children are stored in ascending parent order (ps_sim sorts the parents it draws: DESIGN.md 3.5).  A wave owns one
// 1024-child SEGMENT for the whole launch; sorted parents mean the segment's parents lie in one window
// [idx[first child], idx[last child]] of the parent row -- ~1024 + a few dozen bytes under drift, whatever N is -- so
// the gather needs a ~1.1 KB window in wave-private LDS instead of the whole row (64 KB at N = 65536) in
// workgroup-shared LDS: no workgroup barriers, 24 waves per CU instead of 16, the wave sweep's instruction stream.
//   * the new generation goes to a second buffer (the windows of neighbouring segments overlap, and HR donors are
//     read from the old generation), i.e. the sweep is out of place;
//   * HR: the donor of a cell may sit in any segment.  Its post-mutation value is RECOMPUTED instead of read from a
//     shared snapshot: parent byte old[row][idx[donor]] (two dependent global loads, L2 / Infinity Cache hits: the
//     row is being streamed by the other waves) and the donor's own level-1 / level-2 words (the
//     mutation of a cell is a pure function of (seed, generation, site, individual));
//   * a window wider than WCAP (parents far apart: strong selection against a stretch of the population) is staged
//     and gathered in pieces.
// Everything from the candidate queue on is the wave sweep's code (same entry formats; child rows at a 1024-byte
// stride in LDS, the window buffer beside them).
// ---------------------------------------------------------------------------
#ifndef PS_WCAP
#define PS_WCAP 1216u     // bytes of parent window staged per row (under drift a window spans 1024 + 15 +- 32 parents: 5.5 sigma)
#endif
#ifndef PS_WBPC
#define PS_WBPC 6         // workgroups per CU the kernel is launched at (6 x 23.3 KB of LDS at the default rates)
#endif
#ifndef PS_WLB
#define PS_WLB 6          // ... and the waves per SIMD it is BUILT for (80 VGPRs)
#endif
#ifndef PS_WINDOW_NT_LOADS
#define PS_WINDOW_NT_LOADS 0   // window sweep: 1 = the window loads nt as well; 0 = default cache policy (neighbouring windows share lines, HR donors read the rows), stores nt: 4.107 vs 4.142 ms
#endif
#define PS_WSTRIDE (PS_WCAP + 16u)   // row buffer stride in LDS: 16 zero bytes behind the window (what cells past N gather)
// LDS of one wave: 4 row buffers, the queue (a.qcap 16-bit entries) and the batch's HR list (64 x (donor, cell))
#define PS_WHR 64u
__host__ __device__ constexpr uint32_t ps_window_lds(uint32_t qcap) { return PS_BATCH_ROWS * PS_WSTRIDE + ps_queue_bytes(qcap) + PS_WHR * 6u; }

// post-mutation, pre-recombination value of cell (site row, individual donor), from the old generation
__device__ __forceinline__ uint32_t ps_donor_value(const core_sweep_args &a, const ps_core_plan &pl, const uint8_t *old_row,
                                                   uint32_t site, uint32_t donor, bool do_mut)
{
    uint32_t val = old_row[a.idx[donor]];                       // population.rs:450-465: the donor's gathered byte
    if (do_mut) {
        const ps_u4 A = ps_philox_l1a(site >> 1, donor >> 4, a.gen, a.k0, a.k1);
        const uint32_t n = ps_cell_nibble(A, site, donor);
        if ((n >> pl.cshift) == 0u) {
            const ps_u4 B = ps_philox_l1b(site >> 2, donor >> 4, a.gen, a.k0, a.k1);
            const uint32_t code = ps_sym_code(4u * n + ps_cell_pair(B, site, donor), pl);
            if (code > 1u) {
                val = code;                                      // population.rs:511-540
            } else if (code == 1u) {
                const ps_u4 l2 = ps_philox(site, donor, a.gen, PS_STREAM_CORE_L2, a.k0, a.k1);
                const ps_cell cell = ps_classify(l2.x, pl);
                if (cell.mut) val = cell.mut;
            }
        }
    }
    return val;
}


// WIDE = false: the segments whose window fits the row buffer (all of them under drift); WIDE = true: a second launch for
// the others (parents far apart: strong selection against a stretch of the population) -- same code, the bytes gathered
// straight from the old row in global memory; ascending parents keep every load instruction's 64 addresses in one compact
// range.  A wave whose segment belongs to the other launch leaves at once.
template <bool DO_MUT, bool DO_HR, bool NT, bool WIDE>
__global__ void __launch_bounds__(256, PS_WLB) core_sweep_window_kernel(core_sweep_args a)
{
    constexpr uint32_t PS_ROWS = PS_BATCH_ROWS;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // per wave: PS_ROWS row buffers of PS_WCAP bytes -- the parents' window arrives there by LDS-DMA, the child row
    // (1024 bytes) overwrites its start once the gather has read it -- and the candidate queue
    uint8_t *rowbuf = lds + wave * ps_window_lds(a.qcap);
    uint16_t *q = (uint16_t *)(rowbuf + PS_ROWS * PS_WSTRIDE);
    uint32_t *hr_d = (uint32_t *)(rowbuf + PS_ROWS * PS_WSTRIDE + ps_queue_bytes(a.qcap));     // HR list: donors ...
    uint16_t *hr_e = (uint16_t *)(hr_d + PS_WHR);                                              // ... and cells (cell | row << 10)
    if (lane < PS_ROWS) *(uint4 *)(rowbuf + lane * PS_WSTRIDE + PS_WCAP) = make_uint4(0, 0, 0, 0);     // the zero bytes (never rewritten)
    const ps_core_plan pl = a.plan;
    // the wave's segment, fixed for the launch
    const uint32_t segs = (a.N + 1023u) >> 10;
    // Workgroups with equal blockIdx.x % 8 share an XCD (observed; affinity only) and with it an L2: group g takes the
    // batches [batches g / 8, batches (g + 1) / 8) and ALL segments of them, so that a whole row passes through one L2 at
    // about the same time -- the donors' bytes of HR (anywhere in the row), the overlap of neighbouring windows and the
    // patched bytes then hit it.
    const uint32_t grp = blockIdx.x & 7u;
    const uint32_t wg = __builtin_amdgcn_readfirstlane((blockIdx.x >> 3) * 4u + wave);     // wave index inside its group
    const uint32_t sgm = wg % segs;
    // this launch zeroes the chunk counters of the next one (two sets alternate by launch parity) ...
    if (wg < segs && lane == 0) a.work_ctr[(((a.launch_parity ^ 1u) * 8u + grp) * segs + wg) * 32u] = 0u;
    // ... and the second launch (WIDE) leaves here unless the first one met a segment for it (one flag per generation
    // parity, cleared a generation ahead): under drift there is none, and 7168 waves each reading their window bounds and 16
    // parent indices just to find that out cost 31 us between two sweeps.  (Both launches of a generation go to the SAME
    // stream, the first before the second, and nothing in between touches slot gen & 1: launch_core_sweep_window.)
    if (WIDE) {
        if (__builtin_amdgcn_readfirstlane(a.wide_flags[(a.gen & 1u) * 32u]) == 0u) return;
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.wide_flags[((a.gen + 1u) & 1u) * 32u] = 0u;
    }
    const uint32_t c_first = sgm * 1024u, c_last = min(c_first + 1023u, a.N - 1u);
    const uint32_t chunk = sgm * 64u + lane;                // the lane's 16-cell chunk of the row (Philox counter word)
    const bool has_chunk = chunk < a.cpr;
    const uint32_t i0 = lane * 16u, c0 = c_first + i0;      // cell offset inside the segment / global index of the first cell
    const uint32_t nvalid = (c0 >= a.N) ? 0u : min(16u, a.N - c0);
    const uint32_t vm = ps_valid_word(nvalid);              // the lane's cells that exist, at both sites of a plane word
    const uint32_t nE = 3u * pl.k + pl.R;                   // symbols below nE can hold an event (<= 8 here)
    // the parents' window [w_lo, w_hi] of the parent row, staged from its 16-byte aligned start
    const uint32_t w_lo = __builtin_amdgcn_readfirstlane(a.idx[c_first]) & ~15u;
    const uint32_t w_hi = __builtin_amdgcn_readfirstlane(a.idx[c_last]);
    const uint32_t wbytes = w_hi - w_lo + 1u;
    const bool wide = wbytes > PS_WCAP;                     // wave-uniform
    uint32_t pidx[16];      // parents relative to the window start; cells past N gather a zero byte behind the window
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) pidx[k] = (k < nvalid) ? a.idx[c0 + k] - w_lo : PS_WCAP;
    const bool ld0 = !WIDE && i0 < wbytes, ld1 = !WIDE && 1024u + i0 < wbytes;
    // LDS address of a queue entry's cell: (cell | row << 10) -> row * PS_WSTRIDE + cell
    auto cell_addr = [](uint32_t ent) -> uint32_t { return (ent & 1023u) + ((ent >> 10) & 3u) * PS_WSTRIDE; };

    // batches = global site groups (see the wave sweep); the waves of a segment share the segment's chunk counter (two
    // counter sets alternate by launch parity; this launch zeroes the next one's)
    constexpr uint32_t PS_CHUNK = 3u;
    const uint32_t off = a.site_offset & 3u, g0 = a.site_offset >> 2;
    const uint32_t batches = (a.rows + off + PS_ROWS - 1u) / PS_ROWS;
    const uint32_t b_lo = (uint32_t)((uint64_t)batches * grp / 8u), b_hi = (uint32_t)((uint64_t)batches * (grp + 1u) / 8u);
    uint32_t *ctr = a.work_ctr + ((a.launch_parity * 8u + grp) * segs + sgm) * 32u;
    if (wide != WIDE) {                                     // the segment belongs to the other launch (no workgroup barriers here)
        if (!WIDE && lane == 0) a.wide_flags[(a.gen & 1u) * 32u] = 1u;
        return;
    }
    uint32_t next_chunk = 0;
    if (lane == 0) next_chunk = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        const uint32_t chk = __builtin_amdgcn_readfirstlane(next_chunk);
        if (b_lo + chk * PS_CHUNK >= b_hi) break;
        if (lane == 0) next_chunk = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t cb = 0; cb < PS_CHUNK; cb++) {
        const uint32_t batch = b_lo + chk * PS_CHUNK + cb;
        if (batch >= b_hi) break;
        const int lr0 = (int)(batch * PS_ROWS) - (int)off;  // local row of the batch's first site (see the wave sweep)
        const uint32_t sg = g0 + batch;                     // global site group
        auto lrow = [&](uint32_t rr) -> uint32_t { return (uint32_t)min(max(lr0 + (int)rr, 0), (int)a.rows - 1); };
        // the first window piece of every row of the batch by LDS-DMA: two 16-byte pieces per lane, the second only where
        // the window is longer than 1 KiB (rows outside the shard are processed as copies, never stored)
#pragma unroll
        for (uint32_t rr = 0; rr < PS_ROWS; rr++) {
            const uint8_t *src = a.state + (size_t)lrow(rr) * a.pitch + w_lo;
            if (ld0) ps_dma16(src + i0, rowbuf + rr * PS_WSTRIDE, NT && PS_WINDOW_NT_LOADS);
            if (ld1) ps_dma16(src + 1024u + i0, rowbuf + rr * PS_WSTRIDE + 1024u, NT && PS_WINDOW_NT_LOADS);
        }
        // the symbol planes of the batch while the windows are in flight (see the wave sweep)
        uint32_t wl0, wl1;
        ps_apply_words y[2] = {};
        {
            const ps_u4 A0 = ps_philox_l1a(2u * sg, chunk, a.gen, a.k0, a.k1);
            const ps_u4 A1 = ps_philox_l1a(2u * sg + 1u, chunk, a.gen, a.k0, a.k1);
            const ps_u4 B = ps_philox_l1b(sg, chunk, a.gen, a.k0, a.k1);
            const ps_class_words c0 = ps_classes(A0.x, A0.y, A0.z, A0.w, B.x, B.z, pl.k, nE, vm);
            const ps_class_words c1 = ps_classes(A1.x, A1.y, A1.z, A1.w, B.y, B.w, pl.k, nE, vm);
            wl0 = c0.res;
            wl1 = c1.res;
            if (DO_MUT) {
                y[0] = ps_apply_prepare(c0.dec, B.x & c0.dec, B.z & c0.dec);
                y[1] = ps_apply_prepare(c1.dec, B.y & c1.dec, B.w & c1.dec);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the compiler does not track LDS-DMA)
        ps_wave_sync();
#pragma unroll
        for (uint32_t rr = 0; rr < PS_ROWS; rr++) {
            uint8_t *win = rowbuf + rr * PS_WSTRIDE;
            uint32_t w[4] = { 0u, 0u, 0u, 0u };
            if (!WIDE) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    w[j] = ps_pack4(win[pidx[4 * j]], win[pidx[4 * j + 1]], win[pidx[4 * j + 2]], win[pidx[4 * j + 3]]);
            } else {
                const uint8_t *src = a.state + (size_t)lrow(rr) * a.pitch + w_lo;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t bsel[4];
#pragma unroll
                    for (int b = 0; b < 4; b++) bsel[b] = ((uint32_t)(4 * j + b) < nvalid) ? (uint32_t)src[pidx[4 * j + b]] : 0u;
                    w[j] = ps_pack4(bsel[0], bsel[1], bsel[2], bsel[3]);
                }
            }
            uint4 d = make_uint4(w[0], w[1], w[2], w[3]);
            // the mutations the symbols decide, in registers (population.rs:511-540)
            if (DO_MUT) {
                if (rr & 1u) ps_apply_row<1u>(y[rr >> 1], d);
                else ps_apply_row<0u>(y[rr >> 1], d);
            }
            ps_wave_sync();                                     // the row buffer becomes the child row: every gather read precedes
            *(uint4 *)(win + i0) = d;
            __builtin_amdgcn_sched_barrier(0);      // keep the rows apart (see the wave sweep)
        }
        // the push of the wave sweep: one prefix sum and lane-private writes
        const uint32_t qcap = a.qcap_limit ? min(a.qcap, a.qcap_limit) : a.qcap;
        const uint32_t qn = ps_push_scan(wl0, wl1, q, lane, qcap);
        ps_wave_sync();

        if (qn > qcap) {
            // full queue: the batch is redone by the queue-free method (see the wave sweep), donors recomputed
#pragma unroll 1
            for (uint32_t rr = 0; rr < PS_ROWS; rr++) {
                uint8_t *row = rowbuf + rr * PS_WSTRIDE;
                const uint32_t site = 4u * sg + rr;
                const uint8_t *old_row = a.state + (size_t)lrow(rr) * a.pitch;
                const ps_u4 Ar = ps_philox_l1a(site >> 1, chunk, a.gen, a.k0, a.k1);
                const ps_u4 Br = ps_philox_l1b(sg, chunk, a.gen, a.k0, a.k1);
                uint32_t cmr = ps_cand_word(Ar, site, pl.cshift) & vm;
                while (cmr) {
                    const uint32_t k = ps_bit_cell(__builtin_ctz(cmr)), cell = c0 + k;
                    cmr &= cmr - 1u;
                    uint32_t l2y = 0;
                    const ps_cell cl = ps_cell_events(Ar, Br, site, cell, a.gen, a.k0, a.k1, pl, l2y);
                    if (DO_MUT && cl.mut) row[i0 + k] = (uint8_t)cl.mut;
                    if (DO_HR && cl.hr) {
                        uint32_t donor = ps_mulhi(l2y, a.N - 1u);
                        donor += (donor >= cell) ? 1u : 0u;                      // population.rs:618
                        row[i0 + k] = (uint8_t)ps_donor_value(a, pl, old_row, site, donor, DO_MUT);
                    }
                }
            }
            ps_wave_sync();
        } else {
            // exact pass: level-2 Philox.  A cell that receives a donor allele recomputes the donor's post-mutation value
            // from the old generation (two dependent global loads and the donor's own level-1 / level-2 words) -- not inside
            // the pass, where a handful of lanes would run that code once per trip, but from a list the whole batch
            // shares: ONE trip of it per batch (until 64 such cells have met).  Measured in round 3 against parking such
            // cells in records and patching the stored rows a batch later: the patch stores and the later, cache-cold donor
            // reads cost more than the latency they hide.
            uint32_t nh = 0;        // wave-uniform length of the HR list
            auto hr_flush = [&]() {
                ps_wave_sync();
                if (lane < nh) {
                    const uint32_t ent = hr_e[lane], donor = hr_d[lane], rr = ent >> 10;
                    const uint32_t rg = (uint32_t)min(max(lr0 + (int)rr, 0), (int)a.rows - 1);
                    rowbuf[cell_addr(ent)] = (uint8_t)ps_donor_value(a, pl, a.state + (size_t)rg * a.pitch, 4u * sg + rr, donor, DO_MUT);
                }
                nh = 0;
                ps_wave_sync();
            };
            for (uint32_t base = 0; base < qn; base += 64u) {
                const uint32_t e = base + lane;
                bool hr = false;
                uint32_t ent = 0, donor = 0;
                if (e < qn) {
                    ent = ps_entry_cell(q[e]);      // cell | row << 10
                    const uint32_t rr = ent >> 10;
                    const uint32_t cell = c_first + (ent & 1023u);
                    const ps_u4 l2 = ps_philox(4u * sg + rr, cell, a.gen, PS_STREAM_CORE_L2, a.k0, a.k1);
                    const ps_cell cl = ps_classify(l2.x, pl);
                    if (DO_MUT && cl.mut) rowbuf[cell_addr(ent)] = (uint8_t)cl.mut;
                    if (DO_HR && cl.hr) {
                        hr = true;
                        donor = ps_mulhi(l2.y, a.N - 1u);
                        donor += (donor >= cell) ? 1u : 0u;                      // population.rs:618
                    }
                }
                if (DO_HR) {
                    const uint64_t bal = __builtin_amdgcn_ballot_w64(hr);
                    const uint32_t cnt = (uint32_t)__popcll(bal);
                    if (nh + cnt > PS_WHR) hr_flush();
                    if (hr) {
                        const uint32_t pos = nh + ps_lane_prefix(bal);
                        hr_e[pos] = (uint16_t)ent;
                        hr_d[pos] = donor;
                    }
                    nh += cnt;
                }
            }
            if (DO_HR && nh) hr_flush();
            ps_wave_sync();
        }

        if (has_chunk) {
#pragma unroll
            for (uint32_t rr = 0; rr < PS_ROWS; rr++) {
                const int lr = lr0 + (int)rr;
                if (lr >= 0 && lr < (int)a.rows) {
                    const uint4 o = *(const uint4 *)(rowbuf + rr * PS_WSTRIDE + i0);
                    ps_store_row16(a.out + (size_t)lr * a.pitch + c0, o, NT);
                }
            }
        }
        ps_wave_sync();   // the next iteration's DMA overwrites the row buffers
    }
    }
}

// ---------------------------------------------------------------------------
// Block sweep for rows wider than one wavefront (pitch > 1024: cfg4/cfg5 populations).
// A workgroup of nw waves stages R whole site rows in LDS (parent rows and child rows) and
// splits them into 1024-cell segments.  A wave takes PS_SB consecutive segments per
// iteration and treats them like the wave-per-row sweep treats its rows: the symbol planes of
// the slot (two Philox blocks per segment here: a slot is ONE site of a block pair's two / four),
// the symbol-decided mutations in registers, ONE wave-private queue of the batch's residual cells,
// one exact level-2 pass.
// A mutation only touches cells of the wave's own segments, so it needs no block barrier;
// cells that receive a donor allele are collected in a per-wave HR list, and the donor
// reads / writes happen between block barriers once every segment of the row group has
// been mutated (the donor may sit in any segment).
// Queue entry (residual cells only): LDS byte offset of the cell in rowS (20 bits) | slot << 28.
// ---------------------------------------------------------------------------
#define PS_PF 4u   // prefetch registers (uint4) per thread: pf0..pf3 in the kernel
// Workgroup barrier that orders LDS traffic only: outstanding global loads (the prefetch of the
// next row group) and stores (the finished row group) stay in flight across it.
__device__ __forceinline__ void ps_block_sync_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct core_block_geom {
    uint32_t R;        // rows per workgroup iteration
    uint32_t segs;     // 1024-cell segments per row
    uint32_t QW;       // candidate queue entries per wave (one batch of PS_SB segments)
    uint32_t HW;       // HR list entries per wave (one row group)
    uint32_t SB;       // segments per wave batch (template parameter PS_SB: 4, or 2 when LDS is short)
    uint32_t ovf_off;  // LDS byte offset of the workgroup's "a queue or an HR list was full" word (the last 4 bytes)
};

// plane-word mask (both sites) of the cells i0 .. i0+15 that exist (< N)
__device__ __forceinline__ uint32_t ps_valid_cells(uint32_t i0, uint32_t N)
{
    return ps_valid_word(i0 >= N ? 0u : min(16u, N - i0));
}

template <uint32_t PS_SB, bool PRE, bool DO_GATHER, bool DO_MUT, bool DO_HR>
__global__ void __launch_bounds__(1024) core_sweep_block_kernel(core_sweep_args a, core_block_geom g)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // (the wave index through readfirstlane: everything derived from it -- rows, segments, sites of the wave's
    // slots -- then lives in scalar registers instead of being recomputed with vector multiplies per slot)
    const uint32_t tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u, nw = blockDim.x >> 6;
    uint8_t *rowS = lds;                                   // child rows [R][pitch]
    uint8_t *rowA = lds + (DO_GATHER ? g.R * a.pitch : 0); // parent rows [R][pitch] (gather only)
    uint32_t *wbase = (uint32_t *)(lds + (DO_GATHER ? 2u : 1u) * g.R * a.pitch) + wave * (g.QW + 2u * g.HW);
    uint32_t *q = wbase, *hr_a = wbase + g.QW, *hr_b = hr_a + g.HW;
#ifdef PS_STAMP
    unsigned long long st_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory");
#endif
    const ps_core_plan pl = a.plan;
    constexpr bool events = DO_MUT || DO_HR;      // (the host launches these variants only for plans with events)
    uint32_t *ovf = (uint32_t *)(lds + g.ovf_off);
    // (the last entry of an HR list is never used: the last wave's holds the overflow word, see block_sweep_geometry)
    const uint32_t qcap = a.qcap_limit ? min(g.QW, a.qcap_limit) : g.QW, hcap = a.qcap_limit ? min(g.HW - 1u, a.qcap_limit) : g.HW - 1u;
    if (tid == 0) *ovf = 0u;        // (ordered before any use by the barrier that follows the first staging)
    const uint32_t nE = 3u * pl.k + pl.R;       // (the host launches this kernel for the plans of the wave sweep: cshift <= 1, k <= 1)
    // (row, segment) of the first item of this wave's first batch, and the batch-to-batch stride
    const uint32_t first_rr = (wave * PS_SB) / g.segs, first_sg = (wave * PS_SB) % g.segs;
    const uint32_t step_rr = (nw * PS_SB) / g.segs, step_sg = (nw * PS_SB) % g.segs;

    // PRE: every wave has exactly one batch per row group, i.e. the same PS_SB segments for the
    // whole launch, and all parent indices fit 16 bits: the wave keeps the 16 parents of each of
    // its lanes' cells in registers, two per VGPR (idxP[chunk][8], built by idx_pack16_kernel)
    uint32_t pid[PS_SB][8];
    uint32_t vperm_pre[PS_SB];      // PRE: valid-cell masks of the wave's (fixed) segments
    if (PRE && DO_GATHER) {
        uint32_t sg = first_sg;
#pragma unroll
        for (uint32_t s = 0; s < PS_SB; s++) {
            vperm_pre[s] = ps_valid_cells((sg * 64u + lane) * 16u, a.N);
            const uint32_t chunk = min(sg * 64u + lane, a.cpr - 1u);
            const uint4 lo = *(const uint4 *)(a.idxT + 8u * chunk), hi = *(const uint4 *)(a.idxT + 8u * chunk + 4u);
            pid[s][0] = lo.x; pid[s][1] = lo.y; pid[s][2] = lo.z; pid[s][3] = lo.w;
            pid[s][4] = hi.x; pid[s][5] = hi.y; pid[s][6] = hi.z; pid[s][7] = hi.w;
            if (++sg == g.segs) sg = 0;
        }
    }
    // Software pipeline over the row groups of this workgroup: the next group is fetched into
    // registers (PS_PF x 16 bytes per thread) while the current one is processed, and the stores
    // of the finished group are never waited for (the barriers order LDS traffic only).
    uint8_t *stage = DO_GATHER ? rowA : rowS;
    const uint32_t gstride = gridDim.x * g.R;
    const bool pipelined = g.R * a.pitch <= blockDim.x * 16u * PS_PF;
    uint4 pf0 = make_uint4(0, 0, 0, 0), pf1 = pf0, pf2 = pf0, pf3 = pf0;
    const uint32_t po0 = tid * 16u, po1 = po0 + blockDim.x * 16u, po2 = po1 + blockDim.x * 16u, po3 = po2 + blockDim.x * 16u;
    uint32_t r0 = blockIdx.x * g.R;
    if (r0 < a.rows) {
        const uint32_t nr = min(g.R, a.rows - r0);
        const uint8_t *src = a.state + (size_t)r0 * a.pitch;
        for (uint32_t o = tid * 16u; o < nr * a.pitch; o += blockDim.x * 16u)
            *(uint4 *)(stage + o) = *(const uint4 *)(src + o);
    }
    // every load issued so far (parent indices, first row group) has landed: tell the compiler's wait-count
    // model so, or it flushes vmcnt in the batch loop's preheader on EVERY row group -- right behind the
    // prefetch loads of the next group, which then stop being a prefetch
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    ps_block_sync_lds();
    for (; r0 < a.rows; r0 += gstride) {
        const uint32_t nr = min(g.R, a.rows - r0);
        const uint32_t rnext = r0 + gstride;
        const uint32_t nbytes_next = rnext < a.rows ? min(g.R, a.rows - rnext) * a.pitch : 0u;
        if (pipelined) {
            // unconditional loads (offsets past the group read its first bytes, never stored)
            const uint8_t *src = a.state + (nbytes_next ? (size_t)rnext * a.pitch : (size_t)0);
            pf0 = *(const uint4 *)(src + (po0 < nbytes_next ? po0 : 0u));
            pf1 = *(const uint4 *)(src + (po1 < nbytes_next ? po1 : 0u));
            pf2 = *(const uint4 *)(src + (po2 < nbytes_next ? po2 : 0u));
            pf3 = *(const uint4 *)(src + (po3 < nbytes_next ? po3 : 0u));
        }

        PS_T(0);   // loop control + prefetch issue
        uint32_t nhr = 0;       // wave-uniform length of this wave's HR list
        const uint32_t items = nr * g.segs;
        uint32_t rr0 = first_rr, sg0 = first_sg;
        for (uint32_t item0 = wave * PS_SB; item0 < items; item0 += nw * PS_SB) {
            uint32_t s_base[PS_SB], s_site[PS_SB];   // wave-uniform per slot: LDS offset of the row, site
            uint32_t s_seg[PS_SB];                   // LDS offset of the slot's segment
            uint32_t cmv[PS_SB];                     // candidate masks of the batch (one push loop for all slots)
            uint32_t qn = 0;
            uint32_t rr = rr0, sg = sg0;
            // FULL (wave-uniform): every slot of the batch exists and every lane of every segment holds a chunk
            // (N a multiple of 1024, whole row groups): no per-slot branches and no masked stores.  (Going further
            // -- all gathers of the batch, then all Philox chains, then all stores -- needs 128 VGPRs with spills
            // and was slower: 0.91 against 0.79 ms per 1.2e9 cells at N = 65536.)
            auto slot = [&](auto full_tag, uint32_t s) {
                constexpr bool FULL = decltype(full_tag)::value;
                s_base[s] = rr * a.pitch;
                s_seg[s] = rr * a.pitch + sg * 1024u;
                s_site[s] = a.site_offset + r0 + rr;
                cmv[s] = 0u;
                if (FULL || item0 + s < items) {
                    // lanes past the row (chunk >= cpr) compute on the row's last chunk and only their LDS store is
                    // masked; their candidate mask is empty (no valid cells)
                    const uint32_t chunk = sg * 64u + lane;
                    const bool has_chunk = FULL || chunk < a.cpr;
                    const uint32_t chunkc = FULL ? chunk : min(chunk, a.cpr - 1u);
                    const uint32_t i0 = chunk * 16u;
                    uint8_t *row = rowS + s_base[s];
                    uint32_t w[4] = { 0u, 0u, 0u, 0u };
                    if (DO_GATHER) {
                        const uint8_t *par = rowA + s_base[s];
                        if (PRE) {
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const uint32_t pk0 = pid[s][2 * j], pk1 = pid[s][2 * j + 1];
                                w[j] = ps_pack4(par[pk0 & 0xFFFFu], par[pk0 >> 16], par[pk1 & 0xFFFFu], par[pk1 >> 16]);
                            }
                        } else {
                            // idxT[k][chunk]: consecutive lanes read consecutive words; entries beyond N
                            // hold the index of the row's first padding byte (always 0)
                            const uint32_t *ip = a.idxT + chunkc;
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                w[j] = ps_pack4(par[ip[0]], par[ip[a.cpr]], par[ip[2u * a.cpr]], par[ip[3u * a.cpr]]);
                                ip += 4u * a.cpr;
                            }
                        }
                    } else if (events && DO_MUT) {
                        const uint4 cur = *(const uint4 *)(row + chunkc * 16u);      // the row itself (staged)
                        w[0] = cur.x; w[1] = cur.y; w[2] = cur.z; w[3] = cur.w;
                    }
                    if (events) {
                        // the slot is ONE site of a block pair's two / four: both blocks, this site's half of the words
                        const uint32_t vcell = FULL ? 0xFFFFFFFFu : PRE ? vperm_pre[s] : ps_valid_cells(i0, a.N);
                        const uint32_t site = s_site[s];
                        const ps_u4 A = ps_philox_l1a(site >> 1, chunk, a.gen, a.k0, a.k1);
                        const ps_u4 B = ps_philox_l1b(site >> 2, chunk, a.gen, a.k0, a.k1);
                        const uint32_t p4w = (site & 2u) ? B.y : B.x, p5w = (site & 2u) ? B.w : B.z;
                        const ps_class_words cw = ps_classes(A.x, A.y, A.z, A.w, p4w, p5w, pl.k, nE, vcell & ps_site_bits(site));
                        cmv[s] = cw.res;
                        if (DO_MUT) {
                            // the symbol-decided mutations in registers (see the wave sweep)
                            const ps_apply_words y = ps_apply_prepare(cw.dec, p4w & cw.dec, p5w & cw.dec);
                            uint4 d = make_uint4(w[0], w[1], w[2], w[3]);
                            if (site & 1u) ps_apply_row<1u>(y, d);
                            else ps_apply_row<0u>(y, d);
                            w[0] = d.x; w[1] = d.y; w[2] = d.z; w[3] = d.w;
                        }
                    }
                    if ((DO_GATHER || (events && DO_MUT)) && has_chunk) *(uint4 *)(row + i0) = make_uint4(w[0], w[1], w[2], w[3]);
                }
                if (++sg == g.segs) { sg = 0; rr++; }
            };
            if (item0 + PS_SB <= items && a.cpr == g.segs * 64u && a.N == a.pitch) {
#pragma unroll
                for (uint32_t s = 0; s < PS_SB; s++) slot(std::true_type{}, s);
            } else {
#pragma unroll
                for (uint32_t s = 0; s < PS_SB; s++) slot(std::false_type{}, s);
            }
            PS_T(1);   // gather + level-1 Philox + child store (all slots)
            if (events) {
                // ONE push loop for the batch: its trip count is the largest number of candidates any lane holds in
                // one segment, not the sum over the segments.
                for (;;) {
                    uint32_t any = cmv[0];
#pragma unroll
                    for (uint32_t s = 1; s < PS_SB; s++) any |= cmv[s];
                    if (__ballot(any != 0u) == 0ull) break;
#pragma unroll
                    for (uint32_t s = 0; s < PS_SB; s++) {
                        const bool act = cmv[s] != 0u;
                        const uint64_t bal = __builtin_amdgcn_ballot_w64(act);
                        if (act) {
                            const uint32_t p = __builtin_ctz(cmv[s]);
                            cmv[s] &= cmv[s] - 1u;
                            const uint32_t pos = qn + ps_lane_prefix(bal);
                            if (pos < g.QW) q[pos] = (s_seg[s] + lane * 16u + ps_bit_cell(p)) | (s << 28);
                        }
                        qn += (uint32_t)__popcll(bal);
                    }
                }
            }
            rr0 += step_rr;
            sg0 += step_sg;
            if (sg0 >= g.segs) { sg0 -= g.segs; rr0++; }
            PS_T(2);   // push loop
            if (!events) continue;
            ps_wave_sync();
            if (qn > qcap) {
                // nothing is dropped: the whole row group is redone below by the queue-free method
                if (lane == 0) *ovf = 1u;
                continue;
            }
            const uint32_t n2 = qn;      // (only residual cells are queued: the symbol-decided mutations were applied in registers)
            PS_T(3);
            // exact pass
            for (uint32_t base = 0; base < n2; base += 64u) {
                const uint32_t e = base + lane;
                bool hr = false;
                uint32_t off = 0, donor = 0, rbase = 0;
                if (e < n2) {
                    const uint32_t ent = q[e];
                    off = ent & 0xFFFFFu;
                    const uint32_t s = ent >> 28;
                    rbase = s_base[0];
                    uint32_t site = s_site[0];
#pragma unroll
                    for (uint32_t k = 1; k < PS_SB; k++) {
                        rbase = (s == k) ? s_base[k] : rbase;
                        site = (s == k) ? s_site[k] : site;
                    }
                    const uint32_t cellidx = off - rbase;
                    const ps_u4 l2 = ps_philox(site, cellidx, a.gen, PS_STREAM_CORE_L2, a.k0, a.k1);
                    const ps_cell cell = ps_classify(l2.x, pl);
                    if (DO_MUT && cell.mut) rowS[off] = (uint8_t)cell.mut;
                    if (DO_HR && cell.hr) {
                        hr = true;
                        donor = ps_mulhi(l2.y, a.N - 1u);
                        donor += (donor >= cellidx) ? 1u : 0u;           // population.rs:618
                    }
                }
                if (DO_HR) {
                    const uint64_t bal = __builtin_amdgcn_ballot_w64(hr);
                    if (hr) {
                        const uint32_t pos = nhr + ps_lane_prefix(bal);
                        if (pos < hcap) { hr_a[pos] = off; hr_b[pos] = rbase + donor; }
                    }
                    nhr += (uint32_t)__popcll(bal);
                }
            }
            ps_wave_sync();
        }
        PS_T(4);   // exact pass
        bool redo = false;
        if (events) {
            if (DO_HR && nhr > hcap && lane == 0) *ovf = 1u;
            ps_block_sync_lds();    // every segment is mutated: rowS is the snapshot (population.rs:693-695); *ovf is final
            redo = *ovf != 0u;      // workgroup-uniform
        }
        if (events && redo) {
            // A candidate queue or an HR list of this row group was full (the host sizes them for mean + 10 sigma, so
            // in production never; under the test hook `sweep_queue_cap` all the time).  Nothing is dropped: the row group
            // is redone by the queue-free method of core_sweep_inline_kernel -- every candidate handled by its owner
            // lane, HR cells remembered in a 16-bit mask per chunk (in the now dead queue memory), donors read from the
            // post-mutation snapshot in LDS, the final bytes stored straight to global memory.
            uint16_t *hrm = (uint16_t *)(lds + (DO_GATHER ? 2u : 1u) * g.R * a.pitch);
            auto for_chunks = [&](auto fn) {
                for (uint32_t item0 = wave * PS_SB; item0 < items; item0 += nw * PS_SB)
                    for (uint32_t s = 0; s < PS_SB; s++) {
                        const uint32_t item = item0 + s;
                        if (item >= items) break;
                        const uint32_t rr = item / g.segs, chunk = (item % g.segs) * 64u + lane;
                        if (chunk < a.cpr) fn(rr, chunk);
                    }
            };
            for_chunks([&](uint32_t rr, uint32_t chunk) {
                const uint32_t site = a.site_offset + r0 + rr, i0 = chunk * 16u;
                uint8_t *row = rowS + rr * a.pitch;
                uint32_t d[4];
                if (DO_GATHER) {
                    const uint8_t *par = rowA + rr * a.pitch;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint32_t w = 0;
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const uint32_t i = i0 + 4 * j + b;
                            w |= ((i < a.N) ? (uint32_t)par[a.idx[i]] : 0u) << (8 * b);
                        }
                        d[j] = w;
                    }
                } else {
                    // the row itself: cells already mutated keep their (identical) value
                    const uint4 cur = *(const uint4 *)(row + i0);
                    d[0] = cur.x; d[1] = cur.y; d[2] = cur.z; d[3] = cur.w;
                }
                const ps_u4 A = ps_philox_l1a(site >> 1, chunk, a.gen, a.k0, a.k1), B = ps_philox_l1b(site >> 2, chunk, a.gen, a.k0, a.k1);
                uint32_t cm = ps_cand_word(A, site, pl.cshift), hm = 0;
                while (cm) {
                    const uint32_t k = ps_bit_cell(__builtin_ctz(cm));
                    cm &= cm - 1u;
                    const uint32_t i = i0 + k;
                    if (i >= a.N) continue;
                    uint32_t l2y = 0;
                    const ps_cell cell = ps_cell_events(A, B, site, i, a.gen, a.k0, a.k1, pl, l2y);
                    if (DO_MUT && cell.mut) ps_set_byte(d, k, cell.mut);
                    if (DO_HR && cell.hr) hm |= 1u << k;
                }
                *(uint4 *)(row + i0) = make_uint4(d[0], d[1], d[2], d[3]);
                hrm[rr * a.cpr + chunk] = (uint16_t)hm;
            });
            ps_block_sync_lds();    // rowS is the post-mutation snapshot again, the masks are complete
            for_chunks([&](uint32_t rr, uint32_t chunk) {
                const uint32_t site = a.site_offset + r0 + rr, i0 = chunk * 16u;
                const uint8_t *row = rowS + rr * a.pitch;
                const uint4 v = *(const uint4 *)(row + i0);
                uint32_t d[4] = { v.x, v.y, v.z, v.w };
                for (uint32_t hm = hrm[rr * a.cpr + chunk]; hm;) {
                    const uint32_t k = __builtin_ctz(hm), i = i0 + k;
                    hm &= hm - 1u;
                    const ps_u4 l2 = ps_philox(site, i, a.gen, PS_STREAM_CORE_L2, a.k0, a.k1);
                    uint32_t donor = ps_mulhi(l2.y, a.N - 1u);
                    donor += (donor >= i) ? 1u : 0u;                     // population.rs:618
                    ps_set_byte(d, k, (uint32_t)row[donor]);             // snapshot read, :693-695
                }
                *(uint4 *)(a.out + (size_t)(r0 + rr) * a.pitch + i0) = make_uint4(d[0], d[1], d[2], d[3]);
            });
            if (tid == 0) *ovf = 0u;
        } else if (DO_HR && events) {
            for (uint32_t e = lane; e < nhr; e += 64u) hr_b[e] = (uint32_t)rowS[hr_b[e]];
            ps_block_sync_lds();    // all donor reads done
            for (uint32_t e = lane; e < nhr; e += 64u) rowS[hr_a[e]] = (uint8_t)hr_b[e];
        }
        PS_T(5);   // HR (two barriers inside) incl. the wait at its first barrier
        ps_block_sync_lds();        // the child rows are final; nobody reads the parent rows any more
        PS_T(6);   // barrier: child rows final
        // The prefetch has had the whole group's compute time to land.  The wait is explicit and on every
        // path: only the prefetch loads and the previous group's long-finished stores are outstanding
        // here, so vmcnt(0) is exact -- whereas waits the compiler attaches to the conditional LDS stores
        // below leave loads pending in its model on the skipped paths and come back as vmcnt waits at the
        // loop top, i.e. behind this group's stores (vmcnt counts loads and stores in issue order): one
        // exposed HBM write latency per row group.
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        // With a separate parent buffer the prefetched group is staged BEFORE this group's stores are issued.
        const bool stage_first = DO_GATHER && pipelined;
        if (stage_first) {
            if (po0 < nbytes_next) *(uint4 *)(stage + po0) = pf0;
            if (po1 < nbytes_next) *(uint4 *)(stage + po1) = pf1;
            if (po2 < nbytes_next) *(uint4 *)(stage + po2) = pf2;
            if (po3 < nbytes_next) *(uint4 *)(stage + po3) = pf3;
        }
        {
            uint8_t *dstg = a.out + (size_t)r0 * a.pitch;
            const uint32_t nbytes = nr * a.pitch;
            if (redo) {
                // (the redo pass has stored the row group itself)
            } else if (pipelined) {
                // at most PS_PF pieces per thread: the LDS reads together, then the stores
                uint4 o0 = make_uint4(0, 0, 0, 0), o1 = o0, o2 = o0, o3 = o0;
                if (po0 < nbytes) o0 = *(const uint4 *)(rowS + po0);
                if (po1 < nbytes) o1 = *(const uint4 *)(rowS + po1);
                if (po2 < nbytes) o2 = *(const uint4 *)(rowS + po2);
                if (po3 < nbytes) o3 = *(const uint4 *)(rowS + po3);
                if (po0 < nbytes) *(uint4 *)(dstg + po0) = o0;
                if (po1 < nbytes) *(uint4 *)(dstg + po1) = o1;
                if (po2 < nbytes) *(uint4 *)(dstg + po2) = o2;
                if (po3 < nbytes) *(uint4 *)(dstg + po3) = o3;
            } else {
                for (uint32_t o = tid * 16u; o < nbytes; o += blockDim.x * 16u)
                    *(uint4 *)(dstg + o) = *(const uint4 *)(rowS + o);
            }
        }
        // stage the next row group; a thread overwrites only LDS bytes it has just read itself
        if (stage_first) {
        } else if (pipelined) {
            if (po0 < nbytes_next) *(uint4 *)(stage + po0) = pf0;
            if (po1 < nbytes_next) *(uint4 *)(stage + po1) = pf1;
            if (po2 < nbytes_next) *(uint4 *)(stage + po2) = pf2;
            if (po3 < nbytes_next) *(uint4 *)(stage + po3) = pf3;
        } else if (nbytes_next) {
            const uint8_t *src = a.state + (size_t)rnext * a.pitch;
            for (uint32_t o = tid * 16u; o < nbytes_next; o += blockDim.x * 16u)
                *(uint4 *)(stage + o) = *(const uint4 *)(src + o);
        }
        ps_block_sync_lds();        // the next iteration overwrites the child rows
        PS_T(7);   // wait for the prefetch, stage, store, closing barrier
    }
#ifdef PS_STAMP
    if (lane == 0)
        for (int k = 0; k < 8; k++) atomicAdd((unsigned long long *)a.stamps + k, st_acc[k]);
#endif
}

// idxP[chunk][j] = idx[16*chunk + 2j] | idx[16*chunk + 2j + 1] << 16 (N beyond N, as in idxT):
// the parents of a lane's 16 cells as two 16-byte loads (block sweep with N <= 65536)
__global__ void idx_pack16_kernel(const uint32_t *idx, uint32_t *idxP, uint32_t N, uint32_t cpr)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 8u * cpr) return;
    const uint32_t i = 2u * t;
    const uint32_t lo = (i < N) ? idx[i] : N, hi = (i + 1u < N) ? idx[i + 1u] : N;
    idxP[t] = (lo & 0xFFFFu) | (hi << 16);
}

// idxT[k][chunk] = idx[16*chunk + k] (0 beyond N): coalesced parent indices for the block sweep
__global__ void idx_transpose_kernel(const uint32_t *idx, uint32_t *idxT, uint32_t N, uint32_t cpr)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 16u * cpr) return;
    const uint32_t k = t / cpr, chunk = t % cpr, i = chunk * 16u + k;
    idxT[t] = (i < N) ? idx[i] : N;   // N = first padding byte of a row (always 0); unused if N == pitch
}

// clonal start: every individual gets allele_vec[site] (population.rs:206-212)
__global__ void core_init_kernel(uint8_t *state, const uint8_t *allele_vec, uint32_t N,
                                 uint32_t pitch, uint32_t rows)
{
    const uint32_t cpr = pitch >> 4;
    const uint64_t total = (uint64_t)rows * cpr;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(t / cpr), c = (uint32_t)(t % cpr);
        const uint32_t v = allele_vec[row];
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; b++)
                x |= ((c * 16u + 4 * j + b < N) ? v : 0u) << (8 * b);
            w[j] = x;
        }
        *(uint4 *)(state + (size_t)row * pitch + 16u * c) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// individual-major rows[N][L] <-> site-major state[L][pitch], 64x64 tiles through LDS
template <bool TO_STATE>
__global__ void __launch_bounds__(256) core_transpose_kernel(uint8_t *state, uint8_t *rows_im,
                                                             uint32_t N, uint32_t pitch, uint64_t L)
{
    __shared__ uint8_t tile[64][65];
    const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;
    const uint64_t s0 = (uint64_t)blockIdx.x * 64u;
    const uint32_t i0 = blockIdx.y * 64u;
    if (TO_STATE) {
        for (uint32_t r = ty; r < 64; r += 4) {
            const uint32_t i = i0 + r;
            const uint64_t s = s0 + tx;
            tile[r][tx] = (i < N && s < L) ? rows_im[(uint64_t)i * L + s] : 0;
        }
        __syncthreads();
        for (uint32_t r = ty; r < 64; r += 4) {
            const uint64_t s = s0 + r;
            const uint32_t i = i0 + tx;
            if (s < L && i < pitch) state[s * pitch + i] = (i < N) ? tile[tx][r] : 0;
        }
    } else {
        for (uint32_t r = ty; r < 64; r += 4) {
            const uint64_t s = s0 + r;
            const uint32_t i = i0 + tx;
            tile[r][tx] = (s < L && i < N) ? state[s * pitch + i] : 0;
        }
        __syncthreads();
        for (uint32_t r = ty; r < 64; r += 4) {
            const uint32_t i = i0 + r;
            const uint64_t s = s0 + tx;
            if (i < N && s < L) rows_im[(uint64_t)i * L + s] = tile[tx][r];
        }
    }
}

// Population::write for the core matrix (population.rs:865-880) on the device: the text of
// individuals [i0, i0+ni) -- letters A/C/G/T/N (population.rs:154-162) joined by ',' and ended by
// '\n', 2*L bytes per individual -- produced from the site-major state through a 64x64 LDS tile so
// that both the state reads and the text writes are coalesced.
// (row_slot: line k is the individual stored at column row_slot[k] -- a ps_sim's children sit in ascending parent order,
// its lines come in draw order; nullptr = column k)
__global__ void __launch_bounds__(256) core_csv_kernel(const uint8_t *state, uint8_t *text, uint32_t pitch,
                                                       uint64_t L, uint32_t i0, uint32_t ni, uint8_t last_char,
                                                       const uint32_t *row_slot)
{
    __shared__ uint8_t tile[64][65];
    const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;
    const uint64_t s0 = (uint64_t)blockIdx.x * 64u;
    const uint32_t ib = blockIdx.y * 64u;
    const uint32_t col = (ib + tx < ni) ? (row_slot ? row_slot[i0 + ib + tx] : i0 + ib + tx) : 0u;
    for (uint32_t r = ty; r < 64; r += 4) {
        const uint64_t s = s0 + r;
        const uint32_t i = ib + tx;
        tile[r][tx] = (s < L && i < ni) ? state[s * pitch + col] : 0;
    }
    __syncthreads();
    // 128 text bytes per individual and tile: thread tx writes bytes 2*tx' .. for two halves
    for (uint32_t r = ty; r < 64; r += 4) {
        const uint32_t i = ib + r;
        if (i >= ni) continue;
        uint8_t *row = text + (uint64_t)i * 2u * L + 2u * s0;
#pragma unroll
        for (uint32_t h = 0; h < 2; h++) {
            const uint32_t c = h * 64u + tx;          // character index inside the tile's 128 bytes
            const uint64_t s = s0 + (c >> 1);
            if (s >= L) continue;
            uint8_t ch;
            if (c & 1u) ch = (s == L - 1) ? last_char : ',';      // (a site shard that is not the last one ends in ',')
            else {
                const uint8_t v = tile[c >> 1][r];
                ch = (v == 1) ? 'A' : (v == 2) ? 'C' : (v == 4) ? 'G' : (v == 8) ? 'T' : 'N';
            }
            row[c] = ch;
        }
    }
}

// ---------------------------------------------------------------------------
// sampled-pair Hamming numerators: out[k] += sum_s popcount(x[s][i_k] ^ x[s][j_k])
// (distances.rs:22-52 over the columns held by this handle).
//
// Tiled form: a workgroup packs a tile of SITES x all N individuals into LDS as
// individual-major nibble strings (8 sites per dword, alleles are one-hot < 16),
// then every thread compares the two strings of each of its A pairs with 16-byte
// LDS reads.  Each pair keeps its partial count in a register across the tiles
// of the block's site range; one atomicAdd per (range, pair).
// ---------------------------------------------------------------------------
// acc + popcount(x ^ y) over 16 bytes as a chain of four v_bcnt_u32_b32 (the instruction adds its
// second operand: no separate v_add3)
__device__ __forceinline__ uint32_t ps_bcnt_add(uint32_t v, uint32_t acc)
{
    uint32_t r;   // (the compiler re-associates popcount + add chains into v_bcnt x, 0 plus v_add3 trees)
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(acc));
    return r;
}

__device__ __forceinline__ uint32_t ps_popc_acc4(const uint4 &x, const uint4 &y, uint32_t acc)
{
    acc = ps_bcnt_add(x.x ^ y.x, acc);
    acc = ps_bcnt_add(x.y ^ y.y, acc);
    acc = ps_bcnt_add(x.z ^ y.z, acc);
    acc = ps_bcnt_add(x.w ^ y.w, acc);
    return acc;
}

// Nibble strings of 4 individuals from 8 site rows: v[b] holds the bytes (alleles < 16) of
// individuals 4*qd .. 4*qd+3 at site b; out[j] gets site b of individual j in bits 4b..4b+3.
// Two rows share a byte (v_lshl_or), then a 4 x 4 byte transpose (8 v_perm).
__device__ __forceinline__ void ps_nibble_pack8(const uint32_t *v, uint32_t (&out)[4])
{
    const uint32_t t0 = v[0] | (v[1] << 4), t1 = v[2] | (v[3] << 4), t2 = v[4] | (v[5] << 4), t3 = v[6] | (v[7] << 4);
    // __builtin_amdgcn_perm(hi, lo, sel): byte k of the result is byte sel[k] of {hi:lo} (lo = bytes 0-3)
    const uint32_t a01 = __builtin_amdgcn_perm(t1, t0, 0x05010400u);   // t0.b0 t1.b0 t0.b1 t1.b1
    const uint32_t b01 = __builtin_amdgcn_perm(t1, t0, 0x07030602u);   // t0.b2 t1.b2 t0.b3 t1.b3
    const uint32_t a23 = __builtin_amdgcn_perm(t3, t2, 0x05010400u);
    const uint32_t b23 = __builtin_amdgcn_perm(t3, t2, 0x07030602u);
    out[0] = __builtin_amdgcn_perm(a23, a01, 0x05040100u);             // individual 0: t0.b0 t1.b0 t2.b0 t3.b0
    out[1] = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
    out[2] = __builtin_amdgcn_perm(b23, b01, 0x05040100u);
    out[3] = __builtin_amdgcn_perm(b23, b01, 0x07060302u);
}

template <int A>
__global__ void __launch_bounds__(1024) core_pair_counts_tiled(
    const uint8_t *state, uint32_t N, uint32_t pitch, uint32_t rows, const uint32_t *r1,
    const uint32_t *r2, uint32_t P,
    const uint32_t *tstart, const uint32_t *tcount, uint32_t T_threads,
    uint32_t *out /* part[range][P] */, uint32_t W /* dwords per individual per tile */, uint32_t tiles_per_range)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t T[];
    const uint32_t RS4 = (W >> 2) + 1u;      // row stride in 16-byte units (odd: spreads the banks)
    const uint32_t tid = threadIdx.x;
    // Thread g owns pairs [tstart[g], tstart[g] + tcount[g]) of the list sorted by first individual,
    // all with the SAME first individual (the host splits every run evenly over ceil(run / A)
    // threads).  The first individual's 16-byte chunk is read once per chunk index and the compare
    // loop has no data-dependent branch, so the second individuals' reads are issued in batches.
    // blockIdx.x = pair block, blockIdx.y = site range: the pair blocks of one site range are
    // dispatched together, so that all but the first read their tiles from L2 / Infinity Cache
    const uint32_t g = blockIdx.x * blockDim.x + tid;
    const uint32_t start = g < T_threads ? tstart[g] : 0u;
    const uint32_t count = g < T_threads ? tcount[g] : 0u;
    const uint32_t ri = count ? r1[start] * RS4 : 0u;
    // pairs of this wave beyond the lane's own count compare the row with itself (adds nothing);
    // the loop stops at the wave's largest count
    uint32_t wmax = count;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, off, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    // per-pair counts of this workgroup's site range, two 16-bit counters per register (the host
    // keeps a range below 65536 / 2 sites)
    uint32_t pj[A], acc[A / 2];
#pragma unroll
    for (int q = 0; q < A; q++) {
        if (!(q & 1)) acc[q >> 1] = 0;
        pj[q] = ((uint32_t)q < count) ? r2[start + q] * RS4 : ri;
    }
    const uint4 *T4 = (const uint4 *)T;
    for (uint32_t t = 0; t < tiles_per_range; t++) {
        const uint32_t s0 = (blockIdx.y * tiles_per_range + t) * W * 8u;
        if (s0 >= rows) break;
        __syncthreads();
        // work item = (4 consecutive individuals, 16 consecutive sites): 16 independent dword
        // loads (a wave reads 256 contiguous bytes of a site row), repacked into one 8-byte
        // nibble string per individual
        const uint32_t quads = (N + 3u) >> 2, sgs = W >> 1;
        for (uint32_t it = tid; it < quads * sgs; it += blockDim.x) {
            const uint32_t qd = it % quads, sg = it / quads;
            const uint32_t sb = min(s0 + 16u * sg, rows - 1u);   // (a whole item past the end packs zeros)
            const bool item_valid = s0 + 16u * sg < rows;
            const uint8_t *base = state + (size_t)sb * pitch + 4u * qd;
            // 16 unconditional loads in flight (rows past the end re-read the last row and are
            // zeroed afterwards: only the last tile takes that branch)
            uint32_t v[16];
#pragma unroll
            for (int b = 0; b < 16; b++)
                v[b] = *(const uint32_t *)(base + (size_t)min((uint32_t)b, rows - 1u - sb) * pitch);
            if (sb + 16u > rows) {
#pragma unroll
                for (int b = 0; b < 16; b++)
                    if (!item_valid || sb + b >= rows) v[b] = 0u;
            }
            uint32_t lo[4], hi[4];
            ps_nibble_pack8(v, lo);
            ps_nibble_pack8(v + 8, hi);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (4u * qd + j < N) ((uint2 *)T)[(4u * qd + j) * (RS4 * 2u) + sg] = make_uint2(lo[j], hi[j]);
        }
        __syncthreads();
        for (uint32_t w4 = 0; w4 < (W >> 2); w4++) {
            const uint4 x = T4[ri + w4];
#pragma unroll
            for (int q0 = 0; q0 < A; q0 += 4) {
                if ((uint32_t)q0 < wmax) {
                    uint4 y[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) y[u] = T4[pj[q0 + u] + w4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if ((q0 + u) & 1) acc[(q0 + u) >> 1] += ps_popc_acc4(x, y[u], 0u) << 16;
                        else acc[(q0 + u) >> 1] = ps_popc_acc4(x, y[u], acc[(q0 + u) >> 1]);
                    }
                }
                // keep one batch of 4 reads in flight (the scheduler would hoist all A of them)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // partial counts of this site range, in sorted pair order: part[range][P] (every sorted position
    // belongs to exactly one thread, so plain stores; core_pair_reduce_kernel sums the ranges --
    // one atomicAdd per (range, pair) was 2.6e7 atomics = 1.3 ms at cfg2)
    uint32_t *part = out + (size_t)blockIdx.y * P;
#pragma unroll
    for (int q = 0; q < A; q++) {
        const uint32_t c = (q & 1) ? (acc[q >> 1] >> 16) : (acc[q >> 1] & 0xFFFFu);
        if ((uint32_t)q < count) part[start + q] = c;
    }
}

// ---------------------------------------------------------------------------
// One-hot matrices (every byte is 1, 2, 4 or 8 -- the simulator's own states): two bits per site.
// core_pack2_kernel writes the whole matrix once as tiles packed2[tile][individual][W] (16 sites per
// dword, code = (b >> 1) - (b >> 3): 1,2,4,8 -> 0,1,2,3); core_pair_counts_packed2 then copies a
// tile into LDS with plain 16-byte loads and counts differing sites: for d = x ^ y a site differs
// iff one of its two bits is set, popcount((d | d >> 1) & 0x5555...).  The reference's byte
// popcount of one-hot alleles is twice that (distances.rs:22-52), which is what is accumulated
// into `out`.  Compared with the nibble form: half the LDS bytes per pair (the compare is LDS
// bound) and the matrix is packed once instead of once per pair block.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ps_code2(uint32_t v)     // four bytes -> four 2-bit codes, one per byte
{
    return ((v >> 1) & 0x7F7F7F7Fu) - ((v >> 3) & 0x1F1F1F1Fu);
}

// grid.x = tile; a workgroup packs 16*W sites x all N individuals through LDS and streams the tile out
__global__ void __launch_bounds__(1024) core_pack2_kernel(const uint8_t *state, uint32_t N, uint32_t pitch,
                                                          uint32_t rows, uint32_t *packed, uint32_t W)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t T[];      // [N][W]
    const uint32_t tid = threadIdx.x;
    const uint32_t s0 = blockIdx.x * W * 16u;
    const uint32_t quads = (N + 3u) >> 2;
    // work item = (4 consecutive individuals, 16 consecutive sites) -> one dword per individual
    for (uint32_t it = tid; it < quads * W; it += blockDim.x) {
        const uint32_t qd = it % quads, w = it / quads;
        const uint32_t sb = min(s0 + 16u * w, rows - 1u);
        const bool item_valid = s0 + 16u * w < rows;
        const uint8_t *base = state + (size_t)sb * pitch + 4u * qd;
        uint32_t v[16];
#pragma unroll
        for (int b = 0; b < 16; b++)
            v[b] = *(const uint32_t *)(base + (size_t)min((uint32_t)b, rows - 1u - sb) * pitch);
        if (sb + 16u > rows || !item_valid) {
#pragma unroll
            for (int b = 0; b < 16; b++)
                if (!item_valid || sb + b >= rows) v[b] = 0u;
        }
        uint32_t t[4];
#pragma unroll
        for (int m = 0; m < 4; m++)      // byte j of t[m]: sites 4m .. 4m+3 of individual j
            t[m] = ps_code2(v[4 * m]) | (ps_code2(v[4 * m + 1]) << 2) | (ps_code2(v[4 * m + 2]) << 4) | (ps_code2(v[4 * m + 3]) << 6);
        const uint32_t a01 = __builtin_amdgcn_perm(t[1], t[0], 0x05010400u), b01 = __builtin_amdgcn_perm(t[1], t[0], 0x07030602u);
        const uint32_t a23 = __builtin_amdgcn_perm(t[3], t[2], 0x05010400u), b23 = __builtin_amdgcn_perm(t[3], t[2], 0x07030602u);
        const uint32_t o[4] = { __builtin_amdgcn_perm(a23, a01, 0x05040100u), __builtin_amdgcn_perm(a23, a01, 0x07060302u),
                                __builtin_amdgcn_perm(b23, b01, 0x05040100u), __builtin_amdgcn_perm(b23, b01, 0x07060302u) };
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (4u * qd + j < N) T[(4u * qd + j) * W + w] = o[j];
    }
    __syncthreads();
    uint4 *dst = (uint4 *)(packed + (size_t)blockIdx.x * N * W);
    for (uint32_t k = tid; k < N * (W >> 2); k += blockDim.x) dst[k] = ((const uint4 *)T)[k];
}

template <int A>
__global__ void __launch_bounds__(1024) core_pair_counts_packed2(
    const uint32_t *packed, uint32_t N, uint32_t n_tiles, const uint32_t *r1, const uint32_t *r2,
    uint32_t P, const uint32_t *tstart, const uint32_t *tcount,
    uint32_t T_threads, uint32_t *out /* part[range][P] */, uint32_t W, uint32_t tiles_per_range)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t T[];
    const uint32_t RS4 = (W >> 2) + 1u;      // LDS row stride in 16-byte units (odd: spreads the banks)
    const uint32_t tid = threadIdx.x;
    // thread table, wave-uniform loop bound and pair order: see core_pair_counts_tiled
    const uint32_t g = blockIdx.x * blockDim.x + tid;
    const uint32_t start = g < T_threads ? tstart[g] : 0u;
    const uint32_t count = g < T_threads ? tcount[g] : 0u;
    const uint32_t ri = count ? r1[start] * RS4 : 0u;
    uint32_t wmax = count;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, off, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    uint32_t pj[A], acc[A / 2];              // 16-bit halves: the host keeps a range below 65536 sites
#pragma unroll
    for (int q = 0; q < A; q++) {
        if (!(q & 1)) acc[q >> 1] = 0;
        pj[q] = ((uint32_t)q < count) ? r2[start + q] * RS4 : ri;
    }
    uint4 *T4 = (uint4 *)T;
    for (uint32_t t = 0; t < tiles_per_range; t++) {
        const uint32_t tile = blockIdx.y * tiles_per_range + t;
        if (tile >= n_tiles) break;
        __syncthreads();
        // copy the packed tile into LDS, eight 16-byte loads in flight per thread (a load-store loop
        // would pay one global-load latency per iteration: 8 x ~2 us per tile, as long as the compare)
        const uint4 *src = (const uint4 *)(packed + (size_t)tile * N * W);
        const uint32_t w4s = W >> 2, total = N * w4s;
        const uint32_t sh = 31u - (uint32_t)__builtin_clz(w4s);          // w4s is a power of two
        for (uint32_t k0 = tid; k0 < total; k0 += 8u * blockDim.x) {
            uint4 buf[8];
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) buf[u] = src[min(k0 + u * blockDim.x, total - 1u)];
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                const uint32_t k = k0 + u * blockDim.x;
                if (k < total) T4[(k >> sh) * RS4 + (k & (w4s - 1u))] = buf[u];
            }
        }
        __syncthreads();
        for (uint32_t w4 = 0; w4 < w4s; w4++) {
            const uint4 x = T4[ri + w4];
#pragma unroll
            for (int q0 = 0; q0 < A; q0 += 4) {
                if ((uint32_t)q0 < wmax) {
                    uint4 y[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) y[u] = T4[pj[q0 + u] + w4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        uint32_t c = 0;
                        uint32_t d;
                        d = x.x ^ y[u].x; c = ps_bcnt_add((d | (d >> 1)) & 0x55555555u, c);
                        d = x.y ^ y[u].y; c = ps_bcnt_add((d | (d >> 1)) & 0x55555555u, c);
                        d = x.z ^ y[u].z; c = ps_bcnt_add((d | (d >> 1)) & 0x55555555u, c);
                        d = x.w ^ y[u].w; c = ps_bcnt_add((d | (d >> 1)) & 0x55555555u, c);
                        acc[(q0 + u) >> 1] += ((q0 + u) & 1) ? (c << 16) : c;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    uint32_t *part = out + (size_t)blockIdx.y * P;       // see core_pair_counts_tiled
#pragma unroll
    for (int q = 0; q < A; q++) {
        const uint32_t c = (q & 1) ? (acc[q >> 1] >> 16) : (acc[q >> 1] & 0xFFFFu);
        if ((uint32_t)q < count) part[start + q] = 2u * c;
    }
}

// out[perm[k]] = sum over the site ranges of part[range][k]
__global__ void __launch_bounds__(256) core_pair_reduce_kernel(const uint32_t *part, uint32_t ranges, uint32_t P,
                                                               const uint32_t *perm, uint32_t *out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    uint32_t s = 0;
    for (uint32_t r = 0; r < ranges; r++) s += part[(size_t)r * P + k];
    out[perm ? perm[k] : k] = s;
}

// ---------------------------------------------------------------------------
// All-pairs Hamming numerators, register tiled: H[i][j] += sum_s popcount(x[s][i] ^ x[s][j])
// for every pair of a 128 x 128 tile of individuals (tiles with ti <= tj only).  Cost is
// independent of the number of requested pairs, so it serves P close to N^2/2 (cfg5) and
// per-generation statistics; core_pair_lookup_kernel then picks the sampled pairs.
// A workgroup packs the two individual tiles of a 256-site chunk into LDS as nibble
// strings (same packing as the sampled kernel); thread (ty,tx) owns the 8 x 8 pairs
// (ti*128 + a*16 + ty, tj*128 + b*16 + tx): the 16 lanes that share ty read the same A
// rows (broadcast) and rows tx + 16 b of B, 16 bytes apart by an odd stride (no conflict).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void ps_pack_tile(uint4 *T, const uint8_t *state, uint32_t N, uint32_t pitch,
                                             uint32_t rows, uint32_t ibase, uint32_t s0, uint32_t W)
{
    const uint32_t RS4 = (W >> 2) + 1u, sgs = W >> 2;
    for (uint32_t it = threadIdx.x; it < 32u * sgs; it += blockDim.x) {
        const uint32_t qd = it & 31u, sg = it >> 5;
        const uint32_t sb = s0 + 32u * sg;
        const uint32_t icol = ibase + 4u * qd;
        uint32_t v[32];
        const bool item_valid = sb < rows && icol < pitch;
        const uint32_t sbc = min(sb, rows - 1u);
        const uint8_t *base = state + (size_t)sbc * pitch + min(icol, pitch - 4u);
#pragma unroll
        for (int b = 0; b < 32; b++)
            v[b] = *(const uint32_t *)(base + (size_t)min((uint32_t)b, rows - 1u - sbc) * pitch);
        if (!item_valid || sb + 32u > rows) {
#pragma unroll
            for (int b = 0; b < 32; b++)
                if (!item_valid || sb + b >= rows) v[b] = 0u;
        }
        uint32_t o0[4], o1[4], o2[4], o3[4];
        ps_nibble_pack8(v, o0);
        ps_nibble_pack8(v + 8, o1);
        ps_nibble_pack8(v + 16, o2);
        ps_nibble_pack8(v + 24, o3);
#pragma unroll
        for (int j = 0; j < 4; j++) T[(4u * qd + j) * RS4 + sg] = make_uint4(o0[j], o1[j], o2[j], o3[j]);
    }
    (void)N;
}

__global__ void __launch_bounds__(256) core_allpairs_kernel(const uint8_t *state, uint32_t N, uint32_t pitch,
                                                            uint32_t rows, uint32_t *H, uint32_t W,
                                                            uint32_t chunks_per_range, uint32_t ntile)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t T[];
    const uint32_t RS4 = (W >> 2) + 1u;
    uint4 *TA = (uint4 *)T, *TB = TA + 128u * RS4;
    // blockIdx.x enumerates tile pairs ti <= tj row by row
    uint32_t ti = 0, rem = blockIdx.x;
    while (rem >= ntile - ti) { rem -= ntile - ti; ti++; }
    const uint32_t tj = ti + rem;
    const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
    uint32_t acc[8][8];
#pragma unroll
    for (int a = 0; a < 8; a++)
#pragma unroll
        for (int b = 0; b < 8; b++) acc[a][b] = 0;
    const uint32_t sites_per_chunk = W * 8u;
    for (uint32_t c = 0; c < chunks_per_range; c++) {
        const uint32_t s0 = (blockIdx.y * chunks_per_range + c) * sites_per_chunk;
        if (s0 >= rows) break;
        __syncthreads();
        ps_pack_tile(TA, state, N, pitch, rows, ti * 128u, s0, W);
        ps_pack_tile(TB, state, N, pitch, rows, tj * 128u, s0, W);
        __syncthreads();
        for (uint32_t w4 = 0; w4 < (W >> 2); w4++) {
            uint4 xb[8];
#pragma unroll
            for (int b = 0; b < 8; b++) xb[b] = TB[(tx + 16u * b) * RS4 + w4];
#pragma unroll
            for (int a = 0; a < 8; a++) {
                const uint4 xa = TA[(ty + 16u * a) * RS4 + w4];
#pragma unroll
                for (int b = 0; b < 8; b++)
                    acc[a][b] = ps_popc_acc4(xa, xb[b], acc[a][b]);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const uint32_t i = ti * 128u + ty + 16u * a;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint32_t j = tj * 128u + tx + 16u * b;
            if (i < N && j < N && acc[a][b]) atomicAdd(&H[(size_t)i * N + j], acc[a][b]);
        }
    }
}

// out[slot] = H[i][j] with the tile-ordered lookup (only tiles ti <= tj were computed)
__global__ void core_pair_lookup_kernel(const uint32_t *H, uint32_t N, const uint32_t *r1, const uint32_t *r2,
                                        const uint32_t *perm, uint64_t P, uint32_t *out)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const uint32_t i = r1[k], j = r2[k];
    const bool fwd = (i >> 7) <= (j >> 7);
    out[perm ? perm[k] : k] = fwd ? H[(size_t)i * N + j] : H[(size_t)j * N + i];
}

// ---------------------------------------------------------------------------
// Transposed form for populations too wide for an LDS tile of all N individuals (cfg4: N = 65536).
// core_packT_kernel rewrites the site-major matrix once per call as individual-major bit strings
// packT[individual][WT] (one-hot matrices: 2 bits per site, 16 sites per dword, code as in
// core_pack2_kernel; other nibble-safe matrices: 4 bits per site), then core_pair_counts_rows streams
// the two strings of every sampled pair -- regime (i) of SURVEY 8(d), "no reuse", but on strings 4x
// (2x) smaller than the reference's byte rows: HBM traffic N*L + N*L/4 for the transposition plus
// P * 2 * L/4 for the pairs (cfg4: 98 GB + 60 GB instead of 240 GB of byte rows that the site-major
// layout cannot even stream).
// Tile of a workgroup: 256 individuals x 32 dwords.  Work item = (4 consecutive individuals, 16 or 8
// consecutive sites), as in core_pack2_kernel: a wave reads 256 contiguous bytes of 16 (8) site rows;
// the four dwords of an item go to LDS as one 16-byte store (word-major tile T[word][individual]), and
// the tile leaves as 128 contiguous bytes per individual.
// ---------------------------------------------------------------------------
#ifndef PS_PT_IB
#define PS_PT_IB 256u   // individuals per tile
#endif
#ifndef PS_PT_WB
#define PS_PT_WB 32u    // dwords per individual per tile
#endif
// BLOCKED (the matrix-core all-pairs kernels): the same strings cut into the pieces one operand fragment loads at a time --
// packT[group of 32 individuals][chunk of 8 dwords][half h][r][4 dwords], 1 KB per (group, chunk), so that a wave's 64 x 16-byte
// load is 1 KB of consecutive addresses (individual-major strings gave it 32 bytes of 32 different 128-byte lines per chunk:
// every line crossed the L2 -> L1 path four times).  Rows of the last group past N are written as zeros.
// PLANES (with BLOCKED; the signed all-pairs kernel): the 16 bytes of a lane hold bit PLANES instead of 2-bit codes -- the high
// code bits of its first and second 32 sites, then the low code bits of the same sites (lane (h, r): sites [64 h, 64 h + 64)
// of the chunk's 128).
template <bool NIB, bool BLOCKED = false, bool PLANES = false>
__global__ void __launch_bounds__(256) core_packT_kernel(const uint8_t *state, uint32_t N, uint32_t pitch, uint32_t rows,
                                                         uint32_t *packT, uint32_t WT)
{
    constexpr uint32_t SPW = NIB ? 8u : 16u;          // sites per dword
    constexpr uint32_t RS = PS_PT_IB + 4u;            // LDS row stride in dwords (16-byte aligned, banks shifted by 4 per word)
    __shared__ __attribute__((aligned(16))) uint32_t T[PS_PT_WB * RS];   // T[w][individual]
    static_assert(!PLANES || (BLOCKED && !NIB), "bit planes: blocked strings of one-hot matrices");
    const uint32_t tid = threadIdx.x;
    // consecutive workgroups take consecutive individual blocks of the same 512 (256) site rows
    const uint32_t nib = (N + PS_PT_IB - 1u) / PS_PT_IB;
    const uint32_t i0 = (blockIdx.x % nib) * PS_PT_IB, w0 = (blockIdx.x / nib) * PS_PT_WB;
    // 64 quads x 32 words = 2048 items, 8 per thread; consecutive lanes take consecutive quads
    for (uint32_t it = tid; it < (PS_PT_IB / 4u) * PS_PT_WB; it += 256u) {
        const uint32_t qd = it % (PS_PT_IB / 4u), w = it / (PS_PT_IB / 4u);
        const uint32_t s_first = (w0 + w) * SPW;
        const bool item_valid = s_first < rows;
        const uint32_t sb = min(s_first, rows - 1u);
        const uint32_t col = min(i0 + 4u * qd, pitch - 4u);       // (columns beyond N are padding zeros or unused)
        const uint8_t *base = state + (size_t)sb * pitch + col;
        uint32_t v[SPW];
#pragma unroll
        for (uint32_t b = 0; b < SPW; b++)
            v[b] = *(const uint32_t *)(base + (size_t)min(b, rows - 1u - sb) * pitch);
        if (!item_valid || sb + SPW > rows) {
#pragma unroll
            for (uint32_t b = 0; b < SPW; b++)
                if (!item_valid || sb + b >= rows) v[b] = 0u;
        }
        uint32_t o[4];
        if (NIB) {
            ps_nibble_pack8(v, o);
        } else {
            uint32_t t[4];
            if (PLANES) {
                // byte j of t[0], t[1]: the high code bit (one-hot byte 4 or 8) of sites 0..7, 8..15 of individual j;
                // t[2], t[3]: the low code bit (byte 2 or 8).  o[j] = high bits of the 16 sites | low bits << 16
                t[0] = t[1] = t[2] = t[3] = 0u;
#pragma unroll
                for (uint32_t b = 0; b < 8u; b++) {
                    const uint32_t x = v[b], y = v[8u + b];
                    t[0] |= (((x >> 2) | (x >> 3)) & 0x01010101u) << b;
                    t[1] |= (((y >> 2) | (y >> 3)) & 0x01010101u) << b;
                    t[2] |= (((x >> 1) | (x >> 3)) & 0x01010101u) << b;
                    t[3] |= (((y >> 1) | (y >> 3)) & 0x01010101u) << b;
                }
            } else {
#pragma unroll
            for (int m = 0; m < 4; m++)      // byte j of t[m]: sites 4m .. 4m+3 of individual j
                t[m] = ps_code2(v[4 * m]) | (ps_code2(v[4 * m + 1]) << 2) | (ps_code2(v[4 * m + 2]) << 4) | (ps_code2(v[4 * m + 3]) << 6);
            }
            const uint32_t a01 = __builtin_amdgcn_perm(t[1], t[0], 0x05010400u), b01 = __builtin_amdgcn_perm(t[1], t[0], 0x07030602u);
            const uint32_t a23 = __builtin_amdgcn_perm(t[3], t[2], 0x05010400u), b23 = __builtin_amdgcn_perm(t[3], t[2], 0x07030602u);
            o[0] = __builtin_amdgcn_perm(a23, a01, 0x05040100u);
            o[1] = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
            o[2] = __builtin_amdgcn_perm(b23, b01, 0x05040100u);
            o[3] = __builtin_amdgcn_perm(b23, b01, 0x07060302u);
        }
        *(uint4 *)(T + w * RS + 4u * qd) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    if (BLOCKED) {
        // a wave writes the 1 KB of one (group, chunk): lane = h * 32 + r, 4 chunks of the tile = 4 waves, 8 groups = 8 passes
        static_assert(PS_PT_WB == 32u && PS_PT_IB == 256u, "tile = 8 groups x 4 chunks");
        const uint32_t lane = tid & 63u, cc = tid >> 6, r = lane & 31u, h = lane >> 5;
        const uint32_t n_chunks = WT / 8u, ngroups = (N + 31u) / 32u;
        for (uint32_t gi = 0; gi < 8u; gi++) {
            const uint32_t g = i0 / 32u + gi, ind = gi * 32u + r;
            if (g >= ngroups) break;
            const uint32_t *src = T + (8u * cc + 4u * h) * RS + ind;
            uint4 o = make_uint4(src[0], src[RS], src[2u * RS], src[3u * RS]);
            if (PLANES)         // four (high | low << 16) words of 16 sites -> high bits of 2 x 32 sites, low bits of 2 x 32 sites
                o = make_uint4(__builtin_amdgcn_perm(o.y, o.x, 0x05040100u), __builtin_amdgcn_perm(o.w, o.z, 0x05040100u),
                               __builtin_amdgcn_perm(o.y, o.x, 0x07060302u), __builtin_amdgcn_perm(o.w, o.z, 0x07060302u));
            if (i0 + ind >= N) o = make_uint4(0u, 0u, 0u, 0u);
            *(uint4 *)(packT + ((size_t)g * n_chunks + w0 / 8u + cc) * 256u + lane * 4u) = o;
        }
        return;
    }
    // 8 lanes write the 128 bytes of one individual; 32 individuals per pass of the workgroup
    constexpr uint32_t LPI = PS_PT_WB / 4u;             // lanes per individual (16 bytes each)
    const uint32_t k4 = tid % LPI;
    for (uint32_t ind = tid / LPI; ind < PS_PT_IB; ind += 256u / LPI) {
        if (i0 + ind >= N) break;
        const uint32_t *src = T + (4u * k4) * RS + ind;
        const uint4 o = make_uint4(src[0], src[RS], src[2u * RS], src[3u * RS]);
        *(uint4 *)(packT + (size_t)(i0 + ind) * WT + w0 + 4u * k4) = o;
    }
}

// one wave per pair: both strings streamed with 16-byte loads, four of each in flight per lane
template <bool NIB>
__global__ void __launch_bounds__(256) core_pair_counts_rows(const uint32_t *packT, uint32_t WT, const uint32_t *r1,
                                                             const uint32_t *r2, const uint32_t *perm, uint64_t P,
                                                             uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint32_t n4 = WT >> 2;
    auto cnt = [](const uint4 &x, const uint4 &y, uint32_t c) -> uint32_t {
        if (NIB) return ps_popc_acc4(x, y, c);
        uint32_t d;
        d = x.x ^ y.x; c = ps_bcnt_add((d | (d >> 1)) & 0x55555555u, c);
        d = x.y ^ y.y; c = ps_bcnt_add((d | (d >> 1)) & 0x55555555u, c);
        d = x.z ^ y.z; c = ps_bcnt_add((d | (d >> 1)) & 0x55555555u, c);
        d = x.w ^ y.w; c = ps_bcnt_add((d | (d >> 1)) & 0x55555555u, c);
        return c;
    };
    for (uint64_t k = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < P; k += nwaves) {
        const uint4 *x = (const uint4 *)(packT + (size_t)r1[k] * WT), *y = (const uint4 *)(packT + (size_t)r2[k] * WT);
        uint32_t c = 0, w = lane;
        for (; w + 192u < n4; w += 256u) {
            uint4 xa[4], ya[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) { xa[u] = x[w + 64u * u]; ya[u] = y[w + 64u * u]; }
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) c = cnt(xa[u], ya[u], c);
        }
        for (; w < n4; w += 64u) c = cnt(x[w], y[w], c);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        // one-hot: a differing site counts 2 in the reference's byte popcount (distances.rs:22-52)
        if (lane == 0) out[perm ? perm[k] : k] = NIB ? c : 2u * c;
    }
}

// ---------------------------------------------------------------------------
// All-pairs Hamming numerators on the matrix cores (one-hot matrices, P > ~N^2 / 4: cfg5).
// For one-hot alleles the number of MATCHING sites of two individuals is the inner product of their one-hot
// expansions (4 x {0,1} per site), so all pairs are one X X^T contraction -- SURVEY 8(d) notes this form is
// MFMA-eligible.  It is exact: v_mfma_i32_32x32x32_i8 multiplies i8 {0,1} and accumulates in i32 (<= L matches);
// the reference's numerator (distances.rs:22-52: byte popcount, 2 per differing site) is 2 * (sites - matches).
// Operands come from the blocked 2-bit strings of core_packT_kernel<false, true> (16 sites per dword; padding sites
// hold code 0 for everybody, i.e. they match and drop out of sites - matches).
//   * A workgroup of 8 waves owns a 256 x 256 tile of pairs (tiles with ti <= tj only) over a range of 128-site
//     chunks; a wave owns 128 x 64 pairs: 4 + 2 operand fragments and 4 x 2 accumulator blocks of 32 x 32.
//   * Lane l (r = l & 31, h = l >> 5) of a fragment loads 16 bytes = 64 sites of individual base + r: half h of
//     the chunk.  Byte t of those 16 bytes (4 sites) is K-step t: a 256-entry table in LDS turns the byte into the
//     lane's 16 i8 operand values (4 sites x 4 alleles) with ONE ds_read_b128 -- the K order (half, site, allele)
//     is the same for both operands, which is all the contraction needs.
//   * The table is stored 16 times, entry e of copy c at byte e * 256 + c * 16: a lane always reads copy
//     c(lane), and the 16 lanes the LDS serves together (MI355X_MICROARCH.md, ds_read_b128 lane groups) have 16
//     different copies = 16 different bank groups, so the data-dependent reads never conflict.  The address is
//     one v_perm: (byte << 8) | (copy << 4).
//   * Partial counts of a (tile, chunk range) are stored to slice `range` of H (gridDim.y slices of N x N words; 32
//     consecutive lanes store 32 consecutive words); core_allpairs_sum_slices_kernel leaves their sum in slice 0.
// ---------------------------------------------------------------------------
typedef int ps_v4i __attribute__((ext_vector_type(4)));
typedef int ps_v16i __attribute__((ext_vector_type(16)));
#define PS_MF_TILE 256u          // individuals per workgroup tile side
#define PS_MF_CHUNK_DW 8u        // dwords of a 2-bit string per chunk (128 sites)

__device__ __forceinline__ ps_v4i ps_mf_lut_read(const uint8_t *lut, uint32_t raw, uint32_t colofs, uint32_t b)
{
    // address = (byte b of raw) << 8 | colofs  (colofs = copy * 16 < 256)
    uint32_t addr;
    switch (b) {
    case 0: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0400u); break;
    case 1: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0500u); break;
    case 2: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0600u); break;
    default: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0700u); break;
    }
    // (the table sits at LDS offset 0: the 32-bit value IS the LDS address; going through `lut + addr` costs a v_add of 0)
    (void)lut;
    return *(const __attribute__((address_space(3))) ps_v4i *)(uintptr_t)addr;
}

__global__ void __launch_bounds__(512) core_allpairs_mfma_kernel(const uint32_t *packT, uint32_t WT, uint32_t N, uint32_t *H,
                                                                 uint32_t chunks_per_range, uint32_t n_chunks, uint32_t ntile)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lut[];      // 256 entries x 16 copies x 16 bytes = 64 KB, at LDS offset 0
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the table: entry e = four 2-bit codes c0..c3 (site k in bits 2k, 2k+1) -> dword k = 1 << (8 * ck)
    for (uint32_t x = tid; x < 256u * 16u; x += 512u) {
        const uint32_t e = x >> 4;
        uint4 v;
        v.x = 1u << (8u * (e & 3u));
        v.y = 1u << (8u * ((e >> 2) & 3u));
        v.z = 1u << (8u * ((e >> 4) & 3u));
        v.w = 1u << (8u * ((e >> 6) & 3u));
        *(uint4 *)(lut + (size_t)x * 16u) = v;       // x = e * 16 + copy
    }
    // copy of this lane: distinct inside each group of 16 lanes a ds_read_b128 serves in one pass
    // ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32)
    const uint32_t l5 = lane & 31u;
    const uint32_t copy = l5 < 4u ? l5 : l5 < 12u ? l5 - 4u : l5 < 20u ? l5 - 8u : l5 < 28u ? l5 - 12u : l5 - 16u;
    const uint32_t colofs = copy << 4;
    // tile pair ti <= tj, row by row
    uint32_t ti = 0, rem = blockIdx.x;
    while (rem >= ntile - ti) { rem -= ntile - ti; ti++; }
    const uint32_t tj = ti + rem;
    // wave (wi, wj) of the 2 x 4 arrangement: rows ti*256 + wi*128 .. +127, columns tj*256 + wj*64 .. +63
    const uint32_t wi = wave >> 2, wj = wave & 3u;
    const uint32_t r = lane & 31u, h = lane >> 5;
    const uint32_t *src[6];
    const uint32_t ngroups = (N + 31u) / 32u;
#pragma unroll
    for (uint32_t f = 0; f < 6u; f++) {
        const uint32_t grp = f < 4u ? ti * (PS_MF_TILE / 32u) + wi * 4u + f : tj * (PS_MF_TILE / 32u) + wj * 2u + (f - 4u);
        // (groups past the population repeat the last one; never stored.  Blocked strings: 256 dwords per (group, chunk))
        src[f] = packT + (size_t)min(grp, ngroups - 1u) * (WT / PS_MF_CHUNK_DW) * 256u + lane * 4u;
    }
    ps_v16i acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[a][b][v] = 0;
    const uint32_t c_lo = blockIdx.y * chunks_per_range, c_hi = min(n_chunks, c_lo + chunks_per_range);
    if (c_lo >= c_hi) return;       // (wave-uniform: the whole workgroup leaves)
    uint4 cur[6], nxt[6];
#pragma unroll
    for (uint32_t f = 0; f < 6u; f++) cur[f] = *(const uint4 *)(src[f] + (size_t)c_lo * 256u);
    __syncthreads();        // the table is complete
    // Software pipeline over the K-steps: the six table reads of step t + 1 are issued before the eight MFMAs of
    // step t (two operand sets, ping-pong by the parity of t; 16 steps per chunk, so the parity carries over the
    // chunk loop), and the 16-byte loads of the next chunk are in flight during the whole chunk.
    ps_v4i opA[6], opB[6];
#pragma unroll
    for (uint32_t f = 0; f < 6u; f++) opA[f] = ps_mf_lut_read(lut, cur[f].x, colofs, 0u);
    for (uint32_t c = c_lo; c < c_hi; c++) {
        const uint32_t cn = min(c + 1u, c_hi - 1u);
#pragma unroll
        for (uint32_t f = 0; f < 6u; f++) nxt[f] = *(const uint4 *)(src[f] + (size_t)cn * 256u);
#pragma unroll
        for (uint32_t t = 0; t < 16u; t++) {
            const uint32_t tn = (t + 1u) & 15u;
#pragma unroll
            for (uint32_t f = 0; f < 6u; f++) {
                const uint4 &w = (t == 15u) ? nxt[f] : cur[f];
                const uint32_t raw = (tn >> 2) == 0u ? w.x : (tn >> 2) == 1u ? w.y : (tn >> 2) == 2u ? w.z : w.w;
                if (t & 1u) opA[f] = ps_mf_lut_read(lut, raw, colofs, tn & 3u);
                else opB[f] = ps_mf_lut_read(lut, raw, colofs, tn & 3u);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
                    acc[a][b] = (t & 1u) ? __builtin_amdgcn_mfma_i32_32x32x32_i8(opB[a], opB[4 + b], acc[a][b], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_i32_32x32x32_i8(opA[a], opA[4 + b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (uint32_t f = 0; f < 6u; f++) cur[f] = nxt[f];
    }
    // C layout of the 32 x 32 blocks (dtype independent): col = lane & 31, row = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)
    // (plain stores into this chunk range's own slice of H; core_allpairs_sum_slices_kernel adds the slices.  One atomic per
    // pair and range into a single N x N array was 47 of the kernel's 51 ms at N = 8192: device-scope atomics do not run in
    // the L2 of an XCD)
    const uint32_t sites = (c_hi - c_lo) * PS_MF_CHUNK_DW * 16u;
    uint32_t *Hs = H + (size_t)blockIdx.y * N * N;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const uint32_t j = tj * PS_MF_TILE + wj * 64u + (uint32_t)b * 32u + r;
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const uint32_t i = ti * PS_MF_TILE + wi * 128u + (uint32_t)a * 32u + (uint32_t)((v & 3) + 8 * (v >> 2)) + 4u * h;
                const uint32_t mism = sites - (uint32_t)acc[a][b][v];
                if (i < N && j < N) Hs[(size_t)i * N + j] = 2u * mism;
            }
        }
}

// The same contraction on the block-scaled FP4 path of the matrix cores (v_mfma_scale_f32_32x32x64_f8f6f4, both operands
// E2M1, both block scales 2^0): a one-hot allele is the FP4 value 1.0 (0b0010) in one of a site's four nibbles, so a K = 64
// instruction covers 8 sites per lane half where the i8 form covers 4, at the same cycles (FP4 runs at 4x the bf16 rate,
// i8 at 2x).  {0, 1} products and f32 sums of at most 2^24 matches are exact, so the integers equal the i8 form's.
// Table: byte (4 sites) -> 8 bytes (4 sites x 4 nibbles), stored 32 times (64 KB): a lane reads copy (lane & 31), and the
// 32 lanes a ds_read_b64 serves together cover all 64 banks exactly once.  Two table reads per operand fragment and step.
typedef int ps_v8i __attribute__((ext_vector_type(8)));
typedef float ps_v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint2 ps_mf_lut_read64(uint32_t raw, uint32_t colofs, uint32_t b)
{
    uint32_t addr;      // (byte b of raw) << 8 | colofs  (colofs = copy * 8 < 256); the table sits at LDS offset 0
    switch (b) {
    case 0: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0400u); break;
    case 1: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0500u); break;
    case 2: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0600u); break;
    default: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0700u); break;
    }
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = *(const __attribute__((address_space(3))) u32x2 *)(uintptr_t)addr;
    return make_uint2(v.x, v.y);
}

__global__ void __launch_bounds__(512) core_allpairs_mfma_fp4_kernel(const uint32_t *packT, uint32_t WT, uint32_t N, uint32_t *H,
                                                                     uint32_t chunks_per_range, uint32_t n_chunks, uint32_t ntile)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lut[];      // 256 entries x 32 copies x 8 bytes = 64 KB, at LDS offset 0
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t x = tid; x < 256u * 32u; x += 512u) {
        const uint32_t e = x >> 5;
        uint2 v;      // site k: 16 bits, nibble ck = 0x2 (E2M1 1.0)
        v.x = (2u << (4u * (e & 3u))) | (2u << (16u + 4u * ((e >> 2) & 3u)));
        v.y = (2u << (4u * ((e >> 4) & 3u))) | (2u << (16u + 4u * ((e >> 6) & 3u)));
        *(uint2 *)(lut + (size_t)x * 8u) = v;        // x = e * 32 + copy
    }
    const uint32_t colofs = (lane & 31u) << 3;
    uint32_t ti = 0, rem = blockIdx.x;
    while (rem >= ntile - ti) { rem -= ntile - ti; ti++; }
    const uint32_t tj = ti + rem;
    const uint32_t wi = wave >> 2, wj = wave & 3u;
    const uint32_t r = lane & 31u, h = lane >> 5;
    const uint32_t *src[6];
    const uint32_t ngroups = (N + 31u) / 32u;
#pragma unroll
    for (uint32_t f = 0; f < 6u; f++) {
        const uint32_t grp = f < 4u ? ti * (PS_MF_TILE / 32u) + wi * 4u + f : tj * (PS_MF_TILE / 32u) + wj * 2u + (f - 4u);
        // (groups past the population repeat the last one; never stored.  Blocked strings: 256 dwords per (group, chunk))
        src[f] = packT + (size_t)min(grp, ngroups - 1u) * (WT / PS_MF_CHUNK_DW) * 256u + lane * 4u;
    }
    ps_v16f acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[a][b][v] = 0.0f;
    const uint32_t c_lo = blockIdx.y * chunks_per_range, c_hi = min(n_chunks, c_lo + chunks_per_range);
    if (c_lo >= c_hi) return;
    uint4 cur[6], nxt[6];
#pragma unroll
    for (uint32_t f = 0; f < 6u; f++) cur[f] = *(const uint4 *)(src[f] + (size_t)c_lo * 256u);
    __syncthreads();
    // 8 K-steps per chunk: step t takes bytes 2t, 2t+1 of the lane's 16 (= dword t >> 1, byte pair t & 1)
    uint2 opA[6][2], opB[6][2];
    auto fetch = [&](const uint4 &w, uint32_t t, uint2 (&op)[2]) {
        const uint32_t raw = (t >> 1) == 0u ? w.x : (t >> 1) == 1u ? w.y : (t >> 1) == 2u ? w.z : w.w;
        op[0] = ps_mf_lut_read64(raw, colofs, 2u * (t & 1u));
        op[1] = ps_mf_lut_read64(raw, colofs, 2u * (t & 1u) + 1u);
    };
#pragma unroll
    for (uint32_t f = 0; f < 6u; f++) fetch(cur[f], 0u, opA[f]);
    const int one = 0x7f7f7f7f;          // E8M0 block scale 2^0 in every byte
    for (uint32_t c = c_lo; c < c_hi; c++) {
        const uint32_t cn = min(c + 1u, c_hi - 1u);
#pragma unroll
        for (uint32_t f = 0; f < 6u; f++) nxt[f] = *(const uint4 *)(src[f] + (size_t)cn * 256u);
#pragma unroll
        for (uint32_t t = 0; t < 8u; t++) {
            const uint32_t tn = (t + 1u) & 7u;
#pragma unroll
            for (uint32_t f = 0; f < 6u; f++) {
                if (t & 1u) fetch((t == 7u) ? nxt[f] : cur[f], tn, opA[f]);
                else fetch((t == 7u) ? nxt[f] : cur[f], tn, opB[f]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const uint2 *xa = (t & 1u) ? opB[a] : opA[a], *xb = (t & 1u) ? opB[4 + b] : opA[4 + b];
                    const ps_v8i va = { (int)xa[0].x, (int)xa[0].y, (int)xa[1].x, (int)xa[1].y, 0, 0, 0, 0 };
                    const ps_v8i vb = { (int)xb[0].x, (int)xb[0].y, (int)xb[1].x, (int)xb[1].y, 0, 0, 0, 0 };
                    acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc[a][b], 4, 4, 0, one, 0, one);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (uint32_t f = 0; f < 6u; f++) cur[f] = nxt[f];
    }
    // (plain stores into this chunk range's own slice of H; core_allpairs_sum_slices_kernel adds the slices.  One atomic per
    // pair and range into a single N x N array was 47 of the kernel's 51 ms at N = 8192: device-scope atomics do not run in
    // the L2 of an XCD)
    const uint32_t sites = (c_hi - c_lo) * PS_MF_CHUNK_DW * 16u;
    uint32_t *Hs = H + (size_t)blockIdx.y * N * N;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const uint32_t j = tj * PS_MF_TILE + wj * 64u + (uint32_t)b * 32u + r;
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const uint32_t i = ti * PS_MF_TILE + wi * 128u + (uint32_t)a * 32u + (uint32_t)((v & 3) + 8 * (v >> 2)) + 4u * h;
                const uint32_t mism = sites - (uint32_t)acc[a][b][v];
                if (i < N && j < N) Hs[(size_t)i * N + j] = 2u * mism;
            }
        }
}

// The same counts from 3 products per site instead of 4 (round 6).  With the 2-bit allele code (h, l) as signs s_h = 1 - 2 h,
// s_l = 1 - 2 l, two alleles are equal iff (1 + s_h s_h')(1 + s_l s_l') = 4, i.e. the three products s_h s_h' + s_l s_l' +
// (s_h s_l)(s_h' s_l') add to 3 for equal alleles and to -1 for different ones: over a range of `sites` sites the contraction of
// the three +-1 features per site is S = 4 matches - sites, mismatches = (3 sites - S) / 4.  +-1.0 are E2M1 values (0b0010,
// 0b1010), products and f32 sums of at most 3 * sites < 2^24 in magnitude are exact, so the integers equal the one-hot forms'.
// Operands: the bit planes of core_packT_kernel<false, true, true>; a K-step is one feature of 32 sites per lane (the third plane
// is the xor of the two stored ones), expanded bit -> nibble by four reads of a byte -> 8-nibble table (256 entries x 64
// copies x 4 bytes = 64 KB at LDS offset 0, copy = lane: 64 different banks).  6 K-steps per 128-site chunk where the one-hot
// form takes 8: 25 % fewer matrix-core cycles and LDS bytes for the same pairs.
__device__ __forceinline__ uint32_t ps_mf_lut_read32(uint32_t raw, uint32_t colofs, uint32_t b)
{
    uint32_t addr;      // (byte b of raw) << 8 | colofs  (colofs = lane * 4 < 256)
    switch (b) {
    case 0: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0400u); break;
    case 1: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0500u); break;
    case 2: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0600u); break;
    default: addr = __builtin_amdgcn_perm(raw, colofs, 0x0c0c0700u); break;
    }
    return *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)addr;
}

__global__ void __launch_bounds__(512) core_allpairs_mfma_signed_kernel(const uint32_t *packT, uint32_t WT, uint32_t N, uint32_t *H,
                                                                        uint32_t chunks_per_range, uint32_t n_chunks, uint32_t ntile)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lut[];      // 256 entries x 64 copies x 4 bytes = 64 KB, at LDS offset 0
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t x = tid; x < 256u * 64u; x += 512u) {
        const uint32_t e = x >> 6;
        uint32_t v = 0x22222222u;        // bit k of the byte set -> nibble k = -1.0 (0b1010), clear -> +1.0 (0b0010)
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) v |= ((e >> k) & 1u) << (4u * k + 3u);
        *(uint32_t *)(lut + (size_t)x * 4u) = v;         // x = e * 64 + copy
    }
    const uint32_t colofs = lane << 2;
    uint32_t ti = 0, rem = blockIdx.x;
    while (rem >= ntile - ti) { rem -= ntile - ti; ti++; }
    const uint32_t tj = ti + rem;
    const uint32_t wi = wave >> 2, wj = wave & 3u;
    const uint32_t r = lane & 31u, h = lane >> 5;
    const uint32_t *src[6];
    const uint32_t ngroups = (N + 31u) / 32u;
#pragma unroll
    for (uint32_t f = 0; f < 6u; f++) {
        const uint32_t grp = f < 4u ? ti * (PS_MF_TILE / 32u) + wi * 4u + f : tj * (PS_MF_TILE / 32u) + wj * 2u + (f - 4u);
        src[f] = packT + (size_t)min(grp, ngroups - 1u) * (WT / PS_MF_CHUNK_DW) * 256u + lane * 4u;
    }
    ps_v16f acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[a][b][v] = 0.0f;
    const uint32_t c_lo = blockIdx.y * chunks_per_range, c_hi = min(n_chunks, c_lo + chunks_per_range);
    if (c_lo >= c_hi) return;
    uint4 cur[6], nxt[6];
#pragma unroll
    for (uint32_t f = 0; f < 6u; f++) cur[f] = *(const uint4 *)(src[f] + (size_t)c_lo * 256u);
    __syncthreads();
    // 6 K-steps per chunk: (first 32 sites: high plane, low plane, their xor), (second 32 sites: the same)
    ps_v4i opA[6], opB[6];
    auto fetch = [&](const uint4 &w, uint32_t t, ps_v4i &op) {
        const uint32_t hi = t < 3u ? w.x : w.y, lo = t < 3u ? w.z : w.w;
        const uint32_t raw = (t % 3u) == 0u ? hi : (t % 3u) == 1u ? lo : (hi ^ lo);
        op = ps_v4i{ (int)ps_mf_lut_read32(raw, colofs, 0u), (int)ps_mf_lut_read32(raw, colofs, 1u), (int)ps_mf_lut_read32(raw, colofs, 2u),
                     (int)ps_mf_lut_read32(raw, colofs, 3u) };
    };
#pragma unroll
    for (uint32_t f = 0; f < 6u; f++) fetch(cur[f], 0u, opA[f]);
    const int one = 0x7f7f7f7f;          // E8M0 block scale 2^0 in every byte
    for (uint32_t c = c_lo; c < c_hi; c++) {
        const uint32_t cn = min(c + 1u, c_hi - 1u);
#pragma unroll
        for (uint32_t f = 0; f < 6u; f++) nxt[f] = *(const uint4 *)(src[f] + (size_t)cn * 256u);
#pragma unroll
        for (uint32_t t = 0; t < 6u; t++) {
            const uint32_t tn = t == 5u ? 0u : t + 1u;
#pragma unroll
            for (uint32_t f = 0; f < 6u; f++) {
                if (t & 1u) fetch((t == 5u) ? nxt[f] : cur[f], tn, opA[f]);
                else fetch((t == 5u) ? nxt[f] : cur[f], tn, opB[f]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const ps_v4i &xa = (t & 1u) ? opB[a] : opA[a], &xb = (t & 1u) ? opB[4 + b] : opA[4 + b];
                    const ps_v8i va = { xa.x, xa.y, xa.z, xa.w, 0, 0, 0, 0 };
                    const ps_v8i vb = { xb.x, xb.y, xb.z, xb.w, 0, 0, 0, 0 };
                    acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc[a][b], 4, 4, 0, one, 0, one);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (uint32_t f = 0; f < 6u; f++) cur[f] = nxt[f];
    }
    const int32_t sites3 = (int32_t)(3u * (c_hi - c_lo) * PS_MF_CHUNK_DW * 16u);
    uint32_t *Hs = H + (size_t)blockIdx.y * N * N;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const uint32_t j = tj * PS_MF_TILE + wj * 64u + (uint32_t)b * 32u + r;
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const uint32_t i = ti * PS_MF_TILE + wi * 128u + (uint32_t)a * 32u + (uint32_t)((v & 3) + 8 * (v >> 2)) + 4u * h;
                // S = 4 matches - sites; the reference's numerator counts 2 per differing site (distances.rs:22-52)
                const uint32_t mism = (uint32_t)(sites3 - (int32_t)acc[a][b][v]) >> 2;
                if (i < N && j < N) Hs[(size_t)i * N + j] = 2u * mism;
            }
        }
}

// H[0][i][j] += H[1 ..][i][j] over the 256-tiles with ti <= tj (the only ones the kernels above store): one workgroup per
// row i, 16 bytes per lane from the row's first computed column on
__global__ void __launch_bounds__(256) core_allpairs_sum_slices_kernel(uint32_t *H, uint32_t N, uint32_t slices)
{
    const uint32_t i = blockIdx.x;
    const size_t NN = (size_t)N * N;
    uint32_t *row = H + (size_t)i * N;
    const uint32_t j_lo = (i >> 8) << 8;
    if ((N & 3u) == 0u) {
        for (uint32_t j = j_lo + 4u * threadIdx.x; j < N; j += 1024u) {
            uint4 a = *(const uint4 *)(row + j);
            for (uint32_t s = 1; s < slices; s++) {
                const uint4 b = *(const uint4 *)(row + s * NN + j);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            *(uint4 *)(row + j) = a;
        }
    } else {
        for (uint32_t j = j_lo + threadIdx.x; j < N; j += 256u) {
            uint32_t a = row[j];
            for (uint32_t s = 1; s < slices; s++) a += row[s * NN + j];
            row[j] = a;
        }
    }
}

// out[slot] = H[i][j] for the 256-tiles of core_allpairs_mfma_kernel (tiles ti <= tj were computed; a diagonal
// tile holds both orders)
__global__ void core_pair_lookup256_kernel(const uint32_t *H, uint32_t N, const uint32_t *r1, const uint32_t *r2,
                                           const uint32_t *perm, uint64_t P, uint32_t *out)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const uint32_t i = r1[k], j = r2[k];
    const bool fwd = (i >> 8) <= (j >> 8);
    out[perm ? perm[k] : k] = fwd ? H[(size_t)i * N + j] : H[(size_t)j * N + i];
}

// generic form (any N, any byte values): one thread per pair, blockIdx.y splits the sites
__global__ void __launch_bounds__(256) core_pair_counts_simple(
    const uint8_t *state, uint32_t pitch, uint32_t rows, const uint32_t *r1, const uint32_t *r2,
    const uint32_t *perm, uint64_t P, uint32_t *out, uint32_t rows_per_slice)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const uint32_t i = r1[k], j = r2[k];
    const uint32_t sb = blockIdx.y * rows_per_slice;
    const uint32_t se = min(rows, sb + rows_per_slice);
    uint32_t acc = 0;
    for (uint32_t s = sb; s < se; s++) {
        const uint8_t *row = state + (size_t)s * pitch;
        acc += __popc((uint32_t)(row[i] ^ row[j]));
    }
    if (acc) atomicAdd(&out[perm ? perm[k] : k], acc);
}

// dst[k] += src[k]: partial Hamming numerators of the site shards of one process (ps_multi)
__global__ void __launch_bounds__(256) u32_add_kernel(uint32_t *dst, const uint32_t *src, uint64_t n)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) dst[k] += src[k];
}

// distances.rs:22-52 / :55-77 on two byte slices already in device memory
__global__ void __launch_bounds__(256) slice_counts_kernel(const uint8_t *x, const uint8_t *y,
                                                           uint64_t n, uint32_t *out3)
{
    uint32_t h = 0, in = 0, un = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t a = x[t], b = y[t];
        h += __popc(a ^ b);
        in += __popc(a & b);
        un += __popc(a | b);
    }
    for (int off = 32; off > 0; off >>= 1) {
        h += __shfl_down(h, off, 64);
        in += __shfl_down(in, off, 64);
        un += __shfl_down(un, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (h) atomicAdd(&out3[0], h);
        if (in) atomicAdd(&out3[1], in);
        if (un) atomicAdd(&out3[2], un);
    }
}
