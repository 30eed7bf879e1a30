// ps_common.h -- Philox4x32-10, RNG stream ids and the keyed dense plans shared by
// the host code and the gfx950 kernels of libpansim_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PS_HD __host__ __device__ __forceinline__

// RNG streams (Philox counter word 3; word 2 is the generation)
enum : uint32_t {
    PS_STREAM_CORE_L1 = 1,    // ctr = (site/2, individual/16, gen): symbol planes 0-3 of 2 sites x 16 individuals
    PS_STREAM_CORE_L2 = 2,    // ctr = (site, individual, gen): the residual word + donor
    PS_STREAM_ACC_MUT = 3,    // ctr = (gene/4, individual, gen): 4 flip words
    PS_STREAM_HGT = 4,        // ctr = (event / 2, donor, gen), | compartment << 8: (x, y) event 2m, (z, w) event 2m + 1
    PS_STREAM_CORE_L1B = 5,   // ctr = (site/4, individual/16, gen): symbol planes 4-5 of 4 sites x 16 individuals
    PS_STREAM_INIT_CORE = 16, // host sequential streams: ctr = (n lo, n hi, gen)
    PS_STREAM_INIT_ACC = 17,
    PS_STREAM_SELECTION = 18,
    PS_STREAM_PAIRS = 19,
    PS_STREAM_PARENTS = 20,
    PS_STREAM_HGT_COUNT = 21
};

struct ps_u4 { uint32_t x, y, z, w; };

PS_HD uint32_t ps_mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

// a ^ b ^ c: one v_bitop3_b32 on gfx950 (truth table 0x96) instead of two v_xor_b32
PS_HD uint32_t ps_xor3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__) && __has_builtin(__builtin_amdgcn_bitop3_b32)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

template <int ROUNDS>
PS_HD ps_u4 ps_philox_r(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = ps_xor3((uint32_t)(p1 >> 32), c1, k0);
        uint32_t n2 = ps_xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    ps_u4 o = { c0, c1, c2, c3 };
    return o;
}

// Philox4x32-10: every stream of the build
PS_HD ps_u4 ps_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    return ps_philox_r<10>(c0, c1, c2, c3, k0, k1);
}

// Level 1 of the core cell plan: a 6-bit SYMBOL per cell, bit-sliced.  Plane k (0..5) holds bit k of the
// symbols; a plane word covers the 16 individuals of one chunk at two consecutive sites, cell (i, site) at bit
// ps_plane_pos = 8 * (i % 4) + (i / 4) % 4 + 4 * (site % 2): the four individuals 4j .. 4j + 3 -- one dword of a site row --
// sit at the same bit of the four BYTES of the word, so that a shift and a mask turn plane bits into byte-lane flags.
//   block A = Philox(site / 2, i / 16, gen, CORE_L1):  (x, y, z, w) = planes 0, 1, 2, 3 of the sites 2 * (site / 2) + {0, 1}
//   block B = Philox(site / 4, i / 16, gen, CORE_L1B): x / y = plane 4 of the sites 4 * (site / 4) + {0, 1} / {2, 3},
//                                                      z / w = plane 5 of the same
// i.e. three Philox blocks per 4 sites x 16 individuals -- 6 bits per cell, every bit used once -- and a lane of the sweeps,
// which owns 16 individuals of 4 consecutive sites per trip, computes exactly the blocks it consumes.
PS_HD uint32_t ps_plane_pos(uint32_t ind, uint32_t site) { return 8u * (ind & 3u) + ((ind >> 2) & 3u) + 4u * (site & 1u); }
PS_HD ps_u4 ps_philox_l1a(uint32_t site2, uint32_t chunk, uint32_t gen, uint32_t k0, uint32_t k1)
{
    return ps_philox(site2, chunk, gen, PS_STREAM_CORE_L1, k0, k1);
}
PS_HD ps_u4 ps_philox_l1b(uint32_t site4, uint32_t chunk, uint32_t gen, uint32_t k0, uint32_t k1)
{
    return ps_philox(site4, chunk, gen, PS_STREAM_CORE_L1B, k0, k1);
}

// n-th f64 of a seeded host stream: two per Philox block, top 53 bits x 2^-53
PS_HD double ps_hs_f64(uint32_t k0, uint32_t k1, uint32_t stream, uint32_t gen, uint64_t n)
{
    const uint64_t blk = n >> 1;
    const ps_u4 w = ps_philox((uint32_t)blk, (uint32_t)(blk >> 32), gen, stream, k0, k1);
    const uint64_t x = (n & 1) ? (((uint64_t)w.w << 32) | w.z) : (((uint64_t)w.y << 32) | w.x);
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

// Core cell plan.  Symbol s = 4 * n + t, n = planes 0-3 (bit k of n = plane k), t = planes 4-5:
//   s < k, < 2k, < 3k    : mutate to 2, 4, 8 -- decided by the symbol alone          (population.rs:511-540)
//   3k <= s < 3k + R     : RESIDUAL -- one 32-bit word u (level 2) against seven cumulative thresholds:
//       u < T[0],T[1],T[2] : mutate to 2,4,8
//       u < T[3],T[4],T[5] : mutate to 2,4,8 AND receive a donor allele
//       u < T[6]           : receive a donor allele only                              (population.rs:544-751)
//   otherwise            : nothing happens to the cell
// k = floor(64 a) symbols carry 1/64 each of an allele's mutate-only mass a; what is left of the event mass lives, scaled by
// 64 / R, in the R residual symbols.  A cell whose n has a bit at or above cshift cannot hold an event (3k + R <= 4 << cshift).
struct ps_core_plan {
    uint32_t T[7];
    uint32_t has_events;
    uint32_t k, R;
    uint32_t cshift;      // 0..4 (4: every cell is a candidate)
};

// what a symbol decides: 2 / 4 / 8 = that allele, 1 = residual, 0 = nothing
PS_HD uint32_t ps_sym_code(uint32_t s, const ps_core_plan &pl)
{
    if (s < pl.k) return 2u;
    if (s < 2u * pl.k) return 4u;
    if (s < 3u * pl.k) return 8u;
    return (s < 3u * pl.k + pl.R) ? 1u : 0u;
}

// the symbol of cell (site, individual) out of its two blocks
PS_HD uint32_t ps_cell_nibble(const ps_u4 &A, uint32_t site, uint32_t ind)
{
    const uint32_t pos = ps_plane_pos(ind, site);
    return ((A.x >> pos) & 1u) | (((A.y >> pos) & 1u) << 1) | (((A.z >> pos) & 1u) << 2) | (((A.w >> pos) & 1u) << 3);
}
PS_HD uint32_t ps_cell_pair(const ps_u4 &B, uint32_t site, uint32_t ind)
{
    const uint32_t pos = ps_plane_pos(ind, site);
    const uint32_t p4 = (site & 2u) ? B.y : B.x, p5 = (site & 2u) ? B.w : B.z;
    return ((p4 >> pos) & 1u) | (((p5 >> pos) & 1u) << 1);
}

struct ps_cell { uint32_t mut; uint32_t hr; };

// classify the level-2 word of a residual cell against the plan
PS_HD ps_cell ps_classify(uint32_t u, const ps_core_plan &pl)
{
    ps_cell o = { 0u, 0u };
    if (u >= pl.T[6]) return o;
    if (u < pl.T[2]) {
        o.mut = (u < pl.T[0]) ? 2u : (u < pl.T[1]) ? 4u : 8u;
    } else if (u < pl.T[5]) {
        o.mut = (u < pl.T[3]) ? 2u : (u < pl.T[4]) ? 4u : 8u;
        o.hr = 1u;
    } else {
        o.hr = 1u;
    }
    return o;
}

#define PS_MAX_COMP 2
struct ps_acc_plan {
    int32_t n_comp;
    uint32_t comp_begin[PS_MAX_COMP], comp_end[PS_MAX_COMP];
    uint32_t flip_thr[PS_MAX_COMP];   // M-acc: flip iff word < thr (population.rs:486-510)
    double lam_rec[PS_MAX_COMP];      // HGT events per donor (population.rs:555)
};
