/*
 * pansim_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C restatement of the per-generation hot path of bacpop/Pansim
 * (reference files: pansim/src/population.rs, pansim/src/distances.rs,
 * pansim/src/main.rs).  It exists to CHECK the HIP product path; it is never
 * shipped, never linked into libpansim_hip.so and never imported by the
 * product package.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it.
 *
 * Storage follows the reference: one row per individual, row-major u8
 * (`Array2<u8>` shape (N, ncols), population.rs:164-178).
 *
 * PARITY STATUS
 *  - deterministic functions (distances, gene frequencies, gather, weight
 *    arithmetic, writers, parameter derivation) restate the reference line by
 *    line and are pinned by the hand-derived known-answer vectors of
 *    SURVEY.md section 8(c) (tests/golden/kat.json).  The reference ships no
 *    tests, fixtures or golden vectors of its own, and cannot be built here
 *    (no Rust toolchain), so these are the only pins that exist.
 *  - third-party arithmetic that is NOT under /root/reference (rand 0.8.5,
 *    statrs 0.16, rand_distr 0.4, logsumexp 0.1; versions from
 *    pansim/Cargo.toml:10-17, no Cargo.lock) is "parity unpinned": only the
 *    published distributions are contractual.  The reference draws mutation
 *    and recombination randomness from thread_rng() (population.rs:493, :517,
 *    :596) and is itself not reproducible under --seed.
 *  - stochastic operators are restated in the keyed dense form of DESIGN.md
 *    section 3 (Philox4x32-10 keyed on seed/generation/individual/site), which
 *    is what the HIP kernels implement; oracle <-> HIP must be bit-exact.
 *  - orc_ref_* functions restate the reference's event-driven algorithm
 *    (Poisson count per row, weighted-index binary search per event, serial
 *    gather, serial scatter) with a sequential generator; they are the
 *    "port" CPU baseline and the distributional cross-check.
 */
#ifndef PANSIM_ORACLE_H
#define PANSIM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG streams ---------------------------------- */
enum {
    ORC_STREAM_CORE_L1 = 1,
    ORC_STREAM_CORE_L2 = 2,
    ORC_STREAM_ACC_MUT = 3,
    ORC_STREAM_HGT = 4,
    ORC_STREAM_CORE_L1B = 5,
    ORC_STREAM_INIT_CORE = 16,
    ORC_STREAM_INIT_ACC = 17,
    ORC_STREAM_SELECTION = 18,
    ORC_STREAM_PAIRS = 19,
    ORC_STREAM_PARENTS = 20,
    ORC_STREAM_HGT_COUNT = 21
};

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* n-th f64 / u32 of host stream (seed, stream, gen): stateless, indexed. */
double orc_hs_f64(uint64_t seed, uint32_t stream, uint32_t gen, uint64_t n);
uint32_t orc_hs_u32(uint64_t seed, uint32_t stream, uint32_t gen, uint64_t n);
/* Poisson(mean) from host stream (seed, stream, gen), Knuth product for mean<10
 * else Hoermann PTRS; *n_used returns how many f64 draws were consumed. */
uint64_t orc_poisson(double mean, uint64_t seed, uint32_t stream, uint32_t gen, uint64_t *n_used);
/* Poisson by inversion over an integer threshold table (HGT event counts per donor) */
uint32_t orc_poisson_table(double lambda, uint32_t *kmin_out, uint32_t *thr, uint32_t cap);
uint32_t orc_poisson_from_table(uint32_t u, uint32_t kmin, const uint32_t *thr, uint32_t len);

/* ---- parameter derivation: main.rs:259-367 ------------------------------ */
typedef struct {
    uint64_t pop_size, core_size, pan_genes, core_genes;
    double avg_gene_freq, HR_rate, HGT_rate, core_mu;
    double rate_genes1, rate_genes2, prop_genes2;
} orc_params;

typedef struct {
    uint64_t pan_size;           /* G = pan_genes - core_genes            main.rs:259 */
    double avg_gene_freq_adj;    /* main.rs:263-268 */
    int32_t avg_gene_num;        /* main.rs:272 */
    double n_core_mutations;     /* main.rs:275-276 */
    double n_recombinations_core;/* main.rs:279 */
    double n_recombinations_pan_total; /* main.rs:280 */
    int n_comp;                  /* compartments actually pushed main.rs:341,355 */
    uint64_t comp_begin[2], comp_end[2]; /* gene ranges main.rs:342-345,356-359 */
    double n_pan_mutations[2];   /* main.rs:348,361 */
    double n_recombinations_pan[2]; /* main.rs:349-351,364-366 */
} orc_derived;

void orc_derive(const orc_params *p, orc_derived *d);

/* ---- keyed dense plans --------------------------------------------------- */
typedef struct {
    uint32_t T[7];      /* cumulative 32-bit thresholds of a RESIDUAL cell's level-2 word, DESIGN.md section 3.2 */
    uint32_t has_events;
    uint32_t k;         /* 6-bit symbols s < k, < 2k, < 3k mutate to 2, 4, 8 */
    uint32_t R;         /* symbols 3k <= s < 3k + R are residual */
    uint32_t cshift;    /* a cell whose symbol's high part n = s / 4 has a bit at or above cshift holds no event */
} orc_core_plan;
void orc_core_plan_make(double lam_mut, double lam_hr, uint64_t L, orc_core_plan *plan);
/* flip threshold for one accessory compartment, DESIGN.md section 3.3 */
uint32_t orc_acc_flip_threshold(double lam, uint64_t n_genes_in_comp);

/* ---- initial state: population.rs:181-242 (clonal start) ---------------- */
void orc_init_core_vec(uint64_t seed, uint64_t L, uint8_t *allele_vec);
void orc_init_acc_vec(uint64_t seed, uint64_t G, double avg_gene_freq_adj, uint8_t *acc_vec);
void orc_replicate(const uint8_t *vec, uint64_t N, uint64_t ncols, uint8_t *pop);
/* selection coefficients main.rs:287-319; returns draws used */
uint64_t orc_selection_coefficients(uint64_t seed, uint64_t G, double prop_positive,
                                    double pos_lambda, double neg_lambda, double *out);
/* sampled pair list main.rs:413-427 */
void orc_sample_pairs(uint64_t seed, uint64_t N, uint64_t P, uint32_t *range1, uint32_t *range2);

/* ---- deterministic reference functions ---------------------------------- */
uint32_t orc_hamming_bitwise_fast(const uint8_t *x, const uint8_t *y, size_t n); /* distances.rs:22-52 */
void orc_jaccard_distance_fast(const uint8_t *x, const uint8_t *y, size_t n,
                               uint32_t *inter, uint32_t *uni);                   /* distances.rs:55-77 */
void orc_pairwise_distances(const uint8_t *pop, uint64_t N, uint64_t ncols, int core,
                            uint64_t core_genes, uint64_t P, const uint32_t *r1,
                            const uint32_t *r2, double *out);                     /* population.rs:787-837 */
/* integer form of the core distance numerators over a site range (for sharding) */
void orc_pairwise_hamming_counts(const uint8_t *pop, uint64_t N, uint64_t ncols,
                                 uint64_t col_begin, uint64_t col_end, uint64_t P,
                                 const uint32_t *r1, const uint32_t *r2, uint32_t *out);
void orc_average_distance(const uint8_t *pop, uint64_t N, uint64_t ncols, int core,
                          uint64_t core_genes, double *out);                      /* population.rs:753-784 */
void orc_gene_frequencies(const uint8_t *pop, uint64_t N, uint64_t G, uint64_t core_genes,
                          double *out /* G + core_genes */);                      /* population.rs:840-863 */
double orc_calc_gene_freq(const uint8_t *pop, uint64_t N, uint64_t ncols);        /* population.rs:244-268 */
void orc_next_generation(const uint8_t *pop, uint64_t N, uint64_t ncols,
                         const uint32_t *sample, uint8_t *next);                  /* population.rs:450-465 */
void orc_standard_deviation(const double *v, uint64_t n, double *std, double *mean); /* population.rs:87-94 */
char orc_int_to_base(uint8_t n);                                                  /* population.rs:154-162 */

/* fitness terms: population.rs:282-322 (num_genes, log_sum with the -inf reset) */
void orc_fitness_terms(const uint8_t *pop, uint64_t N, uint64_t G, const double *sel_coeff,
                       int32_t *num_genes, double *logw);
/* weights: population.rs:293-437.  returns 0 ok, <0 invalid (reference would panic) */
int orc_sample_weights(const int32_t *num_genes, const double *logw, uint64_t N, uint64_t G,
                       int32_t avg_gene_num, const double *avg_pairwise_dists,
                       int no_control_genome_size, double genome_size_penalty,
                       double competition_strength, double *weights);
/* draws: population.rs:440-443 with the build's parent stream */
int orc_draw_parents(const double *weights, uint64_t N, uint64_t seed, uint32_t gen, uint32_t *idx);
int orc_sample_indices(const uint8_t *pop, uint64_t N, uint64_t G, uint64_t seed, uint32_t gen,
                       int32_t avg_gene_num, const double *avg_pairwise_dists,
                       const double *sel_coeff, int no_control_genome_size,
                       double genome_size_penalty, double competition_strength, uint32_t *idx);

/* ---- stochastic operators, keyed dense form ------------------------------ */
/* site_offset: global index of column 0 (site sharding keeps global keys) */
void orc_mutate_core(uint8_t *pop, uint64_t N, uint64_t L, uint64_t site_offset, uint64_t seed,
                     uint32_t gen, const orc_core_plan *plan);   /* population.rs:511-540 */
void orc_recombine_core(uint8_t *pop, uint64_t N, uint64_t L, uint64_t site_offset, uint64_t seed,
                        uint32_t gen, const orc_core_plan *plan); /* population.rs:544-751 core */
void orc_mutate_acc(uint8_t *pop, uint64_t N, uint64_t G, uint64_t seed, uint32_t gen, int n_comp,
                    const uint64_t *comp_begin, const uint64_t *comp_end,
                    const double *lambdas);                      /* population.rs:486-510 */
/* returns total events drawn over all compartments */
uint64_t orc_recombine_acc(uint8_t *pop, uint64_t N, uint64_t G, uint64_t seed, uint32_t gen,
                           int n_comp, const uint64_t *comp_begin, const uint64_t *comp_end,
                           const double *lambdas);               /* population.rs:544-751 acc */

/* ---- writers: population.rs:865-897, main.rs:471-498 -------------------- */
/* Rust `{}` Display for f64: shortest round-trip, never exponent form. returns strlen */
int orc_fmt_f64(double v, char *buf, size_t cap);
int orc_write_matrix(const uint8_t *pop, uint64_t N, uint64_t ncols, int core, uint64_t core_genes,
                     const char *outpref);

/* ---- reference-algorithm (event-driven) mode: the CPU baseline ---------- */
typedef struct orc_ref_sim orc_ref_sim;
orc_ref_sim *orc_ref_create(const orc_params *p, uint64_t seed, int threads);
void orc_ref_destroy(orc_ref_sim *s);
/* one generation of main.rs:429-464 (select, gather x2, mutate x2, HR, HGT) */
int orc_ref_generation(orc_ref_sim *s, uint32_t gen);
void orc_ref_set_competition(orc_ref_sim *s, double strength);   /* main.rs:438-440 */
const uint8_t *orc_ref_core(const orc_ref_sim *s);
const uint8_t *orc_ref_acc(const orc_ref_sim *s);
/* threaded pairwise distances (population.rs:797-799 is a par_iter) */
void orc_ref_pairwise(const orc_ref_sim *s, int core, uint64_t P, const uint32_t *r1,
                      const uint32_t *r2, double *out);

#ifdef __cplusplus
}
#endif
#endif
