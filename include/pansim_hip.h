/*
 * pansim_hip.h -- C ABI of libpansim_hip.so, the MI355X (gfx950) drop-in for the
 * per-generation hot path of bacpop/Pansim.
 *
 * The reference has no FFI layer; its library seam is `pub mod population`
 * (pansim/src/lib.rs:4) consumed by main() (pansim/src/main.rs:8).  Every entry
 * point below replaces one item of that seam and cites it.  A Rust host binds
 * this header with `extern "C"` declarations (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - every function returns 0 on success and a negative ps_status on failure;
 *    ps_last_error() returns the message of the calling thread's last failure.
 *    Nothing aborts or throws across the boundary (the reference panics:
 *    population.rs:440, :484, :562, :584).
 *  - the caller owns every buffer it passes; the library never keeps a caller
 *    pointer after the call returns.  A handle owns its HBM state.
 *  - handles are not thread-safe: one host thread per handle, as all reference
 *    methods take `&mut self`.
 *  - matrices cross the boundary in the reference's layout: one row per
 *    individual, row-major u8 (`Array2<u8>` (N, ncols), population.rs:164-178).
 *    In HBM the core matrix is site-major u8 and the accessory matrix is
 *    bit-packed.
 *  - there is no CPU fallback: without a HIP device every compute call fails
 *    with PS_ERR_NO_DEVICE.
 */
#ifndef PANSIM_HIP_H
#define PANSIM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PS_OK = 0,
    PS_ERR_INVALID = -1,    /* bad argument (reference: unwrap()/assert panic) */
    PS_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime failure */
    PS_ERR_OOM = -3,
    PS_ERR_WEIGHTS = -4,    /* WeightedIndex::new would panic (population.rs:440) */
    PS_ERR_IO = -5,
    PS_ERR_STATE = -6       /* call sequence invalid (e.g. rates not set) */
} ps_status;

const char *ps_last_error(void);
/* library ABI version (bumped on any signature change) */
int ps_abi_version(void);
/* number of visible HIP devices, <0 on runtime failure.  Does not create a context. */
int ps_device_count(void);

/* ------------------------------------------------------------------------ */
/* struct Population (population.rs:164-170)                                 */
/* ------------------------------------------------------------------------ */
typedef struct ps_population ps_population;

typedef struct {
    uint64_t pop_size;     /* N: `size` of Population::new (population.rs:182) */
    uint64_t ncols;        /* `allele_count` held by THIS handle (a site shard for core) */
    uint64_t global_cols;  /* full core_size / pan_size (== ncols when not sharded) */
    uint64_t col_offset;   /* global index of local column 0 (site sharding) */
    uint64_t core_genes;   /* population.rs:188 */
    uint64_t seed;         /* --seed (main.rs:180); Philox key */
    int32_t core;          /* population.rs:185: 1 = core alleles, 0 = accessory */
    int32_t device;        /* HIP device ordinal, -1 = current device */
} ps_config;

/* Population::new (population.rs:181-242).  The reference draws ONE random
 * vector and copies it to every individual (clonal start, :206-229); here the
 * caller passes that vector (`init_vec`, ncols bytes; core: 1/2/4/8, accessory:
 * 0/1).  ps_init_vector() below draws it from the build's seeded host stream. */
int ps_population_create(const ps_config *cfg, const uint8_t *init_vec, ps_population **out);
void ps_population_destroy(ps_population *p);

/* Draw the clonal start vector of Population::new (population.rs:199-219):
 * core: 1 << uniform{0..3}; accessory: uniform() < avg_gene_freq.  Columns
 * [col_offset, col_offset+ncols) of the global vector are returned. */
int ps_init_vector(uint64_t seed, int core, uint64_t col_offset, uint64_t ncols,
                   double avg_gene_freq, uint8_t *out);

/* Replace / read the whole matrix, individual-major u8 (N x ncols).  The
 * reference has no such call (its field is private); tests use it to run the
 * deterministic operators on identical state. */
int ps_load_matrix(ps_population *p, const uint8_t *rows);
int ps_read_matrix(ps_population *p, uint8_t *rows);

/* The per-compartment rates the reference passes at every call:
 * `mutations_vec` of mutate_alleles (population.rs:469; main.rs:275-276, :348,
 * :361) and `recombinations_vec` of recombine (population.rs:546; main.rs:279,
 * :349-351, :364-366), with the gene range of each compartment's weight mask
 * (main.rs:342-345, :356-359; core: one compartment [0, global_cols)).  They are
 * set once because the keyed dense form decides mutation and recombination of a
 * cell from one random word. */
int ps_set_rates(ps_population *p, int n_comp, const double *lam_mut, const double *lam_rec,
                 const uint64_t *comp_begin, const uint64_t *comp_end);

/* Population::next_generation(&sample) (population.rs:450-465) */
int ps_next_generation(ps_population *p, const uint32_t *sample);
/* Population::mutate_alleles (population.rs:467-542), generation = loop index j of main.rs:429 */
int ps_mutate_alleles(ps_population *p, uint32_t generation);
/* Population::recombine (population.rs:544-751) */
int ps_recombine(ps_population *p, uint32_t generation);
/* HGT donors sharded over the ranks / shards of one run (the exchange step of the path).  Events are
 * keyed per donor and the recipient's bit is ORed (population.rs:632: the value is always 1), so any partition of the
 * donors gives the unsharded result: this handle generates the events of donors [N r / K, N (r + 1) / K) into a delta
 * buffer of `n_words` u64 (the individual-major bit matrix, zero padded), calls `fn` -- which must leave the bitwise
 * OR of all shards' buffers in every shard's buffer, ordered on `hip_stream` (hipStream_t) -- and ORs the result into
 * its replica of the matrix.  Providers: ps_multi (device-to-device reads between the shards of one process),
 * pansim_amd/distributed.py (torch.distributed: all-to-all + all-gather over RCCL), or any host's own transport.
 * fn == NULL applies the own donors' events only; shard_count == 1 switches sharding off. */
typedef int (*ps_exchange_fn)(void *ctx, void *d_words, uint64_t n_words, void *hip_stream);
int ps_set_donor_shard(ps_population *acc, uint32_t shard_rank, uint32_t shard_count, ps_exchange_fn fn, void *ctx);
/* Fused next_generation + mutate_alleles + recombine in one pass over HBM
 * (main.rs:445-464 for one matrix); bit-identical to the three calls in order.
 * do_recombine mirrors the `HR_rate > 0.0` / `HGT_rate > 0.0` guards (main.rs:459-464). */
int ps_step(ps_population *p, uint32_t generation, const uint32_t *sample, int do_recombine);

/* Population::sample_indices (population.rs:270-448) on the accessory matrix.
 * avg_pairwise_dists: N values (main.rs:435-440).  out_idx: N parent indices. */
int ps_sample_indices(ps_population *acc, uint32_t generation, int32_t avg_gene_num,
                      const double *avg_pairwise_dists, const double *selection_coefficients,
                      int verbose, int no_control_genome_size, double genome_size_penalty,
                      double competition_strength, uint32_t *out_idx);
/* The device half of sample_indices: num_genes (population.rs:282-291) and the
 * per-row log-fitness with the -inf reset (population.rs:299-322). */
int ps_fitness_terms(ps_population *acc, const double *selection_coefficients,
                     int32_t *num_genes, double *logw);
/* The host half: weights (population.rs:293-437), then draws (:440-443). */
int ps_sample_weights(const int32_t *num_genes, const double *logw, uint64_t n, uint64_t n_genes,
                      int32_t avg_gene_num, const double *avg_pairwise_dists,
                      int no_control_genome_size, double genome_size_penalty,
                      double competition_strength, double *weights);
int ps_draw_parents(const double *weights, uint64_t n, uint64_t seed, uint32_t generation,
                    uint32_t *out_idx);

/* Population::average_distance (population.rs:753-784) */
int ps_average_distance(ps_population *p, double *out);
/* Rows [first, first + count) of average_distance (count values): what one rank of a row-sharded D-avg computes -- every
 * individual's mean is a sum over ALL others in ascending order (:770), so the rows are independent and a run splits
 * them over its ranks and all-gathers the N doubles. */
int ps_average_distance_rows(ps_population *p, uint64_t first, uint64_t count, double *out);
/* Population::pairwise_distances (population.rs:787-837) */
int ps_pairwise_distances(ps_population *p, uint64_t max_distances, const uint32_t *range1,
                          const uint32_t *range2, double *out);
/* Integer numerators of pairwise_distances for this handle's columns: core:
 * out_a = sum popcount(x^y) (distances.rs:22-52, before the /2 of
 * population.rs:817); accessory: out_a = intersection, out_b = union
 * (distances.rs:55-77).  out_is_device != 0: out_a/out_b are device pointers
 * (site-sharded runs all-reduce them over RCCL). */
int ps_pairwise_counts(ps_population *p, uint64_t max_distances, const uint32_t *range1,
                       const uint32_t *range2, uint32_t *out_a, uint32_t *out_b,
                       int out_is_device);
/* Which kernel form the last core pair-count call of this handle ran in (bench.py prices the distance phase
 * against the roofline of that form): */
enum {
    PS_PAIR_FORM_NONE = 0,
    PS_PAIR_FORM_TILED2 = 1,     /* sampled pairs compared from LDS tiles of the 2-bit packed matrix (VALU bound) */
    PS_PAIR_FORM_ALLPAIRS = 2,   /* all-pairs register tiles, xor + popcount on nibble strings (VALU bound), then lookup */
    PS_PAIR_FORM_TILED4 = 3,     /* sampled pairs from LDS tiles of nibble strings */
    PS_PAIR_FORM_ROWS = 4,       /* matrix transposed once to bit strings, two strings streamed per pair (HBM bound) */
    PS_PAIR_FORM_SIMPLE = 5,     /* one thread per pair on the byte matrix (matrices with bytes above 15) */
    PS_PAIR_FORM_ALLPAIRS_MFMA = 6, /* all-pairs one-hot X X^T on the i8 matrix cores (exact i32 counts), then lookup */
    PS_PAIR_FORM_ALLPAIRS_MFMA_FP4 = 7, /* the same on the block-scaled FP4 path (E2M1 {0, 1}, scales 2^0; exact f32 counts) */
    PS_PAIR_FORM_ALLPAIRS_MFMA_SIGNED = 8 /* FP4 path, three +-1 features per site instead of four {0, 1}: S = 4 matches - sites */
};
int ps_last_pair_form(ps_population *p);
/* Which kernel the last core sweep of this handle (ps_step, ps_next_generation, ps_mutate_alleles, ps_recombine, a
 * generation of ps_sim_run) was launched as -- the library chooses by population width, rates, parent order and the room
 * for a second buffer; bench.py labels and prices its roofline line from this: */
enum {
    PS_SWEEP_FORM_NONE = 0,
    PS_SWEEP_FORM_WAVE = 1,        /* core_sweep_wave_kernel, one wave per site row (N <= 1024) */
    /* 2: a former build of the wave sweep (rounds 3-5); never reported now */
    PS_SWEEP_FORM_WINDOW = 3,      /* core_sweep_window_kernel: N > 1024, children in ascending parent order, out of place */
    PS_SWEEP_FORM_BLOCK = 4,       /* core_sweep_block_kernel: N > 1024, whole rows in workgroup-shared LDS, in place */
    PS_SWEEP_FORM_INLINE = 5       /* core_sweep_inline_kernel: the queue-free form for any rates */
};
int ps_last_sweep_form(ps_population *p);
/* Population::gene_frequencies (population.rs:840-863): ncols + core_genes values */
int ps_gene_frequencies(ps_population *p, double *out);
/* Population::calc_gene_freq (population.rs:244-268) */
int ps_calc_gene_freq(ps_population *p, double *out);
/* Population::write (population.rs:865-897): <outpref>_core_genome.csv / _pangenome.csv */
int ps_write(ps_population *p, const char *outpref);
/* wait for all queued device work of this handle */
int ps_sync(ps_population *p);
/* Launch tuning / test hooks (no reference counterpart).  Keys: "sweep_blocks_per_cu"
 * (resident 256-thread blocks per CU of the wave-per-row sweep, 1..8), "sweep_rows"
 * (2..4; accepted and ignored since round 6: a wave of that sweep takes the 4 sites of one level-1 block group per iteration),
 * "force_block_sweep" (0/1: use the block sweep even when a row fits one wavefront),
 * "force_inline_sweep" (0/1: use the queue-free inline block sweep), "pair_mode" (core
 * distances: 0 = choose by cost, 1 = sampled-pair kernel, 2 = all-pairs tiles + lookup, 3 = sampled-pair
 * kernel in its nibble form even for one-hot matrices, 4 = transposed bit strings streamed per pair -- the
 * sampled form of populations too wide for an LDS tile, 5 = all-pairs xor + popcount tiles even for one-hot matrices,
 * 6 = all pairs on the i8 matrix cores, 7 = all pairs on the FP4 path with three +-1 features per site; one-hot matrices go to the
 * matrix cores in modes 0 and 2 (FP4 form on one-hot nibbles), 6 and 7), "pair_ranges" (site ranges of the tiled
 * sampled-pair kernels, 0 = choose; the 16-bit counter cap still applies),
 * "davg_form" (average_distance: 0 = choose, 1 = LDS-tile popcount kernels, 2 = intersections on the matrix cores in one kernel --
 * the choice above pop_size 53248 --, 3 = the same in two phases, u16 counts then division + ordered fold -- the choice for row
 * shards and for 8192 < pop_size <= 53248), "davg_nb" (matrix-core forms: 32-individual fragments per wave, 0 = choose, 1, 2, or --
 * two-phase form only; the one-kernel form then chooses by itself -- 4), "davg_ib" (two-phase form: individuals per workgroup of the
 * division + fold phase, 0 = choose, 16 or 32),
 * "hgt_mode" (accessory recombination: 0 = choose, 1 = one atomic per event, 2 = two passes: bin
 * by recipient partition, OR in LDS images), "hgt_slices" (binned HGT: event slices, 0 = choose), "hgt_list_in_global" (0/1: donor gene lists in
 * global scratch instead of LDS; "hgt_bin_list_in_global": the same for the bin pass of the binned form),
 * "hgt_events_per_thread" (light HGT kernel inside the generation loop: events a thread handles in sequence -- the launch is that
 * narrow; 0 = whole chip; ps_sim sets it from the estimated sweep time),
 * "hgt_apply_threads" (binned HGT: threads per workgroup of the LDS-image pass, 256 / 512 / 1024), "window_blocks_per_cu" (window sweep:
 * workgroups per CU, 0 = choose), "davg_plain_division" (matrix-core D-avg: the compiler's f64 division in the epilogue),
 * "hgt_bin_cap" (tests: the bins of the binned HGT hold at most this many events; the rest take the overflow image),
 * "sweep_queue_cap" (tests: the sweeps treat their candidate queues and HR lists as this short, so that the queue-free
 * redo of a batch / row group -- what a full queue falls back to -- runs; 0 = real size),
 * "sweep_out_of_place" (core sweeps: -1 = choose, 0 = update the matrix in place, 1 = write the new generation to a second
 * buffer that then swaps roles with the first, 2 = the same with nontemporal row loads and stores; results are identical),
 * "lds_limit" (bytes of LDS a workgroup may use), "block_waves" (block sweep: waves per
 * workgroup, 0 = choose), "block_batch" (block sweep: segments per wave batch, 0 = choose, 2 or 4),
 * "no_block_preload" (block sweep: parent indices re-read per batch). */
int ps_set_tuning(ps_population *p, const char *key, int64_t value);

/* ------------------------------------------------------------------------ */
/* free functions of the seam                                                */
/* ------------------------------------------------------------------------ */
/* distances.rs:22-52 and :55-77, computed on the device from host slices */
int ps_hamming_bitwise_fast(const uint8_t *x, const uint8_t *y, size_t n, uint32_t *out);
int ps_jaccard_distance_fast(const uint8_t *x, const uint8_t *y, size_t n, uint32_t *inter,
                             uint32_t *uni);
/* population.rs:87-94: returns (std, mean), population sigma */
int ps_standard_deviation(const double *values, uint64_t n, double *std_out, double *mean_out);
/* population.rs:154-162 */
char ps_int_to_base(uint8_t n);
/* Rust `{}` Display of f64 as used by every writer (main.rs:328, :481, :496, :546) */
int ps_fmt_f64(double v, char *buf, size_t cap);
/* The Poisson sampler behind the per-donor HGT event counts (population.rs:562, :599 use statrs'
 * Poisson; only the distribution is contractual): thr[j] = floor(P(K <= kmin + j) * 2^32) over
 * lambda +- (12 sigma + 12); a 32-bit uniform u gives kmin + #{j : thr[j] <= u}.  Returns the
 * table length, 0 if lambda <= 0 or cap is too small. */
uint32_t ps_poisson_table(double lambda, uint32_t *kmin_out, uint32_t *thr, uint32_t cap);

/* ------------------------------------------------------------------------ */
/* main() as a library: parameter derivation and the generation loop         */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint64_t pop_size, core_size, pan_genes, core_genes;      /* main.rs:155-162 */
    double avg_gene_freq, HR_rate, HGT_rate;                   /* :163-165 */
    int32_t n_gen;                                             /* :166-167 */
    uint64_t max_distances;                                    /* :169 */
    double core_mu, rate_genes1, rate_genes2, prop_genes2;     /* :170-173 */
    double prop_positive, pos_lambda, neg_lambda;              /* :174-176 */
    uint64_t seed;                                             /* :180 */
    int32_t print_dist, print_matrices, print_selection, verbose; /* :178-183 */
    int32_t no_control_genome_size;                            /* :184 */
    double genome_size_penalty, competition_strength;          /* :185-186 */
    /* site sharding: this process holds core sites
     * [core_size*shard_rank/shard_count, core_size*(shard_rank+1)/shard_count) */
    int32_t shard_rank, shard_count;
    int32_t device;                                            /* HIP device, -1 = current */
    /* opt-in, UNPINNED: draw what precedes sample_beta -- the selection coefficients, main.rs:289-319 --
     * from the reference's own seeded stream (rand 0.8.5 StdRng = ChaCha12 after seed_from_u64, rand's Uniform,
     * statrs' ziggurat Exp) instead of the build's Philox stream.  Everything from main.rs:370 on keeps the build's streams. */
    int32_t reference_seed_stream;
} ps_sim_params;

typedef struct {
    uint64_t pan_size;                  /* main.rs:259 */
    double avg_gene_freq_adj;           /* :263-268 */
    int32_t avg_gene_num;               /* :272 */
    double n_core_mutations;            /* :275-276 */
    double n_recombinations_core;       /* :279 */
    double n_recombinations_pan_total;  /* :280 */
    int32_t n_comp;                     /* :341, :355 */
    uint64_t comp_begin[2], comp_end[2];
    double n_pan_mutations[2];          /* :348, :361 */
    double n_recombinations_pan[2];     /* :349-351, :364-366 */
} ps_derived;

void ps_sim_default_params(ps_sim_params *p);                  /* defaults of main.rs:21-151 */
/* main.rs:195-247: 0 if valid; otherwise the reference's stdout text is written to msg */
int ps_sim_validate(const ps_sim_params *p, char *msg, size_t cap);
int ps_sim_derive(const ps_sim_params *p, ps_derived *d);      /* main.rs:259-367 */
/* main.rs:287-319 (build's seeded host stream) */
int ps_selection_coefficients(uint64_t seed, uint64_t n_genes, double prop_positive,
                              double pos_lambda, double neg_lambda, double *out);
/* The same draws from the reference's own seeded stream (ChaCha12 StdRng; see ps_sim_params.reference_seed_stream).
 * Restated from the published algorithms of rand 0.8.5 / rand_chacha 0.3 / statrs 0.16, which are not available here:
 * UNPINNED. */
int ps_reference_selection_coefficients(uint64_t seed, uint64_t n_genes, double prop_positive,
                                        double pos_lambda, double neg_lambda, double *out);
/* the ChaCha block function behind it (rounds = 12 for StdRng; 20 reproduces the RFC 7539 vectors): 16 output words */
void ps_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]);
/* main.rs:413-427 */
int ps_sample_pairs(uint64_t seed, uint64_t pop_size, uint64_t max_distances, uint32_t *range1,
                    uint32_t *range2);

typedef struct ps_sim ps_sim;
/* main.rs:259-427: derive, draw selection coefficients, build both populations and the pair list */
int ps_sim_create(const ps_sim_params *p, ps_sim **out);
void ps_sim_destroy(ps_sim *s);
/* main.rs:429-464 for generations [first, first+count): select, gather x2,
 * mutate x2, HR, HGT.  Asynchronous on the device; ps_sim_sync() waits.
 * Row order.  INSIDE, the loop stores the children of a generation in ascending parent order (a stable sort of the N draws
 * of sample_indices, population.rs:440-443: what lets populations wider than one wavefront gather from a ~1 KB window of the
 * parent row).  AT THE BOUNDARY every output of the simulation comes in the reference's order: row k of
 * ps_read_matrix / ps_write / ps_multi_write on the simulation's handles is the child of draw k (main.rs:445-447), the pair
 * list (main.rs:413-427) and ps_pairwise_counts / _distances on those handles name individuals by that row,
 * ps_fitness_terms / ps_average_distance return their vectors in it, ps_sim_last_parents returns the draws in draw order.
 * A direct ps_load_matrix / ps_next_generation / ps_step on a simulation's handle makes the internal order the output order
 * until the simulation's next generation. */
int ps_sim_run(ps_sim *s, uint32_t first_generation, uint32_t count);
int ps_sim_sync(ps_sim *s);
/* Shard the HGT donors over the site shards of this run (shard_rank / shard_count of the parameters) and exchange the
 * deltas through `fn` once per generation (ps_set_donor_shard).  Every shard of the run must do the same. */
int ps_sim_set_exchange(ps_sim *s, ps_exchange_fn fn, void *ctx);
/* bench.py --emulate-shard K: play shard 0 of K with the exchange stood in for by device-local copies of the same
 * volume plus a kernel that holds the stream for the time the two collectives would take on one xGMI link each
 * (latency + (K - 1) / K x bytes / link rate per collective; timing only: the other shards' events never arrive). */
int ps_sim_emulate_exchange(ps_sim *s, int n_shards);
/* what the emulation charged: modelled link time (microseconds, accumulated since the last reset) and its parameters */
int ps_sim_emulated_link_time(ps_sim *s, int reset, double *modelled_us, double *link_gbps, double *latency_us);
/* exchange calls and bytes this shard sent + received in them since the last reset */
int ps_sim_exchange_stats(ps_sim *s, int reset, uint64_t *calls, uint64_t *bytes);
/* The native provider of ps_exchange_fn for one process per GPU: the OR over RCCL (all-to-all of the K row slices with
 * ncclSend / ncclRecv, a local OR, ncclAllGather of the merged slices -- RCCL has no OR reduction).  librccl.so is
 * opened with dlopen at the first call; without it these calls fail with PS_ERR_NO_DEVICE (ps_rccl_available() = 0)
 * and nothing else of the library is affected.  Rank 0 draws the id, the host carries its 128 bytes to the other ranks
 * (MPI, a file, a socket), every rank creates its handle (collective: ncclCommInitRank) and installs
 *     ps_sim_set_exchange(sim, ps_exchange_rccl, handle).
 * The reference has no counterpart (one process, rayon threads: main.rs:249-257). */
#define PS_RCCL_ID_BYTES 128
typedef struct ps_rccl_exchange ps_rccl_exchange;
int ps_rccl_available(void);
int ps_rccl_unique_id(uint8_t *id_out /* PS_RCCL_ID_BYTES */);
int ps_rccl_exchange_create(const uint8_t *id, int rank, int world, int device, ps_rccl_exchange **out);
void ps_rccl_exchange_destroy(ps_rccl_exchange *x);
/* a ps_exchange_fn: ctx = the ps_rccl_exchange of this rank */
int ps_exchange_rccl(void *ctx, void *d_words, uint64_t n_words, void *hip_stream);
/* calls and bytes this rank sent + received in them since the last reset */
int ps_rccl_exchange_stats(ps_rccl_exchange *x, int reset, uint64_t *calls, uint64_t *bytes);
ps_population *ps_sim_core(ps_sim *s);
ps_population *ps_sim_acc(ps_sim *s);
const double *ps_sim_selection(ps_sim *s);                     /* pan_size values */
const uint32_t *ps_sim_range1(ps_sim *s);
const uint32_t *ps_sim_range2(ps_sim *s);
/* the draws of the most recent generation in draw order (population.rs:443): out_idx[k] = the output row, in the generation
 * before, of the parent of this generation's output row k */
int ps_sim_last_parents(ps_sim *s, uint32_t *out_idx);
/* Device timing of the core sweep kernel, measured with HIP events on the
 * stream it is launched on, accumulated since the last reset: launches, total
 * milliseconds, and algorithmic bytes per launch (2*N*L_local). */
int ps_sim_sweep_timing(ps_sim *s, int reset, uint64_t *launches, double *total_ms,
                        double *bytes_per_launch);
int ps_sim_enable_timing(ps_sim *s, int on);
/* main.rs:467-470: pairwise_distances of both matrices for the run's pair list (P values each), the two
 * kernel chains enqueued together on their own streams.  With site shards the core distances of this call
 * cover this shard's columns only (ps_multi_pairwise_distances sums the shards' numerators first). */
int ps_sim_pairwise_distances(ps_sim *s, double *core_out, double *acc_out);
/* device time of the distance kernels of the last ps_sim_pairwise_distances call, per matrix (HIP events) */
int ps_sim_distance_timing(ps_sim *s, double *core_ms, double *acc_ms);
/* Host half of sample_indices inside ps_sim_run, accumulated since the last reset: generations, milliseconds
 * spent waiting for the device half (gene counts / log-fitness of the previous accessory chain), in the three
 * softmaxes (population.rs:325-393, libm on the host) and in the parent draw (:440-443: the cumulative table on
 * the host; the N draws on the device for pop_size >= 4096, on the host below). */
int ps_sim_host_timing(ps_sim *s, int reset, uint64_t *generations, double *wait_ms, double *weights_ms,
                       double *draw_ms);

/* ------------------------------------------------------------------------ */
/* one process, several devices: the run sharded by core site  */
/* ------------------------------------------------------------------------ */
/* main() of the reference is one process (main.rs:429-553).  ps_multi holds one ps_sim per site shard,
 * shard k on HIP device devices[k] (null: k modulo the visible devices; ordinals may repeat, i.e. several
 * shards on one GPU), each driven by its own host thread inside a call.  A generation needs no exchange
 * between the shards; the distance phase sums the shards' integer Hamming numerators on shard 0's device
 * (device-to-device copies).  All results equal those of the unsharded run bit for bit. */
typedef struct ps_multi ps_multi;
int ps_multi_create(const ps_sim_params *p, int n_shards, const int *devices, ps_multi **out);
void ps_multi_destroy(ps_multi *m);
int ps_multi_shards(ps_multi *m);
/* borrowed, for reading: selection coefficients, pair list, matrices, timings.  A shard's generations and its HGT take
 * part in exchanges between ALL shards (parent weights, HGT deltas): ps_sim_run / ps_recombine on a borrowed shard fail
 * with PS_ERR_STATE -- drive the shards through ps_multi_run. */
ps_sim *ps_multi_shard(ps_multi *m, int k);
/* main.rs:429-464 for generations [first, first+count) on every shard */
int ps_multi_run(ps_multi *m, uint32_t first_generation, uint32_t count);
int ps_multi_sync(ps_multi *m);
/* population.rs:787-837 for the run's pair list: core numerators summed over the shards (out_core: P values) */
int ps_multi_pairwise_counts(ps_multi *m, uint32_t *out_core);
/* main.rs:467-470: core and accessory distances of the run's pair list (P values each) */
int ps_multi_pairwise_distances(ps_multi *m, double *core_out, double *acc_out);
/* main.rs:550-553: <outpref>_core_genome.csv (lines assembled from the shards' columns) and _pangenome.csv */
int ps_multi_write(ps_multi *m, const char *outpref);

#ifdef __cplusplus
}
#endif
#endif
