/*
 * pansim_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see pansim_oracle.h).
 *
 * Every function cites the reference lines (under /root/reference/) it restates.
 * Nothing here is used by the product path.
 */
#define _GNU_SOURCE
#include "pansim_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ========================================================================= */
/* Philox4x32-10 (Salmon et al., SC'11; Random123).  Pinned by the Random123 */
/* known-answer vectors in tests/golden/kat.json.                            */
/* ========================================================================= */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void hs_block(uint64_t seed, uint32_t stream, uint32_t gen, uint64_t blk, uint32_t out[4])
{
    uint32_t ctr[4] = { (uint32_t)blk, (uint32_t)(blk >> 32), gen, stream };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    orc_philox4x32_10(ctr, key, out);
}

/* two f64 per block: (w1:w0) and (w3:w2), top 53 bits * 2^-53 (the usual
 * [0,1) construction; rand 0.8.5 `gen::<f64>` uses the same 53-bit form) */
double orc_hs_f64(uint64_t seed, uint32_t stream, uint32_t gen, uint64_t n)
{
    uint32_t w[4];
    hs_block(seed, stream, gen, n >> 1, w);
    uint64_t x = (n & 1) ? (((uint64_t)w[3] << 32) | w[2]) : (((uint64_t)w[1] << 32) | w[0]);
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

uint32_t orc_hs_u32(uint64_t seed, uint32_t stream, uint32_t gen, uint64_t n)
{
    uint32_t w[4];
    hs_block(seed, stream, gen, n >> 2, w);
    return w[n & 3];
}

static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

/* statrs 0.16 Poisson (population.rs:484, :562) is Knuth for small lambda and
 * a rejection method for large; un-vendored, so only the distribution is
 * contractual.  This is Knuth (<10) / Hoermann PTRS 1993 (>=10). */
uint64_t orc_poisson(double mean, uint64_t seed, uint32_t stream, uint32_t gen, uint64_t *n_used)
{
    uint64_t n = 0, result = 0;
    if (!(mean > 0.0)) { if (n_used) *n_used = 0; return 0; }
    if (mean < 10.0) {
        double lim = exp(-mean), p = 1.0;
        uint64_t k = 0;
        do { k++; p *= orc_hs_f64(seed, stream, gen, n++); } while (p > lim);
        result = k - 1;
    } else {
        double slam = sqrt(mean), loglam = log(mean);
        double b = 0.931 + 2.53 * slam;
        double a = -0.059 + 0.02483 * b;
        double invalpha = 1.1239 + 1.1328 / (b - 3.4);
        double vr = 0.9277 - 3.6224 / (b - 2.0);
        for (;;) {
            double U = orc_hs_f64(seed, stream, gen, n++) - 0.5;
            double V = orc_hs_f64(seed, stream, gen, n++);
            double us = 0.5 - fabs(U);
            double kf = floor((2.0 * a / us + b) * U + mean + 0.43);
            if (us >= 0.07 && V <= vr) { result = (uint64_t)kf; break; }
            if (kf < 0.0 || (us < 0.013 && V > us)) continue;
            if (log(V) + log(invalpha) - log(a / (us * us) + b)
                <= -mean + kf * loglam - lgamma(kf + 1.0)) { result = (uint64_t)kf; break; }
        }
    }
    if (n_used) *n_used = n;
    return result;
}

/* ========================================================================= */
/* parameter derivation: main.rs:259-367                                      */
/* ========================================================================= */
void orc_derive(const orc_params *p, orc_derived *d)
{
    memset(d, 0, sizeof(*d));
    uint64_t pan_size = p->pan_genes - p->core_genes;              /* main.rs:259 */
    d->pan_size = pan_size;
    double core_prop = (double)p->core_genes / (double)p->pan_genes; /* :263 */
    double acc_prop = 1.0 - core_prop;                               /* :264 */
    double agf = (p->avg_gene_freq - core_prop) / acc_prop;          /* :265 */
    if (agf < 0.0) agf = 0.0;                                        /* :266-268 */
    d->avg_gene_freq_adj = agf;
    d->avg_gene_num = (int32_t)round(agf * (double)pan_size);        /* :272 */
    d->n_core_mutations = ceil((double)p->core_size * p->core_mu);   /* :275-276 */
    d->n_recombinations_core = round(d->n_core_mutations * p->HR_rate);   /* :279 */
    d->n_recombinations_pan_total = round(d->n_core_mutations * p->HGT_rate); /* :280 */
    uint64_t g1 = (uint64_t)round((double)pan_size * (1.0 - p->prop_genes2)); /* :334 */
    uint64_t g2 = pan_size - g1;                                     /* :335 */
    double prop1 = (double)g1 / (double)pan_size;                    /* :336 */
    double prop2 = 1.0 - prop1;                                      /* :337 */
    int c = 0;
    if (g1 > 0) {                                                    /* :341-352 */
        d->comp_begin[c] = 0; d->comp_end[c] = g1;
        d->n_pan_mutations[c] = p->rate_genes1 * (double)g1;
        d->n_recombinations_pan[c] = d->n_recombinations_pan_total * prop1;
        c++;
    }
    if (g1 < pan_size) {                                             /* :355-367 */
        d->comp_begin[c] = g1; d->comp_end[c] = pan_size;
        d->n_pan_mutations[c] = p->rate_genes2 * (double)g2;
        d->n_recombinations_pan[c] = d->n_recombinations_pan_total * prop2;
        c++;
    }
    d->n_comp = c;
}

/* ========================================================================= */
/* keyed dense plans                                    */
/* ========================================================================= */
static uint32_t prob_to_u32(double p)
{
    double x = floor(p * 4294967296.0);
    if (!(x > 0.0)) return 0u;
    if (x >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

/* Dense form of population.rs:511-540 and :544-751 (core path).
 * Poisson splitting: a cell (individual, site) receives Poisson(lam_mut/L)
 * mutation events and Poisson(lam_hr/L) incoming HR events, independently; the last mutation wins and is uniform over
 * {2,4,8} (:531), the winning donor is uniform over the others.  Per cell: mutate to 2 / 4 / 8 only with mass a each,
 * mutate AND receive a donor allele with mass b each, receive a donor allele only with mass c.
 * Two levels: a 6-bit symbol decides k / 64 of each a outright; what is left of the event mass lives,
 * scaled by 64 / R, in R residual symbols, inside which a 32-bit word is cut by the cumulative thresholds T. */
void orc_core_plan_make(double lam_mut, double lam_hr, uint64_t L, orc_core_plan *plan)
{
    double p = (lam_mut > 0.0) ? -expm1(-lam_mut / (double)L) : 0.0;
    double q = (lam_hr > 0.0) ? -expm1(-lam_hr / (double)L) : 0.0;
    double a = p * (1.0 - q) / 3.0;
    double b = p * q / 3.0;
    double c = (1.0 - p) * q;
    uint32_t k = (uint32_t)floor(a * 64.0);
    double a_left = a - (double)k / 64.0;
    double m_res = a_left + a_left + a_left + b + b + b + c;
    uint32_t R = (uint32_t)ceil(m_res * 64.0);
    if (3u * k + R > 64u) R = 64u - 3u * k;
    double scale = R ? 64.0 / (double)R : 0.0;
    double cum[7];
    cum[0] = a_left * scale;
    cum[1] = (a_left + a_left) * scale;
    cum[2] = (a_left + a_left + a_left) * scale;
    cum[3] = (a_left + a_left + a_left + b) * scale;
    cum[4] = (a_left + a_left + a_left + b + b) * scale;
    cum[5] = (a_left + a_left + a_left + b + b + b) * scale;
    cum[6] = m_res * scale;
    uint32_t prev = 0;
    for (int j = 0; j < 7; j++) {
        uint32_t t = prob_to_u32(cum[j]);
        if (t < prev) t = prev;
        plan->T[j] = t;
        prev = t;
    }
    if (plan->T[6] == 0u) R = 0u;
    plan->k = k;
    plan->R = R;
    plan->has_events = (k > 0u || R > 0u) ? 1u : 0u;
    uint32_t cs = 0;
    while (cs < 4u && 3u * k + R > (4u << cs)) cs++;
    plan->cshift = cs;
}

/* flip iff the Poisson(lam/n) toggle count of a cell is odd (population.rs:501-508) */
uint32_t orc_acc_flip_threshold(double lam, uint64_t n_genes_in_comp)
{
    if (!(lam > 0.0) || n_genes_in_comp == 0) return 0u;
    double pf = -expm1(-2.0 * lam / (double)n_genes_in_comp) / 2.0;
    return prob_to_u32(pf);
}

typedef struct { uint8_t mut; uint8_t hr; uint32_t donor; } cell_outcome;

/* The 6-bit symbol of cell (site, ind): bit-sliced over two Philox blocks.
 *   A = Philox(site / 2, ind / 16, gen, CORE_L1):  word j = plane j (j = 0..3) of the sites 2 (site / 2) + {0, 1}
 *   B = Philox(site / 4, ind / 16, gen, CORE_L1B): words 0 / 1 = plane 4 of the sites 4 (site / 4) + {0, 1} / {2, 3},
 *                                                  words 2 / 3 = plane 5 of the same
 * and inside a plane word the cell is bit 8 (ind % 4) + (ind / 4) % 4 + 4 (site % 2).  Symbol = 4 n + t, bit j of n = plane j, bit 0 / 1
 * of t = plane 4 / 5. */
static void core_blocks(uint64_t seed, uint32_t gen, uint32_t site, uint32_t chunk, uint32_t A[4], uint32_t B[4])
{
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t ca[4] = { site >> 1, chunk, gen, ORC_STREAM_CORE_L1 };
    uint32_t cb[4] = { site >> 2, chunk, gen, ORC_STREAM_CORE_L1B };
    orc_philox4x32_10(ca, key, A);
    orc_philox4x32_10(cb, key, B);
}

/* (A, B = core_blocks(site, ind / 16): the row loops below compute them once per 16 cells) */
static uint32_t core_symbol(const uint32_t A[4], const uint32_t B[4], uint32_t site, uint32_t ind)
{
    uint32_t pos = 8u * (ind & 3u) + ((ind >> 2) & 3u) + 4u * (site & 1u);
    uint32_t n = ((A[0] >> pos) & 1u) | (((A[1] >> pos) & 1u) << 1) | (((A[2] >> pos) & 1u) << 2) | (((A[3] >> pos) & 1u) << 3);
    uint32_t p4 = (site & 2u) ? B[1] : B[0], p5 = (site & 2u) ? B[3] : B[2];
    uint32_t t = ((p4 >> pos) & 1u) | (((p5 >> pos) & 1u) << 1);
    return 4u * n + t;
}

static cell_outcome core_cell(uint64_t seed, uint32_t gen, uint32_t site, uint32_t ind, uint64_t N,
                              const orc_core_plan *plan, const uint32_t A[4], const uint32_t B[4])
{
    cell_outcome o = { 0, 0, 0 };
    if (!plan->has_events) return o;
    uint32_t s = core_symbol(A, B, site, ind);
    if ((s >> 2) >> plan->cshift) return o;          /* (implied by the ranges below: 3k + R <= 4 << cshift) */
    if (s < plan->k) { o.mut = 2; return o; }
    if (s < 2u * plan->k) { o.mut = 4; return o; }
    if (s < 3u * plan->k) { o.mut = 8; return o; }
    if (s >= 3u * plan->k + plan->R) return o;
    /* residual symbol: one 32-bit word against the thresholds */
    uint32_t ctr[4] = { site, ind, gen, ORC_STREAM_CORE_L2 };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t w[4];
    orc_philox4x32_10(ctr, key, w);
    uint32_t u = w[0];
    const uint32_t *T = plan->T;
    if (u >= T[6]) return o;
    if (u < T[0]) o.mut = 2;
    else if (u < T[1]) o.mut = 4;
    else if (u < T[2]) o.mut = 8;
    else if (u < T[3]) { o.mut = 2; o.hr = 1; }
    else if (u < T[4]) { o.mut = 4; o.hr = 1; }
    else if (u < T[5]) { o.mut = 8; o.hr = 1; }
    else o.hr = 1;
    if (o.hr) {
        if (N < 2) { o.hr = 0; return o; }
        uint32_t d = mulhi32(w[1], (uint32_t)(N - 1));
        if (d >= ind) d++;                       /* population.rs:618 shift past self */
        o.donor = d;
    }
    return o;
}

/* population.rs:511-540: the allele written is uniform over core_vec[1 >> value],
 * and `1 >> value` is 0 for every value >= 1, i.e. always {2,4,8}. */
void orc_mutate_core(uint8_t *pop, uint64_t N, uint64_t L, uint64_t site_offset, uint64_t seed,
                     uint32_t gen, const orc_core_plan *plan)
{
    if (!plan->has_events) return;
    for (uint64_t s = 0; s < L; s++) {
        uint32_t site = (uint32_t)(site_offset + s);
        for (uint64_t c = 0; c * 16 < N; c++) {
            uint32_t A[4], B[4];
            core_blocks(seed, gen, site, (uint32_t)c, A, B);
            for (uint64_t i = c * 16; i < N && i < c * 16 + 16; i++) {
                cell_outcome o = core_cell(seed, gen, site, (uint32_t)i, N, plan, A, B);
                if (o.mut) pop[i * L + s] = o.mut;
            }
        }
    }
}

/* population.rs:544-751 core path: donor alleles are read from the
 * pre-recombination snapshot (:693-695); a cell that receives any event ends
 * up with the allele of a uniformly chosen other individual at the same site. */
void orc_recombine_core(uint8_t *pop, uint64_t N, uint64_t L, uint64_t site_offset, uint64_t seed,
                        uint32_t gen, const orc_core_plan *plan)
{
    if (!plan->has_events || N < 2) return;
    uint8_t *col = (uint8_t *)malloc(N);
    for (uint64_t s = 0; s < L; s++) {
        uint32_t site = (uint32_t)(site_offset + s);
        for (uint64_t i = 0; i < N; i++) col[i] = pop[i * L + s];
        for (uint64_t c = 0; c * 16 < N; c++) {
            uint32_t A[4], B[4];
            core_blocks(seed, gen, site, (uint32_t)c, A, B);
            for (uint64_t i = c * 16; i < N && i < c * 16 + 16; i++) {
                cell_outcome o = core_cell(seed, gen, site, (uint32_t)i, N, plan, A, B);
                if (o.hr) pop[i * L + s] = col[o.donor];
            }
        }
    }
    free(col);
}

/* population.rs:486-510 in dense form: independent flip per (individual, gene) */
void orc_mutate_acc(uint8_t *pop, uint64_t N, uint64_t G, uint64_t seed, uint32_t gen, int n_comp,
                    const uint64_t *comp_begin, const uint64_t *comp_end, const double *lambdas)
{
    uint32_t *thr = (uint32_t *)calloc(G ? G : 1, sizeof(uint32_t));
    for (int c = 0; c < n_comp; c++) {
        if (lambdas[c] == 0.0) continue;                 /* population.rs:480 */
        uint32_t t = orc_acc_flip_threshold(lambdas[c], comp_end[c] - comp_begin[c]);
        for (uint64_t g = comp_begin[c]; g < comp_end[c]; g++) thr[g] = t;
    }
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    for (uint64_t i = 0; i < N; i++) {
        for (uint64_t g4 = 0; g4 * 4 < G; g4++) {
            uint32_t ctr[4] = { (uint32_t)g4, (uint32_t)i, gen, ORC_STREAM_ACC_MUT };
            uint32_t w[4];
            orc_philox4x32_10(ctr, key, w);
            for (uint64_t g = g4 * 4; g < G && g < g4 * 4 + 4; g++) {
                if (w[g & 3] < thr[g]) {
                    uint8_t v = pop[i * G + g];
                    pop[i * G + g] = (v == 0) ? 1 : 0;   /* population.rs:505 */
                }
            }
        }
    }
    free(thr);
}

/* Poisson(lambda) by inversion over an integer threshold table: thr[j] =
 * floor(P(K <= kmin + j) * 2^32) for kmin + j in [lambda - 12 sigma - 12, lambda + 12 sigma + 12]
 * (the mass outside is below 1e-30); a 32-bit uniform u gives k = kmin + #{j : thr[j] <= u},
 * capped at the table's last value.  Integer comparison only at draw time, so the host table and
 * a device kernel give the same counts.  statrs' Poisson (population.rs:562, :599) is
 * un-vendored; only the distribution is contractual.  Returns the table length (0 if cap is too
 * small or lambda <= 0). */
uint32_t orc_poisson_table(double lambda, uint32_t *kmin_out, uint32_t *thr, uint32_t cap)
{
    if (!(lambda > 0.0)) return 0;
    const double spread = 12.0 * sqrt(lambda) + 12.0;
    const double lo = floor(lambda - spread);
    const uint32_t kmin = lo > 0.0 ? (uint32_t)lo : 0u;
    const uint32_t kmax = (uint32_t)ceil(lambda + spread);
    const uint32_t len = kmax - kmin + 1u;
    if (len > cap) return 0;
    const double loglam = log(lambda);
    double cdf = 0.0;
    for (uint32_t j = 0; j < len; j++) {
        const double k = (double)(kmin + j);
        cdf += exp(k * loglam - lambda - lgamma(k + 1.0));
        const double scaled = floor(cdf * 4294967296.0);
        thr[j] = scaled >= 4294967295.0 ? 4294967295u : (uint32_t)scaled;
    }
    *kmin_out = kmin;
    return len;
}

uint32_t orc_poisson_from_table(uint32_t u, uint32_t kmin, const uint32_t *thr, uint32_t len)
{
    uint32_t lo = 0, hi = len;              /* number of thresholds <= u (they ascend) */
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (thr[mid] <= u) lo = mid + 1u; else hi = mid;
    }
    if (lo >= len) lo = len - 1u;
    return kmin + lo;
}

/* population.rs:544-751, accessory path (HGT), keyed per donor like the reference draws it:
 * donor d of compartment c sends k_d ~ Poisson(lambda_c) events (:599; here the threshold table
 * above on word 0 of Philox(d, 0, gen, HGT_COUNT | c << 8)).  Events 2m and 2m + 1 of donor d share the
 * block Philox(m, d, gen, HGT | c << 8): words 0, 1 serve event 2m, words 2, 3 event 2m + 1 ->
 * recipient uniform over the others (:584, :616-619; the first word of the pair),
 * locus uniform among the donor's present genes of the compartment in the pre-recombination
 * snapshot (:636-680; the second word), value always 1 (:632).  A donor without genes in the compartment
 * sends nothing (:672, :740).  Returns the number of events drawn (donors without genes included). */
uint64_t orc_recombine_acc(uint8_t *pop, uint64_t N, uint64_t G, uint64_t seed, uint32_t gen,
                           int n_comp, const uint64_t *comp_begin, const uint64_t *comp_end,
                           const double *lambdas)
{
    if (N < 2 || G == 0) return 0;
    uint8_t *snap = (uint8_t *)malloc(N * G);
    memcpy(snap, pop, N * G);
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint64_t total = 0;
    uint32_t *list = (uint32_t *)malloc(G * sizeof(uint32_t));   /* present genes of the donor */
    for (int c = 0; c < n_comp; c++) {
        if (lambdas[c] == 0.0) continue;                 /* population.rs:558 */
        const uint32_t cap = (uint32_t)(24.0 * sqrt(lambdas[c]) + 64.0);
        uint32_t *thr = (uint32_t *)malloc((size_t)cap * sizeof(uint32_t));
        uint32_t kmin = 0;
        const uint32_t len = orc_poisson_table(lambdas[c], &kmin, thr, cap);
        const uint32_t stream_k = ORC_STREAM_HGT_COUNT | ((uint32_t)c << 8);
        const uint32_t stream_e = ORC_STREAM_HGT | ((uint32_t)c << 8);
        for (uint64_t d = 0; len && d < N; d++) {
            uint32_t ctr[4] = { (uint32_t)d, 0u, gen, stream_k };
            uint32_t w[4];
            orc_philox4x32_10(ctr, key, w);
            const uint32_t k = orc_poisson_from_table(w[0], kmin, thr, len);
            total += k;
            /* the donor's WeightedIndex over its present genes of the compartment (:636-680) */
            uint32_t n = 0;
            for (uint64_t g = comp_begin[c]; g < comp_end[c]; g++)
                if (snap[d * G + g] != 0) list[n++] = (uint32_t)g;
            if (n == 0) continue;                        /* population.rs:672 */
            for (uint32_t j = 0; j < k; j++) {
                /* one block per two events: words 0, 1 = (recipient, gene) of event 2m, words 2, 3 of event 2m + 1 */
                if ((j & 1u) == 0u) {
                    uint32_t ce[4] = { j >> 1, (uint32_t)d, gen, stream_e };
                    orc_philox4x32_10(ce, key, w);
                }
                const uint32_t h = (j & 1u) * 2u;
                uint32_t r = mulhi32(w[h], (uint32_t)(N - 1));
                if (r >= d) r++;                         /* population.rs:618 */
                pop[(uint64_t)r * G + list[mulhi32(w[h + 1u], n)]] = 1;
            }
        }
        free(thr);
    }
    free(list);
    free(snap);
    return total;
}

/* ========================================================================= */
/* initial state / host-stream draws                                          */
/* ========================================================================= */
/* population.rs:201-204: gen_range(0..4) then 1 << i, one vector for everyone */
void orc_init_core_vec(uint64_t seed, uint64_t L, uint8_t *allele_vec)
{
    for (uint64_t s = 0; s < L; s++)
        allele_vec[s] = (uint8_t)(1u << (orc_hs_u32(seed, ORC_STREAM_INIT_CORE, 0, s) >> 30));
}

/* population.rs:215-219: gen::<f64>() < avg_gene_freq */
void orc_init_acc_vec(uint64_t seed, uint64_t G, double avg_gene_freq_adj, uint8_t *acc_vec)
{
    for (uint64_t g = 0; g < G; g++)
        acc_vec[g] = orc_hs_f64(seed, ORC_STREAM_INIT_ACC, 0, g) < avg_gene_freq_adj ? 1 : 0;
}

void orc_replicate(const uint8_t *vec, uint64_t N, uint64_t ncols, uint8_t *pop)
{
    for (uint64_t i = 0; i < N; i++) memcpy(pop + i * ncols, vec, ncols); /* population.rs:206-229 */
}

/* main.rs:287-319.  Exp(lambda) by inversion (statrs Exp is un-vendored). */
uint64_t orc_selection_coefficients(uint64_t seed, uint64_t G, double prop_positive,
                                    double pos_lambda, double neg_lambda, double *out)
{
    uint64_t n = 0;
    for (uint64_t g = 0; g < G; g++) out[g] = 0.0;           /* main.rs:287 */
    if (!(prop_positive >= 0.0)) return 0;                    /* main.rs:292 */
    for (uint64_t g = 0; g < G; g++) {
        double weight = orc_hs_f64(seed, ORC_STREAM_SELECTION, 0, n++);
        double s;
        if (weight <= prop_positive) {                        /* main.rs:303 */
            s = -log(1.0 - orc_hs_f64(seed, ORC_STREAM_SELECTION, 0, n++)) / pos_lambda;
        } else {
            s = -log(1.0 - orc_hs_f64(seed, ORC_STREAM_SELECTION, 0, n++)) / neg_lambda;
            while (s > 1.0)                                   /* main.rs:309-311 */
                s = -log(1.0 - orc_hs_f64(seed, ORC_STREAM_SELECTION, 0, n++)) / neg_lambda;
            s = -1.0 * s;                                     /* main.rs:315 */
        }
        out[g] = s;
    }
    return n;
}

/* main.rs:413-427 */
void orc_sample_pairs(uint64_t seed, uint64_t N, uint64_t P, uint32_t *range1, uint32_t *range2)
{
    for (uint64_t k = 0; k < P; k++)
        range1[k] = mulhi32(orc_hs_u32(seed, ORC_STREAM_PAIRS, 0, k), (uint32_t)N);
    for (uint64_t k = 0; k < P; k++) {
        uint32_t e = mulhi32(orc_hs_u32(seed, ORC_STREAM_PAIRS, 1, k), (uint32_t)(N - 1));
        if (e >= range1[k]) e += 1;
        range2[k] = e;
    }
}

/* ========================================================================= */
/* deterministic reference functions                                          */
/* ========================================================================= */
static inline uint32_t popc64(uint64_t v) { return (uint32_t)__builtin_popcountll(v); }

/* distances.rs:22-52 */
uint32_t orc_hamming_bitwise_fast(const uint8_t *x, const uint8_t *y, size_t n)
{
    uint32_t distance = 0;
    size_t chunks = n / 8;
    for (size_t c = 0; c < chunks; c++) {
        uint64_t xv, yv;
        memcpy(&xv, x + 8 * c, 8);
        memcpy(&yv, y + 8 * c, 8);
        distance += popc64(xv ^ yv);
    }
    if (n % 8 != 0) {
        for (size_t k = chunks * 8; k < n; k++)
            distance += (uint32_t)__builtin_popcount((unsigned)(x[k] ^ y[k]));
    }
    return distance;
}

/* distances.rs:55-77 */
void orc_jaccard_distance_fast(const uint8_t *x, const uint8_t *y, size_t n,
                               uint32_t *inter, uint32_t *uni)
{
    uint32_t intersection = 0, un = 0;
    size_t chunks = n / 8;
    for (size_t c = 0; c < chunks; c++) {
        uint64_t xv, yv;
        memcpy(&xv, x + 8 * c, 8);
        memcpy(&yv, y + 8 * c, 8);
        intersection += popc64(xv & yv);
        un += popc64(xv | yv);
    }
    for (size_t k = chunks * 8; k < n; k++) {
        intersection += (uint32_t)__builtin_popcount((unsigned)(x[k] & y[k]));
        un += (uint32_t)__builtin_popcount((unsigned)(x[k] | y[k]));
    }
    *inter = intersection;
    *uni = un;
}

/* population.rs:787-837 */
void orc_pairwise_distances(const uint8_t *pop, uint64_t N, uint64_t ncols, int core,
                            uint64_t core_genes, uint64_t P, const uint32_t *r1,
                            const uint32_t *r2, double *out)
{
    (void)N;
    for (uint64_t k = 0; k < P; k++) {
        const uint8_t *row1 = pop + (uint64_t)r1[k] * ncols;
        const uint8_t *row2 = pop + (uint64_t)r2[k] * ncols;
        double fd;
        if (core) {
            uint32_t distance = orc_hamming_bitwise_fast(row1, row2, ncols) / 2;   /* :817 */
            fd = (double)distance / (double)ncols;                                /* :822 */
        } else {
            uint32_t inter, uni;
            orc_jaccard_distance_fast(row1, row2, ncols, &inter, &uni);
            fd = 1.0 - (((double)inter + (double)core_genes)
                        / ((double)uni + (double)core_genes));                    /* :828-830 */
        }
        out[k] = fd;
    }
}

void orc_pairwise_hamming_counts(const uint8_t *pop, uint64_t N, uint64_t ncols,
                                 uint64_t col_begin, uint64_t col_end, uint64_t P,
                                 const uint32_t *r1, const uint32_t *r2, uint32_t *out)
{
    (void)N;
    for (uint64_t k = 0; k < P; k++) {
        const uint8_t *row1 = pop + (uint64_t)r1[k] * ncols + col_begin;
        const uint8_t *row2 = pop + (uint64_t)r2[k] * ncols + col_begin;
        uint32_t d = 0;
        for (uint64_t s = 0; s < col_end - col_begin; s++)
            d += (uint32_t)__builtin_popcount((unsigned)(row1[s] ^ row2[s]));
        out[k] = d;
    }
}

/* population.rs:753-784 with get_distance :114-151 */
/* one row of average_distance (the body of the `.map(|i| ...)` closure, population.rs:759-779) */
static double average_distance_row(const uint8_t *pop, uint64_t N, uint64_t ncols, int core, uint64_t core_genes, uint64_t i);

void orc_average_distance(const uint8_t *pop, uint64_t N, uint64_t ncols, int core,
                          uint64_t core_genes, double *out)
{
    for (uint64_t i = 0; i < N; i++) out[i] = average_distance_row(pop, N, ncols, core, core_genes, i);
}

static double average_distance_row(const uint8_t *pop, uint64_t N, uint64_t ncols, int core, uint64_t core_genes, uint64_t i)
{
    double matches = 0.0;                                      /* population.rs:764 */
    {
        const uint8_t *row1 = pop + i * ncols;
        double sum = 0.0;
        uint64_t count = 0;
        for (uint64_t j = 0; j < N; j++) {
            if (j == i) continue;                              /* :128-130 */
            const uint8_t *row2 = pop + j * ncols;
            double pd;
            if (core) {
                uint32_t distance = orc_hamming_bitwise_fast(row1, row2, ncols) / 2; /* :136 */
                pd = (double)distance / (double)ncols;
            } else {
                uint32_t inter, uni;
                orc_jaccard_distance_fast(row1, row2, ncols, &inter, &uni);
                pd = 1.0 - (((double)inter + matches + (double)core_genes)
                            / ((double)uni + matches + (double)core_genes));      /* :144-145 */
            }
            sum = sum + pd;                                    /* :770 */
            count++;
        }
        double fd = sum / (double)count;                       /* :771 */
        if (fd == 0.0) fd = DBL_MIN;                           /* :774-776 f64::MIN_POSITIVE */
        return fd;
    }
}

/* population.rs:840-863 */
void orc_gene_frequencies(const uint8_t *pop, uint64_t N, uint64_t G, uint64_t core_genes,
                          double *out)
{
    double n_individuals = (double)N;
    for (uint64_t g = 0; g < G; g++) {
        uint64_t sum = 0;
        for (uint64_t i = 0; i < N; i++) sum += pop[i * G + g];
        out[g] = (double)sum / n_individuals;
    }
    for (uint64_t k = 0; k < core_genes; k++) out[G + k] = 1.0;
}

/* population.rs:244-268 */
double orc_calc_gene_freq(const uint8_t *pop, uint64_t N, uint64_t ncols)
{
    double sum = 0.0;
    for (uint64_t i = 0; i < N; i++) {
        uint64_t rs = 0;
        for (uint64_t g = 0; g < ncols; g++) rs += pop[i * ncols + g];
        sum += (double)rs / (double)ncols;
    }
    return sum / (double)N;
}

/* population.rs:450-465 */
void orc_next_generation(const uint8_t *pop, uint64_t N, uint64_t ncols, const uint32_t *sample,
                         uint8_t *next)
{
    for (uint64_t i = 0; i < N; i++)
        memcpy(next + i * ncols, pop + (uint64_t)sample[i] * ncols, ncols);
}

/* population.rs:83-94 (population sigma; returns (std, mean)) */
void orc_standard_deviation(const double *v, uint64_t n, double *std, double *mean)
{
    double s = 0.0;
    for (uint64_t i = 0; i < n; i++) s += v[i];
    double m = s / (double)n;
    double ss = 0.0;
    for (uint64_t i = 0; i < n; i++) { double d = v[i] - m; ss += d * d; }
    *std = sqrt(ss / (double)n);
    *mean = m;
}

/* population.rs:154-162 */
char orc_int_to_base(uint8_t n)
{
    switch (n) {
    case 1: return 'A';
    case 2: return 'C';
    case 4: return 'G';
    case 8: return 'T';
    default: return 'N';
    }
}

/* ========================================================================= */
/* parent sampling: population.rs:270-448                                     */
/* ========================================================================= */
/* population.rs:282-322 */
void orc_fitness_terms(const uint8_t *pop, uint64_t N, uint64_t G, const double *sel_coeff,
                       int32_t *num_genes, double *logw)
{
    for (uint64_t i = 0; i < N; i++) {
        const uint8_t *row = pop + i * G;
        int32_t sum = 0;
        for (uint64_t g = 0; g < G; g++) sum += (int32_t)row[g];     /* :286 */
        num_genes[i] = sum;
        int neg_inf = 0;
        double log_sum = 0.0;
        for (uint64_t g = 0; g < G; g++) {
            double lv = log(1.0 + sel_coeff[g] * (double)row[g]);    /* :306 */
            if (lv == -INFINITY) neg_inf = 1;                        /* :312 */
            log_sum += lv;                                           /* :317 */
        }
        logw[i] = neg_inf ? 0.0 : log_sum;                           /* :315-320 */
    }
}

/* logsumexp 0.1 `ln_sum_exp` (un-vendored): restated as the one-pass streaming
 * form (running maximum + rescaled sum). */
static double ln_sum_exp(const double *x, uint64_t n)
{
    double alpha = -INFINITY, r = 0.0;
    for (uint64_t i = 0; i < n; i++) {
        if (x[i] <= alpha) r += exp(x[i] - alpha);
        else { r *= exp(alpha - x[i]); r += 1.0; alpha = x[i]; }
    }
    return log(r) + alpha;
}

static void softmax_norm(double *v, uint64_t n)
{
    double lse = ln_sum_exp(v, n);
    for (uint64_t i = 0; i < n; i++) v[i] = exp(v[i] - lse);          /* :333, :358, :379 */
    double sum = 0.0;
    for (uint64_t i = 0; i < n; i++) sum += v[i];                     /* :338, :360, :381 */
    for (uint64_t i = 0; i < n; i++) v[i] = (v[i] != -INFINITY) ? v[i] / sum : 0.0; /* :340 */
}

int orc_sample_weights(const int32_t *num_genes, const double *logw, uint64_t N, uint64_t G,
                       int32_t avg_gene_num, const double *avg_pairwise_dists,
                       int no_control_genome_size, double genome_size_penalty,
                       double competition_strength, double *weights)
{
    double *sel = (double *)malloc(N * sizeof(double));
    double *tmp = (double *)malloc(N * sizeof(double));
    for (uint64_t i = 0; i < N; i++) sel[i] = 1.0;                   /* :293 */
    if (G > 0) {                                                     /* :296 */
        for (uint64_t i = 0; i < N; i++) sel[i] = logw[i];
        softmax_norm(sel, N);                                        /* :325-340 */
    }
    if (!no_control_genome_size) {                                   /* :346 */
        for (uint64_t i = 0; i < N; i++) {
            double diff = (double)(num_genes[i] - avg_gene_num);     /* :350 */
            tmp[i] = diff * log(genome_size_penalty);                /* :355 */
        }
        softmax_norm(tmp, N);                                        /* :356-361 */
        for (uint64_t i = 0; i < N; i++) weights[i] = tmp[i] * sel[i]; /* :368 */
    } else {
        for (uint64_t i = 0; i < N; i++) weights[i] = sel[i];        /* :371 */
    }
    for (uint64_t i = 0; i < N; i++)
        tmp[i] = competition_strength * log(avg_pairwise_dists[i]);  /* :375 */
    softmax_norm(tmp, N);                                            /* :377-382 */
    for (uint64_t i = 0; i < N; i++) weights[i] = weights[i] * tmp[i]; /* :389-393 */
    double mx = -INFINITY;
    for (uint64_t i = 0; i < N; i++) mx = fmax(mx, weights[i]);      /* :403 */
    if (mx == 0.0)
        for (uint64_t i = 0; i < N; i++) weights[i] = 1.0;           /* :435-437 */
    free(sel);
    free(tmp);
    /* WeightedIndex::new(&weights).unwrap() :440 -- panics on NaN/negative/zero total */
    double total = 0.0;
    for (uint64_t i = 0; i < N; i++) {
        if (!(weights[i] >= 0.0)) return -1;
        total += weights[i];
    }
    if (!(total > 0.0) || isinf(total)) return -2;
    return 0;
}

/* population.rs:440-443.  rand 0.8.5 WeightedIndex: cumulative sums of the
 * first N-1 weights, x uniform in [0,total), index = #cumulative <= x. */
int orc_draw_parents(const double *weights, uint64_t N, uint64_t seed, uint32_t gen, uint32_t *idx)
{
    double *cum = (double *)malloc(N * sizeof(double));
    double total = weights[0];
    for (uint64_t i = 1; i < N; i++) { cum[i - 1] = total; total += weights[i]; }
    for (uint64_t k = 0; k < N; k++) {
        double x = orc_hs_f64(seed, ORC_STREAM_PARENTS, gen, k) * total;
        uint64_t lo = 0, hi = N - 1;            /* partition_point(cum[j] <= x) over N-1 entries */
        while (lo < hi) {
            uint64_t mid = lo + (hi - lo) / 2;
            if (cum[mid] <= x) lo = mid + 1; else hi = mid;
        }
        idx[k] = (uint32_t)lo;
    }
    free(cum);
    return 0;
}

int orc_sample_indices(const uint8_t *pop, uint64_t N, uint64_t G, uint64_t seed, uint32_t gen,
                       int32_t avg_gene_num, const double *avg_pairwise_dists,
                       const double *sel_coeff, int no_control_genome_size,
                       double genome_size_penalty, double competition_strength, uint32_t *idx)
{
    int32_t *num_genes = (int32_t *)malloc(N * sizeof(int32_t));
    double *logw = (double *)malloc(N * sizeof(double));
    double *weights = (double *)malloc(N * sizeof(double));
    orc_fitness_terms(pop, N, G, sel_coeff, num_genes, logw);
    int rc = orc_sample_weights(num_genes, logw, N, G, avg_gene_num, avg_pairwise_dists,
                                no_control_genome_size, genome_size_penalty,
                                competition_strength, weights);
    if (rc == 0) rc = orc_draw_parents(weights, N, seed, gen, idx);
    free(num_genes);
    free(logw);
    free(weights);
    return rc;
}

/* ========================================================================= */
/* writers                                                                    */
/* ========================================================================= */
/* Rust `{}` for f64 (used at main.rs:481, :496, :546, :328): shortest digits
 * that round-trip, positional notation only, no trailing ".0". */
int orc_fmt_f64(double v, char *buf, size_t cap)
{
    if (isnan(v)) return snprintf(buf, cap, "NaN");
    if (isinf(v)) return snprintf(buf, cap, v < 0 ? "-inf" : "inf");
    if (v == 0.0) return snprintf(buf, cap, signbit(v) ? "-0" : "0");
    char tmp[64];
    int prec;
    for (prec = 0; prec < 17; prec++) {
        snprintf(tmp, sizeof tmp, "%.*e", prec, v);
        if (strtod(tmp, NULL) == v) break;
    }
    /* tmp = [-]d[.ddd]e[+-]XX */
    char digits[32];
    int nd = 0, neg = 0;
    const char *p = tmp;
    if (*p == '-') { neg = 1; p++; }
    for (; *p && *p != 'e'; p++) if (*p != '.') digits[nd++] = *p;
    int e10 = atoi(p + 1);
    while (nd > 1 && digits[nd - 1] == '0') nd--;      /* cannot happen for shortest, be safe */
    char out[400];
    int o = 0;
    if (neg) out[o++] = '-';
    if (e10 >= nd - 1) {
        for (int i = 0; i < nd; i++) out[o++] = digits[i];
        for (int i = 0; i < e10 - (nd - 1); i++) out[o++] = '0';
    } else if (e10 >= 0) {
        for (int i = 0; i <= e10; i++) out[o++] = digits[i];
        out[o++] = '.';
        for (int i = e10 + 1; i < nd; i++) out[o++] = digits[i];
    } else {
        out[o++] = '0';
        out[o++] = '.';
        for (int i = 0; i < -e10 - 1; i++) out[o++] = '0';
        for (int i = 0; i < nd; i++) out[o++] = digits[i];
    }
    out[o] = 0;
    return snprintf(buf, cap, "%s", out);
}

/* population.rs:865-897 */
int orc_write_matrix(const uint8_t *pop, uint64_t N, uint64_t ncols, int core, uint64_t core_genes,
                     const char *outpref)
{
    char path[4096];
    snprintf(path, sizeof path, "%s%s", outpref, core ? "_core_genome.csv" : "_pangenome.csv");
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    for (uint64_t i = 0; i < N; i++) {
        const uint8_t *row = pop + i * ncols;
        int first = 1;
        if (core) {
            for (uint64_t s = 0; s < ncols; s++) {
                if (!first) fputc(',', f);
                fputc(orc_int_to_base(row[s]), f);
                first = 0;
            }
        } else {
            for (uint64_t k = 0; k < core_genes; k++) {
                if (!first) fputc(',', f);
                fputc('1', f);
                first = 0;
            }
            for (uint64_t g = 0; g < ncols; g++) {
                if (!first) fputc(',', f);
                fprintf(f, "%u", (unsigned)row[g]);
                first = 0;
            }
        }
        fputc('\n', f);
    }
    fclose(f);
    return 0;
}

/* ========================================================================= */
/* reference-algorithm (event-driven) mode: CPU baseline ("port")             */
/* Same algorithm shape as the reference: Poisson count per row, weighted-    */
/* index binary search per event, serial gather, serial scatter, rows in      */
/* parallel exactly where the reference uses rayon.                           */
/* ========================================================================= */
typedef struct { uint64_t s[4]; } xo_rng;

static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void xo_seed(xo_rng *r, uint64_t a, uint64_t b, uint64_t c)
{
    uint64_t x = a * 0x9E3779B97F4A7C15ull ^ (b + 0x632BE59BD9B4E019ull) * 0xD1342543DE82EF95ull ^ c;
    for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&x);
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t xo_next(xo_rng *r)
{
    uint64_t *s = r->s;
    uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl64(s[3], 45);
    return result;
}
static inline double xo_f64(xo_rng *r) { return (double)(xo_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline float xo_f32(xo_rng *r) { return (float)(xo_next(r) >> 40) * (1.0f / 16777216.0f); }
static inline uint64_t xo_below(xo_rng *r, uint64_t n)
{
    return (uint64_t)(((unsigned __int128)xo_next(r) * n) >> 64);
}
static uint64_t xo_poisson(xo_rng *r, double mean)
{
    if (!(mean > 0.0)) return 0;
    if (mean < 10.0) {
        double lim = exp(-mean), p = 1.0;
        uint64_t k = 0;
        do { k++; p *= xo_f64(r); } while (p > lim);
        return k - 1;
    }
    double slam = sqrt(mean), loglam = log(mean);
    double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
    double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
    for (;;) {
        double U = xo_f64(r) - 0.5, V = xo_f64(r);
        double us = 0.5 - fabs(U);
        double kf = floor((2.0 * a / us + b) * U + mean + 0.43);
        if (us >= 0.07 && V <= vr) return (uint64_t)kf;
        if (kf < 0.0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(invalpha) - log(a / (us * us) + b)
            <= -mean + kf * loglam - lgamma(kf + 1.0)) return (uint64_t)kf;
    }
}

/* rand WeightedIndex<f32>: cumulative weights (len n-1) + total; sample =
 * uniform f32 in [0,total), partition_point(w <= x) (main.rs:394-403). */
typedef struct { float *cum; uint64_t n; float total; } widx_f32;
static void widx_build(widx_f32 *w, const float *weights, uint64_t n)
{
    w->n = n;
    w->cum = (float *)malloc((n > 1 ? n - 1 : 1) * sizeof(float));
    float total = weights[0];
    for (uint64_t i = 1; i < n; i++) { w->cum[i - 1] = total; total += weights[i]; }
    w->total = total;
}
static inline uint64_t widx_sample(const widx_f32 *w, xo_rng *r)
{
    float x = xo_f32(r) * w->total;
    uint64_t lo = 0, hi = w->n - 1;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (w->cum[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct orc_ref_sim {
    orc_params p;
    orc_derived d;
    uint64_t seed;
    int threads;
    uint64_t N, L, G;
    uint8_t *core, *acc;
    double *sel_coeff;
    widx_f32 core_w;
    widx_f32 pan_w[2];
    float *pan_weights[2];
    uint64_t shuffle_ctr;
    double competition;          /* --competition_strength (0: average_distance is never called) */
};

typedef void (*row_fn)(orc_ref_sim *, uint64_t row, void *ctx);
typedef struct { orc_ref_sim *s; row_fn fn; void *ctx; uint64_t begin, end; } par_arg;
static void *par_tramp(void *a_)
{
    par_arg *a = (par_arg *)a_;
    for (uint64_t r = a->begin; r < a->end; r++) a->fn(a->s, r, a->ctx);
    return NULL;
}
/* rows in parallel (the reference's `into_par_iter()` over axis 0) */
static void par_rows(orc_ref_sim *s, uint64_t n, row_fn fn, void *ctx)
{
    int T = s->threads < 1 ? 1 : s->threads;
    if ((uint64_t)T > n) T = (int)n;
    if (T <= 1) { for (uint64_t r = 0; r < n; r++) fn(s, r, ctx); return; }
    pthread_t *th = (pthread_t *)malloc(T * sizeof(pthread_t));
    par_arg *args = (par_arg *)malloc(T * sizeof(par_arg));
    for (int t = 0; t < T; t++) {
        args[t].s = s; args[t].fn = fn; args[t].ctx = ctx;
        args[t].begin = n * t / T; args[t].end = n * (t + 1) / T;
        pthread_create(&th[t], NULL, par_tramp, &args[t]);
    }
    for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
    free(th);
    free(args);
}

orc_ref_sim *orc_ref_create(const orc_params *p, uint64_t seed, int threads)
{
    orc_ref_sim *s = (orc_ref_sim *)calloc(1, sizeof(*s));
    s->p = *p;
    s->seed = seed;
    s->threads = threads;
    orc_derive(p, &s->d);
    s->N = p->pop_size; s->L = p->core_size; s->G = s->d.pan_size;
    s->core = (uint8_t *)malloc(s->N * s->L);
    s->acc = (uint8_t *)malloc(s->N * (s->G ? s->G : 1));
    uint8_t *v = (uint8_t *)malloc(s->L > s->G ? s->L : s->G);
    orc_init_core_vec(seed, s->L, v);
    orc_replicate(v, s->N, s->L, s->core);
    orc_init_acc_vec(seed, s->G, s->d.avg_gene_freq_adj, v);
    orc_replicate(v, s->N, s->G, s->acc);
    free(v);
    s->sel_coeff = (double *)calloc(s->G ? s->G : 1, sizeof(double));
    float *cw = (float *)malloc(s->L * sizeof(float));
    for (uint64_t i = 0; i < s->L; i++) cw[i] = 1.0f;               /* main.rs:284 */
    widx_build(&s->core_w, cw, s->L);
    free(cw);
    for (int c = 0; c < s->d.n_comp; c++) {
        s->pan_weights[c] = (float *)calloc(s->G, sizeof(float));    /* main.rs:342-345 */
        for (uint64_t g = s->d.comp_begin[c]; g < s->d.comp_end[c]; g++) s->pan_weights[c][g] = 1.0f;
        widx_build(&s->pan_w[c], s->pan_weights[c], s->G);
    }
    return s;
}

void orc_ref_destroy(orc_ref_sim *s)
{
    if (!s) return;
    free(s->core); free(s->acc); free(s->sel_coeff); free(s->core_w.cum);
    for (int c = 0; c < s->d.n_comp; c++) { free(s->pan_w[c].cum); free(s->pan_weights[c]); }
    free(s);
}

const uint8_t *orc_ref_core(const orc_ref_sim *s) { return s->core; }
const uint8_t *orc_ref_acc(const orc_ref_sim *s) { return s->acc; }

typedef struct { uint32_t gen; int comp; double lam; } mut_ctx;

/* population.rs:511-540 */
static void ref_mut_core_row(orc_ref_sim *s, uint64_t row, void *ctx_)
{
    mut_ctx *c = (mut_ctx *)ctx_;
    xo_rng r;
    xo_seed(&r, s->seed, ((uint64_t)c->gen << 8) | 1, row);
    uint8_t *p = s->core + row * s->L;
    uint64_t n_sites = xo_poisson(&r, c->lam);
    static const uint8_t core_vec[4][3] = { {2, 4, 8}, {1, 4, 8}, {1, 2, 8}, {1, 2, 4} };
    for (uint64_t k = 0; k < n_sites; k++) {
        uint64_t site = widx_sample(&s->core_w, &r);
        uint8_t value = p[site];
        const uint8_t *values = core_vec[(1 >> value) & 3];          /* :531, value <= 8 */
        p[site] = values[xo_below(&r, 3)];
    }
}
/* population.rs:486-510 */
static void ref_mut_acc_row(orc_ref_sim *s, uint64_t row, void *ctx_)
{
    mut_ctx *c = (mut_ctx *)ctx_;
    xo_rng r;
    xo_seed(&r, s->seed, ((uint64_t)c->gen << 8) | (2 + c->comp), row);
    uint8_t *p = s->acc + row * s->G;
    uint64_t n_sites = xo_poisson(&r, c->lam);
    for (uint64_t k = 0; k < n_sites; k++) {
        uint64_t site = widx_sample(&s->pan_w[c->comp], &r);
        p[site] = (p[site] == 0) ? 1 : 0;
    }
}

typedef struct {
    uint32_t gen; int comp; double lam; int core;
    uint64_t **loci; uint64_t **recips; uint8_t **vals; uint64_t *counts;
} rec_ctx;

/* population.rs:587-721 (phase 1) */
static void ref_rec_row(orc_ref_sim *s, uint64_t row, void *ctx_)
{
    rec_ctx *c = (rec_ctx *)ctx_;
    xo_rng r;
    xo_seed(&r, s->seed, ((uint64_t)c->gen << 8) | (c->core ? 8 : 9 + c->comp), row);
    uint64_t n_sites = xo_poisson(&r, c->lam);
    uint64_t *rec = (uint64_t *)malloc((n_sites ? n_sites : 1) * sizeof(uint64_t));
    uint64_t *loc = (uint64_t *)malloc((n_sites ? n_sites : 1) * sizeof(uint64_t));
    uint8_t *val = (uint8_t *)malloc(n_sites ? n_sites : 1);
    for (uint64_t k = 0; k < n_sites; k++) {
        uint64_t v = xo_below(&r, s->N - 1);
        rec[k] = v + (v >= row);                                   /* :616-619 */
        val[k] = 1;                                                /* :632 */
    }
    uint64_t n_loci = 0;
    if (!c->core) {
        const uint8_t *p = s->acc + row * s->G;
        float *nz = (float *)malloc(s->G * sizeof(float));         /* :636 clone */
        memcpy(nz, s->pan_weights[c->comp], s->G * sizeof(float));
        uint64_t total_0 = 0;
        for (uint64_t g = 0; g < s->G; g++) {                      /* :638-655 */
            int update = 0;
            if (nz[g] == 0.0f) update = 1;
            if (p[g] == 0) { nz[g] = 0.0f; update = 1; }
            if (update) total_0++;
        }
        if (total_0 < s->G) {                                      /* :672 */
            widx_f32 w;
            widx_build(&w, nz, s->G);
            for (uint64_t k = 0; k < n_sites; k++) loc[k] = widx_sample(&w, &r);
            n_loci = n_sites;
            free(w.cum);
        }
        free(nz);
    } else {
        const uint8_t *p = s->core + row * s->L;
        for (uint64_t k = 0; k < n_sites; k++) loc[k] = xo_below(&r, s->L); /* :687-690 */
        for (uint64_t k = 0; k < n_sites; k++) val[k] = p[loc[k]];          /* :693-695 */
        n_loci = n_sites;
    }
    c->loci[row] = loc; c->recips[row] = rec; c->vals[row] = val; c->counts[row] = n_loci;
}

/* population.rs:544-751 */
static void ref_recombine(orc_ref_sim *s, uint32_t gen, int core, int n_comp, const double *lams)
{
    uint64_t N = s->N;
    uint64_t ncols = core ? s->L : s->G;
    uint8_t *pop = core ? s->core : s->acc;
    for (int comp = 0; comp < n_comp; comp++) {
        if (lams[comp] == 0.0) continue;                           /* :558 */
        rec_ctx c;
        c.gen = gen; c.comp = comp; c.lam = lams[comp]; c.core = core;
        c.loci = (uint64_t **)calloc(N, sizeof(void *));
        c.recips = (uint64_t **)calloc(N, sizeof(void *));
        c.vals = (uint8_t **)calloc(N, sizeof(void *));
        c.counts = (uint64_t *)calloc(N, sizeof(uint64_t));
        par_rows(s, N, ref_rec_row, &c);
        /* :725-726 seeded shuffle of donor order */
        uint64_t *order = (uint64_t *)malloc(N * sizeof(uint64_t));
        for (uint64_t i = 0; i < N; i++) order[i] = i;
        xo_rng r;
        xo_seed(&r, s->seed, 0xC0FFEEull + s->shuffle_ctr++, gen);
        for (uint64_t i = N - 1; i >= 1; i--) {
            uint64_t j = xo_below(&r, i + 1);
            uint64_t t = order[i]; order[i] = order[j]; order[j] = t;
        }
        for (uint64_t oi = 0; oi < N; oi++) {                      /* :728-748 serial apply */
            uint64_t d = order[oi];
            for (uint64_t k = 0; k < c.counts[d]; k++)
                pop[c.recips[d][k] * ncols + c.loci[d][k]] = c.vals[d][k];
        }
        for (uint64_t i = 0; i < N; i++) { free(c.loci[i]); free(c.recips[i]); free(c.vals[i]); }
        free(c.loci); free(c.recips); free(c.vals); free(c.counts); free(order);
    }
}

/* main.rs:429-464 */
/* --competition_strength of the run (main.rs:438-440: average_distance is only called when it is > 0) */
void orc_ref_set_competition(orc_ref_sim *s, double strength) { s->competition = strength; }

static void ref_avg_row(orc_ref_sim *s, uint64_t row, void *ctx_)
{
    ((double *)ctx_)[row] = average_distance_row(s->acc, s->N, s->G, 0, s->p.core_genes, row);
}

int orc_ref_generation(orc_ref_sim *s, uint32_t gen)
{
    uint64_t N = s->N;
    double *avg = (double *)malloc(N * sizeof(double));
    for (uint64_t i = 0; i < N; i++) avg[i] = 1.0;                 /* main.rs:435 */
    if (s->competition > 0.0) par_rows(s, N, ref_avg_row, avg);    /* main.rs:438-440; population.rs:757 into_par_iter */
    uint32_t *idx = (uint32_t *)malloc(N * sizeof(uint32_t));
    int rc = orc_sample_indices(s->acc, N, s->G, s->seed, gen, s->d.avg_gene_num, avg,
                                s->sel_coeff, 0, 0.99, s->competition, idx);  /* main.rs:442-443 */
    free(avg);
    if (rc) { free(idx); return rc; }
    /* population.rs:450-465: fresh zeroed array + serial row copies */
    uint8_t *next = (uint8_t *)calloc(N * s->L, 1);
    orc_next_generation(s->core, N, s->L, idx, next);
    free(s->core); s->core = next;
    next = (uint8_t *)calloc(N * (s->G ? s->G : 1), 1);
    orc_next_generation(s->acc, N, s->G, idx, next);
    free(s->acc); s->acc = next;
    free(idx);
    mut_ctx mc;
    mc.gen = gen; mc.comp = 0; mc.lam = s->d.n_core_mutations;
    if (mc.lam != 0.0) par_rows(s, N, ref_mut_core_row, &mc);      /* main.rs:452 */
    for (int c = 0; c < s->d.n_comp; c++) {                        /* main.rs:455 */
        mc.comp = c; mc.lam = s->d.n_pan_mutations[c];
        if (mc.lam != 0.0) par_rows(s, N, ref_mut_acc_row, &mc);
    }
    if (s->p.HR_rate > 0.0) {                                      /* main.rs:459-461 */
        double l = s->d.n_recombinations_core;
        ref_recombine(s, gen, 1, 1, &l);
    }
    if (s->p.HGT_rate > 0.0)                                       /* main.rs:462-464 */
        ref_recombine(s, gen, 0, s->d.n_comp, s->d.n_recombinations_pan);
    return 0;
}

typedef struct { int core; const uint32_t *r1, *r2; double *out; } pw_ctx;
static void ref_pair_row(orc_ref_sim *s, uint64_t k, void *ctx_)
{
    pw_ctx *c = (pw_ctx *)ctx_;
    if (c->core)
        orc_pairwise_distances(s->core, s->N, s->L, 1, s->p.core_genes, 1, c->r1 + k, c->r2 + k, c->out + k);
    else
        orc_pairwise_distances(s->acc, s->N, s->G, 0, s->p.core_genes, 1, c->r1 + k, c->r2 + k, c->out + k);
}
void orc_ref_pairwise(const orc_ref_sim *s, int core, uint64_t P, const uint32_t *r1,
                      const uint32_t *r2, double *out)
{
    pw_ctx c = { core, r1, r2, out };
    par_rows((orc_ref_sim *)s, P, ref_pair_row, &c);
}
