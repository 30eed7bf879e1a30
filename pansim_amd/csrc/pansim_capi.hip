// pansim_capi.hip -- implementation of include/pansim_hip.h for gfx950 (MI355X).
//
// Host side of the per-generation hot path of bacpop/Pansim
// (pansim/src/population.rs, pansim/src/main.rs:259-553) above hand-written HIP
// kernels.  There is no CPU compute path in this library: every operator runs on
// the device and fails with PS_ERR_NO_DEVICE when none is present.
#include "../../include/pansim_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "acc_kernels.h"
#include "core_kernels.h"
#include "ps_common.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int ps_fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return ps_fail(e_ == hipErrorOutOfMemory ? PS_ERR_OOM : PS_ERR_NO_DEVICE,    \
                           "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                    \
    } while (0)

#define PSCHK(expr)              \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != PS_OK) return rc_; \
    } while (0)

extern "C" const char *ps_last_error(void) { return g_err.c_str(); }
extern "C" int ps_abi_version(void) { return 3; }
extern "C" int ps_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

// ---------------------------------------------------------------------------
// seeded host streams: stateless, indexed Philox draws
// ---------------------------------------------------------------------------
static inline ps_u4 hs_block(uint64_t seed, uint32_t stream, uint32_t gen, uint64_t blk)
{
    return ps_philox((uint32_t)blk, (uint32_t)(blk >> 32), gen, stream, (uint32_t)seed,
                     (uint32_t)(seed >> 32));
}
static inline double hs_f64(uint64_t seed, uint32_t stream, uint32_t gen, uint64_t n)
{
    const ps_u4 w = hs_block(seed, stream, gen, n >> 1);
    const uint64_t x = (n & 1) ? (((uint64_t)w.w << 32) | w.z) : (((uint64_t)w.y << 32) | w.x);
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
static inline uint32_t hs_u32(uint64_t seed, uint32_t stream, uint32_t gen, uint64_t n)
{
    const ps_u4 w = hs_block(seed, stream, gen, n >> 2);
    switch (n & 3) {
    case 0: return w.x;
    case 1: return w.y;
    case 2: return w.z;
    default: return w.w;
    }
}

// Poisson(lambda) by inversion over an integer threshold table (include/pansim_hip.h,
// ps_poisson_table): thr[j] = floor(P(K <= kmin + j) * 2^32) over lambda +- (12 sigma + 12).
static void hs_poisson_table(double lambda, uint32_t *kmin_out, std::vector<uint32_t> *thr)
{
    thr->clear();
    *kmin_out = 0;
    if (!(lambda > 0.0)) return;
    const double spread = 12.0 * std::sqrt(lambda) + 12.0;
    const double lo = std::floor(lambda - spread);
    const uint32_t kmin = lo > 0.0 ? (uint32_t)lo : 0u;
    const uint32_t kmax = (uint32_t)std::ceil(lambda + spread);
    const double loglam = std::log(lambda);
    double cdf = 0.0;
    for (uint32_t k = kmin; k <= kmax; k++) {
        cdf += std::exp((double)k * loglam - lambda - std::lgamma((double)k + 1.0));
        const double scaled = std::floor(cdf * 4294967296.0);
        thr->push_back(scaled >= 4294967295.0 ? 4294967295u : (uint32_t)scaled);
    }
    *kmin_out = kmin;
}

extern "C" uint32_t ps_poisson_table(double lambda, uint32_t *kmin_out, uint32_t *thr, uint32_t cap)
{
    std::vector<uint32_t> t;
    uint32_t kmin = 0;
    hs_poisson_table(lambda, &kmin, &t);
    if (t.empty() || t.size() > cap || !kmin_out || !thr) return 0;
    *kmin_out = kmin;
    memcpy(thr, t.data(), t.size() * sizeof(uint32_t));
    return (uint32_t)t.size();
}


// ---------------------------------------------------------------------------
// keyed dense plans
// ---------------------------------------------------------------------------
static uint32_t prob_to_u32(double p)
{
    const double x = std::floor(p * 4294967296.0);
    if (!(x > 0.0)) return 0u;
    if (x >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

// (the tests' CPU restatement evaluates the same expressions in the same order: the plans agree bit for bit)
static void make_core_plan(double lam_mut, double lam_hr, uint64_t L, ps_core_plan *plan)
{
    const double p = (lam_mut > 0.0) ? -std::expm1(-lam_mut / (double)L) : 0.0;
    const double q = (lam_hr > 0.0) ? -std::expm1(-lam_hr / (double)L) : 0.0;
    const double a = p * (1.0 - q) / 3.0;
    const double b = p * q / 3.0;
    const double c = (1.0 - p) * q;
    // k symbols of mass 1/64 per allele are decided by the symbol alone ...
    const uint32_t k = (uint32_t)std::floor(a * 64.0);
    const double a_left = a - (double)k / 64.0;
    // ... and the rest of the event mass goes to R residual symbols, inside which it is scaled by 64 / R
    const double m_res = a_left + a_left + a_left + b + b + b + c;
    uint32_t R = (uint32_t)std::ceil(m_res * 64.0);
    if (3u * k + R > 64u) R = 64u - 3u * k;
    const double scale = R ? 64.0 / (double)R : 0.0;
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
    if (plan->T[6] == 0u) R = 0u;       // nothing left for the residual symbols to decide
    plan->k = k;
    plan->R = R;
    plan->has_events = (k > 0u || R > 0u) ? 1u : 0u;
    uint32_t cs = 0;
    while (cs < 4u && 3u * k + R > (4u << cs)) cs++;
    plan->cshift = cs;
}

static uint32_t acc_flip_threshold(double lam, uint64_t n_genes)
{
    if (!(lam > 0.0) || n_genes == 0) return 0u;
    const double pf = -std::expm1(-2.0 * lam / (double)n_genes) / 2.0;
    return prob_to_u32(pf);
}

// ---------------------------------------------------------------------------
// struct Population (population.rs:164-170)
// ---------------------------------------------------------------------------
struct ps_population {
    ps_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool rates_set = false;
    // core: site-major u8
    uint8_t *state = nullptr;
    uint8_t *state2 = nullptr;       // second buffer of the out-of-place sweep (allocated at its first launch)
    int sweep_oop = -1;              // 0 = in place, 1 = out of place (state <-> state2 alternate), 2 = the same with nontemporal row
                                     // loads / stores, -1 = choose: 2 for the wave-per-row sweep (N <= 1024: at most 1 KiB x L more
                                     // memory; measured -2 % at cfg2, -9 % at cfg3), 0 for the block sweep (no gain at N = 65536)
    uint32_t pitch = 0, cpr = 0;
    ps_core_plan cplan{};
    bool nibble_safe = true;         // every byte is below 16 (the nibble-packed distance kernels; checked on load)
    bool onehot_safe = true;         // every byte is 1, 2, 4 or 8 (true for simulated states; checked on load)
    uint32_t *d_pack2 = nullptr;     // 2-bit packed copy of the matrix for the sampled-pair distances
    uint32_t *d_pair_part = nullptr; // partial pair counts per site range (tiled distance kernels)
    uint64_t pair_part_cap = 0;
    uint64_t pack2_cap = 0;
    // accessory: bit-packed, two views, ping-pong
    acc_dims d{};
    uint64_t *G[2] = { nullptr, nullptr };   // gene-major view (G[0] only): rebuilt from the rows when g_valid is false
    bool g_valid = false;
    // ps_sim, neutral selection: where the reduce pass of a binned HGT leaves the rows' gene counts (host-mapped), and
    // whether they describe the matrix as it is now (any other edit of the matrix clears the flag)
    int32_t *fuse_counts_out = nullptr;
    double *fuse_logw_out = nullptr;
    bool counts_fresh = false;
    hipStream_t stream_of_call = nullptr;   // launch_acc_hgt: its caller's stream
    uint64_t edit_epoch = 0;         // bumped by every edit of the matrix (load, generation step, HGT): what a cached result was computed from
    uint64_t *I[2] = { nullptr, nullptr };
    uint32_t *d_ptab[PS_MAX_COMP] = {};  // Poisson threshold tables of the HGT event counts (ps_set_rates)
    uint32_t ptab_kmin[PS_MAX_COMP] = {}, ptab_len[PS_MAX_COMP] = {};
    void *hgt_ovf_img = nullptr;         // binned HGT: image of the events that met a full bin (zero between launches)
    uint32_t hgt_bin_cap = 0;            // tests: bins of the binned HGT hold at most this many events (0 = sized for mean + 10 sigma)
    void *hgt_scratch = nullptr;         // slice images of the LDS-partitioned HGT kernel
    uint64_t hgt_scratch_cap = 0;
    uint64_t *I_snap = nullptr;          // light HGT: the pre-recombination snapshot written by the step before it (ps_sim)
    bool snap_valid = false;
    uint32_t hgt_slices = 0;             // tuning: event slices of the LDS-partitioned HGT kernel (0 = choose)
    uint32_t hgt_events_per_thread = 0;  // light HGT kernel: 0 = whole chip, else narrow launch (set by ps_sim)
    bool hgt_list_in_global = false;     // light HGT kernel: donor lists in global scratch (no LDS beside the block sweep)
    bool hgt_bin_list_in_global = false; // binned HGT, bin pass: the same (tests)
    uint32_t hgt_bin_prio = 0;           // binned HGT, bin pass: s_setprio level beside the sweep (PANSIM_HGT_BIN_PRIO)
    uint32_t *cnt = nullptr;
    int cur = 0;
    ps_acc_plan aplan{};
    // HGT donors sharded over the ranks / shards of a run (ps_set_donor_shard): this handle generates the events of the
    // donors [donor_lo, donor_lo + donor_cnt) into d_delta, the exchange ORs the shards' deltas, the result is ORed
    // into the matrix
    uint32_t donor_lo = 0, donor_cnt = 0;      // donor_cnt == 0: all donors, events applied directly
    uint64_t *d_delta = nullptr;
    uint64_t delta_words = 0;                  // N * GW rounded up to a multiple of 4096 words
    ps_exchange_fn exchange = nullptr;
    void *exchange_ctx = nullptr;
    // scratch
    uint32_t *d_idx = nullptr;       // N parents
    uint32_t *d_idxT = nullptr;      // transposed parents for the block sweep (16 x cpr)
    uint32_t *d_work = nullptr;      // wave sweep: chunk counters (2 sets x 8 x 128 bytes)
    uint32_t work_flags_ofs = 0;     // dword offset of the window sweep's two wide-segment flags behind the counters
    uint64_t sweep_launches = 0;     // parity selects the counter set
    uint64_t window_launches = 0;    // the same for the window sweep's per-segment counters
    double *d_log1p = nullptr;       // G
    int32_t *d_num_genes = nullptr;  // N
    double *d_logw = nullptr;        // N
    uint32_t *d_H = nullptr;         // all-pairs Hamming numerators N x N
    double *d_Dt = nullptr;          // all-pairs Jaccard distances (D-avg), N x N
    void *d_davg = nullptr;          // matrix-core D-avg: the padded bit rows + row counts
    uint64_t davg_cap = 0;
    int davg_form = 0;               // 0 = choose, 1 = LDS-tile popcount kernels, 2 = matrix cores (one kernel), 3 = matrix cores, two phases (the choice for row shards and N <= 24576)
    void *d_davg_in = nullptr;       // two-phase D-avg: u16 intersection counts of one band of rows
    uint64_t davg_in_cap = 0;
    uint32_t davg_nb = 0;            // matrix-core D-avg: B fragments per wave (0 = choose, 1 or 2; 4 in the two-phase form only)
    uint32_t davg_ib = 0;            // two-phase D-avg: individuals per workgroup of phase 2 (0 = choose, 16 or 32)
    bool davg_plain_division = false; // matrix-core D-avg: the compiler's f64 division in the epilogue ("davg_plain_division": A/B, tests)
    uint64_t H_cap = 0;
    int pair_mode = 0;               // 0 auto, 1 sampled kernel, 2 all-pairs kernel (tuning/tests)
    int last_pair_form = 0;          // kernel form of the last core pair-count call (ps_last_pair_form)
    uint32_t pair_ranges = 0;        // tests: site ranges of the tiled sampled-pair kernels (0 = choose); the 16-bit cap still applies
    int hgt_mode = 0;                // 0 auto, 1 one atomic per event, 2 binned by recipient partition + LDS images
    void *d_pairs = nullptr;         // sorted r1 | r2 | perm | outA | outB | tstart | tcount
    uint32_t pair_threads = 0;       // entries of the thread table (tiled distance kernel)
    uint64_t pairs_cap = 0, pairs_cached = 0;
    const uint32_t *pairs_src1 = nullptr, *pairs_src2 = nullptr;   // caller's arrays the cache was built from (ps_sim's own list only)
    bool pairs_tiled = false;        // the cached list is sorted and has a thread table (tiled distance kernels)
    std::vector<uint32_t> h_r1, h_r2; // the caller's list the device copy was built from
    uint32_t lds_limit = 160 * 1024;
    uint32_t sweep_blocks_per_cu = 8;   // wave-per-row sweep: resident 256-thread blocks per CU (capped by LDS)
    uint32_t sweep_rows = 4;            // (accepted and ignored since round 6: a wave takes the 4 sites of a level-1 block group per iteration)
    bool force_block_sweep = false;     // tests: run the block sweep on small populations
    bool force_inline_sweep = false;    // tests: run the inline (queue-free) block sweep
    uint32_t block_waves = 0;           // block sweep: waves per workgroup (0 = auto: 4, 8 or 16)
    bool no_block_preload = false;      // tests: block sweep reads its parent indices per batch
    uint32_t block_batch = 0;           // block sweep: segments per wave batch (0 = 4, falling back to 2; 2 = force 2)
    uint32_t sweep_queue_cap = 0;       // tests: the sweeps treat their candidate queues / HR lists as this short (0 = real size)
    uint32_t hgt_apply_threads = 1024;  // binned HGT, LDS-image pass: threads per workgroup (256 / 512 / 1024; "hgt_apply_threads")
    uint32_t window_blocks_per_cu = 0;  // window sweep: workgroups per CU (0 = PS_WBPC)
    uint32_t free_cus_per_xcd = 0;      // core stream created with a CU mask that leaves this many CUs per XCD to other streams
    bool exchange_beside_sweep = false; // donor-sharded HGT: the next sweep waits for the LDS-image pass only (ps_sim sets it)
    int last_sweep_form = 0;            // PS_SWEEP_FORM_* of the last core sweep launch (ps_last_sweep_form)
    int window_sweep = -1;              // window sweep for N > 1024 when the parents are sorted: -1 = choose, 0 = never, 1 = whenever possible
    // Row order at the boundary.  A ps_sim stores the children of a generation in ascending parent order;
    // the reference's row k is the child of draw k (population.rs:443, main.rs:445-447).  row_slot[k] = the internal row
    // of output row k (empty = they coincide); rows_refresh (set by the owning ps_sim) brings it up to date before an
    // output.  A direct ps_load_matrix / ps_next_generation / ps_step on the handle makes the two orders coincide again.
    int (*rows_refresh)(void *ctx) = nullptr;
    void *rows_ctx = nullptr;
    std::vector<uint32_t> row_slot;
    uint32_t *d_row_slot = nullptr;
    bool rows_overridden = false;
    uint32_t *h_flag = nullptr, *d_flag = nullptr;   // host-mapped sticky device error word
    unsigned long long *h_stamps = nullptr, *d_stamps = nullptr;   // diagnostic phase stamps
};

static int use_device(const ps_population *p)
{
    HIPCHK(hipSetDevice(p->device));
    return PS_OK;
}

// output rows in the reference's order: row_slot current (see struct ps_population), nullptr = internal order
static int rows_current(ps_population *p, const uint32_t **slot_out)
{
    if (p->rows_overridden) p->row_slot.clear();
    else if (p->rows_refresh) PSCHK(p->rows_refresh(p->rows_ctx));
    *slot_out = p->row_slot.empty() ? nullptr : p->row_slot.data();
    return PS_OK;
}

// v[k] <- v[slot[k]] for a per-individual result
template <typename T>
static void rows_permute(T *v, const uint32_t *slot, uint64_t n)
{
    if (!slot) return;
    std::vector<T> tmp(v, v + n);
    for (uint64_t k = 0; k < n; k++) v[k] = tmp[slot[k]];
}

extern "C" void ps_population_destroy(ps_population *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    void *ptrs[] = { p->d_row_slot, p->state, p->state2, p->d_delta, p->hgt_ovf_img, p->G[0], p->G[1], p->I[0], p->I[1], p->I_snap, p->d_ptab[0], p->d_ptab[1], p->hgt_scratch, p->cnt, p->d_idx, p->d_idxT, p->d_work,
                     p->d_log1p, p->d_num_genes, p->d_logw, p->d_pairs, p->d_H, p->d_Dt, p->d_davg, p->d_davg_in, p->d_pack2, p->d_pair_part };
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    if (p->h_flag) (void)hipHostFree(p->h_flag);
    if (p->h_stamps) {
        if (p->d_stamps) {
            (void)hipMemcpy(p->h_stamps, p->d_stamps, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            (void)hipFree(p->d_stamps);
        }
        if (getenv("PANSIM_PRINT_STAMPS"))
            for (int k = 0; k < 32; k++) fprintf(stderr, "stamp[%d] = %llu\n", k, p->h_stamps[k]);
        (void)hipHostFree(p->h_stamps);
    }
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

static int pop_create_impl(const ps_config *cfg, const uint8_t *init_vec, ps_population *p)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return ps_fail(PS_ERR_NO_DEVICE, "no HIP device is visible: libpansim_hip has no CPU path");
    if (cfg->device >= 0) {
        if (cfg->device >= ndev) return ps_fail(PS_ERR_INVALID, "device %d out of range", cfg->device);
        p->device = cfg->device;
    } else {
        HIPCHK(hipGetDevice(&p->device));
    }
    PSCHK(use_device(p));
    // PANSIM_SWEEP_FREE_CUS=n (experiment, round 5): the core stream is created with a CU mask that leaves n CUs of every
    // XCD to the other streams -- kernels that cannot share a SIMD with resident sweep waves (RCCL's 264-register
    // rcclGenericKernel) then start beside the sweep instead of behind it.  KFD deals mask bits round-robin over the XCCs
    // (bit b -> XCC b % 8): clearing the first 8 n bits takes n CUs from each (scripts/ubench/cu_mask.hip).
    if (cfg->core)
        if (const char *e = getenv("PANSIM_SWEEP_FREE_CUS")) p->free_cus_per_xcd = (uint32_t)std::max(0, std::min(8, atoi(e)));
    if (p->free_cus_per_xcd) {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, p->device));
        std::vector<uint32_t> mask((size_t)(prop.multiProcessorCount + 31) / 32, 0xFFFFFFFFu);
        for (uint32_t b = 0; b < 8u * p->free_cus_per_xcd && b < (uint32_t)prop.multiProcessorCount; b++) mask[b / 32u] &= ~(1u << (b % 32u));
        if (hipExtStreamCreateWithCUMask(&p->stream, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            p->free_cus_per_xcd = 0;            // (not permitted on this box: an ordinary stream, all CUs)
            HIPCHK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        }
    } else
    HIPCHK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    if (const char *e = getenv("PANSIM_SWEEP_BLOCKS_PER_CU")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 8) p->sweep_blocks_per_cu = (uint32_t)v;
    }
    if (const char *e = getenv("PANSIM_HGT_MODE")) p->hgt_mode = atoi(e);
    if (const char *e = getenv("PANSIM_SWEEP_OOP")) p->sweep_oop = std::max(-1, std::min(2, atoi(e)));
    if (const char *e = getenv("PANSIM_WINDOW_SWEEP")) p->window_sweep = std::max(-1, std::min(1, atoi(e)));
    if (const char *e = getenv("PANSIM_ACC_LDS_LIMIT")) { const int v = atoi(e); if (v >= 1024 && v <= 160 * 1024 && !cfg->core) p->lds_limit = (uint32_t)v; }
    if (const char *e = getenv("PANSIM_HGT_APPLY_THREADS")) { const int v = atoi(e); if (v == 256 || v == 512 || v == 1024) p->hgt_apply_threads = (uint32_t)v; }
    if (const char *e = getenv("PANSIM_WINDOW_BPC")) p->window_blocks_per_cu = (uint32_t)std::max(0, std::min(8, atoi(e)));
    if (const char *e = getenv("PANSIM_BLOCK_BATCH")) p->block_batch = (uint32_t)atoi(e);
    if (const char *e = getenv("PANSIM_BLOCK_WAVES")) p->block_waves = (uint32_t)atoi(e);
    if (const char *e = getenv("PANSIM_HGT_SLICES")) p->hgt_slices = (uint32_t)atoi(e);
    if (const char *e = getenv("PANSIM_HGT_BIN_LIST_GLOBAL")) p->hgt_bin_list_in_global = atoi(e) != 0;
    if (const char *e = getenv("PANSIM_HGT_BIN_PRIO")) p->hgt_bin_prio = (uint32_t)std::max(0, std::min(3, atoi(e)));
    if (const char *e = getenv("PANSIM_SWEEP_ROWS")) {
        const int v = atoi(e);
        if (v >= 2 && v <= 4) p->sweep_rows = (uint32_t)v;
    }
    const uint64_t N = cfg->pop_size, C = cfg->ncols;
    HIPCHK(hipMalloc(&p->d_idx, std::max<uint64_t>(N, 1) * sizeof(uint32_t)));
    HIPCHK(hipHostMalloc(&p->h_flag, sizeof(uint32_t), hipHostMallocMapped));
    *p->h_flag = 0;
    HIPCHK(hipHostMalloc(&p->h_stamps, 32 * sizeof(unsigned long long), hipHostMallocMapped));
    memset(p->h_stamps, 0, 32 * sizeof(unsigned long long));
    HIPCHK(hipMalloc(&p->d_stamps, 32 * sizeof(unsigned long long)));      // (device memory: 64-bit atomics on mapped host memory are not dependable)
    HIPCHK(hipMemset(p->d_stamps, 0, 32 * sizeof(unsigned long long)));
    HIPCHK(hipHostGetDevicePointer((void **)&p->d_flag, p->h_flag, 0));
    uint8_t *d_vec = nullptr;
    HIPCHK(hipMalloc(&d_vec, std::max<uint64_t>(C, 1)));
    if (C) HIPCHK(hipMemcpyAsync(d_vec, init_vec, C, hipMemcpyHostToDevice, p->stream));
    if (cfg->core) {
        p->pitch = (uint32_t)((N + 127) / 128 * 128);
        p->cpr = p->pitch / 16;
        HIPCHK(hipMalloc(&p->state, std::max<uint64_t>(C, 1) * p->pitch));
        HIPCHK(hipMalloc(&p->d_idxT, 16ull * p->cpr * sizeof(uint32_t)));
        {
            // chunk counters of the dynamic row assignment: 2 sets (alternating by launch) x 8 groups (wave sweep) or
            // x one per 1024-child segment (window sweep), 128 bytes apart
            const uint64_t nctr = 8 * std::max<uint64_t>(1, std::min<uint64_t>((N + 1023) / 1024, 4096));      // (window sweep: N <= 2^22)
            // (+ 256 bytes: the window sweep's two "a segment needs the second launch" flags)
            HIPCHK(hipMalloc(&p->d_work, 2 * nctr * 128 + 256));
            HIPCHK(hipMemsetAsync(p->d_work, 0, 2 * nctr * 128 + 256, p->stream));
            p->work_flags_ofs = (uint32_t)(2 * nctr * 32);
        }
        if (C) {
            const uint64_t total = C * p->cpr;
            const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 65536);
            core_init_kernel<<<blocks, 256, 0, p->stream>>>(p->state, d_vec, (uint32_t)N, p->pitch,
                                                            (uint32_t)C);
        }
    } else {
        p->d.N = (uint32_t)N;
        p->d.G = (uint32_t)C;
        p->d.W = (uint32_t)((N + 63) / 64);
        p->d.GW = (uint32_t)((C + 63) / 64);
        const uint64_t nG = std::max<uint64_t>((uint64_t)p->d.G * p->d.W, 1) * 8;
        const uint64_t nI = std::max<uint64_t>((uint64_t)p->d.N * p->d.GW, 1) * 8;
        HIPCHK(hipMalloc(&p->G[0], nG));
        for (int k = 0; k < 2; k++) HIPCHK(hipMalloc(&p->I[k], nI));
        if (C > 65536) return ps_fail(PS_ERR_INVALID, "at most 65536 accessory genes are supported");
        // events per (compartment, donor) + the dynamic item counter of the HGT kernel
        HIPCHK(hipMalloc(&p->cnt, (std::max<uint64_t>(N, 1) * PS_MAX_COMP + 64) * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&p->d_log1p, std::max<uint64_t>(C, 1) * sizeof(double)));
        HIPCHK(hipMalloc(&p->d_num_genes, std::max<uint64_t>(N, 1) * sizeof(int32_t)));
        HIPCHK(hipMalloc(&p->d_logw, std::max<uint64_t>(N, 1) * sizeof(double)));
        const uint64_t total = (uint64_t)p->d.G * p->d.W + (uint64_t)p->d.N * p->d.GW;
        if (total)
            acc_init_kernel<<<(uint32_t)((total + 255) / 256), 256, 0, p->stream>>>(p->G[0], p->I[0],
                                                                                d_vec, p->d);
        p->g_valid = true;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipFree(d_vec));
    return PS_OK;
}

extern "C" int ps_population_create(const ps_config *cfg, const uint8_t *init_vec, ps_population **out)
{
    if (!cfg || !out) return ps_fail(PS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->pop_size < 1) return ps_fail(PS_ERR_INVALID, "pop_size must be >= 1");
    if (cfg->pop_size > 0xFFFFFFFFull - 128 || cfg->global_cols > 0xFFFFFFFFull
        || cfg->ncols > 0xFFFFFFFFull)
        return ps_fail(PS_ERR_INVALID, "pop_size and column counts must fit 32 bits");
    if (cfg->col_offset + cfg->ncols > cfg->global_cols)
        return ps_fail(PS_ERR_INVALID, "column shard [%llu,%llu) exceeds global_cols %llu",
                       (unsigned long long)cfg->col_offset,
                       (unsigned long long)(cfg->col_offset + cfg->ncols),
                       (unsigned long long)cfg->global_cols);
    if (!cfg->core && (cfg->col_offset != 0 || cfg->ncols != cfg->global_cols))
        return ps_fail(PS_ERR_INVALID, "the accessory matrix is replicated, not sharded");
    if (cfg->ncols && !init_vec) return ps_fail(PS_ERR_INVALID, "init_vec is null");
    ps_population *p = new ps_population();
    p->cfg = *cfg;
    const int rc = pop_create_impl(cfg, init_vec, p);
    if (rc != PS_OK) {
        const std::string keep = g_err;
        ps_population_destroy(p);
        g_err = keep;
        return rc;
    }
    *out = p;
    return PS_OK;
}

extern "C" int ps_init_vector(uint64_t seed, int core, uint64_t col_offset, uint64_t ncols,
                              double avg_gene_freq, uint8_t *out)
{
    if (!out && ncols) return ps_fail(PS_ERR_INVALID, "null output");
    for (uint64_t c = 0; c < ncols; c++) {
        const uint64_t g = col_offset + c;
        if (core)   // population.rs:201-204
            out[c] = (uint8_t)(1u << (hs_u32(seed, PS_STREAM_INIT_CORE, 0, g) >> 30));
        else        // population.rs:217-218
            out[c] = hs_f64(seed, PS_STREAM_INIT_ACC, 0, g) < avg_gene_freq ? 1 : 0;
    }
    return PS_OK;
}

// device-side sticky errors become loud host errors at every synchronisation point
static int check_device_flag(ps_population *p)
{
    if (p->h_flag && *p->h_flag != 0)
        return ps_fail(PS_ERR_STATE, "a device error flag is set (%u)", *p->h_flag);
    return PS_OK;
}

static int sync_checked(ps_population *p)
{
    HIPCHK(hipStreamSynchronize(p->stream));
    return check_device_flag(p);
}

extern "C" int ps_sync(ps_population *p)
{
    if (!p) return ps_fail(PS_ERR_INVALID, "null handle");
    PSCHK(use_device(p));
    return sync_checked(p);
}

extern "C" int ps_set_tuning(ps_population *p, const char *key, int64_t value)
{
    if (!p || !key) return ps_fail(PS_ERR_INVALID, "null argument");
    const std::string k(key);
    if (k == "sweep_blocks_per_cu") {
        if (value < 1 || value > 8) return ps_fail(PS_ERR_INVALID, "sweep_blocks_per_cu must be 1..8");
        p->sweep_blocks_per_cu = (uint32_t)value;
    } else if (k == "sweep_rows") {
        if (value < 2 || value > 4) return ps_fail(PS_ERR_INVALID, "sweep_rows must be 2..4");
        p->sweep_rows = (uint32_t)value;
    } else if (k == "sweep_out_of_place") {
        if (value < -1 || value > 2) return ps_fail(PS_ERR_INVALID, "sweep_out_of_place must be -1 (choose), 0 (in place), 1 (out of place) or 2 (out of place, nontemporal loads and stores)");
        p->sweep_oop = (int)value;
    } else if (k == "hgt_apply_threads") {
        if (value != 256 && value != 512 && value != 1024) return ps_fail(PS_ERR_INVALID, "hgt_apply_threads must be 256, 512 or 1024");
        p->hgt_apply_threads = (uint32_t)value;
    } else if (k == "window_blocks_per_cu") {
        if (value < 0 || value > 8) return ps_fail(PS_ERR_INVALID, "window_blocks_per_cu must be 0 (choose) or 1..8");
        p->window_blocks_per_cu = (uint32_t)value;
    } else if (k == "davg_form") {
        if (value < 0 || value > 3)
            return ps_fail(PS_ERR_INVALID, "davg_form must be 0 (choose), 1 (LDS-tile popcount kernels), 2 (matrix cores, one kernel) or 3 (matrix cores, two phases)");
        p->davg_form = (int)value;
    } else if (k == "davg_plain_division") {
        p->davg_plain_division = value != 0;
    } else if (k == "davg_ib") {
        if (value != 0 && value != 16 && value != 32) return ps_fail(PS_ERR_INVALID, "davg_ib must be 0 (choose), 16 or 32");
        p->davg_ib = (uint32_t)value;
    } else if (k == "davg_nb") {
        if (value < 0 || value > 4 || value == 3) return ps_fail(PS_ERR_INVALID, "davg_nb must be 0 (choose), 1, 2 or (two-phase form only) 4");
        p->davg_nb = (uint32_t)value;
    } else if (k == "hgt_mode") {
        if (value < 0 || value > 2) return ps_fail(PS_ERR_INVALID, "hgt_mode must be 0 (auto), 1 (one atomic per event) or 2 (binned by recipient partition, two passes)");
        p->hgt_mode = (int)value;
    } else if (k == "pair_mode") {
        if (value < 0 || value > 7) return ps_fail(PS_ERR_INVALID, "pair_mode must be 0 (auto), 1 (sampled), 2 (all pairs), 3 (sampled, nibble form even for one-hot matrices), 4 (transposed bit strings, streamed per pair), 5 (all pairs, xor + popcount tiles even for one-hot matrices), 6 (all pairs, i8 matrix cores when the matrix is one-hot; mode 2 takes the FP4 form) or 7 (the FP4 form on three +-1 features per site)");
        p->pair_mode = (int)value;
    } else if (k == "pair_ranges") {
        if (value < 0 || value > 65535) return ps_fail(PS_ERR_INVALID, "pair_ranges must be 0 (choose)..65535");
        p->pair_ranges = (uint32_t)value;
    } else if (k == "force_block_sweep") {
        p->force_block_sweep = value != 0;
    } else if (k == "force_inline_sweep") {
        p->force_inline_sweep = value != 0;
    } else if (k == "hgt_slices") {
        if (value < 0 || value > 4096) return ps_fail(PS_ERR_INVALID, "hgt_slices must be 0..4096");
        p->hgt_slices = (uint32_t)value;
    } else if (k == "block_batch") {
        if (value != 0 && value != 2 && value != 4) return ps_fail(PS_ERR_INVALID, "block_batch must be 0 (choose), 2 or 4");
        p->block_batch = (uint32_t)value;
    } else if (k == "hgt_bin_cap") {
        if (value < 0 || value > (1 << 30)) return ps_fail(PS_ERR_INVALID, "hgt_bin_cap must be 0 (sized by the rates)..2^30");
        p->hgt_bin_cap = (uint32_t)value;
    } else if (k == "hgt_events_per_thread") {
        if (value < 0 || value > 100000) return ps_fail(PS_ERR_INVALID, "hgt_events_per_thread must be 0 (whole chip)..100000");
        p->hgt_events_per_thread = (uint32_t)value;
    } else if (k == "window_sweep") {
        if (value < -1 || value > 1) return ps_fail(PS_ERR_INVALID, "window_sweep must be -1 (choose), 0 (block sweep) or 1 (window sweep where the parents are sorted)");
        p->window_sweep = (int)value;
    } else if (k == "sweep_queue_cap") {
        if (value < 0 || value > 65536) return ps_fail(PS_ERR_INVALID, "sweep_queue_cap must be 0 (real size)..65536");
        p->sweep_queue_cap = (uint32_t)value;
    } else if (k == "hgt_list_in_global") {
        p->hgt_list_in_global = value != 0;
    } else if (k == "hgt_bin_list_in_global") {
        p->hgt_bin_list_in_global = value != 0;
    } else if (k == "no_block_preload") {
        p->no_block_preload = value != 0;
    } else if (k == "block_waves") {
        if (value != 0 && value != 4 && value != 8 && value != 16) return ps_fail(PS_ERR_INVALID, "block_waves must be 0 (auto), 4, 8 or 16");
        p->block_waves = (uint32_t)value;
    } else if (k == "lds_limit") {
        if (value < 1024 || value > 160 * 1024) return ps_fail(PS_ERR_INVALID, "lds_limit must be 1 KiB..160 KiB");
        p->lds_limit = (uint32_t)value;
        p->pairs_cached = 0;          // the cached pair list is sorted / has a thread table depending on the LDS tile
    } else {
        return ps_fail(PS_ERR_INVALID, "unknown tuning key %s", key);
    }
    return PS_OK;
}

// ---------------------------------------------------------------------------
// whole-matrix load / read (individual-major u8 at the boundary)
// ---------------------------------------------------------------------------
extern "C" int ps_load_matrix(ps_population *p, const uint8_t *rows)
{
    if (!p || !rows) return ps_fail(PS_ERR_INVALID, "null argument");
    PSCHK(use_device(p));
    const uint64_t N = p->cfg.pop_size, C = p->cfg.ncols;
    if (N * C == 0) return PS_OK;
    uint8_t *d_rows = nullptr;
    HIPCHK(hipMalloc(&d_rows, N * C));
    HIPCHK(hipMemcpyAsync(d_rows, rows, N * C, hipMemcpyHostToDevice, p->stream));
    if (p->cfg.core) {
        bool safe = true;
        bool onehot = true;
        for (uint64_t k = 0; k < N * C; k++) {
            const uint8_t b = rows[k];
            if (b > 15) { safe = false; onehot = false; break; }
            if (b != 1 && b != 2 && b != 4 && b != 8) onehot = false;
        }
        p->nibble_safe = safe;
        p->onehot_safe = onehot;
        dim3 grid((uint32_t)((C + 63) / 64), (uint32_t)((p->pitch + 63) / 64));
        core_transpose_kernel<true><<<grid, 256, 0, p->stream>>>(p->state, d_rows, (uint32_t)N,
                                                                 p->pitch, C);
    } else {
        const uint64_t nI = (uint64_t)p->d.N * p->d.GW;
        p->snap_valid = false;
        acc_pack_rows_kernel<<<(uint32_t)((nI + 255) / 256), 256, 0, p->stream>>>(d_rows, p->I[p->cur],
                                                                              p->d);
        p->g_valid = false;
        p->counts_fresh = false;
        p->edit_epoch++;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipFree(d_rows));
    p->rows_overridden = true;       // (of a ps_sim's handle: the loaded order is the order of every output until its next generation)
    return PS_OK;
}

extern "C" int ps_read_matrix(ps_population *p, uint8_t *rows)
{
    if (!p || !rows) return ps_fail(PS_ERR_INVALID, "null argument");
    PSCHK(use_device(p));
    const uint64_t N = p->cfg.pop_size, C = p->cfg.ncols;
    if (N * C == 0) return PS_OK;
    uint8_t *d_rows = nullptr;
    HIPCHK(hipMalloc(&d_rows, N * C));
    if (p->cfg.core) {
        dim3 grid((uint32_t)((C + 63) / 64), (uint32_t)((N + 63) / 64));
        core_transpose_kernel<false><<<grid, 256, 0, p->stream>>>(p->state, d_rows, (uint32_t)N,
                                                                  p->pitch, C);
    } else {
        acc_unpack_rows_kernel<<<(uint32_t)((N * C + 255) / 256), 256, 0, p->stream>>>(
            p->I[p->cur], d_rows, p->d);
    }
    HIPCHK(hipGetLastError());
    const uint32_t *slot = nullptr;
    PSCHK(rows_current(p, &slot));
    HIPCHK(hipMemcpyAsync(rows, d_rows, N * C, hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipFree(d_rows));
    if (slot) {
        // row k of the output = the child of draw k (a ps_sim's handle: population.rs:443, main.rs:445-447)
        std::vector<uint8_t> tmp(rows, rows + N * C);
        for (uint64_t k = 0; k < N; k++) memcpy(rows + k * C, tmp.data() + (uint64_t)slot[k] * C, C);
    }
    return PS_OK;
}

// ---------------------------------------------------------------------------
// rates
// ---------------------------------------------------------------------------
extern "C" int ps_set_rates(ps_population *p, int n_comp, const double *lam_mut, const double *lam_rec,
                            const uint64_t *comp_begin, const uint64_t *comp_end)
{
    if (!p) return ps_fail(PS_ERR_INVALID, "null handle");
    if (n_comp < 0 || n_comp > PS_MAX_COMP) return ps_fail(PS_ERR_INVALID, "n_comp must be 0..2");
    for (int c = 0; c < n_comp; c++) {
        // statrs Poisson::new(lambda).unwrap() panics on lambda <= 0 or NaN; 0 is skipped
        // before (population.rs:480, :558)
        if (!(lam_mut[c] >= 0.0) || !(lam_rec[c] >= 0.0) || std::isinf(lam_mut[c]) || std::isinf(lam_rec[c]))
            return ps_fail(PS_ERR_INVALID, "rates must be finite and >= 0");
        if (comp_begin[c] > comp_end[c] || comp_end[c] > p->cfg.global_cols)
            return ps_fail(PS_ERR_INVALID, "compartment range out of bounds");
    }
    if (p->cfg.core) {
        if (n_comp != 1) return ps_fail(PS_ERR_INVALID, "the core matrix has one compartment (main.rs:276, :279)");
        if (p->cfg.pop_size < 2 && lam_rec[0] > 0.0)
            return ps_fail(PS_ERR_INVALID, "recombination needs pop_size >= 2 (population.rs:584 panics)");
        make_core_plan(lam_mut[0], lam_rec[0], p->cfg.global_cols, &p->cplan);
    } else {
        p->aplan = ps_acc_plan{};
        p->aplan.n_comp = n_comp;
        for (int c = 0; c < n_comp; c++) {
            if (p->cfg.pop_size < 2 && lam_rec[c] > 0.0)
                return ps_fail(PS_ERR_INVALID, "recombination needs pop_size >= 2 (population.rs:584 panics)");
            p->aplan.comp_begin[c] = (uint32_t)comp_begin[c];
            p->aplan.comp_end[c] = (uint32_t)comp_end[c];
            p->aplan.flip_thr[c] = acc_flip_threshold(lam_mut[c], comp_end[c] - comp_begin[c]);
            p->aplan.lam_rec[c] = lam_rec[c];
        }
        // Poisson threshold tables of the per-donor HGT event counts (population.rs:599)
        PSCHK(use_device(p));
        for (int c = 0; c < PS_MAX_COMP; c++) {
            if (p->d_ptab[c]) { HIPCHK(hipStreamSynchronize(p->stream)); HIPCHK(hipFree(p->d_ptab[c])); }
            p->d_ptab[c] = nullptr;
            p->ptab_len[c] = 0;
            if (c >= n_comp || !(lam_rec[c] > 0.0)) continue;
            std::vector<uint32_t> thr;
            hs_poisson_table(lam_rec[c], &p->ptab_kmin[c], &thr);
            HIPCHK(hipMalloc(&p->d_ptab[c], thr.size() * sizeof(uint32_t)));
            HIPCHK(hipMemcpy(p->d_ptab[c], thr.data(), thr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            p->ptab_len[c] = (uint32_t)thr.size();
        }
    }
    p->rates_set = true;
    return PS_OK;
}

// ---------------------------------------------------------------------------
// core sweep launch
// ---------------------------------------------------------------------------
// queue of a wave of the wave / window sweeps: every RESIDUAL cell of one batch (4 rows x `cells` cells, R / 64 of them),
// with 10 standard deviations to spare (a full queue only sends the batch to the queue-free redo); `twice`: the wave
// sweep with HR parks a second word per entry in the queue's upper half.  0 = the plan is not one the queued sweeps
// take: they need every event symbol below 8 (cshift <= 1) and at most one symbol-decided allele class each (k <= 1: the
// allele index is then planes 4-5 as they stand)
static uint32_t sweep_queue_entries(const ps_core_plan &pl, uint32_t cells, bool twice)
{
    if (!pl.has_events) return 16u;
    if (pl.cshift > 1u || pl.k > 1u) return 0u;
    const double m = (double)PS_BATCH_ROWS * (double)cells * (double)pl.R / 64.0;
    const uint32_t n = ((uint32_t)std::ceil(m + 10.0 * std::sqrt(m) + 16.0) + 15u) & ~15u;
    return twice ? 2u * n : n;
}

template <bool GA, bool MU, bool HR>
static int launch_core_sweep_wave(ps_population *p, core_sweep_args a, hipStream_t st)
{
    const uint32_t block = 256u, wpb = block / 64u;
    a.qcap = sweep_queue_entries(a.plan, std::min(1024u, (uint32_t)p->cfg.pop_size), HR);
    // (the two sets of chunk counters alternate among the launches that USE them: a launch zeroes the other set)
    a.launch_parity = (uint32_t)(p->sweep_launches++ & 1u);
    const uint32_t lds = wpb * ps_wave_lds(a.qcap);
    const uint32_t batches = (a.rows + (a.site_offset & 3u) + PS_BATCH_ROWS - 1u) / PS_BATCH_ROWS;
    const uint32_t want = (batches + wpb - 1u) / wpb;
    const uint32_t fit = std::max(1u, std::min(8u, p->lds_limit / lds));
    const uint32_t bpc = std::min(p->sweep_blocks_per_cu, fit);
    const uint32_t grid = std::max(8u, std::min((want + 7u) & ~7u, 256u * bpc));   // a multiple of the 8 groups
    if (a.nt) hipLaunchKernelGGL((core_sweep_wave_kernel<GA, MU, HR, true>), dim3(grid), dim3(block), lds, st, a);
    else hipLaunchKernelGGL((core_sweep_wave_kernel<GA, MU, HR, false>), dim3(grid), dim3(block), lds, st, a);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

// The wave-per-row sweep applies the symbol-decided mutations in registers and queues the residual cells of a batch; it
// is selected for the plans sweep_queue_entries admits, when the queue fits the LDS beside the rows.
static bool wave_sweep_eligible(ps_population *p, bool mu, bool hr)
{
    if (p->pitch > 1024 || p->force_block_sweep) return false;
    const ps_core_plan &pl = p->cplan;
    if (!pl.has_events || (!mu && !hr)) return true;
    const uint32_t qe = sweep_queue_entries(pl, std::min(1024u, (uint32_t)p->cfg.pop_size), hr);
    return qe != 0u && 4u * ps_wave_lds(qe) <= p->lds_limit;
}

// geometry of the block sweep for this population and these rates; false if its queues cannot
// be sized safely (then the inline block kernel, correct for any rates, is used)
static bool block_sweep_geometry(const ps_population *p, bool ga, bool mu, bool hr, core_block_geom *g,
                                 uint32_t *lds, uint32_t *waves)
{
    const ps_core_plan &pl = p->cplan;
    const bool events = pl.has_events && (mu || hr);
    if (events && (pl.cshift > 1u || pl.k > 1u)) return false;       // (the plans of the wave sweep: sweep_queue_entries)
    g->segs = (p->cpr + 63u) / 64u;
    auto up16 = [](double x) { return (uint32_t)(((uint64_t)std::ceil(x) + 15u) & ~15ull); };
    const double hr_frac = (events && hr) ? (double)(pl.T[6] - pl.T[2]) / 4294967296.0 * (double)pl.R / 64.0 : 0.0;
    // a wave works on batches of SB segments (4, or 2 when the queues of 4 do not fit beside the
    // rows); the workgroup is as small as a row allows (4, 8 or 16 waves), so that several
    // workgroups per CU overlap their load / compute / store phases
    for (uint32_t SB = (p->block_batch == 2 ? 2u : 4u); SB >= 2; SB >>= 1) {
        uint32_t nw = 4;
        while (nw < 16u && nw * SB < g->segs) nw *= 2u;
        if (p->block_waves) nw = p->block_waves;
        *waves = nw;
        g->SB = SB;
        // (only the residual cells are queued -- R / 64 of the cells; a full queue only sends the row group to the queue-free redo)
        const double mq = events ? SB * 1024.0 * (double)pl.R / 64.0 : 0.0;
        g->QW = std::max(64u, up16(mq + 10.0 * std::sqrt(mq)));
        for (uint32_t R = std::max(1u, nw * SB / std::max(1u, g->segs)); R >= 1; R >>= 1) {
            const uint32_t batches_per_wave = (R * g->segs + nw * SB - 1u) / (nw * SB);
            const double mh = (double)batches_per_wave * SB * 1024.0 * hr_frac;
            g->R = R;
            g->HW = std::max(16u, up16(mh + 10.0 * std::sqrt(mh) + 16.0));
            // queues and HR lists of the waves; the same memory holds the 16-bit HR mask per chunk when a full queue makes
            // the workgroup redo a row group by the queue-free method; then the workgroup's overflow word
            // (at N = 65536 two 64 KB rows + 16 x 2 KB of queues are exactly the CU's 160 KB: the overflow word takes the last
            // entry of the last wave's HR list -- the kernel uses HW - 1 entries per list -- not 16 bytes of its own)
            const uint32_t rows_bytes = (ga ? 2u : 1u) * R * p->pitch;
            const uint32_t scratch = (std::max(nw * (g->QW + 2u * g->HW) * 4u, R * p->cpr * 2u + 16u) + 15u) & ~15u;
            g->ovf_off = rows_bytes + scratch - 4u;
            *lds = rows_bytes + scratch;
            if (*lds <= p->lds_limit) return true;
            if (R == 1) break;
        }
    }
    return false;
}

// PRE variant of the block sweep: one batch per wave and row group, 16-bit parent indices
static bool block_sweep_preload(const ps_population *p, const core_block_geom &g, uint32_t nw)
{
    return p->pitch <= 65536u && g.R * g.segs <= nw * g.SB && !p->no_block_preload;
}

template <uint32_t SB, bool PRE, bool GA, bool MU, bool HR>
static int launch_block_kernel(const core_sweep_args &a, const core_block_geom &g, uint32_t lds, uint32_t nw, hipStream_t st)
{
    auto kern = core_sweep_block_kernel<SB, PRE, GA, MU, HR>;
    if (lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t groups = (a.rows + g.R - 1) / g.R;
    int bpc = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void *)kern, (int)(64u * nw), lds));
    bpc = std::max(1, bpc);
    const uint32_t grid = std::max(1u, std::min(groups, 256u * (uint32_t)bpc));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64u * nw), lds, st, a, g);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

template <bool GA, bool MU, bool HR>
static int launch_core_sweep_block(ps_population *p, const core_sweep_args &a, const uint32_t *d_idx, hipStream_t st)
{
    core_block_geom g{};
    uint32_t lds = 0, nw = 0;
    if (!p->force_inline_sweep && block_sweep_geometry(p, GA, MU, HR, &g, &lds, &nw)) {
        const bool pre = GA && block_sweep_preload(p, g, nw);
        if (GA) {
            const uint32_t n = (pre ? 8u : 16u) * p->cpr;
            if (pre) idx_pack16_kernel<<<(n + 255) / 256, 256, 0, st>>>(d_idx, p->d_idxT, a.N, p->cpr);
            else idx_transpose_kernel<<<(n + 255) / 256, 256, 0, st>>>(d_idx, p->d_idxT, a.N, p->cpr);
        }
        if (GA && pre)
            return g.SB == 4u ? launch_block_kernel<4, GA, GA, MU, HR>(a, g, lds, nw, st)
                              : launch_block_kernel<2, GA, GA, MU, HR>(a, g, lds, nw, st);
        return g.SB == 4u ? launch_block_kernel<4, false, GA, MU, HR>(a, g, lds, nw, st)
                          : launch_block_kernel<2, false, GA, MU, HR>(a, g, lds, nw, st);
    }
    // inline kernel: every candidate handled by its owner lane, no queues to overflow
    const uint32_t block = 1024u;
    const uint32_t hrm_bytes = (a.cpr * 2u + 15u) & ~15u;
    lds = 2u * a.pitch + hrm_bytes;
    if (lds > p->lds_limit)
        return ps_fail(PS_ERR_INVALID, "pop_size %u needs %u bytes of LDS per row (limit %u)", a.N, lds,
                       p->lds_limit);
    auto kern = core_sweep_inline_kernel<GA, MU, HR>;
    if (lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t cap = 256u * (lds > 80 * 1024 ? 1u : 2u);
    const uint32_t grid = std::max(1u, std::min(a.rows, cap));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, st, a);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

// window sweep (core_kernels.h): N > 1024, fused gather + mutate (+ HR), parents in ascending order, out of place
static uint32_t window_sweep_lds(const ps_core_plan &pl)
{
    const uint32_t qe = sweep_queue_entries(pl, 1024u, false);
    return qe ? 4u * ps_window_lds(qe) : 0u;
}

template <bool HR>
static int launch_core_sweep_window(ps_population *p, const core_sweep_args &a, hipStream_t st)
{
    const uint32_t lds = window_sweep_lds(a.plan);
    // ("window_blocks_per_cu": a donor-sharded run leaves the exchange's kernels room beside the sweep)
    const uint32_t bpc = std::max(1u, std::min(p->window_blocks_per_cu ? p->window_blocks_per_cu : (uint32_t)PS_WBPC, p->lds_limit / lds));
    const uint32_t segs = (a.N + 1023u) / 1024u;
    // at least one wave per (XCD group, segment); a multiple of the 8 groups
    const uint32_t grid = (std::max(8u * ((segs + 3u) / 4u), (256u - 8u * p->free_cus_per_xcd) * bpc) + 7u) & ~7u;
    // Two launches ON THE SAME STREAM, the first before the second: the segments whose parent window fits the row buffer,
    // then the others (usually none: its waves leave at once, told by the flag of this generation's parity that only the
    // first launch sets and that the first launch of the generation before cleared -- core_kernels.h).  They alternate the
    // counter sets like any two consecutive launches (a parity of their own among the launches that use them).
    core_sweep_args a0 = a, b = a;
    a0.qcap = b.qcap = sweep_queue_entries(a.plan, 1024u, false);
    a0.launch_parity = (uint32_t)(p->window_launches++ & 1u);
    b.launch_parity = (uint32_t)(p->window_launches++ & 1u);
#define PS_WLAUNCH(NT_)                                                                                                    \
    {                                                                                                                     \
        hipLaunchKernelGGL((core_sweep_window_kernel<true, HR, NT_, false>), dim3(grid), dim3(256), lds, st, a0);          \
        hipLaunchKernelGGL((core_sweep_window_kernel<true, HR, NT_, true>), dim3(grid), dim3(256), lds, st, b);            \
    }
    if (a.nt) PS_WLAUNCH(true) else PS_WLAUNCH(false)
#undef PS_WLAUNCH
    HIPCHK(hipGetLastError());
    return PS_OK;
}

static int launch_core_sweep(ps_population *p, const uint32_t *d_idx, uint32_t gen, bool ga, bool mu,
                             bool hr, hipStream_t st, bool parents_sorted = false)
{
    if (p->cfg.ncols == 0) return PS_OK;
    core_sweep_args a;
    a.state = p->state;
    a.idx = d_idx;
    a.N = (uint32_t)p->cfg.pop_size;
    a.pitch = p->pitch;
    a.cpr = p->cpr;
    a.rows = (uint32_t)p->cfg.ncols;
    a.site_offset = (uint32_t)p->cfg.col_offset;
    a.gen = gen;
    a.k0 = (uint32_t)p->cfg.seed;
    a.k1 = (uint32_t)(p->cfg.seed >> 32);
    a.plan = p->cplan;
    // the kernels' mutate / HR variants assume a plan with events (straight-line per-row code)
    if (!a.plan.has_events) mu = hr = false;
    if (!mu && !hr) a.plan.has_events = 0;
    if (!ga && !mu && !hr) return PS_OK;
    const bool wave = wave_sweep_eligible(p, mu, hr);
    // Out of place: the new generation goes to the second buffer and the two swap roles (an out-of-place stream has
    // a higher ceiling on this GPU than an in-place one, scripts/ubench/inplace_stream.hip); the inline fallback
    // kernel and a failed allocation of the second buffer keep the in-place form -- results are identical.
    a.out = a.state;
    a.nt = 0;
    core_block_geom g_probe{};
    uint32_t lds_probe = 0, nw_probe = 0;
    const bool inline_form = !wave && (p->force_inline_sweep || !block_sweep_geometry(p, ga, mu, hr, &g_probe, &lds_probe, &nw_probe));
    // window sweep: the wave-per-row design for N > 1024 -- needs the children in ascending parent order (ps_sim's
    // generations), the fused gather + mutate (+ HR) step with events, its queue sized like the wave sweep's, a second
    // buffer (it is out of place by construction) and 4 x 5.2 KB of LDS
    const uint32_t wlds = window_sweep_lds(a.plan);
    bool window = !wave && parents_sorted && ga && mu && a.plan.has_events && p->window_sweep != 0 && p->pitch > 1024
                  && a.N <= (1u << 22)
                  && !p->force_inline_sweep && !p->force_block_sweep
                  && wlds != 0u && wlds <= p->lds_limit;
    if (window && !p->state2 && hipMalloc(&p->state2, (uint64_t)p->cfg.ncols * p->pitch) != hipSuccess) {
        (void)hipGetLastError();
        p->state2 = nullptr;
        window = false;          // no room for the second buffer (cfg4 on a full GPU): the in-place block sweep
    }
    const int oop = window ? (p->sweep_oop == 1 ? 1 : 2) : p->sweep_oop < 0 ? (wave ? 2 : 0) : p->sweep_oop;
    if (oop && (!inline_form || window)) {
        if (!p->state2 && hipMalloc(&p->state2, (uint64_t)p->cfg.ncols * p->pitch) != hipSuccess) {
            (void)hipGetLastError();
            p->state2 = nullptr;
            p->sweep_oop = 0;
        }
        if (p->state2) {
            a.out = p->state2;
            a.nt = oop == 2 ? 1u : 0u;
            std::swap(p->state, p->state2);      // (a.state / a.out hold this launch's roles)
        }
    }
    a.overflow_flag = p->d_flag;
    a.qcap = 0;
    a.qcap_limit = p->sweep_queue_cap;
    a.stamps = p->d_stamps;
    a.work_ctr = p->d_work;
    a.wide_flags = p->d_work + p->work_flags_ofs;
    a.launch_parity = 0;
    a.idxT = p->d_idxT;
    p->last_sweep_form = window ? PS_SWEEP_FORM_WINDOW
                         : wave ? PS_SWEEP_FORM_WAVE
                         : inline_form ? PS_SWEEP_FORM_INLINE : PS_SWEEP_FORM_BLOCK;
    if (window) return hr ? launch_core_sweep_window<true>(p, a, st) : launch_core_sweep_window<false>(p, a, st);
#define PS_DISPATCH(G_, M_, H_)                                              \
    if (ga == G_ && mu == M_ && hr == H_)                                    \
        return wave ? launch_core_sweep_wave<G_, M_, H_>(p, a, st)           \
                    : launch_core_sweep_block<G_, M_, H_>(p, a, d_idx, st);
    PS_DISPATCH(true, false, false)
    PS_DISPATCH(false, true, false)
    PS_DISPATCH(false, false, true)
    PS_DISPATCH(true, true, false)
    PS_DISPATCH(true, true, true)
#undef PS_DISPATCH
    return ps_fail(PS_ERR_INVALID, "unsupported operator combination");
}

// ---------------------------------------------------------------------------
// accessory step launches
// ---------------------------------------------------------------------------
// want_snapshot: the step also leaves the pre-recombination snapshot (and the zeroed work counter) of a light-form HGT that
// follows on the same stream -- ps_sim's generations: no copy kernel and no memset between the two
static int launch_acc_step(ps_population *p, const uint32_t *d_idx, uint32_t gen, bool ga, bool mu,
                           hipStream_t st, uint32_t *idx_out = nullptr, bool want_snapshot = false)
{
    if (p->d.G == 0) return PS_OK;
    acc_step_args a;
    a.srcI = p->I[p->cur];
    a.dstI = p->I[1 - p->cur];
    a.idx = d_idx;
    a.idx_out = idx_out;
    a.snapI = nullptr;
    a.zero_word = nullptr;
    p->snap_valid = false;
    if (want_snapshot) {
        if (!p->I_snap && hipMalloc(&p->I_snap, (uint64_t)p->d.N * p->d.GW * 8) != hipSuccess) {
            (void)hipGetLastError();
            p->I_snap = nullptr;
        }
        if (p->I_snap) {
            a.snapI = p->I_snap;
            a.zero_word = p->cnt + (uint64_t)PS_MAX_COMP * p->d.N;
            p->snap_valid = true;
        }
    }
    a.d = p->d;
    a.gen = gen;
    a.k0 = (uint32_t)p->cfg.seed;
    a.k1 = (uint32_t)(p->cfg.seed >> 32);
    a.plan = p->aplan;
    const uint64_t threads = (uint64_t)p->d.N * p->d.GW;
    const uint32_t grid = (uint32_t)((threads + 255) / 256);
    if (ga && mu) acc_step_rows_kernel<true, true><<<grid, 256, 0, st>>>(a);
    else if (ga) acc_step_rows_kernel<true, false><<<grid, 256, 0, st>>>(a);
    else acc_step_rows_kernel<false, true><<<grid, 256, 0, st>>>(a);
    HIPCHK(hipGetLastError());
    p->cur = 1 - p->cur;
    p->g_valid = false;
    p->counts_fresh = false;
    p->edit_epoch++;
    return PS_OK;
}

// the gene-major view (one ballot per gene word) is only needed for gene frequencies: rebuilt on demand
static int ensure_gene_major(ps_population *p, hipStream_t st)
{
    if (p->g_valid || p->d.G == 0) return PS_OK;
    acc_i_to_g_kernel<<<dim3(p->d.W, p->d.GW), 64, 0, st>>>(p->I[p->cur], p->G[0], p->d);
    HIPCHK(hipGetLastError());
    p->g_valid = true;
    return PS_OK;
}

// recipient partitions of the binned HGT: the rows of one partition (plus 1 KB of static LDS of the
// apply kernel) must fit the LDS of a workgroup
static uint32_t hgt_partitions(const ps_population *p)
{
    const uint64_t row_bytes = (uint64_t)p->d.GW * 8;
    if (row_bytes == 0 || p->lds_limit < 1024u + row_bytes) return 0;
    const uint64_t part_cap = (p->lds_limit - 1024u) / row_bytes;
    return (uint32_t)((p->d.N + part_cap - 1) / part_cap);
}

// which form launch_acc_hgt will take for this handle's rates (its callers decide from it whether the step before the HGT
// should leave the light form's snapshot: the binned form never reads one)
static bool hgt_takes_binned_form(const ps_population *p)
{
    double expected = 0.0;
    uint32_t max_comp = 1;
    for (int c = 0; c < p->aplan.n_comp; c++) {
        if (p->aplan.lam_rec[c] == 0.0 || p->ptab_len[c] == 0) continue;
        expected += (double)p->d.N * p->aplan.lam_rec[c];
        max_comp = std::max(max_comp, p->aplan.comp_end[c] - p->aplan.comp_begin[c]);
    }
    const uint32_t list_lds = ((max_comp * 2u + 15u) & ~15u);
    const uint32_t parts = hgt_partitions(p);
    const uint32_t rpp = parts ? (p->d.N + parts - 1) / parts : 0;
    return parts >= 1 && parts <= 1024 && rpp >= 2 && p->d.G <= 65536u && (uint64_t)p->d.N * rpp < (1ull << 32)
           && (p->hgt_mode == 2 || (p->hgt_mode == 0 && expected >= 1.0e7))
           && list_lds + parts * 4u + 64u <= p->lds_limit;
}

// wait_before_apply / record_after_apply: the turn-taking of a heavy HGT with the core sweep (ps_sim) -- the LDS-image pass
// waits for the previous sweep, and the next sweep waits for it: not for the reduce pass behind it, a small streaming
// kernel that fits beside a sweep (nor, in a donor-sharded run, for the exchange and the merge that follow)
// st_gap / ev_bin (round 5, ps_sim's turn-taking schedule): the passes that sit BETWEEN two sweeps -- the LDS-image pass, and in
// a donor-sharded run the reduce pass, the exchange and the merge as well -- are launched on the sweep's own stream, in order
// with the sweeps, instead of being tied to them by two cross-stream events (a stream that waits for another stream's event
// starts 20 - 50 us after it fired: twice per generation, 22 + 6 of cfg3's 92 us between sweeps, 54 + 7 of 577 at one rank of 8
// at cfg4).  Only the bin pass, which runs beside the sweep, stays on `st`; st_gap waits for it (ev_bin), and `st` waits for
// record_after_apply, recorded on st_gap behind the last of them.
static int launch_acc_hgt(ps_population *p, uint32_t gen, hipStream_t st, hipEvent_t wait_before_apply = nullptr,
                          hipEvent_t record_after_apply = nullptr, hipStream_t st_gap = nullptr, hipEvent_t ev_bin = nullptr)
{
    if (p->d.G == 0 || p->d.N < 2) return PS_OK;
    p->stream_of_call = st;          // (the stream the accessory chain continues on, whatever stream the gap passes take)
    bool counts_left = false;
    acc_hgt_args a{};
    a.n_comp = (uint32_t)p->aplan.n_comp;
    double expected = 0.0;
    uint32_t max_comp = 1;
    for (int c = 0; c < p->aplan.n_comp; c++) {
        a.gb[c] = p->aplan.comp_begin[c];
        a.ge[c] = p->aplan.comp_end[c];
        a.ptab[c] = nullptr;
        if (p->aplan.lam_rec[c] == 0.0 || p->ptab_len[c] == 0) continue;      // population.rs:558
        a.ptab[c] = p->d_ptab[c];
        a.kmin[c] = p->ptab_kmin[c];
        a.plen[c] = p->ptab_len[c];
        expected += (double)p->d.N * p->aplan.lam_rec[c];
        max_comp = std::max(max_comp, a.ge[c] - a.gb[c]);
    }
    if (expected == 0.0) return PS_OK;
    a.d = p->d;
    a.gen = gen;
    a.k0 = (uint32_t)p->cfg.seed;
    a.k1 = (uint32_t)(p->cfg.seed >> 32);
    a.work_ctr = p->cnt + (uint64_t)PS_MAX_COMP * p->d.N;
    a.overflow_flag = p->d_flag;
    const bool sharded = p->donor_cnt != 0;
    a.dn_lo = sharded ? p->donor_lo : 0u;
    a.dn_cnt = sharded ? p->donor_cnt : p->d.N;
    const uint32_t items = a.n_comp * a.dn_cnt;           // work items of THIS launch: (compartment, own donor)
    const uint64_t mat_words = (uint64_t)p->d.N * p->d.GW;
    if (sharded && !p->d_delta) {
        p->delta_words = (mat_words + 4095) & ~4095ull;
        HIPCHK(hipMalloc(&p->d_delta, p->delta_words * 8));
        HIPCHK(hipMemsetAsync(p->d_delta, 0, p->delta_words * 8, st));
    }
    const uint32_t list_lds = ((max_comp * 2u + 15u) & ~15u);
    if (list_lds > p->lds_limit)
        return ps_fail(PS_ERR_INVALID, "a compartment of %u genes needs %u bytes of LDS (limit %u)", max_comp, list_lds, p->lds_limit);

    // Heavy HGT (>= 1e7 expected events per generation: cfg3, and every generation at the cfg4 / cfg5
    // populations): two passes without global atomics -- the donors' events are binned by recipient
    // partition, then ORed into LDS images of the partitions and reduced into the matrix.  Otherwise
    // one 64-bit atomicOr per event; that form needs no static LDS and co-runs with the sweep.
    const uint64_t row_bytes = (uint64_t)p->d.GW * 8;
    const uint32_t parts = hgt_partitions(p);
    const bool binned = hgt_takes_binned_form(p);
    // event counts per donor; the light form's snapshot copy rides along (donors read the
    // pre-recombination matrix, population.rs:693-695, while recipients are edited in place)
    // (both forms draw the count of an item where they serve the item; the light form's snapshot and zeroed work counter
    // come from the step before it when ps_sim asked for them, else from a copy kernel and a memset here)
    const bool have_snapshot = p->snap_valid;
    p->snap_valid = false;                      // (the matrix is about to change)
    if (!binned && !have_snapshot) {
        HIPCHK(hipMemsetAsync(a.work_ctr, 0, sizeof(uint32_t), st));
        acc_snapshot_kernel<<<64, 256, 0, st>>>(p->I[p->cur], p->I[1 - p->cur], mat_words);
    }
    if (binned) {
        const uint32_t rows_per_part = (p->d.N + parts - 1) / parts;
        const uint32_t lds = (uint32_t)(rows_per_part * row_bytes);
        // (the apply pass reads one bin per (donor workgroup, partition): with 2048 donor workgroups the bins of a donor shard
        // held ~60 events and the pass was a chain of small dependent reads -- one rank of 8 at cfg4: 4.78 -> 4.63 ms per
        // generation at 1024, cfg5pop 4.22 -> 4.13, cfg4 33.66 -> 33.45, cfg3 unchanged; 512: cfg3 0.68 against 0.64)
        uint32_t donor_blocks = std::min(items, 1024u);
        if (const char *e = getenv("PANSIM_HGT_DONOR_BLOCKS")) donor_blocks = std::max(1u, std::min(items, (uint32_t)atoi(e)));
        // ~1024 apply workgroups, at most 64 slice images (measured: cfg3 64 of 32/64/128, cfg4 5 of 2/3/5)
        // (a donor shard has 1 / K of the events for the same images: fewer slices mean less to zero, publish and reduce --
        // one rank of 8 at cfg4, with a wave per bin in the LDS-image pass: 5 / 3 / 2 / 1 slices 0.43 / 0.41 / 0.40 / 0.40 ms exposed)
        const uint32_t base_slices = std::min(64u, (1024u + parts - 1) / parts);
        const uint32_t auto_slices = sharded ? (uint32_t)(((uint64_t)base_slices * a.dn_cnt + p->d.N - 1) / p->d.N) : base_slices;
        const uint32_t n_slices = std::max(1u, p->hgt_slices ? p->hgt_slices : auto_slices);
        const uint64_t words = (uint64_t)p->d.N * p->d.GW;
        const uint64_t img_bytes = (uint64_t)n_slices * words * 8;
        // events of a donor workgroup: Poisson with mean <= sum_c lambda_c * ceil(N / blocks); a partition
        // receives rows_per_part / (N - 1) of them
        double per_block = 0.0;
        for (int c = 0; c < p->aplan.n_comp; c++)
            if (a.ptab[c]) per_block += p->aplan.lam_rec[c] * (double)((a.dn_cnt + donor_blocks - 1) / donor_blocks);
        const double mean = per_block * (double)rows_per_part / (double)(p->d.N - 1);
        uint32_t cap = (uint32_t)(((uint64_t)(mean + 10.0 * std::sqrt(mean) + 64.0) + 63) & ~63ull);
        if (p->hgt_bin_cap) cap = std::min(cap, std::max(1u, p->hgt_bin_cap));
        const uint64_t bin_words = (uint64_t)donor_blocks * parts * cap;
        const uint64_t cnt_bytes = ((uint64_t)donor_blocks * parts * 4 + 255) & ~255ull;
        // (measured at the cfg4 population: the bin pass running beside the block sweep this way makes the generation
        // LONGER -- 8.3 -> 9.4 ms per generation at 1/8 of the sites -- so ps_sim leaves it off: test hook only)
        const uint64_t list_bytes = p->hgt_bin_list_in_global ? (((uint64_t)donor_blocks * max_comp * sizeof(uint16_t) + 255) & ~255ull) : 0;
        const uint64_t need = img_bytes + bin_words * 4 + cnt_bytes + list_bytes;
        if (p->hgt_scratch_cap < need) {
            if (p->hgt_scratch) HIPCHK(hipFree(p->hgt_scratch));
            p->hgt_scratch = nullptr;
            p->hgt_scratch_cap = 0;
            HIPCHK(hipMalloc(&p->hgt_scratch, need));
            p->hgt_scratch_cap = need;
        }
        if (!p->hgt_ovf_img) {
            // the overflow image of the bin pass: zero now, and left zero by every reduce pass
            HIPCHK(hipMalloc(&p->hgt_ovf_img, words * 8));
            HIPCHK(hipMemsetAsync(p->hgt_ovf_img, 0, words * 8, st));
        }
        a.ovf_img = (unsigned long long *)p->hgt_ovf_img;
        a.scratch = (uint32_t *)p->hgt_scratch;
        a.bins = (uint32_t *)((uint8_t *)p->hgt_scratch + img_bytes);
        a.counts = (uint32_t *)((uint8_t *)p->hgt_scratch + img_bytes + bin_words * 4);
        a.parts = parts;
        a.rows_per_part = rows_per_part;
        a.part_magic = (uint32_t)(4294967296ull / rows_per_part) + 1u;
        a.bin_cap = cap;
        a.srcI = p->I[p->cur];          // not edited before the reduce pass: it IS the snapshot
        a.dstI = p->I[p->cur];
        a.bin_prio = p->hgt_bin_prio;
        if (list_bytes) {
            // the co-running block sweep owns the CU's LDS: donor lists in global scratch, the bin pass then fits
            // beside it with its fill counters only
            a.list_scratch = (uint16_t *)((uint8_t *)p->hgt_scratch + img_bytes + bin_words * 4 + cnt_bytes);
            a.list_stride = max_comp;
        }
        const uint32_t dlds = ((parts + 3u) & ~3u) * 4u + (list_bytes ? 0u : list_lds);
        if (dlds > 64 * 1024)
            HIPCHK(hipFuncSetAttribute((const void *)acc_hgt_donor_bin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
        hipLaunchKernelGGL(acc_hgt_donor_bin_kernel, dim3(donor_blocks), dim3(256), dlds, st, a);
        const bool in_gap = st_gap && ev_bin && record_after_apply && !p->exchange_beside_sweep;
        hipStream_t run = st;            // where the passes between two sweeps go
        if (in_gap) {
            HIPCHK(hipEventRecord(ev_bin, st));
            HIPCHK(hipStreamWaitEvent(st_gap, ev_bin, 0));
            run = st_gap;                // (in order behind the previous sweep: no wait_before_apply)
        } else if (wait_before_apply) HIPCHK(hipStreamWaitEvent(st, wait_before_apply, 0));
        auto kern = acc_hgt_apply_kernel;
        if (lds > 64 * 1024)
            HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(parts * n_slices), dim3(p->hgt_apply_threads), lds, run, a, donor_blocks, n_slices);
        // (a donor-sharded run keeps the sweep behind the exchange and the merge as well: the kernels of an exchange do not
        // fit beside the sweep's 7 workgroups per CU -- RCCL's need 264 registers per lane -- they would wait for its END, and
        // with them the whole chain of the next generation)
        if (record_after_apply && (!sharded || p->exchange_beside_sweep)) {
            HIPCHK(hipEventRecord(record_after_apply, run));
            if (in_gap) HIPCHK(hipStreamWaitEvent(st, record_after_apply, 0));      // the reduce pass, beside the next sweep, on st
            record_after_apply = nullptr;
            run = st;
        }
        st = run;                        // (sharded: reduce, exchange and merge follow the LDS-image pass on its stream)
        if (!sharded && p->fuse_counts_out) {
            // (ps_sim, neutral selection: the reduce pass also leaves the gene counts the next generation's host half reads)
            acc_hgt_reduce_rows_kernel<<<(uint32_t)((p->d.N + 3) / 4), 256, 0, st>>>((const uint64_t *)p->hgt_scratch, p->I[p->cur], p->d,
                                                                                    n_slices, (unsigned long long *)p->hgt_ovf_img,
                                                                                    p->fuse_counts_out, p->fuse_logw_out);
            counts_left = true;
        } else
        acc_hgt_reduce_kernel<<<(uint32_t)((words + 255) / 256), 256, 0, st>>>((const uint64_t *)p->hgt_scratch,
                                                                              sharded ? p->d_delta : p->I[p->cur], words, n_slices,
                                                                              sharded ? 1 : 0, (unsigned long long *)p->hgt_ovf_img);
    } else {
        a.srcI = have_snapshot ? p->I_snap : p->I[1 - p->cur];          // the snapshot (left by the step, or copied above)
        a.dstI = sharded ? p->d_delta : p->I[p->cur];
        if (sharded) HIPCHK(hipMemsetAsync(p->d_delta, 0, mat_words * 8, st));
        // beside a long core sweep (ps_sim sets hgt_events_per_thread) the kernel is launched narrow --
        // one-wave workgroups, a fixed number of events per thread over the generation: the same events
        // then disturb the sweep for longer but far less; stand-alone calls use the whole chip
        uint32_t grid = std::min(items, 256u * 32u);
        if (p->hgt_events_per_thread) {
            const double expected_here = expected * (double)a.dn_cnt / (double)p->d.N;
            const uint64_t want = (uint64_t)(expected_here / ((double)p->hgt_events_per_thread * 64.0)) + 1;
            grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(want, 1), grid);
        }
        uint32_t dyn_lds = list_lds;
        if (p->hgt_list_in_global) {
            // the co-running block sweep owns the CU's LDS: keep the donor lists in a per-workgroup
            // global scratch instead (hot in L2; reads of a list come from the workgroup that wrote it)
            const uint64_t need = (uint64_t)grid * max_comp * sizeof(uint16_t);
            if (p->hgt_scratch_cap < need) {
                if (p->hgt_scratch) HIPCHK(hipFree(p->hgt_scratch));
                p->hgt_scratch = nullptr;
                p->hgt_scratch_cap = 0;
                HIPCHK(hipMalloc(&p->hgt_scratch, need));
                p->hgt_scratch_cap = need;
            }
            a.list_scratch = (uint16_t *)p->hgt_scratch;
            a.list_stride = max_comp;
            dyn_lds = 0;
        }
        if (dyn_lds > 64 * 1024)
            HIPCHK(hipFuncSetAttribute((const void *)acc_hgt_donor_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
        hipLaunchKernelGGL(acc_hgt_donor_wave_kernel, dim3(grid), dim3(64), dyn_lds, st, a);
    }
    HIPCHK(hipGetLastError());
    if (sharded) {
        // the exchange step of the path: OR of the shards' deltas (in place in d_delta, ordered on st), then
        // the union goes into the matrix.  Without a hook the own donors' events alone are applied.
        if (p->exchange) {
            const std::string prev = g_err;
            g_err.clear();
            const int rc = p->exchange(p->exchange_ctx, p->d_delta, p->delta_words, (void *)st);
            if (rc != 0) {
                const std::string inner = g_err;
                return ps_fail(PS_ERR_STATE, "the HGT delta exchange failed (%d)%s%s", rc, inner.empty() ? "" : ": ", inner.c_str());
            }
            g_err = prev;
        }
        if (p->fuse_counts_out) {
            // (ps_sim, neutral selection: the merge also leaves the gene counts the next generation's host half reads)
            acc_or_rows_kernel<<<(uint32_t)((p->d.N + 3) / 4), 256, 0, st>>>(p->I[p->cur], p->d_delta, p->d, p->fuse_counts_out, p->fuse_logw_out);
            counts_left = true;
        } else
        acc_or_kernel<<<(uint32_t)((mat_words + 255) / 256), 256, 0, st>>>(p->I[p->cur], p->d_delta, mat_words);
        HIPCHK(hipGetLastError());
    }
    if (record_after_apply) {        // (light form under a forced turn-taking schedule; sharded runs)
        HIPCHK(hipEventRecord(record_after_apply, st));
        if (st != p->stream_of_call) HIPCHK(hipStreamWaitEvent(p->stream_of_call, record_after_apply, 0));     // (the gap stream carried the merge)
    }
    p->g_valid = false;       // only the individual-major view is edited (ensure_gene_major)
    p->counts_fresh = counts_left;
    p->edit_epoch++;
    return PS_OK;
}

extern "C" int ps_set_donor_shard(ps_population *p, uint32_t shard_rank, uint32_t shard_count, ps_exchange_fn fn, void *ctx)
{
    if (!p) return ps_fail(PS_ERR_INVALID, "null handle");
    if (p->cfg.core) return ps_fail(PS_ERR_INVALID, "donor shards apply to the accessory matrix (HGT, population.rs:544-751)");
    if (shard_count < 1 || shard_rank >= shard_count) return ps_fail(PS_ERR_INVALID, "bad donor shard %u of %u", shard_rank, shard_count);
    const uint64_t N = p->cfg.pop_size;
    if (shard_count == 1) {
        p->donor_lo = 0;
        p->donor_cnt = 0;
        p->exchange = nullptr;
        p->exchange_ctx = nullptr;
        return PS_OK;
    }
    const uint64_t lo = N * shard_rank / shard_count, hi = N * ((uint64_t)shard_rank + 1) / shard_count;
    if (hi <= lo) return ps_fail(PS_ERR_INVALID, "more donor shards than individuals");
    p->donor_lo = (uint32_t)lo;
    p->donor_cnt = (uint32_t)(hi - lo);
    p->exchange = fn;
    p->exchange_ctx = ctx;
    return PS_OK;
}

static int upload_idx(ps_population *p, const uint32_t *sample)
{
    const uint64_t N = p->cfg.pop_size;
    for (uint64_t i = 0; i < N; i++)
        if (sample[i] >= N) return ps_fail(PS_ERR_INVALID, "parent index %u out of range", sample[i]);
    HIPCHK(hipMemcpyAsync(p->d_idx, sample, N * sizeof(uint32_t), hipMemcpyHostToDevice, p->stream));
    return PS_OK;
}

static int step_device(ps_population *p, const uint32_t *d_idx, uint32_t gen, bool ga, bool mu, bool hr,
                       hipStream_t st, uint32_t *idx_out = nullptr, bool parents_sorted = false)
{
    if (p->cfg.core) return launch_core_sweep(p, d_idx, gen, ga, mu, hr, st, parents_sorted);
    if (ga || mu) PSCHK(launch_acc_step(p, d_idx, gen, ga, mu, st, idx_out));
    if (hr) PSCHK(launch_acc_hgt(p, gen, st));
    return PS_OK;
}

extern "C" int ps_next_generation(ps_population *p, const uint32_t *sample)
{
    if (!p || !sample) return ps_fail(PS_ERR_INVALID, "null argument");
    PSCHK(use_device(p));
    PSCHK(upload_idx(p, sample));
    PSCHK(step_device(p, p->d_idx, 0, true, false, false, p->stream));
    p->rows_overridden = true;
    return sync_checked(p);
}

extern "C" int ps_mutate_alleles(ps_population *p, uint32_t generation)
{
    if (!p) return ps_fail(PS_ERR_INVALID, "null handle");
    if (!p->rates_set) return ps_fail(PS_ERR_STATE, "ps_set_rates has not been called");
    PSCHK(use_device(p));
    PSCHK(step_device(p, nullptr, generation, false, true, false, p->stream));
    return sync_checked(p);
}

extern "C" int ps_recombine(ps_population *p, uint32_t generation)
{
    if (!p) return ps_fail(PS_ERR_INVALID, "null handle");
    if (!p->rates_set) return ps_fail(PS_ERR_STATE, "ps_set_rates has not been called");
    PSCHK(use_device(p));
    PSCHK(step_device(p, nullptr, generation, false, false, true, p->stream));
    return sync_checked(p);
}

extern "C" int ps_step(ps_population *p, uint32_t generation, const uint32_t *sample, int do_recombine)
{
    if (!p || !sample) return ps_fail(PS_ERR_INVALID, "null argument");
    if (!p->rates_set) return ps_fail(PS_ERR_STATE, "ps_set_rates has not been called");
    PSCHK(use_device(p));
    PSCHK(upload_idx(p, sample));
    // an ascending sample lets a wide core population take the window sweep (what ps_sim_run's own generations do)
    const bool sorted = p->cfg.core && std::is_sorted(sample, sample + p->cfg.pop_size);
    PSCHK(step_device(p, p->d_idx, generation, true, true, do_recombine != 0, p->stream, nullptr, sorted));
    p->rows_overridden = true;
    return sync_checked(p);
}

// ---------------------------------------------------------------------------
// parent sampling (population.rs:270-448)
// ---------------------------------------------------------------------------
static int fitness_terms_device(ps_population *acc, const double *sel, int32_t *num_genes, double *logw,
                                hipStream_t st)
{
    const uint64_t N = acc->cfg.pop_size, G = acc->cfg.ncols;
    int need = 0;
    if (G) {
        std::vector<double> l1p(G);
        for (uint64_t g = 0; g < G; g++) {
            l1p[g] = std::log(1.0 + sel[g] * 1.0);           // population.rs:306 with col_val = 1
            if (l1p[g] != 0.0) need = 1;
            if (std::isnan(l1p[g])) need = 1;
        }
        if (need)
            HIPCHK(hipMemcpyAsync(acc->d_log1p, l1p.data(), G * sizeof(double), hipMemcpyHostToDevice, st));
        // l1p must outlive the async copy: pageable memcpy is staged before return
        HIPCHK(hipStreamSynchronize(st));
    }
    acc_fitness_kernel<<<(uint32_t)((N + 255) / 256), 256, 0, st>>>(acc->I[acc->cur], acc->d_log1p, need,
                                                                 acc->d_num_genes, acc->d_logw, acc->d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(num_genes, acc->d_num_genes, N * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(logw, acc->d_logw, N * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return PS_OK;
}

extern "C" int ps_fitness_terms(ps_population *acc, const double *sel, int32_t *num_genes, double *logw)
{
    if (!acc || !num_genes || !logw) return ps_fail(PS_ERR_INVALID, "null argument");
    if (acc->cfg.core) return ps_fail(PS_ERR_INVALID, "sample_indices runs on the accessory matrix (main.rs:442)");
    if (acc->cfg.ncols && !sel) return ps_fail(PS_ERR_INVALID, "null selection coefficients");
    PSCHK(use_device(acc));
    PSCHK(fitness_terms_device(acc, sel, num_genes, logw, acc->stream));
    const uint32_t *slot = nullptr;
    PSCHK(rows_current(acc, &slot));
    rows_permute(num_genes, slot, acc->cfg.pop_size);
    rows_permute(logw, slot, acc->cfg.pop_size);
    return PS_OK;
}

// Host threads for the element-wise libm calls of the softmaxes (the sums stay sequential)
static unsigned host_threads_for(uint64_t n)
{
    if (n < 16384) return 1;
    unsigned hw = std::thread::hardware_concurrency();
    if (const char *e = getenv("PANSIM_HOST_THREADS")) hw = (unsigned)std::max(1, atoi(e));
    return std::max(1u, std::min(std::min(hw, 16u), (unsigned)(n / 8192)));
}
template <typename F>
static void par_for(uint64_t n, F fn)      // fn(begin, end) on contiguous slices
{
    const unsigned T = host_threads_for(n);
    if (T <= 1) { fn((uint64_t)0, n); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++) th.emplace_back(fn, n * t / T, n * (t + 1) / T);
    for (auto &x : th) x.join();
}

// x[i] = exp(x[i] - shift) for all i.  The arguments of the size softmax are (gene count - average) * ln(penalty): a few
// hundred distinct values among N, so exp is looked up in a small direct-mapped table keyed by the argument's bits (the same
// bits give the same result: nothing about the output depends on the table's state); vectors without repeats (selection
// coefficients drawn per gene) keep the threads.
static void exp_all(double *x, uint64_t n, double shift)
{
    constexpr uint32_t TB = 4096;
    if (n >= 2048) {
        static thread_local std::vector<uint64_t> key;
        static thread_local std::vector<double> val;
        key.assign(TB, ~0ull);           // (~0 is a NaN pattern: its exp is computed, never looked up)
        val.resize(TB);
        auto one = [&](double a, uint64_t &miss) -> double {
            uint64_t bits;
            memcpy(&bits, &a, 8);
            const uint32_t h = (uint32_t)((bits * 0x9E3779B97F4A7C15ull) >> 52);
            if (key[h] == bits && bits != ~0ull) return val[h];
            miss++;
            const double e = std::exp(a);
            key[h] = bits;
            val[h] = e;
            return e;
        };
        // a prefix decides: mostly repeats -> the table for everything
        const uint64_t probe = 1024;
        uint64_t miss = 0;
        for (uint64_t i = 0; i < probe; i++) x[i] = one(x[i] - shift, miss);
        if (miss * 4 <= probe) {
            for (uint64_t i = probe; i < n; i++) x[i] = one(x[i] - shift, miss);
            return;
        }
        par_for(n - probe, [&](uint64_t a, uint64_t b) { for (uint64_t i = probe + a; i < probe + b; i++) x[i] = std::exp(x[i] - shift); });
        return;
    }
    par_for(n, [&](uint64_t a, uint64_t b) { for (uint64_t i = a; i < b; i++) x[i] = std::exp(x[i] - shift); });
}

// softmax as population.rs:325-340 / :356-361 / :377-382 write it: lse = ln_sum_exp(v) (logsumexp 0.1, in
// its one-pass streaming form: for every x, `if x <= alpha { r += exp(x - alpha) } else { r *= exp(alpha - x);
// r += 1; alpha = x }`, result ln(r) + alpha), v = exp(v - lse), sum = left-to-right sum of v, v = v / sum
// (0 where v is -inf).  Every floating-point operation below is one of those, on the same operands and in the
// same order for the sequential ones; only the independent exp calls run on several threads, and a vector of
// N identical finite values (neutral selection; no competition) needs one exp instead of 2 N.
static void softmax_norm(double *v, uint64_t n, std::vector<double> &scratch)
{
    if (n == 0) return;
    bool equal = std::isfinite(v[0]);
    for (uint64_t i = 1; equal && i < n; i++) equal = (v[i] == v[0]);
    if (equal) {
        // streaming form on N equal values x: r = 0 * exp(-inf - x) + 1 = 1, then N - 1 times r += exp(0) = 1
        // (the result is a function of (N, x) alone, and under neutral selection without competition two of the three
        // softmaxes of EVERY generation are this one with the same x: the N sequential additions -- 0.08 ms of dependent adds
        // at N = 65536, on the chain beside the sweep -- are done once per (N, x) and thread)
        static thread_local uint64_t memo_n = 0, memo_x = 0;
        static thread_local double memo_q = 0.0;
        uint64_t xbits;
        memcpy(&xbits, &v[0], 8);
        double q;
        if (memo_n == n && memo_x == xbits) {
            q = memo_q;
        } else {
            const double lse = std::log((double)n) + v[0];
            const double e = std::exp(v[0] - lse);
            double sum = 0.0;
            for (uint64_t i = 0; i < n; i++) sum += e;
            q = (e != -INFINITY) ? e / sum : 0.0;
            memo_n = n; memo_x = xbits; memo_q = q;
        }
        for (uint64_t i = 0; i < n; i++) v[i] = q;
        return;
    }
    // exp arguments of the streaming form: they depend on the running maximum only
    scratch.resize(n);
    double *arg = scratch.data();
    static thread_local std::vector<uint8_t> upd;
    upd.resize(n);
    double alpha = -INFINITY;
    for (uint64_t i = 0; i < n; i++) {
        if (v[i] <= alpha) { arg[i] = v[i] - alpha; upd[i] = 0; }
        else { arg[i] = alpha - v[i]; upd[i] = 1; alpha = v[i]; }
    }
    exp_all(arg, n, 0.0);
    double r = 0.0;
    for (uint64_t i = 0; i < n; i++) {
        if (upd[i]) { r *= arg[i]; r += 1.0; }
        else r += arg[i];
    }
    const double lse = std::log(r) + alpha;
    exp_all(v, n, lse);
    double sum = 0.0;
    for (uint64_t i = 0; i < n; i++) sum += v[i];
    for (uint64_t i = 0; i < n; i++) v[i] = (v[i] != -INFINITY) ? v[i] / sum : 0.0;
}

// The genome-size softmax (population.rs:346-361) on its integer structure: v_i = (double)(count_i - average) * ln(penalty) takes
// one value per distinct gene count (a few hundred among N), so every exp and every division of softmax_norm is done once per
// distinct count and looked up by the count itself -- the same libm calls on the same operands, the sequential sums in the same
// order, hence the same doubles as softmax_norm(v) (tests/test_host_logic.py compares with the plain loops of the CPU
// restatement).  The exp arguments of the streaming ln_sum_exp depend on the running maximum: the table is stamped with the
// maximum's epoch and starts over when it moves (expected O(log N) times).
static bool softmax_size(const int32_t *cnt, int32_t avg, double lp, uint64_t n, double *v)
{
    int32_t lo = cnt[0], hi = cnt[0];
    for (uint64_t i = 1; i < n; i++) { lo = std::min(lo, cnt[i]); hi = std::max(hi, cnt[i]); }
    const uint64_t span = (uint64_t)((int64_t)hi - (int64_t)lo) + 1;
    if (lo == hi || span > (1u << 20)) return false;          // (equal vector: softmax_norm's own path; absurd spans: the generic code)
    static thread_local std::vector<double> val, tab;
    static thread_local std::vector<uint32_t> stamp;
    static thread_local std::vector<uint8_t> upd;
    val.resize(span);
    tab.resize(span);
    stamp.assign(span, 0u);
    upd.resize(n);
    for (uint64_t k = 0; k < span; k++) val[k] = (double)((int32_t)((int64_t)lo + (int64_t)k) - avg) * lp;      // :350-355
    // streaming ln_sum_exp: r over exp(v - alpha) / exp(alpha - v), alpha the running maximum
    uint32_t epoch = 1;
    double alpha = -INFINITY, r = 0.0;
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t k = (uint32_t)(cnt[i] - lo);
        const double x = val[k];
        if (x <= alpha) {
            if (stamp[k] != epoch) { tab[k] = std::exp(x - alpha); stamp[k] = epoch; }
            r += tab[k];
        } else {
            r *= std::exp(alpha - x);
            r += 1.0;
            alpha = x;
            epoch++;
        }
    }
    const double lse = std::log(r) + alpha;
    epoch++;
    double sum = 0.0;
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t k = (uint32_t)(cnt[i] - lo);
        if (stamp[k] != epoch) { tab[k] = std::exp(val[k] - lse); stamp[k] = epoch; }
        sum += tab[k];
    }
    for (uint64_t k = 0; k < span; k++)
        if (stamp[k] == epoch) tab[k] = (tab[k] != -INFINITY) ? tab[k] / sum : 0.0;
    for (uint64_t i = 0; i < n; i++) v[i] = tab[(uint32_t)(cnt[i] - lo)];
    return true;
}

extern "C" int ps_sample_weights(const int32_t *num_genes, const double *logw, uint64_t n, uint64_t n_genes,
                                 int32_t avg_gene_num, const double *avg_pairwise_dists,
                                 int no_control_genome_size, double genome_size_penalty,
                                 double competition_strength, double *weights)
{
    if (!num_genes || !logw || !avg_pairwise_dists || !weights || n == 0)
        return ps_fail(PS_ERR_INVALID, "null argument");
    // (work vectors kept per thread: four 512 KB allocations per call at N = 65536 were page-faulted in every generation, on
    // the chain that has to fit beside the sweep)
    static thread_local std::vector<double> sel, tmp, scratch;
    sel.assign(n, 1.0);                                            // population.rs:293
    tmp.resize(n);
    if (n_genes > 0) {                                             // :296
        for (uint64_t i = 0; i < n; i++) sel[i] = logw[i];
        softmax_norm(sel.data(), n, scratch);                      // :325-340
    }
    if (!no_control_genome_size) {                                 // :346
        const double lp = std::log(genome_size_penalty);
        if (n < 4096 || !std::isfinite(lp) || !softmax_size(num_genes, avg_gene_num, lp, n, tmp.data())) {
            for (uint64_t i = 0; i < n; i++) tmp[i] = (double)(num_genes[i] - avg_gene_num) * lp; // :350-355
            softmax_norm(tmp.data(), n, scratch);                  // :356-361
        }
        for (uint64_t i = 0; i < n; i++) weights[i] = tmp[i] * sel[i]; // :368
    } else {
        for (uint64_t i = 0; i < n; i++) weights[i] = sel[i];      // :371
    }
    {
        // :375 -- the logarithm of N equal distances (1.0 without competition, main.rs:435) is taken once
        bool equal = true;
        for (uint64_t i = 1; equal && i < n; i++) equal = (avg_pairwise_dists[i] == avg_pairwise_dists[0]);
        if (equal) {
            const double t = competition_strength * std::log(avg_pairwise_dists[0]);
            for (uint64_t i = 0; i < n; i++) tmp[i] = t;
        } else {
            double *tp = tmp.data();          // (tmp is thread local: the workers must not name it)
            par_for(n, [&, tp](uint64_t a, uint64_t b) {
                for (uint64_t i = a; i < b; i++) tp[i] = competition_strength * std::log(avg_pairwise_dists[i]);
            });
        }
    }
    softmax_norm(tmp.data(), n, scratch);                          // :377-382
    for (uint64_t i = 0; i < n; i++) weights[i] = weights[i] * tmp[i]; // :389-393
    double mx = -INFINITY;
    for (uint64_t i = 0; i < n; i++) mx = std::fmax(mx, weights[i]); // :403
    if (mx == 0.0)
        for (uint64_t i = 0; i < n; i++) weights[i] = 1.0;         // :435-437
    double total = 0.0;                                            // WeightedIndex::new, :440
    for (uint64_t i = 0; i < n; i++) {
        if (!(weights[i] >= 0.0))
            return ps_fail(PS_ERR_WEIGHTS, "invalid sampling weight %g at %llu (reference panics, population.rs:440)",
                           weights[i], (unsigned long long)i);
        total += weights[i];
    }
    if (!(total > 0.0) || std::isinf(total))
        return ps_fail(PS_ERR_WEIGHTS, "sampling weights sum to %g (reference panics, population.rs:440)", total);
    return PS_OK;
}

extern "C" int ps_draw_parents(const double *weights, uint64_t n, uint64_t seed, uint32_t generation,
                               uint32_t *out_idx)
{
    if (!weights || !out_idx || n == 0) return ps_fail(PS_ERR_INVALID, "null argument");
    static thread_local std::vector<double> cum;
    cum.resize(n);
    double total = weights[0];
    for (uint64_t i = 1; i < n; i++) { cum[i - 1] = total; total += weights[i]; }
    // number of cumulative weights <= x among the first m = n - 1 (WeightedIndex::sample): fixed power-of-two steps, the
    // same for every draw, so that 8 draws advance in lockstep (independent dependency chains; the loop over the lanes
    // is what the host compiler unrolls / vectorises)
    const double *A = cum.data();
    const uint64_t m = n - 1;
    uint32_t top = 1;
    while ((uint64_t)top * 2 <= m) top *= 2;
    constexpr uint32_t LN = 8;
    auto batch = [&](uint64_t k0, uint32_t *dst) {        // draws k0 .. k0 + 7 (k0 a multiple of 8)
        // draws 2b and 2b + 1 are the two f64 of block b of the PARENTS stream (hs_f64): one Philox call serves both
        double x[LN];
        uint32_t pos[LN];
        for (uint32_t b = 0; b < LN / 2; b++) {
            const ps_u4 w = hs_block(seed, PS_STREAM_PARENTS, generation, k0 / 2 + b);
            const uint64_t x0 = ((uint64_t)w.y << 32) | w.x, x1 = ((uint64_t)w.w << 32) | w.z;
            x[2 * b] = (double)(x0 >> 11) * (1.0 / 9007199254740992.0) * total;
            x[2 * b + 1] = (double)(x1 >> 11) * (1.0 / 9007199254740992.0) * total;
            pos[2 * b] = pos[2 * b + 1] = 0;
        }
        for (uint32_t step = top; step; step >>= 1)
            for (uint32_t l = 0; l < LN; l++) {
                const uint32_t j = pos[l] + step;
                const uint32_t jj = j <= m ? j : (uint32_t)m;        // (clamped load, the comparison below rejects j > m)
                const uint32_t ok = (uint32_t)(j <= m) & (uint32_t)(A[jj - 1] <= x[l]);     // (arithmetic: clang turns a ?: into branches)
                pos[l] += step & (0u - ok);
            }
        for (uint32_t l = 0; l < LN; l++) dst[l] = pos[l];
    };
    if (m == 0) {
        out_idx[0] = 0;
        return PS_OK;
    }
    uint64_t k0 = 0;
    for (; k0 + LN <= n; k0 += LN) batch(k0, out_idx + k0);
    if (k0 < n) {
        uint32_t tail[LN];
        batch(k0, tail);
        for (uint64_t k = k0; k < n; k++) out_idx[k] = tail[k - k0];
    }
    return PS_OK;
}

extern "C" int ps_sample_indices(ps_population *acc, uint32_t generation, int32_t avg_gene_num,
                                 const double *avg_pairwise_dists, const double *sel, int verbose,
                                 int no_control_genome_size, double genome_size_penalty,
                                 double competition_strength, uint32_t *out_idx)
{
    (void)verbose;
    if (!acc || !out_idx || !avg_pairwise_dists) return ps_fail(PS_ERR_INVALID, "null argument");
    const uint64_t N = acc->cfg.pop_size;
    std::vector<int32_t> ng(N);
    std::vector<double> lw(N), w(N);
    PSCHK(ps_fitness_terms(acc, sel, ng.data(), lw.data()));
    PSCHK(ps_sample_weights(ng.data(), lw.data(), N, acc->cfg.ncols, avg_gene_num, avg_pairwise_dists,
                            no_control_genome_size, genome_size_penalty, competition_strength, w.data()));
    return ps_draw_parents(w.data(), N, acc->cfg.seed, generation, out_idx);
}

// ---------------------------------------------------------------------------
// distances
// ---------------------------------------------------------------------------
// D-avg (population.rs:753-784) into a device buffer of N doubles.  Rows [i_lo, i_lo + i_cnt) only (a row shard of a
// sharded run, DESIGN.md 6: the other entries of d_out are left alone); the whole population when i_cnt == N.
static int average_distance_device(ps_population *p, double *d_out, hipStream_t st, uint64_t i_lo = 0, uint64_t i_cnt = ~0ull)
{
    const uint64_t N = p->cfg.pop_size;
    if (i_cnt == ~0ull) i_cnt = N - i_lo;
    // wide populations (and every row shard): intersections on the matrix cores, ordered f64 fold in the accumulator layout
    // (acc_kernels.h); "davg_form": 0 = choose, 1 = LDS-tile popcount kernels, 2 = matrix cores in one kernel (round 4),
    // 3 = matrix cores in two phases (round 5: contraction on every SIMD -> u16 counts -> division + ordered fold)
    const bool whole = i_lo == 0 && i_cnt == N;
    const bool mfma = p->d.G > 0 && N >= 2 && (p->davg_form >= 2 || !whole || (p->davg_form == 0 && N > 8192));
    // Choice between the two matrix-core forms (davg_form 0).  The one-kernel form needs 32 NB rows per wave for the whole fold:
    // a row shard leaves its SIMDs idle (a rank of 8 at N = 65536: 7.6 ms for 8192 rows) and always takes the two phases (1.96 ms);
    // a whole population takes them up to about 52 K rows (round 6, blocked rows + pipelined phase 1 + 16-individual phase 2:
    // N = 32768 3.57 against 4.46 ms, 49152 7.68 against 8.31) -- above, the one-kernel form's fused epilogue is cheaper than the
    // counts written and read back (N = 57344: 9.95 against 10.29 ms; N = 65536: 11.1 against 13.3)
    const bool two_phase = mfma && p->davg_form != 2 && p->d.G <= 65535 && (p->davg_form == 3 || !whole || i_cnt <= 53248);      // (u16 counts)
    if (mfma) {
        const uint32_t WP = (2u * p->d.GW + 7u) & ~7u, Npad = (uint32_t)((N + 127) & ~127ull);
        const uint64_t need = (uint64_t)Npad * WP * 4 + (uint64_t)Npad * 4 + 64;
        if (p->davg_cap < need) {
            if (p->d_davg) HIPCHK(hipFree(p->d_davg));
            p->d_davg = nullptr;
            p->davg_cap = 0;
            HIPCHK(hipMalloc(&p->d_davg, need));
            p->davg_cap = need;
        }
        uint32_t *rowsP = (uint32_t *)p->d_davg, *rowcnt = rowsP + (uint64_t)Npad * WP;
        acc_rows_pad_kernel<<<(Npad + 3u) / 4u, 256, 0, st>>>(p->I[p->cur], rowsP, rowcnt, p->d, WP, Npad);
        const uint32_t lds = 256u * 64u * 4u;
        // (the lean division needs core_genes + the largest union below 2^32: always, short of an absurd --core_genes)
        const bool fast = p->cfg.core_genes < (1ull << 31) && !p->davg_plain_division;
        const uint32_t cgi = fast ? (uint32_t)p->cfg.core_genes : 0u;
        if (two_phase) {
            // phase 1 (contraction, every SIMD) -> u16 counts In[row][j]; phase 2 (division + the ordered fold).  Rows in
            // bands so that the scratch stays below ~9 GB (N = 65536 whole: one band of 8.6 GB)
            // (row pitch of the counts: Npad u16 + 256 bytes -- with a power-of-two pitch the 32 rows one store instruction
            // touches, and the 16 rows a phase-2 workgroup reads, fall on ONE memory channel)
            const uint32_t nb = p->davg_nb ? p->davg_nb : 2u, ld = Npad + 128u;
            // (a wave of phase 1 stores 32 * nb whole rows, at most 128: the band -- the scratch's row count -- is a multiple of that)
            uint64_t band = std::min<uint64_t>((i_cnt + 127) & ~127ull, std::max<uint64_t>(256, ((9ull << 30) / ((uint64_t)ld * 2)) & ~255ull));
            const uint64_t need_in = band * ld * 2;
            if (p->davg_in_cap < need_in) {
                if (p->d_davg_in) HIPCHK(hipFree(p->d_davg_in));
                p->d_davg_in = nullptr;
                p->davg_in_cap = 0;
                HIPCHK(hipMalloc(&p->d_davg_in, need_in));
                p->davg_in_cap = need_in;
            }
            uint16_t *In = (uint16_t *)p->d_davg_in;
            const uint32_t steps = Npad / 128u;
            for (uint64_t b0 = 0; b0 < i_cnt; b0 += band) {
                const uint32_t rows = (uint32_t)std::min<uint64_t>(band, i_cnt - b0), lo = (uint32_t)(i_lo + b0);
                const uint32_t gx = (rows + 128u * nb - 1u) / (128u * nb);
                uint32_t jsteps = 8u;
                while (jsteps > 1u && (uint64_t)gx * ((steps + jsteps - 1u) / jsteps) < 2048u) jsteps >>= 1;
                const dim3 grid(gx, (steps + jsteps - 1u) / jsteps);
                if (nb == 4u) {
                    HIPCHK(hipFuncSetAttribute((const void *)acc_intersections_mfma_kernel<4u>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    acc_intersections_mfma_kernel<4u><<<grid, 256, lds, st>>>(rowsP, WP, Npad, lo, rows, jsteps, In, ld);
                } else if (nb == 2u) {
                    HIPCHK(hipFuncSetAttribute((const void *)acc_intersections_mfma_kernel<2u>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    acc_intersections_mfma_kernel<2u><<<grid, 256, lds, st>>>(rowsP, WP, Npad, lo, rows, jsteps, In, ld);
                } else {
                    HIPCHK(hipFuncSetAttribute((const void *)acc_intersections_mfma_kernel<1u>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    acc_intersections_mfma_kernel<1u><<<grid, 256, lds, st>>>(rowsP, WP, Npad, lo, rows, jsteps, In, ld);
                }
                // (workgroups of 16 individuals -- four per CU, half the staging per fold -- except where 32 make exactly one round of
                // one per CU.  N = 65536: whole 4.99 against 5.42 ms, a shard of 8 0.889 against 0.836; N = 32768 whole 3.29
                // against 3.58 with phase 1, its shard of 8 0.65 against 0.73; N = 16384 shard 0.28 against 0.33)
                const uint32_t blocks32 = (rows + 31u) / 32u;
                const bool small = p->davg_ib ? p->davg_ib == 16u : !(blocks32 >= 256u && blocks32 < 512u);
                const uint32_t ib = small ? 16u : 32u, g2 = (rows + ib - 1u) / ib;
#define PS_AC_LAUNCH(FAST_, IB_) acc_average_from_counts_kernel<FAST_, IB_><<<g2, IB_ * 8u + 64u, 0, st>>>(In, ld, rowcnt, (uint32_t)N, lo, rows, (double)p->cfg.core_genes, cgi, d_out)
                if (fast) { if (small) PS_AC_LAUNCH(true, 16u); else PS_AC_LAUNCH(true, 32u); }
                else { if (small) PS_AC_LAUNCH(false, 16u); else PS_AC_LAUNCH(false, 32u); }
#undef PS_AC_LAUNCH
            }
        } else {
        // 64 individuals per wave (two B fragments: fewer table reads per MFMA) when that still gives every SIMD a wave
        const uint32_t nb = (p->davg_nb == 1u || p->davg_nb == 2u) ? p->davg_nb : (i_cnt >= 64u * 1024u ? 2u : 1u);
        const uint32_t waves = (uint32_t)((i_cnt + 32u * nb - 1) / (32u * nb)), grid = (waves + 3u) / 4u;
#define PS_DAVG_LAUNCH(NB_, FAST_)                                                                                                    \
        {                                                                                                                             \
            HIPCHK(hipFuncSetAttribute((const void *)acc_average_distance_mfma_kernel<NB_, FAST_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            acc_average_distance_mfma_kernel<NB_, FAST_><<<grid, 256, lds, st>>>(rowsP, WP, rowcnt, (uint32_t)N, Npad, (uint32_t)i_lo, (uint32_t)i_cnt, \
                                                                                (double)p->cfg.core_genes, cgi, d_out);                \
        }
        if (nb == 2u) { if (fast) PS_DAVG_LAUNCH(2u, true) else PS_DAVG_LAUNCH(2u, false) }
        else { if (fast) PS_DAVG_LAUNCH(1u, true) else PS_DAVG_LAUNCH(1u, false) }
#undef PS_DAVG_LAUNCH
        }
    } else if (N <= 8192) {
        if (!p->d_Dt) HIPCHK(hipMalloc(&p->d_Dt, N * N * sizeof(double)));
        const uint32_t nt = (uint32_t)((N + 63) / 64);
        if (nt * (nt + 1u) / 2u < 512u) {
            // few 64 x 64 tiles: 32 x 32 tiles (2 x 2 pairs per thread), four times the workgroups
            const uint32_t nt32 = (uint32_t)((N + 31) / 32);
            acc_pair_matrix_tiled_kernel<32u, 2u><<<dim3(nt32, nt32), 256, 0, st>>>(p->I[p->cur], p->d_Dt, p->d, (double)p->cfg.core_genes);
        } else
        acc_pair_matrix_tiled_kernel<64u, 4u><<<dim3(nt, nt), 256, 0, st>>>(p->I[p->cur], p->d_Dt, p->d, (double)p->cfg.core_genes);
        acc_average_from_matrix_kernel<<<(uint32_t)((N + PS_AV_IB - 1) / PS_AV_IB), 256, 0, st>>>(p->d_Dt, d_out, p->d);
    } else {
        acc_average_distance_tiled_kernel<<<(uint32_t)((N + 63) / 64), 256, 0, st>>>(p->I[p->cur], d_out, p->d,
                                                                                (double)p->cfg.core_genes);
    }
    HIPCHK(hipGetLastError());
    return PS_OK;
}

static int ensure_pairs(ps_population *p, uint64_t P)
{
    if (P <= p->pairs_cap) return PS_OK;
    if (p->d_pairs) HIPCHK(hipFree(p->d_pairs));
    p->d_pairs = nullptr;
    p->pairs_cap = 0;
    p->pairs_cached = 0;
    p->pairs_src1 = p->pairs_src2 = nullptr;      // the pointer-identity shortcut of upload_pairs dies with the device copy
    HIPCHK(hipMalloc(&p->d_pairs, P * 9 * sizeof(uint32_t)));   // r1 | r2 | perm | outA | outB | tstart | tcount | r1, r2 mapped to internal rows (ps_sim)
    p->pairs_cap = P;
    return PS_OK;
}

// Upload the pair list sorted by its first individual (consecutive lanes then read the same
// LDS row: a broadcast instead of a bank conflict) together with the permutation that
// restores the caller's order.  The list is fixed for a whole run (main.rs:413-427), so the
// device copy is reused while the caller keeps passing the same list.
struct core_pair_plan;
static bool core_pairs_take_a_tiled_kernel(const ps_population *p, uint64_t P);

static int upload_pairs(ps_population *p, uint64_t P, const uint32_t *r1, const uint32_t *r2, bool trusted = false)
{
    PSCHK(ensure_pairs(p, P));
    const uint64_t N = p->cfg.pop_size;
    // the sorted order and the thread table only serve the tiled kernel: the core matrix with a
    // population that fits its LDS tile (elsewhere -- accessory pairs, cfg5's all-pairs tiles -- the
    // caller's order is uploaded as it is)
    // (... and only when the launch will take one of those kernels: a ps_sim with --print_dist uploads a re-mapped list
    // every generation, and the sort + thread table + bank schedule of 100 000 pairs is host time the all-pairs form never uses)
    const bool tiled_possible = p->cfg.core && (uint64_t)N * (4 + 4) * 4 <= p->lds_limit && core_pairs_take_a_tiled_kernel(p, P);
    // (a ps_sim passes its own immutable list: same pointers as the cached copy was built from)
    if (p->pairs_cached == P && p->pairs_tiled == tiled_possible && p->h_r1.size() == P
        && ((trusted && p->pairs_src1 == r1 && p->pairs_src2 == r2)
            || (memcmp(p->h_r1.data(), r1, P * 4) == 0 && memcmp(p->h_r2.data(), r2, P * 4) == 0)))
        return PS_OK;
    p->pairs_cached = 0;
    p->pairs_tiled = tiled_possible;
    p->pairs_src1 = trusted ? r1 : nullptr;
    p->pairs_src2 = trusted ? r2 : nullptr;
    p->h_r1.assign(r1, r1 + P);
    p->h_r2.assign(r2, r2 + P);
    std::vector<uint32_t> s1(P), s2(P), perm(P);
    if (tiled_possible) {
        std::vector<uint32_t> start(N + 1, 0);
        for (uint64_t k = 0; k < P; k++) start[r1[k] + 1]++;
        for (uint64_t i = 0; i < N; i++) start[i + 1] += start[i];
        for (uint64_t k = 0; k < P; k++) {
            const uint32_t pos = start[r1[k]]++;
            s1[pos] = r1[k];
            s2[pos] = r2[k];
            perm[pos] = (uint32_t)k;
        }
    } else {
        memcpy(s1.data(), r1, P * 4);
        memcpy(s2.data(), r2, P * 4);
        for (uint64_t k = 0; k < P; k++) perm[k] = (uint32_t)k;
    }
    // thread table of the tiled kernel: every run of equal first individual is split evenly over
    // ceil(run / 32) threads, so that a thread's pairs share their first individual
    std::vector<uint32_t> tstart, tcount;
    for (uint64_t r0 = 0; tiled_possible && r0 < P;) {
        uint64_t r1e = r0;
        while (r1e < P && s1[r1e] == s1[r0]) r1e++;
        const uint64_t run = r1e - r0, nthr = (run + 31) / 32;
        for (uint64_t t = 0; t < nthr; t++) {
            const uint64_t a = r0 + run * t / nthr, b = r0 + run * (t + 1) / nthr;
            tstart.push_back((uint32_t)a);
            tcount.push_back((uint32_t)(b - a));
        }
        r0 = r1e;
    }
    // LDS bank scheduling: at step q the lanes of a wave read the 16-byte chunks of their q-th second
    // individuals; the LDS serves 8 such chunks per clock when their rows differ mod 8 (odd row stride
    // in 16-byte units).  The order inside a thread is free, so lane l takes residue (q + l) mod 8 at
    // step q whenever it has one left: the 8 lanes of a group then hit 8 different bank groups.
    // (measured: 2.95 -> 2.86 ms at cfg2; taking residues by lane / 8 or lane / 4 instead is slower)
    {
        std::vector<uint32_t> bucket[8], t2, tp;
        for (size_t g = 0; g < tstart.size(); g++) {
            const uint32_t a = tstart[g], n = tcount[g], l8 = (uint32_t)(g & 7u);
            if (n < 2) continue;
            for (auto &b : bucket) b.clear();
            for (uint32_t k = a; k < a + n; k++) bucket[s2[k] & 7u].push_back(k);
            t2.resize(n);
            tp.resize(n);
            for (uint32_t q = 0; q < n; q++) {
                uint32_t want = (q + l8) & 7u;
                if (bucket[want].empty()) {
                    size_t best = 0;
                    for (uint32_t b = 0; b < 8; b++)
                        if (bucket[b].size() > best) { best = bucket[b].size(); want = b; }
                }
                const uint32_t src = bucket[want].back();
                bucket[want].pop_back();
                t2[q] = s2[src];
                tp[q] = perm[src];
            }
            std::copy(t2.begin(), t2.end(), s2.begin() + a);
            std::copy(tp.begin(), tp.end(), perm.begin() + a);
        }
    }
    p->pair_threads = (uint32_t)tstart.size();
    uint32_t *d = (uint32_t *)p->d_pairs;
    HIPCHK(hipMemcpyAsync(d + 5 * P, tstart.data(), tstart.size() * 4, hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipMemcpyAsync(d + 6 * P, tcount.data(), tcount.size() * 4, hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipMemcpyAsync(d, s1.data(), P * 4, hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipMemcpyAsync(d + P, s2.data(), P * 4, hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipMemcpyAsync(d + 2 * P, perm.data(), P * 4, hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));   // the staging vectors die here
    p->pairs_cached = P;
    return PS_OK;
}

// partial counts of the tiled distance kernels, part[range][P]
static int pair_partials(ps_population *p, uint64_t words, uint32_t **out)
{
    if (p->pair_part_cap < words) {
        if (p->d_pair_part) HIPCHK(hipFree(p->d_pair_part));
        p->d_pair_part = nullptr;
        p->pair_part_cap = 0;
        HIPCHK(hipMalloc(&p->d_pair_part, words * sizeof(uint32_t)));
        p->pair_part_cap = words;
    }
    *out = p->d_pair_part;
    return PS_OK;
}

// which form the core pair counts of P pairs take (one decision, shared by the launch below and by upload_pairs, which
// only sorts the list and builds the thread table when a tiled sampled kernel will read them)
struct core_pair_plan { uint32_t W; bool use_rows, use_all, mfma_ok; };
static core_pair_plan plan_core_pairs(const ps_population *p, uint64_t P)
{
    const uint32_t N = (uint32_t)p->cfg.pop_size;
    const uint32_t rows = (uint32_t)p->cfg.ncols;
        // tiled kernel: LDS holds N * (W+4) dwords; prefer a tile that lets two workgroups share
        // a CU so that one packs its next tile (HBM) while the other compares (LDS)
        uint32_t W = 0;
        for (uint32_t w : { 32u, 16u, 8u, 4u })
            if ((uint64_t)N * (w + 4) * 4 <= p->lds_limit) { W = w; break; }
        // all-pairs tiles cost ~N^2/2 * L regardless of P; the sampled kernel ~P * L with a
        // worse constant (it re-packs the matrix once per 32768 pairs): measured crossover at
        // N = 1000 is P ~ 250 k = half of all pairs
        const double all_pairs = 0.5 * (double)N * (double)N;
        const bool all_fits = p->nibble_safe && (uint64_t)N * N * 4 <= (8ull << 30);
        // Populations too wide for an LDS tile (W == 0): the transposed form (pack once, stream two bit
        // strings per pair: ~1.25 * N * L + P * L / 2 (nibbles: P * L) bytes at ~5 TB/s) against the
        // all-pairs tiles (~0.9e15 pair-sites/s, N = 8192: 45 ms; crossover there at P ~ 350 k)
        // (all-pairs on the matrix cores for one-hot matrices: ~3.5e14 pair-sites/s; the xor + popcount tiles: 1.3e14)
        const bool mfma_ok = p->onehot_safe && p->pair_mode != 5;
        const double t_rows = (1.25 * (double)N * rows + (double)P * rows * (p->onehot_safe ? 0.5 : 1.0)) / 5.0e12;
        const double t_all = all_pairs * rows / (mfma_ok ? 3.5e14 : 1.3e14);
        const bool force_all = p->pair_mode == 2 || p->pair_mode == 5 || p->pair_mode == 6 || p->pair_mode == 7;
        const bool use_rows = p->nibble_safe && !force_all
                              && (p->pair_mode == 4 || (!W && (p->pair_mode == 1 || p->pair_mode == 3 || !all_fits || t_rows < t_all)));
        // against the sampled LDS-tile kernels (one-hot: ~6.4e13 pair-sites/s): the xor + popcount all-pairs tiles from
        // P > N^2 / 4 on; the matrix-core form (FP4: ~5.5e14 pair-sites/s over whole 256 x 256 tiles) by the time model --
        // cfg2 (N = 1000, P = 100 k): 1.53 ms against 2.0 ms
        const double mf_tiles = std::ceil(N / 256.0) * (std::ceil(N / 256.0) + 1.0) * 0.5 * 65536.0;
        const bool all_wins = mfma_ok ? mf_tiles / 5.5e14 < (double)P / 6.4e13 : (double)P * 2.0 > all_pairs;
        const bool use_all = !use_rows && all_fits && p->pair_mode != 1 && p->pair_mode != 3
                             && (force_all || all_wins || !W);
    core_pair_plan o = { W, use_rows, use_all, mfma_ok };
    return o;
}

static bool core_pairs_take_a_tiled_kernel(const ps_population *p, uint64_t P)
{
    if (!p->cfg.core || p->cfg.ncols == 0) return false;
    const core_pair_plan pp = plan_core_pairs(p, P);
    return !pp.use_rows && !pp.use_all && pp.W != 0u;
}

static int pair_counts_device(ps_population *p, uint64_t P, const uint32_t *d_r1, const uint32_t *d_r2,
                              const uint32_t *d_perm, uint32_t *d_a, uint32_t *d_b, hipStream_t st)
{
    const uint32_t N = (uint32_t)p->cfg.pop_size;
    if (p->cfg.core) {
        HIPCHK(hipMemsetAsync(d_a, 0, P * sizeof(uint32_t), st));
        const uint32_t rows = (uint32_t)p->cfg.ncols;
        if (rows == 0) return PS_OK;
        const core_pair_plan pp = plan_core_pairs(p, P);
        const uint32_t W = pp.W;
        const bool use_rows = pp.use_rows, use_all = pp.use_all, mfma_ok = pp.mfma_ok;
        if (use_rows) {
            p->last_pair_form = PS_PAIR_FORM_ROWS;
            const bool nib = !p->onehot_safe;
            const uint32_t spw = nib ? 8u : 16u;
            const uint32_t WT = ((rows + spw - 1u) / spw + PS_PT_WB - 1u) / PS_PT_WB * PS_PT_WB;
            const uint64_t need = (uint64_t)N * WT;
            if (p->pack2_cap < need) {
                if (p->d_pack2) HIPCHK(hipFree(p->d_pack2));
                p->d_pack2 = nullptr;
                p->pack2_cap = 0;
                HIPCHK(hipMalloc(&p->d_pack2, need * sizeof(uint32_t)));
                p->pack2_cap = need;
            }
            const uint64_t ptiles = (uint64_t)((N + PS_PT_IB - 1u) / PS_PT_IB) * (WT / PS_PT_WB);
            if (ptiles > 0x7FFFFFFFull) return ps_fail(PS_ERR_INVALID, "matrix too large for the transposed distance form");
            const dim3 pgrid((uint32_t)ptiles);
            const uint32_t blocks = (uint32_t)std::min<uint64_t>((P + 3) / 4, 256u * 8u * 4u);
            if (nib) {
                core_packT_kernel<true><<<pgrid, 256, 0, st>>>(p->state, N, p->pitch, rows, p->d_pack2, WT);
                core_pair_counts_rows<true><<<blocks, 256, 0, st>>>(p->d_pack2, WT, d_r1, d_r2, d_perm, P, d_a);
            } else {
                core_packT_kernel<false><<<pgrid, 256, 0, st>>>(p->state, N, p->pitch, rows, p->d_pack2, WT);
                core_pair_counts_rows<false><<<blocks, 256, 0, st>>>(p->d_pack2, WT, d_r1, d_r2, d_perm, P, d_a);
            }
        } else
        if (use_all && mfma_ok) {
            // one-hot matrix: X X^T on the matrix cores from the blocked 2-bit strings (or bit planes: the signed form)
            p->last_pair_form = PS_PAIR_FORM_ALLPAIRS_MFMA;
            const uint32_t WT = ((rows + 15u) / 16u + PS_PT_WB - 1u) / PS_PT_WB * PS_PT_WB;
            const uint64_t need = (uint64_t)((N + 31u) / 32u * 32u) * WT;       // blocked strings: whole groups of 32 individuals
            if (p->pack2_cap < need) {
                if (p->d_pack2) HIPCHK(hipFree(p->d_pack2));
                p->d_pack2 = nullptr;
                p->pack2_cap = 0;
                HIPCHK(hipMalloc(&p->d_pack2, need * sizeof(uint32_t)));
                p->pack2_cap = need;
            }
            const uint64_t ptiles = (uint64_t)((N + PS_PT_IB - 1u) / PS_PT_IB) * (WT / PS_PT_WB);
            if (ptiles > 0x7FFFFFFFull) return ps_fail(PS_ERR_INVALID, "matrix too large for the transposed distance form");
            const uint32_t ntile = (N + PS_MF_TILE - 1u) / PS_MF_TILE;
            const uint32_t tile_pairs = ntile * (ntile + 1u) / 2u;
            const uint32_t n_chunks = WT / PS_MF_CHUNK_DW;
            // (tile, chunk range) workgroups, one 8-wave workgroup per CU at a time (242 VGPRs): the number of ranges that
            // minimises rounds x (time of a workgroup + its fixed cost) -- whole rounds of 256 workgroups, e.g. N = 8192:
            // 528 tiles x 16 ranges = 33 rounds exactly, where 3 ranges left the 7th round 19 % full
            uint32_t ranges = 1;
            {
                const double cu_rate = 7.0e14 / 256.0, fixed = 10.0e-6;
                double best = 1.0e300;
                // (every range stores its own N x N slice of partial counts: at most ~8 GB of them)
                const uint32_t r_mem = (uint32_t)std::max<uint64_t>(1, (8ull << 30) / ((uint64_t)N * N * 4));
                const uint32_t r_hi = std::min(std::min(n_chunks, r_mem), std::max(1u, (256u * 64u) / tile_pairs));
                for (uint32_t r = 1; r <= r_hi; r++) {
                    const uint32_t c = (n_chunks + r - 1u) / r, rr = (n_chunks + c - 1u) / c;     // ranges actually launched
                    const double blocks = (double)tile_pairs * rr;
                    // rounds x (a workgroup's chunks + table, stores) + the slices written and read once more
                    const double t = std::ceil(blocks / 256.0) * (65536.0 * (double)c * PS_MF_CHUNK_DW * 16.0 / cu_rate + fixed)
                                     + (rr > 1u ? (double)rr * N * N * 4.0 * 1.5 / 4.0e12 : 0.0);
                    if (t < best) { best = t; ranges = rr; }
                }
            }
            const uint32_t cpr = (n_chunks + ranges - 1u) / ranges;
            ranges = (n_chunks + cpr - 1u) / cpr;          // (every range holds at least one chunk: every slice is written)
            if (p->H_cap < (uint64_t)N * N * ranges) {
                if (p->d_H) HIPCHK(hipFree(p->d_H));
                p->d_H = nullptr;
                p->H_cap = 0;
                HIPCHK(hipMalloc(&p->d_H, (uint64_t)N * N * ranges * sizeof(uint32_t)));
                p->H_cap = (uint64_t)N * N * ranges;
            }
            const uint32_t lds = 256u * 16u * 16u;
            // i8 (pair_mode 6), the block-scaled FP4 form on one-hot {0, 1} nibbles (the default: twice the sites per instruction
            // at the same cycles; a range must stay below 2^24 sites for the f32 sums to be exact) or (pair_mode 7) the FP4 form
            // on three +-1 features per site: 6 instructions per 128 sites instead of 8, but twice the table reads per
            // instruction -- no faster (N = 8192: 51.9 ms both; N = 3000: 3.5 against 2.7 ms), kept as a cross-check
            const bool fp4 = p->pair_mode != 6 && (uint64_t)cpr * PS_MF_CHUNK_DW * 16u < (1u << 24);
            const bool sgn = fp4 && p->pair_mode == 7 && 3ull * cpr * PS_MF_CHUNK_DW * 16u < (1u << 24);
            if (sgn) core_packT_kernel<false, true, true><<<dim3((uint32_t)ptiles), 256, 0, st>>>(p->state, N, p->pitch, rows, p->d_pack2, WT);
            else core_packT_kernel<false, true><<<dim3((uint32_t)ptiles), 256, 0, st>>>(p->state, N, p->pitch, rows, p->d_pack2, WT);
            if (sgn) {
                HIPCHK(hipFuncSetAttribute((const void *)core_allpairs_mfma_signed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(core_allpairs_mfma_signed_kernel, dim3(tile_pairs, ranges), dim3(512), lds, st, p->d_pack2, WT, N, p->d_H, cpr, n_chunks, ntile);
                p->last_pair_form = PS_PAIR_FORM_ALLPAIRS_MFMA_SIGNED;
            } else
            if (fp4) {
                HIPCHK(hipFuncSetAttribute((const void *)core_allpairs_mfma_fp4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(core_allpairs_mfma_fp4_kernel, dim3(tile_pairs, ranges), dim3(512), lds, st, p->d_pack2, WT, N, p->d_H, cpr, n_chunks, ntile);
                p->last_pair_form = PS_PAIR_FORM_ALLPAIRS_MFMA_FP4;
            } else {
            HIPCHK(hipFuncSetAttribute((const void *)core_allpairs_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(core_allpairs_mfma_kernel, dim3(tile_pairs, ranges), dim3(512), lds, st, p->d_pack2, WT, N, p->d_H, cpr, n_chunks, ntile);
            }
            if (ranges > 1u) core_allpairs_sum_slices_kernel<<<N, 256, 0, st>>>(p->d_H, N, ranges);
            core_pair_lookup256_kernel<<<(uint32_t)((P + 255) / 256), 256, 0, st>>>(p->d_H, N, d_r1, d_r2, d_perm, P, d_a);
        } else
        if (use_all) {
            p->last_pair_form = PS_PAIR_FORM_ALLPAIRS;
            const uint32_t WA = 32u, ntile = (N + 127u) / 128u;
            const uint32_t lds = 2u * 128u * ((WA >> 2) + 1u) * 16u;
            if (p->H_cap < (uint64_t)N * N) {
                if (p->d_H) HIPCHK(hipFree(p->d_H));
                p->d_H = nullptr;
                p->H_cap = 0;
                HIPCHK(hipMalloc(&p->d_H, (uint64_t)N * N * sizeof(uint32_t)));
                p->H_cap = (uint64_t)N * N;
            }
            HIPCHK(hipMemsetAsync(p->d_H, 0, (uint64_t)N * N * sizeof(uint32_t), st));
            const uint32_t tile_pairs = ntile * (ntile + 1u) / 2u;
            const uint32_t n_chunks = (rows + WA * 8u - 1u) / (WA * 8u);
            uint32_t ranges = std::max(1u, std::min(n_chunks, (256u * 16u + tile_pairs - 1u) / tile_pairs));
            const uint32_t cpr = (n_chunks + ranges - 1u) / ranges;
            ranges = (n_chunks + cpr - 1u) / cpr;
            core_allpairs_kernel<<<dim3(tile_pairs, ranges), 256, lds, st>>>(p->state, N, p->pitch, rows, p->d_H, WA,
                                                                          cpr, ntile);
            core_pair_lookup_kernel<<<(uint32_t)((P + 255) / 256), 256, 0, st>>>(p->d_H, N, d_r1, d_r2, d_perm, P, d_a);
        } else if (p->onehot_safe && W && p->pair_mode != 3) {
            // one-hot matrix: pack it once at 2 bits per site, then compare from packed tiles
            p->last_pair_form = PS_PAIR_FORM_TILED2;
            constexpr int A = 32;
            constexpr uint32_t PT = 1024;
            const uint32_t n_tiles = (rows + W * 16 - 1) / (W * 16);
            const uint64_t need = (uint64_t)n_tiles * N * W;
            if (p->pack2_cap < need) {
                if (p->d_pack2) HIPCHK(hipFree(p->d_pack2));
                p->d_pack2 = nullptr;
                p->pack2_cap = 0;
                HIPCHK(hipMalloc(&p->d_pack2, need * sizeof(uint32_t)));
                p->pack2_cap = need;
            }
            const uint32_t pack_lds = N * W * 4u;
            if (pack_lds > 64 * 1024)
                HIPCHK(hipFuncSetAttribute((const void *)core_pack2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pack_lds));
            hipLaunchKernelGGL(core_pack2_kernel, dim3(n_tiles), dim3(PT), pack_lds, st, p->state, N, p->pitch, rows, p->d_pack2, W);
            const uint32_t lds = N * ((W >> 2) + 1u) * 16u;
            auto kern = core_pair_counts_packed2<A>;
            if (lds > 64 * 1024)
                HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const uint32_t pair_blocks = std::max(1u, (p->pair_threads + PT - 1) / PT);
            uint32_t ranges = std::max(1u, std::min(n_tiles, p->pair_ranges ? p->pair_ranges : (256u * 4u + pair_blocks - 1) / pair_blocks));
            // the kernel counts a range in 16 bits: fewer than 65536 sites per range
            const uint32_t max_tpr = std::max(1u, 65535u / (W * 16u));
            ranges = std::max(ranges, (n_tiles + max_tpr - 1) / max_tpr);
            const uint32_t tpr = (n_tiles + ranges - 1) / ranges;
            ranges = (n_tiles + tpr - 1) / tpr;
            uint32_t *part = nullptr;
            PSCHK(pair_partials(p, (uint64_t)ranges * P, &part));
            hipLaunchKernelGGL(kern, dim3(pair_blocks, ranges), dim3(PT), lds, st, p->d_pack2, N, n_tiles, d_r1, d_r2,
                               (uint32_t)P, d_r1 + 5 * P, d_r1 + 6 * P, p->pair_threads, part, W, tpr);
            core_pair_reduce_kernel<<<(uint32_t)((P + 255) / 256), 256, 0, st>>>(part, ranges, (uint32_t)P, d_perm, d_a);
        } else if (p->nibble_safe && W) {
            p->last_pair_form = PS_PAIR_FORM_TILED4;
            constexpr int A = 32;
            constexpr uint32_t PT = 1024;
            const uint32_t lds = N * ((W >> 2) + 1u) * 16u;
            auto kern = core_pair_counts_tiled<A>;
            if (lds > 64 * 1024)
                HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const uint32_t n_tiles = (rows + W * 8 - 1) / (W * 8);
            // every workgroup re-packs its site tiles: 1024 threads of the thread table per pair block
            const uint32_t pair_blocks = std::max(1u, (p->pair_threads + PT - 1) / PT);
            // enough site ranges to fill the chip a few times over
            uint32_t ranges = std::max(1u, std::min(n_tiles, p->pair_ranges ? p->pair_ranges : (256u * 4u + pair_blocks - 1) / pair_blocks));
            // the kernel counts a range in 16 bits; a nibble pair differs in up to 4 bits per site
            // (only one-hot bytes are limited to 2): 4 * sites per range < 65536
            const uint32_t max_tpr = std::max(1u, 16383u / (W * 8u));
            ranges = std::max(ranges, (n_tiles + max_tpr - 1) / max_tpr);
            const uint32_t tpr = (n_tiles + ranges - 1) / ranges;
            ranges = (n_tiles + tpr - 1) / tpr;
            uint32_t *part = nullptr;
            PSCHK(pair_partials(p, (uint64_t)ranges * P, &part));
            hipLaunchKernelGGL(kern, dim3(pair_blocks, ranges), dim3(PT), lds, st, p->state, N, p->pitch,
                               rows, d_r1, d_r2, (uint32_t)P, d_r1 + 5 * P, d_r1 + 6 * P, p->pair_threads, part, W, tpr);
            core_pair_reduce_kernel<<<(uint32_t)((P + 255) / 256), 256, 0, st>>>(part, ranges, (uint32_t)P, d_perm, d_a);
        } else {
            p->last_pair_form = PS_PAIR_FORM_SIMPLE;
            const uint32_t slices = std::max(1u, std::min(rows, 64u));
            const uint32_t rps = (rows + slices - 1) / slices;
            dim3 grid((uint32_t)((P + 255) / 256), (rows + rps - 1) / rps);
            core_pair_counts_simple<<<grid, 256, 0, st>>>(p->state, p->pitch, rows, d_r1, d_r2, d_perm, P, d_a, rps);
        }
    } else {
        // more sampled pairs than a quarter of all pairs (cfg5) and room for the N x N matrix: all intersections from LDS
        // tiles, then a lookup; otherwise one thread per sampled pair
        const bool all = p->pair_mode != 1 && (uint64_t)N * N * 4 <= (2ull << 30)
                         && (p->pair_mode == 2 || (double)P * 4.0 > (double)N * (double)N);
        if (all) {
            if (p->H_cap < (uint64_t)N * N + N) {
                if (p->d_H) HIPCHK(hipFree(p->d_H));
                p->d_H = nullptr;
                p->H_cap = 0;
                HIPCHK(hipMalloc(&p->d_H, ((uint64_t)N * N + N) * sizeof(uint32_t)));
                p->H_cap = (uint64_t)N * N + N;
            }
            uint32_t *rowcnt = p->d_H + (uint64_t)N * N;
            const uint32_t nt = (N + 63u) / 64u;
            acc_pair_inter_tiled_kernel<<<dim3(nt, nt), 256, 0, st>>>(p->I[p->cur], p->d_H, rowcnt, p->d);
            acc_pair_lookup_kernel<<<(uint32_t)((P + 255) / 256), 256, 0, st>>>(p->d_H, rowcnt, N, d_r1, d_r2, d_perm, P, d_a, d_b);
        } else {
            acc_pair_counts_kernel<<<(uint32_t)((P * 8 + 255) / 256), 256, 0, st>>>(p->I[p->cur], d_r1, d_r2, d_perm, P,
                                                                             d_a, d_b, p->d);
        }
    }
    HIPCHK(hipGetLastError());
    return PS_OK;
}

// the run's pair list, drawn over the output rows, as internal rows
__global__ void __launch_bounds__(256) pairs_map_kernel(const uint32_t *r1, const uint32_t *r2, const uint32_t *slot, uint32_t *m1,
                                                        uint32_t *m2, uint64_t P)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    m1[k] = slot[r1[k]];
    m2[k] = slot[r2[k]];
}

// The device copy of a pair list whose individuals are named by OUTPUT rows (a ps_sim's handle after a generation), as the
// kernels want it: internal rows.  Mapped on the device from the cached upload by one small kernel -- unless a tiled sampled
// kernel will run, which wants the list sorted by first individual: then mapped on the host and uploaded with its tables.
static int pairs_for_launch(ps_population *p, uint64_t P, const uint32_t *r1, const uint32_t *r2, bool trusted, const uint32_t *slot,
                            uint32_t **d_r1, uint32_t **d_r2, hipStream_t st)
{
    const bool dev_map = slot && !core_pairs_take_a_tiled_kernel(p, P);
    if (slot && !dev_map) {
        std::vector<uint32_t> m1(P), m2(P);
        for (uint64_t k = 0; k < P; k++) { m1[k] = slot[r1[k]]; m2[k] = slot[r2[k]]; }
        PSCHK(upload_pairs(p, P, m1.data(), m2.data()));
    } else {
        PSCHK(upload_pairs(p, P, r1, r2, trusted));
    }
    uint32_t *d = (uint32_t *)p->d_pairs;
    *d_r1 = d;
    *d_r2 = d + P;
    if (dev_map) {
        *d_r1 = d + 7 * P;
        *d_r2 = d + 8 * P;
        pairs_map_kernel<<<(uint32_t)((P + 255) / 256), 256, 0, st>>>(d, d + P, p->d_row_slot, *d_r1, *d_r2, P);
        HIPCHK(hipGetLastError());
    }
    return PS_OK;
}

extern "C" int ps_pairwise_counts(ps_population *p, uint64_t P, const uint32_t *range1, const uint32_t *range2,
                                  uint32_t *out_a, uint32_t *out_b, int out_is_device)
{
    if (!p || !range1 || !range2 || !out_a) return ps_fail(PS_ERR_INVALID, "null argument");
    if (!p->cfg.core && !out_b) return ps_fail(PS_ERR_INVALID, "out_b is required for the accessory matrix");
    if (P == 0) return PS_OK;
    const uint64_t N = p->cfg.pop_size;
    for (uint64_t k = 0; k < P; k++)
        if (range1[k] >= N || range2[k] >= N)
            return ps_fail(PS_ERR_INVALID, "pair %llu out of range", (unsigned long long)k);
    PSCHK(use_device(p));
    const uint32_t *slot = nullptr;
    PSCHK(rows_current(p, &slot));
    // (individuals are named by their output row -- the child of draw k -- on a ps_sim's handle: their internal rows)
    uint32_t *d_r1 = nullptr, *d_r2 = nullptr;
    PSCHK(pairs_for_launch(p, P, range1, range2, false, slot, &d_r1, &d_r2, p->stream));
    uint32_t *d_perm = (uint32_t *)p->d_pairs + 2 * P, *d_a = d_perm + P, *d_b = d_a + P;
    uint32_t *ka = out_is_device ? out_a : d_a;
    uint32_t *kb = out_is_device ? out_b : d_b;
    PSCHK(pair_counts_device(p, P, d_r1, d_r2, d_perm, ka, kb, p->stream));
    if (!out_is_device) {
        HIPCHK(hipMemcpyAsync(out_a, d_a, P * 4, hipMemcpyDeviceToHost, p->stream));
        if (!p->cfg.core) HIPCHK(hipMemcpyAsync(out_b, d_b, P * 4, hipMemcpyDeviceToHost, p->stream));
    }
    HIPCHK(hipStreamSynchronize(p->stream));
    return PS_OK;
}

extern "C" int ps_last_sweep_form(ps_population *p)
{
    if (!p) return ps_fail(PS_ERR_INVALID, "null handle");
    return p->last_sweep_form;
}

extern "C" int ps_last_pair_form(ps_population *p) { return p ? p->last_pair_form : 0; }

extern "C" int ps_pairwise_distances(ps_population *p, uint64_t P, const uint32_t *range1,
                                     const uint32_t *range2, double *out)
{
    if (!p || !out) return ps_fail(PS_ERR_INVALID, "null argument");
    std::vector<uint32_t> a(P), b(P);
    PSCHK(ps_pairwise_counts(p, P, range1, range2, a.data(), b.data(), 0));
    const double ncols = (double)p->cfg.ncols, cg = (double)p->cfg.core_genes;
    for (uint64_t k = 0; k < P; k++) {
        if (p->cfg.core) {
            const uint32_t distance = a[k] / 2;                      // population.rs:817
            out[k] = (double)distance / ncols;                       // :822
        } else {
            out[k] = 1.0 - (((double)a[k] + cg) / ((double)b[k] + cg)); // :828-830
        }
    }
    return PS_OK;
}

extern "C" int ps_average_distance(ps_population *p, double *out)
{
    if (!p || !out) return ps_fail(PS_ERR_INVALID, "null argument");
    if (p->cfg.core)
        return ps_fail(PS_ERR_INVALID, "average_distance is only reached on the accessory matrix (main.rs:439)");
    if (p->cfg.pop_size < 2) return ps_fail(PS_ERR_INVALID, "average_distance needs pop_size >= 2");
    PSCHK(use_device(p));
    double *d_out = nullptr;
    const uint64_t N = p->cfg.pop_size;
    HIPCHK(hipMalloc(&d_out, N * sizeof(double)));
    PSCHK(average_distance_device(p, d_out, p->stream));
    HIPCHK(hipMemcpyAsync(out, d_out, N * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipFree(d_out));
    const uint32_t *slot = nullptr;
    PSCHK(rows_current(p, &slot));
    rows_permute(out, slot, N);
    return PS_OK;
}

// the rows [first, first + count) of average_distance: what one rank of a row-sharded D-avg computes
extern "C" int ps_average_distance_rows(ps_population *p, uint64_t first, uint64_t count, double *out)
{
    if (!p || !out) return ps_fail(PS_ERR_INVALID, "null argument");
    if (p->cfg.core)
        return ps_fail(PS_ERR_INVALID, "average_distance is only reached on the accessory matrix (main.rs:439)");
    const uint64_t N = p->cfg.pop_size;
    if (N < 2) return ps_fail(PS_ERR_INVALID, "average_distance needs pop_size >= 2");
    if (count < 1 || first >= N || count > N - first) return ps_fail(PS_ERR_INVALID, "rows [%llu, +%llu) outside the population",
                                                                     (unsigned long long)first, (unsigned long long)count);
    if (p->d.G == 0) {
        // no accessory genes: every pair is 1 - core_genes / core_genes (get_distance, population.rs:144-145)
        std::vector<double> all(N);
        PSCHK(ps_average_distance(p, all.data()));
        memcpy(out, all.data() + first, count * sizeof(double));
        return PS_OK;
    }
    PSCHK(use_device(p));
    double *d_out = nullptr;
    HIPCHK(hipMalloc(&d_out, N * sizeof(double)));
    PSCHK(average_distance_device(p, d_out, p->stream, first, count));
    HIPCHK(hipMemcpyAsync(out, d_out + first, count * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipFree(d_out));
    return PS_OK;
}

extern "C" int ps_gene_frequencies(ps_population *p, double *out)
{
    if (!p || !out) return ps_fail(PS_ERR_INVALID, "null argument");
    if (p->cfg.core) return ps_fail(PS_ERR_INVALID, "gene_frequencies runs on the accessory matrix (main.rs:492)");
    PSCHK(use_device(p));
    const uint64_t G = p->cfg.ncols;
    if (G) {
        uint32_t *d_c = nullptr;
        HIPCHK(hipMalloc(&d_c, G * sizeof(uint32_t)));
        PSCHK(ensure_gene_major(p, p->stream));
        acc_gene_counts_kernel<<<(uint32_t)((G + 255) / 256), 256, 0, p->stream>>>(p->G[0], d_c, p->d);
        HIPCHK(hipGetLastError());
        std::vector<uint32_t> c(G);
        HIPCHK(hipMemcpyAsync(c.data(), d_c, G * 4, hipMemcpyDeviceToHost, p->stream));
        HIPCHK(hipStreamSynchronize(p->stream));
        HIPCHK(hipFree(d_c));
        const double n_individuals = (double)p->cfg.pop_size;      // population.rs:843
        for (uint64_t g = 0; g < G; g++) out[g] = (double)c[g] / n_individuals; // :852
    }
    for (uint64_t k = 0; k < p->cfg.core_genes; k++) out[G + k] = 1.0;  // :858-860
    return PS_OK;
}

extern "C" int ps_calc_gene_freq(ps_population *p, double *out)
{
    if (!p || !out) return ps_fail(PS_ERR_INVALID, "null argument");
    if (p->cfg.core) return ps_fail(PS_ERR_INVALID, "calc_gene_freq runs on the accessory matrix (main.rs:524)");
    const uint64_t N = p->cfg.pop_size, G = p->cfg.ncols;
    std::vector<int32_t> ng(N);
    std::vector<double> lw(N);
    std::vector<double> zero(G, 0.0);
    PSCHK(ps_fitness_terms(p, zero.data(), ng.data(), lw.data()));
    double sum = 0.0;
    for (uint64_t i = 0; i < N; i++) sum += (double)ng[i] / (double)G;  // population.rs:250-252
    *out = sum / (double)N;                                             // :265
    return PS_OK;
}

// ---------------------------------------------------------------------------
// free functions
// ---------------------------------------------------------------------------
static int slice_counts(const uint8_t *x, const uint8_t *y, size_t n, uint32_t out[3])
{
    out[0] = out[1] = out[2] = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return ps_fail(PS_ERR_NO_DEVICE, "no HIP device is visible: libpansim_hip has no CPU path");
    if (n == 0) return PS_OK;
    if (!x || !y) return ps_fail(PS_ERR_INVALID, "null slice");
    uint8_t *d = nullptr;
    uint32_t *d_out = nullptr;
    HIPCHK(hipMalloc(&d, 2 * n));
    HIPCHK(hipMalloc(&d_out, 3 * sizeof(uint32_t)));
    HIPCHK(hipMemcpy(d, x, n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d + n, y, n, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(d_out, 0, 3 * sizeof(uint32_t)));
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 2048);
    slice_counts_kernel<<<blocks, 256>>>(d, d + n, n, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, d_out, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIPCHK(hipFree(d));
    HIPCHK(hipFree(d_out));
    return PS_OK;
}

extern "C" int ps_hamming_bitwise_fast(const uint8_t *x, const uint8_t *y, size_t n, uint32_t *out)
{
    if (!out) return ps_fail(PS_ERR_INVALID, "null output");
    uint32_t c[3];
    PSCHK(slice_counts(x, y, n, c));
    *out = c[0];
    return PS_OK;
}

extern "C" int ps_jaccard_distance_fast(const uint8_t *x, const uint8_t *y, size_t n, uint32_t *inter,
                                        uint32_t *uni)
{
    if (!inter || !uni) return ps_fail(PS_ERR_INVALID, "null output");
    uint32_t c[3];
    PSCHK(slice_counts(x, y, n, c));
    *inter = c[1];
    *uni = c[2];
    return PS_OK;
}

extern "C" int ps_standard_deviation(const double *v, uint64_t n, double *std_out, double *mean_out)
{
    if (!v || !std_out || !mean_out) return ps_fail(PS_ERR_INVALID, "null argument");
    double s = 0.0;
    for (uint64_t i = 0; i < n; i++) s += v[i];                     // population.rs:84
    const double mean = s / (double)n;
    double ss = 0.0;
    for (uint64_t i = 0; i < n; i++) { const double dd = v[i] - mean; ss += dd * dd; } // :90
    *std_out = std::sqrt(ss / (double)n);                           // :92-93
    *mean_out = mean;
    return PS_OK;
}

extern "C" char ps_int_to_base(uint8_t n)
{
    switch (n) {                                                    // population.rs:154-162
    case 1: return 'A';
    case 2: return 'C';
    case 4: return 'G';
    case 8: return 'T';
    default: return 'N';
    }
}

// Rust `{}` for f64: shortest round-trip digits, positional notation, no ".0"
static std::string fmt_f64(double v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    if (v == 0.0) return std::signbit(v) ? "-0" : "0";
    char tmp[64];
    auto r = std::to_chars(tmp, tmp + sizeof tmp, v, std::chars_format::scientific);
    std::string s(tmp, r.ptr);
    bool neg = false;
    size_t pos = 0;
    if (s[0] == '-') { neg = true; pos = 1; }
    const size_t epos = s.find('e');
    std::string digits;
    for (size_t k = pos; k < epos; k++)
        if (s[k] != '.') digits.push_back(s[k]);
    const int e10 = std::atoi(s.c_str() + epos + 1);
    const int nd = (int)digits.size();
    std::string out = neg ? "-" : "";
    if (e10 >= nd - 1) {
        out += digits;
        out.append((size_t)(e10 - (nd - 1)), '0');
    } else if (e10 >= 0) {
        out += digits.substr(0, (size_t)e10 + 1);
        out.push_back('.');
        out += digits.substr((size_t)e10 + 1);
    } else {
        out += "0.";
        out.append((size_t)(-e10 - 1), '0');
        out += digits;
    }
    return out;
}

extern "C" int ps_fmt_f64(double v, char *buf, size_t cap)
{
    if (!buf || cap == 0) return ps_fail(PS_ERR_INVALID, "null buffer");
    const std::string s = fmt_f64(v);
    if (s.size() + 1 > cap) return ps_fail(PS_ERR_INVALID, "buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

// Population::write (population.rs:865-897)
extern "C" int ps_write(ps_population *p, const char *outpref)
{
    if (!p || !outpref) return ps_fail(PS_ERR_INVALID, "null argument");
    const uint64_t N = p->cfg.pop_size, C = p->cfg.ncols;
    const std::string path = std::string(outpref) + (p->cfg.core ? "_core_genome.csv" : "_pangenome.csv");
    FILE *f = fopen(path.c_str(), "w");
    if (!f) return ps_fail(PS_ERR_IO, "cannot create %s", path.c_str());
    if (p->cfg.core && C > 0) {
        // the text is expanded on the device, individuals in chunks of <= 256 MiB, two pinned
        // buffers so that the D2H copy of one chunk overlaps the fwrite of the previous one
        PSCHK(use_device(p));
        const uint64_t row_bytes = 2 * C;
        const uint32_t chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(N, (256ull << 20) / row_bytes));
        uint8_t *d_text = nullptr, *h_text[2] = { nullptr, nullptr };
        int rc = PS_OK;
        if (hipMalloc(&d_text, (uint64_t)chunk * row_bytes) != hipSuccess
            || hipHostMalloc(&h_text[0], (uint64_t)chunk * row_bytes) != hipSuccess
            || hipHostMalloc(&h_text[1], (uint64_t)chunk * row_bytes) != hipSuccess)
            rc = ps_fail(PS_ERR_OOM, "cannot allocate the text buffers of ps_write");
        uint64_t pending = 0;
        int pending_buf = -1, k = 0;
        const uint32_t *slot = nullptr;
        if (rc == PS_OK) rc = rows_current(p, &slot);
        for (uint64_t i0 = 0; rc == PS_OK && i0 < N; i0 += chunk, k ^= 1) {
            const uint32_t ni = (uint32_t)std::min<uint64_t>(chunk, N - i0);
            dim3 grid((uint32_t)((C + 63) / 64), (ni + 63) / 64);
            core_csv_kernel<<<grid, 256, 0, p->stream>>>(p->state, d_text, p->pitch, C, (uint32_t)i0, ni, (uint8_t)'\n', slot ? p->d_row_slot : nullptr);
            if (hipMemcpyAsync(h_text[k], d_text, (uint64_t)ni * row_bytes, hipMemcpyDeviceToHost, p->stream) != hipSuccess)
                rc = ps_fail(PS_ERR_NO_DEVICE, "D2H copy failed in ps_write");
            if (pending_buf >= 0 && fwrite(h_text[pending_buf], 1, pending, f) != pending)
                rc = ps_fail(PS_ERR_IO, "short write to %s", path.c_str());
            if (hipStreamSynchronize(p->stream) != hipSuccess && rc == PS_OK)
                rc = ps_fail(PS_ERR_NO_DEVICE, "device failure in ps_write");
            pending = (uint64_t)ni * row_bytes;
            pending_buf = k;
        }
        if (rc == PS_OK && pending_buf >= 0 && fwrite(h_text[pending_buf], 1, pending, f) != pending)
            rc = ps_fail(PS_ERR_IO, "short write to %s", path.c_str());
        if (d_text) (void)hipFree(d_text);
        for (auto *h : h_text) if (h) (void)hipHostFree(h);
        fclose(f);
        return rc;
    }
    std::vector<uint8_t> rows(std::max<uint64_t>(N * C, 1));
    {
        const int rc = ps_read_matrix(p, rows.data());
        if (rc != PS_OK) { fclose(f); return rc; }
    }
    std::string line;
    for (uint64_t i = 0; i < N; i++) {
        line.clear();
        const uint8_t *row = rows.data() + i * C;
        if (p->cfg.core) {
            line.reserve(2 * C + 1);
            for (uint64_t s = 0; s < C; s++) {
                if (s) line.push_back(',');
                line.push_back(ps_int_to_base(row[s]));
            }
        } else {
            bool first = true;
            for (uint64_t k = 0; k < p->cfg.core_genes; k++) {
                if (!first) line.push_back(',');
                line.push_back('1');
                first = false;
            }
            for (uint64_t g = 0; g < C; g++) {
                if (!first) line.push_back(',');
                line += std::to_string((unsigned)row[g]);
                first = false;
            }
        }
        line.push_back('\n');
        if (fwrite(line.data(), 1, line.size(), f) != line.size()) {
            fclose(f);
            return ps_fail(PS_ERR_IO, "short write to %s", path.c_str());
        }
    }
    fclose(f);
    return PS_OK;
}

// ---------------------------------------------------------------------------
// main() as a library (main.rs:155-553)
// ---------------------------------------------------------------------------
extern "C" void ps_sim_default_params(ps_sim_params *p)
{
    memset(p, 0, sizeof(*p));
    p->pop_size = 1000; p->core_size = 1200000; p->pan_genes = 6000; p->core_genes = 2000;
    p->avg_gene_freq = 0.5; p->HR_rate = 0.05; p->HGT_rate = 0.05; p->n_gen = 100;
    p->max_distances = 100000; p->core_mu = 0.05; p->rate_genes1 = 1.0; p->rate_genes2 = 1000.0;
    p->prop_genes2 = 0.1; p->prop_positive = -0.1; p->pos_lambda = 10.0; p->neg_lambda = 10.0;
    p->seed = 0; p->genome_size_penalty = 0.99; p->competition_strength = 0.0;
    p->shard_rank = 0; p->shard_count = 1; p->device = -1;
}

extern "C" int ps_sim_validate(const ps_sim_params *p, char *msg, size_t cap)
{
    std::string m;
    auto f = [](double v) { return fmt_f64(v); };
    if (p->core_genes > p->pan_genes) {                              // main.rs:195-198
        m = "core_genes must be less than or equal to pan_size\n";
    } else if (p->HR_rate < 0.0 || p->HGT_rate < 0.0) {              // :200-205
        m = "HR_rate and HGT_rate must be above 0.0\nHR_rate: " + f(p->HR_rate) + "\nHGT_rate: " + f(p->HGT_rate) + "\n";
    } else if (p->pos_lambda <= 0.0 || p->neg_lambda <= 0.0) {       // :207-212
        m = "pos_lambda and neg_lambda must be above 0.0\npos_lambda: " + f(p->pos_lambda) + "\nneg_lambda: "
            + f(p->neg_lambda) + "\n";
    } else if (p->rate_genes1 < 0.0 || p->rate_genes2 < 0.0) {       // :214-219
        m = "rate_genes1 and rate_genes2 must be >= 0\nrate_genes1: " + f(p->rate_genes1) + "\nrate_genes2: "
            + f(p->rate_genes2) + "\n";
    } else if (p->prop_genes2 < 0.0 || p->prop_genes2 > 1.0) {       // :221-225
        m = "prop_genes2 must be 0.0 <= prop_genes2 <= 1.0\nprop_genes2: " + f(p->prop_genes2) + "\n";
    } else if (p->pop_size < 1 || p->core_size < 1 || p->pan_genes < 1 || p->n_gen < 1
               || p->max_distances < 1) {                            // :227-235
        m = "pop_size, core_size, pan_genes, n_gen and max_distances must all be above 1\npop_size: "
            + std::to_string(p->pop_size) + "\ncore_size: " + std::to_string(p->core_size) + "\npan_genes: "
            + std::to_string(p->pan_genes) + "\nn_gen: " + std::to_string(p->n_gen) + "\nmax_distances: "
            + std::to_string(p->max_distances) + "\n";
    } else if (p->core_mu < 0.0 || p->core_mu > 1.0) {               // :237-241
        m = "core_mu must be between 0.0 and 1.0\ncore_mu: " + f(p->core_mu) + "\n";
    } else if (p->avg_gene_freq <= 0.0 || p->avg_gene_freq > 1.0) {  // :243-247
        m = "avg_gene_freq must be above 0.0 and below or equal to 1.0\navg_gene_freq: " + f(p->avg_gene_freq) + "\n";
    }
    if (msg && cap) {
        const size_t n = std::min(cap - 1, m.size());
        memcpy(msg, m.data(), n);
        msg[n] = 0;
    }
    return m.empty() ? PS_OK : PS_ERR_INVALID;
}

extern "C" int ps_sim_derive(const ps_sim_params *p, ps_derived *d)
{
    if (!p || !d) return ps_fail(PS_ERR_INVALID, "null argument");
    memset(d, 0, sizeof(*d));
    if (p->core_genes > p->pan_genes || p->pan_genes == 0) return ps_fail(PS_ERR_INVALID, "core_genes > pan_genes");
    const uint64_t pan_size = p->pan_genes - p->core_genes;          // main.rs:259
    d->pan_size = pan_size;
    const double core_prop = (double)p->core_genes / (double)p->pan_genes; // :263
    const double acc_prop = 1.0 - core_prop;                         // :264
    double agf = (p->avg_gene_freq - core_prop) / acc_prop;          // :265
    if (agf < 0.0) agf = 0.0;                                        // :266-268
    d->avg_gene_freq_adj = agf;
    d->avg_gene_num = (int32_t)std::round(agf * (double)pan_size);   // :272
    d->n_core_mutations = std::ceil((double)p->core_size * p->core_mu); // :275-276
    d->n_recombinations_core = std::round(d->n_core_mutations * p->HR_rate); // :279
    d->n_recombinations_pan_total = std::round(d->n_core_mutations * p->HGT_rate); // :280
    const uint64_t g1 = (uint64_t)std::round((double)pan_size * (1.0 - p->prop_genes2)); // :334
    const uint64_t g2 = pan_size - g1;                               // :335
    const double prop1 = (double)g1 / (double)pan_size;              // :336
    const double prop2 = 1.0 - prop1;                                // :337
    int c = 0;
    if (g1 > 0) {                                                    // :341-352
        d->comp_begin[c] = 0; d->comp_end[c] = g1;
        d->n_pan_mutations[c] = p->rate_genes1 * (double)g1;
        d->n_recombinations_pan[c] = d->n_recombinations_pan_total * prop1;
        c++;
    }
    if (g1 < pan_size) {                                             // :355-367
        d->comp_begin[c] = g1; d->comp_end[c] = pan_size;
        d->n_pan_mutations[c] = p->rate_genes2 * (double)g2;
        d->n_recombinations_pan[c] = d->n_recombinations_pan_total * prop2;
        c++;
    }
    d->n_comp = c;
    return PS_OK;
}

extern "C" int ps_selection_coefficients(uint64_t seed, uint64_t G, double prop_positive, double pos_lambda,
                                         double neg_lambda, double *out)
{
    if (!out && G) return ps_fail(PS_ERR_INVALID, "null output");
    for (uint64_t g = 0; g < G; g++) out[g] = 0.0;                   // main.rs:287
    if (!(prop_positive >= 0.0)) return PS_OK;                       // :292
    uint64_t n = 0;
    for (uint64_t g = 0; g < G; g++) {
        const double weight = hs_f64(seed, PS_STREAM_SELECTION, 0, n++);
        double s;
        if (weight <= prop_positive) {                               // :303
            s = -std::log(1.0 - hs_f64(seed, PS_STREAM_SELECTION, 0, n++)) / pos_lambda;
        } else {
            s = -std::log(1.0 - hs_f64(seed, PS_STREAM_SELECTION, 0, n++)) / neg_lambda;
            while (s > 1.0)                                          // :309-311
                s = -std::log(1.0 - hs_f64(seed, PS_STREAM_SELECTION, 0, n++)) / neg_lambda;
            s = -1.0 * s;                                            // :315
        }
        out[g] = s;
    }
    return PS_OK;
}

// ---------------------------------------------------------------------------
// SURVEY 8f-3, opt-in and UNPINNED: the reference's own seeded stream for what precedes sample_beta (main.rs:289-319).
// `StdRng::seed_from_u64(seed)` (rand 0.8.5): the u64 is expanded to a 32-byte key by PCG32 steps (rand_core 0.6),
// StdRng = ChaCha12 (rand_chacha 0.3: 64-bit block counter in words 12-13, stream id 0 in words 14-15, four blocks
// per refill of a 64-word buffer, next_u64 = two consecutive words, low word first).  `Uniform::new(0.0, 1.0)` takes the
// top 52 bits of a u64 as the mantissa of [1, 2) and subtracts 1; statrs 0.16 `Exp::sample` is ziggurat exp(1) / rate
// with the 256-layer tables of rand's utils/ziggurat_tables.py (recomputed here with libm, as that script does), `gen::<f64>`
// = top 53 bits x 2^-53.  All of this is restated from the published algorithms of un-vendored crates that cannot be
// built or read here: nothing pins it against a Pansim binary.  Everything from sample_beta on (main.rs:370: a rejection
// sampler with a data-dependent number of draws) stays on the build's own streams.
// ---------------------------------------------------------------------------
static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

extern "C" void ps_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16])
{
    uint32_t st[16] = { 0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6],
                        key[7], (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32) };
    uint32_t x[16];
    memcpy(x, st, sizeof x);
    auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
    };
    for (int r = 0; r < rounds; r += 2) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int k = 0; k < 16; k++) out[k] = x[k] + st[k];
}

struct ref_std_rng {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t buf[64];
    uint32_t index = 64;
    explicit ref_std_rng(uint64_t state)               // SeedableRng::seed_from_u64
    {
        for (int k = 0; k < 8; k++) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
            key[k] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
    }
    void refill(uint32_t index_after)
    {
        for (int b = 0; b < 4; b++) ps_chacha_block(key, counter + (uint64_t)b, 0, 12, buf + 16 * b);
        counter += 4;
        index = index_after;
    }
    uint64_t next_u64()                                 // BlockRng::next_u64
    {
        if (index < 63) { const uint64_t v = ((uint64_t)buf[index + 1] << 32) | buf[index]; index += 2; return v; }
        if (index >= 64) { refill(2); return ((uint64_t)buf[1] << 32) | buf[0]; }
        const uint64_t lo = buf[63];
        refill(1);
        return ((uint64_t)buf[0] << 32) | lo;
    }
    double gen_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }                 // Standard
    double uniform01()                                                                                   // Uniform::new(0.0, 1.0)
    {
        const uint64_t bits = (next_u64() >> 12) | 0x3FF0000000000000ull;
        double v;
        memcpy(&v, &bits, 8);
        return (v - 1.0) * 1.0 + 0.0;
    }
};

static void zig_exp_tables(double *x, double *f)        // rand utils/ziggurat_tables.py, exponential, 256 layers
{
    const double R = 7.69711747013104972, V = 0.0039496598225815571993;
    x[0] = V / std::exp(-R);
    x[1] = R;
    for (int i = 2; i < 256; i++) x[i] = -std::log(V / x[i - 1] + std::exp(-x[i - 1]));
    x[256] = 0.0;
    for (int i = 0; i <= 256; i++) f[i] = std::exp(-x[i]);
}

static double zig_exp_1(ref_std_rng &rng, const double *xt, const double *ft)     // statrs ziggurat::sample_exp_1
{
    for (;;) {
        const uint64_t bits = rng.next_u64();
        const unsigned i = (unsigned)(bits & 0xff);
        const double u = (double)(bits >> 11) / 9007199254740992.0;
        const double x = u * xt[i];
        if (x < xt[i + 1]) return x;
        if (i == 0) return 7.69711747013104972 - std::log(rng.gen_f64());
        if (ft[i + 1] + (ft[i] - ft[i + 1]) * rng.gen_f64() < std::exp(-x)) return x;
    }
}

extern "C" int ps_reference_selection_coefficients(uint64_t seed, uint64_t G, double prop_positive, double pos_lambda,
                                                   double neg_lambda, double *out)
{
    if (!out && G) return ps_fail(PS_ERR_INVALID, "null output");
    for (uint64_t g = 0; g < G; g++) out[g] = 0.0;                   // main.rs:287
    if (!(prop_positive >= 0.0)) return PS_OK;                       // :292
    if (!(pos_lambda > 0.0) || !(neg_lambda > 0.0)) return ps_fail(PS_ERR_INVALID, "Exp::new(lambda).unwrap() panics for lambda <= 0");
    static double xt[257], ft[257];
    static std::once_flag once;
    std::call_once(once, [] { zig_exp_tables(xt, ft); });
    ref_std_rng rng(seed);                                            // :289
    for (uint64_t g = 0; g < G; g++) {
        const double weight = rng.uniform01();                        // :298
        double s;
        if (weight <= prop_positive) {                                // :303
            s = zig_exp_1(rng, xt, ft) / pos_lambda;
        } else {
            s = zig_exp_1(rng, xt, ft) / neg_lambda;
            while (s > 1.0) s = zig_exp_1(rng, xt, ft) / neg_lambda;  // :309-311
            s = -1.0 * s;                                             // :315
        }
        out[g] = s;
    }
    return PS_OK;
}

extern "C" int ps_sample_pairs(uint64_t seed, uint64_t N, uint64_t P, uint32_t *range1, uint32_t *range2)
{
    if (!range1 || !range2) return ps_fail(PS_ERR_INVALID, "null output");
    if (N < 2) return ps_fail(PS_ERR_INVALID, "pair sampling needs pop_size >= 2 (main.rs:421 panics)");
    for (uint64_t k = 0; k < P; k++)                                 // main.rs:413-415
        range1[k] = ps_mulhi(hs_u32(seed, PS_STREAM_PAIRS, 0, k), (uint32_t)N);
    for (uint64_t k = 0; k < P; k++) {                               // :420-427
        uint32_t e = ps_mulhi(hs_u32(seed, PS_STREAM_PAIRS, 1, k), (uint32_t)(N - 1));
        if (e >= range1[k]) e += 1;
        range2[k] = e;
    }
    return PS_OK;
}

// ---------------------------------------------------------------------------
// generation loop (main.rs:429-464)
// ---------------------------------------------------------------------------
#define PS_RING 4
struct ps_sim {
    ps_sim_params prm{};
    ps_derived der{};
    ps_population *core = nullptr, *acc = nullptr;
    std::vector<double> sel;
    std::vector<uint32_t> r1, r2;
    // ring of parent-index slots so that the accessory chain (and the host) can run
    // ahead of the long core sweep
    uint32_t *d_idx[PS_RING] = {};
    uint32_t *h_idx[PS_RING] = {};         // pinned, host-mapped
    uint32_t *m_idx[PS_RING] = {};         // device alias of h_idx
    hipEvent_t ev_idx[PS_RING] = {}, ev_core[PS_RING] = {};
    hipEvent_t ev_hgt = nullptr;
    hipEvent_t ev_gap[PS_RING][2] = {};   // timestamped events around the sweep when timing is off
    std::vector<uint32_t> h_kids;       // children per parent (host counting sort of the drawn parents)
    // The draws of each ring slot's generation IN DRAW ORDER (parents named by their internal row): the reference's child k
    // is the child of draw k (population.rs:443, main.rs:445-447), the engine stores the children in ascending parent order
    // -- a stable counting sort of the draws -- so row k of every output is the internal row rank(k) (sim_refresh_rows)
    uint32_t *h_draw[PS_RING] = {};        // pinned, host-mapped (the device draw writes it)
    uint32_t *m_draw[PS_RING] = {};        // device alias of h_draw
    int prev_slot = -1;                    // ring slot of the generation before the last one (-1: none)
    std::vector<uint32_t> sigma, sigma_prev_inv;   // last generation: output row -> internal row; the one before: internal row -> output row
    uint64_t sigma_step = ~0ull;           // step_count the two were computed for
    bool heavy_hgt = false;             // expected HGT events per generation >= 1e7: HGT and sweep take turns
    bool slot_used[PS_RING] = {};
    int32_t *h_num_genes = nullptr, *m_num_genes = nullptr;   // pinned + its device alias
    double *h_logw = nullptr, *m_logw = nullptr, *h_avg = nullptr;
    double *d_avg = nullptr;
    // D-avg of the NEXT generation computed ahead of the sweep (sim_one_generation): valid while the accessory matrix is the
    // one it was computed from
    bool avg_prefetched = false;
    uint64_t avg_epoch = 0;
    hipEvent_t ev_avg = nullptr;
    std::vector<double> h_w;         // the weights of the generation being drawn (kept: no allocation per generation)
    hipEvent_t ev_bin = nullptr;     // the bin pass of the binned HGT is complete (the LDS-image pass on the core stream waits for it)
    double *d_log1p = nullptr;          // ln(1 + s_g) of THIS run (the accessory handle's own table belongs to its Population API)
    uint64_t step_count = 0;
    bool need_logw = false;
    // P-draw on the device (populations of >= 4096: the N binary searches over the cumulative table are the
    // largest part of the host half there, and the host half does not shrink with the number of site shards)
    bool device_draw = false;
    double *h_cum = nullptr, *d_cum = nullptr;   // cumulative weights: pinned host copy, device copy
    uint32_t *d_idx_tsum = nullptr;              // ... and the totals / prefixes of its 1024-parent tiles
    uint32_t *d_idx_cnt = nullptr;               // device draw: children per parent, then their inclusive prefix sums (counting sort)
    int last_slot = 0;
    // distance phase (ps_sim_pairwise_distances): pinned numerators, events around the kernels of each matrix
    uint32_t *h_cnt = nullptr;           // 3 x P: core numerators | accessory intersections | unions
    uint64_t h_cnt_cap = 0;
    hipEvent_t ev_dist[4] = {};          // core begin / end, accessory begin / end
    double dist_core_ms = 0.0, dist_acc_ms = 0.0;
    // host half of a generation, accumulated since the last reset (ps_sim_host_timing)
    uint64_t host_calls = 0;
    double host_wait_ms = 0.0, host_weights_ms = 0.0, host_draw_ms = 0.0;
    // several shards in one process: the weights of sample_indices computed once (ps_multi)
    int (*weights_hook)(void *ctx, ps_sim *s, uint32_t gen, double *w) = nullptr;
    void *weights_ctx = nullptr;
    // exchange of the donor-sharded HGT deltas: emulation (bench.py --emulate-shard) and traffic counters
    int emu_shards = 0;
    void *emu_buf = nullptr;
    uint64_t emu_cap = 0;
    uint32_t emu_clock_khz = 0;                        // wall_clock64() rate; 0 = not yet read
    double emu_link_gbps = 153.0, emu_latency_us = 20.0, emu_modelled_us = 0.0;
    uint64_t exchange_calls = 0, exchange_bytes = 0;   // bytes this rank SENDS + RECEIVES in the exchange (providers add to it)
    // sweep timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tev;
    std::vector<hipEvent_t> ev_pool;
};

extern "C" void ps_sim_destroy(ps_sim *s)
{
    if (!s) return;
    if (s->core) (void)hipSetDevice(s->core->device);
    if (s->core) (void)hipStreamSynchronize(s->core->stream);
    if (s->acc) (void)hipStreamSynchronize(s->acc->stream);
    for (int k = 0; k < PS_RING; k++) {
        if (s->d_idx[k]) (void)hipFree(s->d_idx[k]);
        if (s->h_idx[k]) (void)hipHostFree(s->h_idx[k]);
        if (s->h_draw[k]) (void)hipHostFree(s->h_draw[k]);
        if (s->ev_idx[k]) (void)hipEventDestroy(s->ev_idx[k]);
        if (s->ev_core[k]) (void)hipEventDestroy(s->ev_core[k]);
        if (k == 0 && s->ev_hgt) (void)hipEventDestroy(s->ev_hgt);
        if (k == 0 && s->ev_avg) (void)hipEventDestroy(s->ev_avg);
        if (k == 0 && s->ev_bin) (void)hipEventDestroy(s->ev_bin);
        for (int j = 0; j < 2; j++) if (s->ev_gap[k][j]) (void)hipEventDestroy(s->ev_gap[k][j]);
    }
    if (s->h_num_genes) (void)hipHostFree(s->h_num_genes);
    if (s->h_logw) (void)hipHostFree(s->h_logw);
    if (s->h_avg) (void)hipHostFree(s->h_avg);
    if (s->h_cum) (void)hipHostFree(s->h_cum);
    if (s->d_cum) (void)hipFree(s->d_cum);
    if (s->d_idx_cnt) (void)hipFree(s->d_idx_cnt);
    if (s->d_idx_tsum) (void)hipFree(s->d_idx_tsum);
    if (s->h_cnt) (void)hipHostFree(s->h_cnt);
    for (auto e : s->ev_dist) if (e) (void)hipEventDestroy(e);
    if (s->d_avg) (void)hipFree(s->d_avg);
    if (s->emu_buf) (void)hipFree(s->emu_buf);
    if (s->d_log1p) (void)hipFree(s->d_log1p);
    for (auto &pr : s->tev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto e : s->ev_pool) (void)hipEventDestroy(e);
    ps_population_destroy(s->core);
    ps_population_destroy(s->acc);
    delete s;
}

static int sim_refresh_rows(void *ctx);

static int sim_create_impl(const ps_sim_params *p, ps_sim *s)
{
    char msg[1024];
    if (ps_sim_validate(p, msg, sizeof msg) != PS_OK) return ps_fail(PS_ERR_INVALID, "%s", msg);
    if (p->shard_count < 1 || p->shard_rank < 0 || p->shard_rank >= p->shard_count)
        return ps_fail(PS_ERR_INVALID, "bad shard %d of %d", p->shard_rank, p->shard_count);
    if (p->pop_size < 2) return ps_fail(PS_ERR_INVALID, "pop_size must be >= 2 (main.rs:421 / population.rs:584 panic)");
    s->prm = *p;
    PSCHK(ps_sim_derive(p, &s->der));
    const ps_derived &d = s->der;
    const uint64_t N = p->pop_size, L = p->core_size, G = d.pan_size;
    s->sel.assign(std::max<uint64_t>(G, 1), 0.0);
    if (p->reference_seed_stream)     // opt-in, unpinned: the reference's own ChaCha12 stream for main.rs:289-319
        PSCHK(ps_reference_selection_coefficients(p->seed, G, p->prop_positive, p->pos_lambda, p->neg_lambda, s->sel.data()));
    else
        PSCHK(ps_selection_coefficients(p->seed, G, p->prop_positive, p->pos_lambda, p->neg_lambda, s->sel.data()));
    for (uint64_t g = 0; g < G; g++)
        if (std::log(1.0 + s->sel[g]) != 0.0) s->need_logw = true;
    // site shard of this process
    const uint64_t sb = L * (uint64_t)p->shard_rank / (uint64_t)p->shard_count;
    const uint64_t se = L * ((uint64_t)p->shard_rank + 1) / (uint64_t)p->shard_count;
    ps_config cc{};
    cc.pop_size = N; cc.ncols = se - sb; cc.global_cols = L; cc.col_offset = sb;
    cc.core_genes = p->core_genes; cc.seed = p->seed; cc.core = 1; cc.device = p->device;
    {
        std::vector<uint8_t> v(std::max<uint64_t>(cc.ncols, 1));
        PSCHK(ps_init_vector(p->seed, 1, sb, cc.ncols, 0.0, v.data()));
        PSCHK(ps_population_create(&cc, v.data(), &s->core));       // main.rs:372-381
    }
    ps_config ca{};
    ca.pop_size = N; ca.ncols = G; ca.global_cols = G; ca.col_offset = 0;
    ca.core_genes = p->core_genes; ca.seed = p->seed; ca.core = 0; ca.device = s->core->device;
    {
        std::vector<uint8_t> v(std::max<uint64_t>(G, 1));
        PSCHK(ps_init_vector(p->seed, 0, 0, G, d.avg_gene_freq_adj, v.data()));
        PSCHK(ps_population_create(&ca, v.data(), &s->acc));        // main.rs:382-391
    }
    // rows of every output of the two handles in the reference's order (the child of draw k), whatever the engine's own
    s->core->rows_refresh = s->acc->rows_refresh = sim_refresh_rows;
    s->core->rows_ctx = s->acc->rows_ctx = s;
    {
        const uint64_t b0 = 0, e0 = L;
        const double lm = d.n_core_mutations;
        const double lr = (p->HR_rate > 0.0) ? d.n_recombinations_core : 0.0;   // main.rs:459
        PSCHK(ps_set_rates(s->core, 1, &lm, &lr, &b0, &e0));
        double lrec[2] = { 0, 0 };
        for (int c = 0; c < d.n_comp; c++) lrec[c] = (p->HGT_rate > 0.0) ? d.n_recombinations_pan[c] : 0.0; // :462
        PSCHK(ps_set_rates(s->acc, d.n_comp, d.n_pan_mutations, lrec, d.comp_begin, d.comp_end));
    }
    s->r1.resize(p->max_distances);
    s->r2.resize(p->max_distances);
    PSCHK(ps_sample_pairs(p->seed, N, p->max_distances, s->r1.data(), s->r2.data())); // main.rs:413-427
    PSCHK(use_device(s->core));
    for (int k = 0; k < PS_RING; k++) {
        HIPCHK(hipMalloc(&s->d_idx[k], N * sizeof(uint32_t)));
        HIPCHK(hipHostMalloc(&s->h_idx[k], N * sizeof(uint32_t), hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void **)&s->m_idx[k], s->h_idx[k], 0));
        HIPCHK(hipHostMalloc(&s->h_draw[k], N * sizeof(uint32_t), hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void **)&s->m_draw[k], s->h_draw[k], 0));
        HIPCHK(hipEventCreateWithFlags(&s->ev_idx[k], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&s->ev_core[k], hipEventDisableTiming));
    }
    // the pair list is fixed for the whole run (main.rs:413-427): sort and upload it now, so that
    // the distance phase starts with the device copy in place
    if (p->max_distances > 0) {
        PSCHK(upload_pairs(s->core, p->max_distances, s->r1.data(), s->r2.data(), true));
        PSCHK(upload_pairs(s->acc, p->max_distances, s->r1.data(), s->r2.data(), true));
    }
    // a block sweep that fills the CU's LDS (cfg4 population) leaves none for the donor lists
    if (!wave_sweep_eligible(s->core, true, p->HR_rate > 0.0)) {
        core_block_geom g{};
        uint32_t lds = 0, nw = 0;
        if (block_sweep_geometry(s->core, true, true, p->HR_rate > 0.0, &g, &lds, &nw))
            s->acc->hgt_list_in_global = lds + 16384u > s->core->lds_limit;
    }
    // light HGT co-runs with the sweep: launch it narrow when the sweep is long enough to hide it.
    // The chain of a generation is ~0.3 ms of latency-bound work plus ~1 us per event a thread handles
    // in sequence; it is given half of what the sweep (estimated at 4.2 TB/s) leaves, at most 112
    // events per thread (cfg2, generations/s at 32 / 64 / 128 / 192 / 256 / 384 events per thread:
    // 1690 / 1736 / 1781 / 1775 / 1727 / 1480).  A short sweep leaves the accessory chain critical,
    // which wants the whole chip (0).  An adaptive controller (widen when the core stream is found
    // idle) was tried and lost to its own overshoots.
    {
        const double sweep_ms = 2.0 * (double)N * (double)s->core->cfg.ncols / 4.2e12 * 1e3;
        const double ept = (sweep_ms - 0.3) / 1.0e-3 * 0.5;
        s->acc->hgt_events_per_thread = ept >= 16.0 ? (uint32_t)std::min(ept, 112.0) : 0u;   // (at 7 sweep blocks per CU: 96 / 112 / 128 / 160 / 192 -> 1785 / 1815 / 1821 / 1826 / 1700)
        if (const char *e = getenv("PANSIM_HGT_EVENTS_PER_THREAD")) s->acc->hgt_events_per_thread = (uint32_t)atoi(e);
    }
    HIPCHK(hipEventCreateWithFlags(&s->ev_hgt, hipEventDisableTiming));
    {
        // take turns where the binned HGT kernels are used (>= 1e7 expected events): their LDS images
        // could not share a CU with the sweep anyway
        const uint32_t parts = hgt_partitions(s->acc);
        // (N = 1000, 6000 genes, by HGT_rate -- events per generation: 0.1 -- 6e6 -- 1906 beside the sweep / 1882 in turns;
        // 0.15 -- 9e6 -- 1567 / 1653; 0.2 -- 1.2e7 -- 1728 / 1884 generations/s: profiles/r05_sweep_experiments.md 11)
        s->heavy_hgt = (double)N * s->der.n_recombinations_pan_total >= 7.5e6 && parts >= 1 && parts <= 1024;
    }
    if (const char *e = getenv("PANSIM_HEAVY_HGT")) s->heavy_hgt = atoi(e) != 0;
    // Inside the generation loop the sweep shares the GPU with the accessory chain of the next
    // generation: 7 resident workgroups per CU leave that chain 4 wave slots and 20 KB of LDS on
    // every CU (the sweep's rows are assigned dynamically, so lower residency costs no tail).
    // cfg2, with the narrow one-wave HGT kernel: 8 -> chain starves (1285), 7 -> 1830, 6 -> 1808,
    // 5 -> 1680 generations/s (before that kernel 6 was the optimum: 7 -> 1490, 6 -> 1620-1690).
    // (When HGT and sweep take turns, 8 was best while the whole HGT waited for the sweep: 1292 / 1322 / 1352
    // at 6 / 7 / 8; with the bin pass running beside the sweep's tail, 7: 1388 / 1411 / 1386.)
    // A large accessory genome needs more room: the HGT kernel's donor list (2 bytes per gene of the
    // largest compartment) must fit the LDS the sweep leaves (pan_genes 20000: 1463 / 1664 / 1684 at 7 / 6 / 5).
    if (!getenv("PANSIM_SWEEP_BLOCKS_PER_CU")) {
        uint64_t max_comp = 0;
        for (int c = 0; c < d.n_comp; c++) max_comp = std::max<uint64_t>(max_comp, d.comp_end[c] - d.comp_begin[c]);
        const uint64_t list_lds = 2 * max_comp;
        if (list_lds > 36 * 1024) s->acc->hgt_list_in_global = true;
        // Round 5 (scan push, leaner dense pass; profiles/r05_sweep_experiments.md 11): the sweep by itself is fastest at 5-6
        // workgroups per CU (0.475 against 0.505 ms at 7, alone on the GPU), and the chain beside it decides: light HGT
        // (cfg2) 4 / 5 / 6 / 7 -> 1993 / 2080 / 2093 / 1972 generations/s; with the D-avg of --competition_strength in the
        // chain 1957 / 2007 / 1921 / 1848 (the authors' run 1942 / 1897 / 1841 / 1692); sweep and HGT in turns (cfg3)
        // - / 1537 / 1617 / 1645.
        // Round 6 (bit-sliced level 1, symbol-decided mutations in registers, 4-row batches; profiles/r06_sweep_experiments.md
        // 5): the sweep is bound by its access pattern, and the fewer of its workgroups sit on a CU the more of the chain runs beside
        // it -- cfg2 3 / 4 / 5 -> 2070 / 2121 / 2093 generations/s (6 / 7 / 8 on another box: 1843 / 1841 / 1829 against 1966 at 5);
        // cfg3 (sweep and HGT in turns) 3 / 4 / 5 / 6 / 7 -> 1693 / 1871 / 1836 / 1700 / 1725; with the D-avg of
        // --competition_strength in the chain 3 / 4 / 5 / 6 -> 2027 / 2107 / 1957 / 1786; the authors' run 1972 / 1954 at 4 / 5.
        const uint32_t light = 4u;
        s->core->sweep_blocks_per_cu = list_lds <= 16 * 1024 ? light : std::min(light, 6u);
    }
    HIPCHK(hipHostMalloc(&s->h_num_genes, N * sizeof(int32_t), hipHostMallocMapped));
    HIPCHK(hipHostMalloc(&s->h_logw, N * sizeof(double), hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer((void **)&s->m_num_genes, s->h_num_genes, 0));
    HIPCHK(hipHostGetDevicePointer((void **)&s->m_logw, s->h_logw, 0));
    HIPCHK(hipHostMalloc(&s->h_avg, N * sizeof(double)));
    s->device_draw = N >= 4096;
    if (const char *e = getenv("PANSIM_DEVICE_DRAW")) s->device_draw = atoi(e) != 0;
    HIPCHK(hipHostMalloc(&s->h_cum, N * sizeof(double)));
    HIPCHK(hipMalloc(&s->d_cum, N * sizeof(double)));
    HIPCHK(hipMalloc(&s->d_idx_tsum, ((N + 1023) / 1024 + 1) * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&s->d_idx_cnt, N * sizeof(uint32_t)));
    HIPCHK(hipMemset(s->d_idx_cnt, 0, N * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&s->d_avg, N * sizeof(double)));
    if (G) {
        std::vector<double> l1p(G);
        for (uint64_t g = 0; g < G; g++) l1p[g] = std::log(1.0 + s->sel[g] * 1.0);
        HIPCHK(hipMalloc(&s->d_log1p, G * sizeof(double)));
        HIPCHK(hipMemcpy(s->d_log1p, l1p.data(), G * sizeof(double), hipMemcpyHostToDevice));
    }
    // neutral selection: the reduce pass of a binned HGT leaves the gene counts the next generation's host half reads
    // (PANSIM_FUSE_COUNTS=0: the separate kernel, for A/B runs)
    if (G && !s->need_logw && !(getenv("PANSIM_FUSE_COUNTS") && atoi(getenv("PANSIM_FUSE_COUNTS")) == 0)) {
        s->acc->fuse_counts_out = s->m_num_genes;
        s->acc->fuse_logw_out = s->m_logw;
    }
    return PS_OK;
}

extern "C" int ps_sim_create(const ps_sim_params *p, ps_sim **out)
{
    if (!p || !out) return ps_fail(PS_ERR_INVALID, "null argument");
    *out = nullptr;
    ps_sim *s = new ps_sim();
    const int rc = sim_create_impl(p, s);
    if (rc != PS_OK) {
        const std::string keep = g_err;
        ps_sim_destroy(s);
        g_err = keep;
        return rc;
    }
    *out = s;
    return PS_OK;
}

// D-avg of a generation (main.rs:438-440) into s->d_avg on the accessory stream.  In a run whose shards exchange (the
// donor-sharded HGT's hook is installed) the rows are sharded the same way: this shard computes the rows of ITS individuals
// [N r / K, N (r + 1) / K) -- every individual's mean is an ordered sum over all others, so rows are independent -- into a
// zeroed vector, and the OR exchange of the N doubles' bit patterns is the all-gather (disjoint slices, zeros elsewhere).
static int sim_average_distance(ps_sim *s)
{
    ps_population *acc = s->acc;
    const uint64_t N = s->prm.pop_size;
    hipStream_t sa = acc->stream;
    const bool sharded = acc->exchange && acc->donor_cnt != 0 && acc->d.G > 0 && N >= 2;
    if (!sharded) return average_distance_device(acc, s->d_avg, sa);
    HIPCHK(hipMemsetAsync(s->d_avg, 0, N * sizeof(double), sa));
    PSCHK(average_distance_device(acc, s->d_avg, sa, acc->donor_lo, acc->donor_cnt));
    const std::string prev = g_err;
    g_err.clear();
    const int rc = acc->exchange(acc->exchange_ctx, s->d_avg, N, (void *)sa);
    if (rc != 0) {
        const std::string inner = g_err;
        return ps_fail(PS_ERR_STATE, "the all-gather of the average distances failed (%d)%s%s", rc, inner.empty() ? "" : ": ", inner.c_str());
    }
    g_err = prev;
    return PS_OK;
}

// main.rs:435-443 up to the weights: D-avg when competition is on, the device half of sample_indices (gene counts and
// log-fitness per individual, written straight into host-mapped memory), then the three softmaxes on the host
static int sim_host_weights(ps_sim *s, uint32_t gen, double *w, bool avg_ready = false)
{
    (void)gen;
    ps_population *acc = s->acc;
    const ps_sim_params &p = s->prm;
    const uint64_t N = p.pop_size;
    hipStream_t sa = acc->stream;
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    for (uint64_t i = 0; i < N; i++) s->h_avg[i] = 1.0;                 // main.rs:435
    if (p.competition_strength > 0.0) {                                  // :438-440
        const bool ahead = s->avg_prefetched && s->avg_epoch == acc->edit_epoch;       // (computed before the previous sweep was launched)
        if (!avg_ready && !ahead) PSCHK(sim_average_distance(s));
        s->avg_prefetched = false;
        HIPCHK(hipMemcpyAsync(s->h_avg, s->d_avg, N * sizeof(double), hipMemcpyDeviceToHost, sa));
    }
    if (s->need_logw)
        acc_fitness_kernel<<<(uint32_t)((N + 255) / 256), 256, 0, sa>>>(acc->I[acc->cur], s->d_log1p, 1,
                                                                     s->m_num_genes, s->m_logw, acc->d);
    else if (!acc->counts_fresh)         // (else: left by the reduce pass of the previous generation's HGT)
        acc_gene_count_rows_kernel<<<(uint32_t)((N + 3) / 4), 256, 0, sa>>>(acc->I[acc->cur], s->m_num_genes,
                                                                         s->m_logw, acc->d);
    HIPCHK(hipGetLastError());
    auto th0 = clk::now();
    HIPCHK(hipStreamSynchronize(sa));
    s->host_wait_ms += ms_since(th0);
    th0 = clk::now();
    PSCHK(ps_sample_weights(s->h_num_genes, s->h_logw, N, acc->cfg.ncols, s->der.avg_gene_num, s->h_avg,
                            p.no_control_genome_size, p.genome_size_penalty, p.competition_strength, w));
    s->host_weights_ms += ms_since(th0);
    return PS_OK;
}

static int sim_one_generation(ps_sim *s, uint32_t gen)
{
    ps_population *core = s->core, *acc = s->acc;
    const ps_sim_params &p = s->prm;
    const uint64_t N = p.pop_size, G = acc->cfg.ncols;
    hipStream_t sa = acc->stream, sc = core->stream;
    const int slot = (int)(s->step_count % PS_RING);
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    // the slot's previous core sweep must have consumed its indices
    auto th0 = clk::now();
    if (s->slot_used[slot]) HIPCHK(hipEventSynchronize(s->ev_core[slot]));
    s->host_wait_ms += ms_since(th0);
    // main.rs:435-443: the weights of sample_indices -- from this shard's own replica of the accessory matrix, or (several
    // shards in one process) computed once by shard 0 and handed to the others, whose replicas are bit-identical
    std::vector<double> &w = s->h_w;
    w.resize(N);
    if (s->weights_hook) PSCHK(s->weights_hook(s->weights_ctx, s, gen, w.data()));
    else PSCHK(sim_host_weights(s, gen, w.data()));
    th0 = clk::now();
    s->prev_slot = s->step_count ? s->last_slot : -1;
    s->last_slot = slot;
    core->rows_overridden = acc->rows_overridden = false;      // (rows in the order of THIS generation's draws from here on)
    // the step leaves the pre-recombination snapshot only for the light HGT form, the one that reads it (the binned form of
    // cfg3 / cfg4 / cfg5 does not: a third N x GW x 8-byte buffer written per generation with no reader)
    const bool want_snap = p.HGT_rate > 0.0 && !hgt_takes_binned_form(acc);
    if (s->device_draw) {
        // cumulative table on the host (sequential f64 sums, as WeightedIndex::new builds it), copied to the
        // device (8*N bytes; the 16 dependent reads of every search would otherwise cross PCIe: 155 us at
        // N = 65536); the kernel writes the parents both to device memory and to the host-mapped slot
        double total = w[0];
        for (uint64_t i = 1; i < N; i++) { s->h_cum[i - 1] = total; total += w[i]; }
        HIPCHK(hipMemcpyAsync(s->d_cum, s->h_cum, (N - 1) * sizeof(double), hipMemcpyHostToDevice, sa));
        // the N draws, then a counting sort: children are stored in ascending parent order
        // (the counts are zero: zeroed at creation, and by idx_zero_kernel after every use)
        acc_draw_parents_kernel<<<(uint32_t)((N + 255) / 256), 256, 0, sa>>>(s->d_cum, total, (uint32_t)N, (uint32_t)p.seed,
                                                                         (uint32_t)(p.seed >> 32), gen, s->d_idx_cnt, s->m_draw[slot]);
        {
            const uint32_t tiles = (uint32_t)((N + 1023) / 1024);
            idx_tile_sums_kernel<<<tiles, 256, 0, sa>>>(s->d_idx_cnt, (uint32_t)N, s->d_idx_tsum);
            idx_tile_prefix_kernel<<<1, 256, 0, sa>>>(s->d_idx_tsum, tiles);
            idx_scan_kernel<<<tiles, 256, 0, sa>>>(s->d_idx_cnt, (uint32_t)N, s->d_idx_tsum);
        }
        idx_fill_kernel<<<(uint32_t)((N + 255) / 256), 256, 0, sa>>>(s->d_idx_cnt, (uint32_t)N, s->d_idx[slot], s->m_idx[slot]);
        idx_zero_kernel<<<(uint32_t)((N + 255) / 256), 256, 0, sa>>>(s->d_idx_cnt, (uint32_t)N);
        HIPCHK(hipGetLastError());
        s->host_draw_ms += ms_since(th0);
        s->host_calls++;
        PSCHK(launch_acc_step(acc, s->d_idx[slot], gen, true, true, sa, nullptr, want_snap));
    } else {
    PSCHK(ps_draw_parents(w.data(), N, p.seed, gen, s->h_idx[slot]));
    memcpy(s->h_draw[slot], s->h_idx[slot], N * sizeof(uint32_t));      // (draw order: sim_refresh_rows, ps_sim_last_parents)
    {
        // children in ascending parent order: a counting sort, as on the device
        s->h_kids.assign(N, 0u);
        uint32_t *ix = s->h_idx[slot];
        for (uint64_t k = 0; k < N; k++) s->h_kids[ix[k]]++;
        uint64_t j = 0;
        for (uint64_t par = 0; par < N; par++)
            for (uint32_t c = s->h_kids[par]; c; c--) ix[j++] = (uint32_t)par;
    }
    s->host_draw_ms += ms_since(th0);
    s->host_calls++;
    // main.rs:447, :455, :462-464 on the accessory stream.  The gather kernel reads the parents
    // straight from the host-mapped slot (4*N bytes over PCIe) and publishes the device copy the
    // core sweep uses: no copy kernel has to fight the sweep for a CU.
    if (G == 0) HIPCHK(hipMemcpyAsync(s->d_idx[slot], s->h_idx[slot], N * sizeof(uint32_t), hipMemcpyHostToDevice, sa));
    PSCHK(launch_acc_step(acc, s->m_idx[slot], gen, true, true, sa, s->d_idx[slot], want_snap));
    }
    HIPCHK(hipEventRecord(s->ev_idx[slot], sa));
    // Heavy HGT (cfg3-like rates, >= 1e7 expected events): its scattered loads and the streaming
    // sweep slow each other down far more than their sum, so they take turns on the chip:
    // HGT(g) runs after sweep(g-1) has finished and before sweep(g) starts, and the host half of
    // generation g+1 hides behind sweep(g).
    const bool heavy_hgt = p.HGT_rate > 0.0 && s->heavy_hgt;
    if (heavy_hgt) {
        const int prev = (slot + PS_RING - 1) % PS_RING;
        // only the LDS-image passes (apply, reduce) wait for the previous sweep: the bin pass -- LDS-local
        // gathers, streaming appends -- runs beside its tail (cfg3: 1359 -> 1411 generations/s)
        HIPCHK(hipEventRecord(s->ev_hgt, sa));       // (recorded again after the LDS-image pass; this one covers an HGT that launches nothing)
        // (measured, not kept as the default: cfg3 exposed 0.092 / 0.094 against 0.087 / 0.092 ms, one rank of 8 at cfg4 0.69 / 0.63 against
        // 0.66 / 0.66 -- what sits between two sweeps is the bin pass finishing late, not the event latency; profiles/r05_sweep_experiments.md 8)
        static const bool gap_on_core = getenv("PANSIM_GAP_ON_CORE_STREAM") && atoi(getenv("PANSIM_GAP_ON_CORE_STREAM")) != 0;
        if (gap_on_core && !s->ev_bin) HIPCHK(hipEventCreateWithFlags(&s->ev_bin, hipEventDisableTiming));
        PSCHK(launch_acc_hgt(acc, gen, sa, s->slot_used[prev] ? s->ev_core[prev] : nullptr, s->ev_hgt, gap_on_core ? sc : nullptr, s->ev_bin));
        HIPCHK(hipStreamWaitEvent(sc, s->ev_hgt, 0));      // (a no-op when the event was recorded on sc itself)
    }

    // --competition_strength with a D-avg that cannot share a CU with the sweep (the matrix-core forms: wide populations and
    // every row shard): D-avg of generation g + 1 needs the accessory matrix after HGT(g) and nothing of sweep(g), and its
    // kernels would wait for sweep(g)'s END and head the whole chain of g + 1 behind it.  Run it NOW, ahead of sweep(g): the
    // two take turns (D-avg is matrix-core work, the sweep HBM traffic) and the host half, the draw, the gather and the bin
    // pass of g + 1 hide beside the sweep again (round 5: one rank of 8 at cfg4 + competition 10: 114 -> 141 generations/s; 41 in round 4).
    {
        const bool sharded_rows = acc->exchange && acc->donor_cnt != 0;
        const bool big_davg = acc->d.G > 0 && N >= 2 && (sharded_rows || N > 8192 || acc->davg_form >= 2) && acc->davg_form != 1;
        static const bool off = getenv("PANSIM_DAVG_AHEAD") && atoi(getenv("PANSIM_DAVG_AHEAD")) == 0;
        if (!off && p.competition_strength > 0.0 && big_davg && (heavy_hgt || p.HGT_rate <= 0.0) && (int64_t)gen + 1 != (int64_t)p.n_gen) {      // (not behind the run's last generation)
            PSCHK(sim_average_distance(s));
            s->avg_prefetched = true;
            s->avg_epoch = acc->edit_epoch;
            if (!s->ev_avg) HIPCHK(hipEventCreateWithFlags(&s->ev_avg, hipEventDisableTiming));
            HIPCHK(hipEventRecord(s->ev_avg, sa));
            HIPCHK(hipStreamWaitEvent(sc, s->ev_avg, 0));
        }
    }

    // main.rs:445, :452, :459-461 on the core stream, one fused pass
    HIPCHK(hipStreamWaitEvent(sc, s->ev_idx[slot], 0));
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (s->timing) {
        auto take = [&](hipEvent_t *e) -> int {
            if (!s->ev_pool.empty()) { *e = s->ev_pool.back(); s->ev_pool.pop_back(); return PS_OK; }
            HIPCHK(hipEventCreate(e));
            return PS_OK;
        };
        PSCHK(take(&t0));
        PSCHK(take(&t1));
        HIPCHK(hipEventRecord(t0, sc));
    } else {
        // Two timestamped events around every sweep even when nobody reads them (a ring of 2 per
        // slot): with them consecutive sweeps do not chain back to back and the accessory kernels of
        // the next generation get their CU slots sooner -- measured 1815 / 1806 generations/s with the
        // events against 1784 / 1754 without (cfg2, 300 generations).
        if (!s->ev_gap[slot][0]) {
            HIPCHK(hipEventCreate(&s->ev_gap[slot][0]));
            HIPCHK(hipEventCreate(&s->ev_gap[slot][1]));
        }
        HIPCHK(hipEventRecord(s->ev_gap[slot][0], sc));
    }
    PSCHK(step_device(core, s->d_idx[slot], gen, true, true, p.HR_rate > 0.0, sc, nullptr, true));
    if (s->timing) {
        HIPCHK(hipEventRecord(t1, sc));
        s->tev.emplace_back(t0, t1);
    } else {
        HIPCHK(hipEventRecord(s->ev_gap[slot][1], sc));
    }
    HIPCHK(hipEventRecord(s->ev_core[slot], sc));
    // HGT is enqueued after the sweep so that the sweep's launch is not queued behind it
    if (p.HGT_rate > 0.0 && !heavy_hgt) PSCHK(launch_acc_hgt(acc, gen, sa));
    s->slot_used[slot] = true;
    s->step_count++;
    return PS_OK;
}

// PANSIM_EXCHANGE_BESIDE_SWEEP=1 (experiment): a donor-sharded run lets sweep(g) start behind the LDS-image pass of HGT(g)
// like an unsharded one, with the reduce pass, the exchange and the merge BESIDE it -- for which the window sweep leaves a
// workgroup's worth of wave slots, registers and LDS per CU free (6 workgroups per CU instead of 7)
static void sim_exchange_schedule(ps_sim *s)
{
    const char *e = getenv("PANSIM_EXCHANGE_BESIDE_SWEEP");
    const bool on = e && atoi(e) != 0 && s->acc->exchange != nullptr;
    s->acc->exchange_beside_sweep = on;
    // A donor-sharded run has a longer chain between two sweeps (LDS-image pass, reduce, exchange, merge): the window sweep
    // leaves it more of every CU -- 5 workgroups per CU instead of 6 (round 6, one rank of 8 at cfg4: the sweep 3.60 against
    // 3.56 ms, the period 3.93 against 4.05 ms; profiles/r06_d_ab_cfg4_shard8.json)
    if (s->acc->exchange != nullptr && s->core->window_blocks_per_cu == 0 && !getenv("PANSIM_WINDOW_BPC")) s->core->window_blocks_per_cu = 5;
}

extern "C" int ps_sim_set_exchange(ps_sim *s, ps_exchange_fn fn, void *ctx)
{
    if (!s) return ps_fail(PS_ERR_INVALID, "null handle");
    if (s->prm.shard_count < 2) return ps_fail(PS_ERR_INVALID, "the HGT donors are sharded over the site shards of a run: shard_count must be >= 2");
    PSCHK(ps_set_donor_shard(s->acc, (uint32_t)s->prm.shard_rank, (uint32_t)s->prm.shard_count, fn, ctx));
    sim_exchange_schedule(s);
    return PS_OK;
}

// bench.py --emulate-shard K: this process plays shard 0 of K.  Its HGT serves donors [0, N / K) and the exchange is
// stood in for by (i) device-local copies of the volume a K-rank all-to-all + gather of the delta moves per rank
// (2 x (K - 1) / K of the buffer: the pack / unpack traffic in HBM) and (ii) a kernel of RCCL's shape that holds the stream
// for the time the two collectives would take on the links.  Both steps of the providers are DIRECT all-to-alls (round 5):
// one slice of buffer / K on each of the K - 1 point-to-point links at the same time, so a collective is charged a launch
// latency + (buffer / K) / the rate of one xGMI link (153 GB/s, 20 us; PANSIM_EMU_XGMI_GBPS / PANSIM_EMU_COLL_LATENCY_US
// override).  PANSIM_EMU_RING=1 prices round 4's ring all-gather instead ((K - 1) / K x buffer through ONE link).  A timing
// stand-in only: the other shards' events never arrive.
// The hold kernel has the FOOTPRINT of the kernel a real exchange runs: librccl's rcclGenericKernel on gfx950 is 256
// threads, 261-280 unified registers per lane and 19744 bytes of LDS (read from the code object of torch's librccl,
// RCCL 2.26.6) -- one wave per SIMD, and only on a SIMD that holds no sweep wave.  With a lighter stand-in the
// exchange-beside-the-sweep schedule (PANSIM_EXCHANGE_BESIDE_SWEEP) measured what RCCL's kernels cannot do.
__global__ void __launch_bounds__(256) emu_hold_kernel(unsigned long long ticks)
{
    __shared__ uint32_t pad[19744 / 4];
    pad[threadIdx.x] = threadIdx.x;
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a7, 0" ::: "v255", "a7");
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (pad[(threadIdx.x + 1u) & 255u] == 0xFFFFFFFFu) __builtin_trap();
}

static int emulated_exchange(void *ctx, void *d_words, uint64_t n_words, void *hip_stream)
{
    ps_sim *s = (ps_sim *)ctx;
    const uint64_t bytes = n_words * 8;
    if (s->emu_cap < bytes) {
        if (s->emu_buf) HIPCHK(hipFree(s->emu_buf));
        s->emu_buf = nullptr;
        s->emu_cap = 0;
        HIPCHK(hipMalloc(&s->emu_buf, bytes));
        s->emu_cap = bytes;
    }
    if (s->emu_clock_khz == 0) {
        int khz = 0, dev = 0;
        HIPCHK(hipGetDevice(&dev));
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
        s->emu_clock_khz = (uint32_t)khz;
        const char *e = getenv("PANSIM_EMU_XGMI_GBPS");
        if (e && atof(e) > 0.0) s->emu_link_gbps = atof(e);
        e = getenv("PANSIM_EMU_COLL_LATENCY_US");
        if (e && atof(e) >= 0.0) s->emu_latency_us = atof(e);
    }
    const uint64_t part = bytes / (uint64_t)s->emu_shards * (uint64_t)(s->emu_shards - 1);
    static const bool ring = getenv("PANSIM_EMU_RING") && atoi(getenv("PANSIM_EMU_RING")) != 0;
    const double slice = (double)bytes / (double)s->emu_shards;
    const double a2a_us = s->emu_latency_us + slice / (s->emu_link_gbps * 1e3);             // direct: one slice per link, all links at once
    const double gather_us = ring ? s->emu_latency_us + (double)part / (s->emu_link_gbps * 1e3) : a2a_us;
    const unsigned long long ticks = (unsigned long long)(a2a_us * 1e-6 * (double)s->emu_clock_khz * 1e3);
    const unsigned long long ticks2 = (unsigned long long)(gather_us * 1e-6 * (double)s->emu_clock_khz * 1e3);
    hipStream_t st = (hipStream_t)hip_stream;
    HIPCHK(hipMemcpyAsync(s->emu_buf, d_words, part, hipMemcpyDeviceToDevice, st));                        // "all-to-all"
    emu_hold_kernel<<<8, 256, 0, st>>>(ticks);
    HIPCHK(hipMemcpyAsync((uint8_t *)s->emu_buf + (bytes - part), (uint8_t *)d_words + (bytes - part), part,
                          hipMemcpyDeviceToDevice, st));                                                   // "all-gather"
    emu_hold_kernel<<<8, 256, 0, st>>>(ticks2);
    HIPCHK(hipGetLastError());
    if (d_words == (void *)s->d_avg) {       // (told apart by the buffer itself: an HGT delta of N words exists, pan_genes <= 64)
        // the row-sharded D-avg vector (sim_average_distance): the other shards' slices never arrive, and zeros there would
        // give 7/8 of the population the weight 0 -- a different simulation (every parent from shard 0's individuals, narrow
        // windows).  Stand-in values: shard 0's slice repeated, so that the run keeps the dynamics of the unsharded one.
        const uint64_t own = n_words / (uint64_t)s->emu_shards;
        for (uint64_t k = 1; own && k * own < n_words; k++)
            HIPCHK(hipMemcpyAsync((uint64_t *)d_words + k * own, d_words, std::min(own, n_words - k * own) * 8, hipMemcpyDeviceToDevice, st));
    }
    s->exchange_calls++;
    s->exchange_bytes += 2 * part;
    s->emu_modelled_us += a2a_us + gather_us;
    return PS_OK;
}

extern "C" int ps_sim_emulate_exchange(ps_sim *s, int n_shards)
{
    if (!s) return ps_fail(PS_ERR_INVALID, "null handle");
    if (n_shards < 2) return ps_fail(PS_ERR_INVALID, "n_shards must be >= 2");
    s->emu_shards = n_shards;
    PSCHK(ps_set_donor_shard(s->acc, 0u, (uint32_t)n_shards, emulated_exchange, s));
    sim_exchange_schedule(s);
    return PS_OK;
}

extern "C" int ps_sim_exchange_stats(ps_sim *s, int reset, uint64_t *calls, uint64_t *bytes)
{
    if (!s) return ps_fail(PS_ERR_INVALID, "null handle");
    if (calls) *calls = s->exchange_calls;
    if (bytes) *bytes = s->exchange_bytes;
    if (reset) s->exchange_calls = s->exchange_bytes = 0;
    return PS_OK;
}

extern "C" int ps_sim_emulated_link_time(ps_sim *s, int reset, double *modelled_us, double *link_gbps, double *latency_us)
{
    if (!s) return ps_fail(PS_ERR_INVALID, "null handle");
    if (modelled_us) *modelled_us = s->emu_modelled_us;
    if (link_gbps) *link_gbps = s->emu_link_gbps;
    if (latency_us) *latency_us = s->emu_latency_us;
    if (reset) s->emu_modelled_us = 0.0;
    return PS_OK;
}

extern "C" int ps_sim_run(ps_sim *s, uint32_t first_generation, uint32_t count)
{
    if (!s) return ps_fail(PS_ERR_INVALID, "null handle");
    PSCHK(use_device(s->core));
    for (uint32_t g = 0; g < count; g++) PSCHK(sim_one_generation(s, first_generation + g));
    return PS_OK;
}

extern "C" int ps_sim_sync(ps_sim *s)
{
    if (!s) return ps_fail(PS_ERR_INVALID, "null handle");
    PSCHK(use_device(s->core));
    HIPCHK(hipStreamSynchronize(s->acc->stream));
    HIPCHK(hipStreamSynchronize(s->core->stream));
    PSCHK(check_device_flag(s->acc));
    return check_device_flag(s->core);
}

extern "C" ps_population *ps_sim_core(ps_sim *s) { return s ? s->core : nullptr; }
extern "C" ps_population *ps_sim_acc(ps_sim *s) { return s ? s->acc : nullptr; }
extern "C" const double *ps_sim_selection(ps_sim *s) { return s ? s->sel.data() : nullptr; }
extern "C" const uint32_t *ps_sim_range1(ps_sim *s) { return s ? s->r1.data() : nullptr; }
extern "C" const uint32_t *ps_sim_range2(ps_sim *s) { return s ? s->r2.data() : nullptr; }

// rank[k] = position of draw k after a stable sort of the draws by parent = the internal row of the child of draw k
static void stable_rank(const uint32_t *draw, uint64_t N, std::vector<uint32_t> &start, uint32_t *rank)
{
    start.assign(N + 1, 0u);
    for (uint64_t k = 0; k < N; k++) start[draw[k] + 1]++;
    for (uint64_t i = 0; i < N; i++) start[i + 1] += start[i];
    for (uint64_t k = 0; k < N; k++) rank[k] = start[draw[k]]++;
}

// rows_refresh of the simulation's two handles: output row k = the child of draw k of the last generation
static int sim_refresh_rows(void *ctx)
{
    ps_sim *s = (ps_sim *)ctx;
    if (s->sigma_step == s->step_count) return PS_OK;
    const uint64_t N = s->prm.pop_size;
    PSCHK(use_device(s->core));
    HIPCHK(hipStreamSynchronize(s->acc->stream));       // (with the draw on the device the slot is written by a kernel of that stream)
    std::vector<uint32_t> start;
    s->sigma.clear();
    s->sigma_prev_inv.clear();
    if (s->step_count) {
        s->sigma.resize(N);
        stable_rank(s->h_draw[s->last_slot], N, start, s->sigma.data());
        if (s->prev_slot >= 0) {
            std::vector<uint32_t> sp(N);
            stable_rank(s->h_draw[s->prev_slot], N, start, sp.data());
            s->sigma_prev_inv.resize(N);
            for (uint64_t k = 0; k < N; k++) s->sigma_prev_inv[sp[k]] = (uint32_t)k;
        }
    }
    for (ps_population *p : { s->core, s->acc }) {
        p->row_slot = s->sigma;
        if (!s->sigma.empty()) {
            if (!p->d_row_slot) HIPCHK(hipMalloc(&p->d_row_slot, N * sizeof(uint32_t)));
            HIPCHK(hipMemcpyAsync(p->d_row_slot, s->sigma.data(), N * sizeof(uint32_t), hipMemcpyHostToDevice, p->stream));
            HIPCHK(hipStreamSynchronize(p->stream));
        }
    }
    s->sigma_step = s->step_count;
    return PS_OK;
}

// the draws of the last generation in draw order, every parent named by ITS output row of the generation before (the
// reference's sample_indices: population.rs:443) -- child k of this generation's outputs descends from row out_idx[k] of
// the previous generation's
extern "C" int ps_sim_last_parents(ps_sim *s, uint32_t *out_idx)
{
    if (!s || !out_idx) return ps_fail(PS_ERR_INVALID, "null argument");
    if (s->step_count == 0) { memset(out_idx, 0, s->prm.pop_size * sizeof(uint32_t)); return PS_OK; }
    PSCHK(sim_refresh_rows(s));
    const uint32_t *d = s->h_draw[s->last_slot];
    for (uint64_t k = 0; k < s->prm.pop_size; k++) out_idx[k] = s->sigma_prev_inv.empty() ? d[k] : s->sigma_prev_inv[d[k]];
    return PS_OK;
}

// main.rs:467-470 for this process's matrices: both distance kernels chains are enqueued before anything is
// waited for (core on the core stream, accessory on its own), the numerators come back through pinned memory,
// and the reference's f64 expressions run on the host.  With site shards the core numerators of this call
// cover the shard's sites only (ps_multi_pairwise_distances sums them).
static int sim_pair_counts(ps_sim *s, uint32_t **core_cnt, uint32_t **acc_in, uint32_t **acc_un)
{
    const uint64_t P = s->prm.max_distances;
    ps_population *core = s->core, *acc = s->acc;
    PSCHK(use_device(core));
    if (s->h_cnt_cap < P) {
        if (s->h_cnt) HIPCHK(hipHostFree(s->h_cnt));
        s->h_cnt = nullptr;
        s->h_cnt_cap = 0;
        HIPCHK(hipHostMalloc(&s->h_cnt, std::max<uint64_t>(3 * P, 1) * sizeof(uint32_t)));
        s->h_cnt_cap = P;
    }
    for (auto &e : s->ev_dist)
        if (!e) HIPCHK(hipEventCreate(&e));
    // the pair list names individuals by their output row (main.rs:413-427 draws it once, over the labels 0 .. N - 1)
    const uint32_t *slot = nullptr;
    PSCHK(rows_current(core, &slot));
    uint32_t *c_r1 = nullptr, *c_r2 = nullptr, *a_r1 = nullptr, *a_r2 = nullptr;
    PSCHK(pairs_for_launch(core, P, s->r1.data(), s->r2.data(), true, slot, &c_r1, &c_r2, core->stream));
    PSCHK(pairs_for_launch(acc, P, s->r1.data(), s->r2.data(), true, slot, &a_r1, &a_r2, acc->stream));
    uint32_t *c1 = (uint32_t *)core->d_pairs, *a1 = (uint32_t *)acc->d_pairs;
    // everything the generation loop queued on either stream precedes the distance kernels of both
    HIPCHK(hipStreamSynchronize(acc->stream));
    HIPCHK(hipStreamSynchronize(core->stream));
    HIPCHK(hipEventRecord(s->ev_dist[0], core->stream));
    PSCHK(pair_counts_device(core, P, c_r1, c_r2, c1 + 2 * P, c1 + 3 * P, c1 + 4 * P, core->stream));
    HIPCHK(hipEventRecord(s->ev_dist[1], core->stream));
    HIPCHK(hipMemcpyAsync(s->h_cnt, c1 + 3 * P, P * 4, hipMemcpyDeviceToHost, core->stream));
    HIPCHK(hipEventRecord(s->ev_dist[2], acc->stream));
    if (acc->cfg.ncols) PSCHK(pair_counts_device(acc, P, a_r1, a_r2, a1 + 2 * P, a1 + 3 * P, a1 + 4 * P, acc->stream));
    else HIPCHK(hipMemsetAsync(a1 + 3 * P, 0, 2 * P * 4, acc->stream));
    HIPCHK(hipEventRecord(s->ev_dist[3], acc->stream));
    HIPCHK(hipMemcpyAsync(s->h_cnt + P, a1 + 3 * P, 2 * P * 4, hipMemcpyDeviceToHost, acc->stream));
    HIPCHK(hipStreamSynchronize(acc->stream));
    HIPCHK(hipStreamSynchronize(core->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, s->ev_dist[0], s->ev_dist[1]));
    s->dist_core_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, s->ev_dist[2], s->ev_dist[3]));
    s->dist_acc_ms = ms;
    *core_cnt = s->h_cnt;
    *acc_in = s->h_cnt + P;
    *acc_un = s->h_cnt + 2 * P;
    return PS_OK;
}

static void acc_distances_host(const uint32_t *in, const uint32_t *un, uint64_t P, double cg, double *out)
{
    par_for(P < (1u << 20) ? 1 : P, [&](uint64_t a, uint64_t b) {
        if (P < (1u << 20)) { a = 0; b = P; }
        for (uint64_t k = a; k < b; k++) out[k] = 1.0 - (((double)in[k] + cg) / ((double)un[k] + cg));   // population.rs:828-830
    });
}

extern "C" int ps_sim_pairwise_distances(ps_sim *s, double *core_out, double *acc_out)
{
    if (!s || !core_out || !acc_out) return ps_fail(PS_ERR_INVALID, "null argument");
    const uint64_t P = s->prm.max_distances;
    uint32_t *cc = nullptr, *ai = nullptr, *au = nullptr;
    PSCHK(sim_pair_counts(s, &cc, &ai, &au));
    const double ncols = (double)s->core->cfg.ncols;
    par_for(P < (1u << 20) ? 1 : P, [&](uint64_t a, uint64_t b) {
        if (P < (1u << 20)) { a = 0; b = P; }
        for (uint64_t k = a; k < b; k++) core_out[k] = (double)(cc[k] / 2) / ncols;                      // population.rs:817-822
    });
    acc_distances_host(ai, au, P, (double)s->prm.core_genes, acc_out);
    return PS_OK;
}

// device time of the kernels of the last ps_sim_pairwise_distances call, per matrix (HIP events on their streams)
extern "C" int ps_sim_distance_timing(ps_sim *s, double *core_ms, double *acc_ms)
{
    if (!s) return ps_fail(PS_ERR_INVALID, "null handle");
    if (core_ms) *core_ms = s->dist_core_ms;
    if (acc_ms) *acc_ms = s->dist_acc_ms;
    return PS_OK;
}

extern "C" int ps_sim_host_timing(ps_sim *s, int reset, uint64_t *generations, double *wait_ms, double *weights_ms,
                                  double *draw_ms)
{
    if (!s) return ps_fail(PS_ERR_INVALID, "null handle");
    if (generations) *generations = s->host_calls;
    if (wait_ms) *wait_ms = s->host_wait_ms;
    if (weights_ms) *weights_ms = s->host_weights_ms;
    if (draw_ms) *draw_ms = s->host_draw_ms;
    if (reset) { s->host_calls = 0; s->host_wait_ms = s->host_weights_ms = s->host_draw_ms = 0.0; }
    return PS_OK;
}

extern "C" int ps_sim_enable_timing(ps_sim *s, int on)
{
    if (!s) return ps_fail(PS_ERR_INVALID, "null handle");
    s->timing = on != 0;
    return PS_OK;
}

extern "C" int ps_sim_sweep_timing(ps_sim *s, int reset, uint64_t *launches, double *total_ms,
                                   double *bytes_per_launch)
{
    if (!s) return ps_fail(PS_ERR_INVALID, "null handle");
    PSCHK(use_device(s->core));
    HIPCHK(hipStreamSynchronize(s->core->stream));
    double tot = 0.0;
    for (auto &pr : s->tev) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, pr.first, pr.second));
        tot += ms;
    }
    if (launches) *launches = s->tev.size();
    if (total_ms) *total_ms = tot;
    if (bytes_per_launch) *bytes_per_launch = 2.0 * (double)s->prm.pop_size * (double)s->core->cfg.ncols;
    if (reset) {
        for (auto &pr : s->tev) { s->ev_pool.push_back(pr.first); s->ev_pool.push_back(pr.second); }
        s->tev.clear();
    }
    return PS_OK;
}

// ---------------------------------------------------------------------------
// Several site shards in one process (main.rs:429-553 is one process): ps_multi
// ---------------------------------------------------------------------------
// One ps_sim per shard of the core sites, each on its own device (ordinals may repeat: two
// shards on one GPU), each driven by its own host thread for the duration of a call.  The shards never
// talk to each other during a generation (the accessory matrix is replicated and every shard draws the
// same parents); the one exchange step is the sum of the P Hamming numerators of the distance phase.
struct ps_multi;
struct multi_ctx { ps_multi *m; size_t k; };
struct ps_multi {
    ps_sim_params prm{};
    std::vector<ps_sim *> shard;
    std::vector<uint32_t *> d_cnt;      // per shard, on its device: P numerators
    uint32_t *d_tmp = nullptr;          // on shard 0's device: landing buffers of the peer copies (K - 1 x P)
    uint64_t cnt_cap = 0;
    // rendezvous of the shard threads inside one call, and the exchange of the donor-sharded HGT deltas between the
    // shards' devices (ps_set_donor_shard): every shard ORs its peers' delta buffers into its own, reading them where
    // they are (peer access over xGMI between GPUs, plain loads inside one)
    std::mutex mu;
    std::condition_variable cv;
    size_t bar_count = 0;
    uint64_t bar_gen = 0;
    bool failed = false;
    bool in_call = false;               // a ps_multi_* call is driving every shard (multi_for_each): the barriers are complete
    bool donor_sharded = false;
    bool peers_ok = true;               // every shard's device can read every other's memory
    std::vector<uint64_t *> delta;
    std::vector<hipEvent_t> ev_ready, ev_read;
    // sliced form of the exchange (round 5; PANSIM_MULTI_EXCHANGE=or keeps the reading kernels): per shard a copy stream,
    // a landing buffer for the K - 1 slices it merges, and events "my slices have arrived" / "my slice is merged" / "my
    // gather copies are done"
    bool sliced = true;
    std::vector<hipStream_t> copy_stream;
    std::vector<uint64_t *> landing;
    std::vector<uint64_t> landing_cap;
    std::vector<hipEvent_t> ev_a2a, ev_merged, ev_done;
    std::vector<multi_ctx> ctx;
    // the host half of a generation (three softmaxes over N individuals) is computed by shard 0 and shared: the shards'
    // replicas of the accessory matrix are bit-identical, and K shards x up to 16 threads each oversubscribe the host
    std::vector<double> shared_w;
    int shared_rc = PS_OK;
    std::string shared_err;
};

// barrier of the shard threads; fails (instead of hanging) once any shard has failed
static int multi_barrier(ps_multi *m)
{
    std::unique_lock<std::mutex> lk(m->mu);
    // a shard handle borrowed with ps_multi_shard and driven on its own (ps_sim_run, ps_recombine) would wait here for
    // peers that never come: fail instead
    if (!m->in_call)
        return ps_fail(PS_ERR_STATE, "this shard belongs to a ps_multi: its generations and its HGT take part in exchanges between "
                                     "all shards and can only be driven through ps_multi_run");
    if (m->failed) return ps_fail(PS_ERR_STATE, "another shard failed");
    const uint64_t gen = m->bar_gen;
    if (++m->bar_count == m->shard.size()) {
        m->bar_count = 0;
        m->bar_gen++;
        m->cv.notify_all();
        return PS_OK;
    }
    m->cv.wait(lk, [&] { return m->bar_gen != gen || m->failed; });
    if (m->bar_gen == gen) return ps_fail(PS_ERR_STATE, "another shard failed");
    return PS_OK;
}

// dst[w] |= OR over j of land[j * part + w]   (the K - 1 slices that arrived for this shard, into its own slice in place)
__global__ void __launch_bounds__(256) multi_or_slices_kernel(uint64_t *dst, const uint64_t *land, uint64_t len, uint64_t part, uint32_t n_land)
{
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= len) return;
    uint64_t v = dst[w];
    for (uint32_t j = 0; j < n_land; j++) v |= land[(uint64_t)j * part + w];
    dst[w] = v;
}

// ps_exchange_fn of the shards of one process, SLICED form (round 5): the same three steps as the RCCL provider, with the
// copy engines doing the moving -- hipMemcpyPeerAsync on a copy stream of its own needs no wave slot, so the exchange
// runs whatever the CUs are busy with, and 2 (K - 1) / K x the buffer crosses the links per shard instead of the K - 1
// whole buffers the reading kernels pulled:
//   1. slice k of every peer's buffer is copied into shard k's landing buffer (K - 1 peer copies, one per link),
//   2. one small kernel ORs them into slice k of shard k's own buffer, in place (peers only ever read the OTHER slices),
//   3. the merged slice j of every peer j is copied over slice j of shard k's buffer (K - 1 peer copies).
static int multi_exchange_sliced(multi_ctx *c, void *d_words, uint64_t n_words, hipStream_t st)
{
    ps_multi *m = c->m;
    const size_t k = c->k, K = m->shard.size();
    const uint64_t part = (n_words + K - 1) / K;
    auto lo = [&](size_t j) { return std::min<uint64_t>(n_words, (uint64_t)j * part); };
    auto len = [&](size_t j) { return std::min<uint64_t>(n_words, (uint64_t)(j + 1) * part) - lo(j); };
    const int dev_k = m->shard[k]->core->device;
    hipStream_t cs = m->copy_stream[k];
    if (m->landing_cap[k] < (K - 1) * part) {
        HIPCHK(hipStreamSynchronize(cs));
        if (m->landing[k]) HIPCHK(hipFree(m->landing[k]));
        m->landing[k] = nullptr;
        m->landing_cap[k] = 0;
        HIPCHK(hipMalloc(&m->landing[k], (K - 1) * part * 8));
        m->landing_cap[k] = (K - 1) * part;
    }
    m->delta[k] = (uint64_t *)d_words;
    HIPCHK(hipEventRecord(m->ev_ready[k], st));
    PSCHK(multi_barrier(m));                     // every shard's buffer is published and its "complete" event recorded
    uint32_t slot = 0;
    for (size_t j = 0; j < K; j++) {
        if (j == k) continue;
        HIPCHK(hipStreamWaitEvent(cs, m->ev_ready[j], 0));
        if (len(k))
            HIPCHK(hipMemcpyPeerAsync(m->landing[k] + (uint64_t)slot * part, dev_k, m->delta[j] + lo(k), m->shard[j]->core->device, len(k) * 8, cs));
        slot++;
    }
    HIPCHK(hipEventRecord(m->ev_a2a[k], cs));
    HIPCHK(hipStreamWaitEvent(st, m->ev_a2a[k], 0));
    if (len(k)) {
        multi_or_slices_kernel<<<(uint32_t)((len(k) + 255) / 256), 256, 0, st>>>((uint64_t *)d_words + lo(k), m->landing[k], len(k), part, (uint32_t)(K - 1));
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(m->ev_merged[k], st));
    PSCHK(multi_barrier(m));                     // every shard has recorded "my slice is merged" (and, before it, "my slices have arrived")
    for (size_t j = 0; j < K; j++) {
        if (j == k) continue;
        // (ev_merged[j] also says that shard j has finished READING slice j of this buffer: its step 1 precedes its merge)
        HIPCHK(hipStreamWaitEvent(cs, m->ev_merged[j], 0));
        if (len(j))
            HIPCHK(hipMemcpyPeerAsync((uint64_t *)d_words + lo(j), dev_k, m->delta[j] + lo(j), m->shard[j]->core->device, len(j) * 8, cs));
    }
    HIPCHK(hipEventRecord(m->ev_done[k], cs));
    HIPCHK(hipStreamWaitEvent(st, m->ev_done[k], 0));
    PSCHK(multi_barrier(m));                     // every shard has queued its gather copies
    for (size_t j = 0; j < K; j++)               // nobody rewrites its buffer (next generation) before its peers have copied its slice
        if (j != k) HIPCHK(hipStreamWaitEvent(st, m->ev_done[j], 0));
    m->shard[k]->exchange_calls++;
    m->shard[k]->exchange_bytes += 2 * (uint64_t)(K - 1) * part * 8;
    return PS_OK;
}

// ps_exchange_fn of the shards of one process.  Reading-kernel form (PANSIM_MULTI_EXCHANGE=or): OR is idempotent and the
// buffers only ever gain bits that belong to the union, so a shard may OR into its buffer while its peers read it (aligned
// 8-byte accesses): in place, no staging -- but K - 1 whole buffers cross the links per shard, pulled by kernels on the CUs.
static int multi_exchange(void *vctx, void *d_words, uint64_t n_words, void *hip_stream)
{
    multi_ctx *c = (multi_ctx *)vctx;
    ps_multi *m = c->m;
    const size_t k = c->k, K = m->shard.size();
    hipStream_t st = (hipStream_t)hip_stream;
    if (m->sliced) return multi_exchange_sliced(c, d_words, n_words, st);
    m->delta[k] = (uint64_t *)d_words;
    HIPCHK(hipEventRecord(m->ev_ready[k], st));
    PSCHK(multi_barrier(m));                     // every shard's buffer is published and its "complete" event recorded
    for (size_t j = 0; j < K; j++) {
        if (j == k) continue;
        HIPCHK(hipStreamWaitEvent(st, m->ev_ready[j], 0));
        acc_or_kernel<<<(uint32_t)((n_words + 255) / 256), 256, 0, st>>>((uint64_t *)d_words, m->delta[j], n_words);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(m->ev_read[k], st));
    PSCHK(multi_barrier(m));                     // every shard has queued its reads
    for (size_t j = 0; j < K; j++)               // nobody rewrites its buffer (next generation) before its peers have read it
        if (j != k) HIPCHK(hipStreamWaitEvent(st, m->ev_read[j], 0));
    m->shard[k]->exchange_calls++;
    m->shard[k]->exchange_bytes += (uint64_t)(K - 1) * n_words * 8;
    return PS_OK;
}

static int multi_weights(void *vctx, ps_sim *s, uint32_t gen, double *w)
{
    multi_ctx *c = (multi_ctx *)vctx;
    ps_multi *m = c->m;
    const uint64_t N = m->prm.pop_size;
    // D-avg sharded by rows over the shards like the HGT donors: every shard computes its rows and takes part in the exchange
    bool avg_ready = false;
    if (m->prm.competition_strength > 0.0 && s->acc->exchange && s->acc->donor_cnt != 0 && s->acc->d.G > 0) {
        // (every shard takes the same branch: the flags are set in lockstep by sim_one_generation)
        if (!(s->avg_prefetched && s->avg_epoch == s->acc->edit_epoch)) PSCHK(sim_average_distance(s));
        avg_ready = true;
    }
    if (c->k == 0) {
        m->shared_rc = sim_host_weights(s, gen, m->shared_w.data(), avg_ready);
        if (m->shared_rc != PS_OK) m->shared_err = g_err;
    }
    // the prefetched vector is consumed by this generation on EVERY shard (only shard 0 goes through sim_host_weights, which
    // clears the flag too): a shard must never carry it into a later generation and skip a collective its peers enter
    s->avg_prefetched = false;
    PSCHK(multi_barrier(m));                     // shard 0 has published the weights
    const int rc = m->shared_rc;
    if (rc == PS_OK) memcpy(w, m->shared_w.data(), N * sizeof(double));
    else g_err = m->shared_err;
    PSCHK(multi_barrier(m));                     // every shard has its copy: shard 0 may compute the next generation's
    return rc;
}

// run fn(k) for every shard on its own host thread; the first failure (with its message) is returned
template <typename F>
static int multi_for_each(ps_multi *m, F fn)
{
    const size_t n = m->shard.size();
    std::vector<int> rc(n, PS_OK);
    std::vector<std::string> err(n);
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->failed = false;
        m->bar_count = 0;
        m->in_call = true;
    }
    struct leave { ps_multi *m; ~leave() { std::lock_guard<std::mutex> lk(m->mu); m->in_call = false; } } leave_guard{ m };
    auto body = [&](size_t k) {
        rc[k] = fn(k);
        if (rc[k] != PS_OK) {
            err[k] = g_err;      // g_err is thread local
            std::lock_guard<std::mutex> lk(m->mu);
            m->failed = true;    // peers waiting in multi_barrier give up
            m->cv.notify_all();
        }
    };
    if (n == 1) {
        body(0);
    } else {
        std::vector<std::thread> th;
        for (size_t k = 0; k < n; k++) th.emplace_back(body, k);
        for (auto &t : th) t.join();
    }
    // report the root cause, not a peer's "another shard failed"
    for (size_t k = 0; k < n; k++)
        if (rc[k] != PS_OK && err[k].find("another shard failed") == std::string::npos) { g_err = err[k]; return rc[k]; }
    for (size_t k = 0; k < n; k++)
        if (rc[k] != PS_OK) { g_err = err[k]; return rc[k]; }
    return PS_OK;
}

extern "C" void ps_multi_destroy(ps_multi *m)
{
    if (!m) return;
    for (size_t k = 0; k < m->shard.size(); k++) {
        if (m->shard[k] && m->shard[k]->core) (void)hipSetDevice(m->shard[k]->core->device);
        if (k < m->d_cnt.size() && m->d_cnt[k]) (void)hipFree(m->d_cnt[k]);
        if (k == 0 && m->d_tmp) (void)hipFree(m->d_tmp);
        if (k < m->ev_ready.size() && m->ev_ready[k]) (void)hipEventDestroy(m->ev_ready[k]);
        if (k < m->ev_read.size() && m->ev_read[k]) (void)hipEventDestroy(m->ev_read[k]);
        if (k < m->ev_a2a.size() && m->ev_a2a[k]) (void)hipEventDestroy(m->ev_a2a[k]);
        if (k < m->ev_merged.size() && m->ev_merged[k]) (void)hipEventDestroy(m->ev_merged[k]);
        if (k < m->ev_done.size() && m->ev_done[k]) (void)hipEventDestroy(m->ev_done[k]);
        if (k < m->copy_stream.size() && m->copy_stream[k]) { (void)hipStreamSynchronize(m->copy_stream[k]); (void)hipStreamDestroy(m->copy_stream[k]); }
        if (k < m->landing.size() && m->landing[k]) (void)hipFree(m->landing[k]);
        ps_sim_destroy(m->shard[k]);
    }
    delete m;
}

extern "C" int ps_multi_create(const ps_sim_params *p, int n_shards, const int *devices, ps_multi **out)
{
    if (!p || !out) return ps_fail(PS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (n_shards < 1 || n_shards > 1024) return ps_fail(PS_ERR_INVALID, "n_shards must be 1..1024");
    if ((uint64_t)n_shards > p->core_size) return ps_fail(PS_ERR_INVALID, "more shards than core sites");
    if (p->shard_count != 1 || p->shard_rank != 0)
        return ps_fail(PS_ERR_INVALID, "ps_multi_create shards the run itself: pass shard_rank 0, shard_count 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return ps_fail(PS_ERR_NO_DEVICE, "no HIP device is visible: libpansim_hip has no CPU path");
    ps_multi *m = new ps_multi();
    m->prm = *p;
    m->shard.assign((size_t)n_shards, nullptr);
    m->d_cnt.assign((size_t)n_shards, nullptr);
    // (created one after the other: each creation synchronises its device anyway)
    for (int k = 0; k < n_shards; k++) {
        ps_sim_params q = *p;
        q.shard_rank = k;
        q.shard_count = n_shards;
        q.device = devices ? devices[k] : (p->device >= 0 && n_shards == 1 ? p->device : k % ndev);
        const int rc = ps_sim_create(&q, &m->shard[(size_t)k]);
        if (rc != PS_OK) {
            const std::string keep = g_err;
            ps_multi_destroy(m);
            g_err = keep;
            return rc;
        }
    }
    // direct xGMI access between the shards' devices (the sum of the distance numerators, the HGT delta exchange);
    // without peer access the copies are staged by the runtime, and the delta exchange is not switched on
    bool &peers_ok = m->peers_ok;
    for (int a = 0; a < n_shards; a++)
        for (int b = 0; b < n_shards; b++) {
            const int da = m->shard[(size_t)a]->core->device, db = m->shard[(size_t)b]->core->device;
            if (da == db) continue;
            int can = 0;
            if (hipSetDevice(da) == hipSuccess && hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); peers_ok = false; }
            } else {
                peers_ok = false;
            }
        }
    (void)hipGetLastError();
    // HGT donors sharded over the shards (each generates 1/K of the events, the deltas are ORed across the devices);
    // PANSIM_MULTI_REPLICATED_HGT=1 keeps round 2's form (every shard generates every event, no exchange)
    if (n_shards > 1 && peers_ok && !getenv("PANSIM_MULTI_REPLICATED_HGT") && (uint64_t)n_shards <= p->pop_size) {
        m->delta.assign((size_t)n_shards, nullptr);
        m->ev_ready.assign((size_t)n_shards, nullptr);
        m->ev_read.assign((size_t)n_shards, nullptr);
        m->ev_a2a.assign((size_t)n_shards, nullptr);
        m->ev_merged.assign((size_t)n_shards, nullptr);
        m->ev_done.assign((size_t)n_shards, nullptr);
        m->copy_stream.assign((size_t)n_shards, nullptr);
        m->landing.assign((size_t)n_shards, nullptr);
        m->landing_cap.assign((size_t)n_shards, 0);
        if (const char *e = getenv("PANSIM_MULTI_EXCHANGE")) m->sliced = strcmp(e, "or") != 0;
        m->ctx.resize((size_t)n_shards);
        int rc = PS_OK;
        for (int k = 0; k < n_shards && rc == PS_OK; k++) {
            m->ctx[(size_t)k] = multi_ctx{ m, (size_t)k };
            if (hipSetDevice(m->shard[(size_t)k]->core->device) != hipSuccess
                || hipEventCreateWithFlags(&m->ev_ready[(size_t)k], hipEventDisableTiming) != hipSuccess
                || hipEventCreateWithFlags(&m->ev_a2a[(size_t)k], hipEventDisableTiming) != hipSuccess
                || hipEventCreateWithFlags(&m->ev_merged[(size_t)k], hipEventDisableTiming) != hipSuccess
                || hipEventCreateWithFlags(&m->ev_done[(size_t)k], hipEventDisableTiming) != hipSuccess
                || hipStreamCreateWithFlags(&m->copy_stream[(size_t)k], hipStreamNonBlocking) != hipSuccess
                || hipEventCreateWithFlags(&m->ev_read[(size_t)k], hipEventDisableTiming) != hipSuccess)
                rc = ps_fail(PS_ERR_NO_DEVICE, "cannot create the exchange events of shard %d", k);
            else
                rc = ps_sim_set_exchange(m->shard[(size_t)k], multi_exchange, &m->ctx[(size_t)k]);
        }
        if (rc != PS_OK) {
            const std::string keep = g_err;
            ps_multi_destroy(m);
            g_err = keep;
            return rc;
        }
        m->donor_sharded = true;
    }
    if (n_shards > 1 && !getenv("PANSIM_MULTI_OWN_WEIGHTS")) {
        if (m->ctx.empty()) {
            m->ctx.resize((size_t)n_shards);
            for (int k = 0; k < n_shards; k++) m->ctx[(size_t)k] = multi_ctx{ m, (size_t)k };
        }
        m->shared_w.assign(p->pop_size, 0.0);
        for (int k = 0; k < n_shards; k++) {
            m->shard[(size_t)k]->weights_hook = multi_weights;
            m->shard[(size_t)k]->weights_ctx = &m->ctx[(size_t)k];
        }
    }
    (void)hipGetLastError();
    *out = m;
    return PS_OK;
}

extern "C" int ps_multi_shards(ps_multi *m) { return m ? (int)m->shard.size() : 0; }
extern "C" ps_sim *ps_multi_shard(ps_multi *m, int k)
{
    return (m && k >= 0 && (size_t)k < m->shard.size()) ? m->shard[(size_t)k] : nullptr;
}

extern "C" int ps_multi_run(ps_multi *m, uint32_t first_generation, uint32_t count)
{
    if (!m) return ps_fail(PS_ERR_INVALID, "null handle");
    return multi_for_each(m, [&](size_t k) { return ps_sim_run(m->shard[k], first_generation, count); });
}

extern "C" int ps_multi_sync(ps_multi *m)
{
    if (!m) return ps_fail(PS_ERR_INVALID, "null handle");
    return multi_for_each(m, [&](size_t k) { return ps_sim_sync(m->shard[k]); });
}

// Hamming numerators of the run's sampled pairs over ALL core sites: every shard counts its sites into a
// buffer on its own device; the partial counts are then copied device to device onto shard 0's GPU
// (hipMemcpyPeerAsync: xGMI between GPUs, a plain copy inside one) and added there.
static int multi_core_counts(ps_multi *m, uint32_t **d_total)
{
    const uint64_t P = m->prm.max_distances;
    ps_population *c0 = m->shard[0]->core;
    if (m->cnt_cap < P) {
        for (size_t k = 0; k < m->shard.size(); k++) {
            PSCHK(use_device(m->shard[k]->core));
            if (m->d_cnt[k]) HIPCHK(hipFree(m->d_cnt[k]));
            m->d_cnt[k] = nullptr;
            HIPCHK(hipMalloc(&m->d_cnt[k], std::max<uint64_t>(P, 1) * sizeof(uint32_t)));
        }
        PSCHK(use_device(c0));
        if (m->d_tmp) HIPCHK(hipFree(m->d_tmp));
        m->d_tmp = nullptr;
        HIPCHK(hipMalloc(&m->d_tmp, std::max<uint64_t>(P, 1) * sizeof(uint32_t)));
        m->cnt_cap = P;
    }
    PSCHK(multi_for_each(m, [&](size_t k) {
        ps_sim *s = m->shard[k];
        PSCHK(ps_sim_sync(s));
        return ps_pairwise_counts(s->core, P, s->r1.data(), s->r2.data(), m->d_cnt[k], nullptr, 1);
    }));
    PSCHK(use_device(c0));
    for (size_t k = 1; k < m->shard.size(); k++) {
        if (m->peers_ok) {
            // shard 0's device adds the peer's numerators where they are (xGMI reads; the shards' kernels have completed:
            // ps_pairwise_counts synchronises)
            u32_add_kernel<<<(uint32_t)((P + 255) / 256), 256, 0, c0->stream>>>(m->d_cnt[0], m->d_cnt[k], P);
        } else {
            HIPCHK(hipMemcpyPeerAsync(m->d_tmp, c0->device, m->d_cnt[k], m->shard[k]->core->device, P * sizeof(uint32_t), c0->stream));
            u32_add_kernel<<<(uint32_t)((P + 255) / 256), 256, 0, c0->stream>>>(m->d_cnt[0], m->d_tmp, P);
        }
        HIPCHK(hipGetLastError());
    }
    *d_total = m->d_cnt[0];
    return PS_OK;
}

extern "C" int ps_multi_pairwise_counts(ps_multi *m, uint32_t *out_core)
{
    if (!m || !out_core) return ps_fail(PS_ERR_INVALID, "null argument");
    const uint64_t P = m->prm.max_distances;
    uint32_t *d_total = nullptr;
    PSCHK(multi_core_counts(m, &d_total));
    ps_population *c0 = m->shard[0]->core;
    HIPCHK(hipMemcpyAsync(out_core, d_total, P * sizeof(uint32_t), hipMemcpyDeviceToHost, c0->stream));
    HIPCHK(hipStreamSynchronize(c0->stream));
    return PS_OK;
}

// Population::pairwise_distances of both matrices for the run's pair list (main.rs:467-470)
extern "C" int ps_multi_pairwise_distances(ps_multi *m, double *core_out, double *acc_out)
{
    if (!m || !core_out || !acc_out) return ps_fail(PS_ERR_INVALID, "null argument");
    const uint64_t P = m->prm.max_distances;
    if (m->shard.size() == 1) return ps_sim_pairwise_distances(m->shard[0], core_out, acc_out);
    std::vector<uint32_t> cnt(P);
    PSCHK(ps_multi_pairwise_counts(m, cnt.data()));
    const double ncols = (double)m->prm.core_size;
    for (uint64_t k = 0; k < P; k++) {
        const uint32_t distance = cnt[k] / 2;                       // population.rs:817
        core_out[k] = (double)distance / ncols;                     // :822
    }
    ps_sim *s0 = m->shard[0];
    return ps_pairwise_distances(s0->acc, P, s0->r1.data(), s0->r2.data(), acc_out);
}

// Population::write for both matrices (main.rs:550-553): every line of <outpref>_core_genome.csv is the
// concatenation of the shards' columns, expanded on each shard's device
extern "C" int ps_multi_write(ps_multi *m, const char *outpref)
{
    if (!m || !outpref) return ps_fail(PS_ERR_INVALID, "null argument");
    const size_t K = m->shard.size();
    if (K == 1) {
        PSCHK(ps_write(m->shard[0]->core, outpref));
        return ps_write(m->shard[0]->acc, outpref);
    }
    PSCHK(ps_multi_sync(m));
    const uint64_t N = m->prm.pop_size, L = m->prm.core_size;
    const std::string path = std::string(outpref) + "_core_genome.csv";
    FILE *f = fopen(path.c_str(), "w");
    if (!f) return ps_fail(PS_ERR_IO, "cannot create %s", path.c_str());
    const uint32_t chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(N, (256ull << 20) / (2 * L)));
    std::vector<uint8_t *> d_text(K, nullptr), h_text(K, nullptr);
    int rc = PS_OK;
    for (size_t k = 0; k < K && rc == PS_OK; k++) {
        ps_population *c = m->shard[k]->core;
        const uint64_t bytes = std::max<uint64_t>((uint64_t)chunk * 2 * c->cfg.ncols, 1);
        if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&d_text[k], bytes) != hipSuccess
            || hipHostMalloc(&h_text[k], bytes) != hipSuccess)
            rc = ps_fail(PS_ERR_OOM, "cannot allocate the text buffers of ps_multi_write");
    }
    for (uint64_t i0 = 0; rc == PS_OK && i0 < N; i0 += chunk) {
        const uint32_t ni = (uint32_t)std::min<uint64_t>(chunk, N - i0);
        rc = multi_for_each(m, [&](size_t k) {
            ps_population *c = m->shard[k]->core;
            const uint64_t C = c->cfg.ncols;
            PSCHK(use_device(c));
            dim3 grid((uint32_t)((C + 63) / 64), (ni + 63) / 64);
            const uint32_t *slot = nullptr;
            PSCHK(rows_current(c, &slot));
            core_csv_kernel<<<grid, 256, 0, c->stream>>>(c->state, d_text[k], c->pitch, C, (uint32_t)i0, ni,
                                                         (uint8_t)(k + 1 == K ? '\n' : ','), slot ? c->d_row_slot : nullptr);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(h_text[k], d_text[k], (uint64_t)ni * 2 * C, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            return (int)PS_OK;
        });
        for (uint32_t i = 0; rc == PS_OK && i < ni; i++)
            for (size_t k = 0; k < K; k++) {
                const uint64_t rb = 2 * m->shard[k]->core->cfg.ncols;
                if (fwrite(h_text[k] + (uint64_t)i * rb, 1, rb, f) != rb) {
                    rc = ps_fail(PS_ERR_IO, "short write to %s", path.c_str());
                    break;
                }
            }
    }
    for (size_t k = 0; k < K; k++) {
        (void)hipSetDevice(m->shard[k]->core->device);
        if (d_text[k]) (void)hipFree(d_text[k]);
        if (h_text[k]) (void)hipHostFree(h_text[k]);
    }
    fclose(f);
    PSCHK(rc);
    return ps_write(m->shard[0]->acc, outpref);
}

// the native RCCL provider of ps_exchange_fn (ps_rccl_*, ps_exchange_rccl)
#include "exchange_rccl.h"
