// pansim_cli.cpp -- the `pansim` executable: the reference's command line
// (pansim/src/main.rs:17-152), validation (:195-247), generation loop (:429-528) and
// output files (:321-331, :467-499, :531-553) driving libpansim_hip.so through its C ABI.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pansim_hip.h"

struct Flag { const char *name; const char *help; const char *def; bool takes_value; };

// main.rs:21-151, in declaration order
static const Flag FLAGS[] = {
    { "pop_size", "Number of individuals in population.", "1000", true },
    { "core_size", "Number of nucleotides in core genome.", "1200000", true },
    { "pan_genes", "Total number of genes in pangenome (core + accessory).", "6000", true },
    { "core_genes", "Number of core genes in pangenome.", "2000", true },
    { "avg_gene_freq", "Average proportion of genes in pangenome present in an individual. Includes core and accessory genes.", "0.5", true },
    { "n_gen", "Number of generations to simulate.", "100", true },
    { "max_distances", "Maximum number of pairwise distances to calculate.", "100000", true },
    { "core_mu", "Average core SNP mutation rate (per site per genome per generation in core genome). Must be > 0.0.", "0.05", true },
    { "HR_rate", "Homologous recombination rate, as number of core sites transferred per core genome mutation.", "0.05", true },
    { "HGT_rate", "HGT rate, as number of accessory sites transferred per core genome mutation.", "0.05", true },
    { "rate_genes1", "Average number of accessory genes that are gained/lost per site per genome per generation in gene compartment 1. Must be >= 0.0.", "1.0", true },
    { "rate_genes2", "Average number of accessory genes that are gained/lost per site per genome per generation in gene compartment 2. Must be >= 0.0.", "1000.0", true },
    { "prop_genes2", "Proportion of pangenome made up of compartment 2 genes. Must be 0.0 <= X <= 1.0.", "0.1", true },
    { "prop_positive", "Proportion of pangenome made up of positively selected genes. Must be 0.0 <= X <= 1.0. If negative, neutral selection is simulated.", "-0.1", true },
    { "pos_lambda", "Lambda value for exponential distribution of positively selected genes. Must be > 0.0.", "10.0", true },
    { "neg_lambda", "Lambda value for exponential distribution of negatively selected genes. Must be > 0.0.", "10.0", true },
    { "seed", "Seed for random number generation.", "0", true },
    { "outpref", "Output prefix path.", "distances", true },
    { "print_dist", "Print per-generation average pairwise distances.", nullptr, false },
    { "print_matrices", "Prints core and accessory matrices.", nullptr, false },
    { "print_selection", "Prints selection coefficients.", nullptr, false },
    { "threads", "Number of threads.", "1", true },
    { "verbose", "Prints per generation information.", nullptr, false },
    { "no_control_genome_size", "Removes penalisation of genome sizes deviating from average.", nullptr, false },
    { "genome_size_penalty", "Multiplier for each gene difference between avg_gene_freq and observed value.", "0.99", true },
    { "competition_strength", "Strength of competition felt by strain to all others. 0.0 = no competition", "0.0", true },
};

// not a flag of the reference: number of core-site shards of this process, shard k on device k modulo the
// visible GPUs (include/pansim_hip.h, ps_multi); results do not depend on it
static const Flag EXT_FLAGS[] = {
    { "gpus", "Number of core-site shards, one per GPU (more shards than GPUs share them). Results do not depend on it.", "1", true },
    { "reference_seed_stream", "Draw the selection coefficients from the reference's own seeded stream (ChaCha12 StdRng, restated from the published algorithms of the rand / statrs crates, ziggurat tables recomputed: UNVERIFIED against a Pansim binary) instead of the build's Philox stream.", nullptr, false },
};

// clap 3's layout (the reference's own `pansim --help`, /root/reference/README.md:40-138): the options sorted by clap's
// key in byte order (uppercase first; -h / -V by their letters), the help text behind a 12-space indent,
// "[default: X]" appended as ordinary words, the whole filled greedily to 100 columns.
static void print_wrapped(const std::string &text, size_t indent, size_t width)
{
    std::string line;
    size_t i = 0;
    while (i < text.size()) {
        size_t j = text.find(' ', i);
        if (j == std::string::npos) j = text.size();
        const std::string word = text.substr(i, j - i);
        i = j + 1;
        if (!line.empty() && indent + line.size() + 1 + word.size() > width) {
            printf("%*s%s\n", (int)indent, "", line.c_str());
            line.clear();
        }
        if (!line.empty()) line += ' ';
        line += word;
    }
    if (!line.empty()) printf("%*s%s\n", (int)indent, "", line.c_str());
}

static void print_flag(const Flag &f, const char *short_name)
{
    if (short_name) printf("    -%s, --%s", short_name, f.name);
    else printf("        --%s", f.name);
    if (f.takes_value) printf(" <%s>", f.name);
    printf("\n");
    std::string text = f.help;
    if (f.takes_value) text += std::string(" [default: ") + f.def + "]";
    print_wrapped(text, 12, 100);
}

static void print_help(bool extensions)
{
    printf("pansim 0.1.0\nSamuel Horsfield shorsfield@ebi.ac.uk\n");
    print_wrapped("Runs Wright-Fisher simulation, simulating neutral core genome evolution and two-speed accessory genome evolution.", 0, 100);
    printf("\nUSAGE:\n    pansim [OPTIONS]\n\nOPTIONS:\n");
    static const Flag HELP = { "help", "Print help information", nullptr, false }, VERSION = { "version", "Print version information", nullptr, false };
    std::vector<std::pair<const Flag *, const char *>> all;
    for (const Flag &f : FLAGS) all.push_back({ &f, nullptr });
    all.push_back({ &HELP, "h" });
    all.push_back({ &VERSION, "V" });
    // (clap's sort key: the long name, or for an option with a short one that letter in lower case followed by '0' if it
    // is a lower-case letter and '1' if not: "h0" lands behind genome_size_penalty, "v1" between threads and verbose)
    auto key = [](const std::pair<const Flag *, const char *> &x) {
        if (!x.second) return std::string(x.first->name);
        const char c = x.second[0];
        return std::string(1, (char)tolower(c)) + (islower(c) ? '0' : '1');
    };
    std::sort(all.begin(), all.end(), [&](const std::pair<const Flag *, const char *> &x, const std::pair<const Flag *, const char *> &y) { return key(x) < key(y); });
    for (size_t k = 0; k < all.size(); k++) {
        if (k) printf("\n");
        print_flag(*all[k].first, all[k].second);
    }
    if (extensions) {
        // (not part of the reference's --help: shown by --help-extensions only, so that --help stays the reference's text)
        printf("\nMI355X OPTIONS (not in the reference):\n");
        for (const Flag &f : EXT_FLAGS) {
            print_flag(f, nullptr);
            printf("\n");
        }
        printf("        --help-extensions\n            Print this help with the options above.\n");
    }
}

[[noreturn]] static void die(int code, const std::string &msg)
{
    fprintf(stderr, "%s\n", msg.c_str());
    exit(code);
}

// value_of_t::<f64>(..).unwrap() (main.rs:155-186): a value that does not parse panics (exit 101)
static double as_f64(const std::map<std::string, std::string> &v, const char *name)
{
    const std::string &s = v.at(name);
    char *end = nullptr;
    const double x = strtod(s.c_str(), &end);
    if (s.empty() || *end != 0)
        die(101, "error: Invalid value \"" + s + "\" for '--" + name + "': invalid float literal");
    return x;
}
static uint64_t as_u64(const std::map<std::string, std::string> &v, const char *name)
{
    const std::string &s = v.at(name);
    char *end = nullptr;
    if (s.empty() || s[0] == '-') die(101, "error: Invalid value \"" + s + "\" for '--" + name + "': invalid digit found in string");
    const unsigned long long x = strtoull(s.c_str(), &end, 10);
    if (*end != 0) die(101, "error: Invalid value \"" + s + "\" for '--" + name + "': invalid digit found in string");
    return x;
}
// `raw_f64.round() as usize` (main.rs:155-162): saturating cast
static uint64_t round_usize(double x)
{
    const double r = std::round(x);
    if (!(r > 0.0)) return 0;
    if (r >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)r;
}

static std::string fmt(double v)
{
    char buf[512];
    ps_fmt_f64(v, buf, sizeof buf);
    return buf;
}

// "<core>\t<acc>\n" per pair (main.rs:471-482).  The shortest-round-trip formatting of 2 P doubles
// is the whole cost of the file at cfg5 (33.5 M lines), so chunks of pairs are formatted by the
// host's threads into buffers that are then written in order.
static void write_pairs_tsv(FILE *f, const std::vector<double> &cd, const std::vector<double> &ad)
{
    const uint64_t P = cd.size();
    const uint64_t chunk = 1u << 16;
    const uint64_t nchunks = (P + chunk - 1) / chunk;
    const uint64_t hw = std::thread::hardware_concurrency();
    const unsigned nthreads = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(hw, nchunks), 32));
    for (uint64_t c0 = 0; c0 < nchunks; c0 += nthreads) {
        const unsigned n = (unsigned)std::min<uint64_t>(nthreads, nchunks - c0);
        std::vector<std::string> out(n);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n; t++)
            pool.emplace_back([&, t] {
                const uint64_t a = (c0 + t) * chunk, b = std::min(P, a + chunk);
                std::string &o = out[t];
                o.reserve((size_t)(b - a) * 40);
                char buf[512];
                for (uint64_t k = a; k < b; k++) {
                    o.append(buf, (size_t)ps_fmt_f64(cd[k], buf, sizeof buf));
                    o.push_back('\t');
                    o.append(buf, (size_t)ps_fmt_f64(ad[k], buf, sizeof buf));
                    o.push_back('\n');
                }
            });
        for (auto &th : pool) th.join();
        for (unsigned t = 0; t < n; t++) fwrite(out[t].data(), 1, out[t].size(), f);
    }
}

#define CK(call)                                                        \
    do {                                                                \
        if ((call) != PS_OK) die(101, std::string("pansim: ") + ps_last_error()); \
    } while (0)

int main(int argc, char **argv)
{
    std::map<std::string, std::string> val;
    std::map<std::string, bool> present;
    for (const Flag &f : FLAGS) {
        if (f.takes_value) val[f.name] = f.def;
        else present[f.name] = false;
    }
    for (const Flag &f : EXT_FLAGS) {
        if (f.takes_value) val[f.name] = f.def;
        else present[f.name] = false;
    }
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") { print_help(false); return 0; }
        if (a == "--help-extensions") { print_help(true); return 0; }
        if (a == "-V" || a == "--version") { printf("pansim 0.1.0\n"); return 0; }
        if (a.rfind("--", 0) != 0)
            die(2, "error: Found argument '" + a + "' which wasn't expected, or isn't valid in this context\n\nUSAGE:\n    pansim [OPTIONS]\n\nFor more information try --help");
        std::string name = a.substr(2), value;
        bool has_eq = false;
        const size_t eq = name.find('=');
        if (eq != std::string::npos) { value = name.substr(eq + 1); name = name.substr(0, eq); has_eq = true; }
        const Flag *fl = nullptr;
        for (const Flag &f : FLAGS)
            if (name == f.name) fl = &f;
        for (const Flag &f : EXT_FLAGS)
            if (name == f.name) fl = &f;
        if (!fl)
            die(2, "error: Found argument '--" + name + "' which wasn't expected, or isn't valid in this context\n\nUSAGE:\n    pansim [OPTIONS]\n\nFor more information try --help");
        if (!fl->takes_value) {
            if (has_eq) die(2, "error: The argument '--" + name + "' takes no value");
            present[name] = true;
            continue;
        }
        if (!has_eq) {
            if (i + 1 >= argc)
                die(2, "error: The argument '--" + name + " <" + name + ">' requires a value but none was supplied");
            value = argv[++i];
            // only prop_positive allows a leading hyphen (main.rs:91)
            if (value.rfind("-", 0) == 0 && name != "prop_positive" && value.size() > 1
                && !(value[1] >= '0' && value[1] <= '9') && value[1] != '.')
                die(2, "error: The argument '--" + name + " <" + name + ">' requires a value but none was supplied");
        }
        val[name] = value;
    }

    ps_sim_params p;
    ps_sim_default_params(&p);
    p.pop_size = round_usize(as_f64(val, "pop_size"));                 // main.rs:155-156
    p.core_size = round_usize(as_f64(val, "core_size"));
    p.pan_genes = round_usize(as_f64(val, "pan_genes"));
    p.core_genes = round_usize(as_f64(val, "core_genes"));
    p.avg_gene_freq = as_f64(val, "avg_gene_freq");
    p.HR_rate = as_f64(val, "HR_rate");
    p.HGT_rate = as_f64(val, "HGT_rate");
    {
        const double r = std::round(as_f64(val, "n_gen"));               // `as i32` saturates
        p.n_gen = r >= 2147483647.0 ? 2147483647 : r <= -2147483648.0 ? INT32_MIN : (int32_t)r;
    }
    const std::string outpref = val["outpref"];
    p.max_distances = as_u64(val, "max_distances");                    // main.rs:169 (usize)
    p.core_mu = as_f64(val, "core_mu");
    p.rate_genes1 = as_f64(val, "rate_genes1");
    p.rate_genes2 = as_f64(val, "rate_genes2");
    p.prop_genes2 = as_f64(val, "prop_genes2");
    p.prop_positive = as_f64(val, "prop_positive");
    p.pos_lambda = as_f64(val, "pos_lambda");
    p.neg_lambda = as_f64(val, "neg_lambda");
    (void)as_u64(val, "threads");                                      // main.rs:177; the GPU grid replaces the rayon pool
    (void)as_f64(val, "seed");                                         // main.rs:179
    p.seed = as_u64(val, "seed");                                      // main.rs:180
    p.verbose = present["verbose"];
    p.print_dist = present["print_dist"];
    p.print_matrices = present["print_matrices"];
    p.print_selection = present["print_selection"];
    p.no_control_genome_size = present["no_control_genome_size"];
    p.genome_size_penalty = as_f64(val, "genome_size_penalty");
    p.competition_strength = as_f64(val, "competition_strength");
    p.reference_seed_stream = present["reference_seed_stream"];

    // main.rs:195-247: message on stdout, exit status 0, no files
    char msg[2048];
    if (ps_sim_validate(&p, msg, sizeof msg) != PS_OK) {
        fputs(msg, stdout);
        return 0;
    }
    ps_derived d;
    CK(ps_sim_derive(&p, &d));
    if (p.verbose) printf("avg_gene_freq adjusted to %s\n", fmt(d.avg_gene_freq_adj).c_str()); // main.rs:269-271

    // one ps_sim per core-site shard (one shard: the plain run); main.rs:372-427
    const uint64_t n_shards = as_u64(val, "gpus");
    if (n_shards < 1 || n_shards > 1024) die(101, "pansim: --gpus must be 1..1024");
    ps_multi *multi = nullptr;
    CK(ps_multi_create(&p, (int)n_shards, nullptr, &multi));
    ps_sim *sim = ps_multi_shard(multi, 0);
    const uint64_t G = d.pan_size, P = p.max_distances;

    if (p.print_selection) {                                           // main.rs:321-331
        FILE *f = fopen((outpref + "_selection.tsv").c_str(), "w");
        if (!f) die(1, "Error: cannot create " + outpref + "_selection.tsv");
        const double *sel = ps_sim_selection(sim);
        for (uint64_t g = 0; g < G; g++) fprintf(f, "%s%s", g ? "\n" : "", fmt(sel[g]).c_str());
        fputc('\n', f);
        fclose(f);
    }

    ps_population *acc = ps_sim_acc(sim);      // replicated on every shard
    std::vector<double> avg_core(p.n_gen), avg_acc(p.n_gen), std_core(p.n_gen), std_acc(p.n_gen);
    std::vector<double> cd(P), ad(P);
    const bool stepwise = p.print_dist || p.verbose;
    if (!stepwise) CK(ps_multi_run(multi, 0, (uint32_t)p.n_gen));      // main.rs:429-464
    for (int32_t j = 0; j < p.n_gen; j++) {
        if (stepwise) CK(ps_multi_run(multi, (uint32_t)j, 1));
        if (j == p.n_gen - 1) {                                        // main.rs:467-499
            CK(ps_multi_sync(multi));
            CK(ps_multi_pairwise_distances(multi, cd.data(), ad.data()));
            FILE *f = fopen((outpref + ".tsv").c_str(), "w");
            if (!f) die(1, "Error: cannot create " + outpref + ".tsv");
            write_pairs_tsv(f, cd, ad);
            fclose(f);
            std::vector<double> freqs(G + p.core_genes);
            CK(ps_gene_frequencies(acc, freqs.data()));
            f = fopen((outpref + "_freqs.txt").c_str(), "w");
            if (!f) die(1, "Error: cannot create " + outpref + "_freqs.txt");
            for (double x : freqs) fprintf(f, "%s\n", fmt(x).c_str());
            fclose(f);
        }
        if (p.print_dist) {                                            // main.rs:502-519
            CK(ps_multi_sync(multi));
            CK(ps_multi_pairwise_distances(multi, cd.data(), ad.data()));
            CK(ps_standard_deviation(cd.data(), P, &std_core[j], &avg_core[j]));
            CK(ps_standard_deviation(ad.data(), P, &std_acc[j], &avg_acc[j]));
        }
        if (p.verbose) {                                               // main.rs:522-526
            printf("Finished gen: %d\n", j + 1);
            double gf = 0.0;
            CK(ps_multi_sync(multi));
            CK(ps_calc_gene_freq(acc, &gf));
            printf("avg_gene_freq: %s\n", fmt(gf).c_str());
        }
    }
    if (p.print_dist) {                                                // main.rs:531-548
        FILE *f = fopen((outpref + "_per_gen.tsv").c_str(), "w");
        if (!f) die(1, "Error: cannot create " + outpref + "_per_gen.tsv");
        for (int32_t j = 0; j < p.n_gen; j++)
            fprintf(f, "%s\t%s\t%s\t%s\n", fmt(avg_core[j]).c_str(), fmt(std_core[j]).c_str(),
                    fmt(avg_acc[j]).c_str(), fmt(std_acc[j]).c_str());
        fclose(f);
    }
    if (p.print_matrices) {                                            // main.rs:550-553 (errors ignored)
        (void)ps_multi_write(multi, outpref.c_str());
    }
    ps_multi_destroy(multi);
    return 0;
}
