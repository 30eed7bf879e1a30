"""Oracle-side generation loop (main.rs:429-464) used by the parity tests."""
import numpy as np

from oracle import oracle as o


class OracleSim:
    def __init__(self, seed=0, site_begin=None, site_end=None, prop_positive=-0.1, pos_lambda=10.0,
                 neg_lambda=10.0, no_control_genome_size=False, genome_size_penalty=0.99,
                 competition_strength=0.0, **kw):
        self.p = o.make_params(**kw)
        self.d = o.derive(self.p)
        self.seed = seed
        N, L, G = self.p.pop_size, self.p.core_size, self.d.pan_size
        self.N, self.L, self.G = N, L, G
        self.sb = 0 if site_begin is None else site_begin
        self.se = L if site_end is None else site_end
        cv = o.init_core_vec(seed, L)[self.sb:self.se]
        self.core = np.ascontiguousarray(np.tile(cv, (N, 1)))
        av = o.init_acc_vec(seed, G, self.d.avg_gene_freq_adj)
        self.acc = np.ascontiguousarray(np.tile(av, (N, 1)))
        self.sel = o.selection_coefficients(seed, G, prop_positive, pos_lambda, neg_lambda)
        self.no_control = no_control_genome_size
        self.penalty = genome_size_penalty
        self.competition = competition_strength
        lam_hr = self.d.n_recombinations_core if self.p.HR_rate > 0.0 else 0.0
        self.plan = o.core_plan(self.d.n_core_mutations, lam_hr, L)
        self.cb = [self.d.comp_begin[c] for c in range(self.d.n_comp)]
        self.ce = [self.d.comp_end[c] for c in range(self.d.n_comp)]
        self.lm = [self.d.n_pan_mutations[c] for c in range(self.d.n_comp)]
        self.lr = [self.d.n_recombinations_pan[c] if self.p.HGT_rate > 0.0 else 0.0
                   for c in range(self.d.n_comp)]
        self.last_idx = None

    def generation(self, gen):
        avg = np.ones(self.N)
        if self.competition > 0.0:
            avg = o.average_distance(self.acc, False, self.p.core_genes)
        rc, idx = o.sample_indices(self.acc, self.seed, gen, self.d.avg_gene_num, avg, self.sel,
                                   self.no_control, self.penalty, self.competition)
        assert rc == 0
        # the build's generation loop stores the children in ascending parent order (a relabeling of
        # exchangeable individuals that lets the gather of wide populations work from a window of the parent row);
        # Population::sample_indices itself (orc.sample_indices) returns the draws in draw order, as the reference does
        idx = np.sort(idx)
        self.last_idx = idx
        self.core = o.next_generation(self.core, idx)
        self.acc = o.next_generation(self.acc, idx)
        o.mutate_core(self.core, self.sb, self.seed, gen, self.plan)
        o.mutate_acc(self.acc, self.seed, gen, self.cb, self.ce, self.lm)
        if self.p.HR_rate > 0.0:
            o.recombine_core(self.core, self.sb, self.seed, gen, self.plan)
        if self.p.HGT_rate > 0.0:
            o.recombine_acc(self.acc, self.seed, gen, self.cb, self.ce, self.lr)
