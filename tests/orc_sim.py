"""Oracle-side generation loop (main.rs:429-464) used by the parity tests.

At its boundary this double speaks the reference's language: `core` / `acc` hold row k = the child of draw k of the last
generation and `last_idx[k]` names that child's parent by the parent's row of the generation before (population.rs:443,
main.rs:445-447).  INSIDE it keeps its rows the way the library's engine does -- children in ascending parent order, a stable
sort of the draws (the gather of wide populations then works from a window of the parent row) -- because the
keyed randomness of a cell and the order of the weights the parents are drawn from follow the internal row."""
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
        self._core = np.ascontiguousarray(np.tile(cv, (N, 1)))       # internal rows (ascending parent order)
        av = o.init_acc_vec(seed, G, self.d.avg_gene_freq_adj)
        self._acc = np.ascontiguousarray(np.tile(av, (N, 1)))
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
        self.sigma = np.arange(N)            # output row k -> internal row, from the last generation's draws
        self._over = {"core": False, "acc": False}     # a matrix set from outside sits in the order it was given
        self.internal_idx = None             # the last draws as the engine sees them: ascending, parents by internal row

    # the matrices as the reference would hold them: row k = the child of draw k
    @property
    def core(self):
        return self._core if self._over["core"] else np.ascontiguousarray(self._core[self.sigma])

    @core.setter
    def core(self, m):                       # (ps_load_matrix on the simulation's handle: rows as given until the next generation)
        self._core, self._over["core"] = np.ascontiguousarray(m), True

    @property
    def acc(self):
        return self._acc if self._over["acc"] else np.ascontiguousarray(self._acc[self.sigma])

    @acc.setter
    def acc(self, m):
        self._acc, self._over["acc"] = np.ascontiguousarray(m), True

    def generation(self, gen):
        avg = np.ones(self.N)
        if self.competition > 0.0:
            avg = o.average_distance(self._acc, False, self.p.core_genes)
        # Population::sample_indices (orc.sample_indices) returns the draws in draw order, as the reference does; the
        # weights are those of the internal rows, so a draw names its parent by the parent's internal row
        rc, draw = o.sample_indices(self._acc, self.seed, gen, self.d.avg_gene_num, avg, self.sel,
                                    self.no_control, self.penalty, self.competition)
        assert rc == 0
        draw = draw.astype(np.uint32)
        row_of_slot = np.empty(self.N, np.int64)         # internal row -> output row, of the generation that is the parents'
        row_of_slot[self.sigma] = np.arange(self.N)
        self.last_idx = row_of_slot[draw].astype(np.uint32)
        # the engine's own order: a stable sort of the draws by parent
        order = np.argsort(draw, kind="stable")           # order[j] = the draw whose child sits in internal row j
        idx = draw[order]
        self.internal_idx = idx
        self.sigma = np.empty(self.N, np.int64)
        self.sigma[order] = np.arange(self.N)
        self._over = {"core": False, "acc": False}
        self._core = o.next_generation(self._core, idx)
        self._acc = o.next_generation(self._acc, idx)
        o.mutate_core(self._core, self.sb, self.seed, gen, self.plan)
        o.mutate_acc(self._acc, self.seed, gen, self.cb, self.ce, self.lm)
        if self.p.HR_rate > 0.0:
            o.recombine_core(self._core, self.sb, self.seed, gen, self.plan)
        if self.p.HGT_rate > 0.0:
            o.recombine_acc(self._acc, self.seed, gen, self.cb, self.ce, self.lr)
