"""Online adaptive enrichment (reference python/dune/pylrbms/online_enrichment.py:9-93).

``doerfler_marking`` is restated in full (pure host logic).  ``AdaptiveEnrichment`` keeps the reference's constructor
and ``solve`` loop.  The enrichment step differs in how it is executed, not in what it computes: the reference calls
``reductor.enrich_local`` for one marked subdomain after the other (:49-50); here all corrector problems of a round are
solved by ONE kernel launch (one workgroup per marked neighbourhood, ``lrbms_local_correction_solve``) and the bases
are extended by one masked Gram-Schmidt step, then ``reductor.reduce(touched=marked)`` re-runs the fused project+estimate pass
over the marked subdomains and their neighbours only (pylrbms_amd/reductor.py: incremental re-projection)."""
import numpy as np


def doerfler_marking(indicators, theta):
    """online_enrichment.py:9-22: smallest set of subdomains whose squared indicators sum to > theta * total."""
    assert 0.0 < theta <= 1.0
    indices = list(range(len(indicators)))
    indicators = [ii ** 2 for ii in indicators]
    indicators, indices = [list(x) for x in zip(*sorted(zip(indicators, indices), key=lambda pair: pair[0],
                                                        reverse=True))]
    total = np.sum(indicators)
    sums = np.array([np.sum(indicators[:ii + 1]) for ii in np.arange(len(indicators))])
    where = sums > theta * total
    if np.any(where):
        return indices[:np.argmax(where) + 1]
    return indices


class AdaptiveEnrichment:

    def __init__(self, grid_and_problem_data, discretization, block_space, reductor, rd, target_error,
                 marking_doerfler_theta, marking_max_age):
        self.grid_and_problem_data = grid_and_problem_data
        self.discretization = discretization
        self.block_space = block_space
        self.reductor = reductor
        self.rd = rd
        self.target_error = target_error
        self.marking_doerfler_theta = marking_doerfler_theta
        self.marking_max_age = marking_max_age

    def _global_indicators(self, indicators):
        """Sharded discretization: the marking is global, so every rank needs the indicators of all subdomains (one
        all-gather of a double per subdomain); unsharded: the array as it is."""
        eng = self.discretization.engine
        if eng.S_ext == eng.S:
            return np.asarray(indicators)
        from pylrbms_amd.grid import DDSubdomainsGrid
        from pylrbms_amd.parallel import gather_subdomain_rows
        g = eng.grid
        owned = [list(DDSubdomainsGrid(g.lower_left, g.upper_right, g.K, g.P, rank=r, world_size=g.world_size).subdomains_on_rank)
                 for r in range(g.world_size)]
        loc = eng.ctx.from_numpy(np.asarray(indicators, dtype=np.float64).reshape(-1, 1))
        glob = gather_subdomain_rows(loc, owned, g.num_subdomains, getattr(self.discretization.mpi_comm, 'group', None))
        return glob[:, 0].cpu().numpy()

    def _enrich_once(self, U, mu, indicators, age_count):
        indicators = self._global_indicators(indicators)
        marked_subdomains = set(doerfler_marking(indicators, self.marking_doerfler_theta))
        for ii in np.where(age_count > self.marking_max_age)[0]:
            marked_subdomains.add(ii)
        mine = sorted(ii for ii in marked_subdomains if ii in set(self.discretization.engine.local))   # this rank's share
        if getattr(self, '_reserve', 0) and hasattr(self.reductor, 'reserve'):
            self.reductor.reserve(self.reductor.basis_size() + self._reserve)      # (every rank: the widths agree in reduce())
            self._reserve = 0
        if hasattr(self.reductor, 'enrich_local_batch'):
            self.reductor.enrich_local_batch(mine, U, mu)
        else:
            for ii in mine:
                self.reductor.enrich_local(ii, U, mu)
        # re-projection (online_enrichment.py:52 calls reductor.reduce()): only the marked subdomains' bases changed, so only they
        # and their neighbours are projected again -- into the arrays of self.rd -- unless the basis slab had to grow
        try:
            self.rd = self.reductor.reduce(touched=sorted(marked_subdomains))
        except TypeError:                                     # a reductor without the incremental form
            self.rd = self.reductor.reduce()
        for ii in range(self.block_space.num_blocks):
            age_count[ii] = 1 if ii in marked_subdomains else age_count[ii] + 1
        return len(marked_subdomains)

    def estimate(self, U, mu, decompose=False):
        return self.rd.estimate(U, mu=mu, decompose=decompose)

    def solve(self, mu, enrichment_steps=np.inf, callback=None):
        mu = self.discretization.parse_parameter(mu)
        # room for the vectors this loop can add (one per subdomain and round), reserved in front of the FIRST enrichment: the slab
        # then keeps its width and every later round re-projects marked + neighbours only (zero columns change no result)
        self._reserve = int(min(enrichment_steps, 16)) if np.isfinite(enrichment_steps) else 8
        enrichment_step = 1
        age_count = np.ones(self.block_space.num_blocks)
        local_problem_solves = 0
        while True:
            U = self.rd.solve(mu)
            eta, _, indicators = self.estimate(U, mu=mu, decompose=True)
            if callback:
                callback(self.rd, U, mu, {'eta': eta, 'local_problem_solves': local_problem_solves,
                                          'global RB size': self.rd.solution_space.dim,
                                          'local RB sizes': self.reductor.local_sizes()})
            if eta <= self.target_error:
                return U, self.rd, self.reductor
            if enrichment_step > enrichment_steps:
                return U, self.rd, self.reductor
            enrichment_step += 1
            local_problem_solves = self._enrich_once(U, mu, indicators[:, 0] if np.ndim(indicators) > 1 else indicators,
                                                     age_count)
