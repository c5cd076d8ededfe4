"""Online adaptive enrichment (reference python/dune/pylrbms/online_enrichment.py:9-93).

``doerfler_marking`` is restated in full (pure host logic).  ``AdaptiveEnrichment`` keeps the reference's constructor
and ``solve`` loop.  The enrichment step differs in how it is executed, not in what it computes: the reference calls
``reductor.enrich_local`` for one marked subdomain after the other (:49-50); here all corrector problems of a round are
solved by ONE kernel launch (one workgroup per marked neighbourhood, ``lrbms_local_correction_solve``) and the bases
are extended by one masked Gram-Schmidt step, then ``reductor.reduce()`` re-runs the fused project+estimate pass."""
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

    def _enrich_once(self, U, mu, indicators, age_count):
        marked_subdomains = set(doerfler_marking(indicators, self.marking_doerfler_theta))
        for ii in np.where(age_count > self.marking_max_age)[0]:
            marked_subdomains.add(ii)
        if hasattr(self.reductor, 'enrich_local_batch'):
            self.reductor.enrich_local_batch(sorted(marked_subdomains), U, mu)
        else:
            for ii in marked_subdomains:
                self.reductor.enrich_local(ii, U, mu)
        self.rd = self.reductor.reduce()
        for ii in range(self.block_space.num_blocks):
            age_count[ii] = 1 if ii in marked_subdomains else age_count[ii] + 1
        return len(marked_subdomains)

    def estimate(self, U, mu, decompose=False):
        return self.rd.estimate(U, mu=mu, decompose=decompose)

    def solve(self, mu, enrichment_steps=np.inf, callback=None):
        mu = self.discretization.parse_parameter(mu)
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
