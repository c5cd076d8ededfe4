import sys
import numpy as np
sys.path.insert(0, '.')
from pylrbms_amd import OS2015_academic_problem as prob
from pylrbms_amd.discretize_elliptic_block_swipdg import discretize
from pylrbms_amd.online_enrichment import AdaptiveEnrichment
from pylrbms_amd.reductor import LRBMSReductor
p = prob.init_grid_and_problem({'num_subdomains': [4, 4], 'half_num_fine_elements_per_subdomain_and_dim': 16})
d, data = discretize(p)
reductor = LRBMSReductor(d, order=0)
if len(sys.argv) > 1:
    reductor.extend_basis(d.solve(1.0))
rd = reductor.reduce()
mu = d.parse_parameter(0.1)
Ufom = d.solve(mu)
print('eta_fom', d.estimate(Ufom, mu=mu, decompose=True)[:2])
hist = []
def cb(rd_, U_, mu_, info):
    e, (nc, r, df), ind = rd_.estimate(U_, mu=mu_, decompose=True)
    err = (reductor.reconstruct(U_) - Ufom).data
    print(info['eta'], 'nc', np.linalg.norm(nc), 'r', np.linalg.norm(r), 'df', np.linalg.norm(df), 'solves', info['local_problem_solves'],
          'sizes', info['local RB sizes'], 'err', np.abs(err).max() / np.abs(Ufom.data).max())
loop = AdaptiveEnrichment(p, d, data['block_space'], reductor, rd, target_error=0.1, marking_doerfler_theta=0.8, marking_max_age=2)
loop.solve(mu, enrichment_steps=6, callback=cb)
