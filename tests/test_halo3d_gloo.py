"""The N > 1 path of the 3D / P2 configuration (BASELINE.json config 5: 8 x 8 x 8 subdomains over 8 ranks) on the CPU:
world_size-2 and -4 gloo processes exchange the halo rows of a 3D tile partition with the same HaloExchange as in 2D; every row
the 3D kernels read from a neighbour's slab must equal the owner's row."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pylrbms_amd.grid3d import SIDE_TO_SLOT, DDSubdomainsGrid3D, tile_grid3d
from pylrbms_amd.parallel import HaloExchange, HaloPlan, side_rows3d

P, KC, N = (4, 2, 2), 2, 3


def _mk(world):
    return lambda r: DDSubdomainsGrid3D([0] * 3, [1] * 3, [p * KC for p in P], P, rank=r, world_size=world)


def _worker(rank, world, port, results):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        mk = _mk(world)
        grid = mk(rank)
        plan = HaloPlan(mk, world, rank)
        Vg = np.random.default_rng(99).standard_normal((grid.num_subdomains, grid.template.n, N))
        local = grid.subdomains_on_rank
        halo = sorted({j for s in local for j in grid.neighboring_subdomains(s)} - set(local))
        V = torch.zeros(len(local) + len(halo), grid.template.n, N, dtype=torch.float64)
        V[:len(local)] = torch.from_numpy(Vg[local])
        HaloExchange(plan, N, V.device)(V)
        rows = side_rows3d(grid.template)
        ok, checked = True, 0
        for h, s in enumerate(halo):
            for sd in range(6):
                j = grid.neighbor_slots[s, SIDE_TO_SLOT[sd]]
                if j >= 0 and int(j) in local:
                    ok &= bool(np.array_equal(V[len(local) + h, rows[sd]].numpy(), Vg[s][rows[sd]]))
                    checked += 1
        ok &= bool(np.array_equal(V[:len(local)].numpy(), Vg[local]))
        results[rank] = (ok, checked, plan.a2a_send_splits, plan.a2a_recv_splits)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_halo_exchange_3d_gloo(world):
    port = 28700 + (os.getpid() % 1000) + world
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
    assert len(results) == world
    for r in range(world):
        ok, checked, send, recv = results[r]
        assert ok and checked > 0
        for r2 in range(world):                      # what r sends to r2 is what r2 expects from r
            assert send[r2] == results[r2][3][r]


def test_side_rows_cover_what_the_3d_kernels_read():
    t = DDSubdomainsGrid3D([0] * 3, [1] * 3, [4, 4, 4], [2, 2, 2]).template
    rows = side_rows3d(t)
    for sd in range(6):
        osd = 5 - sd                                  # the neighbour across our side sd shows us its side osd
        need = set()
        for p in range(t.side_count[sd]):             # coupling blocks and flux image: the neighbour's side elements
            e = t.side_elem_out[sd, p]
            need |= {10 * e + i for i in range(10)}
        for p in range(t.nvs):                        # node averages: the neighbour's DoFs at the shared nodes
            sp = sd * t.nvs + p
            need |= {int(d) for d in t.sn_dofs[t.sn_ptr[sp]:t.sn_ptr[sp + 1]]}
        assert need <= set(rows[osd].tolist()), sd


def test_config5_partition_over_eight_ranks():
    assert tile_grid3d(8, [8, 8, 8]) == (2, 2, 2)
    plans = [HaloPlan(lambda r: DDSubdomainsGrid3D([0] * 3, [1] * 3, [8] * 3, [8] * 3, rank=r, world_size=8), 8, r) for r in (0, 7)]
    for pl in plans:
        assert pl.S == 64 and pl.S_ext == 64 + 3 * 16          # a corner tile of 4 x 4 x 4 subdomains: three neighbour tiles
        assert sum(pl.a2a_send_splits) == sum(pl.a2a_recv_splits) == 3 * 16 * 60      # k_c = 1 here: 6 elements x 10 rows per side layer
