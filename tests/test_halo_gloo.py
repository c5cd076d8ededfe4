"""The N > 1 path on CPU: world_size-2 (and 4) gloo processes exchange halo basis rows with HaloExchange and check
that every row the kernels read from a neighbour's slab equals the owner's row."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pylrbms_amd.grid import DDSubdomainsGrid
from pylrbms_amd.parallel import HaloExchange, HaloPlan, side_rows

P, KC, N = (4, 4), 2, 3


def _global_V(grid):
    rng = np.random.default_rng(1234)
    return rng.standard_normal((grid.num_subdomains, grid.template.n, N))


def _worker(rank, world, port, results, mode):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        mk = lambda r: DDSubdomainsGrid([0, 0], [1, 1], (P[0] * KC, P[1] * KC), P, rank=r, world_size=world)  # noqa: E731
        grid = mk(rank)
        plan = HaloPlan(mk, world, rank)
        Vg = _global_V(grid)
        local = grid.subdomains_on_rank
        halo = sorted({j for s in local for j in grid.neighboring_subdomains(s)} - set(local))
        V = torch.zeros(len(local) + len(halo), grid.template.n, N, dtype=torch.float64)
        V[:len(local)] = torch.from_numpy(Vg[local])
        HaloExchange(plan, N, V.device, mode=mode)(V)
        rows = side_rows(grid.template)
        ok = True
        checked = 0
        for h, s in enumerate(halo):
            for sd in range(4):
                j = grid.neighbor_slots[s, (0, 1, 3, 4)[sd]]
                if j >= 0 and int(j) in local:
                    got = V[len(local) + h, rows[sd]].numpy()
                    ok &= bool(np.array_equal(got, Vg[s][rows[sd]]))
                    checked += 1
        ok &= bool(np.array_equal(V[:len(local)].numpy(), Vg[local]))
        # fused estimator norms (estimators.py:100-101)
        from pylrbms_amd.parallel import global_norms
        a = torch.tensor([float(s) for s in local], dtype=torch.float64)
        norms = global_norms(a, 2 * a)
        want = np.sqrt(sum(s * s for s in range(grid.num_subdomains)))
        ok &= abs(float(norms[0]) - want) < 1e-12 and abs(float(norms[1]) - 2 * want) < 1e-12
        # gather of per-subdomain rows into global order (the reduced system of a sharded run, rd.solve)
        from pylrbms_amd.parallel import gather_subdomain_rows
        owned = [list(mk(r).subdomains_on_rank) for r in range(world)]
        rows = torch.tensor([[float(s), 10.0 * s] for s in local], dtype=torch.float64)
        allrows = gather_subdomain_rows(rows, owned, grid.num_subdomains)
        want_rows = np.array([[float(s), 10.0 * s] for s in range(grid.num_subdomains)])
        ok &= bool(np.array_equal(allrows.numpy(), want_rows))
        results[rank] = (ok, checked)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,mode', [(2, 'alltoall'), (4, 'alltoall'), (2, 'allgather'), (4, 'allgather')])
def test_halo_exchange_gloo(world, mode):
    port = 27500 + (os.getpid() % 1000) + world + (10 if mode == "allgather" else 0)
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, results, mode), nprocs=world, join=True)
    assert len(results) == world
    for r in range(world):
        ok, checked = results[r]
        assert ok and checked > 0


def _diag_worker(rank, world, port, results):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pylrbms_amd.parallel import corner_rows
        mk = lambda r: DDSubdomainsGrid([0, 0], [1, 1], (P[0] * KC, P[1] * KC), P, rank=r, world_size=world)  # noqa: E731
        grid = mk(rank)
        Vg = _global_V(grid)
        local = grid.subdomains_on_rank
        halo = grid.halo_subdomains(diagonal=True)
        crow = corner_rows(grid.template)
        ok, checked, corner_items = True, 0, 0
        for mode in ('alltoall', 'allgather'):
            plan = HaloPlan(mk, world, rank, diagonal=True)
            assert plan.S_ext == len(local) + len(halo)
            corner_items = sum(1 for (_, kind) in plan.send_items[rank] if kind >= 4)
            V = torch.full((len(local) + len(halo), grid.template.n, N), float('nan'), dtype=torch.float64)
            V[:len(local)] = torch.from_numpy(Vg[local])
            HaloExchange(plan, N, V.device, mode=mode)(V)
            hpos = {g: len(local) + i for i, g in enumerate(halo)}
            for s in local:                                           # what the vertex patch reads of a diagonal neighbour
                for c, dgn in enumerate(grid.diagonal_neighbors(s)):
                    if dgn >= 0 and dgn not in local:
                        rows = crow[3 - c]                            # the opposite corner of the diagonal subdomain
                        ok &= bool(np.array_equal(V[hpos[dgn], rows].numpy(), Vg[dgn][rows]))
                        checked += 1
        results[rank] = (ok, checked, corner_items)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_halo_exchange_with_diagonal_neighbours_gloo(world):
    """conventions oswald_vertex_patch on a sharded grid: the halo also holds the diagonal neighbours, and their DoF rows at the
    shared cross point arrive -- with a side item where one passes by anyway, as a corner item of their own (two rows) where four
    tiles meet (world 4: 2 x 2 tiles of the 4 x 4 grid have one such cross point, every rank sends exactly one corner item)."""
    port = 27200 + (os.getpid() % 1000) + world
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_diag_worker, args=(world, port, results), nprocs=world, join=True)
    assert len(results) == world
    for r in range(world):
        ok, checked, corner_items = results[r]
        assert ok and checked > 0
        assert corner_items == (1 if world == 4 else 0)


def test_side_rows_cover_what_the_kernels_read():
    t = DDSubdomainsGrid([0, 0], [1, 1], (8, 8), (2, 2)).template
    rows = side_rows(t)
    opposite = {0: 3, 1: 2, 2: 1, 3: 0}
    for sd in range(4):
        need = set()
        osd = opposite[sd]           # the neighbour across our side `sd` shows us its side `osd`
        for p in range(t.side_count[sd]):
            e = t.side_elem_out[sd, p]
            need |= {3 * e, 3 * e + 1, 3 * e + 2}            # flux reconstruction + coupling blocks
        nvx, nvy = t.nvx, t.nvy
        for v in range(t.n_vertices):
            lx, ly = v % nvx, v // nvx
            on = [ly == 0, lx == 0, lx == nvx - 1, ly == nvy - 1][osd]
            if on:
                need |= set(int(i) for i in t.vdof_idx[t.vdof_ptr[v]:t.vdof_ptr[v + 1]])   # Oswald vertex stars
        assert need <= set(int(r) for r in rows[osd])


def test_alltoall_plan_is_consistent_across_ranks():
    """What rank a sends to rank b is, row for row, what rank b expects from rank a (no handshake at run time)."""
    world = 8
    mk = lambda r: DDSubdomainsGrid([0, 0], [1, 1], (32, 32), (8, 8), rank=r, world_size=world)  # noqa: E731
    plans = [HaloPlan(mk, world, r) for r in range(world)]
    for a in range(world):
        assert plans[a].a2a_send_splits[a] == 0 and plans[a].a2a_recv_splits[a] == 0
        assert sum(plans[a].a2a_send_splits) == len(plans[a].a2a_pack_index)
        assert sum(plans[a].a2a_recv_splits) == len(plans[a].a2a_unpack_dst)
        for b in range(world):
            assert plans[a].a2a_send_splits[b] == plans[b].a2a_recv_splits[a]
        # point-to-point volume is far below the all-gather volume
        assert sum(plans[a].a2a_recv_splits) < plans[a].max_rows * (world - 1)


def _loopback_case(device, backend, port):
    """One rank, the plan of rank 0 of a two-tile partition: every packed row goes through the asynchronous
    all_to_all to the rank itself and lands in the halo slabs (the collective branch of HaloExchange.start / finish)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    if backend == 'nccl':
        from pylrbms_amd.parallel import init_rccl
        init_rccl(torch.device(device), rank=0, world_size=1)          # what bench.py calls on every rank
    else:
        dist.init_process_group(backend, rank=0, world_size=1)
    try:
        mk = lambda r: DDSubdomainsGrid([0, 0], [1, 1], (P[0] * KC, P[1] * KC), P, rank=r, world_size=2)  # noqa: E731
        grid = mk(0)
        plan = HaloPlan(mk, 2, 0)
        local = grid.subdomains_on_rank
        halo = sorted({j for s in local for j in grid.neighboring_subdomains(s)} - set(local))
        Vg = _global_V(grid)
        V = torch.full((len(local) + len(halo), grid.template.n, N), float('nan'), dtype=torch.float64, device=device)
        V[:len(local)] = torch.from_numpy(Vg[local]).to(device)
        hx = HaloExchange(plan, N, V.device, loopback=True)
        assert hx.send_bytes == hx.recv_bytes == 8 * N * len(plan.a2a_pack_index) > 0
        finish = hx.start(V)
        # work enqueued between start and finish may only touch the local slabs (it overlaps with the collective)
        checksum = V[:len(local)].sum()
        finish()
        flat = V.view(-1, N)
        sent = flat[torch.from_numpy(plan.a2a_pack_index).to(device)]
        got = flat[torch.from_numpy(plan.a2a_unpack_dst).to(device)]
        assert torch.equal(sent, got)                                    # every halo row the kernels read was delivered
        assert torch.isfinite(checksum)
        assert torch.equal(V[:len(local)].cpu(), torch.from_numpy(Vg[local]))
        # second exchange on the same buffers (steady state of bench.py's loop)
        V[len(local):] = float('nan')
        hx(V)
        assert torch.equal(flat[torch.from_numpy(plan.a2a_unpack_dst).to(device)], sent)
    finally:
        dist.destroy_process_group()


def test_collective_branch_runs_on_one_rank_gloo():
    _loopback_case('cpu', 'gloo', 29700 + (os.getpid() % 100))


@pytest.mark.gpu
def test_collective_branch_runs_on_one_rank_rccl():
    """The same under the nccl (= RCCL) backend on one MI355X: the asynchronous all_to_all_single with split sizes, its
    work.wait() as a device-side wait on the current stream, pack / unpack on the GPU -- the branch the 8-GPU run takes,
    exercised once without an 8-GPU node (world_size 1: RCCL's self exchange)."""
    _loopback_case('cuda:0', 'nccl', 29800 + (os.getpid() % 100))
