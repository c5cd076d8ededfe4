"""Multi-GPU sharding of the hot path: one process per GPU, subdomains tiled over ranks, ONE exchange step.

The reference's only parallelism is MPI domain decomposition: subdomains are the unit of
independent work (discretize_elliptic_block_swipdg.py:66-70, estimators.py:70) and the two ``mpi_norm`` calls at
estimators.py:100-101 are its only live collectives on this path.  Here:

* every rank owns a contiguous 2D tile of subdomains (``DDSubdomainsGrid._partition``) and computes everything for
  its own target subdomains ``ii`` -- including the images W^{kk}|_ii, R^{kk}|_ii of its neighbours' bases -- so
  the projection needs a single halo exchange of neighbour basis rows and no second exchange;
* the halo carries only the DoF rows of elements that touch the shared side (what the coupling blocks, the Oswald
  vertex stars and the flux reconstruction read), packed into one buffer and moved by one RCCL all-to-all with
  per-peer splits over xGMI (point-to-point: only the owner of the neighbour receives a row);
* the estimator norms (C2/C3) are one all-reduce of 2 doubles per estimated vector.

Works with any torch.distributed backend (``nccl`` = RCCL on the GPUs, ``gloo`` in the CPU tests).
"""
import numpy as np


def init_rccl(device, **kw):
    """``torch.distributed.init_process_group('nccl', device_id=device, ...)`` with the collectives' stream taken from the
    HIGH-priority pool.  HIP maps the streams of a priority level onto four hardware queues; the caller's stream and the
    library's side streams (csrc/lrbms_dev.h) can fill them, and a stream that shares a queue runs behind that queue's kernels
    -- an exchange that lands behind the pass's long MFMA kernel overlaps with nothing.  High-priority
    streams have queues of their own, so the exchange never depends on the order in which streams were created."""
    import torch.distributed as dist
    try:
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    except Exception:                 # a build of torch without the option: the default stream pool
        opts = None
    dist.init_process_group('nccl', device_id=device, pg_options=opts, **kw)


class Communicator:
    """Minimal stand-in for the ``mpi_comm`` argument of the reference API (rank / size only)."""

    def __init__(self, rank=0, size=1, group=None):
        self.rank, self.size, self.group = rank, size, group

    @classmethod
    def from_torch_distributed(cls, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            return cls()
        return cls(dist.get_rank(group), dist.get_world_size(group), group)


def side_rows(template):
    """For each side 0..3 the sorted DoF rows of all elements having a vertex on that side."""
    t = template
    lat = t.tri_lattice
    on = [lat[:, :, 1] == 0, lat[:, :, 0] == 0, lat[:, :, 0] == 2 * t.kx, lat[:, :, 1] == 2 * t.ky]
    rows = []
    for sd in range(4):
        elems = np.nonzero(on[sd].any(axis=1))[0]
        rows.append(np.sort((3 * elems[:, None] + np.arange(3)[None, :]).ravel()).astype(np.int64))
    return rows


def side_rows3d(template):
    """3D (pylrbms_amd.grid3d.SubdomainTemplate3D): for each side 0..5 the sorted DoF rows of all elements with a Lagrange node
    on that side -- the cube layer next to it; what the coupling blocks, the node averages and the flux image of the pass read
    of a neighbour's basis."""
    t = template
    from pylrbms_amd.grid3d import NLOC, SIDE_AXIS, SIDE_DIR
    rows = []
    for a in range(6):
        ax = SIDE_AXIS[a]
        layer = 0 if SIDE_DIR[a] < 0 else t.kc[ax] - 1
        elems = np.nonzero(t.elem_cube[:, ax] == layer)[0]
        rows.append(np.sort((NLOC * elems[:, None] + np.arange(NLOC)[None, :]).ravel()).astype(np.int64))
    return rows


def corner_rows(template):
    """2D: for each corner 0 SW, 1 SE, 2 NW, 3 NE of the subdomain the sorted DoF rows sitting AT that lattice vertex (one per
    element of its star) -- all the Oswald vertex patch reads of a diagonal neighbour's basis."""
    t = template
    rows = []
    for c in range(4):
        v = ((c & 1) and t.nvx - 1) + t.nvx * ((c & 2) and t.nvy - 1)
        rows.append(np.sort(np.asarray(t.vdof_idx[t.vdof_ptr[v]:t.vdof_ptr[v + 1]], dtype=np.int64)))
    return rows


class HaloPlan:
    """Who sends which rows of which subdomain: derived by every rank from the grid partition alone (no handshake)."""

    def __init__(self, grid_factory, world_size, rank, diagonal=False):
        """``grid_factory(rank)`` returns the DDSubdomainsGrid (2D) or DDSubdomainsGrid3D as seen by ``rank`` (same global
        grid, other tile).  ``diagonal`` (2D, conventions oswald_vertex_patch): the halo also holds the diagonal neighbours, and a
        diagonal neighbour whose corner rows do not already travel with a side item sends them as an item of their own
        (kinds 4 .. 7 = corner 0 .. 3 behind the four sides): at a cross point of four ranks' tiles, two DoF rows per corner."""
        grids = [grid_factory(r) for r in range(world_size)]
        g = grids[rank]
        self.rank, self.world_size = rank, world_size
        t = g.template
        if getattr(g, 'dim', 2) == 3:
            rows = side_rows3d(t)
            slot_of_side = (0, 1, 2, 4, 5, 6)
        else:
            rows = side_rows(t)
            slot_of_side = (0, 1, 3, 4)
        nsides = len(rows)
        opposite = {sd: nsides - 1 - sd for sd in range(nsides)}
        owner = {}
        for r, gr in enumerate(grids):
            for s in gr.subdomains_on_rank:
                owner[s] = r
        diagonal = bool(diagonal) and getattr(g, 'dim', 2) == 2
        if diagonal:
            rows = list(rows) + corner_rows(t)
            corner_sides = ((0, 1), (0, 2), (3, 1), (3, 2))       # the two sides that meet in corner SW, SE, NW, NE

        def peer_of(s, kind):
            """Global id of the subdomain that reads item (s, kind): the face neighbour across side ``kind`` or the diagonal
            neighbour at corner ``kind - nsides``."""
            if kind < nsides:
                return int(g.neighbor_slots[s, slot_of_side[kind]])
            return int(g.diagonal_neighbors(s)[kind - nsides])
        # send list of rank r: for each owned subdomain s and each side whose neighbour lives elsewhere,
        # the rows of s touching that side.  Deterministic order: (s ascending, side ascending).
        self.send_items = []      # per rank: list of (subdomain, side)
        for r, gr in enumerate(grids):
            items = []
            for s in gr.subdomains_on_rank:
                for sd in range(nsides):
                    j = g.neighbor_slots[s, slot_of_side[sd]]
                    if j >= 0 and owner[int(j)] != r:
                        items.append((s, sd))
                if diagonal:
                    for c, dgn in enumerate(g.diagonal_neighbors(s)):
                        if dgn < 0 or owner[dgn] == r:
                            continue
                        # the corner rows are part of the side rows of both sides that meet there: they already reach owner[dgn] if
                        # that rank also owns the face neighbour of s across one of the two
                        if any(owner.get(int(g.neighbor_slots[s, slot_of_side[sd]]), -1) == owner[dgn] for sd in corner_sides[c]):
                            continue
                        items.append((s, nsides + c))
            self.send_items.append(items)
        self.row_counts = [len(rw) for rw in rows]
        self.rows = rows
        self.send_sizes = [sum(self.row_counts[sd] for (_, sd) in items) for items in self.send_items]
        self.max_rows = max(self.send_sizes) if self.send_sizes else 0
        # local gather index (into the flattened [S_ext * n] row space of the local V) for my own send buffer
        local = list(g.subdomains_on_rank)
        lpos = {s: i for i, s in enumerate(local)}
        if diagonal:
            halo = g.halo_subdomains(diagonal=True)
        else:
            halo = sorted({int(j) for s in local for j in g.neighboring_subdomains(s)} - set(local))
        hpos = {s: len(local) + i for i, s in enumerate(halo)}
        self.S, self.S_ext, self.n = len(local), len(local) + len(halo), t.n
        idx = []
        for (s, sd) in self.send_items[rank]:
            idx.append(lpos[s] * t.n + rows[sd])
        self.pack_index = np.concatenate(idx) if idx else np.zeros(0, dtype=np.int64)
        # scatter: for every other rank's send buffer, which of its rows land in my halo slabs and where
        src, dst = [], []
        for r in range(world_size):
            if r == rank:
                continue
            off = 0
            for (s, sd) in self.send_items[r]:
                cnt = self.row_counts[sd]
                j = peer_of(s, sd)
                if s in hpos and j in lpos:          # s is my halo because its neighbour j on that side is mine
                    src.append(r * self.max_rows + off + np.arange(cnt))
                    dst.append(hpos[s] * t.n + rows[sd])
                off += cnt
        self.unpack_src = np.concatenate(src) if src else np.zeros(0, dtype=np.int64)
        self.unpack_dst = np.concatenate(dst) if dst else np.zeros(0, dtype=np.int64)
        self.opposite = opposite
        # point-to-point form of the same exchange (one all_to_all with per-peer splits): every (s, side) item goes
        # to exactly one peer, the owner of the neighbour across that side.  Send buffer ordered by (peer, s, side).
        def dest(s, sd):
            return owner[peer_of(s, sd)]
        self.a2a_send_splits = [0] * world_size
        self.a2a_recv_splits = [0] * world_size
        pidx, udst = [], []
        for r in range(world_size):
            if r == rank:
                continue
            for (s, sd) in self.send_items[rank]:
                if dest(s, sd) == r:
                    pidx.append(lpos[s] * t.n + rows[sd])
                    self.a2a_send_splits[r] += self.row_counts[sd]
            for (s, sd) in self.send_items[r]:
                if dest(s, sd) == rank:
                    udst.append(hpos[s] * t.n + rows[sd])
                    self.a2a_recv_splits[r] += self.row_counts[sd]
        self.a2a_pack_index = np.concatenate(pidx) if pidx else np.zeros(0, dtype=np.int64)
        self.a2a_unpack_dst = np.concatenate(udst) if udst else np.zeros(0, dtype=np.int64)


class HaloExchange:
    """Fills the halo slabs V[S:] from the neighbours' owners with ONE collective per pass.

    ``mode='alltoall'`` (default): ``all_to_all_single`` with per-peer row splits -- every rank sends each packed row
    only to the one peer that reads it (point-to-point over xGMI, no traffic to non-neighbours; at the 8-GPU tile of
    config 3 this is 1.3 MB per rank instead of the 10 MB an all-gather delivers to everyone).
    ``mode='allgather'``: the same rows through one ``all_gather_into_tensor``."""

    def __init__(self, plan, N, device, dtype=None, group=None, mode=None, loopback=False):
        """``loopback=True`` (alltoall mode, test use): run the collective on a one-rank process group, every packed row
        sent to the rank itself and unpacked into its halo slabs -- exercises the asynchronous collective, its stream
        semantics and pack / unpack on a single GPU (needs as many packed rows as halo rows: symmetric tiles)."""
        import os
        import torch
        self.torch, self.plan, self.group = torch, plan, group
        self.loopback = bool(loopback)
        self.mode = mode or 'alltoall'
        if self.mode not in ('alltoall', 'allgather'):
            raise ValueError('halo mode must be alltoall or allgather')
        dtype = dtype or torch.float64
        if self.mode == 'allgather':
            self.send = torch.zeros(max(plan.max_rows, 1), N, dtype=dtype, device=device)
            self.recv = torch.zeros(plan.world_size * max(plan.max_rows, 1), N, dtype=dtype, device=device)
            self.pack_index = torch.from_numpy(plan.pack_index).to(device)
            self.unpack_src = torch.from_numpy(plan.unpack_src).to(device)
            self.unpack_dst = torch.from_numpy(plan.unpack_dst).to(device)
        else:
            self.send = torch.zeros(len(plan.a2a_pack_index), N, dtype=dtype, device=device)
            self.recv = torch.zeros(len(plan.a2a_unpack_dst), N, dtype=dtype, device=device)
            self.pack_index = torch.from_numpy(plan.a2a_pack_index).to(device)
            self.unpack_dst = torch.from_numpy(plan.a2a_unpack_dst).to(device)
        self.count = len(self.pack_index)
        if self.loopback and (self.mode != 'alltoall' or self.send.shape != self.recv.shape):
            raise ValueError('loopback needs alltoall mode and equally many packed and halo rows')

    @property
    def send_bytes(self):
        """Bytes this rank hands to the collective per exchange."""
        return int(self.send.numel() * self.send.element_size())

    @property
    def recv_bytes(self):
        return int(self.recv.numel() * self.recv.element_size())

    def start(self, V):
        """Pack and launch the collective asynchronously; returns ``finish()``, which makes the current stream wait for
        it and unpacks into the halo slabs of V.  Kernels enqueued between the two calls overlap with the exchange and
        may read the LOCAL slabs of V only (the halo slabs are written by ``finish``)."""
        import torch.distributed as dist
        torch = self.torch
        if self.plan.world_size == 1 and not self.loopback:
            return lambda: V
        flat = V.view(-1, V.shape[2])
        if self.count:
            torch.index_select(flat, 0, self.pack_index, out=self.send[:self.count])
        # gloo has no device collectives: stage through host memory (CPU rehearsals of the multi-rank path on one GPU)
        staged = self.send.is_cuda and dist.get_backend(self.group) == 'gloo'
        send, recv = (self.send.cpu(), torch.empty_like(self.recv, device='cpu')) if staged else (self.send, self.recv)
        if self.mode == 'allgather':
            work = dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
        else:
            if self.loopback:
                out_splits = in_splits = [int(send.shape[0])]
            else:
                out_splits, in_splits = self.plan.a2a_recv_splits, self.plan.a2a_send_splits
            work = dist.all_to_all_single(recv, send, output_split_sizes=out_splits, input_split_sizes=in_splits,
                                          group=self.group, async_op=True)

        def finish():
            work.wait()          # device-side wait on the current stream for RCCL; blocking for gloo
            if staged:
                self.recv.copy_(recv)
            if self.mode == 'allgather':
                if len(self.unpack_src):
                    flat.index_copy_(0, self.unpack_dst, self.recv.index_select(0, self.unpack_src))
            elif len(self.unpack_dst):
                flat.index_copy_(0, self.unpack_dst, self.recv)
            return V
        return finish

    def __call__(self, V):
        """V [S_ext, n, N] contiguous; rows of the halo slabs that the kernels read are overwritten in place."""
        return self.start(V)()


def global_norms(local_eta_nc, local_eta_r_plus_df, group=None):
    """The two ``mpi_norm`` collectives of estimators.py:100-101 fused into one all-reduce (sum of squares)."""
    import torch
    import torch.distributed as dist
    sq = torch.stack([(local_eta_nc ** 2).sum(), (local_eta_r_plus_df ** 2).sum()])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sq, group=group)
    return torch.sqrt(sq)


def gather_subdomain_rows(t_local, owned, total, group=None):
    """All-gather per-subdomain rows into global subdomain order.

    ``t_local`` [S_loc, ...] holds the rows of the subdomains ``owned[rank]`` (every rank passes the same ``owned``:
    one list of global subdomain indices per rank).  Returns [total, ...] on every rank.  Ranks may own different
    numbers of subdomains (rows are padded to the largest share for the collective)."""
    import torch
    import torch.distributed as dist
    world = len(owned)
    if world == 1:
        return t_local
    smax = max(len(o) for o in owned)
    pad = torch.zeros((smax,) + tuple(t_local.shape[1:]), dtype=t_local.dtype, device=t_local.device)
    pad[:t_local.shape[0]] = t_local
    staged = pad.is_cuda and dist.get_backend(group) == 'gloo'     # gloo has no device all-gather
    src = pad.cpu() if staged else pad
    out = torch.empty((world * smax,) + tuple(pad.shape[1:]), dtype=pad.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src.contiguous(), group=group)
    if staged:
        out = out.to(pad.device)
    res = torch.empty((total,) + tuple(pad.shape[1:]), dtype=pad.dtype, device=pad.device)
    for r, ids in enumerate(owned):
        if len(ids):
            res[torch.as_tensor(list(ids), device=res.device)] = out[r * smax:r * smax + len(ids)]
    return res
