"""Device-resident stand-ins for the pyMOR vector-array types the path touches
(``BlockVectorSpace`` / ``ListVectorArray`` / ``NumpyVectorArray``: reference
discretize_elliptic_block_swipdg.py:61,80,142; reductor.py:9; online_adaptive_lrbms.py:92,113,141).

A ``BlockVectorArray`` holds ``len(U)`` vectors of a block space with equally sized blocks as ONE torch tensor
``[num_blocks, block_dim, len]`` in HBM (vector index contiguous) -- the layout the HIP kernels consume as a basis.
Only the few operations the reference's scripts use are provided."""
import numpy as np


class SubSpace:
    def __init__(self, dim, id_):
        self.dim, self.id = int(dim), id_


class BlockVectorSpace:
    def __init__(self, subspaces, id_=None):
        self.subspaces = list(subspaces)
        self.id = id_
        self.dim = sum(s.dim for s in self.subspaces)

    def __eq__(self, other):
        return isinstance(other, BlockVectorSpace) and [s.dim for s in self.subspaces] == [s.dim for s in other.subspaces]

    def zeros(self, count, ctx):
        return BlockVectorArray(ctx.zeros(len(self.subspaces), self.subspaces[0].dim, count), self)

    def from_data(self, data, ctx):
        data = np.atleast_2d(np.asarray(data, dtype=np.float64))
        nb, bd = len(self.subspaces), self.subspaces[0].dim
        t = ctx.from_numpy(data.reshape(data.shape[0], nb, bd).transpose(1, 2, 0))
        return BlockVectorArray(t, self)


class BlockVectorArray:
    def __init__(self, tensor, space):
        assert tensor.dim() == 3
        self._t = tensor.contiguous()
        self.space = space

    def __len__(self):
        return int(self._t.shape[2])

    @property
    def dim(self):
        return int(self._t.shape[0] * self._t.shape[1])

    @property
    def tensor(self):
        """[num_blocks, block_dim, len] device tensor (no copy)."""
        return self._t

    @property
    def data(self):
        """NumPy ``[len, dim]`` copy, block-mapper ordering (what ``U.data`` is in the reference)."""
        return self._t.permute(2, 0, 1).reshape(len(self), -1).cpu().numpy()

    def block(self, ii):
        return BlockVectorArray(self._t[ii:ii + 1], BlockVectorSpace([self.space.subspaces[ii]]))

    def copy(self):
        return BlockVectorArray(self._t.clone(), self.space)

    def append(self, other):
        import torch
        assert other.space == self.space
        self._t = torch.cat([self._t, other._t], dim=2).contiguous()

    def __getitem__(self, idx):
        idx = [idx] if np.isscalar(idx) else list(range(len(self)))[idx] if isinstance(idx, slice) else list(idx)
        return BlockVectorArray(self._t[:, :, idx], self.space)

    def _binary(self, other, op):
        assert other.space == self.space and (len(other) == len(self) or len(other) == 1 or len(self) == 1)
        return BlockVectorArray(op(self._t, other._t), self.space)

    def __add__(self, other):
        return self._binary(other, lambda a, b: a + b)

    def __sub__(self, other):
        return self._binary(other, lambda a, b: a - b)

    def __mul__(self, scalar):
        return BlockVectorArray(self._t * float(scalar), self.space)

    __rmul__ = __mul__

    def sup_norm(self):
        return self._t.abs().amax(dim=(0, 1)).cpu().numpy()

    def l2_norm(self):
        """Euclidean norm of the coefficient vectors (pyMOR ``l2_norm``; python/scripts/parabolic.py:71)."""
        return (self._t ** 2).sum(dim=(0, 1)).sqrt().cpu().numpy()


class ReducedVectorArray:
    """Reduced coefficient vectors ``[num_blocks, N, len]`` (what ``rd.solve`` returns)."""

    def __init__(self, tensor):
        assert tensor.dim() == 3
        self._t = tensor.contiguous()

    def __len__(self):
        return int(self._t.shape[2])

    @property
    def tensor(self):
        return self._t

    @property
    def data(self):
        return self._t.permute(2, 0, 1).reshape(len(self), -1).cpu().numpy()

    def append(self, other):
        import torch
        self._t = torch.cat([self._t, other._t], dim=2).contiguous()

    def __getitem__(self, idx):
        idx = [idx] if np.isscalar(idx) else list(range(len(self)))[idx] if isinstance(idx, slice) else list(idx)
        return ReducedVectorArray(self._t[:, :, idx])

    def __sub__(self, other):
        return ReducedVectorArray(self._t - other._t)

    def __add__(self, other):
        return ReducedVectorArray(self._t + other._t)
