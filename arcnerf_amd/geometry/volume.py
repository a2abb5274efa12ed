"""Volume: the axis-aligned occupancy grid bounding the object (the part of arcnerf/geometry/volume.py the hot path
uses: range, voxel index maths, bool bitfield + fp32 opacity field and their update, ray/volume intersection).

Differences by design (MI355X, 288 GB is not a reason to stream 50 MB of lattice points per step):
  - `grid_pts` ((n+1)^3,3) and `volume_pts` (n^3,3) are computed on demand from indices instead of being resident
    buffers that DDP re-broadcasts every forward;
  - voxel flat index is x*n*n + y*n + z everywhere (volume_func.h:59-66), the bitfield is kept flat + bool like the
    reference (checkpoint compatible) and a packed 1-bit copy is derived for the marcher.
"""

import torch
import torch.nn as nn

from ..ops import functional as F
from .ray import aabb_ray_intersection


def mix_constants(n, rng):
    """the two (multiplier, increment) pairs of mix_permutation for a cell range of n = 2^k, drawn from `rng` (multipliers odd and
    not tiny); also what arcn_refresh_cells_points takes as perm_a / perm_c"""
    k = n.bit_length() - 1
    out = []
    for _ in range(2):
        a = int(rng.integers(0, n // 2)) * 2 + 1
        a |= 1 << max(1, k // 2)           # keep the multiplier from being tiny
        c = int(rng.integers(0, n))
        out.append((a & (n - 1), c))
    return out


def mix_permutation(idx, n, rng):
    """idx (int64 tensor of values in [0, n), n a power of two) -> pi(idx) for a bijection pi of [0, n) drawn from `rng`:
    x -> a*x + c (a odd), x -> x ^ (x >> s): both invertible modulo 2^k; two rounds with independent constants."""
    k = n.bit_length() - 1
    m = n - 1
    x = idx
    for a, c in (rng if isinstance(rng, list) else mix_constants(n, rng)):
        x = (x * a + c) & m
        x = x ^ (x >> max(1, (k + 1) // 2))
        x = (x * 0x9E3779B1) & m           # odd constant: a second multiply after the shift spreads the high bits
        x = x ^ (x >> max(1, k // 3))
    return x


# The random draws of an occupancy refresh (VolumeBound.optimize, volume_bound.py:178-193: torch.randperm for the uniformly chosen quarter,
# torch.rand_like for the jitter inside a cell).  By default they come from seeded on-device generators (mix_permutation, the pcg32 of
# arcn_refresh_cells_points, torch.rand_like).  A tape installed with set_refresh_tape() supplies them instead: an object with
# draws(cur_epoch, n_cells, device) -> (perm int64 (n_cells,), uniforms float32 (n_cells, 3)) - e.g. the draws a run of the reference's
# own loop consumed, so that the run can be reproduced cell for cell (tests/test_gpu_trajectory.py).
_refresh_tape = None


def set_refresh_tape(tape):
    global _refresh_tape
    _refresh_tape = tape


def refresh_tape():
    return _refresh_tape


def select_refresh_cells(bitfield_flat, n_cells, cache, rng, perm=None):
    """Cells refreshed by VolumeBound.optimize after its warm-up (volume_bound.py:178-190): n/4 cells drawn uniformly
    without repetition + the first n/4 occupied cells in flat-index order (`get_occupied_voxel_idx()[:n]`).

    Sync-free: the uniform part is the first n/4 images of a seeded BIJECTION of the (power-of-two) cell range - two rounds of
    odd-multiply / add / xor-shift, each a permutation of [0, 2^k) - instead of torch.randperm (a 2M-key sort); a plain affine map
    a*i + c would be an arithmetic progression (one contiguous quarter of the grid for a = 1), the mixing rounds make the subset
    look uniform in (x, y, z) (tests/test_host_api.py checks the per-octant / per-slab counts).  The occupied part is an ordered
    compaction through cumsum + scatter, and the number
    of valid entries is returned as a DEVICE int32 scalar — torch.nonzero/torch.where would stall the launch queue at every
    refresh.  Returns (cells int64 (2*(n/4),), n_valid int32 (1,)).  `cache` is a dict for persistent buffers, `rng` a
    numpy Generator (host side, only the two affine constants are drawn from it); perm: a given permutation of the cells (a tape)
    instead of the seeded bijection."""
    dev = bitfield_flat.device
    n_s = n_cells // 4
    if 'cell_buf' not in cache:
        cache['cell_buf'] = torch.zeros(2 * n_s + 1, dtype=torch.int64, device=dev)
        cache['arange'] = torch.arange(n_cells, device=dev)
    buf, ar = cache['cell_buf'], cache['arange']
    if perm is not None:
        buf[:n_s] = perm[:n_s]
    elif n_cells & (n_cells - 1) == 0:
        buf[:n_s] = mix_permutation(ar[:n_s], n_cells, rng)
    else:
        buf[:n_s] = torch.randperm(n_cells, device=dev)[:n_s]
    csum = torch.cumsum(bitfield_flat.to(torch.int32), 0)
    dst = torch.where(bitfield_flat & (csum <= n_s), csum.long() + (n_s - 1), torch.full_like(ar, 2 * n_s))
    buf.scatter_(0, dst, ar)
    n_valid = (torch.clamp(csum[-1:], max=n_s) + n_s).to(torch.int32)
    return buf[:2 * n_s], n_valid


class Volume(nn.Module):
    def __init__(self, n_grid=None, origin=(0, 0, 0), side=None, xyz_len=None, dtype=torch.float32, requires_grad=False,
                 **kwargs):
        super().__init__()
        self.n_grid = n_grid
        self.dtype = dtype if isinstance(dtype, torch.dtype) else torch.float32
        self.contains_bitfield = False
        assert side is not None or xyz_len is not None, 'Specify at least side or xyz_len'
        lens = [float(side)] * 3 if side is not None else [float(v) for v in xyz_len]
        self.origin = nn.Parameter(torch.tensor([float(v) for v in origin], dtype=torch.float32), requires_grad=requires_grad)
        self.xyz_len = nn.Parameter(torch.tensor(lens, dtype=torch.float32), requires_grad=requires_grad)
        self.cal_range()

    # The reference keeps three derived point sets as buffers (corner (8,3), grid_pts ((n+1)^3,3), volume_pts (n^3,3):
    # geometry/volume.py:200-232 there), 50 MB at n_grid 128 that nothing on this path reads; they are recomputed on demand
    # here (get_volume_pts ...).  A reference checkpoint carries them: accept and drop them so it loads with strict=True.
    DERIVED_REFERENCE_BUFFERS = ('corner', 'grid_pts', 'volume_pts')

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for name in self.DERIVED_REFERENCE_BUFFERS:
            state_dict.pop(prefix + name, None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    # ---- geometry -----------------------------------------------------------------------------------
    def cal_range(self):
        mn = self.origin.detach() - self.xyz_len.detach() / 2.0
        mx = self.origin.detach() + self.xyz_len.detach() / 2.0
        self.register_buffer('range', torch.stack([mn, mx], dim=-1))  # (3, 2)
        self.__dict__.pop('_host_cache', None)    # (a re-registered buffer may reuse the address AND version of the one it replaces)

    def get_range(self):
        return self.range

    def get_range23(self):
        """the (2, 3) [min | max] form of the range the sampling kernels take, kept while the range tensor is the same object at the same
        version (a (3, 2) -> (2, 3) transposed copy per sampler launch otherwise)"""
        r = self.range
        c = getattr(self, '_range23', None)
        if c is None or c[0] is not r or c[1] != r._version or c[2].device != r.device:
            c = self._range23 = (r, r._version, r.permute(1, 0).contiguous())
        return c[2]

    def get_device(self):
        return self.origin.device

    def get_n_grid(self):
        return self.n_grid

    def set_n_grid(self, n_grid, reset_pts=True):
        self.n_grid = n_grid

    def get_n_voxel(self):
        return self.n_grid ** 3

    def _host_value(self, name, src, make):
        """A Python-number view of a small device tensor, read back ONCE per version of the tensor: every float() of a device scalar is a
        host / device synchronisation, and the samplers ask for the diagonal on every call (the reference re-reads it each time,
        geometry/volume.py:330-360 there).  In-place writers and re-registered buffers move the key."""
        key = (src.data_ptr(), src._version, str(src.device))
        if src.requires_grad:      # a learnable tensor is updated by raw-pointer kernels (FusedAdam) that do not move its version: no cache
            return make()
        cache = self.__dict__.setdefault('_host_cache', {})
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            hit = cache[name] = (key, make())
        return hit[1]

    def _range_key(self):
        """the tensor whose version keys the cached values derived from `range`: a learnable origin / xyz_len turns the cache off"""
        return self.origin if self.origin.requires_grad else (self.xyz_len if self.xyz_len.requires_grad else self.range)

    def get_len(self):
        return self._host_value('len', self.xyz_len, lambda: tuple(float(v) for v in self.xyz_len.detach().tolist()))

    def get_origin(self):
        return self.origin

    def get_diag_len(self):
        return self._host_value('diag', self._range_key(), lambda: float(torch.sqrt(((self.range[:, 1] - self.range[:, 0]) ** 2).sum())))

    def get_voxel_size(self, to_list=True):
        if to_list:
            return self._host_value('voxel%d' % self.n_grid, self._range_key(),
                                    lambda: tuple(float(v) for v in ((self.range[:, 1] - self.range[:, 0]) / self.n_grid).tolist()))
        return (self.range[:, 1] - self.range[:, 0]) / self.n_grid

    @staticmethod
    def convert_flatten_index_to_xyz_index(flat, n):
        z = flat % n
        y = torch.div(flat, n, rounding_mode='trunc') % n
        x = torch.div(flat, n * n, rounding_mode='trunc')
        return torch.stack([x, y, z], dim=-1)

    @staticmethod
    def convert_xyz_index_to_flatten_index(xyz, n):
        return xyz[:, 0] * (n * n) + xyz[:, 1] * n + xyz[:, 2]

    def get_full_voxel_idx(self, flatten=False):
        idx = self.convert_flatten_index_to_xyz_index(torch.arange(self.get_n_voxel(), device=self.get_device()), self.n_grid)
        return idx if flatten else idx.view(self.n_grid, self.n_grid, self.n_grid, 3)

    def get_voxel_pts_by_voxel_idx(self, voxel_idx):
        vs = self.get_voxel_size(to_list=False)
        return voxel_idx * vs + 0.5 * vs + self.range[:, 0]

    def get_volume_pts(self, in_grid=False):
        """voxel centres (n^3, 3): linspace(min + v/2, max - v/2, n) per axis like cal_volume_pts (volume.py:216-225)"""
        vs = self.get_voxel_size()
        ax = [torch.linspace(float(self.range[k, 0]) + 0.5 * vs[k], float(self.range[k, 1]) - 0.5 * vs[k], self.n_grid,
                             device=self.get_device()) for k in range(3)]
        pts = torch.stack(torch.meshgrid(*ax, indexing='ij'), -1)
        return pts if in_grid else pts.view(-1, 3)

    def ray_volume_intersection(self, rays_o, rays_d, in_occ_voxel=False, force=False):
        """near, far (N_rays,1), pts (N_rays,2,3), mask (N_rays,1) against the outer box (volume.py:624-651)"""
        assert not in_occ_voxel, 'per-voxel intersection is outside the hot path'
        near, far, pts, mask = aabb_ray_intersection(rays_o, rays_d, self.range[None].to(rays_o.device))
        return near, far, pts[:, 0], mask

    # ---- occupancy ----------------------------------------------------------------------------------
    def set_up_voxel_bitfield(self, init_occ=True):
        self.contains_bitfield = True
        fill = torch.ones if init_occ else torch.zeros
        self.register_buffer('bitfield', fill((self.n_grid,) * 3, dtype=torch.bool))

    def set_up_voxel_opafield(self):
        self.register_buffer('opafield', torch.zeros((self.n_grid,) * 3, dtype=torch.float32))

    def get_voxel_bitfield(self, flatten=False):
        if not self.contains_bitfield:
            return None
        return self.bitfield.view(-1) if flatten else self.bitfield

    def get_voxel_opafield(self, flatten=False):
        return self.opafield.view(-1) if flatten else self.opafield

    def reset_voxel_bitfield(self, occ=True):
        self.bitfield.fill_(bool(occ))

    def update_bitfield(self, occupancy, ops='and'):
        occ = occupancy.view_as(self.bitfield)
        if ops == 'and':
            self.bitfield &= occ
        elif ops == 'or':
            self.bitfield |= occ
        elif ops == 'overwrite':
            self.bitfield.copy_(occ)
        else:
            raise NotImplementedError('Invalid ops {}'.format(ops))
        return self.bitfield

    def get_n_occupied_voxel(self):
        return self.bitfield.sum()

    def get_occupied_voxel_idx(self, flatten=False):
        idx = torch.where(self.bitfield.view(-1))[0]
        return idx if flatten else self.convert_flatten_index_to_xyz_index(idx, self.n_grid)

    def check_pts_in_occ_voxel(self, pts):
        """(B,) bool: pts inside an occupied voxel (K1)"""
        return F.check_pts_in_occ_voxel(pts, self.bitfield, self.range.permute(1, 0).contiguous(), self.n_grid)

    def update_opafield_by_voxel_idx(self, voxel_idx, opacity, ema=None):
        """opafield[idx] = max(old*ema, new) where old >= 0; voxel_idx (B,3) without repetition (volume.py:983-1003)"""
        flat = self.convert_xyz_index_to_flatten_index(voxel_idx, self.n_grid)
        F.update_opafield(self.opafield.view(-1), flat, opacity, ema)

    def update_opafield_by_flat_idx(self, flat_idx, opacity, ema=None, n_dev=None):
        """same as unique + segmented max + update_opafield_by_voxel_idx for possibly repeated cells, sort-free;
        n_dev (device int32) = number of valid entries"""
        F.opafield_scatter_update(self.opafield.view(-1), flat_idx, opacity, ema, n_dev=n_dev)

    def get_mean_voxel_opacity(self):
        return float(self.opafield.clamp(min=0).mean())

    def update_bitfield_by_opafield(self, threshold=0.01, ops='and'):
        """bitfield (ops) (opafield >= min(mean(clamp(opa,0)), threshold))  (volume.py:1013-1017), no host sync"""
        new = torch.empty_like(self.bitfield)
        F.update_bitfield_by_opafield(self.opafield.view(-1), new.view(-1), threshold)
        self.update_bitfield(new, ops)
