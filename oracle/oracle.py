"""ORACLE — TEST INFRASTRUCTURE ONLY (numpy front-end of oracle/_build/liborc.so).

CPU restatement of the reference's volumetric-rendering hot path; every C function cites the
reference file:line it follows (see oracle/src/*.c).  Allowed importers: tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never the product package.

Pinning: tests/test_oracle_golden.py checks this against tests/golden/*.npz, which were produced
by importing the reference's torch path (tests/golden/make_golden.py).  The CUDA-only sampler
(K3) and segmented max (K4) have no runnable reference in this image: "parity unpinned" for those
two, they are cross-checked through the torch-side invariants of SURVEY.md §8(c) instead.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'liborc.so')

ACT = {'none': 0, None: 0, 'relu': 1, 'sigmoid': 2, 'truncexp': 3, 'softplus': 4}


def build(force=False):
    """Compile the C restatement (gcc, a second or two)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(['make', '-C', _HERE] + (['-B'] if force else []), stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_update_bitfield_by_opafield.restype = C.c_float
        _lib.orc_ray_marching_fwd.restype = C.c_int
        _lib.orc_get_max_threads.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_num_threads(n):
    lib().orc_set_num_threads(C.c_int(int(n)))


def get_max_threads():
    return int(lib().orc_get_max_threads())


# ------------------------------------------------------------------------------------------------
# pcg32
# ------------------------------------------------------------------------------------------------
class Pcg32:
    """Host generator of arcnerf/ops/include/common.h:22-23 (`static pcg32 rng{9121}`)."""

    def __init__(self, seed=9121, seq=1):
        self.si = np.zeros(2, dtype=np.uint64)
        lib().orc_pcg32_init(C.c_uint64(seed), C.c_uint64(seq), _p(self.si))

    @property
    def state(self):
        return int(self.si[0])

    @property
    def inc(self):
        return int(self.si[1])

    def advance(self, delta=1 << 32):
        lib().orc_pcg32_advance_state(_p(self.si), C.c_int64(delta))

    def next_uint(self, n):
        out = np.zeros(n, dtype=np.uint32)
        lib().orc_pcg32_draw(_p(self.si), C.c_int(n), _p(out), None)
        return out

    def next_float(self, n):
        out = np.zeros(n, dtype=np.float32)
        lib().orc_pcg32_draw(_p(self.si), C.c_int(n), None, _p(out))
        return out

    def copy(self):
        r = Pcg32.__new__(Pcg32)
        r.si = self.si.copy()
        return r


# ------------------------------------------------------------------------------------------------
# sampling / bounds
# ------------------------------------------------------------------------------------------------
def check_pts_in_occ_voxel(xyz, bitfield, aabb23, n_grid):
    xyz = _f32(xyz)
    bf = np.ascontiguousarray(bitfield, dtype=np.uint8).reshape(-1)
    aabb = _f32(aabb23)
    out = np.zeros(xyz.shape[0], dtype=np.uint8)
    lib().orc_check_pts_in_occ_voxel(_p(xyz), _p(bf), _p(aabb), C.c_int(n_grid), _p(out), C.c_int64(xyz.shape[0]))
    return out.astype(bool)


def aabb_intersection(rays_o, rays_d, aabb_v23):
    """K2 semantics; aabb (V,2,3)."""
    o, d, bb = _f32(rays_o), _f32(rays_d), _f32(aabb_v23)
    R, V = o.shape[0], bb.shape[0]
    near = np.zeros((R, V), np.float32)
    far = np.zeros((R, V), np.float32)
    pts = np.zeros((R, V, 2, 3), np.float32)
    mask = np.zeros((R, V), np.uint8)
    lib().orc_aabb_intersection(_p(o), _p(d), _p(bb), _p(near), _p(far), _p(pts), _p(mask), C.c_int64(R), C.c_int64(V))
    return near, far, pts, mask.astype(bool)


def aabb_intersection_torch(rays_o, rays_d, aabb_v32, eps=1e-7):
    """torch-path semantics (geometry/ray.py:295-339); aabb (V,3,2)."""
    o, d, bb = _f32(rays_o), _f32(rays_d), _f32(aabb_v32)
    R, V = o.shape[0], bb.shape[0]
    near = np.zeros((R, V), np.float32)
    far = np.zeros((R, V), np.float32)
    pts = np.zeros((R, V, 2, 3), np.float32)
    mask = np.zeros((R, V), np.uint8)
    lib().orc_aabb_intersection_torch(_p(o), _p(d), _p(bb), C.c_float(eps), _p(near), _p(far), _p(pts), _p(mask),
                                      C.c_int64(R), C.c_int64(V))
    return near, far, pts, mask.astype(bool)


def sphere_intersection(rays_o, rays_d, radius, origin=(0.0, 0.0, 0.0)):
    """sphere_ray_intersection (geometry/ray.py:180-255); radius scalar or (n_r,)."""
    o, d = _f32(rays_o), _f32(rays_d)
    rad = _f32(np.atleast_1d(radius))
    org = _f32(np.asarray(origin, np.float32).reshape(3))
    R, K = o.shape[0], rad.shape[0]
    near = np.zeros((R, K), np.float32)
    far = np.zeros((R, K), np.float32)
    pts = np.zeros((R, K, 2, 3), np.float32)
    mask = np.zeros((R, K), np.uint8)
    lib().orc_sphere_intersection(_p(o), _p(d), _p(rad), _p(org), _p(near), _p(far), _p(pts), _p(mask), C.c_int64(R), C.c_int64(K))
    return near, far, pts, mask.astype(bool)


def get_rays(W, H, intrinsic, c2w, wh_order=True, index=None, center_pixel=False, normalize_rays_d=True, ndc=False, ndc_near=1.0):
    """-> rays_o (n,3), rays_d (n,3), rays_r (n,1) or None (index given).  index: (n,2) (i, j) pairs or None (full image)."""
    K, M = _f32(intrinsic).reshape(9), _f32(c2w).reshape(16)
    flat = None
    if index is not None:
        index = np.asarray(index, dtype=np.int64)
        flat = np.ascontiguousarray(index[:, 0] * H + index[:, 1])
    n = W * H if flat is None else flat.shape[0]
    o, d = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    r = np.zeros((n, 1), np.float32) if flat is None else None
    lib().orc_get_rays(C.c_int(W), C.c_int(H), _p(K), _p(M), C.c_int(int(wh_order)), _p(flat), C.c_int64(n), C.c_int(int(center_pixel)),
                       C.c_int(int(normalize_rays_d)), C.c_int(int(ndc)), C.c_float(ndc_near), _p(o), _p(d), _p(r))
    return o, d, r


def sparse_volume_sampling(rays_o, rays_d, near, far, n_pts, dt, aabb23, n_grid, bitfield, near_distance,
                           rng_state, rng_inc, with_trace=False):
    """K3.  Returns zvals (R,n_pts), mask (R,n_pts) bool, counts (R) [, voxel trace (R,n_pts) int32]."""
    o, d = _f32(rays_o), _f32(rays_d)
    nr, fr = _f32(near).reshape(-1), _f32(far).reshape(-1)
    bf = np.ascontiguousarray(bitfield, dtype=np.uint8).reshape(-1)
    aabb = _f32(aabb23)
    R = o.shape[0]
    zvals = np.zeros((R, n_pts), np.float32)
    mask = np.zeros((R, n_pts), np.uint8)
    counts = np.zeros(R, np.int32)
    trace = np.full((R, n_pts), -1, np.int32) if with_trace else None
    lib().orc_sparse_volume_sampling(_p(o), _p(d), _p(nr), _p(fr), C.c_int(n_pts), C.c_float(dt), _p(aabb),
                                     C.c_int(n_grid), _p(bf), C.c_float(near_distance), C.c_uint64(rng_state),
                                     C.c_uint64(rng_inc), _p(zvals), _p(mask), _p(trace), _p(counts), C.c_int64(R))
    if with_trace:
        return zvals, mask.astype(bool), counts, trace
    return zvals, mask.astype(bool), counts


def tensor_reduce_max(full, idx, n_group):
    full = _f32(full)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    out = np.zeros(n_group, np.float32)
    lib().orc_tensor_reduce_max(_p(full), _p(idx), C.c_int(n_group), _p(out), C.c_int64(full.shape[0]))
    return out


# ---- `_bitfield_func` family (K5-K10): Morton-order packed bitfield -------------------------------------------------
def morton3d(xyz):
    xyz = np.ascontiguousarray(xyz, dtype=np.uint32)
    out = np.zeros(xyz.shape[0], np.uint32)
    lib().orc_morton3d(_p(xyz), _p(out), C.c_int64(xyz.shape[0]))
    return out


def morton3d_invert(idx):
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    out = np.zeros((idx.shape[0], 3), np.uint32)
    lib().orc_morton3d_invert(_p(idx), _p(out), C.c_int64(idx.shape[0]))
    return out


def sparse_volume_sampling_bit(rays_o, rays_d, near, far, n_pts, dt, aabb23, n_grid, bitfield, near_distance,
                               rng_state, rng_inc, with_trace=False):
    """K5.  bitfield: packed (n_grid**3 / 8) uint8 in Morton order.  Returns like sparse_volume_sampling."""
    o, d = _f32(rays_o), _f32(rays_d)
    nr, fr = _f32(near).reshape(-1), _f32(far).reshape(-1)
    bf = np.ascontiguousarray(bitfield, dtype=np.uint8).reshape(-1)
    assert bf.shape[0] == n_grid ** 3 // 8
    aabb = _f32(aabb23)
    R = o.shape[0]
    zvals = np.zeros((R, n_pts), np.float32)
    mask = np.zeros((R, n_pts), np.uint8)
    counts = np.zeros(R, np.int32)
    trace = np.full((R, n_pts), -1, np.int32) if with_trace else None
    lib().orc_sparse_volume_sampling_bit(_p(o), _p(d), _p(nr), _p(fr), C.c_int(n_pts), C.c_float(dt), _p(aabb),
                                         C.c_int(n_grid), _p(bf), C.c_float(near_distance), C.c_uint64(rng_state),
                                         C.c_uint64(rng_inc), _p(zvals), _p(mask), _p(trace), _p(counts), C.c_int64(R))
    if with_trace:
        return zvals, mask.astype(bool), counts, trace
    return zvals, mask.astype(bool), counts


def generate_grid_samples(density_grid, n_elements, ema_step, n_grid, thresh, rng_state, rng_inc):
    """K6.  positions (n,3) in [0,1), indices (n) int32 (Morton cell)."""
    g = _f32(density_grid).reshape(-1)
    pos = np.zeros((n_elements, 3), np.float32)
    idx = np.zeros(n_elements, np.int32)
    lib().orc_generate_grid_samples(_p(g), C.c_int(ema_step), C.c_int(n_elements), C.c_int(n_grid), C.c_float(thresh),
                                    C.c_uint64(rng_state), C.c_uint64(rng_inc), _p(pos), _p(idx))
    return pos, idx


def splat_grid_samples(density, indices, grid_tmp):
    """K7, in place on grid_tmp (contiguous float32)."""
    assert grid_tmp.dtype == np.float32 and grid_tmp.flags.c_contiguous
    den = _f32(density).reshape(-1)
    idx = np.ascontiguousarray(indices, dtype=np.int32)
    lib().orc_splat_grid_samples(_p(den), _p(idx), C.c_int(idx.shape[0]), _p(grid_tmp))
    return grid_tmp


def ema_grid_samples_nerf(grid_tmp, grid, decay):
    """K8, in place on grid."""
    assert grid.dtype == np.float32 and grid.flags.c_contiguous
    tmp = _f32(grid_tmp).reshape(-1)
    lib().orc_ema_grid_samples_nerf(_p(tmp), C.c_int(grid.shape[0]), C.c_float(decay), _p(grid))
    return grid


def update_bitfield(grid, mean, opa_thres, n_grid):
    """K9 -> packed bitfield (n_grid**3/8) uint8."""
    g = _f32(grid).reshape(-1)
    bf = np.zeros(n_grid ** 3 // 8, np.uint8)
    lib().orc_update_bitfield(_p(g), C.c_float(mean), _p(bf), C.c_float(opa_thres), C.c_int(n_grid))
    return bf


def count_bitfield(bitfield, n_grid):
    """K10 (reference semantics: 8 per non-zero byte)."""
    bf = np.ascontiguousarray(bitfield, dtype=np.uint8).reshape(-1)
    cnt = np.zeros(1, np.float32)
    lib().orc_count_bitfield(_p(bf), _p(cnt), C.c_int(n_grid))
    return float(cnt[0])


# ---- `_multivol_func` family (K11, K12, cascaded K9) -------------------------------------------------------------------
def sparse_sampling_in_multivol_bitfield(rays_o, rays_d, near, far, n_pts, cone_angle, min_step, max_step, min_aabb23, aabb23,
                                         n_grid, n_cascade, bitfield, near_distance, inclusive, rng_state, rng_inc):
    """K11 -> zvals (R,n_pts), mask (R,n_pts) bool, counts (R)"""
    o, d = _f32(rays_o), _f32(rays_d)
    nr, fr = _f32(near).reshape(-1), _f32(far).reshape(-1)
    bf = np.ascontiguousarray(bitfield, dtype=np.uint8).reshape(-1)
    assert bf.shape[0] == n_grid ** 3 // 8 * (n_cascade if inclusive else n_cascade - 1)
    R = o.shape[0]
    zvals = np.zeros((R, n_pts), np.float32)
    mask = np.zeros((R, n_pts), np.uint8)
    counts = np.zeros(R, np.int32)
    lib().orc_sparse_sampling_in_multivol_bitfield(
        _p(o), _p(d), _p(nr), _p(fr), C.c_int(n_pts), C.c_float(cone_angle), C.c_float(min_step), C.c_float(max_step),
        _p(_f32(min_aabb23)), _p(_f32(aabb23)), C.c_int(n_grid), C.c_int(n_cascade), _p(bf), C.c_float(near_distance),
        C.c_int(int(bool(inclusive))), C.c_uint64(rng_state), C.c_uint64(rng_inc), _p(zvals), _p(mask), _p(counts), C.c_int64(R))
    return zvals, mask.astype(bool), counts


def generate_grid_samples_multivol(density_grid, n_elements, aabb23, ema_step, n_cascade, n_grid, thresh, inclusive, rng_state,
                                   rng_inc):
    """K12 -> positions (n,3) world space, indices (n) int32 = slot * n_grid^3 + Morton cell"""
    g = _f32(density_grid).reshape(-1)
    pos = np.zeros((n_elements, 3), np.float32)
    idx = np.zeros(n_elements, np.int32)
    lib().orc_generate_grid_samples_multivol(_p(g), C.c_int(ema_step), C.c_int(n_elements), _p(_f32(aabb23)), C.c_int(n_cascade),
                                             C.c_int(n_grid), C.c_float(thresh), C.c_int(int(bool(inclusive))),
                                             C.c_uint64(rng_state), C.c_uint64(rng_inc), _p(pos), _p(idx))
    return pos, idx


def update_bitfield_multivol(grid, mean, opa_thres, n_grid, n_cascade, inclusive):
    g = _f32(grid).reshape(-1)
    bf = np.zeros(n_grid ** 3 // 8 * (n_cascade if inclusive else n_cascade - 1), np.uint8)
    lib().orc_update_bitfield_multivol(_p(g), C.c_float(mean), _p(bf), C.c_float(opa_thres), C.c_int(n_grid), C.c_int(n_cascade),
                                       C.c_int(int(bool(inclusive))))
    return bf


def update_opafield(opafield, flat_idx, opacity, ema=None):
    """In place on a contiguous float32 array."""
    assert opafield.dtype == np.float32 and opafield.flags.c_contiguous
    idx = np.ascontiguousarray(flat_idx, dtype=np.int64)
    op = _f32(opacity)
    lib().orc_update_opafield(_p(opafield), _p(idx), _p(op), C.c_int64(idx.shape[0]),
                              C.c_float(-1.0 if ema is None else ema))
    return opafield


def update_bitfield_by_opafield(opafield, threshold):
    opa = _f32(opafield)
    bf = np.zeros(opa.size, np.uint8)
    thres = lib().orc_update_bitfield_by_opafield(_p(opa), _p(bf), C.c_int64(opa.size), C.c_float(threshold))
    return bf.astype(bool).reshape(opa.shape), float(thres)


def voxel_grid_info(pts, min_xyz, max_xyz, n_grid):
    pts = _f32(pts)
    n = pts.shape[0]
    mn, mx = _f32(min_xyz), _f32(max_xyz)
    vidx = np.zeros((n, 3), np.int64)
    valid = np.zeros(n, np.uint8)
    cidx = np.zeros((n, 8, 3), np.int64)
    w = np.zeros((n, 8), np.float32)
    lib().orc_voxel_grid_info(_p(pts), C.c_int64(n), _p(mn), _p(mx), C.c_int(n_grid), _p(vidx), _p(valid), _p(cidx), _p(w))
    return vidx, valid.astype(bool), cidx, w


# ------------------------------------------------------------------------------------------------
# encoders
# ------------------------------------------------------------------------------------------------
def hashgrid_levels(n_levels=16, hashmap_size=19, base_res=16, max_res=2048):
    """Level table of HashGridEmbedder.init_embeddings (hashgrid_encoder.py:126-158).

    per_level_scale is a float32 torch scalar in the reference (exp(log(max/base)/(L-1)) evaluated in
    fp32, :84) and math.log2 of it is taken in double: reproduced with numpy float32.
    """
    import math
    # evaluated in double and rounded once to fp32 (numpy's and torch's fp32 exp differ in the last ulp, which
    # matters when 2^(i*log2 s)*base lands within 1e-6 of an integer, e.g. base 4 / max 64 / L 6)
    pls = np.float32(math.exp(math.log(max_res / base_res) / (float(n_levels) - 1)))
    res, offs, total = [], [], 0
    T = 2 ** hashmap_size
    for i in range(n_levels):
        offs.append(total)
        r = math.ceil(2 ** (i * math.log2(float(pls))) * base_res - 1.0)
        res.append(r)
        total += min(T, (r + 1) ** 3)
    offs.append(total)
    return np.array(res, np.int32), np.array(offs, np.int64)


def hashgrid_fwd(xyz, table, resolutions, offsets, min_xyz, max_xyz, with_idx=False):
    xyz, table = _f32(xyz), _f32(table)
    S, L, F = xyz.shape[0], len(resolutions), table.shape[1]
    res = np.ascontiguousarray(resolutions, np.int32)
    off = np.ascontiguousarray(offsets, np.int64)
    mn, mx = _f32(min_xyz), _f32(max_xyz)
    out = np.zeros((S, L * F), np.float32)
    idx = np.zeros((S, L, 8), np.int64) if with_idx else None
    lib().orc_hashgrid_fwd(_p(xyz), C.c_int64(S), _p(table), C.c_int(L), C.c_int(F), _p(res), _p(off), _p(mn), _p(mx),
                           _p(out), _p(idx))
    return (out, idx) if with_idx else out


def hashgrid_bwd(xyz, table, dout, resolutions, offsets, min_xyz, max_xyz, want_dxyz=False):
    xyz, table, dout = _f32(xyz), _f32(table), _f32(dout)
    S, L, F = xyz.shape[0], len(resolutions), table.shape[1]
    res = np.ascontiguousarray(resolutions, np.int32)
    off = np.ascontiguousarray(offsets, np.int64)
    mn, mx = _f32(min_xyz), _f32(max_xyz)
    dtable = np.zeros_like(table)
    dxyz = np.zeros((S, 3), np.float32) if want_dxyz else None
    lib().orc_hashgrid_bwd(_p(xyz), C.c_int64(S), _p(table), _p(dout), C.c_int(L), C.c_int(F), _p(res), _p(off), _p(mn),
                           _p(mx), _p(dtable), _p(dxyz))
    return (dtable, dxyz) if want_dxyz else dtable


def hashgrid_bwd_bwd(xyz, gdx, table, dout, resolutions, offsets, min_xyz, max_xyz):
    """second-order pieces of the encoding's input gradient -> ddout (S, L*F), dtable (n_total, F), d2xyz (S, 3)"""
    xyz, gdx, table, dout = _f32(xyz), _f32(gdx), _f32(table), _f32(dout)
    res = np.ascontiguousarray(resolutions, dtype=np.int32)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    S, L, F = xyz.shape[0], len(res), table.shape[1]
    ddout = np.zeros((S, L * F), np.float32)
    dtable = np.zeros_like(table)
    d2x = np.zeros((S, 3), np.float32)
    lib().orc_hashgrid_bwd_bwd(_p(xyz), C.c_int64(S), _p(gdx), _p(table), _p(dout), C.c_int(L), C.c_int(F), _p(res), _p(off),
                               _p(_f32(min_xyz)), _p(_f32(max_xyz)), _p(ddout), _p(dtable), _p(d2x))
    return ddout, dtable, d2x


def freq_fwd(x, n_freqs, include_input=True):
    x = _f32(x)
    S, D = x.shape
    out = np.zeros((S, D * (1 if include_input else 0) + 2 * D * n_freqs), np.float32)
    lib().orc_freq_fwd(_p(x), C.c_int64(S), C.c_int(D), C.c_int(n_freqs), C.c_int(int(include_input)), _p(out))
    return out


def freq_bwd(x, dout, n_freqs, include_input=True):
    x, dout = _f32(x), _f32(dout)
    S, D = x.shape
    dx = np.zeros_like(x)
    lib().orc_freq_bwd(_p(x), _p(dout), C.c_int64(S), C.c_int(D), C.c_int(n_freqs), C.c_int(int(include_input)), _p(dx))
    return dx


def sh_fwd(dirs, degree, include_input=False):
    dirs = _f32(dirs)
    S = dirs.shape[0]
    out = np.zeros((S, degree * degree + (3 if include_input else 0)), np.float32)
    lib().orc_sh_fwd(_p(dirs), C.c_int64(S), C.c_int(degree), C.c_int(int(include_input)), _p(out))
    return out


# ------------------------------------------------------------------------------------------------
# dense layers
# ------------------------------------------------------------------------------------------------
def act_fwd(x, act, beta=1.0):
    x = _f32(x)
    y = np.zeros_like(x)
    lib().orc_act_fwd(_p(x), _p(y), C.c_int64(x.size), C.c_int(ACT[act]), C.c_float(beta))
    return y


def act_bwd(x, y, dy, act, beta=1.0):
    x, y, dy = _f32(x), _f32(y), _f32(dy)
    dx = np.zeros_like(x)
    lib().orc_act_bwd(_p(x), _p(y), _p(dy), _p(dx), C.c_int64(x.size), C.c_int(ACT[act]), C.c_float(beta))
    return dx


def linear_fwd(x, W, b=None, act=None, beta=1.0, want_pre=False):
    x, W = _f32(x), _f32(W)
    b = None if b is None else _f32(b)
    S, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K
    y = np.zeros((S, N), np.float32)
    pre = np.zeros((S, N), np.float32) if want_pre else None
    lib().orc_linear_fwd(_p(x), _p(W), _p(b), C.c_int64(S), C.c_int(K), C.c_int(N), C.c_int(ACT[act]), C.c_float(beta),
                         _p(y), _p(pre))
    return (y, pre) if want_pre else y


def linear_bwd(x, W, pre, y, dy, act=None, beta=1.0, has_bias=False):
    x, W, pre, y, dy = _f32(x), _f32(W), _f32(pre), _f32(y), _f32(dy)
    S, K = x.shape
    N = W.shape[0]
    dx = np.zeros((S, K), np.float32)
    dW = np.zeros((N, K), np.float32)
    db = np.zeros(N, np.float32) if has_bias else None
    lib().orc_linear_bwd(_p(x), _p(W), _p(pre), _p(y), _p(dy), C.c_int64(S), C.c_int(K), C.c_int(N), C.c_int(ACT[act]),
                         C.c_float(beta), _p(dx), _p(dW), _p(db))
    return dx, dW, db


# ------------------------------------------------------------------------------------------------
# compositing / resampling
# ------------------------------------------------------------------------------------------------
def sdf_to_alpha_fwd(mid_sdf, zvals, mid_slope, s, clip=True):
    """NeuS sdf_to_alpha (models/neus_model.py:242-265): alpha (R, P-1)."""
    sd, z, sl = _f32(mid_sdf), _f32(zvals), _f32(mid_slope)
    R, P = z.shape
    alpha = np.zeros((R, P - 1), np.float32)
    lib().orc_sdf_to_alpha_fwd(_p(sd), _p(z), _p(sl), C.c_float(s), C.c_int(int(clip)), _p(alpha), C.c_int64(R), C.c_int(P))
    return alpha


def sdf_to_alpha_bwd(mid_sdf, zvals, mid_slope, s, d_alpha, clip=True):
    """-> d mid_sdf, d mid_slope (R, P-1), d s (float)"""
    sd, z, sl, da = _f32(mid_sdf), _f32(zvals), _f32(mid_slope), _f32(d_alpha)
    R, P = z.shape
    d_sdf, d_slope = np.zeros_like(sd), np.zeros_like(sl)
    d_s = C.c_double(0.0)
    lib().orc_sdf_to_alpha_bwd(_p(sd), _p(z), _p(sl), C.c_float(s), C.c_int(int(clip)), _p(da), _p(d_sdf), _p(d_slope),
                               C.byref(d_s), C.c_int64(R), C.c_int(P))
    return d_sdf, d_slope, float(d_s.value)


def ray_marching_fwd(sigma, radiance, zvals, add_inf_z=False, white_bkg=False, alpha=None, bkg_color=None, noise=None):
    """Returns dict(rgb, depth, mask, alpha, trans_shift, weights) like ray_helper.ray_marching."""
    z = _f32(zvals)
    R, P = z.shape
    sg = None if sigma is None else _f32(sigma)
    al = None if alpha is None else _f32(alpha)
    rad = None if radiance is None else _f32(radiance)
    ns = None if noise is None else _f32(noise)
    bk = None if bkg_color is None else _f32(bkg_color).reshape(-1, 3)
    Pe = P if (add_inf_z or al is not None) else P - 1
    rgb = np.zeros((R, 3), np.float32) if rad is not None else None
    depth, mask = np.zeros(R, np.float32), np.zeros(R, np.float32)
    a_o, t_o, w_o = (np.zeros((R, Pe), np.float32) for _ in range(3))
    rc = lib().orc_ray_marching_fwd(_p(sg), _p(al), _p(rad), _p(z), _p(ns), _p(bk),
                                    C.c_int64(0 if bk is None else bk.shape[0]), C.c_int64(R), C.c_int(P),
                                    C.c_int(int(add_inf_z)), C.c_int(int(white_bkg)), _p(rgb), _p(depth), _p(mask),
                                    _p(a_o), _p(t_o), _p(w_o))
    if rc != 0:
        raise AssertionError('zvals is not all increase....')
    return {'rgb': rgb, 'depth': depth, 'mask': mask, 'alpha': a_o, 'trans_shift': t_o, 'weights': w_o}


def ray_marching_bwd(sigma, radiance, zvals, d_rgb, d_depth=None, d_mask=None, add_inf_z=False, white_bkg=False,
                     alpha=None, bkg_color=None, noise=None):
    z = _f32(zvals)
    R, P = z.shape
    sg = None if sigma is None else _f32(sigma)
    al = None if alpha is None else _f32(alpha)
    rad = None if radiance is None else _f32(radiance)
    ns = None if noise is None else _f32(noise)
    bk = None if bkg_color is None else _f32(bkg_color).reshape(-1, 3)
    g_rgb = None if d_rgb is None else _f32(d_rgb)
    g_d = None if d_depth is None else _f32(d_depth)
    g_m = None if d_mask is None else _f32(d_mask)
    d_geo = np.zeros((R, P), np.float32)
    d_rad = np.zeros((R, P, 3), np.float32) if rad is not None else None
    lib().orc_ray_marching_bwd(_p(sg), _p(al), _p(rad), _p(z), _p(ns), _p(bk),
                               C.c_int64(0 if bk is None else bk.shape[0]), C.c_int64(R), C.c_int(P),
                               C.c_int(int(add_inf_z)), C.c_int(int(white_bkg)), _p(g_rgb), _p(g_d), _p(g_m), _p(d_geo),
                               _p(d_rad))
    return d_geo, d_rad


def sample_cdf(bins, cdf, u, eps=1e-5, sort=True):
    bins, cdf, u = _f32(bins), _f32(cdf), _f32(u)
    R, n_pts = bins.shape
    n_sample = u.shape[1]
    samples = np.zeros((R, n_sample), np.float32)
    inds = np.zeros((R, n_sample), np.int64)
    lib().orc_sample_cdf(_p(bins), _p(cdf), _p(u), C.c_int64(R), C.c_int(n_pts), C.c_int(n_sample), C.c_float(eps),
                         C.c_int(int(sort)), _p(samples), _p(inds))
    return samples, inds


def weights_to_cdf(weights, eps=1e-5):
    w = _f32(weights)
    R, n_w = w.shape
    cdf = np.zeros((R, n_w + 1), np.float32)
    lib().orc_weights_to_cdf(_p(w), C.c_int64(R), C.c_int(n_w), C.c_float(eps), _p(cdf))
    return cdf


def sample_pdf(bins, weights, n_sample, u=None, eps=1e-5):
    """det=True when u is None (u = linspace(0,1,n_sample))."""
    cdf = weights_to_cdf(weights, eps)
    if u is None:
        u = np.broadcast_to(linspace01(n_sample)[None], (cdf.shape[0], n_sample))
    return sample_cdf(bins, cdf, u, eps)[0]


def linspace01(n):
    """torch.linspace(0,1,n) in fp32 (symmetric evaluation of the torch kernel)."""
    if n == 1:
        return np.zeros(1, np.float32)
    step = np.float32(1.0) / np.float32(n - 1)
    idx = np.arange(n)
    lo = (np.float32(0.0) + step * idx.astype(np.float32)).astype(np.float32)
    hi = (np.float32(1.0) - step * (n - 1 - idx).astype(np.float32)).astype(np.float32)
    return np.where(idx < n // 2, lo, hi).astype(np.float32)


def zvals_from_near_far(near, far, n_pts, inclusive=True, inverse_linear=False):
    near, far = _f32(near).reshape(-1), _f32(far).reshape(-1)
    R = near.shape[0]
    z = np.zeros((R, n_pts), np.float32)
    lib().orc_zvals_from_near_far(_p(near), _p(far), C.c_int64(R), C.c_int(n_pts), C.c_int(int(inclusive)),
                                  C.c_int(int(inverse_linear)), _p(z))
    return z
