"""Thin torch-tensor front-end of the C ABI (include/arcnerf_hip.h).  No autograd here (see ops/autograd.py).

torch is used for device memory and the current HIP stream only; every numerical operation is a hand-written
gfx950 kernel in libarcnerf_hip.so.  All functions require CUDA(HIP) tensors and raise RuntimeError otherwise —
there is deliberately no CPU fallback.
"""
import contextlib
import os
import ctypes as C

import torch

from .. import _native as N


def _req(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('arcnerf_amd ops run on the GPU only (got a {} tensor); there is no CPU fallback'.format(t.device))


def _f32(t):
    return None if t is None else t.contiguous().float()


def _nptr(n_dev):
    """optional device-side element count (int32 tensor with one element)"""
    if n_dev is None:
        return None
    assert n_dev.dtype == torch.int32 and n_dev.is_cuda
    return n_dev.data_ptr()


_POISON = os.environ.get('ARCN_POISON_OUTPUTS') == '1'


def _fresh(shape, dtype, device):
    """an output buffer its kernel writes EVERY element of: torch.empty - or, with ARCN_POISON_OUTPUTS=1 (the test suite's check that no kernel
    leaves an element of such a buffer unwritten), filled with NaN / 0x7f bytes first"""
    t = torch.empty(shape, dtype=dtype, device=device)
    if _POISON:
        if t.dtype.is_floating_point:
            t.fill_(float('nan'))
        elif t.dtype == torch.bool:
            t.fill_(True)
        else:
            t.fill_(0x7f7f7f7f if t.dtype in (torch.int32, torch.int64) else 0x7f)
    return t


# ------------------------------------------------------------------------------------------------
# _volume_func
# ------------------------------------------------------------------------------------------------
def check_pts_in_occ_voxel(xyz, bitfield, aabb23, n_grid):
    _req(xyz, bitfield, aabb23)
    xyz = _f32(xyz)
    bf = bitfield.contiguous().view(torch.uint8)
    aabb = _f32(aabb23)
    out = torch.zeros(xyz.shape[0], dtype=torch.bool, device=xyz.device)
    N.check(N.lib().arcn_check_pts_in_occ_voxel(N.ptr(xyz), N.ptr(bf), N.ptr(aabb), int(n_grid), out.data_ptr(),
                                               xyz.shape[0], N.stream()), 'check_pts_in_occ_voxel')
    return out


def aabb_intersection(rays_o, rays_d, aabb_v23, want_pts=True):
    """K2 semantics, aabb (V,2,3)."""
    _req(rays_o, rays_d, aabb_v23)
    o, d, bb = _f32(rays_o), _f32(rays_d), _f32(aabb_v23)
    R, V = o.shape[0], bb.shape[0]
    near = _fresh((R, V), torch.float32, o.device)       # (the kernel writes every (ray, volume) entry of the four outputs)
    far = _fresh((R, V), torch.float32, o.device)
    pts = _fresh((R, V, 2, 3), torch.float32, o.device) if want_pts else None
    mask = _fresh((R, V), torch.bool, o.device)
    N.check(N.lib().arcn_aabb_intersection(N.ptr(o), N.ptr(d), N.ptr(bb), N.ptr(near), N.ptr(far), N.ptr(pts),
                                          mask.data_ptr(), R, V, N.stream()), 'aabb_intersection')
    return near, far, pts, mask


def aabb_intersection_torch(rays_o, rays_d, aabb_v32, eps=1e-7, want_pts=True):
    """torch-path semantics of geometry/ray.py:295-339, aabb (V,3,2)."""
    _req(rays_o, rays_d, aabb_v32)
    o, d, bb = _f32(rays_o), _f32(rays_d), _f32(aabb_v32)
    R, V = o.shape[0], bb.shape[0]
    near = _fresh((R, V), torch.float32, o.device)       # (the kernel writes every (ray, volume) entry of the four outputs)
    far = _fresh((R, V), torch.float32, o.device)
    pts = _fresh((R, V, 2, 3), torch.float32, o.device) if want_pts else None
    mask = _fresh((R, V), torch.bool, o.device)
    N.check(N.lib().arcn_aabb_intersection_torch(N.ptr(o), N.ptr(d), N.ptr(bb), float(eps), N.ptr(near), N.ptr(far),
                                                N.ptr(pts), mask.data_ptr(), R, V, N.stream()), 'aabb_intersection_torch')
    return near, far, pts, mask


_SCALARS = {}


def scalar_tensor(value, device):
    """(1,) float32 device tensor holding a Python number, made once per (value, device): torch.tensor([v], device=cuda) is a pageable
    host-to-device copy - the host waits for it - and the samplers / sdf_to_alpha ask with the same few values on every call."""
    key = (float(value), str(device))
    t = _SCALARS.get(key)
    if t is None:
        if len(_SCALARS) > 256:
            _SCALARS.clear()
        t = _SCALARS[key] = torch.tensor([float(value)], dtype=torch.float32, device=device)
    return t


def sphere_intersection(rays_o, rays_d, radius, origin=(0.0, 0.0, 0.0), want_pts=True):
    """sphere_ray_intersection of geometry/ray.py:180-255; radius float or (N_r,) tensor."""
    _req(rays_o, rays_d)
    o, d = _f32(rays_o), _f32(rays_d)
    if not torch.is_tensor(radius):
        radius = scalar_tensor(radius, o.device)
    rad = _f32(radius.to(o.device)).view(-1)
    R, K = o.shape[0], rad.shape[0]
    near = torch.zeros((R, K), dtype=torch.float32, device=o.device)
    far = torch.zeros((R, K), dtype=torch.float32, device=o.device)
    pts = torch.zeros((R, K, 2, 3), dtype=torch.float32, device=o.device) if want_pts else None
    mask = torch.zeros((R, K), dtype=torch.bool, device=o.device)
    org = (C.c_float * 3)(*[float(v) for v in origin])
    N.check(N.lib().arcn_sphere_intersection(N.ptr(o), N.ptr(d), N.ptr(rad), C.addressof(org), N.ptr(near), N.ptr(far), N.ptr(pts),
                                            mask.data_ptr(), R, K, N.stream()), 'sphere_intersection')
    return near, far, pts, mask


def get_rays(W, H, intrinsic, c2w, wh_order=True, flat_index=None, center_pixel=False, normalize_rays_d=True, ndc=False,
             ndc_near=1.0):
    """-> rays_o (n,3), rays_d (n,3), rays_r (n,1) or None (when flat_index, (n,) int64 column-major pixel ids, is given)"""
    _req(intrinsic, c2w, flat_index)
    K, M = _f32(intrinsic), _f32(c2w)
    if tuple(K.shape) != (3, 3) or tuple(M.shape) != (4, 4):
        raise RuntimeError('intrinsic must be (3,3) and c2w (4,4)')
    n = int(W) * int(H) if flat_index is None else flat_index.shape[0]
    idx = None if flat_index is None else flat_index.contiguous().long()
    o = torch.empty((n, 3), dtype=torch.float32, device=K.device)
    d = torch.empty((n, 3), dtype=torch.float32, device=K.device)
    r = torch.empty((n, 1), dtype=torch.float32, device=K.device) if idx is None else None
    N.check(N.lib().arcn_get_rays(int(W), int(H), N.ptr(K), N.ptr(M), int(bool(wh_order)), N.ptr(idx), n, int(bool(center_pixel)),
                                 int(bool(normalize_rays_d)), int(bool(ndc)), float(ndc_near), N.ptr(o), N.ptr(d), N.ptr(r),
                                 N.stream()), 'get_rays')
    return o, d, r


def fetch_train_batch(ids, n_img, H, W, window=None, rgba=None, img=None, mask=None, intrinsic=None, c2w=None, center_pixel=True,
                      normalize_rays_d=True, bkg_rand=None, bkg_const=None, want_rays_r=True, want_src=False, bad_ids=None):
    """One launch for a training batch (arcn_fetch_train_batch): ids (n,) int64 rows of the (cropped) dataset -> dict with `rays_o`,
    `rays_d` (n,3), `rays_r` (n,1) when cameras are given; `img` (n,3) (blended with the background colour when the data has a mask and a
    colour is given), `mask` (n,), `bkg_color` (n,3) when colours are given; `src` (n,) int64 rows of the uncropped tensors on request.
    window = (y0, x0, Hc, Wc) of the centre crop (default: the whole image).  bad_ids: int32 device counter of ids outside the dataset."""
    _req(ids, rgba, img, mask, intrinsic, c2w, bkg_rand, bad_ids)
    ids = ids.contiguous()
    if ids.dtype != torch.int64:
        raise RuntimeError('fetch_train_batch: ids must be int64')
    n, dev = ids.shape[0], ids.device
    y0, x0, Hc, Wc = (0, 0, int(H), int(W)) if window is None else [int(v) for v in window]
    if rgba is not None and (rgba.dtype != torch.uint8 or rgba.numel() != n_img * H * W * 4):
        raise RuntimeError('fetch_train_batch: rgba must be (n_img, H, W, 4) bytes')
    if img is not None and img.numel() != n_img * H * W * 3:
        raise RuntimeError('fetch_train_batch: img must be (n_img, H, W, 3)')
    if mask is not None and mask.numel() != n_img * H * W:
        raise RuntimeError('fetch_train_batch: mask must be (n_img, H, W)')
    out = {}
    has_cam = intrinsic is not None and c2w is not None
    if has_cam:
        Kc, Mc = _f32(intrinsic), _f32(c2w)
        if Kc.numel() != n_img * 9 or Mc.numel() != n_img * 16:
            raise RuntimeError('fetch_train_batch: intrinsic must be (n_img, 3, 3) and c2w (n_img, 4, 4)')
        out['rays_o'] = torch.empty((n, 3), dtype=torch.float32, device=dev)
        out['rays_d'] = torch.empty((n, 3), dtype=torch.float32, device=dev)
        if want_rays_r:
            out['rays_r'] = torch.empty((n, 1), dtype=torch.float32, device=dev)
    else:
        Kc = Mc = None
    has_col = rgba is not None or img is not None
    has_mask = rgba is not None or mask is not None
    imgc, maskc = _f32(img), _f32(mask)
    rgbac = rgba.contiguous() if rgba is not None else None
    blend = has_col and has_mask and (bkg_rand is not None or bkg_const is not None)
    if has_col:
        out['img'] = torch.empty((n, 3), dtype=torch.float32, device=dev)
        if has_mask:
            out['mask'] = torch.empty((n,), dtype=torch.float32, device=dev)
        if blend:
            out['bkg_color'] = torch.empty((n, 3), dtype=torch.float32, device=dev)
    if want_src:
        out['src'] = torch.empty((n,), dtype=torch.int64, device=dev)
    br = _f32(bkg_rand) if blend and bkg_rand is not None else None
    if br is not None and br.numel() != 3 * n:
        raise RuntimeError('fetch_train_batch: bkg_rand must be (n, 3)')
    bc = (C.c_float * 3)(*[float(v) for v in bkg_const]) if blend and br is None else None
    N.check(N.lib().arcn_fetch_train_batch(N.ptr(rgbac), N.ptr(imgc), N.ptr(maskc), N.ptr(Kc), N.ptr(Mc), int(n_img), int(H), int(W), y0, x0, Hc, Wc,
                                          N.ptr(ids), n, int(bool(center_pixel)), int(bool(normalize_rays_d)), N.ptr(br),
                                          C.addressof(bc) if bc is not None else None, N.ptr(out.get('rays_o')), N.ptr(out.get('rays_d')),
                                          N.ptr(out.get('rays_r')), N.ptr(out.get('img')), N.ptr(out.get('mask')), N.ptr(out.get('bkg_color')),
                                          N.ptr(out.get('src')), N.ptr(bad_ids), N.stream()), 'fetch_train_batch')
    return out


def sparse_volume_sampling(rays_o, rays_d, near, far, n_pts, dt, aabb23, n_grid, bitfield, near_distance, rng_state,
                           rng_inc, want_counts=False, dense=True):
    _req(rays_o, rays_d, near, far, aabb23, bitfield)
    o, d = _f32(rays_o), _f32(rays_d)
    nr, fr = _f32(near).view(-1), _f32(far).view(-1)
    aabb = _f32(aabb23)
    bf = bitfield.contiguous().view(torch.uint8)
    R = o.shape[0]
    alloc = torch.zeros if (dense or not want_counts) else torch.empty
    zvals = alloc((R, n_pts), dtype=torch.float32, device=o.device)
    mask = alloc((R, n_pts), dtype=torch.bool, device=o.device)
    counts = torch.zeros(R, dtype=torch.int32, device=o.device) if want_counts else None
    N.check(N.lib().arcn_sparse_volume_sampling(N.ptr(o), N.ptr(d), N.ptr(nr), N.ptr(fr), int(n_pts), float(dt),
                                               N.ptr(aabb), int(n_grid), N.ptr(bf), float(near_distance),
                                               int(rng_state), int(rng_inc), N.ptr(zvals), mask.data_ptr(),
                                               N.ptr(counts), R, N.stream()), 'sparse_volume_sampling')
    return (zvals, mask, counts) if want_counts else (zvals, mask)


def tensor_reduce_max(full, idx, n_group):
    _req(full, idx)
    full = _f32(full)
    idx = idx.contiguous().long()
    out = torch.zeros(n_group, dtype=torch.float32, device=full.device)
    N.check(N.lib().arcn_tensor_reduce_max(N.ptr(full), N.ptr(idx), int(n_group), N.ptr(out), full.shape[0], N.stream()),
            'tensor_reduce_max')
    return out


# ------------------------------------------------------------------------------------------------
# `_bitfield_func` family (K5-K10): Morton-order density grid + packed bitfield
# ------------------------------------------------------------------------------------------------
def sparse_volume_sampling_bit(rays_o, rays_d, near, far, n_pts, dt, aabb23, n_grid, bitfield, near_distance, rng_state,
                               rng_inc, want_counts=False, dense=True):
    _req(rays_o, rays_d, near, far, aabb23, bitfield)
    o, d = _f32(rays_o), _f32(rays_d)
    nr, fr = _f32(near).view(-1), _f32(far).view(-1)
    aabb = _f32(aabb23)
    if bitfield.dtype != torch.uint8 or bitfield.numel() != int(n_grid) ** 3 // 8:
        raise RuntimeError('bitfield should be uint8 in (n_grid**3/8,)')
    bf = bitfield.contiguous()
    R = o.shape[0]
    alloc = torch.zeros if (dense or not want_counts) else torch.empty
    zvals = alloc((R, n_pts), dtype=torch.float32, device=o.device)
    mask = alloc((R, n_pts), dtype=torch.bool, device=o.device)
    counts = torch.zeros(R, dtype=torch.int32, device=o.device) if want_counts else None
    N.check(N.lib().arcn_sparse_volume_sampling_bit(N.ptr(o), N.ptr(d), N.ptr(nr), N.ptr(fr), int(n_pts), float(dt),
                                                   N.ptr(aabb), int(n_grid), N.ptr(bf), float(near_distance),
                                                   int(rng_state), int(rng_inc), N.ptr(zvals), mask.data_ptr(),
                                                   N.ptr(counts), R, N.stream()), 'sparse_volume_sampling_bit')
    return (zvals, mask, counts) if want_counts else (zvals, mask)


def generate_grid_samples(density_grid, n_elements, ema_step, n_grid, thresh, rng_state, rng_inc):
    _req(density_grid)
    g = _f32(density_grid)
    pos = torch.empty((n_elements, 3), dtype=torch.float32, device=g.device)
    idx = torch.empty((n_elements,), dtype=torch.int32, device=g.device)
    N.check(N.lib().arcn_generate_grid_samples(N.ptr(g), int(ema_step), int(n_elements), int(n_grid), float(thresh),
                                              int(rng_state), int(rng_inc), N.ptr(pos), N.ptr(idx), N.stream()),
            'generate_grid_samples')
    return pos, idx


def splat_grid_samples(density, indices, n_samples, density_grid_tmp):
    """In place on density_grid_tmp (float32, contiguous)."""
    _req(density, indices, density_grid_tmp)
    if density_grid_tmp.dtype != torch.float32 or not density_grid_tmp.is_contiguous():
        raise RuntimeError('density_grid_tmp must be a contiguous float32 tensor')
    if indices.dtype != torch.int32:
        raise RuntimeError('density_grid_indices must be int32')
    den = _f32(density).view(-1)
    if n_samples > den.shape[0] or n_samples > indices.shape[0]:
        raise RuntimeError('n_samples exceeds the inputs')
    N.check(N.lib().arcn_splat_grid_samples(N.ptr(den), N.ptr(indices.contiguous()), int(n_samples), N.ptr(density_grid_tmp),
                                           N.stream()), 'splat_grid_samples')
    return density_grid_tmp


def ema_grid_samples_nerf(density_grid_tmp, density_grid, n_elements, decay):
    """In place on density_grid."""
    _req(density_grid_tmp, density_grid)
    for t in (density_grid_tmp, density_grid):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() < n_elements:
            raise RuntimeError('grids must be contiguous float32 tensors of at least n_elements')
    N.check(N.lib().arcn_ema_grid_samples_nerf(N.ptr(density_grid_tmp), int(n_elements), float(decay), N.ptr(density_grid),
                                              N.stream()), 'ema_grid_samples_nerf')
    return density_grid


def update_bitfield(density_grid, density_grid_mean, density_grid_bitfield, thres, n_grid):
    """In place on density_grid_bitfield.  density_grid_mean: python float, or a 1-element device tensor (read on the device:
    no host round trip)."""
    _req(density_grid, density_grid_bitfield)
    if density_grid.dtype != torch.float32 or not density_grid.is_contiguous() or density_grid.numel() < int(n_grid) ** 3:
        raise RuntimeError('density_grid must be a contiguous float32 tensor of n_grid**3')
    if density_grid_bitfield.dtype != torch.uint8 or density_grid_bitfield.numel() < int(n_grid) ** 3 // 8:
        raise RuntimeError('bitfield should be uint8 in (n_grid**3/8,)')
    mean_dev, mean_host = None, 0.0
    if torch.is_tensor(density_grid_mean) and density_grid_mean.is_cuda:
        mean_dev = _f32(density_grid_mean).view(-1)
    else:
        mean_host = float(density_grid_mean)
    N.check(N.lib().arcn_update_bitfield(N.ptr(density_grid), mean_host, N.ptr(mean_dev), N.ptr(density_grid_bitfield),
                                        float(thres), int(n_grid), N.stream()), 'update_bitfield')
    return density_grid_bitfield


def count_bitfield(density_grid_bitfield, n_grid, counter=None):
    """Returns the device counter (1,) float32 (reference semantics: 8 per non-zero byte)."""
    _req(density_grid_bitfield)
    if counter is None:
        counter = torch.zeros((1,), dtype=torch.float32, device=density_grid_bitfield.device)
    N.check(N.lib().arcn_count_bitfield(N.ptr(density_grid_bitfield.contiguous()), N.ptr(counter), int(n_grid), N.stream()),
            'count_bitfield')
    return counter


# ------------------------------------------------------------------------------------------------
# `_multivol_func` family (K11, K12, cascaded K9)
# ------------------------------------------------------------------------------------------------
def _multivol_levels(n_cascade, inclusive):
    return int(n_cascade) if inclusive else int(n_cascade) - 1


def sparse_sampling_in_multivol_bitfield(rays_o, rays_d, near, far, n_pts, cone_angle, min_step, max_step, min_aabb23, aabb23,
                                         n_grid, n_cascade, bitfield, near_distance, inclusive, rng_state, rng_inc,
                                         want_counts=False, dense=True):
    """dense=False (with want_counts): only the first counts[r] entries of a row of zvals are meaningful - the form the packed paths
    compact (pack_dense_samples); the (rays, n_pts) outputs are then not zero-filled first (21 MB of fills per 4096-ray batch)"""
    _req(rays_o, rays_d, near, far, min_aabb23, aabb23, bitfield)
    o, d = _f32(rays_o), _f32(rays_d)
    nr, fr = _f32(near).view(-1), _f32(far).view(-1)
    if bitfield.dtype != torch.uint8 or bitfield.numel() != int(n_grid) ** 3 // 8 * _multivol_levels(n_cascade, inclusive):
        raise RuntimeError('bitfield should be uint8 in (n_grid**3/8 * levels,)')
    R = o.shape[0]
    alloc = torch.zeros if (dense or not want_counts) else torch.empty
    zvals = alloc((R, n_pts), dtype=torch.float32, device=o.device)
    mask = alloc((R, n_pts), dtype=torch.bool, device=o.device)
    counts = _fresh((R,), torch.int32, o.device) if want_counts else None      # (every ray's wave writes its count)
    N.check(N.lib().arcn_sparse_sampling_in_multivol_bitfield(
        N.ptr(o), N.ptr(d), N.ptr(nr), N.ptr(fr), int(n_pts), float(cone_angle), float(min_step), float(max_step),
        N.ptr(_f32(min_aabb23)), N.ptr(_f32(aabb23)), int(n_grid), int(n_cascade), N.ptr(bitfield.contiguous()),
        float(near_distance), int(bool(inclusive)), int(rng_state), int(rng_inc), N.ptr(zvals), mask.data_ptr(), N.ptr(counts), R,
        N.stream()), 'sparse_sampling_in_multivol_bitfield')
    return (zvals, mask, counts) if want_counts else (zvals, mask)


def generate_grid_samples_multivol(density_grid, n_elements, aabb23, ema_step, n_cascade, n_grid, thresh, inclusive, rng_state,
                                   rng_inc):
    _req(density_grid, aabb23)
    g = _f32(density_grid)
    if g.numel() < int(n_grid) ** 3 * _multivol_levels(n_cascade, inclusive):
        raise RuntimeError('density_grid should hold n_grid**3 * levels cells')
    pos = torch.empty((n_elements, 3), dtype=torch.float32, device=g.device)
    idx = torch.empty((n_elements,), dtype=torch.int32, device=g.device)
    N.check(N.lib().arcn_generate_grid_samples_multivol(N.ptr(g), int(ema_step), int(n_elements), N.ptr(_f32(aabb23)),
                                                       int(n_cascade), int(n_grid), float(thresh), int(bool(inclusive)),
                                                       int(rng_state), int(rng_inc), N.ptr(pos), N.ptr(idx), N.stream()),
            'generate_grid_samples_multivol')
    return pos, idx


def update_bitfield_multivol(density_grid, density_grid_mean, density_grid_bitfield, thres, n_grid, n_cascade, inclusive):
    """In place on density_grid_bitfield; density_grid_mean a python float or a 1-element device tensor."""
    _req(density_grid, density_grid_bitfield)
    cells = int(n_grid) ** 3 * _multivol_levels(n_cascade, inclusive)
    if density_grid.dtype != torch.float32 or not density_grid.is_contiguous() or density_grid.numel() < cells:
        raise RuntimeError('density_grid must be a contiguous float32 tensor of n_grid**3 * levels')
    if density_grid_bitfield.dtype != torch.uint8 or density_grid_bitfield.numel() < cells // 8:
        raise RuntimeError('bitfield should be uint8 in (n_grid**3/8 * levels,)')
    mean_dev, mean_host = None, 0.0
    if torch.is_tensor(density_grid_mean) and density_grid_mean.is_cuda:
        mean_dev = _f32(density_grid_mean).view(-1)
    else:
        mean_host = float(density_grid_mean)
    N.check(N.lib().arcn_update_bitfield_multivol(N.ptr(density_grid), mean_host, N.ptr(mean_dev), N.ptr(density_grid_bitfield),
                                                 float(thres), int(n_grid), int(n_cascade), int(bool(inclusive)), N.stream()),
            'update_bitfield_multivol')
    return density_grid_bitfield


class Pcg32Host:
    """The explicit (seed, call counter) replacement of the reference's file-static `pcg32 rng{9121}`
    (arcnerf/ops/include/common.h:22-23): state before launch k is seed-state advanced k * 2^32."""

    def __init__(self, seed=9121, seq=1):
        self._si = (C.c_uint64 * 2)()
        N.lib().arcn_pcg32_seed(seed, seq, C.addressof(self._si))

    @property
    def state(self):
        return int(self._si[0])

    @property
    def inc(self):
        return int(self._si[1])

    def advance(self, delta=1 << 32):
        N.lib().arcn_pcg32_advance(C.addressof(self._si), delta)

    def set_state(self, state):
        """rewind / restore (e.g. to repeat a launch with larger buffers)"""
        self._si[0] = state


# ------------------------------------------------------------------------------------------------
# compacted sampler
# ------------------------------------------------------------------------------------------------
def march_packed(rays_o, rays_d, aabb23, n_grid, bitfield, n_pts, dt, near_distance, rng_state, rng_inc,
                 packed_bits=False, torch_aabb=False, capacity=None, scratch=None):
    """Bounds + occupancy marching in packed form.

    Returns dict(t (cap,), ray_id (cap,), offsets (R+1,) int32 with offsets[R] = total, counts (R,), near, far).
    No host synchronisation: `total` stays on the device (offsets[-1]); capacity defaults to R*n_pts.
    """
    _req(rays_o, rays_d, aabb23, bitfield)
    o, d = _f32(rays_o), _f32(rays_d)
    aabb = _f32(aabb23)
    bf = bitfield.contiguous().view(torch.uint8)
    R = o.shape[0]
    dev = o.device
    if scratch is None:
        scratch = torch.empty((R, n_pts), dtype=torch.float32, device=dev)
    counts = torch.empty(R, dtype=torch.int32, device=dev)
    near = torch.empty(R, dtype=torch.float32, device=dev)
    far = torch.empty(R, dtype=torch.float32, device=dev)
    offsets = torch.empty(R + 1, dtype=torch.int32, device=dev)
    L = N.lib()
    N.check(L.arcn_march_count(N.ptr(o), N.ptr(d), N.ptr(aabb), int(n_grid), N.ptr(bf), int(packed_bits), int(n_pts),
                               float(dt), float(near_distance), int(torch_aabb), int(rng_state), int(rng_inc),
                               N.ptr(scratch), N.ptr(counts), N.ptr(near), N.ptr(far), R, N.stream()), 'march_count')
    if capacity is None:
        capacity = R * n_pts
    N.check(L.arcn_exclusive_scan_i32(N.ptr(counts), N.ptr(offsets), R, int(capacity), None, N.stream()), 'exclusive_scan_i32')
    t = torch.empty(capacity, dtype=torch.float32, device=dev)
    ray_id = torch.empty(capacity, dtype=torch.int32, device=dev)
    N.check(L.arcn_march_write(N.ptr(scratch), N.ptr(counts), N.ptr(offsets), int(n_pts), N.ptr(t), N.ptr(ray_id), R,
                               capacity, N.stream()), 'march_write')
    return {'t': t, 'ray_id': ray_id, 'offsets': offsets, 'counts': counts, 'near': near, 'far': far}


def pack_dense_samples_begin(zvals, counts):
    """First half of pack_dense_samples: the scan, and the total on its way to pinned host memory (asynchronous copy + event) - the
    caller may queue other work before pack_dense_samples_end waits for it."""
    _req(zvals, counts)
    z = _f32(zvals)
    R, n_pts = z.shape
    cnt = counts.contiguous().to(torch.int32)
    offsets = torch.empty(R + 1, dtype=torch.int32, device=z.device)
    p_dense = (_fresh if R > 0 else torch.zeros)((1,), dtype=torch.int32, device=z.device)       # (the scan's one workgroup writes it)
    N.check(N.lib().arcn_exclusive_scan_i32(N.ptr(cnt), N.ptr(offsets), R, int(R * n_pts), N.ptr(p_dense), N.stream()), 'exclusive_scan_i32')
    host = torch.empty(1, dtype=torch.int32).pin_memory()
    host.copy_(offsets[R:R + 1], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return {'z': z, 'cnt': cnt, 'offsets': offsets, 'p_dense': p_dense, 'host': host, 'event': ev}


def pack_dense_samples_end(h):
    """Second half: wait for the total (the ONE host read), size the packed tensors, compact."""
    h['event'].synchronize()
    total = int(h['host'][0])
    z, cnt, offsets = h['z'], h['cnt'], h['offsets']
    R, n_pts = z.shape
    t = torch.empty(max(total, 1), dtype=torch.float32, device=z.device)
    ray_id = torch.empty(max(total, 1), dtype=torch.int32, device=z.device)
    if total > 0:
        N.check(N.lib().arcn_march_write(N.ptr(z), N.ptr(cnt), N.ptr(offsets), int(n_pts), N.ptr(t), N.ptr(ray_id), R, total, N.stream()),
                'march_write')
    return t[:total], ray_id[:total], offsets, h['p_dense'], total


def pack_dense_samples(zvals, counts):
    """Dense sampler output (R, n_pts) with per-ray counts (valid samples first) -> the packed form the compositor and
    packed_points consume: t (total,), ray_id (total,) int32, offsets (R+1,) int32, p_dense (1,) int32 = max(counts), total.
    ONE host read (the total, to size the packed tensors); exclusive scan + compaction are kernels."""
    return pack_dense_samples_end(pack_dense_samples_begin(zvals, counts))


def packed_points(rays_o, rays_d, t, ray_id, n=None, n_dev=None, want_dirs=True):
    _req(rays_o, rays_d, t, ray_id)
    o, d = _f32(rays_o), _f32(rays_d)
    n = t.shape[0] if n is None else int(n)
    xyz = torch.empty((n, 3), dtype=torch.float32, device=o.device)
    dirs = torch.empty((n, 3), dtype=torch.float32, device=o.device) if want_dirs else None
    N.check(N.lib().arcn_packed_points(N.ptr(o), N.ptr(d), N.ptr(t), N.ptr(ray_id), N.ptr(xyz), N.ptr(dirs), n,
                                      _nptr(n_dev), N.stream()), 'packed_points')
    return xyz, dirs


# ------------------------------------------------------------------------------------------------
# encoders
# ------------------------------------------------------------------------------------------------
_LM_ROWS_MIN = 16384      # below this many points the XCD-affine gather writes its rows itself (one launch; strided 8-byte stores)


def hashgrid_lm_to_rows(lm, desc, n, n_cap, n_dev=None, out=None):
    """level-major features (L, n_cap, F) -> rows (n, L F) (arcn_hashgrid_lm_to_rows)"""
    LF = int(desc.n_levels) * int(desc.n_feat)
    if out is None:
        out = torch.empty((n, LF), dtype=torch.float32, device=lm.device)
    N.check(N.lib().arcn_hashgrid_lm_to_rows(N.ptr(lm), int(desc.n_levels), int(desc.n_feat), int(n_cap), N.ptr(out), LF, n, _nptr(n_dev), N.stream()),
            'hashgrid_lm_to_rows')
    return out


def hashgrid_fwd(xyz, table, desc, want_idx=False, n_dev=None, out=None):
    """HashGridEmbedder.forward's (n, L F) features.  n_feat 1 | 2: the XCD-affine cost-balanced gather (arcn_hashgrid_fwd_xcd, bit-identical to
    the plain one) - level-major + one transposing pass for large launches; want_idx (debug rows) or n_feat 4: the plain kernel"""
    _req(xyz, table)
    xyz, table = _f32(xyz), _f32(table)
    n = xyz.shape[0]
    LF = desc.n_levels * desc.n_feat
    if out is None:
        out = torch.empty((n, LF), dtype=torch.float32, device=xyz.device)
    if not want_idx and desc.n_feat <= 2:
        if n >= _LM_ROWS_MIN:
            lm = torch.empty(n * LF, dtype=torch.float32, device=xyz.device)
            N.check(N.lib().arcn_hashgrid_fwd_xcd(N.ptr(xyz), N.ptr(table), C.addressof(desc), N.ptr(lm), 1, n, n, _nptr(n_dev), N.stream()), 'hashgrid_fwd_xcd')
            return hashgrid_lm_to_rows(lm, desc, n, n, n_dev=n_dev, out=out)
        N.check(N.lib().arcn_hashgrid_fwd_xcd(N.ptr(xyz), N.ptr(table), C.addressof(desc), N.ptr(out), 0, n, n, _nptr(n_dev), N.stream()), 'hashgrid_fwd_xcd')
        return out
    idx = torch.empty((n, desc.n_levels, 8), dtype=torch.int32, device=xyz.device) if want_idx else None
    N.check(N.lib().arcn_hashgrid_fwd(N.ptr(xyz), N.ptr(table), C.addressof(desc), N.ptr(out), N.ptr(idx), n, _nptr(n_dev),
                                     N.stream()), 'hashgrid_fwd')
    return (out, idx) if want_idx else out


def hashgrid_fwd_plain(xyz, table, desc, n_dev=None):
    """the plain one-lane-per-(sample, level) gather (arcn_hashgrid_fwd): what the XCD-affine kernel is checked against"""
    _req(xyz, table)
    xyz, table = _f32(xyz), _f32(table)
    n = xyz.shape[0]
    out = torch.empty((n, desc.n_levels * desc.n_feat), dtype=torch.float32, device=xyz.device)
    N.check(N.lib().arcn_hashgrid_fwd(N.ptr(xyz), N.ptr(table), C.addressof(desc), N.ptr(out), None, n, _nptr(n_dev), N.stream()), 'hashgrid_fwd')
    return out


def hashgrid_bwd_workspace(desc, n, device):
    """scratch of the binned scatter (bin counters + 16-byte corner-pair records); the library knows the size"""
    floats = int(N.lib().arcn_hashgrid_bwd_workspace_floats(C.addressof(desc), int(n)))
    return torch.empty(max(1, floats), dtype=torch.float32, device=device)


def deterministic():
    """True when the library runs its order-independent scatter (ARCN_DETERMINISTIC=1, read once at load)"""
    return bool(N.lib().arcn_deterministic())


def hashgrid_bwd_status(desc, n, workspace):
    """(largest |gradient| of the last binned scatter run with this workspace at capacity n, whether a bin overflowed into the
    order-dependent direct atomics) - a host read; deterministic runs check the flag is False"""
    off = int(N.lib().arcn_hashgrid_bwd_status_offset(C.addressof(desc), int(n)))
    words = workspace[off:off + 2].view(torch.int32).cpu()
    return float(words[:1].view(torch.float32)[0]), bool(int(words[1]) != 0)


def hashgrid_bwd(xyz, table, dout, desc, want_dtable=True, want_dxyz=False, n_dev=None, dtable=None, workspace=None, level_stride=0):
    """workspace: True (allocate) or a float tensor from hashgrid_bwd_workspace -> owner-computes scatter; None -> atomics.
    level_stride > 0: dout is LEVEL-major with that many samples per level (binned scatter only, no dxyz: arcn_hashgrid_bwd_lm)"""
    _req(xyz, table, dout)
    xyz, table, dout = _f32(xyz), _f32(table), _f32(dout)
    n = xyz.shape[0]
    if workspace is True:
        workspace = hashgrid_bwd_workspace(desc, n, xyz.device)
    if workspace is not None:
        assert workspace.dtype == torch.float32
    if want_dtable and dtable is None:
        dtable = torch.zeros_like(table)
    if level_stride:
        assert workspace is not None and want_dtable and not want_dxyz
        N.check(N.lib().arcn_hashgrid_bwd_lm(N.ptr(xyz), N.ptr(dout), int(level_stride), C.addressof(desc), N.ptr(dtable), N.ptr(workspace), workspace.numel(),
                                            n, _nptr(n_dev), N.stream()), 'hashgrid_bwd_lm')
        return dtable, None
    dxyz = torch.zeros((n, 3), dtype=torch.float32, device=xyz.device) if want_dxyz else None
    N.check(N.lib().arcn_hashgrid_bwd(N.ptr(xyz), N.ptr(table), N.ptr(dout), C.addressof(desc),
                                     N.ptr(dtable) if want_dtable else None, N.ptr(dxyz), N.ptr(workspace),
                                     0 if workspace is None else workspace.numel(), n,
                                     _nptr(n_dev), N.stream()), 'hashgrid_bwd')
    return dtable, dxyz


def hashgrid_bwd_bwd(xyz, gdx, table, dout, desc, want_ddout=True, want_dtable=True, want_d2xyz=False, workspace=True, dtable=None):
    """second-order pieces of the encoding's input gradient (see arcn_hashgrid_bwd_bwd) -> ddout, dtable, d2xyz (None if unwanted).
    workspace True: binned table scatter (allocates its scratch); None: one float atomic per corner."""
    _req(xyz, gdx, table, dout)
    xyz, gdx, table, dout = _f32(xyz), _f32(gdx), _f32(table), _f32(dout)
    n = xyz.shape[0]
    ddout = torch.empty_like(dout) if want_ddout else None
    if not want_dtable:
        dtable = None
    elif dtable is None:      # (a given buffer is accumulated into)
        dtable = torch.zeros_like(table)
    d2xyz = torch.zeros((n, 3), dtype=torch.float32, device=xyz.device) if want_d2xyz else None
    ws = None
    if want_dtable and workspace is not None and table.shape[-1] <= 2 and n > 0:
        ws = hashgrid_bwd_workspace(desc, 2 * n, xyz.device) if workspace is True else workspace
    N.check(N.lib().arcn_hashgrid_bwd_bwd(N.ptr(xyz), N.ptr(gdx), N.ptr(table), N.ptr(dout), C.addressof(desc), N.ptr(ddout),
                                         N.ptr(dtable), N.ptr(d2xyz), N.ptr(ws), 0 if ws is None else ws.numel(), n, None,
                                         N.stream()), 'hashgrid_bwd_bwd')
    return ddout, dtable, d2xyz


def hashgrid_fwd_corners(xyz, table, desc, n_dev=None):
    """hashgrid_fwd that also keeps the gathered rows: -> (enc (n, L F), corners) for hashgrid_dxyz_corners / hashgrid_ddout_corners.
    corners: (L, 2 F, n, 4) floats - level-major 16-byte quads of the eight rows of every (sample, level), see arcn_hashgrid_fwd_corners"""
    _req(xyz, table)
    xyz, table = _f32(xyz), _f32(table)
    n, L, Fq = xyz.shape[0], int(desc.n_levels), int(desc.n_feat)
    out = torch.empty((n, L * Fq), dtype=torch.float32, device=xyz.device)
    corners = torch.empty((L, 2 * Fq, n, 4), dtype=torch.float32, device=xyz.device)
    if n >= _LM_ROWS_MIN:
        lm = torch.empty(n * L * Fq, dtype=torch.float32, device=xyz.device)
        N.check(N.lib().arcn_hashgrid_fwd_corners(N.ptr(xyz), N.ptr(table), C.addressof(desc), N.ptr(lm), 1, N.ptr(corners), n, n, _nptr(n_dev), N.stream()),
                'hashgrid_fwd_corners')
        hashgrid_lm_to_rows(lm, desc, n, n, n_dev=n_dev, out=out)
    else:
        N.check(N.lib().arcn_hashgrid_fwd_corners(N.ptr(xyz), N.ptr(table), C.addressof(desc), N.ptr(out), 0, N.ptr(corners), n, n, _nptr(n_dev), N.stream()),
                'hashgrid_fwd_corners')
    return out, corners


def hashgrid_dxyz_corners(xyz, corners, dout, desc, n_dev=None):
    """d <dout, enc(x)> / d x (n, 3) from the forward's corners: the dxyz of hashgrid_bwd, bit for bit, without the table reads"""
    _req(xyz, corners, dout)
    xyz, dout = _f32(xyz), _f32(dout)
    n = xyz.shape[0]
    dxyz = _fresh((n, 3), torch.float32, xyz.device)
    N.check(N.lib().arcn_hashgrid_dxyz_corners(N.ptr(xyz), N.ptr(corners), N.ptr(dout), C.addressof(desc), N.ptr(dxyz), int(corners.shape[2]), n,
                                              _nptr(n_dev), N.stream()), 'hashgrid_dxyz_corners')
    return dxyz


def hashgrid_ddout_corners(xyz, gdx, corners, desc, n_dev=None):
    """the ddout (n, L F) of hashgrid_bwd_bwd from the forward's corners, bit for bit, without the table reads"""
    _req(xyz, gdx, corners)
    xyz, gdx = _f32(xyz), _f32(gdx)
    n = xyz.shape[0]
    ddout = _fresh((n, int(desc.n_levels) * int(desc.n_feat)), torch.float32, xyz.device)
    N.check(N.lib().arcn_hashgrid_ddout_corners(N.ptr(xyz), N.ptr(gdx), N.ptr(corners), C.addressof(desc), N.ptr(ddout), int(corners.shape[2]), n,
                                               _nptr(n_dev), N.stream()), 'hashgrid_ddout_corners')
    return ddout


def hashgrid_bwd_first_second(xyz, dout, gdx, dout_dx, desc, dtable, workspace):
    """dtable += the table gradient through the encoding (dout) AND through its input gradient (gdx on J^T dout_dx): both binned scatters
    with ONE accumulation pass; workspace from hashgrid_bwd_workspace(desc, 3 * n)"""
    _req(xyz, dout, gdx, dout_dx, dtable, workspace)
    xyz, dout, gdx, dout_dx = _f32(xyz), _f32(dout), _f32(gdx), _f32(dout_dx)
    n = xyz.shape[0]
    N.check(N.lib().arcn_hashgrid_bwd_first_second(N.ptr(xyz), N.ptr(dout), N.ptr(gdx), N.ptr(dout_dx), C.addressof(desc), N.ptr(dtable),
                                                  N.ptr(workspace), workspace.numel(), n, N.stream()), 'hashgrid_bwd_first_second')
    return dtable


def _adam_args(h):
    """(lr, beta1, beta2, eps, weight_decay, ema_decay (< 0: none), grad_scale, step, ema_step) of a FusedAdam.begin_step() record"""
    return (float(h['lr']), float(h['betas'][0]), float(h['betas'][1]), float(h['eps']), float(h['weight_decay']),
            float(h['ema_decay']) if h.get('ema_decay') is not None else -1.0, float(h.get('grad_scale', 1.0)), int(h['step']), int(h['ema_step'] or h['step']))


def hashgrid_bwd_adam(xyz, dout, desc, dtable, table, exp_avg, exp_avg_sq, hyper, workspace, level_stride=0):
    """the binned table scatter of hashgrid_bwd (dout (n, L F) sample-major) whose chunk owners also apply Adam (+ the EMA aliased onto the
    parameter) to the table levels they own alone: table / exp_avg / exp_avg_sq = the table's views of a flattened FusedAdam's buffers, hyper =
    FusedAdam.begin_step().  -> the bit mask of the levels done (their gradient never reaches dtable); the rest is accumulated into dtable"""
    _req(xyz, dout, dtable, table, exp_avg, exp_avg_sq, workspace)
    xyz, dout = _f32(xyz), _f32(dout)
    n = xyz.shape[0]
    mask = C.c_uint32(0)
    # (level_stride > 0: dout is LEVEL-major with that many samples per level - geo2_bwd(dx_level_major=True) - instead of (n, L F) rows)
    N.check(N.lib().arcn_hashgrid_bwd_lm_adam(N.ptr(xyz), N.ptr(dout), int(level_stride), C.addressof(desc), N.ptr(dtable), N.ptr(table), N.ptr(exp_avg), N.ptr(exp_avg_sq),
                                             *_adam_args(hyper), N.ptr(workspace), workspace.numel(), 0, n, None, C.cast(C.pointer(mask), C.c_void_p),
                                             N.stream()), 'hashgrid_bwd_lm_adam')
    return int(mask.value)


def hashgrid_bwd_first_second_adam(xyz, dout, gdx, dout_dx, desc, dtable, table, exp_avg, exp_avg_sq, hyper, workspace):
    """hashgrid_bwd_first_second with the optimiser applied by the chunk owners (see hashgrid_bwd_adam) -> mask of the levels done"""
    _req(xyz, dout, gdx, dout_dx, dtable, table, exp_avg, exp_avg_sq, workspace)
    xyz, dout, gdx, dout_dx = _f32(xyz), _f32(dout), _f32(gdx), _f32(dout_dx)
    n = xyz.shape[0]
    mask = C.c_uint32(0)
    N.check(N.lib().arcn_hashgrid_bwd_first_second_adam(N.ptr(xyz), N.ptr(dout), N.ptr(gdx), N.ptr(dout_dx), C.addressof(desc), N.ptr(dtable), N.ptr(table),
                                                       N.ptr(exp_avg), N.ptr(exp_avg_sq), *_adam_args(hyper), N.ptr(workspace), workspace.numel(), n,
                                                       C.cast(C.pointer(mask), C.c_void_p), N.stream()), 'hashgrid_bwd_first_second_adam')
    return int(mask.value)


def freq_fwd(x, n_freqs, include_input=True):
    _req(x)
    x = _f32(x)
    n, D = x.shape
    out = torch.empty((n, D * (1 if include_input else 0) + 2 * D * n_freqs), dtype=torch.float32, device=x.device)
    N.check(N.lib().arcn_freq_fwd(N.ptr(x), D, int(n_freqs), int(include_input), N.ptr(out), n, N.stream()), 'freq_fwd')
    return out


def refresh_cells_points(bitfield_bool, n_grid, perm, voxel_size, min_xyz, rng_state, rng_inc, cells, pts, n_valid, workspace):
    """cells / jittered points of an occupancy refresh (arcn_refresh_cells_points): perm = [(a0, c0), (a1, c1)] of
    geometry.volume.mix_constants; cells (2 n_s) int64, pts (2 n_s, 3) float32, n_valid (1) int32, workspace uint8 - all caller-owned"""
    _req(bitfield_bool, cells, pts, n_valid, workspace)
    assert bitfield_bool.dtype in (torch.bool, torch.uint8) and bitfield_bool.is_contiguous() and cells.dtype == torch.int64 and n_valid.dtype == torch.int32
    pa = (C.c_uint64 * 2)(int(perm[0][0]), int(perm[1][0]))
    pc = (C.c_uint64 * 2)(int(perm[0][1]), int(perm[1][1]))
    mn = (C.c_float * 3)(*[float(v) for v in min_xyz])
    vsz = (C.c_float * 3)(*([float(voxel_size)] * 3 if isinstance(voxel_size, (int, float)) else [float(v) for v in voxel_size]))
    N.check(N.lib().arcn_refresh_cells_points(bitfield_bool.data_ptr(), int(n_grid), C.cast(pa, C.c_void_p), C.cast(pc, C.c_void_p), C.cast(vsz, C.c_void_p),
                                            C.cast(mn, C.c_void_p), int(rng_state), int(rng_inc), N.ptr(cells), N.ptr(pts), N.ptr(n_valid), N.ptr(workspace),
                                            workspace.numel(), N.stream()), 'refresh_cells_points')
    return cells, pts, n_valid


def freq_fwd_cols(x, n_freqs, include_input, out):
    """the encoding of x (n, D) into the columns of `out`, a (n, n_cols) column slice of a wider row-major buffer; columns behind the
    encoding's own width are zeroed"""
    _req(x, out)
    x = _f32(x)
    n, D = x.shape
    assert out.dim() == 2 and out.shape[0] == n and out.dtype == torch.float32 and out.stride(1) == 1
    N.check(N.lib().arcn_freq_fwd_cols(N.ptr(x), D, int(n_freqs), int(include_input), out.data_ptr(), out.stride(0) if n > 1 else out.shape[1],
                                     out.shape[1], n, N.stream()), 'freq_fwd_cols')
    return out


def freq_jvp_cols(x, v, n_freqs, include_input, out):
    """(d enc / d x) v into the columns of `out` (n, n_cols), columns behind the encoding's width zeroed: the adjoint of freq_bwd"""
    _req(x, v, out)
    x, v = _f32(x), _f32(v)
    n, D = x.shape
    assert v.shape == x.shape and out.dim() == 2 and out.shape[0] == n and out.dtype == torch.float32 and out.stride(1) == 1
    N.check(N.lib().arcn_freq_jvp_cols(N.ptr(x), N.ptr(v), D, int(n_freqs), int(include_input), out.data_ptr(), out.stride(0) if n > 1 else out.shape[1],
                                     out.shape[1], n, N.stream()), 'freq_jvp_cols')
    return out


def freq_bwd(x, dout, n_freqs, include_input=True):
    _req(x, dout)
    x, dout = _f32(x), _f32(dout)
    n, D = x.shape
    dx = torch.empty_like(x)
    N.check(N.lib().arcn_freq_bwd(N.ptr(x), N.ptr(dout), D, int(n_freqs), int(include_input), N.ptr(dx), n, N.stream()),
            'freq_bwd')
    return dx


def sh_fwd(dirs, degree, include_input=False):
    _req(dirs)
    dirs = _f32(dirs)
    n = dirs.shape[0]
    out = torch.empty((n, degree * degree + (3 if include_input else 0)), dtype=torch.float32, device=dirs.device)
    N.check(N.lib().arcn_sh_fwd(N.ptr(dirs), int(degree), int(include_input), N.ptr(out), n, N.stream()), 'sh_fwd')
    return out


def radiance_inputs(mode, pts=None, dirs=None, normals=None, feat=None, sh_degree=4):
    """[p | v | n | f] blocks of a radiance net's input in the order of `mode`, one kernel (v = SH(normalize(dirs))); feat may be a column
    slice of a wider row-major tensor"""
    _req(pts, dirs, normals, feat)
    ref = next(t for t in (pts, dirs, normals, feat) if t is not None)
    n = ref.shape[0]
    ld, nf = 0, 0
    if feat is not None:
        assert feat.dim() == 2 and feat.stride(1) == 1 and feat.dtype == torch.float32
        ld, nf = feat.stride(0), feat.shape[1]
    width = sum({'p': 3, 'v': sh_degree * sh_degree, 'n': 3, 'f': nf}[c] for c in mode)
    out = torch.empty((n, width), dtype=torch.float32, device=ref.device)
    N.check(N.lib().arcn_radiance_inputs(mode.encode(), N.ptr(_f32(pts)), N.ptr(_f32(dirs)), N.ptr(_f32(normals)),
                                        None if feat is None else feat.data_ptr(), int(ld), int(nf), int(sh_degree), N.ptr(out), n, N.stream()),
            'radiance_inputs')
    return out


def act_fwd(x, act, beta=1.0):
    _req(x)
    x = _f32(x)
    y = torch.empty_like(x)
    N.check(N.lib().arcn_act_fwd(N.ptr(x), N.ptr(y), x.numel(), N.ACT[act], float(beta), N.stream()), 'act_fwd')
    return y


def act_col_scale(x, act, scale, n_dev=None, out=None, beta=1.0):
    """act(x[:, 0]) * scale for a (n, C) row-major x, rows behind the device count left alone"""
    _req(x, out)
    assert x.dim() == 2 and x.is_contiguous() and x.dtype == torch.float32
    n = x.shape[0]
    y = torch.empty(n, dtype=torch.float32, device=x.device) if out is None else out
    N.check(N.lib().arcn_act_col_scale(N.ptr(x), x.shape[1], N.ptr(y), n, _nptr(n_dev), N.ACT[act], float(beta), float(scale), N.stream()), 'act_col_scale')
    return y


def sdf_jac_dz(dh, u, s, c):
    """(dh s + c u s (1 - s), s u) over an (n, H) hidden layer, c (H): written over dh and u"""
    _req(dh, u, s, c)
    n, H = dh.shape
    N.check(N.lib().arcn_sdf_jac_dz(N.ptr(dh), N.ptr(u), N.ptr(s), N.ptr(c), N.ptr(dh), N.ptr(u), n, H, N.stream()), 'sdf_jac_dz')
    return dh, u


def tonemap_fwd(x, params):
    """x (n, C), params (C, 3 W + 1) = [w1 | b1 | w2 | b2] per channel -> sigmoid(b2 + w2 . relu(w1 x + b1)) (n, C)"""
    _req(x, params)
    x, params = _f32(x), _f32(params)
    n, Cc = x.shape
    W = (params.shape[1] - 1) // 3
    y = torch.empty_like(x)
    N.check(N.lib().arcn_tonemap_fwd(N.ptr(x), N.ptr(params), N.ptr(y), n, Cc, W, N.stream()), 'tonemap_fwd')
    return y


def tonemap_bwd(x, y, dy, params, want_dx=True):
    _req(x, y, dy, params)
    x, y, dy, params = _f32(x), _f32(y), _f32(dy), _f32(params)
    n, Cc = x.shape
    W = (params.shape[1] - 1) // 3
    dx = torch.empty_like(x) if want_dx else None
    dparams = torch.empty_like(params)
    nf = max(1, int(N.lib().arcn_tonemap_scratch_floats(n, Cc, W)))
    scratch = torch.empty(nf, dtype=torch.float32, device=x.device)
    N.check(N.lib().arcn_tonemap_bwd(N.ptr(x), N.ptr(y), N.ptr(dy), N.ptr(params), N.ptr(dx), N.ptr(dparams), N.ptr(scratch), nf, n, Cc, W,
                                   N.stream()), 'tonemap_bwd')
    return dx, dparams


def softplus_grad(z, g, beta, from_y=False):
    """g * sigmoid(beta z) (torch softplus threshold 20; g None: the sigmoid itself); from_y: `z` is y = softplus(z)
    (sigmoid(beta z) = 1 - exp(-beta y))"""
    _req(z, g)
    z, g = _f32(z), _f32(g)
    out = torch.empty_like(z)
    N.check(N.lib().arcn_softplus_grad(N.ptr(z), N.ptr(g), N.ptr(out), z.numel(), float(beta), int(from_y), N.stream()), 'softplus_grad')
    return out


def softplus_grad_row(z, g_row, beta, from_y=False):
    """g_row[col] * sigmoid(beta z) for z (n, H) and ONE gradient row (H) shared by all samples"""
    _req(z, g_row)
    z, g_row = _f32(z), _f32(g_row)
    n, H = z.shape
    assert g_row.numel() == H
    out = torch.empty_like(z)
    N.check(N.lib().arcn_softplus_grad_row(N.ptr(z), N.ptr(g_row), N.ptr(out), n, H, float(beta), int(from_y), N.stream()), 'softplus_grad_row')
    return out


def softplus_grad2_row(z, g_row, h, colsum, beta, from_y=False):
    """backward of softplus_grad_row for an incoming h (n, H): -> dz = h * g_row * ds; colsum (H) += the column sums of h * s"""
    _req(z, g_row, h, colsum)
    z, g_row, h = _f32(z), _f32(g_row), _f32(h)
    n, H = z.shape
    assert g_row.numel() == H and colsum.numel() == H and colsum.is_contiguous() and colsum.dtype == torch.float32
    dz = torch.empty_like(z)
    N.check(N.lib().arcn_softplus_grad2_row(N.ptr(z), N.ptr(g_row), N.ptr(h), N.ptr(dz), N.ptr(colsum), n, H, float(beta), int(from_y), N.stream()),
            'softplus_grad2_row')
    return dz


def concat2_div(a, b, div, n_cols):
    """(n, n_cols) = [a / div | b / div | 0]; a (n, na), b (n, nb) | None: column slices of row-major tensors; `/ div` as torch's CUDA kernels
    divide by a python scalar (a product with the float reciprocal)"""
    _req(a, b)
    n = a.shape[0]
    assert a.dim() == 2 and a.dtype == torch.float32 and a.stride(1) == 1
    nb, ld_b = 0, 0
    if b is not None:
        assert b.dim() == 2 and b.dtype == torch.float32 and b.stride(1) == 1 and b.shape[0] == n
        nb, ld_b = b.shape[1], b.stride(0)
    out = torch.empty((n, int(n_cols)), dtype=torch.float32, device=a.device)
    N.check(N.lib().arcn_concat2_div(a.data_ptr(), int(a.stride(0)), int(a.shape[1]), None if b is None else b.data_ptr(), int(ld_b), int(nb), float(div),
                                    N.ptr(out), int(n_cols), n, N.stream()), 'concat2_div')
    return out


def softplus_grad_sum(z, g, g2, beta, from_y=False):
    """(g + g2) * sigmoid(beta z) in one pass (softplus_grad of the sum of two incoming gradients)"""
    _req(z, g, g2)
    z, g, g2 = _f32(z), _f32(g), _f32(g2)
    out = torch.empty_like(z)
    N.check(N.lib().arcn_softplus_grad_sum(N.ptr(z), N.ptr(g), N.ptr(g2), N.ptr(out), z.numel(), float(beta), int(from_y), N.stream()), 'softplus_grad_sum')
    return out


def softplus_grad2(z, g, h, beta, want_dg=True, want_dz=True, from_y=False):
    """backward of softplus_grad for an incoming h: (h * s, h * g * beta s (1 - s)), s = sigmoid(beta z), in one pass; from_y: the second
    is the gradient with respect to y, h * g * beta (1 - s)"""
    _req(z, g, h)
    z, g, h = _f32(z), _f32(g), _f32(h)
    dg = torch.empty_like(z) if want_dg else None
    dz = torch.empty_like(z) if want_dz else None
    N.check(N.lib().arcn_softplus_grad2(N.ptr(z), N.ptr(g), N.ptr(h), N.ptr(dg), N.ptr(dz), z.numel(), float(beta), int(from_y), N.stream()),
            'softplus_grad2')
    return dg, dz


def act_bwd_bwd(x, dy, g, act, beta=1.0, want_ddy=True, want_d2x=True):
    """second backward of an elementwise activation: (ddy, d2x) = (g f'(x), g dy f''(x))"""
    _req(x, dy, g)
    x, dy, g = _f32(x), _f32(dy), _f32(g)
    ddy = torch.empty_like(x) if want_ddy else None
    d2x = torch.empty_like(x) if want_d2x else None
    N.check(N.lib().arcn_act_bwd_bwd(N.ptr(x), N.ptr(dy), N.ptr(g), N.ptr(ddy), N.ptr(d2x), x.numel(), N.ACT[act], float(beta), N.stream()), 'act_bwd_bwd')
    return ddy, d2x


def act_bwd(x, y, dy, act, beta=1.0):
    _req(x, dy)
    x, y, dy = _f32(x), (None if y is None else _f32(y)), _f32(dy)     # y None: the kernel evaluates the activation itself
    dx = torch.empty_like(x)
    N.check(N.lib().arcn_act_bwd(N.ptr(x), N.ptr(y), N.ptr(dy), N.ptr(dx), x.numel(), N.ACT[act], float(beta), N.stream()),
            'act_bwd')
    return dx


def ngp_glue_fwd(geo_out, dirs, feat_off, Wf, sh_degree, feat_first=True, sigma_act='truncexp', n_dev=None,
                 rad_in=None, sigma=None):
    _req(geo_out, dirs)
    n, Wg = geo_out.shape
    W = Wf + sh_degree * sh_degree
    if rad_in is None:
        rad_in = torch.empty((n, W), dtype=torch.float32, device=geo_out.device)
    if sigma is None:
        sigma = torch.empty(n, dtype=torch.float32, device=geo_out.device)
    N.check(N.lib().arcn_ngp_glue_fwd(N.ptr(geo_out), N.ptr(dirs), Wg, int(feat_off), int(Wf), int(sh_degree),
                                     int(feat_first), N.ACT[sigma_act], N.ptr(rad_in), N.ptr(sigma), n, _nptr(n_dev),
                                     N.stream()), 'ngp_glue_fwd')
    return rad_in, sigma


def ngp_glue_bwd(geo_out, d_rad_in, d_sigma, feat_off, Wf, sh_degree, feat_first=True, sigma_act='truncexp', n_dev=None,
                 d_geo_out=None):
    _req(geo_out, d_rad_in, d_sigma)
    n, Wg = geo_out.shape
    if d_geo_out is None:
        d_geo_out = torch.empty_like(geo_out)
    N.check(N.lib().arcn_ngp_glue_bwd(N.ptr(geo_out), N.ptr(d_rad_in), N.ptr(d_sigma), Wg, int(feat_off), int(Wf),
                                     int(sh_degree), int(feat_first), N.ACT[sigma_act], N.ptr(d_geo_out), n, _nptr(n_dev),
                                     N.stream()), 'ngp_glue_bwd')
    return d_geo_out


# ------------------------------------------------------------------------------------------------
# fused MLP
# ------------------------------------------------------------------------------------------------
def mlp_acts_floats(desc, n_cap):
    return int(N.lib().arcn_mlp_acts_floats(C.addressof(desc), int(n_cap)))


def mlp_scratch_floats(desc, n_cap):
    return int(N.lib().arcn_mlp_scratch_floats(C.addressof(desc), int(n_cap)))


def mlp_fwd(x, weights, biases, desc, save_acts=False, n_dev=None, out=None, acts=None):
    """weights: flat fp32 tensor of all W_i (row-major (out,in)) concatenated; biases likewise or None."""
    _req(x, weights, biases)
    x = _f32(x)
    n = x.shape[0]
    if out is None:
        out = torch.empty((n, desc.dims[desc.n_layers]), dtype=torch.float32, device=x.device)
    if save_acts and acts is None:
        acts = torch.empty(max(1, mlp_acts_floats(desc, n)), dtype=torch.float32, device=x.device)
    N.check(N.lib().arcn_mlp_fwd(N.ptr(x), N.ptr(weights), N.ptr(biases), C.addressof(desc), N.ptr(out),
                                N.ptr(acts) if save_acts else None, n, n, _nptr(n_dev), N.stream()), 'mlp_fwd')
    return (out, acts) if save_acts else out


# ------------------------------------------------------------------------------------------------
# dense layers of the wide nets: the three f32-MFMA products of csrc/gemm.hip
# ------------------------------------------------------------------------------------------------
# layers with more than 64 outputs run on the bf16 matrix rate with every operand split into three bf16 planes (six products, f32
# accuracy; csrc/gemm.hip); narrower ones on the exact-f32 MFMA kernels
_GEMM_SPLIT_MIN_OUT = 65


def _use_split(rows, k_red, n_out):
    return n_out >= _GEMM_SPLIT_MIN_OUT and k_red % 4 == 0 and k_red >= 32 and rows.data_ptr() % 16 == 0


def _split_ws(n_out, k_red, device):
    return torch.empty(int(N.lib().arcn_gemm_split_bytes(n_out, k_red)), dtype=torch.uint8, device=device)


# The weights of a layer as split bf16 planes, ONCE for all the chunks of samples one forward pass runs the layer on (chunk_processing
# opens the scope around its loop: the arguments it does not slice - the networks - are the same objects in every iteration).  Inside
# a scope the split of a weight tensor is kept under its storage address (the tensor itself is kept too, so the address cannot be
# handed out again) together with the zero-padded forms ops.autograd makes of odd-width layers; outside nothing is cached and every
# product splits for itself.
_SPLIT_SCOPE = None


@contextlib.contextmanager
def split_weight_scope():
    global _SPLIT_SCOPE
    opened = _SPLIT_SCOPE is None
    if opened:
        _SPLIT_SCOPE = {}
    try:
        yield
    finally:
        if opened:
            _SPLIT_SCOPE = None


def scope_cached(key, keep, make):
    """make() once per split_weight_scope for `key` (anything hashable naming tensors by address; `keep` = those tensors, held so
    that the addresses stay theirs); plain make() outside a scope"""
    if _SPLIT_SCOPE is None:
        return make()
    hit = _SPLIT_SCOPE.get(key)
    if hit is None:
        hit = (make(), keep)
        _SPLIT_SCOPE[key] = hit
    return hit[0]


def split_weights(w, transposed):
    """ws of arcn_gemm_split_weights for the layer weight w (N, K): transposed = False for gemm_nt (outputs N, reduction K), True for
    gemm_nn (outputs K, reduction N)"""
    Nn, K = w.shape
    n_out, k_red = (K, Nn) if transposed else (Nn, K)

    def make():
        ws = _split_ws(n_out, k_red, w.device)
        N.check(N.lib().arcn_gemm_split_weights(N.ptr(w), K, 1 if transposed else 0, n_out, k_red, N.ptr(ws), ws.numel(), N.stream()), 'gemm_split_weights')
        return ws
    return scope_cached(('split', w.data_ptr(), w._version, Nn, K, bool(transposed)), w, make)


def in_split_scope():
    return _SPLIT_SCOPE is not None


def relu_bits_supported(rows, k_red, n_out):
    """whether gemm_nt(..., want_bits=True) can write the ReLU mask of its output as bits (split kernels, outputs in multiples of 32)"""
    return _use_split(rows, k_red, n_out) and n_out % 4 == 0


def _rows(t):
    """(tensor, leading dimension) of a 2-D fp32 row operand: a column slice of a wider row-major buffer (unit column stride) is taken
    as it is - the kernels read and write rows at any stride -, anything else is made contiguous"""
    if t is None:
        return None, 0
    if t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] and t.shape[0] > 1:
        return t, t.stride(0)
    t = t.contiguous().float()
    return t, t.shape[1]


def _aligned_rows(t, ld):
    return t.data_ptr() % 16 == 0 and ld % 4 == 0


def gemm_nt(x, w, bias=None, act=None, beta=1.0, want_bits=False, ws=None, out=None):
    """y (S,N) = act(x (S,K) @ w (N,K).T + bias); want_bits (act = relu, relu_bits_supported): also the (ceil(S / 8), N / 4) int32 words
    [s // 8, f // 4] whose bit 4 (s % 8) + (f % 4) is (y[s, f] > 0) - the mask gemm_nn / gemm_tn take as `mask_bits` (1/32 of y's bytes).
    x and `out` (where the result goes if given) may be column slices of wider row-major buffers; ws: split_weights(w, False)"""
    _req(x, w, bias, out)
    (x, ld_x), w, bias = _rows(x), _f32(w), _f32(bias)
    S, K = x.shape
    Nn = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        y, ld_y = torch.empty((S, Nn), dtype=torch.float32, device=x.device), Nn
    else:
        y, ld_y = out, out.stride(0)
        assert out.shape == (S, Nn) and out.dtype == torch.float32 and out.stride(1) == 1
    if _use_split(x, K, Nn) and ld_x % 4 == 0:
        ready = ws is not None or in_split_scope()
        if ws is None:
            ws = split_weights(w, False) if ready else _split_ws(Nn, K, x.device)
        bits = None
        if want_bits:
            assert act == 'relu' and Nn % 4 == 0
            bits = torch.empty(((S + 7) // 8, Nn // 4), dtype=torch.int32, device=x.device)
        N.check(N.lib().arcn_gemm_nt_split(x.data_ptr(), ld_x, N.ptr(w), N.ptr(bias), y.data_ptr(), N.ptr(bits), ld_y, S, None, K, Nn, N.ACT[act],
                                         float(beta), N.ptr(ws), ws.numel(), 1 if ready else 0, N.stream()), 'gemm_nt_split')
        return (y, bits) if want_bits else y
    assert not want_bits
    N.check(N.lib().arcn_gemm_nt(x.data_ptr(), ld_x, N.ptr(w), N.ptr(bias), y.data_ptr(), ld_y, S, None, K, Nn, N.ACT[act], float(beta), N.stream()),
            'gemm_nt')
    return y


def gemm_nn(dy, w, mask=None, mask_bits=None, ws=None):
    """dx (S,K) = (dy * (mask > 0)) (S,N) @ w (N,K); mask_bits: the forward's ReLU bit words instead of the float mask.  dy (and a float
    mask, at the SAME row stride) may be column slices of wider buffers; ws: split_weights(w, True)"""
    _req(dy, w, mask, mask_bits)
    (dy, ld), w = _rows(dy), _f32(w)
    if mask is not None:
        mask, ld_m = _rows(mask)
        if ld_m != ld:      # the kernels address dy and its mask with one stride
            dy, mask = dy.contiguous(), mask.contiguous()
            ld = dy.shape[1]
    S, Nn = dy.shape
    K = w.shape[1]
    assert w.shape[0] == Nn
    dx = torch.empty((S, K), dtype=torch.float32, device=dy.device)
    if _use_split(dy, Nn, K) and ld % 4 == 0:
        ready = ws is not None or in_split_scope()
        if ws is None:
            ws = split_weights(w, True) if ready else _split_ws(K, Nn, dy.device)
        N.check(N.lib().arcn_gemm_nn_split(dy.data_ptr(), None if mask is None else mask.data_ptr(), N.ptr(mask_bits), ld, N.ptr(w), N.ptr(dx), K, S,
                                         None, Nn, K, N.ptr(ws), ws.numel(), 1 if ready else 0, N.stream()), 'gemm_nn_split')
        return dx
    assert mask_bits is None, 'bit masks are read by the split products only'
    N.check(N.lib().arcn_gemm_nn(dy.data_ptr(), None if mask is None else mask.data_ptr(), ld, N.ptr(w), N.ptr(dx), K, S, None, Nn, K, N.stream()),
            'gemm_nn')
    return dx


def gemm_tn(dy, x, mask=None, want_colsum=False, mask_bits=None, out=None, db_out=None, accumulate=False, head=None):
    """dw (N,K) = (dy * (mask > 0)) (S,N).T @ x (S,K), reduced over the rows in a fixed order; want_colsum: also the column sums (N) of
    dy * (mask > 0) - a layer's bias gradient - from the same pass where the split kernel runs, else from a second product with ones;
    mask_bits: the forward's ReLU bit words instead of the float mask.  dy (+ float mask, same stride) and x may be column slices.
    out / db_out (contiguous) receive the results; accumulate: they are ADDED to (a caller that sums a layer's gradient over chunks);
    head: only the first `head` of the N rows are kept, dw is (head, K) (a layer with padded output columns; plain product only)"""
    _req(dy, x, mask, mask_bits)
    (dy, ld), (x, ld_x) = _rows(dy), _rows(x)
    if mask is not None:
        mask, ld_m = _rows(mask)
        if ld_m != ld:
            dy, mask = dy.contiguous(), mask.contiguous()
            ld = dy.shape[1]
    S, Nn = dy.shape
    K = x.shape[1]
    assert x.shape[0] == S
    if head is not None:
        assert not want_colsum and mask_bits is None and 1 <= head <= Nn
        dw = torch.empty((head, K), dtype=torch.float32, device=dy.device) if out is None else out
        assert dw.shape == (head, K) and dw.is_contiguous() and dw.dtype == torch.float32 and (out is not None or not accumulate)
        nf = max(1, int(N.lib().arcn_gemm_tn_scratch_floats(S, Nn, K)))
        scratch = torch.empty(nf, dtype=torch.float32, device=dy.device)
        N.check(N.lib().arcn_gemm_tn_head(dy.data_ptr(), None if mask is None else mask.data_ptr(), ld, x.data_ptr(), ld_x, N.ptr(dw), N.ptr(scratch), nf,
                                         S, None, Nn, K, int(head), 1 if accumulate else 0, N.stream()), 'gemm_tn_head')
        return dw
    dw = torch.empty((Nn, K), dtype=torch.float32, device=dy.device) if out is None else out
    assert dw.shape == (Nn, K) and dw.is_contiguous() and dw.dtype == torch.float32 and (out is not None or not accumulate)
    acc = 1 if accumulate else 0
    nf = max(1, int(N.lib().arcn_gemm_tn_scratch_floats(S, Nn, K)))
    scratch = torch.empty(nf, dtype=torch.float32, device=dy.device)
    mptr = None if mask is None else mask.data_ptr()
    split = (Nn > 64 or K > 64) and Nn % 4 == 0 and K % 4 == 0 and _aligned_rows(dy, ld) and _aligned_rows(x, ld_x)
    if split:
        db = (torch.empty(Nn, dtype=torch.float32, device=dy.device) if db_out is None else db_out) if want_colsum else None
        N.check(N.lib().arcn_gemm_tn_split(dy.data_ptr(), mptr, N.ptr(mask_bits), ld, x.data_ptr(), ld_x, N.ptr(dw), N.ptr(db), N.ptr(scratch), nf, S,
                                         None, Nn, K, acc, N.stream()), 'gemm_tn_split')
        return (dw, db) if want_colsum else dw
    assert mask_bits is None, 'bit masks are read by the split products only'
    N.check(N.lib().arcn_gemm_tn(dy.data_ptr(), mptr, ld, x.data_ptr(), ld_x, N.ptr(dw), N.ptr(scratch), nf, S, None, Nn, K, acc, N.stream()), 'gemm_tn')
    if not want_colsum:
        return dw
    ones = _ones_cols(S, dy.device)
    db = gemm_tn(dy, ones, mask)[:, 0]
    if db_out is None:
        return dw, db.contiguous()
    if accumulate:
        db_out += db
    else:
        db_out.copy_(db)
    return dw, db_out


_ONES = {}


def _ones_cols(n_rows, device):
    key = str(device)
    t = _ONES.get(key)
    if t is None or t.shape[0] < n_rows:
        t = torch.ones((max(n_rows, 1 << 16), 4), dtype=torch.float32, device=device)
        _ONES[key] = t
    return t[:n_rows]


def mlp_bwd(x, weights, biases, desc, out, acts, dout, want_dx=True, n_dev=None, dweights=None, dbiases=None, scratch=None):
    _req(x, weights, out, dout)
    x, dout = _f32(x), _f32(dout)
    n = x.shape[0]
    dx = torch.empty_like(x) if want_dx else None
    if dweights is None:
        dweights = torch.zeros_like(weights)
    if biases is not None and dbiases is None:
        dbiases = torch.zeros_like(biases)
    if scratch is None:
        scratch = torch.empty(mlp_scratch_floats(desc, n), dtype=torch.float32, device=x.device)
    N.check(N.lib().arcn_mlp_bwd(N.ptr(x), N.ptr(weights), N.ptr(biases), C.addressof(desc), N.ptr(out), N.ptr(acts),
                                N.ptr(dout), N.ptr(dx), N.ptr(dweights), N.ptr(dbiases), N.ptr(scratch), n, n,
                                _nptr(n_dev), N.stream()), 'mlp_bwd')
    return dx, dweights, dbiases


# ------------------------------------------------------------------------------------------------
# compositing
# ------------------------------------------------------------------------------------------------
def _bkg(bkg_color, R):
    if bkg_color is None:
        return None, 0
    b = _f32(bkg_color).view(-1, 3)
    if b.shape[0] not in (1, R):
        raise RuntimeError('Only bkg with N_rays/1 allowed..')
    return b, b.shape[0]


def ray_marching_fwd(sigma, radiance, zvals, add_inf_z=False, white_bkg=False, alpha=None, bkg_color=None, noise=None,
                     want_samples=True, check_order=True):
    _req(sigma, radiance, zvals, alpha, bkg_color, noise)
    z = _f32(zvals)
    R, P = z.shape
    sg, al, rad, ns = _f32(sigma), _f32(alpha), _f32(radiance), _f32(noise)
    bk, bk_rows = _bkg(bkg_color, R)
    Pe = P if (add_inf_z or al is not None) else P - 1
    dev = z.device
    rgb = torch.empty((R, 3), dtype=torch.float32, device=dev) if rad is not None else None
    depth = torch.empty(R, dtype=torch.float32, device=dev)
    mask = torch.empty(R, dtype=torch.float32, device=dev)
    a_o = torch.empty((R, Pe), dtype=torch.float32, device=dev) if want_samples else None
    t_o = torch.empty((R, Pe), dtype=torch.float32, device=dev) if want_samples else None
    w_o = torch.empty((R, Pe), dtype=torch.float32, device=dev) if want_samples else None
    status = torch.zeros(1, dtype=torch.int32, device=dev) if check_order else None
    N.check(N.lib().arcn_ray_marching_fwd(N.ptr(sg), N.ptr(al), N.ptr(rad), N.ptr(z), N.ptr(ns), N.ptr(bk), bk_rows, R, P,
                                         int(add_inf_z), int(white_bkg), N.ptr(rgb), N.ptr(depth), N.ptr(mask),
                                         N.ptr(a_o), N.ptr(t_o), N.ptr(w_o), N.ptr(status), N.stream()), 'ray_marching_fwd')
    return {'rgb': rgb, 'depth': depth, 'mask': mask, 'alpha': a_o, 'trans_shift': t_o, 'weights': w_o, 'status': status}


def ray_marching_bwd(sigma, radiance, zvals, d_rgb, d_depth=None, d_mask=None, add_inf_z=False, white_bkg=False,
                     alpha=None, bkg_color=None, noise=None, d_tlast=None):
    _req(sigma, radiance, zvals, alpha, bkg_color, noise, d_rgb, d_depth, d_mask, d_tlast)
    z = _f32(zvals)
    R, P = z.shape
    sg, al, rad, ns = _f32(sigma), _f32(alpha), _f32(radiance), _f32(noise)
    bk, bk_rows = _bkg(bkg_color, R)
    d_geo = torch.empty((R, P), dtype=torch.float32, device=z.device)
    d_rad = torch.empty((R, P, 3), dtype=torch.float32, device=z.device) if rad is not None else None
    N.check(N.lib().arcn_ray_marching_bwd(N.ptr(sg), N.ptr(al), N.ptr(rad), N.ptr(z), N.ptr(ns), N.ptr(bk), bk_rows, R, P,
                                         int(add_inf_z), int(white_bkg), N.ptr(_f32(d_rgb)), N.ptr(_f32(d_depth)),
                                         N.ptr(_f32(d_mask)), N.ptr(_f32(d_tlast)), N.ptr(d_geo), N.ptr(d_rad), N.stream()),
            'ray_marching_bwd')
    return d_geo, d_rad


def composite_packed_fwd(sigma, radiance, t, offsets, p_dense=2, p_dense_dev=None, add_inf_z=False, white_bkg=False,
                         bkg_color=None, noise=None, want_weights=False, counts=None):
    _req(sigma, radiance, t, offsets, bkg_color, noise, counts)
    sg, rad, t, ns = _f32(sigma), _f32(radiance), _f32(t), _f32(noise)
    R = offsets.shape[0] - 1
    bk, bk_rows = _bkg(bkg_color, R)
    dev = t.device
    rgb = torch.empty((R, 3), dtype=torch.float32, device=dev) if rad is not None else None
    depth = torch.empty(R, dtype=torch.float32, device=dev)
    mask = torch.empty(R, dtype=torch.float32, device=dev)
    w = torch.zeros(sg.shape[0], dtype=torch.float32, device=dev) if want_weights else None
    N.check(N.lib().arcn_composite_packed_fwd(N.ptr(sg), N.ptr(rad), N.ptr(t), N.ptr(offsets), N.ptr(ns), N.ptr(bk), bk_rows,
                                             R, int(p_dense), _nptr(p_dense_dev), int(add_inf_z), int(white_bkg), N.ptr(rgb),
                                             N.ptr(depth), N.ptr(mask), N.ptr(w), N.ptr(counts), N.stream()), 'composite_packed_fwd')
    return {'rgb': rgb, 'depth': depth, 'mask': mask, 'weights': w}


def composite_packed_bwd(sigma, radiance, t, offsets, d_rgb, d_depth=None, d_mask=None, p_dense=2, p_dense_dev=None,
                         add_inf_z=False, white_bkg=False, bkg_color=None, noise=None, counts=None):
    _req(sigma, radiance, t, offsets, bkg_color, noise, d_rgb, d_depth, d_mask, counts)
    sg, rad, t, ns = _f32(sigma), _f32(radiance), _f32(t), _f32(noise)
    R = offsets.shape[0] - 1
    bk, bk_rows = _bkg(bkg_color, R)
    # (every sample of every segment is written: the visited columns, the dropped last column, the samples of a truncated ray - render.hip)
    d_sigma = _fresh(sg.shape, torch.float32, sg.device)
    d_rad = _fresh(rad.shape, torch.float32, rad.device) if rad is not None else None
    N.check(N.lib().arcn_composite_packed_bwd(N.ptr(sg), N.ptr(rad), N.ptr(t), N.ptr(offsets), N.ptr(ns), N.ptr(bk), bk_rows,
                                             R, int(p_dense), _nptr(p_dense_dev), int(add_inf_z), int(white_bkg),
                                             N.ptr(_f32(d_rgb)), N.ptr(_f32(d_depth)), N.ptr(_f32(d_mask)), N.ptr(d_sigma),
                                             N.ptr(d_rad), N.ptr(counts), N.stream()), 'composite_packed_bwd')
    return d_sigma, d_rad


def huber_loss_grad(x, y, delta, weight=1.0, dx=None, loss=None):
    """weight * mean(Huber_delta(x - y)) and its gradient wrt x, one kernel, no host sync (loss stays on the device)"""
    _req(x, y)
    x, y = _f32(x), _f32(y)
    if dx is None:
        dx = torch.empty_like(x)
    if loss is None:
        loss = torch.zeros(1, dtype=torch.float32, device=x.device)
    N.check(N.lib().arcn_huber_loss_grad(N.ptr(x), N.ptr(y), x.numel(), float(delta), float(weight), N.ptr(dx), N.ptr(loss),
                                        N.stream()), 'huber_loss_grad')
    return loss, dx


def _scale_tensor(s, device):
    """the NeuS scale as a device scalar (a python float is uploaded once; a tensor - exp(10 * inv_s) - is used in place)"""
    if torch.is_tensor(s):
        _req(s)
        return _f32(s).reshape(1)
    return scalar_tensor(s, device)


def sdf_to_alpha_fwd(mid_sdf, zvals, mid_slope, s, clip=True):
    """NeuS sdf_to_alpha (models/neus_model.py:242-265): mid_sdf, mid_slope (R,P-1), zvals (R,P) -> alpha (R,P-1)"""
    _req(mid_sdf, zvals, mid_slope)
    sd, z, sl = _f32(mid_sdf), _f32(zvals), _f32(mid_slope)
    R, P = z.shape
    assert sd.shape == (R, P - 1) and sl.shape == (R, P - 1)
    alpha = torch.empty((R, P - 1), dtype=torch.float32, device=z.device)
    sv = _scale_tensor(s, z.device)
    N.check(N.lib().arcn_sdf_to_alpha_fwd(N.ptr(sd), N.ptr(z), N.ptr(sl), N.ptr(sv), int(bool(clip)), N.ptr(alpha), R, P, N.stream()),
            'sdf_to_alpha_fwd')
    return alpha


def sdf_to_alpha_bwd(mid_sdf, zvals, mid_slope, s, d_alpha, clip=True):
    """-> d mid_sdf, d mid_slope (R,P-1), d s (1,) device tensor"""
    _req(mid_sdf, zvals, mid_slope, d_alpha)
    sd, z, sl, da = _f32(mid_sdf), _f32(zvals), _f32(mid_slope), _f32(d_alpha)
    R, P = z.shape
    d_sdf, d_slope = torch.empty_like(sd), torch.empty_like(sl)
    d_s = torch.zeros(1, dtype=torch.float32, device=z.device)
    sv = _scale_tensor(s, z.device)
    N.check(N.lib().arcn_sdf_to_alpha_bwd(N.ptr(sd), N.ptr(z), N.ptr(sl), N.ptr(sv), int(bool(clip)), N.ptr(da), N.ptr(d_sdf),
                                         N.ptr(d_slope), N.ptr(d_s), R, P, N.stream()), 'sdf_to_alpha_bwd')
    return d_sdf, d_slope, d_s


def sample_cdf(bins, cdf, u, eps=1e-5, sort=True, want_inds=False):
    _req(bins, cdf, u)
    bins, cdf, u = _f32(bins), _f32(cdf), _f32(u)
    R, n_pts = bins.shape
    n_sample = u.shape[1]
    samples = torch.empty((R, n_sample), dtype=torch.float32, device=bins.device)
    inds = torch.empty((R, n_sample), dtype=torch.int32, device=bins.device) if want_inds else None
    N.check(N.lib().arcn_sample_cdf(N.ptr(bins), N.ptr(cdf), N.ptr(u), R, n_pts, n_sample, float(eps), int(sort),
                                   N.ptr(samples), N.ptr(inds), N.stream()), 'sample_cdf')
    return (samples, inds) if want_inds else samples


# ------------------------------------------------------------------------------------------------
# occupancy update, optimiser
# ------------------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------------
# NeuS on packed samples (csrc/neus.hip)
# ------------------------------------------------------------------------------------------------
def march_count(rays_o, rays_d, aabb23, n_grid, bitfield, n_pts, dt, near_distance, rng_state, rng_inc, packed_bits=False, torch_aabb=False):
    """bounds (K2) + occupancy marching (K3), one launch: -> zvals_dense (R, n_pts) valid-first (tails NOT filled), counts (R) int32, near, far"""
    _req(rays_o, rays_d, aabb23, bitfield)
    o, d, aabb = _f32(rays_o), _f32(rays_d), _f32(aabb23)
    bf = bitfield.contiguous().view(torch.uint8)
    R, dev = o.shape[0], o.device
    z = torch.empty((R, n_pts), dtype=torch.float32, device=dev)
    counts = torch.empty(R, dtype=torch.int32, device=dev)
    near = torch.empty(R, dtype=torch.float32, device=dev)
    far = torch.empty(R, dtype=torch.float32, device=dev)
    N.check(N.lib().arcn_march_count(N.ptr(o), N.ptr(d), N.ptr(aabb), int(n_grid), N.ptr(bf), int(packed_bits), int(n_pts), float(dt),
                                    float(near_distance), int(torch_aabb), int(rng_state), int(rng_inc), N.ptr(z), N.ptr(counts),
                                    N.ptr(near), N.ptr(far), R, N.stream()), 'march_count')
    return z, counts, near, far


def neus_pack_begin(zvals_dense, counts):
    """First half of neus_pack: the scans, and (total, longest ray) on their way to pinned host memory (asynchronous copy + event) -
    the caller may queue other work, or run this on a side stream for the NEXT batch, before neus_pack_end waits for it."""
    _req(zvals_dense, counts)
    R, n_pts = zvals_dense.shape
    dev = zvals_dense.device
    L = N.lib()
    kmax = (_fresh if R > 0 else torch.zeros)((1,), dtype=torch.int32, device=dev)       # (the scan's one workgroup writes it)
    tmp = torch.empty(R + 1, dtype=torch.int32, device=dev)
    N.check(L.arcn_exclusive_scan_i32(N.ptr(counts), N.ptr(tmp), R, int(R * n_pts), N.ptr(kmax), N.stream()), 'exclusive_scan_i32')
    n_eval = torch.empty(R, dtype=torch.int32, device=dev)
    N.check(L.arcn_neus_count(N.ptr(counts), N.ptr(kmax), R, N.ptr(n_eval), N.stream()), 'neus_count')
    offsets = torch.empty(R + 1, dtype=torch.int32, device=dev)
    N.check(L.arcn_exclusive_scan_i32(N.ptr(n_eval), N.ptr(offsets), R, int(R * (n_pts + 1)), None, N.stream()), 'exclusive_scan_i32')
    both = torch.stack([offsets[R], kmax[0]])
    host = torch.empty(2, dtype=torch.int32).pin_memory()
    host.copy_(both, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return {'z': zvals_dense, 'counts': counts, 'offsets': offsets, 'kmax': kmax, 'n_eval': n_eval, 'host': host, 'event': ev, 'keep': (tmp, both)}


def neus_pack_end(h, n_sample_cfg, want_map=False):
    """Second half: wait for the totals (the ONE host read), size the section tensors, lay the sections out."""
    h['event'].synchronize()
    total, k_h = int(h['host'][0]), int(h['host'][1])
    zvals_dense, counts, offsets = h['z'], h['counts'], h['offsets']
    R, n_pts = zvals_dense.shape
    dev = zvals_dense.device
    p_dense = max(2, int(k_h))
    out = {'offsets': offsets, 'kmax': h['kmax'], 'p_dense': p_dense, 'total': int(total), 'n_eval': h['n_eval']}
    n_alloc = max(1, int(total))
    out['t_mid'] = torch.empty(n_alloc, dtype=torch.float32, device=dev)
    out['lo'] = torch.empty(n_alloc, dtype=torch.float32, device=dev)
    out['hi'] = torch.empty(n_alloc, dtype=torch.float32, device=dev)
    out['ray_id'] = torch.empty(n_alloc, dtype=torch.int32, device=dev)
    out['slot_map'] = torch.empty((R, p_dense), dtype=torch.int64, device=dev) if want_map else None
    N.check(N.lib().arcn_neus_sections(N.ptr(zvals_dense), N.ptr(counts), N.ptr(offsets), int(n_pts), float(n_sample_cfg), R, p_dense,
                                       N.ptr(out['t_mid']), N.ptr(out['lo']), N.ptr(out['hi']), N.ptr(out['ray_id']), N.ptr(out['slot_map']),
                                       N.stream()), 'neus_sections')
    if total > 0:      # (total == 0: the one-element buffers stay, nothing reads them)
        for k in ('t_mid', 'lo', 'hi', 'ray_id'):
            out[k] = out[k][:int(total)]
    return out


def neus_pack(zvals_dense, counts, n_sample_cfg, want_map=False):
    """The section layout of NeuS for marched samples (see csrc/neus.hip) -> dict(t_mid, lo, hi, ray_id (total), offsets (R+1) int32,
    kmax (1) int32 device, p_dense int, total int, slot_map (R, p_dense) int64 | None).  ONE host read (total and the longest ray)."""
    return neus_pack_end(neus_pack_begin(zvals_dense, counts), n_sample_cfg, want_map)


def neus_slots_fwd(packed, offsets, p_dense, dflt):
    _req(packed, offsets)
    R = offsets.shape[0] - 1
    dense = torch.empty((R, int(p_dense), 3), dtype=torch.float32, device=packed.device)
    N.check(N.lib().arcn_neus_slots_fwd(N.ptr(_f32(packed)), N.ptr(offsets), R, int(p_dense), _vec3(dflt), N.ptr(dense), N.stream()), 'neus_slots_fwd')
    return dense


def neus_slots_bwd(d_dense, offsets, p_dense, n_points):
    _req(d_dense, offsets)
    R = offsets.shape[0] - 1
    d_packed = torch.zeros((max(1, int(n_points)), 3), dtype=torch.float32, device=d_dense.device)
    N.check(N.lib().arcn_neus_slots_bwd(N.ptr(_f32(d_dense)), N.ptr(offsets), R, int(p_dense), N.ptr(d_packed), N.stream()), 'neus_slots_bwd')
    return d_packed


def _vec3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def eikonal_packed(normal, pk, n_rays, weight, d_normal=None, loss=None, add_src=None, loss_is_clear=False):
    """EikonalLoss on the dense `normal_pts` of a packed NeuS batch without building it: -> (loss (1,) device, d_normal (S,3)); a given
    d_normal is ADDED to; add_src ((S, 3) column slice of a row-major tensor): a second incoming gradient joined in the same pass;
    loss_is_clear: the given loss buffer is an accumulator already cleared (arcn_neus_blend_loss's loss[1]) - no clearing launch"""
    _req(normal, d_normal, loss, add_src)
    normal = _f32(normal)
    S = normal.shape[0]
    acc = (1 if d_normal is not None else 0) | (2 if (loss is not None and loss_is_clear) else 0)
    if d_normal is None:
        d_normal = torch.empty_like(normal)
    if loss is None:
        loss = torch.empty(1, dtype=torch.float32, device=normal.device)
    ld = 0
    if add_src is not None:
        assert add_src.shape == (S, 3) and add_src.dtype == torch.float32 and add_src.stride(1) == 1
        ld = add_src.stride(0)
    N.check(N.lib().arcn_eikonal_packed(N.ptr(normal), N.ptr(pk['ray_id']), N.ptr(pk['offsets']), S, int(n_rays), int(pk['p_dense']), float(weight),
                                       int(acc), None if add_src is None else add_src.data_ptr(), int(ld), N.ptr(d_normal), N.ptr(loss),
                                       N.stream()), 'eikonal_packed')
    return loss, d_normal


# ---- the passes between the kernels of trainer.FusedNeusNgpStep (csrc/step_glue.hip) ----------------------------------------------------------
def neus_step_prep(w1, l1w, beta, inv_s=None, speed=1.0, bkg_l1w=None, pad_to=4):
    """the per-step derived weights of the sdf net (first layer w1 (H, E), last layer l1w (n_out, H)) and of a background density net's last
    layer, one launch -> dict(w2p (n_pad, H), w1j (H, E) = w1 * l1w[0][:, None], bw20 (H) = beta * l1w[0], scale (1) = exp(inv_s * speed) |
    None, wb1p (nb_pad, Hb) | None)"""
    _req(w1, l1w, inv_s, bkg_l1w)
    w1, l1w = _f32(w1), _f32(l1w)
    H, E = w1.shape
    n_out = l1w.shape[0]
    assert l1w.shape[1] == H
    n_pad = (n_out + pad_to - 1) // pad_to * pad_to
    dev = w1.device
    flat = torch.empty(n_pad * H + H * E + H + 4, dtype=torch.float32, device=dev)       # (one allocation, every piece 16-byte aligned when H % 4 == 0)
    o = {'w2p': flat[:n_pad * H].view(n_pad, H), 'w1j': flat[n_pad * H:n_pad * H + H * E].view(H, E),
         'bw20': flat[n_pad * H + H * E:n_pad * H + H * E + H], 'scale': flat[n_pad * H + H * E + H:n_pad * H + H * E + H + 1] if inv_s is not None else None,
         'wb1p': None}
    Hb = nb_out = nb_pad = 0
    if bkg_l1w is not None:
        bkg_l1w = _f32(bkg_l1w)
        nb_out, Hb = bkg_l1w.shape
        nb_pad = (nb_out + pad_to - 1) // pad_to * pad_to
        o['wb1p'] = torch.empty((nb_pad, Hb), dtype=torch.float32, device=dev)
    N.check(N.lib().arcn_neus_step_prep(N.ptr(w1), N.ptr(l1w), H, E, n_out, n_pad, float(beta), N.ptr(_f32(inv_s)), float(speed), N.ptr(o['w2p']),
                                       N.ptr(o['w1j']), N.ptr(o['bw20']), N.ptr(o['scale']), N.ptr(bkg_l1w), Hb, nb_out, nb_pad, N.ptr(o['wb1p']),
                                       N.stream()), 'neus_step_prep')
    return o


def geo_out_grad(d_col0, d_feat, n_pad, out=None, act=None, beta=1.0, y_col0=None):
    """g_out (n, n_pad) = [d_col0 * act'(out[:, 0]) | d_feat | 0]: d_feat (n, n_feat) and out (n, >= 1) may be column slices of wider row-major
    tensors; act None: column 0 = d_col0"""
    _req(d_col0, d_feat, out, y_col0)
    n = d_col0.shape[0]
    assert d_col0.is_contiguous() and d_col0.dtype == torch.float32 and d_feat.dtype == torch.float32 and d_feat.stride(1) == 1 and d_feat.shape[0] == n
    g = torch.empty((n, int(n_pad)), dtype=torch.float32, device=d_col0.device)
    ld_out = 0
    if N.ACT[act] != 0:
        assert out is not None and out.dtype == torch.float32 and out.shape[0] == n and (out.dim() == 1 or out.stride(1) == 1)
        ld_out = out.stride(0)
    N.check(N.lib().arcn_geo_out_grad(N.ptr(d_col0), None if out is None else out.data_ptr(), int(ld_out), N.ptr(y_col0), N.ACT[act], float(beta),
                                     d_feat.data_ptr(), int(d_feat.stride(0)), int(d_feat.shape[1]), int(n_pad), N.ptr(g), n, N.stream()), 'geo_out_grad')
    return g


def hashgrid_fwd_lm(xyz, table, desc, want_corners=False, n_dev=None):
    """the XCD-affine gather leaving its features LEVEL-major: -> (lm (L * n * F,) = lm[(l * n + s) * F + f], corners | None); the operand layout
    of geo2_fwd / geo2_bwd and of the fused MLP's level-major entry points (level stride = n)"""
    _req(xyz, table)
    xyz, table = _f32(xyz), _f32(table)
    n, L, Fq = xyz.shape[0], int(desc.n_levels), int(desc.n_feat)
    lm = _fresh((n * L * Fq,), torch.float32, xyz.device)
    if want_corners:
        corners = _fresh((L, 2 * Fq, n, 4), torch.float32, xyz.device)
        N.check(N.lib().arcn_hashgrid_fwd_corners(N.ptr(xyz), N.ptr(table), C.addressof(desc), N.ptr(lm), 1, N.ptr(corners), n, n, _nptr(n_dev), N.stream()),
                'hashgrid_fwd_corners')
        return lm, corners
    N.check(N.lib().arcn_hashgrid_fwd_xcd(N.ptr(xyz), N.ptr(table), C.addressof(desc), N.ptr(lm), 1, n, n, _nptr(n_dev), N.stream()), 'hashgrid_fwd_xcd')
    return lm, None


def geo2_fwd(x_lm, n, w1, w2, jac, beta=1.0, pad_to=4, rows=False):
    """a two-layer geometry net on level-major hash features (level stride n) in one launch (arcn_geo2_fwd): w1 (64, 32), w2 (n_out, 64).
    jac True: the sdf net - softplus(beta) hidden layer -> (out (n, n_pad), sdf (n) = out[:, 0], jac (n, 32) = d sdf / d features);
    jac False: a density net - ReLU hidden layer -> (out, exp(out[:, 0]), None)"""
    _req(x_lm, w1, w2)
    assert tuple(w1.shape) == (64, 32) and w2.shape[1] == 64 and w1.is_contiguous() and w2.is_contiguous()
    n_out = int(w2.shape[0])
    n_pad = (n_out + pad_to - 1) // pad_to * pad_to
    dev = x_lm.device
    out = _fresh((n, n_pad), torch.float32, dev)
    head = _fresh((n,), torch.float32, dev)
    jc = _fresh((n, 32), torch.float32, dev) if jac else None
    # (rows: x_lm is the (n, 32) row-major feature tensor instead of the level-major one)
    N.check(N.lib().arcn_geo2_fwd(N.ptr(x_lm), 0 if rows else int(n), N.ptr(_f32(w1)), N.ptr(_f32(w2)), n_out, n_pad, int(bool(jac)), float(beta), N.ptr(out), N.ptr(head),
                                 N.ptr(jc), int(n), None, N.stream()), 'geo2_fwd')
    return out, head, jc


def geo2_bwd(x_lm, n, w1, w2, jac, beta, d_col0, d_feat, dw1, dw2, d_jac=None, out=None, dx_level_major=False, scratch=None, rows=False):
    """the backward of geo2_fwd in one launch + its weight-gradient reduction (arcn_geo2_bwd): d_col0 (n) = the gradient of the head (sdf, or the
    density through its exp), d_feat (n, n_out - 1) = the gradient of the feature columns (may be a column slice of a wider row-major tensor),
    d_jac (n, 32) = the gradient of the Jacobian row (jac True), out = the forward's output (jac False: its column 0 is the TruncExp's input).
    dw1 (64, 32) += and dw2 (n_out, 64) += (views of one flat gradient buffer).  -> dx: (n, 32) rows, or level-major (32 n,) with dx_level_major"""
    _req(x_lm, w1, w2, d_col0, d_feat, dw1, dw2, d_jac, out)
    n_out = int(w2.shape[0])
    assert d_col0.dtype == torch.float32 and d_col0.dim() == 1 and d_col0.numel() == n        # (may be column 0 of a wider row-major gradient)
    assert d_feat.dtype == torch.float32 and d_feat.stride(1) == 1 and d_feat.shape == (n, n_out - 1)
    assert dw1.is_contiguous() and dw2.is_contiguous() and tuple(dw1.shape) == (64, 32) and tuple(dw2.shape) == (n_out, 64)
    dev = x_lm.device
    dx = _fresh((n * 32,), torch.float32, dev) if dx_level_major else _fresh((n, 32), torch.float32, dev)
    need = int(N.lib().arcn_geo2_bwd_scratch_floats(int(n)))
    if scratch is None or scratch.numel() < need:
        scratch = torch.empty(max(1, need), dtype=torch.float32, device=dev)
    ld_out = 0
    if not jac:
        assert out is not None and out.dtype == torch.float32 and out.stride(1) == 1 and out.shape[0] == n
        ld_out = out.stride(0)
    else:
        assert d_jac is not None and d_jac.is_contiguous() and tuple(d_jac.shape) == (n, 32)
    N.check(N.lib().arcn_geo2_bwd(N.ptr(x_lm), 0 if rows else int(n), N.ptr(_f32(w1)), N.ptr(_f32(w2)), n_out, int(bool(jac)), float(beta), d_col0.data_ptr(),
                                 max(1, int(d_col0.stride(0))), None if out is None else out.data_ptr(), int(ld_out), d_feat.data_ptr(), int(d_feat.stride(0)), N.ptr(d_jac), N.ptr(dx),
                                 int(n) if dx_level_major else 0, N.ptr(dw1), N.ptr(dw2), N.ptr(scratch), int(n), None, N.stream()), 'geo2_bwd')
    return dx


_BLEND_WS = {}


def neus_blend_loss(rgb_f, depth_f, t_last, rgb_b, depth_b, target, huber_delta, weight):
    """rgb = rgb_f + T rgb_b, depth likewise, the image loss (Huber: huber_delta > 0, MSE: None / <= 0; plain mean x weight) and the gradients, one
    launch -> dict(rgb, depth, loss (2,): [image loss, 0 = the cleared accumulator of the next loss pass], d_rgb, d_tlast, d_rgb_b)"""
    _req(rgb_f, depth_f, t_last, rgb_b, depth_b, target)
    R = rgb_f.shape[0]
    dev = rgb_f.device
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    o = {'rgb': f(R, 3), 'depth': f(R), 'loss': f(2), 'd_rgb': f(R, 3), 'd_tlast': f(R), 'd_rgb_b': f(R, 3)}
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)      # (ticket + partials: one per stream, zeroed once - the kernel leaves the ticket zero)
    ws = _BLEND_WS.get(key)
    if ws is None:
        ws = _BLEND_WS[key] = torch.zeros(int(N.lib().arcn_neus_blend_loss_workspace_words()), dtype=torch.int32, device=dev)
    N.check(N.lib().arcn_neus_blend_loss(N.ptr(_f32(rgb_f)), N.ptr(_f32(depth_f)), N.ptr(_f32(t_last)), N.ptr(_f32(rgb_b)), N.ptr(_f32(depth_b)),
                                        N.ptr(_f32(target)), R, float(huber_delta or 0.0), float(weight), N.ptr(o['rgb']), N.ptr(o['depth']),
                                        N.ptr(o['d_rgb']), N.ptr(o['d_tlast']), N.ptr(o['d_rgb_b']), N.ptr(o['loss']), N.ptr(ws), N.stream()), 'neus_blend_loss')
    return o


def sdf_jac_dz2(dh, u, s, c, w, colsum):
    """(dh s + c u s (1 - s), s * w) written over dh and u; colsum (H) += the column sums of s u"""
    _req(dh, u, s, c, w, colsum)
    n, H = dh.shape
    assert colsum.numel() == H and colsum.is_contiguous()
    N.check(N.lib().arcn_sdf_jac_dz2(N.ptr(dh), N.ptr(u), N.ptr(s), N.ptr(c), N.ptr(w), N.ptr(dh), N.ptr(u), N.ptr(colsum), n, H, N.stream()), 'sdf_jac_dz2')
    return dh, u


def sum_scale_add(src, dst, factor=1.0, scale_dev=None):
    """dst[0] += factor * scale_dev[0] * sum(src), one launch"""
    _req(src, dst, scale_dev)
    N.check(N.lib().arcn_sum_scale_add(N.ptr(_f32(src)), src.numel(), N.ptr(scale_dev), float(factor), N.ptr(dst), N.stream()), 'sum_scale_add')
    return dst


def neus_render_fwd(sdf, radiance, normal, pk, rays_d, s_dev, cos_anneal, bkg_color, depth_far, dflt_rgb, dflt_nrm):
    _req(sdf, radiance, normal, rays_d, s_dev, bkg_color)
    R = rays_d.shape[0]
    dev = rays_d.device
    bk, bk_rows = _bkg(bkg_color, R)
    rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
    nrm = torch.empty((R, 3), dtype=torch.float32, device=dev)
    depth = torch.empty(R, dtype=torch.float32, device=dev)
    mask = torch.empty(R, dtype=torch.float32, device=dev)
    t_last = torch.empty(R, dtype=torch.float32, device=dev)
    N.check(N.lib().arcn_neus_render_fwd(N.ptr(sdf), N.ptr(radiance), N.ptr(normal), N.ptr(pk['t_mid']), N.ptr(pk['lo']), N.ptr(pk['hi']),
                                        N.ptr(pk['offsets']), N.ptr(rays_d), N.ptr(s_dev), float(cos_anneal), N.ptr(bk), bk_rows,
                                        N.ptr(pk['kmax']), float(depth_far), _vec3(dflt_rgb), _vec3(dflt_nrm), R, N.ptr(rgb), N.ptr(depth),
                                        N.ptr(mask), N.ptr(nrm), N.ptr(t_last), N.stream()), 'neus_render_fwd')
    return rgb, depth, mask, nrm, t_last


def neus_render_bwd(sdf, radiance, normal, pk, rays_d, s_dev, cos_anneal, bkg_color, d_rgb, d_depth, d_mask, d_nrm, d_tlast):
    _req(sdf, radiance, normal, rays_d, s_dev, bkg_color, d_rgb, d_depth, d_mask, d_nrm, d_tlast)
    R = rays_d.shape[0]
    bk, bk_rows = _bkg(bkg_color, R)
    d_sdf = torch.empty_like(sdf)
    d_rad = torch.empty_like(radiance)
    d_normal = torch.empty_like(normal)
    d_s_ray = torch.empty(R, dtype=torch.float32, device=sdf.device)
    N.check(N.lib().arcn_neus_render_bwd(N.ptr(sdf), N.ptr(radiance), N.ptr(normal), N.ptr(pk['t_mid']), N.ptr(pk['lo']), N.ptr(pk['hi']),
                                        N.ptr(pk['offsets']), N.ptr(rays_d), N.ptr(s_dev), float(cos_anneal), N.ptr(bk), bk_rows,
                                        N.ptr(pk['kmax']), R, N.ptr(_f32(d_rgb)), N.ptr(_f32(d_depth)), N.ptr(_f32(d_mask)),
                                        N.ptr(_f32(d_nrm)), N.ptr(_f32(d_tlast)), N.ptr(d_sdf), N.ptr(d_rad), N.ptr(d_normal),
                                        N.ptr(d_s_ray), N.stream()), 'neus_render_bwd')
    return d_sdf, d_rad, d_normal, d_s_ray


def sample_pdf(bins, weights, u, eps=1e-5, sort=True, want_cdf=False):
    """bins (R, n_pts), weights (R, n_pts-1), u (1 | R, n_sample) -> samples (R, n_sample) sorted [, cdf (R, n_pts)]"""
    _req(bins, weights, u)
    R, n_pts = bins.shape
    assert weights.shape == (R, n_pts - 1) and u.dim() == 2 and u.shape[0] in (1, R)
    n_sample = u.shape[1]
    samples = torch.empty((R, n_sample), dtype=torch.float32, device=bins.device)
    cdf = torch.empty((R, n_pts), dtype=torch.float32, device=bins.device) if want_cdf else None
    N.check(N.lib().arcn_sample_pdf(N.ptr(bins), N.ptr(weights), N.ptr(u), R, n_pts, n_sample, int(u.shape[0]), float(eps), int(sort),
                                   N.ptr(samples), N.ptr(cdf), N.stream()), 'sample_pdf')
    return (samples, cdf) if want_cdf else samples


def update_opafield(opafield, flat_idx, opacity, ema=None):
    _req(opafield, flat_idx, opacity)
    assert opafield.is_contiguous() and opafield.dtype == torch.float32
    idx = flat_idx.contiguous().long()
    op = _f32(opacity)
    N.check(N.lib().arcn_update_opafield(N.ptr(opafield), N.ptr(idx), N.ptr(op), idx.shape[0],
                                        -1.0 if ema is None else float(ema), N.stream()), 'update_opafield')
    return opafield


def opafield_scatter_update(opafield, cell_idx, opacity, ema=None, cell_max=None, touched=None, n_dev=None):
    """unique + segmented max + EMA update of the occupancy field for repeated flat cell indices, without a sort."""
    _req(opafield, cell_idx, opacity)
    assert opafield.is_contiguous() and opafield.dtype == torch.float32
    idx = cell_idx.contiguous().long()
    op = _f32(opacity)
    nc = opafield.numel()
    if cell_max is None:
        cell_max = torch.empty(nc, dtype=torch.float32, device=opafield.device)
    if touched is None:
        touched = torch.empty(nc, dtype=torch.uint8, device=opafield.device)
    N.check(N.lib().arcn_opafield_scatter_update(N.ptr(opafield), N.ptr(idx), N.ptr(op), idx.shape[0], _nptr(n_dev), nc,
                                                -1.0 if ema is None else float(ema), N.ptr(cell_max), N.ptr(touched),
                                                N.stream()), 'opafield_scatter_update')
    return opafield


def update_bitfield_by_opafield(opafield, bitfield, threshold):
    _req(opafield, bitfield)
    assert opafield.is_contiguous() and bitfield.is_contiguous()
    ws = torch.zeros(2, dtype=torch.float32, device=opafield.device)
    N.check(N.lib().arcn_update_bitfield_by_opafield(N.ptr(opafield), bitfield.view(torch.uint8).data_ptr(), opafield.numel(),
                                                    float(threshold), N.ptr(ws), N.stream()), 'update_bitfield_by_opafield')
    return bitfield


def adam_ema_step_runs(param, grad, exp_avg, exp_avg_sq, ema, runs, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_decay=0.95,
                       grad_scale=1.0, ema_step=None, zero_grad=False):
    """adam_ema_step on the runs [(lo, hi), ...] (at most four, every lo a multiple of 4) of the flat buffers in ONE launch"""
    _req(param, grad, exp_avg, exp_avg_sq, ema)
    flat = (C.c_int64 * (2 * len(runs)))(*[v for lo, hi in runs for v in (int(lo), int(hi) - int(lo))])
    N.check(N.lib().arcn_adam_ema_step_runs(N.ptr(param), N.ptr(grad), N.ptr(exp_avg), N.ptr(exp_avg_sq), N.ptr(ema), C.cast(flat, C.c_void_p), len(runs),
                                           float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), float(ema_decay),
                                           float(grad_scale), int(step), int(step if ema_step is None else ema_step), int(zero_grad), N.stream()),
            'adam_ema_step_runs')


def adam_ema_step(param, grad, exp_avg, exp_avg_sq, ema, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                  ema_decay=0.95, grad_scale=1.0, ema_step=None, zero_grad=False):
    """torch.optim.Adam step + the reference's EMA.ema_step (average written back into param), one pass."""
    _req(param, grad, exp_avg, exp_avg_sq, ema)
    N.check(N.lib().arcn_adam_ema_step(N.ptr(param), N.ptr(grad), N.ptr(exp_avg), N.ptr(exp_avg_sq), N.ptr(ema),
                                      param.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                      float(weight_decay), float(ema_decay), float(grad_scale), int(step),
                                      int(step if ema_step is None else ema_step), int(zero_grad), N.stream()),
            'adam_ema_step')
