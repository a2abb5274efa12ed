"""ORACLE - TEST INFRASTRUCTURE ONLY: the reference's training-batch fetch restated with numpy.

Never imported by the product package.  What the reference does to its concatenated per-pixel tensors between the dataset and the model
(arcnerf/trainer/pipeline.py): `step_crop_center_image` (:95-130: view(N, H, W, ...)[:, dh:-dh, dw:-dw] -> (N, Hc*Wc, ...)), the row gather of
`step_ray_sample` / `fetch_step_ray_sample` (:132-174, :243-277: rows of the (1, N*Hc*Wc, ...) tensor) and `fetch_step_bkg_color` (:279-300:
img * mask + (1 - mask) * bkg_color), on float32, in the reference's operation order.  The per-pixel rays of a view are the oracle's get_rays
(orc_volume.c, pinned to the reference's get_rays by golden G20) in the order Base3dDataset.precache_ray stores them (wh_order=False).
Pinned to a run of the reference's own Pipeline by tests/test_oracle_psnr_golden.py (golden G27: every batch of 600 iterations by checksums, two
batches in full)."""
import numpy as np

from . import oracle as orc

F32 = np.float32


def dataset_rays(H, W, intrinsic, c2w, center_pixel=True, normalize_rays_d=True):
    """(N,3,3), (N,4,4) -> rays_o, rays_d (N, H*W, 3), rays_r (N, H*W, 1): cameras[i].get_rays(wh_order=False, ...) per view"""
    o, d, r = [], [], []
    for K, M in zip(np.asarray(intrinsic, F32), np.asarray(c2w, F32)):
        ro, rd, rr = orc.get_rays(W, H, K, M, wh_order=False, center_pixel=center_pixel, normalize_rays_d=normalize_rays_d)
        o.append(ro), d.append(rd), r.append(rr)
    return np.stack(o), np.stack(d), np.stack(r)


def crop(t, H, W, window):
    """step_crop_center_image on one (N, H*W, ...) tensor -> (N * Hc * Wc, ...) rows"""
    y0, x0, hc, wc = window
    full = t.reshape(t.shape[0], H, W, *t.shape[2:])
    c = full[:, y0:y0 + hc, x0:x0 + wc]
    return np.ascontiguousarray(c).reshape(t.shape[0] * hc * wc, *t.shape[2:])


def fetch_train_batch(ids, n_img, H, W, window=None, rgba=None, img=None, mask=None, intrinsic=None, c2w=None, center_pixel=True,
                      normalize_rays_d=True, bkg_rand=None, bkg_const=None, rays=None):
    """ids (n,) rows of the cropped dataset tensor -> dict of (n, ...) float32 arrays: rays_o / rays_d / rays_r (cameras given, or `rays` =
    a cached dataset_rays result), img / mask / bkg_color (colours given; blended only when the data has a mask and a colour is asked for)"""
    window = (0, 0, H, W) if window is None else tuple(int(v) for v in window)
    ids = np.asarray(ids, np.int64)
    out = {}
    if rays is None and intrinsic is not None:
        rays = dataset_rays(H, W, intrinsic, c2w, center_pixel, normalize_rays_d)
    if rays is not None:
        for k, t in zip(('rays_o', 'rays_d', 'rays_r'), rays):
            out[k] = crop(t, H, W, window)[ids]
    if rgba is not None:        # NeRF.read_image_list (nerf_dataset.py:107-119): astype(float32) / 255.0, mask = alpha
        f = np.asarray(rgba).reshape(n_img, H * W, 4).astype(F32) / F32(255.0)
        img, mask = f[..., :3], f[..., 3]
    if img is not None:
        out['img'] = crop(np.asarray(img, F32).reshape(n_img, H * W, 3), H, W, window)[ids]
        if mask is not None:
            out['mask'] = crop(np.asarray(mask, F32).reshape(n_img, H * W), H, W, window)[ids]
            if bkg_rand is not None or bkg_const is not None:
                bkg = np.asarray(bkg_rand, F32) if bkg_rand is not None else np.ones_like(out['img']) * np.asarray(bkg_const, F32)[None]
                m = out['mask'][:, None]
                out['img'] = (out['img'] * m + (F32(1.0) - m) * bkg).astype(F32)
                out['bkg_color'] = bkg.astype(F32)
    return out
