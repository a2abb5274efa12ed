"""Inputs of golden G27 (tests/golden/make_golden_psnr.py: the reference's training loop WITH ITS OWN Pipeline - centre precrop, cross-view
shuffle, random background colours, dynamic batch size - run for 600 iterations x 4 seeds on a small analytic scene, held-out PSNR at
checkpoints) - shared by the generator (build container, imports the reference) and by the CPU / GPU tests.

Data generators only - no reference code.  The scene is the analytic one of tools/psnr_curve.py (six soft blobs, textured, mildly view
dependent) rendered ONCE by the generator to 8-bit RGBA images (what a Blender dataset is: straight colours + alpha) that travel in the
fixture; everything random the loop draws comes from numpy PCG64 streams keyed by (seed, purpose, counter), so that every implementation
can be fed the same numbers.
"""
import math

import numpy as np

H = W = 100
N_TRAIN, N_TEST = 24, 4
ANGLE, RADIUS = 0.6911, 3.0 / 1.05                               # Blender's camera_angle_x; scale_radius 3.0 / 1.05 (base_3d_dataset.py:208-224)
N_GRID, N_SAMPLE = 32, 256
EPOCH_OPTIM, EPOCH_WARMUP, UPDATE_EPOCH = 8, 32, 8
LOG_MAX_ALLOWANCE = 14
N_RAYS0, N_RAYS_MAX = 1024, 2048
PRECROP_RATIO, PRECROP_MAX_EPOCH = 0.5, 50
QUIRK_MAX_EPOCH, QUIRK_EPOCHS = 100, 130                         # the data-only leg: a pass over the cropped rays (59 batches) ends before precrop.max_epoch
N_EPOCH = 600
CHECKPOINTS = (50, 100, 200, 400, 600)                           # held-out PSNR after this many iterations
SEEDS = (0, 1, 2, 3)
N_KEEP_REFRESH = 12                                              # bitfields + near-threshold masks kept for the first refreshes of a run
SUMMARY_STEPS = (1, 8, 24)
NEAR_BAND = 1e-4
TABLE_AMP = 1e-4                                                 # HashGridEmbedder's own init range (hashgrid_encoder.py:155-156)


def _rng(seed, purpose, counter=0):
    return np.random.default_rng([27, int(seed), int(purpose), int(counter)])


# ---- cameras ------------------------------------------------------------------------------------------------------------------------
def camera(view):
    """view -> (K (3,3), c2w (4,4)) float32: a pinhole on the sphere of radius 3 / 1.05 looking at the origin, x right, y down, z forward"""
    rng = _rng(0, 0, view)
    c = rng.standard_normal(3)
    c = c / np.linalg.norm(c) * RADIUS
    fwd = -c / np.linalg.norm(c)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right = right / np.linalg.norm(right)
    up = np.cross(right, fwd)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, -up, fwd, c
    focal = 0.5 * W / math.tan(0.5 * ANGLE)
    K = np.array([[focal, 0.0, W / 2.0], [0.0, focal, H / 2.0], [0.0, 0.0, 1.0]])
    return K.astype(np.float32), c2w.astype(np.float32)


def cameras(first, n):
    ks, ms = zip(*[camera(v) for v in range(first, first + n)])
    return np.stack(ks), np.stack(ms)


TRAIN_VIEWS, TEST_VIEWS = (0, N_TRAIN), (1000, N_TEST)


# ---- the analytic scene (tools/psnr_curve.py:build_scene, in double) -------------------------------------------------------------------
def _scene():
    rng = np.random.default_rng(3)
    centers = (rng.random((6, 3)) - 0.5) * 1.0
    radii = rng.random(6) * 0.18 + 0.12
    phase = rng.random((6, 3)) * 6.28
    return centers, radii, phase


def scene_field(x, d):
    """density (N,), colour (N, 3) at positions x (N, 3) seen along d (N, 3)"""
    centers, radii, phase = _scene()
    r2 = ((x[:, None, :] - centers[None]) ** 2).sum(-1) / (radii[None] ** 2)
    w = 1.0 / (1.0 + np.exp(np.minimum(-(1.0 - r2) * 12.0, 700.0)))
    sigma = 60.0 * w.max(axis=1)
    base = 0.5 + 0.5 * np.sin(phase[None] + 4.0 * x[:, None, :])
    col = (w[..., None] * base).sum(1) / (w.sum(1, keepdims=True) + 1e-6)
    col = np.clip(col * (0.75 + 0.25 * np.tanh((d * x).sum(-1, keepdims=True))), 0.0, 1.0)
    return sigma, col


def render_view(K, c2w, n=768, rows=10):
    """the scene through one camera -> (H, W, 4) uint8: straight colours + alpha, as a Blender RGBA image holds them"""
    K, c2w = K.astype(np.float64), c2w.astype(np.float64)
    out = np.zeros((H, W, 4), np.uint8)
    for y0 in range(0, H, rows):
        yy, xx = np.meshgrid(np.arange(y0, min(H, y0 + rows)) + 0.5, np.arange(W) + 0.5, indexing='ij')
        cam = np.stack([(xx - K[0, 2]) / K[0, 0], (yy - K[1, 2]) / K[1, 1], np.ones_like(xx)], -1).reshape(-1, 3)
        d = cam @ c2w[:3, :3].T
        d = d / np.linalg.norm(d, axis=-1, keepdims=True)
        o = np.broadcast_to(c2w[:3, 3], d.shape)
        with np.errstate(divide='ignore', invalid='ignore'):
            t0, t1 = (-1.0 - o) / d, (1.0 - o) / d
        near = np.minimum(t0, t1).max(-1)
        far = np.maximum(t0, t1).min(-1)
        hit = far > np.maximum(near, 0.0)
        near, far = np.where(hit, np.maximum(near, 0.0), 0.0), np.where(hit, far, 0.0)
        z = near[:, None] + (far - near)[:, None] * np.linspace(0.0, 1.0, n)[None]
        x = (o[:, None] + d[:, None] * z[..., None]).reshape(-1, 3)
        s, c = scene_field(x, np.repeat(d, n, axis=0))
        s = s.reshape(-1, n) * hit[:, None]
        delta = ((far - near) / (n - 1))[:, None]
        alpha = 1.0 - np.exp(-s * delta)
        T = np.cumprod(np.concatenate([np.ones((alpha.shape[0], 1)), 1.0 - alpha + 1e-10], 1), 1)[:, :-1]
        wgt = alpha * T
        acc = wgt.sum(1)
        rgb = (wgt[..., None] * c.reshape(-1, n, 3)).sum(1)
        straight = np.where(acc[:, None] > 1e-6, rgb / np.maximum(acc[:, None], 1e-6), 0.0)
        px = np.concatenate([np.clip(straight, 0, 1), np.clip(acc, 0, 1)[:, None]], -1)
        out[y0:y0 + rows] = np.rint(px * 255.0).astype(np.uint8).reshape(-1, W, 4)
    return out


def dataset_tensors(rgba):
    """(N, H, W, 4) uint8 -> img (N, H*W, 3), mask (N, H*W) float32 as NeRF.read_image_list makes them (nerf_dataset.py:107-119: / 255.0)"""
    f = rgba.astype(np.float32) / np.float32(255.0)
    n = rgba.shape[0]
    return np.ascontiguousarray(f[..., :3].reshape(n, -1, 3)), np.ascontiguousarray(f[..., 3].reshape(n, -1))


# ---- the loop's random draws ---------------------------------------------------------------------------------------------------------------
def shuffle_perm(seed, k, n):
    """the k-th torch.randperm(total_samples) of Pipeline.step_ray_sample (trainer/pipeline.py:150) in the run of `seed`"""
    return _rng(seed, 1, k).permutation(n).astype(np.int64)


def bkg_draw(seed, epoch, n_rays):
    """torch.rand_like(img) of Pipeline.fetch_step_bkg_color (trainer/pipeline.py:286) at `epoch`: (n_rays, 3) float32 in [0, 1)"""
    return _rng(seed, 2, epoch).random((n_rays, 3), dtype=np.float32)


def refresh_draws(seed, epoch, n_cells):
    """the two draws of VolumeBound.optimize (volume_bound.py:178-193) at `epoch`: a permutation of the cells, one uniform per coordinate"""
    rng = _rng(seed, 3, epoch)
    return rng.permutation(n_cells).astype(np.int64), rng.random((n_cells, 3), dtype=np.float32)


def table_from_seed(n_rows, n_feat, seed):
    rng = _rng(seed, 4)
    return ((rng.random((n_rows, n_feat), dtype=np.float32) - np.float32(0.5)) * np.float32(2.0 * TABLE_AMP)).astype(np.float32)


# ---- what the fixture keeps of a batch -------------------------------------------------------------------------------------------------------
BATCH_KEYS = ('rays_o', 'rays_d', 'img', 'mask', 'bkg_color')


def batch_summary(batch):
    """{key: (n, ...) array} -> (len(BATCH_KEYS), 2) float64: per key the sum and the sum weighted by (row index + 1) (order sensitive)"""
    out = np.zeros((len(BATCH_KEYS), 2))
    for i, k in enumerate(BATCH_KEYS):
        v = np.asarray(batch[k], np.float64).reshape(len(batch[k]), -1)
        out[i, 0] = v.sum()
        out[i, 1] = (v.sum(1) * (np.arange(v.shape[0]) + 1.0)).sum()
    return out


def psnr(pred, target):
    """img_metric.py:50-56: -10 log10(mean squared error)"""
    mse = float(np.mean((np.asarray(pred, np.float64) - np.asarray(target, np.float64)) ** 2))
    return -10.0 * math.log10(mse)


def white_targets(rgba):
    """held-out targets: blend_bkg_color [1, 1, 1] (the val / eval augmentation of nerf_lego_nerf_ngp.yaml:100-112): img * mask + (1 - mask)"""
    img, mask = dataset_tensors(rgba)
    return (img * mask[..., None] + (np.float32(1.0) - mask[..., None])).astype(np.float32)


# ---- test side -----------------------------------------------------------------------------------------------------------------------------
def golden():
    from conftest import load_golden
    return load_golden('g27_psnr')


class Tape:
    """arcnerf_amd.geometry.volume.set_refresh_tape(Tape(seed)) / trainer.Pipeline(tape=Tape(seed)): the loop draws what the reference run drew"""

    def __init__(self, seed):
        self.seed = seed

    def draws(self, epoch, n_cells, device):
        import torch
        perm, uni = refresh_draws(self.seed, epoch, n_cells)
        return torch.from_numpy(perm).to(device), torch.from_numpy(uni).to(device)

    def shuffle(self, k, n, device):
        import torch
        return torch.from_numpy(shuffle_perm(self.seed, k, n)).to(device)

    def bkg(self, epoch, n_rays, device):
        import torch
        return torch.from_numpy(bkg_draw(self.seed, epoch, n_rays)).to(device)
