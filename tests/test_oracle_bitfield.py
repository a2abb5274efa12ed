"""CPU checks of the oracle's `_bitfield_func` restatement (K5-K10).

The reference holds these kernels as CUDA only (arcnerf/ops/src/bitfield_func/bitfield_func_kernel.cu) and ships no test
vectors for them, so parity is UNPINNED: what is checked here is the oracle against the published definition of the Morton
code, against numpy restatements of the per-element rules, and — for the sampler K5 — against the oracle's K3, which walks the
same loop over the same occupancy in the x-major bool layout."""
import numpy as np
import pytest


def _morton_perm(n_grid, oracle):
    """flat x-major index -> Morton index, for every cell"""
    ax = np.arange(n_grid, dtype=np.uint32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing='ij')
    return oracle.morton3d(np.stack([X.ravel(), Y.ravel(), Z.ravel()], -1))


def to_morton_bits(bf_bool, oracle):
    """(n,n,n) bool occupancy -> packed Morton bitfield (n^3/8) uint8"""
    n = bf_bool.shape[0]
    m = _morton_perm(n, oracle)
    cells = np.zeros(n ** 3, np.uint8)
    cells[m] = bf_bool.reshape(-1)
    return np.packbits(cells, bitorder='little')


def test_morton_known_answers_and_round_trip(oracle):
    xyz = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [3, 0, 0], [1, 1, 1], [2, 3, 5], [1023, 1023, 1023]], np.uint32)
    want = []
    for x, y, z in xyz:   # bit k of x -> bit 3k, of y -> 3k+1, of z -> 3k+2
        v = 0
        for k in range(10):
            v |= ((int(x) >> k) & 1) << (3 * k) | ((int(y) >> k) & 1) << (3 * k + 1) | ((int(z) >> k) & 1) << (3 * k + 2)
        want.append(v)
    got = oracle.morton3d(xyz)
    assert got.tolist() == want and got[1] == 1 and got[2] == 2 and got[3] == 4 and got[4] == 9 and got[-1] == (1 << 30) - 1
    rng = np.random.default_rng(0)
    pts = rng.integers(0, 1024, size=(5000, 3)).astype(np.uint32)
    assert np.array_equal(oracle.morton3d_invert(oracle.morton3d(pts)), pts)
    perm = _morton_perm(16, oracle)
    assert np.array_equal(np.sort(perm), np.arange(16 ** 3))   # a bijection on the grid


def _rays(rng, R):
    o = rng.normal(size=(R, 3)).astype(np.float32)
    o = o / np.linalg.norm(o, axis=-1, keepdims=True) * 2.5
    tgt = (rng.random((R, 3)).astype(np.float32) - 0.5) * 1.6
    d = tgt - o
    return o, (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)


def test_k5_equals_k3_on_the_same_occupancy(oracle):
    rng = np.random.default_rng(5)
    n_grid, n_pts, R = 16, 256, 400
    bf = rng.random((n_grid,) * 3) < 0.15
    o, d = _rays(rng, R)
    aabb23 = np.array([[-1, -1, -1], [1, 1, 1]], np.float32)
    near, far, _, _ = oracle.aabb_intersection(o, d, aabb23[None])
    dt = np.float32(2 * np.sqrt(3.0) / n_pts)
    h = oracle.Pcg32(9121)
    z3, m3, c3, tr3 = oracle.sparse_volume_sampling(o, d, near, far, n_pts, dt, aabb23, n_grid, bf, 0.2, h.state, h.inc, with_trace=True)
    z5, m5, c5, tr5 = oracle.sparse_volume_sampling_bit(o, d, near, far, n_pts, dt, aabb23, n_grid, to_morton_bits(bf, oracle), 0.2,
                                                        h.state, h.inc, with_trace=True)
    # the two differ only for a point exactly on the max faces (K3 rejects index n, K5 clamps it): none with these rays
    assert np.array_equal(c3, c5) and np.array_equal(m3, m5)
    assert np.array_equal(z3.view(np.uint32), z5.view(np.uint32))
    assert c5.sum() > 1000
    # traced cell: K5 reports the Morton index of the x-major cell K3 reports
    perm = _morton_perm(n_grid, oracle)
    rr, jj = np.nonzero(m5)
    assert np.array_equal(perm[tr3[rr, jj]], tr5[rr, jj].astype(np.uint32))


def test_k5_clamps_points_on_the_max_face(oracle):
    # a ray that starts exactly on the +x face and runs along it: voxel coordinate n is clamped to n-1 and sampled
    n_grid, n_pts = 4, 8
    bits = np.full(n_grid ** 3 // 8, 255, np.uint8)
    o = np.array([[1.0, -0.9, 0.1]], np.float32)
    d = np.array([[0.0, 1.0, 0.0]], np.float32)
    aabb23 = np.array([[-1, -1, -1], [1, 1, 1]], np.float32)
    h = oracle.Pcg32(9121)
    z, m, c = oracle.sparse_volume_sampling_bit(o, d, np.zeros((1, 1), np.float32), np.full((1, 1), 1.5, np.float32), n_pts, 0.25,
                                                aabb23, n_grid, bits, 0.0, h.state, h.inc)
    assert c[0] >= 6 and m[0, :c[0]].all()
    z3, m3, c3 = oracle.sparse_volume_sampling(o, d, np.zeros((1, 1), np.float32), np.full((1, 1), 1.5, np.float32), n_pts, 0.25,
                                               aabb23, n_grid, np.ones((n_grid,) * 3, bool), 0.0, h.state, h.inc)
    assert c3[0] == 0   # K3 treats coordinate n as outside


def test_k6_grid_samples(oracle):
    n_grid, n = 16, 3000
    rng = np.random.default_rng(1)
    grid = (rng.random(n_grid ** 3).astype(np.float32) - 0.5)
    h = oracle.Pcg32(9121)
    for step, thresh in ((0, -0.01), (3, 0.2), (70000, 0.49)):
        pos, idx = oracle.generate_grid_samples(grid, n, step, n_grid, thresh, h.state, h.inc)
        assert idx.min() >= 0 and idx.max() < n_grid ** 3
        # uint32 wrap-around probe sequence, restated with python integers
        for i in (0, 1, 17, n - 1):
            want = None
            for j in range(10):
                want = ((((i + step * n) & 0xffffffff) * 56924617 + j * 19349663 + 96925573) & 0xffffffff) % (n_grid ** 3)
                if grid[want] > thresh:
                    break
            assert idx[i] == want
        # the position lies inside its cell, and the jitter of sample i is draws 4i..4i+2 of the host stream
        cell = oracle.morton3d_invert(idx.astype(np.uint32))
        f = pos * n_grid - cell
        assert (f >= 0).all() and (f < 1.0 + 1e-5).all() and (pos >= 0).all() and (pos < 1).all()
        h2 = oracle.Pcg32(9121)
        h2.advance(4 * 17)
        u = h2.next_float(3)
        assert np.allclose((cell[17] + u) / n_grid, pos[17], rtol=0, atol=1e-7)
        if thresh > 0:   # up to 10 probes for a dense cell: hit rate 1 - (1 - p)^10 for a base rate p
            p = (grid > thresh).mean()
            assert abs((grid[idx] > thresh).mean() - (1 - (1 - p) ** 10)) < 0.03


def test_k7_k8_k9_k10_rules(oracle):
    n_grid = 8
    n = n_grid ** 3
    rng = np.random.default_rng(2)
    idx = rng.integers(0, n, size=4000).astype(np.int32)
    den = rng.random(4000).astype(np.float32)
    tmp = np.zeros(n, np.float32)
    oracle.splat_grid_samples(den, idx, tmp)
    want = np.zeros(n, np.float32)
    np.maximum.at(want, idx, den)
    assert np.array_equal(tmp, want)
    grid = (rng.random(n).astype(np.float32) - 0.3).astype(np.float32)   # some negative cells: frozen by the EMA
    g0 = grid.copy()
    oracle.ema_grid_samples_nerf(tmp, grid, 0.95)
    want = np.where(g0 < 0, g0, np.maximum(g0 * np.float32(0.95), tmp)).astype(np.float32)
    assert np.array_equal(grid, want)
    mean = float(np.clip(grid, 0, None).mean())
    for thres in (0.01, 10.0):
        bf = oracle.update_bitfield(grid, mean, thres, n_grid)
        assert np.array_equal(np.unpackbits(bf, bitorder='little').astype(bool), grid > min(thres, mean))
    # K10 as written in the reference: 8 per non-zero byte
    bf = np.zeros(n // 8, np.uint8)
    bf[[1, 5, 9]] = [1, 255, 16]
    assert oracle.count_bitfield(bf, n_grid) == 24.0
    assert oracle.count_bitfield(np.zeros(n // 8, np.uint8), n_grid) == 0.0


# ---- `_multivol_func` (K11, K12, cascaded K9): CUDA only in the reference, parity unpinned ----------------------------
def _level_of(pts, half):
    """volume_func.h:201-226 restated with numpy: exponent of the largest |coordinate| / half side (frexp), floored at 0"""
    a = np.abs(pts.astype(np.float32)) * np.float32(1.0 / half)
    return np.maximum(np.frexp(a)[1].max(-1), 0)


def test_k11_degenerates_to_k5(oracle):
    """one inclusive level, zero cone angle, min_step = max_step: the cascade marcher is the single-volume marcher"""
    rng = np.random.default_rng(6)
    n_grid, n_pts, R = 16, 128, 300
    bits = to_morton_bits(rng.random((n_grid,) * 3) < 0.2, oracle)
    o, d = _rays(rng, R)
    aabb23 = np.array([[-1, -1, -1], [1, 1, 1]], np.float32)
    near, far, _, _ = oracle.aabb_intersection(o, d, aabb23[None])
    dt = np.float32(2 * np.sqrt(3.0) / n_pts)
    h = oracle.Pcg32(9121)
    z5, m5, c5 = oracle.sparse_volume_sampling_bit(o, d, near, far, n_pts, dt, aabb23, n_grid, bits, 0.1, h.state, h.inc)
    z11, m11, c11 = oracle.sparse_sampling_in_multivol_bitfield(o, d, near, far, n_pts, 0.0, dt, dt, aabb23, aabb23, n_grid, 1, bits,
                                                               0.1, True, h.state, h.inc)
    assert c5.sum() > 1000 and np.array_equal(c5, c11) and np.array_equal(m5, m11)
    assert np.array_equal(z5.view(np.uint32), z11.view(np.uint32))


@pytest.mark.parametrize('inclusive', [True, False])
def test_k11_invariants(oracle, inclusive):
    rng = np.random.default_rng(8)
    n_grid, n_cascade, n_pts, R = 16, 4, 192, 400
    levels = n_cascade if inclusive else n_cascade - 1
    cells = rng.random(levels * n_grid ** 3) < 0.3
    bits = np.packbits(cells, bitorder='little')
    o = ((rng.random((R, 3)) - 0.5) * 3.0).astype(np.float32)
    o[: R // 2] *= 0.2   # half of the cameras inside the inner volume
    d = rng.normal(size=(R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    inner = np.array([[-0.5, -0.5, -0.5], [0.5, 0.5, 0.5]], np.float32)
    outer = inner * 2 ** (n_cascade - 1)
    near, far, _, hit = oracle.aabb_intersection_torch(o, d, np.stack([outer[0], outer[1]], -1)[None])
    cone, min_step, max_step = 1.0 / 16, np.float32(np.sqrt(3.0) / n_pts), np.float32(np.sqrt(3.0) * 8 / n_grid)
    h = oracle.Pcg32(9121)
    z, m, c = oracle.sparse_sampling_in_multivol_bitfield(o, d, near, far, n_pts, cone, min_step, max_step, inner, outer, n_grid,
                                                          n_cascade, bits, 0.05, inclusive, h.state, h.inc)
    assert (m.sum(1) == c).all() and c.sum() > 2000
    perm_cache = {}
    for r in range(R):
        k = c[r]
        assert m[r, :k].all() and not m[r, k:].any()
        if k == 0:
            assert (z[r] == 0).all()
            continue
        zz = z[r, :k]
        assert (z[r, k:] == zz[-1]).all() and (np.diff(zz) > 0).all()
        assert zz[0] >= max(near[r, 0], 0.05) and zz[-1] <= far[r, 0]
        # consecutive samples are at least one cone step apart: dt = clamp(t * cone, min_step, max_step)
        dts = np.clip(zz[:-1] * np.float32(cone), min_step, max_step)
        assert (np.diff(zz) >= dts * (1 - 1e-6)).all()
        pts = o[r] + d[r] * zz[:, None]
        lvl = np.minimum(_level_of(pts, 0.5), n_cascade - 1)
        if not inclusive:
            assert (lvl >= 1).all()   # nothing is sampled inside the excluded inner volume ...
            # ... and nothing is kept from before the ray's last visit to it
            t_all = np.arange(0, far[r, 0], float(min_step) / 2, dtype=np.float32)
            inside = (np.abs(o[r] + d[r] * t_all[:, None]).max(-1) < 0.5) & (t_all > max(near[r, 0], 0.05) + max_step)
            if inside.any():
                assert zz[0] > t_all[inside].max() - max_step
        # every sample sits in an occupied cell of its level's grid
        q = pts / (2.0 ** lvl)[:, None]
        ijk = np.clip(((q + 0.5) * n_grid).astype(np.int64), 0, n_grid - 1).astype(np.uint32)
        mort = oracle.morton3d(ijk)
        slot = lvl if inclusive else lvl - 1
        assert cells[slot * n_grid ** 3 + mort].all()


@pytest.mark.parametrize('inclusive', [True, False])
def test_k12_and_cascaded_k9(oracle, inclusive):
    n_grid, n_cascade, n = 8, 4, 5000
    levels = n_cascade if inclusive else n_cascade - 1
    rng = np.random.default_rng(9)
    grid = (rng.random(levels * n_grid ** 3).astype(np.float32) - 0.5)
    inner = np.array([[-0.5, -1.0, 0.0], [0.5, 1.0, 3.0]], np.float32)   # anisotropic, off-centre box
    h = oracle.Pcg32(9121)
    pos, idx = oracle.generate_grid_samples_multivol(grid, n, inner, 3, n_cascade, n_grid, 0.2, inclusive, h.state, h.inc)
    slot = idx // n_grid ** 3
    assert slot.min() >= 0 and slot.max() == levels - 1 and len(np.unique(slot)) == levels
    level = slot if inclusive else slot + 1
    # the point lies in its Morton cell of the level's volume (inner box scaled 2^level about its centre)
    center, length = (inner[0] + inner[1]) / 2, inner[1] - inner[0]
    u = (pos - center) / (length * (2.0 ** level)[:, None]) + 0.5
    cell = oracle.morton3d_invert((idx % n_grid ** 3).astype(np.uint32))
    f = u * n_grid - cell
    assert (f > -1e-4).all() and (f < 1 + 1e-4).all()
    # level draw of sample i = first draws of stream position 4i (redrawn while 0 when the inner volume is excluded)
    for i in (0, 5, 77):
        h2 = oracle.Pcg32(9121)
        h2.advance(4 * i)
        lv = 0
        while True:
            lv = int(np.float32(h2.next_float(1)[0]) * np.float32(n_cascade)) % n_cascade
            if inclusive or lv != 0:
                break
        assert lv == level[i]
    p = (grid > 0.2).mean()
    assert abs((grid[idx] > 0.2).mean() - (1 - (1 - p) ** 10)) < 0.03
    mean = float(np.clip(grid, 0, None).mean())
    bf = oracle.update_bitfield_multivol(grid, mean, 0.01, n_grid, n_cascade, inclusive)
    assert bf.shape[0] == levels * n_grid ** 3 // 8
    assert np.array_equal(np.unpackbits(bf, bitorder='little').astype(bool), grid > min(0.01, mean))
