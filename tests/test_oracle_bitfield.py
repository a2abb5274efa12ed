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
