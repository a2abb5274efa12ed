"""Big weight matrices of the full-width model fixtures (G22-G24) are not stored: each matrix is regenerated from a seed and two
per-column vectors, W = col_mean[None, :] + col_std[None, :] * Z with Z ~ N(0,1) from numpy's PCG64 (platform stable).  The column
statistics are those of the reference's own initialisation (so geometric-init structure - zero columns for the embedded inputs, the
sqrt(pi)/sqrt(n) last layer - survives), taken by tests/golden/make_golden_fullwidth.py, which then loads W into the reference model
before running it.  Gradients of those matrices are stored as a summary: a few full rows, every 16th row, sums and projections.

Shared by the generator (build container) and the tests; imports numpy only."""
import zlib

import numpy as np

BIG = 4096   # matrices with more elements than this are seeded; everything else is stored verbatim


def _seed(base, name):
    return (int(base) * 1000003 + zlib.crc32(name.encode())) % (1 << 63)


def make_weight(name, shape, base_seed, col_mean, col_std):
    z = np.random.default_rng(_seed(base_seed, name)).standard_normal(shape, dtype=np.float32)
    return (col_mean[None, :].astype(np.float32) + col_std[None, :].astype(np.float32) * z).astype(np.float32)


def grad_summary(g):
    g = np.asarray(g, np.float32)
    g2 = g.reshape(g.shape[0], -1)
    sgn = np.random.default_rng(99).integers(0, 2, size=(3,) + g2.shape, dtype=np.int8).astype(np.float64) * 2 - 1
    return {'head': g2[:4].copy(), 'mod16': g2[5::16].copy(), 'sum': np.array(g2.astype(np.float64).sum()),
            'abs': np.array(np.abs(g2).astype(np.float64).sum()), 'max': np.array(np.abs(g2).max()),
            'proj': np.array([(sgn[i] * g2).sum() for i in range(3)])}


def check_grad(summary, g, rtol=1e-3, name=''):
    g2 = np.asarray(g, np.float32).reshape(g.shape[0], -1)
    tol = rtol * float(summary['max']) + 1e-7
    assert np.abs(g2[:4] - summary['head']).max() <= tol, name
    assert np.abs(g2[5::16] - summary['mod16']).max() <= tol, name
    assert abs(float(np.abs(g2).max()) - float(summary['max'])) <= tol, name
    assert abs(g2.astype(np.float64).sum() - float(summary['sum'])) <= rtol * float(summary['abs']) + 1e-7, name
    assert abs(np.abs(g2).astype(np.float64).sum() - float(summary['abs'])) <= rtol * float(summary['abs']) + 1e-7, name
    sgn = np.random.default_rng(99).integers(0, 2, size=(3,) + g2.shape, dtype=np.int8).astype(np.float64) * 2 - 1
    proj = np.array([(sgn[i] * g2).sum() for i in range(3)])
    assert np.abs(proj - summary['proj']).max() <= rtol * float(summary['abs']) + 1e-7, name


def state_dict_from_fixture(g, prefix='sd.'):
    """{name: numpy array}: verbatim entries + the regenerated seeded matrices"""
    sd = {}
    seed = int(g['weight_seed'])
    for k in g.files:
        if k.startswith(prefix):
            sd[k[len(prefix):]] = g[k]
        elif k.startswith('seeded_mean.'):
            name = k[len('seeded_mean.'):]
            mean, std = g[k], g['seeded_std.' + name]
            shape = tuple(int(v) for v in g['seeded_shape.' + name])
            sd[name] = make_weight(name, shape, seed, mean, std)
    for k in g.files:
        if k.startswith('rowpatch.'):     # leading rows edited after seeding (stored verbatim)
            patch = g[k]
            sd[k[len('rowpatch.'):]][:patch.shape[0]] = patch
    return sd
