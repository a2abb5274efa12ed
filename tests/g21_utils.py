"""Helpers shared by the CPU and GPU tests of golden G21 (tests/golden/make_golden_ngp.py: the reference's own NGP stack run end to
end on CPU with only its two CUDA-only calls replaced by the oracle)."""
import numpy as np

from conftest import load_golden

NETS = {
    # tag -> NgpConfig overrides describing the reference nets of that variant
    'lin': dict(geo_fused_semantics=False, has_bias=True, W_feat=16, add_inf_z=False),
    'nb': dict(geo_fused_semantics=False, has_bias=False, W_feat=15, add_inf_z=True),
    # the `nb` nets expressed as the config's own FUSED nets (feat = the whole 16-wide geometry output, column 0 included): the radiance
    # net's first layer gets a zero column for output 0, which makes the two networks the same function
    'nb_fused': dict(geo_fused_semantics=True, has_bias=False, W_feat=16, add_inf_z=True),
}
GEO = 'fg_model.coarse_geo_net.layers.{}.{}'
RAD = 'fg_model.coarse_radiance_net.layers.{}.{}'


def golden():
    return load_golden('g21_ngp_model')


def table(g):
    """the fixture's table from its seed (make_golden_ngp.py:table_from_seed), checked against the stored checksums"""
    rng = np.random.default_rng(int(g['table_seed']))
    n_rows = int(g['offsets'][-1])
    t = ((rng.random((n_rows, 2), dtype=np.float32) - np.float32(0.5)) * np.float32(2.0 * float(g['table_amp']))).astype(np.float32)
    assert abs(t.astype(np.float64).sum() - float(g['table_sum'])) < 1e-6 and np.array_equal(t[::100003], g['table_probe'])
    return t


def bitfield(g, n_grid=128):
    return np.unpackbits(g['bitfield_packed'], bitorder='little').astype(bool).reshape(n_grid, n_grid, n_grid)


def mask_pts(g, key, n_col):
    return np.unpackbits(g[key], axis=1, bitorder='little')[:, :n_col].astype(bool)


def net_weights(g, variant, net):
    """[(W (out,in), b or None), ...] for geo and rad in the layout of NgpField.export_numpy(); net 'nb_fused' = the zero-column map"""
    src = 'nb' if net == 'nb_fused' else net
    pre = '{}_{}_sd.'.format(variant, src)
    out = {}
    for name, fmt, n in (('geo', GEO, 2), ('rad', RAD, 3)):
        layers = []
        for i in range(n):
            W = g[pre + fmt.format(i, 'weight')]
            b = g[pre + fmt.format(i, 'bias')] if (pre + fmt.format(i, 'bias')) in g.files else None
            layers.append((W, b))
        out[name] = layers
    if net == 'nb_fused':
        W0 = out['rad'][0][0]                                   # (64, 15 + 16), inputs [feat 1..15 | SH]
        out['rad'][0] = (np.concatenate([np.zeros((W0.shape[0], 1), np.float32), W0], 1), None)
    return out


def fill_field(fld, g, variant, net, tbl):
    """copy the fixture's table and weights into an NgpField's flat parameter buffer"""
    import torch
    w = net_weights(g, variant, net)
    fld.view('table').copy_(torch.from_numpy(tbl.reshape(-1)))
    for name in ('geo', 'rad'):
        fld.view(name + '_w').copy_(torch.from_numpy(np.concatenate([W.reshape(-1) for W, _ in w[name]])))
        if fld.cfg.has_bias:
            fld.view(name + '_b').copy_(torch.from_numpy(np.concatenate([b for _, b in w[name]])))


def packed_noise(g, variant, net, counts):
    """the reference's dense draw (valid rays, P' or P'-1) -> one value per packed sample (row-major over (ray, sample)); the draw has
    no column for a ray's sample in the LAST dense column when add_inf_z is off (that sample is dropped by the compositor)"""
    src = 'nb' if net == 'nb_fused' else net
    dense = g['{}_{}_train1_noise'.format(variant, src)]
    rows = np.nonzero(counts > 0)[0]
    assert dense.shape[0] == rows.shape[0]
    out = []
    for k, r in enumerate(rows):
        v = np.zeros(int(counts[r]), np.float32)
        n = min(int(counts[r]), dense.shape[1])
        v[:n] = dense[k, :n]
        out.append(v)
    return np.concatenate(out)


def check_outputs(g, pre, rgb, depth, mask, atol=1e-4, depth_far=10.0, train=True):
    """rgb / depth / mask (R,...) against the reference's FullModel output incl. its defaults for rays without samples"""
    sfx = '_coarse' if train else ''
    for k, v in (('rgb', rgb), ('depth', depth), ('mask', mask)):
        ref = g[pre + k + sfx][0]
        assert np.abs(ref - v).max() < atol, (pre, k, float(np.abs(ref - v).max()))


def check_table_grad(g, pre, grad, rtol=1e-3):
    """grad (n_rows, 2) fp32 against the stored summary of the reference's dense table gradient"""
    offs = g['offsets']
    mx = float(g[pre + 'max'])
    tol = rtol * mx
    assert np.abs(grad[:offs[3]] - g[pre + 'low_levels']).max() <= tol
    rows = np.arange(5, grad.shape[0], 16)
    sub = g[pre + 'rows_mod16']
    assert np.abs(grad[rows] - sub).max() <= tol
    assert np.array_equal(np.abs(grad[rows]).sum(1) > 0, np.abs(sub).sum(1) > 0)       # same rows touched
    assert int((np.abs(grad).sum(1) > 0).sum()) == int(g[pre + 'nnz_rows'])
    L = len(offs) - 1
    la = g[pre + 'level_abs']
    ls = np.stack([grad[offs[l]:offs[l + 1]].astype(np.float64).sum(0) for l in range(L)])
    assert np.abs(ls - g[pre + 'level_sum']).max() <= 1e-4 * la.max()
    labs = np.stack([np.abs(grad[offs[l]:offs[l + 1]]).astype(np.float64).sum(0) for l in range(L)])
    assert np.abs(labs - la).max() <= 1e-4 * la.max()
    sgn = np.random.default_rng(77).integers(0, 2, size=(4,) + grad.shape, dtype=np.int8)
    proj = np.array([((sgn[i].astype(np.float64) * 2 - 1) * grad).sum() for i in range(4)])
    assert np.abs(proj - g[pre + 'proj']).max() <= 1e-4 * la.sum()


def check_net_grads(g, variant, net, tag, got, rtol=1e-3):
    """got: {'geo': [(dW, db)...], 'rad': [...]} against the reference's autograd gradients"""
    src = 'nb' if net == 'nb_fused' else net
    pre = '{}_{}_{}_grad.'.format(variant, src, tag)
    for name, fmt in (('geo', GEO), ('rad', RAD)):
        for i, (dW, db) in enumerate(got[name]):
            ref = g[pre + fmt.format(i, 'weight')]
            if net == 'nb_fused' and name == 'rad' and i == 0:
                dW = dW[:, 1:]                    # the zero column's gradient has no counterpart in the reference net
            assert dW.shape == ref.shape
            assert np.abs(dW - ref).max() <= rtol * np.abs(ref).max() + 1e-7, (name, i, 'weight')
            if db is not None:
                refb = g[pre + fmt.format(i, 'bias')]
                assert np.abs(db - refb).max() <= rtol * np.abs(refb).max() + 1e-7, (name, i, 'bias')


def split_flat_grads(fld, flat):
    """flat gradient buffer (numpy) -> table (n,2) and {'geo': [(dW, db)], 'rad': [...]}"""
    off, n = fld._seg['table']
    tbl = flat[off:off + n].reshape(-1, fld.cfg.n_feat_per_entry)
    nets = {}
    for name, dims in (('geo', fld.geo_dims), ('rad', fld.rad_dims)):
        ow, _ = fld._seg[name + '_w']
        ob, nb = fld._seg.get(name + '_b', (0, 0))
        layers = []
        for i in range(len(dims) - 1):
            k = dims[i] * dims[i + 1]
            dW = flat[ow:ow + k].reshape(dims[i + 1], dims[i])
            ow += k
            db = None
            if nb:
                db = flat[ob:ob + dims[i + 1]]
                ob += dims[i + 1]
            layers.append((dW, db))
        nets[name] = layers
    return tbl, nets
