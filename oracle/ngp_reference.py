"""ORACLE — TEST INFRASTRUCTURE ONLY: the NGP step restated with the CPU oracle (dense reference view).

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg to CHECK / TIME against; never imported by the
product package.  Follows the reference call stack of SURVEY.md §3.1: K2 bounds -> K3 marching -> boolean-mask compaction
(fg_model.py:283-292) -> hash grid -> geo net -> TruncExp -> [feat, SH] -> radiance net -> dense padded scatter
(fg_model.py:305-316) -> ray_marching, and the matching backward.
"""
import numpy as np


def oracle_step(orc, fld, cfg, P, o, d, bkg, bf, rng_state, rng_inc, g_rgb=None, huber_target=None, noise=None, torch_bounds=False):
    """Forward (and backward when g_rgb or huber_target is given) of one NGP step on the CPU.

    fld: an arcnerf_amd.pipeline.NgpField (only its metadata: level table, bounds, segment layout); P = fld.export_numpy().
    Returns dict(rgb, depth, mask (R,), valid (R,), n_samples, counts, zvals, mask_pts [, grads (flat), loss]).
    Pinned to a run of the reference itself by tests/test_oracle_ngp_golden.py (golden G21).
    """
    aabb23 = np.array([fld.min_xyz, fld.max_xyz], np.float32)
    if torch_bounds:  # the reference's own torch AABB code (geometry/ray.py:295-339) instead of K2
        near, far, _, _ = orc.aabb_intersection_torch(o, d, np.ascontiguousarray(aabb23.T)[None])
    else:
        near, far, _, _ = orc.aabb_intersection(o, d, aabb23[None])
    z, m, cnt = orc.sparse_volume_sampling(o, d, near, far, cfg.n_sample, np.float32(cfg.dt), aabb23, cfg.n_grid, bf,
                                           cfg.near_distance, rng_state, rng_inc)
    R = o.shape[0]
    out = {'n_samples': int(cnt.sum()), 'counts': cnt}
    Pd = max(2, int(cnt.max()))
    z, m = np.ascontiguousarray(z[:, :Pd]), np.ascontiguousarray(m[:, :Pd])
    valid = cnt > 0
    out['valid'] = valid
    out['zvals'], out['mask_pts'] = z, m
    rr, jj = np.nonzero(m)
    pts = (o[rr] + z[rr, jj][:, None] * d[rr]).astype(np.float32)
    res = np.array(fld.resolutions, np.int32)
    offs = np.array(fld.offsets, np.int64)
    mn, mx = np.array(fld.min_xyz, np.float32), np.array(fld.max_xyz, np.float32)
    feat = orc.hashgrid_fwd(pts, P['table'], res, offs, mn, mx)
    hs, pres = [feat], []
    for i, (W, b_) in enumerate(P['geo']):
        y, pre = orc.linear_fwd(hs[-1], W, b_, 'relu' if i < len(P['geo']) - 1 else None, want_pre=True)
        hs.append(y)
        pres.append(pre)
    geo_out = hs[-1]
    sigma_s = orc.act_fwd(geo_out[:, 0], cfg.sigma_act)
    dn = d[rr] / (np.linalg.norm(d[rr], axis=-1, keepdims=True) + 1e-8)
    sh = orc.sh_fwd(dn.astype(np.float32), cfg.sh_degree, False)
    fpart = geo_out[:, fld.feat_off:fld.feat_off + cfg.W_feat]
    rin = np.concatenate([fpart, sh] if cfg.rad_mode == 'fv' else [sh, fpart], 1).astype(np.float32)
    rs, rpres = [rin], []
    for i, (W, b_) in enumerate(P['rad']):
        y, pre = orc.linear_fwd(rs[-1], W, b_, 'relu' if i < len(P['rad']) - 1 else 'sigmoid', want_pre=True)
        rs.append(y)
        rpres.append(pre)
    rgb_s = rs[-1]
    # dense padded view (fg_model.py:305-316): tail columns repeat the ray's last valid sample
    sg = np.zeros((R, Pd), np.float32)
    rd = np.zeros((R, Pd, 3), np.float32)
    last = np.cumsum(cnt) - 1
    vi = np.nonzero(valid)[0]
    sg[vi, :] = sigma_s[last[vi]][:, None]
    rd[vi, :, :] = rgb_s[last[vi]][:, None, :]
    sg[m], rd[m] = sigma_s, rgb_s
    ns = None
    if noise is not None:  # packed per-sample noise -> dense (columns of the padded tail get the last sample's noise)
        ns_full = np.zeros((R, Pd), np.float32)
        ns_full[vi, :] = noise[last[vi]][:, None]
        ns_full[m] = noise[:rr.shape[0]]
        ns = np.ascontiguousarray(ns_full[valid][:, :Pd if cfg.add_inf_z else Pd - 1])
    bk = None if bkg is None else np.ascontiguousarray(bkg[valid])
    ref = orc.ray_marching_fwd(sg[valid], rd[valid], z[valid], add_inf_z=cfg.add_inf_z, white_bkg=cfg.white_bkg,
                               bkg_color=bk, noise=ns)
    rgb = np.zeros((R, 3), np.float32) if bkg is None else bkg.astype(np.float32).copy()
    depth, mask = np.zeros(R, np.float32), np.zeros(R, np.float32)
    rgb[valid], depth[valid], mask[valid] = ref['rgb'], ref['depth'], ref['mask']
    out.update(rgb=rgb, depth=depth, mask=mask)
    if huber_target is not None:
        diff = rgb - huber_target
        ad = np.abs(diff)
        out['loss'] = float(np.where(ad < cfg.huber_delta, 0.5 / cfg.huber_delta * ad * ad, ad - 0.5 * cfg.huber_delta).mean()
                            * cfg.loss_weight)
        g_rgb = (np.where(ad < cfg.huber_delta, diff / cfg.huber_delta, np.sign(diff)) * (cfg.loss_weight / diff.size)).astype(np.float32)
    if g_rgb is None:
        return out
    dsg, drd = orc.ray_marching_bwd(sg[valid], rd[valid], z[valid], g_rgb[valid], None, None, add_inf_z=cfg.add_inf_z,
                                    white_bkg=cfg.white_bkg, bkg_color=bk, noise=ns)
    # gradients of the padded columns flow to the last valid sample (they are a gather of it)
    DS = np.zeros((R, Pd), np.float32)
    DR = np.zeros((R, Pd, 3), np.float32)
    DS[valid], DR[valid] = dsg, drd
    d_sigma = DS[m].copy()
    d_rgb_s = DR[m].copy()
    pad_s = (DS * ~m).sum(1)
    pad_r = (DR * (~m)[..., None]).sum(1)
    d_sigma[last[vi]] += pad_s[vi]
    d_rgb_s[last[vi]] += pad_r[vi]
    dy = d_rgb_s
    g_rad, b_rad, b_geo = [], [], []
    for i in reversed(range(len(P['rad']))):
        dy, dW, db = orc.linear_bwd(rs[i], P['rad'][i][0], rpres[i], rs[i + 1], dy, 'relu' if i < len(P['rad']) - 1 else 'sigmoid',
                                    has_bias=P['rad'][i][1] is not None)
        g_rad.insert(0, dW)
        b_rad.insert(0, db)
    d_rin = dy
    d_geo_out = np.zeros_like(geo_out)
    sl = slice(0, cfg.W_feat) if cfg.rad_mode == 'fv' else slice(cfg.sh_degree ** 2, None)
    d_geo_out[:, fld.feat_off:fld.feat_off + cfg.W_feat] = d_rin[:, sl]
    d_geo_out[:, 0] += orc.act_bwd(geo_out[:, 0], sigma_s, d_sigma, cfg.sigma_act)
    dy = d_geo_out
    g_geo = []
    for i in reversed(range(len(P['geo']))):
        dy, dW, db = orc.linear_bwd(hs[i], P['geo'][i][0], pres[i], hs[i + 1], dy, 'relu' if i < len(P['geo']) - 1 else None,
                                    has_bias=P['geo'][i][1] is not None)
        g_geo.insert(0, dW)
        b_geo.insert(0, db)
    d_table = orc.hashgrid_bwd(pts, P['table'], dy, res, offs, mn, mx)
    grads = np.zeros(fld.n_params, np.float32)
    for name, arr in (('table', d_table.reshape(-1)), ('geo_w', np.concatenate([g.reshape(-1) for g in g_geo])),
                      ('rad_w', np.concatenate([g.reshape(-1) for g in g_rad]))):
        off, n = fld._seg[name]
        grads[off:off + n] = arr
    for name, bs in (('geo_b', b_geo), ('rad_b', b_rad)):
        off, n = fld._seg.get(name, (0, 0))
        if n and all(b is not None for b in bs):
            grads[off:off + n] = np.concatenate(bs)
    out['grads'] = grads
    return out


def oracle_train_step(orc, fld, cfg, P, o, d, bf, rng_state, rng_inc):
    """fwd + Huber loss + bwd for timing (bench.py cpu_baseline).  Returns the number of valid samples processed."""
    rng = np.random.default_rng(0)
    R = o.shape[0]
    tgt = rng.random((R, 3)).astype(np.float32)
    bkg = rng.random((R, 3)).astype(np.float32)
    res = oracle_step(orc, fld, cfg, P, o, d, bkg, bf, rng_state, rng_inc, huber_target=tgt)
    return res['n_samples']


def oracle_train_step_sharded(orc, fld, cfg, P, o, d, bf, rng_state, rng_inc, shards, threads_per_shard=1):
    """oracle_train_step over `shards` contiguous ray shards run CONCURRENTLY (Python threads: the C kernels and numpy's heavy operations
    release the interpreter lock), each with `threads_per_shard` OpenMP threads, then the shards' flat gradients are summed - the same
    work as one call (rays are independent; a shard's sampler stream starts at its first ray's position, pcg32 advanced 8 per ray like
    the kernel does).  The single-call form keeps its numpy glue (boolean-mask compaction, dense padded scatter) on ONE core and stopped
    scaling at ~3x (round 3); this is the all-core form bench.py's cpu_baseline times.  Returns (valid samples, summed flat gradient)."""
    from concurrent.futures import ThreadPoolExecutor
    R = o.shape[0]
    rng = np.random.default_rng(0)
    tgt = rng.random((R, 3)).astype(np.float32)
    bkg = rng.random((R, 3)).astype(np.float32)
    bounds = [R * k // shards for k in range(shards + 1)]

    def work(k):
        lo, hi = bounds[k], bounds[k + 1]
        if hi <= lo:
            return 0, None
        orc.set_num_threads(threads_per_shard)          # (an OpenMP setting of the calling thread)
        st = orc.Pcg32(9121)
        st.si[0] = np.uint64(rng_state)
        st.advance(8 * lo)
        res = oracle_step(orc, fld, cfg, P, o[lo:hi], d[lo:hi], bkg[lo:hi], bf, st.state, rng_inc, huber_target=tgt[lo:hi])
        # (the loss is a mean over the shard's rays: weight the gradient back to the whole batch's mean)
        return res['n_samples'], res['grads'] * np.float32((hi - lo) / R)
    with ThreadPoolExecutor(max_workers=shards) as ex:
        outs = list(ex.map(work, range(shards)))
    total = sum(n for n, _ in outs)
    grads = None
    for _, g in outs:
        if g is not None:
            grads = g if grads is None else grads + g
    return total, grads


def compare(ref, rgb, depth, mask, grads, fld):
    """max abs errors of GPU results against an oracle_step() result"""
    valid = ref['valid']
    out = {'rgb': float(np.abs(ref['rgb'] - rgb).max()), 'depth': float(np.abs(ref['depth'][valid] - depth[valid]).max()),
           'mask': float(np.abs(ref['mask'] - mask).max())}
    if grads is not None and 'grads' in ref:
        rg = ref['grads']
        out['grad_rel'] = float(np.abs(rg - grads).max() / (np.abs(rg).max() + 1e-12))
        for name in ('table', 'geo_w', 'rad_w'):
            off, n = fld._seg[name]
            out['grad_rel_' + name] = float(np.abs(rg[off:off + n] - grads[off:off + n]).max() / (np.abs(rg[off:off + n]).max() + 1e-12))
    return out


def ngp_smoke_check(device='cuda:0', n_rays=512):
    """One small fwd+bwd of the hot path on the GPU checked against the CPU oracle (used by __graft_entry__.smoke())."""
    import torch
    from oracle import oracle as orc
    from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays
    torch.cuda.set_device(device)
    cfg = NgpConfig(n_levels=8, hashmap_size=14, max_res=256, n_grid=32, n_sample=256, noise_std=0.0)
    fld = NgpField(cfg, device=device, seed=1)
    fld.view('table').mul_(2000.0)  # make the features matter
    pipe = NgpPipeline(fld, max_rays=n_rays, max_samples=n_rays * 64, packed_bits=True)
    bf = synthetic_bitfield(cfg.n_grid, 0.08, seed=2)
    pipe.set_bitfield(torch.from_numpy(bf))
    o, d = synthetic_rays(n_rays, seed=3, device=device)
    bkg = torch.rand(n_rays, 3, device=device)
    state, inc = pipe.rng.state, pipe.rng.inc
    rgb, depth, mask = pipe.forward(o, d, bkg, train=True)
    g_rgb = torch.randn(n_rays, 3, device=device)
    pipe.backward(o, d, g_rgb.contiguous())
    torch.cuda.synchronize()
    ref = oracle_step(orc, fld, cfg, fld.export_numpy(), o.cpu().numpy(), d.cpu().numpy(), bkg.cpu().numpy(), bf, state, inc,
                      g_rgb=g_rgb.cpu().numpy())
    errs = compare(ref, rgb.cpu().numpy(), depth.cpu().numpy(), mask.cpu().numpy(), fld.grads.cpu().numpy(), fld)
    errs['n_samples'] = (ref['n_samples'], int(pipe.n_dev.item()))
    assert ref['n_samples'] == int(pipe.n_dev.item()), errs
    assert errs['rgb'] < 1e-4 and errs['depth'] < 1e-4 and errs['mask'] < 1e-4, errs
    assert errs['grad_rel'] < 1e-3, errs
    return errs
