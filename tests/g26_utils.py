"""Inputs of golden G26 (tests/golden/make_golden_trajectory.py: the reference's training LOOP run for 48 steps) - shared by the generator
(build container, imports the reference) and by the CPU / GPU tests.  Everything here is numpy PCG64 + IEEE double arithmetic rounded
once to float32: the same bits on every box, pinned by checksums stored in the fixture.  Data generators only - no reference code.
"""
import numpy as np

N_GRID = 32
N_SAMPLE = 1024
EPOCH_OPTIM, EPOCH_WARMUP, UPDATE_EPOCH = 4, 8, 4
LOG_MAX_ALLOWANCE = 15
N_RAYS0, N_RAYS_MAX = 256, 1024
# two jobs.  'a': a fresh start, epochs 0..19, the table at HashGridEmbedder's own init U(-1e-4, 1e-4).  'b': a job started with
# progress.start_epoch = 496 from a model-only checkpoint (a rough table, U(-0.1, 0.1), density row x 3 as in G21): EMA.set_n_step(496)
# while Adam starts at step 1, every refresh past the warm-up, the `epoch > 500` rule of the dynamic batch size applies
LEGS = {'a': dict(epochs=list(range(0, 20)), table_seed=2626, table_amp=1e-4, sigma_row_scale=1.0),
        'b': dict(epochs=list(range(496, 516)), table_seed=2627, table_amp=0.1, sigma_row_scale=3.0)}
SUMMARY_STEPS = (1, 4, 8, 12, 20)                                # parameter summaries after this many steps of a leg
NEAR_BAND = 1e-4                                                 # cells with |opacity - thres| <= NEAR_BAND * thres: two fp32 evaluations of the refresh may decide them differently


def table_from_seed(n_rows, n_feat, seed, amp):
    rng = np.random.default_rng(seed)
    return ((rng.random((n_rows, n_feat), dtype=np.float32) - np.float32(0.5)) * np.float32(2.0 * amp)).astype(np.float32)


def step_inputs(k, n_rays):
    """rays, random background colours and ground-truth colours of training step k (= the epoch number): cameras uniform on the
    sphere of radius 3 / 1.05 looking at a point near the centre, target = an analytic scene (a shaded ball of radius 0.45 in front of
    the per-ray background colour)."""
    rng = np.random.default_rng(26000 + k)
    c = rng.standard_normal((N_RAYS_MAX, 3))
    c = c / np.linalg.norm(c, axis=-1, keepdims=True) * (3.0 / 1.05)
    p = rng.uniform(-0.7, 0.7, (N_RAYS_MAX, 3))
    d = p - c
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    d[7::16] = -d[7::16]                                          # one ray in 16 looks away and misses the volume
    bkg = rng.random((N_RAYS_MAX, 3))
    o32, d32, bkg32 = c.astype(np.float32), d.astype(np.float32), bkg.astype(np.float32)
    # ball |x| = 0.45: first hit of the (float32-rounded) ray in double arithmetic
    o, dd = o32.astype(np.float64), d32.astype(np.float64)
    b = (o * dd).sum(-1)
    disc = b * b - (dd * dd).sum(-1) * ((o * o).sum(-1) - 0.45 ** 2)
    t = (-b - np.sqrt(np.where(disc > 0, disc, 0.0))) / (dd * dd).sum(-1)
    hit = (disc > 0) & (t > 0)
    nrm = (o + t[:, None] * dd) / 0.45
    shade = 0.5 + 0.5 * nrm
    img = np.where(hit[:, None], shade, bkg32.astype(np.float64)).astype(np.float32)
    return {'rays_o': o32[:n_rays].copy(), 'rays_d': d32[:n_rays].copy(), 'bkg_color': bkg32[:n_rays].copy(), 'img': img[:n_rays].copy()}


def refresh_draws(epoch, n_cells):
    """the two random draws of VolumeBound.optimize (volume_bound.py:178-193) at `epoch`: a permutation of the cells (its first n / 4
    entries are the uniformly chosen cells) and one uniform per cell coordinate for the jitter (the reference draws as many rows as it
    has cells to evaluate: the first rows of this array)"""
    rng = np.random.default_rng(26900 + epoch)
    perm = rng.permutation(n_cells).astype(np.int64)
    uni = rng.random((n_cells, 3), dtype=np.float32)
    return perm, uni


def table_summary(tbl, offsets):
    """what the fixture keeps of a (n_rows, 2) table-shaped array: level sums / abs sums, 4 seeded +-1 projections, level 0 and every
    1024th row"""
    t = np.asarray(tbl, np.float32).reshape(-1, 2)
    L = len(offsets) - 1
    out = {'level_sum': np.stack([t[offsets[l]:offsets[l + 1]].astype(np.float64).sum(0) for l in range(L)]),
           'level_abs': np.stack([np.abs(t[offsets[l]:offsets[l + 1]]).astype(np.float64).sum(0) for l in range(L)]),
           'level0': t[:offsets[1]].copy(), 'rows_mod1024': t[5::1024].copy()}
    sgn = np.random.default_rng(77).integers(0, 2, size=(4,) + t.shape, dtype=np.int8)
    out['proj'] = np.array([((sgn[i].astype(np.float64) * 2 - 1) * t).sum() for i in range(4)])
    return out


# ---- test side ------------------------------------------------------------------------------------------------------------------
def golden():
    from conftest import load_golden
    return load_golden('g26_trajectory')


def check_inputs(g, leg):
    """the generators above give the fixture's inputs on this box too"""
    epochs = LEGS[leg]['epochs']
    for k, (e, n) in enumerate(zip(epochs, g[leg + '_n_rays'])):
        inp = step_inputs(e, int(n))
        sums = [float(np.float64(inp[key]).sum()) for key in ('rays_o', 'rays_d', 'bkg_color', 'img')]
        assert np.allclose(sums, g[leg + '_input_sums'][k], rtol=0, atol=1e-9), (leg, k)


def table(g, leg, n_rows):
    t = table_from_seed(n_rows, 2, LEGS[leg]['table_seed'], LEGS[leg]['table_amp'])
    assert abs(t.astype(np.float64).sum() - float(g[leg + '_table_sum'])) < 1e-9 and np.array_equal(t[::100003], g[leg + '_table_probe'])
    return t


class Tape:
    """arcnerf_amd.geometry.volume.set_refresh_tape(Tape()): VolumeBound.optimize / NgpPipeline.update_occupancy draw what the reference
    run drew (refresh_draws)"""

    def draws(self, epoch, n_cells, device):
        import torch
        perm, uni = refresh_draws(epoch, n_cells)
        return torch.from_numpy(perm).to(device), torch.from_numpy(uni).to(device)


def loss_bars(g, leg, floor=1e-4, factor=5.0):
    """per step: how far a loss may be from the fixture's, relative.  `floor` until a refresh has had near-threshold cells to decide;
    from then on `factor` x the running maximum of the distance between the reference's two runs (the second one decides every such cell
    the other way: tests/golden/make_golden_trajectory.py), never below the floor."""
    la, lb = g[leg + '_loss'], g['alt_' + leg + '_loss']
    env = np.maximum.accumulate(np.abs(la - lb) / np.abs(la))
    return np.maximum(floor, factor * env)


def count_bars(g, leg, factor=5.0):
    """per step: relative bar on the number of valid samples once an occupancy decision has differed (equal before): `factor` x the
    running maximum of the distance between the reference's two runs, at least 1 %"""
    a, b = g[leg + '_n_valid'].astype(np.float64), g['alt_' + leg + '_n_valid'].astype(np.float64)
    return np.maximum(0.01, factor * np.maximum.accumulate(np.abs(a - b) / a))


def check_bitfield(g, leg, i, bits, flips_so_far=0):
    """bits (n_cells,) bool after the i-th refresh of the leg.  While no decision has differed yet (flips_so_far == 0): equal to the
    reference's except on cells the fixture marks as within NEAR_BAND of the threshold.  Once one has, the two runs train on different
    samples and later bitfields differ away from the band too - exactly what the reference's second run (which decides the near cells the
    other way) shows: then at most 5 x its count of differing cells + 16.  Returns the number of differing cells."""
    ref = np.unpackbits(g[leg + '_bitfields'][i], bitorder='little').astype(bool)
    near = np.unpackbits(g[leg + '_near'][i], bitorder='little').astype(bool)
    diff = np.asarray(bits, bool).reshape(-1) != ref
    if flips_so_far == 0:
        assert not (diff & ~near).any(), (leg, 'refresh', i, 'cells decided differently away from the threshold:', int((diff & ~near).sum()))
    else:
        alt = int(np.unpackbits(g[leg + '_bitfields'][i] ^ g['alt_' + leg + '_bitfields'][i]).sum())
        assert int(diff.sum()) <= 5 * alt + 16, (leg, 'refresh', i, int(diff.sum()), 'cells differ; the reference\'s two runs differ in', alt)
    return int(diff.sum())


def param_report(g, leg, step, tbl, nets):
    """distances of a parameter state after `step` steps to the fixture's summary: {'level_sum': max over levels of |sum - ref| / ref abs
    sum, 'level_abs': same for the abs sums, 'rows': share of the stored rows (level 0, every 1024th) further than 2 % of the level's mean
    |value| away, 'net.<name>': max |W - ref| / max |ref|}"""
    pre = '{}_p{}.'.format(leg, step)
    offs = g['offsets']
    s = table_summary(tbl, offs)
    la = g[pre + 'table.level_abs']
    out = {'level_sum': float((np.abs(s['level_sum'] - g[pre + 'table.level_sum']) / la).max()),
           'level_abs': float((np.abs(s['level_abs'] - la) / la).max()),
           'proj': float(np.abs(s['proj'] - g[pre + 'table.proj']).max() / la.sum())}
    ref_rows = np.concatenate([g[pre + 'table.level0'], g[pre + 'table.rows_mod1024']])
    rows = np.concatenate([s['level0'], s['rows_mod1024']])
    scale = np.abs(ref_rows).mean() + 1e-12
    out['rows'] = float((np.abs(rows - ref_rows).max(1) > 0.02 * scale).mean())
    for name, W in nets.items():
        ref = g[pre + name]
        d = np.abs(np.asarray(W).reshape(ref.shape) - ref) / np.abs(ref).max()
        out['net.' + name] = float(d.max())
        out['netshare.' + name] = float((d > 1e-3).mean())
    return out


def check_params(rep, tight, where):
    """bars on a param_report.  Adam at eps 1e-15 turns the SIGN of a gradient entry that is rounding noise (a first-layer weight whose
    inputs are 1e-4 features, a table row touched with cancelling contributions) into a step of size lr, so single entries may sit a few
    1e-3 (of the tensor's max) away even when everything else agrees to 1e-5: the bars are on sums, on the share of entries that moved,
    and a looser one on the maximum."""
    sums = 2e-4 if tight else 2e-2
    assert rep['level_sum'] <= sums and rep['level_abs'] <= sums and rep['proj'] <= sums, (where, rep)
    assert rep['rows'] <= (2e-3 if tight else 5e-2), (where, rep)
    for k, v in rep.items():
        if k.startswith('net.'):
            assert v <= (2e-2 if tight else 1e-1), (where, k, rep)
        if k.startswith('netshare.'):
            assert v <= (1e-2 if tight else 2e-1), (where, k, rep)
