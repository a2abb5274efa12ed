#!/usr/bin/env python
"""The hash-grid forward gather on its own, on the bench's batch (8320 rays, ~2.6e5 samples, 5 % occupancy, config 2 table):
whole-launch time, time per level on ONE XCD (a one-level descriptor puts all workgroups on XCD slot 0), bit-identity of the
XCD-affine kernel against the plain row-major kernel.  Used with tools/pmc_kernel.sh for the TA / TCP / TCC counters.

    python tools/exp_gather.py [--levels] [--iters 50]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--levels', action='store_true', help='time every level alone')
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--rays', type=int, default=8320)
    ap.add_argument('--check', action='store_true')
    ap.add_argument('--only', type=int, default=-1, help='launch this level alone (one XCD), for counter runs')
    args = ap.parse_args()
    from arcnerf_amd import _native as N
    from arcnerf_amd.ops import functional as F
    from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays
    dev = torch.device('cuda:0')
    cfg = NgpConfig()
    fld = NgpField(cfg, device=dev, seed=0)
    pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20, packed_bits=True)
    pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.05, seed=0)))
    o, d = synthetic_rays(args.rays, seed=1000, device=dev)
    pipe.sample(o, d)
    n = int(pipe.n_dev.item())
    b = pipe.buf
    S = pipe.cap
    L = N.lib()
    st = N.stream()
    table = fld.view('table')
    feat = b['feat']

    def launch(desc, out=feat, lm=1):
        N.check(L.arcn_hashgrid_fwd_xcd(N.ptr(b['xyz']), N.ptr(table), N.C.addressof(desc), N.ptr(out), lm, S, S, pipe.n_dev.data_ptr(), st), 'fwd_xcd')

    def timeit(fn, iters):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, c in ev:
            a.record()
            fn()
            c.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(c) for a, c in ev)
        return t[len(t) // 2] * 1e3, t[0] * 1e3

    if args.only >= 0:
        l = args.only
        desc = N.make_hashgrid_desc([fld.resolutions[l]], [fld.offsets[l], fld.offsets[l + 1]], cfg.n_feat_per_entry, fld.min_xyz, fld.max_xyz)
        m, bst = timeit(lambda: launch(desc), args.iters)
        print('samples %d level %d alone: median %.1f us best %.1f' % (n, l, m, bst))
        return
    med, best = timeit(lambda: launch(fld.grid_desc), args.iters)
    alg = 1164.0 * n
    print('samples %d  variant %s  whole launch: median %.1f us  best %.1f us  -> %.2f TB/s algorithmic, %.3f of 8 TB/s' % (
        n, os.environ.get('ARCN_GATHER_VARIANT', '-'), med, best, alg / med / 1e6, alg / med / 1e6 / 8.0))
    if args.check:
        ref0 = torch.zeros((S, 32), device=dev)
        mp, bp = timeit(lambda: F.hashgrid_fwd(b['xyz'], table, fld.grid_desc, n_dev=pipe.n_dev, out=ref0), args.iters)
        print('plain kernel (one lane per (sample, level), every XCD touches all 16 levels, row-major output): median %.1f us best %.1f' % (mp, bp))
        ref = torch.zeros((S, 32), device=dev)
        F.hashgrid_fwd(b['xyz'], table, fld.grid_desc, n_dev=pipe.n_dev, out=ref)
        launch(fld.grid_desc)
        torch.cuda.synchronize()
        got = feat.view(16, S, 2)[:, :n].permute(1, 0, 2).reshape(n, 32)
        print('bit-identical to the row-major kernel:', bool(torch.equal(got, ref[:n])))
    if args.levels:
        tot = 0.0
        for l in range(cfg.n_levels):
            desc = N.make_hashgrid_desc([fld.resolutions[l]], [fld.offsets[l], fld.offsets[l + 1]], cfg.n_feat_per_entry, fld.min_xyz, fld.max_xyz)
            m, bst = timeit(lambda: launch(desc), max(10, args.iters // 3))
            tot += m
            print('level %2d res %4d rows %7d: %.1f us on one XCD (best %.1f)' % (l, fld.resolutions[l], fld.offsets[l + 1] - fld.offsets[l], m, bst))
        print('sum over levels %.1f us -> /8 XCDs = %.1f us if perfectly balanced' % (tot, tot / 8))


if __name__ == '__main__':
    main()
