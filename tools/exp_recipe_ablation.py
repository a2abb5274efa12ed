"""Which part of the reference's data recipe keeps the NGP run from the all-white collapse (round-4 VERDICT, missing #1): tools/psnr_recipe.py
with the scheduler's precrop and / or random background colour switched off, seeds 0..5, held-out PSNR after 100 / 500 / 2000 iterations."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import psnr_recipe as P

variants = {'recipe': dict(P.SCHEDULER),
            'no precrop': {k: v for k, v in P.SCHEDULER.items() if k != 'precrop'},
            'white bkg instead of random': dict(P.SCHEDULER, bkg_color={'color': [1.0, 1.0, 1.0]}),
            'neither': {'bkg_color': {'color': [1.0, 1.0, 1.0]}, 'dynamic_batch_size': {'update_epoch': 16}}}
out = {}
for name, sch in variants.items():
    P.SCHEDULER = sch
    rows = []
    for seed in range(6):
        r = P.run(2000, seed=seed, report=(100, 500, 2000))
        rows.append([round(p['psnr'], 2) for p in r['points']])
    out[name] = rows
    print(name, rows, file=sys.stderr, flush=True)
print(json.dumps(out))
