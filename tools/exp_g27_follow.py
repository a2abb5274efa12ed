"""debug: the module path / fused step on G27's data next to the reference's recorded losses and sample counts, iteration by iteration"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import g27_utils as U
import test_gpu_psnr as T
g = U.golden()
gpu = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fused = len(sys.argv) > 2 and sys.argv[2] == 'fused'
seed = 0
def on_step(epoch, res):
    tag = 's0_'
    print(epoch, 'loss', round(res['loss'][-1], 3), 'ref', round(float(g[tag + 'loss'][epoch]), 3), 'samples', res['n_valid'][-1], 'ref', int(g[tag + 'n_valid'][epoch]), flush=True)
r = T.run_module_api(g, gpu, seed, fused=fused, n_epoch=n, on_step=on_step)
print('rays', r['n_rays'][:n] == g['s0_n_rays'].tolist()[:n], 'psnr', r['psnr'], 'occupied', r['occupied'])
