"""debug: train the module path (reference net semantics) 50 iterations on G27's data, evaluate, dump the parameters and predictions"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import g27_utils as U
import test_gpu_psnr as T
g = U.golden()
gpu = torch.device('cuda:0')
keep = {}
orig = T.evaluate
def evaluate(m, rays, target):
    ps, ws = orig(m, rays, target)
    m.eval()
    with torch.no_grad():
        out = m({'rays_o': rays[0][0][None], 'rays_d': rays[1][0][None], 'rays_r': rays[2][0][None]}, inference_only=True)
        keep['pred0'] = out['rgb'][0].cpu().numpy()
        keep['mask0'] = out['mask'][0].cpu().numpy() if 'mask' in out else None
        keep['keys'] = sorted(out.keys())
    m.train()
    keep['sd'] = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if 'volume_pts' not in k and 'grid_pts' not in k}
    keep['psnr'] = ps
    return ps, ws
T.evaluate = evaluate
r = T.run_module_api(g, gpu, 0, fused=False, n_epoch=50)
print('psnr', r['psnr'], keep['keys'], 'pred mean', keep['pred0'].mean(0), 'mask mean', None if keep['mask0'] is None else keep['mask0'].mean())
os.makedirs('gpurun_out', exist_ok=True)
np.savez_compressed('gpurun_out/g27_dump.npz', pred0=keep['pred0'], **{'sd.' + k: v for k, v in keep['sd'].items()})
