"""where a module-path step synchronises the host with the device: python tools/exp_sync_sites.py <config>   (GPU box)
torch.cuda.set_sync_debug_mode('warn') + a traceback per warning"""
import os, sys, traceback, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ['neus_ngp_multivol'])
import runpy
import torch

g = runpy.run_path(os.path.join(ROOT, 'tools', 'exp_host_profile.py'), run_name='not_main') if False else None
# (re-using the set-up of exp_host_profile.py without its timing loops)
src = open(os.path.join(ROOT, 'tools', 'exp_host_profile.py')).read().split("for i in range(6):")[0]
ns = {'__file__': os.path.join(ROOT, 'tools', 'exp_host_profile.py'), '__name__': 'setup'}
exec(compile(src, 'exp_host_profile.py', 'exec'), ns)
step = ns['step']
for i in range(4):
    step(i)
torch.cuda.synchronize()
sites = {}


def show(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if '/arcnerf_amd/' in f.filename or 'bench' in f.filename]
    key = ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(st[-4:]))
    if not key:
        key = 'outside the package: ' + ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(traceback.extract_stack()[-8:-2]))
    sites[key] = sites.get(key, 0) + 1


warnings.showwarning = show
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
step(5)
torch.cuda.set_sync_debug_mode('default')
for k, v in sites.items():
    print(v, 'x', k)
