"""call sites of zero-filled allocations (torch.zeros / zeros_like / new_zeros / zero_ / fill_) in one module-path step:
python tools/exp_alloc_sites.py <config>   (GPU box)"""
import os, sys, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ['neus_ngp_multivol'])
import torch
src = open(os.path.join(ROOT, 'tools', 'exp_host_profile.py')).read().split("for i in range(6):")[0]
ns = {'__file__': os.path.join(ROOT, 'tools', 'exp_host_profile.py'), '__name__': 'setup'}
exec(compile(src, 'exp_host_profile.py', 'exec'), ns)
step = ns['step']
for i in range(4):
    step(i)
torch.cuda.synchronize()
sites = collections.Counter()


def site():
    st = [f for f in traceback.extract_stack()[:-2] if '/arcnerf_amd/' in f.filename]
    return ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(st[-2:])) or 'outside'


def wrap(mod, name):
    real = getattr(mod, name)

    def f(*a, **k):
        r = real(*a, **k)
        t = r if isinstance(r, torch.Tensor) else (a[0] if a and isinstance(a[0], torch.Tensor) else None)
        if t is not None and t.is_cuda:
            sites[(name, site(), t.numel())] += 1
        return r
    setattr(mod, name, f)


for n in ('zeros', 'zeros_like', 'ones', 'ones_like', 'full', 'full_like'):
    wrap(torch, n)
for n in ('new_zeros', 'zero_', 'fill_', 'new_ones', 'new_full'):
    wrap(torch.Tensor, n)
step(5)
for (n, s, numel), c in sorted(sites.items(), key=lambda kv: -kv[0][2]):
    print('%2d x %-11s %10d elems  %s' % (c, n, numel, s))
