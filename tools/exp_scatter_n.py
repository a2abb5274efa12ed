"""producer / consumer of the binned scatter (arcn_hashgrid_bwd_lm) against the number of samples: kernel durations from rocprofv3
    cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/sn -o sn --output-format csv -- python tools/exp_scatter_n.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd import _native as N
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays

dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev)
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, 0)))
S = pipe.cap
b = pipe.buf
dt = torch.zeros_like(fld.view('table'))
b['d_feat'].normal_()
ws = pipe.hash_ws
L, st = N.lib(), N.stream()
for rays in (64, 1024, 2080, 4160, 8320, 16640):
    o, d = synthetic_rays(rays, seed=1000, device=dev)
    pipe.sample(o, d)
    n = int(pipe.n_dev.item())

    def run():
        N.check(L.arcn_hashgrid_bwd_lm(N.ptr(b['xyz']), N.ptr(b['d_feat']), S, N.C.addressof(fld.grid_desc), N.ptr(dt), N.ptr(ws), ws.numel(), S,
                                       pipe.n_dev.data_ptr(), st), 'bwd_lm')
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(20):
        run()
    e[1].record(); torch.cuda.synchronize()
    print('rays %6d samples %7d  scatter_lm %.1f us' % (rays, n, e[0].elapsed_time(e[1]) / 20 * 1e3))
