#!/bin/bash
# Shader clock and package power (rocm-smi) while (a) the headline training step runs in a loop and (b) its hash-grid gather runs alone back to
# back on the same batch: the gather's "alone" time (bench line roofline_lookup.alone_launch_ms) is taken at the clock of an idle chip, its
# in-step time at the clock the whole step is held to by the package power limit.  usage (GPU box): tools/exp_clock_step.sh
run() {
python - "$1" <<'PY' &
import sys, time, torch
sys.path.insert(0, '.')
from arcnerf_amd import _native as N
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays
kind = sys.argv[1]
dev = torch.device('cuda:0')
cfg = NgpConfig()
fld = NgpField(cfg, device=dev, seed=0)
pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 20, packed_bits=True, prefetch_depth=2)
pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.05, seed=0)))
pool = [synthetic_rays(8320, seed=i, device=dev) for i in range(4)]
tgt = torch.rand(8320, 3, device=dev)
L, st = N.lib(), N.stream()
pipe.sample(*pool[0])
b, S = pipe.buf, pipe.cap
t0 = time.time()
n, ev = 0, None
while time.time() - t0 < 9:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200):
        if kind == 'step':
            pipe.train_step(pool[i % 4][0], pool[i % 4][1], tgt, next_rays=pool[(i + 2) % 4])
        else:
            N.check(L.arcn_hashgrid_fwd_xcd(N.ptr(b['xyz']), N.ptr(fld.view('table')), N.C.addressof(fld.grid_desc), N.ptr(b['feat']), 1, S, S, pipe.n_dev.data_ptr(), st), 'fwd')
    e1.record()
    torch.cuda.synchronize()
    ev = e0.elapsed_time(e1) / 200
print('# %s: %.4f ms per launch (last 200)' % (kind, ev))
PY
PID=$!
sleep 5
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -2; sleep 1; done
wait $PID
}
echo "# (a) the training step in a loop"; run step
echo "# (b) the gather alone, back to back"; run gather
