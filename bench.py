#!/usr/bin/env python
"""bench.py — ray-samples/sec (train) of the NGP-Lego hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full training pass of the hot path over one batch of synthetic rays: occupancy
marching -> hash-grid encode -> fused geo/radiance MLPs -> compositing -> Huber loss -> backward -> (one RCCL
all-reduce of the flat gradient buffer when N > 1) -> fused Adam + EMA, plus the occupancy-grid refresh every 16 steps
at the reference cadence (its result is kept off the marching bitfield so the workload stays the named configuration).
Inputs (rays, targets, occupancy, parameters) are resident in HBM before the timed region.
value = valid samples evaluated by the nets over all ranks / max-over-ranks wall time of the K timed steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# This process builds SIX models one after another (the headline pipeline and five side legs), each with its own side streams.  The HIP
# runtime hands streams to GPU_MAX_HW_QUEUES (default 4) hardware queues round robin: by the fourth leg the sampling stream of config 4
# lands on the queue of the stream it is meant to run BESIDE and the prefetched samplers serialise with the step (1.96 instead of 1.82 ms;
# a process that builds one model does not see it - DESIGN.md 12c).  Eight queues keep them apart; the headline step is unchanged by it
# (0.564 - 0.567 ms with 4 and with 8, alternated).  Must be set before the runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X HBM3E (MI355X_MICROARCH.md)
# algorithmic bytes per valid sample (NGP config fp32 table)
BYTES_HASH_FWD = 16 * 8 * 2 * 4 + 12 + 32 * 4          # 1164
BYTES_HASH_BWD = 32 * 4 + 12 + 2 * (16 * 8 * 2 * 4)     # 128 grad in + xyz + atomic payload counted as RMW = 2188


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=64)
    ap.add_argument('--warmup', type=int, default=16)
    ap.add_argument('--target-samples', type=int, default=1 << 18, help='valid samples per step (log_max_allowance 18)')
    ap.add_argument('--occupancy', type=float, default=0.05)
    ap.add_argument('--bkg-occupancy', type=float, default=-1.0, help='config 4: occupied fraction of every level of the background cascade (default: the same as --occupancy; 1.0 = the unpruned start-of-training grid)')
    ap.add_argument('--no-occ-update', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-psnr', action='store_true', help='skip the short PSNR@iter run on the analytic scene appended to the default line')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the short runs of BASELINE configs 1 / 3 / 4 / 5 appended to the default line')
    ap.add_argument('--cpu-rays', type=int, default=131072)
    ap.add_argument('--config', default='ngp', choices=['ngp', 'ngp_module', 'nerf', 'neus', 'neus_ngp_multivol', 'neus_ngp_nerfpp', 'hdrnerf'],
                    help='BASELINE.json configs: ngp = config 2 (default, the headline; ngp_module = the same model through the drop-in module path build_model(nerf_ngp.yaml)), nerf = 1, neus = 3, neus_ngp_multivol = 4, hdrnerf = 5')
    ap.add_argument('--rays', type=int, default=0, help='rays per step per GPU for the module-path configs (0 = the config default)')
    ap.add_argument('--chunk-pts', type=int, default=0, help='points per net evaluation chunk of the module-path configs (0 = the yaml value, the reference\'s 4096*32: a memory knob sized for an 11 GB card; the chunks are independent, so it changes launch sizes only)')
    return ap.parse_args()


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` without a launcher (RANK / WORLD_SIZE unset) starts the N ranks itself, one process per GPU, the way the
    reference's launcher does (scripts/gpu.sh:9-21 -> common/trainer/basic_trainer.py:73-111: mp.spawn of one worker per gpu id with a
    tcp://127.0.0.1 rendezvous): this process re-executes the same command line under `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1`, rank 0 of the children prints the ONE JSON line, and the exit code is passed on.
    Under torchrun (the driver's N > 1 form) RANK / WORLD_SIZE are set and this is a no-op."""
    if args.gpus <= 1 or ('RANK' in os.environ and 'WORLD_SIZE' in os.environ):
        return
    import socket
    import subprocess
    backend = os.environ.get('ARCN_DIST_BACKEND', 'nccl')
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        sys.exit('bench.py needs a GPU (there is no CPU fallback for the product path)')
    if backend == 'nccl' and args.gpus > n_dev:
        sys.exit('bench.py --gpus {}: only {} GPU(s) visible and RCCL wants one GPU per rank '
                 '(ARCN_DIST_BACKEND=gloo lets several ranks share a GPU for functional tests)'.format(args.gpus, n_dev))
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL's intra-node transport on this host driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    env['ARCN_BENCH_SELF_SPAWNED'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node={}'.format(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.exit(subprocess.call(cmd, env=env))


def dist_report(dist, world, launch, payload_bytes, n_collectives, per_rank, extra=None):
    """The `rccl` object of the JSON line: which backend carried the gradient exchange, how many ranks it saw, bytes per step."""
    if dist is None:
        return None
    r = {'backend': dist.get_backend(), 'world_size_seen': dist.get_world_size(), 'launcher': launch,
         'allreduce_bytes_per_step': int(payload_bytes), 'collectives_per_step': int(n_collectives),
         'ring_bytes_on_the_wire_per_gpu': int(2 * (world - 1) / world * payload_bytes), 'per_rank_samples_per_step': per_rank}
    if extra:
        r.update(extra)
    return r


def time_allreduce_alone(dist, buf, iters=10):
    """The gradient all-reduce with nothing beside it, HIP events on the current stream (a synchronous collective makes the current
    stream wait for the communicator's): ms per call and the ring's bus bandwidth.  After the timed region, on a scratch buffer."""
    scratch = torch.zeros_like(buf)
    for _ in range(3):
        dist.all_reduce(scratch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        dist.all_reduce(scratch)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    w = dist.get_world_size()
    return {'allreduce_alone_ms': ms, 'allreduce_busbw_GBps': 2 * (w - 1) / w * scratch.numel() * 4 / (ms * 1e-3) / 1e9}


COLD_STEPS = 96   # untimed steps of the workload before the W warmup steps (see main); a multiple of the ray pool size (8)
ROOFLINE_KERNELS = ('hashgrid_fwd', 'hashgrid_bwd')
TABLE_KERNELS = ROOFLINE_KERNELS + ('mlp_fwd', 'mlp_bwd', 'mlp_bwd_dw', 'march_count', 'composite_packed_train', 'composite_packed_fwd',
                                    'composite_packed_bwd', 'adam_ema_step')


# entry points of the level-major path are reported under the name of the op they implement
ALIASES = {'hashgrid_fwd_xcd': 'hashgrid_fwd', 'hashgrid_bwd_lm': 'hashgrid_bwd', 'hashgrid_bwd_lm_adam': 'hashgrid_bwd', 'hashgrid_bwd_lm_adam_planned': 'hashgrid_bwd', 'hashgrid_bwd_lm_planned': 'hashgrid_bwd', 'adam_ema_step_runs': 'adam_ema_step', 'ngp_step_tail': 'adam_ema_step', 'mlp_fwd_lm': 'mlp_fwd', 'mlp_bwd_lm': 'mlp_bwd',
           'mlp_fwd_cat': 'mlp_fwd', 'mlp_bwd_cat': 'mlp_bwd', 'ngp_nets_fwd': 'mlp_fwd', 'march_count_culled': 'march_count', 'march_count_waves': 'march_count'}


class KernelTimers:
    """HIP events (torch.cuda.Event on the launch stream == torch's current stream) around individual C-ABI entry points.
    Only the names in `enabled` are bracketed, and of those every `every`-th launch: two event records per launch are not free on a
    0.6 ms step, so the timed region brackets the roofline candidates only (a quarter of their launches) and the full per-kernel table
    comes from a separate, untimed pass."""

    def __init__(self):
        self.pairs = {}
        self.enabled = set()
        # every 4th launch of a bracketed entry point carries the two event records: bracketing EVERY launch cost 1.6 % of the step (0.625 vs
        # 0.612 ms with none, three alternations in one session; the records keep the neighbouring kernels from overlapping), every 4th 0.5 %
        self.every = 4
        self.calls = {}
        self.main = None

    def wrap(self, name, fn):
        def inner(*a, **k):
            if name not in self.enabled:
                return fn(*a, **k)
            # (the launches of the occupancy refresh on its side stream - another size, another stream - are not the step's)
            if self.main is not None and torch.cuda.current_stream() != self.main:
                return fn(*a, **k)
            c = self.calls.get(name, 0)
            self.calls[name] = c + 1
            if c % self.every:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            self.pairs.setdefault(name, []).append((e0, e1))
            return r
        return inner

    def reset(self, enabled):
        self.pairs = {}
        self.calls = {}
        self.enabled = set(enabled)
        self.main = torch.cuda.current_stream() if torch.cuda.is_available() else None

    def summary(self):
        return {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in self.pairs.items()}


def instrument(timers):
    """Route the C-ABI entry points that make up the step through the timers (the library handle is shared by the package)."""
    from arcnerf_amd import _native as N
    lib = N.lib()

    class Proxy:
        def __init__(self, real):
            self._real = real
            self._cache = {}

        def __getattr__(self, name):
            if name not in self._cache:
                fn = getattr(self._real, name)
                label = ALIASES.get(name[5:], name[5:])
                self._cache[name] = timers.wrap(label, fn) if label in TABLE_KERNELS else fn
            return self._cache[name]

    proxy = Proxy(lib)
    N._lib = proxy
    return lib


# ---- configs 1 / 3 / 4 / 5: the reference-shaped module path (build_model -> FullModel.forward -> loss -> backward -> Adam) --------
F32_MFMA_PEAK = 157.3e12   # FLOP/s, v_mfma_f32_* (MI355X_MICROARCH.md): the nets of these configs compute in exact f32
MODULE_CONFIGS = {
    # name: yaml, default rays, fg net evaluations per ray and step, fwd FLOP per fg net evaluation (2 x MACs of the linear stacks)
    'nerf': dict(yaml='nerf.yaml', rays=4096, evals=64 + 64 + 128, flop=2 * (63 * 256 + 3 * 256 * 256 + 319 * 256 + 3 * 256 * 256 + 256 * 257 + 283 * 128 + 128 * 3),
                 desc='vanilla NeRF, freq encoder, 64+128 samples/ray (coarse 64 + fine 192 net evaluations), 8x256 + 128 nets, Adam'),
    'hdrnerf': dict(yaml='hdrnerf.yaml', rays=4096, evals=64 + 64 + 128, flop=2 * (63 * 256 + 3 * 256 * 256 + 319 * 256 + 3 * 256 * 256 + 256 * 257 + 283 * 128 + 128 * 3 + 3 * 2 * 128),
                    desc='HDR-NeRF: nerf nets + three 1->128->1 tone mappers, per-ray exposure, LDR + HDR compositing, Adam'),
    'neus': dict(yaml='neus.yaml', rays=2048, evals=64 + 64, flop=2 * (39 * 256 + 3 * 256 * 256 + 256 * 217 + 3 * 256 * 256 + 256 * 257 + (3 + 27 + 3 + 256) * 256 + 3 * 256 * 256 + 256 * 3),
                 desc='NeuS sdf net 8x256 softplus-100 + radiance 4x256, 64+64 samples/ray, 4 up-sampling rounds, normals + Eikonal (double backward), Adam'),
    # config 2 as a user of the reference's API gets it: configs/nerf_ngp.yaml (= the reference's configs/models/nerf_ngp.yaml) through
    # build_model -> FullModel.forward -> loss.backward() -> optimizer.step(): marching inline (the module API does not see the next
    # batch), scatter and optimiser as two passes, the occupancy refresh of VolumeBound.optimize at its cadence
    'ngp_module': dict(yaml='nerf_ngp.yaml', rays=8320, evals=None, flop=None,
                       desc='instant-ngp of config 2 through the drop-in module path (hash grid + fused MLPs + volume prune 128^3, packed samples inside NeRF._forward_packed), Adam'),
    'neus_ngp_multivol': dict(yaml='neus_ngp_multivol.yaml', rays=4096, evals=None, flop=None,
                              desc='NeuS on the hash grid in the pruned volume + MultiVol background (hash grid + fused MLPs), Adam'),
    # BASELINE.json's own wording of config 4: the same foreground with the NeRF++ inverted-sphere background (8 x 256 + 128 nets on 32 shell
    # samples per ray) - through the module path (the hand-ordered stepper covers the MultiVol background only)
    'neus_ngp_nerfpp': dict(yaml='neus_ngp_nerfpp.yaml', rays=4096, evals=None, flop=None,
                            desc='NeuS on the hash grid in the pruned volume + NeRF++ background (freq encoder, 8x256 + 128 nets, 32 samples/ray), Adam'),
}


def bench_module(args, name, emit=True):
    """One training step of the module path per `step`; samples = foreground net evaluations (rays x samples per ray of the final
    differentiated pass; the hierarchical up-sampling passes of NeuS are extra work inside the step, not counted)."""
    spec = MODULE_CONFIGS[name]
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count())
    assert torch.cuda.is_available(), 'bench.py needs a GPU (there is no CPU fallback for the product path)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get('ARCN_DIST_FORCE', '0') == '1'   # ARCN_DIST_FORCE=1: a one-rank communicator runs the same path
    if use_dist:
        import torch.distributed as dist
        from arcnerf_amd import distributed as D
        D.init_from_env(backend=os.environ.get('ARCN_DIST_BACKEND', 'nccl'), device=dev)
    from arcnerf_amd.models import build_model
    from arcnerf_amd.optim import FusedAdam
    from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_rays
    from arcnerf_amd.utils.cfgs_utils import load_configs
    n_rays = args.rays or spec['rays']
    torch.manual_seed(0)   # identical initial parameters on every rank
    m = build_model(load_configs(os.path.join(ROOT, 'configs', spec['yaml']), [])).to(dev)
    if args.chunk_pts:
        m.set_chunk_pts(args.chunk_pts)
    fg = m.fg_model
    bkg_occ = None
    if name == 'neus_ngp_multivol':
        # both occupancy structures in the same (converged) pruning state: the foreground volume AND every level of the background cascade
        # filled to --occupancy by synthetic blobs.  Round 2 left the cascade at its all-occupied initial state (--bkg-occupancy 1.0), which
        # put 1.9e6 background samples next to 1.2e5 foreground ones.
        from arcnerf_amd.pipeline import synthetic_cascade_bits
        fg.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, args.occupancy, seed=0)).to(dev), ops='overwrite')
        bkg_occ = args.occupancy if args.bkg_occupancy < 0 else args.bkg_occupancy
        if bkg_occ < 1.0:
            bits = synthetic_cascade_bits(m.bkg_model.n_grid, m.bkg_model.n_levels, bkg_occ, seed=5)
            m.bkg_model.density_bitfield.copy_(torch.from_numpy(bits).to(dev))
    if name in ('ngp_module', 'neus_ngp_nerfpp'):
        fg.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, args.occupancy, seed=0)).to(dev), ops='overwrite')
    radius = 2.2 if name in ('neus_ngp_multivol', 'neus_ngp_nerfpp') else (3.0 if name == 'neus' else (3.0 / 1.05 if name == 'ngp_module' else 4.0))
    pool = []
    g = torch.Generator(device='cpu').manual_seed(77 + rank)
    for i in range(4):
        o, d = synthetic_rays(n_rays, seed=10 * rank + i, device=dev, radius=radius)
        inp = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
               'bkg_color': torch.rand(1, n_rays, 3, generator=g).to(dev), 'img': torch.rand(1, n_rays, 3, generator=g).to(dev)}
        if name == 'hdrnerf':
            inp['exp_time'] = (torch.rand(1, n_rays, 1, generator=g) * 4.0 + 0.1).to(dev)
        pool.append(inp)
    params = [p for p in m.parameters() if p.requires_grad]
    # parameters, gradients and Adam state in ONE flat buffer: p.grad are views autograd accumulates into, so the data-parallel exchange
    # is one collective on that buffer with no gather / scatter copies, DDP's 1 / world is the optimiser's grad_scale, and the
    # optimiser and the gradient clear are one launch each
    opt = FusedAdam(params, lr=5e-4, eps=1e-15, zero_grad_on_step=True).flatten()      # (the kernel clears the gradients while it reads them: no 48.8 MB memset per step)
    opt.grad_scale = 1.0 / world
    flat_grads = opt.flat_grads()
    flat_numel = sum(p.numel() for p in params)
    if use_dist:
        D.broadcast_params(opt.flat_params(), src=0)   # what DDP does at construction

    ngp_loss = None
    if name == 'ngp_module':
        from arcnerf_amd import trainer as T_
        lc = type('C', (), {})()
        lc.loss = type('C', (), {})()
        lc.loss.ImgLoss = type('C', (), dict(keys=['rgb_coarse'], loss_type='Huber', delta=0.1, weight=3000.0))()
        ngp_loss = T_.build_loss(lc)

    neus_loss = None
    if name in ('neus_ngp_multivol', 'neus_ngp_nerfpp'):
        # the loss block of the reference's capture_qqtiger_neusngp_multivol.yaml:262-269: ImgLoss Huber 0.1 x 5 on rgb + EikonalLoss x 0.1 on normal_pts
        from arcnerf_amd import trainer as T_
        from arcnerf_amd.utils.cfgs_utils import dict_to_obj
        neus_loss = T_.build_loss(dict_to_obj({'loss': {'ImgLoss': {'loss_type': 'Huber', 'delta': 0.1, 'weight': 5.0},
                                                        'EikonalLoss': {'key': 'normal_pts', 'weight': 0.1}}}))

    def loss_of(out, inp):
        if neus_loss is not None:
            return neus_loss(inp, out)['sum']
        if name == 'ngp_module':      # ImgLoss(Huber, delta 0.1, weight 3000) of the reference's NGP recipe (arcnerf/loss/img_loss.py:60-100, nerf_lego_nerf_ngp.yaml:192-197)
            return ngp_loss(inp, out)['sum']
        if name in ('nerf', 'hdrnerf'):
            l = ((out['rgb_fine'] - inp['img']) ** 2).mean() + ((out['rgb_coarse'] - inp['img']) ** 2).mean()
            if name == 'hdrnerf':
                l = l + 0.5 * sum(((out['unit_exp_' + s] - 0.5) ** 2).mean() for s in ('coarse', 'fine'))
            return l
        return ((out['rgb'] - inp['img']) ** 2).mean() + 0.1 * ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()

    n_eval = [0]
    # pruned volume (evals None): the valid foreground samples are counted where the model itself measures them - FgModel.adjust_dynamicbs_factor
    # (fg_model.py:100-115) keeps every step's count in a device ring until the dynamic batch size is read; the bench reads that ring
    # once after the timed region (no extra launch inside it)
    if spec['evals'] is None and fg.render_cfgs['max_allowance'] <= 0:
        fg.render_cfgs['max_allowance'] = 1 << 18   # a yaml without dynamic batch size: switch the MEASUREMENT on (nothing resizes the batches here)
    bkg_samples = [0, 0]
    if name == 'neus_ngp_multivol':
        # the background's packed samples: the one host read its path makes anyway (ops.functional.pack_dense_samples) is tallied
        from arcnerf_amd.ops import functional as Fn
        real_pack = Fn.pack_dense_samples_end

        def counting_pack(handle):
            r = real_pack(handle)
            bkg_samples[0] += int(r[4])
            bkg_samples[1] += 1
            return r
        Fn.pack_dense_samples_end = counting_pack

    # the drop-in NGP step (ARCN_MODULE_STEP): `fused` (default) = trainer.FusedNgpStep, the module API on the pipeline's fused step (loss in the
    # compositor, optimiser in the scatter, the next batch marched a step early); `eager` = every kernel of the module path issued eagerly
    fused = None
    mode = os.environ.get('ARCN_MODULE_STEP', 'fused')
    if name == 'ngp_module' and mode == 'fused' and (not use_dist or world > 1):
        # (N > 1: every rank its rays through the same stepper, the gradient summed in level groups overlapped with the scatter)
        from arcnerf_amd.trainer import FusedNgpStep
        fused = FusedNgpStep(m, ngp_loss, opt, None, max_rays=n_rays, ahead=int(os.environ.get('ARCN_MODULE_AHEAD', '2')), world_size=world,
                             grad_sync=os.environ.get('ARCN_GRAD_SYNC', 'flat'))

    prefetch = name in ('neus_ngp_multivol', 'neus_ngp_nerfpp') and os.environ.get('ARCN_PREFETCH_SAMPLES', '1') != '0'

    fused_neus = None
    if name == 'neus_ngp_multivol' and mode == 'fused':
        # config 4 as a hand-ordered kernel chain (trainer.FusedNeusNgpStep): no autograd engine between the kernels, the next batch's samplers
        # on a second stream; N > 1: every rank its rays through the same chain, the flat gradient summed between the scatters and the
        # optimiser (ARCN_GRAD_SYNC=flat: one all-reduce, the default | sharded); ARCN_MODULE_STEP=eager: the module path
        from arcnerf_amd.trainer import FusedNeusNgpStep
        gs = os.environ.get('ARCN_GRAD_SYNC', 'flat')
        # (ARCN_NEUS_FUSED_GEO=0: both geometry nets as round 5's chains of dense products instead of arcn_geo2_fwd / _bwd - the A/B switch)
        fused_neus = FusedNeusNgpStep(m, neus_loss, opt, world_size=world, grad_sync=gs if gs in ('flat', 'sharded') else 'flat',
                                      fused_geo=os.environ.get('ARCN_NEUS_FUSED_GEO', '1') != '0', march_at=os.environ.get('ARCN_NEUS_MARCH_AT', 'opt'), bkg_stream=os.environ.get('ARCN_NEUS_BKG_STREAM', '1') != '0')

    def step(i):
        inp = pool[i % len(pool)]
        if fused_neus is not None:
            # (the samplers of the next TWO batches on the sampling stream: the host can issue a whole step ahead of the device, ARCN_NEUS_AHEAD=1: one)
            ahead = [pool[(i + k) % len(pool)] for k in range(1, 1 + int(os.environ.get('ARCN_NEUS_AHEAD', '2')))]
            return fused_neus(inp, 20000 + i, next_feed_in=ahead if prefetch else None)[1]['sum']
        if fused is not None:
            return fused(inp, 20000 + i, next_feed_in=[pool[(i + k) % len(pool)] for k in range(1, fused.depth + 1)])[1]['sum']
        out = m({k: v for k, v in inp.items()}, inference_only=False, cur_epoch=20000 + i)
        if prefetch:      # the samplers of the NEXT batch on the sampling stream, beside this step's backward (FullModel.prefetch_samples)
            m.prefetch_samples(pool[(i + 1) % len(pool)])
        loss = loss_of(out, inp)
        loss.backward()
        if use_dist:   # DDP semantics (average of the ranks' gradients): SUM all-reduce of the flat buffer, 1 / world in the optimiser
            dist.all_reduce(flat_grads)
        opt.step()
        return loss

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if spec['evals'] is None:
        fg.reset_measurement()
    bkg_samples[0] = bkg_samples[1] = 0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trace, ev_trace = [], [torch.cuda.Event(enable_timing=True)]
    ev_trace[0].record()
    for i in range(args.steps):
        step(args.warmup + i)
        trace.append(time.perf_counter() - t0)
        ev_trace.append(torch.cuda.Event(enable_timing=True))      # (one event per step: the per-step device times below, ~2 us each)
        ev_trace[-1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_gpu_ms = sorted(a.elapsed_time(b) for a, b in zip(ev_trace[:-1], ev_trace[1:]))
    if os.environ.get('ARCN_BENCH_TRACE'):
        print('TRACE', name, [round(t * 1e3, 3) for t in trace], round(dt * 1e3, 3), 'gpu', [round(a.elapsed_time(b), 3) for a, b in zip(ev_trace[:-1], ev_trace[1:])], 'rebuilds', fused.rebuilds if fused is not None else None, file=sys.stderr, flush=True)
    if spec['evals'] is None:
        meter = fg._meter()
        k = int(meter._pending)
        n_eval[0] = int(sum(meter._ring[:k].tolist()) / max(1, args.steps)) if k else n_rays
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    evals_per_step = n_rays * spec['evals'] if spec['evals'] else n_eval[0]
    per_rank = [evals_per_step]
    rccl_extra = None
    if dist is not None:
        mine = torch.tensor([evals_per_step], dtype=torch.int64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [int(t.item()) for t in allr]
        rccl_extra = time_allreduce_alone(dist, flat_grads)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    total = sum(per_rank) * args.steps
    evals_per_step = sum(per_rank) / world
    roofline = None
    if spec['flop']:
        # fwd + bwd (dX and dW) = 3x the forward FLOP of the linear stacks; NeuS additionally differentiates the sdf net twice
        ach = 3.0 * spec['flop'] * evals_per_step / (wall / args.steps)
        roofline = {'kernel': 'linear stacks of the geometry / radiance nets (csrc/gemm.hip: split-bf16 MFMA products at f32 accuracy + fused MFMA kernels)', 'bound': 'mfma',
                    'achieved': ach / 1e12, 'peak': F32_MFMA_PEAK / 1e12, 'unit': 'TFLOP/s', 'frac': ach / F32_MFMA_PEAK, 'traffic': None,
                    'peak_split': 2500.0 / 6.0, 'frac_of_split_peak': ach / (2.5e15 / 6.0),
                    'note': 'algorithmic f32 FLOP = 3 x forward MACs x 2 per net evaluation, over the WHOLE step time; priced against the exact-f32 MFMA peak '
                            '(the products are f32-accurate; each is six bf16 MFMAs, so the matrix pipe itself does 6/8 of the f32 MFMA cycles); '
                            'peak_split = the dense bf16 MFMA peak / 6 terms = what the split form could do at full clock (the kernels sit on the '
                            '1400 W package limit at 1.93-1.97 GHz, DESIGN.md 5b)'}
    if name == 'neus_ngp_multivol':
        Fn.pack_dense_samples_end = real_pack
        s_bkg = bkg_samples[0] / max(1, bkg_samples[1])
        # HBM accounting of the hash-grid passes over the WHOLE step (SURVEY.md 8d per-sample figures): the background model encodes and
        # scatters once per sample; the foreground (sdf net with normals through the encoder) gathers twice (values, d enc / d x) and
        # scatters twice (first order, second order with 8 single-row records per sample and level)
        alg = (BYTES_HASH_FWD + BYTES_HASH_BWD) * s_bkg + (2 * BYTES_HASH_FWD + 2 * BYTES_HASH_BWD) * evals_per_step
        ach = alg / (wall / args.steps)
        roofline = {'kernel': 'hash-grid gathers + binned scatters of the foreground (first and second order) and the background model, over the WHOLE step',
                    'bound': 'hbm', 'achieved': ach / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': ach / HBM_PEAK, 'traffic': None,
                    'fg_points_per_step': evals_per_step, 'bkg_samples_per_step': s_bkg,
                    'note': 'a module-path step: 2.7 ms of kernels in 3.6 ms, of which the hash passes are ~1.2 ms (profiles/r3d_*); the step is '
                            'launch / host bound, not bandwidth bound - the fraction says how far, it is not a kernel figure'}
    if name == 'neus_ngp_nerfpp':
        alg = (2 * BYTES_HASH_FWD + 2 * BYTES_HASH_BWD) * evals_per_step
        ach = alg / (wall / args.steps)
        roofline = {'kernel': 'hash-grid gathers + binned scatters of the foreground (first and second order), over the WHOLE step', 'bound': 'hbm',
                    'achieved': ach / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': ach / HBM_PEAK, 'traffic': None, 'fg_points_per_step': evals_per_step,
                    'bkg_samples_per_step': n_rays * 32,
                    'note': 'a module-path step (autograd engine between the kernels); the background is 4096 x 32 evaluations of the 8 x 256 + 128 NeRF++ nets '
                            '(split-bf16 MFMA products): the fraction says how far the step is from the foreground\'s HBM bound, it is not a kernel figure'}
    if name == 'ngp_module':
        alg = (BYTES_HASH_FWD + BYTES_HASH_BWD) * evals_per_step
        ach = alg / (wall / args.steps)
        roofline = {'kernel': 'hash-grid gather + binned scatter, over the WHOLE step of the module path', 'bound': 'hbm', 'achieved': ach / 1e9,
                    'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': ach / HBM_PEAK, 'traffic': None,
                    'note': ('the headline step behind the reference-shaped API (build_model + FusedAdam.flatten + ImgLoss -> trainer.FusedNgpStep): plain Adam '
                             '(no EMA in this config)' if fused is not None else
                             'the same kernels as the headline step behind the reference-shaped API: marching inline on the step\'s stream, '
                             'scatter and optimiser as two passes (FusedAdam over the flat buffer, no EMA pass)')}
    cpu = None
    if world == 1 and not args.no_cpu_baseline and name == 'nerf' and emit:
        cpu = cpu_baseline_nerf()
    out = {'metric': 'ray-samples/sec (train)', 'value': total / wall, 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': wall / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'f32 (dense layers wider than 64: f32 operands as three bf16 planes, six MFMA terms, f32 accumulate - f32 accuracy)', 'data': 'synthetic',
           'config': {'workload': '{} ({}), {} rays/step/GPU, {} net evaluations/step/GPU, module path build_model({}) + FusedAdam'.format(
               name, spec['desc'], n_rays, evals_per_step, spec['yaml']), 'rays_per_step_per_gpu': n_rays,
               'samples_per_step_per_gpu': evals_per_step, 'n_params': flat_numel, 'parallelism': 'ray-sharded dp{}'.format(world),
               'chunk_pts': int(m.get_chunk_pts()), 'occupancy': args.occupancy, 'bkg_occupancy': bkg_occ,
               # device time between the steps' end events, median / slowest: a step whose HOST side was stalled (shared hosts: this step is ~65 %
               # host time on a quiet machine) shows in ms_per_step and in the maximum, not in the median
               'step_ms_device': {'p50': step_gpu_ms[len(step_gpu_ms) // 2], 'max': step_gpu_ms[-1]} if step_gpu_ms else None,
               'launch': ('trainer.FusedNeusNgpStep: the step as a hand-ordered kernel chain (no autograd engine), the next batch\'s samplers on a second stream ({} steps{})'.format(
                              fused_neus.steps, ', gradient exchange: ' + fused_neus.grad_sync if fused_neus.dist_step else '') if fused_neus is not None
                          else ('trainer.FusedNgpStep: the module API on NgpPipeline.train_step over the flattened optimiser\'s buffers, next {} batches marched '
                                'early ({} steps in this run, {} eager warm-up steps before)'.format(fused.depth, fused.steps, 2) if fused is not None
                                else 'every kernel of the module path issued eagerly'))},
           'rccl': dist_report(dist, world, LAUNCH, flat_grads.numel() * 4, 1, per_rank, rccl_extra),
           'roofline': roofline, 'cpu_baseline': cpu}
    if emit:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    return out


def module_leg_child(name, steps, warmup):
    """`bench.py --config name` in a child process -> its JSON line (the dict bench_module returns)"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'ARCN_DIST_FORCE')}
    cmd = [sys.executable, os.path.abspath(__file__), '--config', name, '--steps', str(steps), '--warmup', str(warmup), '--no-cpu-baseline']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if r.returncode != 0 or not line:
        raise RuntimeError('child bench of {} failed, rc {}: {}'.format(name, r.returncode, r.stderr[-400:]))
    out = json.loads(line[-1])
    out['config']['launch'] = out['config'].get('launch', '') + ' [run in a child process of the default bench]'
    return out


def dist1_leg(sync, steps, warmup, config=None):
    """The N > 1 code path at its single-GPU price: the headline command re-run in a child with a ONE-rank RCCL communicator (ARCN_DIST_FORCE=1) -
    scatter -> collective on the 48.8 MB flat gradient -> optimiser pass, instead of the optimiser fused into the scatter's consumer.  The only
    multi-GPU number a one-GPU box can give (common/trainer/basic_trainer.py:192-198 wraps the model in DDP: this is that step's per-GPU cost)."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.update({'ARCN_DIST_FORCE': '1', 'ARCN_GRAD_SYNC': sync, 'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '1', 'MASTER_ADDR': '127.0.0.1',
                'MASTER_PORT': str(port), 'HSA_ENABLE_IPC_MODE_LEGACY': env.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')})
    cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(steps), '--warmup', str(warmup), '--no-cpu-baseline', '--no-other-configs', '--no-psnr']
    if config is not None:      # (a module config: BASELINE config 4 through trainer.FusedNeusNgpStep(world_size, grad_sync) on the one-rank communicator)
        cmd += ['--config', config]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if r.returncode != 0 or not line:
        return {'error': 'rc {}: {}'.format(r.returncode, r.stderr[-400:])}
    j = json.loads(line[-1])
    rc = j.get('rccl') or {}
    p50 = (j.get('step_ms_spread') or {}).get('p50') or ((j.get('config') or {}).get('step_ms_device') or {}).get('p50')
    return {'ms_per_step': j['ms_per_step'], 'ms_per_step_p50': p50, 'samples_per_s': j['value'], 'steps': steps, 'warmup': warmup,
            'grad_sync': rc.get('grad_sync'), 'backend': rc.get('backend'), 'world_size_seen': rc.get('world_size_seen'),
            'exposed_ms': rc.get('exposed_ms'), 'exposed_ms_max': rc.get('exposed_ms_max'), 'allreduce_alone_ms': rc.get('allreduce_alone_ms'),
            'allreduce_bytes_per_step': rc.get('allreduce_bytes_per_step'),
            'workload': 'the {} step through the N > 1 path on a one-rank RCCL communicator (ARCN_DIST_FORCE=1, ARCN_GRAD_SYNC={})'.format(config or 'headline', sync)}


def inference_leg(dev, occupancy, images=4):
    """Inference / eval of the path (arcnerf/eval/infer_func.py:355-446, arcnerf_trainer.py:363): one 800 x 800 view = 640 000 rays through
    the drop-in module, model(inputs, inference_only=True) in eval mode - FullModel chunks them by the yaml's chunk_rays - on the same 5 %
    occupancy as the headline.  Rays from get_rays (on the GPU) are part of the timed image."""
    import math
    from arcnerf_amd.models import build_model
    from arcnerf_amd.pipeline import synthetic_bitfield
    from arcnerf_amd.render.ray_helper import get_rays
    from arcnerf_amd.utils.cfgs_utils import load_configs
    torch.manual_seed(0)
    m = build_model(load_configs(os.path.join(ROOT, 'configs', 'nerf_ngp.yaml'), ['--model.rays.white_bkg', 'True', '--model.obj_bound.bkg_color', '[1.0,1.0,1.0]'])).to(dev)
    fg = m.fg_model
    fg.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, occupancy, seed=0)).to(dev), ops='overwrite')
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if 'embeddings' in n_:
                p_.mul_(3000.0)     # a table with features of order 0.3 instead of the 1e-4 init: densities that vary over the volume
    m.eval()
    HW = 800
    focal = 0.5 * HW / math.tan(0.5 * 0.6911)
    K = torch.tensor([[focal, 0, HW / 2], [0, focal, HW / 2], [0, 0, 1.0]], device=dev)

    def cam(v):
        gg = torch.Generator(device='cpu').manual_seed(100 + v)
        c = torch.randn(3, generator=gg)
        c = c / c.norm() * (3.0 / 1.05)
        fwd = -c / c.norm()
        right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0]))
        right = right / right.norm()
        up = torch.linalg.cross(right, fwd)
        c2w = torch.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, -up, fwd, c
        return c2w.to(dev)
    cams = [cam(v) for v in range(images)]

    @torch.no_grad()
    def render(c2w):
        o, d, _, r = get_rays(HW, HW, K, c2w, wh_order=False, center_pixel=True)
        return m({'rays_o': o[None], 'rays_d': d[None], 'rays_r': r[None]}, inference_only=True)

    fg.reset_measurement() if hasattr(fg, 'reset_measurement') else None
    out = render(cams[0])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(images + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for v in range(images):
        out = render(cams[v])
        ev[v + 1].record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / images
    dev_ms = sorted(a.elapsed_time(b) for a, b in zip(ev[:-1], ev[1:]))
    return {'ms_per_image': dt * 1e3, 'ms_per_image_device_p50': dev_ms[len(dev_ms) // 2], 'rays_per_s': HW * HW / dt, 'images': images, 'rays_per_image': HW * HW,
            'finite': bool(torch.isfinite(out['rgb']).all()), 'keys': sorted(k for k in out.keys() if not k.startswith('progress'))[:8],
            'workload': 'one 800 x 800 view (640 000 rays: get_rays + FullModel.forward(inference_only=True), chunk_rays 32768) of build_model(nerf_ngp.yaml) in '
                        'eval mode, occupancy {:.0%}, white background'.format(occupancy)}


def cpu_baseline_nerf(rays_all=4096):
    """BASELINE.md section 3 for config 1: (a) the PyTorch-CPU-eager restatement of the reference modules (oracle/torch_cpu_nerf.py, the
    stand-in for scripts/cpu.sh) on all host cores and on one; bounded samples."""
    from oracle.torch_cpu_nerf import time_train_steps
    import multiprocessing
    cores = multiprocessing.cpu_count()
    legs = []
    # eager PyTorch on this many-core host is fastest well below the hardware thread count (small GEMMs): the 'all cores' leg is the
    # best of a few thread counts on a bounded sample
    best_t, best_v = None, 0.0
    for threads in sorted({cores, max(1, cores // 2), 64, 32}):
        if threads > cores:
            continue
        n, dt = time_train_steps(128, steps=1, threads=threads)
        if n / dt > best_v:
            best_t, best_v = threads, n / dt
    for threads, rays in ((best_t, rays_all), (1, 32)):   # 4096 rays = the n_rays of scripts/cpu.sh's configs/default.yaml
        n, dt = time_train_steps(rays, steps=1, threads=threads)
        legs.append({'value': n / dt, 'unit': 'samples/s', 'cores': threads, 'kind': 'port',
                     'sample': 'fwd+bwd+Adam of one config-1 step, {} rays = {} net evaluations, {:.1f} s, PyTorch CPU eager'.format(rays, n, dt)})
    best = dict(legs[0])
    best['threads_1'] = legs[1]
    return best


LAUNCH = 'single process'


def main():
    global LAUNCH
    args = parse()
    spawn_ranks_if_needed(args)
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        LAUNCH = 'bench.py --gpus N re-executed itself under torch.distributed.run' if os.environ.get('ARCN_BENCH_SELF_SPAWNED') else 'external torchrun'
    if args.config != 'ngp':
        return bench_module(args, args.config)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count())
    assert torch.cuda.is_available(), 'bench.py needs a GPU (there is no CPU fallback for the product path)'
    torch.cuda.set_device(local_rank)
    # the step's kernels go to a high-priority stream: the dispatcher then prefers their workgroups over those of the sampling
    # stream (marching of a later batch), which only fills what is left; -1 % on the step (ARCN_MAIN_PRIORITY=0: default stream)
    # Same for every world size (the collectives run on the communicator's own stream, between the backward and the optimiser pass that
    # waits for them: the only thing they share the chip with is the sampling stream's marcher).
    main_priority = int(os.environ.get('ARCN_MAIN_PRIORITY', '-1'))
    if main_priority != 0:
        torch.cuda.set_stream(torch.cuda.Stream(priority=main_priority))
    dev = torch.device('cuda', local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get('ARCN_DIST_FORCE', '0') == '1'   # ARCN_DIST_FORCE=1: a one-rank communicator runs the same path
    if use_dist:
        import torch.distributed as dist
        from arcnerf_amd import distributed as D
        # ARCN_DIST_BACKEND=gloo lets several ranks share one GPU for functional testing; the real runs use RCCL
        backend = os.environ.get('ARCN_DIST_BACKEND', 'nccl')
        D.init_from_env(backend=backend, device=dev if backend == 'nccl' else None)

    from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays

    cfg = NgpConfig()  # configs/models/nerf_ngp.yaml + nerf_lego_nerf_ngp.yaml
    field = NgpField(cfg, device=dev, seed=0)  # identical init on every rank (DDP broadcasts rank 0's, same effect)
    # the ray batches of a run are known in advance (the reference precaches and shuffles them on the GPU): march two steps ahead
    pipe = NgpPipeline(field, max_rays=32768, max_samples=1 << 20, packed_bits=True, prefetch_depth=2,
                       planned_scatter=os.environ.get('ARCN_PLANNED_SCATTER', '0') == '1',      # (=1: the planned scatter, the A/B of DESIGN.md)
                       fused_nets=os.environ.get('ARCN_FUSED_NETS', '1') != '0',                # (=0: the two nets' forward as two launches)
                       prefetch_at=(int(os.environ['ARCN_PREFETCH_AT']) if os.environ.get('ARCN_PREFETCH_AT') else None))   # (where the step queues the coming batch's marching)
    bf = synthetic_bitfield(cfg.n_grid, args.occupancy, seed=0)
    pipe.set_bitfield(torch.from_numpy(bf))

    # dynamic batch size (pipeline.py:222-241): n_rays such that the valid samples hit the allowance
    o, d = synthetic_rays(8192, seed=1000 + rank, device=dev)
    pipe.sample(o, d)
    per_ray = float(pipe.n_dev.item()) / 8192.0
    n_rays = int(min(32768, max(128, (int(args.target_samples / max(per_ray, 1e-3)) + 127) // 128 * 128)))
    n_pool = 8
    pool = []
    g = torch.Generator(device='cpu').manual_seed(77 + rank)
    for i in range(n_pool):
        o, d = synthetic_rays(n_rays, seed=10 * rank + i, device=dev)
        pool.append((o, d, torch.rand(n_rays, 3, generator=g).to(dev), torch.rand(n_rays, 3, generator=g).to(dev)))

    timers = KernelTimers()
    instrument(timers)
    # gradient sync of the N > 1 step, ARCN_GRAD_SYNC:
    #   flat (default, north_star's "single RCCL all-reduce of gradients"): ONE all-reduce of the flat buffer after the backward;
    #   levels: the scatter in level groups, each group's slice on the wire while the next is scattered, the optimiser per group as it
    #           arrives (distributed.LevelGroupedGradSync; ARCN_GRAD_LEVEL_CUTS=8 / =11,5);
    #   sharded: reduce-scatter, Adam + EMA on this rank's 1/N of the buffer, all-gather (distributed.ShardedGradSync).
    sync_mode = os.environ.get('ARCN_GRAD_SYNC', 'flat')
    if sync_mode not in ('flat', 'levels', 'sharded'):
        raise SystemExit('ARCN_GRAD_SYNC must be flat, levels or sharded')
    exposed_marks = []
    all_reduce, grad_sync, sync_name = None, None, None
    if use_dist:
        if sync_mode == 'levels' and pipe.level_major:
            cuts = tuple(int(v) for v in os.environ.get('ARCN_GRAD_LEVEL_CUTS', '8').split(',') if v.strip())
            grad_sync = D.LevelGroupedGradSync(field, cuts)
            grad_sync.timing = True
            spans = ['{}-{}'.format(min(l for l in range(32) if (m >> l) & 1), max(l for l in range(32) if (m >> l) & 1)) for m, _, _ in grad_sync.groups]
            sync_name = 'level groups {} overlapped with the scatter, optimiser per group'.format(spans)
        elif sync_mode == 'sharded':
            grad_sync = D.ShardedGradSync(field.n_params, world, rank)
            grad_sync.timing = True
            sync_name = 'reduce-scatter + optimiser on 1/{} of the buffer + all-gather'.format(world)
        else:
            def all_reduce(t):      # the flat form: everything between these two events is exposed
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                D.allreduce_grads(t, world)
                e1.record()
                exposed_marks.append((e0, e1))
            sync_name = 'one flat all-reduce'
    sample_log = torch.zeros(args.steps + args.warmup + 16, dtype=torch.int64, device=dev)

    def run(step_idx, epoch):
        o, d, tgt, bkg = pool[step_idx % n_pool]
        nxt = pool[(step_idx + pipe.prefetch_depth) % n_pool]
        # the marching of the batch `prefetch_depth` steps ahead is issued on a second stream during this step (next to the
        # optimiser pass at depth 2, next to the backward at depth 1); every step still marches exactly one batch
        pipe.train_step(o, d, tgt, bkg_color=bkg, all_reduce=all_reduce, world_size=world, next_rays=(nxt[0], nxt[1]),
                        grad_sync=grad_sync)
        # bookkeeping copy of this step's device-side sample count: on the sampling stream (which produced it), not in the
        # serial chain of the step's kernels
        with torch.cuda.stream(pipe.aux_stream):
            sample_log[step_idx] = pipe.n_dev[0]
        if not args.no_occ_update:
            pipe.update_occupancy(epoch, apply=False)

    epoch0 = 512  # steady-state regime of VolumeBound.optimize (after epoch_optim_warmup = 256)
    # a fresh box hands over an idle GPU: COLD_STEPS untimed steps of the workload before the W warmup steps (a multiple of the ray pool, so the prefetched
    # batches line up with warmup step 0): with W = 5 the warmup alone is 3 ms of device work on a box that has run nothing yet.
    # Reported as config.cold_start_steps.
    for i in range(COLD_STEPS):
        run(i % n_pool, epoch0 + 1 + (i % 15))     # (no occupancy refresh among them: that cadence belongs to the counted steps)
    torch.cuda.synchronize()
    # The FIRST process on a fresh box showed ONE 24-38 ms step inside the timed region (never in a second process on the same box).
    # The stall is on the HOST (ARCN_BENCH_TRACE=1: 38 ms inside one step's enqueue calls) and always at timed step 12-13, i.e. after
    # ~50 of the per-launch timing events of the timed region: the runtime's event pool growing for the first time on that box.  Grow
    # it here: as many events as the timed region will create, recorded once and released.
    _ev = [torch.cuda.Event(enable_timing=True) for _ in range(6 * args.steps + 512)]
    for e in _ev:
        e.record()
    torch.cuda.synchronize()
    del _ev
    for i in range(args.warmup):
        if i == max(0, args.warmup - 2):
            # the event brackets of the timed region are exercised in the last two warmup steps already: their first use in a process
            # (event pool, lazily loaded runtime pages on a fresh box) showed up as ONE 24 ms step inside the timed region
            timers.reset(ROOFLINE_KERNELS)
            torch.cuda.Event(enable_timing=True).record()
        run(i, epoch0 + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    timers.reset(ROOFLINE_KERNELS)
    torch.cuda.synchronize()
    step_events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    step_events[0].record()
    cpu_marks = [time.perf_counter()]
    for i in range(args.steps):
        run(args.warmup + i, epoch0 + args.warmup + i)
        step_events[i + 1].record()   # per-step spread (the driver's default K makes a 40 ms timed region)
        cpu_marks.append(time.perf_counter())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_step_raw = [step_events[i].elapsed_time(step_events[i + 1]) for i in range(args.steps)]
    per_step = sorted(per_step_raw)
    slowest = int(np.argmax(per_step_raw))
    every = cfg.epoch_optim
    refresh_next = [] if (args.no_occ_update or not every) else [i + 1 for i in range(args.steps - 1) if (epoch0 + args.warmup + i) % every == 0]
    if os.environ.get('ARCN_BENCH_TRACE'):   # where a slow step lost its time: GPU span vs host enqueue span per step
        print(json.dumps({'gpu_ms': [round(v, 3) for v in per_step_raw], 'host_ms': [round((b - a) * 1e3, 3) for a, b in zip(cpu_marks[:-1], cpu_marks[1:])]}), file=sys.stderr)

    samples = sample_log[args.warmup:args.warmup + args.steps].sum().reshape(1)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    per_rank = [int(samples.item()) / args.steps]
    rccl_extra = None
    if dist is not None:
        allr = [torch.zeros_like(samples) for _ in range(world)]
        dist.all_gather(allr, samples)
        per_rank = [int(t.item()) / args.steps for t in allr]
        dist.all_reduce(samples)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rccl_extra = time_allreduce_alone(dist, field.grads)
        rccl_extra['grad_sync'] = sync_name
        # the part of the exchange nothing hid: end of the last scatter group (flat form: of the backward) -> end of the last collective
        if grad_sync is not None and hasattr(grad_sync, 'exposed_ms'):
            ex = grad_sync.exposed_ms(last_n=args.steps)
        else:
            ex = [a.elapsed_time(b) for a, b in exposed_marks[-args.steps:]]
        if ex:
            rccl_extra['exposed_ms'] = sum(ex) / len(ex)
            rccl_extra['exposed_ms_max'] = max(ex)
    total_samples = int(samples.item())
    wall = float(tmax.item())
    ksum = timers.summary()
    n_launch = {k: len(v) for k, v in timers.pairs.items()}
    # per-kernel table: a few extra, untimed steps with every entry point bracketed (the step itself runs slower like this)
    ktable = None
    if world == 1:
        timers.reset(TABLE_KERNELS)
        timers.every = 1
        for i in range(min(16, args.steps)):
            run(args.warmup + args.steps + i, epoch0 + args.warmup + args.steps + i)
        torch.cuda.synchronize()
        ktable = timers.summary()
        timers.every = 4
    # the training gather on its own (nothing else on the device): in the step it runs next to the sampling stream's marcher
    alone_ms = alone_pts = None
    if world == 1:
        from arcnerf_amd import _native as NV
        import ctypes
        torch.cuda.synchronize()
        b, st, S = pipe.buf, NV.stream(), pipe.cap
        alone_pts = int(pipe.n_dev.item())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(24):
            if it == 4:
                e0.record()
            NV.check(NV.lib().arcn_hashgrid_fwd_xcd(NV.ptr(b['xyz']), NV.ptr(field.view('table')), ctypes.addressof(field.grid_desc),
                                                    NV.ptr(b['feat']), 1, S, S, pipe.n_dev.data_ptr(), st), 'hashgrid_fwd_xcd')
        e1.record()
        torch.cuda.synchronize()
        alone_ms = e0.elapsed_time(e1) / 20.0
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    s_per_launch = total_samples / world / args.steps
    # roofline of the dominant kernel (by measured time): hash-grid gather / scatter are HBM-bound random 8-byte accesses
    hash_kernels = {'hashgrid_fwd': BYTES_HASH_FWD, 'hashgrid_bwd': BYTES_HASH_BWD}
    dom = max(hash_kernels, key=lambda k: ksum.get(k, 0.0))
    # hashgrid_fwd is also launched by the occupancy refresh (different size): use the per-step average of its launches
    traffic, traffic_stale, step_bytes, mfma_busy = None, None, None, None
    pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(pmc):
        try:
            import hashlib
            rec = json.load(open(pmc))
            traffic = rec.get(dom)
            step_bytes, mfma_busy = rec.get('_step_total_bytes'), rec.get('_mfma_busy')
            # the counters were collected on a particular version of the kernels: say so when the sources have moved since
            shas = rec.get('_source_sha256')
            if shas is None:
                traffic_stale = True
            else:
                csrc = os.path.join(ROOT, 'arcnerf_amd', 'csrc')
                traffic_stale = any(hashlib.sha256(open(os.path.join(csrc, f), 'rb').read()).hexdigest() != h for f, h in shas.items())
        except Exception:
            traffic, traffic_stale = None, None
    dur_s = ksum[dom] * 1e-3
    # achieved = the kernel's ALGORITHMIC bytes per launch (2188 B per sample for the scatter, 1164 for the gather) / its
    # average launch duration.  On one GPU the scatter's consumer also applies the optimiser to the table levels it owns
    # (arcn_hashgrid_bwd_lm_adam: parameter + two moments in and out once, 24 B per fused parameter): those bytes are real traffic of the
    # same launch but not part of 8(d)'s figure - they are reported beside it (`with_fused_optimizer`), not in `frac`.
    ach = hash_kernels[dom] * s_per_launch / dur_s
    n_fused = 0
    fused_side = None
    if dom == 'hashgrid_bwd' and world == 1 and getattr(pipe, '_adam_rest', None) is not None:
        n_fused = field.n_params - sum(b_ - a_ for a_, b_ in pipe._adam_rest)
        ach2 = (hash_kernels[dom] * s_per_launch + 24.0 * n_fused) / dur_s
        fused_side = {'what': 'the same launch also runs Adam + EMA on the table levels its chunk owners hold: 24 B per fused parameter', 'fused_optimizer_params': n_fused,
                      'algorithmic_bytes_per_launch': {'scatter': hash_kernels[dom] * s_per_launch, 'fused_optimizer': 24.0 * n_fused},
                      'achieved': ach2 / 1e9, 'frac': ach2 / HBM_PEAK}
    step_s = wall / args.steps
    roofline = {'kernel': dom if not n_fused else 'hashgrid_bwd (binned scatter; its consumer also applies the optimiser to the levels it owns, see with_fused_optimizer)', 'bound': 'hbm', 'achieved': ach / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                'frac': ach / HBM_PEAK, 'traffic': traffic, 'avg_launch_ms': ksum[dom], 'launches_timed': n_launch.get(dom), 'launches_in_region': args.steps,
                'algorithmic_bytes_per_sample': hash_kernels[dom],
                # PMC counters cannot be collected inside a timed run: the figure is the per-launch HBM bytes of the committed
                # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (tools/profile_round.sh)
                'traffic_source': 'profiles/pmc_traffic.json' if traffic is not None else None, 'traffic_stale': traffic_stale,
                'traffic_frac': (traffic / dur_s / HBM_PEAK) if traffic else None,
                'with_fused_optimizer': fused_side,
                # the WHOLE step against the HBM roof: counter bytes of every kernel of a step / the step time.  Nothing in this step is
                # bandwidth-bound (the gather waits on L2-miss latency, the scatter on instruction issue and LDS atomics, the nets on MFMA
                # issue): this is the number that says so
                'hbm_step': {'bytes_per_step': step_bytes, 'frac': (step_bytes / step_s / HBM_PEAK) if step_bytes else None, 'source': 'profiles/pmc_traffic.json:_step_total_bytes', 'stale': traffic_stale},
                'mfma_busy': {'per_kernel': mfma_busy, 'of': 'exact-f32 MFMA issue slots (v_mfma_f32_16x16x4_f32, 157 TFLOP/s)', 'source': 'profiles/pmc_traffic.json:_mfma_busy', 'stale': traffic_stale} if mfma_busy else None}

    # The hash LOOKUP on its own (north_star names it): besides the algorithmic HBM accounting, the bound this gather actually
    # sits on.  Every 8-byte corner read of a hashed level drags one 128-byte line from the XCD's L2 into the CU's L1
    # (MI355X_MICROARCH.md: L2 ~34.5 TB/s aggregate): 16 levels x 8 corners x 128 B per sample is the line traffic if no
    # gather hit L1; the coarse levels do hit, which is why the kernel can finish faster than that bound.
    lookup = None
    if ksum.get('hashgrid_fwd'):
        pts = s_per_launch          # (only the step's own launches are bracketed: KernelTimers skips the refresh's side stream)
        sec = ksum['hashgrid_fwd'] * 1e-3
        l2_lines = 16 * 8 * 128.0 * pts
        lookup = {'kernel': 'hashgrid_fwd', 'bound': 'hbm', 'achieved': BYTES_HASH_FWD * pts / sec / 1e9, 'peak': HBM_PEAK / 1e9,
                  'unit': 'GB/s', 'frac': BYTES_HASH_FWD * pts / sec / HBM_PEAK, 'avg_launch_ms': ksum['hashgrid_fwd'],
                  'l2_line_bytes_upper': l2_lines, 'l2_peak_GBps': 34500.0, 'l2_line_frac_upper': l2_lines / sec / 34.5e12}
        if alone_ms:
            # same kernel, same batch, back to back with nothing beside it (in the step it shares the chip with the marcher of a
            # later batch on the sampling stream)
            lookup['alone_launch_ms'] = alone_ms
            lookup['alone_frac'] = BYTES_HASH_FWD * alone_pts / (alone_ms * 1e-3) / HBM_PEAK
            lookup['alone_l2_line_frac_upper'] = 16 * 8 * 128.0 * alone_pts / (alone_ms * 1e-3) / 34.5e12

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, field, bf, args.cpu_rays)
        # north_star: "the reference CPU path (scripts/cpu.sh) timed on the host cores of the same box in the same run" - that script
        # trains the vanilla-NeRF config (config 1) on the CPU; its PyTorch-eager stand-in (oracle/torch_cpu_nerf.py, pinned to G22)
        # rides along as a bounded leg (the full config-1 numbers come from `bench.py --config nerf`)
        try:
            cpu['scripts_cpu_sh'] = cpu_baseline_nerf(rays_all=1024)
        except Exception as e:   # never lose the headline line to the side leg
            cpu['scripts_cpu_sh'] = {'error': repr(e)}

    # BASELINE configs 1 / 3 / 4 / 5 through the module path, AFTER the headline measurement (which they cannot disturb): a few steps each,
    # so that the driver's record carries them too (the reference times every model: tests_benchmark.py:22-185).  `bench.py --config X`
    # is the full line of one of them.
    others = None
    if world == 1 and not args.no_other_configs:
        import copy
        others = {}
        timers.reset(())           # no event brackets around the module-path configs' launches
        # (the headline's step runs on a high-priority stream, above; these legs run the way `bench.py --config <name>` does and a trainer would:
        # on the default stream.  The drop-in NGP step marches ONE batch ahead on a second stream of the same priority - under a high-priority
        # main stream that chain starves and every step waits for it: 1.05 ms instead of 0.58)
        torch.cuda.set_stream(torch.cuda.default_stream())
        for name in (os.environ.get('ARCN_OTHER_CONFIGS', 'ngp_module,nerf,neus,neus_ngp_multivol,neus_ngp_nerfpp,hdrnerf').split(',')):
            a2 = copy.copy(args)
            # (24 steps after 6: the 8-after-3 legs of rounds 2 - 4 read the wide configs 1 - 3 % high - allocator growth and clocks still settling)
            a2.steps, a2.warmup, a2.rays, a2.chunk_pts, a2.no_cpu_baseline = 24, 6, 0, 0, True
            if name in ('nerf', 'neus', 'hdrnerf'):
                # the yaml's chunk_pts (4096 * 32 points per net evaluation) is the reference's memory knob for an 11 GB card; the chunks are
                # independent, so on 288 GB the whole batch goes through the nets in one or two of them: half the launches, 2 - 4.5 % off the
                # step (profiles/r6_exp_chunk_pts.txt); the value is reported as config.chunk_pts
                a2.chunk_pts = 1 << 20
            if name == 'ngp_module':    # the headline's model through the drop-in API: the driver's own K / W (+ the stepper's two eager steps)
                a2.steps, a2.warmup = min(args.steps, 2000), min(args.warmup, 500) + 2
            elif name in ('neus_ngp_multivol', 'neus_ngp_nerfpp'):     # 2 ms steps: eight of them are not a steady state (buffers still growing, 2.19 vs 1.96 ms stand-alone)
                a2.steps, a2.warmup = 32, 8
            try:
                # config 4's legs run in a CHILD process: three streams side by side (foreground, background, samplers) want three hardware
                # queues of their own, and in this process - six models built one after another, torch's stream pool handing their streams
                # round robin - they land on shared ones (1.59 ms here against 1.10 ms in a process that builds the one model, what a training run is)
                r = module_leg_child(name, a2.steps, a2.warmup) if name in ('neus_ngp_multivol', 'neus_ngp_nerfpp') else bench_module(a2, name, emit=False)
                others[name] = {'ms_per_step': r['ms_per_step'], 'ms_per_step_p50': (r['config'].get('step_ms_device') or {}).get('p50'), 'samples_per_s': r['value'], 'steps': a2.steps, 'warmup': a2.warmup,
                                'rays_per_step': r['config']['rays_per_step_per_gpu'], 'samples_per_step': r['config']['samples_per_step_per_gpu'], 'chunk_pts': r['config'].get('chunk_pts'),
                                'roofline_frac': (r['roofline'] or {}).get('frac_of_split_peak', (r['roofline'] or {}).get('frac')), 'roofline_peak': 'dense bf16 MFMA / 6 terms (417 TFLOP/s of f32-accurate work)' if 'frac_of_split_peak' in (r['roofline'] or {}) else 'HBM 8 TB/s',
                                'roofline_frac_of_f32_mfma_peak': (r['roofline'] or {}).get('frac') if 'frac_of_split_peak' in (r['roofline'] or {}) else None, 'roofline_bound': (r['roofline'] or {}).get('bound'), 'bkg_samples_per_step': (r['roofline'] or {}).get('bkg_samples_per_step'),
                                'workload': r['config']['workload']}
            except Exception as e:      # never lose the headline line to a side leg
                others[name] = {'error': repr(e)}
            torch.cuda.empty_cache()
        try:
            others['inference'] = inference_leg(dev, args.occupancy)
        except Exception as e:
            others['inference'] = {'error': repr(e)}
        torch.cuda.empty_cache()
        # the N > 1 step at world 1 (children of this process, after everything timed here): `flat` = north_star's single all-reduce, `sharded` =
        # reduce-scatter + 1/N optimiser + all-gather; 'delta_ms_to_fused_step' = what the exchange-shaped step costs a GPU before any wire time
        for sync in ('flat', 'sharded'):
            try:
                leg = dist1_leg(sync, args.steps, args.warmup)
                if 'ms_per_step' in leg:
                    leg['delta_ms_to_fused_step'] = leg['ms_per_step'] - wall / args.steps * 1e3
                others['ngp_dist1_' + sync] = leg
            except Exception as e:
                others['ngp_dist1_' + sync] = {'error': repr(e)}
        # BASELINE config 4 - the 8-GPU config - likewise: trainer.FusedNeusNgpStep(world_size, grad_sync='flat') on the one-rank communicator
        # (scatters -> ONE all-reduce of the 97.6 MB flat gradient -> one optimiser pass) against the fused single-GPU step of the leg above
        try:
            leg = dist1_leg('flat', 32, 8, config='neus_ngp_multivol')
            if 'ms_per_step' in leg and 'ms_per_step' in others.get('neus_ngp_multivol', {}):
                leg['delta_ms_to_fused_step'] = leg['ms_per_step'] - others['neus_ngp_multivol']['ms_per_step']
            others['neus_ngp_multivol_dist1_flat'] = leg
        except Exception as e:
            others['neus_ngp_multivol_dist1_flat'] = {'error': repr(e)}

    # PSNR@iter, the second half of BASELINE's metric, with the reference's RECIPE through the drop-in API (tools/psnr_recipe.py): no dataset
    # on the box, so the scene is analytic - six soft textured blobs rendered once to 100 training views of 320 x 320 RGBA bytes + 4 held
    # out - and the loop is the one golden G27 pins to a run of the reference's own loop (tests/test_gpu_psnr.py): build_model(nerf_ngp.yaml)
    # + trainer.train_epoch + trainer.FusedNgpStep + trainer.TrainBatches on a trainer.Pipeline with the Lego yaml's scheduler (centre
    # precrop 0.5 / 500 iterations, random background colours, cross-view shuffle, dynamic batch size).  A short run rides along after
    # everything that is timed.
    psnr = None
    if world == 1 and not args.no_other_configs and not args.no_psnr:
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location('psnr_recipe', os.path.join(ROOT, 'tools', 'psnr_recipe.py'))
            pc = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(pc)
            timers.reset(())
            torch.cuda.empty_cache()
            r = pc.run(3000, seed=0, verbose=False, window=(2488, 2988))
            psnr = {'metric': 'PSNR@iter, scene: analytic (NOT Lego - there is no dataset on the box; not comparable with docs/benchmark.md:50-54)',
                    'scene': r['scene'] + ' (analytic: there is no dataset on the box)', 'recipe': r['recipe'], 'path': r['path'], 'seed': 0,
                    'psnr_at_iter': {str(p_['iter']): round(p_['psnr'], 2) for p_ in r['points']},
                    'train_seconds_at_iter': {str(p_['iter']): round(p_['train_seconds'], 2) for p_ in r['points']},
                    'rays_per_step_at_iter': {str(p_['iter']): p_['rays_per_step'] for p_ in r['points']},
                    'occupied_at_iter': {str(p_['iter']): round(p_['occupied'], 4) for p_ in r['points']},
                    'data_seconds': round(r['data_seconds'], 2),
                    # the same loop, iterations 2488 - 2988, between two device synchronisations: every refresh of the occupancy grid APPLIED, a fresh
                    # shuffled batch against the scene's pixels every step, dynamic batch size - the headline's timed loop computes its refreshes
                    # without applying them (a random-init field has no stationary occupancy) and cycles eight ray batches against random targets
                    'training_loop': r.get('training_loop'),
                    'anchor': 'golden G27 (tests/golden/make_golden_psnr.py): the reference\'s own loop + Pipeline, 600 iterations x 4 seeds on 100 x 100 views reach '
                              '33.6 - 34.3 dB; the module path reproduces every seed to 0.02 - 0.5 dB at 50 / 100 / 200 / 400 / 600 iterations, this '
                              'stepper stays inside the seed band (tests/test_gpu_psnr.py); profiles/r5_psnr_recipe.json: three seeds of this leg'}
        except Exception as e:
            psnr = {'error': repr(e)}

    out = {
        'metric': 'ray-samples/sec (train), NGP Lego 800x800', 'value': total_samples / wall, 'unit': 'samples/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': wall / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'instant-ngp (hashgrid L16 F2 T2^19 + fused MLP 32-64-16 / 32-64-64-3 + volume prune 128^3), '
                               'Blender-Lego-like 800x800 rays, {} rays/step/GPU, ~{:.0f} valid samples/step/GPU, '
                               'occupancy {:.0%}, occupancy refresh every 16 steps{}'.format(
                                   n_rays, s_per_launch, args.occupancy, ' (off)' if args.no_occ_update else ''),
                   'rays_per_step_per_gpu': n_rays, 'samples_per_step_per_gpu': s_per_launch, 'cold_start_steps': COLD_STEPS,
                   'parallelism': 'ray-sharded dp{}'.format(world)},
        'timed_region_ms': wall * 1e3,
        # the steps behind an occupancy refresh (VolumeBound.optimize's cadence, cfg.epoch_optim = 16: the refresh of step i is queued on its own stream
        # and shares the chip with step i + 1 - ~0.3 ms of geometry-net work on 2^19 cells every sixteenth step) are the slow ones: part of the
        # workload, inside `value`; 'max_other' is the slowest step that has no refresh beside it
        'step_ms_spread': {'slowest_step': slowest, 'min': per_step[0], 'p50': per_step[len(per_step) // 2], 'p90': per_step[min(len(per_step) - 1, int(0.9 * len(per_step)))],
                           'max': per_step[-1], 'steps_beside_a_refresh': refresh_next,
                           'p50_beside_a_refresh': (sorted(per_step_raw[i] for i in refresh_next)[len(refresh_next) // 2] if refresh_next else None),
                           'max_other': max([v for i, v in enumerate(per_step_raw) if i not in refresh_next] or [None])},
        'rccl': dist_report(dist, world, LAUNCH, field.n_params * 4, (len(grad_sync.groups) if hasattr(grad_sync, 'groups') else 1 + len(grad_sync.segments)) if grad_sync is not None else 1, per_rank, rccl_extra),
        'roofline': roofline,
        'roofline_lookup': lookup,
        'cpu_baseline': cpu,
        'kernel_ms': ktable,
        'other_configs': others,
        'psnr_analytic_scene': psnr,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(cfg, field, bf, n_rays):
    """The CPU oracle (C restatement of the reference path, OpenMP) timed on the host cores on a bounded sample of the
    same workload: fwd + bwd of one NGP training step, on all host cores (about 10-20 s of CPU work) and on ONE thread (a smaller
    sample); BASELINE.md section 3."""
    from oracle import oracle as orc
    from oracle.ngp_reference import oracle_train_step
    from arcnerf_amd.pipeline import synthetic_rays
    orc.build()
    cores = orc.get_max_threads()
    P = field.export_numpy()
    legs = []
    from oracle.ngp_reference import oracle_train_step_sharded
    # all cores: the batch in ray shards run concurrently (each shard: the C kernels on its OpenMP threads + its own numpy glue), gradients
    # summed at the end - the single-call form keeps the glue (mask compaction, padded scatter) on ONE core and stopped at ~3x one thread
    shards = max(1, min(32, cores // 4))
    for threads, rays in ((cores, n_rays), (1, max(256, n_rays // 64))):
        o, d = synthetic_rays(rays, seed=4242, device='cpu')
        o, d = o.numpy(), d.numpy()
        rng = orc.Pcg32(9121)
        t0 = time.perf_counter()
        if threads > 1:
            n, _ = oracle_train_step_sharded(orc, field, cfg, P, o, d, bf, rng.state, rng.inc, shards, max(1, threads // shards))
            how = '{} ray shards x {} OpenMP threads, gradients summed'.format(shards, max(1, threads // shards))
        else:
            orc.set_num_threads(1)
            n = oracle_train_step(orc, field, cfg, P, o, d, bf, rng.state, rng.inc)
            how = 'one call'
        dt = time.perf_counter() - t0
        legs.append({'value': n / dt, 'unit': 'samples/s', 'cores': threads, 'kind': 'port',
                     'sample': 'fwd+bwd of one NGP training step for {} rays = {} valid samples ({:.1f} s), C oracle with OpenMP + numpy glue '
                               '({})'.format(rays, n, dt, how)})
    orc.set_num_threads(cores)
    out = dict(legs[0])
    out['threads_1'] = legs[1]
    return out


if __name__ == '__main__':
    main()
