"""Compacted-sample NGP pipeline: the MI355X fast path of the reference's instant-ngp configuration
(configs/models/nerf_ngp.yaml + configs/expr/NeRF/lego/nerf_lego_nerf_ngp.yaml).

One training step = the reference call stack of SURVEY.md §3.1 without its dense (R, 1024) tensors and host syncs:

    rays --march_count/scan/write--> (offsets[R+1], t[S], ray_id[S])            FgModel.forward :178-186 (K2 + K3)
         --packed_points-->           xyz[S,3], dirs[S,3]                        get_sigma_radiance_by_mask_pts :283-292
         --hashgrid_fwd-->            feat[S,32]                                 HashGridEmbedder.forward
         --mlp_fwd (geo)-->           geo_out[S,16]                              FusedMLPGeoNet.forward
         --ngp_glue_fwd-->            sigma[S] = exp(geo_out[:,0]), rad_in[S,32] = [geo_out, SH(dir)]
         --mlp_fwd (rad)-->           rgb[S,3] (sigmoid)                         FusedMLPRadianceNet.forward
         --composite_packed_fwd-->    rgb/depth/mask per ray                     Base3dModel.ray_marching
         <-- loss (Huber) / composite_packed_bwd / mlp_bwd / ngp_glue_bwd / mlp_bwd / hashgrid_bwd
         [all-reduce of the flat gradient buffer when world_size > 1]
         --adam_ema_step-->           params (Adam + the reference's write-back EMA), gradient cleared

The sample count S lives on the device (offsets[R]); every buffer is pre-allocated at capacity and every kernel takes the
device-side count, so the whole step issues no host synchronisation.  All parameters live in ONE flat fp32 buffer
[hash table | geo W | radiance W] so the optimiser is one kernel and data-parallel training needs one collective.
"""
import math

import numpy as np
import torch

from . import _native as N
from .ops import functional as F
from .utils import param_epoch


class NgpConfig:
    """Hyper-parameters of configs/models/nerf_ngp.yaml (defaults) — field names follow the yaml keys."""

    def __init__(self, **kw):
        # encoder (HashGridEmbedder)
        self.n_levels = 16
        self.n_feat_per_entry = 2
        self.hashmap_size = 19
        self.base_res = 16
        self.max_res = 2048
        self.side = 2.0
        self.origin = (0.0, 0.0, 0.0)
        # geometry net (FusedMLPGeoNet): W, D hidden layers, W_feat outputs, sigma = TruncExp(out[:,0])
        self.geo_W = 64
        self.geo_D = 1
        self.W_feat = 16
        self.geo_fused_semantics = True  # True: feat = whole output (tcnn module); False: out = [sigma | feat] (GeoNet)
        self.sigma_act = 'truncexp'
        # radiance net (FusedMLPRadianceNet, mode 'fv'): SH degree 4 view encoding
        self.rad_W = 64
        self.rad_D = 2
        self.sh_degree = 4
        self.rad_mode = 'fv'
        self.has_bias = False
        # obj_bound.volume + rays
        self.n_grid = 128
        self.n_sample = 1024
        self.near_distance = 0.2
        self.add_inf_z = False
        self.white_bkg = False
        self.noise_std = 1.0
        self.opa_thres = 0.01
        self.ema_optim_decay = 0.95
        self.epoch_optim = 16
        self.epoch_optim_warmup = 256
        # optimiser block of nerf_lego_nerf_ngp.yaml
        self.lr = 1e-1
        self.eps = 1e-15
        self.weight_decay = 1e-6
        self.betas = (0.9, 0.999)
        self.ema_decay = 0.95
        self.huber_delta = 0.1
        self.loss_weight = 3000.0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise KeyError(k)
            setattr(self, k, v)

    @property
    def dt(self):
        """const_dt = volume.get_diag_len() / n_pts (volume_bound.py:112)"""
        return float(np.float32(math.sqrt(3.0 * self.side * self.side)) / np.float32(self.n_sample))


def hashgrid_level_table(n_levels, hashmap_size, base_res, max_res):
    """HashGridEmbedder.init_embeddings (hashgrid_encoder.py:126-158): per_level_scale is a fp32 torch scalar."""
    pls = torch.exp((torch.log(torch.tensor(max_res / base_res))) / (float(n_levels) - 1))
    res, offs, total = [], [], 0
    for i in range(n_levels):
        offs.append(total)
        r = math.ceil(2 ** (i * math.log2(pls)) * base_res - 1.0)
        res.append(r)
        total += min(2 ** hashmap_size, (r + 1) ** 3)
    offs.append(total)
    return res, offs


class NgpField:
    """Parameters of the NGP radiance field in one flat buffer + the kernel descriptors."""

    def __init__(self, cfg, device='cuda', seed=0):
        self.cfg = cfg
        self.device = torch.device(device)
        res, offs = hashgrid_level_table(cfg.n_levels, cfg.hashmap_size, cfg.base_res, cfg.max_res)
        self.resolutions, self.offsets = res, offs
        half = cfg.side / 2.0
        self.min_xyz = [cfg.origin[k] - half for k in range(3)]
        self.max_xyz = [cfg.origin[k] + half for k in range(3)]
        self.grid_desc = N.make_hashgrid_desc(res, offs, cfg.n_feat_per_entry, self.min_xyz, self.max_xyz)
        enc_dim = cfg.n_levels * cfg.n_feat_per_entry
        self.geo_out_dim = cfg.W_feat if cfg.geo_fused_semantics else 1 + cfg.W_feat
        self.feat_off = 0 if cfg.geo_fused_semantics else 1
        geo_dims = [enc_dim] + [cfg.geo_W] * cfg.geo_D + [self.geo_out_dim]
        rad_in = cfg.W_feat + cfg.sh_degree ** 2
        rad_dims = [rad_in] + [cfg.rad_W] * cfg.rad_D + [3]
        self.geo_dims, self.rad_dims = geo_dims, rad_dims
        self.geo_desc = N.make_mlp_desc(geo_dims, 'relu', None, has_bias=cfg.has_bias)
        self.rad_desc = N.make_mlp_desc(rad_dims, 'relu', 'sigmoid', has_bias=cfg.has_bias)
        # flat layout, every segment a multiple of 4 floats (16-byte aligned views for the fused optimiser)
        def pad4(n):
            return (n + 3) // 4 * 4
        self.n_table = offs[-1] * cfg.n_feat_per_entry
        self.n_geo_w = sum(geo_dims[i] * geo_dims[i + 1] for i in range(len(geo_dims) - 1))
        self.n_rad_w = sum(rad_dims[i] * rad_dims[i + 1] for i in range(len(rad_dims) - 1))
        self.n_geo_b = sum(geo_dims[1:]) if cfg.has_bias else 0
        self.n_rad_b = sum(rad_dims[1:]) if cfg.has_bias else 0
        self._seg = {}
        off = 0
        for name, n in (('table', self.n_table), ('geo_w', self.n_geo_w), ('rad_w', self.n_rad_w), ('geo_b', self.n_geo_b),
                        ('rad_b', self.n_rad_b)):
            self._seg[name] = (off, n)
            off += pad4(n)
        self.n_params = off
        self.params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.reset_parameters(seed)

    def view(self, name, buf=None):
        off, n = self._seg[name]
        if n == 0:
            return None
        return (self.params if buf is None else buf)[off:off + n]

    def reset_parameters(self, seed=0):
        """table ~ U(-1e-4, 1e-4) (hashgrid_encoder.py:155-156); dense layers: torch.nn.Linear default init."""
        g = torch.Generator(device='cpu').manual_seed(seed)
        cfg = self.cfg
        self.view('table').copy_((torch.rand(self.n_table, generator=g) * 2e-4 - 1e-4).to(self.device))
        for name, dims in (('geo', self.geo_dims), ('rad', self.rad_dims)):
            ws, bs = [], []
            for i in range(len(dims) - 1):
                bound = 1.0 / math.sqrt(dims[i])
                ws.append((torch.rand(dims[i + 1] * dims[i], generator=g) * 2 - 1) * bound)
                bs.append((torch.rand(dims[i + 1], generator=g) * 2 - 1) * bound)
            self.view(name + '_w').copy_(torch.cat(ws).to(self.device))
            if cfg.has_bias:
                self.view(name + '_b').copy_(torch.cat(bs).to(self.device))

    # ---- numpy export for the oracle-based checks ------------------------------------------------
    def export_numpy(self):
        out = {'table': self.view('table').detach().cpu().numpy().reshape(-1, self.cfg.n_feat_per_entry)}
        for name, dims in (('geo', self.geo_dims), ('rad', self.rad_dims)):
            w = self.view(name + '_w').detach().cpu().numpy()
            b = self.view(name + '_b').detach().cpu().numpy() if self.cfg.has_bias else None
            layers, o, ob = [], 0, 0
            for i in range(len(dims) - 1):
                n = dims[i] * dims[i + 1]
                layers.append((w[o:o + n].reshape(dims[i + 1], dims[i]), None if b is None else b[ob:ob + dims[i + 1]]))
                o += n
                ob += dims[i + 1]
            out[name] = layers
        return out


class StepLoss:
    """Loss of one training step, reduced on first use from the per-workgroup partial sums the fused compositor stored (nothing is
    launched for it during the step).  float(loss) / loss.item() / loss.tensor(); valid while its ring slot is (64 steps)."""

    def __init__(self, pipe, slot, generation, n_wgs):
        self._pipe, self._slot, self._generation, self._n, self._value = pipe, slot, generation, n_wgs, None

    def tensor(self):
        if self._value is None:
            if self._pipe._loss_step[self._slot] != self._generation:
                raise RuntimeError('this step\'s loss partials have been overwritten (read a loss within 64 steps of its step)')
            self._value = self._pipe.buf['loss_ring'][self._slot, :self._n].sum()
        return self._value

    def item(self):
        return float(self.tensor().item())

    __float__ = item

    def detach(self):
        return self.tensor().detach()

    def __repr__(self):
        return 'StepLoss({})'.format(self.item())


class NgpPipeline:
    """Pre-allocated buffers + the kernel sequence of one render / train step for a fixed ray capacity."""

    def __init__(self, field, max_rays=32768, max_samples=1 << 19, packed_bits=True, torch_aabb=False, xcd_scatter=True, level_major=True, fused_glue=True,
                 prefetch_depth=None, prefetch_at=None, march_waves=None, aux_priority=None, occ_async=True, fused_composite=True, fuse_adam=True,
                 step_tail=True, march_cull=True, planned_scatter=False, fused_nets=True):
        """The keyword switches select the measured alternatives of the step's schedule (defaults = the product path): prefetch_depth
        batches marched ahead, prefetch_at = where in the step the next marching is issued (_prefetch_point), march_waves = persistent
        wavefronts of a marching launch with slack, aux_priority = priority of the sampling stream, occ_async = the occupancy refresh on its
        own stream, fused_composite = compositing + loss + their backward as one kernel, fuse_adam = the scatter's chunk owners apply the
        optimiser, step_tail = the end of the step as one launch, march_cull = the marcher's ray-culling grid, planned_scatter = the
        position-only half of the table scatter (arcn_hashgrid_bwd_plan) computed with a batch marched ahead, on the sampling stream (OFF:
        the scatter's own bracket drops from 0.178 to 0.145 ms, but on one GPU the plan pass shares the chip with the step's kernels and the
        step is 0.655 against 0.540 ms - DESIGN.md; it pays where the sampling stream has idle compute units beside it), fused_nets = both
        nets' forward as one kernel (arcn_ngp_nets_fwd: bit-identical to the two launches it replaces)."""
        cfg = field.cfg
        self.field, self.cfg = field, cfg
        dev = field.device
        self.max_rays, self.cap = int(max_rays), int(max_samples)
        self.packed_bits = packed_bits
        self.torch_aabb = torch_aabb
        f32, i32 = torch.float32, torch.int32
        S, R = self.cap, self.max_rays
        self.aabb23 = torch.tensor([field.min_xyz, field.max_xyz], dtype=f32, device=dev)
        self.rng = F.Pcg32Host(9121)
        E = cfg.n_levels * cfg.n_feat_per_entry
        b = self.buf = {}
        # sample buffers exist twice: the marcher of step i+1 (it depends only on rays + occupancy) can run on a second
        # HIP stream while step i's backward is still scattering (prefetch_samples)
        # prefetch_depth 2: three sets, two batches in flight - train_step(next_rays=) is then
        # given the rays of step i+2 and marches them next to step i's optimiser pass (pure HBM streaming, idle VALUs) instead of next
        # to the backward kernels; a refreshed occupancy takes effect two steps later instead of one
        self.prefetch_depth = max(1, int(prefetch_depth or 1))
        self.march_cull = bool(march_cull)
        self._sets = []
        for _ in range(1 + self.prefetch_depth):
            self._sets.append({
                'scratch_t': torch.empty((R, cfg.n_sample), dtype=f32, device=dev),
                'counts': torch.zeros(R, dtype=i32, device=dev), 'offsets': torch.zeros(R + 1, dtype=i32, device=dev),
                'near': torch.empty(R, dtype=f32, device=dev), 'far': torch.empty(R, dtype=f32, device=dev),
                't': torch.zeros(S, dtype=f32, device=dev), 'ray_id': torch.zeros(S, dtype=i32, device=dev),
                'p_dense': torch.full((1,), 2, dtype=i32, device=dev),
                # everything else that depends on the rays and the occupancy only is produced with the samples (second stream
                # when prefetched): sample positions / directions, per-ray harmonics, the density noise of a training step
                'xyz': torch.zeros((S, 3), dtype=f32, device=dev), 'dirs': torch.zeros((S, 3), dtype=f32, device=dev),
                'sh_ray': torch.zeros((R, max(1, cfg.sh_degree ** 2)), dtype=f32, device=dev),
                'noise': torch.zeros(S, dtype=f32, device=dev)})
        self._noise_ready = [False] * len(self._sets)
        b.update(self._sets[0])
        self._cur_set = 0
        self._prefetched = []   # FIFO of (rays_o ptr, rays_d ptr, R, set index, event), oldest first
        self.sampled_ahead = False     # the batch of the last sample() was marched ahead on the sampling stream (its count was produced there)
        self.sample_event = None       # ... or inline on the caller's stream: recorded behind its marcher
        self._next_rays = None
        self._carry_rays = None  # prefetch point 5: rays handed over by the previous train_step, marched behind this step's gather
        self.occ_stream = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
        self.occ_async = bool(occ_async) and self.occ_stream is not None
        self._occ_params_event = None   # the refresh still reads the parameters: the optimiser waits for it
        self._occ_bits_event = None     # the refreshed bitfield is ready: the marcher waits for it
        self._occ_state_event = None    # last refresh finished: readers of .bitfield / .opafield wait for it
        self.prefetch_at = int(prefetch_at) if prefetch_at is not None else (1 if self.prefetch_depth == 1 else 3)
        # two batches ahead the marching has a whole step of slack: it runs as 4096 PERSISTENT wavefronts (arcn_march_count_waves: wave w takes
        # the rays w, w + 4096, ...) - half as many long-lived marcher waves on every SIMD beside the step's kernels, for twice as long: the
        # step loses 2 % less to them (0.569 -> 0.556 - 0.558 ms, three alternations; 3072: the same, 2048 / 5120 / 6144: less, 1024: the chain
        # is late and the step waits).  One batch ahead (or inline) the chain's latency is on the clock: +3 %, so not there.
        self.march_waves = int(march_waves) if march_waves is not None else (4096 if self.prefetch_depth >= 2 else 0)
        # multi-rank: the compute units idle while the gradient all-reduce is on the wire - march the next batch there
        self.prefetch_at_dist = 3
        self._prefetch_now = self.prefetch_at
        self.aux_stream = None
        if dev.type == 'cuda':
            # ONE batch ahead, the marching chain has exactly the backward of the current step to finish in: on a lower priority than the
            # step's stream it only gets what the step's kernels leave, and every step then waits for it (measured under a priority -1
            # current stream: 1.05 ms per step against 0.58) - so it takes the priority of the stream the pipeline is built on.  Two
            # batches ahead it has a whole step of slack, and a lower priority is what one wants (bench.py's headline: -1 %).
            prio = torch.cuda.current_stream(dev).priority if self.prefetch_depth == 1 else 0
            prio = int(aux_priority) if aux_priority is not None else prio
            self.aux_stream = torch.cuda.Stream(device=dev, priority=prio) if prio != 0 else torch.cuda.Stream(device=dev)
        self.use_streams = self.aux_stream is not None
        b['feat'] = torch.zeros((S, E), dtype=f32, device=dev)
        b['geo_out'] = torch.zeros((S, field.geo_out_dim), dtype=f32, device=dev)
        b['geo_acts'] = torch.zeros(max(1, F.mlp_acts_floats(field.geo_desc, S)), dtype=f32, device=dev)
        b['rad_in'] = torch.zeros((S, field.rad_dims[0]), dtype=f32, device=dev)
        b['sigma'] = torch.zeros(S, dtype=f32, device=dev)
        b['rgb_s'] = torch.zeros((S, 3), dtype=f32, device=dev)
        b['rad_acts'] = torch.zeros(max(1, F.mlp_acts_floats(field.rad_desc, S)), dtype=f32, device=dev)
        # backward
        b['d_sigma'] = torch.zeros(S, dtype=f32, device=dev)
        b['d_rgb_s'] = torch.zeros((S, 3), dtype=f32, device=dev)
        b['d_rad_in'] = torch.zeros((S, field.rad_dims[0]), dtype=f32, device=dev)
        b['d_geo_out'] = torch.zeros((S, field.geo_out_dim), dtype=f32, device=dev)
        b['d_feat'] = torch.zeros((S, E), dtype=f32, device=dev)
        # backward scratch per network: per-workgroup dW partials (and dpre when the two-kernel backward runs)
        b['geo_scratch'] = torch.zeros(F.mlp_scratch_floats(field.geo_desc, S), dtype=f32, device=dev)
        b['rad_scratch'] = torch.zeros(F.mlp_scratch_floats(field.rad_desc, S), dtype=f32, device=dev)
        # per-ray outputs
        b['rgb'] = torch.zeros((R, 3), dtype=f32, device=dev)
        b['depth'] = torch.zeros(R, dtype=f32, device=dev)
        b['mask'] = torch.zeros(R, dtype=f32, device=dev)
        b['d_rgb'] = torch.zeros((R, 3), dtype=f32, device=dev)
        b['loss'] = torch.zeros(1, dtype=f32, device=dev)
        # fused compositor: every workgroup (4 rays) stores its share of the step's loss; the scalar is summed only when somebody
        # reads it (StepLoss).  A ring of slots keeps the last 64 steps' partials readable.
        self._loss_wgs = (R + 3) // 4
        b['loss_ring'] = torch.zeros((64, self._loss_wgs), dtype=f32, device=dev)
        self._loss_slot = 0
        self._loss_step = [-1] * 64
        self.fused_composite = bool(fused_composite)
        # XCD-owned-levels scatter workspace (owner + tile counters); None selects the plain agent-scope kernel
        self.hash_ws = F.hashgrid_bwd_workspace(self.field.grid_desc, S, dev) if xcd_scatter else None  # scatter bins
        # single-GPU step: the optimiser of the table levels whose chunks have ONE owner is applied by that owner inside the scatter
        # (arcn_hashgrid_bwd_lm_adam); `_adam_rest` = the slices of the flat buffer the plain kernel still has to visit
        self._adam_rest = None
        self._fused_step = False
        self._fuse_next = False
        if xcd_scatter and self.hash_ws is not None and fuse_adam and field.n_params > 0:
            self._adam_rest = self._plan_fused_adam(S)
        # level-major features between the hash grid and the geometry net (XCD-affine gather, coalesced everywhere): the shapes
        # the *_lm entry points are wired for; anything else keeps the row-major buffers
        self.ray_sh = (cfg.sh_degree >= 1 and field.feat_off == 0 and field.geo_out_dim % 4 == 0 and cfg.W_feat % 4 == 0 and
                       (cfg.sh_degree ** 2) % 4 == 0)
        rd = [field.rad_desc.dims[i] for i in range(field.rad_desc.n_layers + 1)]
        self.fused_glue = bool(fused_glue and self.ray_sh and field.geo_out_dim == 16 and cfg.W_feat == 16 and cfg.sh_degree == 4 and
                               not field.rad_desc.has_bias and field.rad_desc.n_layers in (2, 3) and rd[0] == 32 and
                               all(48 < w <= 64 for w in rd[1:-1]) and rd[-1] <= 16)
        gd = [field.geo_desc.dims[i] for i in range(field.geo_desc.n_layers + 1)]
        self._want_fused_nets = bool(fused_nets)
        self.level_major = bool(level_major and xcd_scatter and cfg.n_feat_per_entry == 2 and field.geo_desc.n_layers == 2 and
                                not field.geo_desc.has_bias and gd[0] in (32, 64) and 48 < gd[1] <= 64 and gd[2] <= 16)
        gdsc, rdsc = field.geo_desc, field.rad_desc
        self.fused_nets = bool(self._want_fused_nets and self.level_major and self.fused_glue and gd == [32, 64, 16] and rd[:3] == [32, 64, 64] and
                               len(rd) == 4 and gdsc.act_hidden == N.ACT['relu'] and gdsc.act_out == N.ACT[None] and
                               rdsc.act_hidden == N.ACT['relu'] and rdsc.act_out == N.ACT['sigmoid'])
        # planned scatter: a batch marched AHEAD (prefetch_samples, sampling stream) also gets the scatter's position-only half - cells, runs,
        # rows, bins, ranks, the records' index halves - into a plan workspace of its own (one per sample-buffer set); the step then runs the
        # fill pass + the chunk owners (arcn_hashgrid_bwd_lm[_adam]_planned).  A batch sampled inline keeps the one-pass scatter.
        self._plan_floats = 0
        if planned_scatter and self.level_major and self.hash_ws is not None and self.use_streams and not F.deterministic():
            self._plan_floats = int(N.lib().arcn_hashgrid_plan_workspace_floats(N.C.addressof(field.grid_desc), int(S)))
        for st_ in self._sets:
            st_['plan_ws'] = torch.empty(self._plan_floats, dtype=f32, device=dev) if self._plan_floats else None
        self._planned = [False] * len(self._sets)
        # single-GPU step with the optimiser inside the scatter: the two dW reductions, the optimiser on the rest of the flat buffer and
        # the clearing of the scatter's bin counters as ONE launch at the end of the step (arcn_ngp_step_tail; step_tail=False: four)
        self._tail = None
        self._tail_step = False
        self._ws_clear = False
        if self._adam_rest is not None and self.level_major and self.fused_glue and not cfg.has_bias and step_tail:
            gw, rw = field._seg['geo_w'], field._seg['rad_w']
            inside = lambda run, seg: seg[0] >= run[0] and seg[0] + seg[1] <= run[1]
            rest, ok = [], True
            for run in self._adam_rest:     # the weight segments leave the plain optimiser's runs (they sit at the end of the last one)
                segs = sorted(sg for sg in (gw, rw) if inside(run, sg))
                lo = run[0]
                for sg in segs:
                    if sg[0] > lo:
                        rest.append((lo, sg[0]))
                    lo = sg[0] + sg[1]
                if run[1] > lo:
                    rest.append((lo, run[1]))
            ok = (sum(1 for run in self._adam_rest for sg in (gw, rw) if inside(run, sg)) == 2 and len(rest) <= 4 and
                  all(lo % 4 == 0 for lo, _ in rest))
            if ok:
                self._tail = {'runs': rest, 'geo_w': gw[0], 'rad_w': rw[0],
                              'clear_words': int(N.lib().arcn_hashgrid_bwd_counter_words(N.C.addressof(field.grid_desc), int(S)))}
                # the tail launch is pure load latency (13 us alone, 50 us with the marcher's waves resident next to it): the next batch's
                # marching is issued behind it, not in front (0.607 against 0.609 ms per step, 0.618 with the four launches)
                if prefetch_at is None and self.prefetch_depth != 1:
                    self.prefetch_at = 4
        # optimiser state
        n = field.n_params
        self.exp_avg = torch.zeros(n, dtype=f32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=f32, device=dev)
        # EMA shadow (ema.py's old_avg): equal to the parameters after every step because the average is written back, and this
        # pipeline is the only writer of the flat buffer -> the shadow IS the buffer (arcn_adam_ema_step with ema == param); a caller
        # that keeps a separate shadow (FusedAdam without ema_in_param) assigns its buffer to `.ema`
        self.ema = field.params
        self.step_count = 0
        self.ema_n_step = 0     # EMA.n_step (ema.py:14,25-27): equal to step_count unless a resumed job set it (set_ema_n_step)
        # occupancy (Volume bitfield/opafield, volume.py:741-760,959-969)
        ng = cfg.n_grid
        self._bitfield = torch.ones(ng ** 3, dtype=torch.bool, device=dev)
        self._opafield = torch.zeros(ng ** 3, dtype=f32, device=dev)
        self._bits = None
        self._occ_scratch = None
        self._pb = self._gb = None
        self.generation = 0
        NgpPipeline._built += 1
        self.generation_id = NgpPipeline._built      # (unlike id(): never handed out twice in a process)
        self.occupancy_sync = None      # callable(opafield, new_bitfield) run on the refresh stream before a refreshed bitfield is applied
        self.set_bitfield(self.bitfield)

    def set_ema_n_step(self, n_step):
        """EMA.set_n_step (arcnerf/trainer/ema.py:25-27): the trainer calls it with progress.start_epoch (arcnerf_trainer.py:70), so the
        running average of a resumed job is de-biased with the epoch count while Adam's own step count comes from its state"""
        self.ema_n_step = int(n_step)

    def sample_count(self):
        """Valid samples of the batch marched LAST (with prefetch: the batch marched ahead on the sampling stream) as a python int.
        A host read: it waits for the sampling stream first - `int(pipe.n_dev.item())` from the main stream races with a marcher that
        is still running and returns whatever the buffer holds (the dynamic batch size of tools/psnr_curve.py was fed that way in
        rounds 1-2: one source of its run-to-run spread)."""
        if self.aux_stream is not None:
            self.aux_stream.synchronize()
        return int(self.n_dev.item())

    # ---- parameter binding -------------------------------------------------------------------------
    def bind_params(self, tensors):
        """Run on externally owned flat fp32 tensors {'table','geo_w','rad_w'[,'geo_b','rad_b']} (e.g. nn.Parameters of the
        model mirror) instead of the field's own flat buffer.  None restores the field."""
        self._pb = tensors

    def bind_grads(self, tensors):
        self._gb = tensors

    def _p(self, name):
        if self._pb is not None:
            return self._pb.get(name)
        return self.field.view(name)

    def _g(self, name):
        if self._gb is not None:
            return self._gb.get(name)
        return self.field.view(name, self.field.grads)

    # ---- occupancy ------------------------------------------------------------------------------
    # the occupancy state is produced on the refresh stream: reading it makes the CURRENT stream wait for the last refresh
    @property
    def bitfield(self):
        self._wait_occupancy_state()
        return self._bitfield

    @bitfield.setter
    def bitfield(self, value):
        self._bitfield = value

    @property
    def opafield(self):
        self._wait_occupancy_state()
        return self._opafield

    def set_bitfield(self, bitfield_bool):
        """bitfield (n_grid^3 | n_grid,n_grid,n_grid) bool, flat index x*n*n + y*n + z"""
        self.bitfield = bitfield_bool.reshape(-1).to(self.field.device).contiguous()
        if self.packed_bits:
            ng3 = self.bitfield.numel()
            assert ng3 % 8 == 0
            w = (2 ** torch.arange(8, device=self.field.device, dtype=torch.int32))
            bits = (self.bitfield.view(-1, 8).to(torch.int32) * w).sum(-1).to(torch.uint8).contiguous()
            self._bits = bits
            self._share_bits_with_aux()

    def set_occupancy_bits(self, bits, mode):
        """Hand over an already packed occupancy (uint8, 1 bit per voxel) without conversion: mode 1 = x-major order, mode 2 =
        Morton order with clamped coordinates (BitfieldBound.density_bitfield, the `_bitfield_func` layout)."""
        assert bits.dtype == torch.uint8 and bits.is_contiguous() and mode in (1, 2)
        assert bits.numel() * 8 == self.cfg.n_grid ** 3
        self._bits, self.packed_bits = bits, mode
        self._share_bits_with_aux()

    def _share_bits_with_aux(self):
        """occupancy bits produced / handed over on the current stream are read by marchers on the sampling stream: order the
        sampling stream behind their producer and tell the allocator about the second consumer"""
        self._build_cull_grid()
        aux = getattr(self, 'aux_stream', None)
        if aux is not None and self._bits is not None and self._bits.is_cuda:
            aux.wait_stream(torch.cuda.current_stream())
            self._bits.record_stream(aux)
            if self._coarse is not None:
                self._coarse.record_stream(aux)

    def _build_cull_grid(self):
        """the marcher's ray-culling grid (arcn_march_cull_grid) of the occupancy just handed over; march_cull=False: none"""
        self._coarse = None
        ng = self.cfg.n_grid
        # (only for the bits this pipeline packs itself, set_bitfield: a Morton bitfield handed over by set_occupancy_bits is updated in
        # place by its owner's kernels, behind any version counter - a stale culling grid would drop samples)
        if self._bits is None or not self._bits.is_cuda or self.packed_bits != 1 or ng < 16 or ng % 4 or not getattr(self, 'march_cull', True):
            return
        cells = (ng // 4) ** 3
        coarse = torch.empty(cells, dtype=torch.uint8, device=self._bits.device)
        tmp = torch.empty(cells, dtype=torch.uint8, device=self._bits.device)
        N.check(N.lib().arcn_march_cull_grid(N.ptr(self._bits), int(self.packed_bits), ng, N.ptr(coarse), N.ptr(tmp), N.stream()), 'march_cull_grid')
        self._coarse = coarse

    def _occ(self):
        return self._bits if self.packed_bits else self._bitfield

    def _cached(self, key, make):
        if not hasattr(self, '_cache'):
            self._cache = {}
        if key not in self._cache:
            self._cache[key] = make()
        return self._cache[key]

    def _select_cells(self, n_cells, perm=None):
        from .geometry.volume import select_refresh_cells
        cache = self._cached('refresh_cache', dict)
        rng = self._cached('np_rng', lambda: np.random.default_rng(12345))
        return select_refresh_cells(self.bitfield, n_cells, cache, rng, perm=perm)

    # ---- forward --------------------------------------------------------------------------------
    def prefetch_samples(self, rays_o, rays_d, noise=False):
        """March `rays` on the auxiliary stream into the spare sample-buffer set (with the sample positions, the per-ray harmonics
        and - noise=True - the density noise of a training step); the next forward() on the SAME tensors picks the result up
        instead of marching again.  Safe to call right after the current step's forward was issued."""
        if not self.use_streams:
            return
        main = torch.cuda.current_stream()
        self.aux_stream.wait_stream(main)  # occupancy / previous consumers of the spare set are ordered before us
        if len(self._prefetched) >= self.prefetch_depth:
            self._prefetched.pop(0)   # never picked up: its set is free again
        busy = {self._cur_set} | {pf[3] for pf in self._prefetched}
        spare = next(i for i in range(len(self._sets)) if i not in busy)
        with torch.cuda.stream(self.aux_stream):
            self._sample_into(self._sets[spare], rays_o, rays_d, waves=self.march_waves)
            self._planned[spare] = False
            if noise and self._plan_floats and self.level_major:      # (noise: the batch of a training step - its samples will be scattered)
                bs = self._sets[spare]
                R_ = rays_o.shape[0]
                N.check(N.lib().arcn_hashgrid_bwd_plan(N.ptr(bs['xyz']), N.C.addressof(self.field.grid_desc), N.ptr(bs['plan_ws']), self._plan_floats,
                                                       self.cap, bs['offsets'][R_:R_ + 1].data_ptr(), N.stream()), 'hashgrid_bwd_plan')
                self._planned[spare] = True
            self._noise_ready[spare] = bool(noise and self.cfg.noise_std > 0)
            if self._noise_ready[spare]:
                self._sets[spare]['noise'].normal_(0.0, self.cfg.noise_std)
            ev = torch.cuda.Event()
            ev.record(self.aux_stream)
        # the entry keeps the ray tensors alive (their addresses cannot be handed to another batch by the caching allocator) and
        # remembers their versions (an in-place refill of a staging buffer invalidates the prefetch)
        self._prefetched.append((rays_o.data_ptr(), rays_d.data_ptr(), rays_o.shape[0], spare, ev, rays_o, rays_d,
                                 rays_o._version, rays_d._version))

    def sample(self, rays_o, rays_d):
        """[A] bounds + occupancy marching in packed form (no host sync).  Advances the pcg32 like the reference."""
        key = (rays_o.data_ptr(), rays_d.data_ptr(), rays_o.shape[0])
        hit = next((k for k, pf in enumerate(self._prefetched)
                    if pf[:3] == key and pf[5]._version == pf[7] and pf[6]._version == pf[8]), None)
        if hit is not None:
            pf = self._prefetched.pop(hit)
            del self._prefetched[:hit]   # older entries were skipped by the caller: drop them
            torch.cuda.current_stream().wait_event(pf[4])
            self._cur_set = pf[3]
            self.buf.update(self._sets[self._cur_set])
            self.sampled_ahead = True
        else:
            self.sampled_ahead = False
            if self.prefetch_depth == 1:
                self._prefetched = []
            elif any(pf[3] == self._cur_set for pf in self._prefetched):   # cannot happen: the current set is never handed out
                raise RuntimeError('sample buffer set in use by a prefetch')
            self._noise_ready[self._cur_set] = False
            self._planned[self._cur_set] = False
            self._sample_into(self._sets[self._cur_set], rays_o, rays_d)
            self.buf.update(self._sets[self._cur_set])
            if self.use_streams:      # (whoever reads this batch's count on the sampling stream waits for this, not for the step behind it)
                self.sample_event = torch.cuda.Event()
                self.sample_event.record()
        R = rays_o.shape[0]
        self.n_dev = self.buf['offsets'][R:R + 1]  # device-side sample count (view, no sync)
        return self.n_dev

    def _sample_into(self, b, rays_o, rays_d, waves=0):
        cfg = self.cfg
        self._wait_occupancy_bits()
        R = rays_o.shape[0]
        assert R <= self.max_rays
        L = N.lib()
        st = N.stream()
        if waves > 0 and R > waves:     # a launch with time to spare beside other kernels: persistent wavefronts (same outputs)
            N.check(L.arcn_march_count_waves(N.ptr(rays_o), N.ptr(rays_d), N.ptr(self.aabb23), cfg.n_grid, N.ptr(self._occ()),
                                             int(self.packed_bits), N.ptr(getattr(self, '_coarse', None)), cfg.n_sample, cfg.dt, cfg.near_distance,
                                             int(self.torch_aabb), self.rng.state, self.rng.inc, N.ptr(b['scratch_t']),
                                             N.ptr(b['counts']), N.ptr(b['near']), N.ptr(b['far']), R, int(waves), st), 'march_count_waves')
        elif getattr(self, '_coarse', None) is not None:     # rays that pass no occupied block leave before they march (same outputs)
            N.check(L.arcn_march_count_culled(N.ptr(rays_o), N.ptr(rays_d), N.ptr(self.aabb23), cfg.n_grid, N.ptr(self._occ()),
                                              int(self.packed_bits), N.ptr(self._coarse), cfg.n_sample, cfg.dt, cfg.near_distance,
                                              int(self.torch_aabb), self.rng.state, self.rng.inc, N.ptr(b['scratch_t']),
                                              N.ptr(b['counts']), N.ptr(b['near']), N.ptr(b['far']), R, st), 'march_count_culled')
        else:
            N.check(L.arcn_march_count(N.ptr(rays_o), N.ptr(rays_d), N.ptr(self.aabb23), cfg.n_grid, N.ptr(self._occ()),
                                       int(self.packed_bits), cfg.n_sample, cfg.dt, cfg.near_distance, int(self.torch_aabb),
                                       self.rng.state, self.rng.inc, N.ptr(b['scratch_t']), N.ptr(b['counts']), N.ptr(b['near']),
                                       N.ptr(b['far']), R, st), 'march_count')
        self.rng.advance()
        # offsets (clamped to the capacity) and the dense width the reference would have used, max(2, max count)
        N.check(L.arcn_exclusive_scan_i32(N.ptr(b['counts']), N.ptr(b['offsets']), R, self.cap, N.ptr(b['p_dense']), st), 'scan')
        N.check(L.arcn_march_write(N.ptr(b['scratch_t']), N.ptr(b['counts']), N.ptr(b['offsets']), cfg.n_sample, N.ptr(b['t']),
                                   N.ptr(b['ray_id']), R, self.cap, st), 'march_write')
        N.check(L.arcn_packed_points(N.ptr(rays_o), N.ptr(rays_d), N.ptr(b['t']), N.ptr(b['ray_id']), N.ptr(b['xyz']),
                                     N.ptr(b['dirs']), self.cap, b['offsets'][R:R + 1].data_ptr(), st), 'packed_points')
        if self.ray_sh:
            N.check(L.arcn_ngp_ray_sh(N.ptr(rays_d), cfg.sh_degree, N.ptr(b['sh_ray']), R, st), 'ngp_ray_sh')

    def forward(self, rays_o, rays_d, bkg_color=None, train=False, noise=None, huber_target=None, presampled=False):
        """Render rays: returns rgb (R,3), depth (R), mask (R) views of the internal buffers.
        presampled: sample(rays_o, rays_d) has just been called by the caller (e.g. to check the capacity): do not march again.
        huber_target (R,3), training only: compositing, the Huber image loss and the compositor's backward run as ONE kernel; the
        loss lands in self.last_loss and backward() starts at the radiance net."""
        cfg, b, fld = self.cfg, self.buf, self.field
        R = rays_o.shape[0]
        rays_o, rays_d = rays_o.contiguous().float(), rays_d.contiguous().float()
        n_dev = self.n_dev if presampled else self.sample(rays_o, rays_d)   # + sample positions / directions, per-ray harmonics
        S = self.cap
        L, st = N.lib(), N.stream()
        if noise == 'auto':   # training noise: the prefetched set brings it along, otherwise draw it now
            noise = None
            if cfg.noise_std > 0:
                noise = b['noise'] if self._noise_ready[self._cur_set] else b['noise'].normal_(0.0, cfg.noise_std)
                self._noise_ready[self._cur_set] = False
        self.generation += 1
        if self.level_major:
            # encode -> geometry net through a LEVEL-MAJOR feature buffer (L, S, 2): the XCD-affine gather writes it coalesced and
            # the net's tile loads read 128 contiguous bytes per level
            N.check(L.arcn_hashgrid_fwd_xcd(N.ptr(b['xyz']), N.ptr(self._p('table')), N.C.addressof(fld.grid_desc), N.ptr(b['feat']),
                                            1, S, S, n_dev.data_ptr(), st), 'hashgrid_fwd_xcd')
            if train and self._carry_rays is not None:
                # prefetch point 5 (two batches ahead only): the marching handed over by the PREVIOUS train_step starts behind this step's
                # gather instead of beside it
                carry, self._carry_rays = self._carry_rays, None
                self.prefetch_samples(*carry, noise=True)
            if self.fused_nets:
                # both nets in one kernel: the geometry net's output tile is the radiance net's operand without leaving the registers
                N.check(L.arcn_ngp_nets_fwd(N.ptr(b['feat']), S, N.ptr(self._p('geo_w')), N.C.addressof(fld.geo_desc), N.ptr(b['geo_out']),
                                            N.ptr(b['sh_ray']), N.ptr(b['ray_id']), int(cfg.rad_mode == 'fv'), N.ptr(self._p('rad_w')),
                                            N.C.addressof(fld.rad_desc), N.ptr(b['rgb_s']), N.ptr(b['rad_acts']) if train else None, N.ptr(b['sigma']),
                                            N.ACT[cfg.sigma_act], S, S, n_dev.data_ptr(), st), 'ngp_nets_fwd')
            else:
                N.check(L.arcn_mlp_fwd_lm(N.ptr(b['feat']), S, N.ptr(self._p('geo_w')), N.C.addressof(fld.geo_desc), N.ptr(b['geo_out']),
                                          N.ptr(b['geo_acts']) if train else None, S, S, n_dev.data_ptr(), st), 'mlp_fwd_lm(geo)')
        else:
            F.hashgrid_fwd(b['xyz'], self._p('table'), fld.grid_desc, n_dev=n_dev, out=b['feat'])
            F.mlp_fwd(b['feat'], self._p('geo_w'), self._p('geo_b'), fld.geo_desc, save_acts=train, n_dev=n_dev, out=b['geo_out'],
                      acts=b['geo_acts'])
        if self.level_major and self.fused_nets:
            pass
        elif self.fused_glue:
            # view-direction harmonics once per ray; the radiance net assembles [geo features | SH(ray)] in its operand load and
            # writes sigma = act(geo_out[:, 0]) on the way: no rad_in buffer, no glue kernel
            N.check(L.arcn_mlp_fwd_cat(N.ptr(b['geo_out']), N.ptr(b['sh_ray']), N.ptr(b['ray_id']), int(cfg.rad_mode == 'fv'),
                                       N.ptr(self._p('rad_w')), N.C.addressof(fld.rad_desc), N.ptr(b['rgb_s']),
                                       N.ptr(b['rad_acts']) if train else None, N.ptr(b['sigma']), N.ACT[cfg.sigma_act], S, S,
                                       n_dev.data_ptr(), st), 'mlp_fwd_cat(rad)')
        else:
            if self.ray_sh:
                N.check(L.arcn_ngp_glue_fwd_rays(N.ptr(b['geo_out']), N.ptr(b['sh_ray']), N.ptr(b['ray_id']), fld.geo_out_dim,
                                                 fld.feat_off, cfg.W_feat, cfg.sh_degree, int(cfg.rad_mode == 'fv'),
                                                 N.ACT[cfg.sigma_act], N.ptr(b['rad_in']), N.ptr(b['sigma']), S, n_dev.data_ptr(), st),
                        'ngp_glue_fwd_rays')
            else:
                F.ngp_glue_fwd(b['geo_out'], b['dirs'], fld.feat_off, cfg.W_feat, cfg.sh_degree, feat_first=(cfg.rad_mode == 'fv'),
                               sigma_act=cfg.sigma_act, n_dev=n_dev, rad_in=b['rad_in'], sigma=b['sigma'])
            F.mlp_fwd(b['rad_in'], self._p('rad_w'), self._p('rad_b'), fld.rad_desc, save_acts=train, n_dev=n_dev, out=b['rgb_s'],
                      acts=b['rad_acts'])
        bk, bk_rows = (None, 0) if bkg_color is None else (bkg_color.contiguous().float().view(-1, 3), bkg_color.view(-1, 3).shape[0])
        self._bkg = bk
        self._noise = noise
        self._composite_bwd_done = False
        if huber_target is not None and train:
            ring, k = b['loss_ring'], self._loss_slot
            self._loss_slot = (k + 1) % ring.shape[0]
            self._loss_step[k] = self.generation
            N.check(L.arcn_composite_packed_train(N.ptr(b['sigma']), N.ptr(b['rgb_s']), N.ptr(b['t']), N.ptr(b['offsets']), N.ptr(noise),
                                                  N.ptr(bk), bk_rows, R, 2, b['p_dense'].data_ptr(), int(cfg.add_inf_z),
                                                  int(cfg.white_bkg), N.ptr(huber_target.contiguous().float()), cfg.huber_delta,
                                                  cfg.loss_weight * self.loss_scale, N.ptr(b['rgb']), N.ptr(b['depth']), N.ptr(b['mask']), N.ptr(b['d_rgb']),
                                                  ring[k].data_ptr(),
                                                  N.ptr(b['d_sigma']), N.ptr(b['d_rgb_s']), N.ptr(b['counts']), st), 'composite_packed_train')
            self.last_loss = StepLoss(self, k, self.generation, (R + 3) // 4)
            self._composite_bwd_done = True
            return b['rgb'][:R], b['depth'][:R], b['mask'][:R]
        N.check(L.arcn_composite_packed_fwd(N.ptr(b['sigma']), N.ptr(b['rgb_s']), N.ptr(b['t']), N.ptr(b['offsets']),
                                            N.ptr(noise), N.ptr(bk), bk_rows, R, 2, b['p_dense'].data_ptr(),
                                            int(cfg.add_inf_z), int(cfg.white_bkg), N.ptr(b['rgb']), N.ptr(b['depth']),
                                            N.ptr(b['mask']), None, N.ptr(b['counts']), st), 'composite_packed_fwd')
        return b['rgb'][:R], b['depth'][:R], b['mask'][:R]

    # ---- backward + optimiser -----------------------------------------------------------------------
    def backward(self, rays_o, rays_d, d_rgb, d_depth=None, d_mask=None):
        """Accumulate d loss / d params into field.grads given the per-ray output gradients of the last forward(train=True)."""
        cfg, b, fld = self.cfg, self.buf, self.field
        R = rays_o.shape[0]
        L, st = N.lib(), N.stream()
        n_dev = self.n_dev
        bk = self._bkg
        bk_rows = 0 if bk is None else bk.shape[0]
        if not getattr(self, '_composite_bwd_done', False):   # the fused compositor already produced d_sigma / d_rgb_s
            N.check(L.arcn_composite_packed_bwd(N.ptr(b['sigma']), N.ptr(b['rgb_s']), N.ptr(b['t']), N.ptr(b['offsets']),
                                                N.ptr(self._noise), N.ptr(bk), bk_rows, R, 2, b['p_dense'].data_ptr(),
                                                int(cfg.add_inf_z), int(cfg.white_bkg), N.ptr(d_rgb), N.ptr(d_depth), N.ptr(d_mask),
                                                N.ptr(b['d_sigma']), N.ptr(b['d_rgb_s']), N.ptr(b['counts']), st), 'composite_packed_bwd')
        self._composite_bwd_done = False
        S = self.cap
        # the step's tail launch (optimizer_step) sums the dW partials and applies the optimiser: the nets leave them in their scratch
        tail = bool(self._fuse_next and self._tail is not None and self._adam_rest is not None and self._pb is None and self._gb is None and
                    self.ema is self.field.params)
        # dx and dW of each net come out of ONE fused kernel (arcn_mlp_bwd with dweights): dpre never leaves the registers
        if self.fused_glue:
            # ... and the radiance net writes d geo_out directly (feature half of its input gradient + sigma's gradient in col 0)
            N.check(L.arcn_mlp_bwd_cat(N.ptr(b['geo_out']), N.ptr(b['sh_ray']), N.ptr(b['ray_id']), int(cfg.rad_mode == 'fv'),
                                       N.ptr(self._p('rad_w')), N.C.addressof(fld.rad_desc), N.ptr(b['rgb_s']), N.ptr(b['rad_acts']),
                                       N.ptr(b['d_rgb_s']), N.ptr(b['d_geo_out']), N.ptr(b['d_sigma']), N.ACT[cfg.sigma_act],
                                       N.ptr(self._g('rad_w')), N.ptr(b['rad_scratch']), int(tail), S, S, n_dev.data_ptr(), st),
                    'mlp_bwd_cat(rad)')
        else:
            N.check(L.arcn_mlp_bwd(N.ptr(b['rad_in']), N.ptr(self._p('rad_w')), N.ptr(self._p('rad_b')), N.C.addressof(fld.rad_desc),
                                   N.ptr(b['rgb_s']), N.ptr(b['rad_acts']), N.ptr(b['d_rgb_s']), N.ptr(b['d_rad_in']),
                                   N.ptr(self._g('rad_w')), N.ptr(self._g('rad_b')), N.ptr(b['rad_scratch']), S, S, n_dev.data_ptr(), st),
                    'mlp_bwd(rad)')
            F.ngp_glue_bwd(b['geo_out'], b['d_rad_in'], b['d_sigma'], fld.feat_off, cfg.W_feat, cfg.sh_degree,
                           feat_first=(cfg.rad_mode == 'fv'), sigma_act=cfg.sigma_act, n_dev=n_dev, d_geo_out=b['d_geo_out'])
        self._prefetch_point(1)
        if self.level_major:
            N.check(L.arcn_mlp_bwd_lm(N.ptr(b['feat']), S, N.ptr(self._p('geo_w')), N.C.addressof(fld.geo_desc), N.ptr(b['geo_out']),
                                      N.ptr(b['geo_acts']), N.ptr(b['d_geo_out']), N.ptr(b['d_feat']), N.ptr(self._g('geo_w')),
                                      N.ptr(b['geo_scratch']), int(tail), S, S, n_dev.data_ptr(), st), 'mlp_bwd_lm(geo)')
            self._prefetch_point(2)
            if self._fuse_next and self._adam_rest is not None and self._pb is None and self._gb is None and self.ema is fld.params:
                # scatter + optimiser of the one-owner levels in one pass (the refresh on its own stream still reads the parameters)
                if self._occ_params_event is not None:
                    torch.cuda.current_stream().wait_event(self._occ_params_event)
                    self._occ_params_event = None
                fused = N.C.c_uint32(0)
                t_lo = fld._seg['table'][0]
                planned = self._planned[self._cur_set]
                self._planned[self._cur_set] = False      # (one plan serves one scatter)
                if planned:
                    N.check(L.arcn_hashgrid_bwd_lm_adam_planned(N.ptr(b['xyz']), N.ptr(b['d_feat']), S, N.C.addressof(fld.grid_desc), N.ptr(self._g('table')),
                                                                N.ptr(fld.view('table')), self.exp_avg[t_lo:].data_ptr(), self.exp_avg_sq[t_lo:].data_ptr(),
                                                                float(cfg.lr), float(cfg.betas[0]), float(cfg.betas[1]), float(cfg.eps),
                                                                float(cfg.weight_decay), -1.0 if cfg.ema_decay is None else float(cfg.ema_decay), 1.0,
                                                                self.step_count + 1, self.ema_n_step + 1, N.ptr(self._sets[self._cur_set]['plan_ws']),
                                                                self._plan_floats, N.ptr(self.hash_ws), self.hash_ws.numel(), S, n_dev.data_ptr(),
                                                                N.C.byref(fused), st), 'hashgrid_bwd_lm_adam_planned')
                else:
                    N.check(L.arcn_hashgrid_bwd_lm_adam(N.ptr(b['xyz']), N.ptr(b['d_feat']), S, N.C.addressof(fld.grid_desc), N.ptr(self._g('table')),
                                                        N.ptr(fld.view('table')), self.exp_avg[t_lo:].data_ptr(), self.exp_avg_sq[t_lo:].data_ptr(),
                                                        float(cfg.lr), float(cfg.betas[0]), float(cfg.betas[1]), float(cfg.eps),
                                                        float(cfg.weight_decay), -1.0 if cfg.ema_decay is None else float(cfg.ema_decay), 1.0, self.step_count + 1,
                                                        self.ema_n_step + 1, N.ptr(self.hash_ws), self.hash_ws.numel(), int(self._ws_clear), S,
                                                        n_dev.data_ptr(), N.C.byref(fused), st), 'hashgrid_bwd_lm_adam')
                if fused.value != self._fused_mask:
                    raise RuntimeError('hashgrid_bwd_lm_adam fused levels {:#x}, the optimiser plan expects {:#x}'.format(fused.value, self._fused_mask))
                self._fused_step = True
                self._tail_step = tail
                self._ws_clear = False
            elif getattr(self, '_level_sync', None) is not None:
                # data-parallel step: the scatter in level groups, each group's slice of the flat gradient on the wire (asynchronous
                # all-reduce on the communicator's stream) while the next group is scattered (distributed.LevelGroupedGradSync)
                if tail:
                    raise RuntimeError('the step tail was planned for a step whose scatter does not apply the optimiser')
                sync = self._level_sync
                for gi, (mask, _, _) in enumerate(sync.groups):
                    N.check(L.arcn_hashgrid_bwd_lm_levels(N.ptr(b['xyz']), N.ptr(b['d_feat']), S, N.C.addressof(fld.grid_desc), N.ptr(self._g('table')),
                                                          N.ptr(self.hash_ws), self.hash_ws.numel(), S, n_dev.data_ptr(), int(mask), int(gi > 0), st),
                            'hashgrid_bwd_lm_levels')
                    sync.launch_group(gi, fld.grads)
                self._ws_clear = False
            else:
                if tail:
                    raise RuntimeError('the step tail was planned for a step whose scatter does not apply the optimiser')
                planned = self._planned[self._cur_set]
                self._planned[self._cur_set] = False
                if planned:
                    N.check(L.arcn_hashgrid_bwd_lm_planned(N.ptr(b['xyz']), N.ptr(b['d_feat']), S, N.C.addressof(fld.grid_desc), N.ptr(self._g('table')),
                                                           N.ptr(self._sets[self._cur_set]['plan_ws']), self._plan_floats, N.ptr(self.hash_ws),
                                                           self.hash_ws.numel(), S, n_dev.data_ptr(), st), 'hashgrid_bwd_lm_planned')
                else:
                    N.check(L.arcn_hashgrid_bwd_lm(N.ptr(b['xyz']), N.ptr(b['d_feat']), S, N.C.addressof(fld.grid_desc), N.ptr(self._g('table')),
                                                   N.ptr(self.hash_ws), self.hash_ws.numel(), S, n_dev.data_ptr(), st), 'hashgrid_bwd_lm')
                self._ws_clear = False
            self._fuse_next = False
            return
        N.check(L.arcn_mlp_bwd(N.ptr(b['feat']), N.ptr(self._p('geo_w')), N.ptr(self._p('geo_b')), N.C.addressof(fld.geo_desc),
                               N.ptr(b['geo_out']), N.ptr(b['geo_acts']), N.ptr(b['d_geo_out']), N.ptr(b['d_feat']),
                               N.ptr(self._g('geo_w')), N.ptr(self._g('geo_b')), N.ptr(b['geo_scratch']), S, S, n_dev.data_ptr(), st),
                'mlp_bwd(geo)')
        self._prefetch_point(2)
        N.check(L.arcn_hashgrid_bwd(N.ptr(b['xyz']), N.ptr(self._p('table')), N.ptr(b['d_feat']), N.C.addressof(fld.grid_desc),
                                    N.ptr(self._g('table')), None, N.ptr(self.hash_ws),
                                    0 if self.hash_ws is None else self.hash_ws.numel(), S, n_dev.data_ptr(), st),
                'hashgrid_bwd')

    loss_scale = 1.0
    _built = 0

    def huber_grad(self, rgb, target):
        """ImgLoss(Huber, delta, weight) of arcnerf/loss/img_loss.py:60-100: loss value and d loss / d rgb (mean over R*3)."""
        cfg, b = self.cfg, self.buf
        R = rgb.shape[0]
        loss, d = F.huber_loss_grad(rgb, target, cfg.huber_delta, cfg.loss_weight * self.loss_scale, dx=b['d_rgb'][:R], loss=b['loss'])
        return loss[0], d

    def _plan_fused_adam(self, S):
        """-> list of (lo, hi) slices of the flat buffer left to the plain Adam kernel when the scatter applies the optimiser to its
        one-owner levels, or None when that form does not apply (deterministic mode, a separate EMA shadow, no such level, a slice
        that would not start 16-byte aligned)"""
        fld = self.field
        mask = int(N.lib().arcn_hashgrid_bwd_fusable_levels(N.C.addressof(fld.grid_desc), int(S)))
        if mask == 0 or 'table' not in fld._seg:
            return None
        t_lo, t_n = fld._seg['table'][0], fld._seg['table'][1]
        Fq = self.cfg.n_feat_per_entry
        cuts, pos = [], 0
        for l in range(len(fld.resolutions)):
            a, b_ = t_lo + fld.offsets[l] * Fq, t_lo + fld.offsets[l + 1] * Fq
            if (mask >> l) & 1:
                if a > pos:
                    cuts.append((pos, a))
                pos = b_
        if pos < fld.n_params:
            cuts.append((pos, fld.n_params))
        if any(lo % 4 for lo, _ in cuts) or t_lo % 4:
            return None
        self._fused_mask = mask
        return cuts

    def optimizer_step(self, world_size=1, lo=None, hi=None, advance=True):
        """fused Adam + EMA (+ gradient clear) on the whole flat buffer or on its slice [lo, hi) (pipelined gradient sync:
        one call per segment, `advance` only on the first so every segment sees the same step count)."""
        cfg, fld = self.cfg, self.field
        param_epoch.bump()      # (the parameters change here: whatever was cached from them since train_step() began is old)
        if self._occ_params_event is not None:
            torch.cuda.current_stream().wait_event(self._occ_params_event)
            self._occ_params_event = None
        if advance:
            self.step_count += 1
            self.ema_n_step += 1
        if self._fused_step:        # the scatter of this step already updated its levels: the rest of the flat buffer
            self._fused_step = False
            if lo is not None or hi is not None or world_size != 1:
                raise RuntimeError('the scatter of this step already applied the optimiser to its levels: single-GPU, whole-buffer step only')
            if self._tail_step:
                self._tail_step = False
                t, b = self._tail, self.buf
                S = self.cap
                flat = (N.C.c_int64 * max(2, 2 * len(t['runs'])))(*[v for a, b_ in t['runs'] for v in (int(a), int(b_) - int(a))])
                N.check(N.lib().arcn_ngp_step_tail(N.C.addressof(fld.geo_desc), N.ptr(b['geo_scratch']), t['geo_w'], N.C.addressof(fld.rad_desc),
                                                   N.ptr(b['rad_scratch']), t['rad_w'], S, S, N.ptr(fld.params), N.ptr(fld.grads),
                                                   N.ptr(self.exp_avg), N.ptr(self.exp_avg_sq), N.ptr(self.ema), N.C.cast(flat, N.C.c_void_p),
                                                   len(t['runs']), float(cfg.lr), float(cfg.betas[0]), float(cfg.betas[1]), float(cfg.eps),
                                                   float(cfg.weight_decay), float(-1.0 if cfg.ema_decay is None else cfg.ema_decay), 1.0,
                                                   self.step_count, self.ema_n_step, N.ptr(self.hash_ws), t['clear_words'], N.stream()),
                        'ngp_step_tail')
                self._ws_clear = True
                return
            if 1 < len(self._adam_rest) <= 4:      # the small levels in front of the fused ones and the MLP weights behind them: one launch
                F.adam_ema_step_runs(fld.params, fld.grads, self.exp_avg, self.exp_avg_sq, self.ema if cfg.ema_decay is not None else None, self._adam_rest,
                                     self.step_count, lr=cfg.lr, betas=cfg.betas, eps=cfg.eps, weight_decay=cfg.weight_decay,
                                     ema_decay=cfg.ema_decay if cfg.ema_decay is not None else 0.0, grad_scale=1.0,
                                     ema_step=self.ema_n_step, zero_grad=True)
                return
            slices = [slice(a, b_) for a, b_ in self._adam_rest]
        else:
            slices = [slice(lo, hi)]
        with_ema = cfg.ema_decay is not None      # (None: plain Adam - the drop-in step of an optimiser built without an EMA)
        for sl in slices:
            F.adam_ema_step(fld.params[sl], fld.grads[sl], self.exp_avg[sl], self.exp_avg_sq[sl], self.ema[sl] if with_ema else None, self.step_count,
                            lr=cfg.lr, betas=cfg.betas, eps=cfg.eps, weight_decay=cfg.weight_decay, ema_decay=cfg.ema_decay if with_ema else 0.0,
                            grad_scale=1.0 / world_size, ema_step=self.ema_n_step, zero_grad=True)

    def train_step(self, rays_o, rays_d, target_rgb, bkg_color=None, all_reduce=None, world_size=1, next_rays=None, grad_sync=None, loss_scale=1.0):
        """fwd + loss + bwd (+ one gradient all-reduce) + Adam/EMA.  Returns the loss tensor (device, no sync).
        next_rays = (rays_o, rays_d) of the FOLLOWING step: their marching is overlapped with this step's backward.
        grad_sync: a distributed.LevelGroupedGradSync or ShardedGradSync (takes precedence over the flat `all_reduce` callable).
        loss_scale: factor on the loss weight of THIS step.  The loss is a mean over this rank's rays and the optimiser divides the summed
        gradients by the world size: with UNEQUAL ray shards (distributed.balanced_shards) a rank passes
        loss_scale = R_local * world_size / R_global so that the step optimises the mean over the GLOBAL batch (every ray the same weight,
        as in one process on the whole batch); equal shards: 1."""
        cfg, b = self.cfg, self.buf
        self.loss_scale = float(loss_scale)
        param_epoch.bump()      # (this step's kernels rewrite the parameters through raw pointers)
        fused = self.fused_composite
        rgb, _, _ = self.forward(rays_o, rays_d, bkg_color, train=True, noise='auto', huber_target=target_rgb if fused else None)
        if self.prefetch_at == 5 and self.prefetch_depth >= 2 and grad_sync is None and all_reduce is None:
            self._carry_rays = next_rays
        self._next_rays = next_rays
        self._prefetch_now = self.prefetch_at if (grad_sync is None and all_reduce is None) else self.prefetch_at_dist
        if self._prefetch_now == 5 and self.prefetch_depth >= 2:
            self._next_rays = None          # issued inside the NEXT step's forward, behind its gather (see forward)
        self._prefetch_point(0)
        if fused:
            loss, d_rgb = self.last_loss, b['d_rgb'][:rays_o.shape[0]]
        else:
            loss, d_rgb = self.huber_grad(rgb, target_rgb)
        # one GPU: nothing has to be summed across ranks between the scatter and the optimiser, so the scatter applies it (see backward)
        self._fuse_next = grad_sync is None and all_reduce is None and world_size == 1
        grouped = grad_sync is not None and hasattr(grad_sync, 'launch_group')
        if grouped and not (self.level_major and self._pb is None and self._gb is None):
            raise RuntimeError('LevelGroupedGradSync needs the level-major scatter on the field\'s own flat buffers')
        self._level_sync = grad_sync if grouped else None
        self.backward(rays_o, rays_d, d_rgb)
        self._level_sync = None
        if grouped:
            # the groups are already on the wire (issued inside the backward, behind their part of the scatter)
            self._prefetch_point(3)
            for i, (lo, hi) in enumerate(grad_sync.segments):
                grad_sync.wait(i)
                self.optimizer_step(world_size, lo, hi, advance=(i == 0))
            return loss
        if grad_sync is not None:
            # distributed.ShardedGradSync: reduce-scatter, Adam + EMA on this rank's shard (and the replicated tail), all-gather
            if not hasattr(grad_sync, 'gather'):
                raise RuntimeError('grad_sync must be a distributed.LevelGroupedGradSync or a distributed.ShardedGradSync')
            if self._pb is not None or self._gb is not None:
                raise RuntimeError('ShardedGradSync needs the step on the field\'s own flat buffers')
            grad_sync.launch(self.field.grads)
            self._prefetch_point(3)
            grad_sync.wait()
            for i, (lo, hi) in enumerate(grad_sync.segments):
                self.optimizer_step(world_size, lo, hi, advance=(i == 0))
            grad_sync.clear_foreign(self.field.grads)
            if cfg.ema_decay is not None and self.ema is not self.field.params:
                grad_sync.gather(self.field.params, self.ema)
            else:
                grad_sync.gather(self.field.params)
            return loss
        self._prefetch_point(3)  # before a blocking collective is queued: the second stream only waits for the backward
        if all_reduce is not None:
            all_reduce(self.field.grads)
        self.optimizer_step(world_size)
        self._prefetch_point(4)
        return loss

    def _prefetch_point(self, where):
        """Issue the next batch's marching (second stream) at point `where` of the step: 0 after the forward, 1 before the
        geometry-net backward, 2 before the hash-grid scatter, 3 before the optimiser, 4 after it.  The marcher is pure VALU work with
        a 256 KiB working set; it costs least next to the LDS / HBM-bound kernels.  One batch ahead (prefetch_depth 1) its chain has
        to be done when the next forward starts, so it goes next to the backward (1); two batches ahead it goes next to the
        optimiser pass (3): 0.711 vs 0.722 ms/step, A/B in one session (`tools/ab_prefetch.sh`); with the step tail, behind it (4).
        5 (two batches ahead, an experiment): handed to the NEXT step, which issues it behind its gather (forward): the gather then runs at
        its alone time, the forward nets pay more than it gains."""
        if getattr(self, '_next_rays', None) is not None and where == self._prefetch_now:
            self.prefetch_samples(*self._next_rays, noise=True)
            self._next_rays = None

    # ---- occupancy update (VolumeBound.optimize, volume_bound.py:160-212) -----------------------------
    def update_occupancy(self, cur_epoch, apply=True):
        """Sample voxels, evaluate opacity = sigma * dt with the geo net, EMA-max into the opacity field and re-threshold
        the bitfield.  apply=False runs all the work but leaves the marching bitfield untouched (fixed-workload benches).
        The refresh is queued on its own stream (ARCN_OCC_ASYNC=0: on the caller's): it only needs the parameters as they are
        now, so it runs next to the following step; the next optimiser step waits for it before overwriting them, and - when
        the new bitfield is applied - so does the next marching pass."""
        cfg = self.cfg
        if cur_epoch <= 0 or cfg.epoch_optim is None or cur_epoch % cfg.epoch_optim != 0:
            return
        if not self.occ_async:
            self._refresh_occupancy(cur_epoch, apply)
            return
        main = torch.cuda.current_stream()
        side = self.occ_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._refresh_occupancy(cur_epoch, apply)
            ev = torch.cuda.Event()
            ev.record(side)
        self._occ_params_event = ev
        self._occ_state_event = ev  # readers of .bitfield / .opafield wait for this refresh
        if apply:
            self._occ_bits_event = ev  # ... and so does the marcher when the bitfield it reads was replaced
            for t in (self._bitfield, getattr(self, '_bits', None)):   # allocated on the side stream, read on the others
                if t is not None:
                    t.record_stream(main)
                    if self.aux_stream is not None:
                        t.record_stream(self.aux_stream)

    def _wait_occupancy_bits(self):
        if self._occ_bits_event is not None:
            torch.cuda.current_stream().wait_event(self._occ_bits_event)

    def _wait_occupancy_state(self):
        if self._occ_state_event is not None:
            torch.cuda.current_stream().wait_event(self._occ_state_event)

    def _refresh_occupancy(self, cur_epoch, apply):
        cfg, fld = self.cfg, self.field
        dev = fld.device
        ng = cfg.n_grid
        n_cells = ng ** 3
        n_dev = None
        vs = cfg.side / ng
        warm = cfg.epoch_optim_warmup is not None and cur_epoch < cfg.epoch_optim_warmup
        from .geometry.volume import refresh_tape
        tape, perm, uni = refresh_tape(), None, None
        if tape is not None:     # the draws of a recorded run instead of the seeded generators (geometry/volume.py:set_refresh_tape)
            perm, uni = tape.draws(cur_epoch, n_cells, dev)
        if tape is None and not warm and ng >= 16 and ng & (ng - 1) == 0 and self.bitfield.data_ptr() % 8 == 0:
            # cells (n / 4 uniform + the first n / 4 occupied, both in flat order) and their jittered points by four small launches
            from .geometry.volume import mix_constants
            rng = self._cached('np_rng', lambda: np.random.default_rng(12345))
            rb = self._cached('refresh_native', lambda: {
                'cells': torch.zeros(2 * (n_cells // 4), dtype=torch.int64, device=dev),
                'pts': torch.zeros((2 * (n_cells // 4), 3), dtype=torch.float32, device=dev),
                'n_valid': torch.zeros(1, dtype=torch.int32, device=dev),
                'ws': torch.empty(n_cells + 8 * (n_cells // 4096 + 2), dtype=torch.uint8, device=dev)})
            F.refresh_cells_points(self.bitfield, ng, mix_constants(n_cells, rng), vs, fld.min_xyz, int(rng.integers(0, 1 << 62)),
                                   int(rng.integers(0, 1 << 62)) * 2 + 1, rb['cells'], rb['pts'], rb['n_valid'], rb['ws'])
            cell, pts, n_dev = rb['cells'], rb['pts'], rb['n_valid']
        else:
            if warm:
                cell = self._cached('arange_cells', lambda: torch.arange(n_cells, device=dev))
            else:
                cell, n_dev = self._select_cells(n_cells, perm)
            ix = torch.div(cell, ng * ng, rounding_mode='floor')
            iy = torch.div(cell, ng, rounding_mode='floor') % ng
            iz = cell % ng
            idx3 = torch.stack([ix, iy, iz], -1).float()
            mn = self._cached('mn', lambda: torch.tensor(fld.min_xyz, device=dev))
            pts = idx3 * vs + 0.5 * vs + mn
            pts = pts + ((torch.rand_like(pts) if uni is None else uni[:pts.shape[0]]) - 0.5) * vs
        n = pts.shape[0]
        if self._occ_scratch is None or self._occ_scratch['feat'].shape[0] < n:
            self._occ_scratch = {
                'feat': torch.empty((n, cfg.n_levels * cfg.n_feat_per_entry), dtype=torch.float32, device=dev),
                'geo_out': torch.zeros((n, fld.geo_out_dim), dtype=torch.float32, device=dev),
                'cell_max': torch.empty(n_cells, dtype=torch.float32, device=dev),
                'touched': torch.empty(n_cells, dtype=torch.uint8, device=dev),
            }
        sc = self._occ_scratch
        pts = pts.contiguous()
        if self.level_major:
            L, st, cap = N.lib(), N.stream(), sc['feat'].shape[0]
            nd = None if n_dev is None else n_dev.data_ptr()
            N.check(L.arcn_hashgrid_fwd_xcd(N.ptr(pts), N.ptr(self._p('table')), N.C.addressof(fld.grid_desc), N.ptr(sc['feat']), 1,
                                            cap, n, nd, st), 'hashgrid_fwd_xcd(occ)')
            N.check(L.arcn_mlp_fwd_lm(N.ptr(sc['feat']), cap, N.ptr(self._p('geo_w')), N.C.addressof(fld.geo_desc),
                                      N.ptr(sc['geo_out']), None, cap, n, nd, st), 'mlp_fwd_lm(occ)')
        else:
            F.hashgrid_fwd(pts, self._p('table'), fld.grid_desc, out=sc['feat'][:n], n_dev=n_dev)
            F.mlp_fwd(sc['feat'][:n], self._p('geo_w'), self._p('geo_b'), fld.geo_desc, out=sc['geo_out'][:n], n_dev=n_dev)
        if n_dev is not None:
            opacity = F.act_col_scale(sc['geo_out'][:n], cfg.sigma_act, cfg.dt, n_dev=n_dev, out=self._cached('occ_opacity', lambda: torch.zeros(n, dtype=torch.float32, device=dev)))
        else:
            sigma = F.act_fwd(sc['geo_out'][:n, 0].contiguous(), cfg.sigma_act)
            opacity = sigma * cfg.dt  # get_est_opacity (base_3d_model.py:386-389)
        F.opafield_scatter_update(self.opafield, cell, opacity, ema=cfg.ema_optim_decay, cell_max=sc['cell_max'],
                                  touched=sc['touched'], n_dev=n_dev)
        new_bits = torch.empty_like(self.bitfield)
        F.update_bitfield_by_opafield(self.opafield, new_bits, cfg.opa_thres)
        if apply:
            if self.occupancy_sync is not None:
                # data-parallel training: every rank marches rank 0's occupancy, like the buffers DistributedDataParallel re-broadcasts
                # on every forward (common/trainer/basic_trainer.py:198, broadcast_buffers) - here once per APPLIED refresh, on the
                # refresh's own stream (distributed.broadcast_occupancy: the opacity field + the bool bitfield, 10 MiB at 128^3)
                self.occupancy_sync(self._opafield, new_bits)
            self.set_bitfield(new_bits)


# ----------------------------------------------------------------------------------------------------
# synthetic workload: Blender-like cameras on a sphere, blob occupancy
# ----------------------------------------------------------------------------------------------------
def synthetic_rays(n_rays, seed=0, device='cuda', radius=3.0 / 1.05, hw=800, camera_angle_x=0.6911):
    """Pinhole rays from cameras uniform on a sphere looking at the origin (center_pixel=True)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    focal = 0.5 * hw / math.tan(0.5 * camera_angle_x)
    c = torch.randn(n_rays, 3, generator=g)
    c = c / c.norm(dim=-1, keepdim=True) * radius
    fwd = -c / c.norm(dim=-1, keepdim=True)
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = torch.cross(fwd, up, dim=-1)
    right = right / (right.norm(dim=-1, keepdim=True) + 1e-8)
    up2 = torch.cross(right, fwd, dim=-1)
    px = (torch.rand(n_rays, 2, generator=g) * hw).floor() + 0.5
    x = (px[:, 0:1] - hw / 2) / focal
    y = (px[:, 1:2] - hw / 2) / focal
    d = fwd + x * right - y * up2
    d = d / d.norm(dim=-1, keepdim=True)
    return c.float().to(device).contiguous(), d.float().to(device).contiguous()


def morton3d(x, y, z):
    """instant-ngp's cell order of the Morton density grids (bitfield_func / multivol_func, arcnerf/ops/include/volume_func.h:142-160):
    bit i of x, y, z -> bits 3i, 3i+1, 3i+2.  numpy integer arrays, coordinates < 1024."""
    def spread(v):
        v = v.astype(np.uint32) & 0x3ff
        v = (v | (v << 16)) & 0x30000ff
        v = (v | (v << 8)) & 0x300f00f
        v = (v | (v << 4)) & 0x30c30c3
        return (v | (v << 2)) & 0x9249249
    return spread(x) | (spread(y) << 1) | (spread(z) << 2)


def synthetic_cascade_bits(n_grid=128, n_levels=4, frac=0.05, seed=0):
    """Packed occupancy bits of a MultiVol cascade (level after level, Morton order inside a level, 8 cells per byte, LSB first): every
    level gets its own union of blobs filling about `frac` of it - the converged state of a pruned background, like synthetic_bitfield
    for the foreground volume."""
    ax = np.arange(n_grid)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing='ij')
    order = morton3d(X.reshape(-1), Y.reshape(-1), Z.reshape(-1)).astype(np.int64)
    out = []
    for lv in range(n_levels):
        bf = synthetic_bitfield(n_grid, frac, seed=seed + 17 * lv + 1).reshape(-1)
        cells = np.zeros(n_grid ** 3, bool)
        cells[order] = bf
        out.append(np.packbits(cells, bitorder='little'))
    return np.concatenate(out)


def synthetic_bitfield(n_grid=128, frac=0.05, seed=0):
    """Union of a few boxes / spheres filling about `frac` of the grid (numpy, flat index x*n*n+y*n+z)."""
    rng = np.random.default_rng(seed)
    ax = (np.arange(n_grid) + 0.5) / n_grid * 2 - 1
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing='ij')
    bf = np.zeros((n_grid,) * 3, bool)
    while bf.mean() < frac:
        c = (rng.random(3) - 0.5) * 1.0
        r = rng.random() * 0.2 + 0.08
        if rng.random() < 0.5:
            bf |= ((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) < r * r
        else:
            bf |= (np.abs(X - c[0]) < r) & (np.abs(Y - c[1]) < r * 0.7) & (np.abs(Z - c[2]) < r * 0.5)
    return bf
