"""The drop-in NGP training step on the packed pipeline's FUSED step.

`build_model(configs/nerf_ngp.yaml)` + `FusedAdam(...).flatten()` + the reference's ImgLoss(Huber) describe exactly what
`NgpPipeline.train_step` computes - the module path just spells it as ~45 launches (model.forward, the loss chain, autograd, the
optimiser) where the pipeline has 14: compositing + Huber loss + their backward in one kernel, the optimiser inside the scatter's
consumer, the step's tail as one launch, the marching of the NEXT batch on a second stream.  The flattened optimiser already keeps the
parameters, gradients and moments in the pipeline's flat layout [table | geometry weights | radiance weights], so this class binds an
NgpPipeline to THOSE buffers and runs the step there: the model, its state_dict, the optimiser's state_dict and `model.optimize` (the
occupancy refresh of the module's own Volume) stay what they are, inference goes through the module as before.

    stepper = FusedNgpStep(model, loss_factory, optimizer, ema)          # raises if the combination is not the NGP recipe
    output, loss = stepper(feed_in, epoch, next_feed_in=batch_of_the_next_step)   # in place of trainer.step_optimize

The first steps (an all-ones bitfield asks for R x n_sample samples) run through the module path, which sizes the sample buffers;
a step that overflows them is detected a step later (the sample total travels to pinned memory) and the buffers grow.  Such a step is
never a step on truncated rays: the compositor leaves the rays behind the fill point out (arcn_composite_packed_train, `counts`): they
render as background and send no gradient.  The loss stays the mean over ALL R rays of the batch - the left-out rays contribute their
(background vs target) term to the reported value, and the rays that fit keep the 1 / (3 R) weight they have in a complete step: the
update is the gradient of that batch mean with the left-out rays' terms constant, i.e. the fitting rays' exact gradients scaled by
R_fit / R relative to a batch made of them alone.
"""
import copy
import warnings

import torch

from ..models.base_modules.obj_bound.volume_bound import VolumeBound
from ..models.nerf_model import NeRF
from ..ops.volume_func import sampler_rng
from ..optim import FusedAdam
from ..pipeline import NgpField, NgpPipeline
from ..utils.device_copy import copy_words
from .loss import AllLoss, HuberLoss, ImgLoss
from .step import step_optimize


class FusedNgpStep:
    SENTINEL = -(1 << 30)

    @staticmethod
    def why_not(model, loss_factory, optimizer):
        """None when the combination is the NGP recipe this step implements, else the reason"""
        fg = model.fg_model
        if not (isinstance(fg, NeRF) and fg.packed_path_eligible() and model.bkg_model is None):
            return 'the model is not the packed instant-ngp NeRF without a background model'
        if not isinstance(fg.obj_bound, VolumeBound):       # (a BitfieldBound keeps Morton bits and refreshes them itself: the module path)
            return 'the object bound is not a VolumeBound'
        if not (isinstance(optimizer, FusedAdam) and optimizer._flat is not None and len(optimizer._flat) == 1):
            return 'the optimiser is not a FusedAdam with ONE flattened parameter group'
        want = [fg.coarse_geo_net.embed_fn.embeddings, fg.coarse_geo_net.layers.params, fg.coarse_radiance_net.layers.params]
        have = optimizer._flat[0]['list']
        if len(have) != 3 or any(a is not b for a, b in zip(have, want)):
            return 'the optimiser does not hold exactly [hash table, geometry weights, radiance weights]'
        if not (isinstance(loss_factory, AllLoss) and len(loss_factory.funcs) == 1 and isinstance(loss_factory.funcs[0], ImgLoss)):
            return 'the loss is not a single ImgLoss'
        il = loss_factory.funcs[0]
        if not (isinstance(il.loss, HuberLoss) and list(il.keys) == ['rgb_coarse'] and il.do_mean and not il.use_mask and il.internal_weights is None):
            return 'the ImgLoss is not the plain Huber mean on rgb_coarse'
        if optimizer.ema_decay is not None and not optimizer.ema_in_param and optimizer._flat[0]['ema'] is None:
            return 'the optimiser has an EMA decay but no shadow'
        return None

    def __init__(self, model, loss_factory, optimizer, ema=None, warmup=2, total_epoch=300000, max_rays=None, clip_value=0.0, ahead=2,
                 world_size=1, grad_sync='flat', grad_level_cuts=(8,), sync_occupancy=True):
        reason = self.why_not(model, loss_factory, optimizer)
        if reason is None and clip_value > 0.0:
            reason = 'gradient clipping (clip_value > 0) needs the gradients between backward and the optimiser: use trainer.step_optimize'
        if reason is not None:
            raise RuntimeError('FusedNgpStep: ' + reason)
        self.model, self.fg, self.loss_factory, self.opt, self.ema = model, model.fg_model, loss_factory, optimizer, ema
        self.total_epoch = total_epoch
        self._eager_left = int(warmup)
        self.pipe = None
        self._bits_key = None
        self._pending = []
        self._host_total = torch.zeros(256, dtype=torch.int32).pin_memory()
        self._np_total = self._host_total.numpy()
        self._slot = 0
        self.steps = 0
        self.samples_total = 0
        self.max_rays = max_rays        # ray capacity of the buffers (default: the model's chunk_rays, else 32768; grows with the batches)
        if self.max_rays is None:
            self.max_rays = int(self.fg.chunk_rays) if getattr(self.fg, 'chunk_rays', None) and self.fg.chunk_rays > 0 else 32768
        # batches in flight: trainer.train_epoch draws up to `ahead` batches early (FIFO of (epoch, feed_in)), their marching runs on the second
        # stream.  Two ahead, the chain has a whole step of slack: lower stream priority, persistent wavefronts (NgpPipeline.march_waves)
        self.depth = max(1, int(ahead))
        self._queue = []
        # data parallel (one process per GPU, every rank its shard of the rays; the reference wraps the model in DistributedDataParallel,
        # common/trainer/basic_trainer.py:197-198): the flat gradient is summed over the ranks - grad_sync 'flat': ONE all-reduce after the
        # backward (the default); 'levels': in level groups overlapped with the scatter (distributed.LevelGroupedGradSync); 'sharded':
        # reduce-scatter, the optimiser on this rank's 1/N, all-gather (distributed.ShardedGradSync) -, the optimiser divides by the world
        # size (FusedAdam.grad_scale = 1 / world_size, DDP's average), and a refreshed occupancy is rank 0's on every rank (DDP's
        # broadcast_buffers)
        self.world = max(1, int(world_size))
        if grad_sync not in ('flat', 'levels', 'sharded'):
            raise RuntimeError('FusedNgpStep: grad_sync must be flat, levels or sharded')
        self.grad_sync = grad_sync
        self.grad_level_cuts = tuple(grad_level_cuts)
        self.sync_occupancy = bool(sync_occupancy)
        self._sync = None
        self._bits_seen = None
        self._after_eager = False
        self.rebuilds = 0

    # ---- the pipeline on the optimiser's flat buffers --------------------------------------------------------------------------------
    def _build(self, device, n_rays, min_samples=0):
        fg, fb = self.fg, self.opt._flat[0]
        mp = fg._packed_pipeline(device)                      # the module's own pipeline: configuration + descriptors + sized buffers
        cfg = copy.copy(mp.cfg)
        il = self.loss_factory.funcs[0]
        cfg.huber_delta, cfg.loss_weight = float(il.loss.delta), float(self.loss_factory.weights[0])
        fld = NgpField.__new__(NgpField)
        src = mp.field
        fld.cfg, fld.device = cfg, device
        for k in ('resolutions', 'offsets', 'min_xyz', 'max_xyz', 'grid_desc', 'geo_desc', 'rad_desc', 'geo_dims', 'rad_dims', 'geo_out_dim', 'feat_off'):
            setattr(fld, k, getattr(src, k))
        slots = fb['slots']
        fld._seg = {'table': slots[0], 'geo_w': slots[1], 'rad_w': slots[2], 'geo_b': (0, 0), 'rad_b': (0, 0)}
        fld.n_table, fld.n_geo_w, fld.n_rad_w, fld.n_geo_b, fld.n_rad_b = slots[0][1], slots[1][1], slots[2][1], 0, 0
        fld.n_params = fb['params'].numel()
        fld.params, fld.grads = fb['params'], fb['grads']
        # (the module's own pipeline keeps >= 2^20 sample slots for 32768-ray inference chunks; every launch of the step is sized by the
        # capacity: this one holds twice what the last step asked for; _check_capacity says when it is rebuilt)
        max_rays = max(int(n_rays), int(self.max_rays or 0))
        cap = max(int(min_samples), 2 * max_rays, 1 << 16)
        self.max_rays = max_rays
        if self.pipe is not None:
            # batches marched ahead into the buffers being dropped are marched again, as the same launches of the sampler's stream
            # (volume_func_kernel.cu:283-289: one 2^32 jump per launch)
            for _ in self.pipe._prefetched:
                sampler_rng().advance((1 << 64) - (1 << 32))
            self.pipe._prefetched = []
        pipe = NgpPipeline(fld, max_rays=max_rays, max_samples=cap, packed_bits=True, prefetch_depth=self.depth)
        pipe.exp_avg, pipe.exp_avg_sq = fb['exp_avg'], fb['exp_avg_sq']
        if self.opt.ema_decay is None:
            cfg.ema_decay = None
            pipe.ema = fld.params
        else:
            cfg.ema_decay = float(self.opt.ema_decay)
            pipe.ema = fld.params if self.opt.ema_in_param else fb['ema']
        pipe.rng = sampler_rng()                                # the process-wide sampler stream, shared with the module path
        self._sync = None
        if self.world > 1 and self.grad_sync == 'levels' and pipe.level_major:
            from .. import distributed as D
            self._sync = D.LevelGroupedGradSync(fld, self.grad_level_cuts)
        elif self.world > 1 and self.grad_sync == 'sharded':
            from .. import distributed as D
            self._sync = D.ShardedGradSync(fld.n_params, self.world)
            self.opt.shard_sync = self._sync       # (state_dict() then insists on gather_sharded_state() first)
        self.pipe, self._bits_key = pipe, None
        self._pending.clear()
        self.rebuilds += 1

    def _sync_occupancy(self):
        bf = self.fg.obj_bound.volume.get_voxel_bitfield(flatten=True)
        if self.world > 1 and self.sync_occupancy and self._bits_seen != (bf.data_ptr(), bf._version):
            # a collective: decided by the Volume's state alone (every rank refreshes at the same epochs), never by this rank's buffers
            # (a rebuild of the sample buffers happens on one rank at a time)
            from .. import distributed as D
            D.broadcast_occupancy(self.fg.obj_bound.volume.get_voxel_opafield(flatten=True), bf)     # in place: the Volume's own buffers
            self._bits_seen = (bf.data_ptr(), bf._version)
        key = (id(self.pipe), bf.data_ptr(), bf._version)
        if self._bits_key != key:
            self.pipe.set_bitfield(bf)
            self._bits_key = key

    @staticmethod
    def _slots_for(need):
        """capacity built for a demand of `need` samples: twice that (the launches are sized by the capacity, but their idle workgroups
        leave at once - the headline bench runs 2.6e5 samples in 2^20 slots - while a rebuild costs milliseconds)"""
        return (int(need * 2.0) + 4095) // 4096 * 4096

    def _check_capacity(self, n_rays):
        """the sample totals of the steps that have finished (pinned memory, written by a kernel at the end of each step): the rate that
        sizes the buffers, and the report of a step that filled them"""
        while self._pending:
            slot, cap, rays = self._pending[0]
            need = int(self._np_total[slot])
            if need == self.SENTINEL:
                if len(self._pending) <= 8:
                    break
                torch.cuda.current_stream().synchronize()
                need = int(self._np_total[slot])
            self._pending.pop(0)
            self.samples_total += min(need, cap)
            self.fg._samples_per_ray = need / max(1, rays)
            if need >= cap:
                # (the compositor never renders a ray from a truncated sample set: the rays behind the fill point took no part in that
                # step - background colour, zero gradient; the loss of that step is still the mean over all its rays, so the rays that
                # fit kept their 1 / (3 R) weight and the reported loss includes the left-out rays' background term)
                warnings.warn('FusedNgpStep: a training step filled the sample buffers ({} slots for {} rays): the rays behind the fill point '
                              'were left out of that step (no gradient from them); the buffers grow now'.format(cap, rays))
                self.fg._samples_per_ray = max(self.fg._samples_per_ray, 2.0 * cap / max(1, rays))
        rate = getattr(self.fg, '_samples_per_ray', None)
        pipe = self.pipe
        if pipe is None or pipe.max_rays < n_rays:
            return True
        if pipe.field.params is not self.opt._flat[0]['params']:      # FusedAdam.flatten() ran again (load_state_dict): new flat buffers
            return True
        if rate is None:
            return False
        # grow before a step that could come within 25 % of the capacity (the batches of one run differ by a few percent in samples per
        # ray), shrink when an eighth would do (the all-ones bitfield of a fresh model against the pruned one a few hundred steps later)
        full = n_rays * pipe.cfg.n_sample
        return min(1.25 * rate * n_rays, full) > pipe.cap or 8 * self._target_cap(rate, n_rays, pipe.max_rays) <= pipe.cap

    def drain(self):
        """wait for the steps issued so far and take their sample totals in: -> samples_total, the valid samples (mask_pts.sum() of the
        reference's metric) of every fused step since construction"""
        torch.cuda.synchronize()
        if self.pipe is not None:
            self._check_capacity(1)
        return self.samples_total

    def _target_cap(self, rate, n_rays, max_rays):
        full = n_rays * self.fg.get_n_coarse_sample()
        need = full if rate is None else min(self._slots_for(rate * n_rays), full)
        return max(int(need), 2 * int(max_rays), 1 << 16)

    # ---- the batches drawn early (trainer.train_epoch) -----------------------------------------------------------------------------------------
    def can_run_ahead(self, epoch):
        """True when nothing the trainer does at `epoch` before the step can change what that step's marching reads: no refresh of the
        bound's occupancy (VolumeBound.optimize: epoch % epoch_optim == 0) - then the batch can be drawn and marched early"""
        if self._eager_left > 0 or self.pipe is None:
            return False
        every = self.fg.obj_bound.get_optim_cfgs('epoch_optim')
        return not (epoch > 0 and every is not None and epoch % every == 0)

    def ahead_room(self):
        return self.depth - len(self._queue)

    def next_ahead_epoch(self, epoch):
        """the epoch whose batch would be drawn next, given that the step of `epoch` is about to run"""
        return self._queue[-1][0] + 1 if self._queue else epoch + 1

    def hold_ahead(self, epoch, feed_in):
        self._queue.append((epoch, feed_in))

    def take_ahead(self, epoch):
        if self._queue and self._queue[0][0] == epoch:
            return self._queue.pop(0)[1]
        self._queue = []        # (a trainer that left the epoch order: the batches drawn early are dropped, the pipeline forgets their samples)
        return None

    def ahead(self):
        return [f for _, f in self._queue]

    def _module_step(self, feed_in, epoch, get_progress):
        """an iteration on the module path (the first steps of a run, progress iterations); data parallel: the flat gradient summed over the
        ranks between backward and the optimiser, which divides by the world size - what DistributedDataParallel amounts to"""
        if self.world == 1:
            return step_optimize(self.model, feed_in, self.loss_factory, self.opt, self.ema, epoch, self.total_epoch, get_progress=get_progress)
        from .. import distributed as D
        output = self.model(feed_in, get_progress=get_progress, cur_epoch=epoch, total_epoch=self.total_epoch)
        loss = self.loss_factory(feed_in, output)
        self.opt.zero_grad()
        loss['sum'].backward()
        D.allreduce_grads(self.opt.flat_grads(), self.world)
        self.opt.step()
        if self.ema is not None:
            self.ema.ema_step()
        return output, loss

    # ---- the iteration --------------------------------------------------------------------------------------------------------------------
    def __call__(self, feed_in, epoch=0, next_feed_in=None, get_progress=False):
        """one training iteration on `feed_in` ((B, N, 3) rays_o / rays_d / img [/ bkg_color]), like trainer.step_optimize: -> (output, loss).
        next_feed_in: the batch of the FOLLOWING call, or the batches of the following `ahead` calls in order (the same tensors must be passed
        then): marched on the second stream meanwhile.
        get_progress: the per-sample outputs of the reference's progress dumps exist on the module path only - that iteration runs there."""
        dev = feed_in['rays_o'].device
        n_rays = feed_in['rays_o'].shape[0] * feed_in['rays_o'].shape[1]
        if self._eager_left > 0 or get_progress:      # the module path: its exact first step sizes everything (all-ones bitfield: R x n_sample samples)
            self._eager_left = max(0, self._eager_left - 1)
            self._after_eager = True
            return self._module_step(feed_in, epoch, get_progress)
        if self.pipe is None or self._after_eager:
            self.fg._check_deferred_overflow(dev)          # (the sample total of the last eager step)
            if not self.opt.zero_grad_on_step:
                self.opt.zero_grad()                       # the step accumulates into the flat gradient and clears it in the optimiser pass
            self._after_eager = False
        if self._check_capacity(n_rays):
            rate = getattr(self.fg, '_samples_per_ray', None)
            need = self._target_cap(rate, n_rays, max(n_rays, self.max_rays))
            if self.pipe is not None:
                torch.cuda.synchronize()
            self._build(dev, n_rays, min_samples=need)
        pipe, fb, group = self.pipe, self.opt._flat[0], self.opt.param_groups[0]
        for p, (o, n) in zip(fb['list'], fb['slots']):
            if p.data_ptr() != fb['params'].data_ptr() + 4 * o:
                raise RuntimeError('FusedNgpStep: a parameter was re-assigned after FusedAdam.flatten()')
        cfg = pipe.cfg
        cfg.lr, cfg.betas, cfg.eps, cfg.weight_decay = float(group['lr']), tuple(group['betas']), float(group['eps']), float(group['weight_decay'])
        if abs(float(self.opt.grad_scale) - 1.0 / self.world) > 1e-12:
            raise RuntimeError('FusedNgpStep(world_size={}): FusedAdam.grad_scale must be 1 / world_size (the gradients are SUMMED over the '
                               'ranks), it is {}'.format(self.world, self.opt.grad_scale))
        pipe.step_count = int(fb['step'])
        pipe.ema_n_step = int(fb['step']) if self.opt.ema_n_step is None else int(self.opt.ema_n_step)
        self._sync_occupancy()
        o, d, img = feed_in['rays_o'].reshape(-1, 3), feed_in['rays_d'].reshape(-1, 3), feed_in['img'].reshape(-1, 3)
        bkg = feed_in['bkg_color'].reshape(-1, 3) if feed_in.get('bkg_color') is not None else None
        # the batches of the following calls, in order: those not marched yet are marched now - the first at the step's own prefetch point, a
        # second one (the step after a refresh, when two are drawn at once) behind the optimiser.  Marching order = call order: every batch
        # is the launch of the sampler's stream it would have been in the eager loop.
        ahead = [] if next_feed_in is None else (list(next_feed_in) if isinstance(next_feed_in, (list, tuple)) else [next_feed_in])
        todo = []
        for f in ahead[:pipe.prefetch_depth]:
            fo, fd = f['rays_o'].reshape(-1, 3), f['rays_d'].reshape(-1, 3)
            if fo.shape[0] > pipe.max_rays:
                break
            if not any(pf[0] == fo.data_ptr() and pf[1] == fd.data_ptr() and pf[2] == fo.shape[0] for pf in pipe._prefetched):
                todo.append((fo, fd))
        if self.world == 1:
            loss = pipe.train_step(o, d, img, bkg_color=bkg, next_rays=todo[0] if todo else None)
        elif self._sync is not None:
            loss = pipe.train_step(o, d, img, bkg_color=bkg, next_rays=todo[0] if todo else None, world_size=self.world, grad_sync=self._sync)
        else:
            from .. import distributed as D
            loss = pipe.train_step(o, d, img, bkg_color=bkg, next_rays=todo[0] if todo else None, world_size=self.world,
                                   all_reduce=lambda t: D.allreduce_grads(t, self.world))
        for fo, fd in todo[1:]:
            pipe.prefetch_samples(fo, fd, noise=True)
        # the counters the optimiser / EMA objects expose
        fb['step'] = pipe.step_count
        for p in fb['list']:
            self.opt.state[p]['step'] = fb['step']
        if self.opt.ema_n_step is not None:
            self.opt.ema_n_step = pipe.ema_n_step
        if self.ema is not None:
            self.ema.n_step += 1
        self.steps += 1
        # the measurement of the dynamic batch size + this step's sample total (a kernel writing pinned memory, read a step later)
        # (the count of a batch marched ahead was final steps ago, on the sampling stream: recorded and later read there, the batch-size
        # update every 16 steps does not drain the step's stream)
        aux = pipe.aux_stream if pipe.use_streams else None
        if aux is not None and not pipe.sampled_ahead:
            if pipe.sample_event is not None:
                aux.wait_event(pipe.sample_event)      # (marched inline: behind its marcher on this stream, not behind the step)
            else:
                aux = None
        self.fg.adjust_dynamicbs_factor(n_valid=pipe.n_dev[0], stream=aux)
        slot = self._slot
        self._slot = (slot + 1) % self._host_total.numel()
        self._np_total[slot] = self.SENTINEL
        copy_words(pipe.n_dev, self._host_total[slot:slot + 1])
        self._pending.append((slot, pipe.cap, n_rays))
        b, n = feed_in['rays_o'].shape[:2]
        rgb, depth = self.fg._packed_defaults(pipe.buf['rgb'][:n_rays], pipe.buf['depth'][:n_rays], pipe.buf['counts'][:n_rays], bkg)
        out = {'rgb_coarse': rgb.view(b, n, 3), 'depth_coarse': depth.view(b, n), 'mask_coarse': pipe.buf['mask'][:n_rays].view(b, n)}
        name = self.loss_factory.names[0] if hasattr(self.loss_factory, 'names') else 'ImgLoss'
        return out, {'sum': loss, 'names': [name], name: loss}
