"""torch.optim-shaped front end of the fused Adam (+ EMA) kernel.

The reference builds `torch.optim.Adam` in common/trainer/optimizer.py and applies `EMA.ema_step` afterwards
(arcnerf/trainer/ema.py:29-43: a debiased running average written back into the parameter).  torch's
multi-tensor Adam streams the 12.6 M NGP parameters through about ten kernels per step; `arcn_adam_ema_step` does Adam, the EMA
and the gradient clear in one pass (488 MB of HBM traffic, ~75 us; 390 MB with `ema_in_param`).  FusedAdam is a drop-in for torch.optim.Adam on CUDA float32
parameters: same update rule (L2 weight decay added to the gradient, bias correction, eps outside the square root), same
`state_dict` layout (`step`, `exp_avg`, `exp_avg_sq`), plus an optional fused EMA and zero-on-step."""
import torch

from .ops import functional as F
from .utils import param_epoch


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_decay=None, zero_grad_on_step=False,
                 ema_in_param=False):
        """ema_decay: None = plain Adam; a float also applies the reference's EMA.ema_step in the same pass (debiased running
        average of the parameter, stored in state['ema'] AND written back into the parameter, arcnerf/trainer/ema.py:29-43).
        zero_grad_on_step: the kernel clears .grad while it reads it (then `optimizer.zero_grad()` can be skipped).
        ema_in_param: the written-back average makes state['ema'] equal to the parameter after every step; when nothing else writes
        the parameters between steps (no clipping of weights, no reload mid-run) the shadow can be the parameter itself: no
        state['ema'], one 8 B/param sweep less, same bits."""
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError('invalid Adam hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.ema_decay = ema_decay
        self.zero_grad_on_step = zero_grad_on_step
        self.ema_in_param = bool(ema_in_param and ema_decay is not None)
        self.grad_scale = 1.0     # multiplies every gradient as it is read: 1 / world after a SUM all-reduce (DDP's average, no extra pass)
        self.ema_n_step = None    # EMA.n_step when it differs from Adam's step count (set_ema_n_step: a resumed job); None = Adam's
        self._flat = None         # per param group: dict(params, grads, exp_avg, exp_avg_sq, ema, slots) once flatten() ran
        self.shard_sync = None    # a distributed.ShardedGradSync whose steps update only this rank's shard of the moments (set by the stepper)

    # ---- one flat buffer per parameter group --------------------------------------------------------------------------------------
    def flatten(self, direct_grads=None):
        """Re-home the parameters of every group into ONE contiguous fp32 buffer per group, their gradients into one flat gradient
        buffer (`p.grad` becomes a view; autograd accumulates into it in place) and the Adam state into flat buffers too: `step()` is
        then one kernel launch per group, `flat_grads()` is what a data-parallel step all-reduces with ONE collective and no
        gather / scatter copies (the reference wraps the model in DistributedDataParallel, common/trainer/basic_trainer.py:197-198,
        whose buckets are the same idea), and `zero_grad()` is one memset.  Segments are padded to 16 bytes; the pad elements have zero
        parameter, gradient and state, which Adam leaves at zero.  state_dict() keeps torch.optim.Adam's per-parameter layout (the
        entries are views).  direct_grads: mark the parameters so that hand-written backward nodes whose kernels accumulate (the packed NGP
        render) add their gradient straight into the flat buffer instead of handing autograd a zero-filled temporary for AccumulateGrad
        to add (tensor hooks on such a parameter do not see that contribution; pass False when hooks must).  Returns self."""
        if direct_grads is None:
            direct_grads = getattr(self, '_direct_grads', None)
            if direct_grads is None:
                direct_grads = True
        self._direct_grads = bool(direct_grads)      # a later re-flatten (load_state_dict) keeps the caller's choice
        self._flat = []
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.requires_grad]
            for p in ps:
                if not (p.is_cuda and p.dtype == torch.float32):
                    raise RuntimeError('FusedAdam handles float32 CUDA parameters (there is no CPU fallback)')
            slots, off = [], 0
            for p in ps:
                slots.append((off, p.numel()))
                off += (p.numel() + 3) // 4 * 4
            dev = ps[0].device if ps else None
            mk = lambda: torch.zeros(max(off, 4), dtype=torch.float32, device=dev)   # noqa: E731
            fb = dict(list=ps, slots=slots, params=mk(), grads=mk(), exp_avg=mk(), exp_avg_sq=mk(),
                      ema=mk() if (self.ema_decay is not None and not self.ema_in_param) else None, step=0)
            with torch.no_grad():
                for p, (o, n) in zip(ps, slots):
                    fb['params'][o:o + n].copy_(p.detach().reshape(-1))
                    st = self.state[p]
                    if 'exp_avg' in st:
                        fb['exp_avg'][o:o + n].copy_(st['exp_avg'].reshape(-1))
                        fb['exp_avg_sq'][o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
                        fb['step'] = max(fb['step'], int(st['step']))
                    if p.grad is not None:
                        fb['grads'][o:o + n].copy_(p.grad.reshape(-1))
                    if fb['ema'] is not None:
                        src = st['ema'] if st.get('ema') is not None else p.detach()
                        fb['ema'][o:o + n].copy_(src.reshape(-1))
                    p.data = fb['params'][o:o + n].view(p.shape)
                    p.grad = fb['grads'][o:o + n].view(p.shape)
                    p._arcn_direct_grad = bool(direct_grads)
                    st['step'] = fb['step']
                    st['exp_avg'] = fb['exp_avg'][o:o + n].view(p.shape)
                    st['exp_avg_sq'] = fb['exp_avg_sq'][o:o + n].view(p.shape)
                    if fb['ema'] is not None:
                        st['ema'] = fb['ema'][o:o + n].view(p.shape)
            self._flat.append(fb)
        return self

    def set_ema_n_step(self, n_step):
        """EMA.set_n_step (arcnerf/trainer/ema.py:25-27; the trainer passes progress.start_epoch, arcnerf_trainer.py:70): the running
        average is de-biased with ITS step count, Adam's bias correction with the optimiser state's"""
        self.ema_n_step = int(n_step)

    def _next_ema_step(self):
        if self.ema_n_step is None:
            return None
        self.ema_n_step += 1
        return self.ema_n_step

    def flat_grads(self, group=0):
        """The flat gradient buffer of a parameter group (after flatten()): the tensor to all-reduce."""
        if self._flat is None:
            raise RuntimeError('FusedAdam.flat_grads() needs flatten() first')
        return self._flat[group]['grads']

    def flat_params(self, group=0):
        if self._flat is None:
            raise RuntimeError('FusedAdam.flat_params() needs flatten() first')
        return self._flat[group]['params']

    def zero_grad(self, set_to_none=False):
        if self._flat is None:
            return super().zero_grad(set_to_none=set_to_none)
        for fb in self._flat:      # the views stay in place (set_to_none would detach them from the flat buffer)
            fb['grads'].zero_()

    def gather_sharded_state(self):
        """Under sharded gradient sync (distributed.ShardedGradSync: every rank runs Adam on its 1/N of the flat buffer) a rank's moments are
        current for its own shard only.  Call this on EVERY rank (a collective) before any rank takes state_dict(): all-gather of the
        moment shards, in place.  No-op without an attached sharded sync."""
        if self.shard_sync is None or self._flat is None:
            return
        fb = self._flat[0]
        self.shard_sync.gather_moments(fb['exp_avg'], fb['exp_avg_sq'])

    def state_dict(self):
        if self.shard_sync is not None and not self.shard_sync.moments_current:
            raise RuntimeError('FusedAdam.state_dict(): the steps since the last gather ran with sharded gradient sync - this rank holds the '
                               'Adam moments of its own 1/{} of the parameters only.  Call optimizer.gather_sharded_state() on EVERY rank first '
                               '(a collective), then save on whichever rank saves.'.format(self.shard_sync.world))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self._flat is not None:   # the loaded per-parameter tensors replaced the views: fold them back into the flat buffers
            self.flatten()

    # ---- a step in two halves: for callers whose table scatters apply the optimiser to part of the buffer themselves ----------------------
    def begin_step(self):
        """First half of step() for a flattened single-group optimiser: the checks and the counters of THIS step, and the numbers a kernel that
        applies the update itself needs (ops.functional.hashgrid_bwd_adam: the scatter's chunk owners on the table levels they own) ->
        dict(step, ema_step, lr, betas, eps, weight_decay, ema_decay | None, grad_scale).  finish_step(exclude) must follow."""
        if self._flat is None or len(self._flat) != 1 or len(self.param_groups) != 1:
            raise RuntimeError('FusedAdam.begin_step() needs flatten() and a single parameter group')
        if self.ema_decay is not None and not self.ema_in_param:
            raise RuntimeError('FusedAdam.begin_step(): a fused EMA needs ema_in_param (the shadow aliased onto the parameter)')
        param_epoch.bump()
        group, fb = self.param_groups[0], self._flat[0]
        for p, (o, n) in zip(fb['list'], fb['slots']):
            if p.grad is None or p.grad.data_ptr() != fb['grads'].data_ptr() + 4 * o or p.data_ptr() != fb['params'].data_ptr() + 4 * o:
                raise RuntimeError('FusedAdam (flat): a parameter or its .grad was re-assigned after flatten(); '
                                   'use zero_grad(set_to_none=False) and in-place updates, or call flatten() again')
        ema_step = self._next_ema_step()
        fb['step'] += 1
        for p in fb['list']:
            self.state[p]['step'] = fb['step']
        return dict(step=fb['step'], ema_step=ema_step, lr=group['lr'], betas=group['betas'], eps=group['eps'], weight_decay=group['weight_decay'],
                    ema_decay=self.ema_decay, grad_scale=self.grad_scale)

    def finish_step(self, hyper, exclude=()):
        """Second half: Adam (+ EMA) on everything of the flat buffer OUTSIDE the float ranges `exclude` = [(lo, hi), ...] (already updated by
        the caller's kernels with the numbers of begin_step()); ranges start and end on multiples of 4 floats."""
        fb = self._flat[0]
        total = fb['params'].numel()
        runs, at = [], 0
        for lo, hi in sorted((int(a), int(b)) for a, b in exclude if b > a):
            if lo % 4 or hi % 4 or lo < at:
                raise RuntimeError('FusedAdam.finish_step(): excluded ranges must be disjoint and 4-float aligned')
            if lo > at:
                runs.append((at, lo))
            at = hi
        if at < total:
            runs.append((at, total))
        param_epoch.bump()      # (again: anything cached between begin_step() and here was computed from the old parameters)
        for k in range(0, len(runs), 4):
            F.adam_ema_step_runs(fb['params'], fb['grads'], fb['exp_avg'], fb['exp_avg_sq'], fb['params'] if self.ema_in_param else None, runs[k:k + 4],
                                 hyper['step'], lr=hyper['lr'], betas=hyper['betas'], eps=hyper['eps'], weight_decay=hyper['weight_decay'],
                                 ema_decay=self.ema_decay if self.ema_decay is not None else 0.0, grad_scale=hyper['grad_scale'],
                                 ema_step=hyper['ema_step'], zero_grad=self.zero_grad_on_step)

    def table_views(self, p):
        """(exp_avg, exp_avg_sq, first float of p in the flat buffers) of a parameter of the flattened group"""
        fb = self._flat[0]
        for q, (o, n) in zip(fb['list'], fb['slots']):
            if q is p:
                return fb['exp_avg'][o:o + n], fb['exp_avg_sq'][o:o + n], o
        raise KeyError('not a parameter of this optimiser')

    def _step_flat(self):
        if len(self.param_groups) != len(self._flat):
            raise RuntimeError('FusedAdam (flat): a parameter group was added after flatten(); call flatten() again')
        ema_step = self._next_ema_step()
        for group, fb in zip(self.param_groups, self._flat):
            for p, (o, n) in zip(fb['list'], fb['slots']):
                if p.grad is None or p.grad.data_ptr() != fb['grads'].data_ptr() + 4 * o or p.data_ptr() != fb['params'].data_ptr() + 4 * o:
                    raise RuntimeError('FusedAdam (flat): a parameter or its .grad was re-assigned after flatten(); '
                                       'use zero_grad(set_to_none=False) and in-place updates, or call flatten() again')
            fb['step'] += 1
            for p in fb['list']:
                self.state[p]['step'] = fb['step']
            F.adam_ema_step(fb['params'], fb['grads'], fb['exp_avg'], fb['exp_avg_sq'], fb['params'] if self.ema_in_param else fb['ema'],
                            fb['step'], lr=group['lr'], betas=group['betas'], eps=group['eps'], weight_decay=group['weight_decay'],
                            ema_decay=self.ema_decay if self.ema_decay is not None else 0.0, grad_scale=self.grad_scale,
                            ema_step=ema_step, zero_grad=self.zero_grad_on_step)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        param_epoch.bump()      # (the kernels write the parameters through raw pointers: caches keyed on tensor versions must hear of it)
        if self._flat is not None:
            self._step_flat()
            return loss
        ema_step = self._next_ema_step()
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                        and p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0):
                    raise RuntimeError('FusedAdam handles contiguous, 16-byte aligned float32 CUDA parameters (there is no CPU fallback)')
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                if self.ema_decay is not None and not self.ema_in_param and st.get('ema') is None:
                    # also after load_state_dict of a checkpoint written by torch.optim.Adam (step / exp_avg / exp_avg_sq only)
                    st['ema'] = p.detach().clone()
                st['step'] = int(st['step']) + 1
                F.adam_ema_step(p, p.grad, st['exp_avg'], st['exp_avg_sq'], p if self.ema_in_param else st.get('ema'), st['step'], lr=group['lr'],
                                betas=group['betas'], eps=group['eps'], weight_decay=group['weight_decay'],
                                ema_decay=self.ema_decay if self.ema_decay is not None else 0.0, grad_scale=self.grad_scale,
                                ema_step=ema_step, zero_grad=self.zero_grad_on_step)
        return loss
