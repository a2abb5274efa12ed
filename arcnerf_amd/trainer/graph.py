"""One training iteration of the drop-in module path as a HIP graph.

`model(feed_in) -> loss -> backward -> FusedAdam.step` of configs/nerf_ngp.yaml is ~45 kernel launches of 5 - 100 us plus the Python of
the module tree: issued eagerly the host needs 0.7 ms for a step whose kernels take 0.6.  Every buffer of the packed
path sits at its capacity and every count lives on the device, so the whole iteration can be RECORDED once (`torch.cuda.graph`) and
replayed with one launch.  What a replay cannot take from the host it reads from device memory that is rewritten before each replay:
the optimiser's step-dependent scalars (FusedAdam.prepare_step) and the sampler's generator state (NgpPipeline.prepare_replay).
What stays outside the graph, eagerly, in the reference trainer's order: `model.optimize(epoch)` (the occupancy refresh - its result is
written INTO the buffers the recorded marcher reads), the dynamic-batch measurement (one device copy), and the capacity check of the
packed buffers (the sample total travels to pinned memory behind an event and is read a step later, like the eager path).

A graph is recorded per batch shape; a batch size the dynamic batch size has not produced before records a new one (a few ms, once).
"""
import os
import time
import warnings

import torch

from ..models.nerf_model import NeRF
from ..optim import FusedAdam
from ..utils.replay import copy_words


class GraphedTrainStep:
    def __init__(self, model, loss_factory, optimizer, ema=None, warmup=2, total_epoch=300000):
        """model: a FullModel whose foreground is the packed instant-ngp NeRF and which has no background model; optimizer: a
        flattened FusedAdam (its EMA fused: pass the trainer.EMA object as `ema` only to keep its counter in step)."""
        fg = model.fg_model
        if not (isinstance(fg, NeRF) and fg.packed_path_eligible() and model.bkg_model is None):
            raise RuntimeError('GraphedTrainStep records the packed instant-ngp module path (NeRF + volume / bitfield bound, no background model)')
        if not (isinstance(optimizer, FusedAdam) and optimizer._flat is not None):
            raise RuntimeError('GraphedTrainStep needs a FusedAdam with flatten()')
        self.model, self.fg, self.loss_factory, self.opt, self.ema = model, fg, loss_factory, optimizer, ema
        self.warmup, self.total_epoch = int(warmup), total_epoch
        self.graphs = {}           # (n_rays, pipeline id, capacity) -> (graph, static inputs, static outputs, static loss dict)
        self._eager_left = self.warmup
        self._pending = []         # (pinned slot, capacity, n_rays) of the previous steps' sample totals, oldest first
        self._host_total = torch.zeros(256, dtype=torch.int32).pin_memory()
        self._np_total = self._host_total.numpy()
        self._slot = 0
        self.opt.enable_replay()
        self.replays = 0
        # steps the host may run ahead of the device.  Measured (bench.py --config ngp_module, 200 steps): 2 -> 0.761 ms per step, 6 - 12 ->
        # 0.741 - 0.745, 16 and more -> 4.5 - 4.8 ms: past ~150 queued operations the stream falls into a slow path (every step then takes
        # six times as long ON THE DEVICE), so the recorder never lets the queue get that deep
        self.max_ahead = int(os.environ.get('ARCN_GRAPH_MAX_AHEAD', '8'))
        self.host_s = {}           # host seconds spent per phase (diagnostics)

    # ---- pieces of one iteration ---------------------------------------------------------------------------------------------------------
    def _eager(self, feed_in, epoch):
        out = self.model(feed_in, get_progress=False, cur_epoch=epoch, total_epoch=self.total_epoch)
        loss = self.loss_factory(feed_in, out)
        total = loss['sum'] if isinstance(loss, dict) else loss
        if not self.opt.zero_grad_on_step:
            self.opt.zero_grad()
        total.backward()
        self.opt.step()
        if self.ema is not None:
            self.ema.ema_step()
        return out, loss

    SENTINEL = -(1 << 30)

    def _check_capacity(self):
        """the packed buffers never drop a sample silently: the sample totals of the previous steps against the capacity.  A step's
        total travels to a pinned slot holding a sentinel; the host just LOOKS at the slot - no HIP event and no memcpy operation (both stall a stream
        of graph launches on this stack, utils/replay.py).  Slots that have arrived are read; the
        host is held back (one stream synchronisation) only when it is more than `max_ahead` steps ahead of the device."""
        while self._pending:
            slot, cap, n_rays = self._pending[0]
            need = int(self._np_total[slot])
            if need == self.SENTINEL:
                if len(self._pending) <= self.max_ahead:
                    return
                torch.cuda.current_stream().synchronize()      # (a stream wait, once per max_ahead steps: the device idles for one host iteration)
                need = int(self._np_total[slot])
            self._pending.pop(0)
            self.fg._samples_per_ray = need / max(1, n_rays)
            if need >= cap and cap == self.fg._pipe.cap:
                warnings.warn('packed NGP path (recorded step): a step filled the sample buffers ({} of {}); growing them and recording '
                              'the step again'.format(need, cap))
                torch.cuda.synchronize()
                self._pending.clear()
                self.fg._packed_pipeline(self.fg._pipe.field.device, min_samples=(need * 3 // 2 + 1023) // 1024 * 1024)
                self.graphs.clear()
                self._eager_left = 1

    def _after(self, n_rays):
        pipe = self.fg._pipe
        self.fg.adjust_dynamicbs_factor(n_valid=pipe.n_dev[0])
        slot = self._slot
        self._slot = (slot + 1) % self._host_total.numel()
        self._np_total[slot] = self.SENTINEL
        copy_words(pipe.n_dev, self._host_total[slot:slot + 1])        # (a kernel writing pinned host memory: no memcpy operation)
        self._pending.append((slot, pipe.cap, n_rays))

    # ---- the iteration ----------------------------------------------------------------------------------------------------------------------
    def __call__(self, feed_in, epoch=0):
        """feed_in: the reference's dict (rays_o / rays_d / rays_r / img / bkg_color ..., (B, N, ...)).  Returns (output, loss) like
        trainer.step_optimize; after the warm-up steps the tensors are the graph's static outputs (valid until the next call)."""
        t_call = time.perf_counter()
        self._check_capacity()
        self.host_s['check'] = self.host_s.get('check', 0.0) + time.perf_counter() - t_call
        dev = feed_in['rays_o'].device
        n_rays = feed_in['rays_o'].shape[0] * feed_in['rays_o'].shape[1]
        if self._eager_left > 0:           # the first steps size the buffers (all-ones occupancy: R * n_sample samples) with host reads
            self._eager_left -= 1
            out, loss = self._eager(feed_in, epoch)
            self._after(n_rays)
            return out, loss
        pipe = self.fg._packed_pipeline(dev)      # applies a refreshed occupancy (in place once enable_replay() ran)
        if pipe.replay is None:
            pipe.enable_replay()
        # a recorded launch is bound to the pipeline's buffers AND to the optimiser's flat buffers (FusedAdam.load_state_dict flattens
        # again into new ones): either moving invalidates every graph
        fb0 = self.opt._flat[0] if getattr(self.opt, '_flat', None) else None
        home = (getattr(pipe, 'generation_id', None) or id(pipe), pipe.cap, None if fb0 is None else (fb0['params'].data_ptr(), fb0['grads'].data_ptr()))
        if getattr(self, '_home', home) != home:
            self.graphs.clear()
        self._home = home
        key = tuple(sorted((k, tuple(v.shape)) for k, v in feed_in.items() if torch.is_tensor(v)))
        rec = self.graphs.get(key)
        self.opt.prepare_step()
        pipe.prepare_replay()
        if rec is None:
            static_in = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in feed_in.items()}
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self.model(static_in, get_progress=False, cur_epoch=epoch, total_epoch=self.total_epoch)
                loss = self.loss_factory(static_in, out)
                total = loss['sum'] if isinstance(loss, dict) else loss
                if not self.opt.zero_grad_on_step:
                    self.opt.zero_grad()
                total.backward()
                self.opt.step()
            rec = self.graphs[key] = (g, static_in, out, loss)
            # (the capture does not execute: run the recorded step once now - it IS this call's step)
        g, static_in, out, loss = rec
        for k, v in feed_in.items():
            if torch.is_tensor(v) and static_in[k] is not v:
                copy_words(v.contiguous(), static_in[k]) if (v.element_size() * v.numel()) % 4 == 0 else static_in[k].copy_(v, non_blocking=True)
        t_r = time.perf_counter()
        g.replay()
        self.host_s['replay'] = self.host_s.get('replay', 0.0) + time.perf_counter() - t_r
        self.host_s['before_replay'] = self.host_s.get('before_replay', 0.0) + t_r - t_call
        self.replays += 1
        if self.ema is not None:
            self.ema.ema_step()
        self._after(n_rays)
        self.host_s['call'] = self.host_s.get('call', 0.0) + time.perf_counter() - t_call
        return out, loss
