"""Data-parallel training of the hot path: one process per GPU, rays sharded across ranks, ONE all-reduce of the flat
gradient buffer per step over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

What the reference does: DistributedDataParallel buckets + per-forward broadcast of ~100 MB of
Volume buffers, an image-level DistributedSampler.  What this does instead:
  - rays, not images, are sharded (`shard_range`), optionally balanced by last step's per-ray sample counts
    (`balanced_shards`) because the dynamic-batch-size regime makes rays unequal;
  - parameters live in one flat fp32 buffer (NgpField.params), so gradients are one contiguous tensor:
    `allreduce_grads` is a single collective (48.8 MB for the NGP config; the 1/world factor is folded into the fused
    Adam kernel's grad_scale, no extra pass);
  - the occupancy grid is refreshed from the same seed on every rank or, when ranks may diverge, only the packed
    256 KiB bitfield is broadcast (`broadcast_bitfield`) — never the lattice buffers.
There is no data-path collective: rays are independent, a ray's samples never leave its rank.
"""
import os

import torch
import torch.distributed as dist


def forced():
    """ARCN_DIST_FORCE=1: build the process group and issue every collective even at world size 1 (a one-rank RCCL communicator:
    the whole distributed code path - communicator setup, the collectives' stream hand-over, the segmented gradient sync - runs on a
    one-GPU box; tests/test_gpu_distributed.py)."""
    return os.environ.get('ARCN_DIST_FORCE', '0') == '1'


def _active(group=None):
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or forced())


def env_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def init_from_env(backend=None, device=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun style).
    Returns (rank, world).  No-op for world == 1 (unless ARCN_DIST_FORCE=1)."""
    rank, world, local = env_world()
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl' and device is not None:
            kw['device_id'] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_range(n_total, rank, world):
    """Contiguous [lo, hi) slice of n_total rays for `rank`; sizes differ by at most one, every ray exactly once."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def balanced_shards(counts, world):
    """Split rays into `world` contiguous shards of (nearly) equal expected SAMPLE count.

    counts: 1-D tensor of per-ray sample counts of the previous step (same rays or same distribution).
    Returns a list of `world + 1` boundaries (python ints), boundaries[k]..boundaries[k+1] is rank k's slice."""
    c = counts.to(torch.float64).cpu()
    n = c.numel()
    if n == 0:
        return [0] * (world + 1)
    csum = torch.cumsum(c + 1e-9, 0)  # strictly increasing so searchsorted is well defined
    total = float(csum[-1])
    targets = torch.tensor([total * k / world for k in range(1, world)], dtype=torch.float64)
    cuts = torch.searchsorted(csum, targets, right=False) + 1
    bounds = [0] + [int(min(max(v, 0), n)) for v in cuts.tolist()] + [n]
    for k in range(1, len(bounds)):
        bounds[k] = max(bounds[k], bounds[k - 1])
    return bounds


def allreduce_grads(flat_grads, world=None, group=None):
    """SUM all-reduce of the flat gradient buffer in place (the average is applied by the optimiser's grad_scale)."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1 or _active(group):
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return flat_grads


class ShardedGradSync:
    """The N > 1 step without a replicated optimiser: reduce-scatter of the flat gradient, every rank runs Adam + EMA on ITS 1/N of the
    parameters (only that shard's moments are ever touched), all-gather of the updated shards.

    Same wire bytes as the all-reduce it replaces (a ring all-reduce IS a reduce-scatter followed by an all-gather: 2 (N-1)/N x 48.8 MB
    per GPU); what goes away is the optimiser pass over the whole buffer on every rank (62 us for the NGP parameter set -> 8 us at 8
    ranks) between the two halves of the exchange.  The flat buffer [hash table | geometry weights | radiance weights] is cut into N equal
    16-byte aligned shards [r * per, (r + 1) * per) plus a tail of fewer than 4 N floats, which is all-reduced and updated on every rank.
    Both collectives run in place on the flat buffers (the shard of rank r is where reduce_scatter / all_gather expect it).
    Same arithmetic as the flat form: every gradient element is summed over the ranks once, every parameter sees the same update
    (tests/test_distributed_gloo.py: bit-identical parameters on 2 and 4 ranks).

    CHECKPOINTS: a rank's Adam moments are current for ITS shard only (the other shards hold zeros or what the last gather_moments()
    brought).  Before an optimiser state is saved - the reference saves on rank 0 only (common/trainer/basic_trainer.py:416-440) - EVERY
    rank calls gather_moments(exp_avg, exp_avg_sq) (a collective; FusedAdam.gather_sharded_state() does it for an attached optimiser);
    FusedAdam.state_dict() refuses to hand out moments that are not current (`moments_current`)."""

    def __init__(self, n_params, world=None, rank=None, group=None, align=4):
        self.group = group
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world, self.rank, self.n = int(world), int(rank), int(n_params)
        self.per = (self.n // self.world) // align * align
        self.body = self.per * self.world
        self.lo, self.hi = self.rank * self.per, (self.rank + 1) * self.per
        # what this rank's optimiser updates: its shard, and the replicated tail
        self.segments = [(self.lo, self.hi)] + ([(self.body, self.n)] if self.body < self.n else [])
        self.works = []
        self.timing = False
        self._marks = None
        self.moments_current = True     # no sharded step since the last gather_moments(): every rank holds every shard's moments

    def gather_moments(self, *flat_moments):
        """all-gather of the shards of the optimiser's moment buffers, in place (exp_avg, exp_avg_sq of the flat layout): afterwards every
        rank holds the complete state - what a checkpoint needs.  A collective: every rank calls it at the same point of its program."""
        if _active(self.group) and self.per > 0:
            works = [dist.all_gather_into_tensor(b[:self.body], b[self.lo:self.hi], group=self.group, async_op=True) for b in flat_moments]
            for w in works:
                w.wait()
        self.moments_current = True

    def launch(self, flat_grads):
        """after the backward: the sums of this rank's shard land in flat_grads[lo:hi] (in place), the tail is all-reduced"""
        self.works = []
        self.moments_current = self.world == 1
        if self.timing and flat_grads.is_cuda:
            import collections
            if self._marks is None:
                self._marks = collections.deque(maxlen=1024)
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self._marks.append([e0, None])
        if not _active(self.group):
            return
        if self.per > 0:
            self.works.append(dist.reduce_scatter_tensor(flat_grads[self.lo:self.hi], flat_grads[:self.body], op=dist.ReduceOp.SUM,
                                                         group=self.group, async_op=True))
        if self.body < self.n:
            self.works.append(dist.all_reduce(flat_grads[self.body:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []

    def clear_foreign(self, flat_grads):
        """the other ranks' shards of the gradient buffer hold this rank's partial sums: cleared for the next backward (the optimiser
        clears what it reads)"""
        if self.lo > 0:
            flat_grads[:self.lo].zero_()
        if self.hi < self.body:
            flat_grads[self.hi:self.body].zero_()

    def gather(self, *flat_buffers):
        """the updated shards of every rank, in place (parameters; the EMA shadow when it is kept beside them); the current stream
        continues behind the collective (the next forward reads every shard)"""
        if _active(self.group) and self.per > 0:
            for b in flat_buffers:
                self.works.append(dist.all_gather_into_tensor(b[:self.body], b[self.lo:self.hi], group=self.group, async_op=True))
        self.wait()
        if self.timing and self._marks and self._marks[-1][1] is None and flat_buffers and flat_buffers[0].is_cuda:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._marks[-1][1] = e1

    def exposed_ms(self, last_n=None):
        """per step: launch of the reduce-scatter -> end of the all-gather, minus nothing: the sharded optimiser sits between the two (host
        read: synchronises); needs timing=True"""
        marks = [m for m in (self._marks or []) if m[1] is not None]
        marks = marks if last_n is None else marks[-last_n:]
        out = []
        for e0, e1 in marks:
            e1.synchronize()
            out.append(max(0.0, e0.elapsed_time(e1)))
        return out


class LevelGroupedGradSync:
    """The gradient exchange overlapped with the hash-grid scatter (NgpPipeline.train_step(grad_sync=...)).

    The table gradient is 99.9 % of the flat buffer and it is the LAST thing the backward produces, so a flat all-reduce starts when
    the compute is over (48.8 MB, ~0.28 ms at 8 GPUs, nothing to hide behind).  DistributedDataParallel hides its buckets
    behind the rest of the backward (common/trainer/basic_trainer.py:197-198); the equivalent here: the scatter runs in level GROUPS
    (arcn_hashgrid_bwd_lm_levels), finest levels first, and each group's slice of the flat buffer goes on the wire (`async_op=True`: on
    the communicator's stream, behind the kernels already queued) while the next group is still being scattered; the optimiser then
    updates a group's slice as soon as it has arrived while the later groups are still in flight.  Groups of 8 levels keep the
    scatter's producer at one full wave of workgroups per launch (32 per level x 8 = 256 CUs).

    The MLP weights' gradients are complete before the scatter starts and sit BEHIND the table in the flat buffer: they travel with the
    first group, whose slice is [first level of the group, end of the buffer).  Same arithmetic as one flat SUM all-reduce: every
    element is summed across ranks exactly once (tests/test_distributed_gloo.py: bit-identical)."""

    def __init__(self, field, boundaries=(8,), group=None):
        """field: an NgpField (level offsets + flat layout); boundaries: the first level of every group but the last, descending
        (default (8,): levels 8..L-1 + everything behind the table first, then levels 0..7)"""
        self.group = group
        L = len(field.resolutions)
        Fq = field.cfg.n_feat_per_entry
        t_lo, t_n = field._seg['table']
        if t_lo != 0:
            raise ValueError('LevelGroupedGradSync expects the table at the start of the flat buffer')
        cuts = sorted({int(b) for b in boundaries if 0 < int(b) < L}, reverse=True)
        levels_hi = L
        self.groups = []        # (level mask, lo, hi): what is scattered together and the slice that then goes on the wire
        hi = field.n_params
        for b in cuts + [0]:
            mask = sum(1 << l for l in range(b, levels_hi))
            lo = field.offsets[b] * Fq
            self.groups.append((mask, lo, hi))
            hi, levels_hi = lo, b
        # the optimiser's slices start 16-byte aligned: a group's pass begins at the next multiple of 4 floats, the (at most 3) elements in
        # front of it are updated with the following group, whose wait comes later - they have arrived by then
        self.segments = []
        ahi = field.n_params
        for _, lo, _ in self.groups:
            alo = -(-lo // 4) * 4 if lo > 0 else 0
            self.segments.append((alo, ahi))
            ahi = alo
        self.works = [None] * len(self.groups)
        # timing=True: per step two events - the end of the last group's scatter (compute stream) and the end of the last collective
        # (a probe stream that only waits for it) - whose distance is the part of the exchange nothing hid (exposed_ms)
        self.timing = False
        self._probe = None
        import collections
        self._marks = collections.deque(maxlen=1024)      # (bounded: a long run with timing on keeps the last 1024 steps)

    def launch_group(self, i, flat_grads):
        _, lo, hi = self.groups[i]
        self.works[i] = None
        last = i == len(self.groups) - 1
        t0 = None
        if self.timing and last and flat_grads.is_cuda:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record()
        if _active(self.group):
            self.works[i] = dist.all_reduce(flat_grads[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if t0 is not None:
            if self._probe is None:
                self._probe = torch.cuda.Stream(device=flat_grads.device)
            t1 = torch.cuda.Event(enable_timing=True)
            self._probe.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._probe):
                if self.works[i] is not None:
                    self.works[i].wait()
                t1.record(self._probe)
            self._marks.append((t0, t1))

    def exposed_ms(self, last_n=None):
        """per step: time from the end of the compute that produced the last group's gradient to the end of the last collective
        (host read: synchronises); needs timing=True"""
        marks = list(self._marks) if last_n is None else list(self._marks)[-last_n:]
        out = []
        for t0, t1 in marks:
            t1.synchronize()
            out.append(max(0.0, t0.elapsed_time(t1)))
        return out

    def wait(self, i):
        if self.works[i] is not None:
            self.works[i].wait()
            self.works[i] = None


def broadcast_bitfield(bits, src=0, group=None):
    """Make every rank march the same occupancy: broadcast the packed bitfield (uint8, n_grid^3/8 bytes) from `src`."""
    if _active(group):
        dist.broadcast(bits, src=src, group=group)
    return bits


def broadcast_occupancy(opafield, bitfield, src=0, group=None):
    """NgpPipeline.occupancy_sync: rank `src`'s opacity field and bool bitfield to every rank before a refreshed occupancy is applied
    (the reference's DDP re-broadcasts these buffers on every forward, common/trainer/basic_trainer.py:198).  With the same refresh seed
    and bit-identical parameters the ranks compute the same fields anyway; this makes it hold by construction.
    ORDER: these are collectives on the default communicator, which the (asynchronous) gradient collectives of the step share - every rank
    must issue them at the same point of its program.  They do when the refresh is driven by the step count (`update_occupancy(epoch,
    apply=True)` / `VolumeBound.optimize(epoch)` at `epoch % epoch_optim == 0`, the same epoch on every rank) and never by rank-local
    state such as a buffer rebuild; pass a dedicated `group` if a caller cannot guarantee that."""
    if _active(group):
        dist.broadcast(opafield, src=src, group=group)
        as_bytes = bitfield.view(torch.uint8) if bitfield.dtype == torch.bool else bitfield
        dist.broadcast(as_bytes, src=src, group=group)
    return opafield, bitfield


def broadcast_params(flat_params, src=0, group=None):
    """Initial parameter sync (what DDP does at construction)."""
    if _active(group):
        dist.broadcast(flat_params, src=src, group=group)
    return flat_params


def max_over_ranks(value, device=None, group=None):
    """max of a python float over ranks (timing in bench.py)"""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if _active(group):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
