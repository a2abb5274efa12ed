"""The drop-in training step of BASELINE config 4 - NeuS on the hash grid in a pruned volume + the MultiVol background
(configs/neus_ngp_multivol.yaml, the model block of the reference's capture_qqtiger_neusngp_multivol.yaml) - as a hand-ordered chain of
the HIP kernels, without the autograd engine between them.

`build_model(configs/neus_ngp_multivol.yaml)` + `FusedAdam(...).flatten()` + the yaml's losses (ImgLoss Huber on `rgb`, EikonalLoss on
`normal_pts`) describe a fixed computation; the module path spells it as ~200 launches per step (a third of them torch fills, adds, copies
and concatenations between the kernels, 25 % of the kernel time) issued through autograd, and the step is HOST bound: 2.9 ms for 1.75 ms
of hand-written kernels on the critical path.  This class runs the same kernels in the same order on the same tensors - the forward
chain, the loss gradients, then every node's backward by hand - writes the parameter gradients straight into the flattened optimiser's
buffers and calls the optimiser: the model, its state_dict, `model.optimize` and inference through the module stay what they are.

    stepper = FusedNeusNgpStep(model, loss_factory, optimizer)           # raises if the combination is not the config-4 recipe
    output, loss = stepper(feed_in, epoch, next_feed_in=batch_of_the_next_step)   # in place of trainer.step_optimize (or a list: the next TWO batches)

What the chain is (reference: neus_model.py:63-104, sdf_model.py:42-101, base_network.py:30-44, multivol_bkg_model.py:74-148,
full_model.py:278-330):
  foreground   march (K2 + K3) -> section layout -> hash encode -> sdf net 32 -> 64 (softplus 100) -> 1 + 16 WITH the Jacobian row of its first
               output (ops.autograd.SdfMlpJacFn's arithmetic) -> normals = (d enc / d x)^T J -> radiance net [p | SH(v) | n | f] 38 -> 64 -> 64 -> 3 ->
               the NeuS render kernel (slope, cos annealing, sdf_to_alpha, weights, sums, defaults) with the last transmittance
  background   cascade march (K11) -> compaction -> hash encode (side 24) -> density net 32 -> 64 (ReLU) -> 1 + 16 (TruncExp density) -> radiance net
               [f | SH(v)] 32 -> 64 -> 64 -> 3 -> packed compositor
  blend        rgb = rgb_fg + T_last rgb_bkg, depth likewise (`rgb` blending)
  loss         ImgLoss (Huber | MSE, plain mean) on rgb + EikonalLoss (MSE, plain mean) on the dense normal_pts - the latter evaluated on the
               PACKED normals with the dense layout's weights (arcn_eikonal_packed): the (rays, P, 3) tensor is never built
  backward     every node in reverse, second-order pieces included: the normals' gradient reaches the table through arcn_hashgrid_bwd_bwd and
               the sdf net's weights through the gradient of J.
The samplers of the next one or two batches run on a second stream meanwhile (_march_ahead).
`output` holds rgb / depth / mask / normal and params like the module's; `normal_pts` is not materialised (the loss it exists for is inside).
"""
import torch

from ..models.base_modules.encoding.hashgrid_encoder import HashGridEmbedder
from ..models.multivol_bkg_model import MultiVol
from ..models.neus_model import Neus
from ..ops import functional as F
from ..optim import FusedAdam
from .loss import AllLoss, EikonalLoss, HuberLoss, ImgLoss


def _sdf_net_form(geo):
    """(embedder, first layer, last layer) when the geometry net is the one GeoNet._jacobian_path handles, else None"""
    import torch.nn as nn
    from ..models.base_modules.geo_rad_model.linear_network_module import DenseLayer, GeoNet
    from ..models.base_modules.linear import Linear
    if not isinstance(geo, GeoNet) or geo.D != 1 or geo.skips or geo.W_feat <= 0 or geo.out_act is not None:
        return None
    emb, l0, l1 = geo.embed_fn, geo.layers[0], geo.layers[1]
    if type(emb) is not HashGridEmbedder or emb.include_input:
        return None
    if type(l0) is not DenseLayer or type(l0.activation) is not nn.Softplus or l0.activation.threshold != 20 or type(l1) is not Linear:
        return None
    if l0.bias is not None or l1.bias is not None or hasattr(l0, 'weight_g') or hasattr(l1, 'weight_g'):
        return None
    return emb, l0, l1


def _density_net_form(geo):
    """(embedder, first layer, last layer) when the geometry net is hash grid -> one ReLU layer -> linear [sigma | feat] with a TruncExp
    density, without biases (the background nets of configs/neus_ngp_multivol.yaml), else None"""
    import torch.nn as nn
    from ..models.base_modules.geo_rad_model.linear_network_module import DenseLayer, GeoNet
    from ..models.base_modules.linear import Linear
    if not isinstance(geo, GeoNet) or geo.D != 1 or geo.skips or geo.W_feat <= 0 or type(geo.out_act).__name__ != 'TruncExp':
        return None
    emb, l0, l1 = geo.embed_fn, geo.layers[0], geo.layers[1]
    if type(emb) is not HashGridEmbedder or emb.include_input:
        return None
    if type(l0) is not DenseLayer or type(l0.activation) is not nn.ReLU or type(l1) is not Linear:
        return None
    if l0.bias is not None or l1.bias is not None or hasattr(l0, 'weight_g') or hasattr(l1, 'weight_g'):
        return None
    return emb, l0, l1


def _flat_view(tensors):
    """the tensors as ONE flat tensor when they sit back to back in memory (consecutive parameters of a flattened optimiser whose sizes
    are multiples of four floats), else None"""
    t0 = tensors[0]
    ptr = t0.data_ptr()
    for t in tensors:
        if t.data_ptr() != ptr or not t.is_contiguous():
            return None
        ptr += 4 * t.numel()
    n = sum(t.numel() for t in tensors)
    try:
        return t0.as_strided((n,), (1,), t0.storage_offset())
    except RuntimeError:
        return None


class FusedNeusNgpStep:
    @staticmethod
    def why_not(model, loss_factory, optimizer):
        from ..models.base_modules.encoding.sh_encoder import SHEmbedder
        from ..models.base_modules.geo_rad_model.linear_network_module import RadianceNet
        fg, bkg = model.fg_model, model.bkg_model
        if not (isinstance(fg, Neus) and fg.packed_path_eligible()):
            return 'the foreground is not the packed NeuS (occupancy-marched volume, no importance sampling)'
        if _sdf_net_form(fg.geo_net) is None:
            return 'the sdf net is not hash grid -> one softplus layer -> linear, without biases'
        r = fg.radiance_net
        if not (isinstance(r, RadianceNet) and r._fused_desc is not None and r.mode == 'pvnf' and isinstance(r.embed_fn_view, SHEmbedder)
                and not r.embed_fn_view.include_input and r.embed_fn_pts.get_output_dim() == 3):
            return 'the radiance net is not the bias-free pvnf stack of widths <= 64 on [p | SH(v) | n | f]'
        if not (isinstance(bkg, MultiVol) and bkg.use_packed_path and model.bkg_blend == 'rgb' and not model.fg_only):
            return 'the background is not a MultiVol blended by rgb'
        rb = bkg.radiance_net
        if not (_density_net_form(bkg.geo_net) is not None and isinstance(rb, RadianceNet) and rb._fused_desc is not None and rb.mode == 'fv'
                and isinstance(rb.embed_fn_view, SHEmbedder) and not rb.embed_fn_view.include_input):
            return 'the background nets are not hash grid -> ReLU layer -> [TruncExp density | features] + the bias-free fv radiance stack of widths <= 64'
        if float(bkg.get_ray_cfgs('noise_std') or 0.0) > 0 or float(fg.get_ray_cfgs('noise_std') or 0.0) > 0:
            return 'density noise is on'
        if not (isinstance(optimizer, FusedAdam) and optimizer._flat is not None):
            return 'the optimiser is not a flattened FusedAdam'
        if any(p.grad is None for p in model.parameters() if p.requires_grad):
            return 'a parameter has no gradient buffer (FusedAdam.flatten() gives every parameter a view of the flat one)'
        if not isinstance(loss_factory, AllLoss) or sorted(type(f).__name__ for f in loss_factory.funcs) != ['EikonalLoss', 'ImgLoss']:
            return 'the loss is not ImgLoss + EikonalLoss'
        il = next(f for f in loss_factory.funcs if isinstance(f, ImgLoss))
        el = next(f for f in loss_factory.funcs if isinstance(f, EikonalLoss))
        if not (list(il.keys) == ['rgb'] and il.do_mean and not il.use_mask and il.internal_weights is None
                and (isinstance(il.loss, HuberLoss) or type(il.loss).__name__ == 'MSELoss')):
            return 'the ImgLoss is not the plain Huber / MSE mean on rgb'
        if not (el.key == 'normal_pts' and el.do_mean and not el.use_mask and type(el.loss).__name__ == 'MSELoss'):
            return 'the EikonalLoss is not the plain MSE mean on normal_pts'
        return None

    def __init__(self, model, loss_factory, optimizer, ema=None, total_epoch=300000, prefetch=True, world_size=1, grad_sync='flat',
                 sync_occupancy=True, keep_corners=True, fuse_adam=True, fused_geo=True, march_at='opt', bkg_stream=True):
        """world_size > 1: data parallel, one process per GPU, every rank its shard of the rays (the reference wraps the model in
        DistributedDataParallel, common/trainer/basic_trainer.py:192-198).  The flat gradient is SUMMED over the ranks between the backward
        and the optimiser - grad_sync 'flat': ONE all-reduce of the flattened optimiser's gradient buffer, then FusedAdam.step() on every
        rank; 'sharded': reduce-scatter, Adam on this rank's 1/N of the buffer, all-gather of the parameters (distributed.ShardedGradSync) -,
        FusedAdam.grad_scale must be 1 / world_size (DDP's average), and a refreshed occupancy (the foreground Volume's fields, the
        background cascade's grid and bits) is rank 0's on every rank (DDP's broadcast_buffers).  The scatters' chunk owners cannot apply
        the optimiser then (the summed gradient only exists after the exchange): scatter, exchange, one optimiser pass."""
        reason = self.why_not(model, loss_factory, optimizer)
        if reason is not None:
            raise RuntimeError('FusedNeusNgpStep: ' + reason)
        if grad_sync not in ('flat', 'sharded'):
            raise RuntimeError('FusedNeusNgpStep: grad_sync must be flat or sharded')
        self.model, self.fg, self.bkg, self.opt, self.ema = model, model.fg_model, model.bkg_model, optimizer, ema
        self.loss_factory, self.total_epoch, self.prefetch = loss_factory, total_epoch, bool(prefetch)
        il = next(i for i, f in enumerate(loss_factory.funcs) if isinstance(f, ImgLoss))
        el = 1 - il
        self.img_loss, self.img_w, self.img_name = loss_factory.funcs[il], float(loss_factory.weights[il]), loss_factory.names[il]
        self.eik_w, self.eik_name = float(loss_factory.weights[el]), loss_factory.names[el]
        self.steps = 0
        self._ws = {}
        self._ahead = []                 # batches marched ahead: (rays key, occupancy key, foreground handle, background handle), oldest first
        self.apply_optimizer = True      # (False: the gradients stay in the flat buffer - tests compare them with autograd's)
        from .. import distributed as D
        self.world = max(1, int(world_size))
        # (ARCN_DIST_FORCE=1 with an initialised process group: a ONE-rank communicator runs the multi-rank form, collectives included)
        self.dist_step = self.world > 1 or D._active()
        self.grad_sync, self.sync_occupancy = grad_sync, bool(sync_occupancy)
        self._sync = None
        self._occ_seen = None
        if self.dist_step and grad_sync == 'sharded':
            self._sync = D.ShardedGradSync(optimizer._flat[0]['params'].numel(), self.world)
            optimizer.shard_sync = self._sync      # (state_dict() then insists on gather_sharded_state() first)
        # the gathered table rows of the forward are kept for the two second-order gathers (keep_corners=False: three gathers from the table;
        # measured in round 5, DESIGN.md: -46 us per step with the rows kept)
        self.keep_corners = bool(keep_corners)
        # both two-layer geometry nets as one forward and one backward kernel each on the gather's level-major features (arcn_geo2_fwd / _bwd);
        # fused_geo=False: the chain of dense products + glue passes they replace (round 5's form, the A/B reference)
        self.fused_geo = bool(fused_geo) and self.keep_corners
        # where in the step the coming batches' samplers are queued on the sampling stream (they share the chip with whatever follows):
        # 'start' (before the foreground forward) | 'blend' (after both forwards: round 5's place) | 'fg_bwd' (after the background's backward) |
        # 'opt' (behind the last scatter, beside the optimiser pass and the NEXT step's forward).  Measured (profiles/r6_ab_cfg4_march_at.txt):
        # 1.36 / 1.34 / 1.38 / 1.26 ms - the cascade marcher (275 us of long serial waves) costs the MFMA-bound backward kernels it runs beside
        # more than it costs the gathers
        self.bkg_stream = bool(bkg_stream)      # (the background's forward and backward chains beside the foreground's, on a stream of their own)
        if march_at not in ('start', 'blend', 'fg_bwd', 'opt'):
            raise RuntimeError('FusedNeusNgpStep: march_at must be start | blend | fg_bwd | opt')
        self.march_at = march_at
        # the table scatters' chunk owners apply Adam to the levels they own alone (arcn_hashgrid_bwd_lm_adam / _first_second_adam): those levels'
        # gradients never go to HBM and the optimiser pass shrinks to the rest of the flat buffer.  fuse_adam=False (and every multi-rank
        # step): scatter, then one pass
        self.fuse_adam = (bool(fuse_adam) and not self.dist_step and len(optimizer._flat) == 1 and len(optimizer.param_groups) == 1
                          and (optimizer.ema_decay is None or optimizer.ema_in_param))

    def _default_normal(self):
        """the unit default normal of rays without samples (render_cfgs, neus_model.py) as three floats, computed once"""
        d = self._ws.get('dflt_nrm')
        if d is None:
            nv = torch.tensor(self.fg.render_cfgs['normal'], dtype=torch.float32)
            d = self._ws['dflt_nrm'] = (nv / (nv.norm() + 1e-8)).tolist()
        return d

    def _zeros(self, n, device):
        """n read-only zero floats (kept: no fill kernel per step)"""
        z = self._ws.get('zeros')
        if z is None or z.numel() < n or z.device != device:
            z = self._ws['zeros'] = torch.zeros(max(int(n * 1.25), 1 << 16), dtype=torch.float32, device=device)
        return z[:n]

    # ---- the samplers of coming batches, on a second stream ---------------------------------------------------------------------------------
    @staticmethod
    def _rays_key(rays_o, rays_d):
        return (rays_o.data_ptr(), rays_d.data_ptr(), rays_o.shape[0], rays_o._version, rays_d._version)

    def _occupancy_key(self):
        """what the two samplers read besides the rays: (pointer, version) of the foreground volume's bitfield and of the cascade's"""
        bf = self.fg.obj_bound.volume.get_voxel_bitfield()
        db = self.bkg.density_bitfield
        # (the cascade's bits are rewritten by a kernel, behind torch's version counter: MortonDensityGrid counts its refreshes in ema_step)
        return (bf.data_ptr(), bf._version, db.data_ptr(), db._version, int(getattr(self.bkg, 'ema_step', 0)))

    def _sync_occupancy(self):
        """data parallel: when either occupancy structure has changed since the last step (model.optimize refreshed it - at the same epochs
        on every rank), rank 0's state replaces every rank's, in place: the foreground Volume's opacity field + bitfield, the background
        cascade's density grid + packed bits.  A collective decided by the model's state alone, never by rank-local buffers."""
        if not self.dist_step or not self.sync_occupancy:
            return
        key = self._occupancy_key()
        if key == self._occ_seen:
            return
        from .. import distributed as D
        vol = self.fg.obj_bound.volume
        D.broadcast_occupancy(vol.get_voxel_opafield(flatten=True), vol.get_voxel_bitfield(flatten=True))
        D.broadcast_occupancy(self.bkg.density_grid, self.bkg.density_bitfield)
        self._occ_seen = self._occupancy_key()

    def _march_ahead(self, feeds):
        """Queue both samplers (+ their scans, the totals on their way to pinned memory) for the batches of `feeds` that are not marched yet, in
        order, on the sampling stream - behind everything the main stream has been given so far.  One batch ahead the step's start finds totals
        that arrived half a step ago; TWO batches ahead (a list of two) the host can issue a whole step while the device is still in the one
        before - a host that stalls for a millisecond (shared machines) no longer stalls the device.  Each sampler advances its generator once
        per batch in batch order, as without any of this: the same samples."""
        have = {e[0] for e in self._ahead}
        todo = []
        for f in feeds:
            o = f['rays_o'].reshape(-1, 3)
            d = f['rays_d'].reshape(-1, 3)
            if not (o.is_cuda and o.is_contiguous() and d.is_contiguous() and o.dtype == torch.float32 and d.dtype == torch.float32):
                break
            k = self._rays_key(o, d)
            if k in have:
                continue
            todo.append((k, o, d))
            have.add(k)
        if not todo:
            return
        st = getattr(self.model, '_sample_stream', None)
        if st is None or st.device != todo[0][1].device:
            st = self.model._sample_stream = torch.cuda.Stream(device=todo[0][1].device)
        st.wait_stream(torch.cuda.current_stream())
        occ = self._occupancy_key()
        with torch.cuda.stream(st):
            for k, o, d in todo:
                hb = self.bkg._sample_begin(o, d)       # (the handles hold o and d: their addresses cannot be handed to another batch while
                hf = self.fg._sample_begin(o, d)        # an entry waits here, and the allocator knows the sampling stream reads them)
                self._ahead.append((k, occ, hf, hb))

    @staticmethod
    def _level_ranges(emb, mask, first):
        """float ranges [lo, hi) in the flat buffers of the table levels in `mask` (first = the table's first float), neighbours merged"""
        F_, offs = int(emb.desc.n_feat), emb.desc.offsets
        out = []
        for l in range(int(emb.desc.n_levels)):
            if (mask >> l) & 1:
                lo, hi = first + int(offs[l]) * F_, first + int(offs[l + 1]) * F_
                if out and out[-1][1] == lo:
                    out[-1] = (out[-1][0], hi)
                else:
                    out.append((lo, hi))
        return out

    def _geo_scratch(self, n, device, key='geo2_fg'):
        """the weight-gradient partials of arcn_geo2_bwd, kept - one buffer per net (the two backward chains may run on two streams)"""
        need = int(F.N.lib().arcn_geo2_bwd_scratch_floats(int(n)))
        w = self._ws.get(key)
        if w is None or w.numel() < need or w.device != device:
            w = self._ws[key] = torch.empty(max(1, int(need * 1.25)), dtype=torch.float32, device=device)
        return w

    def _bkg_side(self, cur):
        """the context the background model's chains are issued in: its own stream behind everything `cur` has been given so far (bkg_stream), or
        nothing.  The two models meet at the blend (forward) and at the optimiser (backward) only: different tables, nets and gradient
        segments in between - and most of their kernels are too small to fill the chip alone (512 - 768 workgroups at two per CU).
        Memory: every tensor of the step stays referenced until __call__ returns, and by then `cur` has been told to wait for the background
        stream (_bkg_join in front of the optimiser); a block freed there is only handed out again to work that is queued behind that wait (the
        step's own stream) or behind the next step's `st.wait_stream(cur)` (the background stream's pool) - no record_stream needed."""
        import contextlib
        if not self.bkg_stream:
            return contextlib.nullcontext()
        st = self._ws.get('bkg_stream')
        if st is None or st.device != cur.device:
            st = self._ws['bkg_stream'] = torch.cuda.Stream(device=cur.device)
        st.wait_stream(cur)
        return torch.cuda.stream(st)

    def _bkg_join(self, cur):
        if self.bkg_stream and self._ws.get('bkg_stream') is not None:
            cur.wait_stream(self._ws['bkg_stream'])

    def _scatter_ws(self, key, desc, n, device):
        """scratch of a binned table scatter, kept while it is large enough (the library knows the size for n samples)"""
        w = self._ws.get(key)
        need = int(F.N.lib().arcn_hashgrid_bwd_workspace_floats(F.C.addressof(desc), int(n)))
        if w is None or w.numel() < need or w.device != device:
            w = self._ws[key] = torch.empty(max(1, int(need * 1.25)), dtype=torch.float32, device=device)
        return w

    # ---- the iteration ----------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, feed_in, epoch=0, next_feed_in=None):
        model, fg, bkg = self.model, self.fg, self.bkg
        rays_o = feed_in['rays_o'].reshape(-1, 3).contiguous().float()
        rays_d = feed_in['rays_d'].reshape(-1, 3).contiguous().float()
        img = feed_in['img'].reshape(-1, 3).contiguous().float()
        bkg_color = feed_in['bkg_color'].reshape(-1, 3).contiguous().float() if feed_in.get('bkg_color') is not None else None
        b, n = feed_in['rays_o'].shape[:2]
        R, dev = rays_o.shape[0], rays_o.device
        if not self.opt.zero_grad_on_step:
            self.opt.zero_grad()
        if self.world > 1 and abs(float(self.opt.grad_scale) - 1.0 / self.world) > 1e-12:
            raise RuntimeError('FusedNeusNgpStep(world_size={}): FusedAdam.grad_scale must be 1 / world_size (the gradients are SUMMED over the '
                               'ranks), it is {}'.format(self.world, self.opt.grad_scale))
        self._sync_occupancy()
        # this step's optimiser numbers, for the scatters whose owners apply it: asked for right before the first of them (begin_step advances the
        # Adam / EMA counters - an exception in the forward or the sampler must not leave them a step ahead of the parameters)
        fuse = bool(self.apply_optimizer and self.fuse_adam)
        hyper = None
        done = []                                                                                 # float ranges the scatters' owners have updated
        cur = torch.cuda.current_stream()
        # ---- samples of both models: marched one or two steps ago on the sampling stream (self._ahead, oldest first), or now
        h_fg = h_bkg = None
        key, occ = self._rays_key(rays_o, rays_d), self._occupancy_key()
        while self._ahead:
            k_, occ_, hf_, hb_ = self._ahead.pop(0)
            if k_ == key and occ_ == occ:
                for h in (hf_, hb_):
                    cur.wait_event(h['event'])
                    for t_ in h.values():
                        for u in (t_ if isinstance(t_, tuple) else (t_,)):
                            if isinstance(u, torch.Tensor) and u.is_cuda:
                                u.record_stream(cur)
                h_fg, h_bkg = hf_, hb_
                break
            # (another batch, or an occupancy structure changed since: what was marched ahead is dropped, like FullModel's one-slot form does)
        if h_fg is None:
            self._ahead = []
            h_bkg = bkg._sample_begin(rays_o, rays_d)      # (the order FullModel runs the two samplers in: each has its own generator anyway)
            h_fg = fg._sample_begin(rays_o, rays_d)
        def ahead():
            if self.prefetch and next_feed_in is not None:
                self._march_ahead(next_feed_in if isinstance(next_feed_in, (list, tuple)) else [next_feed_in])
        pk = F.neus_pack_end(h_fg, float(fg.get_ray_cfgs('n_sample')))
        t_b, ray_b, off_b, pd_b, total_b = F.pack_dense_samples_end(h_bkg)
        S = pk['total']
        if self.march_at == 'start':
            ahead()
        # ---- foreground forward
        emb, l0, l1 = _sdf_net_form(fg.geo_net)
        rad = fg.radiance_net
        table = emb.embeddings
        beta = float(l0.activation.beta)
        n_out = l1.weight.shape[0]
        emb_b, b0, b1 = _density_net_form(bkg.geo_net)
        # the weights both geometry nets' kernels take this step - the padded last layers, the Jacobian row's folded first layer, beta W2[0] -
        # and the NeuS scale exp(inv_s * speed): one launch (arcn_neus_step_prep)
        prep = F.neus_step_prep(l0.weight, l1.weight, beta, fg.inv_s.detach().reshape(1), float(fg.speed_factor), b1.weight)
        w1, w2, wb1 = l0.weight, prep['w2p'], prep['wb1p']
        s_dev = prep['scale']
        cos_anneal = fg.get_cos_anneal(epoch)
        dflt_nrm = self._default_normal()
        dflt_rgb = fg.render_cfgs['bkg_color']
        # ---- background forward: issued FIRST so that, on its own stream (bkg_stream), it runs beside the foreground's forward
        rb = bkg.radiance_net
        tb = emb_b.embeddings
        nb_out = b1.weight.shape[0]
        with self._bkg_side(cur):      # (bkg_stream: beside the foreground's forward, on the background's own stream)
            if total_b > 0:
                xyz_b, dirs_b = F.packed_points(rays_o, rays_d, t_b, ray_b)
                if self.fused_geo:
                    enc_b, _ = F.hashgrid_fwd_lm(xyz_b, tb, emb_b.desc)
                    out_b, sigma_b, _ = F.geo2_fwd(enc_b, total_b, b0.weight, b1.weight, False)
                else:
                    enc_b = F.hashgrid_fwd(xyz_b, tb, emb_b.desc)
                    hid_b = F.gemm_nt(enc_b, b0.weight, None, act='relu')
                    out_b = F.gemm_nt(hid_b, wb1, None)
                    sigma_b = F.act_col_scale(out_b, 'truncexp', 1.0)       # (the density from column 0 of the padded output)
                rin_b = F.radiance_inputs('fv', None, dirs_b, None, out_b[:, 1:nb_out], rb.embed_fn_view.n_freqs)
                rb_w = _flat_view([layer.weight for layer in rb.layers])
                rb_g = _flat_view([layer.weight.grad for layer in rb.layers]) if rb_w is not None else None
                if rb_w is None:
                    rb_w = torch.cat([layer.weight.reshape(-1) for layer in rb.layers])
                rgb_sb, rb_acts = F.mlp_fwd(rin_b, rb_w, None, rb._fused_desc, save_acts=True)
                comp = F.composite_packed_fwd(sigma_b, rgb_sb, t_b, off_b, p_dense_dev=pd_b, add_inf_z=bool(bkg.add_inf_z),
                                              white_bkg=bool(bkg.get_ray_cfgs('white_bkg')))
                rgb_b, depth_b = comp['rgb'], comp['depth']
            else:
                rgb_b = rays_o.new_ones((R, 3)) if bkg.get_ray_cfgs('white_bkg') else rays_o.new_zeros((R, 3))
                depth_b = rays_o.new_zeros((R,))
        # ---- foreground forward (continued)
        if S > 0:
            fg.adjust_dynamicbs_factor(n_valid=pk['offsets'][R])
            pts, dirs = F.packed_points(rays_o, rays_d, pk['t_mid'], pk['ray_id'])
            # (the gathered rows are kept: the normals and the gradient of the Jacobian row stream them instead of gathering again;
            # ARCN_NEUS_CORNERS=0: three gathers from the table)
            if self.fused_geo:
                enc, corners = F.hashgrid_fwd_lm(pts, table, emb.desc, want_corners=True)     # (level-major features: what the net's kernels read)
                out, sdf, jac = F.geo2_fwd(enc, S, w1, l1.weight, True, beta)
            elif self.keep_corners:
                enc, corners = F.hashgrid_fwd_corners(pts, table, emb.desc)
            else:
                enc, corners = F.hashgrid_fwd(pts, table, emb.desc), None
            if not self.fused_geo:
                hid = F.gemm_nt(enc, w1, None, act='softplus', beta=beta)
                out = F.gemm_nt(hid, w2, None)
                sg = F.softplus_grad(hid, None, beta, from_y=True)
                jac = F.gemm_nn(sg, prep['w1j'])
            if corners is not None:
                normal = F.hashgrid_dxyz_corners(pts, corners, jac, emb.desc)
            else:
                _, normal = F.hashgrid_bwd(pts, table, jac, emb.desc, want_dtable=False, want_dxyz=True)
            if not self.fused_geo:
                sdf = F.act_col_scale(out, None, 1.0)       # (column 0 of the padded output)
            n_sh = rad.embed_fn_view.n_freqs ** 2
            rad_in = F.radiance_inputs('pvnf', pts, dirs, normal, out[:, 1:n_out], rad.embed_fn_view.n_freqs)       # [p | SH(normalize(v)) | n | f], one pass
            rad_w = _flat_view([layer.weight for layer in rad.layers])
            rad_g = _flat_view([layer.weight.grad for layer in rad.layers]) if rad_w is not None else None
            if rad_w is None:
                rad_w = torch.cat([layer.weight.reshape(-1) for layer in rad.layers])
            rgb_s, rad_acts = F.mlp_fwd(rad_in, rad_w, None, rad._fused_desc, save_acts=True)
        else:
            sdf = rays_o.new_zeros((1,))
            rgb_s = normal = rays_o.new_zeros((1, 3))
        rgb_f, depth_f, mask_f, nrm_f, t_last = F.neus_render_fwd(sdf, rgb_s, normal, pk, rays_d, s_dev, cos_anneal, bkg_color,
                                                                 float(fg.render_cfgs['depth_far']), dflt_rgb, dflt_nrm)
        # ---- (the background's forward was issued above, before the foreground's)
        self._bkg_join(cur)
        # the samplers of the next batch, beside everything that follows
        if self.march_at == 'blend':
            ahead()
        # ---- blend + losses
        il = self.img_loss
        bl = F.neus_blend_loss(rgb_f, depth_f, t_last, rgb_b, depth_b, img, float(il.loss.delta) if isinstance(il.loss, HuberLoss) else None, self.img_w)
        rgb, depth, d_rgb, d_tlast, d_rgb_b = bl['rgb'], bl['depth'], bl['d_rgb'], bl['d_tlast'], bl['d_rgb_b']
        losses = bl['loss']          # [image loss, 0 = the Eikonal pass's accumulator]
        # ---- background backward
        with self._bkg_side(cur):      # (bkg_stream: the background's whole backward, its table scatter + optimiser included, beside the foreground's)
            if total_b > 0:
                d_sig_b, d_rad_b = F.composite_packed_bwd(sigma_b, rgb_sb, t_b, off_b, d_rgb_b, None, None, p_dense_dev=pd_b,
                                                          add_inf_z=bool(bkg.add_inf_z), white_bkg=bool(bkg.get_ray_cfgs('white_bkg')))
                dx_b, dw_b, _ = F.mlp_bwd(rin_b, rb_w, None, rb._fused_desc, rgb_sb, rb_acts, d_rad_b, want_dx=True, dweights=rb_g)     # (added into rb_g)
                if rb_g is None:
                    k = 0
                    for layer in rb.layers:
                        m_ = layer.weight.numel()
                        layer.weight.grad.add_(dw_b[k:k + m_].view_as(layer.weight))
                        k += m_
                # [d density through TruncExp | d features | 0] of the padded output, one pass
                lm_b = 0
                if self.fused_geo:
                    # [d density through TruncExp | d features] enters the net's backward in pieces; d_enc_b level-major, as the scatter reads it
                    d_enc_b = F.geo2_bwd(enc_b, total_b, b0.weight, b1.weight, False, 1.0, d_sig_b, dx_b[:, :nb_out - 1], b0.weight.grad, b1.weight.grad,
                                         out=out_b, dx_level_major=True, scratch=self._geo_scratch(total_b, dev, 'geo2_bkg'))
                    lm_b = total_b
                else:
                    g_out_b = F.geo_out_grad(d_sig_b, dx_b[:, :nb_out - 1], wb1.shape[0], out=out_b, act='truncexp', y_col0=sigma_b)
                    F.gemm_tn(g_out_b, hid_b, out=b1.weight.grad, accumulate=True, head=nb_out)
                    d_hid_b = F.gemm_nn(g_out_b, wb1)
                    F.gemm_tn(d_hid_b, enc_b, mask=hid_b, out=b0.weight.grad, accumulate=True)
                    d_enc_b = F.gemm_nn(d_hid_b, b0.weight, mask=hid_b)
                ws_b = self._scatter_ws('bkg', emb_b.desc, total_b, dev)
                if fuse:
                    hyper = hyper or self.opt.begin_step()
                    m_, v_, o_ = self.opt.table_views(tb)
                    done += self._level_ranges(emb_b, F.hashgrid_bwd_adam(xyz_b, d_enc_b, emb_b.desc, tb.grad, tb, m_, v_, hyper, ws_b, level_stride=lm_b), o_)
                else:
                    F.hashgrid_bwd(xyz_b, tb, d_enc_b, emb_b.desc, dtable=tb.grad, workspace=ws_b, level_stride=lm_b)
        if self.march_at == 'fg_bwd':
            ahead()
        # ---- foreground backward
        if S > 0:
            zr = self._zeros(4 * R, dev)       # (read-only zero upstream gradients)
            d_sdf, d_rad, d_normal, d_s_ray = F.neus_render_bwd(sdf, rgb_s, normal, pk, rays_d, s_dev, cos_anneal, bkg_color, d_rgb,
                                                                zr[:R], zr[:R], zr[R:4 * R].view(R, 3), d_tlast)
            dx_r, dw_r, _ = F.mlp_bwd(rad_in, rad_w, None, rad._fused_desc, rgb_s, rad_acts, d_rad, want_dx=True, dweights=rad_g)
            if rad_g is None:     # (the radiance weights are separate nn.Linear tensors: not back to back, the kernel's flat gradient is split back)
                k = 0
                for layer in rad.layers:
                    m_ = layer.weight.numel()
                    layer.weight.grad.add_(dw_r[k:k + m_].view_as(layer.weight))
                    k += m_
            # Eikonal value + gradient and the radiance net's gradient of its normal inputs, both added into d_normal in one pass
            F.eikonal_packed(normal, pk, R, self.eik_w, d_normal=d_normal, loss=losses[1:2], add_src=dx_r[:, 3 + n_sh:6 + n_sh], loss_is_clear=True)
            # the normals' gradient: to the Jacobian row they were built from (its table part joins the first-order scatter below)
            if corners is not None:
                d_jac = F.hashgrid_ddout_corners(pts, d_normal, corners, emb.desc)
            else:
                d_jac, _, _ = F.hashgrid_bwd_bwd(pts, d_normal, table, jac, emb.desc, want_ddout=True, want_dtable=False, want_d2xyz=False)
            # the sdf net, first output's Jacobian included (ops.autograd.SdfMlpJacFn.backward)
            if self.fused_geo:
                # [d sdf | d features] and d_jac in, d_enc and both layers' gradients (ordinary + Jacobian path) out: one kernel + its reduction
                d_enc = F.geo2_bwd(enc, S, w1, l1.weight, True, beta, d_sdf, dx_r[:, 6 + n_sh:6 + n_sh + n_out - 1], l0.weight.grad, l1.weight.grad,
                                   d_jac=d_jac, scratch=self._geo_scratch(S, dev))
            else:
                g_out = F.geo_out_grad(d_sdf, dx_r[:, 6 + n_sh:6 + n_sh + n_out - 1], w2.shape[0])       # [d sdf | d features | 0]
                dz = F.gemm_nn(g_out, w2)
                F.gemm_tn(g_out, hid, out=l1.weight.grad, accumulate=True, head=n_out)
                u = F.gemm_nt(d_jac, w1, None)
                # dz through the softplus + the Jacobian path's curvature term; s * W2[0] (the operand of that path's first-layer gradient) in u's
                # place; the path's gradient of W2[0] - the column sums of s u - straight into the last layer's gradient row
                dz, sw = F.sdf_jac_dz2(dz, u, sg, prep['bw20'], w2[0], l1.weight.grad[0])
                F.gemm_tn(sw, d_jac, out=l0.weight.grad, accumulate=True)
                d_enc = F.gemm_nn(dz, w1)
                F.gemm_tn(dz, enc, out=l0.weight.grad, accumulate=True)
            # the table: through the encoding (d_enc) and through its input gradient (d_normal on J^T jac), ONE accumulation pass for both
            ws_f = self._scatter_ws('fg', emb.desc, 3 * S, dev)
            if fuse:
                hyper = hyper or self.opt.begin_step()
                m_, v_, o_ = self.opt.table_views(table)
                done += self._level_ranges(emb, F.hashgrid_bwd_first_second_adam(pts, d_enc, d_normal, jac, emb.desc, table.grad, table, m_, v_, hyper, ws_f), o_)
            else:
                F.hashgrid_bwd_first_second(pts, d_enc, d_normal, jac, emb.desc, table.grad, ws_f)
            # scale = exp(inv_s * speed)
            F.sum_scale_add(d_s_ray, fg.inv_s.grad, float(fg.speed_factor), s_dev)
        self._bkg_join(cur)
        if self.march_at == 'opt':
            ahead()
        # ---- optimiser
        if self.apply_optimizer:
            if fuse:
                hyper = hyper or self.opt.begin_step()
                self.opt.finish_step(hyper, done)      # Adam on what the scatters' owners did not update: the coarse levels, the nets, inv_s
            elif self._sync is not None:
                # reduce-scatter of the flat gradient, Adam on this rank's shard (+ the replicated tail), all-gather of the parameters
                fb = self.opt._flat[0]
                self._sync.launch(fb['grads'])
                self._sync.wait()
                hyper = self.opt.begin_step()
                n_all = fb['params'].numel()
                keep, at = [], 0
                for lo, hi in sorted(self._sync.segments):
                    if lo > at:
                        keep.append((at, lo))
                    at = hi
                if at < n_all:
                    keep.append((at, n_all))
                self.opt.finish_step(hyper, keep)      # (`keep` = everything that is NOT this rank's: left to its owner)
                self._sync.clear_foreign(fb['grads'])
                self._sync.gather(fb['params'])
            else:
                if self.dist_step:
                    from .. import distributed as D
                    D.allreduce_grads(self.opt.flat_grads(), self.world)       # ONE collective on the flat buffer; 1 / world is the optimiser's grad_scale
                self.opt.step()
            if self.ema is not None:
                self.ema.ema_step()
        self.steps += 1
        total = losses.sum()
        out = {'rgb': rgb.view(b, n, 3), 'depth': depth.view(b, n), 'mask': mask_f.view(b, n), 'normal': nrm_f.view(b, n, 3),
               'params': {'scale': s_dev.reshape(())}}
        return out, {'sum': total, 'names': [self.img_name, self.eik_name], self.img_name: losses[0], self.eik_name: losses[1]}
