"""ORACLE — TEST INFRASTRUCTURE ONLY: the reference's instant-ngp TRAINING LOOP restated on the CPU oracle.

Never imported by the product package.  One iteration, in the reference trainer's order (arcnerf/trainer/arcnerf_trainer.py:494-548 train_epoch,
:319-333 step_optimize):

    optimize(epoch)          VolumeBound.optimize (volume_bound.py:160-212): warm-up = every cell, afterwards n/4 cells of a permutation +
                             the first n/4 occupied cells; jitter; opacity = sigma * dt (base_3d_model.py:368-389); unique + K4 max;
                             update_opafield_by_voxel_idx / update_bitfield_by_opafield (volume.py:983-1017)
    n_rays(epoch)            Pipeline.fetch_step_update_dynamic_bs (trainer/pipeline.py:222-241) on FgModel.get_dynamicbs_factor (fg_model.py:105-130)
    step(rays, colours)      oracle_step (ngp_reference.py: K2, K3, hash grid, nets, compositing, Huber loss, backward) ->
                             torch.optim.Adam's single-tensor update (torch/optim/adam.py: weight decay added to the gradient, lerp_ first
                             moment, addcmul_ second, bias corrections as Python doubles, eps outside the root) -> EMA.ema_step (ema.py:29-43)

Pinned to a run of the reference's own loop by tests/test_oracle_trajectory_golden.py (golden G26, reference semantics `nb`: geometry
output = [sigma | 15 features]); the GPU tests then use it with the config's FUSED semantics (features = the whole 16-wide output), which no
CPU run of the reference can produce (tiny-cuda-nn), to check NgpPipeline.train_step over a trajectory.
"""
import numpy as np

from .ngp_reference import oracle_step

F32 = np.float32


class OracleNgpTrainer:
    def __init__(self, orc, fld, cfg, flat_params, max_allowance, n_rays, update_epoch, max_batch_size, start_epoch=0):
        """fld: an arcnerf_amd.pipeline.NgpField built on the CPU or GPU (metadata only: level table, bounds, flat segment layout);
        flat_params: float32 numpy copy of its flat parameter buffer (owned by this object from here on)"""
        self.orc, self.fld, self.cfg = orc, fld, cfg
        self.p = np.ascontiguousarray(flat_params, F32).copy()
        self.m = np.zeros_like(self.p)
        self.v = np.zeros_like(self.p)
        self.adam_step = 0
        self.ema_n_step = int(start_epoch)              # ArcNerfTrainer.__init__: ema.set_n_step(progress.start_epoch)
        ng = cfg.n_grid
        self.opafield = np.zeros(ng ** 3, F32)
        self.bitfield = np.ones(ng ** 3, bool)
        self.rng = orc.Pcg32(9121)
        self.max_allowance = max_allowance
        self.measured_batch_size, self.measured_count = 0.0, 0
        self.n_rays, self.update_epoch, self.max_batch_size = n_rays, update_epoch, max_batch_size
        self.last = {}

    # ---- views of the flat buffer in the layout oracle_step wants --------------------------------------------------------------
    def views(self, flat=None):
        fld, flat = self.fld, (self.p if flat is None else flat)
        out = {}
        off, n = fld._seg['table']
        out['table'] = flat[off:off + n].reshape(-1, self.cfg.n_feat_per_entry)
        for name, dims in (('geo', fld.geo_dims), ('rad', fld.rad_dims)):
            ow, _ = fld._seg[name + '_w']
            ob, nb = fld._seg.get(name + '_b', (0, 0))
            layers = []
            for i in range(len(dims) - 1):
                k = dims[i] * dims[i + 1]
                W = flat[ow:ow + k].reshape(dims[i + 1], dims[i])
                ow += k
                b = None
                if nb:
                    b = flat[ob:ob + dims[i + 1]]
                    ob += dims[i + 1]
                layers.append((W, b))
            out[name] = layers
        return out

    # ---- VolumeBound.optimize ---------------------------------------------------------------------------------------------------
    def optimize(self, epoch, perm, uni):
        """perm (n_cells,) int64, uni (n_cells, 3) float32: the run's torch.randperm / torch.rand_like draws.  Returns True if refreshed."""
        cfg, orc = self.cfg, self.orc
        if epoch <= 0 or cfg.epoch_optim is None or epoch % cfg.epoch_optim != 0:
            return False
        ng = cfg.n_grid
        n_cells = ng ** 3
        if cfg.epoch_optim_warmup is not None and epoch < cfg.epoch_optim_warmup:
            cell = np.arange(n_cells, dtype=np.int64)
        else:
            n_s = n_cells // 4
            occ = np.nonzero(self.bitfield)[0][:n_s]
            cell = np.concatenate([perm[:n_s], occ]).astype(np.int64)
        ix, iy, iz = cell // (ng * ng), (cell // ng) % ng, cell % ng
        idx3 = np.stack([ix, iy, iz], -1).astype(F32)
        vs = F32(cfg.side) / F32(ng)                                            # get_voxel_size: (max - min) / n_grid in float32
        mn = np.array(self.fld.min_xyz, F32)
        pts = idx3 * vs + F32(0.5) * vs + mn                                    # get_voxel_pts_by_voxel_idx (volume.py:437-452)
        noise = (uni[:cell.shape[0]] - F32(0.5)) * vs
        pts = (pts + noise).astype(F32)
        P = self.views()
        res, offs = np.array(self.fld.resolutions, np.int32), np.array(self.fld.offsets, np.int64)
        h = orc.hashgrid_fwd(pts, P['table'], res, offs, mn, np.array(self.fld.max_xyz, F32))
        for i, (W, b) in enumerate(P['geo']):
            h = orc.linear_fwd(h, W, b, 'relu' if i < len(P['geo']) - 1 else None)
        sigma = orc.act_fwd(np.ascontiguousarray(h[:, 0]), cfg.sigma_act)
        opacity = (sigma * F32(cfg.dt)).astype(F32)                             # get_est_opacity: density * dt
        uniq, inv = np.unique(cell, return_inverse=True)
        uni_opa = orc.tensor_reduce_max(opacity, inv.astype(np.int64), uniq.shape[0])      # K4
        orc.update_opafield(self.opafield, uniq, uni_opa, ema=cfg.ema_optim_decay)
        bf, thres = orc.update_bitfield_by_opafield(self.opafield, cfg.opa_thres)
        self.bitfield = bf.reshape(-1)
        self.last['thres'], self.last['n_refresh_pts'] = thres, int(cell.shape[0])
        return True

    # ---- dynamic batch size ---------------------------------------------------------------------------------------------------------
    def update_n_rays(self, epoch):
        if self.update_epoch > 0 and epoch % self.update_epoch == 0 and epoch > 500:
            factor = self.measured_batch_size / self.measured_count if self.measured_count > 0 else 1
            self.measured_batch_size, self.measured_count = 0.0, 0
            val = self.n_rays * factor
            self.n_rays = min(int((val + 128 - 1) // 128 * 128), self.max_batch_size)
            self.last['dyn_factor'] = factor
        return self.n_rays

    # ---- step_optimize ----------------------------------------------------------------------------------------------------------------
    def step(self, rays_o, rays_d, bkg, img):
        cfg = self.cfg
        res = oracle_step(self.orc, self.fld, cfg, self.views(), rays_o, rays_d, bkg, self.bitfield.reshape(cfg.n_grid, cfg.n_grid, cfg.n_grid),
                          self.rng.state, self.rng.inc, huber_target=img)
        self.rng.advance()
        if res['n_samples'] > 0 and self.max_allowance > 0:                     # adjust_dynamicbs_factor (only reached with valid rays)
            self.measured_count += 1
            self.measured_batch_size += float(self.max_allowance) / (float(res['n_samples']) + 1)
        self.adam(res['grads'])
        self.ema()
        return res

    def adam(self, grad):
        cfg = self.cfg
        self.adam_step += 1
        b1, b2 = cfg.betas
        g = grad.astype(F32)
        if cfg.weight_decay != 0:
            g = g + F32(cfg.weight_decay) * self.p
        self.m += (g - self.m) * F32(1 - b1)                                   # exp_avg.lerp_(grad, 1 - beta1)
        self.v *= F32(b2)
        self.v += F32(1 - b2) * g * g                                           # addcmul_(grad, grad, value=1 - beta2)
        bc1 = 1 - b1 ** self.adam_step
        bc2_sqrt = (1 - b2 ** self.adam_step) ** 0.5
        denom = np.sqrt(self.v) / F32(bc2_sqrt) + F32(cfg.eps)
        self.p -= F32(cfg.lr / bc1) * (self.m / denom)                          # addcdiv_(exp_avg, denom, value=-step_size)

    def ema(self):
        d = self.cfg.ema_decay
        if d is None:
            return
        self.ema_n_step += 1
        deb_old = 1 - d ** (self.ema_n_step - 1)
        deb_new = 1.0 / (1 - d ** self.ema_n_step)
        if not hasattr(self, 'old_avg'):
            raise RuntimeError('call start_ema() once the parameters are in place (EMA.__init__ clones them)')
        new = (F32(1 - d) * self.p + F32(d) * self.old_avg * F32(deb_old)) * F32(deb_new)
        self.p[:] = new
        self.old_avg = new

    def start_ema(self):
        self.old_avg = self.p.copy()
        return self
