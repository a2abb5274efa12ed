"""RadianceNet of the linear family in any mode over {p, v, n, f} (encoder_mlp_network.py:62-118, linear_network_module.py:318-335) as ONE
first-order autograd node over all the points of a pass: the input blocks - encoded position, encoded unit view direction, normal,
geometry feature - are written into their column ranges of one (chunk, width) buffer (the encoders by their kernels, normal and feature
by one strided copy each: no torch.cat, no zero-pad tensor), the ReLU DenseLayers run on the products of csrc/gemm.hip with bit masks,
the chunk loop (model.chunk_pts) is inside the node so that weight / bias gradients are summed by the products' accumulate flag, and the
gradients of normal and feature - what the sdf node of NeuS differentiates a second time (ops/sdf_chain.py) - leave as two slices of the
first layer's input gradient.  Weight norm stays outside (the node takes the effective weights).  config 3 of BASELINE.json (mode
'pvnf', 4 x 256, weight norm); the NeRF family's 'vf' nets take ops/field_chain.py together with their geometry net.
ARCN_RADIANCE_CHAIN=0 turns the node off.  tests/test_gpu_kernels.py::test_radiance_chain_equals_the_layer_by_layer_module."""

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable

from . import functional as F
from .field_chain import _relu_layer


class RadSpec:
    def __init__(self, rad, chunk):
        self.chunk = int(chunk)
        self.mode = rad.mode
        self.blocks = []          # (kind, first column, width)
        off = 0
        for m in rad.mode:
            w = {'p': rad.embed_fn_pts.get_output_dim() if 'p' in rad.mode else 0, 'v': rad.embed_fn_view.get_output_dim() if 'v' in rad.mode else 0,
                 'n': 3, 'f': max(int(rad.W_feat_in), 0)}[m]
            self.blocks.append((m, off, w))
            off += w
        self.width = off
        self.ld = (off + 3) // 4 * 4
        self.pts_enc = (rad.embed_fn_pts.n_freqs, bool(rad.embed_fn_pts.include_input)) if 'p' in rad.mode else None
        self.view_enc = (rad.embed_fn_view.n_freqs, bool(rad.embed_fn_view.include_input)) if 'v' in rad.mode else None
        self.widths = [layer.out_features for layer in rad.layers]
        self.sigmoid = isinstance(rad.layers[-1].activation, nn.Sigmoid)


def make_rad_spec(rad, chunk):
    from ..models.base_modules.encoding.freq_encoder import FreqEmbedder
    from ..models.base_modules.geo_rad_model.linear_network_module import RadianceNet
    from ..models.base_modules.linear import DenseLayer
    if type(rad) is not RadianceNet or rad._fused_desc is not None:
        return None
    if len(set(rad.mode)) != len(rad.mode) or len(rad.layers) < 2:
        return None
    for m, emb in (('p', rad.embed_fn_pts), ('v', rad.embed_fn_view)):
        if m in rad.mode and (type(emb) is not FreqEmbedder or emb.input_dim != 3 or (emb.n_freqs == 0 and not emb.include_input)):
            return None
    spec = RadSpec(rad, chunk)
    for j, layer in enumerate(rad.layers):
        last = j == len(rad.layers) - 1
        if type(layer) is not DenseLayer:
            return None
        if layer.in_features != (spec.width if j == 0 else rad.layers[j - 1].out_features):
            return None
        if last:
            if layer.out_features != 3 or not isinstance(layer.activation, (nn.Sigmoid, nn.Identity)):
                return None
        elif type(layer.activation) is not nn.ReLU or layer.out_features % 4:
            return None
    return spec


def effective_params(rad):
    params = []
    for layer in rad.layers:
        w = torch._weight_norm(layer.weight_v, layer.weight_g, 0) if hasattr(layer, 'weight_g') else layer.weight
        params += [w, layer.bias]
    return params


class RadianceChainFn(torch.autograd.Function):
    """pre-activation rgb padded to 4 columns (n, 4) = radiance net on [blocks of spec.mode]; x (n, 3) positions, dirs (n, 3) UNIT view
    directions, normal (n, 3), feat (n, W_feat) (any of them None when the mode does not use it); params = weight, bias (or None) per layer"""

    @staticmethod
    def forward(ctx, x, dirs, normal, feat, spec, *params):
        src = next(t for t in (x, dirs, normal, feat) if t is not None)
        n, dev = src.shape[0], src.device
        want = any(ctx.needs_input_grad)
        nl = len(spec.widths)
        pad = torch.nn.functional.pad
        with torch.no_grad():
            ws = [params[2 * j] for j in range(nl)]
            bs = [params[2 * j + 1] for j in range(nl)]
            ws[0] = pad(ws[0], (0, spec.ld - spec.width))
            ws[nl - 1] = pad(ws[nl - 1], (0, 0, 0, 1))
            bs[nl - 1] = None if bs[nl - 1] is None else pad(bs[nl - 1], (0, 1))
            ws = [w.contiguous() for w in ws]
        rgb4 = torch.empty((n, 4), dtype=torch.float32, device=dev)
        chunk = spec.chunk if spec.chunk > 0 else n
        saved = []
        with F.split_weight_scope():
            for lo in range(0, n, chunk):
                hi = min(n, lo + chunk)
                S = hi - lo
                R = torch.empty((S, spec.ld), dtype=torch.float32, device=dev)
                for kind, off, w in spec.blocks:
                    dst = R[:, off:off + w]
                    if kind == 'p':
                        F.freq_fwd_cols(x[lo:hi], spec.pts_enc[0], spec.pts_enc[1], dst)
                    elif kind == 'v':
                        F.freq_fwd_cols(dirs[lo:hi], spec.view_enc[0], spec.view_enc[1], dst)
                    elif kind == 'n':
                        dst.copy_(normal[lo:hi])
                    else:
                        dst.copy_(feat[lo:hi])
                if spec.ld > spec.width:
                    R[:, spec.width:].zero_()
                ins, masks = [], []
                cur = R
                for j in range(nl - 1):
                    out = torch.empty((S, spec.widths[j]), dtype=torch.float32, device=dev)
                    masks.append(_relu_layer(cur, ws[j], bs[j], out, want))
                    ins.append(cur)
                    cur = out
                F.gemm_nt(cur, ws[nl - 1], bs[nl - 1], out=rgb4[lo:hi])
                ins.append(cur)
                if want:
                    saved.append((lo, hi, ins, masks))
            if want:
                probe = saved[0][2][0]
                ctx.ws_nn = [F.split_weights(w, True) if F._use_split(probe, w.shape[0], w.shape[1]) else None for w in ws]
        ctx.spec, ctx.saved, ctx.ws, ctx.bs, ctx.n = spec, saved, ws, bs, n
        ctx.shapes = [None if p is None else p.shape for p in params]
        return rgb4

    @staticmethod
    @once_differentiable
    def backward(ctx, d_rgb4):
        if ctx.saved is None:
            raise RuntimeError('RadianceChainFn: the activations of this pass were released by its first backward (retain_graph is not supported)')
        spec, ws, bs, n = ctx.spec, ctx.ws, ctx.bs, ctx.n
        nl = len(spec.widths)
        dev = ws[0].device
        dws = [torch.empty_like(w) for w in ws]
        dbs = [None if b is None else torch.empty_like(b) for b in bs]
        need_n, need_f = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        off_n = next((off for kind, off, _ in spec.blocks if kind == 'n'), None)
        off_f, w_f = next(((off, w) for kind, off, w in spec.blocks if kind == 'f'), (None, 0))
        d_normal = torch.empty((n, 3), dtype=torch.float32, device=dev) if (need_n and off_n is not None) else None
        d_feat = torch.empty((n, w_f), dtype=torch.float32, device=dev) if (need_f and off_f is not None) else None
        need_in = d_normal is not None or d_feat is not None
        for c, (lo, hi, ins, masks) in enumerate(ctx.saved):
            acc = c > 0
            g = d_rgb4[lo:hi]

            def tn(dy, xin, j, kw):
                if dbs[j] is not None:
                    F.gemm_tn(dy, xin, want_colsum=True, out=dws[j], db_out=dbs[j], accumulate=acc, **kw)
                else:
                    F.gemm_tn(dy, xin, out=dws[j], accumulate=acc, **kw)
            j = nl - 1
            tn(g, ins[j], j, {})
            d = F.gemm_nn(g, ws[j], ws=ctx.ws_nn[j])
            for j in range(nl - 2, -1, -1):
                m, bits = masks[j]
                kw = {'mask_bits': m} if bits else {'mask': m}
                tn(d, ins[j], j, kw)
                if j > 0 or need_in:
                    d = F.gemm_nn(d, ws[j], ws=ctx.ws_nn[j], **kw)
            if d_normal is not None:
                d_normal[lo:hi].copy_(d[:, off_n:off_n + 3])
            if d_feat is not None:
                d_feat[lo:hi].copy_(d[:, off_f:off_f + w_f])
        ctx.saved = None
        grads = []
        for j in range(nl):
            shp = ctx.shapes[2 * j]
            grads += [dws[j][:shp[0], :shp[1]], None if dbs[j] is None else dbs[j][:shp[0]]]
        grads = [gr if need else None for gr, need in zip(grads, ctx.needs_input_grad[5:])]
        return (None, None, d_normal, d_feat, None) + tuple(grads)


def radiance_chain(radiance_net, x, view_dirs, normals, geo_feat, chunk_pts):
    """radiance (n, 3) of RadianceNet.forward(x, view_dirs, normals, geo_feat) over all points through RadianceChainFn, or None where the
    node does not apply"""
    spec = make_rad_spec(radiance_net, chunk_pts)
    if spec is None:
        return None
    given = {'p': x, 'v': view_dirs, 'n': normals, 'f': geo_feat}
    n = None
    for m in spec.mode:
        t = given[m]
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2):
            return None
        n = t.shape[0] if n is None else n
        if t.shape[0] != n or n == 0:
            return None
    if ('p' in spec.mode and x.requires_grad and x.grad_fn is not None) or ('v' in spec.mode and view_dirs.requires_grad):
        return None      # positions / directions that depend on something learnable: the module path differentiates the encoders
    for kind, _, w in spec.blocks:
        if given[kind].shape[1] != (3 if kind in 'pv' else w):
            return None
    params = effective_params(radiance_net)
    if any(p is not None and (p.dtype != torch.float32 or not p.is_cuda) for p in params):
        return None
    from ..geometry.ray import normalize
    rgb4 = RadianceChainFn.apply(x.detach().contiguous() if 'p' in spec.mode else None,
                                 normalize(view_dirs).contiguous() if 'v' in spec.mode else None,
                                 normals if 'n' in spec.mode else None, geo_feat if 'f' in spec.mode else None, spec, *params)
    rgb = rgb4[:, :3]
    return torch.sigmoid(rgb) if spec.sigmoid else rgb.contiguous()
