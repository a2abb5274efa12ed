"""The field of a vanilla-NeRF family model - GeoNet (frequency encoding, ReLU DenseLayers with skip concatenation, a final Linear giving
[sigma | feature]) followed by RadianceNet in mode 'vf' - as ONE autograd node over all the points of a pass
(linear_network_module.py:16-335 under base_3d_model.py's chunk_processing; configs 1 and 5 of BASELINE.json).

The products are the very launches of ops.autograd.linear_relu / linear (csrc/gemm.hip: split-bf16 forms with ReLU bit masks, exact f32
for the narrow ones); what the node removes is everything BETWEEN them, which was a seventh of the step's device time:
  * the buffers are laid out so that no concatenation, slice copy or pad exists: the positional encoding is written by its kernel into
    the tail columns of the skip buffer (layer 0 reads it there, the skip layer's product writes the head columns); the last geometry
    layer - rows permuted to [feature | sigma | 0 0 0] - writes straight into the radiance net's input buffer, whose first layer has its
    columns permuted to [feature | view] to match, and the view encoding lands behind the feature by its own kernel; on the way back the
    same buffers are read in place (sigma's gradient is dropped into its column);
  * the chunk loop (model.chunk_pts points per launch, as the yaml has it) runs INSIDE the node: weight and bias gradients are summed
    over the chunks by the weight-gradient product itself (accumulate flag), not by one autograd add per parameter and chunk, the weights
    are padded / permuted / split once per pass, and the outputs are written into their slices of one tensor.
Column permutations reorder the k-sum of the first radiance layer, so the outputs equal the layer-by-layer path to f32 summation order
(1e-6), not bit for bit; tests/test_gpu_kernels.py::test_field_chain_equals_the_layer_by_layer_modules, G22 / G24.
First order only, inputs without gradient; anything else (NeuS: softplus + double backward, weight norm, other modes) keeps the
module-by-module path.  ARCN_FIELD_CHAIN=0 turns the node off."""

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable

from . import functional as F


def _r4(v):
    return (v + 3) // 4 * 4


class ChainSpec:
    """static description of an eligible (GeoNet, RadianceNet) pair"""

    def __init__(self, geo, rad, chunk):
        self.chunk = int(chunk)
        self.D, self.W, self.W_feat, self.skips = geo.D, geo.W, geo.W_feat, sorted(set(geo.skips))
        self.pos_freqs, self.pos_input = geo.embed_fn.n_freqs, bool(geo.embed_fn.include_input)
        self.ed = geo.embed_dim
        self.edp = _r4(self.ed)
        self.view_freqs, self.view_input = rad.embed_fn_view.n_freqs, bool(rad.embed_fn_view.include_input)
        self.vd = rad.embed_fn_view.get_output_dim()
        self.Np = self.W_feat + 4                               # final geometry layer, permuted: [feature | sigma | 0 0 0]
        self.ldR = max(_r4(self.W_feat + self.vd), self.Np)     # radiance input, permuted: [feature | view | 0 ...]
        self.rad_widths = [layer.out_features for layer in rad.layers]
        self.rad_sigmoid = isinstance(rad.layers[-1].activation, nn.Sigmoid)
        self.geo_bias = [layer.bias is not None for layer in geo.layers]
        self.rad_bias = [layer.bias is not None for layer in rad.layers]


def make_spec(geo, rad, chunk):
    """ChainSpec when the two nets are exactly the stack the node computes, else None"""
    from ..models.base_modules.encoding.freq_encoder import FreqEmbedder
    from ..models.base_modules.geo_rad_model.linear_network_module import GeoNet, RadianceNet
    from ..models.base_modules.linear import DenseLayer, Linear
    if type(geo) is not GeoNet or type(rad) is not RadianceNet:
        return None
    if type(geo.embed_fn) is not FreqEmbedder or geo.embed_fn.n_freqs < 1:      # (any input width: NeRF++ feeds (x / r, 1 / r))
        return None
    if geo.norm_skip or geo.out_act is not None or geo.W_feat <= 0 or geo.W_feat % 4 or geo.W % 4 or geo.D < 1:
        return None
    if any(s < 0 or s >= geo.D for s in geo.skips):
        return None
    for i, layer in enumerate(geo.layers):
        if hasattr(layer, 'weight_g'):
            return None
        want_in = geo.embed_dim if i == 0 else (geo.W + geo.embed_dim if (i - 1) in geo.skips else geo.W)
        if i == geo.D:
            if type(layer) is not Linear or layer.out_features != 1 + geo.W_feat or layer.in_features != want_in:
                return None
        elif type(layer) is not DenseLayer or type(layer.activation) is not nn.ReLU or layer.out_features != geo.W or layer.in_features != want_in:
            return None
    if rad.mode != 'vf' or rad._fused_desc is not None or type(rad.embed_fn_view) is not FreqEmbedder or rad.embed_fn_view.input_dim != 3:
        return None
    if rad.embed_fn_view.n_freqs < 1 or rad.W_feat_in != geo.W_feat or len(rad.layers) < 2:
        return None
    vd = rad.embed_fn_view.get_output_dim()
    for j, layer in enumerate(rad.layers):
        last = j == len(rad.layers) - 1
        if type(layer) is not DenseLayer or hasattr(layer, 'weight_g'):
            return None
        want_in = vd + geo.W_feat if j == 0 else rad.layers[j - 1].out_features
        if layer.in_features != want_in:
            return None
        if last:
            if layer.out_features != 3 or not isinstance(layer.activation, (nn.Sigmoid, nn.Identity)):
                return None
        elif type(layer.activation) is not nn.ReLU or layer.out_features % 4:
            return None
    return ChainSpec(geo, rad, chunk)


def _prepare(spec, params):
    """the weights in the layouts of the node's buffers (detached copies, made once per pass)"""
    with torch.no_grad():
        D, nr = spec.D, len(spec.rad_widths)
        pad = torch.nn.functional.pad
        gw, gb, rw, rb = [], [], [], []
        for i in range(D):
            w, b = params[2 * i], params[2 * i + 1]
            gw.append(pad(w, (0, (-w.shape[1]) % 4)).contiguous())       # 63 -> 64, 319 -> 320 input columns
            gb.append(b)
        w, b = params[2 * D], params[2 * D + 1]
        gw.append(torch.cat([w[1:], w[:1], w.new_zeros(3, w.shape[1])], dim=0).contiguous())
        gb.append(None if b is None else torch.cat([b[1:], b[:1], b.new_zeros(3)]))
        base = 2 * (D + 1)
        for j in range(nr):
            w, b = params[base + 2 * j], params[base + 2 * j + 1]
            if j == 0:
                w = torch.cat([w[:, spec.vd:], w[:, :spec.vd], w.new_zeros(w.shape[0], spec.ldR - spec.W_feat - spec.vd)], dim=1)
            if j == nr - 1:
                w = pad(w, (0, 0, 0, 1))
                b = None if b is None else pad(b, (0, 1))
            rw.append(w.contiguous())
            rb.append(b)
    return gw, gb, rw, rb


def _relu_layer(x_in, w, b, out, want_mask):
    """out = relu(x_in @ w.T + b) written in place; the backward's mask: the bit words where the split kernels write them, else `out`"""
    n_out, k_in = w.shape
    if want_mask and F.relu_bits_supported(x_in, k_in, n_out) and F._use_split(x_in, n_out, k_in) and n_out > 64:
        _, bits = F.gemm_nt(x_in, w, b, act='relu', want_bits=True, out=out)
        return bits, True
    F.gemm_nt(x_in, w, b, act='relu', out=out)
    return out, False


class FieldChainFn(torch.autograd.Function):
    """(sigma (n), pre-activation rgb padded to 4 columns (n, 4)) = field(pts (n, D), unit dirs (n, 3)); params = weight, bias (or None)
    of the geometry layers 0 .. D, then of the radiance layers"""

    @staticmethod
    def forward(ctx, pts, dirs, spec, *params):
        n, dev = pts.shape[0], pts.device
        want = any(ctx.needs_input_grad[3:])
        gw, gb, rw, rb = _prepare(spec, params)
        D, W, Wf, skips, nr = spec.D, spec.W, spec.W_feat, spec.skips, len(spec.rad_widths)
        sigma = torch.empty(n, dtype=torch.float32, device=dev)
        rgb4 = torch.empty((n, 4), dtype=torch.float32, device=dev)
        chunk = spec.chunk if spec.chunk > 0 else n
        saved = []
        with F.split_weight_scope():
            for lo in range(0, n, chunk):
                hi = min(n, lo + chunk)
                S = hi - lo
                sk = {i: torch.empty((S, W + spec.edp), dtype=torch.float32, device=dev) for i in skips}
                xe = sk[skips[0]][:, W:] if skips else torch.empty((S, spec.edp), dtype=torch.float32, device=dev)
                F.freq_fwd_cols(pts[lo:hi], spec.pos_freqs, spec.pos_input, xe)
                for i in skips[1:]:
                    sk[i][:, W:].copy_(xe)
                ins, masks = [], []
                cur = xe
                for i in range(D):
                    out = sk[i][:, :W] if i in sk else torch.empty((S, W), dtype=torch.float32, device=dev)
                    m = _relu_layer(cur, gw[i], gb[i], out, want)
                    ins.append(cur)
                    masks.append(m)
                    cur = sk[i] if i in sk else out
                R = torch.empty((S, spec.ldR), dtype=torch.float32, device=dev)
                F.gemm_nt(cur, gw[D], gb[D], out=R[:, :spec.Np])
                ins.append(cur)
                sigma[lo:hi].copy_(R[:, Wf])
                F.freq_fwd_cols(dirs[lo:hi], spec.view_freqs, spec.view_input, R[:, Wf:])
                cur = R
                for j in range(nr - 1):
                    out = torch.empty((S, spec.rad_widths[j]), dtype=torch.float32, device=dev)
                    m = _relu_layer(cur, rw[j], rb[j], out, want)
                    ins.append(cur)
                    masks.append(m)
                    cur = out
                F.gemm_nt(cur, rw[nr - 1], rb[nr - 1], out=rgb4[lo:hi])
                ins.append(cur)
                if want:
                    saved.append((lo, hi, ins, masks))
            if want:
                # the weight splits of the input-gradient products, once for all chunks (None: that product is exact f32 and splits nothing)
                probe = saved[0][2]
                ctx.ws_nn = [F.split_weights(w, True) if F._use_split(probe[0], w.shape[0], w.shape[1]) else None for w in gw + rw]
        ctx.spec, ctx.saved, ctx.prep = spec, saved, (gw, gb, rw, rb)
        ctx.shapes = [None if p is None else p.shape for p in params]
        return sigma, rgb4

    @staticmethod
    @once_differentiable
    def backward(ctx, d_sigma, d_rgb4):
        if ctx.saved is None:
            raise RuntimeError('FieldChainFn: the activations of this pass were released by its first backward (retain_graph is not supported)')
        spec, (gw, gb, rw, rb) = ctx.spec, ctx.prep
        D, W, Wf, skips, nr = spec.D, spec.W, spec.W_feat, set(spec.skips), len(spec.rad_widths)
        dev = gw[0].device
        dgw = [torch.empty_like(w) for w in gw]
        dgb = [None if b is None else torch.empty_like(b) for b in gb]
        drw = [torch.empty_like(w) for w in rw]
        drb = [None if b is None else torch.empty_like(b) for b in rb]
        ws_g, ws_r = ctx.ws_nn[:D + 1], ctx.ws_nn[D + 1:]
        for c, (lo, hi, ins, masks) in enumerate(ctx.saved):
            acc = c > 0
            S = hi - lo
            g = d_rgb4[lo:hi] if d_rgb4 is not None else torch.zeros((S, 4), dtype=torch.float32, device=dev)

            def tn(dy, x, dw, db, kw):
                if db is not None:
                    F.gemm_tn(dy, x, want_colsum=True, out=dw, db_out=db, accumulate=acc, **kw)
                else:
                    F.gemm_tn(dy, x, out=dw, accumulate=acc, **kw)
            # radiance net, last layer first
            j = nr - 1
            tn(g, ins[D + 1 + j], drw[j], drb[j], {})
            d = F.gemm_nn(g, rw[j], ws=ws_r[j])
            for j in range(nr - 2, -1, -1):
                m, bits = masks[D + j]
                kw = {'mask_bits': m} if bits else {'mask': m}
                tn(d, ins[D + 1 + j], drw[j], drb[j], kw)
                d = F.gemm_nn(d, rw[j], ws=ws_r[j], **kw)
            # d = gradient of the radiance input [feature | view | pad]: the geometry layer's gradient [feature | sigma | 0 0 0] in place
            if d_sigma is not None:
                d[:, Wf].copy_(d_sigma[lo:hi])
                d[:, Wf + 1:spec.Np].zero_()
            else:
                d[:, Wf:spec.Np].zero_()
            dy = d[:, :spec.Np]
            tn(dy, ins[D], dgw[D], dgb[D], {})
            d = F.gemm_nn(dy, gw[D], ws=ws_g[D])
            for i in range(D - 1, -1, -1):
                dy = d[:, :W] if i in skips else d
                m, bits = masks[i]
                kw = {'mask_bits': m} if bits else {'mask': m}
                tn(dy, ins[i], dgw[i], dgb[i], kw)
                if i > 0:
                    d = F.gemm_nn(dy, gw[i], ws=ws_g[i], **kw)
        ctx.saved = None
        # back to the parameters' own layouts
        grads = []
        for i in range(D):
            shp = ctx.shapes[2 * i]
            grads += [dgw[i][:, :shp[1]], dgb[i]]
        grads += [torch.cat([dgw[D][Wf:Wf + 1], dgw[D][:Wf]], dim=0), None if dgb[D] is None else torch.cat([dgb[D][Wf:Wf + 1], dgb[D][:Wf]])]
        for j in range(nr):
            w, b = drw[j], drb[j]
            if j == nr - 1:
                w, b = w[:3], (None if b is None else b[:3])
            if j == 0:
                w = torch.cat([w[:, Wf:Wf + spec.vd], w[:, :Wf]], dim=1)
            grads += [w, b]
        grads = [gr if need else None for gr, need in zip(grads, ctx.needs_input_grad[3:])]
        return (None, None, None) + tuple(grads)


def field_chain(geo_net, radiance_net, pts, dirs, chunk_pts):
    """(sigma (n), radiance (n, 3)) of _forward_pts_dir over all points through FieldChainFn, or None where the node does not apply"""
    if not (torch.is_tensor(pts) and pts.is_cuda and pts.dtype == torch.float32 and pts.dim() == 2 and pts.shape[0] > 0):
        return None
    if dirs is None or dirs.dim() != 2 or dirs.shape[0] != pts.shape[0] or dirs.shape[1] != 3 or dirs.dtype != torch.float32:
        return None
    if pts.requires_grad or dirs.requires_grad:
        return None
    spec = make_spec(geo_net, radiance_net, chunk_pts)
    if spec is None or pts.shape[1] != geo_net.embed_fn.input_dim:
        return None
    params = []
    for layer in list(geo_net.layers) + list(radiance_net.layers):
        params += [layer.weight, layer.bias]
    if any(p is not None and (p.dtype != torch.float32 or not p.is_cuda) for p in params):
        return None
    from ..geometry.ray import normalize
    sigma, rgb4 = FieldChainFn.apply(pts.contiguous(), normalize(dirs).contiguous(), spec, *params)
    rgb = rgb4[:, :3]
    return sigma, (torch.sigmoid(rgb) if spec.rad_sigmoid else rgb.contiguous())
