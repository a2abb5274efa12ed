"""The sdf net of NeuS on the frequency encoding - GeoNet with softplus DenseLayers, skip concatenation (optionally reduced and
1 / sqrt(2)-normalised), weight norm, final Linear giving [sdf | feature] - TOGETHER WITH ITS NORMAL d sdf / d x as one first-order
autograd node over all the points of a pass (linear_network_module.py:16-197 evaluated through base_network.py:30-44, i.e. differentiated
with create_graph = True and then differentiated again by the Eikonal / radiance losses; config 3 of BASELINE.json).

    forward      a_0 = e(x);  h_i = sp(W_i a_i + b_i);  a_{i+1} = h_i, or [h_i | e] / sqrt2 behind a skip;  out = W_D a_D + b_D
    normal       g_{D-1} = W_D[0];  p_i = g_i * s_i  (s_i = sp'(z_i) = 1 - exp(-beta h_i));  abar_i = p_i W_i;  g_{i-1} = abar_i (its head / sqrt2
                 behind a skip, the tail / sqrt2 joins ebar);  ebar += abar_0;  n = (d e / d x)^T ebar
    backward     for incoming (d_out, d_n):  UP the layers the adjoint of the normal chain - ehat = (d e / d x) d_n, ahat_0 = ehat,
                 phat_i = ahat_i W_i^T, dW_i += p_i^T ahat_i, (ghat_i, dy_i) = (phat_i s_i, phat_i g_i beta (1 - s_i)), ahat_{i+1} = ghat_i (or
                 [ghat_i | ehat] / sqrt2), dW_D[0] += sum_s ghat_{D-1} - then DOWN the ordinary chain with the second-order term joined in:
                 hbar_i = abar'_{i+1} + dy_i, zbar_i = hbar_i s_i, dW_i += zbar_i^T a_i, db_i += sum zbar_i, abar'_i = zbar_i W_i.
The products are the ones torch's autograd would issue for the same graph (six per layer and sample chunk); what the node removes is the
graph around them: the encoding built from torch sin / cos so that it can be differentiated twice, one `add_` per parameter, chunk and
differentiation order, the concatenations / slices / fills of the double backward (a quarter of the step's device time).  Weight-norm
parameters stay outside (the node takes the effective weights, torch differentiates `_weight_norm`).  x gets no gradient (sample positions
are not learnable); inputs that need one keep the module path.  ARCN_SDF_CHAIN=0 turns the node off.
tests/test_gpu_kernels.py::test_sdf_chain_equals_the_double_backward_of_the_modules, golden G23."""
import math

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable

from . import functional as F


def _r4(v):
    return (v + 3) // 4 * 4


class SdfSpec:
    def __init__(self, geo, chunk):
        self.chunk = int(chunk)
        self.D, self.W_feat, self.skips, self.norm_skip = geo.D, geo.W_feat, sorted(set(geo.skips)), bool(geo.norm_skip)
        self.beta = float(geo.layers[0].activation.beta)
        self.pos_freqs, self.pos_input, self.ed = geo.embed_fn.n_freqs, bool(geo.embed_fn.include_input), geo.embed_dim
        self.out_dims = [layer.out_features for layer in geo.layers]
        self.in_dims = [layer.in_features for layer in geo.layers]
        self.has_bias = [layer.bias is not None for layer in geo.layers]


def make_sdf_spec(geo, chunk):
    from ..models.base_modules.encoding.freq_encoder import FreqEmbedder
    from ..models.base_modules.geo_rad_model.linear_network_module import GeoNet
    from ..models.base_modules.linear import DenseLayer, Linear
    if type(geo) is not GeoNet or geo.use_siren:
        return None
    if type(geo.embed_fn) is not FreqEmbedder or geo.embed_fn.input_dim != 3 or geo.embed_fn.n_freqs < 1:
        return None
    if geo.out_act is not None or geo.W_feat <= 0 or geo.D < 1 or any(s < 0 or s >= geo.D - 1 for s in geo.skips):
        return None
    beta = None
    for i, layer in enumerate(geo.layers):
        if i == geo.D:
            if type(layer) is not Linear or layer.out_features != 1 + geo.W_feat:
                return None
            continue
        if type(layer) is not DenseLayer or type(layer.activation) is not nn.Softplus or layer.activation.threshold != 20:
            return None
        if beta is not None and layer.activation.beta != beta:
            return None
        beta = layer.activation.beta
        if layer.out_features % 4 and i not in geo.skips:
            return None
    for i in range(geo.D + 1):       # input widths as GeoNet.__init__ builds them, concatenations a multiple of 4 wide or padded up to one
        want = geo.embed_dim if i == 0 else (geo.layers[i - 1].out_features + geo.embed_dim if (i - 1) in geo.skips else geo.layers[i - 1].out_features)
        if geo.layers[i].in_features != want:
            return None
    return SdfSpec(geo, chunk)


def effective_params(geo):
    """weight (through weight norm where the layer has it: the module's own pre-forward hook is not run by the node) and bias per layer"""
    params = []
    for layer in geo.layers:
        w = torch._weight_norm(layer.weight_v, layer.weight_g, 0) if hasattr(layer, 'weight_g') else layer.weight
        params += [w, layer.bias]
    return params


_SQ2 = math.sqrt(2)


def padded_params(geo):
    """(ws, bs, w_sdf): the effective (weight-normed) weights and biases of every layer as the kernels take them - input columns and output
    rows padded to multiples of 4 (63 -> 64 / 319 -> 320 columns, 257 -> 260 rows; zero rows and columns that never leave the node) - and the
    first row of the last layer, under no_grad.  Built ONCE per parameter state: the key is every parameter's (pointer, version) plus
    utils.param_epoch (the optimiser kernels write through raw pointers).  A NeuS step evaluates this net in seven graph-free passes and one
    differentiated one; per pass it was 9 weight-norm launches + 18 pads."""
    from ..utils import param_epoch
    key = (param_epoch.current(),) + tuple((p.data_ptr(), p._version) for p in geo.layers.parameters())
    c = getattr(geo, '_padded_cache', None)
    if c is not None and c[0] == key:
        return c[1]
    pad = torch.nn.functional.pad
    with torch.no_grad():
        params = effective_params(geo)
        D = geo.D
        ws = [pad(params[2 * i], (0, (-params[2 * i].shape[1]) % 4, 0, (-params[2 * i].shape[0]) % 4)).contiguous() for i in range(D)]
        bs = [None if params[2 * i + 1] is None else pad(params[2 * i + 1], (0, (-params[2 * i + 1].shape[0]) % 4)) for i in range(D)]
        n_last = params[2 * D].shape[0]
        ws.append(pad(params[2 * D], (0, (-params[2 * D].shape[1]) % 4, 0, (-n_last) % 4)).contiguous())
        bs.append(None if params[2 * D + 1] is None else pad(params[2 * D + 1], (0, (-n_last) % 4)))
        w_sdf = ws[D][0].contiguous()
    val = (ws, bs, w_sdf)
    object.__setattr__(geo, '_padded_cache', (key, val))
    return val


def sdf_forward_nograd(geo, x):
    """GeoNet.forward (linear_network_module.py:174-197) -> (sdf (..., 1), feature (..., W_feat)) for a pass that builds no graph, on the
    cached padded weights: encoding, D softplus layers with the activation in the product's epilogue, the skip concatenation [h | e] / sqrt2
    as one kernel, the last layer - no module hooks (weight norm), no per-layer pads, no torch cat / div.  None where it does not apply."""
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in geo.layers.parameters())):
        return None
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.numel() > 0 and x.shape[-1] == 3):
        return None
    spec = make_sdf_spec(geo, 0)
    if spec is None or any(p.dtype != torch.float32 or not p.is_cuda for p in geo.layers.parameters()):
        return None
    ws, bs, _ = padded_params(geo)
    shp = x.shape
    xc = x.detach().reshape(-1, 3).contiguous()
    S, dev = xc.shape[0], xc.device
    div = _SQ2 if spec.norm_skip else 1.0
    with torch.no_grad(), F.split_weight_scope():
        a0 = torch.empty((S, ws[0].shape[1]), dtype=torch.float32, device=dev)
        F.freq_fwd_cols(xc, spec.pos_freqs, spec.pos_input, a0)
        cur = a0
        for i in range(spec.D):
            h = F.gemm_nt(cur, ws[i], bs[i], act='softplus', beta=spec.beta)
            cur = F.concat2_div(h[:, :spec.out_dims[i]], a0[:, :spec.ed], div, ws[i + 1].shape[1]) if i in spec.skips else h
        out = F.gemm_nt(cur, ws[spec.D], bs[spec.D])
    n_out = 1 + spec.W_feat
    out = out[:, :n_out].reshape(*shp[:-1], n_out)
    return out[..., :1], out[..., 1:]


class SdfChainFn(torch.autograd.Function):
    """(out (n, r4(1 + W_feat)) = [sdf | feature | 0 ...], normal (n, 3)) = sdf net and its input gradient at x (n, 3)"""

    @staticmethod
    def forward(ctx, x, spec, *params):
        n, dev = x.shape[0], x.device
        D, skips, ed, beta = spec.D, spec.skips, spec.ed, spec.beta
        want = any(ctx.needs_input_grad[2:])
        # (63 -> 64 / 319 -> 320 input columns; a reduced skip layer's 193 -> 196 output rows, 257 -> 260 rows of the last layer: zero rows whose
        # columns never leave the node; w_sdf = g of the last hidden layer.  The values of `params`, padded once per parameter state)
        ws, bs, w_sdf = spec.padded
        div = _SQ2 if spec.norm_skip else 1.0
        No = ws[D].shape[0]
        out_all = torch.empty((n, No), dtype=torch.float32, device=dev)
        normal = torch.empty((n, 3), dtype=torch.float32, device=dev)
        chunk = spec.chunk if spec.chunk > 0 else n
        saved = []
        with F.split_weight_scope():
            for lo in range(0, n, chunk):
                hi = min(n, lo + chunk)
                S = hi - lo
                xc = x[lo:hi]
                a0 = torch.empty((S, ws[0].shape[1]), dtype=torch.float32, device=dev)
                F.freq_fwd_cols(xc, spec.pos_freqs, spec.pos_input, a0)
                ins, hs = [], []
                cur = a0
                for i in range(D):
                    h = F.gemm_nt(cur, ws[i], bs[i], act='softplus', beta=beta)
                    ins.append(cur)
                    hs.append(h)
                    if i in skips:      # [h | e] (/ sqrt2) at the next layer's padded input width, one pass (e = the first layer's input)
                        cur = F.concat2_div(h[:, :spec.out_dims[i]], a0[:, :ed], div, ws[i + 1].shape[1])
                    else:
                        cur = h
                F.gemm_nt(cur, ws[D], bs[D], out=out_all[lo:hi])
                ins.append(cur)
                # the normal: d out[:, 0] / d x down the layers
                gs, ps = [None] * D, [None] * D
                ebar = None
                g = None
                for i in range(D - 1, -1, -1):
                    if i == D - 1:
                        p = F.softplus_grad_row(hs[i], w_sdf[:hs[i].shape[1]], beta, True)      # (s * W_D[0], the row broadcast inside the kernel)
                        gs[i] = None      # (the row W_D[0] itself, kept implicit)
                    else:
                        gs[i] = g
                        p = F.softplus_grad(hs[i], g, beta, True)
                    ps[i] = p
                    abar = F.gemm_nn(p, ws[i])
                    if i == 0:
                        e0 = abar[:, :ed]
                        ebar = e0.contiguous() if ebar is None else ebar + e0
                    elif (i - 1) in skips:
                        wo = spec.out_dims[i - 1]
                        g = F.concat2_div(abar[:, :wo], None, div, ws[i - 1].shape[0])      # (the head / sqrt2, zero pad behind it)
                        tail = abar[:, wo:wo + ed] / _SQ2 if spec.norm_skip else abar[:, wo:wo + ed].contiguous()
                        ebar = tail if ebar is None else ebar + tail
                    else:
                        g = abar
                normal[lo:hi].copy_(F.freq_bwd(xc, ebar, spec.pos_freqs, spec.pos_input))
                if want:
                    saved.append((lo, hi, ins, hs, gs, ps))
            if want:
                probe = saved[0][2][0]
                ctx.ws_nn = [F.split_weights(w, True) if F._use_split(probe, w.shape[0], w.shape[1]) else None for w in ws]
                ctx.ws_nt = [F.split_weights(w, False) if F._use_split(probe, w.shape[1], w.shape[0]) else None for w in ws]
        ctx.spec, ctx.saved, ctx.ws, ctx.bs, ctx.w_sdf = spec, saved, ws, bs, w_sdf
        ctx.x = x
        ctx.shapes = [None if p is None else p.shape for p in params]
        return out_all, normal

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out, d_n):
        if ctx.saved is None:
            raise RuntimeError('SdfChainFn: the activations of this pass were released by its first backward (retain_graph is not supported)')
        spec, ws, bs = ctx.spec, ctx.ws, ctx.bs
        D, skips, ed, beta = spec.D, set(spec.skips), spec.ed, spec.beta
        dev = ws[0].device
        dws = [torch.zeros_like(w) for w in ws]              # every product below ADDS (two per layer and chunk)
        dbs = [None if b is None else torch.zeros_like(b) for b in bs]
        d_row0 = torch.zeros_like(ctx.w_sdf)                 # the adjoint of g_{D-1} = W_D[0], summed over the samples
        div = _SQ2 if spec.norm_skip else 1.0
        for c, (lo, hi, ins, hs, gs, ps) in enumerate(ctx.saved):
            S = hi - lo

            def tn(dy, xin, i, with_bias):
                if with_bias and dbs[i] is not None:
                    F.gemm_tn(dy, xin, want_colsum=True, out=dws[i], db_out=dbs[i], accumulate=True)
                else:
                    F.gemm_tn(dy, xin, out=dws[i], accumulate=True)
            dys = [None] * D
            if d_n is not None:
                # UP: the adjoint of the normal chain
                ehat = torch.empty((S, ws[0].shape[1]), dtype=torch.float32, device=dev)
                F.freq_jvp_cols(ctx.x[lo:hi], d_n[lo:hi], spec.pos_freqs, spec.pos_input, ehat)
                ahat = ehat
                for i in range(D):
                    phat = F.gemm_nt(ahat, ws[i], None, ws=ctx.ws_nt[i])
                    tn(ps[i], ahat, i, False)
                    if i == D - 1:
                        # g is the row W_D[0]: its adjoint (the column sums of phat * s) goes straight into d_row0, ghat itself is not needed
                        H = hs[i].shape[1]
                        dys[i] = F.softplus_grad2_row(hs[i], ctx.w_sdf[:H], phat, d_row0[:H], beta, True)
                        continue
                    ghat, dys[i] = F.softplus_grad2(hs[i], gs[i], phat, beta, from_y=True)
                    if i in skips:
                        ahat = F.concat2_div(ghat[:, :spec.out_dims[i]], ehat[:, :ed], div, ws[i + 1].shape[1])
                    else:
                        ahat = ghat
            # DOWN: the ordinary backward with the second-order term joined in
            obar = d_out[lo:hi] if d_out is not None else torch.zeros((S, ws[D].shape[0]), dtype=torch.float32, device=dev)
            tn(obar, ins[D], D, True)
            abar = F.gemm_nn(obar, ws[D], ws=ctx.ws_nn[D])
            for i in range(D - 1, -1, -1):
                if i in skips:
                    hbar = F.concat2_div(abar[:, :spec.out_dims[i]], None, div, ws[i].shape[0])
                else:
                    hbar = abar
                zbar = F.softplus_grad(hs[i], hbar, beta, True) if dys[i] is None else F.softplus_grad_sum(hs[i], hbar, dys[i], beta, True)
                tn(zbar, ins[i], i, True)
                if i > 0:
                    abar = F.gemm_nn(zbar, ws[i], ws=ctx.ws_nn[i])
        ctx.saved = None
        dws[D][0] += d_row0
        grads = []
        for i in range(D + 1):
            shp = ctx.shapes[2 * i]
            grads += [dws[i][:shp[0], :shp[1]], None if dbs[i] is None else dbs[i][:shp[0]]]
        grads = [gr if need else None for gr, need in zip(grads, ctx.needs_input_grad[2:])]
        return (None, None) + tuple(grads)


def sdf_chain(geo_net, pts, chunk_pts):
    """(sdf (n, 1), feature (n, W_feat), normal (n, 3)) of GeoNet.forward_with_grad over all points through SdfChainFn, or None where
    the node does not apply"""
    if not (torch.is_tensor(pts) and pts.is_cuda and pts.dtype == torch.float32 and pts.dim() == 2 and pts.shape[0] > 0):
        return None
    if pts.requires_grad and pts.grad_fn is not None:      # positions that depend on something learnable want d / d x as well
        return None
    spec = make_sdf_spec(geo_net, chunk_pts)
    if spec is None:
        return None
    params = effective_params(geo_net)
    if any(p is not None and (p.dtype != torch.float32 or not p.is_cuda) for p in params):
        return None
    spec.padded = padded_params(geo_net)
    out, normal = SdfChainFn.apply(pts.detach().contiguous(), spec, *params)
    # (one split node: its backward is one concatenation of the two gradients and the zero pad)
    geo, feat, _ = torch.split(out, [1, spec.W_feat, out.shape[1] - 1 - spec.W_feat], dim=-1)
    return geo, feat, normal
