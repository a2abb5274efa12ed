"""torch.autograd.Function wrappers around the HIP kernels (forward + hand-written backward each).

The reference gets its backward from torch autograd over ~25 tiny kernels per encoder level; here
every Function is one forward kernel and one backward kernel.
"""
import torch
from torch.autograd.function import once_differentiable

from . import functional as F


def _accumulate_node(p):
    """the AccumulateGrad node of a leaf parameter (autograd keeps one per leaf and hands the same one out while it lives).  Looked up
    per call, NOT cached on the tensor: the node holds the parameter strongly, so parameter -> node -> parameter would be a cycle through
    C++ that Python's collector cannot see - the flat optimiser storage behind the parameter (200 MB for the NGP set) would never be freed."""
    with torch.enable_grad():
        return p.view_as(p).grad_fn.next_functions[0][0]


def direct_grad(p):
    """The gradient buffer a hand-written backward may ADD a parameter's gradient into directly, or None: the .grad of a parameter that
    lives in FusedAdam's flat buffers (optim.FusedAdam.flatten marks it `_arcn_direct_grad`).  A node that used it returns None for that
    input: no zero-filled temporary, no AccumulateGrad pass over it (48.8 MB written, read and added again per table scatter).

    Only when the engine run that called us WOULD accumulate into that .grad: `loss.backward()` does, `torch.autograd.grad(geo, x)`
    (BaseGeoNet.forward_with_grad: normals) does not and must leave every .grad alone, and `torch.autograd.grad(loss, [p])` wants the
    gradient returned.  Asked from the engine itself (`_will_engine_execute_node` on the parameter's AccumulateGrad node; it raises for
    a captured leaf) - to be called from inside a backward()."""
    if not getattr(p, '_arcn_direct_grad', False):     # (asked first: reading .grad of a non-leaf tensor warns)
        return None
    g = p.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.shape != p.shape:
        return None
    try:
        if not torch._C._will_engine_execute_node(_accumulate_node(p)):
            return None
    except (RuntimeError, AttributeError):
        return None
    return g


class HashGridFn(torch.autograd.Function):
    """out (n, L*F) = hash-grid encode(xyz; table): the node that carries the TABLE gradient (binned scatter).

    The xyz gradient lives on a second node (HashGridXyzFn, value 0) added by `hashgrid_encode` when xyz requires grad.  Two
    nodes because autograd then prunes by itself: torch.autograd.grad(sdf, xyz) - the normals of an sdf model - never reaches
    this node, so no table scatter runs for it, and a plain backward() of a density model never touches the xyz node."""

    @staticmethod
    def forward(ctx, xyz, table, desc, scatter_ws):
        xyz = xyz.contiguous().float()
        out = F.hashgrid_fwd(xyz, table, desc)
        ctx.save_for_backward(xyz, table)
        ctx.desc, ctx.ws = desc, scatter_ws
        return out

    @staticmethod
    @once_differentiable   # first order only: a second differentiation through this node raises instead of being silently wrong
    def backward(ctx, dout):
        xyz, table = ctx.saved_tensors
        dtable = dxyz = None
        if ctx.needs_input_grad[1]:
            if ctx.needs_input_grad[0]:   # direct use with a differentiable xyz (first order only): both in one kernel
                dtable, dxyz = F.hashgrid_bwd(xyz, table, dout.contiguous(), ctx.desc, want_dtable=True, want_dxyz=True)
            else:
                into = direct_grad(table)
                dtable, _ = F.hashgrid_bwd(xyz, table, dout.contiguous(), ctx.desc, workspace=ctx.ws, dtable=into)
                if into is not None:
                    return dxyz, None, None, None
            dtable = dtable.view_as(table)
        elif ctx.needs_input_grad[0]:
            _, dxyz = F.hashgrid_bwd(xyz, table, dout.contiguous(), ctx.desc, want_dtable=False, want_dxyz=True)
        return dxyz, dtable, None, None


class HashGridDxFn(torch.autograd.Function):
    """dxyz (n,3) = J(xyz; table)^T dout, itself differentiable in dout, table and xyz (arcn_hashgrid_bwd_bwd)."""

    @staticmethod
    def forward(ctx, xyz, table, dout, desc):
        dout = dout.contiguous().float()
        _, dxyz = F.hashgrid_bwd(xyz, table, dout, desc, want_dtable=False, want_dxyz=True)
        ctx.save_for_backward(xyz, table, dout)
        ctx.desc = desc
        return dxyz

    @staticmethod
    @once_differentiable   # first order only: a second differentiation through this node raises instead of being silently wrong
    def backward(ctx, gdx):
        xyz, table, dout = ctx.saved_tensors
        need_x, need_t, need_d = ctx.needs_input_grad[:3]
        into = direct_grad(table) if need_t else None
        ddout, dtable, d2x = F.hashgrid_bwd_bwd(xyz, gdx.contiguous(), table, dout, ctx.desc, want_ddout=need_d, want_dtable=need_t,
                                                want_d2xyz=need_x, dtable=into)
        return d2x, (dtable.view_as(table) if (dtable is not None and into is None) else None), ddout, None


class HashGridXyzFn(torch.autograd.Function):
    """Value 0, gradient d encode / d xyz: the xyz-side node of the encoding (see HashGridFn).  Its backward is built from a
    differentiable function, so create_graph=True gives a graph through which a loss on the input gradient reaches the table."""

    @staticmethod
    def forward(ctx, xyz, table, desc, n_out):
        ctx.save_for_backward(xyz, table)
        ctx.desc = desc
        return xyz.new_zeros((xyz.shape[0], n_out))

    @staticmethod
    def backward(ctx, dout):
        xyz, table = ctx.saved_tensors
        return HashGridDxFn.apply(xyz, table, dout, ctx.desc), None, None, None


def hashgrid_encode(xyz, table, desc, scatter_ws=True):
    """(n, L*F) hash-grid features, differentiable in table and - to second order - in xyz"""
    xyz = xyz.contiguous().float()
    out = HashGridFn.apply(xyz.detach(), table, desc, scatter_ws)
    if xyz.requires_grad and torch.is_grad_enabled():
        out = out + HashGridXyzFn.apply(xyz, table, desc, out.shape[1])
    return out


class FreqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_freqs, include_input):
        x = x.contiguous().float()
        ctx.save_for_backward(x)
        ctx.cfg = (n_freqs, include_input)
        return F.freq_fwd(x, n_freqs, include_input)

    @staticmethod
    @once_differentiable   # first order only: a second differentiation through this node raises instead of being silently wrong
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        return F.freq_bwd(x, dout.contiguous(), *ctx.cfg), None, None


class TruncExpFn(torch.autograd.Function):
    """arcnerf/ops/trunc_exp.py:7-37: fwd exp(x); bwd g * exp(clamp(x, -15, 15))"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        y = F.act_fwd(x, 'truncexp')
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    @once_differentiable   # first order only: a second differentiation through this node raises instead of being silently wrong
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        return F.act_bwd(x, y, g.contiguous(), 'truncexp')


class ActGradFn(torch.autograd.Function):
    """dx = dy f'(x) of an elementwise activation, itself differentiable in dy and x (arcn_act_bwd_bwd)"""

    @staticmethod
    def forward(ctx, x, dy, act, beta):
        ctx.save_for_backward(x, dy)
        ctx.act, ctx.beta = act, beta
        return F.act_bwd(x, None, dy.contiguous(), act, beta)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, dy = ctx.saved_tensors
        ddy, d2x = F.act_bwd_bwd(x, dy, g.contiguous(), ctx.act, ctx.beta, want_ddy=ctx.needs_input_grad[1], want_d2x=ctx.needs_input_grad[0])
        return d2x, ddy, None, None


class ActFn(torch.autograd.Function):
    """y = f(x) for an activation of the kernels' table (relu / sigmoid / truncexp / softplus / squareplus / sine), twice differentiable:
    the elementwise forms of the activations the reference's fused-MLP map names (tcnn_fusedmlp_module.py:195-213)"""

    @staticmethod
    def forward(ctx, x, act, beta=1.0):
        x = x.contiguous().float()
        ctx.save_for_backward(x)
        ctx.act, ctx.beta = act, beta
        return F.act_fwd(x, act, beta)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return ActGradFn.apply(x, g, ctx.act, ctx.beta), None, None


class HuberMeanFn(torch.autograd.Function):
    """mean(Huber_delta(x - y)) of the reference's ImgLoss (arcnerf/loss/img_loss.py:80-100: 0.5 / delta d^2 inside the band, |d| - 0.5 delta
    outside) with its gradient in the same kernel (arcn_huber_loss_grad): three launches for what the elementwise chain does in fourteen"""

    @staticmethod
    def forward(ctx, x, y, delta):
        loss, dx = F.huber_loss_grad(x.contiguous(), y.contiguous(), delta, 1.0)
        ctx.save_for_backward(dx)
        ctx.shape = x.shape
        return loss[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        dx, = ctx.saved_tensors
        return (dx * g).view(ctx.shape), None, None


class FusedMlpFn(torch.autograd.Function):
    """y = MLP(x; flat weights[, flat biases]) on the f32-MFMA fused kernel."""

    @staticmethod
    def forward(ctx, x, weights, biases, desc):
        x = x.contiguous().float()
        need = any(ctx.needs_input_grad[:3])
        if need:
            out, acts = F.mlp_fwd(x, weights, biases, desc, save_acts=True)
            ctx.save_for_backward(x, weights, biases if biases is not None else x.new_zeros(0), out, acts)
        else:
            out = F.mlp_fwd(x, weights, biases, desc)
        ctx.desc = desc
        ctx.has_bias = biases is not None
        return out

    @staticmethod
    @once_differentiable   # first order only: a second differentiation through this node raises instead of being silently wrong
    def backward(ctx, dout):
        x, weights, biases, out, acts = ctx.saved_tensors
        b = biases if ctx.has_bias else None
        gw, gb = direct_grad(weights), (direct_grad(b) if b is not None else None)   # (the kernels add into the buffers they are given)
        dx, dw, db = F.mlp_bwd(x, weights, b, ctx.desc, out, acts, dout.contiguous(), want_dx=ctx.needs_input_grad[0], dweights=gw, dbiases=gb)
        return dx, (None if gw is not None else dw), ((None if gb is not None else db) if ctx.has_bias else None), None


class RayMarchingFn(torch.autograd.Function):
    """ray_marching of arcnerf/render/ray_helper.py:476-593 on dense (R,P) tensors; differentiable wrt sigma (or alpha)
    and radiance.  Returns rgb, depth, mask, alpha, trans_shift, weights, status and t_last = trans_shift[:, -1] as a
    DIFFERENTIABLE output (FullModel scales the background model's colour and depth with it); the other per-sample outputs
    are detached views."""

    @staticmethod
    def forward(ctx, sigma, radiance, zvals, alpha, bkg_color, noise, add_inf_z, white_bkg):
        out = F.ray_marching_fwd(sigma, radiance, zvals, add_inf_z=add_inf_z, white_bkg=white_bkg, alpha=alpha,
                                 bkg_color=bkg_color, noise=noise, want_samples=True, check_order=True)
        ctx.flags = (add_inf_z, white_bkg)
        ctx.set_materialize_grads(False)
        ctx.has = (sigma is not None, radiance is not None, alpha is not None, bkg_color is not None, noise is not None)
        dummy = zvals.new_zeros(0)
        ctx.save_for_backward(*[t if t is not None else dummy for t in (sigma, radiance, zvals, alpha, bkg_color, noise)])
        ctx.status = out['status']
        rgb = out['rgb'] if out['rgb'] is not None else zvals.new_zeros(0)
        ctx.mark_non_differentiable(out['alpha'], out['trans_shift'])
        t_last = out['trans_shift'][:, -1].clone()
        return rgb, out['depth'], out['mask'], out['alpha'], out['trans_shift'], out['weights'], out['status'], t_last

    @staticmethod
    @once_differentiable   # first order only: a second differentiation through this node raises instead of being silently wrong
    def backward(ctx, d_rgb, d_depth, d_mask, _da, _dt, d_w, _ds, d_tlast):
        sigma, radiance, zvals, alpha, bkg, noise = [t if h else None for t, h in
                                                     zip(ctx.saved_tensors, (ctx.has[0], ctx.has[1], True, ctx.has[2], ctx.has[3], ctx.has[4]))]
        add_inf_z, white_bkg = ctx.flags
        R, P = zvals.shape
        # gradients are NOT materialised (forward sets set_materialize_grads(False)): an output nobody differentiated arrives as None,
        # so "is there a gradient on the per-sample weights" is a host-side test, no reduction, no device read
        zr = zvals.new_zeros(R)
        d_depth = zr if d_depth is None else d_depth.contiguous()
        d_mask = zr if d_mask is None else d_mask.contiguous()
        if radiance is not None:
            d_rgb = zvals.new_zeros((R, 3)) if d_rgb is None else d_rgb.contiguous()
        else:
            d_rgb = None
        d_geo, d_rad = F.ray_marching_bwd(sigma, radiance, zvals, d_rgb, d_depth, d_mask, add_inf_z=add_inf_z, white_bkg=white_bkg,
                                          alpha=alpha, bkg_color=bkg, noise=noise,
                                          d_tlast=None if d_tlast is None else d_tlast.contiguous())
        if d_w is not None:
            # weights_i = alpha_i T_i: their gradient is the gradient of sum_i w_i * d_w_i, i.e. of a compositing pass whose "colour" is
            # d_w (first channel) with d_rgb = (1, 0, 0) and no background term (ray_helper.py:596-620; e.g. the NeuS normal map)
            Pe = d_w.shape[1]
            col = zvals.new_zeros((R, P, 3))
            col[:, :Pe, 0] = d_w
            e0 = zvals.new_zeros((R, 3))
            e0[:, 0] = 1.0
            d_geo_w, _ = F.ray_marching_bwd(sigma, col, zvals, e0, zr, zr, add_inf_z=add_inf_z, white_bkg=False, alpha=alpha,
                                            bkg_color=None, noise=noise)
            d_geo = d_geo + d_geo_w
        # with alpha= given, sigma is only recorded (NeuS passes the sdf there, ray_helper.py:550-556): it gets no gradient
        d_sigma = d_geo if (sigma is not None and alpha is None) else None
        return d_sigma, d_rad, None, (d_geo if alpha is not None else None), None, None, None, None


class PackedCompositeFn(torch.autograd.Function):
    """rgb (R,3), depth (R), mask (R) = alpha compositing of PACKED samples (sigma (S), radiance (S,3), t (S), offsets (R+1)):
    the result the reference's padded dense tensors would give (width p_dense = the longest ray, tails repeating the last sample),
    without materialising them.  Gradients for sigma and radiance."""

    @staticmethod
    def forward(ctx, sigma, radiance, t, offsets, p_dense_dev, add_inf_z, white_bkg, noise):
        out = F.composite_packed_fwd(sigma, radiance, t, offsets, p_dense_dev=p_dense_dev, add_inf_z=add_inf_z, white_bkg=white_bkg,
                                     noise=noise)
        ctx.save_for_backward(sigma, radiance, t, offsets, p_dense_dev, noise)
        ctx.flags = (bool(add_inf_z), bool(white_bkg))
        return out['rgb'], out['depth'], out['mask']

    @staticmethod
    @once_differentiable   # first order only: a second differentiation through this node raises instead of being silently wrong
    def backward(ctx, d_rgb, d_depth, d_mask):
        sigma, radiance, t, offsets, p_dense_dev, noise = ctx.saved_tensors
        d_sigma, d_rad = F.composite_packed_bwd(sigma, radiance, t, offsets, d_rgb.contiguous(), d_depth.contiguous(), d_mask.contiguous(),
                                                p_dense_dev=p_dense_dev, add_inf_z=ctx.flags[0], white_bkg=ctx.flags[1], noise=noise)
        return d_sigma, d_rad, None, None, None, None, None, None


class NeusPackedRenderFn(torch.autograd.Function):
    """rgb (R,3), depth, mask (R), normal (R,3), t_last (R) of NeuS from PACKED per-point sdf / radiance / normal (csrc/neus.hip): slope,
    cos annealing, sdf_to_alpha, weights and sums in one kernel per direction; the numbers of the reference's padded chain
    (neus_model.py:63-104).  Gradients for sdf, radiance, normal and the scale s."""

    @staticmethod
    def forward(ctx, sdf, radiance, normal, s, pk, rays_d, cos_anneal, bkg_color, depth_far, dflt_rgb, dflt_nrm):
        sdf, radiance, normal = sdf.contiguous(), radiance.contiguous(), normal.contiguous()
        s_dev = s.detach().reshape(1).contiguous()
        out = F.neus_render_fwd(sdf, radiance, normal, pk, rays_d, s_dev, cos_anneal, bkg_color, depth_far, dflt_rgb, dflt_nrm)
        ctx.save_for_backward(sdf, radiance, normal, s_dev, rays_d, *( [bkg_color] if bkg_color is not None else []))
        ctx.pk, ctx.cos_anneal, ctx.has_bkg, ctx.s_shape = pk, float(cos_anneal), bkg_color is not None, s.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d_rgb, d_depth, d_mask, d_nrm, d_tlast):
        sdf, radiance, normal, s_dev, rays_d = ctx.saved_tensors[:5]
        bkg = ctx.saved_tensors[5] if ctx.has_bkg else None
        d_sdf, d_rad, d_normal, d_s_ray = F.neus_render_bwd(sdf, radiance, normal, ctx.pk, rays_d, s_dev, ctx.cos_anneal, bkg,
                                                             d_rgb.contiguous(), d_depth.contiguous(), d_mask.contiguous(),
                                                             d_nrm.contiguous(), d_tlast.contiguous())
        d_s = d_s_ray.sum().reshape(ctx.s_shape) if ctx.needs_input_grad[3] else None
        return d_sdf, d_rad, d_normal, d_s, None, None, None, None, None, None, None


class NeusSlotsFn(torch.autograd.Function):
    """packed per-point normals (S, 3) -> the dense (rays, P, 3) `normal_pts` of the reference's output (padded slots repeat a ray's last
    point, rays without samples hold the default normal); backward = the transpose, without atomics"""

    @staticmethod
    def forward(ctx, packed, offsets, p_dense, dflt):
        ctx.save_for_backward(offsets)
        ctx.p_dense, ctx.n = int(p_dense), packed.shape[0]
        return F.neus_slots_fwd(packed.contiguous(), offsets, p_dense, dflt)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        offsets, = ctx.saved_tensors
        return F.neus_slots_bwd(g.contiguous(), offsets, ctx.p_dense, ctx.n)[:ctx.n], None, None, None


class SdfToAlphaFn(torch.autograd.Function):
    """NeuS sdf_to_alpha (arcnerf/models/neus_model.py:242-265) with gradients w.r.t. the mid-point sdf, the slope and the
    scale s (a tensor: exp(10 * inv_s) of the learnable parameter; a float scale gets no gradient).  zvals carry no gradient,
    as in the reference's use (they are detached samples)."""

    @staticmethod
    def forward(ctx, mid_sdf, zvals, mid_slope, s, clip):
        ctx.clip = bool(clip)
        ctx.s_is_tensor = torch.is_tensor(s)
        s_t = s if ctx.s_is_tensor else F.scalar_tensor(s, zvals.device)
        ctx.save_for_backward(mid_sdf, zvals, mid_slope, s_t)
        return F.sdf_to_alpha_fwd(mid_sdf, zvals, mid_slope, s_t, clip=ctx.clip)

    @staticmethod
    @once_differentiable   # first order only: a second differentiation through this node raises instead of being silently wrong
    def backward(ctx, d_alpha):
        mid_sdf, zvals, mid_slope, s_t = ctx.saved_tensors
        d_sdf, d_slope, d_s = F.sdf_to_alpha_bwd(mid_sdf, zvals, mid_slope, s_t, d_alpha.contiguous(), clip=ctx.clip)
        d_s = d_s.reshape(s_t.shape) if (ctx.s_is_tensor and ctx.needs_input_grad[3]) else None
        return d_sdf, None, d_slope, d_s, None


# ------------------------------------------------------------------------------------------------------------------------------------
# nn.Linear on the hand-written f32-MFMA products (csrc/gemm.hip).  The three products are each other's gradients, so the graph is
# differentiable to any order: NeuS takes d sdf / d x with create_graph=True and differentiates the Eikonal loss through it again
# (base_network.py:30-44), which needs the double backward of every layer of the sdf net.
# ------------------------------------------------------------------------------------------------------------------------------------
class GemmNT(torch.autograd.Function):
    """y (S,N) = x (S,K) @ w (N,K).T [+ bias]"""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return F.gemm_nt(x, w, bias)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        dx = GemmNN.apply(g, w) if ctx.needs_input_grad[0] else None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and want_db:
            dw, db = GemmTNB.apply(g, x)      # the bias gradient (column sums of g) from the weight-gradient pass
        else:
            dw = GemmTN.apply(g, x) if ctx.needs_input_grad[1] else None
            db = g.sum(0) if want_db else None
        return dx, dw, db


class GemmNN(torch.autograd.Function):
    """dx (S,K) = dy (S,N) @ w (N,K)"""

    @staticmethod
    def forward(ctx, dy, w):
        ctx.save_for_backward(dy, w)
        return F.gemm_nn(dy, w)

    @staticmethod
    def backward(ctx, g):
        dy, w = ctx.saved_tensors
        g = g.contiguous()
        d_dy = GemmNT.apply(g, w, None) if ctx.needs_input_grad[0] else None
        d_w = GemmTN.apply(dy, g) if ctx.needs_input_grad[1] else None
        return d_dy, d_w


class GemmTN(torch.autograd.Function):
    """dw (N,K) = a (S,N).T @ b (S,K)"""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return F.gemm_tn(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        d_a = GemmNT.apply(b, g, None) if ctx.needs_input_grad[0] else None
        d_b = GemmNN.apply(a, g) if ctx.needs_input_grad[1] else None
        return d_a, d_b


class GemmTNB(torch.autograd.Function):
    """(dw (N,K), db (N)) = (a (S,N).T @ b (S,K), column sums of a): a layer's weight and bias gradients in one pass over a"""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return F.gemm_tn(a, b, want_colsum=True)

    @staticmethod
    def backward(ctx, g_w, g_b):
        a, b = ctx.saved_tensors
        d_a = d_b = None
        if ctx.needs_input_grad[0]:
            d_a = GemmNT.apply(b, g_w.contiguous(), None) if g_w is not None else None
            if g_b is not None:
                d_a = g_b.unsqueeze(0).expand_as(a) if d_a is None else d_a + g_b.unsqueeze(0)
        if ctx.needs_input_grad[1] and g_w is not None:
            d_b = GemmNN.apply(a, g_w.contiguous())
        return d_a, d_b


class ToneMapFn(torch.autograd.Function):
    """HDR-NeRF's per-channel 1 -> W -> 1 tone mappers (hdrnerf_model.py:44-75) as one kernel per direction: x (n, C), params
    (C, 3 W + 1) = [w1 | b1 | w2 | b2] per channel.  First order only."""

    @staticmethod
    def forward(ctx, x, params):
        y = F.tonemap_fwd(x, params)
        ctx.save_for_backward(x, params, y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, params, y = ctx.saved_tensors
        dx, dparams = F.tonemap_bwd(x, y, g.contiguous(), params, want_dx=ctx.needs_input_grad[0])
        return dx, (dparams if ctx.needs_input_grad[1] else None)


class SoftplusFn(torch.autograd.Function):
    """torch.nn.functional.softplus(z, beta, threshold=20) differentiable twice on fused kernels: the sdf nets of NeuS evaluate it,
    take d sdf / d x through it with create_graph=True (SoftplusGradFn) and differentiate the Eikonal loss through that again."""

    @staticmethod
    def forward(ctx, z, beta):
        ctx.save_for_backward(z)
        ctx.beta = beta
        return F.act_fwd(z, 'softplus', beta)

    @staticmethod
    def backward(ctx, g):
        z, = ctx.saved_tensors
        return SoftplusGradFn.apply(g.contiguous(), z, ctx.beta, False), None


class SoftplusGradFn(torch.autograd.Function):
    """out = g * sigmoid(beta z); backward (first order from here: a third differentiation raises): dg = h s, dz = h g beta s (1 - s)
    from ONE pass over z, g, h.  from_y: the second argument is y = softplus(z) (sigmoid(beta z) = 1 - exp(-beta y)) and the second
    gradient is the one with respect to y."""

    @staticmethod
    def forward(ctx, g, z, beta, from_y=False):
        ctx.save_for_backward(g, z)
        ctx.beta, ctx.from_y = beta, from_y
        return F.softplus_grad(z, g, beta, from_y)

    @staticmethod
    @once_differentiable
    def backward(ctx, h):
        g, z = ctx.saved_tensors
        dg, dz = F.softplus_grad2(z, g, h.contiguous(), ctx.beta, want_dg=ctx.needs_input_grad[0], want_dz=ctx.needs_input_grad[1],
                                  from_y=ctx.from_y)
        return dg, dz, None, None


class LinearSoftplusFn(torch.autograd.Function):
    """y = softplus_beta(x @ w.T + bias) with the activation in the product's epilogue: the pre-activation is never written.  The
    backward is a composition of twice-differentiable nodes (SoftplusGradFn from y, GemmNN, GemmTNB), so the normals of NeuS
    (create_graph=True) and the Eikonal loss through them differentiate it again."""

    @staticmethod
    def forward(ctx, x, w, bias, beta):
        y = F.gemm_nt(x, w, bias, act='softplus', beta=beta)
        ctx.save_for_backward(x, w, y)
        ctx.beta, ctx.has_bias = beta, bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        gs = SoftplusGradFn.apply(g.contiguous(), y, ctx.beta, True)
        dx = GemmNN.apply(gs, w) if ctx.needs_input_grad[0] else None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and want_db:
            dw, db = GemmTNB.apply(gs, x)
        else:
            dw = GemmTN.apply(gs, x) if ctx.needs_input_grad[1] else None
            db = gs.sum(0) if want_db else None
        return dx, dw, db, None


def linear_softplus(x, weight, bias, beta):
    """softplus(F.linear(x, weight, bias), beta) as one fused, twice-differentiable layer; None where the HIP products do not apply"""
    if not _use_hip_linear(x, weight):
        return None
    shp = x.shape
    x2, w, b, n_out, npad = _padded_operands(x, weight, bias)
    y = LinearSoftplusFn.apply(x2, w, b, float(beta))
    if npad:
        y = y[:, :n_out]
    return y.reshape(*shp[:-1], n_out)


def softplus(z, beta):
    """softplus on the fused twice-differentiable kernels for fp32 CUDA tensors; CPU tensors (host-side tests) and empty ones: torch"""
    _no_silent_cuda_fallback(z)
    if not (z.is_cuda and z.numel() > 0):
        return torch.nn.functional.softplus(z, beta=beta)
    return SoftplusFn.apply(z.contiguous(), float(beta))


FUSED_SDF_NET = True      # (False: SdfMlpJacFn as its chain of dense products - the reference the fused kernels are tested against)


class SdfMlpJacFn(torch.autograd.Function):
    """The two-layer softplus sdf net of NeuS on the hash grid (GeoNet D = 1, no bias: out = W2 softplus_beta(W1 f)) together with the
    Jacobian row of its first output, J = d out[:, 0] / d f = W1^T (s * W2[0]) with s = sigmoid(beta W1 f) - as EXPLICIT outputs of a
    first-order node.  The normals are then d enc / d x applied to J (HashGridDxFn), and the Eikonal / radiance losses reach the
    weights through the gradient of J - no create_graph, no second differentiation of every layer (base_network.py:30-44 builds the
    same quantity by differentiating the layer stack twice).  Products on the exact-f32 kernels of csrc/gemm.hip.
        backward, for incoming g_out (S, O) and g_J (S, K):   dh = g_out W2,  u = g_J W1^T,
        dz = dh s + W2[0] u beta s (1 - s),   df = dz W1,   dW1 = dz^T f + diag(W2[0]) s^T g_J,   dW2 = g_out^T h,  dW2[0] += sum_s s u"""

    @staticmethod
    def _fused(f, w1, w2):
        """the NGP shape - 32 features, 64 hidden, <= 32 outputs, f32 on the GPU: one forward and one backward kernel (arcn_geo2_fwd / _bwd on the
        row-major features), the hidden layer in registers; anything else keeps the chain of dense products below"""
        return (FUSED_SDF_NET and tuple(w1.shape) == (64, 32) and w2.shape[1] == 64 and w2.shape[0] <= 32 and f.shape[0] > 0 and f.is_cuda
                and f.dtype == w1.dtype == w2.dtype == torch.float32 and f.data_ptr() % 16 == 0)

    @staticmethod
    def forward(ctx, f, w1, w2, beta):
        f = f.contiguous()
        ctx.fused = SdfMlpJacFn._fused(f, w1, w2)
        if ctx.fused:
            w1c, w2c = w1.contiguous(), w2.contiguous()
            out, _, jac = F.geo2_fwd(f, f.shape[0], w1c, w2c, True, beta, rows=True)
            ctx.save_for_backward(f, w1c, w2c)
            ctx.beta = beta
            return out[:, :w2.shape[0]], jac
        h = F.gemm_nt(f, w1, None, act='softplus', beta=beta)             # (S, H)
        out = F.gemm_nt(h, w2, None)                                      # (S, O)
        s = F.softplus_grad(h, None, beta, from_y=True)                   # sigmoid(beta z) from y = softplus(z), one pass
        jac = F.gemm_nn(s, (w1 * w2[0][:, None]).contiguous())             # (S, K) = s (diag(W2[0]) W1)
        ctx.save_for_backward(f, w1, w2, h)
        ctx.beta = beta
        return out, jac

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out, g_jac):
        if ctx.fused:
            f, w1, w2 = ctx.saved_tensors
            n, n_out = f.shape[0], w2.shape[0]
            g_out = g_out if (g_out.dim() == 2 and g_out.stride(1) == 1) else g_out.contiguous()
            if g_jac is None:
                g_jac = torch.zeros((n, 32), dtype=torch.float32, device=f.device)
            flat = torch.zeros(64 * 32 + n_out * 64, dtype=torch.float32, device=f.device)      # the kernel ADDS both layers' gradients into views of one buffer
            dw1, dw2 = flat[:2048].view(64, 32), flat[2048:].view(n_out, 64)
            df = F.geo2_bwd(f, n, w1, w2, True, ctx.beta, g_out[:, 0], g_out[:, 1:], dw1, dw2, d_jac=g_jac.contiguous(), rows=True)
            return (df if ctx.needs_input_grad[0] else None), dw1, dw2, None
        f, w1, w2, h = ctx.saved_tensors
        beta = ctx.beta
        s = F.softplus_grad(h, None, beta, from_y=True)
        w20 = w2[0]
        dz = F.gemm_nn(g_out.contiguous(), w2)                            # dh
        dw2 = F.gemm_tn(g_out.contiguous(), h)
        dw1 = None
        if g_jac is not None:
            g_jac = g_jac.contiguous()
            u = F.gemm_nt(g_jac, w1, None)                                # (S, H) = g_J W1^T
            dz, su = F.sdf_jac_dz(dz, u, s, (beta * w20).contiguous())    # dz = dh s + W2[0] u beta s (1 - s), su = s u: one pass, in place
            dw2[0] += su.sum(0)
            dw1 = F.gemm_tn(s, g_jac) * w20[:, None]
        else:
            dz = dz * s
        df = F.gemm_nn(dz, w1) if ctx.needs_input_grad[0] else None
        dw1_a = F.gemm_tn(dz, f)
        dw1 = dw1_a if dw1 is None else dw1 + dw1_a
        return df, dw1, dw2, None


class LinearReluFn(torch.autograd.Function):
    """y = relu(x @ w.T + bias) in ONE kernel (bias + activation in the product's epilogue); backward with the activation's mask
    folded into the operand loads of the two gradient products (dpre = dy * (y > 0) is never written): per layer and direction one
    launch instead of the library's GEMM + elementwise passes.  Under create_graph (a second differentiation through the input gradient
    of a ReLU stack: Eikonal / normal losses) the backward is rebuilt from the products that are closed under differentiation."""

    @staticmethod
    def forward(ctx, x, w, bias):
        # where the split kernels run and the width allows, the mask of the backward is kept as BITS (1/32 of y's bytes): the two
        # masked gradient products read dy + bits instead of dy + y
        n_out, k_in = w.shape
        ctx.has_bias = bias is not None
        # the weight split of the input-gradient product, taken now where one forward runs the layer on several chunks (one split for
        # all of them, F.split_weight_scope); None: the backward product splits for itself
        ctx.ws_nn = None
        if ctx.needs_input_grad[0] and F.in_split_scope() and F._use_split(x, n_out, k_in):
            ctx.ws_nn = F.split_weights(w, True)
        if any(ctx.needs_input_grad) and F.relu_bits_supported(x, k_in, n_out) and F._use_split(x, n_out, k_in) and n_out > 64:
            y, bits = F.gemm_nt(x, w, bias, act='relu', want_bits=True)
            ctx.use_bits = True
            ctx.save_for_backward(x, w, bits, bias if bias is not None else x.new_empty(0))
        else:
            y = F.gemm_nt(x, w, bias, act='relu')
            ctx.use_bits = False
            ctx.save_for_backward(x, w, y, x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, m, bias = ctx.saved_tensors
        g = g.contiguous()
        if torch.is_grad_enabled():
            # create_graph: somebody differentiates THROUGH this backward (an Eikonal / normal loss on a ReLU sdf net, a radiance mode
            # with 'n' on a ReLU geometry net: base_network.py:30-44).  The masked kernels are first order only, so the gradient is built
            # from the three products that are closed under differentiation; the mask itself is piecewise constant.
            with torch.no_grad():     # only the bits were kept: the mask is recomputed from the layer's own forward
                y = F.gemm_nt(x, w, bias if ctx.has_bias else None, act='relu') if ctx.use_bits else m
                mask = (y > 0).to(g.dtype)
            gm = g * mask
            dx = GemmNN.apply(gm, w) if ctx.needs_input_grad[0] else None
            dw = GemmTN.apply(gm, x) if ctx.needs_input_grad[1] else None
            db = gm.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
            return dx, dw, db
        if g.data_ptr() % 16 != 0:
            g = g.clone()      # a view at an odd offset: the masked forms of the products want 16-byte aligned rows (a fresh buffer is)
        kw = {'mask_bits': m} if ctx.use_bits else {'mask': m}
        dx = F.gemm_nn(g, w, ws=ctx.ws_nn, **kw) if ctx.needs_input_grad[0] else None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        dw = db = None
        if ctx.needs_input_grad[1] and want_db:
            dw, db = F.gemm_tn(g, x, want_colsum=True, **kw)   # bias gradient = column sums of dy * (y > 0), from the same pass
        elif ctx.needs_input_grad[1]:
            dw = F.gemm_tn(g, x, **kw)
        elif want_db:
            db = F.gemm_tn(g, F._ones_cols(g.shape[0], g.device), **kw)[:, 0].contiguous()
        return dx, dw, db


class LinearReluCatFn(torch.autograd.Function):
    """cat([relu(x @ w.T + bias), tail], -1) - a DenseLayer followed by the skip concatenation of GeoNet
    (linear_network_module.py:174-197) - without the concatenation pass: the product writes its columns of the (S, N + T) result at
    that row stride, the tail is one strided copy of T columns (64 of 320 for the positional embedding), and the gradient products read
    their N columns of the incoming gradient in place.  N and N + T multiples of 4 (16-byte aligned rows)."""

    @staticmethod
    def forward(ctx, x, w, bias, tail):
        n_out, k_in = w.shape
        S, T = x.shape[0], tail.shape[1]
        ctx.has_bias, ctx.n_out = bias is not None, n_out
        buf = torch.empty((S, n_out + T), dtype=torch.float32, device=x.device)
        y = buf[:, :n_out]
        ctx.ws_nn = None
        if ctx.needs_input_grad[0] and F.in_split_scope() and F._use_split(x, n_out, k_in):
            ctx.ws_nn = F.split_weights(w, True)
        if any(ctx.needs_input_grad[:3]) and F.relu_bits_supported(x, k_in, n_out) and F._use_split(x, n_out, k_in) and n_out > 64:
            _, bits = F.gemm_nt(x, w, bias, act='relu', want_bits=True, out=y)
            ctx.use_bits = True
            ctx.save_for_backward(x, w, bits, bias if bias is not None else x.new_empty(0))
        else:
            F.gemm_nt(x, w, bias, act='relu', out=y)
            ctx.use_bits = False
            ctx.save_for_backward(x, w, buf, x.new_empty(0))
        buf[:, n_out:].copy_(tail)
        return buf

    @staticmethod
    def backward(ctx, g):
        x, w, m, bias = ctx.saved_tensors
        n = ctx.n_out
        g = g.contiguous()
        d_tail = g[:, n:] if ctx.needs_input_grad[3] else None
        if torch.is_grad_enabled():      # create_graph: see LinearReluFn.backward
            with torch.no_grad():
                y = F.gemm_nt(x, w, bias if ctx.has_bias else None, act='relu') if ctx.use_bits else m[:, :n]
                mask = (y > 0).to(g.dtype)
            gm = g[:, :n] * mask
            dx = GemmNN.apply(gm, w) if ctx.needs_input_grad[0] else None
            dw = GemmTN.apply(gm, x) if ctx.needs_input_grad[1] else None
            db = gm.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
            return dx, dw, db, d_tail
        if g.data_ptr() % 16 != 0:
            g = g.clone()
            d_tail = g[:, n:] if ctx.needs_input_grad[3] else None
        gv = g[:, :n]
        kw = {'mask_bits': m} if ctx.use_bits else {'mask': m[:, :n]}
        dx = F.gemm_nn(gv, w, ws=ctx.ws_nn, **kw) if ctx.needs_input_grad[0] else None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        dw = db = None
        if ctx.needs_input_grad[1] and want_db:
            dw, db = F.gemm_tn(gv, x, want_colsum=True, **kw)
        elif ctx.needs_input_grad[1]:
            dw = F.gemm_tn(gv, x, **kw)
        elif want_db:
            db = F.gemm_tn(gv, F._ones_cols(g.shape[0], g.device), **kw)[:, 0].contiguous()
        return dx, dw, db, d_tail


def linear_relu_cat(x, weight, bias, tail):
    """cat([relu(F.linear(x, weight, bias)), tail], -1) with the layer's product writing straight into the concatenated buffer
    (LinearReluCatFn); None where that form does not apply (the caller then concatenates as the reference does)"""
    n_out, k_in = weight.shape
    if not _use_hip_linear(x, weight):
        return None
    if x.dim() != 2 or tail.dim() != 2 or tail.dtype != torch.float32 or n_out % 4 or (n_out + tail.shape[1]) % 4 or tail.shape[0] != x.shape[0]:
        return None
    x2, w, b, _, npad = _padded_operands(x, weight, bias)
    return LinearReluCatFn.apply(x2, w, b, tail)


def _padded_operands(x, weight, bias):
    """the kernels move 16 bytes per lane when every row starts 16-byte aligned: feature dimensions that are not multiples of 4 (63,
    319, 283 inputs; 257, 17, 3 outputs) are zero-padded - the weight / bias pads are tiny, the input pad is one copy, and the padded
    OUTPUT is sliced outside the autograd node so that its gradient arrives padded (aligned) as well"""
    n_out, k_in = weight.shape
    kp, npad = (-k_in) % 4, (-n_out) % 4
    pad = torch.nn.functional.pad
    # (under chunk_processing the padded forms are made once for all chunks: one pad, one node for the chunks' gradients to meet in)
    w = F.scope_cached(('padw', weight.data_ptr(), weight._version, n_out, k_in), weight, lambda: pad(weight, (0, kp, 0, npad))) if (kp or npad) else weight
    b = F.scope_cached(('padb', bias.data_ptr(), bias._version, n_out), bias, lambda: pad(bias, (0, npad))) if (bias is not None and npad) else bias
    x2 = x.reshape(-1, x.shape[-1])
    if x2.shape[-1] == k_in + kp:      # already padded by the caller (pad_cols4: one pad shared by a layer input and a skip concat)
        x2 = x2.contiguous()
    else:
        x2 = pad(x2, (0, kp)) if kp else x2.contiguous()
    return x2, w.contiguous(), b, n_out, npad


def pad_cols4(x):
    """x with zero columns appended up to a multiple of 4 where the dense layers run on the HIP products (their padded input width),
    x itself otherwise: lets a module pad ONCE what it feeds to several layers / concatenations instead of one pad copy per layer"""
    kp = (-x.shape[-1]) % 4
    if not kp or not (x.is_cuda and x.dtype == torch.float32):
        return x
    return torch.nn.functional.pad(x, (0, kp))


def _no_silent_cuda_fallback(*tensors):
    """The dense layers of the product path are the hand-written MFMA products, for float32.  A CUDA tensor of another dtype used to take
    torch's route (hipBLASLt) without a word - a dtype slip would then measure the library, not the product: it raises instead.  CPU
    tensors (the host-side tests that run the module logic without a GPU) go to torch."""
    for t in tensors:
        if t is not None and t.is_cuda and t.dtype != torch.float32:
            raise RuntimeError('arcnerf_amd: a {} CUDA tensor reached a dense layer of the product path - the HIP kernels compute in float32 and '
                               'there is no library fallback (cast the input / parameters to float32)'.format(t.dtype))


def _use_hip_linear(x, weight):
    """True: the HIP products run.  False: CPU tensors (torch, host-side tests) or an empty batch.  CUDA tensors that are not float32 raise."""
    _no_silent_cuda_fallback(x, weight)
    return x.is_cuda and weight.is_cuda and x.numel() > 0


def linear_relu(x, weight, bias=None):
    """relu(torch.nn.functional.linear(x, weight, bias)) as one fused layer (LinearReluFn); torch for CPU tensors (host-side tests)"""
    if not _use_hip_linear(x, weight):
        return torch.relu(linear(x, weight, bias))
    shp = x.shape
    x2, w, b, n_out, npad = _padded_operands(x, weight, bias)
    y = LinearReluFn.apply(x2, w, b)
    if npad:
        y = y[:, :n_out]
    return y.reshape(*shp[:-1], n_out)


def linear_act_nograd(x, weight, bias, act, beta=1.0):
    """act(F.linear(x, weight, bias)) with the activation in the product's epilogue, for passes that build no graph (the importance
    sampling rounds of NeuS evaluate the softplus sdf net four times per step under no_grad); None when the fast path does not apply"""
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return None
    if not _use_hip_linear(x, weight):
        return None
    shp = x.shape
    x2, w, b, n_out, npad = _padded_operands(x.detach(), weight.detach(), None if bias is None else bias.detach())
    y = F.gemm_nt(x2, w, b, act=act, beta=beta)
    if npad:
        y = y[:, :n_out]
    return y.reshape(*shp[:-1], n_out)


def linear(x, weight, bias=None, keep_pad=False):
    """torch.nn.functional.linear on the MFMA products (fp32 CUDA tensors); CPU tensors (the host-side tests) and empty batches go to
    torch, a CUDA tensor of another dtype raises (no silent library route).  keep_pad: return the product's own (rows, out + pad) tensor
    (zero columns up to a multiple of 4) for a caller that splits it itself."""
    if not _use_hip_linear(x, weight):
        return torch.nn.functional.linear(x[..., :weight.shape[1]], weight, bias)    # (a pad_cols4 input on an empty batch)
    shp = x.shape
    x2, w, b, n_out, npad = _padded_operands(x, weight, bias)
    y = GemmNT.apply(x2, w, b)
    if keep_pad:
        return y.reshape(*shp[:-1], n_out + npad)
    if npad:
        y = y[:, :n_out]
    return y.reshape(*shp[:-1], n_out)
