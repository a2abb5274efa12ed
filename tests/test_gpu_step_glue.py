"""The passes between the kernels of trainer.FusedNeusNgpStep (csrc/step_glue.hip) against the torch expressions they replace - the reference's
own elementwise code (sdf_model.py:42-101, base_network.py:30-44, full_model.py:278-330, img_loss.py:60-100, geo_loss.py:12-70) - plus the
output-poison check: the buffers this round stopped zero-filling are written in full by their kernels."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def test_neus_step_prep(gpu):
    from arcnerf_amd.ops import functional as F
    g = torch.Generator().manual_seed(0)
    w1, l1w, bl1 = torch.randn(64, 32, generator=g).to(gpu), torch.randn(17, 64, generator=g).to(gpu), torch.randn(17, 64, generator=g).to(gpu)
    inv_s = torch.tensor([0.3], device=gpu)
    o = F.neus_step_prep(w1, l1w, 100.0, inv_s, 10.0, bl1)
    assert o['w2p'].shape == (20, 64) and torch.equal(o['w2p'][:17], l1w) and float(o['w2p'][17:].abs().max()) == 0
    assert torch.equal(o['wb1p'][:17], bl1) and float(o['wb1p'][17:].abs().max()) == 0
    assert torch.equal(o['w1j'], w1 * l1w[0][:, None]) and torch.equal(o['bw20'], 100.0 * l1w[0])
    assert abs(float(o['scale']) - float(torch.exp(inv_s * 10.0))) <= 2e-6 * float(torch.exp(inv_s * 10.0))
    for k in ('w2p', 'w1j', 'bw20'):
        assert o[k].data_ptr() % 16 == 0
    o = F.neus_step_prep(w1, l1w, 100.0)       # no scale, no background net
    assert o['scale'] is None and o['wb1p'] is None and torch.equal(o['w2p'][:17], l1w)


@pytest.mark.parametrize('act', [None, 'truncexp', 'relu'])
def test_geo_out_grad(gpu, act):
    from arcnerf_amd.ops import functional as F
    g = torch.Generator().manual_seed(1)
    n = 5003
    out = torch.randn(n, 20, generator=g).to(gpu) * 3
    out[:5, 0] = torch.tensor([-20.0, 20.0, 14.9, 15.1, 0.0], device=gpu)
    dx = torch.randn(n, 38, generator=g).to(gpu)
    d0 = torch.randn(n, generator=g).to(gpu)
    got = F.geo_out_grad(d0, dx[:, 22:38], 20, out=out, act=act)
    if act is None:
        c0 = d0
    else:
        x = out[:, 0].contiguous()
        c0 = F.act_bwd(x, F.act_fwd(x, act), d0, act)
    want = torch.cat([c0[:, None], dx[:, 22:38], torch.zeros(n, 3, device=gpu)], dim=-1)
    assert torch.equal(got, want)
    if act is not None:     # with the forward's own output handed over
        y = F.act_col_scale(out, act, 1.0)
        assert torch.equal(F.geo_out_grad(d0, dx[:, 22:38], 20, out=out, act=act, y_col0=y), want)


@pytest.mark.parametrize('delta', [0.1, None])
def test_neus_blend_loss(gpu, delta):
    from arcnerf_amd.ops import functional as F
    g = torch.Generator().manual_seed(2)
    for R in (1, 777, 4096, 9001):
        rf, rb, img = (torch.rand(R, 3, generator=g).to(gpu) for _ in range(3))
        df, db, T = (torch.rand(R, generator=g).to(gpu) for _ in range(3))
        o = F.neus_blend_loss(rf, df, T, rb, db, img, delta, 5.0)
        rgb = rf + T[:, None] * rb
        assert torch.equal(o['rgb'], rgb) and torch.equal(o['depth'], df + T * db)
        if delta is not None:
            loss, d_rgb = F.huber_loss_grad(rgb, img, delta, 5.0)
            loss = float(loss[0])
        else:
            diff = rgb - img
            loss, d_rgb = float((diff.double() ** 2).mean() * 5.0), diff * (2.0 * 5.0 / diff.numel())
        assert abs(float(o['loss'][0]) - loss) <= 2e-6 * abs(loss) and float(o['loss'][1]) == 0.0
        assert torch.allclose(o['d_rgb'], d_rgb, rtol=1e-6, atol=0)
        assert torch.allclose(o['d_tlast'], (d_rgb * rb).sum(-1), rtol=1e-5, atol=1e-9)
        assert torch.allclose(o['d_rgb_b'], d_rgb * T[:, None], rtol=1e-6, atol=0)
        o2 = F.neus_blend_loss(rf, df, T, rb, db, img, delta, 5.0)       # one workgroup, fixed order: the same bits
        assert torch.equal(o['loss'], o2['loss'])


@pytest.mark.parametrize('H', [64, 128, 256])
def test_sdf_jac_dz2(gpu, H):
    from arcnerf_amd.ops import functional as F
    g = torch.Generator().manual_seed(3)
    for n in (1, 333, 70001):
        dh, u = torch.randn(n, H, generator=g).to(gpu), torch.randn(n, H, generator=g).to(gpu)
        s = torch.rand(n, H, generator=g).to(gpu)
        c, w = torch.randn(H, generator=g).to(gpu), torch.randn(H, generator=g).to(gpu)
        dz0, su = F.sdf_jac_dz(dh.clone(), u.clone(), s, c)
        acc0 = torch.randn(H, generator=g).to(gpu)
        acc = acc0.clone()
        dz, sw = F.sdf_jac_dz2(dh.clone(), u.clone(), s, c, w, acc)
        assert torch.equal(dz, dz0) and torch.equal(sw, s * w)
        want = su.double().sum(0)
        assert float((acc.double() - acc0.double() - want).abs().max()) <= 1e-5 * max(1.0, float(su.double().abs().sum(0).max()))


def test_sum_scale_add_and_gemm_tn_head(gpu):
    from arcnerf_amd.ops import functional as F
    g = torch.Generator().manual_seed(4)
    src = torch.randn(4096, generator=g).to(gpu)
    dst = torch.tensor([0.25], device=gpu)
    sc = torch.tensor([3.0], device=gpu)
    F.sum_scale_add(src, dst, 10.0, sc)
    assert abs(float(dst) - (0.25 + 30.0 * float(src.double().sum()))) <= 1e-4 * (1 + abs(30.0 * float(src.double().abs().sum())) * 1e-2)
    for n in (100, 50001):
        dy, x = torch.randn(n, 20, generator=g).to(gpu), torch.randn(n, 64, generator=g).to(gpu)
        full = F.gemm_tn(dy, x)
        base = torch.randn(17, 64, generator=g).to(gpu)
        out = base.clone()
        guard = torch.cat([out.reshape(-1), torch.full((3 * 64,), 7.0, device=gpu)])      # the rows behind the head must stay untouched
        view = guard[:17 * 64].view(17, 64)
        F.gemm_tn(dy, x, out=view, accumulate=True, head=17)
        assert torch.allclose(view, base + full[:17], rtol=1e-6, atol=1e-6) and float((guard[17 * 64:] - 7.0).abs().max()) == 0
        assert torch.equal(F.gemm_tn(dy, x, head=17), full[:17])


def test_eikonal_packed_joins_a_second_gradient(gpu):
    from arcnerf_amd.ops import functional as F
    g = torch.Generator().manual_seed(5)
    R = 300
    cnt = torch.randint(0, 40, (R,), generator=g)
    cnt[::7] = 0
    off = torch.zeros(R + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(cnt, 0)
    S = int(off[-1])
    ray_id = torch.repeat_interleave(torch.arange(R, dtype=torch.int32), cnt)
    pk = {'ray_id': ray_id.to(gpu), 'offsets': off.to(gpu), 'p_dense': int(cnt.max())}
    normal = torch.randn(S, 3, generator=g).to(gpu)
    wide = torch.randn(S, 38, generator=g).to(gpu)
    d0 = torch.randn(S, 3, generator=g).to(gpu)
    loss_a, d_a = F.eikonal_packed(normal, pk, R, 0.1, d_normal=(d0 + wide[:, 19:22]).contiguous())
    acc = torch.tensor([123.0, 0.0], device=gpu)
    loss_b, d_b = F.eikonal_packed(normal, pk, R, 0.1, d_normal=d0.clone(), loss=acc[1:2], add_src=wide[:, 19:22], loss_is_clear=True)
    assert torch.allclose(d_a, d_b, rtol=1e-6, atol=1e-7)
    assert abs(float(loss_a) - float(acc[1])) <= 1e-5 * abs(float(loss_a)) and float(acc[0]) == 123.0
    F.eikonal_packed(normal, pk, R, 0.1, d_normal=d0.clone(), loss=acc[1:2], loss_is_clear=True)      # an accumulator: twice the value now
    assert abs(float(acc[1]) - 2 * float(loss_a)) <= 2e-5 * abs(float(loss_a))


def test_unfilled_outputs_are_written_in_full():
    """ARCN_POISON_OUTPUTS=1 fills every buffer that ops.functional hands to a kernel WITHOUT clearing it (the box intersections, the
    cascade marcher's counts, the scans' maxima, the packed compositor's gradients) with NaN / 0x7f first: the tests that consume them -
    samplers against the oracle, the packed compositor, the fused config-4 step against the module path - still pass"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = dict(os.environ, ARCN_POISON_OUTPUTS='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', 'tests/test_gpu_composite.py', 'tests/test_gpu_models.py', 'tests/test_gpu_kernels.py',
                        'tests/test_gpu_geo2.py', '-k', 'fused_neus or packed or multivol or aabb or neus or sdf_net or density_net or module_path_sdf'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout or '')[-1500:]
    assert r.returncode == 0, tail
    assert ' passed' in tail and 'no tests ran' not in tail, tail


# ---- the wide sdf net's glue (ops/sdf_chain.py) ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('from_y', [True, False])
def test_softplus_row_kernels(gpu, from_y):
    from arcnerf_amd.ops import functional as F
    g = torch.Generator().manual_seed(6)
    for n, H in ((1, 256), (4099, 256), (1000, 64), (70001, 128)):
        z = (torch.randn(n, H, generator=g) * 0.05).to(gpu)
        if from_y:
            z = torch.nn.functional.softplus(z, beta=100.0)
        z[0, :4] = torch.tensor([0.0, 0.3, 1e-4, 0.21], device=gpu)       # beta z = 0 / 30 (beyond torch's threshold) / 0.01 / 21
        row, h = torch.randn(H, generator=g).to(gpu), torch.randn(n, H, generator=g).to(gpu)
        full = row.unsqueeze(0).expand(n, -1).contiguous()
        assert torch.equal(F.softplus_grad_row(z, row, 100.0, from_y), F.softplus_grad(z, full, 100.0, from_y))
        dg, dz = F.softplus_grad2(z, full, h, 100.0, from_y=from_y)
        acc0 = torch.randn(H, generator=g).to(gpu)
        acc = acc0.clone()
        got = F.softplus_grad2_row(z, row, h, acc, 100.0, from_y)
        assert torch.equal(got, dz)
        want = dg.double().sum(0)
        assert float((acc.double() - acc0.double() - want).abs().max()) <= 1e-5 * max(1.0, float(dg.double().abs().sum(0).max()))


def test_concat2_div_is_torchs_scalar_division(gpu):
    import math
    from arcnerf_amd.ops import functional as F
    g = torch.Generator().manual_seed(7)
    n = 5003
    a, b = torch.randn(n, 260, generator=g).to(gpu), torch.randn(n, 64, generator=g).to(gpu)
    for div in (math.sqrt(2), 1.0):
        got = F.concat2_div(a[:, :193], b[:, :63], div, 260)
        want = torch.cat([a[:, :193] / div, b[:, :63] / div, torch.zeros(n, 4, device=gpu)], dim=-1)
        assert torch.equal(got, want)          # bit for bit: torch's CUDA kernel multiplies by the float reciprocal too
        got = F.concat2_div(a[:, :193], None, div, 196)
        assert torch.equal(got[:, :193], a[:, :193] / div) and float(got[:, 193:].abs().max()) == 0


def test_sdf_net_graph_free_pass_and_its_weight_cache(gpu, monkeypatch):
    """GeoNet.forward under no_grad (ops.sdf_chain.sdf_forward_nograd: cached weight-normed padded weights, one-kernel skip concatenation)
    against the module path of the same net; the cache follows the parameters - through an optimiser that writes them by raw pointer
    (FusedAdam -> utils.param_epoch) and through torch's own in-place updates (tensor versions)"""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops import sdf_chain
    from arcnerf_amd.optim import FusedAdam
    from arcnerf_amd.utils.cfgs_utils import load_configs
    torch.manual_seed(0)
    m = build_model(load_configs(os.path.join(ROOT, 'configs', 'neus.yaml'), [])).to(gpu)
    geo = m.fg_model.geo_net
    x = (torch.rand(10007, 3, device=gpu) - 0.5) * 2.0

    def both():
        with torch.no_grad():
            geo.nograd_fast = False          # the layer-by-layer modules
            ref = geo(x)
            geo.nograd_fast = True
            assert sdf_chain.sdf_forward_nograd(geo, x) is not None
            got = geo(x)
        for r, o in zip(ref, got):
            assert r.shape == o.shape and torch.allclose(r, o, rtol=1e-5, atol=1e-5), float((r - o).abs().max())
        return got[0].clone()

    s0 = both()
    key0 = geo._padded_cache[0]
    with torch.no_grad():
        geo(x)
    assert geo._padded_cache[0] == key0                      # the same parameter state: the cached weights are reused
    opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-2, eps=1e-15).flatten()
    for _ in range(2):
        opt.zero_grad()
        sdf, _ = geo(x)
        (sdf ** 2).mean().backward()
        opt.step()
        s1 = both()
        assert float((s1 - s0).abs().max()) > 1e-4           # the parameters moved, and both paths saw it
        s0 = s1
    with torch.no_grad():
        for p in geo.layers.parameters():
            p.mul_(1.01)                                      # a torch in-place update: the tensors' own versions
    both()


def test_second_order_gathers_from_the_forwards_corners(gpu):
    """arcn_hashgrid_fwd_corners (the XCD-affine gather) keeps the eight gathered rows of every (sample, level) in level-major quads; the
    normal's gather and the gradient of the Jacobian row computed from them are the table forms bit for bit (points outside the grid
    included), for launches on both sides of the level-major threshold, with a device-side count, and for both table geometries of config 4"""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops import functional as F
    from arcnerf_amd.utils.cfgs_utils import load_configs
    torch.manual_seed(0)
    m = build_model(load_configs(os.path.join(ROOT, 'configs', 'neus_ngp_multivol.yaml'), [])).to(gpu)
    g = torch.Generator().manual_seed(8)
    for emb in (m.fg_model.geo_net.embed_fn, m.bkg_model.geo_net.embed_fn):
        table = (torch.randn_like(emb.embeddings) * 0.1).contiguous()
        L, Fq = int(emb.desc.n_levels), int(emb.desc.n_feat)
        for n in (1, 1000, 50021):
            lo, hi = torch.tensor(list(emb.desc.min_xyz)), torch.tensor(list(emb.desc.max_xyz))
            x = (lo + (hi - lo) * (torch.rand(n, 3, generator=g) * 1.1 - 0.05)).to(gpu)       # 5 % beyond the box on every side
            enc0 = F.hashgrid_fwd_plain(x, table, emb.desc)
            enc, corners = F.hashgrid_fwd_corners(x, table, emb.desc)
            assert torch.equal(enc, enc0) and corners.shape == (L, 2 * Fq, n, 4)
            jac = torch.randn(n, enc.shape[1], generator=g).to(gpu)
            _, dx0 = F.hashgrid_bwd(x, table, jac, emb.desc, want_dtable=False, want_dxyz=True)
            assert torch.equal(F.hashgrid_dxyz_corners(x, corners, jac, emb.desc), dx0)
            gdx = torch.randn(n, 3, generator=g).to(gpu)
            dd0, _, _ = F.hashgrid_bwd_bwd(x, gdx, table, jac, emb.desc, want_ddout=True, want_dtable=False, want_d2xyz=False)
            assert torch.equal(F.hashgrid_ddout_corners(x, gdx, corners, emb.desc), dd0)
            assert float(dx0.abs().max()) > 0 and float(dd0.abs().max()) > 0
            # the quads hold the table's rows: rebuilt from the debug indices of the plain kernel
            _, idx = F.hashgrid_fwd(x, table, emb.desc, want_idx=True)
            rows = table.view(-1, Fq)[idx.clamp(min=0).long()] * (idx >= 0).unsqueeze(-1)          # (n, L, 8, F)
            assert torch.equal(corners.permute(2, 0, 1, 3).reshape(n, L, 8, Fq), rows)
            # a device-side count: the rows behind it stay untouched
            k = max(1, n // 3)
            n_dev = torch.tensor([k], dtype=torch.int32, device=gpu)
            enc_k, corners_k = F.hashgrid_fwd_corners(x, table, emb.desc, n_dev=n_dev)
            assert torch.equal(enc_k[:k], enc0[:k]) and torch.equal(corners_k[:, :, :k], corners[:, :, :k])
            assert torch.equal(F.hashgrid_dxyz_corners(x, corners, jac, emb.desc, n_dev=n_dev)[:k], dx0[:k])
            assert torch.equal(F.hashgrid_ddout_corners(x, gdx, corners, emb.desc, n_dev=n_dev)[:k], dd0[:k])


def test_fused_adam_step_in_two_halves(gpu):
    """FusedAdam.begin_step() / finish_step(exclude): the ranges a caller's kernels updated with the step's own numbers + the rest of the flat
    buffer = one step() (same bits, same counters, EMA aliased onto the parameter included); misaligned or overlapping ranges are refused"""
    from arcnerf_amd.ops import functional as F
    from arcnerf_amd.optim import FusedAdam

    def make():
        torch.manual_seed(1)
        ps = [torch.nn.Parameter(torch.randn(n, device=gpu)) for n in (1, 4096, 333, 20000, 64)]
        opt = FusedAdam(ps, lr=1e-2, eps=1e-15, weight_decay=1e-6, ema_decay=0.95, ema_in_param=True, zero_grad_on_step=True).flatten()
        return ps, opt

    (pa, oa), (pb, ob) = make(), make()
    g = torch.Generator().manual_seed(2)
    for it in range(3):
        grads = [torch.randn(p.numel(), generator=g).to(gpu) for p in pa]
        for ps in (pa, pb):
            for p, gr in zip(ps, grads):
                p.grad.copy_(gr.view_as(p))
        oa.step()
        h = ob.begin_step()
        fb = ob._flat[0]
        _, _, o1 = ob.table_views(pb[1])
        _, _, o3 = ob.table_views(pb[3])
        mine = [(o1 + 1024, o1 + 4096), (o3, o3 + 8000)]            # "the caller's kernels": the same Adam on two ranges, by hand
        F.adam_ema_step_runs(fb['params'], fb['grads'], fb['exp_avg'], fb['exp_avg_sq'], fb['params'], mine, h['step'], lr=h['lr'], betas=h['betas'],
                             eps=h['eps'], weight_decay=h['weight_decay'], ema_decay=h['ema_decay'], grad_scale=h['grad_scale'], ema_step=h['ema_step'],
                             zero_grad=True)
        ob.finish_step(h, mine)
        assert torch.equal(oa.flat_params(), ob.flat_params()) and torch.equal(oa._flat[0]['exp_avg_sq'], ob._flat[0]['exp_avg_sq'])
        assert oa._flat[0]['step'] == ob._flat[0]['step'] == it + 1 and float(ob.flat_grads().abs().max()) == 0
    h = ob.begin_step()
    with pytest.raises(RuntimeError, match='aligned'):
        ob.finish_step(h, [(6, 10)])
    with pytest.raises(RuntimeError, match='disjoint'):
        ob.finish_step(h, [(8, 64), (32, 128)])
