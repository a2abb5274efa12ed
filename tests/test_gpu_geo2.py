"""arcn_geo2_fwd / arcn_geo2_bwd - the two-layer geometry nets of config 4 (NeuS on the hash grid + MultiVol background) as one forward and one
backward kernel - against float64 torch autograd of the reference's own expressions: GeoNet.forward_with_grad on a softplus DenseLayer
(sdf_model.py:42-101, base_network.py:30-44: the Jacobian row of the first output differentiated again) and the density net with TruncExp
(linear_network_module.py:174-197, arcnerf/ops/trunc_exp.py).  Bars: 1e-4 of each tensor's max for the forward, 2e-4 for the gradients (f32
MFMA chains against float64)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _lm(rows):
    n = rows.shape[0]
    return rows.view(n, 16, 2).permute(1, 0, 2).contiguous().reshape(-1)


def _rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('n,beta,n_out', [(5003, 100.0, 17), (128, 5.0, 17), (40000, 100.0, 16), (777, 20.0, 5)])
def test_sdf_net_with_its_jacobian_row(gpu, n, beta, n_out):
    from arcnerf_amd.ops import functional as F
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(n, 32, generator=g) * 0.5).to(gpu)
    w1 = (torch.randn(64, 32, generator=g) * 0.05).to(gpu)
    w2 = (torch.randn(n_out, 64, generator=g) * 0.3).to(gpu)
    d0 = torch.randn(n, generator=g).to(gpu)
    wide = torch.randn(n, n_out + 6, generator=g).to(gpu)          # d_feat as a column slice of a wider tensor
    dfeat = wide[:, 3:3 + n_out - 1]
    dj = torch.randn(n, 32, generator=g).to(gpu)
    lm = _lm(x)
    out, sdf, jac = F.geo2_fwd(lm, n, w1, w2, True, beta)
    n_pad = (n_out + 3) // 4 * 4
    assert out.shape == (n, n_pad) and jac.shape == (n, 32)
    # float64 reference
    xd, w1d, w2d = x.double().requires_grad_(), w1.double().requires_grad_(), w2.double().requires_grad_()
    h = torch.nn.functional.softplus(xd @ w1d.t(), beta=beta)
    o = h @ w2d.t()
    jr, = torch.autograd.grad(o[:, 0].sum(), xd, create_graph=True)
    assert _rel(out[:, :n_out], o.detach()) <= 1e-4 and float(out[:, n_out:].abs().max() if n_pad > n_out else 0.0) == 0.0
    assert torch.equal(sdf, out[:, 0])
    assert _rel(jac, jr.detach()) <= 1e-4
    loss = (o[:, 0] * d0.double()).sum() + (o[:, 1:] * dfeat.double()).sum() + (jr * dj.double()).sum()
    gx, gw1, gw2 = torch.autograd.grad(loss, [xd, w1d, w2d])
    flat = torch.full((64 * 32 + n_out * 64 + 8,), 0.25, device=gpu)       # the gradients are ADDED into views of one flat buffer
    dw1, dw2 = flat[:2048].view(64, 32), flat[2052:2052 + n_out * 64].view(n_out, 64)
    dx = F.geo2_bwd(lm, n, w1, w2, True, beta, d0, dfeat, dw1, dw2, d_jac=dj)
    assert _rel(dx, gx) <= 2e-4, _rel(dx, gx)
    assert _rel(dw1 - 0.25, gw1) <= 2e-4, _rel(dw1 - 0.25, gw1)
    assert _rel(dw2 - 0.25, gw2) <= 2e-4, _rel(dw2 - 0.25, gw2)
    assert float((flat[2048:2052] - 0.25).abs().max()) == 0 and float((flat[2052 + n_out * 64:] - 0.25).abs().max()) == 0
    # level-major dx = the rows transposed, bit for bit
    flat2 = torch.zeros_like(flat)
    dx_lm = F.geo2_bwd(lm, n, w1, w2, True, beta, d0, dfeat, flat2[:2048].view(64, 32), flat2[2052:2052 + n_out * 64].view(n_out, 64), d_jac=dj, dx_level_major=True)
    assert torch.equal(dx_lm, _lm(dx))


@pytest.mark.parametrize('n,n_out', [(6001, 17), (64, 17), (33333, 9)])
def test_density_net_with_truncexp_head(gpu, n, n_out):
    from arcnerf_amd.ops import functional as F
    g = torch.Generator().manual_seed(n + 1)
    x = (torch.randn(n, 32, generator=g) * 0.5).to(gpu)
    w1 = (torch.randn(64, 32, generator=g) * 0.3).to(gpu)
    w2 = (torch.randn(n_out, 64, generator=g) * 0.2).to(gpu)
    w2[0] *= 8.0                                                    # some pre-activations beyond the +-15 clamp of the TruncExp backward
    d0 = torch.randn(n, generator=g).to(gpu)
    dfeat = torch.randn(n, n_out - 1, generator=g).to(gpu)
    lm = _lm(x)
    out, sigma, none = F.geo2_fwd(lm, n, w1, w2, False)
    assert none is None
    xd, w1d, w2d = x.double(), w1.double(), w2.double()
    h = torch.relu(xd @ w1d.t())
    o = h @ w2d.t()
    assert _rel(out[:, :n_out], o) <= 1e-4
    assert torch.allclose(sigma, torch.exp(out[:, 0]), rtol=2e-6, atol=0)
    assert n < 1000 or float(out[:, 0].abs().max()) > 15.0
    go = torch.cat([(d0.double() * torch.exp(out[:, 0].double().clamp(-15, 15)))[:, None], dfeat.double()], dim=1)
    gw2 = go.t() @ h
    dz = (go @ w2d) * (h > 0)
    gw1 = dz.t() @ xd
    gx = dz @ w1d
    flat = torch.zeros(64 * 32 + n_out * 64, device=gpu)
    dw1, dw2 = flat[:2048].view(64, 32), flat[2048:].view(n_out, 64)
    dx = F.geo2_bwd(lm, n, w1, w2, False, 1.0, d0, dfeat, dw1, dw2, out=out, dx_level_major=True)
    # (a sample whose float32 hidden pre-activation has the other sign than the float64 one flips a ReLU gate: bars on the sums, looser per row)
    assert _rel(dw1, gw1) <= 2e-4 and _rel(dw2, gw2) <= 2e-4, (_rel(dw1, gw1), _rel(dw2, gw2))
    assert _rel(dx, _lm(gx.float()).double()) <= 1e-3


def test_geo2_argument_checks(gpu):
    from arcnerf_amd.ops import functional as F
    x = torch.zeros(32 * 16, device=gpu)
    w1, w2 = torch.zeros(64, 32, device=gpu), torch.zeros(40, 64, device=gpu)
    with pytest.raises(RuntimeError):
        F.geo2_fwd(x, 16, w1, w2, True, 1.0)          # more than 32 outputs


@pytest.mark.parametrize('S,feat_first,rgb_w', [(7001, True, 3), (128, False, 3), (50000, True, 16)])
def test_both_ngp_nets_in_one_kernel_equal_the_two_launches_bit_for_bit(gpu, S, feat_first, rgb_w):
    """arcn_ngp_nets_fwd = arcn_mlp_fwd_lm (geometry net) + arcn_mlp_fwd_cat (radiance net on [geo_out | SH(ray)]): the same fragments in the
    same MFMA order, the geometry net's output tile handed over in registers - geo_out, sigma, the saved activations and rgb must be the SAME
    BITS (base_3d_model.py:233-254 on the fused MLPs of tcnn_fusedmlp_module.py:66-77)."""
    import ctypes as C
    from arcnerf_amd import _native as N
    from arcnerf_amd.ops import functional as F
    g = torch.Generator().manual_seed(S)
    R = 97
    cap = S + 13                                      # capacity above the count: the level stride and the activation layout follow the capacity
    lm = (torch.randn(16 * cap * 2, generator=g) * 0.5).to(gpu)
    ray_id = torch.sort(torch.randint(0, R, (cap,), generator=g)).values.int().to(gpu)
    sh_ray = torch.randn(R, 16, generator=g).to(gpu)
    gd, rd = N.make_mlp_desc([32, 64, 16], 'relu', None), N.make_mlp_desc([32, 64, 64, rgb_w], 'relu', 'sigmoid')
    gw = (torch.randn(32 * 64 + 64 * 16, generator=g) * 0.2).to(gpu)
    rw = (torch.randn(32 * 64 + 64 * 64 + 64 * rgb_w, generator=g) * 0.2).to(gpu)
    n_dev = torch.tensor([S], dtype=torch.int32, device=gpu)
    lib, st = N.lib(), N.stream()

    def bufs():
        return (torch.full((cap, 16), 7.0, device=gpu), torch.full((cap, rgb_w), 7.0, device=gpu), torch.full((F.mlp_acts_floats(rd, cap),), 7.0, device=gpu),
                torch.full((cap,), 7.0, device=gpu))
    geo_a, rgb_a, acts_a, sig_a = bufs()
    N.check(lib.arcn_mlp_fwd_lm(N.ptr(lm), cap, N.ptr(gw), C.addressof(gd), N.ptr(geo_a), None, cap, cap, N.ptr(n_dev), st))
    N.check(lib.arcn_mlp_fwd_cat(N.ptr(geo_a), N.ptr(sh_ray), N.ptr(ray_id), int(feat_first), N.ptr(rw), C.addressof(rd), N.ptr(rgb_a), N.ptr(acts_a),
                                 N.ptr(sig_a), N.ACT['truncexp'], cap, cap, N.ptr(n_dev), st))
    geo_b, rgb_b, acts_b, sig_b = bufs()
    N.check(lib.arcn_ngp_nets_fwd(N.ptr(lm), cap, N.ptr(gw), C.addressof(gd), N.ptr(geo_b), N.ptr(sh_ray), N.ptr(ray_id), int(feat_first), N.ptr(rw),
                                  C.addressof(rd), N.ptr(rgb_b), N.ptr(acts_b), N.ptr(sig_b), N.ACT['truncexp'], cap, cap, N.ptr(n_dev), st))
    assert torch.equal(geo_a, geo_b) and torch.equal(sig_a, sig_b) and torch.equal(rgb_a, rgb_b) and torch.equal(acts_a, acts_b)
    assert float(geo_b[S:].min()) == 7.0 and float(rgb_b[S:].min()) == 7.0          # rows behind the device-side count are left alone
    # inference: no saved activations; other nets are refused
    rgb_c = torch.zeros_like(rgb_b)
    N.check(lib.arcn_ngp_nets_fwd(N.ptr(lm), cap, N.ptr(gw), C.addressof(gd), N.ptr(geo_b), N.ptr(sh_ray), N.ptr(ray_id), int(feat_first), N.ptr(rw),
                                  C.addressof(rd), N.ptr(rgb_c), None, None, 0, cap, S, None, st))
    assert torch.equal(rgb_c[:S], rgb_a[:S])
    bad = N.make_mlp_desc([32, 64, 16], 'softplus', None)
    assert lib.arcn_ngp_nets_fwd(N.ptr(lm), cap, N.ptr(gw), C.addressof(bad), N.ptr(geo_b), N.ptr(sh_ray), N.ptr(ray_id), 1, N.ptr(rw), C.addressof(rd),
                                 N.ptr(rgb_c), None, None, 0, cap, S, None, st) == -1


def test_module_path_sdf_node_on_the_fused_kernels_equals_its_chain_of_products(gpu):
    """ops.autograd.SdfMlpJacFn (what GeoNet.forward_with_grad runs for NeuS on the hash grid, sdf_model.py:42-101) routes the NGP shape through
    arcn_geo2_fwd / _bwd on row-major features; FUSED_SDF_NET = False keeps the chain of dense products it replaces: same outputs and gradients
    (the fused kernels take softplus and its slope from one hardware exponential: 2e-6 of each tensor's max, not bits)."""
    from arcnerf_amd.ops import autograd as A
    g = torch.Generator().manual_seed(11)
    S, O, beta = 4099, 20, 100.0
    f0 = (torch.randn(S, 32, generator=g) * 0.3).to(gpu)
    w10 = (torch.randn(64, 32, generator=g) / 32 ** 0.5 * 0.3).to(gpu)
    w20 = (torch.randn(O, 64, generator=g) / 8.0).to(gpu)
    go, gj = torch.randn(S, O, generator=g).to(gpu), torch.randn(S, 32, generator=g).to(gpu)
    res = {}
    for fused in (True, False):
        A.FUSED_SDF_NET = fused
        try:
            f, w1, w2 = (t.clone().requires_grad_(True) for t in (f0, w10, w20))
            out, jac = A.SdfMlpJacFn.apply(f, w1, w2, beta)
            loss = (out * go).sum() + (jac * gj).sum()
            res[fused] = (out.detach(), jac.detach()) + torch.autograd.grad(loss, (f, w1, w2))
        finally:
            A.FUSED_SDF_NET = True
    for a, b, name in zip(res[True], res[False], ('out', 'jac', 'df', 'dw1', 'dw2')):
        assert a.shape == b.shape, name
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), (name, float((a - b).abs().max() / b.abs().max()))


@pytest.mark.parametrize('n', [0, 1, 5, 4095, 4096, 4097, 32768, 100003])
def test_exclusive_scan_tiles(gpu, n):
    """arcn_exclusive_scan_i32 (the packed samplers' offsets: the reference's boolean-mask compaction order, fg_model.py:289-292) on sizes around
    its 4096-count tiles: offsets, the total in offsets[n], the clamp to a capacity and the largest count, against torch.cumsum"""
    from arcnerf_amd import _native as N
    g = torch.Generator().manual_seed(n)
    counts = torch.randint(0, 70, (max(n, 1),), generator=g, dtype=torch.int32)[:n].to(gpu)
    ref = torch.zeros(n + 1, dtype=torch.int64)
    ref[1:] = torch.cumsum(counts.cpu().long(), 0)
    for cap in (0, int(ref[-1]) // 2 + 1):
        off = torch.full((n + 1,), -7, dtype=torch.int32, device=gpu)
        mx = torch.full((1,), -7, dtype=torch.int32, device=gpu)
        N.check(N.lib().arcn_exclusive_scan_i32(N.ptr(counts) if n else N.ptr(off), N.ptr(off), n, cap, N.ptr(mx), N.stream()))
        want = ref.clamp(max=cap) if cap > 0 else ref
        assert torch.equal(off.cpu().long(), want), (n, cap)
        assert int(mx) == (int(counts.max()) if n else 0)
