"""Near-tie probe for the inverse-CDF resampling of the reference (build container only; imported by the generators).

`sample_pdf` / `sample_cdf` (arcnerf/render/ray_helper.py:410-473) are discontinuous in their inputs in three places:
 (1) the bin a uniform `u` falls into (`searchsorted(cdf, u, right=True)`) when the neighbouring bin uses the other form of (2);
 (2) the `denom[denom < eps] = 1` rule: a bin whose cdf mass is an ulp either side of 1e-5 switches between `t = (u - c0) / denom`
     and `t = u - c0`;
 (3) with `det=True` the lattice u = linspace(0, 1, n) contains u = 1.0 EXACTLY and it is compared with cdf[-1], a float cumsum that
     is 1.0 to within an ulp: `cdf[-1] > 1.0` sends the last sample into the last bin - where (2) usually applies, an empty tail bin
     has mass 1e-5 / sum - i.e. to ~bins[-2] instead of bins[-1].  torch's CPU cumsum accumulates in double and rounds every prefix to
     float, so cdf[-1] > 1 iff the double sum of the float pdf values exceeds 1 + 2^-24; that sum is 1 + (rounding error of
     `weights.sum()`) + (rounding noise of the divisions), a few 1e-8 either way.
Different arithmetic (the reference's own CUDA path, this repo's kernels) takes the other side of such a decision now and then; the
sample moves by a bin, and the gradients of the matrices fed by the 2^9-frequency position embedding feel a single moved sample at the
1e-2 level.  A parity fixture must therefore hold only rays whose decisions have a margin.  `Probe` wraps the reference's
`sample_pdf`, recomputes its decisions and records per ray:
   last[r]   = (1 + 2^-24) - sum_double(pdf[r])    (> 0: cdf[-1] <= 1.0; the decision's margin in cdf units, a few 1e-8)
   inner[r]  = smallest distance of any u (except the u = 1 / u = 0 ends) to a cdf edge that borders a bin on the `denom < eps` rule,
               and of any used denominator to eps where the two forms give positions more than 1e-6 apart
"""
import numpy as np
import torch


def decision_margin(bins, cdf, u, eps=1e-5):
    """per ray: the smallest margin (cdf units) of decisions (1) and (2) for the uniforms `u` (R, n): distance of any u to a cdf edge that
    borders a bin on (or within 5 % of) the `denom < eps` rule or to the clamped last edge; 0 for a u INSIDE a bin whose mass is within
    5 % of eps when the two forms of the rule give positions more than 1e-6 apart (near cdf = 1 such masses are quantised to 167 or 168
    ulps, either side of eps, so the rule flips with the last bit of the cdf)."""
    n_pts = cdf.shape[-1]
    inds = torch.searchsorted(cdf, u, right=True)
    below, above = torch.clamp(inds - 1, 0, n_pts - 1), torch.clamp(inds, 0, n_pts - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = c1 - c0
    dc = cdf[:, 1:] - cdf[:, :-1]
    irregular = dc < eps * 1.05
    edge_bad = torch.zeros_like(cdf, dtype=torch.bool)
    edge_bad[:, :-1] |= irregular
    edge_bad[:, 1:] |= irregular
    edge_bad[:, -1] = True
    d_edge = (u[:, :, None] - cdf[:, None, :]).abs()
    d_edge = torch.where(edge_bad[:, None, :].expand_as(d_edge), d_edge, torch.full_like(d_edge, 1.0))
    m_edge = d_edge.amin(dim=(1, 2))
    differs = ((u - c0) * (1.0 / denom.clamp_min(1e-12) - 1.0)).abs() * (b1 - b0).abs() > 1e-6
    near_eps = (denom - eps).abs() < 0.05 * eps
    m_den = torch.where(differs & near_eps & (above != below), torch.zeros_like(denom), torch.full_like(denom, 1.0)).amin(dim=1)
    return torch.minimum(m_edge, m_den)


CDF_NOISE = 3e-7   # what two fp32 evaluations of the same cdf differ by near cdf = 1: the normaliser's last ulps + a prefix's rounding


def position_noise(bins, cdf, u, eps=1e-5):
    """per ray: how far the WORST-conditioned sample of the ray moves when the cdf moves by CDF_NOISE.  The inverse CDF divides by the
    bin's mass, `t = (u - c0) / (c1 - c0)`: a u that lands in a bin of mass 1e-4 turns 3e-7 of cdf noise into 3e-3 of a bin width.
    This is the reference's own conditioning (its CPU and CUDA paths disagree by as much), so a parity fixture holds rays whose every
    uniform lands in a bin massive enough - or in an empty one, where the `denom < eps` rule makes t ~ 0."""
    n_pts = cdf.shape[-1]
    inds = torch.searchsorted(cdf, u, right=True)
    below, above = torch.clamp(inds - 1, 0, n_pts - 1), torch.clamp(inds, 0, n_pts - 1)
    denom = torch.gather(cdf, 1, above) - torch.gather(cdf, 1, below)
    width = (torch.gather(bins, 1, above) - torch.gather(bins, 1, below)).abs()
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    return (width / denom).amax(dim=1) * CDF_NOISE


class Probe:
    def __init__(self, *modules):
        """modules: every module object that holds a reference to `sample_pdf` (ray_helper itself and the models that imported it)"""
        self.modules = modules
        self.orig = modules[0].sample_pdf
        self.last, self.inner, self.last_gt1, self.noise = [], [], [], []

    def __enter__(self):
        probe = self

        def wrapped(bins, weights, n_sample, det=False, eps=1e-5):
            assert det, 'the probe assumes the deterministic lattice (perturb off)'
            w = weights.detach() + eps
            pdf = w / torch.sum(w, -1, keepdim=True)
            cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
            total = pdf.double().sum(-1)
            probe.last.append(((1.0 + 2.0 ** -24) - total).numpy())
            probe.last_gt1.append((cdf[:, -1] > 1.0).numpy())
            u = torch.linspace(0.0, 1.0, steps=n_sample)[None, 1:-1].expand(cdf.shape[0], n_sample - 2).contiguous()
            probe.inner.append(decision_margin(bins.detach(), cdf, u, eps).numpy())
            probe.noise.append(position_noise(bins.detach(), cdf, u, eps).numpy())
            return probe.orig(bins, weights, n_sample, det, eps)
        for m in self.modules:
            m.sample_pdf = wrapped
        return self

    def __exit__(self, *a):
        for m in self.modules:
            m.sample_pdf = self.orig

    def per_ray(self):
        """(last margin, inner margin, any cdf[-1] > 1) per ray: the minimum over the calls (all calls must see the same rays)"""
        return (np.min(np.stack(self.last), 0), np.min(np.stack(self.inner), 0), np.any(np.stack(self.last_gt1), 0))

    def per_ray_noise(self):
        """per ray: the largest position_noise over the calls"""
        return np.max(np.stack(self.noise), 0)


class RandTape:
    """torch.rand of a reference run on tape.  record(): every call draws from a private generator and is appended to `draws`;
    replay(draws): the calls return the given tensors in order (shape-checked).  The reference's sampling helpers take their uniforms
    from torch.rand (ray_helper.py:375 perturb_interval, :453 sample_cdf); the mirror takes them from render/ray_helper.uniform, which the
    tests feed with the same tape (tests/rand_feed.py)."""

    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.draws = []
        self._real = torch.rand
        self._feed = None

    def _shape(self, a, k):
        size = k.get('size', a[0] if len(a) == 1 and not isinstance(a[0], int) else a)
        return tuple(int(v) for v in size)

    def __call__(self, *a, **k):
        shape = self._shape(a, k)
        if self._feed is None:
            t = self._real(shape, generator=self.gen, dtype=torch.float32)
            self.draws.append(t.clone())
            return t
        t = self._feed.pop(0)
        assert tuple(t.shape) == shape, (tuple(t.shape), shape)
        self.draws.append(t.clone())
        return t.clone()

    def record(self):
        self._feed, self.draws = None, []
        return self

    def replay(self, draws):
        self._feed, self.draws = [torch.as_tensor(d) for d in draws], []
        return self

    def __enter__(self):
        torch.rand = self
        return self

    def __exit__(self, *a):
        torch.rand = self._real
        assert not self._feed, 'replay tape not consumed: {} draws left'.format(len(self._feed))


class ProbeU(Probe):
    """Probe for perturb=True runs: the uniforms are the ones the RandTape handed out (no u = 1, no lattice)."""

    def __init__(self, tape, *modules):
        super().__init__(*modules)
        self.tape = tape

    def __enter__(self):
        probe = self

        def wrapped(bins, weights, n_sample, det=False, eps=1e-5):
            assert not det
            out = probe.orig(bins, weights, n_sample, det, eps)
            u = probe.tape.draws[-1]
            assert u.shape == (bins.shape[0], n_sample)
            w = weights.detach() + eps
            pdf = w / torch.sum(w, -1, keepdim=True)
            cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
            probe.inner.append(decision_margin(bins.detach(), cdf, u, eps).numpy())
            probe.noise.append(position_noise(bins.detach(), cdf, u, eps).numpy())
            probe.last.append(np.ones(bins.shape[0]))
            probe.last_gt1.append(np.zeros(bins.shape[0], bool))
            return out
        for m in self.modules:
            m.sample_pdf = wrapped
        return self


class FlipLast:
    """Forces decision (3) of the module docstring - u = 1.0 against cdf[-1] on the deterministic lattice - one way for every ray:
    mode 'down' sets cdf[:, -1] = 1.0 (the last sample is bins[-1]), mode 'up' sets it to the next float above 1.0 (the last sample
    falls into the last bin).  Running the reference's inference pass both ways and comparing the outputs per ray tells which rays do
    not depend on that coin flip (`inference_flip_sensitivity`); only those go into a fixture."""

    def __init__(self, mode, *modules):
        self.mode, self.modules = mode, modules
        self.orig = modules[0].sample_cdf

    def __enter__(self):
        flip = self

        def wrapped(bins, cdf, n_sample, det=False, eps=1e-5):
            assert det
            cdf = cdf.clone()
            cdf[:, -1] = 1.0 if flip.mode == 'down' else float(np.nextafter(np.float32(1.0), np.float32(2.0)))
            return flip.orig(bins, cdf, n_sample, det, eps)
        for m in self.modules:
            m.sample_cdf = wrapped
        return self

    def __exit__(self, *a):
        for m in self.modules:
            m.sample_cdf = self.orig


def inference_lattice_margin(model, inputs, *modules):
    """per HIT ray: (decision margin, position noise) of the inner lattice points u = k / (n - 1), 0 < k < n - 1, over the up-sampling rounds of an inference pass"""
    with Probe(*modules) as p:
        model({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    return p.per_ray()[1], p.per_ray_noise()


def inference_flip_sensitivity(model, inputs, ray_helper_module, keys=('rgb', 'depth', 'mask', 'normal')):
    """per ray: the largest change of any inference output between the two forced outcomes of the u = 1.0 decision"""
    res = {}
    for mode in ('down', 'up'):
        with FlipLast(mode, ray_helper_module):
            with torch.no_grad() if False else torch.enable_grad():
                out = model({k: v.clone() for k, v in inputs.items()}, inference_only=True)
        res[mode] = {k: out[k].detach() for k in keys if k in out}
    sens = None
    for k in res['down']:
        d = (res['down'][k] - res['up'][k]).abs()
        d = d.reshape(d.shape[0] * d.shape[1], -1).amax(dim=1)
        sens = d if sens is None else torch.maximum(sens, d)
    return sens.numpy()


def float64_gradients(model, inputs, draws, loss_fn, **run_kw):
    """The training pass of a reference model re-run in FLOAT64 (parameters, inputs and the taped uniforms cast up; same decisions as
    long as none sits on a tie, which the ray selection guarantees) -> {parameter name: gradient as float64 numpy}.  What it is for:
    `|grad_fp32 - grad_fp64| / max|grad_fp64|` is the reference's OWN fp32 error on that tensor.  For NeuS it is ~1e-2 on the two sdf-net
    matrices fed by the 2^9-frequency position embedding (the Eikonal term differentiates sin(512 x) twice and the up-sampled positions
    carry the sdf's rounding error times the sharpness 64..512) and <= 3e-4 everywhere else; the parity tests hold every gradient at
    max(1e-3, 1.25 x that error) and additionally require the mirror to be as close to the float64 gradient as that.  The model is
    returned to float32."""
    flipped = []
    for mod in model.modules():
        if getattr(mod, 'dtype', None) is torch.float32:      # encoders that cast their output to a configured dtype
            mod.dtype = torch.float64
            flipped.append(mod)
    model.double()
    try:
        tape = RandTape(0)
        with tape.replay([torch.as_tensor(d).double() for d in draws]):
            res = model({k: v.clone().double() for k, v in inputs.items()}, inference_only=False, **run_kw)
        loss = loss_fn(res, {k: v.double() for k, v in inputs.items()})
        loss = loss[0] if isinstance(loss, tuple) else loss
        model.zero_grad()
        loss.backward()
        assert res['rgb'].dtype == torch.float64
        grads = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
        outs = {k: v.detach().numpy().copy() for k, v in res.items() if torch.is_tensor(v)}
    finally:
        model.float()
        model.zero_grad()
        for mod in flipped:
            mod.dtype = torch.float32
    return grads, outs, float(loss)


def store_fp32_error(out, tag, grads32, grads64, outs64, res32):
    """`f64err.<param>` = max|g32 - g64| / max|g64| (grads32: {name: fp32 gradient tensor} taken before the float64 re-run);
    `f64out.<key>` = max|out32 - out64|; `g64sum.<param>.*` = seeded_weights-style summary of the float64 gradient"""
    import seeded_weights as SW
    worst = {}
    for k, g32 in grads32.items():
        g64 = grads64[k]
        err = float(np.abs(g32.detach().double().numpy() - g64).max() / (np.abs(g64).max() + 1e-300))
        out[tag + 'f64err.' + k] = np.array(err)
        worst[k] = err
        if err > 5e-4:      # the tensors whose bar is set by this error: the float64 gradient travels too (rows 0-3 and every 16th)
            for kk, vv in SW.grad_summary(g64.astype(np.float32)).items():
                if kk in ('head', 'mod16', 'max'):
                    out[tag + 'g64sum.' + k + '.' + kk] = vv
    for k, v in res32.items():
        if torch.is_tensor(v) and k in outs64:
            out[tag + 'f64out.' + k] = np.array(float(np.abs(v.detach().double().numpy() - outs64[k]).max()))
    return worst
