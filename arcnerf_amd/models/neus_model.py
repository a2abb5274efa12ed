"""NeuS per-interval math (arcnerf/models/neus_model.py:221-265): sdf_to_cdf, sdf_to_pdf, sdf_to_alpha.

Second piece of the NeuS row (the first is the sphere bound).  sdf_to_alpha runs as one HIP
kernel forward and one backward (gradients to the mid sdf, the slope and the learnable scale).  The `Neus` model class
itself needs the geometry net's input gradient (normals) with a second-order backward for the Eikonal term and is not built
yet: `build_model` raises NotImplementedError for `type: NeuS`."""
import torch

from ..ops.autograd import SdfToAlphaFn


def sdf_to_cdf(sdf, s):
    """sigmoid(sdf * s)  (neus_model.py:221-228)"""
    return torch.sigmoid(sdf * s)


def sdf_to_pdf(sdf, s):
    """s e^{-s sdf} / (1 + e^{-s sdf})^2  (neus_model.py:231-239)"""
    esx = torch.exp(-sdf * s)
    return s * esx / ((1 + esx) ** 2)


def sdf_to_alpha(mid_sdf, zvals, mid_slope, s, clip=True):
    """mid_sdf, mid_slope (B, N_pts-1), zvals (B, N_pts), s float or scalar tensor -> alpha (B, N_pts-1)
    (neus_model.py:242-265); differentiable w.r.t. mid_sdf, mid_slope and a tensor s."""
    return SdfToAlphaFn.apply(mid_sdf.contiguous(), zvals.contiguous(), mid_slope.contiguous(), s, clip)
