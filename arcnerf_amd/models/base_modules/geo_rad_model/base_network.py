"""BaseGeoNet / BaseRadianceNet (arcnerf/models/base_modules/geo_rad_model/base_network.py:7-71)."""
import torch
import torch.nn as nn


class BaseGeoNet(nn.Module):
    """forward(x (B,C)) -> (geo value (B,1), feature (B,W_feat) | None)"""

    def forward_geo_value(self, x):
        return self.forward(x)[0][:, 0]

    def forward_with_grad(self, x):
        with torch.enable_grad():
            x = x.requires_grad_(True)
            geo, h = self.forward(x)
            grad = torch.autograd.grad(outputs=geo, inputs=x, grad_outputs=torch.ones_like(geo), create_graph=True,
                                       retain_graph=True, only_inputs=True)[0]
        return geo, h, grad

    def pretrain_siren(self, n_iter=5000, lr=1e-4, thres=0.01, n_pts=5000):
        return


class BaseRadianceNet(nn.Module):
    """forward(x, view_dirs, normals, geo_feat) -> rgb (B,3)"""
