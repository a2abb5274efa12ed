from .linear_network_module import GeoNet, RadianceNet
from .tcnn_fusedmlp_module import FusedMLPGeoNet, FusedMLPRadianceNet

__all__ = ['GeoNet', 'RadianceNet', 'FusedMLPGeoNet', 'FusedMLPRadianceNet']
