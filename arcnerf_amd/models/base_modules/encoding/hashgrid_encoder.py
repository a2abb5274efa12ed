"""HashGridEmbedder (arcnerf/models/base_modules/encoding/hashgrid_encoder.py:21-249): multiresolution hash grid with
the reference TORCH-backend semantics (always hashed, int64 hash, non power-of-two low levels, fp32) on one gather
kernel forward and one scatter kernel backward.  `backend: tcnn | torch` both select the HIP kernels; the parameter is
named `embeddings` (n_total_embed, F) like the reference's torch backend so state_dicts line up."""
import torch
import torch.nn as nn

from .... import _native as N
from ....geometry.volume import Volume
from ....ops.autograd import hashgrid_encode
from ....pipeline import hashgrid_level_table
from ....utils.registry import ENCODER_REGISTRY


@ENCODER_REGISTRY.register()
class HashGridEmbedder(nn.Module):
    def __init__(self, input_dim=3, n_levels=16, n_feat_per_entry=2, hashmap_size=19, base_res=16, max_res=2048,
                 origin=(0, 0, 0), side=None, xyz_len=None, dtype='torch.float16', include_input=True, backend=None,
                 *args, **kwargs):
        super().__init__()
        assert input_dim == 3, 'HashGridEmbedder should has input_dim==3...'
        assert side is not None or xyz_len is not None, 'You must set the size of volume...'
        backend = backend or 'torch'
        assert backend in ('torch', 'tcnn'), 'Invalid backend used, only torch/tcnn allowed'
        self.input_dim, self.include_input, self.backend = input_dim, include_input, backend
        self.n_levels, self.n_feat_per_entry, self.hashmap_size = n_levels, n_feat_per_entry, 2 ** hashmap_size
        self.base_res, self.max_res = base_res, max_res
        self.resolutions, self.offsets = hashgrid_level_table(n_levels, hashmap_size, base_res, max_res)
        self.n_total_embed = self.offsets[-1]
        self.embeddings = nn.Parameter(torch.empty(self.n_total_embed, n_feat_per_entry))
        nn.init.uniform_(self.embeddings, -1e-4, 1e-4)
        lens = [float(side)] * 3 if side is not None else [float(v) for v in xyz_len]
        mn = torch.tensor([float(origin[k]) - lens[k] / 2.0 for k in range(3)])
        mx = torch.tensor([float(origin[k]) + lens[k] / 2.0 for k in range(3)])
        self.register_buffer('min_xyz', mn)
        self.register_buffer('max_xyz', mx)
        # the reference's torch backend keeps its box as a Volume submodule (hashgrid_encoder.py:90): same keys in the state_dict
        self.volume = Volume(n_grid=base_res, origin=origin, side=side, xyz_len=xyz_len)
        self.desc = N.make_hashgrid_desc(self.resolutions, self.offsets, n_feat_per_entry, mn.tolist(), mx.tolist())
        self.out_dim = n_levels * n_feat_per_entry + include_input * input_dim
        self._ws = None

    def get_output_dim(self):
        return self.out_dim

    def get_embeddings(self):
        return self.embeddings.data

    def set_embeddings(self, data):
        self.embeddings.data = data

    def forward(self, xyz):
        assert xyz.dim() == 2 and xyz.shape[-1] == 3, 'Must be (B, 3) tensor'
        emb = hashgrid_encode(xyz, self.embeddings, self.desc, True)
        return torch.cat([xyz, emb], dim=-1) if self.include_input else emb
