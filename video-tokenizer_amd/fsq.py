"""Finite scalar quantizer with the reference's class surface (models/model_new/quantizer/fsq.py:54-131).

The reference's TiTok-style autoencoders build `FSQ(levels=[8,8,8,5,5,5])` directly (model_new/autoencoder.py:59,640;
[8,8,8,8,5,5,5,5] at :140) and use `forward(z) -> (codes, {'indices': int32})` and `indices_to_codes(indices)`
(:654,661).  Here the same calls run as single libvt_hip launches (csrc/vt_fsq.hip); there is no parameter and no
state dict entry (the reference registers its buffers non-persistently, :63,66,75).  GPU tensors only.
"""
import torch
from torch import nn

from . import hip
from .functional import FiniteScalarQuantize


class FSQ(nn.Module):
    def __init__(self, levels, dim=None):
        super().__init__()
        self.levels = tuple(int(v) for v in levels)
        self.codebook_dim = len(self.levels)
        self.dim = dim if dim is not None else len(self.levels)
        size = 1
        for v in self.levels:
            size *= v
        self.codebook_size = size
        basis, b = [], 1
        for v in self.levels:
            basis.append(b)
            b *= v
        self.register_buffer("_levels", torch.tensor(self.levels, dtype=torch.int32), persistent=False)
        self.register_buffer("_basis", torch.tensor(basis, dtype=torch.int32), persistent=False)

    def quantize(self, z):
        """codes in [-1, 1] with the straight-through gradient (fsq.py:83-88)"""
        return FiniteScalarQuantize.apply(z, self.levels)[0]

    def codes_to_indices(self, zhat):
        """fsq.py:103-107: exact integer arithmetic on the level indices (codes are k / half_width)"""
        half = (self._levels // 2).to(zhat.device)
        lvl = torch.round(zhat.float() * half + half).to(torch.int32)
        return (lvl * self._basis.to(zhat.device)).sum(dim=-1).to(torch.int32)

    def indices_to_level_indices(self, indices):
        """fsq.py:109-113"""
        return (indices.unsqueeze(-1) // self._basis.to(indices.device)) % self._levels.to(indices.device)

    def indices_to_codes(self, indices):
        """fsq.py:115-118; always fp32 like the reference (callers cast, autoencoder.py:661)"""
        assert indices is not None
        return hip.fsq_indices_to_codes(indices, self.levels, torch.float32)

    def forward(self, z):
        """z [..., d] (fp32 or bf16) -> (codes like z, {'indices': int32 [...]})"""
        if z.dtype not in (torch.float32, torch.bfloat16):
            z32 = z.float()
            codes, idx = FiniteScalarQuantize.apply(z32, self.levels)
            return codes.to(z.dtype), {"indices": idx}
        codes, idx = FiniteScalarQuantize.apply(z, self.levels)
        return codes, {"indices": idx}

    def extra_repr(self):
        return f"levels={list(self.levels)}, codebook_size={self.codebook_size}"
