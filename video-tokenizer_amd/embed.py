"""Patch embedding parameter holder and sin-cos position tables.

Mirrors /root/reference/models/embed.py: `PatchEmbed3D` (:37-116; Conv3d kernel=stride=(pt,p,p),
flatten to B x N x D) and the float64 numpy builders get_3d/2d/1d_sincos_pos_embed (:269-331).
The tables are module buffers that live in checkpoints, so they are rebuilt with the same
float64 arithmetic (tests pin them bit-for-bit against the reference's outputs).  The Conv3d
itself never runs: the engine gathers patches and multiplies by `proj.weight.view(D,-1)` on MFMA.
"""
import numpy as np
import torch.nn as nn


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos, scale_factor=10000):
    assert embed_dim % 2 == 0
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / scale_factor ** omega
    out = np.einsum("m,d->md", np.asarray(pos).reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size):
    assert embed_dim % 2 == 0
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)  # w first
    return np.concatenate([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0]),
                           get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])], axis=1)


def get_3d_sincos_pos_embed(embed_dim, grid_size, frame_num):
    e2 = get_2d_sincos_pos_embed(embed_dim, grid_size).reshape(1, grid_size, grid_size, embed_dim)
    e1 = get_1d_sincos_pos_embed_from_grid(embed_dim, np.arange(frame_num, dtype=np.float32)).reshape(frame_num, 1, 1, embed_dim)
    return (e2 + e1).reshape(-1, embed_dim)


class PatchEmbed3D(nn.Module):
    """Same constructor, attributes and state-dict keys (`proj.weight [D,C,pt,p,p]`, `proj.bias`) as the
    reference class; `strict_vid_size` is read/written by the evaluator (eval/rfvd_evaluator.py:33)."""

    def __init__(self, spatial_vid_size=224, temporal_vid_size=8, spatial_patch_size=16, temporal_patch_size=4,
                 in_chans=3, embed_dim=768, norm_layer=None, flatten=True, output_fmt=None, bias=True,
                 strict_vid_size=True, dynamic_vid_pad=False):
        super().__init__()
        if norm_layer is not None or not flatten or output_fmt is not None or dynamic_vid_pad or not bias:
            raise NotImplementedError("PatchEmbed3D: only the configuration LARPTokenizer uses is built (flatten, bias, no norm/pad)")
        self.patch_size = (temporal_patch_size, spatial_patch_size, spatial_patch_size)
        self.vid_size = (temporal_vid_size, spatial_vid_size, spatial_vid_size)
        self.grid_size = tuple(s // p for s, p in zip(self.vid_size, self.patch_size))
        self.num_patches = self.grid_size[0] * self.grid_size[1] * self.grid_size[2]
        self.num_spatial_patches = self.num_patches_per_frame = self.grid_size[1] * self.grid_size[2]
        self.num_temporal_patches = self.grid_size[0]
        self.flatten = True
        self.output_fmt = "bcthw"
        self.strict_vid_size = strict_vid_size
        self.dynamic_vid_pad = False
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size, bias=True)

    def check_input(self, x):
        """The reference's size asserts (embed.py:87-104)."""
        _, _, T, H, W = x.shape
        if self.strict_vid_size:
            assert T == self.vid_size[0], f"Input depth ({T}) doesn't match model ({self.vid_size[0]})."
            assert H == self.vid_size[1], f"Input height ({H}) doesn't match model ({self.vid_size[1]})."
            assert W == self.vid_size[2], f"Input width ({W}) doesn't match model ({self.vid_size[2]})."
        else:
            assert T % self.patch_size[0] == 0, f"Input depth ({T}) should be divisible by patch size ({self.patch_size[0]})."
            assert H % self.patch_size[1] == 0, f"Input height ({H}) should be divisible by patch size ({self.patch_size[1]})."
            assert W % self.patch_size[2] == 0, f"Input width ({W}) should be divisible by patch size ({self.patch_size[2]})."

    def forward(self, x, pos_embed=None):
        """embed.py:85-116 -> [B, N, D] fp32 (values bf16-rounded like the conv under autocast).  `pos_embed` (an
        extension): an fp32 [N, D] table added in the GEMM epilogue, as every caller adds one right after."""
        from .functional import PatchEmbed
        self.check_input(x)
        return PatchEmbed.apply(x, self.proj.weight, self.proj.bias, pos_embed)


class VideoPatchEmbed(nn.Module):
    """models/embed.py:16-34 (temporal_patch_size == 1): timm's 2-D PatchEmbed applied to every frame -- Conv2d(kernel = stride = p), tokens
    ordered (t, h, w).  Same state-dict keys (`proj.weight [D, C, p, p]`, `proj.bias`); the arithmetic is PatchEmbed3D's gather + GEMM with
    a temporal patch of one frame."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True, frame_num=None):
        super().__init__()
        assert frame_num is not None and bias
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.num_patches_per_frame = self.num_spatial_patches = self.grid_size[0] * self.grid_size[1]
        self.num_temporal_patches = frame_num
        self.flatten = True
        self.strict_vid_size = True
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=True)

    def forward(self, x, pos_embed=None):
        from .functional import PatchEmbed
        _, _, T, H, W = x.shape
        assert (H, W) == self.img_size, f"Input size ({H}x{W}) doesn't match model ({self.img_size[0]}x{self.img_size[1]})."
        return PatchEmbed.apply(x, self.proj.weight.unsqueeze(2), self.proj.bias, pos_embed)
