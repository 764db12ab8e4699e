"""Transformer stacks: parameter holders with timm `Block` attribute names.

Mirrors /root/reference/models/transformer.py: `transformer_encoder_parallel` (:34-70) builds `depth`
x timm Block(dim, heads, mlp_ratio=4, qkv_bias=False); forward(context, query) = cat -> blocks ->
last len(query) rows.  The state-dict keys (`blocks.{i}.norm1.weight`, `attn.qkv.weight`,
`attn.proj.{weight,bias}`, `norm2.*`, `mlp.fc1.*`, `mlp.fc2.*`) are the checkpoint contract.  Inside
LARPTokenizer the arithmetic runs in the fused HIP engine (vt_tokenizer_encode/decode); called on their own
(`models.make({'name': 'transformer_encoder_parallel', ...})(context, query)`, or the discriminator's
`transformer_encoder_fused`, :9-31) the stacks run the same kernels through functional.BlockStack.
"""
import torch
import torch.nn as nn

from .registry import register


class _Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Block(nn.Module):
    """timm.models.vision_transformer.Block(dim, num_heads, mlp_ratio=4, qkv_bias=False) parameter layout."""

    def __init__(self, dim, num_heads, mlp_ratio=4, qkv_bias=False, proj_drop=0.0, attn_drop=0.0):
        super().__init__()
        if qkv_bias or proj_drop or attn_drop:
            raise NotImplementedError("Block: qkv_bias/dropout are not used by the LARP tokenizer and are not built")
        self.norm1 = nn.LayerNorm(dim)  # eps 1e-5 (timm default)
        self.attn = _Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


@register("transformer_encoder_parallel")
class TransformerEncoderParallel(nn.Module):
    def __init__(self, dim, depth, n_head, head_dim, ff_dim=None, dropout=0.0):
        super().__init__()
        self.is_encoder_decoder = True
        assert ff_dim is None
        assert dim == head_dim * n_head
        if dropout:
            raise NotImplementedError("dropout > 0 is not built")
        if head_dim not in (32, 64):
            raise NotImplementedError(f"head_dim {head_dim}: the gfx950 attention kernels are built for head_dim 64 and 32")
        self.dim, self.depth, self.n_head = dim, depth, n_head
        self.blocks = nn.ModuleList([Block(dim, n_head) for _ in range(depth)])

    def forward(self, context, query):
        """transformer.py:62-70: cat([context, query]) -> blocks -> last len(query) rows (fp32 in, fp32 out)"""
        from .functional import block_stack
        nq = query.size(1)
        h = block_stack(torch.cat([context.float(), query.float()], dim=1), self.blocks, self.n_head)
        return h[:, -nq:, :]


@register("transformer_encoder_fused")
class TransformerEncoderFused(nn.Module):
    """models/transformer.py:8-31: nn.Sequential of `depth` timm Blocks, forward(x) = blocks(x)"""

    def __init__(self, dim, depth, n_head, head_dim, ff_dim=None, dropout=0.0):
        super().__init__()
        self.layers = nn.ModuleList()
        assert ff_dim is None
        assert dim == head_dim * n_head
        if dropout:
            raise NotImplementedError("dropout > 0 is not built")
        if head_dim not in (32, 64):
            raise NotImplementedError(f"head_dim {head_dim}: the gfx950 attention kernels are built for head_dim 64 and 32")
        self.dim, self.depth, self.n_head = dim, depth, n_head
        self.blocks = nn.Sequential(*[Block(dim, n_head) for _ in range(depth)])

    def forward(self, x):
        from .functional import block_stack
        return block_stack(x.float(), self.blocks, self.n_head)
