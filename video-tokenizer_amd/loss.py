"""GAN branch of the tokenizer step: `TransformerDiscriminator` and the `lpips_disc_loss` module.

Mirrors /root/reference/models/loss.py: TransformerDiscriminator (:119-204; PatchEmbed3D + sincos PE + cls token +
`transformer_encoder_fused` stack of timm Blocks + LayerNorm(1e-6) + Linear(D, 1) on the cls row) and
VQLPIPSWithDiscriminator (:207-456; generator branch: pixel (+ perceptual) + g_loss_weight * g_loss, discriminator
branch: hinge / non-saturating / one-side-smoothed non-saturating loss + LeCam regulariser).  Same constructor
keywords, state-dict keys and return tuples, so trainers/larp_tokenizer_trainer.py:263-301 can drive it.

Where the arithmetic runs: patch gather, patch-embed GEMM (+PE), every Block (head_dim 32 at the shipped
disc_tran_hidden_size 384 / 12 heads, odd L = 1025 with the cls token) forward and backward incl. the gradient
w.r.t. the input video (the generator update back-propagates through the discriminator) are libvt_hip kernels via
functional.{PatchEmbed,BlockStack}.  The cls-row head (B x D LayerNorm + a D-vector dot) and the scalar GAN loss
formulas over B logits are torch glue.

LPIPS (`perceptual_loss='lpips'`, weight 1.0 in the shipped yamls) is lpips.py: a torch-ops VGG-16 metric with the `lpips`
package's state-dict layout, frozen, weights from a user-supplied state dict (env VT_LPIPS_WEIGHTS or the checkpoint's
`loss` entry) -- parity unpinned, the package is not importable here; a callable `perceptual_loss(input_frames,
recon_frames) -> tensor` is accepted too.  r1_gp_weight > 0 (no shipped yaml): the penalty needs the discriminator differentiated twice,
which the single-backward HIP functions cannot give, so that one term evaluates it as plain torch ops (_discriminator_torch_ops).
Not built (raise at construction): disc_type other than 'transformer'.
"""
import os
from itertools import chain

import torch
import torch.nn as nn
import torch.nn.functional as F

from .embed import PatchEmbed3D, VideoPatchEmbed, get_3d_sincos_pos_embed
from .registry import register
from .transformer import TransformerEncoderFused


class _GanObjective:
    """The scalar GAN formulas of `lpips_disc_loss` over the B discriminator logits, in one place.

    `kind` is the yaml's `disc_loss` (/root/reference/models/loss.py:64-96, 288-299):
      hinge      d = (mean relu(1 - real) + mean relu(1 + fake)) / 2                 g = -mean fake
      ns         d = bce(real, 1) + bce(fake, 0)                                       g = mean softplus(-fake)
      ns_smooth  the same with one-sided smoothed targets: real max(1 - 0.15 |n|, 0.7), fake min(0.15 |n|, 0.3), n ~ N(0, 1)
    with bce(x, t) = mean(softplus(x) - t x), the logit form of binary cross entropy (no sigmoid, no log of it)."""

    KINDS = ("hinge", "ns", "ns_smooth")

    def __init__(self, kind):
        if kind not in self.KINDS:
            raise AssertionError(f"disc_loss must be one of {self.KINDS}, got {kind!r}")
        self.kind = kind

    @staticmethod
    def _bce(logits, target):
        return (F.softplus(logits) - target * logits).mean()

    def discriminator(self, real, fake):
        if self.kind == "hinge":
            return 0.5 * (F.relu(1.0 - real).mean() + F.relu(1.0 + fake).mean())
        if self.kind == "ns":
            return self._bce(real, 1.0) + self._bce(fake, 0.0)
        t_real = (1.0 - 0.15 * torch.randn_like(real).abs()).clamp(min=0.7)
        t_fake = (0.15 * torch.randn_like(fake).abs()).clamp(max=0.3)
        return self._bce(real, t_real) + self._bce(fake, t_fake)

    def generator(self, fake):
        return -fake.mean() if self.kind == "hinge" else F.softplus(-fake).mean()


def _lecam(real_mean, fake_mean, ema_real, ema_fake):
    """LeCam regulariser (arXiv 2104.03310; reference :17-35) on the batch means and their running averages (all 0-dim)"""
    return F.relu(real_mean - ema_fake).square() + F.relu(ema_real - fake_mean).square()


def _started(weight, step, start):
    """a loss weight that is 0 before `start` (the reference's adopt_weight, :98-101)"""
    return weight if step >= start else 0.0


def _frames(x):
    """'b c t h w -> (b t) c h w'"""
    b, c, t, h, w = x.shape
    return x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w).contiguous()


class TransformerDiscriminator(nn.Module):
    def __init__(self, hidden_size, n_heads, n_layers, input_size, temporal_patch_size, patch_size, in_channels, frame_num=16):
        super().__init__()
        self.hidden_size = hidden_size
        self.n_head = n_heads
        self.n_layers = n_layers
        self.input_size = input_size
        self.temporal_patch_size = temporal_patch_size
        self.patch_size = patch_size
        self.in_channels = in_channels
        self.frame_num = frame_num
        if temporal_patch_size == 1:      # loss.py:137-138: every frame through a 2-D patch embed (the constructor's default; the yamls set 4)
            self.x_embedder = VideoPatchEmbed(input_size, patch_size, in_channels, hidden_size, bias=True, frame_num=frame_num)
        else:
            assert temporal_patch_size > 1
            self.x_embedder = PatchEmbed3D(input_size, frame_num, patch_size, temporal_patch_size, in_channels, hidden_size, bias=True)
        self.token_t = self.x_embedder.num_temporal_patches
        self.token_h = self.token_w = int(self.x_embedder.num_spatial_patches ** 0.5)
        self.video_token_num = video_token_num = self.x_embedder.num_spatial_patches * self.token_t
        self.cls_token = nn.Parameter(torch.randn(1, 1, hidden_size))
        self.register_buffer("encoder_pos_embed", torch.zeros(1, video_token_num, hidden_size))
        self.get_encoder_pos_embed = lambda: self.encoder_pos_embed
        self.transformer_encoder = TransformerEncoderFused(dim=hidden_size, depth=n_layers, n_head=n_heads, head_dim=hidden_size // n_heads)
        self.norm_final = nn.LayerNorm(hidden_size, eps=1e-6)
        self.fc = nn.Linear(hidden_size, 1)
        self.initialize_weights()

    def initialize_weights(self):
        """loss.py:165-186"""
        def _basic_init(module):
            if isinstance(module, nn.Linear):
                torch.nn.init.xavier_uniform_(module.weight)
                if module.bias is not None:
                    nn.init.constant_(module.bias, 0)
            elif isinstance(module, nn.LayerNorm):
                module.reset_parameters()

        self.apply(_basic_init)
        pe = get_3d_sincos_pos_embed(self.hidden_size, self.token_h, self.token_t)
        self.encoder_pos_embed.data.copy_(torch.from_numpy(pe).float().reshape_as(self.encoder_pos_embed))
        w = self.x_embedder.proj.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.constant_(self.x_embedder.proj.bias, 0)
        nn.init.xavier_uniform_(self.cls_token)

    def forward(self, x):
        """x: (b, c, t, h, w) -> logits (b, 1)   (loss.py:188-201)"""
        b = x.shape[0]
        tok = self.x_embedder(x, pos_embed=self.encoder_pos_embed[0])           # (b, n, d): conv (bf16) + PE in fp32
        seq = torch.cat((self.cls_token.float().expand(b, -1, -1), tok), dim=1)  # (b, n+1, d)
        z = self.transformer_encoder(seq)
        z_cls = F.layer_norm(z[:, 0], (self.hidden_size,), self.norm_final.weight, self.norm_final.bias, 1e-6)
        rb = lambda t: t.to(torch.bfloat16).float()                              # Linear under autocast: bf16 operands and output
        return rb(F.linear(rb(z_cls), rb(self.fc.weight)) + self.fc.bias)


def _spectral_normalise(module):
    """loss.py:59-64, 275-276: every Conv3d / Linear of the discriminator gets torch's spectral-norm parametrization (`weight` = original /
    sigma, one power-iteration step per training forward; state-dict keys `...parametrizations.weight.original` / `.0._u` / `.0._v`, the
    reference's).  The layers of this build are parameter holders whose `.weight` the HIP autograd functions read once per forward, so the
    parametrized tensor -- ordinary differentiable torch ops on a [out, in] matrix -- is what the kernels consume and what their
    weight gradients flow back through."""
    from torch.nn.utils.parametrizations import spectral_norm
    wrap = [(parent, name) for parent in module.modules() for name, child in parent.named_children()
            if isinstance(child, (nn.Conv2d, nn.Conv3d, nn.Linear))]
    for parent, name in wrap:
        setattr(parent, name, spectral_norm(getattr(parent, name)))


def _discriminator_torch_ops(D, x):
    """TransformerDiscriminator.forward (loss.py:188-201) out of plain torch ops on the GPU, for the one caller that needs to differentiate
    the discriminator TWICE: the R1 gradient penalty (loss.py:36-56, autograd.grad(..., create_graph=True) w.r.t. the real clip, then a
    backward through that gradient into the weights).  The HIP autograd functions of this build are single-backward, so this term -- and
    only this term, when r1_gp_weight > 0, which no shipped yaml sets -- takes the unfused path: Conv3d, LayerNorm, Linear, an explicit
    softmax(q k^T / sqrt(d)) v and erf-GELU under bf16 autocast like the reference's modules; not a kernel of this build and never timed."""
    b = x.shape[0]
    H, hd = D.n_head, D.hidden_size // D.n_head
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        emb = D.x_embedder
        w = emb.proj.weight      # read ONCE: under the spectral-norm parametrization every read of `.weight` in train mode is a power iteration
        w = w if w.dim() == 5 else w.unsqueeze(2)      # VideoPatchEmbed: a Conv2d per frame
        tok = F.conv3d(x, w, emb.proj.bias, stride=w.shape[2:]).flatten(2).transpose(1, 2).float() + D.encoder_pos_embed
        h = torch.cat((D.cls_token.float().expand(b, -1, -1), tok), dim=1)
        for blk in D.transformer_encoder.blocks:
            y = F.layer_norm(h, (D.hidden_size,), blk.norm1.weight, blk.norm1.bias, 1e-5)
            q, k, v = F.linear(y, blk.attn.qkv.weight).reshape(b, -1, 3, H, hd).permute(2, 0, 3, 1, 4)
            att = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1) @ v
            h = h + F.linear(att.transpose(1, 2).reshape(b, -1, D.hidden_size), blk.attn.proj.weight, blk.attn.proj.bias).float()
            y = F.layer_norm(h, (D.hidden_size,), blk.norm2.weight, blk.norm2.bias, 1e-5)
            h = h + F.linear(F.gelu(F.linear(y, blk.mlp.fc1.weight, blk.mlp.fc1.bias)), blk.mlp.fc2.weight, blk.mlp.fc2.bias).float()
        z = F.layer_norm(h[:, 0], (D.hidden_size,), D.norm_final.weight, D.norm_final.bias, 1e-6)
        return F.linear(z, D.fc.weight, D.fc.bias).float()


def _r1_gradient_penalty(D, real, cost):
    """loss.py:36-56: (logits of the real clips, cost * mean_b || d logits / d real ||^2), the gradient taken with create_graph=True"""
    leaf = real.detach().clone().requires_grad_(True)
    logits = _discriminator_torch_ops(D, leaf)
    (slope,) = torch.autograd.grad(logits.sum(), leaf, create_graph=True)          # d(sum of logits) / d clip = every clip's own input gradient
    return logits, cost * slope.float().flatten(1).square().sum(dim=1).mean()


@register("lpips_disc_loss")
class VQLPIPSWithDiscriminator(nn.Module):
    def __init__(self, disc_start, disc_self_start=None, pixelloss_weight=1.0, disc_type="transformer", disc_in_channels=3,
                 disc_factor=1.0, disc_weight=1.0, perceptual_weight=1.0, disc_loss="hing", disc_tran_hidden_size=256,
                 disc_tran_n_heads=8, disc_tran_n_layers=6, disc_tran_temporal_patch_size=1, disc_tran_patch_size=16, frame_num=16,
                 perceptual_loss="lpips", perceptual_fp16=False, pixel_loss="l1", lecam_weight=0.0, input_spatial_size=128,
                 r1_gp_weight=0.0, d_update_freq=1, d_update_loss_threshold=-1.0e6, spectral_norm=False):
        super().__init__()
        self.objective = _GanObjective(disc_loss)
        assert pixel_loss in ["l1", "l2"]
        self.pixel_weight = pixelloss_weight
        self.perceptual_weight = perceptual_weight
        if callable(perceptual_loss):
            self.perceptual_loss = perceptual_loss
        elif perceptual_loss == "lpips":
            # the shipped yamls set perceptual_weight: 1.0 (cfgs/larp_tokenizer.yaml:120): a torch-ops LPIPS (lpips.py, parity
            # unpinned) with the package's state-dict layout; weights come from VT_LPIPS_WEIGHTS or a checkpoint's `loss` entry
            from .lpips import LPIPS, load_lpips_state_dict
            self.perceptual_loss = LPIPS(net="vgg")
            if os.environ.get("VT_LPIPS_WEIGHTS"):
                load_lpips_state_dict(self.perceptual_loss, os.environ["VT_LPIPS_WEIGHTS"])
            if perceptual_fp16:
                self.perceptual_loss = self.perceptual_loss.to(dtype=torch.float16)
        else:
            raise ValueError(f"Unknown perceptual loss: >> {perceptual_loss} <<")
        self.set_perceptual_eval()
        self.pixel_power = 1 if pixel_loss == "l1" else 2
        self.input_spatial_size = input_spatial_size
        self.r1_gp_weight = r1_gp_weight       # > 0: the R1 term runs the discriminator as twice-differentiable torch ops (_r1_gradient_penalty)
        self.d_update_freq = d_update_freq
        self.d_update_loss_threshold = d_update_loss_threshold
        if disc_type.lower() != "transformer":
            raise ValueError(f"Unknown discriminator type: >> {disc_type} <<")
        self.discriminator = TransformerDiscriminator(hidden_size=disc_tran_hidden_size, n_heads=disc_tran_n_heads, n_layers=disc_tran_n_layers,
                                                      input_size=input_spatial_size, temporal_patch_size=disc_tran_temporal_patch_size,
                                                      patch_size=disc_tran_patch_size, in_channels=disc_in_channels, frame_num=frame_num)
        self.disc_type = "3d"
        self.spectral_norm = bool(spectral_norm)
        if spectral_norm:
            _spectral_normalise(self.discriminator)
        self.discriminator_iter_start = disc_start
        self.discriminator_self_start = disc_self_start if (disc_self_start is not None and disc_self_start >= 0) else disc_start
        self.disc_factor = disc_factor
        self.discriminator_weight = disc_weight
        self.lecam_weight = lecam_weight
        if self.lecam_weight > 0.0:
            self.register_buffer("lecam_ema_real", torch.tensor(0.0))
            self.register_buffer("lecam_ema_fake", torch.tensor(0.0))

    def _perceptual(self, a, b):
        if isinstance(self.perceptual_loss, nn.Module) and hasattr(self.perceptual_loss, "scaling_layer"):
            dt = self.perceptual_loss.scaling_layer.shift.dtype
            return self.perceptual_loss(a.to(dt), b.to(dt), normalize=True).float()       # the reference's call, :335, 370-372
        return self.perceptual_loss(a, b)

    def _pixel(self, a, b):
        d = a - b
        return d.abs() if self.pixel_power == 1 else d * d

    # ---- what trainers/larp_tokenizer_trainer.py:121-122, 163, 267-292 call on the loss module ----
    def trainable_modules(self):
        return [self.discriminator]

    def trainable_parameters(self):
        return chain.from_iterable(m.parameters() for m in self.trainable_modules())

    def trainable_requires_grad_(self, requires_grad):
        for p in self.trainable_parameters():
            p.requires_grad_(requires_grad)

    def set_perceptual_eval(self):
        metric = self.perceptual_loss
        if isinstance(metric, nn.Module):
            metric.requires_grad_(False).eval()

    def set_training_mode(self, trainable_mode, others_mode=False):
        self.train(others_mode)                       # everything (the frozen metric stays in eval by its own train())
        for m in self.trainable_modules():
            m.train(trainable_mode)

    @torch.no_grad()
    def update_lecam_ema(self, real, fake, decay=0.999):
        for buf, logits in ((self.lecam_ema_real, real), (self.lecam_ema_fake, fake)):
            buf.lerp_(logits.float().mean(), 1.0 - decay)

    def forward_perceptual(self, inputs, reconstructions):
        if self.perceptual_loss is None:
            raise NotImplementedError("no perceptual loss configured")
        return {"loss_prior": self._perceptual(_frames(inputs), _frames(reconstructions)).mean()}

    def forward(self, inputs, reconstructions, global_step, for_discriminator=False, last_layer=None):
        """loss.py:338-456.  Returns (loss, info_dict, p_loss_per_sample | None); info values are detached 0-dim tensors
        (the reference calls .item() on each: a host sync per entry) except `g_loss_weight`, a Python float."""
        input_frames, recon_frames = _frames(inputs), _frames(reconstructions)
        zero = input_frames.new_zeros(1)
        if not for_discriminator:
            disc_factor = _started(self.disc_factor, global_step, self.discriminator_iter_start)
            rec_loss = self._pixel(input_frames, recon_frames) if self.pixel_weight > 0 else zero
            p_loss = self._perceptual(input_frames, recon_frames) if self.perceptual_weight > 0 else zero
            nll_loss = torch.mean(self.pixel_weight * rec_loss + self.perceptual_weight * p_loss)
            if disc_factor > 0.0:
                logits_fake = self.discriminator(reconstructions)
                g_loss = self.objective.generator(logits_fake)
                d_weight = self.discriminator_weight
            else:
                d_weight, g_loss = 0.0, zero
            g_loss_weight = float(d_weight * disc_factor)
            loss = nll_loss + g_loss_weight * g_loss
            info = {"rec_loss": rec_loss.mean().detach(), "perceptual_loss": p_loss.mean().detach(), "rp_loss": nll_loss.detach(),
                    "g_loss": g_loss.mean().detach(), "g_loss_weight": g_loss_weight}
            return loss, info, 0
        disc_factor = _started(self.disc_factor, global_step, self.discriminator_self_start)
        if disc_factor > 0.0:
            # one pass over [real ; fake]: the discriminator has no batch-coupled op (LayerNorm and attention are per
            # clip), so this equals the reference's two calls (:417-424) and halves the launches
            nb = inputs.shape[0]
            r1_gp = zero
            if self.training and self.r1_gp_weight > 0.0:      # loss.py:415-418: the real logits come out of the penalty's own forward
                logits_real, r1_gp = _r1_gradient_penalty(self.discriminator, inputs.contiguous(), self.r1_gp_weight)
                logits_fake = self.discriminator(reconstructions.detach().contiguous())
            elif self.spectral_norm:
                # the parametrization advances its power iteration once per training forward: the reference's two calls (:417-424) take
                # two steps per update and the fake logits see the second sigma, so the two calls are kept apart here as well
                logits_real = self.discriminator(inputs.contiguous())
                logits_fake = self.discriminator(reconstructions.detach().contiguous())
            else:
                logits = self.discriminator(torch.cat([inputs, reconstructions.detach()], dim=0))
                logits_real, logits_fake = logits[:nb], logits[nb:]
            if self.lecam_weight > 0.0:
                lecam_loss = self.lecam_weight * _lecam(logits_real.mean(), logits_fake.mean(), self.lecam_ema_real, self.lecam_ema_fake)
                self.update_lecam_ema(logits_real, logits_fake)
            else:
                lecam_loss = zero
            d_loss = self.objective.discriminator(logits_real, logits_fake)
            total_loss = d_loss + self.lecam_weight * lecam_loss + r1_gp     # (sic) the reference applies lecam_weight twice, loss.py:426-437
        else:
            d_loss = lecam_loss = total_loss = logits_real = logits_fake = r1_gp = zero
        info = {"d_total_loss": total_loss.mean().detach(), "d_lecam_loss": lecam_loss.mean().detach(), "d_loss": d_loss.mean().detach(),
                "logits_real": logits_real.mean().detach(), "logits_fake": logits_fake.mean().detach()}
        if self.r1_gp_weight > 0.0:
            info["r1_gp"] = r1_gp.mean().detach()
        return total_loss, info, None
