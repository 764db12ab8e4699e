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

Not built (raise at construction): LPIPS -- the `lpips` package and its VGG weights are not available offline
(SURVEY §8c); pass perceptual_weight=0 or a callable `perceptual_loss(input_frames, recon_frames) -> tensor`;
r1_gp_weight > 0 (needs double backward through the HIP ops); spectral_norm; temporal_patch_size == 1
(VideoPatchEmbed); disc_type other than 'transformer'.
"""
from itertools import chain

import torch
import torch.nn as nn
import torch.nn.functional as F

from .embed import PatchEmbed3D, get_3d_sincos_pos_embed
from .registry import register
from .transformer import TransformerEncoderFused


def lecam_reg(real_pred, fake_pred, ema_real_pred, ema_fake_pred):
    """loss.py:17-35 (https://arxiv.org/abs/2104.03310)"""
    assert real_pred.ndim == 0 and ema_fake_pred.ndim == 0
    lecam_loss = torch.mean(torch.pow(torch.relu(real_pred - ema_fake_pred), 2))
    return lecam_loss + torch.mean(torch.pow(torch.relu(ema_real_pred - fake_pred), 2))


def hinge_d_loss(logits_real, logits_fake):
    return 0.5 * (torch.mean(F.relu(1.0 - logits_real)) + torch.mean(F.relu(1.0 + logits_fake)))


def hinge_g_loss(logits_fake):
    return -torch.mean(logits_fake)


def ns_d_loss(logits_real, logits_fake):
    real_loss = F.binary_cross_entropy_with_logits(logits_real, torch.ones_like(logits_real))
    fake_loss = F.binary_cross_entropy_with_logits(logits_fake, torch.zeros_like(logits_fake))
    return real_loss + fake_loss


def ns_d_loss_single_side_smooth(logits_real, logits_fake):
    """loss.py:82-92: targets 1 - |N(0, 0.15)| clamped at 0.7 for real, |N(0, 0.15)| clamped at 0.3 for fake"""
    real_target = torch.ones_like(logits_real) - torch.randn_like(logits_real).abs() * 0.15
    real_target.clamp_min_(0.7)
    fake_target = torch.randn_like(logits_fake).abs() * 0.15
    fake_target.clamp_max_(0.3)
    return F.binary_cross_entropy_with_logits(logits_real, real_target) + F.binary_cross_entropy_with_logits(logits_fake, fake_target)


def ns_g_loss(logits_fake):
    return -torch.mean(F.logsigmoid(logits_fake))


def adopt_weight(weight, global_step, threshold=0, value=0.0):
    return value if global_step < threshold else weight


def measure_perplexity(predicted_indices, n_embed):
    encodings = F.one_hot(predicted_indices, n_embed).float().reshape(-1, n_embed)
    avg_probs = encodings.mean(0)
    perplexity = (-(avg_probs * torch.log(avg_probs + 1e-10)).sum()).exp()
    return perplexity, torch.sum(avg_probs > 0)


def l1(x, y):
    return torch.abs(x - y)


def l2(x, y):
    return torch.pow((x - y), 2)


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
        if temporal_patch_size == 1:
            raise NotImplementedError("TransformerDiscriminator: temporal_patch_size == 1 (VideoPatchEmbed) is not built; "
                                      "every shipped yaml sets disc_tran_temporal_patch_size: 4")
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


@register("lpips_disc_loss")
class VQLPIPSWithDiscriminator(nn.Module):
    def __init__(self, disc_start, disc_self_start=None, pixelloss_weight=1.0, disc_type="transformer", disc_in_channels=3,
                 disc_factor=1.0, disc_weight=1.0, perceptual_weight=1.0, disc_loss="hing", disc_tran_hidden_size=256,
                 disc_tran_n_heads=8, disc_tran_n_layers=6, disc_tran_temporal_patch_size=1, disc_tran_patch_size=16, frame_num=16,
                 perceptual_loss="lpips", perceptual_fp16=False, pixel_loss="l1", lecam_weight=0.0, input_spatial_size=128,
                 r1_gp_weight=0.0, d_update_freq=1, d_update_loss_threshold=-1.0e6, spectral_norm=False):
        super().__init__()
        assert disc_loss in ["hinge", "ns", "ns_smooth"]
        assert pixel_loss in ["l1", "l2"]
        self.pixel_weight = pixelloss_weight
        self.perceptual_weight = perceptual_weight
        if callable(perceptual_loss):
            self.perceptual_loss = perceptual_loss
            self.set_perceptual_eval()
        elif perceptual_weight > 0:
            raise NotImplementedError("perceptual_loss='lpips': the lpips package / VGG weights are not available offline; "
                                      "pass perceptual_weight=0 or a callable perceptual_loss(input_frames, recon_frames)")
        else:
            self.perceptual_loss = None
        self.pixel_loss = l1 if pixel_loss == "l1" else l2
        self.input_spatial_size = input_spatial_size
        if r1_gp_weight > 0.0:
            raise NotImplementedError("r1_gp_weight > 0 needs double backward through the HIP ops; not built (shipped yamls set 0.0)")
        self.r1_gp_weight = r1_gp_weight
        self.d_update_freq = d_update_freq
        self.d_update_loss_threshold = d_update_loss_threshold
        if disc_type.lower() != "transformer":
            raise ValueError(f"Unknown discriminator type: >> {disc_type} <<")
        self.discriminator = TransformerDiscriminator(hidden_size=disc_tran_hidden_size, n_heads=disc_tran_n_heads, n_layers=disc_tran_n_layers,
                                                      input_size=input_spatial_size, temporal_patch_size=disc_tran_temporal_patch_size,
                                                      patch_size=disc_tran_patch_size, in_channels=disc_in_channels, frame_num=frame_num)
        self.disc_type = "3d"
        if spectral_norm:
            raise NotImplementedError("spectral_norm=True is not built (shipped yamls set false)")
        self.discriminator_iter_start = disc_start
        self.discriminator_self_start = disc_self_start if (disc_self_start is not None and disc_self_start >= 0) else disc_start
        if disc_loss == "hinge":
            self.disc_loss, self.g_loss = hinge_d_loss, hinge_g_loss
        elif disc_loss == "ns":
            self.disc_loss, self.g_loss = ns_d_loss, ns_g_loss
        else:
            self.disc_loss, self.g_loss = ns_d_loss_single_side_smooth, ns_g_loss
        self.disc_factor = disc_factor
        self.discriminator_weight = disc_weight
        self.lecam_weight = lecam_weight
        if self.lecam_weight > 0.0:
            self.register_buffer("lecam_ema_real", torch.tensor(0.0))
            self.register_buffer("lecam_ema_fake", torch.tensor(0.0))

    def set_perceptual_eval(self):
        if isinstance(self.perceptual_loss, nn.Module):
            self.perceptual_loss.eval()
            for param in self.perceptual_loss.parameters():
                param.requires_grad_(False)

    def trainable_requires_grad_(self, requires_grad):
        for param in self.trainable_parameters():
            param.requires_grad_(requires_grad)

    def trainable_modules(self):
        return [self.discriminator]

    def trainable_parameters(self):
        return chain(*(m.parameters() for m in self.trainable_modules()))

    def set_training_mode(self, trainable_mode, others_mode=False):
        self.train(others_mode)
        for m in self.trainable_modules():
            m.train(trainable_mode)

    @torch.no_grad()
    def update_lecam_ema(self, real, fake, decay=0.999):
        real, fake = real.float().mean(), fake.float().mean()
        self.lecam_ema_real.mul_(decay).add_(real, alpha=1 - decay)
        self.lecam_ema_fake.mul_(decay).add_(fake, alpha=1 - decay)

    def forward_perceptual(self, inputs, reconstructions):
        if self.perceptual_loss is None:
            raise NotImplementedError("no perceptual loss configured")
        return {"loss_prior": self.perceptual_loss(_frames(inputs), _frames(reconstructions)).mean()}

    def forward(self, inputs, reconstructions, global_step, for_discriminator=False, last_layer=None):
        """loss.py:338-456.  Returns (loss, info_dict, p_loss_per_sample | None); info values are detached 0-dim tensors
        (the reference calls .item() on each: a host sync per entry) except `g_loss_weight`, a Python float."""
        input_frames, recon_frames = _frames(inputs), _frames(reconstructions)
        zero = input_frames.new_zeros(1)
        if not for_discriminator:
            disc_factor = adopt_weight(self.disc_factor, global_step, threshold=self.discriminator_iter_start)
            rec_loss = self.pixel_loss(input_frames, recon_frames) if self.pixel_weight > 0 else zero
            if self.perceptual_weight > 0:
                p_loss = self.perceptual_loss(input_frames, recon_frames)
            else:
                p_loss = zero
            nll_loss = torch.mean(self.pixel_weight * rec_loss + self.perceptual_weight * p_loss)
            if disc_factor > 0.0:
                logits_fake = self.discriminator(reconstructions)
                g_loss = self.g_loss(logits_fake)
                d_weight = self.discriminator_weight
            else:
                d_weight, g_loss = 0.0, zero
            g_loss_weight = float(d_weight * disc_factor)
            loss = nll_loss + g_loss_weight * g_loss
            info = {"rec_loss": rec_loss.mean().detach(), "perceptual_loss": p_loss.mean().detach(), "rp_loss": nll_loss.detach(),
                    "g_loss": g_loss.mean().detach(), "g_loss_weight": g_loss_weight}
            return loss, info, 0
        disc_factor = adopt_weight(self.disc_factor, global_step, threshold=self.discriminator_self_start)
        if disc_factor > 0.0:
            # one pass over [real ; fake]: the discriminator has no batch-coupled op (LayerNorm and attention are per
            # clip), so this equals the reference's two calls (:417-424) and halves the launches
            nb = inputs.shape[0]
            logits = self.discriminator(torch.cat([inputs, reconstructions.detach()], dim=0))
            logits_real, logits_fake = logits[:nb], logits[nb:]
            if self.lecam_weight > 0.0:
                lecam_loss = self.lecam_weight * lecam_reg(real_pred=logits_real.mean(), fake_pred=logits_fake.mean(),
                                                           ema_real_pred=self.lecam_ema_real, ema_fake_pred=self.lecam_ema_fake)
                self.update_lecam_ema(logits_real, logits_fake)
            else:
                lecam_loss = zero
            d_loss = self.disc_loss(logits_real, logits_fake)
            total_loss = d_loss + self.lecam_weight * lecam_loss     # (sic) the reference applies lecam_weight twice, loss.py:426-437
        else:
            d_loss = lecam_loss = total_loss = logits_real = logits_fake = zero
        info = {"d_total_loss": total_loss.mean().detach(), "d_lecam_loss": lecam_loss.mean().detach(), "d_loss": d_loss.mean().detach(),
                "logits_real": logits_real.mean().detach(), "logits_fake": logits_fake.mean().detach()}
        return total_loss, info, None
