"""`larp_tokenizer`: the LARP video tokenizer behind the reference's model-registry interface.

Drop-in for /root/reference/models/larp_tokenizer.py `LARPTokenizer` (:44-496) with
bottleneck_type='vq': same constructor keywords (unknown extras are accepted and ignored so that
`from_checkpoint(cls(**ckpt['model']['args']))` works with the yaml's extra keys), same state-dict
keys/shapes, same output-dict keys, same public methods (encode / decode / encode_eval /
decode_eval / decode_from_bottleneck / from_checkpoint / set_vq_eval_deterministic ...).
forward/backward run in the fused HIP engine (engine.py -> libvt_hip.so); there is no CPU path.
"""
import itertools
import os

import numpy as np
import torch
import torch.nn as nn

from . import engine as _engine
from . import hip
from . import registry
from .embed import PatchEmbed3D, get_1d_sincos_pos_embed_from_grid, get_3d_sincos_pos_embed
from .pretrained import LocalPretrainedMixin
from .registry import register


class OutputLayer(nn.Module):
    """LayerNorm(eps 1e-6) + Linear(hidden, pt*p*p*c) parameter holder (larp_tokenizer.py:31-41)."""

    def __init__(self, hidden_size, temporal_patch_size, patch_size, out_channels):
        super().__init__()
        self.norm_final = nn.LayerNorm(hidden_size, eps=1e-6)
        self.linear = nn.Linear(hidden_size, temporal_patch_size * patch_size * patch_size * out_channels, bias=True)


@register("larp_tokenizer")
class LARPTokenizer(nn.Module, LocalPretrainedMixin):
    # LocalPretrainedMixin: from_pretrained / save_pretrained on a local directory, the layout of the reference's PyTorchModelHubMixin
    # base (larp_tokenizer.py:45; eval/eval_larp_tokenizer.py:40)
    output_format = "bcthw"

    def __init__(self, bottleneck, prior_model=None, bottleneck_token_num=1024, input_size=128, frame_num=16,
                 temporal_patch_size=4, patch_size=8, decoder_temporal_patch_size=4, decoder_patch_size=8, in_channels=3,
                 bottleneck_type="auto", transformer_name="transformer_encoder_parallel", encoder_name=None, decoder_name=None,
                 latent_pe_scale_factor=10000, query_init_std=0.02, encoder_hidden_size=768, decoder_hidden_size=768,
                 encoder_num_heads=12, decoder_num_heads=12, encoder_depth=6, decoder_depth=6, train_type="simple",
                 learned_encoder_patch_pe=False, learned_encoder_latent_query_embed=True, learned_decoder_latent_pe=False,
                 learned_decoder_patch_query_embed=False, use_encoder_patch_token_type_embed=False,
                 use_encoder_latent_query_token_type_embed=False, use_decoder_latent_token_type_embed=False,
                 use_decoder_patch_query_token_type_embed=False, encoder_query_gaussian_init=True, mrope_args=None, **ignored):
        super().__init__()
        # `ignored`: yaml keys the reference class does not take (e.g. use_pe, cfgs/larp_tokenizer.yaml:75)
        if bottleneck_type not in ("vq", "sq", "fsq"):
            raise NotImplementedError(f"bottleneck_type='{bottleneck_type}': this build implements the 'vq', 'sq' and 'fsq' bottlenecks "
                                      "(pass model.args.bottleneck_type vq); 'auto' builds no bottleneck in the reference either")
        if train_type not in ("simple", "mrope"):
            raise NotImplementedError(f"train_type '{train_type}': 'simple' and 'mrope' are built")
        self.train_type = train_type
        # Options beyond what the fused engine carries (learned / token-type position embeddings, fixed latent queries, a per-frame patch
        # embed, a normalised bottleneck, the 'fsq' branch) run on the COMPOSED path: the same kernels through the sub-modules' own
        # autograd functions, the small embedding sums and the bottleneck LayerNorm as torch glue (forward(): self._composed).
        extra = dict(learned_encoder_patch_pe=learned_encoder_patch_pe, learned_decoder_latent_pe=learned_decoder_latent_pe,
                     learned_decoder_patch_query_embed=learned_decoder_patch_query_embed,
                     use_encoder_patch_token_type_embed=use_encoder_patch_token_type_embed,
                     use_encoder_latent_query_token_type_embed=use_encoder_latent_query_token_type_embed,
                     use_decoder_latent_token_type_embed=use_decoder_latent_token_type_embed,
                     fixed_latent_queries=not learned_encoder_latent_query_embed, per_frame_patch_embed=temporal_patch_size == 1)
        bn = bottleneck.get("args", {}).get("norm") if isinstance(bottleneck, dict) and bottleneck_type == "vq" else None
        extra["bottleneck_norm"] = bn is not None and str(bn).lower() not in ("no", "none")
        extra["entropy_loss"] = bottleneck_type == "vq" and float(bottleneck["args"]["regularizer"]["args"].get("entropy_loss_weight", 0.0)) > 0
        extra["train_type_mrope"] = train_type == "mrope"
        self._composed = bottleneck_type == "fsq" or any(extra.values())
        self._composed_why = ", ".join((["bottleneck_type='fsq'"] if bottleneck_type == "fsq" else []) + [k for k, v in extra.items() if v])
        assert temporal_patch_size >= 1
        assert (temporal_patch_size, patch_size) == (decoder_temporal_patch_size, decoder_patch_size), \
            "unpatchify uses the ENCODER patch sizes (larp_tokenizer.py:447-449): encoder and decoder patch sizes must match"
        assert encoder_hidden_size == decoder_hidden_size and encoder_num_heads == decoder_num_heads, \
            "the fused engine is built for equal encoder/decoder width"

        self.train_type = train_type
        self.bottleneck_type = bottleneck_type
        self.in_channels = in_channels
        self.out_channels = in_channels
        self.input_size = input_size
        self.frame_num = frame_num
        self.bottleneck_token_num = bottleneck_token_num
        self.temporal_patch_size = temporal_patch_size
        self.patch_size = patch_size
        self.decoder_temporal_patch_size = decoder_temporal_patch_size
        self.decoder_patch_size = decoder_patch_size
        self.decoder_latent_len = bottleneck_token_num
        self.encoder_hidden_size = encoder_hidden_size = int(encoder_hidden_size)
        self.decoder_hidden_size = decoder_hidden_size = int(decoder_hidden_size)
        self.encoder_num_heads = encoder_num_heads = int(encoder_num_heads)
        self.decoder_num_heads = decoder_num_heads = int(decoder_num_heads)
        self.latent_pe_scale_factor = latent_pe_scale_factor
        self.query_init_std = query_init_std

        if temporal_patch_size == 1:
            from .embed import VideoPatchEmbed
            self.x_embedder = VideoPatchEmbed(input_size, patch_size, in_channels, encoder_hidden_size, bias=True, frame_num=frame_num)
        else:
            self.x_embedder = PatchEmbed3D(input_size, frame_num, patch_size, temporal_patch_size, in_channels, encoder_hidden_size, bias=True)
        self.token_h = token_h = self.token_w = token_w = int(self.x_embedder.num_spatial_patches ** 0.5)
        self.token_t = token_t = self.x_embedder.num_temporal_patches
        self.video_token_num = video_token_num = self.x_embedder.num_spatial_patches * token_t
        assert input_size % decoder_patch_size == 0, "input_size must be divisible by decoder_patch_size"
        self.decoder_token_t = decoder_token_t = frame_num // decoder_temporal_patch_size
        self.decoder_token_h = self.decoder_token_w = decoder_token_h = input_size // decoder_patch_size
        self.recon_video_token_num = recon_video_token_num = self.decoder_token_h ** 2 * self.decoder_token_t
        E, Dd = encoder_hidden_size, decoder_hidden_size

        # the position / query / token-type embeddings of larp_tokenizer.py:119-180, under the reference's names
        self.learned_encoder_patch_pe = learned_encoder_patch_pe
        if learned_encoder_patch_pe:
            self.encoder_h_embed = nn.Parameter(torch.zeros(1, 1, token_h, 1, E), requires_grad=True)
            self.encode_w_embed = nn.Parameter(torch.zeros(1, 1, 1, token_w, E), requires_grad=True)
            self.encoder_t_embed = nn.Parameter(torch.zeros(1, token_t, 1, 1, E), requires_grad=True)
        else:
            self.register_buffer("encoder_patch_pe", torch.zeros(1, video_token_num, E))
        self.use_encoder_patch_token_type_embed = use_encoder_patch_token_type_embed
        if use_encoder_patch_token_type_embed:
            self.encoder_patch_token_type_embed = nn.Parameter(torch.zeros(1, 1, E), requires_grad=True)
        self.learned_encoder_latent_query_embed = learned_encoder_latent_query_embed
        self.encoder_query_gaussian_init = encoder_query_gaussian_init
        if learned_encoder_latent_query_embed:
            self.encoder_latent_query_embed = nn.Parameter(torch.zeros(bottleneck_token_num, E), requires_grad=True)
        else:
            self.register_buffer("encoder_latent_query_embed", torch.zeros(bottleneck_token_num, E))
            assert not encoder_query_gaussian_init, "encoder_query_gaussian_init requires learned_encoder_latent_query_embed to be True"
        self.use_encoder_latent_query_token_type_embed = use_encoder_latent_query_token_type_embed
        if use_encoder_latent_query_token_type_embed:
            self.encoder_latent_query_token_type_embed = nn.Parameter(torch.zeros(1, 1, E), requires_grad=True)
        self.learned_decoder_latent_pe = learned_decoder_latent_pe
        if learned_decoder_latent_pe:
            self.decoder_latent_pe = nn.Parameter(torch.zeros(1, self.decoder_latent_len, Dd), requires_grad=True)
        else:
            self.register_buffer("decoder_latent_pe", torch.zeros(1, self.decoder_latent_len, Dd))
        self.use_decoder_latent_token_type_embed = use_decoder_latent_token_type_embed
        if use_decoder_latent_token_type_embed:
            self.decoder_latent_token_type_embed = nn.Parameter(torch.zeros(1, 1, Dd), requires_grad=True)
        self.learned_decoder_patch_query_embed = learned_decoder_patch_query_embed
        if learned_decoder_patch_query_embed:
            self.decoder_h_embed = nn.Parameter(torch.zeros(1, 1, decoder_token_h, 1, Dd), requires_grad=True)
            self.decoder_w_embed = nn.Parameter(torch.zeros(1, 1, 1, decoder_token_h, Dd), requires_grad=True)
            self.decoder_t_embed = nn.Parameter(torch.zeros(1, decoder_token_t, 1, 1, Dd), requires_grad=True)
        else:
            self.register_buffer("decoder_patch_query_embed", torch.zeros(1, recon_video_token_num, Dd))
        self.use_decoder_patch_query_token_type_embed = use_decoder_patch_query_token_type_embed
        if use_decoder_patch_query_token_type_embed:
            self.decoder_patch_query_token_type_embed = nn.Parameter(torch.zeros(1, 1, Dd), requires_grad=True)

        def _name(n):
            return transformer_name if n is None or str(n).lower() in ("none", "no", "null", "") else n
        enc_args = {"name": _name(encoder_name), "args": {"dim": encoder_hidden_size, "depth": encoder_depth, "n_head": encoder_num_heads,
                                                          "head_dim": encoder_hidden_size // encoder_num_heads}}
        dec_args = {"name": _name(decoder_name), "args": {"dim": decoder_hidden_size, "depth": decoder_depth, "n_head": decoder_num_heads,
                                                          "head_dim": decoder_hidden_size // decoder_num_heads}}
        self.encoder = registry.make(enc_args)
        self.decoder = registry.make(dec_args)
        if train_type == "mrope":
            # larp_tokenizer.py:242-244, 401-405, 459-461: encode / decode go through Encoder111 / Decoder111 (gated layers with 3-axis RoPE,
            # model_new/base/blocks.py:1110-1178, built with THEIR defaults: 'small' = width 512, a 16x128x128 clip in 4x8x8 patches, 1024
            # latents -- the tokenizer's hidden size, patching and token count have to agree with that, as in the reference); the plain
            # encoder / decoder above stay constructed and in the state dict, unused, as they do there.  `mrope_args` (this build only)
            # overrides those defaults, e.g. for a small test geometry.
            from .titok import Decoder111, Encoder111
            ma = dict(mrope_args or {})
            self.encoder111 = Encoder111(**{k: v for k, v in ma.items() if k in ("model_size", "patch_size", "in_grid", "out_tokens")})
            self.decoder111 = Decoder111(**{{"in_grid": "out_grid", "out_tokens": "in_tokens"}.get(k, k): v for k, v in ma.items()
                                            if k in ("model_size", "patch_size", "in_grid", "out_tokens")})

        if bottleneck_type == "vq":
            self.bottleneck_dim = bottleneck["args"]["bottleneck_dim"]
            self.bottleneck = registry.make(bottleneck, args={"token_nums": self.bottleneck_token_num, "input_dim": encoder_hidden_size,
                                                              "output_dim": decoder_hidden_size})
            self.codebook_size = bottleneck["args"]["regularizer"]["args"]["codebook_size"]
            # parameter names the fused engine binds (engine.py): in/out projection, codebook
            self._bt_names = {"in_w": "bottleneck.in_linear.weight", "in_b": "bottleneck.in_linear.bias", "out_w": "bottleneck.out_linear.weight",
                              "out_b": "bottleneck.out_linear.bias", "codebook": "bottleneck.regularizer.embedding.weight"}
        elif bottleneck_type == "fsq":
            # larp_tokenizer.py:219-228, 412-418: LayerNorm -> Linear(768, 6) -> FSQ([8, 8, 8, 5, 5, 5]) -> Linear(6, 768); `encode` returns
            # {'encoded'} only (the FSQ indices are dropped there).  Not on the fused engine: forward() composes the sub-modules' own
            # autograd functions (one vt_stack_forward / backward call per stack, the LayerNorm / GEMM / FSQ / patch kernels in between).
            from .fsq import FSQ
            self.fsq_in_linear = nn.Linear(encoder_hidden_size, 6)
            self.fsq_out_linear = nn.Linear(6, decoder_hidden_size)
            self.fsq_norm = nn.LayerNorm(encoder_hidden_size)
            self.bottleneck = FSQ(levels=[8, 8, 8, 5, 5, 5])
            self.bottleneck_dim, self.codebook_size = 6, self.bottleneck.codebook_size
            self._bt_names = None
        else:
            # larp_tokenizer.py:225-229: Linear(768, 24) -> VectorQuantizer(196 560 x 24, frozen, cosine) -> Linear(24, 768).  The yaml's
            # `bottleneck` entry is ignored on this branch, as in the reference.  `sq_codebook` (extra keyword of this build): a .npy
            # path or an array; default = the reference's default path, replaced by the generated Leech shell when the file is absent.
            from .sq import VectorQuantizer
            self.sq_in_linear = nn.Linear(encoder_hidden_size, 24)
            self.sq_out_linear = nn.Linear(24, decoder_hidden_size)
            cb = ignored.get("sq_codebook", "/data2/zhxie/myproject/bsq-vit/cache/leech_lattices_normalized.npy")
            self.bottleneck = VectorQuantizer(n_embed=196_560, embed_dim=24, l2_norm=True, beta=0.25, input_format="blc", predefined_codebook=cb)
            self.bottleneck_dim, self.codebook_size = 24, 196_560
            self._bt_names = {"in_w": "sq_in_linear.weight", "in_b": "sq_in_linear.bias", "out_w": "sq_out_linear.weight",
                              "out_b": "sq_out_linear.bias", "codebook": "bottleneck.embedding.weight"}
        self.final_layer = OutputLayer(decoder_hidden_size, decoder_temporal_patch_size, decoder_patch_size, self.out_channels)
        self.prior_model = None  # the reference never builds one (larp_tokenizer.py:239-241); the trainer reads the attribute
        self.initialize_weights()
        self._engine = None if self._composed else _engine.TokenizerEngine(self)

    # ------------------------------------------------------------------------------- init
    def initialize_weights(self):
        """larp_tokenizer.py:249-328: xavier-uniform Linears (and the conv viewed 2-D), zero biases, sin-cos
        buffers, N(0, std^2) latent queries / token-type embed, ZERO output layer."""
        def _basic_init(module):
            if isinstance(module, nn.Linear):
                torch.nn.init.xavier_uniform_(module.weight)
                if module.bias is not None:
                    nn.init.constant_(module.bias, 0)
        self.apply(_basic_init)
        D, Dd = self.encoder_hidden_size, self.decoder_hidden_size

        def tab(dim, n, scale=10000):
            return torch.from_numpy(get_1d_sincos_pos_embed_from_grid(dim, np.arange(n), scale)).float()
        if self.learned_encoder_patch_pe:                                                        # :258-264
            self.encoder_h_embed.data.copy_(tab(D, self.token_h).reshape_as(self.encoder_h_embed))
            self.encode_w_embed.data.copy_(tab(D, self.token_w).reshape_as(self.encode_w_embed))
            self.encoder_t_embed.data.copy_(tab(D, self.token_t).reshape_as(self.encoder_t_embed))
        else:
            pe = get_3d_sincos_pos_embed(D, self.token_h, self.token_t)
            self.encoder_patch_pe.data.copy_(torch.from_numpy(pe).float().reshape_as(self.encoder_patch_pe))
        if self.use_encoder_patch_token_type_embed:
            self.encoder_patch_token_type_embed.data.copy_(torch.randn(1, 1, D) * 0.02)
        if self.learned_encoder_latent_query_embed:                                              # :273-285
            q = torch.randn(self.bottleneck_token_num, D) * self.query_init_std if self.encoder_query_gaussian_init else tab(D, self.bottleneck_token_num)
        else:
            q = tab(D, self.bottleneck_token_num, self.latent_pe_scale_factor)
        self.encoder_latent_query_embed.data.copy_(q)
        if self.use_encoder_latent_query_token_type_embed:
            self.encoder_latent_query_token_type_embed.data.copy_(torch.randn(1, 1, D) * 0.02)
        if self.learned_decoder_latent_pe:                                                       # :291-297
            self.decoder_latent_pe.data.copy_(torch.randn(1, self.decoder_latent_len, Dd) * 0.02)
        else:
            self.decoder_latent_pe.data.copy_(tab(Dd, self.decoder_latent_len, self.latent_pe_scale_factor).reshape_as(self.decoder_latent_pe))
        if self.use_decoder_latent_token_type_embed:
            self.decoder_latent_token_type_embed.data.copy_(torch.randn(1, 1, Dd) * 0.02)
        if self.learned_decoder_patch_query_embed:                                               # :303-309
            self.decoder_h_embed.data.copy_(tab(Dd, self.decoder_token_h).reshape_as(self.decoder_h_embed))
            self.decoder_w_embed.data.copy_(tab(Dd, self.decoder_token_w).reshape_as(self.decoder_w_embed))
            self.decoder_t_embed.data.copy_(tab(Dd, self.decoder_token_t).reshape_as(self.decoder_t_embed))
        else:
            dq = get_3d_sincos_pos_embed(Dd, self.decoder_token_h, self.decoder_token_t)
            self.decoder_patch_query_embed.data.copy_(torch.from_numpy(dq).float().reshape_as(self.decoder_patch_query_embed))
        if self.use_decoder_patch_query_token_type_embed:
            self.decoder_patch_query_token_type_embed.data.copy_(torch.randn(1, 1, Dd) * 0.02)
        w = self.x_embedder.proj.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.constant_(self.x_embedder.proj.bias, 0)
        nn.init.constant_(self.final_layer.linear.weight, 0)
        nn.init.constant_(self.final_layer.linear.bias, 0)

    # ------------------------------------------------------------------------------- embeddings (larp_tokenizer.py:119-180)
    def get_encoder_patch_pe(self):
        pe = (self.encoder_h_embed + self.encode_w_embed + self.encoder_t_embed).reshape(1, self.video_token_num, self.encoder_hidden_size) \
            if self.learned_encoder_patch_pe else self.encoder_patch_pe
        return pe + self.encoder_patch_token_type_embed if self.use_encoder_patch_token_type_embed else pe

    def get_encoder_latent_query_embed(self):
        q = self.encoder_latent_query_embed.unsqueeze(0)
        return q + self.encoder_latent_query_token_type_embed if self.use_encoder_latent_query_token_type_embed else q

    def get_decoder_latent_pe(self):
        return self.decoder_latent_pe + self.decoder_latent_token_type_embed if self.use_decoder_latent_token_type_embed else self.decoder_latent_pe

    def get_decoder_patch_query_embed(self):
        q = (self.decoder_h_embed + self.decoder_w_embed + self.decoder_t_embed).reshape(1, self.recon_video_token_num, self.decoder_hidden_size) \
            if self.learned_decoder_patch_query_embed else self.decoder_patch_query_embed
        return q + self.decoder_patch_query_token_type_embed if self.use_decoder_patch_query_token_type_embed else q

    # ------------------------------------------------------------------------------- small API
    def get_last_layer(self):
        return self.final_layer.linear.weight

    def set_vq_eval_deterministic(self, deterministic=True):
        if self.bottleneck_type == "vq":
            self.bottleneck.regularizer.set_eval_deterministic(deterministic)

    def _vq_engine_cfg(self):
        """(index mode, l2_normalized, 1/tau, beta, codebook weight, frozen codebook) for the fused engine"""
        if self.bottleneck_type == "vq":
            vq = self.bottleneck.regularizer
            return vq.index_mode(), bool(vq.l2_normalized), vq.inv_tau(), float(vq.beta), float(vq.codebook_loss_weight), False
        # 'sq': argmin(-z E^T) == first argmax of the cosine in every mode (model_new/quantizer/fsq.py:176-183), frozen codebook
        return 1, True, 1.0, float(self.bottleneck.beta), 1.0, True

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def decoder_parameters(self):
        return itertools.chain(self.decoder.parameters(), self.final_layer.parameters())

    def decoder_requires_grad_(self, requires_grad):
        for p in self.decoder_parameters():
            p.requires_grad_(requires_grad)

    def others_parameters(self):
        dec = set(self.decoder_parameters())
        return (p for p in self.parameters() if p not in dec)

    def others_requires_grad_(self, requires_grad):
        for p in self.others_parameters():
            p.requires_grad_(requires_grad)

    def unpatchify(self, x):
        """larp_tokenizer.py:441-454: (b, n, pt * p * p * c) rows in the reference's (pt, p1, p2, c) column order -> video (b, c, t pt, h p, w p),
        with the ENCODER-side patch sizes and token_h as the reference uses.  A view + permute of torch (any device); the engine itself
        never calls it -- its head GEMM runs on row-permuted weights and scatters straight into the video (csrc/vt_patch.hip)."""
        c, pt, p = self.out_channels, self.temporal_patch_size, self.patch_size
        h = w = self.token_h
        t = x.size(1) // (h * w)
        x = x.reshape(-1, t, h, w, pt, p, p, c)
        return x.permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(x.shape[0], c, t * pt, h * p, w * p)

    @classmethod
    def from_checkpoint(cls, ckpt, load_state_dict=True, version="sd"):
        """larp_tokenizer.py:376-398.  Files are read with weights_only=True (nothing is executed from them)."""
        if isinstance(ckpt, str):
            assert os.path.exists(ckpt), f"checkpoint {ckpt} does not exist"
            ckpt = torch.load(ckpt, map_location="cpu", weights_only=True)
        else:
            assert isinstance(ckpt, dict), "checkpoint must be a dict or a path to a checkpoint"
        model = cls(**ckpt["model"]["args"])
        if load_state_dict:
            if version == "sd":
                sd = ckpt["model"]["sd"]
            elif version.startswith("ema"):
                assert "_" in version, "ema version must be in the format 'ema_{alpha}'"
                sd = ckpt["model"]["ema_sd"][float(version.split("_")[1])]
            else:
                raise ValueError(f"Unknown version: {version}")
            model.load_state_dict(sd, strict=True)
        return model

    # ------------------------------------------------------------------------------- hot path
    def _bottleneck_dict(self, o):
        if self.bottleneck_type == "sq":
            # fsq.py:195-206: loss = beta * mean_n sum_d (sg(q) - z)^2 + mean_n sum_d (q - sg(z))^2 = d * (engine loss_q); only key besides
            # 'encoded' (larp_tokenizer.py:423-428)
            return {"loss_codebook": o["losses"][0] * float(self.bottleneck_dim)}
        zero = torch.zeros((), device=o["losses"].device)
        return {
            "bottleneck_rep": o["indices"], "projected_z": o["projected_z"],
            "input_norm_first": o["input_norms"][0], "input_norm_last": o["input_norms"][1],
            "unregularized_z": o["unregularized_z"], "emb": o["emb"], "regularized_z": o["regularized_z"],
            "loss_q": o["losses"][0], "loss_commit": o["losses"][1], "loss_codebook": o["losses"][2],
            "loss_entropy": zero, "per_sample_entropy": zero, "codebook_entropy": zero,
        }

    def forward(self, data, **kwargs):
        """larp_tokenizer.py:489-496: {'pred_frames', 'encoded', **bottleneck outputs}.  Differentiable outputs:
        pred_frames, loss_q, loss_commit, loss_codebook (what the trainer back-propagates,
        trainers/larp_tokenizer_trainer.py:294-333,372)."""
        if self._composed:
            enc = self._composed_encode(data)
            return {"pred_frames": self._composed_decode(enc["encoded"]).contiguous(), **enc}
        pred, losses, encoded, idx, pz, uz, rz, emb, norms = _engine.apply(self._engine, data)
        o = {"indices": idx, "projected_z": pz, "input_norms": norms, "unregularized_z": uz, "emb": emb, "regularized_z": rz, "losses": losses}
        if self.bottleneck_type == "sq":
            self.last_indices = idx     # the reference's 'sq' dict carries no token ids (fsq.py:206); kept here for inspection / tests
        return {"pred_frames": pred, "encoded": encoded, **self._bottleneck_dict(o)}

    # ---- the composed path (self._composed): the sub-modules' own autograd functions, differentiable end to end, also through encode / decode
    def _composed_encode(self, x, num_tokens_only=False):
        """larp_tokenizer.py:400-428"""
        from .functional import LayerNormRows, Linear
        if not x.is_cuda:
            raise hip.HipError("LARPTokenizer: input is on the CPU; this build runs on MI355X only (no CPU fallback)")
        B = x.shape[0]
        nv = (x.shape[2] // self.temporal_patch_size) * (x.shape[3] // self.patch_size) ** 2
        pe = self.get_encoder_patch_pe()[:, :nv]
        if self.train_type == "mrope":            # :401-405: no additive position embedding, the layers rotate q and k
            z = self.encoder111(self.x_embedder(x), self.get_encoder_latent_query_embed().expand(B, -1, -1))
        else:
            if pe.requires_grad:                  # learned / token-type embeddings: the add carries their gradient
                tok = self.x_embedder(x) + pe
            else:
                tok = self.x_embedder(x, pos_embed=pe[0])
            z = self.encoder(tok, self.get_encoder_latent_query_embed().expand(B, -1, -1))
        if self.bottleneck_type == "fsq":
            z = LayerNormRows.apply(z, self.fsq_norm.weight, self.fsq_norm.bias, self.fsq_norm.eps)
            z = Linear.apply(z, self.fsq_in_linear.weight, self.fsq_in_linear.bias)
            codes, info = self.bottleneck(z)
            self.last_indices, self.last_codes = info["indices"], codes.detach()     # the reference drops them (:416); kept for inspection / tests
            return {"encoded": Linear.apply(codes, self.fsq_out_linear.weight, self.fsq_out_linear.bias)}
        if self.bottleneck_type == "sq":
            o = self.bottleneck(Linear.apply(z, self.sq_in_linear.weight, self.sq_in_linear.bias))
            self.last_indices = o.pop("indices")
            return {"encoded": Linear.apply(o.pop("output"), self.sq_out_linear.weight, self.sq_out_linear.bias), **o}
        o = self.bottleneck(z)
        return {"encoded": o.pop("output"), **o}

    def _head_perm(self, device):
        """head rows in the patch scatter's (c, dt, dy, dx) order <- the reference's (dt, dy, dx, c) (larp_tokenizer.py:452-453)"""
        hit = getattr(self, "_head_perm_cache", None)
        if hit is None or hit.device != device:
            C, pt, p = self.out_channels, self.decoder_temporal_patch_size, self.decoder_patch_size
            c, dt, dy, dx = torch.meshgrid(torch.arange(C), torch.arange(pt), torch.arange(p), torch.arange(p), indexing="ij")
            hit = ((((dt * p + dy) * p + dx) * C + c).reshape(-1)).to(device)
            self._head_perm_cache = hit
        return hit

    def _composed_decode(self, z, num_x_tokens=None):
        """larp_tokenizer.py:456-469 (:471-482 with fewer query tokens)"""
        from .functional import LayerNormRows, Linear, Unpatchify
        if not z.is_cuda:
            raise hip.HipError("LARPTokenizer.decode: input is on the CPU; no CPU fallback")
        B = z.shape[0]
        nv = self.recon_video_token_num if num_x_tokens is None else int(num_x_tokens)
        dq = self.get_decoder_patch_query_embed()[:, :nv]
        if self.train_type == "mrope":            # :459-461: no latent position embedding either
            h = self.decoder111(z.float(), dq.expand(B, -1, -1))
        else:
            h = self.decoder(z.float() + self.get_decoder_latent_pe(), dq.expand(B, -1, -1))
        fl = self.final_layer
        y = LayerNormRows.apply(h, fl.norm_final.weight, fl.norm_final.bias, fl.norm_final.eps)
        perm = self._head_perm(z.device)
        rows = Linear.apply(y, fl.linear.weight[perm], fl.linear.bias[perm])
        T = nv // self.decoder_token_h ** 2 * self.decoder_temporal_patch_size
        return Unpatchify.apply(rows.reshape(B * nv, -1), (B, self.out_channels, T, self.input_size, self.decoder_temporal_patch_size, self.decoder_patch_size))

    def _graph_expected(self, params, *tensors):
        """True when the caller is recording an autograd graph that this call belongs to: grad mode on and an input or one of
        the parameters the call uses requires a gradient"""
        return torch.is_grad_enabled() and (any(t.requires_grad for t in tensors if torch.is_tensor(t)) or any(p.requires_grad for p in params))

    def encode(self, x):
        """larp_tokenizer.py:400-428: an ordinary differentiable method in the reference.  Forward-only calls (no_grad, eval
        pipelines) run the fused engine; when a graph is being recorded the same kernels run through the sub-modules' autograd
        functions (the composed path), so `encode(x)['encoded']`, the losses and every encoder / bottleneck parameter get gradients."""
        if self._composed or self._graph_expected(self.others_parameters(), x):
            return self._composed_encode(x)
        with torch.no_grad():
            return self._encode(x)

    def _encode(self, x):
        _, _, o = _engine.run_encode(self._engine, x)
        return {"encoded": o["encoded"], **self._bottleneck_dict(o)}

    @torch.no_grad()
    def encode_eval(self, x):
        """larp_tokenizer.py:430-439: may encode fewer frames (PE prefix), returns num_x_tokens."""
        out = self.encode(x)
        _, _, T, S, _ = x.shape
        out["num_x_tokens"] = (T // self.temporal_patch_size) * (S // self.patch_size) ** 2
        return out

    def decode(self, z, num_x_tokens=None):
        """larp_tokenizer.py:456-469 / :471-482 (decode_eval): z (b, Nq, D) -> video.  Differentiable like the reference's (decoder-only
        fine-tuning on cached latents, gradients w.r.t. z): with a graph being recorded the composed path runs, otherwise the engine."""
        extra = [self.decoder_patch_query_token_type_embed] if self.use_decoder_patch_query_token_type_embed else []
        if self._composed or self._graph_expected(itertools.chain(self.decoder_parameters(), extra), z):
            return self._composed_decode(z, num_x_tokens)
        with torch.no_grad():
            return self._decode(z, num_x_tokens)

    def _decode(self, z, num_x_tokens=None):
        if not z.is_cuda:
            raise hip.HipError("LARPTokenizer.decode: input is on the CPU; no CPU fallback")
        eng = self._engine
        B = z.shape[0]
        per_frame = self.decoder_token_h ** 2
        nv = self.recon_video_token_num if num_x_tokens is None else int(num_x_tokens)
        assert nv % per_frame == 0
        T = (nv // per_frame) * self.temporal_patch_size
        st = eng.state_for(B, T, self.input_size, z.device)
        ps = eng.param_struct()
        eng.ensure_packed(st, ps)
        st.fwd_id += 1
        return _engine.run_decode(eng, st, ps, z.contiguous().float(), B, T, self.input_size)

    def decode_eval(self, z, num_x_tokens=None):
        return self.decode(z, num_x_tokens)

    @torch.no_grad()
    def decode_from_bottleneck(self, bottleneck_rep):
        """larp_tokenizer.py:484-487: indices (b, Nq) -> bottleneck.decode -> decode.  ('sq': the reference's VectorQuantizer has
        no .decode and fails there with AttributeError; here the same entry point works -- codebook row -> sq_out_linear.)"""
        import ctypes
        if not bottleneck_rep.is_cuda:
            raise hip.HipError("LARPTokenizer.decode_from_bottleneck: input is on the CPU; no CPU fallback")
        if self._composed:
            from .functional import Linear
            if self.bottleneck_type == "fsq":    # (the reference's FSQ has no .decode; here: indices -> codes -> fsq_out_linear -> decode)
                codes = self.bottleneck.indices_to_codes(bottleneck_rep.contiguous().to(torch.int32))
                return self._composed_decode(Linear.apply(codes, self.fsq_out_linear.weight, self.fsq_out_linear.bias))
            if self.bottleneck_type == "sq":
                q = self.bottleneck.get_codebook_entry(bottleneck_rep)
                return self._composed_decode(Linear.apply(q, self.sq_out_linear.weight, self.sq_out_linear.bias))
            return self._composed_decode(self.bottleneck.decode(bottleneck_rep))
        eng = self._engine
        B = bottleneck_rep.shape[0]
        st = eng.state_for(B, self.frame_num, self.input_size, bottleneck_rep.device)
        ps = eng.param_struct()
        eng.ensure_packed(st, ps)
        ids = bottleneck_rep.contiguous().to(torch.int64)
        enc = torch.empty(B, self.bottleneck_token_num, self.decoder_hidden_size, device=ids.device, dtype=torch.float32)
        hip.check(hip.lib().vt_tokenizer_codes_to_encoded(st.handle, ctypes.byref(ps.struct), hip.ptr(ids), hip.ptr(st.ws), hip.ptr(enc), hip.stream()),
                  "vt_tokenizer_codes_to_encoded")
        return self.decode(enc)
