"""Bottleneck + vector quantiser: parameter holders and option surface.

Mirrors /root/reference/models/bottleneck.py: `Bottleneck` (:65-188; in_linear -> regulariser ->
out_linear; norm 'none' on the fused engine, the LayerNorm / SyncBatchNorm variants as torch glue on the composed path) and `SimpleVectorQuantizer` (:203-344; l2-normalised cosine / L2
codebook search in three index modes).  State-dict keys: `in_linear.*`, `out_linear.*`,
`regularizer.embedding.weight`.  Arithmetic: vt_vq_forward / vt_vq_backward + GEMMs -- fused inside the
engine when called through LARPTokenizer, through functional.{Linear,VectorQuantize} when the modules are
called on their own (forward / decode / get_codebook_entry keep the reference's signatures and dict keys).
"""
import torch
import torch.nn as nn

from .registry import make, register


@register("vq")
class SimpleVectorQuantizer(nn.Module):
    def __init__(self, dim, codebook_size, commitment_loss_weight=0.25, entropy_loss_weight=0.0,
                 entropy_loss_temperature=0.01, l2_normalized=False, same_index_shape=True, stochastic=False,
                 stochastic_temperature=1.0, codebook_loss_weight=1.0, **kwargs):
        super().__init__()
        self.codebook_size = codebook_size
        self.dim = dim
        self.beta = commitment_loss_weight
        self.codebook_loss_weight = codebook_loss_weight
        self.entropy_loss_weight = entropy_loss_weight
        if entropy_loss_weight > 0 and stochastic:
            # the reference computes the term from `d`, which only exists on the stochastic=False branch (bottleneck.py:282-303): NameError there
            raise ValueError("entropy_loss_weight > 0 needs stochastic=False (the reference's entropy term reads the L2 distances of that branch)")
        assert isinstance(l2_normalized, bool)
        self.l2_normalized = l2_normalized
        self.stochastic = stochastic
        self.eval_deterministic = False
        self.default_stochastic_temperature = stochastic_temperature
        if self.stochastic:
            assert self.l2_normalized, "Stochastic sampling requires l2 normalization"
            if stochastic_temperature > 0:
                self.stochastic_temperature_inv = 1 / stochastic_temperature
            else:
                raise NotImplementedError("learnable stochastic temperature is not built")
        self.embedding = nn.Embedding(self.codebook_size, self.dim)
        nn.init.kaiming_uniform_(self.embedding.weight)
        self.same_index_shape = same_index_shape
        self.entropy_loss_temperature = entropy_loss_temperature

    def set_eval_deterministic(self, deterministic=True):
        self.eval_deterministic = deterministic

    def set_stochastic_temperature(self, temperature):
        self.stochastic_temperature_inv = 1 / temperature

    def index_mode(self):
        """engine vq_mode for the current flags (bottleneck.py:272-290): 0 l2-argmin, 1 cos-argmax, 2 cos-sample."""
        if not self.stochastic:
            return 0
        return 1 if (self.eval_deterministic and not self.training) else 2

    def inv_tau(self):
        return float(self.stochastic_temperature_inv) if self.stochastic else 1.0

    def forward(self, z):
        """bottleneck.py:262-324; same dict keys.  The search runs in fp32 on the f32 MFMA path (indices bit-exact
        against the oracle); stochastic sampling is Gumbel-max with a counter RNG seeded from torch.initial_seed()."""
        from .functional import VectorQuantize
        assert len(z.shape) == 3, "Input shape must be (batch, n_tokens, e_dim)"
        self._calls = getattr(self, "_calls", 0) + 1
        seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._calls) & 0xFFFFFFFFFFFFFFFF
        rz, idx, lq, lc, lcb, zn, emb = VectorQuantize.apply(z.float(), self.embedding.weight, self.index_mode(), self.l2_normalized,
                                                              self.inv_tau(), float(self.beta), float(self.codebook_loss_weight), seed)
        if self.same_index_shape:
            idx = idx.reshape(rz.shape[0], rz.shape[1])
        zero = torch.zeros((), device=z.device, dtype=torch.float32)
        le = se = ae = zero
        if self.entropy_loss_weight > 0:
            le, se, ae = self._entropy_loss(z.float())
            lq = lq + self.entropy_loss_weight * le
        return {"unregularized_z": zn, "emb": emb, "regularized_z": rz, "bottleneck_rep": idx, "loss_q": lq, "loss_commit": lc,
                "loss_codebook": lcb, "loss_entropy": le, "per_sample_entropy": se, "codebook_entropy": ae}

    def _entropy_loss(self, z):
        """bottleneck.py:12-33, 298-303: entropy of softmax(-d / T) per token minus the entropy of its batch average.  NOT a kernel of this
        build: no shipped yaml enables it (entropy_loss_weight 0.0), so the term is the reference's own unfused sequence of torch ops on the GPU
        (one library GEMM for the N x K distances, softmax, two reductions: ~1 GB of traffic at N = K = 8192), differentiable through torch
        autograd into z and the codebook next to the fused quantizer's gradients.  A tokenizer that enables it runs on the composed path."""
        import torch.nn.functional as F
        zf = z.reshape(-1, z.shape[-1])
        emb = self.embedding.weight
        if self.l2_normalized:
            zf, emb = F.normalize(zf, p=2, dim=-1), F.normalize(emb, p=2, dim=-1)
        d = zf.pow(2).sum(1, keepdim=True) + emb.pow(2).sum(1) - 2 * zf @ emb.t()
        flat = -d / self.entropy_loss_temperature
        probs = F.softmax(flat, dim=-1)
        log_probs = F.log_softmax(flat + 1e-5, dim=-1)
        avg_probs = probs.mean(dim=0)
        avg_entropy = -torch.sum(avg_probs * torch.log(avg_probs + 1e-5))
        sample_entropy = -torch.mean(torch.sum(probs * log_probs, dim=-1))
        return sample_entropy - avg_entropy, sample_entropy, avg_entropy

    def get_codebook_entry(self, indices, shape=None):
        """bottleneck.py:327-341"""
        from .functional import codebook_entries
        z_q = codebook_entries(indices, self.embedding.weight, self.l2_normalized)
        return z_q.reshape(shape) if shape is not None else z_q

    def decode(self, indices):
        return self.get_codebook_entry(indices)


@register("bottleneck")
class Bottleneck(nn.Module):
    def __init__(self, bottleneck_dim, input_dim, output_dim, token_nums, norm=None, regularizer=None, param_names=None):
        super().__init__()
        self.token_nums = token_nums
        self.param_names = param_names
        self.input_dim = input_dim
        self.output_dim = output_dim
        if bottleneck_dim <= 0:
            raise NotImplementedError("bottleneck_dim <= 0 (identity projections) is not built")
        self.bottleneck_dim = bottleneck_dim
        norm = None if norm is None or norm.lower() in ("no", "none") else norm.lower()
        if norm not in (None, "ln_d", "ln_nd", "ln_d_na", "bn_bn", "bn_b"):
            raise ValueError(f"Normalization type {norm} not supported")
        self.norm = norm
        if regularizer is None or regularizer["name"].lower() != "vq":
            raise NotImplementedError("only the 'vq' regularizer is built")
        self.project_dim = self.bottleneck_dim
        self.in_linear = nn.Linear(self.input_dim, self.project_dim)
        self.out_linear = nn.Linear(self.bottleneck_dim, self.output_dim)
        # bottleneck.py:113-126: LayerNorm over the d (or token x d) entries of the projected latents, fp32 with autocast off (:146-159).
        # A [B, Nq, d <= 64] tensor: torch's LayerNorm is the whole cost model here; the fused engine does not carry it, so a tokenizer
        # with a normalised bottleneck runs on the composed path (LARPTokenizer._composed).
        if norm == "ln_d":
            self.norm_layer = nn.LayerNorm(self.project_dim)
        elif norm == "ln_nd":
            self.norm_layer = nn.LayerNorm((self.token_nums, self.project_dim))
        elif norm == "ln_d_na":
            self.norm_layer = nn.LayerNorm(self.project_dim, elementwise_affine=False)
        elif norm == "bn_bn":      # bottleneck.py:115-116: one statistic per latent channel, over batch and tokens
            self.norm_layer = nn.SyncBatchNorm(self.project_dim)
        elif norm == "bn_b":       # :117-119: one statistic per (token, channel), over the batch only
            assert self.token_nums is not None, "num_tokens must be specified for batch normalization"
            self.norm_layer = nn.SyncBatchNorm(self.project_dim * self.token_nums)
        regularizer["args"]["dim"] = self.bottleneck_dim
        regularizer["args"]["token_nums"] = self.token_nums
        self.regularizer = make(regularizer)

    def project_in(self, x):
        from .functional import Linear
        assert len(x.shape) == 3, "Input shape must be (batch, n_tokens, e_dim)"
        z = Linear.apply(x, self.in_linear.weight, self.in_linear.bias)
        if self.norm in ("ln_d", "ln_nd"):          # (the reference applies no layer for 'ln_d_na' in project_in either, :146-159)
            z = self.norm_layer(z.float())
        elif self.norm == "bn_bn":                  # :149-152  b n d -> b d n, SyncBatchNorm over (b, n), back; fp32, autocast off.  torch's
            z = self.norm_layer(z.float().transpose(1, 2)).transpose(1, 2)      # module: batch statistics are all-reduced over the process group
        elif self.norm == "bn_b":                   # :153-156  b n d -> b (n d)
            z = self.norm_layer(z.float().reshape(z.shape[0], -1)).reshape(z.shape)
        return z

    def project_out(self, z_cat):
        from .functional import Linear
        return Linear.apply(z_cat, self.out_linear.weight, self.out_linear.bias)

    def decode(self, bottleneck_rep):
        return self.project_out(self.regularizer.decode(bottleneck_rep))

    def forward(self, x):
        """bottleneck.py:170-188; the two norm statistics stay 0-dim device tensors (no .item() sync)"""
        input_norm_first = torch.norm(x[:, 0, :].float(), dim=-1).mean().detach()
        input_norm_last = torch.norm(x[:, -1, :].float(), dim=-1).mean().detach()
        z = self.project_in(x)
        reg = self.regularizer(z)
        x_hat = self.project_out(reg["regularized_z"])
        rep = reg.pop("bottleneck_rep")
        return {"output": x_hat, "bottleneck_rep": rep, "projected_z": z, "input_norm_first": input_norm_first,
                "input_norm_last": input_norm_last, **reg}
