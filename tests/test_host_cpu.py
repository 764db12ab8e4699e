"""CPU tests of the host side: registry semantics, state-dict/ckpt layout, PE buffers, config surface,
C-ABI export list, loud failure without a GPU.  No compute kernels are called here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import video_tokenizer_amd as vt
from oracle import larp_oracle as O
from tests.test_model_gpu import spec_from_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_make_filters_kwargs_and_loads_sd():
    @vt.register("_probe")
    class Probe(torch.nn.Module):
        def __init__(self, a, b=2):
            super().__init__()
            self.a, self.b = a, b
            self.w = torch.nn.Parameter(torch.zeros(3))
    m = vt.make({"name": "_probe", "args": {"a": 1, "junk": 5}}, args={"b": 7})
    assert (m.a, m.b) == (1, 7)
    m2 = vt.make({"name": "_probe", "args": {"a": 1}, "sd": {"w": torch.ones(3)}}, load_sd=True)
    assert torch.equal(m2.w.data, torch.ones(3))
    assert {"larp_tokenizer", "transformer_encoder_parallel", "bottleneck", "vq"} <= set(vt.models)


def test_module_layout_matches_reference_state_dict():
    cfg = O.make_cfg("tiny")
    m = vt.make(spec_from_cfg(cfg, stochastic=True))
    sd, ref = m.state_dict(), O.init_state_dict(cfg)
    assert set(sd) == set(ref)
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    m.load_state_dict(ref, strict=True)
    # buffers are in the checkpoint: they must equal the reference-pinned sin-cos tables bit for bit
    assert torch.equal(vt.make(spec_from_cfg(cfg)).encoder_patch_pe, ref["encoder_patch_pe"])
    assert torch.equal(vt.make(spec_from_cfg(cfg)).decoder_latent_pe, ref["decoder_latent_pe"])
    assert torch.equal(vt.make(spec_from_cfg(cfg)).decoder_patch_query_embed, ref["decoder_patch_query_embed"])
    assert m.prior_model is None and m.output_format == "bcthw" and m.codebook_size == cfg["codebook_size"]
    assert m.bottleneck_token_num == cfg["bottleneck_token_num"] and m.x_embedder.strict_vid_size is True
    # zero-initialised head like the reference (larp_tokenizer.py:327-328)
    fresh = vt.make(spec_from_cfg(cfg))
    assert fresh.final_layer.linear.weight.abs().sum() == 0 and fresh.final_layer.norm_final.eps == 1e-6
    assert fresh.encoder.blocks[0].norm1.eps == 1e-5


def test_base_config_parameter_count():
    """config B: 173.4 M parameters (694 MB of fp32 gradients)."""
    m = vt.make(spec_from_cfg(O.make_cfg("B"), stochastic=True))
    n = sum(p.numel() for p in m.parameters())
    assert abs(n - 173.4e6) < 0.2e6, n


def test_from_checkpoint_roundtrip_and_extra_keys(tmp_path):
    cfg = O.make_cfg("tiny")
    spec = spec_from_cfg(cfg)
    m = vt.make(spec)
    m.load_state_dict(O.init_state_dict(cfg), strict=True)
    path = str(tmp_path / "ckpt.pth")
    torch.save({"model": {"name": "larp_tokenizer", "args": spec["args"], "sd": m.state_dict(), "ema_sd": {0.999: m.state_dict()}}, "epoch": 3}, path)
    a = vt.LARPTokenizer.from_checkpoint(path)                       # yaml extras such as use_pe are accepted and ignored
    b = vt.LARPTokenizer.from_checkpoint(path, version="ema_0.999")
    for k, v in m.state_dict().items():
        assert torch.equal(a.state_dict()[k], v) and torch.equal(b.state_dict()[k], v)
    with pytest.raises(ValueError):
        vt.LARPTokenizer.from_checkpoint(path, version="bogus")


def test_from_pretrained_local_directory_and_public_unpatchify(tmp_path):
    """larp_tokenizer.py:45 / larp_ar.py:233 (PyTorchModelHubMixin) as called at eval/eval_larp_tokenizer.py:40, sample.py:409,415,
    trainers/larp_ar_trainer.py:51: `from_pretrained(<local directory>)` reads config.json -> constructor kwargs and model.safetensors ->
    load_state_dict(strict=True); `save_pretrained` writes that layout.  A directory written by huggingface_hub's OWN mixin (installed
    here) loads too; a string that is no directory raises instead of reaching for the network.  unpatchify == the einops form (:441-454)."""
    import einops
    import json
    import os
    from video_tokenizer_amd import larp_ar
    cfg = O.make_cfg("tiny")
    spec = spec_from_cfg(cfg)
    m = vt.make(spec)
    m.load_state_dict(O.init_state_dict(cfg, seed=3), strict=True)
    d = m.save_pretrained(str(tmp_path / "tok"))
    assert sorted(os.listdir(d)) == ["config.json", "model.safetensors"]
    assert json.load(open(os.path.join(d, "config.json")))["bottleneck_token_num"] == cfg["bottleneck_token_num"]
    a = vt.LARPTokenizer.from_pretrained(d)
    assert not a.training and a.bottleneck_token_num == m.bottleneck_token_num
    for k, v in m.state_dict().items():
        assert torch.equal(a.state_dict()[k], v), k
    with pytest.raises(FileNotFoundError, match="never contacts"):
        vt.LARPTokenizer.from_pretrained("hywang66/LARP-L-long-tokenizer")
    # the AR prior: constructor takes a dataclass named `config`
    args = larp_ar.ModelArgs(dim=128, n_layer=2, n_head=2, vocab_size=64, max_seq_len=16, num_classes=5, cls_token_num=1)
    ar = larp_ar.LARP_AR(args)
    d2 = ar.save_pretrained(str(tmp_path / "ar"))
    assert json.load(open(os.path.join(d2, "config.json")))["n_layer"] == 2      # the hub mixin's encoding: the dataclass IS the file
    b = larp_ar.LARP_AR.from_pretrained(d2)
    assert b.config == args
    for k, v in ar.state_dict().items():
        assert torch.equal(b.state_dict()[k], v), k
    # a directory in the layout of the hub library itself
    from huggingface_hub import PyTorchModelHubMixin

    class HubTwin(torch.nn.Module, PyTorchModelHubMixin):
        def __init__(self, config: larp_ar.ModelArgs):
            super().__init__()
            self.inner = larp_ar.LARP_AR(config)

        def state_dict(self, *a, **k):
            return self.inner.state_dict(*a, **k)

    twin = HubTwin(args)
    twin.inner.load_state_dict(ar.state_dict())
    d3 = str(tmp_path / "hub")
    twin.save_pretrained(d3)
    c = larp_ar.LARP_AR.from_pretrained(d3)
    for k, v in ar.state_dict().items():
        assert torch.equal(c.state_dict()[k], v), k
    # unpatchify
    pt, p, hh = m.temporal_patch_size, m.patch_size, m.token_h
    x = torch.randn(2, m.token_t * hh * hh, pt * p * p * 3)
    ref = einops.rearrange(x.reshape(-1, m.token_t, hh, hh, pt, p, p, 3), "b t h w pt p1 p2 c -> b c (t pt) (h p1) (w p2)")
    assert torch.equal(m.unpatchify(x), ref)


def test_unsupported_options_fail_loudly():
    cfg = O.make_cfg("tiny")
    s = spec_from_cfg(cfg)
    s["args"]["bottleneck_type"] = "auto"
    with pytest.raises(NotImplementedError):
        vt.make(s)
    s = spec_from_cfg(cfg)
    s["args"]["train_type"] = "rope2d"
    with pytest.raises(NotImplementedError):
        vt.make(s)
    s = spec_from_cfg(cfg)
    s["args"]["bottleneck"]["args"]["norm"] = "bn_xx"
    with pytest.raises(ValueError):
        vt.make(s)
    for bn, width in (("bn_bn", cfg["bottleneck_dim"]), ("bn_b", cfg["bottleneck_dim"] * cfg["bottleneck_token_num"])):    # bottleneck.py:115-119
        s = spec_from_cfg(cfg)
        s["args"]["bottleneck"]["args"]["norm"] = bn
        m = vt.make(s)
        assert m._composed and isinstance(m.bottleneck.norm_layer, torch.nn.SyncBatchNorm) and m.bottleneck.norm_layer.num_features == width
        assert "bottleneck.norm_layer.running_mean" in m.state_dict()


def test_constructor_options_of_the_composed_path_keep_the_reference_state_dict():
    """larp_tokenizer.py:119-180: learned factorised PEs, token-type embeddings, fixed latent queries, VideoPatchEmbed -- the parameters /
    buffers carry the reference's names and shapes and the model leaves the fused engine for the composed path"""
    cfg = O.make_cfg("tiny")
    s = spec_from_cfg(cfg)
    s["args"].update(learned_encoder_patch_pe=True, use_encoder_patch_token_type_embed=True, use_encoder_latent_query_token_type_embed=True,
                     learned_decoder_latent_pe=True, use_decoder_latent_token_type_embed=True, learned_decoder_patch_query_embed=True)
    m = vt.make(s)
    sd = m.state_dict()
    th, tt, D = m.token_h, m.token_t, 768
    assert m._composed and m._engine is None
    assert sd["encoder_h_embed"].shape == (1, 1, th, 1, D) and sd["encode_w_embed"].shape == (1, 1, 1, th, D) and sd["encoder_t_embed"].shape == (1, tt, 1, 1, D)
    assert sd["decoder_h_embed"].shape == (1, 1, th, 1, D) and sd["decoder_t_embed"].shape == (1, tt, 1, 1, D)
    for k in ("encoder_patch_token_type_embed", "encoder_latent_query_token_type_embed", "decoder_latent_token_type_embed", "decoder_patch_query_token_type_embed"):
        assert sd[k].shape == (1, 1, D) and dict(m.named_parameters())[k].requires_grad
    assert "encoder_patch_pe" not in sd and "decoder_patch_query_embed" not in sd and dict(m.named_parameters())["decoder_latent_pe"].requires_grad
    # the learned factorised PE starts as the sum of three 1-D sin-cos tables (:258-264)
    from video_tokenizer_amd.embed import get_1d_sincos_pos_embed_from_grid
    h = get_1d_sincos_pos_embed_from_grid(D, np.arange(th))
    np.testing.assert_allclose(sd["encoder_h_embed"].reshape(th, D).numpy(), h, atol=1e-6)
    assert m.get_encoder_patch_pe().shape == (1, m.video_token_num, D) and m.get_decoder_patch_query_embed().shape == (1, m.recon_video_token_num, D)
    s = spec_from_cfg(cfg)
    s["args"].update(learned_encoder_latent_query_embed=False, encoder_query_gaussian_init=False, temporal_patch_size=1, decoder_temporal_patch_size=1)
    m = vt.make(s)
    assert "encoder_latent_query_embed" in dict(m.named_buffers()) and m.x_embedder.proj.weight.shape == (768, 3, cfg["patch_size"], cfg["patch_size"])
    assert m.token_t == cfg["frame_num"] and m._composed


def test_cpu_tensors_are_refused():
    cfg = O.make_cfg("tiny")
    m = vt.make(spec_from_cfg(cfg))
    x = torch.zeros(1, 3, cfg["frame_num"], cfg["input_size"], cfg["input_size"])
    for call in (lambda: m(x), lambda: m.encode(x), lambda: m.decode(torch.zeros(1, cfg["bottleneck_token_num"], 768)),
                 lambda: m.decode_from_bottleneck(torch.zeros(1, cfg["bottleneck_token_num"], dtype=torch.long))):
        with pytest.raises(vt.hip.HipError):
            call()
    # standalone sub-module forwards run the same kernels: CPU tensors are refused there too (no CPU path anywhere)
    tok = torch.zeros(1, 4, 768)
    for call in (lambda: m.encoder(tok, tok), lambda: m.bottleneck(tok), lambda: m.bottleneck.regularizer(torch.zeros(1, 4, 24)),
                 lambda: m.x_embedder(x), lambda: m.bottleneck.decode(torch.zeros(1, 4, dtype=torch.long))):
        with pytest.raises(vt.hip.HipError):
            call()
    d = vt.TransformerDiscriminator(128, 4, 1, 32, 2, 8, 3, frame_num=4)
    with pytest.raises(vt.hip.HipError):
        d(torch.zeros(1, 3, 4, 32, 32))
    q = vt.FSQ(levels=[8, 8, 8, 5, 5, 5])
    for call in (lambda: q(torch.zeros(2, 4, 6)), lambda: q.indices_to_codes(torch.zeros(2, 4, dtype=torch.int32))):
        with pytest.raises(vt.hip.HipError):
            call()


def test_titok_autoencoder_host_surface():
    """registry names, state-dict layout and rotary tables of the FSQ autoencoder family (model_new/autoencoder.py:8,89,589)"""
    from oracle import titok_oracle as T
    assert {"autoencoder_convpatchify", "autoencoder_convpatchify_greatfsq", "autoencoder_large"} <= set(vt.models)
    m = vt.make({"name": "autoencoder_convpatchify", "args": {"bottleneck": None, "prior_model": None, "input_size": 128, "encoder_depth": 6,
                                                               "_geometry": dict(in_grid=[8, 32, 32], tokens=32)}})
    cfg = T.make_cfg("small", frames=8, side=32, tokens=32)
    sd = T.init_state_dict(cfg)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    assert m.encoder.model_layers.ffd_layer[0][1].weight.shape == (2 * 1376, 512) and m.prior_model is None
    assert m.encoder.proj_out.bias.abs().sum() == 0 and float(m.encoder.model_layers.attn_layer[0].q_norm.weight.mean()) == 1.0
    cos, sin = vt.titok.rope_tables(1024, [4, 16, 16])
    ang = T.rope_angles(1024, [4, 16, 16], 64)
    assert cos.shape == (2048, 32) and torch.equal(cos, torch.cos(ang).float()) and torch.equal(sin, torch.sin(ang).float())
    assert vt.titok.get_model_dims("large") == (1024, 24, 16, 4.0) and vt.titok.get_model_dims("base_thin") == (1024, 7, 16, 2.0)
    assert vt.titok.ffd_inner_dim(1024) == 2752
    with pytest.raises(vt.hip.HipError):
        m(torch.zeros(2, 3, 8, 32, 32))                      # CPU tensors: no CPU path


def test_fsq_host_surface():
    """FSQ of models/model_new/quantizer/fsq.py:54-75: sizes, non-persistent buffers; the level list is validated by the
    library itself on the host (no GPU needed for that call)"""
    q = vt.FSQ(levels=[8, 8, 8, 8, 5, 5, 5, 5])
    assert q.codebook_size == 2560000 == vt.hip.fsq_codebook_size(q.levels) and q.dim == 8 and len(q.state_dict()) == 0
    assert q._basis.tolist() == [1, 8, 64, 512, 4096, 20480, 102400, 512000]
    for bad in ([8, 1], [8] * 9, [2] * 17):
        with pytest.raises(vt.hip.HipError):
            vt.hip.fsq_codebook_size(bad)


def test_discriminator_and_loss_module_surface():
    """state-dict keys / shapes of models/loss.py:119-204 and the registry names of SURVEY §8b"""
    assert {"transformer_encoder_fused", "transformer_encoder_parallel", "bottleneck", "vq", "larp_tokenizer", "lpips_disc_loss"} <= set(vt.models)
    d = vt.TransformerDiscriminator(384, 12, 8, 128, 4, 8, 3, frame_num=16)   # cfgs/larp_tokenizer.yaml:130-134
    sd = d.state_dict()
    assert sd["x_embedder.proj.weight"].shape == (384, 3, 4, 8, 8) and sd["cls_token"].shape == (1, 1, 384)
    assert sd["encoder_pos_embed"].shape == (1, 1024, 384) and sd["fc.weight"].shape == (1, 384)
    assert sd["transformer_encoder.blocks.7.mlp.fc2.weight"].shape == (384, 1536) and d.video_token_num == 1024
    ref = O.sincos_3d(384, 16, 4)
    assert np.array_equal(sd["encoder_pos_embed"][0].numpy(), ref.astype(np.float32))
    assert set(sd) == set(O.init_discriminator_state_dict(384, 12, 8, 128, 16, 4, 8))
    lm = vt.make({"name": "lpips_disc_loss", "args": {"disc_start": 0, "perceptual_weight": 0.0, "disc_loss": "ns_smooth", "lecam_weight": 0.001,
                                                      "disc_tran_hidden_size": 128, "disc_tran_n_heads": 4, "disc_tran_n_layers": 1,
                                                      "disc_tran_temporal_patch_size": 2, "disc_tran_patch_size": 8, "input_spatial_size": 32, "frame_num": 4}})
    assert [m_ for m_ in lm.trainable_modules()] == [lm.discriminator] and "lecam_ema_real" in lm.state_dict()
    per_frame = vt.TransformerDiscriminator(128, 4, 1, 32, 1, 8, 3, frame_num=4)        # temporal_patch_size 1: VideoPatchEmbed (loss.py:137-138)
    assert tuple(per_frame.x_embedder.proj.weight.shape) == (128, 3, 8, 8) and per_frame.video_token_num == 4 * 16


def test_vq_index_mode_follows_reference_flags():
    from video_tokenizer_amd.bottleneck import SimpleVectorQuantizer as VQ
    assert VQ(24, 16, l2_normalized=True, stochastic=False).index_mode() == 0
    v = VQ(24, 16, l2_normalized=True, stochastic=True, stochastic_temperature=0.03)
    assert v.index_mode() == 2 and abs(v.inv_tau() - 1 / 0.03) < 1e-9
    v.eval()
    assert v.index_mode() == 2          # eval without --det still samples (bottleneck.py:277-280)
    v.set_eval_deterministic(True)
    assert v.index_mode() == 1
    v.train()
    assert v.index_mode() == 2


YAML = """
trainer: larp_tokenizer_trainer
train_dataset:
  name: video_dataset
  args: {csv_file: $csv_file$, frame_num: $frame_num$, input_size: $input_size$}
model:
  name: some_other_model
  args:
    bottleneck:
      name: bottleneck
      args: {bottleneck_dim: 24, norm: 'none', regularizer: {name: vq, args: {codebook_size: 512, l2_normalized: true, stochastic: true, stochastic_temperature: 0.03}}}
    prior_model: {name: none}
    bottleneck_token_num: 56
    bottleneck_type: 'sq'
    input_size: 256
    frame_num: $frame_num$
    temporal_patch_size: 2
    patch_size: 16
    decoder_temporal_patch_size: 2
    decoder_patch_size: 16
    encoder_depth: 12
    decoder_depth: 12
    use_decoder_patch_query_token_type_embed: true
    use_pe: 'yes'
optimizer: {args: {betas: [0.5, 0.9]}}
"""


def test_yaml_surface_vars_and_opts():
    from video_tokenizer_amd.config import load_cfg
    args = {"csv_file": "null128", "frame_num": 4, "input_size": 32}
    cfg = load_cfg(YAML, args, ["model.name", "larp_tokenizer", "model.args.bottleneck_type", "vq", "model.args.input_size", "32",
                                "model.args.encoder_depth", "2", "model.args.decoder_depth", "2",
                                "model.args.use_decoder_patch_query_token_type_embed", "false", "optimizer.args.betas", "0.9_0.95"])
    assert cfg.train_dataset.args.csv_file == "null128" and cfg.model.args.frame_num == 4
    assert cfg.model.args.input_size == 32 and isinstance(cfg.model.args.input_size, int)
    assert cfg.model.args.use_decoder_patch_query_token_type_embed is False
    assert cfg.optimizer.args.betas == [0.9, 0.95]
    m = vt.make(cfg.model)
    assert m.encoder.depth == 2 and m.video_token_num == 8 and not m.use_decoder_patch_query_token_type_embed
    with pytest.raises(KeyError):
        load_cfg(YAML, args, ["model.args.not_a_key", "1"])


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = vt.hip.lib()  # loads libvt_hip.so and binds argtypes for every signature
    bound = set(vt.hip.SIGNATURES) | set(vt.hip.ENGINE_SIGNATURES)
    assert declared == bound, declared ^ bound
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.vt_abi_version() == 8
    # argument validation runs on the host before any launch: bad arguments give a code and a message, no GPU needed
    p = vt.hip.GemmNT()
    assert lib.vt_gemm_nt(ctypes.byref(p), None) == -1
    buf = ctypes.create_string_buffer(256)
    lib.vt_last_error(buf, 256)
    assert b"vt_gemm_nt" in buf.value
    h = ctypes.c_void_p()
    cfg = vt.hip.TokenizerConfig()
    assert lib.vt_tokenizer_create(ctypes.byref(cfg), ctypes.byref(h)) == -1


def test_engine_plan_sizes_host_only():
    """vt_tokenizer_create / workspace_bytes are pure host planning: config B at 8 clips/GPU fits easily in 288 GB."""
    lib = vt.hip.lib()
    c = vt.hip.TokenizerConfig()
    c.B, c.C, c.T, c.S, c.pt, c.p = 8, 3, 16, 128, 2, 16
    c.D, c.H, c.depth_enc, c.depth_dec, c.Nq, c.d, c.K = 768, 12, 12, 12, 1024, 24, 8192
    c.vq_mode, c.l2_normalized, c.inv_tau, c.beta, c.codebook_w = 2, 1, 33.3, 0.25, 1.0
    h = ctypes.c_void_p()
    assert lib.vt_tokenizer_create(ctypes.byref(c), ctypes.byref(h)) == 0
    nbytes = lib.vt_tokenizer_workspace_bytes(h)
    assert 8e9 < nbytes < 20e9, nbytes
    assert lib.vt_tokenizer_num_backward_stages(h) == 27
    lib.vt_tokenizer_destroy(h)


def test_frechet_distance_and_feature_stats():
    """metrics.py against scipy's matrix square root on synthetic Gaussian features (the I3D extractor is absent)."""
    import scipy.linalg
    from video_tokenizer_amd.metrics import FeatureStats, clip_mse, frechet_distance, psnr_given_mse
    rng = np.random.default_rng(0)
    A = rng.normal(size=(2000, 24)).astype(np.float32) @ rng.normal(size=(24, 24)).astype(np.float32)
    Bf = rng.normal(size=(1500, 24)).astype(np.float32) * 1.3 + 0.2
    sa, sb = FeatureStats(), FeatureStats(max_items=1200)
    for chunk in np.array_split(A, 7):
        sa.append(chunk)
    for chunk in np.array_split(Bf, 5):
        sb.append(torch.from_numpy(chunk))
    assert sb.num_items == 1200 and sb.is_full()
    mu_a, cov_a = sa.get_mean_cov()
    np.testing.assert_allclose(mu_a, A.astype(np.float64).mean(0), rtol=1e-10)
    np.testing.assert_allclose(cov_a, np.cov(A.astype(np.float64), rowvar=False, bias=True), rtol=1e-8, atol=1e-8)
    mu_b, cov_b = sb.get_mean_cov()
    covmean = scipy.linalg.sqrtm(cov_a @ cov_b).real
    want = ((mu_a - mu_b) ** 2).sum() + np.trace(cov_a) + np.trace(cov_b) - 2 * np.trace(covmean)
    np.testing.assert_allclose(frechet_distance(sa, sb), want, rtol=1e-6)
    assert abs(frechet_distance(sa, sa)) < 1e-6 * np.trace(cov_a)
    # the eigen-decomposition square root == the SVD form the reference uses (fvd.py:24-33), also on a rank-deficient covariance
    from video_tokenizer_amd.metrics import psd_sqrt, trace_sqrt_product

    def svd_sqrt(m, eps=1e-10):
        u, sv, vt_ = torch.linalg.svd(m)
        return u @ torch.diag(torch.where(sv < eps, sv, sv.sqrt())) @ vt_
    low = rng.normal(size=(10, 24))
    for c in (torch.from_numpy(cov_a), torch.from_numpy(cov_b), torch.from_numpy(low.T @ low / 10)):
        np.testing.assert_allclose(psd_sqrt(c).numpy(), svd_sqrt(c).numpy(), rtol=1e-7, atol=1e-9 * float(c.abs().max()) + 1e-9)
        other = torch.from_numpy(cov_b)
        want_tr = torch.trace(svd_sqrt(svd_sqrt(c) @ other @ svd_sqrt(c)))
        np.testing.assert_allclose(float(trace_sqrt_product(c, other)), float(want_tr), rtol=1e-8)
    with pytest.raises(ValueError):
        sa.update(np.zeros((3, 5), dtype=np.float32))
    v = torch.rand(3, 3, 4, 8, 8)
    r = v + 0.1
    mse = clip_mse(v, r)
    assert mse.shape == (3,) and abs(psnr_given_mse(torch.full((3,), 0.01)) - 20.0) < 1e-5


def test_model_deepcopy_and_requires_grad_helpers():
    """the reference trainer deep-copies the model for EMA and toggles requires_grad on parameter groups"""
    import copy
    cfg = O.make_cfg("tiny")
    m = vt.make(spec_from_cfg(cfg))
    c = copy.deepcopy(m)
    assert c._engine is not m._engine and c._engine.model is c
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), c.state_dict().values()))
    c.requires_grad_(False)
    assert all(not p.requires_grad for p in c.parameters()) and all(p.requires_grad for p in m.parameters())
    m.decoder_requires_grad_(False)
    dec = set(id(p) for p in m.decoder_parameters())
    assert all(p.requires_grad == (id(p) not in dec) for p in m.parameters())
    m.others_requires_grad_(False)
    assert not any(p.requires_grad for p in m.parameters())


def test_plain_c_client_links_against_the_c_abi(tmp_path):
    """gcc (not hipcc, not C++) compiles a client of include/vt_hip.h and links libvt_hip.so: the boundary is a C ABI."""
    import shutil
    import subprocess
    from video_tokenizer_amd import build as vt_build
    lib = vt_build.build()
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(lib)
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
           "-o", exe, "-L", libdir, "-lvt_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    assert shutil.which("gcc"), "gcc missing"
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "abi 8"
    assert lines[1].startswith("gemm_nt rc -1 msg vt_gemm_nt: null operand")
    assert lines[2].startswith("create rc 0 stages 5 ws ") and int(lines[2].split()[-1]) > 1 << 20
    assert lines[3].startswith("bad create rc -1") and "head_dim" in lines[3]
    assert lines[4].startswith("stack rc 0 ws ")


def test_product_geometry_table_and_synthetic_clips_equal_the_checkers():
    """bench.py takes its geometry table, registry spec and synthetic clips from video-tokenizer_amd/config.py; the oracle
    keeps its own copies (it must not import the product): both must describe the same configs and produce the same bytes."""
    from oracle import inputs as gen
    C = vt.config
    assert set(C.GEOMETRIES) == set(O.CONFIGS)
    for n in O.CONFIGS:
        assert C.geometry(n) == O.make_cfg(n), n
    assert C.geometry("tiny", bottleneck_token_num=37) == O.make_cfg("tiny", bottleneck_token_num=37)
    for (b, t, s, seed) in [(2, 4, 32, 11), (1, 2, 64, 3), (3, 16, 16, 100)]:
        assert np.array_equal(C.synthetic_clips(b, t, s, seed), gen.video_clips(b, t, s, seed))
    spec = C.model_spec(C.geometry("B"), stochastic=True)
    assert spec["name"] == "larp_tokenizer" and spec["args"]["bottleneck_type"] == "vq" and spec["args"]["bottleneck"]["args"]["regularizer"]["args"]["stochastic"] is True
    # bench.py touches oracle/ only inside its cpu_baseline() leg, and tests/ nowhere
    src = open(os.path.join(ROOT, "bench.py")).read()
    start, end = src.index("def cpu_baseline"), src.index("\ndef main")
    outside = src[:start] + src[end:]
    assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", outside, re.M)
    assert re.search(r"^\s*from oracle import", src[start:end], re.M)


def test_sq_bottleneck_host_surface():
    """LARPTokenizer(bottleneck_type='sq') (models/larp_tokenizer.py:225-229): parameter names and shapes of the reference's
    state dict, the frozen 196 560 x 24 codebook, and the engine's parameter binding; no GPU needed to construct it."""
    spec = vt.config.model_spec(vt.config.geometry("tiny"))
    spec["args"]["bottleneck_type"] = "sq"
    m = vt.make(spec)
    sd = m.state_dict()
    for k, shp in {"sq_in_linear.weight": (24, 768), "sq_in_linear.bias": (24,), "sq_out_linear.weight": (768, 24), "sq_out_linear.bias": (768,),
                   "bottleneck.embedding.weight": (196560, 24)}.items():
        assert tuple(sd[k].shape) == shp, k
    assert not any(k.startswith("bottleneck.in_linear") or "regularizer" in k for k in sd)
    assert m.bottleneck.embedding.weight.requires_grad is False and m.sq_in_linear.weight.requires_grad
    np.testing.assert_allclose(m.bottleneck.embedding.weight.norm(dim=-1).numpy(), 1.0, atol=1e-6)
    assert m._vq_engine_cfg() == (1, True, 1.0, 0.25, 1.0, True)
    from video_tokenizer_amd.engine import _flat_order
    names = [n for n, _, _ in _flat_order(m)]
    assert set(names) == {n for n, _ in m.named_parameters()} and len(names) == len(set(names))
    with pytest.raises(NotImplementedError):
        spec["args"]["bottleneck_type"] = "auto"
        vt.make(spec)
    with pytest.raises(vt.hip.HipError):
        m(torch.zeros(1, 3, 4, 32, 32))


def test_fsq_bottleneck_branch_host_surface():
    """LARPTokenizer(bottleneck_type='fsq') (models/larp_tokenizer.py:219-228): LayerNorm + Linear(768, 6) + FSQ([8,8,8,5,5,5]) + Linear(6, 768);
    the reference's keys, no engine (composed path), loud errors for the engine-only helpers and for CPU tensors."""
    cfg = O.make_cfg("tiny", bottleneck_type="fsq")
    spec = spec_from_cfg(cfg)
    spec["args"]["bottleneck_type"] = "fsq"
    m = vt.make(spec)
    sd = m.state_dict()
    assert {"fsq_in_linear.weight", "fsq_in_linear.bias", "fsq_out_linear.weight", "fsq_out_linear.bias", "fsq_norm.weight", "fsq_norm.bias"} <= set(sd)
    assert sd["fsq_in_linear.weight"].shape == (6, 768) and sd["fsq_out_linear.weight"].shape == (768, 6)
    assert not any(k.startswith("bottleneck.") for k in sd) and m.codebook_size == 64000 and m._engine is None
    assert set(O.init_state_dict(cfg, seed=3).keys()) == set(sd.keys())
    from video_tokenizer_amd.optim import FusedAdam
    with pytest.raises(NotImplementedError):
        FusedAdam(m, lr=1e-4)
    with pytest.raises(vt.hip.HipError):
        m(torch.zeros(1, 3, cfg["frame_num"], cfg["input_size"], cfg["input_size"]))
    perm = m._head_perm(torch.device("cpu"))
    C, pt, p = 3, cfg["temporal_patch_size"], cfg["patch_size"]
    ref_cols = torch.arange(pt * p * p * C).reshape(pt, p, p, C).permute(3, 0, 1, 2).reshape(-1)      # (c, dt, dy, dx) <- (dt, dy, dx, c)
    assert torch.equal(perm, ref_cols)


def test_gemm_tile_order_visits_every_tile_once(tmp_path):
    """The NT GEMM launchers' tile order (vt_common.h: vt_tile_of / vt_auto_col_block -- column blocks of W tile columns so that an XCD's chunk
    of the tile list is a rectangle) on the host: every tile grid up to 70 x 26 and every block width is a permutation of the tiles, and the
    automatic widths are the ones DESIGN quotes.  hipcc compiles the product header for the host; no kernel runs."""
    import shutil
    import subprocess
    from video_tokenizer_amd import build as vt_build
    if not (shutil.which(vt_build.HIPCC) or os.path.exists(vt_build.HIPCC)):
        pytest.skip("needs hipcc")
    exe = str(tmp_path / "tile_order_check")
    subprocess.check_call([vt_build.HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "c", "tile_order_check.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("OK "), out.stdout + out.stderr
