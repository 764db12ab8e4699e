"""Reconstruction metrics used by the reference's evaluator, without its unavailable dependencies.

Restates the arithmetic of /root/reference/utils/fvd/fvd.py -- `FeatureStats` running mean/covariance in
float64 (:91-109, :132-149), `_symmetric_matrix_square_root` / `trace_sqrt_product` (:24-33) and
`calculate_fvd` (:419-434) -- and of eval/rfvd_evaluator.py (per-clip MSE over (c,t,h,w), PSNR from MSE,
:86-154).  The I3D feature extractor itself (`utils/fvd/i3d_torchscript.pt`) is absent from the reference
tree, so rFVD can only be produced when the operator supplies features; this module is the (feature ->
Frechet distance) half, exact and testable.  Host-side fp64 math: it is not on the GPU hot path.
"""
import numpy as np
import torch


class FeatureStats:
    """Running sum and outer-product sum of feature rows in float64 (capture_mean_cov mode of the reference)."""

    def __init__(self, max_items=None):
        self.max_items = max_items
        self.num_items = 0
        self.num_features = None
        self.raw_mean = None
        self.raw_cov = None

    def is_full(self):
        return (self.max_items is not None) and (self.num_items >= self.max_items)

    def append(self, x):
        x = np.asarray(x.detach().float().cpu().numpy() if isinstance(x, torch.Tensor) else x, dtype=np.float32)
        assert x.ndim == 2
        if (self.max_items is not None) and (self.num_items + x.shape[0] > self.max_items):
            if self.num_items >= self.max_items:
                return
            x = x[:self.max_items - self.num_items]
        if self.num_features is None:
            self.num_features = x.shape[1]
            self.raw_mean = np.zeros([self.num_features], dtype=np.float64)
            self.raw_cov = np.zeros([self.num_features, self.num_features], dtype=np.float64)
        assert x.shape[1] == self.num_features
        self.num_items += x.shape[0]
        x64 = x.astype(np.float64)
        self.raw_mean += x64.sum(axis=0)
        self.raw_cov += x64.T @ x64

    def append_torch(self, x, num_gpus=1):
        """fvd.py:112-124: with several ranks every rank ends up with ALL ranks' rows, interleaved sample-wise
        (world x broadcast; cold path, kept as is)."""
        assert isinstance(x, torch.Tensor) and x.ndim == 2
        if num_gpus > 1:
            ys = []
            for src in range(num_gpus):
                y = x.clone()
                torch.distributed.broadcast(y, src=src)
                ys.append(y)
            x = torch.stack(ys, dim=1).flatten(0, 1)
        self.append(x.float().cpu().numpy())

    def get_mean_cov(self):
        mean = self.raw_mean / self.num_items
        cov = self.raw_cov / self.num_items
        return mean, cov - np.outer(mean, mean)


class FVDCalculator:
    """utils/fvd/fvd.py:324-434 without the bundled weights: `detector(video_in_minus1_1 BCTHW) -> [B, F]` features.
    Default detector = the I3D torchscript file the reference expects next to fvd.py (absent from the tree,
    `.MISSING_LARGE_BLOBS`): pass `i3d_path=` to a copy, or any callable (tests use a fixed random projection)."""

    def __init__(self, i3d_path=None, device="cuda", detector=None):
        self.device = device
        if detector is None:
            import os
            if i3d_path is None or not os.path.exists(i3d_path):
                raise FileNotFoundError("FVDCalculator: the I3D torchscript detector is not shipped; pass i3d_path= or detector=")
            i3d = torch.jit.load(i3d_path).eval().to(device)
            detector = lambda v: i3d(v, resize=True, return_features=True)  # noqa: E731
        self.detector = detector
        self.num_gpus = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1

    @torch.inference_mode()
    def get_feature_stats_for_batch(self, batch, feats=None, num_gpus=None):
        num_gpus = self.num_gpus if num_gpus is None else num_gpus
        feats = FeatureStats() if feats is None else feats
        data = batch.get("gt", batch.get("video")) if isinstance(batch, dict) else batch
        assert isinstance(data, torch.Tensor) and data.ndim == 5  # BCTHW in [0, 1]
        feats.append_torch(self.detector((data - 0.5) * 2), num_gpus=num_gpus)
        return feats

    def calculate_fvd(self, fake_stats, real_stats):
        return frechet_distance(fake_stats, real_stats)


def _symmetric_matrix_square_root(mat, eps=1e-10):
    u, s, v = torch.svd(mat)
    si = torch.where(s < eps, s, torch.sqrt(s))
    return torch.matmul(torch.matmul(u, torch.diag(si)), v.t())


def trace_sqrt_product(sigma, sigma_v):
    sqrt_sigma = _symmetric_matrix_square_root(sigma)
    return torch.trace(_symmetric_matrix_square_root(torch.matmul(sqrt_sigma, torch.matmul(sigma_v, sqrt_sigma))))


def frechet_distance(stats_gen: FeatureStats, stats_real: FeatureStats) -> float:
    """|mu_g - mu_r|^2 + Tr(C_g + C_r - 2 (C_g^1/2 C_r C_g^1/2)^1/2)   (fvd.py:419-434)."""
    mu_g, cov_g = (torch.from_numpy(a) for a in stats_gen.get_mean_cov())
    mu_r, cov_r = (torch.from_numpy(a) for a in stats_real.get_mean_cov())
    mean = torch.sum((mu_g - mu_r) ** 2)
    trace = torch.trace(cov_g + cov_r) - 2.0 * trace_sqrt_product(cov_g, cov_r)
    return float(trace + mean)


def clip_mse(video, recon):
    """per-clip MSE over (c,t,h,w) after clamping the reconstruction to [0,1] (rfvd_evaluator.py:123-131)."""
    return ((video.float() - recon.float().clamp(0.0, 1.0)) ** 2).mean(dim=(1, 2, 3, 4))


def psnr_given_mse(mse):
    """mean over clips of 10 log10(1 / mse)."""
    return float((10.0 * torch.log10(1.0 / mse)).mean())
