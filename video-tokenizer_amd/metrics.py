"""Reconstruction metrics used by the reference's evaluator, without its unavailable dependencies.

Same quantities as /root/reference/utils/fvd/fvd.py -- `FeatureStats` mean / covariance in float64 (:91-109, :132-149), the
Frechet distance of `calculate_fvd` (:419-434) with the matrix square roots of :24-33 -- in this build's own form (one
(count, sum, outer) accumulator; square roots by symmetric eigen-decomposition, tested equal to the SVD form and to scipy's
sqrtm) -- and of eval/rfvd_evaluator.py (per-clip MSE over (c,t,h,w), PSNR from MSE,
:86-154).  The I3D feature extractor itself (`utils/fvd/i3d_torchscript.pt`) is absent from the reference
tree, so rFVD can only be produced when the operator supplies features; this module is the (feature ->
Frechet distance) half, exact and testable.  Host-side fp64 math: it is not on the GPU hot path.
"""
import numpy as np
import torch


class FeatureStats:
    """First and second moments of a stream of feature rows, kept as the triple (count, sum of rows, sum of outer products) in
    float64 -- what utils/fvd/fvd.py's capture_mean_cov mode keeps (:91-109, :132-149).  `max_items` caps the stream: rows beyond
    the cap are dropped, a batch straddling it is cut.  Members the evaluator / tests read: num_items, num_features, is_full(),
    append(), append_torch(), get_mean_cov()."""

    def __init__(self, max_items=None):
        self.max_items = max_items
        self.num_items = 0
        self.num_features = None
        self._sum = None          # [F]    float64
        self._outer = None        # [F, F] float64

    def room(self):
        """rows the stream still accepts (inf without a cap)"""
        return float("inf") if self.max_items is None else max(self.max_items - self.num_items, 0)

    def is_full(self):
        return self.room() == 0

    def update(self, rows):
        """rows: [n, F] array-like or tensor; the features are taken at float32 precision (the detector's), the sums in float64"""
        if isinstance(rows, torch.Tensor):
            rows = rows.detach().to(device="cpu", dtype=torch.float32).numpy()
        rows = np.asarray(rows, dtype=np.float32)
        if rows.ndim != 2:
            raise ValueError(f"FeatureStats.update: expected [n, features], got shape {rows.shape}")
        if self.num_features is None:
            self.num_features = rows.shape[1]
            self._sum = np.zeros(self.num_features, dtype=np.float64)
            self._outer = np.zeros((self.num_features, self.num_features), dtype=np.float64)
        elif rows.shape[1] != self.num_features:
            raise ValueError(f"FeatureStats.update: {rows.shape[1]} features, the stream has {self.num_features}")
        take = int(min(rows.shape[0], self.room()))
        if take == 0:
            return self
        r = rows[:take].astype(np.float64)
        self.num_items += take
        self._sum += r.sum(axis=0)
        self._outer += r.T @ r
        return self

    append = update       # the reference's name (fvd.py:91)

    def append_torch(self, x, num_gpus=1):
        """fvd.py:112-124: with several ranks every rank ends up with ALL ranks' rows, sample i of rank 0, 1, ... before sample i + 1
        (one broadcast per rank; cold path)."""
        if not (isinstance(x, torch.Tensor) and x.ndim == 2):
            raise ValueError("FeatureStats.append_torch: expected a [n, features] tensor")
        if num_gpus > 1:
            gathered = torch.empty((x.shape[0], num_gpus, x.shape[1]), dtype=x.dtype, device=x.device)
            for src in range(num_gpus):
                piece = x.clone()
                torch.distributed.broadcast(piece, src=src)
                gathered[:, src] = piece
            x = gathered.reshape(-1, x.shape[1])
        return self.update(x)

    def get_mean_cov(self):
        """(mean [F], biased covariance [F, F]) = (S / n, O / n - mean mean^T)"""
        mean = self._sum / self.num_items
        return mean, self._outer / self.num_items - mean[:, None] * mean[None, :]

    # the reference's attribute names for the two sums (read-only views)
    raw_mean = property(lambda self: self._sum)
    raw_cov = property(lambda self: self._outer)


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


def psd_sqrt(mat, eps=1e-10):
    """Square root of a symmetric positive semi-definite matrix through its eigen-decomposition: V diag(g(w)) V^T with g the
    reference's rule on the spectrum (fvd.py:24-27 applies it to singular values, which for a PSD matrix ARE the eigenvalues):
    values below `eps` are kept as they are, the others replaced by their square root.  Rounding can leave eigenvalues of a
    rank-deficient covariance slightly negative; their magnitude is what an SVD would report, so |w| enters the rule."""
    sym = 0.5 * (mat + mat.transpose(-1, -2))
    w, vecs = torch.linalg.eigh(sym)
    w = w.abs()
    g = torch.where(w < eps, w, w.sqrt())
    return (vecs * g.unsqueeze(-2)) @ vecs.transpose(-1, -2)


def trace_sqrt_product(sigma, sigma_v):
    """Tr((sigma^1/2 sigma_v sigma^1/2)^1/2)  (fvd.py:30-33); the inner product is symmetric PSD, so only its spectrum is needed"""
    root = psd_sqrt(sigma)
    inner = root @ sigma_v @ root
    w = torch.linalg.eigvalsh(0.5 * (inner + inner.transpose(-1, -2))).abs()
    return torch.where(w < 1e-10, w, w.sqrt()).sum()


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
