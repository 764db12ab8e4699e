"""Reconstruction evaluator: the loop of /root/reference/eval/rfvd_evaluator.py:15-155 (UCFrFVDEvaluator) over this
build's tokenizer.

Per batch: `encode_eval` -> `decode_eval(encoded, num_x_tokens)` (fewer frames than the training length are allowed:
the evaluator clears `x_embedder.strict_vid_size`, :33) -> clamp to [0, 1] -> at most 16 frames -> per-clip MSE,
optional perceptual distance on frames, I3D-feature statistics of real and reconstructed clips when a clip has >= 12
frames (:133-137); at the end mean MSE, PSNR from the per-clip MSEs, Frechet distance (or -1), mean perceptual value.

What differs: the three things the reference constructs from packages/files that are absent offline are injected --
`loader` (any iterable of {'gt': BCTHW in [0,1]}; the reference builds `datasets.make('video_dataset')` + DataLoader),
`detector` (I3D torchscript) and `perceptual_loss` (lpips.LPIPS('vgg')); without them FVD is -1 / the perceptual
value is NaN.  `use_amp/amp_dtype/compile` are accepted and ignored: the engine has its own cast points.
"""
import torch

from .metrics import FVDCalculator, clip_mse


class UCFrFVDEvaluator:
    def __init__(self, model, dataset_csv=None, root_path="data/metadata", frame_num=16, crop_size=128, batch_size=4, num_workers=4,
                 use_amp=True, amp_dtype=torch.float16, compile=False, token_subsample=None, repeat_to_16=False,
                 loader=None, detector=None, i3d_path=None, perceptual_loss=None):
        self.model = model.cuda().eval()
        if hasattr(self.model, "x_embedder"):
            self.model.x_embedder.strict_vid_size = False
        if token_subsample is not None:
            raise NotImplementedError("token_subsample decoding is not built")
        self.dataset_csv, self.root_path, self.frame_num, self.crop_size = dataset_csv, root_path, frame_num, crop_size
        self.batch_size, self.num_workers = batch_size, num_workers
        self.repeat_to_16 = repeat_to_16
        self.psnr_given_mse = lambda m: (-10 * torch.log10(m)).mean()
        self.perceptual_loss = perceptual_loss
        try:
            self.fvdc = FVDCalculator(i3d_path=i3d_path, detector=detector)
        except FileNotFoundError:
            self.fvdc = None
        if loader is None:
            raise ValueError("UCFrFVDEvaluator: pass loader= (an iterable of {'gt': video}); the reference's video_dataset/decord "
                             "pipeline is outside this build")
        self.loader = loader

    @staticmethod
    def repeat_to_16_frames(video):
        """rfvd_evaluator.py:75-82"""
        t = video.shape[2]
        return video if t >= 16 else torch.cat([video, video[:, :, -1:].repeat(1, 1, 16 - t, 1, 1)], dim=2)

    def evaluate(self, no_fvd=False):
        mse_l, lp_l = [], []
        fake_stats = real_stats = None
        with torch.inference_mode():
            for batch in self.loader:
                vb = batch["gt"].cuda()
                n_frames = vb.size(2)
                if self.repeat_to_16:
                    vb = self.repeat_to_16_frames(vb)
                er = self.model.encode_eval(vb)
                rvb = self.model.decode_eval(er["encoded"], num_x_tokens=er.get("num_x_tokens"))
                if isinstance(rvb, dict):
                    rvb = rvb["pred_frames"]
                rvb = rvb.float().clamp(0.0, 1.0)
                vb, rvb = vb[:, :, :16], rvb[:, :, :16]
                if self.repeat_to_16:
                    vb, rvb = vb[:, :, :n_frames], rvb[:, :, :n_frames]
                mse_l.append(clip_mse(vb, rvb))
                if self.perceptual_loss is not None:
                    b, c, t, h, w = vb.shape
                    fr = lambda v: v.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)  # noqa: E731
                    lp_l.append(self.perceptual_loss(fr(vb), fr(rvb)).reshape(-1))
                if vb.size(2) >= 12 and self.fvdc is not None and not no_fvd:
                    fake_stats = self.fvdc.get_feature_stats_for_batch(rvb, fake_stats)
                    real_stats = self.fvdc.get_feature_stats_for_batch(vb, real_stats)
        mse = torch.cat(mse_l)
        assert mse.ndim == 1
        lpips_val = torch.cat(lp_l).mean() if lp_l else torch.tensor(float("nan"))
        psnr_val = self.psnr_given_mse(mse)
        fvd = -1.0 if (no_fvd or fake_stats is None or real_stats is None) else self.fvdc.calculate_fvd(fake_stats, real_stats)
        return mse.mean().item(), psnr_val, fvd, lpips_val
