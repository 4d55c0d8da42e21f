"""Post-processing chain: exposure -> contrast/brightness -> saturation -> global Reinhard ->
gamma (upstream: src/post_processing/post_processing.py:21-82).

``PostProcessing(...).process(img)`` accepts what upstream accepts (a numpy (W, H, 3) array) and,
in addition, a torch tensor.  A float32 (W, H, 3) tensor on a GPU goes through the fused HIP
kernels behind ``rt_postprocess`` (two passes over the image instead of a dozen, both global
reductions in the first); ``process_device`` does the same straight from a renderer's accumulation
buffer, folding in the division by the pass count and the flip to display orientation.  Other
tensors take a stage-by-stage torch path.  Quirks kept because they define the
output: the clip to [0, 1] happens BEFORE tone mapping (:56), the luminance weights are
(0.299, 0.587, 0.114) (:44), Reinhard normalises by the image maximum (:72) and gamma is applied
as ``img ** gamma``, not ``1 / gamma`` (:82).
"""
from __future__ import annotations

import numpy as np

__all__ = ["PostProcessing"]

_LUMA = (0.299, 0.587, 0.114)
_contexts: dict = {}


def _context(device_index: int):
    """One library context per GPU for tensors that do not come with a renderer's."""
    be = _contexts.get(device_index)
    if be is None:
        from .backend.lib import HipBackend

        be = _contexts[device_index] = HipBackend(device_index)
    return be


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class PostProcessing:
    def __init__(self, exposure=1, brightness=0, contrast=1, saturation=1, gamma=1) -> None:
        self.exposure = exposure
        self.brightness = brightness
        self.contrast = contrast
        self.saturation = saturation
        self.gamma = gamma

    # -- public chain ---------------------------------------------------------------------------
    def _fused(self, backend, width, height, src_ptr, scale, flip_y, device):
        import torch

        out = torch.empty((width, height, 3), dtype=torch.float32, device=device)
        backend.postprocess(width, height, out.data_ptr(), src_ptr=src_ptr, scale=scale, flip_y=flip_y,
                            exposure=self.exposure, brightness=self.brightness, contrast=self.contrast,
                            saturation=self.saturation, gamma=self.gamma,
                            stream=torch.cuda.current_stream(device).cuda_stream)
        return out

    def process_device(self, render_result):
        """The post-processed display image (W, H, 3) of a render, as a torch tensor on the renderer's
        GPU: ``process(render_result.img_arr)`` without the image leaving HBM."""
        be = render_result._backend
        if be is None or not be.accum_ptr():
            raise RuntimeError("no device buffer: render first")
        be.synchronize()
        return self._fused(be, render_result.WIDTH, render_result.HEIGHT, 0, 1.0 / max(render_result.rendered_passes, 1),
                           True, f"cuda:{be.device}")

    def process(self, img_arr):
        if _is_torch(img_arr) and img_arr.is_cuda and img_arr.dim() == 3 and img_arr.shape[2] == 3:
            import torch

            x = img_arr.to(torch.float32).contiguous()
            return self._fused(_context(x.device.index or 0), x.shape[0], x.shape[1], x.data_ptr(), 1.0, False, x.device)
        x = self.exposure_correction(img_arr)
        x = self.contrast_and_brightness_correction(x)
        x = self.saturation_correction(x)
        x = self.reinhardt_tone_mapping(x)
        return self.gamma_correction(x)

    # -- stages ---------------------------------------------------------------------------------
    def grayscale(self, img):
        if _is_torch(img):
            return img[..., 0] * _LUMA[0] + img[..., 1] * _LUMA[1] + img[..., 2] * _LUMA[2]
        return np.dot(img, np.array(_LUMA))

    def exposure_correction(self, img):
        return img * self.exposure

    def contrast_and_brightness_correction(self, img):
        y = self.contrast * (img - 0.5) + self.brightness + 0.5
        return y.clamp(0, 1) if _is_torch(y) else np.clip(y, 0, 1)

    def saturation_correction(self, img):
        gray = self.grayscale(img)[..., None]
        y = gray + (img - gray) * self.saturation
        return y.clamp(0, 1) if _is_torch(y) else np.clip(y, 0, 1)

    def reinhardt_tone_mapping(self, img, a=0.18, saturation=1.0):
        img = img + 1e-8
        lw = self.grayscale(img)
        if _is_torch(img):
            lwa = lw.log().mean().exp()
            lm = a / lwa * lw
            ld = lm * (1 + lm / (saturation ** 2)) / (1 + lm)
            ld = ld / ld.max()
            return (img / lw[..., None] * ld[..., None]).clamp(0, 1)
        lwa = np.exp(np.mean(np.log(lw)))
        lm = a / lwa * lw
        ld = lm * (1 + lm / (saturation ** 2)) / (1 + lm)
        ld = ld / np.max(ld)
        return np.clip(img / lw[..., None] * ld[..., None], 0, 1)

    def gamma_correction(self, img):
        return img ** self.gamma if _is_torch(img) else np.power(img, self.gamma)
