"""PIL-backed stand-in for ``pygame.image.load`` (tests-only, see the package docstring)."""
from __future__ import annotations

import numpy as np


class _Image:
    def __init__(self, rgb: np.ndarray):
        self.rgb = rgb  # (H, W, 3) uint8

    def convert(self):
        return self

    def get_width(self):
        return int(self.rgb.shape[1])

    def get_height(self):
        return int(self.rgb.shape[0])

    def get_at(self, xy):
        x, y = xy
        if not (0 <= x < self.rgb.shape[1] and 0 <= y < self.rgb.shape[0]):
            raise IndexError("pixel index out of range")
        r, g, b = self.rgb[y, x]
        return (int(r), int(g), int(b), 255)


def load(path):
    from PIL import Image

    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint8).copy()
    return _Image(rgb)
