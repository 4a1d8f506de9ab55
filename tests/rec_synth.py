"""Seeded text-line crops shared by tests/golden/make_golden.py (inputs of the reference run) and the parity tests: data
generators only."""
import numpy as np


def rec_pp_crops(seed=107):
    """seeded RGB crops of the sizes text lines come in (plus the corner cases of resize_norm_img: a crop wider than the
    1280 limit at height 48, a 1-pixel-wide one, exact 2x decimation, identity size)"""
    rng = np.random.default_rng(seed)
    shapes = [(18, 70), (22, 300), (48, 320), (96, 640), (30, 31), (17, 1), (40, 1400), (12, 500), (25, 160), (48, 97),
              (33, 260), (20, 20), (64, 48), (19, 700)]
    crops = []
    for (h, w) in shapes:
        base = rng.integers(0, 256, (max(1, h // 3), max(1, w // 3), 3))
        img = np.kron(base, np.ones((3, 3, 1)))[:h, :w]
        if img.shape[0] < h or img.shape[1] < w:
            img = np.pad(img, ((0, h - img.shape[0]), (0, w - img.shape[1]), (0, 0)), mode="edge")
        crops.append(np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8))
    return crops
