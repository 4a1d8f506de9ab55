"""Seeded inputs shared by tests/golden/make_golden.py (pplcnet) and the PP-LCNet tests, so that the goldens only have to
carry the reference's OUTPUTS."""
import numpy as np

CLS_GOLDEN_TASKS = {     # task -> (class_num, textline stride list?, (H, W), weight seed)
    "textline_orientation": (2, True, (80, 160), 21),
    "text_image_orientation": (4, False, (224, 224), 22),
    "table_attribute": (6, False, (224, 224), 23),
}
PRE_CASES = [("textline_orientation", (37, 211)), ("textline_orientation", (120, 90)), ("text_image_orientation", (260, 190))]
PIL_CASES = [(37, 211, 80, 160), (80, 160, 80, 160), (200, 900, 80, 160), (30, 30, 224, 224), (113, 57, 224, 224),
             (5, 300, 80, 160), (80, 500, 80, 160), (300, 160, 80, 160), (640, 512, 224, 224)]


def cls_inputs(seed: int, n: int, hw):
    """noise images with per-image gain / offset so that the pooled features (and logits) differ between images"""
    rng = np.random.default_rng(1000 + seed)
    x = rng.standard_normal((n, 3) + tuple(hw)).astype(np.float32)
    a = rng.uniform(0.2, 3.0, (n, 1, 1, 1)).astype(np.float32)
    b = rng.uniform(-1.5, 2.0, (n, 1, 1, 1)).astype(np.float32)
    return x * a + b


def u8_image(seed: int, h: int, w: int):
    return np.random.default_rng(2000 + seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
