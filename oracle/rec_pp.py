"""Oracle for the PP-OCR recognition pre-processor (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates
  PPOcrRecPreProcessor.resize_norm_img   model/ocr_rec_pp/processor_ocr_rec_pp.py:43-67
  PPOcrRecPreProcessor.__call__          model/ocr_rec_pp/processor_ocr_rec_pp.py:69-135
with the defaults of PPOcrRecognitionConfig (configuration_ocr_recognition_pp.py:43-58): rec_image_shape (3, 48, 320),
rec_batch_num 6, limited widths [16, 1280].

Pinned by tests/golden/rec_pp.npz: the reference's own class run on seeded crops (tests/golden/make_golden.py rec_pp) with
``cv2.resize`` -- opencv-python is absent -- replaced by ``db_pre.cv2_resize_linear_u8``, this repository's restatement
of OpenCV's 8-bit INTER_LINEAR path (PARITY UNPINNED for that one call, as everywhere cv2.resize appears; the size
arithmetic, the ordering, the batching, the normalisation and the padding are the reference's own code).
"""
from __future__ import annotations

import math

import numpy as np

from .db_pre import cv2_resize_linear_u8

REC_IMAGE_SHAPE = (3, 48, 320)
REC_BATCH_NUM = 6
LIMITED_MAX_WIDTH = 1280
LIMITED_MIN_WIDTH = 16


def resize_norm_img(img: np.ndarray, max_wh_ratio: float, rec_image_shape=REC_IMAGE_SHAPE, limited_max_width=LIMITED_MAX_WIDTH,
                    limited_min_width=LIMITED_MIN_WIDTH) -> np.ndarray:
    """processor_ocr_rec_pp.py:43-67.  img uint8 [h, w, 3] -> f32 [3, imgH, imgW]"""
    imgC, imgH, imgW = rec_image_shape
    assert imgC == img.shape[2]
    max_wh_ratio = max(max_wh_ratio, imgW / imgH)
    imgW = int((imgH * max_wh_ratio))
    imgW = max(min(imgW, limited_max_width), limited_min_width)
    h, w = img.shape[:2]
    ratio = w / float(h)
    ratio_imgH = math.ceil(imgH * ratio)
    ratio_imgH = max(ratio_imgH, limited_min_width)
    resized_w = imgW if ratio_imgH > imgW else int(ratio_imgH)
    resized = cv2_resize_linear_u8(np.ascontiguousarray(img), resized_w, imgH)
    resized = resized.astype("float32")
    resized = resized.transpose((2, 0, 1)) / 255
    resized -= 0.5
    resized /= 0.5
    padding_im = np.zeros((imgC, imgH, imgW), dtype=np.float32)
    padding_im[:, :, 0:resized_w] = resized
    return padding_im


def rec_pp_preprocess(images, rec_batch_num=REC_BATCH_NUM, **kw):
    """processor_ocr_rec_pp.py:69-135 for a list of RGB uint8 crops: -> list of {'image' f32 [b, 3, 48, imgW], 'indices',
    'batch_beg_img_no'} (the crops sorted by aspect ratio, ``rec_batch_num`` per mini-batch, each mini-batch padded to
    the width of its widest member)."""
    width_list = [img.shape[1] / float(img.shape[0]) for img in images]
    indices = np.argsort(np.array(width_list))
    out = []
    for beg in range(0, len(images), rec_batch_num):
        end = min(len(images), beg + rec_batch_num)
        max_wh_ratio = 0
        for ino in range(beg, end):
            h, w = images[indices[ino]].shape[0:2]
            max_wh_ratio = max(max_wh_ratio, w * 1.0 / h)
        batch = [resize_norm_img(images[indices[ino]], max_wh_ratio, **kw)[np.newaxis, :] for ino in range(beg, end)]
        out.append({"image": np.concatenate(batch), "indices": indices, "batch_beg_img_no": beg})
    return out
