"""PP-OCR flavour of the recognition post-process: ``CTCLabelDecode`` (reference:
src/pdftable/model/ocr_rec_pp/rec_postprocess.py:17-195) over the engine's fused arg-max output.

The reference takes the full ``[batch, T, classes]`` probability tensor, does ``argmax(axis=2)`` / ``max(axis=2)`` on the
host and decodes.  The engine never materialises that tensor: the classifier GEMM's epilogue already yields, per time
step, the winning class id and its value, so this class starts from ``(ids int [B,T], probs float [B,T])``.
Character table: ``['blank'] + dict lines (+ ' ' if use_space_char)``; blank (0) and repeats are dropped; confidence is
the mean of the kept per-step maxima (``[0]`` -> 0.0 for an empty string); Arabic dictionaries reverse the text while
keeping latin/digit runs in order (``pred_reverse``).
"""
from __future__ import annotations

import re
from typing import List, Optional, Sequence, Tuple

import numpy as np

__all__ = ["CTCLabelDecode"]

_LATIN_RUN = re.compile("[a-zA-Z0-9 :*./%+-]")


class CTCLabelDecode:
    def __init__(self, character_dict_path: Optional[str] = None, use_space_char: bool = False,
                 characters: Optional[Sequence[str]] = None):
        self.reverse = False
        if characters is not None:
            chars = list(characters)
        elif character_dict_path is None:
            chars = list("0123456789abcdefghijklmnopqrstuvwxyz")
        else:
            with open(character_dict_path, "rb") as fin:
                chars = [ln.decode("utf-8").strip("\n").strip("\r\n") for ln in fin.readlines()]
            if use_space_char:
                chars.append(" ")
            self.reverse = "arabic" in character_dict_path
        self.character = ["blank"] + chars
        self.dict = {c: i for i, c in enumerate(self.character)}

    @staticmethod
    def pred_reverse(pred: str) -> str:
        out, run = [], ""
        for c in pred:
            if _LATIN_RUN.search(c):
                run += c
            else:
                if run:
                    out.append(run)
                out.append(c)
                run = ""
        if run:
            out.append(run)
        return "".join(out[::-1])

    def decode_ids(self, ids: np.ndarray, probs: Optional[np.ndarray] = None) -> List[Tuple[str, float]]:
        ids = np.asarray(ids)
        res = []
        for b in range(ids.shape[0]):
            row = ids[b]
            keep = np.ones(len(row), dtype=bool)
            keep[1:] = row[1:] != row[:-1]
            keep &= row != 0
            text = "".join(self.character[int(t)] for t in row[keep])
            conf = probs[b][keep] if probs is not None else np.ones(len(row))      # reference: [1] * len(selection), never empty
            if len(conf) == 0:
                conf = [0]
            if self.reverse:
                text = self.pred_reverse(text)
            res.append((text, np.mean(conf).tolist()))
        return res

    def __call__(self, preds, **kwargs):
        """Reference call shape: preds float [B, T, classes] (kept for drop-in use and for the golden test)."""
        if isinstance(preds, (tuple, list)):
            preds = preds[-1]
        preds = np.asarray(preds)
        return self.decode_ids(preds.argmax(axis=2), preds.max(axis=2))
