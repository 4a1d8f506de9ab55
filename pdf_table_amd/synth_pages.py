"""Seeded synthetic PubTabNet-style pages -- there are no datasets offline.

``make_page(idx)``: white 1024x1024x3 uint8 canvas, 1-2 ruled tables (6-14 rows x 3-8 cols, 1-2 px
black rules, ~10 % merged cells), one text line of dark glyph-like blobs per cell, 8-20 free paragraph
lines outside the tables, Gaussian noise sigma=3.  RNG: ``numpy.random.default_rng(20250103 + idx)``.
Returns the page plus the ground-truth layout (table boxes, cell boxes, text-line boxes) so later
stages can be fed realistic crops before real checkpoints are available.
"""
from __future__ import annotations

import numpy as np

__all__ = ["make_page", "make_pages"]


def _text_line(img, rng, x0, y0, x1, hgt):
    """dark bars / glyph blobs of height hgt between x0 and x1 starting at row y0"""
    x = x0
    while x < x1 - 4:
        wlen = int(rng.integers(3, 9))           # glyphs in this word
        for _ in range(wlen):
            gw = int(rng.integers(max(3, hgt // 3), max(5, hgt // 2 + 2)))
            if x + gw >= x1:
                break
            top = y0 + int(rng.integers(0, max(1, hgt // 4)))
            bot = y0 + hgt - int(rng.integers(0, max(1, hgt // 4)))
            shade = int(rng.integers(10, 70))
            img[top:bot, x:x + gw] = shade
            if rng.uniform() < 0.5:               # a counter (hole) inside the glyph
                img[top + 2:bot - 2, x + 1:x + gw - 1] = 255 if gw > 3 else shade
            x += gw + int(rng.integers(1, 3))
        x += int(rng.integers(5, 10))


def make_page(idx: int, size: int = 1024):
    rng = np.random.default_rng(20250103 + idx)
    img = np.full((size, size, 3), 255, dtype=np.uint8)
    gray = np.full((size, size), 255, dtype=np.int16)
    ntab = 1 if rng.uniform() < 0.7 else 2
    tables, cells, lines = [], [], []
    band = size // ntab
    used = np.zeros(size, bool)
    for t in range(ntab):
        rows, cols = int(rng.integers(6, 15)), int(rng.integers(3, 9))
        tw = int(rng.integers(int(size * 0.55), int(size * 0.9)))
        rh = int(rng.integers(26, 40))
        th = min(rows * rh, band - 120)
        rows = max(3, th // rh)
        th = rows * rh
        tx = int(rng.integers(30, size - tw - 30))
        ty = t * band + int(rng.integers(40, max(41, band - th - 60)))
        lw = int(rng.integers(1, 3))
        col_edges = np.sort(rng.choice(np.arange(tx + 40, tx + tw - 40, 20), cols - 1, replace=False)) if cols > 1 else []
        xs = [tx] + [int(c) for c in col_edges] + [tx + tw]
        ys = [ty + r * rh for r in range(rows + 1)]
        merged = set()
        for r in range(rows):
            for c in range(len(xs) - 2):
                if rng.uniform() < 0.10:
                    merged.add((r, c))
        for yy in ys:
            gray[yy:yy + lw, tx:tx + tw + lw] = 0
        for ci, xx in enumerate(xs):
            for r in range(rows):
                if 0 < ci < len(xs) - 1 and (r, ci - 1) in merged:
                    continue
                gray[ys[r]:ys[r + 1] + lw, xx:xx + lw] = 0
        for r in range(rows):
            c = 0
            while c < len(xs) - 1:
                c2 = c + 1
                if (r, c) in merged and c2 < len(xs) - 1:
                    c2 += 1
                cx0, cx1, cy0, cy1 = xs[c], xs[c2], ys[r], ys[r + 1]
                cells.append((cx0, cy0, cx1, cy1))
                hgt = int(rng.integers(14, min(23, rh - 8)))
                lx0 = cx0 + 6
                lx1 = cx0 + 6 + int((cx1 - cx0 - 12) * rng.uniform(0.5, 1.0))
                ly = cy0 + (rh - hgt) // 2 + 1
                if lx1 - lx0 > 12:
                    _text_line(gray, rng, lx0, ly, lx1, hgt)
                    lines.append((lx0, ly, lx1, ly + hgt))
                c = c2
        tables.append((tx, ty, tx + tw + lw, ty + th + lw))
        used[max(0, ty - 10):ty + th + 12] = True
    # free paragraph lines outside the tables
    nfree = int(rng.integers(8, 21))
    tries = 0
    while nfree > 0 and tries < 400:
        tries += 1
        hgt = int(rng.integers(14, 23))
        y = int(rng.integers(10, size - hgt - 10))
        if used[y - 4:y + hgt + 4].any():
            continue
        x0 = int(rng.integers(30, 200))
        x1 = int(rng.integers(size // 2, size - 30))
        _text_line(gray, rng, x0, y, x1, hgt)
        lines.append((x0, y, x1, y + hgt))
        used[y - 4:y + hgt + 4] = True
        nfree -= 1
    noisy = gray[:, :, None].astype(np.float32) + rng.normal(0, 3.0, (size, size, 3)).astype(np.float32)
    img[:] = np.clip(np.rint(noisy), 0, 255).astype(np.uint8)
    meta = {"tables": np.array(tables, dtype=np.int32).reshape(-1, 4), "cells": np.array(cells, dtype=np.int32).reshape(-1, 4),
            "lines": np.array(lines, dtype=np.int32).reshape(-1, 4)}
    return img, meta


def make_pages(start: int, n: int, size: int = 1024) -> np.ndarray:
    return np.stack([make_page(start + i, size)[0] for i in range(n)])
