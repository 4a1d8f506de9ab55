"""Seeded synthetic Lore head maps and PicoDet head outputs shared by tests/golden/make_golden.py (inputs of the reference run) and the
parity tests (inputs of the oracle / HIP run): data generators only."""
import numpy as np


def synth_lore_heads(seed, H=80, W=80):
    """Seeded head maps that look like a table: cell-centre and corner peaks on a jittered grid, quads of about the
    grid pitch, so that the wiz_rev vertex snapping really fires.  NCHW f32, 'hm' pre-sigmoid."""
    rng = np.random.default_rng(seed)
    hm = rng.normal(-6.0, 0.5, (1, 2, H, W)).astype(np.float32)
    wh = np.zeros((1, 8, H, W), np.float32)
    st = np.zeros((1, 8, H, W), np.float32)
    pitch_x, pitch_y = 9, 7
    for gy in range(3, H - 3, pitch_y):
        for gx in range(4, W - 4, pitch_x):
            if rng.uniform() < 0.85:
                hm[0, 0, gy, gx] = rng.normal(1.5, 1.2)             # cell centre (some below the 0.2 threshold)
            for (cy, cx) in ((gy - pitch_y // 2, gx - pitch_x // 2),):
                if 0 <= cy < H and 0 <= cx < W and rng.uniform() < 0.9:
                    hm[0, 1, cy, cx] = rng.normal(1.0, 1.0)         # corner point
    hw, hh = pitch_x / 2.0, pitch_y / 2.0
    base = np.array([hw, hh, -hw, hh, -hw, -hh, hw, -hh], np.float32).reshape(1, 8, 1, 1)
    wh[:] = base + rng.normal(0, 0.6, (1, 8, H, W)).astype(np.float32)
    st[:] = base + rng.normal(0, 0.6, (1, 8, H, W)).astype(np.float32)
    reg = rng.uniform(0.2, 0.8, (1, 2, H, W)).astype(np.float32)
    ax = rng.standard_normal((1, 256, H, W)).astype(np.float32)
    cr = rng.standard_normal((1, 256, H, W)).astype(np.float32)
    return {"hm": hm, "st": st, "wh": wh, "ax": ax, "cr": cr, "reg": reg}


def synth_pico_heads(seed, target=(160, 128), ncls=5, reg_max=7, strides=(8, 16, 32, 64)):
    """seeded head outputs of a layout-like page: a few strong, overlapping detections per class on every level"""
    rng = np.random.default_rng(seed)
    scores, boxes = [], []
    for s in strides:
        fh, fw = int(np.ceil(target[0] / s)), int(np.ceil(target[1] / s))
        a = fh * fw
        sc = rng.uniform(0.0, 0.3, (1, a, ncls)).astype(np.float32)
        hot = rng.choice(a, max(2, a // 6), replace=False)
        sc[0, hot, rng.integers(0, ncls, len(hot))] = rng.uniform(0.45, 0.99, len(hot)).astype(np.float32)
        scores.append(sc)
        boxes.append((rng.standard_normal((1, a, 4 * (reg_max + 1))) * 2.0).astype(np.float32))
    return scores, boxes


def synth_table_grids(seed, n_cases=6):
    """seeded (polygons, logi) cases: regular grids with merged cells, shuffled, plus a few Lore-like irregular ones
    (duplicate / overlapping logical locations, a row whose cells all span the same number of rows)"""
    rng = np.random.default_rng(seed)
    cases = []
    for ci in range(n_cases):
        rows, cols = int(rng.integers(2, 7)), int(rng.integers(2, 6))
        occ = np.zeros((rows, cols), bool)
        polys, logi = [], []
        xs = np.cumsum(np.concatenate([[10.0], rng.uniform(40, 120, cols)]))
        ys = np.cumsum(np.concatenate([[20.0], rng.uniform(18, 40, rows)]))
        for r in range(rows):
            for c in range(cols):
                if occ[r, c]:
                    continue
                rs = int(rng.integers(1, 3)) if rng.uniform() < 0.25 else 1
                cs = int(rng.integers(1, 3)) if rng.uniform() < 0.25 else 1
                rs, cs = min(rs, rows - r), min(cs, cols - c)
                if occ[r:r + rs, c:c + cs].any():
                    rs = cs = 1
                occ[r:r + rs, c:c + cs] = True
                x1, y1, x2, y2 = xs[c], ys[r], xs[c + cs], ys[r + rs]
                j = rng.uniform(-1.5, 1.5, 8)
                polys.append(np.array([x1, y1, x2, y1, x2, y2, x1, y2]) + j)
                logi.append([c, c + cs - 1, r, r + rs - 1])
        polys, logi = np.array(polys, np.float32), np.array(logi, np.float32)
        if ci == n_cases - 2:                       # every cell of the first row spans two rows
            m = logi[:, 2] == 0
            logi[m, 3] = 1
        if ci == n_cases - 1:                       # duplicated logical location
            polys = np.concatenate([polys, polys[:2] + 3.0])
            logi = np.concatenate([logi, logi[:2]])
        perm = rng.permutation(len(polys))
        cases.append((polys[perm], logi[perm]))
    return cases


def synth_table_texts(seed, polys):
    """seeded OCR text lines for a table whose cells are `polys` [n,8] (TL,TR,BR,BL): most cells get 0-3 stacked lines
    inside them (some poking over the border, some wider than the cell), a few lines lie between cells or outside the
    table.  -> (boxes f32 [t,4,2] as OcrCell.bbox expects, texts list[str]); order shuffled like detector output."""
    rng = np.random.default_rng(1000 + seed)
    words = ["total", "0", "O", "1.2.3", "12.5", "n/a", "item", "Q1", "2024", "o", "1.000.000", "x y", "abc\n", "9,5", " lead"]
    boxes, texts = [], []
    p = np.asarray(polys, np.float64)
    for k in range(len(p)):
        x1, y1, x2, y2 = p[k, 0], p[k, 1], p[k, 4], p[k, 5]
        n = int(rng.integers(0, 4))
        h = (y2 - y1) / max(n, 1)
        for i in range(n):
            lx1 = x1 + rng.uniform(1, 0.3 * (x2 - x1))
            lx2 = x2 - rng.uniform(1, 0.3 * (x2 - x1))
            ly1 = y1 + i * h + rng.uniform(0.5, 0.2 * h)
            ly2 = y1 + (i + 1) * h - rng.uniform(0.5, 0.2 * h)
            u = rng.uniform()
            if u < 0.15:        # pokes out of the cell
                lx2 = x2 + rng.uniform(2.5, 12)
            elif u < 0.25:
                ly1 = y1 - rng.uniform(2.5, 6)
            if i > 0 and rng.uniform() < 0.3:       # second fragment on the same text row
                boxes.append([[lx2 + 2, ly1 + 1], [lx2 + 14, ly1 + 1], [lx2 + 14, ly2], [lx2 + 2, ly2]])
                texts.append(str(words[int(rng.integers(len(words)))]))
            boxes.append([[lx1, ly1], [lx2, ly1 + rng.uniform(-0.8, 0.8)], [lx2, ly2], [lx1, ly2 + rng.uniform(-0.8, 0.8)]])
            texts.append(str(words[int(rng.integers(len(words)))]))
    xmin, ymin, xmax, ymax = p[:, 0::2].min(), p[:, 1::2].min(), p[:, 0::2].max(), p[:, 1::2].max()
    for _ in range(4):          # strays: just outside the table and far away
        cx, cy = rng.uniform(xmin - 30, xmax + 30), rng.choice([ymin - rng.uniform(1, 30), ymax + rng.uniform(1, 30)])
        boxes.append([[cx, cy], [cx + 30, cy], [cx + 30, cy + 10], [cx, cy + 10]])
        texts.append("stray")
    perm = rng.permutation(len(boxes))
    return np.asarray(boxes, np.float32)[perm], [texts[i] for i in perm]
