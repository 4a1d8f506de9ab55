"""ORACLE (test infrastructure only -- see oracle/__init__.py): CPU restatement of Lore's detection decode, pre-process
geometry and logical-location post-process.

Follows /root/reference/src/pdftable/model/lore/lineless_table_process.py:
  ``nms_peaks`` :66-73, ``topk_1class`` :76-94, ``corner_decode`` :97-124, ``cell_decode`` = ctdet_4ps_decode :127-267
  (incl. the wiz_rev vertex-snapping loop :178-236 with is_group_faster_faster :357-379, find4ps :329-337),
  ``affine_from_center_scale`` = get_affine_transform :403-438, ``transform_preds`` :471-476,
  ``process_detect_output`` :592-655 (merge_outputs :551-565, filter :568-582, normalized_ps :585-589),
  ``process_logic_output`` :658-663; and lore/processer_lore.py:66-109 for ``lore_preprocess_geometry``.

PINNED by tests/golden/lore_decode.npz: outputs of the reference's own ``process_detect_output`` on seeded head maps
(cv2.getAffineTransform and shapely's Point.within(Polygon) -- unpinned third-party dependencies that are not installed
-- are stood in by ``get_affine_transform_3pt`` / ``point_strictly_in_polygon`` below, so those two are PARITY UNPINNED
and known-answer tested only).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

K_CELLS, K_CORNERS = 3000, 5000          # lineless_table_process.py:593
CELL_SCORE_MIN, CORNER_SCORE_MIN = 0.2, 0.3   # :190, :193 (hard-coded in the wiz_rev loop)


def nms_peaks(heat: torch.Tensor) -> torch.Tensor:
    hmax = F.max_pool2d(heat, 3, 1, 1)
    return heat * (hmax == heat).float()


def topk_1class(scores: torch.Tensor, K: int):
    """scores [1,1,H,W] -> (score [1,K], ind [1,K] int64, ys [1,K], xs [1,K]) in torch.topk order."""
    b, c, H, W = scores.shape
    assert b == 1 and c == 1
    s, ind = torch.topk(scores.view(1, 1, -1), K)
    ind = ind % (H * W)
    ys = (ind / torch.tensor([float(W)])).int().float()
    xs = (ind % W).int().float()
    s2, sel = torch.topk(s.view(1, -1), K)
    return s2, ind.view(1, -1).gather(1, sel), ys.view(1, -1).gather(1, sel), xs.view(1, -1).gather(1, sel)


def _gather(feat: torch.Tensor, ind: torch.Tensor) -> torch.Tensor:
    """feat [1,C,H,W], ind [1,K] -> [1,K,C]"""
    f = feat.permute(0, 2, 3, 1).reshape(1, -1, feat.shape[1])
    return f.gather(1, ind.unsqueeze(2).expand(1, ind.shape[1], f.shape[2]))


def corner_decode(mk: torch.Tensor, st: torch.Tensor, reg: torch.Tensor, K: int = K_CORNERS):
    s, ind, ys, xs = topk_1class(nms_peaks(mk), K)
    r = _gather(reg, ind)
    xs = xs.view(1, K, 1) + r[:, :, 0:1]
    ys = ys.view(1, K, 1) + r[:, :, 1:2]
    sr = _gather(st, ind)
    ctr = torch.cat([xs, ys] * 4, dim=2)
    return {"scores": s.view(1, K, 1), "inds": ind, "xs": xs, "ys": ys, "gboxes": ctr - sr}


def point_strictly_in_polygon(px: float, py: float, poly: np.ndarray) -> bool:
    """shapely ``Point.within(Polygon)``: interior only (a point on the boundary is not within).  Even-odd crossing
    rule in float64 on a simple quadrilateral; boundary points are rejected first."""
    n = len(poly)
    for i in range(n):
        x1, y1 = poly[i]
        x2, y2 = poly[(i + 1) % n]
        cross = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1)
        if cross == 0 and min(x1, x2) <= px <= max(x1, x2) and min(y1, y2) <= py <= max(y1, y2):
            return False
    inside = False
    j = n - 1
    for i in range(n):
        xi, yi = poly[i]
        xj, yj = poly[j]
        if (yi > py) != (yj > py):
            xint = (xj - xi) * (py - yi) / (yj - yi) + xi
            if px < xint:
                inside = not inside
        j = i
    return inside


def is_group(bbox: np.ndarray, gbox: np.ndarray) -> bool:
    """is_group_faster_faster: bounding boxes overlap and at least one vertex of gbox lies strictly inside bbox."""
    b = bbox.reshape(4, 2)
    g = gbox.reshape(4, 2)
    if b[:, 0].min() > g[:, 0].max() or g[:, 0].min() > b[:, 0].max() or b[:, 1].min() > g[:, 1].max() \
            or g[:, 1].min() > b[:, 1].max():
        return False
    bd = b.astype(np.float64)
    return any(point_strictly_in_polygon(float(g[i, 0]), float(g[i, 1]), bd) for i in range(4))


def snap_vertices(bboxes: np.ndarray, scores: np.ndarray, gboxes: np.ndarray, cxs: np.ndarray, cys: np.ndarray,
                  cscores: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """The wiz_rev double loop (:188-236), float32 like the reference's tensors.  bboxes [K,8], gboxes [MK,8].
    Returns (bboxes_rev, scores) -- scores of cells with <= 2 snapped vertices are multiplied by 0.4."""
    rev = bboxes.copy()
    scores = scores.copy()
    n_corner = int(np.argmax(cscores < CORNER_SCORE_MIN)) if (cscores < CORNER_SCORE_MIN).any() else len(cscores)
    for i in range(len(bboxes)):
        if not scores[i] >= CELL_SCORE_MIN:
            break
        count = 0
        bb = bboxes[i]
        for j in range(n_corner):
            if not is_group(bb, gboxes[j]):
                continue
            cx, cy = cxs[j], cys[j]
            d = (bb[0::2] - cx) ** 2 + (bb[1::2] - cy) ** 2          # find4ps: float32 squared distances, first minimum
            k = int(np.argmin(d))
            if rev[i, 2 * k] == bb[2 * k] and rev[i, 2 * k + 1] == bb[2 * k + 1]:
                count += 1
                rev[i, 2 * k], rev[i, 2 * k + 1] = cx, cy
            else:
                ox, oy = bb[2 * k], bb[2 * k + 1]
                d_old = (ox - rev[i, 2 * k]) ** 2 + (oy - rev[i, 2 * k + 1]) ** 2
                d_new = (ox - cx) ** 2 + (oy - cy) ** 2
                if d_old >= d_new:
                    count += 1
                    rev[i, 2 * k], rev[i, 2 * k + 1] = cx, cy
        if count <= 2:
            scores[i] = scores[i] * np.float32(0.4)
    return rev, scores


def cell_decode(heat, wh, ax, cr, corners, reg, K: int = K_CELLS, wiz_rev: bool = True):
    """ctdet_4ps_decode -> (detections [1,K,10] = 8 coords, score, class; ax [1,K,256]; cr_feat [1,K,256])."""
    W = heat.shape[3]
    s, ind, ys, xs = topk_1class(nms_peaks(heat), K)
    r = _gather(reg, ind)
    xs = xs.view(1, K, 1) + r[:, :, 0:1]
    ys = ys.view(1, K, 1) + r[:, :, 1:2]
    whg = _gather(wh, ind)
    axg = _gather(ax, ind)
    bboxes = torch.cat([xs, ys] * 4, dim=2) - whg
    scores = s.view(1, K, 1).clone()
    if wiz_rev:
        rev, sc = snap_vertices(bboxes[0].numpy(), scores[0, :, 0].numpy(), corners["gboxes"][0].numpy(),
                                corners["xs"][0, :, 0].numpy(), corners["ys"][0, :, 0].numpy(),
                                corners["scores"][0, :, 0].numpy())
        boxes_used = torch.from_numpy(rev)[None]
        scores = torch.from_numpy(sc).view(1, K, 1)
    else:
        boxes_used = bboxes
    # corner index x + W * round(y), rounded again (:240-251); torch.round is half-to-even
    cc = torch.cat([boxes_used[:, :, 2 * k:2 * k + 1] + W * torch.round(boxes_used[:, :, 2 * k + 1:2 * k + 2])
                    for k in range(4)], dim=2)
    cc = torch.round(cc).to(torch.int64)
    crf = cr.permute(0, 2, 3, 1).reshape(1, -1, cr.shape[1])
    npix = crf.shape[1]
    # _get_4ps_feat (:39-63) is called with the `cr` TENSOR (not the dict), so its clamp runs: indices >= H*W become
    # (batch - 1) = 0 and negative ones 0 -- an out-of-map corner reads pixel 0
    cc = torch.where(cc < npix, cc, torch.zeros_like(cc))
    cc = torch.where(cc >= 0, cc, torch.zeros_like(cc))
    # NOTE: cr_feat stays in pre-sort order while det/ax are re-sorted below -- as in the reference (:254-263)
    cr_feat = sum(crf[0][cc[0, :, k]] for k in range(4))[None]
    clses = torch.zeros(1, K, 1)
    det = torch.cat([boxes_used, scores, clses], dim=2)
    if wiz_rev:
        _, order = torch.sort(scores, descending=True, dim=1)
        det = det.gather(1, order.expand(1, K, det.shape[2]))
        axg = axg.gather(1, order.expand(1, K, axg.shape[2]))
        ind = ind.gather(1, order[:, :, 0])
    cell_decode.last_inds = ind[0].numpy()        # centre pixel of every output row (tests only)
    return det, axg, cr_feat


def get_affine_transform_3pt(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """cv2.getAffineTransform: the 2x3 map taking three src points to three dst points (6x6 solve in float64)."""
    a = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        a[2 * i, 0:3] = [src[i, 0], src[i, 1], 1]
        a[2 * i + 1, 3:6] = [src[i, 0], src[i, 1], 1]
        b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def affine_from_center_scale(center, scale, out_size, inv: bool = False) -> np.ndarray:
    """get_affine_transform(center, scale, rot=0, output_size) (:403-438) in the reference's float32 steps."""
    sc = np.array([scale, scale], dtype=np.float32)
    src_w = sc[0]
    dst_w, dst_h = out_size
    src_dir = [0 * 1.0 - (src_w * -0.5) * 0.0, 0 * 0.0 + (src_w * -0.5) * 1.0]       # get_dir with rot 0: sn=0, cs=1
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = center
    src[1] = np.asarray(center) + np.asarray(src_dir)
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir

    def third(a, b):
        d = a - b
        return b + np.array([-d[1], d[0]], dtype=np.float32)
    src[2] = third(src[0], src[1])
    dst[2] = third(dst[0], dst[1])
    return get_affine_transform_3pt(dst, src) if inv else get_affine_transform_3pt(src, dst)


def affine_upper_left(center, scale, out_size, inv: bool = False) -> np.ndarray:
    """get_affine_transform_upper_left (:441-468): anchors (cx, cy) -> (0, 0) and, when cx >= cy, (cx, scale) -> (0, out_w)
    (otherwise (scale, cy) -> (out_w, 0)), third point perpendicular."""
    sc = np.array([scale, scale], dtype=np.float32)
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = center
    dst[0] = [0, 0]
    if center[0] < center[1]:
        src[1] = [sc[0], center[1]]
        dst[1] = [out_size[0], 0]
    else:
        src[1] = [center[0], sc[0]]
        dst[1] = [0, out_size[0]]

    def third(a, b):
        d = a - b
        return b + np.array([-d[1], d[0]], dtype=np.float32)
    src[2] = third(src[0], src[1])
    dst[2] = third(dst[0], dst[1])
    return get_affine_transform_3pt(dst, src) if inv else get_affine_transform_3pt(src, dst)


def transform_preds(coords: np.ndarray, center, scale, out_size, upper_left: bool = False) -> np.ndarray:
    """[n,2] feature-map points -> source pixels: float64 matrix times float32 homogeneous point, per point (:471-486)."""
    t = (affine_upper_left if upper_left else affine_from_center_scale)(center, scale, out_size, inv=True)
    p = coords.astype(np.float32).astype(np.float64)
    out = np.zeros(coords.shape)
    out[:, 0] = t[0, 0] * p[:, 0] + t[0, 1] * p[:, 1] + t[0, 2]
    out[:, 1] = t[1, 0] * p[:, 0] + t[1, 1] * p[:, 1] + t[1, 2]
    return out


def process_detect_output(output: Dict[str, torch.Tensor], meta: np.ndarray, wiz_rev: bool = True,
                          vis_thresh: float = 0.2, return_raw: bool = False, upper_left: bool = False):
    """heads (NCHW f32; 'hm' pre-sigmoid) + meta [cx, cy, s, in_h, in_w, out_h, out_w] ->
    (slct_logi_feat f32 [1,n,256], slct_dets_feat i64 [1,n,8], polygons f32 [n,8] in source pixels, dets f32 [K,9])."""
    hm = torch.sigmoid(output["hm"])
    corners = corner_decode(hm[:, 1:2], output["st"], output["reg"])
    det, logi, crf = cell_decode(hm[:, 0:1], output["wh"], output["ax"], output["cr"], corners, output["reg"],
                                 wiz_rev=wiz_rev)
    raw = det[0].numpy()
    d = raw.copy()
    c, s = meta[:2], meta[2]
    out_h, out_w = meta[5], meta[6]
    for k in range(4):
        d[:, 2 * k:2 * k + 2] = transform_preds(d[:, 2 * k:2 * k + 2], c, s, (out_w, out_h), upper_left)
    results = d[:, :9].astype(np.float32)
    n = int((results[:, 8] >= vis_thresh).sum())
    logi = (logi + crf)[:, :n].contiguous()
    ps = torch.from_numpy(raw[:n, :8].astype(np.int32).astype(np.float32))[None]      # filter(): int32 truncation
    ps = torch.round(ps).to(torch.int64).clamp(0, 255)                                 # normalized_ps(.., 256)
    if return_raw:
        return logi, ps, results[:n, :8], results, raw          # raw: [K,10] feature-map quads + score + class
    return logi, ps, results[:n, :8], results


def process_logic_output(logi: torch.Tensor) -> torch.Tensor:
    fl = logi.floor()
    return torch.where(logi - fl > 0.5, fl + 1, fl)


def lore_preprocess_geometry(height: int, width: int, inp_h: int = 1024, inp_w: int = 1024, upper_left: bool = False):
    """TableLorePreProcessor.process (processer_lore.py:66-109)
    -> (trans_input 2x3 float64 for cv2.warpAffine, meta int64 [cx, cy, s, in_h, in_w, out_h, out_w])."""
    s = max(height, width) * 1.0
    if upper_left:
        c = np.array([0, 0], dtype=np.float32)
        trans = affine_upper_left(c, s, (inp_w, inp_h))
    else:
        c = np.array([width / 2.0, height / 2.0], dtype=np.float32)
        trans = affine_from_center_scale(c, s, (inp_w, inp_h))
    meta = np.array([c[0], c[1], s, inp_h, inp_w, inp_h // 4, inp_w // 4]).astype(np.int64)   # torch .long(): truncation
    return trans, meta
