"""TEST INFRASTRUCTURE ONLY -- the four stages of a page composed from the oracle's own pieces, i.e. what the reference's
``OcrSystemTask.__call__`` (src/pdftable/model/ocr_pdf/ocr_system_task.py:549-734) computes for an IMAGE page with the in-tree
architectures: layout (:203-215) -> text detection + reading-order sort (:629, :148-166) -> one recogniser call per detected line
(:630, :296-336) -> table structure on every table region (:184-199, Lore wtw).  Every function called here cites the reference
lines it restates; this file only strings them together (used by tests/golden/make_golden.py::gen_e2e_page and nothing else).
The JPEG round trip of the table crop (:192-198) is not reproduced."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch


def page_chain(page: np.ndarray, sds: Dict[str, dict], table_boxes: Sequence[Sequence[int]], thresh: float = 0.3, box_thresh: float = 0.6,
               unclip_ratio: float = 1.5, vis_thresh: float = 0.2, layout_classes: int = 5) -> Dict:
    """page uint8 [h, w, 3] RGB; sds: state_dicts 'db', 'crnn', 'pico', 'lore', 'proc'; table_boxes: integer x1, y1, x2, y2 regions, or None: the layout stage's own "table" regions (the reference's flow).
    -> dict of numpy arrays / lists (see the keys below)."""
    from . import crnn as ocrnn
    from . import db_net, db_post, db_pre
    from . import lore_decode as od
    from . import lore_net, lore_pre, lore_processor
    from . import picodet as opico
    out: Dict = {}
    with torch.no_grad():
        # ---- layout
        xl, sf = opico.picodet_preprocess(page)
        sc, bx = opico.picodet_forward(sds["pico"], torch.from_numpy(xl)[None], layout_classes)
        lay = opico.picodet_postprocess([s_.numpy() for s_ in sc], [b_.numpy() for b_ in bx], list(page.shape[:2]), sf, [800, 608],
                                        opico.LABELS["en"])
        out["layout"] = lay
        if table_boxes is None:
            # the reference's hand-off (ocr_system_task.py:184-198): TableProcessUtils.get_layout_by_type(layout, "table") -- score >= 0.2, top to
            # bottom (pdf_table/table_common.py:1287-1300) --, each region cropped at its rounded coordinates (crop_image_by_box,
            # utils/ocr/ocr_common_utils.py:279-280)
            tabs = sorted((it for it in lay if it["label"].lower() == "table" and it["score"] >= 0.2), key=lambda it: it["bbox"][1])
            table_boxes = [[round(float(v)) for v in it["bbox"]] for it in tabs]
            table_boxes = [b for b in table_boxes if b[2] > b[0] and b[3] > b[1]]
        out["table_boxes"] = np.asarray(table_boxes, dtype=np.int64).reshape(-1, 4)
        # ---- detection
        chw, shape_list = db_pre.preprocess_db_pp(page)
        logits = db_net.db_forward_fp32(sds["db"], torch.from_numpy(np.ascontiguousarray(chw))[None], return_logits=True)[0, 0]
        prob = torch.sigmoid(logits).numpy()
        boxes = db_post.db_postprocess(prob, shape_list, page.shape, thresh, box_thresh, unclip_ratio, False, 1000)
        boxes = db_post.sort_det_result(boxes) if len(boxes) else boxes
        out["det_boxes"] = np.asarray(boxes).reshape(-1, 8)
        out["det_prob_near_thresh"] = int((np.abs(prob - thresh) <= 1e-3).sum())
        # ---- recognition: one call per line
        ids, margins, wins = [], [], []
        for q in out["det_boxes"]:
            x = ocrnn.rec_preprocess(ocrnn.crop_image(page, ocrnn.order_point(q)))
            lg = ocrnn.crnn_forward_fp32(sds["crnn"], x, native_lstm=True)[0]
            top2 = torch.topk(lg, 2, dim=-1)
            ids.append(top2.indices[:, 0].numpy().astype(np.int32))
            margins.append((top2.values[:, 0] - top2.values[:, 1]).numpy())
            wins.append(top2.values[:, 0].numpy())
        out["rec_ids"] = np.stack(ids) if ids else np.zeros((0, 160), np.int32)
        out["rec_margin"] = np.stack(margins) if margins else np.zeros((0, 160), np.float32)
        out["rec_win"] = np.stack(wins) if wins else np.zeros((0, 160), np.float32)
        # ---- table structure
        tables = []
        for b in table_boxes:
            x1, y1, x2, y2 = (int(v) for v in b)
            crop = np.ascontiguousarray(page[y1:y2, x1:x2][:, :, ::-1])
            x, meta = lore_pre.lore_preprocess(crop, 1024, 1024)
            z = lore_net.dlaseg_forward(sds["lore"], x)
            logi, ps, polys, results, raw = od.process_detect_output(z, meta, wiz_rev=True, vis_thresh=vis_thresh, return_raw=True)
            n = logi.shape[1]
            t = {"n": n, "polys": np.asarray(polys, np.float32).reshape(-1, 8), "scores": np.asarray(raw[:n, 8], np.float32)}
            if n:
                logic, stacked = lore_processor.processor_forward(sds["proc"], logi, None)
                t["stacked"] = stacked[0].numpy()
                t["logi"] = od.process_logic_output(stacked)[0].numpy()
                sig = torch.sigmoid(z["hm"])[0, 0].numpy()
                inds = od.cell_decode.last_inds[:n]
                H, W = sig.shape
                frag = np.zeros(n, bool)
                for k in range(n):
                    y, x0 = divmod(int(inds[k]), W)
                    nb = max(sig[yy, xx] for yy in range(max(y - 1, 0), min(y + 2, H)) for xx in range(max(x0 - 1, 0), min(x0 + 2, W))
                             if (yy, xx) != (y, x0))
                    s = raw[k, 8]
                    frag[k] = (abs(s - vis_thresh) < 3e-3 or abs(s / 0.4 - vis_thresh) < 3e-3 or abs(sig[y, x0] - vis_thresh) < 3e-3
                               or (sig[y, x0] - nb) < 3e-3)
                t["fragile"] = frag
            tables.append(t)
        out["tables"] = tables
    return out
