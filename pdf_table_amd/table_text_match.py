"""Table structure + recognised text -> table HTML on the host.

Mirrors ``OcrTableToHtmlTask.match_table_cell_and_text_cell`` (model/ocr_pdf/ocr_table_to_html_task.py:178-243) for the
Lore path: the OCR lines whose centre lies in the table box (``get_text_in_table_bbox``, pdf_table/table_common.py
:1303-1325) are each assigned to one cell (``find_top1_mach_box`` :48-77), a cell's lines are put in reading order
(``get_one_cell_text`` :297-330) and joined, and ``cell_to_html`` (table_common.py:578-669) emits the rows with text and
column widths.  The reference evaluates the text x cell pairs in a Python double loop; here the pair tests are numpy
matrices (one [texts, cells] pass per table), the rest is the same order-dependent logic.

Reference behaviours kept on purpose:
* a cell box is (x1, y1, x2, y2) = (bottom-left, top-right) of the quad (``build_table_cell_from_table_unit``
  :1583-1614), so y1 > y2 and ``compute_iou_v2`` is 0 for every pair: among non-containing cells the nearest by the
  corner distance wins;
* ``Cell.text`` APPENDS (pdf_table/table_core.py:366-368): matched text follows the "test_text" the structure stage put
  there; unmatched cells keep it."""
from __future__ import annotations

import re
from typing import Dict, List, Sequence, Tuple

import numpy as np

from .table_html import TableCell

__all__ = ["text_boxes", "texts_in_table", "find_top1_match", "one_cell_text", "ocr_post_process",
           "match_table_cells_and_text", "cells_to_html", "page_table_html"]

_NUM = re.compile(r"[-0-9\.,]")      # MatchUtils.PATTERN_OCR_TEXT_ZH_NUMBER (utils/match_utils.py:52)
_ZERO = re.compile(r"[oO]")          # MatchUtils.PATTERN_OCR_TEXT_0 (:50)


def text_boxes(quads: np.ndarray) -> np.ndarray:
    """OCR quads [t,4,2] or [t,8] -> (min_x, min_y, max_x, max_y) [t,4] like OcrCell.parse (entity/table_entity.py:292-307)"""
    q = np.asarray(quads).reshape(len(quads), 4, 2)
    return np.concatenate([q.min(1), q.max(1)], 1)


def texts_in_table(bbox: Sequence[float], tboxes: np.ndarray, diff: float = 2) -> np.ndarray:
    """indices of the lines whose rounded centre lies in the table box grown by diff (get_text_in_table_bbox)"""
    cx = np.rint((tboxes[:, 0] + tboxes[:, 2]) / 2.0)         # OcrCell.center_point: round() = half to even
    cy = np.rint((tboxes[:, 1] + tboxes[:, 3]) / 2.0)
    m = (bbox[1] - diff <= cy) & (cy <= bbox[3] + diff) & (bbox[0] - diff <= cx) & (cx <= bbox[2] + diff)
    return np.flatnonzero(m)


def find_top1_match(tboxes: np.ndarray, cboxes: np.ndarray, diff: float = 2) -> np.ndarray:
    """for every text box the index of its cell: the FIRST cell that contains it (box_in_other_box, table_common.py
    :138-160), else the first minimum of (1 - iou, distance) (compute_iou_v2 :473-512, distance :435-441)"""
    t = np.asarray(tboxes, np.float64)[:, None, :]            # [T,1,4]  x3,y3,x4,y4
    c = np.asarray(cboxes, np.float64)[None, :, :]            # [1,C,4]  x1,y1,x2,y2
    cy_lo, cy_hi = np.minimum(c[..., 1], c[..., 3]), np.maximum(c[..., 1], c[..., 3])
    ty_lo, ty_hi = np.minimum(t[..., 1], t[..., 3]), np.maximum(t[..., 1], t[..., 3])
    inside = (t[..., 0] >= c[..., 0] - diff) & (t[..., 2] <= c[..., 2] + diff) & (cy_lo - diff <= ty_lo) & (ty_hi <= cy_hi + diff)
    d1 = np.abs(c[..., 0] - t[..., 0]) + np.abs(c[..., 1] - t[..., 1])
    d2 = np.abs(c[..., 2] - t[..., 2]) + np.abs(c[..., 3] - t[..., 3])
    dist = (d1 + d2) + np.minimum(d1, d2)
    ix = np.maximum(np.minimum(t[..., 2], c[..., 2]) - np.maximum(t[..., 0], c[..., 0]), 0)
    iy = np.maximum(np.minimum(t[..., 3], c[..., 3]) - np.maximum(t[..., 1], c[..., 1]), 0)
    inter = ix * iy
    a_t = np.abs((t[..., 2] - t[..., 0]) * (t[..., 3] - t[..., 1]))
    a_c = np.abs((c[..., 2] - c[..., 0]) * (c[..., 3] - c[..., 1]))
    k1 = 1.0 - inter / (a_t + a_c - inter + 1e-6)
    out = np.empty(len(tboxes), np.int64)
    for i in range(len(tboxes)):
        hit = np.flatnonzero(inside[i])
        if len(hit):
            out[i] = hit[0]
            continue
        m = np.flatnonzero(k1[i] == k1[i].min())              # lexicographic minimum, first index on ties
        out[i] = m[np.argmin(dist[i][m])]
    return out


def _merge_close_lines(ar: List[float], line_tol: float) -> List[float]:
    """PdfUtils.merge_close_lines (utils/pdf_utils.py:804-826): moving mean of values within np.isclose(atol=line_tol)"""
    ret: List[float] = []
    for a in ar:
        if not ret:
            ret.append(a)
        elif np.isclose(ret[-1], a, atol=line_tol):
            ret[-1] = (ret[-1] + a) / 2.0
        else:
            ret.append(a)
    return ret


def one_cell_text(tboxes: np.ndarray, texts: Sequence[str]) -> Tuple[List[str], List[int]]:
    """get_one_cell_text: the cell's lines sorted by (merged text row, x1); -> (stripped texts, order)"""
    heights = tboxes[:, 3] - tboxes[:, 1]
    line_tol = (float(sum(heights)) / len(heights) * 1.0) / 3
    y1r = [round(float(v)) for v in tboxes[:, 1]]
    norm = _merge_close_lines(sorted(y1r, reverse=True), line_tol)
    keys = []
    for i in range(len(tboxes)):
        y = tboxes[i, 1]
        for nv in norm:                                        # PdfImageProcessor.find_close_norm_x (first close value)
            if np.isclose(y, nv, atol=line_tol):
                y = nv
                break
        keys.append((float(y), float(tboxes[i, 0])))
    order = sorted(range(len(tboxes)), key=lambda i: keys[i])
    return [str(texts[i]).strip("\n") for i in order], order


def ocr_post_process(text: str) -> str:
    """TableProcessUtils.ocr_post_process (table_common.py:1328-1350): a lone o/O -> "0"; in dotted numbers every dot but
    the last -> ","."""
    new_text = text
    clean = re.sub(r"\s", "", text)
    if len(clean) == 1 and len(_ZERO.findall(clean)) > 0:
        new_text = "0"
    is_num = all(len(_NUM.findall(ch)) > 0 for ch in clean) and str(clean).count(".") > 1
    if is_num:
        dots, right = str(text).count("."), text.rfind(".")
        if dots > 1 and right > -1:
            new_text = f"{text[:right].replace('.', ',')}{text[right:]}"
    return new_text


def match_table_cells_and_text(cells: List[TableCell], tboxes: np.ndarray, texts: Sequence[str],
                               post_process: bool = False) -> List[TableCell]:
    """match_table_cell_and_text_cell up to the HTML: returns the cells sorted by (row_index, col_index), matched cells
    first in the reference's intermediate list (irrelevant after the sort, which is stable)"""
    if len(tboxes):
        cb = np.array([[c.x1, c.y1, c.x2, c.y2] for c in cells], np.float64)
        top1 = find_top1_match(tboxes, cb)
    else:
        top1 = np.zeros(0, np.int64)
    matched: Dict[int, List[int]] = {}
    for ti, ci in enumerate(top1.tolist()):
        matched.setdefault(ci, []).append(ti)
    results = []
    for ci, tis in matched.items():
        show, _ = one_cell_text(np.asarray(tboxes)[tis], [texts[i] for i in tis])
        text = "".join(show)
        if post_process:
            text = ocr_post_process(text)
        cells[ci].text = "".join([cells[ci].text, text])       # Cell.text appends
        results.append(cells[ci])
    results.extend(c for i, c in enumerate(cells) if i not in matched)
    results.sort(key=lambda c: (c.row_index, c.col_index))
    return results


def cells_to_html(cells: List[TableCell]) -> Tuple[List[str], List[str]]:
    """TableProcessUtils.cell_to_html(table_cells) with its defaults (first_header is forced False at :590) ->
    (table_html lines, db_table_html lines)"""
    cells = sorted(cells, key=lambda c: (c.row_index, c.col_index))
    rows: Dict[float, List[TableCell]] = {}
    for c in cells:
        rows.setdefault(c.row_index, []).append(c)
    html_rows = []
    for ri in sorted(rows):
        cols = rows[ri]
        spans = [c.row_span for c in cols if c.row_span > 1]
        drop = len(spans) == len(cols) and len(cols) > 0 and all(s == spans[0] for s in spans)
        one = ["<tr>"]
        for c in cols:
            colspan = f'colspan="{int(c.col_span)}" ' if c.col_span > 1 else ""
            rowspan = f'rowspan="{int(c.row_span)}" ' if c.row_span > 1 else ""
            width = f'width="{round(c.width_ratio * 100)}%"' if round(abs(c.x2 - c.x1)) > 0 else ""
            if drop:
                rowspan = ""
            one.append(f"<td {colspan}{rowspan}{width}>{c.text.replace(chr(10), '<br/>')}</td>")
        one.append("</tr>")
        html_rows.append(one)
    table_html = ['<table border="1">', "<tbody>"]
    for r in html_rows:
        table_html.extend(r)
    table_html += ["</tbody>", "</table>"]
    db = ["<table class='pdf-table' border='1' width='100%'>"]
    for r in html_rows:
        r = ['<tr align="center">'] + r[1:]
        db.append("".join(x.replace("<th ", "<td ").replace("</th>", "</td>") for x in r))
    db.append("</table>")
    return table_html, db


def page_table_html(polygons: np.ndarray, logi: np.ndarray, table_box: Sequence[float], text_quads: np.ndarray,
                    texts: Sequence[str], post_process: bool = True) -> Tuple[List[str], List[str]]:
    """One table of a page -> (table_html, db_table_html).  EVERYTHING is in page pixels: ``polygons`` are the cell quads
    after the reference's shift by the crop corner (convert_table_sep_to_merge, table_common.py:1811-1825), ``table_box``
    the layout box the crop was cut from, ``text_quads`` [t,8] the page's detected lines with their ``texts``.  Lines whose
    centre is in the table box are assigned to cells; image pages use ocr_post_process=True (ocr_table_to_html_task.py:93)."""
    from .table_html import table_cells_from_logits
    cells = table_cells_from_logits(polygons, logi)
    tbx = text_boxes(text_quads) if len(text_quads) else np.zeros((0, 4))
    inside = texts_in_table([float(v) for v in table_box], tbx, diff=2) if len(tbx) else np.zeros(0, np.int64)
    res = match_table_cells_and_text(cells, tbx[inside], [texts[i] for i in inside], post_process=post_process)
    return cells_to_html(res)
