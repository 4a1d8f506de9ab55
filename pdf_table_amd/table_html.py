"""Logical locations -> table cells -> structure HTML (the north star's third output) on the host.

Mirrors the reference's result shaping for Lore (model/ocr_pdf/ocr_table_structure_task.py:276-303):
``TableProcessUtils.get_table_cell_from_table_logit`` -> ``build_table_cell_from_axis`` (pdf_table/table_common.py:1568-1663):
``TableEval`` sorts the cells with a bubble sort on (top, left, bottom, right) (entity/table_entity.py:631-657 -- a stable
sort, done here with ``sorted``), ``build_table_cell_from_table_unit`` (:1583-1614) turns every unit into a cell with
1-based row / column index and spans, and ``cell_to_html(first_header=False, add_width=False, add_text=False)``
(:578-669) emits the rows.  The logical columns are read as [left, right, top, bottom] (``TableUnit`` :546-553).
"""
from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass
from typing import Dict, List

import numpy as np

__all__ = ["TableCell", "table_cells_from_logits", "cells_to_structure_html", "structure_html"]


@dataclass
class TableCell:
    """the fields of pdf_table/table_core.py:240-300 ``Cell`` that the structure path fills"""
    x1: float
    y1: float
    x2: float
    y2: float
    row_index: float
    col_index: float
    row_span: float
    col_span: float
    width_ratio: float
    height_ratio: float
    lt: tuple = ()
    rb: tuple = ()
    text: str = "test_text"


def _dist(a, b) -> float:
    return float(np.sqrt((float(a[0]) - float(b[0])) ** 2 + (float(a[1]) - float(b[1])) ** 2))


def table_cells_from_logits(polygons: np.ndarray, logi: np.ndarray) -> List[TableCell]:
    """polygons [n,8] (x1,y1 .. x4,y4: TL, TR, BR, BL), logi [n,4] = [left, right, top, bottom] -> cells in the
    reference's order (sorted by top, left, bottom, right; ties keep the input order)."""
    n = len(polygons)
    if n == 0:
        return []
    order = sorted(range(n), key=lambda i: (logi[i][2], logi[i][0], logi[i][3], logi[i][1]))
    p = np.asarray(polygons)
    first, last = p[order[0]], p[order[-1]]
    table_w = last[4] - first[0]               # point3.x of the last unit - point1.x of the first
    table_h = last[5] - first[1]
    cells = []
    for i in order:
        b = p[i]
        left, right, top, bottom = logi[i][0], logi[i][1], logi[i][2], logi[i][3]
        col_geom = (_dist(b[0:2], b[2:4]) + _dist(b[4:6], b[6:8])) / 2      # TableBBox.col_span (table_entity.py:517-518)
        row_geom = (_dist(b[0:2], b[6:8]) + _dist(b[2:4], b[4:6])) / 2
        with np.errstate(divide="ignore", invalid="ignore"):
            wr, hr = col_geom / table_w, row_geom / table_h
        cells.append(TableCell(x1=b[6], y1=b[7], x2=b[2], y2=b[3], row_index=top + 1, col_index=left + 1,
                               row_span=bottom - top + 1, col_span=right - left + 1, width_ratio=wr, height_ratio=hr,
                               lt=(b[0], b[1]), rb=(b[4], b[5])))
    return cells


def cells_to_structure_html(cells: List[TableCell]) -> str:
    """cell_to_html(first_header=False, add_width=False, add_text=False) joined and wrapped like show_results does
    (ocr_table_structure_task.py:296-303)."""
    cells = sorted(cells, key=lambda c: (c.row_index, c.col_index))          # convert_table_cell_to_dict (:540-554)
    rows: Dict[float, List[TableCell]] = defaultdict(list)
    for c in cells:
        rows[c.row_index].append(c)
    out = ['<table border="1">', "<tbody>"]
    for ri in sorted(rows):
        cols = rows[ri]
        spans = [c.row_span for c in cols if c.row_span > 1]
        same = all(s == spans[0] for s in spans)
        drop_rowspan = len(spans) == len(cols) and len(cols) > 0 and same        # fix_row_span_same (:617-619)
        out.append("<tr>")
        for c in cols:
            colspan = f'colspan="{int(c.col_span)}" ' if c.col_span > 1 else ""
            rowspan = f'rowspan="{int(c.row_span)}" ' if c.row_span > 1 else ""
            if drop_rowspan:
                rowspan = ""
            out.append(f"<td {colspan}{rowspan}></td>")
        out.append("</tr>")
    out += ["</tbody>", "</table>"]
    html = "".join(out)
    return f"<html><body>{html}</body></html>".replace("<td >", "<td>").replace("<tbody>", "").replace("</tbody>", "")


def structure_html(polygons: np.ndarray, logi: np.ndarray) -> str:
    """same string as cells_to_structure_html(table_cells_from_logits(..)) without building the cell objects: the HTML
    only needs the sorted (row, col, row_span, col_span) of every cell"""
    lg = np.asarray(logi)
    n = len(lg)
    if n == 0:
        return cells_to_structure_html([])
    left, right, top, bottom = lg[:, 0], lg[:, 1], lg[:, 2], lg[:, 3]
    order = np.lexsort((right, bottom, left, top))               # stable: (top, left, bottom, right)
    row = (top + 1)[order]
    col = (left + 1)[order]
    rs = (bottom - top + 1)[order]
    cs = (right - left + 1)[order]
    o2 = np.lexsort((col, row))                                  # stable re-sort by (row_index, col_index)
    row, rs, cs = row[o2], rs[o2].tolist(), cs[o2].tolist()
    starts = np.flatnonzero(np.r_[True, row[1:] != row[:-1]]).tolist() + [n]
    out = ['<html><body><table border="1">']
    for a, b in zip(starts[:-1], starts[1:]):
        spans = [x for x in rs[a:b] if x > 1]
        drop = len(spans) == b - a and all(x == spans[0] for x in spans)
        out.append("<tr>")
        for k in range(a, b):
            c_, r_ = cs[k], rs[k]
            if c_ > 1 or (r_ > 1 and not drop):
                out.append("<td " + (f'colspan="{int(c_)}" ' if c_ > 1 else "") + (f'rowspan="{int(r_)}" ' if r_ > 1 and not drop else "")
                           + "></td>")
            else:
                out.append("<td></td>")
        out.append("</tr>")
    out.append("</table></body></html>")
    return "".join(out)
