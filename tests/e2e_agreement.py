"""Agreement of an engine run with the committed end-to-end oracle fixture (tests/golden/e2e_page.npz: the composed fp32 oracle chain on
two 1024 x 1024 synthetic pages, make_golden.py::gen_e2e_page) as FRACTIONS -- the figures VERDICT r03 asked for the headline (bf16)
mode, whose per-stage tests assert drift bounds rather than equality.  Test / bench infrastructure (a checker, like oracle/): used by
tests/test_gpu_e2e.py (both modes) and by bench.py's `tolerance_mode.bf16_e2e_agreement` leg; nothing in pdf_table_amd/ imports it.

What is counted, per the reference's three outputs (north_star: text-box polygons, recognised strings, cell-adjacency HTML):
  boxes      oracle text boxes found by the engine: identical int16 quads / within 2 px on every coordinate
  strings    lines cut from a matched quad whose recognised string equals the oracle's (on identical quads; on quads within 2 px)
  cells      oracle table cells found with >= 3 of 4 vertices within 0.1 / 1 / 4 / 16 px; logical locations equal on the matched ones
  html       tables whose HTML equals the ORACLE CHAIN's -- the reference's own structure + text -> HTML functions run over the oracle's cells, boxes
             and strings when the fixture was generated (one differing token anywhere in the table breaks it) --, and tables whose STRUCTURE
             string (rows, spans; no text) is equal
"""
import numpy as np


def match_rows(want, got, tol):
    """greedy one-to-one matching of rows by max |difference| <= tol -> list of (i_want, j_got)"""
    used = np.zeros(len(got), bool)
    pairs = []
    for i, w in enumerate(want):
        if not len(got):
            break
        d = np.abs(got - w).max(1)
        d[used] = np.inf
        j = int(np.argmin(d))
        if d[j] <= tol:
            used[j] = True
            pairs.append((i, j))
    return pairs


def match_cells(want, got, tol=0.1):
    """one-to-one matching of cell quads: a pair matches when at least three of its four vertices agree within tol (one vertex may have
    been snapped to a different corner point) -> (pairs, indices of the pairs with a differing vertex)"""
    used = np.zeros(len(got), bool)
    pairs, odd = [], []
    for i, w in enumerate(want):
        if not len(got):
            break
        d = np.abs(got - w).reshape(len(got), 4, 2).max(2)          # per vertex
        ok = (d <= tol).sum(1)
        ok[used] = -1
        j = int(np.argmax(ok))
        if ok[j] >= 3:
            used[j] = True
            pairs.append((i, j))
            if ok[j] < 4:
                odd.append(len(pairs) - 1)
    return pairs, odd


def oracle_strings(g, pi, label):
    from pdf_table_amd.rec_stage import ctc_collapse
    ids = g[f"p{pi}_rec_ids"]
    out = ["".join(label.get(t, "") for t in row) for row in ctc_collapse(ids)]
    if f"p{pi}_rec_text" in g:          # the strings the fixture's HTML was built from (make_golden.py: the oracle's own greedy decode)
        assert out == [str(x) for x in g[f"p{pi}_rec_text"]]
    return out


def agreement(g, results, label, table_boxes):
    """g: the loaded fixture; results: PageResult list of the fixture's pages (OcrTablePipeline.predict, table_html=True); label: the
    recogniser's id -> char map; table_boxes: per page int [t,4] regions the fixture's tables were cut from.  -> dict of counts and fractions."""
    from pdf_table_amd.table_html import structure_html
    from pdf_table_amd.table_text_match import page_table_html
    c = dict(boxes=0, boxes_engine=0, boxes_identical=0, boxes_within_2px=0, lines_on_identical_quads=0, strings_identical_on_identical_quads=0,
             lines_on_2px_quads=0, strings_identical_on_2px_quads=0, chars=0, chars_equal=0, cells=0, cells_engine=0, cells_matched_0p1px=0, cells_matched_1px=0, cells_matched_4px=0, cells_matched_16px=0, tables_structure_identical=0,
             logi_rows_compared=0, logi_rows_equal=0, tables=0, tables_cells_and_logi_identical=0, tables_html_identical=0)
    for pi, r in enumerate(results):
        want = g[f"p{pi}_det_boxes"]
        got = np.asarray(r.det_result, np.float32).reshape(-1, 8)
        ref_txt = oracle_strings(g, pi, label)
        same, near = match_rows(want, got, 0.0), match_rows(want, got, 2.0)
        c["boxes"] += len(want)
        c["boxes_engine"] += len(got)
        c["boxes_identical"] += len(same)
        c["boxes_within_2px"] += len(near)
        for pairs, kl, ks in ((same, "lines_on_identical_quads", "strings_identical_on_identical_quads"),
                              (near, "lines_on_2px_quads", "strings_identical_on_2px_quads")):
            c[kl] += len(pairs)
            c[ks] += sum(1 for i, j in pairs if r.ocr_result[j]["text"] == ref_txt[i])
        for i, j in near:           # character level: positions that carry the oracle's character (a whole string fails on ONE flipped token)
            a, b = r.ocr_result[j]["text"], ref_txt[i]
            c["chars"] += max(len(a), len(b))
            c["chars_equal"] += sum(1 for x, y in zip(a, b) if x == y)
        tbs = table_boxes[pi]
        tsr = r.table_structure_result or []
        c["tables"] += int(g[f"p{pi}_n_tables"])
        for ti in range(int(g[f"p{pi}_n_tables"])):
            k = f"p{pi}_t{ti}_"
            polys, logi = g[k + "polys"].astype(np.float64), g[k + "logi"]
            c["cells"] += len(polys)
            if ti >= len(tsr):
                continue
            t = tsr[ti]
            off = np.tile(tbs[ti][:2].astype(np.float64), 4)[None]
            gp = np.asarray(t["polygons"], np.float64).reshape(-1, 8) - off
            gl = np.asarray(t["logi"]).reshape(-1, 4)
            c["cells_engine"] += len(gp)
            tight, _ = match_cells(polys, gp, 0.1)
            loose, _ = match_cells(polys, gp, 1.0)
            c["cells_matched_0p1px"] += len(tight)
            c["cells_matched_1px"] += len(loose)
            c["cells_matched_4px"] += len(match_cells(polys, gp, 4.0)[0])
            c["cells_matched_16px"] += len(match_cells(polys, gp, 16.0)[0])
            c["tables_structure_identical"] += int(len(gp) > 0 and structure_html(polys + off, logi) == structure_html(gp + off, gl))
            if loose:
                wi, gj = np.array([i for i, _ in loose]), np.array([j for _, j in loose])
                eq = (gl[gj] == logi[wi]).all(1)
                c["logi_rows_compared"] += len(eq)
                c["logi_rows_equal"] += int(eq.sum())
            same_cells = len(tight) == len(polys) == len(gp) and all(i == j for i, j in tight) and np.array_equal(gl, logi)
            c["tables_cells_and_logi_identical"] += int(same_cells)
            # the ORACLE CHAIN's HTML: the reference's own OcrTableToHtmlTask code over the oracle's cells, boxes and strings, stored in the fixture
            # (make_golden.py::gen_e2e_page) -- nothing of the product on this side of the comparison
            ref_html = [str(x) for x in g[k + "html"]]
            c["tables_html_identical"] += int(list(t.get("table_html") or []) == ref_html)

    def fr(a, b):
        return round(c[a] / c[b], 4) if c[b] else None
    c["frac"] = {"boxes_identical": fr("boxes_identical", "boxes"), "boxes_within_2px": fr("boxes_within_2px", "boxes"),
                 "strings_identical_on_identical_quads": fr("strings_identical_on_identical_quads", "lines_on_identical_quads"),
                 "strings_identical_on_2px_quads": fr("strings_identical_on_2px_quads", "lines_on_2px_quads"),
                 "chars_equal_on_2px_quads": fr("chars_equal", "chars"),
                 "cells_matched_0p1px": fr("cells_matched_0p1px", "cells"), "cells_matched_1px": fr("cells_matched_1px", "cells"),
                 "cells_matched_4px": fr("cells_matched_4px", "cells"), "cells_matched_16px": fr("cells_matched_16px", "cells"),
                 "tables_structure_identical": fr("tables_structure_identical", "tables"),
                 "logi_rows_equal_on_matched_cells": fr("logi_rows_equal", "logi_rows_compared"),
                 "tables_html_identical": fr("tables_html_identical", "tables")}
    return c
