"""Host half of MtlTabNet (second half): label convertor, HTML post-processor and the stage that drives the
engine's pre-processing kernel, backbone and KV-cached decoders over a batch of tables.

Reference (all under src/pdftable/model/):
  * ``MtlTabNetConvertor`` -- table/mtl_tabnet/master_convertor.py:271-784 (test time: ``output_format`` :756-784 with ``tensor2idx``
    :236-268, ``idx2str`` :431-445, ``_get_pred_bbox_mask`` :586-619, ``_filter_invalid_bbox`` :621-641, ``_decode_bboxes`` :643-674,
    ``_adjsut_bboxes_len`` :676-683, ``tensor2idx_cell`` :551-584, ``idx2str_cell`` :447-462, ``_get_strings_scores`` :685-699);
  * ``MasterPostProcessor`` -- table/mtl_tabnet/master_post_processor.py:14-401 (``text_to_list``, ``merge_span_token``,
    ``insert_text_to_token``, ``deal_eb_token``, ``deal_bb`` with ``deal_duplicate_bb`` / ``deal_isolate_span``);
  * ``MtlTabNetPostProcessor.__call__`` -- mtl_tabnet/processor_mtl_tabnet.py:108-131 (result dict);
  * ``MtlTabNet.simple_test`` -- table/mtl_tabnet/table_master.py:557-590; the test pipeline of mtl_tabnet_config.py:136-160.

Everything here is pinned to the reference's own classes on seeded decoder outputs (tests/golden/mtl_tabnet_host.json,
tests/test_mtl_host.py), quirks included:
  * the structure vocabulary holds ``colspan="2"`` WITHOUT the leading blank ``merge_span_token`` looks for, so span cells come out
    as ``<tdcolspan="2"></td>``;
  * a table with exactly ONE predicted cell has no cell strings (``out_cell_i.size(0) == 1`` is how the convertor recognises the
    ``torch.zeros(1)`` placeholder of a table without cells, master_convertor.py:771-773);
  * the box mask is aligned with the token list AFTER ``<PAD>`` tokens were dropped, the boxes are per position;
  * a table none of whose boxes sums to more than one pixel raises ``IndexError`` in ``box_transform`` (``strict=True`` keeps that).
"""
from __future__ import annotations

import json
import os
import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

__all__ = ["MtlTabNetConvertor", "MasterPostProcessor", "MtlStage", "MtlTabnetConfig", "load_alphabets", "mtl_result", "mtl_image_meta"]

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "mtl_tabnet_alphabet.json")


class MtlTabnetConfig:
    """``MtlTabnetConfig`` (model/mtl_tabnet/configuration_mtl_tabnet.py:23-91) reduced to what the engine path reads, with the
    numbers of table/mtl_tabnet/mtl_tabnet_config.py:12-18,136-160 (sequence limits 500 / 150, 480 x 480 test pipeline)."""

    def __init__(self, model_name: str = "MtlTabNet", backbone: str = "TableResNetExtra", task_type: str = "PubTabNet", model_path: str = "",
                 debug: bool = True, **kwargs):
        self.model_name, self.backbone, self.model_path, self.debug = model_name, backbone, model_path, debug
        self.task_type = "FinTabNet" if (task_type in ["FinTabNet", "fin"] and model_name == "MtlTabNet") else "PubTabNet"
        self.model_provider = "Other"
        self.predictor_type = "hip"
        self.max_seq_len, self.max_seq_len_cell, self.size = 500, 150, 480


def load_alphabets() -> Tuple[List[str], List[str]]:
    """(structure tokens, cell-content tokens) of the PubTabNet checkpoints (table/mtl_tabnet/mtl_tabnet_constants.py; a data file
    of this package, its sha256 pinned in tests/golden/mtl_tabnet_alphabet_hash.json)"""
    with open(_DATA, "rb") as f:
        d = json.loads(f.read().decode("utf-8"))
    return list(d["structure"]), list(d["cell"])


def _softmax_max(logits: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """arg-max and its soft-max probability along the last axis, float32 like ``seq.softmax(-1)`` / ``torch.max``"""
    x = np.asarray(logits, dtype=np.float32)
    idx = x.argmax(-1)
    e = np.exp(x - x.max(-1, keepdims=True))
    return idx, (np.take_along_axis(e, idx[..., None], -1)[..., 0] / e.sum(-1)).astype(np.float32)


class MtlTabNetConvertor:
    """ids <-> tokens for the structure and the cell-content vocabularies, and the test-time output formatting."""

    def __init__(self, dict_file: Optional[Sequence[str]] = None, cell_dict_file: Optional[Sequence[str]] = None, max_seq_len: int = 500,
                 max_seq_len_cell: int = 150, with_unknown: bool = True, start_end_same: bool = False, **kwargs):
        if start_end_same:
            raise AssertionError("MtlTabNetConvertor needs start_end_same = False")          # checker(), :368-372
        if dict_file is None or cell_dict_file is None:
            s, c = load_alphabets()
            dict_file = s if dict_file is None else dict_file
            cell_dict_file = c if cell_dict_file is None else cell_dict_file
        self.max_seq_len, self.max_seq_len_cell, self.with_unknown, self.start_end_same = max_seq_len, max_seq_len_cell, with_unknown, False
        self.idx2char, sp = self._vocabulary(dict_file, with_unknown)
        self.unknown_idx, self.start_idx, self.end_idx, self.padding_idx = sp
        self.idx2char_cell, sp = self._vocabulary(cell_dict_file, with_unknown)
        self.unknown_idx_cell, self.start_idx_cell, self.end_idx_cell, self.padding_idx_cell = sp
        self.char2idx = {c: i for i, c in enumerate(self.idx2char)}
        self.char2idx_cell = {c: i for i, c in enumerate(self.idx2char_cell)}
        self.has_cell = True

    @staticmethod
    def _vocabulary(lines: Sequence[str], with_unknown: bool):
        """BaseConvertor.__init__ (:34-60: strip only the newline, drop empty lines) + update_dict (:170-196): <UKN>, <SOS>, <EOS>, <PAD>"""
        toks = [t for t in (ln.strip("\n") for ln in lines) if t != ""]
        unk = None
        if with_unknown:
            toks.append("<UKN>")
            unk = len(toks) - 1
        toks += ["<SOS>", "<EOS>", "<PAD>"]
        return toks, (unk, len(toks) - 3, len(toks) - 2, len(toks) - 1)

    def num_classes(self) -> int:
        return len(self.idx2char)

    def num_classes_cell(self) -> int:
        return len(self.idx2char_cell)

    def idx_tag_cell(self) -> List[int]:
        return [self.char2idx["<td></td>"], self.char2idx["<td"]]

    def decoder_cfg(self) -> Dict:
        """the integers pack_mtl_decoder stores beside the weights (update_decoder_config, table_master.py:321-340)"""
        if not self.has_cell:
            return dict(N=3, sos=self.start_idx, eos=self.end_idx, pad=self.padding_idx, max_len=self.max_seq_len, idx_tag_cell=self.idx_tag_cell())
        return dict(N=3, sos=self.start_idx, eos=self.end_idx, pad=self.padding_idx, max_len=self.max_seq_len, sos_cell=self.start_idx_cell,
                    eos_cell=self.end_idx_cell, pad_cell=self.padding_idx_cell, max_len_cell=self.max_seq_len_cell, idx_tag_cell=self.idx_tag_cell())

    # -- test-time formatting of ONE table (the reference always runs a batch of one, processor_mtl_tabnet.py:84-89) -------------------
    @staticmethod
    def _kept(ids: Sequence[int], probs: Sequence[float], pad: int, eos: int):
        """ids / probabilities up to the first <EOS>, <PAD> positions dropped (the reference's per-element loop, as two array operations)"""
        ids, probs = np.asarray(ids), np.asarray(probs)
        stop = np.flatnonzero(ids == eos)
        if len(stop):
            ids, probs = ids[:stop[0]], probs[:stop[0]]
        keep = ids != pad
        return ids[keep].tolist(), probs[keep].astype(np.float64).tolist()

    @staticmethod
    def _mean(v: List[float]) -> float:
        return sum(v) / len(v) if len(v) > 0 else 0.0

    def format_ids(self, tag_ids, tag_prob, boxes, cell_ids, cell_prob, img_meta: Dict):
        """tag_ids / tag_prob [L] (arg-max and soft-max probability per decoded position), boxes float32 [L, 4], cell_ids / cell_prob
        [n_cells, steps] (or None) -> (string, score, pred_bbox float64 [len(tokens), 4], cell_strings, cell_scores)."""
        ids, probs = self._kept(tag_ids, tag_prob, self.padding_idx, self.end_idx)
        tokens = [self.idx2char[i] for i in ids]
        string = ",".join(tokens)
        score = self._mean(probs)
        # _get_pred_bbox_mask works on string.split(','): '' -> [''] (one entry, mask 0)
        parts = string.split(",")
        mask = []
        for t in parts:
            if t == "<EOS>":
                mask.append(0)
                break
            mask.append(1 if t in ("<td></td>", "<td") else 0)
        box = np.asarray(boxes, dtype=np.float32)
        inside = ((box >= 0.0) * 1 + (box <= 1.0) * 1).sum(1)
        valid = np.where(inside == 8, 1, 0)
        padded = np.zeros(box.shape[0], dtype="int64")
        padded[:len(mask)] = mask                      # the reference lets numpy raise when the mask is longer than the boxes
        dec = box * valid[:, None] * padded[:, None]   # float32 * int64 -> float64, as in the reference
        pad_shape, sf = img_meta["pad_shape"], img_meta["scale_factor"]
        dec[:, 0::2] = dec[:, 0::2] * pad_shape[1]
        dec[:, 1::2] = dec[:, 1::2] * pad_shape[0]
        dec[:, 0::2] = dec[:, 0::2] / sf[1]
        dec[:, 1::2] = dec[:, 1::2] / sf[0]
        dec = dec[:len(parts), :]
        cell_strings, cell_scores = [], []
        if cell_ids is not None and len(cell_ids) > 1:      # size(0) == 1 is the no-cells placeholder AND a single real cell
            # _kept for every cell at once: positions before the row's first <EOS> that are not <PAD>; the score is the reference's left-to-right
            # float64 sum of the kept probabilities (a cumulative sum adds them in that order, and + 0.0 for a dropped position is exact)
            ci, cp = np.asarray(cell_ids), np.asarray(cell_prob, dtype=np.float64)
            keep = (np.cumsum(ci == self.end_idx_cell, axis=1) == 0) & (ci != self.padding_idx_cell)
            cnt = keep.sum(1)
            tot = np.cumsum(np.where(keep, cp, 0.0), axis=1)[:, -1] if ci.shape[1] else np.zeros(len(ci))
            table = self.idx2char_cell
            for row, km, n_kept, t in zip(ci.tolist(), keep.tolist(), cnt.tolist(), tot.tolist()):
                cell_strings.append("".join([table[i] for i, k in zip(row, km) if k]))
                cell_scores.append(t / n_kept if n_kept > 0 else 0.0)
        return string, score, dec, cell_strings, cell_scores

    def output_format(self, outputs, out_bbox, out_cell, img_metas=None):
        """``output_format`` on raw tensors: outputs [N, T, C] logits, out_bbox [N, T, 4], out_cell a list of [n_cells, steps, C_cell]
        logits (or a 1-element placeholder) -- numpy arrays or anything ``np.asarray`` takes."""
        strings, scores, bboxes, cells, cell_scores = [], [], [], [], []
        for b in range(len(outputs)):
            ti, tp = _softmax_max(np.asarray(outputs[b]))
            oc = np.asarray(out_cell[b])
            ci = cp = None
            if oc.ndim == 3:
                ci, cp = _softmax_max(oc)
            s, sc, bb, cs, css = self.format_ids(ti, tp, np.asarray(out_bbox[b]), ci, cp, img_metas[b])
            strings.append(s), scores.append(sc), bboxes.append(bb), cells.append(cs), cell_scores.append(css)
        return strings, scores, bboxes, cells, cell_scores


class TableMasterConvertor(MtlTabNetConvertor):
    """``TableMasterConvertor`` (table/mtl_tabnet/master_convertor.py:787-1070): the structure vocabulary only -- TableMaster has no cell-content decoder
    (master_decoder.py:532-563), so ``output_format`` returns (strings, scores, boxes) and the post-processor gets no cell texts."""

    def __init__(self, dict_file: Optional[Sequence[str]] = None, max_seq_len: int = 500, with_unknown: bool = True, start_end_same: bool = False, **kwargs):
        super().__init__(dict_file=dict_file, cell_dict_file=[], max_seq_len=max_seq_len, max_seq_len_cell=0, with_unknown=with_unknown,
                         start_end_same=start_end_same)
        self.has_cell = False

    def num_classes_cell(self) -> int:
        return 0

    def output_format(self, outputs, out_bbox, img_metas=None):
        strings, scores, bboxes = [], [], []
        for b in range(len(outputs)):
            ti, tp = _softmax_max(np.asarray(outputs[b]))
            s, sc, bb, _, _ = self.format_ids(ti, tp, np.asarray(out_bbox[b]), None, None, img_metas[b])
            strings.append(s), scores.append(sc), bboxes.append(bb)
        return strings, scores, bboxes


# ----------------------------------------------------------------------------------------------------------------------------------
# master_post_processor.py: structure tokens (+ cell texts) -> HTML
# ----------------------------------------------------------------------------------------------------------------------------------
_N = r'"(\d)+"'
_SPAN_ATTR = rf' rowspan={_N} colspan={_N}| colspan={_N} rowspan={_N}| rowspan={_N}| colspan={_N}'
_RE_SPAN_ATTR = re.compile(_SPAN_ATTR)
_RE_SPAN_OPEN = re.compile("|".join(f"<td{a}>" for a in _SPAN_ATTR.split("|")))
_RE_ISOLATED = re.compile("|".join(f"<td></td>{a}></b></td>" for a in _SPAN_ATTR.split("|")))
_RE_TD = re.compile("|".join(f"<td{a}>(.+?)</td>" for a in _SPAN_ATTR.split("|")) + "|<td>(.*?)</td>")
_EMPTY_BOX_TOKENS = [("<eb></eb>", "<td></td>"), ("<eb1></eb1>", "<td> </td>"), ("<eb2></eb2>", "<td><b> </b></td>"),
                     ("<eb3></eb3>", "<td>\u2028\u2028</td>"), ("<eb4></eb4>", "<td><sup> </sup></td>"), ("<eb5></eb5>", "<td><b></b></td>"),
                     ("<eb6></eb6>", "<td><i> </i></td>"), ("<eb7></eb7>", "<td><b><i></i></b></td>"),
                     ("<eb8></eb8>", "<td><b><i> </i></b></td>"), ("<eb9></eb9>", "<td><i></i></td>"),
                     ("<eb10></eb10>", "<td><b> \u2028 \u2028 </b></td>")]


def text_to_list(master_token: str) -> List[str]:
    """the ','-joined structure string -> token list closed by ``</tr>``, ``</tbody>`` (:283-296)"""
    toks = master_token.split(",")
    if toks[-1] == "<td></td>":
        toks += ["</tr>", "</tbody>"]
    elif toks[-1] != "</tbody>":
        toks.append("</tbody>")
    if toks[-2] != "</tr>":
        toks.insert(-1, "</tr>")
    return toks


def merge_span_token(toks: List[str]) -> List[str]:
    """``<td`` + `` rowspan=..`` [+ `` colspan=..``] + ``>`` + ``</td>`` -> one token, everything up to the first ``</tbody>`` (:163-214).
    Like the reference it appends ``</tbody>`` to the CALLER's list when that is missing."""
    if toks[-1] != "</tbody>":
        toks.append("</tbody>")
    span = lambda t: t.startswith(" colspan=") or t.startswith(" rowspan=")
    out, p, n = [], 0, len(toks)
    while p < n and toks[p] != "</tbody>":
        take = 1
        if toks[p] == "<td":
            if p + 5 <= n and span(toks[p + 2]):
                take = 5
            elif p + 4 <= n and span(toks[p + 1]):
                take = 4
        out.append("".join(toks[p:p + take]))
        p += take
    out.append("</tbody>")
    return out


def deal_eb_token(tok: str) -> str:
    if "<eb" not in tok:
        return tok
    for a, b in _EMPTY_BOX_TOKENS:
        tok = tok.replace(a, b)
    return tok


def insert_text_to_token(toks: List[str], cell_content_list: Optional[List[str]] = None) -> str:
    """cell texts go between ``>`` and ``<`` of the ``<td`` tokens in order; a ``<td`` token beyond the last text is DROPPED (:260-280)"""
    merged, used = [], 0
    for tok in merge_span_token(toks):
        if tok.startswith("<td") and cell_content_list is not None:
            if used > len(cell_content_list) - 1:
                used += 1
                continue
            tok = tok.replace("><", ">{}<".format(cell_content_list[used]))
            used += 1
        merged.append(deal_eb_token(tok))
    return "".join(merged)


def deal_isolate_span(part: str) -> str:
    for item in [m.group() for m in _RE_ISOLATED.finditer(part)]:
        part = part.replace(item, "<td{}></td>".format(_RE_SPAN_ATTR.search(item).group()))
    return part


def deal_duplicate_bb(part: str) -> str:
    tds = [m.group() for m in _RE_TD.finditer(part)]
    for td in tds:
        new = td
        if td.count("<b>") > 1 or td.count("</b>") > 1:
            new = td.replace("<b>", "").replace("</b>", "").replace("<td>", "<td><b>").replace("</td>", "</b></td>")
        part = part.replace(td, new)
    return part


def deal_bb(result_token: str, tag_: str = "thead") -> str:
    """bold every cell of the FIRST ``<tag_>...</tag_>`` section (:90-160)"""
    m = re.search("<" + tag_ + ">(.*?)</" + tag_ + ">", result_token)
    if m is None:
        return result_token
    origin = part = m.group()
    spans = [s.group() for s in _RE_SPAN_OPEN.finditer(part)]
    if not spans:
        part = part.replace("<td>", "<td><b>").replace("</td>", "</b></td>").replace("<b><b>", "<b>").replace("</b></b>", "</b>")
    else:
        for sp in spans:                                # a span that occurs twice is replaced twice; the doubled <b> collapses below
            part = part.replace(sp, sp.replace(">", "><b>"))
        part = part.replace("</td>", "</b></td>")
        part = re.sub("(<b>)+", "<b>", part)
        part = re.sub("(</b>)+", "</b>", part)
        part = part.replace("<td>", "<td><b>").replace("<b><b>", "<b>")
    part = part.replace("<td><b></b></td>", "<td></td>")
    part = deal_isolate_span(deal_duplicate_bb(part))
    return result_token.replace(origin, part)


def html_post_process(text: str) -> str:
    return '<html><body><table border="1">' + text + "</table></body></html>"


class MasterPostProcessor:
    """``MasterPostProcessor`` (:326-401) without the file side effects.  ``strict``: keep the reference's IndexError for a table
    without a surviving box; otherwise such a table gets empty ``bbox`` / ``new_bbox`` arrays."""

    def __init__(self, output_dir=None, strict: bool = True):
        self.output_dir, self.strict = output_dir, strict

    @staticmethod
    def _structure_tokens(pred_structure: str) -> List[str]:
        return merge_span_token(text_to_list(pred_structure))

    def get_table_structure(self, pred_structure: str) -> str:
        html = "".join(self._structure_tokens(pred_structure))
        return html_post_process(deal_bb(deal_bb(html, "thead"), "tbody"))

    def get_table_structure_list(self, pred_structure: str) -> List[str]:
        return ["<html>", "<body>", "<table>"] + self._structure_tokens(pred_structure) + ["</table>", "</body>", "</html>"]

    def box_transform(self, bboxes: np.ndarray) -> np.ndarray:
        """(cx, cy, w, h) -> (x1, y1, x2, y2), truncated to int32"""
        if bboxes.ndim != 2:
            if self.strict:
                raise IndexError("index 0 is out of bounds for axis 0 with size 0")    # np.array([])[..., 0] in the reference (:360)
            return np.zeros((0, 4), np.int32)
        half = bboxes[..., 2:4] / 2
        return np.concatenate([bboxes[..., 0:2] - half, bboxes[..., 0:2] + half], -1).astype(np.int32)

    def __call__(self, result: Dict, file_name=None) -> Dict:
        text, cells = result["text"], result.get("cell", None)
        bb = np.asarray(result["bbox"])
        # rows whose coordinates sum to more than 1 (the reference filters row by row with Python's sum: the same left-to-right float64 additions)
        kept = bb[((bb[:, 0] + bb[:, 1]) + bb[:, 2]) + bb[:, 3] > 1] if bb.ndim == 2 and bb.shape[1] == 4 and len(bb) else \
            np.array([row for row in result["bbox"] if sum(row) > 1])
        result["bbox"] = kept if len(kept) else np.array([])      # no surviving box: the reference's empty 1-D array (box_transform then raises)
        html = insert_text_to_token(text_to_list(text), cells)
        html = deal_bb(deal_bb(html, "thead"), "tbody")
        result["pred_html"], result["html_context"] = html, html_post_process(html)
        toks = self._structure_tokens(text)      # once for both forms (get_table_structure / get_table_structure_list)
        result["structure_str"] = html_post_process(deal_bb(deal_bb("".join(toks), "thead"), "tbody"))
        result["structure_str_list"] = ["<html>", "<body>", "<table>"] + toks + ["</table>", "</body>", "</html>"]
        result["new_bbox"] = self.box_transform(result["bbox"])
        return result


def two_point_to_four_point(bbox_list):
    """``OcrCommonUtils.box_list_two_point_to_four_point`` (utils/ocr/ocr_common_utils.py:569-579)"""
    if len(bbox_list) > 0 and len(bbox_list[0]) == 4:
        b = np.asarray(bbox_list)
        return b[:, [0, 1, 2, 1, 2, 3, 0, 3]]
    return bbox_list


def mtl_result(convertor: MtlTabNetConvertor, post: MasterPostProcessor, tag_ids, tag_prob, boxes, cell_ids, cell_prob, img_meta: Dict,
               inputs=None) -> Dict:
    """decoder outputs of one table -> the reference's result dict (``MtlTabNet.simple_test`` -> ``MtlTabNetPostProcessor.__call__``)"""
    s, sc, bb, cs, _ = convertor.format_ids(tag_ids, tag_prob, boxes, cell_ids, cell_prob, img_meta)
    if not getattr(convertor, "has_cell", True):
        cs = None                                    # TableMaster.simple_test (table_master.py:692-694): dict(text, score, bbox), no 'cell' entry
    pred = post(dict(text=s, score=sc, bbox=bb, cell=cs) if cs is not None else dict(text=s, score=sc, bbox=bb))
    return {"polygons": two_point_to_four_point(pred["new_bbox"]), "structure_str_list": pred["structure_str_list"],
            "structure_str": pred["structure_str"], "html_context": pred["html_context"], "inputs": inputs,
            "text": s, "score": sc, "cell": cs}


def mtl_image_meta(crop_h: int, crop_w: int, resized_w: int, resized_h: int, size: int = 480) -> Dict:
    """the img_metas entries the convertor reads (TableResize._resize_img, model/table/lgpma/lgpma_preprocess.py:1067-1093: scale_factor
    = resized / original per axis as Python floats; TablePad._pad_img :1172-1182: pad_shape = the padded canvas)"""
    return {"scale_factor": (resized_h / crop_h, resized_w / crop_w), "pad_shape": (size, size, 3), "ori_shape": (crop_h, crop_w, 3),
            "img_shape": (resized_h, resized_w, 3)}


class MtlStage:
    """MtlTabNet over table crops of resident pages: ``pt_tsr_mtl_preprocess`` -> ``pt_tsr_mtl_backbone_net`` -> ``pt_tsr_mtl_structure``
    / ``pt_tsr_mtl_cells`` per micro-batch, then the convertor and the post-processor on the host."""

    def __init__(self, eng, convertor: Optional[MtlTabNetConvertor] = None, size: int = 480, micro_batch: int = 32, strict: bool = False):
        self.eng, self.size, self.micro_batch = eng, size, micro_batch
        self.convertor = convertor or MtlTabNetConvertor()
        self.post = MasterPostProcessor(strict=strict)
        self.stats = {"tables": 0, "tokens": 0, "cells": 0, "cell_steps": 0}

    def tables(self, page_shape: Tuple[int, int], boxes_per_page: Sequence[np.ndarray]) -> np.ndarray:
        """integer boxes [k, 4] (x1, y1, x2, y2) per page, cropped like ``crop_image_by_box`` (utils/ocr/ocr_common_utils.py:269-284)"""
        from .engine import TSR_TABLE_DTYPE
        ph, pw = page_shape
        recs = []
        for pi, boxes in enumerate(boxes_per_page):
            for b in np.asarray(boxes).reshape(-1, 4):
                x1, y1, x2, y2 = (int(v) for v in b)
                x1, y1, x2, y2 = max(x1, 0), max(y1, 0), min(x2, pw), min(y2, ph)
                if x2 <= x1 or y2 <= y1:
                    raise ValueError(f"empty table crop {b.tolist()} on a {ph}x{pw} page")
                r = np.zeros((), dtype=TSR_TABLE_DTYPE)
                r["page"], r["x0"], r["y0"], r["crop_w"], r["crop_h"] = pi, x1, y1, x2 - x1, y2 - y1
                recs.append(r)
        return np.array(recs, dtype=TSR_TABLE_DTYPE) if recs else np.zeros(0, dtype=TSR_TABLE_DTYPE)

    def decode(self, pages, tables: np.ndarray) -> List[Dict]:
        """device half + D2H for all tables: one dict of numpy arrays per table (tag ids / probabilities / boxes per decoded position,
        cell ids / probabilities per cell and step)"""
        import torch
        raw = []
        for o in range(0, len(tables), self.micro_batch):
            tb = tables[o:o + self.micro_batch]
            x = self.eng.mtl_preprocess(pages, tb, self.size)
            f3 = self.eng.mtl_backbone_features(x, (self.size, self.size))
            out = self.eng.mtl_decode(f3)
            lens, counts, steps = out["lens"], out["cell_counts"], out["cell_steps"]
            ncls = out["cfg"]["num_classes"]
            lmax = int(lens.max()) if len(lens) else 0
            tag = out["tag_logits"][:, :lmax, :ncls]
            prob = torch.softmax(tag, -1)                    # plumbing: arg-max + its probability instead of shipping the logits
            tp, ti = prob.max(-1)
            ti, tp, bx = ti.cpu().numpy(), tp.cpu().numpy(), out["boxes"][:, :lmax].cpu().numpy()
            cid, cpr = out["cell_ids"].cpu().numpy(), out["cell_prob"].cpu().numpy()
            c0 = 0
            for b in range(len(tb)):
                ln, nc, st = int(lens[b]), int(counts[b]), int(steps[b])
                raw.append(dict(tag_ids=ti[b, :ln], tag_prob=tp[b, :ln], boxes=bx[b, :ln], cell_ids=cid[c0:c0 + nc, :st] if nc else None,
                                cell_prob=cpr[c0:c0 + nc, :st] if nc else None))
                c0 += nc
                self.stats["tables"] += 1
                self.stats["tokens"] += ln
                self.stats["cells"] += nc
                self.stats["cell_steps"] += st
        return raw

    def format(self, raw: List[Dict], tables: np.ndarray, offsets: Optional[np.ndarray] = None) -> List[Dict]:
        """host half: ``decode``'s arrays of every table -> the reference's result dicts (convertor + post-processor)"""
        res = []
        for k, (r, t) in enumerate(zip(raw, tables)):
            rw, rh = self.eng.mtl_resized_size(int(t["crop_w"]), int(t["crop_h"]), self.size)
            meta = mtl_image_meta(int(t["crop_h"]), int(t["crop_w"]), rw, rh, self.size)
            d = mtl_result(self.convertor, self.post, r["tag_ids"], r["tag_prob"], r["boxes"], r["cell_ids"], r["cell_prob"], meta)
            if offsets is not None and len(d["polygons"]):
                d["polygons"] = d["polygons"] + np.tile(offsets[k].astype(d["polygons"].dtype), 4)[None]
            res.append(d)
        return res

    def run(self, pages, tables: np.ndarray, offsets: Optional[np.ndarray] = None) -> List[Dict]:
        return self.format(self.decode(pages, tables), tables, offsets)

    @staticmethod
    def _per_page(flat: List[Dict], boxes_per_page: Sequence[np.ndarray]) -> List[List[Dict]]:
        out, o = [], 0
        for b in boxes_per_page:
            k = len(np.asarray(b).reshape(-1, 4))
            out.append(flat[o:o + k])
            o += k
        return out

    def __call__(self, pages, boxes_per_page: Sequence[np.ndarray], page_frame: bool = False) -> List[List[Dict]]:
        tables = self.tables(tuple(pages.shape[1:3]), boxes_per_page)
        offs = np.stack([tables["x0"], tables["y0"]], 1) if page_frame and len(tables) else None
        flat = self.run(pages, tables, offs) if len(tables) else []
        return self._per_page(flat, boxes_per_page)

    def stream(self, batches, page_frame: bool = False):
        """``(pages, boxes_per_page)`` batches -> ``__call__``'s result per batch, in order.  The host half of batch k (pure Python: ~0.25 ms per table)
        runs on a worker thread while this thread decodes batch k + 1 on the device -- the library calls release the interpreter lock, so the two overlap
        the way ``OcrTablePipeline.predict_stream`` overlaps its host halves."""
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=1) as pool:
            pending = None
            for pages, boxes_per_page in batches:
                tables = self.tables(tuple(pages.shape[1:3]), boxes_per_page)
                offs = np.stack([tables["x0"], tables["y0"]], 1) if page_frame and len(tables) else None
                raw = self.decode(pages, tables) if len(tables) else []
                fut = pool.submit(lambda r=raw, t=tables, o=offs, b=boxes_per_page: self._per_page(self.format(r, t, o), b))
                if pending is not None:
                    yield pending.result()
                pending = fut
            if pending is not None:
                yield pending.result()
