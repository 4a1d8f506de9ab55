"""Seeded MtlTabNet decoder outputs shared by tests/golden/make_golden.py (inputs of the reference's own MtlTabNetConvertor /
MasterPostProcessor run) and tests/test_mtl_host.py: only case names, the seed and the EXPECTED results are stored in
tests/golden/mtl_tabnet_host.json."""
import re

import numpy as np

MTL_HOST_CASES = {
    # structure token strings (',' joined ids come from the alphabet), cell texts, (ori_h, ori_w)
    "plain": ("<thead>,<tr>,<td></td>,<td></td>,<td></td>,</tr>,</thead>,<tbody>,<tr>,<td></td>,<eb></eb>,<td></td>,</tr>,<tr>,<td></td>,"
              "<td></td>,<eb1></eb1>,</tr>,</tbody>", ["Variable", "n (%)", "<b>HR</b>", "Age", "0.5", "x<sup>2</sup>", "12"], (211, 640)),
    "spans": ("<thead>,<tr>,<td,colspan=\"2\",>,</td>,<td></td>,</tr>,<tr>,<td,rowspan=\"2\",colspan=\"3\",>,</td>,<td></td>,</tr>,</thead>,<tbody>,"
              "<tr>,<td></td>,<td,rowspan=\"3\",>,</td>,<eb2></eb2>,</tr>,</tbody>", ["A b", "<b>c</b>", "<b>d</b><b>e</b>", "f", "1", "2"], (300, 333)),
    "no_tbody_end": ("<thead>,<tr>,<td></td>,</tr>,</thead>,<tbody>,<tr>,<td></td>,<td></td>", ["h", "a", "b"], (97, 480)),
    "ends_in_tr": ("<tbody>,<tr>,<td></td>,<eb5></eb5>,<eb10></eb10>,</tr>", ["only"], (480, 200)),
    "one_cell": ("<tbody>,<tr>,<td></td>,</tr>,</tbody>", ["lonely"], (50, 60)),          # out_cell.size(0) == 1 -> no cell strings (quirk)
    "no_cells": ("<tbody>,<tr>,<eb></eb>,<eb3></eb3>,</tr>,</tbody>", [], (64, 64)),
    "more_cells_than_text": ("<thead>,<tr>,<td></td>,<td></td>,</tr>,</thead>,<tbody>,<tr>,<td></td>,<td></td>,<td></td>,</tr>,</tbody>",
                             ["p", "q", "r", "s", "t"], (128, 512)),
    "isolate_span": ("<thead>,<tr>,<td></td>,rowspan=\"2\",>,</td>,<td></td>,</tr>,</thead>,<tbody>,<tr>,<td></td>,</tr>,</tbody>", ["u", "v", "w"], (90, 700)),
    "pad_and_unknown": ("<tbody>,<tr>,<td></td>,<PAD>,<UKN>,<td></td>,</tr>,</tbody>", ["z z", "<i>y</i>"], (200, 200)),
}


def mtl_host_case_tensors(name, convertor_like, seed):
    """seeded decoder outputs for one case: the ids of the case's tokens win every position by a margin, everything else is noise"""
    tokens, cells, (oh, ow) = MTL_HOST_CASES[name]
    rng = np.random.default_rng(seed)
    c2i, c2i_cell = convertor_like["char2idx"], convertor_like["char2idx_cell"]
    ids = [c2i[t] for t in tokens.split(",")] + [convertor_like["end_idx"]]
    ids += [int(rng.integers(0, len(c2i))) for _ in range(3)]            # positions after <EOS> exist in the tensors and are ignored
    T = len(ids)
    tag = (rng.standard_normal((1, T, len(c2i))) * 0.5).astype(np.float32)
    tag[0, np.arange(T), ids] += rng.uniform(4.0, 9.0, T).astype(np.float32)
    box = rng.uniform(0.02, 0.6, (1, T, 4)).astype(np.float32)
    box[0, 4::7, :] *= 0.001                                                # some boxes whose coordinates sum to <= 1 pixel: dropped
    cell_tokens = []
    for text in cells:
        toks = [t for t in re.findall(r"</?[a-z]+>|.", text)]
        cell_tokens.append([c2i_cell[t] for t in toks])
    n_cells = len(cells)
    if n_cells == 0:
        cell = np.zeros(1, np.float32)
    else:
        steps = max(len(t) for t in cell_tokens) + 1
        cid = np.full((n_cells, steps), convertor_like["end_idx_cell"], np.int64)
        for i, t in enumerate(cell_tokens):
            cid[i, :len(t)] = t
        if n_cells > 2:
            cid[1, 0] = convertor_like["padding_idx_cell"]                # an emitted <PAD> is skipped by tensor2idx_cell
        cell = (rng.standard_normal((n_cells, steps, len(c2i_cell))) * 0.5).astype(np.float32)
        ii, jj = np.meshgrid(np.arange(n_cells), np.arange(steps), indexing="ij")
        cell[ii, jj, cid] += rng.uniform(4.0, 9.0, (n_cells, steps)).astype(np.float32)
    size = 480
    if ow < oh:
        rw, rh = int(size / oh * ow), size
    else:
        rw, rh = size, int(size / ow * oh)
    meta = {"scale_factor": (rh / oh, rw / ow), "pad_shape": (size, size, 3), "ori_shape": (oh, ow, 3), "img_shape": (rh, rw, 3)}
    return tag, box, cell, meta
