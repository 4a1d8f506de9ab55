"""Oracle for the PP-LCNet image classifiers (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates, in fp32 functional torch over a ``state_dict``:
  model/cls/cls_pp_lcnet.py:164-283       PPLCNet.__init__/forward: LCNet x1.0 (conv1 + blocks2..6) -> AdaptiveAvgPool2d(1)
                                          -> last_conv 1x1 (512 -> 1280, no bias) -> Hardswish -> [Dropout, eval: identity]
                                          -> Flatten -> Linear(1280, class_num)
  model/cls/cls_pp_lcnet.py:54-66,190-191 NET_CONFIG and the stride_list override of each stage's first block
  model/cls/configuration_cls_pulc.py:20-39 per task: class_num and stride_list ([2,[2,1],[2,1],[2,1],[2,1]] for the
                                          text-line tasks)
  model/cls/image_processing_pplcnet.py:155-192 Topk post-processing;  :109-152 TableAttribute
Pinned by tests/golden/pplcnet.npz (outputs of the reference's own PPLCNet module and Topk, tests/test_oracle_pplcnet.py)."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from oracle.picodet import _conv_bn
from pdf_table_amd.synth_weights import LCNET_CONFIG

CLS_TASKS = {      # configuration_cls_pulc.py:20-39 + image_processing_pplcnet.py:74-107
    "table_attribute": {"class_num": 6, "textline": False, "size": (224, 224), "topk": None},
    "text_image_orientation": {"class_num": 4, "textline": False, "size": (224, 224), "topk": 2},
    "textline_orientation": {"class_num": 2, "textline": True, "size": (80, 160), "topk": 1},
    "language_classification": {"class_num": 10, "textline": True, "size": (80, 160), "topk": 2},
}
CLASS_ID_MAP = {   # image_processing_pplcnet.py:40-72
    "text_image_orientation": {0: "0", 1: "90", 2: "180", 3: "270"},
    "textline_orientation": {0: "0_degree", 1: "180_degree"},
    "language_classification": {0: "arabic", 1: "chinese_cht", 2: "cyrillic", 3: "devanagari", 4: "japan", 5: "ka",
                                6: "korean", 7: "ta", 8: "te", 9: "latin"},
}


def pplcnet_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, textline: bool = False) -> torch.Tensor:
    """x f32 [n,3,H,W] -> logits f32 [n, class_num]"""
    with torch.no_grad():
        x = _conv_bn(sd, "conv1", x, stride=2)
        for blk in ("blocks2", "blocks3", "blocks4", "blocks5", "blocks6"):
            for i, (k, cin, cout, s, se) in enumerate(LCNET_CONFIG[blk]):
                p = f"{blk}.{i}"
                stride = (2, 1) if (textline and s == 2) else s      # cls_pp_lcnet.py:190-191
                x = _conv_bn(sd, p + ".dw_conv", x, stride=stride, groups=cin)
                if se:
                    a = F.adaptive_avg_pool2d(x, 1)
                    a = F.relu(F.conv2d(a, sd[p + ".se.conv1.weight"], sd[p + ".se.conv1.bias"]))
                    a = F.hardsigmoid(F.conv2d(a, sd[p + ".se.conv2.weight"], sd[p + ".se.conv2.bias"]))
                    x = x * a
                x = _conv_bn(sd, p + ".pw_conv", x)
        x = F.adaptive_avg_pool2d(x, 1)
        x = F.hardswish(F.conv2d(x, sd["last_conv.weight"]))
        return F.linear(x.flatten(1), sd["fc.weight"], sd["fc.bias"])


def topk_postprocess(logits: np.ndarray, task: str) -> List[Dict]:
    """Topk.__call__ (image_processing_pplcnet.py:162-192): soft-max, top-k ids (descending), scores rounded to 5 places"""
    topk = CLS_TASKS[task]["topk"]
    x = torch.softmax(torch.from_numpy(np.asarray(logits, np.float32)), dim=-1).numpy()
    out = []
    for probs in x:
        index = probs.argsort(axis=0)[-topk:][::-1].astype("int32")
        out.append({"class_ids": [int(i) for i in index],
                    "scores": np.around([probs[i].item() for i in index], decimals=5).tolist(),
                    "label_names": [CLASS_ID_MAP[task][int(i)] for i in index]})
    return out


def table_attribute_postprocess(outputs: np.ndarray, thresholds: Sequence[float] = (0.5,) * 6) -> List[Dict]:
    """TableAttribute.__call__ (image_processing_pplcnet.py:125-152), including its quirk: `obstruction` and `angle` are
    compared with number_threshold, the `output` vector with each attribute's own threshold"""
    names = (("Scanned", "Photo"), ("Little", "Numerous"), ("Black-and-White", "Multicolor"), ("Clear", "Blurry"),
             ("Without-Obstacles", "With-Obstacles"), ("Horizontal", "Tilted"))
    use = (thresholds[0], thresholds[1], thresholds[2], thresholds[3], thresholds[1], thresholds[1])
    out = []
    for res in np.asarray(outputs).tolist():
        out.append({"attributes": [names[i][0] if res[i] > use[i] else names[i][1] for i in range(6)],
                    "output": (np.array(res) > np.array(thresholds)).astype(np.int8).tolist()})
    return out
