"""ORACLE (test infrastructure only -- see oracle/__init__.py): CPU restatement of the PicoDet layout stage.

Follows /root/reference/src/pdftable/model/picodet/:
  ``lcnet_forward``     LCNet.forward lcnet.py:241-257 (ConvBNLayer :65-90, DepthwiseSeparable :93-123, SEModule :126-153)
  ``csppan_forward``    CSPPAN.forward csp_pan.py:305-345 (Channel_T :212-227, CSPLayer :160-209, DarknetBottleneck
                        :108-157, DPModule :56-105)
  ``picohead_forward``  PicoHead.forward_eval pico_head.py:1108-1160 with export_post_process=False (what the ONNX export
                        the reference runs delivers, ocr_layout_task.py:159-175) over PicoFeat.forward :154-167
  ``picodet_preprocess`` OCRPicodetPreProcessor.__call__ processor_picodet.py:72-113
  ``picodet_postprocess`` OCRPicodetPostProcessor.__call__ :184-298, ``hard_nms`` :301-330, ``iou_of`` :333-348

PINNED by tests/golden/picodet.npz (the three reference modules on seeded weights; the reference post-processor on
seeded head outputs).  The network hyper-parameters are an ASSUMPTION (see pdf_table_amd.synth_weights.picodet_state_dict);
cv2.resize inside the pre-processor is the oracle's restatement (db_pre.cv2_resize_linear_u8, parity unpinned).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from pdf_table_amd.synth_weights import LCNET_CONFIG, PICODET_STANDIN

NORM_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)       # configuration_picodet.py:48-49
NORM_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
LABELS = {"en": ["text", "title", "list", "table", "figure"],       # configuration_picodet.py:79-104
          "ch": ["text", "title", "figure", "figure_caption", "table", "table_caption", "header", "footer", "reference",
                 "equation"],
          "table": ["table"]}


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _conv_bn(sd, p, x, stride=1, groups=1, act=True, norm="bn"):
    w = sd[p + ".conv.weight"]
    y = _bn(sd, f"{p}.{norm}", F.conv2d(x, w, None, stride, (w.shape[2] - 1) // 2, 1, groups))
    return F.hardswish(y) if act else y


def lcnet_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, prefix: str = "backbone") -> List[torch.Tensor]:
    x = _conv_bn(sd, prefix + ".conv1", x, stride=2)
    outs = []
    for blk in ("blocks2", "blocks3", "blocks4", "blocks5", "blocks6"):
        for i, (k, cin, cout, s, se) in enumerate(LCNET_CONFIG[blk]):
            p = f"{prefix}.{blk}.{i}"
            x = _conv_bn(sd, p + ".dw_conv", x, stride=s, groups=cin)
            if se:
                a = F.adaptive_avg_pool2d(x, 1)
                a = F.relu(F.conv2d(a, sd[p + ".se.conv1.weight"], sd[p + ".se.conv1.bias"]))
                a = F.hardsigmoid(F.conv2d(a, sd[p + ".se.conv2.weight"], sd[p + ".se.conv2.bias"]))
                x = x * a
            x = _conv_bn(sd, p + ".pw_conv", x)
        if blk != "blocks2":
            outs.append(x)
    return outs[1:]            # feature_maps [3, 4, 5]: blocks4, blocks5, blocks6 outputs


def _dp(sd, p, x, stride=1):
    w = sd[p + ".dwconv.weight"]
    x = F.hardswish(_bn(sd, p + ".bn1", F.conv2d(x, w, None, stride, (w.shape[2] - 1) // 2, 1, w.shape[0])))
    return F.hardswish(_bn(sd, p + ".bn2", F.conv2d(x, sd[p + ".pwconv.weight"])))


def _csp(sd, p, x):
    short = _conv_bn(sd, p + ".short_conv", x)
    main = _conv_bn(sd, p + ".main_conv", x)
    main = _dp(sd, p + ".blocks.0.conv2", _conv_bn(sd, p + ".blocks.0.conv1", main))     # add_identity=False
    return _conv_bn(sd, p + ".final_conv", torch.cat((main, short), 1))


def csppan_forward(sd, feats: List[torch.Tensor], prefix: str = "neck") -> List[torch.Tensor]:
    ins = [_conv_bn(sd, f"{prefix}.conv_t.convs.{i}", f) for i, f in enumerate(feats)]
    inner = [ins[-1]]
    for idx in range(len(ins) - 1, 0, -1):
        up = F.interpolate(inner[0], size=ins[idx - 1].shape[2:4], mode="nearest")
        inner.insert(0, _csp(sd, f"{prefix}.top_down_blocks.{len(ins) - 1 - idx}", torch.cat([up, ins[idx - 1]], 1)))
    outs = [inner[0]]
    for idx in range(len(ins) - 1):
        down = _dp(sd, f"{prefix}.downsamples.{idx}", outs[-1], stride=2)
        outs.append(_csp(sd, f"{prefix}.bottom_up_blocks.{idx}", torch.cat([down, inner[idx + 1]], 1)))
    top = _dp(sd, prefix + ".first_top_conv", ins[-1], stride=2) + _dp(sd, prefix + ".second_top_conv", outs[-1], stride=2)
    outs.append(top)
    return outs


def picohead_forward(sd, feats: List[torch.Tensor], num_classes: int, prefix: str = "head"):
    """-> (scores: per level [B, A, ncls] after sigmoid, box logits: per level [B, A, 4 * (reg_max + 1)])"""
    scores, boxes = [], []
    for s, f in enumerate(feats):
        x = f
        for i in range(PICODET_STANDIN["num_convs"]):
            x = F.hardswish(_conv_bn(sd, f"{prefix}.conv_feat.cls_conv_dw{s}_{i}", x, groups=x.shape[1], act=False, norm="norm"))
            x = F.hardswish(_conv_bn(sd, f"{prefix}.conv_feat.cls_conv_pw{s}_{i}", x, act=False, norm="norm"))
        y = F.conv2d(x, sd[f"{prefix}.head_cls{s}.weight"], sd[f"{prefix}.head_cls{s}.bias"])
        b = y.shape[0]
        scores.append(torch.sigmoid(y[:, :num_classes]).reshape(b, num_classes, -1).permute(0, 2, 1))
        boxes.append(y[:, num_classes:].reshape(b, y.shape[1] - num_classes, -1).permute(0, 2, 1))
    return scores, boxes


def picodet_forward(sd, x: torch.Tensor, num_classes: int = 5):
    return picohead_forward(sd, csppan_forward(sd, lcnet_forward(sd, x)), num_classes)


def picodet_preprocess(img_rgb: np.ndarray, img_h: int = 800, img_w: int = 608):
    """RGB uint8 HxWx3 -> (f32 [3, img_h, img_w], scale_factor [ratio_h, ratio_w])"""
    from .db_pre import cv2_resize_linear_u8
    img = img_rgb[:, :, ::-1]
    h, w = img.shape[:2]
    resized = cv2_resize_linear_u8(np.ascontiguousarray(img), img_w, img_h)
    x = (resized.astype("float32") * np.float32(1.0 / 255.0) - NORM_MEAN.reshape(1, 1, 3)) / NORM_STD.reshape(1, 1, 3)
    return np.ascontiguousarray(x.transpose(2, 0, 1)), [float(img_h) / h, float(img_w) / w]


def iou_of(boxes0, boxes1, eps=1e-5):
    lt = np.maximum(boxes0[..., :2], boxes1[..., :2])
    rb = np.minimum(boxes0[..., 2:], boxes1[..., 2:])
    hw = np.clip(rb - lt, 0.0, None)
    inter = hw[..., 0] * hw[..., 1]

    def area(b):
        d = np.clip(b[..., 2:] - b[..., :2], 0.0, None)
        return d[..., 0] * d[..., 1]
    return inter / (area(boxes0) + area(boxes1) - inter + eps)


def hard_nms(box_scores, iou_threshold, top_k=-1, candidate_size=200):
    scores, boxes = box_scores[:, -1], box_scores[:, :-1]
    picked = []
    indexes = np.argsort(scores)[-candidate_size:]
    while len(indexes) > 0:
        cur = indexes[-1]
        picked.append(cur)
        if 0 < top_k == len(picked) or len(indexes) == 1:
            break
        indexes = indexes[:-1]
        iou = iou_of(boxes[indexes, :], np.expand_dims(boxes[cur, :], 0))
        indexes = indexes[iou <= iou_threshold]
    return box_scores[picked, :]


def picodet_postprocess(scores: List[np.ndarray], raw_boxes: List[np.ndarray], org_shape, scale_factor, target_shape,
                        labels: List[str], strides=(8, 16, 32, 64), score_threshold=0.5, nms_threshold=0.5,
                        nms_top_k=1000, keep_top_k=100):
    """scores / raw_boxes: per level [1, A, ncls] / [1, A, 4 * (reg_max + 1)] -> list of {'bbox' f32[4], 'label', 'score',
    'category_id'} (the `bboxs` list of the reference's result dict)."""
    from scipy.special import softmax
    reg_max = raw_boxes[0].shape[-1] // 4 - 1
    decode_boxes, select_scores = [], []
    for stride, bd, sc in zip(strides, raw_boxes, scores):
        bd, sc = bd[0], sc[0]
        fm_h, fm_w = target_shape[0] / stride, target_shape[1] / stride
        ww, hh = np.meshgrid(np.arange(fm_w), np.arange(fm_h))
        ct_row, ct_col = (hh.flatten() + 0.5) * stride, (ww.flatten() + 0.5) * stride
        center = np.stack((ct_col, ct_row, ct_col, ct_row), axis=1)
        dist = softmax(bd.reshape((-1, reg_max + 1)), axis=1) * np.expand_dims(np.arange(reg_max + 1), 0)
        dist = np.sum(dist, axis=1).reshape((-1, 4)) * stride
        topk = np.argsort(sc.max(axis=1))[::-1][:nms_top_k]
        decode_boxes.append(center[topk] + [-1, -1, 1, 1] * dist[topk])
        select_scores.append(sc[topk])
    bboxes = np.concatenate(decode_boxes, 0)
    conf = np.concatenate(select_scores, 0)
    picked, picked_labels = [], []
    for c in range(conf.shape[1]):
        probs = conf[:, c]
        mask = probs > score_threshold
        if not mask.any():
            continue
        bp = hard_nms(np.concatenate([bboxes[mask], probs[mask].reshape(-1, 1)], 1), nms_threshold, keep_top_k)
        picked.append(bp)
        picked_labels.extend([c] * bp.shape[0])
    if not picked:
        return []
    pb = np.concatenate(picked)
    ori = np.array(org_shape, dtype=np.float32)
    # warp_boxes (:140-162): min/max over the 4 corners (a no-op re-ordering for axis-aligned boxes), clip, float32
    b = pb[:, :4]
    x = b[:, [0, 2, 0, 2]]
    y = b[:, [1, 3, 3, 1]]
    xy = np.stack([x.min(1), y.min(1), x.max(1), y.max(1)], 1)
    xy[:, [0, 2]] = xy[:, [0, 2]].clip(0, ori[1])
    xy[:, [1, 3]] = xy[:, [1, 3]].clip(0, ori[0])
    pb[:, :4] = xy.astype(np.float32)
    sf = np.array(scale_factor, dtype=np.float32)
    pb[:, :4] /= np.concatenate([sf[::-1], sf[::-1]])
    return [{"bbox": pb[i, :4].copy(), "label": labels[int(c)], "score": pb[i, 4], "category_id": int(c)}
            for i, c in enumerate(picked_labels)]
