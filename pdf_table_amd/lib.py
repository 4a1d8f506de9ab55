"""ctypes binding of libpdftable_hip.so (C ABI: include/pdftable_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, an exception is
raised.  ``import torch`` happens first so that the HIP runtime PyTorch ships (same soname,
``libamdhip64.so.7``) is the one both share -- device pointers and streams then interoperate.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_LIB_PATH") or os.path.join(_HERE, "libpdftable_hip.so")   # PT_LIB_PATH: kernel A/B experiments

PT_MODEL_DB_RESNET18 = 1
PT_MODEL_CRNN = 2
PT_MODEL_LORE_DLA34 = 3
PT_MODEL_LORE_PROCESSOR = 4
PT_MODEL_PICODET = 5
PT_MODEL_LORE_RESNET18 = 6
PT_MODEL_DB_NAS = 7
PT_MODEL_PPLCNET = 8      # + slot (0 .. PT_CLS_SLOTS - 1)
PT_MODEL_CONVNEXT_VIT = 16
PT_MODEL_MTL_BACKBONE = 17
PT_MODEL_MTL_DECODER = 18
PT_CVIT_W, PT_CVIT_CHUNK_W, PT_CVIT_CHUNK_STEP, PT_CVIT_T, PT_CVIT_NCLS = 804, 300, 252, 201, 7644
PT_CLS_SLOTS = 4
PT_CLS_MAX_CLASSES = 16
PT_LAYOUT_HEAD_CS, PT_LAYOUT_CAND_FLOATS = 40, 48
PT_DET_PRE_DB_PP = 0
PT_DET_PRE_DB_TORCH = 1
PT_DET_PRE_NONE = 2
PT_PRECISION_BF16 = 0
PT_PRECISION_BF16X3 = 1
PT_PRECISION_F16X2 = 2
PT_PRECISION_F16 = 3          # single-pass IEEE half (ABI 14): fp16 activations and weight tiles, saturating stores
PT_DET_POST_DB_PP = 0
PT_DET_POST_DB_TORCH = 1
PT_TSR_MAX_CELLS = 3000
PT_REC_H, PT_REC_W, PT_REC_T, PT_REC_NCLS = 32, 640, 160, 7644
PT_PROF_CLASSES = ("conv3x3", "conv1x1", "stem", "other")
EXPECTED_ABI = 14         # pt_abi_version() of the library these prototypes were written against (include/pdftable_hip.h)

_lib = None


class PtError(RuntimeError):
    pass


def _proto(lib):
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    ip = C.POINTER(C.c_int)
    P = {
        "pt_engine_create": (i, [i, C.POINTER(vp)]),
        "pt_engine_destroy": (None, [vp]),
        "pt_last_error": (C.c_char_p, []),
        "pt_abi_version": (i, []),
        "pt_engine_check": (i, [vp]),
        "pt_engine_set_precision": (i, [vp, i]),
        "pt_engine_set_lstm_cluster": (i, [vp, i]),
        "pt_engine_set_mtl_kv_fp8": (i, [vp, i]),
        "pt_engine_set_dcn_mfma": (i, [vp, i]),
        "pt_weights_load": (i, [vp, i, vp, sz]),
        "pt_weights_load_device": (i, [vp, i, vp, sz, vp]),
        "pt_det_plan": (i, [i, i, i, ip, ip]),
        "pt_det_forward": (i, [vp, vp, i, i, i, i, f, i, vp, vp, vp]),
        "pt_det_forward_net": (i, [vp, vp, i, i, i, vp, vp, vp]),
        "pt_det_preprocess": (i, [vp, vp, i, i, i, i, vp, vp]),
        "pt_det_bitmap": (i, [vp, vp, i, i, i, f, i, vp, vp]),
        "pt_det_box_scores": (i, [vp, vp, i, i, i, vp, i, vp, vp]),
        "pt_db_candidates": (i, [vp, i, i, i, f, vp, vp, i, ip]),
        "pt_db_finalize": (i, [vp, vp, i, f, f, f, i, i, i, i, i, vp, vp, i, ip]),
        "pt_db_candidates_batch": (i, [vp, i, i, i, i, f, i, vp, vp, i, vp]),
        "pt_db_finalize_batch": (i, [vp, vp, vp, i, i, f, f, f, i, i, i, i, i, i, i, vp, vp, vp, vp]),
        "pt_rec_forward": (i, [vp, vp, i, i, i, vp, vp, i, vp, vp, vp]),
        "pt_rec_forward_crops": (i, [vp, vp, vp, vp, i, vp, vp, vp]),
        "pt_rec_forward_net": (i, [vp, vp, i, vp, vp, vp]),
        "pt_rec_preprocess": (i, [vp, vp, i, i, i, vp, vp, i, vp, vp]),
        "pt_rec_cvit_forward": (i, [vp, vp, i, i, i, vp, vp, vp, i, vp, vp, vp]),
        "pt_rec_cvit_forward_crops": (i, [vp, vp, vp, vp, vp, i, vp, vp, vp]),
        "pt_rec_cvit_forward_net": (i, [vp, vp, i, i, vp, vp, vp, vp]),
        "pt_rec_cvit_preprocess_crops": (i, [vp, vp, vp, vp, i, vp, vp]),
        "pt_tsr_mtl_backbone_net": (i, [vp, vp, i, i, i, vp, vp]),
        "pt_tsr_mtl_preprocess": (i, [vp, vp, i, i, i, vp, i, i, vp, vp]),
        "pt_tsr_mtl_resized_size": (None, [i, i, i, ip, ip]),
        "pt_tsr_mtl_decoder_config": (i, [vp, ip]),
        "pt_tsr_mtl_structure": (i, [vp, vp, i, i, vp, vp, ip, ip, i, vp]),
        "pt_tsr_mtl_cells": (i, [vp, i, vp, vp, vp, ip, i, vp]),
        "pt_rec_pp_preprocess": (i, [vp, vp, i, i, i, vp, vp, i, vp, i, i, i, vp, vp]),
        "pt_rec_pp_preprocess_crops": (i, [vp, vp, vp, vp, i, vp, i, i, i, vp, vp]),
        "pt_layout_plan": (i, [i, i, ip, ip]),
        "pt_layout_preprocess": (i, [vp, vp, i, i, i, i, i, vp, vp]),
        "pt_layout_forward_net": (i, [vp, vp, i, i, i, vp, vp, vp, vp, vp]),
        "pt_layout_candidates": (i, [vp, vp, vp, vp, vp, i, i, i, i, f, i, vp, vp, vp]),
        "pt_layout_forward": (i, [vp, vp, i, i, i, i, i, i, f, i, vp, vp, vp]),
        "pt_tsr_preprocess": (i, [vp, vp, i, i, i, vp, i, i, i, i, vp, vp]),
        "pt_tsr_forward_net": (i, [vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp]),
        "pt_tsr_forward_net_wireless": (i, [vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp]),
        "pt_tsr_decode": (i, [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, f, vp, vp, vp, vp]),
        "pt_tsr_forward_decode": (i, [vp, vp, i, i, i, i, f, vp, vp, vp, vp]),
        "pt_tsr_process": (i, [vp, vp, vp, vp, i, i, vp, vp, vp]),
        "pt_hard_nms": (i, [vp, vp, vp, i, C.c_double, i, vp, vp]),
        "pt_cls_preprocess": (i, [vp, vp, vp, i, i, i, i, i, vp, vp]),
        "pt_cls_forward_net": (i, [vp, i, vp, i, i, i, i, vp, ip, vp]),
        "pt_cls_forward": (i, [vp, i, vp, vp, i, i, i, i, i, i, vp, ip, vp]),
        "pt_cls_forward_lines": (i, [vp, i, vp, i, i, i, vp, vp, i, i, i, i, i, i, vp, ip, vp]),
        "pt_op_conv2d": (i, [vp, vp, i, i, i, i, vp, vp, i, i, i, vp, i, i, i, i, vp, i, i, i, i, vp]),
        "pt_op_dcn": (i, [vp, vp, vp, i, i, i, i, vp, vp, i, vp, i, i, vp]),
        "pt_op_stem7x7": (i, [vp, vp, i, i, i, vp, vp, vp, i, vp]),
        "pt_op_maxpool3x3s2": (i, [vp, vp, i, i, i, i, vp, i, vp]),
        "pt_op_db_head_final": (i, [vp, vp, i, i, i, vp, vp, vp, vp, i, vp]),
        "pt_op_dwconv": (i, [vp, vp, i, i, i, i, vp, vp, i, i, i, vp, i, vp]),
        "pt_op_add": (i, [vp, vp, vp, vp, C.c_longlong, i, i, vp]),
        "pt_op_maxpool": (i, [vp, vp, i, i, i, i, i, i, i, vp, i, vp]),
        "pt_op_avgpool": (i, [vp, vp, i, i, i, i, i, vp, i, vp]),
        "pt_op_chan_mean": (i, [vp, vp, i, i, i, vp, vp, i, vp]),
        "pt_op_chan_mean_scratch_floats": (i, [i, i]),
        "pt_op_scale_channels": (i, [vp, vp, vp, i, i, i, vp, i, vp]),
        "pt_op_act": (i, [vp, vp, C.c_longlong, i, C.c_float, C.c_float, vp, i, i, vp]),
        "pt_op_copy_channels": (i, [vp, vp, C.c_longlong, i, i, vp, i, i, i, vp]),
        "pt_op_upsample_nearest": (i, [vp, vp, i, i, i, i, i, vp, vp]),
        "pt_op_mul": (i, [vp, vp, vp, vp, C.c_longlong, i, i, vp]),
        "pt_copy_bytes": (i, [vp, vp, vp, C.c_longlong, vp]),
        "pt_op_layernorm": (i, [vp, vp, C.c_longlong, i, i, vp, vp, C.c_float, vp, i, vp]),
        "pt_op_softmax": (i, [vp, vp, C.c_longlong, i, i, vp, vp, i, vp]),
        "pt_op_attention": (i, [vp, vp, i, i, i, i, i, C.c_float, vp, i, i, vp]),
        "pt_profile_enable": (i, [vp, i]),
        "pt_profile_read": (i, [vp, vp, vp, vp]),
        "pt_profile_read_labels": (i, [vp, C.c_char_p, i]),
    }
    for name, (res, args) in P.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    return P


EXPORTS = None


def load():
    """Load the engine library (once).  Raises if it has not been built (python -m pdf_table_amd.build)."""
    global _lib, EXPORTS
    if _lib is None:
        import torch  # noqa: F401  (must precede the CDLL: see module docstring)
        if not os.path.exists(LIB_PATH):
            raise PtError(f"{LIB_PATH} not found: run `python -m pdf_table_amd.build` (hipcc, gfx950). "
                          "There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        EXPORTS = _proto(lib)
        # ABI 12 / 13 inserted arguments into existing pt_op_* signatures: a stale library resolves every name and would take the stream pointer
        # for `split` -- refuse it by version instead of faulting later
        have = int(lib.pt_abi_version())
        if have != EXPECTED_ABI:
            raise PtError(f"{LIB_PATH} reports ABI {have}, this binding was written for ABI {EXPECTED_ABI}: rebuild it "
                          "(`python -m pdf_table_amd.build --force`)")
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().pt_last_error()
        raise PtError(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}")
