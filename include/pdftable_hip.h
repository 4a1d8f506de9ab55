/*
 * pdftable_hip.h -- C ABI of libpdftable_hip.so, the MI355X (gfx950) engine behind pdf_table's
 * four-stage page-vision path.
 *
 * The reference has NO FFI for this path: each stage is a Python ``BaseInferTask`` whose
 * ``_run_model`` calls ``self.infer(...)`` which dispatches to PyTorch-eager or onnxruntime
 * (reference: src/pdftable/model/ocr_pdf/base_infer_task.py:366-381).  The functions below are
 * what a ``predictor_type == "hip"`` branch of that ``infer()`` binds (INTEGRATION.md shows the
 * ctypes stub).  Conventions:
 *   - plain pointers and sizes only; no torch / numpy types cross this boundary;
 *   - every function returns an int status, 0 == PT_OK; pt_last_error() gives the message of the
 *     last failure on the calling thread;
 *   - pointers named d_* are DEVICE pointers on the engine's GPU, h_* are HOST pointers;
 *   - weights, activations and scratch are engine-owned; inputs and outputs are caller-owned;
 *   - one engine per GPU; calls on one engine must be issued by one host thread at a time (the
 *     reference is single-threaded, base_infer_task.py:311-315).  Every stage (layout + cls, det,
 *     rec, the tsr detector pt_tsr_forward* / pt_tsr_decode, the tsr processor pt_tsr_process) has
 *     its own activation arena and scratch inside the engine, so calls of DIFFERENT
 *     stages may be in flight on different streams at the same time (measured: +10 % pages/s with
 *     the recogniser on a second stream); two calls of the SAME stage -- and pt_cls_forward_lines
 *     with pt_rec_forward*, which share the crop buffers -- must be stream-ordered.
 *     Distinct engines own all their state (weights,
 *     arena, scratch, the decode state between the steps of pt_tsr_forward_decode) and are
 *     independent, with ONE restriction per device: the bf16 recognition LSTM (pt_rec_forward*)
 *     is a cluster kernel whose workgroups must all be resident at once, so two such launches may
 *     not overlap in time on one GPU (two engines, streams or processes).  An overlap cannot hang:
 *     a workgroup gives up after a bounded wait, pt_engine_check() (or the next rec call on that
 *     engine) returns PT_ERR_HIP ("not co-resident") once and switches the engine to the streaming
 *     kernel; pt_engine_set_lstm_cluster(e, 0) selects it up front;
 *   - `stream` is a hipStream_t (0 = the null stream).  Calls are asynchronous on that stream
 *     unless stated otherwise.
 */
#ifndef PDFTABLE_HIP_H
#define PDFTABLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_ERR_INVALID 1   /* bad argument / shape */
#define PT_ERR_HIP 2       /* a HIP runtime call failed */
#define PT_ERR_STATE 3     /* e.g. weights for that model not loaded */
#define PT_ERR_FORMAT 4    /* malformed weight blob */

typedef struct pt_engine pt_engine;
typedef void* pt_stream; /* hipStream_t */

/* ---- lifetime (replaces DeployUtils.model_eval / prepare_onnx_model, utils/deploy_utils.py:227-280) ---- */
int pt_engine_create(int device_id, pt_engine** out);
void pt_engine_destroy(pt_engine* e);
const char* pt_last_error(void);
int pt_abi_version(void);
/* Failures the DEVICE detected in work already executed (today: the co-residency time-out of the recognition LSTM, see
 * above).  Call it after synchronising the stream whose results are about to be consumed; PT_OK if nothing was flagged.
 * No counterpart in the reference (a torch/onnxruntime call either returns or raises). */
int pt_engine_check(pt_engine* e);
/* on = 1 (default; PT_LSTM_CLUSTER=0 in the environment changes the default): the bf16 recognition LSTM runs as the
 * weight-stationary cluster kernel, whose workgroups must all be co-resident -- only safe when nothing else runs on the
 * GPU beside the recogniser.  on = 0: the streaming kernel (about twice the LSTM time, no co-residency requirement):
 * use it whenever pt_rec_forward* runs on a stream beside other work.  A failure pt_engine_check reports also
 * switches the engine to 0 and clears the flag, so the failed batch can simply be submitted again. */
int pt_engine_set_lstm_cluster(pt_engine* e, int on);
/* MtlTabNet (pt_tsr_mtl_structure), PT_PRECISION_BF16 only: on = 1 keeps a second copy of the source-attention keys / values of the
 * KV-cached structure loop as fp8 (e4m3, x 8) and streams THAT every step -- half the bytes of the loop's dominant HBM stream
 * (BASELINE.json configs[4] names fp8; this is where fp8 pays on this path).  A throughput option with recorded drift
 * (tests/test_gpu_mtl.py), off by default; ignored in PT_PRECISION_BF16X3.  Environment default: PT_MTL_KV_FP8. */
int pt_engine_set_mtl_kv_fp8(pt_engine* e, int on);
/* Lore detector (pt_tsr_forward*, pt_op_dcn), PT_PRECISION_BF16 only: on = 1 (dcn_mfma_kernel) or 2 (dcn_mfma2_kernel: full-line gathers, one workgroup
 * per CU; layers with 64 outputs, the others keep the VALU blend) runs the modulated deformable convolutions
 * (model/lore/dcnv2.py:71-86, DCNv2_latest/src/cuda/dcn_v2_im2col_cuda.cu:121-191) with the bilinear blend on the matrix pipe
 * (dcn_mfma_kernel: corner lines by LDS-DMA, blend = MFMA against block-diagonal bf16 weights); on = 0 (default) keeps the VALU blend
 * with fp32 weights (dcn_fused64_kernel).  Both are bf16-mode results: the sampled columns differ by the bf16 rounding of the four
 * bilinear x mask weights (<= 2^-9 of a column; tests/test_gpu_dcn_op.py holds both to the oracle).  Measured equal in speed
 * (profiles/r04/experiments.txt: both sit on the vector-memory path's gather rate), hence the exact weights by default.
 * Ignored in PT_PRECISION_BF16X3.  Environment default: PT_DCN_MFMA. */
int pt_engine_set_dcn_mfma(pt_engine* e, int on);

/* Arithmetic of the conv nets (DESIGN.md "numerics").  PT_PRECISION_BF16: bf16 activations/weights, fp32
 * accumulate -- the throughput mode BASELINE.json's configs name.  PT_PRECISION_BF16X3: every activation and
 * weight is a (hi, lo) bf16 pair and each product is three MFMA passes (hi*hi + lo*hi + hi*lo), fp32-class
 * accuracy (~1e-5 relative) at ~1/3 of the throughput: the mode that meets the reference-fp32 parity tolerance.
 * In BF16X3 mode every bf16 NHWC activation tensor of C channels crossing this ABI (pt_det_preprocess output,
 * pt_det_forward_net input, pt_op_* tensors with split=1) has 2C channels laid out [hi(C) | lo(C)].
 * PT_PRECISION_F16X2: the same (hi, lo) bf16 activation pairs -- same tensors across this ABI -- but the convolutions and GEMMs of
 * the DB / CRNN / Lore / PicoDet graphs multiply them with SINGLE fp16 weights in two fp16 MFMA passes (hi*w + lo*w; a bf16 value
 * converts to fp16 exactly): rounding the WEIGHTS to 11 bits moves the logits by ~1.5e-4 of their scale (tools/x2_emulation.py) --
 * inside the 1e-3 tolerance at 2/3 of BF16X3's matrix work; layers without the variant run as in BF16X3.
 * PT_PRECISION_F16 (ABI 14): single-pass IEEE half -- the reference's own GPU arithmetic (base_infer_task.py:56-57 precision="fp16",
 * utils/deploy_utils.py:227-240 model.half()): every 16-bit tensor crossing this ABI (activations, the weight tiles of the blob) holds fp16
 * bits instead of bf16 bits, products accumulate in fp32, stores round to nearest even and SATURATE at +-65504 (never Inf).  Same
 * kernels, bytes and MFMA rate as PT_PRECISION_BF16 with 11 significant bits instead of 8.  The weight blobs must have been packed for
 * it (weights.py fmt="f16"); a bf16 blob under this precision (or the reverse) is refused by every forward call.  The ONNX operator
 * entry points (pt_op_*) follow the engine's precision the same way. */
#define PT_PRECISION_BF16 0
#define PT_PRECISION_BF16X3 1
#define PT_PRECISION_F16X2 2
#define PT_PRECISION_F16 3
int pt_engine_set_precision(pt_engine* e, int precision);

/* Model kinds for pt_weights_load.  A blob is the "PTW1" container written by
 * pdf_table_amd/weights.py (BN-folded, bf16 KRSC-tiled conv weights, fp32 biases). It replaces
 * torch.load + load_state_dict of model/db_net/modeling_db_net.py:53-56 and
 * model/ocr_recognition/modeling_ocr_recognition.py:102-132. */
enum {
  PT_MODEL_DB_RESNET18 = 1, /* db_net/dbnet.py:715-728 */
  PT_MODEL_CRNN = 2,        /* crnn/modeling_crnn.py:36-113 */
  PT_MODEL_LORE_DLA34 = 3,  /* lore/lore_dla_34.py:137-206 (DLASeg on dla34 + DCN), modeling_lore.py:88-95 */
  PT_MODEL_LORE_PROCESSOR = 4, /* lore/lore_processor.py:399-514 (LoreProcessModel) */
  PT_MODEL_LORE_RESNET18 = 6, /* lore/lore_detector.py:155-389 (LoreDetectModel, the 'wireless' detector) */
  PT_MODEL_DB_NAS = 7,        /* db_net/dbnet.py:693-712 (DBNasModel: ProxylessNAS backbone + LightSegDetector) */
  PT_MODEL_PPLCNET = 8,       /* cls/cls_pp_lcnet.py:164-283 (PPLCNet classifier); kinds 8 .. 8 + PT_CLS_SLOTS - 1 = classifier slots */
  PT_MODEL_PICODET = 5,     /* picodet/lcnet.py:159-259 + csp_pan.py:233-347 + pico_head.py:966-1160 (assumed config) */
  PT_MODEL_CONVNEXT_VIT = 16, /* convnext_vit/modeling_convnext_vit.py:20-45 (ConvNextViT recogniser) */
  PT_MODEL_MTL_BACKBONE = 17, /* table/mtl_tabnet/table_resnet_extra.py:205-318 (TableResNetExtra, backbone of MtlTabNet) */
  PT_MODEL_MTL_DECODER = 18,  /* table/mtl_tabnet/master_decoder.py:194-531 (MtlTabNetDecoder: structure, box and cell-content decoders) */
};
int pt_weights_load(pt_engine* e, int model_kind, const void* h_blob, size_t nbytes);
/* Same, but the blob already sits in device memory (e.g. after an RCCL broadcast from rank 0). */
int pt_weights_load_device(pt_engine* e, int model_kind, const void* d_blob, size_t nbytes, pt_stream stream);

/* ---- stage 1: layout detection (PicoDet) -------------------------------------------------------- */
#define PT_LAYOUT_LEVELS 4
#define PT_LAYOUT_HEAD_CS 40   /* fp32 values per anchor: ncls class logits, then 4 * (reg_max + 1) box logits, zero padded */
#define PT_LAYOUT_CAND_FLOATS 48 /* candidate record: int32 level, int32 anchor, 40 head values, padding */

/* Anchors of level l for an inp_h x inp_w input: strides 8, 16, 32 and the extra 5x5/s2 level (csp_pan.py:265-270). */
int pt_layout_plan(int inp_h, int inp_w, int32_t fm_h[PT_LAYOUT_LEVELS], int32_t fm_w[PT_LAYOUT_LEVELS]);

/* OCRPicodetPreProcessor.__call__ (picodet/processor_picodet.py:72-113): BGR flip, cv2.resize to inp_w x inp_h (8-bit
 * bilinear, aspect not kept), (x * 1/255 - mean) / std  ->  bf16 NHWC4 [n, inp_h, inp_w, 4] (8 in BF16X3 mode). */
int pt_layout_preprocess(pt_engine* e, const uint8_t* d_pages_rgb, int n, int h, int w, int inp_h, int inp_w,
                         uint16_t* d_out_bf16, pt_stream stream);

/* Network only: LCNet -> CSP-PAN -> PicoHead.forward_eval(export_post_process=False) before the sigmoid.
 *   d_head[l] : float32 [n, fm_h[l] * fm_w[l], PT_LAYOUT_HEAD_CS] */
int pt_layout_forward_net(pt_engine* e, const uint16_t* d_input_bf16, int n, int inp_h, int inp_w, float* d_head0,
                          float* d_head1, float* d_head2, float* d_head3, pt_stream stream);

/* Anchors whose best class score exceeds thr_lo, with their raw head values -- everything
 * OCRPicodetPostProcessor.__call__ (processor_picodet.py:184-298) can keep when thr_lo <= its score_threshold.
 *   d_counts : int32 [n] candidates found (may exceed max_cands: only the first max_cands records are stored)
 *   d_cands  : float32 [n, max_cands, PT_LAYOUT_CAND_FLOATS] */
int pt_layout_candidates(pt_engine* e, const float* d_head0, const float* d_head1, const float* d_head2,
                         const float* d_head3, int n, int inp_h, int inp_w, int num_classes, float thr_lo, int max_cands,
                         int32_t* d_counts, float* d_cands, pt_stream stream);

/* pt_layout_preprocess + pt_layout_forward_net + pt_layout_candidates with engine-owned intermediates.
 * Replaces OcrLayoutTask._preprocess + _run_model (ocr_layout_task.py:70-123). */
int pt_layout_forward(pt_engine* e, const uint8_t* d_pages_rgb, int n, int h, int w, int inp_h, int inp_w, int num_classes,
                      float thr_lo, int max_cands, int32_t* d_counts, float* d_cands, pt_stream stream);

/* Host: greedy per-class hard NMS of OCRPicodetPostProcessor (processor_picodet.py:301-348) for one page.
 *   h_boxes    : float64 [n, 5] (x1, y1, x2, y2, score)
 *   h_order    : per group (class) the candidate indices in ascending score order (numpy argsort()[-200:]);
 *   h_group_off: int64 [n_groups + 1] offsets into h_order
 *   h_picked   : int64 [len(h_order)] picked indices per group, in pick order;  h_n_picked: int32 [n_groups] */
int pt_hard_nms(const double* h_boxes, const int64_t* h_order, const int64_t* h_group_off, int n_groups,
                double iou_threshold, int top_k, int64_t* h_picked, int32_t* h_n_picked);

/* ---- stage 2: DB text detection ------------------------------------------------------------- */

/* Pre-process flavours: how a page is resized before the net. */
#define PT_DET_PRE_DB_PP 0    /* max side <= 960, sides rounded to /32, ImageNet mean/std on BGR:
                                 db_pp/image_operators.py:269-316,78-102; processor_ocr_db_pp.py:124 */
#define PT_DET_PRE_DB_TORCH 1 /* short side 736, long side ceil to /32, (x - mean)/255 on BGR:
                                 db_net/processor_ocr_dbnet.py:50-102 */
#define PT_DET_PRE_NONE 2     /* no resize (h, w must be multiples of 32); db_pp normalisation */

/* Resized (network) size the reference's pre-processor gives an h x w page. Pure host arithmetic. */
int pt_det_plan(int h, int w, int pre_flavour, int* net_h, int* net_w);

/* Full detection forward for n pages of identical size.  The detector network is the one loaded LAST with
 * pt_weights_load: PT_MODEL_DB_RESNET18 (`DBModel`) or PT_MODEL_DB_NAS (`DBNasModel`), the two backbones
 * modeling_db_net.py:47-52 chooses between.
 *   d_pages_rgb : uint8 [n, h, w, 3] RGB (as np.array(PIL.Image) gives; the BGR flip is done inside)
 *   d_prob      : float32 [n, net_h, net_w]   sigmoid probability map   (may be NULL)
 *   d_bitmap    : uint32  [n, net_h, net_w/32] bit x%32 of word x/32 = (prob > thresh), optionally
 *                 2x2-dilated (DBPostProcess.__call__, db_pp/processor_ocr_db_pp.py:291-311) (may be NULL)
 * Replaces OcrDetectionTask._preprocess + _run_model (ocr_detection_task.py:77-124) and the
 * `pred > thresh` step of the post-processor. */
int pt_det_forward(pt_engine* e, const uint8_t* d_pages_rgb, int n, int h, int w, int pre_flavour,
                   float thresh, int use_dilation, float* d_prob, uint32_t* d_bitmap, pt_stream stream);

/* Network only: d_input is bf16 NHWC4 [n, net_h, net_w, 4] (4th channel zero), already normalised.
 * Used by parity tests to feed the net the exact tensor the oracle sees. d_logits (optional) gets the
 * pre-sigmoid fp32 map. */
int pt_det_forward_net(pt_engine* e, const uint16_t* d_input_bf16, int n, int net_h, int net_w,
                       float* d_prob, float* d_logits, pt_stream stream);

/* Pre-process only (tests / CPU-baseline inputs): writes bf16 NHWC4 [n, net_h, net_w, 4]. */
int pt_det_preprocess(pt_engine* e, const uint8_t* d_pages_rgb, int n, int h, int w, int pre_flavour,
                      uint16_t* d_out_bf16, pt_stream stream);

/* prob -> bit-packed bitmap (see pt_det_forward). */
int pt_det_bitmap(pt_engine* e, const float* d_prob, int n, int net_h, int net_w, float thresh,
                  int use_dilation, uint32_t* d_bitmap, pt_stream stream);

/* box_score_fast (db_pp/processor_ocr_db_pp.py:253-268) for nb quads on the device.
 *   d_boxes : float32 [nb, 9] = (page index, x0,y0,x1,y1,x2,y2,x3,y3) in net-map pixels
 *   d_scores: float32 [nb] mean probability inside the quad (cv2.fillPoly mask semantics) */
int pt_det_box_scores(pt_engine* e, const float* d_prob, int n, int net_h, int net_w,
                      const float* d_boxes, int nb, float* d_scores, pt_stream stream);

/* Host side of DBPostProcess.boxes_from_bitmap (db_pp/processor_ocr_db_pp.py:174-251), split in the
 * two halves that sit either side of pt_det_box_scores.  Pure CPU (C++), no GPU needed.
 *
 * pt_db_candidates: contour tracing (cv2.findContours RETR_LIST / CHAIN_APPROX_SIMPLE), first
 *   max_candidates contours, min-area rectangle + get_mini_boxes ordering, drops sside < min_size.
 *   h_bitmap: uint32 [net_h, net_w/32] of ONE page. h_boxes: float32 [cap, 8]; returns count in *n_out. */
int pt_db_candidates(const uint32_t* h_bitmap, int net_h, int net_w, int max_candidates, float min_size,
                     float* h_boxes, float* h_sside, int cap, int* n_out);
/* pt_db_finalize: score gate, unclip (Clipper round-join offset by area*ratio/perimeter), second
 *   min-area rectangle, sside gate (min_size + 2), rescale to the source page, round, clip.
 *   Output h_out: int32 [cap, 8]; h_out_scores float32 [cap]. */
int pt_db_finalize(const float* h_boxes, const float* h_scores, int nb, float box_thresh, float unclip_ratio,
                   float min_size, int net_h, int net_w, int dest_h, int dest_w, int post_flavour, int32_t* h_out,
                   float* h_out_scores, int cap, int* n_out);
/* Batch forms of the two calls above: n pages in one call, walked by n_threads threads inside the library (0 = all
 * hardware threads).  h_bitmaps uint32 [n, net_h, net_w/32]; h_boxes float32 [n, cap, 8] (page i's candidates at
 * [i, 0 .. n_out[i])), h_sside float32 [n, cap] or NULL.  pt_db_finalize_batch: h_scores float32 [n, cap], nb int [n]
 * candidates per page; h_out int32 [n, cap, 8]; filter != 0 additionally applies PPOcrDetectionPostProcessor.
 * filter_tag_det_res (db_pp/processor_ocr_db_pp.py:344-386: clockwise order, clip to the page, drop boxes <= 3 px) and
 * writes the surviving boxes as float32 [n, cap, 8] to h_out_f32 (n_out then counts those). */
int pt_db_candidates_batch(const uint32_t* h_bitmaps, int n, int net_h, int net_w, int max_candidates, float min_size,
                           int n_threads, float* h_boxes, float* h_sside, int cap, int* n_out);
int pt_db_finalize_batch(const float* h_boxes, const float* h_scores, const int* nb, int n, int cap, float box_thresh,
                         float unclip_ratio, float min_size, int net_h, int net_w, int dest_h, int dest_w, int post_flavour,
                         int filter, int n_threads, int32_t* h_out, float* h_out_f32, float* h_out_scores, int* n_out);
#define PT_DET_POST_DB_PP 0    /* float32 box / W * dest, round, clip, astype(int16): processor_ocr_db_pp.py:211-217 */
#define PT_DET_POST_DB_TORCH 1 /* box.astype(int32) first, then the same in float64: ocr_detection_utils.py:198-206 */

/* ---- stage 3: text-line recognition (CRNN) -------------------------------------------------------- */
#define PT_REC_H 32     /* OCRRecognitionConfig.img_height                                        */
#define PT_REC_W 640    /* img_width for CRNN (configuration_ocr_document.py:47-48)                */
#define PT_REC_T 160    /* time steps = W / 4 (crnn/modeling_crnn.py: two 2x2 pools)               */
#define PT_REC_NCLS 7644 /* crnn/modeling_crnn.py:90                                               */

/* One text line to cut out of a page: the INVERSE perspective matrix (row-major 3x3, destination -> source, what
 * cv2.warpPerspective evaluates) of OcrCommonUtils.crop_image (utils/ocr/ocr_common_utils.py:214-266) and the
 * crop size int(img_width) x int(img_height) computed there. */
typedef struct pt_rec_line {
  double minv[9];
  int32_t page;
  int32_t crop_w;
  int32_t crop_h;
  int32_t reserved;
} pt_rec_line;

/* Recognise n_lines text lines cut from n_pages pages (uint8 [n_pages,h,w,3] RGB, on the GPU).
 *   d_lines   : device array of pt_rec_line;  h_crop_px: HOST array [n_lines] of crop_w*crop_h (sizes the crop
 *               scratch without a device round trip)
 *   d_ids     : int32 [n_lines, PT_REC_T] arg-max class per time step (0 = CTC blank); collapse on the host
 *   d_maxlogit: float [n_lines, PT_REC_T] the winning logit (may be NULL)
 * Replaces the per-line loop OcrSystemTask.text_recognition -> crop_image -> OcrRecognitionTask
 * (ocr_system_task.py:296-336, ocr_recognition_task.py:81-136) and the arg-max of OCRRecognition.postprocess. */
int pt_rec_forward(pt_engine* e, const uint8_t* d_pages_rgb, int n_pages, int h, int w, const pt_rec_line* d_lines,
                   const int64_t* h_crop_px, int n_lines, int32_t* d_ids, float* d_maxlogit, pt_stream stream);
/* Same, for lines that are ALREADY cropped (what OcrRecognitionTask.__call__ receives, ocr_recognition_task.py:67-79):
 * d_crops_rgb is the concatenation of the n_lines uint8 [crop_h, crop_w, 3] images; only crop_w / crop_h of d_lines
 * are read.  Resize (OCRRecognitionPreprocessor.keepratio_resize) + CRNN + arg-max. */
int pt_rec_forward_crops(pt_engine* e, const uint8_t* d_crops_rgb, const pt_rec_line* d_lines, const int64_t* h_crop_px,
                         int n_lines, int32_t* d_ids, float* d_maxlogit, pt_stream stream);
/* Network only: d_gray bf16 [n, 32, 640] (BF16X3 mode: [n, 32, 640, 2] = hi, lo), values in [0, 1]. */
int pt_rec_forward_net(pt_engine* e, const uint16_t* d_gray, int n, int32_t* d_ids, float* d_maxlogit, pt_stream stream);
/* Crop + resize + gray only (tests): writes d_gray as above. */
int pt_rec_preprocess(pt_engine* e, const uint8_t* d_pages_rgb, int n_pages, int h, int w, const pt_rec_line* d_lines,
                      const int64_t* h_crop_px, int n_lines, uint16_t* d_gray, pt_stream stream);

/* ---- stage 3, ConvNextViT recogniser (OcrRecognitionTask(model="ConvNextViT"), BASELINE.json configs[4]) ---------
 * OCRRecognitionPreprocessor with do_chunking (model/ocr_recognition/processor_ocr_recognition.py:44-62,94-113): keep-ratio
 * resize to 32 x 804, three 300-px chunks at 252-px steps; ConvNextViT.forward (model/convnext_vit/modeling_convnext_vit.py:
 * 38-45, modeling_convnext.py:29-131, modeling_vit.py:31-143): every chunk through the ConvNext CNN and the 12-layer ViT, the
 * chunks' 75 tokens stitched to 201 per line, Linear(192 -> 7644); then the arg-max of OCRRecognitionPostProcessor (:147-150).
 *   d_ids / d_maxlogit: [n_lines, PT_CVIT_T] arg-max class (0 = CTC blank, vocabulary from class 2) and its logit.
 *   h_crop_wh (HOST int32 [n_lines][2] = crop_w, crop_h; may be NULL): with the crop sizes known on the host, the 300-px chunks
 *   that hold no text (text width after the resize <= 252 j for chunk j) are not computed line by line: every all-padding
 *   chunk yields the same 75 tokens, so ONE is computed per call and shared -- bit-identical results. */
#define PT_CVIT_W 804          /* OCRRecognitionConfig.img_width with do_chunking (configuration_ocr_recognition.py:47) */
#define PT_CVIT_CHUNK_W 300    /* processor_ocr_recognition.py:105-106 */
#define PT_CVIT_CHUNK_STEP 252 /* 300 - 48 */
#define PT_CVIT_T 201          /* modeling_vit.py:134 */
#define PT_CVIT_NCLS 7644      /* modeling_convnext_vit.py:33 */
/* lines cut from resident pages (as pt_rec_forward) */
int pt_rec_cvit_forward(pt_engine* e, const uint8_t* d_pages_rgb, int n_pages, int h, int w, const pt_rec_line* d_lines,
                        const int64_t* h_crop_px, const int32_t* h_crop_wh, int n_lines, int32_t* d_ids, float* d_maxlogit,
                        pt_stream stream);
/* lines that are already cropped (as pt_rec_forward_crops) */
int pt_rec_cvit_forward_crops(pt_engine* e, const uint8_t* d_crops_rgb, const pt_rec_line* d_lines, const int64_t* h_crop_px,
                              const int32_t* h_crop_wh, int n_lines, int32_t* d_ids, float* d_maxlogit, pt_stream stream);
/* Network only.  d_gray fp32 in [0, 1]: layout 0 = chunks [3 n_lines, 32, 300] (the tensor the reference's model receives,
 * gray), layout 1 = lines [n_lines, 32, 804] (chunk j = columns [252 j, 252 j + 300)).  h_text_w (HOST int32 [n_lines], may be
 * NULL): columns >= h_text_w[i] of line i are zero padding (same chunk sharing as h_crop_wh above). */
int pt_rec_cvit_forward_net(pt_engine* e, const float* d_gray, int layout, int n_lines, const int32_t* h_text_w, int32_t* d_ids,
                            float* d_maxlogit, pt_stream stream);
/* Resize + gray only (tests): d_gray fp32 [n_lines, 32, 804]. */
int pt_rec_cvit_preprocess_crops(pt_engine* e, const uint8_t* d_crops_rgb, const pt_rec_line* d_lines, const int64_t* h_crop_px,
                                 int n_lines, float* d_gray, pt_stream stream);

/* PP-OCR recognition pre-processor -- PPOcrRecPreProcessor (model/ocr_rec_pp/processor_ocr_rec_pp.py:69-135,
 * resize_norm_img :43-67), the pre-processing of the recogniser the reference's system path selects
 * (fix_model_names, model/ocr_pdf/configuration_ocr_document.py:138-141).  The HOST orders the lines by aspect ratio and
 * groups them into mini-batches of rec_batch_num (np.argsort, as the reference; pdf_table_amd/rec_pp_stage.py); one item =
 * one line of that plan: which crop, the width it is resized to (height img_h = 48), the padded width img_w of its
 * mini-batch and where its [3, img_h, img_w] fp32 block starts in d_out (float offset; a mini-batch's items are
 * consecutive, which makes d_out the concatenation of the reference's [b, 3, 48, img_w] NCHW arrays).
 * Pixels: cv2.resize 8-bit INTER_LINEAR semantics, then (x / 255 - 0.5) / 0.5 in fp32, zeros right of resized_w. */
typedef struct pt_rec_pp_item {
  int32_t line;
  int32_t resized_w;
  int32_t img_w;
  int32_t reserved;
  int64_t out_off;
} pt_rec_pp_item;
/* lines cut from resident pages (crop_image on the device, as pt_rec_forward) */
int pt_rec_pp_preprocess(pt_engine* e, const uint8_t* d_pages_rgb, int n_pages, int h, int w, const pt_rec_line* d_lines,
                         const int64_t* h_crop_px, int n_lines, const pt_rec_pp_item* d_items, int n_items, int img_h,
                         int max_img_w, float* d_out, pt_stream stream);
/* lines that are already cropped (what OcrRecognitionTask.__call__ receives); d_crops_rgb / d_lines as pt_rec_forward_crops */
int pt_rec_pp_preprocess_crops(pt_engine* e, const uint8_t* d_crops_rgb, const pt_rec_line* d_lines, const int64_t* h_crop_px,
                               int n_lines, const pt_rec_pp_item* d_items, int n_items, int img_h, int max_img_w, float* d_out,
                               pt_stream stream);

/* ---- stage 4: table structure recognition (Lore) ------------------------------------------------- */
/* One table crop of a resident page and the INVERSE (destination -> crop) affine map of
 * TableLorePreProcessor.process (lore/processer_lore.py:66-90), i.e. cv::invertAffineTransform of
 * get_affine_transform(c, s, 0, [inp_w, inp_h]) (lineless_table_process.py:403-438), computed on the host in float64.
 * The crop is the integer box of crop_image_by_box (utils/ocr/ocr_common_utils.py:269-284). */
typedef struct pt_tsr_table {
  double minv[6];
  int32_t page, x0, y0, crop_w, crop_h, reserved;
} pt_tsr_table;

/* Warp + normalise n table crops into the detector's input (cv2.warpAffine INTER_LINEAR fixed-point semantics, zero
 * border, then ((x / 255) - mean) / std with Lore's constants).
 *   d_pages_rgb : uint8 [n_pages, ph, pw, 3];  d_tables : pt_tsr_table [n];  bgr != 0: feed channels reversed, as the
 *                 reference does for path / PIL inputs (processer_lore.py:48-64,152)
 *   d_out_bf16  : bf16 NHWC4 [n, inp_h, inp_w, 4] (8 channels = hi | lo in BF16X3 mode) */
int pt_tsr_preprocess(pt_engine* e, const uint8_t* d_pages_rgb, int n_pages, int ph, int pw, const pt_tsr_table* d_tables,
                      int n, int inp_h, int inp_w, int bgr, uint16_t* d_out_bf16, pt_stream stream);

/* Detector network only (LoreModel.forward's `self.detect_infer_model(pixel_values)`, lore/modeling_lore.py:147;
 * DLASeg.forward lore/lore_dla_34.py:184-196).
 *   d_input_bf16 : bf16 NHWC4 [n, H, W, 4] (4th channel zero; [hi rgb0 | lo rgb0] = 8 channels in BF16X3 mode),
 *                  already warped + normalised (TableLorePreProcessor.process, lore/processer_lore.py:66-109);
 *                  H, W multiples of 32
 *   outputs      : fp32 NHWC head maps at H/4 x W/4, channel strides PT_TSR_CS_*: hm [.,8] (2 valid: cell centres,
 *                  corners), st [.,8], wh [.,8], ax [.,256], cr [.,256], reg [.,8] (2 valid) -- the dict `z` of
 *                  DLASeg.forward, pre-sigmoid */
#define PT_TSR_CS_SMALL 8
#define PT_TSR_CS_FEAT 256
int pt_tsr_forward_net(pt_engine* e, const uint16_t* d_input_bf16, int n, int H, int W, float* d_hm, float* d_st,
                       float* d_wh, float* d_ax, float* d_cr, float* d_reg, pt_stream stream);

/* Same contract for the 'wireless' detector (LoreDetectModel.forward, lore/lore_detector.py:353-389; weights
 * PT_MODEL_LORE_RESNET18); H, W multiples of 64. */
int pt_tsr_forward_net_wireless(pt_engine* e, const uint16_t* d_input_bf16, int n, int H, int W, float* d_hm, float* d_st,
                                float* d_wh, float* d_ax, float* d_cr, float* d_reg, pt_stream stream);

/* Heat-map + corner-point decode (process_detect_output, lore/lineless_table_process.py:592-655: corner_decode :97-124,
 * ctdet_4ps_decode :127-267 incl. the wiz_rev vertex snapping :188-236, logi = ax + cr_feat :648) for n tables.
 *   head maps   : as written by pt_tsr_forward_net (fp32 NHWC at h x w)
 *   wiz_rev     : LoreConfig.wiz_rev (configuration_lore.py:79-92); vis_thresh: LoreConfig.vis_thresh
 *   d_counts    : int32 [n]            cells with final score >= vis_thresh (= rows of slct_logi_feat, :568-571)
 *   d_dets      : float32 [n, 3000, 9] x0,y0..x3,y3 in feature-map pixels + score, sorted like the reference's `dets`
 *   d_logi      : float32 [n, 3000, 256] logic features of the same rows
 * Only rows [0, number of peaks above vis_thresh) are written; rows past d_counts[i] are not part of the result. */
#define PT_TSR_MAX_CELLS 3000
int pt_tsr_decode(pt_engine* e, const float* d_hm, const float* d_st, const float* d_wh, const float* d_ax,
                  const float* d_cr, const float* d_reg, int n, int h, int w, int wiz_rev, float vis_thresh,
                  int32_t* d_counts, float* d_dets, float* d_logi, pt_stream stream);

/* pt_tsr_forward_net + pt_tsr_decode in one call for the DLA-34 detector (PT_MODEL_LORE_DLA34), same outputs as
 * pt_tsr_decode.  Of the six head maps the decode needs only `hm` everywhere: `reg` / `wh` / `st` are read at the kept
 * peaks (<= 3000 cells + 5000 corners per table, lineless_table_process.py:127-177), `ax` / `cr` at the final cells'
 * centres and corner pixels (:254-263, _get_4ps_feat :39-63).  Here those five heads run on 3x3-pixel patches around
 * exactly these positions instead of on the whole map; every value that is read is produced by the same kernels in the
 * same order as in the dense map (bit-identical to the two-call path when both use one conv kernel family). */
int pt_tsr_forward_decode(pt_engine* e, const uint16_t* d_input_bf16, int n, int in_h, int in_w, int wiz_rev,
                          float vis_thresh, int32_t* d_counts, float* d_dets, float* d_logi, pt_stream stream);

/* Logical-location processor (LoreProcessModel.forward, lore/lore_processor.py:465-514, evaluation branch) for the
 * cells of n_tables tables at once.
 *   d_logi, d_dets : as written by pt_tsr_decode;  h_counts : HOST int32 [n_tables] = its d_counts copied back
 *   use_2dpe       : LoreConfig.wiz_2dpe -- add the x/y position embeddings of the (int-truncated, [0,255]-clamped)
 *                    quad coordinates 0,1,2,5 (:486-491; lineless_table_process.py:576-589)
 *   d_logic, d_stacked : float32 [n_tables, 3000, 4]: rows [0, h_counts[i]) = logic_axis / stacked_axis (post-ReLU,
 *                    not yet rounded: process_logic_output, lineless_table_process.py:658-663, is host work) */
int pt_tsr_process(pt_engine* e, const float* d_logi, const float* d_dets, const int32_t* h_counts, int n_tables,
                   int use_2dpe, float* d_logic, float* d_stacked, pt_stream stream);

/* ---- single operator (parity tests of the conv kernel variants) --------------------------------- */
/* NHWC bf16 convolution on the MFMA implicit-GEMM kernel. d_w_tiled is [N/64][Cin/32][ks*ks][64][32] bf16
 * (pdf_table_amd/weights.py:tile_conv_weight), d_bias fp32 [N]. ks in {1,3} (pad ks/2), stride in {1,2}.
 * rep > 1: every output pixel is replicated rep x rep (fused nn.Upsample(nearest)); shuffle_cout > 0:
 * N = 4*shuffle_cout and the output is the 2x up-sampled ConvTranspose2d(k=2,s=2) image;
 * res_mode 1: + d_res (same shape as the un-replicated output); 2: + nearest-x2-upsampled d_res. */
int pt_op_conv2d(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int Cin, const uint16_t* d_w_tiled,
                 const float* d_bias, int N, int ks, int stride, uint16_t* d_out, int out_cstride, int out_coff,
                 int rep, int shuffle_cout, const uint16_t* d_res, int res_mode, int relu, int split, int out_lo_off,
                 pt_stream stream);
/* split=1 (BF16X3): d_in/d_res/d_out carry (hi | lo) channel groups, d_w_tiled is [N/64][3*Cin/32][ks*ks][64][32]
 * (K chunks: w_hi for x_hi, w_hi for x_lo, w_lo for x_hi), out_lo_off = channel distance hi -> lo in d_out. */

/* Fused modulated deformable 3x3 convolution (pad 1, stride 1, dilation 1, one deformable group) + bias (+ ReLU): the operator
 * DCN.forward (model/lore/dcnv2.py:71-86) hands to torchvision.ops.deform_conv2d, sampling rule of
 * model/lore/DCNv2_latest/src/cpu/dcn_v2_im2col_cpu.cpp:26-55,123-190 (bilinear, zero outside (-1,H)x(-1,W), per-corner bounds).
 *   d_in  bf16 NHWC [B,H,W,C] ([hi(C) | lo(C)] when split), C % 32 == 0
 *   d_om  fp32 [B*H*W][32]: channel 2k / 2k+1 = (dy, dx) of tap k, 18+k = mask LOGIT of tap k (the sigmoid is applied here), 27..31 unused
 *   d_w_tiled: the [N, 9*C, 1, 1] weight (K = tap * C + c) tiled like a 1x1 convolution (weights.py:tile_conv_weight, or
 *              tile_conv_weight_x3 when split), d_bias fp32 [N], N % 64 == 0
 *   d_out bf16 [B,H,W,N] ([hi | lo] when split).  Since ABI 12. */
int pt_op_dcn(pt_engine* e, const uint16_t* d_in, const float* d_om, int B, int H, int W, int C, const uint16_t* d_w_tiled,
              const float* d_bias, int N, uint16_t* d_out, int relu, int split, pt_stream stream);

/* Stem of the ResNet-18 backbone: 7x7 s2 p3 conv (BN folded) + ReLU on a bf16 NHWC4 image (dbnet.py:272-275).
 * d_w is bf16 [64][7][8][4] (K padded: tap s=7 and channel 3 are zero), d_bias fp32 [64]; out bf16 [B,H/2,W/2,64]. */
int pt_op_stem7x7(pt_engine* e, const uint16_t* d_in, int B, int H, int W, const uint16_t* d_w, const float* d_bias,
                  uint16_t* d_out, int split, pt_stream stream);
/* MaxPool2d(3, 2, 1) on bf16 NHWC (dbnet.py:276). */
int pt_op_maxpool3x3s2(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int C, uint16_t* d_out, int split,
                       pt_stream stream);
/* ConvTranspose2d(64 -> 1, k=2, s=2) + Sigmoid (dbnet.py:539): in bf16 [B,H,W,64], d_w bf16 [4][64] (quadrant
 * dy*2+dx major), d_bias fp32 [1]; d_prob / d_logits fp32 [B,2H,2W] (either may be NULL). */
int pt_op_db_head_final(pt_engine* e, const uint16_t* d_in, int B, int H, int W, const void* d_w, const float* d_bias,
                        float* d_prob, float* d_logits, int split, pt_stream stream);

/* ---- single operators of the generic ONNX layer-list executor (pdf_table_amd/onnx_exec.py) --------------------
 * Replaces: onnxruntime's execution of an arbitrary graph behind BaseInferTask.infer (model/ocr_pdf/base_infer_task.py:
 * 366-370, sessions built by utils/deploy_utils.py:243-280).  bf16 NHWC, C a multiple of 8; convolutions go to
 * pt_op_conv2d.  act: 0 none, 1 ReLU, 2 hardswish.
 * split (since ABI 12; every operator below except the pure data movers): 0 = plain bf16 tensors; 1 = the tolerance mode's (hi | lo) tensors --
 * a pixel / token row holds [hi(C) | lo(C)], value = hi + lo, arithmetic in fp32 on the sum, result split again (the convention of
 * PT_PRECISION_BF16X3).  pt_op_act / pt_op_mul then need C (channels of one half; n_elems counts VALUES, i.e. pixels x C); pt_op_copy_channels
 * and pt_op_upsample_nearest move halves like any channels (the caller addresses them through the channel stride / offset). */
/* depthwise k x k (k 3 or 5, pad k/2, stride 1 or 2) + bias + act; d_w_taps fp32 [k*k][C], d_bias fp32 [C] */
int pt_op_dwconv(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int C, const float* d_w_taps, const float* d_bias, int k,
                 int stride, int act, uint16_t* d_out, int split, pt_stream stream);
/* element-wise a + b over npix pixels of C channels */
int pt_op_add(pt_engine* e, const uint16_t* d_a, const uint16_t* d_b, uint16_t* d_out, long long npix, int C, int split, pt_stream stream);
/* MaxPool2d(3, 2, 1), or a non-overlapping k x k pool (stride k, no padding, H and W divisible by k) */
int pt_op_maxpool(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int C, int k, int stride, int pad, uint16_t* d_out,
                  int split, pt_stream stream);
/* AveragePool k x k, stride k, no padding (H and W divisible by k) */
int pt_op_avgpool(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int C, int k, uint16_t* d_out, int split, pt_stream stream);
/* GlobalAveragePool: [B, HW, C] -> bf16 [B, C]; d_scratch: pt_op_chan_mean_scratch_floats(B, C) floats */
int pt_op_chan_mean(pt_engine* e, const uint16_t* d_in, int B, int HW, int C, float* d_scratch, uint16_t* d_mean, int split, pt_stream stream);
int pt_op_chan_mean_scratch_floats(int B, int C);
/* x [B, HW, C] * gate [B, C] (the Mul of a squeeze-and-excitation block) */
int pt_op_scale_channels(pt_engine* e, const uint16_t* d_in, const uint16_t* d_gate, int B, int HW, int C, uint16_t* d_out,
                         int split, pt_stream stream);
/* stand-alone activation over n_elems (multiple of 8) values; kind: 1 ReLU, 2 hardswish, 4 sigmoid,
 * 5 hardsigmoid = max(0, min(1, alpha x + beta)), 6 ReLU6, 7 GELU (erf form), 8 swish x * sigmoid(x) */
int pt_op_act(pt_engine* e, const uint16_t* d_in, long long n_elems, int kind, float alpha, float beta, uint16_t* d_out, int C, int split,
              pt_stream stream);
/* Data movement of the executor (ONNX Concat / Slice / Split over channels, nearest Resize): dst[pix][dst_coff + c] = src[pix][src_coff + c] for
 * c < n; nearest-neighbour up-sampling by an integer factor -> [B, H f, W f, C]; element-wise product of two equally shaped tensors */
int pt_op_copy_channels(pt_engine* e, const uint16_t* d_src, long long npix, int src_cstride, int src_coff, uint16_t* d_dst, int dst_cstride, int dst_coff,
                        int n, pt_stream stream);
int pt_op_upsample_nearest(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int C, int factor, uint16_t* d_out, pt_stream stream);
int pt_op_mul(pt_engine* e, const uint16_t* d_a, const uint16_t* d_b, uint16_t* d_out, long long n_elems, int C, int split, pt_stream stream);
/* nbytes from src to dst (16-byte aligned) by a kernel on `stream`; either side may be PINNED HOST memory (hipHostMalloc / torch pin_memory: mapped into
 * the device's address space).  For the few small transfers a host thread WAITS for while the compute stream holds a backlog: an asynchronous
 * hipMemcpy goes through a DMA engine's in-order queue and can sit there behind copies of other streams that wait for kernels still to run
 * (measured: 15-50 ms in a 7 MB copy); a kernel on a high-priority stream cannot.  Replaces nothing in the reference (torch's .cpu()). */
int pt_copy_bytes(pt_engine* e, const void* src, void* dst, long long nbytes, pt_stream stream);
/* Sequence operators (the attention / LayerNorm / soft-max blocks of SVTR-type recognisers such as PP-OCRv4 rec, the recogniser
 * fix_model_names() selects: model/ocr_pdf/configuration_ocr_document.py:138-141).  Token rows bf16 [rows, c_pad], the first c channels real.
 * pt_op_layernorm: LayerNormalization over the channels (biased variance, eps inside the root), padded channels written as zeros.
 * pt_op_softmax: Softmax over the channels -> fp32 [rows, c] (d_out_f32, network outputs) OR bf16 [rows, c_pad] (d_out_bf16); pass one.
 * pt_op_attention: multi-head self-attention of rows holding [q | k | v], channel = part * heads * d + head * d + j (the fused-qkv layout of
 * timm / PaddleOCR SVTR blocks): out[b, t, head * d + j] = softmax_k(scale q.k) v; T <= 1024, d <= 64. */
int pt_op_layernorm(pt_engine* e, const uint16_t* d_in, long long rows, int c_pad, int c, const float* d_gamma, const float* d_beta, float eps,
                    uint16_t* d_out, int split, pt_stream stream);
int pt_op_softmax(pt_engine* e, const uint16_t* d_in, long long rows, int c_pad, int c, float* d_out_f32, uint16_t* d_out_bf16, int split, pt_stream stream);
int pt_op_attention(pt_engine* e, const uint16_t* d_qkv, int B, int T, int heads, int d, int qkv_cstride, float scale, uint16_t* d_out, int out_cstride,
                    int split, pt_stream stream);

/* ---- introspection used by bench.py (HIP-event timing of the dominant kernel) ------------------ */
/* ---- image classification (PP-LCNet) ------------------------------------------------------------
 * Replaces ClsImagePulcTask._preprocess/_run_model (ocr_pdf/cls_image_pulc_task.py:48-84): PPLCNetImageProcessor
 * (cls/image_processing_pplcnet.py:327-455: Pillow bilinear resize, * 1/255, ImageNet mean/std) and PPLCNet.forward
 * (cls/cls_pp_lcnet.py:262-283).  The soft-max / top-k of the post-processor (:155-192) stays on the host.
 * Up to PT_CLS_SLOTS classifiers are resident at once: load slot k with pt_weights_load(e, PT_MODEL_PPLCNET + k, ..).
 * `textline`: the task's stride_list is [2,[2,1],[2,1],[2,1],[2,1]] (textline_orientation, language_classification;
 * cls/configuration_cls_pulc.py:20-39). */
#define PT_CLS_SLOTS 4
#define PT_CLS_MAX_CLASSES 16
typedef struct pt_cls_image {
  int64_t offset; /* bytes from d_base to an RGB uint8 [h, w, 3] image */
  int32_t h, w;
} pt_cls_image;

/* d_images: DEVICE array of n descriptors; max_h / max_w: upper bounds of their sizes (host-known);
 * d_out_bf16: bf16 NHWC4 [n, out_h, out_w, 4] (8 channels [hi rgb0 | lo rgb0] in PT_PRECISION_BF16X3). */
int pt_cls_preprocess(pt_engine* e, const uint8_t* d_base, const pt_cls_image* d_images, int n, int max_h, int max_w,
                      int out_h, int out_w, uint16_t* d_out_bf16, pt_stream stream);
/* d_logits: float32 [n, PT_CLS_MAX_CLASSES], the first *n_classes columns valid (n_classes may be NULL). */
int pt_cls_forward_net(pt_engine* e, int slot, const uint16_t* d_input_bf16, int n, int in_h, int in_w, int textline,
                       float* d_logits, int* n_classes, pt_stream stream);
/* pre-process + network for n images. */
int pt_cls_forward(pt_engine* e, int slot, const uint8_t* d_base, const pt_cls_image* d_images, int n, int max_h, int max_w,
                   int out_h, int out_w, int textline, float* d_logits, int* n_classes, pt_stream stream);
/* Text lines cut from resident pages (the per-line loop of OcrSystemTask.text_line_orientation,
 * ocr_system_task.py:395-439: order_point -> crop_image -> classifier): d_lines / h_crop_px as for pt_rec_forward;
 * max_crop_h / max_crop_w: upper bounds of the lines' crop sizes. */
int pt_cls_forward_lines(pt_engine* e, int slot, const uint8_t* d_pages_rgb, int n_pages, int h, int w,
                         const pt_rec_line* d_lines, const int64_t* h_crop_px, int n_lines, int max_crop_h, int max_crop_w,
                         int out_h, int out_w, int textline, float* d_logits, int* n_classes, pt_stream stream);

/* on = 1: every kernel launch of the forward calls is bracketed by hipEvents on `stream`; on = 2 + class: only the
 * launches of that kernel class (an event pair costs a few microseconds of idle GPU per launch, which adds up over the
 * ~1000 launches of a four-stage step); 0: off.  pt_profile_read returns accumulated milliseconds, launch count and
 * algorithmic FLOP per kernel class. */
#define PT_PROF_CONV3X3 0
#define PT_PROF_CONV1X1 1
#define PT_PROF_STEM 2
#define PT_PROF_OTHER 3
#define PT_PROF_NCLASS 4
int pt_profile_enable(pt_engine* e, int on);
int pt_profile_read(pt_engine* e, double* ms_per_class, long long* launches_per_class, double* flop_per_class);
/* Per-label readout of the launches recorded since the last read (pt_profile_enable(e, 1)): text lines
 * "label<TAB>launches<TAB>ms<TAB>algorithmic FLOP<TAB>algorithmic bytes<NL>" into buf (NUL-terminated; PT_ERR_INVALID if cap is too small).
 * Consumes the records like pt_profile_read.  bench.py's roofline.by_class is built from it.  Since ABI 12. */
int pt_profile_read_labels(pt_engine* e, char* buf, int cap);

/* ---- MtlTabNet (second half): backbone, then the decoders below -------------------------------
 * TableResNetExtra.forward (model/table/mtl_tabnet/table_resnet_extra.py:268-318) of the configuration in
 * mtl_tabnet_config.py:41-53: d_x bf16 [n, H, W, 32] NHWC (channels 0..2 = the normalised image, the rest zero; BF16X3:
 * [hi 32 | lo 32]), H and W multiples of 8 -> d_f3 fp32 [n, H/8, W/8, 512], the feature map the decoders read (feat[-1]). */
int pt_tsr_mtl_backbone_net(pt_engine* e, const uint16_t* d_x, int n, int H, int W, float* d_f3, pt_stream stream);

/* MtlTabNet test-time pre-processing (mtl_tabnet_config.py:136-160: TableResize(keep_ratio, long_size = 480) -> TablePad(480 x 480,
 * pad_val 0) -> ToTensorOCR -> NormalizeOCR(mean 0.5, std 0.5); model/table/lgpma/lgpma_preprocess.py:1067-1093, 1128-1152) of n
 * table crops given as rectangles of resident pages (pt_tsr_table.page / x0 / y0 / w / h; the affine fields are not read).  The
 * long side becomes `size`, the short side int(size / long * short) (cv2.resize INTER_LINEAR, 8-bit), the rest of the size x size
 * canvas is 0 BEFORE the normalisation, i.e. -1 after it.  d_out bf16 [n, size, size, 32] (BF16X3: [hi 32 | lo 32]), channels
 * 0..2 in the crop's own channel order (mean and std are the same for all three).  The resized size of crop i is what
 * pt_tsr_mtl_resized_size returns (host arithmetic in double, as the reference's Python floats). */
int pt_tsr_mtl_preprocess(pt_engine* e, const uint8_t* d_pages, int n_pages, int h, int w, const pt_tsr_table* d_tables, int n, int size,
                          uint16_t* d_out, pt_stream stream);
void pt_tsr_mtl_resized_size(int crop_w, int crop_h, int size, int32_t* out_w, int32_t* out_h);

/* The three decoders of MtlTabNet (master_decoder.py:354-517: greedy_forward / decode_test) for a batch of tables, KV-cached.
 * pt_weights_load(PT_MODEL_MTL_DECODER) first.  pt_tsr_mtl_decoder_config: the blob's 13 configuration integers (structure
 * classes, cell classes, <SOS>, <EOS>, <PAD>, max_seq_len, the same three ids and the limit of the cell alphabet, the two
 * structure ids that open a non-empty cell, padded d_ff).
 * pt_tsr_mtl_structure: d_f3 fp32 [n, hw, 512] = the backbone's last map (pt_tsr_mtl_backbone_net; NHWC = flattened row-major
 * like the reference's view + permute).  Every table is decoded as the reference decodes a batch of ONE: it stops at its own <EOS>
 * (or after max_seq_len + 1 positions).  d_tag_logits fp32 [n, max_seq_len + 1, classes] (raw cls_fc outputs), d_boxes fp32 [n,
 * max_seq_len + 1, 4] (sigmoid); h_lens[n] = positions written per table (the length of the reference's output), h_cell_counts[n] =
 * positions whose arg-max is '<td></td>' or '<td' (decode_test :389-392).  Synchronous: the call polls the sequences' end flags.
 * pt_tsr_mtl_cells: the cell-content decoder for those cells (ordered by table, then position; total = sum of h_cell_counts):
 * d_cell_ids int32 / d_cell_prob fp32 [total, max_seq_len_cell + 1] = arg-max and its soft-max probability per position,
 * d_cell_logits fp32 [total, max_seq_len_cell + 1, cell classes] or NULL; h_steps[n] = positions decoded per table (all cells of a
 * table share it: the loop ends when every cell emitted <EOS> in the same step, :451-453; 0 = no cells).
 * force_redecode = 1 runs the reference's own O(L^2) schedule (every step decodes the whole prefix again) instead of the cache:
 * the engine switches to it by itself when a <PAD> token is emitted (the reference's padding mask is not causal); tests use the
 * flag to show that both schedules agree.
 * TableMasterDecoder (master_decoder.py:532-645, model="TableMaster" of ocr_table_structure_task.py:71) is the same decoder without the
 * cell-content layer: a blob that says "0 cell classes" (pdf_table_amd/weights.py::pack_mtl_decoder on a state_dict without cell tensors).
 * pt_tsr_mtl_structure then reports zero cells for every table and pt_tsr_mtl_cells(total = 0) is a no-op.  Its greedy_forward (:599-608)
 * never stops at <EOS> (the output is the last of max_seq_len + 1 passes; the convertor cuts at <EOS>): the KV-cached loop's stop at
 * <EOS> is that output's prefix, and a table still running when a <PAD> appears runs on to the length limit as the reference's does. */
int pt_tsr_mtl_decoder_config(pt_engine* e, int32_t out13[13]);
int pt_tsr_mtl_structure(pt_engine* e, const float* d_f3, int n, int hw, float* d_tag_logits, float* d_boxes, int32_t* h_lens,
                         int32_t* h_cell_counts, int force_redecode, pt_stream stream);
int pt_tsr_mtl_cells(pt_engine* e, int total, int32_t* d_cell_ids, float* d_cell_prob, float* d_cell_logits, int32_t* h_steps,
                     int force_redecode, pt_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* PDFTABLE_HIP_H */
