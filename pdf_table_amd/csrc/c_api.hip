// c_api.hip -- the entry points of libpdftable_hip.so (declared in include/pdftable_hip.h), once per activation format: this file is compiled
// into namespace pt_bf16::api and, with -DPT_ACT_F16=1, into pt_f16::api (act16.h).  The exported C symbols are the thin dispatchers of
// api_dispatch.cpp, which pdf_table_amd/build.py generates from the header's prototypes: an entry point whose first argument is the engine runs
// the namespace of pt_engine::precision (PT_PRECISION_F16 -> pt_f16), every other entry point (plans, sizes) runs pt_bf16's copy.
// pt_set_error / pt_last_error / pt_abi_version live in api_common.cpp (compiled once).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <cmath>

#include "common.h"

namespace PT_FMT_NS {
namespace api {

static int ensure(void** buf, size_t* cap, size_t need);     // grow an engine-owned device buffer (defined below)

int pt_engine_check(pt_engine* e) {
  PT_REQUIRE(e, "pt_engine_check: null engine");
  if (e->lstm_err && *e->lstm_err) {
    // reported ONCE: the flag is cleared and the engine switches to the streaming LSTM, so the caller can re-run the batch
    *e->lstm_err = 0;
    e->lstm_cluster = 0;
    pt_set_error("lstm_cluster_kernel: a workgroup waited > 2^22 polls for its peers -- the launch was not co-resident "
                 "(GPU shared with another process or stream?).  The recognition results since the last check are invalid; "
                 "this engine now uses the streaming LSTM kernel (pt_engine_set_lstm_cluster(e, 0)): run the batch again");
    return PT_ERR_HIP;
  }
  return PT_OK;
}

int pt_engine_set_lstm_cluster(pt_engine* e, int on) {
  PT_REQUIRE(e && (on == 0 || on == 1), "pt_engine_set_lstm_cluster: bad arguments");
  e->lstm_cluster = on;
  return PT_OK;
}

int pt_engine_set_dcn_mfma(pt_engine* e, int on) {
  PT_REQUIRE(e && on >= 0 && on <= 2, "pt_engine_set_dcn_mfma: bad arguments");
  e->dcn_mfma = on;
  return PT_OK;
}

int pt_engine_set_mtl_kv_fp8(pt_engine* e, int on) {
  PT_REQUIRE(e && (on == 0 || on == 1), "pt_engine_set_mtl_kv_fp8: bad arguments");
  e->mtl_kv_fp8 = on;
  return PT_OK;
}

int pt_engine_set_precision(pt_engine* e, int precision) {
  PT_REQUIRE(e && (precision == PT_PRECISION_BF16 || precision == PT_PRECISION_BF16X3 || precision == PT_PRECISION_F16X2 || precision == PT_PRECISION_F16),
             "pt_engine_set_precision: bad arguments");
  e->precision = precision;      // the dispatchers read it on every call: PT_PRECISION_F16 runs namespace pt_f16 from the next call on
  return PT_OK;
}

int pt_engine_create(int device_id, pt_engine** out) {
  PT_REQUIRE(out != nullptr, "pt_engine_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  PT_HIP_CHECK(hipGetDeviceCount(&ndev));
  PT_REQUIRE(device_id >= 0 && device_id < ndev, "pt_engine_create: device %d not in [0,%d)", device_id, ndev);
  PT_HIP_CHECK(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  PT_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
  pt_engine* e = new pt_engine();
  e->device = device_id;
  e->num_cu = prop.multiProcessorCount;
  {
    const char* ev = getenv("PT_LSTM_CLUSTER");     // default of pt_engine_set_lstm_cluster
    e->lstm_cluster = ev ? (atoi(ev) != 0) : 1;
    ev = getenv("PT_MTL_KV_FP8");                   // default of pt_engine_set_mtl_kv_fp8
    e->mtl_kv_fp8 = ev ? (atoi(ev) != 0) : 0;
    ev = getenv("PT_DCN_MFMA");                     // default of pt_engine_set_dcn_mfma
    e->dcn_mfma = ev ? (atoi(ev) < 0 || atoi(ev) > 2 ? 0 : atoi(ev)) : 0;
    ev = getenv("PT_REC_RAGGED");                   // 0: the recogniser's conv stack also computes the padding (A/B switch)
    e->rec_ragged = ev ? (atoi(ev) != 0) : 1;
  }
  *out = e;
  return PT_OK;
}

void pt_engine_destroy(pt_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipDeviceSynchronize();
  for (auto& kv : e->models)
    if (kv.second.d_blob) (void)hipFree(kv.second.d_blob);
  for (auto& a : e->arenas)
    if (a.base) (void)hipFree(a.base);
  if (e->rec_crops) (void)hipFree(e->rec_crops);
  if (e->rec_gray) (void)hipFree(e->rec_gray);
  if (e->rec_off) (void)hipFree(e->rec_off);
  if (e->zero_page) (void)hipFree(e->zero_page);
  if (e->tsr_scratch) (void)hipFree(e->tsr_scratch);
  if (e->tsr_lut) (void)hipFree(e->tsr_lut);
  if (e->cls_lut) (void)hipFree(e->cls_lut);
  if (e->rec_pp_lut) (void)hipFree(e->rec_pp_lut);
  if (e->rec_zero[0]) (void)hipFree(e->rec_zero[0]);
  if (e->rec_zero[1]) (void)hipFree(e->rec_zero[1]);
  for (int i = 0; i < 2; ++i)
    if (e->rec_zero_ready[i]) (void)hipEventDestroy(e->rec_zero_ready[i]);
  if (e->rec_limits) (void)hipFree(e->rec_limits);
  if (e->cls_scratch) (void)hipFree(e->cls_scratch);
  if (e->layout_scratch) (void)hipFree(e->layout_scratch);
  if (e->det_in) (void)hipFree(e->det_in);
  pt_mtl_release(e);
  e->stage_ring.destroy();
  if (e->lstm_scratch) (void)hipFree(e->lstm_scratch);
  if (e->lstm_scratch8) (void)hipFree(e->lstm_scratch8);
  if (e->lstm_err) (void)hipHostFree(e->lstm_err);
  if (e->prof.h_lims) (void)hipHostFree(e->prof.h_lims);
  for (auto& p : e->prof.pending) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  delete e;
}

// ---- weights ---------------------------------------------------------------------------------------
#pragma pack(push, 1)
struct BlobEntry {
  char name[96];
  uint32_t dtype, ndim;
  uint32_t dims[6];
  uint64_t offset, nbytes;
};
#pragma pack(pop)

static int register_blob(pt_engine* e, int kind, const uint8_t* h_head, size_t head_bytes, void* d_blob, size_t nbytes) {
  if (head_bytes < 8 || memcmp(h_head, "PTW1", 4) != 0) {
    pt_set_error("weight blob: bad magic");
    return PT_ERR_FORMAT;
  }
  uint32_t nt;
  memcpy(&nt, h_head + 4, 4);
  if (8 + (size_t)nt * sizeof(BlobEntry) > head_bytes) {
    pt_set_error("weight blob: truncated table (%u tensors)", nt);
    return PT_ERR_FORMAT;
  }
  PtModel m;
  m.d_blob = d_blob;
  m.nbytes = nbytes;
  for (uint32_t i = 0; i < nt; ++i) {
    BlobEntry be;
    memcpy(&be, h_head + 8 + (size_t)i * sizeof(BlobEntry), sizeof(be));
    be.name[95] = 0;
    if (be.offset + be.nbytes > nbytes || (be.offset & 255) || be.ndim > 6) {
      pt_set_error("weight blob: tensor '%s' out of range", be.name);
      return PT_ERR_FORMAT;
    }
    PtTensor t;
    t.dtype = (int)be.dtype;
    t.ndim = (int)be.ndim;
    for (int k = 0; k < 6; ++k) t.dims[k] = be.dims[k];
    t.nbytes = be.nbytes;
    t.d_ptr = reinterpret_cast<const char*>(d_blob) + be.offset;
    m.tensors[be.name] = t;
    if (strcmp(be.name, "__act_f16__") == 0) m.act_f16 = 1;       // weights.py fmt="f16": every 16-bit tensor of the blob holds IEEE-half bits
  }
  auto old = e->models.find(kind);
  if (old != e->models.end()) {
    (void)hipDeviceSynchronize();
    (void)hipFree(old->second.d_blob);
  }
  e->models[kind] = m;
  if (kind == PT_MODEL_DB_RESNET18 || kind == PT_MODEL_DB_NAS) e->det_kind = kind;   // the active detector
  if (kind == PT_MODEL_CRNN) e->rec_zero_valid[0] = e->rec_zero_valid[1] = false;     // cached all-padding-line activations
  return PT_OK;
}

int pt_weights_load(pt_engine* e, int model_kind, const void* h_blob, size_t nbytes) {
  PT_REQUIRE(e && h_blob && nbytes >= 8, "pt_weights_load: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  void* d = nullptr;
  PT_HIP_CHECK(hipMalloc(&d, nbytes));
  hipError_t he = hipMemcpy(d, h_blob, nbytes, hipMemcpyHostToDevice);
  if (he != hipSuccess) {
    (void)hipFree(d);
    pt_set_error("hipMemcpy(weights) failed: %s", hipGetErrorString(he));
    return PT_ERR_HIP;
  }
  uint32_t nt = 0;
  memcpy(&nt, reinterpret_cast<const uint8_t*>(h_blob) + 4, 4);
  size_t head = 8 + (size_t)nt * sizeof(BlobEntry);
  if (head > nbytes) head = nbytes;
  int rc = register_blob(e, model_kind, reinterpret_cast<const uint8_t*>(h_blob), head, d, nbytes);
  if (rc != PT_OK) (void)hipFree(d);
  return rc;
}

int pt_weights_load_device(pt_engine* e, int model_kind, const void* d_blob, size_t nbytes, pt_stream stream) {
  PT_REQUIRE(e && d_blob && nbytes >= 8, "pt_weights_load_device: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PT_HIP_CHECK(hipSetDevice(e->device));
  void* d = nullptr;
  PT_HIP_CHECK(hipMalloc(&d, nbytes));
  PT_HIP_CHECK(hipMemcpyAsync(d, d_blob, nbytes, hipMemcpyDeviceToDevice, s));
  uint8_t h8[8];
  PT_HIP_CHECK(hipMemcpyAsync(h8, d, 8, hipMemcpyDeviceToHost, s));
  PT_HIP_CHECK(hipStreamSynchronize(s));
  uint32_t nt = 0;
  memcpy(&nt, h8 + 4, 4);
  size_t head = 8 + (size_t)nt * sizeof(BlobEntry);
  if (memcmp(h8, "PTW1", 4) != 0 || head > nbytes) {
    (void)hipFree(d);
    pt_set_error("weight blob: bad magic or table");
    return PT_ERR_FORMAT;
  }
  std::vector<uint8_t> hh(head);
  PT_HIP_CHECK(hipMemcpy(hh.data(), d, head, hipMemcpyDeviceToHost));
  int rc = register_blob(e, model_kind, hh.data(), head, d, nbytes);
  if (rc != PT_OK) (void)hipFree(d);
  return rc;
}

// ---- detection ---------------------------------------------------------------------------------------
// the detector network loaded last: `DBModel` or `DBNasModel` (modeling_db_net.py:47-52)
static int det_net(pt_engine* e, const bf16_t* x, int n, int H, int W, float* prob, float* logits, hipStream_t s, uint32_t* bitmap = nullptr,
                   float thresh = 0.f, int* bitmap_done = nullptr) {
  if (bitmap_done) *bitmap_done = 0;
  if (e->det_kind == PT_MODEL_DB_NAS) return pt_dbnas_forward_net(e, x, n, H, W, prob, logits, s);
  return pt_db_forward_net(e, x, n, H, W, prob, logits, s, bitmap, thresh, bitmap_done);
}

int pt_det_plan(int h, int w, int pre_flavour, int* net_h, int* net_w) {
  PT_REQUIRE(h > 0 && w > 0 && net_h && net_w, "pt_det_plan: bad arguments");
  if (pre_flavour == PT_DET_PRE_DB_PP) {
    // DetResizeForTest.resize_image_type0, limit_type 'max', limit_side_len 960 (image_operators.py:269-316)
    const int limit = 960;
    double ratio = 1.0;
    if ((h > w ? h : w) > limit) ratio = (h > w) ? (double)limit / h : (double)limit / w;
    int rh = (int)(h * ratio), rw = (int)(w * ratio);
    // python round(): half to even
    rh = (int)std::nearbyint(rh / 32.0) * 32;
    rw = (int)std::nearbyint(rw / 32.0) * 32;
    *net_h = rh < 32 ? 32 : rh;
    *net_w = rw < 32 ? 32 : rw;
  } else if (pre_flavour == PT_DET_PRE_DB_TORCH) {
    // OCRDetectionPreprocessor.resize, short side 736 (processor_ocr_dbnet.py:50-60)
    const int side = 736;
    if (h < w) {
      *net_h = side;
      *net_w = (int)(std::ceil((double)side / h * w / 32.0) * 32);
    } else {
      *net_w = side;
      *net_h = (int)(std::ceil((double)side / w * h / 32.0) * 32);
    }
  } else if (pre_flavour == PT_DET_PRE_NONE) {
    PT_REQUIRE(h % 32 == 0 && w % 32 == 0, "PT_DET_PRE_NONE needs sizes that are multiples of 32");
    *net_h = h;
    *net_w = w;
  } else {
    pt_set_error("unknown pre-process flavour %d", pre_flavour);
    return PT_ERR_INVALID;
  }
  return PT_OK;
}

int pt_det_preprocess(pt_engine* e, const uint8_t* d_pages_rgb, int n, int h, int w, int pre_flavour,
                      uint16_t* d_out_bf16, pt_stream stream) {
  PT_REQUIRE(e && d_pages_rgb && d_out_bf16 && n > 0, "pt_det_preprocess: bad arguments");
  int nh, nw, rc;
  if ((rc = pt_det_plan(h, w, pre_flavour, &nh, &nw)) != PT_OK) return rc;
  return pt_launch_det_preprocess(d_pages_rgb, n, h, w, nh, nw, pre_flavour, pt_split(e), d_out_bf16,
                                  reinterpret_cast<hipStream_t>(stream));
}

int pt_det_forward_net(pt_engine* e, const uint16_t* d_input_bf16, int n, int net_h, int net_w, float* d_prob,
                       float* d_logits, pt_stream stream) {
  PT_REQUIRE(e && d_input_bf16 && n > 0 && (d_prob || d_logits), "pt_det_forward_net: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return det_net(e, d_input_bf16, n, net_h, net_w, d_prob, d_logits, reinterpret_cast<hipStream_t>(stream));
}

// ---- layout (PicoDet) ---------------------------------------------------------------------------------
int pt_layout_plan(int inp_h, int inp_w, int32_t fm_h[PT_LAYOUT_LEVELS], int32_t fm_w[PT_LAYOUT_LEVELS]) {
  PT_REQUIRE(inp_h > 0 && inp_w > 0 && inp_h % 32 == 0 && inp_w % 32 == 0 && fm_h && fm_w,
             "pt_layout_plan: input %dx%d must be positive multiples of 32", inp_h, inp_w);
  for (int l = 0; l < 3; ++l) { fm_h[l] = inp_h >> (3 + l); fm_w[l] = inp_w >> (3 + l); }
  fm_h[3] = (fm_h[2] + 4 - 5) / 2 + 1;      // 5x5, stride 2, padding 2
  fm_w[3] = (fm_w[2] + 4 - 5) / 2 + 1;
  return PT_OK;
}

int pt_layout_preprocess(pt_engine* e, const uint8_t* d_pages_rgb, int n, int h, int w, int inp_h, int inp_w,
                         uint16_t* d_out_bf16, pt_stream stream) {
  PT_REQUIRE(e && d_pages_rgb && d_out_bf16 && n > 0, "pt_layout_preprocess: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  // same arithmetic as the PP-OCR detection pre-process (BGR flip, (x * 1/255 - mean) / std), fixed target size
  return pt_launch_det_preprocess(d_pages_rgb, n, h, w, inp_h, inp_w, PT_DET_PRE_DB_PP, pt_split(e),
                                  d_out_bf16, reinterpret_cast<hipStream_t>(stream));
}

int pt_layout_forward_net(pt_engine* e, const uint16_t* d_input_bf16, int n, int inp_h, int inp_w, float* d_head0,
                          float* d_head1, float* d_head2, float* d_head3, pt_stream stream) {
  PT_REQUIRE(e && d_input_bf16 && n > 0, "pt_layout_forward_net: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return pt_picodet_forward_net(e, d_input_bf16, n, inp_h, inp_w, d_head0, d_head1, d_head2, d_head3,
                                reinterpret_cast<hipStream_t>(stream));
}

int pt_layout_candidates(pt_engine* e, const float* d_head0, const float* d_head1, const float* d_head2,
                         const float* d_head3, int n, int inp_h, int inp_w, int num_classes, float thr_lo, int max_cands,
                         int32_t* d_counts, float* d_cands, pt_stream stream) {
  PT_REQUIRE(e && d_head0 && d_head1 && d_head2 && d_head3 && d_counts && d_cands && n > 0 && max_cands > 0 &&
                 num_classes > 0 && num_classes <= 8, "pt_layout_candidates: bad arguments");
  int32_t fh[4], fw[4];
  int rc = pt_layout_plan(inp_h, inp_w, fh, fw);
  if (rc != PT_OK) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PT_HIP_CHECK(hipMemsetAsync(d_counts, 0, (size_t)n * 4, s));
  const float* heads[4] = {d_head0, d_head1, d_head2, d_head3};
  for (int l = 0; l < 4; ++l)
    if ((rc = pt_launch_pico_candidates(heads[l], n, fh[l] * fw[l], num_classes, l, thr_lo, max_cands, d_cands, d_counts, s)) != PT_OK)
      return rc;
  return PT_OK;
}

int pt_layout_forward(pt_engine* e, const uint8_t* d_pages_rgb, int n, int h, int w, int inp_h, int inp_w, int num_classes,
                      float thr_lo, int max_cands, int32_t* d_counts, float* d_cands, pt_stream stream) {
  PT_REQUIRE(e && d_pages_rgb && n > 0, "pt_layout_forward: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  int32_t fh[4], fw[4];
  int rc = pt_layout_plan(inp_h, inp_w, fh, fw);
  if (rc != PT_OK) return rc;
  const int m = pt_split(e) ? 2 : 1;
  size_t off = 0, o_head[4];
  auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~size_t(255); return o; };
  const size_t o_x = carve((size_t)n * inp_h * inp_w * 4 * m * sizeof(uint16_t));
  for (int l = 0; l < 4; ++l) o_head[l] = carve((size_t)n * fh[l] * fw[l] * PT_LAYOUT_HEAD_CS * sizeof(float));
  if (off > e->layout_scratch_cap) {
    PT_HIP_CHECK(hipDeviceSynchronize());
    if (e->layout_scratch) PT_HIP_CHECK(hipFree(e->layout_scratch));
    e->layout_scratch = nullptr; e->layout_scratch_cap = 0;
    PT_HIP_CHECK(hipMalloc(&e->layout_scratch, off));
    e->layout_scratch_cap = off;
  }
  char* base = reinterpret_cast<char*>(e->layout_scratch);
  uint16_t* x = reinterpret_cast<uint16_t*>(base + o_x);
  float* hd[4];
  for (int l = 0; l < 4; ++l) hd[l] = reinterpret_cast<float*>(base + o_head[l]);
  if ((rc = api::pt_layout_preprocess(e, d_pages_rgb, n, h, w, inp_h, inp_w, x, stream)) != PT_OK) return rc;
  if ((rc = api::pt_layout_forward_net(e, x, n, inp_h, inp_w, hd[0], hd[1], hd[2], hd[3], stream)) != PT_OK)
    return rc;
  return api::pt_layout_candidates(e, hd[0], hd[1], hd[2], hd[3], n, inp_h, inp_w, num_classes, thr_lo, max_cands, d_counts, d_cands,
                              stream);
}

int pt_tsr_preprocess(pt_engine* e, const uint8_t* d_pages_rgb, int n_pages, int ph, int pw, const pt_tsr_table* d_tables,
                      int n, int inp_h, int inp_w, int bgr, uint16_t* d_out_bf16, pt_stream stream) {
  PT_REQUIRE(e && d_pages_rgb && d_tables && d_out_bf16 && n > 0 && n_pages > 0, "pt_tsr_preprocess: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  if (!e->tsr_lut) {
    // ((x / 255.) - mean) / std in float64 then cast, as numpy evaluates processer_lore.py:67-70,90
    const float mean[3] = {0.408f, 0.447f, 0.470f}, stdv[3] = {0.289f, 0.274f, 0.278f};
    float lut[768];
    for (int c = 0; c < 3; ++c)
      for (int v = 0; v < 256; ++v) lut[c * 256 + v] = (float)(((double)v / 255.0 - (double)mean[c]) / (double)stdv[c]);
    PT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->tsr_lut), sizeof(lut)));
    PT_HIP_CHECK(hipMemcpy(e->tsr_lut, lut, sizeof(lut), hipMemcpyHostToDevice));
  }
  return pt_launch_tsr_preprocess(d_pages_rgb, ph, pw, d_tables, n, inp_h, inp_w, bgr, e->tsr_lut, d_out_bf16,
                                  pt_split(e), reinterpret_cast<hipStream_t>(stream));
}

int pt_tsr_forward_net(pt_engine* e, const uint16_t* d_input_bf16, int n, int H, int W, float* d_hm, float* d_st,
                       float* d_wh, float* d_ax, float* d_cr, float* d_reg, pt_stream stream) {
  PT_REQUIRE(e && d_input_bf16 && n > 0, "pt_tsr_forward_net: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return pt_lore_forward_net(e, d_input_bf16, n, H, W, d_hm, d_st, d_wh, d_ax, d_cr, d_reg,
                             reinterpret_cast<hipStream_t>(stream));
}

int pt_tsr_forward_net_wireless(pt_engine* e, const uint16_t* d_input_bf16, int n, int H, int W, float* d_hm, float* d_st,
                                float* d_wh, float* d_ax, float* d_cr, float* d_reg, pt_stream stream) {
  PT_REQUIRE(e && d_input_bf16 && n > 0, "pt_tsr_forward_net_wireless: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return pt_lore_wireless_forward_net(e, d_input_bf16, n, H, W, d_hm, d_st, d_wh, d_ax, d_cr, d_reg,
                                      reinterpret_cast<hipStream_t>(stream));
}

int pt_tsr_decode(pt_engine* e, const float* d_hm, const float* d_st, const float* d_wh, const float* d_ax,
                  const float* d_cr, const float* d_reg, int n, int h, int w, int wiz_rev, float vis_thresh,
                  int32_t* d_counts, float* d_dets, float* d_logi, pt_stream stream) {
  PT_REQUIRE(e && n > 0 && h > 0 && w > 0, "pt_tsr_decode: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return pt_lore_decode(e, d_hm, d_st, d_wh, d_ax, d_cr, d_reg, n, h, w, wiz_rev, vis_thresh, d_counts, d_dets, d_logi,
                        reinterpret_cast<hipStream_t>(stream));
}

int pt_tsr_forward_decode(pt_engine* e, const uint16_t* d_input_bf16, int n, int in_h, int in_w, int wiz_rev, float vis_thresh,
                          int32_t* d_counts, float* d_dets, float* d_logi, pt_stream stream) {
  PT_REQUIRE(e && d_input_bf16 && n > 0, "pt_tsr_forward_decode: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return pt_lore_forward_decode(e, d_input_bf16, n, in_h, in_w, wiz_rev, vis_thresh, d_counts, d_dets, d_logi,
                                reinterpret_cast<hipStream_t>(stream));
}

int pt_tsr_process(pt_engine* e, const float* d_logi, const float* d_dets, const int32_t* h_counts, int n_tables,
                   int use_2dpe, float* d_logic, float* d_stacked, pt_stream stream) {
  PT_REQUIRE(e, "pt_tsr_process: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return pt_lore_process(e, d_logi, d_dets, h_counts, n_tables, use_2dpe, d_logic, d_stacked,
                         reinterpret_cast<hipStream_t>(stream));
}

int pt_det_bitmap(pt_engine* e, const float* d_prob, int n, int net_h, int net_w, float thresh, int use_dilation,
                  uint32_t* d_bitmap, pt_stream stream) {
  PT_REQUIRE(e && d_prob && d_bitmap && n > 0, "pt_det_bitmap: bad arguments");
  return pt_launch_bitmap(d_prob, n, net_h, net_w, thresh, use_dilation, d_bitmap, reinterpret_cast<hipStream_t>(stream));
}

static int microbatch() {
  static int mb = -1;
  if (mb < 0) {
    const char* s = getenv("PT_DET_MICROBATCH");
    // 64 pages per launch: the 30x30 / 60x60 layers of an 8-page batch are only 256-1000 workgroups, i.e. one (partial)
    // round on 256 CUs x 2; measured det-only 3314 -> 3996 pages/s (8 -> 32), 4096 at 64; four stages 513 -> 519 pages/s
    // (32 -> 64, with 128-table Lore micro-batches); ~190 MB of activations per page = 12 GB of the 288 GB at 64
    mb = s ? atoi(s) : 64;
    if (mb < 1) mb = 1;
  }
  return mb;
}

int pt_det_forward(pt_engine* e, const uint8_t* d_pages_rgb, int n, int h, int w, int pre_flavour, float thresh,
                   int use_dilation, float* d_prob, uint32_t* d_bitmap, pt_stream stream) {
  PT_REQUIRE(e && d_pages_rgb && n > 0 && d_prob, "pt_det_forward: bad arguments (d_prob is required)");
  PT_HIP_CHECK(hipSetDevice(e->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int nh, nw, rc;
  if ((rc = pt_det_plan(h, w, pre_flavour, &nh, &nw)) != PT_OK) return rc;
  const int mb = microbatch();
  // the pre-processed pages live in their own engine-owned buffer (not the arena, which the net resets)
  const int x3 = pt_split(e);
  const size_t need = (size_t)(mb < n ? mb : n) * nh * nw * (x3 ? 8 : 4) * sizeof(bf16_t);
  if ((rc = ensure(&e->det_in, &e->det_in_cap, need)) != PT_OK) return rc;
  void* xbuf = e->det_in;
  for (int i0 = 0; i0 < n; i0 += mb) {
    const int nb = (n - i0) < mb ? (n - i0) : mb;
    {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "preprocess");
      rc = pt_launch_det_preprocess(d_pages_rgb + (size_t)i0 * h * w * 3, nb, h, w, nh, nw, pre_flavour, x3,
                                    reinterpret_cast<bf16_t*>(xbuf), s);
      if (rc != PT_OK) return rc;
    }
    float* prob_i = d_prob + (size_t)i0 * nh * nw;
    // without dilation the DB-ResNet18 head thresholds its own output (bf16 mode): no second pass over the probability map
    int bm_done = 0;
    uint32_t* bm_i = d_bitmap ? d_bitmap + (size_t)i0 * nh * (nw / 32) : nullptr;
    rc = det_net(e, reinterpret_cast<const bf16_t*>(xbuf), nb, nh, nw, prob_i, nullptr, s, use_dilation ? nullptr : bm_i, thresh, &bm_done);
    if (rc != PT_OK) return rc;
    if (d_bitmap && !bm_done) {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "bitmap");
      rc = pt_launch_bitmap(prob_i, nb, nh, nw, thresh, use_dilation, d_bitmap + (size_t)i0 * nh * (nw / 32), s);
      if (rc != PT_OK) return rc;
    }
  }
  return PT_OK;
}

int pt_det_box_scores(pt_engine* e, const float* d_prob, int n, int net_h, int net_w, const float* d_boxes, int nb,
                      float* d_scores, pt_stream stream) {
  PT_REQUIRE(e && d_prob && (nb == 0 || (d_boxes && d_scores)), "pt_det_box_scores: bad arguments");
  return pt_launch_box_scores(d_prob, n, net_h, net_w, d_boxes, nb, d_scores, reinterpret_cast<hipStream_t>(stream));
}

// ---- recognition ---------------------------------------------------------------------------------------
static int ensure(void** buf, size_t* cap, size_t need) {
  if (need <= *cap) return PT_OK;
  PT_HIP_CHECK(hipDeviceSynchronize());
  if (*buf) PT_HIP_CHECK(hipFree(*buf));
  *buf = nullptr;
  *cap = 0;
  PT_HIP_CHECK(hipMalloc(buf, need));
  *cap = need;
  return PT_OK;
}

static int rec_microbatch() {
  static int mb = -1;
  if (mb < 0) {
    const char* s = getenv("PT_REC_MICROBATCH");
    mb = s ? atoi(s) : 6144;  // many lines per launch; 6144 = one launch of the cluster LSTM at 192 lines per cluster on 256 CUs
    if (mb < 1) mb = 1;
  }
  return mb;
}

// crop + resize + gray for lines [i0, i0+nb) into d_gray (bf16 [nb,32,640] or hi/lo pairs)
static int rec_pre_chunk(pt_engine* e, const uint8_t* d_pages, int n_pages, int h, int w, const pt_rec_line* d_lines,
                         const int64_t* h_crop_px, int i0, int nb, bf16_t* d_gray, hipStream_t s) {
  (void)n_pages;
  long long maxpx = 0, total = 0;
  for (int i = 0; i < nb; ++i) {
    const long long px = h_crop_px[i0 + i] > 0 ? h_crop_px[i0 + i] : 0;
    total += px;
    if (px > maxpx) maxpx = px;
  }
  int rc;
  if ((rc = ensure(&e->rec_off, &e->rec_off_cap, (size_t)(nb + 1) * sizeof(long long))) != PT_OK) return rc;
  if ((rc = ensure(&e->rec_crops, &e->rec_crops_cap, (size_t)total * 3 + 16)) != PT_OK) return rc;
  // crop start offsets by a device scan over the line records: nothing is staged on the host, so the enqueue does not
  // have to wait for the stream (a host staging vector would have to outlive the asynchronous copy)
  if ((rc = pt_launch_rec_offsets(d_lines + i0, nb, reinterpret_cast<long long*>(e->rec_off), s)) != PT_OK) return rc;
  const long long* d_off = reinterpret_cast<const long long*>(e->rec_off);
  {
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "rec warp");
    rc = pt_launch_rec_warp(d_pages, h, w, d_lines + i0, nb, d_off, reinterpret_cast<uint8_t*>(e->rec_crops), (int)maxpx, s);
    if (rc != PT_OK) return rc;
  }
  PtProfScope ps(e, s, PT_PROF_OTHER, 0, "rec resize+gray");
  return pt_launch_rec_resize_gray(reinterpret_cast<const uint8_t*>(e->rec_crops), d_lines + i0, d_off, nb,
                                   pt_split(e), d_gray, s);
}

int pt_rec_preprocess(pt_engine* e, const uint8_t* d_pages_rgb, int n_pages, int h, int w, const pt_rec_line* d_lines,
                      const int64_t* h_crop_px, int n_lines, uint16_t* d_gray, pt_stream stream) {
  PT_REQUIRE(e && d_pages_rgb && d_lines && h_crop_px && d_gray && n_lines > 0, "pt_rec_preprocess: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return rec_pre_chunk(e, d_pages_rgb, n_pages, h, w, d_lines, h_crop_px, 0, n_lines, d_gray, reinterpret_cast<hipStream_t>(stream));
}

// crop offsets of a recognition call -> e->rec_off through a pinned slot of the staging ring: no stream synchronise in the enqueue path
// (a pipelined caller has the next batch's detection queued on s by now; waiting for it here stalled the host one detection per batch)
static int upload_crop_offsets(pt_engine* e, const std::vector<long long>& off, hipStream_t s) {
  const size_t nb = off.size() * sizeof(long long);
  void* hs = e->stage_ring.acquire(nb);
  PT_REQUIRE(hs, "recognition: pinned staging (%s)", hipGetErrorString(hipGetLastError()));
  memcpy(hs, off.data(), nb);
  PT_HIP_CHECK(hipMemcpyAsync(e->rec_off, hs, nb, hipMemcpyHostToDevice, s));
  PT_REQUIRE(e->stage_ring.release(s) == 0, "recognition: event record failed");
  return PT_OK;
}

int pt_rec_forward_net(pt_engine* e, const uint16_t* d_gray, int n, int32_t* d_ids, float* d_maxlogit, pt_stream stream) {
  PT_REQUIRE(e && d_gray && d_ids && n > 0, "pt_rec_forward_net: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int mb = rec_microbatch();
  const size_t per_line = (size_t)PT_REC_H * PT_REC_W * (pt_split(e) ? 2 : 1);
  for (int i0 = 0; i0 < n; i0 += mb) {
    const int nb = (n - i0) < mb ? (n - i0) : mb;
    int rc = pt_crnn_forward_net(e, d_gray + (size_t)i0 * per_line, nb, d_ids + (size_t)i0 * PT_REC_T,
                                 d_maxlogit ? d_maxlogit + (size_t)i0 * PT_REC_T : nullptr, s);
    if (rc != PT_OK) return rc;
  }
  return PT_OK;
}

int pt_rec_forward_crops(pt_engine* e, const uint8_t* d_crops_rgb, const pt_rec_line* d_lines, const int64_t* h_crop_px,
                         int n_lines, int32_t* d_ids, float* d_maxlogit, pt_stream stream) {
  PT_REQUIRE(e && d_crops_rgb && d_lines && h_crop_px && d_ids && n_lines > 0, "pt_rec_forward_crops: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int mb = rec_microbatch();
  const int x3 = pt_split(e);
  const size_t per_line = (size_t)PT_REC_H * PT_REC_W * (x3 ? 2 : 1);
  int rc;
  if ((rc = ensure(&e->rec_gray, &e->rec_gray_cap, (size_t)(mb < n_lines ? mb : n_lines) * per_line * sizeof(bf16_t))) != PT_OK)
    return rc;
  std::vector<long long>& off = e->rec_off_host;
  off.assign((size_t)n_lines + 1, 0);
  for (int i = 0; i < n_lines; ++i) off[i + 1] = off[i] + (h_crop_px[i] > 0 ? h_crop_px[i] : 0);
  if ((rc = ensure(&e->rec_off, &e->rec_off_cap, (size_t)(n_lines + 1) * sizeof(long long))) != PT_OK) return rc;
  if ((rc = upload_crop_offsets(e, off, s)) != PT_OK) return rc;
  const long long* d_off = reinterpret_cast<const long long*>(e->rec_off);
  for (int i0 = 0; i0 < n_lines; i0 += mb) {
    const int nb = (n_lines - i0) < mb ? (n_lines - i0) : mb;
    rc = pt_launch_rec_resize_gray(d_crops_rgb, d_lines + i0, d_off + i0, nb, x3, reinterpret_cast<bf16_t*>(e->rec_gray), s);
    if (rc != PT_OK) return rc;
    rc = pt_crnn_forward_net(e, reinterpret_cast<const bf16_t*>(e->rec_gray), nb, d_ids + (size_t)i0 * PT_REC_T,
                             d_maxlogit ? d_maxlogit + (size_t)i0 * PT_REC_T : nullptr, s, d_lines + i0);
    if (rc != PT_OK) return rc;
  }
  return PT_OK;
}

int pt_rec_forward(pt_engine* e, const uint8_t* d_pages_rgb, int n_pages, int h, int w, const pt_rec_line* d_lines,
                   const int64_t* h_crop_px, int n_lines, int32_t* d_ids, float* d_maxlogit, pt_stream stream) {
  PT_REQUIRE(e && d_pages_rgb && d_lines && h_crop_px && d_ids && n_lines > 0, "pt_rec_forward: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int mb = rec_microbatch();
  const size_t per_line = (size_t)PT_REC_H * PT_REC_W * (pt_split(e) ? 2 : 1);
  int rc;
  if ((rc = ensure(&e->rec_gray, &e->rec_gray_cap, (size_t)(mb < n_lines ? mb : n_lines) * per_line * sizeof(bf16_t))) != PT_OK)
    return rc;
  for (int i0 = 0; i0 < n_lines; i0 += mb) {
    const int nb = (n_lines - i0) < mb ? (n_lines - i0) : mb;
    rc = rec_pre_chunk(e, d_pages_rgb, n_pages, h, w, d_lines, h_crop_px, i0, nb, reinterpret_cast<bf16_t*>(e->rec_gray), s);
    if (rc != PT_OK) return rc;
    rc = pt_crnn_forward_net(e, reinterpret_cast<const bf16_t*>(e->rec_gray), nb, d_ids + (size_t)i0 * PT_REC_T,
                             d_maxlogit ? d_maxlogit + (size_t)i0 * PT_REC_T : nullptr, s, d_lines + i0);
    if (rc != PT_OK) return rc;
  }
  return PT_OK;
}

// ---- ConvNextViT recogniser ---------------------------------------------------------------------------------------
static int cvit_microbatch() {
  static int mb = -1;
  if (mb < 0) {
    const char* s = getenv("PT_CVIT_MICROBATCH");
    mb = s ? atoi(s) : 512;
    if (mb < 1) mb = 1;
  }
  return mb;
}

int pt_rec_cvit_forward_net(pt_engine* e, const float* d_gray, int layout, int n_lines, const int32_t* h_text_w, int32_t* d_ids,
                            float* d_maxlogit, pt_stream stream) {
  PT_REQUIRE(e && d_gray && d_ids && n_lines > 0, "pt_rec_cvit_forward_net: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return pt_cvit_forward_net(e, d_gray, layout, n_lines, d_ids, d_maxlogit, reinterpret_cast<hipStream_t>(stream), h_text_w);
}

// text width of every line after OCRRecognitionPreprocessor.keepratio_resize (processor_ocr_recognition.py:44-53), the formula
// of rec_resize_gray_kernel in the same IEEE double arithmetic
static void cvit_text_widths(const int32_t* h_crop_wh, int i0, int n, std::vector<int>& tw) {
  tw.resize((size_t)n);
  for (int i = 0; i < n; ++i) {
    const int cw = h_crop_wh[2 * (i0 + i)], ch = h_crop_wh[2 * (i0 + i) + 1];
    int nw = 0;
    if (cw > 0 && ch > 0) {
      const double ratio = (double)cw / (double)ch;
      nw = ratio > (double)PT_CVIT_W / (double)PT_REC_H ? PT_CVIT_W : (int)((double)PT_REC_H * ratio);
    }
    tw[(size_t)i] = nw;
  }
}

// crop offsets of already-cropped lines -> device (host prefix sum; the copy is waited for: `off` is reused by the next call)
static int cvit_crop_offsets(pt_engine* e, const int64_t* h_crop_px, int n_lines, hipStream_t s) {
  std::vector<long long>& off = e->rec_off_host;
  off.assign((size_t)n_lines + 1, 0);
  for (int i = 0; i < n_lines; ++i) off[i + 1] = off[i] + (h_crop_px[i] > 0 ? h_crop_px[i] : 0);
  int rc;
  if ((rc = ensure(&e->rec_off, &e->rec_off_cap, (size_t)(n_lines + 1) * sizeof(long long))) != PT_OK) return rc;
  return upload_crop_offsets(e, off, s);
}

int pt_rec_cvit_preprocess_crops(pt_engine* e, const uint8_t* d_crops_rgb, const pt_rec_line* d_lines, const int64_t* h_crop_px,
                                 int n_lines, float* d_gray, pt_stream stream) {
  PT_REQUIRE(e && d_crops_rgb && d_lines && h_crop_px && d_gray && n_lines > 0, "pt_rec_cvit_preprocess_crops: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc;
  if ((rc = cvit_crop_offsets(e, h_crop_px, n_lines, s)) != PT_OK) return rc;
  return pt_launch_rec_resize_gray_f32(d_crops_rgb, d_lines, reinterpret_cast<const long long*>(e->rec_off), n_lines, PT_CVIT_W, d_gray, s);
}

int pt_rec_cvit_forward_crops(pt_engine* e, const uint8_t* d_crops_rgb, const pt_rec_line* d_lines, const int64_t* h_crop_px,
                              const int32_t* h_crop_wh, int n_lines, int32_t* d_ids, float* d_maxlogit, pt_stream stream) {
  PT_REQUIRE(e && d_crops_rgb && d_lines && h_crop_px && d_ids && n_lines > 0, "pt_rec_cvit_forward_crops: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int mb = cvit_microbatch();
  const size_t per_line = (size_t)PT_REC_H * PT_CVIT_W * sizeof(float);
  int rc;
  if ((rc = ensure(&e->rec_gray, &e->rec_gray_cap, (size_t)(mb < n_lines ? mb : n_lines) * per_line)) != PT_OK) return rc;
  if ((rc = cvit_crop_offsets(e, h_crop_px, n_lines, s)) != PT_OK) return rc;
  const long long* d_off = reinterpret_cast<const long long*>(e->rec_off);
  for (int i0 = 0; i0 < n_lines; i0 += mb) {
    const int nb = (n_lines - i0) < mb ? (n_lines - i0) : mb;
    rc = pt_launch_rec_resize_gray_f32(d_crops_rgb, d_lines + i0, d_off + i0, nb, PT_CVIT_W, reinterpret_cast<float*>(e->rec_gray), s);
    if (rc != PT_OK) return rc;
    std::vector<int> tw;
    if (h_crop_wh) cvit_text_widths(h_crop_wh, i0, nb, tw);
    rc = pt_cvit_forward_net(e, reinterpret_cast<const float*>(e->rec_gray), 1, nb, d_ids + (size_t)i0 * PT_CVIT_T,
                             d_maxlogit ? d_maxlogit + (size_t)i0 * PT_CVIT_T : nullptr, s, h_crop_wh ? tw.data() : nullptr);
    if (rc != PT_OK) return rc;
  }
  return PT_OK;
}

int pt_rec_cvit_forward(pt_engine* e, const uint8_t* d_pages_rgb, int n_pages, int h, int w, const pt_rec_line* d_lines,
                        const int64_t* h_crop_px, const int32_t* h_crop_wh, int n_lines, int32_t* d_ids, float* d_maxlogit,
                        pt_stream stream) {
  PT_REQUIRE(e && d_pages_rgb && d_lines && h_crop_px && d_ids && n_lines > 0, "pt_rec_cvit_forward: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  (void)n_pages;
  const int mb = cvit_microbatch();
  const size_t per_line = (size_t)PT_REC_H * PT_CVIT_W * sizeof(float);
  int rc;
  if ((rc = ensure(&e->rec_gray, &e->rec_gray_cap, (size_t)(mb < n_lines ? mb : n_lines) * per_line)) != PT_OK) return rc;
  for (int i0 = 0; i0 < n_lines; i0 += mb) {
    const int nb = (n_lines - i0) < mb ? (n_lines - i0) : mb;
    long long maxpx = 0, total = 0;
    for (int i = 0; i < nb; ++i) {
      const long long px = h_crop_px[i0 + i] > 0 ? h_crop_px[i0 + i] : 0;
      total += px;
      if (px > maxpx) maxpx = px;
    }
    if ((rc = ensure(&e->rec_off, &e->rec_off_cap, (size_t)(nb + 1) * sizeof(long long))) != PT_OK) return rc;
    if ((rc = ensure(&e->rec_crops, &e->rec_crops_cap, (size_t)total * 3 + 16)) != PT_OK) return rc;
    if ((rc = pt_launch_rec_offsets(d_lines + i0, nb, reinterpret_cast<long long*>(e->rec_off), s)) != PT_OK) return rc;
    const long long* d_off = reinterpret_cast<const long long*>(e->rec_off);
    if ((rc = pt_launch_rec_warp(d_pages_rgb, h, w, d_lines + i0, nb, d_off, reinterpret_cast<uint8_t*>(e->rec_crops), (int)maxpx, s)) != PT_OK)
      return rc;
    rc = pt_launch_rec_resize_gray_f32(reinterpret_cast<const uint8_t*>(e->rec_crops), d_lines + i0, d_off, nb, PT_CVIT_W,
                                       reinterpret_cast<float*>(e->rec_gray), s);
    if (rc != PT_OK) return rc;
    std::vector<int> tw;
    if (h_crop_wh) cvit_text_widths(h_crop_wh, i0, nb, tw);
    rc = pt_cvit_forward_net(e, reinterpret_cast<const float*>(e->rec_gray), 1, nb, d_ids + (size_t)i0 * PT_CVIT_T,
                             d_maxlogit ? d_maxlogit + (size_t)i0 * PT_CVIT_T : nullptr, s, h_crop_wh ? tw.data() : nullptr);
    if (rc != PT_OK) return rc;
  }
  return PT_OK;
}

// ---- MtlTabNet backbone ------------------------------------------------------------------------------------------
int pt_tsr_mtl_backbone_net(pt_engine* e, const uint16_t* d_x, int n, int H, int W, float* d_f3, pt_stream stream) {
  PT_REQUIRE(e && d_x && d_f3 && n > 0, "pt_tsr_mtl_backbone_net: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return pt_mtl_backbone_forward_net(e, d_x, n, H, W, d_f3, reinterpret_cast<hipStream_t>(stream));
}

// ---- MtlTabNet pre-processing + decoders (mtl_decoder.hip) ---------------------------------------------------------
int pt_tsr_mtl_preprocess(pt_engine* e, const uint8_t* d_pages, int n_pages, int h, int w, const pt_tsr_table* d_tables, int n, int size,
                          uint16_t* d_out, pt_stream stream) {
  PT_REQUIRE(e && d_pages && d_tables && d_out && n_pages > 0 && n > 0 && h > 0 && w > 0 && size > 0 && size % 8 == 0, "pt_tsr_mtl_preprocess: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return pt_mtl_preprocess(e, d_pages, h, w, d_tables, n, size, d_out, reinterpret_cast<hipStream_t>(stream));
}

int pt_tsr_mtl_decoder_config(pt_engine* e, int32_t out13[13]) {
  PT_REQUIRE(e && out13, "pt_tsr_mtl_decoder_config: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return pt_mtl_decoder_config(e, out13);
}

int pt_tsr_mtl_structure(pt_engine* e, const float* d_f3, int n, int hw, float* d_tag_logits, float* d_boxes, int32_t* h_lens,
                         int32_t* h_cell_counts, int force_redecode, pt_stream stream) {
  PT_REQUIRE(e, "pt_tsr_mtl_structure: null engine");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return pt_mtl_structure(e, d_f3, n, hw, d_tag_logits, d_boxes, h_lens, h_cell_counts, force_redecode, reinterpret_cast<hipStream_t>(stream));
}

int pt_tsr_mtl_cells(pt_engine* e, int total, int32_t* d_cell_ids, float* d_cell_prob, float* d_cell_logits, int32_t* h_steps,
                     int force_redecode, pt_stream stream) {
  PT_REQUIRE(e, "pt_tsr_mtl_cells: null engine");
  PT_HIP_CHECK(hipSetDevice(e->device));
  return pt_mtl_cells(e, total, d_cell_ids, d_cell_prob, d_cell_logits, h_steps, force_redecode, reinterpret_cast<hipStream_t>(stream));
}

// ---- PP-OCR recognition pre-processor --------------------------------------------------------------------------
static int rec_pp_lut(pt_engine* e) {
  if (e->rec_pp_lut) return PT_OK;
  float lut[256];
  for (int v = 0; v < 256; ++v) {      // resized.astype('float32') / 255; -= 0.5; /= 0.5  (fp32 at every step)
    float x = (float)v / 255.0f;
    x -= 0.5f;
    x /= 0.5f;
    lut[v] = x;
  }
  PT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->rec_pp_lut), sizeof(lut)));
  PT_HIP_CHECK(hipMemcpy(e->rec_pp_lut, lut, sizeof(lut), hipMemcpyHostToDevice));
  return PT_OK;
}

int pt_rec_pp_preprocess(pt_engine* e, const uint8_t* d_pages_rgb, int n_pages, int h, int w, const pt_rec_line* d_lines,
                         const int64_t* h_crop_px, int n_lines, const pt_rec_pp_item* d_items, int n_items, int img_h,
                         int max_img_w, float* d_out, pt_stream stream) {
  PT_REQUIRE(e && d_pages_rgb && d_lines && h_crop_px && d_items && d_out && n_lines > 0 && n_items > 0 && img_h > 0 && max_img_w > 0,
             "pt_rec_pp_preprocess: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  (void)n_pages;
  long long maxpx = 0, total = 0;
  for (int i = 0; i < n_lines; ++i) {
    const long long px = h_crop_px[i] > 0 ? h_crop_px[i] : 0;
    total += px;
    if (px > maxpx) maxpx = px;
  }
  int rc;
  if ((rc = rec_pp_lut(e)) != PT_OK) return rc;
  if ((rc = ensure(&e->rec_off, &e->rec_off_cap, (size_t)(n_lines + 1) * sizeof(long long))) != PT_OK) return rc;
  if ((rc = ensure(&e->rec_crops, &e->rec_crops_cap, (size_t)total * 3 + 16)) != PT_OK) return rc;
  if ((rc = pt_launch_rec_offsets(d_lines, n_lines, reinterpret_cast<long long*>(e->rec_off), s)) != PT_OK) return rc;
  const long long* d_off = reinterpret_cast<const long long*>(e->rec_off);
  if ((rc = pt_launch_rec_warp(d_pages_rgb, h, w, d_lines, n_lines, d_off, reinterpret_cast<uint8_t*>(e->rec_crops), (int)maxpx, s)) != PT_OK)
    return rc;
  return pt_launch_rec_pp_resize_norm(reinterpret_cast<const uint8_t*>(e->rec_crops), d_lines, d_off, d_items, n_items, img_h, max_img_w,
                                      e->rec_pp_lut, d_out, s);
}

int pt_rec_pp_preprocess_crops(pt_engine* e, const uint8_t* d_crops_rgb, const pt_rec_line* d_lines, const int64_t* h_crop_px,
                               int n_lines, const pt_rec_pp_item* d_items, int n_items, int img_h, int max_img_w, float* d_out,
                               pt_stream stream) {
  PT_REQUIRE(e && d_crops_rgb && d_lines && h_crop_px && d_items && d_out && n_lines > 0 && n_items > 0 && img_h > 0 && max_img_w > 0,
             "pt_rec_pp_preprocess_crops: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc;
  if ((rc = rec_pp_lut(e)) != PT_OK) return rc;
  if ((rc = ensure(&e->rec_off, &e->rec_off_cap, (size_t)(n_lines + 1) * sizeof(long long))) != PT_OK) return rc;
  if ((rc = pt_launch_rec_offsets(d_lines, n_lines, reinterpret_cast<long long*>(e->rec_off), s)) != PT_OK) return rc;
  return pt_launch_rec_pp_resize_norm(d_crops_rgb, d_lines, reinterpret_cast<const long long*>(e->rec_off), d_items, n_items, img_h,
                                      max_img_w, e->rec_pp_lut, d_out, s);
}

// ---- image classification (PP-LCNet) -----------------------------------------------------------------------------
static int cls_microbatch() {
  static int v = -1;
  if (v < 0) {
    const char* s = getenv("PT_CLS_MICROBATCH");
    v = s ? atoi(s) : 1024;
    if (v < 1) v = 1;
  }
  return v;
}

static int cls_lut(pt_engine* e) {
  if (e->cls_lut) return PT_OK;
  // transformers.image_transforms.rescale (fp64 product cast to fp32) then normalize in fp32 with
  // IMAGENET_DEFAULT_MEAN / _STD (image_processing_pplcnet.py:267-268, 444-448)
  static const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  float lut[3 * 256];
  for (int c = 0; c < 3; ++c)
    for (int v = 0; v < 256; ++v) {
      const float r = (float)((double)v * (1.0 / 255.0));
      lut[c * 256 + v] = (r - mean[c]) / stdv[c];
    }
  PT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->cls_lut), sizeof(lut)));
  PT_HIP_CHECK(hipMemcpy(e->cls_lut, lut, sizeof(lut), hipMemcpyHostToDevice));
  return PT_OK;
}

int pt_cls_preprocess(pt_engine* e, const uint8_t* d_base, const pt_cls_image* d_images, int n, int max_h, int max_w,
                      int out_h, int out_w, uint16_t* d_out_bf16, pt_stream stream) {
  PT_REQUIRE(e && d_base && d_images && d_out_bf16 && n > 0, "pt_cls_preprocess: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  int rc;
  if ((rc = cls_lut(e)) != PT_OK) return rc;
  return pt_launch_cls_resize_norm(d_base, d_images, n, max_h, max_w, out_h, out_w, e->cls_lut,
                                   pt_split(e), d_out_bf16, reinterpret_cast<hipStream_t>(stream));
}

int pt_cls_forward_net(pt_engine* e, int slot, const uint16_t* d_input_bf16, int n, int in_h, int in_w, int textline,
                       float* d_logits, int* n_classes, pt_stream stream) {
  PT_REQUIRE(e && d_input_bf16 && d_logits && n > 0, "pt_cls_forward_net: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int mb = cls_microbatch();
  const size_t per = (size_t)in_h * in_w * (pt_split(e) ? 8 : 4);
  for (int i0 = 0; i0 < n; i0 += mb) {
    const int nb = (n - i0) < mb ? (n - i0) : mb;
    const int rc = pt_pplcnet_forward_net(e, slot, d_input_bf16 + (size_t)i0 * per, nb, in_h, in_w, textline,
                                          d_logits + (size_t)i0 * PT_CLS_MAX_CLASSES, n_classes, s);
    if (rc != PT_OK) return rc;
  }
  return PT_OK;
}

int pt_cls_forward(pt_engine* e, int slot, const uint8_t* d_base, const pt_cls_image* d_images, int n, int max_h, int max_w,
                   int out_h, int out_w, int textline, float* d_logits, int* n_classes, pt_stream stream) {
  PT_REQUIRE(e && d_base && d_images && d_logits && n > 0, "pt_cls_forward: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc;
  if ((rc = cls_lut(e)) != PT_OK) return rc;
  const int mb = cls_microbatch(), x3 = pt_split(e);
  const size_t per = (size_t)out_h * out_w * (x3 ? 8 : 4);
  if ((rc = ensure(&e->cls_scratch, &e->cls_scratch_cap, (size_t)(mb < n ? mb : n) * per * sizeof(bf16_t))) != PT_OK) return rc;
  bf16_t* xin = reinterpret_cast<bf16_t*>(e->cls_scratch);
  for (int i0 = 0; i0 < n; i0 += mb) {
    const int nb = (n - i0) < mb ? (n - i0) : mb;
    {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "cls resize+norm");
      rc = pt_launch_cls_resize_norm(d_base, d_images + i0, nb, max_h, max_w, out_h, out_w, e->cls_lut, x3, xin, s);
      if (rc != PT_OK) return rc;
    }
    rc = pt_pplcnet_forward_net(e, slot, xin, nb, out_h, out_w, textline, d_logits + (size_t)i0 * PT_CLS_MAX_CLASSES, n_classes, s);
    if (rc != PT_OK) return rc;
  }
  return PT_OK;
}

int pt_cls_forward_lines(pt_engine* e, int slot, const uint8_t* d_pages_rgb, int n_pages, int h, int w,
                         const pt_rec_line* d_lines, const int64_t* h_crop_px, int n_lines, int max_crop_h, int max_crop_w,
                         int out_h, int out_w, int textline, float* d_logits, int* n_classes, pt_stream stream) {
  PT_REQUIRE(e && d_pages_rgb && d_lines && h_crop_px && d_logits && n_lines > 0 && n_pages > 0, "pt_cls_forward_lines: bad arguments");
  PT_HIP_CHECK(hipSetDevice(e->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc;
  if ((rc = cls_lut(e)) != PT_OK) return rc;
  const int mb = cls_microbatch(), x3 = pt_split(e);
  const size_t per = (size_t)out_h * out_w * (x3 ? 8 : 4);
  const int cap = mb < n_lines ? mb : n_lines;
  const size_t desc_bytes = ((size_t)cap * sizeof(pt_cls_image) + 255) & ~(size_t)255;
  if ((rc = ensure(&e->cls_scratch, &e->cls_scratch_cap, desc_bytes + (size_t)cap * per * sizeof(bf16_t))) != PT_OK) return rc;
  pt_cls_image* desc = reinterpret_cast<pt_cls_image*>(e->cls_scratch);
  bf16_t* xin = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(e->cls_scratch) + desc_bytes);
  for (int i0 = 0; i0 < n_lines; i0 += mb) {
    const int nb = (n_lines - i0) < mb ? (n_lines - i0) : mb;
    long long maxpx = 0, total = 0;
    for (int i = 0; i < nb; ++i) {
      const long long px = h_crop_px[i0 + i] > 0 ? h_crop_px[i0 + i] : 0;
      total += px;
      if (px > maxpx) maxpx = px;
    }
    if ((rc = ensure(&e->rec_off, &e->rec_off_cap, (size_t)(nb + 1) * sizeof(long long))) != PT_OK) return rc;
    if ((rc = ensure(&e->rec_crops, &e->rec_crops_cap, (size_t)total * 3 + 16)) != PT_OK) return rc;
    if ((rc = pt_launch_rec_offsets(d_lines + i0, nb, reinterpret_cast<long long*>(e->rec_off), s)) != PT_OK) return rc;
    const long long* d_off = reinterpret_cast<const long long*>(e->rec_off);
    {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "cls warp");
      rc = pt_launch_rec_warp(d_pages_rgb, h, w, d_lines + i0, nb, d_off, reinterpret_cast<uint8_t*>(e->rec_crops), (int)maxpx, s);
      if (rc != PT_OK) return rc;
      if ((rc = pt_launch_cls_desc_from_lines(d_lines + i0, d_off, nb, desc, s)) != PT_OK) return rc;
      rc = pt_launch_cls_resize_norm(reinterpret_cast<const uint8_t*>(e->rec_crops), desc, nb, max_crop_h, max_crop_w, out_h,
                                     out_w, e->cls_lut, x3, xin, s);
      if (rc != PT_OK) return rc;
    }
    rc = pt_pplcnet_forward_net(e, slot, xin, nb, out_h, out_w, textline, d_logits + (size_t)i0 * PT_CLS_MAX_CLASSES, n_classes, s);
    if (rc != PT_OK) return rc;
  }
  return PT_OK;
}

int pt_op_conv2d(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int Cin, const uint16_t* d_w_tiled,
                 const float* d_bias, int N, int ks, int stride, uint16_t* d_out, int out_cstride, int out_coff,
                 int rep, int shuffle_cout, const uint16_t* d_res, int res_mode, int relu, int split, int out_lo_off,
                 pt_stream stream) {
  PT_REQUIRE(e != nullptr, "pt_op_conv2d: null engine");
  ConvDesc d;
  d.in = d_in; d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.w = d_w_tiled; d.bias = d_bias; d.N = N; d.ks = ks;
  d.stride = stride; d.out = d_out; d.out_cstride = out_cstride; d.out_coff = out_coff; d.rep = rep;
  d.shuffle_cout = shuffle_cout; d.res = d_res; d.res_mode = res_mode; d.relu = relu;
  d.split = split; d.out_lo_off = out_lo_off;
  return pt_launch_conv(e, d, reinterpret_cast<hipStream_t>(stream));
}

int pt_op_stem7x7(pt_engine* e, const uint16_t* d_in, int B, int H, int W, const uint16_t* d_w, const float* d_bias,
                  uint16_t* d_out, int split, pt_stream stream) {
  PT_REQUIRE(e != nullptr, "pt_op_stem7x7: null engine");
  return pt_launch_stem7x7(e, d_in, B, H, W, d_w, d_bias, d_out, split, reinterpret_cast<hipStream_t>(stream));
}

int pt_op_maxpool3x3s2(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int C, uint16_t* d_out, int split,
                       pt_stream stream) {
  PT_REQUIRE(e && d_in && d_out, "pt_op_maxpool3x3s2: bad arguments");
  return pt_launch_maxpool3x3s2(d_in, B, H, W, C, d_out, split, reinterpret_cast<hipStream_t>(stream));
}

int pt_op_db_head_final(pt_engine* e, const uint16_t* d_in, int B, int H, int W, const void* d_w, const float* d_bias,
                        float* d_prob, float* d_logits, int split, pt_stream stream) {
  PT_REQUIRE(e && d_in && d_w && d_bias && (d_prob || d_logits), "pt_op_db_head_final: bad arguments");
  return pt_launch_db_head_final(d_in, B, H, W, d_w, d_bias, d_prob, d_logits, split, reinterpret_cast<hipStream_t>(stream));
}

int pt_op_dcn(pt_engine* e, const uint16_t* d_in, const float* d_om, int B, int H, int W, int C, const uint16_t* d_w_tiled,
              const float* d_bias, int N, uint16_t* d_out, int relu, int split, pt_stream stream) {
  PT_REQUIRE(e && d_in && d_om && d_w_tiled && d_bias && d_out, "pt_op_dcn: null argument");
  PT_REQUIRE(B > 0 && H > 0 && W > 0, "pt_op_dcn: empty map (B=%d H=%d W=%d)", B, H, W);
  return pt_launch_dcn_fused(e, d_in, d_om, d_w_tiled, d_bias, d_out, B, H, W, C, N, split, relu, reinterpret_cast<hipStream_t>(stream));
}

// ---- profiling ---------------------------------------------------------------------------------------
int pt_profile_enable(pt_engine* e, int on) {
  PT_REQUIRE(e != nullptr, "pt_profile_enable: null engine");
  PT_REQUIRE(on >= 0 && on < 2 + PT_PROF_NCLASS, "pt_profile_enable: mode %d out of range", on);
  if (on && !e->prof.h_lims)
    PT_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&e->prof.h_lims), PtProfile::MAX_LIMS * sizeof(int), hipHostMallocDefault));
  e->prof.on = on;
  return PT_OK;
}

// per-label readout of the launches recorded since the last read (PT profile mode 1): one line per label,
// "label<TAB>launches<TAB>ms<TAB>flop<TAB>bytes<NL>", NUL-terminated; resets the records like pt_profile_read.  Returns PT_ERR_INVALID when the
// text does not fit `cap` (nothing is consumed then).
int pt_profile_read_labels(pt_engine* e, char* buf, int cap) {
  PT_REQUIRE(e && buf && cap > 0, "pt_profile_read_labels: bad arguments");
  PT_HIP_CHECK(hipDeviceSynchronize());
  struct Row { double ms = 0, flop = 0, bytes = 0; long long n = 0; };
  std::map<std::string, Row> rows;
  for (auto& p : e->prof.pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) != hipSuccess) continue;
    double fl = p.flop;
    if (p.lim_slot >= 0 && p.rows > 0) {
      const int lim = e->prof.h_lims[p.lim_slot];
      fl *= (double)(lim < 0 ? 0 : (lim > p.rows ? p.rows : lim)) / p.rows;
    }
    Row& r = rows[p.label];
    r.ms += ms; r.flop += fl; r.bytes += p.bytes; r.n += 1;
  }
  std::string text;
  char line[160];
  for (auto& kv : rows) {
    snprintf(line, sizeof(line), "%s\t%lld\t%.6f\t%.6e\t%.6e\n", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flop, kv.second.bytes);
    text += line;
  }
  PT_REQUIRE((int)text.size() + 1 <= cap, "pt_profile_read_labels: %d bytes needed, %d given", (int)text.size() + 1, cap);
  memcpy(buf, text.c_str(), text.size() + 1);
  for (auto& p : e->prof.pending) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  e->prof.pending.clear();
  e->prof.n_lims = 0;
  for (int i = 0; i < PT_PROF_NCLASS; ++i) { e->prof.ms[i] = 0; e->prof.launches[i] = 0; e->prof.flop[i] = 0; }
  return PT_OK;
}

int pt_profile_read(pt_engine* e, double* ms_per_class, long long* launches_per_class, double* flop_per_class) {
  PT_REQUIRE(e != nullptr, "pt_profile_read: null engine");
  PT_HIP_CHECK(hipDeviceSynchronize());
  const bool verbose = getenv("PT_PROF_VERBOSE") != nullptr;
  std::map<std::string, std::pair<double, int>> by_label;
  for (auto& p : e->prof.pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      if (verbose) { auto& r = by_label[p.label]; r.first += ms; r.second += 1; }
      e->prof.ms[p.cls] += ms;
      e->prof.launches[p.cls] += 1;
      double fl = p.flop;
      if (p.lim_slot >= 0 && p.rows > 0) {      // row-limited launch: only the rows below the device limit were computed
        const int lim = e->prof.h_lims[p.lim_slot];
        fl *= (double)(lim < 0 ? 0 : (lim > p.rows ? p.rows : lim)) / p.rows;
      }
      e->prof.flop[p.cls] += fl;
    }
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  e->prof.pending.clear();
  e->prof.n_lims = 0;
  if (verbose)
    for (auto& kv : by_label)
      fprintf(stderr, "[pt_prof] %-40s n=%5d total=%9.3f ms avg=%8.4f ms\n", kv.first.c_str(), kv.second.second, kv.second.first,
              kv.second.first / kv.second.second);
  for (int i = 0; i < PT_PROF_NCLASS; ++i) {
    if (ms_per_class) ms_per_class[i] = e->prof.ms[i];
    if (launches_per_class) launches_per_class[i] = e->prof.launches[i];
    if (flop_per_class) flop_per_class[i] = e->prof.flop[i];
    e->prof.ms[i] = 0;
    e->prof.launches[i] = 0;
    e->prof.flop[i] = 0;
  }
  return PT_OK;
}

}  // namespace api
}  // namespace PT_FMT_NS
