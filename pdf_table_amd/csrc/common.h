// common.h -- engine internals shared by the HIP translation units of libpdftable_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/pdftable_hip.h"
#include "act16.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits

void pt_set_error(const char* fmt, ...);

#define PT_HIP_CHECK(expr)                                                                  \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      pt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return PT_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)

#define PT_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      pt_set_error(__VA_ARGS__);   \
      return PT_ERR_INVALID;       \
    }                              \
  } while (0)

// ---- weight container ("PTW1") ----------------------------------------------------------------
// header: char magic[4]="PTW1"; uint32 n_tensors; then n_tensors entries of
//   char name[96]; uint32 dtype (0=bf16, 1=f32, 2=i32); uint32 ndim; uint32 dims[6]; uint64 offset; uint64 nbytes
// data offsets are relative to the blob start and 256-byte aligned.
struct PtTensor {
  int dtype = 0;
  int ndim = 0;
  uint32_t dims[6] = {0, 0, 0, 0, 0, 0};
  const void* d_ptr = nullptr;  // device pointer into the engine-owned copy of the blob
  size_t nbytes = 0;
};

struct PtModel {
  void* d_blob = nullptr;
  size_t nbytes = 0;
  std::map<std::string, PtTensor> tensors;
  // host copies of small integer tensors (configuration words stored beside the weights), read from the device ONCE per loaded blob:
  // a synchronous hipMemcpy per call runs on the null stream and waits for everything queued on the caller's default stream
  std::map<std::string, std::vector<int32_t>> host_words;
  int act_f16 = 0;   // 1: the blob's 16-bit tensors are IEEE half (packed for PT_PRECISION_F16), 0: bfloat16
  const PtTensor* find(const std::string& n) const {
    auto it = tensors.find(n);
    return it == tensors.end() ? nullptr : &it->second;
  }
};

// ---- bump arena for activations ------------------------------------------------------------------
struct PtArena {
  char* base = nullptr;
  size_t cap = 0;
  size_t off = 0;
  size_t high = 0;
  void reset() { off = 0; }
  // returns nullptr when out of space (caller grows and retries)
  void* take(size_t n) {
    size_t a = (off + 255) & ~size_t(255);
    if (a + n > cap) {
      high = a + n > high ? a + n : high;
      off = a + n;
      return nullptr;
    }
    off = a + n;
    if (off > high) high = off;
    return base + a;
  }
};

// Capacity an arena is (re)allocated with: its high-water mark plus 1 MiB of slack
static inline size_t pt_arena_round(size_t high) { return high + (1u << 20); }

// Pinned host staging for small per-call tables (token maps, tile lists) that are the SOURCE of an asynchronous host-to-device copy:
// a stack vector dies before a deferred copy reads it, and waiting for the copy means a stream synchronise per call.  A slot is
// re-used only after the event recorded behind its copy has completed (the wait is real only if the ring wrapped with work pending).
struct PtPinnedRing {
  static constexpr int SLOTS = 8;
  void* buf[SLOTS] = {};
  size_t cap[SLOTS] = {};
  hipEvent_t ev[SLOTS] = {};
  bool pending[SLOTS] = {};
  int next = 0;
  int cur = -1;
  // -> pinned pointer of at least `bytes`, or nullptr (hip error left for the caller to report)
  void* acquire(size_t bytes) {
    cur = next;
    next = (next + 1) % SLOTS;
    if (pending[cur]) {
      if (hipEventSynchronize(ev[cur]) != hipSuccess) return nullptr;
      pending[cur] = false;
    }
    if (bytes > cap[cur]) {
      if (buf[cur]) (void)hipHostFree(buf[cur]);
      buf[cur] = nullptr;
      cap[cur] = 0;
      const size_t want = (bytes + 4095) & ~size_t(4095);
      if (hipHostMalloc(&buf[cur], want) != hipSuccess) return nullptr;
      cap[cur] = want;
    }
    return buf[cur];
  }
  // record "the copies out of the slot acquired last are queued on s"
  int release(hipStream_t s) {
    if (cur < 0) return 0;
    if (!ev[cur] && hipEventCreateWithFlags(&ev[cur], hipEventDisableTiming) != hipSuccess) return 1;
    if (hipEventRecord(ev[cur], s) != hipSuccess) return 1;
    pending[cur] = true;
    return 0;
  }
  void destroy() {
    for (int i = 0; i < SLOTS; ++i) {
      if (buf[i]) (void)hipHostFree(buf[i]);
      if (ev[i]) (void)hipEventDestroy(ev[i]);
      buf[i] = nullptr; cap[i] = 0; ev[i] = nullptr; pending[i] = false;
    }
  }
};

enum { PT_ARENA_DET = 0, PT_ARENA_REC, PT_ARENA_TSR, PT_ARENA_TSRP, PT_ARENA_LAYOUT, PT_ARENA_COUNT };   // TSRP: the Lore processor

struct PtProfile {
  int on = 0;         // 0 off, 1 every launch, 2 + class: only the launches of kernel class (on - 2)
  double ms[PT_PROF_NCLASS] = {0, 0, 0, 0};
  long long launches[PT_PROF_NCLASS] = {0, 0, 0, 0};
  double flop[PT_PROF_NCLASS] = {0, 0, 0, 0};
  struct Pending {
    hipEvent_t a, b;
    int cls;
    double flop;          // algorithmic FLOP of the launch (for a row-limited launch: of its full extent)
    double bytes = 0;     // algorithmic HBM bytes of the launch (inputs once + outputs once + weights), 0 = not stated by the launcher
    int lim_slot = -1;    // >= 0: the launch was row-limited on the device (ConvDesc.ylimit); h_lims[lim_slot] receives the limit
    int rows = 0;         // ... and this is the launch's full extent in output rows: credited flop * min(limit, rows) / rows
    char label[48];
  };
  std::vector<Pending> pending;
  double next_bytes = 0;  // set by a launcher right before it opens its PtProfScope, which takes (and clears) it
  int* h_lims = nullptr;  // pinned: device row limits copied back behind their launches (only while profiling)
  int n_lims = 0;
  static constexpr int MAX_LIMS = 1 << 16;
};

struct pt_engine {
  int device = 0;
  int num_cu = 256;
  // one activation arena per stage, so that calls of DIFFERENT stages may be in flight on different streams (two calls of
  // one stage share its arena and must be stream-ordered); layout and the PP-LCNet classifiers share PT_ARENA_LAYOUT
  PtArena arenas[PT_ARENA_COUNT];
  std::map<int, PtModel> models;
  PtProfile prof;
  int precision = 0;  // PT_PRECISION_* (PT_PRECISION_F16: the exported functions run namespace pt_f16, every other value pt_bf16)
  int det_kind = PT_MODEL_DB_RESNET18;   // detector network pt_det_forward* runs: the one loaded last
  // engine-owned scratch of the recognition stage (outside the arena, which every net forward resets)
  void* rec_crops = nullptr; size_t rec_crops_cap = 0;
  void* rec_gray = nullptr; size_t rec_gray_cap = 0;
  void* rec_off = nullptr; size_t rec_off_cap = 0;
  std::vector<long long> rec_off_host;
  void* det_in = nullptr; size_t det_in_cap = 0;             // pre-processed pages of pt_det_forward (one micro-batch)
  void* zero_page = nullptr;  // 8 KiB of zeros: DMA source for halo pixels outside the image
  void* tsr_scratch = nullptr; size_t tsr_scratch_cap = 0;   // candidate lists of the Lore decode
  void* layout_scratch = nullptr; size_t layout_scratch_cap = 0;   // layout input + head maps (pt_layout_forward)
  float* tsr_lut = nullptr;                                  // [3][256] normalisation table of the Lore pre-process
  float* cls_lut = nullptr;                                  // [3][256] ... of the PP-LCNet pre-process
  float* rec_pp_lut = nullptr;                               // [256] (v / 255 - 0.5) / 0.5 of the PP-OCR recognition pre-process
  void* cls_scratch = nullptr; size_t cls_scratch_cap = 0;   // network input + image descriptors of pt_cls_forward*
  alignas(8) unsigned char tsr_decode_state[128] = {};       // lore_decode.hip: DecodeState of the sparse-head decode in flight
  void* lstm_scratch = nullptr;                              // rec_kernels.hip: h exchange buffers + step counters of the cluster LSTM
  void* lstm_scratch8 = nullptr; size_t lstm_scratch8_cap = 0;   // ... of the eight-member hi/lo cluster LSTM
  int* lstm_err = nullptr;                                   // pinned, device-visible: set by a cluster member that gave up waiting
  int lstm_max_cl = 0;                                       // clusters per direction per launch (num_cu / 8)
  int dcn_mfma = 0;                                          // pt_engine_set_dcn_mfma: bf16-mode deformable convolutions blend on the matrix pipe (dcn_mfma_kernel)
  int mtl_kv_fp8 = 0;                                        // pt_engine_set_mtl_kv_fp8: MtlTabNet source-attention keys / values of the structure loop as fp8 (bf16 mode only)
  int lstm_cluster = 1;                                      // 1: weight-stationary cluster kernel (bf16 mode); 0: streaming kernel
  // crnn_model.hip: activations of an all-padding text line after every limited conv layer, per precision (bf16 / hi-lo);
  // valid until the CRNN weights are loaded again
  void* rec_zero[2] = {nullptr, nullptr};
  bool rec_zero_valid[2] = {false, false};
  hipEvent_t rec_zero_ready[2] = {nullptr, nullptr};         // recorded behind the build of rec_zero[i]: calls on other streams wait for it
  void* rec_limits = nullptr; size_t rec_limits_cap = 0;     // per-line column limits of the call in flight
  int rec_ragged = 1;                                        // PT_REC_RAGGED=0: compute the padding too (A/B switch)
  // cvit_model.hip: host images of the chunk maps of the last 16 micro-batches.  They are the sources of asynchronous
  // host-to-device copies: a stack vector could die before a deferred copy reads it; a slot is reused only 16 micro-batches
  // (thousands of launches on the same stream) later
  std::vector<int> cvit_maps[16][2];
  int cvit_slot = 0;
  PtPinnedRing stage_ring;                                   // pinned sources of small asynchronous uploads (Lore processor token maps, ConvNextViT chunk maps)
  void* mtl_state = nullptr;                                 // mtl_decoder.hip: buffers + cell lists between pt_tsr_mtl_structure and pt_tsr_mtl_cells
};

// Everything below exists once per activation format (act16.h): the launchers and the model drivers of namespace pt_bf16 and of namespace pt_f16.
namespace PT_FMT_NS {

// ---- conv launcher (conv_igemm.hip) -----------------------------------------------------------------
struct ConvDesc {
  // input NHWC bf16
  const bf16_t* in = nullptr;
  int B = 0, H = 0, W = 0, Cin = 0;  // Cin multiple of 32 (channel stride == Cin)
  // 1x1 stride-1 GEMM over a channel concatenation that is never materialised (DLA-34 `Root`: conv(torch.cat(children)),
  // center_net/modeling_centernet.py:196-214): K walks `nseg` tensors of seg_c[i] channels each (multiples of 32, sum = Cin),
  // `in` is segment 0, in_more[i - 1] segment i; every segment is its own NHWC tensor (channel stride seg_c[i])
  const bf16_t* in_more[3] = {nullptr, nullptr, nullptr};
  int seg_c[4] = {0, 0, 0, 0};
  int nseg = 1;
  // packed weights [N/64][Cin/32][taps][64][32] bf16, bias fp32 [N]
  const bf16_t* w = nullptr;
  const float* bias = nullptr;
  int N = 0;       // GEMM N: Cout, or 4*Cout for a 2x2/s2 transposed conv
  int ks = 3;      // 1 or 3 (square), padding = ks/2
  int stride = 1;  // 1 or 2
  // output
  bf16_t* out = nullptr;
  int out_cstride = 0;  // channels per pixel of the output buffer
  int out_coff = 0;     // channel offset inside it (concat fusion)
  int rep = 1;          // nearest-neighbour replicate factor (fused nn.Upsample)
  int shuffle_cout = 0; // >0: pixel-shuffle epilogue of ConvTranspose2d(k=2,s=2) with this Cout
  // residual
  const bf16_t* res = nullptr;
  int res_mode = 0;  // 0 none, 1 same resolution, 2 half resolution (fused nearest x2 upsample + add)
  int relu = 0;      // activation: 0 none, 1 ReLU, 2 hardswish, 3 PReLU(slope), 4 GELU (erf)
  const float* slope = nullptr;   // device pointer to the PReLU slope (relu == 3)
  int pool = 0;                   // 1: MaxPool2d(2,2), 2: MaxPool2d((2,1)), 3: (2,1) with rows -> channel groups; fused behind bias + ReLU (plain 3x3 stride-1 layers)
  const int* ylimit = nullptr;    // device int: output rows >= *ylimit are not computed (whole tiles; v1 kernel only)
  // ragged images (text lines padded to a common width): device int [B], output COLUMNS >= xlimit[b] of image b are not
  // computed (whole tiles; v1 kernel only) -- the caller fills them (pt_launch_crnn_fill).  xlimit_cols: device int, the
  // number of columns below the limits summed over the images, tile-rounded (roofline accounting only)
  const int* xlimit = nullptr;
  const int* xlimit_cols = nullptr;
  // the same for a [1, lines, T, C] view (1x1 GEMMs over the sequence): device int [H], limit of ROW oy; a tile of several
  // rows is skipped when it lies right of all its rows' limits (rows with a smaller limit get values the caller overwrites)
  const int* xlimit_rows = nullptr;
  // with xlimit, maps <= 4 rows high (the 4 x 64 patch): the compacted list of live 32-column blocks (rows_live_list_kernel over the same limits:
  // [0] = count, [1 + i] = image * (Wo / 32) + block); a workgroup multiplies two of them -- possibly of two images -- instead of one image's 64 columns
  const int* block_list = nullptr;
  // bf16x3 precision mode: in/res/out hold (hi | lo) channel groups; w is [N/64][3*Cin/32][taps][64][32]
  int split = 0;
  // split = 2 (PT_PRECISION_F16X2): same (hi | lo) tensors, but w holds fp16 tiles [N/64][2*Cin/32][taps][64][32] (blob suffix .wh: the
  // same fp16 tile for the x_hi and the x_lo K chunks) and the launch takes the two-pass fp16 variant of conv_igemm_kernel
  int out_lo_off = 0;  // channel distance between the hi and lo halves in the output buffer
  // fused DB head: see ConvK in conv_igemm.hip (requires shuffle_cout == 64)
  const void* head_w = nullptr;
  const float* head_b = nullptr;
  float* head_prob = nullptr;
  float* head_logits = nullptr;
  // fused arg-max over N: float2 (max, index-as-bits) per (output row, 64-wide N tile); no activation tensor is written
  float* argmax_part = nullptr;
  // only output channels [0, n_valid) are stored (0 = all N); out_f32: fp32 [pixel][out_cstride] output instead of bf16
  int n_valid = 0;
  float* out_f32 = nullptr;
  const float* res_f32 = nullptr;   // fp32 residual laid out like out_f32 (may alias it)
  // roofline accounting only (PtProfile): the layer's real output channels when N / n_valid are padded (0 = n_valid, else N),
  // and the fraction of the launch's output pixels the algorithm needs (patch mosaics compute 9 pixels to use one)
  // 3x3 stride-1 layers: bit (r * 3 + s) of tap_mask[t] set = tap (r, s) of the 64-channel output tile t has non-zero weights; 0 = all
  // taps.  Tiles past the eighth use every tap.  (Phase convolutions of an up-sampled input: 4 of 9 taps per tile.)
  unsigned tap_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // register-epilogue launches: a row-tile's 32 pixels x 64 channels leave through a wave-private 4 KB LDS tile as whole 128-byte lines
  // instead of 16-byte pieces per lane (one more barrier per tile: pays for large write-heavy layers, not for small GEMMs)
  int xp_store = 0;
  int alg_n = 0;
  double alg_scale = 1.0;
};
int pt_launch_conv(pt_engine* e, const ConvDesc& d, hipStream_t s);
// CRNN conv0 + pool + conv1 + pool in one launch (conv_igemm.hip: crnn_conv01_kernel), single-pass modes
int pt_launch_crnn_conv01(pt_engine* e, const bf16_t* gray, int n, const float* w64x9, const float* b0, const bf16_t* w1, const float* b1, bf16_t* p1,
                          const int* xlimit, const int* xlimit_cols, hipStream_t s);
// depthwise k x k + pointwise 1x1 in one launch (conv_igemm.hip: dwpw_kernel); PT_ERR_INVALID: outside its shapes, run the two launches
int pt_launch_dwpw(pt_engine* e, const bf16_t* in, int B, int H, int W, int C, const float* dw_w, const float* dw_b, int k, int stride, int dw_act,
                   const ConvDesc& pw, hipStream_t s);

// stem: 7x7 p3 conv (stride 1 or 2) on NHWC4 bf16 input, 64 GEMM outputs of which the first n_valid (0 = 64) are
// stored, bias + ReLU (conv_igemm.hip)
int pt_launch_stem7x7(pt_engine* e, const bf16_t* in, int B, int H, int W, const bf16_t* w, const float* bias,
                      bf16_t* out, int split, hipStream_t s, int stride = 2, int n_valid = 0);

// ---- misc kernels (det_kernels.hip) ---------------------------------------------------------------------
int pt_launch_det_preprocess(const uint8_t* pages, int n, int h, int w, int nh, int nw, int flavour, int split,
                             bf16_t* out, hipStream_t s);
// fused modulated deformable 3x3 convolution (lore_kernels.hip): x NHWC bf16, om fp32 [pixel][32], w tiled as a 1x1 conv over K = 9C
// omw / omb != null (allowed when pt_dcn_fuses_om): the layer's 27-channel offset / mask conv runs in the kernel's prologue from its `.om` weight tiles; om is not read
int pt_launch_dcn_fused(pt_engine* e, const bf16_t* x, const float* om, const bf16_t* w, const float* bias, bf16_t* out,
                        int B, int H, int W, int C, int N, int split, int relu, hipStream_t s, const bf16_t* omw = nullptr, const float* omb = nullptr);
bool pt_dcn_fuses_om(const pt_engine* e, int C, int split);
int pt_launch_maxpool3x3s2(const bf16_t* in, int B, int H, int W, int C, bf16_t* out, int split, hipStream_t s);
int pt_launch_stem7x7_pool(pt_engine* e, const bf16_t* in, int B, int H, int W, const bf16_t* w, const float* bias, bf16_t* out,
                           hipStream_t s);   // ResNet-18 stem + MaxPool2d(3,2,1) in one kernel (bf16 mode)
int pt_launch_db_head_final(const bf16_t* in, int B, int H, int W, const void* w4x64, const float* bias, float* prob,
                            float* logits, int split, hipStream_t s);
// bitmap != null: prob > thresh bit-packed by the same kernel (pt_launch_bitmap's words, no dilation)
int pt_launch_db_head_mfma(const bf16_t* in, int B, int H, int W, const bf16_t* w3, const float* b3, const bf16_t* w6,
                           const float* b6, float* prob, float* logits, hipStream_t s, uint32_t* bitmap = nullptr, float thresh = 0.f);
int pt_launch_bitmap(const float* prob, int n, int H, int W, float thresh, int dilate, uint32_t* bitmap,
                     hipStream_t s);
int pt_launch_box_scores(const float* prob, int n, int H, int W, const float* boxes, int nb, float* scores,
                         hipStream_t s);

// ---- mobile-net kernels shared by PicoDet and DB-ProxylessNAS (layout_kernels.hip) ----------------------
#define PT_SE_CHUNKS 64
int pt_launch_stem3x3s2(const bf16_t* in, const float* w, const float* b, bf16_t* out, int B, int H, int W, int split,
                        hipStream_t s, int variant);
int pt_launch_dwconv(const bf16_t* in, const float* w, const float* b, bf16_t* out, int B, int H, int W, int C, int k,
                     int stride, int act, int split, hipStream_t s, const float* slope);
int pt_launch_se(const bf16_t* x, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                 bf16_t* out, int B, int HW, int C, int split, hipStream_t s, int hidden, int mode, float* part);
int pt_launch_add(const bf16_t* a, const bf16_t* b, bf16_t* out, long long npix, int C, int split, hipStream_t s);
int pt_launch_chan_mean(const bf16_t* x, int B, int HW, int C, int split, float* part, bf16_t* mean, int rows, hipStream_t s);
int pt_pplcnet_forward_net(pt_engine* e, int slot, const bf16_t* x, int n, int H, int W, int textline, float* logits,
                           int* n_classes, hipStream_t s);
int pt_launch_cls_resize_norm(const uint8_t* base, const pt_cls_image* images, int n, int max_h, int max_w, int OH, int OW,
                              const float* lut, int split, bf16_t* out, hipStream_t s);
int pt_launch_cls_desc_from_lines(const pt_rec_line* lines, const long long* off, int n, pt_cls_image* images, hipStream_t s);
int pt_launch_dbnas_tail(const bf16_t* y, const float* tw, int B, int H4, int W4, int split, float* prob, float* logits,
                         hipStream_t s);
int pt_dbnas_forward_net(pt_engine* e, const bf16_t* x, int n, int H, int W, float* prob, float* logits, hipStream_t s);

// ---- recognition kernels (rec_kernels.hip) --------------------------------------------------------------
int pt_launch_rec_offsets(const pt_rec_line* lines, int n, long long* off, hipStream_t s);
int pt_launch_rec_warp(const uint8_t* pages, int ph, int pw, const pt_rec_line* lines, int n_lines,
                       const long long* pix_off, uint8_t* crops, int max_crop_px, hipStream_t s);
int pt_launch_rec_resize_gray(const uint8_t* crops, const pt_rec_line* lines, const long long* pix_off, int n_lines,
                              int split, bf16_t* out, hipStream_t s);
int pt_launch_rec_pp_resize_norm(const uint8_t* crops, const pt_rec_line* lines, const long long* pix_off,
                                 const pt_rec_pp_item* items, int n_items, int img_h, int max_img_w, const float* lut, float* out,
                                 hipStream_t s);
// per-layer column limits of the ragged CRNN conv stack (crnn_model.hip) from the lines' crop sizes, and the fill of the
// columns a limited conv skipped with the activations an all-padding line has there
struct PtCrnnLimits {
  int* lim[6];        // device int [n] each: conv1, conv2a, conv2b, conv3a, conv3b output-column limits; [5]: conv0 (pooled columns)
  int* cols;          // device int [8]: sum over lines of the tile-rounded limits ([5]: the sequence GEMMs, 32-step tiles)
  int* glist;         // device int [1 + 5 n]: the live 32-step row groups of the sequence GEMMs, compacted (rows_live_list_kernel; = conv3b's block list)
  int* blist3a;       // device int [1 + 5 n]: the live 32-column blocks of conv3a (limit lim[3])
  int* blist2a;       // the same for conv2a (lim[1]) and conv2b (lim[2]): the 8-row maps' tiles of the blocked DMA kernel (conv3x3_pipe_kernel<.., 8>)
  int* blist2b;
};
int pt_launch_rows_live_list(const int* lim, int n, int* glist, hipStream_t s);
int pt_launch_crnn_limits(const pt_rec_line* lines, int n, const PtCrnnLimits& L, hipStream_t s);
// end_lim / end_tile: fill only up to column roundup(end_lim[b], end_tile) (inclusive: the halo column the next limited conv
// reads); end_lim == null: to the end of the row
int pt_launch_crnn_fill(bf16_t* out, const bf16_t* ref, const int* lim, int tile_w, int div, int n, int rows, int W, int cs,
                        hipStream_t s, const int* end_lim = nullptr, int end_tile = 1);
int pt_launch_crnn_conv0_pool(const bf16_t* in, int n, int H, int W, const float* w64x9, const float* bias, int split,
                              bf16_t* out, hipStream_t s, const int* xlim = nullptr);
int pt_launch_maxpool_kxk(const bf16_t* in, int n, int H, int W, int C, int kh, int kw, int h2c, int split, bf16_t* out,
                          hipStream_t s);
int pt_launch_lstm(pt_engine* e, const bf16_t* gx, const bf16_t* whh, bf16_t* hout, int B, int T, int split, hipStream_t s);
int pt_launch_gemm_argmax(const bf16_t* A, long long M, int K, const bf16_t* W, const float* bias, int N, int* ids, float* maxv,
                          hipStream_t s);
int pt_launch_gemm_argmax_x3(const bf16_t* A, long long M, int K, const bf16_t* W3, const float* bias, int N, int n_real, int* ids, float* maxv,
                             void* scratch, hipStream_t s);
// tlim != null: rows are (line, t) with T = 160 steps per line; only the 32-step row groups of the list are computed (PtCrnnLimits.glist)
int pt_launch_gemm_rows_x3(const bf16_t* A, long long M, int K, const bf16_t* W3, const float* bias, int N, bf16_t* out, int relu, hipStream_t s,
                           const int* tlim = nullptr);
int pt_launch_gemm_rows(const bf16_t* A, long long M, int K, const bf16_t* W, const float* bias, int N, bf16_t* out, int relu,
                        hipStream_t s, const int* tlim = nullptr);
int pt_launch_argmax_reduce(const float* part, long long rows, int ntiles, int* ids, float* maxv, hipStream_t s);
// d_lines != null: the lines' crop sizes are known, so the conv stack does no work on the zero padding right of the text
// (bit-identical results: skipped columns are filled with what an all-padding line has there)
int pt_crnn_forward_net(pt_engine* e, const bf16_t* gray, int n, int32_t* ids, float* maxlogit, hipStream_t s,
                        const pt_rec_line* d_lines = nullptr);

// gray fp32, layout 0 = chunks [3 n, 32, 300], 1 = lines [n, 32, 804]; ids int32 [n, PT_CVIT_T] (cvit_model.hip)
// h_text_w (HOST, may be null): text width of every line after the keep-ratio resize -- all-padding chunks are then computed once
int pt_cvit_forward_net(pt_engine* e, const float* gray, int layout, int n, int32_t* ids, float* maxlogit, hipStream_t s,
                        const int* h_text_w = nullptr);
int pt_launch_rec_resize_gray_f32(const uint8_t* crops, const pt_rec_line* lines, const long long* pix_off, int n_lines, int tw,
                                  float* out, hipStream_t s);

int pt_mtl_backbone_forward_net(pt_engine* e, const bf16_t* x, int n, int H, int W, float* f3, hipStream_t s);
// mtl_decoder.hip
void pt_mtl_release(pt_engine* e);
int pt_mtl_preprocess(pt_engine* e, const uint8_t* pages, int ph, int pw, const pt_tsr_table* tabs, int n, int size, bf16_t* out, hipStream_t s);
int pt_mtl_decoder_config(pt_engine* e, int32_t* out13);
int pt_mtl_structure(pt_engine* e, const float* f3, int n, int hw, float* d_tag_logits, float* d_boxes, int32_t* h_lens, int32_t* h_cell_counts,
                     int force_redecode, hipStream_t s);
int pt_mtl_cells(pt_engine* e, int total, int32_t* d_cell_ids, float* d_cell_prob, float* d_cell_logits, int32_t* h_steps, int force_redecode,
                 hipStream_t s);

// ---- models ---------------------------------------------------------------------------------------------
// bitmap / bitmap_done (optional): where the head kernel can threshold its own output it writes the bit-packed map too and sets *bitmap_done = 1
int pt_db_forward_net(pt_engine* e, const bf16_t* x, int n, int H, int W, float* prob, float* logits, hipStream_t s, uint32_t* bitmap = nullptr,
                      float thresh = 0.f, int* bitmap_done = nullptr);
int pt_launch_tsr_preprocess(const uint8_t* pages, int ph, int pw, const pt_tsr_table* tabs, int n, int H, int W, int bgr,
                             const float* lut, bf16_t* out, int split, hipStream_t s);
int pt_picodet_forward_net(pt_engine* e, const bf16_t* x, int n, int H, int W, float* h0, float* h1, float* h2, float* h3,
                          hipStream_t s);
int pt_launch_pico_candidates(const float* head, int B, int A, int ncls, int level, float thr_lo, int max_cands, float* cands,
                              int* counts, hipStream_t s);
int pt_lore_decode(pt_engine* e, const float* hm, const float* st, const float* wh, const float* ax, const float* cr,
                   const float* reg, int B, int H, int W, int wiz_rev, float vis_thresh, int* d_counts, float* d_dets,
                   float* d_logi, hipStream_t s);
void pt_lore_mosaic_rows(int B, int* rows_ax, int* rows_cr, int* rows_cell, int* rows_corner);
int pt_lore_decode_peaks(pt_engine* e, const float* hm, int B, int H, int W, int wiz_rev, float vis_thresh, int* d_counts,
                         const int** d_lim_cell, const int** d_lim_corner, hipStream_t s);
int pt_lore_peak_patches(pt_engine* e, const bf16_t* feat, int B, int H, int W, int C, int split, bf16_t* mos_cell,
                         bf16_t* mos_corner, hipStream_t s);
int pt_lore_decode_boxes(pt_engine* e, const float* reg_cell, const float* wh, const float* reg_corner, const float* st, int B,
                         int H, int W, const int** d_lim_ax, const int** d_lim_cr, hipStream_t s);
int pt_lore_patch_gather(pt_engine* e, const bf16_t* feat, int B, int H, int W, int C, int split, bf16_t* mos_ax, bf16_t* mos_cr,
                         hipStream_t s);
int pt_lore_decode_sparse(pt_engine* e, const float* ax_mos, const float* cr_mos, int B, int H, int W, float vis_thresh,
                          int* d_counts, float* d_dets, float* d_logi, hipStream_t s);
int pt_lore_forward_decode(pt_engine* e, const bf16_t* x, int n, int H, int W, int wiz_rev, float vis_thresh, int* d_counts,
                           float* d_dets, float* d_logi, hipStream_t s);
int pt_lore_wireless_forward_net(pt_engine* e, const bf16_t* x, int n, int H, int W, float* hm, float* st, float* wh,
                                 float* ax, float* cr, float* reg, hipStream_t s);
int pt_lore_process(pt_engine* e, const float* d_logi, const float* d_dets, const int32_t* h_counts, int n_tables,
                    int use_2dpe, float* d_logic, float* d_stacked, hipStream_t s);
int pt_lore_forward_net(pt_engine* e, const bf16_t* x, int n, int H, int W, float* hm, float* st, float* wh, float* ax,
                        float* cr, float* reg, hipStream_t s);

// storage: both tolerance modes keep every activation as a (hi | lo) bf16 pair ("split" layout); PT_PRECISION_F16 (namespace pt_f16 only) is
// single-pass like PT_PRECISION_BF16
static inline int pt_split(const pt_engine* e) { return e->precision == PT_PRECISION_BF16X3 || e->precision == PT_PRECISION_F16X2; }
// arithmetic: F16X2 = the conv / GEMM launches that have the variant multiply the pair (converted to fp16 while it is staged in LDS:
// exact, a bf16 value has 8 significant bits) with SINGLE fp16 weights in two MFMA passes; every other kernel runs as in BF16X3
static inline int pt_f16x2(const pt_engine* e) { return e->precision == PT_PRECISION_F16X2; }

// profiling helper: bracket a launch with events when enabled
struct PtProfScope {
  pt_engine* e;
  hipStream_t s;
  int idx = -1;
  PtProfScope(pt_engine* e_, hipStream_t s_, int cls, double flop, const char* label = "") : e(e_), s(s_) {
    double bytes = 0;
    if (e) { bytes = e->prof.next_bytes; e->prof.next_bytes = 0; }
    if (e && e->prof.on && (e->prof.on == 1 || e->prof.on - 2 == cls)) {
      PtProfile::Pending p;
      p.bytes = bytes;
      if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
      p.cls = cls;
      p.flop = flop;
      strncpy(p.label, label, sizeof(p.label) - 1);
      p.label[sizeof(p.label) - 1] = 0;
      (void)hipEventRecord(p.a, s);
      e->prof.pending.push_back(p);
      idx = (int)e->prof.pending.size() - 1;
    }
  }
  ~PtProfScope() {
    if (idx >= 0) (void)hipEventRecord(e->prof.pending[idx].b, s);
  }
};

// a weight blob is packed for ONE storage format (weights.py: the "__act_f16__" tensor; absent = bf16): refuse the other one loudly
static inline int pt_model_format_ok(const PtModel& m, const char* what) {
  if (m.act_f16 != PT_ACT_F16) {
    pt_set_error("%s: the loaded weight blob holds %s tiles but the engine computes in %s (pack it with fmt=\"%s\" / set the matching precision)", what,
                 m.act_f16 ? "fp16" : "bf16", PT_FMT_NAME, PT_FMT_NAME);
    return 0;
  }
  return 1;
}

}  // namespace PT_FMT_NS
