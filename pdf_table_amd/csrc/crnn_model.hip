// crnn_model.hip -- launch graph of the CRNN text-line recogniser on the engine's kernels.
//
// Reference graph: CRNN.forward (model/crnn/modeling_crnn.py:92-113): RGB->gray, 7 conv(+BN+ReLU) layers with
// four max-pools, two BidirectionalLSTM blocks (nn.LSTM + Linear), Linear(512 -> 7644); then the arg-max of
// OCRRecognition.postprocess (model/ocr_recognition/modeling_ocr_recognition.py:168-171).
// Mapping onto kernels:
//   conv0 + pool0            crnn_conv0_pool_kernel (VALU, K = 9)
//   conv1..conv3 (3x3)       conv_igemm_kernel<3,1> (8x32 patches; 4x64 patches for the 4-row maps of conv3.*)
//   pools                    maxpool_kxk_kernel; the last (2,1) pool writes H into channels so that
//   conv4 (2,1)/s(2,1)       becomes a 1x1 GEMM with K = 1024
//   LSTM input projections   1x1 GEMM (both directions and both biases in one N = 2048 launch)
//   LSTM recurrences         lstm_dir_kernel (32 lines per workgroup, W_hh streamed from L2)
//   embeddings               1x1 GEMMs
//   classifier + arg-max     1x1 GEMM with the arg-max epilogue (N = 7680 = 120 tiles; padded classes carry a
//                            -3e38 bias) + argmax_reduce_kernel: the [lines, 160, 7644] logits never exist in HBM
// All 1x1 GEMMs view the [n, 1, 160, C] activations as one image [1, n, 160, C] so that a 4x32 patch is 4 lines x
// 32 time steps (no empty MFMA rows).
#include <stdlib.h>

#include "common.h"

namespace {

struct ConvW {
  const PtTensor *w = nullptr, *b = nullptr;
};

int getw(const PtModel& m, const std::string& base, bool x3, ConvW& out) {
  out.w = m.find(base + (x3 ? ".w3" : ".w"));
  out.b = m.find(base + ".b");
  if (!out.w || !out.b) {
    pt_set_error("CRNN weight blob lacks '%s'", base.c_str());
    return PT_ERR_FORMAT;
  }
  return PT_OK;
}

inline const bf16_t* W(const PtTensor* t) { return reinterpret_cast<const bf16_t*>(t->d_ptr); }
inline const float* Bv(const PtTensor* t) { return reinterpret_cast<const float*>(t->d_ptr); }

}  // namespace

int pt_crnn_forward_net(pt_engine* e, const bf16_t* gray, int n, int32_t* ids, float* maxlogit, hipStream_t s) {
  PT_REQUIRE(e && gray && ids && n > 0, "crnn: bad arguments");
  auto it = e->models.find(PT_MODEL_CRNN);
  if (it == e->models.end()) {
    pt_set_error("CRNN weights not loaded (pt_weights_load(PT_MODEL_CRNN))");
    return PT_ERR_STATE;
  }
  const PtModel& M = it->second;
  const int x3 = e->precision == PT_PRECISION_BF16X3 ? 1 : 0;
  const int m = x3 ? 2 : 1;
  int rc;
  ConvW c1, c2a, c2b, c3a, c3b, c4, xp1, em1, xp2, em2, cls;
#define G(name, dst) if ((rc = getw(M, name, x3 != 0, dst)) != PT_OK) return rc
  G("conv1", c1); G("conv2a", c2a); G("conv2b", c2b); G("conv3a", c3a); G("conv3b", c3b); G("conv4", c4);
  G("lstm1.xproj", xp1); G("lstm1.emb", em1); G("lstm2.xproj", xp2); G("lstm2.emb", em2); G("cls", cls);
#undef G
  const PtTensor* c0w = M.find(x3 ? "conv0.wf32" : "conv0.wbf");
  const PtTensor* c0b = M.find("conv0.b");
  const PtTensor* whh1 = M.find("lstm1.whh");
  const PtTensor* whh2 = M.find("lstm2.whh");
  if (!c0w || !c0b || !whh1 || !whh2) {
    pt_set_error("CRNN weight blob lacks conv0 / lstm tensors");
    return PT_ERR_FORMAT;
  }
  const int T = PT_REC_T, NT = 7680 / 64;

  struct {
    bf16_t *a0, *a1, *p1, *c2a, *c2b, *p2, *c3a, *c3b, *p3, *f, *gx, *h, *e1, *e2;
    float* part;
  } bf;
  for (int attempt = 0; attempt < 2; ++attempt) {
    e->arenas[PT_ARENA_REC].reset();
    bool ok = true;
    auto take = [&](size_t elems) {
      void* p = e->arenas[PT_ARENA_REC].take(elems * m * sizeof(bf16_t));
      if (!p) ok = false;
      return reinterpret_cast<bf16_t*>(p);
    };
    const size_t N = (size_t)n;
    bf.a0 = take(N * 16 * 320 * 64);
    bf.a1 = take(N * 16 * 320 * 128);
    bf.p1 = take(N * 8 * 160 * 128);
    bf.c2a = take(N * 8 * 160 * 256);
    bf.c2b = take(N * 8 * 160 * 256);
    bf.p2 = take(N * 4 * 160 * 256);
    bf.c3a = take(N * 4 * 160 * 512);
    bf.c3b = take(N * 4 * 160 * 512);
    bf.p3 = take(N * 160 * 1024);
    bf.f = take(N * T * 512);
    bf.gx = take(N * T * 2048);
    bf.h = take(N * T * 512);
    bf.e1 = take(N * T * 256);
    bf.e2 = take(N * T * 512);
    void* pp = e->arenas[PT_ARENA_REC].take(N * T * NT * 2 * sizeof(float));
    if (!pp) ok = false;
    bf.part = reinterpret_cast<float*>(pp);
    if (ok) break;
    if (attempt == 1) {
      pt_set_error("activation arena allocation failed");
      return PT_ERR_HIP;
    }
    PT_HIP_CHECK(hipDeviceSynchronize());
    if (e->arenas[PT_ARENA_REC].base) PT_HIP_CHECK(hipFree(e->arenas[PT_ARENA_REC].base));
    e->arenas[PT_ARENA_REC].base = nullptr;
    const size_t want = e->arenas[PT_ARENA_REC].high + (1u << 20);
    PT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->arenas[PT_ARENA_REC].base), want));
    e->arenas[PT_ARENA_REC].cap = want;
  }

#define RUN(call) do { if ((rc = (call)) != PT_OK) return rc; } while (0)
  auto conv = [&](const bf16_t* in, int B, int hh, int ww, int cin, const ConvW& cw, int N, int ks, bf16_t* out, int relu) {
    ConvDesc c;
    c.in = in; c.B = B; c.H = hh; c.W = ww; c.Cin = cin; c.w = W(cw.w); c.bias = Bv(cw.b); c.N = N; c.ks = ks;
    c.stride = 1; c.out = out; c.out_cstride = N * m; c.relu = relu; c.split = x3; c.out_lo_off = N;
    return c;
  };
  // [rows, K] x [N, K]^T GEMMs of the sequence head: streaming kernel in bf16 mode (PT_CLS_FUSED=0 or hi/lo mode: 1x1 conv kernel)
  static int fused = -1;
  if (fused < 0) {
    const char* ev = getenv("PT_CLS_FUSED");
    fused = ev ? atoi(ev) : 1;
  }
  auto rows_gemm = [&](const bf16_t* in, int cin, const ConvW& cw, int N, bf16_t* out, int relu, const char* label) -> int {
    if (!x3 && fused && (cin == 512 || cin == 256)) {
      PtProfScope ps(e, s, PT_PROF_CONV1X1, 2.0 * n * T * (double)cin * N, label);
      return pt_launch_gemm_rows(in, (long long)n * T, cin, W(cw.w), Bv(cw.b), N, out, relu, s);
    }
    return pt_launch_conv(e, conv(in, 1, n, T, cin, cw, N, 1, out, relu), s);
  };
  {
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "crnn conv0+pool");
    RUN(pt_launch_crnn_conv0_pool(gray, n, PT_REC_H, PT_REC_W, Bv(c0w), Bv(c0b), x3, bf.a0, s));
  }
  // conv1 + pool(2,2) and conv2.3 + pool((2,1)): pooling in the conv epilogue (PT_POOL_FUSED=0: separate pool kernels)
  static int pool_fused = -1;
  if (pool_fused < 0) {
    const char* ev = getenv("PT_POOL_FUSED");
    pool_fused = ev ? atoi(ev) : 1;
  }
  if (pool_fused) {
    ConvDesc c1d = conv(bf.a0, n, 16, 320, 64, c1, 128, 3, bf.p1, 1);
    c1d.pool = 1;
    RUN(pt_launch_conv(e, c1d, s));
  } else {
    RUN(pt_launch_conv(e, conv(bf.a0, n, 16, 320, 64, c1, 128, 3, bf.a1, 1), s));
    RUN(pt_launch_maxpool_kxk(bf.a1, n, 16, 320, 128, 2, 2, 0, x3, bf.p1, s));
  }
  RUN(pt_launch_conv(e, conv(bf.p1, n, 8, 160, 128, c2a, 256, 3, bf.c2a, 1), s));
  if (pool_fused) {
    ConvDesc c2d = conv(bf.c2a, n, 8, 160, 256, c2b, 256, 3, bf.p2, 1);
    c2d.pool = 2;
    RUN(pt_launch_conv(e, c2d, s));
  } else {
    RUN(pt_launch_conv(e, conv(bf.c2a, n, 8, 160, 256, c2b, 256, 3, bf.c2b, 1), s));
    RUN(pt_launch_maxpool_kxk(bf.c2b, n, 8, 160, 256, 2, 1, 0, x3, bf.p2, s));
  }
  RUN(pt_launch_conv(e, conv(bf.p2, n, 4, 160, 256, c3a, 512, 3, bf.c3a, 1), s));
  if (pool_fused) {
    ConvDesc c3d = conv(bf.c3a, n, 4, 160, 512, c3b, 512, 3, bf.p3, 1);
    c3d.pool = 3;          // (2,1) pool, rows -> channel groups: [n][160][2 * 512]
    RUN(pt_launch_conv(e, c3d, s));
  } else {
    RUN(pt_launch_conv(e, conv(bf.c3a, n, 4, 160, 512, c3b, 512, 3, bf.c3b, 1), s));
    RUN(pt_launch_maxpool_kxk(bf.c3b, n, 4, 160, 512, 2, 1, /*h2c=*/1, x3, bf.p3, s));
  }
  // from here on: [1, n, 160, C] views
  RUN(pt_launch_conv(e, conv(bf.p3, 1, n, T, 1024, c4, 512, 1, bf.f, 1), s));
  RUN(rows_gemm(bf.f, 512, xp1, 2048, bf.gx, 0, "rows gemm 512->2048"));
  {
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "lstm1");
    RUN(pt_launch_lstm(e, bf.gx, W(whh1), bf.h, n, T, x3, s));
  }
  RUN(rows_gemm(bf.h, 512, em1, 256, bf.e1, 0, "rows gemm 512->256"));
  RUN(rows_gemm(bf.e1, 256, xp2, 2048, bf.gx, 0, "rows gemm 256->2048"));
  {
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "lstm2");
    RUN(pt_launch_lstm(e, bf.gx, W(whh2), bf.h, n, T, x3, s));
  }
  RUN(rows_gemm(bf.h, 512, em2, 512, bf.e2, 0, "rows gemm 512->512"));
  // classifier + arg-max: fused kernel in bf16 mode (PT_CLS_FUSED=0: tiled GEMM with per-tile partials + reduce, which is
  // also the hi/lo path)
  if (!x3 && fused) {
    PtProfScope ps(e, s, PT_PROF_CONV1X1, 2.0 * n * T * 512.0 * 7680.0, "classifier gemm+argmax");
    RUN(pt_launch_gemm_argmax(bf.e2, (long long)n * T, 512, W(cls.w), Bv(cls.b), 7680, ids, maxlogit, s));
  } else {
    {
      ConvDesc c = conv(bf.e2, 1, n, T, 512, cls, 7680, 1, nullptr, 0);
      c.argmax_part = bf.part;
      RUN(pt_launch_conv(e, c, s));
    }
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "argmax");
    RUN(pt_launch_argmax_reduce(bf.part, (long long)n * T, NT, ids, maxlogit, s));
  }
#undef RUN
  return PT_OK;
}
