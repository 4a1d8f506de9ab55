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

namespace PT_FMT_NS {

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

// element offsets (per channel multiplier m) of the single-line tensors inside pt_engine::rec_zero[precision]
namespace {
struct ZeroLine {
  static constexpr size_t GRAY = 0, A0 = GRAY + 32 * 640, P1 = A0 + 16 * 320 * 64, C2A = P1 + 8 * 160 * 128,
                          P2 = C2A + 8 * 160 * 256, C3A = P2 + 4 * 160 * 256, P3 = C3A + 4 * 160 * 512, F = P3 + 160 * 1024,
                          GX = F + 160 * 512, END = GX + 160 * 2048;
};
}  // namespace

int pt_crnn_forward_net(pt_engine* e, const bf16_t* gray, int n, int32_t* ids, float* maxlogit, hipStream_t s,
                        const pt_rec_line* d_lines) {
  PT_REQUIRE(e && gray && ids && n > 0, "crnn: bad arguments");
  auto it = e->models.find(PT_MODEL_CRNN);
  if (it == e->models.end()) {
    pt_set_error("CRNN weights not loaded (pt_weights_load(PT_MODEL_CRNN))");
    return PT_ERR_STATE;
  }
  if (!pt_model_format_ok(it->second, "PT_MODEL_CRNN")) return PT_ERR_STATE;
  const PtModel& M = it->second;
  const int x3 = pt_split(e) ? 1 : 0;
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
    const size_t want = pt_arena_round(e->arenas[PT_ARENA_REC].high);
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
  // hi/lo mode: the streaming three-pass kernel (gemm_rows_x3_kernel: the conv kernel's sums in its order; PT_ROWS_X3=0, read per call: the conv kernel)
  auto rows_x3 = [&]() {
    const char* ev = getenv("PT_ROWS_X3");
    return x3 && fused && !pt_f16x2(e) && !(ev && ev[0] == '0');
  };
  // single-pass modes: the pipelined GEMM of pt_launch_conv (gemm_pipe_kernel) takes these layers; PT_GEMM_PIPE=0 (read per call): the streaming kernel
  auto gemm_pipe = [&]() {
    const char* ev = getenv("PT_GEMM_PIPE");
    return !x3 && !(ev && ev[0] == '0');
  };
  auto rows_gemm = [&](const bf16_t* in, int cin, const ConvW& cw, int N, bf16_t* out, int relu, const char* label) -> int {
    if (!x3 && fused && (cin == 256 || (cin == 512 && !gemm_pipe()))) {      // (K = 256 stays on the streaming kernel: pt_launch_conv's rule)
      PtProfScope ps(e, s, PT_PROF_CONV1X1, 2.0 * n * T * (double)cin * N, label);
      return pt_launch_gemm_rows(in, (long long)n * T, cin, W(cw.w), Bv(cw.b), N, out, relu, s);
    }
    if (rows_x3() && (cin == 512 || cin == 256)) {
      char lb[48];
      snprintf(lb, sizeof(lb), "%s x3", label);
      PtProfScope ps(e, s, PT_PROF_CONV1X1, 2.0 * n * T * (double)cin * N, lb);
      return pt_launch_gemm_rows_x3(in, (long long)n * T, cin, W(cw.w), Bv(cw.b), N, out, relu, s);
    }
    return pt_launch_conv(e, conv(in, 1, n, T, cin, cw, N, 1, out, relu), s);
  };
  // conv0 .. conv3b (+ their pools) for nn lines; lim != null: no work right of every line's text (see crnn_limits_kernel),
  // the skipped columns are filled from the all-padding line's activations `zl`
  int pool_fused = 1;      // (read per call: tests switch it in process)
  {
    const char* ev = getenv("PT_POOL_FUSED");
    if (ev) pool_fused = atoi(ev);
  }
  auto conv_stack = [&](const bf16_t* g, int nn, bf16_t* a0, bf16_t* a1, bf16_t* p1, bf16_t* c2a_o, bf16_t* c2b_o, bf16_t* p2,
                        bf16_t* c3a_o, bf16_t* c3b_o, bf16_t* p3, bf16_t* f_o, bf16_t* gx_o, const PtCrnnLimits* lim,
                        const bf16_t* zl) -> int {
    auto limited = [&](ConvDesc c, int k) {
      if (lim) {
        c.xlimit = lim->lim[k]; c.xlimit_cols = lim->cols + k;
        if (k == 3) c.block_list = lim->blist3a;      // conv3.* (4-row maps): pairs of live 32-column blocks per workgroup
        if (k == 4) c.block_list = lim->glist;
        if (k == 1) c.block_list = lim->blist2a;      // conv2.* (8-row maps): taken by the blocked DMA kernel only (pt_launch_conv)
        if (k == 2) c.block_list = lim->blist2b;
      }
      return c;
    };
    // fill the columns layer k left, up to the last column the NEXT limited layer (limit kn, tile tn) reads; kn < 0: to the end
    auto fill = [&](bf16_t* out, size_t zoff, int k, int tile_w, int div, int rows, int Wd, int C, int kn, int tn) -> int {
      if (!lim) return PT_OK;
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "crnn fill");
      return pt_launch_crnn_fill(out, zl + zoff * m, lim->lim[k], tile_w, div, nn, rows, Wd, C * m, s, kn >= 0 ? lim->lim[kn] : nullptr, tn);
    };
    // single-pass modes: conv0 + pool + conv1 + pool in ONE launch, the 64-channel map never leaves the CU (crnn_conv01_kernel; PT_CONV01=0, read per
    // call: the two launches below -- same values, tests/test_gpu_rec.py)
    const char* ev01 = getenv("PT_CONV01");
    const bool conv01 = pool_fused && !x3 && !(ev01 && atoi(ev01) == 0);
    if (conv01) {
      RUN(pt_launch_crnn_conv01(e, g, nn, Bv(c0w), Bv(c0b), W(c1.w), Bv(c1.b), p1, lim ? lim->lim[0] : nullptr, lim ? lim->cols + 0 : nullptr, s));
      RUN(fill(p1, ZeroLine::P1, 0, 32, 2, 8, 160, 128, 1, 32));
    } else {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "crnn conv0+pool");
      RUN(pt_launch_crnn_conv0_pool(g, nn, PT_REC_H, PT_REC_W, Bv(c0w), Bv(c0b), x3, a0, s, lim ? lim->lim[5] : nullptr));
    }
    // conv1 + pool(2,2) and conv2.3 + pool((2,1)): pooling in the conv epilogue (PT_POOL_FUSED=0: separate pool kernels,
    // which need the full maps: no column limits then)
    if (conv01) {
    } else if (pool_fused) {
      ConvDesc c1d = limited(conv(a0, nn, 16, 320, 64, c1, 128, 3, p1, 1), 0);
      c1d.pool = 1;
      RUN(pt_launch_conv(e, c1d, s));
      RUN(fill(p1, ZeroLine::P1, 0, 32, 2, 8, 160, 128, 1, 32));
    } else {
      RUN(pt_launch_conv(e, conv(a0, nn, 16, 320, 64, c1, 128, 3, a1, 1), s));
      RUN(pt_launch_maxpool_kxk(a1, nn, 16, 320, 128, 2, 2, 0, x3, p1, s));
    }
    RUN(pt_launch_conv(e, limited(conv(p1, nn, 8, 160, 128, c2a, 256, 3, c2a_o, 1), 1), s));
    RUN(fill(c2a_o, ZeroLine::C2A, 1, 32, 1, 8, 160, 256, 2, 32));
    if (pool_fused) {
      ConvDesc c2d = limited(conv(c2a_o, nn, 8, 160, 256, c2b, 256, 3, p2, 1), 2);
      c2d.pool = 2;
      RUN(pt_launch_conv(e, c2d, s));
      RUN(fill(p2, ZeroLine::P2, 2, 32, 1, 4, 160, 256, 3, 64));
    } else {
      RUN(pt_launch_conv(e, conv(c2a_o, nn, 8, 160, 256, c2b, 256, 3, c2b_o, 1), s));
      RUN(pt_launch_maxpool_kxk(c2b_o, nn, 8, 160, 256, 2, 1, 0, x3, p2, s));
    }
    RUN(pt_launch_conv(e, limited(conv(p2, nn, 4, 160, 256, c3a, 512, 3, c3a_o, 1), 3), s));
    RUN(fill(c3a_o, ZeroLine::C3A, 3, 32, 1, 4, 160, 512, 4, 64));      // (conv3.* multiply 32-column row-tiles up to the limit inside their 64-column patches)
    if (pool_fused) {
      ConvDesc c3d = limited(conv(c3a_o, nn, 4, 160, 512, c3b, 512, 3, p3, 1), 4);
      c3d.pool = 3;          // (2,1) pool, rows -> channel groups: [n][160][2 * 512]
      RUN(pt_launch_conv(e, c3d, s));      // p3 needs no fill: conv4 below is per position and limited the same way
    } else {
      RUN(pt_launch_conv(e, conv(c3a_o, nn, 4, 160, 512, c3b, 512, 3, c3b_o, 1), s));
      RUN(pt_launch_maxpool_kxk(c3b_o, nn, 4, 160, 512, 2, 1, /*h2c=*/1, x3, p3, s));
    }
    // conv4 ((2,1) kernel = 1x1 GEMM over [1, nn, 160, 1024]) and the first LSTM's input projection: per time step, so right of
    // the text they give the all-padding line's values too; computed in 32-step groups up to the conv3b limit, and only the
    // projection's output -- which the recurrence reads at every step -- is filled
    {
      ConvDesc c4d = conv(p3, 1, nn, T, 1024, c4, 512, 1, f_o, 1);
      if (lim) { c4d.xlimit_rows = lim->lim[4]; c4d.xlimit_cols = lim->cols + 5; c4d.block_list = lim->glist; }      // (the list: gemm_pipe_kernel's live row groups)
      RUN(pt_launch_conv(e, c4d, s));
    }
    if ((!x3 && fused && !gemm_pipe()) || rows_x3()) {
      int lim_slot = -1;
      {
        PtProfScope ps(e, s, PT_PROF_CONV1X1, 2.0 * nn * T * 512.0 * 2048, x3 ? "rows gemm 512->2048 x3" : "rows gemm 512->2048");
        if (x3) RUN(pt_launch_gemm_rows_x3(f_o, (long long)nn * T, 512, W(xp1.w), Bv(xp1.b), 2048, gx_o, 0, s, lim ? lim->glist : nullptr));
        else RUN(pt_launch_gemm_rows(f_o, (long long)nn * T, 512, W(xp1.w), Bv(xp1.b), 2048, gx_o, 0, s, lim ? lim->glist : nullptr));
        if (lim && ps.idx >= 0 && e->prof.h_lims && e->prof.n_lims < PtProfile::MAX_LIMS) {      // credit the executed row groups, like the limited convs
          auto& pd = e->prof.pending[ps.idx];
          pd.lim_slot = lim_slot = e->prof.n_lims++;
          pd.rows = nn * T;
        }
      }
      if (lim_slot >= 0) (void)hipMemcpyAsync(e->prof.h_lims + lim_slot, lim->cols + 5, sizeof(int), hipMemcpyDeviceToHost, s);
    } else {
      ConvDesc xd = conv(f_o, 1, nn, T, 512, xp1, 2048, 1, gx_o, 0);
      if (lim) { xd.xlimit_rows = lim->lim[4]; xd.xlimit_cols = lim->cols + 5; xd.block_list = lim->glist; }
      RUN(pt_launch_conv(e, xd, s));
    }
    RUN(fill(gx_o, ZeroLine::GX, 4, 32, 1, 1, 160, 2048, -1, 1));
    return PT_OK;
  };
  // ragged path: the lines' sizes are known, the pools are fused (a limited conv never writes the full-resolution map)
  const bool ragged = d_lines != nullptr && pool_fused && e->rec_ragged;
  PtCrnnLimits lim;
  const bf16_t* zl = nullptr;
  if (ragged) {
    if (!e->rec_zero_valid[x3]) {        // once per (weights, precision): an all-padding line through the conv stack, in full
      const size_t bytes = ZeroLine::END * m * sizeof(bf16_t);
      if (!e->rec_zero[x3]) PT_HIP_CHECK(hipMalloc(&e->rec_zero[x3], bytes));
      PT_HIP_CHECK(hipMemsetAsync(e->rec_zero[x3], 0, bytes, s));
      bf16_t* z = reinterpret_cast<bf16_t*>(e->rec_zero[x3]);
      // full-resolution conv outputs are not kept with fused pools: a1 / c2b / c3b are unused there
      RUN(conv_stack(z + ZeroLine::GRAY * m, 1, z + ZeroLine::A0 * m, nullptr, z + ZeroLine::P1 * m, z + ZeroLine::C2A * m, nullptr,
                     z + ZeroLine::P2 * m, z + ZeroLine::C3A * m, nullptr, z + ZeroLine::P3 * m, z + ZeroLine::F * m,
                     z + ZeroLine::GX * m, nullptr, nullptr));
      e->rec_zero_valid[x3] = true;
      // later calls may run on ANOTHER stream (the pipeline's overlap recogniser): they wait for this build through the event
      if (!e->rec_zero_ready[x3]) PT_HIP_CHECK(hipEventCreateWithFlags(&e->rec_zero_ready[x3], hipEventDisableTiming));
      PT_HIP_CHECK(hipEventRecord(e->rec_zero_ready[x3], s));
    } else if (e->rec_zero_ready[x3]) {
      PT_HIP_CHECK(hipStreamWaitEvent(s, e->rec_zero_ready[x3], 0));       // a no-op once the build has completed
    }
    zl = reinterpret_cast<const bf16_t*>(e->rec_zero[x3]);
    const size_t need = ((size_t)26 * n + 40) * sizeof(int);
    if (need > e->rec_limits_cap) {
      PT_HIP_CHECK(hipStreamSynchronize(s));
      if (e->rec_limits) PT_HIP_CHECK(hipFree(e->rec_limits));
      e->rec_limits = nullptr;
      PT_HIP_CHECK(hipMalloc(&e->rec_limits, need * 2));
      e->rec_limits_cap = need * 2;
    }
    int* base = reinterpret_cast<int*>(e->rec_limits);
    for (int k = 0; k < 6; ++k) lim.lim[k] = base + (size_t)k * n;
    lim.cols = base + (size_t)6 * n;
    lim.glist = base + (size_t)6 * n + 8;
    lim.blist3a = base + (size_t)11 * n + 16;
    lim.blist2a = base + (size_t)16 * n + 24;
    lim.blist2b = base + (size_t)21 * n + 32;
    {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "crnn limits");
      RUN(pt_launch_crnn_limits(d_lines, n, lim, s));
      RUN(pt_launch_rows_live_list(lim.lim[4], n, lim.glist, s));
      RUN(pt_launch_rows_live_list(lim.lim[3], n, lim.blist3a, s));
      RUN(pt_launch_rows_live_list(lim.lim[1], n, lim.blist2a, s));
      RUN(pt_launch_rows_live_list(lim.lim[2], n, lim.blist2b, s));
    }
  }
  RUN(conv_stack(gray, n, bf.a0, bf.a1, bf.p1, bf.c2a, bf.c2b, bf.p2, bf.c3a, bf.c3b, bf.p3, bf.f, bf.gx, ragged ? &lim : nullptr, zl));
  // from here on everything runs over all 160 steps of every line: the recurrences make padded steps line-dependent
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
  // (PT_CLS_PIPE=1, read per call: the pipelined GEMM with the arg-max in its register epilogue + the partial reduce -- the conv path below.  Measured
  // SLOWER for this shape, 6.87 + 1.57 ms against the streaming kernel's 5.02: with N = 7680 a row tile is sixty 8-slice tiles that re-stream A and
  // write 780 MB of partial maxima, where the streaming kernel keeps a wave's rows in registers and sweeps W once)
  const char* cp_ev = getenv("PT_CLS_PIPE");
  const bool cls_pipe = gemm_pipe() && cp_ev && cp_ev[0] == '1';
  if (!x3 && fused && !cls_pipe) {
    PtProfScope ps(e, s, PT_PROF_CONV1X1, 2.0 * n * T * 512.0 * 7680.0, "classifier gemm+argmax");
    RUN(pt_launch_gemm_argmax(bf.e2, (long long)n * T, 512, W(cls.w), Bv(cls.b), 7680, ids, maxlogit, s));
  } else if (x3 && fused && !pt_f16x2(e) && !(getenv("PT_CLS_X3_REFINE") && atoi(getenv("PT_CLS_X3_REFINE")) == 0)) {
    // hi/lo mode: two single-pass sweeps (maximum, then the classes within the rounding bound of it) + exact logits of those
    // candidates (rec_kernels.hip: gemm_cand_kernel / cand_eval_kernel); PT_CLS_X3_REFINE=0: the tiled three-pass GEMM (A/B switch)
    PtProfScope ps(e, s, PT_PROF_CONV1X1, 2.0 * n * T * 512.0 * 7680.0, "classifier bound+refine x3");
    RUN(pt_launch_gemm_argmax_x3(bf.e2, (long long)n * T, 512, W(cls.w), Bv(cls.b), 7680, 7680, ids, maxlogit, bf.part, s));
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

}  // namespace PT_FMT_NS
