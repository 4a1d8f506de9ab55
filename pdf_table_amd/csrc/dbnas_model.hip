// dbnas_model.hip -- launch graph of the DB-ProxylessNAS text detector (`DBNasModel`, db_net/dbnet.py:693-712):
// CompactDetBackbone (db_net/proxyless.py:92-178, searched block list :118-127) -> LightSegDetector (dbnet.py:338-481),
// eval branch (`binary` only).
//
// Engine mapping (weights.pack_db_nas folds every BatchNorm):
//   * first_conv 3x3 s2 (3 -> 32) + ReLU: direct VALU kernel;
//   * each inverted-residual block = 1x1 expand on the MFMA kernel (PReLU epilogue), ONE depthwise k x k kernel -- the
//     `rep` blocks' sum of 1x1 / 3x3 / 5x5 depthwise branches (layers.py:732-745) is a single 5x5 depthwise conv after
//     folding, the same re-parameterisation the reference's own `deploy` switch stands for -- with PReLU, and the 1x1
//     projection on the MFMA kernel with the identity shortcut added in its epilogue;
//   * SE blocks: two-level average pool, gate = 1 + sigmoid(..) (the block's identity shortcut folded in), scale;
//   * decoder: the four 1x1 lateral convs run coarse to fine, each adding the nearest-x2 up-sampled running sum in its
//     epilogue (up8(a) + up4(b) + up2(c) + d == up2(up2(up2(a) + b) + c) + d for nearest up-sampling); depthwise 5x5 +
//     ReLU; 1x1 64 -> 16 + ReLU on the MFMA kernel; everything after that in dbnas_tail_kernel.
// Channel counts are stored padded to multiples of 64 (32 -> 64, 96 -> 128) with zero weights, so that a residual has
// the GEMM's own channel stride; PReLU(0) = 0 and x * gate keep the padding zero.
#include <string>

#include "common.h"

namespace PT_FMT_NS {

namespace {

struct T {
  bf16_t* p = nullptr;
  int H = 0, W = 0, C = 0;
};

// kind (0 conv block, 1 SE), cin, cout, expand ratio, depthwise kernel after folding, stride, SE squeeze factor
// proxyless.py:107-127: conv_candidates[conv_op_ids[i]] per block; '135'/'35' RepConv fold to k = 5
struct Blk { int kind, cin, cout, expand, k, stride, squeeze; };
const Blk kBlocks[24] = {
    {0, 32, 32, 2, 5, 2, 0},  {0, 32, 32, 2, 5, 1, 0},  {0, 32, 32, 2, 5, 1, 0},  {0, 32, 32, 2, 5, 1, 0},  {0, 32, 32, 2, 5, 1, 0},
    {1, 32, 32, 0, 0, 1, 2},
    {0, 32, 64, 4, 5, 2, 0},  {0, 64, 64, 4, 5, 1, 0},  {0, 64, 64, 4, 5, 1, 0},  {0, 64, 64, 4, 5, 1, 0},  {0, 64, 64, 4, 5, 1, 0},
    {1, 64, 64, 0, 0, 1, 8},
    {0, 64, 96, 4, 5, 2, 0},  {0, 96, 96, 4, 5, 1, 0},  {0, 96, 96, 4, 5, 1, 0},  {0, 96, 96, 4, 5, 1, 0},  {0, 96, 96, 4, 5, 1, 0},
    {1, 96, 96, 0, 0, 1, 8},
    {0, 96, 128, 4, 5, 2, 0}, {0, 128, 128, 4, 5, 1, 0}, {0, 128, 128, 4, 5, 1, 0}, {0, 128, 128, 4, 5, 1, 0}, {0, 128, 128, 4, 5, 1, 0},
    {1, 128, 128, 0, 0, 1, 8}};

inline int pad64(int c) { return (c + 63) / 64 * 64; }

struct Ctx {
  pt_engine* e;
  const PtModel* m;
  hipStream_t s;
  int n, x3, mul;
  bool dry, ok;
  int rc;

  T alloc(int H, int W, int C) {
    T t;
    t.H = H; t.W = W; t.C = C;
    t.p = reinterpret_cast<bf16_t*>(e->arenas[PT_ARENA_DET].take((size_t)n * H * W * C * mul * sizeof(bf16_t)));
    if (!t.p) ok = false;
    return t;
  }
  const PtTensor* get(const std::string& name) {
    const PtTensor* t = m->find(name);
    if (!t && rc == PT_OK) {
      pt_set_error("DB-NAS weight blob lacks tensor '%s'", name.c_str());
      rc = PT_ERR_FORMAT;
    }
    return t;
  }
  bool go() const { return rc == PT_OK && !dry && ok; }
  const float* F(const PtTensor* t) { return reinterpret_cast<const float*>(t->d_ptr); }

  // 1x1 conv (+ folded BN); act 0 none / 1 ReLU / 3 PReLU(slope tensor q.slope); res_mode 1: + res, 2: + nearest-x2(res)
  void pw(const T& in, const std::string& q, int N, const T& out, int act, const T* res = nullptr, int res_mode = 1, int nv = 0) {
    const PtTensor* w = get(q + (x3 ? ".w3" : ".w"));
    const PtTensor* b = get(q + ".b");
    const PtTensor* sl = act == 3 ? get(q + ".slope") : nullptr;
    if (!go()) return;
    ConvDesc c;
    c.in = in.p; c.B = n; c.H = in.H; c.W = in.W; c.Cin = in.C;
    c.w = reinterpret_cast<const bf16_t*>(w->d_ptr); c.bias = F(b);
    c.N = N; c.ks = 1; c.stride = 1; c.relu = act; c.split = x3; c.n_valid = nv;
    if (sl) c.slope = F(sl);
    c.out = out.p; c.out_cstride = out.C * mul; c.out_lo_off = out.C;
    if (res) { c.res = res->p; c.res_mode = res_mode; }
    const int r = pt_launch_conv(e, c, s);
    if (r != PT_OK) rc = r;
  }
  T dw(const T& in, const std::string& q, int k, int stride, int act) {
    const int pad = k / 2;
    T o = alloc((in.H + 2 * pad - k) / stride + 1, (in.W + 2 * pad - k) / stride + 1, in.C);
    const PtTensor* w = get(q + ".wf32");
    const PtTensor* b = get(q + ".b");
    const PtTensor* sl = act == 3 ? get(q + ".slope") : nullptr;
    if (go()) {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "dbnas dwconv");
      const int r = pt_launch_dwconv(in.p, F(w), F(b), o.p, n, in.H, in.W, in.C, k, stride, act, x3, s, sl ? F(sl) : nullptr);
      if (r != PT_OK) rc = r;
    }
    return o;
  }
};

}  // namespace

// x: NHWC4 bf16 [n, H, W, 4] (8 channels in BF16X3 mode); prob / logits: fp32 [n, H, W] (either may be null)
int pt_dbnas_forward_net(pt_engine* e, const bf16_t* x, int n, int H, int W, float* prob, float* logits, hipStream_t s) {
  PT_REQUIRE(x && (prob || logits) && n > 0, "DB-NAS net: null pointer");
  PT_REQUIRE(H % 32 == 0 && W % 32 == 0 && H > 0 && W > 0, "DB-NAS net: input %dx%d must be multiples of 32", H, W);
  auto it = e->models.find(PT_MODEL_DB_NAS);
  if (it == e->models.end()) {
    pt_set_error("DB-NAS weights not loaded (pt_weights_load(PT_MODEL_DB_NAS))");
    return PT_ERR_STATE;
  }
  if (!pt_model_format_ok(it->second, "PT_MODEL_DB_NAS")) return PT_ERR_STATE;
  Ctx c;
  c.e = e; c.m = &it->second; c.s = s; c.n = n;
  c.x3 = pt_split(e) ? 1 : 0;
  c.mul = c.x3 ? 2 : 1;
  c.rc = PT_OK;
  for (int pass = 0; pass < 2; ++pass) {
    c.dry = pass == 0;
    c.ok = true;
    e->arenas[PT_ARENA_DET].reset();
    float* gate = reinterpret_cast<float*>(e->arenas[PT_ARENA_DET].take((size_t)n * 512 * sizeof(float)));
    float* part = reinterpret_cast<float*>(e->arenas[PT_ARENA_DET].take((size_t)n * PT_SE_CHUNKS * 512 * sizeof(float)));
    if (!gate || !part) c.ok = false;
    T t = c.alloc(H / 2, W / 2, 64);
    {
      const PtTensor* w = c.get("stem.wf32");
      const PtTensor* b = c.get("stem.b");
      if (c.go()) {
        PtProfScope ps(e, s, PT_PROF_STEM, 0, "dbnas stem3x3");
        const int r = pt_launch_stem3x3s2(x, c.F(w), c.F(b), t.p, n, H, W, c.x3, s, 1);
        if (r != PT_OK) c.rc = r;
      }
    }
    T feats[4];
    for (int i = 0; i < 24; ++i) {
      const Blk& b = kBlocks[i];
      const std::string q = "b" + std::to_string(i);
      if (b.kind == 1) {
        const PtTensor *w1 = c.get(q + ".se.w1"), *b1 = c.get(q + ".se.b1"), *w2 = c.get(q + ".se.w2"), *b2 = c.get(q + ".se.b2");
        T g = c.alloc(t.H, t.W, t.C);
        if (c.go()) {
          PtProfScope ps(e, s, PT_PROF_OTHER, 0, "dbnas SE");
          const int r = pt_launch_se(t.p, c.F(w1), c.F(b1), c.F(w2), c.F(b2), gate, g.p, n, t.H * t.W, t.C, c.x3, s,
                                     b.cin / b.squeeze, 1, part);
          if (r != PT_OK) c.rc = r;
        }
        t = g;
        feats[i / 6] = t;       // NasRecBackbone.forward: outputs after blocks 5, 11, 17, 23 (proxyless.py:21-31)
        continue;
      }
      const int mid = b.cin * b.expand, coutp = pad64(b.cout);
      T ex = c.alloc(t.H, t.W, mid);
      c.pw(t, q + ".exp", mid, ex, 3);
      T d = c.dw(ex, q + ".dw", b.k, b.stride, 3);
      T o = c.alloc(d.H, d.W, coutp);
      const bool shortcut = b.stride == 1 && b.cin == b.cout;
      c.pw(d, q + ".proj", coutp, o, 0, shortcut ? &t : nullptr, 1);
      t = o;
    }
    // ---- LightSegDetector (dbnet.py:458-469)
    T acc = c.alloc(feats[3].H, feats[3].W, 64);
    c.pw(feats[3], "dec.in5", 64, acc, 0);
    static const char* lat[3] = {"dec.in4", "dec.in3", "dec.in2"};
    for (int l = 2; l >= 0; --l) {
      PT_REQUIRE(feats[l].H == 2 * acc.H && feats[l].W == 2 * acc.W, "DB-NAS net: feature maps are not exact halves");
      T nx = c.alloc(feats[l].H, feats[l].W, 64);
      c.pw(feats[l], lat[2 - l], 64, nx, 0, &acc, 2);
      acc = nx;
    }
    T d = c.dw(acc, "dec.dw", 5, 1, 1);
    T y16 = c.alloc(d.H, d.W, 16);
    c.pw(d, "dec.pw", 64, y16, 1, nullptr, 1, 16);
    {
      const PtTensor* tw = c.get("dec.tail");
      if (c.go()) {
        PtProfScope ps(e, s, PT_PROF_OTHER, 0, "dbnas tail");
        const int r = pt_launch_dbnas_tail(y16.p, c.F(tw), n, y16.H, y16.W, c.x3, prob, logits, s);
        if (r != PT_OK) c.rc = r;
      }
    }
    if (c.rc != PT_OK) return c.rc;
    if (pass == 0) {
      if (c.ok) continue;
      PT_HIP_CHECK(hipDeviceSynchronize());
      if (e->arenas[PT_ARENA_DET].base) PT_HIP_CHECK(hipFree(e->arenas[PT_ARENA_DET].base));
      e->arenas[PT_ARENA_DET].base = nullptr;
      const size_t want = pt_arena_round(e->arenas[PT_ARENA_DET].high);
      PT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->arenas[PT_ARENA_DET].base), want));
      e->arenas[PT_ARENA_DET].cap = want;
      continue;
    }
    if (!c.ok) {
      pt_set_error("DB-NAS net: activation arena allocation failed");
      return PT_ERR_HIP;
    }
  }
  return PT_OK;
}

}  // namespace PT_FMT_NS
