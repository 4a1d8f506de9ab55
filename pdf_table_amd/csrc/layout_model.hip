// layout_model.hip -- launch graphs over the LCNet x1.0 backbone: the PicoDet layout detector (LCNet -> 4-level CSP-PAN ->
// PicoHead) and the PP-LCNet image classifiers (`PPLCNet`, model/cls/cls_pp_lcnet.py:164-283: the same backbone -> global
// average pool -> 1x1 conv 512 -> 1280 + hardswish -> Linear; text-line variants run the first block of blocks3..6 with
// stride (2, 1), configuration_cls_pulc.py:20-39).
//
// Reference graph: LCNet.forward picodet/lcnet.py:241-257, CSPPAN.forward csp_pan.py:305-345, PicoHead.forward_eval
// pico_head.py:1108-1160 (export_post_process=False: per level sigmoid scores and raw box-distribution logits, which is
// what the ONNX export the reference runs hands to OCRPicodetPostProcessor, ocr_layout_task.py:159-175).
// Engine mapping: every Conv+BN folded; 1x1 convs on the MFMA kernel with a hardswish epilogue; depthwise convs, the
// 3-channel stem, SE gates and the one tensor add are bandwidth kernels (layout_kernels.hip).  A 1x1 conv over
// `cat(a, b)` is two GEMMs accumulating through the residual path; when `a` is a nearest x2 up-sample the first GEMM
// runs at the low resolution and replicates its stores (1x1 conv and nearest up-sampling commute).
// 16-channel tensors are stored 32 wide with a zero upper half.  Head outputs: fp32 [n, A_level, 40] (ncls class logits,
// then 4 x (reg_max + 1) box logits, padded to 40).
#include <stdlib.h>

#include <string>
#include <vector>

#include "common.h"

namespace PT_FMT_NS {

namespace {

struct T {
  bf16_t* p = nullptr;
  int H = 0, W = 0, C = 0;
};

struct Ctx {
  pt_engine* e;
  const PtModel* m;
  hipStream_t s;
  int n, x3, mul;
  bool dry, ok;
  int rc;
  float* gate = nullptr;
  float* part = nullptr;      // PT_SE_CHUNKS * n * 512 floats: two-level average pool of the SE blocks (null: single-workgroup scan)
  const char* what = "PicoDet";

  T alloc(int H, int W, int C) {
    T t;
    t.H = H; t.W = W; t.C = C;
    t.p = reinterpret_cast<bf16_t*>(e->arenas[PT_ARENA_LAYOUT].take((size_t)n * H * W * C * mul * sizeof(bf16_t)));
    if (!t.p) ok = false;
    return t;
  }
  const PtTensor* get(const std::string& name) {
    const PtTensor* t = m->find(name);
    if (!t && rc == PT_OK) {
      pt_set_error("%s weight blob lacks tensor '%s'", what, name.c_str());
      rc = PT_ERR_FORMAT;
    }
    return t;
  }
  bool go() const { return rc == PT_OK && !dry && ok; }
  const float* F(const PtTensor* t) { return reinterpret_cast<const float*>(t->d_ptr); }

  // 1x1 conv (+ folded BN) with activation act (0 none, 2 hardswish); rep: nearest replicate factor of the stores;
  // res: added before the activation; nv: stored channels; out_f32: fp32 [.., f32_cs] output
  void pw(const T& in, const std::string& q, int N, const T& out, int act, const T* res = nullptr, int rep = 1, int nv = 0,
          float* out_f32 = nullptr, int f32_cs = 0) {
    const PtTensor* w = get(q + (x3 ? ".w3" : ".w"));
    const PtTensor* b = get(q + ".b");
    if (!go()) return;
    ConvDesc c;
    c.in = in.p; c.B = n; c.H = in.H; c.W = in.W; c.Cin = in.C;
    c.w = reinterpret_cast<const bf16_t*>(w->d_ptr); c.bias = F(b);
    c.N = N; c.ks = 1; c.stride = 1; c.relu = act; c.split = x3; c.n_valid = nv; c.rep = rep;
    if (out_f32) {
      c.out_f32 = out_f32; c.out_cstride = f32_cs;
    } else {
      c.out = out.p; c.out_cstride = out.C * mul; c.out_lo_off = out.C;
    }
    if (res) { c.res = res->p; c.res_mode = 1; }
    const int r = pt_launch_conv(e, c, s);
    if (r != PT_OK) rc = r;
  }
  T dw(const T& in, const std::string& q, int k, int stride, int act) {     // stride: s or (sy << 8) | sx
    const int pad = k / 2, sy = stride > 255 ? stride >> 8 : stride, sx = stride > 255 ? stride & 255 : stride;
    T o = alloc((in.H + 2 * pad - k) / sy + 1, (in.W + 2 * pad - k) / sx + 1, in.C);
    const PtTensor* w = get(q + ".wf32");
    const PtTensor* b = get(q + ".b");
    if (go()) {
      e->prof.next_bytes = 2.0 * (x3 ? 2 : 1) * n * ((double)in.H * in.W + (double)o.H * o.W) * in.C;
      char label[48];
      snprintf(label, sizeof(label), "layout dwconv k%d s%d %d @%dx%d", k, stride, in.C, o.H, o.W);
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, label);
      const int r = pt_launch_dwconv(in.p, F(w), F(b), o.p, n, in.H, in.W, in.C, k, stride, act, x3, s, nullptr);
      if (r != PT_OK) rc = r;
    }
    return o;
  }
  // depthwise k x k (stride 1) + pointwise 1x1 as ONE launch (conv_igemm.hip: dwpw_kernel) in the single-pass modes; false: not taken (pair mode, stride 2,
  // more than 256 outputs, PT_DWPW=0 -- read per call), the caller runs the two launches
  bool dwpw(const T& in, const std::string& qd, int k, int stride, int act_dw, const std::string& qp, int N, const T& out, int act_pw, int nv = 0) {
    // Measured (MI355X, 64 pages, profiles/r06/dwpw_ab.txt): the fused launch wins where the depthwise conv is 3 x 3 (32 -> 32 @400x304: 0.74 -> 0.61 ms,
    // 64 -> 64 @200x152: 0.32 -> 0.25, 128 -> 128 @100x76: 0.18 -> 0.16) and LOSES at 5 x 5 (256 -> 256 @50x38 x 5: 0.81 -> 1.15 ms, 128 -> 128 @100x76 x 4:
    // 0.94 -> 1.02): 25 taps per value are VALU / LDS work that two workgroups of 210-256 registers per CU hide worse than the depthwise tile kernel's
    // four to five, and these maps sit in the Infinity Cache -- the traffic the fusion removes was not what the pair waited for.  PT_DWPW=0: never,
    // PT_DWPW=2: every stride-1 pair (A/B switch, read per call)
    const char* ev = getenv("PT_DWPW");
    const int mode = ev ? atoi(ev) : 1;
    if (x3 || stride != 1 || N > 256 || mode == 0 || (mode == 1 && k != 3)) return false;
    const PtTensor* dww = get(qd + ".wf32");
    const PtTensor* dwb = get(qd + ".b");
    const PtTensor* w = get(qp + ".w");
    const PtTensor* b = get(qp + ".b");
    if (!go()) return true;
    ConvDesc c;
    c.B = n; c.H = in.H; c.W = in.W; c.Cin = in.C;
    c.w = reinterpret_cast<const bf16_t*>(w->d_ptr); c.bias = F(b);
    c.N = N; c.ks = 1; c.stride = 1; c.relu = act_pw; c.n_valid = nv;
    c.out = out.p; c.out_cstride = out.C; c.out_lo_off = out.C;
    const int r = pt_launch_dwpw(e, in.p, n, in.H, in.W, in.C, F(dww), F(dwb), k, 1, act_dw, c, s);
    if (r != PT_OK) rc = r;
    return true;
  }
  // DPModule (csp_pan.py:56-105): dw k5 + BN + hswish, pw + BN + hswish
  T dp(const T& in, const std::string& q, int stride) {
    if (stride == 1 && !x3) {
      T o = alloc(in.H, in.W, in.C);
      if (dwpw(in, q + ".dw", 5, 1, 2, q + ".pw", in.C, o, 2)) return o;
    }
    T d = dw(in, q + ".dw", 5, stride, 2);
    T o = alloc(d.H, d.W, in.C);
    pw(d, q + ".pw", in.C, o, 2);
    return o;
  }
  // hswish(bn(conv1x1(cat(a', b)))) where a' = a (up == 1) or nearest-x2(a) (up == 2)
  T cat_pw(const T& a, const T& b, const std::string& q, int N, int up) {
    T o = alloc(b.H, b.W, N);
    pw(a, q + ".a", N, o, 0, nullptr, up);
    pw(b, q + ".b", N, o, 2, &o);
    return o;
  }
  // CSPLayer (csp_pan.py:160-209) on cat(a', b)
  T csp(const T& a, const T& b, const std::string& q, int up) {
    T sh = cat_pw(a, b, q + ".short", 64, up);
    T mn = cat_pw(a, b, q + ".main", 64, up);
    T c1 = alloc(mn.H, mn.W, 64);
    pw(mn, q + ".conv1", 64, c1, 2);
    T d = dp(c1, q + ".dp", 1);
    return cat_pw(d, sh, q + ".final", 128, 1);
  }
};

// LCNet x1.0 (picodet/lcnet.py:241-257 == cls_pp_lcnet.py:262-273): conv1 3x3 s2 + 13 depthwise-separable blocks.
// stage_stride[i]: stride of the first block of blocks3 + i -- 2, or (sy << 8) | sx.  feats: outputs of blocks4, 5, 6.
T lcnet_backbone(Ctx& c, const bf16_t* x, int H, int W, const int* stage_stride, T* feats) {
  // k, cin, cout, stage index of a stage's first block (-1: stride 1), se -- lcnet.py:25-46 / cls_pp_lcnet.py:54-66
  static const int cfg[][5] = {{3, 16, 32, -1, 0},
                               {3, 32, 64, 0, 0}, {3, 64, 64, -1, 0},
                               {3, 64, 128, 1, 0}, {3, 128, 128, -1, 0},
                               {3, 128, 256, 2, 0}, {5, 256, 256, -1, 0}, {5, 256, 256, -1, 0}, {5, 256, 256, -1, 0},
                               {5, 256, 256, -1, 0}, {5, 256, 256, -1, 0},
                               {5, 256, 512, 3, 1}, {5, 512, 512, -1, 1}};
  static const char* names[] = {"blocks2.0", "blocks3.0", "blocks3.1", "blocks4.0", "blocks4.1", "blocks5.0", "blocks5.1",
                                "blocks5.2", "blocks5.3", "blocks5.4", "blocks5.5", "blocks6.0", "blocks6.1"};
  pt_engine* e = c.e;
  hipStream_t s = c.s;
  const int n = c.n;
  T t = c.alloc((H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1, 32);
  {
    const PtTensor* w = c.get("stem.wf32");
    const PtTensor* b = c.get("stem.b");
    if (c.go()) {
      PtProfScope ps(e, s, PT_PROF_STEM, 0, "lcnet stem3x3");
      const int r = pt_launch_stem3x3s2(x, c.F(w), c.F(b), t.p, n, H, W, c.x3, s, 0);
      if (r != PT_OK) c.rc = r;
    }
  }
  for (int i = 0; i < 13; ++i) {
    const int k = cfg[i][0], cout = cfg[i][2], se = cfg[i][4];
    const int st = cfg[i][3] < 0 ? 1 : stage_stride[cfg[i][3]];
    const std::string q = names[i];
    if (!se && st == 1 && !c.x3) {      // plain DepthwiseSeparable block at stride 1: one launch
      const int cstore = cout < 32 ? 32 : cout;
      T o = c.alloc(t.H, t.W, cstore);
      if (c.dwpw(t, q + ".dw", k, 1, 2, q + ".pw", cout < 64 ? 64 : cout, o, 2, cout < 64 ? cstore : 0)) {
        t = o;
        if (i == 4) feats[0] = t;
        if (i == 10) feats[1] = t;
        if (i == 12) feats[2] = t;
        continue;
      }
    }
    T d = c.dw(t, q + ".dw", k, st, 2);
    if (se) {
      const PtTensor *w1 = c.get(q + ".se.w1"), *b1 = c.get(q + ".se.b1"), *w2 = c.get(q + ".se.w2"), *b2 = c.get(q + ".se.b2");
      T g = c.alloc(d.H, d.W, d.C);
      if (c.go()) {
        PtProfScope ps(e, s, PT_PROF_OTHER, 0, "lcnet SE");
        const int r = pt_launch_se(d.p, c.F(w1), c.F(b1), c.F(w2), c.F(b2), c.gate, g.p, n, d.H * d.W, d.C, c.x3, s, d.C / 4, 0, c.part);
        if (r != PT_OK) c.rc = r;
      }
      d = g;
    }
    const int cstore = cout < 32 ? 32 : cout;
    T o = c.alloc(d.H, d.W, cstore);
    c.pw(d, q + ".pw", cout < 64 ? 64 : cout, o, 2, nullptr, 1, cout < 64 ? cstore : 0);
    t = o;
    if (i == 4) feats[0] = t;
    if (i == 10) feats[1] = t;
    if (i == 12) feats[2] = t;
  }
  return t;
}

}  // namespace

// x: NHWC4 bf16 [n, H, W, 4] (8 channels in BF16X3 mode); heads[l]: fp32 [n, A_l, 40], A_l = ceil-chain of the strides
int pt_picodet_forward_net(pt_engine* e, const bf16_t* x, int n, int H, int W, float* h0, float* h1, float* h2, float* h3,
                          hipStream_t s) {
  PT_REQUIRE(x && h0 && h1 && h2 && h3 && n > 0, "layout net: null pointer");
  PT_REQUIRE(H % 32 == 0 && W % 32 == 0 && H > 0 && W > 0, "layout net: input %dx%d must be multiples of 32", H, W);
  auto it = e->models.find(PT_MODEL_PICODET);
  if (it == e->models.end()) {
    pt_set_error("PicoDet weights not loaded (pt_weights_load(PT_MODEL_PICODET))");
    return PT_ERR_STATE;
  }
  if (!pt_model_format_ok(it->second, "PT_MODEL_PICODET")) return PT_ERR_STATE;
  Ctx c;
  c.e = e; c.m = &it->second; c.s = s; c.n = n;
  c.x3 = pt_split(e) ? 1 : 0;
  c.mul = c.x3 ? 2 : 1;
  c.rc = PT_OK;
  float* heads[4] = {h0, h1, h2, h3};
  static const int st22[4] = {2, 2, 2, 2};
  for (int pass = 0; pass < 2; ++pass) {
    c.dry = pass == 0;
    c.ok = true;
    e->arenas[PT_ARENA_LAYOUT].reset();
    c.gate = reinterpret_cast<float*>(e->arenas[PT_ARENA_LAYOUT].take((size_t)n * 512 * sizeof(float)));
    c.part = reinterpret_cast<float*>(e->arenas[PT_ARENA_LAYOUT].take((size_t)n * PT_SE_CHUNKS * 512 * sizeof(float)));
    if (!c.gate || !c.part) c.ok = false;
    T feats[3];
    lcnet_backbone(c, x, H, W, st22, feats);
    // ---- CSP-PAN (csp_pan.py:305-345)
    T ins[3];
    for (int i = 0; i < 3; ++i) {
      ins[i] = c.alloc(feats[i].H, feats[i].W, 128);
      c.pw(feats[i], "neck.t" + std::to_string(i), 128, ins[i], 2);
    }
    PT_REQUIRE(ins[1].H == 2 * ins[2].H && ins[1].W == 2 * ins[2].W && ins[0].H == 2 * ins[1].H && ins[0].W == 2 * ins[1].W,
               "layout net: feature maps %dx%d / %dx%d / %dx%d are not exact halves", ins[0].H, ins[0].W, ins[1].H, ins[1].W,
               ins[2].H, ins[2].W);
    T inner1 = c.csp(ins[2], ins[1], "neck.td0", 2);
    T inner0 = c.csp(inner1, ins[0], "neck.td1", 2);
    T outs[4];
    outs[0] = inner0;
    outs[1] = c.csp(c.dp(outs[0], "neck.down0", 2), inner1, "neck.bu0", 1);
    outs[2] = c.csp(c.dp(outs[1], "neck.down1", 2), ins[2], "neck.bu1", 1);
    {
      T a = c.dp(ins[2], "neck.top1", 2), b = c.dp(outs[2], "neck.top2", 2);
      outs[3] = c.alloc(a.H, a.W, 128);
      if (c.go()) {
        const int r = pt_launch_add(a.p, b.p, outs[3].p, (long long)n * a.H * a.W, 128, c.x3, s);
        if (r != PT_OK) c.rc = r;
      }
    }
    // ---- PicoFeat towers + head_cls (pico_head.py:154-167, 1114-1140)
    for (int l = 0; l < 4; ++l) {
      T f = outs[l];
      for (int i = 0; i < 4; ++i) {
        const std::string q = "head." + std::to_string(l) + "." + std::to_string(i);
        T o = c.alloc(f.H, f.W, 128);
        if (!c.dwpw(f, q + ".dw", 5, 1, 2, q + ".pw", 128, o, 2)) {
          T d = c.dw(f, q + ".dw", 5, 1, 2);
          c.pw(d, q + ".pw", 128, o, 2);
        }
        f = o;
      }
      c.pw(f, "head." + std::to_string(l) + ".out", 64, T(), 0, nullptr, 1, 40, heads[l], 40);
    }
    if (c.rc != PT_OK) return c.rc;
    if (pass == 0) {
      if (c.ok) continue;
      PT_HIP_CHECK(hipDeviceSynchronize());
      if (e->arenas[PT_ARENA_LAYOUT].base) PT_HIP_CHECK(hipFree(e->arenas[PT_ARENA_LAYOUT].base));
      e->arenas[PT_ARENA_LAYOUT].base = nullptr;
      const size_t want = pt_arena_round(e->arenas[PT_ARENA_LAYOUT].high);
      PT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->arenas[PT_ARENA_LAYOUT].base), want));
      e->arenas[PT_ARENA_LAYOUT].cap = want;
      continue;
    }
    if (!c.ok) {
      pt_set_error("layout net: activation arena allocation failed");
      return PT_ERR_HIP;
    }
  }
  return PT_OK;
}

// PP-LCNet classifier (`PPLCNet.forward`, cls_pp_lcnet.py:262-283).  x: NHWC4 bf16 [n, H, W, 4] (8 channels in BF16X3
// mode); textline != 0: stride_list [2, [2,1], [2,1], [2,1], [2,1]] (textline_orientation / language_classification).
// logits: fp32 [n, 16], the first *n_classes columns valid.  slot: which of the PT_CLS_SLOTS loaded classifiers
// (model kind PT_MODEL_PPLCNET + slot) -- the reference keeps several alive at once (ocr_system_task.py:116-146).
int pt_pplcnet_forward_net(pt_engine* e, int slot, const bf16_t* x, int n, int H, int W, int textline, float* logits,
                           int* n_classes, hipStream_t s) {
  PT_REQUIRE(x && logits && n > 0 && H > 0 && W > 0 && slot >= 0 && slot < PT_CLS_SLOTS, "PP-LCNet: bad arguments");
  auto it = e->models.find(PT_MODEL_PPLCNET + slot);
  if (it == e->models.end()) {
    pt_set_error("PP-LCNet weights not loaded (pt_weights_load(PT_MODEL_PPLCNET + %d))", slot);
    return PT_ERR_STATE;
  }
  if (!pt_model_format_ok(it->second, "PT_MODEL_PPLCNET")) return PT_ERR_STATE;
  Ctx c;
  c.e = e; c.m = &it->second; c.s = s; c.n = n;
  c.x3 = pt_split(e) ? 1 : 0;
  c.mul = c.x3 ? 2 : 1;
  c.rc = PT_OK;
  c.what = "PP-LCNet";
  const PtTensor* nc = c.get("fc.nclass");
  if (!nc) return c.rc;
  if (n_classes) *n_classes = (int)nc->dims[0];
  const int st22[4] = {2, 2, 2, 2}, st21[4] = {(2 << 8) | 1, (2 << 8) | 1, (2 << 8) | 1, (2 << 8) | 1};
  const int rows = (n + 31) / 32 * 32;
  for (int pass = 0; pass < 2; ++pass) {
    c.dry = pass == 0;
    c.ok = true;
    e->arenas[PT_ARENA_LAYOUT].reset();
    c.gate = reinterpret_cast<float*>(e->arenas[PT_ARENA_LAYOUT].take((size_t)n * 512 * sizeof(float)));
    float* part = reinterpret_cast<float*>(e->arenas[PT_ARENA_LAYOUT].take((size_t)n * PT_SE_CHUNKS * 512 * sizeof(float)));
    c.part = part;
    if (!c.gate || !part) c.ok = false;
    T feats[3];
    T t = lcnet_backbone(c, x, H, W, textline ? st21 : st22, feats);
    // avg_pool -> last_conv (1x1, no bias) + hardswish -> fc: the pooled vectors form a [rows/32, 32] "image" of 512 channels
    const int keep = c.n;
    T mean;
    mean.H = rows / 32; mean.W = 32; mean.C = 512;
    mean.p = reinterpret_cast<bf16_t*>(e->arenas[PT_ARENA_LAYOUT].take((size_t)rows * 512 * c.mul * sizeof(bf16_t)));
    T hid;
    hid.H = rows / 32; hid.W = 32; hid.C = 1280;
    hid.p = reinterpret_cast<bf16_t*>(e->arenas[PT_ARENA_LAYOUT].take((size_t)rows * 1280 * c.mul * sizeof(bf16_t)));
    float* lg = reinterpret_cast<float*>(e->arenas[PT_ARENA_LAYOUT].take((size_t)rows * 16 * sizeof(float)));
    if (!mean.p || !hid.p || !lg) c.ok = false;
    if (c.go()) {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "pplcnet avgpool");
      const int r = pt_launch_chan_mean(t.p, n, t.H * t.W, 512, c.x3, part, mean.p, rows, s);
      if (r != PT_OK) c.rc = r;
    }
    c.n = 1;     // the two head GEMMs see one [rows/32, 32] map
    c.pw(mean, "last_conv", 1280, hid, 2);
    c.pw(hid, "fc", 64, T(), 0, nullptr, 1, 16, lg, 16);
    c.n = keep;
    if (c.go()) PT_HIP_CHECK(hipMemcpyAsync(logits, lg, (size_t)n * 16 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (c.rc != PT_OK) return c.rc;
    if (pass == 0) {
      if (c.ok) continue;
      PT_HIP_CHECK(hipDeviceSynchronize());
      if (e->arenas[PT_ARENA_LAYOUT].base) PT_HIP_CHECK(hipFree(e->arenas[PT_ARENA_LAYOUT].base));
      e->arenas[PT_ARENA_LAYOUT].base = nullptr;
      const size_t want = pt_arena_round(e->arenas[PT_ARENA_LAYOUT].high);
      PT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->arenas[PT_ARENA_LAYOUT].base), want));
      e->arenas[PT_ARENA_LAYOUT].cap = want;
      continue;
    }
    if (!c.ok) {
      pt_set_error("PP-LCNet: activation arena allocation failed");
      return PT_ERR_HIP;
    }
  }
  return PT_OK;
}

}  // namespace PT_FMT_NS
