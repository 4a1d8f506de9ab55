// lore_model.hip -- launch graph of Lore's table-cell detector (DLA-34 backbone + DCN up-sampling + six heads).
//
// Reference graph: DLASeg.forward lore/lore_dla_34.py:184-196
//   base    = dla34: DLA.forward center_net/modeling_centernet.py:382-402 (Tree.forward :259-271, Root :167-175,
//             BasicBlock :58-72)
//   dla_up  = DLAUp.forward :128-134 over IDAUp.forward :106-112; every proj/node is DeformConv :65-83 =
//             DCN (lore/dcnv2.py:71-86) -> BN -> ReLU; every up is a depthwise ConvTranspose2d
//   heads   = conv3x3(64->256)+ReLU+conv1x1 for hm, st, wh, ax, cr, reg (:159-182)
// Engine mapping: Conv+BN(+ReLU) folded, residual add in the conv epilogue; Root's conv over a channel concat is
// evaluated child by child, accumulating through the residual path (no concat tensor); DCN = offset/mask conv (fp32
// out) -> dcn_im2col_kernel -> 1x1 GEMM over 9*C columns; the `up(x) + skip` add is fused into the up-sampler.
// The 16-channel levels (base_layer, level0, level1) run on dedicated thin kernels and keep their real 16 channels.
#include <stdlib.h>

#include <string>
#include <vector>

#include "common.h"

namespace PT_FMT_NS {

int pt_launch_dcn_im2col(const bf16_t* x, const float* om, bf16_t* cols, int B, int H, int W, int C, int split,
                         hipStream_t s);
int pt_launch_conv3x3_c16(pt_engine* e, const bf16_t* in, const bf16_t* w, const float* bias, bf16_t* out, int B, int H, int W,
                          int N, int stride, int split, hipStream_t s);
int pt_launch_dwconvt_up_add(const bf16_t* in, const float* w, const bf16_t* add, bf16_t* out, int B, int h, int wd,
                             int C, int f, int split, hipStream_t s);
int pt_launch_dla_thin_chain(pt_engine* e, const bf16_t* in, int B, int H, int W, const bf16_t* w_stem, const float* b_stem, const bf16_t* w0,
                             const float* b0, const bf16_t* w1, const float* b1, bf16_t* out, hipStream_t s);

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
  bool dry;      // planning pass: only arena accounting, no launches
  bool ok;       // arena had room for everything so far
  int rc;
  bf16_t* cols = nullptr;   // shared DCN column scratch (largest site)
  float* om = nullptr;      // shared offset/mask scratch, fp32 [pixel][32]
  const int* ylimit = nullptr;   // when set: convs skip output tiles at rows >= *ylimit (sparse-head mosaics)
  double alg_scale = 1.0;        // roofline accounting: fraction of a launch's output pixels the algorithm needs

  T alloc(int H, int W, int C) {
    T t;
    t.H = H; t.W = W; t.C = C;
    t.p = reinterpret_cast<bf16_t*>(e->arenas[PT_ARENA_TSR].take((size_t)n * H * W * C * mul * sizeof(bf16_t)));
    if (!t.p) ok = false;
    return t;
  }
  const PtTensor* get(const std::string& name) {
    const PtTensor* t = m->find(name);
    if (!t && rc == PT_OK) {
      pt_set_error("Lore DLA-34 weight blob lacks tensor '%s'", name.c_str());
      rc = PT_ERR_FORMAT;
    }
    return t;
  }
  // conv with folded bias; q = weight name prefix; N = GEMM width (multiple of 64), nv = channels stored (0 = N)
  void conv(const T& in, const std::string& q, int N, int ks, int stride, const T& out, int relu, const T* res = nullptr,
            int nv = 0, float* out_f32 = nullptr, int f32_cs = 0, int cin_override = 0, int alg_n = 0) {
    const PtTensor* w = get(q + (x3 ? ".w3" : ".w"));
    const PtTensor* b = get(q + ".b");
    if (rc != PT_OK || dry || !ok) return;
    ConvDesc c;
    c.in = in.p; c.B = n; c.H = in.H; c.W = in.W; c.Cin = cin_override ? cin_override : in.C;
    c.w = reinterpret_cast<const bf16_t*>(w->d_ptr); c.bias = reinterpret_cast<const float*>(b->d_ptr);
    c.N = N; c.ks = ks; c.stride = stride; c.relu = relu; c.split = x3; c.n_valid = nv; c.ylimit = ylimit;
    c.alg_n = alg_n; c.alg_scale = alg_scale;
    if (out_f32) {
      c.out_f32 = out_f32; c.out_cstride = f32_cs;
    } else {
      c.out = out.p; c.out_cstride = out.C * mul; c.out_lo_off = out.C;
    }
    if (res) { c.res = res->p; c.res_mode = 1; }
    const int r = pt_launch_conv(e, c, s);
    if (r != PT_OK) rc = r;
  }
  // 1x1 conv over the channel concatenation of `ins` (never materialised): one GEMM whose K walks the tensors
  void conv_cat(const std::vector<T>& ins, const std::string& q, int N, const T& out, int relu) {
    const PtTensor* w = get(q + (x3 ? ".w3" : ".w"));
    const PtTensor* b = get(q + ".b");
    if (rc != PT_OK || dry || !ok) return;
    ConvDesc c;
    c.in = ins[0].p; c.B = n; c.H = ins[0].H; c.W = ins[0].W;
    c.nseg = (int)ins.size();
    c.Cin = 0;
    for (int i = 0; i < c.nseg; ++i) {
      c.seg_c[i] = ins[i].C;
      c.Cin += ins[i].C;
      if (i) c.in_more[i - 1] = ins[i].p;
    }
    c.w = reinterpret_cast<const bf16_t*>(w->d_ptr); c.bias = reinterpret_cast<const float*>(b->d_ptr);
    c.N = N; c.ks = 1; c.stride = 1; c.relu = relu; c.split = x3; c.alg_scale = alg_scale;
    c.out = out.p; c.out_cstride = out.C * mul; c.out_lo_off = out.C;
    const int r = pt_launch_conv(e, c, s);
    if (r != PT_OK) rc = r;
  }
  T maxpool2(const T& x) {
    T o = alloc(x.H / 2, x.W / 2, x.C);
    if (rc == PT_OK && !dry && ok) {
      e->prof.next_bytes = (double)n * x.H * x.W * x.C * 2.0 * mul * 1.25;
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "maxpool2x2");
      const int r = pt_launch_maxpool_kxk(x.p, n, x.H, x.W, x.C, 2, 2, 0, x3, o.p, s);
      if (r != PT_OK) rc = r;
    }
    return o;
  }
  T block(const std::string& q, const T& x, const T& residual, int stride, int cout) {
    T t = alloc(x.H / stride, x.W / stride, cout);
    conv(x, q + ".conv1", cout, 3, stride, t, 1);
    T o = alloc(t.H, t.W, cout);
    conv(t, q + ".conv2", cout, 3, 1, o, 1, &residual);
    return o;
  }
  T tree(const std::string& q, int levels, const T& x, int cin, int cout, int stride, bool level_root,
         std::vector<T> children) {
    T bottom = stride > 1 ? maxpool2(x) : x;
    if (level_root) children.push_back(bottom);
    if (levels == 1) {
      T residual = bottom;
      if (cin != cout) {
        residual = alloc(bottom.H, bottom.W, cout);
        conv(bottom, q + ".project", cout, 1, 1, residual, 0);
      }
      T x1 = block(q + ".tree1", x, residual, stride, cout);
      T x2 = block(q + ".tree2", x1, x1, 1, cout);
      std::vector<T> ins = {x2, x1};
      ins.insert(ins.end(), children.begin(), children.end());
      T o = alloc(x1.H, x1.W, cout);
      conv_cat(ins, q + ".root", cout, o, 1);      // Root: conv(cat(x2, x1, *children)) + BN + ReLU, one launch
      return o;
    }
    T x1 = tree(q + ".tree1", levels - 1, x, cin, cout, stride, false, {});
    children.push_back(x1);
    return tree(q + ".tree2", levels - 1, x1, cout, cout, 1, false, children);
  }
  // DeformConv (lore_dla_34.py:65-83)
  T dcn(const std::string& q, const T& x, int cout) {
    T o = alloc(x.H, x.W, cout);
    static const bool fused = !(getenv("PT_DCN_FUSED") && atoi(getenv("PT_DCN_FUSED")) == 0);
    // the 27-channel offset / mask conv: inside the deformable conv's kernel where that kernel can (dcn_fused64_kernel<..., OMF = 1>: no launch, no fp32 map
    // through HBM), else as a launch of its own writing `om`
    const bool om_inside = fused && pt_dcn_fuses_om(e, x.C, x3);
    if (!om_inside) conv(x, q + ".om", 64, 3, 1, T(), 0, nullptr, 32, om, 32, 0, 27);     // 18 offsets + 9 masks are the layer's real outputs
    if (fused) {
      const PtTensor* w = get(q + (x3 ? ".dcn.w3" : ".dcn.w"));
      const PtTensor* b = get(q + ".dcn.b");
      const PtTensor* ow = om_inside ? get(q + (x3 ? ".om.w3" : ".om.w")) : nullptr;
      const PtTensor* ob = om_inside ? get(q + ".om.b") : nullptr;
      if (rc == PT_OK && !dry && ok) {
        const int r = pt_launch_dcn_fused(e, x.p, om, reinterpret_cast<const bf16_t*>(w->d_ptr),
                                          reinterpret_cast<const float*>(b->d_ptr), o.p, n, x.H, x.W, x.C, cout, x3, 1, s,
                                          ow ? reinterpret_cast<const bf16_t*>(ow->d_ptr) : nullptr, ob ? reinterpret_cast<const float*>(ob->d_ptr) : nullptr);
        if (r != PT_OK) rc = r;
      }
      return o;
    }
    // two-kernel form (PT_DCN_FUSED=0): columns through HBM, then a 1x1 GEMM
    if (rc == PT_OK && !dry && ok) {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "dcn im2col");
      const int r = pt_launch_dcn_im2col(x.p, om, cols, n, x.H, x.W, x.C, x3, s);
      if (r != PT_OK) rc = r;
    }
    T c;
    c.p = cols; c.H = x.H; c.W = x.W; c.C = 9 * x.C;
    conv(c, q + ".dcn", cout, 1, 1, o, 1);
    return o;
  }
  // IDAUp.forward (lore_dla_34.py:106-112) on layers[startp .. endp)
  void ida(const std::string& q, std::vector<T>& layers, int startp, int endp, int o_ch, const int* up_f) {
    for (int i = startp + 1; i < endp; ++i) {
      const int j = i - startp;
      const std::string js = std::to_string(j);
      T p = dcn(q + ".proj_" + js, layers[i], o_ch);
      const int f = up_f[j];
      T u = alloc(p.H * f, p.W * f, o_ch);
      const PtTensor* wu = get(q + ".up_" + js + ".wf32");
      if (rc == PT_OK && !dry && ok) {
        e->prof.next_bytes = (double)n * p.H * p.W * o_ch * 2.0 * mul * (1.0 + 2.0 * f * f);      // in once, skip + out at f x f the pixels
        char label[48];
        snprintf(label, sizeof(label), "dw convT up + add %d @%dx%d", o_ch, p.H * f, p.W * f);
        PtProfScope ps(e, s, PT_PROF_OTHER, 0, label);
        const int r = pt_launch_dwconvt_up_add(p.p, reinterpret_cast<const float*>(wu->d_ptr), layers[i - 1].p, u.p, n,
                                               p.H, p.W, o_ch, f, x3, s);
        if (r != PT_OK) rc = r;
      }
      layers[i] = dcn(q + ".node_" + js, u, o_ch);
    }
  }
};

}  // namespace

// x: NHWC4 bf16 [n, H, W, 4] ([hi rgb0 | lo rgb0] in BF16X3 mode); heads: fp32 NHWC at H/4 x W/4 with channel
// strides 8 (hm: 2 valid), 8 (st), 8 (wh), 256 (ax), 256 (cr), 8 (reg: 2 valid)
struct SparseArgs {      // fused forward + decode: the ax / cr heads run on patch mosaics around the decoded positions
  int wiz_rev;
  float vis_thresh;
  int* d_counts;
  float *d_dets, *d_logi;
};

static int lore_dla_run(pt_engine* e, const bf16_t* x, int n, int H, int W, float* hm, float* st, float* wh, float* ax,
                        float* cr, float* reg, const SparseArgs* sp, hipStream_t s) {
  PT_REQUIRE(H % 32 == 0 && W % 32 == 0 && H > 0 && W > 0, "Lore net: input %dx%d must be multiples of 32", H, W);
  PT_REQUIRE(x && n > 0 && (sp || (hm && st && wh && ax && cr && reg)), "Lore net: null pointer");
  auto it = e->models.find(PT_MODEL_LORE_DLA34);
  if (it == e->models.end()) {
    pt_set_error("Lore DLA-34 weights not loaded (pt_weights_load(PT_MODEL_LORE_DLA34))");
    return PT_ERR_STATE;
  }
  if (!pt_model_format_ok(it->second, "PT_MODEL_LORE_DLA34")) return PT_ERR_STATE;
  Ctx c;
  c.e = e; c.m = &it->second; c.s = s; c.n = n;
  c.x3 = pt_split(e) ? 1 : 0;
  c.mul = c.x3 ? 2 : 1;
  c.rc = PT_OK;
  const int ch[6] = {16, 32, 64, 128, 256, 512};
  const int lv[6] = {1, 1, 1, 2, 2, 1};
  float* heads[6] = {hm, st, wh, ax, cr, reg};
  const char* hname[6] = {"hm", "st", "wh", "ax", "cr", "reg"};
  const int hcs[6] = {8, 8, 8, 256, 256, 8};
  const int hreal[6] = {2, 8, 8, 256, 256, 2};     // the heads' real channel counts (roofline accounting)

  for (int pass = 0; pass < 2; ++pass) {
    c.dry = pass == 0;     // pass 0 only plans the arena (and grows it if needed), pass 1 launches
    c.ok = true;
    e->arenas[PT_ARENA_TSR].reset();
    // shared DCN scratch: the largest site is 64 channels at H/4 x W/4 (9*64 columns); 128 ch at H/8 is half of it
    const size_t px4 = (size_t)n * (H / 4) * (W / 4);
    c.cols = reinterpret_cast<bf16_t*>(e->arenas[PT_ARENA_TSR].take(px4 * 576 * c.mul * sizeof(bf16_t)));
    c.om = reinterpret_cast<float*>(e->arenas[PT_ARENA_TSR].take(px4 * 32 * sizeof(float)));
    if (!c.cols || !c.om) c.ok = false;

    // bf16 mode: base_layer -> level0 -> level1 in one launch, the two full-resolution 16-channel maps never leave the CU
    // (lore_kernels.hip: dla_thin_chain_kernel; PT_DLA_CHAIN=0: the three launches, A/B switch read per call)
    const char* chain_env = getenv("PT_DLA_CHAIN");
    const bool chain = !c.x3 && H % 2 == 0 && W % 2 == 0 && !(chain_env && atoi(chain_env) == 0);
    std::vector<T> layers(6);
    if (chain) {
      layers[1] = c.alloc(H / 2, W / 2, 32);
      const PtTensor *ws = c.get("base_layer.w"), *bs = c.get("base_layer.b"), *w0 = c.get("level0.wt"), *b0 = c.get("level0.bt"),
                     *w1 = c.get("level1.wt"), *b1 = c.get("level1.bt");
      if (c.rc == PT_OK && !c.dry && c.ok) {
        const int r = pt_launch_dla_thin_chain(e, x, n, H, W, reinterpret_cast<const bf16_t*>(ws->d_ptr), reinterpret_cast<const float*>(bs->d_ptr),
                                               reinterpret_cast<const bf16_t*>(w0->d_ptr), reinterpret_cast<const float*>(b0->d_ptr),
                                               reinterpret_cast<const bf16_t*>(w1->d_ptr), reinterpret_cast<const float*>(b1->d_ptr), layers[1].p, s);
        if (r != PT_OK) c.rc = r;
      }
    } else {
    T t0 = c.alloc(H, W, 16);
    if (!c.dry && c.ok) {
      const PtTensor* w = c.get(c.x3 ? "base_layer.w3" : "base_layer.w");
      const PtTensor* b = c.get("base_layer.b");
      if (c.rc == PT_OK) {
        const int r = pt_launch_stem7x7(e, x, n, H, W, reinterpret_cast<const bf16_t*>(w->d_ptr),
                                        reinterpret_cast<const float*>(b->d_ptr), t0.p, c.x3, s, 1, 16);
        if (r != PT_OK) c.rc = r;
      }
    }
    layers[0] = c.alloc(H, W, 16);
    layers[1] = c.alloc(H / 2, W / 2, 32);
    for (int lv1 = 0; lv1 < 2; ++lv1) {
      const std::string q = lv1 ? "level1" : "level0";
      const PtTensor* w = c.get(q + (c.x3 ? ".wt3" : ".wt"));
      const PtTensor* b = c.get(q + ".bt");
      if (c.rc == PT_OK && !c.dry && c.ok) {
        const int r = pt_launch_conv3x3_c16(e, lv1 ? layers[0].p : t0.p, reinterpret_cast<const bf16_t*>(w->d_ptr),
                                            reinterpret_cast<const float*>(b->d_ptr), layers[lv1].p, n, H, W, lv1 ? 32 : 16,
                                            lv1 ? 2 : 1, c.x3, s);
        if (r != PT_OK) c.rc = r;
      }
    }
    }
    for (int l = 2; l < 6; ++l)
      layers[l] = c.tree("level" + std::to_string(l), lv[l], layers[l - 1], ch[l - 1], ch[l], 2, l > 2, {});

    // DLAUp.forward (lore_dla_34.py:128-134): ida_0 on [4,6), ida_1 on [3,6), ida_2 on [2,6)
    std::vector<T> out = {layers[5]};
    const int f2[4] = {1, 2, 2, 2};
    c.ida("dla_up.ida_0", layers, 4, 6, 256, f2);
    out.insert(out.begin(), layers[5]);
    c.ida("dla_up.ida_1", layers, 3, 6, 128, f2);
    out.insert(out.begin(), layers[5]);
    c.ida("dla_up.ida_2", layers, 2, 6, 64, f2);
    out.insert(out.begin(), layers[5]);
    // ida_up on clones of out[0..3) (strides 4, 8, 16): up factors 2 and 4
    std::vector<T> y = {out[0], out[1], out[2]};
    const int f24[3] = {1, 2, 4};
    c.ida("ida_up", y, 0, 3, 64, f24);
    const T feat = y[2];
    T hid = c.alloc(feat.H, feat.W, 256);
    if (sp) {      // only `hm` is needed everywhere: the other five heads run on patch mosaics (lore_decode.hip)
      heads[0] = reinterpret_cast<float*>(e->arenas[PT_ARENA_TSR].take((size_t)n * feat.H * feat.W * 8 * sizeof(float)));
      if (!heads[0]) c.ok = false;
    }
    for (int h = 0; h < 6; ++h) {
      if (sp && h != 0) continue;
      c.conv(feat, std::string(hname[h]) + ".0", 256, 3, 1, hid, 1);
      c.conv(hid, std::string(hname[h]) + ".2", hcs[h] < 64 ? 64 : hcs[h], 1, 1, T(), 0, nullptr, hcs[h], heads[h], hcs[h], 0, hreal[h]);
    }
    if (sp) {
      int rows_ax = 0, rows_cr = 0, rows_cell = 0, rows_corner = 0;
      pt_lore_mosaic_rows(n, &rows_ax, &rows_cr, &rows_cell, &rows_corner);
      const int MW = 32;          // patches per row of the patch image (lore_decode.hip: MOS_PW); a patch = one row of 9 * 64 values
      auto take = [&](size_t bytes) { void* p_ = e->arenas[PT_ARENA_TSR].take(bytes); if (!p_) c.ok = false; return p_; };
      auto mosaic = [&](int rows, int C) {
        T t;
        t.H = rows; t.W = MW; t.C = C;
        t.p = reinterpret_cast<bf16_t*>(take(((size_t)rows * MW * C * c.mul + 64) * sizeof(bf16_t)));
        return t;
      };
      T mcell = mosaic(rows_cell, 9 * 64), mcorner = mosaic(rows_corner, 9 * 64), max_ = mosaic(rows_ax, 9 * 64), mcr_ = mosaic(rows_cr, 9 * 64);
      T mhid = mosaic(rows_cr, 256);                 // hidden layer of whichever head is running (rows_cr is the largest)
      float* o_wh = reinterpret_cast<float*>(take((size_t)rows_cell * MW * 8 * sizeof(float)));
      float* o_regc = reinterpret_cast<float*>(take((size_t)rows_cell * MW * 8 * sizeof(float)));
      float* o_st = reinterpret_cast<float*>(take((size_t)rows_corner * MW * 8 * sizeof(float)));
      float* o_regk = reinterpret_cast<float*>(take((size_t)rows_corner * MW * 8 * sizeof(float)));
      float* oax = reinterpret_cast<float*>(take((size_t)rows_ax * MW * 256 * sizeof(float)));
      float* ocr = reinterpret_cast<float*>(take((size_t)rows_cr * MW * 256 * sizeof(float)));
      if (c.rc == PT_OK && !c.dry && c.ok) {
        const int keep = c.n;
        // one head on one mosaic: 3x3 (64 -> 256) + ReLU, then 1x1 to nc channels, fp32 out; tiles below `lim` rows exit
        auto head = [&](const T& mos, int rows, const char* name, int nc, float* outp, const int* lim, int real_nc) {
          T hh = mhid;
          hh.H = rows;
          c.n = 1;
          c.ylimit = lim;
          // the 3x3 layer as a GEMM over patch rows: K = 9 * 64 in the conv kernel's own (chunk, tap, channel) order, so the
          // layer's packed 3x3 weight tiles are the 18 K-chunks of this 1x1 launch as they stand
          c.conv(mos, std::string(name) + ".0", 256, 1, 1, hh, 1);
          c.conv(hh, std::string(name) + ".2", nc < 64 ? 64 : nc, 1, 1, T(), 0, nullptr, nc, outp, nc, 0, real_nc);
          c.ylimit = nullptr;
          c.n = keep;
        };
        const int *lim_cell = nullptr, *lim_corner = nullptr, *lim_ax = nullptr, *lim_cr = nullptr;
        int r = pt_lore_decode_peaks(e, heads[0], n, feat.H, feat.W, sp->wiz_rev, sp->vis_thresh, sp->d_counts, &lim_cell,
                                     &lim_corner, s);
        if (r == PT_OK) r = pt_lore_peak_patches(e, feat.p, n, feat.H, feat.W, 64, c.x3, mcell.p, mcorner.p, s);
        if (r != PT_OK) return r;
        head(mcell, rows_cell, "wh", 8, o_wh, lim_cell, 8);
        head(mcell, rows_cell, "reg", 8, o_regc, lim_cell, 2);
        head(mcorner, rows_corner, "st", 8, o_st, lim_corner, 8);
        head(mcorner, rows_corner, "reg", 8, o_regk, lim_corner, 2);
        if (c.rc != PT_OK) return c.rc;
        r = pt_lore_decode_boxes(e, o_regc, o_wh, o_regk, o_st, n, feat.H, feat.W, &lim_ax, &lim_cr, s);
        if (r == PT_OK) r = pt_lore_patch_gather(e, feat.p, n, feat.H, feat.W, 64, c.x3, max_.p, mcr_.p, s);
        if (r != PT_OK) return r;
        head(max_, rows_ax, "ax", 256, oax, lim_ax, 256);
        head(mcr_, rows_cr, "cr", 256, ocr, lim_cr, 256);
        if (c.rc != PT_OK) return c.rc;
        r = pt_lore_decode_sparse(e, oax, ocr, n, feat.H, feat.W, sp->vis_thresh, sp->d_counts, sp->d_dets, sp->d_logi, s);
        if (r != PT_OK) return r;
      }
    }
    if (c.rc != PT_OK) return c.rc;
    if (pass == 0) {
      if (c.ok) continue;      // everything fits: next pass launches
      PT_HIP_CHECK(hipDeviceSynchronize());
      if (e->arenas[PT_ARENA_TSR].base) PT_HIP_CHECK(hipFree(e->arenas[PT_ARENA_TSR].base));
      e->arenas[PT_ARENA_TSR].base = nullptr;
      const size_t want = pt_arena_round(e->arenas[PT_ARENA_TSR].high);
      PT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->arenas[PT_ARENA_TSR].base), want));
      e->arenas[PT_ARENA_TSR].cap = want;
      continue;
    }
    if (!c.ok) {
      pt_set_error("Lore net: activation arena allocation failed");
      return PT_ERR_HIP;
    }
    break;
  }
  return PT_OK;
}


int pt_lore_forward_net(pt_engine* e, const bf16_t* x, int n, int H, int W, float* hm, float* st, float* wh, float* ax,
                        float* cr, float* reg, hipStream_t s) {
  PT_REQUIRE(hm && st && wh && ax && cr && reg, "Lore net: null pointer");
  return lore_dla_run(e, x, n, H, W, hm, st, wh, ax, cr, reg, nullptr, s);
}

// DLA-34 forward and decode in one call with sparse ax / cr heads (see lore_decode.hip); outputs as pt_lore_decode
int pt_lore_forward_decode(pt_engine* e, const bf16_t* x, int n, int H, int W, int wiz_rev, float vis_thresh, int* d_counts,
                           float* d_dets, float* d_logi, hipStream_t s) {
  PT_REQUIRE(d_counts && d_dets && d_logi, "Lore forward+decode: null pointer");
  SparseArgs sp{wiz_rev, vis_thresh, d_counts, d_dets, d_logi};
  return lore_dla_run(e, x, n, H, W, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &sp, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// 'wireless' detector: LoreDetectModel.forward (lore/lore_detector.py:353-389) -- ResNet-18-style backbone whose every
// stage is strided, four ConvTranspose2d(4x4, stride 2) + BN + ReLU up-samplers (run as 3x3 convolutions with 4 x 256
// outputs and a pixel-shuffle epilogue), 1x1 lateral `adaption` convs added through the residual path, heads of
// four 3x3 (-> 64) + ReLU and a 1x1.  Same head-map outputs as pt_lore_forward_net.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct WCtx {
  pt_engine* e;
  const PtModel* m;
  hipStream_t s;
  int n, x3, mul;
  bool dry, ok;
  int rc;
  T alloc(int H, int W, int C) {
    T t;
    t.H = H; t.W = W; t.C = C;
    t.p = reinterpret_cast<bf16_t*>(e->arenas[PT_ARENA_TSR].take((size_t)n * H * W * C * mul * sizeof(bf16_t)));
    if (!t.p) ok = false;
    return t;
  }
  const PtTensor* get(const std::string& name) {
    const PtTensor* t = m->find(name);
    if (!t && rc == PT_OK) {
      pt_set_error("Lore wireless weight blob lacks tensor '%s'", name.c_str());
      rc = PT_ERR_FORMAT;
    }
    return t;
  }
  void conv(const T& in, const std::string& q, int N, int ks, int stride, const T& out, int relu, const T* res = nullptr,
            int shuffle = 0, int nv = 0, float* out_f32 = nullptr, int f32_cs = 0) {
    const PtTensor* w = get(q + (x3 ? ".w3" : ".w"));
    const PtTensor* b = get(q + ".b");
    if (rc != PT_OK || dry || !ok) return;
    ConvDesc c;
    c.in = in.p; c.B = n; c.H = in.H; c.W = in.W; c.Cin = in.C;
    c.w = reinterpret_cast<const bf16_t*>(w->d_ptr); c.bias = reinterpret_cast<const float*>(b->d_ptr);
    c.N = N; c.ks = ks; c.stride = stride; c.relu = relu; c.split = x3; c.n_valid = nv; c.shuffle_cout = shuffle;
    if (shuffle && ks == 3) c.alg_scale = 16.0 / 36.0;   // ConvTranspose2d(k=4,s=2) has 16 taps per (input pixel, Cout); the 3x3 x 4-phase GEMM spends 36
    if (out_f32) {
      c.out_f32 = out_f32; c.out_cstride = f32_cs;
    } else {
      c.out = out.p; c.out_cstride = out.C * mul; c.out_lo_off = out.C;
    }
    if (res) { c.res = res->p; c.res_mode = 1; }
    const int r = pt_launch_conv(e, c, s);
    if (r != PT_OK) rc = r;
  }
};

}  // namespace

int pt_lore_wireless_forward_net(pt_engine* e, const bf16_t* x, int n, int H, int W, float* hm, float* st, float* wh,
                                 float* ax, float* cr, float* reg, hipStream_t s) {
  PT_REQUIRE(H % 64 == 0 && W % 64 == 0 && H > 0 && W > 0, "Lore wireless net: input %dx%d must be multiples of 64", H, W);
  PT_REQUIRE(x && hm && st && wh && ax && cr && reg && n > 0, "Lore wireless net: null pointer");
  auto it = e->models.find(PT_MODEL_LORE_RESNET18);
  if (it == e->models.end()) {
    pt_set_error("Lore wireless weights not loaded (pt_weights_load(PT_MODEL_LORE_RESNET18))");
    return PT_ERR_STATE;
  }
  if (!pt_model_format_ok(it->second, "PT_MODEL_LORE_RESNET18")) return PT_ERR_STATE;
  WCtx c;
  c.e = e; c.m = &it->second; c.s = s; c.n = n;
  c.x3 = pt_split(e) ? 1 : 0;
  c.mul = c.x3 ? 2 : 1;
  c.rc = PT_OK;
  float* heads[6] = {hm, st, wh, ax, cr, reg};
  const char* hname[6] = {"hm", "st", "wh", "ax", "cr", "reg"};
  const int hcs[6] = {8, 8, 8, 256, 256, 8};
  const int planes[4] = {64, 128, 256, 256};
  for (int pass = 0; pass < 2; ++pass) {
    c.dry = pass == 0;
    c.ok = true;
    e->arenas[PT_ARENA_TSR].reset();
    T s0 = c.alloc(H / 2, W / 2, 64);
    if (!c.dry && c.ok) {
      const PtTensor* w = c.get(c.x3 ? "stem.w3" : "stem.w");
      const PtTensor* b = c.get("stem.b");
      if (c.rc == PT_OK) {
        const int r = pt_launch_stem7x7(e, x, n, H, W, reinterpret_cast<const bf16_t*>(w->d_ptr),
                                        reinterpret_cast<const float*>(b->d_ptr), s0.p, c.x3, s, 2, 0);
        if (r != PT_OK) c.rc = r;
      }
    }
    T x0 = c.alloc(H / 4, W / 4, 64);
    if (c.rc == PT_OK && !c.dry && c.ok) {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "maxpool");
      const int r = pt_launch_maxpool3x3s2(s0.p, n, H / 2, W / 2, 64, x0.p, c.x3, s);
      if (r != PT_OK) c.rc = r;
    }
    T feat[4];
    T cur = x0;
    for (int l = 0; l < 4; ++l) {
      for (int b = 0; b < 2; ++b) {
        const std::string q = "layer" + std::to_string(l + 1) + "." + std::to_string(b);
        const int stride = b == 0 ? 2 : 1;
        T t = c.alloc(cur.H / stride, cur.W / stride, planes[l]);
        c.conv(cur, q + ".conv1", planes[l], 3, stride, t, 1);
        T res = cur;
        if (b == 0) {
          res = c.alloc(t.H, t.W, planes[l]);
          c.conv(cur, q + ".down", planes[l], 1, 2, res, 0);
        }
        T o = c.alloc(t.H, t.W, planes[l]);
        c.conv(t, q + ".conv2", planes[l], 3, 1, o, 1, &res);
        cur = o;
      }
      feat[l] = cur;
    }
    // top-down: deconv (3x3 + pixel shuffle + BN + ReLU), then `adaption(lateral) + deconv`
    T up = feat[3];
    const T* lateral[4] = {&feat[2], &feat[1], &feat[0], &x0};
    const char* adapt[4] = {"adaption3", "adaption2", "adaption1", "adaption0"};
    for (int i = 0; i < 4; ++i) {
      T d = c.alloc(up.H * 2, up.W * 2, 256);
      c.conv(up, "deconv" + std::to_string(i + 1), 1024, 3, 1, d, 1, nullptr, 256);
      T o = c.alloc(d.H, d.W, 256);
      c.conv(*lateral[i], adapt[i], 256, 1, 1, o, 0, &d);
      up = o;
    }
    T f = c.alloc(up.H, up.W, 256);
    c.conv(up, "adaptionU1", 256, 1, 1, f, 0);
    T h1 = c.alloc(f.H, f.W, 64), h2 = c.alloc(f.H, f.W, 64);
    for (int h = 0; h < 6; ++h) {
      const int n3 = h == 5 ? 1 : 4;
      const T* in = &f;
      for (int j = 0; j < n3; ++j) {
        const T& o = (j & 1) ? h2 : h1;
        c.conv(*in, std::string(hname[h]) + ".c" + std::to_string(j), 64, 3, 1, o, 1);
        in = &o;
      }
      c.conv(*in, std::string(hname[h]) + ".out", hcs[h] < 64 ? 64 : hcs[h], 1, 1, T(), 0, nullptr, 0, hcs[h], heads[h], hcs[h]);
    }
    if (c.rc != PT_OK) return c.rc;
    if (pass == 0) {
      if (c.ok) continue;
      PT_HIP_CHECK(hipDeviceSynchronize());
      if (e->arenas[PT_ARENA_TSR].base) PT_HIP_CHECK(hipFree(e->arenas[PT_ARENA_TSR].base));
      e->arenas[PT_ARENA_TSR].base = nullptr;
      const size_t want = pt_arena_round(e->arenas[PT_ARENA_TSR].high);
      PT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->arenas[PT_ARENA_TSR].base), want));
      e->arenas[PT_ARENA_TSR].cap = want;
      continue;
    }
    if (!c.ok) {
      pt_set_error("Lore wireless net: activation arena allocation failed");
      return PT_ERR_HIP;
    }
  }
  return PT_OK;
}

}  // namespace PT_FMT_NS
