// db_model.hip -- launch graph of the DB-ResNet18 text detector on the engine's kernels.
//
// Reference graph: DBModel.forward = SegDetector(ResNet(BasicBlock,[2,2,2,2]))
//   model/db_net/dbnet.py:324-335 (backbone), :615-638 (decoder, eval branch), :533-539 (binarize head).
// Fusions relative to the reference's op list (all arithmetic-preserving up to the numerics contract in
// DESIGN.md): Conv+BN(+ReLU) folded; residual add + ReLU in the conv epilogue; the top-down
// `up(x) + lateral` adds fused into the lateral 1x1 conv's epilogue (nearest x2 read of the residual);
// the nn.Upsample(x8/x4/x2) + torch.cat fused into the out5/out4/out3/out2 conv epilogues (replicated
// stores straight into the 256-channel concat buffer); ConvTranspose2d(2,2) as a GEMM with a
// pixel-shuffle epilogue; the last ConvTranspose2d(64->1) + Sigmoid as one streaming kernel.
//
// Two precisions over the same graph and kernels (pt_engine_set_precision):
//   PT_PRECISION_BF16   activations bf16, one MFMA pass                       (throughput mode)
//   PT_PRECISION_BF16X3 activations (hi, lo) bf16 pairs, K = (x_hi,w_hi)+(x_lo,w_hi)+(x_hi,w_lo): ~2^-16
//                       relative error per product, fp32 accumulate           (fp32-class parity mode)
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace PT_FMT_NS {

namespace {

struct DbWeights {
  const PtTensor *stem_w, *stem_b;
  struct Block {
    const PtTensor *w1, *b1, *w2, *b2, *wd, *bd;
  } blk[4][2];
  const PtTensor *in_w[4], *in_b[4];    // in2..in5 (index 0 = in2)
  const PtTensor *out_w[4], *out_b[4];  // out2..out5
  const PtTensor *bin0_w, *bin0_b, *bin3_w, *bin3_b, *bin6_w, *bin6_b;
  const PtTensor *out2f_w = nullptr, *out2f_b = nullptr, *out2p_w = nullptr, *out2p_b = nullptr;
  const PtTensor *bin0p_w = nullptr, *bin0p_b = nullptr, *bin0c_w = nullptr, *bin0c_b = nullptr;
};

int get(const PtModel& m, const std::string& name, const PtTensor** out, bool optional = false) {
  *out = m.find(name);
  if (!*out && !optional) {
    pt_set_error("weight blob lacks tensor '%s'", name.c_str());
    return PT_ERR_FORMAT;
  }
  return PT_OK;
}

int bind(const PtModel& m, DbWeights& w, bool x3, bool f16) {
  int rc;
  const std::string ws3 = x3 ? ".w3" : ".w";  // split-precision weight tiles carry the suffix .w3
  const std::string ws = f16 ? ".wh" : ws3;   // ... the fp16 tiles of PT_PRECISION_F16X2 .wh (implicit-GEMM layers only)
#define G(name, field) if ((rc = get(m, name, &w.field)) != PT_OK) return rc
  G("stem" + ws3, stem_w); G("stem.b", stem_b);
  for (int l = 0; l < 4; ++l)
    for (int b = 0; b < 2; ++b) {
      const std::string p = "layer" + std::to_string(l + 1) + "." + std::to_string(b);
      G(p + ".conv1" + ws, blk[l][b].w1); G(p + ".conv1.b", blk[l][b].b1);
      G(p + ".conv2" + ws, blk[l][b].w2); G(p + ".conv2.b", blk[l][b].b2);
      if ((rc = get(m, p + ".down" + ws, &w.blk[l][b].wd, true)) != PT_OK) return rc;
      if ((rc = get(m, p + ".down.b", &w.blk[l][b].bd, true)) != PT_OK) return rc;
    }
  for (int i = 0; i < 4; ++i) {
    const std::string k = std::to_string(i + 2);
    G("in" + k + ws, in_w[i]); G("in" + k + ".b", in_b[i]);
    G("out" + k + ws, out_w[i]); G("out" + k + ".b", out_b[i]);
  }
  G("bin0" + ws, bin0_w); G("bin0.b", bin0_b); G("bin3" + ws, bin3_w); G("bin3.b", bin3_b);
  // fused out2 (packer: conv3x3(W_o . W_i, c2) + phase conv of o3); older blobs do not carry it
  if ((rc = get(m, "out2f" + ws, &w.out2f_w, true)) != PT_OK) return rc;
  if ((rc = get(m, "out2f.b", &w.out2f_b, true)) != PT_OK) return rc;
  if ((rc = get(m, "out2p" + ws, &w.out2p_w, true)) != PT_OK) return rc;
  if ((rc = get(m, "out2p.b", &w.out2p_b, true)) != PT_OK) return rc;
  // binarize.0 without the concat (packer: phase conv of the 1/8-resolution concat + 64 -> 64 conv of p2); older blobs do not carry it
  if ((rc = get(m, "bin0p" + ws, &w.bin0p_w, true)) != PT_OK) return rc;
  if ((rc = get(m, "bin0p.b", &w.bin0p_b, true)) != PT_OK) return rc;
  if ((rc = get(m, "bin0c" + ws, &w.bin0c_w, true)) != PT_OK) return rc;
  if ((rc = get(m, "bin0c.b", &w.bin0c_b, true)) != PT_OK) return rc;
  G(x3 ? "bin6.wf32" : "bin6.w", bin6_w); G("bin6.b", bin6_b);
#undef G
  return PT_OK;
}

inline const bf16_t* W(const PtTensor* t) { return reinterpret_cast<const bf16_t*>(t->d_ptr); }
inline const float* Bv(const PtTensor* t) { return reinterpret_cast<const float*>(t->d_ptr); }

struct Bufs {
  bf16_t *s, *p, *t[4], *a[4], *c[4], *d[4], *in5, *o4, *o3, *o2, *fuse, *y0, *y1, *f8, *p2, *yb;
};

}  // namespace

int pt_db_forward_net(pt_engine* e, const bf16_t* x, int n, int H, int W_, float* prob, float* logits, hipStream_t s, uint32_t* bitmap, float thresh,
                      int* bitmap_done) {
  if (bitmap_done) *bitmap_done = 0;
  PT_REQUIRE(H % 32 == 0 && W_ % 32 == 0 && H > 0 && W_ > 0, "det net: input %dx%d must be multiples of 32", H, W_);
  auto it = e->models.find(PT_MODEL_DB_RESNET18);
  if (it == e->models.end()) {
    pt_set_error("DB-ResNet18 weights not loaded (pt_weights_load(PT_MODEL_DB_RESNET18))");
    return PT_ERR_STATE;
  }
  if (!pt_model_format_ok(it->second, "PT_MODEL_DB_RESNET18")) return PT_ERR_STATE;
  const int x3 = pt_split(e) ? 1 : 0;
  const int m = x3 ? 2 : 1;  // channel-group multiplier of every activation buffer
  DbWeights w;
  const int f16 = pt_f16x2(e) && it->second.find("bin0.wh") ? 1 : 0;      // blobs packed without the fp16 tiles run as BF16X3
  if (pt_f16x2(e) && !f16) {
    static bool said = false;      // once per process: the caller asked for F16X2 and gets BF16X3 numbers and speed
    if (!said) {
      said = true;
      fprintf(stderr, "[pdftable_hip] PT_PRECISION_F16X2 requested but the detector blob has no '.wh' (fp16) tiles: running PT_PRECISION_BF16X3 "
                      "(pack with x3=True to get them)\n");
    }
  }
  int rc = bind(it->second, w, x3 != 0, f16 != 0);
  if (rc != PT_OK) return rc;

  const int ch[4] = {64, 128, 256, 512};
  // out2 without its lateral (PT_DB_FUSE_OUT2=0: the layer-by-layer path; A/B switch): see the packer for the algebra
  static int fuse_out2 = -1;
  if (fuse_out2 < 0) {
    const char* ev = getenv("PT_DB_FUSE_OUT2");
    fuse_out2 = ev ? atoi(ev) : 1;
  }
  const bool fused2 = fuse_out2 && w.out2f_w && w.out2f_b && w.out2p_w && w.out2p_b;
  // binarize.0 without the 256-channel concat at 1/4 resolution (PT_DB_FUSE_BIN0=0: the concat path; A/B switch, read per call)
  const char* fb_env = getenv("PT_DB_FUSE_BIN0");
  const bool fused0 = fused2 && !(fb_env && atoi(fb_env) == 0) && w.bin0p_w && w.bin0p_b && w.bin0c_w && w.bin0c_b;
  Bufs bf;
  for (int attempt = 0; attempt < 2; ++attempt) {
    e->arenas[PT_ARENA_DET].reset();
    bool ok = true;
    auto take = [&](size_t elems) {
      void* p = e->arenas[PT_ARENA_DET].take(elems * m * sizeof(bf16_t));
      if (!p) ok = false;
      return reinterpret_cast<bf16_t*>(p);
    };
    const size_t px2 = (size_t)n * (H / 2) * (W_ / 2), px4 = px2 / 4;
    bf.s = take(px2 * 64);
    bf.p = take(px4 * 64);
    size_t px = px4;
    for (int l = 0; l < 4; ++l) {
      bf.t[l] = take(px * ch[l]); bf.a[l] = take(px * ch[l]); bf.c[l] = take(px * ch[l]);
      bf.d[l] = l ? take(px * ch[l]) : nullptr;
      px /= 4;
    }
    bf.in5 = take(px4 / 64 * 256); bf.o4 = take(px4 / 16 * 256); bf.o3 = take(px4 / 4 * 256);
    bf.o2 = fused2 ? nullptr : take(px4 * 256);
    bf.fuse = fused0 ? nullptr : take(px4 * 256);
    bf.y0 = take(px4 * 64);
    bf.y1 = take(px2 * 64);
    bf.f8 = bf.p2 = bf.yb = nullptr;
    if (fused0) { bf.f8 = take(px4 / 4 * 192); bf.p2 = take(px4 * 64); bf.yb = take(px4 * 64); }
    if (ok) break;
    if (attempt == 1) {
      pt_set_error("activation arena allocation failed");
      return PT_ERR_HIP;
    }
    PT_HIP_CHECK(hipDeviceSynchronize());
    if (e->arenas[PT_ARENA_DET].base) PT_HIP_CHECK(hipFree(e->arenas[PT_ARENA_DET].base));
    e->arenas[PT_ARENA_DET].base = nullptr;
    const size_t want = pt_arena_round(e->arenas[PT_ARENA_DET].high);
    PT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->arenas[PT_ARENA_DET].base), want));
    e->arenas[PT_ARENA_DET].cap = want;
  }

#define RUN(call) do { if ((rc = (call)) != PT_OK) return rc; } while (0)
  // every conv below: out buffer of C channels is laid out [hi(C) | lo(C)] in x3 mode
  auto conv = [&](const bf16_t* in, int hh, int ww, int cin, const PtTensor* wt, const PtTensor* bs, int N, int ks,
                  int stride, bf16_t* out, int out_c, int relu) {
    ConvDesc c;
    c.in = in; c.B = n; c.H = hh; c.W = ww; c.Cin = cin; c.w = W(wt); c.bias = Bv(bs); c.N = N; c.ks = ks;
    c.stride = stride; c.out = out; c.out_cstride = out_c * m; c.relu = relu; c.split = f16 ? 2 : x3; c.out_lo_off = out_c;
    c.xp_store = 1;      // full-line stores where the launch takes the register epilogue (conv_igemm.hip: +0.9 % det-only)
    return c;
  };
  // bf16 mode: stem + max-pool in one kernel (the 64-channel half-resolution map stays in LDS; bit-identical);
  // PT_STEM_POOL=0 (read per call) keeps the two launches
  const char* sp_env = getenv("PT_STEM_POOL");
  if (!x3 && !(sp_env && atoi(sp_env) == 0)) {
    RUN(pt_launch_stem7x7_pool(e, x, n, H, W_, W(w.stem_w), Bv(w.stem_b), bf.p, s));
  } else {
    RUN(pt_launch_stem7x7(e, x, n, H, W_, W(w.stem_w), Bv(w.stem_b), bf.s, x3, s));
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "maxpool");
    RUN(pt_launch_maxpool3x3s2(bf.s, n, H / 2, W_ / 2, 64, bf.p, x3, s));
  }
  const bf16_t* cur = bf.p;
  int ch_in = 64, hh = H / 4, ww = W_ / 4;
  for (int l = 0; l < 4; ++l) {
    for (int b = 0; b < 2; ++b) {
      const int stride = (l > 0 && b == 0) ? 2 : 1;
      const DbWeights::Block& bw = w.blk[l][b];
      ConvDesc c1 = conv(cur, hh, ww, ch_in, bw.w1, bw.b1, ch[l], 3, stride, bf.t[l], ch[l], 1);
      RUN(pt_launch_conv(e, c1, s));
      const bf16_t* res = cur;
      if (bw.wd) {
        ConvDesc cd = conv(cur, hh, ww, ch_in, bw.wd, bw.bd, ch[l], 1, stride, bf.d[l], ch[l], 0);
        RUN(pt_launch_conv(e, cd, s));
        res = bf.d[l];
      }
      hh /= stride; ww /= stride;
      bf16_t* dst = (b == 0) ? bf.a[l] : bf.c[l];
      ConvDesc c2 = conv(bf.t[l], hh, ww, ch[l], bw.w2, bw.b2, ch[l], 3, 1, dst, ch[l], 1);
      c2.res = res; c2.res_mode = 1;
      RUN(pt_launch_conv(e, c2, s));
      cur = dst;
      ch_in = ch[l];
    }
  }
  // decoder: lateral 1x1 convs with the top-down add fused (in5 first, then in4 + up(in5), ...)
  bf16_t* lat[4] = {bf.o2, bf.o3, bf.o4, bf.in5};  // index i <-> feature c[i]
  for (int i = 3; i >= (fused2 ? 1 : 0); --i) {
    ConvDesc c = conv(bf.c[i], H >> (2 + i), W_ >> (2 + i), ch[i], w.in_w[i], w.in_b[i], 256, 1, 1, lat[i], 256, 0);
    if (i < 3) { c.res = lat[i + 1]; c.res_mode = 2; }
    RUN(pt_launch_conv(e, c, s));
  }
  // phase masks of conv3x3(W, up2(x)) run at the resolution of x: output phase (dy, dx) uses taps (dy..dy+1) x (dx..dx+1)
  auto phase_masks = [](ConvDesc& c) {
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < 2; ++dx) {
        unsigned mk = 0;
        for (int r = dy; r < dy + 2; ++r)
          for (int q = dx; q < dx + 2; ++q) mk |= 1u << (r * 3 + q);
        c.tap_mask[dy * 2 + dx] = mk;
      }
    c.shuffle_cout = 64;
    c.alg_scale = 4.0 / 9.0;
  };
  // out5/out4/out3/out2: 3x3 256->64, nearest-upsampled x8/x4/x2/x1 and concatenated as (p5,p4,p3,p2); with the fused binarize.0
  // p5/p4/p3 are up-sampled x4/x2/x1 into a 192-channel concat at 1/8 resolution instead
  for (int i = 3; i >= (fused2 ? 1 : 0); --i) {
    ConvDesc c = conv(lat[i], H >> (2 + i), W_ >> (2 + i), 256, w.out_w[i], w.out_b[i], 64, 3, 1, fused0 ? bf.f8 : bf.fuse, fused0 ? 192 : 256, 0);
    c.out_coff = (3 - i) * 64; c.rep = fused0 ? 1 << (i - 1) : 1 << i;
    RUN(pt_launch_conv(e, c, s));
  }
  if (fused2) {
    // (a) conv3x3(W_o, up2(o3)) as a phase convolution at 1/8 resolution: 4 x 64 outputs, pixel-shuffled to 1/4 resolution, 4 of 9 taps
    //     per phase (bf.y0 is free until bin0 writes it)
    ConvDesc cp = conv(lat[1], H >> 3, W_ >> 3, 256, w.out2p_w, w.out2p_b, 256, 3, 1, bf.y0, 64, 0);
    phase_masks(cp);
    RUN(pt_launch_conv(e, cp, s));
    // (b) + conv3x3(W_o . W_i, c2) with (a) as its residual: p2, straight into its slice of the concat (or, fused binarize.0, on its own)
    ConvDesc cf = conv(bf.c[0], H >> 2, W_ >> 2, 64, w.out2f_w, w.out2f_b, 64, 3, 1, fused0 ? bf.p2 : bf.fuse, fused0 ? 64 : 256, 0);
    cf.out_coff = fused0 ? 0 : 192;
    cf.res = bf.y0; cf.res_mode = 1;
    RUN(pt_launch_conv(e, cf, s));
  }
  if (fused0) {
    // binarize.0 = conv3x3(W[:, :192], up2(concat at 1/8)) as a phase convolution + conv3x3(W[:, 192:], p2) + BN bias, ReLU
    ConvDesc cb = conv(bf.f8, H >> 3, W_ >> 3, 192, w.bin0p_w, w.bin0p_b, 256, 3, 1, bf.yb, 64, 0);
    phase_masks(cb);
    RUN(pt_launch_conv(e, cb, s));
    ConvDesc c = conv(bf.p2, H / 4, W_ / 4, 64, w.bin0c_w, w.bin0c_b, 64, 3, 1, bf.y0, 64, 1);
    c.res = bf.yb; c.res_mode = 1;
    RUN(pt_launch_conv(e, c, s));
  } else {
    ConvDesc c = conv(bf.fuse, H / 4, W_ / 4, 256, w.bin0_w, w.bin0_b, 64, 3, 1, bf.y0, 64, 1);
    RUN(pt_launch_conv(e, c, s));
  }
  static int head_mfma = -1;     // PT_DB_HEAD_MFMA=0: the head as an implicit-GEMM launch with the fused epilogue (A/B switch)
  if (head_mfma < 0) {
    const char* ev = getenv("PT_DB_HEAD_MFMA");
    head_mfma = ev ? atoi(ev) : 1;
  }
  if (!x3 && head_mfma) {
    // both transposed convs of the head in one streaming kernel (det_kernels.hip: db_head_mfma_kernel)
    PtProfScope ps(e, s, PT_PROF_CONV1X1, 2.0 * n * (H / 4) * (W_ / 4) * (64.0 * 256 + 256 * 4), "db head (2 x convT) mfma");
    static int fuse_bm = -1;      // PT_DB_FUSE_BITMAP=0: the separate bitmap pass (A/B switch)
    if (fuse_bm < 0) {
      const char* ev = getenv("PT_DB_FUSE_BITMAP");
      fuse_bm = ev ? atoi(ev) : 1;
    }
    uint32_t* bm = (fuse_bm && bitmap && bitmap_done && prob) ? bitmap : nullptr;
    RUN(pt_launch_db_head_mfma(bf.y0, n, H / 4, W_ / 4, W(w.bin3_w), Bv(w.bin3_b), W(w.bin6_w), Bv(w.bin6_b), prob, logits, s, bm, thresh));
    if (bm) *bitmap_done = 1;
  } else {
    // ConvTranspose2d(64,64,2,2)+BN+ReLU with the final ConvTranspose2d(64,1,2,2)+Sigmoid fused into its epilogue
    ConvDesc c = conv(bf.y0, H / 4, W_ / 4, 64, w.bin3_w, w.bin3_b, 256, 1, 1, bf.y1, 64, 1);
    c.shuffle_cout = 64;
    c.head_w = w.bin6_w->d_ptr; c.head_b = Bv(w.bin6_b); c.head_prob = prob; c.head_logits = logits;
    RUN(pt_launch_conv(e, c, s));
  }
#undef RUN
  return PT_OK;
}

}  // namespace PT_FMT_NS
