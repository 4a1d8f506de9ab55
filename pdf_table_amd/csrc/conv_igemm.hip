// conv_igemm.hip -- implicit-GEMM convolution for gfx950 (MI355X), bf16 in / fp32 accumulate.
//
// Covers every dense conv of the hot-path nets (reference layer lists: SURVEY.md appendix A):
//   3x3 s1/s2 and 1x1 s1/s2 convolutions of ResNet-18 / SegDetector (db_net/dbnet.py:102-171,
//   260-336, 513-539) and the CRNN conv stack (crnn/modeling_crnn.py:40-87), plus
//   ConvTranspose2d(k=2,s=2) as a 1x1 GEMM with N = 4*Cout and a pixel-shuffle epilogue.
//
// Design (DESIGN.md "conv kernel"):
//   * activations NHWC bf16, weights pre-tiled [N/64][Cin/32][taps][64][32] bf16 (BN folded on host)
//   * one workgroup = 4 waves = an 8x32 (stride 1) or 4x32 (stride 2) patch of output pixels x 64
//     output channels; the input halo patch for a 32-channel slice is staged ONCE in LDS and all
//     KSxKS taps read it from there (im2col never touches HBM/L2)
//   * v_mfma_f32_32x32x16_bf16: the 32 MFMA rows are 32 consecutive output pixels of one image row,
//     so an A fragment is one ds_read_b128 per lane at a fixed stride (80 B per pixel: 64 B of
//     channels + 16 B pad => conflict-free across the 16-lane groups of ds_read_b128)
//   * next K-slice is prefetched global->VGPR while the current one is multiplied (2 workgroups per
//     CU cover the rest of the latency)
//   * epilogue goes through LDS as fp32 so that bias + residual (+ fused nearest-x2 upsample of the
//     residual) + ReLU + bf16 rounding + (replicated / pixel-shuffled / channel-offset) stores are
//     all 16-byte, channel-contiguous accesses
//   * blockIdx -> tile mapping is XCD-aware: consecutive logical tiles (all N-tiles of a patch, then
//     the neighbouring patch) stay on one XCD so the halo and the weights hit that XCD's L2.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace PT_FMT_NS {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

struct ConvK {
  const bf16_t* in;
  const bf16_t* w;
  const float* bias;
  bf16_t* out;
  const bf16_t* res;
  int B, H, W, Cin, Ho, Wo, N;
  int out_cstride, out_coff, rep, shuffle_cout, res_mode, relu;
  int tiles_x, tiles_y, n_tiles;
  // bf16x3 ("split") precision mode: every activation is a (hi, lo) bf16 pair stored as two channel groups
  // [hi(C) | lo(C)]; K runs over (x_hi, w_hi), (x_lo, w_hi), (x_hi, w_lo); out_lo_off = channel distance hi -> lo
  int split, out_lo_off;
  // fused DB head (dbnet.py:537-539): this conv is ConvTranspose2d(64,64,2,2)+BN+ReLU as a pixel-shuffle GEMM and
  // the epilogue applies the last ConvTranspose2d(64->1,2,2) + Sigmoid on the rounded activations, so the
  // 64-channel 1/2-resolution tensor never goes to HBM.  head_w: [4][64] bf16 (fp32 in split mode).
  const void* head_w;
  const float* head_b;
  float* head_prob;
  float* head_logits;
  // fused per-row arg-max over the N (class) dimension: partial (max, index) per 64-class tile -> [rows][N/64] float2
  float* argmax_part;
  // n_valid > 0: only output channels [0, n_valid) are stored (N is padded to a multiple of 64 with zero weights);
  // out_f32 != null: the plain store path writes fp32 [pixel][out_cstride] there instead of bf16 (network outputs)
  int n_valid;
  float* out_f32;
  const float* res_f32;   // fp32 residual with the layout of out_f32 (transformer residual stream), added before ReLU
  int n_group;            // > 0: column tiles are walked in groups of n_group so that a group's weights stay in one XCD's L2
  const float* slope;     // relu == 3: PReLU, slope[0] = the (single, layer-wide) negative slope, read on the device
  int pool;               // 1: MaxPool2d(2,2), 2: MaxPool2d((2,1)) fused behind bias + ReLU (bf16-rounded first, like the stored map)
  int reps, total_tiles;  // reps > 1: a workgroup walks reps consecutive tiles of total_tiles (see the kernel)
  const int* ylimit;      // device int: tiles whose first output row is >= *ylimit do nothing (data-dependent extents)
  const int* xlimit;      // device int [B]: tiles of image b whose first output column is >= xlimit[b] do nothing (ragged lines)
  const int* xlimit_rows; // device int [Ho]: the same per output ROW (sequence views [1, lines, T, C])
  const int* xcols;       // host-side bookkeeping only (launch_cfg): ConvDesc.xlimit_cols
  // 1x1 stride-1 only: K over nseg > 1 tensors (ConvDesc.in_more / seg_c); `in` is segment 0
  const bf16_t* in1;
  const bf16_t* in2;
  const bf16_t* in3;
  int segc0, segc1, segc2, segc3, nseg;
  unsigned tap_mask[8];   // ConvDesc.tap_mask (0 = every tap)
  int xp_store;           // register epilogue: rows leave through a wave-private LDS tile as whole 128-byte lines (PT_CONV_XP)
  const int* blist;       // 4 x 64 patch only (ConvDesc.block_list): blist[0] live 32-column blocks, blist[1 + i] = image * (Wo / 32) + block; a workgroup takes two
};

// The storage format of activations and single-pass weights is act16.h's (bf16 in namespace pt_bf16, IEEE half in pt_f16); these are this
// file's names for its conversions.  pack_bf16x2 is ONE instruction (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32; the half format saturates through MODE.FP16_OVFL, act16.h) where
// an integer rounding takes nine -- the epilogues are VALU-bound (s_memtime phase stamps: 40 % of a short-K tile's time)
__device__ __forceinline__ float bf16_to_f32(uint32_t bits16) { return a16_to_f32(bits16); }
__device__ __forceinline__ uint32_t f32_to_bf16(float f) { return f32_to_a16(f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) { return pack_a16x2(a, b); }
__device__ __forceinline__ float bf16lo_f32(uint32_t pk) { return a16lo_f32(pk); }
__device__ __forceinline__ float bf16hi_f32(uint32_t pk) { return a16hi_f32(pk); }

// PT_PRECISION_F16X2: two bf16 values -> two fp16 values.  Exact (a bf16 value has 8 significant bits, fp16 holds 11) only inside fp16's
// NORMAL range 2^-14 <= |x| <= 65504: v_cvt_pkrtz clamps larger magnitudes to 65504 and truncates smaller ones toward zero (subnormals
// keep fewer bits, < 2^-24 becomes 0).  The mode therefore assumes activations below 6.5e4 -- true of every BN-folded net on this path
// (DB-ResNet18 activations are O(1..100)); it is an experiment outside the 1e-3 contract (tests/test_gpu_fullsize.py records its drift),
// not a tolerance mode.  The round-toward-zero pack is enough; the hi / lo halves of a pair are converted while the slice is staged.
__device__ __forceinline__ uint32_t bf16x2_to_f16x2(uint32_t v) {
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(v << 16), __uint_as_float(v & 0xFFFF0000u)));
}
__device__ __forceinline__ u32x4 bf16x8_to_f16x8(u32x4 v) {
  u32x4 r;
  r.x = bf16x2_to_f16x2(v.x); r.y = bf16x2_to_f16x2(v.y); r.z = bf16x2_to_f16x2(v.z); r.w = bf16x2_to_f16x2(v.w);
  return r;
}
// The lo halves are ~2^-9 of their values: below 2^-14 they would be fp16 subnormals, which the matrix pipe flushes (measured: without
// the scaling the mode read 1.2e-3 of the logit scale instead of 1.5e-4).  They are multiplied by 2^8 on the way in (exact), their K
// chunks run FIRST, and the accumulators are multiplied by 2^-8 once before the hi chunks follow.
#define PT_F16_LO_SCALE 256.0f
__device__ __forceinline__ uint32_t bf16x2_to_f16x2_scaled(uint32_t v) {
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(v << 16) * PT_F16_LO_SCALE, __uint_as_float(v & 0xFFFF0000u) * PT_F16_LO_SCALE));
}
__device__ __forceinline__ u32x4 bf16x8_to_f16x8_scaled(u32x4 v) {
  u32x4 r;
  r.x = bf16x2_to_f16x2_scaled(v.x); r.y = bf16x2_to_f16x2_scaled(v.y); r.z = bf16x2_to_f16x2_scaled(v.z); r.w = bf16x2_to_f16x2_scaled(v.w);
  return r;
}
// one 32x32x16 MFMA on 16-bit operands held as bf16x8 bit patterns: bf16, or fp16 in the F16X2 kernels
template <bool F16>
__device__ __forceinline__ f32x16 mma16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return mfma_32x32x16_a16(a, b, c);
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

// XCD-aware bijective remap of the flat block id (8 XCDs; block b is observed to run on XCD b % 8)
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (id >> 3);
}

// Block order for wide-N layers: (column group, spatial tile, column inside the group).  With all N tiles of a spatial
// tile consecutive, a 512 -> 7680 classifier streams its whole 7.8 MB weight matrix through every XCD's 4 MB L2 once per
// spatial tile (20 GB per launch); walking the columns in groups whose weights fit L2 re-reads the activations a few
// times instead (n_tiles / group) and keeps the weights resident.
__device__ __forceinline__ void tile_order(int L, int n_tiles, int n_group, int& nt, int& sp) {
  if (n_group <= 0 || n_group >= n_tiles) { nt = L % n_tiles; sp = L / n_tiles; return; }
  const int spatial = gridDim.x / n_tiles;
  const int per_group = n_group * spatial;
  const int ng = L / per_group;
  const int rem = L - ng * per_group;
  const int gcur = min(n_group, n_tiles - ng * n_group);
  sp = rem / gcur;
  nt = ng * n_group + rem - sp * gcur;
}

// Shared epilogue: fp32 tile [TH*32 pixels][64 ch] in LDS -> bias/residual/ReLU -> bf16 stores.
// b_second >= 0 (TW == 64, the 4 x 64 patch of conv_igemm_kernel<3, 1, 1>): the patch's columns 32 .. 63 are a 32-column block of image b_second at ox_second
// (another text line's, from the compacted block list; ox_second = Wo: no second block)
template <int TH, int TW, int NTHR = 256, int EXTRAS = 1>      // EXTRAS: 0 plain stores only, 1 every fused epilogue, 2 plain + fused pooling
__device__ __forceinline__ void epilogue_store(const ConvK& p, const float* stage, int tid, int b_first, int oy0, int ox0,
                                               int n0, int b_second = -1, int ox_second = 0) {
  // fused-head weights of this thread's 8 channels (idx & 7 == tid & 7 for every j)
  float hw[4][8];
  if (EXTRAS == 1 && p.head_w) {
    const int cgw = tid & 7;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      if (p.split) {
        const float* wf = reinterpret_cast<const float*>(p.head_w) + qd * 64 + cgw * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) hw[qd][k] = wf[k];
      } else {
        const u32x4 wv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.head_w) + qd * 64 + cgw * 8);
        const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) hw[qd][k] = bf16_to_f32((k & 1) ? (ww[k >> 1] >> 16) : (ww[k >> 1] & 0xFFFFu));
      }
    }
  }
  if (EXTRAS != 0 && p.pool) {
    // conv + BN + ReLU + max-pool in one pass: the pooled tile is (TH/2) x (TW/pw); every source value is biased, ReLU'd and
    // rounded exactly as it would have been stored, then the maximum is stored (identical to conv -> store -> pool)
    // pool 3: (2,1) with the pooled rows written as channel groups, [n][Wo][Ho/2 * C] (the layout the (2,1)-kernel conv4 of
    // the CRNN reads as a 1x1 GEMM, maxpool_kxk_kernel's h2c)
    const int pw = p.pool == 1 ? 2 : 1;
    const int PTW = TW / pw, PHo = p.Ho >> 1, PWo = p.Wo / pw;
    for (int idx = tid; idx < (TH / 2) * PTW * 8; idx += NTHR) {
      const int ppix = idx >> 3, cg = idx & 7;
      const int py = ppix / PTW, px = ppix - py * PTW;
      const bool second = b_second >= 0 && px >= 32 / pw;
      const int b = second ? b_second : b_first;
      const int oy = (oy0 >> 1) + py, ox = second ? ox_second / pw + px - 32 / pw : ox0 / pw + px;
      if (oy >= PHo || ox >= PWo) continue;
      const int n = n0 + cg * 8;
      const f32x4* bp = reinterpret_cast<const f32x4*>(p.bias + n);
      const f32x4 b0 = bp[0], b1 = bp[1];
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float m[8];
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < pw; ++dx) {
          const f32x4* sp = reinterpret_cast<const f32x4*>(stage + ((2 * py + dy) * TW + pw * px + dx) * 64 + cg * 8);
          const f32x4 v0 = sp[0], v1 = sp[1];
          const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float v = vv[k] + bb[k];
            if (p.relu == 1) v = fmaxf(v, 0.f);
            if (!p.split) v = bf16_to_f32(f32_to_bf16(v));
            m[k] = (dy == 0 && dx == 0) ? v : fmaxf(m[k], v);
          }
        }
      const u32x4 o = {pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]), pack_bf16x2(m[6], m[7])};
      size_t oo;
      int lo_off = p.out_lo_off;
      if (p.pool == 3) {
        const int Ct = PHo * p.N;
        oo = ((size_t)b * PWo + ox) * (p.split ? 2 * Ct : Ct) + oy * p.N + n;
        lo_off = Ct;
      } else {
        oo = (((size_t)b * PHo + oy) * PWo + ox) * p.out_cstride + p.out_coff + n;
      }
      *reinterpret_cast<u32x4*>(p.out + oo) = o;
      if (p.split) {
        const uint32_t ow[4] = {o.x, o.y, o.z, o.w};
        uint32_t lw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) lw[k] = pack_bf16x2(m[2 * k] - bf16lo_f32(ow[k]), m[2 * k + 1] - bf16hi_f32(ow[k]));
        const u32x4 ol = {lw[0], lw[1], lw[2], lw[3]};
        *reinterpret_cast<u32x4*>(p.out + oo + lo_off) = ol;
      }
    }
    return;
  }
  static_assert(NTHR % 8 == 0, "a thread keeps its channel group over the passes");
  const int cg = tid & 7, n = n0 + cg * 8;
  // the thread's 8 bias values: once, not once per pixel (the compiler may not hoist the loads over the stores)
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
  for (int j = 0; j < TH * TW * 8 / NTHR; ++j) {
    const int pix = (tid + j * NTHR) >> 3;
    const int ty = pix / TW, tx = pix % TW;
    const bool second = b_second >= 0 && tx >= 32;
    const int b = second ? b_second : b_first;
    const int oy = oy0 + ty, ox = second ? ox_second + tx - 32 : ox0 + tx;
    if (oy >= p.Ho || ox >= p.Wo) continue;
    const f32x4* sp = reinterpret_cast<const f32x4*>(stage + pix * 64 + cg * 8);
    f32x4 v0 = sp[0], v1 = sp[1];
    float v[8] = {v0.x + b0.x, v0.y + b0.y, v0.z + b0.z, v0.w + b0.w, v1.x + b1.x, v1.y + b1.y, v1.z + b1.z, v1.w + b1.w};
    if (p.res_mode) {
      const int rcs = p.split ? 2 * p.N : p.N;
      size_t ro;
      if (p.res_mode == 1)
        ro = (((size_t)b * p.Ho + oy) * p.Wo + ox) * rcs + n;
      else
        ro = (((size_t)b * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * rcs + n;
      u32x4 r = *reinterpret_cast<const u32x4*>(p.res + ro);
      v[0] += bf16_to_f32(r.x & 0xFFFFu); v[1] += bf16_to_f32(r.x >> 16);
      v[2] += bf16_to_f32(r.y & 0xFFFFu); v[3] += bf16_to_f32(r.y >> 16);
      v[4] += bf16_to_f32(r.z & 0xFFFFu); v[5] += bf16_to_f32(r.z >> 16);
      v[6] += bf16_to_f32(r.w & 0xFFFFu); v[7] += bf16_to_f32(r.w >> 16);
      if (p.split) {
        r = *reinterpret_cast<const u32x4*>(p.res + ro + p.N);
        v[0] += bf16_to_f32(r.x & 0xFFFFu); v[1] += bf16_to_f32(r.x >> 16);
        v[2] += bf16_to_f32(r.y & 0xFFFFu); v[3] += bf16_to_f32(r.y >> 16);
        v[4] += bf16_to_f32(r.z & 0xFFFFu); v[5] += bf16_to_f32(r.z >> 16);
        v[6] += bf16_to_f32(r.w & 0xFFFFu); v[7] += bf16_to_f32(r.w >> 16);
      }
    }
    if (EXTRAS == 1 && p.res_f32) {
      const f32x4* rp = reinterpret_cast<const f32x4*>(p.res_f32 + (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.out_cstride + p.out_coff + n);
      const f32x4 r0 = rp[0], r1 = rp[1];
      v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    }
    if (p.relu == 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
    } else if (EXTRAS == 1 && p.relu == 2) {   // hardswish: x * relu6(x + 3) / 6 (PicoDet's LCNet / CSP-PAN / head)
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = v[k] * fminf(fmaxf(v[k] + 3.f, 0.f), 6.f) / 6.f;
    } else if (EXTRAS == 1 && p.relu == 3) {   // nn.PReLU() with one shared slope (DB-ProxylessNAS, db_net/layers.py:696,722)
      const float sl = p.slope[0];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.f ? v[k] : sl * v[k];
    } else if (EXTRAS == 1 && p.relu == 4) {   // exact GELU, x * 0.5 * (1 + erf(x / sqrt(2))): ConvNext / ViT MLPs (cvit_model.hip)
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = v[k] * 0.5f * (1.f + erff(v[k] * 0.70710678118654752f));
    }
    u32x4 o, ol;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    if (p.split) {
      ol.x = pack_bf16x2(v[0] - bf16lo_f32(o.x), v[1] - bf16hi_f32(o.x));
      ol.y = pack_bf16x2(v[2] - bf16lo_f32(o.y), v[3] - bf16hi_f32(o.y));
      ol.z = pack_bf16x2(v[4] - bf16lo_f32(o.z), v[5] - bf16hi_f32(o.z));
      ol.w = pack_bf16x2(v[6] - bf16lo_f32(o.w), v[7] - bf16hi_f32(o.w));
    }
    if (EXTRAS == 1 && p.argmax_part) {
      // fused arg-max over classes (CTC greedy decode, modeling_ocr_recognition.py:168-171): best (value, index)
      // of this pixel's 64-class slice; ties keep the LOWEST class index, like torch.argmax
      float bv = v[0];
      int bi = n;
#pragma unroll
      for (int k = 1; k < 8; ++k)
        if (v[k] > bv) { bv = v[k]; bi = n + k; }
#pragma unroll
      for (int off = 1; off < 8; off <<= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (cg == 0) {
        const size_t row = ((size_t)b * p.Ho + oy) * p.Wo + ox;
        float2 pr;
        pr.x = bv;
        pr.y = __int_as_float(bi);
        reinterpret_cast<float2*>(p.argmax_part)[row * p.n_tiles + (n0 >> 6)] = pr;
      }
    } else if (EXTRAS == 1 && p.head_w) {
      // 8 consecutive lanes hold the 64 channels of one output pixel of quadrant `quad` (n0 == quad * 64)
      float xs[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t ow = k < 2 ? o.x : k < 4 ? o.y : k < 6 ? o.z : o.w;
        xs[k] = p.split ? v[k] : ((k & 1) ? bf16hi_f32(ow) : bf16lo_f32(ow));
      }
      float acc4[4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a = fmaf(xs[k], hw[qd][k], a);
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        a += __shfl_xor(a, 4);
        acc4[qd] = a;
      }
      if (cg < 4) {
        const int quad = n0 >> 6;
        const float lg = (cg == 0 ? acc4[0] : cg == 1 ? acc4[1] : cg == 2 ? acc4[2] : acc4[3]) + p.head_b[0];
        const int Y = 2 * oy + (quad >> 1), X = 2 * ox + (quad & 1);
        const size_t o = ((size_t)b * (4 * p.Ho) + 2 * Y + (cg >> 1)) * (size_t)(4 * p.Wo) + 2 * X + (cg & 1);
        if (p.head_logits) p.head_logits[o] = lg;
        if (p.head_prob) p.head_prob[o] = 1.f / (1.f + expf(-lg));
      }
    } else if (p.shuffle_cout) {
      const int quad = n0 / p.shuffle_cout;
      const int co = n0 - quad * p.shuffle_cout + cg * 8;
      const int OH = p.Ho * 2, OW = p.Wo * 2;
      const size_t oo = (((size_t)b * OH + 2 * oy + (quad >> 1)) * OW + 2 * ox + (quad & 1)) * p.out_cstride + p.out_coff + co;
      *reinterpret_cast<u32x4*>(p.out + oo) = o;
      if (p.split) *reinterpret_cast<u32x4*>(p.out + oo + p.out_lo_off) = ol;
    } else if (EXTRAS == 1 && p.n_valid && n >= p.n_valid) {
      // padded output channel: nothing to store
    } else if (EXTRAS == 1 && p.out_f32) {
      const size_t oo = (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.out_cstride + p.out_coff + n;
      f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
      *reinterpret_cast<f32x4*>(p.out_f32 + oo) = o0;
      *reinterpret_cast<f32x4*>(p.out_f32 + oo + 4) = o1;
    } else {
      const int f = p.rep;
      const int OH = p.Ho * f, OW = p.Wo * f;
      for (int fy = 0; fy < f; ++fy)
        for (int fx = 0; fx < f; ++fx) {
          const size_t oo = (((size_t)b * OH + oy * f + fy) * OW + ox * f + fx) * p.out_cstride + p.out_coff + n;
          *reinterpret_cast<u32x4*>(p.out + oo) = o;
          if (p.split) *reinterpret_cast<u32x4*>(p.out + oo + p.out_lo_off) = ol;
        }
    }
  }
}

// Register epilogue for kernels that run the MFMA with the WEIGHTS as its A operand (D = [channel][pixel]): lane (lx, q) owns
// pixel lx of a 32-pixel row-tile, register r of 32-channel block nb holds channel nb*32 + (r & 3) + 8 * (r >> 2) + 4 * q.
// bias / residual / activation / bf16 rounding happen where the accumulators are; one v_permlane32_swap per dword pair
// then gives every lane two 16-byte channel runs per block (q = 0: channels 0-7 and 16-23, q = 1: 8-15 and 24-31).
// Against the staged epilogue above (fp32 through LDS, two barriers per pass): no LDS, no barrier, ~1/3 of the
// instructions.  Used by the register-staged kernel's 1x1, stride-2 and short 3x3 launches (K is two to sixteen chunks there: a
// tile is mostly epilogue); plain layers only (no split / pool / fused head / arg-max / pixel shuffle / fp32 residual: the staged epilogue's full-row
// fp32 read-modify-write beats 16-byte pieces per lane, measured on the ConvNextViT residual GEMMs).
struct DirectBias { f32x4 v[2][4]; };
template <int NB>
__device__ __forceinline__ DirectBias direct_bias(const ConvK& p, int n0, int q) {
  DirectBias bs;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int g = 0; g < 4; ++g) bs.v[nb][g] = *reinterpret_cast<const f32x4*>(p.bias + n0 + nb * 32 + 8 * g + 4 * q);
  return bs;
}
// xp != null: the row-tile's 32 pixels x 64 channels go through a wave-private 4 KB LDS tile ([pixel][128 B], 16-byte pieces
// XOR-swizzled by the pixel) so that the global stores are whole 128-byte lines, 8 per wave instruction -- stored straight
// from the accumulator layout an instruction touches 32 lines with 16 bytes each.
template <int NB>
__device__ __forceinline__ void epilogue_direct_row(const ConvK& p, const f32x16 (&acc)[2], const DirectBias& bs, int b, int oy,
                                                    int ox0, int lx, int n0, int q, char* xp) {
  const int ox = ox0 + lx;
  const bool inside = oy < p.Ho && ox < p.Wo;      // both lanes of a (pixel, q) pair agree
  const bf16_t* rp = nullptr;
  if (p.res_mode == 1)
    rp = p.res + (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.N + n0 + 4 * q;
  else if (p.res_mode == 2)
    rp = p.res + (((size_t)b * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.N + n0 + 4 * q;
  const int f = p.rep, OW = p.Wo * f;
  const float sl = p.relu == 3 ? p.slope[0] : 0.f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      v[4 * g + 0] = acc[nb][4 * g + 0] + bs.v[nb][g].x;
      v[4 * g + 1] = acc[nb][4 * g + 1] + bs.v[nb][g].y;
      v[4 * g + 2] = acc[nb][4 * g + 2] + bs.v[nb][g].z;
      v[4 * g + 3] = acc[nb][4 * g + 3] + bs.v[nb][g].w;
    }
    if (rp && inside) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u32x2 rr = *reinterpret_cast<const u32x2*>(rp + nb * 32 + 8 * g);
        v[4 * g + 0] += bf16lo_f32(rr.x); v[4 * g + 1] += bf16hi_f32(rr.x);
        v[4 * g + 2] += bf16lo_f32(rr.y); v[4 * g + 3] += bf16hi_f32(rr.y);
      }
    }
    if (p.relu == 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = fmaxf(v[k], 0.f);
    } else if (p.relu == 2) {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = v[k] * fminf(fmaxf(v[k] + 3.f, 0.f), 6.f) * 0.16666667f;      // (the rounded reciprocal, like layout_kernels.hip's
                                                                                                        // hswish: the IEEE division is ten instructions per value)
    } else if (p.relu == 3) {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = v[k] > 0.f ? v[k] : sl * v[k];
    } else if (p.relu == 4) {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = v[k] * 0.5f * (1.f + erff(v[k] * 0.70710678118654752f));
    }
    if (p.out_f32) {
      // fp32 output (network outputs, offset / mask maps): the lane's runs of four channels are 16-byte stores as they are
      if (inside) {
        float* of = p.out_f32 + (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.out_cstride + p.out_coff + n0 + nb * 32 + 4 * q;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (!p.n_valid || n0 + nb * 32 + 8 * g + 4 * q < p.n_valid)
            *reinterpret_cast<f32x4*>(of + 8 * g) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
      }
      continue;
    }
    uint32_t d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
    // (d0, d1 | d2, d3) = channels 8g + 4q + (0..3) of g = 0, 1: swap the q = 1 half of (d0, d1) with the q = 0 half of (d2, d3)
    const u32x2 s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
    const u32x2 s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
    const u32x2 s2 = __builtin_amdgcn_permlane32_swap(d[4], d[6], false, false);
    const u32x2 s3 = __builtin_amdgcn_permlane32_swap(d[5], d[7], false, false);
    const u32x4 run0 = {s0.x, s1.x, s0.y, s1.y};     // channels nb*32 + 8q .. + 7
    const u32x4 run1 = {s2.x, s3.x, s2.y, s3.y};     // channels nb*32 + 16 + 8q .. + 7
    if (xp) {
      *reinterpret_cast<u32x4*>(xp + lx * 128 + (((nb * 4 + q) ^ (lx & 7)) << 4)) = run0;
      *reinterpret_cast<u32x4*>(xp + lx * 128 + (((nb * 4 + 2 + q) ^ (lx & 7)) << 4)) = run1;
    } else if (inside) {
      bf16_t* op = p.out + (((size_t)b * p.Ho * f + oy * f) * OW + ox * f) * p.out_cstride + p.out_coff + n0 + 8 * q + nb * 32;
      const bool ok0 = !p.n_valid || n0 + nb * 32 + 8 * q < p.n_valid;
      const bool ok1 = !p.n_valid || n0 + nb * 32 + 16 + 8 * q < p.n_valid;
      for (int fy = 0; fy < f; ++fy)
        for (int fx = 0; fx < f; ++fx) {
          bf16_t* o = op + ((size_t)fy * OW + fx) * p.out_cstride;
          if (ok0) *reinterpret_cast<u32x4*>(o) = run0;
          if (ok1) *reinterpret_cast<u32x4*>(o + 16) = run1;
        }
    }
  }
  if (xp) {
    // wave-private tile: the wave's own writes are visible to it after lgkmcnt(0) (in-order LDS), no barrier
    const int lane = q * 32 + lx, k = lane & 7;
    const bool okc = k < NB * 4 && (!p.n_valid || n0 + 8 * k < p.n_valid);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int P = (lane >> 3) + 8 * i;
      const u32x4 run = *reinterpret_cast<const u32x4*>(xp + P * 128 + ((k ^ (P & 7)) << 4));
      const int oxp = ox0 + P;
      if (okc && oy < p.Ho && oxp < p.Wo) {
        bf16_t* op = p.out + (((size_t)b * p.Ho * f + oy * f) * OW + oxp * f) * p.out_cstride + p.out_coff + n0 + 8 * k;
        for (int fy = 0; fy < f; ++fy)
          for (int fx = 0; fx < f; ++fx) *reinterpret_cast<u32x4*>(op + ((size_t)fy * OW + fx) * p.out_cstride) = run;
      }
    }
  }
}

template <int KS, int STRIDE, int GEOM = 0>
struct ConvCfg {
  // GEOM 0: 3x3/s1: 8x32 patch (2 MFMA row-tiles per wave); 1x1 and stride-2: 4x32 (smaller LDS image -> more
  //         workgroups per CU, which is what the bandwidth-bound 1x1 layers need)
  // GEOM 1: 4x64 patch for feature maps that are only <= 4 rows high (CRNN conv3.*, crnn/modeling_crnn.py:66-77)
  static constexpr int TW = GEOM ? 64 : 32;
  static constexpr int TH = GEOM ? 4 : ((STRIDE == 1 && KS == 3) ? 8 : 4);
  static constexpr int CT = TW / 32;            // 32-pixel MFMA row-tiles per patch row
  static constexpr int MT = TH * CT / 4;        // row-tiles per wave
  // a strided 1x1 conv reads only the pixels it samples: its patch is TH x TW pixels gathered at stride GS and laid out densely in LDS
  // (SS = 1); the 7 x 63 window of a 4 x 32 stride-2 tile would stage 3.4x the pixels it uses
  static constexpr int SS = KS == 1 ? 1 : STRIDE;      // pixel stride of an A fragment inside the LDS patch
  static constexpr int GS = KS == 1 ? STRIDE : 1;      // stride of the patch's pixels in the input map
  static constexpr int THIN = (TH - 1) * SS + KS;
  // GEOM 1: the patch is TWO 32-column blocks with their own halo columns (34 + 34), so that the blocks may come from different text lines
  static constexpr int TWIN = GEOM == 1 ? 68 : (TW - 1) * SS + KS;
  static constexpr int TAPS = KS * KS;
  static constexpr int XEVEN = (TWIN + 1) / 2;   // stride 2: number of even input columns of a patch row (stored first)
  // bytes per staged pixel / weight row: 32 bf16 + 16 B pad.  The stride-2 3x3 patch (9 x 65 pixels) plus its weight slice
  // is 93 KB at that pitch -- ONE workgroup per CU, one wave per SIMD, nothing to hide the K-slice hand-over behind.  There
  // the rows are un-padded (64 B) with the 16-byte slots XOR-swizzled by ((row >> 2) & 3) (conflict-free for 16 lanes on
  // consecutive rows, like the DMA kernel's images): 74 KB, two workgroups per CU.
  static constexpr bool SWZ = KS == 3 && STRIDE == 2;
  static constexpr int PIXB = SWZ ? 64 : 80;
  static constexpr int IN_BYTES = THIN * TWIN * PIXB;
  static constexpr int W_BYTES = TAPS * 64 * PIXB;
  static constexpr int STAGE_BYTES = TH * TW * 64 * 4;
  static constexpr int SMEM = (IN_BYTES + W_BYTES) > STAGE_BYTES ? (IN_BYTES + W_BYTES) : STAGE_BYTES;
  static constexpr int NP_IN = THIN * TWIN * 4;  // 16-byte pieces of the input patch
  static constexpr int NI = (NP_IN + 255) / 256;
  static constexpr int NP_W = TAPS * 64 * 4;
  static constexpr int NWP = NP_W / 256;
  static_assert(NP_W % 256 == 0, "weight slice must be a whole number of 256-thread passes");
};

// NHALF = 1: only the first 32 of the tile's 64 output columns are computed (layers with <= 32 real outputs, e.g. the
// 27-channel offset / mask convs of the deformable layers): half the MFMAs and half the B-fragment reads.
// DIRECT: the weights are the MFMA's A operand and the epilogue runs from the accumulators (epilogue_direct_row)
// F16 (PT_PRECISION_F16X2, split layers only): K = (x_hi, w) + (x_lo, w) with fp16 weight tiles; the activation halves are converted to
// fp16 on their way into LDS and the products run on v_mfma_f32_32x32x16_f16
template <int KS, int STRIDE, int GEOM, int NHALF = 2, bool DIRECT = false, bool F16 = false>
__global__ __launch_bounds__(256, KS == 1 ? 4 : 2) void conv_igemm_kernel(ConvK p) {
  a16_kernel_enter();
  using C = ConvCfg<KS, STRIDE, GEOM>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_in = smem;
  char* s_w = smem + C::IN_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;

  // reps > 1 (data-dependent extents, ConvDesc.ylimit): a workgroup walks `reps` consecutive tiles and stops at the first
  // one beyond the row limit -- the worst-case grid of a mostly empty map is reps times smaller
  const int reps = p.reps > 1 ? p.reps : 1;
  for (int rep = 0; rep < reps; ++rep) {
  int L, nt;
  if (reps == 1) {
    // (block list: the live pairs are the FRONT of the walk -- the XCD remap would hand all of them to the first XCDs; in launch order every
    // output-channel tile of a pair lands on another XCD instead)
    L = (GEOM == 1 && p.blist) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    tile_order(L, p.n_tiles, p.n_group, nt, L);
  } else {
    L = blockIdx.x * reps + rep;
    if (L >= p.total_tiles) break;
    nt = L % p.n_tiles;
    L /= p.n_tiles;
    if (rep) __syncthreads();                      // the previous tile's epilogue has finished with the LDS image
  }
  const int txi = L % p.tiles_x;
  L /= p.tiles_x;
  const int tyi = L % p.tiles_y;
  const int b = L / p.tiles_y;
  const int oy0 = tyi * C::TH;
  int ox0 = txi * C::TW;
  // GEOM 1: the patch's two 32-column blocks, (image, first output column); column Wo = no block.  From the compacted list of live blocks
  // (ragged text lines: a workgroup always has two full blocks to multiply, possibly of two lines), or the two halves of a 64-column tile
  int blk_b[2] = {b, b}, blk_x[2] = {ox0, ox0 + 32};
  if (GEOM == 1) {
    if (p.blist) {
      const int cnt = p.blist[0], bpl = p.Wo >> 5;
      if (2 * txi >= cnt) continue;
      const int e0 = p.blist[1 + 2 * txi], e1 = 2 * txi + 1 < cnt ? p.blist[2 + 2 * txi] : -1;
      blk_b[0] = e0 / bpl; blk_x[0] = (e0 - blk_b[0] * bpl) * 32;
      blk_b[1] = e1 >= 0 ? e1 / bpl : blk_b[0];
      blk_x[1] = e1 >= 0 ? (e1 - blk_b[1] * bpl) * 32 : p.Wo;
      ox0 = blk_x[0];
    } else {
      if (p.xlimit && ox0 >= p.xlimit[b]) continue;
      if (blk_x[1] >= p.Wo || (p.xlimit && blk_x[1] >= p.xlimit[b])) blk_x[1] = p.Wo;      // the second half is beyond the map or all padding response
    }
  }
  const int bq = GEOM == 1 ? blk_b[0] : b;         // image of the (first block of the) tile
  if (p.ylimit && oy0 >= *p.ylimit) break;         // uniform over the workgroup; later tiles of the walk are further down
  if (GEOM != 1 && p.xlimit && ox0 >= p.xlimit[b]) continue;    // ragged image: this tile is all padding response, filled by the caller
  if (p.xlimit_rows) {
    int mx = 0;
#pragma unroll
    for (int r = 0; r < C::TH; ++r)
      if (oy0 + r < p.Ho) mx = max(mx, p.xlimit_rows[oy0 + r]);
    if (ox0 >= mx) continue;
  }
  const int iy0 = oy0 * STRIDE - (KS / 2), ix0 = ox0 * STRIDE - (KS / 2);
  const int nchunks = p.split ? (F16 ? 2 : 3) * (p.Cin >> 5) : (p.Cin >> 5);
  const int in_cs = p.split ? 2 * p.Cin : p.Cin;   // channels per input pixel in memory
  const bf16_t* in_b = p.in + (size_t)bq * p.H * p.W * in_cs;
  const bf16_t* in_b1 = p.in + (size_t)blk_b[1] * p.H * p.W * in_cs;      // GEOM 1: the second block's image
  const bf16_t* wt = p.w + (size_t)nt * nchunks * (C::TAPS * 64 * 32);

  u32x4 rin[C::NI];
  u32x4 rw[C::NWP];

  auto prefetch = [&](int chunk) {
    // split mode: K chunks walk [x_hi | x_lo] against w_hi, then x_hi again against w_lo; F16: x_lo first, then x_hi, both against w
    int c0 = chunk << 5;
    if (F16) c0 = chunk < (p.Cin >> 5) ? p.Cin + c0 : c0 - p.Cin;
    else if (c0 >= in_cs) c0 -= in_cs;
    const bf16_t* src = in_b;
    int src_cs = in_cs;
    if (KS == 1 && STRIDE == 1 && p.nseg > 1) {
      // K over a concatenation of tensors: logical channel -> (segment, channel inside it); uniform over the workgroup
      int cc = chunk << 5, lo = 0;
      if (F16) {
        if (cc < p.Cin) lo = 1;
        else cc -= p.Cin;
      } else if (p.split) {
        if (cc >= 2 * p.Cin) cc -= 2 * p.Cin;
        else if (cc >= p.Cin) { cc -= p.Cin; lo = 1; }
      }
      const bf16_t* sp = p.in;
      int sc = p.segc0;
      if (cc >= sc) { cc -= sc; sp = p.in1; sc = p.segc1;
        if (cc >= sc) { cc -= sc; sp = p.in2; sc = p.segc2;
          if (cc >= sc) { cc -= sc; sp = p.in3; sc = p.segc3; } } }
      src_cs = p.split ? 2 * sc : sc;
      src = sp + (size_t)b * p.H * p.W * src_cs;
      c0 = cc + (lo ? sc : 0);
    }
#pragma unroll
    for (int j = 0; j < C::NI; ++j) {
      const int idx = tid + j * 256;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (idx < C::NP_IN) {
        const int pix = idx >> 2, part = idx & 3;
        const int iy = pix / C::TWIN, ix = pix - iy * C::TWIN;
        int gy = iy0 + iy * C::GS, gx = ix0 + ix * C::GS;
        const bf16_t* sp_ = src;
        if (GEOM == 1) {       // sub-patch ix / 34 with its own halo: columns blk_x[j] - 1 .. blk_x[j] + 32 of image blk_b[j]
          const int j = ix >= 34;
          gx = blk_x[j] - 1 + ix - 34 * j;
          if (blk_x[j] >= p.Wo) gx = -1;
          if (j) sp_ = in_b1;
        }
        if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
          v = *reinterpret_cast<const u32x4*>(sp_ + ((size_t)gy * p.W + gx) * src_cs + c0 + part * 8);
      }
      rin[j] = v;
    }
    const bf16_t* wc = wt + (size_t)chunk * (C::TAPS * 64 * 32);
#pragma unroll
    for (int j = 0; j < C::NWP; ++j) {
      const int idx = tid + j * 256;
      // NHALF = 1: the MFMAs read weight rows 0 .. 31 of every tap only -- rows 32 .. 63 (the second half of each 256-piece tap: threads 128 .. 255) are
      // the zero padding of a <= 32-output layer and are neither fetched nor written to LDS
      if (NHALF == 2 || tid < 128) rw[j] = *reinterpret_cast<const u32x4*>(wc + idx * 8);
    }
  };
  auto commit = [&](int chunk) {
    const bool lo_half = F16 && chunk < (p.Cin >> 5);
#pragma unroll
    for (int j = 0; j < C::NI; ++j) {
      const int idx = tid + j * 256;
      if (idx < C::NP_IN) {
        int slot = idx >> 2;
        if (C::SS == 2) {       // even columns first, then the odd ones: the 32 pixels of an A fragment (input columns 2*lx + s)
          const int iy = slot / C::TWIN, ix = slot - iy * C::TWIN;   // are then CONSECUTIVE 80-byte slots, as at stride 1 --
          slot = iy * C::TWIN + (ix & 1) * C::XEVEN + (ix >> 1);      // no 2-pixel stride, no bank conflicts on ds_read_b128
        }
        *reinterpret_cast<u32x4*>(s_in + slot * C::PIXB + ((C::SWZ ? ((idx & 3) ^ ((slot >> 2) & 3)) : (idx & 3)) * 16)) =
            F16 ? (lo_half ? bf16x8_to_f16x8_scaled(rin[j]) : bf16x8_to_f16x8(rin[j])) : rin[j];
      }
    }
#pragma unroll
    for (int j = 0; j < C::NWP; ++j) {
      const int idx = tid + j * 256;
      if (NHALF == 2 || tid < 128)
        *reinterpret_cast<u32x4*>(s_w + (idx >> 2) * C::PIXB + ((C::SWZ ? ((idx & 3) ^ ((idx >> 4) & 3)) : (idx & 3)) * 16)) = rw[j];
    }
  };

  f32x16 acc[C::MT][2];
#pragma unroll
  for (int m = 0; m < C::MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // MFMA row-tile t = wave * MT + m covers patch row t / CT, columns (t % CT) * 32 .. + 31
  const char* a_base[C::MT];
  int a_slot[C::MT];
#pragma unroll
  for (int m = 0; m < C::MT; ++m) {
    const int t = wave * C::MT + m;
    a_slot[m] = ((t / C::CT) * C::SS) * C::TWIN + ((t % C::CT) * (GEOM == 1 ? 34 : 32) + lx);      // SS = 2: even columns are stored first, consecutively
    a_base[m] = s_in + a_slot[m] * C::PIXB + (C::SWZ ? 0 : q * 16);
  }
  const char* b_base = s_w + lx * C::PIXB + (C::SWZ ? 0 : q * 16);
  const int b_sw = (lx >> 2) & 3;      // SWZ: weight row tap * 64 (+ 32) + lx -> ((row >> 2) & 3) == ((lx >> 2) & 3) for every tap

  const unsigned tmask = (KS == 3 && STRIDE == 1 && nt < 8 && p.tap_mask[nt]) ? p.tap_mask[nt] : 0x1FFu;
  // ragged lines on the 4 x 64 patch: a wave's 32-column row-tile that starts at or behind the line's limit is not multiplied (the caller fills
  // from the limit rounded up to 32 columns, not 64: lines are ~1/4 text, and the last tile of a line was half padding on average)
  bool dead[C::MT];
#pragma unroll
  for (int m = 0; m < C::MT; ++m) dead[m] = GEOM == 1 && blk_x[(wave * C::MT + m) % C::CT] >= p.Wo;
  prefetch(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();  // everyone is done reading the previous slice
    commit(c);
    __syncthreads();
    if (c + 1 < nchunks) prefetch(c + 1);
    if (F16 && c == (p.Cin >> 5)) {      // the scaled lo chunks are in: back to the scale of the hi products
#pragma unroll
      for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int n = 0; n < NHALF; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] *= (1.0f / PT_F16_LO_SCALE);
    }
#pragma unroll
    for (int r = 0; r < KS; ++r) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int tap = r * KS + s;
        if (KS == 3 && STRIDE == 1 && !((tmask >> tap) & 1u)) continue;      // uniform: this tile's weights are zero on the tap
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int b_off = C::SWZ ? (((q + 2 * kk) ^ b_sw) * 16) : kk * 32;
          const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(b_base + (tap * 64) * C::PIXB + b_off);
          bf16x8 b1 = b0;
          if (NHALF == 2) b1 = *reinterpret_cast<const bf16x8*>(b_base + (tap * 64 + 32) * C::PIXB + b_off);
#ifdef PT_SETPRIO
          __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
          for (int m = 0; m < C::MT; ++m) {
            if (GEOM == 1 && dead[m]) continue;      // wave-uniform
            // stride 2: tap s reads input column 2*lx + s = slot lx (s = 0), XEVEN + lx (s = 1), lx + 1 (s = 2)
            const int soff = C::SS == 2 ? ((s & 1) * C::XEVEN + (s >> 1)) : s;
            const int a_off = C::SWZ ? (((q + 2 * kk) ^ (((a_slot[m] + r * C::TWIN + soff) >> 2) & 3)) * 16) : kk * 32;
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_base[m] + (r * C::TWIN + soff) * C::PIXB + a_off);
            if (DIRECT) {
              acc[m][0] = mma16<F16>(b0, a, acc[m][0]);      // D = [channel][pixel]
              if (NHALF == 2) acc[m][1] = mma16<F16>(b1, a, acc[m][1]);
            } else {
              acc[m][0] = mma16<F16>(a, b0, acc[m][0]);
              if (NHALF == 2) acc[m][1] = mma16<F16>(a, b1, acc[m][1]);
            }
          }
#ifdef PT_SETPRIO
          __builtin_amdgcn_s_setprio(0);
#endif
        }
      }
    }
  }

  if (DIRECT) {
    // ---- epilogue from the accumulators: lane (lx, q) owns pixel lx of its row-tile ----
    const DirectBias bs = direct_bias<NHALF>(p, nt * 64, q);
    char* xp = nullptr;
    if (p.xp_store) {      // uniform: the operand images are dead once every wave has left the K loop
      __syncthreads();
      xp = smem + wave * 4096;
    }
#pragma unroll
    for (int m = 0; m < C::MT; ++m) {
      const int t = wave * C::MT + m;
      epilogue_direct_row<NHALF>(p, acc[m], bs, b, oy0 + t / C::CT, ox0 + (t % C::CT) * 32, lx, nt * 64, q, xp);
    }
    continue;      // next tile of a rep walk (none for the layers that take this path)
  }
  // ---- epilogue through LDS (fp32 [pixel][64]) ----
  __syncthreads();
  float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int m = 0; m < C::MT; ++m) {
    const int t = wave * C::MT + m;
    const int pbase = (t / C::CT) * C::TW + (t % C::CT) * 32;
#pragma unroll
    for (int n = 0; n < NHALF; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tx = (r & 3) + 8 * (r >> 2) + 4 * q;
        stage[(pbase + tx) * 64 + n * 32 + lx] = acc[m][n][r];
      }
  }
  __syncthreads();
  if (GEOM == 1) epilogue_store<C::TH, C::TW>(p, stage, tid, bq, oy0, ox0, nt * 64, blk_b[1], blk_x[1]);
  else epilogue_store<C::TH, C::TW>(p, stage, tid, b, oy0, ox0, nt * 64);
  }   // rep
}

// ---------------------------------------------------------------------------------------------------
// 1x1 stride-1 layers with a plain epilogue, K staged KG 32-channel chunks at a time.  conv_igemm_kernel<1, 1> hands its 4 x 32 x 64 tile one
// 32-channel chunk per barrier pair: 12 KB in flight per workgroup and four MFMAs per wave between two barriers -- a 128 -> 128 layer is four
// load latencies in a row per tile and ran at 2.2 TB/s.  Here a stage is KG chunks (KG = 4: the whole K of a 128-channel layer is requested at
// once, one barrier pair per tile); same tile, same chunk order inside the accumulators (bit-identical), same register epilogue.
// ---------------------------------------------------------------------------------------------------
template <int KG>
struct Conv1WideCfg {
  static constexpr int PIXB = 64 * KG + 16;       // 36 / 68 dwords: a quarter-wave's ds_read_b128 covers every bank twice
  static constexpr int IN_BYTES = 128 * PIXB;
  static constexpr int W_BYTES = 64 * PIXB;
  static constexpr int SMEM = IN_BYTES + W_BYTES > 16384 ? IN_BYTES + W_BYTES : 16384;      // (16 KB: the xp_store tiles of the epilogue)
};

template <int KG>
__global__ __launch_bounds__(256, KG == 4 ? 3 : 4) void conv1x1_wide_kernel(ConvK p) {
  a16_kernel_enter();
  using C = Conv1WideCfg<KG>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_in = smem;
  char* s_w = smem + C::IN_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  int L, nt;
  L = xcd_remap(blockIdx.x, gridDim.x);
  tile_order(L, p.n_tiles, p.n_group, nt, L);
  const int txi = L % p.tiles_x;
  L /= p.tiles_x;
  const int tyi = L % p.tiles_y;
  const int b = L / p.tiles_y;
  const int oy0 = tyi * 4, ox0 = txi * 32;
  if (p.xlimit && ox0 >= p.xlimit[b]) return;
  if (p.xlimit_rows) {
    int mx = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (oy0 + r < p.Ho) mx = max(mx, p.xlimit_rows[oy0 + r]);
    if (ox0 >= mx) return;
  }
  const int nstages = (p.Cin >> 5) / KG;
  const bf16_t* wt = p.w + (size_t)nt * (p.Cin >> 5) * (64 * 32);

  u32x4 rin[2 * KG];
  u32x4 rw[KG];
  // a thread's piece of a chunk: 16-byte part tid & 3 of pixel (row) pix0.  Consecutive groups of four lanes take pixels p and p + 4, not p and p + 1: a
  // ds_write_b128 is served eight lanes at a time, a pixel row is 68 (36) dwords -- four banks further than its neighbour -- so neighbours overlap in 12 of
  // their 16 banks and rows four apart do not (the 1x1 kernels spent a third of their LDS cycles in bank conflicts: profiles/r06/lds_activity_all_kernels.txt)
  const int t4 = tid >> 2, part = tid & 3;
  const int pix0 = (t4 & ~7) | ((t4 & 1) << 2) | ((t4 >> 1) & 3);
  const int gy0 = oy0 + (pix0 >> 5), gx = ox0 + (pix0 & 31);      // j = 1: two rows further down
  const bool in0 = gy0 < p.H && gx < p.W, in1 = gy0 + 2 < p.H && gx < p.W;
  auto prefetch = [&](int stage) {
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      int cc = (stage * KG + g) << 5;
      const bf16_t* sp = p.in;
      int sc = p.Cin;
      if (p.nseg > 1) {      // K over a concatenation of tensors: logical channel -> (segment, channel inside it); uniform over the workgroup
        sc = p.segc0;
        if (cc >= sc) { cc -= sc; sp = p.in1; sc = p.segc1;
          if (cc >= sc) { cc -= sc; sp = p.in2; sc = p.segc2;
            if (cc >= sc) { cc -= sc; sp = p.in3; sc = p.segc3; } } }
      }
      const bf16_t* src = sp + ((size_t)b * p.H * p.W) * sc + cc + part * 8;
      u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
      if (in0) v0 = *reinterpret_cast<const u32x4*>(src + ((size_t)gy0 * p.W + gx) * sc);
      if (in1) v1 = *reinterpret_cast<const u32x4*>(src + ((size_t)(gy0 + 2) * p.W + gx) * sc);
      rin[2 * g] = v0;
      rin[2 * g + 1] = v1;
    }
    const bf16_t* wc = wt + (size_t)stage * KG * (64 * 32);
#pragma unroll
    for (int g = 0; g < KG; ++g) rw[g] = *reinterpret_cast<const u32x4*>(wc + (g * 256 + pix0 * 4 + part) * 8);
  };
  auto commit = [&]() {
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      *reinterpret_cast<u32x4*>(s_in + pix0 * C::PIXB + g * 64 + part * 16) = rin[2 * g];
      *reinterpret_cast<u32x4*>(s_in + (pix0 + 64) * C::PIXB + g * 64 + part * 16) = rin[2 * g + 1];
      *reinterpret_cast<u32x4*>(s_w + pix0 * C::PIXB + g * 64 + part * 16) = rw[g];
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const char* a_base = s_in + (wave * 32 + lx) * C::PIXB + q * 16;
  const char* b_base = s_w + lx * C::PIXB + q * 16;

  prefetch(0);
  for (int st = 0; st < nstages; ++st) {
    if (st) __syncthreads();      // everyone is done reading the previous stage
    commit();
    __syncthreads();
    if (st + 1 < nstages) prefetch(st + 1);
#pragma unroll
    for (int k = 0; k < 2 * KG; ++k) {
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(b_base + k * 32);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(b_base + 32 * C::PIXB + k * 32);
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_base + k * 32);
      acc[0] = mma16<false>(b0, a, acc[0]);      // D = [channel][pixel]
      acc[1] = mma16<false>(b1, a, acc[1]);
    }
  }
  const DirectBias bs = direct_bias<2>(p, nt * 64, q);
  char* xp = nullptr;
  if (p.xp_store) {      // uniform: the operand images are dead once every wave has left the K loop
    __syncthreads();
    xp = smem + wave * 4096;
  }
  epilogue_direct_row<2>(p, acc, bs, b, oy0 + wave, ox0, lx, nt * 64, q, xp);
}

// ---------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution, LDS-DMA pipeline with 16-channel K-slices ("v3").
// Fill traffic per FLOP is what limits v1/v2 (PMC analysis), so v3 makes the tile as large as two
// K-slices in LDS allow:  NT == 1: 32x32 pixels x 64 channels;  NT == 2: 16x32 pixels x 128 channels.
// NWV = 8 waves (or 4: a 16x32 x 64 tile, two workgroups per CU -- see the dispatch rule in pt_launch_conv); each wave owns
// 4 MFMA row-tiles x 64 channels (acc 4x2, 6 ds_read_b128 per 8 MFMAs); a K-slice is
// 16 channels (one MFMA k-step per tap), ~56 KB for both operands, double buffered; rows are 32 B with the two
// 16-byte halves swapped on rows with bit 3 set (applied on the DMA source and on the ds_read address).
// Weights are read straight from the 32-channel tiling of v1 (half of every 64-byte row per slice).
// ---------------------------------------------------------------------------------------------------
// S = 2 (stride-2 layers with N % 128 == 0: the ResNet / DLA down-sampling convs): a wave owns TWO row-tiles, the workgroup 8 x 32 output pixels x 128
// channels from a 17 x 65 input patch whose rows are stored even columns first (tap s of output pixel lx reads column 2 lx + s: slot lx,
// 33 + lx, lx + 1 -- 32 consecutive slots per fragment, as at stride 1).  The register-staged stride-2 tile (4 x 32 x 64, a 37 KB weight slice
// per 36 MFMAs of a wave) ran these layers at 16-25 % of the peak.
template <int NT, int NWV = 8, int S = 1>
struct Dma16Cfg {
  static constexpr int NTHR = 64 * NWV;
  static constexpr int MT = S == 1 ? 4 : 2;                   // MFMA row-tiles (patch rows) per wave
  static constexpr int TH = (NT == 1 ? NWV : NWV / 2) * MT, TW = 32;
  static constexpr int NW = 64 * NT;                          // output channels per workgroup
  static constexpr int THIN = (TH - 1) * S + 3, TWIN = (TW - 1) * S + 3;
  static constexpr int XEVEN = (TWIN + 1) / 2;                // S = 2: even input columns of a patch row (stored first)
  static constexpr int NPIX = THIN * TWIN;                    // 1156 / 612; S = 2: 1105
  static constexpr int IN_BYTES = NPIX * 32;
  static constexpr int W_BYTES = 9 * NW * 32;
  static constexpr int BUF_BYTES = IN_BYTES + W_BYTES;        // 55424 / 56448
  static constexpr int PASS_ROWS = NT == 1 ? TH / 2 : TH;    // patch rows per epilogue pass (half of the waves / one 64-channel half)
  static constexpr int STAGE_BYTES = PASS_ROWS * 32 * 64 * 4; // one epilogue pass: 64 channels fp32
  static constexpr int SMEM = 2 * BUF_BYTES > STAGE_BYTES ? 2 * BUF_BYTES : STAGE_BYTES;
  static constexpr int IN_UNITS = NPIX * 2;
  static constexpr int IN_INSTR = (IN_UNITS + 63) / 64;
  static constexpr int W_UNITS = 9 * NW * 2;
  static constexpr int W_INSTR = W_UNITS / 64;
  static constexpr int IN_SLOTS = (IN_INSTR + NWV - 1) / NWV;
  static constexpr int W_SLOTS = (W_INSTR + NWV - 1) / NWV;
};

template <int NT, int NWV, int S = 1>
__global__ __launch_bounds__(64 * NWV, 2) void conv3x3_dma16_kernel(ConvK p, const bf16_t* __restrict__ zero_page) {
  a16_kernel_enter();
  using C = Dma16Cfg<NT, NWV, S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = lane & 31, qh = lane >> 5;
  const int wm = NT == 1 ? wave : (wave >> 1);   // which group of 4 patch rows
  const int wn = NT == 1 ? 0 : (wave & 1);       // which 64-channel half

  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int nb = L % p.n_tiles;                  // n_tiles counts NW-wide blocks here
  L /= p.n_tiles;
  const int txi = L % p.tiles_x;
  L /= p.tiles_x;
  const int tyi = L % p.tiles_y;
  const int b = L / p.tiles_y;
  const int oy0 = tyi * C::TH, ox0 = txi * C::TW;
  const int in_cs = p.split ? 2 * p.Cin : p.Cin;
  const int nch32 = p.split ? 3 * (p.Cin >> 5) : (p.Cin >> 5);
  const int nslices = nch32 * 2;
  const bf16_t* in_b = p.in + (size_t)b * p.H * p.W * in_cs;

  const bf16_t* src_in[C::IN_SLOTS];
  bool on_in[C::IN_SLOTS];
#pragma unroll
  for (int j = 0; j < C::IN_SLOTS; ++j) {
    const int k = wave + NWV * j;
    const int U = k * 64 + lane;
    on_in[j] = (k < C::IN_INSTR) && (U < C::IN_UNITS);
    const int pix = U >> 1;
    const int q = (U & 1) ^ ((pix >> 3) & 1);
    const int iy = pix / C::TWIN, sx = pix - iy * C::TWIN;
    const int ix = S == 1 ? sx : (sx < C::XEVEN ? 2 * sx : 2 * (sx - C::XEVEN) + 1);     // slot of the row -> input column of the patch
    const int gy = oy0 * S - 1 + iy, gx = ox0 * S - 1 + ix;
    const bool inside = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    src_in[j] = inside ? in_b + ((size_t)gy * p.W + gx) * in_cs + q * 8 : zero_page;
  }
  // weight unit U -> row = U >> 1 = tap * NW + n, half = (U & 1) ^ ((row >> 3) & 1); element offset inside one
  // 32-channel chunk of the v1 tiling [N/64][Cin/32][9][64][32], relative to (nt64 = NT * nb, chunk32 = 0)
  // a DMA instruction moves 32 weight rows of ONE tap and ONE 64-channel output tile: instructions whose tap is masked out for that
  // tile (phase convolutions: 5 of 9) are not issued -- their LDS rows are never read.  Weights are 36 of the 55 KB a 128-wide
  // slice stages, and two workgroups per CU fill at ~54 B per clock against the ~64 the L2 delivers: the phase convs were fill-bound
  int src_w[C::W_SLOTS];
  bool on_w[C::W_SLOTS];
#pragma unroll
  for (int j = 0; j < C::W_SLOTS; ++j) {
    const int k = wave + NWV * j;
    const int U = k * 64 + lane;
    const int row = U >> 1;
    const int tap = row / C::NW, n = row - tap * C::NW;
    const int hq = (U & 1) ^ ((row >> 3) & 1);
    src_w[j] = (n >> 6) * nch32 * (9 * 64 * 32) + (tap * 64 + (n & 63)) * 32 + hq * 8;
    const int ktap = (k * 32) / C::NW, kt64 = NT * nb + (((k * 32) % C::NW) >> 6);      // wave-uniform
    const unsigned km = (kt64 < 8 && p.tap_mask[kt64]) ? p.tap_mask[kt64] : 0x1FFu;
    on_w[j] = k < C::W_INSTR && ((km >> ktap) & 1u);
  }
  const bf16_t* wt = p.w + (size_t)(NT * nb) * nch32 * (9 * 64 * 32);

  auto issue = [&](int slice, int buf) {
    int c0 = slice << 4;
    if (c0 >= in_cs) c0 -= in_cs;
    if (c0 >= in_cs) c0 -= in_cs;
    char* lds_in = smem + buf * C::BUF_BYTES;
    char* lds_w = lds_in + C::IN_BYTES;
    const bf16_t* wc = wt + (size_t)(slice >> 1) * (9 * 64 * 32) + (slice & 1) * 16;
#pragma unroll
    for (int j = 0; j < C::IN_SLOTS; ++j) {
      if (on_in[j])
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_in[j] + c0),
                                         (__attribute__((address_space(3))) void*)(lds_in + (wave + NWV * j) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < C::W_SLOTS; ++j) {
      if (on_w[j])
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wc + src_w[j]),
                                         (__attribute__((address_space(3))) void*)(lds_w + (wave + NWV * j) * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[C::MT][2];
#pragma unroll
  for (int m = 0; m < C::MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int pa0 = (S * C::MT * wm) * C::TWIN + lx;                      // pixel of row-tile 0, tap (0,0)
  const int boff = (wn * 64 + lx) * 32 + ((qh ^ ((lx >> 3) & 1)) << 4);  // weight row n = wn*64 + nt*32 + lx

  const int t64 = NT * nb + (NT == 2 ? wn : 0);            // this wave's 64-channel output tile
  const unsigned tmask = (t64 < 8 && p.tap_mask[t64]) ? p.tap_mask[t64] : 0x1FFu;
  issue(0, 0);
  for (int c = 0; c < nslices; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c + 1 < nslices) issue(c + 1, (c + 1) & 1);
    const char* s_in = smem + (c & 1) * C::BUF_BYTES;
    const char* s_w = s_in + C::IN_BYTES;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int tap = r * 3 + s;
        if (!((tmask >> tap) & 1u)) continue;      // wave-uniform: this wave's 64 output channels have zero weights on the tap
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(s_w + (tap * C::NW) * 32 + boff);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(s_w + (tap * C::NW + 32) * 32 + boff);
        const int soff = S == 1 ? s : ((s & 1) * C::XEVEN + (s >> 1));
#pragma unroll
        for (int m = 0; m < C::MT; ++m) {
          const int pp = pa0 + (S * m + r) * C::TWIN + soff;
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(s_in + pp * 32 + ((qh ^ ((pp >> 3) & 1)) << 4));
          acc[m][0] = mfma_32x32x16_a16(a, b0, acc[m][0]);
          acc[m][1] = mfma_32x32x16_a16(a, b1, acc[m][1]);
        }
      }
    }
  }

  // epilogue in two passes of PASS_ROWS x 32 pixels x 64 channels (fp32 in LDS)
  float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    const bool mine = NT == 1 ? ((wave / (NWV / 2)) == pass) : (wn == pass);
    if (mine) {
      const int rbase = NT == 1 ? C::MT * (wave % (NWV / 2)) : C::MT * wm;   // local patch row inside this pass' rows
#pragma unroll
      for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int tx = (r & 3) + 8 * (r >> 2) + 4 * qh;
            stage[((rbase + m) * 32 + tx) * 64 + n * 32 + lx] = acc[m][n][r];
          }
    }
    __syncthreads();
    if (NT == 1)
      epilogue_store<C::PASS_ROWS, 32, C::NTHR, 0>(p, stage, tid, b, oy0 + C::PASS_ROWS * pass, ox0, nb * 64);
    else
      epilogue_store<C::PASS_ROWS, 32, C::NTHR, 0>(p, stage, tid, b, oy0, ox0, (nb * 2 + pass) * 64);
  }
}

// ---------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution, LDS-DMA pipeline with 16-channel K-slices and a SOFTWARE-PIPELINED tap loop ("v4").
// Same tiles, same K order inside the accumulators (bit-identical results) and the same two-buffer slice ring as conv3x3_dma16_kernel; what
// changes is what the matrix pipe waits for.  Counters on v3 (profiles/r06/conv3x3_stalls.txt): 37-42 % MFMA-busy.  Its ISA shows why: the
// per-tap mask test makes every tap a basic block, so a tap is {4 ds_read_b128, lgkmcnt(0), 4 MFMA, 2 ds_read_b128, lgkmcnt(0), 4 MFMA} --
// two exposed LDS round trips per 8 MFMAs -- and the seven LDS-DMA instructions of the next slice are issued in one burst behind the barrier
// while neither wave of the SIMD feeds the pipe.  Here:
//   * the slice body is straight-line code (no tap masks: masked layers stay on v3): the fragments of tap t+1 are requested before the
//     MFMAs of tap t are issued, so a read has eight MFMAs (256 cycles of the pipe) to come back;
//   * the image's 16-byte halves are swizzled by the patch COLUMN instead of the linear pixel index: a fragment address is one of three
//     per-lane bases (tap column s) plus an immediate (78 address VGPRs in v3, 4 here);
//   * every wave issues exactly SLOTS DMA instructions per slice, one behind each of the first taps' MFMA groups (the input and weight
//     instructions are one list; a slot past its end repeats the last instruction), out-of-image lanes read the zero page: no exec masks,
//     no branches between the MFMAs.
// ---------------------------------------------------------------------------------------------------
#ifndef PT_PIPE_INTERLEAVE
#define PT_PIPE_INTERLEAVE 1
#endif
// Scheduling groups of one tap of the pipelined loop: NM MFMAs of this tap, NR ds_reads (the NEXT tap's fragments) and ND LDS-DMA requests, in
// the order M R M R ... M [D] M: every read and every DMA request is issued in the shadow of an MFMA that is already in the pipe (a wave that
// issues its six reads and a DMA request back to back leaves the pipe ~100 cycles without work unless its SIMD partner happens to be out of phase)
template <int NR, int NM, int ND>
__device__ __forceinline__ void tap_groups() {
#if PT_PIPE_INTERLEAVE
  static_for<NM>([&](auto i_c) {
    constexpr int i = decltype(i_c)::value;
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if constexpr (i >= NR && i - NR < ND) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
  });
  // (more requests than MFMAs to hide them behind -- the stride-2 tile's 4 MFMAs per tap: the rest follow the last MFMA)
  if constexpr (NR > NM) __builtin_amdgcn_sched_group_barrier(0x100, NR - NM, 0);
  if constexpr (NR + ND > NM) __builtin_amdgcn_sched_group_barrier(0x020, NR >= NM ? ND : NR + ND - NM, 0);
#else
  if constexpr (NR > 0) __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
  if constexpr (ND > 0) __builtin_amdgcn_sched_group_barrier(0x020, ND, 0);
#endif
}

// BR (block rows): the tile's TH x 32 output pixels are NBLK = TH / BR blocks of BR rows x 32 columns, each with its own halo ((BR + 2) x 34 input
// pixels) and its own (image, first column).  BR == TH (0 = default) is the ordinary tile.  BR = 4 / 8 serve maps that are exactly BR rows high --
// the CRNN conv stack's 4 x 160 and 8 x 160 maps (crnn/modeling_crnn.py:66-77), K = 1 152 ... 4 608 -- whose blocks come from the compacted list of
// live 32-column blocks of the ragged text lines (ConvDesc.block_list), or from NBLK consecutive column blocks of one line.
template <int NT, int NWV, int BR_ = 0, int S_ = 1>
struct PipeCfg {
  static constexpr int NTHR = 64 * NWV;
  static constexpr int S = S_;                                // stride (2: the ResNet / DLA down-sampling convs; ordinary tiles only)
  static constexpr int MT = S == 1 ? 4 : 2;                   // MFMA row-tiles (patch rows) per wave
  static constexpr int TH = (NT == 1 ? NWV : NWV / 2) * MT, TW = 32;
  static constexpr int BR = BR_ ? BR_ : TH;
  static constexpr int NBLK = TH / BR, PR = (BR - 1) * S + 3; // blocks per tile, patch rows per block
  static_assert(TH % BR == 0 && BR % MT == 0, "a wave's rows lie inside one block");
  static_assert(S == 1 || NBLK == 1, "stride 2: ordinary tiles");
  static constexpr int NW = 64 * NT;                          // output channels per workgroup
  static constexpr int THIN = NBLK * PR, TWIN = (TW - 1) * S + 3;
  // stride 2: a patch row is stored even input columns first (XEVEN of them), then the odd ones: tap s of output pixel lx reads input column 2 lx + s,
  // i.e. slot lx (s = 0), XEVEN + lx (s = 1), lx + 1 (s = 2) -- 32 consecutive slots per fragment, as at stride 1
  static constexpr int XEVEN = (TWIN + 1) / 2;
  static constexpr int NPIX = THIN * TWIN;
  static constexpr int IN_UNITS = NPIX * 2;
  static constexpr int IN_INSTR = (IN_UNITS + 63) / 64;
  static constexpr int W_INSTR = 9 * NW * 2 / 64;
  static constexpr int T_INSTR = IN_INSTR + W_INSTR;
  static constexpr int SLOTS = (T_INSTR + NWV - 1) / NWV;
  static constexpr int IN_BYTES = IN_INSTR * 1024;            // (the last instruction's tail lanes land in padding)
  static constexpr int W_BYTES = W_INSTR * 1024;
  static constexpr int BUF_BYTES = IN_BYTES + W_BYTES;
  static constexpr int PASS_ROWS = NT == 1 ? TH / 2 : TH;
  static constexpr int STAGE_BYTES = PASS_ROWS * 32 * 64 * 4;
  static constexpr int SMEM = 2 * BUF_BYTES > STAGE_BYTES ? 2 * BUF_BYTES : STAGE_BYTES;
  static_assert(SLOTS <= 18, "at most two DMA instructions per tap");
  static_assert(SMEM <= 163840, "LDS budget");
};

// DIRECT (plain layers: no hi/lo pairs, no fused pooling): the weights are the MFMA's A operand (D = [channel][pixel], the same sums in the same order)
// and the epilogue runs from the accumulators (epilogue_direct_row: bias, residual, activation, rounding, v_permlane32_swap into 16-byte channel
// runs) -- no fp32 round trip through LDS, no barrier behind the K loop
// PHASE (N = 4 x 64, the phase convolutions of conv3x3(W, up2(x)) run at the resolution of x: db_model.hip phase_masks): 64-channel output tile ph = (dy, dx)
// has non-zero weights on taps (dy .. dy + 1) x (dx .. dx + 1) only -- a wave multiplies those four taps (runtime fragment bases, the same immediates),
// the weight rows of the other five are neither fetched (out-of-range DMA lanes) nor read
template <int NT, int NWV, bool SKEW, int BR = 0, bool DIRECT = false, int S = 1, bool PHASE = false>
__global__ __launch_bounds__(64 * NWV, 2) void conv3x3_pipe_kernel(ConvK p, const bf16_t* __restrict__ zero_page) {
  // (device pass only: the buffer-resource builtins do not exist for the host target, and a kernel template whose body fails to instantiate there
  // silently loses its launch stub -- "undefined symbol ... conv3x3_pipe_kernel" at dlopen)
#if defined(__HIP_DEVICE_COMPILE__)
  a16_kernel_enter();
  using C = PipeCfg<NT, NWV, BR, S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = lane & 31, qh = lane >> 5;
  const int wm = NT == 1 ? wave : (wave >> 1);   // which group of MT output rows
  const int wn = NT == 1 ? 0 : (wave & 1);       // which 64-channel half

  // the tile's blocks: (image, first output column); column Wo = no block (nothing fetched, nothing stored)
  int blk_b[C::NBLK], blk_x[C::NBLK];
  int nb, oy0 = 0;
  if constexpr (C::NBLK == 1) {
    int L = xcd_remap(blockIdx.x, gridDim.x);
    nb = L % p.n_tiles;                          // n_tiles counts NW-wide blocks here
    L /= p.n_tiles;
    const int txi = L % p.tiles_x;
    L /= p.tiles_x;
    const int tyi = L % p.tiles_y;
    blk_b[0] = L / p.tiles_y;
    blk_x[0] = txi * C::TW;
    oy0 = tyi * C::TH;
  } else {
    // (launch order, no XCD remap: with a block list the live tiles are the FRONT of the walk -- the remap would hand all of them to the first XCDs)
    const int L = blockIdx.x;
    nb = L % p.n_tiles;
    const int ti = L / p.n_tiles;
    if (p.blist) {
      const int cnt = p.blist[0], bpl = p.Wo >> 5, first = ti * C::NBLK;
      if (first >= cnt) return;
#pragma unroll
      for (int j = 0; j < C::NBLK; ++j) {
        const int e = first + j < cnt ? p.blist[1 + first + j] : -1;
        blk_b[j] = e >= 0 ? e / bpl : 0;
        blk_x[j] = e >= 0 ? (e - blk_b[j] * bpl) * 32 : p.Wo;
        if (e < 0) blk_b[j] = blk_b[0];
      }
    } else {
      const int txi = ti % p.tiles_x, bb = ti / p.tiles_x, x0 = txi * C::NBLK * 32;
      if (p.xlimit && x0 >= p.xlimit[bb]) return;      // ragged line: this tile is all padding response, filled by the caller
#pragma unroll
      for (int j = 0; j < C::NBLK; ++j) {
        const int x = x0 + 32 * j;
        blk_b[j] = bb;
        blk_x[j] = (x < p.Wo && (!p.xlimit || x < p.xlimit[bb])) ? x : p.Wo;
      }
    }
  }
  const int b = blk_b[0], ox0 = blk_x[0];
  const int in_cs = p.split ? 2 * p.Cin : p.Cin;
  const int nch32 = p.split ? 3 * (p.Cin >> 5) : (p.Cin >> 5);
  const int nslices = nch32 * 2;
  const bf16_t* in_b = p.in + (size_t)b * p.H * p.W * in_cs;      // (every block's image is at or behind the first block's: offsets stay non-negative)
  const bf16_t* wt = p.w + (size_t)(NT * nb) * nch32 * (9 * 64 * 32);

  // DMA slot j of this wave = instruction k = wave + NWV * j of the list [IN_INSTR input | W_INSTR weight]; lane -> 16-byte unit k * 64 + lane.
  // Input unit U: pixel U >> 1 of the patch, stored half U & 1 holds channel half (U & 1) ^ ((column >> 3) & 1).
  // Weight unit U: row = U >> 1 = tap * NW + n, channel half (U & 1) ^ ((row >> 3) & 1); source = the v1 tiling [N/64][Cin/32][9][64][32].
  // The requests are MUBUF (buffer_load_dwordx4 ... offen lds), not global_load_lds: a FLAT-encoded LDS load in flight makes hipcc's wait-count
  // pass treat every later LDS dependency as out of order (s_waitcnt lgkmcnt(0) in front of each tap: the prefetched fragments of the NEXT tap
  // were waited for too -- seen in the ISA of the first build of this kernel); a buffer load does not.  It also zero-fills by itself: a lane
  // whose byte offset is beyond num_records writes zeros (halo pixels outside the image; no zero page).
  constexpr int OOB = 0x7FFFF000;                 // == num_records of both descriptors
  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(in_b), 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wt), 0, OOB, 0x00020000);
  int voff[C::SLOTS];                             // byte offset of the lane's unit at slice 0
  bool is_in[C::SLOTS];
  int dst[C::SLOTS];
#pragma unroll
  for (int j = 0; j < C::SLOTS; ++j) {
    int k = wave + NWV * j;
    if (k >= C::T_INSTR) k = C::T_INSTR - 1;
    is_in[j] = k < C::IN_INSTR;                  // wave-uniform
    dst[j] = k * 1024;
    if (is_in[j]) {
      const int U = k * 64 + lane;
      const int pix = U >> 1;
      const int pr = pix / C::TWIN, sx = pix - pr * C::TWIN;      // patch row, SLOT inside the row
      const int q = (U & 1) ^ ((sx >> 3) & 1);
      const int ix = S == 1 ? sx : (sx < C::XEVEN ? 2 * sx : 2 * (sx - C::XEVEN) + 1);      // slot -> input column of the patch
      const int bk = pr / C::PR, iy = pr - bk * C::PR;      // block of the patch row, row inside the block's patch
      int bj = blk_b[0], xj = blk_x[0];
#pragma unroll
      for (int t = 1; t < C::NBLK; ++t)
        if (bk == t) { bj = blk_b[t]; xj = blk_x[t]; }
      const int gy = oy0 * S - 1 + iy, gx = xj * S - 1 + ix;
      const bool inside = U < C::IN_UNITS && xj < p.Wo && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      voff[j] = inside ? (int)(((((size_t)(bj - b) * p.H + gy) * p.W + gx) * in_cs + q * 8) * 2) : OOB;
    } else {
      const int U = (k - C::IN_INSTR) * 64 + lane;
      const int row = U >> 1;
      const int tap = row / C::NW, n = row - tap * C::NW;
      const int hq = (U & 1) ^ ((row >> 3) & 1);
      voff[j] = (int)((((size_t)(n >> 6) * nch32 * (9 * 64 * 32)) + (tap * 64 + (n & 63)) * 32 + hq * 8) * 2);
      if constexpr (PHASE) {      // the tile's phase (dy, dx) = its index among the four 64-channel tiles: taps outside its 2 x 2 window are not fetched
        const int ph = NT * nb + (n >> 6), dr = tap / 3 - (ph >> 1), dc = tap % 3 - (ph & 1);
        if ((unsigned)dr > 1u || (unsigned)dc > 1u) voff[j] = OOB;
      }
    }
  }
  // element offsets of slice c: input channels (split mode: [x_hi | x_lo] against w_hi, then x_hi again against w_lo), weight chunk + half
  auto in_off = [&](int slice) {
    int c0 = slice << 4;
    if (c0 >= in_cs) c0 -= in_cs;
    if (c0 >= in_cs) c0 -= in_cs;
    return c0;
  };
  auto w_off = [&](int slice) { return (slice >> 1) * (9 * 64 * 32) + (slice & 1) * 16; };
  auto issue_one = [&](int j, int oi, int ow, int buf) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(is_in[j] ? r_in : r_w, (__attribute__((address_space(3))) void*)(smem + buf * C::BUF_BYTES + dst[j]), 16,
                                             voff[j], (is_in[j] ? oi : ow) * 2, 0, 0);
  };

  f32x16 acc[C::MT][2];
#pragma unroll
  for (int m = 0; m < C::MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // fragment addresses: A(tap (r, s), row-tile m) = a_lane[s] + ((m + r) * TWIN) * 32; B(tap, half) = b_lane + (tap * NW + half * 32) * 32
  int a_lane[3];
#pragma unroll
  for (int s = 0; s < 3; ++s)
  {
    const int soff = S == 1 ? s : ((s & 1) * C::XEVEN + (s >> 1));      // slot offset of tap column s
    a_lane[s] = ((((C::MT * wm) / C::BR) * C::PR + ((C::MT * wm) % C::BR) * S) * C::TWIN + lx + soff) * 32 + ((qh ^ (((lx + soff) >> 3) & 1)) << 4);
  }
  const int b_lane = C::IN_BYTES + (wn * 64 + lx) * 32 + ((qh ^ ((lx >> 3) & 1)) << 4);

  // PHASE: this wave's 64-channel tile is phase (dy, dx); its tap t4 = (a, b) is tap (dy + a, dx + b) of the 3 x 3 kernel
  constexpr int NTAP = PHASE ? 4 : 9;
  int a_ph[2] = {0, 0}, b_ph = 0;
  if constexpr (PHASE) {
    const int ph = NT * nb + wn, dy = ph >> 1, dx = ph & 1;
    a_ph[0] = (dx ? a_lane[1] : a_lane[0]) + dy * C::TWIN * 32;
    a_ph[1] = (dx ? a_lane[2] : a_lane[1]) + dy * C::TWIN * 32;
    b_ph = b_lane + ((dy * 3 + dx) * C::NW) * 32;
  }
  bf16x8 fa[2][C::MT], fb[2][2];
  auto load_frags = [&](const char* sb, int tap, int slot) {
    if constexpr (PHASE) {
      const int a = tap >> 1, b = tap & 1;
      fb[slot][0] = *reinterpret_cast<const bf16x8*>(sb + b_ph + ((a * 3 + b) * C::NW) * 32);
      fb[slot][1] = *reinterpret_cast<const bf16x8*>(sb + b_ph + ((a * 3 + b) * C::NW + 32) * 32);
#pragma unroll
      for (int m = 0; m < C::MT; ++m)
        fa[slot][m] = *reinterpret_cast<const bf16x8*>(sb + a_ph[b] + ((S * m + a) * C::TWIN) * 32);
    } else {
      const int r = tap / 3, s = tap - 3 * r;
      fb[slot][0] = *reinterpret_cast<const bf16x8*>(sb + b_lane + (tap * C::NW) * 32);
      fb[slot][1] = *reinterpret_cast<const bf16x8*>(sb + b_lane + (tap * C::NW + 32) * 32);
#pragma unroll
      for (int m = 0; m < C::MT; ++m)
        fa[slot][m] = *reinterpret_cast<const bf16x8*>(sb + a_lane[s] + ((S * m + r) * C::TWIN) * 32);
    }
  };
  auto mma_tap = [&](int slot) {
#pragma unroll
    for (int m = 0; m < C::MT; ++m) {
      if constexpr (DIRECT) {
        acc[m][0] = mfma_32x32x16_a16(fb[slot][0], fa[slot][m], acc[m][0]);
        acc[m][1] = mfma_32x32x16_a16(fb[slot][1], fa[slot][m], acc[m][1]);
      } else {
        acc[m][0] = mfma_32x32x16_a16(fa[slot][m], fb[slot][0], acc[m][0]);
        acc[m][1] = mfma_32x32x16_a16(fa[slot][m], fb[slot][1], acc[m][1]);
      }
    }
  };

#pragma unroll
  for (int j = 0; j < C::SLOTS; ++j) issue_one(j, in_off(0), w_off(0), 0);
  // one slice: wait for its data, then nine taps with the next tap's fragments and (MORE) the next slice's DMA requests between the MFMA groups
  auto slice_body = [&](int c, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* sb = smem + (c & 1) * C::BUF_BYTES;
    const int oi = in_off(c + 1), ow = w_off(c + 1), nbuf = (c + 1) & 1;
    load_frags(sb, 0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) load_frags(sb, tap + 1, (tap + 1) & 1);
      mma_tap(tap & 1);
      if constexpr (MORE) {
#pragma unroll
        for (int j = tap; j < C::SLOTS; j += 9) issue_one(j, oi, ow, nbuf);
      }
    }
    // pin the order the source asks for (left alone, the scheduler sinks every ds_read to just above its first use and waits lgkmcnt(0) there):
    // 6 reads of tap 0, then per tap {6 reads of the next tap, 8 MFMAs, this tap's DMA requests}
    __builtin_amdgcn_sched_group_barrier(0x100, 2 + C::MT, 0);
    static_for<9>([&](auto tap_c) {
      constexpr int tap = decltype(tap_c)::value;
      constexpr int nd = MORE ? (C::SLOTS > tap ? 1 : 0) + (C::SLOTS > tap + 9 ? 1 : 0) : 0;
      tap_groups<tap < 8 ? 2 + C::MT : 0, 2 * C::MT, nd>();
    });
  };
  // SKEW: the hand-over barrier sits in front of tap 8 instead of behind it.  Tap 8's fragments are in registers by then (so every read of
  // slice c is complete: the barrier still frees its buffer), the next slice's own DMA pieces were requested during taps 0 .. 3, and the first
  // fragments of slice c + 1 are requested right behind the barrier -- barrier skew and that first LDS round trip run under tap 8's eight MFMAs
  // instead of an idle pipe.  Fragment slot of (slice parity P, tap t) = (P + t) & 1, so two slices make one loop iteration.
  auto skew_slice = [&](int c, auto par_tag, auto more_tag) {
    constexpr int P = decltype(par_tag)::value;
    constexpr bool MORE = decltype(more_tag)::value;
    constexpr int DPT = (C::SLOTS + NTAP - 2) / (NTAP - 1) > 2 ? (C::SLOTS + NTAP - 2) / (NTAP - 1) : 2;      // DMA requests per tap, over the first taps
    const char* sb = smem + P * C::BUF_BYTES;
    const int oi = in_off(c + 1), ow = w_off(c + 1);
    static_for<NTAP - 1>([&](auto tap_c) {
      constexpr int tap = decltype(tap_c)::value;
      load_frags(sb, tap + 1, (P * NTAP + tap + 1) & 1);
      mma_tap((P * NTAP + tap) & 1);
      if constexpr (MORE) {
#pragma unroll
        for (int j = DPT * tap; j < DPT * tap + DPT && j < C::SLOTS; ++j) issue_one(j, oi, ow, P ^ 1);
        static_assert(C::SLOTS <= DPT * (NTAP - 1), "every DMA request has a tap");
      }
    });
    static_for<NTAP - 1>([&](auto tap_c) {
      constexpr int tap = decltype(tap_c)::value;
      constexpr int left = C::SLOTS - DPT * tap;
      constexpr int nd = MORE ? (left > DPT ? DPT : (left > 0 ? left : 0)) : 0;
      tap_groups<2 + C::MT, 2 * C::MT, nd>();
    });
    if constexpr (MORE) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      load_frags(smem + (P ^ 1) * C::BUF_BYTES, 0, ((P + 1) * NTAP) & 1);
    }
    mma_tap((P * NTAP + NTAP - 1) & 1);
    tap_groups<MORE ? 2 + C::MT : 0, 2 * C::MT, 0>();
  };
  if constexpr (SKEW) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_frags(smem, 0, 0);
    for (int c = 0; c + 2 < nslices; c += 2) {
      skew_slice(c, std::integral_constant<int, 0>{}, std::true_type{});
      skew_slice(c + 1, std::integral_constant<int, 1>{}, std::true_type{});
    }
    skew_slice(nslices - 2, std::integral_constant<int, 0>{}, std::true_type{});
    skew_slice(nslices - 1, std::integral_constant<int, 1>{}, std::false_type{});
  } else {
    static_assert(!PHASE, "phase convolutions: the SKEW loop only");
    for (int c = 0; c + 1 < nslices; ++c) slice_body(c, std::true_type{});
    slice_body(nslices - 1, std::false_type{});
  }

  if constexpr (DIRECT) {
    const int n0 = (NT == 1 ? nb : nb * 2 + wn) * 64;
    const DirectBias bs = direct_bias<2>(p, n0, qh);
    // the wave's four rows lie in block (MT wm) / BR of the tile, from row (MT wm) % BR of that block's map
    int bj = blk_b[0], xj = blk_x[0];
#pragma unroll
    for (int t = 1; t < C::NBLK; ++t)
      if ((C::MT * wm) / C::BR == t) { bj = blk_b[t]; xj = blk_x[t]; }
    if (xj >= p.Wo) return;      // (wave-uniform) no block
#pragma unroll
    for (int m = 0; m < C::MT; ++m) epilogue_direct_row<2>(p, acc[m], bs, bj, oy0 + (C::MT * wm) % C::BR + m, xj, lx, n0, qh, nullptr);
    return;
  }
  // epilogue in two passes of PASS_ROWS x 32 pixels x 64 channels (fp32 in LDS)
  float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    const bool mine = NT == 1 ? ((wave / (NWV / 2)) == pass) : (wn == pass);
    if (mine) {
      const int rbase = NT == 1 ? C::MT * (wave % (NWV / 2)) : C::MT * wm;   // local patch row inside this pass' rows
#pragma unroll
      for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int tx = (r & 3) + 8 * (r >> 2) + 4 * qh;
            stage[((rbase + m) * 32 + tx) * 64 + n * 32 + lx] = acc[m][n][r];
          }
    }
    __syncthreads();
    if constexpr (C::NBLK > 1) {
      static_assert(NT == 2 || C::NBLK == 1, "blocked tiles: one pass = all rows of one 64-channel half");
#pragma unroll 1
      for (int j = 0; j < C::NBLK; ++j) {      // (uniform: a dead block stores nothing)
        int bj = blk_b[0], xj = blk_x[0];
#pragma unroll
        for (int t = 1; t < C::NBLK; ++t)
          if (j == t) { bj = blk_b[t]; xj = blk_x[t]; }
        if (xj < p.Wo) epilogue_store<C::BR, 32, C::NTHR, 2>(p, stage + j * C::BR * 32 * 64, tid, bj, 0, xj, (nb * 2 + pass) * 64);
      }
    } else if (NT == 1) {
      epilogue_store<C::PASS_ROWS, 32, C::NTHR, 0>(p, stage, tid, b, oy0 + C::PASS_ROWS * pass, ox0, nb * 64);
    } else {
      epilogue_store<C::PASS_ROWS, 32, C::NTHR, 0>(p, stage, tid, b, oy0, ox0, (nb * 2 + pass) * 64);
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// conv3x3_pipe_kernel<NT, NWV, SKEW, 0, DIRECT, S> as a PERSISTENT kernel ("v5"): a workgroup walks a contiguous run of tiles and the slice pipeline
// runs across the tile boundary.  One tile per workgroup pays, per tile, the launch of the workgroup, the address arithmetic of its DMA requests, a DMA
// round trip with nothing to multiply, and the epilogue with nothing in flight: T(tile) = 3.48 us x slices + 9.1 us on the 8-wave tile (r06) -- 14 % of
// a 256-channel layer, 25 % of a 128-channel one.  Here the first slice of tile i + 1 is requested during the LAST slice of tile i (into the buffer
// that slice does not use: the slice count is even), the hand-over barrier in front of tile i's last tap is also tile i + 1's first one, its first
// fragments are read behind it, and the epilogue of tile i (registers -> global, nothing staged) runs while that slice is already in LDS.
// Per tile the per-lane DMA offsets are eight integer operations per slot from the slot's (row, column, half) in the patch, kept packed in a register.
// Same tiles, same K order, same epilogue: every output bit equals conv3x3_pipe_kernel's (tests/test_gpu_det.py).
// ---------------------------------------------------------------------------------------------------
template <int NT, int NWV, int S>
__global__ __launch_bounds__(64 * NWV, 2) void conv3x3_pipe_persist_kernel(ConvK p) {
#if defined(__HIP_DEVICE_COMPILE__)      // (buffer-resource builtins: device pass only, see conv3x3_pipe_kernel)
  a16_kernel_enter();
  using C = PipeCfg<NT, NWV, 0, S>;
  static_assert(C::NBLK == 1, "ordinary tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = lane & 31, qh = lane >> 5;
  const int wm = NT == 1 ? wave : (wave >> 1);   // which group of MT output rows
  const int wn = NT == 1 ? 0 : (wave & 1);       // which 64-channel half
  const int G = gridDim.x;
  const int Lw = xcd_remap(blockIdx.x, G);
  const int t0 = (int)((long long)Lw * p.total_tiles / G), t1 = (int)((long long)(Lw + 1) * p.total_tiles / G);
  if (t0 >= t1) return;
  const int in_cs = p.Cin;
  const int nch32 = p.Cin >> 5, nslices = nch32 * 2;

  // DMA slot j of this wave = instruction k = wave + NWV * j of the list [IN_INSTR input | W_INSTR weight] (conv3x3_pipe_kernel's list and LDS layout).
  // Tile-independent part of a slot: an input unit's (patch row, input column of the patch, stored half), packed; a weight unit's byte offset.
  constexpr int OOB = 0x7FFFF000;
  int meta[C::SLOTS];
  bool is_in[C::SLOTS];
  int dst[C::SLOTS];
#pragma unroll
  for (int j = 0; j < C::SLOTS; ++j) {
    int k = wave + NWV * j;
    if (k >= C::T_INSTR) k = C::T_INSTR - 1;
    is_in[j] = k < C::IN_INSTR;                  // wave-uniform
    dst[j] = k * 1024;
    if (is_in[j]) {
      const int U = k * 64 + lane;
      const int pix = U >> 1;
      const int pr = pix / C::TWIN, sx = pix - pr * C::TWIN;
      const int q = (U & 1) ^ ((sx >> 3) & 1);
      const int ix = S == 1 ? sx : (sx < C::XEVEN ? 2 * sx : 2 * (sx - C::XEVEN) + 1);
      meta[j] = U < C::IN_UNITS ? (pr | (ix << 8) | (q << 16)) : -1;
    } else {
      const int U = (k - C::IN_INSTR) * 64 + lane;
      const int row = U >> 1;
      const int tap = row / C::NW, n = row - tap * C::NW;
      const int hq = (U & 1) ^ ((row >> 3) & 1);
      meta[j] = (int)((((size_t)(n >> 6) * nch32 * (9 * 64 * 32)) + (tap * 64 + (n & 63)) * 32 + hq * 8) * 2);
    }
  }
  struct Tile { int b, oy0, ox0, nb; };
  auto decode = [&](int L) {
    Tile t;
    t.nb = L % p.n_tiles;
    L /= p.n_tiles;
    const int txi = L % p.tiles_x;
    L /= p.tiles_x;
    const int tyi = L % p.tiles_y;
    t.b = L / p.tiles_y;
    t.ox0 = txi * C::TW;
    t.oy0 = tyi * C::TH;
    return t;
  };
  auto offsets = [&](const Tile& t, int (&vo)[C::SLOTS]) {
#pragma unroll
    for (int j = 0; j < C::SLOTS; ++j) {
      if (is_in[j]) {
        const int m = meta[j];
        const int gy = t.oy0 * S - 1 + (m & 0xFF), gx = t.ox0 * S - 1 + ((m >> 8) & 0xFF);
        const bool inside = m >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        vo[j] = inside ? ((gy * p.W + gx) * in_cs + ((m >> 16) & 1) * 8) * 2 : OOB;
      } else {
        vo[j] = meta[j];
      }
    }
  };
  auto rsrc_in = [&](const Tile& t) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.in + (size_t)t.b * p.H * p.W * in_cs), 0, OOB, 0x00020000); };
  auto rsrc_w = [&](const Tile& t) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w + (size_t)(NT * t.nb) * nch32 * (9 * 64 * 32)), 0, OOB, 0x00020000); };
  auto in_off = [&](int slice) { return slice << 4; };
  auto w_off = [&](int slice) { return (slice >> 1) * (9 * 64 * 32) + (slice & 1) * 16; };

  f32x16 acc[C::MT][2];
#pragma unroll
  for (int m = 0; m < C::MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  int a_lane[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int soff = S == 1 ? s : ((s & 1) * C::XEVEN + (s >> 1));      // slot offset of tap column s
    a_lane[s] = ((C::MT * wm * S) * C::TWIN + lx + soff) * 32 + ((qh ^ (((lx + soff) >> 3) & 1)) << 4);
  }
  const int b_lane = C::IN_BYTES + (wn * 64 + lx) * 32 + ((qh ^ ((lx >> 3) & 1)) << 4);
  bf16x8 fa[2][C::MT], fb[2][2];
  auto load_frags = [&](const char* sb, int tap, int slot) {
    const int r = tap / 3, s = tap - 3 * r;
    fb[slot][0] = *reinterpret_cast<const bf16x8*>(sb + b_lane + (tap * C::NW) * 32);
    fb[slot][1] = *reinterpret_cast<const bf16x8*>(sb + b_lane + (tap * C::NW + 32) * 32);
#pragma unroll
    for (int m = 0; m < C::MT; ++m) fa[slot][m] = *reinterpret_cast<const bf16x8*>(sb + a_lane[s] + ((S * m + r) * C::TWIN) * 32);
  };
  auto mma_tap = [&](int slot) {
#pragma unroll
    for (int m = 0; m < C::MT; ++m) {
      acc[m][0] = mfma_32x32x16_a16(fb[slot][0], fa[slot][m], acc[m][0]);
      acc[m][1] = mfma_32x32x16_a16(fb[slot][1], fa[slot][m], acc[m][1]);
    }
  };

  Tile cur = decode(t0);
  int vo[C::SLOTS];
  offsets(cur, vo);
  __amdgpu_buffer_rsrc_t r_in = rsrc_in(cur), r_w = rsrc_w(cur);
#pragma unroll
  for (int j = 0; j < C::SLOTS; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(is_in[j] ? r_in : r_w, (__attribute__((address_space(3))) void*)(smem + dst[j]), 16, vo[j], 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  load_frags(smem, 0, 0);

  for (int ti = t0; ti < t1; ++ti) {
    // one slice of the K loop (conv3x3_pipe_kernel's skew_slice): the DMA requests of the slice after it go out behind the first taps' MFMAs.
    // LAST (the tile's last slice): the requests are slice 0 of the NEXT tile (of the last tile: all out of range -- the buffer is zero-filled and
    // never read), and the first fragments of that slice are read after the epilogue, not before it (they would be 24 live registers across it)
    auto skew_slice = [&](int c, auto par_tag, auto last_tag, int oi, int ow) {
      constexpr int P = decltype(par_tag)::value;
      constexpr bool LAST = decltype(last_tag)::value;
      constexpr int NTAP = 9;
      constexpr int DPT = (C::SLOTS + NTAP - 2) / (NTAP - 1) > 2 ? (C::SLOTS + NTAP - 2) / (NTAP - 1) : 2;
      const char* sb = smem + P * C::BUF_BYTES;
      static_for<NTAP - 1>([&](auto tap_c) {
        constexpr int tap = decltype(tap_c)::value;
        load_frags(sb, tap + 1, (P * NTAP + tap + 1) & 1);
        mma_tap((P * NTAP + tap) & 1);
#pragma unroll
        for (int j = DPT * tap; j < DPT * tap + DPT && j < C::SLOTS; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(is_in[j] ? r_in : r_w, (__attribute__((address_space(3))) void*)(smem + (P ^ 1) * C::BUF_BYTES + dst[j]), 16, vo[j],
                                                   (is_in[j] ? oi : ow) * 2, 0, 0);
        static_assert(C::SLOTS <= DPT * (NTAP - 1), "every DMA request has a tap");
      });
      static_for<NTAP - 1>([&](auto tap_c) {
        constexpr int tap = decltype(tap_c)::value;
        constexpr int left = C::SLOTS - DPT * tap;
        constexpr int nd = left > DPT ? DPT : (left > 0 ? left : 0);
        tap_groups<2 + C::MT, 2 * C::MT, nd>();
      });
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if constexpr (!LAST) load_frags(smem + (P ^ 1) * C::BUF_BYTES, 0, ((P + 1) * NTAP) & 1);
      mma_tap((P * NTAP + NTAP - 1) & 1);
      tap_groups<LAST ? 0 : 2 + C::MT, 2 * C::MT, 0>();
    };
    for (int c = 0; c + 2 < nslices; c += 2) {
      skew_slice(c, std::integral_constant<int, 0>{}, std::false_type{}, in_off(c + 1), w_off(c + 1));
      skew_slice(c + 1, std::integral_constant<int, 1>{}, std::false_type{}, in_off(c + 2), w_off(c + 2));
    }
    skew_slice(nslices - 2, std::integral_constant<int, 0>{}, std::false_type{}, in_off(nslices - 1), w_off(nslices - 1));
    // this tile's requests are all out: the offsets and descriptors become the next tile's
    const Tile done = cur;
    {
      const bool has_next = ti + 1 < t1;
      cur = decode(has_next ? ti + 1 : ti);
      offsets(cur, vo);
      if (!has_next) {
#pragma unroll
        for (int j = 0; j < C::SLOTS; ++j) vo[j] = OOB;
      }
      r_in = rsrc_in(cur);
      r_w = rsrc_w(cur);
    }
    skew_slice(nslices - 1, std::integral_constant<int, 1>{}, std::true_type{}, 0, 0);
    // epilogue from the accumulators; the next tile's first slice is in LDS
    {
      const int n0 = (NT == 1 ? done.nb : done.nb * 2 + wn) * 64;
      const DirectBias bs = direct_bias<2>(p, n0, qh);
#pragma unroll
      for (int m = 0; m < C::MT; ++m) {
        epilogue_direct_row<2>(p, acc[m], bs, done.b, done.oy0 + C::MT * wm + m, done.ox0, lx, n0, qh, nullptr);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
      }
    }
    load_frags(smem, 0, 0);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// 1x1 stride-1 layers with K >= 256 as a pipelined GEMM: out [M][N] = A [M][K] . W^T (+ bias, residual, activation), single-pass modes.
// These layers -- the CRNN's conv4 (K = 1024) and sequence-head projections (512 -> 2048, 256 -> 2048, 512 -> 512 / 256), Lore's wide 1x1 convs --
// ran on conv_igemm_kernel<1, 1> / conv1x1_wide_kernel (4 x 32 x 64 tiles, two barriers per 32 or 128 channels) or on the streaming row GEMM
// (gemm_argmax_kernel<., 1>: a wave's 32 rows x the whole W through LDS, one ds_read per MFMA, 64-byte store segments) at 0.14-0.24 of the
// matrix peak and 0.27-0.41 of HBM.  Here: conv3x3_pipe_kernel's structure on a GEMM tile --
//   * 512 rows (16 groups of 32 consecutive rows: an MFMA row-tile each) x 128 output channels per 8-wave workgroup, a wave 4 row-tiles x 64 channels;
//   * K in 64-channel slices, two LDS buffers of (64 KB of A + 16 KB of W), requested by MUBUF LDS-DMA (ten instructions per wave and slice) between
//     the MFMA groups of the slice before; rows are 128 bytes with the 16-byte slots XOR-swizzled by ((row >> 1) & 7) on the DMA source and on
//     the read address (conflict-free ds_read_b128 for any 16 rows of a fragment);
//   * four k-steps per slice, the next k-step's six fragments requested between this one's eight MFMAs, the hand-over barrier in front of the
//     last k-step (SKEW);
//   * the weights are the MFMA's A operand: bias / residual / activation / rounding from the accumulators, 16-byte channel runs.
// LIST (ragged sequence views: ConvDesc.xlimit_rows + block_list): the tile's row groups come from the compacted list of live 32-step groups.
// ---------------------------------------------------------------------------------------------------
struct GemmPipeCfg {
  static constexpr int NWV = 8, NTHR = 512, MT = 4, NG = 16;       // row groups (MFMA row-tiles) per workgroup
  static constexpr int KS = 64, NW = 128;
  static constexpr int A_BYTES = NG * 32 * KS * 2, W_BYTES = NW * KS * 2, BUF_BYTES = A_BYTES + W_BYTES;      // 65 536 + 16 384
  static constexpr int A_INSTR = A_BYTES / 1024, W_INSTR = W_BYTES / 1024, SLOTS = (A_INSTR + W_INSTR) / NWV;  // 64 + 16 -> 10 per wave
  static constexpr int SMEM = 2 * BUF_BYTES;                       // 163 840: the whole LDS of a CU
  static_assert(A_INSTR % NWV == 0 && W_INSTR % NWV == 0, "a slot is an A or a W request for every wave alike");
};

template <bool LIST>
__global__ __launch_bounds__(512, 2) void gemm_pipe_kernel(ConvK p) {
#if defined(__HIP_DEVICE_COMPILE__)      // (buffer-resource builtins: device pass only, see conv3x3_pipe_kernel)
  a16_kernel_enter();
  using C = GemmPipeCfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = lane & 31, qh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;       // which four row groups, which 64-channel half

  // tile -> (128-channel block, 16 row groups); consecutive workgroups = the channel blocks of one row tile (its A slices come from that XCD's L2)
  const int L = LIST ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  const int nb = L % p.n_tiles, ti = L / p.n_tiles;
  const long long M = (long long)p.B * p.H * p.W;
  const int n_groups = (int)(M >> 5);
  int cnt = n_groups;
  if (LIST) {
    cnt = p.blist[0];
    if (ti * C::NG >= cnt) return;
  }
  // row group of the tile's j-th MFMA row-tile (-1: none)
  auto group_of = [&](int j) {
    const int i = ti * C::NG + j;
    if (i >= cnt) return -1;
    return LIST ? p.blist[1 + i] : i;
  };
  const int K = p.Cin, nslices = K >> 6;
  const bf16_t* wt = p.w + (size_t)(2 * nb) * (K >> 5) * 2048;
  constexpr int OOB = 0x7FFFF000;
  // A rows can lie anywhere in a > 2 GB tensor: the descriptor starts at the tile's first group, offsets are relative to it (a list is ascending)
  const int g0 = group_of(0);
  const __amdgpu_buffer_rsrc_t r_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.in + (size_t)g0 * 32 * K), 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wt), 0, OOB, 0x00020000);
  int voff[C::SLOTS];
#pragma unroll
  for (int j = 0; j < C::SLOTS; ++j) {
    const int k = wave + C::NWV * j;
    if (j < C::A_INSTR / C::NWV) {
      // A unit U = k * 64 + lane: row r = U >> 3 of the tile, stored slot U & 7 holds channel run (U & 7) ^ ((r >> 1) & 7) of the slice
      const int U = k * 64 + lane, r = U >> 3, c = (U & 7) ^ ((r >> 1) & 7);
      const int g = group_of(r >> 5);            // (r >> 5 is uniform over an instruction: 8 rows of one group)
      const long long off = ((long long)(g - g0) * 32 + (r & 31)) * K + c * 8;
      voff[j] = (g >= 0 && off * 2 < OOB) ? (int)(off * 2) : OOB;
    } else {
      // W unit: row n = U >> 3 of the 128, slot U & 7 holds channel run (U & 7) ^ ((n >> 1) & 7); source = the 1x1 tiling [N/64][K/32][64][32]
      const int U = (k - C::A_INSTR) * 64 + lane, n = U >> 3, c = (U & 7) ^ ((n >> 1) & 7);
      voff[j] = (int)((((size_t)(n >> 6) * (K >> 5) + (c >> 2)) * 2048 + (n & 63) * 32 + (c & 3) * 8) * 2);
    }
  }
  auto issue_one = [&](int j, int slice, int buf) {
    const bool is_a = j < C::A_INSTR / C::NWV;      // compile-time per slot
    __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? r_a : r_w, (__attribute__((address_space(3))) void*)(smem + buf * C::BUF_BYTES + (wave + C::NWV * j) * 1024), 16,
                                             voff[j], is_a ? slice * 128 : slice * 2 * 2048 * 2, 0, 0);
  };

  f32x16 acc[C::MT][2];
#pragma unroll
  for (int m = 0; m < C::MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // fragment addresses of k-step kk: A(row-tile m) = a_lane[kk] + (4 wm + m) * 4096; W(half h) = b_lane[kk] + h * 4096
  int a_lane[4], b_lane[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int slot = ((kk * 2 + qh) ^ ((lx >> 1) & 7)) << 4;
    a_lane[kk] = (C::MT * wm) * 4096 + lx * 128 + slot;
    b_lane[kk] = C::A_BYTES + (wn * 64 + lx) * 128 + slot;
  }
  bf16x8 fa[2][C::MT], fb[2][2];
  auto load_frags = [&](const char* sb, int kk, int slot) {
    fb[slot][0] = *reinterpret_cast<const bf16x8*>(sb + b_lane[kk]);
    fb[slot][1] = *reinterpret_cast<const bf16x8*>(sb + b_lane[kk] + 4096);
#pragma unroll
    for (int m = 0; m < C::MT; ++m) fa[slot][m] = *reinterpret_cast<const bf16x8*>(sb + a_lane[kk] + m * 4096);
  };
  auto mma_step = [&](int slot) {
#pragma unroll
    for (int m = 0; m < C::MT; ++m) {
      acc[m][0] = mfma_32x32x16_a16(fb[slot][0], fa[slot][m], acc[m][0]);      // D = [channel][row]
      acc[m][1] = mfma_32x32x16_a16(fb[slot][1], fa[slot][m], acc[m][1]);
    }
  };

#pragma unroll
  for (int j = 0; j < C::SLOTS; ++j) issue_one(j, 0, 0);
  // one slice: three k-steps with the next k-step's fragments and (MORE) the next slice's DMA requests between the MFMAs, the hand-over barrier,
  // the next slice's first fragments, the fourth k-step.  Four k-steps per slice: the fragment slot of k-step kk is kk & 1 in every slice.
  auto slice = [&](int c, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    const char* sb = smem + (c & 1) * C::BUF_BYTES;
    static_for<3>([&](auto kk_c) {
      constexpr int kk = decltype(kk_c)::value;
      load_frags(sb, kk + 1, (kk + 1) & 1);
      mma_step(kk & 1);
      if constexpr (MORE) {
#pragma unroll
        for (int j = 4 * kk; j < 4 * kk + 4 && j < C::SLOTS; ++j) issue_one(j, c + 1, (c + 1) & 1);
      }
    });
    static_for<3>([&](auto kk_c) {
      constexpr int kk = decltype(kk_c)::value;
      constexpr int nd = MORE ? (C::SLOTS - 4 * kk > 4 ? 4 : (C::SLOTS - 4 * kk > 0 ? C::SLOTS - 4 * kk : 0)) : 0;
      tap_groups<2 + C::MT, 2 * C::MT, nd>();
    });
    if constexpr (MORE) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      load_frags(smem + ((c + 1) & 1) * C::BUF_BYTES, 0, 0);
    }
    mma_step(1);
    tap_groups<MORE ? 2 + C::MT : 0, 2 * C::MT, 0>();
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  load_frags(smem, 0, 0);
  for (int c = 0; c + 1 < nslices; ++c) slice(c, std::true_type{});
  slice(nslices - 1, std::false_type{});

  // epilogue from the accumulators: lane (lx, q) owns row lx of its row-tile; the map is addressed as [1][M / 32][32] (oy = row group, ox = lx)
  const int n0 = (nb * 2 + wn) * 64;
  const DirectBias bs = direct_bias<2>(p, n0, qh);
  ConvK q = p;
  q.B = 1; q.Ho = n_groups; q.Wo = 32;
  // bf16 outputs leave as whole 128-byte lines through a wave-private 4 KB LDS tile (epilogue_direct_row's xp path): the operand buffers are dead once
  // every wave has left the K loop.  Straight from the accumulator layout a store instruction touches 32 rows with 32 bytes each -- the 2048-wide
  // projections wrote at 2.0 TB/s that way
  if (p.argmax_part) {
    // fused arg-max over classes (CTC greedy decode, modeling_ocr_recognition.py:168-171): best (value, index) of every row's 64-class slice from the
    // accumulators -- lane (lx, q) holds 32 of the 64 logits of row lx: register r of block nb = class n0 + nb * 32 + (r & 3) + 8 (r >> 2) + 4 q --,
    // then the two lanes of a row; ties keep the LOWEST class index, like torch.argmax.  Partial results [row][N / 64] float2, as epilogue_store writes them
    const int nt64 = p.N >> 6;
#pragma unroll
    for (int m = 0; m < C::MT; ++m) {
      const int g = group_of(C::MT * wm + m);
      if (g < 0) continue;
      float bv = -INFINITY;
      int bi = 0x7FFFFFFF;
#pragma unroll
      for (int nbk = 0; nbk < 2; ++nbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float bias = r < 4 ? bs.v[nbk][0][r & 3] : r < 8 ? bs.v[nbk][1][r & 3] : r < 12 ? bs.v[nbk][2][r & 3] : bs.v[nbk][3][r & 3];
          const float v = acc[m][nbk][r] + bias;
          const int ci = n0 + nbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * qh;
          if (v > bv || (v == bv && ci < bi)) { bv = v; bi = ci; }
        }
      const float ov = __shfl_xor(bv, 32);
      const int oi = __shfl_xor(bi, 32);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      if (qh == 0) {
        float2 pr;
        pr.x = bv;
        pr.y = __int_as_float(bi);
        reinterpret_cast<float2*>(p.argmax_part)[((size_t)g * 32 + lx) * nt64 + (n0 >> 6)] = pr;
      }
    }
    return;
  }
  char* xp = nullptr;
  if (!p.out_f32 && p.xp_store) {
    __syncthreads();
    xp = smem + wave * 4096;
  }
#pragma unroll
  for (int m = 0; m < C::MT; ++m) {
    const int g = group_of(C::MT * wm + m);
    if (g >= 0) epilogue_direct_row<2>(q, acc[m], bs, 0, g, 0, lx, n0, qh, xp);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// CRNN conv0 (3x3, 1 -> 64, + BN + ReLU) + MaxPool2d(2) + conv1 (3x3, 64 -> 128, + BN + ReLU) + MaxPool2d(2) in ONE kernel (crnn/modeling_crnn.py:44-55),
// single-pass modes.  As two launches the 64-channel 16 x 320 map of every text line went to HBM and came back; conv1 (pool in a staged epilogue) sat on the
// register-staged kernel at 0.23 of the matrix peak and conv0 + pool was a launch of its own.  The map is cheap to RE-COMPUTE where it is needed: a 16 x 32
// conv1 tile reads an 18 x 34 patch of it = 77 MFMA row tiles of conv0 (K = 9 taps padded to 16) beside the tile's 2 304 MFMAs.  A workgroup
//   1. stages a 38 x 72 gray patch in LDS and computes conv0 + bias + ReLU + 2x2 max + rounding for the patch (crnn_conv0_pool_mfma_kernel's operands: the
//      32 rows of an M tile are 8 pooling windows x 4 pixels, the max is in-lane), writing the result -- zero outside the 16 x 320 map, which is conv1's
//      padding -- straight into the four 16-channel slice images of conv3x3_pipe_kernel's layout, resident for the whole K loop;
//   2. runs conv1's K loop on those images with the weights streamed through a two-buffer ring by MUBUF LDS-DMA (software-pipelined taps, hand-over
//      barrier in front of the last tap), 16 x 32 pixels x 128 channels per 8-wave workgroup, weights as the MFMA's A operand;
//   3. pools in registers (the lane's two rows, the neighbouring lane's column by DPP) and stores 16-byte runs -- no LDS stage.
// K order = conv3x3_pipe_kernel's (16-channel slices outer, taps inner): every bit equals conv0+pool -> the v4 conv -> MaxPool2d(2) as three launches
// (PT_CONV01=0 PT_POOL_FUSED=0; tests/test_gpu_rec.py).  Measured on the recognition stage alone (2 227 lines): 3.86 ms (0.93 + 2.93) -> 2.24 ms, of which
// 1.0 ms is the K loop, 0.85 ms the conv0 phase (VALU-bound: ~100 instructions per row tile, two waves per SIMD) and 0.2 ms the epilogue.
// ConvK: in = gray [n][32][640] (16 bit), head_w = conv0 weights fp32 [64][9], head_b = conv0 bias fp32 [64]; w / bias / out / xlimit = conv1's.
// ---------------------------------------------------------------------------------------------------
struct Conv01Cfg {
  static constexpr int NWV = 8, NTHR = 512, MT = 4, TH = 16, TW = 32, NW = 128;
  static constexpr int THIN = TH + 2, TWIN = TW + 2, NPIX = THIN * TWIN;           // 18 x 34 pooled-conv0 pixels
  static constexpr int NMT = (NPIX + 7) / 8;                                       // conv0 M tiles (8 pooling windows each)
  static constexpr int IMG_BYTES = NPIX * 32 + 64;                                 // one 16-channel slice image; + 64: the two slices a conv0 store touches lie in different banks
  static constexpr int W_INSTR = 9 * NW * 2 / 64, W_BYTES = W_INSTR * 1024;        // 36 KB of weights per slice
  static constexpr int SLOTS = (W_INSTR + NWV - 1) / NWV;                          // 5 DMA requests per wave and slice
  static constexpr int GH = 2 * THIN + 2, GW = 72, GQ = GW / 4;                    // gray patch: 38 rows x 72 columns from column 64 tile - 4 (8-byte aligned quads)
  static constexpr int OFF_W = ((4 * IMG_BYTES + 1023) / 1024) * 1024, OFF_G = OFF_W + 2 * W_BYTES;
  static constexpr int SMEM = OFF_G + GH * GW * 2;
  static_assert(SMEM <= 163840, "LDS budget");
};

__device__ __forceinline__ float dpp_xor1(float v) {      // the value of lane ^ 1 (quad_perm [1, 0, 3, 2])
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}

__global__ __launch_bounds__(512, 2) void crnn_conv01_kernel(ConvK p) {
#if defined(__HIP_DEVICE_COMPILE__)
  a16_kernel_enter();
  using C = Conv01Cfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = lane & 31, qh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int txi = blockIdx.x % p.tiles_x, b = blockIdx.x / p.tiles_x;
  const int ox0 = txi * C::TW;
  if (p.xlimit && ox0 >= p.xlimit[b]) return;      // ragged line: this tile is all padding response, filled by the caller
  const int GHI = 2 * p.H, GWI = 2 * p.W;            // gray image (32 x 640): the conv0 map is p.H x p.W = 16 x 320 after its pool

  // ---- weights: MUBUF LDS-DMA ring (conv3x3_pipe_kernel's weight half)
  constexpr int OOB = 0x7FFFF000;
  const int nch32 = p.Cin >> 5, nslices = nch32 * 2;      // 64 input channels: 4 slices
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, OOB, 0x00020000);
  int voff[C::SLOTS], dst[C::SLOTS];
#pragma unroll
  for (int j = 0; j < C::SLOTS; ++j) {
    int k = wave + C::NWV * j;
    if (k >= C::W_INSTR) k = C::W_INSTR - 1;
    dst[j] = k * 1024;
    const int U = k * 64 + lane, row = U >> 1;
    const int tap = row / C::NW, n = row - tap * C::NW;
    const int hq = (U & 1) ^ ((row >> 3) & 1);
    voff[j] = (int)((((size_t)(n >> 6) * nch32 * (9 * 64 * 32)) + (tap * 64 + (n & 63)) * 32 + hq * 8) * 2);
  }
  auto w_off = [&](int slice) { return (slice >> 1) * (9 * 64 * 32) + (slice & 1) * 16; };
  auto issue_one = [&](int j, int slice, int buf) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)(smem + C::OFF_W + buf * C::W_BYTES + dst[j]), 16, voff[j],
                                             w_off(slice) * 2, 0, 0);
  };
#pragma unroll
  for (int j = 0; j < C::SLOTS; ++j) issue_one(j, 0, 0);      // slice 0's weights fly while conv0 is computed

  // ---- conv0 + pool into the slice images
  {
    // gray patch: rows -3 .. 34, columns 64 tile - 4 .. + 67 of the line's image as aligned quads of pixels (zeros outside the image: conv0's padding)
    const bf16_t* gin = p.in + (size_t)b * GHI * GWI;
#pragma unroll
    for (int it = 0; it < (C::GH * C::GQ + C::NTHR - 1) / C::NTHR; ++it) {
      const int i = tid + it * C::NTHR;
      if (i < C::GH * C::GQ) {
        const int yy = -3 + i / C::GQ, xx = 2 * ox0 - 4 + 4 * (i % C::GQ);
        u32x2 v = {0u, 0u};
        if ((unsigned)yy < (unsigned)GHI && (unsigned)xx < (unsigned)GWI) v = *reinterpret_cast<const u32x2*>(gin + (size_t)yy * GWI + xx);
        *reinterpret_cast<u32x2*>(smem + C::OFF_G + i * 8) = v;
      }
    }
    const float* w64x9 = reinterpret_cast<const float*>(p.head_w);
    bf16x8 bw[2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      uint32_t pk[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k0 = qh * 8 + 2 * i, k1 = k0 + 1;
        const uint32_t a0 = k0 < 9 ? f32_to_bf16(w64x9[(nh * 32 + lx) * 9 + k0]) : 0u;
        const uint32_t a1 = k1 < 9 ? f32_to_bf16(w64x9[(nh * 32 + lx) * 9 + k1]) : 0u;
        pk[i] = a0 | (a1 << 16);
      }
      const u32x4 v = {pk[0], pk[1], pk[2], pk[3]};
      bw[nh] = __builtin_bit_cast(bf16x8, v);
    }
    const float bs0[2] = {p.head_b[lx], p.head_b[32 + lx]};
    __syncthreads();
    // M tile mt = pooled pixels 8 mt .. 8 mt + 7 of the 18 x 34 patch (linear index P); MFMA row 4 w + 2 dy + dx = conv0 pixel (dy, dx) of window w.
    // A fragment: k = tap 3 r + s.  A lane reads, per tap row, the two dwords that hold its three taps (patch columns 2 px + 1 + dx + s; the pair starts at
    // column 2 (px + dx)) and shifts them into place; the upper half of the wave (k = 8 ..) keeps tap 8 only.
    const int wnd = lx >> 2, dy = (lx >> 1) & 1, dx = lx & 1, odd = lx & 1;
    const uint32_t* sgd = reinterpret_cast<const uint32_t*>(smem + C::OFF_G);
    auto conv0_tile = [&](int mt) {
      const int Pm = min(mt * 8 + wnd, C::NPIX - 1);
      const int py = Pm / C::TWIN, pxx = Pm - py * C::TWIN;
      const uint32_t* gp = sgd + (2 * py + dy) * (C::GW / 2) + pxx + dx;
      uint32_t L[3], H[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const uint32_t d0 = gp[r * (C::GW / 2)], d1 = gp[r * (C::GW / 2) + 1];
        L[r] = dx ? d0 : __builtin_amdgcn_alignbit(d1, d0, 16);
        H[r] = dx ? (d1 & 0xFFFFu) : (d1 >> 16);
      }
      const u32x4 av = {qh ? H[2] : L[0], qh ? 0u : (H[0] | (L[1] << 16)), qh ? 0u : ((L[1] >> 16) | (H[1] << 16)), qh ? 0u : L[2]};
      const bf16x8 a = __builtin_bit_cast(bf16x8, av);
      f32x16 c0[2];
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[nh][r] = 0.f;
        c0[nh] = mfma_32x32x16_a16(a, bw[nh], c0[nh]);
      }
      // the lane's two store positions (windows g = odd, 2 + odd): pixel P of the patch -> slot, in the map or conv1's zero padding
      int soff[2];
      bool in_map[2];
#pragma unroll
      for (int gp2 = 0; gp2 < 2; ++gp2) {
        const int P = min(mt * 8 + 2 * (2 * gp2 + odd) + qh, C::NPIX - 1);      // (past the end: the last pixel once more -- only in the last tile, whose
        const int qy = P / C::TWIN, qx = P - qy * C::TWIN;                       //  lanes of the clamped windows hold that pixel's values: Pm is clamped alike)
        in_map[gp2] = (unsigned)(qy - 1) < (unsigned)p.H && (unsigned)(ox0 - 1 + qx) < (unsigned)p.W;
        soff[gp2] = P * 32 + (((qx >> 3) & 1) << 4);
      }
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        // accumulator r: row (r & 3) + 8 (r >> 2) + 4 q = window 2 (r >> 2) + q, pixel r & 3; lane lx = channel nh * 32 + lx.
        // + bias on the sums, then max(window, 0): the rounding is monotonic, so round(max(..)) == max(round(..)) bit for bit (and the additions come
        // first because a maximum of raw MFMA results costs a canonicalising instruction per operand)
        float mv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float t0 = c0[nh][g * 4] + bs0[nh], t1 = c0[nh][g * 4 + 1] + bs0[nh], t2 = c0[nh][g * 4 + 2] + bs0[nh], t3 = c0[nh][g * 4 + 3] + bs0[nh];
          mv[g] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(t0, t1), t2), __builtin_fmaxf(t3, 0.f));
        }
        // lane pairs (channel 2 c, 2 c + 1) trade windows so that each stores dwords: the even lane windows g = 0, 2, the odd lane g = 1, 3
        const int ch = nh * 32 + (lx & ~1), sl = ch >> 4, kq = (ch >> 3) & 1, e = ch & 7;
#pragma unroll
        for (int gp2 = 0; gp2 < 2; ++gp2) {
          const float keep = odd ? mv[2 * gp2 + 1] : mv[2 * gp2], send = odd ? mv[2 * gp2] : mv[2 * gp2 + 1];
          const float recv = dpp_xor1(send);
          const uint32_t dw = pack_bf16x2(odd ? recv : keep, odd ? keep : recv);
          *reinterpret_cast<uint32_t*>(smem + sl * C::IMG_BYTES + (soff[gp2] ^ (kq << 4)) + e * 2) = in_map[gp2] ? dw : 0u;
        }
      }
    };
    for (int mt = wave; mt < C::NMT; mt += 2 * C::NWV) {      // two tiles per turn: independent chains (two waves per SIMD hide nothing)
      conv0_tile(mt);
      conv0_tile(min(mt + C::NWV, C::NMT - 1));               // (past the end: the last tile once more, same values)
    }
  }

  // ---- conv1: K loop over the resident images; the weights are the MFMA's A operand: D = [channel][pixel]
  f32x16 acc[C::MT][2];
#pragma unroll
  for (int m = 0; m < C::MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  int a_lane[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) a_lane[s] = ((C::MT * wm) * C::TWIN + lx + s) * 32 + ((qh ^ (((lx + s) >> 3) & 1)) << 4);
  const int b_lane = (wn * 64 + lx) * 32 + ((qh ^ ((lx >> 3) & 1)) << 4);
  bf16x8 fa[2][C::MT], fb[2][2];
  auto load_frags = [&](int c, int tap, int slot) {
    const char* sa = smem + c * C::IMG_BYTES;
    const char* sw = smem + C::OFF_W + (c & 1) * C::W_BYTES;
    const int r = tap / 3, s = tap - 3 * r;
    fb[slot][0] = *reinterpret_cast<const bf16x8*>(sw + b_lane + (tap * C::NW) * 32);
    fb[slot][1] = *reinterpret_cast<const bf16x8*>(sw + b_lane + (tap * C::NW + 32) * 32);
#pragma unroll
    for (int m = 0; m < C::MT; ++m) fa[slot][m] = *reinterpret_cast<const bf16x8*>(sa + a_lane[s] + ((m + r) * C::TWIN) * 32);
  };
  auto mma_tap = [&](int slot) {
#pragma unroll
    for (int m = 0; m < C::MT; ++m) {
      acc[m][0] = mfma_32x32x16_a16(fb[slot][0], fa[slot][m], acc[m][0]);
      acc[m][1] = mfma_32x32x16_a16(fb[slot][1], fa[slot][m], acc[m][1]);
    }
  };
  auto slice = [&](int c, auto par_tag, auto more_tag) {
    constexpr int P = decltype(par_tag)::value;
    constexpr bool MORE = decltype(more_tag)::value;
    static_for<8>([&](auto tap_c) {
      constexpr int tap = decltype(tap_c)::value;
      load_frags(c, tap + 1, (P + tap + 1) & 1);
      mma_tap((P + tap) & 1);
      if constexpr (MORE) {
#pragma unroll
        for (int j = 2 * tap; j < 2 * tap + 2 && j < C::SLOTS; ++j) issue_one(j, c + 1, (c + 1) & 1);
      }
    });
    static_for<8>([&](auto tap_c) {
      constexpr int tap = decltype(tap_c)::value;
      constexpr int nd = MORE ? (C::SLOTS > 2 * tap ? 1 : 0) + (C::SLOTS > 2 * tap + 1 ? 1 : 0) : 0;
      tap_groups<2 + C::MT, 2 * C::MT, nd>();
    });
    if constexpr (MORE) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      load_frags(c + 1, 0, (P + 1) & 1);
    }
    mma_tap((P + 8) & 1);
    tap_groups<MORE ? 2 + C::MT : 0, 2 * C::MT, 0>();
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();            // the images are written, slice 0's weights have landed
  load_frags(0, 0, 0);
  for (int c = 0; c + 2 < nslices; c += 2) {
    slice(c, std::integral_constant<int, 0>{}, std::true_type{});
    slice(c + 1, std::integral_constant<int, 1>{}, std::true_type{});
  }
  slice(nslices - 2, std::integral_constant<int, 0>{}, std::true_type{});
  slice(nslices - 1, std::integral_constant<int, 1>{}, std::false_type{});

  // ---- epilogue from the accumulators: MaxPool2d(2) = the lane's two rows (in-lane) and the neighbouring lane's column (DPP), taken on the raw sums --
  // bias, ReLU and the rounding are monotonic, the result equals pooling the stored map bit for bit.  Of a lane pair the even lane keeps the wave's first
  // 32 channels, the odd lane the second 32: every lane stores.
  {
    const int odd = lx & 1;
    const int n0 = wn * 64 + odd * 32;
    f32x4 bsv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bsv[g] = *reinterpret_cast<const f32x4*>(p.bias + n0 + 8 * g + 4 * qh);
    const int PHo = p.Ho >> 1, PWo = p.Wo >> 1;
#pragma unroll
    for (int rp = 0; rp < C::MT / 2; ++rp) {
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float t0 = fmaxf(acc[2 * rp][0][k], acc[2 * rp + 1][0][k]), t1 = fmaxf(acc[2 * rp][1][k], acc[2 * rp + 1][1][k]);
        const float keep = odd ? t1 : t0, send = odd ? t0 : t1;
        v[k] = fmaxf(keep, dpp_xor1(send));
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v[4 * g + 0] = fmaxf(v[4 * g + 0] + bsv[g].x, 0.f);
        v[4 * g + 1] = fmaxf(v[4 * g + 1] + bsv[g].y, 0.f);
        v[4 * g + 2] = fmaxf(v[4 * g + 2] + bsv[g].z, 0.f);
        v[4 * g + 3] = fmaxf(v[4 * g + 3] + bsv[g].w, 0.f);
      }
      uint32_t d[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
      const u32x2 s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
      const u32x2 s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
      const u32x2 s2 = __builtin_amdgcn_permlane32_swap(d[4], d[6], false, false);
      const u32x2 s3 = __builtin_amdgcn_permlane32_swap(d[5], d[7], false, false);
      const u32x4 run0 = {s0.x, s1.x, s0.y, s1.y};     // channels n0 + 8 q .. + 7
      const u32x4 run1 = {s2.x, s3.x, s2.y, s3.y};     // channels n0 + 16 + 8 q .. + 7
      const int oy = (C::MT * wm) / 2 + rp, ox = (ox0 + lx) >> 1;
      if (oy < PHo && ox < PWo) {
        bf16_t* op = p.out + (((size_t)b * PHo + oy) * PWo + ox) * p.out_cstride + p.out_coff + n0 + 8 * qh;
        *reinterpret_cast<u32x4*>(op) = run0;
        *reinterpret_cast<u32x4*>(op + 16) = run1;
      }
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// Depthwise k x k (+ BN + activation) -> pointwise 1x1 (+ BN + activation) in ONE kernel: LCNet's DepthwiseSeparable (picodet/lcnet.py:64-90), CSP-PAN's
// DPModule (csp_pan.py:56-105) and the PicoFeat towers (pico_head.py:98-107), single-pass modes.  Run as two launches, every block writes its depthwise
// output to HBM for the pointwise conv to read back -- half of the pair's traffic -- and the layout net is ~130 launches of 50-150 us.  Here a
// workgroup (four waves) owns 4 x 32 output pixels and ALL output channels: per 32-channel chunk it stages the input halo in LDS, computes the
// depthwise taps on the VALU (fp32, the same (ky, kx) order and the same fused multiply-adds as dwconv_tile_kernel, bias, activation, rounded to the
// storage format), writes that 128 x 32 tile to LDS as the MFMA's pixel operand (64-byte rows, slots XOR-swizzled by ((pixel >> 2) & 3)) and multiplies
// it with the chunk's pointwise weights (fragments straight from global memory / L2, requested before the taps are computed); a wave keeps its 32
// pixels x Cout accumulators, the epilogue is epilogue_direct_row.  The depthwise tensor never exists in HBM; every output bit equals the two launches'.
// ---------------------------------------------------------------------------------------------------
struct DwPwK {
  const bf16_t* in;      // [B, H, W, C]
  const float* dw_w;     // fp32 [K * K][C]
  const float* dw_b;     // fp32 [C]
  int dw_act;            // 0 none, 1 ReLU, 2 hardswish
  int B, H, W, C, Ho, Wo, tiles_x, tiles_y;
  ConvK pw;              // the pointwise layer as a 1x1 ConvK on the depthwise output map (w, bias, out, out_cstride, out_coff, relu, N, n_valid, Ho, Wo)
};

template <int K, int NB>      // NB: 32-channel blocks of the pointwise output (Cout / 32); stride 1 (a stride-2 halo of 11 x 67 pixels is 70 KB per chunk)
__global__ __launch_bounds__(256, (NB == 2 && K == 3) ? 4 : 2) void dwpw_kernel(DwPwK p) {
#pragma clang fp contract(fast)
  a16_kernel_enter();
  constexpr int PAD = K / 2, TH = 4, TW = 32, PX = 2, CB = 32, CGN = CB / 8, S = 1;
  constexpr int THIN = (TH - 1) * S + K, TWIN = (TW - 1) * S + K, NCOL = (PX - 1) * S + K;
  constexpr int PITCH = CB * 2 + 32;             // 96 bytes per staged pixel (dwconv_tile_kernel's pitch at 32 channels)
  constexpr int NPIECE = THIN * TWIN * CGN;
  __shared__ __attribute__((aligned(16))) char s_in[THIN * TWIN * PITCH];
  __shared__ __attribute__((aligned(16))) char s_a[TH * TW * 64];          // depthwise output tile: [pixel][32 channels], swizzled 16-byte slots
  __shared__ __attribute__((aligned(16))) float s_w[K * K * CB + CB];
  __shared__ __attribute__((aligned(16))) char s_pw[NB * 32 * 64];         // the chunk's pointwise weights: [output][32 channels], the same swizzle
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  int L = blockIdx.x;
  const int txi = L % p.tiles_x;
  L /= p.tiles_x;
  const int tyi = L % p.tiles_y;
  const int bi = L / p.tiles_y;
  const int oy0 = tyi * TH, ox0 = txi * TW;
  const bf16_t* in_b = p.in + (size_t)bi * p.H * p.W * p.C;
  const int nchunks = p.C >> 5;

  f32x16 acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  // depthwise work item of this thread: 8 channels x 2 horizontally adjacent output pixels
  const int cg = tid & (CGN - 1), g = tid >> 2;
  const int row = g >> 4, colp = g & 15;

  for (int c = 0; c < nchunks; ++c) {
    // the chunk's pointwise weights, 16-byte pieces: output n = piece >> 2 (tile n >> 6 of the 1x1 tiling [N/64][C/32][64][32]), channel run piece & 3;
    // requested now, stored to LDS behind the barrier below -- the loads fly while the halo is staged
    constexpr int WP = NB * 32 * 4 / 256;          // pieces per thread
    u32x4 wreg[WP];
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const int piece = tid + j * 256, n = piece >> 2, cr = piece & 3;
      wreg[j] = *reinterpret_cast<const u32x4*>(p.pw.w + ((size_t)(n >> 6) * nchunks + c) * 2048 + (n & 63) * 32 + cr * 8);
    }
    if (c) __syncthreads();      // the previous chunk's taps have been read (s_in, s_w) and its A tile multiplied (s_a)
#pragma unroll
    for (int j = 0; j < (NPIECE + 255) / 256; ++j) {
      const int idx = tid + j * 256;
      if (idx < NPIECE) {
        const int pix = idx / CGN, cc = idx - pix * CGN;
        const int iy = pix / TWIN, ix = pix - iy * TWIN;
        const int gy = oy0 * S - PAD + iy, gx = ox0 * S - PAD + ix;
        u32x4 v = {0u, 0u, 0u, 0u};
        if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) v = *reinterpret_cast<const u32x4*>(in_b + ((size_t)gy * p.W + gx) * p.C + c * CB + cc * 8);
        *reinterpret_cast<u32x4*>(s_in + pix * PITCH + cc * 16) = v;
      }
    }
    for (int i = tid; i < K * K * CB; i += 256) s_w[i] = p.dw_w[(size_t)(i / CB) * p.C + c * CB + (i % CB)];
    if (tid < CB) s_w[K * K * CB + tid] = p.dw_b[c * CB + tid];
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const int piece = tid + j * 256, n = piece >> 2, cr = piece & 3;
      *reinterpret_cast<u32x4*>(s_pw + n * 64 + ((cr ^ ((n >> 2) & 3)) << 4)) = wreg[j];
    }
    __syncthreads();
    float a2[PX][8];
#pragma unroll
    for (int px = 0; px < PX; ++px)
#pragma unroll
      for (int k8 = 0; k8 < 8; ++k8) a2[px][k8] = 0.f;
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
      float col[NCOL][8];
#pragma unroll
      for (int j = 0; j < NCOL; ++j) {
        const u32x4 h = *reinterpret_cast<const u32x4*>(s_in + ((row * S + ky) * TWIN + colp * PX * S + j) * PITCH + cg * 16);
        const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) col[j][k8] = bf16_to_f32((k8 & 1) ? (hw[k8 >> 1] >> 16) : (hw[k8 >> 1] & 0xFFFFu));
      }
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const float4 w0 = *reinterpret_cast<const float4*>(s_w + (ky * K + kx) * CB + cg * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(s_w + (ky * K + kx) * CB + cg * 8 + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int px = 0; px < PX; ++px)
#pragma unroll
          for (int k8 = 0; k8 < 8; ++k8) a2[px][k8] += col[px * S + kx][k8] * wv[k8];
      }
    }
    {
      const float* bv = s_w + K * K * CB + cg * 8;
#pragma unroll
      for (int px = 0; px < PX; ++px) {
        uint32_t hb[8];
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
          float v = a2[px][k8] + bv[k8];
          if (p.dw_act == 2) v = v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * 0.16666667f;      // (layout_kernels.hip's hswish)
          else if (p.dw_act == 1) v = fmaxf(v, 0.f);
          hb[k8] = f32_to_bf16(v);
        }
        const int P = row * TW + colp * PX + px;
        const u32x4 o = {hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16)};
        *reinterpret_cast<u32x4*>(s_a + P * 64 + ((cg ^ ((P >> 2) & 3)) << 4)) = o;
      }
    }
    __syncthreads();
    // pointwise: wave w multiplies row-tile w (pixels 32 w .. 32 w + 31) with every output block; D = [channel][pixel]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int P = wave * TW + lx;
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(s_a + P * 64 + (((kk * 2 + q) ^ ((P >> 2) & 3)) << 4));
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = nb * 32 + lx;
        const bf16x8 fb = *reinterpret_cast<const bf16x8*>(s_pw + n * 64 + (((kk * 2 + q) ^ ((n >> 2) & 3)) << 4));
        acc[nb] = mfma_32x32x16_a16(fb, fa, acc[nb]);
      }
    }
  }
  // epilogue from the accumulators, 64 output channels at a time
  const int oy = oy0 + wave;
#pragma unroll
  for (int j = 0; j < NB / 2; ++j) {
    const DirectBias bs = direct_bias<2>(p.pw, 64 * j, q);
    epilogue_direct_row<2>(p.pw, *reinterpret_cast<const f32x16(*)[2]>(&acc[2 * j]), bs, bi, oy, ox0, lx, 64 * j, q, nullptr);
  }
}

// ---------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution 64 -> 64, weight-stationary and persistent ("ws64", bf16 mode).
// The five 64 -> 64 @240^2 layers of DB-ResNet18 (layer1 + the fused out2; db_net/dbnet.py:102-140, 615-638) are the detector's
// largest item and the furthest from the matrix roofline (0.54 PF on the v3 4-wave tile): K is only 576, so a 16x32 tile
// is four K-slices -- a DMA round trip of prologue, two fp32 passes through LDS of epilogue -- and every tile re-fetches the
// layer's whole 72 KB weight matrix, which is HALF of its LDS fill traffic (73.7 KB weights against 78 KB of input patch).
// Here ONE workgroup per CU (8 waves) keeps the whole weight matrix in LDS as ready-made MFMA fragments for the kernel's
// lifetime and walks a contiguous run of 16x32-pixel tiles.  The input patches stream through two 32-channel slice buffers
// (global_load_lds, 64-byte pixel rows with the 16-byte slots XOR-swizzled by ((pixel >> 2) & 3): conflict-free ds_read_b128
// for 16 consecutive pixels); the slice pipeline runs ACROSS tile boundaries, so only the first tile of a workgroup pays a
// prologue.  The MFMA takes the weights as its A operand (D = [channel][pixel]): a lane owns a pixel, the epilogue (bias,
// residual, ReLU, bf16 packing, v_permlane32_swap into 16-byte channel runs) runs from the accumulators while the NEXT
// tile's first slice is already in LDS, its stores drain under that slice's MFMAs, and the residual is fetched into
// registers one slice ahead.  LDS: 72 KB weights + 2 x 39 KB slices = 150 KB.
// ---------------------------------------------------------------------------------------------------
#ifndef PT_WS_ABL
#define PT_WS_ABL 0      // ablation bits (timing only): 1 no stores, 4 no epilogue, 8 no input DMA after the first tile, 16 no MFMA loop
#endif
struct Ws64Cfg {
  static constexpr int NWV = 8, NTHR = 512;
  static constexpr int TH = 16, TW = 32, THIN = TH + 2, TWIN = TW + 2;
  static constexpr int NPIX = THIN * TWIN;                     // 612
  static constexpr int W_FRAGS = 9 * 4 * 2;                    // (tap, 16-channel k-step, 32-output half): 1 KB each
  static constexpr int W_BYTES = W_FRAGS * 1024;               // 73 728
  static constexpr int IN_UNITS = NPIX * 4;                    // 16-byte units of a 32-channel slice
  static constexpr int IN_INSTR = (IN_UNITS + 63) / 64;        // 39 wave-wide DMA instructions
  static constexpr int IN_SLOTS = (IN_INSTR + NWV - 1) / NWV;  // 5 per wave
  static constexpr int IN_BYTES = IN_INSTR * 1024;             // 39 936
  static constexpr int SMEM = W_BYTES + 2 * IN_BYTES;          // 153 600
};

__global__ __launch_bounds__(512, 1) void conv3x3_ws64_kernel(ConvK p, const bf16_t* __restrict__ zero_page) {
#if defined(__HIP_DEVICE_COMPILE__)      // (buffer-resource builtins: device pass only, see conv3x3_pipe_kernel)
  a16_kernel_enter();
  using C = Ws64Cfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_w = smem;
  char* s_in0 = smem + C::W_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = lane & 31, q = lane >> 5;

  // this workgroup's run of tiles (tile = (image, tile row, tile column), x fastest)
  const int G = gridDim.x;
  const int Lw = xcd_remap(blockIdx.x, G);
  const int t0 = (int)((long long)Lw * p.total_tiles / G), t1 = (int)((long long)(Lw + 1) * p.total_tiles / G);
  if (t0 >= t1) return;
  const int nsl = 2 * (t1 - t0);
  const int tiles_img = p.tiles_x * p.tiles_y;

  // weights -> LDS, once: fragment f = (tap * 4 + ks) * 2 + nh holds, for lane (lx, q), channels ks * 16 + 8 q .. + 7 of output
  // nh * 32 + lx; source = the v1 tiling [Cin/32 = 2][9][64][32]
  // LDS-DMA requests are MUBUF (buffer_load ... lds), like conv3x3_pipe_kernel's: a FLAT-encoded LDS load in flight makes hipcc wait lgkmcnt(0) in front of
  // every dependent ds_read result (no read could stay in flight across an MFMA group), and a lane beyond num_records writes zeros (no zero page)
  constexpr int OOB = 0x7FFFF000;
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, OOB, 0x00020000);
#pragma unroll
  for (int j = 0; j < C::W_FRAGS / C::NWV; ++j) {
    const int f = wave + C::NWV * j;
    const int tap = f >> 3, ks = (f >> 1) & 3, nh = f & 1;
    const int off = ((((ks >> 1) * 9 + tap) * 64 + nh * 32 + lx) * 32 + (ks & 1) * 16 + q * 8) * 2;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)(s_w + f * 1024), 16, off, 0, 0, 0);
  }

  // input units: 16-byte slot U & 3 of pixel U >> 2 holds channel quarter (U & 3) ^ ((patch column >> 2) & 3): conflict-free ds_read_b128 for 16 lanes on
  // consecutive columns, and a fragment address is (per tap column, per k-step) base + row * immediate
  int voff[C::IN_SLOTS];
  const bf16_t* in_b = p.in;
  auto issue = [&](int sl) {
    if (!(sl & 1)) {      // first slice of a tile: where its patch lies
      const int t = t0 + (sl >> 1);
      const int b = t / tiles_img, r = t - b * tiles_img;
      const int tyi = r / p.tiles_x, txi = r - tyi * p.tiles_x;
      const int oy0 = tyi * C::TH, ox0 = txi * C::TW;
      in_b = p.in + (size_t)b * p.H * p.W * 64;
#pragma unroll
      for (int j = 0; j < C::IN_SLOTS; ++j) {
        int k = wave + C::NWV * j;
        if (k >= C::IN_INSTR) k = C::IN_INSTR - 1;      // (a slot past the list repeats the last instruction: no branch in the issue path)
        const int U = k * 64 + lane;
        const int pix = U >> 2;
        const int iy = pix / C::TWIN, ix = pix - iy * C::TWIN;
        const int qq = (U & 3) ^ ((ix >> 2) & 3);
        const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
        const bool inside = U < C::IN_UNITS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        voff[j] = inside ? (int)((((size_t)gy * p.W + gx) * 64 + qq * 8) * 2) : OOB;
      }
    }
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(in_b), 0, OOB, 0x00020000);
    char* lds = s_in0 + (sl & 1) * C::IN_BYTES;
#pragma unroll
    for (int j = 0; j < C::IN_SLOTS; ++j) {
      int k = wave + C::NWV * j;
      if (k >= C::IN_INSTR) k = C::IN_INSTR - 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (__attribute__((address_space(3))) void*)(lds + k * 1024), 16, voff[j], (sl & 1) * 64, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  u32x4 rres[2][2][2];                                   // residual of the tile in flight: [row][32-channel block][16-byte run]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int h = 0; h < 2; ++h) rres[m][n][h] = u32x4{0u, 0u, 0u, 0u};

  // bias in the accumulator layout: register r of block nb = channel nb * 32 + (r & 3) + 8 (r >> 2) + 4 q
  f32x4 bs[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int g = 0; g < 4; ++g) bs[nb][g] = *reinterpret_cast<const f32x4*>(p.bias + nb * 32 + 8 * g + 4 * q);

  // pixel of this lane in tile t: -> (image, row of patch row 2 wave, column); false: outside the map
  auto tile_pos = [&](int t, int& b, int& oy, int& ox) {
    b = t / tiles_img;
    const int r = t - b * tiles_img;
    const int tyi = r / p.tiles_x, txi = r - tyi * p.tiles_x;
    oy = tyi * C::TH + 2 * wave;
    ox = txi * C::TW + lx;
  };
  auto load_res = [&](int t) {
    int b, oy, ox;
    tile_pos(t, b, oy, ox);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (oy + m < p.Ho && ox < p.Wo) {
        const bf16_t* rp = p.res + (((size_t)b * p.Ho + oy + m) * p.Wo + ox) * 64 + 8 * q;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          rres[m][nb][0] = *reinterpret_cast<const u32x4*>(rp + nb * 32);
          rres[m][nb][1] = *reinterpret_cast<const u32x4*>(rp + nb * 32 + 16);
        }
      }
    }
  };
  auto epilogue = [&](int t) {
    int b, oy, ox;
    tile_pos(t, b, oy, ox);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const bool inside = oy + m < p.Ho && ox < p.Wo;
      bf16_t* op = p.out + (((size_t)b * p.Ho + oy + m) * p.Wo + ox) * p.out_cstride + p.out_coff + 8 * q;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        float v[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          v[4 * g + 0] = acc[m][nb][4 * g + 0] + bs[nb][g].x;
          v[4 * g + 1] = acc[m][nb][4 * g + 1] + bs[nb][g].y;
          v[4 * g + 2] = acc[m][nb][4 * g + 2] + bs[nb][g].z;
          v[4 * g + 3] = acc[m][nb][4 * g + 3] + bs[nb][g].w;
        }
        if (p.res_mode) {
          // the 16-byte runs back into the accumulator layout: the swap of the store path is its own inverse
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const u32x4 rr = rres[m][nb][h];
            const u32x2 a = __builtin_amdgcn_permlane32_swap(rr.x, rr.z, false, false);
            const u32x2 c = __builtin_amdgcn_permlane32_swap(rr.y, rr.w, false, false);
            v[8 * h + 0] += bf16lo_f32(a.x); v[8 * h + 1] += bf16hi_f32(a.x);
            v[8 * h + 2] += bf16lo_f32(c.x); v[8 * h + 3] += bf16hi_f32(c.x);
            v[8 * h + 4] += bf16lo_f32(a.y); v[8 * h + 5] += bf16hi_f32(a.y);
            v[8 * h + 6] += bf16lo_f32(c.y); v[8 * h + 7] += bf16hi_f32(c.y);
          }
        }
        if (p.relu == 1) {
#pragma unroll
          for (int k = 0; k < 16; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        uint32_t d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
        const u32x2 s2 = __builtin_amdgcn_permlane32_swap(d[4], d[6], false, false);
        const u32x2 s3 = __builtin_amdgcn_permlane32_swap(d[5], d[7], false, false);
#if PT_WS_ABL & 1
        if (inside && s0.x == 0x12345u) {
#else
        if (inside) {
#endif
          *reinterpret_cast<u32x4*>(op + nb * 32) = u32x4{s0.x, s1.x, s0.y, s1.y};
          *reinterpret_cast<u32x4*>(op + nb * 32 + 16) = u32x4{s2.x, s3.x, s2.y, s3.y};
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][nb][r] = 0.f;
      }
    }
  };

  int a_lane[3][2];      // fragment base of (tap column s, k-step kk): pixel (2 wave, lx + s) of the patch, 16-byte slot (kk * 2 + q) ^ ((column >> 2) & 3)
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) a_lane[s][kk] = ((2 * wave) * C::TWIN + lx + s) * 64 + (((kk * 2 + q) ^ (((lx + s) >> 2) & 3)) << 4);
  issue(0);
  for (int sl = 0; sl < nsl; ++sl) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#if PT_WS_ABL & 8
    if (sl + 1 < nsl && sl < 2) issue(sl + 1);
#else
    if (sl + 1 < nsl) issue(sl + 1);
#endif
    if (sl & 1) {
#if !(PT_WS_ABL & 2)
      if (p.res_mode) load_res(t0 + (sl >> 1));          // lands under this slice's MFMAs, used after the next barrier
#endif
    } else if (sl) {
#if !(PT_WS_ABL & 4)
      epilogue(t0 + (sl >> 1) - 1);                      // the stores drain under this slice's MFMAs
#endif
    }
#if PT_WS_ABL & 16
    if (sl > 1 && p.B != 12345) continue;
#endif
    const char* s_in = s_in0 + (sl & 1) * C::IN_BYTES;
    const char* s_wc = s_w + (sl & 1) * 4096 + lane * 16;   // k-steps 2 c, 2 c + 1 of every tap
    // 18 steps (tap, k-step) of {2 weight + 2 image fragments, 4 MFMAs}; the fragments of step i + 1 are requested between the MFMAs of step i
    // (M R M R M R M R, pinned with scheduling groups: left alone the reads sink to just above their use and every step waits for LDS)
    bf16x8 fw[2][2], fa[2][2];
    auto load_step = [&](int st, int slot) {
      const int tap = st >> 1, kk = st & 1, r = tap / 3, s = tap - 3 * r;
      fw[slot][0] = *reinterpret_cast<const bf16x8*>(s_wc + (tap * 8 + kk * 2) * 1024);
      fw[slot][1] = *reinterpret_cast<const bf16x8*>(s_wc + (tap * 8 + kk * 2 + 1) * 1024);
#pragma unroll
      for (int m = 0; m < 2; ++m) fa[slot][m] = *reinterpret_cast<const bf16x8*>(s_in + a_lane[s][kk] + ((m + r) * C::TWIN) * 64);
    };
    load_step(0, 0);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
      if (st < 17) load_step(st + 1, (st + 1) & 1);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        acc[m][0] = mfma_32x32x16_a16(fw[st & 1][0], fa[st & 1][m], acc[m][0]);      // D = [channel][pixel]
        acc[m][1] = mfma_32x32x16_a16(fw[st & 1][1], fa[st & 1][m], acc[m][1]);
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    static_for<18>([&](auto st_c) {
      constexpr int st = decltype(st_c)::value;
      tap_groups<st < 17 ? 4 : 0, 4, 0>();
    });
  }
  epilogue(t1 - 1);
#endif
}

// ---------------------------------------------------------------------------------------------------
// Stem: 7x7 stride-2 pad-3 conv on a 4-channel (RGB0) bf16 image, 64 outputs, bias + ReLU.
// K is laid out [r=7][s=8][c=4] = 224 (tap s=7 and channel 3 carry zero weights), so that one MFMA
// k-step (16) = 4 horizontally adjacent pixels x 4 channels = 32 contiguous bytes of the image row.
// ---------------------------------------------------------------------------------------------------
template <int S>
struct StemCfg {
  static constexpr int TH = 8, TW = 32;
  static constexpr int THIN = (TH - 1) * S + 7;         // 21 (S=2) / 14 (S=1)
  static constexpr int TWIN = ((TW - 1) * S + 8 + 1) & ~1;  // 70 / 40 (even: staged as pixel pairs)
  static constexpr int IN_BYTES = THIN * TWIN * 8;
  static constexpr int WROW = 464;                  // 224 bf16 = 448 B + 16 B pad (odd number of 16-B slots)
  static constexpr int W_BYTES = 64 * WROW;
  static constexpr int STAGE_BYTES = TH * TW * 64 * 4;
  static constexpr int SMEM = STAGE_BYTES;  // > IN_BYTES + W_BYTES (41456 at S=2)
  static constexpr int HP = TWIN / 2;              // pixel pairs per patch row
  static constexpr int NP_IN = THIN * HP;
  static constexpr int NI = (NP_IN + 255) / 256;
  static constexpr int NP_W = 64 * 28;             // 1792 16-byte pieces
  static constexpr int NWP = NP_W / 256;           // 7
};

// 7x7 pad-3 convolution of a 3-channel image stored NHWC4 ([r g b 0] bf16 per pixel), stride S, 64 GEMM outputs
// (ResNet-18 stem: S=2, 64 channels, db_net/dbnet.py:272; DLA-34 base_layer: S=1, 16 channels + zero padding,
// center_net/modeling_centernet.py:291-294).  K = [7 ky][8 kx][4 c] = 224: an A fragment is two horizontally
// adjacent input pixels, so at S=1 fragments are only 8-byte aligned and are read as two ds_read_b64.
template <int S, int NH>   // NH: 32-column halves of the 64 GEMM outputs that are computed (1 when n_valid <= 32)
__global__ __launch_bounds__(256, 2) void conv_stem7x7_kernel(ConvK p) {
  a16_kernel_enter();
  using C = StemCfg<S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_in = smem;
  char* s_w = smem + C::IN_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;

  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int txi = L % p.tiles_x;
  L /= p.tiles_x;
  const int tyi = L % p.tiles_y;
  const int b = L / p.tiles_y;
  const int oy0 = tyi * C::TH, ox0 = txi * C::TW;
  const int iy0 = oy0 * S - 3, ix0 = ox0 * S - 3;
  const int ps = p.split ? 8 : 4;  // bf16 elements per input pixel: [r g b 0] or [hi rgb0 | lo rgb0]
  const bf16_t* in_b = p.in + (size_t)b * p.H * p.W * ps;

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const char* a_base = s_in + (((wave * 2) * S) * C::TWIN + S * lx + 2 * q) * 8;
  const char* b_base = s_w + lx * C::WROW + q * 16;
  // split mode: three passes (x_hi, w_hi), (x_lo, w_hi), (x_hi, w_lo) into the same accumulators
  const int npass = p.split ? 3 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const int po = (pass == 1) ? 4 : 0;
    const bf16_t* wsrc = p.w + (pass == 2 ? 64 * 224 : 0);
    if (pass) __syncthreads();
#pragma unroll
    for (int j = 0; j < C::NI; ++j) {
      const int idx = tid + j * 256;
      if (idx < C::NP_IN) {
        const int iy = idx / C::HP, ip = idx - iy * C::HP;
        const int gy = iy0 + iy, gx = ix0 + 2 * ip;
        u32x2 v0 = {0u, 0u}, v1 = {0u, 0u};
        if ((unsigned)gy < (unsigned)p.H) {
          const bf16_t* rowp = in_b + (size_t)gy * p.W * ps + po;
          if ((unsigned)gx < (unsigned)p.W) v0 = *reinterpret_cast<const u32x2*>(rowp + (size_t)gx * ps);
          if ((unsigned)(gx + 1) < (unsigned)p.W) v1 = *reinterpret_cast<const u32x2*>(rowp + (size_t)(gx + 1) * ps);
        }
        u32x4 v = {v0.x, v0.y, v1.x, v1.y};
        *reinterpret_cast<u32x4*>(s_in + (iy * C::TWIN + 2 * ip) * 8) = v;
      }
    }
    if (pass != 1) {  // pass 1 re-uses w_hi
#pragma unroll
      for (int j = 0; j < C::NWP; ++j) {
        const int idx = tid + j * 256;
        const int row = idx / 28, part = idx - row * 28;
        *reinterpret_cast<u32x4*>(s_w + row * C::WROW + part * 16) = *reinterpret_cast<const u32x4*>(wsrc + idx * 8);
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 7; ++r) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ks = r * 2 + h;
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(b_base + ks * 32);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(b_base + 32 * C::WROW + ks * 32);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const char* ap = a_base + ((m * S + r) * C::TWIN + 4 * h) * 8;
          bf16x8 a;
          if (S == 2) {
            a = *reinterpret_cast<const bf16x8*>(ap);
          } else {
            const u32x2 lo = *reinterpret_cast<const u32x2*>(ap), hi = *reinterpret_cast<const u32x2*>(ap + 8);
            const u32x4 av = {lo.x, lo.y, hi.x, hi.y};
            a = __builtin_bit_cast(bf16x8, av);
          }
          acc[m][0] = mfma_32x32x16_a16(a, b0, acc[m][0]);
          if (NH == 2) acc[m][1] = mfma_32x32x16_a16(a, b1, acc[m][1]);
        }
      }
    }
  }
  __syncthreads();
  float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int ty = wave * 2 + m;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tx = (r & 3) + 8 * (r >> 2) + 4 * q;
        stage[(ty * 32 + tx) * 64 + n * 32 + lx] = acc[m][n][r];
      }
  }
  __syncthreads();
  epilogue_store<C::TH, C::TW>(p, stage, tid, b, oy0, ox0, 0);
}

// ---------------------------------------------------------------------------------------------------
// ResNet-18 stem + max-pool in one kernel (db_net/dbnet.py:272-275: conv 7x7/s2 + BN + ReLU, MaxPool2d(3, 2, 1)), bf16 mode.
// conv_stem7x7_kernel<2,2> writes the 64-channel half-resolution map (29.5 MB per 960x960 page) and maxpool3x3s2_kernel
// reads it back: 1.9 GB of the detector's traffic per 32 pages, and two latency-bound launches (one 8x32 tile per
// workgroup, nothing overlapped).  Here a workgroup computes a 16x32 patch of the stem map (4 MFMA row-tiles per wave),
// rounds it to bf16 exactly as the stored map would be, keeps it in LDS and writes the 7x15 pooled pixels it covers
// (window rows 2py-1..2py+1: patch rows 2(py-py0)..+2): the stem map never exists in HBM.  1.22x the stem's MFMA work
// (16/14 x 32/30 re-computed halo) against -1.9 GB.  Results are bit-identical to stem -> store -> pool: the conv sums
// in the same order, the rounding is the same instruction, and a max over non-negative bf16 values is a max over their
// bit patterns (positions outside the stem map hold 0, which never wins against the always-valid window centre).
// The weights are the MFMA's A operand (D = [channel][pixel]): a lane owns a pixel and packs runs of four channels.
// ---------------------------------------------------------------------------------------------------
// max of the two 16-bit halves of a dword as unsigned integers, one instruction (v_pk_max_u16): the order of non-negative 16-bit floats is the order of their bits
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}
struct StemPoolCfg {
  static constexpr int TH = 16, TW = 32, PH = 7, PW = 15;
  static constexpr int THIN = (TH - 1) * 2 + 7;          // 37
  static constexpr int TWIN = 70;                        // (TW - 1) * 2 + 8, as StemCfg<2>
  static constexpr int IN_BYTES = THIN * TWIN * 8;       // 20 720
  static constexpr int WROW = 464;
  static constexpr int W_BYTES = 64 * WROW;              // 29 696
  static constexpr int PIX = 136;                        // staged pixel: 64 bf16 + 8 B (conflict-free 8-byte writes of 16 lanes)
  static constexpr int STAGE_BYTES = TH * TW * PIX;      // 69 632
  static constexpr int SMEM = STAGE_BYTES;               // > IN_BYTES + W_BYTES
  static constexpr int HP = TWIN / 2;
  static constexpr int NP_IN = THIN * HP;                // 1295 pixel pairs
  static constexpr int NI = (NP_IN + 255) / 256;         // 6
  static constexpr int NWP = 64 * 28 / 256;              // 7
};

__global__ __launch_bounds__(256, 2) void conv_stem7x7_pool_kernel(ConvK p) {
  a16_kernel_enter();
  using C = StemPoolCfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_in = smem;
  char* s_w = smem + C::IN_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  // p.Ho x p.Wo is the STEM map (H/2 x W/2); tiles_x / tiles_y count pooled tiles of PH x PW
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int txi = L % p.tiles_x;
  L /= p.tiles_x;
  const int tyi = L % p.tiles_y;
  const int b = L / p.tiles_y;
  const int py0 = tyi * C::PH, px0 = txi * C::PW;
  const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;        // first stem row / column of the patch
  const int iy0 = sy0 * 2 - 3, ix0 = sx0 * 2 - 3;
  const bf16_t* in_b = p.in + (size_t)b * p.H * p.W * 4;

#pragma unroll
  for (int j = 0; j < C::NI; ++j) {
    const int idx = tid + j * 256;
    if (idx < C::NP_IN) {
      const int iy = idx / C::HP, ip = idx - iy * C::HP;
      const int gy = iy0 + iy, gx = ix0 + 2 * ip;
      u32x2 v0 = {0u, 0u}, v1 = {0u, 0u};
      if ((unsigned)gy < (unsigned)p.H) {
        const bf16_t* rowp = in_b + (size_t)gy * p.W * 4;
        if ((unsigned)gx < (unsigned)p.W) v0 = *reinterpret_cast<const u32x2*>(rowp + (size_t)gx * 4);
        if ((unsigned)(gx + 1) < (unsigned)p.W) v1 = *reinterpret_cast<const u32x2*>(rowp + (size_t)(gx + 1) * 4);
      }
      const u32x4 v = {v0.x, v0.y, v1.x, v1.y};
      *reinterpret_cast<u32x4*>(s_in + (iy * C::TWIN + 2 * ip) * 8) = v;
    }
  }
#pragma unroll
  for (int j = 0; j < C::NWP; ++j) {
    const int idx = tid + j * 256;
    const int row = idx / 28, part = idx - row * 28;
    *reinterpret_cast<u32x4*>(s_w + row * C::WROW + part * 16) = *reinterpret_cast<const u32x4*>(p.w + idx * 8);
  }
  __syncthreads();

  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  const char* a_base = s_in + (((wave * 4) * 2) * C::TWIN + 2 * lx + 2 * q) * 8;
  const char* b_base = s_w + lx * C::WROW + q * 16;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ks = r * 2 + h;
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(b_base + ks * 32);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(b_base + 32 * C::WROW + ks * 32);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_base + ((m * 2 + r) * C::TWIN + 4 * h) * 8);
        acc[m][0] = mfma_32x32x16_a16(b0, a, acc[m][0]);     // D = [channel][pixel]
        acc[m][1] = mfma_32x32x16_a16(b1, a, acc[m][1]);
      }
    }
  }
  __syncthreads();      // the operand images are dead: the patch takes their place
  // lane (lx, q): stem pixel (sy0 + 4 wave + m, sx0 + lx); register r of block n = channel 32 n + (r & 3) + 8 (r >> 2) + 4 q
  const bool col_ok = (unsigned)(sx0 + lx) < (unsigned)p.Wo;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int row = wave * 4 + m;
    const bool ok = col_ok && (unsigned)(sy0 + row) < (unsigned)p.Ho;
    char* dst = smem + (row * C::TW + lx) * C::PIX + 8 * q;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n * 32 + 8 * g + 4 * q);
        u32x2 o = {0u, 0u};
        if (ok) {
          o.x = pack_bf16x2(fmaxf(acc[m][n][4 * g + 0] + bv.x, 0.f), fmaxf(acc[m][n][4 * g + 1] + bv.y, 0.f));
          o.y = pack_bf16x2(fmaxf(acc[m][n][4 * g + 2] + bv.z, 0.f), fmaxf(acc[m][n][4 * g + 3] + bv.w, 0.f));
        }
        *reinterpret_cast<u32x2*>(dst + (n * 32 + 8 * g) * 2) = o;
      }
  }
  __syncthreads();
  // pooled pixel (py0 + py, px0 + px), channels 4 c .. 4 c + 3: max over the 3x3 window of bit patterns (all values >= +0)
  const int PHo = p.Ho >> 1, PWo = p.Wo >> 1;
  for (int idx = tid; idx < C::PH * C::PW * 16; idx += 256) {
    const int c4 = idx & 15, pp = idx >> 4;
    const int py = pp / C::PW, px = pp - py * C::PW;
    const int oy = py0 + py, ox = px0 + px;
    if (oy >= PHo || ox >= PWo) continue;
    u32x2 mx = {0u, 0u};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(smem + ((2 * py + dy) * C::TW + 2 * px + dx) * C::PIX + c4 * 8);
        mx.x = pk_max_u16(mx.x, v.x);
        mx.y = pk_max_u16(mx.y, v.y);
      }
    *reinterpret_cast<u32x2*>(p.out + (((size_t)b * PHo + oy) * PWo + ox) * 64 + c4 * 4) = mx;
  }
}

// ---------------------------------------------------------------------------------------------------
// The same stem + max-pool, persistent and weight-stationary.  conv_stem7x7_pool_kernel spends 85 % of a tile in latency: one
// tile per workgroup, whose 29.7 KB weight image (more than its 20.7 KB input patch) and input are fetched, staged and
// waited for before the first MFMA (0.86 ms per 64 pages for 3.6 k cycles of MFMA per tile and SIMD).  Here ONE workgroup
// per CU (8 waves, two stem rows each) keeps the whole weight matrix in REGISTERS as the MFMA's A fragments (14 k-steps x
// 2 x 16 B per lane: no weight traffic and no weight ds_reads after the first tile), walks a contiguous run of tiles, and
// fetches the next tile's input patch into registers while the current one is multiplied and pooled.  Same MFMA sequence,
// same rounding, same max: bit-identical to the kernel above (test_stem_pool_fused_is_bit_identical covers both).
// ---------------------------------------------------------------------------------------------------
struct StemPoolWsCfg {
  static constexpr int NTHR = 512;
  static constexpr int NI = (StemPoolCfg::NP_IN + NTHR - 1) / NTHR;      // 3 pixel pairs per thread
};

__global__ __launch_bounds__(512, 1) void conv_stem7x7_pool_ws_kernel(ConvK p) {
  a16_kernel_enter();
  using C = StemPoolCfg;
  using W = StemPoolWsCfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // the input patch (20.7 KB), then the bf16 stem patch (69.6 KB) over it
  char* s_in = smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  const int G = gridDim.x;
  const int Lw = xcd_remap(blockIdx.x, G);
  const int t0 = (int)((long long)Lw * p.total_tiles / G), t1 = (int)((long long)(Lw + 1) * p.total_tiles / G);
  if (t0 >= t1) return;
  const int tiles_img = p.tiles_x * p.tiles_y;

  bf16x8 wreg[14][2];
#pragma unroll
  for (int ks = 0; ks < 14; ++ks)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) wreg[ks][nh] = *reinterpret_cast<const bf16x8*>(p.w + (nh * 32 + lx) * 224 + ks * 16 + q * 8);
  u32x4 rin[W::NI];
  auto prefetch = [&](int t) {
    const int b = t / tiles_img, r = t - b * tiles_img;
    const int tyi = r / p.tiles_x, txi = r - tyi * p.tiles_x;
    const int iy0 = (2 * tyi * C::PH - 1) * 2 - 3, ix0 = (2 * txi * C::PW - 1) * 2 - 3;
    const bf16_t* in_b = p.in + (size_t)b * p.H * p.W * 4;
#pragma unroll
    for (int j = 0; j < W::NI; ++j) {
      const int idx = tid + j * W::NTHR;
      u32x2 v0 = {0u, 0u}, v1 = {0u, 0u};
      if (idx < C::NP_IN) {
        const int iy = idx / C::HP, ip = idx - iy * C::HP;
        const int gy = iy0 + iy, gx = ix0 + 2 * ip;
        if ((unsigned)gy < (unsigned)p.H) {
          const bf16_t* rowp = in_b + (size_t)gy * p.W * 4;
          if ((unsigned)gx < (unsigned)p.W) v0 = *reinterpret_cast<const u32x2*>(rowp + (size_t)gx * 4);
          if ((unsigned)(gx + 1) < (unsigned)p.W) v1 = *reinterpret_cast<const u32x2*>(rowp + (size_t)(gx + 1) * 4);
        }
      }
      rin[j] = u32x4{v0.x, v0.y, v1.x, v1.y};
    }
  };
  const char* a_base = s_in + (((wave * 2) * 2) * C::TWIN + 2 * lx + 2 * q) * 8;
  const int PHo = p.Ho >> 1, PWo = p.Wo >> 1;
  prefetch(t0);
  for (int t = t0; t < t1; ++t) {
#pragma unroll
    for (int j = 0; j < W::NI; ++j) {
      const int idx = tid + j * W::NTHR;
      if (idx < C::NP_IN) {
        const int iy = idx / C::HP, ip = idx - iy * C::HP;
        *reinterpret_cast<u32x4*>(s_in + (iy * C::TWIN + 2 * ip) * 8) = rin[j];
      }
    }
    __syncthreads();
    if (t + 1 < t1) prefetch(t + 1);      // in flight under the MFMAs and the pooling of this tile
    const int b = t / tiles_img, rr = t - b * tiles_img;
    const int tyi = rr / p.tiles_x, txi = rr - tyi * p.tiles_x;
    const int py0 = tyi * C::PH, px0 = txi * C::PW;
    const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ks = r * 2 + h;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_base + ((m * 2 + r) * C::TWIN + 4 * h) * 8);
          acc[m][0] = mfma_32x32x16_a16(wreg[ks][0], a, acc[m][0]);     // D = [channel][pixel]
          acc[m][1] = mfma_32x32x16_a16(wreg[ks][1], a, acc[m][1]);
        }
      }
    }
    __syncthreads();      // the input patch is dead: the stem patch takes its place
    const bool col_ok = (unsigned)(sx0 + lx) < (unsigned)p.Wo;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int row = wave * 2 + m;
      const bool ok = col_ok && (unsigned)(sy0 + row) < (unsigned)p.Ho;
      char* dst = smem + (row * C::TW + lx) * C::PIX + 8 * q;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n * 32 + 8 * g + 4 * q);      // 256 B, cache-resident: not worth 32 registers
          u32x2 o = {0u, 0u};
          if (ok) {
            o.x = pack_bf16x2(fmaxf(acc[m][n][4 * g + 0] + bv.x, 0.f), fmaxf(acc[m][n][4 * g + 1] + bv.y, 0.f));
            o.y = pack_bf16x2(fmaxf(acc[m][n][4 * g + 2] + bv.z, 0.f), fmaxf(acc[m][n][4 * g + 3] + bv.w, 0.f));
          }
          *reinterpret_cast<u32x2*>(dst + (n * 32 + 8 * g) * 2) = o;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < C::PH * C::PW * 16; idx += W::NTHR) {
      const int c4 = idx & 15, pp = idx >> 4;
      const int py = pp / C::PW, px = pp - py * C::PW;
      const int oy = py0 + py, ox = px0 + px;
      if (oy >= PHo || ox >= PWo) continue;
      u32x2 mx = {0u, 0u};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const u32x2 v = *reinterpret_cast<const u32x2*>(smem + ((2 * py + dy) * C::TW + 2 * px + dx) * C::PIX + c4 * 8);
          mx.x = pk_max_u16(mx.x, v.x);
          mx.y = pk_max_u16(mx.y, v.y);
        }
      *reinterpret_cast<u32x2*>(p.out + (((size_t)b * PHo + oy) * PWo + ox) * 64 + c4 * 4) = mx;
    }
    __syncthreads();      // the stem patch has been read: the next input patch may land on it
  }
}

// ---------------------------------------------------------------------------------------------------
// Thin stride-1 stem (DLA-34 base_layer: 3 -> 16 channels at full resolution, bf16 mode).  conv_stem7x7_kernel<1,1> spends
// most of its time around the MFMAs: every 8x32-pixel workgroup re-loads the 29 KB weight image from L2 (3.7 GB per
// 32 tables) and sends 256 x 64 fp32 through LDS to store 16 channels.  Here a workgroup walks STEM_NT tiles of a row
// strip with the 32 computed weight rows staged once, and the epilogue is done from the accumulators: the MFMA runs with
// the weights as its A operand, so a lane owns one pixel and stores its channels as 8-byte runs of four.
// ---------------------------------------------------------------------------------------------------
constexpr int STEM_NT = 8;

// SPLIT (PT_PRECISION_BF16X3): pixels are [hi rgb0 | lo rgb0], the weights [hi | lo][64][224]; three passes (x_hi, w_hi), (x_lo, w_hi),
// (x_hi, w_lo) into the same accumulators, in the general stem kernel's order (bit-identical to it), the 16 channels stored as (hi | lo) halves
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void conv_stem7x7_thin_kernel(ConvK p) {
  a16_kernel_enter();
  using C = StemCfg<1>;
  constexpr int NPL = SPLIT ? 2 : 1;
  __shared__ __attribute__((aligned(16))) char s_in[NPL][C::IN_BYTES];
  __shared__ __attribute__((aligned(16))) char s_w[NPL][32 * C::WROW];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  const int strips_x = (p.tiles_x + STEM_NT - 1) / STEM_NT;
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int sxi = L % strips_x;
  L /= strips_x;
  const int tyi = L % p.tiles_y;
  const int b = L / p.tiles_y;
  const int oy0 = tyi * C::TH, iy0 = oy0 - 3;
  constexpr int PS = SPLIT ? 8 : 4;                       // bf16 elements per input pixel
  const bf16_t* in_b = p.in + (size_t)b * p.H * p.W * PS;
  for (int idx = tid; idx < 32 * 28; idx += 256) {        // weight rows 0..31 (channels >= n_valid are zero rows)
    const int row = idx / 28, part = idx - row * 28;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
      *reinterpret_cast<u32x4*>(s_w[pl] + row * C::WROW + part * 16) = *reinterpret_cast<const u32x4*>(p.w + (size_t)pl * 64 * 224 + (size_t)idx * 8);
  }
  const int a_off = ((wave * 2) * C::TWIN + lx + 2 * q) * 8;
  const int b_off = lx * C::WROW + q * 16;
  for (int it = 0; it < STEM_NT; ++it) {
    const int txi = sxi * STEM_NT + it;
    if (txi >= p.tiles_x) break;
    const int ox0 = txi * C::TW, ix0 = ox0 - 3;
    __syncthreads();                                      // previous tile's fragments have been read
#pragma unroll
    for (int j = 0; j < C::NI; ++j) {
      const int idx = tid + j * 256;
      if (idx < C::NP_IN) {
        const int iy = idx / C::HP, ip = idx - iy * C::HP;
        const int gy = iy0 + iy, gx = ix0 + 2 * ip;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          u32x2 v0 = {0u, 0u}, v1 = {0u, 0u};
          if ((unsigned)gy < (unsigned)p.H) {
            const bf16_t* rowp = in_b + (size_t)gy * p.W * PS + pl * 4;
            if ((unsigned)gx < (unsigned)p.W) v0 = *reinterpret_cast<const u32x2*>(rowp + (size_t)gx * PS);
            if ((unsigned)(gx + 1) < (unsigned)p.W) v1 = *reinterpret_cast<const u32x2*>(rowp + (size_t)(gx + 1) * PS);
          }
          u32x4 v = {v0.x, v0.y, v1.x, v1.y};
          *reinterpret_cast<u32x4*>(s_in[pl] + (iy * C::TWIN + 2 * ip) * 8) = v;
        }
      }
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    // weights as the A operand: D is [channel][pixel] -- a lane owns one pixel, its accumulators are 4-channel runs
#pragma unroll
    for (int pass = 0; pass < (SPLIT ? 3 : 1); ++pass) {
      const char* a_base = s_in[SPLIT && pass == 1 ? NPL - 1 : 0] + a_off;
      const char* b_base = s_w[SPLIT && pass == 2 ? NPL - 1 : 0] + b_off;
#pragma unroll
      for (int r = 0; r < 7; ++r) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(b_base + (r * 2 + h) * 32);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const char* ap = a_base + ((m + r) * C::TWIN + 4 * h) * 8;
            const u32x2 lo = *reinterpret_cast<const u32x2*>(ap), hi = *reinterpret_cast<const u32x2*>(ap + 8);
            const u32x4 av = {lo.x, lo.y, hi.x, hi.y};
            acc[m] = mfma_32x32x16_a16(b0, __builtin_bit_cast(bf16x8, av), acc[m]);
          }
        }
      }
    }
    const int ox = ox0 + lx;
    if (ox < p.Wo) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int oy = oy0 + wave * 2 + m;
        if (oy >= p.Ho) continue;
        bf16_t* op = p.out + (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.out_cstride;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int ch = 8 * rg + 4 * q;
          if (ch >= p.n_valid) continue;
          const f32x4 bs = *reinterpret_cast<const f32x4*>(p.bias + ch);
          const float v0 = fmaxf(acc[m][rg * 4 + 0] + bs.x, 0.f), v1 = fmaxf(acc[m][rg * 4 + 1] + bs.y, 0.f),
                      v2 = fmaxf(acc[m][rg * 4 + 2] + bs.z, 0.f), v3 = fmaxf(acc[m][rg * 4 + 3] + bs.w, 0.f);
          if (SPLIT) {
            const uint32_t h01 = pack_bf16x2(v0, v1), h23 = pack_bf16x2(v2, v3);
            *reinterpret_cast<u32x2*>(op + ch) = u32x2{h01, h23};
            *reinterpret_cast<u32x2*>(op + p.out_lo_off + ch) =
                u32x2{pack_bf16x2(v0 - bf16lo_f32(h01), v1 - bf16hi_f32(h01)), pack_bf16x2(v2 - bf16lo_f32(h23), v3 - bf16hi_f32(h23))};
          } else {
            const uint32_t h0 = f32_to_bf16(v0), h1 = f32_to_bf16(v1), h2 = f32_to_bf16(v2), h3 = f32_to_bf16(v3);
            *reinterpret_cast<u32x2*>(op + ch) = u32x2{h0 | (h1 << 16), h2 | (h3 << 16)};
          }
        }
      }
    }
  }
}

static void launch_half(ConvK& k, unsigned nblk, hipStream_t s) {
  using C = ConvCfg<3, 1, 0>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<3, 1, 0, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<3, 1, 0, 1>), dim3(nblk), dim3(256), C::SMEM, s, k);
}

// layers with a plain epilogue: the register-epilogue instances of the kernel (1x1, stride-2 3x3, 3x3 at both widths)
template <int KS, int STRIDE, int NHALF>
static void launch_direct(const ConvK& k, unsigned nblk, hipStream_t s) {
  using C = ConvCfg<KS, STRIDE, 0>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<KS, STRIDE, 0, NHALF, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<KS, STRIDE, 0, NHALF, true>), dim3(nblk), dim3(256), C::SMEM, s, k);
}

static bool conv1_wide() {      // PT_CONV1_WIDE=0: conv_igemm_kernel<1, 1> (one 32-channel chunk per stage) for every plain 1x1 layer (A/B switch, read per call)
  const char* ev = getenv("PT_CONV1_WIDE");
  return !(ev && ev[0] == '0');
}
template <int KG>
static void launch_wide1(const ConvK& k, unsigned nblk, hipStream_t s) {
  using C = Conv1WideCfg<KG>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_wide_kernel<KG>), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv1x1_wide_kernel<KG>), dim3(nblk), dim3(256), C::SMEM, s, k);
}

template <int KS, int STRIDE, int GEOM = 0>
static int launch_cfg(pt_engine* e, ConvK& k, hipStream_t s, double flop) {
  using C = ConvCfg<KS, STRIDE, GEOM>;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<KS, STRIDE, GEOM>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM));
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<KS, STRIDE, GEOM, 2, false, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM));
    attr_done = true;
  }
  k.tiles_x = (k.Wo + C::TW - 1) / C::TW;
  k.tiles_y = (k.Ho + C::TH - 1) / C::TH;
  k.n_tiles = k.N / 64;
  long long nblk = (long long)k.B * k.tiles_x * k.tiles_y * k.n_tiles;
  if (GEOM == 1 && k.blist) {      // pairs of live 32-column blocks: the worst case is every block of every image
    k.tiles_x = (int)(((long long)k.B * (k.Wo / 32) + 1) / 2);
    k.tiles_y = 1;
    nblk = (long long)k.tiles_x * k.n_tiles;
  }
  PT_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv grid out of range (%lld blocks)", nblk);
  if (k.ylimit) {      // mostly empty worst-case extents: 16 tiles per workgroup
    k.reps = 16;
    k.total_tiles = (int)nblk;
    k.n_group = 0;
    nblk = (nblk + 15) / 16;
  }
  char label[48];
  snprintf(label, sizeof(label), "conv%dx%d s%d %d->%d @%dx%d%s", KS, KS, STRIDE, k.Cin, k.N, k.Ho, k.Wo, k.head_w ? " +head" : (k.split == 2 ? " h2" : (k.split ? " x3" : "")));
  int lim_slot = -1;
  {
    PtProfScope prof(e, s, KS == 3 ? PT_PROF_CONV3X3 : PT_PROF_CONV1X1, flop, label);
    static int direct1 = -1;      // PT_CONV1_DIRECT=0: the staged epilogue for every launch of this kernel (A/B switch)
    if (direct1 < 0) { const char* ev = getenv("PT_CONV1_DIRECT"); direct1 = ev ? atoi(ev) : 1; }
    const bool plain = direct1 && GEOM == 0 && !k.split && !k.pool && !k.head_w && !k.argmax_part && !k.res_f32 && !k.shuffle_cout;
    const bool narrow = KS == 3 && STRIDE == 1 && GEOM == 0 && k.n_valid > 0 && k.n_valid <= 32 && k.N == 64 && !k.split && !k.pool &&
                        !k.head_w && !k.argmax_part;      // <= 32 real output channels: half-width variant
    if (narrow && plain)
      launch_direct<3, 1, 1>(k, (unsigned)nblk, s);
    else if (narrow)
      launch_half(k, (unsigned)nblk, s);
    else if (plain && KS == 1 && STRIDE == 1 && !k.ylimit && k.Cin % 128 == 0 && conv1_wide())
      launch_wide1<4>(k, (unsigned)nblk, s);
    else if (plain && KS == 1 && STRIDE == 1 && !k.ylimit && k.Cin % 64 == 0 && conv1_wide())
      launch_wide1<2>(k, (unsigned)nblk, s);
    else if (plain && KS == 1 && STRIDE == 1)
      launch_direct<1, 1, 2>(k, (unsigned)nblk, s);
    else if (plain && KS == 3 && STRIDE == 2)
      launch_direct<3, 2, 2>(k, (unsigned)nblk, s);
    else if (plain && KS == 1 && STRIDE == 2)
      launch_direct<1, 2, 2>(k, (unsigned)nblk, s);
    else if (plain && KS == 3 && STRIDE == 1)
      launch_direct<3, 1, 2>(k, (unsigned)nblk, s);
    else if (k.split == 2)
      hipLaunchKernelGGL((conv_igemm_kernel<KS, STRIDE, GEOM, 2, false, true>), dim3((unsigned)nblk), dim3(256), C::SMEM, s, k);
    else
      hipLaunchKernelGGL((conv_igemm_kernel<KS, STRIDE, GEOM>), dim3((unsigned)nblk), dim3(256), C::SMEM, s, k);
    if ((k.ylimit || k.xcols) && prof.idx >= 0 && e->prof.h_lims && e->prof.n_lims < PtProfile::MAX_LIMS) {
      // the launch covers the worst case and stops at a device-side row limit (or per-image column limits): remember
      // where the limit will land
      auto& pd = e->prof.pending[prof.idx];
      pd.lim_slot = lim_slot = e->prof.n_lims++;
      pd.rows = k.ylimit ? k.Ho : (k.xlimit_rows ? k.Ho * k.Wo : k.B * k.Wo);
    }
  }
  if (lim_slot >= 0)      // behind the launch (and outside its event pair): the limit the kernel saw, to pinned memory
    (void)hipMemcpyAsync(e->prof.h_lims + lim_slot, k.ylimit ? k.ylimit : k.xcols, sizeof(int), hipMemcpyDeviceToHost, s);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

static bool use_dma_kernel() {
  static int v = -1;
  if (v < 0) {
    const char* s = getenv("PT_CONV_DMA");
    v = s ? atoi(s) : 1;
  }
  return v != 0;
}

template <int NT, int NWV, int S = 1>
static int launch_dma16(pt_engine* e, ConvK& k, hipStream_t s, double flop) {
  using C = Dma16Cfg<NT, NWV, S>;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_dma16_kernel<NT, NWV, S>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM));
    attr_done = true;
  }
  if (!e->zero_page) {
    PT_HIP_CHECK(hipMalloc(&e->zero_page, 8192));
    PT_HIP_CHECK(hipMemset(e->zero_page, 0, 8192));
  }
  k.tiles_x = (k.Wo + C::TW - 1) / C::TW;
  k.tiles_y = (k.Ho + C::TH - 1) / C::TH;
  k.n_tiles = k.N / C::NW;
  const long long nblk = (long long)k.B * k.tiles_x * k.tiles_y * k.n_tiles;
  PT_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv grid out of range (%lld blocks)", nblk);
  char label[48];
  snprintf(label, sizeof(label), "conv3x3 %s%s %d->%d @%dx%d%s", S == 2 ? "s2 v3" : "v3", NWV == 4 ? "h" : "", k.Cin, k.N, k.Ho, k.Wo, k.split ? " x3" : "");
  PtProfScope prof(e, s, PT_PROF_CONV3X3, flop, label);
  hipLaunchKernelGGL((conv3x3_dma16_kernel<NT, NWV, S>), dim3((unsigned)nblk), dim3(C::NTHR), C::SMEM, s, k,
                     reinterpret_cast<const bf16_t*>(e->zero_page));
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// v4: the software-pipelined tap loop (conv3x3_pipe_kernel); same tiling arithmetic and labels as launch_dma16
template <int NT, int NWV, bool SKEW, int BR = 0, bool DIRECT = false, int S = 1, bool PHASE = false>
static int launch_pipe(pt_engine* e, ConvK& k, hipStream_t s, double flop) {
  using C = PipeCfg<NT, NWV, BR, S>;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pipe_kernel<NT, NWV, SKEW, BR, DIRECT, S, PHASE>), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM));
    attr_done = true;
  }
  k.n_tiles = k.N / C::NW;
  long long nblk;
  if (C::NBLK == 1) {
    k.tiles_x = (k.Wo + C::TW - 1) / C::TW;
    k.tiles_y = (k.Ho + C::TH - 1) / C::TH;
    nblk = (long long)k.B * k.tiles_x * k.tiles_y * k.n_tiles;
  } else if (k.blist) {      // NBLK live 32-column blocks per workgroup: the worst case is every block of every image
    k.tiles_x = (int)(((long long)k.B * (k.Wo / 32) + C::NBLK - 1) / C::NBLK);
    k.tiles_y = 1;
    nblk = (long long)k.tiles_x * k.n_tiles;
  } else {                   // NBLK consecutive column blocks of one image
    k.tiles_x = (k.Wo + C::NBLK * 32 - 1) / (C::NBLK * 32);
    k.tiles_y = 1;
    nblk = (long long)k.B * k.tiles_x * k.n_tiles;
  }
  PT_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv grid out of range (%lld blocks)", nblk);
  char label[48];
  if (C::NBLK == 1) snprintf(label, sizeof(label), "conv3x3 %sv4%s%s %d->%d @%dx%d%s", S == 2 ? "s2 " : "", NWV == 4 ? "h" : "", PHASE ? "p" : "", k.Cin, k.N, k.Ho, k.Wo, k.split ? " x3" : "");
  else snprintf(label, sizeof(label), "conv3x3 v4b %d->%d @%dx%d%s", k.Cin, k.N, k.Ho, k.Wo, k.split ? " x3" : "");
  int lim_slot = -1;
  {
    PtProfScope prof(e, s, PT_PROF_CONV3X3, flop, label);
    hipLaunchKernelGGL((conv3x3_pipe_kernel<NT, NWV, SKEW, BR, DIRECT, S, PHASE>), dim3((unsigned)nblk), dim3(C::NTHR), C::SMEM, s, k, reinterpret_cast<const bf16_t*>(e->zero_page));
    if (k.xcols && prof.idx >= 0 && e->prof.h_lims && e->prof.n_lims < PtProfile::MAX_LIMS) {
      // per-image column limits: the launch covers the worst case; remember where the limit will land (credited at read-out, like launch_cfg)
      auto& pd = e->prof.pending[prof.idx];
      pd.lim_slot = lim_slot = e->prof.n_lims++;
      pd.rows = k.B * k.Wo;
    }
  }
  if (lim_slot >= 0) (void)hipMemcpyAsync(e->prof.h_lims + lim_slot, k.xcols, sizeof(int), hipMemcpyDeviceToHost, s);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// the persistent form of launch_pipe<NT, NWV, true, 0, true, S> (conv3x3_pipe_persist_kernel): as many workgroups as fit the chip, a contiguous run of tiles each
template <int NT, int NWV, int S>
static int launch_pipe_persist(pt_engine* e, ConvK& k, hipStream_t s, double flop) {
  using C = PipeCfg<NT, NWV, 0, S>;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pipe_persist_kernel<NT, NWV, S>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C::BUF_BYTES));
    attr_done = true;
  }
  k.n_tiles = k.N / C::NW;
  k.tiles_x = (k.Wo + C::TW - 1) / C::TW;
  k.tiles_y = (k.Ho + C::TH - 1) / C::TH;
  const long long total = (long long)k.B * k.tiles_x * k.tiles_y * k.n_tiles;
  PT_REQUIRE(total > 0 && total < (1ll << 31), "conv grid out of range (%lld tiles)", total);
  k.total_tiles = (int)total;
  const int per_cu = 2 * C::BUF_BYTES <= 80 * 1024 ? 2 : 1;      // workgroups that fit a CU's LDS
  long long grid = (long long)e->num_cu * per_cu;
  if (const char* gv = getenv("PT_CONV_PERSIST_GRID")) {          // tests: fewer workgroups, longer runs
    const int gg = atoi(gv);
    if (gg > 0) grid = gg;
  }
  if (grid > total) grid = total;
  char label[48];
  snprintf(label, sizeof(label), "conv3x3 %sv5%s %d->%d @%dx%d", S == 2 ? "s2 " : "", NWV == 4 ? "h" : "", k.Cin, k.N, k.Ho, k.Wo);
  {
    PtProfScope prof(e, s, PT_PROF_CONV3X3, flop, label);
    hipLaunchKernelGGL((conv3x3_pipe_persist_kernel<NT, NWV, S>), dim3((unsigned)grid), dim3(C::NTHR), 2 * C::BUF_BYTES, s, k);
  }
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// 1x1 stride-1 plain layers with K >= 256, N % 128 == 0, M % 32 == 0 in the single-pass modes: the pipelined GEMM (gemm_pipe_kernel)
static int launch_gemm_pipe(pt_engine* e, ConvK& k, hipStream_t s, double flop) {
  using C = GemmPipeCfg;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM));
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM));
    attr_done = true;
  }
  const long long M = (long long)k.B * k.H * k.W, groups = M / 32;
  {
    const char* xv = getenv("PT_GEMM_XP");      // 0: 16-byte runs straight from the accumulators (A/B switch, read per call)
    k.xp_store = !(xv && xv[0] == '0') && !k.out_f32;
  }
  k.n_tiles = k.N / C::NW;
  k.tiles_x = (int)((groups + C::NG - 1) / C::NG);      // worst case with a list: every group live
  k.tiles_y = 1;
  const long long nblk = (long long)k.tiles_x * k.n_tiles;
  PT_REQUIRE(nblk > 0 && nblk < (1ll << 31), "gemm grid out of range (%lld blocks)", nblk);
  char label[48];
  if (k.argmax_part) snprintf(label, sizeof(label), "classifier gemm+argmax");      // (the label bench.py's by_class files the CTC classifier under)
  else snprintf(label, sizeof(label), "gemm %d->%d @%dx%d", k.Cin, k.N, k.Ho, k.Wo);
  int lim_slot = -1;
  {
    PtProfScope prof(e, s, PT_PROF_CONV1X1, flop, label);
    if (k.blist) hipLaunchKernelGGL((gemm_pipe_kernel<true>), dim3((unsigned)nblk), dim3(C::NTHR), C::SMEM, s, k);
    else hipLaunchKernelGGL((gemm_pipe_kernel<false>), dim3((unsigned)nblk), dim3(C::NTHR), C::SMEM, s, k);
    if (k.xcols && prof.idx >= 0 && e->prof.h_lims && e->prof.n_lims < PtProfile::MAX_LIMS) {      // ragged rows: credited at read-out, like launch_cfg
      auto& pd = e->prof.pending[prof.idx];
      pd.lim_slot = lim_slot = e->prof.n_lims++;
      pd.rows = k.xlimit_rows ? k.Ho * k.Wo : k.B * k.Wo;
    }
  }
  if (lim_slot >= 0) (void)hipMemcpyAsync(e->prof.h_lims + lim_slot, k.xcols, sizeof(int), hipMemcpyDeviceToHost, s);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// weight-stationary persistent kernel for plain 64 -> 64 layers (bf16 mode): one workgroup per CU walks total_tiles / grid tiles
static int launch_ws64(pt_engine* e, ConvK& k, hipStream_t s, double flop) {
  using C = Ws64Cfg;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ws64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM));
    attr_done = true;
  }
  if (!e->zero_page) {
    PT_HIP_CHECK(hipMalloc(&e->zero_page, 8192));
    PT_HIP_CHECK(hipMemset(e->zero_page, 0, 8192));
  }
  k.tiles_x = (k.Wo + C::TW - 1) / C::TW;
  k.tiles_y = (k.Ho + C::TH - 1) / C::TH;
  k.n_tiles = 1;
  const long long total = (long long)k.B * k.tiles_x * k.tiles_y;
  PT_REQUIRE(total > 0 && total < (1ll << 30), "conv grid out of range (%lld tiles)", total);
  k.total_tiles = (int)total;
  unsigned nblk = (unsigned)(total < e->num_cu ? total : e->num_cu);
  if (const char* gv = getenv("PT_CONV_WS64_GRID")) {      // tests: fewer workgroups, longer tile runs
    const int g = atoi(gv);
    if (g > 0 && (unsigned)g < nblk) nblk = (unsigned)g;
  }
  char label[48];
  snprintf(label, sizeof(label), "conv3x3 ws %d->%d @%dx%d", k.Cin, k.N, k.Ho, k.Wo);
  PtProfScope prof(e, s, PT_PROF_CONV3X3, flop, label);
  hipLaunchKernelGGL(conv3x3_ws64_kernel, dim3(nblk), dim3(C::NTHR), C::SMEM, s, k, reinterpret_cast<const bf16_t*>(e->zero_page));
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// PT_CONV_VARIANT: 0 = v1 only, 1 = micro-benchmark rule, 2 = v2 wherever it applies, 3 = v3 wherever it applies (default).
// History: in the DB-ResNet18 graph at 8-page micro-batches v1-only measured fastest (det-only, no post: 3836 pages/s vs
// 3764 with rule 1 and 3719 with v2) and was the default for most of round 1.  With 32-page det and 80-table Lore
// micro-batches the 16-channel-slice DMA kernel wins where it applies (tools/ab_env.sh, two alternating runs each:
// det-only 3503 / 3607 -> 3685 / 3647 pages/s, TSR-only 833 / 842 -> 845 / 859, four stages 354 / 358 -> 361 / 361).
static int conv_variant() {
  static int v = -1;
  if (v < 0) {
    const char* s = getenv("PT_CONV_VARIANT");
    v = s ? atoi(s) : 3;
  }
  return v;
}

int pt_launch_conv(pt_engine* e, const ConvDesc& d, hipStream_t s) {
  PT_REQUIRE(d.in && d.w && d.bias && (d.out || d.out_f32 || d.head_w || d.argmax_part), "conv: null pointer");
  PT_REQUIRE(d.Cin % 32 == 0 && d.Cin > 0, "conv: Cin=%d must be a positive multiple of 32", d.Cin);
  PT_REQUIRE(d.N % 64 == 0 && d.N > 0, "conv: N=%d must be a positive multiple of 64", d.N);
  PT_REQUIRE((d.ks == 1 || d.ks == 3) && (d.stride == 1 || d.stride == 2), "conv: ks=%d stride=%d unsupported", d.ks,
             d.stride);
  PT_REQUIRE(!(d.shuffle_cout && (d.rep != 1 || d.shuffle_cout % 64 != 0 || d.N != 4 * d.shuffle_cout)),
             "conv: bad pixel-shuffle configuration");
  PT_REQUIRE(d.rep >= 1 && d.out_cstride % 8 == 0 && d.out_coff % 8 == 0, "conv: bad output layout");
  ConvK k;
  memset(&k, 0, sizeof(k));
  k.n_valid = d.n_valid; k.out_f32 = d.out_f32; k.res_f32 = d.res_f32;
  PT_REQUIRE(!d.res_f32 || d.out_f32, "conv: fp32 residual needs the fp32 output path");
  PT_REQUIRE(d.n_valid % 8 == 0 && d.n_valid <= d.N, "conv: n_valid=%d must be a multiple of 8 and <= N", d.n_valid);
  PT_REQUIRE(!(d.out_f32 && (d.rep != 1 || d.shuffle_cout)), "conv: fp32 output only on the plain store path");
  k.in = d.in; k.w = d.w; k.bias = d.bias; k.out = d.out; k.res = d.res;
  k.nseg = d.nseg;
  if (d.nseg > 1) {
    PT_REQUIRE(d.nseg <= 4 && d.ks == 1 && d.stride == 1, "conv: K over several tensors needs a 1x1 stride-1 layer and <= 4 segments");
    int sum = 0;
    for (int i = 0; i < d.nseg; ++i) {
      PT_REQUIRE(d.seg_c[i] > 0 && d.seg_c[i] % 32 == 0 && (i == 0 || d.in_more[i - 1]), "conv: bad K segment %d", i);
      sum += d.seg_c[i];
    }
    PT_REQUIRE(sum == d.Cin, "conv: K segments sum to %d channels, Cin is %d", sum, d.Cin);
    k.in1 = d.in_more[0]; k.in2 = d.in_more[1]; k.in3 = d.in_more[2];
    k.segc0 = d.seg_c[0]; k.segc1 = d.seg_c[1]; k.segc2 = d.seg_c[2]; k.segc3 = d.seg_c[3];
  }
  k.B = d.B; k.H = d.H; k.W = d.W; k.Cin = d.Cin; k.N = d.N;
  const int pad = d.ks / 2;
  k.Ho = (d.H + 2 * pad - d.ks) / d.stride + 1;
  k.Wo = (d.W + 2 * pad - d.ks) / d.stride + 1;
  k.out_cstride = d.out_cstride; k.out_coff = d.out_coff; k.rep = d.rep; k.shuffle_cout = d.shuffle_cout;
  k.res_mode = d.res ? d.res_mode : 0; k.relu = d.relu; k.slope = d.slope; k.ylimit = d.ylimit; k.pool = d.pool;
  k.xlimit = d.xlimit; k.xlimit_rows = d.xlimit_rows; k.xcols = (d.xlimit || d.xlimit_rows) ? d.xlimit_cols : nullptr;
  {
    const char* bl = getenv("PT_CONV_BLOCK_LIST");      // PT_CONV_BLOCK_LIST=0: one image's 64 columns per workgroup (A/B switch, read per call)
    k.blist = (d.xlimit && d.block_list && d.ks == 3 && d.stride == 1 && k.Ho <= 4 && k.Wo > 32 && k.Wo % 32 == 0 && !(bl && bl[0] == '0')) ? d.block_list : nullptr;
  }
  PT_REQUIRE(!d.xlimit_rows || (d.ks == 1 && d.stride == 1 && !d.ylimit && !d.xlimit), "conv: row-wise column limits need a 1x1 stride-1 layer");
  PT_REQUIRE(!d.xlimit || (d.ks == 3 && d.stride == 1 && !d.ylimit), "conv: column limits need a 3x3 stride-1 layer");
  PT_REQUIRE(!d.pool || (d.ks == 3 && d.stride == 1 && !d.res && !d.shuffle_cout && d.rep == 1 && !d.out_f32 && !d.argmax_part && !d.head_w && !d.n_valid && d.relu <= 1 && k.Ho % 2 == 0 && (d.pool != 1 || k.Wo % 2 == 0)),
             "conv: fused pooling needs a plain 3x3 stride-1 layer with even output size");
  PT_REQUIRE(d.relu != 3 || d.slope, "conv: PReLU needs the slope tensor");
  k.split = d.split; k.out_lo_off = d.out_lo_off;
  for (int i = 0; i < 8; ++i) k.tap_mask[i] = d.tap_mask[i];
  {
    // the layer asks for it (ConvDesc.xp_store: the detector's graph, +0.9 % det-only; the layout net's many small hardswish GEMMs lose
    // 1.9 % to the extra barrier, the table nets are indifferent); PT_CONV_XP=0 / 1 forces it off / on everywhere (A/B switch, read per call)
    const char* xv = getenv("PT_CONV_XP");
    k.xp_store = (xv ? atoi(xv) : d.xp_store) && !d.out_f32;
  }
  k.head_w = d.head_w; k.head_b = d.head_b; k.head_prob = d.head_prob; k.head_logits = d.head_logits;
  k.argmax_part = d.argmax_part;
  if (d.head_w) PT_REQUIRE(d.shuffle_cout == 64 && d.head_b && (d.head_prob || d.head_logits), "conv: bad fused-head configuration");
  if (k.res_mode == 2) PT_REQUIRE(k.Ho % 2 == 0 && k.Wo % 2 == 0, "conv: half-res residual needs even output size");
  // algorithmic FLOP (not x3 in split mode): the layer's real output channels, not the padded GEMM width; for a transposed
  // conv run as a pixel-shuffle GEMM N = 4 * Cout IS the work.  Row-limited launches are credited at read-out with the rows
  // the device limit let through (launch_cfg)
  const int n_alg = d.alg_n ? d.alg_n : (d.n_valid ? d.n_valid : d.N);
  const double flop = 2.0 * k.B * k.Ho * k.Wo * (double)n_alg * d.Cin * d.ks * d.ks * d.alg_scale;
  {
    // algorithmic bytes of the launch (roofline.by_class of bench.py): every input pixel the layer samples once, every stored output once,
    // the residual once, the weights once; (hi | lo) tensors carry both halves
    const double mact = d.split ? 2.0 : 1.0, osz = d.out_f32 ? 4.0 : 2.0 * mact;
    const double in_px = (d.ks == 1 && d.stride == 2) ? (double)k.B * k.Ho * k.Wo : (double)d.B * d.H * d.W;
    const double nstore = d.shuffle_cout ? d.N : (d.n_valid ? d.n_valid : d.N);
    double by = in_px * d.Cin * 2.0 * mact + (double)k.B * k.Ho * k.Wo * d.rep * d.rep * nstore * osz * d.alg_scale;
    if (d.res || d.res_f32) by += (double)k.B * k.Ho * k.Wo * nstore * (d.res_f32 ? 4.0 : 2.0 * mact) / (d.res_mode == 2 ? 4.0 : 1.0);
    by += (double)d.N * d.Cin * d.ks * d.ks * 2.0 * (d.split == 2 ? 2 : (d.split ? 3 : 1));
    e->prof.next_bytes = by;
  }
  // stride-2 3x3 layers with 128-wide output blocks (ResNet-18 / DLA-34 down-sampling convs): the 16-channel-slice DMA kernel on an 8 x 32 x 128 tile
  // (PT_CONV_S2_DMA=0: the register-staged 4 x 32 x 64 tile, A/B switch)
  if (d.ks == 3 && d.stride == 2 && !d.split && d.N % 128 == 0 && d.Cin % 32 == 0 && d.Cin >= 64 && k.Ho >= 8 && !d.head_w && !d.argmax_part && !d.n_valid &&
      !d.out_f32 && !d.res_f32 && !d.res && d.relu < 2 && !d.ylimit && !d.xlimit && !d.xlimit_rows && !d.pool && d.rep == 1 && !d.shuffle_cout && use_dma_kernel()) {
    static int s2 = -1;
    if (s2 < 0) { const char* ev = getenv("PT_CONV_S2_DMA"); s2 = ev ? atoi(ev) : 1; }
    bool masked = false;
    for (int i = 0; i < 8; ++i) masked = masked || d.tap_mask[i] != 0;
    const char* pv2 = getenv("PT_CONV_PIPE");      // 0: the v3 stride-2 tile (A/B switch, read per call)
    if (s2 && !masked && !(pv2 && pv2[0] == '0')) return launch_pipe<2, 8, true, 0, true, 2>(e, k, s, flop);
    if (s2 && !masked) return launch_dma16<2, 8, 2>(e, k, s, flop);
  }
  if (d.ks == 3 && d.stride == 1 && !d.head_w && !d.argmax_part && !d.n_valid && !d.out_f32 && !d.res_f32 && d.relu < 2 && !d.ylimit && !d.xlimit && !d.pool && d.split != 2 && use_dma_kernel()) {
    // steady-state A/B on MI355X (tools/ab3.sh, round 1): the 16-channel-slice DMA kernel (v3) wins on >= 120-row maps
    // with K >= 128 channels, the 32-channel-slice DMA kernel (v2) on 60..119-row maps, the register-staged kernel
    // (v1) on short-K layers and on small maps, where the big DMA tiles leave CUs idle
    const int cv = conv_variant();
    // plain 64 -> 64 layers with enough tiles to give every CU a run of them: the weight-stationary persistent kernel
    // (PT_CONV_WS64=0: the v3 tiles, A/B switch)
    const char* ws_ev = getenv("PT_CONV_WS64");      // read per call: 2 forces the kernel for any tile count (tests)
    const int ws64 = ws_ev ? atoi(ws_ev) : 1;
    bool masked = false;
    for (int i = 0; i < 8; ++i) masked = masked || d.tap_mask[i] != 0;
    if (ws64 && cv == 3 && !d.split && d.Cin == 64 && d.N == 64 && d.rep == 1 && !d.shuffle_cout && d.res_mode <= 1 && !masked &&
        (ws64 == 2 || (long long)k.B * ((k.Ho + 15) / 16) * ((k.Wo + 31) / 32) >= 4ll * e->num_cu))
      return launch_ws64(e, k, s, flop);
    const bool v3ok = (d.N % 128 == 0) ? k.Ho >= 12 : k.Ho >= 24;
    const bool wide = d.Cin >= 128;
    int pick = 0;
    if (cv == 3 && v3ok) pick = 3;
    else if (cv == 1 && wide && k.Ho >= 120) pick = 3;
    // 4-wave workgroups (16x32 pixels x 64 channels, 74 KB of LDS: two per CU) where the 8-wave tile is mostly prologue and
    // epilogue: short K (Cin <= 128: +3..8 % on 64->64 @240^2 / @256^2, 128->128 @120^2 / @128^2, 64->256 @256^2), and grids
    // that leave the 8-wave tiling with a ragged last round (512->512 @32^2 x 44: +12 %); K >= 2304 layers whose grid is whole
    // rounds stay on the 8-wave tiles (512->512 @30^2 x 32: -3 % as 4-wave).  PT_CONV_HALF = 0 / 1 forces one of them.
    static int half = -2;
    if (half == -2) { const char* ev = getenv("PT_CONV_HALF"); half = ev ? atoi(ev) : -1; }
    bool use_half = half == 1;
    if (half < 0 && pick == 3) {
      const bool nt2 = d.N % 128 == 0;
      const long long full = (long long)k.B * ((k.Ho + (nt2 ? 15 : 31)) / (nt2 ? 16 : 32)) * ((k.Wo + 31) / 32) * (d.N / (nt2 ? 128 : 64));
      use_half = d.Cin <= 128 || (full < 2ll * e->num_cu && full % e->num_cu != 0);
    }
    // v4 (software-pipelined tap loop) for the unmasked layers; PT_CONV_PIPE=0: v3 everywhere (A/B switch, read per call)
    const char* pv = getenv("PT_CONV_PIPE");
    // the phase convolutions (db_model.hip phase_masks: four 64-channel tiles, taps (dy .. dy + 1) x (dx .. dx + 1)) have their own v4 variant
    bool phase = masked && d.N == 256;
    for (int i = 0; i < 8 && phase; ++i) {
      unsigned mk = 0;
      if (i < 4)
        for (int r = i >> 1; r < (i >> 1) + 2; ++r)
          for (int q = i & 1; q < (i & 1) + 2; ++q) mk |= 1u << (r * 3 + q);
      phase = d.tap_mask[i] == mk;
    }
    if (phase && cv == 3 && v3ok && !(pv && atoi(pv) < 3)) return launch_pipe<2, 8, true, 0, false, 1, true>(e, k, s, flop);
    const int pipe = masked ? 0 : (pv ? atoi(pv) : 4);      // 1: barrier behind tap 8, 2: in front of it (SKEW), 3: + register epilogue on plain layers, 4: + persistent workgroups
    if (pick == 3 && pipe == 1) {
      if (use_half) return launch_pipe<1, 4, false>(e, k, s, flop);
      return d.N % 128 == 0 ? launch_pipe<2, 8, false>(e, k, s, flop) : launch_pipe<1, 8, false>(e, k, s, flop);
    }
    if (pick == 3 && pipe >= 2) {
      // register epilogue for the plain layers (PT_CONV_PIPE=2: the staged fp32 epilogue everywhere, A/B switch)
      const bool direct = pipe >= 3 && !k.split && !k.pool && !k.shuffle_cout;
      // 4: the persistent form for the 8-wave tiles where a workgroup gets at least two tiles (interleaved A/B, tools/conv_ab.py: 256 -> 256 @64^2 1305 -> 1327
      // TF/s, @60^2 1174 -> 1198, 512 -> 512 @32^2 1379 -> 1404, @30^2 1267 -> 1275; the 4-wave tile LOSES 14 % as a persistent kernel -- 128 -> 128 @128^2
      // 1134 -> 975: two of its workgroups per CU already cover each other's prologues, and the run's bookkeeping costs it registers it does not have)
      const bool nt2 = d.N % 128 == 0;
      const long long tiles = (long long)k.B * ((k.Ho + (nt2 ? 15 : 31)) / (nt2 ? 16 : 32)) * ((k.Wo + 31) / 32) * (d.N / (nt2 ? 128 : 64));
      // (a four-slice layer without a residual is better off on the persistent 8-wave tile than on the 4-wave one: 64 -> 256 @256^2 900 -> 955 TF/s; with
      // eight slices the two are level -- 128 -> 128 @128^2 1101 / 1120 -- and with a residual the 4-wave tile wins, 922 against 862)
      if (use_half && half < 0 && direct && pipe >= 4 && d.Cin == 64 && nt2 && !d.res && tiles >= 2ll * e->num_cu) use_half = false;
      if (direct && pipe >= 4 && (!use_half || pipe >= 5) && !k.xlimit && !k.xlimit_rows && !k.blist) {
        if (tiles >= 2ll * e->num_cu || pipe >= 5)      // (5: whatever the tile count -- tests)
          return nt2 ? launch_pipe_persist<2, 8, 1>(e, k, s, flop) : launch_pipe_persist<1, 8, 1>(e, k, s, flop);
      }
      if (direct) {
        if (use_half) return launch_pipe<1, 4, true, 0, true>(e, k, s, flop);
        return d.N % 128 == 0 ? launch_pipe<2, 8, true, 0, true>(e, k, s, flop) : launch_pipe<1, 8, true, 0, true>(e, k, s, flop);
      }
      if (use_half) return launch_pipe<1, 4, true>(e, k, s, flop);
      return d.N % 128 == 0 ? launch_pipe<2, 8, true>(e, k, s, flop) : launch_pipe<1, 8, true>(e, k, s, flop);
    }
    if (pick == 3 && use_half) return launch_dma16<1, 4>(e, k, s, flop);
    if (pick == 3) return d.N % 128 == 0 ? launch_dma16<2, 8>(e, k, s, flop) : launch_dma16<1, 8>(e, k, s, flop);
  }
  // maps that are exactly 4 or 8 rows high with K >= 1152 and 128-wide output blocks (the CRNN conv2.* / conv3.* layers): the pipelined DMA kernel on
  // tiles of four 4 x 32 / two 8 x 32 blocks -- from the compacted list of live blocks when the layer has one (PT_CONV_PIPE_BLOCKS=0: the
  // register-staged kernels, A/B switch read per call)
  if (d.ks == 3 && d.stride == 1 && (k.Ho == 4 || k.Ho == 8) && k.H == k.Ho && k.Wo % 32 == 0 && k.Wo > 32 && d.N % 128 == 0 && d.Cin >= 128 && !d.head_w &&
      !d.argmax_part && !d.n_valid && !d.out_f32 && !d.res_f32 && !d.res && d.relu < 2 && !d.ylimit && !d.xlimit_rows && d.rep == 1 && !d.shuffle_cout &&
      d.split != 2 && use_dma_kernel()) {
    const char* pb = getenv("PT_CONV_PIPE_BLOCKS");
    const char* pv = getenv("PT_CONV_PIPE");
    bool masked = false;
    for (int i = 0; i < 8; ++i) masked = masked || d.tap_mask[i] != 0;
    if (!masked && !(pb && pb[0] == '0') && !(pv && pv[0] == '0')) {
      if (d.xlimit && d.block_list) k.blist = d.block_list;      // (8-row maps too: the GEOM 1 rule above only takes lists for <= 4 rows)
      if (!k.split && !k.pool && !(pv && atoi(pv) < 3))           // plain layer: register epilogue
        return k.Ho == 4 ? launch_pipe<2, 8, true, 4, true>(e, k, s, flop) : launch_pipe<2, 8, true, 8, true>(e, k, s, flop);
      return k.Ho == 4 ? launch_pipe<2, 8, true, 4>(e, k, s, flop) : launch_pipe<2, 8, true, 8>(e, k, s, flop);
    }
  }
  if (d.ks == 3 && d.stride == 1 && k.Ho <= 4 && k.Wo > 32) return launch_cfg<3, 1, 1>(e, k, s, flop);
  if (d.ks == 3 && d.stride == 1) return launch_cfg<3, 1>(e, k, s, flop);
  if (d.ks == 3 && d.stride == 2) return launch_cfg<3, 2>(e, k, s, flop);
  // (K >= 512: at K = 256 a tile is four slices -- mostly prologue and epilogue of a workgroup that has the CU to itself -- and the 2048-wide
  // projection measured 1.65 ms against the streaming row GEMM's 1.42; K = 512 ... 1024: 1.43 -> 1.35, 0.70 -> 0.60, 0.42 -> 0.34, 1.17 -> 0.63 ms)
  if (d.ks == 1 && d.stride == 1 && !d.split && d.nseg <= 1 && d.Cin >= 512 && d.Cin % 64 == 0 && d.N % 128 == 0 && ((long long)d.B * d.H * d.W) % 32 == 0 &&
      !d.head_w && (!d.argmax_part || (!d.res && !d.relu)) && !d.res_f32 && !d.shuffle_cout && !d.pool && !d.ylimit && !d.xlimit && d.rep == 1 && (!d.res || d.res_mode == 1) &&
      (!d.xlimit_rows || (d.block_list && d.W % 32 == 0)) && use_dma_kernel()) {
    const char* gv = getenv("PT_GEMM_PIPE");      // 0: the 4 x 32 x 64 tile kernels (A/B switch, read per call)
    if (!(gv && gv[0] == '0')) {
      k.blist = d.xlimit_rows ? d.block_list : nullptr;
      return launch_gemm_pipe(e, k, s, flop);
    }
  }
  if (d.ks == 1) {
    // column tiles per group: ~2 MB of weights (half of an XCD's L2); PT_N_GROUP overrides (0 = off)
    static int ng_env = -2;
    if (ng_env == -2) { const char* s_ = getenv("PT_N_GROUP"); ng_env = s_ ? atoi(s_) : -1; }
    const long long tile_bytes = 64ll * d.Cin * 2 * (d.split == 2 ? 2 : (d.split ? 3 : 1));
    int g = ng_env >= 0 ? ng_env : (int)((2ll << 20) / tile_bytes);
    if (g < 1) g = ng_env == 0 ? 0 : 1;
    k.n_group = g;
  }
  if (d.ks == 1 && d.stride == 1) return launch_cfg<1, 1>(e, k, s, flop);
  return launch_cfg<1, 2>(e, k, s, flop);
}

// CRNN conv0 + pool + conv1 + pool in one launch (crnn_conv01_kernel): gray [n][32][640] -> p1 [n][8][160][128]; xlimit: conv1's per-line column limits
int pt_launch_crnn_conv01(pt_engine* e, const bf16_t* gray, int n, const float* w64x9, const float* b0, const bf16_t* w1, const float* b1, bf16_t* p1,
                          const int* xlimit, const int* xlimit_cols, hipStream_t s) {
  using C = Conv01Cfg;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&crnn_conv01_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM));
    attr_done = true;
  }
  ConvK k;
  memset(&k, 0, sizeof(k));
  k.in = gray; k.head_w = w64x9; k.head_b = b0; k.w = w1; k.bias = b1; k.out = p1;
  k.B = n; k.H = 16; k.W = 320; k.Ho = 16; k.Wo = 320; k.Cin = 64; k.N = 128;
  k.out_cstride = 128; k.out_coff = 0; k.rep = 1; k.relu = 1; k.pool = 1;
  k.xlimit = xlimit; k.xcols = xlimit ? xlimit_cols : nullptr;
  k.tiles_x = 10; k.tiles_y = 1; k.n_tiles = 1;
  const double flop = 2.0 * n * 16 * 320 * 128.0 * 64 * 9;
  e->prof.next_bytes = 2.0 * n * (32.0 * 640 + 8.0 * 160 * 128);
  int lim_slot = -1;
  {
    PtProfScope prof(e, s, PT_PROF_CONV3X3, flop, "conv0+pool+conv1+pool 1->64->128 @16x320");
    hipLaunchKernelGGL(crnn_conv01_kernel, dim3((unsigned)(n * 10)), dim3(C::NTHR), C::SMEM, s, k);
    if (k.xcols && prof.idx >= 0 && e->prof.h_lims && e->prof.n_lims < PtProfile::MAX_LIMS) {
      auto& pd = e->prof.pending[prof.idx];
      pd.lim_slot = lim_slot = e->prof.n_lims++;
      pd.rows = k.B * k.Wo;
    }
  }
  if (lim_slot >= 0) (void)hipMemcpyAsync(e->prof.h_lims + lim_slot, k.xcols, sizeof(int), hipMemcpyDeviceToHost, s);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// depthwise k x k (stride 1 / 2) + pointwise 1x1 in one launch (dwpw_kernel); returns PT_ERR_INVALID when the pair is outside the kernel's
// shapes (the caller then runs the two launches).  pw: the pointwise layer's ConvDesc on the depthwise OUTPUT map (its `in` is ignored).
int pt_launch_dwpw(pt_engine* e, const bf16_t* in, int B, int H, int W, int C, const float* dw_w, const float* dw_b, int k, int stride, int dw_act,
                   const ConvDesc& pw, hipStream_t s) {
  if (!(k == 3 || k == 5) || stride != 1 || C % 32 != 0 || pw.N % 64 != 0 || pw.N > 256 || pw.split || pw.res || pw.rep != 1 ||
      pw.shuffle_cout || pw.out_f32 || pw.relu > 2 || dw_act > 2 || pw.Cin != C)
    return PT_ERR_INVALID;
  const int pad = k / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (pw.H != Ho || pw.W != Wo || pw.B != B) return PT_ERR_INVALID;
  DwPwK p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.dw_w = dw_w; p.dw_b = dw_b; p.dw_act = dw_act;
  p.B = B; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo;
  p.tiles_x = (Wo + 31) / 32; p.tiles_y = (Ho + 3) / 4;
  ConvK& q = p.pw;
  q.w = pw.w; q.bias = pw.bias; q.out = pw.out; q.B = B; q.H = Ho; q.W = Wo; q.Ho = Ho; q.Wo = Wo; q.Cin = C; q.N = pw.N;
  q.out_cstride = pw.out_cstride; q.out_coff = pw.out_coff; q.rep = 1; q.relu = pw.relu; q.n_valid = pw.n_valid;
  const long long nblk = (long long)B * p.tiles_x * p.tiles_y;
  PT_REQUIRE(nblk > 0 && nblk < (1ll << 31), "dwpw grid out of range");
  char label[48];
  snprintf(label, sizeof(label), "dw%d s%d + pw %d->%d @%dx%d", k, stride, C, pw.n_valid ? pw.n_valid : pw.N, Ho, Wo);
  e->prof.next_bytes = 2.0 * B * ((double)H * W * C + (double)Ho * Wo * (pw.n_valid ? pw.n_valid : pw.N));
  PtProfScope prof(e, s, PT_PROF_CONV1X1, 2.0 * B * Ho * Wo * (double)C * (pw.n_valid ? pw.n_valid : pw.N), label);
#define PT_DWPW(KK, NN) hipLaunchKernelGGL((dwpw_kernel<KK, NN>), dim3((unsigned)nblk), dim3(256), 0, s, p)
#define PT_DWPW_N(KK) do { if (pw.N == 64) PT_DWPW(KK, 2); else if (pw.N == 128) PT_DWPW(KK, 4); else if (pw.N == 192) PT_DWPW(KK, 6); else PT_DWPW(KK, 8); } while (0)
  if (k == 3) PT_DWPW_N(3);
  else PT_DWPW_N(5);
#undef PT_DWPW_N
#undef PT_DWPW
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

template <int S, int NH>
static int launch_stem(pt_engine* e, ConvK& k, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem7x7_kernel<S, NH>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, StemCfg<S>::SMEM));
    attr_done = true;
  }
  k.tiles_x = (k.Wo + 31) / 32; k.tiles_y = (k.Ho + 7) / 8; k.n_tiles = 1;
  const long long nblk = (long long)k.B * k.tiles_x * k.tiles_y;
  PT_REQUIRE(nblk > 0 && nblk < (1ll << 31), "stem grid out of range");
  PtProfScope prof(e, s, PT_PROF_STEM, 2.0 * k.B * k.Ho * k.Wo * 64.0 * 147.0, S == 2 ? "stem7x7 s2" : "stem7x7 s1");
  hipLaunchKernelGGL((conv_stem7x7_kernel<S, NH>), dim3((unsigned)nblk), dim3(256), StemCfg<S>::SMEM, s, k);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ResNet-18 stem + MaxPool2d(3, 2, 1) fused (bf16 mode): in NHWC4 [B, H, W, 4], out [B, H/4, W/4, 64]
int pt_launch_stem7x7_pool(pt_engine* e, const bf16_t* in, int B, int H, int W, const bf16_t* w, const float* bias, bf16_t* out,
                           hipStream_t s) {
  PT_REQUIRE(in && w && bias && out, "stem+pool: null pointer");
  PT_REQUIRE(H % 4 == 0 && W % 4 == 0 && H > 0 && W > 0, "stem+pool: H, W must be multiples of 4");
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem7x7_pool_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, StemPoolCfg::SMEM));
    attr_done = true;
  }
  ConvK k;
  memset(&k, 0, sizeof(k));
  k.in = in; k.w = w; k.bias = bias; k.out = out;
  k.B = B; k.H = H; k.W = W; k.Cin = 4; k.N = 64;
  k.Ho = H / 2; k.Wo = W / 2;
  k.tiles_x = (W / 4 + StemPoolCfg::PW - 1) / StemPoolCfg::PW;
  k.tiles_y = (H / 4 + StemPoolCfg::PH - 1) / StemPoolCfg::PH;
  const long long nblk = (long long)B * k.tiles_x * k.tiles_y;
  PT_REQUIRE(nblk > 0 && nblk < (1ll << 31), "stem+pool grid out of range");
  // FLOP of the layer (the re-computed halo is not algorithmic work)
  PtProfScope prof(e, s, PT_PROF_STEM, 2.0 * B * k.Ho * k.Wo * 64.0 * 147.0, "stem7x7 s2 + maxpool");
  const char* ws_ev = getenv("PT_STEM_POOL_WS");      // 0: one tile per workgroup (A/B switch, read per call)
  if (!(ws_ev && atoi(ws_ev) == 0)) {
    static bool ws_attr = false;
    if (!ws_attr) {
      PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem7x7_pool_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, StemPoolCfg::SMEM));
      ws_attr = true;
    }
    k.total_tiles = (int)nblk;
    unsigned g = (unsigned)(nblk < e->num_cu ? nblk : e->num_cu);
    if (const char* gv = getenv("PT_STEM_POOL_WS_GRID")) {      // tests: fewer workgroups, longer tile runs
      const int gg = atoi(gv);
      if (gg > 0 && (unsigned)gg < g) g = (unsigned)gg;
    }
    hipLaunchKernelGGL(conv_stem7x7_pool_ws_kernel, dim3(g), dim3(StemPoolWsCfg::NTHR), StemPoolCfg::SMEM, s, k);
    PT_HIP_CHECK(hipGetLastError());
    return PT_OK;
  }
  hipLaunchKernelGGL(conv_stem7x7_pool_kernel, dim3((unsigned)nblk), dim3(256), StemPoolCfg::SMEM, s, k);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_launch_stem7x7(pt_engine* e, const bf16_t* in, int B, int H, int W, const bf16_t* w, const float* bias,
                      bf16_t* out, int split, hipStream_t s, int stride, int n_valid) {
  PT_REQUIRE(in && w && bias && out, "stem: null pointer");
  PT_REQUIRE(stride == 1 || stride == 2, "stem: stride %d unsupported", stride);
  PT_REQUIRE(stride == 1 || (H % 2 == 0 && W % 2 == 0), "stem: H, W must be even");
  PT_REQUIRE(n_valid % 8 == 0 && n_valid >= 0 && n_valid <= 64, "stem: bad n_valid");
  const int nv = n_valid ? n_valid : 64;
  ConvK k;
  memset(&k, 0, sizeof(k));
  k.in = in; k.w = w; k.bias = bias; k.out = out; k.res = nullptr;
  k.B = B; k.H = H; k.W = W; k.Cin = 4; k.N = 64;
  k.Ho = H / stride; k.Wo = W / stride;
  k.out_cstride = split ? 2 * nv : nv; k.out_coff = 0; k.rep = 1; k.shuffle_cout = 0; k.res_mode = 0; k.relu = 1;
  k.split = split; k.out_lo_off = nv; k.n_valid = n_valid;
  if (stride == 2) return launch_stem<2, 2>(e, k, s);
  static int thin = -1;          // PT_STEM_THIN=0: the general kernel for the thin stride-1 stem too (A/B switch)
  if (thin < 0) {
    const char* ev = getenv("PT_STEM_THIN");
    thin = ev ? atoi(ev) : 1;
  }
  static int thin_x3 = -1;       // PT_STEM_THIN_X3=0: the general kernel for the hi/lo mode's thin stem (A/B switch)
  if (thin_x3 < 0) {
    const char* ev = getenv("PT_STEM_THIN_X3");
    thin_x3 = ev ? atoi(ev) : 1;
  }
  if (thin && (!split || (split == 1 && thin_x3)) && n_valid && n_valid <= 32) {
    k.tiles_x = (k.Wo + 31) / 32; k.tiles_y = (k.Ho + 7) / 8; k.n_tiles = 1;
    const long long nblk = (long long)k.B * ((k.tiles_x + STEM_NT - 1) / STEM_NT) * k.tiles_y;
    PT_REQUIRE(nblk > 0 && nblk < (1ll << 31), "stem grid out of range");
    PtProfScope prof(e, s, PT_PROF_STEM, 2.0 * k.B * k.Ho * k.Wo * 64.0 * 147.0, split ? "stem7x7 s1 thin x3" : "stem7x7 s1 thin");
    if (split)
      hipLaunchKernelGGL(conv_stem7x7_thin_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, s, k);
    else
      hipLaunchKernelGGL(conv_stem7x7_thin_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, s, k);
    PT_HIP_CHECK(hipGetLastError());
    return PT_OK;
  }
  return (n_valid && n_valid <= 32) ? launch_stem<1, 1>(e, k, s) : launch_stem<1, 2>(e, k, s);
}

}  // namespace PT_FMT_NS
