// lore_kernels.hip -- bandwidth-type kernels of the Lore table-structure detector (DLA-34 + DCN).
//
//   dcn_im2col_kernel     modulated deformable sampling of torchvision.ops.deform_conv2d as called by
//                         lore/dcnv2.py:71-86: bilinear rule and zero fill of DCNv2_latest/src/cpu/
//                         dcn_v2_im2col_cpu.cpp:26-55,160-185, value * sigmoid(mask).  Writes the sampled columns
//                         [pixel][tap * C + c]; the deformable convolution itself is then a 1x1 GEMM on the MFMA kernel.
//   dwconvt_up_add_kernel depthwise ConvTranspose2d(C, C, 2f, stride f, padding f/2, groups=C) up-sampler of IDAUp
//                         (lore_dla_34.py:96-110) fused with the `+ layers[i-1]` of IDAUp.forward.
// Activations are NHWC bf16; in the BF16X3 precision mode every tensor is [hi(C) | lo(C)] per pixel and values are
// hi + lo in fp32.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace PT_FMT_NS {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

namespace {

// 16 stored bits <-> fp32 in the storage format of this namespace (act16.h: bf16, or IEEE half in pt_f16)
__device__ __forceinline__ float bf2f(uint32_t b) { return a16_to_f32(b); }
__device__ __forceinline__ uint32_t f2bf(float f) { return f32_to_a16(f); }

__device__ __forceinline__ void load8(const bf16_t* p, int lo_off, int split, float* v) {
  const u32x4 h = *reinterpret_cast<const u32x4*>(p);
  const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = bf2f((k & 1) ? (hw[k >> 1] >> 16) : (hw[k >> 1] & 0xFFFFu));
  if (split) {
    const u32x4 l = *reinterpret_cast<const u32x4*>(p + lo_off);
    const uint32_t lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += bf2f((k & 1) ? (lw[k >> 1] >> 16) : (lw[k >> 1] & 0xFFFFu));
  }
}

__device__ __forceinline__ void store8(bf16_t* p, int lo_off, int split, const float* v) {
  uint32_t hb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) hb[k] = f2bf(v[k]);
  u32x4 o;
  o.x = hb[0] | (hb[1] << 16); o.y = hb[2] | (hb[3] << 16); o.z = hb[4] | (hb[5] << 16); o.w = hb[6] | (hb[7] << 16);
  *reinterpret_cast<u32x4*>(p) = o;
  if (split) {
    uint32_t lb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) lb[k] = f2bf(v[k] - bf2f(hb[k]));
    o.x = lb[0] | (lb[1] << 16); o.y = lb[2] | (lb[3] << 16); o.z = lb[4] | (lb[5] << 16); o.w = lb[6] | (lb[7] << 16);
    *reinterpret_cast<u32x4*>(p + lo_off) = o;
  }
}

// om: fp32 [pixel][32]: channels 2k / 2k+1 = (dy, dx) of tap k, 18 + k = mask logit of tap k (dcnv2.py:72-75)
__global__ __launch_bounds__(256) void dcn_im2col_kernel(const bf16_t* __restrict__ x, const float* __restrict__ om,
                                                          bf16_t* __restrict__ cols, int B, int H, int W, int C,
                                                          int split) {
  a16_kernel_enter();
  const int cgn = C >> 3;
  const int cs = split ? 2 * C : C;
  const int kc = 9 * C;
  const long long total = (long long)B * H * W * 9 * cgn;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cgn);
    long long t = i / cgn;
    const int tap = (int)(t % 9);
    const long long pix = t / 9;
    const int xw = (int)(pix % W);
    const int yh = (int)((pix / W) % H);
    const int b = (int)(pix / ((long long)W * H));
    const float* o = om + pix * 32;
    const float off_h = o[2 * tap], off_w = o[2 * tap + 1];
    const float mask = 1.f / (1.f + expf(-o[18 + tap]));
    const float h_im = (float)(yh - 1 + tap / 3) + off_h;
    const float w_im = (float)(xw - 1 + tap % 3) + off_w;
    float val[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) val[k] = 0.f;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
      float v1[8], v2[8], v3[8], v4[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v1[k] = v2[k] = v3[k] = v4[k] = 0.f;
      const bf16_t* xb = x + (size_t)b * H * W * cs + cg * 8;
      if (h_low >= 0 && w_low >= 0) load8(xb + ((size_t)h_low * W + w_low) * cs, C, split, v1);
      if (h_low >= 0 && w_high <= W - 1) load8(xb + ((size_t)h_low * W + w_high) * cs, C, split, v2);
      if (h_high <= H - 1 && w_low >= 0) load8(xb + ((size_t)h_high * W + w_low) * cs, C, split, v3);
      if (h_high <= H - 1 && w_high <= W - 1) load8(xb + ((size_t)h_high * W + w_high) * cs, C, split, v4);
#pragma unroll
      for (int k = 0; k < 8; ++k) val[k] = (w1 * v1[k] + w2 * v2[k] + w3 * v3[k] + w4 * v4[k]) * mask;
    }
    store8(cols + (size_t)pix * (split ? 2 * kc : kc) + tap * C + cg * 8, kc, split, val);
  }
}

// out[b, oy, ox, c] = sum_{iy, ix} in[b, iy, ix, c] * w[ky * k + kx][c]  (+ add[b, oy, ox, c]),  ky = oy + p - iy * f
__global__ __launch_bounds__(256) void dwconvt_up_add_kernel(const bf16_t* __restrict__ in, const float* __restrict__ w,
                                                              const bf16_t* __restrict__ add, bf16_t* __restrict__ out,
                                                              int B, int h, int wd, int C, int f, int split) {
  a16_kernel_enter();
  const int cgn = C >> 3;
  const int cs = split ? 2 * C : C;
  const int OH = h * f, OW = wd * f, p = f / 2, k = 2 * f;
  const long long total = (long long)B * OH * OW * cgn;
  const bool small = total < (1ll << 31);      // 32-bit index arithmetic (the 64-bit divides were half of the kernel)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int cg, ox, oy, b;
    if (small) {
      const unsigned iu = (unsigned)i;
      unsigned t = iu / (unsigned)cgn;
      cg = (int)(iu - t * (unsigned)cgn);
      const unsigned t2 = t / (unsigned)OW;
      ox = (int)(t - t2 * (unsigned)OW);
      b = (int)(t2 / (unsigned)OH);
      oy = (int)(t2 - (unsigned)b * (unsigned)OH);
    } else {
      cg = (int)(i % cgn);
      long long t = i / cgn;
      ox = (int)(t % OW);
      t /= OW;
      oy = (int)(t % OH);
      b = (int)(t / OH);
    }
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    const int iy1 = (oy + p) / f, ix1 = (ox + p) / f;
    for (int dy = 0; dy < 2; ++dy) {
      const int iy = iy1 - dy, ky = oy + p - iy * f;
      if (iy < 0 || iy >= h || ky >= k) continue;
      for (int dx = 0; dx < 2; ++dx) {
        const int ix = ix1 - dx, kx = ox + p - ix * f;
        if (ix < 0 || ix >= wd || kx >= k) continue;
        float v[8];
        load8(in + (((size_t)b * h + iy) * wd + ix) * cs + cg * 8, C, split, v);
        const float* wp = w + (size_t)(ky * k + kx) * C + cg * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += v[q] * wp[q];
      }
    }
    const size_t oo = (((size_t)b * OH + oy) * OW + ox) * cs + cg * 8;
    if (add) {
      float a[8];
      load8(add + oo, C, split, a);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += a[q];
    }
    store8(out + oo, C, split, acc);
  }
}

// f = 2 (kernel 4, padding 1), the up-sampler of every IDAUp step but one: a thread owns a 2x2 output block of one 8-channel
// group -- the four outputs read the same 3x3 input neighbourhood (9 loads instead of 16) and the 16 x C weight table comes
// from LDS instead of 128 B per output through L1.  Per output the terms and their order are those of the kernel above
// (dy = 0, 1 outer, dx = 0, 1 inner; terms outside the map skipped), so the results are bit-identical.
// PACKED (plain bf16 maps only): the 3x3 neighbourhood and the block's four `add` vectors stay packed in registers (52 instead of 72 + one add vector at a
// time) and are all requested before the first multiply-add: ~90 registers instead of 122, and 13 loads in flight per thread instead of 9 + 1 + 1 + 1 + 1.
// Same products, same order.
template <int PACKED>
__global__ __launch_bounds__(256) void dwconvt_up2_add_kernel(const bf16_t* __restrict__ in, const float* __restrict__ w,
                                                               const bf16_t* __restrict__ add, bf16_t* __restrict__ out,
                                                               int B, int h, int wd, int C, int split) {
  a16_kernel_enter();
  extern __shared__ __attribute__((aligned(16))) float s_w[];                    // [16][C]
  for (int i = threadIdx.x; i < 16 * C; i += 256) s_w[i] = w[i];
  __syncthreads();
  const int cgn = C >> 3;
  const int cs = split ? 2 * C : C;
  const int OW = wd * 2;
  const unsigned total = (unsigned)B * h * wd * cgn;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned t = i / (unsigned)cgn;
    const int cg = (int)(i - t * (unsigned)cgn);
    const unsigned t2 = t / (unsigned)wd;
    const int ax = (int)(t - t2 * (unsigned)wd);
    const int b = (int)(t2 / (unsigned)h);
    const int ay = (int)(t2 - (unsigned)b * (unsigned)h);
    constexpr bool SP = PACKED == 2;      // PACKED 2: (hi | lo) maps, both halves packed
    float v[PACKED ? 1 : 3][PACKED ? 1 : 3][8];
    u32x4 vraw[PACKED ? 3 : 1][PACKED ? 3 : 1], vlo[SP ? 3 : 1][SP ? 3 : 1], araw[2][2], alo[SP ? 2 : 1][SP ? 2 : 1];
    bool ok[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int iy = ay - 1 + r, ix = ax - 1 + c;
        ok[r][c] = iy >= 0 && iy < h && ix >= 0 && ix < wd;
        if (PACKED) {
          // no branches: the clamped pixel is loaded and zeroed when it is outside the map -- its products are +-0, and an accumulator that started
          // at +0 is never -0, so adding them changes no bit (the float variant skips those terms)
          const int iyc = min(max(iy, 0), h - 1), ixc = min(max(ix, 0), wd - 1);
          const bf16_t* ip = in + (((size_t)b * h + iyc) * wd + ixc) * cs + cg * 8;
          u32x4 rv = *reinterpret_cast<const u32x4*>(ip);
          const uint32_t m = ok[r][c] ? 0xFFFFFFFFu : 0u;
          rv.x &= m; rv.y &= m; rv.z &= m; rv.w &= m;
          vraw[PACKED ? r : 0][PACKED ? c : 0] = rv;
          if (SP) {
            u32x4 rl = *reinterpret_cast<const u32x4*>(ip + C);
            rl.x &= m; rl.y &= m; rl.z &= m; rl.w &= m;
            vlo[SP ? r : 0][SP ? c : 0] = rl;
          }
        } else {
          if (ok[r][c]) load8(in + (((size_t)b * h + iy) * wd + ix) * cs + cg * 8, C, split, v[PACKED ? 0 : r][PACKED ? 0 : c]);
        }
      }
    if (PACKED && add) {
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const bf16_t* ap = add + (((size_t)b * (2 * h) + 2 * ay + py) * OW + 2 * ax + px) * cs + cg * 8;
          araw[py][px] = *reinterpret_cast<const u32x4*>(ap);
          if (SP) alo[SP ? py : 0][SP ? px : 0] = *reinterpret_cast<const u32x4*>(ap + C);
        }
    }
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        // output (2 ay + py, 2 ax + px): rows iy1 = ay + py (ky = 1 - py ... see the general kernel), then iy1 - 1
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const int r = py + 1 - dy;                 // neighbourhood row of iy = ay + py - dy
          const int ky = py + 1 - 2 * (py - dy);     // oy + 1 - 2 iy with oy = 2 ay + py
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const int c = px + 1 - dx;
            const int kx = px + 1 - 2 * (px - dx);
            if (!PACKED && !ok[r][c]) continue;
            const float* wp = s_w + (ky * 4 + kx) * C + cg * 8;
            if (PACKED) {
              const u32x4 rv = vraw[PACKED ? r : 0][PACKED ? c : 0];
              const uint32_t hw[4] = {rv.x, rv.y, rv.z, rv.w};
              if (SP) {      // the value is hi + lo (one rounding), as load8 forms it
                const u32x4 rl = vlo[SP ? r : 0][SP ? c : 0];
                const uint32_t lw[4] = {rl.x, rl.y, rl.z, rl.w};
#pragma unroll
                for (int q = 0; q < 8; ++q)
                  acc[q] += (bf2f((q & 1) ? (hw[q >> 1] >> 16) : (hw[q >> 1] & 0xFFFFu)) + bf2f((q & 1) ? (lw[q >> 1] >> 16) : (lw[q >> 1] & 0xFFFFu))) * wp[q];
              } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += bf2f((q & 1) ? (hw[q >> 1] >> 16) : (hw[q >> 1] & 0xFFFFu)) * wp[q];
              }
            } else {
#pragma unroll
              for (int q = 0; q < 8; ++q) acc[q] += v[PACKED ? 0 : r][PACKED ? 0 : c][q] * wp[q];
            }
          }
        }
        const size_t oo = (((size_t)b * (2 * h) + 2 * ay + py) * OW + 2 * ax + px) * cs + cg * 8;
        if (add) {
          float a[8];
          if (PACKED) {
            const uint32_t hw[4] = {araw[py][px].x, araw[py][px].y, araw[py][px].z, araw[py][px].w};
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = bf2f((k & 1) ? (hw[k >> 1] >> 16) : (hw[k >> 1] & 0xFFFFu));
            if (SP) {
              const u32x4 rl = alo[SP ? py : 0][SP ? px : 0];
              const uint32_t lw[4] = {rl.x, rl.y, rl.z, rl.w};
#pragma unroll
              for (int k = 0; k < 8; ++k) a[k] += bf2f((k & 1) ? (lw[k >> 1] >> 16) : (lw[k >> 1] & 0xFFFFu));
            }
          } else {
            load8(add + oo, C, split, a);
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) acc[q] += a[q];
        }
        store8(out + oo, C, PACKED == 1 ? 0 : split, acc);
      }
  }
}

// cv2.warpAffine(crop, M, (W, H), INTER_LINEAR, BORDER_CONSTANT 0) + (x/255 - mean)/std of TableLorePreProcessor.process
// (lore/processer_lore.py:85-90), sampling the table crop straight from the resident page.  OpenCV's WarpAffineInvoker
// arithmetic: inverse map in fp64, 10-bit fixed-point coordinates, 1/32-pixel positions, 15-bit weights.
// lut[c][v] = float(((v / 255.) - mean[c]) / std[c]) evaluated in fp64 on the host like numpy does.
__global__ __launch_bounds__(256) void tsr_preprocess_kernel(const uint8_t* __restrict__ pages, int ph, int pw,
                                                              const pt_tsr_table* __restrict__ tabs, int H, int W, int bgr,
                                                              const float* __restrict__ lut, bf16_t* __restrict__ out,
                                                              int split) {
  a16_kernel_enter();
  const int t = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  const pt_tsr_table tb = tabs[t];
  auto sat = [](double v) { return (long long)fmin(fmax(rint(v), -2147483648.0), 2147483647.0); };
  const long long adelta = sat(tb.minv[0] * x * 1024.0), bdelta = sat(tb.minv[3] * x * 1024.0);
  const long long X0 = sat((tb.minv[1] * y + tb.minv[2]) * 1024.0) + 16, Y0 = sat((tb.minv[4] * y + tb.minv[5]) * 1024.0) + 16;
  const long long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
  long long sx = X >> 5, sy = Y >> 5;
  sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
  sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
  const int ax = (int)(X & 31), ay = (int)(Y & 31);
  const int wt[4] = {(32 - ay) * (32 - ax) * 32, (32 - ay) * ax * 32, ay * (32 - ax) * 32, ay * ax * 32};
  int acc[3] = {0, 0, 0};
  const uint8_t* pg = pages + (size_t)tb.page * ph * pw * 3;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long yy = sy + (k >> 1), xx = sx + (k & 1);
    if (yy < 0 || yy >= tb.crop_h || xx < 0 || xx >= tb.crop_w) continue;
    const long long py = tb.y0 + yy, px = tb.x0 + xx;
    if (py < 0 || py >= ph || px < 0 || px >= pw) continue;
    const uint8_t* s = pg + ((size_t)py * pw + px) * 3;
    acc[0] += s[0] * wt[k]; acc[1] += s[1] * wt[k]; acc[2] += s[2] * wt[k];
  }
  float v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sc = bgr ? 2 - c : c;
    int u = (acc[sc] + (1 << 14)) >> 15;
    u = u < 0 ? 0 : (u > 255 ? 255 : u);
    v[c] = lut[c * 256 + u];
  }
  bf16_t* o = out + ((size_t)t * H * W + i) * (split ? 8 : 4);
  uint32_t hb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { hb[c] = f2bf(v[c]); o[c] = (bf16_t)hb[c]; }
  o[3] = 0;
  if (split) {
#pragma unroll
    for (int c = 0; c < 3; ++c) o[4 + c] = (bf16_t)f2bf(v[c] - bf2f(hb[c]));
    o[7] = 0;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused modulated deformable 3x3 convolution (+ folded BN + ReLU): the sampled columns never leave the CU.
//   workgroup = 128 consecutive pixels x NB <= 128 output channels, 4 waves (32 pixels each); K = 9 taps x C channels
//   walked in 32-channel chunks.  Per chunk every thread blends the 4 bilinear corners of ONE pixel for 16 channels
//   (two 16-byte loads per corner, position / weights computed once per tap) into a bf16 [128][32] LDS tile (80-byte
//   rows: conflict-free ds_read_b128), the weight chunk [NB][32] is staged next to it, and each wave issues
//   2 x NB/32 v_mfma_f32_32x32x16_bf16.  Gathers of chunk c+1 are in flight while chunk c is multiplied.
//   Same arithmetic as dcn_im2col_kernel + the 1x1 GEMM (columns rounded to bf16 / hi+lo), weights in the same
//   [N/64][chunks][64][32] tiling (K = tap * C + c); BF16X3: chunks [w_hi | w_hi | w_lo], three MFMA passes.
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 dbf16x8;
typedef __attribute__((ext_vector_type(16))) float df32x16;
typedef __attribute__((ext_vector_type(2))) float df2;
// two fp32 -> one dword of two stored values (act16.h: v_cvt_pk_bf16_f32, or clamp + v_cvt_pk_f16_f32 in pt_f16), and back
__device__ __forceinline__ uint32_t pack_df2(df2 v) { return pack_a16x2(v.x, v.y); }
__device__ __forceinline__ df2 unpack_df2(uint32_t pk) { return df2{a16lo_f32(pk), a16hi_f32(pk)}; }

template <int SPLIT, int NB>
__global__ __launch_bounds__(256, (SPLIT || NB > 64) ? 2 : 4) void dcn_fused_kernel(const bf16_t* __restrict__ x, const float* __restrict__ om,
                                                            const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                            bf16_t* __restrict__ out, long long npix, int H, int W, int C,
                                                            int N, int relu) {
#pragma clang fp contract(fast)
  a16_kernel_enter();
  constexpr int ROW = 80;                       // bytes per staged row (32 bf16 + 16 B pad)
  constexpr int NP = SPLIT ? 2 : 1;
  constexpr int NT = NB / 32;                   // 32-wide MFMA column tiles
  __shared__ __attribute__((aligned(16))) char s_a[NP][128 * ROW];
  __shared__ __attribute__((aligned(16))) char s_w[NP][NB * ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  const long long p0 = (long long)blockIdx.x * 128;
  const int n0 = blockIdx.y * NB;
  const int cs = SPLIT ? 2 * C : C;
  const int nslice = C >> 5, nk = 9 * nslice;
  // gather role: pixel gp, channels [16 * gh, 16 * gh + 16) of the current 32-channel slice
  const int gp = tid & 127, gh = tid >> 7;
  long long pix = p0 + gp;
  const bool live = pix < npix;
  if (!live) pix = npix - 1;
  const int xw = (int)(pix % W), yh = (int)((pix / W) % H);
  const bf16_t* xb = x + (size_t)(pix / ((long long)W * H)) * H * W * cs;
  const float* omp = om + pix * 32;
  // weight staging role: 16-byte piece `tid` (and tid + 256 ...) of the [NB][32] chunk
  constexpr int WPIECES = NB * 4 / 256;         // per thread
  const bf16_t* wbase = w + (size_t)(n0 >> 6) * (SPLIT ? 3 : 1) * nk * (64 * 32);

  df32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  int coff[4];
  float cwt[4], mask = 0.f;
  u32x4 rc[NP][4][2];
  u32x4 rw[NP][WPIECES];

  auto geometry = [&](int tap) {
    const float off_h = omp[2 * tap], off_w = omp[2 * tap + 1];
    mask = 1.f / (1.f + expf(-omp[18 + tap]));
    const float h_im = (float)(yh - 1 + tap / 3) + off_h;
    const float w_im = (float)(xw - 1 + tap % 3) + off_w;
#pragma unroll
    for (int k = 0; k < 4; ++k) { coff[k] = -1; cwt[k] = 0.f; }
    if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      cwt[0] = hh * hw; cwt[1] = hh * lw; cwt[2] = lh * hw; cwt[3] = lh * lw;
      if (h_low >= 0 && w_low >= 0) coff[0] = (h_low * W + w_low) * cs;
      if (h_low >= 0 && w_high <= W - 1) coff[1] = (h_low * W + w_high) * cs;
      if (h_high <= H - 1 && w_low >= 0) coff[2] = (h_high * W + w_low) * cs;
      if (h_high <= H - 1 && w_high <= W - 1) coff[3] = (h_high * W + w_high) * cs;
    }
  };
  auto prefetch = [&](int kc) {
    const int tap = kc / nslice, sl = kc - tap * nslice;
    if (sl == 0) geometry(tap);
    const int ch = sl * 32 + gh * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        rc[pp][k][0] = z; rc[pp][k][1] = z;
        if (coff[k] >= 0) {
          const bf16_t* sp = xb + coff[k] + ch + pp * C;
          rc[pp][k][0] = *reinterpret_cast<const u32x4*>(sp);
          rc[pp][k][1] = *reinterpret_cast<const u32x4*>(sp + 8);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < WPIECES; ++j) {
      const int idx = tid + j * 256;            // row = idx >> 2 (0 .. NB-1), 16-byte part = idx & 3
      const int row = idx >> 2, part = idx & 3;
      const bf16_t* wc = wbase + (size_t)(row >> 6) * (SPLIT ? 3 : 1) * nk * (64 * 32) + (size_t)kc * (64 * 32) + (row & 63) * 32 + part * 8;
      rw[0][j] = *reinterpret_cast<const u32x4*>(wc);
      if (SPLIT) rw[1][j] = *reinterpret_cast<const u32x4*>(wc + (size_t)2 * nk * (64 * 32));
    }
  };
  auto commit = [&]() {
    // packed fp32 math (v_pk_fma_f32) and the hardware RNE bf16 conversion (v_cvt_pk_bf16_f32); FMA contraction is on
    // in this kernel: ((w1 v1 + w2 v2) + w3 v3) + w4 v4 with fused roundings, then * mask
    const df2 w0 = {cwt[0], cwt[0]}, w1 = {cwt[1], cwt[1]}, w2 = {cwt[2], cwt[2]}, w3 = {cwt[3], cwt[3]}, mk = {mask, mask};
    uint32_t oh[8], ol[8];
#pragma unroll
    for (int hgrp = 0; hgrp < 2; ++hgrp) {
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        df2 c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t u = e2 == 0 ? rc[0][k][hgrp].x : e2 == 1 ? rc[0][k][hgrp].y : e2 == 2 ? rc[0][k][hgrp].z : rc[0][k][hgrp].w;
          c[k] = df2{a16lo_f32(u), a16hi_f32(u)};
          if (SPLIT) {
            const uint32_t ul = e2 == 0 ? rc[NP - 1][k][hgrp].x : e2 == 1 ? rc[NP - 1][k][hgrp].y
                                : e2 == 2 ? rc[NP - 1][k][hgrp].z : rc[NP - 1][k][hgrp].w;
            c[k] += df2{a16lo_f32(ul), a16hi_f32(ul)};
          }
        }
        df2 v = w0 * c[0];
        v = w1 * c[1] + v;
        v = w2 * c[2] + v;
        v = w3 * c[3] + v;
        v = v * mk;
        const uint32_t hb = pack_df2(v);
        oh[hgrp * 4 + e2] = hb;
        if (SPLIT) ol[hgrp * 4 + e2] = pack_df2(v - unpack_df2(hb));
      }
    }
    char* ap = s_a[0] + gp * ROW + gh * 32;
    *reinterpret_cast<u32x4*>(ap) = u32x4{oh[0], oh[1], oh[2], oh[3]};
    *reinterpret_cast<u32x4*>(ap + 16) = u32x4{oh[4], oh[5], oh[6], oh[7]};
    if (SPLIT) {
      char* al = s_a[NP - 1] + gp * ROW + gh * 32;
      *reinterpret_cast<u32x4*>(al) = u32x4{ol[0], ol[1], ol[2], ol[3]};
      *reinterpret_cast<u32x4*>(al + 16) = u32x4{ol[4], ol[5], ol[6], ol[7]};
    }
#pragma unroll
    for (int j = 0; j < WPIECES; ++j) {
      const int idx = tid + j * 256;
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) *reinterpret_cast<u32x4*>(s_w[pp] + (idx >> 2) * ROW + (idx & 3) * 16) = rw[pp][j];
    }
  };

  const char* a_rd = s_a[0] + (wave * 32 + lx) * ROW + q * 16;
  const char* b_rd = s_w[0] + lx * ROW + q * 16;
  constexpr int A_PLANE = 128 * ROW, W_PLANE = NB * ROW;
  prefetch(0);
  for (int kc = 0; kc < nk; ++kc) {
    __syncthreads();
    commit();
    __syncthreads();
    if (kc + 1 < nk) prefetch(kc + 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const dbf16x8 ah = *reinterpret_cast<const dbf16x8*>(a_rd + kk * 32);
      dbf16x8 al;
      if (SPLIT) al = *reinterpret_cast<const dbf16x8*>(a_rd + A_PLANE + kk * 32);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const dbf16x8 bh = *reinterpret_cast<const dbf16x8*>(b_rd + t * 32 * ROW + kk * 32);
        acc[t] = mfma_32x32x16_a16(ah, bh, acc[t]);
        if (SPLIT) {
          acc[t] = mfma_32x32x16_a16(al, bh, acc[t]);
          const dbf16x8 bl = *reinterpret_cast<const dbf16x8*>(b_rd + W_PLANE + t * 32 * ROW + kk * 32);
          acc[t] = mfma_32x32x16_a16(ah, bl, acc[t]);
        }
      }
    }
  }
  // epilogue: D[row = pixel (r & 3) + 8 (r >> 2) + 4 q][col = channel lx] -> + bias, ReLU, bf16 (hi | lo)
  const int ocs = SPLIT ? 2 * N : N;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = n0 + t * 32 + lx;
    const float bv = bias[n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long op = p0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * q;
      if (op >= npix) continue;
      float v = acc[t][r] + bv;
      if (relu) v = fmaxf(v, 0.f);
      const uint32_t hb = f2bf(v);
      out[(size_t)op * ocs + n] = (bf16_t)hb;
      if (SPLIT) out[(size_t)op * ocs + N + n] = (bf16_t)f2bf(v - bf2f(hb));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 fast path of the fused deformable convolution.  The variant above gives every lane a different pixel, so one
// 16-byte wave load touches 64 cache lines and the kernel runs at the L1 tag rate (measured: 0.246 ms for 64->64 @256^2
// x 8 tables, whatever the VALU work or occupancy).  Here eight consecutive lanes read the eight 16-byte pieces of ONE
// corner pixel's 64-channel run (a full 128-byte line): 8 lines per wave load.  K is walked in (tap, 64-channel) stages;
// a thread blends 4 (pixel, piece) items per stage; LDS rows are 144 bytes (conflict-free ds_read_b128).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef PT_DCN_ABL
#define PT_DCN_ABL 0
#endif
// NTHR = 256 or 512 threads: with 512 a thread blends two (pixel, piece) items per stage instead of four and a wave owns
// one 32-pixel x (NB / 2) block of the product -- the same arithmetic in the same order, ~100 VGPRs instead of ~190, so
// that 16 waves per CU (instead of 8) keep gathers in flight.
// SPLIT = 1 (PT_PRECISION_BF16X3): x / out carry (hi | lo) channel groups, the corner lines of both halves are gathered and
// summed in fp32 before the blend, the blended value is split again, and the product runs as three MFMA passes
// (a_hi w_hi + a_lo w_hi + a_hi w_lo) -- the arithmetic of dcn_fused_kernel<1, .> on this kernel's gather / LDS layout.
// EARLY = 1 (bf16 mode): an item's corner loads for stage st + 1 are issued right after the item has been blended for stage st -- its registers are
// free from then on -- instead of after the stage's second barrier: the loads then fly during the rest of the blend, the barrier and the product
// (the kernel is bound by its phases -- gather wait, blend, barrier, product, barrier -- not by a resource: profiles/r04/dcn_op_bench.txt).  Same
// arithmetic in the same order, no extra registers: same bits.
// OMF = 1: the layer's 27-channel offset / mask convolution (3x3, C -> 18 offsets + 9 mask logits, lore/dcnv2.py:71-75) runs in THIS kernel's prologue
// instead of as a launch of its own whose fp32 output travels through HBM (8.4 MB per 256 x 256 map out and back in; sixteen launches per table batch):
// the (8 + 2) x (16 + 2) halo of the tile is staged 32 channels at a time in the LDS the sampling tables will occupy, each of the first four waves multiplies
// one 32-pixel row tile against the `.om` weight tiles (32 of their 64 rows: the rest is zero padding) straight from L2, chunk -> tap -> k-step like
// conv_igemm_kernel<3, 1, 0, NHALF = 1> (same sums in the same order), and the 27 values per pixel land in s_om where the table build reads them.  With
// eight waves the taps are split between two wave groups and summed through s_om.  SPLIT: K walks (x_hi, w_hi), (x_lo, w_hi), (x_hi, w_lo) over the `.om.w3` tiles.
template <int NB, int NTHR, int SPLIT = 0, int EARLY = 0, int OMF = 0>
__global__ __launch_bounds__(NTHR, SPLIT ? (NB == 128 ? 1 : 2) : NTHR / 128) void dcn_fused64_kernel(const bf16_t* __restrict__ x, const float* __restrict__ om,
                                                              const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                              bf16_t* __restrict__ out, long long npix, int H, int W,
                                                              int C, int N, int relu, const bf16_t* __restrict__ omw = nullptr,
                                                              const float* __restrict__ omb = nullptr) {
#pragma clang fp contract(fast)
  a16_kernel_enter();
  constexpr int ROW = 144;                      // 64 bf16 + 16 B pad
  constexpr int IT = 1024 / NTHR;               // (pixel, 16-byte piece) items per thread per stage
  constexpr int PSTEP = NTHR / 8;               // pixel distance between a thread's items
  constexpr int WP = NB * 8 / NTHR;             // 16-byte weight pieces per thread per stage
  constexpr int NT = NB * 8 / NTHR;             // 32-column tiles of the product per wave
  constexpr int NP = SPLIT ? 2 : 1;             // operand planes in LDS (hi, lo)
  // sampling geometry of the tile, computed ONCE per (pixel, tap) -- the eight lanes that share a pixel used to redo it
  // every stage, which was half of the kernel's VALU time: element offsets of the four corners (-1 = outside the map),
  // their bilinear weights, and the sigmoid mask
  __shared__ __attribute__((aligned(16))) unsigned s_goff[128 * 9][4];
  __shared__ __attribute__((aligned(16))) float s_gwt[128 * 9][4];
  // the offset / mask values are only needed while the table is built: they share LDS with the operand images
  __shared__ __attribute__((aligned(16))) char s_ab[NP * (128 * ROW + NB * ROW)];
  static_assert(128 * ROW + NB * ROW >= 128 * 28 * 4, "operand images must cover the staged offset/mask rows");
  constexpr int A_PLANE = 128 * ROW, W_PLANE = NB * ROW;
  char* s_a = s_ab;                              // plane p at + p * A_PLANE
  char* s_w = s_ab + NP * A_PLANE;               // plane p at + p * W_PLANE
  float* s_om = reinterpret_cast<float*>(s_ab);   // 27 offset / mask values per pixel (row pitch 28)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  // the workgroup's 128 pixels are an 8-row x 16-column patch of one map: its 9-tap sampling footprint (~10 x 18 lines
  // of 128 B) fits the 32 KB L1, where a 128-pixel row segment (3 x 130 lines) does not and every corner went to L2
  const int tiles_x = (W + 15) >> 4, tiles_y = (H + 7) >> 3;
  int Lb = blockIdx.x;
  const int nblk = blockIdx.y;
  const int tx0 = (Lb % tiles_x) * 16;
  Lb /= tiles_x;
  const int ty0 = (Lb % tiles_y) * 8;
  const long long img0 = (long long)(Lb / tiles_y) * H * W;     // first pixel of this tile's map
  // local pixel pl -> (y, x), clamped into the map for loads; ok = inside the map
  auto locate = [&](int pl, int& y, int& xq) -> bool {
    y = ty0 + (pl >> 4);
    xq = tx0 + (pl & 15);
    const bool ok = y < H && xq < W;
    y = y < H ? y : H - 1;
    xq = xq < W ? xq : W - 1;
    return ok;
  };
  const int n0 = nblk * NB;
  const int nss = C >> 6, nst = 9 * nss, nk = 9 * (C >> 5);
  const int cs = SPLIT ? 2 * C : C;             // channels per pixel in memory
  const int piece = tid & 7, prow = tid >> 3;   // items: pixels prow + 32 j, j = 0..3, 16-byte piece `piece`
  const char* xmap = reinterpret_cast<const char*>(x + (size_t)img0 * cs);     // wave-uniform: the gathers are scalar base + 32-bit lane offset
  if constexpr (OMF) {
    constexpr int HP = 180, HROW = 80;            // halo pixels (10 rows x 18 columns); bytes per staged pixel: 32 channels + 16 B pad (conflict-free ds_read_b128)
    constexpr int HIT = (HP * 4 + NTHR - 1) / NTHR;   // 16-byte halo pieces per thread per chunk
    constexpr int NG = NTHR / 256;                // wave groups that share the taps
    static_assert(sizeof(s_goff) >= HP * HROW && sizeof(s_gwt) >= HP * HROW, "a halo buffer must fit each table's space");
    char* s_h = reinterpret_cast<char*>(&s_goff[0][0]);     // the two sampling tables' space holds the two halo buffers: the tables are built after the last use
    char* s_h2 = reinterpret_cast<char*>(&s_gwt[0][0]);
    const int nch = (SPLIT ? 3 : 1) * (C >> 5);
    const int grp = wave >> 2, mt_o = wave & 3;
    const int t_lo = NG == 2 ? (grp ? 5 : 0) : 0, t_hi = NG == 2 ? (grp ? 9 : 5) : 9;
    // halo piece i -> pixel i >> 2 (row hy, column hx of the halo), 16-byte part i & 3; outside the map: zeros (the conv's padding)
    u32x4 hreg[HIT];
    auto halo_load = [&](int kc) {
      int c0 = kc * 32;                           // element offset into a pixel's [hi(C) | lo(C)] channels
      if (SPLIT && kc >= 2 * (C >> 5)) c0 -= 2 * C;      // third pass: x_hi again (against w_lo)
#pragma unroll
      for (int j = 0; j < HIT; ++j) {
        const int i = tid + j * NTHR;
        const int hp = i >> 2, hy = hp / 18, hx = hp - hy * 18;
        const int y = ty0 - 1 + hy, xq = tx0 - 1 + hx;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (i < HP * 4 && (unsigned)y < (unsigned)H && (unsigned)xq < (unsigned)W)
          v = *reinterpret_cast<const u32x4*>(xmap + ((size_t)y * W + xq) * (2 * cs) + (size_t)c0 * 2 + (i & 3) * 16);
        hreg[j] = v;
      }
    };
    auto halo_store = [&](char* buf) {
#pragma unroll
      for (int j = 0; j < HIT; ++j) {
        const int i = tid + j * NTHR;
        if (i < HP * 4) *reinterpret_cast<u32x4*>(buf + (i >> 2) * HROW + (i & 3) * 16) = hreg[j];
      }
    };
    df32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    const int pl_o = mt_o * 32 + lx;              // this lane's pixel of the A fragment: row pl_o >> 4, column pl_o & 15 of the tile
    const int a_off = ((pl_o >> 4) * 18 + (pl_o & 15)) * HROW + q * 16;
    const bf16_t* wl = omw + (size_t)lx * 32 + q * 8;     // row lx (< 32) of a [64][32] tile, k-half q
    // B fragments (rows lx of the weight tiles) of ALL the group's taps of a chunk live in registers and are re-loaded for the next chunk right behind
    // their last use: every load has a whole chunk (halo store, barrier, the other taps' MFMAs) to arrive -- under the gathers of the co-resident
    // workgroups an L2 hit takes a microsecond, and a tap-by-tap prefetch made the prologue a chain of nine of them per chunk
    constexpr int NTAP = NG == 2 ? 5 : 9;
    dbf16x8 bfr[NTAP][2];
    auto load_b = [&](int kc, int t) {
      const bf16_t* wt = wl + ((size_t)kc * 9 + (t_lo + t)) * 2048;
      bfr[t][0] = *reinterpret_cast<const dbf16x8*>(wt);
      bfr[t][1] = *reinterpret_cast<const dbf16x8*>(wt + 16);
    };
    halo_load(0);
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
      if (t_lo + t < t_hi) load_b(0, t);
    for (int kc = 0; kc < nch; ++kc) {
      char* buf = (kc & 1) ? s_h2 : s_h;
      halo_store(buf);
      if (kc + 1 < nch) halo_load(kc + 1);
      __syncthreads();            // buffer kc & 1 is complete; its previous readers (chunk kc - 2) passed the barrier of chunk kc - 1
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const int tap = t_lo + t;
        if (tap < t_hi) {         // wave-uniform
          const char* ap = buf + a_off + ((tap / 3) * 18 + (tap % 3)) * HROW;
          const dbf16x8 a0 = *reinterpret_cast<const dbf16x8*>(ap);
          const dbf16x8 a1 = *reinterpret_cast<const dbf16x8*>(ap + 32);
          oacc = mfma_32x32x16_a16(a0, bfr[t][0], oacc);      // D = [pixel][offset / mask channel]
          oacc = mfma_32x32x16_a16(a1, bfr[t][1], oacc);
          if (kc + 1 < nch) load_b(kc + 1, t);
        }
      }
    }
    // s_om aliases the operand images, not the halo buffers: no barrier needed before it is written.  A lane holds channel lx of 16 pixels: group 0 stores
    // (+ bias), group 1 adds its taps' share
    if (wave < 4 && lx < 27) {
      const float bo = omb[lx];
#pragma unroll
      for (int r = 0; r < 16; ++r) s_om[(mt_o * 32 + (r & 3) + 8 * (r >> 2) + 4 * q) * 28 + lx] = oacc[r] + bo;
    }
    if (NG == 2) {
      __syncthreads();
      if (wave >= 4 && lx < 27) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_om[(mt_o * 32 + (r & 3) + 8 * (r >> 2) + 4 * q) * 28 + lx] += oacc[r];
      }
    }
  } else {
    for (int i = tid; i < 128 * 7; i += NTHR) {     // 7 float4 per pixel (channels 0..27; 27 is padding)
      int y, xq;
      locate(i / 7, y, xq);
      const long long pix = img0 + (long long)y * W + xq;
      *reinterpret_cast<float4*>(s_om + (i / 7) * 28 + (i % 7) * 4) = *reinterpret_cast<const float4*>(om + pix * 32 + (i % 7) * 4);
    }
  }
  __syncthreads();
  for (int i = tid; i < 128 * 9; i += NTHR) {
    const int pl = i / 9, tap = i - pl * 9;
    int yh, xw;
    locate(pl, yh, xw);
    const float* o = s_om + pl * 28;
    const float off_h = o[2 * tap], off_w = o[2 * tap + 1];
    // the sigmoid mask is folded into the bilinear weights here: one multiply per (pixel, tap) instead of one per blended value
    const float gm = 1.f / (1.f + expf(-o[18 + tap]));
    const float h_im = (float)(yh - 1 + tap / 3) + off_h;
    const float w_im = (float)(xw - 1 + tap % 3) + off_w;
    // A corner outside the map contributes zero (dcn_v2_im2col_cpu.cpp:26-55): its WEIGHT is zeroed here and its offset
    // points at the map's first pixel, so that the stage loop gathers unconditionally -- no predicate, no zero-filled
    // registers, no branch per corner (those were 40 % of the kernel's VALU instructions).  0 * finite = +-0: same sums.
    unsigned co[4] = {0u, 0u, 0u, 0u};            // BYTE offsets from the map's first pixel (< 2^32, checked by the launcher)
    float cw[4] = {0.f, 0.f, 0.f, 0.f};
    if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      if (h_low >= 0 && w_low >= 0) { co[0] = (unsigned)(h_low * W + w_low) * (unsigned)(2 * cs); cw[0] = hh * hw * gm; }
      if (h_low >= 0 && w_high <= W - 1) { co[1] = (unsigned)(h_low * W + w_high) * (unsigned)(2 * cs); cw[1] = hh * lw * gm; }
      if (h_high <= H - 1 && w_low >= 0) { co[2] = (unsigned)(h_high * W + w_low) * (unsigned)(2 * cs); cw[2] = lh * hw * gm; }
      if (h_high <= H - 1 && w_high <= W - 1) { co[3] = (unsigned)(h_high * W + w_high) * (unsigned)(2 * cs); cw[3] = lh * lw * gm; }
    }
    *reinterpret_cast<uint4*>(s_goff[i]) = make_uint4(co[0], co[1], co[2], co[3]);
    *reinterpret_cast<float4*>(s_gwt[i]) = make_float4(cw[0], cw[1], cw[2], cw[3]);
  }
  __syncthreads();     // table complete; s_om is dead from here on (s_a / s_w take its place at the first commit)
  const bf16_t* wbase = w + (size_t)(n0 >> 6) * (SPLIT ? 3 : 1) * nk * (64 * 32);

  df32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  unsigned coff[IT][4];
  float cwt[IT][4];
  u32x4 rc[IT][4], rcl[SPLIT ? IT : 1][4];
  u32x4 rw[WP], rwl[SPLIT ? WP : 1];

  auto geometry = [&](int tap) {
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int gi = (prow + PSTEP * j) * 9 + tap;
      const uint4 o = *reinterpret_cast<const uint4*>(s_goff[gi]);
      const float4 wv = *reinterpret_cast<const float4*>(s_gwt[gi]);
      coff[j][0] = o.x + piece * 16; coff[j][1] = o.y + piece * 16; coff[j][2] = o.z + piece * 16; coff[j][3] = o.w + piece * 16;
      cwt[j][0] = wv.x; cwt[j][1] = wv.y; cwt[j][2] = wv.z; cwt[j][3] = wv.w;
    }
  };
  auto prefetch = [&](int st) {
    const int tap = st / nss, ss = st - tap * nss;
    if (ss == 0) geometry(tap);
    const char* xs = xmap + ss * 128;             // this stage's 64-channel slice (uniform)
#pragma unroll
    for (int j = 0; j < IT; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#if PT_DCN_ABL == 1     /* ablation: no gather traffic */
        rc[j][k] = u32x4{coff[j][k], 0u, 0u, 0u};
#else
        rc[j][k] = *reinterpret_cast<const u32x4*>(xs + coff[j][k]);
#endif
        if (SPLIT) rcl[j][k] = *reinterpret_cast<const u32x4*>(xs + 2 * C + coff[j][k]);
      }
    const int kc = tap * (C >> 5) + 2 * ss;     // the stage's two 32-channel weight chunks are adjacent
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const int idx = tid + j * NTHR;           // row = idx >> 3, piece = idx & 7 (0..3: chunk kc, 4..7: chunk kc + 1)
      const int row = idx >> 3, pc = idx & 7;
      const bf16_t* wsrc = wbase + (size_t)(row >> 6) * (SPLIT ? 3 : 1) * nk * (64 * 32) + (size_t)(kc + (pc >> 2)) * (64 * 32) +
                           (row & 63) * 32 + (pc & 3) * 8;
      rw[j] = *reinterpret_cast<const u32x4*>(wsrc);
      if (SPLIT) rwl[j] = *reinterpret_cast<const u32x4*>(wsrc + (size_t)2 * nk * (64 * 32));     // chunks [w_hi | w_hi | w_lo]
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < IT; ++j) {
#if PT_DCN_ABL == 2     /* ablation: no blend arithmetic */
      *reinterpret_cast<u32x4*>(s_a + (prow + PSTEP * j) * ROW + piece * 16) = rc[j][0] ^ rc[j][1] ^ rc[j][2] ^ rc[j][3];
      continue;
#endif
      const df2 w0 = {cwt[j][0], cwt[j][0]}, w1 = {cwt[j][1], cwt[j][1]}, w2 = {cwt[j][2], cwt[j][2]},
                w3 = {cwt[j][3], cwt[j][3]};
      uint32_t o[4], ol[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        if constexpr (PT_A16_HAS_MIX_BLEND && !SPLIT) {      // half storage: eight mixed-precision FMAs per dword, no unpack, no pack (act16.h)
          o[e2] = a16_blend4(rc[j][0][e2], rc[j][1][e2], rc[j][2][e2], rc[j][3][e2], cwt[j][0], cwt[j][1], cwt[j][2], cwt[j][3]);
          continue;
        }
        df2 c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t u = e2 == 0 ? rc[j][k].x : e2 == 1 ? rc[j][k].y : e2 == 2 ? rc[j][k].z : rc[j][k].w;
          c[k] = df2{a16lo_f32(u), a16hi_f32(u)};
          if (SPLIT) {
            const uint32_t ul = e2 == 0 ? rcl[j][k].x : e2 == 1 ? rcl[j][k].y : e2 == 2 ? rcl[j][k].z : rcl[j][k].w;
            c[k] += df2{a16lo_f32(ul), a16hi_f32(ul)};
          }
        }
        df2 v = w0 * c[0];
        v = w1 * c[1] + v;
        v = w2 * c[2] + v;
        v = w3 * c[3] + v;
        const uint32_t hb = pack_df2(v);
        o[e2] = hb;
        if (SPLIT) ol[e2] = pack_df2(v - unpack_df2(hb));
      }
      *reinterpret_cast<u32x4*>(s_a + (prow + PSTEP * j) * ROW + piece * 16) = u32x4{o[0], o[1], o[2], o[3]};
      if (SPLIT) *reinterpret_cast<u32x4*>(s_a + A_PLANE + (prow + PSTEP * j) * ROW + piece * 16) = u32x4{ol[0], ol[1], ol[2], ol[3]};
    }
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const int idx = tid + j * NTHR;
      *reinterpret_cast<u32x4*>(s_w + (idx >> 3) * ROW + (idx & 7) * 16) = rw[j];
      if (SPLIT) *reinterpret_cast<u32x4*>(s_w + W_PLANE + (idx >> 3) * ROW + (idx & 7) * 16) = rwl[j];
    }
  };

  const int mt = wave & 3, ct0 = (wave >> 2) * NT;      // this wave's 32-pixel row tile and first 32-column tile
  const char* a_rd = s_a + (mt * 32 + lx) * ROW + q * 16;
  const char* b_rd = s_w + (ct0 * 32 + lx) * ROW + q * 16;
  auto product = [&]() {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const dbf16x8 a = *reinterpret_cast<const dbf16x8*>(a_rd + kk * 32);
      dbf16x8 al;
      if (SPLIT) al = *reinterpret_cast<const dbf16x8*>(a_rd + A_PLANE + kk * 32);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const dbf16x8 b = *reinterpret_cast<const dbf16x8*>(b_rd + t * 32 * ROW + kk * 32);
#if PT_DCN_ABL == 3     /* ablation: one MFMA per stage instead of eight */
        if (kk == 0 && t == 0)
#endif
        acc[t] = mfma_32x32x16_a16(b, a, acc[t]);   // D = [channel][pixel]
        if (SPLIT) {
          acc[t] = mfma_32x32x16_a16(b, al, acc[t]);
          const dbf16x8 bl = *reinterpret_cast<const dbf16x8*>(b_rd + W_PLANE + t * 32 * ROW + kk * 32);
          acc[t] = mfma_32x32x16_a16(bl, a, acc[t]);
        }
      }
    }
  };
  if constexpr (EARLY && !SPLIT) {
    prefetch(0);
    for (int st = 0; st < nst; ++st) {
      const bool more = st + 1 < nst;
      const int tap_n = (st + 1) / nss, ss_n = (st + 1) - tap_n * nss;
      const char* xs_n = xmap + ss_n * 128;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < IT; ++j) {
        const df2 w0 = {cwt[j][0], cwt[j][0]}, w1 = {cwt[j][1], cwt[j][1]}, w2 = {cwt[j][2], cwt[j][2]}, w3 = {cwt[j][3], cwt[j][3]};
        uint32_t o[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          if constexpr (PT_A16_HAS_MIX_BLEND) {
            o[e2] = a16_blend4(rc[j][0][e2], rc[j][1][e2], rc[j][2][e2], rc[j][3][e2], cwt[j][0], cwt[j][1], cwt[j][2], cwt[j][3]);
            continue;
          }
          df2 c[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t u = e2 == 0 ? rc[j][k].x : e2 == 1 ? rc[j][k].y : e2 == 2 ? rc[j][k].z : rc[j][k].w;
            c[k] = df2{a16lo_f32(u), a16hi_f32(u)};
          }
          df2 v = w0 * c[0];
          v = w1 * c[1] + v;
          v = w2 * c[2] + v;
          v = w3 * c[3] + v;
          o[e2] = pack_df2(v);
        }
        *reinterpret_cast<u32x4*>(s_a + (prow + PSTEP * j) * ROW + piece * 16) = u32x4{o[0], o[1], o[2], o[3]};
        if (more) {          // this item's registers are free: its loads of the next stage go out now
          if (ss_n == 0) {
            const int gi = (prow + PSTEP * j) * 9 + tap_n;
            const uint4 og = *reinterpret_cast<const uint4*>(s_goff[gi]);
            const float4 wv = *reinterpret_cast<const float4*>(s_gwt[gi]);
            coff[j][0] = og.x + piece * 16; coff[j][1] = og.y + piece * 16; coff[j][2] = og.z + piece * 16; coff[j][3] = og.w + piece * 16;
            cwt[j][0] = wv.x; cwt[j][1] = wv.y; cwt[j][2] = wv.z; cwt[j][3] = wv.w;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) rc[j][k] = *reinterpret_cast<const u32x4*>(xs_n + coff[j][k]);
        }
      }
#pragma unroll
      for (int j = 0; j < WP; ++j) {
        const int idx = tid + j * NTHR;
        *reinterpret_cast<u32x4*>(s_w + (idx >> 3) * ROW + (idx & 7) * 16) = rw[j];
      }
      if (more) {
        const int kc = tap_n * (C >> 5) + 2 * ss_n;
#pragma unroll
        for (int j = 0; j < WP; ++j) {
          const int idx = tid + j * NTHR;
          const int row = idx >> 3, pc = idx & 7;
          rw[j] = *reinterpret_cast<const u32x4*>(wbase + (size_t)(row >> 6) * nk * (64 * 32) + (size_t)(kc + (pc >> 2)) * (64 * 32) + (row & 63) * 32 + (pc & 3) * 8);
        }
      }
      __syncthreads();
      product();
    }
  } else {
    prefetch(0);
    for (int st = 0; st < nst; ++st) {
      __syncthreads();
      commit();
      __syncthreads();
      if (st + 1 < nst) prefetch(st + 1);
      product();
    }
  }
  // weights were the MFMA's A operand: a lane owns pixel mt * 32 + lx, its accumulators are runs of four channels
  int y, xq;
  if (locate(mt * 32 + lx, y, xq)) {
    bf16_t* op = out + (size_t)(img0 + (long long)y * W + xq) * (SPLIT ? 2 * N : N) + n0 + ct0 * 32;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int ch = t * 32 + 8 * rg + 4 * q;
        const float4 bs = *reinterpret_cast<const float4*>(bias + n0 + ct0 * 32 + ch);
        float v[4] = {acc[t][rg * 4 + 0] + bs.x, acc[t][rg * 4 + 1] + bs.y, acc[t][rg * 4 + 2] + bs.z, acc[t][rg * 4 + 3] + bs.w};
        if (relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        const uint32_t h0 = f2bf(v[0]), h1 = f2bf(v[1]), h2 = f2bf(v[2]), h3 = f2bf(v[3]);
        *reinterpret_cast<uint2*>(op + ch) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
        if (SPLIT)
          *reinterpret_cast<uint2*>(op + N + ch) = make_uint2(f2bf(v[0] - bf2f(h0)) | (f2bf(v[1] - bf2f(h1)) << 16),
                                                               f2bf(v[2] - bf2f(h2)) | (f2bf(v[3] - bf2f(h3)) << 16));
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dcn_mfma_kernel (bf16 mode): the bilinear blend of the deformable convolution on the matrix pipe.
// dcn_fused64_kernel is VALU-bound on unpack -> fp32 blend -> repack (profiles/r03/dcn_counters.txt: 67 % VALU busy, matrix
// pipe 10 %; profiles/r04/valu_rate_bench.txt: ~4 cycles per VALU instruction, v_pk_fma_f32 no faster than two v_fma_f32,
// v_dot2_f32_bf16 half rate).  Here no gathered value passes through the VALU at all:
//   * a wave owns 32 pixels x 32 channels of a (tap, 64-channel slice) stage.  The four corner lines of four pixels -- 16 rows
//     of 64 bytes -- are ONE global_load_lds_dwordx4 (1 KB, lane-linear in LDS: lane 4 r + j moves piece j of row r =
//     (pixel r >> 2, corner r & 3)); a ring of RING such slots per wave keeps RING - 1 gathers in flight, counted with
//     s_waitcnt vmcnt (no VGPR holds gathered data, so the depth costs LDS only).
//   * blend: D[channel][pixel] = G^T[channel][(pixel', corner)] x Wt[(pixel', corner)][pixel] as v_mfma_f32_32x32x16_bf16 --
//     one slot is one K = 16 step.  G^T is read with ds_read_b64_tr_b16 (the 16-lane groups transpose a [4 rows][16 channels]
//     block: lane = channel, registers = rows); Wt is block-diagonal and lives in registers: lane (pixel n, k-group) holds the
//     four bilinear x mask weights of ITS pixel (bf16, rounded once) in the K slots of that pixel and zeros elsewhere -- one
//     compare + four selects per step.  Corners outside the map carry weight zero and point at the map's first pixel.
//   * product: a lane's 16 accumulators are 16 channels of one pixel; the tr-read hands channel sigma(m) to row m (4-channel
//     pieces 1 and 2 of every 16 swapped), which makes accumulators [8 t, 8 t + 8) eight CONSECUTIVE channels: converted to
//     bf16 in place they ARE the B operand of the 1x1 product (K = 32 channels of the wave; the weight fragment is a plain
//     ds_read_b128 of the tiled weights, as in dcn_fused64_kernel).  No LDS round trip for the sampled columns.
//   * the two waves that share a pixel tile (channel halves of the slice) add their partial sums through LDS once per tile.
// The fp32 sums are associated differently from dcn_fused64_kernel (blend exact in fp32 there, bf16 weights here; K split in
// halves): both are bf16-mode results within tests/test_gpu_dcn_op.py's bounds.  Opt-in (pt_engine_set_dcn_mfma / PT_DCN_MFMA=1): measured
// equal in speed to dcn_fused64_kernel on the bench's offset fields (profiles/r04/experiments.txt).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef PT_DCN_MABL
#define PT_DCN_MABL 0      /* ablations of dcn_mfma_kernel (tools/dcn_mfma_abl.sh): 1 no gathers, 2 no transpose-reads / blend products */
#endif
#ifndef PT_DCN_RING64
#define PT_DCN_RING64 4    /* ring slots per wave of the 64-wide blocks: 4 -> two workgroups per CU, 8 -> one */
#endif
template <int NB, int RING>
struct DcnMfmaSmem {
  static constexpr int BYTES = 128 * 9 * 24 + 2 * NB * 128 + 8 * RING * 1024;     // sampling table + two weight images + the waves' gather rings
};
template <int NB, int RING>
__global__ __launch_bounds__(512, (NB == 64 && RING == 4) ? 2 : 1) void dcn_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ om,
                                                                            const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                                            bf16_t* __restrict__ out, long long npix, int H, int W, int C,
                                                                            int N, int relu) {
#pragma clang fp contract(fast)
  a16_kernel_enter();
  static_assert(RING == 4 || RING == 8, "ring slots must divide the eight steps of a stage");
  constexpr int NT = NB / 32;                   // 32-column tiles of the product (every wave: all NB outputs over its 32 channels)
  constexpr int WP = NB / 64;                   // weight DMAs per wave per stage (1 KB each)
  constexpr int D = RING - 1;                   // gathers in flight ahead of the step being multiplied
  constexpr int GOFF_BYTES = 128 * 9 * 16, GW_BYTES = 128 * 9 * 8, W_BYTES = NB * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_goff = smem;                          // [pixel * 9 + tap][4] byte offsets of the corners from the map's first pixel
  char* s_gw = smem + GOFF_BYTES;               // [pixel * 9 + tap][2] the four weights as bf16 pairs (w0 | w1 << 16, w2 | w3 << 16)
  char* s_w = s_gw + GW_BYTES;                  // two weight images
  char* s_ring = s_w + 2 * W_BYTES;             // 8 waves x RING x 1 KB
  float* s_om = reinterpret_cast<float*>(s_ring);   // offset / mask staging while the table is built (14 KB)
  static_assert(8 * RING * 1024 >= 128 * 28 * 4, "ring must cover the staged offset / mask rows");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  const int pt = wave & 3, hh = wave >> 2;      // 32-pixel tile, channel half of the slice
  const int tiles_x = (W + 15) >> 4, tiles_y = (H + 7) >> 3;
  int Lb = blockIdx.x;
  const int tx0 = (Lb % tiles_x) * 16;
  Lb /= tiles_x;
  const int ty0 = (Lb % tiles_y) * 8;
  const long long img0 = (long long)(Lb / tiles_y) * H * W;
  auto locate = [&](int pl, int& y, int& xq) -> bool {
    y = ty0 + (pl >> 4);
    xq = tx0 + (pl & 15);
    const bool ok = y < H && xq < W;
    y = y < H ? y : H - 1;
    xq = xq < W ? xq : W - 1;
    return ok;
  };
  const int n0 = blockIdx.y * NB;
  const int nss = C >> 6, nst = 9 * nss, nk = 9 * (C >> 5);
  for (int i = tid; i < 128 * 7; i += 512) {
    int y, xq;
    locate(i / 7, y, xq);
    const long long pix = img0 + (long long)y * W + xq;
    *reinterpret_cast<float4*>(s_om + (i / 7) * 28 + (i % 7) * 4) = *reinterpret_cast<const float4*>(om + pix * 32 + (i % 7) * 4);
  }
  __syncthreads();
  for (int i = tid; i < 128 * 9; i += 512) {      // the sampling rule of dcn_fused64_kernel's table (dcn_v2_im2col_cpu.cpp:26-55)
    const int pl = i / 9, tap = i - pl * 9;
    int yh, xw;
    locate(pl, yh, xw);
    const float* o = s_om + pl * 28;
    const float off_h = o[2 * tap], off_w = o[2 * tap + 1];
    const float gm = 1.f / (1.f + expf(-o[18 + tap]));
    const float h_im = (float)(yh - 1 + tap / 3) + off_h;
    const float w_im = (float)(xw - 1 + tap % 3) + off_w;
    unsigned co[4] = {0u, 0u, 0u, 0u};
    float cw[4] = {0.f, 0.f, 0.f, 0.f};
    if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - hf, lw = w_im - wf, hhh = 1.f - lh, hw = 1.f - lw;
      if (h_low >= 0 && w_low >= 0) { co[0] = (unsigned)(h_low * W + w_low) * (unsigned)(2 * C); cw[0] = hhh * hw * gm; }
      if (h_low >= 0 && w_high <= W - 1) { co[1] = (unsigned)(h_low * W + w_high) * (unsigned)(2 * C); cw[1] = hhh * lw * gm; }
      if (h_high <= H - 1 && w_low >= 0) { co[2] = (unsigned)(h_high * W + w_low) * (unsigned)(2 * C); cw[2] = lh * hw * gm; }
      if (h_high <= H - 1 && w_high <= W - 1) { co[3] = (unsigned)(h_high * W + w_high) * (unsigned)(2 * C); cw[3] = lh * lw * gm; }
    }
    *reinterpret_cast<uint4*>(s_goff + i * 16) = make_uint4(co[0], co[1], co[2], co[3]);
    *reinterpret_cast<uint2*>(s_gw + i * 8) = make_uint2(f2bf(cw[0]) | (f2bf(cw[1]) << 16), f2bf(cw[2]) | (f2bf(cw[3]) << 16));
  }
  // weight image of a stage: NB rows x 128 bytes (the stage's 64 channels), moved by LDS-DMA too -- wave v moves rows [8 v, 8 v + 8) of every
  // 64-row block: lane = (row, 16-byte slot), and the slot holds source piece slot ^ ((row >> 1) & 7): with rows 128 bytes apart that XOR
  // puts the 16 rows a ds_read_b128 lane group touches on 16 different bank quads (no padding is possible in a lane-linear DMA image)
  const bf16_t* wbase = w + (size_t)(n0 >> 6) * nk * (64 * 32);
  const int w_row = wave * 8 + (lane >> 3), w_pc = (lane & 7) ^ ((w_row >> 1) & 7);
  const unsigned w_lane = (unsigned)(((w_pc >> 2) * (64 * 32) + w_row * 32 + (w_pc & 3) * 8) * 2);   // bytes from (64-row block, first chunk of the stage)
  auto issue_w = [&](int st, int tap, int ss) {
    const int kc = tap * (C >> 5) + 2 * ss;
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      // scalar base + 32-bit lane offset, written out: hipcc folds the lane part of this one into a 64-bit vector address and then guards
      // the next writes of those address registers with vmcnt(0) -- which drains the gather ring once per stage
      const unsigned long long u = (unsigned long long)(wbase + ((size_t)j * nk + kc) * (64 * 32));
      const unsigned long long ub = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)u);     // (the builtin returns int: no sign extension)
      const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_w + (st & 1) * W_BYTES + (j * 64 + wave * 8) * 128));
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(w_lane), "s"(ub), "s"(dst) : "memory");
    }
  };
  issue_w(0, 0, 0);
  __syncthreads();       // table complete; s_om (in the ring) is dead

  // gather side: lane 4 r + j moves 16-byte piece j of row r of a step = corner (r & 3) of pixel 4 s + (r >> 2) of the wave's tile
  const char* xmap = reinterpret_cast<const char*>(x + (size_t)img0 * C);
  const char* g_tab = s_goff + ((pt * 32 + (lane >> 4)) * 9) * 16 + ((lane >> 2) & 3) * 4;     // this lane's entry of step 0, tap 0
  const unsigned g_lane = (unsigned)(hh * 64 + (lane & 3) * 16);                                // bytes inside the 128-byte slice line
  // (wave-uniform values through readfirstlane: the ring's slot addresses then live in scalar registers -- one s_add + s_mov m0 per gather)
  const unsigned ring_lds = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_ring + wave * (RING * 1024)));
  auto issue = [&](unsigned off, int slot) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xmap + off),
                                     (__attribute__((address_space(3))) void*)(size_t)(ring_lds + (unsigned)(slot * 1024)), 16, 0, 0);
  };
  // blend side.  A operand (gathered, transposed): 16-lane group g = lane >> 4 reads rows 8 (g >> 1) + {0..3} (then + 4) x the 16
  // channels 16 (g & 1) ..; lane a of the group SUPPLIES the address of row a >> 2, 4-channel piece swap(a & 3) (1 <-> 2) and
  // RECEIVES column a: channel sigma(16 (g & 1) + a)
  const int la = lane & 15, lg = lane >> 4;
  const unsigned tr_addr = ring_lds + (unsigned)((8 * (lg >> 1) + (la >> 2)) * 64 + (lg & 1) * 32 + ((((la & 1) << 1) | ((la >> 1) & 1)) * 8));
  const int key = (lx >> 1) - q;                // step s holds this lane's pixel in its K group iff key == 2 s
  const bool odd = lx & 1;
  const unsigned gw_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_gw + ((pt * 32 + lx) * 9) * 8);
  const char* b_rd = s_w + lx * 128;           // + 16-byte slot (channel piece ^ ((row >> 1) & 7))
  const int b_key = (lx >> 1) & 7;

  df32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

#pragma unroll
  for (int s = 0; s < D; ++s) issue(*reinterpret_cast<const unsigned*>(g_tab + s * 576) + g_lane, s);
  int tap = 0, ss = 0;
  for (int st = 0; st < nst; ++st) {
    int tap_n = tap, ss_n = ss + 1;             // the stage after this one (the gathers run D steps ahead)
    if (ss_n == nss) { ss_n = 0; tap_n = tap + 1; }
    const bool last = st + 1 == nst;
    // weight image st: every wave waits for its own piece (the newest D gathers were issued after it), then the barrier publishes all of them
    // and frees image st - 1 for stage st + 1
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
    __builtin_amdgcn_s_barrier();              // (no fence: __syncthreads() would drain the gathers in flight with vmcnt(0))
    if (!last) issue_w(st + 1, tap_n, ss_n);
    u32x2 tw;                                   // (inline: hipcc puts a vmcnt(0) in front of a plain LDS load of this table here, draining the ring)
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(tw) : "v"(gw_addr + (unsigned)(tap * 8)) : "memory");
    const uint32_t e01 = odd ? 0u : tw.x, e23 = odd ? 0u : tw.y, o01 = odd ? tw.x : 0u, o23 = odd ? tw.y : 0u;
    // the eight table entries of the gathers this stage issues (steps D .. 7 of this stage, then 0 .. D - 1 of the next one)
    const char* tb_c = g_tab + tap * 16;
    const char* tb_n = g_tab + tap_n * 16;      // (tap_n = 9 in the last stage: read, never used)
    const unsigned gc = g_lane + (unsigned)(ss * 128), gn = g_lane + (unsigned)(ss_n * 128);
    unsigned gv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      gv[i] = (i + D < 8 ? *reinterpret_cast<const unsigned*>(tb_c + (i + D) * 576) + gc : *reinterpret_cast<const unsigned*>(tb_n + (i + D - 8) * 576) + gn);
    df32x16 bl;
#pragma unroll
    for (int r = 0; r < 16; ++r) bl[r] = 0.f;
    // one K = 16 step: issue the gather D steps ahead, wait for this step's slot (everything but the newest D gathers -- and, while they are
    // younger than the slot, the WP weight DMAs of this stage -- has landed), transpose-read it, multiply by the block-diagonal weights
#define PT_DCN_STEP(S, LAST)                                                                                                             \
    {                                                                                                                                    \
      if (PT_DCN_MABL != 1 && (S + D < 8 || !LAST)) issue(gv[S], (S + D) % RING);                                                        \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LAST ? (S + D < 8 ? D : 7 - S) : (S < D ? D + WP : D)) : "memory");                       \
      u32x2 a_lo = {gv[S], gc}, a_hi = {gn, gv[S]};                                                                                      \
      if (PT_DCN_MABL != 2)                                                                                                              \
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"               \
                     : "=&v"(a_lo), "=&v"(a_hi)                                                                                          \
                     : "v"(tr_addr), "n"((S % RING) * 1024), "n"((S % RING) * 1024 + 256)                                                \
                     : "memory");                                                                                                        \
      const bool mine = key == 2 * S;                                                                                                    \
      const u32x4 bw = {mine ? e01 : 0u, mine ? e23 : 0u, mine ? o01 : 0u, mine ? o23 : 0u};                                             \
      const u32x4 aw = {a_lo.x, a_lo.y, a_hi.x, a_hi.y};                                                                                 \
      if (PT_DCN_MABL != 2 || S == 0)                                                                                                    \
        bl = mfma_32x32x16_a16(__builtin_bit_cast(dbf16x8, aw), __builtin_bit_cast(dbf16x8, bw), bl);     \
    }
    if (!last) {
      PT_DCN_STEP(0, false) PT_DCN_STEP(1, false) PT_DCN_STEP(2, false) PT_DCN_STEP(3, false)
      PT_DCN_STEP(4, false) PT_DCN_STEP(5, false) PT_DCN_STEP(6, false) PT_DCN_STEP(7, false)
    } else {
      PT_DCN_STEP(0, true) PT_DCN_STEP(1, true) PT_DCN_STEP(2, true) PT_DCN_STEP(3, true)
      PT_DCN_STEP(4, true) PT_DCN_STEP(5, true) PT_DCN_STEP(6, true) PT_DCN_STEP(7, true)
    }
#undef PT_DCN_STEP
    // bl[8 t + e] = channel 32 hh + 16 t + 8 q + e of pixel lx: the product's B operand after one conversion
    const char* wrd = b_rd + (st & 1) * W_BYTES;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      u32x4 cf;
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const df2 v = {bl[8 * t + 2 * e2], bl[8 * t + 2 * e2 + 1]};
        cf[e2] = pack_df2(v);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const dbf16x8 wf = *reinterpret_cast<const dbf16x8*>(wrd + nt * 32 * 128 + (((hh * 4 + t * 2 + q) ^ b_key) << 4));
        acc[nt] = mfma_32x32x16_a16(wf, __builtin_bit_cast(dbf16x8, cf), acc[nt]);   // D = [channel][pixel]
      }
    }
    tap = tap_n;
    ss = ss_n;
  }
  // the two channel halves of a pixel tile meet in LDS: wave hh finishes the 32-column tiles [hh NT / 2, (hh + 1) NT / 2)
  __syncthreads();
  float* scr = reinterpret_cast<float*>(smem) + pt * (NT * 16 * 64);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    if ((nt >= NT / 2) != (hh == 1)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) scr[(nt * 16 + r) * 64 + lane] = acc[nt][r];
    }
  __syncthreads();
  int y, xq;
  const bool ok = locate(pt * 32 + lx, y, xq);
  bf16_t* op = out + (size_t)(img0 + (long long)y * W + xq) * N + n0;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    if ((nt >= NT / 2) == (hh == 1)) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int ch = nt * 32 + 8 * rg + 4 * q;
        const float4 bs = *reinterpret_cast<const float4*>(bias + n0 + ch);
        float v[4] = {acc[nt][rg * 4 + 0] + scr[(nt * 16 + rg * 4 + 0) * 64 + lane] + bs.x, acc[nt][rg * 4 + 1] + scr[(nt * 16 + rg * 4 + 1) * 64 + lane] + bs.y,
                      acc[nt][rg * 4 + 2] + scr[(nt * 16 + rg * 4 + 2) * 64 + lane] + bs.z, acc[nt][rg * 4 + 3] + scr[(nt * 16 + rg * 4 + 3) * 64 + lane] + bs.w};
        if (relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (ok) *reinterpret_cast<uint2*>(op + ch) = make_uint2(f2bf(v[0]) | (f2bf(v[1]) << 16), f2bf(v[2]) | (f2bf(v[3]) << 16));
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dcn_mfma2_kernel (bf16 mode, 64 output channels per workgroup): dcn_mfma_kernel re-cut around what gather_rate_bench measured.
//   * FULL 128-byte lines: a wave owns 32 pixels x the slice's 64 channels (two 32-channel M tiles share one block-diagonal weight operand),
//     a K = 16 step is two LDS-DMA instructions (8 rows x 128 B each: 2 pixels x 4 corners) -- half-line rows cost the path 35 % at the bench's
//     offsets.
//   * ONE workgroup of eight waves per CU on a 16 x 16 pixel tile, an 8-step ring of 2 KB slots per wave: 7 steps = 14 KB per wave = 112 KB per
//     CU in flight (the two-workgroup cut had 48 KB); LDS = 8 x 16 KB rings + two 8 KB weight images + 8 x 1 KB of corner offsets + ... = 160 KB.
//   * every wave computes the sampling geometry of ITS pixels itself, one tap ahead (the 27 offset / mask values of a pixel sit in registers):
//     no workgroup table, no table barrier; the only barrier per stage publishes the weight image.
//   * a wave has all 64 channels of its pixels: the product needs no exchange between waves, the epilogue stores from the accumulators.
// All DMAs are inline asm (scalar base + 32-bit lane offset): hipcc guards registers that served as the vector address of an LDS-DMA builtin with
// vmcnt(0), which drains the ring.  Same arithmetic as dcn_mfma_kernel (bf16 bilinear x mask weights) with the K of the product in one piece.
// ---------------------------------------------------------------------------------------------------------------------
struct DcnMfma2Smem {
  static constexpr int W_BYTES = 64 * 128, OFF_BYTES = 8 * 1024, RING_BYTES = 8 * 8 * 2048;
  static constexpr int BYTES = 2 * W_BYTES + OFF_BYTES + RING_BYTES;       // 155 648
};

__global__ __launch_bounds__(512, 2) void dcn_mfma2_kernel(const bf16_t* __restrict__ x, const float* __restrict__ om, const bf16_t* __restrict__ w,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ out, long long npix, int H, int W,
                                                            int C, int N, int relu) {
#pragma clang fp contract(fast)
  a16_kernel_enter();
  constexpr int NT = 2, D = 7;                  // 32-column tiles of the product; K = 16 steps in flight ahead of the one being multiplied
  constexpr int W_BYTES = DcnMfma2Smem::W_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_w = smem;                                            // two weight images (64 rows x 128 B, 16-byte slots XOR-swizzled by (row >> 1) & 7)
  char* s_off = smem + 2 * W_BYTES;                            // per wave: [2 taps][32 pixels][4 corners] byte offsets from the map's first pixel
  char* s_ring = s_off + DcnMfma2Smem::OFF_BYTES;              // per wave: 8 slots x 2 KB (16 rows x 128 B, 16-byte slots XOR 4 on rows with bit 1 set)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = lane & 31, q = lane >> 5;
  const int tiles_x = (W + 15) >> 4, tiles_y = (H + 15) >> 4;
  int Lb = blockIdx.x;
  const int tx0 = (Lb % tiles_x) * 16;
  Lb /= tiles_x;
  const int ty0 = (Lb % tiles_y) * 16;
  const long long img0 = (long long)(Lb / tiles_y) * H * W;
  const int n0 = blockIdx.y * 64;
  const int nss = C >> 6, nk = 9 * (C >> 5);
  // this lane's pixel (lanes lx and lx + 32 share it): patch rows 2 wave, 2 wave + 1
  int py = ty0 + 2 * wave + (lx >> 4), px = tx0 + (lx & 15);
  const bool inside = py < H && px < W;
  py = py < H ? py : H - 1;
  px = px < W ? px : W - 1;
  float omv[28];
  {
    const float* o = om + (img0 + (long long)py * W + px) * 32;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(o + 4 * i);
      omv[4 * i] = v.x; omv[4 * i + 1] = v.y; omv[4 * i + 2] = v.z; omv[4 * i + 3] = v.w;
    }
    // the 27 values pass through an asm statement: hipcc then knows they have landed.  It cannot see the inline-asm DMAs below, and where it still
    // tracked one of these loads as outstanding it guarded the value's first use -- once per tap -- with vmcnt(0), draining the gather ring
    asm volatile("" : "+v"(omv[0]), "+v"(omv[1]), "+v"(omv[2]), "+v"(omv[3]), "+v"(omv[4]), "+v"(omv[5]), "+v"(omv[6]), "+v"(omv[7]), "+v"(omv[8]),
                 "+v"(omv[9]), "+v"(omv[10]), "+v"(omv[11]), "+v"(omv[12]), "+v"(omv[13]));
    asm volatile("" : "+v"(omv[14]), "+v"(omv[15]), "+v"(omv[16]), "+v"(omv[17]), "+v"(omv[18]), "+v"(omv[19]), "+v"(omv[20]), "+v"(omv[21]),
                 "+v"(omv[22]), "+v"(omv[23]), "+v"(omv[24]), "+v"(omv[25]), "+v"(omv[26]));
  }
  // scalar bases of the DMAs (readfirstlane returns int: no sign extension into the high half)
  auto sbase = [](const void* p) -> unsigned long long {
    const unsigned long long u = (unsigned long long)p;
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)u);
  };
  const unsigned long long xbase = sbase(x + (size_t)img0 * C);
  const bf16_t* wbase = w + (size_t)(n0 >> 6) * nk * (64 * 32);
  const int w_row = wave * 8 + (lane >> 3), w_pc = (lane & 7) ^ ((w_row >> 1) & 7);
  const unsigned w_lane = (unsigned)(((w_pc >> 2) * (64 * 32) + w_row * 32 + (w_pc & 3) * 8) * 2);
  const unsigned w_dst = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_w + wave * 1024));
  auto issue_w = [&](int buf, int tap, int ss) {
    const unsigned long long ub = sbase(wbase + (size_t)(tap * (C >> 5) + 2 * ss) * (64 * 32));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(w_lane), "s"(ub), "s"(w_dst + (unsigned)(buf * W_BYTES)) : "memory");
  };
  // sampling geometry of this lane's pixel for one tap (dcn_v2_im2col_cpu.cpp:26-55, as dcn_fused64_kernel's table): corner byte offsets -> the wave's
  // offset table, the four bilinear x mask weights as bf16 pairs -> registers
  char* my_off = s_off + wave * 1024;
  const unsigned my_off_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)my_off;
  auto geometry = [&](float off_h, float off_w, float mlogit, int tap, uint32_t& w01, uint32_t& w23) {
    const float gm = 1.f / (1.f + expf(-mlogit));
    const float h_im = (float)(py - 1 + tap / 3) + off_h;
    const float w_im = (float)(px - 1 + tap % 3) + off_w;
    unsigned co[4] = {0u, 0u, 0u, 0u};
    float cw[4] = {0.f, 0.f, 0.f, 0.f};
    if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      if (h_low >= 0 && w_low >= 0) { co[0] = (unsigned)(h_low * W + w_low) * (unsigned)(2 * C); cw[0] = hh * hw * gm; }
      if (h_low >= 0 && w_high <= W - 1) { co[1] = (unsigned)(h_low * W + w_high) * (unsigned)(2 * C); cw[1] = hh * lw * gm; }
      if (h_high <= H - 1 && w_low >= 0) { co[2] = (unsigned)(h_high * W + w_low) * (unsigned)(2 * C); cw[2] = lh * hw * gm; }
      if (h_high <= H - 1 && w_high <= W - 1) { co[3] = (unsigned)(h_high * W + w_high) * (unsigned)(2 * C); cw[3] = lh * lw * gm; }
    }
    if (q == 0) *reinterpret_cast<uint4*>(my_off + (tap & 1) * 512 + lx * 16) = make_uint4(co[0], co[1], co[2], co[3]);
    w01 = f2bf(cw[0]) | (f2bf(cw[1]) << 16);
    w23 = f2bf(cw[2]) | (f2bf(cw[3]) << 16);
  };
  // gather side: instruction h of a step moves pixels 4 s + 2 h, + 1 (lane >> 5) x corners ((lane >> 3) & 3) x eight 16-byte pieces; the slot a
  // piece lands in is piece ^ 4 on rows with bit 1 set (two transposed reads of four rows then touch all 64 banks)
  const unsigned g_tab = my_off_lds + (unsigned)((lane >> 5) * 16 + ((lane >> 3) & 3) * 4);
  const unsigned g_piece = (unsigned)(((lane & 7) ^ (((lane >> 4) & 1) << 2)) * 16);
  const unsigned ring_lds = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_ring + wave * 16384));
  auto dma = [&](unsigned off, unsigned dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(xbase), "s"(dst) : "memory");
  };
  // blend side (see dcn_mfma_kernel): lane a of a 16-lane group supplies row 8 (g >> 1) + (a >> 2) (+ 4), 4-channel piece swap(a & 3) of the
  // channels 32 mt + 16 (g & 1) .. + 15, and receives column a
  const int la = lane & 15, lg = lane >> 4;
  const int sw4 = ((la & 1) << 1) | ((la >> 1) & 1);
  const unsigned tr_row = (unsigned)((8 * (lg >> 1) + (la >> 2)) * 128 + (sw4 & 1) * 8);
  const unsigned tr_f = (unsigned)((la >> 3) & 1) << 2;             // the row's slot swizzle (bit 1 of the row)
  const unsigned tr0 = ring_lds + tr_row + (((unsigned)(2 * (lg & 1) + (sw4 >> 1)) ^ tr_f) << 4);
  const unsigned tr1 = ring_lds + tr_row + (((unsigned)(4 + 2 * (lg & 1) + (sw4 >> 1)) ^ tr_f) << 4);
  const int key = (lx >> 1) - q;                // step s holds this lane's pixel in its K group iff key == 2 s
  const bool odd = lx & 1;
  const char* b_rd = s_w + lx * 128;
  const int b_key = (lx >> 1) & 7;

  df32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  uint32_t wc01, wc23, wn01 = 0, wn23 = 0;       // weights of the current / the next tap
  geometry(omv[0], omv[1], omv[18], 0, wc01, wc23);
  issue_w(0, 0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the wave's own offsets are in LDS (wave-private: no barrier)
#pragma unroll
  for (int s = 0; s < D; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      unsigned o;
      asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(o) : "v"(g_tab), "n"((4 * s + 2 * h) * 16) : "memory");
      dma(o + g_piece, ring_lds + (unsigned)(s * 2048 + h * 1024));
    }

  // one stage = (tap, 64-channel slice).  GEO: the geometry of tap TAP + 1 is computed at the top (its gathers start in this stage)
#define PT_DCN2_STEP(S, LAST)                                                                                                          \
    {                                                                                                                                  \
      if (S == 0 || !LAST) {                                                                                                           \
        dma(gv[2 * S], ring_lds + (unsigned)(((S + D) % 8) * 2048));                                                                   \
        dma(gv[2 * S + 1], ring_lds + (unsigned)(((S + D) % 8) * 2048 + 1024));                                                        \
      }                                                                                                                                \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LAST ? (S == 0 ? 2 * D : 2 * (7 - S)) : (S < D ? 2 * D + 1 : 2 * D)) : "memory");     \
      u32x2 a0l, a0h, a1l, a1h;                                                                                                        \
      asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t"                                    \
                   "ds_read_b64_tr_b16 %2, %5 offset:%6\n\tds_read_b64_tr_b16 %3, %5 offset:%7\n\ts_waitcnt lgkmcnt(0)"                \
                   : "=&v"(a0l), "=&v"(a0h), "=&v"(a1l), "=&v"(a1h)                                                                    \
                   : "v"(tr0), "v"(tr1), "n"(S * 2048), "n"(S * 2048 + 512)                                                            \
                   : "memory");                                                                                                        \
      const bool mine = key == 2 * S;                                                                                                  \
      const u32x4 bw = {mine ? e01 : 0u, mine ? e23 : 0u, mine ? o01 : 0u, mine ? o23 : 0u};                                           \
      const u32x4 aw0 = {a0l.x, a0l.y, a0h.x, a0h.y}, aw1 = {a1l.x, a1l.y, a1h.x, a1h.y};                                              \
      bl0 = mfma_32x32x16_a16(__builtin_bit_cast(dbf16x8, aw0), __builtin_bit_cast(dbf16x8, bw), bl0);  \
      bl1 = mfma_32x32x16_a16(__builtin_bit_cast(dbf16x8, aw1), __builtin_bit_cast(dbf16x8, bw), bl1);  \
    }
#define PT_DCN2_TAP(TAP)                                                                                                               \
  for (int ss = 0; ss < nss; ++ss) {                                                                                                   \
    const bool last = TAP == 8 && ss + 1 == nss;                                                                                       \
    const bool geo = TAP < 8 && ss + 1 == nss;                                                                                         \
    const int tap_n = geo ? TAP + 1 : TAP, ss_n = geo ? 0 : ss + 1;                                                                    \
    const int st = TAP * nss + ss;                                                                                                     \
    /* this wave's piece of weight image st has landed (14 gathers were issued after it); the barrier publishes all pieces */          \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * D) : "memory");                                                                       \
    __builtin_amdgcn_s_barrier();                                                                                                      \
    if (!last) issue_w((st + 1) & 1, tap_n, ss_n);                                                                                     \
    if (geo) geometry(omv[2 * (TAP < 8 ? TAP + 1 : 0)], omv[2 * (TAP < 8 ? TAP + 1 : 0) + 1], omv[18 + (TAP < 8 ? TAP + 1 : 0)], TAP + 1, wn01, wn23); \
    const uint32_t e01 = odd ? 0u : wc01, e23 = odd ? 0u : wc23, o01 = odd ? wc01 : 0u, o23 = odd ? wc23 : 0u;                         \
    /* table entries of the gathers this stage issues: step 7 of this stage, then steps 0 .. 6 of the next one */                      \
    unsigned gv[16];                                                                                                                   \
    {                                                                                                                                  \
      const unsigned tc = g_tab + (unsigned)((TAP & 1) * 512), tn = g_tab + (unsigned)((tap_n & 1) * 512);                             \
      const unsigned gc = g_piece + (unsigned)(ss * 128), gn = g_piece + (unsigned)(ss_n * 128);                                       \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                               \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                                                 \
        const int stp = i >> 1, h = i & 1;                                                                                             \
        const unsigned a_ = (stp == 0 ? tc + (unsigned)((28 + 2 * h) * 16) : tn + (unsigned)((4 * (stp - 1) + 2 * h) * 16));           \
        unsigned o_;                                                                                                                   \
        asm volatile("ds_read_b32 %0, %1" : "=v"(o_) : "v"(a_) : "memory");                                                            \
        gv[i] = o_;                                                                                                                    \
      }                                                                                                                                \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                               \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) gv[i] += (i < 2 ? gc : gn);                                                       \
    }                                                                                                                                  \
    df32x16 bl0, bl1;                                                                                                                  \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) { bl0[r] = 0.f; bl1[r] = 0.f; }                                                     \
    if (!last) {                                                                                                                       \
      PT_DCN2_STEP(0, false) PT_DCN2_STEP(1, false) PT_DCN2_STEP(2, false) PT_DCN2_STEP(3, false)                                      \
      PT_DCN2_STEP(4, false) PT_DCN2_STEP(5, false) PT_DCN2_STEP(6, false) PT_DCN2_STEP(7, false)                                      \
    } else {                                                                                                                           \
      PT_DCN2_STEP(0, true) PT_DCN2_STEP(1, true) PT_DCN2_STEP(2, true) PT_DCN2_STEP(3, true)                                          \
      PT_DCN2_STEP(4, true) PT_DCN2_STEP(5, true) PT_DCN2_STEP(6, true) PT_DCN2_STEP(7, true)                                          \
    }                                                                                                                                  \
    /* bl<mt>[8 t + e] = channel 32 mt + 16 t + 8 q + e of pixel lx: after one conversion the B operand of the product's k-step 2 mt + t */ \
    const char* wrd = b_rd + (st & 1) * W_BYTES;                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                                 \
      u32x4 cf;                                                                                                                        \
      _Pragma("unroll") for (int e2 = 0; e2 < 4; ++e2) {                                                                               \
        const int r_ = 8 * (ks & 1) + 2 * e2;                                                                                          \
        const df2 v = (ks < 2) ? df2{bl0[r_], bl0[r_ + 1]} : df2{bl1[r_], bl1[r_ + 1]};                                                \
        cf[e2] = pack_df2(v);                                                        \
      }                                                                                                                                \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                                              \
        const dbf16x8 wf = *reinterpret_cast<const dbf16x8*>(wrd + nt * 32 * 128 + (((ks * 2 + q) ^ b_key) << 4));                     \
        acc[nt] = mfma_32x32x16_a16(wf, __builtin_bit_cast(dbf16x8, cf), acc[nt]);                      \
      }                                                                                                                                \
    }                                                                                                                                  \
    if (geo) { wc01 = wn01; wc23 = wn23; }                                                                                             \
  }
  PT_DCN2_TAP(0) PT_DCN2_TAP(1) PT_DCN2_TAP(2) PT_DCN2_TAP(3) PT_DCN2_TAP(4) PT_DCN2_TAP(5) PT_DCN2_TAP(6) PT_DCN2_TAP(7) PT_DCN2_TAP(8)
#undef PT_DCN2_TAP
#undef PT_DCN2_STEP
  // weights were the MFMA's A operand: a lane owns its pixel, the accumulators are runs of four channels
  if (inside) {
    bf16_t* op = out + (size_t)(img0 + (long long)py * W + px) * N + n0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int ch = nt * 32 + 8 * rg + 4 * q;
        const float4 bs = *reinterpret_cast<const float4*>(bias + n0 + ch);
        float v[4] = {acc[nt][rg * 4 + 0] + bs.x, acc[nt][rg * 4 + 1] + bs.y, acc[nt][rg * 4 + 2] + bs.z, acc[nt][rg * 4 + 3] + bs.w};
        if (relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        *reinterpret_cast<uint2*>(op + ch) = make_uint2(f2bf(v[0]) | (f2bf(v[1]) << 16), f2bf(v[2]) | (f2bf(v[3]) << 16));
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 convolution for 16 input channels (DLA-34 level0: 16 -> 16 at full resolution, level1: 16 -> 32 stride 2,
// center_net/modeling_centernet.py:295-298,370-380) + folded BN + ReLU.  The general implicit-GEMM kernel would pad these
// to 32 -> 64 (4-8x the work and twice the bytes at 1024 x 1024); here K = 9 taps x 16 channels = nine MFMA k-steps:
//   * activations are stored with their real 16 channels ([hi16 | lo16] in BF16X3 mode)
//   * the weights of all 9 taps live in registers as B fragments (lane = output channel, N <= 32), no LDS for them
//   * the input patch of a (TH x 64)-pixel tile is staged in LDS with a 48-byte pixel pitch (conflict-free
//     ds_read_b128 at stride 1); a wave owns TH/4 rows = two 32-pixel MFMA tiles per row
// w: bf16 [9][32][16] (output channels >= N zero), [2][9][32][16] = (hi, lo) in BF16X3 mode; bias fp32 [32].
// ---------------------------------------------------------------------------------------------------------------------
template <int STRIDE, int SPLIT>
__global__ __launch_bounds__(256) void conv3x3_c16_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ w,
                                                           const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                           int B, int H, int W, int Ho, int Wo, int N, int tiles_x,
                                                           int tiles_y) {
  a16_kernel_enter();
  constexpr int TH = STRIDE == 1 ? 8 : 4, TW = (STRIDE == 2 && SPLIT) ? 32 : 64;   // (LDS: <= 64 KB static)
  constexpr int RPW = TH / 4;                              // output rows per wave
  constexpr int PH = (TH - 1) * STRIDE + 3, PW = (TW - 1) * STRIDE + 3;
  constexpr int PITCH = 48;                                // bytes per staged pixel (32 data + 16 pad)
  constexpr int NP = SPLIT ? 2 : 1;
  __shared__ __attribute__((aligned(16))) char s_in[NP][PH * PW * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lx = lane & 31, q = lane >> 5;
  int L = blockIdx.x;
  const int txi = L % tiles_x;
  L /= tiles_x;
  const int tyi = L % tiles_y;
  const int b = L / tiles_y;
  const int oy0 = tyi * TH, ox0 = txi * TW, iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
  const int ics = SPLIT ? 32 : 16;
  const bf16_t* in_b = in + (size_t)b * H * W * ics;
  // stage the patch: 2 16-byte pieces per pixel (and per hi / lo plane)
  for (int i = tid; i < PH * PW * 2 * NP; i += 256) {
    const int plane = i / (PH * PW * 2), r = i - plane * (PH * PW * 2);
    const int pix = r >> 1, part = r & 1;
    const int py = pix / PW, px = pix - py * PW;
    const int gy = iy0 + py, gx = ix0 + px;
    u32x4 v = {0u, 0u, 0u, 0u};
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
      v = *reinterpret_cast<const u32x4*>(in_b + ((size_t)gy * W + gx) * ics + plane * 16 + part * 8);
    *reinterpret_cast<u32x4*>(s_in[plane] + pix * PITCH + part * 16) = v;
  }
  // B fragments: lane = output channel lx, k = channels 8q .. 8q+7 of tap t
  dbf16x8 wh[9], wl[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    wh[t] = *reinterpret_cast<const dbf16x8*>(w + ((size_t)t * 32 + lx) * 16 + q * 8);
    if (SPLIT) wl[t] = *reinterpret_cast<const dbf16x8*>(w + (size_t)9 * 32 * 16 + ((size_t)t * 32 + lx) * 16 + q * 8);
  }
  __syncthreads();
  // The MFMA runs with the weights as its A operand: D is [channel][pixel], i.e. a lane owns ONE pixel and its
  // accumulators are runs of four consecutive channels (rows (r & 3) + 8 (r >> 2) + 4 q) -- 8-byte stores from every lane
  // instead of 2-byte stores from the 16 or 32 lanes that hold a valid channel.
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int ty = wave * RPW + rr;
#pragma unroll
    for (int half = 0; half < TW / 32; ++half) {
      df32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int tx = half * 32 + lx;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int off = ((ty * STRIDE + t / 3) * PW + tx * STRIDE + t % 3) * PITCH + q * 16;
        const dbf16x8 ah = *reinterpret_cast<const dbf16x8*>(s_in[0] + off);
        acc = mfma_32x32x16_a16(wh[t], ah, acc);
        if (SPLIT) {
          const dbf16x8 al = *reinterpret_cast<const dbf16x8*>(s_in[NP - 1] + off);
          acc = mfma_32x32x16_a16(wh[t], al, acc);
          acc = mfma_32x32x16_a16(wl[t], ah, acc);
        }
      }
      const int oy = oy0 + ty, ox = ox0 + tx;
      if (oy < Ho && ox < Wo) {
        const int ocs = SPLIT ? 2 * N : N;
        bf16_t* op = out + (((size_t)b * Ho + oy) * Wo + ox) * ocs;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int ch = 8 * rg + 4 * q;
          if (ch >= N) continue;
          const float4 bs = *reinterpret_cast<const float4*>(bias + ch);
          const float v[4] = {fmaxf(acc[rg * 4 + 0] + bs.x, 0.f), fmaxf(acc[rg * 4 + 1] + bs.y, 0.f),
                              fmaxf(acc[rg * 4 + 2] + bs.z, 0.f), fmaxf(acc[rg * 4 + 3] + bs.w, 0.f)};
          uint32_t hb[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) hb[k] = f2bf(v[k]);
          *reinterpret_cast<uint2*>(op + ch) = make_uint2(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16));
          if (SPLIT) {
            uint32_t lb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) lb[k] = f2bf(v[k] - bf2f(hb[k]));
            *reinterpret_cast<uint2*>(op + N + ch) = make_uint2(lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16));
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// DLA-34's three thin levels as ONE kernel (bf16 mode): base_layer (7x7, 3 -> 16, full resolution) -> level0 (3x3, 16 -> 16) ->
// level1 (3x3 stride 2, 16 -> 32), center_net/modeling_centernet.py:291-298,382-402.  As three launches the two 16-channel
// full-resolution maps (33.5 MB per 1024^2 table each) are written to HBM and read back, and the step spends 5.0 ms (r03) on 9.7
// GFLOP per table.  Here a workgroup owns an 8 x 30 tile of the LEVEL1 output and keeps what it needs of the two intermediate maps in
// LDS: the 19 x 63 patch of base_layer's output (halo of level0 + level1 recomputed: 1.19x rows, 1.05x columns) and the 17 x 61 patch
// of level0's.  HBM sees the image once (25 x 69 pixels per tile) and the 32-channel half-resolution map once.
//   * The arithmetic of each level is the stand-alone kernels' (conv_stem7x7_thin_kernel, conv3x3_c16_kernel: same operand layout,
//     same K order, bf16 rounding of every intermediate) -- results are bit-identical to the three-launch path.
//   * A position of an intermediate patch that lies OUTSIDE its map is stored as zero: the next level pads with zeros, not with the
//     response to a padded input.
//   * A workgroup (8 waves) walks all x-tiles of one tile row with the stem's weight rows in LDS and the two levels' nine B fragments in
//     registers.
// in: NHWC4 bf16 [B,H,W,4] (rgb0); w_stem bf16 [64][224] (rows >= 16 zero), w0 / w1 bf16 [9][32][16]; biases fp32 [>= 32]; out bf16
// [B,H/2,W/2,32].  H, W even.
// ---------------------------------------------------------------------------------------------------------------------
struct ThinChainCfg {
  static constexpr int TH1 = 8, TW1 = 30;                 // level1 output tile
  static constexpr int R0 = 2 * TH1 + 1, C0 = 2 * TW1 + 1; // level0 patch: 17 x 61
  static constexpr int RB = R0 + 2, CB = C0 + 2;           // base_layer patch: 19 x 63
  static constexpr int RI = RB + 6, CI = 72;               // image patch: 25 rows x (69 needed; 72 staged: the stem's fragments of column 63 reach 70)
  static constexpr int PW = 64;                            // pixels per LDS row of the two intermediate patches (two 32-pixel MFMA tiles)
  static constexpr int PITCH = 48;                         // bytes per intermediate pixel: 16 bf16 + 16 B pad (conflict-free ds_read_b128)
  static constexpr int WROW = 464;                         // stem weight row: 224 bf16 + 16 B pad
  static constexpr int IN_BYTES = RI * CI * 8, W_BYTES = 32 * WROW;
  static constexpr int B_BYTES = (RB * PW + 4) * PITCH, L0_BYTES = (R0 * PW + 4) * PITCH;      // + 4 pixels: fragments of discarded columns read past a row
  static constexpr int SMEM = IN_BYTES + W_BYTES + B_BYTES + L0_BYTES;
};

__global__ __launch_bounds__(512, 1) void dla_thin_chain_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ w_stem,
                                                                const float* __restrict__ b_stem, const bf16_t* __restrict__ w0,
                                                                const float* __restrict__ b0, const bf16_t* __restrict__ w1,
                                                                const float* __restrict__ b1, bf16_t* __restrict__ out, int B, int H, int W,
                                                                int tiles_x, int tiles_y) {
  a16_kernel_enter();
  using C = ThinChainCfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_in = smem;
  char* s_w = s_in + C::IN_BYTES;
  char* s_b = s_w + C::W_BYTES;
  char* s_l0 = s_b + C::B_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lx = lane & 31, q = lane >> 5;
  const int H1 = H >> 1, W1 = W >> 1;
  const int tyi = blockIdx.x % tiles_y, b = blockIdx.x / tiles_y;
  const int oy1 = tyi * C::TH1;
  const bf16_t* in_b = in + (size_t)b * H * W * 4;
  // stem weight rows 0..31 (channels >= 16 are zero rows), once per workgroup
  for (int idx = tid; idx < 32 * 28; idx += 512) {
    const int row = idx / 28, part = idx - row * 28;
    *reinterpret_cast<u32x4*>(s_w + row * C::WROW + part * 16) = *reinterpret_cast<const u32x4*>(w_stem + (size_t)idx * 8);
  }
  // B fragments of the two 3x3 levels: lane = output channel lx, k = channels 8q .. 8q+7 of tap t
  dbf16x8 wf0[9], wf1[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    wf0[t] = *reinterpret_cast<const dbf16x8*>(w0 + ((size_t)t * 32 + lx) * 16 + q * 8);
    wf1[t] = *reinterpret_cast<const dbf16x8*>(w1 + ((size_t)t * 32 + lx) * 16 + q * 8);
  }
  const int yb0 = 2 * oy1 - 2, y00 = 2 * oy1 - 1, yi0 = 2 * oy1 - 5;      // first map row of the base / level0 / image patches
  // image patch of a tile: zero outside the image (the 7x7 convolution's padding).  The NEXT tile's patch is fetched into registers while
  // this tile's stem runs and lands in LDS behind the stem's barrier (s_in is dead from there to the next tile): no exposed load latency
  constexpr int NLD = (C::RI * C::CI + 511) / 512;
  u32x2 pre[NLD];
  auto fetch = [&](int txi) {
    const int xi0 = 2 * txi * C::TW1 - 5;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int idx = tid + j * 512;
      const int iy = idx / C::CI, ix = idx - iy * C::CI;
      const int gy = yi0 + iy, gx = xi0 + ix;
      pre[j] = u32x2{0u, 0u};
      if (idx < C::RI * C::CI && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
        pre[j] = *reinterpret_cast<const u32x2*>(in_b + ((size_t)gy * W + gx) * 4);
    }
  };
  auto land = [&]() {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int idx = tid + j * 512;
      if (idx < C::RI * C::CI) *reinterpret_cast<u32x2*>(s_in + idx * 8) = pre[j];
    }
  };
  fetch(0);
  land();
  for (int txi = 0; txi < tiles_x; ++txi) {
    const int ox1 = txi * C::TW1;
    const int xb0 = 2 * ox1 - 2, x00 = 2 * ox1 - 1;
    __syncthreads();                       // this tile's image patch is in LDS; the previous tile's level0 patch has been read
    if (txi + 1 < tiles_x) fetch(txi + 1);
    // ---- base_layer: 19 rows x 2 column tiles, 14 k-steps each (7 kernel rows x 2 halves of the 8-wide kernel row)
    for (int u = wave; u < C::RB * 2; u += 8) {
      const int i = u >> 1, ct = u & 1;
      df32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const char* a_base = s_in + (i * C::CI + ct * 32 + lx + 2 * q) * 8;
      const char* b_base = s_w + lx * C::WROW + q * 16;
#pragma unroll
      for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const dbf16x8 wv = *reinterpret_cast<const dbf16x8*>(b_base + (r * 2 + h) * 32);
          const char* ap = a_base + (r * C::CI + 4 * h) * 8;
          const u32x2 lo = *reinterpret_cast<const u32x2*>(ap), hi = *reinterpret_cast<const u32x2*>(ap + 8);
          const u32x4 av = {lo.x, lo.y, hi.x, hi.y};
          acc = mfma_32x32x16_a16(wv, __builtin_bit_cast(dbf16x8, av), acc);
        }
      const int j = ct * 32 + lx, gy = yb0 + i, gx = xb0 + j;
      const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      char* dst = s_b + (i * C::PW + j) * C::PITCH;
#pragma unroll
      for (int rg = 0; rg < 2; ++rg) {      // channels 8 rg + 4 q .. + 3 (rows 16..31 of the product are zero weight rows)
        const int ch = 8 * rg + 4 * q;
        const float4 bs = *reinterpret_cast<const float4*>(b_stem + ch);
        const float v0 = fmaxf(acc[rg * 4 + 0] + bs.x, 0.f), v1 = fmaxf(acc[rg * 4 + 1] + bs.y, 0.f), v2 = fmaxf(acc[rg * 4 + 2] + bs.z, 0.f),
                    v3 = fmaxf(acc[rg * 4 + 3] + bs.w, 0.f);
        uint2 o = make_uint2(f2bf(v0) | (f2bf(v1) << 16), f2bf(v2) | (f2bf(v3) << 16));
        if (!inside) o = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(dst + ch * 2) = o;
      }
    }
    __syncthreads();
    if (txi + 1 < tiles_x) land();         // every stem fragment of this tile has been read
    // ---- level0: 3x3 stride 1 on the base patch, 17 rows x 2 column tiles, 9 taps
    for (int u = wave; u < C::R0 * 2; u += 8) {
      const int i = u >> 1, ct = u & 1;
      df32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const dbf16x8 av = *reinterpret_cast<const dbf16x8*>(s_b + ((i + t / 3) * C::PW + ct * 32 + lx + t % 3) * C::PITCH + q * 16);
        acc = mfma_32x32x16_a16(wf0[t], av, acc);
      }
      const int j = ct * 32 + lx, gy = y00 + i, gx = x00 + j;
      const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && j < C::C0;
      char* dst = s_l0 + (i * C::PW + j) * C::PITCH;
#pragma unroll
      for (int rg = 0; rg < 2; ++rg) {
        const int ch = 8 * rg + 4 * q;
        const float4 bs = *reinterpret_cast<const float4*>(b0 + ch);
        const float v0 = fmaxf(acc[rg * 4 + 0] + bs.x, 0.f), v1 = fmaxf(acc[rg * 4 + 1] + bs.y, 0.f), v2 = fmaxf(acc[rg * 4 + 2] + bs.z, 0.f),
                    v3 = fmaxf(acc[rg * 4 + 3] + bs.w, 0.f);
        uint2 o = make_uint2(f2bf(v0) | (f2bf(v1) << 16), f2bf(v2) | (f2bf(v3) << 16));
        if (!inside) o = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(dst + ch * 2) = o;
      }
    }
    __syncthreads();
    // ---- level1: 3x3 stride 2 on the level0 patch, one row of the tile per wave
    {
      const int oy = wave;
      df32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const dbf16x8 av = *reinterpret_cast<const dbf16x8*>(s_l0 + ((2 * oy + t / 3) * C::PW + 2 * lx + t % 3) * C::PITCH + q * 16);
        acc = mfma_32x32x16_a16(wf1[t], av, acc);
      }
      const int gy = oy1 + oy, gx = ox1 + lx;
      if (lx < C::TW1 && gy < H1 && gx < W1) {
        bf16_t* op = out + (((size_t)b * H1 + gy) * W1 + gx) * 32;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int ch = 8 * rg + 4 * q;
          const float4 bs = *reinterpret_cast<const float4*>(b1 + ch);
          const float v0 = fmaxf(acc[rg * 4 + 0] + bs.x, 0.f), v1 = fmaxf(acc[rg * 4 + 1] + bs.y, 0.f), v2 = fmaxf(acc[rg * 4 + 2] + bs.z, 0.f),
                      v3 = fmaxf(acc[rg * 4 + 3] + bs.w, 0.f);
          *reinterpret_cast<uint2*>(op + ch) = make_uint2(f2bf(v0) | (f2bf(v1) << 16), f2bf(v2) | (f2bf(v3) << 16));
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same chain on the 16 x 16 x 32 MFMA ("v2", the default).  Counters of the kernel above (profiles/r06/thin_chain_stalls.txt): matrix pipe 43 % busy, LDS
// port 44 %, and SIX VALU instructions per LDS instruction -- base_layer and level0 have 16 output channels, the 32 x 32 x 16 MFMA computes 32 rows, so half of
// the matrix time and half of every epilogue (bias, ReLU, rounding of 16 accumulators per lane) is spent on rows of zeros.  Here those two levels run on
// v_mfma_f32_16x16x32: A = the 16 channels' weights (K = 32: one 8-tap kernel row of the 7x7, two taps of the 3x3), B = 16 pixels, D = 4 accumulators
// per lane (four consecutive channels of one pixel: one 8-byte LDS store).  Per 32 pixels: 14 MFMAs of half the cycles for base_layer, 10 (nine taps
// paired, the tenth half zero) against 9 full ones for level0; the epilogue's VALU work halves.  level1 (32 channels) keeps the 32 x 32 x 16 MFMA.
// K is summed in another association than in the stand-alone kernels (32 products per instruction, taps paired): results agree with the three launches
// to fp32 rounding, not bit for bit (tests/test_gpu_tsr.py bounds it; the kernel above stays selectable: PT_DLA_CHAIN=1).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void dla_thin_chain16_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ w_stem,
                                                                  const float* __restrict__ b_stem, const bf16_t* __restrict__ w0,
                                                                  const float* __restrict__ b0, const bf16_t* __restrict__ w1,
                                                                  const float* __restrict__ b1, bf16_t* __restrict__ out, int B, int H, int W,
                                                                  int tiles_x, int tiles_y) {
  a16_kernel_enter();
  using C = ThinChainCfg;
  typedef a16_f32x4 df32x4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_in = smem;
  char* s_b = s_in + C::IN_BYTES + C::W_BYTES;      // (the layout of the kernel above; its weight rows are unused here)
  char* s_l0 = s_b + C::B_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lx = lane & 31, q = lane >> 5;
  const int c16 = lane & 15, kb = lane >> 4;        // 16-wide MFMA: row / column, k-block (= the lane's four output channels 4 kb .. + 3)
  const int H1 = H >> 1, W1 = W >> 1;
  const int tyi = blockIdx.x % tiles_y, b = blockIdx.x / tiles_y;
  const int oy1 = tyi * C::TH1;
  const bf16_t* in_b = in + (size_t)b * H * W * 4;
  // A fragments.  base_layer: kernel row r = K block r of the [7][8][4] row: lane (channel c16, k-block kb) holds taps 2 kb, 2 kb + 1 (x 4 channels)
  // (which eight K indices a k-block means is ours to choose, the same for both operands: lanes l and l + 32 -- k-blocks kb and kb + 2 -- are given
  // ADJACENT 16-byte pieces of LDS, as the q halves of the 32-wide kernels are)
  const int tp = (kb & 1) * 2 + (kb >> 1);          // base_layer: the lane's pair of taps (2 tp, 2 tp + 1) of a kernel row
  dbf16x8 wst[7];
#pragma unroll
  for (int r = 0; r < 7; ++r) wst[r] = *reinterpret_cast<const dbf16x8*>(w_stem + (size_t)c16 * 224 + r * 32 + tp * 8);
  // level0: K block p = taps 2 p, 2 p + 1 (tap 9 does not exist: zero weights); k-block kb = tap 2 p + (kb & 1), channels 8 (kb >> 1) .. + 7
  dbf16x8 wl0[5];
  int toff[5];                                      // the lane's pixel fragment of K block p, relative to the unit's first pixel
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    const int t = 2 * p + (kb & 1);
    const int tc = t < 9 ? t : 8;
    const dbf16x8 wv = *reinterpret_cast<const dbf16x8*>(w0 + ((size_t)tc * 32 + c16) * 16 + (kb >> 1) * 8);
    const u32x4 z = {0u, 0u, 0u, 0u};
    wl0[p] = t < 9 ? wv : __builtin_bit_cast(dbf16x8, z);
    toff[p] = ((tc / 3) * C::PW + tc % 3) * C::PITCH + (kb >> 1) * 16;
  }
  dbf16x8 wf1[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wf1[t] = *reinterpret_cast<const dbf16x8*>(w1 + ((size_t)t * 32 + lx) * 16 + q * 8);
  const float4 bsv = *reinterpret_cast<const float4*>(b_stem + 4 * kb), b0v = *reinterpret_cast<const float4*>(b0 + 4 * kb);
  const int yb0 = 2 * oy1 - 2, y00 = 2 * oy1 - 1, yi0 = 2 * oy1 - 5;      // first map row of the base / level0 / image patches
  constexpr int NLD = (C::RI * C::CI + 511) / 512;
  u32x2 pre[NLD];
  auto fetch = [&](int txi) {
    const int xi0 = 2 * txi * C::TW1 - 5;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int idx = tid + j * 512;
      const int iy = idx / C::CI, ix = idx - iy * C::CI;
      const int gy = yi0 + iy, gx = xi0 + ix;
      pre[j] = u32x2{0u, 0u};
      if (idx < C::RI * C::CI && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
        pre[j] = *reinterpret_cast<const u32x2*>(in_b + ((size_t)gy * W + gx) * 4);
    }
  };
  auto land = [&]() {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int idx = tid + j * 512;
      if (idx < C::RI * C::CI) *reinterpret_cast<u32x2*>(s_in + idx * 8) = pre[j];
    }
  };
  // bias + ReLU + rounding of a lane's four channels of one pixel -> the 8 bytes it stores (zero outside the map: the next level's padding)
  auto finish = [&](const df32x4& acc, const float4& bv, bool inside) {
    uint2 o = make_uint2(pack_a16x2(fmaxf(acc[0] + bv.x, 0.f), fmaxf(acc[1] + bv.y, 0.f)), pack_a16x2(fmaxf(acc[2] + bv.z, 0.f), fmaxf(acc[3] + bv.w, 0.f)));
    if (!inside) o = make_uint2(0u, 0u);
    return o;
  };
  fetch(0);
  land();
  for (int txi = 0; txi < tiles_x; ++txi) {
    const int ox1 = txi * C::TW1;
    const int xb0 = 2 * ox1 - 2, x00 = 2 * ox1 - 1;
    __syncthreads();                       // this tile's image patch is in LDS; the previous tile's level0 patch has been read
    if (txi + 1 < tiles_x) fetch(txi + 1);
    // ---- base_layer: 19 rows x 2 halves of 32 columns = two 16-pixel blocks each (two accumulator chains), 7 K blocks
    for (int u = wave; u < C::RB * 2; u += 8) {
      const int i = u >> 1, ct = u & 1;
      df32x4 acc[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) acc[k] = df32x4{0.f, 0.f, 0.f, 0.f};
      const char* a_base = s_in + (i * C::CI + ct * 32 + c16 + 2 * tp) * 8;
      // the unit's 14 fragments are requested up front (pinned below: left alone hipcc sinks each read to its MFMA and waits for it there)
      u32x4 fr[7][2];
#pragma unroll
      for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const char* ap = a_base + (r * C::CI + 16 * k) * 8;
          const u32x2 lo = *reinterpret_cast<const u32x2*>(ap), hi = *reinterpret_cast<const u32x2*>(ap + 8);
          fr[r][k] = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
#pragma unroll
      for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int k = 0; k < 2; ++k) acc[k] = mfma_16x16x32_a16(wst[r], __builtin_bit_cast(dbf16x8, fr[r][k]), acc[k]);
      __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 14, 0);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int j = ct * 32 + 16 * k + c16, gy = yb0 + i, gx = xb0 + j;
        const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        *reinterpret_cast<uint2*>(s_b + (i * C::PW + j) * C::PITCH + kb * 8) = finish(acc[k], bsv, inside);
      }
    }
    __syncthreads();
    if (txi + 1 < tiles_x) land();         // every stem fragment of this tile has been read
    // ---- level0: 3x3 stride 1 on the base patch, 17 rows x 2 halves, 5 K blocks of two taps
    for (int u = wave; u < C::R0 * 2; u += 8) {
      const int i = u >> 1, ct = u & 1;
      df32x4 acc[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) acc[k] = df32x4{0.f, 0.f, 0.f, 0.f};
      const char* a_base = s_b + (i * C::PW + ct * 32 + c16) * C::PITCH;
      dbf16x8 fr[5][2];
#pragma unroll
      for (int p = 0; p < 5; ++p)
#pragma unroll
        for (int k = 0; k < 2; ++k) fr[p][k] = *reinterpret_cast<const dbf16x8*>(a_base + toff[p] + 16 * k * C::PITCH);
#pragma unroll
      for (int p = 0; p < 5; ++p)
#pragma unroll
        for (int k = 0; k < 2; ++k) acc[k] = mfma_16x16x32_a16(wl0[p], fr[p][k], acc[k]);
      __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int j = ct * 32 + 16 * k + c16, gy = y00 + i, gx = x00 + j;
        const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && j < C::C0;
        // a level0 row is stored even columns first (32 slots), then the odd ones: level1's stride-2 fragments then read 32 CONSECUTIVE slots (column
        // 2 lx + s = slot lx, 32 + lx, lx + 1 for s = 0, 1, 2) -- read at a stride of two 48-byte pixels they hit every bank quad twice
        const int slot = (j & 1) * 32 + (j >> 1);
        *reinterpret_cast<uint2*>(s_l0 + (i * C::PW + slot) * C::PITCH + kb * 8) = finish(acc[k], b0v, inside);
      }
    }
    __syncthreads();
    // ---- level1: 3x3 stride 2 on the level0 patch, one row of the tile per wave (32 channels: the 32 x 32 x 16 MFMA, as above)
    {
      const int oy = wave;
      df32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int sl = (t % 3 == 1 ? 32 : 0) + lx + (t % 3 == 2 ? 1 : 0);      // slot of column 2 lx + t % 3
        const dbf16x8 av = *reinterpret_cast<const dbf16x8*>(s_l0 + ((2 * oy + t / 3) * C::PW + sl) * C::PITCH + q * 16);
        acc = mfma_32x32x16_a16(wf1[t], av, acc);
      }
      const int gy = oy1 + oy, gx = ox1 + lx;
      if (lx < C::TW1 && gy < H1 && gx < W1) {
        bf16_t* op = out + (((size_t)b * H1 + gy) * W1 + gx) * 32;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int ch = 8 * rg + 4 * q;
          const float4 bs = *reinterpret_cast<const float4*>(b1 + ch);
          *reinterpret_cast<uint2*>(op + ch) = make_uint2(pack_a16x2(fmaxf(acc[rg * 4 + 0] + bs.x, 0.f), fmaxf(acc[rg * 4 + 1] + bs.y, 0.f)),
                                                          pack_a16x2(fmaxf(acc[rg * 4 + 2] + bs.z, 0.f), fmaxf(acc[rg * 4 + 3] + bs.w, 0.f)));
        }
      }
    }
  }
}

inline int grid_for(long long total) {
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace

int pt_launch_dcn_im2col(const bf16_t* x, const float* om, bf16_t* cols, int B, int H, int W, int C, int split,
                         hipStream_t s) {
  PT_REQUIRE(x && om && cols && C % 8 == 0, "dcn im2col: bad arguments");
  hipLaunchKernelGGL(dcn_im2col_kernel, dim3(grid_for((long long)B * H * W * 9 * (C / 8))), dim3(256), 0, s, x, om, cols,
                     B, H, W, C, split);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_launch_dwconvt_up_add(const bf16_t* in, const float* w, const bf16_t* add, bf16_t* out, int B, int h, int wd,
                             int C, int f, int split, hipStream_t s) {
  PT_REQUIRE(in && w && out && C % 8 == 0 && (f == 2 || f == 4), "dwconvT: bad arguments");
  const long long blocks2 = (long long)B * h * wd * (C / 8);
  const char* sw = getenv("PT_DWCONVT2");       // PT_DWCONVT2=0: the general kernel also for f = 2 (A/B switch, read per call)
  if (f == 2 && blocks2 < (1ll << 31) && C <= 512 && !(sw && sw[0] == '0')) {
    static int packed = -1;                      // PT_DWCONVT2_PACKED=0: the float-register variant also for plain bf16 maps (A/B switch)
    if (packed < 0) {
      const char* ev = getenv("PT_DWCONVT2_PACKED");
      packed = ev ? atoi(ev) : 1;
    }
    if (!split && packed)
      hipLaunchKernelGGL(dwconvt_up2_add_kernel<1>, dim3(grid_for(blocks2)), dim3(256), (size_t)16 * C * sizeof(float), s, in, w, add, out,
                         B, h, wd, C, 0);
    else if (split && packed)
      hipLaunchKernelGGL(dwconvt_up2_add_kernel<2>, dim3(grid_for(blocks2)), dim3(256), (size_t)16 * C * sizeof(float), s, in, w, add, out,
                         B, h, wd, C, 1);
    else
      hipLaunchKernelGGL(dwconvt_up2_add_kernel<0>, dim3(grid_for(blocks2)), dim3(256), (size_t)16 * C * sizeof(float), s, in, w, add, out,
                         B, h, wd, C, split);
    PT_HIP_CHECK(hipGetLastError());
    return PT_OK;
  }
  hipLaunchKernelGGL(dwconvt_up_add_kernel, dim3(grid_for((long long)B * h * f * wd * f * (C / 8))), dim3(256), 0, s, in,
                     w, add, out, B, h, wd, C, f, split);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_launch_tsr_preprocess(const uint8_t* pages, int ph, int pw, const pt_tsr_table* tabs, int n, int H, int W, int bgr,
                             const float* lut, bf16_t* out, int split, hipStream_t s) {
  PT_REQUIRE(pages && tabs && lut && out && n > 0 && H > 0 && W > 0, "tsr preprocess: bad arguments");
  hipLaunchKernelGGL(tsr_preprocess_kernel, dim3((H * W + 255) / 256, n), dim3(256), 0, s, pages, ph, pw, tabs, H, W, bgr, lut,
                     out, split);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

template <int SPLIT, int NB>
static int launch_dcn_fused(pt_engine* e, const bf16_t* x, const float* om, const bf16_t* w, const float* bias, bf16_t* out,
                            int B, int H, int W, int C, int N, int relu, hipStream_t s) {
  const long long npix = (long long)B * H * W;
  char label[48];
  snprintf(label, sizeof(label), "dcn fused %d->%d @%dx%d%s", C, N, H, W, SPLIT ? " x3" : "");
  PtProfScope prof(e, s, PT_PROF_OTHER, 0, label);   // gather-bound, kept out of the implicit-GEMM class
  hipLaunchKernelGGL((dcn_fused_kernel<SPLIT, NB>), dim3((unsigned)((npix + 127) / 128), N / NB), dim3(256), 0, s, x, om, w,
                     bias, out, npix, H, W, C, N, relu);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// x NHWC bf16 [B,H,W,C], om fp32 [pixel][32], w tiled like a 1x1 conv over K = 9*C (".dcn" tensors), out [B,H,W,N]
// omw / omb != null (pt_dcn_fuses_om says when): the offset / mask convolution runs in the kernel's prologue from the layer's `.om` tiles and `om` is not read
bool pt_dcn_fuses_om(const pt_engine* e, int C, int split) {
  // PT_DCN_FUSE_OM: 0 = the offset conv as its own launch everywhere; 1 (default) = inside the DCN kernel where that measured faster -- single-pass modes,
  // C <= 128: the 64 -> 64 @256^2 layers 10.68 -> 9.59 ms per step, the C = 128 layers 5.95 -> 5.63 (profiles/r05/dcn_offset_conv_fusion.txt); at C >= 256
  // the prologue's 8+ chunk barriers cost what the launch did, and in the pair modes (one or two workgroups per CU: nobody to hide behind) it measured
  // equal; 2 = everywhere it can run (tests).  Read at every call: tests flip it.
  const char* ev = getenv("PT_DCN_FUSE_OM");
  const int on = ev ? atoi(ev) : 1;
  const char* nt = getenv("PT_DCN_THREADS");
  const char* ea = getenv("PT_DCN_EARLY");
  const char* xf = getenv("PT_DCN_X3_FAST");
  const bool defaults = !(nt && atoi(nt) != 512) && !(ea && atoi(ea) == 0) && !(xf && atoi(xf) == 0);
  if (!on || !e || C % 64 != 0 || !defaults || (!split && e->dcn_mfma != 0)) return false;
  return on == 2 || (!split && C <= 128);
}

int pt_launch_dcn_fused(pt_engine* e, const bf16_t* x, const float* om, const bf16_t* w, const float* bias, bf16_t* out,
                        int B, int H, int W, int C, int N, int split, int relu, hipStream_t s, const bf16_t* omw, const float* omb) {
  PT_REQUIRE(e && x && (om || (omw && omb)) && w && bias && out && C % 32 == 0 && N % 64 == 0, "dcn fused: bad arguments (C=%d N=%d)", C, N);      // e: every path below reads it
  PT_REQUIRE(!omw || pt_dcn_fuses_om(e, C, split), "dcn fused: the offset conv can only be fused on the default fused64 paths (C=%d)", C);
  PT_REQUIRE((long long)H * W * (split ? 2 * C : C) < (1ll << 31), "dcn fused: image too large for 32-bit offsets");
  // algorithmic bytes: input map once, offsets / masks once, output once, weights once (the 36 corner lines per pixel are cache traffic)
  const double dcn_bytes = (double)B * H * W * ((split ? 2.0 : 1.0) * 2.0 * (C + N) + 27 * 4.0) + (double)N * 9 * C * 2.0 * (split ? 3 : 1);
  e->prof.next_bytes = dcn_bytes;
  // (the hi/lo mode keeps 64-channel blocks: with 128 the corner registers of both halves spill)
  static int x3fast = -1;        // PT_DCN_X3_FAST=0: the hi/lo mode on dcn_fused_kernel<1, 64> (A/B switch)
  if (x3fast < 0) {
    const char* ev = getenv("PT_DCN_X3_FAST");
    x3fast = ev ? atoi(ev) : 1;
  }
  if (split && C % 64 == 0 && x3fast) {
    const long long npix = (long long)B * H * W;
    char label[48];
    snprintf(label, sizeof(label), "dcn fused %d->%d @%dx%d x3", C, N, H, W);
    PtProfScope prof(e, s, PT_PROF_OTHER, 0, label);
    const unsigned tiles = (unsigned)((long long)B * ((H + 7) / 8) * ((W + 15) / 16));
    // N >= 128: one workgroup computes 128 output channels from one gather + blend, as in bf16 mode (PT_DCN_NB=64: 64-wide blocks everywhere).
    // 110 KB of LDS for both operand planes: one workgroup of eight waves per CU, which the 64-wide hi/lo blocks (92 KB) are too
    const char* nbv = getenv("PT_DCN_NB");
    if (N % 128 == 0 && !(nbv && atoi(nbv) == 64)) {
      if (omw) hipLaunchKernelGGL((dcn_fused64_kernel<128, 512, 1, 0, 1>), dim3(tiles, N / 128), dim3(512), 0, s, x, om, w, bias, out, npix, H, W, C, N, relu, omw, omb);
      else hipLaunchKernelGGL((dcn_fused64_kernel<128, 512, 1>), dim3(tiles, N / 128), dim3(512), 0, s, x, om, w, bias, out, npix, H, W, C, N, relu, omw, omb);
    } else {
      if (omw) hipLaunchKernelGGL((dcn_fused64_kernel<64, 512, 1, 0, 1>), dim3(tiles, N / 64), dim3(512), 0, s, x, om, w, bias, out, npix, H, W, C, N, relu, omw, omb);
      else hipLaunchKernelGGL((dcn_fused64_kernel<64, 512, 1>), dim3(tiles, N / 64), dim3(512), 0, s, x, om, w, bias, out, npix, H, W, C, N, relu, omw, omb);
    }
    PT_HIP_CHECK(hipGetLastError());
    return PT_OK;
  }
  if (split) { e->prof.next_bytes = dcn_bytes; return launch_dcn_fused<1, 64>(e, x, om, w, bias, out, B, H, W, C, N, relu, s); }
  if (C % 64 == 0) {
    const long long npix = (long long)B * H * W;
    char label[48];
    snprintf(label, sizeof(label), "dcn fused %d->%d @%dx%d", C, N, H, W);
    PtProfScope prof(e, s, PT_PROF_OTHER, 0, label);   // gather-bound, kept out of the implicit-GEMM class
    // N >= 128: one workgroup computes 128 output channels from one gather + blend (the expensive half of the kernel)
    // instead of two workgroups repeating it for 64 each (PT_DCN_NB=64: the 64-wide blocks everywhere)
    static int nb128 = -1;
    if (nb128 < 0) {
      const char* ev = getenv("PT_DCN_NB");
      nb128 = ev ? (atoi(ev) == 128) : 1;
    }
    static int nthr = -1;          // PT_DCN_THREADS=256: four waves per workgroup (A/B switch)
    if (nthr < 0) {
      const char* ev = getenv("PT_DCN_THREADS");
      nthr = ev ? atoi(ev) : 512;
    }
    const unsigned tiles = (unsigned)((long long)B * ((H + 7) / 8) * ((W + 15) / 16));
    // pt_engine_set_dcn_mfma (default off, PT_DCN_MFMA): the blend on the matrix pipe (dcn_mfma_kernel); off: dcn_fused64_kernel (VALU blend,
    // fp32 weights); PT_DCN_MFMA_NB=64: 64-wide blocks for every layer
    static int mfma_nb = -1;
    if (mfma_nb < 0) {
      const char* nv = getenv("PT_DCN_MFMA_NB");
      PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcn_mfma_kernel<64, PT_DCN_RING64>), hipFuncAttributeMaxDynamicSharedMemorySize, DcnMfmaSmem<64, PT_DCN_RING64>::BYTES));
      PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcn_mfma_kernel<128, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, DcnMfmaSmem<128, 8>::BYTES));
      mfma_nb = nv ? atoi(nv) : 128;
    }
    const int mfma = e->dcn_mfma;
    if (mfma == 2 && N == 64) {      // full-line gathers, one workgroup per CU (dcn_mfma2_kernel); wider layers keep dcn_fused64_kernel in this setting
      static bool attr2 = false;
      if (!attr2) {
        PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcn_mfma2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DcnMfma2Smem::BYTES));
        attr2 = true;
      }
      const unsigned tiles2 = (unsigned)((long long)B * ((H + 15) / 16) * ((W + 15) / 16));
      hipLaunchKernelGGL(dcn_mfma2_kernel, dim3(tiles2, 1), dim3(512), DcnMfma2Smem::BYTES, s, x, om, w, bias, out, npix, H, W, C, N, relu);
      PT_HIP_CHECK(hipGetLastError());
      return PT_OK;
    }
    if (mfma == 1) {
      constexpr int smem64 = DcnMfmaSmem<64, PT_DCN_RING64>::BYTES, smem128 = DcnMfmaSmem<128, 8>::BYTES;
      if (N % 128 == 0 && mfma_nb == 128)
        hipLaunchKernelGGL((dcn_mfma_kernel<128, 8>), dim3(tiles, N / 128), dim3(512), smem128, s, x, om, w, bias, out, npix, H, W, C, N, relu);
      else
        hipLaunchKernelGGL((dcn_mfma_kernel<64, PT_DCN_RING64>), dim3(tiles, N / 64), dim3(512), smem64, s, x, om, w, bias, out, npix, H, W, C, N, relu);
      PT_HIP_CHECK(hipGetLastError());
      return PT_OK;
    }
    if (nb128 && N % 128 == 0) {     // 128-wide blocks stay at 4 waves: with 8 they need 136 VGPRs (> 128: spills), measured 1 % slower

      if (omw) hipLaunchKernelGGL((dcn_fused64_kernel<128, 256, 0, 0, 1>), dim3(tiles, N / 128), dim3(256), 0, s, x, om, w, bias, out, npix, H, W, C, N, relu, omw, omb);
      else hipLaunchKernelGGL((dcn_fused64_kernel<128, 256>), dim3(tiles, N / 128), dim3(256), 0, s, x, om, w, bias, out, npix, H, W, C, N, relu, omw, omb);
    } else if (nthr == 512) {
      static int early = -1;          // PT_DCN_EARLY=0: next-stage loads after the second barrier (A/B switch)
      if (early < 0) {
        const char* ev = getenv("PT_DCN_EARLY");
        early = ev ? atoi(ev) : 1;
      }
      if (early && omw) hipLaunchKernelGGL((dcn_fused64_kernel<64, 512, 0, 1, 1>), dim3(tiles, N / 64), dim3(512), 0, s, x, om, w, bias, out, npix, H, W, C, N, relu, omw, omb);
      else if (early) hipLaunchKernelGGL((dcn_fused64_kernel<64, 512, 0, 1>), dim3(tiles, N / 64), dim3(512), 0, s, x, om, w, bias, out, npix, H, W, C, N, relu, omw, omb);
      else hipLaunchKernelGGL((dcn_fused64_kernel<64, 512>), dim3(tiles, N / 64), dim3(512), 0, s, x, om, w, bias, out, npix, H, W, C, N, relu, omw, omb);
    } else {
      hipLaunchKernelGGL((dcn_fused64_kernel<64, 256>), dim3(tiles, N / 64), dim3(256), 0, s, x, om, w, bias, out, npix, H, W, C, N, relu, omw, omb);
    }
    PT_HIP_CHECK(hipGetLastError());
    return PT_OK;
  }
  if (N % 128 == 0) return launch_dcn_fused<0, 128>(e, x, om, w, bias, out, B, H, W, C, N, relu, s);
  return launch_dcn_fused<0, 64>(e, x, om, w, bias, out, B, H, W, C, N, relu, s);
}

// in [B,H,W,16] bf16 ([hi16|lo16] when split) -> out [B,Ho,Wo,N] (N = 16 or 32), 3x3 pad 1, stride 1 or 2, bias + ReLU
int pt_launch_conv3x3_c16(pt_engine* e, const bf16_t* in, const bf16_t* w, const float* bias, bf16_t* out, int B, int H, int W,
                          int N, int stride, int split, hipStream_t s) {
  PT_REQUIRE(in && w && bias && out && (N == 16 || N == 32) && (stride == 1 || stride == 2), "conv3x3_c16: bad arguments");
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const int th = stride == 1 ? 8 : 4;
  const int tw = (stride == 2 && split) ? 32 : 64;
  const int tiles_x = (Wo + tw - 1) / tw, tiles_y = (Ho + th - 1) / th;
  const long long nblk = (long long)B * tiles_x * tiles_y;
  PT_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv3x3_c16: grid out of range");
  char label[48];
  snprintf(label, sizeof(label), "conv3x3 c16 s%d 16->%d @%dx%d%s", stride, N, Ho, Wo, split ? " x3" : "");
  PtProfScope prof(e, s, PT_PROF_STEM, 2.0 * B * Ho * Wo * (double)N * 144, label);
  if (stride == 1) {
    if (split) hipLaunchKernelGGL((conv3x3_c16_kernel<1, 1>), dim3((unsigned)nblk), dim3(256), 0, s, in, w, bias, out, B, H, W, Ho, Wo, N, tiles_x, tiles_y);
    else hipLaunchKernelGGL((conv3x3_c16_kernel<1, 0>), dim3((unsigned)nblk), dim3(256), 0, s, in, w, bias, out, B, H, W, Ho, Wo, N, tiles_x, tiles_y);
  } else {
    if (split) hipLaunchKernelGGL((conv3x3_c16_kernel<2, 1>), dim3((unsigned)nblk), dim3(256), 0, s, in, w, bias, out, B, H, W, Ho, Wo, N, tiles_x, tiles_y);
    else hipLaunchKernelGGL((conv3x3_c16_kernel<2, 0>), dim3((unsigned)nblk), dim3(256), 0, s, in, w, bias, out, B, H, W, Ho, Wo, N, tiles_x, tiles_y);
  }
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// DLA-34 base_layer -> level0 -> level1 in one launch (bf16 mode): in NHWC4 [B,H,W,4] -> out [B,H/2,W/2,32]
int pt_launch_dla_thin_chain(pt_engine* e, const bf16_t* in, int B, int H, int W, const bf16_t* w_stem, const float* b_stem, const bf16_t* w0,
                             const float* b0, const bf16_t* w1, const float* b1, bf16_t* out, hipStream_t s) {
  PT_REQUIRE(in && w_stem && b_stem && w0 && b0 && w1 && b1 && out && B > 0, "dla thin chain: bad arguments");
  PT_REQUIRE(H % 2 == 0 && W % 2 == 0 && H >= 2 && W >= 2, "dla thin chain: H, W must be even (got %d x %d)", H, W);
  using C = ThinChainCfg;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dla_thin_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM));
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dla_thin_chain16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM));
    attr_done = true;
  }
  const int tiles_x = (W / 2 + C::TW1 - 1) / C::TW1, tiles_y = (H / 2 + C::TH1 - 1) / C::TH1;
  const long long nblk = (long long)B * tiles_y;
  PT_REQUIRE(nblk < (1ll << 31), "dla thin chain: grid out of range");
  const double px = (double)B * H * W;
  e->prof.next_bytes = px * 8.0 + px / 4 * 64.0;
  PtProfScope prof(e, s, PT_PROF_STEM, 2.0 * px * (16.0 * 147 + 16.0 * 144) + 2.0 * px / 4 * 32.0 * 144, "dla thin chain (stem + level0 + level1)");
  const char* cv = getenv("PT_DLA_CHAIN");      // 1: the 32 x 32 x 16 kernel (bit-identical to the three launches); default: the 16 x 16 x 32 one (read per call)
  if (cv && atoi(cv) == 1)
    hipLaunchKernelGGL(dla_thin_chain_kernel, dim3((unsigned)nblk), dim3(512), C::SMEM, s, in, w_stem, b_stem, w0, b0, w1, b1, out, B, H, W, tiles_x, tiles_y);
  else
    hipLaunchKernelGGL(dla_thin_chain16_kernel, dim3((unsigned)nblk), dim3(512), C::SMEM, s, in, w_stem, b_stem, w0, b0, w1, b1, out, B, H, W, tiles_x, tiles_y);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

}  // namespace PT_FMT_NS
