// det_kernels.hip -- HBM-bound kernels around the DB detector: page pre-process (resize + normalise),
// 3x3/s2 max-pool, the 64->1 transposed-conv + sigmoid head, prob->bitmap, and box_score_fast.
// Compiled with -ffp-contract=off: the float sequences below restate numpy/OpenCV op-by-op.
#include <math.h>

#include "common.h"

namespace PT_FMT_NS {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// 16 stored bits <-> fp32 in the storage format of this namespace (act16.h: bf16, or IEEE half in pt_f16)
__device__ __forceinline__ float bf2f(uint32_t bits16) { return a16_to_f32(bits16); }
__device__ __forceinline__ uint32_t f2bf(float f) { return f32_to_a16(f); }

// ---------------------------------------------------------------------------------------------------
// Pre-process.  Restates, per output pixel:
//   img[:, :, ::-1]                                    db_pp/processor_ocr_db_pp.py:124
//   cv2.resize(img, (nw, nh))  (INTER_LINEAR, 8-bit)   db_pp/image_operators.py:310, db_net/processor_ocr_dbnet.py:58
//   (x * (1/255) - mean) / std   [db_pp]               db_pp/image_operators.py:100-101
//   (x - mean) / 255             [db torch]            db_net/processor_ocr_dbnet.py:62-65
// OpenCV's 8-bit bilinear path is fixed point: 11-bit coefficients, horizontal pass in int32, vertical
// pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.  Exact 2x decimation switches to INTER_AREA.
// ---------------------------------------------------------------------------------------------------
struct ResizeCoef {
  int s0, s1;
  int a0, a1;
};
__device__ __forceinline__ ResizeCoef resize_coef(int d, double scale, int ssize, bool clamp_frac) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  ResizeCoef c;
  if (clamp_frac) {  // horizontal direction: OpenCV pins fx at the borders
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  }
  c.a0 = (int)rintf((1.f - f) * 2048.f);
  c.a1 = (int)rintf(f * 2048.f);
  int t0 = s, t1 = s + 1;
  t0 = t0 < 0 ? 0 : (t0 >= ssize ? ssize - 1 : t0);
  t1 = t1 < 0 ? 0 : (t1 >= ssize ? ssize - 1 : t1);
  c.s0 = t0;
  c.s1 = t1;
  return c;
}

constexpr int PRE_ROWS = 8;      // output rows per workgroup

__global__ __launch_bounds__(256) void det_preprocess_kernel(const uint8_t* __restrict__ pages, int n, int h, int w,
                                                              int nh, int nw, int flavour, int split, bf16_t* __restrict__ out,
                                                              double sx, double sy) {
  a16_kernel_enter();
  // grid (column blocks, groups of PRE_ROWS output rows, page): no 64-bit index division per pixel, a thread keeps its column --
  // the horizontal coefficients are computed once for PRE_ROWS pixels.  sx = (double)w / nw, sy = (double)h / nh come from the host
  // (the same IEEE division).  The normalisation is a function of one byte: a [3][256] table of (bf16 hi | bf16 lo << 16) is built per
  // workgroup with the per-pixel float sequence (-ffp-contract=off: the same bits) -- the fp32 division and the integer rounding were
  // a third of the kernel's instructions, and the kernel is VALU-bound (190 instructions per 11 bytes moved).
  __shared__ uint32_t lut[3][256];
  {
    const int u = threadIdx.x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float o;
      if (flavour == PT_DET_PRE_DB_TORCH) {
        const float mean[3] = {123.68f, 116.78f, 103.94f};
        o = ((float)u - mean[c]) / 255.f;
      } else {
        const float scale = (float)(1.0 / 255.0);
        const float mean[3] = {0.485f, 0.456f, 0.406f};
        const float stdv[3] = {0.229f, 0.224f, 0.225f};
        o = ((float)u * scale - mean[c]) / stdv[c];
      }
      const uint32_t hi = f2bf(o);
      lut[c][u] = hi | (f2bf(o - bf2f(hi)) << 16);
    }
  }
  __syncthreads();
  const bool area2 = (w == 2 * nw) && (h == 2 * nh);
  const int b = blockIdx.z;
  const uint8_t* src = pages + (size_t)b * h * w * 3;
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < nw; x += gridDim.x * blockDim.x) {
    const ResizeCoef cx = resize_coef(x, sx, w, true);
    const int o0 = cx.s0 * 3, d1 = (cx.s1 - cx.s0) * 24;
    for (int ry = 0; ry < PRE_ROWS; ++ry) {
      const int y = blockIdx.y * PRE_ROWS + ry;
      if (y >= nh) break;
      const long long i = ((long long)b * nh + y) * nw + x;
      int v[3];
      if (w == nw && h == nh) {
        const uint8_t* p = src + ((size_t)y * w + x) * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
      } else if (area2) {
        const uint8_t* p0 = src + ((size_t)(2 * y) * w + 2 * x) * 3;
        const uint8_t* p1 = p0 + (size_t)w * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2;
      } else {
        const ResizeCoef cy = resize_coef(y, sy, h, false);
        const uint8_t* r0 = src + (size_t)cy.s0 * w * 3;
        const uint8_t* r1 = src + (size_t)cy.s1 * w * 3;
        // the two source pixels of a row are adjacent (or the same one at the clamped borders): ONE unaligned 8-byte load per row covers
        // their 6 bytes; the last two pixels of a row take byte loads (the window would end past the row -- past the buffer on the last one)
        unsigned long long q0, q1;
        if (o0 + 8 <= w * 3) {
          __builtin_memcpy(&q0, r0 + o0, 8);
          __builtin_memcpy(&q1, r1 + o0, 8);
        } else {
          q0 = q1 = 0;
          for (int k = 0; k < 3; ++k) {
            q0 |= (unsigned long long)r0[o0 + k] << (8 * k) | (unsigned long long)r0[cx.s1 * 3 + k] << (d1 + 8 * k);
            q1 |= (unsigned long long)r1[o0 + k] << (8 * k) | (unsigned long long)r1[cx.s1 * 3 + k] << (d1 + 8 * k);
          }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int S0 = (int)((q0 >> (8 * c)) & 0xFF) * cx.a0 + (int)((q0 >> (d1 + 8 * c)) & 0xFF) * cx.a1;
          const int S1 = (int)((q1 >> (8 * c)) & 0xFF) * cx.a0 + (int)((q1 >> (d1 + 8 * c)) & 0xFF) * cx.a1;
          v[c] = (((cy.a0 * (S0 >> 4)) >> 16) + ((cy.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
          v[c] = v[c] < 0 ? 0 : (v[c] > 255 ? 255 : v[c]);
        }
      }
      // channel c of the output is channel 2 - c of the page (the BGR flip), through the table
      const uint32_t e0 = lut[0][v[2]], e1 = lut[1][v[1]], e2 = lut[2][v[0]];
      if (!split) {
        u32x2 pk;
        pk.x = (e0 & 0xFFFFu) | (e1 << 16);
        pk.y = e2 & 0xFFFFu;
        *reinterpret_cast<u32x2*>(out + (size_t)i * 4) = pk;
      } else {  // (hi | lo) pairs for the bf16x3 precision mode: lo = bf16(x - hi)
        u32x4 pk;
        pk.x = (e0 & 0xFFFFu) | (e1 << 16);
        pk.y = e2 & 0xFFFFu;
        pk.z = (e0 >> 16) | (e1 & 0xFFFF0000u);
        pk.w = e2 >> 16;
        *reinterpret_cast<u32x4*>(out + (size_t)i * 8) = pk;
      }
    }
  }
}

int pt_launch_det_preprocess(const uint8_t* pages, int n, int h, int w, int nh, int nw, int flavour, int split,
                             bf16_t* out, hipStream_t s) {
  PT_REQUIRE(n > 0 && nh > 0 && nw > 0 && nh < 65536 && n < 65536, "det pre-process: bad extents %d x %d x %d", n, nh, nw);
  hipLaunchKernelGGL(det_preprocess_kernel, dim3((nw + 255) / 256, (nh + PRE_ROWS - 1) / PRE_ROWS, n), dim3(256), 0, s, pages, n, h, w, nh, nw, flavour, split, out,
                     (double)w / nw, (double)h / nh);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, padding 1) on NHWC bf16 (dbnet.py:276).  8 channels (16 B) per thread.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bfmax2(uint32_t a, uint32_t b) {
  const float al = a16lo_f32(a), bl = a16lo_f32(b);
  const float ah = a16hi_f32(a), bh = a16hi_f32(b);
  const uint32_t lo = (bl > al) ? (b & 0xFFFFu) : (a & 0xFFFFu);
  const uint32_t hi = (bh > ah) ? (b & 0xFFFF0000u) : (a & 0xFFFF0000u);
  return lo | hi;
}

// split mode: pixels hold [hi(C) | lo(C)]; the max is taken on hi + lo and the winning PAIR is copied
__global__ __launch_bounds__(256) void maxpool3x3s2_split_kernel(const bf16_t* __restrict__ in, int B, int H, int W,
                                                                  int C, bf16_t* __restrict__ out) {
  a16_kernel_enter();
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int cg = C >> 3;
  const long long total = (long long)B * Ho * Wo * cg;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    long long t = i / cg;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    uint32_t bh[8], bl[8];
    float bv[8];
    bool first = true;
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = oy * 2 - 1 + dy;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = ox * 2 - 1 + dx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const bf16_t* px = in + (((size_t)b * H + iy) * W + ix) * (2 * C) + g * 8;
        const u32x4 vh = *reinterpret_cast<const u32x4*>(px);
        const u32x4 vl = *reinterpret_cast<const u32x4*>(px + C);
        const uint32_t hw[4] = {vh.x, vh.y, vh.z, vh.w}, lw[4] = {vl.x, vl.y, vl.z, vl.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t hb = (k & 1) ? (hw[k >> 1] >> 16) : (hw[k >> 1] & 0xFFFFu);
          const uint32_t lb = (k & 1) ? (lw[k >> 1] >> 16) : (lw[k >> 1] & 0xFFFFu);
          const float v = bf2f(hb) + bf2f(lb);
          if (first || v > bv[k]) { bv[k] = v; bh[k] = hb; bl[k] = lb; }
        }
        first = false;
      }
    }
    u32x4 oh, ol;
    oh.x = bh[0] | (bh[1] << 16); oh.y = bh[2] | (bh[3] << 16); oh.z = bh[4] | (bh[5] << 16); oh.w = bh[6] | (bh[7] << 16);
    ol.x = bl[0] | (bl[1] << 16); ol.y = bl[2] | (bl[3] << 16); ol.z = bl[4] | (bl[5] << 16); ol.w = bl[6] | (bl[7] << 16);
    bf16_t* po = out + (((size_t)b * Ho + oy) * Wo + ox) * (2 * C) + g * 8;
    *reinterpret_cast<u32x4*>(po) = oh;
    *reinterpret_cast<u32x4*>(po + C) = ol;
  }
}

__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const bf16_t* __restrict__ in, int B, int H, int W, int C,
                                                            bf16_t* __restrict__ out) {
  a16_kernel_enter();
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int cg = C >> 3;
  const long long total = (long long)B * Ho * Wo * cg;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    long long t = i / cg;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    u32x4 m;
    bool first = true;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = oy * 2 - 1 + dy;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = ox * 2 - 1 + dx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(in + (((size_t)b * H + iy) * W + ix) * C + g * 8);
        if (first) {
          m = v;
          first = false;
        } else {
          m.x = bfmax2(m.x, v.x); m.y = bfmax2(m.y, v.y); m.z = bfmax2(m.z, v.z); m.w = bfmax2(m.w, v.w);
        }
      }
    }
    *reinterpret_cast<u32x4*>(out + (size_t)i * 8) = m;
  }
}

int pt_launch_maxpool3x3s2(const bf16_t* in, int B, int H, int W, int C, bf16_t* out, int split, hipStream_t s) {
  PT_REQUIRE(C % 8 == 0, "maxpool: C must be a multiple of 8");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)B * Ho * Wo * (C / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (split)
    hipLaunchKernelGGL(maxpool3x3s2_split_kernel, dim3(blocks), dim3(256), 0, s, in, B, H, W, C, out);
  else
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(blocks), dim3(256), 0, s, in, B, H, W, C, out);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// DB head tail: ConvTranspose2d(64 -> 1, k=2, s=2) + Sigmoid (dbnet.py:539).
// in [B,H,W,64] bf16 -> prob fp32 [B,2H,2W].  8 lanes share one input pixel (16 B each, coalesced);
// partial dot products are combined with a 3-step xor-shuffle; lanes 0..3 of the group store the
// 2x2 output patch.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void db_head_final_kernel(const bf16_t* __restrict__ in, int B, int H, int W,
                                                             const bf16_t* __restrict__ w4x64, const float* __restrict__ bias_p,
                                                             float* __restrict__ prob, float* __restrict__ logits) {
  a16_kernel_enter();
  const int sub = threadIdx.x & 7;
  const float bias = bias_p[0];
  float wq[4][8];
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const u32x4 wv = *reinterpret_cast<const u32x4*>(w4x64 + qd * 64 + sub * 8);
    wq[qd][0] = bf2f(wv.x & 0xFFFFu); wq[qd][1] = bf2f(wv.x >> 16);
    wq[qd][2] = bf2f(wv.y & 0xFFFFu); wq[qd][3] = bf2f(wv.y >> 16);
    wq[qd][4] = bf2f(wv.z & 0xFFFFu); wq[qd][5] = bf2f(wv.z >> 16);
    wq[qd][6] = bf2f(wv.w & 0xFFFFu); wq[qd][7] = bf2f(wv.w >> 16);
  }
  const long long npix = (long long)B * H * W;
  const long long gstride = ((long long)gridDim.x * blockDim.x) >> 3;
  for (long long pix = (((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3); pix < npix; pix += gstride) {
    const u32x4 xv = *reinterpret_cast<const u32x4*>(in + (size_t)pix * 64 + sub * 8);
    float x[8] = {bf2f(xv.x & 0xFFFFu), bf2f(xv.x >> 16), bf2f(xv.y & 0xFFFFu), bf2f(xv.y >> 16),
                  bf2f(xv.z & 0xFFFFu), bf2f(xv.z >> 16), bf2f(xv.w & 0xFFFFu), bf2f(xv.w >> 16)};
    float acc[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) a = fmaf(x[k], wq[qd][k], a);
      acc[qd] = a;
    }
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      acc[qd] += __shfl_xor(acc[qd], 1);
      acc[qd] += __shfl_xor(acc[qd], 2);
      acc[qd] += __shfl_xor(acc[qd], 4);
    }
    if (sub < 4) {
      const float lg = (sub == 0 ? acc[0] : sub == 1 ? acc[1] : sub == 2 ? acc[2] : acc[3]) + bias;
      const int xx = (int)(pix % W);
      const long long t = pix / W;
      const int yy = (int)(t % H);
      const int b = (int)(t / H);
      const size_t o = ((size_t)b * (2 * H) + 2 * yy + (sub >> 1)) * (size_t)(2 * W) + 2 * xx + (sub & 1);
      if (logits) logits[o] = lg;
      if (prob) prob[o] = 1.f / (1.f + expf(-lg));
    }
  }
}

// bf16x3 precision mode: in [B,H,W,128] = (hi | lo), weights fp32 [4][64]
__global__ __launch_bounds__(256) void db_head_final_split_kernel(const bf16_t* __restrict__ in, int B, int H, int W,
                                                                   const float* __restrict__ w4x64,
                                                                   const float* __restrict__ bias_p,
                                                                   float* __restrict__ prob, float* __restrict__ logits) {
  a16_kernel_enter();
  const int sub = threadIdx.x & 7;
  const float bias = bias_p[0];
  float wq[4][8];
#pragma unroll
  for (int qd = 0; qd < 4; ++qd)
#pragma unroll
    for (int k = 0; k < 8; ++k) wq[qd][k] = w4x64[qd * 64 + sub * 8 + k];
  const long long npix = (long long)B * H * W;
  const long long gstride = ((long long)gridDim.x * blockDim.x) >> 3;
  for (long long pix = (((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3); pix < npix; pix += gstride) {
    const u32x4 xh = *reinterpret_cast<const u32x4*>(in + (size_t)pix * 128 + sub * 8);
    const u32x4 xl = *reinterpret_cast<const u32x4*>(in + (size_t)pix * 128 + 64 + sub * 8);
    float x[8] = {bf2f(xh.x & 0xFFFFu) + bf2f(xl.x & 0xFFFFu), bf2f(xh.x >> 16) + bf2f(xl.x >> 16),
                  bf2f(xh.y & 0xFFFFu) + bf2f(xl.y & 0xFFFFu), bf2f(xh.y >> 16) + bf2f(xl.y >> 16),
                  bf2f(xh.z & 0xFFFFu) + bf2f(xl.z & 0xFFFFu), bf2f(xh.z >> 16) + bf2f(xl.z >> 16),
                  bf2f(xh.w & 0xFFFFu) + bf2f(xl.w & 0xFFFFu), bf2f(xh.w >> 16) + bf2f(xl.w >> 16)};
    float acc[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) a = fmaf(x[k], wq[qd][k], a);
      acc[qd] = a;
    }
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      acc[qd] += __shfl_xor(acc[qd], 1);
      acc[qd] += __shfl_xor(acc[qd], 2);
      acc[qd] += __shfl_xor(acc[qd], 4);
    }
    if (sub < 4) {
      const float lg = (sub == 0 ? acc[0] : sub == 1 ? acc[1] : sub == 2 ? acc[2] : acc[3]) + bias;
      const int xx = (int)(pix % W);
      const long long t = pix / W;
      const int yy = (int)(t % H);
      const int b = (int)(t / H);
      const size_t o = ((size_t)b * (2 * H) + 2 * yy + (sub >> 1)) * (size_t)(2 * W) + 2 * xx + (sub & 1);
      if (logits) logits[o] = lg;
      if (prob) prob[o] = 1.f / (1.f + expf(-lg));
    }
  }
}

int pt_launch_db_head_final(const bf16_t* in, int B, int H, int W, const void* w4x64, const float* bias, float* prob,
                            float* logits, int split, hipStream_t s) {
  const long long nthreads = (long long)B * H * W * 8;
  int blocks = (int)((nthreads + 255) / 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (split)
    hipLaunchKernelGGL(db_head_final_split_kernel, dim3(blocks), dim3(256), 0, s, in, B, H, W,
                       reinterpret_cast<const float*>(w4x64), bias, prob, logits);
  else
    hipLaunchKernelGGL(db_head_final_kernel, dim3(blocks), dim3(256), 0, s, in, B, H, W,
                       reinterpret_cast<const bf16_t*>(w4x64), bias, prob, logits);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// DB head in one streaming kernel (bf16 mode): ConvTranspose2d(64,64,2,2)+BN+ReLU and ConvTranspose2d(64,1,2,2)+Sigmoid
// (dbnet.py:537-539) from the 64-channel map at 1/4 resolution straight to the full-resolution probability map.
// The implicit-GEMM kernel ran this as four 64-column tiles with K = 64 -- two K-chunks of MFMA work per workgroup around
// a full prologue / fp32-LDS epilogue, and 4-byte scattered stores: 0.6 ms per 32 pages for 355 MB of traffic.  Here a wave
// owns 32 pixels at a time and never leaves its registers:
//   GEMM 1  D^T[256 ch][32 px] = W3^T (A operand, from LDS) x X^T (B operand, 16-byte loads straight from HBM):
//           a lane owns ONE pixel, its accumulators are that pixel's channels;
//   bias + ReLU + bf16 rounding in the lane (the rounding the stored 64-channel map would have had);
//   GEMM 2  per quadrant: D2[4 sub-pixels (of 32 rows)][32 px] = W6^T x h -- the accumulator registers of GEMM 1 ARE the B
//           operand (the K order of a dot product is free: W6 is permuted to the accumulator order once);
//   a pixel's 4x4 output block leaves as four 16-byte stores, 512 B contiguous per wave and output row.
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 hbf16x8;
typedef __attribute__((ext_vector_type(2))) float hcf2;
typedef __attribute__((ext_vector_type(2))) __bf16 hcb2;
// two fp32 -> one dword of two stored values, round-to-nearest-even (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32; equal to f2bf for finite values)
__device__ __forceinline__ uint32_t pack2bf(float a, float b) { return pack_a16x2(a, b); }
typedef __attribute__((ext_vector_type(16))) float hf32x16;
typedef __attribute__((ext_vector_type(4))) float hf32x4;

__global__ __launch_bounds__(256, 4) void db_head_mfma_kernel(const bf16_t* __restrict__ in, long long npix, int H, int W,
                                                               const bf16_t* __restrict__ w3, const float* __restrict__ b3,
                                                               const bf16_t* __restrict__ w6, const float* __restrict__ b6,
                                                               float* __restrict__ prob, float* __restrict__ logits,
                                                               uint32_t* __restrict__ bitmap, float thresh) {
  a16_kernel_enter();
  constexpr int PITCH = 144;                       // 64 k x 2 B + 16 B pad: conflict-free ds_read_b128 over 32 rows
  __shared__ __attribute__((aligned(16))) char s_w[256 * PITCH];
  __shared__ float s_b[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  for (int idx = tid; idx < 256 * 8; idx += 256) {  // W3: packed [4 tiles of 64][2 chunks of 32 k][64 rows][32 k] -> [256 rows][64 k]
    const int n = idx >> 3, piece = idx & 7;        // piece: 8 k-values
    const int k = piece * 8;
    const size_t src = ((size_t)((n >> 6) * 2 + (k >> 5)) * 64 + (n & 63)) * 32 + (k & 31);
    *reinterpret_cast<u32x4*>(s_w + n * PITCH + piece * 16) = *reinterpret_cast<const u32x4*>(w3 + src);
  }
  s_b[tid] = b3[tid];
  // W6^T fragments (A operand of GEMM 2), in the accumulator order of GEMM 1: k-slot (q, i) of step (tt, half) is channel
  // tt*32 + 16*half + (i & 3) + 8*(i >> 2) + 4*q of the quadrant; rows >= 4 are zero
  hbf16x8 w6f[2][2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint32_t pk[4];
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        uint32_t lo = 0, hi = 0;
        if (lx < 4) {
          lo = w6[lx * 64 + tt * 32 + 16 * half + (i & 3) + 8 * (i >> 2) + 4 * q];
          hi = w6[lx * 64 + tt * 32 + 16 * half + ((i + 1) & 3) + 8 * ((i + 1) >> 2) + 4 * q];
        }
        pk[i >> 1] = lo | (hi << 16);
      }
      const u32x4 v = {pk[0], pk[1], pk[2], pk[3]};
      w6f[tt][half] = __builtin_bit_cast(hbf16x8, v);
    }
  const float bias6 = b6[0];
  __syncthreads();
  const long long nbatch = (npix + 31) >> 5;
  // the next batch's pixels are fetched under this batch's arithmetic: with four waves per SIMD and a ~2.5 us load at the top of every
  // ~1 us batch the kernel sat at 2 TB/s
  const long long bstep = (long long)gridDim.x * 4;
  hbf16x8 xn[4];
  {
    const long long bt0 = (long long)blockIdx.x * 4 + wave;
    const long long p0 = bt0 * 32 + lx;
    const long long pc0 = p0 < npix ? p0 : npix - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) xn[j] = *reinterpret_cast<const hbf16x8*>(in + (size_t)pc0 * 64 + j * 16 + q * 8);
  }
  for (long long bt = (long long)blockIdx.x * 4 + wave; bt < nbatch; bt += bstep) {
    const long long pix = bt * 32 + lx;
    hbf16x8 xf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xf[j] = xn[j];
    if (bt + bstep < nbatch) {
      const long long pn = (bt + bstep) * 32 + lx;
      const long long pcn = pn < npix ? pn : npix - 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) xn[j] = *reinterpret_cast<const hbf16x8*>(in + (size_t)pcn * 64 + j * 16 + q * 8);
    }
    // the weight fragments are re-read from LDS for every batch: hoisted out of the loop they would take 128 VGPRs and spill
    // (LDS-typed pointers: made opaque as generic pointers, every one of the 48 fragment reads of a batch was a FLAT load -- slower than ds_read, and with
    // FLAT loads in flight hipcc waits lgkmcnt(0) / vmcnt(0) in front of each MFMA)
    typedef __attribute__((address_space(3))) const char* lds_cptr;
    typedef __attribute__((address_space(3))) const float* lds_fptr;
    lds_cptr sw = (lds_cptr)(s_w + lx * PITCH + q * 16);
    lds_fptr sb = (lds_fptr)(s_b + 4 * q);
    asm volatile("" : "+v"(sw), "+v"(sb));
    float outv[4][4];                               // [quadrant][sub-pixel] of this lane's pixel (valid in the q == 0 lanes)
#pragma unroll
    for (int Q = 0; Q < 4; ++Q) {                   // quadrant by quadrant: 32 accumulators live, not 128
      hf32x16 d2;
#pragma unroll
      for (int r = 0; r < 16; ++r) d2[r] = 0.f;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * Q + tt;
        hf32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const hbf16x8 a = *reinterpret_cast<__attribute__((address_space(3))) const hbf16x8*>(sw + t * 32 * PITCH + j * 32);
          acc = mfma_32x32x16_a16(a, xf[j], acc);
        }
        // bias + ReLU + round-to-nearest-even to bf16, two values per v_cvt_pk_bf16_f32: the integer rounding (five instructions per
        // value, 640 per 32-pixel batch against 48 MFMAs) had made this streaming kernel VALU-bound
        float hv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = fmaxf(acc[r] + sb[t * 32 + (r & 3) + 8 * (r >> 2)], 0.f);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const u32x4 bv = {pack2bf(hv[8 * half + 0], hv[8 * half + 1]), pack2bf(hv[8 * half + 2], hv[8 * half + 3]),
                            pack2bf(hv[8 * half + 4], hv[8 * half + 5]), pack2bf(hv[8 * half + 6], hv[8 * half + 7])};
          d2 = mfma_32x32x16_a16(w6f[tt][half], __builtin_bit_cast(hbf16x8, bv), d2);
        }
      }
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) outv[Q][sp] = d2[sp] + bias6;      // rows 0..3 live in registers 0..3 of the q == 0 lanes
    }
    if (q == 0 && pix < npix) {
      const int ox = (int)(pix % W);
      const long long tq = pix / W;
      const int oy = (int)(tq % H);
      const long long b = tq / H;
      const size_t OW = (size_t)4 * W;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {              // output row 4*oy + rr: quadrant row a = rr >> 1, sub-pixel row c = rr & 1
        const int a = rr >> 1, c = rr & 1;
        const hf32x4 lg = {outv[2 * a][2 * c], outv[2 * a][2 * c + 1], outv[2 * a + 1][2 * c], outv[2 * a + 1][2 * c + 1]};
        const size_t o = ((size_t)b * (4 * H) + 4 * oy + rr) * OW + (size_t)4 * ox;
        if (logits) *reinterpret_cast<hf32x4*>(logits + o) = lg;
        if (prob) {
          const hf32x4 pr = {1.f / (1.f + expf(-lg.x)), 1.f / (1.f + expf(-lg.y)), 1.f / (1.f + expf(-lg.z)), 1.f / (1.f + expf(-lg.w))};
          *reinterpret_cast<hf32x4*>(prob + o) = pr;
          if (bitmap) {
            // prob > thresh, bit-packed (bitmap_kernel's words: bit k of word i = pixel 32 i + k): a lane holds 4 pixels of the row, the 8 lanes of
            // an aligned group (W % 8 == 0, batches start at multiples of 32: the group is one 32-pixel word, all in or all out of the map) meet
            // by three exchanges -- the separate pass re-read the 236 MB probability map of 64 pages for this
            uint32_t wv = ((pr.x > thresh ? 1u : 0u) | (pr.y > thresh ? 2u : 0u) | (pr.z > thresh ? 4u : 0u) | (pr.w > thresh ? 8u : 0u)) << (4 * (lx & 7));
            wv |= __shfl_xor(wv, 1);
            wv |= __shfl_xor(wv, 2);
            wv |= __shfl_xor(wv, 4);
            if ((lx & 7) == 0) bitmap[o >> 5] = wv;
          }
        }
      }
    }
  }
}

int pt_launch_db_head_mfma(const bf16_t* in, int B, int H, int W, const bf16_t* w3, const float* b3, const bf16_t* w6,
                           const float* b6, float* prob, float* logits, hipStream_t s, uint32_t* bitmap, float thresh) {
  PT_REQUIRE(!bitmap || (prob && W % 8 == 0), "db head: the fused bitmap needs the probability map and a width that is a multiple of 8");
  const long long npix = (long long)B * H * W;
  long long blocks = (npix / 32 + 4 * 12 - 1) / (4 * 12);     // ~12 batches per wave: the 37 KB weight image is staged once per workgroup
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;                            // whole rounds of 256 CUs x 4 workgroups (2 400 were 2.3 rounds)
  hipLaunchKernelGGL(db_head_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, npix, H, W, w3, b3, w6, b6, prob, logits, bitmap, thresh);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// prob > thresh -> bit-packed bitmap (DBPostProcess.__call__, processor_ocr_db_pp.py:296), with the
// optional 2x2 all-ones cv2.dilate (anchor (1,1): out(x,y) = max over x-1..x, y-1..y; :301-304).
// One lane per pixel, wave ballot -> two 32-bit words.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bitmap_kernel(const float* __restrict__ prob, int n, int H, int W, float thresh,
                                                      int dilate, uint32_t* __restrict__ bitmap) {
  a16_kernel_enter();
  const long long total = (long long)n * H * W;  // W % 32 == 0 -> total % 32 == 0
  const long long rounded = (total + 63) & ~63ll;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += (long long)gridDim.x * blockDim.x) {
    bool on = false;
    if (i < total) {
      if (!dilate) {
        on = prob[i] > thresh;
      } else {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        on = prob[i] > thresh;
        if (x > 0) on = on || (prob[i - 1] > thresh);
        if (y > 0) on = on || (prob[i - W] > thresh);
        if (x > 0 && y > 0) on = on || (prob[i - W - 1] > thresh);
      }
    }
    const unsigned long long m = __ballot(on);
    const int lane = threadIdx.x & 63;
    if (lane == 0 && i < total) bitmap[i >> 5] = (uint32_t)m;
    if (lane == 32 && i < total) bitmap[i >> 5] = (uint32_t)(m >> 32);
  }
}

int pt_launch_bitmap(const float* prob, int n, int H, int W, float thresh, int dilate, uint32_t* bitmap,
                     hipStream_t s) {
  PT_REQUIRE(W % 32 == 0, "bitmap: W must be a multiple of 32");
  const long long total = (long long)n * H * W;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(bitmap_kernel, dim3(blocks), dim3(256), 0, s, prob, n, H, W, thresh, dilate, bitmap);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// box_score_fast (processor_ocr_db_pp.py:253-268): mean of prob over the cv2.fillPoly mask of the
// quad inside its clipped bounding rectangle.  fillPoly semantics restated from OpenCV's drawing.cpp
// (CollectPolyEdges + FillEdgeCollection, shift = 0, 8-connected): Bresenham outline of every edge
// (iterated left-to-right, error term rounds exact halves toward the start point) united with the
// scan-line interior x in [ (xa+0x8000)>>16 , (xb+0x8000)>>16 ] for ymin <= y < ymax, edges advanced
// in 16.16 fixed point with a truncated slope.  One wave per box; double accumulation like cv2.mean.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool on_bresenham(int px, int py, int x0, int y0, int x1, int y1) {
  // left-to-right start point
  if (x1 < x0) { int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
  const int dx = x1 - x0;
  const int dyabs = y1 >= y0 ? y1 - y0 : y0 - y1;
  const int sy = y1 >= y0 ? 1 : -1;
  // minor-axis position after j major steps is floor((2*minor*j + major - 1) / (2*major)); the membership test
  //   m == u  <=>  u*2*major <= 2*minor*j + major - 1 < (u+1)*2*major   needs no division (|coords| < 2^12)
  if (dyabs > dx) {  // y-major
    const int j = (py - y0) * sy, u = px - x0;
    if (j < 0 || j > dyabs || u < 0) return false;
    const int num = 2 * dx * j + dyabs - 1;
    return u * 2 * dyabs <= num && num < (u + 1) * 2 * dyabs;
  }
  if (dx == 0) return px == x0 && py == y0;  // single point
  const int j = px - x0, u = (py - y0) * sy;
  if (j < 0 || j > dx || u < 0) return false;
  const int num = 2 * dyabs * j + dx - 1;
  return u * 2 * dx <= num && num < (u + 1) * 2 * dx;
}

// one workgroup per box: the waves take the rows of the bounding rectangle in turn (the scan-line spans of a row are
// computed once, wave-uniform), the lanes its pixels; double sums, reduced through LDS.  One wave per box left the
// largest box of a page batch (a ruled table is one connected component) as a 3 ms tail.
constexpr int BS_THREADS = 256;

__global__ __launch_bounds__(BS_THREADS) void box_score_kernel(const float* __restrict__ prob, int n, int H, int W,
                                                        const float* __restrict__ boxes, int nb,
                                                        float* __restrict__ scores) {
  a16_kernel_enter();
  const int bi = blockIdx.x;
  if (bi >= nb) return;
  const float* bx = boxes + (size_t)bi * 9;
  const int page = (int)bx[0];
  float fx[4], fy[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { fx[k] = bx[1 + 2 * k]; fy[k] = bx[2 + 2 * k]; }
  const float mnx = fminf(fminf(fx[0], fx[1]), fminf(fx[2], fx[3]));
  const float mxx = fmaxf(fmaxf(fx[0], fx[1]), fmaxf(fx[2], fx[3]));
  const float mny = fminf(fminf(fy[0], fy[1]), fminf(fy[2], fy[3]));
  const float mxy = fmaxf(fmaxf(fy[0], fy[1]), fmaxf(fy[2], fy[3]));
  auto clampi = [](long long v, int lo, int hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); };
  const int xmin = clampi((long long)floorf(mnx), 0, W - 1), xmax = clampi((long long)ceilf(mxx), 0, W - 1);
  const int ymin = clampi((long long)floorf(mny), 0, H - 1), ymax = clampi((long long)ceilf(mxy), 0, H - 1);
  const int mw = xmax - xmin + 1, mh = ymax - ymin + 1;
  int vx[4], vy[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    vx[k] = (int)(fx[k] - (float)xmin);  // astype(int32): truncation toward zero
    vy[k] = (int)(fy[k] - (float)ymin);
  }
  // polygon edges for the scan-line part
  int ey0[4], ey1[4];
  long long ex[4], edx[4];
  int ne = 0;
  int pymin = vy[0], pymax = vy[0];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pymin = vy[k] < pymin ? vy[k] : pymin;
    pymax = vy[k] > pymax ? vy[k] : pymax;
    const int k0 = (k + 3) & 3;
    const int ax = vx[k0], ay = vy[k0], cx = vx[k], cy = vy[k];
    if (ay == cy) continue;
    const long long X0 = (long long)ax << 16, X1 = (long long)cx << 16;
    if (ay < cy) { ey0[ne] = ay; ey1[ne] = cy; ex[ne] = X0; } else { ey0[ne] = cy; ey1[ne] = ay; ex[ne] = X1; }
    edx[ne] = (X1 - X0) / (cy - ay);
    ++ne;
  }
  const float* pm = prob + (size_t)page * H * W;
  double sum = 0.0;
  int cnt = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int py = wave; py < mh; py += BS_THREADS / 64) {
    int x1s[2] = {1, 1}, x2s[2] = {0, 0};      // up to two spans [x1, x2] of the row
    if (py >= pymin && py < pymax) {
      long long xs[4];
      int na = 0;
      for (int k = 0; k < ne; ++k)
        if (py >= ey0[k] && py < ey1[k]) xs[na++] = ex[k] + (long long)(py - ey0[k]) * edx[k];
      // sort (na <= 4)
      for (int a = 1; a < na; ++a)
        for (int c = a; c > 0 && xs[c] < xs[c - 1]; --c) { long long tt = xs[c]; xs[c] = xs[c - 1]; xs[c - 1] = tt; }
      for (int a = 0; a + 1 < na; a += 2) {
        x1s[a >> 1] = (int)((xs[a] + 32768) >> 16);
        x2s[a >> 1] = (int)((xs[a + 1] + 32768) >> 16);
      }
    }
    const float* prow = pm + (size_t)(ymin + py) * W + xmin;
    for (int px = lane; px < mw; px += 64) {
      bool in = (px >= x1s[0] && px <= x2s[0]) || (px >= x1s[1] && px <= x2s[1]);
      if (!in) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int k0 = (k + 3) & 3;
          if (on_bresenham(px, py, vx[k0], vy[k0], vx[k], vy[k])) in = true;
        }
      }
      if (in) {
        sum += (double)prow[px];
        ++cnt;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_xor(sum, off);
    cnt += __shfl_xor(cnt, off);
  }
  __shared__ double s_sum[BS_THREADS / 64];
  __shared__ int s_cnt[BS_THREADS / 64];
  if (lane == 0) { s_sum[wave] = sum; s_cnt[wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < BS_THREADS / 64; ++k) { sum += s_sum[k]; cnt += s_cnt[k]; }
    scores[bi] = cnt > 0 ? (float)(sum / (double)cnt) : 0.f;
  }
}

int pt_launch_box_scores(const float* prob, int n, int H, int W, const float* boxes, int nb, float* scores,
                         hipStream_t s) {
  if (nb <= 0) return PT_OK;
  hipLaunchKernelGGL(box_score_kernel, dim3(nb), dim3(BS_THREADS), 0, s, prob, n, H, W, boxes, nb, scores);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

}  // namespace PT_FMT_NS
