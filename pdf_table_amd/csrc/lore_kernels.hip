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
#include "common.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

namespace {

__device__ __forceinline__ float bf2f(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}

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
  const int cgn = C >> 3;
  const int cs = split ? 2 * C : C;
  const int OH = h * f, OW = wd * f, p = f / 2, k = 2 * f;
  const long long total = (long long)B * OH * OW * cgn;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cgn);
    long long t = i / cgn;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
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

// cv2.warpAffine(crop, M, (W, H), INTER_LINEAR, BORDER_CONSTANT 0) + (x/255 - mean)/std of TableLorePreProcessor.process
// (lore/processer_lore.py:85-90), sampling the table crop straight from the resident page.  OpenCV's WarpAffineInvoker
// arithmetic: inverse map in fp64, 10-bit fixed-point coordinates, 1/32-pixel positions, 15-bit weights.
// lut[c][v] = float(((v / 255.) - mean[c]) / std[c]) evaluated in fp64 on the host like numpy does.
__global__ __launch_bounds__(256) void tsr_preprocess_kernel(const uint8_t* __restrict__ pages, int ph, int pw,
                                                              const pt_tsr_table* __restrict__ tabs, int H, int W, int bgr,
                                                              const float* __restrict__ lut, bf16_t* __restrict__ out,
                                                              int split) {
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
