// rec_kernels.hip -- kernels of the text-line recognition stage that are not dense convolutions:
//   * perspective crop of every detected quad straight from the page image (replaces the per-line host
//     cv2.warpPerspective of OcrCommonUtils.crop_image, utils/ocr/ocr_common_utils.py:214-266)
//   * keep-ratio resize to 32 x W, zero pad to 640, /255, RGB->gray (processor_ocr_recognition.py:44-62,111;
//     crnn/modeling_crnn.py:94)
//   * CRNN conv0 (1->64, 3x3) + BN + ReLU + 2x2 max-pool fused (crnn/modeling_crnn.py:40-47)
//   * max-pools 2x2 / (2,1) on NHWC (:55,:68,:81), the last one also folding H into channels for the (2,1) conv
//   * one direction of an LSTM layer for 32 text lines per workgroup (nn.LSTM equations; modeling_crnn.py:19-33)
//   * arg-max reduction over the class tiles produced by the fused classifier GEMM epilogue
// Compiled with -ffp-contract=off.
#include <math.h>

#include "common.h"

namespace PT_FMT_NS {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// 16 stored bits <-> fp32 in the storage format of this namespace (act16.h: bf16, or IEEE half in pt_f16)
__device__ __forceinline__ float rbf2f(uint32_t bits16) { return a16_to_f32(bits16); }
__device__ __forceinline__ uint32_t rf2bf(float f) { return f32_to_a16(f); }

// ---------------------------------------------------------------------------------------------------
// Perspective crop.  cv2.warpPerspective(img, M, (w, h)) semantics for 8-bit, INTER_LINEAR, constant border 0:
// destination (x, y) -> (X, Y, W) = Minv * (x, y, 1) in double; X*32/W, Y*32/W rounded to nearest integer give
// the source position in 1/32 pixel; the four neighbours are blended with 15-bit weights
// (32-ay)(32-ax)*32 ... and the sum is rounded with +2^14 >> 15.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rec_warp_kernel(const uint8_t* __restrict__ pages, int ph, int pw,
                                                        const pt_rec_line* __restrict__ lines, int n_lines,
                                                        const long long* __restrict__ pix_off, uint8_t* __restrict__ crops) {
  a16_kernel_enter();
  const int li = blockIdx.y;
  if (li >= n_lines) return;
  const pt_rec_line L = lines[li];
  const int cw = L.crop_w, chh = L.crop_h;
  const long long npx = (long long)cw * chh;
  uint8_t* dst = crops + pix_off[li] * 3;
  const uint8_t* src = pages + (size_t)L.page * ph * pw * 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % cw), y = (int)(i / cw);
    const double X0 = L.minv[0] * x + L.minv[1] * y + L.minv[2];
    const double Y0 = L.minv[3] * x + L.minv[4] * y + L.minv[5];
    double W = L.minv[6] * x + L.minv[7] * y + L.minv[8];
    W = W != 0. ? 32. / W : 0.;
    const double fX = fmax(-2147483648., fmin(2147483647., X0 * W));
    const double fY = fmax(-2147483648., fmin(2147483647., Y0 * W));
    const long long Xi = (long long)rint(fX), Yi = (long long)rint(fY);
    const long long sx = Xi >> 5, sy = Yi >> 5;
    const int ax = (int)(Xi & 31), ay = (int)(Yi & 31);
    const int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
    int acc[3] = {0, 0, 0};
    auto tap = [&](long long yy, long long xx, int wgt) {
      if (wgt && yy >= 0 && yy < ph && xx >= 0 && xx < pw) {
        const uint8_t* p = src + ((size_t)yy * pw + xx) * 3;
        acc[0] += p[0] * wgt; acc[1] += p[1] * wgt; acc[2] += p[2] * wgt;
      }
    };
    tap(sy, sx, w00); tap(sy, sx + 1, w01); tap(sy + 1, sx, w10); tap(sy + 1, sx + 1, w11);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int v = (acc[c] + (1 << 14)) >> 15;
      dst[i * 3 + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}

// off[i] = sum_{j < i} max(crop_w_j * crop_h_j, 0), i = 0 .. n: where line i's crop starts in the ragged crop buffer.
// One workgroup, chunked Hillis-Steele scan in LDS with a running carry (n is a micro-batch: a few thousand lines).
__global__ __launch_bounds__(1024) void rec_offsets_kernel(const pt_rec_line* __restrict__ lines, int n,
                                                           long long* __restrict__ off) {
  a16_kernel_enter();
  __shared__ long long sc[1024];
  __shared__ long long carry;
  const int tid = threadIdx.x;
  if (tid == 0) { carry = 0; off[0] = 0; }
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    long long v = 0;
    if (i < n) {
      v = (long long)lines[i].crop_w * (long long)lines[i].crop_h;
      if (v < 0) v = 0;
    }
    sc[tid] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const long long t = tid >= d ? sc[tid - d] : 0;
      __syncthreads();
      sc[tid] += t;
      __syncthreads();
    }
    if (i < n) off[i + 1] = carry + sc[tid];
    __syncthreads();
    if (tid == 1023) carry += sc[1023];
    __syncthreads();
  }
}

int pt_launch_rec_offsets(const pt_rec_line* lines, int n, long long* off, hipStream_t s) {
  hipLaunchKernelGGL(rec_offsets_kernel, dim3(1), dim3(1024), 0, s, lines, n, off);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_launch_rec_warp(const uint8_t* pages, int ph, int pw, const pt_rec_line* lines, int n_lines,
                       const long long* pix_off, uint8_t* crops, int max_crop_px, hipStream_t s) {
  if (n_lines <= 0) return PT_OK;
  int bx = (max_crop_px + 255) / 256;
  bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
  hipLaunchKernelGGL(rec_warp_kernel, dim3(bx, n_lines), dim3(256), 0, s, pages, ph, pw, lines, n_lines, pix_off, crops);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// keepratio_resize (cv2.resize 8-bit bilinear, see det_kernels.hip) + zero pad + /255 + gray.
// out: bf16 [n, 32, 640] (split: [n, 32, 640, 2] = hi, lo)
// ---------------------------------------------------------------------------------------------------
struct RCoef {
  int s0, s1, a0, a1;
};
__device__ __forceinline__ RCoef rcoef(int d, double scale, int ssize, bool clamp_frac) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (clamp_frac) {
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  }
  RCoef c;
  c.a0 = (int)rintf((1.f - f) * 2048.f);
  c.a1 = (int)rintf(f * 2048.f);
  int t0 = s, t1 = s + 1;
  c.s0 = t0 < 0 ? 0 : (t0 >= ssize ? ssize - 1 : t0);
  c.s1 = t1 < 0 ? 0 : (t1 >= ssize ? ssize - 1 : t1);
  return c;
}

__global__ __launch_bounds__(256) void rec_resize_gray_kernel(const uint8_t* __restrict__ crops,
                                                               const pt_rec_line* __restrict__ lines,
                                                               const long long* __restrict__ pix_off, int n_lines, int TH,
                                                               int TWID, int split, bf16_t* __restrict__ out,
                                                               float* __restrict__ out_f32) {
  a16_kernel_enter();
  const long long total = (long long)n_lines * TH * TWID;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % TWID);
    const long long t = i / TWID;
    const int y = (int)(t % TH);
    const int li = (int)(t / TH);
    const int cw = lines[li].crop_w, chh = lines[li].crop_h;
    float gray = 0.f;
    if (cw > 0 && chh > 0) {
      // cur_ratio > target_w / target_h ? (32, 640) : (32, int(32 * ratio))   (python float arithmetic)
      const double ratio = (double)cw / (double)chh;
      const int nw = ratio > (double)TWID / (double)TH ? TWID : (int)((double)TH * ratio);
      if (x < nw) {
        const uint8_t* src = crops + pix_off[li] * 3;
        int v[3];
        if (cw == nw && chh == TH) {
          const uint8_t* p = src + ((size_t)y * cw + x) * 3;
          v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
        } else if (cw == 2 * nw && chh == 2 * TH) {
          const uint8_t* p0 = src + ((size_t)(2 * y) * cw + 2 * x) * 3;
          const uint8_t* p1 = p0 + (size_t)cw * 3;
#pragma unroll
          for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2;
        } else {
          const RCoef cx = rcoef(x, (double)cw / nw, cw, true);
          const RCoef cy = rcoef(y, (double)chh / TH, chh, false);
          const uint8_t* r0 = src + (size_t)cy.s0 * cw * 3;
          const uint8_t* r1 = src + (size_t)cy.s1 * cw * 3;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int S0 = r0[cx.s0 * 3 + c] * cx.a0 + r0[cx.s1 * 3 + c] * cx.a1;
            const int S1 = r1[cx.s0 * 3 + c] * cx.a0 + r1[cx.s1 * 3 + c] * cx.a1;
            int r = (((cy.a0 * (S0 >> 4)) >> 16) + ((cy.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
            v[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
          }
        }
        const float R = (float)v[0] / 255.f, G = (float)v[1] / 255.f, B = (float)v[2] / 255.f;
        gray = (R * 0.2989f + G * 0.5870f) + B * 0.1140f;  // modeling_crnn.py:94, left-to-right
      }
    }
    if (out_f32) {                 // ConvNextViT (fp32 stream from the first layer on, cvit_model.hip)
      out_f32[i] = gray;
      continue;
    }
    const uint32_t hb = rf2bf(gray);
    if (split) {
      out[i * 2] = (bf16_t)hb;
      out[i * 2 + 1] = (bf16_t)rf2bf(gray - rbf2f(hb));
    } else {
      out[i] = (bf16_t)hb;
    }
  }
}

int pt_launch_rec_resize_gray(const uint8_t* crops, const pt_rec_line* lines, const long long* pix_off, int n_lines,
                              int split, bf16_t* out, hipStream_t s) {
  if (n_lines <= 0) return PT_OK;
  const long long total = (long long)n_lines * 32 * 640;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(rec_resize_gray_kernel, dim3(blocks), dim3(256), 0, s, crops, lines, pix_off, n_lines, 32, 640, split, out,
                     static_cast<float*>(nullptr));
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// the same resize to 32 x tw (804 for the chunking ConvNextViT pre-processor, processor_ocr_recognition.py:38-62) as fp32 gray
int pt_launch_rec_resize_gray_f32(const uint8_t* crops, const pt_rec_line* lines, const long long* pix_off, int n_lines, int tw,
                                  float* out, hipStream_t s) {
  if (n_lines <= 0) return PT_OK;
  const long long total = (long long)n_lines * 32 * tw;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(rec_resize_gray_kernel, dim3(blocks), dim3(256), 0, s, crops, lines, pix_off, n_lines, 32, tw, 0,
                     static_cast<bf16_t*>(nullptr), out);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// Ragged CRNN conv stack.  A text line is resized to height 32 and zero-padded to 640 columns (keepratio_resize,
// processor_ocr_recognition.py:44-62); right of the text every conv layer produces what an ALL-padding line produces at
// the same position (the kernels are position-independent and see the same zeros), so those columns need not be
// computed: they are copied from the cached activations of an all-padding line.  crnn_limits_kernel derives, per line,
// the first "clean" output column of each limited conv layer from the resized text width nw (the same formula as
// rec_resize_gray_kernel):
//   conv0 (3x3) + pool 2x2   clean from e0  = ceil((nw + 1) / 2)        (computed in full: K = 9, 1.6 ms per step)
//   conv1 (3x3)              clean from e0 + 1;      after its 2x2 pool: e1 = ceil((e0 + 1) / 2)
//   conv2a / conv2b (3x3)    e1 + 1 / e1 + 2         (the (2,1) pool keeps columns)
//   conv3a / conv3b (3x3)    e1 + 3 / e1 + 4
// lim[k][b] is that column (in the conv's own output columns); cols[k] sums the tile-rounded limits (roofline accounting).
// ---------------------------------------------------------------------------------------------------
__global__ void crnn_limits_kernel(const pt_rec_line* __restrict__ lines, int n, int* l1, int* l2a, int* l2b, int* l3a, int* l3b,
                                   int* l0, int* __restrict__ cols) {
  a16_kernel_enter();
  __shared__ int sums[6];
  if (threadIdx.x < 6) sums[threadIdx.x] = 0;
  __syncthreads();
  int acc[6] = {0, 0, 0, 0, 0, 0};
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < n; b += gridDim.x * blockDim.x) {
    const int cw = lines[b].crop_w, chh = lines[b].crop_h;
    int nw = 0;
    if (cw > 0 && chh > 0) {
      const double ratio = (double)cw / (double)chh;
      nw = ratio > 640.0 / 32.0 ? 640 : (int)(32.0 * ratio);
    }
    const int e0 = (nw + 2) >> 1;                 // ceil((nw + 1) / 2)
    const int e1 = (e0 + 2) >> 1;                 // ceil((e0 + 1) / 2)
    const int v[5] = {min(e0 + 1, 320), min(e1 + 1, 160), min(e1 + 2, 160), min(e1 + 3, 160), min(e1 + 4, 160)};
    l1[b] = v[0]; l2a[b] = v[1]; l2b[b] = v[2]; l3a[b] = v[3]; l3b[b] = v[4];
    // conv0 (computed in 64-column pooled tiles) must deliver the columns the limited conv1 reads: its tiles [0, R1) + the halo
    l0[b] = min((v[0] + 31) / 32 * 32 + 1, 320);
    const int tw[5] = {32, 32, 32, 32, 32}, wo[5] = {320, 160, 160, 160, 160};      // (conv3.*: 64-column patches, 32-column row-tiles)
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] += min(wo[k], (v[k] + tw[k] - 1) / tw[k] * tw[k]);
    acc[5] += min(160, (v[4] + 31) / 32 * 32);     // the sequence GEMMs (conv4, first LSTM projection): 32-step tiles
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) atomicAdd(&sums[k], acc[k]);
  __syncthreads();
  if (threadIdx.x < 6) atomicAdd(&cols[threadIdx.x], sums[threadIdx.x]);
}

// The live 32-step row groups of the sequence GEMMs, compacted: glist[0] = their number, glist[1 + g] = row group (line * 5 + k) of the g-th,
// in line order (k * 32 < lim[line]).  One workgroup: per-thread runs of lines, an exclusive scan of the runs' counts through LDS.
__global__ __launch_bounds__(1024) void rows_live_list_kernel(const int* __restrict__ lim, int n, int* __restrict__ glist) {
  a16_kernel_enter();
  __shared__ int part[1024];
  const int tid = threadIdx.x, per = (n + 1023) / 1024, b0 = tid * per, b1 = min(n, b0 + per);
  int cnt = 0;
  for (int b = b0; b < b1; ++b) cnt += min(PT_REC_T / 32, (lim[b] + 31) / 32);
  part[tid] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int pos = part[tid] - cnt;
  if (tid == 1023) glist[0] = part[1023];
  for (int b = b0; b < b1; ++b) {
    const int c = min(PT_REC_T / 32, (lim[b] + 31) / 32);
    for (int k = 0; k < c; ++k) glist[1 + pos + k] = b * (PT_REC_T / 32) + k;
    pos += c;
  }
}

int pt_launch_rows_live_list(const int* lim, int n, int* glist, hipStream_t s) {
  hipLaunchKernelGGL(rows_live_list_kernel, dim3(1), dim3(1024), 0, s, lim, n, glist);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_launch_crnn_limits(const pt_rec_line* lines, int n, const PtCrnnLimits& L, hipStream_t s) {
  PT_HIP_CHECK(hipMemsetAsync(L.cols, 0, 8 * sizeof(int), s));
  int blocks = (n + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(crnn_limits_kernel, dim3(blocks), dim3(256), 0, s, lines, n, L.lim[0], L.lim[1], L.lim[2], L.lim[3], L.lim[4], L.lim[5], L.cols);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// out [n][rows][W][cs] (bf16; cs = channels incl. the lo half in hi/lo mode), ref [rows][W][cs]: for every line b copy
// ref -> out for the columns x >= xf(b) = (roundup(lim[b], tile_w)) / div -- the columns the limited conv left untouched
// (div = 2 when a 2x2 pool follows the conv in its epilogue).  16-byte pieces; a block walks (row, column, piece) of one line.
__global__ __launch_bounds__(256) void crnn_fill_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ ref,
                                                         const int* __restrict__ lim, int tile_w, int div, int rows, int W, int cs,
                                                         const int* __restrict__ end_lim, int end_tile) {
  a16_kernel_enter();
  const int b = blockIdx.y;
  const int xf = ((lim[b] + tile_w - 1) / tile_w * tile_w) / div;
  int xe = W;
  if (end_lim) xe = min(W, (end_lim[b] + end_tile - 1) / end_tile * end_tile + 1);    // + 1: the next conv's halo column
  if (xf >= xe) return;
  const int pc = cs >> 3, wcols = xe - xf;
  const long long total = (long long)rows * wcols * pc;
  u32x4* o = reinterpret_cast<u32x4*>(out + (size_t)b * rows * W * cs);
  const u32x4* r = reinterpret_cast<const u32x4*>(ref);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int piece = (int)(i % pc);
    const long long t = i / pc;
    const int x = xf + (int)(t % wcols), y = (int)(t / wcols);
    const size_t idx = ((size_t)y * W + x) * pc + piece;
    o[idx] = r[idx];
  }
}

int pt_launch_crnn_fill(bf16_t* out, const bf16_t* ref, const int* lim, int tile_w, int div, int n, int rows, int W, int cs,
                        hipStream_t s, const int* end_lim, int end_tile) {
  if (n <= 0) return PT_OK;
  PT_REQUIRE(cs % 8 == 0, "crnn fill: channel stride must be a multiple of 8");
  long long per = (long long)rows * W * (cs >> 3);
  int bx = (int)((per + 255) / 256);
  bx = bx < 1 ? 1 : (bx > 16 ? 16 : bx);
  hipLaunchKernelGGL(crnn_fill_kernel, dim3(bx, n), dim3(256), 0, s, out, ref, lim, tile_w, div, rows, W, cs, end_lim, end_tile);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// PPOcrRecPreProcessor.resize_norm_img (ocr_rec_pp/processor_ocr_rec_pp.py:43-67) for every line of a width-sorted plan:
// cv2.resize (8-bit bilinear, as above) of the crop to img_h x resized_w, (x / 255 - 0.5) / 0.5 through a 256-entry fp32
// table built on the host with the reference's operation order, zeros right of resized_w up to the mini-batch width.
// out: fp32, one [3, img_h, img_w] block per item at item.out_off (the reference's NCHW mini-batch layout).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rec_pp_resize_norm_kernel(const uint8_t* __restrict__ crops,
                                                                  const pt_rec_line* __restrict__ lines,
                                                                  const long long* __restrict__ pix_off,
                                                                  const pt_rec_pp_item* __restrict__ items, int img_h,
                                                                  const float* __restrict__ lut, float* __restrict__ out) {
  a16_kernel_enter();
  __shared__ float slut[256];
  slut[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const pt_rec_pp_item it = items[blockIdx.y];
  const int li = it.line, nw = it.resized_w, iw = it.img_w;
  const int cw = lines[li].crop_w, chh = lines[li].crop_h;
  const uint8_t* src = crops + pix_off[li] * 3;
  float* o = out + it.out_off;
  const long long plane = (long long)img_h * iw;
  const bool same = cw == nw && chh == img_h, half = cw == 2 * nw && chh == 2 * img_h;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % iw), y = (int)(i / iw);
    float r[3] = {0.f, 0.f, 0.f};
    if (x < nw && cw > 0 && chh > 0) {
      int v[3];
      if (same) {
        const uint8_t* p = src + ((size_t)y * cw + x) * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
      } else if (half) {
        const uint8_t* p0 = src + ((size_t)(2 * y) * cw + 2 * x) * 3;
        const uint8_t* p1 = p0 + (size_t)cw * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2;
      } else {
        const RCoef cx = rcoef(x, (double)cw / nw, cw, true);
        const RCoef cy = rcoef(y, (double)chh / img_h, chh, false);
        const uint8_t* r0 = src + (size_t)cy.s0 * cw * 3;
        const uint8_t* r1 = src + (size_t)cy.s1 * cw * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int S0 = r0[cx.s0 * 3 + c] * cx.a0 + r0[cx.s1 * 3 + c] * cx.a1;
          const int S1 = r1[cx.s0 * 3 + c] * cx.a0 + r1[cx.s1 * 3 + c] * cx.a1;
          const int q = (((cy.a0 * (S0 >> 4)) >> 16) + ((cy.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
          v[c] = q < 0 ? 0 : (q > 255 ? 255 : q);
        }
      }
      r[0] = slut[v[0]]; r[1] = slut[v[1]]; r[2] = slut[v[2]];
    }
    o[i] = r[0];
    o[plane + i] = r[1];
    o[2 * plane + i] = r[2];
  }
}

int pt_launch_rec_pp_resize_norm(const uint8_t* crops, const pt_rec_line* lines, const long long* pix_off,
                                 const pt_rec_pp_item* items, int n_items, int img_h, int max_img_w, const float* lut, float* out,
                                 hipStream_t s) {
  if (n_items <= 0) return PT_OK;
  int bx = (int)(((long long)img_h * max_img_w + 255) / 256);
  bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
  hipLaunchKernelGGL(rec_pp_resize_norm_kernel, dim3(bx, n_items), dim3(256), 0, s, crops, lines, pix_off, items, img_h, lut, out);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// CRNN conv0: Conv2d(1, 64, 3, pad 1) + BN + ReLU, then MaxPool2d(2, 2).  Direct fp32 convolution on the VALU
// (K = 9 is far too thin for MFMA).  in: gray bf16 [n, H, W] (split: hi/lo pairs); w fp32 [64][9] (BN folded, rounded
// to bf16 by the packer in bf16 mode), bias fp32 [64]; out bf16 [n, H/2, W/2, 64] (split: [hi(64) | lo(64)]).
// One thread = one pooled pixel x 8 channels.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crnn_conv0_pool_kernel(const bf16_t* __restrict__ in, int n, int H, int W,
                                                               const float* __restrict__ w64x9,
                                                               const float* __restrict__ bias, int split,
                                                               bf16_t* __restrict__ out, const int* __restrict__ xlim) {
  a16_kernel_enter();
  __shared__ float sw[64 * 9 + 64];
  for (int i = threadIdx.x; i < 64 * 9; i += blockDim.x) sw[i] = w64x9[i];
  for (int i = threadIdx.x; i < 64; i += blockDim.x) sw[576 + i] = bias[i];
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2;
  const long long total = (long long)n * Ho * Wo * 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i & 7);
    long long t = i >> 3;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    if (xlim && (ox & ~63) >= xlim[b]) continue;      // ragged line: the 64-column groups the MFMA kernel below skips (nothing downstream reads them)
    float win[4][4];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        const int yy = 2 * oy - 1 + dy, xx = 2 * ox - 1 + dx;
        float v = 0.f;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
          const size_t o = ((size_t)b * H + yy) * W + xx;
          v = split ? rbf2f(in[o * 2]) + rbf2f(in[o * 2 + 1]) : rbf2f(in[o]);
        }
        win[dy][dx] = v;
      }
    float best[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = cg * 8 + k;
      const float* wk = sw + c * 9;
      float m = 0.f;
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          float a = 0.f;
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) a = fmaf(win[py + r][px + s2], wk[r * 3 + s2], a);
          a = fmaxf(a + sw[576 + c], 0.f);
          if (!split) a = rbf2f(rf2bf(a));  // bf16 contract: the conv output is stored in bf16 before pooling
          m = (py == 0 && px == 0) ? a : fmaxf(m, a);
        }
      best[k] = m;
    }
    uint32_t hb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) hb[k] = rf2bf(best[k]);
    u32x4 o;
    o.x = hb[0] | (hb[1] << 16); o.y = hb[2] | (hb[3] << 16); o.z = hb[4] | (hb[5] << 16); o.w = hb[6] | (hb[7] << 16);
    const size_t pix = ((size_t)b * Ho + oy) * Wo + ox;
    if (!split) {
      *reinterpret_cast<u32x4*>(out + pix * 64 + cg * 8) = o;
    } else {
      uint32_t lb[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) lb[k] = rf2bf(best[k] - rbf2f(hb[k]));
      u32x4 ol;
      ol.x = lb[0] | (lb[1] << 16); ol.y = lb[2] | (lb[3] << 16); ol.z = lb[4] | (lb[5] << 16); ol.w = lb[6] | (lb[7] << 16);
      *reinterpret_cast<u32x4*>(out + pix * 128 + cg * 8) = o;
      *reinterpret_cast<u32x4*>(out + pix * 128 + 64 + cg * 8) = ol;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// bf16 fast path of conv0 + pool on the matrix cores.  The VALU kernel above already runs at ~85 % of the fp32 FMA
// peak (57 GFMA per 4096 lines in 1.7 ms); as a GEMM the layer is [pixels] x [K = 9 taps, padded to 16] x [64]: one
// v_mfma_f32_32x32x16_bf16 per 32 pixels and 32 channels.  The 32 rows of an M tile are 8 pooling windows x 4 pixels
// (row m = 4 w + 2 dy + dx), so the four conv outputs of a window are the four accumulators (r & 3) of ONE lane and the
// 2x2 max is in-lane; the weights are two B fragments held in registers; A fragments are gathered from a gray patch in
// LDS (lane q = 0: taps 0..7, q = 1: tap 8 and zeros).  Same bf16 contract: bias, ReLU, round to bf16, then max.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crnn_conv0_pool_mfma_kernel(const bf16_t* __restrict__ in, int n, int H, int W,
                                                                    const float* __restrict__ w64x9,
                                                                    const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                                    const int* __restrict__ xlim) {
  a16_kernel_enter();
  constexpr int PR = 4, PC = 64;                  // pooled rows x cols per workgroup
  constexpr int LW = 2 * PC + 2, LH = 2 * PR + 2; // gray patch with a 1-pixel halo
  __shared__ bf16_t sg[LH * LW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  const int Ho = H / 2, Wo = W / 2;
  const int tcx = (Wo + PC - 1) / PC, tcy = (Ho + PR - 1) / PR;
  int L = blockIdx.x;
  const int tx = L % tcx;
  L /= tcx;
  const int ty = L % tcy;
  const int b = L / tcy;
  const int oy0 = ty * PR, ox0 = tx * PC;
  if (xlim && ox0 >= xlim[b]) return;             // ragged line: nothing downstream reads these pooled columns
  for (int i = tid; i < LH * LW; i += 256) {
    const int yy = 2 * oy0 - 1 + i / LW, xx = 2 * ox0 - 1 + i % LW;
    sg[i] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? in[((size_t)b * H + yy) * W + xx] : (bf16_t)0;
  }
  // B fragments: lane = output channel lx (+32), k = q*8 .. q*8+7 -> taps (weights already rounded to bf16 by the packer)
  bf16x8 bw[2];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    uint32_t pk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k0 = q * 8 + 2 * i, k1 = k0 + 1;
      const uint32_t a0 = k0 < 9 ? rf2bf(w64x9[(nh * 32 + lx) * 9 + k0]) : 0u;
      const uint32_t a1 = k1 < 9 ? rf2bf(w64x9[(nh * 32 + lx) * 9 + k1]) : 0u;
      pk[i] = a0 | (a1 << 16);
    }
    const u32x4 v = {pk[0], pk[1], pk[2], pk[3]};
    bw[nh] = __builtin_bit_cast(bf16x8, v);
  }
  const float bs[2] = {bias[lx], bias[32 + lx]};
  __syncthreads();
  // M tiles of this workgroup: PR row pairs x (PC / 8) groups of 8 windows; wave w takes tiles w, w + 4, ...
  const int wnd = lx >> 2, dy = (lx >> 1) & 1, dx = lx & 1;      // this lane's A row: window, pixel inside it
  for (int mt = wave; mt < PR * (PC / 8); mt += 4) {
    const int pr = mt / (PC / 8), pc0 = (mt % (PC / 8)) * 8;
    const int ly = 2 * pr + dy, lxx = 2 * (pc0 + wnd) + dx;       // conv pixel in patch coordinates minus the halo
    uint32_t pk[4] = {0u, 0u, 0u, 0u};
    if (q == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k0 = 2 * i, k1 = 2 * i + 1;
        pk[i] = (uint32_t)sg[(ly + k0 / 3) * LW + lxx + k0 % 3] | ((uint32_t)sg[(ly + k1 / 3) * LW + lxx + k1 % 3] << 16);
      }
    } else {
      pk[0] = (uint32_t)sg[(ly + 2) * LW + lxx + 2];
    }
    const u32x4 av = {pk[0], pk[1], pk[2], pk[3]};
    const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = mfma_32x32x16_a16(a, bw[nh], acc);
      // accumulator r: row (r & 3) + 8 (r >> 2) + 4 q = window 2 (r >> 2) + q, pixel r & 3
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = rbf2f(rf2bf(fmaxf(acc[g * 4 + j] + bs[nh], 0.f)));
          m = j == 0 ? v : fmaxf(m, v);
        }
        const int oy = oy0 + pr, ox = ox0 + pc0 + 2 * g + q;
        if (oy < Ho && ox < Wo) out[(((size_t)b * Ho + oy) * Wo + ox) * 64 + nh * 32 + lx] = (bf16_t)rf2bf(m);
      }
    }
  }
}

int pt_launch_crnn_conv0_pool(const bf16_t* in, int n, int H, int W, const float* w64x9, const float* bias, int split,
                              bf16_t* out, hipStream_t s, const int* xlim) {
  PT_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv0: H, W must be even");
  static int use_mfma = -1;      // PT_CONV0_MFMA=0: the VALU kernel in bf16 mode too (A/B switch)
  if (use_mfma < 0) {
    const char* ev = getenv("PT_CONV0_MFMA");
    use_mfma = ev ? atoi(ev) : 1;
  }
  if (!split && use_mfma) {
    const long long nb = (long long)n * ((H / 2 + 3) / 4) * ((W / 2 + 63) / 64);
    hipLaunchKernelGGL(crnn_conv0_pool_mfma_kernel, dim3((unsigned)nb), dim3(256), 0, s, in, n, H, W, w64x9, bias, out, xlim);
    PT_HIP_CHECK(hipGetLastError());
    return PT_OK;
  }
  const long long total = (long long)n * (H / 2) * (W / 2) * 8;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(crnn_conv0_pool_kernel, dim3(blocks), dim3(256), 0, s, in, n, H, W, w64x9, bias, split, out, xlim);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// MaxPool2d(kernel = stride = (kh, kw)) on NHWC bf16, kh in {2}, kw in {1, 2}.  h2c: the pooled rows are written
// as extra channel groups ([n][Wo][Ho*C]) so that the following (Ho,1)-kernel conv is a plain 1x1 GEMM.
// split: channels are [hi(C) | lo(C)] pairs and the max is taken on hi + lo.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kxk_kernel(const bf16_t* __restrict__ in, int n, int H, int W, int C,
                                                           int kh, int kw, int h2c, int split,
                                                           bf16_t* __restrict__ out) {
  a16_kernel_enter();
  const int Ho = H / kh, Wo = W / kw, cgn = C >> 3;
  const int cs = split ? 2 * C : C;
  const long long total = (long long)n * Ho * Wo * cgn;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cgn);
    long long t = i / cgn;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    uint32_t bh[8], bl[8];
    float bv[8];
    bool first = true;
    for (int dy = 0; dy < kh; ++dy)
      for (int dx = 0; dx < kw; ++dx) {
        const bf16_t* px = in + (((size_t)b * H + oy * kh + dy) * W + ox * kw + dx) * cs + g * 8;
        const u32x4 vh = *reinterpret_cast<const u32x4*>(px);
        u32x4 vl = {0u, 0u, 0u, 0u};
        if (split) vl = *reinterpret_cast<const u32x4*>(px + C);
        const uint32_t hw[4] = {vh.x, vh.y, vh.z, vh.w}, lw[4] = {vl.x, vl.y, vl.z, vl.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t hb = (k & 1) ? (hw[k >> 1] >> 16) : (hw[k >> 1] & 0xFFFFu);
          const uint32_t lb = (k & 1) ? (lw[k >> 1] >> 16) : (lw[k >> 1] & 0xFFFFu);
          const float v = rbf2f(hb) + rbf2f(lb);
          if (first || v > bv[k]) { bv[k] = v; bh[k] = hb; bl[k] = lb; }
        }
        first = false;
      }
    u32x4 oh, ol;
    oh.x = bh[0] | (bh[1] << 16); oh.y = bh[2] | (bh[3] << 16); oh.z = bh[4] | (bh[5] << 16); oh.w = bh[6] | (bh[7] << 16);
    ol.x = bl[0] | (bl[1] << 16); ol.y = bl[2] | (bl[3] << 16); ol.z = bl[4] | (bl[5] << 16); ol.w = bl[6] | (bl[7] << 16);
    size_t o;
    int lo_off;
    if (h2c) {
      const int Ct = Ho * C;
      o = ((size_t)b * Wo + ox) * (split ? 2 * Ct : Ct) + oy * C + g * 8;
      lo_off = Ct;
    } else {
      o = (((size_t)b * Ho + oy) * Wo + ox) * cs + g * 8;
      lo_off = C;
    }
    *reinterpret_cast<u32x4*>(out + o) = oh;
    if (split) *reinterpret_cast<u32x4*>(out + o + lo_off) = ol;
  }
}

int pt_launch_maxpool_kxk(const bf16_t* in, int n, int H, int W, int C, int kh, int kw, int h2c, int split, bf16_t* out,
                          hipStream_t s) {
  PT_REQUIRE(C % 8 == 0 && H % kh == 0 && W % kw == 0, "maxpool: bad shape");
  const long long total = (long long)n * (H / kh) * (W / kw) * (C / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(maxpool_kxk_kernel, dim3(blocks), dim3(256), 0, s, in, n, H, W, C, kh, kw, h2c, split, out);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// LSTM, one direction, 32 text lines per workgroup, all T steps inside the kernel.
//   gates_t = gx[line, t, dir] (input projection + both biases, computed beforehand by the 1x1 GEMM kernel)
//             + h_{t-1} . W_hh^T                                   (MFMA: M = 32 lines, N = 1024, K = 256)
//   i, f, g, o = split(gates);  c = sig(f) c + sig(i) tanh(g);  h = sig(o) tanh(c)
// Wave w owns hidden units [64w, 64w+64): its 8 MFMA tiles are (gate 0..3) x (32-unit half 0..1), so the four
// gate pre-activations of one (line, unit) sit in the same lane and register index and the cell update is
// lane-local; c stays in registers (fp32) for the whole sequence; h goes through LDS (double-buffered, bf16 or
// hi/lo pair) as next step's A operand and to HBM as the layer output.  W_hh streams from L2 in MFMA-fragment order
// (packed that way by weights.pack_crnn), one contiguous 1 KB record per load instruction.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  // 2 / (1 + 2^(-2 log2(e) x)) - 1 ; saturates cleanly for large |x| (exp2 -> 0 or inf)
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.f;
}

#ifndef PT_LSTM_ABL
#define PT_LSTM_ABL 0
#endif
// NG = 2: two independent groups of 32 lines per workgroup (waves 0-3 and 4-7, two waves per SIMD): while one group waits
// for a W_hh burst from L2 the other multiplies -- with one group a SIMD holds one wave and every round trip is exposed --
// and a launch needs half the workgroups (5082 lines x 2 directions: 160 instead of 318, one round on 256 CUs instead of two).
template <int SPLIT, int NG = 1>
__global__ __launch_bounds__(256 * NG, 1) void lstm_dir_kernel(const bf16_t* __restrict__ gx, const bf16_t* __restrict__ whh,
                                                           bf16_t* __restrict__ hout, int B, int T) {
  a16_kernel_enter();
  constexpr int NP = SPLIT ? 2 : 1;
  constexpr int HROW = 264;  // 256 + 8 bf16: 528-byte rows = 33 16-byte slots (odd) -> conflict-free b128 reads
  __shared__ __attribute__((aligned(16))) bf16_t hbuf_all[NG][NP][32][HROW];  // read by the group's waves (MFMA), then rewritten
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 8, tid = threadIdx.x & 255, wave = tid >> 6;
  bf16_t (*hbuf)[32][HROW] = hbuf_all[grp];
  const int lx = lane & 31, q = lane >> 5;
  const int line0 = (blockIdx.x * NG + grp) * 32, dir = blockIdx.y;
  const int gcs = (SPLIT ? 2 : 1) * 2048;  // gx channels per (line, t): [dir0 1024 | dir1 1024] (x2 for hi|lo)
  const int hcs = (SPLIT ? 2 : 1) * 512;   // hout channels per (line, t): [fw 256 | bw 256] (x2 for hi|lo)
  const bf16_t* whh_d = whh + (size_t)dir * 1024 * 256;           // hi part; lo part at + 2*1024*256
  for (int i = tid; i < NP * 32 * HROW; i += 256) (&hbuf[0][0][0])[i] = 0;
  float c[2][16];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[h][r] = 0.f;
  __syncthreads();
  // h_t goes to HBM from its LDS copy at the top of the NEXT step: 16-byte row pieces (4 per thread) instead of 32
  // two-byte stores per lane from the MFMA layout; the copy is stable until this step's cell update (after the barrier)
  auto flush_h = [&](int tt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256, m = idx >> 5, j = idx & 31;
      const int line = line0 + m;
      if (line >= B) continue;
      bf16_t* hp = hout + ((size_t)line * T + tt) * hcs + dir * 256 + j * 8;
      *reinterpret_cast<u32x4*>(hp) = *reinterpret_cast<const u32x4*>(&hbuf[0][m][j * 8]);
      if (SPLIT) *reinterpret_cast<u32x4*>(hp + 512) = *reinterpret_cast<const u32x4*>(&hbuf[NP - 1][m][j * 8]);
    }
  };
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    if (s > 0 && PT_LSTM_ABL != 4) flush_h(dir ? t + 1 : t - 1);
    // input-projection terms of this step: gx is laid out [line][t][dir][unit][gate] (gate fastest, set up by the
    // weight packer), so the four gates of a (line, unit) are one 8-byte load; issued before the MFMA phase
    // NG = 2 (256 registers per wave instead of 512): the gx terms are fetched per 32-unit half AFTER the MFMA phase (the
    // other group's MFMAs cover the latency) and W_hh comes in bursts of two k-steps instead of four
    u32x2 gxv[2][16], gxl[2][16];
    auto load_gx = [&](int h) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * q;
        const int line = line0 + m;
        const int lc = line < B ? line : B - 1;
        const bf16_t* gp = gx + ((size_t)lc * T + t) * gcs + dir * 1024 + (wave * 64 + h * 32 + lx) * 4;
#if PT_LSTM_ABL == 3      /* ablation: no gx traffic (wrong results) */
        gxv[h][r] = u32x2{(uint32_t)s, (uint32_t)r};
        if (SPLIT) gxl[h][r] = gxv[h][r];
        (void)gp;
#else
        gxv[h][r] = *reinterpret_cast<const u32x2*>(gp);
        if (SPLIT) gxl[h][r] = *reinterpret_cast<const u32x2*>(gp + 2048);
#endif
      }
    };
    if (NG == 1) {
      load_gx(0);
      load_gx(1);
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][h][r] = 0.f;
    constexpr int NPASS = SPLIT ? 3 : 1;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int pa = (pass == 1) ? 1 : 0;  // A part: h_hi, h_lo, h_hi
      const bf16_t* wsrc = whh_d + (pass == 2 ? (size_t)2 * 1024 * 256 : 0);
      // W_hh streams from L2 and the loop is latency-bound: issue the 32 fragment loads of a quarter step (128
      // VGPRs) in one burst, then multiply -- four L2 round trips per pass instead of sixteen
      constexpr int KQ = NG == 2 ? 2 : 4;      // k-steps per burst
#pragma unroll 1
      for (int half = 0; half < 16 / KQ; ++half) {
        bf16x8 bq[KQ][4][2];
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              // fragment-ordered by the packer: [wave][half][kq][gate][h][lane][8] = column gate*256 + wave*64 +
              // h*32 + lx, k = (half*4 + kq)*16 + q*8 ..+8
#if PT_LSTM_ABL == 1      /* ablation: no W_hh traffic (wrong results) */
              bq[kq][g][h] = *reinterpret_cast<const bf16x8*>(&hbuf[pa][lx][((kq * 4 + g) * 2 + h) * 8]);
#else
              bq[kq][g][h] = *reinterpret_cast<const bf16x8*>(
                  wsrc + (size_t)((((wave * 16 + half * KQ + kq) * 4 + g) * 2 + h) * 64 + lane) * 8);
#endif
            }
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(&hbuf[pa][lx][(half * KQ + kq) * 16 + q * 8]);
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h)
              acc[g][h] = mfma_32x32x16_a16(a, bq[kq][g][h], acc[g][h]);
        }
      }
    }
    __syncthreads();  // every wave has finished reading h_{t-1}
    // cell update (lane-local) + publish h
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int unit = wave * 64 + h * 32 + lx;
      if (NG == 2) load_gx(h);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * q;
        const int line = line0 + m;
        float gi = acc[0][h][r] + rbf2f(gxv[h][r].x & 0xFFFFu);
        float gf = acc[1][h][r] + rbf2f(gxv[h][r].x >> 16);
        float gg = acc[2][h][r] + rbf2f(gxv[h][r].y & 0xFFFFu);
        float go = acc[3][h][r] + rbf2f(gxv[h][r].y >> 16);
        if (SPLIT) {
          gi += rbf2f(gxl[h][r].x & 0xFFFFu); gf += rbf2f(gxl[h][r].x >> 16);
          gg += rbf2f(gxl[h][r].y & 0xFFFFu); go += rbf2f(gxl[h][r].y >> 16);
        }
        // sigmoid / tanh through the hardware exp2 + reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp, far inside the 1e-3
        // contract); tanh(x) = 2 sigmoid(2x) - 1
#if PT_LSTM_ABL == 2      /* ablation: no transcendentals (wrong results) */
        const float cn = gf * c[h][r] + gi * gg;
        c[h][r] = cn;
        const float hn = go * cn;
#else
        const float si = fast_sigmoid(gi), sf = fast_sigmoid(gf), so = fast_sigmoid(go);
        const float cn = sf * c[h][r] + si * fast_tanh(gg);
        c[h][r] = cn;
        const float hn = so * fast_tanh(cn);
#endif
        const uint32_t hb = rf2bf(hn);
        hbuf[0][m][unit] = (bf16_t)hb;
        uint32_t lb = 0;
        if (SPLIT) {
          lb = rf2bf(hn - rbf2f(hb));
          hbuf[NP - 1][m][unit] = (bf16_t)lb;
        }
        (void)line;
      }
    }
    __syncthreads();
  }
  if (T > 0 && PT_LSTM_ABL != 4) flush_h(dir ? 0 : T - 1);
}

// ---------------------------------------------------------------------------------------------------
// bf16 fast path of the LSTM: the same step as lstm_dir_kernel<0>, but the gate pre-activations gx of step s + 1 are
// pre-fetched into LDS (global_load_lds, 64 KB per step, two buffers) while step s runs.  In the register version every
// workgroup of the chip asks HBM for its 128 scattered 256-byte gx rows at the same moment of every step and -- vector
// loads return in order -- the first W_hh burst cannot be consumed before they have arrived (7.8 us of a 25 us step by
// ablation).  Here the DMA of step s + 1 is issued right after the mid-step barrier of step s, has the cell update and
// the four W_hh bursts of the next step to land, and is awaited (vmcnt(0), already satisfied) at the next mid-step
// barrier.  Barriers are raw s_barrier + lgkmcnt(0): a __syncthreads() fence would wait for the DMA in flight.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void lstm_dir_dma_kernel(const bf16_t* __restrict__ gx, const bf16_t* __restrict__ whh,
                                                               bf16_t* __restrict__ hout, int B, int T) {
  a16_kernel_enter();
  constexpr int HROW = 264;
  extern __shared__ __attribute__((aligned(16))) char lsm[];
  bf16_t (*hbuf)[HROW] = reinterpret_cast<bf16_t (*)[HROW]>(lsm);                     // [32][HROW]
  char* gxs = lsm + 32 * HROW * 2;                                                    // [2][32 lines][2048 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  const int line0 = blockIdx.x * 32, dir = blockIdx.y;
  constexpr int gcs = 2048, hcs = 512;
  const bf16_t* whh_d = whh + (size_t)dir * 1024 * 256;
  for (int i = tid; i < 32 * HROW; i += 256) (&hbuf[0][0])[i] = 0;
  float c[2][16];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[h][r] = 0.f;
  // DMA of one step: 64 wave-instructions of 1 KB (line m = ii >> 1, half ii & 1), 16 per wave
  auto issue_gx = [&](int tt, int buf) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ii = wave * 16 + k, m = ii >> 1, part = ii & 1;
      const int line = line0 + m;
      const int lc = line < B ? line : B - 1;
      const bf16_t* src = gx + ((size_t)lc * T + tt) * gcs + dir * 1024 + part * 512 + lane * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(gxs + buf * 65536 + m * 2048 + part * 1024), 16, 0, 0);
    }
  };
  auto flush_h = [&](int tt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256, m = idx >> 5, j = idx & 31;
      const int line = line0 + m;
      if (line >= B) continue;
      *reinterpret_cast<u32x4*>(hout + ((size_t)line * T + tt) * hcs + dir * 256 + j * 8) =
          *reinterpret_cast<const u32x4*>(&hbuf[m][j * 8]);
    }
  };
  if (T > 0) issue_gx(dir ? T - 1 : 0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    if (s > 0) flush_h(dir ? t + 1 : t - 1);
    f32x16 acc[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][h][r] = 0.f;
#pragma unroll 1
    for (int half = 0; half < 4; ++half) {
      bf16x8 bq[4][4][2];
#pragma unroll
      for (int kq = 0; kq < 4; ++kq)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            bq[kq][g][h] = *reinterpret_cast<const bf16x8*>(
                whh_d + (size_t)(((((wave * 4 + half) * 4 + kq) * 4 + g) * 2 + h) * 64 + lane) * 8);
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(&hbuf[lx][(half * 4 + kq) * 16 + q * 8]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            acc[g][h] = mfma_32x32x16_a16(a, bq[kq][g][h], acc[g][h]);
      }
    }
    // every wave has finished reading h_{t-1}; this step's gx (issued one step ago) has landed
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 1 < T) issue_gx(dir ? t - 1 : t + 1, (s + 1) & 1);
    const char* gcur = gxs + (s & 1) * 65536;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int unit = wave * 64 + h * 32 + lx;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * q;
        const u32x2 gv = *reinterpret_cast<const u32x2*>(gcur + m * 2048 + unit * 8);
        const float gi = acc[0][h][r] + rbf2f(gv.x & 0xFFFFu);
        const float gf = acc[1][h][r] + rbf2f(gv.x >> 16);
        const float gg = acc[2][h][r] + rbf2f(gv.y & 0xFFFFu);
        const float go = acc[3][h][r] + rbf2f(gv.y >> 16);
        const float si = fast_sigmoid(gi), sf = fast_sigmoid(gf), so = fast_sigmoid(go);
        const float cn = sf * c[h][r] + si * fast_tanh(gg);
        c[h][r] = cn;
        hbuf[m][unit] = (bf16_t)rf2bf(so * fast_tanh(cn));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (T > 0) flush_h(dir ? 0 : T - 1);
}

// ---------------------------------------------------------------------------------------------------
// Weight-stationary LSTM (bf16 mode).  lstm_dir_kernel streams the 512 KB W_hh of a direction from L2 in every one of
// the T steps and every workgroup (128 MB per step over the chip: the step is bound by that traffic).
// Here four workgroups ("members") share a block of 128 lines; member j keeps the W_hh columns of hidden units
// [64 j, 64 j + 64) -- 128 KB, already contiguous in the packer's fragment order -- in LDS for the whole sequence,
// computes those units' gates and cell update for all 128 lines, and publishes its quarter of h_t (bf16, in MFMA
// A-fragment order) to an L2-resident exchange buffer; the members meet once per step on four step counters.
//   * exchange: agent-scope relaxed atomic 8-byte stores / loads (global_* sc1: L1 bypassed, coherent across XCDs), the
//     writer drains its stores (vmcnt(0)), barriers, then one lane bumps the member's counter; readers poll the four
//     counters (one lane each), barrier, then load the fragments.  Two parities: a member can only be one step ahead.
//   * wave w: 32-unit half h = w & 1, line tiles 2 (w >> 1) and 2 (w >> 1) + 1: a W fragment read from LDS feeds two MFMAs.
//   * all workgroups of a launch must be co-resident (1 per CU, 144 KB of LDS): the launcher issues at most
//     num_cu / 8 clusters per direction per launch; a member that waits > 2^22 polls gives up (sets *err) instead of hanging.
// ---------------------------------------------------------------------------------------------------
// MI = 32-line tiles per wave: 2 (128 lines per cluster, a W fragment read from LDS feeds two MFMAs) or 3 (192 lines per
// cluster: the tiles go through the MFMAs one after the other so that the accumulators and the A fragments of one tile
// are live at a time; 1.5x the LDS reads, but a launch holds 6144 lines instead of 4096 -- the step is latency-bound, so
// a launch costs about the same whatever it holds).  Same sums in the same order for every MI.
constexpr int CLL_MAX = 192;  // lines per cluster at MI = 3 (scratch is sized for it)

// (global, not generic: the opaque per-step pointers below would otherwise make every h fragment load a FLAT instruction, and with FLAT loads in flight hipcc
// waits lgkmcnt(0) in front of every LDS-fed MFMA)
#ifndef PT_CL_ABL
#define PT_CL_ABL 0      // ablation bits of lstm_cluster_kernel (timing only, results wrong): 1 no MFMA loop, 2 no transcendentals in the cell update, 4 no waiting for the other members
#endif
typedef const __attribute__((address_space(1))) unsigned long long* gptr_u64;

template <int MI>
__global__ __launch_bounds__(256, 1) void lstm_cluster_kernel(const bf16_t* __restrict__ gx, const bf16_t* __restrict__ whh,
                                                               bf16_t* __restrict__ hout, int B, int T, int ncl,
                                                               bf16_t* __restrict__ hx, int* __restrict__ flags,
                                                               int* __restrict__ err) {
  a16_kernel_enter();
  constexpr int CLL = 64 * MI;                                 // lines per cluster
  extern __shared__ __attribute__((aligned(16))) char lsm[];
  char* wl = lsm;                                              // [16 ks][4 g][2 h][64 lanes][16 B] = 128 KB
  constexpr int SROW = 72;                                     // 64 + 8 bf16: 144-byte rows
  bf16_t* stage = reinterpret_cast<bf16_t*>(lsm + 131072);     // [CLL][SROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  const int L = blockIdx.x, dir = blockIdx.y;
  const int member = (L >> 3) & 3, cl = (L >> 5) * 8 + (L & 7);   // the 4 members of a cluster: ids b, b+8, b+16, b+24
  if (cl >= ncl) return;
  const int line0 = cl * CLL;
  const int h = wave & 1, mp = wave >> 1;
  {   // W slice of this member: one contiguous 128 KB block of the fragment-ordered tensor
    const u32x4* src = reinterpret_cast<const u32x4*>(whh + (size_t)dir * 1024 * 256 + (size_t)member * 65536);
    for (int i = tid; i < 8192; i += 256) reinterpret_cast<u32x4*>(wl)[i] = src[i];
  }
  unsigned long long* hxc = reinterpret_cast<unsigned long long*>(hx + (size_t)(dir * ncl + cl) * 2 * (CLL * 256));
  int* fl = flags + (dir * ncl + cl) * 4;
  float c[MI][16];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[mi][r] = 0.f;
  bool dead = false;
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    // gate inputs of this step (independent of the other members): 8 bytes per (line, unit)
    u32x2 gv[MI][16];
    int last = B - 1;
    asm volatile("" : "+v"(last));      // opaque per step: the 16 MI row addresses are recomputed (a min and a mad each) instead
                                        // of living in 32 MI registers across the whole sequence
    const bf16_t* gxt = gx + (size_t)t * 2048 + dir * 1024 + (member * 64 + h * 32 + lx) * 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (MI * mp + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * q;
        const int line = line0 + m;
        const int lc = line < last ? line : last;
        gv[mi][r] = *reinterpret_cast<const u32x2*>(gxt + (size_t)lc * T * 2048);
      }
    if (s > 0 && !(PT_CL_ABL & 4)) {
      if (tid < 4 && !dead) {
        int spins = 0;
        while (__hip_atomic_load(fl + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < s) {
          if (++spins > (1 << 22)) { dead = true; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
    }
    const unsigned long long* hp = hxc + (size_t)((s - 1) & 1) * (CLL * 256 / 4);
    asm volatile("" : "+s"(hp));        // opaque per step, for the same reason: no per-fragment addresses kept across steps
    // cell update of line tile mi (lane-local) from its four gate accumulators, h to the staging tile
    auto cell = [&](int mi, const f32x16* acc) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (MI * mp + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * q;
        const float gi = acc[0][r] + rbf2f(gv[mi][r].x & 0xFFFFu);
        const float gf = acc[1][r] + rbf2f(gv[mi][r].x >> 16);
        const float gg = acc[2][r] + rbf2f(gv[mi][r].y & 0xFFFFu);
        const float go = acc[3][r] + rbf2f(gv[mi][r].y >> 16);
#if PT_CL_ABL & 2
        const float si = gi * 0.25f, sf = gf * 0.25f, so = go * 0.25f;
        const float cn = sf * c[mi][r] + si * gg;
        c[mi][r] = cn;
        stage[m * SROW + h * 32 + lx] = (bf16_t)rf2bf(so * cn);
#else
        const float si = fast_sigmoid(gi), sf = fast_sigmoid(gf), so = fast_sigmoid(go);
        const float cn = sf * c[mi][r] + si * fast_tanh(gg);
        c[mi][r] = cn;
        stage[m * SROW + h * 32 + lx] = (bf16_t)rf2bf(so * fast_tanh(cn));
#endif
      }
    };
    if constexpr (MI == 2) {
      f32x16 acc[2][4];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][g][r] = 0.f;
      if (s > 0) {
        // A fragments of both line tiles, all 16 k-steps (2 x 16 x 16 B per lane), then the MFMAs
        unsigned long long af[2][16][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) {
            const gptr_u64 pp = (gptr_u64)hp + ((size_t)((2 * mp + mi) * 16 + ks) * 64 + lane) * 2;
            af[mi][ks][0] = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            af[mi][ks][1] = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        // the four gate fragments of k-step ks + 1 are requested between the eight MFMAs of k-step ks (pinned: left alone, with ONE wave per SIMD,
        // hipcc emits read, wait, MFMA, MFMA on one fragment register -- 64 exposed LDS round trips per step)
        bf16x8 wf[2][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) wf[0][g] = *reinterpret_cast<const bf16x8*>(wl + (g * 2 + h) * 1024 + lane * 16);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          bf16x8 a[2];
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const unsigned long long two[2] = {af[mi][ks][0], af[mi][ks][1]};
            a[mi] = __builtin_bit_cast(bf16x8, two);
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (ks + 1 < 16) wf[(ks + 1) & 1][g] = *reinterpret_cast<const bf16x8*>(wl + (((ks + 1) * 4 + g) * 2 + h) * 1024 + lane * 16);
            acc[0][g] = mfma_32x32x16_a16(a[0], wf[ks & 1][g], acc[0][g]);
            acc[1][g] = mfma_32x32x16_a16(a[1], wf[ks & 1][g], acc[1][g]);
          }
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 64, 0);      // both tiles' A fragments (global loads) first
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (ks + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
      }
      cell(0, acc[0]);
      cell(1, acc[1]);
    } else {
      // one tile at a time; the A fragments of the next tile are fetched under the MFMAs of this one (two buffers), and a
      // scheduling barrier keeps the compiler from hoisting the third tile's loads on top (that spilled)
      unsigned long long af[2][16][2];
      auto fetch = [&](int mi, int buf) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const gptr_u64 pp = (gptr_u64)hp + ((size_t)((MI * mp + mi) * 16 + ks) * 64 + lane) * 2;
          af[buf][ks][0] = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          af[buf][ks][1] = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      };
      if (s > 0) fetch(0, 0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        f32x16 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        if (s > 0 && !(PT_CL_ABL & 1)) {
          if (mi + 1 < MI) fetch(mi + 1, (mi + 1) & 1);
          bf16x8 wf[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) wf[g] = *reinterpret_cast<const bf16x8*>(wl + (g * 2 + h) * 1024 + lane * 16);
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) {
            const unsigned long long two[2] = {af[mi & 1][ks][0], af[mi & 1][ks][1]};
            const bf16x8 a = __builtin_bit_cast(bf16x8, two);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              acc[g] = mfma_32x32x16_a16(a, wf[g], acc[g]);
              if (ks + 1 < 16) wf[g] = *reinterpret_cast<const bf16x8*>(wl + (((ks + 1) * 4 + g) * 2 + h) * 1024 + lane * 16);
            }
          }
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
          for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (ks + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        cell(mi, acc);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    // publish: CLL lines x 8 pieces of 16 B; piece i of line m = units 64 member + 8 i .. + 8.  The exchange stores go
    // first and only they are drained before the counter is bumped; the layer output (HBM) follows behind the counter,
    // off the step's critical chain
    unsigned long long* hw = hxc + (size_t)(s & 1) * (CLL * 256 / 4);
    int tio = tid;
    asm volatile("" : "+v"(tio));       // opaque per step: the store addresses below are recomputed, not kept (or spilled)
    u32x4 pv[2 * MI];
#pragma unroll
    for (int i = 0; i < 2 * MI; ++i) {
      const int idx = tio + i * 256, m = idx >> 3, pc = idx & 7;
      pv[i] = *reinterpret_cast<const u32x4*>(stage + m * SROW + pc * 8);
      const int ks = 4 * member + (pc >> 1), qq = pc & 1;
      unsigned long long* dp = hw + ((size_t)((m >> 5) * 16 + ks) * 64 + qq * 32 + (m & 31)) * 2;
      __hip_atomic_store(dp, (unsigned long long)pv[i].x | ((unsigned long long)pv[i].y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dp + 1, (unsigned long long)pv[i].z | ((unsigned long long)pv[i].w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(fl + member, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < 2 * MI; ++i) {
      const int idx = tio + i * 256, m = idx >> 3, pc = idx & 7;
      const int line = line0 + m;
      if (line < B) *reinterpret_cast<u32x4*>(hout + ((size_t)line * T + t) * 512 + dir * 256 + member * 64 + pc * 8) = pv[i];
    }
  }
}

template <int MI>
static int launch_lstm_cluster(const bf16_t* gx, const bf16_t* whh, bf16_t* hout, int B, int T, int max_cl, bf16_t* hx, int* flags,
                               int* err, hipStream_t s) {
  constexpr int CLL = 64 * MI;
  constexpr int SMEM = 131072 + CLL * 72 * 2;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_cluster_kernel<MI>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_done = true;
  }
  // equal chunks: a launch costs the same whatever it holds, so the last one should not be a sliver
  const int nlaunch = (B + max_cl * CLL - 1) / (max_cl * CLL);
  const int per = (((B + nlaunch - 1) / nlaunch) + CLL - 1) / CLL * CLL;
  for (int b0 = 0; b0 < B; b0 += per) {
    const int nb = (B - b0) < per ? (B - b0) : per;
    const int ncl = (nb + CLL - 1) / CLL;
    PT_HIP_CHECK(hipMemsetAsync(flags, 0, (size_t)(2 * max_cl * 4) * sizeof(int), s));
    hipLaunchKernelGGL(lstm_cluster_kernel<MI>, dim3(((ncl + 7) / 8) * 32, 2), dim3(256), SMEM, s, gx + (size_t)b0 * T * 2048, whh,
                       hout + (size_t)b0 * T * 512, nb, T, ncl, hx, flags, err);
  }
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// Weight-stationary LSTM of the hi/lo (BF16X3) mode: EIGHT workgroups share a block of lines.  The streaming kernel re-reads
// the 1 MB of (hi, lo) W_hh of a direction from L2 in every step (94 us per step, 30 ms per 64-page step for the two layers); a
// 64-unit slice of the pair is 256 KB and does not fit LDS, a 32-unit slice (64 KB hi + 64 KB lo) does.  Member j keeps the W_hh
// columns of hidden units [32 j, 32 j + 32) for the whole sequence; its four waves take the cluster's 32-line tiles MI apiece,
// compute gates = h_hi W_hi + h_lo W_hi + h_hi W_lo (per k-step, fp32 accumulate) + gx_hi + gx_lo and the lane-local cell update,
// and publish their 32 units of h_t as (hi, lo) A fragments (k-steps 2 j, 2 j + 1 of every tile) to an L2-resident exchange buffer;
// the members meet once per step on eight counters, exactly like lstm_cluster_kernel.  Members of a cluster sit on one XCD
// (ids b, b + 8, ..., b + 56).  The packer's fragment order already has the 1 KB records this needs: member j = (wave j >> 1,
// half j & 1) of the four-member layout.
// ---------------------------------------------------------------------------------------------------
// NW = 8 waves (two per SIMD, 256 registers each): while one wave waits for its tile's gate inputs and fragments the other multiplies --
// with NW = 4 (one wave per SIMD) a tile's fetch, 192 MFMAs, cell update and publish run back to back (42.6 us per step at three tiles)
template <int MI, int NW>
__global__ __launch_bounds__(64 * NW, 1) void lstm_cluster8_x3_kernel(const bf16_t* __restrict__ gx, const bf16_t* __restrict__ whh,
                                                                   bf16_t* __restrict__ hout, int B, int T, int ncl,
                                                                   unsigned long long* __restrict__ hx, int* __restrict__ flags,
                                                                   int* __restrict__ err) {
  a16_kernel_enter();
  constexpr int CLL = 32 * NW * MI, NTILE = NW * MI, NTHR = 64 * NW;      // lines / 32-line tiles per cluster
  extern __shared__ __attribute__((aligned(16))) char lsm[];
  char* wl = lsm;                                              // [2 (hi, lo)][16 ks][4 g][64 lanes][16 B] = 128 KB
  constexpr int SROW = 32;                                     // un-padded 64-byte rows: four consecutive lanes read a row's four 16-byte pieces
  // wave-private staging tile [2 (hi, lo)][32 lines][SROW] = 4 KB: a wave publishes each of its tiles right after the cell update (its own
  // LDS writes are visible to it in order: no barrier); staging whole clusters (2 x CLL x 64 B) would not fit beside the 128 KB of weights
  bf16_t* stage = reinterpret_cast<bf16_t*>(lsm + 131072) + (threadIdx.x >> 6) * (2 * 32 * SROW);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  const int L = blockIdx.x, dir = blockIdx.y;
  const int member = (L >> 3) & 7, cl = (L >> 6) * 8 + (L & 7);
  if (cl >= ncl) return;
  const int line0 = cl * CLL;
  {   // W slice of this member: 64 + 64 records of 1 KB out of the fragment-ordered tensor [hi | lo][dir][wave 4][16 ks][4 g][2 h][64][8]
    for (int i = tid; i < 2 * 64 * 64; i += NTHR) {
      const int pl = i >> 12, rec = (i >> 6) & 63, ln = i & 63;
      const bf16_t* src = whh + (size_t)pl * 2 * 1024 * 256 + (size_t)dir * 1024 * 256 + (size_t)(member >> 1) * 65536 +
                          (size_t)(rec * 2 + (member & 1)) * 512 + ln * 8;
      *reinterpret_cast<u32x4*>(wl + (size_t)pl * 65536 + rec * 1024 + ln * 16) = *reinterpret_cast<const u32x4*>(src);
    }
  }
  // exchange buffer of the cluster: [2 buffers][NTILE][16 ks][2 (hi, lo)][64 lanes][2 u64]
  constexpr size_t XBUF = (size_t)NTILE * 16 * 2 * 64 * 2;
  unsigned long long* hxc = hx + (size_t)(dir * ncl + cl) * 2 * XBUF;
  int* fl = flags + (dir * ncl + cl) * 8;
  float c[MI][16];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[mi][r] = 0.f;
  bool dead = false;
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    int last = B - 1;
    asm volatile("" : "+v"(last));      // opaque per step: row addresses are recomputed, not kept across the sequence
    const bf16_t* gxt = gx + (size_t)t * 4096 + dir * 1024 + (member * 32 + lx) * 4;
    if (s > 0) {
      if (tid < 8 && !dead) {
        int spins = 0;
        while (__hip_atomic_load(fl + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < s) {
          if (++spins > (1 << 22)) { dead = true; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
    }
    const unsigned long long* hp = hxc + (size_t)((s - 1) & 1) * XBUF;
    asm volatile("" : "+s"(hp));
#pragma unroll 1
    for (int mi = 0; mi < MI; ++mi) {
      const int tile = wave * MI + mi;
      // gate inputs of this tile (hi, lo): issued first, they land under the fragment fetch and the MFMAs
      u32x2 gh[16], gl[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * q;
        const int line = line0 + m;
        const int lc = line < last ? line : last;
        const bf16_t* gp = gxt + (size_t)lc * T * 4096;
        gh[r] = *reinterpret_cast<const u32x2*>(gp);
        gl[r] = *reinterpret_cast<const u32x2*>(gp + 2048);
      }
      f32x16 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
      if (s > 0) {
        // NW = 8: the tile's fragments in two bursts of eight k-steps (64 registers instead of 128: the wave has 256, and its
        // SIMD partner covers the second round trip)
        constexpr int KH = NW == 8 ? 8 : 16;
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += KH) {
          unsigned long long af[KH][2][2];
#pragma unroll
          for (int ks = 0; ks < KH; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
              const gptr_u64 pp = (gptr_u64)hp + ((size_t)((tile * 16 + k0 + ks) * 2 + pl) * 64 + lane) * 2;
              af[ks][pl][0] = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              af[ks][pl][1] = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
          for (int ks = 0; ks < KH; ++ks) {
            const unsigned long long th[2] = {af[ks][0][0], af[ks][0][1]}, tl[2] = {af[ks][1][0], af[ks][1][1]};
            const bf16x8 ah = __builtin_bit_cast(bf16x8, th), al = __builtin_bit_cast(bf16x8, tl);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wl + ((k0 + ks) * 4 + g) * 1024 + lane * 16);
              const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wl + 65536 + ((k0 + ks) * 4 + g) * 1024 + lane * 16);
              acc[g] = mfma_32x32x16_a16(ah, bh, acc[g]);
              acc[g] = mfma_32x32x16_a16(al, bh, acc[g]);
              acc[g] = mfma_32x32x16_a16(ah, bl, acc[g]);
            }
          }
          if (NW == 8) __builtin_amdgcn_sched_barrier(0);      // the second burst's loads are not hoisted above the first burst's MFMAs
        }
      }
      // cell update (lane-local: the four gates of a (line, unit) share lane and register), h as a (hi, lo) pair to the staging tiles
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * q;
        const float gi = acc[0][r] + rbf2f(gh[r].x & 0xFFFFu) + rbf2f(gl[r].x & 0xFFFFu);
        const float gf = acc[1][r] + rbf2f(gh[r].x >> 16) + rbf2f(gl[r].x >> 16);
        const float gg = acc[2][r] + rbf2f(gh[r].y & 0xFFFFu) + rbf2f(gl[r].y & 0xFFFFu);
        const float go = acc[3][r] + rbf2f(gh[r].y >> 16) + rbf2f(gl[r].y >> 16);
        const float si = fast_sigmoid(gi), sf = fast_sigmoid(gf), so = fast_sigmoid(go);
        const float cn = sf * c[mi][r] + si * fast_tanh(gg);
        c[mi][r] = cn;
        const float hn = so * fast_tanh(cn);
        const uint32_t hb = rf2bf(hn);
        const int ml = (r & 3) + 8 * (r >> 2) + 4 * q;
        stage[ml * SROW + lx] = (bf16_t)hb;
        stage[(32 + ml) * SROW + lx] = (bf16_t)rf2bf(hn - rbf2f(hb));
        (void)m;
      }
      // publish the tile: 32 lines x 4 pieces of 16 B x (hi, lo); piece pc of a line = units 32 member + 8 pc .. + 8 = k-step 2 member + (pc >> 1),
      // half q = pc & 1 of the consumers' A fragments
      {
        unsigned long long* hw = hxc + (size_t)(s & 1) * XBUF;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int idx = lane + i * 64, ml = idx >> 2, pc = idx & 3;
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            const u32x4 pv = *reinterpret_cast<const u32x4*>(stage + (pl * 32 + ml) * SROW + pc * 8);
            unsigned long long* dp = hw + ((size_t)((tile * 16 + 2 * member + (pc >> 1)) * 2 + pl) * 64 + (pc & 1) * 32 + ml) * 2;
            __hip_atomic_store(dp, (unsigned long long)pv.x | ((unsigned long long)pv.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dp + 1, (unsigned long long)pv.z | ((unsigned long long)pv.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // only the exchange stores are outstanding here: drain them, meet, bump the counter; the layer output (HBM) follows behind the
    // counter, off the step's critical chain -- copied from this member's own fragments in the exchange buffer (L2), which frees the
    // registers a held copy would take
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(fl + member, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
      const unsigned long long* hr = hxc + (size_t)(s & 1) * XBUF;
      int tio = tid;
      asm volatile("" : "+v"(tio));
#pragma unroll
      for (int i = 0; i < 2 * MI; ++i) {
        const int idx = tio + i * NTHR, m = idx >> 2, pc = idx & 3;      // m: line inside the cluster
        const int line = line0 + m;
        if (line < B) {
          bf16_t* ho = hout + ((size_t)line * T + t) * 1024 + dir * 256 + member * 32 + pc * 8;
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            const gptr_u64 sp = (gptr_u64)hr + ((size_t)(((m >> 5) * 16 + 2 * member + (pc >> 1)) * 2 + pl) * 64 + (pc & 1) * 32 + (m & 31)) * 2;
            const unsigned long long v0 = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long v1 = __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *reinterpret_cast<u32x4*>(ho + pl * 512) = u32x4{(uint32_t)v0, (uint32_t)(v0 >> 32), (uint32_t)v1, (uint32_t)(v1 >> 32)};
          }
        }
      }
    }
  }
}

template <int MI, int NW>
static int launch_lstm_cluster8_x3(pt_engine* e, const bf16_t* gx, const bf16_t* whh, bf16_t* hout, int B, int T, hipStream_t s) {
  constexpr int CLL = 32 * NW * MI;
  constexpr int SMEM = 131072 + NW * 2 * 32 * 32 * 2;      // W slices + a wave-private staging tile per wave (160 KB exactly at eight waves)
  static_assert(SMEM <= 163840, "LDS");
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_cluster8_x3_kernel<MI, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_done = true;
  }
  const int max_cl = e->num_cu / 16 < 1 ? 1 : e->num_cu / 16;      // 2 dirs x 8 members x max_cl <= num_cu
  const size_t xbuf = (size_t)NW * MI * 16 * 2 * 64 * 2 * sizeof(unsigned long long);
  const size_t need = (size_t)2 * max_cl * 2 * xbuf + (size_t)2 * max_cl * 8 * sizeof(int) + 256;
  if (need > e->lstm_scratch8_cap) {
    PT_HIP_CHECK(hipStreamSynchronize(s));
    if (e->lstm_scratch8) PT_HIP_CHECK(hipFree(e->lstm_scratch8));
    e->lstm_scratch8 = nullptr; e->lstm_scratch8_cap = 0;
    PT_HIP_CHECK(hipMalloc(&e->lstm_scratch8, need));
    e->lstm_scratch8_cap = need;
  }
  unsigned long long* hx = reinterpret_cast<unsigned long long*>(e->lstm_scratch8);
  int* flags = reinterpret_cast<int*>(reinterpret_cast<char*>(e->lstm_scratch8) + (size_t)2 * max_cl * 2 * xbuf);
  int* err = nullptr;
  PT_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&err), e->lstm_err, 0));
  const int nlaunch = (B + max_cl * CLL - 1) / (max_cl * CLL);
  const int per = (((B + nlaunch - 1) / nlaunch) + CLL - 1) / CLL * CLL;
  for (int b0 = 0; b0 < B; b0 += per) {
    const int nb = (B - b0) < per ? (B - b0) : per;
    const int ncl = (nb + CLL - 1) / CLL;
    PT_HIP_CHECK(hipMemsetAsync(flags, 0, (size_t)(2 * max_cl * 8) * sizeof(int), s));
    hipLaunchKernelGGL((lstm_cluster8_x3_kernel<MI, NW>), dim3(((ncl + 7) / 8) * 64, 2), dim3(64 * NW), SMEM, s, gx + (size_t)b0 * T * 4096, whh,
                       hout + (size_t)b0 * T * 1024, nb, T, ncl, hx, flags, err);
  }
  return PT_OK;
}

int pt_launch_lstm(pt_engine* e, const bf16_t* gx, const bf16_t* whh, bf16_t* hout, int B, int T, int split, hipStream_t s) {
  if (B <= 0) return PT_OK;
  dim3 grid((B + 31) / 32, 2);
  static int ng2 = -1;           // PT_LSTM_NG=1: one 32-line group per workgroup in the hi/lo kernel too (A/B switch)
  if (ng2 < 0) {
    const char* ev = getenv("PT_LSTM_NG");
    ng2 = ev ? (atoi(ev) == 2) : 1;
  }
  static int use_dma = -1;       // PT_LSTM_DMA=0: the register-staged kernel also in bf16 mode (A/B switch)
  if (use_dma < 0) {
    const char* ev = getenv("PT_LSTM_DMA");
    use_dma = ev ? atoi(ev) : 0;      // measured 3.97 vs 3.89 ms: no gain (the step is bound by L2 traffic, not by gx latency)
  }
  if (!split && e->lstm_cluster) {     // pt_engine_set_lstm_cluster / PT_LSTM_CLUSTER=0: the streaming kernel
    if (!e->lstm_scratch) {     // per engine: exchange buffers + step counters, pinned error word, clusters per launch
      e->lstm_max_cl = e->num_cu / 8 < 1 ? 1 : e->num_cu / 8;     // 2 dirs x 4 members x max_cl <= num_cu
      PT_HIP_CHECK(hipMalloc(&e->lstm_scratch, (size_t)2 * e->lstm_max_cl * 2 * CLL_MAX * 256 * sizeof(bf16_t) + (size_t)2 * e->lstm_max_cl * 4 * sizeof(int) + 256));
      PT_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&e->lstm_err), sizeof(int), hipHostMallocMapped));
      *e->lstm_err = 0;
    }
    void* scratch = e->lstm_scratch;
    int* h_err = e->lstm_err;
    const int max_cl = e->lstm_max_cl;
    if (*h_err) {      // an EARLIER launch timed out (its output was wrong): fail loudly now, once, and stop using the kernel
      *h_err = 0;
      e->lstm_cluster = 0;
      pt_set_error("lstm_cluster_kernel: a workgroup waited > 2^22 polls for its peers -- the launch was not co-resident "
                   "(GPU shared with another process or stream?).  Results of that call are invalid; this engine now uses "
                   "the streaming LSTM kernel: run the batch again");
      return PT_ERR_HIP;
    }
    bf16_t* hx = reinterpret_cast<bf16_t*>(scratch);
    int* flags = reinterpret_cast<int*>(reinterpret_cast<char*>(scratch) + (size_t)2 * max_cl * 2 * CLL_MAX * 256 * sizeof(bf16_t));
    int* err = nullptr;
    PT_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&err), h_err, 0));
    // 192-line clusters when they save a launch (e.g. 4872 lines: 1 launch instead of 2); PT_LSTM_MI = 2 / 3 forces one
    const char* ev = getenv("PT_LSTM_MI");
    const int force = ev ? atoi(ev) : 0;
    const int n2 = (B + max_cl * 128 - 1) / (max_cl * 128), n3 = (B + max_cl * 192 - 1) / (max_cl * 192);
    const bool mi3 = force == 3 || (force != 2 && n3 < n2);
    const int rc = mi3 ? launch_lstm_cluster<3>(gx, whh, hout, B, T, max_cl, hx, flags, err, s)
                       : launch_lstm_cluster<2>(gx, whh, hout, B, T, max_cl, hx, flags, err, s);
    if (rc != PT_OK) return rc;
  } else if (split == 1 && e->lstm_cluster && !(getenv("PT_LSTM_CLUSTER_X3") && atoi(getenv("PT_LSTM_CLUSTER_X3")) == 0)) {
    // hi/lo mode: the eight-member weight-stationary kernel (PT_LSTM_CLUSTER_X3=0: the streaming kernel; A/B switch, read per call)
    if (!e->lstm_err) {
      PT_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&e->lstm_err), sizeof(int), hipHostMallocMapped));
      *e->lstm_err = 0;
    }
    if (*e->lstm_err) {      // an EARLIER launch timed out (its output was wrong): fail loudly now, once, and stop using the kernels
      *e->lstm_err = 0;
      e->lstm_cluster = 0;
      pt_set_error("lstm_cluster8_x3_kernel: a workgroup waited > 2^22 polls for its peers -- the launch was not co-resident "
                   "(GPU shared with another process or stream?).  Results of that call are invalid; this engine now uses "
                   "the streaming LSTM kernel: run the batch again");
      return PT_ERR_HIP;
    }
    // PT_LSTM_X3_WAVES=4: one wave per SIMD with 256- / 384-line clusters (PT_LSTM_MI = 2 / 3; 384 when it saves a launch); default: eight
    // waves (two per SIMD, each hiding the other's fetches) with 256- / 512-line clusters (512 when it saves a launch or PT_LSTM_MI=2)
    const int max_cl = e->num_cu / 16 < 1 ? 1 : e->num_cu / 16;
    const char* ev = getenv("PT_LSTM_MI");
    const int force = ev ? atoi(ev) : 0;
    const char* wv = getenv("PT_LSTM_X3_WAVES");
    int rc;
    if (wv && atoi(wv) == 4) {
      const int n2 = (B + max_cl * 256 - 1) / (max_cl * 256), n3 = (B + max_cl * 384 - 1) / (max_cl * 384);
      const bool mi3 = force == 3 || (force != 2 && n3 < n2);
      rc = mi3 ? launch_lstm_cluster8_x3<3, 4>(e, gx, whh, hout, B, T, s) : launch_lstm_cluster8_x3<2, 4>(e, gx, whh, hout, B, T, s);
    } else {
      const int n1 = (B + max_cl * 256 - 1) / (max_cl * 256), n2 = (B + max_cl * 512 - 1) / (max_cl * 512);
      const bool mi2 = force == 2 || (force != 1 && n2 < n1);
      rc = mi2 ? launch_lstm_cluster8_x3<2, 8>(e, gx, whh, hout, B, T, s) : launch_lstm_cluster8_x3<1, 8>(e, gx, whh, hout, B, T, s);
    }
    if (rc != PT_OK) return rc;
  } else if (split) {
    if (ng2)
      hipLaunchKernelGGL((lstm_dir_kernel<1, 2>), dim3((B + 63) / 64, 2), dim3(512), 0, s, gx, whh, hout, B, T);
    else
      hipLaunchKernelGGL((lstm_dir_kernel<1, 1>), grid, dim3(256), 0, s, gx, whh, hout, B, T);
  } else if (use_dma) {
    constexpr int SMEM = 32 * 264 * 2 + 2 * 65536;
    static bool attr_done = false;
    if (!attr_done) {
      PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_dir_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
      attr_done = true;
    }
    hipLaunchKernelGGL(lstm_dir_dma_kernel, grid, dim3(256), SMEM, s, gx, whh, hout, B, T);
  } else {
    if (ng2)
      hipLaunchKernelGGL((lstm_dir_kernel<0, 2>), dim3((B + 63) / 64, 2), dim3(512), 0, s, gx, whh, hout, B, T);
    else
      hipLaunchKernelGGL((lstm_dir_kernel<0, 1>), grid, dim3(256), 0, s, gx, whh, hout, B, T);
  }
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// arg-max over the per-tile partials written by the classifier GEMM epilogue: [rows][ntiles] float2(max, idx bits)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_reduce_kernel(const float2* __restrict__ part, long long rows, int ntiles,
                                                             int* __restrict__ ids, float* __restrict__ maxv) {
  a16_kernel_enter();
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
    const float2* p = part + r * ntiles;
    float bv = p[0].x;
    int bi = __float_as_int(p[0].y);
    for (int k = 1; k < ntiles; ++k) {
      const float v = p[k].x;
      const int i = __float_as_int(p[k].y);
      if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    ids[r] = bi;
    if (maxv) maxv[r] = bv;
  }
}

// ---------------------------------------------------------------------------------------------------
// Classifier GEMM with the arg-max inside (bf16 mode): ids[row] = argmax_n (A[row, :] . W[n, :] + bias[n]).
// The implicit-GEMM 1x1 kernel runs this layer (M = lines x 160, K = 512, N = 7680) at ~500 TFLOP/s: with four MFMAs per
// K-slice and barrier it is bound by its own hand-over, and a third of every (128 x 64) tile goes to an fp32-through-LDS
// epilogue that only feeds a compare.  Here the roles are turned around:
//   * a wave keeps its 32 rows of A (all of K) in registers for the whole kernel: 32 k-steps x 16 B per lane;
//   * W streams through LDS in 64-class stages (64 KB, next stage pre-fetched to registers, rows pitched K*2 + 16 bytes);
//   * the MFMA runs with W as the A operand, so D is [class][row]: a lane owns ONE row and its 16 accumulators are 16
//     classes in increasing order -- the running (max, index) is a compare/select per accumulator in the lane, no LDS,
//     no partials; the two lanes that share a row meet once at the end (lane ^ 32).
// Ties keep the lowest class index (torch.argmax); sums are the same MFMA sequence over K as the conv kernel's.
// ---------------------------------------------------------------------------------------------------
// MODE 1: the same streaming GEMM with a store epilogue instead (out bf16 [M][N] = A . W^T + bias, optional ReLU): a lane
// writes its row's classes as 8-byte runs of four (the LSTM input projections and embeddings of the CRNN head).
// GELU for the bf16 mode's row GEMM (ConvNextViT stage 3, cvit_model.hip): x * Phi(x), Phi - 1/2 = x * Q(x^2) with the degree-9
// Q of a Chebyshev fit on [-4, 4] (the polynomial of cvit_model.hip's gelu_pair; |error| < 3e-6, clamped outside) -- the value
// is rounded to bf16 right after
__device__ __forceinline__ float gelu_poly(float x) {
  const float xp = fmaxf(x, -4.f), xc = fminf(xp, 4.f), t = xc * xc;
  float q = -3.658831230e-12f;
  q = fmaf(q, t, 3.561182861e-10f); q = fmaf(q, t, -1.572596130e-08f); q = fmaf(q, t, 4.224180292e-07f);
  q = fmaf(q, t, -7.841504780e-06f); q = fmaf(q, t, 1.084709610e-04f); q = fmaf(q, t, -1.168552637e-03f);
  q = fmaf(q, t, 9.945140159e-03f); q = fmaf(q, t, -6.647037283e-02f); q = fmaf(q, t, 3.989380888e-01f);
  return xp * fmaf(xc, q, 0.5f);
}

#ifndef PT_ROWS_ABL
#define PT_ROWS_ABL 0
#endif
// NW: waves per workgroup (4 or 8).  Every workgroup streams ALL of W through its LDS stage: with eight waves on one stage instead of two
// workgroups of four, a CU moves half the weight bytes (global -> registers -> ds_write at ~79 B/clk, which was 40 % of the MFMA time).
template <int KSTEPS, int MODE, int NW = 4>
__global__ __launch_bounds__(NW * 64, 2) void gemm_argmax_kernel(const bf16_t* __restrict__ A, long long M,
                                                             const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                             int N, int* __restrict__ ids, float* __restrict__ maxv,
                                                             bf16_t* __restrict__ out, int relu, const int* __restrict__ tlim,
                                                             int lda, long long wts) {
  a16_kernel_enter();
  // lda: elements between rows of A (K, or 2 K for the hi halves of (hi | lo) rows); wts: elements between 64-class tiles of W
  // (NCH * 2048, or three times that for the first third -- the w_hi chunks -- of the three-pass tiling)
  constexpr int K = KSTEPS * 16, NCH = K / 32, P = K * 2 + 16;     // P: LDS row pitch in bytes (odd number of 16-B slots)
  constexpr int NTHR = NW * 64, NPF = 64 * K * 2 / 16 / NTHR;          // 16-byte pieces per thread per stage
  static_assert(64 * K * 2 / 16 % NTHR == 0, "a stage is whole passes of the workgroup");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sw = smem;                                                  // [64 classes][P]
  float* sb = reinterpret_cast<float*>(smem + 64 * P);              // [64] bias of the stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  // ragged sequences (MODE 1, tlim = the list of rows_live_list_kernel): a wave's 32 rows are 32 consecutive time steps of ONE line
  // (T = 160 = 5 x 32); only the groups in front of their line's limit are computed (the caller fills the others), and they are
  // COMPACTED over the waves -- tlim[0] live groups, tlim[1 + g] = row group of the g-th -- so that no workgroup streams W for
  // one live wave out of four (lines are ~ 1/4 text on the bench pages: skipping whole workgroups only left 70 % of them running)
  long long row0 = ((long long)blockIdx.x * NW + wave) * 32;
  bool live = true;
  if (MODE == 1 && tlim) {
    const int total = tlim[0], g = blockIdx.x * NW + wave;
    if ((int)blockIdx.x * NW >= total) return;
    live = g < total;
    row0 = live ? (long long)tlim[1 + g] * 32 : 0;
  }
  const long long row = row0 + lx;
  const long long rc = row < M ? row : M - 1;
  bf16x8 areg[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) areg[ks] = *reinterpret_cast<const bf16x8*>(A + rc * lda + ks * 16 + q * 8);
  u32x4 pf[NPF];
  float pb = 0.f;
  auto prefetch = [&](int t) {
    const bf16_t* wt = W + (size_t)t * wts;
#pragma unroll
    for (int j = 0; j < NPF; ++j) pf[j] = *reinterpret_cast<const u32x4*>(wt + (size_t)(tid + j * NTHR) * 8);
    if (tid < 64) pb = bias[t * 64 + tid];
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int idx = tid + j * NTHR, c = idx >> 8, r = (idx & 255) >> 2, part = idx & 3;    // chunk, class row, 16-B part
      *reinterpret_cast<u32x4*>(sw + r * P + c * 64 + part * 16) = pf[j];
    }
    if (tid < 64) sb[tid] = pb;
  };
  float bv = -INFINITY;
  int bi = 0;
  const int NT = N / 64;
  // (walking the tiles from a per-workgroup start, to spread the stores' addresses over the memory channels, measured SLOWER: 1.02 -> 1.48 ms for
  // K = 512 -- the workgroups then stream different weight tiles at the same time)
  prefetch(0);
  for (int t = 0; t < NT; ++t) {
    __syncthreads();
    commit();
    __syncthreads();
    if (t + 1 < NT) prefetch(t + 1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const char* wr = sw + (half * 32 + lx) * P + q * 16;
      if (MODE == 1 && !live) continue;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
#if PT_ROWS_ABL == 2     /* ablation: one MFMA per half stage */
        if (MODE == 1 && ks) break;
#endif
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wr + ks * 32);
        acc = mfma_32x32x16_a16(wf, areg[ks], acc);
      }
      if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cl = half * 32 + (r & 3) + 8 * (r >> 2) + 4 * q;
          const float v = acc[r] + sb[cl];
          if (v > bv) { bv = v; bi = t * 64 + cl; }
        }
      } else {
        // store epilogue through a wave-private LDS tile: a lane owns a ROW, so storing from the accumulators touches 64 lines with 8 bytes
        // each per instruction (measured: the 256 -> 2048 projection took 1.76 ms with its stores and 0.23 ms without).  The half stage's
        // 32 rows x 32 classes are transposed to 64-byte row segments instead: four lanes per row, 16 rows per 16-byte-per-lane store.
        char* tile = smem + 64 * P + 256 + wave * (32 * 80);          // row pitch 80 bytes
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int cl = half * 32 + rg * 8 + 4 * q;
          uint32_t hb[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float v = acc[rg * 4 + k] + sb[cl + k];
            if (relu == 1) v = fmaxf(v, 0.f);
            else if (relu == 4) v = gelu_poly(v);
            hb[k] = rf2bf(v);
          }
          *reinterpret_cast<u32x2*>(tile + lx * 80 + (rg * 8 + 4 * q) * 2) = u32x2{hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the tile is wave-private: LDS operations of a wave complete in order
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = (lane >> 2) + 16 * i;
          const u32x4 v = *reinterpret_cast<const u32x4*>(tile + r * 80 + (lane & 3) * 16);
#if PT_ROWS_ABL == 1     /* ablation: no stores */
          if (v.x == 0x12345u)
#endif
#if PT_ROWS_ABL == 3     /* ablation (timing only, wrong layout): the wave's 32 x 32 half tile stored as 2 KB of consecutive bytes */
          if (row0 + r < M) *reinterpret_cast<u32x4*>(out + ((row0 >> 5) * (N / 64) + t) * 2048 + half * 1024 + r * 32 + (lane & 3) * 8) = v;
#else
          if (row0 + r < M) *reinterpret_cast<u32x4*>(out + (row0 + r) * N + t * 64 + half * 32 + (lane & 3) * 8) = v;
#endif
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // reads done before the next half overwrites the tile
      }
    }
  }
  if (MODE != 0) return;
  const float ov = __shfl_xor(bv, 32);
  const int oi = __shfl_xor(bi, 32);
  if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  if (q == 0 && row < M) {
    ids[row] = bi;
    if (maxv) maxv[row] = bv;
  }
}

// ---------------------------------------------------------------------------------------------------
// The K = 512 classifier arg-max (gemm_argmax_kernel<32, 0, 8>: a wave's 32 rows resident in registers, the class tiles streamed through LDS) with the
// weight stream on LDS-DMA and a pinned inner loop.  In that kernel's ISA every pair of MFMAs waits for an LDS round trip requested just in front of
// it (read, read, wait, MFMA, wait, MFMA: hipcc sinks the reads to their uses), all 64 MFMAs of a class tile hang on ONE accumulator chain, the next
// tile travels global -> 32 registers -> ds_write, and a tile costs two barriers: matrix pipe 51 % busy, LDS port 42 % (profiles/r06).  Here
//   * the next tile is requested by MUBUF LDS-DMA (one 1 KB class row per instruction, eight per wave) into the other of two LDS buffers: no staging
//     registers, no ds_writes, ONE barrier per tile;
//   * both 32-class halves of a tile are multiplied together (two accumulator chains: an MFMA never waits for the one in front of it), the
//     fragments of step ks + 2 are requested between the MFMAs of step ks (pinned with scheduling groups).
// Same operands, same K order per (row, class), same visiting order of the classes in the arg-max: ids and maxima equal gemm_argmax_kernel's bit for bit.
// ---------------------------------------------------------------------------------------------------
constexpr int CAND_SLOTS = 8;      // candidate slots per (row, half of the classes a lane pair splits): 16 per row (the hi/lo mode's bound-and-refine arg-max below)

template <int MODE>      // 0: arg-max (ids, maxv); 1: the hi/lo mode's second sweep -- every class within the bound of the row's maximum into cand (gemm_cand_kernel's lists)
__global__ __launch_bounds__(512, 1) void cls_argmax_dma_kernel(const bf16_t* __restrict__ A, long long M, const bf16_t* __restrict__ W,
                                                                const float* __restrict__ bias, int N, int* __restrict__ ids, float* __restrict__ maxv,
                                                                int lda, long long wts, const float* __restrict__ wmax, const float* __restrict__ rowmax,
                                                                int* __restrict__ cand) {
#if defined(__HIP_DEVICE_COMPILE__)      // (buffer-resource builtins: device pass only)
  a16_kernel_enter();
  constexpr int KSTEPS = 32, K = 512, P = K * 2 + 16, TILE = 64 * P, D = 2;      // P: LDS row pitch (odd number of 16-byte slots); D: fragment prefetch distance
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sb = reinterpret_cast<float*>(smem + 2 * TILE);            // [2][64] bias of the tile in each buffer
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = lane & 31, q = lane >> 5;
  const long long row0 = ((long long)blockIdx.x * 8 + wave) * 32;
  const long long row = row0 + lx;
  const long long rc = row < M ? row : M - 1;
  bf16x8 areg[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) areg[ks] = *reinterpret_cast<const bf16x8*>(A + rc * lda + ks * 16 + q * 8);
  float thr = 0.f;
  int* cp = nullptr;
  int cnt = 0;
  if (MODE == 1) {      // (gemm_cand_kernel's bound: see the derivation above it)
    float ss = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = (float)areg[ks][i];
        ss += v * v;
      }
    ss += __shfl_xor(ss, 32);
    thr = rowmax[rc] - 2.f * (0.00390625f * 1.02f + 6.103515625e-05f) * sqrtf(ss) * 1.0001f * wmax[0];
    cp = cand + (rc * 2 + q) * (1 + CAND_SLOTS);
  }
  // DMA slot j of this wave = class row r = wave + 8 j of the tile; lane L fetches 16-byte part L & 3 of 32-channel chunk L >> 2 (tiling [K/32][64][32])
  constexpr int OOB = 0x7FFFF000;
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W), 0, OOB, 0x00020000);
  int voff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) voff[j] = ((((lane >> 2) * 64 + wave + 8 * j) * 32) + (lane & 3) * 8) * 2;
  const int NT = N / 64;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)(smem + (wave + 8 * j) * P), 16, voff[j], 0, 0, 0);
  if (tid < 64) sb[tid] = bias[tid];
  float bv = -INFINITY;
  int bi = 0;
  for (int t = 0; t < NT; ++t) {
    const int b = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // tile t is in buffer b (every wave's rows), tile t - 1 has been read
    const bool more = t + 1 < NT;
    const int soff = more ? (int)((long long)(t + 1) * wts * 2) : 0;
    if (tid < 64) sb[(b ^ 1) * 64 + tid] = bias[(more ? t + 1 : t) * 64 + tid];
    const char* wr = smem + b * TILE + lx * P + q * 16;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    bf16x8 f0[D + 1], f1[D + 1];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      f0[d] = *reinterpret_cast<const bf16x8*>(wr + d * 32);
      f1[d] = *reinterpret_cast<const bf16x8*>(wr + 32 * P + d * 32);
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (ks + D < KSTEPS) {
        f0[(ks + D) % (D + 1)] = *reinterpret_cast<const bf16x8*>(wr + (ks + D) * 32);
        f1[(ks + D) % (D + 1)] = *reinterpret_cast<const bf16x8*>(wr + 32 * P + (ks + D) * 32);
      }
      acc0 = mfma_32x32x16_a16(f0[ks % (D + 1)], areg[ks], acc0);
      acc1 = mfma_32x32x16_a16(f1[ks % (D + 1)], areg[ks], acc1);
      if ((ks & 3) == 0)      // the next tile's eight rows of this wave, one request every four steps (the last tile: out of range, zero-filled, never read)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)(smem + (b ^ 1) * TILE + (wave + 8 * (ks >> 2)) * P), 16,
                                                 more ? voff[ks >> 2] : OOB, soff, 0, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * D, 0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (ks + D < KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (ks + D < KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if ((ks & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    const float* sbt = sb + b * 64;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = half * 32 + (r & 3) + 8 * (r >> 2) + 4 * q;
        const float v = (half ? acc1[r] : acc0[r]) + sbt[cl];
        if (MODE == 0) {
          if (v > bv) { bv = v; bi = t * 64 + cl; }
        } else if (v >= thr) {      // rare (1.4 per row): straight to memory
          if (cnt < CAND_SLOTS && row < M) cp[1 + cnt] = t * 64 + cl;
          ++cnt;
        }
      }
  }
  if (MODE == 1) {
    if (row < M) cp[0] = cnt;
    return;
  }
  const float ov = __shfl_xor(bv, 32);
  const int oi = __shfl_xor(bi, 32);
  if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  if (q == 0 && row < M) {
    ids[row] = bi;
    if (maxv) maxv[row] = bv;
  }
#endif
}

static bool cls_dma() {      // PT_CLS_DMA=0: gemm_argmax_kernel<32, 0, 8> for the K = 512 classifier (A/B switch, read per call)
  const char* ev = getenv("PT_CLS_DMA");
  return !(ev && ev[0] == '0');
}
constexpr int CLS_DMA_SMEM = 2 * 64 * (512 * 2 + 16) + 2 * 64 * 4;
static int cls_dma_attr() {
  static bool done = false;
  if (!done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cls_argmax_dma_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, CLS_DMA_SMEM));
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cls_argmax_dma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, CLS_DMA_SMEM));
    done = true;
  }
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// Classifier arg-max of the hi/lo (BF16X3) mode by bound and refine.  The tiled three-pass GEMM with per-tile partials + reduce
// spends 33 ms per 5 k lines (560 TF/s of three-pass work) to find ONE class per row.  With a = a_hi + a_lo, w = w_hi + w_lo:
//     |a . w - a_hi . w_hi|  <=  |a_lo . w_hi| + |a . w_lo|  <=  2^-9 |a| |w_hi| + 2^-9 |a| |w| (1 + 2^-9)  <  2^-8 * 1.02 |a_hi| |w|
// (bf16 rounding to nearest: |x_lo| <= 2^-9 |x| element-wise; Cauchy-Schwarz), plus the fp32 accumulation of the single-pass
// product (<= 512 * 2^-24 |a| |w| = 2^-15 |a| |w|).  So with m = (2^-8 * 1.02 + 2^-14) |a_hi| max_n |w_n| the class that maximises the
// exact logit has a single-pass logit within 2 m of the single-pass maximum: the kernel below sweeps the classes TWICE at the bf16
// rate with the row's a_hi resident in registers -- first the maximum, then every class within 2 m of it into per-row slots (a lane
// owns its row: no atomics) -- and cand_eval_kernel computes the exact four-term logits of those few classes (1.4 per row on
// the bench's lines) in fp32 and takes their arg-max (lowest index on ties).  A row with more candidates than slots is evaluated
// over all classes (rare: the margin is 0.18 standard deviations of a row's logits).  The result is the arg-max of the exact
// (hi + lo) x (hi + lo) products: at least as close to the fp32 oracle as the three-pass sum it replaces.
// ---------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void wnorm_max_kernel(const bf16_t* __restrict__ W3, int N, int K, unsigned* __restrict__ out) {
  a16_kernel_enter();
  // max over classes of |w_hi + w_lo|: a workgroup per 64-class tile of the [N/64][3 K/32][64][32] tensor (chunks: hi, hi, lo), a wave per
  // class row, one atomic per workgroup
  __shared__ float smax[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nch = K >> 5;
  float best = 0.f;
  for (int r = wave; r < 64; r += 4) {
    const int n = blockIdx.x * 64 + r;
    if (n >= N) break;
    const bf16_t* wt = W3 + (size_t)blockIdx.x * 3 * nch * 2048 + r * 32;
    float ss = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float v = rbf2f(wt[(size_t)(k >> 5) * 2048 + (k & 31)]) + rbf2f(wt[(size_t)(2 * nch + (k >> 5)) * 2048 + (k & 31)]);
      ss += v * v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
    best = fmaxf(best, sqrtf(ss) * 1.0001f);
  }
  if (lane == 0) smax[wave] = best;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(out, __float_as_uint(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]))));      // non-negative floats order like their bit patterns
}

template <int KSTEPS>
__global__ __launch_bounds__(256, 2) void gemm_cand_kernel(const bf16_t* __restrict__ A, long long M, int lda,
                                                           const bf16_t* __restrict__ W, long long wts, const float* __restrict__ bias,
                                                           int N, const float* __restrict__ wmax, const float* __restrict__ rowmax,
                                                           int* __restrict__ cand) {
  a16_kernel_enter();
  constexpr int K = KSTEPS * 16, P = K * 2 + 16;
  constexpr int NPF = 64 * K * 2 / 16 / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sw = smem;
  float* sb = reinterpret_cast<float*>(smem + 64 * P);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  const long long row = ((long long)blockIdx.x * 4 + wave) * 32 + lx;
  const long long rc = row < M ? row : M - 1;
  bf16x8 areg[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) areg[ks] = *reinterpret_cast<const bf16x8*>(A + rc * lda + ks * 16 + q * 8);
  float thr;
  {
    float ss = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = (float)areg[ks][i];
        ss += v * v;
      }
    ss += __shfl_xor(ss, 32);
    thr = rowmax[rc] - 2.f * (0.00390625f * 1.02f + 6.103515625e-05f) * sqrtf(ss) * 1.0001f * wmax[0];
  }
  // [row][q][1 + CAND_SLOTS]: count (may exceed the slots: the row is then evaluated over all classes), class ids
  int* cp = cand + (rc * 2 + q) * (1 + CAND_SLOTS);
  int cnt = 0;
  u32x4 pf[NPF];
  float pb = 0.f;
  auto prefetch = [&](int t) {
    const bf16_t* wt = W + (size_t)t * wts;
#pragma unroll
    for (int j = 0; j < NPF; ++j) pf[j] = *reinterpret_cast<const u32x4*>(wt + (size_t)(tid + j * 256) * 8);
    if (tid < 64) pb = bias[t * 64 + tid];
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int idx = tid + j * 256, c = idx >> 8, r = (idx & 255) >> 2, part = idx & 3;
      *reinterpret_cast<u32x4*>(sw + r * P + c * 64 + part * 16) = pf[j];
    }
    if (tid < 64) sb[tid] = pb;
  };
  const int NT = N / 64;
  prefetch(0);
  for (int t = 0; t < NT; ++t) {
    __syncthreads();
    commit();
    __syncthreads();
    if (t + 1 < NT) prefetch(t + 1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const char* wr = sw + (half * 32 + lx) * P + q * 16;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wr + ks * 32);
        acc = mfma_32x32x16_a16(wf, areg[ks], acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = half * 32 + (r & 3) + 8 * (r >> 2) + 4 * q;
        if (acc[r] + sb[cl] >= thr) {      // rare (1.4 per row): straight to memory
          if (cnt < CAND_SLOTS && row < M) cp[1 + cnt] = t * 64 + cl;
          ++cnt;
        }
      }
    }
  }
  if (row < M) cp[0] = cnt;
}

// exact logits (a_hi + a_lo) . (w_hi + w_lo) + bias of every row's candidate classes and their arg-max.  Sixteen lanes per row (four rows
// per wave): lane l holds the 32 activations of K-chunk l and reads the 64 + 64 bytes of that chunk of a class's w_hi / w_lo rows; K = 512
__global__ __launch_bounds__(256) void cand_eval_kernel(const bf16_t* __restrict__ A, long long M, int lda, int K, const bf16_t* __restrict__ W3,
                                                        const float* __restrict__ bias, int N, int n_real, const int* __restrict__ cand,
                                                        int* __restrict__ ids, float* __restrict__ maxv, int* __restrict__ ovf_count,
                                                        int* __restrict__ ovf_rows) {
  a16_kernel_enter();
  const int lane = threadIdx.x & 63, l = lane & 15;
  const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const long long rc = row < M ? row : M - 1;
  const int nch = K >> 5;
  float a[32];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const u32x4 ah = *reinterpret_cast<const u32x4*>(A + rc * lda + l * 32 + i * 8);
    const u32x4 al = *reinterpret_cast<const u32x4*>(A + rc * lda + K + l * 32 + i * 8);
    const uint32_t hw[4] = {ah.x, ah.y, ah.z, ah.w}, lw[4] = {al.x, al.y, al.z, al.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[8 * i + 2 * j] = rbf2f(hw[j] & 0xFFFFu) + rbf2f(lw[j] & 0xFFFFu);
      a[8 * i + 2 * j + 1] = rbf2f(hw[j] >> 16) + rbf2f(lw[j] >> 16);
    }
  }
  auto logit = [&](int n) {
    const bf16_t* wt = W3 + (size_t)(n >> 6) * 3 * nch * 2048 + ((size_t)l * 64 + (n & 63)) * 32;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 wh = *reinterpret_cast<const u32x4*>(wt + i * 8);
      const u32x4 wl = *reinterpret_cast<const u32x4*>(wt + (size_t)2 * nch * 2048 + i * 8);
      const uint32_t hw[4] = {wh.x, wh.y, wh.z, wh.w}, lw[4] = {wl.x, wl.y, wl.z, wl.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sum = fmaf(a[8 * i + 2 * j], rbf2f(hw[j] & 0xFFFFu) + rbf2f(lw[j] & 0xFFFFu), sum);
        sum = fmaf(a[8 * i + 2 * j + 1], rbf2f(hw[j] >> 16) + rbf2f(lw[j] >> 16), sum);
      }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);      // inside the row's 16 lanes
    return sum + bias[n];
  };
  const int* cp = cand + rc * 2 * (1 + CAND_SLOTS);
  const int c0 = cp[0], c1 = cp[1 + CAND_SLOTS];
  // more candidates than slots (a row whose logits lie closer together than the rounding bound): every class has to be evaluated -- 7 680
  // dependent gathers would make one such row the kernel's tail (measured: 5-10 ms), so it goes to cand_full_kernel's list instead
  const bool all = c0 > CAND_SLOTS || c1 > CAND_SLOTS;
  if (all) {
    if (l == 0 && row < M) ovf_rows[atomicAdd(ovf_count, 1)] = (int)row;
  }
  const int total = all ? 0 : c0 + c1;
  float bv = -INFINITY;
  int bi = 0x7FFFFFFF;
  for (int i = 0; __any(i < total); ++i) {
    if (i < total) {      // uniform over the row's 16 lanes
      const int n = i < c0 ? cp[1 + i] : cp[(1 + CAND_SLOTS) + 1 + (i - c0)];
      const float v = logit(n);
      if (v > bv || (v == bv && n < bi)) { bv = v; bi = n; }
    }
  }
  if (l == 0 && row < M && !all) {
    ids[row] = bi;
    if (maxv) maxv[row] = bv;
  }
}

// the rows cand_eval_kernel could not settle from their slots: a workgroup per row, its 16 lane groups take the classes g, g + 16, ... in
// ascending order (the first maximum of a group wins), then the groups meet in LDS (lowest class index on equal values)
__global__ __launch_bounds__(256) void cand_full_kernel(const bf16_t* __restrict__ A, int lda, int K, const bf16_t* __restrict__ W3,
                                                        const float* __restrict__ bias, int n_real, const int* __restrict__ ovf_count,
                                                        const int* __restrict__ ovf_rows, int* __restrict__ ids, float* __restrict__ maxv) {
  a16_kernel_enter();
  __shared__ float sv[16];
  __shared__ int si[16];
  const int lane = threadIdx.x & 63, l = lane & 15, g = threadIdx.x >> 4;
  const int nch = K >> 5, count = ovf_count[0];
  for (int o = blockIdx.x; o < count; o += gridDim.x) {
    const long long row = ovf_rows[o];
    float a[32];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 ah = *reinterpret_cast<const u32x4*>(A + row * lda + l * 32 + i * 8);
      const u32x4 al = *reinterpret_cast<const u32x4*>(A + row * lda + K + l * 32 + i * 8);
      const uint32_t hw[4] = {ah.x, ah.y, ah.z, ah.w}, lw[4] = {al.x, al.y, al.z, al.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[8 * i + 2 * j] = rbf2f(hw[j] & 0xFFFFu) + rbf2f(lw[j] & 0xFFFFu);
        a[8 * i + 2 * j + 1] = rbf2f(hw[j] >> 16) + rbf2f(lw[j] >> 16);
      }
    }
    float bv = -INFINITY;
    int bi = 0x7FFFFFFF;
    for (int n = g; n < n_real; n += 16) {
      const bf16_t* wt = W3 + (size_t)(n >> 6) * 3 * nch * 2048 + ((size_t)l * 64 + (n & 63)) * 32;
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 wh = *reinterpret_cast<const u32x4*>(wt + i * 8);
        const u32x4 wl = *reinterpret_cast<const u32x4*>(wt + (size_t)2 * nch * 2048 + i * 8);
        const uint32_t hw[4] = {wh.x, wh.y, wh.z, wh.w}, lw[4] = {wl.x, wl.y, wl.z, wl.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sum = fmaf(a[8 * i + 2 * j], rbf2f(hw[j] & 0xFFFFu) + rbf2f(lw[j] & 0xFFFFu), sum);
          sum = fmaf(a[8 * i + 2 * j + 1], rbf2f(hw[j] >> 16) + rbf2f(lw[j] >> 16), sum);
        }
      }
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
      const float v = sum + bias[n];
      if (v > bv) { bv = v; bi = n; }
    }
    __syncthreads();      // the previous row's sv / si have been read
    if (l == 0) { sv[g] = bv; si[g] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float b = sv[0];
      int ix = si[0];
      for (int k = 1; k < 16; ++k)
        if (sv[k] > b || (sv[k] == b && si[k] < ix)) { b = sv[k]; ix = si[k]; }
      ids[row] = ix;
      if (maxv) maxv[row] = b;
    }
  }
}

// hi/lo mode: A bf16 [M][hi(K) | lo(K)], W3 the three-pass tiling [N/64][3 K/32][64][32]; scratch: >= 256 + M * (2 * (1 + CAND_SLOTS) + 2) * 4 bytes.
// scratch additionally holds M floats (the first sweep's maxima).  n_real: classes below it carry weights.  K == 512 only (PT_ERR_INVALID otherwise: the caller falls back)
int pt_launch_gemm_argmax_x3(const bf16_t* A, long long M, int K, const bf16_t* W3, const float* bias, int N, int n_real, int* ids, float* maxv,
                             void* scratch, hipStream_t s) {
  if (K != 512 || N % 64 != 0 || M <= 0) return PT_ERR_INVALID;
  constexpr int SMEM = 64 * (512 * 2 + 16) + 64 * 4;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_cand_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_done = true;
  }
  unsigned* wmax = reinterpret_cast<unsigned*>(scratch);
  int* cand = reinterpret_cast<int*>(reinterpret_cast<char*>(scratch) + 256);
  float* rowmax = reinterpret_cast<float*>(cand + (size_t)M * 2 * (1 + CAND_SLOTS));
  int* ovf_count = reinterpret_cast<int*>(scratch) + 1;
  int* ovf_rows = reinterpret_cast<int*>(rowmax + M);
  static bool attr0 = false;
  if (!attr0) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_argmax_kernel<32, 0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr0 = true;
  }
  PT_HIP_CHECK(hipMemsetAsync(wmax, 0, 8, s));
  hipLaunchKernelGGL(wnorm_max_kernel, dim3(N / 64), dim3(256), 0, s, W3, N, K, wmax);
  const long long wts = (long long)3 * (K / 32) * 2048;
  // sweep 1: the single-pass maximum of every row (the bf16 mode's kernel on the hi halves); sweep 2: the classes within the bound of it
  if (cls_dma()) {
    if (cls_dma_attr() != PT_OK) return PT_ERR_HIP;
    hipLaunchKernelGGL(cls_argmax_dma_kernel<0>, dim3((unsigned)((M + 255) / 256)), dim3(512), CLS_DMA_SMEM, s, A, M, W3, bias, N, ids, rowmax, 2 * K, wts, nullptr, nullptr, nullptr);
  } else {
    hipLaunchKernelGGL((gemm_argmax_kernel<32, 0, 8>), dim3((unsigned)((M + 255) / 256)), dim3(512), SMEM, s, A, M, W3, bias, N, ids, rowmax, nullptr, 0,
                       nullptr, 2 * K, wts);      // eight waves per weight stage, as the bf16 mode's classifier
  }
  if (cls_dma())
    hipLaunchKernelGGL(cls_argmax_dma_kernel<1>, dim3((unsigned)((M + 255) / 256)), dim3(512), CLS_DMA_SMEM, s, A, M, W3, bias, N, nullptr, nullptr, 2 * K, wts,
                       reinterpret_cast<const float*>(wmax), rowmax, cand);
  else
    hipLaunchKernelGGL((gemm_cand_kernel<32>), dim3((unsigned)((M + 127) / 128)), dim3(256), SMEM, s, A, M, 2 * K, W3, wts, bias, N,
                       reinterpret_cast<const float*>(wmax), rowmax, cand);
  hipLaunchKernelGGL(cand_eval_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, s, A, M, 2 * K, K, W3, bias, N, n_real, cand, ids, maxv, ovf_count,
                     ovf_rows);
  hipLaunchKernelGGL(cand_full_kernel, dim3(2048), dim3(256), 0, s, A, 2 * K, K, W3, bias, n_real, ovf_count, ovf_rows, ids, maxv);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// A: bf16 [M][K] row-major, W: conv-tiled [N/64][K/32][64][32], bias fp32 [N]; K == 512 only (returns PT_ERR_INVALID otherwise
// so that the caller can fall back to the tiled kernel + reduce)
// Waves per workgroup of the streaming GEMMs.  Measured (rec-only bench, 5 082 lines): the K = 512 classifier 6.34 -> 5.68 ms with eight waves on
// one weight stage; the store-epilogue GEMMs do not gain (512 -> 2048: 1.00 -> 1.03 ms) or lose (256 -> 2048: 1.48 -> 1.81 ms: half as many
// workgroups hide each other's store tails), so they and the K = 192 head keep four.  PT_GEMM_NW=4 / 8 forces one size everywhere (A/B switch).
static int gemm_nw(int dflt) {
  static int nw = -1;
  if (nw < 0) {
    const char* ev = getenv("PT_GEMM_NW");
    nw = ev ? (atoi(ev) == 8 ? 8 : 4) : 0;
  }
  return nw ? nw : dflt;
}

int pt_launch_gemm_argmax(const bf16_t* A, long long M, int K, const bf16_t* W, const float* bias, int N, int* ids, float* maxv,
                          hipStream_t s) {
  if ((K != 512 && K != 192) || N % 64 != 0 || M <= 0) return PT_ERR_INVALID;      // 512: CRNN head; 192: ConvNextViT head
  constexpr int SMEM = 64 * (512 * 2 + 16) + 64 * 4;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_argmax_kernel<32, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_argmax_kernel<12, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_argmax_kernel<32, 0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_argmax_kernel<12, 0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_done = true;
  }
  const int nw = gemm_nw(K == 512 ? 8 : 4);
  const dim3 grid((unsigned)((M + nw * 32 - 1) / (nw * 32))), blk(nw * 64);
  if (K == 192) {
    if (nw == 8) hipLaunchKernelGGL((gemm_argmax_kernel<12, 0, 8>), grid, blk, 64 * (192 * 2 + 16) + 64 * 4, s, A, M, W, bias, N, ids, maxv, nullptr, 0, nullptr, 192, 6ll * 2048);
    else hipLaunchKernelGGL((gemm_argmax_kernel<12, 0>), grid, blk, 64 * (192 * 2 + 16) + 64 * 4, s, A, M, W, bias, N, ids, maxv, nullptr, 0, nullptr, 192, 6ll * 2048);
  } else {
    if (nw == 8 && cls_dma()) {
      if (cls_dma_attr() != PT_OK) return PT_ERR_HIP;
      hipLaunchKernelGGL(cls_argmax_dma_kernel<0>, grid, blk, CLS_DMA_SMEM, s, A, M, W, bias, N, ids, maxv, 512, 16ll * 2048, nullptr, nullptr, nullptr);
    } else if (nw == 8) hipLaunchKernelGGL((gemm_argmax_kernel<32, 0, 8>), grid, blk, SMEM, s, A, M, W, bias, N, ids, maxv, nullptr, 0, nullptr, 512, 16ll * 2048);
    else hipLaunchKernelGGL((gemm_argmax_kernel<32, 0>), grid, blk, SMEM, s, A, M, W, bias, N, ids, maxv, nullptr, 0, nullptr, 512, 16ll * 2048);
  }
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// out bf16 [M][N] = A [M][K] . W^T + bias (relu: 0 none, 1 ReLU, 4 GELU by gelu_poly); K in {256, 512}; PT_ERR_INVALID otherwise (caller falls back)
int pt_launch_gemm_rows(const bf16_t* A, long long M, int K, const bf16_t* W, const float* bias, int N, bf16_t* out, int relu,
                        hipStream_t s, const int* tlim) {
  if ((K != 512 && K != 256) || N % 64 != 0 || M <= 0) return PT_ERR_INVALID;
  const int nw = gemm_nw(4);
  const int smem = 64 * (K * 2 + 16) + 64 * 4 + nw * 32 * 80;       // weight stage + bias + the waves' store tiles
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_argmax_kernel<32, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (512 * 2 + 16) + 256 + 4 * 32 * 80));
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_argmax_kernel<32, 1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (512 * 2 + 16) + 256 + 8 * 32 * 80));
    attr_done = true;
  }
  const dim3 grid((unsigned)((M + nw * 32 - 1) / (nw * 32))), blk(nw * 64);
  if (K == 512) {
    if (nw == 8) hipLaunchKernelGGL((gemm_argmax_kernel<32, 1, 8>), grid, blk, smem, s, A, M, W, bias, N, nullptr, nullptr, out, relu, tlim, 512, 16ll * 2048);
    else hipLaunchKernelGGL((gemm_argmax_kernel<32, 1>), grid, blk, smem, s, A, M, W, bias, N, nullptr, nullptr, out, relu, tlim, 512, 16ll * 2048);
  } else {
    if (nw == 8) hipLaunchKernelGGL((gemm_argmax_kernel<16, 1, 8>), grid, blk, smem, s, A, M, W, bias, N, nullptr, nullptr, out, relu, tlim, 256, 8ll * 2048);
    else hipLaunchKernelGGL((gemm_argmax_kernel<16, 1>), grid, blk, smem, s, A, M, W, bias, N, nullptr, nullptr, out, relu, tlim, 256, 8ll * 2048);
  }
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// Row GEMM of the hi/lo (BF16X3) mode: out (hi | lo) [M][2 N] = A (hi | lo) [M][2 K] . W^T + bias over the three-pass tiling
// [w_hi | w_hi | w_lo] of the 1x1 conv kernel, in that kernel's order -- all (a_hi, w_hi) k-steps, then (a_lo, w_hi), then (a_hi, w_lo)
// -- so the accumulators hold the same sums.  The streaming shape of gemm_argmax_kernel<., 1>: a wave keeps BOTH halves of its 32
// rows in registers for the whole kernel (2 x KSTEPS x 4: 256 of the 512 registers a lone wave of a SIMD may hold at K = 512; two
// workgroups per CU at K = 256), W streams through LDS in 32-class stages of (w_hi, w_lo) rows (66 KB at K = 512, the next stage
// pre-fetched to registers), 3 KSTEPS MFMAs per stage and wave, and the stage's 32 x 32 (hi, lo) values leave through wave-private
// LDS tiles as 64-byte row segments.  The conv kernel ran these layers (the CRNN head's LSTM projections and embeddings) with a
// 128 x 64 tile, twelve MFMAs per wave between two barriers and an fp32-through-LDS epilogue.
// ---------------------------------------------------------------------------------------------------
template <int KSTEPS>
__global__ __launch_bounds__(256, KSTEPS == 32 ? 1 : 2) void gemm_rows_x3_kernel(const bf16_t* __restrict__ A, long long M, const bf16_t* __restrict__ W3,
                                                                                 const float* __restrict__ bias, int N, bf16_t* __restrict__ out, int relu,
                                                                                 const int* __restrict__ tlim) {
  a16_kernel_enter();
  constexpr int K = KSTEPS * 16, NCH = K / 32, P = K * 2 + 16;
  constexpr int NPF = NCH;                              // 16-byte pieces per thread per stage: 2 planes x 32 rows x K * 2 / 16 / 256
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sw = smem;                                      // [2 planes: w_hi, w_lo][32 classes][P]
  float* sb = reinterpret_cast<float*>(smem + 64 * P);  // [32] bias of the stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 31, q = lane >> 5;
  long long row0 = ((long long)blockIdx.x * 4 + wave) * 32;
  bool live = row0 < M;
  if (tlim) {      // ragged sequences: the compacted list of live row groups, as in gemm_argmax_kernel<., 1>
    const int total = tlim[0], g = blockIdx.x * 4 + wave;
    if ((int)blockIdx.x * 4 >= total) return;
    live = g < total;
    row0 = live ? (long long)tlim[1 + g] * 32 : 0;
  }
  const long long row = row0 + lx;
  const long long rc = row < M ? row : M - 1;
  bf16x8 ahi[KSTEPS], alo[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    ahi[ks] = *reinterpret_cast<const bf16x8*>(A + rc * (2 * K) + ks * 16 + q * 8);
    alo[ks] = *reinterpret_cast<const bf16x8*>(A + rc * (2 * K) + K + ks * 16 + q * 8);
  }
  u32x4 pf[NPF];
  float pb = 0.f;
  // stage st = 32 classes: half st & 1 of the 64-class tile st >> 1; piece idx of the stage: plane idx / (NCH * 128), chunk, class row, 16-byte part
  auto prefetch = [&](int st) {
    const bf16_t* wt = W3 + (size_t)(st >> 1) * (3 * NCH * 2048) + (st & 1) * (32 * 32);
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int idx = tid + j * 256, plane = idx / (NCH * 128), rem = idx - plane * (NCH * 128), c = rem >> 7, rp = rem & 127;
      pf[j] = *reinterpret_cast<const u32x4*>(wt + (size_t)(plane * 2 * NCH + c) * 2048 + rp * 8);
    }
    if (tid < 32) pb = bias[st * 32 + tid];
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int idx = tid + j * 256, plane = idx / (NCH * 128), rem = idx - plane * (NCH * 128), c = rem >> 7, r = (rem & 127) >> 2, part = rem & 3;
      *reinterpret_cast<u32x4*>(sw + (plane * 32 + r) * P + c * 64 + part * 16) = pf[j];
    }
    if (tid < 32) sb[tid] = pb;
  };
  char* tile = smem + 64 * P + 128 + wave * (2 * 32 * 80);      // hi rows (pitch 80 bytes), then the lo rows
  const int NS = N / 32;
  prefetch(0);
  for (int st = 0; st < NS; ++st) {
    __syncthreads();
    commit();
    __syncthreads();
    if (st + 1 < NS) prefetch(st + 1);
    if (!live) continue;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const char* wh = sw + lx * P + q * 16;
    const char* wl = sw + (32 + lx) * P + q * 16;
    // (a scheduling barrier every eight k-steps: left alone, hipcc hoists a pass's 32 fragment reads in front of its MFMAs and spills the rows)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      acc = mfma_32x32x16_a16(*reinterpret_cast<const bf16x8*>(wh + ks * 32), ahi[ks], acc);
      if ((ks & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      acc = mfma_32x32x16_a16(*reinterpret_cast<const bf16x8*>(wh + ks * 32), alo[ks], acc);
      if ((ks & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      acc = mfma_32x32x16_a16(*reinterpret_cast<const bf16x8*>(wl + ks * 32), ahi[ks], acc);
      if ((ks & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    // a lane owns a ROW: its 16 accumulators are classes (r & 3) + 8 (r >> 2) + 4 q of the stage; (hi, lo) through the wave's tiles
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int cl = rg * 8 + 4 * q;
      uint32_t hb[4], lb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v = acc[rg * 4 + k] + sb[cl + k];
        if (relu == 1) v = fmaxf(v, 0.f);
        hb[k] = rf2bf(v);
        lb[k] = rf2bf(v - rbf2f(hb[k]));
      }
      *reinterpret_cast<u32x2*>(tile + lx * 80 + cl * 2) = u32x2{hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
      *reinterpret_cast<u32x2*>(tile + 32 * 80 + lx * 80 + cl * 2) = u32x2{lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the tiles are wave-private: LDS operations of a wave complete in order
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (lane >> 2) + 16 * i;
      const u32x4 vh = *reinterpret_cast<const u32x4*>(tile + r * 80 + (lane & 3) * 16);
      const u32x4 vl = *reinterpret_cast<const u32x4*>(tile + 32 * 80 + r * 80 + (lane & 3) * 16);
      if (row0 + r < M) {
        bf16_t* op = out + (row0 + r) * (2 * (long long)N) + st * 32 + (lane & 3) * 8;
        *reinterpret_cast<u32x4*>(op) = vh;
        *reinterpret_cast<u32x4*>(op + N) = vl;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // reads done before the next stage overwrites the tiles
  }
}

// hi/lo rows (see the kernel); K in {256, 512}, N % 64 == 0; PT_ERR_INVALID otherwise (the caller falls back to the 1x1 conv kernel)
int pt_launch_gemm_rows_x3(const bf16_t* A, long long M, int K, const bf16_t* W3, const float* bias, int N, bf16_t* out, int relu, hipStream_t s,
                           const int* tlim) {
  if ((K != 512 && K != 256) || N % 64 != 0 || M <= 0 || (relu != 0 && relu != 1)) return PT_ERR_INVALID;
  const int smem = 64 * (K * 2 + 16) + 128 + 4 * 2 * 32 * 80;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rows_x3_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (512 * 2 + 16) + 128 + 4 * 2 * 32 * 80));
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rows_x3_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (256 * 2 + 16) + 128 + 4 * 2 * 32 * 80));
    attr_done = true;
  }
  const dim3 grid((unsigned)((M + 127) / 128)), blk(256);
  if (K == 512) hipLaunchKernelGGL((gemm_rows_x3_kernel<32>), grid, blk, smem, s, A, M, W3, bias, N, out, relu, tlim);
  else hipLaunchKernelGGL((gemm_rows_x3_kernel<16>), grid, blk, smem, s, A, M, W3, bias, N, out, relu, tlim);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_launch_argmax_reduce(const float* part, long long rows, int ntiles, int* ids, float* maxv, hipStream_t s) {
  if (rows <= 0) return PT_OK;
  int blocks = (int)((rows + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(argmax_reduce_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float2*>(part), rows, ntiles,
                     ids, maxv);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

}  // namespace PT_FMT_NS
