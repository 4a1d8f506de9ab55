// cvit_model.hip -- the ConvNextViT text-line recogniser on the engine's kernels.
//
// Reference graph: ConvNextViT.forward (model/convnext_vit/modeling_convnext_vit.py:38-45) = RGB -> gray, ConvNextModel
// (model/convnext_vit/modeling_convnext.py:29-131: patch embedding 4x4/4 + LayerNorm, four ConvNextStages of depths
// 3/3/8/3 and widths 96/192/256/512, stages 1..3 preceded by LayerNorm + a (2,1)-kernel (2,1)-stride conv; each ConvNextLayer
// = depthwise 7x7 -> LayerNorm -> Linear(C, 4C) -> GELU -> Linear(4C, C) -> layer scale -> + residual), then ViTForSTR
// (model/convnext_vit/modeling_vit.py:31-143: 1x1 patch projection 512 -> 192, position embeddings [1:], twelve pre-norm ViT
// layers with 3 heads of 64, final LayerNorm, the three 75-token chunks of a line stitched to 201 tokens, Linear(192 -> 7644))
// and the arg-max of OCRRecognitionPostProcessor (model/ocr_recognition/processor_ocr_recognition.py:147-150).
//
// Mapping onto kernels.  A line is three 32 x 300 chunks (processor_ocr_recognition.py:103-109); every chunk is one image of
// the CNN and one 75-token sequence of the ViT.  The residual stream is fp32 [pixels, C] for the whole network (the reference
// is fp32; both precision modes round only the GEMM / attention operands), token-wise work runs over ALL pixels / tokens of a
// micro-batch of lines at once:
//   patch embedding + LayerNorm          cvit_embed_kernel (K = 16: VALU; a lane owns one output pixel and its 96 channels)
//   depthwise 7x7 + LayerNorm            cvit_dwconv_ln_kernel<H> (a workgroup owns all H <= 8 rows of 8 columns of a chunk, a thread one
//                                        channel of them: every input loaded once; LayerNorm statistics through an LDS tile)
//   Linear C->4C + GELU, 4C->C + scale   bf16 mode, C <= 256 and the ViT MLP: ONE kernel, cvit_mlp_kernel (the hidden layer goes from
//   + residual                           MFMA accumulators to MFMA operands inside the lane); C = 512: row GEMM + tiled GEMM; hi/lo
//                                        mode: two 1x1 GEMMs on conv_igemm_kernel.  The layer scale is folded into the second
//                                        product's weights and bias; its epilogue adds the fp32 residual in place
//   down-sampler                         cvit_ln_kernel writes LayerNorm(x) of rows 2y / 2y+1 side by side: the (2,1) conv is a
//                                        1x1 GEMM with K = 2C
//   ViT LayerNorms, chunk stitching      cvit_ln_kernel (one wave per token; mode 2 gathers the 201 tokens of a line from its chunks)
//   q/k/v (one 192 -> 576 GEMM, 1/8 folded into q), attention output      1x1 GEMMs, fp32 residual epilogue
//   attention                            cvit_attention_kernel: one wave = 32 queries x one head, all 75 keys, on the matrix
//                                        cores (the Lore processor's scheme, lore_processor.hip, with d = 64)
//   classifier + arg-max                 bf16 mode: gemm_argmax_kernel (rec_kernels.hip, K = 192); hi/lo mode: 1x1 GEMM with the
//                                        arg-max epilogue + argmax_reduce_kernel.  No logits in HBM either way
// Chunks that hold no text (the line's resized width ends before them) are computed once per micro-batch and shared (forward_batch).
// PT_PRECISION_BF16X3: activations [hi | lo], weights [hi | hi | lo], three MFMA passes -- as everywhere in the engine.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "common.h"

namespace PT_FMT_NS {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8;
typedef __attribute__((ext_vector_type(16))) float af32x16;

constexpr int CV_T = 75;          // tokens per chunk = 300 / 4
constexpr int CV_PIX0 = 8 * CV_T; // pixels per chunk after the patch embedding (8 rows x 75)

// 16 stored bits <-> fp32 in the storage format of this namespace (act16.h: bf16, or IEEE half in pt_f16)
__device__ __forceinline__ float bf2f(uint32_t b) { return a16_to_f32(b); }
__device__ __forceinline__ uint32_t f2bf(float f) { return f32_to_a16(f); }
__device__ __forceinline__ void put(bf16_t* p, int lo_off, int split, float v) {
  const uint32_t h = f2bf(v);
  p[0] = (bf16_t)h;
  if (split) p[lo_off] = (bf16_t)f2bf(v - bf2f(h));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// ConvNextEmbeddings: Conv2d(1, 96, 4, stride 4) + LayerNorm(96, eps 1e-6) (HF modeling_convnext.py ConvNextEmbeddings).
// gray fp32; chunk c = 3 * line + j starts at gray + line * lstride + j * jstride, rows `pitch` apart.  A LANE owns one output
// pixel and all 96 channels of it: the weights are wave-uniform (scalar loads), the LayerNorm is in-lane -- no cross-lane step.
__global__ __launch_bounds__(256) void cvit_embed_kernel(const float* __restrict__ gray, int pitch, long long jstride, long long lstride,
                                                         int nchunks, const int* __restrict__ csrc, const float* __restrict__ w,
                                                         const float* __restrict__ b, const float* __restrict__ g,
                                                         const float* __restrict__ beta, float* __restrict__ x) {
  a16_kernel_enter();
  const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (long long)nchunks * CV_PIX0) return;
  // csrc != null: the batch holds only the chunks with text in them; csrc[k] = 3 line + j of compact chunk k, -1 = the
  // all-padding chunk (zeros)
  const int k = (int)(pix / CV_PIX0), r = (int)(pix % CV_PIX0), oy = r / CV_T, ox = r % CV_T;
  const int chunk = csrc ? csrc[k] : k;
  float in[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) in[i] = 0.f;
  if (chunk >= 0) {
    const float* src = gray + (long long)(chunk / 3) * lstride + (long long)(chunk % 3) * jstride + (size_t)(oy * 4) * pitch + ox * 4;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) in[dy * 4 + dx] = src[dy * pitch + dx];
  }
  float v[96];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 96; ++c) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) a = fmaf(w[c * 16 + k], in[k], a);
    v[c] = a + b[c];
    sum += v[c];
  }
  const float mean = sum / 96.f;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < 96; ++c) {
    v[c] -= mean;
    q += v[c] * v[c];
  }
  const float rstd = 1.f / sqrtf(q / 96.f + 1e-6f);
  float4* op = reinterpret_cast<float4*>(x + pix * 96);
#pragma unroll
  for (int c = 0; c < 96; c += 4)
    op[c >> 2] = make_float4(v[c] * rstd * g[c] + beta[c], v[c + 1] * rstd * g[c + 1] + beta[c + 1], v[c + 2] * rstd * g[c + 2] + beta[c + 2],
                             v[c + 3] * rstd * g[c + 3] + beta[c + 3]);
}

// ConvNextLayer, first half: depthwise Conv2d(C, C, 7, padding 3) + LayerNorm(C, eps 1e-6) over the fp32 stream
// x [B, H, W, C] -> bf16 [pixel][C] ([hi C | lo C] in the hi/lo mode).  The maps are H <= 8 rows high, less than the kernel:
// a workgroup owns ALL H rows of 8 consecutive columns of one chunk, thread c owns channel c of those H x 8 pixels, so every
// input value is loaded once (H x 14 loads for up to H x 8 x 49 multiply-adds).  LayerNorm: the H x 8 pixels go through an LDS
// tile [pixel][C]; a wave reduces whole pixels (two passes: mean, squared deviations), the statistics come back through LDS.
// grid (ceil(W / 8), B), block = C rounded up to waves, dynamic LDS = (H * 8 * C + 2 * H * 8) floats.  wt: [49][C] (tap-major).
constexpr int CV_TX = 8;
template <int H>
__global__ __launch_bounds__(512) void cvit_dwconv_ln_kernel(const float* __restrict__ x, int W, int C, const float* __restrict__ wt,
                                                             const float* __restrict__ bias, const float* __restrict__ g,
                                                             const float* __restrict__ beta, bf16_t* __restrict__ out, int split) {
  a16_kernel_enter();
  extern __shared__ float cv_lds[];
  constexpr int NP = H * CV_TX;
  float* tile = cv_lds;                 // [NP][C]
  float* stats = cv_lds + NP * C;       // [NP][2] = (mean, rstd)
  const int c = threadIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int x0 = blockIdx.x * CV_TX, b = blockIdx.y;
  const bool active = c < C;
  float acc[H][CV_TX];
#pragma unroll
  for (int oy = 0; oy < H; ++oy)
#pragma unroll
    for (int j = 0; j < CV_TX; ++j) acc[oy][j] = 0.f;
  if (active) {
    float wv[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) wv[k] = wt[k * C + c];
    const float* base = x + (size_t)b * H * W * C + c;
#pragma unroll
    for (int iy = 0; iy < H; ++iy) {
#pragma unroll
      for (int dx = 0; dx < CV_TX + 6; ++dx) {
        const int ix = x0 + dx - 3;
        const float v = (ix >= 0 && ix < W) ? base[((size_t)iy * W + ix) * C] : 0.f;
#pragma unroll
        for (int oy = 0; oy < H; ++oy) {
          const int dy = iy - oy + 3;
          if (dy < 0 || dy >= 7) continue;
#pragma unroll
          for (int j = 0; j < CV_TX; ++j) {
            const int k = dx - j;
            if (k >= 0 && k < 7) acc[oy][j] = fmaf(wv[dy * 7 + k], v, acc[oy][j]);      // explicit: the build runs with -ffp-contract=off
          }
        }
      }
    }
    const float bb = bias[c];
#pragma unroll
    for (int oy = 0; oy < H; ++oy)
#pragma unroll
      for (int j = 0; j < CV_TX; ++j) {
        acc[oy][j] += bb;
        tile[(oy * CV_TX + j) * C + c] = acc[oy][j];
      }
  }
  __syncthreads();
  for (int px = wave; px < NP; px += nw) {
    float v[8], s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int cc = lane + 64 * k;
      v[k] = cc < C ? tile[px * C + cc] : 0.f;
      s += v[k];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float d = (lane + 64 * k) < C ? v[k] - mean : 0.f;
      q += d * d;
    }
    q = wave_sum(q);
    if (lane == 0) {
      stats[2 * px] = mean;
      stats[2 * px + 1] = 1.f / sqrtf(q / (float)C + 1e-6f);
    }
  }
  __syncthreads();
  if (!active) return;
  const float gg = g[c], be = beta[c];
  const int mul = split ? 2 : 1;
#pragma unroll
  for (int oy = 0; oy < H; ++oy)
#pragma unroll
    for (int j = 0; j < CV_TX; ++j) {
      if (x0 + j >= W) break;
      const size_t pix = ((size_t)b * H + oy) * W + x0 + j;
      const int px = oy * CV_TX + j;
      put(out + pix * C * mul + c, C, split, (acc[oy][j] - stats[2 * px]) * stats[2 * px + 1] * gg + be);
    }
}

// LayerNorm (biased variance, eps inside the root: nn.LayerNorm) of fp32 rows [rows, C], C <= 512 -> bf16 (hi | lo).  One wave
// per row.  g == null: conversion only.
//   mode 0: out row = row
//   mode 1: ConvNextStage down-sampler input: row = (b, y, x) of a [B, H, W] map goes to row (b, y / 2, x) of a [B, H/2, W]
//           map with 2C channels, at channel offset (y & 1) * C -- the K layout of the (2,1)-kernel conv as a 1x1 GEMM
//   mode 2: chunk stitching (modeling_vit.py:133-138), gather form: OUTPUT row = (line, pos) of the 201-token sequence takes
//           token t of chunk j (pos < 69: chunk 0; < 132: chunk 1 from its token 6; else chunk 2 from its token 6); cmap != null:
//           chunk 3 line + j of the line is compact chunk cmap[3 line + j] of the batch (all-padding chunks share one)
__global__ __launch_bounds__(256) void cvit_ln_kernel(const float* __restrict__ x, long long rows, int C, const float* __restrict__ g,
                                                      const float* __restrict__ beta, float eps, bf16_t* __restrict__ out, int split,
                                                      int mode, int H, int W, const int* __restrict__ cmap) {
  a16_kernel_enter();
  long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  long long orow = row;
  int ocs = C, coff = 0;
  if (mode == 1) {
    const int xx = (int)(row % W);
    const long long t = row / W;
    const int y = (int)(t % H);
    const long long b = t / H;
    orow = (b * (H >> 1) + (y >> 1)) * W + xx;
    ocs = 2 * C;
    coff = (y & 1) * C;
  } else if (mode == 2) {
    const long long line = row / 201;
    const int pos = (int)(row % 201);
    const int j = pos < 69 ? 0 : (pos < 132 ? 1 : 2);
    const int t = j == 0 ? pos : (j == 1 ? pos - 69 + 6 : pos - 132 + 6);
    const long long chunk = cmap ? cmap[3 * line + j] : 3 * line + j;
    row = chunk * CV_T + t;                 // input row; orow stays (line, pos)
  }
  float v[8];
  const int nk = (C + 63) >> 6;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + 64 * k;
    v[k] = (k < nk && c < C) ? x[row * C + c] : 0.f;
    s += v[k];
  }
  float mean = 0.f, rstd = 1.f;
  if (g) {
    mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = lane + 64 * k;
      const float d = (k < nk && c < C) ? v[k] - mean : 0.f;
      q += d * d;
    }
    rstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
  }
  bf16_t* op = out + orow * ocs * (split ? 2 : 1) + coff;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + 64 * k;
    if (k < nk && c < C) put(op + c, ocs, split, g ? (v[k] - mean) * rstd * g[c] + beta[c] : v[k]);
  }
}

// x[row, c] += pos[row % 75, c]: embeddings + position_embeddings[:, 1:, :] (modeling_vit.py:78-79)
__global__ __launch_bounds__(256) void cvit_add_pos_kernel(float* __restrict__ x, const float* __restrict__ pos, long long total, int C) {
  a16_kernel_enter();
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long row = i / C;
  x[i] += pos[(row % CV_T) * C + (i % C)];
}

// ViT self-attention for one head (d = 64) and 32 queries of one 75-token chunk: one wave, on the matrix cores.
// qkv [tokens, 576] = [q | k | v] (hi/lo: [hi 576 | lo 576]), q already carries the 1/sqrt(64); out [tokens, 192].
//   S^T (32 keys x 32 queries) = K_tile Q^T over d = 64: four v_mfma_f32_32x32x16_bf16.  In the D layout a lane owns ONE
//   query and 16 of the 32 keys: the online soft-max is in-lane plus one exchange with lane ^ 32.
//   O^T (64 d x 32 queries) += V^T P^T as two 32-row blocks; P is the B operand exactly as the lane holds it (keys in the D
//   layout's own order), V is gathered in that key order.  Hi/lo mode: three passes each, fp32 soft-max.
__device__ __forceinline__ abf16x8 ld8(const bf16_t* p) { return *reinterpret_cast<const abf16x8*>(p); }

template <int SPLIT>
__global__ __launch_bounds__(64) void cvit_attention_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out) {
  a16_kernel_enter();
  const int q0 = blockIdx.x * 32, head = blockIdx.y, lane = threadIdx.x;
  const long long tok0 = (long long)blockIdx.z * CV_T;
  constexpr int E = 192, LO = 3 * E, cs = SPLIT ? 2 * LO : LO;
  const int col = lane & 31, half = lane >> 5;
  const int qi = q0 + col;
  const bf16_t* qrow = qkv + (size_t)(tok0 + (qi < CV_T ? qi : CV_T - 1)) * cs + head * 64 + half * 8;
  abf16x8 qh[4], ql[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qh[s] = ld8(qrow + 16 * s);
    if (SPLIT) ql[s] = ld8(qrow + LO + 16 * s);
  }
  af32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < CV_T; k0 += 32) {
    const int krow = k0 + col;
    const bf16_t* kp = qkv + (size_t)(tok0 + (krow < CV_T ? krow : CV_T - 1)) * cs + E + head * 64 + half * 8;
    af32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const abf16x8 kh = ld8(kp + 16 * s);
      sc = mfma_32x32x16_a16(kh, qh[s], sc);
      if (SPLIT) {
        sc = mfma_32x32x16_a16(kh, ql[s], sc);
        sc = mfma_32x32x16_a16(ld8(kp + LO + 16 * s), qh[s], sc);
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (key >= CV_T) sc[r] = -INFINITY;
      mt = fmaxf(mt, sc[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float mn = fmaxf(m, mt);
    const float scale = expf(m - mn);
    float p[16], lt = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { p[r] = expf(sc[r] - mn); lt += p[r]; }
    lt += __shfl_xor(lt, 32);
    l = l * scale + lt;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] *= scale; acc[1][r] *= scale; }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      abf16x8 ph, pl;
      int keys[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float pv = p[8 * s2 + j];
        const uint32_t hb = f2bf(pv);
        ph[j] = __builtin_bit_cast(__bf16, (uint16_t)hb);
        if (SPLIT) pl[j] = __builtin_bit_cast(__bf16, (uint16_t)f2bf(pv - bf2f(hb)));
        const int key = k0 + (j & 3) + 8 * (2 * s2 + (j >> 2)) + 4 * half;
        keys[j] = key < CV_T ? key : CV_T - 1;              // its probability is exactly 0
      }
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        abf16x8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bf16_t* vp = qkv + (size_t)(tok0 + keys[j]) * cs + 2 * E + head * 64 + db * 32 + col;     // A row = d
          vh[j] = __builtin_bit_cast(__bf16, vp[0]);
          if (SPLIT) vl[j] = __builtin_bit_cast(__bf16, vp[LO]);
        }
        acc[db] = mfma_32x32x16_a16(vh, ph, acc[db]);
        if (SPLIT) {
          acc[db] = mfma_32x32x16_a16(vh, pl, acc[db]);
          acc[db] = mfma_32x32x16_a16(vl, ph, acc[db]);
        }
      }
    }
  }
  if (qi < CV_T) {
    bf16_t* op = out + (size_t)(tok0 + qi) * (SPLIT ? 2 * E : E) + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) put(op + db * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, E, SPLIT, acc[db][r] / l);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused MLP of a ConvNextLayer / ViTLayer (bf16 mode): x += W2 . GELU(W1 . xb + b1) + b2 without the [rows, 4C] hidden tensor
// ever leaving the CU (unfused, it is written and read back once per layer: 16 of the 26 C bytes per row a layer moves).
//   workgroup = 128 rows, a wave owns 32 of them and keeps their xb rows (K = C) in registers as MFMA B fragments for the
//   whole kernel; the hidden layer is walked in chunks of 32 units.  Per chunk the W1 rows [32][C] and the W2 columns
//   [C][32] sit in LDS (double-buffered, next chunk pre-fetched to registers while this one is multiplied):
//     D1 [32 hidden][32 rows] = W1_chunk . xb^T        C / 16 v_mfma_f32_32x32x16_bf16, W1 as the A operand
//     bias + GELU + bf16 in the lane: in the D layout a lane owns ONE row and 16 hidden units -- which is exactly a B operand
//     of the second product if its K order is the D layout's own order (a sum does not care), so the hidden values go from
//     accumulators to operands without leaving the lane; W2's columns are stored in that order (weights.py)
//     D2 [C out][32 rows] += W2_chunk . H              2 x C / 32 MFMAs, W2 as the A operand
//   An MFMA here reads 1 KB of LDS (the weight fragment; the row operand is in registers): LDS and matrix pipe are balanced,
//   neither HBM (10 C bytes per row instead of 26 C) nor the launch count (one kernel instead of two) is the bound.
//   Epilogue: the lane's 4-channel fp32 runs of its row are added to the residual stream in place.
// GELU: a packed-fp32 polynomial (gelu_pair below, absolute error < 3e-6) -- the hidden value is rounded to bf16 right after,
// 2^-9 relative.  The hi/lo mode keeps the two-GEMM path with erff.
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) uint32_t cu32x4;

typedef __attribute__((ext_vector_type(2))) float mlp_f2;
typedef __attribute__((ext_vector_type(2))) __bf16 mlp_b2;

// GELU of two values at once in packed fp32 (v_pk_fma_f32): x * Phi(x) with Phi(x) - 1/2 = x * Q(x^2), Q the degree-9 polynomial
// of a Chebyshev fit on [-4, 4] (|error of Phi| < 8e-7 there); outside, x is clamped in Phi (Phi(4) = 1 - 3.2e-5) and, on the
// negative side, in the product too (GELU(x < -4) = -1.3e-4 instead of -> 0).  Absolute error < 3e-6 on [-4, 4], relative error
// < 5e-4 wherever |GELU| > 1e-3 -- the value is rounded to bf16 (2^-9) right after.  14 packed instructions per pair against
// ~34 per value for the erf form (exp, reciprocal, unfused multiply-adds under -ffp-contract=off): the kernel is VALU-bound here.
__device__ __forceinline__ mlp_f2 gelu_pair(mlp_f2 x) {
  const mlp_f2 xp = __builtin_elementwise_max(x, (mlp_f2){-4.f, -4.f});
  const mlp_f2 xc = __builtin_elementwise_min(xp, (mlp_f2){4.f, 4.f});
  const mlp_f2 t = xc * xc;
  constexpr float A[10] = {3.989380888e-01f, -6.647037283e-02f, 9.945140159e-03f, -1.168552637e-03f, 1.084709610e-04f,
                           -7.841504780e-06f, 4.224180292e-07f, -1.572596130e-08f, 3.561182861e-10f, -3.658831230e-12f};
  mlp_f2 q = {A[9], A[9]};
#pragma unroll
  for (int k = 8; k >= 0; --k) q = __builtin_elementwise_fma(q, t, (mlp_f2){A[k], A[k]});
  return xp * __builtin_elementwise_fma(xc, q, (mlp_f2){0.5f, 0.5f});
}

template <int C, bool PIPE, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void cvit_mlp_kernel(const bf16_t* __restrict__ xb, const bf16_t* __restrict__ w1,
                                                                            const float* __restrict__ b1, const bf16_t* __restrict__ w2p,
                                                                            const float* __restrict__ b2, float* __restrict__ x, long long rows) {
  a16_kernel_enter();
  constexpr int KS = C / 16, NT = C / 32, NCH = 4 * C / 32;
  constexpr int U1 = C / 8;                            // 16-byte units per W1 row
  constexpr int S1 = 32 * C * 2, S2 = C * 64;          // one W1 chunk [32][C], one W2 chunk [C][32], bytes
  constexpr int NI = C / 16;                           // DMA wave-instructions (64 units of 16 bytes) per chunk and matrix
  constexpr int SLOTS = (NI + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) char cv_mlp_lds[];
  char* const w1buf = cv_mlp_lds;                      // two W1 chunks, then two W2 chunks, then b1
  char* const w2buf = cv_mlp_lds + 2 * S1;
  float* const b1s = reinterpret_cast<float*>(cv_mlp_lds + 2 * S1 + 2 * S2);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 31, half = lane >> 5;
  // NW waves of 32 rows share one copy of the weight chunks (the DMA traffic per row halves from 4 to 8 waves); waves past the
  // last row still take part in the DMA and the barriers
  const long long row = (long long)blockIdx.x * (32 * NW) + wave * 32 + col;
  const bool live = row < rows;
  for (int i = tid; i < 4 * C; i += 64 * NW) b1s[i] = b1[i];
  // The weight chunks go global -> LDS by DMA (global_load_lds: no staging registers, no ds_write): a wave instruction
  // fills 1 KB of LDS in lane order, the lane picks the SOURCE.  No row padding is possible that way, so the 16-byte units
  // of a row are XOR-swizzled with the row number instead (rows are a multiple of 64 bytes apart): eight consecutive rows
  // read at the same unit index then sit in eight different 16-byte bank groups -- conflict-free ds_read_b128.
  const int sw1 = C == 96 ? ((col >> 1) & 3) : (col & 7);            // W1: row = the lane's hidden unit
  const int sw2 = (col >> 1) & 3;                                     // W2: row = the lane's output channel
  int src1[SLOTS], src2[SLOTS];
#pragma unroll
  for (int j = 0; j < SLOTS; ++j) {
    const int U = (wave + NW * j) * 64 + lane;
    const int r1 = U / U1, u1 = U % U1;
    src1[j] = r1 * (2 * C) + ((u1 ^ (C == 96 ? ((r1 >> 1) & 3) : (r1 & 7))) << 4);
    const int r2 = U >> 2, u2 = U & 3;
    src2[j] = r2 * 64 + ((u2 ^ ((r2 >> 1) & 3)) << 4);
  }
  // DMA addresses as (kernel-argument base in SGPRs) + (32-bit lane offset incl. the chunk offset): written as pointer + 64-bit
  // chunk offset, the loop's chunk pointers become 64-bit per-lane induction variables -- sixteen more VGPRs in a kernel that
  // has none to spare; they were spilled, and every reload waited (vmcnt, in order) for the DMA issued just before it: 3.8 k of
  // the 8.9 k cycles a chunk took at C = 256 (s_memtime stamps)
  auto issue1 = [&](int hc, char* buf) {
    const char* g = reinterpret_cast<const char*>(w1);
    unsigned base = (unsigned)hc * S1;
    asm volatile("" : "+s"(base));        // opaque to loop strength reduction
#pragma unroll
    for (int j = 0; j < SLOTS; ++j)
      if (wave + NW * j < NI)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (base + (unsigned)src1[j])),
                                         (__attribute__((address_space(3))) void*)(buf + (wave + NW * j) * 1024), 16, 0, 0);
  };
  auto issue2 = [&](int hc, char* buf) {
    const char* g = reinterpret_cast<const char*>(w2p);
    unsigned base = (unsigned)hc * S2;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int j = 0; j < SLOTS; ++j)
      if (wave + NW * j < NI)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (base + (unsigned)src2[j])),
                                         (__attribute__((address_space(3))) void*)(buf + (wave + NW * j) * 1024), 16, 0, 0);
  };
  abf16x8 xf[KS];
  {
    const bf16_t* xr = xb + (live ? row : 0) * C + half * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[s] = ld8(xr + 16 * s);
  }
  af32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  auto gemm1 = [&](const char* buf) {
    af32x16 d;
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = 0.f;
    const char* a1 = buf + col * (2 * C);
#pragma unroll
    for (int s = 0; s < KS; ++s)
      d = mfma_32x32x16_a16(*reinterpret_cast<const abf16x8*>(a1 + (((2 * s + half) ^ sw1) << 4)), xf[s], d);
    return d;
  };
  // second product of one chunk: the two k-steps of a tile are dependent, so all first steps are issued before the second ones
  auto gemm2 = [&](const char* buf, const abf16x8 (&h)[2]) {
    const char* a2 = buf + col * 64;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = mfma_32x32x16_a16(*reinterpret_cast<const abf16x8*>(a2 + t * 2048 + (((2 * s2 + half) ^ sw2) << 4)), h[s2],
                                                          acc[t]);
  };
  // bias + GELU + bf16 (v_cvt_pk_bf16_f32: round to nearest even, two values per instruction) of the value PAIRS [p0, p1)
  auto gelu16 = [&](const af32x16& d, const float* bp, abf16x8 (&h)[2], int p0, int p1) {
#pragma unroll
    for (int q = p0; q < p1; ++q) {
      const int r = 2 * q;
      const mlp_f2 v = gelu_pair((mlp_f2){d[r] + bp[8 * (r >> 2) + (r & 3)], d[r + 1] + bp[8 * (r >> 2) + (r & 3) + 1]});
      const mlp_b2 pk = __builtin_bit_cast(mlp_b2, pack_a16x2(v[0], v[1]));      // bit containers: the storage format is act16.h's
      h[r >> 3][r & 7] = pk[0];
      h[r >> 3][(r & 7) + 1] = pk[1];
    }
  };
  abf16x8 hf[2];
  if (PIPE) {
    // software pipeline (C = 96 / 192): in iteration k the second product of chunk k (MFMA) and the bias + GELU of chunk k + 1
    // (VALU) are independent instruction streams of the SAME wave, interleaved in program order so that the matrix pipe and
    // the vector ALU work at the same time.  W1 is staged two chunks ahead, W2 one chunk ahead: two buffers each, one
    // barrier per chunk.  (C = 256 has no registers left for the second set of hidden values.)
    issue1(0, w1buf);
    issue2(0, w2buf);
    issue1(1, w1buf + S1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    abf16x8 hn[2];
    gelu16(gemm1(w1buf), b1s + 4 * half, hf, 0, 8);
    for (int hc = 0; hc < NCH; ++hc) {
      const bool more = hc + 1 < NCH;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();            // chunk hc's W2 and chunk hc + 1's W1 have landed; the buffers of hc - 1 are free
      if (hc + 2 < NCH) issue1(hc + 2, w1buf + (hc & 1) * S1);
      if (more) issue2(hc + 1, w2buf + ((hc + 1) & 1) * S2);
      af32x16 d1;
      const float* bp = b1s + (more ? hc + 1 : hc) * 32 + 4 * half;
      if (more) d1 = gemm1(w1buf + ((hc + 1) & 1) * S1);
      const char* a2 = w2buf + (hc & 1) * S2 + col * 64;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[t] = mfma_32x32x16_a16(*reinterpret_cast<const abf16x8*>(a2 + t * 2048 + (((2 * s2 + half) ^ sw2) << 4)),
                                                            hf[s2], acc[t]);
          if (more) gelu16(d1, bp, hn, (8 * (s2 * NT + t)) / (2 * NT), (8 * (s2 * NT + t + 1)) / (2 * NT));
        }
      hf[0] = hn[0];
      hf[1] = hn[1];
    }
  } else {
    issue1(0, w1buf);
    issue2(0, w2buf);
    for (int hc = 0; hc < NCH; ++hc) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (hc + 1 < NCH) {
        issue1(hc + 1, w1buf + ((hc + 1) & 1) * S1);
        issue2(hc + 1, w2buf + ((hc + 1) & 1) * S2);
      }
      gelu16(gemm1(w1buf + (hc & 1) * S1), b1s + hc * 32 + 4 * half, hf, 0, 8);
      gemm2(w2buf + (hc & 1) * S2, hf);
    }
  }
  if (!live) return;
  float* xr = x + row * C + 4 * half;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int o = 32 * t + 8 * g4;
      float4 r = *reinterpret_cast<const float4*>(xr + o);
      const float4 bb = *reinterpret_cast<const float4*>(b2 + o + 4 * half);
      r.x += acc[t][4 * g4] + bb.x; r.y += acc[t][4 * g4 + 1] + bb.y; r.z += acc[t][4 * g4 + 2] + bb.z; r.w += acc[t][4 * g4 + 3] + bb.w;
      *reinterpret_cast<float4*>(xr + o) = r;
    }
}

template <int C, bool PIPE, int NW>
int launch_mlp(const bf16_t* xb, const bf16_t* w1, const float* b1, const bf16_t* w2p, const float* b2, float* x, long long rows_pad,
               hipStream_t s) {
  constexpr int SMEM = 2 * (32 * C * 2 + C * 64) + 4 * C * 4;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cvit_mlp_kernel<C, PIPE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_done = true;
  }
  hipLaunchKernelGGL((cvit_mlp_kernel<C, PIPE, NW>), dim3((unsigned)((rows_pad + 32 * NW - 1) / (32 * NW))), dim3(64 * NW), SMEM, s, xb, w1, b1, w2p,
                     b2, x, rows_pad);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

struct Net {
  pt_engine* e;
  const PtModel* m;
  hipStream_t s;
  int x3, mul, rc;
  const PtTensor* get(const std::string& n) {
    const PtTensor* t = m->find(n);
    if (!t && rc == PT_OK) {
      pt_set_error("ConvNextViT weight blob lacks tensor '%s'", n.c_str());
      rc = PT_ERR_FORMAT;
    }
    return t;
  }
  const float* f32(const std::string& n) {
    const PtTensor* t = get(n);
    return t ? reinterpret_cast<const float*>(t->d_ptr) : nullptr;
  }
  // y = x W^T + b over `rows` (padded to 128) rows: x bf16 [rows, cin] -> bf16 [rows, N] (act: 0 none, 4 GELU), or the fp32
  // stream [rows, f32_cs] (+ fp32 residual `res`, in place when res == out_f32); nv: real output channels when N is padded
  void gemm(const bf16_t* x, long long rows, int cin, const std::string& q, int N, int act, bf16_t* out, float* out_f32 = nullptr,
            int f32_cs = 0, const float* res = nullptr, int nv = 0, float* argmax_part = nullptr) {
    const PtTensor* w = get(q + (x3 ? ".w3" : ".w"));
    const PtTensor* b = get(q + ".b");
    if (rc != PT_OK) return;
    ConvDesc c;
    c.in = x; c.B = 1; c.H = (int)(rows / 32); c.W = 32; c.Cin = cin;
    c.w = reinterpret_cast<const bf16_t*>(w->d_ptr); c.bias = reinterpret_cast<const float*>(b->d_ptr);
    c.N = N; c.ks = 1; c.stride = 1; c.relu = act; c.split = x3; c.n_valid = nv;
    if (argmax_part) {
      c.argmax_part = argmax_part;
    } else if (out_f32) {
      c.out_f32 = out_f32; c.out_cstride = f32_cs; c.res_f32 = res;
    } else {
      c.out = out; c.out_cstride = N * mul; c.out_coff = 0; c.out_lo_off = N;
    }
    const int r = pt_launch_conv(e, c, s);
    if (r != PT_OK) rc = r;
  }
};

inline long long pad128(long long r) { return (r + 127) / 128 * 128; }

// the MLP of layer `q` (ConvNextLayer: pw1 / pw2 with the layer scale folded in; ViTLayer: fc1 / fc2) over the fp32 stream x:
// one fused kernel in bf16 mode for C = 96 / 192 / 256 (PT_CVIT_FUSED_MLP=0: never), else two GEMMs through the hidden tensor
void mlp(Net& p, const bf16_t* xb, bf16_t* hb, float* x, long long rows_pad, int C, const std::string& q, const char* n1, const char* n2) {
  static int fused = -1;
  if (fused < 0) {
    const char* ev = getenv("PT_CVIT_FUSED_MLP");
    fused = ev ? atoi(ev) : 1;
  }
  if (fused && !p.x3 && (C == 96 || C == 192 || C == 256)) {
    const PtTensor *w1 = p.get(q + ".mlp.w1"), *w2 = p.get(q + ".mlp.w2p");
    const float *b1 = p.f32(q + "." + n1 + ".b"), *b2 = p.f32(q + "." + n2 + ".b");
    if (p.rc != PT_OK) return;
    PtProfScope ps(p.e, p.s, PT_PROF_CONV1X1, 16.0 * rows_pad * (double)C * C, "cvit fused mlp");
    const bf16_t* W1 = reinterpret_cast<const bf16_t*>(w1->d_ptr);
    const bf16_t* W2 = reinterpret_cast<const bf16_t*>(w2->d_ptr);
    // 128-row workgroups (NW = 4), two per CU; 256-row ones (NW = 8, one per CU, half the weight DMA per row) measured equal at
    // C = 256 and 5-10 % slower at C = 96 / 192
    int r;
    if (C == 96) r = launch_mlp<96, true, 4>(xb, W1, b1, W2, b2, x, rows_pad, p.s);
    else if (C == 192) r = launch_mlp<192, true, 4>(xb, W1, b1, W2, b2, x, rows_pad, p.s);
    else r = launch_mlp<256, false, 4>(xb, W1, b1, W2, b2, x, rows_pad, p.s);
    if (r != PT_OK) p.rc = r;
    return;
  }
  const int Np = (C + 63) / 64 * 64;
  if (fused && !p.x3 && C == 512) {
    // stage 3 (the fused kernel's accumulators do not fit at C = 512): the first product on the streaming row GEMM of the CRNN
    // head (rows of A in registers, W through LDS; 2048-wide N: 0.64 -> 0.3x ms per 115 k rows) with the GELU in its epilogue
    const PtTensor *w = p.get(q + "." + n1 + ".w"), *b = p.get(q + "." + n1 + ".b");
    if (p.rc != PT_OK) return;
    PtProfScope ps(p.e, p.s, PT_PROF_CONV1X1, 2.0 * rows_pad * (double)C * 4 * C, "cvit rows gemm + gelu");
    const int r = pt_launch_gemm_rows(xb, rows_pad, C, reinterpret_cast<const bf16_t*>(w->d_ptr), reinterpret_cast<const float*>(b->d_ptr),
                                      4 * C, hb, 4, p.s);
    if (r != PT_OK) p.rc = r;
  } else {
    p.gemm(xb, rows_pad, C, q + "." + n1, 4 * C, 4, hb);
  }
  p.gemm(hb, rows_pad, 4 * C, q + "." + n2, Np, 0, nullptr, x, C, x, Np != C ? C : 0);
}

// lines [0, n) of one micro-batch
int forward_batch(pt_engine* e, const PtModel& M, const float* gray, int pitch, long long jstride, long long lstride, int n, int32_t* ids,
                  float* maxlogit, hipStream_t s, const int* h_tw) {
  Net p;
  p.e = e; p.m = &M; p.s = s; p.rc = PT_OK;
  p.x3 = pt_split(e) ? 1 : 0;
  p.mul = p.x3 ? 2 : 1;
  const int x3 = p.x3, mul = p.mul, NT = 7680 / 64;
  // h_tw != null: the lines' text widths after the keep-ratio resize are known.  Chunk j of a line is columns [252 j, 252 j +
  // 300): with a text width <= 252 j it is all padding, and every all-padding chunk yields the same 75 tokens -- the CNN and
  // the ViT never look across chunks.  The batch then holds the chunks with text plus ONE all-padding chunk; the stitching
  // gathers through cmap.  Bit-identical to computing every chunk (a chunk's result does not depend on its batch).
  std::vector<int>& cmap = e->cvit_maps[e->cvit_slot][0];
  std::vector<int>& csrc = e->cvit_maps[e->cvit_slot][1];
  e->cvit_slot = (e->cvit_slot + 1) & 15;
  csrc.clear();
  if (h_tw) {
    cmap.assign((size_t)3 * n, -1);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < 3; ++j)
        if (h_tw[i] > PT_CVIT_CHUNK_STEP * j) {
          cmap[3 * i + j] = (int)csrc.size();
          csrc.push_back(3 * i + j);
        }
    if ((int)csrc.size() < 3 * n) {
      const int zero = (int)csrc.size();
      csrc.push_back(-1);
      for (int& c : cmap)
        if (c < 0) c = zero;
    }
  }
  const int nchunks = h_tw ? (int)csrc.size() : 3 * n;
  static const int DEPTH[4] = {3, 3, 8, 3}, DIM[4] = {96, 192, 256, 512};
  const long long rows_cls = pad128((long long)n * 201), Tpad = pad128((long long)nchunks * CV_T);
  // the stream / its bf16 image / the MLP hidden layer are reused by every stage: size them for the widest (rows are padded
  // to 128 per stage, so a later, shorter stage can be the larger one)
  long long xel = Tpad * 192;
  for (int st = 0; st < 4; ++st) {
    const long long el = pad128((long long)nchunks * (CV_PIX0 >> st)) * DIM[st];
    if (el > xel) xel = el;
  }
  float* x = nullptr;
  bf16_t *xb = nullptr, *hb = nullptr, *qkv = nullptr, *att = nullptr, *feat = nullptr;
  float* part = nullptr;
  int *d_cmap = nullptr, *d_csrc = nullptr;
  for (int attempt = 0; attempt < 2; ++attempt) {
    PtArena& A = e->arenas[PT_ARENA_REC];
    A.reset();
    bool ok = true;
    auto take = [&](size_t bytes) { void* q = A.take(bytes); if (!q) ok = false; return q; };
    x = reinterpret_cast<float*>(take((size_t)(xel + 128) * sizeof(float)));      // + 128: the padded-N epilogue reads up to 32 floats
    xb = reinterpret_cast<bf16_t*>(take((size_t)xel * mul * sizeof(bf16_t)));      //   past the last 96-wide row
    hb = reinterpret_cast<bf16_t*>(take((size_t)xel * 4 * mul * sizeof(bf16_t)));
    qkv = reinterpret_cast<bf16_t*>(take((size_t)Tpad * 576 * mul * sizeof(bf16_t)));
    att = reinterpret_cast<bf16_t*>(take((size_t)Tpad * 192 * mul * sizeof(bf16_t)));
    feat = reinterpret_cast<bf16_t*>(take((size_t)rows_cls * 192 * mul * sizeof(bf16_t)));
    part = reinterpret_cast<float*>(take((size_t)rows_cls * NT * 2 * sizeof(float)));
    if (h_tw) {
      d_cmap = reinterpret_cast<int*>(take(cmap.size() * sizeof(int)));
      d_csrc = reinterpret_cast<int*>(take(csrc.size() * sizeof(int)));
    }
    if (ok) break;
    if (attempt == 1) {
      pt_set_error("activation arena allocation failed");
      return PT_ERR_HIP;
    }
    PT_HIP_CHECK(hipDeviceSynchronize());
    if (A.base) PT_HIP_CHECK(hipFree(A.base));
    A.base = nullptr;
    const size_t want = pt_arena_round(A.high);
    PT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&A.base), want));
    A.cap = want;
  }
  if (h_tw) {
    // sources of asynchronous copies: a pinned slot of the engine's staging ring, re-used only after the event recorded behind these
    // copies has completed (a pageable vector in a fixed-depth ring could be rewritten by a host that runs far ahead of the GPU)
    int* hs = static_cast<int*>(e->stage_ring.acquire((cmap.size() + csrc.size()) * sizeof(int)));
    PT_REQUIRE(hs, "ConvNextViT: pinned staging (%s)", hipGetErrorString(hipGetLastError()));
    memcpy(hs, cmap.data(), cmap.size() * sizeof(int));
    memcpy(hs + cmap.size(), csrc.data(), csrc.size() * sizeof(int));
    PT_HIP_CHECK(hipMemcpyAsync(d_cmap, hs, cmap.size() * sizeof(int), hipMemcpyHostToDevice, s));
    PT_HIP_CHECK(hipMemcpyAsync(d_csrc, hs + cmap.size(), csrc.size() * sizeof(int), hipMemcpyHostToDevice, s));
    PT_REQUIRE(e->stage_ring.release(s) == 0, "ConvNextViT: event record failed");
  }
  // rows past the real ones are never written by the row kernels: clear once so that no NaN bit pattern reaches a GEMM
  PT_HIP_CHECK(hipMemsetAsync(x, 0, (size_t)(xel + 128) * sizeof(float), s));
  PT_HIP_CHECK(hipMemsetAsync(xb, 0, (size_t)xel * mul * sizeof(bf16_t), s));
  PT_HIP_CHECK(hipMemsetAsync(att, 0, (size_t)Tpad * 192 * mul * sizeof(bf16_t), s));
  PT_HIP_CHECK(hipMemsetAsync(feat, 0, (size_t)rows_cls * 192 * mul * sizeof(bf16_t), s));
  {
    const float *w = p.f32("embed.w"), *b = p.f32("embed.b"), *g = p.f32("embed.ln.g"), *be = p.f32("embed.ln.b");
    if (p.rc != PT_OK) return p.rc;
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "cvit embed");
    const long long pix = (long long)nchunks * CV_PIX0;
    hipLaunchKernelGGL(cvit_embed_kernel, dim3((unsigned)((pix + 255) / 256)), dim3(256), 0, s, gray, pitch, jstride, lstride, nchunks, d_csrc, w, b, g, be, x);
  }
  auto ln = [&](const std::string& q, long long rows, int C, float eps, bf16_t* out, int mode, int H, int W) {
    const float* g = q.empty() ? nullptr : p.f32(q + ".g");
    const float* be = q.empty() ? nullptr : p.f32(q + ".b");
    if (p.rc != PT_OK) return;
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "cvit layernorm");
    hipLaunchKernelGGL(cvit_ln_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, C, g, be, eps, out, x3, mode, H, W, d_cmap);
  };
  int H = 8;
  for (int st = 0; st < 4; ++st) {
    const int C = DIM[st];
    const std::string sq = "s" + std::to_string(st);
    if (st > 0) {
      const int Cp = DIM[st - 1];
      ln(sq + ".down.ln", (long long)nchunks * H * CV_T, Cp, 1e-6f, xb, 1, H, CV_T);
      H >>= 1;
      p.gemm(xb, pad128((long long)nchunks * H * CV_T), 2 * Cp, sq + ".down", C, 0, nullptr, x, C);
    }
    const long long rows = (long long)nchunks * H * CV_T, rp = pad128(rows);
    for (int l = 0; l < DEPTH[st]; ++l) {
      const std::string lq = sq + ".l" + std::to_string(l);
      const float *w = p.f32(lq + ".dw.w"), *b = p.f32(lq + ".dw.b"), *g = p.f32(lq + ".ln.g"), *be = p.f32(lq + ".ln.b");
      if (p.rc != PT_OK) return p.rc;
      {
        PtProfScope ps(e, s, PT_PROF_OTHER, 0, "cvit dwconv7+ln");
        const dim3 grid((CV_T + CV_TX - 1) / CV_TX, nchunks), block((C + 63) / 64 * 64);
        const size_t lds = ((size_t)H * CV_TX * C + 2 * H * CV_TX) * sizeof(float);
        if (H == 8) hipLaunchKernelGGL(cvit_dwconv_ln_kernel<8>, grid, block, lds, s, x, CV_T, C, w, b, g, be, xb, x3);
        else if (H == 4) hipLaunchKernelGGL(cvit_dwconv_ln_kernel<4>, grid, block, lds, s, x, CV_T, C, w, b, g, be, xb, x3);
        else if (H == 2) hipLaunchKernelGGL(cvit_dwconv_ln_kernel<2>, grid, block, lds, s, x, CV_T, C, w, b, g, be, xb, x3);
        else hipLaunchKernelGGL(cvit_dwconv_ln_kernel<1>, grid, block, lds, s, x, CV_T, C, w, b, g, be, xb, x3);
      }
      mlp(p, xb, hb, x, rp, C, lq, "pw1", "pw2");
      if (p.rc != PT_OK) return p.rc;
    }
  }
  // ---- ViT over [nchunks * 75, 192]; the CNN's last_hidden_state is the raw stream (modeling_convnext.py:117,127)
  const long long T = (long long)nchunks * CV_T, Tp = Tpad;
  ln("", T, 512, 0.f, xb, 0, 1, 1);
  p.gemm(xb, Tp, 512, "vit.embed", 192, 0, nullptr, x, 192);
  {
    const float* pos = p.f32("vit.pos");
    if (p.rc != PT_OK) return p.rc;
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "cvit add pos");
    hipLaunchKernelGGL(cvit_add_pos_kernel, dim3((unsigned)((T * 192 + 255) / 256)), dim3(256), 0, s, x, pos, T * 192, 192);
  }
  for (int l = 0; l < 12; ++l) {
    const std::string lq = "vit.l" + std::to_string(l);
    ln(lq + ".ln1", T, 192, 1e-12f, xb, 0, 1, 1);
    p.gemm(xb, Tp, 192, lq + ".qkv", 576, 0, qkv);
    if (p.rc != PT_OK) return p.rc;
    {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "cvit attention");
      if (x3) hipLaunchKernelGGL(cvit_attention_kernel<1>, dim3(3, 3, nchunks), dim3(64), 0, s, qkv, att);
      else hipLaunchKernelGGL(cvit_attention_kernel<0>, dim3(3, 3, nchunks), dim3(64), 0, s, qkv, att);
    }
    p.gemm(att, Tp, 192, lq + ".out", 192, 0, nullptr, x, 192, x);
    ln(lq + ".ln2", T, 192, 1e-12f, xb, 0, 1, 1);
    mlp(p, xb, hb, x, Tp, 192, lq, "fc1", "fc2");
    if (p.rc != PT_OK) return p.rc;
  }
  ln("vit.ln", (long long)n * 201, 192, 1e-12f, feat, 2, 1, 1);
  static int cls_fused = -1;      // PT_CLS_FUSED=0: tiled GEMM with per-tile arg-max partials + reduce in bf16 mode too (A/B switch)
  if (cls_fused < 0) {
    const char* ev = getenv("PT_CLS_FUSED");
    cls_fused = ev ? atoi(ev) : 1;
  }
  if (!x3 && cls_fused) {
    // bf16 mode: the streaming classifier of the CRNN head (gemm_argmax_kernel, K = 192 here): W as the MFMA A operand, the
    // running arg-max in the lane, no partials
    const PtTensor *w = p.get("cls.w"), *b = p.get("cls.b");
    if (p.rc != PT_OK) return p.rc;
    PtProfScope ps(e, s, PT_PROF_CONV1X1, 2.0 * n * 201 * 192.0 * 7680.0, "cvit classifier gemm+argmax");
    return pt_launch_gemm_argmax(feat, (long long)n * 201, 192, reinterpret_cast<const bf16_t*>(w->d_ptr), reinterpret_cast<const float*>(b->d_ptr),
                                 7680, ids, maxlogit, s);
  }
  p.gemm(feat, rows_cls, 192, "cls", 7680, 0, nullptr, nullptr, 0, nullptr, 0, part);
  if (p.rc != PT_OK) return p.rc;
  PT_HIP_CHECK(hipGetLastError());
  PtProfScope ps(e, s, PT_PROF_OTHER, 0, "argmax");
  return pt_launch_argmax_reduce(part, (long long)n * 201, NT, ids, maxlogit, s);
}

}  // namespace

// gray fp32: layout 0 = chunks [3 n, 32, 300] (what the reference's model receives), 1 = lines [n, 32, 804] (what its
// pre-processor cuts the chunks from: chunk j = columns [252 j, 252 j + 300))
int pt_cvit_forward_net(pt_engine* e, const float* gray, int layout, int n, int32_t* ids, float* maxlogit, hipStream_t s, const int* h_text_w) {
  PT_REQUIRE(e && gray && ids && n > 0 && (layout == 0 || layout == 1), "convnext-vit: bad arguments");
  auto it = e->models.find(PT_MODEL_CONVNEXT_VIT);
  if (it == e->models.end()) {
    pt_set_error("ConvNextViT weights not loaded (pt_weights_load(PT_MODEL_CONVNEXT_VIT))");
    return PT_ERR_STATE;
  }
  if (!pt_model_format_ok(it->second, "PT_MODEL_CONVNEXT_VIT")) return PT_ERR_STATE;
  static int mb = -1, skip = -1;
  if (mb < 0) {
    const char* ev = getenv("PT_CVIT_MICROBATCH");
    mb = ev ? atoi(ev) : 512;
    if (mb < 1) mb = 1;
    ev = getenv("PT_CVIT_SKIP_EMPTY");       // 0: compute the all-padding chunks too (A/B switch)
    skip = ev ? atoi(ev) : 1;
  }
  if (!skip) h_text_w = nullptr;
  const int pitch = layout ? PT_CVIT_W : PT_CVIT_CHUNK_W;
  const long long jstride = layout ? PT_CVIT_CHUNK_STEP : (long long)PT_REC_H * PT_CVIT_CHUNK_W;
  const long long lstride = layout ? (long long)PT_REC_H * PT_CVIT_W : 3ll * PT_REC_H * PT_CVIT_CHUNK_W;
  // micro-batches of about 3 * mb chunks that are really computed (at most 4 * mb lines)
  for (int i0 = 0; i0 < n;) {
    int nb = 0, chunks = 0;
    while (i0 + nb < n && nb < 4 * mb) {
      int c = 3;
      if (h_text_w) c = (h_text_w[i0 + nb] > 0) + (h_text_w[i0 + nb] > PT_CVIT_CHUNK_STEP) + (h_text_w[i0 + nb] > 2 * PT_CVIT_CHUNK_STEP);
      if (nb > 0 && chunks + c > 3 * mb) break;
      chunks += c;
      ++nb;
    }
    const int rc = forward_batch(e, it->second, gray + (long long)i0 * lstride, pitch, jstride, lstride, nb, ids + (size_t)i0 * PT_CVIT_T,
                                 maxlogit ? maxlogit + (size_t)i0 * PT_CVIT_T : nullptr, s, h_text_w ? h_text_w + i0 : nullptr);
    if (rc != PT_OK) return rc;
    i0 += nb;
  }
  return PT_OK;
}

}  // namespace PT_FMT_NS
