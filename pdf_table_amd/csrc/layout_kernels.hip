// layout_kernels.hip -- the non-GEMM kernels of the two mobile-style nets, the PicoDet layout detector and the
// DB-ProxylessNAS text detector (all bandwidth-type work).
//
//   stem3x3s2_kernel   LCNet.conv1: conv 3x3 s2 (3 -> 16) + BN + hardswish (picodet/lcnet.py:165-170), direct VALU
//   dwconv_kernel      depthwise k x k (3 / 5), stride 1 / 2, + BN + optional hardswish: DepthwiseSeparable.dw_conv
//                      (lcnet.py:104-110), DPModule.dwconv (csp_pan.py:76-84), PicoFeat.cls_conv_dw (pico_head.py:98-107)
//   se_gate_kernel     SEModule (lcnet.py:126-153): global average pool -> 1x1 (C/4) + ReLU -> 1x1 + hardsigmoid -> [B, C]
//   se_scale_kernel    x * gate
//   add_kernel         CSPPAN's `top_features = first_top_conv(..) + second_top_conv(..)` (csp_pan.py:338-340)
//   pico_candidates_kernel  anchors whose best class score can pass the post-processor's threshold, with their raw head
//                      outputs (processor_picodet.py:250-262 only ever looks at those)
// DB-ProxylessNAS (db_net/proxyless.py, layers.py, dbnet.py:338-481) re-uses the stem (32 outputs + ReLU), dwconv (PReLU
// / ReLU) and SE kernels (sigmoid gate inside a residual block: x * (1 + gate), hidden width C / squeeze) and adds
//   chan_partial_sum_kernel  deterministic two-level global average pool for SE at 1/4 .. 1/32 resolution
//   dbnas_tail_kernel        everything after the 64->16 pointwise conv of LightSegDetector.binarize in one pass:
//                            ConvT(dw 2x2 s2)+BN+ReLU, 1x1 16->16+BN+ReLU, ConvT(dw 2x2 s2)+BN+ReLU, 1x1 16->1, sigmoid
// NHWC bf16; BF16X3 mode: [hi(C) | lo(C)] per pixel, arithmetic on hi + lo in fp32.
#include "common.h"

namespace PT_FMT_NS {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

namespace {

// 16 stored bits <-> fp32 in the storage format of this namespace (act16.h: bf16, or IEEE half in pt_f16)
__device__ __forceinline__ float bf2f(uint32_t b) { return a16_to_f32(b); }
__device__ __forceinline__ uint32_t f2bf(float f) { return f32_to_a16(f); }
// x * relu6(x + 3) / 6 with the division as a multiplication by the rounded reciprocal (<= 1 ulp from the quotient): the IEEE division was ten
// instructions per value -- a fifth of the depthwise kernels' VALU work
__device__ __forceinline__ float hswish(float v) { return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * 0.16666667f; }

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
  u32x4 o = {hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16)};
  *reinterpret_cast<u32x4*>(p) = o;
  if (split) {
    uint32_t lb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) lb[k] = f2bf(v[k] - bf2f(hb[k]));
    u32x4 l = {lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16), lb[4] | (lb[5] << 16), lb[6] | (lb[7] << 16)};
    *reinterpret_cast<u32x4*>(p + lo_off) = l;
  }
}

// in: NHWC4 [B,H,W,4] ([hi rgb0 | lo rgb0] when split); w fp32 [NOUT][3][3][4]; out [B,Ho,Wo,OST], channels NOUT..OST-1
// zero.  ACT 2: hardswish (LCNet.conv1), 1: ReLU (CompactDetBackbone.first_conv, proxyless.py:101-110)
template <int NOUT, int OST, int ACT>
__global__ __launch_bounds__(256) void stem3x3s2_kernel(const bf16_t* __restrict__ in, const float* __restrict__ w,
                                                         const float* __restrict__ b, bf16_t* __restrict__ out, int B,
                                                         int H, int W, int Ho, int Wo, int split) {
  a16_kernel_enter();
  __shared__ float sw[NOUT * 36];
  __shared__ float sb[NOUT];
  for (int i = threadIdx.x; i < NOUT * 36; i += 256) sw[i] = w[i];
  if (threadIdx.x < NOUT) sb[threadIdx.x] = b[threadIdx.x];
  __syncthreads();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * Ho * Wo) return;
  const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho), bi = (int)(i / ((long long)Wo * Ho));
  const int ps = split ? 8 : 4;
  float px[9][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
      float* d = px[ky * 3 + kx];
      d[0] = d[1] = d[2] = 0.f;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
        const bf16_t* p = in + (((size_t)bi * H + iy) * W + ix) * ps;
#pragma unroll
        for (int c = 0; c < 3; ++c) d[c] = bf2f(p[c]) + (split ? bf2f(p[4 + c]) : 0.f);
      }
    }
  bf16_t* op = out + (size_t)i * (split ? 2 * OST : OST);
#pragma unroll
  for (int n0 = 0; n0 < NOUT; n0 += 8) {
    float o[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) a += px[t][c] * sw[(n0 + n) * 36 + t * 4 + c];
      a += sb[n0 + n];
      o[n] = ACT == 2 ? hswish(a) : fmaxf(a, 0.f);
    }
    store8(op + n0, OST, split, o);
  }
  const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int n0 = NOUT; n0 < OST; n0 += 8) store8(op + n0, OST, split, z);
}

// w fp32 [k*k][C] (BN scale folded), b fp32 [C].  One thread = 8 channels x PX consecutive output pixels of a row: the
// input columns of a tap row are loaded once (16 B each) and shared by the PX outputs, the 8 weights of a tap once per
// thread; per output pixel the taps are still accumulated in (ky, kx) order in fp32.
template <int K, int SX>
__global__ __launch_bounds__(256) void dwconv_kernel(const bf16_t* __restrict__ in, const float* __restrict__ w,
                                                      const float* __restrict__ b, bf16_t* __restrict__ out, int B, int H,
                                                      int W, int C, int sy, int Ho, int Wo, int act, int split,
                                                      const float* __restrict__ slope) {
  // fused multiply-adds for the tap sums (the library is built with -ffp-contract=off for the bit-exact pre-processing
  // arithmetic; here mul + add as two VALU operations was half of the kernel's time)
#pragma clang fp contract(fast)
  a16_kernel_enter();
  constexpr int PX = 4, PAD = K / 2, NCOL = (PX - 1) * SX + K;
  const int cgn = C >> 3, cs = split ? 2 * C : C, wq = (Wo + PX - 1) / PX;
  const float sl = act == 3 ? slope[0] : 0.f;
  const long long total = (long long)B * Ho * wq * cgn;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    // (32-bit index arithmetic: the launcher keeps the item count below 2^31; three 64-bit divisions were ~400 instructions per item)
    const unsigned iu = (unsigned)i;
    const int cg = (int)(iu % (unsigned)cgn);
    unsigned t = iu / (unsigned)cgn;
    const int ox0 = (int)(t % (unsigned)wq) * PX;
    t /= (unsigned)wq;
    const int oy = (int)(t % (unsigned)Ho);
    const int bi = (int)(t / (unsigned)Ho);
    float acc[PX][8];
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[p][q] = 0.f;
    const int ix0 = ox0 * SX - PAD;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy * sy - PAD + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
      float col[NCOL][8];
#pragma unroll
      for (int j = 0; j < NCOL; ++j) {
        const int ix = ix0 + j;
        if ((unsigned)ix < (unsigned)W) {
          load8(in + (((size_t)bi * H + iy) * W + ix) * cs + cg * 8, C, split, col[j]);
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) col[j][q] = 0.f;
        }
      }
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const float4 w0 = *reinterpret_cast<const float4*>(w + (size_t)(ky * K + kx) * C + cg * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(w + (size_t)(ky * K + kx) * C + cg * 8 + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int p = 0; p < PX; ++p)
#pragma unroll
          for (int q = 0; q < 8; ++q) acc[p][q] += col[p * SX + kx][q] * wv[q];
      }
    }
    const float4 b0 = *reinterpret_cast<const float4*>(b + cg * 8), b1 = *reinterpret_cast<const float4*>(b + cg * 8 + 4);
    const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      if (ox0 + p >= Wo) break;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float v = acc[p][q] + bv[q];
        if (act == 2) v = hswish(v);
        else if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 3) v = v > 0.f ? v : sl * v;
        acc[p][q] = v;
      }
      store8(out + (((size_t)bi * Ho + oy) * Wo + ox0 + p) * cs + cg * 8, C, split, acc[p]);
    }
  }
}

// Stride-1 variant with TWO output rows per thread: the K + 1 input rows of the pair are loaded and unpacked once (6 x 8
// column loads for 8 outputs at K = 5 instead of 2 x 5 x 8), every output still accumulates its taps in (ky, kx) order in
// fp32 -- results identical to dwconv_kernel<K, 1>.  The layout net's 5x5 depthwise layers were its largest item
// (4.9 ms per 64-page step: ~1200 VALU instructions and 90 loads per 32 outputs in the one-row kernel).
template <int K>
__global__ __launch_bounds__(256) void dwconv2_kernel(const bf16_t* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ b, bf16_t* __restrict__ out, int B, int H, int W,
                                                       int C, int Ho, int Wo, int act, int split, const float* __restrict__ slope) {
#pragma clang fp contract(fast)
  a16_kernel_enter();
  constexpr int PX = 4, PAD = K / 2, NCOL = PX - 1 + K;
  const int cgn = C >> 3, cs = split ? 2 * C : C, wq = (Wo + PX - 1) / PX, hq = (Ho + 1) >> 1;
  const float sl = act == 3 ? slope[0] : 0.f;
  const long long total = (long long)B * hq * wq * cgn;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const unsigned iu = (unsigned)i;            // (32-bit index arithmetic, as in dwconv_kernel)
    const int cg = (int)(iu % (unsigned)cgn);
    unsigned t = iu / (unsigned)cgn;
    const int ox0 = (int)(t % (unsigned)wq) * PX;
    t /= (unsigned)wq;
    const int oy0 = (int)(t % (unsigned)hq) * 2;
    const int bi = (int)(t / (unsigned)hq);
    float acc[2][PX][8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[r][p][q] = 0.f;
    const int ix0 = ox0 - PAD;
#pragma unroll
    for (int ir = 0; ir < K + 1; ++ir) {          // input row oy0 - PAD + ir feeds output row r with ky = ir - r
      const int iy = oy0 - PAD + ir;
      if ((unsigned)iy >= (unsigned)H) continue;
      float col[NCOL][8];
#pragma unroll
      for (int j = 0; j < NCOL; ++j) {
        const int ix = ix0 + j;
        if ((unsigned)ix < (unsigned)W) {
          load8(in + (((size_t)bi * H + iy) * W + ix) * cs + cg * 8, C, split, col[j]);
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) col[j][q] = 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int ky = ir - r;
        if (ky < 0 || ky >= K) continue;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float4 w0 = *reinterpret_cast<const float4*>(w + (size_t)(ky * K + kx) * C + cg * 8);
          const float4 w1 = *reinterpret_cast<const float4*>(w + (size_t)(ky * K + kx) * C + cg * 8 + 4);
          const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int p = 0; p < PX; ++p)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[r][p][q] += col[p + kx][q] * wv[q];
        }
      }
    }
    const float4 b0 = *reinterpret_cast<const float4*>(b + cg * 8), b1 = *reinterpret_cast<const float4*>(b + cg * 8 + 4);
    const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (oy0 + r >= Ho) break;
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        if (ox0 + p >= Wo) break;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float v = acc[r][p][q] + bv[q];
          if (act == 2) v = hswish(v);
          else if (act == 1) v = fmaxf(v, 0.f);
          else if (act == 3) v = v > 0.f ? v : sl * v;
          acc[r][p][q] = v;
        }
        store8(out + (((size_t)bi * Ho + oy0 + r) * Wo + ox0 + p) * cs + cg * 8, C, split, acc[r][p]);
      }
    }
  }
}

// Stride-1 depthwise convolution on an LDS tile (bf16 mode).  dwconv2_kernel is load-latency-bound (PMC, profiles/r04/experiments.txt: its waves wait 66 % of
// their cycles, the VALU is busy 25 %: 240 registers -> two waves per SIMD, and a thread's 48 loads are issued a row at a time right before they are
// needed).  Here a workgroup stages the (8 + K - 1) x (TW + K - 1) pixel neighbourhood of an 8 x TW output tile x CB channels in LDS -- every thread issues
// its share of the 16-byte loads back to back --, and a thread then reads the K x (4 + K - 1) columns of its 4 output pixels x 8 channels from LDS.
// A third of the registers, 30-40 KB of LDS: four to five workgroups per CU.  Per output the taps are accumulated in (ky, kx) order in fp32 with the
// same fused multiply-adds: the results equal dwconv_kernel's / dwconv2_kernel's bit for bit.
// CB = 64 channels per workgroup (TW = 16) or 32 (TW = 32); w fp32 [K * K][C], b fp32 [C].
template <int K, int CB, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? 128 : 256, 4) void dwconv_tile_kernel(const bf16_t* __restrict__ in, const float* __restrict__ w, const float* __restrict__ b,
                                                          bf16_t* __restrict__ out, int B, int H, int W, int C, int act, const float* __restrict__ slope,
                                                          int tiles_x, int tiles_y) {
#pragma clang fp contract(fast)
  a16_kernel_enter();
  // SPLIT (the pair modes: a pixel is [hi(C) | lo(C)], the value hi + lo): the staged tile holds the fp32 SUMS (256 + 32 bytes per pixel at CB = 64), four
  // output rows per workgroup of 128 threads instead of eight / 256 (the static LDS limit), the same taps in the same (ky, kx) order on the same fp32
  // values as dwconv_kernel / dwconv2_kernel in their split mode: bit-identical results
  constexpr int PAD = K / 2, CGN = CB / 8, TH = SPLIT ? 4 : 8, TW = 16 * 64 / CB, TWIN = TW + K - 1, THIN = TH + K - 1, PX = 4, NCOL = PX + K - 1;
  constexpr int NTHR = SPLIT ? 128 : 256;
  constexpr int PITCH = SPLIT ? CB * 4 + 32 : CB * 2 + 32;             // bytes per staged pixel: the 16 lanes of a ds_read_b128 group then fall on distinct bank quads
  constexpr int NPIECE = THIN * TWIN * CGN;
  static_assert(TH * (TW / PX) * CGN == NTHR, "one thread per (row, 4-pixel column group, 8-channel group)");
  __shared__ __attribute__((aligned(16))) char s_in[THIN * TWIN * PITCH];
  __shared__ __attribute__((aligned(16))) float s_w[K * K * CB + CB];
  const int tid = threadIdx.x;
  int L = blockIdx.x;
  const int cb = L % (C / CB);
  L /= (C / CB);
  const int txi = L % tiles_x;
  L /= tiles_x;
  const int tyi = L % tiles_y;
  const int bi = L / tiles_y;
  const int oy0 = tyi * TH, ox0 = txi * TW, c0 = cb * CB;
  const int cs = SPLIT ? 2 * C : C;               // channels per pixel in memory
  const bf16_t* in_b = in + (size_t)bi * H * W * cs + c0;
#pragma unroll
  for (int j = 0; j < (NPIECE + NTHR - 1) / NTHR; ++j) {
    const int idx = tid + j * NTHR;
    if (idx < NPIECE) {
      const int pix = idx / CGN, c = idx - pix * CGN;
      const int iy = pix / TWIN, ix = pix - iy * TWIN;
      const int gy = oy0 - PAD + iy, gx = ox0 - PAD + ix;
      const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      if (SPLIT) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (inside) load8(in_b + ((size_t)gy * W + gx) * cs + c * 8, C, 1, v);
        float4* d = reinterpret_cast<float4*>(s_in + pix * PITCH + c * 32);
        d[0] = make_float4(v[0], v[1], v[2], v[3]);
        d[1] = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (inside) v = *reinterpret_cast<const u32x4*>(in_b + ((size_t)gy * W + gx) * cs + c * 8);
        *reinterpret_cast<u32x4*>(s_in + pix * PITCH + c * 16) = v;
      }
    }
  }
  for (int i = tid; i < K * K * CB; i += NTHR) s_w[i] = w[(size_t)(i / CB) * C + c0 + (i % CB)];
  if (tid < CB) s_w[K * K * CB + tid] = b[c0 + tid];
  __syncthreads();
  const int cg = tid % CGN, g = tid / CGN;
  const int colg = g % (TW / PX), row = g / (TW / PX);
  float acc[PX][8];
#pragma unroll
  for (int p = 0; p < PX; ++p)
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[p][q] = 0.f;
#pragma unroll 1      // (one tap row at a time: unrolled, hipcc hoists every row's LDS reads and needs 256 registers)
  for (int ky = 0; ky < K; ++ky) {
    float col[NCOL][8];
#pragma unroll
    for (int j = 0; j < NCOL; ++j) {
      const char* sp = s_in + ((row + ky) * TWIN + colg * PX + j) * PITCH;
      if (SPLIT) {
        const float4 f0 = *reinterpret_cast<const float4*>(sp + cg * 32), f1 = *reinterpret_cast<const float4*>(sp + cg * 32 + 16);
        col[j][0] = f0.x; col[j][1] = f0.y; col[j][2] = f0.z; col[j][3] = f0.w;
        col[j][4] = f1.x; col[j][5] = f1.y; col[j][6] = f1.z; col[j][7] = f1.w;
      } else {
        const u32x4 h = *reinterpret_cast<const u32x4*>(sp + cg * 16);
        const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) col[j][k] = bf2f((k & 1) ? (hw[k >> 1] >> 16) : (hw[k >> 1] & 0xFFFFu));
      }
    }
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const float4 w0 = *reinterpret_cast<const float4*>(s_w + (ky * K + kx) * CB + cg * 8);
      const float4 w1 = *reinterpret_cast<const float4*>(s_w + (ky * K + kx) * CB + cg * 8 + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[p][q] += col[p + kx][q] * wv[q];
    }
  }
  const int oy = oy0 + row;
  if (oy >= H) return;
  const float sl = act == 3 ? slope[0] : 0.f;
  const float* bv = s_w + K * K * CB + cg * 8;
#pragma unroll
  for (int p = 0; p < PX; ++p) {
    const int ox = ox0 + colg * PX + p;
    if (ox >= W) break;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = acc[p][q] + bv[q];
      if (act == 2) v = hswish(v);
      else if (act == 1) v = fmaxf(v, 0.f);
      else if (act == 3) v = v > 0.f ? v : sl * v;
      acc[p][q] = v;
    }
    store8(out + (((size_t)bi * H + oy) * W + ox) * cs + c0 + cg * 8, C, SPLIT ? 1 : 0, acc[p]);
  }
}

// global-average-pool partial sums, deterministic: block (chunk, image) sums its pixel range per channel
// part: fp32 [B][gridDim.x][C]
__global__ __launch_bounds__(256) void chan_partial_sum_kernel(const bf16_t* __restrict__ x, int HW, int C, int split,
                                                                float* __restrict__ part) {
  a16_kernel_enter();
  __shared__ float s_acc[256][8];
  const int cgn = C >> 3, cs = split ? 2 * C : C, nslot = 256 / cgn;
  const int tid = threadIdx.x, cg = tid % cgn, slot = tid / cgn;
  const int bi = blockIdx.y, nchunk = gridDim.x;
  const int per = (HW + nchunk - 1) / nchunk, p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (slot < nslot)
    for (int p = p0 + slot; p < p1; p += nslot) {
      float v[8];
      load8(x + ((size_t)bi * HW + p) * cs + cg * 8, C, split, v);
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] += v[q];
    }
#pragma unroll
  for (int q = 0; q < 8; ++q) s_acc[tid][q] = a[q];
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float t = 0.f;
    for (int sl = 0; sl < nslot; ++sl) t += s_acc[sl * cgn + (c >> 3)][c & 7];
    part[((size_t)bi * nchunk + blockIdx.x) * C + c] = t;
  }
}

// one workgroup per image: gate[b][c] = G(W2 relu(W1 mean_hw(x) + b1) + b2); C <= 512, hidden width Ch <= 128.
// mode 0: G = hardsigmoid (LCNet SEModule); mode 1: G = 1 + sigmoid (SELayer inside an identity-shortcut block:
// x + x * sigmoid(..), db_net/layers.py:480-489 + :50-57).  part != null: mean from nchunk partial sums.
__global__ __launch_bounds__(256) void se_gate_kernel(const bf16_t* __restrict__ x, int HW, int C, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, float* __restrict__ gate, int split,
                                                       int Ch, int mode, const float* __restrict__ part, int nchunk) {
  a16_kernel_enter();
  __shared__ float s_mean[512];
  __shared__ float s_hid[128];
  const int bi = blockIdx.x, tid = threadIdx.x, cs = split ? 2 * C : C;
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    if (part) {
      for (int i = 0; i < nchunk; ++i) s += part[((size_t)bi * nchunk + i) * C + c];
    } else {
      const bf16_t* p = x + (size_t)bi * HW * cs + c;
      for (int i = 0; i < HW; ++i) s += bf2f(p[(size_t)i * cs]) + (split ? bf2f(p[(size_t)i * cs + C]) : 0.f);
    }
    s_mean[c] = s / (float)HW;
  }
  __syncthreads();
  for (int h = tid; h < Ch; h += 256) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += w1[(size_t)h * C + c] * s_mean[c];
    s_hid[h] = fmaxf(a + b1[h], 0.f);
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f;
    for (int h = 0; h < Ch; ++h) a += w2[(size_t)c * Ch + h] * s_hid[h];
    a += b2[c];
    gate[(size_t)bi * C + c] = mode == 0 ? fminf(fmaxf(a + 3.f, 0.f), 6.f) / 6.f      // nn.Hardsigmoid
                                         : 1.f + 1.f / (1.f + expf(-a));
  }
}

__global__ __launch_bounds__(256) void se_scale_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gate,
                                                        bf16_t* __restrict__ out, int B, int HW, int C, int split) {
  a16_kernel_enter();
  const int cgn = C >> 3, cs = split ? 2 * C : C;
  const long long total = (long long)B * HW * cgn;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cg = (int)(i % cgn);
  const long long pix = i / cgn;
  const int bi = (int)(pix / HW);
  float v[8];
  load8(x + (size_t)pix * cs + cg * 8, C, split, v);
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] *= gate[(size_t)bi * C + cg * 8 + q];
  store8(out + (size_t)pix * cs + cg * 8, C, split, v);
}

__global__ __launch_bounds__(256) void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                   bf16_t* __restrict__ out, long long npix, int C, int split) {
  a16_kernel_enter();
  const int cgn = C >> 3, cs = split ? 2 * C : C;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix * cgn) return;
  const int cg = (int)(i % cgn);
  const long long pix = i / cgn;
  float va[8], vb[8];
  load8(a + (size_t)pix * cs + cg * 8, C, split, va);
  load8(b + (size_t)pix * cs + cg * 8, C, split, vb);
#pragma unroll
  for (int q = 0; q < 8; ++q) va[q] += vb[q];
  store8(out + (size_t)pix * cs + cg * 8, C, split, va);
}

// head: fp32 [B][A][40] of one level (ncls class logits, then 4 x (reg_max + 1) box logits).  An anchor whose best
// sigmoid score exceeds thr_lo is appended to cands[b] as (level, anchor, 40 raw values) -- 48 floats per record.
__global__ __launch_bounds__(256) void pico_candidates_kernel(const float* __restrict__ head, int B, int A, int ncls,
                                                               int level, float thr_lo, int max_cands,
                                                               float* __restrict__ cands, int* __restrict__ counts) {
  a16_kernel_enter();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * A) return;
  const int bi = (int)(i / A), a = (int)(i % A);
  const float* h = head + (size_t)i * 40;
  float mx = -INFINITY;
  for (int c = 0; c < ncls; ++c) mx = fmaxf(mx, h[c]);
  if (!(1.f / (1.f + expf(-mx)) > thr_lo)) return;
  const int slot = atomicAdd(&counts[bi], 1);
  if (slot >= max_cands) return;
  float* o = cands + ((size_t)bi * max_cands + slot) * 48;
  o[0] = __int_as_float(level);
  o[1] = __int_as_float(a);
  for (int c = 0; c < 40; ++c) o[2 + c] = h[c];
}

// LightSegDetector.binarize after its first pointwise conv (dbnet.py:383-386 with DwPwConvTranspose :75-99), one thread
// per 1/4-resolution pixel.  y: bf16 [B,H4,W4,16] (ReLU already applied); tw fp32, BN folded:
//   [0,64) W1[q][c]  [64,80) B1[c]  [80,336) P1[j][c]  [336,352) pb1[j]  [352,416) W2[r][c]  [416,432) B2[c]
//   [432,448) P2[c]  [448] pb2          (q, r = 2*dy + dx sub-pixel of the two transposed convs)
// prob / logits: fp32 [B, 4*H4, 4*W4]
__global__ __launch_bounds__(256) void dbnas_tail_kernel(const bf16_t* __restrict__ y, const float* __restrict__ tw, int B,
                                                          int H4, int W4, int split, float* __restrict__ prob,
                                                          float* __restrict__ logits) {
  a16_kernel_enter();
  __shared__ float sw[449];
  for (int i = threadIdx.x; i < 449; i += 256) sw[i] = tw[i];
  __syncthreads();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * H4 * W4) return;
  const int x4 = (int)(i % W4), y4 = (int)((i / W4) % H4), bi = (int)(i / ((long long)W4 * H4));
  float v[16];
  const bf16_t* yp = y + (size_t)i * (split ? 32 : 16);
  load8(yp, 16, split, v);
  load8(yp + 8, 16, split, v + 8);
  const int OH = 4 * H4, OW = 4 * W4;
  float res[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float a[16], u[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = fmaxf(v[c] * sw[q * 16 + c] + sw[64 + c], 0.f);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) t += sw[80 + j * 16 + c] * a[c];
      u[j] = fmaxf(t + sw[336 + j], 0.f);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) t += sw[432 + c] * fmaxf(u[c] * sw[352 + r * 16 + c] + sw[416 + c], 0.f);
      res[2 * (q >> 1) + (r >> 1)][2 * (q & 1) + (r & 1)] = t + sw[448];
    }
  }
#pragma unroll
  for (int ry = 0; ry < 4; ++ry) {
    const size_t o = ((size_t)bi * OH + 4 * y4 + ry) * OW + 4 * x4;
    if (logits) *reinterpret_cast<float4*>(logits + o) = make_float4(res[ry][0], res[ry][1], res[ry][2], res[ry][3]);
    if (prob)
      *reinterpret_cast<float4*>(prob + o) = make_float4(1.f / (1.f + expf(-res[ry][0])), 1.f / (1.f + expf(-res[ry][1])),
                                                         1.f / (1.f + expf(-res[ry][2])), 1.f / (1.f + expf(-res[ry][3])));
  }
}

// global average pool, second level: part [B][nchunk][C] -> mean bf16 [rows >= B][C] ([hi | lo] when split); rows B..rows-1
// are written as zeros (the classifier GEMM works on whole 32-row tiles)
__global__ __launch_bounds__(256) void chan_mean_kernel(const float* __restrict__ part, int B, int rows, int nchunk, int C,
                                                        int HW, int split, bf16_t* __restrict__ mean) {
  a16_kernel_enter();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)rows * C) return;
  const int r = (int)(i / C), c = (int)(i % C);
  float s = 0.f;
  if (r < B)
    for (int k = 0; k < nchunk; ++k) s += part[((size_t)r * nchunk + k) * C + c];
  s /= (float)HW;
  const uint32_t hb = f2bf(s);
  const int cs = split ? 2 * C : C;
  mean[(size_t)r * cs + c] = (bf16_t)hb;
  if (split) mean[(size_t)r * cs + C + c] = (bf16_t)f2bf(s - bf2f(hb));
}

inline int blocks_for(long long total, int cap = 256 * 64) {
  long long b = (total + 255) / 256;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

int pt_launch_stem3x3s2(const bf16_t* in, const float* w, const float* b, bf16_t* out, int B, int H, int W, int split,
                        hipStream_t s, int variant) {
  PT_REQUIRE(in && w && b && out, "stem3x3: null pointer");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const dim3 grid((unsigned)(((long long)B * Ho * Wo + 255) / 256));
  if (variant == 0)       // LCNet: 16 outputs stored 32 wide, hardswish
    hipLaunchKernelGGL((stem3x3s2_kernel<16, 32, 2>), grid, dim3(256), 0, s, in, w, b, out, B, H, W, Ho, Wo, split);
  else                    // ProxylessNAS: 32 outputs stored 64 wide, ReLU
    hipLaunchKernelGGL((stem3x3s2_kernel<32, 64, 1>), grid, dim3(256), 0, s, in, w, b, out, B, H, W, Ho, Wo, split);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// stride: s (both directions) or, for PP-LCNet's text-line classifiers, (sy << 8) | sx (cls_pp_lcnet.py:190-191)
int pt_launch_dwconv(const bf16_t* in, const float* w, const float* b, bf16_t* out, int B, int H, int W, int C, int k,
                     int stride, int act, int split, hipStream_t s, const float* slope) {
  const int sy = stride > 255 ? stride >> 8 : stride, sx = stride > 255 ? stride & 255 : stride;
  PT_REQUIRE(in && w && b && out && C % 8 == 0 && (k == 3 || k == 5) && (sy == 1 || sy == 2) && (sx == 1 || sx == 2),
             "dwconv: bad arguments");
  PT_REQUIRE(act != 3 || slope, "dwconv: PReLU needs the slope tensor");
  const int pad = k / 2, Ho = (H + 2 * pad - k) / sy + 1, Wo = (W + 2 * pad - k) / sx + 1;
  PT_REQUIRE((long long)B * Ho * ((Wo + 3) / 4) * (C / 8) < (1ll << 31), "dwconv: more than 2^31 work items (the kernels index in 32 bits)");
  const dim3 grid(blocks_for((long long)B * Ho * ((Wo + 3) / 4) * (C / 8)));
  static int two = -1;           // PT_DWCONV2=0: one output row per thread everywhere (A/B switch)
  if (two < 0) {
    const char* ev = getenv("PT_DWCONV2");
    two = ev ? atoi(ev) : 1;
  }
  const char* tile_ev = getenv("PT_DWCONV_TILE");      // PT_DWCONV_TILE=0: the register kernels everywhere (A/B switch, read per call)
  const bool tile = !(tile_ev && tile_ev[0] == '0');
  if (tile && sy == 1 && sx == 1 && C % 32 == 0) {      // stride 1, pad k / 2: Ho == H, Wo == W; the pair modes take the fp32-staged instantiation
    const int cb = C % 64 == 0 ? 64 : 32, tw = cb == 64 ? 16 : 32, th = split ? 4 : 8;
    const int tiles_x = (Wo + tw - 1) / tw, tiles_y = (Ho + th - 1) / th;
    const long long nblk = (long long)B * tiles_x * tiles_y * (C / cb);
    PT_REQUIRE(nblk > 0 && nblk < (1ll << 31), "dwconv: grid out of range");
#define PT_DWT(KK, CC, SP) hipLaunchKernelGGL((dwconv_tile_kernel<KK, CC, SP>), dim3((unsigned)nblk), dim3(SP ? 128 : 256), 0, s, in, w, b, out, B, H, W, C, act, slope, tiles_x, tiles_y)
    if (split) {
      if (k == 3 && cb == 64) PT_DWT(3, 64, true);
      else if (k == 3) PT_DWT(3, 32, true);
      else if (cb == 64) PT_DWT(5, 64, true);
      else PT_DWT(5, 32, true);
    } else {
      if (k == 3 && cb == 64) PT_DWT(3, 64, false);
      else if (k == 3) PT_DWT(3, 32, false);
      else if (cb == 64) PT_DWT(5, 64, false);
      else PT_DWT(5, 32, false);
    }
#undef PT_DWT
    PT_HIP_CHECK(hipGetLastError());
    return PT_OK;
  }
  if (two && sy == 1 && sx == 1 && Ho >= 2) {
    const dim3 grid2(blocks_for((long long)B * ((Ho + 1) / 2) * ((Wo + 3) / 4) * (C / 8)));
    if (k == 3)
      hipLaunchKernelGGL((dwconv2_kernel<3>), grid2, dim3(256), 0, s, in, w, b, out, B, H, W, C, Ho, Wo, act, split, slope);
    else
      hipLaunchKernelGGL((dwconv2_kernel<5>), grid2, dim3(256), 0, s, in, w, b, out, B, H, W, C, Ho, Wo, act, split, slope);
    PT_HIP_CHECK(hipGetLastError());
    return PT_OK;
  }
#define PT_DW(KK, SS) hipLaunchKernelGGL((dwconv_kernel<KK, SS>), grid, dim3(256), 0, s, in, w, b, out, B, H, W, C, sy, Ho, Wo, act, split, slope)
  if (k == 3 && sx == 1) PT_DW(3, 1);
  else if (k == 3) PT_DW(3, 2);
  else if (sx == 1) PT_DW(5, 1);
  else PT_DW(5, 2);
#undef PT_DW
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// hidden: width of the squeeze layer; mode 0 hardsigmoid gate, 1: 1 + sigmoid (see se_gate_kernel); part: scratch of
// PT_SE_CHUNKS * B * C floats for the two-level average pool, or null for the single-workgroup scan (tiny maps)
int pt_launch_se(const bf16_t* x, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                 bf16_t* out, int B, int HW, int C, int split, hipStream_t s, int hidden, int mode, float* part) {
  PT_REQUIRE(x && w1 && b1 && w2 && b2 && gate && out && C <= 512 && C % 32 == 0 && hidden > 0 && hidden <= 128,
             "SE: bad arguments");
  if (part)
    hipLaunchKernelGGL(chan_partial_sum_kernel, dim3(PT_SE_CHUNKS, B), dim3(256), 0, s, x, HW, C, split, part);
  hipLaunchKernelGGL(se_gate_kernel, dim3(B), dim3(256), 0, s, x, HW, C, w1, b1, w2, b2, gate, split, hidden, mode, part,
                     PT_SE_CHUNKS);
  hipLaunchKernelGGL(se_scale_kernel, dim3((unsigned)(((long long)B * HW * (C / 8) + 255) / 256)), dim3(256), 0, s, x, gate, out, B,
                     HW, C, split);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// x [B, HW, C] -> mean over HW as bf16 [rows, C]; part: scratch of PT_SE_CHUNKS * B * C floats
int pt_launch_chan_mean(const bf16_t* x, int B, int HW, int C, int split, float* part, bf16_t* mean, int rows, hipStream_t s) {
  PT_REQUIRE(x && part && mean && C % 8 == 0 && C <= 2048 && rows >= B, "channel mean: bad arguments");
  hipLaunchKernelGGL(chan_partial_sum_kernel, dim3(PT_SE_CHUNKS, B), dim3(256), 0, s, x, HW, C, split, part);
  hipLaunchKernelGGL(chan_mean_kernel, dim3((unsigned)(((long long)rows * C + 255) / 256)), dim3(256), 0, s, part, B, rows,
                     PT_SE_CHUNKS, C, HW, split, mean);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_launch_dbnas_tail(const bf16_t* y, const float* tw, int B, int H4, int W4, int split, float* prob, float* logits,
                         hipStream_t s) {
  PT_REQUIRE(y && tw && (prob || logits), "dbnas tail: null pointer");
  hipLaunchKernelGGL(dbnas_tail_kernel, dim3((unsigned)(((long long)B * H4 * W4 + 255) / 256)), dim3(256), 0, s, y, tw, B, H4, W4,
                     split, prob, logits);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_launch_add(const bf16_t* a, const bf16_t* b, bf16_t* out, long long npix, int C, int split, hipStream_t s) {
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)((npix * (C / 8) + 255) / 256)), dim3(256), 0, s, a, b, out, npix, C, split);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_launch_pico_candidates(const float* head, int B, int A, int ncls, int level, float thr_lo, int max_cands, float* cands,
                              int* counts, hipStream_t s) {
  hipLaunchKernelGGL(pico_candidates_kernel, dim3((unsigned)(((long long)B * A + 255) / 256)), dim3(256), 0, s, head, B, A, ncls,
                     level, thr_lo, max_cands, cands, counts);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

}  // namespace PT_FMT_NS
