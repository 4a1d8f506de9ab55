// graph_ops.hip -- single-operator entry points for the generic ONNX layer-list executor (pdf_table_amd/onnx_exec.py;
// the reference runs its default models as arbitrary ONNX graphs through onnxruntime,
// utils/deploy_utils.py:243-280).  The executor walks the importer's engine layer list and issues one call per layer:
// the convolutions go to pt_op_conv2d (MFMA implicit GEMM, conv_igemm.hip), everything else to the entry points below --
// thin wrappers around the launchers the dedicated launch graphs use (depthwise conv, pooling, channel mean, add), plus two
// element-wise kernels the dedicated graphs fuse into epilogues (channel scale of an SE block, a stand-alone activation).
// bf16 NHWC activations, channel counts padded to multiples of 8 by the caller (zero padding stays zero through every op
// here except sigmoid / hardsigmoid, whose padded lanes the caller never reads).
#include "common.h"

namespace PT_FMT_NS {

namespace {

// 16 stored bits <-> fp32 in the storage format of this namespace (act16.h: bf16, or IEEE half in pt_f16)
__device__ __forceinline__ float g_bf(uint32_t bits16) { return a16_to_f32(bits16); }
__device__ __forceinline__ uint32_t g_f2bf(float f) { return f32_to_a16(f); }

// (hi | lo) tensors of the executor's tolerance mode (PT_PRECISION_BF16X3 convention: a pixel / row holds [hi(C) | lo(C)], value = hi + lo,
// arithmetic in fp32, result split again): lo = 0 reads / writes a plain bf16 tensor
__device__ __forceinline__ void g_load8(const bf16_t* p, int lo, float* v) {
  const uint4 h = *reinterpret_cast<const uint4*>(p);
  const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[2 * k] = g_bf(hw[k] & 0xFFFFu); v[2 * k + 1] = g_bf(hw[k] >> 16); }
  if (lo) {
    const uint4 l = *reinterpret_cast<const uint4*>(p + lo);
    const uint32_t lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] += g_bf(lw[k] & 0xFFFFu); v[2 * k + 1] += g_bf(lw[k] >> 16); }
  }
}
__device__ __forceinline__ void g_store8(bf16_t* p, int lo, const float* v) {
  uint32_t h[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) h[k] = g_f2bf(v[k]);
  *reinterpret_cast<uint4*>(p) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
  if (lo) {
    uint32_t l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) l[k] = g_f2bf(v[k] - g_bf(h[k]));
    *reinterpret_cast<uint4*>(p + lo) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
  }
}
__device__ __forceinline__ float g_ld(const bf16_t* p, int lo) { return lo ? g_bf(p[0]) + g_bf(p[lo]) : g_bf(p[0]); }
__device__ __forceinline__ void g_st(bf16_t* p, int lo, float v) {
  const uint32_t h = g_f2bf(v);
  p[0] = (bf16_t)h;
  if (lo) p[lo] = (bf16_t)g_f2bf(v - g_bf(h));
}

// x [B, HW, C] *= gate [B, C] (the Mul of a squeeze-and-excitation block)
__global__ __launch_bounds__(256) void scale_channels_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gate,
                                                             bf16_t* __restrict__ out, long long total8, int HW, int C, int split) {
  a16_kernel_enter();
  const int cg = C >> 3, lo = split ? C : 0, cs = split ? 2 * C : C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cg);
    const long long pix = i / cg;
    const int b = (int)(pix / HW);
    float xv[8], gv[8];
    g_load8(x + pix * cs + c8 * 8, lo, xv);
    g_load8(gate + (size_t)b * cs + c8 * 8, lo, gv);
#pragma unroll
    for (int k = 0; k < 8; ++k) xv[k] *= gv[k];
    g_store8(out + pix * cs + c8 * 8, lo, xv);
  }
}

// kind: 1 relu, 2 hardswish, 4 sigmoid, 5 hardsigmoid (max(0, min(1, alpha x + beta))), 6 relu6, 7 GELU (erf), 8 swish
__global__ __launch_bounds__(256) void act_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, long long total8, int kind,
                                                  float alpha, float beta, int C, int split) {
  a16_kernel_enter();
  const int cg = C >> 3, lo = split ? C : 0, cs = split ? 2 * C : C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const long long off = (i / cg) * cs + (i % cg) * 8;
    float v[8];
    g_load8(x + off, lo, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = v[j];
      if (kind == 1) t = fmaxf(t, 0.f);
      else if (kind == 2) t = t * fminf(fmaxf(t + 3.f, 0.f), 6.f) / 6.f;
      else if (kind == 4) t = 1.f / (1.f + expf(-t));
      else if (kind == 5) t = fmaxf(0.f, fminf(1.f, alpha * t + beta));
      else if (kind == 6) t = fminf(fmaxf(t, 0.f), 6.f);
      else if (kind == 7) t = t * 0.5f * (1.f + erff(t * 0.70710678118654752f));      // GELU (erf form: nn.GELU())
      else if (kind == 8) t = t / (1.f + expf(-t));                                      // swish / SiLU: x * sigmoid(x)
      v[j] = t;
    }
    g_store8(out + off, lo, v);
  }
}

// AveragePool k x k, stride k, no padding: out [B, H/k, W/k, C]
__global__ __launch_bounds__(256) void avgpool_kxk_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int H, int W, int C,
                                                          int k, int split) {
  a16_kernel_enter();
  const int Ho = H / k, Wo = W / k, cg = C >> 3, lo = split ? C : 0, cs = split ? 2 * C : C;
  const long long total = (long long)B * Ho * Wo * cg;
  const float inv = 1.f / (float)(k * k);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cg);
    long long t = i / cg;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho), b = (int)(t / Ho);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        float v[8];
        g_load8(x + (((size_t)b * H + oy * k + dy) * W + ox * k + dx) * cs + c8 * 8, lo, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += v[q];
      }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] *= inv;
    g_store8(out + (i / cg) * cs + c8 * 8, lo, acc);
  }
}


// ---- sequence operators (SVTR-type recognisers: LayerNorm / attention / soft-max blocks; rows = tokens, channels padded like every tensor here) ----
// out[pix][dst_off + c] = src[pix][src_off + c], c < n: channel concat / slice without arithmetic
__global__ __launch_bounds__(256) void copy_channels_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, long long npix, int scs, int soff,
                                                            int dcs, int doff, int n) {
  a16_kernel_enter();
  const long long total = npix * n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long pix = i / n;
    const int c = (int)(i - pix * n);
    dst[pix * dcs + doff + c] = src[pix * scs + soff + c];
  }
}

// nearest-neighbour up-sampling by an integer factor: out [B, H f, W f, C]
__global__ __launch_bounds__(256) void upsample_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int H, int W, int C, int f) {
  a16_kernel_enter();
  const int cg = C >> 3, Wo = W * f, Ho = H * f;
  const long long total = (long long)B * Ho * Wo * cg;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cg);
    long long t = i / cg;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho), b = (int)(t / Ho);
    *reinterpret_cast<uint4*>(out + i * 8) = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + oy / f) * W + ox / f) * C + c8 * 8);
  }
}

__global__ __launch_bounds__(256) void mul_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ out, long long total8,
                                                  int C, int split) {
  a16_kernel_enter();
  const int cg = C >> 3, lo = split ? C : 0, cs = split ? 2 * C : C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const long long off = (i / cg) * cs + (i % cg) * 8;
    float va[8], vb[8];
    g_load8(a + off, lo, va);
    g_load8(b + off, lo, vb);
#pragma unroll
    for (int k = 0; k < 8; ++k) va[k] *= vb[k];
    g_store8(out + off, lo, va);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// LayerNorm over the first C of Cp channels of every row (biased variance, eps inside the root: nn.LayerNorm); one wave per row, padded
// channels are written as zeros.  split: rows are [hi(Cp) | lo(Cp)]
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, long long rows, int Cp, int C,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int split) {
  a16_kernel_enter();
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, lo = split ? Cp : 0, cs = split ? 2 * Cp : Cp;
  if (row >= rows) return;
  const bf16_t* xr = x + row * cs;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += g_ld(xr + c, lo);
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = g_ld(xr + c, lo) - mean; q += d * d; }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
  bf16_t* orow = out + row * cs;
  for (int c = lane; c < Cp; c += 64) g_st(orow + c, lo, c < C ? (g_ld(xr + c, lo) - mean) * rstd * gamma[c] + beta[c] : 0.f);
}

// soft-max over the first C of Cp channels of every row -> fp32 probabilities [rows][C] (network outputs: CTC heads) or bf16 [rows][Cp] ([hi | lo] when split)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const bf16_t* __restrict__ x, long long rows, int Cp, int C, float* __restrict__ out_f32,
                                                           bf16_t* __restrict__ out_bf, int split) {
  a16_kernel_enter();
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, lo = split ? Cp : 0, cs = split ? 2 * Cp : Cp;
  if (row >= rows) return;
  const bf16_t* xr = x + row * cs;
  float m = -3.0e38f;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, g_ld(xr + c, lo));
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(g_ld(xr + c, lo) - m);
  const float inv = 1.f / wave_sum(s);
  for (int c = lane; c < (out_f32 ? C : Cp); c += 64) {
    const float p = c < C ? expf(g_ld(xr + c, lo) - m) * inv : 0.f;
    if (out_f32) out_f32[row * C + c] = p;
    else g_st(out_bf + row * cs + c, lo, p);
  }
}

// Multi-head self-attention of token rows holding [q | k | v] (each heads * d channels; channel = part * heads * d + head * d + j): one wave per
// (batch, head, query); scores of all T keys in LDS (T <= 1024), soft-max, weighted sum of the values; out[b, t, head * d + j].  d <= 64.
// split: rows are [hi(qcs) | lo(qcs)] / [hi(ocs) | lo(ocs)]; scores, soft-max and the weighted sum in fp32 on hi + lo
__global__ __launch_bounds__(64) void attention_rows_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int T, int heads, int d, int qcs,
                                                            int ocs, float scale, int split) {
  a16_kernel_enter();
  __shared__ float p[1024];
  __shared__ float qs[64];
  const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  const int C = heads * d, lo = split ? qcs : 0, rs = split ? 2 * qcs : qcs, olo = split ? ocs : 0, ors = split ? 2 * ocs : ocs;
  const bf16_t* base = qkv + (size_t)b * T * rs;
  if (lane < d) qs[lane] = g_ld(base + (size_t)t * rs + h * d + lane, lo) * scale;
  __syncthreads();
  float m = -3.0e38f;
  for (int k = lane; k < T; k += 64) {
    const bf16_t* kr = base + (size_t)k * rs + C + h * d;
    float acc = 0.f;
    for (int j = 0; j < d; ++j) acc += qs[j] * g_ld(kr + j, lo);
    p[k] = acc;
    m = fmaxf(m, acc);
  }
  m = wave_max(m);
  float s = 0.f;
  for (int k = lane; k < T; k += 64) { const float e_ = expf(p[k] - m); p[k] = e_; s += e_; }
  const float inv = 1.f / wave_sum(s);
  __syncthreads();
  if (lane < d) {
    float acc = 0.f;
    for (int k = 0; k < T; ++k) acc += p[k] * g_ld(base + (size_t)k * rs + 2 * C + h * d + lane, lo);
    g_st(out + ((size_t)b * T + t) * ors + h * d + lane, olo, acc * inv);
  }
}

// byte copy by the compute units; either pointer may be pinned host memory (mapped into the device's address space)
__global__ __launch_bounds__(256) void copy_bytes_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long n16, const unsigned char* __restrict__ src_tail,
                                                         unsigned char* __restrict__ dst_tail, int tail) {
  a16_kernel_enter();
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

inline unsigned grid_for(long long n) {
  const long long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace

namespace api {

int pt_op_dwconv(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int C, const float* d_w_taps, const float* d_bias, int k,
                 int stride, int act, uint16_t* d_out, int split, pt_stream stream) {
  PT_REQUIRE(e && d_in && d_w_taps && d_bias && d_out, "pt_op_dwconv: null pointer");
  PT_REQUIRE((k == 3 || k == 5) && (stride == 1 || stride == 2) && C % 8 == 0 && act >= 0 && act <= 2,
             "pt_op_dwconv: k=%d stride=%d C=%d act=%d unsupported (k 3/5, stride 1/2, C multiple of 8, act 0/1/2)", k, stride, C, act);
  return pt_launch_dwconv(d_in, d_w_taps, d_bias, d_out, B, H, W, C, k, stride, act, split ? 1 : 0, reinterpret_cast<hipStream_t>(stream), nullptr);
}

int pt_op_add(pt_engine* e, const uint16_t* d_a, const uint16_t* d_b, uint16_t* d_out, long long npix, int C, int split, pt_stream stream) {
  PT_REQUIRE(e && d_a && d_b && d_out && npix > 0 && C % 8 == 0, "pt_op_add: bad arguments");
  return pt_launch_add(d_a, d_b, d_out, npix, C, split ? 1 : 0, reinterpret_cast<hipStream_t>(stream));
}

int pt_op_maxpool(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int C, int k, int stride, int pad, uint16_t* d_out, int split,
                  pt_stream stream) {
  PT_REQUIRE(e && d_in && d_out && C % 8 == 0, "pt_op_maxpool: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (k == 3 && stride == 2 && pad == 1) return pt_launch_maxpool3x3s2(d_in, B, H, W, C, d_out, split ? 1 : 0, s);
  PT_REQUIRE(k == stride && pad == 0 && k >= 2 && H % k == 0 && W % k == 0,
             "pt_op_maxpool: only MaxPool(3, 2, 1) and non-overlapping k x k pools on sizes divisible by k (got k=%d stride=%d pad=%d)", k,
             stride, pad);
  return pt_launch_maxpool_kxk(d_in, B, H, W, C, k, k, 0, split ? 1 : 0, d_out, s);
}

int pt_op_avgpool(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int C, int k, uint16_t* d_out, int split, pt_stream stream) {
  PT_REQUIRE(e && d_in && d_out && C % 8 == 0 && k >= 2 && H % k == 0 && W % k == 0, "pt_op_avgpool: k x k / stride k on sizes divisible by k");
  hipLaunchKernelGGL(avgpool_kxk_kernel, dim3(grid_for((long long)B * (H / k) * (W / k) * (C / 8))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), d_in, d_out, B, H, W, C, k, split);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_op_chan_mean(pt_engine* e, const uint16_t* d_in, int B, int HW, int C, float* d_scratch, uint16_t* d_mean, int split, pt_stream stream) {
  PT_REQUIRE(e && d_in && d_scratch && d_mean, "pt_op_chan_mean: null pointer");
  return pt_launch_chan_mean(d_in, B, HW, C, split ? 1 : 0, d_scratch, d_mean, B, reinterpret_cast<hipStream_t>(stream));
}

int pt_op_chan_mean_scratch_floats(int B, int C) { return PT_SE_CHUNKS * B * C; }

int pt_op_scale_channels(pt_engine* e, const uint16_t* d_in, const uint16_t* d_gate, int B, int HW, int C, uint16_t* d_out, int split,
                         pt_stream stream) {
  PT_REQUIRE(e && d_in && d_gate && d_out && C % 8 == 0 && B > 0 && HW > 0, "pt_op_scale_channels: bad arguments");
  const long long total8 = (long long)B * HW * (C >> 3);
  hipLaunchKernelGGL(scale_channels_kernel, dim3(grid_for(total8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_in, d_gate,
                     d_out, total8, HW, C, split);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_op_act(pt_engine* e, const uint16_t* d_in, long long n_elems, int kind, float alpha, float beta, uint16_t* d_out, int C, int split,
              pt_stream stream) {
  PT_REQUIRE(e && d_in && d_out && n_elems > 0 && n_elems % 8 == 0, "pt_op_act: bad arguments");
  PT_REQUIRE(kind == 1 || kind == 2 || (kind >= 4 && kind <= 8), "pt_op_act: activation kind %d unsupported", kind);
  PT_REQUIRE(!split || (C > 0 && C % 8 == 0 && n_elems % C == 0), "pt_op_act: a (hi | lo) tensor needs its channel count (C=%d)", C);
  hipLaunchKernelGGL(act_kernel, dim3(grid_for(n_elems / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_in, d_out,
                     n_elems / 8, kind, alpha, beta, split ? C : 8, split);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_op_copy_channels(pt_engine* e, const uint16_t* d_src, long long npix, int src_cstride, int src_coff, uint16_t* d_dst, int dst_cstride, int dst_coff,
                        int n, pt_stream stream) {
  PT_REQUIRE(e && d_src && d_dst && npix > 0 && n > 0 && src_coff >= 0 && dst_coff >= 0 && src_coff + n <= src_cstride && dst_coff + n <= dst_cstride,
             "pt_op_copy_channels: bad arguments");
  hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(npix * n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_src, d_dst, npix, src_cstride,
                     src_coff, dst_cstride, dst_coff, n);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_op_upsample_nearest(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int C, int factor, uint16_t* d_out, pt_stream stream) {
  PT_REQUIRE(e && d_in && d_out && C % 8 == 0 && factor >= 1 && B > 0 && H > 0 && W > 0, "pt_op_upsample_nearest: bad arguments");
  hipLaunchKernelGGL(upsample_kernel, dim3(grid_for((long long)B * H * factor * W * factor * (C / 8))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     d_in, d_out, B, H, W, C, factor);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_copy_bytes(pt_engine* e, const void* src, void* dst, long long nbytes, pt_stream stream) {
  PT_REQUIRE(e && src && dst && nbytes > 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "pt_copy_bytes: null or unaligned pointer");
  const long long n16 = nbytes / 16;
  const int tail = (int)(nbytes - n16 * 16);
  long long g = (n16 + 255) / 256;
  g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
  hipLaunchKernelGGL(copy_bytes_kernel, dim3((unsigned)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const uint4*>(src),
                     reinterpret_cast<uint4*>(dst), n16, reinterpret_cast<const unsigned char*>(src) + n16 * 16, reinterpret_cast<unsigned char*>(dst) + n16 * 16, tail);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_op_mul(pt_engine* e, const uint16_t* d_a, const uint16_t* d_b, uint16_t* d_out, long long n_elems, int C, int split, pt_stream stream) {
  PT_REQUIRE(e && d_a && d_b && d_out && n_elems > 0 && n_elems % 8 == 0, "pt_op_mul: bad arguments");
  PT_REQUIRE(!split || (C > 0 && C % 8 == 0 && n_elems % C == 0), "pt_op_mul: a (hi | lo) tensor needs its channel count (C=%d)", C);
  hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n_elems / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_a, d_b, d_out, n_elems / 8,
                     split ? C : 8, split);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_op_layernorm(pt_engine* e, const uint16_t* d_in, long long rows, int c_pad, int c, const float* d_gamma, const float* d_beta, float eps,
                    uint16_t* d_out, int split, pt_stream stream) {
  PT_REQUIRE(e && d_in && d_gamma && d_beta && d_out && rows > 0 && c > 0 && c <= c_pad, "pt_op_layernorm: bad arguments");
  hipLaunchKernelGGL(layernorm_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_in, d_out, rows, c_pad,
                     c, d_gamma, d_beta, eps, split);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_op_softmax(pt_engine* e, const uint16_t* d_in, long long rows, int c_pad, int c, float* d_out_f32, uint16_t* d_out_bf16, int split, pt_stream stream) {
  PT_REQUIRE(e && d_in && rows > 0 && c > 0 && c <= c_pad && ((d_out_f32 != nullptr) != (d_out_bf16 != nullptr)), "pt_op_softmax: bad arguments (one output)");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_in, rows, c_pad, c,
                     d_out_f32, d_out_bf16, split);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_op_attention(pt_engine* e, const uint16_t* d_qkv, int B, int T, int heads, int d, int qkv_cstride, float scale, uint16_t* d_out, int out_cstride,
                    int split, pt_stream stream) {
  PT_REQUIRE(e && d_qkv && d_out && B > 0 && T > 0 && T <= 1024 && heads > 0 && d > 0 && d <= 64 && 3 * heads * d <= qkv_cstride && heads * d <= out_cstride,
             "pt_op_attention: bad arguments (T <= 1024, head size <= 64)");
  hipLaunchKernelGGL(attention_rows_kernel, dim3(T, heads, B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), d_qkv, d_out, T, heads, d, qkv_cstride,
                     out_cstride, scale, split);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

}  // namespace api
}  // namespace PT_FMT_NS
