// graph_ops.hip -- single-operator entry points for the generic ONNX layer-list executor (pdf_table_amd/onnx_exec.py;
// the reference runs its default models as arbitrary ONNX graphs through onnxruntime,
// utils/deploy_utils.py:243-280).  The executor walks the importer's engine layer list and issues one call per layer:
// the convolutions go to pt_op_conv2d (MFMA implicit GEMM, conv_igemm.hip), everything else to the entry points below --
// thin wrappers around the launchers the dedicated launch graphs use (depthwise conv, pooling, channel mean, add), plus two
// element-wise kernels the dedicated graphs fuse into epilogues (channel scale of an SE block, a stand-alone activation).
// bf16 NHWC activations, channel counts padded to multiples of 8 by the caller (zero padding stays zero through every op
// here except sigmoid / hardsigmoid, whose padded lanes the caller never reads).
#include "common.h"

namespace {

__device__ __forceinline__ float g_bf(uint32_t bits16) { return __uint_as_float(bits16 << 16); }
__device__ __forceinline__ uint32_t g_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}

// x [B, HW, C] *= gate [B, C] (the Mul of a squeeze-and-excitation block)
__global__ __launch_bounds__(256) void scale_channels_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gate,
                                                             bf16_t* __restrict__ out, long long total8, int HW, int C) {
  const int cg = C >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cg);
    const long long pix = i / cg;
    const int b = (int)(pix / HW);
    const uint4 xv = *reinterpret_cast<const uint4*>(x + i * 8);
    const uint4 gv = *reinterpret_cast<const uint4*>(gate + ((size_t)b * C + c8 * 8));
    const uint32_t xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float lo = g_bf(xs[k] & 0xFFFFu) * g_bf(gs[k] & 0xFFFFu);
      const float hi = g_bf(xs[k] >> 16) * g_bf(gs[k] >> 16);
      o[k] = g_f2bf(lo) | (g_f2bf(hi) << 16);
    }
    *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// kind: 1 relu, 2 hardswish, 4 sigmoid, 5 hardsigmoid (max(0, min(1, alpha x + beta))), 6 relu6
__global__ __launch_bounds__(256) void act_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, long long total8, int kind,
                                                  float alpha, float beta) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const uint4 xv = *reinterpret_cast<const uint4*>(x + i * 8);
    const uint32_t xs[4] = {xv.x, xv.y, xv.z, xv.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v[2] = {g_bf(xs[k] & 0xFFFFu), g_bf(xs[k] >> 16)};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float t = v[j];
        if (kind == 1) t = fmaxf(t, 0.f);
        else if (kind == 2) t = t * fminf(fmaxf(t + 3.f, 0.f), 6.f) / 6.f;
        else if (kind == 4) t = 1.f / (1.f + expf(-t));
        else if (kind == 5) t = fmaxf(0.f, fminf(1.f, alpha * t + beta));
        else if (kind == 6) t = fminf(fmaxf(t, 0.f), 6.f);
        v[j] = t;
      }
      o[k] = g_f2bf(v[0]) | (g_f2bf(v[1]) << 16);
    }
    *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// AveragePool k x k, stride k, no padding: out [B, H/k, W/k, C]
__global__ __launch_bounds__(256) void avgpool_kxk_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int H, int W, int C,
                                                          int k) {
  const int Ho = H / k, Wo = W / k, cg = C >> 3;
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
        const uint4 v = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + oy * k + dy) * W + ox * k + dx) * C + c8 * 8);
        const uint32_t vs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[2 * q] += g_bf(vs[q] & 0xFFFFu);
          acc[2 * q + 1] += g_bf(vs[q] >> 16);
        }
      }
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = g_f2bf(acc[2 * q] * inv) | (g_f2bf(acc[2 * q + 1] * inv) << 16);
    *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

inline unsigned grid_for(long long n) {
  const long long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace

extern "C" {

int pt_op_dwconv(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int C, const float* d_w_taps, const float* d_bias, int k,
                 int stride, int act, uint16_t* d_out, pt_stream stream) {
  PT_REQUIRE(e && d_in && d_w_taps && d_bias && d_out, "pt_op_dwconv: null pointer");
  PT_REQUIRE((k == 3 || k == 5) && (stride == 1 || stride == 2) && C % 8 == 0 && act >= 0 && act <= 2,
             "pt_op_dwconv: k=%d stride=%d C=%d act=%d unsupported (k 3/5, stride 1/2, C multiple of 8, act 0/1/2)", k, stride, C, act);
  return pt_launch_dwconv(d_in, d_w_taps, d_bias, d_out, B, H, W, C, k, stride, act, 0, reinterpret_cast<hipStream_t>(stream), nullptr);
}

int pt_op_add(pt_engine* e, const uint16_t* d_a, const uint16_t* d_b, uint16_t* d_out, long long npix, int C, pt_stream stream) {
  PT_REQUIRE(e && d_a && d_b && d_out && npix > 0 && C % 8 == 0, "pt_op_add: bad arguments");
  return pt_launch_add(d_a, d_b, d_out, npix, C, 0, reinterpret_cast<hipStream_t>(stream));
}

int pt_op_maxpool(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int C, int k, int stride, int pad, uint16_t* d_out,
                  pt_stream stream) {
  PT_REQUIRE(e && d_in && d_out && C % 8 == 0, "pt_op_maxpool: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (k == 3 && stride == 2 && pad == 1) return pt_launch_maxpool3x3s2(d_in, B, H, W, C, d_out, 0, s);
  PT_REQUIRE(k == stride && pad == 0 && k >= 2 && H % k == 0 && W % k == 0,
             "pt_op_maxpool: only MaxPool(3, 2, 1) and non-overlapping k x k pools on sizes divisible by k (got k=%d stride=%d pad=%d)", k,
             stride, pad);
  return pt_launch_maxpool_kxk(d_in, B, H, W, C, k, k, 0, 0, d_out, s);
}

int pt_op_avgpool(pt_engine* e, const uint16_t* d_in, int B, int H, int W, int C, int k, uint16_t* d_out, pt_stream stream) {
  PT_REQUIRE(e && d_in && d_out && C % 8 == 0 && k >= 2 && H % k == 0 && W % k == 0, "pt_op_avgpool: k x k / stride k on sizes divisible by k");
  hipLaunchKernelGGL(avgpool_kxk_kernel, dim3(grid_for((long long)B * (H / k) * (W / k) * (C / 8))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), d_in, d_out, B, H, W, C, k);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_op_chan_mean(pt_engine* e, const uint16_t* d_in, int B, int HW, int C, float* d_scratch, uint16_t* d_mean, pt_stream stream) {
  PT_REQUIRE(e && d_in && d_scratch && d_mean, "pt_op_chan_mean: null pointer");
  return pt_launch_chan_mean(d_in, B, HW, C, 0, d_scratch, d_mean, B, reinterpret_cast<hipStream_t>(stream));
}

int pt_op_chan_mean_scratch_floats(int B, int C) { return PT_SE_CHUNKS * B * C; }

int pt_op_scale_channels(pt_engine* e, const uint16_t* d_in, const uint16_t* d_gate, int B, int HW, int C, uint16_t* d_out,
                         pt_stream stream) {
  PT_REQUIRE(e && d_in && d_gate && d_out && C % 8 == 0 && B > 0 && HW > 0, "pt_op_scale_channels: bad arguments");
  const long long total8 = (long long)B * HW * (C >> 3);
  hipLaunchKernelGGL(scale_channels_kernel, dim3(grid_for(total8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_in, d_gate,
                     d_out, total8, HW, C);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_op_act(pt_engine* e, const uint16_t* d_in, long long n_elems, int kind, float alpha, float beta, uint16_t* d_out,
              pt_stream stream) {
  PT_REQUIRE(e && d_in && d_out && n_elems > 0 && n_elems % 8 == 0, "pt_op_act: bad arguments");
  PT_REQUIRE(kind == 1 || kind == 2 || kind == 4 || kind == 5 || kind == 6, "pt_op_act: activation kind %d unsupported", kind);
  hipLaunchKernelGGL(act_kernel, dim3(grid_for(n_elems / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_in, d_out,
                     n_elems / 8, kind, alpha, beta);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

}  // extern "C"
