// mtl_model.hip -- backbone of MtlTabNet / TableMaster (second half) on the engine's kernels.
//
// Reference graph: TableResNetExtra.forward (model/table/mtl_tabnet/table_resnet_extra.py:205-318, configuration
// mtl_tabnet_config.py:41-53): conv1 (3 -> 64) / conv2 (64 -> 128) + BN + ReLU, max-pool, four stages of BasicBlocks [1, 2, 5, 3]
// (:164-202) each followed by a 3x3 conv + BN + ReLU, max-pools after stages 1 and 2; the FIRST block of stages 2-4 carries a
// global-context block (ContextBlock :36-161: attention pooling over H*W, 1x1 -> LayerNorm -> ReLU -> 1x1, broadcast add).
// The decoders run on the LAST feature map (table_master.py: feat[-1]), which is what this entry point returns.
// Mapping: every conv (+ folded BN, + ReLU, + residual where no context block sits in between) is conv_igemm_kernel;
// max-pools are maxpool_kxk_kernel; the context block is four small kernels (gc_logits / gc_softmax / gc_pool / gc_mlp, below)
// + gc_add_relu_kernel (out = ReLU(x + t[c] + residual)).
// Only the backbone is on the engine so far: the decoders exist as the pinned oracle (oracle/mtl_tabnet.py).
#include <math.h>

#include <string>

#include "common.h"

namespace PT_FMT_NS {

namespace {

// 16 stored bits <-> fp32 in the storage format of this namespace (act16.h: bf16, or IEEE half in pt_f16)
__device__ __forceinline__ float mbf2f(uint32_t b) { return a16_to_f32(b); }
__device__ __forceinline__ uint32_t mf2bf(float f) { return f32_to_a16(f); }
__device__ __forceinline__ float mget(const bf16_t* p, int lo_off, int split) {
  float v = mbf2f(p[0]);
  if (split) v += mbf2f(p[lo_off]);
  return v;
}
__device__ __forceinline__ void mput(bf16_t* p, int lo_off, int split, float v) {
  const uint32_t h = mf2bf(v);
  p[0] = (bf16_t)h;
  if (split) p[lo_off] = (bf16_t)mf2bf(v - mbf2f(h));
}

__device__ float block_reduce(float v, float* red, bool is_max) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const float o = __shfl_xor(v, m);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
  return r;
}

// ContextBlock (pooling 'att', one header, fusion 'channel_add'; table_resnet_extra.py:89-141) in four small kernels, x [B, HW, C]
// (hi | lo), C <= 512:
//   gc_logits_kernel    logit[b][px] = wm . x[b][px] + bm                  one wave per pixel, lanes over channels
//   gc_softmax_kernel   p[b][:] = softmax over HW                          one workgroup per image (HW <= 14 400 floats)
//   gc_pool_kernel      part[b][chunk][c] = sum over the chunk's pixels of x[px][c] p[px]    thread = channel, 128-pixel chunks
//   gc_mlp_kernel       ctx = sum of the partials (fixed order: deterministic), t = W3 . ReLU(LayerNorm(W0 . ctx + b0)) + b3
// (a first version did all of it in ONE workgroup per image: 4.5 ms per call at 16 tables of 120x120x256, 55 % of the backbone)
constexpr int GC_CHUNK = 128;

__global__ __launch_bounds__(256) void gc_logits_kernel(const bf16_t* __restrict__ x, long long npix, int C, const float* __restrict__ wm,
                                                        const float* __restrict__ bm, float* __restrict__ logit, int split) {
  a16_kernel_enter();
  const long long px = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (px >= npix) return;
  const bf16_t* r = x + (size_t)px * (split ? 2 * C : C);
  float a = 0.f;
  for (int c = lane; c < C; c += 64) a = fmaf(mget(r + c, C, split), wm[c], a);
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m);
  if (lane == 0) logit[px] = a + bm[0];
}

__global__ __launch_bounds__(512) void gc_softmax_kernel(float* __restrict__ logit, int HW) {
  a16_kernel_enter();
  __shared__ float red[8];
  float* l = logit + (size_t)blockIdx.x * HW;
  float mx = -INFINITY;
  for (int px = threadIdx.x; px < HW; px += blockDim.x) mx = fmaxf(mx, l[px]);
  mx = block_reduce(mx, red, true);
  float sum = 0.f;
  for (int px = threadIdx.x; px < HW; px += blockDim.x) {
    const float e = expf(l[px] - mx);
    l[px] = e;
    sum += e;
  }
  sum = block_reduce(sum, red, false);
  __syncthreads();
  for (int px = threadIdx.x; px < HW; px += blockDim.x) l[px] = l[px] / sum;
}

__global__ __launch_bounds__(512) void gc_pool_kernel(const bf16_t* __restrict__ x, const float* __restrict__ p, int HW, int C, int nchunk,
                                                      float* __restrict__ part, int split) {
  a16_kernel_enter();
  const int b = blockIdx.y, ch = blockIdx.x, c = threadIdx.x;
  if (c >= C) return;
  const int cs = split ? 2 * C : C;
  const int p0 = ch * GC_CHUNK, p1 = min(HW, p0 + GC_CHUNK);
  const bf16_t* xb = x + ((size_t)b * HW + p0) * cs + c;
  const float* pb = p + (size_t)b * HW;
  float a = 0.f;
  for (int px = p0; px < p1; ++px, xb += cs) a = fmaf(mget(xb, C, split), pb[px], a);
  part[((size_t)b * nchunk + ch) * C + c] = a;
}

__global__ __launch_bounds__(512) void gc_mlp_kernel(const float* __restrict__ part, int nchunk, int C, int hid, const float* __restrict__ w0,
                                                     const float* __restrict__ b0, const float* __restrict__ lg, const float* __restrict__ lb,
                                                     const float* __restrict__ w3, const float* __restrict__ b3, float* __restrict__ t) {
  a16_kernel_enter();
  __shared__ float ctx[512];
  __shared__ float hbuf[64];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  for (int c = tid; c < C; c += nt) {
    float a = 0.f;
    for (int k = 0; k < nchunk; ++k) a += part[((size_t)b * nchunk + k) * C + c];
    ctx[c] = a;
  }
  __syncthreads();
  if (tid < hid) {
    float a = b0[tid];
    for (int c = 0; c < C; ++c) a = fmaf(w0[tid * C + c], ctx[c], a);
    hbuf[tid] = a;
  }
  __syncthreads();
  if (tid == 0) {                        // LayerNorm([hid, 1, 1]) + ReLU over <= 64 values
    float m = 0.f, q = 0.f;
    for (int j = 0; j < hid; ++j) m += hbuf[j];
    m /= (float)hid;
    for (int j = 0; j < hid; ++j) q += (hbuf[j] - m) * (hbuf[j] - m);
    const float rstd = 1.f / sqrtf(q / (float)hid + 1e-5f);
    for (int j = 0; j < hid; ++j) hbuf[j] = fmaxf((hbuf[j] - m) * rstd * lg[j] + lb[j], 0.f);
  }
  __syncthreads();
  for (int c = tid; c < C; c += nt) {
    float a = b3[c];
    for (int j = 0; j < hid; ++j) a = fmaf(w3[c * hid + j], hbuf[j], a);
    t[(size_t)b * C + c] = a;
  }
}

// out = ReLU(x + t[image][c] + res)   (BasicBlock.forward :191-200 with the context block's broadcast add)
__global__ __launch_bounds__(256) void gc_add_relu_kernel(const bf16_t* __restrict__ x, const float* __restrict__ t, const bf16_t* __restrict__ res,
                                                          bf16_t* __restrict__ out, long long total, int HW, int C, int split) {
  a16_kernel_enter();
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long px = i / C;
    const int b = (int)(px / HW);
    const size_t o = (size_t)px * (split ? 2 * C : C) + c;
    mput(out + o, C, split, fmaxf(mget(x + o, C, split) + t[(size_t)b * C + c] + mget(res + o, C, split), 0.f));
  }
}

}  // namespace

// x bf16 [n, H, W, 32] (3 real channels, the rest zero; hi/lo mode: [hi 32 | lo 32]) -> f3 fp32 [n, H/8, W/8, 512]
int pt_mtl_backbone_forward_net(pt_engine* e, const bf16_t* x, int n, int H, int W_, float* f3, hipStream_t s) {
  PT_REQUIRE(e && x && f3 && n > 0 && H > 0 && W_ > 0 && H % 8 == 0 && W_ % 8 == 0, "mtl backbone: input %dx%d must be multiples of 8", H, W_);
  auto it = e->models.find(PT_MODEL_MTL_BACKBONE);
  if (it == e->models.end()) {
    pt_set_error("MtlTabNet backbone weights not loaded (pt_weights_load(PT_MODEL_MTL_BACKBONE))");
    return PT_ERR_STATE;
  }
  if (!pt_model_format_ok(it->second, "PT_MODEL_MTL_BACKBONE")) return PT_ERR_STATE;
  const PtModel& M = it->second;
  const int x3 = pt_split(e) ? 1 : 0, m = x3 ? 2 : 1;
  int rc = PT_OK;
  auto get = [&](const std::string& name) -> const PtTensor* {
    const PtTensor* t = M.find(name);
    if (!t && rc == PT_OK) {
      pt_set_error("MtlTabNet backbone weight blob lacks tensor '%s'", name.c_str());
      rc = PT_ERR_FORMAT;
    }
    return t;
  };
  PtArena& A = e->arenas[PT_ARENA_TSR];
  const size_t big = (size_t)n * H * W_ * 128;            // the largest activation: conv2's output at full resolution
  bf16_t* buf[4] = {nullptr, nullptr, nullptr, nullptr};
  float *tvec = nullptr, *scratch = nullptr, *part = nullptr;
  for (int attempt = 0; attempt < 2; ++attempt) {
    A.reset();
    bool ok = true;
    for (int i = 0; i < 4; ++i) {
      buf[i] = reinterpret_cast<bf16_t*>(A.take(big * m * sizeof(bf16_t)));
      if (!buf[i]) ok = false;
    }
    tvec = reinterpret_cast<float*>(A.take((size_t)n * 512 * sizeof(float)));
    scratch = reinterpret_cast<float*>(A.take((size_t)n * (H / 4) * (W_ / 4) * sizeof(float)));
    part = reinterpret_cast<float*>(A.take((size_t)n * (((H / 4) * (W_ / 4) + GC_CHUNK - 1) / GC_CHUNK) * 512 * sizeof(float)));
    if (!tvec || !scratch || !part) ok = false;
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
  auto conv = [&](const bf16_t* in, int hh, int ww, int cin, const std::string& q, int N, int ks, bf16_t* out, int relu, const bf16_t* res,
                  float* out_f32 = nullptr) {
    const PtTensor* w = get(q + (x3 ? ".w3" : ".w"));
    const PtTensor* b = get(q + ".b");
    if (rc != PT_OK) return;
    ConvDesc c;
    c.in = in; c.B = n; c.H = hh; c.W = ww; c.Cin = cin;
    c.w = reinterpret_cast<const bf16_t*>(w->d_ptr); c.bias = reinterpret_cast<const float*>(b->d_ptr);
    c.N = N; c.ks = ks; c.stride = 1; c.relu = relu; c.split = x3;
    if (out_f32) {
      c.out_f32 = out_f32; c.out_cstride = N;
    } else {
      c.out = out; c.out_cstride = N * m; c.out_lo_off = N;
    }
    if (res) { c.res = res; c.res_mode = 1; }
    const int r = pt_launch_conv(e, c, s);
    if (r != PT_OK) rc = r;
  };
  // one BasicBlock: in (buf[a]) -> out (returned buffer index); uses the other three buffers as temporaries
  auto block = [&](int a, int hh, int ww, int cin, int planes, const std::string& q, bool gcb) -> int {
    const int t1 = (a + 1) & 3, t2 = (a + 2) & 3, t3 = (a + 3) & 3;
    conv(buf[a], hh, ww, cin, q + ".conv1", planes, 3, buf[t1], 1, nullptr);
    const bf16_t* res = buf[a];
    if (cin != planes) {
      conv(buf[a], hh, ww, cin, q + ".down", planes, 1, buf[t2], 0, nullptr);
      res = buf[t2];
    }
    if (!gcb) {
      conv(buf[t1], hh, ww, planes, q + ".conv2", planes, 3, buf[t3], 1, res);
      return t3;
    }
    conv(buf[t1], hh, ww, planes, q + ".conv2", planes, 3, buf[t3], 0, nullptr);
    const PtTensor *wm = get(q + ".gc.wm"), *bm = get(q + ".gc.bm"), *w0 = get(q + ".gc.w0"), *b0 = get(q + ".gc.b0"), *lg = get(q + ".gc.lg"),
                   *lb = get(q + ".gc.lb"), *w3 = get(q + ".gc.w3"), *b3 = get(q + ".gc.b3");
    if (rc != PT_OK) return t3;
    const int hid = (int)w0->dims[0];
    // the context kernels hold ctx[512] / hidden[64] in LDS and launch `planes` threads: a checkpoint with another gcb ratio or width must fail here
    if (!(hid > 0 && hid <= 64 && planes <= 512 && (int)w0->dims[1] == planes && (int)w3->dims[0] == planes && (int)w3->dims[1] == hid)) {
      pt_set_error("MtlTabNet backbone: context block '%s' has hidden %d / %u x %u weights for %d planes (built: hidden <= 64, planes <= 512)", q.c_str(), hid,
                   w0->dims[0], w0->dims[1], planes);
      rc = PT_ERR_FORMAT;
      return t3;
    }
    auto F = [](const PtTensor* t) { return reinterpret_cast<const float*>(t->d_ptr); };
    {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "mtl gc context");
      const int HW = hh * ww, nchunk = (HW + GC_CHUNK - 1) / GC_CHUNK;
      const long long npix = (long long)n * HW;
      hipLaunchKernelGGL(gc_logits_kernel, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, s, buf[t3], npix, planes, F(wm), F(bm), scratch, x3);
      hipLaunchKernelGGL(gc_softmax_kernel, dim3(n), dim3(512), 0, s, scratch, HW);
      hipLaunchKernelGGL(gc_pool_kernel, dim3(nchunk, n), dim3(planes), 0, s, buf[t3], scratch, HW, planes, nchunk, part, x3);
      hipLaunchKernelGGL(gc_mlp_kernel, dim3(n), dim3(512), 0, s, part, nchunk, planes, hid, F(w0), F(b0), F(lg), F(lb), F(w3), F(b3), tvec);
    }
    const long long total = (long long)n * hh * ww * planes;
    long long g = (total + 255) / 256;
    if (g > 65536) g = 65536;
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "mtl gc add+relu");
    hipLaunchKernelGGL(gc_add_relu_kernel, dim3((unsigned)g), dim3(256), 0, s, buf[t3], tvec, res, buf[t1], total, hh * ww, planes, x3);
    return t1;
  };
  static const int LAYERS[4] = {1, 2, 5, 3}, PLANES[4] = {256, 256, 512, 512};
  int hh = H, ww = W_;
  conv(x, hh, ww, 32, "conv1", 64, 3, buf[0], 1, nullptr);
  conv(buf[0], hh, ww, 64, "conv2", 128, 3, buf[1], 1, nullptr);
  if (rc != PT_OK) return rc;
  if ((rc = pt_launch_maxpool_kxk(buf[1], n, hh, ww, 128, 2, 2, 0, x3, buf[0], s)) != PT_OK) return rc;
  hh /= 2; ww /= 2;
  int cur = 0, cin = 128;
  for (int st = 0; st < 4; ++st) {
    for (int j = 0; j < LAYERS[st]; ++j) {
      cur = block(cur, hh, ww, cin, PLANES[st], "layer" + std::to_string(st + 1) + "." + std::to_string(j), st > 0 && j == 0);
      cin = PLANES[st];
      if (rc != PT_OK) return rc;
    }
    const int nxt = (cur + 1) & 3;
    if (st == 3) {
      conv(buf[cur], hh, ww, cin, "conv6", 512, 3, nullptr, 1, nullptr, f3);
    } else {
      conv(buf[cur], hh, ww, cin, "conv" + std::to_string(st + 3), PLANES[st], 3, buf[nxt], 1, nullptr);
      cur = nxt;
      if (st < 2) {
        const int p = (cur + 1) & 3;
        if (rc != PT_OK) return rc;
        if ((rc = pt_launch_maxpool_kxk(buf[cur], n, hh, ww, PLANES[st], 2, 2, 0, x3, buf[p], s)) != PT_OK) return rc;
        cur = p;
        hh /= 2; ww /= 2;
      }
    }
    if (rc != PT_OK) return rc;
  }
  PT_HIP_CHECK(hipGetLastError());
  return rc;
}

}  // namespace PT_FMT_NS
