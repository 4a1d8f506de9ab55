// lore_processor.hip -- Lore's logical-location processor (two 4-layer pre-norm transformer regressors) for a batch
// of tables in one pass.
//
// Reference: LoreProcessModel.forward lore/lore_processor.py:465-514 (evaluation branch): Transformer :81-114 =
// Linear -> L x EncoderLayer :286-313 (x += MHA(Norm(x)); x += FFN(Norm(x))) -> Decoder :64-78; Norm :117-131 (unbiased
// std, eps added to the std); MultiHeadAttention :172-226 (8 heads x 32, softmax(QK^T / sqrt(32)) V); Stacker :342-396.
// The reference runs one table at a time; here the cells of all tables of a batch are concatenated into one token
// matrix for every token-wise op (all GEMMs on the MFMA conv kernel as 1x1 convolutions, fp32 residual stream), and
// only the attention kernel looks at table boundaries (a tile list built on the host from the decode's counts).
#include <math.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "common.h"

namespace PT_FMT_NS {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

namespace {

// 16 stored bits <-> fp32 in the storage format of this namespace (act16.h: bf16, or IEEE half in pt_f16)
__device__ __forceinline__ float bf2f(uint32_t b) { return a16_to_f32(b); }
__device__ __forceinline__ uint32_t f2bf(float f) { return f32_to_a16(f); }
__device__ __forceinline__ void put(bf16_t* p, int lo_off, int split, float v) {
  const uint32_t h = f2bf(v);
  p[0] = (bf16_t)h;
  if (split) p[lo_off] = (bf16_t)f2bf(v - bf2f(h));
}
__device__ __forceinline__ float get(const bf16_t* p, int lo_off, int split) {
  float v = bf2f(p[0]);
  if (split) v += bf2f(p[lo_off]);
  return v;
}

// token t of the batch = row tok[2t+1] of table tok[2t]; writes its 256 features (+ the four 2-D position embeddings
// when pe_dets != null, lore_processor.py:486-491) to x0 [Npad,256] and to channels [256,512) of cat [Npad,512]
__global__ __launch_bounds__(256) void tok_prepare_kernel(const float* __restrict__ logi, const float* __restrict__ dets,
                                                           const int* __restrict__ tok, int N, int use_pe,
                                                           const float* __restrict__ x_pe, const float* __restrict__ y_pe,
                                                           bf16_t* __restrict__ x0, bf16_t* __restrict__ cat, int split) {
  a16_kernel_enter();
  const int t = blockIdx.x, c = threadIdx.x;
  float v = 0.f;
  if (t < N) {
    const int tb = tok[2 * t], r = tok[2 * t + 1];
    v = logi[((size_t)tb * PT_TSR_MAX_CELLS + r) * 256 + c];
    if (use_pe) {
      // filter() casts the float quad to int32 (truncation, :576-580), normalized_ps rounds and clamps to [0, 255] (:585-589)
      const float* d = dets + ((size_t)tb * PT_TSR_MAX_CELLS + r) * 9;
      int p0 = (int)d[0], p1 = (int)d[1], p2 = (int)d[2], p5 = (int)d[5];
      p0 = min(max(p0, 0), 255); p1 = min(max(p1, 0), 255); p2 = min(max(p2, 0), 255); p5 = min(max(p5, 0), 255);
      v = v + x_pe[p0 * 256 + c] + y_pe[p1 * 256 + c] + x_pe[p2 * 256 + c] + y_pe[p5 * 256 + c];
    }
  }
  put(x0 + (size_t)t * (split ? 512 : 256) + c, 256, split, v);
  put(cat + (size_t)t * (split ? 1024 : 512) + 256 + c, 512, split, v);
}

// Norm.forward (:126-131) over 256 channels, or a plain fp32 -> bf16 conversion when alpha == null.  One wave per token.
__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                    const float* __restrict__ bias, bf16_t* __restrict__ out, int Npad,
                                                    int split) {
  a16_kernel_enter();
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= Npad) return;
  const float4 v = *reinterpret_cast<const float4*>(x + (size_t)t * 256 + lane * 4);
  float o[4] = {v.x, v.y, v.z, v.w};
  if (alpha) {
    float s = v.x + v.y + v.z + v.w;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
    const float mean = s / 256.f;
    float d[4] = {v.x - mean, v.y - mean, v.z - mean, v.w - mean};
    float q = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) q += __shfl_xor(q, m);
    const float den = sqrtf(q / 255.f) + 1e-6f;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = alpha[lane * 4 + k] * d[k] / den + bias[lane * 4 + k];
  }
  bf16_t* op = out + (size_t)t * (split ? 512 : 256) + lane * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) put(op + k, 256, split, o[k]);
}

// fp32 [Npad, 8] (4 valid) -> bf16 [Npad, 32] zero padded: the stacker's logi_encoder input (:383)
__global__ __launch_bounds__(256) void cvt_logic_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int Npad,
                                                         int split) {
  a16_kernel_enter();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Npad * 32) return;
  const int t = i >> 5, c = i & 31;
  put(out + (size_t)t * (split ? 64 : 32) + c, 32, split, c < 4 ? x[(size_t)t * 8 + c] : 0.f);
}

// attention (:134-163) for one head and 32 queries of one table, one wave, on the matrix cores.
// tiles[3i..] = (first token of the table, query offset inside it, cells in the table).
// qkv [Npad, 768] = [q | k | v] (split: [hi 768 | lo 768]); out [Npad, 256].
//   S^T (32 keys x 32 queries) = K_tile Q^T: two v_mfma_f32_32x32x16_bf16 over d = 32 (A = key rows, B = query rows,
//   both 16-byte row chunks straight from global).  In the D layout a lane owns ONE query (column lane & 31) and 16 of
//   the 32 keys, so the online soft-max is in-lane plus one exchange with lane ^ 32.
//   O^T (32 d x 32 queries) += V^T P^T: the B operand is exactly the 8 probabilities a lane already holds per
//   k-step (keys taken in the D layout's own order -- a sum does not care), the A operand gathers V[key][d] in that
//   same key order.  BF16X3 mode: three passes each (hi*hi + hi*lo + lo*hi), fp32 soft-max.
typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8;
typedef __attribute__((ext_vector_type(16))) float af32x16;

__device__ __forceinline__ abf16x8 ld8(const bf16_t* p) { return *reinterpret_cast<const abf16x8*>(p); }

template <int SPLIT>
__global__ __launch_bounds__(64) void attention_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ tiles,
                                                        bf16_t* __restrict__ out) {
  a16_kernel_enter();
  const int tile = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
  const int tok0 = tiles[3 * tile], q0 = tiles[3 * tile + 1], nseg = tiles[3 * tile + 2];
  constexpr int cs = SPLIT ? 1536 : 768;
  const int col = lane & 31, half = lane >> 5;
  const int qi = q0 + col;
  const bf16_t* qrow = qkv + (size_t)(tok0 + (qi < nseg ? qi : nseg - 1)) * cs + head * 32 + half * 8;
  abf16x8 qh[2], ql[2];
  qh[0] = ld8(qrow); qh[1] = ld8(qrow + 16);
  if (SPLIT) { ql[0] = ld8(qrow + 768); ql[1] = ld8(qrow + 768 + 16); }
  af32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < nseg; k0 += 32) {
    // ---- scores: rows = keys k0 + (r & 3) + 8 (r >> 2) + 4 half, column = this lane's query
    const int krow = k0 + col;
    const bf16_t* kp = qkv + (size_t)(tok0 + (krow < nseg ? krow : nseg - 1)) * cs + 256 + head * 32 + half * 8;
    af32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
    {
      const abf16x8 k0h = ld8(kp), k1h = ld8(kp + 16);
      sc = mfma_32x32x16_a16(k0h, qh[0], sc);
      sc = mfma_32x32x16_a16(k1h, qh[1], sc);
      if (SPLIT) {
        sc = mfma_32x32x16_a16(k0h, ql[0], sc);
        sc = mfma_32x32x16_a16(k1h, ql[1], sc);
        const abf16x8 k0l = ld8(kp + 768), k1l = ld8(kp + 768 + 16);
        sc = mfma_32x32x16_a16(k0l, qh[0], sc);
        sc = mfma_32x32x16_a16(k1l, qh[1], sc);
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      sc[r] = key < nseg ? sc[r] / 5.656854249492381f : -INFINITY;      // scores / math.sqrt(32) (:135)
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
    for (int r = 0; r < 16; ++r) acc[r] *= scale;
    // ---- O^T += V^T P^T; k-step s covers this half's registers 8s .. 8s+7, i.e. keys k0 + (j & 3) + 8 (2s + (j >> 2)) + 4 half
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      abf16x8 ph, pl, vh, vl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float pv = p[8 * s2 + j];
        const uint32_t hb = f2bf(pv);
        ph[j] = __builtin_bit_cast(__bf16, (uint16_t)hb);
        if (SPLIT) pl[j] = __builtin_bit_cast(__bf16, (uint16_t)f2bf(pv - bf2f(hb)));
        int key = k0 + (j & 3) + 8 * (2 * s2 + (j >> 2)) + 4 * half;
        if (key >= nseg) key = nseg - 1;                 // its probability is exactly 0
        const bf16_t* vp = qkv + (size_t)(tok0 + key) * cs + 512 + head * 32 + col;     // A row = d = lane & 31
        vh[j] = __builtin_bit_cast(__bf16, vp[0]);
        if (SPLIT) vl[j] = __builtin_bit_cast(__bf16, vp[768]);
      }
      acc = mfma_32x32x16_a16(vh, ph, acc);
      if (SPLIT) {
        acc = mfma_32x32x16_a16(vh, pl, acc);
        acc = mfma_32x32x16_a16(vl, ph, acc);
      }
    }
  }
  if (qi < nseg) {
    bf16_t* op = out + (size_t)(tok0 + qi) * (SPLIT ? 512 : 256) + head * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = (r & 3) + 8 * (r >> 2) + 4 * half;
      put(op + d, 256, SPLIT, acc[r] / l);
    }
  }
}

// fp32 [Npad, 8] token rows -> out[table][row][4]
__global__ __launch_bounds__(256) void scatter4_kernel(const float* __restrict__ x, const int* __restrict__ tok, int N,
                                                        float* __restrict__ out) {
  a16_kernel_enter();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * 4) return;
  const int t = i >> 2, c = i & 3;
  out[((size_t)tok[2 * t] * PT_TSR_MAX_CELLS + tok[2 * t + 1]) * 4 + c] = x[(size_t)t * 8 + c];
}

struct P {
  pt_engine* e;
  const PtModel* m;
  hipStream_t s;
  int Npad, x3, mul, rc;
  const PtTensor* get(const std::string& n) {
    const PtTensor* t = m->find(n);
    if (!t && rc == PT_OK) {
      pt_set_error("Lore processor weight blob lacks tensor '%s'", n.c_str());
      rc = PT_ERR_FORMAT;
    }
    return t;
  }
  // y = x W^T + b over all tokens: x bf16 [Npad, cin] -> bf16 [Npad, out_c] at channel out_coff, or fp32 [Npad, f32_cs]
  void gemm(const bf16_t* x, int cin, const std::string& q, int N, int relu, bf16_t* out, int out_c, int out_coff,
            float* out_f32 = nullptr, int f32_cs = 0, const float* res_f32 = nullptr, int nv = 0) {
    const PtTensor* w = get(q + (x3 ? ".w3" : ".w"));
    const PtTensor* b = get(q + ".b");
    if (rc != PT_OK) return;
    ConvDesc c;
    c.in = x; c.B = 1; c.H = Npad / 32; c.W = 32; c.Cin = cin;
    c.w = reinterpret_cast<const bf16_t*>(w->d_ptr); c.bias = reinterpret_cast<const float*>(b->d_ptr);
    c.N = N; c.ks = 1; c.stride = 1; c.relu = relu; c.split = x3; c.n_valid = nv;
    if (out_f32) {
      c.out_f32 = out_f32; c.out_cstride = f32_cs; c.res_f32 = res_f32;
    } else {
      c.out = out; c.out_cstride = out_c * mul; c.out_coff = out_coff; c.out_lo_off = out_c;
    }
    const int r = pt_launch_conv(e, c, s);
    if (r != PT_OK) rc = r;
  }
};

}  // namespace

int pt_lore_process(pt_engine* e, const float* d_logi, const float* d_dets, const int32_t* h_counts, int n_tables,
                    int use_2dpe, float* d_logic, float* d_stacked, hipStream_t s) {
  PT_REQUIRE(d_logi && h_counts && n_tables > 0 && d_logic && d_stacked, "tsr process: null pointer");
  PT_REQUIRE(!use_2dpe || d_dets, "tsr process: 2-D position embeddings need the cell quads");
  auto it = e->models.find(PT_MODEL_LORE_PROCESSOR);
  if (it == e->models.end()) {
    pt_set_error("Lore processor weights not loaded (pt_weights_load(PT_MODEL_LORE_PROCESSOR))");
    return PT_ERR_STATE;
  }
  if (!pt_model_format_ok(it->second, "PT_MODEL_LORE_PROCESSOR")) return PT_ERR_STATE;
  std::vector<int> tok, tiles;
  int N = 0;
  for (int t = 0; t < n_tables; ++t) {
    const int c = h_counts[t];
    PT_REQUIRE(c >= 0 && c <= PT_TSR_MAX_CELLS, "tsr process: bad cell count %d", c);
    for (int q0 = 0; q0 < c; q0 += 32) { tiles.push_back(N); tiles.push_back(q0); tiles.push_back(c); }
    for (int r = 0; r < c; ++r) { tok.push_back(t); tok.push_back(r); }
    N += c;
  }
  if (N == 0) return PT_OK;
  P p;
  p.e = e; p.m = &it->second; p.s = s; p.rc = PT_OK;
  p.x3 = pt_split(e) ? 1 : 0;
  p.mul = p.x3 ? 2 : 1;
  p.Npad = (N + 127) / 128 * 128;
  const int Npad = p.Npad, x3 = p.x3;
  const PtTensor* meta = p.get("meta");
  if (p.rc != PT_OK) return p.rc;
  int layers[2];
  {
    // the two layer counts: read from the blob once per loaded model (a per-call hipMemcpy would wait for the whole default stream)
    std::vector<int32_t>& hw = it->second.host_words["meta"];
    if (hw.size() != 2) {
      hw.assign(2, 0);
      PT_HIP_CHECK(hipMemcpy(hw.data(), meta->d_ptr, 8, hipMemcpyDeviceToHost));
    }
    layers[0] = hw[0];
    layers[1] = hw[1];
  }

  // activations from the arena (grown once if needed)
  const size_t be = sizeof(bf16_t) * p.mul;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~size_t(255); return o; };
  const size_t o_x = carve((size_t)Npad * 256 * 4), o_x0 = carve((size_t)Npad * 256 * be), o_xb = carve((size_t)Npad * 256 * be),
               o_qkv = carve((size_t)Npad * 768 * be), o_att = carve((size_t)Npad * 256 * be),
               o_h = carve((size_t)Npad * 2048 * be), o_cat = carve((size_t)Npad * 512 * be),
               o_l32 = carve((size_t)Npad * 32 * be), o_le = carve((size_t)Npad * 256 * be),
               o_lg = carve((size_t)Npad * 8 * 4), o_sk = carve((size_t)Npad * 8 * 4),
               o_tok = carve(tok.size() * 4), o_tiles = carve(tiles.size() * 4);
  if (off > e->arenas[PT_ARENA_TSRP].cap) {
    PT_HIP_CHECK(hipDeviceSynchronize());
    if (e->arenas[PT_ARENA_TSRP].base) PT_HIP_CHECK(hipFree(e->arenas[PT_ARENA_TSRP].base));
    e->arenas[PT_ARENA_TSRP].base = nullptr; e->arenas[PT_ARENA_TSRP].cap = 0;
    PT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->arenas[PT_ARENA_TSRP].base), off + (1u << 20)));
    e->arenas[PT_ARENA_TSRP].cap = off + (1u << 20);
  }
  char* base = e->arenas[PT_ARENA_TSRP].base;
  float* x = reinterpret_cast<float*>(base + o_x);
  bf16_t* x0 = reinterpret_cast<bf16_t*>(base + o_x0);
  bf16_t* xb = reinterpret_cast<bf16_t*>(base + o_xb);
  bf16_t* qkv = reinterpret_cast<bf16_t*>(base + o_qkv);
  bf16_t* att = reinterpret_cast<bf16_t*>(base + o_att);
  bf16_t* hbuf = reinterpret_cast<bf16_t*>(base + o_h);
  bf16_t* cat = reinterpret_cast<bf16_t*>(base + o_cat);
  bf16_t* l32 = reinterpret_cast<bf16_t*>(base + o_l32);
  bf16_t* le = reinterpret_cast<bf16_t*>(base + o_le);
  float* lg = reinterpret_cast<float*>(base + o_lg);
  float* sk = reinterpret_cast<float*>(base + o_sk);
  int* d_tok = reinterpret_cast<int*>(base + o_tok);
  int* d_tiles = reinterpret_cast<int*>(base + o_tiles);
  {
    // token and tile maps go through an engine-owned pinned slot (the vectors die with this call, and a stream synchronise here would
    // make the host wait for everything queued on s)
    char* hs = static_cast<char*>(e->stage_ring.acquire((tok.size() + tiles.size()) * 4));
    PT_REQUIRE(hs, "tsr process: pinned staging (%s)", hipGetErrorString(hipGetLastError()));
    memcpy(hs, tok.data(), tok.size() * 4);
    memcpy(hs + tok.size() * 4, tiles.data(), tiles.size() * 4);
    PT_HIP_CHECK(hipMemcpyAsync(d_tok, hs, tok.size() * 4, hipMemcpyHostToDevice, s));
    PT_HIP_CHECK(hipMemcpyAsync(d_tiles, hs + tok.size() * 4, tiles.size() * 4, hipMemcpyHostToDevice, s));
    PT_REQUIRE(e->stage_ring.release(s) == 0, "tsr process: event record failed");
  }

  const PtTensor *xpe = p.get("x_pe"), *ype = p.get("y_pe");
  if (p.rc != PT_OK) return p.rc;
  const int ntiles = (int)tiles.size() / 3;
  {
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "tsr tok prepare");
    hipLaunchKernelGGL(tok_prepare_kernel, dim3(Npad), dim3(256), 0, s, d_logi, d_dets, d_tok, N, use_2dpe,
                       reinterpret_cast<const float*>(xpe->d_ptr), reinterpret_cast<const float*>(ype->d_ptr), x0, cat, x3);
  }
  auto norm = [&](const std::string& q) {
    const PtTensor* a = q.empty() ? nullptr : p.get(q + ".alpha");
    const PtTensor* b = q.empty() ? nullptr : p.get(q + ".bias");
    if (p.rc != PT_OK) return;
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "tsr norm");
    hipLaunchKernelGGL(norm_kernel, dim3((Npad + 3) / 4), dim3(256), 0, s, x, a ? reinterpret_cast<const float*>(a->d_ptr) : nullptr,
                       b ? reinterpret_cast<const float*>(b->d_ptr) : nullptr, xb, Npad, x3);
  };
  auto transformer = [&](const std::string& q, const bf16_t* in, int cin, int nl, float* out4) {
    p.gemm(in, cin, q + ".linear", 256, 0, nullptr, 0, 0, x, 256);
    for (int l = 0; l < nl; ++l) {
      const std::string lq = q + ".l" + std::to_string(l);
      norm(lq + ".norm_1");
      p.gemm(xb, 256, lq + ".qkv", 768, 0, qkv, 768, 0);
      if (p.rc == PT_OK) {
        PtProfScope ps(e, s, PT_PROF_OTHER, 0, "tsr attention");
        if (x3) hipLaunchKernelGGL(attention_kernel<1>, dim3(ntiles, 8), dim3(64), 0, s, qkv, d_tiles, att);
        else hipLaunchKernelGGL(attention_kernel<0>, dim3(ntiles, 8), dim3(64), 0, s, qkv, d_tiles, att);
      }
      p.gemm(att, 256, lq + ".out", 256, 0, nullptr, 0, 0, x, 256, x);
      norm(lq + ".norm_2");
      p.gemm(xb, 256, lq + ".ff1", 2048, 1, hbuf, 2048, 0);
      p.gemm(hbuf, 2048, lq + ".ff2", 256, 0, nullptr, 0, 0, x, 256, x);
    }
    norm("");                                                         // fp32 stream -> bf16 for the decoder GEMMs
    p.gemm(xb, 256, q + ".dec0", 256, 1, att, 256, 0);
    p.gemm(att, 256, q + ".dec2", 64, 1, nullptr, 0, 0, out4, 8, nullptr, 8);
  };
  // attention output rows of pad tokens are never written: clear once so that no NaN bit pattern reaches a GEMM
  PT_HIP_CHECK(hipMemsetAsync(att, 0, (size_t)Npad * 256 * be, s));
  transformer("axis", x0, 256, layers[0], lg);
  {
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "tsr cvt logic");
    hipLaunchKernelGGL(cvt_logic_kernel, dim3((Npad * 32 + 255) / 256), dim3(256), 0, s, lg, l32, Npad, x3);
  }
  p.gemm(l32, 32, "stk.le0", 256, 1, le, 256, 0);
  p.gemm(le, 256, "stk.le2", 256, 1, cat, 512, 0);                    // channels [0, 256) of the concat (:385)
  transformer("stk", cat, 512, layers[1], sk);
  if (p.rc != PT_OK) return p.rc;
  hipLaunchKernelGGL(scatter4_kernel, dim3((N * 4 + 255) / 256), dim3(256), 0, s, lg, d_tok, N, d_logic);
  hipLaunchKernelGGL(scatter4_kernel, dim3((N * 4 + 255) / 256), dim3(256), 0, s, sk, d_tok, N, d_stacked);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

}  // namespace PT_FMT_NS
