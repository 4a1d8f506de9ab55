// mtl_decoder.hip -- the three autoregressive decoders of MtlTabNet (second half) as a KV-cached greedy
// loop over a BATCH of tables.
//
// Reference (model/table/mtl_tabnet/master_decoder.py): MtlTabNetDecoder.forward_test :503-517 -> greedy_forward :463-491 ->
// decode_test :354-461.  N = 3: two shared DecoderLayers (:99-114: x += self_attn(LN x); x += src_attn(LN x, feature);
// x += FFN(LN x)), then one more layer + LayerNorm + Linear per head: structure tokens (cls_layer / cls_fc), cell boxes (bbox_layer /
// bbox_fc + sigmoid), and -- after the structure loop ended -- the cell-content decoder (:387-459): every position whose structure
// token is '<td></td>' or '<td' becomes a row [emb_cell(token) + pe | x2 of that position] -> cell_input_fc -> DecoderLayerCell
// (keys / values of ONE table for all its cells, :117-144) -> LayerNorm -> cell_fc, greedy until ALL cells of the table emit
// <EOS> in the same step.  The reference is called with ONE table (processor_mtl_tabnet.py:84-89) and decodes the WHOLE prefix
// again at every step (O(L^2) layer passes).
//
// Here: rows are (position, sequence); a step computes only the NEW position of every sequence (tables in the structure loop,
// cells in the content loop), the self-attention reads keys / values of earlier positions from a cache [position][sequence][q|k|v]
// that the q/k/v GEMM of each step extends in place, and the source attention reads keys / values of the table's 3600 feature
// vectors that ONE GEMM produced for all five layers before the loop (`kv`).  Per-table semantics are the reference's at batch 1:
// a table stops at its own <EOS> (fin[]), the others go on.
//
// The one place where the reference is NOT causal: make_mask (:264-278) masks the QUERY rows of <PAD> tokens, so a <PAD> the
// decoder emitted itself attends uniformly to the whole current prefix, future positions included.  A cache cannot express that;
// the loop therefore watches for an emitted <PAD> (first_pad), rolls back to that step and continues in RE-DECODE mode, which runs
// the same kernels over positions 0..t of every sequence at every step -- the reference's own schedule.  Trained checkpoints never
// emit <PAD> (it is the loss's ignore_index); seeded random weights do, and tests/test_gpu_mtl.py covers both paths.
//
// Kernels: a Linear over <= 512 rows (every step of the KV-cached loops) is ONE launch of mtl_rowfused_kernel (LayerNorm in the prologue where the layer
// has one, K split over the waves of a workgroup, operands straight into the MFMA registers), over more rows (the key / value projection of the feature
// map, the re-decode mode) mtl_ln_kernel + a 1x1 GEMM on conv_igemm_kernel -- fp32 residual stream, hi/lo operands in BF16X3 like the rest of the engine;
// mtl_self_decode_kernel / mtl_cross_decode_kernel (one query per sequence: self- and source attention of the KV-cached loops as pure key / value
// streams, fp32 on the VALU); mtl_cross_attn_kernel (many queries per table -- re-decode mode, cell loop: the MFMA flash scheme of lore_processor.hip /
// cvit_model.hip with d = 64: a wave = up to 32 queries of ONE table x one head, the keys optionally split over several waves whose partial (max, sum,
// acc) triples mtl_cross_combine_kernel merges -- with one query per table and step, the split is what fills the chip); pick kernels (arg-max, soft-max
// probability, <EOS> / <PAD> bookkeeping, next-token embedding).  The structure-token and the box layer of a KV-cached step share every launch
// (run_layer, nb = 2: blockIdx.z picks the layer).
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "common.h"

namespace PT_FMT_NS {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8;
typedef __attribute__((ext_vector_type(16))) float af32x16;

constexpr int D = 512, HEADS = 8, DK = 64, NL = 5;      // d_model, heads, d_k, decoder layers with a source attention
constexpr int KVC = NL * 2 * D;                          // channels of the cross key / value tensor: [k_l | v_l] per layer
constexpr int PE_ROWS = 4096;

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
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ abf16x8 ld8(const bf16_t* p) { return *reinterpret_cast<const abf16x8*>(p); }
__device__ __forceinline__ uint4 ldu4(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
// eight bf16 values of a 16-byte load -> fp32, in memory order
__device__ __forceinline__ void unpack8(const uint4 u, float* f) {
  f[0] = a16lo_f32(u.x); f[1] = a16hi_f32(u.x);
  f[2] = a16lo_f32(u.y); f[3] = a16hi_f32(u.y);
  f[4] = a16lo_f32(u.z); f[5] = a16hi_f32(u.z);
  f[6] = a16lo_f32(u.w); f[7] = a16hi_f32(u.w);
}

// out_enc = PositionalEncoding(feat) (:182-188): f3 fp32 [n * hw, 512] (NHWC = the reference's view(b, c, h*w).permute(0, 2, 1))
// + pe[token] -> bf16 (hi | lo) rows; rows >= rows_valid are zero.  One wave per row.
__global__ __launch_bounds__(256) void mtl_feature_kernel(const float* __restrict__ f3, const float* __restrict__ pe, long long rows_valid,
                                                          long long rows_pad, int hw, bf16_t* __restrict__ out, int split) {
  a16_kernel_enter();
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows_pad) return;
  bf16_t* op = out + row * (split ? 2 * D : D);
  const int t = (int)(row % hw);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + 64 * k;
    put(op + c, D, split, row < rows_valid ? f3[row * D + c] + pe[(size_t)t * D + c] : 0.f);
  }
}

// Embeddings (* sqrt(d_model), folded into the table) + PositionalEncoding for positions [p0, p0 + npos) of every sequence.
// Row r = (p - p0) * Mp + s.  xout != null: fp32 rows [rows, 512] (structure decoder).  cin != null (cell-content decoder): bf16
// rows [rows, 1024] = [embedding + pe | x2 of the cell's structure position] -- the input of cell_input_fc (:417-418).
__global__ __launch_bounds__(256) void mtl_embed_kernel(const int* __restrict__ tok, const float* __restrict__ emb, const float* __restrict__ pe, int p0,
                                                        int npos, int Mp, int M, float* __restrict__ xout, bf16_t* __restrict__ cin,
                                                        const float* __restrict__ x2keep, const int* __restrict__ src_row, int split) {
  a16_kernel_enter();
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)npos * Mp) return;
  const int pi = (int)(row / Mp), s = (int)(row % Mp), p = p0 + pi;
  const bool valid = s < M;
  const int id = valid ? tok[(size_t)p * Mp + s] : 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + 64 * k;
    const float v = valid ? emb[(size_t)id * D + c] + pe[(size_t)p * D + c] : 0.f;
    if (xout) xout[row * D + c] = v;
    if (cin) {
      bf16_t* op = cin + row * (split ? 4 * D : 2 * D);
      put(op + c, 2 * D, split, v);
      put(op + D + c, 2 * D, split, valid ? x2keep[(size_t)src_row[s] * D + c] : 0.f);
    }
  }
}

// nn.LayerNorm(512) (biased variance, eps = 1e-5 inside the root) of fp32 rows -> bf16 (hi | lo).  One wave per row.
__global__ __launch_bounds__(256) void mtl_ln_kernel(const float* __restrict__ x, long long rows, const float* __restrict__ g,
                                                     const float* __restrict__ b, bf16_t* __restrict__ out, int split) {
  a16_kernel_enter();
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[8], s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v[k] = x[row * D + lane + 64 * k];
    s += v[k];
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) q += (v[k] - mean) * (v[k] - mean);
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
  bf16_t* op = out + row * (split ? 2 * D : D);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + 64 * k;
    put(op + c, D, split, (v[k] - mean) * rstd * g[c] + b[c]);
  }
}

// a [rows, 512] -> b, c (the cls and bbox layers start from the shared layers' output) and, for sequences still running at step
// p1, -> keep[p][s] (x2 of every position: what the cell-content decoder reads, :399)
__global__ __launch_bounds__(256) void mtl_fork_kernel(const float* __restrict__ a, float* __restrict__ b, float* __restrict__ c, float* __restrict__ keep,
                                                       const int* __restrict__ fin, int p0, int p1, int Mp, int M) {
  a16_kernel_enter();
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)(p1 - p0 + 1) * Mp * (D / 4);
  if (i >= total) return;
  const float4 v = reinterpret_cast<const float4*>(a)[i];
  reinterpret_cast<float4*>(b)[i] = v;
  reinterpret_cast<float4*>(c)[i] = v;
  const long long row = i / (D / 4);
  const int pi = (int)(row / Mp), s = (int)(row % Mp);
  if (s < M && fin[s] >= p1) reinterpret_cast<float4*>(keep)[((size_t)(p0 + pi) * Mp + s) * (D / 4) + (i % (D / 4))] = v;
}

// Self-attention of DecoderLayer (:109-110, self_attention :57-72) as a key / value stream: the query is position p of sequence s, keys and values are
// positions 0..p of the same sequence in the cache [position][Mp][q 512 | k 512 | v 512] (k already / 8, hi/lo: [hi 1536 | lo 1536]).  make_mask
// (:264-278): a query whose own token is <PAD> has every score replaced by -6.55e4, i.e. attends uniformly to ALL Lcur positions of the current prefix.
// A workgroup = the 8 heads of one row: wave = head, lane = (g = key within an octet, c = 8-channel piece), 16-byte loads of the cached k and v rows, one
// online soft-max stream per g merged at the end -- mtl_cross_decode_kernel's scheme over the cache (10 us at 160 cached positions; a wave per (row, head)
// walking the values one key and two bytes per lane at a time took 17).
template <int SPLIT>
__global__ __launch_bounds__(512) void mtl_self_decode_kernel(const bf16_t* __restrict__ cache, const int* __restrict__ tok, int pad, int p0, int Mp, int M,
                                                             int Lcur, bf16_t* __restrict__ att, long long cache_bs, long long att_bs) {
  a16_kernel_enter();
  cache += blockIdx.z * cache_bs;
  att += blockIdx.z * att_bs;
  const int pi = blockIdx.x / M, s = blockIdx.x % M, p = p0 + pi;
  const int head = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 3, c = lane & 7;
  constexpr int LO = 3 * D, cs = SPLIT ? 2 * LO : LO;
  const bool is_pad = tok[(size_t)p * Mp + s] == pad;
  const int nk = is_pad ? Lcur : p + 1;
  float qf[8];
  {
    const bf16_t* qp = cache + ((size_t)p * Mp + s) * cs + head * DK + 8 * c;
    unpack8(ldu4(qp), qf);
    if (SPLIT) {
      float t[8];
      unpack8(ldu4(qp + LO), t);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[j] += t[j];
    }
  }
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  float m = -INFINITY, l = 0.f;
  const bf16_t* kbase = cache + (size_t)s * cs + D + head * DK + 8 * c;
  const size_t pstride = (size_t)Mp * cs;
  for (int k0 = 0; k0 < nk; k0 += 64) {
    uint4 kh[8], vh[8], kl[8], vl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int key = k0 + 8 * i + g;
      const bf16_t* kp = kbase + (size_t)(key < nk ? key : nk - 1) * pstride;
      kh[i] = ldu4(kp);
      vh[i] = ldu4(kp + D);
      if (SPLIT) {
        kl[i] = ldu4(kp + LO);
        vl[i] = ldu4(kp + LO + D);
      }
    }
    float sc[8], mt = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float kf[8];
      unpack8(kh[i], kf);
      if (SPLIT) {
        float t[8];
        unpack8(kl[i], t);
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[j] += t[j];
      }
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d = fmaf(qf[j], kf[j], d);
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      d += __shfl_xor(d, 4);
      if (is_pad) d = -6.55e4f;
      sc[i] = k0 + 8 * i + g < nk ? d : -INFINITY;
      mt = fmaxf(mt, sc[i]);
    }
    if (mt > -INFINITY) {
      const float mn = fmaxf(m, mt);
      const float scale = expf(m - mn);
      l *= scale;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] *= scale;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float pv = expf(sc[i] - mn);
        l += pv;
        float vf[8];
        unpack8(vh[i], vf);
        if (SPLIT) {
          float t[8];
          unpack8(vl[i], t);
#pragma unroll
          for (int j = 0; j < 8; ++j) vf[j] += t[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(pv, vf[j], acc[j]);
      }
      m = mn;
    }
  }
  float mx = m;
  mx = fmaxf(mx, __shfl_xor(mx, 8));
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float w = m == -INFINITY ? 0.f : expf(m - mx);
  l *= w;
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] *= w;
#pragma unroll
  for (int sh = 8; sh < 64; sh <<= 1) {
    l += __shfl_xor(l, sh);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], sh);
  }
  if (g != 0) return;
  bf16_t* op = att + ((size_t)pi * Mp + s) * (SPLIT ? 2 * D : D) + head * DK + 8 * c;
#pragma unroll
  for (int j = 0; j < 8; ++j) put(op + j, D, SPLIT, acc[j] / l);
}

// Source attention (:111-112; MultiHeadAttentionCell :117-144 for the cell decoder: same arithmetic, keys of ONE table) for a tile
// of up to 32 queries of one table x one head x one slice of the keys, on the matrix cores.  tiles[i] = (table, first query row,
// queries, row stride).  q [rows, 512] (hi | lo), kv [n * hw, KVC] with the layer's keys (already / 8) at channel koff and values at
// koff + 512.  S^T (32 keys x 32 queries) = K Q^T over d = 64 (four v_mfma_f32_32x32x16_bf16); in the D layout a lane owns ONE
// query and 16 keys, so the online soft-max is in-lane + one exchange with lane ^ 32; O^T (64 x 32) += V^T P^T with P fed from the
// registers it is in.  nsplit == 1: normalised output rows -> att (hi | lo).  nsplit > 1: (max, sum) and the un-normalised
// accumulator of this key slice -> mlpart / opart for mtl_cross_combine_kernel.
template <int SPLIT>
__global__ __launch_bounds__(64) void mtl_cross_attn_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kv, int koff, const int4* __restrict__ tiles,
                                                            int hw, int keys_per_split, int nsplit, long long R, float* __restrict__ opart,
                                                            float* __restrict__ mlpart, bf16_t* __restrict__ att) {
  a16_kernel_enter();
  const int4 tile = tiles[blockIdx.x];
  const int head = blockIdx.y, z = blockIdx.z, lane = threadIdx.x;
  const int tab = tile.x, row0 = tile.y, cnt = tile.z, rstride = tile.w;
  constexpr int LOQ = D, qcs = SPLIT ? 2 * D : D, LOK = KVC, kcs = SPLIT ? 2 * KVC : KVC;
  const int col = lane & 31, half = lane >> 5;
  const long long qrow = row0 + (long long)(col < cnt ? col : cnt - 1) * rstride;
  const bf16_t* qp = q + qrow * qcs + head * DK + half * 8;
  abf16x8 qh[4], ql[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qh[s] = ld8(qp + 16 * s);
    if (SPLIT) ql[s] = ld8(qp + LOQ + 16 * s);
  }
  af32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const int kbeg = z * keys_per_split, kend = min(hw, kbeg + keys_per_split);
  const bf16_t* kbase = kv + (size_t)tab * hw * kcs + koff + head * DK;
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    const int krow = k0 + col;
    const bf16_t* kp = kbase + (size_t)(krow < kend ? krow : kend - 1) * kcs + half * 8;
    af32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const abf16x8 kh = ld8(kp + 16 * s);
      sc = mfma_32x32x16_a16(kh, qh[s], sc);
      if (SPLIT) {
        sc = mfma_32x32x16_a16(kh, ql[s], sc);
        sc = mfma_32x32x16_a16(ld8(kp + LOK + 16 * s), qh[s], sc);
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (key >= kend) sc[r] = -INFINITY;
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
        keys[j] = key < kend ? key : kend - 1;              // its probability is exactly 0
      }
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        abf16x8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bf16_t* vp = kbase + (size_t)keys[j] * kcs + D + db * 32 + col;     // A row = d
          vh[j] = __builtin_bit_cast(__bf16, vp[0]);
          if (SPLIT) vl[j] = __builtin_bit_cast(__bf16, vp[LOK]);
        }
        acc[db] = mfma_32x32x16_a16(vh, ph, acc[db]);
        if (SPLIT) {
          acc[db] = mfma_32x32x16_a16(vh, pl, acc[db]);
          acc[db] = mfma_32x32x16_a16(vl, ph, acc[db]);
        }
      }
    }
  }
  if (col >= cnt) return;
  if (nsplit == 1) {
    bf16_t* op = att + qrow * qcs + head * DK;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) put(op + db * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, D, SPLIT, acc[db][r] / l);
  } else {
    const size_t slot = ((size_t)z * R + qrow) * HEADS + head;
    float* op = opart + slot * DK;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) op[db * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = acc[db][r];
    if (half == 0) {
      mlpart[slot * 2] = m;
      mlpart[slot * 2 + 1] = l;
    }
  }
}

// Source attention of the KV-cached structure loop: ONE query per table and step, so the work is a stream over that table's keys and values and nothing
// else.  A workgroup = the 8 heads of one (table, key slice): wave = head, so the eight waves read the same 1 KB key rows and 1 KB value rows (the
// MFMA kernel above, launched per head, reads 128-byte pieces 10 KB apart, its values two bytes at a time).  lane = (g = key within an octet, c = 8-channel
// piece): 16-byte loads, 8 lanes cover the 128 bytes of a key's head slice; fp32 arithmetic on the VALU -- q . k over the lane's 8 channels, summed over the
// 8 lanes of the group; p * v accumulated per lane; hi + lo operands are added before the products in BF16X3 (exact in fp32) -- one independent online
// soft-max stream per g, merged at the end.  Output as mtl_cross_attn_kernel's: normalised rows (nsplit == 1) or (max, sum, accumulator) per key slice.
template <int SPLIT>
__global__ __launch_bounds__(512) void mtl_cross_decode_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kv, int koff, const int4* __restrict__ tiles,
                                                              int hw, int keys_per_split, int nsplit, long long R, float* __restrict__ opart,
                                                              float* __restrict__ mlpart, bf16_t* __restrict__ att, long long q_bs, long long part_bs,
                                                              long long ml_bs) {
  a16_kernel_enter();
  q += blockIdx.z * q_bs; att += blockIdx.z * q_bs; koff += blockIdx.z * 2 * D;      // blockIdx.z: second layer of a pair (the next key / value slot)
  opart += blockIdx.z * part_bs; mlpart += blockIdx.z * ml_bs;
  const int4 tile = tiles[blockIdx.x];
  const int head = threadIdx.x >> 6, lane = threadIdx.x & 63, z = blockIdx.y;
  const int g = lane >> 3, c = lane & 7;
  const int tab = tile.x;
  const long long qrow = tile.y;
  constexpr int LOQ = D, qcs = SPLIT ? 2 * D : D, LOK = KVC, kcs = SPLIT ? 2 * KVC : KVC;
  float qf[8];
  {
    const bf16_t* qp = q + qrow * qcs + head * DK + 8 * c;
    unpack8(ldu4(qp), qf);
    if (SPLIT) {
      float t[8];
      unpack8(ldu4(qp + LOQ), t);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[j] += t[j];
    }
  }
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int kbeg = z * keys_per_split, kend = min(hw, kbeg + keys_per_split);
  const bf16_t* kbase = kv + (size_t)tab * hw * kcs + koff + head * DK + 8 * c;
  for (int k0 = kbeg; k0 < kend; k0 += 64) {
    uint4 kh[8], vh[8], kl[8], vl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int key = k0 + 8 * i + g;
      const bf16_t* kp = kbase + (size_t)(key < kend ? key : kend - 1) * kcs;
      kh[i] = ldu4(kp);
      vh[i] = ldu4(kp + D);
      if (SPLIT) {
        kl[i] = ldu4(kp + LOK);
        vl[i] = ldu4(kp + LOK + D);
      }
    }
    float sc[8], mt = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float kf[8];
      unpack8(kh[i], kf);
      if (SPLIT) {
        float t[8];
        unpack8(kl[i], t);
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[j] += t[j];
      }
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d = fmaf(qf[j], kf[j], d);
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      d += __shfl_xor(d, 4);
      sc[i] = k0 + 8 * i + g < kend ? d : -INFINITY;
      mt = fmaxf(mt, sc[i]);
    }
    if (mt > -INFINITY) {                       // else: this group's keys of the chunk are all past the slice
      const float mn = fmaxf(m, mt);
      const float scale = expf(m - mn);         // m == -inf: 0
      l *= scale;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] *= scale;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float pv = expf(sc[i] - mn);      // masked keys: exp(-inf) = 0
        l += pv;
        float vf[8];
        unpack8(vh[i], vf);
        if (SPLIT) {
          float t[8];
          unpack8(vl[i], t);
#pragma unroll
          for (int j = 0; j < 8; ++j) vf[j] += t[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(pv, vf[j], acc[j]);
      }
      m = mn;
    }
  }
  // merge the eight streams (lane bits 3..5)
  float mx = m;
  mx = fmaxf(mx, __shfl_xor(mx, 8));
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float w = m == -INFINITY ? 0.f : expf(m - mx);
  l *= w;
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] *= w;
#pragma unroll
  for (int sh = 8; sh < 64; sh <<= 1) {
    l += __shfl_xor(l, sh);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], sh);
  }
  if (nsplit == 1) {
    if (g != 0) return;
    bf16_t* op = att + qrow * qcs + head * DK + 8 * c;
#pragma unroll
    for (int j = 0; j < 8; ++j) put(op + j, D, SPLIT, acc[j] / l);
    return;
  }
  if (g == 0) {
    const size_t slot = ((size_t)z * R + qrow) * HEADS + head;
    float* op = opart + slot * DK + 8 * c;
#pragma unroll
    for (int j = 0; j < 8; ++j) op[j] = acc[j];
    if (c == 0) {
      mlpart[slot * 2] = mx;
      mlpart[slot * 2 + 1] = l;
    }
  }
}

// ---- fp8 keys / values (pt_engine_set_mtl_kv_fp8, bf16 mode) --------------------------------------------------------------------------
// The KV-cached structure loop is a stream over the tables' keys and values (641 MB per layer and step for 87 tables at ~5 TB/s: 60 % of a
// step).  mtl_kv_fp8_kernel writes a second copy of the projected [rows, 5120] tensor as e4m3 bytes (x KV8_SCALE: the projections of LayerNorm-ed
// features are O(1), the scale keeps small values out of the subnormals), mtl_cross_decode8_kernel is mtl_cross_decode_kernel over that copy: 4 lanes
// per key (16 channels = one 16-byte load each), 16 independent soft-max streams per wave, hardware conversions (v_cvt_pk_f32_fp8).  Half the bytes;
// the drift against the bf16 keys / values is recorded in tests/test_gpu_mtl.py.  The cell loop and the re-decode mode read the bf16 tensor.
constexpr float KV8_SCALE = 8.f;

__global__ __launch_bounds__(256) void mtl_kv_fp8_kernel(const bf16_t* __restrict__ kv, long long n8, unsigned char* __restrict__ out) {
  a16_kernel_enter();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  float f[8];
  unpack8(ldu4(kv + i * 8), f);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(f[j] * KV8_SCALE, -448.f), 448.f);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  *reinterpret_cast<int2*>(out + i * 8) = make_int2(lo, hi);
}

// sixteen fp8 values of a 16-byte load -> fp32, in memory order
__device__ __forceinline__ void unpack16_fp8(const uint4 u, float* f) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const auto a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[k], false);
    const auto b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[k], true);
    f[4 * k] = a[0]; f[4 * k + 1] = a[1]; f[4 * k + 2] = b[0]; f[4 * k + 3] = b[1];
  }
}

__global__ __launch_bounds__(512) void mtl_cross_decode8_kernel(const bf16_t* __restrict__ q, const unsigned char* __restrict__ kv8, int koff,
                                                               const int4* __restrict__ tiles, int hw, int keys_per_split, int nsplit, long long R,
                                                               float* __restrict__ opart, float* __restrict__ mlpart, bf16_t* __restrict__ att,
                                                               long long q_bs, long long part_bs, long long ml_bs) {
  a16_kernel_enter();
  q += blockIdx.z * q_bs; att += blockIdx.z * q_bs; koff += blockIdx.z * 2 * D;
  opart += blockIdx.z * part_bs; mlpart += blockIdx.z * ml_bs;
  const int4 tile = tiles[blockIdx.x];
  const int head = threadIdx.x >> 6, lane = threadIdx.x & 63, z = blockIdx.y;
  const int g = lane >> 2, c = lane & 3;
  const int tab = tile.x;
  const long long qrow = tile.y;
  float qf[16];
  {
    const bf16_t* qp = q + qrow * D + head * DK + 16 * c;
    unpack8(ldu4(qp), qf);
    unpack8(ldu4(qp + 8), qf + 8);
#pragma unroll
    for (int j = 0; j < 16; ++j) qf[j] *= 1.f / KV8_SCALE;
  }
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int kbeg = z * keys_per_split, kend = min(hw, kbeg + keys_per_split);
  const unsigned char* kbase = kv8 + (size_t)tab * hw * KVC + koff + head * DK + 16 * c;
  for (int k0 = kbeg; k0 < kend; k0 += 128) {
    uint4 kr[8], vr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int key = k0 + 16 * i + g;
      const unsigned char* kp = kbase + (size_t)(key < kend ? key : kend - 1) * KVC;
      kr[i] = *reinterpret_cast<const uint4*>(kp);
      vr[i] = *reinterpret_cast<const uint4*>(kp + D);
    }
    float sc[8], mt = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float kf[16];
      unpack16_fp8(kr[i], kf);
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) d = fmaf(qf[j], kf[j], d);
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      sc[i] = k0 + 16 * i + g < kend ? d : -INFINITY;
      mt = fmaxf(mt, sc[i]);
    }
    if (mt > -INFINITY) {
      const float mn = fmaxf(m, mt);
      const float scale = expf(m - mn);
      l *= scale;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] *= scale;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float pv = expf(sc[i] - mn);
        l += pv;
        float vf[16];
        unpack16_fp8(vr[i], vf);
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = fmaf(pv, vf[j], acc[j]);
      }
      m = mn;
    }
  }
  float mx = m;
#pragma unroll
  for (int sh = 4; sh < 64; sh <<= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
  const float w = m == -INFINITY ? 0.f : expf(m - mx);
  l *= w;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] *= w * (1.f / KV8_SCALE);
#pragma unroll
  for (int sh = 4; sh < 64; sh <<= 1) {
    l += __shfl_xor(l, sh);
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] += __shfl_xor(acc[j], sh);
  }
  if (nsplit == 1) {
    if (g != 0) return;
    bf16_t* op = att + qrow * D + head * DK + 16 * c;
#pragma unroll
    for (int j = 0; j < 16; ++j) put(op + j, D, 0, acc[j] / l);
    return;
  }
  if (g == 0) {
    const size_t slot = ((size_t)z * R + qrow) * HEADS + head;
    float* op = opart + slot * DK + 16 * c;
#pragma unroll
    for (int j = 0; j < 16; ++j) op[j] = acc[j];
    if (c == 0) {
      mlpart[slot * 2] = mx;
      mlpart[slot * 2 + 1] = l;
    }
  }
}

// merges the key slices of mtl_cross_attn_kernel: out = sum_z e^(m_z - M) acc_z / sum_z e^(m_z - M) l_z.  thread = (row, channel)
__global__ __launch_bounds__(512) void mtl_cross_combine_kernel(const float* __restrict__ opart, const float* __restrict__ mlpart, int nsplit, long long R, int Mp,
                                                                int M, bf16_t* __restrict__ att, int split, long long att_bs, long long part_bs,
                                                                long long ml_bs) {
  a16_kernel_enter();
  att += blockIdx.z * att_bs; opart += blockIdx.z * part_bs; mlpart += blockIdx.z * ml_bs;
  const int pi = blockIdx.x / M, s = blockIdx.x % M, c = threadIdx.x, head = c >> 6;
  const long long row = (long long)pi * Mp + s;
  float mx = -INFINITY;
  for (int z = 0; z < nsplit; ++z) mx = fmaxf(mx, mlpart[(((size_t)z * R + row) * HEADS + head) * 2]);
  float num = 0.f, den = 0.f;
  for (int z = 0; z < nsplit; ++z) {
    const size_t slot = ((size_t)z * R + row) * HEADS + head;
    const float w = expf(mlpart[slot * 2] - mx);
    den += w * mlpart[slot * 2 + 1];
    num += w * opart[slot * DK + (c & 63)];
  }
  put(att + row * (split ? 2 * D : D) + c, D, split, num / den);
}

// Structure heads of positions [p0, p1] (decode_test :372-380, 461 + greedy_forward :476-490), one thread per row.  For sequences
// still running at step p1 (fin[s] >= p1): raw logits -> out_logits [M, T, ncls], sigmoid boxes -> out_boxes [M, T, 4], arg-max ->
// ids; the row of position p1 decides: <EOS> or the length limit ends the sequence (fin[s] = p1, i.e. p1 + 1 output positions),
// anything else is appended (tok[p1 + 1]); an appended <PAD> is reported through first_pad (see the header comment).
__global__ __launch_bounds__(256) void mtl_tag_pick_kernel(const float* __restrict__ lg, const float* __restrict__ bx, int p0, int p1, int Mp, int M, int ncls,
                                                           int ncls_p, int eos, int pad, int max_len, int T, int* __restrict__ tok, int* __restrict__ ids,
                                                           int* __restrict__ fin, int* __restrict__ first_pad, float* __restrict__ out_logits,
                                                           float* __restrict__ out_boxes) {
  a16_kernel_enter();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (p1 - p0 + 1) * M) return;
  const int pi = i / M, s = i % M, p = p0 + pi;
  const bool running = fin[s] >= p1;
  if (!running) {
    if (p == p1) tok[(size_t)(p1 + 1) * Mp + s] = eos;        // a valid id for the embedding of rows nobody reads
    return;
  }
  const float* r = lg + ((size_t)pi * Mp + s) * ncls_p;
  float best = r[0];
  int bi = 0;
  float* ol = out_logits + ((size_t)s * T + p) * ncls;
  ol[0] = best;
  for (int c = 1; c < ncls; ++c) {
    const float v = r[c];
    ol[c] = v;
    if (v > best) { best = v; bi = c; }
  }
  ids[(size_t)p * Mp + s] = bi;
  const float* b = bx + ((size_t)pi * Mp + s) * 8;
#pragma unroll
  for (int k = 0; k < 4; ++k) out_boxes[((size_t)s * T + p) * 4 + k] = 1.f / (1.f + expf(-b[k]));
  if (p == p1) {
    if (bi == eos || p1 == max_len) {
      fin[s] = p1;
      tok[(size_t)(p1 + 1) * Mp + s] = eos;
    } else {
      tok[(size_t)(p1 + 1) * Mp + s] = bi;
      if (bi == pad) atomicMin(first_pad, p1);
    }
  }
}

// Cell-content head of positions [p0, p1], one wave per row (cell): soft-max probability of the arg-max (what tensor2idx_cell reads,
// master_convertor.py:551-584) -> cell_ids / cell_prob [Mc, Tc], raw logits -> cell_logits [Mc, Tc, ncell] when requested; the arg-max
// of position p1 -> nxt[cell].  Written only for cells whose table is still running at step p1.
__global__ __launch_bounds__(256) void mtl_cell_pick_kernel(const float* __restrict__ lg, int p0, int p1, int Mp, int M, int ncell, int ncell_p, int Tc,
                                                            const int* __restrict__ cell_tab, const int* __restrict__ finc, int* __restrict__ nxt,
                                                            int* __restrict__ cell_ids, float* __restrict__ cell_prob, float* __restrict__ cell_logits) {
  a16_kernel_enter();
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= (long long)(p1 - p0 + 1) * M) return;
  const int pi = (int)(w / M), s = (int)(w % M), p = p0 + pi;
  if (finc[cell_tab[s]] < p1) return;
  const float* r = lg + ((size_t)pi * Mp + s) * ncell_p;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < ncell; c += 64) {
    const float v = r[c];
    if (cell_logits) cell_logits[((size_t)s * Tc + p) * ncell + c] = v;
    if (v > best) { best = v; bi = c; }
  }
#pragma unroll
  for (int mm = 32; mm > 0; mm >>= 1) {
    const float ob = __shfl_xor(best, mm);
    const int oi = __shfl_xor(bi, mm);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  float sum = 0.f;
  for (int c = lane; c < ncell; c += 64) sum += expf(r[c] - best);
  sum = wave_sum(sum);
  if (lane == 0) {
    cell_ids[(size_t)s * Tc + p] = bi;
    cell_prob[(size_t)s * Tc + p] = 1.f / sum;
    if (p == p1) nxt[s] = bi;
  }
}

// One workgroup per table after the pick of step p1 (:446-455): the content loop of a table ends when ALL its cells emitted <EOS> in
// this step, or at the length limit; otherwise every cell's arg-max is appended.
__global__ __launch_bounds__(256) void mtl_cell_next_kernel(const int* __restrict__ tab_first, const int* __restrict__ nxt, int p1, int Mp, int eos, int pad,
                                                            int max_len, int* __restrict__ tok, int* __restrict__ finc, int* __restrict__ first_pad) {
  a16_kernel_enter();
  __shared__ int cnt;
  const int b = blockIdx.x, c0 = tab_first[b], c1 = tab_first[b + 1];
  const bool running = finc[b] >= p1;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  if (running) {
    int mine = 0;
    for (int c = c0 + threadIdx.x; c < c1; c += blockDim.x) mine += nxt[c] == eos;
    if (mine) atomicAdd(&cnt, mine);
  }
  __syncthreads();
  const bool stop = !running || cnt == c1 - c0 || p1 == max_len;
  for (int c = c0 + threadIdx.x; c < c1; c += blockDim.x) {
    const int v = stop ? eos : nxt[c];
    tok[(size_t)(p1 + 1) * Mp + c] = v;
    if (!stop && v == pad) atomicMin(first_pad, p1);
  }
  if (running && stop && threadIdx.x == 0) finc[b] = p1;
}

__global__ void mtl_fill_kernel(int* __restrict__ p, int n, int v) {
  a16_kernel_enter();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// roll back to step `step`: sequences that ended later are running again
__global__ void mtl_rollback_kernel(int* __restrict__ fin, int n, int step) {
  a16_kernel_enter();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && fin[i] > step) fin[i] = INT_MAX;
}

// TableResize(keep_ratio, long_size) (lgpma_preprocess.py:1067-1093): Python floats = IEEE double; int() truncates
__host__ __device__ inline void mtl_resized(int w, int h, int size, int* ow, int* oh) {
  double fw = (double)w, fh = (double)h;
  if (fw < fh) {
    fw = (double)size / fh * fw;
    fh = (double)size;
  } else {
    fh = (double)size / fw * fh;
    fw = (double)size;
  }
  int iw = (int)fw, ih = (int)fh;
  *ow = iw < 1 ? 1 : iw;       // cv2.resize raises on an empty size; a 1-pixel line is the nearest well-defined answer
  *oh = ih < 1 ? 1 : ih;
}

struct RCoef {
  int s0, s1, a0, a1;
};
// OpenCV's 8-bit INTER_LINEAR coefficients (see det_kernels.hip: resize_coef)
__device__ __forceinline__ RCoef mtl_coef(int d, double scale, int ssize, bool clamp_frac) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  RCoef c;
  if (clamp_frac) {
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  }
  c.a0 = (int)rintf((1.f - f) * 2048.f);
  c.a1 = (int)rintf(f * 2048.f);
  int t0 = s, t1 = s + 1;
  c.s0 = t0 < 0 ? 0 : (t0 >= ssize ? ssize - 1 : t0);
  c.s1 = t1 < 0 ? 0 : (t1 >= ssize ? ssize - 1 : t1);
  return c;
}

// TableResize + TablePad + ToTensorOCR + NormalizeOCR of one table crop per blockIdx.y: out bf16 [n, size, size, 32] (hi | lo)
__global__ __launch_bounds__(256) void mtl_preprocess_kernel(const uint8_t* __restrict__ pages, int ph, int pw, const pt_tsr_table* __restrict__ tabs, int size,
                                                             bf16_t* __restrict__ out, int split) {
  a16_kernel_enter();
  const int b = blockIdx.y;
  const pt_tsr_table t = tabs[b];
  int nw, nh;
  mtl_resized(t.crop_w, t.crop_h, size, &nw, &nh);
  const double sx = (double)t.crop_w / nw, sy = (double)t.crop_h / nh;
  const bool area2 = (t.crop_w == 2 * nw) && (t.crop_h == 2 * nh);
  const uint8_t* src = pages + ((size_t)t.page * ph + t.y0) * pw * 3 + (size_t)t.x0 * 3;
  const size_t pitch = (size_t)pw * 3;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < size * size; i += gridDim.x * blockDim.x) {
    const int x = i % size, y = i / size;
    float o[3] = {-1.f, -1.f, -1.f};                      // pad value 0: (0 - 0.5) / 0.5
    if (x < nw && y < nh) {
      int v[3];
      if (t.crop_w == nw && t.crop_h == nh) {
        const uint8_t* p = src + y * pitch + x * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
      } else if (area2) {
        const uint8_t* p0 = src + (size_t)(2 * y) * pitch + 2 * x * 3;
        const uint8_t* p1 = p0 + pitch;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2;
      } else {
        const RCoef cx = mtl_coef(x, sx, t.crop_w, true), cy = mtl_coef(y, sy, t.crop_h, false);
        const uint8_t* r0 = src + (size_t)cy.s0 * pitch;
        const uint8_t* r1 = src + (size_t)cy.s1 * pitch;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int S0 = r0[cx.s0 * 3 + c] * cx.a0 + r0[cx.s1 * 3 + c] * cx.a1;
          const int S1 = r1[cx.s0 * 3 + c] * cx.a0 + r1[cx.s1 * 3 + c] * cx.a1;
          int r = (((cy.a0 * (S0 >> 4)) >> 16) + ((cy.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
          v[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = ((float)v[c] / 255.f - 0.5f) / 0.5f;       // to_tensor: / 255; normalize: (x - mean) / std
    }
    // 64 bytes per pixel (three values + the zero padding of conv1's 32-channel chunk) as four 16-byte stores; the lo half likewise
    uint4* op = reinterpret_cast<uint4*>(out + ((size_t)b * size * size + i) * (split ? 64 : 32));
    const uint4 zero = {0u, 0u, 0u, 0u};
    const uint4 hi = {pack_a16x2(o[0], o[1]), pack_a16x2(o[2], 0.f), 0u, 0u};
    op[0] = hi; op[1] = zero; op[2] = zero; op[3] = zero;
    if (split) {
      const uint4 lo = {pack_a16x2(o[0] - a16lo_f32(hi.x), o[1] - a16hi_f32(hi.x)), pack_a16x2(o[2] - a16lo_f32(hi.y), 0.f), 0u, 0u};
      op[4] = lo; op[5] = zero; op[6] = zero; op[7] = zero;
    }
  }
}

struct DevBuf {
  char* base = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return PT_OK;
    PT_HIP_CHECK(hipDeviceSynchronize());
    if (base) PT_HIP_CHECK(hipFree(base));
    base = nullptr;
    cap = 0;
    PT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&base), bytes + (1u << 20)));
    cap = bytes + (1u << 20);
    return PT_OK;
  }
  void release() {
    if (base) (void)hipFree(base);
    base = nullptr;
    cap = 0;
  }
};

struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off = (off + bytes + 255) & ~size_t(255);
    return o;
  }
};

// ---- skinny GEMM of the decoding loops -------------------------------------------------------------------------------------------------
// A decoding step multiplies M = 128 .. 256 rows (one per table / cell) by 512 x 512 .. 2048 x 512 weights: on conv_igemm_kernel that is 8 workgroups
// walking 16 .. 192 K-chunks one global-load latency at a time (18 .. 55 us per Linear in bf16), and the KV-cached loops are a chain of dependent
// few-microsecond kernels, so the launch count IS the step time.  mtl_rowfused_kernel does a Linear in ONE launch: workgroup = 32 rows x one 64-channel
// weight tile; its four waves split K (a wave = kc / 4 chunks, in groups of ROWGEMM_CH with the next group's loads in flight; operands straight from global
// memory into the MFMA registers), their fp32 partial tiles meet in LDS and are added in wave order (deterministic), then bias, fp32 residual, ReLU and
// the store of Ctx::gemm's contract (bf16 hi | lo rows or fp32 rows, n_valid).  Weights are read in conv_igemm's tiling ([N/64][chunks][64][32]; BF16X3:
// chunks = [hi | hi | lo] against the activations' [hi | lo | hi]).  MFMA roles: A = weights (M = output channel), B = activations (N = row), so a lane
// owns four consecutive channels of one row.  LNIN: the rows come as the fp32 residual stream and nn.LayerNorm(512) runs in the prologue -- a wave owns the
// 128-channel slice its K-share needs, so the values are read once and normalised in registers (two-pass variance; row statistics cross the waves through
// LDS); BF16X3 builds the hi and lo operands there.  Every weight tile of the layer recomputes the statistics of its 32 rows (64 KB from L2) -- cheaper than
// the launch it replaces.  (Round 4's split-K pair of kernels + mtl_ln_kernel: 85 launches per step; this: 43; with the layer pairing of run_layer: 31.)
constexpr int ROWGEMM_CH = 4;
constexpr int RF_LD = 68;      // floats per row of the partial tiles in LDS

// one Linear's operands; the kernel takes two sets and blockIdx.z picks (the structure-token and the box layer of a step are independent and equal in
// shape: one launch per Linear for both)
struct RowB {
  const bf16_t* x; const float* xf; const float* lng; const float* lnb; const bf16_t* w; const float* bias;
  bf16_t* out; float* out_f32; const float* res_f32; int f32_cs; int n_valid;
};

template <int SPLIT, int LNIN>
__global__ __launch_bounds__(256) void mtl_rowfused_kernel(const RowB pa, const RowB pb, int cin, int N, int relu, int out_cs) {
#define RB(f) (blockIdx.z ? pb.f : pa.f)
  const bf16_t* __restrict__ x = RB(x);
  const float* __restrict__ xf = RB(xf);
  const float* __restrict__ lng = RB(lng);
  const float* __restrict__ lnb = RB(lnb);
  const bf16_t* __restrict__ w = RB(w);
  const float* __restrict__ bias = RB(bias);
  bf16_t* __restrict__ out = RB(out);
  float* out_f32 = RB(out_f32);
  const float* res_f32 = RB(res_f32);
  const int f32_cs = RB(f32_cs), n_valid = RB(n_valid);
#undef RB
  a16_kernel_enter();
  __shared__ float s_red[4][32][RF_LD];
  __shared__ float s_stat[2][4][32];
  const int nt = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = lane & 31, half = lane >> 5;
  const long long row = (long long)blockIdx.y * 32 + col;
  const int kc1 = cin >> 5, kc = SPLIT ? 3 * kc1 : kc1;
  const bf16_t* wt = w + (size_t)nt * kc * (64 * 32) + col * 32 + 8 * half;
  af32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  auto load_a = [&](abf16x8 (&a)[ROWGEMM_CH][2][2], int wc0) {
#pragma unroll
    for (int i = 0; i < ROWGEMM_CH; ++i) {
      const bf16_t* wp = wt + (size_t)(wc0 + i) * (64 * 32);
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        a[i][0][st] = ld8(wp + 16 * st);
        a[i][1][st] = ld8(wp + 32 * 32 + 16 * st);
      }
    }
  };
  auto mma = [&](const abf16x8 (&a)[ROWGEMM_CH][2][2], const abf16x8 (&b)[ROWGEMM_CH][2]) {
#pragma unroll
    for (int i = 0; i < ROWGEMM_CH; ++i)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        acc[0] = mfma_32x32x16_a16(a[i][0][st], b[i][st], acc[0]);
        acc[1] = mfma_32x32x16_a16(a[i][1][st], b[i][st], acc[1]);
      }
  };
  abf16x8 a0[ROWGEMM_CH][2][2], a1[ROWGEMM_CH][2][2];
  if constexpr (LNIN) {
    // cin == D: chunks 4 * wave .. 4 * wave + 3 of every pass
    load_a(a0, 4 * wave);
    float v[ROWGEMM_CH][2][8];
    const float* xr = xf + row * D + 128 * wave + 8 * half;
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < ROWGEMM_CH; ++i)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const float4 p = *reinterpret_cast<const float4*>(xr + 32 * i + 16 * st), q = *reinterpret_cast<const float4*>(xr + 32 * i + 16 * st + 4);
        float* vv = v[i][st];
        vv[0] = p.x; vv[1] = p.y; vv[2] = p.z; vv[3] = p.w; vv[4] = q.x; vv[5] = q.y; vv[6] = q.z; vv[7] = q.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) sm += vv[e];
      }
    sm += __shfl_xor(sm, 32);
    if (half == 0) s_stat[0][wave][col] = sm;
    __syncthreads();
    const float mean = (((s_stat[0][0][col] + s_stat[0][1][col]) + s_stat[0][2][col]) + s_stat[0][3][col]) / (float)D;
    float qd = 0.f;
#pragma unroll
    for (int i = 0; i < ROWGEMM_CH; ++i)
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int e = 0; e < 8; ++e) qd += (v[i][st][e] - mean) * (v[i][st][e] - mean);
    qd += __shfl_xor(qd, 32);
    if (half == 0) s_stat[1][wave][col] = qd;
    __syncthreads();
    const float rstd = 1.f / sqrtf((((s_stat[1][0][col] + s_stat[1][1][col]) + s_stat[1][2][col]) + s_stat[1][3][col]) / (float)D + 1e-5f);
    abf16x8 bh[ROWGEMM_CH][2], bl[ROWGEMM_CH][2];
#pragma unroll
    for (int i = 0; i < ROWGEMM_CH; ++i)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const int c0 = 128 * wave + 32 * i + 16 * st + 8 * half;
        const float4 g0 = *reinterpret_cast<const float4*>(lng + c0), g1 = *reinterpret_cast<const float4*>(lng + c0 + 4);
        const float4 e0 = *reinterpret_cast<const float4*>(lnb + c0), e1 = *reinterpret_cast<const float4*>(lnb + c0 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, ee[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (v[i][st][e] - mean) * rstd * gg[e] + ee[e];
        uint4 hi = {pack_a16x2(y[0], y[1]), pack_a16x2(y[2], y[3]), pack_a16x2(y[4], y[5]), pack_a16x2(y[6], y[7])};
        bh[i][st] = __builtin_bit_cast(abf16x8, hi);
        if constexpr (SPLIT) {
          uint4 lo = {pack_a16x2(y[0] - a16lo_f32(hi.x), y[1] - a16hi_f32(hi.x)), pack_a16x2(y[2] - a16lo_f32(hi.y), y[3] - a16hi_f32(hi.y)),
                      pack_a16x2(y[4] - a16lo_f32(hi.z), y[5] - a16hi_f32(hi.z)), pack_a16x2(y[6] - a16lo_f32(hi.w), y[7] - a16hi_f32(hi.w))};
          bl[i][st] = __builtin_bit_cast(abf16x8, lo);
        }
      }
    if constexpr (SPLIT) {      // weights [hi | hi | lo] against the row's [hi | lo | hi]
      load_a(a1, kc1 + 4 * wave);
      mma(a0, bh);
      load_a(a0, 2 * kc1 + 4 * wave);
      mma(a1, bl);
      mma(a0, bh);
    } else {
      mma(a0, bh);
    }
  } else {
    const int cpw = kc >> 2, ng = cpw / ROWGEMM_CH;      // host: kc % 16 == 0
    const bf16_t* xr = x + row * (SPLIT ? 2 * cin : cin) + 8 * half;
    abf16x8 b0[ROWGEMM_CH][2], b1[ROWGEMM_CH][2];
    auto load_b = [&](abf16x8 (&b)[ROWGEMM_CH][2], int c0) {
#pragma unroll
      for (int i = 0; i < ROWGEMM_CH; ++i) {
        const int c = c0 + i;
        const int xc = !SPLIT ? c : (c < 2 * kc1 ? c : c - 2 * kc1);      // [hi | lo] rows against weight chunks [hi | hi | lo]
#pragma unroll
        for (int st = 0; st < 2; ++st) b[i][st] = ld8(xr + xc * 32 + 16 * st);
      }
    };
    const int cw = wave * cpw;
    load_a(a0, cw); load_b(b0, cw);
    for (int g = 0; g < ng; g += 2) {
      if (g + 1 < ng) { load_a(a1, cw + (g + 1) * ROWGEMM_CH); load_b(b1, cw + (g + 1) * ROWGEMM_CH); }
      mma(a0, b0);
      if (g + 2 < ng) { load_a(a0, cw + (g + 2) * ROWGEMM_CH); load_b(b0, cw + (g + 2) * ROWGEMM_CH); }
      if (g + 1 < ng) mma(a1, b1);
    }
  }
  // partial tiles -> LDS ([wave][row][channel]), summed in wave order by (row, eight channels) threads
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 t = {acc[h][4 * g], acc[h][4 * g + 1], acc[h][4 * g + 2], acc[h][4 * g + 3]};
      *reinterpret_cast<float4*>(&s_red[wave][col][h * 32 + 8 * g + 4 * half]) = t;
    }
  __syncthreads();
  const int r = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8, n0 = nt * 64 + c8;
  if (n_valid && n0 >= n_valid) return;
  const long long grow = (long long)blockIdx.y * 32 + r;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = s_red[0][r][c8 + e];
#pragma unroll
  for (int z = 1; z < 4; ++z)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += s_red[z][r][c8 + e];
  const bool second = !n_valid || n0 + 4 < n_valid;      // n_valid % 4 == 0
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] += bias[n0 + e];
  if (second)
#pragma unroll
    for (int e = 4; e < 8; ++e) o[e] += bias[n0 + e];
  if (res_f32) {
    const float* rp = res_f32 + grow * f32_cs + n0;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] += rp[e];
    if (second)
#pragma unroll
      for (int e = 4; e < 8; ++e) o[e] += rp[e];
  }
  if (relu)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaxf(o[e], 0.f);
  if (out_f32) {
    float* op = out_f32 + grow * f32_cs + n0;
    *reinterpret_cast<float4*>(op) = float4{o[0], o[1], o[2], o[3]};
    if (second) *reinterpret_cast<float4*>(op + 4) = float4{o[4], o[5], o[6], o[7]};
  } else {
    bf16_t* op = out + grow * (SPLIT ? 2 * out_cs : out_cs) + n0;
    const uint4 hi = {pack_a16x2(o[0], o[1]), pack_a16x2(o[2], o[3]), pack_a16x2(o[4], o[5]), pack_a16x2(o[6], o[7])};
    *reinterpret_cast<uint2*>(op) = uint2{hi.x, hi.y};
    if (second) *reinterpret_cast<uint2*>(op + 4) = uint2{hi.z, hi.w};
    if constexpr (SPLIT) {
      const uint2 l0 = {pack_a16x2(o[0] - a16lo_f32(hi.x), o[1] - a16hi_f32(hi.x)), pack_a16x2(o[2] - a16lo_f32(hi.y), o[3] - a16hi_f32(hi.y))};
      const uint2 l1 = {pack_a16x2(o[4] - a16lo_f32(hi.z), o[5] - a16hi_f32(hi.z)), pack_a16x2(o[6] - a16lo_f32(hi.w), o[7] - a16hi_f32(hi.w))};
      *reinterpret_cast<uint2*>(op + out_cs) = l0;
      if (second) *reinterpret_cast<uint2*>(op + out_cs + 4) = l1;
    }
  }
}

// what survives between pt_tsr_mtl_structure and pt_tsr_mtl_cells
struct MtlState {
  DevBuf persist, work, cwork;
  int n = 0, hw = 0, Mp = 0, T = 0, x3 = 0;
  size_t o_kv8 = 0;             // fp8 copy of the keys / values (0 bytes when the option is off)
  size_t o_kv = 0, o_keep = 0, o_tok = 0, o_ids = 0, o_fin = 0;
  std::vector<int> lens;        // output positions per table
  std::vector<int> cell_tab;    // table of every cell, cells ordered by (table, position)
  std::vector<int> cell_src;    // row of x2keep ( = position * Mp + table) of every cell
  std::vector<int> tab_first;   // [n + 1]
  bool structure_done = false;
  int* h_poll = nullptr;        // pinned
};

struct Ctx {
  pt_engine* e;
  const PtModel* m;
  hipStream_t s;
  int x3, mul, rc;
  bool rowfused = true;        // false: every Linear on conv_igemm_kernel + mtl_ln_kernel
  const PtTensor* get(const std::string& n) {
    const PtTensor* t = m->find(n);
    if (!t && rc == PT_OK) {
      pt_set_error("MtlTabNet decoder weight blob lacks tensor '%s'", n.c_str());
      rc = PT_ERR_FORMAT;
    }
    return t;
  }
  const float* F(const std::string& n) {
    const PtTensor* t = get(n);
    return t ? reinterpret_cast<const float*>(t->d_ptr) : nullptr;
  }
  // y = x W^T + b over `rows` rows (multiple of 128): x bf16 [rows, cin] (hi | lo) -> bf16 [rows, out_cs] or fp32 [rows, f32_cs]
  void gemm(const bf16_t* x, long long rows, int cin, const std::string& q, int N, int relu, bf16_t* out, int out_cs, float* out_f32 = nullptr,
            int f32_cs = 0, const float* res_f32 = nullptr, int nv = 0) {
    const PtTensor* w = get(q + (x3 ? ".w3" : ".w"));
    const PtTensor* b = get(q + ".b");
    if (rc != PT_OK) return;
    if (fusable(rows, cin, N, out_f32, f32_cs, nv, out_cs)) {
      Lin l{x, nullptr, "", q, out, out_f32, res_f32, f32_cs, nv};
      lin(1, &l, rows, cin, N, relu, out_cs, nullptr);
      return;
    }
    ConvDesc c;
    c.in = x; c.B = 1; c.H = (int)(rows / 32); c.W = 32; c.Cin = cin;
    c.w = reinterpret_cast<const bf16_t*>(w->d_ptr); c.bias = reinterpret_cast<const float*>(b->d_ptr);
    c.N = N; c.ks = 1; c.stride = 1; c.relu = relu; c.split = x3; c.n_valid = nv;
    if (out_f32) {
      c.out_f32 = out_f32; c.out_cstride = f32_cs; c.res_f32 = res_f32;
    } else {
      c.out = out; c.out_cstride = out_cs * mul; c.out_coff = 0; c.out_lo_off = out_cs;
    }
    const int r = pt_launch_conv(e, c, s);
    if (r != PT_OK) rc = r;
  }
  // one Linear: bf16 rows `x`, or (xf != null) LayerNorm `lnq` of the fp32 rows xf [rows, 512]; weights `q`; bf16 rows `out` or fp32 rows `out_f32` (+ res)
  struct Lin {
    const bf16_t* x; const float* xf; std::string lnq, q;
    bf16_t* out; float* out_f32; const float* res; int f32_cs, nv;
  };
  // nb (1 or 2) Linears of one shape: ONE mtl_rowfused_kernel launch when the shape allows, else one after the other on the general kernels
  void lin(int nb, const Lin* l, long long rows, int cin, int N, int relu, int out_cs, bf16_t* xb) {
    bool fuse = true;
    for (int b = 0; b < nb; ++b) fuse = fuse && fusable(rows, cin, N, l[b].out_f32, l[b].f32_cs, l[b].nv, out_cs) && (l[b].xf != nullptr) == (l[0].xf != nullptr);
    if (!fuse) {
      for (int b = 0; b < nb; ++b) {
        if (l[b].xf) gemm_ln(l[b].xf, rows, l[b].lnq, xb, l[b].q, N, relu, l[b].out, out_cs, l[b].out_f32, l[b].f32_cs, l[b].nv);
        else gemm(l[b].x, rows, cin, l[b].q, N, relu, l[b].out, out_cs, l[b].out_f32, l[b].f32_cs, l[b].res, l[b].nv);
      }
      return;
    }
    RowB rb[2];
    for (int b = 0; b < nb; ++b) {
      const PtTensor* w = get(l[b].q + (x3 ? ".w3" : ".w"));
      const PtTensor* bi = get(l[b].q + ".b");
      const float *g = nullptr, *be = nullptr;
      if (l[b].xf) { g = F(l[b].lnq + ".g"); be = F(l[b].lnq + ".b"); }
      if (rc != PT_OK) return;
      rb[b] = RowB{l[b].x, l[b].xf, g, be, reinterpret_cast<const bf16_t*>(w->d_ptr), reinterpret_cast<const float*>(bi->d_ptr), l[b].out, l[b].out_f32,
                   l[b].res, l[b].f32_cs, l[b].nv};
    }
    if (nb == 1) rb[1] = rb[0];
    PtProfScope ps(e, s, PT_PROF_OTHER, 0, "mtl row gemm");
    const dim3 grid(N / 64, (unsigned)(rows / 32), nb);
    const bool lnin = l[0].xf != nullptr;
    if (x3 && lnin) hipLaunchKernelGGL((mtl_rowfused_kernel<1, 1>), grid, dim3(256), 0, s, rb[0], rb[1], cin, N, relu, out_cs);
    else if (x3) hipLaunchKernelGGL((mtl_rowfused_kernel<1, 0>), grid, dim3(256), 0, s, rb[0], rb[1], cin, N, relu, out_cs);
    else if (lnin) hipLaunchKernelGGL((mtl_rowfused_kernel<0, 1>), grid, dim3(256), 0, s, rb[0], rb[1], cin, N, relu, out_cs);
    else hipLaunchKernelGGL((mtl_rowfused_kernel<0, 0>), grid, dim3(256), 0, s, rb[0], rb[1], cin, N, relu, out_cs);
  }
  // one launch per Linear (mtl_rowfused_kernel) for the KV-cached loops' row counts; PT_MTL_ROWFUSED=0: conv_igemm_kernel + mtl_ln_kernel
  bool fusable(long long rows, int cin, int N, const float* out_f32, int f32_cs, int nv, int out_cs) const {
    static const bool on = !(getenv("PT_MTL_ROWFUSED") && atoi(getenv("PT_MTL_ROWFUSED")) == 0);
    static const int skinny_rows = getenv("PT_MTL_ROWGEMM_MAX") ? atoi(getenv("PT_MTL_ROWGEMM_MAX")) : 512;
    return on && rowfused && rows <= skinny_rows && rows % 32 == 0 && N % 64 == 0 && ((x3 ? 3 : 1) * (cin / 32)) % 16 == 0 && cin % 32 == 0 &&
           (out_f32 ? f32_cs % 4 == 0 : out_cs % 8 == 0) && (nv == 0 || nv % 4 == 0);
  }
  // y = LayerNorm(x; lnq) W^T + b: x fp32 [rows, 512] (the residual stream).  `xb`: where the normalised rows go when the two cannot share a launch
  void gemm_ln(const float* x, long long rows, const std::string& lnq, bf16_t* xb, const std::string& q, int N, int relu, bf16_t* out, int out_cs,
               float* out_f32 = nullptr, int f32_cs = 0, int nv = 0) {
    if (fusable(rows, D, N, out_f32, f32_cs, nv, out_cs)) {
      Lin l{nullptr, x, lnq, q, out, out_f32, nullptr, f32_cs, nv};
      lin(1, &l, rows, D, N, relu, out_cs, xb);
      return;
    }
    ln(x, rows, lnq, xb);
    gemm(xb, rows, D, q, N, relu, out, out_cs, out_f32, f32_cs, nullptr, nv);
  }
  void ln(const float* x, long long rows, const std::string& q, bf16_t* out) {
    const float *g = F(q + ".g"), *b = F(q + ".b");
    if (rc != PT_OK) return;
    hipLaunchKernelGGL(mtl_ln_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, g, b, out, x3);
  }
};

// work buffers of one decoding loop (structure or cells): R rows
struct Work {
  long long R = 0;
  long long Rs = 0;      // rows in use this step = stride between the key-split slices of opart / mlpart (<= R)
  bf16_t *xb = nullptr, *qc = nullptr, *att = nullptr, *hb = nullptr;
  float *opart = nullptr, *mlpart = nullptr;
  int4* tiles = nullptr;
  int ntiles = 0, nsplit = 1, kps = 0;
  bool single = false;      // every tile is ONE query (the KV-cached structure loop): mtl_cross_decode_kernel
  // buffers of a second layer run in the same launches (structure-token + box layer of a KV-cached step): element strides from the first one's (0: none)
  long long bs_row = 0, bs_hb = 0, bs_opart = 0, bs_ml = 0;
};

struct Seqs {          // the sequences of one loop
  int M = 0, Mp = 0;   // sequences, padded to 128
  const int* tok = nullptr;
  int pad = 0;
  int ffp = 2048;
  int hw = 0;
  const unsigned char* kv8 = nullptr;      // fp8 copy of the cross keys / values (pt_engine_set_mtl_kv_fp8; null: off)
};

// One DecoderLayer over positions [p0, p1] of all sequences -- or (nb == 2) two independent layers of one shape in the same launches: the structure-token
// and the box layer of a KV-cached step.  Per layer: weights `q`, its place `slot` in the cross K / V tensor (the second layer's = slot + 1), x_in: the
// fp32 residual rows of those positions ((p - p0) * Mp + s) the layer starts from, x: where its residual stream lives from the first residual on
// (x == x_in: in place).  cache: the first layer's [position][Mp][1536] q/k/v cache, the second's cache_bs elements behind it.
struct LayerB { std::string q; const float* x_in; float* x; };

void run_layer(Ctx& c, int nb, const LayerB* lb, int slot, bf16_t* cache, long long cache_bs, const bf16_t* kv, const Seqs& S, const Work& W, int p0, int p1) {
  const int npos = p1 - p0 + 1;
  const long long rows = (long long)npos * S.Mp;
  const int mul = c.mul;
  const size_t slab_off = (size_t)p0 * S.Mp * 3 * D * mul;
  Ctx::Lin l[2];
  for (int b = 0; b < nb; ++b) l[b] = Ctx::Lin{nullptr, lb[b].x_in, lb[b].q + ".ln0", lb[b].q + ".qkv", cache + b * cache_bs + slab_off, nullptr, nullptr, 0, 0};
  c.lin(nb, l, rows, D, 3 * D, 0, 3 * D, W.xb);
  if (c.rc != PT_OK) return;
  {
    PtProfScope ps(c.e, c.s, PT_PROF_OTHER, 0, "mtl self attention");
    const dim3 sgrid(npos * S.M, 1, nb);
    if (c.x3) hipLaunchKernelGGL(mtl_self_decode_kernel<1>, sgrid, dim3(512), 0, c.s, cache, S.tok, S.pad, p0, S.Mp, S.M, p1 + 1, W.att, cache_bs, W.bs_row);
    else hipLaunchKernelGGL(mtl_self_decode_kernel<0>, sgrid, dim3(512), 0, c.s, cache, S.tok, S.pad, p0, S.Mp, S.M, p1 + 1, W.att, cache_bs, W.bs_row);
  }
  for (int b = 0; b < nb; ++b) l[b] = Ctx::Lin{W.att + b * W.bs_row, nullptr, "", lb[b].q + ".so", nullptr, lb[b].x, lb[b].x_in, D, 0};
  c.lin(nb, l, rows, D, D, 0, 0, nullptr);
  for (int b = 0; b < nb; ++b) l[b] = Ctx::Lin{nullptr, lb[b].x, lb[b].q + ".ln1", lb[b].q + ".cq", W.qc + b * W.bs_row, nullptr, nullptr, 0, 0};
  c.lin(nb, l, rows, D, D, 0, D, W.xb);
  if (c.rc != PT_OK) return;
  {
    PtProfScope ps(c.e, c.s, PT_PROF_OTHER, 0, "mtl source attention");
    const dim3 grid(W.ntiles, HEADS, W.nsplit), dgrid(W.ntiles, W.nsplit, nb);
    static const bool decode_kernel = !getenv("PT_MTL_CROSS_MFMA");
    if (W.single && S.kv8 && !c.x3) {
      hipLaunchKernelGGL(mtl_cross_decode8_kernel, dgrid, dim3(512), 0, c.s, W.qc, S.kv8, slot * 2 * D, W.tiles, S.hw, W.kps, W.nsplit, W.Rs, W.opart, W.mlpart, W.att,
                         W.bs_row, W.bs_opart, W.bs_ml);
    } else if (W.single && (decode_kernel || nb > 1)) {
      if (c.x3) hipLaunchKernelGGL(mtl_cross_decode_kernel<1>, dgrid, dim3(512), 0, c.s, W.qc, kv, slot * 2 * D, W.tiles, S.hw, W.kps, W.nsplit, W.Rs, W.opart, W.mlpart, W.att,
                                   W.bs_row, W.bs_opart, W.bs_ml);
      else hipLaunchKernelGGL(mtl_cross_decode_kernel<0>, dgrid, dim3(512), 0, c.s, W.qc, kv, slot * 2 * D, W.tiles, S.hw, W.kps, W.nsplit, W.Rs, W.opart, W.mlpart, W.att,
                              W.bs_row, W.bs_opart, W.bs_ml);
    } else if (c.x3) hipLaunchKernelGGL(mtl_cross_attn_kernel<1>, grid, dim3(64), 0, c.s, W.qc, kv, slot * 2 * D, W.tiles, S.hw, W.kps, W.nsplit, W.Rs, W.opart, W.mlpart, W.att);
    else hipLaunchKernelGGL(mtl_cross_attn_kernel<0>, grid, dim3(64), 0, c.s, W.qc, kv, slot * 2 * D, W.tiles, S.hw, W.kps, W.nsplit, W.Rs, W.opart, W.mlpart, W.att);
    if (W.nsplit > 1)
      hipLaunchKernelGGL(mtl_cross_combine_kernel, dim3(npos * S.M, 1, nb), dim3(512), 0, c.s, W.opart, W.mlpart, W.nsplit, W.Rs, S.Mp, S.M, W.att, c.x3, W.bs_row,
                         W.bs_opart, W.bs_ml);
  }
  for (int b = 0; b < nb; ++b) l[b] = Ctx::Lin{W.att + b * W.bs_row, nullptr, "", lb[b].q + ".co", nullptr, lb[b].x, lb[b].x, D, 0};
  c.lin(nb, l, rows, D, D, 0, 0, nullptr);
  for (int b = 0; b < nb; ++b) l[b] = Ctx::Lin{nullptr, lb[b].x, lb[b].q + ".ln2", lb[b].q + ".ff1", W.hb + b * W.bs_hb, nullptr, nullptr, 0, 0};
  c.lin(nb, l, rows, D, S.ffp, 1, S.ffp, W.xb);
  for (int b = 0; b < nb; ++b) l[b] = Ctx::Lin{W.hb + b * W.bs_hb, nullptr, "", lb[b].q + ".ff2", nullptr, lb[b].x, lb[b].x, D, 0};
  c.lin(nb, l, rows, S.ffp, D, 0, 0, nullptr);
}

void run_layer(Ctx& c, const std::string& q, int slot, float* x, bf16_t* cache, const bf16_t* kv, const Seqs& S, const Work& W, int p0, int p1) {
  const LayerB lb{q, x, x};
  run_layer(c, 1, &lb, slot, cache, 0, kv, S, W, p0, p1);
}

MtlState* state_of(pt_engine* e) {
  if (!e->mtl_state) e->mtl_state = new MtlState();
  return reinterpret_cast<MtlState*>(e->mtl_state);
}

int upload_tiles(std::vector<int4>& host, int4* dev, hipStream_t s) {
  PT_HIP_CHECK(hipMemcpyAsync(dev, host.data(), host.size() * sizeof(int4), hipMemcpyHostToDevice, s));
  PT_HIP_CHECK(hipStreamSynchronize(s));      // `host` is a stack vector
  return PT_OK;
}

int pick_split(int ntiles, int hw, int* kps) {
  // waves that stream their key slice with a handful of loads in flight (PT_MTL_SPLIT_TARGET waves per launch).  Measured on 87 tables x 3600 keys,
  // KV-cached loop: 2048 .. 16384 all within 3 % (1.43 .. 1.49 ms per step): at 641 MB per launch the stream runs at ~5 TB/s either way
  static int target = -1;
  if (target < 0) { const char* ev = getenv("PT_MTL_SPLIT_TARGET"); target = ev ? atoi(ev) : 4096; }
  int ns = target / (ntiles * HEADS > 0 ? ntiles * HEADS : 1);
  if (ns < 1) ns = 1;
  if (ns > 16) ns = 16;
  int per = ((hw + ns - 1) / ns + 31) / 32 * 32;
  ns = (hw + per - 1) / per;
  *kps = per;
  return ns;
}

struct Meta {
  int ncls, ncell, sos, eos, pad, max_len, sos_c, eos_c, pad_c, max_len_c, tag0, tag1, ffp;
};

int read_meta(Ctx& c, Meta* mt) {
  const PtTensor* t = c.get("meta");
  if (c.rc != PT_OK) return c.rc;
  PT_REQUIRE(t->nbytes >= sizeof(Meta), "MtlTabNet decoder blob: short meta tensor");
  {
    std::vector<int32_t>& hw = const_cast<PtModel*>(c.m)->host_words["meta"];      // read from the device once per loaded blob
    if (hw.size() != sizeof(Meta) / 4) {
      hw.assign(sizeof(Meta) / 4, 0);
      PT_HIP_CHECK(hipMemcpy(hw.data(), t->d_ptr, sizeof(Meta), hipMemcpyDeviceToHost));
    }
    memcpy(mt, hw.data(), sizeof(Meta));
  }
  // ncell == 0: a TableMasterDecoder blob (no cell-content decoder: pt_tsr_mtl_structure reports zero cells)
  PT_REQUIRE(mt->ncls > 0 && mt->ncell >= 0 && mt->max_len > 0 && mt->max_len + 2 <= PE_ROWS &&
                 (mt->ncell == 0 || (mt->max_len_c > 0 && mt->max_len_c + 2 <= PE_ROWS)) && mt->ffp % 64 == 0, "MtlTabNet decoder blob: bad meta");
  return PT_OK;
}

}  // namespace

namespace api {      // an entry point of include/pdftable_hip.h (reached through api_dispatch.cpp)
void pt_tsr_mtl_resized_size(int crop_w, int crop_h, int size, int32_t* out_w, int32_t* out_h) {
  int w = 0, h = 0;
  mtl_resized(crop_w, crop_h, size, &w, &h);
  *out_w = w;
  *out_h = h;
}
}  // namespace api

int pt_mtl_preprocess(pt_engine* e, const uint8_t* pages, int ph, int pw, const pt_tsr_table* tabs, int n, int size, bf16_t* out, hipStream_t s) {
  const int x3 = pt_split(e) ? 1 : 0;
  PtProfScope ps(e, s, PT_PROF_OTHER, 0, "mtl preprocess");
  hipLaunchKernelGGL(mtl_preprocess_kernel, dim3((size * size + 255) / 256, n), dim3(256), 0, s, pages, ph, pw, tabs, size, out, x3);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

void pt_mtl_release(pt_engine* e) {
  if (!e || !e->mtl_state) return;
  MtlState* st = reinterpret_cast<MtlState*>(e->mtl_state);
  st->persist.release();
  st->work.release();
  st->cwork.release();
  if (st->h_poll) (void)hipHostFree(st->h_poll);
  delete st;
  e->mtl_state = nullptr;
}

int pt_mtl_decoder_config(pt_engine* e, int32_t* out13) {
  auto it = e->models.find(PT_MODEL_MTL_DECODER);
  if (it == e->models.end()) {
    pt_set_error("MtlTabNet decoder weights not loaded (pt_weights_load(PT_MODEL_MTL_DECODER))");
    return PT_ERR_STATE;
  }
  if (!pt_model_format_ok(it->second, "PT_MODEL_MTL_DECODER")) return PT_ERR_STATE;
  Ctx c{e, &it->second, nullptr, 0, 1, PT_OK};
  Meta mt;
  const int rc = read_meta(c, &mt);
  if (rc != PT_OK) return rc;
  memcpy(out13, &mt, sizeof(mt));
  return PT_OK;
}

// Structure + box decoders.  f3 fp32 [n, hw, 512] (the backbone's last map, NHWC).  Outputs (device, caller-owned):
// tag_logits [n, T, ncls], boxes [n, T, 4] with T = max_len + 1; host: lens[n] (positions written per table), cell_counts[n].
int pt_mtl_structure(pt_engine* e, const float* f3, int n, int hw, float* d_tag_logits, float* d_boxes, int32_t* h_lens, int32_t* h_cell_counts,
                     int force_redecode, hipStream_t s) {
  PT_REQUIRE(e && f3 && n > 0 && hw > 0 && hw <= PE_ROWS && d_tag_logits && d_boxes && h_lens && h_cell_counts, "pt_tsr_mtl_structure: bad arguments");
  auto it = e->models.find(PT_MODEL_MTL_DECODER);
  if (it == e->models.end()) {
    pt_set_error("MtlTabNet decoder weights not loaded (pt_weights_load(PT_MODEL_MTL_DECODER))");
    return PT_ERR_STATE;
  }
  if (!pt_model_format_ok(it->second, "PT_MODEL_MTL_DECODER")) return PT_ERR_STATE;
  Ctx c{e, &it->second, s, pt_split(e) ? 1 : 0, pt_split(e) ? 2 : 1, PT_OK};
  Meta mt;
  int rc = read_meta(c, &mt);
  if (rc != PT_OK) return rc;
  PT_REQUIRE(mt.ncls <= 1024, "MtlTabNet decoder: %d structure classes", mt.ncls);
  MtlState* st = state_of(e);
  st->structure_done = false;
  const int mul = c.mul, T = mt.max_len + 1, Mp = (n + 127) / 128 * 128, ncls_p = (mt.ncls + 63) / 64 * 64;
  const long long Fr = ((long long)n * hw + 127) / 128 * 128;
  st->n = n; st->hw = hw; st->Mp = Mp; st->T = T; st->x3 = c.x3;
  if (!st->h_poll) PT_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&st->h_poll), 4096 * sizeof(int)));
  PT_REQUIRE(n + 1 <= 4096, "pt_tsr_mtl_structure: at most 4095 tables per call");

  const bool kv_fp8 = e->mtl_kv_fp8 && !c.x3;
  // ---- persistent buffers
  {
    Carver cv;
    st->o_kv = cv.take((size_t)Fr * KVC * mul * sizeof(bf16_t));
    st->o_kv8 = cv.take(kv_fp8 ? (size_t)Fr * KVC : 0);
    st->o_keep = cv.take((size_t)T * Mp * D * sizeof(float));
    st->o_tok = cv.take((size_t)(T + 1) * Mp * sizeof(int));
    st->o_ids = cv.take((size_t)T * Mp * sizeof(int));
    st->o_fin = cv.take((size_t)(Mp + 64) * sizeof(int));
    if ((rc = st->persist.ensure(cv.off)) != PT_OK) return rc;
  }
  char* pb = st->persist.base;
  bf16_t* kv = reinterpret_cast<bf16_t*>(pb + st->o_kv);
  float* keep = reinterpret_cast<float*>(pb + st->o_keep);
  int* tok = reinterpret_cast<int*>(pb + st->o_tok);
  int* ids = reinterpret_cast<int*>(pb + st->o_ids);
  int* fin = reinterpret_cast<int*>(pb + st->o_fin);
  int* first_pad = fin + Mp;

  // ---- work buffers: cached mode needs one position per buffer, re-decode mode all T
  struct Lay { size_t x[3], cache, xb, qc, att, hb, lg, bx, opart, mlpart, tiles, featb, total; long long R, prow; int nsplit_cap; bool pair; };
  auto plan = [&](bool all_positions) {
    Lay L;
    Carver cv;
    const long long R = all_positions ? (long long)T * Mp : Mp;
    L.R = R;
    L.pair = !all_positions;
    for (int i = 0; i < 3; ++i) L.x[i] = cv.take((size_t)R * D * sizeof(float));
    L.cache = cv.take((size_t)4 * T * Mp * 3 * D * mul * sizeof(bf16_t));
    const int nbuf = all_positions ? 1 : 2;      // KV-cached mode: the structure-token and the box layer of a step share their launches (run_layer, nb = 2)
    L.xb = cv.take((size_t)R * D * mul * sizeof(bf16_t));
    L.qc = cv.take((size_t)nbuf * R * D * mul * sizeof(bf16_t));
    L.att = cv.take((size_t)nbuf * R * D * mul * sizeof(bf16_t));
    L.hb = cv.take((size_t)nbuf * R * mt.ffp * mul * sizeof(bf16_t));
    L.lg = cv.take((size_t)R * ncls_p * sizeof(float));
    L.bx = cv.take((size_t)R * 8 * sizeof(float));
    // split-key partials: a step with `npos` positions in flight uses nsplit(npos) slices of npos * Mp rows each (stride = the rows in use,
    // Work::Rs).  pick_split() shrinks as the tile count grows, so the product peaks around 32 positions -- sized for that peak instead of
    // 16 x T x Mp rows (2.1 GB of fp32 for one 128-table micro-batch in re-decode mode)
    long long prow = 0;
    L.nsplit_cap = 1;
    for (int np_ = 1; np_ <= (all_positions ? T : 1); ++np_) {
      int kps_ = 0;
      const int ns_ = pick_split(all_positions ? n * ((np_ + 31) / 32) : n, hw, &kps_);
      if (ns_ > L.nsplit_cap) L.nsplit_cap = ns_;
      if ((long long)ns_ * np_ * Mp > prow) prow = (long long)ns_ * np_ * Mp;
    }
    L.prow = prow;
    L.opart = cv.take((size_t)nbuf * prow * D * sizeof(float));
    L.mlpart = cv.take((size_t)nbuf * prow * HEADS * 2 * sizeof(float));
    L.tiles = cv.take(((size_t)n * ((T + 31) / 32) + 16) * sizeof(int4));
    L.featb = all_positions ? 0 : cv.take((size_t)Fr * D * mul * sizeof(bf16_t));
    L.total = cv.off;
    return L;
  };
  Lay L = plan(false);
  if ((rc = st->work.ensure(L.total)) != PT_OK) return rc;

  // ---- out_enc, then keys / values of the five source attentions in one GEMM
  {
    const float* pe = c.F("pe");
    if (c.rc != PT_OK) return c.rc;
    bf16_t* featb = reinterpret_cast<bf16_t*>(st->work.base + L.featb);
    {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "mtl feature + pe");
      hipLaunchKernelGGL(mtl_feature_kernel, dim3((unsigned)((Fr + 3) / 4)), dim3(256), 0, s, f3, pe, (long long)n * hw, Fr, hw, featb, c.x3);
    }
    c.gemm(featb, Fr, D, "kv", KVC, 0, kv, KVC);
    if (c.rc != PT_OK) return c.rc;
    if (kv_fp8) {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "mtl keys / values -> fp8");
      const long long n8 = Fr * KVC / 8;
      hipLaunchKernelGGL(mtl_kv_fp8_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, kv, n8, reinterpret_cast<unsigned char*>(pb + st->o_kv8));
    }
  }
  hipLaunchKernelGGL(mtl_fill_kernel, dim3((Mp + 255) / 256), dim3(256), 0, s, tok, Mp, mt.sos);
  hipLaunchKernelGGL(mtl_fill_kernel, dim3((Mp + 64 + 255) / 256), dim3(256), 0, s, fin, Mp + 64, INT_MAX);

  Seqs S;
  S.M = n; S.Mp = Mp; S.tok = tok; S.pad = mt.pad; S.ffp = mt.ffp; S.hw = hw;
  S.kv8 = kv_fp8 ? reinterpret_cast<const unsigned char*>(pb + st->o_kv8) : nullptr;
  const float *emb = c.F("emb"), *pe = c.F("pe");
  if (c.rc != PT_OK) return c.rc;
  static const char* LN[4] = {"l0", "l1", "cls", "bbox"};
  bool redecode = force_redecode != 0;
  Work W;
  auto bind = [&](const Lay& l) {
    char* wb = st->work.base;
    W.R = l.R;
    W.xb = reinterpret_cast<bf16_t*>(wb + l.xb); W.qc = reinterpret_cast<bf16_t*>(wb + l.qc); W.att = reinterpret_cast<bf16_t*>(wb + l.att);
    W.hb = reinterpret_cast<bf16_t*>(wb + l.hb); W.opart = reinterpret_cast<float*>(wb + l.opart); W.mlpart = reinterpret_cast<float*>(wb + l.mlpart);
    W.tiles = reinterpret_cast<int4*>(wb + l.tiles);
    W.bs_row = l.pair ? l.R * D * mul : 0; W.bs_hb = l.pair ? l.R * mt.ffp * mul : 0;
    W.bs_opart = l.pair ? l.prow * D : 0; W.bs_ml = l.pair ? l.prow * HEADS * 2 : 0;
  };
  auto enter_redecode = [&]() -> int {
    PT_HIP_CHECK(hipStreamSynchronize(s));
    L = plan(true);
    int r = st->work.ensure(L.total);
    if (r != PT_OK) return r;
    bind(L);
    PT_HIP_CHECK(hipMemsetAsync(W.att, 0, (size_t)L.R * D * mul * sizeof(bf16_t), s));
    return PT_OK;
  };
  bind(L);
  PT_HIP_CHECK(hipMemsetAsync(W.att, 0, (size_t)2 * L.R * D * mul * sizeof(bf16_t), s));
  if (redecode && (rc = enter_redecode()) != PT_OK) return rc;
  if (!redecode) {           // one query per table and step
    std::vector<int4> tl(n);
    for (int b = 0; b < n; ++b) tl[b] = make_int4(b, b, 1, Mp);
    W.ntiles = n;
    W.nsplit = pick_split(n, hw, &W.kps);
    W.single = true;
    if ((rc = upload_tiles(tl, W.tiles, s)) != PT_OK) return rc;
  }
  // finish / <PAD> poll every POLL steps (4, 8, 16 and a synchronisation-free ring of device-written host slots all measured within noise of each other)
  static const int POLL = getenv("PT_MTL_POLL") && atoi(getenv("PT_MTL_POLL")) > 0 ? atoi(getenv("PT_MTL_POLL")) : 16;
  int t = 0;
  while (t <= mt.max_len) {
    const int p0 = redecode ? 0 : t, npos = t - p0 + 1;
    const long long rows = (long long)npos * Mp;
    W.Rs = rows;
    char* wb = st->work.base;
    float* xs = reinterpret_cast<float*>(wb + L.x[0]);
    float* xc = reinterpret_cast<float*>(wb + L.x[1]);
    float* xx = reinterpret_cast<float*>(wb + L.x[2]);
    bf16_t* cache = reinterpret_cast<bf16_t*>(wb + L.cache);
    const size_t cache_layer = (size_t)T * Mp * 3 * D * mul;
    if (redecode) {           // tiles of up to 32 positions of one table
      std::vector<int4> tl;
      for (int b = 0; b < n; ++b)
        for (int q0 = 0; q0 < npos; q0 += 32) tl.push_back(make_int4(b, q0 * Mp + b, npos - q0 < 32 ? npos - q0 : 32, Mp));
      W.ntiles = (int)tl.size();
      W.nsplit = pick_split(W.ntiles, hw, &W.kps);
      W.single = false;
      if ((rc = upload_tiles(tl, W.tiles, s)) != PT_OK) return rc;
    }
    // KV-cached step: the shared layers work in place on keep[t] (x2 of every position, what the cell-content decoder reads: no copy; rows of finished
    // sequences are past their length and never read), then the structure-token and the box layer run as ONE chain of launches (nb = 2)
    static const bool pair_on = !(getenv("PT_MTL_PAIR") && atoi(getenv("PT_MTL_PAIR")) == 0);
    const bool pair = !redecode && pair_on && L.pair && W.single && c.fusable(rows, D, D, xc, D, 0, 0);
    if (pair) xs = keep + (size_t)t * Mp * D;
    {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "mtl embed");
      hipLaunchKernelGGL(mtl_embed_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, tok, emb, pe, p0, npos, Mp, n, xs, (bf16_t*)nullptr,
                         (const float*)nullptr, (const int*)nullptr, c.x3);
    }
    run_layer(c, LN[0], 0, xs, cache, kv, S, W, p0, t);
    run_layer(c, LN[1], 1, xs, cache + cache_layer, kv, S, W, p0, t);
    if (c.rc != PT_OK) return c.rc;
    float* lg = reinterpret_cast<float*>(wb + L.lg);
    float* bx = reinterpret_cast<float*>(wb + L.bx);
    if (pair) {
      const LayerB lb[2] = {{LN[2], xs, xc}, {LN[3], xs, xx}};
      run_layer(c, 2, lb, 2, cache + 2 * cache_layer, (long long)cache_layer, kv, S, W, p0, t);
      const Ctx::Lin heads[2] = {{nullptr, xc, "norm", "cls_fc", nullptr, lg, nullptr, ncls_p, 0}, {nullptr, xx, "norm", "bbox_fc", nullptr, bx, nullptr, 8, 8}};
      if (ncls_p == 64) c.lin(2, heads, rows, D, 64, 0, 0, W.xb);
      else { c.lin(1, &heads[0], rows, D, ncls_p, 0, 0, W.xb); c.lin(1, &heads[1], rows, D, 64, 0, 0, W.xb); }
    } else {
      {
        const long long tot = rows * (D / 4);
        hipLaunchKernelGGL(mtl_fork_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, xs, xc, xx, keep, fin, p0, t, Mp, n);
      }
      run_layer(c, LN[2], 2, xc, cache + 2 * cache_layer, kv, S, W, p0, t);
      run_layer(c, LN[3], 3, xx, cache + 3 * cache_layer, kv, S, W, p0, t);
      c.gemm_ln(xc, rows, "norm", W.xb, "cls_fc", ncls_p, 0, nullptr, 0, lg, ncls_p);
      c.gemm_ln(xx, rows, "norm", W.xb, "bbox_fc", 64, 0, nullptr, 0, bx, 8, 8);
    }
    if (c.rc != PT_OK) return c.rc;
    {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "mtl tag pick");
      // TableMasterDecoder.greedy_forward (:599-608) has no <EOS> stop: its output is the LAST of max_len + 1 passes over the whole prefix.  For the
      // causal rows that equals stopping at <EOS>; once a <PAD> was emitted (re-decode mode: its row attends to ALL positions of the prefix, later ones
      // included) it does not, so a table still running then goes on to the length limit like the reference's -- the convertor cuts at <EOS> either way
      const int eos_eff = (mt.ncell == 0 && redecode) ? -1 : mt.eos;
      hipLaunchKernelGGL(mtl_tag_pick_kernel, dim3((npos * n + 255) / 256), dim3(256), 0, s, lg, bx, p0, t, Mp, n, mt.ncls, ncls_p, eos_eff, mt.pad, mt.max_len, T,
                         tok, ids, fin, first_pad, d_tag_logits, d_boxes);
    }
    ++t;
    if (t % POLL == 0 || t > mt.max_len) {
      PT_HIP_CHECK(hipMemcpyAsync(st->h_poll, fin, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
      PT_HIP_CHECK(hipMemcpyAsync(st->h_poll + n, first_pad, sizeof(int), hipMemcpyDeviceToHost, s));
      PT_HIP_CHECK(hipStreamSynchronize(s));
      const int fp = st->h_poll[n];
      if (!redecode && fp != INT_MAX) {
        // a <PAD> was appended at step fp: everything after it was computed with a causal cache the reference does not have
        redecode = true;
        hipLaunchKernelGGL(mtl_rollback_kernel, dim3((n + 255) / 256), dim3(256), 0, s, fin, n, fp);
        if ((rc = enter_redecode()) != PT_OK) return rc;
        t = fp + 1;
        continue;
      }
      bool all = true;
      for (int b = 0; b < n; ++b) all = all && st->h_poll[b] != INT_MAX;
      if (all) break;
    }
  }
  PT_HIP_CHECK(hipGetLastError());
  // ---- lengths, cell positions (bbox_masks of decode_test :389-392 from the arg-max of the final logits)
  std::vector<int> h_ids((size_t)T * Mp), h_fin(n);
  PT_HIP_CHECK(hipMemcpyAsync(h_ids.data(), ids, h_ids.size() * sizeof(int), hipMemcpyDeviceToHost, s));
  PT_HIP_CHECK(hipMemcpyAsync(h_fin.data(), fin, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
  PT_HIP_CHECK(hipStreamSynchronize(s));
  st->lens.assign(n, 0);
  st->cell_tab.clear(); st->cell_src.clear(); st->tab_first.assign(n + 1, 0);
  for (int b = 0; b < n; ++b) {
    PT_REQUIRE(h_fin[b] != INT_MAX, "pt_tsr_mtl_structure: table %d did not finish (internal error)", b);
    const int len = h_fin[b] + 1;
    st->lens[b] = len;
    h_lens[b] = len;
    st->tab_first[b] = (int)st->cell_tab.size();
    for (int p = 0; p < len; ++p) {
      const int id = h_ids[(size_t)p * Mp + b];
      if (mt.ncell > 0 && (id == mt.tag0 || id == mt.tag1)) {
        st->cell_tab.push_back(b);
        st->cell_src.push_back(p * Mp + b);
      }
    }
    h_cell_counts[b] = (int)st->cell_tab.size() - st->tab_first[b];
  }
  st->tab_first[n] = (int)st->cell_tab.size();
  st->structure_done = true;
  return PT_OK;
}

// Cell-content decoder for the cells pt_mtl_structure found (total = sum of its cell_counts, cells ordered by (table, position)).
// Device outputs: cell_ids int32 / cell_prob fp32 [total, Tc] (Tc = max_len_cell + 1), cell_logits fp32 [total, Tc, ncell] or null;
// host: steps[n] = positions decoded per table (0 for a table without cells: the reference returns torch.zeros(1) there, :395-398).
int pt_mtl_cells(pt_engine* e, int total, int32_t* d_cell_ids, float* d_cell_prob, float* d_cell_logits, int32_t* h_steps, int force_redecode,
                 hipStream_t s) {
  PT_REQUIRE(e && h_steps, "pt_tsr_mtl_cells: bad arguments");
  MtlState* st = e->mtl_state ? reinterpret_cast<MtlState*>(e->mtl_state) : nullptr;
  if (!st || !st->structure_done) {
    pt_set_error("pt_tsr_mtl_cells: call pt_tsr_mtl_structure first");
    return PT_ERR_STATE;
  }
  auto it = e->models.find(PT_MODEL_MTL_DECODER);
  if (it == e->models.end()) {
    pt_set_error("MtlTabNet decoder weights not loaded (pt_weights_load(PT_MODEL_MTL_DECODER))");
    return PT_ERR_STATE;
  }
  if (!pt_model_format_ok(it->second, "PT_MODEL_MTL_DECODER")) return PT_ERR_STATE;
  Ctx c{e, &it->second, s, st->x3, st->x3 ? 2 : 1, PT_OK};
  PT_REQUIRE((pt_split(e) ? 1 : 0) == st->x3, "pt_tsr_mtl_cells: the precision changed since pt_tsr_mtl_structure");
  Meta mt;
  int rc = read_meta(c, &mt);
  if (rc != PT_OK) return rc;
  const int n = st->n, Mc = (int)st->cell_tab.size();
  PT_REQUIRE(total == Mc, "pt_tsr_mtl_cells: %d cells announced, the structure pass found %d", total, Mc);
  for (int b = 0; b < n; ++b) h_steps[b] = 0;
  if (Mc == 0) return PT_OK;
  PT_REQUIRE(d_cell_ids && d_cell_prob, "pt_tsr_mtl_cells: null output");
  const int mul = c.mul, Tc = mt.max_len_c + 1, Mp = (Mc + 127) / 128 * 128, ncell_p = (mt.ncell + 63) / 64 * 64, hw = st->hw;
  char* pb = st->persist.base;
  const bf16_t* kv = reinterpret_cast<const bf16_t*>(pb + st->o_kv);
  const float* keep = reinterpret_cast<const float*>(pb + st->o_keep);

  struct Lay { size_t x, cache, cin, xb, qc, att, hb, lg, tiles, total; long long R; };
  const int max_tiles_pos = Mc / 32 + n + 1;
  auto plan = [&](bool all_positions) {
    Lay L;
    Carver cv;
    const long long R = all_positions ? (long long)Tc * Mp : Mp;
    L.R = R;
    L.x = cv.take((size_t)R * D * sizeof(float));
    L.cache = cv.take((size_t)Tc * Mp * 3 * D * mul * sizeof(bf16_t));
    L.cin = cv.take((size_t)R * 2 * D * mul * sizeof(bf16_t));
    L.xb = cv.take((size_t)R * D * mul * sizeof(bf16_t));
    L.qc = cv.take((size_t)R * D * mul * sizeof(bf16_t));
    L.att = cv.take((size_t)R * D * mul * sizeof(bf16_t));
    L.hb = cv.take((size_t)R * mt.ffp * mul * sizeof(bf16_t));
    L.lg = cv.take((size_t)R * ncell_p * sizeof(float));
    L.tiles = cv.take((size_t)max_tiles_pos * (all_positions ? Tc : 1) * sizeof(int4));
    L.total = cv.off;
    return L;
  };
  Lay L = plan(false);
  if ((rc = st->cwork.ensure(L.total)) != PT_OK) return rc;
  // bookkeeping lives in `work` (the structure loop is over, its buffers are free): growing `cwork` for the re-decode mode must not lose it
  struct { size_t tok, nxt, tab, src, first, fin; } B;
  {
    Carver cv;
    B.tok = cv.take((size_t)(Tc + 1) * Mp * sizeof(int));
    B.nxt = cv.take((size_t)Mp * sizeof(int));
    B.tab = cv.take((size_t)Mp * sizeof(int));
    B.src = cv.take((size_t)Mp * sizeof(int));
    B.first = cv.take((size_t)(n + 1) * sizeof(int));
    B.fin = cv.take((size_t)(n + 64) * sizeof(int));
    if ((rc = st->work.ensure(cv.off)) != PT_OK) return rc;
  }
  char* bk = st->work.base;
  int* tok = reinterpret_cast<int*>(bk + B.tok);
  int* nxt = reinterpret_cast<int*>(bk + B.nxt);
  int* d_tab = reinterpret_cast<int*>(bk + B.tab);
  int* d_src = reinterpret_cast<int*>(bk + B.src);
  int* d_first = reinterpret_cast<int*>(bk + B.first);
  int* finc = reinterpret_cast<int*>(bk + B.fin);
  int* first_pad = finc + n;
  PT_HIP_CHECK(hipMemcpyAsync(d_tab, st->cell_tab.data(), (size_t)Mc * sizeof(int), hipMemcpyHostToDevice, s));
  PT_HIP_CHECK(hipMemcpyAsync(d_src, st->cell_src.data(), (size_t)Mc * sizeof(int), hipMemcpyHostToDevice, s));
  PT_HIP_CHECK(hipMemcpyAsync(d_first, st->tab_first.data(), (size_t)(n + 1) * sizeof(int), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(mtl_fill_kernel, dim3((Mp + 255) / 256), dim3(256), 0, s, tok, Mp, mt.sos_c);
  hipLaunchKernelGGL(mtl_fill_kernel, dim3((n + 64 + 255) / 256), dim3(256), 0, s, finc, n + 64, INT_MAX);
  PT_HIP_CHECK(hipStreamSynchronize(s));       // the three vectors above belong to the state, but keep the copies simple

  Seqs S;
  S.M = Mc; S.Mp = Mp; S.tok = tok; S.pad = mt.pad_c; S.ffp = mt.ffp; S.hw = hw;
  const float *emb = c.F("emb_cell"), *pe = c.F("pe");
  if (c.rc != PT_OK) return c.rc;
  Work W;
  auto bind = [&](const Lay& l) {
    char* wb = st->cwork.base;
    W.R = l.R;
    W.xb = reinterpret_cast<bf16_t*>(wb + l.xb); W.qc = reinterpret_cast<bf16_t*>(wb + l.qc); W.att = reinterpret_cast<bf16_t*>(wb + l.att);
    W.hb = reinterpret_cast<bf16_t*>(wb + l.hb); W.tiles = reinterpret_cast<int4*>(wb + l.tiles);
    W.opart = nullptr; W.mlpart = nullptr; W.nsplit = 1; W.kps = (hw + 31) / 32 * 32;
  };
  auto make_tiles = [&](int npos) -> int {
    std::vector<int4> tl;
    for (int pi = 0; pi < npos; ++pi)
      for (int b = 0; b < n; ++b)
        for (int c0 = st->tab_first[b]; c0 < st->tab_first[b + 1]; c0 += 32)
          tl.push_back(make_int4(b, pi * Mp + c0, st->tab_first[b + 1] - c0 < 32 ? st->tab_first[b + 1] - c0 : 32, 1));
    W.ntiles = (int)tl.size();
    return upload_tiles(tl, W.tiles, s);
  };
  bool redecode = force_redecode != 0;
  auto enter_redecode = [&]() -> int {
    PT_HIP_CHECK(hipStreamSynchronize(s));
    L = plan(true);
    int r = st->cwork.ensure(L.total);
    if (r != PT_OK) return r;
    bind(L);
    PT_HIP_CHECK(hipMemsetAsync(W.att, 0, (size_t)L.R * D * mul * sizeof(bf16_t), s));
    return PT_OK;
  };
  bind(L);
  PT_HIP_CHECK(hipMemsetAsync(W.att, 0, (size_t)L.R * D * mul * sizeof(bf16_t), s));
  if (redecode && (rc = enter_redecode()) != PT_OK) return rc;
  if (!redecode && (rc = make_tiles(1)) != PT_OK) return rc;
  const int POLL = 2;      // a cell step is ~1 ms of device work (every cell of every table): the poll's bubble is nothing, a step after the last <EOS> is
  int t = 0;
  while (t <= mt.max_len_c) {
    const int p0 = redecode ? 0 : t, npos = t - p0 + 1;
    const long long rows = (long long)npos * Mp;
    char* wb = st->cwork.base;
    float* x = reinterpret_cast<float*>(wb + L.x);
    bf16_t* cache = reinterpret_cast<bf16_t*>(wb + L.cache);
    bf16_t* cin = reinterpret_cast<bf16_t*>(wb + L.cin);
    float* lg = reinterpret_cast<float*>(wb + L.lg);
    if (redecode && (rc = make_tiles(npos)) != PT_OK) return rc;
    {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "mtl cell embed");
      hipLaunchKernelGGL(mtl_embed_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, tok, emb, pe, p0, npos, Mp, Mc, (float*)nullptr, cin, keep, d_src, c.x3);
    }
    c.gemm(cin, rows, 2 * D, "cell_in", D, 0, nullptr, 0, x, D);
    run_layer(c, "cell", 4, x, cache, kv, S, W, p0, t);
    c.gemm_ln(x, rows, "norm", W.xb, "cell_fc", ncell_p, 0, nullptr, 0, lg, ncell_p);
    if (c.rc != PT_OK) return c.rc;
    {
      PtProfScope ps(e, s, PT_PROF_OTHER, 0, "mtl cell pick");
      const long long waves = (long long)npos * Mc;
      hipLaunchKernelGGL(mtl_cell_pick_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, lg, p0, t, Mp, Mc, mt.ncell, ncell_p, Tc, d_tab, finc, nxt,
                         d_cell_ids, d_cell_prob, d_cell_logits);
      hipLaunchKernelGGL(mtl_cell_next_kernel, dim3(n), dim3(256), 0, s, d_first, nxt, t, Mp, mt.eos_c, mt.pad_c, mt.max_len_c, tok, finc, first_pad);
    }
    ++t;
    if (t % POLL == 0 || t > mt.max_len_c) {
      PT_HIP_CHECK(hipMemcpyAsync(st->h_poll, finc, (size_t)(n + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
      PT_HIP_CHECK(hipStreamSynchronize(s));
      const int fp = st->h_poll[n];
      if (!redecode && fp != INT_MAX) {
        redecode = true;
        hipLaunchKernelGGL(mtl_rollback_kernel, dim3((n + 255) / 256), dim3(256), 0, s, finc, n, fp);
        if ((rc = enter_redecode()) != PT_OK) return rc;
        t = fp + 1;
        continue;
      }
      bool all = true;
      for (int b = 0; b < n; ++b) all = all && (st->h_poll[b] != INT_MAX || st->tab_first[b + 1] == st->tab_first[b]);
      if (all) break;
    }
  }
  PT_HIP_CHECK(hipGetLastError());
  PT_HIP_CHECK(hipMemcpyAsync(st->h_poll, finc, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
  PT_HIP_CHECK(hipStreamSynchronize(s));
  for (int b = 0; b < n; ++b) {
    const bool has = st->tab_first[b + 1] > st->tab_first[b];
    PT_REQUIRE(!has || st->h_poll[b] != INT_MAX, "pt_tsr_mtl_cells: table %d did not finish (internal error)", b);
    h_steps[b] = has ? st->h_poll[b] + 1 : 0;
  }
  return PT_OK;
}

}  // namespace PT_FMT_NS
