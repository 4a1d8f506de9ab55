// lore_decode.hip -- Lore's heat-map + corner-point decode on the device.
//
// Reference: process_detect_output, lore/lineless_table_process.py:592-655
//   corner_decode :97-124, ctdet_4ps_decode :127-267 (incl. the wiz_rev vertex-snapping double loop :188-236 that the
//   reference runs in Python with per-element tensors), _nms :66-73, _topk :76-94, _get_4ps_feat :39-63.
// What the reference materialises as 3000 / 5000 top-K rows (mostly zero-score non-peaks) is kept here as the list
// of real peaks whose score passes the thresholds that can matter downstream: cells >= vis_thresh, corners >= 0.3
// (:193); both sorted by (score desc, pixel index asc) and capped at K = 3000 / MK = 5000 like torch.topk.
//
//   lore_sigmoid_kernel   hm logits -> scores (2 classes)
//   lore_peaks_kernel     3x3 max-pool equality (-inf padding) + threshold -> (score, index) keys, atomically appended
//   lore_sort_kernel      one workgroup per list: bitonic sort of <= 16384 64-bit keys in LDS
//   lore_boxes_kernel     centre + reg, 4 corner points = centre - wh / st  (fp32, same operation order)
//   lore_snap_kernel      one wave per cell: bbox overlap + strict point-in-quad (fp64) against all corners, 64 at a
//                         time, then the order-dependent "snap the nearest vertex" update replayed in corner order
//   lore_gather_kernel    final order (re-sorted by the x0.4-demoted scores), logic features ax[centre] + sum of cr at
//                         the 4 corner pixels -- with the reference's quirk that cr_feat is NOT re-sorted (:254-263)
#include "common.h"

namespace PT_FMT_NS {

namespace {

constexpr int CAP = 16384;         // candidate capacity per (table, class)
constexpr int K_CELLS = 3000, K_CORNERS = 5000;

__global__ __launch_bounds__(256) void lore_sigmoid_kernel(const float* __restrict__ hm, float* __restrict__ sig,
                                                            long long npix) {
  a16_kernel_enter();
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  sig[2 * i] = 1.f / (1.f + expf(-hm[8 * i]));
  sig[2 * i + 1] = 1.f / (1.f + expf(-hm[8 * i + 1]));
}

// keys[(b * 2 + cls) * CAP + slot] = score_bits << 32 | (0xFFFFFFFF - pixel index): descending key order is
// (score desc, index asc)
template <int NTHR>
__global__ __launch_bounds__(NTHR) void lore_peaks_kernel(const float* __restrict__ sig, int B, int H, int W, float thr_cell,
                                                          float thr_corner, unsigned long long* __restrict__ keys,
                                                          int* __restrict__ counts) {
  a16_kernel_enter();
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < (long long)B * H * W;
  const long long ic = live ? i : 0;
  const int x = (int)(ic % W), y = (int)((ic / W) % H), b = (int)(ic / ((long long)W * H));
  const float* sb = sig + (size_t)b * H * W * 2;
  // a workgroup never straddles two tables: one counter and ONE atomic per workgroup and class (the waves' counts meet in LDS).  With one
  // atomic per wave a 256 x 256 map sent up to 1024 of them to the same address -- serialised in L2, 0.5 ms per 87 tables
  const bool uniform = ((long long)H * W) % NTHR == 0;
  __shared__ int s_cnt[2][NTHR / 64];
  __shared__ int s_base[2];
  const int wave = threadIdx.x >> 6;
  for (int cls = 0; cls < 2; ++cls) {
    const float s = sb[((size_t)y * W + x) * 2 + cls];
    bool peak = live && (s >= (cls ? thr_corner : thr_cell));
    if (peak) {
      for (int dy = -1; dy <= 1 && peak; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = y + dy, xx = x + dx;
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
          if (sb[((size_t)yy * W + xx) * 2 + cls] > s) { peak = false; break; }
        }
    }
    const int list = b * 2 + cls;
    int slot = -1;
    if (uniform) {
      // plateaus of equal scores make every pixel of them a peak: thousands of appends per list (the order inside a list is
      // irrelevant, it is sorted next)
      const unsigned long long m = __ballot(peak);
      if ((threadIdx.x & 63) == 0) s_cnt[cls][wave] = __popcll(m);
      __syncthreads();
      if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < NTHR / 64; ++k) tot += s_cnt[cls][k];
        s_base[cls] = tot ? atomicAdd(&counts[list], tot) : 0;
      }
      __syncthreads();
      if (peak) {
        int base = s_base[cls];
        for (int k = 0; k < wave; ++k) base += s_cnt[cls][k];
        slot = base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
      }
    } else if (peak) {
      slot = atomicAdd(&counts[list], 1);
    }
    if (peak && slot < CAP)
      keys[(size_t)list * CAP + slot] = ((unsigned long long)__float_as_uint(s) << 32) |
                                        (unsigned long long)(0xFFFFFFFFu - (unsigned)(y * W + x));
  }
}

// Sorts list `blockIdx.x` (n = min(counts, CAP) keys) descending; writes the first min(n, kmax) to out and that count.
__global__ __launch_bounds__(1024) void lore_sort_kernel(const unsigned long long* __restrict__ keys,
                                                          const int* __restrict__ counts, int count_stride,
                                                          int stride_in, int kmax_even, int kmax_odd,
                                                          unsigned long long* __restrict__ out, int stride_out,
                                                          int* __restrict__ out_counts) {
  a16_kernel_enter();
  extern __shared__ unsigned long long sk[];
  const int list = blockIdx.x;
  int n = counts[list * count_stride];
  if (n > stride_in) n = stride_in;
  if (n > CAP) n = CAP;
  int P = 2;
  while (P < n) P <<= 1;
  for (int i = threadIdx.x; i < P; i += blockDim.x) sk[i] = i < n ? keys[(size_t)list * stride_in + i] : 0ull;
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = sk[i], c = sk[l];
          const bool desc = (i & k) == 0;
          if (desc ? a < c : a > c) { sk[i] = c; sk[l] = a; }
        }
      }
      __syncthreads();
    }
  const int kmax = (list & 1) ? kmax_odd : kmax_even;
  const int m = n < kmax ? n : kmax;
  for (int i = threadIdx.x; i < m; i += blockDim.x) out[(size_t)list * stride_out + i] = sk[i];
  if (threadIdx.x == 0) out_counts[list] = m;
}

// boxes[(list, k)][0..7] = centre - offsets, [8] = score, [9] = centre x, [10] = centre y, [11] = pixel index
__device__ __forceinline__ size_t mosaic_centre(long long j);

// reg0 / reg1: the `reg` head for the cell / corner lists, wh, st: dense maps [B,H,W,8] -- or, when pk_base is given, the
// head outputs on the peak-patch mosaics (patch pk_base[list] + k, value at its centre pixel)
__global__ __launch_bounds__(256) void lore_boxes_kernel(const unsigned long long* __restrict__ sorted,
                                                          const int* __restrict__ counts, int stride, int B, int H, int W,
                                                          const float* __restrict__ reg0, const float* __restrict__ reg1,
                                                          const float* __restrict__ wh, const float* __restrict__ st,
                                                          const int* __restrict__ pk_base, float* __restrict__ boxes) {
  a16_kernel_enter();
  const int list = blockIdx.y, b = list >> 1, cls = list & 1;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= counts[list]) return;
  const unsigned long long key = sorted[(size_t)list * stride + k];
  const int idx = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
  const float score = __uint_as_float((unsigned)(key >> 32));
  const size_t pix = pk_base ? mosaic_centre((long long)pk_base[list] + k) : (size_t)b * H * W + idx;
  const float* rg = (cls ? reg1 : reg0) + pix * 8;
  const float xs = (float)(idx % W) + rg[0];
  const float ys = (float)(idx / W) + rg[1];
  const float* off = (cls ? st : wh) + pix * 8;
  float* o = boxes + ((size_t)list * stride + k) * 12;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    o[2 * m] = xs - off[2 * m];
    o[2 * m + 1] = ys - off[2 * m + 1];
  }
  o[8] = score; o[9] = xs; o[10] = ys; o[11] = __int_as_float(idx);
}

// shapely Point.within(Polygon): strictly interior (even-odd rule, boundary excluded), in fp64 on fp32 coordinates
__device__ bool point_in_quad(double px, double py, const float* q) {
  bool inside = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = (i + 1) & 3;
    const double x1 = q[2 * i], y1 = q[2 * i + 1], x2 = q[2 * j], y2 = q[2 * j + 1];
    const double cross = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1);
    if (cross == 0.0 && fmin(x1, x2) <= px && px <= fmax(x1, x2) && fmin(y1, y2) <= py && py <= fmax(y1, y2))
      return false;
  }
  int jj = 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double xi = q[2 * i], yi = q[2 * i + 1], xj = q[2 * jj], yj = q[2 * jj + 1];
    if ((yi > py) != (yj > py)) {
      const double xint = (xj - xi) * (py - yi) / (yj - yi) + xi;
      if (px < xint) inside = !inside;
    }
    jj = i;
  }
  return inside;
}

// one wave per cell (cells with score >= 0.2 only, lineless_table_process.py:190); rev[(b, k)][0..7], new score [8]
__global__ __launch_bounds__(64) void lore_snap_kernel(const float* __restrict__ boxes, const int* __restrict__ counts,
                                                        int stride, float* __restrict__ rev) {
  a16_kernel_enter();
  const int b = blockIdx.y, k = blockIdx.x, lane = threadIdx.x;
  const int ncell = counts[2 * b], ncorner = counts[2 * b + 1];
  if (k >= ncell) return;
  const float* bbp = boxes + ((size_t)(2 * b) * stride + k) * 12;
  float bb[8], rv[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) bb[m] = rv[m] = bbp[m];
  float score = bbp[8];
  if (score >= 0.2f) {
    const float bxmin = fminf(fminf(bb[0], bb[2]), fminf(bb[4], bb[6])), bxmax = fmaxf(fmaxf(bb[0], bb[2]), fmaxf(bb[4], bb[6]));
    const float bymin = fminf(fminf(bb[1], bb[3]), fminf(bb[5], bb[7])), bymax = fmaxf(fmaxf(bb[1], bb[3]), fmaxf(bb[5], bb[7]));
    int count = 0;
    const float* cbase = boxes + (size_t)(2 * b + 1) * stride * 12;
    for (int base = 0; base < ncorner; base += 64) {
      const int j = base + lane;
      bool hit = false;
      float cx = 0.f, cy = 0.f;
      if (j < ncorner) {
        const float* g = cbase + (size_t)j * 12;
        float gq[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) gq[m] = g[m];
        cx = g[9]; cy = g[10];
        const float gxmin = fminf(fminf(gq[0], gq[2]), fminf(gq[4], gq[6])), gxmax = fmaxf(fmaxf(gq[0], gq[2]), fmaxf(gq[4], gq[6]));
        const float gymin = fminf(fminf(gq[1], gq[3]), fminf(gq[5], gq[7])), gymax = fmaxf(fmaxf(gq[1], gq[3]), fmaxf(gq[5], gq[7]));
        if (!(bxmin > gxmax || gxmin > bxmax || bymin > gymax || gymin > bymax)) {
#pragma unroll
          for (int m = 0; m < 4 && !hit; ++m) hit = point_in_quad((double)gq[2 * m], (double)gq[2 * m + 1], bb);
        }
      }
      unsigned long long mask = __ballot(hit);
      while (mask) {
        const int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const float px = __shfl(cx, src), py = __shfl(cy, src);
        // find4ps (:329-337): nearest of the ORIGINAL four vertices, first minimum
        int kk = 0;
        float best = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float dx = bb[2 * m] - px, dy = bb[2 * m + 1] - py;
          const float d = dx * dx + dy * dy;
          if (m == 0 || d < best) { best = d; kk = m; }
        }
        float ox = 0.f, oy = 0.f, rx = 0.f, ry = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (m == kk) { ox = bb[2 * m]; oy = bb[2 * m + 1]; rx = rv[2 * m]; ry = rv[2 * m + 1]; }
        bool take;
        if (rx == ox && ry == oy) {
          take = true;
        } else {
          const float dox = ox - rx, doy = oy - ry, dnx = ox - px, dny = oy - py;
          take = (dox * dox + doy * doy) >= (dnx * dnx + dny * dny);
        }
        if (take) {
          ++count;
#pragma unroll
          for (int m = 0; m < 4; ++m)
            if (m == kk) { rv[2 * m] = px; rv[2 * m + 1] = py; }
        }
      }
    }
    if (count <= 2) score = score * 0.4f;
  }
  if (lane < 8) rev[((size_t)b * stride + k) * 12 + lane] = rv[lane];
  if (lane == 8) rev[((size_t)b * stride + k) * 12 + 8] = score;
}

// key of cell k after the snap: (new score desc, previous rank asc)
__global__ __launch_bounds__(256) void lore_rekey_kernel(const float* __restrict__ rev, const int* __restrict__ counts,
                                                          int stride, unsigned long long* __restrict__ keys) {
  a16_kernel_enter();
  const int b = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= counts[2 * b]) return;
  const float s = rev[((size_t)b * stride + k) * 12 + 8];
  keys[(size_t)b * stride + k] = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)k);
}

// Patch rows.  The sparse heads are 3x3 (64 -> 256) + ReLU + 1x1 convolutions evaluated at single pixels: the 3x3
// neighbourhood of such a pixel, laid out as ONE row of 9 * C values in the K order of the conv kernel -- [32-channel chunk]
// [tap][32 channels], the (hi | lo) halves one after the other in hi/lo mode -- turns the 3x3 conv into a plain GEMM over
// the patches whose K-chunks ARE the 3x3 layer's weight tiles in their packed order: same weights, same MFMA sequence, so
// the same bits as the dense conv at that pixel, for 1/9 of the FLOPs a 3x3-pixel mosaic patch needs.  The rows form an
// "image" of MOS_PW patches per row for the row-limited 1x1 launches.
constexpr int MOS_PW = 32;         // patches per row of the patch image
__device__ __forceinline__ size_t mosaic_centre(long long j) { return (size_t)j; }
// element offset of (tap, channel ch of cs) inside a patch row; C = real channels (cs = C or 2C)
__device__ __forceinline__ int patch_off(int tap, int ch, int C) {
  const int half = ch >= C ? 1 : 0, cl = ch - half * C;
  return half * 9 * C + ((cl >> 5) * 9 + tap) * 32 + (cl & 31);
}

// exclusive prefix of the kept cell counts over the tables + the two row limits of the mosaics (one thread: B <= 64)
// base[b] = sum_{b' < b} ncell[b'];  lim[0] / lim[1] = pixel rows of the ax / cr mosaic that hold patches
__global__ void lore_sparse_base_kernel(const int* __restrict__ counts, int B, int* __restrict__ base, int* __restrict__ lim) {
  a16_kernel_enter();
  if (threadIdx.x || blockIdx.x) return;
  long long t = 0;
  for (int b = 0; b < B; ++b) {
    base[b] = (int)t;
    t += counts[2 * b];
  }
  lim[0] = (int)((t + MOS_PW - 1) / MOS_PW);
  lim[1] = (int)((4 * t + MOS_PW - 1) / MOS_PW);
}

// the same for the kept peaks of the first sort: pk_base[2 b + cls] = number of kept peaks of class cls in the tables
// before b; lim[0] / lim[1] = pixel rows of the cell / corner mosaic that hold patches
__global__ void lore_peak_base_kernel(const int* __restrict__ kept, int B, int* __restrict__ pk_base, int* __restrict__ lim) {
  a16_kernel_enter();
  if (threadIdx.x || blockIdx.x) return;
  long long t0 = 0, t1 = 0;
  for (int b = 0; b < B; ++b) {
    pk_base[2 * b] = (int)t0;
    pk_base[2 * b + 1] = (int)t1;
    t0 += kept[2 * b];
    t1 += kept[2 * b + 1];
  }
  lim[0] = (int)((t0 + MOS_PW - 1) / MOS_PW);
  lim[1] = (int)((t1 + MOS_PW - 1) / MOS_PW);
}

// 3x3 neighbourhoods of the kept peaks (cells and corners, in sorted order) -> the cell / corner patch mosaics
__global__ __launch_bounds__(64) void lore_peak_patch_kernel(const unsigned long long* __restrict__ sorted,
                                                             const int* __restrict__ kept, int stride, int H, int W,
                                                             const bf16_t* __restrict__ feat, int C, int split,
                                                             const int* __restrict__ pk_base, bf16_t* __restrict__ mos_cell,
                                                             bf16_t* __restrict__ mos_corner) {
  a16_kernel_enter();
  const int list = blockIdx.y, b = list >> 1, cls = list & 1, k = blockIdx.x;
  if (k >= kept[list]) return;
  const int idx = (int)(0xFFFFFFFFu - (unsigned)(sorted[(size_t)list * stride + k] & 0xFFFFFFFFull));
  const int cs = split ? 2 * C : C, ppx = cs / 8;
  const long long pj = (long long)pk_base[list] + k;
  bf16_t* mos = cls ? mos_corner : mos_cell;
  for (int i = threadIdx.x; i < 9 * ppx; i += 64) {
    const int piece = i % ppx, tap = i / ppx;
    const int y = idx / W + tap / 3 - 1, x = idx % W + tap % 3 - 1;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
      v = *reinterpret_cast<const uint4*>(feat + (((size_t)b * H + y) * W + x) * cs + piece * 8);
    *reinterpret_cast<uint4*>(mos + (size_t)pj * 9 * cs + patch_off(tap, piece * 8, C)) = v;
  }
}

// The 3x3 neighbourhoods (zero outside the map) of the five feature-map positions lore_gather_kernel reads for output
// row p of table b -- the cell's centre for `ax`, its four corner pixels for `cr` -- copied into the two patch mosaics.
// feat: bf16 [B,H,W,C] ([hi | lo] halves when split).  One workgroup per (p, b), 16-byte pieces.
__global__ __launch_bounds__(256) void lore_patch_gather_kernel(const float* __restrict__ rev, const float* __restrict__ boxes,
                                                                const unsigned long long* __restrict__ order,
                                                                const int* __restrict__ counts, int stride, int H, int W,
                                                                const bf16_t* __restrict__ feat, int C, int split,
                                                                const int* __restrict__ sp_base, bf16_t* __restrict__ mos_ax,
                                                                bf16_t* __restrict__ mos_cr) {
  a16_kernel_enter();
  const int b = blockIdx.y, p = blockIdx.x;
  const int ncell = counts[2 * b];
  if (p >= ncell) return;
  const int r = order ? (int)(0xFFFFFFFFu - (unsigned)(order[(size_t)b * stride + p] & 0xFFFFFFFFull)) : p;
  const float* cellbox = boxes + ((size_t)(2 * b) * stride) * 12;
  const float* here = rev ? rev + ((size_t)b * stride + p) * 12 : cellbox + (size_t)p * 12;
  __shared__ int pos[5];
  if (threadIdx.x < 5) {
    const int m = (int)threadIdx.x - 1;
    long long cc;
    if (m < 0) {
      cc = __float_as_int(cellbox[(size_t)r * 12 + 11]);
    } else {
      const float f = here[2 * m] + (float)W * rintf(here[2 * m + 1]);
      cc = (long long)rintf(f);
      if (!(cc < (long long)H * W)) cc = 0;
      if (cc < 0) cc = 0;
    }
    pos[threadIdx.x] = (int)cc;
  }
  __syncthreads();
  const int cs = split ? 2 * C : C, ppx = cs / 8;          // 16-byte pieces per pixel
  const long long j = sp_base[b] + p;
  for (int i = threadIdx.x; i < 5 * 9 * ppx; i += 256) {
    const int piece = i % ppx, tap = (i / ppx) % 9, k = i / (9 * ppx);
    const int y = pos[k] / W + tap / 3 - 1, x = pos[k] % W + tap % 3 - 1;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
      v = *reinterpret_cast<const uint4*>(feat + (((size_t)b * H + y) * W + x) * cs + piece * 8);
    const long long pj = k == 0 ? j : 4 * j + (k - 1);
    bf16_t* dst = (k == 0 ? mos_ax : mos_cr) + (size_t)pj * 9 * cs + patch_off(tap, piece * 8, C);
    *reinterpret_cast<uint4*>(dst) = v;
  }
}

// one workgroup per output row p of table b.  order[p] (or p itself when there was no snap pass) is the pre-sort rank
// whose box / score / centre go to row p; the corner features use the box that sat at rank p BEFORE the re-sort.
__global__ __launch_bounds__(256) void lore_gather_kernel(const float* __restrict__ rev, const float* __restrict__ boxes,
                                                           const unsigned long long* __restrict__ order,
                                                           const int* __restrict__ counts, int stride, int H, int W,
                                                           const float* __restrict__ ax, const float* __restrict__ cr,
                                                           float vis_thresh, float* __restrict__ dets,
                                                           float* __restrict__ logi, int* __restrict__ n_valid,
                                                           const int* __restrict__ sp_base) {
  a16_kernel_enter();
  const int b = blockIdx.y, p = blockIdx.x, c = threadIdx.x;
  const int ncell = counts[2 * b];
  if (p >= ncell) return;
  const int r = order ? (int)(0xFFFFFFFFu - (unsigned)(order[(size_t)b * stride + p] & 0xFFFFFFFFull)) : p;
  const float* cellbox = boxes + ((size_t)(2 * b) * stride) * 12;
  const float* src = rev ? rev + ((size_t)b * stride + r) * 12 : cellbox + (size_t)r * 12;
  const float* here = rev ? rev + ((size_t)b * stride + p) * 12 : cellbox + (size_t)p * 12;
  const float score = src[8];
  if (c < 8) dets[((size_t)b * K_CELLS + p) * 9 + c] = src[c];
  if (c == 8) {
    dets[((size_t)b * K_CELLS + p) * 9 + 8] = score;
    if (score >= vis_thresh) atomicAdd(&n_valid[b], 1);
  }
  if (sp_base) {
    // sparse heads: ax / cr hold the head outputs of 3x3-pixel patches in a mosaic, 256 patches per mosaic row; the value
    // of patch j is its centre pixel (mosaic_centre); patch (sp_base[b] + p) of ax, 4 (sp_base[b] + p) + m of cr
    const long long j = sp_base[b] + p;
    float v = ax[mosaic_centre(j) * 256 + c];
    float crs = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) crs += cr[mosaic_centre(4 * j + m) * 256 + c];
    logi[((size_t)b * K_CELLS + p) * 256 + c] = v + crs;
    return;
  }
  const int centre = __float_as_int(cellbox[(size_t)r * 12 + 11]);
  const size_t fb = (size_t)b * H * W;
  float v = ax[(fb + centre) * 256 + c];
  float crs = 0.f;
  const long long npix = (long long)H * W;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float f = here[2 * m] + (float)W * rintf(here[2 * m + 1]);
    long long cc = (long long)rintf(f);
    if (!(cc < npix)) cc = 0;        // _get_4ps_feat's clamp when called with the cr tensor (:53-59)
    if (cc < 0) cc = 0;
    crs += cr[(fb + cc) * 256 + c];
  }
  logi[((size_t)b * K_CELLS + p) * 256 + c] = v + crs;
}

}  // namespace

// Scratch layout (engine-owned): sig f32 [B*H*W*2] | counts int [5B] | keys u64 [2B*CAP] | sorted u64 [2B*CAP] |
// boxes f32 [2B*CAP*12] | rev f32 [B*CAP*12] | keys2/sorted2 u64 [B*CAP] each | sparse bases / limits int [3B + 4]
struct DecodeState {
  int* cnt;
  unsigned long long *sorted, *keys2, *sorted2;
  float *boxes, *rev;
  int *sp_base, *sp_lim;      // final-order cells: base per table, row limits of the ax / cr mosaics
  int *pk_base, *pk_lim;      // kept peaks: base per (table, class), row limits of the cell / corner mosaics
  int wiz_rev;
};

// sigmoid, peak test, first sort: the kept cells / corners of every table in score order
static int decode_peaks(pt_engine* e, const float* hm, int B, int H, int W, int wiz_rev, float vis_thresh, int* d_counts,
                        DecodeState* ds, hipStream_t s) {
  PT_REQUIRE((long long)H * W < (1ll << 31), "tsr decode: map too large");
  const size_t npix = (size_t)B * H * W;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~size_t(255); return o; };
  const size_t o_sig = carve(npix * 2 * 4), o_cnt = carve((size_t)B * 5 * 4), o_keys = carve((size_t)2 * B * CAP * 8),
               o_sorted = carve((size_t)2 * B * CAP * 8), o_boxes = carve((size_t)2 * B * CAP * 12 * 4),
               o_rev = carve((size_t)B * CAP * 12 * 4), o_keys2 = carve((size_t)B * CAP * 8),
               o_sorted2 = carve((size_t)B * CAP * 8), o_sp = carve((size_t)(3 * B + 4) * 4);
  if (off > e->tsr_scratch_cap) {
    PT_HIP_CHECK(hipDeviceSynchronize());
    if (e->tsr_scratch) PT_HIP_CHECK(hipFree(e->tsr_scratch));
    e->tsr_scratch = nullptr; e->tsr_scratch_cap = 0;
    PT_HIP_CHECK(hipMalloc(&e->tsr_scratch, off));
    e->tsr_scratch_cap = off;
  }
  char* base = reinterpret_cast<char*>(e->tsr_scratch);
  float* sig = reinterpret_cast<float*>(base + o_sig);
  int* cnt = reinterpret_cast<int*>(base + o_cnt);   // [0,2B): raw peak counts, [2B,4B): kept counts, [4B,5B): dump
  auto* keys = reinterpret_cast<unsigned long long*>(base + o_keys);
  ds->cnt = cnt;
  ds->sorted = reinterpret_cast<unsigned long long*>(base + o_sorted);
  ds->boxes = reinterpret_cast<float*>(base + o_boxes);
  ds->rev = wiz_rev ? reinterpret_cast<float*>(base + o_rev) : nullptr;
  ds->keys2 = reinterpret_cast<unsigned long long*>(base + o_keys2);
  ds->sorted2 = wiz_rev ? reinterpret_cast<unsigned long long*>(base + o_sorted2) : nullptr;
  ds->sp_base = reinterpret_cast<int*>(base + o_sp);
  ds->sp_lim = ds->sp_base + B;
  ds->pk_base = ds->sp_lim + 2;
  ds->pk_lim = ds->pk_base + 2 * B;
  ds->wiz_rev = wiz_rev;
  static bool attr_done = false;
  if (!attr_done) {
    PT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lore_sort_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, CAP * 8));
    attr_done = true;
  }
  PT_HIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)B * 5 * 4, s));
  PT_HIP_CHECK(hipMemsetAsync(d_counts, 0, (size_t)B * 4, s));
  hipLaunchKernelGGL(lore_sigmoid_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, hm, sig, (long long)npix);
  // with wiz_rev the snap loop runs for cells >= 0.2 (:190) and the final filter is vis_thresh (:568-571): cells below
  // vis_thresh can never reach the output, whatever the snap does
  hipLaunchKernelGGL(lore_peaks_kernel<1024>, dim3((unsigned)((npix + 1023) / 1024)), dim3(1024), 0, s, sig, B, H, W, vis_thresh,
                     0.3f, keys, cnt);
  hipLaunchKernelGGL(lore_sort_kernel, dim3(2 * B), dim3(1024), CAP * 8, s, keys, cnt, 1, CAP, K_CELLS, K_CORNERS, ds->sorted,
                     CAP, cnt + 2 * B);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// boxes from the reg / wh / st heads (dense maps, or their values on the peak-patch mosaics), vertex snapping, final order
static int decode_boxes(const DecodeState* ds, const float* reg0, const float* reg1, const float* wh, const float* st,
                        bool sparse, int B, int H, int W, hipStream_t s) {
  int* cnt = ds->cnt;
  hipLaunchKernelGGL(lore_boxes_kernel, dim3((K_CORNERS + 255) / 256, 2 * B), dim3(256), 0, s, ds->sorted, cnt + 2 * B, CAP, B,
                     H, W, reg0, reg1, wh, st, sparse ? (const int*)ds->pk_base : (const int*)nullptr, ds->boxes);
  if (ds->wiz_rev) {
    hipLaunchKernelGGL(lore_snap_kernel, dim3(K_CELLS, B), dim3(64), 0, s, ds->boxes, cnt + 2 * B, CAP, ds->rev);
    hipLaunchKernelGGL(lore_rekey_kernel, dim3((K_CELLS + 255) / 256, B), dim3(256), 0, s, ds->rev, cnt + 2 * B, CAP, ds->keys2);
    // one list per table; its length is the kept cell count (the even entries of the kept counts)
    hipLaunchKernelGGL(lore_sort_kernel, dim3(B), dim3(1024), CAP * 8, s, ds->keys2, cnt + 2 * B, 2, CAP, K_CELLS, K_CELLS,
                       ds->sorted2, CAP, cnt + 4 * B);
  }
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_lore_decode(pt_engine* e, const float* hm, const float* st, const float* wh, const float* ax, const float* cr,
                   const float* reg, int B, int H, int W, int wiz_rev, float vis_thresh, int* d_counts, float* d_dets,
                   float* d_logi, hipStream_t s) {
  PT_REQUIRE(hm && st && wh && ax && cr && reg && d_counts && d_dets && d_logi && B > 0, "tsr decode: null pointer");
  PtProfScope ps(e, s, PT_PROF_OTHER, 0, "tsr decode");
  DecodeState ds;
  int rc = decode_peaks(e, hm, B, H, W, wiz_rev, vis_thresh, d_counts, &ds, s);
  if (rc == PT_OK) rc = decode_boxes(&ds, reg, reg, wh, st, false, B, H, W, s);
  if (rc != PT_OK) return rc;
  hipLaunchKernelGGL(lore_gather_kernel, dim3(K_CELLS, B), dim3(256), 0, s, ds.rev, ds.boxes, ds.sorted2, ds.cnt + 2 * B, CAP, H,
                     W, ax, cr, vis_thresh, d_dets, d_logi, d_counts, (const int*)nullptr);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

// ---- sparse heads.  Of the six head maps the decode needs only `hm` everywhere: `reg`, `wh`, `st` are read at the kept
// peaks (<= 3000 cells + 5000 corners per table), `ax` / `cr` at the final cells' centres and corner pixels (<= 5 x 3000
// positions).  The fused forward+decode therefore evaluates five heads on 3x3-pixel patches around exactly those
// positions, in two rounds around the head launches (lore_model.hip):
//   pt_lore_decode_peaks    sigmoid, peaks, first sort + row limits of the cell / corner mosaics
//   pt_lore_peak_patches    feature-map neighbourhoods of the kept peaks -> mosaics [3 * rows][768][C]
//   pt_lore_decode_boxes    boxes from the wh / st / reg outputs on the mosaics, snapping, final order + limits of round two
//   pt_lore_patch_gather    neighbourhoods of the final cells' centres / corners -> the ax / cr mosaics
//   pt_lore_decode_sparse   dets + logic features from the ax / cr outputs on the mosaics
// A patch's centre pixel sees only its own 3x3 pixels, through the same conv kernels in the same order as in the dense
// map: with one conv kernel family the results are bit-identical to pt_lore_decode's (tests/test_gpu_tsr.py).
void pt_lore_mosaic_rows(int B, int* rows_ax, int* rows_cr, int* rows_cell, int* rows_corner) {
  *rows_ax = (int)(((long long)B * K_CELLS + MOS_PW - 1) / MOS_PW);
  *rows_cr = (int)(((long long)B * K_CELLS * 4 + MOS_PW - 1) / MOS_PW);
  *rows_cell = *rows_ax;
  *rows_corner = (int)(((long long)B * K_CORNERS + MOS_PW - 1) / MOS_PW);
}

// state of the earlier steps, consumed by the calls that follow on the same stream: kept in the engine, so that two
// engines can interleave their decodes
static_assert(sizeof(DecodeState) <= sizeof(pt_engine::tsr_decode_state), "DecodeState outgrew its slot in pt_engine");
#define g_ds (*reinterpret_cast<DecodeState*>(e->tsr_decode_state))

int pt_lore_decode_peaks(pt_engine* e, const float* hm, int B, int H, int W, int wiz_rev, float vis_thresh, int* d_counts,
                         const int** d_lim_cell, const int** d_lim_corner, hipStream_t s) {
  PT_REQUIRE(hm && d_counts && B > 0, "tsr decode peaks: null pointer");
  PtProfScope ps(e, s, PT_PROF_OTHER, 0, "tsr decode");
  int rc = decode_peaks(e, hm, B, H, W, wiz_rev, vis_thresh, d_counts, &g_ds, s);
  if (rc != PT_OK) return rc;
  hipLaunchKernelGGL(lore_peak_base_kernel, dim3(1), dim3(1), 0, s, g_ds.cnt + 2 * B, B, g_ds.pk_base, g_ds.pk_lim);
  PT_HIP_CHECK(hipGetLastError());
  *d_lim_cell = g_ds.pk_lim;
  *d_lim_corner = g_ds.pk_lim + 1;
  return PT_OK;
}

int pt_lore_peak_patches(pt_engine* e, const bf16_t* feat, int B, int H, int W, int C, int split, bf16_t* mos_cell,
                         bf16_t* mos_corner, hipStream_t s) {
  PT_REQUIRE(feat && mos_cell && mos_corner && C % 8 == 0, "tsr peak patches: bad arguments");
  PtProfScope ps(e, s, PT_PROF_OTHER, 0, "tsr patch gather");
  hipLaunchKernelGGL(lore_peak_patch_kernel, dim3(K_CORNERS, 2 * B), dim3(64), 0, s, g_ds.sorted, g_ds.cnt + 2 * B, CAP, H, W, feat,
                     C, split, g_ds.pk_base, mos_cell, mos_corner);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_lore_decode_boxes(pt_engine* e, const float* reg_cell, const float* wh, const float* reg_corner, const float* st, int B,
                         int H, int W, const int** d_lim_ax, const int** d_lim_cr, hipStream_t s) {
  PT_REQUIRE(reg_cell && wh && reg_corner && st, "tsr decode boxes: null pointer");
  PtProfScope ps(e, s, PT_PROF_OTHER, 0, "tsr decode");
  int rc = decode_boxes(&g_ds, reg_cell, reg_corner, wh, st, true, B, H, W, s);
  if (rc != PT_OK) return rc;
  hipLaunchKernelGGL(lore_sparse_base_kernel, dim3(1), dim3(1), 0, s, g_ds.cnt + 2 * B, B, g_ds.sp_base, g_ds.sp_lim);
  PT_HIP_CHECK(hipGetLastError());
  *d_lim_ax = g_ds.sp_lim;
  *d_lim_cr = g_ds.sp_lim + 1;
  return PT_OK;
}

int pt_lore_patch_gather(pt_engine* e, const bf16_t* feat, int B, int H, int W, int C, int split, bf16_t* mos_ax, bf16_t* mos_cr,
                         hipStream_t s) {
  PT_REQUIRE(feat && mos_ax && mos_cr && C % 8 == 0, "tsr patch gather: bad arguments");
  PtProfScope ps(e, s, PT_PROF_OTHER, 0, "tsr patch gather");
  hipLaunchKernelGGL(lore_patch_gather_kernel, dim3(K_CELLS, B), dim3(256), 0, s, g_ds.rev, g_ds.boxes, g_ds.sorted2,
                     g_ds.cnt + 2 * B, CAP, H, W, feat, C, split, g_ds.sp_base, mos_ax, mos_cr);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_lore_decode_sparse(pt_engine* e, const float* ax_mos, const float* cr_mos, int B, int H, int W, float vis_thresh,
                          int* d_counts, float* d_dets, float* d_logi, hipStream_t s) {
  PT_REQUIRE(ax_mos && cr_mos && d_counts && d_dets && d_logi, "tsr decode sparse: null pointer");
  PtProfScope ps(e, s, PT_PROF_OTHER, 0, "tsr decode");
  hipLaunchKernelGGL(lore_gather_kernel, dim3(K_CELLS, B), dim3(256), 0, s, g_ds.rev, g_ds.boxes, g_ds.sorted2, g_ds.cnt + 2 * B,
                     CAP, H, W, ax_mos, cr_mos, vis_thresh, d_dets, d_logi, d_counts, (const int*)g_ds.sp_base);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

}  // namespace PT_FMT_NS
