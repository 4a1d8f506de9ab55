// cls_kernels.hip -- pre-processing of the PP-LCNet image classifiers (model/cls/image_processing_pplcnet.py:327-455):
// Pillow's bilinear Image.resize (what transformers.image_transforms.resize calls) + * 1/255 + (x - mean) / std, from
// RGB uint8 images that are already in HBM (whole pages, or the ragged text-line crops the recognition stage cuts) to the
// bf16 NHWC4 network input.
//
// Pillow's ImagingResample (libImaging/Resample.c) is integer work after the coefficients and is reproduced bit for bit:
//   * per output coordinate xx: centre = (xx + 0.5) * scale, support = max(scale, 1), taps [int(centre - support + 0.5),
//     int(centre + support + 0.5)) clipped to the image, triangle weights normalised in fp64 and quantised to 22
//     fractional bits with (int)(0.5 + k * 2^22);
//   * horizontal pass first, 8-bit intermediate: clip8((2^21 + sum px * k) >> 22), then the vertical pass the same way;
//   * a pass whose input and output sizes are equal is skipped.
// One workgroup = one image x RT output rows.  The coefficient tables of the image (all output columns; the RT rows) are
// built in LDS by the workgroup itself, then every thread produces output pixels; the horizontal sums of an input row are
// recomputed for each output row that taps it (2-11 times) instead of staging a ragged intermediate image in HBM.
#include "common.h"

namespace PT_FMT_NS {

namespace {

constexpr int RT = 8;            // output rows per workgroup
constexpr int PBITS = 22;        // PRECISION_BITS = 32 - 8 - 2

__device__ __forceinline__ uint32_t f2bf_(float f) { return f32_to_a16(f); }      // act16.h: the storage format of this namespace

// precompute_coeffs + normalize_coeffs_8bpc for output coordinate xx; k must hold ksize ints
__device__ void coeffs(int in_size, int out_size, int xx, int ksize, int* xmin_out, int* n_out, int* k) {
  const double scale = (double)in_size / (double)out_size;
  const double fs = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * fs, ss = 1.0 / fs;
  const double center = ((double)xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double a = ((double)(x + xmin) - center + 0.5) * ss;
    if (a < 0.0) a = -a;
    ww += a < 1.0 ? 1.0 - a : 0.0;
  }
  for (int x = 0; x < ksize; ++x) {
    double w = 0.0;
    if (x < xmax) {
      double a = ((double)(x + xmin) - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      w = a < 1.0 ? 1.0 - a : 0.0;
      if (ww != 0.0) w = w / ww;
    }
    k[x] = (int)(0.5 + w * (double)(1 << PBITS));
  }
  *xmin_out = xmin;
  *n_out = xmax;
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PBITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// images: per image {byte offset from base, h, w}; out: bf16 [n, OH, OW, 4] ([hi rgb0 | lo rgb0], 8 wide, when split)
// lut: fp32 [3][256] = ((v * 1/255) - mean[c]) / std[c]; dynamic LDS: OW * (ksx + 2) + RT * (ksy + 2) ints
__global__ __launch_bounds__(256) void cls_resize_norm_kernel(const uint8_t* __restrict__ base,
                                                              const pt_cls_image* __restrict__ images, int OH, int OW,
                                                              int ksx, int ksy, const float* __restrict__ lut, int split,
                                                              bf16_t* __restrict__ out) {
  a16_kernel_enter();
  extern __shared__ int s_tab[];
  int* kx = s_tab;                      // [OW][ksx]
  int* bx = kx + OW * ksx;              // [OW][2] xmin, count
  int* ky = bx + OW * 2;                // [RT][ksy]
  int* by = ky + RT * ksy;              // [RT][2]
  const int img = blockIdx.y, y0 = blockIdx.x * RT, tid = threadIdx.x;
  const pt_cls_image d = images[img];
  const int h = d.h, w = d.w;
  const uint8_t* src = base + d.offset;
  for (int xx = tid; xx < OW; xx += 256) coeffs(w, OW, xx, ksx, &bx[2 * xx], &bx[2 * xx + 1], kx + xx * ksx);
  for (int r = tid; r < RT; r += 256)
    if (y0 + r < OH) coeffs(h, OH, y0 + r, ksy, &by[2 * r], &by[2 * r + 1], ky + r * ksy);
  __syncthreads();
  const bool pass_x = w != OW, pass_y = h != OH;
  const int ps = split ? 8 : 4;
  for (int i = tid; i < RT * OW; i += 256) {
    const int r = i / OW, xx = i - r * OW, yy = y0 + r;
    if (yy >= OH) continue;
    const int xmin = bx[2 * xx], xn = bx[2 * xx + 1], ymin = by[2 * r], yn = by[2 * r + 1];
    const int* cx = kx + xx * ksx;
    const int* cy = ky + r * ksy;
    int v[3];
    if (pass_y) {
      int acc[3] = {1 << (PBITS - 1), 1 << (PBITS - 1), 1 << (PBITS - 1)};
      for (int yi = 0; yi < yn; ++yi) {
        const uint8_t* row = src + (size_t)(ymin + yi) * w * 3;
        int hv[3];
        if (pass_x) {
          int a0 = 1 << (PBITS - 1), a1 = a0, a2 = a0;
          for (int xi = 0; xi < xn; ++xi) {
            const uint8_t* p = row + (size_t)(xmin + xi) * 3;
            const int k = cx[xi];
            a0 += p[0] * k; a1 += p[1] * k; a2 += p[2] * k;
          }
          hv[0] = clip8(a0); hv[1] = clip8(a1); hv[2] = clip8(a2);
        } else {
          const uint8_t* p = row + (size_t)xx * 3;
          hv[0] = p[0]; hv[1] = p[1]; hv[2] = p[2];
        }
        const int k = cy[yi];
        acc[0] += hv[0] * k; acc[1] += hv[1] * k; acc[2] += hv[2] * k;
      }
      v[0] = clip8(acc[0]); v[1] = clip8(acc[1]); v[2] = clip8(acc[2]);
    } else {
      const uint8_t* row = src + (size_t)yy * w * 3;
      if (pass_x) {
        int a0 = 1 << (PBITS - 1), a1 = a0, a2 = a0;
        for (int xi = 0; xi < xn; ++xi) {
          const uint8_t* p = row + (size_t)(xmin + xi) * 3;
          const int k = cx[xi];
          a0 += p[0] * k; a1 += p[1] * k; a2 += p[2] * k;
        }
        v[0] = clip8(a0); v[1] = clip8(a1); v[2] = clip8(a2);
      } else {
        const uint8_t* p = row + (size_t)xx * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
      }
    }
    bf16_t* o = out + (((size_t)img * OH + yy) * OW + xx) * ps;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float f = lut[c * 256 + v[c]];
      const uint32_t hb = f2bf_(f);
      o[c] = (bf16_t)hb;
      if (split) o[4 + c] = (bf16_t)f2bf_(f - a16_to_f32(hb));
    }
    o[3] = 0;
    if (split) o[7] = 0;
  }
}

// text-line crops of the recognition stage (ragged uint8 RGB, pixel offsets off[]) -> image descriptors
__global__ void cls_desc_from_lines_kernel(const pt_rec_line* __restrict__ lines, const long long* __restrict__ off, int n,
                                           pt_cls_image* __restrict__ images) {
  a16_kernel_enter();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pt_cls_image d;
  d.offset = off[i] * 3;
  d.h = lines[i].crop_h > 0 ? lines[i].crop_h : 1;
  d.w = lines[i].crop_w > 0 ? lines[i].crop_w : 1;
  images[i] = d;
}

inline int ksize_for(int in_size, int out_size) {
  double scale = (double)in_size / (double)out_size;
  if (scale < 1.0) scale = 1.0;
  return (int)ceil(scale) * 2 + 1;
}

}  // namespace

// max_h / max_w: upper bounds of the images' sizes (they size the coefficient tables)
int pt_launch_cls_resize_norm(const uint8_t* base, const pt_cls_image* images, int n, int max_h, int max_w, int OH, int OW,
                              const float* lut, int split, bf16_t* out, hipStream_t s) {
  PT_REQUIRE(base && images && lut && out && n > 0 && OH > 0 && OW > 0 && max_h > 0 && max_w > 0, "cls resize: bad arguments");
  const int ksx = ksize_for(max_w, OW), ksy = ksize_for(max_h, OH);
  const size_t smem = ((size_t)OW * (ksx + 2) + (size_t)RT * (ksy + 2)) * sizeof(int);
  PT_REQUIRE(smem <= 64 * 1024, "cls resize: %dx%d -> %dx%d needs %zu bytes of coefficient tables (limit 64 KB)", max_h, max_w,
             OH, OW, smem);
  hipLaunchKernelGGL(cls_resize_norm_kernel, dim3((OH + RT - 1) / RT, n), dim3(256), smem, s, base, images, OH, OW, ksx, ksy, lut,
                     split, out);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

int pt_launch_cls_desc_from_lines(const pt_rec_line* lines, const long long* off, int n, pt_cls_image* images, hipStream_t s) {
  hipLaunchKernelGGL(cls_desc_from_lines_kernel, dim3((n + 255) / 256), dim3(256), 0, s, lines, off, n, images);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

}  // namespace PT_FMT_NS
