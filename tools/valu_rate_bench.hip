// tools/valu_rate_bench.hip -- issue rates of the instructions the deformable convolution's bilinear blend is (or could be) made of, on
// gfx950: the blend of dcn_fused64_kernel is VALU-bound (profiles/r03/dcn_counters.txt: 67 % VALU busy, matrix pipe 10 %), and which
// rewrite pays depends on numbers the guides do not hold -- v_dot2_f32_bf16 (an FMA straight from a packed bf16 operand: no unpack),
// v_pk_fma_f32 vs two v_fma_f32, v_perm_b32, v_cvt_pk_bf16_f32, the 4x4x4 MFMA, ds_read_b64_tr_b16.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_bench.hip -o tools/scratch/valu_rate_bench && tools/scratch/valu_rate_bench
// Each kernel runs REP x 64 independent instructions of one kind per wave (8 rotating destinations) between two s_memtime reads; printed:
// shader cycles per wave-instruction per SIMD at 1 / 2 / 4 waves per SIMD (= wall cycles of the slowest wave x / instructions issued on
// that SIMD).  Then the two candidate blend bodies as the compiler schedules them: cycles per (pixel, 8-channel piece).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) __bf16 b2;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;

#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define R64(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I)

enum { K_FMA, K_PKFMA, K_SHIFT, K_AND, K_DOT2, K_DOT2C, K_CVTPK, K_PERM, K_MFMA444, K_TR, K_NKIND };
static const char* kind_name[K_NKIND] = {"v_fma_f32", "v_pk_fma_f32", "v_lshlrev_b32", "v_and_b32", "v_dot2_f32_bf16", "v_dot2c_f32_bf16",
                                        "v_cvt_pk_bf16_f32", "v_perm_b32", "v_mfma_f32_4x4x4_16b_bf16", "ds_read_b64_tr_b16"};

template <int KIND>
__global__ __launch_bounds__(1024) void rate_kernel(uint32_t* out, long long* cyc, int rep, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * seed;
  __syncthreads();
  float a0 = seed * 1e-9f + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
  f4 m0 = {a0, a1, a2, a3}, m1 = m0, m2 = m0, m3 = m0, m4 = m0, m5 = m0, m6 = m0, m7 = m0;
  f2 t0 = p0, t1 = p0, t2 = p0, t3 = p0, t4 = p0, t5 = p0, t6 = p0, t7 = p0;
  const float x = 1.0001f, y = 0.5f;
  const f2 px = {x, x}, py = {y, y};
  const uint32_t ux = 0x3f803f80u + threadIdx.x, uy = 0x00003f80u;
  const uint32_t la = (threadIdx.x & 63) * 8;
  const long long t_beg = __builtin_readcyclecounter();
  for (int r = 0; r < rep; ++r) {
    if (KIND == K_FMA) {
#define I(n) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##n) : "v"(x), "v"(y));
      R64(I)
#undef I
    } else if (KIND == K_PKFMA) {
#define I(n) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p##n) : "v"(px), "v"(py));
      R64(I)
#undef I
    } else if (KIND == K_SHIFT) {
#define I(n) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(a##n) : "v"(ux));
      R64(I)
#undef I
    } else if (KIND == K_AND) {
#define I(n) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(a##n) : "v"(ux));
      R64(I)
#undef I
    } else if (KIND == K_DOT2) {
#define I(n) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a##n) : "v"(ux), "v"(uy));
      R64(I)
#undef I
    } else if (KIND == K_DOT2C) {
#define I(n) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a##n) : "v"(ux), "v"(uy));
      R64(I)
#undef I
    } else if (KIND == K_CVTPK) {
#define I(n) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a##n) : "v"(x), "v"(y));
      R64(I)
#undef I
    } else if (KIND == K_PERM) {
#define I(n) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a##n) : "v"(ux), "v"(uy), "v"(0x07060302u));
      R64(I)
#undef I
    } else if (KIND == K_MFMA444) {
#define I(n) asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(m##n) : "v"(px), "v"(py));
      R64(I)
#undef I
    } else if (KIND == K_TR) {
#define I(n) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #n "*512" : "=v"(t##n) : "v"(la));
      R64(I)
#undef I
      asm volatile("s_waitcnt lgkmcnt(0)");
    }
  }
  const long long t_end = __builtin_readcyclecounter();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + m0.x + m1.y + m2.z + m3.w + m4.x +
            m5.x + m6.x + m7.x + t0.x + t1.x + t2.x + t3.x + t4.x + t5.x + t6.x + t7.x;
  out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s);
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t_end - t_beg;
}

// the two blend bodies, per thread: NITEM (pixel, 16-byte piece) items x 4 corners already in registers -> 16 bytes of blended bf16 per item
template <int MODE>
__global__ __launch_bounds__(512) void blend_kernel(const u4* __restrict__ src, const float* __restrict__ wsrc, u4* __restrict__ dst, long long* cyc, int rep) {
#pragma clang fp contract(fast)
  u4 c[4];
  float w[4];
  for (int k = 0; k < 4; ++k) { c[k] = src[threadIdx.x * 4 + k]; w[k] = wsrc[threadIdx.x * 4 + k]; }
  u4 acc = {0, 0, 0, 0};
  const long long t_beg = __builtin_readcyclecounter();
  for (int r = 0; r < rep; ++r) {
    uint32_t o[4];
    if (MODE == 0) {          // today: unpack to fp32 pairs, packed FMAs, one conversion per dword
      const f2 w0 = {w[0], w[0]}, w1 = {w[1], w[1]}, w2 = {w[2], w[2]}, w3 = {w[3], w[3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f2 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t u = c[k][e];
          const f2 cv = {__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u)};
          const f2 wk = k == 0 ? w0 : k == 1 ? w1 : k == 2 ? w2 : w3;
          v = k == 0 ? wk * cv : wk * cv + v;
        }
        o[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
      }
    } else if (MODE == 1) {   // plain FMAs instead of packed ones
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo = 0.f, hi = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t u = c[k][e];
          lo = __builtin_fmaf(w[k], __uint_as_float(u << 16), lo);
          hi = __builtin_fmaf(w[k], __uint_as_float(u & 0xFFFF0000u), hi);
        }
        const f2 v = {lo, hi};
        o[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
      }
    } else {                  // v_dot2_f32_bf16 with the weight as (w, 0) / (0, w): one instruction per (corner, channel), no unpack
      uint32_t wl[4], wh[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint32_t u = __float_as_uint(w[k]);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
        wl[k] = u;
        wh[k] = u << 16;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo = 0.f, hi = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t u = c[k][e];
          asm("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(lo) : "v"(u), "v"(wl[k]));
          asm("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(hi) : "v"(u), "v"(wh[k]));
        }
        const f2 v = {lo, hi};
        o[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
      }
    }
    acc = acc ^ u4{o[0], o[1], o[2], o[3]};
    for (int k = 0; k < 4; ++k) c[k] = c[k] + acc;       // next round depends on this one's output: nothing hoisted
  }
  const long long t_end = __builtin_readcyclecounter();
  dst[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t_end - t_beg;
}

template <int KIND>
static void run_kind(uint32_t* out, long long* cyc, std::vector<long long>& h) {
  const int rep = 200;
  printf("%-28s", kind_name[KIND]);
  for (int wps = 1; wps <= 4; wps *= 2) {        // waves per SIMD: one workgroup of 256 * wps threads per CU
    const int thr = 256 * wps;
    rate_kernel<KIND><<<256, thr>>>(out, cyc, rep, 12345u);
    rate_kernel<KIND><<<256, thr>>>(out, cyc, rep, 12345u);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), cyc, 256 * (thr / 64) * sizeof(long long), hipMemcpyDeviceToHost));
    long long mx = 0;
    for (int i = 0; i < 256 * (thr / 64); ++i) mx = h[i] > mx ? h[i] : mx;
    printf("  %d w/SIMD: %6.2f cyc", wps, (double)mx / ((double)rep * 64 * wps));
  }
  printf("\n");
}

int main() {
  uint32_t* out;
  long long* cyc;
  CK(hipMalloc(&out, 256 * 1024 * 16));
  CK(hipMalloc(&cyc, 256 * 16 * sizeof(long long)));
  std::vector<long long> h(256 * 16);
  printf("cycles per wave-instruction per SIMD (slowest wave's s_memtime span x waves per SIMD / instructions; s_memtime ticks at 100 MHz x ... see note)\n");
  run_kind<K_FMA>(out, cyc, h);
  run_kind<K_PKFMA>(out, cyc, h);
  run_kind<K_SHIFT>(out, cyc, h);
  run_kind<K_AND>(out, cyc, h);
  run_kind<K_DOT2>(out, cyc, h);
  run_kind<K_DOT2C>(out, cyc, h);
  run_kind<K_CVTPK>(out, cyc, h);
  run_kind<K_PERM>(out, cyc, h);
  run_kind<K_MFMA444>(out, cyc, h);
  run_kind<K_TR>(out, cyc, h);
  // wall-clock calibration of the counter: one long kernel, events around it
  {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    rate_kernel<K_FMA><<<256, 256>>>(out, cyc, 20000, 1u);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), cyc, 256 * 4 * sizeof(long long), hipMemcpyDeviceToHost));
    long long mx = 0;
    for (int i = 0; i < 1024; ++i) mx = h[i] > mx ? h[i] : mx;
    printf("calibration: %lld counter ticks in %.3f ms -> %.1f MHz counter; v_fma_f32 %.3f ns per wave-instruction per SIMD at one wave per SIMD\n", mx, ms,
           mx / (ms * 1e3), ms * 1e6 / (20000.0 * 64));
  }
  u4* src;
  float* wsrc;
  u4* dst;
  CK(hipMalloc(&src, 512 * 4 * sizeof(u4)));
  CK(hipMalloc(&wsrc, 512 * 4 * sizeof(float)));
  CK(hipMalloc(&dst, 256 * 2 * 512 * sizeof(u4)));
  std::vector<uint32_t> hs(512 * 16);
  std::vector<float> hw(512 * 4);
  for (size_t i = 0; i < hs.size(); ++i) hs[i] = 0x3f003e80u + (uint32_t)(i * 2654435761u >> 20 & 0x7f007f);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.25f + (i % 7) * 0.01f;
  CK(hipMemcpy(src, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wsrc, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  const char* mode_name[3] = {"unpack + v_pk_fma_f32 (today)", "unpack + v_fma_f32", "v_dot2_f32_bf16 (w, 0)/(0, w)"};
  for (int mode = 0; mode < 3; ++mode) {
    const int rep = 2000;
    for (int pass = 0; pass < 2; ++pass) {
      if (mode == 0) blend_kernel<0><<<512, 512>>>(src, wsrc, dst, cyc, rep);
      if (mode == 1) blend_kernel<1><<<512, 512>>>(src, wsrc, dst, cyc, rep);
      if (mode == 2) blend_kernel<2><<<512, 512>>>(src, wsrc, dst, cyc, rep);
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), cyc, 512 * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    long long mx = 0;
    for (int i = 0; i < 4096; ++i) mx = h[i] > mx ? h[i] : mx;
    // two 512-thread workgroups per CU = 4 waves per SIMD; an item = one (pixel, 16-byte piece) with its 4 corners
    printf("blend body %-32s %7.1f counter ticks per item-round per wave (4 waves per SIMD)\n", mode_name[mode], (double)mx / rep);
  }
  return 0;
}
