// act16.h -- the 16-bit storage format of activations and single-pass weights: a COMPILE-TIME parameter of every kernel.
//
// Every translation unit of the library is compiled twice (pdf_table_amd/build.py):
//   namespace pt_bf16   bfloat16 (8 significant bits, fp32 range): PT_PRECISION_BF16, and the (hi | lo) pair modes BF16X3 / F16X2
//   namespace pt_f16    IEEE half (11 significant bits, |x| <= 65504), -DPT_ACT_F16=1: PT_PRECISION_F16 -- the reference's own GPU
//                       arithmetic (base_infer_task.py:56-57 precision="fp16", utils/deploy_utils.py:227-240 model.half())
// Same kernels, same tiles, same bytes and the same MFMA rate (v_mfma_f32_32x32x16_f16 == ..._bf16); what changes is this file: how 16 stored
// bits become an fp32 value, how an fp32 value is rounded for storage (round-to-nearest-even in both; the half format SATURATES at +-65504
// instead of producing Inf: MODE.FP16_OVFL, set by a16_kernel_enter() at the top of every kernel), and which matrix instruction multiplies them.  The exported C functions (api_dispatch.cpp, generated from
// include/pdftable_hip.h) pick the namespace from pt_engine::precision.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef PT_ACT_F16
#define PT_ACT_F16 0
#endif
#if PT_ACT_F16
#define PT_FMT_NS pt_f16
#define PT_FMT_NAME "f16"
#else
#define PT_FMT_NS pt_bf16
#define PT_FMT_NAME "bf16"
#endif

namespace PT_FMT_NS {

typedef __attribute__((ext_vector_type(2))) float a16_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 a16_bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 a16_f16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 a16_bf16x8;     // the kernels' 16-byte operand container (bit pattern only)
typedef __attribute__((ext_vector_type(8))) _Float16 a16_f16x8;
typedef __attribute__((ext_vector_type(16))) float a16_f32x16;
typedef __attribute__((ext_vector_type(4))) float a16_f32x4;

#if PT_ACT_F16
#define PT_A16_MAX 65504.0f
// bits of 1.0 / the most negative finite value in the storage format (pool identities, canvas fills)
#define PT_A16_ONE 0x3C00u
#define PT_A16_LOWEST 0xFBFFu
// First statement of EVERY kernel of this namespace: MODE.FP16_OVFL = 1 -- "an overflowed FP16 VALU result is clamped to +/- MAX_FP16 regardless of the
// round mode" (the bit exists for exactly this) -- so v_cvt_f16_f32 / v_cvt_pk_f16_f32 SATURATE at +-65504 instead of producing Inf, at no instruction per
// stored value (two v_med3_f32 clamps per pair cost 3 % of the four-stage step: 674 -> pages/s in profiles/r05/f16_ovfl.txt).  Measured on gfx950
// (tools/scratch/ovfl.hip, recorded in the same file): 70000 -> 0x7bff, -1e9 -> 0xfbff, 65520 -> 0x7bff with the bit, Inf without.  MODE is per wave and
// starts from the kernel descriptor (bit clear), so a kernel that skips this call would store Inf again: tests/test_gpu_f16.py drives the stores of every
// kernel family over the edge.
// The "memory" clobber keeps every later store (and the conversions feeding it, which the compiler schedules with their stores) behind the mode write; the
// build refuses a translation unit whose count of __global__ differs from its count of a16_kernel_enter() (build.py: lint_kernel_enter).
// DEVIATION from the reference, stated here where the precision is defined: its model.half() (utils/deploy_utils.py:227-240) produces +-Inf beyond 65504;
// this mode stores +-65504 instead (an Inf would turn into NaN at the next subtraction or 0 x Inf and poison the rest of the page).
__device__ __forceinline__ void a16_kernel_enter() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1" ::: "memory"); }
__device__ __forceinline__ float a16_sat(float f) { return __builtin_amdgcn_fmed3f(f, -PT_A16_MAX, PT_A16_MAX); }      // explicit clamp (fp32 side), where one is wanted
__device__ __forceinline__ float a16_to_f32(uint32_t bits16) { return (float)__builtin_bit_cast(_Float16, (uint16_t)bits16); }
__device__ __forceinline__ float a16lo_f32(uint32_t pk) { return (float)__builtin_bit_cast(a16_f16x2, pk).x; }
__device__ __forceinline__ float a16hi_f32(uint32_t pk) { return (float)__builtin_bit_cast(a16_f16x2, pk).y; }
// round-to-nearest-even; saturating through MODE.FP16_OVFL (a16_kernel_enter): an activation beyond 65504 is stored as 65504, never as Inf
__device__ __forceinline__ uint32_t f32_to_a16(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
// two values -> one dword: v_cvt_pk_f16_f32
__device__ __forceinline__ uint32_t pack_a16x2(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(a16_f32x2{a, b}, a16_f16x2));
}
__device__ __forceinline__ a16_f32x16 mfma_32x32x16_a16(a16_bf16x8 a, a16_bf16x8 b, a16_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(a16_f16x8, a), __builtin_bit_cast(a16_f16x8, b), c, 0, 0, 0);
}
// 16 x 16 outputs, K = 32 (lane l: row / column l & 15, k = 8 (l >> 4) .. + 7; D: column l & 15, rows 4 (l >> 4) .. + 3): thin layers with 16 output channels
__device__ __forceinline__ a16_f32x4 mfma_16x16x32_a16(a16_bf16x8 a, a16_bf16x8 b, a16_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(a16_f16x8, a), __builtin_bit_cast(a16_f16x8, b), c, 0, 0, 0);
}
// Weighted sum of four stored dwords (two values each) with fp32 weights, fp32 accumulation, one rounding: ((w0 c0 + w1 c1) + w2 c2) + w3 c3, fused.
// The half format has mixed-precision FMAs that read a 16-bit half of a register directly (v_fma_mix_f32, op_sel picks the half) and write a rounded half
// of the result (v_fma_mixlo / mixhi_f16): eight instructions per dword where unpack + packed fp32 FMA + pack takes thirteen -- the deformable conv's blend
// is bound by exactly this instruction count (profiles/r05/experiments.txt).  Saturation: MODE.FP16_OVFL covers the mixlo / mixhi results like any VALU half.
#define PT_A16_HAS_MIX_BLEND 1
__device__ __forceinline__ uint32_t a16_blend4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, float w0, float w1, float w2, float w3) {
  float lo, hi;
  uint32_t o;
  asm("v_fma_mix_f32 %0, %3, %7, 0 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %3, %7, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %0, %4, %8, %0 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %4, %8, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %0, %5, %9, %0 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %5, %9, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %2, %6, %10, %0 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %2, %6, %10, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(lo), "=&v"(hi), "=&v"(o)
      : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
  return o;
}
#else
#define PT_A16_ONE 0x3F80u
#define PT_A16_LOWEST 0xFF7Fu
__device__ __forceinline__ void a16_kernel_enter() {}       // bf16 has the fp32 range: nothing to set up (see the pt_f16 side)
__device__ __forceinline__ float a16_sat(float f) { return f; }
__device__ __forceinline__ float a16_to_f32(uint32_t bits16) { return __uint_as_float(bits16 << 16); }
__device__ __forceinline__ float a16lo_f32(uint32_t pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float a16hi_f32(uint32_t pk) { return __uint_as_float(pk & 0xFFFF0000u); }
// round-to-nearest-even fp32 -> bf16 bits (inputs are finite on this path)
__device__ __forceinline__ uint32_t f32_to_a16(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
// two values -> one dword: v_cvt_pk_bf16_f32 (equal to f32_to_a16 for finite values)
__device__ __forceinline__ uint32_t pack_a16x2(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(a16_f32x2{a, b}, a16_bf16x2));
}
__device__ __forceinline__ a16_f32x16 mfma_32x32x16_a16(a16_bf16x8 a, a16_bf16x8 b, a16_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ a16_f32x4 mfma_16x16x32_a16(a16_bf16x8 a, a16_bf16x8 b, a16_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#define PT_A16_HAS_MIX_BLEND 0      // no mixed-precision FMA reads bf16: callers keep their unpack / packed-FMA / pack sequence
__device__ __forceinline__ uint32_t a16_blend4(uint32_t, uint32_t, uint32_t, uint32_t, float, float, float, float) { return 0u; }
#endif

}  // namespace PT_FMT_NS
