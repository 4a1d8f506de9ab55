// Ceiling probe for the conv kernels' inner loop on one MI355X (not part of the product):
//   A: back-to-back v_mfma_f32_32x32x16_bf16 from registers (4 independent accumulators per wave)
//   B: the same with every operand fragment re-read from LDS (1 ds_read_b128 per MFMA, the conv kernels' ratio)
//   C: B with half the fragment reads (2 MFMAs per ds_read_b128, the 128-pixel-per-wave tiling)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/scratch/mfma_ceiling tools/mfma_ceiling.hip ; run: mfma_ceiling [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ uint4 lds[4096];                       // 64 KB
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  // 80-byte row stride like the kernels' pixel stride (conflict-free for ds_read_b128)
  const char* base = reinterpret_cast<const char*>(lds) + wave * 8192 + (lane & 31) * 80 + (lane >> 5) * 16;
  bf16x8 a0 = *reinterpret_cast<const bf16x8*>(base), a1 = *reinterpret_cast<const bf16x8*>(base + 2560);
  bf16x8 b0 = *reinterpret_cast<const bf16x8*>(base + 32), b1 = *reinterpret_cast<const bf16x8*>(base + 2592);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 1) {
        const int o = ((it + u) & 7) * 64;
        a0 = *reinterpret_cast<const bf16x8*>(base + o);
        a1 = *reinterpret_cast<const bf16x8*>(base + 2560 + o);
        b0 = *reinterpret_cast<const bf16x8*>(base + 32 + o);
        b1 = *reinterpret_cast<const bf16x8*>(base + 2592 + o);
      } else if (MODE == 2) {
        const int o = ((it + u) & 7) * 64;
        if (u & 1) { a0 = *reinterpret_cast<const bf16x8*>(base + o); a1 = *reinterpret_cast<const bf16x8*>(base + 2560 + o); }
        else       { b0 = *reinterpret_cast<const bf16x8*>(base + 32 + o); b1 = *reinterpret_cast<const bf16x8*>(base + 2592 + o); }
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int blocks, int iters, const uint4* src, float* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<MODE><<<blocks, 256>>>(src, out, iters);
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    probe<MODE><<<blocks, 256>>>(src, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = double(blocks) * 4 * iters * 8 * 4 * 32768.0;
    printf("%-44s blocks=%4d  %8.3f ms  %7.0f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
  }
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 2;
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  uint4* src; float* out;
  CK(hipMalloc(&src, 65536)); CK(hipMalloc(&out, 256 * 8 * 256 * 4));
  std::vector<unsigned short> h(32768);
  const int rnd = argc > 3 ? atoi(argv[3]) : 0;
  unsigned st = 12345u;
  for (int i = 0; i < 32768; ++i) {
    st = st * 1664525u + 1013904223u;
    // rnd: random sign / mantissa / 4 exponent bits around 1.0 (like activations); else a low-entropy pattern
    h[i] = rnd ? (unsigned short)(((st >> 16) & 0x83ff) | (0x3c00 + (((st >> 8) & 7) << 7))) : (unsigned short)(0x3c00 + (i * 37 % 64));
  }
  CK(hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice));
  const int blocks = 256 * wps;
  run<0>("A mfma only (registers)", blocks, iters, src, out);
  run<1>("B mfma + 1 ds_read_b128 per mfma", blocks, iters, src, out);
  run<2>("C mfma + 1 ds_read_b128 per 2 mfma", blocks, iters, src, out);
  return 0;
}
