// tools/gather_rate_bench.hip -- what the vector-memory path delivers for the deformable convolution's corner gather, by transport:
//   A  global_load_lds_dwordx4, 16 rows of 64 B per instruction (dcn_mfma_kernel's slots: half a channel slice per wave), DEPTH instructions in flight
//   B  global_load_lds_dwordx4, 8 rows of 128 B per instruction (a full 64-channel line per eight lanes)
//   C  global_load_dwordx4 into registers, 8 rows of 128 B per instruction, eight loads in flight per wave (dcn_fused64_kernel's gather)
// Every row is a pixel of a [T][256][256][64] bf16 map at a pseudo-random position within +-R pixels of the wave's own spot in an 8 x 16 tile -- the
// footprint of the bench's Lore offsets (R = 9) or of a trained net's (R = 2).  Nothing else runs: no blend, no product.  16 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_rate_bench.hip -o tools/scratch/gather_rate_bench && tools/scratch/gather_rate_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) uint32_t u4;

__device__ __forceinline__ uint32_t hash(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
  return a;
}

// MODE 0: A, MODE 1: B (LDS-DMA); MODE 2: C (registers).  One workgroup = one 8 x 16 tile, 8 waves; STEPS instructions per wave.
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 2) void gather_kernel(const char* __restrict__ x, uint32_t* __restrict__ out, int tiles_per_map, int R, int steps) {
  __shared__ __attribute__((aligned(16))) char ring[8 * 4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x % tiles_per_map, map = blockIdx.x / tiles_per_map;
  const int ty0 = 24 + (tile / 13) * 8, tx0 = 24 + (tile % 13) * 16;      // interior tiles: rows 24 .. 231, columns 24 .. 231 (no clamping up to R = 24)
  const char* xm = x + (size_t)map * 256 * 256 * 128;
  const int rows_per = MODE == 0 ? 16 : 8;                    // rows per instruction
  const int row = MODE == 0 ? lane >> 2 : lane >> 3;
  const unsigned piece = MODE == 0 ? (unsigned)((wave >> 2) * 64 + (lane & 3) * 16) : (unsigned)((lane & 7) * 16);
  const unsigned ring_lds = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(ring + wave * 4096));
  u4 acc = {0, 0, 0, 0};
  u4 pend[8];
  // the row's pixel: the wave's pixel tile (wave & 3) x a pixel inside it that advances with the step, displaced by a pseudo-random (dy, dx) in
  // [-R, R] (a per-lane LCG: ~12 VALU instructions per address, so that the address arithmetic is far from being the limit); tiles at the map's
  // border are not launched, so no clamping
  uint32_t lcg = hash((uint32_t)(blockIdx.x * 9176 + row * 7 + (MODE == 0 ? 0 : wave * 977)));
  const unsigned span = (unsigned)(2 * R + 1);
  const int base_y = ty0 + (wave & 3) * 2 - R, base_x = tx0 - R;
  auto addr = [&](int s) -> unsigned {
    lcg = lcg * 1664525u + 1013904223u;
    const unsigned dy = ((lcg >> 24) * span) >> 8, dx = (((lcg >> 12) & 255u) * span) >> 8;
    const int pl = (s * rows_per + row) >> 2;                   // 0 .. 31 within the wave's two tile rows, cyclic
    const unsigned y = (unsigned)(base_y + ((pl >> 4) & 1)) + dy, xq = (unsigned)(base_x + (pl & 15)) + dx;
    return ((y << 8) + xq) * 128u + piece;
  };
  if (MODE < 2) {
    for (int s = 0; s < DEPTH && s < steps; ++s)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xm + addr(s)),
                                       (__attribute__((address_space(3))) void*)(size_t)(ring_lds + (unsigned)((s & 3) * 1024)), 16, 0, 0);
    for (int s = 0; s < steps; ++s) {
      if (s + DEPTH < steps)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xm + addr(s + DEPTH)),
                                         (__attribute__((address_space(3))) void*)(size_t)(ring_lds + (unsigned)(((s + DEPTH) & 3) * 1024)), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = *reinterpret_cast<const u4*>(ring + wave * 4096 + lane * 16);
  } else {
    for (int s0 = 0; s0 < steps; s0 += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) pend[j] = *reinterpret_cast<const u4*>(xm + addr(s0 + j));
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = acc ^ pend[j];
    }
  }
  out[(size_t)blockIdx.x * 512 + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* x, uint32_t* out, int maps, int R) {
  const int tiles_per_map = 26 * 13, steps = 72 * 2;      // 9 taps x 8 steps x 2 (a 128-channel layer's worth of instructions per wave)
  const int grid = maps * tiles_per_map;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  gather_kernel<MODE, DEPTH><<<grid, 512>>>(x, out, tiles_per_map, R, steps);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) gather_kernel<MODE, DEPTH><<<grid, 512>>>(x, out, tiles_per_map, R, steps);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 3;
  const double bytes = (double)grid * 8 * steps * 1024;
  printf("R = %2d  %-58s %8.3f ms  %7.2f TB/s  %6.1f GB/s per CU  (%5.1f B/clk/CU at 2.1 GHz)\n", R, name, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256,
         bytes / ms / 1e6 / 256 / 2.1);
}

int main() {
  const int maps = 24;
  char* x;
  uint32_t* out;
  CK(hipMalloc(&x, (size_t)maps * 256 * 256 * 128));
  CK(hipMemset(x, 1, (size_t)maps * 256 * 256 * 128));
  CK(hipMalloc(&out, (size_t)maps * 512 * 512 * 4));
  for (int R : {2, 9, 16, 24}) {
    run<0, 3>("A  LDS-DMA, 16 x 64 B rows, 3 in flight per wave", x, out, maps, R);
    run<0, 7>("A  LDS-DMA, 16 x 64 B rows, 7 in flight (slots reused: rate only)", x, out, maps, R);
    run<0, 15>("A  LDS-DMA, 16 x 64 B rows, 15 in flight (rate only)", x, out, maps, R);
    run<1, 3>("B  LDS-DMA, 8 x 128 B rows, 3 in flight per wave", x, out, maps, R);
    run<1, 7>("B  LDS-DMA, 8 x 128 B rows, 7 in flight (rate only)", x, out, maps, R);
    run<1, 15>("B  LDS-DMA, 8 x 128 B rows, 15 in flight (rate only)", x, out, maps, R);
    run<2, 8>("C  registers, 8 x 128 B rows, 8 loads in flight per wave", x, out, maps, R);
  }
  return 0;
}
