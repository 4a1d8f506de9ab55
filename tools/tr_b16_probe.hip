// tools/tr_b16_probe.hip -- semantics of gfx950's ds_read_b64_tr_b16 with ARBITRARY per-lane addresses, checked on the GPU against the
// model dcn_mfma_kernel relies on:   within each 16-lane group, lane i receives element j = the (i & 3)-th bf16 of the 8 bytes that lane
// 4 j + (i >> 2) of the group addressed.   hipcc --offload-arch=gfx950 -O2 tools/tr_b16_probe.hip -o tools/scratch/tr_probe && tools/scratch/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(const int* offs, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)(i * 7 + 3);
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)((char*)lds + offs[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  int* d_off; unsigned short* d_out;
  hipMalloc(&d_off, 64 * 4); hipMalloc(&d_out, 256 * 2);
  int bad = 0;
  for (int trial = 0; trial < 200; ++trial) {
    std::vector<int> off(64);
    for (int l = 0; l < 64; ++l) off[l] = trial == 0 ? l * 8 : (rand() % 4096) * 8;      // 8-byte aligned, anywhere in 32 KB
    hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_off, d_out);
    std::vector<unsigned short> out(256);
    hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
      const int g = l & ~15, i = l & 15;
      for (int j = 0; j < 4; ++j) {
        const int src = g + 4 * j + (i >> 2);
        const unsigned short want = (unsigned short)((off[src] / 2 + (i & 3)) * 7 + 3);
        if (out[l * 4 + j] != want) { if (bad < 8) printf("trial %d lane %d elem %d: got %u want %u\n", trial, l, j, out[l * 4 + j], want); ++bad; }
      }
    }
  }
  printf(bad ? "tr_b16 model MISMATCH (%d)\n" : "tr_b16 model OK (%d mismatches over 200 trials)\n", bad);
  return bad != 0;
}
