// tools/conv_harness.cpp -- one conv shape through the C ABI (pt_op_conv2d) without Python: starts in a fraction of a second, so that
// `rocprofv3 --pmc ... -- tools/conv_harness ...` passes take seconds (the torch-hosted tools/conv_bench.py takes minutes per counter pass).
//   hipcc -O2 -o tools/conv_harness tools/conv_harness.cpp -Iinclude -Lpdf_table_amd -lpdftable_hip -Wl,-rpath,'$ORIGIN/../pdf_table_amd'
//   tools/conv_harness B H W Cin N [ks] [stride] [iters] [res]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include <vector>

#include "pdftable_hip.h"

static uint16_t bf16_of(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s B H W Cin N [ks] [stride] [iters] [res]\n", argv[0]);
    return 2;
  }
  const int B = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), Cin = atoi(argv[4]), N = atoi(argv[5]);
  const int ks = argc > 6 ? atoi(argv[6]) : 3, stride = argc > 7 ? atoi(argv[7]) : 1, iters = argc > 8 ? atoi(argv[8]) : 20;
  const int with_res = argc > 9 ? atoi(argv[9]) : 0;
  const int pad = ks / 2, Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
  pt_engine* e = nullptr;
  if (pt_engine_create(0, &e) != 0) {
    fprintf(stderr, "pt_engine_create: %s\n", pt_last_error());
    return 1;
  }
  const size_t n_in = (size_t)B * H * W * Cin, n_w = (size_t)N * Cin * ks * ks, n_out = (size_t)B * Ho * Wo * N;
  std::vector<uint16_t> h_in(n_in), h_w(n_w);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };      // uniform [-1, 1): full-range operands (DVFS)
  for (auto& v : h_in) v = bf16_of(rnd());
  const float wsc = 1.0f / sqrtf((float)Cin * ks * ks);
  for (auto& v : h_w) v = bf16_of(rnd() * wsc);
  uint16_t *d_in, *d_w, *d_out, *d_res = nullptr;
  float* d_b;
  hipMalloc(&d_in, n_in * 2); hipMalloc(&d_w, n_w * 2); hipMalloc(&d_out, n_out * 2); hipMalloc(&d_b, N * 4);
  hipMemcpy(d_in, h_in.data(), n_in * 2, hipMemcpyHostToDevice);
  hipMemcpy(d_w, h_w.data(), n_w * 2, hipMemcpyHostToDevice);
  hipMemset(d_b, 0, N * 4);
  if (with_res) { hipMalloc(&d_res, n_out * 2); hipMemcpy(d_res, h_in.data(), (n_out < n_in ? n_out : n_in) * 2, hipMemcpyHostToDevice); }
  hipStream_t st;
  hipStreamCreate(&st);
  auto run = [&]() {
    return pt_op_conv2d(e, d_in, B, H, W, Cin, d_w, d_b, N, ks, stride, d_out, N, 0, 1, 0, d_res, with_res ? 1 : 0, 1, 0, 0, st);
  };
  for (int i = 0; i < 3; ++i)
    if (run() != 0) { fprintf(stderr, "pt_op_conv2d: %s\n", pt_last_error()); return 1; }
  hipStreamSynchronize(st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) run();
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  const double fl = 2.0 * B * Ho * Wo * (double)N * Cin * ks * ks;
  printf("conv %dx%d s%d %d->%d @%dx%d B=%d%s: %.1f us  %.0f TFLOP/s  (PT_CONV_PIPE=%s)\n", ks, ks, stride, Cin, N, H, W, B, with_res ? " +res" : "",
         ms * 1e3, fl / ms / 1e9, getenv("PT_CONV_PIPE") ? getenv("PT_CONV_PIPE") : "default");
  pt_engine_destroy(e);
  return 0;
}
