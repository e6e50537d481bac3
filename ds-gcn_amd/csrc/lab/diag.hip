// Diagnostics (tools/ only): raw f32 MFMA issue-rate probe used to calibrate the roofline of K-C.
#include "../common.h"
#ifdef DSGCN_LAB
namespace {
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_probe(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace
extern "C" int dsgcn_diag_mfma_probe(float* out, int blocks, int iters, int nacc, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (nacc == 1) hipLaunchKernelGGL(k_mfma_probe<1>, dim3(blocks), dim3(256), 0, st, out, iters, 1.f, 2.f);
  else if (nacc == 2) hipLaunchKernelGGL(k_mfma_probe<2>, dim3(blocks), dim3(256), 0, st, out, iters, 1.f, 2.f);
  else hipLaunchKernelGGL(k_mfma_probe<4>, dim3(blocks), dim3(256), 0, st, out, iters, 1.f, 2.f);
  DSGCN_LAUNCH_CHECK();
  return 0;
}
#endif
