// Probe: are dword-aligned (not 16-byte aligned) 16-byte buffer loads / stores and global loads legal and correct on gfx950?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/unaligned_b128.hip -o /tmp/unaligned && /tmp/unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const float* src, float* dst, float* dst2, int shift, int n4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), (short)0, (n4 * 4 + 8) * 4, 0x00020000);
  __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, (short)0, (n4 * 4 + 8) * 4, 0x00020000);
  const int off = (4 * i + shift) * 4;                                  // bytes: 4-byte aligned only when shift % 4 != 0
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rd, off, 0, 0);
  f32x4 w = *reinterpret_cast<const f32x4*>(src + 4 * i + shift);      // plain global_load_dwordx4 at the same address
  *reinterpret_cast<f32x4*>(dst2 + 4 * i + shift) = w;
}

int main() {
  const int n4 = 1 << 20, n = n4 * 4 + 8;
  std::vector<float> h(n), o(n), o2(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *s, *d, *d2;
  hipMalloc(&s, n * 4); hipMalloc(&d, n * 4); hipMalloc(&d2, n * 4);
  hipMemcpy(s, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; ++shift) {
    hipMemset(d, 0, n * 4); hipMemset(d2, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(n4 / 256), dim3(256), 0, 0, s, d, d2, shift, n4);
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k, dim3(n4 / 256), dim3(256), 0, 0, s, d, d2, shift, n4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipDeviceSynchronize();
    hipMemcpy(o.data(), d, n * 4, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), d2, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad2 = 0;
    for (int i = shift; i < n4 * 4 + shift; ++i) { bad += o[i] != h[i]; bad2 += o2[i] != h[i]; }
    printf("shift %d: err %d  buffer mismatches %ld  global mismatches %ld  %.1f us/launch (%.2f TB/s of 3 x %d MB)\n", shift, (int)err, bad, bad2,
           ms / 20 * 1e3, 3.0 * n4 * 16 / (ms / 20 * 1e-3) / 1e12, n4 * 16 >> 20);
  }
  return 0;
}
