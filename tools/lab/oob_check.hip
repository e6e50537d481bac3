// How does a raw buffer load of 8 / 16 bytes behave when only its FIRST dwords lie inside num_records, and at 4-byte
// alignment?  (gfx950; run on the GPU box: hipcc --offload-arch=gfx950 tools/lab/oob_check.hip -o /tmp/oob && /tmp/oob)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* p, int nbytes, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), (short)0, nbytes, 0x00020000);
  const int l = threadIdx.x;                       // lane l loads 16 bytes starting at dword (n - 4 + l), l = 0..7
  const int n = nbytes / 4;
  f4 a = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (n - 4 + l) * 4, 0, 0));
  f2 b = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, (n - 2 + l) * 4, 0, 0));
  for (int i = 0; i < 4; ++i) out[l * 8 + i] = a[i];
  out[l * 8 + 4] = b[0]; out[l * 8 + 5] = b[1];
  // negative offset (wraps to a huge unsigned): everything must be zero
  f4 c = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (l - 2) * 4, 0, 0));
  out[l * 8 + 6] = c[0]; out[l * 8 + 7] = c[3];
}
int main() {
  const int n = 64;
  float h[n], *d, *o, ho[64];
  for (int i = 0; i < n; ++i) h[i] = 100.f + i;
  hipMalloc(&d, n * 4); hipMalloc(&o, 64 * 4);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, (n - 1) * 4, o);      // num_records = 63 dwords: dword 62 is the last valid
  hipMemcpy(ho, o, 64 * 4, hipMemcpyDeviceToHost);
  for (int l = 0; l < 8; ++l)
    printf("lane %d: b128 @dword %d -> %g %g %g %g | b64 @dword %d -> %g %g | neg/low b128 -> %g %g\n", l, 63 - 4 + l, ho[l * 8], ho[l * 8 + 1],
           ho[l * 8 + 2], ho[l * 8 + 3], 63 - 2 + l, ho[l * 8 + 4], ho[l * 8 + 5], ho[l * 8 + 6], ho[l * 8 + 7]);
  return 0;
}
