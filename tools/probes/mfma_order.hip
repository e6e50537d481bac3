// Which fp32 sums do the fp32 MFMA shapes compute?  One wave, C = A (32 x K) * B (K x 32) on random fp32 operands of mixed
// magnitude, computed by v_mfma_f32_32x32x2_f32 over k pairs, by v_mfma_f32_16x16x4_f32 over k quads (four quadrants), and
// by scalar chains on the vector ALU: sequential fmaf over k, "pairs first" (a0*b0 + a1*b1, then added) and "quads first".
// Prints how many of the 1024 results agree bit for bit.  Result on gfx950 (profiles/r06/mfma_order.txt): both MFMA shapes
// equal the sequential fmaf chain on every element.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o mfma_order tools/probes/mfma_order.hip && ./mfma_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_probe(const float* A, const float* B, int K, float* C32, float* C16, float* Cseq, float* Cpair,
                        float* Cquad) {
  const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
  f32x16 acc = {};
  for (int k = 0; k < K; k += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + half], B[(k + half) * 32 + l31], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C32[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
  const int l15 = lane & 15, lk = lane >> 4;
  for (int qi = 0; qi < 2; ++qi)
    for (int qj = 0; qj < 2; ++qj) {
      f32x4 a4 = {};
      for (int k = 0; k < K; k += 4)
        a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(16 * qi + l15) * K + k + lk], B[(k + lk) * 32 + 16 * qj + l15], a4, 0, 0, 0);
      for (int r = 0; r < 4; ++r) C16[(16 * qi + 4 * lk + r) * 32 + 16 * qj + l15] = a4[r];
    }
  for (int e = lane; e < 1024; e += 64) {
    const int i = e >> 5, j = e & 31;
    float s = 0.f, p = 0.f, q = 0.f;
    for (int k = 0; k < K; ++k) s = __builtin_fmaf(A[i * K + k], B[k * 32 + j], s);
    for (int k = 0; k < K; k += 2) {
      const float t = __builtin_fmaf(A[i * K + k + 1], B[(k + 1) * 32 + j], A[i * K + k] * B[k * 32 + j]);
      p += t;
    }
    for (int k = 0; k < K; k += 4) {
      float t = A[i * K + k] * B[k * 32 + j];
      for (int d = 1; d < 4; ++d) t = __builtin_fmaf(A[i * K + k + d], B[(k + d) * 32 + j], t);
      q += t;
    }
    Cseq[e] = s; Cpair[e] = p; Cquad[e] = q;
  }
}

static int same(const std::vector<float>& a, const std::vector<float>& b) {
  int n = 0;
  for (size_t i = 0; i < a.size(); ++i) n += std::memcmp(&a[i], &b[i], 4) == 0;
  return n;
}

int main() {
  const int Ks[3] = {4, 64, 400};
  for (int K : Ks)
    for (int mode = 0; mode < 2; ++mode) {
      std::vector<float> A(32 * K), B(K * 32);
      srand(17 + K + mode);
      auto rnd = [&]() {
        float v = (float)rand() / RAND_MAX * 2.f - 1.f;
        if (mode) v *= std::ldexp(1.f, rand() % 24 - 12);          // mixed binades: cancellation exposes the order
        return v;
      };
      for (auto& v : A) v = rnd();
      for (auto& v : B) v = rnd();
      float *dA, *dB, *d[5];
      if (hipMalloc(&dA, A.size() * 4) != hipSuccess || hipMalloc(&dB, B.size() * 4) != hipSuccess) return 1;
      for (auto& p : d)
        if (hipMalloc(&p, 4096) != hipSuccess) return 1;
      (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
      (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, K, d[0], d[1], d[2], d[3], d[4]);
      std::vector<float> h[5];
      for (int i = 0; i < 5; ++i) {
        h[i].resize(1024);
        (void)hipMemcpy(h[i].data(), d[i], 4096, hipMemcpyDeviceToHost);
      }
      printf("K=%3d %s: 32x32x2 vs 16x16x4 %4d | 32x32x2 vs seq-fma %4d, vs pairs %4d, vs quads %4d | 16x16x4 vs seq-fma %4d, "
             "vs pairs %4d, vs quads %4d  (of 1024)\n",
             K, mode ? "mixed" : "unit ", same(h[0], h[1]), same(h[0], h[2]), same(h[0], h[3]), same(h[0], h[4]),
             same(h[1], h[2]), same(h[1], h[3]), same(h[1], h[4]));
      (void)hipFree(dA); (void)hipFree(dB);
      for (auto& p : d) (void)hipFree(p);
    }
  return 0;
}
