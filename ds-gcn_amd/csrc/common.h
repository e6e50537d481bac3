// Shared device helpers for the DS-GCN HIP kernels (gfx950 / CDNA4 only: wave64, MFMA f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DSGCN_WAVE 64

// Error codes returned by every C-ABI entry point: 0 = ok, >0 = hipError_t of the launch,
// <0 = argument rejected before any launch (see include/dsgcn.h).
#define DSGCN_EINVAL (-1)
#define DSGCN_EUNSUPPORTED (-2)

#define DSGCN_LAUNCH_CHECK()                      \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// relu?(x*s+h)
__device__ __forceinline__ float affine_act(float x, float s, float h, int relu) {
  float v = fmaf(x, s, h);
  return relu ? fmaxf(v, 0.f) : v;
}

// Single-wave workgroups: LDS traffic of one wave is executed in issue order, so cross-lane exchange through
// LDS needs no s_barrier — only "all my DS ops are done" and a compiler fence.  Unlike __syncthreads() this does
// NOT emit s_waitcnt vmcnt(0), so global prefetch loads stay in flight across it.
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// t = e / V, v = e % V for 0 <= e < 2^22 with invV = 1.f / V (float reciprocal + one fix-up step: exact)
__device__ __forceinline__ void divmod_small(int e, int V, float invV, int& t, int& v) {
  t = (int)((float)e * invV);
  v = e - t * V;
  if (v < 0) { v += V; --t; }
  else if (v >= V) { v -= V; ++t; }
}

// One wave copies a 16-B aligned run of L4 float4 from global memory into LDS; the loads of a 512-float4 chunk are all
// issued before the first LDS write (whole (T,V) planes are <= a few KB: the wave keeps the entire plane in flight).
__device__ __forceinline__ void plane_to_lds(const float* __restrict__ src, float* __restrict__ dst, int L4, int lane) {
  const f32x4* __restrict__ s4 = reinterpret_cast<const f32x4*>(src);
  f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(dst);
  for (int base = 0; base < L4; base += 512) {
    f32x4 r[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = base + q * 64 + lane;
      if (i < L4) r[q] = s4[i];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = base + q * 64 + lane;
      if (i < L4) d4[i] = r[q];
    }
  }
}

// ---- three-term bf16 form of the fp32 product (B3) -----------------------------------------------------------------
// An fp32 value is the exact sum of three bf16 terms (8 significand bits each, round-to-nearest residues: x0 = bf16(x),
// x1 = bf16(x - x0), x2 = x - x0 - x1; both subtractions are exact in fp32 and the last residue has at most 6 bits).
// a*b = sum_{i,j} a_i*b_j; the six terms with i + j <= 2 are kept (each exact in fp32, accumulated in the MFMA's fp32
// accumulator), the dropped ones are below 2^-24 |a||b| — the rounding class of the fp32 MFMA it replaces, at 2.7x its
// rate: six v_mfma_f32_32x32x16_bf16 (32 cycles each) per 16 channels against eight v_mfma_f32_32x32x2_f32 (64 cycles).
// Non-finite inputs give NaN where fp32 would keep an infinity (inf - inf in the residue).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned b3_pack(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, bf16x2));   // v_cvt_pk_bf16_f32: lo = a, hi = b
}

// packed bf16 pairs p[s] = (lo: term s of a, hi: term s of b)
__device__ __forceinline__ void b3_split(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = b3_pack(a, b);
  a -= __builtin_bit_cast(float, p0 << 16);
  b -= __builtin_bit_cast(float, p0 & 0xffff0000u);
  p1 = b3_pack(a, b);
  a -= __builtin_bit_cast(float, p1 << 16);
  b -= __builtin_bit_cast(float, p1 & 0xffff0000u);
  p2 = b3_pack(a, b);
}
