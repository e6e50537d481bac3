// Measurement build only (libdsgcn_lab.so, -DDSGCN_LAB): the first formulation of K-A's forward — lane-per-frame FMAs with
// the adjacency read through the scalar cache — kept for A/B runs against the MFMA kernels (tools/ka_variants.py).
// Included by csrc/aggregate.hip inside its anonymous namespace; not part of the product library.
#pragma once

template <int V, int UNR>
__global__ __launch_bounds__(64) void k_aggregate_fwd_valu(const float* __restrict__ zp, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu,
                                                      const float* __restrict__ ahat, long ahat_unit_stride,
                                                      int ahat_mod, float* __restrict__ y, int KC, int T, int vec) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int unit = blockIdx.x;
  const int t0 = blockIdx.y * 64;
  const int c = unit % KC;
  const float s = scale ? scale[c] : 1.f;
  const float h = shift ? shift[c] : 0.f;
  // ahat_mod > 0: adjacency shared across samples (index = unit % ahat_mod), else per unit
  const long aidx = ahat_mod > 0 ? (long)(unit % ahat_mod) : (long)unit;
  const float* __restrict__ A = ahat + aidx * ahat_unit_stride;
  const float* __restrict__ src = zp + ((size_t)unit * T + t0) * V;
  float* __restrict__ dst = y + ((size_t)unit * T + t0) * V;
  const int rows = min(64, T - t0);
  const int cnt = rows * V;
  if (vec) {
    const f32x4* __restrict__ s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* l4 = reinterpret_cast<f32x4*>(lds);
    const int c4 = cnt >> 2;
    for (int i = lane; i < c4; i += 64) {
      f32x4 v = s4[i];
      v.x = affine_act(v.x, s, h, relu);
      v.y = affine_act(v.y, s, h, relu);
      v.z = affine_act(v.z, s, h, relu);
      v.w = affine_act(v.w, s, h, relu);
      l4[i] = v;
    }
  } else {
    for (int i = lane; i < cnt; i += 64) lds[i] = affine_act(src[i], s, h, relu);
  }
  __syncthreads();
  const int r = lane < rows ? lane : 0;
  float acc[V];
#pragma unroll
  for (int w = 0; w < V; ++w) acc[w] = 0.f;
#pragma unroll UNR
  for (int u = 0; u < V; ++u) {
    const float p = lds[r * V + u];
#pragma unroll
    for (int w = 0; w < V; ++w) acc[w] = fmaf(p, A[u * V + w], acc[w]);
  }
  __syncthreads();
  if (lane < rows) {
#pragma unroll
    for (int w = 0; w < V; ++w) lds[lane * V + w] = acc[w];
  }
  __syncthreads();
  if (vec) {
    f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(dst);
    const f32x4* l4 = reinterpret_cast<const f32x4*>(lds);
    const int c4 = cnt >> 2;
    for (int i = lane; i < c4; i += 64) d4[i] = l4[i];
  } else {
    for (int i = lane; i < cnt; i += 64) dst[i] = lds[i];
  }
}
