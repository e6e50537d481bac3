// K-A: gather-aggregate over joints (reference: pyskl/models/gcns/utils/gcn.py:2341-2354,
//      einsum('nkctv,nkcvw->nkctw'); ST-GCN form gcn.py:88; CTR-GCN form gcn.py:658).
//
//   Y[b,t,w] = sum_u P[b,t,u] * Ahat[b,u,w],   P = relu?(Zp*scale[c]+shift[c])   b = (n, c) "unit"
//
// One wave64 owns one unit (a T x V plane of P/Y plus its V x V adjacency):
//   * the plane is streamed HBM -> LDS with 16 B/lane coalesced loads, the deferred BatchNorm affine
//     and ReLU applied in flight (P never exists in HBM);
//   * lane = frame t; its P row is read from LDS with stride V (V=25/17 is odd -> conflict-free);
//   * Ahat[b] is wave-uniform, so it is fetched through the SCALAR cache (s_load) and fed to the
//     FMAs as SGPR operands: V*V v_fma per lane, no LDS/VGPR traffic for the adjacency at all;
//   * the Y row goes back through LDS and leaves as coalesced 16 B/lane stores.
// Algorithmic HBM bytes per unit: 4*(2*T*V + V*V); nothing is read twice.
//
// Backward (one wave per unit as well):
//   dP[t,u]   = sum_w dY[t,w] * Ahat[u,w]          (same SGPR-broadcast FMA form)
//   dAhat[u,w]= sum_t P[t,u] * dY[t,w]             (V x V x T: f32 MFMA 32x32x2, k = frames)
//   dZp = dP * 1[P>0] * scale ; per-unit partial sums of dP*1[P>0] and dP*1[P>0]*Zp feed the
//   deferred-BN backward (d shift, d scale).
#include "common.h"

namespace {

template <int V>
__device__ __forceinline__ void load_plane_to_lds(const float* __restrict__ src, float* lds, int cnt, int lane,
                                                  bool vec) {
  if (vec) {
    const f32x4* __restrict__ s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* l4 = reinterpret_cast<f32x4*>(lds);
    const int c4 = cnt >> 2;
    for (int i = lane; i < c4; i += DSGCN_WAVE) l4[i] = s4[i];
  } else {
    for (int i = lane; i < cnt; i += DSGCN_WAVE) lds[i] = src[i];
  }
}

template <int V, int UNR>
__global__ __launch_bounds__(64) void k_aggregate_fwd(const float* __restrict__ zp, const float* __restrict__ scale,
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

template <int V, int UNR>
__global__ __launch_bounds__(64) void k_aggregate_bwd(const float* __restrict__ zp, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu,
                                                      const float* __restrict__ ahat, long ahat_unit_stride,
                                                      int ahat_mod, const float* __restrict__ dy,
                                                      float* __restrict__ dzp, float* __restrict__ dahat,
                                                      float* __restrict__ partial, int KC, int T, int vec) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsZ = lds;
  float* ldsG = lds + 64 * V;
  const int lane = threadIdx.x;
  const int unit = blockIdx.x;
  const int c = unit % KC;
  const float s = scale ? scale[c] : 1.f;
  const float h = shift ? shift[c] : 0.f;
  const long aidx = ahat_mod > 0 ? (long)(unit % ahat_mod) : (long)unit;
  const float* __restrict__ A = ahat + aidx * ahat_unit_stride;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float sum_h = 0.f, sum_s = 0.f;
  const int mi = lane & 31;           // MFMA row/col index inside the 32x32 tile (= joint)
  const int mk = lane >> 5;           // which of the 2 k-slices (= frame parity)
  const int mic = mi < V ? mi : V - 1;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int rows = min(64, T - t0);
    const int cnt = rows * V;
    const size_t off = ((size_t)unit * T + t0) * V;
    load_plane_to_lds<V>(zp + off, ldsZ, cnt, lane, vec);
    load_plane_to_lds<V>(dy + off, ldsG, cnt, lane, vec);
    __syncthreads();
    // dAhat += P^T dY   (A operand: P[t][u] at lane (u, k); B operand: dY[t][w] at lane (w, k))
    for (int j = 0; j < rows; j += 2) {
      const int t = j + mk;
      const bool ok = (mi < V) && (t < rows);
      const int idx = (t < rows ? t : rows - 1) * V + mic;
      float a = affine_act(ldsZ[idx], s, h, relu);
      float b = ldsG[idx];
      a = ok ? a : 0.f;
      b = ok ? b : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
    // dP row per lane, then through the ReLU mask and the BN scale, in place over ldsZ
    if (lane < rows) {
      float g[V];
#pragma unroll
      for (int w = 0; w < V; ++w) g[w] = ldsG[lane * V + w];
      // keep the scalar loads of Ahat inside this chunk iteration (no hoisting into 625 live SGPRs)
      int opaque0 = 0;
      asm volatile("" : "+s"(opaque0));
      const float* __restrict__ Aq = A + opaque0;
#pragma unroll UNR
      for (int u = 0; u < V; ++u) {
        float dp = 0.f;
#pragma unroll
        for (int w = 0; w < V; ++w) dp = fmaf(g[w], Aq[u * V + w], dp);
        const float z = ldsZ[lane * V + u];
        const float pre = fmaf(z, s, h);
        const float dpre = (!relu || pre > 0.f) ? dp : 0.f;
        sum_h += dpre;
        sum_s = fmaf(dpre, z, sum_s);
        ldsZ[lane * V + u] = dpre * s;
      }
    }
    __syncthreads();
    if (vec) {
      f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(dzp + off);
      const f32x4* l4 = reinterpret_cast<const f32x4*>(ldsZ);
      const int c4 = cnt >> 2;
      for (int i = lane; i < c4; i += 64) d4[i] = l4[i];
    } else {
      for (int i = lane; i < cnt; i += 64) dzp[off + i] = ldsZ[i];
    }
    __syncthreads();
  }
  sum_s = wave_sum(sum_s);
  sum_h = wave_sum(sum_h);
  if (lane == 0) {
    partial[(size_t)unit * 2 + 0] = sum_s;
    partial[(size_t)unit * 2 + 1] = sum_h;
  }
  // D[i=u][j=w]: j = lane&31, i = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* __restrict__ dA = dahat + (size_t)unit * V * V;
  if (mi < V) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int u = (r & 3) + 8 * (r >> 2) + 4 * mk;
      if (u < V) dA[u * V + mi] = acc[r];
    }
  }
}

template <int V>
int launch_fwd(const float* zp, const float* scale, const float* shift, int relu, const float* ahat,
               long a_stride, int a_mod, float* y, long units, int KC, int T, hipStream_t st) {
  const int vec = ((T * V) % 4 == 0) ? 1 : 0;
  dim3 grid((unsigned)units, (unsigned)((T + 63) / 64));
  const size_t lds = (size_t)64 * V * sizeof(float);
  hipLaunchKernelGGL((k_aggregate_fwd<V, 5>), grid, dim3(64), lds, st, zp, scale, shift, relu, ahat, a_stride, a_mod,
                     y, KC, T, vec);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

template <int V>
int launch_bwd(const float* zp, const float* scale, const float* shift, int relu, const float* ahat,
               long a_stride, int a_mod, const float* dy, float* dzp, float* dahat, float* partial, long units,
               int KC, int T, hipStream_t st) {
  const int vec = ((T * V) % 4 == 0) ? 1 : 0;
  const size_t lds = (size_t)2 * 64 * V * sizeof(float);
  hipLaunchKernelGGL((k_aggregate_bwd<V, 5>), dim3((unsigned)units), dim3(64), lds, st, zp, scale, shift, relu, ahat,
                     a_stride, a_mod, dy, dzp, dahat, partial, KC, T, vec);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" {

// See include/dsgcn.h for the contract.
int dsgcn_aggregate_fwd(const float* zp, const float* scale, const float* shift, int relu, const float* ahat,
                        float* y, int n, int KC, int T, int V, void* stream) {
  if (!zp || !ahat || !y || n <= 0 || KC <= 0 || T <= 0) return DSGCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long units = (long)n * KC;
  switch (V) {
    case 25: return launch_fwd<25>(zp, scale, shift, relu, ahat, 25 * 25, 0, y, units, KC, T, st);
    case 17: return launch_fwd<17>(zp, scale, shift, relu, ahat, 17 * 17, 0, y, units, KC, T, st);
    case 18: return launch_fwd<18>(zp, scale, shift, relu, ahat, 18 * 18, 0, y, units, KC, T, st);
    default: return DSGCN_EUNSUPPORTED;
  }
}

int dsgcn_aggregate_bwd(const float* zp, const float* scale, const float* shift, int relu, const float* ahat,
                        const float* dy, float* dzp, float* dahat, float* partial, int n, int KC, int T, int V,
                        void* stream) {
  if (!zp || !ahat || !dy || !dzp || !dahat || !partial || n <= 0 || KC <= 0 || T <= 0) return DSGCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long units = (long)n * KC;
  switch (V) {
    case 25: return launch_bwd<25>(zp, scale, shift, relu, ahat, 625, 0, dy, dzp, dahat, partial, units, KC, T, st);
    case 17: return launch_bwd<17>(zp, scale, shift, relu, ahat, 289, 0, dy, dzp, dahat, partial, units, KC, T, st);
    case 18: return launch_bwd<18>(zp, scale, shift, relu, ahat, 324, 0, dy, dzp, dahat, partial, units, KC, T, st);
    default: return DSGCN_EUNSUPPORTED;
  }
}

}  // extern "C"
