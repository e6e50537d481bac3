// K-A: gather-aggregate over joints (reference: pyskl/models/gcns/utils/gcn.py:2341-2354,
//      einsum('nkctv,nkcvw->nkctw'); CTR-GCN form gcn.py:658).
//
//   Y[b,t,w] = sum_u P[b,t,u] * Ahat[b,u,w],   P = relu?(Zp*scale[c]+shift[c])   b = (n, c) "unit"
//
// One wave64 owns one unit (a T x V plane of P/Y plus its V x V adjacency; 64 frames per pass):
//   * the plane is streamed HBM -> LDS with 16 B/lane coalesced loads, the deferred BatchNorm affine
//     and ReLU applied in flight (P never exists in HBM); Ahat[b] (V*V floats) is staged next to it;
//   * the (frames x V) . (V x V) product runs on the f32 matrix core: v_mfma_f32_32x32x2_f32 with
//     i = frame, j = joint w, k = joint u (V=25 -> 13 k-steps, padded lanes fed zeros).  Operands are
//     single ds_read_b32 per lane: P rows at stride V (odd -> conflict-free), Ahat rows contiguous;
//   * the Y tile leaves the accumulators through LDS as coalesced 16 B/lane stores.
// Algorithmic HBM bytes per unit: 4*(2*T*V + V*V); nothing is read twice.
// (A first version kept lane = frame and fed Ahat through the scalar cache as SGPR FMA operands;
//  it measured 24 % of the HBM roofline — k_aggregate_fwd_valu below is kept for that A/B only.)
//
// Backward (one wave per unit as well), three MFMA products out of the same LDS tiles:
//   dAhat[u,w] = sum_t P[t,u] * dY[t,w]            (i=u, j=w, k=frames)
//   dP[t,u]    = sum_w dY[t,w] * Ahat[u,w]         (i=frame, j=u, k=w)
//   dZp = dP * 1[P>0] * scale ; per-unit partial sums of dP*1[P>0] and dP*1[P>0]*Zp feed the
//   deferred-BN backward (d shift, d scale).
#include "common.h"

namespace {

// The persistent kernels' operand prefetch (planes and adjacency are read exactly once per launch).  KA_NT_LOADS: issue
// them as non-temporal loads (streamed data need not displace what the next launch will find in L2 / Infinity Cache).
#ifndef KA_NT_LOADS
#define KA_NT_LOADS 0
#endif
template <typename TT>
__device__ __forceinline__ TT ka_ld(const TT* p) {
#if KA_NT_LOADS
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

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

#ifdef DSGCN_LAB
#include "lab/aggregate_valu.h"   // superseded VALU formulation, A/B measurements only
#endif

// MFMA C/D row of accumulator register r for this lane (32x32 tile): (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int V>
__global__ __launch_bounds__(64) void k_aggregate_fwd(const float* __restrict__ zp, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu,
                                                      const float* __restrict__ ahat, float* __restrict__ y, int KC,
                                                      int T, int vec) {
  constexpr int KS = (V + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsP = lds;
  float* ldsA = lds + 64 * V;
  const int lane = threadIdx.x;
  const int unit = blockIdx.x;
  const int t0 = blockIdx.y * 64;
  const int c = unit % KC;
  const float s = scale ? scale[c] : 1.f;
  const float h = shift ? shift[c] : 0.f;
  const float* __restrict__ A = ahat + (size_t)unit * V * V;
  const float* __restrict__ src = zp + ((size_t)unit * T + t0) * V;
  float* __restrict__ dst = y + ((size_t)unit * T + t0) * V;
  const int rows = min(64, T - t0);
  const int cnt = rows * V;
  for (int i = lane; i < V * V; i += 64) ldsA[i] = A[i];
  if (vec) {
    const f32x4* __restrict__ s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* l4 = reinterpret_cast<f32x4*>(ldsP);
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
    for (int i = lane; i < cnt; i += 64) ldsP[i] = affine_act(src[i], s, h, relu);
  }
  __syncthreads();
  const int mi = lane & 31, mk = lane >> 5;
  const int mic = mi < V ? mi : V - 1;
  float b[KS];
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    const int u = 2 * q + mk;
    const float v = ldsA[(u < V ? u : V - 1) * V + mic];
    b[q] = (u < V && mi < V) ? v : 0.f;
  }
  f32x16 acc[2];
#pragma unroll
  for (int tile = 0; tile < 2; ++tile) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[tile][i] = 0.f;
    if (tile * 32 < rows) {
      const int t = tile * 32 + mi;
      const int tc = t < rows ? t : rows - 1;
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        const int u = 2 * q + mk;
        const float v = ldsP[tc * V + (u < V ? u : V - 1)];
        const float a = (u < V && t < rows) ? v : 0.f;
        acc[tile] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[q], acc[tile], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  if (mi < V) {
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = tile * 32 + mfma_row(r, mk);
        if (t < rows) ldsP[t * V + mi] = acc[tile][r];
      }
    }
  }
  __syncthreads();
  if (vec) {
    f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(dst);
    const f32x4* l4 = reinterpret_cast<const f32x4*>(ldsP);
    const int c4 = cnt >> 2;
    for (int i = lane; i < c4; i += 64) d4[i] = l4[i];
  } else {
    for (int i = lane; i < cnt; i += 64) dst[i] = ldsP[i];
  }
}

// Persistent form of k_aggregate_fwd: each wave walks work items (unit, 64-frame chunk) with a grid
// stride and keeps the NEXT item's P plane and adjacency in flight (global -> VGPR) while the matrix core
// works on the current one, so HBM requests, MFMA and stores of neighbouring items overlap inside one wave.
// RW: the exact row count of every work item when the shape makes it one (loop bounds and tail guards fold), 0 = runtime
// KA_WPB independent waves share one workgroup (its LDS cut into per-wave slices, no workgroup barrier anywhere): the
// launch dispatches a quarter of the workgroups.
constexpr int KA_WPB = 4;

template <int V, int CH, int RW = 0>
__global__ __launch_bounds__(64 * KA_WPB) void k_aggregate_fwd_pipe(const float* __restrict__ zp,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu,
                                                           const float* __restrict__ ahat, float* __restrict__ y,
                                                           int KC, int T, int chunks, long items, int direct) {
  constexpr int KS = (V + 1) / 2;
  constexpr int NP4 = (CH * V / 4 + 63) / 64;     // float4 loads per lane for a full CH-frame chunk
  constexpr int NA = (V * V + 63) / 64;           // dword loads per lane for the adjacency
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* ldsP = lds + wv * (CH * V + V * V);
  float* ldsA = ldsP + CH * V;
  const int mi = lane & 31, mk = lane >> 5;
  const int mic = mi < V ? mi : V - 1;
  f32x4 pre[NP4];
  float prea[NA];

  auto issue = [&](long item) {
    const long unit = item / chunks;
    const int t0 = (int)(item - unit * chunks) * CH;
    const int rows = RW ? RW : min(CH, T - t0);
    const int c4 = (rows * V) >> 2;
    const f32x4* __restrict__ s4 = reinterpret_cast<const f32x4*>(zp + ((size_t)unit * T + t0) * V);
    const float* __restrict__ A = ahat + (size_t)unit * V * V;
#pragma unroll
    for (int q = 0; q < NP4; ++q) {
      const int i = lane + q * 64;
      if (i < c4) pre[q] = ka_ld(&s4[i]);
    }
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int i = lane + q * 64;
      if (i < V * V) prea[q] = ka_ld(&A[i]);
    }
  };

  long item = (long)blockIdx.x * KA_WPB + wv;
  if (item < items) issue(item);
  while (item < items) {
    const long unit = item / chunks;
    const int t0 = (int)(item - unit * chunks) * CH;
    const int rows = RW ? RW : min(CH, T - t0);
    const int c4 = (rows * V) >> 2;
    const int c = (int)(unit % KC);
    const float s = scale ? scale[c] : 1.f;
    const float h = shift ? shift[c] : 0.f;
    // staged registers -> LDS (deferred BN affine + ReLU applied here)
    {
      f32x4* l4 = reinterpret_cast<f32x4*>(ldsP);
#pragma unroll
      for (int q = 0; q < NP4; ++q) {
        const int i = lane + q * 64;
        if (i < c4) {
          f32x4 v = pre[q];
          v.x = affine_act(v.x, s, h, relu);
          v.y = affine_act(v.y, s, h, relu);
          v.z = affine_act(v.z, s, h, relu);
          v.w = affine_act(v.w, s, h, relu);
          l4[i] = v;
        }
      }
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int i = lane + q * 64;
        if (i < V * V) ldsA[i] = prea[q];
      }
    }
    const long next = item + (long)gridDim.x * KA_WPB;
    if (next < items) issue(next);
    wave_lds_sync();
    float b[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int u = 2 * q + mk;
      const float v = ldsA[(u < V ? u : V - 1) * V + mic];
      b[q] = (u < V && mi < V) ? v : 0.f;
    }
    constexpr int NTILE = CH / 32;
    f32x16 acc[NTILE];
#pragma unroll
    for (int tile = 0; tile < NTILE; ++tile) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[tile][i] = 0.f;
      if (tile * 32 < rows) {
        const int t = tile * 32 + mi;
        const int tc = t < rows ? t : rows - 1;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          const int u = 2 * q + mk;
          const float v = ldsP[tc * V + (u < V ? u : V - 1)];
          const float a = (u < V && t < rows) ? v : 0.f;
          acc[tile] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[q], acc[tile], 0, 0, 0);
        }
      }
    }
    if (direct) {
      // accumulator layout -> HBM directly: per instruction two 4*V-byte row segments (rows t and t+4); a unit's rows are
      // contiguous, so every 128-B line is completed by the same wave within a few instructions
      float* __restrict__ yo = y + ((size_t)unit * T + t0) * V;
      if (mi < V) {
#pragma unroll
        for (int tile = 0; tile < NTILE; ++tile) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = tile * 32 + mfma_row(r, mk);
            if (t < rows) yo[t * V + mi] = acc[tile][r];
          }
        }
      }
      wave_lds_sync();
    } else {
      wave_lds_sync();
      if (mi < V) {
#pragma unroll
        for (int tile = 0; tile < NTILE; ++tile) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = tile * 32 + mfma_row(r, mk);
            if (t < rows) ldsP[t * V + mi] = acc[tile][r];
          }
        }
      }
      wave_lds_sync();
      {
        f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(y + ((size_t)unit * T + t0) * V);
        const f32x4* l4 = reinterpret_cast<const f32x4*>(ldsP);
#pragma unroll
        for (int q = 0; q < NP4; ++q) {
          const int i = lane + q * 64;
          if (i < c4) __builtin_nontemporal_store(l4[i], &d4[i]);
        }
      }
      wave_lds_sync();
    }
    item = next;
  }
}

template <int V>
__global__ __launch_bounds__(64) void k_aggregate_bwd(const float* __restrict__ zp, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu,
                                                      const float* __restrict__ ahat, const float* __restrict__ dy,
                                                      float* __restrict__ dzp, float* __restrict__ dahat,
                                                      float* __restrict__ partial, int KC, int T, int vec) {
  constexpr int KS = (V + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsZ = lds;
  float* ldsG = lds + 64 * V;
  float* ldsA = lds + 128 * V;
  const int lane = threadIdx.x;
  const int unit = blockIdx.x;
  const int c = unit % KC;
  const float s = scale ? scale[c] : 1.f;
  const float h = shift ? shift[c] : 0.f;
  const float* __restrict__ A = ahat + (size_t)unit * V * V;
  const int mi = lane & 31;           // MFMA row (A operand) / column (B operand, C/D) index inside the tile
  const int mk = lane >> 5;           // which of the 2 k-slices
  const int mic = mi < V ? mi : V - 1;
  for (int i = lane; i < V * V; i += 64) ldsA[i] = A[i];
  f32x16 accA;
#pragma unroll
  for (int i = 0; i < 16; ++i) accA[i] = 0.f;
  float sum_h = 0.f, sum_s = 0.f;
  float bt[KS];                       // B operand of dP = dY . Ahat^T : B[k=w][j=u] = Ahat[u][w]
  bool have_bt = false;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int rows = min(64, T - t0);
    const int cnt = rows * V;
    const size_t off = ((size_t)unit * T + t0) * V;
    load_plane_to_lds<V>(zp + off, ldsZ, cnt, lane, vec);
    load_plane_to_lds<V>(dy + off, ldsG, cnt, lane, vec);
    __syncthreads();
    if (!have_bt) {
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        const int w = 2 * q + mk;
        const float v = ldsA[mic * V + (w < V ? w : V - 1)];
        bt[q] = (w < V && mi < V) ? v : 0.f;
      }
      have_bt = true;
    }
    // dAhat += P^T dY   (A operand: P[t][u] at lane (u, k); B operand: dY[t][w] at lane (w, k))
    for (int j = 0; j < rows; j += 2) {
      const int t = j + mk;
      const bool ok = (mi < V) && (t < rows);
      const int idx = (t < rows ? t : rows - 1) * V + mic;
      float a = affine_act(ldsZ[idx], s, h, relu);
      float b = ldsG[idx];
      a = ok ? a : 0.f;
      b = ok ? b : 0.f;
      accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accA, 0, 0, 0);
    }
    // dP tiles (i = frame, j = u, k = w), then ReLU mask / BN scale in place over ldsZ
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
      if (tile * 32 < rows) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const int t = tile * 32 + mi;
        const int tc = t < rows ? t : rows - 1;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          const int w = 2 * q + mk;
          const float v = ldsG[tc * V + (w < V ? w : V - 1)];
          const float a = (w < V && t < rows) ? v : 0.f;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bt[q], acc, 0, 0, 0);
        }
        if (tile == 0) __syncthreads();   // every lane's P reads of the dAhat product are done before ldsZ is rewritten
        if (mi < V) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int tt = tile * 32 + mfma_row(r, mk);
            if (tt < rows) {
              const float z = ldsZ[tt * V + mi];
              const float pre = fmaf(z, s, h);
              const float dpre = (!relu || pre > 0.f) ? acc[r] : 0.f;
              sum_h += dpre;
              sum_s = fmaf(dpre, z, sum_s);
              ldsZ[tt * V + mi] = dpre * s;
            }
          }
        }
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
  float* __restrict__ dA = dahat + (size_t)unit * V * V;
  if (mi < V) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int u = mfma_row(r, mk);
      if (u < V) dA[u * V + mi] = accA[r];
    }
  }
}

// Persistent/pipelined backward for T <= TM (TM = 64, or 128 for the long clips of the K400 config: T = 100 / 50): same
// products as k_aggregate_bwd, with the next unit's Zp / dY planes and adjacency in flight while the current unit is on
// the matrix core.
// TE: the exact frame count when it is one of the model's (32 / 16: loop bounds and tail guards fold), 0 = runtime T <= TM
template <int V, bool DA_LDS, int TM, int TE = 0>
__global__ __launch_bounds__(64 * KA_WPB) void k_aggregate_bwd_pipe(const float* __restrict__ zp,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu,
                                                           const float* __restrict__ ahat,
                                                           const float* __restrict__ dy, float* __restrict__ dzp,
                                                           float* __restrict__ dahat, float* __restrict__ partial,
                                                           int KC, int T, long units) {
  constexpr int KS = (V + 1) / 2;
  constexpr int NP4 = (TM * V / 4 + 63) / 64;
  constexpr int NA = (V * V + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* ldsZ = lds + wv * (2 * (TE ? TE : T) * V + V * V);
  float* ldsG = ldsZ + (TE ? TE : T) * V;         // LDS sized by T (T*V % 4 == 0): short layers keep more waves resident
  float* ldsA = ldsZ + 2 * (TE ? TE : T) * V;
  const int mi = lane & 31, mk = lane >> 5;
  const int mic = mi < V ? mi : V - 1;
  const int rows = TE ? TE : T;       // T <= TM
  const int c4 = (rows * V) >> 2;
  f32x4 prez[NP4], preg[NP4];
  float prea[NA];

  auto issue = [&](long unit) {
    const f32x4* __restrict__ z4 = reinterpret_cast<const f32x4*>(zp + (size_t)unit * (TE ? TE : T) * V);
    const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(dy + (size_t)unit * (TE ? TE : T) * V);
    const float* __restrict__ A = ahat + (size_t)unit * V * V;
#pragma unroll
    for (int q = 0; q < NP4; ++q) {
      const int i = lane + q * 64;
      if (i < c4) { prez[q] = ka_ld(&z4[i]); preg[q] = ka_ld(&g4[i]); }
    }
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int i = lane + q * 64;
      if (i < V * V) prea[q] = ka_ld(&A[i]);
    }
  };

  long unit = (long)blockIdx.x * KA_WPB + wv;
  if (unit < units) issue(unit);
  while (unit < units) {
    const int c = (int)(unit % KC);
    const float s = scale ? scale[c] : 1.f;
    const float h = shift ? shift[c] : 0.f;
    {
      f32x4* lz = reinterpret_cast<f32x4*>(ldsZ);
      f32x4* lg = reinterpret_cast<f32x4*>(ldsG);
#pragma unroll
      for (int q = 0; q < NP4; ++q) {
        const int i = lane + q * 64;
        if (i < c4) { lz[i] = prez[q]; lg[i] = preg[q]; }
      }
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int i = lane + q * 64;
        if (i < V * V) ldsA[i] = prea[q];
      }
    }
    const long next = unit + (long)gridDim.x * KA_WPB;
    if (next < units) issue(next);
    wave_lds_sync();
    float bt[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int w = 2 * q + mk;
      const float v = ldsA[mic * V + (w < V ? w : V - 1)];
      bt[q] = (w < V && mi < V) ? v : 0.f;
    }
    f32x16 accA, accB;            // two chains: consecutive MFMAs on one accumulator wait for each other
#pragma unroll
    for (int i = 0; i < 16; ++i) { accA[i] = 0.f; accB[i] = 0.f; }
    // dAhat = P^T dY : k runs over frames; operands of 8 k-steps are read from LDS before their MFMAs are issued
    // (a read->use chain per step left the matrix pipe idle ~2/3 of the time: round-1 ablation)
    for (int j0 = 0; j0 < rows; j0 += 16) {
      float av[8], bv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int t = j0 + 2 * q + mk;
        const bool ok = (mi < V) && (t < rows);
        const int idx = (t < rows ? t : rows - 1) * V + mic;
        const float a = affine_act(ldsZ[idx], s, h, relu);
        const float b = ldsG[idx];
        av[q] = ok ? a : 0.f;
        bv[q] = ok ? b : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (j0 + 2 * q < rows) {
          if (q & 1) accB = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], accB, 0, 0, 0);
          else accA = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], accA, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) accA[i] += accB[i];
    if (!DA_LDS) {
      float* __restrict__ dA = dahat + (size_t)unit * V * V;
      if (mi < V) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int u = mfma_row(r, mk);
          if (u < V) __builtin_nontemporal_store(accA[r], &dA[u * V + mi]);
        }
      }
    } else {
      // ldsA is free once bt has been read: bounce dAhat through it so the 2.5 KB leave as contiguous 256-B stores
      // instead of 16 x two 100-B row segments
      if (mi < V) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int u = mfma_row(r, mk);
          if (u < V) ldsA[u * V + mi] = accA[r];
        }
      }
    }
    float sum_h = 0.f, sum_s = 0.f;
#pragma unroll
    for (int tile = 0; tile < TM / 32; ++tile) {
      if (tile * 32 < rows) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const int t = tile * 32 + mi;
        const int tc = t < rows ? t : rows - 1;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          const int w = 2 * q + mk;
          const float v = ldsG[tc * V + (w < V ? w : V - 1)];
          const float a = (w < V && t < rows) ? v : 0.f;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bt[q], acc, 0, 0, 0);
        }
        wave_lds_sync();
        if (mi < V) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int tt = tile * 32 + mfma_row(r, mk);
            if (tt < rows) {
              const float z = ldsZ[tt * V + mi];
              const float pre = fmaf(z, s, h);
              const float dpre = (!relu || pre > 0.f) ? acc[r] : 0.f;
              sum_h += dpre;
              sum_s = fmaf(dpre, z, sum_s);
              ldsZ[tt * V + mi] = dpre * s;
            }
          }
        }
      }
    }
    wave_lds_sync();
    if (DA_LDS) {
      float* __restrict__ dA = dahat + (size_t)unit * V * V;
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int i = lane + q * 64;
        if (i < V * V) __builtin_nontemporal_store(ldsA[i], &dA[i]);
      }
    }
    {
      f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(dzp + (size_t)unit * (TE ? TE : T) * V);
      const f32x4* l4 = reinterpret_cast<const f32x4*>(ldsZ);
#pragma unroll
      for (int q = 0; q < NP4; ++q) {
        const int i = lane + q * 64;
        if (i < c4) __builtin_nontemporal_store(l4[i], &d4[i]);
      }
    }
    sum_s = wave_sum(sum_s);
    sum_h = wave_sum(sum_h);
    if (lane == 0) {
      partial[(size_t)unit * 2 + 0] = sum_s;
      partial[(size_t)unit * 2 + 1] = sum_h;
    }
    wave_lds_sync();
    unit = next;
  }
}

// Two waves per unit for 32 < T <= 64: each wave owns 32 frames (its half of Zp / dY / dZp and half of the adjacency
// load), so the serial load -> MFMA -> store chain per wave is half as long and twice as many waves are resident
// (round-1 ablation: at n=128 the one-wave form spends half its time in the per-wave MFMA chains, not on HBM).
// dAhat = P^T dY is summed over the two halves through LDS; two raw s_barriers per unit (no vmcnt drain: the next
// unit's prefetch stays in flight).  partial has 2 rows per unit: [wave][unit][2].
// FULL: T == NW*HR exactly (every wave has HR rows: bounds and tail guards fold)
template <int V, int HR, int NW, bool FULL = false>
__global__ __launch_bounds__(64 * NW) void k_aggregate_bwd_pair(const float* __restrict__ zp,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu,
                                                            const float* __restrict__ ahat,
                                                            const float* __restrict__ dy, float* __restrict__ dzp,
                                                            float* __restrict__ dahat, float* __restrict__ partial,
                                                            int KC, int T, long units) {
  constexpr int KS = (V + 1) / 2;
  constexpr int NP4 = (HR * V / 4 + 63) / 64;
  constexpr int AH = (V * V + NW - 1) / NW;               // adjacency dwords each wave loads
  constexpr int NAH = (AH + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* ldsZ = lds + wave * HR * V;
  float* ldsG = lds + NW * HR * V + wave * HR * V;
  float* ldsA = lds + 2 * NW * HR * V;
  float* ldsD = ldsA + V * V;                              // [NW-1][V*V] partial dAhat of waves 1..NW-1
  const int mi = lane & 31, mk = lane >> 5;
  const int mic = mi < V ? mi : V - 1;
  const int rows = FULL ? HR : min(HR, T - wave * HR);
  const int c4 = (rows * V) >> 2;
  const int a0 = wave * AH, a1 = min(V * V, a0 + AH);
  f32x4 prez[NP4], preg[NP4];
  float prea[NAH];

  auto issue = [&](long unit) {
    const size_t off = ((size_t)unit * T + wave * HR) * V;
    const f32x4* __restrict__ z4 = reinterpret_cast<const f32x4*>(zp + off);
    const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(dy + off);
    const float* __restrict__ A = ahat + (size_t)unit * V * V;
#pragma unroll
    for (int q = 0; q < NP4; ++q) {
      const int i = lane + q * 64;
      if (i < c4) { prez[q] = ka_ld(&z4[i]); preg[q] = ka_ld(&g4[i]); }
    }
#pragma unroll
    for (int q = 0; q < NAH; ++q) {
      const int i = a0 + lane + q * 64;
      if (i < a1) prea[q] = ka_ld(&A[i]);
    }
  };

  long unit = blockIdx.x;
  if (unit < units) issue(unit);
  while (unit < units) {
    const int c = (int)(unit % KC);
    const float s = scale ? scale[c] : 1.f;
    const float h = shift ? shift[c] : 0.f;
    {
      f32x4* lz = reinterpret_cast<f32x4*>(ldsZ);
      f32x4* lg = reinterpret_cast<f32x4*>(ldsG);
#pragma unroll
      for (int q = 0; q < NP4; ++q) {
        const int i = lane + q * 64;
        if (i < c4) { lz[i] = prez[q]; lg[i] = preg[q]; }
      }
#pragma unroll
      for (int q = 0; q < NAH; ++q) {
        const int i = a0 + lane + q * 64;
        if (i < a1) ldsA[i] = prea[q];
      }
    }
    const long next = unit + gridDim.x;
    if (next < units) issue(next);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float bt[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int w = 2 * q + mk;
      const float v = ldsA[mic * V + (w < V ? w : V - 1)];
      bt[q] = (w < V && mi < V) ? v : 0.f;
    }
    f32x16 accA, accB;            // two chains: consecutive MFMAs on one accumulator wait for each other
#pragma unroll
    for (int i = 0; i < 16; ++i) { accA[i] = 0.f; accB[i] = 0.f; }
    for (int j0 = 0; j0 < rows; j0 += 16) {
      float av[8], bv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int t = j0 + 2 * q + mk;
        const bool ok = (mi < V) && (t < rows);
        const int idx = (t < rows ? t : rows - 1) * V + mic;
        const float a = affine_act(ldsZ[idx], s, h, relu);
        const float b = ldsG[idx];
        av[q] = ok ? a : 0.f;
        bv[q] = ok ? b : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (j0 + 2 * q < rows) {
          if (q & 1) accB = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], accB, 0, 0, 0);
          else accA = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], accA, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) accA[i] += accB[i];
    if (wave > 0 && mi < V) {
      float* dd = ldsD + (wave - 1) * V * V;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int u = mfma_row(r, mk);
        if (u < V) dd[u * V + mi] = accA[r];
      }
    }
    float sum_h = 0.f, sum_s = 0.f;
    {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const int tc = mi < rows ? mi : rows - 1;
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        const int w = 2 * q + mk;
        const float v = ldsG[tc * V + (w < V ? w : V - 1)];
        const float a = (w < V && mi < rows) ? v : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bt[q], acc, 0, 0, 0);
      }
      wave_lds_sync();
      if (mi < V) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tt = mfma_row(r, mk);
          if (tt < rows) {
            const float z = ldsZ[tt * V + mi];
            const float pre = fmaf(z, s, h);
            const float dpre = (!relu || pre > 0.f) ? acc[r] : 0.f;
            sum_h += dpre;
            sum_s = fmaf(dpre, z, sum_s);
            ldsZ[tt * V + mi] = dpre * s;
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave == 0 && mi < V) {
      float* __restrict__ dA = dahat + (size_t)unit * V * V;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int u = mfma_row(r, mk);
        if (u < V) {
          float v = accA[r];
          for (int w = 0; w < NW - 1; ++w) v += ldsD[w * V * V + u * V + mi];
          __builtin_nontemporal_store(v, &dA[u * V + mi]);
        }
      }
    }
    {
      f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(dzp + ((size_t)unit * T + wave * HR) * V);
      const f32x4* l4 = reinterpret_cast<const f32x4*>(ldsZ);
#pragma unroll
      for (int q = 0; q < NP4; ++q) {
        const int i = lane + q * 64;
        if (i < c4) __builtin_nontemporal_store(l4[i], &d4[i]);
      }
    }
    sum_s = wave_sum(sum_s);
    sum_h = wave_sum(sum_h);
    if (lane == 0) {
      partial[((size_t)wave * units + unit) * 2 + 0] = sum_s;
      partial[((size_t)wave * units + unit) * 2 + 1] = sum_h;
    }
    wave_lds_sync();
    unit = next;
  }
}

int g_pipe_waves = 0;       // tuning knobs (dsgcn_set_tuning)
int g_pipe_waves_bwd = 0;
int g_fwd_direct = 0;      // accumulator -> LDS -> 16-byte stores (direct accumulator stores measured 1-2 % slower once the kernels were specialised on T)
int g_fwd_chunk = 32;
int g_bwd_variant = 0;      // 1 = one-shot kernel (A/B)
int g_bwd_da_lds = 0;       // dAhat leaves through LDS (A/B)
int g_bwd_pair = 1;         // two waves per unit when 32 < T <= 64
int g_pair_wgs = 0;

template <int V>
int launch_fwd(const float* zp, const float* scale, const float* shift, int relu, const float* ahat, float* y,
               long units, int KC, int T, int variant, hipStream_t st) {
  const int vec = ((T * V) % 4 == 0) ? 1 : 0;
  dim3 grid((unsigned)units, (unsigned)((T + 63) / 64));
#ifdef DSGCN_LAB
  if (variant == 1) {
    const size_t lds = (size_t)64 * V * sizeof(float);
    hipLaunchKernelGGL((k_aggregate_fwd_valu<V, 5>), grid, dim3(64), lds, st, zp, scale, shift, relu, ahat,
                       (long)V * V, 0, y, KC, T, vec);
  } else
#endif
  if (variant == 2 || !vec) {
    const size_t lds = (size_t)(64 * V + V * V) * sizeof(float);
    hipLaunchKernelGGL((k_aggregate_fwd<V>), grid, dim3(64), lds, st, zp, scale, shift, relu, ahat, y, KC, T, vec);
  } else {
    // 32-frame work items when a unit has more than 32 frames (two items per wave keep the prefetch pipeline busy on
    // the small early layers); (32*V) % 4 == 0 keeps every chunk 16-B aligned
    const bool half = (T > 32) && ((32 * V) % 4 == 0) && g_fwd_chunk != 64;
    const int CHv = half ? 32 : 64;
    const size_t lds = (size_t)(CHv * V + V * V) * sizeof(float);
    const int chunks = (T + CHv - 1) / CHv;
    const long items = units * chunks;
    int waves = g_pipe_waves > 0 ? g_pipe_waves : 3072;                     // measured: tools/ka_sweep.py
    // equal items per wave where possible
    const long per = (items + waves - 1) / waves;
    const long g = (items + per - 1) / per;
    if (half && T % 32 == 0)
      hipLaunchKernelGGL((k_aggregate_fwd_pipe<V, 32, 32>), dim3((unsigned)((g + KA_WPB - 1) / KA_WPB)), dim3(64 * KA_WPB), lds * KA_WPB, st, zp, scale, shift, relu,
                         ahat, y, KC, T, chunks, items, g_fwd_direct);
    else if (half)
      hipLaunchKernelGGL((k_aggregate_fwd_pipe<V, 32>), dim3((unsigned)((g + KA_WPB - 1) / KA_WPB)), dim3(64 * KA_WPB), lds * KA_WPB, st, zp, scale, shift, relu,
                         ahat, y, KC, T, chunks, items, g_fwd_direct);
    else if (T == 32)
      hipLaunchKernelGGL((k_aggregate_fwd_pipe<V, 64, 32>), dim3((unsigned)((g + KA_WPB - 1) / KA_WPB)), dim3(64 * KA_WPB), lds * KA_WPB, st, zp, scale, shift, relu,
                         ahat, y, KC, T, chunks, items, g_fwd_direct);
    else if (T == 16)
      hipLaunchKernelGGL((k_aggregate_fwd_pipe<V, 64, 16>), dim3((unsigned)((g + KA_WPB - 1) / KA_WPB)), dim3(64 * KA_WPB), lds * KA_WPB, st, zp, scale, shift, relu,
                         ahat, y, KC, T, chunks, items, g_fwd_direct);
    else
      hipLaunchKernelGGL((k_aggregate_fwd_pipe<V, 64>), dim3((unsigned)((g + KA_WPB - 1) / KA_WPB)), dim3(64 * KA_WPB), lds * KA_WPB, st, zp, scale, shift, relu,
                         ahat, y, KC, T, chunks, items, g_fwd_direct);
  }
  DSGCN_LAUNCH_CHECK();
  return 0;
}

template <int V>
int bwd_pair_hr(int T) {      // frames per wave of the multi-wave backward, 0 = not used
  if (!g_bwd_pair || g_bwd_variant != 0 || T > 64 || (T * V) % 4 != 0) return 0;
  const int hr = 32;            // 16-frame pieces (4 waves per unit) measured slower: tools/ka_variants.py
  if (T <= hr || (hr * V) % 4 != 0) return 0;
  return hr;
}

template <int V>
int launch_bwd(const float* zp, const float* scale, const float* shift, int relu, const float* ahat, const float* dy,
               float* dzp, float* dahat, float* partial, long units, int KC, int T, hipStream_t st) {
  const int vec = ((T * V) % 4 == 0) ? 1 : 0;
  const size_t lds = (size_t)(2 * 64 * V + V * V) * sizeof(float);
  const int hr = bwd_pair_hr<V>(T);
  if (hr) {
    const int nw = (T + hr - 1) / hr;
    const size_t lds2 = (size_t)(2 * nw * hr * V + nw * V * V) * sizeof(float);
    const int wgs = g_pair_wgs > 0 ? g_pair_wgs : 1536;
    const long per = (units + wgs - 1) / wgs;
    const long g = (units + per - 1) / per;
    if (T == 64)
      hipLaunchKernelGGL((k_aggregate_bwd_pair<V, 32, 2, true>), dim3((unsigned)g), dim3(64 * nw), lds2, st, zp, scale,
                         shift, relu, ahat, dy, dzp, dahat, partial, KC, T, units);
    else
      hipLaunchKernelGGL((k_aggregate_bwd_pair<V, 32, 2>), dim3((unsigned)g), dim3(64 * nw), lds2, st, zp, scale, shift,
                         relu, ahat, dy, dzp, dahat, partial, KC, T, units);
  } else if (vec && T <= 128 && g_bwd_variant == 0) {
    const size_t lds = (size_t)(2 * T * V + V * V) * sizeof(float);
    int waves = g_pipe_waves_bwd > 0 ? g_pipe_waves_bwd : (T == 32 ? 3072 : 2048);      // tools/ka_variants.py
    const long per = (units + waves - 1) / waves;
    const long g = (units + per - 1) / per;
    if (T > 64) {      // four waves' slices of up to 128 frames can pass 64 KB (V = 25, T = 100: 90 KB)
      static bool raised = false;
      if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_aggregate_bwd_pipe<V, false, 128>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
      }
    }
    if (T > 64)        // the long clips of BASELINE config 5 (V = 17, T = 100 / 50): four 32-frame tiles per unit
      hipLaunchKernelGGL((k_aggregate_bwd_pipe<V, false, 128>), dim3((unsigned)((g + KA_WPB - 1) / KA_WPB)), dim3(64 * KA_WPB), lds * KA_WPB, st, zp, scale, shift,
                         relu, ahat, dy, dzp, dahat, partial, KC, T, units);
    else if (g_bwd_da_lds)
      hipLaunchKernelGGL((k_aggregate_bwd_pipe<V, true, 64>), dim3((unsigned)((g + KA_WPB - 1) / KA_WPB)), dim3(64 * KA_WPB), lds * KA_WPB, st, zp, scale, shift, relu,
                         ahat, dy, dzp, dahat, partial, KC, T, units);
    else if (T == 32)
      hipLaunchKernelGGL((k_aggregate_bwd_pipe<V, false, 64, 32>), dim3((unsigned)((g + KA_WPB - 1) / KA_WPB)), dim3(64 * KA_WPB), lds * KA_WPB, st, zp, scale, shift,
                         relu, ahat, dy, dzp, dahat, partial, KC, T, units);
    else if (T == 16)
      hipLaunchKernelGGL((k_aggregate_bwd_pipe<V, false, 64, 16>), dim3((unsigned)((g + KA_WPB - 1) / KA_WPB)), dim3(64 * KA_WPB), lds * KA_WPB, st, zp, scale, shift,
                         relu, ahat, dy, dzp, dahat, partial, KC, T, units);
    else
      hipLaunchKernelGGL((k_aggregate_bwd_pipe<V, false, 64>), dim3((unsigned)((g + KA_WPB - 1) / KA_WPB)), dim3(64 * KA_WPB), lds * KA_WPB, st, zp, scale, shift, relu,
                         ahat, dy, dzp, dahat, partial, KC, T, units);
  } else {
    hipLaunchKernelGGL((k_aggregate_bwd<V>), dim3((unsigned)units), dim3(64), lds, st, zp, scale, shift, relu, ahat,
                       dy, dzp, dahat, partial, KC, T, vec);
  }
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" {

static int aggregate_fwd_dispatch(const float* zp, const float* scale, const float* shift, int relu, const float* ahat,
                                  float* y, int n, int KC, int T, int V, int variant, void* stream) {
  if (!zp || !ahat || !y || n <= 0 || KC <= 0 || T <= 0) return DSGCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long units = (long)n * KC;
  switch (V) {
    case 25: return launch_fwd<25>(zp, scale, shift, relu, ahat, y, units, KC, T, variant, st);
    case 17: return launch_fwd<17>(zp, scale, shift, relu, ahat, y, units, KC, T, variant, st);
    case 18: return launch_fwd<18>(zp, scale, shift, relu, ahat, y, units, KC, T, variant, st);
    default: return DSGCN_EUNSUPPORTED;
  }
}

// See include/dsgcn.h for the contract.
int dsgcn_aggregate_fwd(const float* zp, const float* scale, const float* shift, int relu, const float* ahat,
                        float* y, int n, int KC, int T, int V, void* stream) {
  return aggregate_fwd_dispatch(zp, scale, shift, relu, ahat, y, n, KC, T, V, 0, stream);
}

#ifdef DSGCN_LAB
// A/B only (tools/ka_variants.py): variant 1 = scalar-cache/VALU formulation, 2 = one-shot MFMA (no pipelining).
int dsgcn_aggregate_fwd_variant(const float* zp, const float* scale, const float* shift, int relu, const float* ahat,
                                float* y, int n, int KC, int T, int V, int variant, void* stream) {
  return aggregate_fwd_dispatch(zp, scale, shift, relu, ahat, y, n, KC, T, V, variant, stream);
}

int dsgcn_set_tuning(int key, int value) {
  if (key == 0) { g_pipe_waves = value; return 0; }
  if (key == 1) { g_pipe_waves_bwd = value; return 0; }
  if (key == 2) { g_bwd_variant = value; return 0; }
  if (key == 4) { g_fwd_direct = value; return 0; }
  if (key == 5) { g_fwd_chunk = value; return 0; }
  if (key == 6) { g_bwd_da_lds = value; return 0; }
  if (key == 7) { g_bwd_pair = value; return 0; }
  if (key == 8) { g_pair_wgs = value; return 0; }
  return DSGCN_EINVAL;
}

// A/B only: the scalar-cache/VALU formulation of the same product.
int dsgcn_aggregate_fwd_valu(const float* zp, const float* scale, const float* shift, int relu, const float* ahat,
                             float* y, int n, int KC, int T, int V, void* stream) {
  return aggregate_fwd_dispatch(zp, scale, shift, relu, ahat, y, n, KC, T, V, 1, stream);
}

#endif  // DSGCN_LAB

// rows of the backward's `partial` buffer: (rows, KC, 2); the sum over rows gives [d scale | d shift]
int dsgcn_aggregate_bwd_partial_rows(int n, int T, int V) {
  int hr = 0;
  switch (V) {
    case 25: hr = bwd_pair_hr<25>(T); break;
    case 17: hr = bwd_pair_hr<17>(T); break;
    case 18: hr = bwd_pair_hr<18>(T); break;
    default: break;
  }
  return hr ? n * ((T + hr - 1) / hr) : n;
}

int dsgcn_aggregate_bwd(const float* zp, const float* scale, const float* shift, int relu, const float* ahat,
                        const float* dy, float* dzp, float* dahat, float* partial, int n, int KC, int T, int V,
                        void* stream) {
  if (!zp || !ahat || !dy || !dzp || !dahat || !partial || n <= 0 || KC <= 0 || T <= 0) return DSGCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long units = (long)n * KC;
  switch (V) {
    case 25: return launch_bwd<25>(zp, scale, shift, relu, ahat, dy, dzp, dahat, partial, units, KC, T, st);
    case 17: return launch_bwd<17>(zp, scale, shift, relu, ahat, dy, dzp, dahat, partial, units, KC, T, st);
    case 18: return launch_bwd<18>(zp, scale, shift, relu, ahat, dy, dzp, dahat, partial, units, KC, T, st);
    default: return DSGCN_EUNSUPPORTED;
  }
}

}  // extern "C"
