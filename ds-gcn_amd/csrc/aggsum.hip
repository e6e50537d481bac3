// K-A': subset-summed gather-aggregate — the ST-GCN / CTR-GCN form of K-A.
//
//   Y[n,c,t,w] = sum_k sum_u P[n, k*Co+c, t, u] * Ahat_k[u,w]
//
//   ST-GCN  unit_gcn   (reference: pyskl/models/gcns/utils/gcn.py:81-85, einsum('nkctv,kvw->nctw')): Ahat_k = A[k], shared
//   CTR-GCN unit_ctrgcn (gcn.py:658 einsum('ncuv,nctu->nctv') + the sum over subsets gcn.py:917-919): Ahat_k = Ahat[n,k,c]
// The adjacency address is ahat + n*a_ns + k*a_ks + c*a_cs, so both are the same kernel.  One wave64 owns one (n,c)
// output plane: the K input planes stream through LDS one after the other and accumulate in the same f32 MFMA tile
// (v_mfma_f32_32x32x2_f32, i = frame, j = joint w, k = joint u), so the K partial products never exist in HBM
// (the reference writes and re-reads them: 2*K extra planes).  The epilogue stores straight from the accumulators and
// emits the per-plane sum / sum of squares for the BatchNorm that follows (gcn.py:86 / gcn.py:921).
// Algorithmic HBM bytes per (n,c): 4*((K+1)*T*V + K*V*V) forward.
//
// Backward (one wave per (n,c) as well): G = gy + A0[c] + B0[c]*y (deferred-BN statistics terms), then per subset
//   dP_k[t,u]   = sum_w G[t,w] * Ahat_k[u,w]      (i = frame, j = u, k = w)
//   dAhat_k[u,w] = sum_t P_k[t,u] * G[t,w]        (i = u, j = w, k = frames)
// dAhat_k is written at dahat + n*d_ns + k*d_ks + c*d_cs (for the shared form the caller sums the per-(n,c) pieces).
#include "common.h"

namespace {

__device__ __forceinline__ int as_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int V>
__device__ __forceinline__ void as_load_plane(const float* __restrict__ src, float* lds, int cnt, int lane, bool vec) {
  constexpr int NP4 = (64 * V / 4 + 63) / 64;
  if (vec) {
    const f32x4* __restrict__ s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* l4 = reinterpret_cast<f32x4*>(lds);
    const int c4 = cnt >> 2;
    f32x4 v[NP4];
#pragma unroll
    for (int q = 0; q < NP4; ++q) {
      const int i = lane + q * 64;
      if (i < c4) v[q] = s4[i];
    }
#pragma unroll
    for (int q = 0; q < NP4; ++q) {
      const int i = lane + q * 64;
      if (i < c4) l4[i] = v[q];
    }
  } else {
    for (int i = lane; i < cnt; i += 64) lds[i] = src[i];
  }
}

template <int V>
__device__ __forceinline__ void as_load_adj(const float* __restrict__ A, float* ldsA, int lane) {
  constexpr int NA = (V * V + 63) / 64;
  float v[NA];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int i = lane + q * 64;
    if (i < V * V) v[q] = A[i];
  }
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int i = lane + q * 64;
    if (i < V * V) ldsA[i] = v[q];
  }
}

template <int V>
__global__ __launch_bounds__(64) void k_aggsum_fwd(const float* __restrict__ p, const float* __restrict__ ahat,
                                                   long a_ns, long a_ks, long a_cs, float* __restrict__ y,
                                                   float* __restrict__ partial, int K, int Co, int T, int vec) {
  constexpr int KS = (V + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsP = lds;
  float* ldsA = lds + 64 * V;
  const int lane = threadIdx.x;
  const long unit = blockIdx.x;
  const int n = (int)(unit / Co), c = (int)(unit - (long)n * Co);
  const int mi = lane & 31, mk = lane >> 5;
  const int mic = mi < V ? mi : V - 1;
  float sum = 0.f, sq = 0.f;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int rows = min(64, T - t0);
    const int cnt = rows * V;
    f32x16 acc[2];
#pragma unroll
    for (int tile = 0; tile < 2; ++tile)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[tile][i] = 0.f;
    for (int k = 0; k < K; ++k) {
      const size_t plane = ((size_t)n * K + k) * Co + c;
      wave_lds_sync();                                   // the previous subset's operand reads are done
      as_load_plane<V>(p + (plane * T + t0) * V, ldsP, cnt, lane, vec);
      as_load_adj<V>(ahat + n * a_ns + k * a_ks + c * a_cs, ldsA, lane);
      wave_lds_sync();
      float b[KS];
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        const int u = 2 * q + mk;
        const float v = ldsA[(u < V ? u : V - 1) * V + mic];
        b[q] = (u < V && mi < V) ? v : 0.f;
      }
#pragma unroll
      for (int tile = 0; tile < 2; ++tile) {
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
    }
    float* __restrict__ yo = y + ((size_t)unit * T + t0) * V;
    if (mi < V) {
#pragma unroll
      for (int tile = 0; tile < 2; ++tile) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = tile * 32 + as_row(r, mk);
          if (t < rows) {
            const float v = acc[tile][r];
            yo[t * V + mi] = v;
            sum += v;
            sq = fmaf(v, v, sq);
          }
        }
      }
    }
  }
  if (partial) {
    sum = wave_sum(sum);
    sq = wave_sum(sq);
    if (lane == 0) {
      partial[unit * 2 + 0] = sum;
      partial[unit * 2 + 1] = sq;
    }
  }
}

template <int V>
__global__ __launch_bounds__(64) void k_aggsum_bwd(const float* __restrict__ p, const float* __restrict__ ahat,
                                                   long a_ns, long a_ks, long a_cs, const float* __restrict__ gy,
                                                   const float* __restrict__ y, const float* __restrict__ A0,
                                                   const float* __restrict__ B0, float* __restrict__ dp,
                                                   float* __restrict__ dahat, long d_ns, long d_ks, long d_cs, int K,
                                                   int Co, int T, int vec) {
  constexpr int KS = (V + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsZ = lds;                 // P_k chunk
  float* ldsG = lds + 64 * V;        // G chunk
  float* ldsA = lds + 128 * V;
  const int lane = threadIdx.x;
  const long unit = blockIdx.x;
  const int n = (int)(unit / Co), c = (int)(unit - (long)n * Co);
  const int mi = lane & 31, mk = lane >> 5;
  const int mic = mi < V ? mi : V - 1;
  const float a0 = A0 ? A0[c] : 0.f, b0 = (B0 && y) ? B0[c] : 0.f;
  const int chunks = (T + 63) / 64;
  for (int k = 0; k < K; ++k) {
    const size_t plane = ((size_t)n * K + k) * Co + c;
    wave_lds_sync();
    as_load_adj<V>(ahat + n * a_ns + k * a_ks + c * a_cs, ldsA, lane);
    f32x16 accA;
#pragma unroll
    for (int i = 0; i < 16; ++i) accA[i] = 0.f;
    float bt[KS];
    for (int t0 = 0; t0 < T; t0 += 64) {
      const int rows = min(64, T - t0);
      const int cnt = rows * V;
      const size_t offy = ((size_t)unit * T + t0) * V;
      if (t0 > 0) wave_lds_sync();
      if (k == 0 || chunks > 1) {
        // G = gy + A0 + B0*y
        if (vec) {
          const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(gy + offy);
          const f32x4* __restrict__ y4 = reinterpret_cast<const f32x4*>(y ? y + offy : gy + offy);
          f32x4* l4 = reinterpret_cast<f32x4*>(ldsG);
          const int c4 = cnt >> 2;
          for (int i = lane; i < c4; i += 64) {
            f32x4 g = g4[i];
            const f32x4 yy = y4[i];
            g.x += fmaf(b0, yy.x, a0); g.y += fmaf(b0, yy.y, a0); g.z += fmaf(b0, yy.z, a0); g.w += fmaf(b0, yy.w, a0);
            l4[i] = g;
          }
        } else {
          for (int i = lane; i < cnt; i += 64) ldsG[i] = gy[offy + i] + fmaf(b0, y ? y[offy + i] : 0.f, a0);
        }
      }
      as_load_plane<V>(p + (plane * T + t0) * V, ldsZ, cnt, lane, vec);
      wave_lds_sync();
      if (t0 == 0) {
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          const int w = 2 * q + mk;
          const float v = ldsA[mic * V + (w < V ? w : V - 1)];
          bt[q] = (w < V && mi < V) ? v : 0.f;
        }
      }
      // dAhat_k += P_k^T G  (operands of 8 k-steps read before their MFMAs are issued)
      for (int j0 = 0; j0 < rows; j0 += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int t = j0 + 2 * q + mk;
          const bool ok = (mi < V) && (t < rows);
          const int idx = (t < rows ? t : rows - 1) * V + mic;
          const float a = ldsZ[idx], b = ldsG[idx];
          av[q] = ok ? a : 0.f;
          bv[q] = ok ? b : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (j0 + 2 * q < rows) accA = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], accA, 0, 0, 0);
      }
      // dP_k = G . Ahat_k^T, stored straight from the accumulators
      float* __restrict__ dpo = dp + (plane * T + t0) * V;
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
          if (mi < V) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int tt = tile * 32 + as_row(r, mk);
              if (tt < rows) dpo[tt * V + mi] = acc[r];
            }
          }
        }
      }
    }
    float* __restrict__ dA = dahat + n * d_ns + k * d_ks + c * d_cs;
    if (mi < V) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int u = as_row(r, mk);
        if (u < V) dA[u * V + mi] = accA[r];
      }
    }
  }
}

// ---- pipelined forms (T*V % 4 == 0) -----------------------------------------------------------------------------
// Forward: persistent single-wave workgroups walk (unit, 32-frame chunk) items; inside an item the K subset planes
// stream through LDS one after the other into the same accumulator tile, and the NEXT plane / adjacency (of this
// item or of the wave's next item) is already in flight while the matrix core works on the current one.
// partial has one row per chunk: [chunk][unit][2].
// (Round 5: a variant that keeps a SHARED adjacency's fragments in registers for the whole launch measured the same —
// 69 vs 66-72 us per ST-GCN layer; the forward is bound by its loads in flight, see the geometry note at the launch.  The
// backward does gain from it, below.)
template <int V>
__global__ __launch_bounds__(64) void k_aggsum_fwd_pipe(const float* __restrict__ p, const float* __restrict__ ahat,
                                                        long a_ns, long a_ks, long a_cs, float* __restrict__ y,
                                                        float* __restrict__ partial, int K, int Co, int T, int chunks,
                                                        long items, long units, int half) {
  // half (shared adjacency, 16-frame planes): the launch sees pairs of channels as 32-frame planes (Co / 2 "channels" of
  // T = 32: the rows of channels 2c and 2c + 1 are adjacent in memory and share A), so no MFMA tile is half empty; only the
  // statistics know — rows 0-15 belong to channel 2c, rows 16-31 to 2c + 1
  constexpr int KS = (V + 1) / 2;
  constexpr int CH = 32;
  constexpr int NP4 = (CH * V / 4 + 63) / 64;
  constexpr int NA = (V * V + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsP = lds;
  float* ldsA = lds + CH * V;
  const int lane = threadIdx.x;
  const int mi = lane & 31, mk = lane >> 5;
  const int mic = mi < V ? mi : V - 1;
  f32x4 pre[NP4];
  float prea[NA];

  auto issue = [&](long item, int k) {
    const long unit = item / chunks;
    const int t0 = (int)(item - unit * chunks) * CH;
    const int c4 = (min(CH, T - t0) * V) >> 2;
    const long n = unit / Co;
    const int c = (int)(unit - n * Co);
    const size_t plane = ((size_t)n * K + k) * Co + c;
    const f32x4* __restrict__ s4 = reinterpret_cast<const f32x4*>(p + (plane * T + t0) * V);
    const float* __restrict__ A = ahat + n * a_ns + k * a_ks + c * a_cs;
#pragma unroll
    for (int q = 0; q < NP4; ++q) {
      const int i = lane + q * 64;
      if (i < c4) pre[q] = s4[i];
    }
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int i = lane + q * 64;
      if (i < V * V) prea[q] = A[i];
    }
  };

  long item = blockIdx.x;
  int k = 0;
  if (item < items) issue(item, 0);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  while (item < items) {
    const long unit = item / chunks;
    const int ch = (int)(item - unit * chunks);
    const int t0 = ch * CH;
    const int rows = min(CH, T - t0);
    const int c4 = (rows * V) >> 2;
    {
      f32x4* l4 = reinterpret_cast<f32x4*>(ldsP);
#pragma unroll
      for (int q = 0; q < NP4; ++q) {
        const int i = lane + q * 64;
        if (i < c4) l4[i] = pre[q];
      }
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int i = lane + q * 64;
        if (i < V * V) ldsA[i] = prea[q];
      }
    }
    int nk = k + 1;
    long nitem = item;
    if (nk == K) { nk = 0; nitem = item + gridDim.x; }
    if (nitem < items) issue(nitem, nk);
    wave_lds_sync();
    {
      float b[KS];
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        const int u = 2 * q + mk;
        const float v = ldsA[(u < V ? u : V - 1) * V + mic];
        b[q] = (u < V && mi < V) ? v : 0.f;
      }
      const int tc = mi < rows ? mi : rows - 1;
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        const int u = 2 * q + mk;
        const float v = ldsP[tc * V + (u < V ? u : V - 1)];
        const float a = (u < V && mi < rows) ? v : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[q], acc, 0, 0, 0);
      }
    }
    if (k == K - 1) {
      float* __restrict__ yo = y + ((size_t)unit * T + t0) * V;
      float sum = 0.f, sq = 0.f, sum1 = 0.f, sq1 = 0.f;
      // Y leaves through the wave's P slice (dead after the last subset's products): 16-byte row-major stores instead of
      // sixteen 4-byte stores of 25 lanes each (what the backward got in round 3)
      wave_lds_sync();
      if (mi < V) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = as_row(r, mk);
          if (t < rows) {
            const float v = acc[r];
            ldsP[t * V + mi] = v;
            if (half && r >= 8) {                  // (rows 16-31: accumulator registers 8-15)
              sum1 += v;
              sq1 = fmaf(v, v, sq1);
            } else {
              sum += v;
              sq = fmaf(v, v, sq);
            }
          }
        }
      }
      wave_lds_sync();
      {
        const f32x4* l4 = reinterpret_cast<const f32x4*>(ldsP);
        f32x4* __restrict__ y4 = reinterpret_cast<f32x4*>(yo);
#pragma unroll
        for (int q = 0; q < NP4; ++q) {
          const int i = lane + q * 64;
          if (i < c4) y4[i] = l4[i];
        }
      }
      if (partial) {
        sum = wave_sum(sum);
        sq = wave_sum(sq);
        if (half) {
          sum1 = wave_sum(sum1);
          sq1 = wave_sum(sq1);
          if (lane == 0) {
            partial[(size_t)unit * 4 + 0] = sum;
            partial[(size_t)unit * 4 + 1] = sq;
            partial[(size_t)unit * 4 + 2] = sum1;
            partial[(size_t)unit * 4 + 3] = sq1;
          }
        } else if (lane == 0) {
          partial[((size_t)ch * units + unit) * 2 + 0] = sum;
          partial[((size_t)ch * units + unit) * 2 + 1] = sq;
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    }
    wave_lds_sync();
    item = nitem;
    k = nk;
  }
}

// Backward: NW waves per unit (32 frames each, NW = ceil(T/32) <= 2).  G is staged once per unit, then the K subsets
// stream through: per subset each wave forms its part of dAhat_k = P_k^T G and its rows of dP_k = G Ahat_k^T.
// PER_UNIT: dAhat_k is summed over the waves through LDS and written per (n,c) (CTR-GCN).  Otherwise (shared A,
// K == 3) every wave keeps three running dA_k accumulators over all the units it walks and writes them once at
// the end: pieces[(wg*NW + wave)][k][V*V] -> dsgcn_colsum.
template <int V, int NW, bool PER_UNIT, bool SHR = false>
__global__ __launch_bounds__(64 * NW, NW == 2 ? 3 : 2) void k_aggsum_bwd_pipe(const float* __restrict__ p, const float* __restrict__ ahat,
                                                             long a_ns, long a_ks, long a_cs,
                                                             const float* __restrict__ gy, const float* __restrict__ y,
                                                             const float* __restrict__ A0, const float* __restrict__ B0,
                                                             float* __restrict__ dp, float* __restrict__ dahat,
                                                             long d_ns, long d_ks, long d_cs, int K, int Co, int T,
                                                             long units, int CHK, int half) {
  // CHK > 1 (shared adjacency, NW == 1): a plane's 32-frame chunks are items of their own — with the running dA accumulators
  // nothing ties the two halves of a 64-frame plane to one workgroup (the two-wave form measured 172 us where the same
  // bytes as one-wave items take 143)
  constexpr int KS = (V + 1) / 2;
  constexpr int HR = 32;
  constexpr int NP4 = (HR * V / 4 + 63) / 64;
  constexpr int AH = (V * V + NW - 1) / NW;
  constexpr int NAH = (AH + 63) / 64;
  constexpr int KACC = PER_UNIT ? 1 : 3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* ldsZ = lds + wave * HR * V;
  float* ldsG = lds + NW * HR * V + wave * HR * V;
  float* ldsA = lds + 2 * NW * HR * V;
  float* ldsD = ldsA + V * V;
  const int mi = lane & 31, mk = lane >> 5;
  const int mic = mi < V ? mi : V - 1;
  int ch = 0;                                     // chunk of the current item (CHK > 1 only)
  int rows = min(HR, T - wave * HR);
  int c4 = (rows * V) >> 2;
  const int a0 = wave * AH, a1 = min(V * V, a0 + AH);
  const bool has_y = (y != nullptr) && (B0 != nullptr);
  f32x4 prez[NP4], preg[NP4], prey[NP4];
  float prea[NAH];
  // shared adjacency (PER_UNIT == false, K == 3): the three A_k^T fragment sets stay in registers (see k_aggsum_fwd_pipe)
  const bool shreg = !PER_UNIT && SHR;
  float btS[PER_UNIT ? 1 : 3][KS];
  if (!PER_UNIT) {
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        const int w = 2 * q + mk;
        btS[kk][q] = (SHR && kk < K && w < V && mi < V) ? ahat[(size_t)kk * a_ks + mi * V + w] : 0.f;
      }
  }
  f32x16 accS[KACC];
#pragma unroll
  for (int j = 0; j < KACC; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) accS[j][i] = 0.f;

  auto issueG = [&](long unit, int chn) {
    const int c4 = (min(HR, T - (wave + chn) * HR) * V) >> 2;
    const size_t off = ((size_t)unit * T + (wave + chn) * HR) * V;
    const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(gy + off);
    const f32x4* __restrict__ y4 = reinterpret_cast<const f32x4*>((has_y ? y : gy) + off);
#pragma unroll
    for (int q = 0; q < NP4; ++q) {
      const int i = lane + q * 64;
      if (i < c4) { preg[q] = g4[i]; if (has_y) prey[q] = y4[i]; }
    }
  };
  auto issueK = [&](long unit, int chn, int k) {
    const int c4 = (min(HR, T - (wave + chn) * HR) * V) >> 2;
    const long n = unit / Co;
    const int c = (int)(unit - n * Co);
    const size_t plane = ((size_t)n * K + k) * Co + c;
    const f32x4* __restrict__ z4 = reinterpret_cast<const f32x4*>(p + (plane * T + (wave + chn) * HR) * V);
    const float* __restrict__ A = ahat + n * a_ns + k * a_ks + c * a_cs;
#pragma unroll
    for (int q = 0; q < NP4; ++q) {
      const int i = lane + q * 64;
      if (i < c4) prez[q] = z4[i];
    }
    if (!shreg) {
#pragma unroll
      for (int q = 0; q < NAH; ++q) {
        const int i = a0 + lane + q * 64;
        if (i < a1) prea[q] = A[i];
      }
    }
  };

  const long items = units * CHK;
  long item = blockIdx.x;
  if (item < items) { issueG(item / CHK, (int)(item % CHK)); issueK(item / CHK, (int)(item % CHK), 0); }
  while (item < items) {
    const long unit = item / CHK;
    ch = (int)(item - unit * CHK);
    rows = min(HR, T - (wave + ch) * HR);
    c4 = (rows * V) >> 2;
    const long n = unit / Co;
    const int c = (int)(unit - n * Co);
    // half: this "channel" is the pair (2c, 2c + 1) of 16-frame planes — float4 i of the tile belongs to the first while
    // 4 i < 16 V (the launcher checks 16 V % 4 == 0)
    const float ca = A0 ? A0[half ? 2 * c : c] : 0.f, cb = has_y ? B0[half ? 2 * c : c] : 0.f;
    const float ca1 = (half && A0) ? A0[2 * c + 1] : ca, cb1 = (half && has_y) ? B0[2 * c + 1] : cb;
    const int split4 = half ? 4 * V : (1 << 30);
    {
      f32x4* lg = reinterpret_cast<f32x4*>(ldsG);
#pragma unroll
      for (int q = 0; q < NP4; ++q) {
        const int i = lane + q * 64;
        if (i < c4) {
          f32x4 g = preg[q];
          const float ca_ = i < split4 ? ca : ca1, cb_ = i < split4 ? cb : cb1;
          if (has_y) {
            const f32x4 yy = prey[q];
            g.x += fmaf(cb_, yy.x, ca_); g.y += fmaf(cb_, yy.y, ca_); g.z += fmaf(cb_, yy.z, ca_); g.w += fmaf(cb_, yy.w, ca_);
          } else {
            g.x += ca_; g.y += ca_; g.z += ca_; g.w += ca_;
          }
          lg[i] = g;
        }
      }
    }
    const long next = item + gridDim.x;
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      {
        f32x4* lz = reinterpret_cast<f32x4*>(ldsZ);
#pragma unroll
        for (int q = 0; q < NP4; ++q) {
          const int i = lane + q * 64;
          if (i < c4) lz[i] = prez[q];
        }
        if (!shreg) {
#pragma unroll
          for (int q = 0; q < NAH; ++q) {
            const int i = a0 + lane + q * 64;
            if (i < a1) ldsA[i] = prea[q];
          }
        }
      }
      if (k + 1 < K) {
        issueK(unit, ch, k + 1);
      } else if (next < items) {
        issueG(next / CHK, (int)(next % CHK));
        issueK(next / CHK, (int)(next % CHK), 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (NW > 1) __builtin_amdgcn_s_barrier(); else __builtin_amdgcn_wave_barrier();
      float bt[KS];
      if (shreg) {
#pragma unroll
        for (int q = 0; q < KS; ++q)
          bt[q] = k == 0 ? btS[0][q] : (k == 1 ? btS[PER_UNIT ? 0 : 1][q] : btS[PER_UNIT ? 0 : 2][q]);
      } else {
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          const int w = 2 * q + mk;
          const float v = ldsA[mic * V + (w < V ? w : V - 1)];
          bt[q] = (w < V && mi < V) ? v : 0.f;
        }
      }
      f32x16 accA;
      if (PER_UNIT) {
#pragma unroll
        for (int i = 0; i < 16; ++i) accA[i] = 0.f;
      } else {
        accA = k == 0 ? accS[0] : (k == 1 ? accS[KACC > 1 ? 1 : 0] : accS[KACC > 2 ? 2 : 0]);
      }
      for (int j0 = 0; j0 < rows; j0 += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int t = j0 + 2 * q + mk;
          const bool ok = (mi < V) && (t < rows);
          const int idx = (t < rows ? t : rows - 1) * V + mic;
          const float a = ldsZ[idx], b = ldsG[idx];
          av[q] = ok ? a : 0.f;
          bv[q] = ok ? b : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (j0 + 2 * q < rows) accA = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], accA, 0, 0, 0);
      }
      if (!PER_UNIT) {
        if (k == 0) accS[0] = accA;
        else if (k == 1) accS[KACC > 1 ? 1 : 0] = accA;
        else accS[KACC > 2 ? 2 : 0] = accA;
      } else if (NW > 1 && wave > 0 && mi < V) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int u = as_row(r, mk);
          if (u < V) ldsD[u * V + mi] = accA[r];
        }
      }
      {
        const size_t plane = ((size_t)n * K + k) * Co + c;
        float* __restrict__ dpo = dp + (plane * T + (wave + ch) * HR) * V;
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
        // dP leaves through the wave's own P_k slice (dead after the dAhat product above): 16-byte row-major stores
        // instead of sixteen 4-byte stores of 25 lanes each (the scalar stores bounded this kernel: 1.6 TB/s)
        if (mi < V) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int tt = as_row(r, mk);
            if (tt < rows) ldsZ[tt * V + mi] = acc[r];
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        {
          const f32x4* lz = reinterpret_cast<const f32x4*>(ldsZ);
          f32x4* __restrict__ dp4 = reinterpret_cast<f32x4*>(dpo);
#pragma unroll
          for (int q = 0; q < NP4; ++q) {
            const int i = lane + q * 64;
            if (i < c4) dp4[i] = lz[i];
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (NW > 1) __builtin_amdgcn_s_barrier(); else __builtin_amdgcn_wave_barrier();
      if (PER_UNIT && wave == 0 && mi < V) {
        float* __restrict__ dA = dahat + n * d_ns + k * d_ks + c * d_cs;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int u = as_row(r, mk);
          if (u < V) dA[u * V + mi] = accA[r] + (NW > 1 ? ldsD[u * V + mi] : 0.f);
        }
      }
    }
    item = next;
  }
  if (!PER_UNIT && mi < V) {
    float* __restrict__ out = dahat + ((size_t)blockIdx.x * NW + wave) * 3 * V * V;
#pragma unroll
    for (int j = 0; j < KACC; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int u = as_row(r, mk);
        if (u < V) out[(size_t)j * V * V + u * V + mi] = accS[j][r];
      }
  }
}

int g_as_pipe = 1;         // 0: one-shot kernels only (A/B)
int g_as_half = 1;         // shared adjacency, 16-frame planes: channel pairs as 32-frame planes (0: half-empty tiles, A/B)
int g_as_shreg = 1;        // shared adjacency: its MFMA fragments held in registers (0: reloaded per unit and subset, A/B)
int g_as_waves_fwd = 0;
int g_as_wgs_bwd = 0;

template <int V>
bool as_pipe_ok(int T) { return g_as_pipe && (T * V) % 4 == 0 && (32 * V) % 4 == 0; }

inline long as_grid(long items, int want) {
  const long per = (items + want - 1) / want;
  return (items + per - 1) / per;
}

template <int V>
int as_launch_fwd(const float* p, const float* ahat, long a_ns, long a_ks, long a_cs, float* y, float* partial, int n,
                  int K, int Co, int T, hipStream_t st) {
  const int vec = ((T * V) % 4 == 0) ? 1 : 0;
  if (as_pipe_ok<V>(T)) {
    // shared adjacency on 16-frame planes: pairs of channels as one 32-frame plane (see the kernel)
    const int half = (a_ns == 0 && a_cs == 0 && T == 16 && Co % 2 == 0 && (16 * V) % 4 == 0 && g_as_half) ? 1 : 0;
    if (half) { Co /= 2; T = 32; }
    const int chunks = (T + 31) / 32;
    const long units = (long)n * Co, items = units * chunks;
    // persistent waves: 120 VGPRs allow four per SIMD = 4096 waves, a whole number of rounds.  Per-channel adjacency
    // (CTR-GCN): 4096 beat 3072 at 64 and 16 frames and tied at 32 (tools/kap_sweep.py, round 5: 67.8 -> 58.9, 124 -> 113,
    // 123 -> 105 us per layer); shared adjacency (ST-GCN; tools/kap_shared.py): 69 -> 56, 69 -> 56, 117 -> 100 us (2048:
    // 78 / 78 / 143; 6144 = one and a half rounds: 66 / 66 / 119; 8192: 58 / 57 / 101)
    const long g = as_grid(items, g_as_waves_fwd > 0 ? g_as_waves_fwd : 4096);
    const size_t lds = (size_t)(32 * V + V * V) * sizeof(float);
    hipLaunchKernelGGL((k_aggsum_fwd_pipe<V>), dim3((unsigned)g), dim3(64), lds, st, p, ahat, a_ns, a_ks, a_cs, y,
                       partial, K, Co, T, chunks, items, units, half);
  } else {
    const size_t lds = (size_t)(64 * V + V * V) * sizeof(float);
    hipLaunchKernelGGL((k_aggsum_fwd<V>), dim3((unsigned)((long)n * Co)), dim3(64), lds, st, p, ahat, a_ns, a_ks, a_cs,
                       y, partial, K, Co, T, vec);
  }
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// pieces rows of the shared-A pipelined backward (0: the one-shot kernel writes per-(n,c) pieces instead)
template <int V>
long as_bwd_piece_rows(int n, int K, int Co, int T) {
  if (!as_pipe_ok<V>(T) || T > 64 || K != 3) return 0;
  if (T == 16 && Co % 2 == 0 && (16 * V) % 4 == 0 && g_as_half) { Co /= 2; T = 32; }
  return as_grid((long)n * Co * ((T + 31) / 32), g_as_wgs_bwd > 0 ? g_as_wgs_bwd : 2048);      // one-wave workgroups
}

template <int V>
int as_launch_bwd(const float* p, const float* ahat, long a_ns, long a_ks, long a_cs, const float* gy, const float* y,
                  const float* A0, const float* B0, float* dp, float* dahat, long d_ns, long d_ks, long d_cs, int n,
                  int K, int Co, int T, hipStream_t st) {
  const int vec = ((T * V) % 4 == 0) ? 1 : 0;
  const bool shared = (a_ns == 0 && a_cs == 0);
  const int half = (shared && K == 3 && T == 16 && Co % 2 == 0 && (16 * V) % 4 == 0 && as_pipe_ok<V>(T) && g_as_half) ? 1 : 0;
  if (half) { Co /= 2; T = 32; }                      // channel pairs as 32-frame planes (see k_aggsum_fwd_pipe)
  const long units = (long)n * Co;
  if (as_pipe_ok<V>(T) && T <= 64 && (!shared || K == 3)) {
    const int nw = (T > 32 && !shared) ? 2 : 1;
    const int chk = shared ? (T + 31) / 32 : 1;       // shared adjacency: 32-frame chunks as one-wave items
    // per-(n,c) adjacency (CTR-GCN): 140 VGPRs = three waves per SIMD, which 2048 single-wave workgroups left one short of
    // (tools/kap_sweep.py, round 5: 3072 workgroups 210 -> 158, 373 -> 286, 315 -> 249 us on the 32- / 16-frame layers,
    // 142 -> 130, 252 -> 245 on the 64-frame ones).  The shared-adjacency form (190 / 168 VGPRs: two per SIMD) keeps its
    // geometry — dsgcn_aggsum_bwd_piece_rows sizes its pieces by it.
    const long g = as_grid(units * chk, g_as_wgs_bwd > 0 ? g_as_wgs_bwd : (shared ? 2048 : 3072));
    const size_t lds = (size_t)(2 * nw * 32 * V + 2 * V * V) * sizeof(float);
#define AS_BWD(NWV, PU, SR)                                                                                         \
  hipLaunchKernelGGL((k_aggsum_bwd_pipe<V, NWV, PU, SR>), dim3((unsigned)g), dim3(64 * NWV), lds, st, p, ahat, a_ns, a_ks, \
                     a_cs, gy, y, A0, B0, dp, dahat, d_ns, d_ks, d_cs, K, Co, T, units, chk, half)
    // (shared adjacency: its fragments in registers — 170 -> 143, 233 -> 196 us on ST-GCN's 32- and 16-frame layers; the
    // two-wave form of the 64-frame layers measured the same with it, 172 vs 172-178, and was replaced by chunk items)
    if (nw == 2) AS_BWD(2, true, false);
    else { if (shared) { if (g_as_shreg) AS_BWD(1, false, true); else AS_BWD(1, false, false); } else AS_BWD(1, true, false); }
#undef AS_BWD
  } else {
    const size_t lds = (size_t)(128 * V + V * V) * sizeof(float);
    hipLaunchKernelGGL((k_aggsum_bwd<V>), dim3((unsigned)units), dim3(64), lds, st, p, ahat, a_ns, a_ks, a_cs, gy, y,
                       A0, B0, dp, dahat, d_ns, d_ks, d_cs, K, Co, T, vec);
  }
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" {

// See include/dsgcn.h for the contract.
int dsgcn_aggsum_fwd(const float* p, const float* ahat, long a_ns, long a_ks, long a_cs, float* y, float* partial,
                     int n, int K, int Co, int T, int V, void* stream) {
  if (!p || !ahat || !y || n <= 0 || K <= 0 || Co <= 0 || T <= 0) return DSGCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  switch (V) {
    case 25: return as_launch_fwd<25>(p, ahat, a_ns, a_ks, a_cs, y, partial, n, K, Co, T, st);
    case 17: return as_launch_fwd<17>(p, ahat, a_ns, a_ks, a_cs, y, partial, n, K, Co, T, st);
    case 18: return as_launch_fwd<18>(p, ahat, a_ns, a_ks, a_cs, y, partial, n, K, Co, T, st);
    default: return DSGCN_EUNSUPPORTED;
  }
}

int dsgcn_aggsum_bwd(const float* p, const float* ahat, long a_ns, long a_ks, long a_cs, const float* gy,
                     const float* y, const float* A0, const float* B0, float* dp, float* dahat, long d_ns, long d_ks,
                     long d_cs, int n, int K, int Co, int T, int V, void* stream) {
  if (!p || !ahat || !gy || !dp || !dahat || n <= 0 || K <= 0 || Co <= 0 || T <= 0) return DSGCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  switch (V) {
    case 25: return as_launch_bwd<25>(p, ahat, a_ns, a_ks, a_cs, gy, y, A0, B0, dp, dahat, d_ns, d_ks, d_cs, n, K, Co, T, st);
    case 17: return as_launch_bwd<17>(p, ahat, a_ns, a_ks, a_cs, gy, y, A0, B0, dp, dahat, d_ns, d_ks, d_cs, n, K, Co, T, st);
    case 18: return as_launch_bwd<18>(p, ahat, a_ns, a_ks, a_cs, gy, y, A0, B0, dp, dahat, d_ns, d_ks, d_cs, n, K, Co, T, st);
    default: return DSGCN_EUNSUPPORTED;
  }
}

// rows of the forward's `partial` buffer (rows, Co, 2)
int dsgcn_aggsum_partial_rows(int n, int T, int V) {
  bool pipe = false;
  switch (V) {
    case 25: pipe = as_pipe_ok<25>(T); break;
    case 17: pipe = as_pipe_ok<17>(T); break;
    case 18: pipe = as_pipe_ok<18>(T); break;
    default: break;
  }
  return pipe ? n * ((T + 31) / 32) : n;
}

// Shared adjacency (a_ns == a_cs == 0): rows R > 0 means the backward writes dahat as (R, K, V, V) per-wave pieces
// (strides ignored); 0 means per-(n,c) pieces through the d_* strides.
int dsgcn_aggsum_bwd_piece_rows(int n, int K, int Co, int T, int V) {
  switch (V) {
    case 25: return (int)as_bwd_piece_rows<25>(n, K, Co, T);
    case 17: return (int)as_bwd_piece_rows<17>(n, K, Co, T);
    case 18: return (int)as_bwd_piece_rows<18>(n, K, Co, T);
    default: return 0;
  }
}

#ifdef DSGCN_LAB
int dsgcn_aggsum_tuning(int key, int value) {
  if (key == 0) { g_as_pipe = value; return 0; }
  if (key == 1) { g_as_waves_fwd = value; return 0; }
  if (key == 2) { g_as_wgs_bwd = value; return 0; }
  if (key == 3) { g_as_shreg = value; return 0; }
  if (key == 4) { g_as_half = value; return 0; }
  return DSGCN_EINVAL;
}
#endif

}  // extern "C"
