// K-B: dynamic-semantic adjacency build (reference: pyskl/models/gcns/utils/gcn.py:2240-2337).
//
//   proj (n, 4m + m*P, V) = [conv1 (2m) | conv2 (2m) | conv1_se (m*P, row c*P+p)] applied to xbar = time mean of the input
//   a_k = rows k*m+c, b_k = rows 2m+k*m+c (k = 0,1);  s[c,v] = row 4m + c*P + tau(v)                 (node-typed select)
//   D0 = a0[u]-b0[w];  D1 = We[eps(u,w)] (a1[u]-b1[w]) + be[eps(u,w)];  D2 = s[u]-s[w]                (edge-typed linear)
//   G_k[u,w] = sum_c x1_k[c,u] x2_k[c,w]  (x1=(a0,a1,s), x2=(b0,b1,s): conv2_se is dead, quirk Q1)
//   Ahat[k,c,u,w] = A[k,u,w] + alpha_k tanh(D_k[c,u,w]) + beta_k softmax_u(G_k)[u,w]
//
// The reference spends ~40 launches plus a 625-iteration host loop per layer on this and computes all 15 edge-typed /
// 5 node-typed variants before selecting one.  Here: the three mean-pooled projections are a K-C launch on xbar, the rest
// one launch each way.
//   * the edge-typed linear is factored: P_e = We[e] a1 + be[e], Q_e = We[e] b1 for all E classes,
//     D1[c,u,w] = P_eps(u,w)[c,u] - Q_eps(u,w)[c,w]: two LDS gathers per element instead of a m x m product per pair;
//   * the backward mirrors it: dP_e / dQ_e are masked row / column sums of dD1, then
//       d a1 = sum_e We[e]^T dP_e,  d b1 = sum_e We[e]^T dQ_e,  dWe[e] = dP_e a1^T + dQ_e b1^T,  dbe[e] = sum_u dP_e
//     with per-sample partials for the parameters (deterministic: summed by dsgcn_colsum, no float atomics).
// Round 3: the three subsets share nothing but the launch, so a workgroup is (sample, subset) — 3n workgroups instead of
// n (64 samples left three quarters of the chip idle: 190 us per 256-channel layer backward, 1.1 ms of the 13.3 ms
// step) — and the subset-1 workgroup's four small GEMMs (P/Q, the masked sums as a product with the one-hot class
// matrix, We^T dP / We^T dQ, dWe) run on the matrix core (v_mfma_f32_16x16x4_f32: full fp32 products) instead of
// per-thread dot products over LDS.
// Bound: HBM writes of Ahat forward, reads of dAhat backward.
#include "common.h"
#include "bn_jobs.h"

namespace {

constexpr int KSUB = 3;     // subsets: 2 plain + 1 semantic
constexpr int NT = 1024;    // the subset-1 LDS image allows one workgroup per CU, so it brings all 16 waves
constexpr int NW = NT / DSGCN_WAVE;
constexpr int CPC = 4;      // channels per thread in the backward's elementwise pass

struct DynDims { int n, ld, mid, V, P, E, pc0, pm; };     // ld: joint stride of the proj / dproj rows (>= V; 32 = padded)
// pc0, pm: the channel window [pc0, pc0+pm) of the subset this workgroup evaluates (forward: gridDim.z windows per
// subset; backward: all mid channels)

// projection slots: x1_k = (a0, a1, s)[k], x2_k = (b0, b1, s)[k]
__device__ __forceinline__ int proj_row(const DynDims& d, int k, int second, int c, int tau_v) {
  const int m = d.mid;
  if (k == 2) return 4 * m + c * d.P + tau_v;        // s (typed), both operands
  return (second ? 2 * m : 0) + k * m + c;           // a_k | b_k
}

// LDS carve (floats):  X1 [m][V] | X2 [m][V] | G [V][V] | col [V][2] | ET [V][V] (int) | PQ [2][E][pm][V]
//                      (bwd:  | SC [V][V] | SCp [ceil(m/CPC)][V][V] | dX1 [m][V] | dX2 [m][V])
__device__ __forceinline__ int lds_G(const DynDims& d) { return 2 * d.mid * d.V; }
__device__ __forceinline__ int lds_col(const DynDims& d) { return lds_G(d) + d.V * d.V; }
__device__ __forceinline__ int lds_ET(const DynDims& d) { return lds_col(d) + 2 * d.V; }
__device__ __forceinline__ int lds_PQ(const DynDims& d) { return lds_ET(d) + d.V * d.V; }
__device__ __forceinline__ int lds_end(const DynDims& d) { return lds_PQ(d) + 2 * d.E * d.pm * d.V; }

// One 16x16 tile of C = A(M x K) B(K x N) on v_mfma_f32_16x16x4_f32, operands through accessors (which return 0 outside
// their matrices).  Lane l holds A[i0 + (l&15)][k + (l>>4)], B[k + (l>>4)][j0 + (l&15)]; result register r is
// C[i0 + 4*(l>>4) + r][j0 + (l&15)].  The operands of UNR k-steps are fetched before their MFMAs: the accessors read
// global memory (L2) in three of the four products, and a wave that waits for every step's pair pays the round trip per
// step (measured: 31 + 17 + 35 us of a 95 us subset-1 workgroup with four steps in flight).
template <int UNR, class FA, class FB, class FS>
__device__ __forceinline__ void mm16(int i0, int j0, int kbeg, int kend, int lane, FA fa, FB fb, FS fs) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, lk = lane >> 4;
  for (int k0 = kbeg; k0 < kend; k0 += 4 * UNR) {
    float a[UNR], b[UNR];
#pragma unroll
    for (int s = 0; s < UNR; ++s) a[s] = fa(i0 + li, k0 + 4 * s + lk);
#pragma unroll
    for (int s = 0; s < UNR; ++s) b[s] = fb(k0 + 4 * s + lk, j0 + li);
#pragma unroll
    for (int s = 0; s < UNR; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) fs(i0 + 4 * lk + r, j0 + li, acc[r]);
}

// P_e / Q_e of the channel window for all classes (rows i = e*pm + c - pc0 of We, K = m = 4*KL): the tiles of a wave run
// as a software pipeline — the A rows (two 16-byte loads per lane at m = 32: lane group l>>4 owns KL consecutive k, the
// k order of a product is free as long as both operands agree) and the bias values of the NEXT tile are in flight while
// the current one is multiplied; fetched per tile they cost a round trip to L2 each (120 tiles over 16 waves: 31 us).
template <int KL>
__device__ __forceinline__ void pq_tiles(const DynDims& d, const float* __restrict__ we, const float* __restrict__ be,
                                         const float* X1, const float* X2, float* PQ, int lane, int wave) {
  const int m = d.mid, V = d.V;
  const int M = d.E * d.pm, mt = (M + 15) >> 4, nt = (V + 15) >> 4, njobs = 2 * mt * nt;
  const int li = lane & 15, lk = lane >> 4;
  auto row = [&](int i) { const int e_ = i / d.pm; return e_ * m + d.pc0 + (i - e_ * d.pm); };   // row of We / be
  float a[KL], bz[4], an[KL], bn[4];
  auto issue = [&](int job, float (&av)[KL], float (&bv)[4]) {
    const int pq = job / (mt * nt), rem = job - pq * mt * nt, it = rem / nt;
    const float* arow = we + (size_t)row(min(16 * it + li, M - 1)) * m + lk * KL;
    if constexpr (KL % 4 == 0) {
#pragma unroll
      for (int q = 0; q < KL / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(arow + 4 * q);
        av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < KL; ++q) av[q] = arow[q];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = be[row(min(16 * it + 4 * lk + r, M - 1))];
  };
  int job = wave;
  if (job < njobs) issue(job, a, bz);
  for (; job < njobs; job += NW) {
    if (job + NW < njobs) issue(job + NW, an, bn);
    const int pq = job / (mt * nt), rem = job - pq * mt * nt, it = rem / nt, jt = rem - it * nt;
    const float* src = pq ? X2 : X1;
    const int j = 16 * jt + li, jc = min(j, V - 1);
    const bool live = 16 * it + li < M;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float b[KL];
#pragma unroll
    for (int s_ = 0; s_ < KL; ++s_) b[s_] = src[(lk * KL + s_) * V + jc];
#pragma unroll
    for (int s_ = 0; s_ < KL; ++s_)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(live ? a[s_] : 0.f, j < V ? b[s_] : 0.f, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * it + 4 * lk + r;
      if (i < M && j < V) PQ[(pq * M + i) * V + j] = pq ? acc[r] : acc[r] + bz[r];
    }
#pragma unroll
    for (int q = 0; q < KL; ++q) a[q] = an[q];
#pragma unroll
    for (int r = 0; r < 4; ++r) bz[r] = bn[r];
  }
}

// Masked row (COL = 0) / column (COL = 1) sums of dD1 over the classes:
//   dP_e[c,u] = sum_{w: eps(u,w)=e} dD1[c,u,w];   dQ_e[c,w] = -sum_{u: eps(u,w)=e} dD1[c,u,w]
// = for every fixed joint x the product of the (m x V) slice of dD1 with the one-hot class matrix (V x E), one 16x16 tile
// per (x, channel tile, class tile); NS k-steps of 4 cover V.  dD1 is read back from global memory (L2): the slice of
// the next tile is in flight while the current one is multiplied.
template <int COL, int NS>
__device__ __forceinline__ void masked_sums(const float* dd1, const int* ET, float* dst, int m, int V, int E, int lane,
                                            int wave) {
  const int VV = V * V, ct = (m + 15) >> 4, et = (E + 15) >> 4, njobs = V * ct * et;
  const int li = lane & 15, lk = lane >> 4;
  int ko[NS];                               // element offset of this lane's k of step s inside a slice row (clamped)
  bool kv[NS];
#pragma unroll
  for (int s_ = 0; s_ < NS; ++s_) {
    const int kk = 4 * s_ + lk;
    kv[s_] = kk < V;
    ko[s_] = min(kk, V - 1) * (COL ? V : 1);
  }
  float a[NS], an[NS];
  auto issue = [&](int job, float (&av)[NS]) {
    const int x = job / (ct * et), r2 = job - x * ct * et, ic = r2 / et;
    const float* base = dd1 + (size_t)min(16 * ic + li, m - 1) * VV + (COL ? x : x * V);
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) av[s_] = base[ko[s_]];
  };
  int job = wave;
  if (job < njobs) issue(job, a);
  for (; job < njobs; job += NW) {
    if (job + NW < njobs) issue(job + NW, an);
    const int x = job / (ct * et), r2 = job - x * ct * et, ic = r2 / et, ie = r2 - ic * et;
    const int* erow = ET + (COL ? x : x * V);
    const int e = 16 * ie + li;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int cls[NS];
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) cls[s_] = erow[ko[s_]];
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s_], (kv[s_] && cls[s_] == e) ? 1.f : 0.f, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * ic + 4 * lk + r;
      if (c < m && e < E) dst[(e * m + c) * V + x] = COL ? -acc[r] : acc[r];
    }
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) a[s_] = an[s_];
  }
}

// Steps shared by forward and backward for subset k: projections -> LDS (node-typed row pick), Gram + column softmax,
// and for k = 1 the factored edge-typed linear of the channel window.
template <int MD>
__device__ __forceinline__ void dyn_prepare(const DynDims& d, int k, float* lds, const float* __restrict__ proj_n,
                                            const float* __restrict__ we, const float* __restrict__ be,
                                            const int* __restrict__ node_type, const int* __restrict__ edge_type) {
  const int tid = threadIdx.x;
  const int m = d.mid, V = d.V, VV = V * V;
  float* X1 = lds;
  float* X2 = lds + m * V;
  float* G = lds + lds_G(d);
  float* col = lds + lds_col(d);
  int* ET = reinterpret_cast<int*>(lds + lds_ET(d));
  float* PQ = lds + lds_PQ(d);
  for (int o = tid; o < 2 * m * V; o += NT) {
    const int second = o >= m * V, r = o - second * m * V;
    const int c = r / V, v = r - c * V;
    lds[o] = proj_n[proj_row(d, k, second, c, node_type[v]) * d.ld + v];
  }
  for (int i = tid; i < VV; i += NT) ET[i] = edge_type[i];
  __syncthreads();
  for (int i = tid; i < VV; i += NT) {
    const int u = i / V, w = i - u * V;
    float g = 0.f;
    for (int c = 0; c < m; ++c) g = fmaf(X1[c * V + u], X2[c * V + w], g);
    G[i] = g;
  }
  if (k == 1) {
    // P_e[c,v] = be[e,c] + sum_cc We[e,c,cc] a1[cc,v];  Q_e[c,v] = sum_cc We[e,c,cc] b1[cc,v]   (rows i = e*pm + c - pc0)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the tile decode runs on the SALU
    if constexpr (MD > 0 && MD % 4 == 0) {
      pq_tiles<MD / 4>(d, we, be, X1, X2, PQ, lane, wave);
    } else {
      const int M = d.E * d.pm, mt = (M + 15) >> 4, nt = (V + 15) >> 4;
      for (int job = wave; job < 2 * mt * nt; job += NW) {
        const int pq = job / (mt * nt), rem = job - pq * mt * nt, it = rem / nt, jt = rem - it * nt;
        const float* src = pq ? X2 : X1;
        auto row = [&](int i) { const int e_ = i / d.pm; return e_ * m + d.pc0 + (i - e_ * d.pm); };
        mm16<4>(16 * it, 16 * jt, 0, m, lane,
                [&](int i, int kk) {
                  const float v = we[(size_t)row(min(i, M - 1)) * m + min(kk, m - 1)];
                  return (i < M && kk < m) ? v : 0.f;
                },
                [&](int kk, int j) { return (kk < m && j < V) ? src[kk * V + j] : 0.f; },
                [&](int i, int j, float v) {
                  if (i < M && j < V) PQ[(pq * M + i) * V + j] = pq ? v : v + be[row(i)];
                });
      }
    }
  }
  __syncthreads();
  if (tid < V) {
    const int w = tid;
    float mx = -INFINITY;
    for (int u = 0; u < V; ++u) mx = fmaxf(mx, G[u * V + w]);
    float ssum = 0.f;
    for (int u = 0; u < V; ++u) ssum += expf(G[u * V + w] - mx);
    col[w * 2 + 0] = mx;
    col[w * 2 + 1] = 1.f / ssum;
  }
  __syncthreads();
  for (int i = tid; i < VV; i += NT) {
    const int w = i % V;
    G[i] = expf(G[i] - col[w * 2]) * col[w * 2 + 1];     // G now holds softmax_u
  }
  __syncthreads();
}

#ifdef DSGCN_LAB
// phase timestamps of the backward (100 MHz wall clock) of sample 0: [subset][phase]; read by dsgcn_dynadj_phases
__device__ long long g_dyn_stamp[KSUB][8];
#define DYN_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_dyn_stamp[k][i] = wall_clock64(); } while (0)
#else
#define DYN_STAMP(i) do {} while (0)
#endif

// the launch puts the heavier subset (1: it carries the edge-typed linear) first
__device__ __forceinline__ int subset_of(int y) { return y == 0 ? 1 : (y == 1 ? 0 : 2); }

// VT / MD: the joint count and the mid width as compile-time constants for the model's shapes (25 / 17 joints, mid
// 8 / 16 / 32), 0 = take them from the arguments.  Every phase decodes (c, u, w) from a flat index: with run-time V and
// mid that is two or three integer divisions (~30 instructions each) per element.
// grid (n, 3, windows): workgroup = (sample, subset, channel window).
template <int VT, int MD>
__global__ __launch_bounds__(NT) void k_dynadj_fwd(DynDims d_, const float* __restrict__ proj,
                                                   const float* __restrict__ A, const float* __restrict__ alpha,
                                                   const float* __restrict__ beta, const float* __restrict__ we,
                                                   const float* __restrict__ be, const int* __restrict__ node_type,
                                                   const int* __restrict__ edge_type, float* __restrict__ ahat,
                                                   BnFinTable jobs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= d_.n) {                   // hosted BatchNorm finalize jobs (bn_jobs.h): extra workgroups past the samples
    if (blockIdx.y == 0 && blockIdx.z == 0) {
      double (*jred)[8][2] = reinterpret_cast<double (*)[8][2]>(lds);        // 2 KB of the launch's dynamic LDS (>= 4 KB)
      bnj_dispatch(jobs, (int)blockIdx.x - d_.n, [&](const BnFinJob& J, int b) { bn_finalize_block(J, b, jred); });
    }
    return;
  }
  DynDims d = d_;
  if (VT) d.V = VT;
  if (MD) d.mid = MD;
  const int n = blockIdx.x, k = subset_of(blockIdx.y);
  const int m = d.mid, V = d.V, VV = V * V;
  d.pc0 = (m * (int)blockIdx.z) / (int)gridDim.z;
  d.pm = (m * ((int)blockIdx.z + 1)) / (int)gridDim.z - d.pc0;
  dyn_prepare<MD>(d, k, lds, proj + (size_t)n * (4 + d.P) * m * d.ld, we, be, node_type, edge_type);
  const float* X1 = lds;
  const float* X2 = lds + m * V;
  const float* Sm = lds + lds_G(d);
  const int* ET = reinterpret_cast<const int*>(lds + lds_ET(d));
  const float* Pe = lds + lds_PQ(d);
  const float* Qe = Pe + d.E * d.pm * V;
  float* out = ahat + ((size_t)n * KSUB + k) * m * VV;
  const float al = alpha[k], bt = beta[k];
  const float* Ak = A + k * VV;
  const int total = d.pm * VV;
  for (int i = threadIdx.x; i < total; i += NT) {
    const int cl = i / VV, r = i - cl * VV;
    const int u = r / V, w = r - u * V;
    const int c = d.pc0 + cl;
    float dk;
    if (k == 1) {
      const int e = ET[r];
      dk = Pe[(e * d.pm + cl) * V + u] - Qe[(e * d.pm + cl) * V + w];
    } else {
      dk = X1[c * V + u] - X2[c * V + w];
    }
    out[c * VV + r] = Ak[r] + al * tanhf(dk) + bt * Sm[r];
  }
}

// Backward, workgroup = (sample, subset).  Workspace `dd` (n,3,mid,V,V) receives dD_k so the row / column sums can read
// it back (it stays in L2).  Outputs: dproj (n, 4m+mP, V) (typed rows: only row c*P+tau(v) of joint v is non-zero); ppar
// (n, pstride) per-sample partials [sum_c dAhat (3VV) | dalpha (3) | dbeta (3) | dWe (E*m*m) | dbe (E*m)] — the sum over
// samples gives the parameter gradients.  Every element of both has exactly one writer.
template <int VT, int MD>
__global__ __launch_bounds__(NT) void k_dynadj_bwd(
    DynDims d_, const float* __restrict__ proj, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ we, const float* __restrict__ be, const int* __restrict__ node_type,
    const int* __restrict__ edge_type, const float* __restrict__ dahat, float* dd, float* __restrict__ dproj,
    float* __restrict__ ppar, int pstride, BnCoefTable jobs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[2][NW];
  if ((int)blockIdx.x >= d_.n) {                   // hosted BatchNorm coefficient jobs: extra workgroups past the samples
    if (blockIdx.y == 0) {
      double (*jred)[8][2] = reinterpret_cast<double (*)[8][2]>(lds);
      bnj_dispatch(jobs, (int)blockIdx.x - d_.n, [&](const BnCoefJob& J, int b) { bn_coef_rows_block(J, b, jred); });
    }
    return;
  }
  DynDims d = d_;
  if (VT) d.V = VT;
  if (MD) d.mid = MD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x, k = subset_of(blockIdx.y);
  const int m = d.mid, V = d.V, VV = V * V, E = d.E;
  d.pc0 = 0;
  d.pm = m;
  DYN_STAMP(0);
  dyn_prepare<MD>(d, k, lds, proj + (size_t)n * (4 + d.P) * m * d.ld, we, be, node_type, edge_type);
  DYN_STAMP(1);
  const float* X1 = lds;
  const float* X2 = lds + m * V;
  const float* Sm = lds + lds_G(d);
  const int* ET = reinterpret_cast<const int*>(lds + lds_ET(d));
  float* PQ = lds + lds_PQ(d);
  const int nch = (m + CPC - 1) / CPC;
  float* SC = lds + lds_end(d);         // [V][V]  sum_c dAhat, then dG
  float* SCp = SC + VV;                 // [nch][V][V]  its per-channel-group pieces
  float* dX1 = SCp + nch * VV;          // [m][V] grads of x1_k, x2_k
  float* dX2 = dX1 + m * V;
  const float* g_k = dahat + ((size_t)n * KSUB + k) * m * VV;
  float* dd_k = dd + ((size_t)n * KSUB + k) * m * VV;
  float* par = ppar + (size_t)n * pstride;
  const float al = alpha[k], bk = beta[k];
  const int ksplit = (k == 1 && VV * (1 + nch) >= 4 * m * V) ? 2 : 1;

  // pass 1: thread = (channel group, u, w): dD -> workspace, the group's share of sum_c dAhat, dalpha partial
  float pal = 0.f;
  for (int i = tid; i < nch * VV; i += NT) {
    const int ch = i / VV, r = i - ch * VV, u = r / V, w = r - u * V;
    const int e = ET[r];
    float gv[CPC];
#pragma unroll
    for (int j = 0; j < CPC; ++j) gv[j] = (CPC * ch + j < m) ? g_k[(CPC * ch + j) * VV + r] : 0.f;
    float sc = 0.f;
#pragma unroll
    for (int j = 0; j < CPC; ++j) {
      const int c = CPC * ch + j;
      if (c < m) {
        const float dk = (k == 1) ? PQ[(e * m + c) * V + u] - PQ[((E + e) * m + c) * V + w] : X1[c * V + u] - X2[c * V + w];
        const float th = tanhf(dk);
        sc += gv[j];
        pal = fmaf(th, gv[j], pal);
        dd_k[c * VV + r] = al * (1.f - th * th) * gv[j];
      }
    }
    SCp[i] = sc;
  }
  __syncthreads();                      // (also: every dd_k element this block wrote is visible to the block below)
  DYN_STAMP(2);
  // sum_c dAhat (groups in order) -> partial row; dbeta partial
  float pbe = 0.f;
  for (int r = tid; r < VV; r += NT) {
    float sc = 0.f;
    for (int ch = 0; ch < nch; ++ch) sc += SCp[ch * VV + r];
    SC[r] = sc;
    par[k * VV + r] = sc;
    pbe = fmaf(Sm[r], sc, pbe);
  }
  __syncthreads();
  // softmax backward per column w: SC <- dG = Sm * (beta*SC - sum_u Sm*beta*SC)
  if (tid < V) {
    const int w = tid;
    float dot = 0.f;
    for (int u = 0; u < V; ++u) dot = fmaf(Sm[u * V + w], bk * SC[u * V + w], dot);
    for (int u = 0; u < V; ++u) SC[u * V + w] = Sm[u * V + w] * (bk * SC[u * V + w] - dot);
  }
  float* dP = PQ;                       // (P / Q are dead after pass 1)
  float* dQ = PQ + E * m * V;
  if (k == 1) {
    constexpr int NS = VT ? (VT + 3) / 4 : 8;          // k-steps that cover the joints (V <= 32)
    masked_sums<0, NS>(dd_k, ET, dP, m, V, E, lane, wave);
    masked_sums<1, NS>(dd_k, ET, dQ, m, V, E, lane, wave);
  }
  __syncthreads();
  DYN_STAMP(3);
  // Gram backward (+ for the plain subsets the row / column sums of dD): thread = (c, j)
  for (int o = tid; o < m * V; o += NT) {
    const int c = o / V, j = o - c * V;
    float rs = 0.f, cs = 0.f;
    if (k != 1) {
      const float* dk = dd_k + (size_t)c * VV;
#pragma unroll
      for (int w0 = 0; w0 < 32; w0 += 8) {
        float rv[8], cv[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          rv[b] = (w0 + b < V) ? dk[j * V + w0 + b] : 0.f;
          cv[b] = (w0 + b < V) ? dk[(w0 + b) * V + j] : 0.f;
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) { rs += rv[b]; cs += cv[b]; }
      }
    }
    float g1 = 0.f, g2 = 0.f;
    for (int w = 0; w < V; ++w) g1 = fmaf(SC[j * V + w], X2[c * V + w], g1);   // d x1_k[c,j]
    for (int u = 0; u < V; ++u) g2 = fmaf(SC[u * V + j], X1[c * V + u], g2);   // d x2_k[c,j]
    dX1[o] = rs + g1;
    dX2[o] = g2 - cs;
  }
  if (k == 1) {
    __syncthreads();
    DYN_STAMP(4);
    // d a1[c,j] += sum_ec We[ec,c] dP[ec,j];  d b1[c,j] += sum_ec We[ec,c] dQ[ec,j]          (2 * ct * vt long tiles)
    // dWe[ec,cc] = sum_u dP[ec,u] a1[cc,u] + sum_w dQ[ec,w] b1[cc,w]                          (mt * ct short tiles)
    const int EM = E * m, ct = (m + 15) >> 4, vt = (V + 15) >> 4, mt = (EM + 15) >> 4, Vp = (V + 3) & ~3;
    float* pwe = par + KSUB * VV + 6;
    float* pbe_ = pwe + EM * m;
    // the long tiles split K in two when the scratch T (SC | SCp: dead after the Gram step) holds both halves of both
    // operands; the halves are added, in order, when dproj is written
    const int KS = ksplit, kh = ((EM / KS + 3) & ~3);
    const int nlong = 2 * ct * vt * KS;
    for (int job = wave; job < nlong + mt * ct; job += NW) {
      if (job < nlong) {
        const int ks = job / (2 * ct * vt), r0 = job - ks * 2 * ct * vt;
        const int pq = r0 / (ct * vt), rem = r0 - pq * ct * vt, ic = rem / vt, jv = rem - ic * vt;
        const float* src = pq ? dQ : dP;
        float* dst = KS == 2 ? SC + (ks * 2 + pq) * m * V : (pq ? dX2 : dX1);
        const int kend = KS == 2 ? min(EM, (ks + 1) * kh) : EM;
        mm16<8>(16 * ic, 16 * jv, KS == 2 ? ks * kh : 0, kend, lane,
                [&](int c, int ec) {
                  const float v = we[(size_t)min(ec, kend - 1) * m + min(c, m - 1)];
                  return (c < m && ec < kend) ? v : 0.f;
                },
                [&](int ec, int j) { return (ec < kend && j < V) ? src[ec * V + j] : 0.f; },
                [&](int c, int j, float v) {
                  if (c < m && j < V) dst[c * V + j] = KS == 2 ? v : dst[c * V + j] + v;
                });
      } else {
        const int rem = job - nlong, it = rem / ct, jc = rem - it * ct;
        mm16<7>(16 * it, 16 * jc, 0, 2 * Vp, lane,
                [&](int ec, int kk) {
                  const int q = kk >= Vp, x = kk - q * Vp;
                  return (ec < EM && x < V) ? (q ? dQ : dP)[ec * V + x] : 0.f;
                },
                [&](int kk, int cc) {
                  const int q = kk >= Vp, x = kk - q * Vp;
                  return (cc < m && x < V) ? (q ? X2 : X1)[cc * V + x] : 0.f;
                },
                [&](int ec, int cc, float v) { if (ec < EM && cc < m) pwe[ec * m + cc] = v; });
      }
    }
    for (int ec = tid; ec < EM; ec += NT) {          // dbe[e,c] = sum_u dP_e[c,u]
      float acc = 0.f;
      for (int u = 0; u < V; ++u) acc += dP[ec * V + u];
      pbe_[ec] = acc;
    }
  }
  __syncthreads();
  DYN_STAMP(5);
  // this subset's dproj rows, joint stride ld (padding columns are written as zeros)
  {
    const int ld = d.ld;
    float* dp_n = dproj + (size_t)n * (4 + d.P) * m * ld;
    if (k < 2) {
      for (int o = tid; o < 2 * m * ld; o += NT) {
        const int q = o / ld, v = o - q * ld;
        const int second = q >= m, c = q - second * m;
        float val = v < V ? (second ? dX2 : dX1)[c * V + v] : 0.f;
        if (ksplit == 2 && v < V) val += SC[second * m * V + c * V + v] + SC[(2 + second) * m * V + c * V + v];
        dp_n[(size_t)proj_row(d, k, second, c, 0) * ld + v] = val;
      }
    } else {
      for (int o = tid; o < m * d.P * ld; o += NT) {
        const int q = o / ld, v = o - q * ld;
        const int c = q / d.P, p = q - c * d.P;
        dp_n[(size_t)(4 * m + q) * ld + v] = (v < V && node_type[v] == p) ? dX1[c * V + v] + dX2[c * V + v] : 0.f;
      }
    }
  }
  // block-reduce the two scalar partials (waves in order)
  {
    const float ra = wave_sum(pal), rb = wave_sum(pbe);
    if (lane == 0) { red[0][wave] = ra; red[1][wave] = rb; }
    __syncthreads();
    if (tid < 2) {
      float r = 0.f;
      for (int i = 0; i < NW; ++i) r += red[tid][i];
      par[KSUB * VV + 3 * tid + k] = r;
    }
  }
  DYN_STAMP(6);
}

size_t dyn_lds_bytes(int mid, int V, int E, bool bwd, int pm) {
  size_t f = 2 * (size_t)mid * V + 2 * (size_t)V * V + 2 * (size_t)V + 2 * (size_t)E * pm * V;
  if (bwd) f += (size_t)V * V * (1 + (mid + CPC - 1) / CPC) + 2 * (size_t)mid * V;
  return f * sizeof(float);
}

}  // namespace

// the model's (V, mid) combinations get their own instantiation, everything else the run-time form
#define DYN_DISPATCH(L)                                   \
  if (V == 25 && mid == 8) L(25, 8)                       \
  else if (V == 25 && mid == 16) L(25, 16)                \
  else if (V == 25 && mid == 32) L(25, 32)                \
  else if (V == 17 && mid == 8) L(17, 8)                  \
  else if (V == 17 && mid == 16) L(17, 16)                \
  else if (V == 17 && mid == 32) L(17, 32)                \
  else L(0, 0)

extern "C" {

// floats per sample of the backward's parameter-partial buffer
int dsgcn_dynadj_partial_stride(int mid, int V, int E) { return KSUB * V * V + 6 + E * mid * mid + E * mid; }

int dsgcn_dynadj_fwd_jobs(const float* proj, const float* A, const float* alpha, const float* beta, const float* we,
                          const float* be, const int* node_type, const int* edge_type, float* ahat, int n, int mid, int V,
                          int ld, int P, int E, const dsgcn_bn_fin_job* jobs, int njobs, void* stream) {
  if (!proj || !A || !alpha || !beta || !ahat || !we || !be || !node_type || !edge_type || n <= 0 || mid <= 0)
    return DSGCN_EINVAL;
  if (njobs < 0 || njobs > BNJ_MAX || (njobs > 0 && !jobs)) return DSGCN_EINVAL;
  BnFinTable jt = {};
  jt.n = njobs;
  for (int i = 0; i < njobs; ++i) {
    const dsgcn_bn_fin_job& q = jobs[i];
    jt.j[i] = BnFinJob{q.partial, q.gamma, q.beta, q.mean, q.var, q.scale, q.shift, q.count, q.eps, q.nblk, q.C, q.c_affine};
  }
  if (!bnj_fin_ok(jt)) return DSGCN_EINVAL;
  const int jblocks = bnj_total_blocks(jt);
  if (V > 32 || mid > 64 || ld < V) return DSGCN_EUNSUPPORTED;      // (the LDS check below is what bounds mid for a given E)
  // channel windows per (sample, subset): enough workgroups to fill the chip at small batches, windows of >= 8 channels
  int split = (n <= 128 && mid >= 16) ? 2 : 1;
  size_t lds = dyn_lds_bytes(mid, V, E, false, (mid + split - 1) / split);
  if (njobs && lds < 2048) lds = 2048;          // the hosted job blocks reduce through 2 KB of it
  if (lds > 158 * 1024) return DSGCN_EUNSUPPORTED;
  DynDims d{n, ld, mid, V, P, E, 0, mid};
#define DYN_FWD(VT, MD)                                                                                               \
  {                                                                                                                   \
    static size_t attr = 64 * 1024;                                                                                   \
    if (lds > attr) {                                                                                                 \
      hipError_t e = hipFuncSetAttribute((const void*)k_dynadj_fwd<VT, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         158 * 1024);                                                                 \
      if (e != hipSuccess) return (int)e;                                                                             \
      attr = 158 * 1024;                                                                                              \
    }                                                                                                                 \
    hipLaunchKernelGGL((k_dynadj_fwd<VT, MD>), dim3(n + jblocks, KSUB, split), dim3(NT), lds, (hipStream_t)stream, d, proj, A,  \
                       alpha, beta, we, be, node_type, edge_type, ahat, jt);                                           \
  }
  DYN_DISPATCH(DYN_FWD)
#undef DYN_FWD
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_dynadj_fwd(const float* proj, const float* A, const float* alpha, const float* beta, const float* we,
                     const float* be, const int* node_type, const int* edge_type, float* ahat, int n, int mid, int V,
                     int ld, int P, int E, void* stream) {
  return dsgcn_dynadj_fwd_jobs(proj, A, alpha, beta, we, be, node_type, edge_type, ahat, n, mid, V, ld, P, E, nullptr, 0, stream);
}

int dsgcn_dynadj_bwd_jobs(const float* proj, const float* alpha, const float* beta, const float* we, const float* be,
                          const int* node_type, const int* edge_type, const float* dahat, float* dd_ws, float* dproj,
                          float* ppar, int pstride, int n, int mid, int V, int ld, int P, int E,
                          const dsgcn_bn_coef_job* jobs, int njobs, void* stream) {
  if (!proj || !alpha || !beta || !we || !be || !node_type || !edge_type || !dahat || !dd_ws || !dproj || !ppar ||
      n <= 0 || mid <= 0)
    return DSGCN_EINVAL;
  if (njobs < 0 || njobs > BNJ_MAX || (njobs > 0 && !jobs)) return DSGCN_EINVAL;
  BnCoefTable jt = {};
  jt.n = njobs;
  for (int i = 0; i < njobs; ++i) {
    const dsgcn_bn_coef_job& q = jobs[i];
    jt.j[i] = BnCoefJob{q.part, q.mean, q.var, q.gamma, q.coef, q.count, q.eps, q.R, q.C, q.k, q.i_ds, q.i_dh, q.c_affine,
                        q.accumulate};
  }
  if (!bnj_coef_ok(jt)) return DSGCN_EINVAL;
  const int jblocks = bnj_total_blocks(jt);
  if (V > 32 || mid > 64 || ld < V) return DSGCN_EUNSUPPORTED;      // (the LDS check below is what bounds mid for a given E)
  if (pstride < dsgcn_dynadj_partial_stride(mid, V, E)) return DSGCN_EINVAL;
  size_t lds = dyn_lds_bytes(mid, V, E, true, mid);
  if (njobs && lds < 2048) lds = 2048;
  if (lds > 158 * 1024) return DSGCN_EUNSUPPORTED;
  DynDims d{n, ld, mid, V, P, E, 0, mid};
#define DYN_BWD(VT, MD)                                                                                               \
  {                                                                                                                   \
    static size_t attr = 64 * 1024;                                                                                   \
    if (lds > attr) {                                                                                                 \
      hipError_t e = hipFuncSetAttribute((const void*)k_dynadj_bwd<VT, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         158 * 1024);                                                                 \
      if (e != hipSuccess) return (int)e;                                                                             \
      attr = 158 * 1024;                                                                                              \
    }                                                                                                                 \
    hipLaunchKernelGGL((k_dynadj_bwd<VT, MD>), dim3(n + jblocks, KSUB), dim3(NT), lds, (hipStream_t)stream, d, proj, alpha, beta, we, \
                       be, node_type, edge_type, dahat, dd_ws, dproj, ppar, pstride, jt);                             \
  }
  DYN_DISPATCH(DYN_BWD)
#undef DYN_BWD
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_dynadj_bwd(const float* proj, const float* alpha, const float* beta, const float* we, const float* be,
                     const int* node_type, const int* edge_type, const float* dahat, float* dd_ws, float* dproj,
                     float* ppar, int pstride, int n, int mid, int V, int ld, int P, int E, void* stream) {
  return dsgcn_dynadj_bwd_jobs(proj, alpha, beta, we, be, node_type, edge_type, dahat, dd_ws, dproj, ppar, pstride, n, mid,
                               V, ld, P, E, nullptr, 0, stream);
}

#ifdef DSGCN_LAB
// out[3][8]: wall-clock stamps (10 ns units) of the last backward launch's sample-0 workgroups, per subset
int dsgcn_dynadj_phases(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dyn_stamp), sizeof(long long) * KSUB * 8);
}
#endif

}  // extern "C"
