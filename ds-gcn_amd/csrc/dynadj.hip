// K-B: dynamic-semantic adjacency build (reference: pyskl/models/gcns/utils/gcn.py:2240-2337).
//
//   proj (n,9*mid,V) = [W1;W2;Wse] xbar + bias (the K-C kernel on the time-mean xbar)  ->  Ahat (n, 3*mid, V, V)
//   a_k = proj rows k*mid+c, b_k = rows 2mid+k*mid+c (k=0,1);  s[c,v] = proj row 4mid + c*P+tau(v)
//   D0 = a0[u]-b0[w];  D1 = We[eps(u,w)] (a1[u]-b1[w]) + be[eps(u,w)];  D2 = s[u]-s[w]
//   G_k[u,w] = sum_c x1_k[c,u] x2_k[c,w]  (x1=(a0,a1,s), x2=(b0,b1,s): conv2_se is dead, quirk Q1)
//   Ahat[k,c,u,w] = A[k,u,w] + alpha_k tanh(D_k[c,u,w]) + beta_k softmax_u(G_k)[u,w]
//
// The reference spends ~40 launches plus a 625-iteration host loop per layer on this and computes all
// 15 edge-typed / 5 node-typed variants before selecting one.  Here one workgroup per sample keeps
// everything V x V in LDS, indexes the typed weight slice directly with tau(v) / eps(u,w) (625-entry
// integer tables), and only Ahat ever reaches HBM (coalesced).  Bound: HBM writes of Ahat.
//
// Backward = the same recompute + the chain rule in LDS (k_dynadj_bwd); the projection weights' gradients come from
// the K-C backward applied to dproj.
#include "common.h"

namespace {

constexpr int KSUB = 3;     // subsets: 2 plain + 1 semantic
constexpr int NT_DWE = 256; // the per-(sample, edge class) weight-gradient kernel has many small workgroups
constexpr int NT = 1024;    // threads per workgroup: the LDS image allows one workgroup per CU, so it has to bring all 16 waves

struct DynDims {
  int n, Ci, mid, V, P, E;
};

// LDS carve (floats): X[5*mid*V] | G[3*V*V] | col[3*V*2] | We[E*mid*mid] be[E*mid] | (bwd: SC[3*V*V] dX[5*mid*V])
__device__ __forceinline__ int lds_X(const DynDims& d) { return 0; }
__device__ __forceinline__ int lds_G(const DynDims& d) { return lds_X(d) + 5 * d.mid * d.V; }
__device__ __forceinline__ int lds_col(const DynDims& d) { return lds_G(d) + KSUB * d.V * d.V; }
__device__ __forceinline__ int lds_We(const DynDims& d) { return lds_col(d) + KSUB * d.V * 2; }
__device__ __forceinline__ int lds_end(const DynDims& d) { return lds_We(d) + d.E * d.mid * d.mid + d.E * d.mid; }

// X slot of x1_k / x2_k :  X[0]=a0 X[1]=a1 X[2]=s X[3]=b0 X[4]=b1
__device__ __forceinline__ int slot_x1(int k) { return k; }
__device__ __forceinline__ int slot_x2(int k) { return k < 2 ? 3 + k : 2; }

// Steps shared by forward and backward: projections -> LDS, edge-typed weights -> LDS, Gram, column softmax.
// proj_n (9*mid, V): rows [a (2mid) | b (2mid) | s-typed (mid*P, row c*P+p)] = the three mean-pooled 1x1 convs
// (conv1 / conv2 / conv1_se of gcn.py:2248-2253) evaluated by the K-C kernel on xbar; the node-typed select
// (gcn.py:2256-2259) is the row pick c*P+tau(v) done here.
__device__ void dyn_prepare(const DynDims& d, float* lds, const float* __restrict__ proj_n,
                            const float* __restrict__ we, const float* __restrict__ be,
                            const int* __restrict__ node_type) {
  const int tid = threadIdx.x;
  const int mid = d.mid, V = d.V;
  float* X = lds + lds_X(d);
  float* G = lds + lds_G(d);
  float* col = lds + lds_col(d);
  float* Wl = lds + lds_We(d);
  for (int o = tid; o < 5 * mid * V; o += NT) {
    const int q = o / V, v = o - q * V;
    int slot, c, row;
    if (q < 2 * mid) { slot = q / mid; c = q - slot * mid; row = q; }
    else if (q < 4 * mid) { const int r = q - 2 * mid; slot = 3 + r / mid; c = r % mid; row = q; }
    else { c = q - 4 * mid; slot = 2; row = 4 * mid + c * d.P + node_type[v]; }
    X[(slot * mid + c) * V + v] = proj_n[row * V + v];
  }
  {
    const int nw = d.E * mid * mid, nb = d.E * mid;
    for (int i = tid; i < nw; i += NT) Wl[i] = we[i];
    for (int i = tid; i < nb; i += NT) Wl[nw + i] = be[i];
  }
  __syncthreads();
  for (int i = tid; i < KSUB * V * V; i += NT) {
    const int k = i / (V * V), r = i - k * V * V, u = r / V, w = r - u * V;
    const float* x1 = X + slot_x1(k) * mid * V;
    const float* x2 = X + slot_x2(k) * mid * V;
    float g = 0.f;
    for (int c = 0; c < mid; ++c) g = fmaf(x1[c * V + u], x2[c * V + w], g);
    G[i] = g;
  }
  __syncthreads();
  for (int j = tid; j < KSUB * V; j += NT) {
    const int k = j / V, w = j - k * V;
    const float* g = G + k * V * V + w;
    float m = -INFINITY;
    for (int u = 0; u < V; ++u) m = fmaxf(m, g[u * V]);
    float ssum = 0.f;
    for (int u = 0; u < V; ++u) ssum += expf(g[u * V] - m);
    col[j * 2 + 0] = m;
    col[j * 2 + 1] = 1.f / ssum;
  }
  __syncthreads();
  for (int i = tid; i < KSUB * V * V; i += NT) {
    const int k = i / (V * V), r = i - k * V * V, w = r % V;
    const int j = k * V + w;
    G[i] = expf(G[i] - col[j * 2]) * col[j * 2 + 1];   // G now holds softmax_u
  }
  __syncthreads();
}

// D_k[c,u,w] before tanh (Wl = edge-typed weights then biases, in LDS)
__device__ __forceinline__ float dyn_D(const DynDims& d, const float* X, const float* Wl,
                                       const int* __restrict__ edge_type, int k, int c, int u, int w) {
  const int mid = d.mid, V = d.V;
  if (k == 0) return X[(0 * mid + c) * V + u] - X[(3 * mid + c) * V + w];
  if (k == 2) return X[(2 * mid + c) * V + u] - X[(2 * mid + c) * V + w];
  const int e = edge_type[u * V + w];
  const float* wr = Wl + (e * mid + c) * mid;
  float acc = Wl[d.E * mid * mid + e * mid + c];
  const float* a1 = X + 1 * mid * V;
  const float* b1 = X + 4 * mid * V;
  for (int cc = 0; cc < mid; ++cc) acc = fmaf(wr[cc], a1[cc * V + u] - b1[cc * V + w], acc);
  return acc;
}

__global__ __launch_bounds__(NT) void k_dynadj_fwd(DynDims d, const float* __restrict__ proj,
                                                   const float* __restrict__ A, const float* __restrict__ alpha,
                                                   const float* __restrict__ beta, const float* __restrict__ we,
                                                   const float* __restrict__ be, const int* __restrict__ node_type,
                                                   const int* __restrict__ edge_type, float* __restrict__ ahat) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n = blockIdx.x;
  const int mid = d.mid, V = d.V, VV = V * V;
  dyn_prepare(d, lds, proj + (size_t)n * 9 * mid * V, we, be, node_type);
  const float* X = lds + lds_X(d);
  const float* Sm = lds + lds_G(d);
  const float* Wl = lds + lds_We(d);
  float* out = ahat + (size_t)n * KSUB * mid * VV;
  const int total = KSUB * mid * VV;
  const float al0 = alpha[0], al1 = alpha[1], al2 = alpha[2], be0 = beta[0], be1 = beta[1], be2 = beta[2];
  // gridDim.y workgroups share one sample (each repeats the small prepare step): at n = 128 one workgroup per sample
  // would leave half of the 256 CUs idle
  for (int i = threadIdx.x + NT * blockIdx.y; i < total; i += NT * gridDim.y) {
    const int k = i / (mid * VV);
    int r = i - k * mid * VV;
    const int c = r / VV;
    r -= c * VV;
    const int u = r / V, w = r - u * V;
    const float dk = dyn_D(d, X, Wl, edge_type, k, c, u, w);
    const float al = k == 0 ? al0 : (k == 1 ? al1 : al2), bt = k == 0 ? be0 : (k == 1 ? be1 : be2);
    out[i] = A[k * VV + r] + al * tanhf(dk) + bt * Sm[k * VV + r];
  }
}

// Backward, per sample.  Workspace `dd` (n,3,mid,V,V) receives dD_k (k=1: later d(delta)) so the
// row/column reductions and the per-edge-class weight gradient can read it back; it aliases nothing.
// Outputs: dproj (n,9mid,V) (typed rows: only row c*P+tau(v) of joint v is non-zero); pA (n,3,V,V) = sum_c dAhat;
// pab (n,6) = [dalpha_k | dbeta_k] partials.  ddelta (n,mid,V,V): second workspace for d(a1[u]-b1[w]).
// The edge-typed weight gradient is a separate, wider launch (k_dynadj_dwe) reading dd's k=1 slice.
__global__ __launch_bounds__(NT) void k_dynadj_bwd(
    DynDims d, const float* __restrict__ proj, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ we, const float* __restrict__ be, const int* __restrict__ node_type,
    const int* __restrict__ edge_type, const int* __restrict__ pair_order, const int* __restrict__ class_start,
    const float* __restrict__ dahat, float* dd, float* ddelta, float* __restrict__ dproj, float* __restrict__ pA,
    float* __restrict__ pab, int pstride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[6][NT / DSGCN_WAVE];
  const int tid = threadIdx.x;
  const int n = blockIdx.x;
  const int mid = d.mid, V = d.V, VV = V * V;
  dyn_prepare(d, lds, proj + (size_t)n * 9 * mid * V, we, be, node_type);
  const float* X = lds + lds_X(d);
  const float* Sm = lds + lds_G(d);
  const float* Wl = lds + lds_We(d);
  float* SC = lds + lds_end(d);       // [3][V][V]  sum_c dAhat, then dG
  float* dX = SC + KSUB * VV;         // [5][mid][V] grads of a0,a1,s,b0,b1
  const float* g_n = dahat + (size_t)n * KSUB * mid * VV;
  float* dd_n = dd + (size_t)n * KSUB * mid * VV;

  // pass 1: thread = (k,u,w), loop channels: SC = sum_c dAhat, dalpha partial, dD -> workspace
  float pal0 = 0.f, pal1 = 0.f, pal2 = 0.f;
  for (int i = tid; i < KSUB * VV; i += NT) {
    const int k = i / VV, r = i - k * VV, u = r / V, w = r - u * V;
    const float al = alpha[k];
    float sc = 0.f, pa = 0.f;
    for (int c0 = 0; c0 < mid; c0 += 8) {
      float gv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) gv[j] = (c0 + j < mid) ? g_n[(k * mid + c0 + j) * VV + r] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        if (c < mid) {
          const float th = tanhf(dyn_D(d, X, Wl, edge_type, k, c, u, w));
          sc += gv[j];
          pa = fmaf(th, gv[j], pa);
          dd_n[(k * mid + c) * VV + r] = al * (1.f - th * th) * gv[j];
        }
      }
    }
    SC[i] = sc;
    pal0 += (k == 0) ? pa : 0.f;
    pal1 += (k == 1) ? pa : 0.f;
    pal2 += (k == 2) ? pa : 0.f;
  }
  __syncthreads();

  // dbeta partial, pA write-out
  float pbe0 = 0.f, pbe1 = 0.f, pbe2 = 0.f;
  for (int i = tid; i < KSUB * VV; i += NT) {
    const int k = i / VV;
    const float v = Sm[i] * SC[i];
    pbe0 += (k == 0) ? v : 0.f;
    pbe1 += (k == 1) ? v : 0.f;
    pbe2 += (k == 2) ? v : 0.f;
    pA[(size_t)n * pstride + i] = SC[i];
  }
  __syncthreads();
  // softmax backward per column (k,w): SC <- dG = Sm * (beta*SC - sum_u Sm*beta*SC)
  for (int j = tid; j < KSUB * V; j += NT) {
    const int k = j / V, w = j - k * V;
    const float bk = beta[k];
    float dot = 0.f;
    for (int u = 0; u < V; ++u) dot = fmaf(Sm[k * VV + u * V + w], bk * SC[k * VV + u * V + w], dot);
    for (int u = 0; u < V; ++u) {
      const int idx = k * VV + u * V + w;
      SC[idx] = Sm[idx] * (bk * SC[idx] - dot);
    }
  }
  // d(delta)[cc,u,w] = sum_c We[eps,c,cc] dD1[c,u,w]  -> second workspace (dD1 itself is kept for k_dynadj_dwe)
  {
    const float* dd1 = dd_n + (size_t)1 * mid * VV;
    float* dl = ddelta + (size_t)n * mid * VV;
    for (int r = tid; r < VV; r += NT) {
      const int e = edge_type[r];
      float dv[32], out[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) dv[c] = (c < mid) ? dd1[c * VV + r] : 0.f;
#pragma unroll
      for (int cc = 0; cc < 32; ++cc) {
        float acc = 0.f;
        if (cc < mid) {
#pragma unroll
          for (int c = 0; c < 32; ++c)
            if (c < mid) acc = fmaf(Wl[(e * mid + c) * mid + cc], dv[c], acc);
        }
        out[cc] = acc;
      }
#pragma unroll
      for (int cc = 0; cc < 32; ++cc)
        if (cc < mid) dl[cc * VV + r] = out[cc];
    }
  }
  __syncthreads();
  // row/col sums + Gram backward: thread = (k,c,j)
  for (int o = tid; o < KSUB * mid * V; o += NT) {
    const int k = o / (mid * V), rr = o - k * mid * V, c = rr / V, j = rr - c * V;
    const float* dk = (k == 1) ? ddelta + ((size_t)n * mid + c) * VV : dd_n + (size_t)(k * mid + c) * VV;
    float rs = 0.f, cs = 0.f;
    {
      float rv[32], cv[32];
#pragma unroll
      for (int w = 0; w < 32; ++w) {
        rv[w] = (w < V) ? dk[j * V + w] : 0.f;
        cv[w] = (w < V) ? dk[w * V + j] : 0.f;
      }
#pragma unroll
      for (int w = 0; w < 32; ++w) { rs += rv[w]; cs += cv[w]; }
    }
    const float* x1 = X + (slot_x1(k) * mid + c) * V;
    const float* x2 = X + (slot_x2(k) * mid + c) * V;
    const float* dG = SC + k * VV;
    float g1 = 0.f, g2 = 0.f;
    for (int w = 0; w < V; ++w) g1 = fmaf(dG[j * V + w], x2[w], g1);   // d x1_k[c,j]
    for (int u = 0; u < V; ++u) g2 = fmaf(dG[u * V + j], x1[u], g2);   // d x2_k[c,j]
    if (k < 2) {
      dX[(slot_x1(k) * mid + c) * V + j] = rs + g1;
      dX[(slot_x2(k) * mid + c) * V + j] = g2 - cs;
    } else {
      dX[(2 * mid + c) * V + j] = rs - cs + g1 + g2;
    }
  }
  __syncthreads();
  // dproj rows [a (2mid) | b (2mid) | s-typed (mid*P)]
  float* dp_n = dproj + (size_t)n * 9 * mid * V;
  for (int o = tid; o < 9 * mid * V; o += NT) {
    const int q = o / V, v = o - q * V;
    float val;
    if (q < 2 * mid) { const int slot = q / mid, c = q - slot * mid; val = dX[(slot * mid + c) * V + v]; }
    else if (q < 4 * mid) { const int r = q - 2 * mid; val = dX[((3 + r / mid) * mid + r % mid) * V + v]; }
    else {
      const int rr = q - 4 * mid, c = rr / d.P, p = rr - c * d.P;
      val = (node_type[v] == p) ? dX[(2 * mid + c) * V + v] : 0.f;
    }
    dp_n[o] = val;
  }
  // block-reduce the six scalar partials
  float vals[6] = {pal0, pal1, pal2, pbe0, pbe1, pbe2};
  const int wv = tid / DSGCN_WAVE, ln = tid % DSGCN_WAVE;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float r = wave_sum(vals[i]);
    if (ln == 0) red[i][wv] = r;
  }
  __syncthreads();
  if (tid < 6) {
    float r = 0.f;
    for (int i = 0; i < NT / DSGCN_WAVE; ++i) r += red[tid][i];
    pab[(size_t)n * pstride + tid] = r;
  }
}

// Edge-typed weight gradient: one workgroup per (sample, edge class e).
//   dWe[e,c,cc] += sum_{(u,w) in class e} dD1[n,c,u,w] * (a1[n,cc,u] - b1[n,cc,w]);   dbe[e,c] += sum dD1[n,c,u,w]
// The class's pairs (<= 256) are staged in LDS once (dD1 column and delta column per pair), then every (c,cc) output is a
// short LDS dot product; results are accumulated into dwe/dbe with float atomics (128 adders per address).
__global__ __launch_bounds__(NT_DWE) void k_dynadj_dwe(DynDims d, const float* __restrict__ proj,
                                                   const float* __restrict__ dd, const int* __restrict__ pair_order,
                                                   const int* __restrict__ class_start, float* __restrict__ dwe,
                                                   float* __restrict__ dbe) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n = blockIdx.x, e = blockIdx.y;
  const int mid = d.mid, V = d.V, VV = V * V;
  const int p0 = class_start[e], np = class_start[e + 1] - p0;
  if (np <= 0) return;
  const int NPs = np | 1;
  float* Dl = lds;                      // [mid][NPs]  dD1[c][pair]
  float* El = lds + mid * NPs;          // [mid][NPs]  delta[cc][pair]
  const float* dd1 = dd + ((size_t)n * KSUB + 1) * mid * VV;
  const float* a1 = proj + ((size_t)n * 9 * mid + 1 * mid) * V;           // rows mid..2mid-1  (a, k=1)
  const float* b1 = proj + ((size_t)n * 9 * mid + 3 * mid) * V;           // rows 3mid..4mid-1 (b, k=1)
  for (int i0 = threadIdx.x; i0 < mid * np; i0 += NT_DWE * 4) {
    float dv[4], ev[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q * NT_DWE;
      dv[q] = 0.f; ev[q] = 0.f;
      if (i < mid * np) {
        const int c = i / np, p = i - c * np;
        const int r = pair_order[p0 + p];
        const int u = r / V, w = r - u * V;
        dv[q] = dd1[c * VV + r];
        ev[q] = a1[c * V + u] - b1[c * V + w];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q * NT_DWE;
      if (i < mid * np) {
        const int c = i / np, p = i - c * np;
        Dl[c * NPs + p] = dv[q];
        El[c * NPs + p] = ev[q];
      }
    }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < mid * mid; o += NT_DWE) {
    const int c = o / mid, cc = o - c * mid;
    float acc = 0.f, accb = 0.f;
    for (int p = 0; p < np; ++p) {
      const float g = Dl[c * NPs + p];
      acc = fmaf(g, El[cc * NPs + p], acc);
      accb += g;
    }
    atomicAdd(dwe + (size_t)e * mid * mid + o, acc);
    if (cc == 0) atomicAdd(dbe + e * mid + c, accb);
  }
}

size_t dyn_lds_bytes(int mid, int V, int E, bool bwd) {
  size_t f = 5 * (size_t)mid * V + 3 * (size_t)V * V + 3 * (size_t)V * 2 + (size_t)E * mid * mid + (size_t)E * mid;
  if (bwd) f += 3 * (size_t)V * V + 5 * (size_t)mid * V;
  return f * sizeof(float);
}

}  // namespace

extern "C" {

int dsgcn_dynadj_fwd(const float* proj, const float* A, const float* alpha, const float* beta, const float* we,
                     const float* be, const int* node_type, const int* edge_type, float* ahat, int n, int mid, int V,
                     int P, int E, void* stream) {
  if (!proj || !A || !ahat || !we || !be || n <= 0 || mid <= 0) return DSGCN_EINVAL;
  if (V > 32 || mid > 32) return DSGCN_EUNSUPPORTED;
  const size_t lds = dyn_lds_bytes(mid, V, E, false);
  if (lds > 156 * 1024) return DSGCN_EUNSUPPORTED;
  DynDims d{n, 0, mid, V, P, E};
  static size_t attr_fwd = 64 * 1024;      // raised once per size class (not a stream op: keep it out of graph capture)
  if (lds > attr_fwd) {
    hipError_t e = hipFuncSetAttribute((const void*)k_dynadj_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_fwd = 156 * 1024;
  }
  hipLaunchKernelGGL(k_dynadj_fwd, dim3(n, n <= 128 ? 2 : 1), dim3(NT), lds, (hipStream_t)stream, d, proj, A, alpha, beta, we, be,
                     node_type, edge_type, ahat);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_dynadj_bwd(const float* proj, const float* alpha, const float* beta, const float* we, const float* be,
                     const int* node_type, const int* edge_type, const int* pair_order, const int* class_start,
                     const float* dahat, float* dd_ws, float* ddelta_ws, float* dproj, float* pA, float* pab,
                     int pstride, float* dwe, float* dbe, int n, int mid, int V, int P, int E, void* stream) {
  if (!proj || !dahat || !dd_ws || !ddelta_ws || !dproj || !pA || !pab || !dwe || !dbe) return DSGCN_EINVAL;
  if (V > 32 || mid > 32) return DSGCN_EUNSUPPORTED;
  const size_t lds = dyn_lds_bytes(mid, V, E, true);
  if (lds > 156 * 1024) return DSGCN_EUNSUPPORTED;
  DynDims d{n, 0, mid, V, P, E};
  hipStream_t st = (hipStream_t)stream;
  static size_t attr_bwd = 64 * 1024;
  if (lds > attr_bwd) {
    hipError_t e = hipFuncSetAttribute((const void*)k_dynadj_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_bwd = 156 * 1024;
  }
  hipLaunchKernelGGL(k_dynadj_bwd, dim3(n), dim3(NT), lds, st, d, proj, alpha, beta, we, be, node_type, edge_type,
                     pair_order, class_start, dahat, dd_ws, ddelta_ws, dproj, pA, pab, pstride);
  DSGCN_LAUNCH_CHECK();
  const size_t lds2 = (size_t)2 * mid * ((V * V) | 1) * sizeof(float);   // worst case: one class holds every pair
  static size_t attr_dwe = 64 * 1024;
  if (lds2 > attr_dwe) {
    hipError_t e2 = hipFuncSetAttribute((const void*)k_dynadj_dwe, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (e2 != hipSuccess) return (int)e2;
    attr_dwe = 156 * 1024;
  }
  hipLaunchKernelGGL(k_dynadj_dwe, dim3(n, E), dim3(NT_DWE), lds2, st, d, proj, dd_ws, pair_order, class_start, dwe, dbe);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
