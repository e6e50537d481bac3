// CTR-GCN channel-wise topology refinement (reference: pyskl/models/gcns/utils/gcn.py:634-666, 882-929), the pieces
// that are not a 1x1 conv (those run through K-C) or the aggregate (K-A'):
//
//   tanhdiff   : d[k,n,r,u,v] = tanh(x1[n,k,r,u] - x2[n,k,r,v])            gcn.py:653-655 (after the mean over T)
//                proj (n, 2*K*R, V): rows [k*R + r] = conv1_k(xbar), rows [K*R + k*R + r] = conv2_k(xbar)
//   ctr_affine : Ahat[n, k*Co + c, u, v] = alpha_k * S_k[n,c,u,v] + A[k,u,v] (+ beta_k * G[n,k,u,v])
//                gcn.py:657 (classic: one shared alpha, no G; S_k = conv4_k(d[k]) from K-C); gcn.py:755-760 (CTRHGC:
//                alpha per subset and the "ada" Gram term G_k = x1_k^T x2_k scaled by the subset's beta)
//   edge_select: out[n,r,u,v] = in[n, eps(u,v)*R + r, u, v]                 gcn.py:737-745 (CTRHGC edge attention: the
//                edge-typed conv computes all E variants; each joint pair keeps the one of its edge class)
//   plane_stats: per-(n,c) sum / sum of squares of a tensor (the BatchNorms that follow a temporal conv / max-pool in
//                MSTCN, msg3d_utils.py:101-117) -> dsgcn_bn_finalize
// All are HBM-bound streaming passes, one wave per V*V (or T*V) plane.
#include "common.h"
#include "dsgcn_jobs.h"

namespace {

constexpr int CTR_MAXK = 4;
#define DSGCN_CTR_SLICES 4      // channel slices per (n, k) in ctr_affine_bwd: prow has n*4 rows

struct CtrPtrs {
  const float* s[CTR_MAXK];
  float* ds[CTR_MAXK];
};

// one wave per (k, n, r)
// Rd = R: d (K, n, R, V, V).  Rd = R + 2 (the augmented form): d (K, n, R + 2, V, V) with two constant channels behind the
// R difference channels — A[k] and ones — so that alpha * conv4(d) + A[k] is ONE 1x1 conv over R + 2 channels
// (weights [W | 1 | b], input scale [alpha .. alpha, 1, alpha]): the refinement's affine pass and its backward disappear.
__global__ __launch_bounds__(64) void k_tanhdiff_fwd(const float* __restrict__ proj, float* __restrict__ d, int n, int K,
                                                     int R, int V, int Rd, const float* __restrict__ A) {
  __shared__ float xs[64];
  const int lane = threadIdx.x;
  const long row = blockIdx.x;                 // (k*n + i)*Rd + r
  const int r = (int)(row % Rd);
  const long ki = row / Rd;
  const int i = (int)(ki % n), k = (int)(ki / n);
  float* __restrict__ out = d + (size_t)row * V * V;
  if (r >= R) {
    const float* __restrict__ a = A + (size_t)k * V * V;
    for (int e = lane; e < V * V; e += 64) out[e] = r == R ? a[e] : 1.f;
    return;
  }
  const float* __restrict__ p1 = proj + ((size_t)i * 2 * K * R + (size_t)k * R + r) * V;
  const float* __restrict__ p2 = p1 + (size_t)K * R * V;
  if (lane < V) xs[lane] = p1[lane];
  else if (lane >= 32 && lane - 32 < V) xs[lane] = p2[lane - 32];
  wave_lds_sync();
  for (int e = lane; e < V * V; e += 64) {
    const int u = e / V, v = e - u * V;
    out[e] = tanhf(xs[u] - xs[32 + v]);
  }
}

// dproj rows: x1 gets sum_v dd*(1-d^2), x2 gets -sum_u dd*(1-d^2)
// (dd: one gradient tensor (n, R, V, V) per subset, NULL = none reached it)
// (Rd = R + 2, the augmented form: rows r == R carry the gradient of the A[k] channel = sum_c dAhat of sample i — copied
// to dAp (n, K, V, V), whose column sum over n is dA)
__global__ __launch_bounds__(64) void k_tanhdiff_bwd(const float* __restrict__ d, CtrPtrs dd,
                                                     float* __restrict__ dproj, int n, int K, int R, int V, int Rd,
                                                     float* __restrict__ dAp) {
  extern __shared__ float ts[];                // [V*V]
  const int lane = threadIdx.x;
  const int rows = Rd > R ? R + 1 : R;         // (the ones channel's gradient is not needed)
  const long row = blockIdx.x;
  const int r = (int)(row % rows);
  const long ki = row / rows;
  const int i = (int)(ki % n), k = (int)(ki / n);
  const float* __restrict__ pd = d + ((size_t)ki * Rd + r) * V * V;
  const float* __restrict__ pg = dd.s[k] ? dd.s[k] + ((size_t)i * Rd + r) * V * V : nullptr;
  if (r == R) {
    float* __restrict__ o = dAp + ((size_t)i * K + k) * V * V;
    for (int e = lane; e < V * V; e += 64) o[e] = pg ? pg[e] : 0.f;
    return;
  }
  for (int e = lane; e < V * V; e += 64) {
    const float t = pd[e];
    ts[e] = pg ? pg[e] * (1.f - t * t) : 0.f;
  }
  wave_lds_sync();
  float* __restrict__ o1 = dproj + ((size_t)i * 2 * K * R + (size_t)k * R + r) * V;
  float* __restrict__ o2 = o1 + (size_t)K * R * V;
  if (lane < V) {
    float acc = 0.f;
    for (int v = 0; v < V; ++v) acc += ts[lane * V + v];
    o1[lane] = acc;
  } else if (lane >= 32 && lane - 32 < V) {
    const int v = lane - 32;
    float acc = 0.f;
    for (int u = 0; u < V; ++u) acc += ts[u * V + v];
    o2[v] = -acc;
  }
}

// one wave per (n, k, c)
__global__ __launch_bounds__(64) void k_ctr_affine_fwd(CtrPtrs p, const float* __restrict__ alpha, int astride,
                                                       const float* __restrict__ A, const float* __restrict__ beta,
                                                       const float* __restrict__ G, float* __restrict__ ahat, int K,
                                                       int Co, int VV) {
  const int lane = threadIdx.x;
  const long unit = blockIdx.x;                // (n*K + k)*Co + c
  const int c = (int)(unit % Co);
  const long nk = unit / Co;
  const int k = (int)(nk % K);
  const long i = nk / K;
  const float al = alpha[k * astride];
  const float* __restrict__ s = p.s[k] + ((size_t)i * Co + c) * VV;
  const float* __restrict__ a = A + (size_t)k * VV;
  float* __restrict__ o = ahat + (size_t)unit * VV;
  if (G) {
    const float bt = beta[k];
    const float* __restrict__ g = G + (size_t)nk * VV;
    for (int e = lane; e < VV; e += 64) o[e] = fmaf(bt, g[e], fmaf(al, s[e], a[e]));
  } else {
    for (int e = lane; e < VV; e += 64) o[e] = fmaf(al, s[e], a[e]);
  }
}

// one block per (n, k, channel slice): dS_k = alpha*dAhat ; prow (n*CS, K*VV + K): [sum_c dAhat[k,e] | sum dAhat*S] over
// the slice's channels (row = n*CS + slice; the caller's column sum adds the slices and the samples)
__global__ __launch_bounds__(256) void k_ctr_affine_bwd(CtrPtrs p, const float* __restrict__ alpha, int astride,
                                                        const float* __restrict__ dahat, float* __restrict__ prow,
                                                        int K, int Co, int VV) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const long nk = blockIdx.x;
  const int k = (int)(nk % K);
  const long i = nk / K;
  const float al = alpha[k * astride];
  const float* __restrict__ g = dahat + (size_t)nk * Co * VV;
  const float* __restrict__ s = p.s[k] + (size_t)i * Co * VV;
  float* __restrict__ ds = p.ds[k] + (size_t)i * Co * VV;
  float accA[4] = {0.f, 0.f, 0.f, 0.f};
  float acca = 0.f;
  const int CS = gridDim.y, sl = blockIdx.y;
  const int cper = (Co + CS - 1) / CS;
  const int cbeg = sl * cper, cend = min(Co, cbeg + cper);
  for (int c = cbeg; c < cend; ++c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + 256 * j;
      if (e < VV) {
        const float gv = g[(size_t)c * VV + e];
        const float sv = s[(size_t)c * VV + e];
        ds[(size_t)c * VV + e] = al * gv;
        accA[j] += gv;
        acca = fmaf(gv, sv, acca);
      }
    }
  }
  float* __restrict__ out = prow + ((size_t)i * CS + sl) * (K * VV + K);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = tid + 256 * j;
    if (e < VV) out[(size_t)k * VV + e] = accA[j];
  }
  acca = wave_sum(acca);
  if ((tid & 63) == 0) red[tid >> 6] = acca;
  __syncthreads();
  if (tid == 0) out[(size_t)K * VV + k] = red[0] + red[1] + red[2] + red[3];
}

// edge-typed select: thread = (n, r, pair); in (n, E*R, VV), out (n, R, VV)
__global__ __launch_bounds__(256) void k_edge_select_fwd(const float* __restrict__ in, const int* __restrict__ et,
                                                         float* __restrict__ out, int R, int E, int VV, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int e = (int)(idx % VV);
  const long nr = idx / VV;
  const int r = (int)(nr % R);
  const long i = nr / R;
  out[idx] = in[((size_t)i * E * R + (size_t)et[e] * R + r) * VV + e];
}

// backward: din (n, E*R, VV) = dout at the selected class, zero elsewhere (every element is written)
__global__ __launch_bounds__(256) void k_edge_select_bwd(const float* __restrict__ dout, const int* __restrict__ et,
                                                         float* __restrict__ din, int R, int E, int VV, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // over (n, E*R, VV)
  if (idx >= total) return;
  const int e = (int)(idx % VV);
  const long nc = idx / VV;
  const int ch = (int)(nc % (E * R));
  const long i = nc / (E * R);
  const int cls = ch / R, r = ch - cls * R;
  din[idx] = (et[e] == cls) ? dout[((size_t)i * R + r) * VV + e] : 0.f;
}

// one wave per plane of L elements: partial[plane] = [sum, sum of squares]
__global__ __launch_bounds__(64) void k_plane_stats(const float* __restrict__ x, float* __restrict__ partial, int L,
                                                    int vec) {
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const float* __restrict__ p = x + (size_t)plane * L;
  double s = 0.0, q = 0.0;
  if (vec) {
    const f32x4* __restrict__ p4 = reinterpret_cast<const f32x4*>(p);
    const int L4 = L >> 2;
#pragma unroll 4
    for (int i = lane; i < L4; i += 64) {
      const f32x4 v = p4[i];
      s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
      q = fma((double)v.x, (double)v.x, q); q = fma((double)v.y, (double)v.y, q);
      q = fma((double)v.z, (double)v.z, q); q = fma((double)v.w, (double)v.w, q);
    }
  } else {
#pragma unroll 4
    for (int i = lane; i < L; i += 64) {
      const double v = (double)p[i];
      s += v;
      q = fma(v, v, q);
    }
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  if (lane == 0) {
    partial[plane * 2 + 0] = (float)s;
    partial[plane * 2 + 1] = (float)q;
  }
}

// The augmented conv4 operands of a CTR-GCN unit (see k_tanhdiff_fwd), as JOBS (include/dsgcn_jobs.h: one record per
// unit, blockIdx.z = record — every unit of a model in one launch at the head of the step, they depend on parameters
// only): block k of a record builds W'_k (Co, R + 2) = [W_k | 1 | b_k] and its input affine (scale [alpha x R, 1, alpha],
// shift 0) at sh + k*2*(R+2).
struct CtrPrepTable {
  dsgcn_ctr_prep_job j[DSGCN_CTR_JOBS_MAX];
};
struct CtrFinTable {
  dsgcn_ctr_fin_job j[DSGCN_CTR_JOBS_MAX];
};

__global__ __launch_bounds__(256) void k_ctr_wprep(CtrPrepTable t) {
  const dsgcn_ctr_prep_job& a = t.j[blockIdx.z];
  const int k = blockIdx.x, R = a.R, Co = a.Co, R2 = R + 2;
  if (k >= a.K) return;
  const float* __restrict__ wk = a.w[k];
  const float* __restrict__ bk = a.b[k];
  float* __restrict__ o = a.wout + (size_t)k * Co * R2;
  const int e = blockIdx.y * 256 + threadIdx.x;                    // (grid.y covers Co * (R + 2): one element per thread)
  if (e < Co * R2) {
    const int c = e / R2, r = e - c * R2;
    o[e] = r < R ? wk[(size_t)c * R + r] : (r == R ? 1.f : (bk ? bk[c] : 0.f));
  }
  if (blockIdx.y == 0) {
    const float al = a.alpha[0];
    for (int r = threadIdx.x; r < R2; r += 256) {
      a.sh[(size_t)k * 2 * R2 + r] = r == R ? 1.f : al;
      a.sh[(size_t)k * 2 * R2 + R2 + r] = 0.f;
    }
  }
}

// Its backward: dW_k = dW'_k[:, :R], db_k = dW'_k[:, R + 1] (the input scale already put alpha on those columns), and
// block K: dalpha = sum_k (sum_{r<R} ds_k[r] + ds_k[R + 1]) from the input-scale gradients (fixed order).
__global__ __launch_bounds__(256) void k_ctr_wfin(CtrFinTable t) {
  const dsgcn_ctr_fin_job& a = t.j[blockIdx.z];
  const int K = a.K, Co = a.Co, R = a.R, dss = a.ds_stride, R2 = R + 2;
  if ((int)blockIdx.x > K) return;
  if ((int)blockIdx.x == K) {
    if (threadIdx.x == 0 && blockIdx.y == 0) {
      float s = 0.f;
      for (int k = 0; k < K; ++k)
        if (a.ds[k]) {
          for (int r = 0; r < R; ++r) s += a.ds[k][(size_t)r * dss];
          s += a.ds[k][(size_t)(R + 1) * dss];
        }
      a.dalpha[0] = s;
    }
    return;
  }
  const int k = blockIdx.x;
  const float* __restrict__ g = a.dwp[k];
  float* __restrict__ dw = a.out[k];
  float* __restrict__ db = a.out[k] + (size_t)Co * R;
  const int e = blockIdx.y * 256 + threadIdx.x;                    // (grid.y covers Co * R)
  if (e < Co * R) {
    const int c = e / R, r = e - c * R;
    dw[e] = g ? g[(size_t)c * R2 + r] : 0.f;
  }
  if (e < Co) db[e] = g ? g[(size_t)e * R2 + R + 1] : 0.f;
}

}  // namespace

extern "C" {

int dsgcn_tanhdiff_fwd(const float* proj, float* d, int n, int K, int R, int V, void* stream) {
  if (!proj || !d || n <= 0 || K <= 0 || R <= 0 || V <= 0 || V > 32) return DSGCN_EINVAL;
  hipLaunchKernelGGL(k_tanhdiff_fwd, dim3((unsigned)((long)K * n * R)), dim3(64), 0, (hipStream_t)stream, proj, d, n, K,
                     R, V, R, (const float*)nullptr);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// d (K, n, R + 2, V, V): the R tanh-difference channels, then A[k] (A (K, V, V)) and a channel of ones.
int dsgcn_tanhdiff_aug_fwd(const float* proj, const float* A, float* d, int n, int K, int R, int V, void* stream) {
  if (!proj || !A || !d || n <= 0 || K <= 0 || R <= 0 || V <= 0 || V > 32) return DSGCN_EINVAL;
  hipLaunchKernelGGL(k_tanhdiff_fwd, dim3((unsigned)((long)K * n * (R + 2))), dim3(64), 0, (hipStream_t)stream, proj, d, n,
                     K, R, V, R + 2, A);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// dd: K gradients (n, R + 2, V, V) (NULL = zero); dproj as dsgcn_tanhdiff_bwd; dAp (n, K, V, V) = the A-channel rows.
int dsgcn_tanhdiff_aug_bwd(const float* d, const float* const* dd, float* dproj, float* dAp, int n, int K, int R, int V,
                           void* stream) {
  if (!d || !dd || !dproj || !dAp || n <= 0 || K <= 0 || K > CTR_MAXK || R <= 0 || V <= 0 || V > 32) return DSGCN_EINVAL;
  CtrPtrs p = {};
  for (int k = 0; k < K; ++k) p.s[k] = dd[k];
  hipLaunchKernelGGL(k_tanhdiff_bwd, dim3((unsigned)((long)K * n * (R + 1))), dim3(64), (size_t)V * V * sizeof(float),
                     (hipStream_t)stream, d, p, dproj, n, K, R, V, R + 2, dAp);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// The augmented conv4 operands of njobs units in one launch (records: include/dsgcn_jobs.h).
int dsgcn_ctr_wprep_multi(const dsgcn_ctr_prep_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs <= 0) return DSGCN_EINVAL;
  for (int i = 0; i < njobs; ++i) {
    const dsgcn_ctr_prep_job& a = jobs[i];
    if (!a.alpha || !a.wout || !a.sh || a.K <= 0 || a.K > CTR_MAXK || a.Co <= 0 || a.R <= 0) return DSGCN_EINVAL;
    for (int k = 0; k < a.K; ++k)
      if (!a.w[k]) return DSGCN_EINVAL;
  }
  for (int i0 = 0; i0 < njobs; i0 += DSGCN_CTR_JOBS_MAX) {
    const int m = njobs - i0 < DSGCN_CTR_JOBS_MAX ? njobs - i0 : DSGCN_CTR_JOBS_MAX;
    CtrPrepTable t = {};
    int kmax = 0, ymax = 0;
    for (int i = 0; i < m; ++i) {
      t.j[i] = jobs[i0 + i];
      const int y = (t.j[i].Co * (t.j[i].R + 2) + 255) / 256;
      kmax = t.j[i].K > kmax ? t.j[i].K : kmax;
      ymax = y > ymax ? y : ymax;
    }
    hipLaunchKernelGGL(k_ctr_wprep, dim3((unsigned)kmax, (unsigned)ymax, (unsigned)m), dim3(256), 0, (hipStream_t)stream,
                       t);
    DSGCN_LAUNCH_CHECK();
  }
  return 0;
}

// w / b: K device pointers (Co, R) / (Co) or NULL; wout (K, Co, R + 2); sh (K, 2, R + 2) = per subset [scale; shift].
int dsgcn_ctr_wprep(const float* const* w, const float* const* b, const float* alpha, float* wout, float* sh, int K,
                    int Co, int R, void* stream) {
  if (!w || !b || K <= 0 || K > CTR_MAXK) return DSGCN_EINVAL;
  dsgcn_ctr_prep_job j = {};
  for (int k = 0; k < K; ++k) {
    j.w[k] = w[k];
    j.b[k] = b[k];
  }
  j.alpha = alpha, j.wout = wout, j.sh = sh, j.K = K, j.Co = Co, j.R = R;
  return dsgcn_ctr_wprep_multi(&j, 1, stream);
}

// The finishing launches of njobs units as one (records: include/dsgcn_jobs.h).
int dsgcn_ctr_wfin_multi(const dsgcn_ctr_fin_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs <= 0) return DSGCN_EINVAL;
  for (int i = 0; i < njobs; ++i) {
    const dsgcn_ctr_fin_job& a = jobs[i];
    if (!a.dalpha || a.K <= 0 || a.K > CTR_MAXK || a.Co <= 0 || a.R <= 0 || a.ds_stride <= 0) return DSGCN_EINVAL;
    for (int k = 0; k < a.K; ++k)
      if (!a.out[k]) return DSGCN_EINVAL;
  }
  for (int i0 = 0; i0 < njobs; i0 += DSGCN_CTR_JOBS_MAX) {
    const int m = njobs - i0 < DSGCN_CTR_JOBS_MAX ? njobs - i0 : DSGCN_CTR_JOBS_MAX;
    CtrFinTable t = {};
    int kmax = 0, ymax = 0;
    for (int i = 0; i < m; ++i) {
      t.j[i] = jobs[i0 + i];
      const int y = (t.j[i].Co * t.j[i].R + 255) / 256;
      kmax = t.j[i].K > kmax ? t.j[i].K : kmax;
      ymax = y > ymax ? y : ymax;
    }
    hipLaunchKernelGGL(k_ctr_wfin, dim3((unsigned)(kmax + 1), (unsigned)ymax, (unsigned)m), dim3(256), 0,
                       (hipStream_t)stream, t);
    DSGCN_LAUNCH_CHECK();
  }
  return 0;
}

// dwp: K gradients of W'_k (Co, R + 2) or NULL; ds: K gradients of the input scales (R + 2 elements, ds_stride floats
// apart: 3 when they are read straight out of the K-C backward's summed (channel, 3) rows) or NULL;
// out: K buffers (Co*R + Co) = [dW_k | db_k]; dalpha (1).
int dsgcn_ctr_wfin(const float* const* dwp, const float* const* ds, int ds_stride, float* const* out, float* dalpha, int K,
                   int Co, int R, void* stream) {
  if (!dwp || !ds || !out || K <= 0 || K > CTR_MAXK) return DSGCN_EINVAL;
  dsgcn_ctr_fin_job j = {};
  for (int k = 0; k < K; ++k) {
    j.dwp[k] = dwp[k];
    j.ds[k] = ds[k];
    j.out[k] = out[k];
  }
  j.dalpha = dalpha, j.K = K, j.Co = Co, j.R = R, j.ds_stride = ds_stride;
  return dsgcn_ctr_wfin_multi(&j, 1, stream);
}

int dsgcn_tanhdiff_bwd(const float* d, const float* dd, float* dproj, int n, int K, int R, int V, void* stream) {
  if (!d || !dd || !dproj || n <= 0 || K <= 0 || K > CTR_MAXK || R <= 0 || V <= 0 || V > 32) return DSGCN_EINVAL;
  CtrPtrs p = {};
  for (int k = 0; k < K; ++k) p.s[k] = dd + (size_t)k * n * R * V * V;
  hipLaunchKernelGGL(k_tanhdiff_bwd, dim3((unsigned)((long)K * n * R)), dim3(64), (size_t)V * V * sizeof(float),
                     (hipStream_t)stream, d, p, dproj, n, K, R, V, R, (float*)nullptr);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// dd: K pointers, each the gradient (n, R, V, V) of one subset's slice of d or NULL (no gradient reached that slice): the
// K conv4 backward passes hand their results over as they are (stacking them was a fill + copy + add per subset)
int dsgcn_tanhdiff_bwd_k(const float* d, const float* const* dd, float* dproj, int n, int K, int R, int V, void* stream) {
  if (!d || !dd || !dproj || n <= 0 || K <= 0 || K > CTR_MAXK || R <= 0 || V <= 0 || V > 32) return DSGCN_EINVAL;
  CtrPtrs p = {};
  for (int k = 0; k < K; ++k) p.s[k] = dd[k];
  hipLaunchKernelGGL(k_tanhdiff_bwd, dim3((unsigned)((long)K * n * R)), dim3(64), (size_t)V * V * sizeof(float),
                     (hipStream_t)stream, d, p, dproj, n, K, R, V, R, (float*)nullptr);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_ctr_affine_fwd(const float* const* s, const float* alpha, int alpha_stride, const float* A, const float* beta,
                         const float* G, float* ahat, int n, int K, int Co, int V, void* stream) {
  if (!s || !alpha || !A || !ahat || n <= 0 || K <= 0 || K > CTR_MAXK || Co <= 0 || V <= 0 || (G && !beta) ||
      (alpha_stride != 0 && alpha_stride != 1))
    return DSGCN_EINVAL;
  CtrPtrs p = {};
  for (int k = 0; k < K; ++k) {
    if (!s[k]) return DSGCN_EINVAL;
    p.s[k] = s[k];
  }
  hipLaunchKernelGGL(k_ctr_affine_fwd, dim3((unsigned)((long)n * K * Co)), dim3(64), 0, (hipStream_t)stream, p, alpha,
                     alpha_stride, A, beta, G, ahat, K, Co, V * V);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_ctr_affine_bwd(const float* const* s, const float* alpha, int alpha_stride, const float* dahat,
                         float* const* ds, float* prow, int n, int K, int Co, int V, void* stream) {
  if (!s || !alpha || !dahat || !ds || !prow || n <= 0 || K <= 0 || K > CTR_MAXK || Co <= 0 || V <= 0 || V * V > 1024 ||
      (alpha_stride != 0 && alpha_stride != 1))
    return DSGCN_EINVAL;
  CtrPtrs p = {};
  for (int k = 0; k < K; ++k) {
    if (!s[k] || !ds[k]) return DSGCN_EINVAL;
    p.s[k] = s[k];
    p.ds[k] = ds[k];
  }
  hipLaunchKernelGGL(k_ctr_affine_bwd, dim3((unsigned)((long)n * K), (unsigned)DSGCN_CTR_SLICES), dim3(256), 0,
                     (hipStream_t)stream, p, alpha, alpha_stride, dahat, prow, K, Co, V * V);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_edge_select_fwd(const float* in, const int* edge_type, float* out, int n, int R, int E, int V, void* stream) {
  if (!in || !edge_type || !out || n <= 0 || R <= 0 || E <= 0 || V <= 0) return DSGCN_EINVAL;
  const long total = (long)n * R * V * V;
  hipLaunchKernelGGL(k_edge_select_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                     edge_type, out, R, E, V * V, total);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_edge_select_bwd(const float* dout, const int* edge_type, float* din, int n, int R, int E, int V, void* stream) {
  if (!dout || !edge_type || !din || n <= 0 || R <= 0 || E <= 0 || V <= 0) return DSGCN_EINVAL;
  const long total = (long)n * E * R * V * V;
  hipLaunchKernelGGL(k_edge_select_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout,
                     edge_type, din, R, E, V * V, total);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_plane_stats(const float* x, float* partial, long planes, int L, void* stream) {
  if (!x || !partial || planes <= 0 || L <= 0) return DSGCN_EINVAL;
  hipLaunchKernelGGL(k_plane_stats, dim3((unsigned)planes), dim3(64), 0, (hipStream_t)stream, x, partial, L,
                     (L % 4 == 0) ? 1 : 0);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
