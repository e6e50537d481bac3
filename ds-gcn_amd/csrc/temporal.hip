// K-D (part): the pre- and post-stages of the multi-scale temporal unit dgmstcn (reference:
// pyskl/models/gcns/utils/tcn.py:379-428).
//
//   branch_act : h[n,c,t,0..V) = act_c(z*scale[c]+shift[c]),  h[n,c,t,V] = act_c(zaug*scale[c]+shift[c])
//                (act_c = ReLU for the BN'd branches c < n_act, identity for the '1x1' branch) — replaces
//                cat([x, x.mean(-1)]) (tcn.py:409) + the five BatchNorm2d+ReLU (tcn.py:389,394) in one pass.
//   combine    : f[n,c,t,v] = o[n,c,t,v] + o[n,c,t,V]*add_coeff[v]  (tcn.py:416-420) + per-plane sum / sum of squares of f
//                (the statistics of transform.0's BatchNorm, tcn.py:401) — replaces slice + einsum + add + the BN pass.
// One wave per (n,c) plane, coalesced streaming; both are HBM-bound elementwise passes.  The temporal convolutions and
// the max-pool between the two are csrc/tapconv.hip.  With zaug == NULL branch_act has no global-joint column (h is
// (n,C,T,V)): the form CTR-GCN's MSTCN uses (msg3d_utils.py:84-117).
#include "common.h"

namespace {

__global__ __launch_bounds__(64) void k_branch_act_fwd(const float* __restrict__ z, const float* __restrict__ zaug,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int n_act, float* __restrict__ h, int C, int T, int V) {
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int c = (int)(plane % C);
  const float s = scale[c], b = shift[c];
  const bool relu = c < n_act;
  const int V1 = zaug ? V + 1 : V, Lo = T * V1;
  const float* __restrict__ pz = z + (size_t)plane * T * V;
  const float* __restrict__ pa = zaug ? zaug + (size_t)plane * T : nullptr;
  float* __restrict__ ph = h + (size_t)plane * Lo;
#pragma unroll 4
  for (int o = lane; o < Lo; o += 64) {
    const int t = o / V1, v = o - t * V1;
    const float x = v < V ? pz[t * V + v] : pa[t];
    float y = fmaf(x, s, b);
    if (relu) y = fmaxf(y, 0.f);
    ph[o] = y;
  }
}

// part (n*C, 2): [sum dpre*x, sum dpre]
__global__ __launch_bounds__(64) void k_branch_act_bwd(const float* __restrict__ z, const float* __restrict__ zaug,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int n_act, const float* __restrict__ dh, float* __restrict__ dz,
                                                       float* __restrict__ dzaug, float* __restrict__ part, int C, int T,
                                                       int V) {
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int c = (int)(plane % C);
  const float s = scale[c], b = shift[c];
  const bool relu = c < n_act;
  const int V1 = zaug ? V + 1 : V, Lo = T * V1;
  const float* __restrict__ pz = z + (size_t)plane * T * V;
  const float* __restrict__ pa = zaug ? zaug + (size_t)plane * T : nullptr;
  const float* __restrict__ pg = dh + (size_t)plane * Lo;
  float* __restrict__ oz = dz + (size_t)plane * T * V;
  float* __restrict__ oa = zaug ? dzaug + (size_t)plane * T : nullptr;
  float u0 = 0.f, u1 = 0.f;
#pragma unroll 4
  for (int o = lane; o < Lo; o += 64) {
    const int t = o / V1, v = o - t * V1;
    const float x = v < V ? pz[t * V + v] : pa[t];
    float g = pg[o];
    if (relu && !(fmaf(x, s, b) > 0.f)) g = 0.f;
    if (v < V) oz[t * V + v] = g * s; else oa[t] = g * s;
    u0 = fmaf(g, x, u0);
    u1 += g;
  }
  u0 = wave_sum(u0);
  u1 = wave_sum(u1);
  if (lane == 0) {
    part[(size_t)plane * 2 + 0] = u0;
    part[(size_t)plane * 2 + 1] = u1;
  }
}

// partial (n*C... laid out [n][C][2]) : per-plane sum / sum of squares of f
__global__ __launch_bounds__(64) void k_tms_combine_fwd(const float* __restrict__ o, const float* __restrict__ coeff,
                                                        float* __restrict__ f, float* __restrict__ partial, int C, int T,
                                                        int V) {
  __shared__ float cf[32];
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  if (lane < V) cf[lane] = coeff[lane];
  wave_lds_sync();
  const int V1 = V + 1, L = T * V;
  const float* __restrict__ po = o + (size_t)plane * T * V1;
  float* __restrict__ pf = f + (size_t)plane * L;
  double sv = 0.0, qv = 0.0;               // fp64: the variance formula downstream (E[f^2]-mean^2) amplifies sum errors
#pragma unroll 4
  for (int i = lane; i < L; i += 64) {
    const int t = i / V, v = i - t * V;
    const float val = fmaf(po[t * V1 + V], cf[v], po[t * V1 + v]);
    pf[i] = val;
    sv += (double)val;
    qv = fma((double)val, (double)val, qv);
  }
  if (partial) {
    sv = wave_sum_d(sv);
    qv = wave_sum_d(qv);
    if (lane == 0) {
      partial[(size_t)plane * 2 + 0] = (float)sv;
      partial[(size_t)plane * 2 + 1] = (float)qv;
    }
  }
}

// gf_eff = gf + A0[c] + B0[c]*f ;  do[..,v<V] = gf_eff ; do[..,V] = sum_v gf_eff*coeff[v] ;
// pcoef (n*C, V): per-plane sum_t gf_eff[t,v]*o[t,V]
__global__ __launch_bounds__(64) void k_tms_combine_bwd(const float* __restrict__ o, const float* __restrict__ coeff,
                                                        const float* __restrict__ gf, const float* __restrict__ A0,
                                                        const float* __restrict__ B0, float* __restrict__ dout,
                                                        float* __restrict__ pcoef, int C, int T, int V) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [T][V] gf_eff
  __shared__ float cf[32];
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int c = (int)(plane % C);
  if (lane < V) cf[lane] = coeff[lane];
  const float a0 = A0 ? A0[c] : 0.f, b0 = B0 ? B0[c] : 0.f;
  wave_lds_sync();
  const int V1 = V + 1, L = T * V;
  const float* __restrict__ po = o + (size_t)plane * T * V1;
  const float* __restrict__ pg = gf ? gf + (size_t)plane * L : nullptr;
  float* __restrict__ pd = dout + (size_t)plane * T * V1;
#pragma unroll 4
  for (int i = lane; i < L; i += 64) {
    const int t = i / V, v = i - t * V;
    const float fv = fmaf(po[t * V1 + V], cf[v], po[t * V1 + v]);
    const float g = (pg ? pg[i] : 0.f) + fmaf(b0, fv, a0);
    lds[i] = g;
    pd[t * V1 + v] = g;
  }
  wave_lds_sync();
  for (int t = lane; t < T; t += 64) {
    float acc = 0.f;
    for (int v = 0; v < V; ++v) acc = fmaf(lds[t * V + v], cf[v], acc);
    pd[t * V1 + V] = acc;
  }
  if (lane < V) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc = fmaf(lds[t * V + lane], po[t * V1 + V], acc);
    pcoef[(size_t)plane * V + lane] = acc;
  }
}


// ---- 16-byte forms (planes with T*V % 4 == 0 and T*(V+1) % 4 == 0) --------------------------------------------
// The operand in the "other" row layout (V+1 columns next to V columns) is brought into LDS with aligned 16-B loads —
// the whole plane in flight at once — and read back element-wise; the operand in the iteration layout moves
// directly as float4.  Measured per launch at n = 128 (tools/ew_bench.py, cache-resident): branch_act_bwd 29.5 -> 25.7 us,
// combine forward 21.8 -> 18.9 us, combine backward 57.7 -> 38.7 us (its per-frame / per-joint sums no longer run on
// 25 lanes); a 16-B branch_act forward and fuse_out backward measured no faster than the scalar forms and are not kept.

__global__ __launch_bounds__(64) void k_branch_act_bwd4(const float* __restrict__ z, const float* __restrict__ zaug,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        int n_act, const float* __restrict__ dh, float* __restrict__ dz,
                                                        float* __restrict__ dzaug, float* __restrict__ part, int C, int T,
                                                        int V) {
  extern __shared__ __attribute__((aligned(16))) float lds[];       // [T*(V+1)] dh plane
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int c = (int)(plane % C);
  const float s = scale[c], b = shift[c];
  const bool relu = c < n_act;
  const int L = T * V, V1 = V + 1, L4 = L >> 2;
  const f32x4* __restrict__ qz = reinterpret_cast<const f32x4*>(z + (size_t)plane * L);
  f32x4* __restrict__ qo = reinterpret_cast<f32x4*>(dz + (size_t)plane * L);
  plane_to_lds(dh + (size_t)plane * T * V1, lds, (T * V1) >> 2, lane);
  wave_lds_sync();
  const float invV = 1.f / (float)V;
  float u0 = 0.f, u1 = 0.f;
#pragma unroll 2
  for (int i = lane; i < L4; i += 64) {
    const f32x4 xv = qz[i];
    const float x[4] = {xv.x, xv.y, xv.z, xv.w};
    int t, v;
    divmod_small(4 * i, V, invV, t, v);
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float g = lds[4 * i + k + t];
      if (relu && !(fmaf(x[k], s, b) > 0.f)) g = 0.f;
      r[k] = g * s;
      u0 = fmaf(g, x[k], u0);
      u1 += g;
      if (++v == V) { v = 0; ++t; }
    }
    qo[i] = f32x4{r[0], r[1], r[2], r[3]};
  }
  for (int t = lane; t < T; t += 64) {
    const float x = zaug[(size_t)plane * T + t];
    float g = lds[t * V1 + V];
    if (relu && !(fmaf(x, s, b) > 0.f)) g = 0.f;
    dzaug[(size_t)plane * T + t] = g * s;
    u0 = fmaf(g, x, u0);
    u1 += g;
  }
  u0 = wave_sum(u0);
  u1 = wave_sum(u1);
  if (lane == 0) {
    part[(size_t)plane * 2 + 0] = u0;
    part[(size_t)plane * 2 + 1] = u1;
  }
}

__global__ __launch_bounds__(64) void k_tms_combine_fwd4(const float* __restrict__ o, const float* __restrict__ coeff,
                                                         float* __restrict__ f, float* __restrict__ partial, int C, int T,
                                                         int V) {
  extern __shared__ __attribute__((aligned(16))) float lds[];       // [T*(V+1)] o plane, [32] coeff
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int V1 = V + 1, L = T * V, Lo = T * V1, L4 = L >> 2;
  float* cf = lds + Lo;
  if (lane < V) cf[lane] = coeff[lane];
  plane_to_lds(o + (size_t)plane * Lo, lds, Lo >> 2, lane);
  wave_lds_sync();
  const float invV = 1.f / (float)V;
  f32x4* __restrict__ qf = reinterpret_cast<f32x4*>(f + (size_t)plane * L);
  double sv = 0.0, qv = 0.0;
#pragma unroll 2
  for (int i = lane; i < L4; i += 64) {
    int t, v;
    divmod_small(4 * i, V, invV, t, v);
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      r[k] = fmaf(lds[t * V1 + V], cf[v], lds[4 * i + k + t]);
      if (++v == V) { v = 0; ++t; }
    }
    qf[i] = f32x4{r[0], r[1], r[2], r[3]};
    const float s4 = (r[0] + r[1]) + (r[2] + r[3]);
    const float q4 = fmaf(r[0], r[0], r[1] * r[1]) + fmaf(r[2], r[2], r[3] * r[3]);
    sv += (double)s4;
    qv += (double)q4;
  }
  if (partial) {
    sv = wave_sum_d(sv);
    qv = wave_sum_d(qv);
    if (lane == 0) {
      partial[(size_t)plane * 2 + 0] = (float)sv;
      partial[(size_t)plane * 2 + 1] = (float)qv;
    }
  }
}

__global__ __launch_bounds__(64) void k_tms_combine_bwd4(const float* __restrict__ o, const float* __restrict__ coeff,
                                                         const float* __restrict__ gf, const float* __restrict__ A0,
                                                         const float* __restrict__ B0, float* __restrict__ dout,
                                                         float* __restrict__ pcoef, int C, int T, int V) {
  extern __shared__ __attribute__((aligned(16))) float lds[];       // [T*(V+1)] o, [T*V] gf_eff, [T] aug, [32] coeff
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int c = (int)(plane % C);
  const int V1 = V + 1, L = T * V, Lo = T * V1, L4 = L >> 2;
  float* ge = lds + Lo;
  float* aug = ge + L;
  float* cf = aug + ((T + 3) & ~3);
  if (lane < V) cf[lane] = coeff[lane];
  const float a0 = A0 ? A0[c] : 0.f, b0 = B0 ? B0[c] : 0.f;
  plane_to_lds(o + (size_t)plane * Lo, lds, Lo >> 2, lane);
  wave_lds_sync();
  const float invV = 1.f / (float)V;
  const f32x4* __restrict__ qg = gf ? reinterpret_cast<const f32x4*>(gf + (size_t)plane * L) : nullptr;
  f32x4* ge4 = reinterpret_cast<f32x4*>(ge);
#pragma unroll 2
  for (int i = lane; i < L4; i += 64) {
    const f32x4 gv = qg ? qg[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    const float g[4] = {gv.x, gv.y, gv.z, gv.w};
    int t, v;
    divmod_small(4 * i, V, invV, t, v);
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float fv = fmaf(lds[t * V1 + V], cf[v], lds[4 * i + k + t]);
      r[k] = g[k] + fmaf(b0, fv, a0);
      if (++v == V) { v = 0; ++t; }
    }
    ge4[i] = f32x4{r[0], r[1], r[2], r[3]};
  }
  wave_lds_sync();
  for (int t = lane; t < T; t += 64) {
    float acc = 0.f;
    for (int v = 0; v < V; ++v) acc = fmaf(ge[t * V + v], cf[v], acc);
    aug[t] = acc;
  }
  {
    // pcoef[v] = sum_t gf_eff[t,v] * o[t,V]: the two half-waves take the even / odd frames
    const int v = lane & 31, half = lane >> 5;
    float acc = 0.f;
    if (v < V)
      for (int t = half; t < T; t += 2) acc = fmaf(ge[t * V + v], lds[t * V1 + V], acc);
    acc += __shfl_xor(acc, 32, 64);
    if (lane < V) pcoef[(size_t)plane * V + lane] = acc;
  }
  wave_lds_sync();
  const float invV1 = 1.f / (float)V1;
  f32x4* __restrict__ qd = reinterpret_cast<f32x4*>(dout + (size_t)plane * Lo);
  const int Lo4 = Lo >> 2;
#pragma unroll 2
  for (int i = lane; i < Lo4; i += 64) {
    int t, v;
    divmod_small(4 * i, V1, invV1, t, v);
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      r[k] = v < V ? ge[4 * i + k - t] : aug[t];
      if (++v == V1) { v = 0; ++t; }
    }
    qd[i] = f32x4{r[0], r[1], r[2], r[3]};
  }
}

}  // namespace

extern "C" {

int dsgcn_branch_act_fwd(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                         float* h, int n, int C, int T, int V, void* stream) {
  if (!z || !scale || !shift || !h || n <= 0 || C <= 0 || T <= 0 || V <= 0) return DSGCN_EINVAL;
  hipLaunchKernelGGL(k_branch_act_fwd, dim3((unsigned)((long)n * C)), dim3(64), 0, (hipStream_t)stream, z, zaug, scale,
                     shift, n_act, h, C, T, V);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_branch_act_bwd(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                         const float* dh, float* dz, float* dzaug, float* part, int n, int C, int T, int V,
                         void* stream) {
  if (!z || !scale || !shift || !dh || !dz || (zaug && !dzaug) || !part) return DSGCN_EINVAL;
  if (zaug && (T * V) % 4 == 0 && (T * (V + 1)) % 4 == 0 && (size_t)T * (V + 1) * 4 <= 60 * 1024) {
    hipLaunchKernelGGL(k_branch_act_bwd4, dim3((unsigned)((long)n * C)), dim3(64), (size_t)T * (V + 1) * 4,
                       (hipStream_t)stream, z, zaug, scale, shift, n_act, dh, dz, dzaug, part, C, T, V);
    DSGCN_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_branch_act_bwd, dim3((unsigned)((long)n * C)), dim3(64), 0, (hipStream_t)stream, z, zaug, scale,
                     shift, n_act, dh, dz, dzaug, part, C, T, V);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_tms_combine_fwd(const float* o, const float* coeff, float* f, float* partial, int n, int C, int T, int V,
                          void* stream) {
  if (!o || !coeff || !f || n <= 0 || C <= 0 || T <= 0 || V <= 0 || V > 32) return DSGCN_EINVAL;
  if ((T * V) % 4 == 0 && (T * (V + 1)) % 4 == 0 && (size_t)(T * (V + 1) + 32) * 4 <= 60 * 1024) {
    hipLaunchKernelGGL(k_tms_combine_fwd4, dim3((unsigned)((long)n * C)), dim3(64), (size_t)(T * (V + 1) + 32) * 4,
                       (hipStream_t)stream, o, coeff, f, partial, C, T, V);
    DSGCN_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_tms_combine_fwd, dim3((unsigned)((long)n * C)), dim3(64), 0, (hipStream_t)stream, o, coeff, f,
                     partial, C, T, V);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_tms_combine_bwd(const float* o, const float* coeff, const float* gf, const float* A0, const float* B0,
                          float* dout, float* pcoef, int n, int C, int T, int V, void* stream) {
  if (!o || !coeff || !dout || !pcoef || n <= 0 || C <= 0 || T <= 0 || V <= 0 || V > 32) return DSGCN_EINVAL;
  const size_t lds4 = (size_t)(T * (V + 1) + T * V + ((T + 3) & ~3) + 32) * sizeof(float);
  if ((T * V) % 4 == 0 && (T * (V + 1)) % 4 == 0 && lds4 <= 60 * 1024) {
    hipLaunchKernelGGL(k_tms_combine_bwd4, dim3((unsigned)((long)n * C)), dim3(64), lds4, (hipStream_t)stream, o, coeff,
                       gf, A0, B0, dout, pcoef, C, T, V);
    DSGCN_LAUNCH_CHECK();
    return 0;
  }
  const size_t lds = (size_t)T * V * sizeof(float);
  if (lds > 64 * 1024) return DSGCN_EUNSUPPORTED;
  hipLaunchKernelGGL(k_tms_combine_bwd, dim3((unsigned)((long)n * C)), dim3(64), lds, (hipStream_t)stream, o, coeff, gf,
                     A0, B0, dout, pcoef, C, T, V);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Depthwise causal temporal taps of unitmlp (reference: pyskl/models/gcns/utils/tcn.py:525-614: F.pad on the left, then
// a grouped Conv1d with groups = channels over the frames of every joint, tcn.py:586-592):
//   y[n,c,t',v] = b[c] + sum_j w[c,j] * h[n,c, t'*s - (KM-1-j)*dil[c], v]        (frames < 0 read as zero)
// dil[c] = 0 marks channels outside the mlp windows (the max-pool / pass-through branches of msmlp): y = 0 there.
// One wave per (n,c) plane; the backward also leaves the per-plane sums for dw (KM) and db.
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr int DW_MAXK = 5;      // unitmlp kernel 9 (ST-GCN temporal unit) -> 5 causal taps; msmlp kernel 3 -> 2

__global__ __launch_bounds__(64) void k_dwcausal_fwd(const float* __restrict__ h, const float* __restrict__ w,
                                                     const float* __restrict__ b, const int* __restrict__ dil,
                                                     float* __restrict__ y, int C, int T, int Tout, int V, int stride,
                                                     int KM) {
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int c = (int)(plane % C);
  const int d = dil[c];
  float* __restrict__ py = y + (size_t)plane * Tout * V;
  const int L = Tout * V;
  if (d == 0) {
    for (int i = lane; i < L; i += 64) py[i] = 0.f;
    return;
  }
  const float* __restrict__ ph = h + (size_t)plane * T * V;
  float wk[DW_MAXK];
#pragma unroll
  for (int j = 0; j < DW_MAXK; ++j) wk[j] = j < KM ? w[c * KM + j] : 0.f;
  const float bc = b ? b[c] : 0.f;
  for (int i = lane; i < L; i += 64) {
    const int tp = i / V, v = i - tp * V;
    float acc = bc;
#pragma unroll
    for (int j = 0; j < DW_MAXK; ++j) {
      const int t = tp * stride - (KM - 1 - j) * d;
      if (j < KM && t >= 0 && t < T) acc = fmaf(wk[j], ph[t * V + v], acc);
    }
    py[i] = acc;
  }
}

// dh[n,c,t,v] = sum_j w[c,j] * dy[n,c,t',v] with t'*s - (KM-1-j)*d = t;  part[plane] = [dw_0..dw_{KM-1}, db]
__global__ __launch_bounds__(64) void k_dwcausal_bwd(const float* __restrict__ h, const float* __restrict__ w,
                                                     const int* __restrict__ dil, const float* __restrict__ dy,
                                                     float* __restrict__ dh, float* __restrict__ part, int C, int T,
                                                     int Tout, int V, int stride, int KM) {
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int c = (int)(plane % C);
  const int d = dil[c];
  float* __restrict__ pdh = dh + (size_t)plane * T * V;
  float* __restrict__ pp = part + (size_t)plane * (DW_MAXK + 1);
  if (d == 0) {
    for (int i = lane; i < T * V; i += 64) pdh[i] = 0.f;
    if (lane <= DW_MAXK) pp[lane] = 0.f;
    return;
  }
  const float* __restrict__ ph = h + (size_t)plane * T * V;
  const float* __restrict__ pg = dy + (size_t)plane * Tout * V;
  float wk[DW_MAXK], sw[DW_MAXK];
#pragma unroll
  for (int j = 0; j < DW_MAXK; ++j) { wk[j] = j < KM ? w[c * KM + j] : 0.f; sw[j] = 0.f; }
  float sb = 0.f;
  for (int i = lane; i < T * V; i += 64) {
    const int t = i / V, v = i - t * V;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < DW_MAXK; ++j) {
      const int num = t + (KM - 1 - j) * d;
      if (j < KM && num % stride == 0) {
        const int tp = num / stride;
        if (tp < Tout) acc = fmaf(wk[j], pg[tp * V + v], acc);
      }
    }
    pdh[i] = acc;
  }
  for (int i = lane; i < Tout * V; i += 64) {
    const int tp = i / V, v = i - tp * V;
    const float g = pg[i];
    sb += g;
#pragma unroll
    for (int j = 0; j < DW_MAXK; ++j) {
      const int t = tp * stride - (KM - 1 - j) * d;
      if (j < KM && t >= 0 && t < T) sw[j] = fmaf(g, ph[t * V + v], sw[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < DW_MAXK; ++j) sw[j] = wave_sum(sw[j]);
  sb = wave_sum(sb);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < DW_MAXK; ++j) pp[j] = sw[j];
    pp[DW_MAXK] = sb;
  }
}

}  // namespace

extern "C" {

int dsgcn_dwcausal_fwd(const float* h, const float* w, const float* b, const int* dil, float* y, int n, int C, int T,
                       int V, int stride, int KM, void* stream) {
  if (!h || !w || !dil || !y || n <= 0 || C <= 0 || T <= 0 || V <= 0 || stride <= 0 || KM < 1 || KM > DW_MAXK)
    return DSGCN_EINVAL;
  const int Tout = (T + stride - 1) / stride;
  hipLaunchKernelGGL(k_dwcausal_fwd, dim3((unsigned)((long)n * C)), dim3(64), 0, (hipStream_t)stream, h, w, b, dil, y, C, T,
                     Tout, V, stride, KM);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// part (n*C, 6): per-plane [dw_0 .. dw_4, db]; the sum over n gives the parameter gradients
int dsgcn_dwcausal_bwd(const float* h, const float* w, const int* dil, const float* dy, float* dh, float* part, int n,
                       int C, int T, int V, int stride, int KM, void* stream) {
  if (!h || !w || !dil || !dy || !dh || !part || n <= 0 || C <= 0 || T <= 0 || V <= 0 || stride <= 0 || KM < 1 ||
      KM > DW_MAXK)
    return DSGCN_EINVAL;
  const int Tout = (T + stride - 1) / stride;
  hipLaunchKernelGGL(k_dwcausal_bwd, dim3((unsigned)((long)n * C)), dim3(64), 0, (hipStream_t)stream, h, w, dil, dy, dh,
                     part, C, T, Tout, V, stride, KM);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
