// Train-mode BatchNorm's two micro-reductions as JOBS: a job is what one k_bn_finalize / k_bn_coef_rows launch did
// (pwconv.hip, same arithmetic, same fixed summation order: results are bit-identical to the single launches); a launch
// takes a small table of them in its kernel arguments.  Two uses (round 6):
//   * k_bn_finalize_multi / k_bn_coef_rows_multi: the jobs of INDEPENDENT producers in one launch (`post` + `down`;
//     transform + block-residual conv — gcn.py:2142-2167,2215, tcn.py:389-404: the reference's nn.BatchNorm2d calls);
//   * hosted: a launch that sits between a producer and its consumer anyway (K-B between the `pre` conv and K-A) carries the
//     jobs as extra workgroups (bn_jobs_host_block below) — the `pre` BatchNorm costs no launch of its own.
#pragma once
#include "common.h"
#include "dsgcn_jobs.h"

constexpr int BNJ_MAX = DSGCN_BN_JOBS_MAX;         // jobs per launch
constexpr int BNJ_NT = 1024;                       // threads of a job block: 8 channels x 128 row slices

struct BnFinJob {                                  // partial (nblk, C, 2) -> mean, var, scale, shift (C each)
  const float* partial; const float* gamma; const float* beta;
  float* mean; float* var; float* scale; float* shift;
  double count; float eps; int nblk, C, c_affine;
};

struct BnCoefJob {                                 // part (R, C, k) -> coef (4, C) = [d gamma | d beta | A0 | B0]
  const float* part; const float* mean; const float* var; const float* gamma; float* coef;
  double count; float eps; int R, C, k, ids, idh, c_affine, accumulate;
};

struct BnFinTable { BnFinJob j[BNJ_MAX]; int n; };
struct BnCoefTable { BnCoefJob j[BNJ_MAX]; int n; };

__host__ __device__ inline int bnj_blocks(int C) { return (C + 7) / 8; }

// one NT-thread block of a finalize job (block `blk` of bnj_blocks(C): 8 channels x NT/8 row slices); `red` = NT/64 x 8 x 2
// doubles of LDS.  The summation order depends on NT: a job gives the same bits wherever it runs with the same NT, and
// launches that must agree bit for bit use the same NT (1024: the stand-alone launches and K-B's hosting launches).
template <int NT = BNJ_NT>
__device__ __forceinline__ void bn_finalize_block(const BnFinJob& J, int blk, double (*red)[8][2]) {
  constexpr int CB = 8, NS = NT / CB;
  const int cl = threadIdx.x & (CB - 1), slice = threadIdx.x / CB;
  const int C = J.C, nblk = J.nblk;
  const int c = blk * CB + cl;
  double s = 0.0, q = 0.0;
  if (c < C) {
    const float2* p2 = reinterpret_cast<const float2*>(J.partial);
    for (int b0 = slice; b0 < nblk; b0 += NS * 4) {
      float2 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int b = b0 + NS * j;
        v[j] = b < nblk ? p2[(size_t)b * C + c] : float2{0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { s += (double)v[j].x; q += (double)v[j].y; }
    }
  }
  // the 8 slices of a wave (lanes cl + 8*j) meet through three lane exchanges, the 16 waves through ONE LDS hand-off:
  // fixed order, deterministic
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
    s += __shfl_xor(s, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  if ((threadIdx.x & 63) < CB) {
    red[threadIdx.x >> 6][cl][0] = s;
    red[threadIdx.x >> 6][cl][1] = q;
  }
  __syncthreads();
  if (slice == 0 && c < C) {
    s = 0.0;
    q = 0.0;
#pragma unroll
    for (int w = 0; w < NS / 8; ++w) { s += red[w][cl][0]; q += red[w][cl][1]; }
    const double mean = s / J.count;
    double var = q / J.count - mean * mean;
    if (var < 0.0) var = 0.0;
    J.mean[c] = (float)mean;
    J.var[c] = (float)var;
    if (c < J.c_affine) {
      const float g = J.gamma ? J.gamma[c] : 1.f, bt = J.beta ? J.beta[c] : 0.f;
      const float rstd = (float)(1.0 / sqrt(var + (double)J.eps));
      const float sc = g * rstd;
      J.scale[c] = sc;
      J.shift[c] = bt - (float)mean * sc;
    } else {
      J.scale[c] = 1.f;
      J.shift[c] = 0.f;
    }
  }
}

// one NT-thread block of a coefficient job
template <int NT = BNJ_NT>
__device__ __forceinline__ void bn_coef_rows_block(const BnCoefJob& J, int blk, double (*red)[8][2]) {
  constexpr int CB = 8, NS = NT / CB;
  const int cl = threadIdx.x & (CB - 1), sl = threadIdx.x / CB;
  const int C = J.C, R = J.R, k = J.k;
  const int c = blk * CB + cl;
  // the finishing thread's own operands first: their round trip overlaps the row sums
  float mu_f = 0.f, var_f = 1.f, g_f = 1.f, old[4] = {0.f, 0.f, 0.f, 0.f};
  if (sl == 0 && c < C) {
    mu_f = J.mean[c]; var_f = J.var[c];
    if (J.gamma && c < J.c_affine) g_f = J.gamma[c];
    if (J.accumulate) {
#pragma unroll
      for (int j = 0; j < 4; ++j) old[j] = J.coef[(size_t)j * C + c];
    }
  }
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    const size_t rs = (size_t)C * k;
    const float* p = J.part + (size_t)c * k;
    for (int r0 = sl; r0 < R; r0 += NS * 4) {
      float x[4], y[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = r0 + NS * j;
        const float* q = p + (size_t)(r < R ? r : r0) * rs;           // (clamped: no predicated loads)
        x[j] = q[J.ids]; y[j] = q[J.idh];
        if (r >= R) { x[j] = 0.f; y[j] = 0.f; }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { s0 += (double)x[j]; s1 += (double)y[j]; }
    }
  }
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {          // a wave's 8 slices by lane exchange, the 16 waves through one LDS hand-off
    s0 += __shfl_xor(s0, off, 64);
    s1 += __shfl_xor(s1, off, 64);
  }
  if ((threadIdx.x & 63) < CB) {
    red[threadIdx.x >> 6][cl][0] = s0;
    red[threadIdx.x >> 6][cl][1] = s1;
  }
  __syncthreads();
  if (sl == 0 && c < C) {
    double gs = 0.0, gh = 0.0;
#pragma unroll
    for (int w = 0; w < NS / 8; ++w) { gs += red[w][cl][0]; gh += red[w][cl][1]; }
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < J.c_affine) {
      const double mu = mu_f, r = 1.0 / sqrt((double)var_f + (double)J.eps);
      const double g = g_f;
      const double t = gs - mu * gh;
      const double dmean = -gh * g * r;
      const double dvar = -0.5 * r * r * r * g * t;
      o[0] = (float)(r * t);
      o[1] = (float)gh;
      o[2] = (float)((dmean - 2.0 * dvar * mu) / J.count);
      o[3] = (float)(2.0 * dvar / J.count);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) J.coef[(size_t)j * C + c] = old[j] + o[j];
  }
}

// block `b` of the concatenated job blocks of a table -> run it; false when b lies past the table
template <typename TABLE, typename F>
__device__ __forceinline__ bool bnj_dispatch(const TABLE& t, int b, F&& run) {
#pragma unroll
  for (int i = 0; i < BNJ_MAX; ++i) {
    if (i < t.n) {
      const int nb = bnj_blocks(t.j[i].C);
      if (b < nb) { run(t.j[i], b); return true; }
      b -= nb;
    }
  }
  return false;
}

template <typename TABLE>
inline int bnj_total_blocks(const TABLE& t) {
  int nb = 0;
  for (int i = 0; i < t.n; ++i) nb += bnj_blocks(t.j[i].C);
  return nb;
}

// host-side argument checks shared by the entry points (pwconv.hip, dynadj.hip)
inline bool bnj_fin_ok(const BnFinTable& t) {
  if (t.n < 0 || t.n > BNJ_MAX) return false;
  for (int i = 0; i < t.n; ++i) {
    const BnFinJob& J = t.j[i];
    if (!J.partial || !J.mean || !J.var || !J.scale || !J.shift || J.nblk <= 0 || J.C <= 0) return false;
  }
  return true;
}
inline bool bnj_coef_ok(const BnCoefTable& t) {
  if (t.n < 0 || t.n > BNJ_MAX) return false;
  for (int i = 0; i < t.n; ++i) {
    const BnCoefJob& J = t.j[i];
    if (!J.part || !J.mean || !J.var || !J.coef || J.R <= 0 || J.C <= 0 || J.k <= 0 || J.ids < 0 || J.ids >= J.k || J.idh < 0 ||
        J.idh >= J.k)
      return false;
  }
  return true;
}
