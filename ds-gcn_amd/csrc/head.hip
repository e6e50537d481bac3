// The small ends of the training step as a handful of launches (round 5): what used to be ~45 framework launches of
// 4-6 us each around the backbone (the step's kernels run back to back in a hipGraph: a launch is ~4.7 us however little
// it does).
//   k_head_fwd / k_head_fin / k_head_bwd   person mean + Linear + softmax cross entropy + top-1 / top-5 accuracy
//       (pyskl/models/heads/simple_head.py:60-98 'GCN' mode after the plane mean, heads/base.py:50-84,
//       losses/cross_entropy_loss.py:75-82 with base.py:38-44's loss_weight, core/evaluation.py:63-88 top_k_accuracy)
//   k_sgd                                  the optimizer's five elementwise passes over the flat buffers as one
//   k_bn_running_multi                     running_mean / running_var / num_batches_tracked of every BatchNorm of a
//       forward in one launch (torch.nn.functional.batch_norm's training-mode buffer update, momentum form)
// All sums run in a fixed order (no atomics): the step stays bit-reproducible.
#include "common.h"

namespace {

constexpr int HD_NT = 256;

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {      // red: >= 4 floats; all threads get the sum
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float block_max(float v, float* red, int tid) {
  v = wave_max(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// One workgroup per clip n.  LDS: pl[C] pooled features, sc[K] scores.
//   pooled[n, c] = (1/M) sum_m feat[n*M + m, c];  score[n, k] = <pooled[n], W[k]> + b[k]  (a wave per class: lanes split C)
//   prob = softmax(score);  clip[n] = (-log prob[label], rank < 1, rank < 5) with rank = classes that a stable ascending
//   argsort of the scores puts after the label (greater score, or equal score and greater index).
__global__ __launch_bounds__(HD_NT) void k_head_fwd(const float* __restrict__ feat, const float* __restrict__ w,
                                                    const float* __restrict__ b, const long long* __restrict__ label,
                                                    int M, int C, int K, float* __restrict__ pooled,
                                                    float* __restrict__ score, float* __restrict__ prob,
                                                    float* __restrict__ clip) {
  extern __shared__ float hs[];
  __shared__ float red[4];
  float* pl = hs;
  float* sc = hs + C;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float invM = 1.f / (float)M;
  for (int c = tid; c < C; c += HD_NT) {
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += feat[((size_t)n * M + m) * C + c];
    s *= invM;
    pl[c] = s;
    pooled[(size_t)n * C + c] = s;
  }
  __syncthreads();
  // wave w takes classes w, w + 4, ...: eight of them per pass, all their loads issued before the first product (the
  // loop with one class per trip was a chain of ~16 dependent L2 round trips: 31 us for 1 MFLOP)
  const bool c4 = (C & 3) == 0;
  for (int k0 = wave; k0 < K; k0 += 32) {
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    if (c4) {
      for (int c = 4 * lane; c < C; c += 256) {
        f32x4 wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          wv[j] = (k0 + 4 * j < K) ? *reinterpret_cast<const f32x4*>(w + (size_t)(k0 + 4 * j) * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 p = *reinterpret_cast<const f32x4*>(pl + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += (p.x * wv[j].x + p.y * wv[j].y) + (p.z * wv[j].z + p.w * wv[j].w);
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = (k0 + 4 * j < K) ? w[(size_t)(k0 + 4 * j) * C + c] : 0.f;
        const float p = pl[c];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fmaf(p, wv[j], a[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = wave_sum(a[j]);
      if (lane == 0 && k0 + 4 * j < K) sc[k0 + 4 * j] = s + (b ? b[k0 + 4 * j] : 0.f);
    }
  }
  __syncthreads();
  const long long lb = label[n];
  const bool valid = lb >= 0 && lb < K;
  const float sl = valid ? sc[lb] : __builtin_nanf("");
  float mx = -__builtin_inff();
  for (int k = tid; k < K; k += HD_NT) mx = fmaxf(mx, sc[k]);
  mx = block_max(mx, red, tid);
  float se = 0.f, rk = 0.f;
  for (int k = tid; k < K; k += HD_NT) {
    const float s = sc[k];
    se += __expf(s - mx);
    rk += (s > sl || (s == sl && k > lb)) ? 1.f : 0.f;
  }
  se = block_sum(se, red, tid);
  rk = block_sum(rk, red, tid);
  const float inv = 1.f / se;
  for (int k = tid; k < K; k += HD_NT) {
    const float s = sc[k];
    score[(size_t)n * K + k] = s;
    prob[(size_t)n * K + k] = __expf(s - mx) * inv;
  }
  if (tid == 0) {
    clip[n * 3 + 0] = (__logf(se) + mx) - sl;
    clip[n * 3 + 1] = (valid && rk < 1.f) ? 1.f : 0.f;
    clip[n * 3 + 2] = (valid && rk < 5.f) ? 1.f : 0.f;
  }
}

// loss = loss_weight * mean_n clip[n, 0] (f32), acc = mean_n clip[n, 1..2] (f64): one wave, fp64, fixed order
__global__ __launch_bounds__(64) void k_head_fin(const float* __restrict__ clip, int N, float loss_weight,
                                                 float* __restrict__ loss, double* __restrict__ acc) {
  double s0 = 0., s1 = 0., s2 = 0.;
  for (int n = threadIdx.x; n < N; n += 64) {
    s0 += (double)clip[n * 3];
    s1 += (double)clip[n * 3 + 1];
    s2 += (double)clip[n * 3 + 2];
  }
  s0 = wave_sum_d(s0); s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  if (threadIdx.x == 0) {
    loss[0] = (float)((double)loss_weight * s0 / N);
    acc[0] = s1 / N;
    acc[1] = s2 / N;
  }
}

// Blocks [0, N): dfeat rows of clip n (every person gets dpooled / M);  blocks [N, N + K): row k of dW and db[k].
// dscore[n, k] = (prob[n, k] - [k == label[n]]) * gloss * loss_weight / N.
__global__ __launch_bounds__(HD_NT) void k_head_bwd(const float* __restrict__ prob, const float* __restrict__ pooled,
                                                    const float* __restrict__ w, const long long* __restrict__ label,
                                                    const float* __restrict__ gloss, float loss_weight, int N, int M, int C,
                                                    int K, float* __restrict__ dfeat, float* __restrict__ dw,
                                                    float* __restrict__ db) {
  extern __shared__ float hs[];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const float g = gloss[0] * loss_weight / (float)N;
  if ((int)blockIdx.x < N) {
    const int n = blockIdx.x;
    const long long lb = label[n];
    for (int k = tid; k < K; k += HD_NT) hs[k] = (prob[(size_t)n * K + k] - (k == lb ? 1.f : 0.f)) * g;
    __syncthreads();
    const float invM = 1.f / (float)M;
    for (int c = tid; c < C; c += HD_NT) {
      float a0 = 0.f, a1 = 0.f;
      int k = 0;
      for (; k + 1 < K; k += 2) {
        a0 = fmaf(hs[k], w[(size_t)k * C + c], a0);
        a1 = fmaf(hs[k + 1], w[(size_t)(k + 1) * C + c], a1);
      }
      if (k < K) a0 = fmaf(hs[k], w[(size_t)k * C + c], a0);
      const float d = (a0 + a1) * invM;
      for (int m = 0; m < M; ++m) dfeat[((size_t)n * M + m) * C + c] = d;
    }
  } else {
    const int k = blockIdx.x - N;
    float part = 0.f;
    for (int n = tid; n < N; n += HD_NT) {
      const float d = (prob[(size_t)n * K + k] - (label[n] == k ? 1.f : 0.f)) * g;
      hs[n] = d;
      part += d;
    }
    part = block_sum(part, red, tid);                               // (its barriers also publish hs)
    if (tid == 0) db[k] = part;
    for (int c = tid; c < C; c += HD_NT) {
      float a0 = 0.f, a1 = 0.f;
      int n = 0;
      for (; n + 1 < N; n += 2) {
        a0 = fmaf(hs[n], pooled[(size_t)n * C + c], a0);
        a1 = fmaf(hs[n + 1], pooled[(size_t)(n + 1) * C + c], a1);
      }
      if (n < N) a0 = fmaf(hs[n], pooled[(size_t)n * C + c], a0);
      dw[(size_t)k * C + c] = a0 + a1;
    }
  }
}

// ---- BatchNorm buffers ----
constexpr int RN_MAXJOBS = 64;
struct RnJobs {
  float* rm[RN_MAXJOBS];
  float* rv[RN_MAXJOBS];
  const float* mean[RN_MAXJOBS];
  const float* var[RN_MAXJOBS];
  long long* nbt[RN_MAXJOBS];
  int C[RN_MAXJOBS];
  float unbias[RN_MAXJOBS], mom[RN_MAXJOBS];
  int njobs;
};

// block j: rm = (1 - m) rm + m mean;  rv = (1 - m) rv + m (var * unbias);  nbt += 1   (the op order of the
// multi-tensor form this replaces: scale the buffer, then add the scaled statistic)
__global__ __launch_bounds__(256) void k_bn_running_multi(RnJobs jb) {
  const int j = blockIdx.x;
  const int C = jb.C[j];
  const float m = jb.mom[j], keep = 1.f - m, ub = jb.unbias[j];
  float* __restrict__ rm = jb.rm[j];
  float* __restrict__ rv = jb.rv[j];
  const float* __restrict__ mean = jb.mean[j];
  const float* __restrict__ var = jb.var[j];
  for (int c = threadIdx.x; c < C; c += 256) {
    rm[c] = rm[c] * keep + m * mean[c];
    rv[c] = rv[c] * keep + m * (var[c] * ub);
  }
  if (threadIdx.x == 0 && jb.nbt[j]) jb.nbt[j][0] += 1;
}

// SGD with momentum (dampening 0) over the flat buffers, torch.optim.SGD's update order:
//   g' = g + wd p;  buf = mom buf + g';  step = nesterov ? g' + mom buf : buf;  p -= lr step      (lr read on the device)
__global__ __launch_bounds__(256) void k_sgd(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                             const float* __restrict__ lr, float mom, float wd, int nesterov, long n4,
                                             long n) {
  const float rate = lr[0];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  auto one = [&](float pv, float gv, float& bv) {
    gv = gv + wd * pv;
    float st = gv;
    if (buf) {
      bv = bv * mom + gv;
      st = nesterov ? gv + mom * bv : bv;
    }
    return pv - rate * st;
  };
  if (i < n4) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 bv = buf ? reinterpret_cast<f32x4*>(buf)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float be = bv[e];
      pv[e] = one(pv[e], gv[e], be);
      bv[e] = be;
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    if (buf) reinterpret_cast<f32x4*>(buf)[i] = bv;
  } else if (i == n4) {                                             // the tail (n % 4 elements)
    for (long j = 4 * n4; j < n; ++j) {
      float bv = buf ? buf[j] : 0.f;
      p[j] = one(p[j], g[j], bv);
      if (buf) buf[j] = bv;
    }
  }
}

}  // namespace

extern "C" {

int dsgcn_sgd_step(float* p, const float* g, float* buf, const float* lr, float momentum, float weight_decay,
                   int nesterov, long long n, void* stream) {
  if (!p || !g || !lr || n <= 0 || (momentum != 0.f && !buf)) return DSGCN_EINVAL;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) return DSGCN_EINVAL;
  const long n4 = (long)(n / 4);
  hipLaunchKernelGGL(k_sgd, dim3((unsigned)((n4 + 1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g,
                     momentum != 0.f ? buf : nullptr, lr, momentum, weight_decay, nesterov, n4, (long)n);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_head_loss_fwd(const float* feat, const float* w, const float* b, const long long* label, int N, int M, int C,
                        int K, float loss_weight, float* pooled, float* score, float* prob, float* clip, float* loss,
                        double* acc, void* stream) {
  if (!feat || !w || !label || !pooled || !score || !prob || !clip || !loss || !acc || N <= 0 || M <= 0 || C <= 0 || K <= 0)
    return DSGCN_EINVAL;
  if ((size_t)(C + K) * sizeof(float) > 60 * 1024) return DSGCN_EUNSUPPORTED;
  hipLaunchKernelGGL(k_head_fwd, dim3((unsigned)N), dim3(HD_NT), (size_t)(C + K) * sizeof(float), (hipStream_t)stream,
                     feat, w, b, label, M, C, K, pooled, score, prob, clip);
  DSGCN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_head_fin, dim3(1), dim3(64), 0, (hipStream_t)stream, clip, N, loss_weight, loss, acc);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_head_loss_bwd(const float* prob, const float* pooled, const float* w, const long long* label, const float* gloss,
                        int N, int M, int C, int K, float loss_weight, float* dfeat, float* dw, float* db, void* stream) {
  if (!prob || !pooled || !w || !label || !gloss || !dfeat || !dw || !db || N <= 0 || M <= 0 || C <= 0 || K <= 0)
    return DSGCN_EINVAL;
  const int lds = (K > N ? K : N) * (int)sizeof(float);
  if (lds > 60 * 1024) return DSGCN_EUNSUPPORTED;
  hipLaunchKernelGGL(k_head_bwd, dim3((unsigned)(N + K)), dim3(HD_NT), (size_t)lds, (hipStream_t)stream, prob, pooled, w,
                     label, gloss, loss_weight, N, M, C, K, dfeat, dw, db);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_bn_running_multi(float* const* running_mean, float* const* running_var, const float* const* mean,
                           const float* const* var, long long* const* num_batches_tracked, const int* C,
                           const float* unbias, const float* momentum, int njobs, void* stream) {
  if (!running_mean || !running_var || !mean || !var || !num_batches_tracked || !C || !unbias || !momentum || njobs <= 0)
    return DSGCN_EINVAL;
  for (int j = 0; j < njobs; ++j)
    if (!running_mean[j] || !running_var[j] || !mean[j] || !var[j] || C[j] <= 0) return DSGCN_EINVAL;
  for (int j0 = 0; j0 < njobs; j0 += RN_MAXJOBS) {
    RnJobs jb;
    jb.njobs = njobs - j0 < RN_MAXJOBS ? njobs - j0 : RN_MAXJOBS;
    for (int j = 0; j < jb.njobs; ++j) {
      jb.rm[j] = running_mean[j0 + j]; jb.rv[j] = running_var[j0 + j];
      jb.mean[j] = mean[j0 + j]; jb.var[j] = var[j0 + j];
      jb.nbt[j] = num_batches_tracked[j0 + j];
      jb.C[j] = C[j0 + j]; jb.unbias[j] = unbias[j0 + j]; jb.mom[j] = momentum[j0 + j];
    }
    hipLaunchKernelGGL(k_bn_running_multi, dim3((unsigned)jb.njobs), dim3(256), 0, (hipStream_t)stream, jb);
    DSGCN_LAUNCH_CHECK();
  }
  return 0;
}

}  // extern "C"

// ---- the input BatchNorm (data_bn) ---------------------------------------------------------------------------------
// pyskl/models/gcns/dgstgcn.py:158-164 (stgcn.py:132-139, ctrgcn.py:112-118, aagcn.py:128-135): the clip (N, M, T, V, C) is
// permuted to (N*M, V*C, T) ['VC'] or (N, M*V*C, T) ['MVC'], normalised by a BatchNorm1d over its V*C resp. M*V*C channels
// and permuted to (N*M, C, T, V).  There: a permute copy, the BatchNorm (+ its buffer updates), a permute copy; the same
// again backwards.  Here: per-sample channel sums (k_dbn_stats), then ONE launch that finishes the statistics (every
// workgroup for itself: the table is NM x VC x 2 floats), updates the buffers and writes the normalised clip straight in
// the (N*M, C, T, V) layout through an LDS transpose (k_dbn_apply); backward: one launch of per-sample partial sums of
// dgamma / dbeta (the clip itself needs no gradient).
namespace {

constexpr int DB_NT = 256;

struct DbnArgs {
  const float* x;          // (NM, T, VC)
  double* psum;            // (NM, VC)   fp64: the variance below is E[x^2] - mean^2, which cancels in fp32 when the clip's
  double* psq;             // (NM, VC)   mean is large against its spread (raw pixel coordinates)
  int NM, M, T, V, C, mvc;
};

// thread (slot, col): col = tid % VC, rows slot, slot + slots, ...  (threads beyond slots * VC idle)
__global__ __launch_bounds__(DB_NT) void k_dbn_stats(DbnArgs a) {
  __shared__ double red[2 * DB_NT];
  const int VC = a.V * a.C, slots = DB_NT / VC, tid = threadIdx.x;
  const int slot = tid / VC, col = tid - slot * VC;
  const int nm = blockIdx.x;
  double s = 0., q = 0.;
  if (slot < slots) {
    const float* __restrict__ p = a.x + (size_t)nm * a.T * VC + col;
#pragma unroll 8
    for (int t = slot; t < a.T; t += slots) {
      const double v = (double)p[(size_t)t * VC];
      s += v;
      q = fma(v, v, q);
    }
  }
  red[tid] = s;
  red[DB_NT + tid] = q;
  __syncthreads();
  if (tid < VC) {
    double ts = 0., tq = 0.;
    for (int j = 0; j < slots; ++j) { ts += red[j * VC + tid]; tq += red[DB_NT + j * VC + tid]; }
    a.psum[(size_t)nm * VC + tid] = ts;
    a.psq[(size_t)nm * VC + tid] = tq;
  }
}

struct DbnApply {
  const float* x;
  const double* psum;
  const double* psq;
  const float* gamma;      // (Ch) or NULL
  const float* beta;       // (Ch) or NULL
  float* rmean;            // running buffers (Ch) or NULL
  float* rvar;
  long long* nbt;          // or NULL
  float* y;                // (NM, C, T, V)
  float* save_mean;        // (Ch)   (training: batch statistics for the backward)
  float* save_invstd;      // (Ch)
  int NM, M, T, V, C, mvc, training;
  float eps, momentum;
};

__global__ __launch_bounds__(DB_NT) void k_dbn_apply(DbnApply a) {
  extern __shared__ float tile[];                 // [T][VC] normalised values
  __shared__ double dred[2 * DB_NT];
  __shared__ float sc[DB_NT], sf[DB_NT];
  const int VC = a.V * a.C, slots = DB_NT / VC, tid = threadIdx.x;
  const int slot = tid / VC, col = tid - slot * VC;
  const int nm = blockIdx.x, m = nm % a.M;
  const int ch = a.mvc ? m * VC + col : col;      // BatchNorm channel of this thread's column
  // 1. statistics of this sample's channels
  if (a.training) {
    double s = 0., q = 0.;
    if (slot < slots) {
      // 'VC': every sample contributes; 'MVC': the samples of person m
      const int first = a.mvc ? m : 0, step = a.mvc ? a.M : 1;
#pragma unroll 8
      for (int r = first + slot * step; r < a.NM; r += slots * step) {
        s += a.psum[(size_t)r * VC + col];
        q += a.psq[(size_t)r * VC + col];
      }
    }
    dred[tid] = s;
    dred[DB_NT + tid] = q;
    __syncthreads();
    if (tid < VC) {
      double ts = 0., tq = 0.;
      for (int j = 0; j < slots; ++j) { ts += dred[j * VC + tid]; tq += dred[DB_NT + j * VC + tid]; }
      const double cnt = (double)(a.mvc ? a.NM / a.M : a.NM) * a.T;
      const double mean = ts / cnt;
      double var = tq / cnt - mean * mean;
      var = var > 0. ? var : 0.;
      const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
      const float g = a.gamma ? a.gamma[ch] : 1.f, b = a.beta ? a.beta[ch] : 0.f;
      sc[tid] = g * invstd;
      sf[tid] = (float)((double)b - mean * (double)(g * invstd));
      // one workgroup per BatchNorm channel set keeps the books: sample m ('MVC') / sample 0 ('VC')
      if (nm == (a.mvc ? m : 0)) {
        a.save_mean[ch] = (float)mean;
        a.save_invstd[ch] = invstd;
        if (a.rmean) {
          const double unb = cnt > 1. ? var * cnt / (cnt - 1.) : var;
          a.rmean[ch] = a.rmean[ch] * (1.f - a.momentum) + a.momentum * (float)mean;
          a.rvar[ch] = a.rvar[ch] * (1.f - a.momentum) + a.momentum * (float)unb;
        }
      }
    }
    if (nm == 0 && tid == 0 && a.nbt) a.nbt[0] += 1;
  } else if (tid < VC) {
    const float invstd = rsqrtf(a.rvar[ch] + a.eps);
    const float g = a.gamma ? a.gamma[ch] : 1.f, b = a.beta ? a.beta[ch] : 0.f;
    sc[tid] = g * invstd;
    sf[tid] = b - a.rmean[ch] * g * invstd;
  }
  __syncthreads();
  // 2. normalise into LDS as [t][v*C + c] ...
  const int TV = a.T * VC;
  const float* __restrict__ px = a.x + (size_t)nm * TV;
  const int NA = slots * VC;                      // active threads: a thread keeps its column
  if (tid < NA) {
    const float s_ = sc[col], f_ = sf[col];
#pragma unroll 8
    for (int e = tid; e < TV; e += NA) tile[e] = fmaf(px[e], s_, f_);
  }
  __syncthreads();
  // 3. ... and out as (C, T, V): consecutive threads write consecutive addresses
  float* __restrict__ py = a.y + (size_t)nm * TV;
  const int TVv = a.T * a.V;
#pragma unroll 4
  for (int o = tid; o < TV; o += DB_NT) {
    const int c = o / TVv, r = o - c * TVv;       // r = t*V + v
    py[o] = tile[r * a.C + c];
  }
}

struct DbnBwd {
  const float* x;          // (NM, T, VC)
  const float* dy;         // (NM, C, T, V)
  const float* mean;       // (Ch)
  const float* invstd;     // (Ch)
  float* pg;               // (NM, VC) partial sums of dy * xhat
  float* pb;               // (NM, VC) partial sums of dy
  int NM, M, T, V, C, mvc;
};

__global__ __launch_bounds__(DB_NT) void k_dbn_bwd(DbnBwd a) {
  extern __shared__ float tile[];                 // dy as [t][v*C + c]
  __shared__ float red[2 * DB_NT];
  const int VC = a.V * a.C, slots = DB_NT / VC, tid = threadIdx.x;
  const int slot = tid / VC, col = tid - slot * VC;
  const int nm = blockIdx.x, m = nm % a.M;
  const int TV = a.T * VC, TVv = a.T * a.V;
  const float* __restrict__ pd = a.dy + (size_t)nm * TV;
#pragma unroll 4
  for (int o = tid; o < TV; o += DB_NT) {
    const int c = o / TVv, r = o - c * TVv;
    tile[r * a.C + c] = pd[o];
  }
  __syncthreads();
  float g = 0.f, b = 0.f;
  if (slot < slots) {
    const int ch = a.mvc ? m * VC + col : col;
    const float mu = a.mean[ch], is = a.invstd[ch];
    const float* __restrict__ px = a.x + (size_t)nm * TV + col;
#pragma unroll 8
    for (int t = slot; t < a.T; t += slots) {
      const float d = tile[t * VC + col];
      g = fmaf(d, (px[(size_t)t * VC] - mu) * is, g);
      b += d;
    }
  }
  red[tid] = g;
  red[DB_NT + tid] = b;
  __syncthreads();
  if (tid < VC) {
    float tg = 0.f, tb = 0.f;
    for (int j = 0; j < slots; ++j) { tg += red[j * VC + tid]; tb += red[DB_NT + j * VC + tid]; }
    a.pg[(size_t)nm * VC + tid] = tg;
    a.pb[(size_t)nm * VC + tid] = tb;
  }
}

}  // namespace

extern "C" {

// x (N, M, T, V, C) -> y (N*M, C, T, V).  mvc: 0 = 'VC' (V*C channels, statistics over N*M samples and T), 1 = 'MVC'
// (M*V*C channels, over N and T).  training: batch statistics (saved to save_mean / save_invstd; running_mean /
// running_var / num_batches_tracked updated with `momentum` when given), else the running statistics.
// scratch: 4 * N*M * V*C floats (two fp64 tables), 8-byte aligned.  Two launches (training) / one (eval).
int dsgcn_data_bn_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                      long long* num_batches_tracked, float* y, float* save_mean, float* save_invstd, float* scratch,
                      int N, int M, int T, int V, int C, int mvc, int training, float eps, float momentum,
                      void* stream) {
  if (!x || !y || N <= 0 || M <= 0 || T <= 0 || V <= 0 || C <= 0) return DSGCN_EINVAL;
  if (training ? (!save_mean || !save_invstd || !scratch) : (!running_mean || !running_var)) return DSGCN_EINVAL;
  const int VC = V * C, NM = N * M;
  if (VC > DB_NT || (size_t)T * VC * sizeof(float) > 60 * 1024) return DSGCN_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  double* tab = reinterpret_cast<double*>(scratch);
  if (training) {
    if ((uintptr_t)scratch & 7) return DSGCN_EINVAL;
    DbnArgs s = {x, tab, tab + (size_t)NM * VC, NM, M, T, V, C, mvc};
    hipLaunchKernelGGL(k_dbn_stats, dim3((unsigned)NM), dim3(DB_NT), 0, st, s);
    DSGCN_LAUNCH_CHECK();
  }
  DbnApply a = {x, tab, tab ? tab + (size_t)NM * VC : nullptr, gamma, beta, running_mean, running_var,
                num_batches_tracked, y, save_mean, save_invstd, NM, M, T, V, C, mvc, training, eps, momentum};
  hipLaunchKernelGGL(k_dbn_apply, dim3((unsigned)NM), dim3(DB_NT), (size_t)T * VC * sizeof(float), st, a);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// pg / pb (N*M, V*C): per-sample partial sums of dgamma / dbeta; their column sum over the samples ('VC': all N*M rows;
// 'MVC': view them as (N, M*V*C)) finishes them.  The clip needs no gradient.  One launch.
int dsgcn_data_bn_bwd(const float* x, const float* dy, const float* save_mean, const float* save_invstd, float* pg,
                      float* pb, int N, int M, int T, int V, int C, int mvc, void* stream) {
  if (!x || !dy || !save_mean || !save_invstd || !pg || !pb || N <= 0 || M <= 0 || T <= 0 || V <= 0 || C <= 0)
    return DSGCN_EINVAL;
  const int VC = V * C;
  if (VC > DB_NT || (size_t)T * VC * sizeof(float) > 60 * 1024) return DSGCN_EUNSUPPORTED;
  DbnBwd a = {x, dy, save_mean, save_invstd, pg, pb, N * M, M, T, V, C, mvc};
  hipLaunchKernelGGL(k_dbn_bwd, dim3((unsigned)(N * M)), dim3(DB_NT), (size_t)T * VC * sizeof(float), (hipStream_t)stream, a);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
