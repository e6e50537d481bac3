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
