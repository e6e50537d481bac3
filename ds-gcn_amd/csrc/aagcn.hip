// AAGCN attention gates (reference: pyskl/models/gcns/utils/gcn.py:447-459): three passes y <- y * g + y with the gate g
// broadcast over all but one or two axes — per (sample, joint), per (sample, frame), per (sample, channel) — each gate
// computed from a mean of the tensor the previous gate produced.  One kernel applies gate i and emits the mean gate
// i+1 needs, so the unit's output is read and written once per gate instead of three times (mean, multiply, add).
// One wave per (n, c) plane, the plane through LDS; HBM-bound elementwise work.
#include "common.h"

namespace {

constexpr int GT_NT = 256;

// g index of element (n, c, t, v) for mode 0 (n, V), 1 (n, T), 2 (n, C)
__device__ __forceinline__ int gt_gidx(int mode, int n, int c, int t, int v, int C, int T, int V) {
  return mode == 0 ? n * V + v : (mode == 1 ? n * T + t : n * C + c);
}

// out = y * (1 + g);  rmode 1: rout (n, C, T) = mean over joints of out;  rmode 2: rout (n, C) = mean over the plane
__global__ __launch_bounds__(GT_NT) void k_gate_fwd(const float* __restrict__ y, const float* __restrict__ g, int mode,
                                                    float* __restrict__ out, float* __restrict__ rout, int rmode,
                                                    long planes, int C, int T, int V) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long plane = (long)blockIdx.x * 4 + wave;
  if (plane >= planes) return;
  const int L = T * V;
  float* pl = lds + (size_t)wave * L;
  const int n = (int)(plane / C), c = (int)(plane - (long)n * C);
  const float* py = y + plane * L;
  float* po = out + plane * L;
  const float invV = 1.f / (float)V;
  float acc = 0.f;
  for (int i = lane; i < L; i += 64) {
    int t, v;
    divmod_small(i, V, invV, t, v);
    const float o = py[i] * (1.f + g[gt_gidx(mode, n, c, t, v, C, T, V)]);
    po[i] = o;
    acc += o;
    if (rmode == 1) pl[i] = o;
  }
  if (rmode == 2) {
    acc = wave_sum(acc);
    if (lane == 0) rout[plane] = acc / (float)L;
  } else if (rmode == 1) {
    wave_lds_sync();
    for (int t = lane; t < T; t += 64) {
      float s = 0.f;
      for (int v = 0; v < V; ++v) s += pl[t * V + v];
      rout[plane * T + t] = s * invV;
    }
  }
}

// G = gout + (gradient of the mean this pass emitted);  dy = G * (1 + g);  dgp = sum of G * y over the axes g is
// broadcast along INSIDE the plane: mode 0 -> (n, C, V) sums over frames, 1 -> (n, C, T) sums over joints, 2 -> (n, C)
__global__ __launch_bounds__(GT_NT) void k_gate_bwd(const float* __restrict__ y, const float* __restrict__ g, int mode,
                                                    const float* __restrict__ gout, const float* __restrict__ drout,
                                                    int rmode, float* __restrict__ dy, float* __restrict__ dgp,
                                                    long planes, int C, int T, int V) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long plane = (long)blockIdx.x * 4 + wave;
  if (plane >= planes) return;
  const int L = T * V;
  float* pl = lds + (size_t)wave * L;
  const int n = (int)(plane / C), c = (int)(plane - (long)n * C);
  const float* py = y + plane * L;
  const float* pg = gout ? gout + plane * L : nullptr;        // (no direct gradient of the gated output: only the reduced one)
  float* pd = dy + plane * L;
  const float invV = 1.f / (float)V;
  const float dplane = (rmode == 2 && drout) ? drout[plane] / (float)L : 0.f;
  float acc = 0.f;
  for (int i = lane; i < L; i += 64) {
    int t, v;
    divmod_small(i, V, invV, t, v);
    float G = pg ? pg[i] : 0.f;
    if (rmode == 1 && drout) G += drout[plane * T + t] * invV;
    G += dplane;
    const float yy = py[i];
    pd[i] = G * (1.f + g[gt_gidx(mode, n, c, t, v, C, T, V)]);
    const float gy = G * yy;
    if (mode == 2) acc += gy; else pl[i] = gy;
  }
  if (mode == 2) {
    acc = wave_sum(acc);
    if (lane == 0) dgp[plane] = acc;
    return;
  }
  wave_lds_sync();
  if (mode == 0) {
    for (int v = lane; v < V; v += 64) {
      float s = 0.f;
      for (int t = 0; t < T; ++t) s += pl[t * V + v];
      dgp[plane * V + v] = s;
    }
  } else {
    for (int t = lane; t < T; t += 64) {
      float s = 0.f;
      for (int v = 0; v < V; ++v) s += pl[t * V + v];
      dgp[plane * T + t] = s;
    }
  }
}

}  // namespace

extern "C" {

int dsgcn_gate_fwd(const float* y, const float* g, int mode, float* out, float* rout, int rmode, int n, int C, int T,
                   int V, void* stream) {
  if (!y || !g || !out || mode < 0 || mode > 2 || rmode < 0 || rmode > 2 || (rmode && !rout) || n <= 0 || C <= 0 ||
      T <= 0 || V <= 0)
    return DSGCN_EINVAL;
  const size_t lds = (size_t)4 * T * V * sizeof(float);
  if (lds > 64 * 1024) return DSGCN_EUNSUPPORTED;
  const long planes = (long)n * C;
  hipLaunchKernelGGL(k_gate_fwd, dim3((unsigned)((planes + 3) / 4)), dim3(GT_NT), lds, (hipStream_t)stream, y, g, mode, out,
                     rout, rmode, planes, C, T, V);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_gate_bwd(const float* y, const float* g, int mode, const float* gout, const float* drout, int rmode, float* dy,
                   float* dgp, int n, int C, int T, int V, void* stream) {
  if (!y || !g || !dy || !dgp || mode < 0 || mode > 2 || rmode < 0 || rmode > 2 || n <= 0 || C <= 0 || T <= 0 || V <= 0)
    return DSGCN_EINVAL;
  const size_t lds = (size_t)4 * T * V * sizeof(float);
  if (lds > 64 * 1024) return DSGCN_EUNSUPPORTED;
  const long planes = (long)n * C;
  hipLaunchKernelGGL(k_gate_bwd, dim3((unsigned)((planes + 3) / 4)), dim3(GT_NT), lds, (hipStream_t)stream, y, g, mode,
                     gout, drout, rmode, dy, dgp, planes, C, T, V);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
