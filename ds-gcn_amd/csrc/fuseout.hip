// Block output: the one place an activation is materialised.
//   out[n,c,t,v] = relu?( x1*s1[c]+h1[c]  (+ x2*s2[c]+h2[c] | + x2) )      xbar[n,c,v] = mean_t out[n,c,t,v]
//   relu bit 0 = the outer ReLU, bit 1 = a ReLU on the first term alone (CTR-GCN: MSTCN ends in its own ReLU before the
//   block adds the residual, msg3d_utils.py:139-141 + ctrgcn.py:60)
// Replaces BatchNorm2d (tcn.py:427 / gcn.py:2365) + residual add + ReLU (dgstgcn.py:63-65) and the
// x.mean(dim=-2) of the NEXT block's dynamic adjacency (gcn.py:2246) — one read of each operand, one write.
// One wave per (n,c) plane: coalesced 16-B loads, the plane passes through LDS only to form the per-joint time mean.
// HBM-bound: 4*(2 or 3)*T*V bytes per plane.
//
// Backward: dv = (dout + dxbar/T) * 1[pre>0];  dx1 = dv*s1, dx2 = dv*s2 (or dv);
//           per-plane partial sums of dv1*x1, dv, dv*x2, dv1  (-> d s1, d h2, d s2, d h1 after the sum over n;
//           dv1 = dv unless the first term has its own ReLU).
#include "common.h"
#include "dropout.h"
#include "dsgcn_jobs.h"

namespace {

__global__ __launch_bounds__(64) void k_fuse_out_fwd(const float* __restrict__ x1, const float* __restrict__ s1,
                                                     const float* __restrict__ h1, const float* __restrict__ x2,
                                                     const float* __restrict__ s2, const float* __restrict__ h2,
                                                     int relu, float* __restrict__ out, float* __restrict__ xbar, int C,
                                                     int T, int V, int vec, int ld, float* __restrict__ out_s2,
                                                     float* __restrict__ pmean, DropArgs dr) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int c = (int)(plane % C);
  const int L = T * V;
  const bool drop = dr.thresh != 0;                // (wave-uniform)
  const unsigned long long dstep = (drop && dr.step) ? (unsigned long long)*dr.step : 0ull;
  const float a1 = s1 ? s1[c] : 1.f, b1 = s1 ? h1[c] : 0.f;
  const float a2 = s2 ? s2[c] : 1.f, b2 = s2 ? h2[c] : 0.f;
  const float* p1 = x1 + (size_t)plane * L;
  const float* p2 = x2 ? x2 + (size_t)plane * L : nullptr;
  float* po = out ? out + (size_t)plane * L : nullptr;      // (out == NULL: only the plane mean is wanted — the last block)
  float psum = 0.f;
  if (vec) {
    const int L4 = L >> 2;
    const f32x4* q1 = reinterpret_cast<const f32x4*>(p1);
    const f32x4* q2 = reinterpret_cast<const f32x4*>(p2);
    f32x4* qo = reinterpret_cast<f32x4*>(po);
    f32x4* ql = reinterpret_cast<f32x4*>(lds);
#pragma unroll 4
    for (int i = lane; i < L4; i += 64) {
      f32x4 v = q1[i];
      v.x = fmaf(v.x, a1, b1); v.y = fmaf(v.y, a1, b1); v.z = fmaf(v.z, a1, b1); v.w = fmaf(v.w, a1, b1);
      if (relu & 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (drop) {                                 // dropout on the first term (the temporal unit's output), before the residual
        unsigned rw[4];
        drop_words(dr, dstep, (unsigned long long)plane * L4 + i, rw);
        v.x *= rw[0] >= dr.thresh ? dr.inv : 0.f; v.y *= rw[1] >= dr.thresh ? dr.inv : 0.f;
        v.z *= rw[2] >= dr.thresh ? dr.inv : 0.f; v.w *= rw[3] >= dr.thresh ? dr.inv : 0.f;
      }
      if (p2) {
        const f32x4 r = q2[i];
        v.x += fmaf(r.x, a2, b2); v.y += fmaf(r.y, a2, b2); v.z += fmaf(r.z, a2, b2); v.w += fmaf(r.w, a2, b2);
      }
      if (relu & 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (po) qo[i] = v;
      psum += (v.x + v.y) + (v.z + v.w);
      if (xbar || out_s2) ql[i] = v;
    }
  } else {
    for (int i = lane; i < L; i += 64) {
      float v = fmaf(p1[i], a1, b1);
      if (relu & 2) v = fmaxf(v, 0.f);
      if (drop) v *= drop_mult(dr, dstep, (unsigned long long)plane * L + i);
      if (p2) v += fmaf(p2[i], a2, b2);
      if (relu & 1) v = fmaxf(v, 0.f);
      if (po) po[i] = v;
      psum += v;
      if (xbar || out_s2) lds[i] = v;
    }
  }
  if (pmean) {
    psum = wave_sum(psum);
    if (lane == 0) pmean[plane] = psum / (float)L;
  }
  if (xbar || out_s2) wave_lds_sync();
  if (out_s2) {
    // the even frames as a tensor of their own, (n, C, ceil(T/2), V): what a stride-2 block's residual conv reads (it used to
    // be a strided-copy launch of its own, and its backward a scatter into a zero-filled full-size tensor)
    const int T2 = (T + 1) >> 1, L2 = T2 * V;
    float* ps = out_s2 + (size_t)plane * L2;
    const float invV = 1.f / (float)V;
    if ((L2 & 3) == 0) {
      f32x4* q4 = reinterpret_cast<f32x4*>(ps);
      for (int i = lane; i < (L2 >> 2); i += 64) {
        int t, v;
        divmod_small(4 * i, V, invV, t, v);
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          r[k] = lds[2 * t * V + v];
          if (++v == V) { v = 0; ++t; }
        }
        q4[i] = f32x4{r[0], r[1], r[2], r[3]};
      }
    } else {
      for (int i = lane; i < L2; i += 64) {
        int t, v;
        divmod_small(i, V, invV, t, v);
        ps[i] = lds[2 * t * V + v];
      }
    }
  }
  if (xbar) {
    if (lane < ld) {                              // ld >= V: the joint row is zero-padded to ld (ld <= 64)
      float s = 0.f;
      if (lane < V)
        for (int t = 0; t < T; ++t) s += lds[t * V + lane];
      xbar[(size_t)plane * ld + lane] = s / (float)T;
    }
  }
}

__global__ __launch_bounds__(64) void k_fuse_out_bwd(const float* __restrict__ x1, const float* __restrict__ s1,
                                                     const float* __restrict__ h1, const float* __restrict__ x2,
                                                     const float* __restrict__ s2, const float* __restrict__ h2,
                                                     int relu, const float* __restrict__ dout,
                                                     const float* __restrict__ dout2, const float* __restrict__ dout3,
                                                     const float* __restrict__ dxbar, float* __restrict__ dx1,
                                                     float* __restrict__ dx2, float* __restrict__ part, int C, int T,
                                                     int V, int ld, int s3, DropArgs dr) {
  __shared__ float dxb[32];
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int c = (int)(plane % C);
  const int L = T * V;
  const bool drop = dr.thresh != 0;
  const unsigned long long dstep = (drop && dr.step) ? (unsigned long long)*dr.step : 0ull;
  const float a1 = s1 ? s1[c] : 1.f, b1 = s1 ? h1[c] : 0.f;
  const float a2 = s2 ? s2[c] : 1.f, b2 = s2 ? h2[c] : 0.f;
  if (lane < V) dxb[lane] = dxbar ? dxbar[(size_t)plane * ld + lane] / (float)T : 0.f;
  wave_lds_sync();
  const float* __restrict__ p1 = x1 + (size_t)plane * L;
  const float* __restrict__ p2 = x2 ? x2 + (size_t)plane * L : nullptr;
  const float* __restrict__ pg = (dout && s3 != 3) ? dout + (size_t)plane * L : nullptr;
  const float gpl = s3 == 3 ? dout[plane] / (float)L : 0.f;       // s3 = 3: dout is (n*C), the gradient of the plane mean
  const float* __restrict__ pg2 = dout2 ? dout2 + (size_t)plane * L : nullptr;
  // s3 = 2: dout3 is the gradient of the even-frame copy (n, C, ceil(T/2), V): it reaches the even frames only
  const float* __restrict__ pg3 = dout3 ? dout3 + (size_t)plane * (s3 == 2 ? ((T + 1) >> 1) * V : L) : nullptr;
  float* __restrict__ o1 = dx1 + (size_t)plane * L;
  float* __restrict__ o2 = dx2 ? dx2 + (size_t)plane * L : nullptr;
  float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
  int v = lane % V, t = lane / V;        // joint / frame of element `lane`; advance by 64 % V / 64 / V per iteration
  const int step = 64 % V, tstep = 64 / V;
#pragma unroll 5
  for (int i = lane; i < L; i += 64) {
    const float xa = p1[i];
    const float xb = p2 ? p2[i] : 0.f;
    const float pre1 = fmaf(xa, a1, b1);
    float pre = (relu & 2) ? fmaxf(pre1, 0.f) : pre1;
    const float dm = drop ? drop_mult(dr, dstep, (unsigned long long)plane * L + i) : 1.f;     // the forward's dropout multiplier
    pre *= dm;
    if (p2) pre += fmaf(xb, a2, b2);
    float g = s3 == 3 ? gpl : (pg ? pg[i] : 0.f);
    if (pg2) g += pg2[i];              // (a + b) + c, then the time-mean term: the order dsgcn_add3 + this kernel had
    if (pg3) {
      if (s3 == 2) { if (!(t & 1)) g += pg3[(t >> 1) * V + v]; }
      else g += pg3[i];
    }
    g += dxb[v];
    if ((relu & 1) && !(pre > 0.f)) g = 0.f;
    const float g1 = (((relu & 2) && !(pre1 > 0.f)) ? 0.f : g) * dm;     // gradient of the first term (through its dropout)
    o1[i] = g1 * a1;
    if (o2) o2[i] = g * a2;
    u0 = fmaf(g1, xa, u0);
    u1 += g;
    u2 = fmaf(g, xb, u2);
    u3 += g1;
    v += step;
    t += tstep;
    if (v >= V) { v -= V; ++t; }
  }
  if (part) {
    u0 = wave_sum(u0);
    u1 = wave_sum(u1);
    u2 = wave_sum(u2);
    u3 = wave_sum(u3);
    if (lane == 0) {
      part[(size_t)plane * 4 + 0] = u0;
      part[(size_t)plane * 4 + 1] = u1;
      part[(size_t)plane * 4 + 2] = u2;
      part[(size_t)plane * 4 + 3] = u3;
    }
  }
}

// 16-byte form of k_fuse_out_bwd for planes of T*V % 4 == 0: every stream moves as float4, the loads of an iteration
// are all issued before its arithmetic, absent streams are compile-time (no predicated loads: those become branches and
// serial waits).  NG = number of full-size gradient streams (1..3); S3 = the even-frame third stream of fuse_out_fwd2.
template <int NG, bool S3, bool X2>
__global__ __launch_bounds__(64) void k_fuse_out_bwd4(const float* __restrict__ x1, const float* __restrict__ s1,
                                                      const float* __restrict__ h1, const float* __restrict__ x2,
                                                      const float* __restrict__ s2, const float* __restrict__ h2,
                                                      int relu, const float* __restrict__ dout,
                                                      const float* __restrict__ dout2, const float* __restrict__ dout3,
                                                      const float* __restrict__ dxbar, float* __restrict__ dx1,
                                                      float* __restrict__ dx2, float* __restrict__ part, int C, int T,
                                                      int V, int ld, DropArgs dr) {
  __shared__ float dxb[32];
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int c = (int)(plane % C);
  const int L = T * V, L4 = L >> 2;
  const bool drop = dr.thresh != 0;
  const unsigned long long dstep = (drop && dr.step) ? (unsigned long long)*dr.step : 0ull;
  const float a1 = s1 ? s1[c] : 1.f, b1 = s1 ? h1[c] : 0.f;
  const float a2 = s2 ? s2[c] : 1.f, b2 = s2 ? h2[c] : 0.f;
  if (lane < V) dxb[lane] = dxbar ? dxbar[(size_t)plane * ld + lane] / (float)T : 0.f;
  wave_lds_sync();
  const f32x4* __restrict__ p1 = reinterpret_cast<const f32x4*>(x1 + (size_t)plane * L);
  const f32x4* __restrict__ p2 = X2 ? reinterpret_cast<const f32x4*>(x2 + (size_t)plane * L) : nullptr;
  // NG = 0: the gradient is one value per plane (dout (n*C): the backward of the plane mean, already divided by T*V)
  const f32x4* __restrict__ pg = NG >= 1 ? reinterpret_cast<const f32x4*>(dout + (size_t)plane * L) : nullptr;
  const float gplane = NG == 0 ? dout[plane] / (float)L : 0.f;
  const f32x4* __restrict__ pg2 = NG >= 2 ? reinterpret_cast<const f32x4*>(dout2 + (size_t)plane * L) : nullptr;
  const f32x4* __restrict__ pg3 = (NG >= 3 && !S3) ? reinterpret_cast<const f32x4*>(dout3 + (size_t)plane * L) : nullptr;
  const float* __restrict__ ps3 = S3 ? dout3 + (size_t)plane * (((T + 1) >> 1) * V) : nullptr;
  f32x4* __restrict__ o1 = reinterpret_cast<f32x4*>(dx1 + (size_t)plane * L);
  f32x4* __restrict__ o2 = X2 ? reinterpret_cast<f32x4*>(dx2 + (size_t)plane * L) : nullptr;
  const float invV = 1.f / (float)V;
  float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
  for (int i = lane; i < L4; i += 64) {
    const f32x4 xa = p1[i];
    f32x4 xb = {0.f, 0.f, 0.f, 0.f}, g = {gplane, gplane, gplane, gplane};
    if constexpr (NG >= 1) g = pg[i];
    if constexpr (X2) xb = p2[i];
    if constexpr (NG >= 2) { const f32x4 w = pg2[i]; g.x += w.x; g.y += w.y; g.z += w.z; g.w += w.w; }
    if constexpr (NG >= 3 && !S3) { const f32x4 w = pg3[i]; g.x += w.x; g.y += w.y; g.z += w.z; g.w += w.w; }
    int t, v;
    divmod_small(4 * i, V, invV, t, v);
    f32x4 r1, r2;
    float dm[4] = {1.f, 1.f, 1.f, 1.f};            // the forward's dropout multipliers of these four elements
    if (drop) {
      unsigned rw[4];
      drop_words(dr, dstep, (unsigned long long)plane * L4 + i, rw);
#pragma unroll
      for (int k = 0; k < 4; ++k) dm[k] = rw[k] >= dr.thresh ? dr.inv : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gg = g[k];
      if constexpr (S3) { if (!(t & 1)) gg += ps3[(t >> 1) * V + v]; }
      gg += dxb[v];
      const float pre1 = fmaf(xa[k], a1, b1);
      float pre = ((relu & 2) ? fmaxf(pre1, 0.f) : pre1) * dm[k];
      if constexpr (X2) pre += fmaf(xb[k], a2, b2);
      if ((relu & 1) && !(pre > 0.f)) gg = 0.f;
      const float g1 = (((relu & 2) && !(pre1 > 0.f)) ? 0.f : gg) * dm[k];
      r1[k] = g1 * a1;
      r2[k] = gg * a2;
      u0 = fmaf(g1, xa[k], u0);
      u1 += gg;
      u2 = fmaf(gg, xb[k], u2);
      u3 += g1;
      if (++v == V) { v = 0; ++t; }
    }
    o1[i] = r1;
    if constexpr (X2) o2[i] = r2;
  }
  if (part) {
    u0 = wave_sum(u0);
    u1 = wave_sum(u1);
    u2 = wave_sum(u2);
    u3 = wave_sum(u3);
    if (lane == 0) {
      part[(size_t)plane * 4 + 0] = u0;
      part[(size_t)plane * 4 + 1] = u1;
      part[(size_t)plane * 4 + 2] = u2;
      part[(size_t)plane * 4 + 3] = u3;
    }
  }
}

int g_fo_vec = 1;        // 16-byte backward where the plane allows it (lab A/B: 0 = the scalar form everywhere)

}  // namespace

namespace {

// p <= 0 (or no record): dropout off
DropArgs drop_args(const dsgcn_dropout* d) {
  DropArgs a = {};
  if (d && d->p > 0.f) {
    a.step = d->step; a.seed = d->seed; a.call = d->call;
    double t = (double)d->p * 4294967296.0;
    a.thresh = t >= 4294967295.0 ? 4294967295u : (t < 1.0 ? 1u : (unsigned)t);
    a.inv = 1.f / (1.f - d->p);
  }
  return a;
}

__global__ __launch_bounds__(256) void k_dropout_mask(float* __restrict__ mask, long numel, DropArgs dr) {
  const unsigned long long step = dr.step ? (unsigned long long)*dr.step : 0ull;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < numel; e += (long)gridDim.x * 256)
    mask[e] = dr.thresh ? drop_mult(dr, step, (unsigned long long)e) : 1.f;
}

}  // namespace

extern "C" {

// The block output with dropout on its first term: out = relu?( D * relu2?(x1*s1+h1) + (x2*s2+h2 | x2) ), D = keep / (1 - p)
// from the counter-based generator of csrc/dropout.h (reference: the Dropout behind the temporal unit's BatchNorm,
// tcn.py:30,33, MSTCN msg3d_utils.py:141-146, then dgstgcn.py:63-65 / stgcn.py:64-66 add the residual and apply ReLU).
// Any of out / out_s2 / xbar / pmean may be NULL (pmean (n*C): the plane means for a pooling head, the last block);
// d NULL or d->p <= 0: no dropout — the plain fuse_out.
int dsgcn_fuse_out_fwd_drop(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                            const float* h2, int relu, float* out, float* out_s2, float* xbar, float* pmean, int n, int C,
                            int T, int V, int xbar_ld, const dsgcn_dropout* d, void* stream) {
  if (!x1 || (!out && !pmean) || n <= 0 || C <= 0 || T <= 0 || V <= 0 || V > 32 || (s1 && !h1) || (s2 && !h2)) return DSGCN_EINVAL;
  if ((out_s2 || xbar) && !out) return DSGCN_EINVAL;
  if (xbar && (xbar_ld < V || xbar_ld > 64)) return DSGCN_EINVAL;
  if (d && !(d->p >= 0.f && d->p < 1.f)) return DSGCN_EINVAL;
  const int vec = ((T * V) % 4 == 0) ? 1 : 0;
  const size_t lds = (xbar || out_s2) ? (size_t)T * V * sizeof(float) : 0;
  if (lds > 64 * 1024) return DSGCN_EUNSUPPORTED;
  hipLaunchKernelGGL(k_fuse_out_fwd, dim3((unsigned)((long)n * C)), dim3(64), lds, (hipStream_t)stream, x1, s1, h1, x2,
                     s2, h2, relu, out, xbar, C, T, V, vec, xbar ? xbar_ld : V, out_s2, pmean, drop_args(d));
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_fuse_out_fwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                       const float* h2, int relu, float* out, float* xbar, int n, int C, int T, int V, int xbar_ld,
                       void* stream) {
  if (!out) return DSGCN_EINVAL;
  return dsgcn_fuse_out_fwd_drop(x1, s1, h1, x2, s2, h2, relu, out, nullptr, xbar, nullptr, n, C, T, V, xbar_ld, nullptr, stream);
}

// out_s2 (NULL or (n, C, ceil(T/2), V)): the even frames of `out` as a second, contiguous output — the operand of a
// stride-2 block's 1x1 residual conv (reference: unit_tcn(kernel_size=1, stride=2) of the block residual, dgstgcn.py:35-40)
int dsgcn_fuse_out_fwd2(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                        const float* h2, int relu, float* out, float* out_s2, float* xbar, int n, int C, int T, int V,
                        int xbar_ld, void* stream) {
  if (!out) return DSGCN_EINVAL;
  return dsgcn_fuse_out_fwd_drop(x1, s1, h1, x2, s2, h2, relu, out, out_s2, xbar, nullptr, n, C, T, V, xbar_ld, nullptr, stream);
}

// The LAST block's output is only ever averaged over its (T, V) planes by the head (simple_head.py:88-93): pmean (n, C) =
// plane means of the block output, which is never written (one 82 MB write, its read by the pooling launch, and in the
// backward the broadcast of the pooled gradient and its read are gone).
int dsgcn_fuse_out_pool_fwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                            const float* h2, int relu, float* pmean, int n, int C, int T, int V, void* stream) {
  if (!pmean) return DSGCN_EINVAL;
  return dsgcn_fuse_out_fwd_drop(x1, s1, h1, x2, s2, h2, relu, nullptr, nullptr, nullptr, pmean, n, C, T, V, V, nullptr, stream);
}

// Backward of dsgcn_fuse_out_fwd_drop (the same dropout record: the masks are regenerated).  part: (n*C, 4) per-plane
// [sum dv1*x1, sum dv, sum dv*x2, sum dv1]; dout or dxbar may be NULL (treated as zero).  dout2 / dout3 (NULL or like dout):
// the gradients of the other consumers of `out` (the next block reads its input three times) — summed while loading,
// (dout + dout2) + dout3.  stride3 = 2: dout3 is the gradient of the even-frame output, (n, C, ceil(T/2), V);
// stride3 = 3: dout is (n*C), the gradient of the plane means (dsgcn_fuse_out_pool_fwd), dout2 / dout3 / dxbar NULL.
int dsgcn_fuse_out_bwd_drop(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                            const float* h2, int relu, const float* dout, const float* dout2, const float* dout3,
                            int stride3, const float* dxbar, float* dx1, float* dx2, float* part, int n, int C, int T, int V,
                            int xbar_ld, const dsgcn_dropout* d, void* stream) {
  if (!x1 || !dx1 || n <= 0 || C <= 0 || T <= 0 || V <= 0 || V > 32 || (x2 && !dx2)) return DSGCN_EINVAL;
  if (dxbar && xbar_ld < V) return DSGCN_EINVAL;
  if ((dout2 || dout3) && !dout) return DSGCN_EINVAL;
  if (stride3 < 1 || stride3 > 3 || (stride3 == 3 && (!dout || dout2 || dout3 || dxbar))) return DSGCN_EINVAL;
  if (d && !(d->p >= 0.f && d->p < 1.f)) return DSGCN_EINVAL;
  const DropArgs dr = drop_args(d);
  const dim3 grid((unsigned)((long)n * C)), blk(64);
  hipStream_t st = (hipStream_t)stream;
  if (stride3 == 3) {
    if ((T * V) % 4 == 0) {
      if (x2)
        hipLaunchKernelGGL((k_fuse_out_bwd4<0, false, true>), grid, blk, 0, st, x1, s1, h1, x2, s2, h2, relu, dout,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, dx1, dx2, part, C, T, V, V, dr);
      else
        hipLaunchKernelGGL((k_fuse_out_bwd4<0, false, false>), grid, blk, 0, st, x1, s1, h1, x2, s2, h2, relu, dout,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, dx1, dx2, part, C, T, V, V, dr);
    } else {
      hipLaunchKernelGGL(k_fuse_out_bwd, grid, blk, 0, st, x1, s1, h1, x2, s2, h2, relu, dout, (const float*)nullptr,
                         (const float*)nullptr, (const float*)nullptr, dx1, dx2, part, C, T, V, V, 3, dr);
    }
    DSGCN_LAUNCH_CHECK();
    return 0;
  }
  // the summation order of the scalar form: ((dout + dout2) + dout3) + the time-mean term
  if (g_fo_vec && dout && (T * V) % 4 == 0 && !(dout3 && !dout2 && stride3 == 1)) {
    const int ng = dout3 && stride3 == 1 ? 3 : (dout2 ? 2 : 1);
    const bool s3 = dout3 && stride3 == 2;
#define FO4(NGV, S3V, X2V) hipLaunchKernelGGL((k_fuse_out_bwd4<NGV, S3V, X2V>), grid, blk, 0, st, x1, s1, h1, x2, s2, h2, relu, \
                                              dout, dout2, dout3, dxbar, dx1, dx2, part, C, T, V, xbar_ld, dr)
    if (x2) {
      if (s3) { if (ng == 2) FO4(2, true, true); else FO4(1, true, true); }
      else if (ng == 3) FO4(3, false, true);
      else if (ng == 2) FO4(2, false, true);
      else FO4(1, false, true);
    } else {
      if (s3) { if (ng == 2) FO4(2, true, false); else FO4(1, true, false); }
      else if (ng == 3) FO4(3, false, false);
      else if (ng == 2) FO4(2, false, false);
      else FO4(1, false, false);
    }
#undef FO4
    DSGCN_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_fuse_out_bwd, grid, blk, 0, st, x1, s1, h1, x2, s2, h2, relu, dout, dout2, dout3, dxbar, dx1, dx2,
                     part, C, T, V, xbar_ld, stride3, dr);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// dpmean (n, C): gradient of the plane means; dx1 / dx2 / part as in dsgcn_fuse_out_bwd.
int dsgcn_fuse_out_pool_bwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                            const float* h2, int relu, const float* dpmean, float* dx1, float* dx2, float* part, int n,
                            int C, int T, int V, void* stream) {
  if (!dpmean) return DSGCN_EINVAL;
  return dsgcn_fuse_out_bwd_drop(x1, s1, h1, x2, s2, h2, relu, dpmean, nullptr, nullptr, 3, nullptr, dx1, dx2, part, n, C, T, V,
                                 V, nullptr, stream);
}

int dsgcn_fuse_out_bwd3(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                        const float* h2, int relu, const float* dout, const float* dout2, const float* dout3,
                        const float* dxbar, float* dx1, float* dx2, float* part, int n, int C, int T, int V, int xbar_ld,
                        void* stream) {
  return dsgcn_fuse_out_bwd_drop(x1, s1, h1, x2, s2, h2, relu, dout, dout2, dout3, 1, dxbar, dx1, dx2, part, n, C, T, V,
                                 xbar_ld, nullptr, stream);
}

// stride3 = 2: dout3 is the gradient of dsgcn_fuse_out_fwd2's even-frame output, (n, C, ceil(T/2), V)
int dsgcn_fuse_out_bwd3s(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                         const float* h2, int relu, const float* dout, const float* dout2, const float* dout3, int stride3,
                         const float* dxbar, float* dx1, float* dx2, float* part, int n, int C, int T, int V, int xbar_ld,
                         void* stream) {
  if (stride3 != 1 && stride3 != 2) return DSGCN_EINVAL;
  return dsgcn_fuse_out_bwd_drop(x1, s1, h1, x2, s2, h2, relu, dout, dout2, dout3, stride3, dxbar, dx1, dx2, part, n, C, T, V,
                                 xbar_ld, nullptr, stream);
}

// mask (numel) = the multipliers dsgcn_fuse_out_fwd_drop applies to the first term of a tensor of numel elements under
// record d: 1/(1-p) or 0 (tests and debugging: the product never materialises it)
int dsgcn_dropout_mask(float* mask, long numel, const dsgcn_dropout* d, void* stream) {
  if (!mask || numel <= 0 || (d && !(d->p >= 0.f && d->p < 1.f))) return DSGCN_EINVAL;
  const long blocks = (numel + 255) / 256;
  hipLaunchKernelGGL(k_dropout_mask, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, mask,
                     numel, drop_args(d));
  DSGCN_LAUNCH_CHECK();
  return 0;
}

#ifdef DSGCN_LAB
int dsgcn_fuse_out_tuning(int key, int value) {      // key 0: 16-byte backward on (1) / off (0)
  if (key != 0) return DSGCN_EINVAL;
  g_fo_vec = value;
  return 0;
}
#endif

int dsgcn_fuse_out_bwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                       const float* h2, int relu, const float* dout, const float* dxbar, float* dx1, float* dx2,
                       float* part, int n, int C, int T, int V, int xbar_ld, void* stream) {
  return dsgcn_fuse_out_bwd3(x1, s1, h1, x2, s2, h2, relu, dout, nullptr, nullptr, dxbar, dx1, dx2, part, n, C, T, V,
                             xbar_ld, stream);
}

}  // extern "C"

// Gradient packing for the data-parallel exchange: count tensors (device table of source pointers, destination
// offsets and lengths) -> one flat buffer, one launch (the per-tensor path costs ~200 blit kernels per step).
namespace {
__global__ __launch_bounds__(256) void k_pack(const float* const* __restrict__ src, const long* __restrict__ off,
                                              const int* __restrict__ numel, float* __restrict__ dst) {
  const int i = blockIdx.x;
  const float* __restrict__ s = src[i];
  float* __restrict__ d = dst + off[i];
  const int n = numel[i];
  for (int j = blockIdx.y * 256 + threadIdx.x; j < n; j += gridDim.y * 256) d[j] = s[j];
}
}  // namespace

namespace {
// the same with 64-bit lengths and a NULL source meaning "zero-fill": one launch writes the WHOLE flat buffer (parameters
// that received no gradient included), so no fill pass and no length-conversion launch precede it
__global__ __launch_bounds__(256) void k_pack_fill(const float* const* __restrict__ src, const long* __restrict__ off,
                                                   const long* __restrict__ numel, float* __restrict__ dst) {
  const int i = blockIdx.x;
  const float* __restrict__ s = src[i];
  float* __restrict__ d = dst + off[i];
  const long n = numel[i];
  if (s) {
    for (long j = blockIdx.y * 256 + threadIdx.x; j < n; j += gridDim.y * 256) d[j] = s[j];
  } else {
    for (long j = blockIdx.y * 256 + threadIdx.x; j < n; j += gridDim.y * 256) d[j] = 0.f;
  }
}
}  // namespace

extern "C" int dsgcn_pack_fill(const float* const* src_table, const long* dst_offsets, const long* numels, int count,
                               float* dst, void* stream) {
  if (!src_table || !dst_offsets || !numels || !dst || count <= 0) return DSGCN_EINVAL;
  hipLaunchKernelGGL(k_pack_fill, dim3((unsigned)count, 8), dim3(256), 0, (hipStream_t)stream, src_table, dst_offsets,
                     numels, dst);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

extern "C" int dsgcn_pack(const float* const* src_table, const long* dst_offsets, const int* numels, int count,
                          float* dst, void* stream) {
  if (!src_table || !dst_offsets || !numels || !dst || count <= 0) return DSGCN_EINVAL;
  hipLaunchKernelGGL(k_pack, dim3((unsigned)count, 8), dim3(256), 0, (hipStream_t)stream, src_table, dst_offsets, numels,
                     dst);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// out = a + b (+ c): the gradients that reach one block input from its consumers, summed in one pass
// (autograd would add them pairwise: two launches and six plane accesses instead of four).
namespace {
__global__ __launch_bounds__(256) void k_add3(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                              const f32x4* __restrict__ c, f32x4* __restrict__ out, long n4,
                                              const float* as, const float* bs, const float* cs, float* os, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    f32x4 v = a[i];
    const f32x4 w = b[i];
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    if (c) { const f32x4 u = c[i]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    out[i] = v;
  }
  if (blockIdx.x == 0) {                      // tail (n % 4 elements)
    const long j = n4 * 4 + threadIdx.x;
    if (j < n) os[j] = as[j] + bs[j] + (cs ? cs[j] : 0.f);
  }
}
}  // namespace

extern "C" int dsgcn_add3(const float* a, const float* b, const float* c, float* out, long n, void* stream) {
  if (!a || !b || !out || n <= 0) return DSGCN_EINVAL;
  const long n4 = n / 4;
  const unsigned blocks = (unsigned)((n4 + 255) / 256);
  hipLaunchKernelGGL(k_add3, dim3(blocks ? blocks : 1), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4*>(a), reinterpret_cast<const f32x4*>(b),
                     reinterpret_cast<const f32x4*>(c), reinterpret_cast<f32x4*>(out), n4, a, b, c, out, n);
  DSGCN_LAUNCH_CHECK();
  return 0;
}
