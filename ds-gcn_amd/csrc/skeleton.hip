// Skeleton input pipeline, per-element half: one launch turns raw clips resident in HBM into the network
// input (N, clips, M, T, V, C_out).  Replaces, per sample, the numpy transforms of the reference's data-loader workers:
//   pyskl/datasets/pipelines/pose_related.py:250-336 (PreNormalize3D: drop empty frames, swap persons, centre + mask,
//   spine / shoulder alignment), 144-178 (RandomRot), 340-442 (JointToBone / ToMotion / GenSkeFeat: j, b, jm, bm),
//   19-54 (PoseDecode: frame gather), 468-518 (FormatGCNInput: person padding, clip layout),
//   pipelines/sampling.py:10-192 (UniformSample: the frame indices arrive in f0 / f1).
//   pipelines/augmentations.py:21-116 (PoseCompact: the shift by the box origin; the box itself is a host decision).
// The per-clip DECISIONS (kept frames, person order, body centre / box origin, total linear map = rotation x alignment,
// sampled frame indices) are made on the host with the reference's RNG draws (ds-gcn_amd/pipeline.py: SkeletonBatcher.plan) and arrive
// as a few hundred bytes per clip; everything proportional to the clip size happens here.  Pure gather / byte-moving work:
// HBM-bound, one thread per output joint (all its feature channels), coalesced stores.
#include "common.h"

namespace {

struct SkArgs {
  const float* raw; const long* offset; const int* M; const int* T; const int* flags;
  const int* f0; const int* f1; const float* center; const float* matrix; const int* parent;
  float* out;
  int N, clips, Mout, clip_len, V, C, nfeat, fmask, scored, loop;
};

// joint v of person pm, original frame f of clip n, after centre / mask / linear map; score channel passes through
// masked: 0 plain shift, 1 PreNormalize3D (a joint that is zero in every coordinate stays zero), 2 PoseCompact (every
// coordinate that is zero stays zero: augmentations.py:104-105 shifts `kp_x[kp_x != 0]` and `kp_y[kp_y != 0]` separately)
__device__ __forceinline__ void sk_joint(const SkArgs& a, const float* __restrict__ base, int T, int pm, int f, int v,
                                         int masked, const float* c, const float* m, float (&o)[3]) {
  const float* p = base + (((size_t)pm * T + f) * a.V + v) * a.C;
  float x[3] = {p[0], p[1], a.C == 3 ? p[2] : 0.f};
  if (masked == 2) {
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = x[i] != 0.f ? x[i] - c[i] : x[i];
  } else if (masked) {
    const bool nz = x[0] != 0.f || x[1] != 0.f || x[2] != 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = nz ? x[i] - c[i] : 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] -= c[i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = fmaf(m[3 * i + 2], x[2], fmaf(m[3 * i + 1], x[1], m[3 * i] * x[0]));
}

__global__ __launch_bounds__(256) void k_skeleton_prep(SkArgs a) {
  const long total = (long)a.N * a.clips * a.Mout * a.clip_len * a.V;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int v = (int)(idx % a.V);
  long r = idx / a.V;
  const int t = (int)(r % a.clip_len); r /= a.clip_len;
  const int mo = (int)(r % a.Mout); r /= a.Mout;
  const int clip = (int)(r % a.clips);
  const int n = (int)(r / a.clips);
  const int Cout = a.C * a.nfeat;
  float* __restrict__ o = a.out + idx * Cout;
  const int M = a.M[n], T = a.T[n], fl = a.flags[n];
  int m_src = mo;
  if (a.loop && M < a.Mout) {
    // FormatGCNInput(mode='loop') with fewer persons than slots: EVERY slot after the first repeats the first formatted
    // person — the reference assigns keypoint[1:] = keypoint[0] after padding, overwriting a real second person too
    // (pose_related.py:492-497)
    if (mo >= 1) m_src = 0;
  } else if (mo >= M) {
    for (int i = 0; i < Cout; ++i) o[i] = 0.f;
    return;
  }
  const int pm = (fl & 1) ? 1 - m_src : m_src;    // swap: the formatted person 0 is raw person 1
  const int masked = (fl & 8) ? 2 : ((fl & 2) ? 1 : 0);
  const float* c = a.center + 3 * n;
  const float* m = a.matrix + 9 * n;
  const float* base = a.raw + a.offset[n];
  const int F = a.clips * a.clip_len;
  const int fa = a.f0[(size_t)n * F + clip * a.clip_len + t];
  const int fb = a.f1[(size_t)n * F + clip * a.clip_len + t];
  const int par = a.parent[v];
  float j0[3], p0[3] = {0.f, 0.f, 0.f}, j1[3] = {0.f, 0.f, 0.f}, p1[3] = {0.f, 0.f, 0.f};
  sk_joint(a, base, T, pm, fa, v, masked, c, m, j0);
  bool need_b = false, need_m = false;
  for (int i = 0; i < a.nfeat; ++i) {
    const int code = (a.fmask >> (2 * i)) & 3;
    need_b |= (code & 1) != 0;
    need_m |= (code & 2) != 0;
  }
  if (need_b) sk_joint(a, base, T, pm, fa, par, masked, c, m, p0);
  if (need_m && fb >= 0) {
    sk_joint(a, base, T, pm, fb, v, masked, c, m, j1);
    if (need_b) sk_joint(a, base, T, pm, fb, par, masked, c, m, p1);
  }
  // fl bit 2: fp16-sourced clip without rotation — the reference builds bones and joint motion by numpy arithmetic on the
  // fp16 arrays (differences and score sums round to fp16) and stores bones in an fp32 array (bone motion: fp32 arithmetic)
  const bool h16 = (fl & 4) != 0;
  for (int i = 0; i < a.nfeat; ++i) {
    const int code = (a.fmask >> (2 * i)) & 3;    // 0 j, 1 b, 2 jm, 3 bm
    float cur[3], nxt[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      cur[k] = (code & 1) ? j0[k] - p0[k] : j0[k];
      nxt[k] = (code & 1) ? j1[k] - p1[k] : j1[k];
      if ((code & 1) && h16) { cur[k] = (float)(_Float16)cur[k]; nxt[k] = (float)(_Float16)nxt[k]; }
    }
    if ((code & 1) && a.scored) {
      float s0 = j0[2] + p0[2], s1 = j1[2] + p1[2];
      if (h16) { s0 = (float)(_Float16)s0; s1 = (float)(_Float16)s1; }
      cur[2] = 0.5f * s0; nxt[2] = 0.5f * s1;
    }
    float val[3];
    if (code & 2) {
      const bool r16 = h16 && code == 2;          // joint motion stays in the fp16 array; bone motion is fp32 arithmetic
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float d = nxt[k] - cur[k];
        if (r16) d = (float)(_Float16)d;
        val[k] = fb >= 0 ? d : 0.f;
      }
      if (a.scored) {
        float sm = cur[2] + nxt[2];
        if (r16) sm = (float)(_Float16)sm;
        val[2] = fb >= 0 ? 0.5f * sm : 0.f;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) val[k] = cur[k];
    }
    for (int k = 0; k < a.C; ++k) o[i * a.C + k] = val[k];
  }
}

}  // namespace

extern "C" int dsgcn_skeleton_prep(const float* raw, const long* offset, const int* M, const int* T, const int* flags,
                                   const int* f0, const int* f1, const float* center, const float* matrix,
                                   const int* parent, float* out, int N, int clips, int Mout, int clip_len, int V, int C,
                                   int nfeat, int fmask, int scored, int loop, void* stream) {
  if (!raw || !offset || !M || !T || !flags || !f0 || !f1 || !center || !matrix || !parent || !out) return DSGCN_EINVAL;
  if (N <= 0 || clips <= 0 || Mout <= 0 || clip_len <= 0 || V <= 0 || (C != 2 && C != 3) || nfeat < 1 || nfeat > 4)
    return DSGCN_EINVAL;
  SkArgs a{raw, offset, M, T, flags, f0, f1, center, matrix, parent, out, N, clips, Mout, clip_len, V, C, nfeat, fmask,
           scored, loop};
  const long total = (long)N * clips * Mout * clip_len * V;
  hipLaunchKernelGGL(k_skeleton_prep, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  DSGCN_LAUNCH_CHECK();
  return 0;
}
