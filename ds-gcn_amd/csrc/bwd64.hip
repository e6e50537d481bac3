// K-C backward in ONE pass for narrow convs (Ci, Co <= 64, one input stream, stride 1): data gradient and weight
// gradient share the loads of gz, z and the input (reference ops: the backward of the 1x1 Conv2d + BatchNorm + ReLU chains
// cited in pwconv.hip — gcn.py:2165-2169,2209-2215,2363-2365, tcn.py:379-404,422,427).
//
//   dz_eff[co,pos] = gz + A0[co] + B0[co]*z                    v[ci,pos] = relu?(x*s1[ci] + h1[ci])
//   dW[co,ci] = sum_pos dz_eff * v      db[co] = sum_pos dz_eff
//   dv[ci,pos] = sum_co W[co,ci] dz_eff[co,pos]     dx = dv * 1[pre>0] * s1     ipart: sum dvm*x, sum dvm
//
// At these widths both separate kernels are HBM-bound and each reads gz, z and x: 367 MB for a 64 -> 64 conv at 128
// samples where one pass needs 262 MB ("fuse dgrad + wgrad into one pass").  The weight gradient contracts
// over positions, so both of its operands must be staged through LDS as [channel][position] tiles (as in wgrad.hip);
// the data gradient then takes its B operand (dz_eff, k = co, j = position) from the SAME tile and its A operand (W^T)
// from a 64x64 weight image in LDS.  Work unit = (sample, 64 positions); K-splits over the grid, every split writes
// its partial dW / db / ipart row (ordered sums later: dsgcn_colsum).
#include "common.h"

namespace {

constexpr int BF_NT = 256, BF_KC = 64, BF_LS = BF_KC + 2, BF_Q = BF_KC / 4, BF_J = 64 * BF_Q / BF_NT;   // 4 slots
constexpr int BF_OOB = 0x7ffffff0;

struct BfArgs {
  const float* x1; const float* s1; const float* h1;
  const float* x2; const float* s2; const float* h2; int relu;
  const float* w;                                   // (Co, Ci)
  const float* z; const float* gz; const float* A0; const float* B0;
  float* dx; float* dx2; float* dwp; float* dbp; float* ipart;  // ipart (splits, Ci, 3) or NULL
  int pstride, n, Ci, Co, L, cpn, total_chunks, cps;
};

typedef float bf_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bf_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bf_load(__amdgpu_buffer_rsrc_t r, int voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ int bf_row32(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

#ifdef DSGCN_LAB
// wall-clock stamps (10 ns) of workgroup 0, thread 0: start, tables ready, then per unit (committed + barrier, products +
// barrier, finished), partial rows written; [63] = count (dsgcn_bwd64_phases)
__device__ long long g_bf_stamp[64];
#define BF_STAMP() do { if (blockIdx.x == 0 && threadIdx.x == 0 && nst < 62) g_bf_stamp[nst++] = wall_clock64(); } while (0)
#else
#define BF_STAMP() do {} while (0)
#endif

// HASC: batch-statistics terms (A0, B0);  AFF: the input carries an affine and / or a ReLU;  HAS2: second input stream.
template <bool HASC, bool AFF, bool HAS2>
__global__ __launch_bounds__(BF_NT, 2) void k_bwd64(BfArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef DSGCN_LAB
  int nst = 0;
#endif
  BF_STAMP();
  float* Ds = lds;                                  // [64][LS] dz_eff (zero rows >= Co)
  float* Xs = lds + 64 * BF_LS;                     // [64][LS] the ACTIVATED virtual input v (zero rows >= Ci)
  float* Os = lds + 128 * BF_LS;                    // [64][LS] dv tile of the unit (before mask / scale)
  float* Ws = lds + 192 * BF_LS;                    // [64 co][65] W[co][ci], zero padded
  bf_f2* Cs = reinterpret_cast<bf_f2*>(Ws + 64 * 65);      // [64] (A0, B0)
  f32x4* Ps = reinterpret_cast<f32x4*>(Cs + 64);           // [64] (s1, h1, s2, h2)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int Ci = a.Ci, Co = a.Co, L = a.L, L4 = L * 4;
  const int split = blockIdx.x;
  const int ch0 = split * a.cps, ch1 = min(a.total_chunks, ch0 + a.cps);

  for (int i = tid; i < 64 * 65; i += BF_NT) {
    const int co = i / 65, ci = i - co * 65;
    Ws[i] = (co < Co && ci < Ci) ? a.w[(size_t)co * Ci + ci] : 0.f;
  }
  if (tid < 64) {
    Cs[tid] = (HASC && tid < Co) ? bf_f2{a.A0[tid], a.B0[tid]} : bf_f2{0.f, 0.f};
    f32x4 p = {1.f, 0.f, 1.f, 0.f};
    if (AFF && tid < Ci) {
      if (a.s1) { p.x = a.s1[tid]; p.y = a.h1[tid]; }
      if (HAS2 && a.s2) { p.z = a.s2[tid]; p.w = a.h2[tid]; }
    }
    Ps[tid] = p;
  }
  const float lo = a.relu ? 0.f : -__builtin_inff();

  // staging slots: f = tid + 256*j -> row f / 16, positions 4*(f % 16) ..+3 of the 64-position unit.  The same
  // threads finish the data gradient of "their" elements after the MFMA phases (they still hold the raw inputs).
  const int col = (tid % BF_Q) * 4, row0 = tid / BF_Q;          // rows row0 + 16*j
  f32x4 gr[BF_J], zr[HASC ? BF_J : 1], xr[BF_J], yr[HAS2 ? BF_J : 1];
  f32x4 xk[AFF ? BF_J : 1], yk[HAS2 ? BF_J : 1];
  float dsum[BF_J], r0[BF_J], r1[BF_J], r2[BF_J];
#pragma unroll
  for (int j = 0; j < BF_J; ++j) { dsum[j] = 0.f; r0[j] = 0.f; r1[j] = 0.f; r2[j] = 0.f; }
  auto issue = [&](int ch) {
    const int n = ch / a.cpn, c0 = (ch - n * a.cpn) * BF_KC;
    const bool pv = c0 + col < L;                                // L % 4 == 0: a float4 is entirely in or out
    const __amdgpu_buffer_rsrc_t rg = bf_rsrc(a.gz + (size_t)n * Co * L, Co * L4);
    const __amdgpu_buffer_rsrc_t rz = bf_rsrc((HASC ? a.z : a.gz) + (size_t)n * Co * L, HASC ? Co * L4 : 0);
    const __amdgpu_buffer_rsrc_t rx = bf_rsrc(a.x1 + (size_t)n * Ci * L, Ci * L4);
    const __amdgpu_buffer_rsrc_t ry = bf_rsrc((HAS2 ? a.x2 : a.x1) + (size_t)n * Ci * L, HAS2 ? Ci * L4 : 0);
#pragma unroll
    for (int j = 0; j < BF_J; ++j) {
      const int row = row0 + 16 * j;
      const int vd = (pv && row < Co) ? (row * L + c0 + col) * 4 : BF_OOB;
      const int vx = (pv && row < Ci) ? (row * L + c0 + col) * 4 : BF_OOB;
      gr[j] = bf_load(rg, vd);
      if constexpr (HASC) zr[j] = bf_load(rz, vd);
      xr[j] = bf_load(rx, vx);
      if constexpr (HAS2) yr[j] = bf_load(ry, vx);
    }
  };
  auto commit = [&](int ch) {
    const int n = ch / a.cpn, c0 = (ch - n * a.cpn) * BF_KC;
    const bool pv = c0 + col < L;
#pragma unroll
    for (int j = 0; j < BF_J; ++j) {
      const int row = row0 + 16 * j;
      f32x4 d = gr[j];
      if constexpr (HASC) {
        const bf_f2 c = Cs[row];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] += fmaf(c.y, zr[j][e], c.x);
      }
      if (!(pv && row < Co)) d = f32x4{0.f, 0.f, 0.f, 0.f};
      dsum[j] += (d.x + d.y) + (d.z + d.w);
      bf_f2* dd = reinterpret_cast<bf_f2*>(Ds + row * BF_LS + col);
      dd[0] = bf_f2{d.x, d.y};
      dd[1] = bf_f2{d.z, d.w};
      f32x4 v = xr[j];
      if constexpr (AFF) {
        const f32x4 p = Ps[row];
        xk[j] = xr[j];
        if constexpr (HAS2) yk[j] = yr[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = fmaf(xr[j][e], p.x, p.y);
          if constexpr (HAS2) t += fmaf(yr[j][e], p.z, p.w);
          v[e] = fmaxf(t, lo);
        }
      }
      if (!(pv && row < Ci)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      bf_f2* dxs = reinterpret_cast<bf_f2*>(Xs + row * BF_LS + col);
      dxs[0] = bf_f2{v.x, v.y};
      dxs[1] = bf_f2{v.z, v.w};
    }
  };

  f32x16 accw, accw2, accd, accd2;
#pragma unroll
  for (int i = 0; i < 16; ++i) { accw[i] = 0.f; accw2[i] = 0.f; }
  // weight gradient: wave (mt, nt) owns the 32x32 block (co tile mt, ci tile nt) of dW
  // data gradient:   wave (cit, pt) owns (ci tile cit) x (position tile pt) of the unit's 64 x 64 dv
  // (Ci <= 32: only one ci tile exists — waves 0,1 take the two co tiles of dW, waves 2,3 the two position tiles of dv,
  // instead of two waves doing both jobs and two idling)
  const bool narrow_i = Ci <= 32;
  const int mt = narrow_i ? (wave & 1) : (wave >> 1), nt = narrow_i ? 0 : (wave & 1);
  const int cit = narrow_i ? 0 : (wave >> 1), pt = wave & 1;
  const float* Ap = Ds + (32 * mt + l31) * BF_LS + 2 * half;
  const float* Bp = Xs + (32 * nt + l31) * BF_LS + 2 * half;
  const int KSd = (Co + 1) >> 1;                                 // data-gradient k-steps (two output channels each)
  const bool wg_on = 32 * mt < Co && 32 * nt < Ci && !(narrow_i && wave >= 2);      // tiles that hold real channels
  const bool dg_on = 32 * cit < Ci && !(narrow_i && wave < 2);

  __syncthreads();
  BF_STAMP();
  if (ch0 < ch1) issue(ch0);
  for (int ch = ch0; ch < ch1; ++ch) {
    commit(ch);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    BF_STAMP();
    if (ch + 1 < ch1) issue(ch + 1);
    // ---- weight gradient ---- (two accumulators: a single one is a chain of dependent MFMAs, and only two waves
    // share a SIMD here)
    if (wg_on) {
      // operands of steps w+2, w+3 are read from LDS BEFORE the four products of steps w, w+1 are issued (pinned): left to
      // itself hipcc emits read, read, s_waitcnt lgkmcnt(0), four MFMAs — every pair of steps pays an LDS round trip in
      // series with its 256 matrix cycles (round-5 disassembly), and the co-resident workgroup does not always cover it
      bf_f2 av0 = *reinterpret_cast<const bf_f2*>(Ap), bv0 = *reinterpret_cast<const bf_f2*>(Bp);
      bf_f2 av1 = *reinterpret_cast<const bf_f2*>(Ap + 4), bv1 = *reinterpret_cast<const bf_f2*>(Bp + 4);
#pragma unroll
      for (int w = 0; w < BF_Q; w += 2) {
        bf_f2 an0 = av0, bn0 = bv0, an1 = av1, bn1 = bv1;
        if (w + 2 < BF_Q) {
          an0 = *reinterpret_cast<const bf_f2*>(Ap + 4 * (w + 2)); bn0 = *reinterpret_cast<const bf_f2*>(Bp + 4 * (w + 2));
          an1 = *reinterpret_cast<const bf_f2*>(Ap + 4 * (w + 3)); bn1 = *reinterpret_cast<const bf_f2*>(Bp + 4 * (w + 3));
        }
        __builtin_amdgcn_sched_barrier(0);
        accw = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.x, bv0.x, accw, 0, 0, 0);
        accw2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.y, bv0.y, accw2, 0, 0, 0);
        accw = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.x, bv1.x, accw, 0, 0, 0);
        accw2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.y, bv1.y, accw2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        av0 = an0; bv0 = bn0; av1 = an1; bv1 = bn1;
      }
    }
    // ---- data gradient: dv tile -> Os ----
    if (dg_on) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { accd[i] = 0.f; accd2[i] = 0.f; }
      const float* Wa = Ws + half * 65 + 32 * cit + l31;          // A[i = ci][k = co] = W[co][ci]
      const float* Db = Ds + half * BF_LS + 32 * pt + l31;        // B[k = co][j = position]
      // the same pinning: four k-steps per pass, the next pass's eight operands read ahead of this pass's products.  Rows
      // past Co are zero in BOTH images (Ws is zero padded, commit() zeroes the dz_eff rows), so the step count is rounded
      // up to a multiple of four instead of branching on the tail; the read-ahead of the last pass stays inside the
      // weight image / the Xs tile that follows Ds (values never used).
      const int KS4 = (KSd + 3) & ~3;
      float wa[4], db[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { wa[i] = Wa[2 * i * 65]; db[i] = Db[2 * i * BF_LS]; }
      for (int ks = 0; ks < KS4; ks += 4) {
        float wn[4], dn[4];
        const int kn = ks + 4 < 32 ? ks + 4 : 28;                   // (clamped: never past the 64-row images)
#pragma unroll
        for (int i = 0; i < 4; ++i) { wn[i] = Wa[2 * (kn + i) * 65]; dn[i] = Db[2 * (kn + i) * BF_LS]; }
        __builtin_amdgcn_sched_barrier(0);
        accd = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[0], db[0], accd, 0, 0, 0);
        accd2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[1], db[1], accd2, 0, 0, 0);
        accd = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[2], db[2], accd, 0, 0, 0);
        accd2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[3], db[3], accd2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { wa[i] = wn[i]; db[i] = dn[i]; }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Os[(32 * cit + bf_row32(r, half)) * BF_LS + 32 * pt + l31] = accd[r] + accd2[r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // raw barrier: the next unit's loads stay in flight
    BF_STAMP();
    {
      // finish the data gradient in the staging layout (16 B per lane, the raw inputs are still in registers):
      // dx = dv * 1[pre > 0] * s;  (Os is rewritten only after the NEXT unit's first barrier)
      const int n = ch / a.cpn, c0 = (ch - n * a.cpn) * BF_KC;
      if (c0 + col < L) {
#pragma unroll
        for (int j = 0; j < BF_J; ++j) {
          const int row = row0 + 16 * j;
          if (row < Ci) {
            const bf_f2* o = reinterpret_cast<const bf_f2*>(Os + row * BF_LS + col);
            const bf_f2 lo2 = o[0], hi2 = o[1];
            f32x4 dv = {lo2.x, lo2.y, hi2.x, hi2.y};
            f32x4 p = {1.f, 0.f, 1.f, 0.f};
            if constexpr (AFF) {
              p = Ps[row];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float t = fmaf(xk[j][e], p.x, p.y);
                if constexpr (HAS2) t += fmaf(yk[j][e], p.z, p.w);
                if (a.relu && !(t > 0.f)) dv[e] = 0.f;
                r0[j] = fmaf(dv[e], xk[j][e], r0[j]);
                r1[j] += dv[e];
                if constexpr (HAS2) r2[j] = fmaf(dv[e], yk[j][e], r2[j]);
              }
            }
            const size_t go = ((size_t)n * Ci + row) * L + c0 + col;
            __builtin_nontemporal_store(f32x4{dv.x * p.x, dv.y * p.x, dv.z * p.x, dv.w * p.x}, reinterpret_cast<f32x4*>(a.dx + go));
            if constexpr (HAS2) __builtin_nontemporal_store(f32x4{dv.x * p.z, dv.y * p.z, dv.z * p.z, dv.w * p.z}, reinterpret_cast<f32x4*>(a.dx2 + go));
          }
        }
      }
    }
  }

  BF_STAMP();
  // ---- partial rows of this split ----
  float* dw = a.dwp + (size_t)split * a.pstride;
  {
    const int ci = 32 * nt + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = 32 * mt + bf_row32(r, half);
      if (wg_on && co < Co && ci < Ci) dw[(size_t)co * Ci + ci] = accw[r] + accw2[r];
    }
  }
#pragma unroll
  for (int j = 0; j < BF_J; ++j) {
    float s = dsum[j], s0 = r0[j], s1 = r1[j], s2 = r2[j];
#pragma unroll
    for (int off = 1; off < BF_Q; off <<= 1) {
      s += __shfl_xor(s, off, 64);
      if (AFF) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
      if (HAS2) s2 += __shfl_xor(s2, off, 64);
    }
    const int row = row0 + 16 * j;
    if ((tid % BF_Q) == 0) {
      if (row < Co) a.dbp[(size_t)split * a.pstride + row] = s;
      if (AFF && a.ipart && row < Ci) {
        float* o = a.ipart + ((size_t)split * Ci + row) * 3;
        o[0] = s0; o[1] = s1; o[2] = s2;
      }
    }
  }
#ifdef DSGCN_LAB
  BF_STAMP();
  if (blockIdx.x == 0 && threadIdx.x == 0) g_bf_stamp[63] = nst;
#endif
}

// ---- round 5: the same pass on three-term bf16 products ---------------------------------------------------------------
// The fp32 form above is bound by the matrix pipe: 64 v_mfma_f32_32x32x2_f32 per wave and 64-position unit = 4 096 cycles,
// 43 us per 64 -> 64 launch at the chip's 157 TF against 26 us of HBM time.  Here both products run as six
// v_mfma_f32_32x32x16_bf16 per 16-deep k-step on the exact three-way bf16 split of both operands (common.h: fp32-class
// error, what the wide convs have used since round 2): 24 MFMAs of 32 cycles per wave and 32-position unit.
// Work unit = (sample, 32 positions).  LDS (79.5 KB: two workgroups per CU), every image as [term][row][k contiguous]:
//   Dk [64 co][32 pos]  dz_eff           A of the weight gradient (i = co, k = position)
//   Xk [64 ci][32 pos]  activated input  B of the weight gradient (j = ci, k = position)
//   Dt [32 pos][64 co]  dz_eff again     B of the data gradient   (j = position, k = co)
//   Wt [64 ci][64 co]   W^T, once        A of the data gradient   (i = ci, k = co)
//   Os [64 ci][32 pos]  fp32 dv tile, left by the two data-gradient waves for the staging threads to finish.
// A thread stages row tid / 4, positions 8*(tid % 4) ..+7 of both operand rows (two 16-byte loads per stream, split once);
// the transposed image takes its 2-byte elements in pairs: lanes of rows r, r^1 trade halves and each writes a dword.
constexpr int BB_RB = 80, BB_RT = 144, BB_OS = 36;       // bytes per Dk / Xk row, per Dt / Wt row; floats per Os row
constexpr int BB_DK = 0, BB_XK = 3 * 64 * BB_RB, BB_DT = 2 * 3 * 64 * BB_RB, BB_WT = BB_DT + 3 * 32 * BB_RT,
              BB_OSO = BB_WT + 3 * 64 * BB_RT, BB_LDS = BB_OSO + 64 * BB_OS * 4;

template <bool HASC, bool AFF>
__global__ __launch_bounds__(BF_NT, 2) void k_bwd64b(BfArgs a) {
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
#ifdef DSGCN_LAB
  int nst = 0;
#endif
  BF_STAMP();
  char* Dk = ldsb + BB_DK;
  char* Xk = ldsb + BB_XK;
  char* Dt = ldsb + BB_DT;
  char* Wt = ldsb + BB_WT;
  float* Os = reinterpret_cast<float*>(ldsb + BB_OSO);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int Ci = a.Ci, Co = a.Co, L = a.L, L4 = L * 4;
  const int split = blockIdx.x;
  const int ch0 = split * a.cps, ch1 = min(a.total_chunks, ch0 + a.cps);
  // staging role: row (a dz row AND an input row), positions 8*q ..+7 of the unit
  const int row = tid >> 2, q = tid & 3;
  const bool rd = row < Co, rx = row < Ci;
  const float A0v = (HASC && rd) ? a.A0[row] : 0.f, B0v = (HASC && rd) ? a.B0[row] : 0.f;
  float s1v = 1.f, h1v = 0.f;
  if (AFF && rx && a.s1) { s1v = a.s1[row]; h1v = a.h1[row]; }
  const float lo = a.relu ? 0.f : -__builtin_inff();
  // W^T image (pairs of co per thread) and the zeroed dv tile
  for (int i = tid; i < 64 * 32; i += BF_NT) {
    const int ci = i >> 5, co = 2 * (i & 31);
    const float w0 = (ci < Ci && co < Co) ? a.w[(size_t)co * Ci + ci] : 0.f;
    const float w1 = (ci < Ci && co + 1 < Co) ? a.w[(size_t)(co + 1) * Ci + ci] : 0.f;
    unsigned p0, p1, p2;
    b3_split(w0, w1, p0, p1, p2);
    char* base = Wt + ci * BB_RT + co * 2;
    *reinterpret_cast<unsigned*>(base) = p0;
    *reinterpret_cast<unsigned*>(base + 64 * BB_RT) = p1;
    *reinterpret_cast<unsigned*>(base + 2 * 64 * BB_RT) = p2;
  }

  // two units of raw operands in flight (sets A / B by unit parity): with one, a unit's loads had only the ~1 us of the
  // previous unit's products to land in and every commit waited 1-2 us for them (lab stamps, round 5)
  struct Raw { f32x4 g[2], z[HASC ? 2 : 1], x[2]; };
  Raw ra, rb;
  f32x4 xk[AFF ? 2 : 1];
  float dsum = 0.f, r0 = 0.f, r1 = 0.f;
  auto issue = [&](int ch, Raw& R) {
    const int n = ch / a.cpn, c0 = (ch - n * a.cpn) * 32;
    const __amdgpu_buffer_rsrc_t rg = bf_rsrc(a.gz + (size_t)n * Co * L, Co * L4);
    const __amdgpu_buffer_rsrc_t rz = bf_rsrc((HASC ? a.z : a.gz) + (size_t)n * Co * L, HASC ? Co * L4 : 0);
    const __amdgpu_buffer_rsrc_t r1_ = bf_rsrc(a.x1 + (size_t)n * Ci * L, Ci * L4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = c0 + 8 * q + 4 * h;                              // (L % 4 == 0: a quad is inside or outside the plane)
      const int vd = (rd && p < L) ? (row * L + p) * 4 : BF_OOB;
      const int vx = (rx && p < L) ? (row * L + p) * 4 : BF_OOB;
      R.g[h] = bf_load(rg, vd);
      if constexpr (HASC) R.z[h] = bf_load(rz, vd);
      R.x[h] = bf_load(r1_, vx);
    }
  };
  auto commit = [&](int ch, const Raw& R) {
    const int n = ch / a.cpn, c0 = (ch - n * a.cpn) * 32;
    float d[8], v[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool in = c0 + 8 * q + 4 * h < L;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = R.g[h][e];
        if constexpr (HASC) x += fmaf(B0v, R.z[h][e], A0v);
        d[4 * h + e] = (rd && in) ? x : 0.f;
        float t = R.x[h][e];
        if constexpr (AFF) {
          t = fmaf(t, s1v, h1v);
          t = fmaxf(t, lo);
        }
        v[4 * h + e] = (rx && in) ? t : 0.f;
      }
      if constexpr (AFF) xk[h] = R.x[h];
    }
    dsum += ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
    unsigned t0[4], t1[4], t2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b3_split(d[2 * j], d[2 * j + 1], t0[j], t1[j], t2[j]);
    {
      char* base = Dk + row * BB_RB + q * 16;
      *reinterpret_cast<u32x4v*>(base) = u32x4v{t0[0], t0[1], t0[2], t0[3]};
      *reinterpret_cast<u32x4v*>(base + 64 * BB_RB) = u32x4v{t1[0], t1[1], t1[2], t1[3]};
      *reinterpret_cast<u32x4v*>(base + 2 * 64 * BB_RB) = u32x4v{t2[0], t2[1], t2[2], t2[3]};
    }
    // transposed image: word j holds positions (8q + 2j, 8q + 2j + 1) of this row; the lane of row^1 (4 lanes away) holds
    // the same positions of its row.  Even rows write the pair (row, row + 1) of the first position, odd rows of the second.
    {
      const int odd = row & 1;
      char* base = Dt + (8 * q + odd) * BB_RT + (row & ~1) * 2;
      // (v_perm_b32 byte selects instead of a select between two forms: hipcc turned the ternaries into twelve divergent
      // branch diamonds, each waiting for its lane exchange)
      // v_perm_b32 over {src0 = mine : src1 = the other row's}: odd rows take both HIGH halves (other's below mine), even
      // rows both LOW halves (mine below the other's)
      const unsigned sel = odd ? 0x07060302u : 0x01000504u;
      unsigned o0[4], o1[4], o2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o0[j] = __shfl_xor(t0[j], 4, 64); o1[j] = __shfl_xor(t1[j], 4, 64); o2[j] = __shfl_xor(t2[j], 4, 64);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned w0 = __builtin_amdgcn_perm(t0[j], o0[j], sel);
        const unsigned w1 = __builtin_amdgcn_perm(t1[j], o1[j], sel);
        const unsigned w2 = __builtin_amdgcn_perm(t2[j], o2[j], sel);
        char* pj = base + 2 * j * BB_RT;
        *reinterpret_cast<unsigned*>(pj) = w0;
        *reinterpret_cast<unsigned*>(pj + 32 * BB_RT) = w1;
        *reinterpret_cast<unsigned*>(pj + 2 * 32 * BB_RT) = w2;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) b3_split(v[2 * j], v[2 * j + 1], t0[j], t1[j], t2[j]);
    {
      char* base = Xk + row * BB_RB + q * 16;
      *reinterpret_cast<u32x4v*>(base) = u32x4v{t0[0], t0[1], t0[2], t0[3]};
      *reinterpret_cast<u32x4v*>(base + 64 * BB_RB) = u32x4v{t1[0], t1[1], t1[2], t1[3]};
      *reinterpret_cast<u32x4v*>(base + 2 * 64 * BB_RB) = u32x4v{t2[0], t2[1], t2[2], t2[3]};
    }
    (void)n;
  };

  // waves 0, 1: weight gradient — wave mt owns the co tile mt of dW, both ci tiles (K = the unit's 32 positions);
  // waves 2, 3: data gradient   — wave 2 + cit owns ci tile cit of the 64 x 32 dv tile over all of co.
  // (24 MFMAs per wave either way.  The first version split the data gradient's K over wave pairs and added the halves
  // into Os with ds_add_f32: 16 LDS float atomics per lane and unit took 6-10 us — the lab stamps.)
  const bool wgw = wave < 2;
  const int mt = wave & 1, cit = wave & 1;
  const bool wg0 = wgw && 32 * mt < Co, wg1 = wg0 && Ci > 32;
  const bool dg_on = !wgw && 32 * cit < Ci;
  const int KSD = (Co + 15) >> 4;                                 // data-gradient k-steps that hold real channels
  f32x16 accw, accw1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { accw[i] = 0.f; accw1[i] = 0.f; }
  const char* Af = Dk + (32 * mt + l31) * BB_RB + 16 * half;
  const char* Bf = Xk + l31 * BB_RB + 16 * half;
  const char* Wf = Wt + (32 * cit + l31) * BB_RT + 16 * half;
  const char* Df = Dt + l31 * BB_RT + 16 * half;
  auto six = [&](f32x16 c, const char* pa, int sa, const char* pb, int sb) {
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(pa)),
                 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(pa + sa)),
                 a2 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(pa + 2 * sa));
    const bf16x8 b0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(pb)),
                 b1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(pb + sb)),
                 b2 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(pb + 2 * sb));
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
    return c;
  };

  __syncthreads();
  BF_STAMP();
  auto unit = [&](int ch, Raw& R) {
    commit(ch, R);
    if (ch + 2 < ch1) issue(ch + 2, R);                            // (the set is free: commit kept what the finish needs)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // images complete (raw barrier: loads stay in flight)
    BF_STAMP();
    if (wg0) {
      accw = six(accw, Af, 64 * BB_RB, Bf, 64 * BB_RB);
      accw = six(accw, Af + 32, 64 * BB_RB, Bf + 32, 64 * BB_RB);
      if (wg1) {
        accw1 = six(accw1, Af, 64 * BB_RB, Bf + 32 * BB_RB, 64 * BB_RB);
        accw1 = six(accw1, Af + 32, 64 * BB_RB, Bf + 32 * BB_RB + 32, 64 * BB_RB);
      }
    }
    if (dg_on) {
      f32x16 accd;
#pragma unroll
      for (int i = 0; i < 16; ++i) accd[i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (ks < KSD) accd = six(accd, Wf + 32 * ks, 64 * BB_RT, Df + 32 * ks, 32 * BB_RT);
#pragma unroll
      for (int r = 0; r < 16; ++r) Os[(32 * cit + bf_row32(r, half)) * BB_OS + l31] = accd[r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // dv tile complete; the images are free
    BF_STAMP();
    {
      const int n = ch / a.cpn, c0 = (ch - n * a.cpn) * 32;
      f32x4* o = reinterpret_cast<f32x4*>(Os + row * BB_OS + 8 * q);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 dv = o[h];
        if (rx && c0 + 8 * q + 4 * h < L) {
          float sx = 1.f;
          if constexpr (AFF) {
            sx = s1v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t = fmaf(xk[h][e], s1v, h1v);
              if (a.relu && !(t > 0.f)) dv[e] = 0.f;
              r0 = fmaf(dv[e], xk[h][e], r0);
              r1 += dv[e];
            }
          }
          const size_t go = ((size_t)n * Ci + row) * L + c0 + 8 * q + 4 * h;
          __builtin_nontemporal_store(f32x4{dv.x * sx, dv.y * sx, dv.z * sx, dv.w * sx}, reinterpret_cast<f32x4*>(a.dx + go));
        }
      }
    }
  };
  if (ch0 < ch1) issue(ch0, ra);
  if (ch0 + 1 < ch1) issue(ch0 + 1, rb);
  for (int ch = ch0; ch < ch1; ch += 2) {
    unit(ch, ra);
    if (ch + 1 < ch1) unit(ch + 1, rb);
  }

  BF_STAMP();
  // ---- partial rows of this split ----
  float* dw = a.dwp + (size_t)split * a.pstride;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = 32 * mt + bf_row32(r, half);
    if (wg0 && co < Co && l31 < Ci) dw[(size_t)co * Ci + l31] = accw[r];
    if (wg1 && co < Co && 32 + l31 < Ci) dw[(size_t)co * Ci + 32 + l31] = accw1[r];
  }
  {
    float s = dsum, s0 = r0, s1 = r1;
#pragma unroll
    for (int off = 1; off < 4; off <<= 1) {
      s += __shfl_xor(s, off, 64);
      if (AFF) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
    }
    if (q == 0) {
      if (rd) a.dbp[(size_t)split * a.pstride + row] = s;
      if (AFF && a.ipart && rx) {
        float* o = a.ipart + ((size_t)split * Ci + row) * 3;
        o[0] = s0; o[1] = s1; o[2] = 0.f;
      }
    }
  }
#ifdef DSGCN_LAB
  BF_STAMP();
  if (blockIdx.x == 0 && threadIdx.x == 0) g_bf_stamp[63] = nst;
#endif
}

struct BfPlan { int cpn, chunks, splits, cps; };

// 1: one-stream convs with Ci >= 16 take the three-term bf16 form (k_bwd64b, 32-position units); 0: always the fp32 MFMA
// form (lab A/B).  Round-5 same-box timings, fp32 -> bf16 form: 64->24 38.5 -> 37.2 us, 24->64 42.7 -> 39.6, 64->64 59.8 ->
// 52.0; the two-stream 64->64 (72.6 -> 81.0) and the 3->24 input conv (25.7 -> 27.7) are slower there and stay on fp32.
int g_bf_b3 = 1;

bool bf_plan(int n, int Ci, int Co, int L, BfPlan* p) {
  if (Ci <= 0 || Co <= 0 || Ci > 64 || Co > 64 || L % 4) return false;
  if ((long)64 * L * 4 >= (1L << 31) - 64) return false;
  // (the partial-row count is the fp32 form's for both forms: dsgcn_pwconv_bwd_rows does not know the stream count)
  p->cpn = (L + BF_KC - 1) / BF_KC;
  p->chunks = n * p->cpn;
  int target = 512;                                               // two workgroups per CU
  if (target > p->chunks) target = p->chunks;
  p->cps = (p->chunks + target - 1) / target;
  p->splits = (p->chunks + p->cps - 1) / p->cps;
  return true;
}

}  // namespace

// Internal (hidden): partial rows of the fused backward for this shape, 0 = not eligible.
__attribute__((visibility("hidden"))) int dsgcn_bwd64_splits(int n, int Ci, int Co, int L) {
  BfPlan p;
  return bf_plan(n, Ci, Co, L, &p) ? p.splits : 0;
}

__attribute__((visibility("hidden"))) int dsgcn_bwd64(const float* x1, const float* s1, const float* h1, const float* x2,
                                                       const float* s2, const float* h2, int relu, const float* w,
                                                       const float* z, const float* gz, const float* A0, const float* B0,
                                                       float* dx, float* dx2, float* dwp, float* dbp, int pstride,
                                                       float* ipart, int n, int Ci, int Co, int L, hipStream_t st) {
  BfPlan p;
  if (!bf_plan(n, Ci, Co, L, &p)) return 0;
  BfArgs a = {};
  a.x1 = x1; a.s1 = s1; a.h1 = h1; a.x2 = x2; a.s2 = s2; a.h2 = h2; a.relu = relu; a.w = w; a.z = z; a.gz = gz;
  a.A0 = A0; a.B0 = B0; a.dx = dx; a.dx2 = dx2; a.dwp = dwp; a.dbp = dbp; a.ipart = ipart; a.pstride = pstride;
  a.n = n; a.Ci = Ci; a.Co = Co; a.L = L; a.cpn = p.cpn; a.total_chunks = p.chunks; a.cps = p.cps;
  const bool b3 = g_bf_b3 && x2 == nullptr && Ci >= 16;
  if (b3) {                                           // 32-position units over the same splits (a split with none writes zero rows)
    a.cpn = (L + 31) / 32;
    a.total_chunks = n * a.cpn;
    a.cps = (a.total_chunks + p.splits - 1) / p.splits;
  }
  const size_t lds = b3 ? (size_t)BB_LDS : (size_t)(192 * BF_LS + 64 * 65 + 2 * 64 + 4 * 64) * sizeof(float);
  const dim3 grid((unsigned)p.splits), blk(BF_NT);
  const bool hasc = A0 != nullptr, has2 = x2 != nullptr, aff = s1 != nullptr || s2 != nullptr || relu != 0 || has2;
  if (b3) {
#define BB_LAUNCH(HC, AF)                                                                                               \
  {                                                                                                                   \
    static bool raised = false;                                                                                       \
    if (!raised) {                                                                                                    \
      hipError_t e = hipFuncSetAttribute((const void*)k_bwd64b<HC, AF>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)lds);                                                                   \
      if (e != hipSuccess) return (int)e;                                                                             \
      raised = true;                                                                                                  \
    }                                                                                                                 \
    hipLaunchKernelGGL((k_bwd64b<HC, AF>), grid, blk, lds, st, a);                                                \
  }
    if (hasc) {
      if (aff) BB_LAUNCH(true, true) else BB_LAUNCH(true, false)
    } else {
      if (aff) BB_LAUNCH(false, true) else BB_LAUNCH(false, false)
    }
#undef BB_LAUNCH
    DSGCN_LAUNCH_CHECK();
    return 1;
  }
  // the LDS image (69 KB) is above the default dynamic limit: raised once per instantiation (not a stream operation:
  // the first call of a shape class is an eager one, outside any graph capture)
#define BF_LAUNCH(HC, AF, H2)                                                                                          \
  {                                                                                                                   \
    static bool raised = false;                                                                                       \
    if (!raised) {                                                                                                    \
      hipError_t e = hipFuncSetAttribute((const void*)k_bwd64<HC, AF, H2>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)lds);                                                                   \
      if (e != hipSuccess) return (int)e;                                                                             \
      raised = true;                                                                                                  \
    }                                                                                                                 \
    hipLaunchKernelGGL((k_bwd64<HC, AF, H2>), grid, blk, lds, st, a);                                                 \
  }
  if (has2) {
    if (hasc) BF_LAUNCH(true, true, true) else BF_LAUNCH(false, true, true)
  } else if (hasc) {
    if (aff) BF_LAUNCH(true, true, false) else BF_LAUNCH(true, false, false)
  } else {
    if (aff) BF_LAUNCH(false, true, false) else BF_LAUNCH(false, false, false)
  }
#undef BF_LAUNCH
  DSGCN_LAUNCH_CHECK();
  return 1;
}

__attribute__((visibility("hidden"))) int dsgcn_bwd64_tuning(int value) { g_bf_b3 = value; return 0; }

#ifdef DSGCN_LAB
extern "C" int dsgcn_bwd64_phases(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bf_stamp), sizeof(long long) * 64);
}
#endif
