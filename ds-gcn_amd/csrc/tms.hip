// K-D, third generation: the WHOLE multi-scale temporal stage of dgmstcn / mstcn / MSTCN between its two 1x1 convs as
// one launch per direction, on the reference's own (n, C, T, V) layout for any V (odd or even):
//   h      = act(z * scale + shift) (+ the global-joint column act(zaug * scale + shift))      tcn.py:389-394,409
//   o      = per channel window: (KT,1) dilated conv | (3,1) max-pool | strided copy            tcn.py:383-396
//   f      = o[..., :V] + o[..., V] * add_coeff   (+ sum f, sum f^2 per channel: BN statistics)  tcn.py:416-420,401
// and for CTR-GCN's MSTCN / ST-GCN++'s mstcn the same without the global joint (msg3d_utils.py:84-117, tcn.py:104-177).
// It replaces three launches per direction of the second generation (k_branch_act_*, k_tap4 / k_tapw, k_tms_combine_*)
// and their two intermediate tensors h and o of (V+1)-column rows: neither is materialised any more.
//
// Why a new design instead of prologues on k_tap4: k_tap4 feeds its MFMA B operand straight from global memory with
// 16-byte loads at the tap-shifted row, which needs an EVEN row pitch (26 = 25 joints + the global joint) and therefore
// the layout conversions on either side.  Here a workgroup stages the rows its output tile needs ONCE into LDS — aligned
// 16-byte loads of whole row ranges (tiles start at multiples of 4 frames, so every range is 16-byte aligned whatever V),
// the deferred BN affine + ReLU applied on the way, the global-joint column appended IN LDS (pitch V+1 there, V in HBM),
// zero rows outside the clip — and all taps read their shifted operand from that one tile.  The accumulators bounce
// through LDS for the epilogue, which adds the global-joint term, gathers the statistics and stores 16 bytes per lane.
// Matrix work runs on v_mfma_f32_16x16x4_f32: the branch widths are 10..46 channels, 16-row tiles waste half as much
// as 32-row ones.
//
// Work unit = (sample, tile of R output frames); grid = (unit stripes, channel windows); a workgroup keeps its window's
// weights in LDS and walks its stripe of units.  All reductions are per-workgroup partial rows (dsgcn_colsum /
// dsgcn_bn_finalize finish them): deterministic, no atomics.
#include <algorithm>

#include "common.h"

namespace {

constexpr int TM_NT = 512;           // threads per workgroup (8 waves: per-thread staging registers and per-wave
                                     // accumulators stay small enough for two workgroups per CU)
constexpr int TM_MAXBR = 8;
constexpr int TM_H = 4;              // halo frames on either side = largest |tap shift| (KT=3: dil <= 4; KT=5: dil <= 2)
constexpr int TM_CK = 16;            // channels per staged K chunk
constexpr int TM_NTW = 4;            // 16-position tiles per wave (8 waves: R * (V+1) <= 512)
constexpr size_t TM_LDS_MAX = 156 * 1024;

struct TmBranch {
  int type;            // 0 conv, 1 max-pool(3), 2 strided copy
  int c0, bc, dil;     // channel window [c0, c0 + bc) of both input and output
  const float* w;      // (bc, bc, KT, 1)
  const float* b;      // (bc) or NULL
  float* dwp;          // weight-gradient partial rows (row stride pstride)
  float* dbp;
};

struct TmArgs {
  // forward operands
  const float* z;      // (n, C, T, V) raw input
  const float* zaug;   // (n, C, T) global-joint column of the input, or NULL
  const float* scale;  // (C) deferred affine of the input, or NULL (identity)
  const float* shift;
  const float* coeff;  // (V) add_coeff, with zaug
  float* f;            // (n, C, Tout, V)
  float* oaug;         // (n, C, Tout) the global-joint column of o (saved for the backward), with zaug
  float* stats;        // (gridDim.x, C, 2) partial sums of f, f^2, or NULL
  // backward operands
  const float* gf;     // (n, C, Tout, V) gradient of f
  const float* fin;    // f (for the statistics term B0 * f), with A0
  const float* A0;     // (C) statistics terms of the BatchNorm that follows f, or NULL
  const float* B0;
  const float* oaug_in;
  float* dz;           // (n, C, T, V)
  float* dzaug;        // (n, C, T)
  float* paff;         // (gridDim.x, C, 2) partials of d scale, d shift
  float* pcoeff;       // (gridDim.x * nbr, V) partials of d coeff
  int n, C, T, Tout, V, stride, n_act, nbr, R, tiles, units, pstride, dbg;
  TmBranch br[TM_MAXBR];
};

__host__ __device__ inline int tm_xs(int w) { return ((w + 15) / 32) * 32 + 16; }   // >= w, = 16 mod 32: the four k
                                                                                     // rows of a fragment hit distinct banks

// ---- staging: rows [t_lo, t_lo + rows) of TM_CK channels (chunk kc of window br) -> Xl[cl][r * VL + x], act applied ------
// src plane pitch V, LDS pitch VL = V + aug; rows outside [0, Trows) and channels past the window are zero.
// MODE 0: forward input h = act(z * scale + shift);  MODE 1: backward input dfe = gf + A0 + B0 * f (no aug column here)
template <int MODE>
__device__ __forceinline__ void tm_stage(const TmArgs& a, const TmBranch& br, float* Xl, int XS, int n, int kc, int t_lo,
                                         int rows, int Trows, int VL, int tid) {
  const int V = a.V;
  const int F4 = (rows * V) >> 2;                     // float4 per channel row range (rows % 4 == 0)
  const int total = TM_CK * F4;
  const long plane = (long)Trows * V;
  constexpr int MAXJ = 6;
  f32x4 v[MAXJ], w[MAXJ];
  const float invV = 1.f / (float)V;
  for (int base = 0; base < total; base += TM_NT * MAXJ) {
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int id = base + tid + TM_NT * j;
      const int cl = id / F4, q = id - cl * F4;
      const int c = kc * TM_CK + cl;
      const long g = (long)t_lo * V + 4 * q;          // flat index inside the plane
      const bool ok = id < total && c < br.bc && g >= 0 && g < plane;
      const long off = ((long)n * a.C + br.c0 + c) * plane + g;
      if (MODE == 0) {
        v[j] = ok ? *reinterpret_cast<const f32x4*>(a.z + off) : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        v[j] = ok ? *reinterpret_cast<const f32x4*>(a.gf + off) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.A0) w[j] = ok ? *reinterpret_cast<const f32x4*>(a.fin + off) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int id = base + tid + TM_NT * j;
      if (id >= total) continue;
      const int cl = id / F4, q = id - cl * F4;
      const int c = kc * TM_CK + cl;
      const long g = (long)t_lo * V + 4 * q;
      const bool ok = c < br.bc && g >= 0 && g < plane;
      float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
      if (ok) {
        const int cg = br.c0 + c;
        if (MODE == 0) {
          if (a.scale) {
            const float sc = a.scale[cg], sh = a.shift[cg];
            const bool relu = cg < a.n_act;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float t = fmaf(e[i], sc, sh);
              e[i] = relu ? fmaxf(t, 0.f) : t;
            }
          }
        } else if (a.A0) {
          const float a0 = a.A0[cg], b0 = a.B0[cg];
          const float ff[4] = {w[j].x, w[j].y, w[j].z, w[j].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) e[i] = e[i] + fmaf(b0, ff[i], a0);
        }
      }
      int r, x;
      divmod_small(4 * q, V, invV, r, x);
      float* dst = Xl + cl * XS;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dst[r * VL + x] = e[i];
        if (++x == V) { x = 0; ++r; }
      }
    }
  }
}

// global-joint column of the forward input: Xl[cl][r * VL + V] = act(zaug[n, c, t_lo + r])
__device__ __forceinline__ void tm_stage_aug(const TmArgs& a, const TmBranch& br, float* Xl, int XS, int n, int kc, int t_lo,
                                             int rows, int VL, int tid) {
  for (int id = tid; id < TM_CK * rows; id += TM_NT) {
    const int cl = id / rows, r = id - cl * rows;
    const int c = kc * TM_CK + cl, t = t_lo + r;
    float v = 0.f;
    if (c < br.bc && t >= 0 && t < a.T) {
      const int cg = br.c0 + c;
      v = a.zaug[((long)n * a.C + cg) * a.T + t];
      if (a.scale) {
        v = fmaf(v, a.scale[cg], a.shift[cg]);
        if (cg < a.n_act) v = fmaxf(v, 0.f);
      }
    }
    Xl[cl * XS + r * VL + a.V] = v;
  }
}

// weights of window br -> Wl[(tap * CKW + k) * MP + m]: FWD k = ci, m = co; transposed (data gradient) k = co, m = ci
template <int KT, bool FWD>
__device__ __forceinline__ void tm_stage_w(const TmBranch& br, float* Wl, int CKW, int MP, int tid) {
  for (int i = tid; i < KT * CKW * MP; i += TM_NT) Wl[i] = 0.f;
  __syncthreads();
  const int bc = br.bc, total = bc * bc * KT;
  for (int i = tid; i < total; i += TM_NT) {
    const int co = i / (bc * KT), r = i - co * bc * KT, ci = r / KT, tap = r - ci * KT;
    const float v = br.w[i];
    const int k = FWD ? ci : co, m = FWD ? co : ci;
    Wl[(tap * CKW + k) * MP + m] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// software-pipelined staging: the global loads of the NEXT (unit, channel chunk) are issued into registers right after
// the current one has been committed to LDS and land while the matrix product and the epilogue run.  Barriers are raw
// s_barrier + lgkmcnt(0): a __syncthreads() would also drain vmcnt, i.e. wait for the epilogue's stores and the
// prefetch at every phase boundary (measured: the three phases of a unit then run strictly one after the other at
// ~5 us each, none of them near a bound).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tm_bar() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

constexpr int TM_NPF = 5;            // float4 prefetch slots per thread (16 channels x rows x V / 4 <= 2560)

struct TmPf {
  f32x4 v[TM_NPF];
  f32x4 w[TM_NPF];                   // second stream (MODE 1 with statistics terms: f)
  float aug[2];
};

// issue the loads of chunk kc of the tile [t_lo, t_lo + rows) (see tm_stage); SECOND: also the f stream.
// Index arithmetic: float-reciprocal divisions and 32-bit offsets from a wave-uniform base (the launch checks that a
// 16-channel slab stays below 2^31 bytes) — runtime integer divisions cost ~25 instructions and several registers each.
template <int MODE, bool SECOND>
__device__ __forceinline__ void tm_issue(const TmArgs& a, const TmBranch& br, TmPf& pf, int n, int kc, int t_lo, int rows,
                                         int Trows, bool aug, int tid) {
  // (the slot -> (channel, float4) maps do not depend on the unit: left alone, the compiler hoists all of them out of the
  // unit loop and keeps ~60 registers of indices alive across the matrix product; an opaque copy of the thread id pins the
  // few VALU operations per slot inside the loop instead)
  asm volatile("" : "+v"(tid));
  const int V = a.V;
  const int F4 = (rows * V) >> 2, total = TM_CK * F4;
  const int plane = Trows * V;
  const float invF4 = 1.f / (float)F4;
  const long base = ((long)n * a.C + br.c0 + kc * TM_CK) * plane + (long)t_lo * V;     // wave-uniform
  const float* src = (MODE == 0 ? a.z : a.gf) + base;
  const float* src2 = SECOND ? a.fin + base : nullptr;
  const int g0 = t_lo * V;
#pragma unroll
  for (int j = 0; j < TM_NPF; ++j) {
    const int id = tid + TM_NT * j;
    int cl, q;
    divmod_small(id, F4, invF4, cl, q);
    const int g = g0 + 4 * q;
    const bool ok = id < total && kc * TM_CK + cl < br.bc && g >= 0 && g < plane;
    const int off = cl * plane + 4 * q;
    pf.v[j] = ok ? *reinterpret_cast<const f32x4*>(src + off) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (SECOND) pf.w[j] = ok ? *reinterpret_cast<const f32x4*>(src2 + off) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (MODE == 0 && aug) {
    const float invR = 1.f / (float)rows;
    const float* za = a.zaug + ((long)n * a.C + br.c0 + kc * TM_CK) * a.T + t_lo;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int id = tid + TM_NT * j;
      int cl, r;
      divmod_small(id, rows, invR, cl, r);
      const int t = t_lo + r;
      const bool ok = id < TM_CK * rows && kc * TM_CK + cl < br.bc && t >= 0 && t < a.T;
      pf.aug[j] = ok ? za[cl * a.T + r] : 0.f;
    }
  }
}

// registers -> LDS tile (deferred affine + ReLU / statistics terms applied here)
template <int MODE, bool SECOND>
__device__ __forceinline__ void tm_commit(const TmArgs& a, const TmBranch& br, const TmPf& pf, float* Xl, int XS, int kc,
                                          int t_lo, int rows, int Trows, int VL, bool aug, int tid) {
  asm volatile("" : "+v"(tid));                      // (see tm_issue)
  const int V = a.V;
  const int F4 = (rows * V) >> 2, total = TM_CK * F4;
  const int plane = Trows * V;
  const float invV = 1.f / (float)V, invF4 = 1.f / (float)F4;
  const int g0 = t_lo * V;
#pragma unroll
  for (int j = 0; j < TM_NPF; ++j) {
    const int id = tid + TM_NT * j;
    if (id >= total) continue;
    int cl, q;
    divmod_small(id, F4, invF4, cl, q);
    const int c = kc * TM_CK + cl;
    const int g = g0 + 4 * q;
    const bool ok = c < br.bc && g >= 0 && g < plane;
    float e[4] = {pf.v[j].x, pf.v[j].y, pf.v[j].z, pf.v[j].w};
    if (ok) {
      const int cg = br.c0 + c;
      if (MODE == 0) {
        if (a.scale) {
          const float sc = a.scale[cg], sh = a.shift[cg];
          const bool relu = cg < a.n_act;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float t = fmaf(e[i], sc, sh);
            e[i] = relu ? fmaxf(t, 0.f) : t;
          }
        }
      } else if (SECOND) {
        const float a0 = a.A0[cg], b0 = a.B0[cg];
        const float ff[4] = {pf.w[j].x, pf.w[j].y, pf.w[j].z, pf.w[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = e[i] + fmaf(b0, ff[i], a0);
      }
    }
    int r, x;
    divmod_small(4 * q, V, invV, r, x);
    float* dst = Xl + cl * XS + r * VL + x;
    const int wrap = V - x;                          // elements from index `wrap` on belong to the next frame: + (VL - V)
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i + (i >= wrap ? VL - V : 0)] = e[i];
  }
  if (MODE == 0 && aug) {
    const float invR = 1.f / (float)rows;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int id = tid + TM_NT * j;
      if (id >= TM_CK * rows) continue;
      int cl, r;
      divmod_small(id, rows, invR, cl, r);
      const int c = kc * TM_CK + cl, t = t_lo + r;
      float v = pf.aug[j];
      if (c < br.bc && t >= 0 && t < a.T && a.scale) {
        const int cg = br.c0 + c;
        v = fmaf(v, a.scale[cg], a.shift[cg]);
        if (cg < a.n_act) v = fmaxf(v, 0.f);
      }
      Xl[cl * XS + r * VL + V] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <int KT, int MTL>
__global__ __launch_bounds__(TM_NT, MTL >= 3 ? 3 : 4) void k_tms_fwd(TmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const TmBranch& br = a.br[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int V = a.V, VL = V + (a.zaug ? 1 : 0);
  const bool AUG = a.zaug != nullptr;
  const int R = a.R, s = a.stride;
  const int RIN = R * s + 2 * TM_H;
  const int XS = tm_xs(RIN * VL);
  const int bc = br.bc, nchunk = (bc + TM_CK - 1) / TM_CK, CKW = nchunk * TM_CK, MP = MTL * 16;
  float* Xl = lds;                                   // [TM_CK][XS]; the accumulator bounce Dl[16][DS] aliases it
  float* Wl = lds + TM_CK * XS;                      // [KT][CKW][MP]   (conv windows)
  float* Cl = Wl + (br.type == 0 ? KT * CKW * MP : 0);      // coeff[V]
  const int DS = tm_xs(R * VL);
  const int NP = R * VL, NTU = (NP + 15) >> 4;       // positions / 16-position tiles per unit
  const float invVL = 1.f / (float)VL, invV = 1.f / (float)V;
  if (br.type == 0) tm_stage_w<KT, true>(br, Wl, CKW, MP, tid);
  if (AUG) for (int i = tid; i < V; i += TM_NT) Cl[i] = a.coeff[i];
  // the lane's position in each of its wave's tiles: LDS offset of the centre tap (row r*s + H, column x)
  int offP[TM_NTW];
#pragma unroll
  for (int i = 0; i < TM_NTW; ++i) {
    int P = 16 * (wave + 8 * i) + (lane & 15);
    if (P >= NP) P = NP - 1;                         // (columns past the tile: computed, never written)
    int r, x;
    divmod_small(P, VL, invVL, r, x);
    offP[i] = (r * s + TM_H) * VL + x;
  }
  const int co_l = tid >> 5, sub = tid & 31;         // epilogue role: one of 16 output rows, 32 threads per row
  float s1[MTL], s2[MTL];
#pragma unroll
  for (int m = 0; m < MTL; ++m) s1[m] = s2[m] = 0.f;
  const int kq = lane >> 4, l15 = lane & 15;
#ifdef DSGCN_LAB
  const int dbg = a.dbg;             // ablation mask (tools/): 1 no staging, 2 no MFMA, 4 no epilogue
#else
  constexpr int dbg = 0;
#endif
  TmPf pf;
  f32x4 acc[TM_NTW][MTL];

  // the pipelined items are (unit, chunk) pairs in order
  int u = blockIdx.x, kc = 0;
  const float invTiles = 1.f / (float)a.tiles;
  auto tile = [&](int uu, int& n, int& t0) { int k; divmod_small(uu, a.tiles, invTiles, n, k); t0 = k * R; };
  if (u < a.units) {
    int n, t0;
    tile(u, n, t0);
    tm_issue<0, false>(a, br, pf, n, 0, t0 * s - TM_H, RIN, a.T, AUG, tid);
  }
  while (u < a.units) {
    int n, t0;
    tile(u, n, t0);
    const int rows = min(R, a.Tout - t0);                           // live output frames
    const int t_lo = t0 * s - TM_H;
    const int nq = (rows * V) >> 2;
    tm_bar();                                                       // the previous item's readers are done
    if (!(dbg & 1)) tm_commit<0, false>(a, br, pf, Xl, XS, kc, t_lo, RIN, a.T, VL, AUG, tid);
    tm_bar();
    {                                                               // next item's loads
      int un = u, kn = kc + 1;
      if (kn == nchunk) { kn = 0; un += gridDim.x; }
      if (un < a.units && !(dbg & 1)) {
        int nn, tn;
        tile(un, nn, tn);
        tm_issue<0, false>(a, br, pf, nn, kn, tn * s - TM_H, RIN, a.T, AUG, tid);
      }
    }
    if (br.type == 0) {
      if (kc == 0) {
#pragma unroll
        for (int i = 0; i < TM_NTW; ++i)
#pragma unroll
          for (int m = 0; m < MTL; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (!(dbg & 2)) {
        // KT * 4 k-steps of 4 channels; fragments of step st+1 are read from LDS while step st multiplies
        constexpr int NST = KT * (TM_CK / 4);
        float av[2][MTL], bv[2][TM_NTW];
        auto frag = [&](int st, float (&af)[MTL], float (&bf)[TM_NTW]) {
          const int tap = st >> 2, ks = st & 3;
          const float* wrow = Wl + (tap * CKW + kc * TM_CK + 4 * ks + kq) * MP + l15;
          const float* xrow = Xl + (4 * ks + kq) * XS + (tap - KT / 2) * br.dil * VL;
#pragma unroll
          for (int m = 0; m < MTL; ++m) af[m] = wrow[16 * m];
#pragma unroll
          for (int i = 0; i < TM_NTW; ++i) bf[i] = xrow[offP[i]];
        };
        frag(0, av[0], bv[0]);
#pragma unroll 1
        for (int st = 0; st < NST; st += 2) {
          frag(st + 1, av[1], bv[1]);
#pragma unroll
          for (int i = 0; i < TM_NTW; ++i)
            if (wave + 8 * i < NTU)
#pragma unroll
              for (int m = 0; m < MTL; ++m) acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][m], bv[0][i], acc[i][m], 0, 0, 0);
          if (st + 2 < NST) frag(st + 2, av[0], bv[0]);
#pragma unroll
          for (int i = 0; i < TM_NTW; ++i)
            if (wave + 8 * i < NTU)
#pragma unroll
              for (int m = 0; m < MTL; ++m) acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][m], bv[1][i], acc[i][m], 0, 0, 0);
        }
      }
      if (kc == nchunk - 1 && !(dbg & 4)) {
        // epilogue, one 16-row tile at a time through the bounce buffer
#pragma unroll
        for (int m = 0; m < MTL; ++m) {
          if (16 * m >= bc) break;
          tm_bar();
#pragma unroll
          for (int i = 0; i < TM_NTW; ++i) {
            const int P = 16 * (wave + 8 * i) + l15;
            if (wave + 8 * i < NTU && P < NP) {
#pragma unroll
              for (int r = 0; r < 4; ++r) Xl[(4 * kq + r) * DS + P] = acc[i][m][r];
            }
          }
          tm_bar();
          const int co = 16 * m + co_l;
          if (co < bc) {
            const float bias = br.b ? br.b[co] : 0.f;
            const float* drow = Xl + co_l * DS;
            float* dst = a.f + (((long)n * a.C + br.c0 + co) * a.Tout + t0) * V;
            for (int q = sub; q < nq; q += 32) {
              int r, x;
              divmod_small(4 * q, V, invV, r, x);
              float e[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                float v = drow[r * VL + x] + bias;
                if (AUG) v = fmaf(drow[r * VL + V] + bias, Cl[x], v);
                e[i] = v;
                s1[m] += v;
                s2[m] = fmaf(v, v, s2[m]);
                if (++x == V) { x = 0; ++r; }
              }
              *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{e[0], e[1], e[2], e[3]};
            }
            if (AUG)
              for (int r = sub; r < rows; r += 32)
                a.oaug[((long)n * a.C + br.c0 + co) * a.Tout + t0 + r] = drow[r * VL + V] + bias;
          }
        }
      }
    } else if (!(dbg & 4)) {
      // max-pool / strided copy: straight from the staged tile; chunk kc = statistics slot kc
      const int co = kc * TM_CK + co_l;
      if (co < bc) {
        const float* xrow = Xl + co_l * XS;
        auto val = [&](int r, int x) -> float {                    // output frame t0 + r, column x (x = V: global joint)
          const int rr = r * s + TM_H;
          float v = xrow[rr * VL + x];
          if (br.type == 1) {
            const int t = (t0 + r) * s;                             // centre input frame; neighbours only inside the clip
            if (t - 1 >= 0) v = fmaxf(v, xrow[(rr - 1) * VL + x]);
            if (t + 1 < a.T) v = fmaxf(v, xrow[(rr + 1) * VL + x]);
          }
          return v;
        };
        float* dst = a.f + (((long)n * a.C + br.c0 + co) * a.Tout + t0) * V;
        float l1 = 0.f, l2 = 0.f;
        for (int q = sub; q < nq; q += 32) {
          int r, x;
          divmod_small(4 * q, V, invV, r, x);
          float e[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = val(r, x);
            if (AUG) v = fmaf(val(r, V), Cl[x], v);
            e[i] = v;
            l1 += v;
            l2 = fmaf(v, v, l2);
            if (++x == V) { x = 0; ++r; }
          }
          *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{e[0], e[1], e[2], e[3]};
        }
        if (AUG)
          for (int r = sub; r < rows; r += 32) a.oaug[((long)n * a.C + br.c0 + co) * a.Tout + t0 + r] = val(r, V);
#pragma unroll
        for (int m = 0; m < MTL; ++m)
          if (m == kc) { s1[m] += l1; s2[m] += l2; }
      }
    }
    if (++kc == nchunk) { kc = 0; u += gridDim.x; }
  }
  // per-channel partial sums: the 32 threads of a row are 32 consecutive lanes
  if (a.stats) {
#pragma unroll
    for (int m = 0; m < MTL; ++m) {
      float x1 = s1[m], x2 = s2[m];
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        x1 += __shfl_xor(x1, off, 64);
        x2 += __shfl_xor(x2, off, 64);
      }
      const int co = 16 * m + co_l;
      if (sub == 0 && co < bc) {
        float* dst = a.stats + ((long)blockIdx.x * a.C + br.c0 + co) * 2;
        dst[0] = x1;
        dst[1] = x2;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// data gradient.  Work unit = (sample, tile of R INPUT frames).  The tile of dfe = gf + A0 + B0 * f the taps reach
// (output frames [t0/s - 4, t0/s + R/s + 4)) is staged like the forward's input, its global-joint column
// d o_aug[t'] = sum_v dfe[t', v] * coeff[v] is built in LDS, the transposed-weight product gives dh, and the epilogue
// turns it into dz = dh * [z*scale+shift > 0] * scale with the partial sums of d scale / d shift (global joint included)
// and of d coeff[v] = sum dfe[.., v] * o_aug.
// ---------------------------------------------------------------------------------------------------------------
template <int KT, int MTL, int ST>
__global__ __launch_bounds__(TM_NT, 2) void k_tms_dgrad(TmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const TmBranch& br = a.br[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int V = a.V, AUG = a.zaug ? 1 : 0, VL = V + AUG;
  const int R = a.R;
  constexpr int s = ST;
  const int RG = R / s + 2 * TM_H;                   // staged output frames
  const int RX = R + 2 * TM_H;                       // staged input frames (max-pool windows only)
  const int XS = tm_xs(RG * VL), XS2 = tm_xs(RX * VL);
  const int bc = br.bc, nchunk = (bc + TM_CK - 1) / TM_CK, CKW = nchunk * TM_CK, MP = MTL * 16;
  float* Gl = lds;                                   // [TM_CK][XS]   dfe tile; the bounce buffer aliases it
  float* Wl = Gl + TM_CK * XS;                       // [KT][CKW][MP] transposed weights (conv) | [TM_CK][XS2] act(z) (max-pool)
  float* Ol = Wl + (br.type == 0 ? KT * CKW * MP : (br.type == 1 ? TM_CK * XS2 : 0));     // [TM_CK][RG] o_aug
  float* Cl = Ol + TM_CK * RG;                       // coeff[V]
  float* Rl = Cl + 32;                               // [16][32] d coeff reduction
  const int DS = tm_xs(R * VL);
  const int NP = R * VL;
  const float invVL = 1.f / (float)VL, invV = 1.f / (float)V;
  // Stride 2 (round 6): a tap reaches an input frame only when (frame - shift) is even, so with the tile's positions in
  // plane order half of every tile's lanes multiplied zeros for every tap.  The 16-position MFMA tiles now hold frames of
  // ONE parity (even frames first, then odd): a (tile, tap) pair is wholly live or wholly dead, the dead half is skipped
  // (CTR-GCN's stride-2 blocks: 339 / 286 us per launch before).  PAR = that mapping; it needs tiles_e + tiles_o <= 32.
  const int Re = (R + 1) >> 1, Ro = R >> 1;
  const int tiles_e = (Re * VL + 15) >> 4, tiles_o = (Ro * VL + 15) >> 4;
  const bool PAR = s == 2 && tiles_e + tiles_o <= 8 * TM_NTW;
  const int NTU = PAR ? tiles_e + tiles_o : (NP + 15) >> 4;
  if (br.type == 0) tm_stage_w<KT, false>(br, Wl, CKW, MP, tid);
  if (AUG) for (int i = tid; i < V; i += TM_NT) Cl[i] = a.coeff[i];
  int rP[TM_NTW], xP[TM_NTW];
  bool vP[TM_NTW];                                   // this lane's position of tile i exists
#pragma unroll
  for (int i = 0; i < TM_NTW; ++i) {
    const int tile = wave + 8 * i;
    if (PAR) {
      const int cls = tile >= tiles_e ? 1 : 0, nc = (cls ? Ro : Re) * VL;
      int jj = 16 * (tile - cls * tiles_e) + (lane & 15);
      vP[i] = jj < nc;
      if (jj >= nc) jj = nc > 0 ? nc - 1 : 0;
      int fr;
      divmod_small(jj, VL, invVL, fr, xP[i]);
      rP[i] = 2 * fr + cls;
    } else {
      int P = 16 * tile + (lane & 15);
      vP[i] = P < NP;
      if (P >= NP) P = NP - 1;
      divmod_small(P, VL, invVL, rP[i], xP[i]);
    }
  }
  const int ci_l = tid >> 5, sub = tid & 31;
  float ps[MTL], pb[MTL];                            // partial sums: d scale, d shift of channel 16 m + ci_l
#pragma unroll
  for (int m = 0; m < MTL; ++m) ps[m] = pb[m] = 0.f;
  float pc = 0.f;                                    // d coeff[x], x = tid & 31, over the (tid >> 5)-strided (channel, frame) pairs
  const int kq = lane >> 4;
  const int cx = tid & 31, cg8 = tid >> 5;

  for (int u = blockIdx.x; u < a.units; u += gridDim.x) {
    const int n = u / a.tiles, t0 = (u - n * a.tiles) * R;          // first INPUT frame of the tile
    const int rows = min(R, a.T - t0);
    const int g_lo = t0 / s - TM_H;                                 // output frame of LDS row 0
    const int core = min(R / s, a.Tout - t0 / s);                   // output frames this unit owns (d coeff)
    const int nq = (rows * V) >> 2;
    f32x4 acc[TM_NTW][MTL];
#pragma unroll
    for (int i = 0; i < TM_NTW; ++i)
#pragma unroll
      for (int m = 0; m < MTL; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // epilogue of one 16-channel group: dh(r, x) -> dz, partial sums.  slot = statistics slot (m-tile / chunk)
    auto finish = [&](int c, auto&& dh, float& qs, float& qb) {
      const int cg = br.c0 + c;
      const float sc = a.scale ? a.scale[cg] : 1.f, sh = a.scale ? a.shift[cg] : 0.f;
      const bool relu = a.scale && cg < a.n_act;
      const long pbase = (((long)n * a.C + cg) * a.T + t0) * V;
      for (int q = sub; q < nq; q += 32) {
        int r, x;
        divmod_small(4 * q, V, invV, r, x);
        const f32x4 zv = *reinterpret_cast<const f32x4*>(a.z + pbase + 4 * q);
        const float zz[4] = {zv.x, zv.y, zv.z, zv.w};
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float g = dh(r, x);
          if (relu && !(fmaf(zz[i], sc, sh) > 0.f)) g = 0.f;
          qs = fmaf(g, zz[i], qs);
          qb += g;
          e[i] = g * sc;
          if (++x == V) { x = 0; ++r; }
        }
        *reinterpret_cast<f32x4*>(a.dz + pbase + 4 * q) = f32x4{e[0], e[1], e[2], e[3]};
      }
      if (AUG) {
        for (int r = sub; r < rows; r += 32) {
          const long o = ((long)n * a.C + cg) * a.T + t0 + r;
          const float zz = a.zaug[o];
          float g = dh(r, V);
          if (relu && !(fmaf(zz, sc, sh) > 0.f)) g = 0.f;
          qs = fmaf(g, zz, qs);
          qb += g;
          a.dzaug[o] = g * sc;
        }
      }
    };

    for (int kc = 0; kc < nchunk; ++kc) {
      __syncthreads();
      tm_stage<1>(a, br, Gl, XS, n, kc, g_lo, RG, a.Tout, VL, tid);
      if (AUG) {
        for (int id = tid; id < TM_CK * RG; id += TM_NT) {          // o_aug of the staged frames (zero outside the clip)
          const int cl = id / RG, r = id - cl * RG, c = kc * TM_CK + cl, t = g_lo + r;
          Ol[id] = (c < bc && t >= 0 && t < a.Tout) ? a.oaug_in[((long)n * a.C + br.c0 + c) * a.Tout + t] : 0.f;
        }
      }
      if (br.type == 1) {                                           // max-pool routing needs the forward input tile
        tm_stage<0>(a, br, Wl, XS2, n, kc, t0 - TM_H, RX, a.T, VL, tid);
        if (AUG) tm_stage_aug(a, br, Wl, XS2, n, kc, t0 - TM_H, RX, VL, tid);
      }
      __syncthreads();
      if (AUG) {
        // global-joint column of dfe and this unit's share of d coeff
        for (int id = tid; id < TM_CK * RG; id += TM_NT) {
          const int cl = id / RG, r = id - cl * RG;
          const float* row = Gl + cl * XS + r * VL;
          float acc1 = 0.f;
          for (int x = 0; x < V; ++x) acc1 = fmaf(row[x], Cl[x], acc1);
          Gl[cl * XS + r * VL + V] = acc1;
        }
        if (cx < V)
          for (int id = cg8; id < TM_CK * core; id += 16) {
            const int cl = id / core, r = id - cl * core + TM_H;
            pc = fmaf(Gl[cl * XS + r * VL + cx], Ol[cl * RG + r], pc);
          }
        __syncthreads();
      }
      if (br.type == 0) {
#pragma unroll 1
        for (int tap = 0; tap < KT; ++tap) {
          const int shr = (tap - KT / 2) * br.dil;                  // input frame = output frame * s + shr
          int offs[TM_NTW];
          bool okp[TM_NTW], livet[TM_NTW];
#pragma unroll
          for (int i = 0; i < TM_NTW; ++i) {
            const int num = rP[i] - shr;                            // (t0 is a multiple of s: the tile offset drops out)
            okp[i] = s == 1 || !(num & 1);
            offs[i] = ((s == 1 ? num : (num >> 1)) + TM_H) * VL + xP[i];
            // PAR: the parity of (frame - shift) is the tile's (wave-uniform): a dead (tile, tap) pair is skipped whole
            const int tile = wave + 8 * i;
            livet[i] = !PAR || !(((tile >= tiles_e ? 1 : 0) - shr) & 1);
          }
#pragma unroll 1
          for (int ks = 0; ks < TM_CK / 4; ++ks) {
            float av[MTL];
#pragma unroll
            for (int m = 0; m < MTL; ++m) av[m] = Wl[(tap * CKW + kc * TM_CK + 4 * ks + kq) * MP + 16 * m + (lane & 15)];
            const float* grow = Gl + (4 * ks + kq) * XS;
#pragma unroll
            for (int i = 0; i < TM_NTW; ++i) {
              if (wave + 8 * i < NTU && livet[i]) {
                const float bv = okp[i] ? grow[offs[i]] : 0.f;
#pragma unroll
                for (int m = 0; m < MTL; ++m) acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv, acc[i][m], 0, 0, 0);
              }
            }
          }
        }
      } else {
        // strided copy / max-pool: dh straight from the staged tiles, this chunk's 16 channels now
        const int c = kc * TM_CK + ci_l;
        if (c < bc) {
          const float* grow = Gl + ci_l * XS;
          const float* xrow = Wl + ci_l * XS2;
          auto dh = [&](int r, int x) -> float {
            const int t = t0 + r;
            if (br.type == 2) return (t % s == 0 && t / s < a.Tout) ? grow[(t / s - g_lo) * VL + x] : 0.f;
            // max-pool: the gradient of window t' goes to its FIRST maximal frame (ATen's order: ties to the earlier
            // frame).  Frame t can be the last frame of the window centred at t-1, the centre of its own window or the
            // first frame of the window centred at t+1 (centres are the multiples of the stride below Tout*s); it owns
            // a window iff it beats every earlier frame strictly and every later frame weakly.  Outside the clip: -inf.
            auto hv = [&](int tt) -> float { return (tt >= 0 && tt < a.T) ? xrow[(tt - t0 + TM_H) * VL + x] : -INFINITY; };
            const float v = hv(t), m2 = hv(t - 2), m1 = hv(t - 1), p1 = hv(t + 1), p2 = hv(t + 2);
            float g = 0.f;
            if (t >= 1 && (t - 1) % s == 0 && (t - 1) / s < a.Tout && v > m2 && v > m1) g += grow[((t - 1) / s - g_lo) * VL + x];
            if (t % s == 0 && t / s < a.Tout && v > m1 && v >= p1) g += grow[(t / s - g_lo) * VL + x];
            if ((t + 1) % s == 0 && (t + 1) / s < a.Tout && v >= p1 && v >= p2) g += grow[((t + 1) / s - g_lo) * VL + x];
            return g;
          };
#pragma unroll
          for (int m = 0; m < MTL; ++m)
            if (m == kc) finish(c, dh, ps[m], pb[m]);
        }
      }
    }
    if (br.type == 0) {
#pragma unroll
      for (int m = 0; m < MTL; ++m) {
        if (16 * m >= bc) break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM_NTW; ++i) {
          const int P = rP[i] * VL + xP[i];                           // plane order, whatever order the tiles hold
          if (wave + 8 * i < NTU && vP[i]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Gl[(4 * kq + r) * DS + P] = acc[i][m][r];
          }
        }
        __syncthreads();
        const int c = 16 * m + ci_l;
        if (c < bc) {
          const float* drow = Gl + ci_l * DS;
          finish(c, [&](int r, int x) -> float { return drow[r * VL + x]; }, ps[m], pb[m]);
        }
      }
    }
  }
  if (a.paff) {
#pragma unroll
    for (int m = 0; m < MTL; ++m) {
      float x1 = ps[m], x2 = pb[m];
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        x1 += __shfl_xor(x1, off, 64);
        x2 += __shfl_xor(x2, off, 64);
      }
      const int c = 16 * m + ci_l;
      if (sub == 0 && c < bc) {
        float* dst = a.paff + ((long)blockIdx.x * a.C + br.c0 + c) * 2;
        dst[0] = x1;
        dst[1] = x2;
      }
    }
  }
  if (AUG && a.pcoeff) {
    __syncthreads();
    Rl[cg8 * 32 + cx] = pc;
    __syncthreads();
    if (tid < V) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) v += Rl[g * 32 + tid];
      a.pcoeff[((long)blockIdx.y * gridDim.x + blockIdx.x) * V + tid] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient of the conv windows:  dW[co, ci, tap] = sum_{n, t', x} dfe[co, t', x] * h[ci, t'*s + (tap - KT/2)*dil, x]
// over the V joints AND the global-joint column, db[co] = sum dfe.  Work unit = (sample, R output frames): dfe tile
// Gl[co][R x VL] and haloed input tile Xl[ci][(R*s + 8) x VL] (rows of odd stride: the fragment reads walk channels),
// the K loop runs over the tile's positions, the eight waves split it and their accumulators meet in LDS at the end;
// one partial row per workgroup (dsgcn_colsum finishes).  Windows wider than 48 channels split their input-channel tiles
// over two wave pairs instead.
// ---------------------------------------------------------------------------------------------------------------
template <int KT, int MTL, int NPW>     // NPW: input-channel tiles per wave (MTL, or MTL / 2 when two wave pairs split them)
__global__ __launch_bounds__(TM_NT, 2) void k_tms_wgrad(TmArgs a, int nconv) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int bi = 0;
  for (int i = 0, k = 0; i < a.nbr; ++i)
    if (a.br[i].type == 0) { if (k == (int)blockIdx.y) bi = i; ++k; }
  const TmBranch& br = a.br[bi];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int V = a.V, AUG = a.zaug ? 1 : 0, VL = V + AUG;
  const int R = a.R, s = a.stride;
  const int RIN = R * s + 2 * TM_H;
  const int GS = (R * VL) | 1, XS = (RIN * VL) | 1;
  const int bc = br.bc, MP = MTL * 16;
  constexpr int NSPLIT = MTL / NPW;                  // wave groups over the input-channel tiles (1 or 2)
  constexpr int KSPLIT = 8 / NSPLIT;                 // waves sharing the positions
  float* Gl = lds;                                   // [MP][GS]
  float* Xl = Gl + MP * GS;                          // [MP][XS]
  float* Pl = Xl + MP * XS;                          // [R * VL] position -> tile offset (r * s * VL + x)
  float* Cl = Pl + ((R * VL + 3) & ~3);              // coeff[V]
  const int NP = R * VL;
  const float invVL = 1.f / (float)VL;
  for (int i = tid; i < NP; i += TM_NT) {
    int r, x;
    divmod_small(i, VL, invVL, r, x);
    reinterpret_cast<int*>(Pl)[i] = r * s * VL + x;
  }
  if (AUG) for (int i = tid; i < V; i += TM_NT) Cl[i] = a.coeff[i];
  const int kq = lane >> 4, l15 = lane & 15;
  const int ng = wave / KSPLIT, kw = wave - ng * KSPLIT;             // input-tile group, position share
  f32x4 acc[MTL][NPW][KT];
#pragma unroll
  for (int m = 0; m < MTL; ++m)
#pragma unroll
    for (int j = 0; j < NPW; ++j)
#pragma unroll
      for (int k = 0; k < KT; ++k) acc[m][j][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbl = 0.f;                                   // d bias of channel tid (tid < bc)

  for (int u = blockIdx.x; u < a.units; u += gridDim.x) {
    const int n = u / a.tiles, t0 = (u - n * a.tiles) * R;          // first OUTPUT frame of the tile
    __syncthreads();
    // stage all channels of the window, TM_CK at a time (the staging helpers work on 16-channel chunks)
    for (int kc = 0; kc * TM_CK < bc; ++kc) {
      tm_stage<1>(a, br, Gl + kc * TM_CK * GS, GS, n, kc, t0, R, a.Tout, VL, tid);
      tm_stage<0>(a, br, Xl + kc * TM_CK * XS, XS, n, kc, t0 * s - TM_H, RIN, a.T, VL, tid);
      if (AUG) tm_stage_aug(a, br, Xl + kc * TM_CK * XS, XS, n, kc, t0 * s - TM_H, RIN, VL, tid);
    }
    __syncthreads();
    if (AUG) {
      for (int id = tid; id < MP * R; id += TM_NT) {
        const int cl = id / R, r = id - cl * R;
        const float* row = Gl + cl * GS + r * VL;
        float acc1 = 0.f;
        for (int x = 0; x < V; ++x) acc1 = fmaf(row[x], Cl[x], acc1);
        Gl[cl * GS + r * VL + V] = acc1;
      }
      __syncthreads();
    }
    if (tid < bc) {                                                 // d bias: every position of the tile, global joint included
      const float* row = Gl + tid * GS;
      float v = 0.f;
      for (int p = 0; p < NP; ++p) v += row[p];
      dbl += v;
    }
    for (int p0 = 4 * kw; p0 < NP; p0 += 4 * KSPLIT) {
      const int p = p0 + kq;
      const bool live = p < NP;
      const int po = live ? reinterpret_cast<const int*>(Pl)[p] : 0;
      float av[MTL];
#pragma unroll
      for (int m = 0; m < MTL; ++m) av[m] = live ? Gl[(16 * m + l15) * GS + p] : 0.f;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const int off = po + (TM_H + (k - KT / 2) * br.dil) * VL;
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
          const float bv = live ? Xl[(16 * (ng * NPW + j) + l15) * XS + off] : 0.f;
#pragma unroll
          for (int m = 0; m < MTL; ++m) acc[m][j][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv, acc[m][j][k], 0, 0, 0);
        }
      }
    }
  }
  // sum the KSPLIT position shares through LDS, one (m, j, tap) tile at a time, and write the partial row
  float* dw = br.dwp + (long)blockIdx.x * a.pstride;
  float* Rs = lds;                                   // [8 waves][16][17]
#pragma unroll
  for (int m = 0; m < MTL; ++m)
#pragma unroll
    for (int j = 0; j < NPW; ++j)
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) Rs[(wave * 16 + 4 * kq + r) * 17 + l15] = acc[m][j][k][r];
        __syncthreads();
        // thread -> (group ng2, row co, col ci) of the 16x16 tile(s): NSPLIT * 256 outputs, 256 threads
        for (int o = tid; o < NSPLIT * 256; o += TM_NT) {
          const int g2 = o >> 8, rr = (o >> 4) & 15, cc = o & 15;
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < KSPLIT; ++w) v += Rs[((g2 * KSPLIT + w) * 16 + rr) * 17 + cc];
          const int co = 16 * m + rr, ci = 16 * (g2 * NPW + j) + cc;
          if (co < bc && ci < bc) dw[((long)co * bc + ci) * KT + k] = v;
        }
      }
  if (tid < bc) br.dbp[(long)blockIdx.x * a.pstride + tid] = dbl;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
namespace {

int g_tm_dbg = 0, g_tm_rows = 0, g_tm_R = 0;       // lab-only knobs (dsgcn_tms_tuning); 0 = defaults

struct TmPlan {
  int ok, R, Rw, tiles_o, tiles_i, tiles_w, gx_f, gx_d, gx_w, mtl, nconv, VL;
  size_t lds_f, lds_d, lds_w;
};

// geometry shared by the three launches and by dsgcn_tms_rows (the caller sizes its partial buffers from it)
TmPlan tm_plan(int n, int C, int T, int V, int stride, int KT, int nbr, const int* type, const int* bc, const int* dil,
               int aug) {
  TmPlan p = {};
  if (n <= 0 || C <= 0 || T <= 0 || V <= 0 || nbr <= 0 || nbr > TM_MAXBR || !type || !bc || !dil) return p;
  if ((KT != 3 && KT != 5) || (stride != 1 && stride != 2)) return p;
  const int Tout = (T + stride - 1) / stride;
  if (((long)T * V) % 4 || ((long)Tout * V) % 4 || (stride == 2 && (T & 1))) return p;
  const int VL = V + (aug ? 1 : 0);
  if (V > 32) return p;                              // d coeff reduction: one thread per joint in 32-wide groups
  if ((long)TM_CK * T * V * 4 >= (1L << 31)) return p;       // 32-bit offsets inside a 16-channel slab
  int maxbc = 0, maxconv = 0, nconv = 0, maxpool = 0;
  for (int i = 0; i < nbr; ++i) {
    if (bc[i] <= 0 || bc[i] > 64) return p;
    maxbc = std::max(maxbc, bc[i]);
    if (type[i] == 0) {
      ++nconv;
      maxconv = std::max(maxconv, bc[i]);
      if (dil[i] < 1 || dil[i] * (KT / 2) > TM_H) return p;
    } else if (type[i] == 1) {
      maxpool = 1;
    } else if (type[i] != 2) {
      return p;
    }
  }
  p.mtl = (maxbc + 15) / 16;
  p.nconv = nconv;
  p.VL = VL;
  // frames per tile: 16 where the positions of a tile fit the per-wave accumulators and the launch still fills the chip
  int R = 16;
  if (R * VL > 128 * TM_NTW) R = 8;
  if (R * VL > 128 * TM_NTW) return p;
  if (R == 16 && (long)n * ((Tout + 15) / 16) * nbr < 1024 && Tout > 8) R = 8;
  if (R * VL > 128 * TM_NTW) return p;
  // the staged tile of a 16-channel chunk must fit the per-thread prefetch slots (forward: R*stride + 8 frames)
  while (R > 4 && (long)TM_CK * ((R * stride + 2 * TM_H) * V / 4) > (long)TM_NT * TM_NPF) R -= 4;
  if ((long)TM_CK * ((R * stride + 2 * TM_H) * V / 4) > (long)TM_NT * TM_NPF || (long)TM_CK * (R * stride + 2 * TM_H) > 2L * TM_NT)
    return p;
  if (stride == 2 && (R / 2) % 4) return p;          // the backward's staged range must start at a multiple of 4 frames
  if (g_tm_R > 0) R = g_tm_R;
  p.R = R;
  p.tiles_o = (Tout + R - 1) / R;
  p.tiles_i = (T + R - 1) / R;
  const int CKW = ((maxconv + TM_CK - 1) / TM_CK) * TM_CK, MP = p.mtl * 16;
  const size_t wl = nconv ? (size_t)KT * CKW * MP : 0;
  const int RIN = R * stride + 2 * TM_H, RG = R / stride + 2 * TM_H, RX = R + 2 * TM_H;
  p.lds_f = ((size_t)TM_CK * tm_xs(RIN * VL) + wl + 32) * sizeof(float);
  const size_t x2 = maxpool ? (size_t)TM_CK * tm_xs(RX * VL) : 0;
  p.lds_d = ((size_t)TM_CK * tm_xs(RG * VL) + std::max(wl, x2) + (size_t)TM_CK * RG + 32 + 16 * 32) * sizeof(float);
  const int gxt = g_tm_rows > 0 ? g_tm_rows : std::max(1, 1024 / nbr);
  p.gx_f = (int)std::min<long>((long)n * p.tiles_o, gxt);
  p.gx_d = (int)std::min<long>((long)n * p.tiles_i, gxt);
  if (nconv) {
    const int MPc = ((maxconv + 15) / 16) * 16;
    p.Rw = MPc <= 16 ? 8 : 4;
    if (p.Rw * VL > 512) return p;
    p.tiles_w = (Tout + p.Rw - 1) / p.Rw;
    const int RINw = p.Rw * stride + 2 * TM_H;
    p.lds_w = ((size_t)MP * (((p.Rw * VL) | 1) + ((RINw * VL) | 1)) + ((p.Rw * VL + 3) & ~3) + 32) * sizeof(float);
    p.lds_w = std::max(p.lds_w, (size_t)8 * 16 * 17 * sizeof(float));
    p.gx_w = (int)std::min<long>((long)n * p.tiles_w, std::max(1, 512 / nconv));
    if (p.lds_w > TM_LDS_MAX) return p;
  }
  if (p.lds_f > TM_LDS_MAX || p.lds_d > TM_LDS_MAX) return p;
  p.ok = 1;
  return p;
}

int tm_fill(TmArgs& a, int nbr, const int* type, const int* c0, const int* bc, const int* dil, const float* const* w,
            const float* const* b) {
  for (int i = 0; i < nbr; ++i) {
    TmBranch& t = a.br[i];
    t.type = type[i]; t.c0 = c0[i]; t.bc = bc[i]; t.dil = dil[i];
    t.w = w ? w[i] : nullptr; t.b = b ? b[i] : nullptr;
    if (t.c0 < 0 || t.c0 + t.bc > a.C) return DSGCN_EINVAL;
    if (t.type == 0 && w && !t.w) return DSGCN_EINVAL;
  }
  a.nbr = nbr;
  return 0;
}

template <typename F>
int tm_lds(F* kernel, size_t lds) {
  if (lds > 64 * 1024) {       // not a stream operation; idempotent, issued before the launch (outside graph capture on
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TM_LDS_MAX);   // the first eager call)
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

#define TM_LAUNCH(KERNEL, GRID, LDS, ...)                                                   \
  do {                                                                                      \
    const int rc_ = tm_lds(KERNEL, LDS);                                                    \
    if (rc_) return rc_;                                                                    \
    hipLaunchKernelGGL(KERNEL, GRID, dim3(TM_NT), LDS, (hipStream_t)stream, __VA_ARGS__);  \
  } while (0)

#define TM_SWITCH_MTL(MTLV, BODY)                         \
  switch (MTLV) {                                         \
    case 1: { constexpr int M_ = 1; BODY; } break;        \
    case 2: { constexpr int M_ = 2; BODY; } break;        \
    case 3: { constexpr int M_ = 3; BODY; } break;        \
    case 4: { constexpr int M_ = 4; BODY; } break;        \
    default: return DSGCN_EUNSUPPORTED;                   \
  }

}  // namespace

extern "C" {

#ifdef DSGCN_LAB
int dsgcn_tms_tuning(int key, int value) {
  if (key == 0) { g_tm_dbg = value; return 0; }      // ablation mask
  if (key == 1) { g_tm_rows = value; return 0; }     // workgroups per window (0 = default)
  if (key == 2) { g_tm_R = value; return 0; }        // frames per tile (0 = default)
  return DSGCN_EINVAL;
}
#endif

// Partial-row counts of the fused temporal stage for a shape (0 = the shape is not eligible: use the staged kernels).
// which: 0 forward statistics rows, 1 data-gradient rows (paff; pcoeff has rows * nbr), 2 weight-gradient rows.
int dsgcn_tms_rows(int which, int n, int C, int T, int V, int stride, int KT, int nbr, const int* type, const int* bc,
                   const int* dil, int aug) {
  const TmPlan p = tm_plan(n, C, T, V, stride, KT, nbr, type, bc, dil, aug);
  if (!p.ok) return 0;
  return which == 0 ? p.gx_f : (which == 1 ? p.gx_d : p.gx_w);
}

int dsgcn_tms_fwd(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                  const float* coeff, float* f, float* oaug, float* stats, int n, int C, int T, int V, int stride, int KT,
                  int nbr, const int* type, const int* c0, const int* bc, const int* dil, const float* const* w,
                  const float* const* b, void* stream) {
  if (!z || !f || (zaug && (!coeff || !oaug)) || (scale && !shift)) return DSGCN_EINVAL;
  const TmPlan p = tm_plan(n, C, T, V, stride, KT, nbr, type, bc, dil, zaug != nullptr);
  if (!p.ok) return DSGCN_EUNSUPPORTED;
  TmArgs a = {};
  a.z = z; a.zaug = zaug; a.scale = scale; a.shift = shift; a.coeff = coeff; a.f = f; a.oaug = oaug; a.stats = stats;
  a.n = n; a.C = C; a.T = T; a.Tout = (T + stride - 1) / stride; a.V = V; a.stride = stride; a.n_act = n_act;
  a.R = p.R; a.tiles = p.tiles_o; a.units = n * p.tiles_o; a.dbg = g_tm_dbg;
  const int rc = tm_fill(a, nbr, type, c0, bc, dil, w, b);
  if (rc) return rc;
  const dim3 grid((unsigned)p.gx_f, (unsigned)nbr);
  if (KT == 3) {
    TM_SWITCH_MTL(p.mtl, TM_LAUNCH((k_tms_fwd<3, M_>), grid, p.lds_f, a));
  } else {
    TM_SWITCH_MTL(p.mtl, TM_LAUNCH((k_tms_fwd<5, M_>), grid, p.lds_f, a));
  }
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_tms_dgrad(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                    const float* coeff, const float* gf, const float* f, const float* A0, const float* B0,
                    const float* oaug, float* dz, float* dzaug, float* paff, float* pcoeff, int n, int C, int T, int V,
                    int stride, int KT, int nbr, const int* type, const int* c0, const int* bc, const int* dil,
                    const float* const* w, void* stream) {
  if (!z || !gf || !dz || (zaug && (!coeff || !oaug || !dzaug)) || (A0 && (!B0 || !f)) || (scale && !shift))
    return DSGCN_EINVAL;
  const TmPlan p = tm_plan(n, C, T, V, stride, KT, nbr, type, bc, dil, zaug != nullptr);
  if (!p.ok) return DSGCN_EUNSUPPORTED;
  TmArgs a = {};
  a.z = z; a.zaug = zaug; a.scale = scale; a.shift = shift; a.coeff = coeff; a.gf = gf; a.fin = f; a.A0 = A0; a.B0 = B0;
  a.oaug_in = oaug; a.dz = dz; a.dzaug = dzaug; a.paff = paff; a.pcoeff = pcoeff;
  a.n = n; a.C = C; a.T = T; a.Tout = (T + stride - 1) / stride; a.V = V; a.stride = stride; a.n_act = n_act;
  a.R = p.R; a.tiles = p.tiles_i; a.units = n * p.tiles_i;
  const int rc = tm_fill(a, nbr, type, c0, bc, dil, w, nullptr);
  if (rc) return rc;
  const dim3 grid((unsigned)p.gx_d, (unsigned)nbr);
  if (KT == 3 && stride == 1) {
    TM_SWITCH_MTL(p.mtl, TM_LAUNCH((k_tms_dgrad<3, M_, 1>), grid, p.lds_d, a));
  } else if (KT == 3) {
    TM_SWITCH_MTL(p.mtl, TM_LAUNCH((k_tms_dgrad<3, M_, 2>), grid, p.lds_d, a));
  } else if (stride == 1) {
    TM_SWITCH_MTL(p.mtl, TM_LAUNCH((k_tms_dgrad<5, M_, 1>), grid, p.lds_d, a));
  } else {
    TM_SWITCH_MTL(p.mtl, TM_LAUNCH((k_tms_dgrad<5, M_, 2>), grid, p.lds_d, a));
  }
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// Conv window i writes row r of its weight / bias partials at dwp[i] + r * pstride / dbp[i] + r * pstride, r < rows(2)
int dsgcn_tms_wgrad(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                    const float* coeff, const float* gf, const float* f, const float* A0, const float* B0, int n, int C,
                    int T, int V, int stride, int KT, int nbr, const int* type, const int* c0, const int* bc,
                    const int* dil, float* const* dwp, float* const* dbp, int pstride, void* stream) {
  if (!z || !gf || !dwp || !dbp || (zaug && !coeff) || (A0 && (!B0 || !f)) || (scale && !shift)) return DSGCN_EINVAL;
  TmPlan p = tm_plan(n, C, T, V, stride, KT, nbr, type, bc, dil, zaug != nullptr);
  if (!p.ok || !p.nconv) return DSGCN_EUNSUPPORTED;
  TmArgs a = {};
  a.z = z; a.zaug = zaug; a.scale = scale; a.shift = shift; a.coeff = coeff; a.gf = gf; a.fin = f; a.A0 = A0; a.B0 = B0;
  a.n = n; a.C = C; a.T = T; a.Tout = (T + stride - 1) / stride; a.V = V; a.stride = stride; a.n_act = n_act;
  a.R = p.Rw; a.tiles = p.tiles_w; a.units = n * p.tiles_w; a.pstride = pstride;
  const int rc = tm_fill(a, nbr, type, c0, bc, dil, nullptr, nullptr);
  if (rc) return rc;
  int maxconv = 0;
  for (int i = 0; i < nbr; ++i) {
    a.br[i].dwp = dwp[i]; a.br[i].dbp = dbp[i];
    if (type[i] == 0) {
      if (!dwp[i] || !dbp[i]) return DSGCN_EINVAL;
      maxconv = std::max(maxconv, bc[i]);
    }
  }
  const int mtl = (maxconv + 15) / 16;
  const dim3 grid((unsigned)p.gx_w, (unsigned)p.nconv);
  if (mtl == 3 && KT == 5) {                         // runs on the four-tile kernel (below): its LDS rows are 64 wide
    const int VLw = V + (zaug ? 1 : 0), RINw = p.Rw * stride + 2 * TM_H;
    const size_t need = ((size_t)64 * (((p.Rw * VLw) | 1) + ((RINw * VLw) | 1)) + ((p.Rw * VLw + 3) & ~3) + 32) * sizeof(float);
    if (need > TM_LDS_MAX) return DSGCN_EUNSUPPORTED;
    if (need > p.lds_w) p.lds_w = need;
  }
  // five taps x four row tiles: with two input-channel tiles per wave the 160 accumulator registers spilled (71 VGPRs,
  // round-4 metadata); one tile per wave — four wave groups over the input tiles, two waves sharing the positions — holds 80
#define TM_W(KTV, NPW4)                                                                        \
  switch (mtl) {                                                                               \
    case 1: TM_LAUNCH((k_tms_wgrad<KTV, 1, 1>), grid, p.lds_w, a, p.nconv); break;             \
    case 2: TM_LAUNCH((k_tms_wgrad<KTV, 2, 2>), grid, p.lds_w, a, p.nconv); break;             \
    case 3: if (KTV == 3) { TM_LAUNCH((k_tms_wgrad<3, 3, 3>), grid, p.lds_w, a, p.nconv); break; }   \
            /* five taps x three tiles per wave spilled 100 registers: 33..48-channel windows take the four-tile form */ \
    case 4: TM_LAUNCH((k_tms_wgrad<KTV, 4, NPW4>), grid, p.lds_w, a, p.nconv); break;          \
    default: return DSGCN_EUNSUPPORTED;                                                        \
  }
  if (KT == 3) { TM_W(3, 2) } else { TM_W(5, 1) }
#undef TM_W
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
