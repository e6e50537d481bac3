// K-D, split layout: the multi-scale temporal stage of dgmstcn (reference: pyskl/models/gcns/utils/tcn.py:379-428) between
// the unit's two 1x1 convs WITHOUT its (V+1)-column intermediates.  tcn.py:409 appends the global joint (the mean over the
// joints) as a 26th column, runs BatchNorm + ReLU and the six temporal branches on the widened tensor and folds the extra
// column back with `x[..., :V] + einsum(x[..., V], add_coeff)` (tcn.py:416-420).  Every stage in between is column-wise, so
// the stage splits exactly into a V-column part on the (n,C,T,V) layout and a one-column part on (n,C,T) tensors, and the
// conv windows being linear, the one-column part of their forward folds into the operand:
//       o[..., v] + oaug * coeff[v] = conv(h[..., v] + haug * coeff[v]) + bias * (1 + coeff[v]).
//
//   forward   k_tsp<true>   : ONE launch.  Conv windows on the f32 matrix core, B operand = relu(z*scale+shift) + haug*coeff
//                             formed while loading z (16-byte planes; a tap shifts by dil*V positions, an ODD shift is read
//                             with two 4-byte-aligned 8-byte loads per side tap and the two plane-end cases fixed up by
//                             select; haug of the block's frames sits in LDS), epilogue f = acc + bias*(1 + coeff) + the batch
//                             statistics of f (transform.0's BatchNorm, tcn.py:401) — h, o and the old branch_act / combine
//                             passes never exist; the same code on T x 1 "planes" as extra blocks gives oaug (kept for the
//                             backward's d coeff); max-pool / pass-through windows as whole-plane waves in the same launch
//   backward  k_tsp_prep    : ge = gf + A0 + B0*f (one materialised copy: it feeds the data AND the weight gradient), its
//                             global-joint column doaug = sum_v ge*coeff, d coeff partials
//             k_tsp<false>  : ONE launch.  Transposed windows over ge; epilogue dz = relu'(z*scale+shift) * acc * scale and
//                             the sums of the branch BatchNorm's backward (sum dpre*z, sum dpre) — dh / branch_act_bwd gone;
//                             extra blocks do the same for the one-column part (doaug -> dzaug)
//             k_tspw        : weight gradient, k_tapw's scheme (csrc/tapconv.hip) on the split operands: h is formed from z
//                             on the way into LDS, the unit's four global-joint values ride as one more group of the
//                             matrix loop, consecutive windows that fit one 32 x 32 tile share a block (per-lane tap shift)
// Stride 1, kernel 3, windows <= 64 channels, dilation <= 4, V odd, T % 4 == 0, T*V <= 2048; everything else stays on the
// staged path (branch_act -> tapconv -> combine).  Measured: profiles/r04/README.md.
#include <algorithm>

#include "common.h"
#include "bn_jobs.h"

namespace {

constexpr int TS_NT = 256;
constexpr int TS_MAXBR = 8;
constexpr int TS_OOB = 0x7ffffff0;
constexpr int TS_R = 4, TS_H = 4;                 // weight gradient: frames per unit, halo frames per side (max dilation)

typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

struct TSBranch {
  int type;            // 0 conv, 1 max (3,1), 2 pass-through
  int c0, bc, dil;     // channel window (input and output coincide), dilation
  const float* w;      // (bc, bc, 3, 1)
  const float* b;      // (bc) or NULL
  float* dwp;          // (splits, bc*bc*3) partials
  float* dbp;          // (splits, bc)
};

struct TSArgs {
  const float* z; const float* zaug; const float* scale; const float* shift;
  const float* coeff;
  float* f; float* oaug; float* stats;            // forward outputs ((n,C,T,V), (n,C,T)+1 float, (rows,C,2))
  const float* ge; const float* doaug;            // backward inputs
  float* dz; float* dzaug; float* part;           // backward outputs ((rows,C,2): sum dpre*x, sum dpre)
  int n_act, n, C, T, Tout, V, nbr, ngrp, ngrpa, nconv, eplanes, sboff, haw, splits, pstride;   // T: input frames; Tout = T / stride
  int cw[TS_MAXBR];    // table index of conv window w
  int ngroups, gfirst[TS_MAXBR], gcount[TS_MAXBR];   // weight gradient: consecutive conv windows sharing one (co x ci) tile
  int exp;             // lab builds only: timing experiments that skip parts of the work (0 in the product)
  TSBranch br[TS_MAXBR];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ts_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ts_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ f32x2s ts_load2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x2s, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void ts_store4(f32x4 v, __amdgpu_buffer_rsrc_t r, int voff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), r, voff, 0, 0);
}
__device__ __forceinline__ int ts_row32(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
// vector offset of a per-row access (the row depends on the lane's half: it must not reach the scalar operand)
__device__ __forceinline__ int ts_rowoff(bool ok, int base, int rowoff) {
  return (int)((unsigned)(ok ? base : TS_OOB) + (unsigned)rowoff);
}
__device__ __forceinline__ int ts_cp(int bc) { return (min(64, max(bc, 1)) + 7) & ~7; }

__device__ __forceinline__ const TSBranch& ts_conv_window(const TSArgs& a, int w) { return a.br[a.cw[w]]; }

// Sum of half of row l31 of a wave's [32][36] LDS tile (lane (half, l31); the caller adds the two halves).
template <typename ACC>
__device__ __forceinline__ ACC ts_rowread(const float* Tw, int half, int l31) {
  const f32x4* rowp = reinterpret_cast<const f32x4*>(Tw + l31 * 36 + half * 16);
  ACC s = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = rowp[q];
    s += ((ACC)v.x + (ACC)v.y) + ((ACC)v.z + (ACC)v.w);
  }
  return s;
}

// ---------------------------------------------------------------------------------------------------------------
// conv window, forward (FWD) or data gradient: four waves x 128 positions of the planes, all samples back to back; a lane
// owns four consecutive positions of one channel row (tap4's scheme).  AUG = false: the V-column planes (n, C, T, V);
// AUG = true: the same code on the global-joint column, "planes" of T x 1 (zaug -> oaug, doaug -> dzaug), as extra
// blocks of the same launch.
//
// Forward, V columns: the conv is linear and add_coeff multiplies whole columns, so
//       o[..., v] + oaug * coeff[v] = conv(h[..., v] + haug * coeff[v]) + bias * (1 + coeff[v]):
// the global-joint term is folded into the OPERAND (haug of the block's <= 22 frames + halo, activated, sits in LDS) and
// the epilogue needs no second tensor — f = acc + bias * (1 + coeff[v]) and the statistics of f.  (oaug, which only the
// backward's d coeff needs, comes from the AUG blocks.)
//
// LDS: [0, sboff) = the weight tile [co][tap*CP + ci], reused by the epilogue's transposes; then [64] (scale, shift),
// [64] bias, [64][32] haug.
// ---------------------------------------------------------------------------------------------------------------

#ifdef DSGCN_LAB
// wall-clock stamps (10 ns) of one V-column conv block of k_tsp (block g_tsc_block, thread 0): start, operand prefetch +
// staging requests issued, staging done (barrier), main loop done, epilogue done; [15] = count (dsgcn_tms_split_phases(1))
__device__ long long g_tsc_stamp[16];
__device__ int g_tsc_block = 0;
#define TSC_STAMP() do { if (!AUG && (int)blockIdx.x == g_tsc_block && threadIdx.x == 0 && nst < 14) g_tsc_stamp[nst++] = wall_clock64(); } while (0)
#else
#define TSC_STAMP() do {} while (0)
#endif

// S2: the stage at stride 2 (T' = T / 2 output frames; forward tiles / stores run over the OUTPUT planes and read input
// frame 2t' + (tap-1)*dil, the data gradient tiles the INPUT planes and reads output frame (t - (tap-1)*dil) / 2 where that is
// whole).  A lane's four consecutive positions lie in at most two frames (nx marks the second):
//   forward, V columns: per tap two 16-byte loads — LA at the first frame's source row, LB one source frame pair further —
//     element k comes from LB where nx[k], LA otherwise (buffer loads are range-checked per dword: a run past the tensor's
//     end reads zeros for the dwords outside, tools/lab/oob_check.hip; a run that STARTS before the tensor does not, so
//     sources are never addressed below their first element);
//   data gradient, V columns: consecutive frames have opposite parity, so per tap only the lane's first-frame elements OR
//     its second-frame elements have a source: ONE 16-byte load, the others masked (ge / doaug carry a front pad of 32
//     floats so that the second-frame run of a plane's first row may start before the plane);
//   the one-column part (V = 1: four consecutive frames): four 4-byte loads per tap.
template <bool FWD, int MT, bool ODD, bool AUG, bool S2 = false>
__device__ __forceinline__ void ts_conv(const TSArgs& a, const TSBranch& br, bool first, float* lds, int grp, int lane, int wave) {
  constexpr int KT = 3, PDK = 2, NS = KT * PDK;
  constexpr bool FOLD = FWD && !AUG;               // the operand carries haug * coeff
  constexpr int FP = 32;                           // front pad (floats) of the data gradient's sources at stride 2
  static_assert(!(S2 && ODD), "stride 2 has its own load forms");
#ifdef DSGCN_LAB
  int nst = 0;
#endif
  TSC_STAMP();
  const int tid = threadIdx.x;
  const int half = lane >> 5, l31 = lane & 31;
  const int V = AUG ? 1 : a.V, C = a.C, bc = br.bc;
  const int T = (S2 && FWD) ? a.Tout : a.T;        // frames of the tiled / destination planes
  const int Ts = S2 ? (FWD ? a.T : a.Tout) : T;    // frames of the source planes
  const int L = T * V, L4 = L * 4, Ls = Ts * V, Ls4 = Ls * 4;
  const int CP = ts_cp(bc), S = KT * CP + 1;
  float* Ws = lds;
  f32x2s* SB = reinterpret_cast<f32x2s*>(lds + a.sboff);   // [64] (scale, shift): zero past the window
  float* BI = lds + a.sboff + 128;                         // [64] bias
  float* HA = lds + a.sboff + 192;                         // [64][haw]: frames Gs .. Gs + haw of the block's channels
  const int HAW = a.haw;
  const bool relu = br.c0 < a.n_act;
  const int wt = grp * 4 + wave;
  const long total = (long)a.n * L;
  const bool wlive = (long)wt * 128 < total;
  const int g0 = wlive ? wt * 128 : 0;
  const int n0 = g0 / L;
  int p = g0 - n0 * L + 4 * l31, ds = 0;
  while (p >= L) { p -= L; ++ds; }
  const bool pok = wlive && n0 + ds < a.n;
  const float* srcp = FWD ? (AUG ? a.zaug : a.z) : (AUG ? a.doaug : a.ge);
  const int spad = (S2 && !FWD) ? FP : 0;
  const __amdgpu_buffer_rsrc_t rs = ts_rsrc(srcp - spad, ((size_t)a.n * C * Ls + spad) * 4);
  const int rowbase = ((n0 + ds) * C + br.c0 + half) * Ls + spad;
  int t0 = 0, v0 = 0;                              // frame / joint of the lane's first position
  bool nx[4] = {false, false, false, false};       // element k lies in frame t0 + 1
  if constexpr (FOLD || S2) {
    divmod_small(p, V, 1.f / (float)V, t0, v0);
#pragma unroll
    for (int k = 0; k < 4; ++k) nx[k] = v0 + k >= V;
  }
  // Stride 1.  A side tap shifts by +-dil*V positions.  Even shift: two aligned 8-byte pairs, each wholly inside or outside
  // the plane.  Odd shift: the pairs are only 4-byte aligned and one of them can straddle a plane end — (-1, 0) is read as
  // (0, 1) and (L-1, L) as (L-2, L-1), both wholly inside the tensor, and the wanted element moved over by a select.
  const int sh = (FWD ? br.dil : -br.dil) * V;
  int vP[KT][S2 ? 4 : 2];                          // S2: [0] = LA / the one load, [1] = LB; one-column part: four elements
  float mk[KT][4];
  bool fS[KT][2], fE[KT][2];
  if constexpr (!S2) {
#pragma unroll
    for (int tap = 0; tap < KT; ++tap)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = p + (tap - 1) * sh + 2 * j;
        const bool e0 = q >= 0 && q < L, e1 = q + 1 >= 0 && q + 1 < L;
        const bool s = ODD && !e0 && e1, e = ODD && e0 && !e1;
        const int qa = s ? q + 1 : (e ? q - 1 : q);
        fS[tap][j] = s;
        fE[tap][j] = e;
        vP[tap][j] = (pok && (e0 || e1)) ? (rowbase + qa) * 4 : TS_OOB;
        mk[tap][2 * j] = (pok && e0) ? 1.f : 0.f;
        mk[tap][2 * j + 1] = (pok && e1) ? 1.f : 0.f;
      }
  } else {
#pragma unroll
    for (int tap = 0; tap < KT; ++tap) {
      const int sft = (tap - 1) * br.dil;
      fS[tap][0] = fS[tap][1] = fE[tap][0] = fE[tap][1] = false;
      if constexpr (AUG) {
        // four consecutive frames p + k of a T x 1 plane
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int src;
          bool ok;
          if (FWD) {
            src = 2 * (p + k) + sft;
            ok = src >= 0 && src < Ts;
          } else {
            const int num = p + k - sft;
            src = num >> 1;
            ok = num >= 0 && !(num & 1) && src < Ts;
          }
          ok = ok && pok;
          vP[tap][k] = ok ? (rowbase + src) * 4 : TS_OOB;
          mk[tap][k] = ok ? 1.f : 0.f;
        }
      } else if (FWD) {
        const int fA = 2 * t0 + sft, fB = fA + 2;
        const bool okA = pok && fA >= 0 && fA < Ts, okB = pok && fB >= 0 && fB < Ts;
        vP[tap][0] = okA ? (rowbase + fA * V + v0) * 4 : TS_OOB;
        vP[tap][1] = okB ? (rowbase + fA * V + v0 + V) * 4 : TS_OOB;       // element k of frame t0 + 1: (fA + 2)*V + v0 + k - V
        vP[tap][2] = vP[tap][3] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) mk[tap][k] = (nx[k] ? okB : okA) ? 1.f : 0.f;
      } else {
        const int numA = t0 - sft, numB = t0 + 1 - sft;
        const bool useA = !(numA & 1);
        const int num = useA ? numA : numB;
        const int fs = num >> 1;
        const bool ok = pok && num >= 0 && fs < Ts;
        // first-frame elements: source index fs*V + v0 + k;  second-frame elements: fs*V + v0 + k - V
        vP[tap][0] = ok ? (rowbase + fs * V + v0 - (useA ? 0 : V)) * 4 : TS_OOB;
        vP[tap][1] = vP[tap][2] = vP[tap][3] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) mk[tap][k] = (ok && (nx[k] != useA)) ? 1.f : 0.f;
      }
    }
  }
  auto load = [&](int tap, int ks) -> f32x4 {
    const int soff = 2 * ks * Ls4;
    if constexpr (S2) {
      if constexpr (AUG)
        return f32x4{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vP[tap][0], soff, 0)),
                     __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vP[tap][1], soff, 0)),
                     __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vP[tap][2], soff, 0)),
                     __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vP[tap][3], soff, 0))};
      else
        return ts_load4(rs, vP[tap][0], soff);
    } else {
      if (tap == 1) return ts_load4(rs, vP[1][0], soff);
      const f32x2s lo = ts_load2(rs, vP[tap][0], soff), hi = ts_load2(rs, vP[tap][1], soff);
      return f32x4{lo.x, lo.y, hi.x, hi.y};
    }
  };
  auto loadB = [&](int tap, int ks) -> f32x4 { return ts_load4(rs, vP[tap][1], 2 * ks * Ls4); };   // forward, stride 2, V columns
  constexpr bool TWO = S2 && FWD && !AUG;

  // V columns, forward: add_coeff of the lane's four joints and the lane's frame inside the block's haug window
  float cf[4] = {0.f, 0.f, 0.f, 0.f};
  int gl = 0;
  // first frame of the window, in SOURCE frames counted over all samples (stride 2: source frame = 2 * output frame)
  const int Gs = (S2 ? 2 : 1) * ((grp * 512) / V) - TS_H;
  if constexpr (FOLD) {
#pragma unroll
    for (int k = 0; k < 4; ++k) cf[k] = a.coeff[nx[k] ? v0 + k - V : v0 + k];
    gl = pok ? (S2 ? 2 : 1) * ((n0 + ds) * T + t0) - Gs : TS_H;
  }
  const int KS2 = ((bc + 3) >> 2) << 1;                      // k-steps of two channels, even (weights are zero past bc)
  f32x4 buf[NS], bufB[TWO ? NS : 1];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    buf[u] = load(u % KT, u / KT);
    if constexpr (TWO) bufB[u] = loadB(u % KT, u / KT);
    __builtin_amdgcn_sched_barrier(0);
  }
  // Staging (the operand prefetch above is already in flight: the block pays ONE memory round trip before its first
  // product, not one per stage): weights [co][tap*CP + ci] zero padded, per-channel (scale, shift) and bias, haug.
  {
    const int run = bc * KT, wtotal = bc * run;
    float wv[8], hv[FOLD ? 8 : 1];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = tid + q * TS_NT;
      wv[q] = i < wtotal ? br.w[i] : 0.f;
    }
    f32x2s sb = {0.f, 0.f};
    float bi = 0.f;
    if (tid < bc) {
      const int c = br.c0 + tid;
      sb = f32x2s{a.scale ? a.scale[c] : 1.f, a.shift ? a.shift[c] : 0.f};
      if (br.b) bi = br.b[tid];
    }
    // haug: haw is a power of two (32 at V >= 23), thread = (frame tid % haw, channels tid / haw + k * 256 / haw): the
    // frame, hence the sample and the address arithmetic, is fixed per thread
    const int hsh = 31 - __builtin_clz(HAW);
    const int he = tid & (HAW - 1), hc0 = tid >> hsh, hcs = TS_NT >> hsh;
    const int hG = Gs + he, hn = hG / Ts, ht = hG - hn * Ts;          // (source frames: Ts = T at stride 1)
    const bool hok = hG >= 0 && hn < a.n;
    const float* hsrc = a.zaug + ((size_t)(hok ? hn : 0) * C + br.c0) * Ts + (hok ? ht : 0);
    if constexpr (FOLD) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int ci = hc0 + q * hcs;
        hv[q] = (hok && ci < bc) ? hsrc[(size_t)ci * Ts] : 0.f;
      }
    }
    TSC_STAMP();
    {
      f32x4* w4 = reinterpret_cast<f32x4*>(Ws);     // 64 * S + 64 floats: a multiple of 4 (S odd, 64 * S = 0 mod 4)
      for (int i = tid; i < (64 * S + 64) >> 2; i += TS_NT) w4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float invrun = 1.f / (float)run;
    auto wat = [&](int i) -> int {                 // (co*bc + ci)*3 + tap -> co*S + tap*CP + ci  (i < 2^22: exact)
      int co, r;
      divmod_small(i, run, invrun, co, r);
      const int ci = (r * 43691) >> 17, tap = r - ci * KT;
      return co * S + tap * CP + ci;
    };
    if (tid < 64) {
      SB[tid] = sb;
      BI[tid] = bi;
    }
    __syncthreads();
    TSC_STAMP();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = tid + q * TS_NT;
      if (i < wtotal) Ws[wat(i)] = wv[q];
    }
    if constexpr (FOLD) {
      auto put = [&](int ci, float x) {
        const f32x2s e = SB[ci];
        float y = fmaf(x, e.x, e.y);
        if (relu) y = fmaxf(y, 0.f);
        HA[ci * HAW + he] = (hok && ci < bc) ? y : 0.f;
      };
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int ci = hc0 + q * hcs;
        if (ci < 64) put(ci, hv[q]);
      }
      for (int ci = hc0 + 8 * hcs; ci < 64; ci += hcs)               // haw > 32 (V < 23): the rest of the channels
        put(ci, (hok && ci < bc) ? hsrc[(size_t)ci * Ts] : 0.f);
    }
    for (int i0 = 8 * TS_NT; i0 < wtotal; i0 += 8 * TS_NT) {     // windows wider than 26 channels: the rest of the tile
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = i0 + tid + q * TS_NT;
        wv[q] = i < wtotal ? br.w[i] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = i0 + tid + q * TS_NT;
        if (i < wtotal) Ws[wat(i)] = wv[q];
      }
    }
    __syncthreads();
  }
  TSC_STAMP();
  f32x16 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;
  auto afrag = [&](int tap, int ks, int m) -> float {
    const int kl = 2 * ks + half;
    return FWD ? Ws[(32 * m + l31) * S + tap * CP + kl] : Ws[kl * S + tap * CP + 32 * m + l31];
  };
  // haug of channel 2ks+half at the lane's frame shifted by the tap, and the frame after it
  auto hfrag = [&](int tap, int ks) -> f32x2s {
    const float* hp = HA + (2 * ks + half) * HAW + gl + (tap - 1) * br.dil;
    return f32x2s{hp[0], hp[S2 ? 2 : 1]};          // (stride 2: the lane's second frame is two source frames on)
  };
  float avb[2][MT];
  f32x2s sbv[2], hab[2];
#pragma unroll
  for (int m = 0; m < MT; ++m) avb[0][m] = afrag(0, 0, m);
  sbv[0] = SB[half];
  if constexpr (FOLD) hab[0] = hfrag(0, 0);
  for (int base = 0; base < KS2; base += PDK) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int tap = u % KT, ks = base + u / KT;
      const int cur = u & 1, nxt = cur ^ 1;
      const int tn = (u + 1) % KT, kn = base + (u + 1) / KT;
#pragma unroll
      for (int m = 0; m < MT; ++m) avb[nxt][m] = afrag(tn, kn < 32 ? kn : 31, m);
      if (FWD) sbv[nxt] = SB[2 * (kn < 32 ? kn : 31) + half];
      if constexpr (FOLD) hab[nxt] = hfrag(tn, kn < 32 ? kn : 31);
      f32x4 b = buf[u];
      if constexpr (TWO) {
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = nx[q] ? bufB[u][q] : b[q];
      }
      if (tap != 1) {
        if (ODD) {
          const float x0 = b.x, y0 = b.y, x1 = b.z, y1 = b.w;
          b.x = fE[tap][0] ? y0 : x0;
          b.y = fS[tap][0] ? x0 : y0;
          b.z = fE[tap][1] ? y1 : x1;
          b.w = fS[tap][1] ? x1 : y1;
        }
      }
      if (FWD && !(a.exp & 4)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = fmaf(b[q], sbv[cur].x, sbv[cur].y);
          if (relu) v = fmaxf(v, 0.f);
          if constexpr (FOLD) v = fmaf(nx[q] ? hab[cur].y : hab[cur].x, cf[q], v);
          if (S2 || tap != 1) v *= mk[tap][q];
          b[q] = v;
        }
      } else if (S2) {
        // (a select, not a product: the masked elements of the one load may come from the uninitialised front pad)
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = mk[tap][q] != 0.f ? b[q] : 0.f;
      } else if (ODD && tap != 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] *= mk[tap][q];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(avb[cur][m], b[q], acc[m][q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      buf[u] = load(tap, min(ks + PDK, KS2 - 1));
      if constexpr (TWO) bufB[u] = loadB(tap, min(ks + PDK, KS2 - 1));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  TSC_STAMP();
  float* Tw = lds + wave * (32 * 36);               // one transpose tile per wave: the two sums of a row tile go through it in turn
  const int ooff = pok ? (((n0 + ds) * C + br.c0) * L + p) * 4 : TS_OOB;
  if constexpr (FWD) {
    double* Ss = reinterpret_cast<double*>(lds + 4 * 32 * 36);       // [4][MT*32][2]
    const __amdgpu_buffer_rsrc_t ro = ts_rsrc(AUG ? a.oaug : a.f, (size_t)a.n * C * L4);
    const bool stats = !AUG && a.stats != nullptr && !(a.exp & 2);
    float bmul[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bmul[k] = 1.f + cf[k];           // AUG: cf = 0
    __syncthreads();                                // the weight tile is dead: its LDS carries the transposes
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float sq[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = ts_row32(r, half);
        const int ch = 32 * m + row;
        const float bias = BI[ch];
        f32x4 val;
        float s = 0.f, qq = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          val[k] = fmaf(bias, bmul[k], acc[m][k][r]);
          s += val[k];
          qq = fmaf(val[k], val[k], qq);
        }
        ts_store4(val, ro, ts_rowoff(ch < bc, ooff, ch * L4));
        if (stats) {
          const bool ok = ch < bc && pok;
          Tw[row * 36 + l31] = ok ? s : 0.f;
          sq[r] = ok ? qq : 0.f;
        }
      }
      if (stats) {
        // (sums of a wave's 128 values of one row in fp32: 1e-7-class; everything across waves / blocks is fp64)
        wave_lds_sync();
        float sf = ts_rowread<float>(Tw, half, l31);
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 16; ++r) Tw[ts_row32(r, half) * 36 + l31] = sq[r];
        wave_lds_sync();
        float qf = ts_rowread<float>(Tw, half, l31);
        wave_lds_sync();
        sf += __shfl_xor(sf, 32, 64);
        qf += __shfl_xor(qf, 32, 64);
        const double sd = (double)sf, qd = (double)qf;
        if (half == 0) {
          Ss[((wave * MT + m) * 32 + l31) * 2 + 0] = sd;
          Ss[((wave * MT + m) * 32 + l31) * 2 + 1] = qd;
        }
      }
    }
    if (stats) {
      __syncthreads();
      if (tid < 32 * MT && tid < bc) {
        double s4 = 0.0, q4 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s4 += Ss[((w * MT * 32) + tid) * 2]; q4 += Ss[((w * MT * 32) + tid) * 2 + 1]; }
        a.stats[((size_t)grp * C + br.c0 + tid) * 2 + 0] = (float)s4;
        a.stats[((size_t)grp * C + br.c0 + tid) * 2 + 1] = (float)q4;
      }
    }
  } else {
    float* Ss = lds + 4 * 32 * 36;                                     // [4][MT*32][2]
    const __amdgpu_buffer_rsrc_t rx = ts_rsrc(AUG ? a.zaug : a.z, (size_t)a.n * C * L4);
    const __amdgpu_buffer_rsrc_t ro = ts_rsrc(AUG ? a.dzaug : a.dz, (size_t)a.n * C * L4);
    constexpr int G = MT * 4, PD = 4;
    f32x4 xa[PD][4];
    float su[16];
    auto fetch = [&](int g, int slot) {
      const int m = g >> 2, rb = (g & 3) * 4;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int ch = 32 * m + ts_row32(rb + rr, half);
        xa[slot][rr] = ts_load4(rx, ts_rowoff(ch < bc, ooff, ch * L4), 0);
      }
    };
#pragma unroll
    for (int g = 0; g < PD; ++g) fetch(g, g);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                // the weight tile is dead: its LDS carries the transposes
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int m = g >> 2, rb = (g & 3) * 4;
      const int slot = g % PD;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = rb + rr;
        const int row = ts_row32(r, half);
        const int ch = 32 * m + row;
        const f32x2s e = SB[ch];
        float u0 = 0.f, u1 = 0.f;
        f32x4 d;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float x = xa[slot][rr][q];
          const float pre = fmaf(x, e.x, e.y);
          const float dv = (!relu || pre > 0.f) ? acc[m][q][r] : 0.f;
          d[q] = dv * e.x;
          u0 = fmaf(dv, x, u0);
          u1 += dv;
        }
        ts_store4(d, ro, ts_rowoff(ch < bc, ooff, ch * L4));
        const bool ok = ch < bc && pok;
        Tw[row * 36 + l31] = ok ? u0 : 0.f;
        su[r] = ok ? u1 : 0.f;
      }
      if (MT == 2) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + PD < G) fetch(g + PD, slot);
        __builtin_amdgcn_sched_barrier(0);
      }
      if ((g & 3) == 3) {
        wave_lds_sync();
        float s0 = ts_rowread<float>(Tw, half, l31);
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 16; ++r) Tw[ts_row32(r, half) * 36 + l31] = su[r];
        wave_lds_sync();
        float s1 = ts_rowread<float>(Tw, half, l31);
        wave_lds_sync();
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (half == 0) {
          Ss[((wave * MT + m) * 32 + l31) * 2 + 0] = s0;
          Ss[((wave * MT + m) * 32 + l31) * 2 + 1] = s1;
        }
      }
    }
    __syncthreads();
    const size_t prow = (size_t)(AUG ? a.ngrp + a.n + grp : grp) * C;
    if (tid < 32 * MT && tid < bc) {
      float v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { v0 += Ss[((w * MT * 32) + tid) * 2]; v1 += Ss[((w * MT * 32) + tid) * 2 + 1]; }
      a.part[(prow + br.c0 + tid) * 2 + 0] = v0;
      a.part[(prow + br.c0 + tid) * 2 + 1] = v1;
    }
  }
#ifdef DSGCN_LAB
  TSC_STAMP();
  if (!AUG && (int)blockIdx.x == g_tsc_block && threadIdx.x == 0) g_tsc_stamp[15] = nst;
#endif
  // the table rows of this block hold nothing for the pooling / pass-through channels: the first window's block zeroes them
  float* tab = FWD ? (AUG ? nullptr : a.stats) : a.part;
  if (tab && first) {
    const size_t trow = (size_t)((!FWD && AUG) ? a.ngrp + a.n + grp : grp) * C;
    for (int i = 0; i < a.nbr; ++i)
      if (a.br[i].type != 0)
        for (int k = tid; k < a.br[i].bc * 2; k += TS_NT) tab[(trow + a.br[i].c0) * 2 + k] = 0.f;
  }
}

// max-pool / pass-through window, one wave per (n, c) plane.  Every global operand of the wave (the plane(s), the
// global-joint columns) is requested before the first LDS write.  lw: [L] z (forward: h = act(z)), [L] ge (backward),
// then [T] act(zaug), [T] pooled column / doaug, [32] coeff.  Planes up to 8 float4 per lane (T*V <= 2048).
constexpr int TS_PQ = 8;

// Stride 2 (S2): the pooled / copied output has T / 2 frames — out[t'] = max(h[2t'-1], h[2t'], h[2t'+1]) resp. h[2t'].  The
// planes are short (<= 2048 floats): both directions run one element per lane and step through LDS; lw: [Lin] z (forward:
// h), [Lout] ge (backward), [Tin] act(zaug), [Tout] pooled column / doaug, [32] coeff.
template <bool FWD>
__device__ __forceinline__ void ts_elem2(const TSArgs& a, const TSBranch& br, float* lw, int n, int c, int lane) {
  const int V = a.V, Ti = a.T, To = a.Tout, Li = Ti * V, Lo = To * V, C = a.C;
  const int cc = br.c0 + c;
  const size_t plane = (size_t)n * C + cc;
  const bool relu = cc < a.n_act;
  const bool pool = br.type == 1;
  const float invV = 1.f / (float)V;
  const int Tpi = (Ti + 3) & ~3, Tpo = (To + 3) & ~3;
  float* hp = lw;                // forward: act(z); backward: raw z
  float* gp = lw + Li;           // backward: ge plane
  float* ha = gp + Lo;           // act(zaug)
  float* oa = ha + Tpi;          // forward: pooled column; backward: doaug
  float* cf = oa + Tpo;
  const float s = a.scale ? a.scale[cc] : 1.f, b = a.shift ? a.shift[cc] : 0.f;
  auto act = [&](float x) -> float {
    const float y = fmaf(x, s, b);
    return relu ? fmaxf(y, 0.f) : y;
  };
  const f32x4* z4 = reinterpret_cast<const f32x4*>(a.z + plane * Li);
  const f32x4* g4 = FWD ? nullptr : reinterpret_cast<const f32x4*>(a.ge + plane * Lo);
  const float* za = a.zaug + plane * Ti;
  f32x4 zr[TS_PQ], gr[FWD ? 1 : TS_PQ / 2];
#pragma unroll
  for (int q = 0; q < TS_PQ; ++q) {
    zr[q] = z4[min(q * 64 + lane, (Li >> 2) - 1)];            // (clamped, not predicated: a predicated load becomes a branch and a wait)
  }
  if constexpr (!FWD) {
#pragma unroll
    for (int q = 0; q < TS_PQ / 2; ++q) {
      gr[q] = g4[min(q * 64 + lane, (Lo >> 2) - 1)];
    }
  }
  float zav[2], gav = 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int t = lane + 64 * q;
    zav[q] = za[min(t, Ti - 1)];
  }
  if (!FWD) gav = a.doaug[plane * To + min(lane, To - 1)];
  const float cfv = FWD ? a.coeff[min(lane, V - 1)] : 0.f;
#pragma unroll
  for (int q = 0; q < TS_PQ; ++q) {
    const int i = q * 64 + lane;
    if (i < (Li >> 2)) {
      f32x4 v = zr[q];
      if (FWD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = act(v[k]);
      }
      reinterpret_cast<f32x4*>(hp)[i] = v;
    }
  }
  if constexpr (!FWD) {
#pragma unroll
    for (int q = 0; q < TS_PQ / 2; ++q) {
      const int i = q * 64 + lane;
      if (i < (Lo >> 2)) reinterpret_cast<f32x4*>(gp)[i] = gr[q];
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int t = lane + 64 * q;
    if (t < Ti) ha[t] = act(zav[q]);
  }
  if (!FWD && lane < To) oa[lane] = gav;
  if (FWD && lane < V) cf[lane] = cfv;
  wave_lds_sync();
  if constexpr (FWD) {
    for (int tp = lane; tp < To; tp += 64) {
      const int t = 2 * tp;
      float v = ha[t];
      if (pool) {
        if (t >= 1) v = fmaxf(v, ha[t - 1]);
        if (t + 1 < Ti) v = fmaxf(v, ha[t + 1]);
      }
      oa[tp] = v;
      a.oaug[plane * To + tp] = v;
    }
    wave_lds_sync();
    float* fo = a.f + plane * Lo;
    float sf = 0.f, qf = 0.f;                      // (<= 32 values per lane: fp32; the wave's sum in fp64)
    for (int e = lane; e < Lo; e += 64) {
      int tp, v;
      divmod_small(e, V, invV, tp, v);
      const int t = 2 * tp, ei = t * V + v;
      float x = hp[ei];
      if (pool) {
        if (t >= 1) x = fmaxf(x, hp[ei - V]);
        if (t + 1 < Ti) x = fmaxf(x, hp[ei + V]);
      }
      const float r = fmaf(oa[tp], cf[v], x);
      fo[e] = r;
      sf += r;
      qf = fmaf(r, r, qf);
    }
    if (a.stats) {
      const double sv = wave_sum_d((double)sf);
      const double qv = wave_sum_d((double)qf);
      if (lane == 0) {
        a.stats[((size_t)(a.ngrp + n) * C + cc) * 2 + 0] = (float)sv;
        a.stats[((size_t)(a.ngrp + n) * C + cc) * 2 + 1] = (float)qv;
      }
    }
  } else {
    // gradient of source row t of a column: pass-through: g[t / 2] on even rows; max-pool: the windows (centres = the even
    // rows) whose FIRST maximal valid row is t.  hcol: the pooled values are actf(hcol[.])
    auto route = [&](auto&& hval, const float* gcol, int t, int st) -> float {
      if (!pool) return (!(t & 1) && (t >> 1) < To) ? gcol[(t >> 1) * st] : 0.f;
      auto row = [&](int r) -> float { return (r >= 0 && r < Ti) ? hval(r) : -INFINITY; };
      const float v = hval(t);
      const float m2 = row(t - 2), m1 = row(t - 1), p1 = row(t + 1), p2 = row(t + 2);
      float g = 0.f;
      if (t >= 1 && !((t - 1) & 1) && ((t - 1) >> 1) < To && v > m2 && v > m1) g += gcol[((t - 1) >> 1) * st];
      if (!(t & 1) && (t >> 1) < To && v > m1 && v >= p1) g += gcol[(t >> 1) * st];
      if (((t + 1) & 1) == 0 && ((t + 1) >> 1) < To && v >= p1 && v >= p2) g += gcol[((t + 1) >> 1) * st];
      return g;
    };
    float* dzo = a.dz + plane * Li;
    float u0 = 0.f, u1 = 0.f;
    if (pool) {
      for (int e = lane; e < Li; e += 64) {
        int t, v;
        divmod_small(e, V, invV, t, v);
        const float x = hp[e];
        float g = route([&](int r) { return act(hp[r * V + v]); }, gp + v, t, V);
        if (relu && !(fmaf(x, s, b) > 0.f)) g = 0.f;
        dzo[e] = g * s;
        u0 = fmaf(g, x, u0);
        u1 += g;
      }
    } else {
      for (int e = lane; e < Li; e += 64) {
        int t, v;
        divmod_small(e, V, invV, t, v);
        const float x = hp[e];
        float g = (t & 1) ? 0.f : gp[(t >> 1) * V + v];
        if (relu && !(fmaf(x, s, b) > 0.f)) g = 0.f;
        dzo[e] = g * s;
        u0 = fmaf(g, x, u0);
        u1 += g;
      }
    }
    for (int t = lane; t < Ti; t += 64) {
      const float x = za[t];
      float g = route([&](int r) { return ha[r]; }, oa, t, 1);
      if (relu && !(fmaf(x, s, b) > 0.f)) g = 0.f;
      a.dzaug[plane * Ti + t] = g * s;
      u0 = fmaf(g, x, u0);
      u1 += g;
    }
    u0 = wave_sum(u0);
    u1 = wave_sum(u1);
    if (lane == 0) {
      a.part[((size_t)(a.ngrp + n) * C + cc) * 2 + 0] = u0;
      a.part[((size_t)(a.ngrp + n) * C + cc) * 2 + 1] = u1;
    }
  }
}

template <bool FWD>
__device__ __forceinline__ void ts_elem(const TSArgs& a, const TSBranch& br, float* lw, int n, int c, int lane) {
  const int V = a.V, T = a.T, L = T * V, L4n = L >> 2, C = a.C;
  const int cc = br.c0 + c;
  const size_t plane = (size_t)n * C + cc;
  const bool relu = cc < a.n_act;
  const bool pool = br.type == 1;
  const float invV = 1.f / (float)V;
  const int Tp = (T + 3) & ~3;
  float* hp = lw;
  float* gp = lw + L;
  float* ha = lw + (FWD ? 1 : 2) * L;
  float* oa = ha + Tp;
  float* cf = oa + Tp;
  const float* za = a.zaug + plane * T;
  const f32x4* z4 = reinterpret_cast<const f32x4*>(a.z + plane * L);
  const f32x4* g4 = FWD ? nullptr : reinterpret_cast<const f32x4*>(a.ge + plane * L);
  auto act = [&](float x, float s, float b) -> float {
    const float y = fmaf(x, s, b);
    return relu ? fmaxf(y, 0.f) : y;
  };
  // ---- requests ------------------------------------------------------------------------------------------------
  const float s = a.scale ? a.scale[cc] : 1.f, b = a.shift ? a.shift[cc] : 0.f;
  f32x4 zr[TS_PQ], gr[FWD ? 1 : TS_PQ];
#pragma unroll
  for (int q = 0; q < TS_PQ; ++q) {
    // (clamped, not predicated: a predicated load becomes a branch, and the compiler then waits for each load in turn —
    // seven serial round trips per plane in the first version's disassembly)
    const int i = min(q * 64 + lane, L4n - 1);
    zr[q] = z4[i];
    if (!FWD) gr[q] = g4[i];
  }
  float zav[2], gav[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int t = lane + 64 * q;
    zav[q] = za[min(t, T - 1)];
    gav[q] = FWD ? 0.f : a.doaug[plane * T + min(t, T - 1)];
  }
  const float cfv = FWD ? a.coeff[min(lane, V - 1)] : 0.f;
  // ---- LDS -----------------------------------------------------------------------------------------------------
  if (pool) {
    f32x4* h4 = reinterpret_cast<f32x4*>(hp);
    f32x4* gl4 = reinterpret_cast<f32x4*>(gp);
#pragma unroll
    for (int q = 0; q < TS_PQ; ++q) {
      const int i = q * 64 + lane;
      if (i < L4n) {
        f32x4 v = zr[q];
        if (FWD) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = act(v[k], s, b);
        }
        h4[i] = v;
        if (!FWD) gl4[i] = gr[q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int t = lane + 64 * q;
    if (t < T) {
      ha[t] = act(zav[q], s, b);
      if (!FWD) oa[t] = gav[q];
    }
  }
  for (int t = lane + 128; t < T; t += 64) {          // T > 128: the rest of the columns
    ha[t] = act(za[t], s, b);
    if (!FWD) oa[t] = a.doaug[plane * T + t];
  }
  if (FWD && lane < V) cf[lane] = cfv;
  wave_lds_sync();
  if constexpr (FWD) {
    for (int t = lane; t < T; t += 64) {
      float v = ha[t];
      if (pool) {
        if (t >= 1) v = fmaxf(v, ha[t - 1]);
        if (t + 1 < T) v = fmaxf(v, ha[t + 1]);
      }
      a.oaug[plane * T + t] = v;
    }
    f32x4* fp = reinterpret_cast<f32x4*>(a.f + plane * L);
    double sv = 0.0, qv = 0.0;
    if (pool) {
      // lane = consecutive elements: the three rows of a window are conflict-free LDS reads (a lane owning four consecutive
      // elements reads at a stride of four banks: four lanes per bank)
      float* fo = a.f + plane * L;
      for (int e = lane; e < L; e += 64) {
        int t, v;
        divmod_small(e, V, invV, t, v);
        float x = hp[e], o = ha[t];
        if (t >= 1) { x = fmaxf(x, hp[e - V]); o = fmaxf(o, ha[t - 1]); }
        if (t + 1 < T) { x = fmaxf(x, hp[e + V]); o = fmaxf(o, ha[t + 1]); }
        const float r = fmaf(o, cf[v], x);
        fo[e] = r;
        sv += (double)r;
        qv = fma((double)r, (double)r, qv);
      }
    } else
#pragma unroll
    for (int q = 0; q < TS_PQ; ++q) {
      const int i = q * 64 + lane;
      if (i < L4n) {
        int t, v;
        divmod_small(4 * i, V, invV, t, v);
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float x, o;
          if (pool) {
            const int e = t * V + v;
            x = hp[e];
            o = ha[t];
            if (t >= 1) { x = fmaxf(x, hp[e - V]); o = fmaxf(o, ha[t - 1]); }
            if (t + 1 < T) { x = fmaxf(x, hp[e + V]); o = fmaxf(o, ha[t + 1]); }
          } else {
            x = act(zr[q][k], s, b);
            o = ha[t];
          }
          r[k] = fmaf(o, cf[v], x);
          if (++v == V) { v = 0; ++t; }
        }
        fp[i] = f32x4{r[0], r[1], r[2], r[3]};
        const float s4 = (r[0] + r[1]) + (r[2] + r[3]);
        const float q4 = fmaf(r[0], r[0], r[1] * r[1]) + fmaf(r[2], r[2], r[3] * r[3]);
        sv += (double)s4;
        qv += (double)q4;
      }
    }
    if (a.stats) {
      sv = wave_sum_d(sv);
      qv = wave_sum_d(qv);
      if (lane == 0) {
        a.stats[((size_t)(a.ngrp + n) * C + cc) * 2 + 0] = (float)sv;
        a.stats[((size_t)(a.ngrp + n) * C + cc) * 2 + 1] = (float)qv;
      }
    }
  } else {
    // gradient of row t of a column: pass-through = g[t]; max-pool = the windows whose FIRST maximal valid row is t (ATen
    // max_pool2d_with_indices order; rows outside the plane count as -inf).  hcol holds raw z: the pooled values are act(.)
    auto route = [&](const float* hcol, const float* gcol, int t, int st) -> float {
      auto row = [&](int r) -> float { return (r >= 0 && r < T) ? act(hcol[r * st], s, b) : -INFINITY; };
      const float v = act(hcol[t * st], s, b);
      const float m2 = row(t - 2), m1 = row(t - 1), p1 = row(t + 1), p2 = row(t + 2);
      float g = 0.f;
      if (t >= 1 && v > m2 && v > m1) g += gcol[(t - 1) * st];
      if (v > m1 && v >= p1) g += gcol[t * st];
      if (t + 1 < T && v >= p1 && v >= p2) g += gcol[(t + 1) * st];
      return g;
    };
    f32x4* dp = reinterpret_cast<f32x4*>(a.dz + plane * L);
    float u0 = 0.f, u1 = 0.f;
    if (pool) {
      float* dzo = a.dz + plane * L;
      for (int e = lane; e < L; e += 64) {
        int t, v;
        divmod_small(e, V, invV, t, v);
        const float x = hp[e];
        float g = route(hp + v, gp + v, t, V);
        if (relu && !(fmaf(x, s, b) > 0.f)) g = 0.f;
        dzo[e] = g * s;
        u0 = fmaf(g, x, u0);
        u1 += g;
      }
    } else
#pragma unroll
    for (int q = 0; q < TS_PQ; ++q) {
      const int i = q * 64 + lane;
      if (i < L4n) {
        int t, v;
        divmod_small(4 * i, V, invV, t, v);
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float x = zr[q][k];
          float g = pool ? route(hp + v, gp + v, t, V) : gr[q][k];
          if (relu && !(fmaf(x, s, b) > 0.f)) g = 0.f;
          r[k] = g * s;
          u0 = fmaf(g, x, u0);
          u1 += g;
          if (++v == V) { v = 0; ++t; }
        }
        dp[i] = f32x4{r[0], r[1], r[2], r[3]};
      }
    }
    // the column: ha holds act(zaug); the pooled values are ha itself, the pre-activation sign is ha > 0 under ReLU
    auto route1 = [&](int t) -> float {
      if (!pool) return oa[t];
      auto row = [&](int r) -> float { return (r >= 0 && r < T) ? ha[r] : -INFINITY; };
      const float v = ha[t];
      const float m2 = row(t - 2), m1 = row(t - 1), p1 = row(t + 1), p2 = row(t + 2);
      float g = 0.f;
      if (t >= 1 && v > m2 && v > m1) g += oa[t - 1];
      if (v > m1 && v >= p1) g += oa[t];
      if (t + 1 < T && v >= p1 && v >= p2) g += oa[t + 1];
      return g;
    };
    for (int t = lane; t < T; t += 64) {
      const float x = t < 128 ? zav[t >> 6] : za[t];
      float g = route1(t);
      if (relu && !(fmaf(x, s, b) > 0.f)) g = 0.f;
      a.dzaug[plane * T + t] = g * s;
      u0 = fmaf(g, x, u0);
      u1 += g;
    }
    u0 = wave_sum(u0);
    u1 = wave_sum(u1);
    if (lane == 0) {
      a.part[((size_t)(a.ngrp + n) * C + cc) * 2 + 0] = u0;
      a.part[((size_t)(a.ngrp + n) * C + cc) * 2 + 1] = u1;
    }
  }
}

// grid.x = [conv blocks of the V columns: (position group, conv window)] ++ [conv blocks of the global-joint column] ++
// [plane blocks: 4 planes each]
template <bool FWD, int MT, bool S2 = false>
__global__ __launch_bounds__(TS_NT, (MT == 1 && !S2) ? 3 : 2) void k_tsp(TSArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cb = a.nconv * a.ngrp, cba = a.nconv * a.ngrpa, b = blockIdx.x;
  if (b < cb) {
    if (a.exp & 8) return;
    const int w = b % a.nconv, grp = b / a.nconv;
    const TSBranch& br = ts_conv_window(a, w);
    if constexpr (S2) ts_conv<FWD, MT, false, false, true>(a, br, w == 0, lds, grp, lane, wave);
    else if (((br.dil * a.V) & 1) && !(a.exp & 16)) ts_conv<FWD, MT, true, false>(a, br, w == 0, lds, grp, lane, wave);
    else ts_conv<FWD, MT, false, false>(a, br, w == 0, lds, grp, lane, wave);
    return;
  }
  if (b < cb + cba) {
    const int w = (b - cb) % a.nconv, grp = (b - cb) / a.nconv;
    const TSBranch& br = ts_conv_window(a, w);
    if constexpr (S2) ts_conv<FWD, MT, false, true, true>(a, br, w == 0, lds, grp, lane, wave);
    else if (br.dil & 1) ts_conv<FWD, MT, true, true>(a, br, w == 0, lds, grp, lane, wave);
    else ts_conv<FWD, MT, false, true>(a, br, w == 0, lds, grp, lane, wave);
    return;
  }
  int pl = (b - cb - cba) * 4 + wave;
  if (pl >= a.n * a.eplanes || (a.exp & 1)) return;
  const int n = pl / a.eplanes;
  int c = pl - n * a.eplanes;
  float* tab = FWD ? a.stats : a.part;
  if (c == 0 && tab) {                  // the plane rows hold nothing for the conv channels: the first plane's wave zeroes them
    for (int i = 0; i < a.nbr; ++i)
      if (a.br[i].type == 0)
        for (int k = lane; k < a.br[i].bc * 2; k += 64) tab[((size_t)(a.ngrp + n) * a.C + a.br[i].c0) * 2 + k] = 0.f;
  }
  const int per = S2 ? a.T * a.V + a.Tout * a.V + ((a.T + 3) & ~3) + ((a.Tout + 3) & ~3) + 32
                     : (FWD ? 1 : 2) * a.T * a.V + 2 * ((a.T + 3) & ~3) + 32;
  for (int i = 0; i < a.nbr; ++i) {
    if (a.br[i].type == 0) continue;
    if (c < a.br[i].bc) {
      if constexpr (S2) ts_elem2<FWD>(a, a.br[i], lds + (size_t)wave * per, n, c, lane);
      else ts_elem<FWD>(a, a.br[i], lds + (size_t)wave * per, n, c, lane);
      return;
    }
    c -= a.br[i].bc;
  }
}

// ge = gf + A0[c] + B0[c]*f ;  doaug[t] = sum_v ge[t,v]*coeff[v] ;  pcoef (n*C, V): sum_t ge[t,v]*oaug[t]
// one wave per (n, c) plane; LDS: [L] ge, [T] oaug, [32] coeff
__global__ __launch_bounds__(64) void k_tsp_prep(const float* __restrict__ gf, const float* __restrict__ f,
                                                 const float* __restrict__ oaug, const float* __restrict__ coeff,
                                                 const float* __restrict__ A0, const float* __restrict__ B0,
                                                 float* __restrict__ ge, float* __restrict__ doaug,
                                                 float* __restrict__ pcoef, int C, int T, int V) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int c = (int)(plane % C);
  const int L = T * V, L4n = L >> 2, Tp = (T + 3) & ~3;
  float* gl = lds;
  float* ol = lds + L;
  float* cf = ol + Tp;
  if (lane < V) cf[lane] = coeff[lane];
  for (int t = lane; t < T; t += 64) ol[t] = oaug[plane * T + t];
  const float a0 = A0 ? A0[c] : 0.f, b0 = B0 ? B0[c] : 0.f;
  const f32x4* g4 = gf ? reinterpret_cast<const f32x4*>(gf + plane * L) : nullptr;
  const f32x4* f4 = reinterpret_cast<const f32x4*>(f + plane * L);
  f32x4* o4 = reinterpret_cast<f32x4*>(ge + plane * L);
  f32x4* l4 = reinterpret_cast<f32x4*>(gl);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int base = 0; base < L4n; base += 256) {
    f32x4 gv[4], fv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = base + q * 64 + lane;
      const bool ok = i < L4n;
      gv[q] = (ok && g4) ? g4[i] : zero4;
      fv[q] = (ok && A0) ? f4[i] : zero4;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = base + q * 64 + lane;
      if (i < L4n) {
        f32x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = gv[q][k] + fmaf(b0, fv[q][k], a0);
        o4[i] = r;
        l4[i] = r;
      }
    }
  }
  wave_lds_sync();
  for (int t = lane; t < T; t += 64) {
    float acc = 0.f;
    for (int v = 0; v < V; ++v) acc = fmaf(gl[t * V + v], cf[v], acc);
    doaug[plane * T + t] = acc;
  }
  {
    const int v = lane & 31, hf = lane >> 5;
    float acc = 0.f;
    if (v < V)
      for (int t = hf; t < T; t += 2) acc = fmaf(gl[t * V + v], ol[t], acc);
    acc += __shfl_xor(acc, 32, 64);
    if (lane < V) pcoef[plane * V + lane] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient: k_tapw's scheme (csrc/tapconv.hip) on the split operands.  Work unit = (sample, 4 frames); the
// workgroup stages ge (4 frames) and h = relu(z*scale+shift) (the 4 frames plus a 4-frame halo on either side) with
// 16-byte loads into LDS rows [V-column part | one-column part] of stride = 2 mod 4 floats; the matrix loop walks the
// V-column part in groups of four positions (a tap shifts by dil*V: an odd shift reads its operand with two 4-byte LDS
// reads instead of one 8-byte read) and then the one group of the unit's four global-joint values (shift dil), so dW / db
// carry both parts.  grid = (K-splits, conv windows); every split writes its partial row (dsgcn_colsum).
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int tsw_ls(int w) { return ((w + 1) & ~3) + 2; }      // >= w, = 2 mod 4

#ifdef DSGCN_LAB
// wall-clock stamps (10 ns) of workgroup (0, 0), thread 0 of k_tspw: start, LDS cleared, then per unit (committed + barrier,
// next unit requested, products + barrier), partial rows written; [63] = count (dsgcn_tms_split_phases)
__device__ long long g_tsw_stamp[64];
#define TSW_STAMP() do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && nst < 62) g_tsw_stamp[nst++] = wall_clock64(); } while (0)
#else
#define TSW_STAMP() do {} while (0)
#endif

template <int CH>
__global__ __launch_bounds__(TS_NT, CH == 32 ? 2 : 1) void k_tspw(TSArgs a, BnCoefTable jobs) {
  constexpr int KT = 3;
  constexpr int JD = CH == 32 ? 4 : 7, JX = CH == 32 ? 10 : 20;     // float4 staging slots per thread (V <= 25)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= a.splits) {               // hosted BatchNorm coefficient jobs (bn_jobs.h): workgroups past the K-splits
    if (blockIdx.y == 0)
      bnj_dispatch(jobs, (int)blockIdx.x - a.splits,
                   [&](const BnCoefJob& J, int b) { bn_coef_rows_block<TS_NT>(J, b, reinterpret_cast<double (*)[8][2]>(lds)); });
    return;
  }
#ifdef DSGCN_LAB
  int nst = 0;
#endif
  TSW_STAMP();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  // the block's windows: group blockIdx.y = `gcnt` consecutive conv windows (channel-contiguous, together <= CH channels:
  // DS-STGCN's 64-channel layers have windows of 14 + 10 and 10 + 10).  One (co x ci) tile carries them all — its
  // off-diagonal blocks are never read — with the tap shift a per-LANE quantity (the lane's ci row knows its window);
  // per-block cost (staging, barriers, the LDS round trips) is then paid once for the group.
  const int gw0 = a.gfirst[blockIdx.y], gcnt = a.gcount[blockIdx.y];
  const TSBranch& br = ts_conv_window(a, gw0);
  const int V = a.V, T = a.T, L = T * V, C = a.C;
  int bc = 0, ldil = br.dil;                       // channels of the group; dilation of the window holding row l31
  for (int w = 0; w < gcnt; ++w) {
    const TSBranch& bw = ts_conv_window(a, gw0 + w);
    if (l31 >= bc) ldil = bw.dil;
    bc += bw.bc;
  }
  const int GM = TS_R * V, XM = (TS_R + 2 * TS_H) * V;              // V-column part of a row (floats; multiples of 4)
  const int GS4 = GM >> 2, XS4 = XM >> 2, SEG4 = (TS_H * V) >> 2;   // float4 per row; float4 per 4-frame segment
  const int LSd = tsw_ls(GM + TS_R), LSx = tsw_ls(XM + TS_R + 2 * TS_H);
  float* Ds = lds;                                                   // [CH][LSd]
  float* Xs = lds + CH * LSd;                                        // [CH][LSx]
  f32x2s* SB = reinterpret_cast<f32x2s*>(Xs + CH * LSx);             // [CH] (scale, shift)
  const f32x2s sb0 = (tid < CH && tid < bc) ? f32x2s{a.scale ? a.scale[br.c0 + tid] : 1.f, a.shift ? a.shift[br.c0 + tid] : 0.f}
                                            : f32x2s{0.f, 0.f};
  const int rel0 = br.c0;                          // ReLU on channels < n_act (per row: a group may straddle)

  const int units = a.n * (T / TS_R);
  const int per = (units + a.splits - 1) / a.splits;
  const int u0 = blockIdx.x * per, u1 = min(units, u0 + per);

  const __amdgpu_buffer_rsrc_t rg = ts_rsrc(a.ge, (size_t)a.n * C * L * 4);
  const __amdgpu_buffer_rsrc_t rz = ts_rsrc(a.z, (size_t)a.n * C * L * 4);
  const __amdgpu_buffer_rsrc_t rga = ts_rsrc(a.doaug, (size_t)a.n * C * T * 4);
  const __amdgpu_buffer_rsrc_t rza = ts_rsrc(a.zaug, (size_t)a.n * C * T * 4);
  // staging slots: f = tid + 256*j -> (row, float4 column); fixed per thread
  int vD[JD], lD[JD], vX[JX], lX[JX];          // lX: LDS index | segment << 20 | row << 22 | valid << 28
#pragma unroll
  for (int j = 0; j < JD; ++j) {
    const int f = tid + TS_NT * j, row = f / GS4, c4 = f - row * GS4;
    const bool ok = row < bc;
    vD[j] = ok ? (row * L + 4 * c4) * 4 : TS_OOB;
    lD[j] = ok ? row * LSd + 4 * c4 : -1;
  }
#pragma unroll
  for (int j = 0; j < JX; ++j) {
    const int f = tid + TS_NT * j, row = f / XS4, c4 = f - row * XS4;
    const bool ok = row < bc;
    vX[j] = ok ? (row * L + 4 * c4) * 4 : TS_OOB;
    const int seg = c4 / SEG4;                  // 0 = halo before (needs bit 1 of `inval` clear), 1 = the unit, 2 = halo after (bit 2)
    lX[j] = (row * LSx + 4 * c4) | ((seg == 0 ? 1 : seg == 2 ? 2 : 0) << 20) | ((row & 63) << 22) | (ok ? 1 << 28 : 0);
  }
  // the one-column operands: thread < CH: doaug of row tid (4 frames); thread < 3*CH: zaug of row tid/3, segment tid%3
  const int arow = tid / 3, aseg = tid - arow * 3;
  const bool aokD = tid < CH && tid < bc && !(a.exp & 256), aokX = tid < 3 * CH && arow < bc && !(a.exp & 256);
  const int aneed = aseg == 0 ? 1 : aseg == 2 ? 2 : 0;
  f32x4 gr[JD], xr[JX], ga, xa;
  float dsum[JD], dsa = 0.f;
#pragma unroll
  for (int j = 0; j < JD; ++j) dsum[j] = 0.f;
  int inval = 0;                              // bit 1: the unit has no halo before it (t0 < 4), bit 2: none after it
  auto issue = [&](int u) {
    const int n = u / (T / TS_R), t0 = (u - n * (T / TS_R)) * TS_R;
    const int sg = ((n * C + br.c0) * L + t0 * V) * 4;
    const int sx = ((n * C + br.c0) * L + (t0 - TS_H) * V) * 4;          // may be negative: only used with valid slots
    inval = (t0 >= TS_H ? 0 : 1) | (t0 + TS_R < T ? 0 : 2);
#pragma unroll
    for (int j = 0; j < JD; ++j) gr[j] = ts_load4(rg, vD[j], sg);
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const bool ok = (((lX[j] >> 20) & 3) & inval) == 0;
      xr[j] = ts_load4(rz, ok ? vX[j] + sx : TS_OOB, 0);
    }
    ga = ts_load4(rga, aokD ? ((n * C + br.c0 + tid) * T + t0) * 4 : TS_OOB, 0);
    const bool oka = aokX & ((aneed & inval) == 0);
    xa = ts_load4(rza, oka ? ((n * C + br.c0 + arow) * T + t0 + (aseg - 1) * TS_H) * 4 : TS_OOB, 0);
  };
  const bool noact = a.exp & 128;
  auto act = [&](float x, f32x2s sb, bool rl) -> float {
    if (noact) return x;
    const float y = fmaf(x, sb.x, sb.y);
    return rl ? fmaxf(y, 0.f) : y;
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < JD; ++j) {
      if (lD[j] >= 0) {
        f32x2s* d = reinterpret_cast<f32x2s*>(Ds + lD[j]);
        d[0] = f32x2s{gr[j].x, gr[j].y};
        d[1] = f32x2s{gr[j].z, gr[j].w};
        dsum[j] += (gr[j].x + gr[j].y) + (gr[j].z + gr[j].w);
      }
    }
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      if (lX[j] & (1 << 28)) {
        const bool ok = (((lX[j] >> 20) & 3) & inval) == 0;
        const int row = (lX[j] >> 22) & 63;
        const f32x2s sb = SB[row];
        const bool rl = rel0 + row < a.n_act;
        f32x2s* d = reinterpret_cast<f32x2s*>(Xs + (lX[j] & 0xfffff));
        d[0] = ok ? f32x2s{act(xr[j].x, sb, rl), act(xr[j].y, sb, rl)} : f32x2s{0.f, 0.f};
        d[1] = ok ? f32x2s{act(xr[j].z, sb, rl), act(xr[j].w, sb, rl)} : f32x2s{0.f, 0.f};
      }
    }
    if (aokD) {
      f32x2s* d = reinterpret_cast<f32x2s*>(Ds + tid * LSd + GM);
      d[0] = f32x2s{ga.x, ga.y};
      d[1] = f32x2s{ga.z, ga.w};
      dsa += (ga.x + ga.y) + (ga.z + ga.w);
    }
    if (aokX) {
      const bool ok = (aneed & inval) == 0;
      const f32x2s sb = SB[arow];
      const bool rl = rel0 + arow < a.n_act;
      f32x2s* d = reinterpret_cast<f32x2s*>(Xs + arow * LSx + XM + aseg * TS_H);
      d[0] = ok ? f32x2s{act(xa.x, sb, rl), act(xa.y, sb, rl)} : f32x2s{0.f, 0.f};
      d[1] = ok ? f32x2s{act(xa.z, sb, rl), act(xa.w, sb, rl)} : f32x2s{0.f, 0.f};
    }
  };

  f32x16 acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  // CH = 32: wave w takes the position groups [g0, g1) of the unit's GS4 groups of 4 (wave 0 also the global-joint group);
  // CH = 64: wave = (mt, nt), all groups
  const int mt = CH == 64 ? (wave >> 1) : 0, nt = CH == 64 ? (wave & 1) : 0;
  const int g0 = CH == 32 ? (GS4 * wave) / 4 : 0, g1 = CH == 32 ? (GS4 * (wave + 1)) / 4 : GS4;
  const bool augw = CH == 64 || wave == 0;
  const int sh = ldil * V;
  const bool oddsh = gcnt > 1 || ((sh & 1) && !(a.exp & 512));   // per-lane shifts: the 4-byte form takes any parity
  const float* Ap = Ds + (32 * mt + l31) * LSd + 2 * half;
  const float* Bp = Xs + (32 * nt + l31) * LSx + TS_H * V + 2 * half;
  const float* Aq = Ds + (32 * mt + l31) * LSd + GM + 2 * half;
  const float* Bq = Xs + (32 * nt + l31) * LSx + XM + TS_H + 2 * half;
  auto mm = [&](f32x2s av, f32x2s b0, f32x2s b1, f32x2s b2) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b2.x, acc[2], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b2.y, acc[2], 0, 0, 0);
  };

  if (u0 < u1) issue(u0);                           // the first unit travels while the tile is cleared
  {
    f32x4* l4 = reinterpret_cast<f32x4*>(lds);      // rows >= bc and the pad columns stay zero (CH * (LSd + LSx) is a multiple of 4)
    for (int i = tid; i < (CH * (LSd + LSx)) >> 2; i += TS_NT) l4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < CH) SB[tid] = sb0;
  }
  __syncthreads();
  TSW_STAMP();
  for (int u = u0; u < u1; ++u) {
    commit();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    TSW_STAMP();
    if (u + 1 < u1) issue(u + 1);
    TSW_STAMP();
    if (oddsh) {
#pragma unroll 2
      for (int g = g0; g < g1; ++g) {
        const f32x2s av = *reinterpret_cast<const f32x2s*>(Ap + 4 * g);
        const f32x2s b1 = *reinterpret_cast<const f32x2s*>(Bp + 4 * g);
        const f32x2s b0 = {Bp[4 * g - sh], Bp[4 * g - sh + 1]};
        const f32x2s b2 = {Bp[4 * g + sh], Bp[4 * g + sh + 1]};
        mm(av, b0, b1, b2);
      }
    } else {
#pragma unroll 2
      for (int g = g0; g < g1; ++g) {
        const f32x2s av = *reinterpret_cast<const f32x2s*>(Ap + 4 * g);
        const f32x2s b0 = *reinterpret_cast<const f32x2s*>(Bp + 4 * g - sh);
        const f32x2s b1 = *reinterpret_cast<const f32x2s*>(Bp + 4 * g);
        const f32x2s b2 = *reinterpret_cast<const f32x2s*>(Bp + 4 * g + sh);
        mm(av, b0, b1, b2);
      }
    }
    if (augw) {
      const f32x2s av = *reinterpret_cast<const f32x2s*>(Aq);
      const f32x2s b1 = *reinterpret_cast<const f32x2s*>(Bq);
      const f32x2s b0 = {Bq[-ldil], Bq[1 - ldil]};
      const f32x2s b2 = {Bq[ldil], Bq[1 + ldil]};
      mm(av, b0, b1, b2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // raw barrier: the next unit's loads stay in flight
    TSW_STAMP();
  }

  if (CH == 32) {
    float* Rs = lds;                                 // [wave][tap][co][33]
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) Rs[((wave * KT + k) * 32 + ts_row32(r, half)) * 33 + l31] = acc[k][r];
    __syncthreads();
    for (int w = 0, r0 = 0; w < gcnt; ++w) {         // the diagonal block of every window of the group
      const TSBranch& bw = ts_conv_window(a, gw0 + w);
      const int wb = bw.bc;
      float* dw = bw.dwp + (size_t)blockIdx.x * a.pstride;
      for (int o = tid; o < KT * wb * wb; o += TS_NT) {
        const int co = o / (wb * KT), r2 = o - co * wb * KT, ci = r2 / KT, k = r2 - ci * KT;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) v += Rs[((wv * KT + k) * 32 + r0 + co) * 33 + r0 + ci];
        dw[o] = v;                                   // (co*bc + ci)*KT + tap
      }
      r0 += wb;
    }
    __syncthreads();
  } else {
    float* dw = br.dwp + (size_t)blockIdx.x * a.pstride;      // (CH = 64: one window per group)
    const int ci = 32 * nt + l31;
    if (ci < bc) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * mt + ts_row32(r, half);
        if (co < bc) {
#pragma unroll
          for (int k = 0; k < KT; ++k) dw[((size_t)co * bc + ci) * KT + k] = acc[k][r];
        }
      }
    }
    __syncthreads();
  }
  // db: per-thread row pieces -> LDS -> one thread per row
  float* Bs = lds;                                   // [CH][GS4 + 2]
#pragma unroll
  for (int j = 0; j < JD; ++j) {
    const int f = tid + TS_NT * j, row = f / GS4, c4 = f - row * GS4;
    if (row < bc) Bs[row * (GS4 + 2) + c4] = dsum[j];
  }
  if (aokD) Bs[tid * (GS4 + 2) + GS4] = dsa;
  __syncthreads();
  if (tid < bc) {
    float v = 0.f;
    for (int c = 0; c <= GS4; ++c) v += Bs[tid * (GS4 + 2) + c];
    int r0 = 0;
    for (int w = 0; w < gcnt; ++w) {
      const TSBranch& bw = ts_conv_window(a, gw0 + w);
      if (tid >= r0 && tid < r0 + bw.bc) bw.dbp[(size_t)blockIdx.x * a.pstride + tid - r0] = v;
      r0 += bw.bc;
    }
  }
#ifdef DSGCN_LAB
  TSW_STAMP();
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_tsw_stamp[63] = nst;
#endif
}

// Stride-2 form of the weight gradient:  dW[co,ci,tap] = sum ge[n,co,t',v] * h[n,ci, 2t' + (tap-1)*dil, v].  The source rows a
// unit of four output frames needs are every second frame, so the haloed tile does not apply: every tap gets its own
// decimated tile Xs[tap][ci][4 frames | 4 global-joint values] (a frame row is V floats at a 4-byte-aligned address: seven
// 16-byte loads per 25-float row, the dwords past the row dropped when the tile is written) and the matrix loop reads ge
// and h at the SAME tile position.  Same grid, groups, splits and epilogue as k_tspw.
template <int CH>
__global__ __launch_bounds__(TS_NT, CH == 32 ? 2 : 1) void k_tspw2(TSArgs a, BnCoefTable jobs) {
  constexpr int KT = 3;
  constexpr int JD = CH == 32 ? 4 : 7, JX = CH == 32 ? 11 : 21, JA = CH == 32 ? 2 : 3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= a.splits) {               // hosted BatchNorm coefficient jobs (bn_jobs.h): workgroups past the K-splits
    if (blockIdx.y == 0)
      bnj_dispatch(jobs, (int)blockIdx.x - a.splits,
                   [&](const BnCoefJob& J, int b) { bn_coef_rows_block<TS_NT>(J, b, reinterpret_cast<double (*)[8][2]>(lds)); });
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int gw0 = a.gfirst[blockIdx.y], gcnt = a.gcount[blockIdx.y];
  const TSBranch& br = ts_conv_window(a, gw0);
  const int V = a.V, Ti = a.T, To = a.Tout, Li = Ti * V, Lo = To * V, C = a.C;
  int bc = 0;
  for (int w = 0; w < gcnt; ++w) bc += ts_conv_window(a, gw0 + w).bc;
  auto row_dil = [&](int row) -> int {             // dilation of the window that holds row `row` of the group
    int d = br.dil, r0 = 0;
    for (int w = 0; w < gcnt; ++w) {
      const TSBranch& bw = ts_conv_window(a, gw0 + w);
      if (row >= r0) d = bw.dil;
      r0 += bw.bc;
    }
    return d;
  };
  const int GM = TS_R * V;                                           // V-column part of a row (floats; a multiple of 4)
  const int GS4 = GM >> 2, Q = (V + 3) >> 2;                         // float4 per ge row piece; 16-byte loads per frame row of h
  const int LSd = tsw_ls(GM + TS_R);
  float* Ds = lds;                                                   // [CH][LSd]
  float* Xs = lds + CH * LSd;                                        // [3][CH][LSd]
  f32x2s* SB = reinterpret_cast<f32x2s*>(Xs + 3 * CH * LSd);         // [CH] (scale, shift)
  const f32x2s sb0 = (tid < CH && tid < bc) ? f32x2s{a.scale ? a.scale[br.c0 + tid] : 1.f, a.shift ? a.shift[br.c0 + tid] : 0.f}
                                            : f32x2s{0.f, 0.f};
  const int rel0 = br.c0;
  const int units = a.n * (To / TS_R);
  const int per = (units + a.splits - 1) / a.splits;
  const int u0 = blockIdx.x * per, u1 = min(units, u0 + per);
  const __amdgpu_buffer_rsrc_t rg = ts_rsrc(a.ge, (size_t)a.n * C * Lo * 4);
  const __amdgpu_buffer_rsrc_t rz = ts_rsrc(a.z, (size_t)a.n * C * Li * 4);
  const __amdgpu_buffer_rsrc_t rga = ts_rsrc(a.doaug, (size_t)a.n * C * To * 4);
  const __amdgpu_buffer_rsrc_t rza = ts_rsrc(a.zaug, (size_t)a.n * C * Ti * 4);
  int vD[JD], lD[JD];
#pragma unroll
  for (int j = 0; j < JD; ++j) {
    const int f = tid + TS_NT * j, row = f / GS4, c4 = f - row * GS4;
    const bool ok = row < bc;
    vD[j] = ok ? (row * Lo + 4 * c4) * 4 : TS_OOB;
    lD[j] = ok ? row * LSd + 4 * c4 : -1;
  }
  // h slots: f = tid + 256*j -> (row, tap, frame of the unit, 16-byte piece of the frame row)
  int vX[JX], lX[JX], fX[JX];        // source offset inside the sample (bytes, before the unit's frame), LDS index | count << 20 | row << 23 | valid << 29, frame offset
#pragma unroll
  for (int j = 0; j < JX; ++j) {
    const int f = tid + TS_NT * j, row = f / (12 * Q), r1 = f - row * 12 * Q, tap = r1 / (4 * Q), r2 = r1 - tap * 4 * Q,
              fr = r2 / Q, q = r2 - fr * Q;
    const bool ok = row < bc;
    const int fo = 2 * fr + (tap - 1) * row_dil(row);               // source frame minus 2 * (first output frame of the unit)
    fX[j] = fo;
    vX[j] = ok ? (row * Li + fo * V + 4 * q) * 4 : TS_OOB;
    lX[j] = ((tap * CH + row) * LSd + fr * V + 4 * q) | (min(4, V - 4 * q) << 20) | ((row & 63) << 23) | (ok ? 1 << 29 : 0);
  }
  // the one-column operands: thread < CH: doaug of row tid (4 frames); slots tid + 256*j < CH*12: (row, tap, frame) of zaug
  const bool aokD = tid < CH && tid < bc;
  int vA[JA], lA[JA], fA[JA];
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const int f = tid + TS_NT * j, row = f / 12, r1 = f - row * 12, tap = r1 >> 2, fr = r1 & 3;
    const bool ok = f < CH * 12 && row < bc;
    const int fo = 2 * fr + (tap - 1) * row_dil(row);
    fA[j] = fo;
    vA[j] = ok ? (row * Ti + fo) * 4 : TS_OOB;
    lA[j] = ((tap * CH + row) * LSd + GM + fr) | ((row & 63) << 23) | (ok ? 1 << 29 : 0);
  }
  f32x4 gr[JD], xr[JX], ga;
  float xa[JA];
  float dsum[JD], dsa = 0.f;
#pragma unroll
  for (int j = 0; j < JD; ++j) dsum[j] = 0.f;
  int t2 = 0;                                      // 2 * (first output frame) of the unit being committed
  auto issue = [&](int u) {
    const int n = u / (To / TS_R), tp0 = (u - n * (To / TS_R)) * TS_R;
    t2 = 2 * tp0;
    const int sg = ((n * C + br.c0) * Lo + tp0 * V) * 4;
    const int sx = ((n * C + br.c0) * Li + t2 * V) * 4;
    const int sa = ((n * C + br.c0) * Ti + t2) * 4;
#pragma unroll
    for (int j = 0; j < JD; ++j) gr[j] = ts_load4(rg, vD[j], sg);
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const int fr = t2 + fX[j];
      const bool ok = fr >= 0 && fr < Ti;
      xr[j] = ts_load4(rz, ok ? vX[j] + sx : TS_OOB, 0);
    }
    ga = ts_load4(rga, aokD ? ((n * C + br.c0 + tid) * To + tp0) * 4 : TS_OOB, 0);
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      const int fr = t2 + fA[j];
      const bool ok = fr >= 0 && fr < Ti;
      xa[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rza, ok ? vA[j] + sa : TS_OOB, 0, 0));
    }
  };
  auto act = [&](float x, f32x2s sb, bool rl) -> float {
    const float y = fmaf(x, sb.x, sb.y);
    return rl ? fmaxf(y, 0.f) : y;
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < JD; ++j) {
      if (lD[j] >= 0) {
        f32x2s* d = reinterpret_cast<f32x2s*>(Ds + lD[j]);
        d[0] = f32x2s{gr[j].x, gr[j].y};
        d[1] = f32x2s{gr[j].z, gr[j].w};
        dsum[j] += (gr[j].x + gr[j].y) + (gr[j].z + gr[j].w);
      }
    }
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      if (lX[j] & (1 << 29)) {
        const int fr = t2 + fX[j];
        const bool ok = fr >= 0 && fr < Ti;
        const int row = (lX[j] >> 23) & 63, cnt = (lX[j] >> 20) & 7;
        const f32x2s sb = SB[row];
        const bool rl = rel0 + row < a.n_act;
        float* d = Xs + (lX[j] & 0xfffff);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < cnt) d[k] = ok ? act(xr[j][k], sb, rl) : 0.f;
      }
    }
    if (aokD) {
      f32x2s* d = reinterpret_cast<f32x2s*>(Ds + tid * LSd + GM);
      d[0] = f32x2s{ga.x, ga.y};
      d[1] = f32x2s{ga.z, ga.w};
      dsa += (ga.x + ga.y) + (ga.z + ga.w);
    }
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      if (lA[j] & (1 << 29)) {
        const int fr = t2 + fA[j];
        const bool ok = fr >= 0 && fr < Ti;
        const int row = (lA[j] >> 23) & 63;
        Xs[lA[j] & 0xfffff] = ok ? act(xa[j], SB[row], rel0 + row < a.n_act) : 0.f;
      }
    }
  };
  f32x16 acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  const int mt = CH == 64 ? (wave >> 1) : 0, nt = CH == 64 ? (wave & 1) : 0;
  const int G1 = GS4 + 1;                                             // groups of four positions: the V columns, then the global joint
  const int g0 = CH == 32 ? (G1 * wave) / 4 : 0, g1 = CH == 32 ? (G1 * (wave + 1)) / 4 : G1;
  const float* Ap = Ds + (32 * mt + l31) * LSd + 2 * half;
  const float* Bp = Xs + (32 * nt + l31) * LSd + 2 * half;
  if (u0 < u1) issue(u0);
  {
    f32x4* l4 = reinterpret_cast<f32x4*>(lds);
    for (int i = tid; i < (4 * CH * LSd) >> 2; i += TS_NT) l4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < CH) SB[tid] = sb0;
  }
  __syncthreads();
  for (int u = u0; u < u1; ++u) {
    commit();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (u + 1 < u1) issue(u + 1);
#pragma unroll 2
    for (int g = g0; g < g1; ++g) {
      const f32x2s av = *reinterpret_cast<const f32x2s*>(Ap + 4 * g);
      const f32x2s b0 = *reinterpret_cast<const f32x2s*>(Bp + 4 * g);
      const f32x2s b1 = *reinterpret_cast<const f32x2s*>(Bp + CH * LSd + 4 * g);
      const f32x2s b2 = *reinterpret_cast<const f32x2s*>(Bp + 2 * CH * LSd + 4 * g);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b2.x, acc[2], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b2.y, acc[2], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (CH == 32) {
    float* Rs = lds;                                 // [wave][tap][co][33]
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) Rs[((wave * KT + k) * 32 + ts_row32(r, half)) * 33 + l31] = acc[k][r];
    __syncthreads();
    for (int w = 0, r0 = 0; w < gcnt; ++w) {
      const TSBranch& bw = ts_conv_window(a, gw0 + w);
      const int wb = bw.bc;
      float* dw = bw.dwp + (size_t)blockIdx.x * a.pstride;
      for (int o = tid; o < KT * wb * wb; o += TS_NT) {
        const int co = o / (wb * KT), r2 = o - co * wb * KT, ci = r2 / KT, k = r2 - ci * KT;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) v += Rs[((wv * KT + k) * 32 + r0 + co) * 33 + r0 + ci];
        dw[o] = v;
      }
      r0 += wb;
    }
    __syncthreads();
  } else {
    float* dw = br.dwp + (size_t)blockIdx.x * a.pstride;
    const int ci = 32 * nt + l31;
    if (ci < bc) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * mt + ts_row32(r, half);
        if (co < bc) {
#pragma unroll
          for (int k = 0; k < KT; ++k) dw[((size_t)co * bc + ci) * KT + k] = acc[k][r];
        }
      }
    }
    __syncthreads();
  }
  float* Bs = lds;                                   // [CH][GS4 + 2]
#pragma unroll
  for (int j = 0; j < JD; ++j) {
    const int f = tid + TS_NT * j, row = f / GS4, c4 = f - row * GS4;
    if (row < bc) Bs[row * (GS4 + 2) + c4] = dsum[j];
  }
  if (aokD) Bs[tid * (GS4 + 2) + GS4] = dsa;
  __syncthreads();
  if (tid < bc) {
    float v = 0.f;
    for (int c = 0; c <= GS4; ++c) v += Bs[tid * (GS4 + 2) + c];
    int r0 = 0;
    for (int w = 0; w < gcnt; ++w) {
      const TSBranch& bw = ts_conv_window(a, gw0 + w);
      if (tid >= r0 && tid < r0 + bw.bc) bw.dbp[(size_t)blockIdx.x * a.pstride + tid - r0] = v;
      r0 += bw.bc;
    }
  }
}

int g_ts_exp = 0;                                 // lab builds: dsgcn_tms_split_tuning
constexpr size_t TS_LDS_MAX = 156 * 1024;

template <typename F>
int ts_raise_lds(F* kernel, size_t lds, size_t* have) {
  if (lds > TS_LDS_MAX) return DSGCN_EUNSUPPORTED;
  if (lds > *have) {       // not a stream op: done once per kernel, outside any graph capture (first eager call)
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TS_LDS_MAX);
    if (e != hipSuccess) return (int)e;
    *have = TS_LDS_MAX;
  }
  return 0;
}

// fills the window table and the launch geometry; 1 = eligible, 0 = not, < 0 = bad arguments
int ts_fill(TSArgs& a, int n, int C, int T, int V, int stride, int KT, int nbr, const int* type, const int* c0,
            const int* bc, const int* dil) {
  if (n <= 0 || C <= 0 || T <= 0 || V <= 0 || nbr <= 0 || nbr > TS_MAXBR || !type || !c0 || !bc || !dil) return DSGCN_EINVAL;
  a.n = n; a.C = C; a.T = T; a.V = V; a.nbr = nbr; a.exp = g_ts_exp;
  a.Tout = stride == 2 ? T / 2 : T;
  int nconv = 0, eplanes = 0, wmax = 0, next = 0;
  for (int i = 0; i < nbr; ++i) {
    TSBranch& b = a.br[i];
    b.type = type[i]; b.c0 = c0[i]; b.bc = bc[i]; b.dil = dil[i];
    if (b.bc <= 0 || b.c0 != next) return b.bc <= 0 ? DSGCN_EINVAL : 0;     // windows tile the channels in order
    next += b.bc;
    if (b.type == 0) {
      a.cw[nconv] = i;
      ++nconv;
      wmax = std::max(wmax, b.bc);
      if (b.dil < 1 || b.dil > TS_H) return 0;
    } else if (b.type == 1 || b.type == 2) {
      eplanes += b.bc;
    } else {
      return DSGCN_EINVAL;
    }
  }
  if (next != C) return 0;
  if (KT != 3 || (stride != 1 && stride != 2) || nconv == 0 || wmax > 64 || (V & 1) == 0 || V < 5 || V > 25 || T % 4 ||
      (long)T * V > 2048)
    return 0;
  if (stride == 2 && (T % 8 || T > 128)) return 0;   // (four-frame units of the output; the plane waves hold a column in two registers)
  const long L = (long)T * V;
  if ((long)n * C * L * 4 >= (1L << 31) - 4096 || (long)n * L >= (1L << 31) - 256) return 0;
  if (n > 32768) return 0;
  a.nconv = nconv;
  a.eplanes = eplanes;
  // weight-gradient groups: consecutive conv windows (channel-contiguous, the same ReLU flag) that fit one 32 x 32 tile
  a.ngroups = 0;
  for (int i = 0, k = 0, prev = -2, gsum = 0; i < nbr; ++i) {
    if (a.br[i].type != 0) continue;
    if (a.ngroups > 0 && prev == i - 1 && wmax <= 32 && gsum + a.br[i].bc <= 32) {
      ++a.gcount[a.ngroups - 1];
      gsum += a.br[i].bc;
    } else {
      a.gfirst[a.ngroups] = k;
      a.gcount[a.ngroups] = 1;
      ++a.ngroups;
      gsum = a.br[i].bc;
    }
    prev = i;
    ++k;
  }
  a.haw = 32;                                       // source frames of haug per block ((511 / V + 2) x stride, a halo of 4 either side, the pair read), a power of two
  while (a.haw < (511 / V + 2) * stride + 2 * TS_H + 3) a.haw *= 2;
  return wmax <= 32 ? 1 : 2;                          // = MT
}

// position groups of the launch of one direction: the forward tiles the OUTPUT planes, the data gradient the INPUT planes
void ts_groups(TSArgs& a, bool fwd) {
  const int Tt = fwd ? a.Tout : a.T;
  a.ngrp = (int)(((long)a.n * Tt * a.V + 511) / 512);
  a.ngrpa = (int)(((long)a.n * Tt + 511) / 512);
}

// LDS of the main launch: conv blocks (weight tile | epilogue transposes, then the per-channel table) vs plane blocks
size_t ts_lds_main(TSArgs& a, int MT, bool fwd) {
  int cpmax = 8;
  for (int i = 0; i < a.nbr; ++i)
    if (a.br[i].type == 0) cpmax = std::max(cpmax, (std::min(64, a.br[i].bc) + 7) & ~7);
  const size_t wsf = (size_t)64 * (3 * cpmax + 1) + 64;
  const size_t epf = (size_t)4 * 32 * 36 + (size_t)4 * MT * 32 * 2 * 2;          // transposes + [4][MT*32][2] doubles
  const size_t convf = (std::max(wsf, epf) + 3) & ~(size_t)3;
  a.sboff = (int)convf;
  const size_t planef = a.Tout != a.T ? (size_t)4 * (a.T * a.V + a.Tout * a.V + ((a.T + 3) & ~3) + ((a.Tout + 3) & ~3) + 32)
                                      : (size_t)4 * ((fwd ? 1 : 2) * a.T * a.V + 2 * ((a.T + 3) & ~3) + 32);
  return std::max(convf + 192 + (fwd ? (size_t)64 * a.haw : 0), planef) * sizeof(float);
}

template <bool FWD>
int ts_launch_main(TSArgs& a, int MT, hipStream_t st) {
  const size_t lds = ts_lds_main(a, MT, FWD);
  ts_groups(a, FWD);
  const long blocks = (long)a.nconv * (a.ngrp + a.ngrpa) + ((long)a.n * a.eplanes + 3) / 4;
  if (blocks <= 0 || blocks >= (1L << 31)) return DSGCN_EUNSUPPORTED;
  const dim3 grid((unsigned)blocks), blk(TS_NT);
  if (a.Tout != a.T) {
    if (MT == 1) {
      static size_t have = 64 * 1024;
      const int rc = ts_raise_lds(k_tsp<FWD, 1, true>, lds, &have);
      if (rc) return rc;
      hipLaunchKernelGGL((k_tsp<FWD, 1, true>), grid, blk, lds, st, a);
    } else {
      static size_t have = 64 * 1024;
      const int rc = ts_raise_lds(k_tsp<FWD, 2, true>, lds, &have);
      if (rc) return rc;
      hipLaunchKernelGGL((k_tsp<FWD, 2, true>), grid, blk, lds, st, a);
    }
  } else if (MT == 1) {
    static size_t have = 64 * 1024;
    const int rc = ts_raise_lds(k_tsp<FWD, 1>, lds, &have);
    if (rc) return rc;
    hipLaunchKernelGGL((k_tsp<FWD, 1>), grid, blk, lds, st, a);
  } else {
    static size_t have = 64 * 1024;
    const int rc = ts_raise_lds(k_tsp<FWD, 2>, lds, &have);
    if (rc) return rc;
    hipLaunchKernelGGL((k_tsp<FWD, 2>), grid, blk, lds, st, a);
  }
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" {

#ifdef DSGCN_LAB
// timing experiments (include/dsgcn_lab.h): key 0 = bit mask of parts to skip (k_tsp: 1 plane blocks, 2 epilogue statistics,
// 4 the affine + ReLU of the forward operand, 8 V-column conv blocks, 16 odd-shift loads read as even; k_tspw: 128 the
// affine + ReLU, 256 the global-joint column, 512 odd shifts read as even); results are WRONG with any bit set; key 1 = the
// conv block whose phases are stamped
int dsgcn_tms_split_tuning(int key, int value) {
  if (key == 1) return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tsc_block), &value, sizeof(int));
  if (key != 0) return DSGCN_EINVAL;
  g_ts_exp = value;
  return 0;
}
int dsgcn_tms_split_phases(int which, long long* out) {
  if (which == 1) return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tsc_stamp), sizeof(long long) * 16);
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tsw_stamp), sizeof(long long) * 64);
}
#endif

// which: -1 -> 1 when the shape takes the split-layout path, 0 when it does not;  0 -> rows of the forward statistics
// table (rows, C, 2);  1 -> rows of the data gradient's BatchNorm partial table (rows, C, 2);  2 -> K-splits (rows of the
// weight partials) of the weight gradient.  0 for every `which` when the shape is not eligible.
int dsgcn_tms_split_rows(int which, int n, int C, int T, int V, int stride, int KT, int nbr, const int* type,
                         const int* c0, const int* bc, const int* dil) {
  TSArgs a = {};
  const int mt = ts_fill(a, n, C, T, V, stride, KT, nbr, type, c0, bc, dil);
  if (mt <= 0) return 0;
  if (ts_lds_main(a, mt, false) > TS_LDS_MAX || ts_lds_main(a, mt, true) > TS_LDS_MAX) return 0;
  if (which == -1) return 1;
  if (which == 0) { ts_groups(a, true); return a.ngrp + n; }
  if (which == 1) { ts_groups(a, false); return a.ngrp + n + a.ngrpa; }
  if (which == 2) {
    const int units = n * (a.Tout / TS_R);
    int splits = (mt == 1 ? 512 : 256) / a.ngroups;
    if (splits < 1) splits = 1;
    return std::min(splits, units);
  }
  return 0;
}

// f (n,C,T,V), oaug (n,C,T), stats (rows(0), C, 2) or NULL.
int dsgcn_tms_split_fwd(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                        const float* coeff, float* f, float* oaug, float* stats, int n, int C, int T, int V, int stride,
                        int nbr, const int* type, const int* c0, const int* bc, const int* dil, const float* const* w,
                        const float* const* b, void* stream) {
  if (!z || !zaug || !coeff || !f || !oaug || !w) return DSGCN_EINVAL;
  TSArgs a = {};
  const int mt = ts_fill(a, n, C, T, V, stride, 3, nbr, type, c0, bc, dil);
  if (mt < 0) return mt;
  if (mt == 0) return DSGCN_EUNSUPPORTED;
  a.z = z; a.zaug = zaug; a.scale = scale; a.shift = shift; a.n_act = n_act; a.coeff = coeff;
  a.f = f; a.oaug = oaug; a.stats = stats;
  for (int i = 0; i < nbr; ++i) {
    a.br[i].w = w[i]; a.br[i].b = b ? b[i] : nullptr;
    if (type[i] == 0 && !a.br[i].w) return DSGCN_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  return ts_launch_main<true>(a, mt, st);
}

// ge (n,C,T,V), doaug (n,C,T), pcoef (n*C, V).  gf NULL = zero upstream gradient; A0/B0 NULL = no BatchNorm terms.
int dsgcn_tms_split_prep(const float* gf, const float* f, const float* oaug, const float* coeff, const float* A0,
                         const float* B0, float* ge, float* doaug, float* pcoef, int n, int C, int T, int V,
                         void* stream) {
  if (!f || !oaug || !coeff || !ge || !doaug || !pcoef || n <= 0 || C <= 0 || T <= 0 || V <= 0) return DSGCN_EINVAL;
  if ((T * V) % 4 || V > 32 || (A0 && !B0)) return DSGCN_EUNSUPPORTED;
  const size_t lds = ((size_t)T * V + ((T + 3) & ~3) + 32) * sizeof(float);
  if (lds > 64 * 1024) return DSGCN_EUNSUPPORTED;
  hipLaunchKernelGGL(k_tsp_prep, dim3((unsigned)((long)n * C)), dim3(64), lds, (hipStream_t)stream, gf, f, oaug, coeff,
                     A0, B0, ge, doaug, pcoef, C, T, V);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// dz (n,C,T,V), dzaug (n,C,T), part (rows(1), C, 2): [sum dpre*x, sum dpre] of the branch BatchNorm's backward.
int dsgcn_tms_split_dgrad(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                          const float* ge, const float* doaug, float* dz, float* dzaug, float* part, int n, int C, int T,
                          int V, int stride, int nbr, const int* type, const int* c0, const int* bc, const int* dil,
                          const float* const* w, void* stream) {
  if (!z || !zaug || !ge || !doaug || !dz || !dzaug || !part || !w) return DSGCN_EINVAL;
  TSArgs a = {};
  const int mt = ts_fill(a, n, C, T, V, stride, 3, nbr, type, c0, bc, dil);
  if (mt < 0) return mt;
  if (mt == 0) return DSGCN_EUNSUPPORTED;
  a.z = z; a.zaug = zaug; a.scale = scale; a.shift = shift; a.n_act = n_act;
  a.ge = ge; a.doaug = doaug; a.dz = dz; a.dzaug = dzaug; a.part = part;
  for (int i = 0; i < nbr; ++i) {
    a.br[i].w = w[i];
    if (type[i] == 0 && !a.br[i].w) return DSGCN_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  return ts_launch_main<false>(a, mt, st);
}

// Conv window i writes split s of its weight / bias partials at dwp[i] + s*pstride / dbp[i] + s*pstride, s < splits =
// rows(2); NULL entries for the other window types.
int dsgcn_tms_split_wgrad_jobs(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                               const float* ge, const float* doaug, int n, int C, int T, int V, int stride, int nbr,
                               const int* type, const int* c0, const int* bc, const int* dil, float* const* dwp,
                               float* const* dbp, int splits, int pstride, const dsgcn_bn_coef_job* jobs, int njobs,
                               void* stream) {
  if (!z || !zaug || !ge || !doaug || !dwp || !dbp || splits <= 0) return DSGCN_EINVAL;
  if (njobs < 0 || njobs > BNJ_MAX || (njobs > 0 && !jobs)) return DSGCN_EINVAL;
  BnCoefTable jt = {};
  jt.n = njobs;
  for (int i = 0; i < njobs; ++i) {
    const dsgcn_bn_coef_job& q = jobs[i];
    jt.j[i] = BnCoefJob{q.part, q.mean, q.var, q.gamma, q.coef, q.count, q.eps, q.R, q.C, q.k, q.i_ds, q.i_dh, q.c_affine,
                        q.accumulate};
  }
  if (!bnj_coef_ok(jt)) return DSGCN_EINVAL;
  TSArgs a = {};
  const int mt = ts_fill(a, n, C, T, V, stride, 3, nbr, type, c0, bc, dil);
  if (mt < 0) return mt;
  if (mt == 0) return DSGCN_EUNSUPPORTED;
  a.z = z; a.zaug = zaug; a.scale = scale; a.shift = shift; a.n_act = n_act;
  a.ge = ge; a.doaug = doaug; a.splits = splits; a.pstride = pstride;
  for (int i = 0; i < nbr; ++i) {
    a.br[i].dwp = dwp[i]; a.br[i].dbp = dbp[i];
    if (type[i] == 0 && (!a.br[i].dwp || !a.br[i].dbp)) return DSGCN_EINVAL;
  }
  const int ch = mt == 1 ? 32 : 64;
  const size_t tile = stride == 2 ? (size_t)4 * ch * tsw_ls(TS_R * (V + 1)) + 2 * ch
                                  : (size_t)ch * (tsw_ls(TS_R * (V + 1)) + tsw_ls((TS_R + 2 * TS_H) * (V + 1))) + 2 * ch;
  const size_t red = (size_t)4 * 3 * 32 * 33;
  const size_t lds = std::max(tile, ch == 32 ? red : (size_t)0) * sizeof(float);
  const dim3 grid((unsigned)(splits + bnj_total_blocks(jt)), (unsigned)a.ngroups);
  if (stride == 2) {
    if (ch == 32) {
      static size_t have = 64 * 1024;
      const int rc = ts_raise_lds(k_tspw2<32>, lds, &have);
      if (rc) return rc;
      hipLaunchKernelGGL(k_tspw2<32>, grid, dim3(TS_NT), lds, (hipStream_t)stream, a, jt);
    } else {
      static size_t have = 64 * 1024;
      const int rc = ts_raise_lds(k_tspw2<64>, lds, &have);
      if (rc) return rc;
      hipLaunchKernelGGL(k_tspw2<64>, grid, dim3(TS_NT), lds, (hipStream_t)stream, a, jt);
    }
  } else if (ch == 32) {
    static size_t have = 64 * 1024;
    const int rc = ts_raise_lds(k_tspw<32>, lds, &have);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tspw<32>, grid, dim3(TS_NT), lds, (hipStream_t)stream, a, jt);
  } else {
    static size_t have = 64 * 1024;
    const int rc = ts_raise_lds(k_tspw<64>, lds, &have);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tspw<64>, grid, dim3(TS_NT), lds, (hipStream_t)stream, a, jt);
  }
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_tms_split_wgrad(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                          const float* ge, const float* doaug, int n, int C, int T, int V, int stride, int nbr,
                          const int* type, const int* c0, const int* bc, const int* dil, float* const* dwp, float* const* dbp,
                          int splits, int pstride, void* stream) {
  return dsgcn_tms_split_wgrad_jobs(z, zaug, scale, shift, n_act, ge, doaug, n, C, T, V, stride, nbr, type, c0, bc, dil, dwp,
                                    dbp, splits, pstride, nullptr, 0, stream);
}

}  // extern "C"
