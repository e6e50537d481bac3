// K-C, wide-load form ("pw4"): the 1x1 channel mix with 16-byte-per-lane operand loads.
//
// Same math as pwconv.hip (which documents the reference lines this replaces: gcn.py:2165-2169,2209-2215,2363-2365,
// tcn.py:379-404,422,427): out[n,m,pos] = sum_k A[m,k] * B'[n,k,pos], B' = the "virtual" operand (deferred BatchNorm
// affine / ReLU / second stream applied while loading).  Forward: A = W (m = co, k = ci), B = the layer input.
// Data gradient: A = W^T (m = ci, k = co), B' = dz_eff = gz + A0[co] + B0[co]*z, epilogue = ReLU mask / affine of the
// forward's virtual input + the per-channel sums for d scale / d shift.
//
// Why this form (profiles/r02/kc_counters_before.csv): the first kernel read the B operand with 4-byte-per-lane loads
// (two 128-B segments per wave instruction, 64 different channel rows per wave) and reached 2.2 TB/s with the MFMA pipe
// 23 % busy — HBM saw 128-B visits scattered over rows 6.4 KB apart.  Here a lane owns NQ = 4 (or 2) CONSECUTIVE
// positions of one channel: one buffer_load_dwordx4 per k-step brings two 512-B row segments, and register q of the
// loaded vector is directly the B fragment of position sub-tile q (sub-tile q = positions {NQ*j + q}: the position
// order inside the wave's 32*NQ-position tile is a permutation the store undoes for free, because the four results
// of a lane are again 16 consecutive bytes).  Per k-step a wave issues 1 VMEM load, MT LDS reads and MT*NQ MFMAs
// (v_mfma_f32_32x32x2_f32), with PD k-steps of operand prefetch in flight and no barrier inside the K loop (the whole
// weight block of the workgroup's 32*MT output channels sits in LDS).  Sample base folded into the buffer resource:
// offsets stay 32-bit for any batch size and channels past K read as zero through the bounds check.
#include "common.h"
#include <type_traits>

namespace {

constexpr int P4_NT = 256;
constexpr int P4_OOB = 0x7ffffff0;
constexpr int P4_NT_STORE = 2;                   // cache-policy bits of the output stores: non-temporal (written once, read by a later launch)

struct Pw4Args {
  const float* b1; const float* b2;                                        // B streams (n, K, L); b2 NULL unless MODE 2
  const float* ps1; const float* ph1; const float* ps2; const float* ph2;  // per-k affine (NULL = 1 / 0)
  int relu;
  const float* w; int w_ldm, w_ldk;                                        // A[m][k] = w[m*w_ldm + k*w_ldk]
  const float* bias;                                                       // per m, NULL ok
  float* out;                                                              // (n, M, L)
  float* partial;                                                          // EPI 0: [ngrp][M][2] or NULL
  const float* ex1; const float* ex2;                                      // EPI 1: forward operands at (n, M, L)
  const float* es1; const float* eh1; const float* es2; const float* eh2;
  int erelu;
  float* out2; float* ipart;                                               // EPI 1: d x2 or NULL; [ngrp][M][3] or NULL
  int n, K, M, L, span, WT, cc, Kpad;
  int Lq;                                                                  // positions per plane rounded up to a multiple of NQ (ragged planes)
  // (round 6) up to three convs of ONE shape in a launch (blockIdx.y = which): CTR-GCN refines its topology with three
  // conv4's per unit, each too small to fill the chip (k_pw4 only; the GEMM forms ignore it)
  int ngroup;
  struct Grp { const float* b1; const float* ps1; const float* ph1; const float* w; float* out;
               const float* ex1; const float* es1; const float* eh1; float* ipart; } g[3];
};

template <int NQ> struct VQ;
template <> struct VQ<4> { typedef float T __attribute__((ext_vector_type(4))); };
template <> struct VQ<2> { typedef float T __attribute__((ext_vector_type(2))); };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t p4_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, bytes, 0x00020000);
}

template <int NQ>
__device__ __forceinline__ typename VQ<NQ>::T p4_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  if constexpr (NQ == 4) {
    return __builtin_bit_cast(typename VQ<4>::T, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  } else {
    return __builtin_bit_cast(typename VQ<2>::T, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
  }
}

__device__ __forceinline__ int p4_row32(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Vector offset of row `row` (rowoff = row * plane bytes) for the epilogue's per-row accesses.  The row depends on the
// lane's half, so it must NOT go into the scalar offset: hipcc then wraps every access in a readfirstlane loop that runs
// once per distinct value (two passes with half the lanes each — found in the round-4 disassembly: 35-195 such loops per
// kernel).  An invalid row / position keeps an out-of-range offset (unsigned sum: no wrap below 2^32).
__device__ __forceinline__ int p4_rowoff(bool ok, int ooff, int rowoff) {
  return (int)((unsigned)(ok ? ooff : P4_OOB) + (unsigned)rowoff);
}

template <int NQ>
__device__ __forceinline__ void p4_store(typename VQ<NQ>::T v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  if constexpr (NQ == 4) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, P4_NT_STORE);
  } else {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, P4_NT_STORE);
  }
}

// Sum of half of row l31 of a wave's [32][36] LDS tile (lane (half, l31); the caller adds the two halves).
template <typename ACC>
__device__ __forceinline__ ACC p4_rowread(const float* Tw, int half, int l31) {
  const f32x4* rowp = reinterpret_cast<const f32x4*>(Tw + l31 * 36 + half * 16);
  ACC s = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = rowp[q];
    s += ((ACC)v.x + (ACC)v.y) + ((ACC)v.z + (ACC)v.w);
  }
  return s;
}

// MODE 0: B' = b1;  1: relu?(b1*s1+h1);  2: relu?(b1*s1+h1 + b2*s2+h2).   EPI 0: forward (bias, statistics);  1: data gradient.
struct P4Tile { int wave, half, l31, tid, mBase, n, nrem, ds, pos, grp; bool wlive, pok; int skip; };

// Epilogue of a wave's (32*MT rows) x (32*NQ positions, lane-owned runs of NQ) accumulator tile, shared by k_pw4 and
// k_pwg.  OWNROWS = false: the workgroup's four waves hold the SAME rows at different positions (their per-row sums are
// added through LDS, one partial row per workgroup); true: the waves hold different rows of one position tile (each wave
// writes its rows of the workgroup's partial row itself).
// EPD > 0 (data gradient): the forward operands of the ReLU mask / affine sums are fetched EPD row groups (4 rows each)
// ahead of the group being finished, X2 saying at compile time whether a second stream exists — left to itself the loop
// is load -> wait -> compute -> store per group, eight dependent memory round trips per wave (the 15-27 us epilogue of the
// lab stamps, profiles/r03 / r04).  EPD = 0: the original form (the compiler's own schedule).
template <int MT, int NQ, int EPI, bool OWNROWS, int NWV = 4, int EPD = 0, bool X2 = true>
__device__ __forceinline__ void p4_epilogue(const Pw4Args& a, f32x16 (&acc)[MT][NQ], float* lds, const P4Tile& t) {
  typedef typename VQ<NQ>::T vq;
  const int wave = t.wave, half = t.half, l31 = t.l31, tid = t.tid, mBase = t.mBase, n = t.n, nrem = t.nrem, ds = t.ds,
            pos = t.pos, grp = t.grp;
  const bool wlive = t.wlive, pok = t.pok;
  const int M = a.M, L = a.L, L4 = L * 4;
  (void)L;
  const int skip = t.skip;                         // leading elements of the lane's run that the previous run also holds (0
                                                   // except for the last run of a ragged plane): stored, not summed
  // Epilogue.  Stores go through a per-sample buffer resource (invalid rows / positions get an out-of-range offset
  // and are dropped by the bounds check: no branches); per-channel sums through LDS transposes of the wave's tiles.
  constexpr int NTL = EPI == 0 ? 2 : 3;            // transposed tiles per wave
  float* Tw = lds + wave * (NTL * 32 * 36);
  const __amdgpu_buffer_rsrc_t ro = p4_rsrc(a.out + (size_t)n * M * L, wlive ? nrem * M * L4 : 0);
  const int ooff = pok ? ds * M * L4 + pos * 4 : P4_OOB;
  if (EPI == 0) {
    double* Ss = reinterpret_cast<double*>(lds + NWV * NTL * 32 * 36);  // [waves][MT*32][2]
    const bool stats = a.partial != nullptr;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = p4_row32(r, half);
        const int co = mBase + 32 * m + row;
        vq val;
        float s = 0.f, qq = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          val[q] = acc[m][q][r];
          if (q >= skip) {
            s += val[q];
            qq = fmaf(val[q], val[q], qq);
          }
        }
        p4_store<NQ>(val, ro, p4_rowoff(co < M, ooff, co * L4), 0);
        if (stats) {
          const bool ok = co < M && pok;
          Tw[row * 36 + l31] = ok ? s : 0.f;
          Tw[32 * 36 + row * 36 + l31] = ok ? qq : 0.f;
        }
      }
      if (stats) {
        wave_lds_sync();
        double sd = p4_rowread<double>(Tw, half, l31);
        double qd = p4_rowread<double>(Tw + 32 * 36, half, l31);
        wave_lds_sync();
        sd += __shfl_xor(sd, 32, 64);
        qd += __shfl_xor(qd, 32, 64);
        if (half == 0) {
          if constexpr (OWNROWS) {
            const int co = mBase + 32 * m + l31;
            if (co < M) {
              a.partial[((size_t)grp * M + co) * 2 + 0] = (float)sd;
              a.partial[((size_t)grp * M + co) * 2 + 1] = (float)qd;
            }
          } else {
            Ss[((wave * MT + m) * 32 + l31) * 2 + 0] = sd;
            Ss[((wave * MT + m) * 32 + l31) * 2 + 1] = qd;
          }
        }
      }
    }
    if (stats && !OWNROWS) {
      __syncthreads();
      if (tid < 32 * MT) {
        const int co = mBase + tid;
        if (co < M) {
          double s4 = 0.0, q4 = 0.0;
#pragma unroll
          for (int w = 0; w < 4; ++w) { s4 += Ss[((w * MT * 32) + tid) * 2]; q4 += Ss[((w * MT * 32) + tid) * 2 + 1]; }
          a.partial[((size_t)grp * M + co) * 2 + 0] = (float)s4;
          a.partial[((size_t)grp * M + co) * 2 + 1] = (float)q4;
        }
      }
    }
  } else {
    float* Ss = lds + NWV * NTL * 32 * 36;                               // [waves][MT*32][3]
    f32x4* Es = reinterpret_cast<f32x4*>(Ss + NWV * MT * 32 * 3);        // [MT*32] (s1, h1, s2, h2) of the block's rows
    const bool need_x = a.erelu || a.es1 != nullptr || a.ex2 != nullptr;
    const bool has2 = a.ex2 != nullptr;
    const bool sums = a.ipart != nullptr;
    if constexpr (OWNROWS) Es += wave * 32 * MT;       // every wave its own rows
    const int et = OWNROWS ? (tid & 63) : tid;
    if (et < 32 * MT) {
      const int ci = mBase + et;
      f32x4 p = {1.f, 0.f, 1.f, 0.f};
      if (ci < M) {
        if (a.es1) { p.x = a.es1[ci]; p.y = a.eh1[ci]; }
        if (a.es2) { p.z = a.es2[ci]; p.w = a.eh2[ci]; }
      }
      Es[et] = p;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rx1 = p4_rsrc(a.ex1 + (size_t)n * M * L, (wlive && need_x) ? nrem * M * L4 : 0);
    const __amdgpu_buffer_rsrc_t rx2 = p4_rsrc((has2 ? a.ex2 : a.ex1) + (size_t)n * M * L, (wlive && has2) ? nrem * M * L4 : 0);
    const __amdgpu_buffer_rsrc_t ro2 = p4_rsrc((a.out2 ? a.out2 : a.out) + (size_t)n * M * L, (wlive && a.out2) ? nrem * M * L4 : 0);
    constexpr int G = MT * 4;                       // row groups of the wave's tile
    constexpr int PD = EPD > 0 ? (EPD < G ? EPD : G) : 1;
    vq xa[PD][4], xb[(X2 || EPD == 0) ? PD : 1][4];
    auto fetch = [&](int g, int slot) {
      const int m = g >> 2, rb = (g & 3) * 4;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int ci = mBase + 32 * m + p4_row32(rb + rr, half);
        xa[slot][rr] = p4_load<NQ>(rx1, p4_rowoff(ci < M, ooff, ci * L4), 0);      // zeros when the input is not needed
        if constexpr (X2 || EPD == 0) xb[slot][rr] = p4_load<NQ>(rx2, p4_rowoff(ci < M, ooff, ci * L4), 0);
      }
    };
    if constexpr (EPD > 0) {
#pragma unroll
      for (int g = 0; g < PD; ++g) fetch(g, g);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int m = g >> 2, rb = (g & 3) * 4;
      const int slot = EPD > 0 ? g % PD : 0;
      if constexpr (EPD == 0) fetch(g, 0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = rb + rr;
        const int row = p4_row32(r, half);
        const int ci = mBase + 32 * m + row;
        const f32x4 e = Es[32 * m + row];
        float u0 = 0.f, u1 = 0.f, u2 = 0.f;
        vq d1, d2;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          float pre = fmaf(xa[slot][rr][q], e.x, e.y);
          float xbq = 0.f;
          if constexpr (X2 || EPD == 0) {
            xbq = xb[slot][rr][q];
            if (has2) pre += fmaf(xbq, e.z, e.w);
          }
          const float dv = (!a.erelu || pre > 0.f) ? acc[m][q][r] : 0.f;
          d1[q] = dv * e.x;
          d2[q] = dv * e.z;
          if (q >= skip) {
            u0 = fmaf(dv, xa[slot][rr][q], u0);
            u1 += dv;
            u2 = fmaf(dv, xbq, u2);
          }
        }
        p4_store<NQ>(d1, ro, p4_rowoff(ci < M, ooff, ci * L4), 0);
        if constexpr (X2 || EPD == 0) p4_store<NQ>(d2, ro2, p4_rowoff(ci < M, ooff, ci * L4), 0);   // zero-sized resource when there is no dx2
        if (sums) {
          const bool ok = ci < M && pok;
          Tw[row * 36 + l31] = ok ? u0 : 0.f;
          Tw[32 * 36 + row * 36 + l31] = ok ? u1 : 0.f;
          Tw[2 * 32 * 36 + row * 36 + l31] = ok ? u2 : 0.f;
        }
      }
      if constexpr (EPD > 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + PD < G) fetch(g + PD, slot);
        __builtin_amdgcn_sched_barrier(0);
      }
      if ((g & 3) == 3 && sums) {
        wave_lds_sync();
        float s0 = p4_rowread<float>(Tw, half, l31);
        float s1 = p4_rowread<float>(Tw + 32 * 36, half, l31);
        float s2 = p4_rowread<float>(Tw + 2 * 32 * 36, half, l31);
        wave_lds_sync();
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (half == 0) {
          if constexpr (OWNROWS) {
            const int ci = mBase + 32 * m + l31;
            if (ci < M) {
              float* o = a.ipart + ((size_t)grp * M + ci) * 3;
              o[0] = s0; o[1] = s1; o[2] = s2;
            }
          } else {
            float* q = Ss + ((wave * MT + m) * 32 + l31) * 3;
            q[0] = s0; q[1] = s1; q[2] = s2;
          }
        }
      }
    }
    if (sums && !OWNROWS) {
      __syncthreads();
      if (tid < 32 * MT) {
        const int ci = mBase + tid;
        if (ci < M) {
          float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const float* q = Ss + ((w * MT * 32) + tid) * 3;
            v0 += q[0]; v1 += q[1]; v2 += q[2];
          }
          float* o = a.ipart + ((size_t)grp * M + ci) * 3;
          o[0] = v0; o[1] = v1; o[2] = v2;
        }
      }
    }
  }
}

// KSP (round 6): the four waves of a workgroup share ONE position tile and split the K loop (whole prefetch rounds each);
// their accumulators meet in LDS and wave 0 runs the epilogue.  For the tiny-plane launches (the dynamic-adjacency
// projections: n x 32 positions, K = 128 .. 288): 64 wave tiles x 9 row blocks left the chip at 576 waves each walking
// K / 2 dependent k-steps (18-21 us for 0.6 GFLOP); split four ways the chain is a quarter as long on four times the waves.
template <int MT, int NQ, int MODE, int PD, int EPI, bool KSP = false>
__global__ __launch_bounds__(P4_NT, (MT * NQ >= 6 ? 2 : (MT * NQ >= 4 ? 3 : 4))) void k_pw4(Pw4Args a_) {
  typedef typename VQ<NQ>::T vq;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  Pw4Args a = a_;
  if (a_.ngroup > 1) {                             // grouped launch: this workgroup's conv
    const Pw4Args::Grp& q = a_.g[blockIdx.y];
    a.b1 = q.b1; a.ps1 = q.ps1; a.ph1 = q.ph1; a.w = q.w; a.out = q.out;
    a.ex1 = q.ex1; a.es1 = q.es1; a.eh1 = q.eh1; a.ipart = q.ipart;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  // XCD-aware decode: the cc workgroups that read the same position tiles (one per 32*MT output channels) take
  // consecutive slots of one XCD (blockIdx % 8), so the re-reads are served by that XCD's L2
  const int ngrp = KSP ? a.WT : (a.WT + 3) >> 2;
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int cz = slot % a.cc;
  const int grp = (slot / a.cc) * 8 + xcd;
  if (grp >= ngrp) return;
  const int mBase = cz * 32 * MT;
  const int K = a.K, M = a.M, L = a.L;
  const int Kpad = a.Kpad, KP = Kpad + 1;
  float* Ws = lds;                                                       // [32*MT][KP], zero beyond (M, K)
  f32x4* Ps = reinterpret_cast<f32x4*>(lds + ((32 * MT * KP + 2 + 3) & ~3));  // [Kpad + 2] (s1, h1, s2, h2)

  // Position tiles run over the planes of all samples back to back (a tile may straddle samples: L % NQ == 0, so a lane's
  // NQ positions never do): no per-sample tail tile — at L = 400 (256 channels, 16 frames) per-sample tiling left
  // 22 % of the MFMA work on padding.  The wave's buffer resources start at its first sample n; a lane adds ds sample
  // strides in its vector offset.
  const int wt = KSP ? grp : grp * 4 + wave;
  const bool wlive = wt < a.WT;
  // Ragged planes (L % NQ != 0: K400's 25 x 17 and CTR-GCN's 25 x 25 planes): the tile walks Lq = L rounded up to NQ
  // positions per plane, and the plane's last run is moved back to END at the plane's end — it overlaps the run before it
  // by `skip` positions, which both lanes compute and store identically and only the earlier one adds to the per-channel
  // sums.  No load or store ever leaves the plane (a run reaching into the next row would need a per-element bounds
  // check: a 16-byte buffer load that straddles the end of its resource returns zeros from its second dword on, measured),
  // at the price of dword-aligned 16-byte accesses (legal and within 4 % of aligned ones on gfx950:
  // tools/probes/unaligned_b128.hip).
  const int Lq = a.Lq;
  const int g0 = (wlive ? wt : 0) * (32 * NQ);          // < 2^31 (p4_plan)
  const int n = g0 / Lq;
  int pos = g0 - n * Lq + l31 * NQ;
  int ds = 0;
  while (pos >= Lq) { pos -= Lq; ++ds; }
  const bool pok = wlive && n + ds < a.n;
  int skip = 0;
  if (L - pos < NQ) { skip = NQ - (L - pos); pos = L - NQ; }
  const int L4 = L * 4;
  const int nrem = a.n - n < a.span ? a.n - n : a.span;     // samples the wave can touch
  const int voff = pok ? ds * K * L4 + (half * L + pos) * 4 : P4_OOB;
  const __amdgpu_buffer_rsrc_t r1 = p4_rsrc(a.b1 + (size_t)n * K * L, wlive ? nrem * K * L4 : 0);
  const __amdgpu_buffer_rsrc_t r2 = p4_rsrc((MODE == 2 ? a.b2 : a.b1) + (size_t)n * K * L, (wlive && MODE == 2) ? nrem * K * L4 : 0);

  f32x16 acc[MT][NQ];
  const float lo = a.relu ? 0.f : -__builtin_inff();
  // this wave's k-steps: all of them, or (KSP) its share of the Kpad / (2 PD) prefetch rounds
  const int KS = a.Kpad >> 1;                      // k-steps (2 channels each), a multiple of PD
  const int KSr = (K + 1) >> 1;                    // k-steps that hold real channels
  int ks0 = 0, ks1 = KS;
  if constexpr (KSP) {
    const int U = KS / PD;
    ks0 = (U * wave / 4) * PD;
    ks1 = (U * (wave + 1) / 4) * PD;
  }
  const int kse = ks1 < KSr ? ks1 : KSr;           // loads past it: out of range (zeros, no traffic)
  // ---- weights: global -> registers (all loads of a batch issued together), operand prefetch, then LDS ----
  constexpr int WB = 16;
  const bool mfast = a.w_ldm == 1;                 // A = W^T (data gradient): m is the contiguous index of w
  // element e of this thread: k fast: (r, k) = ((tid>>4) + 16*(e % (2*MT)), (tid&15) + 16*(e / (2*MT)))
  //                           m fast: (r, k) = ((tid&31) + 32*(e % MT),     (tid>>5) + 8*(e / MT))
  const int nel = mfast ? (Kpad >> 3) * MT : (Kpad >> 4) * 2 * MT;
  vq buf1[PD], buf2[MODE == 2 ? PD : 1];
  for (int e0 = 0; e0 < nel; e0 += WB) {
    float tmp[WB];
#pragma unroll
    for (int j = 0; j < WB; ++j) {
      const int e = e0 + j;
      int r, k;
      if (mfast) { r = (tid & 31) + 32 * (e % MT); k = (tid >> 5) + 8 * (e / MT); }
      else { r = (tid >> 4) + 16 * (e % (2 * MT)); k = (tid & 15) + 16 * (e / (2 * MT)); }
      const int m = mBase + r;
      tmp[j] = (e < nel && m < M && k < K) ? a.w[(size_t)m * a.w_ldm + (size_t)k * a.w_ldk] : 0.f;
    }
    if (e0 == 0) {
#pragma unroll
      for (int u = 0; u < PD; ++u) {
        const int s0 = ks0 + u < kse ? 2 * (ks0 + u) * L4 : P4_OOB;   // (channels past K: out of range, zeros)
        buf1[u] = p4_load<NQ>(r1, voff, s0);
        if constexpr (MODE == 2) buf2[u] = p4_load<NQ>(r2, voff, s0);
      }
    }
#pragma unroll
    for (int j = 0; j < WB; ++j) {
      const int e = e0 + j;
      int r, k;
      if (mfast) { r = (tid & 31) + 32 * (e % MT); k = (tid >> 5) + 8 * (e / MT); }
      else { r = (tid >> 4) + 16 * (e % (2 * MT)); k = (tid & 15) + 16 * (e / (2 * MT)); }
      if (e < nel) Ws[r * KP + k] = tmp[j];
    }
  }
  // accumulators start at the bias of their output row (forward), so the epilogue has no per-row loads
  // (loaded here, with the affine table: after the barrier they were a memory round trip of their own)
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = mBase + 32 * m + p4_row32(i, half);
      const float b0 = (EPI == 0 && a.bias && row < M && (!KSP || wave == 0)) ? a.bias[row] : 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[m][q][i] = b0;
    }

  if (MODE != 0) {
    for (int i = tid; i < Kpad; i += P4_NT) {
      f32x4 p = {0.f, 0.f, 0.f, 0.f};
      if (i < K) {
        p.x = a.ps1 ? a.ps1[i] : 1.f;
        p.y = a.ph1 ? a.ph1[i] : 0.f;
        p.z = a.ps2 ? a.ps2[i] : 1.f;
        p.w = a.ph2 ? a.ph2[i] : 0.f;
      }
      Ps[i] = p;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                    // raw barrier: the operand prefetch stays in flight

  // Software pipeline, pinned with scheduling barriers.  Step ks: start the LDS reads of step ks+1 (A fragment, affine
  // row; double-buffered by step parity), apply the affine to the operand loaded PD steps ago, run the MT*NQ MFMAs, then
  // re-issue that operand buffer's load for step ks+PD (after the MFMAs: the buffer registers are dead by then, so the
  // load lands in place).  Left to itself the compiler sinks all PD loads to the end of the unrolled body and waits for
  // the first of them at the top of the next one — or, with the load ahead of the MFMAs, rotates the PD buffers through
  // v_mov chains behind a vmcnt(0) (profiles/r02: matrix pipe 61 % busy at 256 -> 256 channels).
  float avb[2][MT];
  f32x4 pb[2] = {{1.f, 0.f, 1.f, 0.f}, {1.f, 0.f, 1.f, 0.f}};
#pragma unroll
  for (int m = 0; m < MT; ++m) avb[0][m] = Ws[(32 * m + l31) * KP + 2 * ks0 + half];
  if (MODE != 0) pb[0] = Ps[2 * ks0 + half];
  for (int base = ks0; base < ks1; base += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int ks = base + u;
      const int cur = u & 1, nxt = cur ^ 1;                              // PD is even: the parity survives the back edge
      const int kn = 2 * (ks + 1) + half;                                // last step: reads the LDS pad, never used
#pragma unroll
      for (int m = 0; m < MT; ++m) avb[nxt][m] = Ws[(32 * m + l31) * KP + kn];
      if (MODE != 0) pb[nxt] = Ps[kn];
      vq b = buf1[u];
      if (MODE != 0) {
        const f32x4 p = pb[cur];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          float v = fmaf(b[q], p.x, p.y);
          if constexpr (MODE == 2) v += fmaf(buf2[u][q], p.z, p.w);
          b[q] = fmaxf(v, lo);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(avb[cur][m], b[q], acc[m][q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // past K: the scalar offset jumps out of the buffer's range (zeros, no traffic: the bounds check covers it)
      const int sn = ks + PD < kse ? 2 * (ks + PD) * L4 : P4_OOB;
      buf1[u] = p4_load<NQ>(r1, voff, sn);
      if constexpr (MODE == 2) buf2[u] = p4_load<NQ>(r2, voff, sn);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();                                 // every wave is done with Ws / Ps: LDS is reused below

  if constexpr (KSP) {
    // the K shares meet: waves 1..3 park their accumulators ([wave][register][lane]: conflict-free), wave 0 adds them in
    // a fixed order (deterministic) and alone runs the epilogue — the others go through it as dead waves (no stores,
    // zero sums: what a partly empty last workgroup's waves do)
    float* Rs = lds;                               // [3][MT * NQ * 16][64]
    if (wave != 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int i = 0; i < 16; ++i) Rs[(((wave - 1) * MT * NQ + m * NQ + q) * 16 + i) * 64 + lane] = acc[m][q][i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][q][i] += Rs[((w * MT * NQ + m * NQ + q) * 16 + i) * 64 + lane];
    }
    __syncthreads();
  }
  const bool elive = KSP ? (wlive && wave == 0) : wlive;
  const P4Tile tile = {wave, half, l31, tid, mBase, n, nrem, ds, pos, grp, elive, KSP ? (pok && wave == 0) : pok, skip};
  if constexpr (EPI == 1) {
    // (operand prefetch of the data-gradient epilogue: two row groups ahead — k_pw4's waves keep their PD operand slots
    // next to MT*NQ accumulator tiles, there is room for two)
    if (a.ex2) p4_epilogue<MT, NQ, EPI, false, 4, (MT * NQ >= 8 || (MT == 2 && NQ == 2) ? 1 : 2), true>(a, acc, lds, tile);
    else p4_epilogue<MT, NQ, EPI, false, 4, (MT * NQ >= 8 || (MT == 2 && NQ == 2) ? 1 : 2), false>(a, acc, lds, tile);
  } else {
    p4_epilogue<MT, NQ, EPI, false>(a, acc, lds, tile);
  }
}


// ---- K-C, GEMM form with three-term bf16 products ("pwg") ----------------------------------------------------------
// For convs wide on both sides (the fp32 MFMA form is bound by the matrix pipe: 85 TF of 157 at 256 -> 256) the product
// runs as six v_mfma_f32_32x32x16_bf16 per 16 channels on the exact three-way bf16 split of both operands (common.h).
// The split costs ~5 VALU operations per value, so it must not be repeated per output-row block the way k_pw4 re-reads
// its B operand: here a workgroup owns 128 output rows x 128 positions, its four waves hold 32 rows each of the SAME
// position tile, and both operand tiles of a 32-channel chunk are split once by the thread that loaded them and shared
// through LDS — rows of 32 bf16 (+16 B pad: conflict-free 16-byte fragment reads) per term, k contiguous.  The loader
// thread holds a 4 x 4 register block (4 channels x its lane's 4 positions; for W^T 4 k x 4 rows), so the k-contiguous
// image is written with 8-byte stores and no transposed reads are needed; position sub-tile q is again {4j + q}: the
// accumulators have k_pw4's layout and the epilogue is shared (p4_epilogue, OWNROWS).
constexpr int PG_KC = 32, PG_RB = PG_KC * 2 + 16, PG_T = 128, PG_TAM = 144;

template <int MODE, int EPI>
__global__ __launch_bounds__(P4_NT, 2) void k_pwg(Pw4Args a) {
  typedef VQ<4>::T vq;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* Ab = reinterpret_cast<char*>(lds);                               // [3][128 rows][RB]
  // W^T arrives four consecutive ROWS per loader thread: written in row order the 8-byte stores of a wave would sit 320 B
  // apart (16-way bank conflicts: profiles/r02 counters); row 4a + rr is therefore kept at LDS row 36*rr + a (lanes 80 B
  // apart on the store side, and the 16-byte fragment reads of a wave's 32 rows — 4 row groups x 8 — stay conflict-free)
  constexpr int TA = EPI == 1 ? PG_TAM : PG_T;                           // A rows per term
  char* Bb = Ab + 3 * TA * PG_RB;                                        // [3][128 position slots][RB]
  f32x4* Ps = reinterpret_cast<f32x4*>(Bb + 3 * PG_T * PG_RB);           // [Kpad] (s1, h1, s2, h2)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int ngrp = a.WT;
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int cz = slot % a.cc;
  const int grp = (slot / a.cc) * 8 + xcd;
  if (grp >= ngrp) return;
  const int mBase0 = cz * PG_T, mBase = mBase0 + 32 * wave;
  const int K = a.K, M = a.M, L = a.L, Kpad = a.Kpad;
  const int Lq = a.Lq;                             // (ragged planes: see k_pw4)
  const int g0 = grp * 128;
  const int n = g0 / Lq;
  int pos = g0 - n * Lq + l31 * 4;
  int ds = 0;
  while (pos >= Lq) { pos -= Lq; ++ds; }
  const bool pok = n + ds < a.n;
  int skip = 0;
  if (L - pos < 4) { skip = 4 - (L - pos); pos = L - 4; }
  const int L4 = L * 4;
  const int nrem = a.n - n < a.span ? a.n - n : a.span;
  const __amdgpu_buffer_rsrc_t r1 = p4_rsrc(a.b1 + (size_t)n * K * L, nrem * K * L4);
  const __amdgpu_buffer_rsrc_t r2 = p4_rsrc((MODE == 2 ? a.b2 : a.b1) + (size_t)n * K * L, MODE == 2 ? nrem * K * L4 : 0);
  // loader roles.  B: channels 4*cg .. 4*cg+3 of the chunk (cg = tid / 32) at the lane's own position quad.
  const int cg = tid >> 5;
  const int voffB = pok ? ds * K * L4 + pos * 4 : P4_OOB;
  // A, k contiguous in memory (forward): rows (tid / 8) + 32*i, k = 4*(tid % 8) ..+3;  m contiguous (data gradient: W^T):
  // k = 4*(tid / 32) + e, rows 4*(tid % 32) ..+3
  constexpr bool mfast = EPI == 1;                 // data gradient: A = W^T, m contiguous in memory
  f32x4 aw[4];
  vq bwA[4], bwB[4], bw2A[MODE == 2 ? 4 : 1], bw2B[MODE == 2 ? 4 : 1];     // two chunks of B in flight (ping-pong sets)
  auto issueB = [&](int ch0, vq (&bw)[4], vq (&bw2)[MODE == 2 ? 4 : 1]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = ch0 + 4 * cg + e;                 // per half-wave: the channel goes into the vector offset
      const int vo = c < K ? voffB + c * L4 : P4_OOB;
      bw[e] = p4_load<4>(r1, vo, 0);
      if constexpr (MODE == 2) bw2[e] = p4_load<4>(r2, vo, 0);
    }
  };
  // weights through a buffer resource too: rows / channels past the matrix are out-of-range offsets (zeros), no branches
  const __amdgpu_buffer_rsrc_t rw = p4_rsrc(a.w, M * K * 4);
  const int offA0 = mfast ? (4 * cg * a.w_ldk + mBase0 + 4 * l31) * 4 : ((mBase0 + (tid >> 3)) * a.w_ldm + 4 * (tid & 7)) * 4;
  const int stepE = mfast ? a.w_ldk * 4 : 32 * a.w_ldm * 4;             // between this thread's four loads
  const int stepC = mfast ? a.w_ldk * 4 : 4;                            // per channel of chunk advance
  const int kA = mfast ? 4 * cg : 4 * (tid & 7), mA = mfast ? mBase0 + 4 * l31 : mBase0 + (tid >> 3);
  auto issueA = [&](int ch0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = (kA + ch0 + (mfast ? e : 0) < K) & (mA + (mfast ? 0 : 32 * e) < M);
      aw[e] = p4_load<4>(rw, ok ? offA0 + e * stepE + ch0 * stepC : P4_OOB, 0);
    }
  };
  const float lo = a.relu ? 0.f : -__builtin_inff();
  auto put = [&](char* base, int tstride, float v0, float v1, float v2, float v3) {   // four consecutive k of one row -> 8 B per term
    unsigned p0, p1, p2, q0, q1, q2;
    b3_split(v0, v1, p0, p1, p2);
    b3_split(v2, v3, q0, q1, q2);
    *reinterpret_cast<u32x2v*>(base) = u32x2v{p0, q0};
    *reinterpret_cast<u32x2v*>(base + tstride) = u32x2v{p1, q1};
    *reinterpret_cast<u32x2v*>(base + 2 * tstride) = u32x2v{p2, q2};
  };
  auto commit = [&](int ch0, vq (&bw)[4], vq (&bw2)[MODE == 2 ? 4 : 1]) {
    if (mfast) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) put(Ab + (36 * rr + l31) * PG_RB + cg * 8, TA * PG_RB, aw[0][rr], aw[1][rr], aw[2][rr], aw[3][rr]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) put(Ab + ((tid >> 3) + 32 * e) * PG_RB + (tid & 7) * 8, TA * PG_RB, aw[e].x, aw[e].y, aw[e].z, aw[e].w);
    }
    f32x4 pr[4];
    if (MODE != 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) pr[e] = Ps[ch0 + 4 * cg + e];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = bw[e][q];
        if (MODE != 0) {
          x = fmaf(x, pr[e].x, pr[e].y);
          if constexpr (MODE == 2) x += fmaf(bw2[e][q], pr[e].z, pr[e].w);
          x = fmaxf(x, lo);
        }
        v[e] = x;
      }
      put(Bb + (32 * q + l31) * PG_RB + cg * 8, PG_T * PG_RB, v[0], v[1], v[2], v[3]);
    }
  };

  if (MODE != 0) {                                 // (before the operand loads: its own loads end in a vmcnt(0))
    for (int i = tid; i < Kpad; i += P4_NT) {
      f32x4 p = {0.f, 0.f, 0.f, 0.f};
      if (i < K) {
        p.x = a.ps1 ? a.ps1[i] : 1.f;
        p.y = a.ph1 ? a.ph1[i] : 0.f;
        p.z = a.ps2 ? a.ps2[i] : 1.f;
        p.w = a.ph2 ? a.ph2[i] : 0.f;
      }
      Ps[i] = p;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  issueA(0);
  issueB(0, bwA, bw2A);
  issueB(PG_KC, bwB, bw2B);
  __builtin_amdgcn_sched_barrier(0);
  f32x16 acc[1][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = mBase + p4_row32(i, half);
    const float b0 = (EPI == 0 && a.bias && row < M) ? a.bias[row] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[0][q][i] = b0;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                    // Ps visible (raw barrier: the operand loads stay in flight)
  const char* Af = Ab + (mfast ? 36 * (l31 & 3) + 8 * wave + (l31 >> 2) : 32 * wave + l31) * PG_RB + 16 * half;
  const char* Bf = Bb + l31 * PG_RB + 16 * half;
  auto chunk = [&](int ch0, vq (&bw)[4], vq (&bw2)[MODE == 2 ? 4 : 1]) {
    commit(ch0, bw, bw2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issueA(ch0 + PG_KC);                           // (past K: zeros)
    issueB(ch0 + 2 * PG_KC, bw, bw2);              // this set is free again: two chunks ahead
    __builtin_amdgcn_sched_barrier(0);             // pinned here, in this order: the next commit's vmcnt leaves this set in flight
#pragma unroll
    for (int ks = 0; ks < PG_KC / 16; ++ks) {
      bf16x8 af[3];
#pragma unroll
      for (int t = 0; t < 3; ++t)
        af[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(Af + t * TA * PG_RB + 32 * ks));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bf16x8 bf[3];
#pragma unroll
        for (int t = 0; t < 3; ++t)
          bf[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(Bf + (t * PG_T + 32 * q) * PG_RB + 32 * ks));
        f32x16 c = acc[0][q];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], c, 0, 0, 0);
        acc[0][q] = c;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // raw barrier: the loads in flight stay in flight
  };
  // (pairs in the loop, an odd last chunk outside it: with the second call conditional inside the loop the compiler
  // must allow for a first-call -> first-call path, on which the set just refilled is the next one consumed: vmcnt(0))
  int ch0 = 0;
  for (; ch0 + PG_KC < Kpad; ch0 += 2 * PG_KC) {
    chunk(ch0, bwA, bw2A);
    chunk(ch0 + PG_KC, bwB, bw2B);
  }
  if (ch0 < Kpad) chunk(ch0, bwA, bw2A);
  __syncthreads();                                 // drains the read-ahead loads before LDS is reused
  const P4Tile tile = {wave, half, l31, tid, mBase, n, nrem, ds, pos, grp, true, pok, skip};
  if constexpr (EPI == 1) {
    if (a.ex2) p4_epilogue<1, 4, EPI, true, 4, 4, true>(a, acc, lds, tile);
    else p4_epilogue<1, 4, EPI, true, 4, 4, false>(a, acc, lds, tile);
  } else {
    p4_epilogue<1, 4, EPI, true>(a, acc, lds, tile);
  }
}


// ---- K-C, GEMM form with the weights split ahead of the launch -----------------------------------------------------------
// k_pwg spends as many VALU cycles on the three-way bf16 split as the matrix core spends on the six products (rocprofv3
// counters, profiles/r03: 9.8 M VALU instructions x 4 cycles against 39 M MFMA-busy cycles per 256 -> 256 launch, matrix
// pipe 27 % busy) and half of that split is the WEIGHT tile, the same for every workgroup and every chunk revisit: the
// weights are therefore split once per conv and step (k_wsplit below, the W and the W^T image).  Round 3's consumer of
// that image (k_pwg2: eight waves on one position tile, the weight chunk copied through LDS, two barriers per chunk) is
// gone; k_pwg3 below replaced it (same-box A/B of the step: 13.03 -> 12.55 ms, profiles/r04).
#ifdef DSGCN_LAB
// wall-clock stamps (10 ns) of one workgroup's thread 0 (include/dsgcn_lab.h: dsgcn_pwg2_phases), [63] = count
__device__ long long g_pwg_stamp[64];
__device__ int g_pwg_stamp_block = 0;            // which workgroup stamps (dsgcn_pwg2_phases_block)
#define PWG_STAMP_BLOCK g_pwg_stamp_block
#define PWG_STAMP() do { if (blockIdx.x == PWG_STAMP_BLOCK && threadIdx.x == 0 && nst < 62) g_pwg_stamp[nst++] = wall_clock64(); } while (0)
#else
#define PWG_STAMP() do {} while (0)
#endif

// ---- K-C, GEMM form, third generation ("pwg3"): two independent workgroups per CU, one barrier per chunk ------------
// What the lab stamps of round 3's k_pwg2 showed (profiles/r03/README.md): a workgroup's life is prologue -> 8 x (commit + barrier +
// products + barrier) -> epilogue with all eight waves of the CU in lockstep; the matrix pipe idles through every commit
// and through the HBM-bound epilogue (15 us of 45 in the data gradient), and 400 position tiles on 256 one-workgroup CUs
// make two rounds.  Here:
//   * a workgroup is FOUR waves that own 32*MT rows each (MT = 2: 256 rows) of the same 128-position tile: 8 accumulator
//     tiles per wave, two workgroups resident per CU (2 x 66 KB of LDS, <= 256 registers), so one workgroup's prologue /
//     epilogue / barrier waits run under the other's products — the tiles of a launch fit in one round;
//   * the weight terms never touch LDS: the image is written once per conv in FRAGMENT order (k_wsplit, frag = 1:
//     [term][row tile][k-step][lane][8 bf16], a wave's A fragment is one contiguous 1 KB) and every wave loads its own
//     rows' fragments straight into registers one k-step ahead (k_pwg2 tried this on the row-major image — 64 row
//     segments of 32 B per load — and lost);
//   * the activation chunk (32 channels x 128 positions, three bf16 terms) is double-buffered in LDS: chunk i+1 is split
//     and written while chunk i is multiplied (the split's VALU work sits in the MFMA shadow), ONE barrier per chunk.
// vmcnt is in order, so a wave's wait for its next A fragments also completes the older activation loads: the
// activation prefetch distance is one chunk by construction (a second register set would buy nothing).
constexpr int G3_BBUF = 3 * PG_T * PG_RB;          // one activation buffer: [3 terms][128 position slots][RB]

template <int MODE, int EPI, int MT>
__global__ __launch_bounds__(256, 2) void k_pwg3(Pw4Args a, const unsigned short* __restrict__ wfr, int RT) {
  typedef VQ<4>::T vq;
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef DSGCN_LAB
  int nst = 0;
#endif
  PWG_STAMP();
  char* Bb = reinterpret_cast<char*>(lds);                               // [2][3][128 position slots][RB]
  f32x4* Ps = reinterpret_cast<f32x4*>(Bb + 2 * G3_BBUF);                // [Kpad] (s1, h1, s2, h2)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int ngrp = a.WT;
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int cz = slot % a.cc;
  const int grp = (slot / a.cc) * 8 + xcd;
  if (grp >= ngrp) return;
  const int mBase0 = cz * (128 * MT), mBase = mBase0 + 32 * MT * wave;
  const int K = a.K, M = a.M, L = a.L, Kpad = a.Kpad;
  const int Lq = a.Lq;                             // (ragged planes: see k_pw4)
  const int g0 = grp * 128;
  const int n = g0 / Lq;
  int pos = g0 - n * Lq + l31 * 4;
  int ds = 0;
  while (pos >= Lq) { pos -= Lq; ++ds; }
  const bool pok = n + ds < a.n;
  int skip = 0;
  if (L - pos < 4) { skip = 4 - (L - pos); pos = L - 4; }
  const int L4 = L * 4;
  const int nrem = a.n - n < a.span ? a.n - n : a.span;
  const __amdgpu_buffer_rsrc_t r1 = p4_rsrc(a.b1 + (size_t)n * K * L, nrem * K * L4);
  const __amdgpu_buffer_rsrc_t r2 = p4_rsrc((MODE == 2 ? a.b2 : a.b1) + (size_t)n * K * L, MODE == 2 ? nrem * K * L4 : 0);
  // B loader: channels 4*cg .. 4*cg+3 of the chunk (cg = tid / 32) at the lane's own position quad
  const int cg = tid >> 5;
  const int voffB = pok ? ds * K * L4 + pos * 4 : P4_OOB;
  vq bw[4], bw2[MODE == 2 ? 4 : 1];
  auto issueB = [&](int ch0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = ch0 + 4 * cg + e;
      const int vo = c < K ? voffB + c * L4 : P4_OOB;
      bw[e] = p4_load<4>(r1, vo, 0);
      if constexpr (MODE == 2) bw2[e] = p4_load<4>(r2, vo, 0);
    }
  };
  // A fragments: (term t, row tile m of this wave, k-step ks) = 1 KB at ((t*RT + rt)*KS + ks) KB of the image
  const int KS = Kpad >> 4;
  const __amdgpu_buffer_rsrc_t rw = p4_rsrc(wfr, 3 * RT * KS * 1024);
  // (the fragment's KB index is wave-uniform: it rides in the scalar offset, one vector register addresses all of them)
  const int fragA0 = __builtin_amdgcn_readfirstlane(mBase >> 5);
  u32x4v af0[MT][3], af1[MT][3];                   // k-steps of even / odd index
  auto issueA = [&](int ks, u32x4v (&af)[MT][3]) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        af[m][t] = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, ((t * RT + fragA0 + m) * KS + ks) * 1024, 0));
  };
  const float lo = a.relu ? 0.f : -__builtin_inff();
  // the chunk in registers -> the operand values, in place (deferred-BatchNorm affine, second stream, ReLU): done as soon
  // as the loads land, so the affine table's rows and the second stream's registers are dead before the products start
  // (with them alive next to 128 accumulator + 48 fragment registers the two-stream kernels spilled INSIDE the loop, and a
  // scratch reload waits, in order, for every prefetch issued before it)
  auto combine = [&](int ch0) {
    if constexpr (MODE != 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 pr = Ps[ch0 + 4 * cg + e];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float x = fmaf(bw[e][q], pr.x, pr.y);
          if constexpr (MODE == 2) x += fmaf(bw2[e][q], pr.z, pr.w);
          bw[e][q] = fmaxf(x, lo);
        }
      }
    }
  };
  // position sub-tile q of the (combined) chunk -> its three term rows of buffer `dst`
  auto commit_q = [&](char* dst, int q) {
    char* base = dst + (32 * q + l31) * PG_RB + cg * 8;
    unsigned p0, p1, p2, q0, q1, q2;
    b3_split(bw[0][q], bw[1][q], p0, p1, p2);
    b3_split(bw[2][q], bw[3][q], q0, q1, q2);
    *reinterpret_cast<u32x2v*>(base) = u32x2v{p0, q0};
    *reinterpret_cast<u32x2v*>(base + PG_T * PG_RB) = u32x2v{p1, q1};
    *reinterpret_cast<u32x2v*>(base + 2 * PG_T * PG_RB) = u32x2v{p2, q2};
  };

  // prologue: the first TWO activation chunks are requested before anything else (the second into a register set that is
  // dead once the loop starts), so the affine table, the bias and chunk 0's split all wait on one memory round trip
  vq bwN[4], bw2N[MODE == 2 ? 4 : 1];
  __builtin_amdgcn_sched_barrier(0);
  issueB(0);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = PG_KC + 4 * cg + e;
    const int vo = c < K ? voffB + c * L4 : P4_OOB;
    bwN[e] = p4_load<4>(r1, vo, 0);
    if constexpr (MODE == 2) bw2N[e] = p4_load<4>(r2, vo, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x16 acc[MT][4];
  const __amdgpu_buffer_rsrc_t rbias = p4_rsrc(a.bias, (EPI == 0 && a.bias) ? M * 4 : 0);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = mBase + 32 * m + p4_row32(i, half);
      float b0 = 0.f;                              // (rows past M / no bias: the bounds check returns zero, no branches)
      if constexpr (EPI == 0) b0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, row * 4, 0, 0));
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[m][q][i] = b0;
    }
  if (MODE != 0) {
    for (int i = tid; i < Kpad; i += 256) {
      f32x4 p = {0.f, 0.f, 0.f, 0.f};
      if (i < K) {
        p.x = a.ps1 ? a.ps1[i] : 1.f;
        p.y = a.ph1 ? a.ph1[i] : 0.f;
        p.z = a.ps2 ? a.ps2[i] : 1.f;
        p.w = a.ph2 ? a.ph2[i] : 0.f;
      }
      Ps[i] = p;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  issueA(0, af0);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                    // Ps visible (raw barrier: the operand loads stay in flight)
  PWG_STAMP();
  combine(0);
#pragma unroll
  for (int q = 0; q < 4; ++q) commit_q(Bb, q);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bw[e] = bwN[e];
    if constexpr (MODE == 2) bw2[e] = bw2N[e];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  PWG_STAMP();

  const int fragoff = l31 * PG_RB + 16 * half;
  // one k-step of products on buffer `cur` (ksl = 0 / 1: the chunk's first / second 16 channels); `fill(q)` runs after
  // the products of position sub-tile q (the next chunk's split: VALU + LDS stores in the MFMA shadow)
  auto products = [&](const char* cur, int ksl, u32x4v (&af)[MT][3], auto&& fill) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bf16x8 bf[3];
#pragma unroll
      for (int t = 0; t < 3; ++t)
        bf[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(cur + fragoff + (t * PG_T + 32 * q) * PG_RB + 32 * ksl));
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, af[m][0]), a1 = __builtin_bit_cast(bf16x8, af[m][1]),
                     a2 = __builtin_bit_cast(bf16x8, af[m][2]);
        f32x16 c = acc[m][q];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf[0], c, 0, 0, 0);
        acc[m][q] = c;
      }
      fill(q);
    }
  };
  const int NC = Kpad / PG_KC;
  // ACT = false: a wave whose rows lie past M (M = 96: the fourth wave) only loads and splits
  auto mainloop = [&](auto act) {
    constexpr bool ACT = decltype(act)::value;
    char* cur = Bb;
    char* nxt = Bb + G3_BBUF;
    for (int i = 0; i < NC; ++i) {
      const int ch1 = (i + 1) * PG_KC;             // the chunk committed in this pass (past K: zeros, never multiplied)
      if constexpr (ACT) issueA(2 * i + 1, af1);
      __builtin_amdgcn_sched_barrier(0);
      combine(ch1 < Kpad ? ch1 : 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ACT) products(cur, 0, af0, [&](int q) { commit_q(nxt, q); });
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q) commit_q(nxt, q);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ACT) issueA(2 * i + 2, af0);
      issueB(ch1 + PG_KC);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ACT) products(cur, 1, af1, [&](int) {});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PWG_STAMP();
      __builtin_amdgcn_s_barrier();                // raw barrier: the loads in flight stay in flight
      PWG_STAMP();
      char* t = cur; cur = nxt; nxt = t;
    }
  };
  if (mBase < M) mainloop(std::true_type{});
  else mainloop(std::false_type{});
  __syncthreads();                                 // drains the read-ahead loads before LDS is reused
  PWG_STAMP();
  const P4Tile tile = {wave, half, l31, tid, mBase, n, nrem, ds, pos, grp, true, pok, skip};
  if constexpr (EPI == 1) {
    // operand prefetch depth of the epilogue: what the registers freed by the main loop hold next to the accumulators
    if (a.ex2) p4_epilogue<MT, 4, EPI, true, 4, MT == 2 ? 1 : 4, true>(a, acc, lds, tile);
    else p4_epilogue<MT, 4, EPI, true, 4, MT == 2 ? 3 : 4, false>(a, acc, lds, tile);
  } else {
    p4_epilogue<MT, 4, EPI, true, 4>(a, acc, lds, tile);
  }
  PWG_STAMP();
#ifdef DSGCN_LAB
  if (blockIdx.x == PWG_STAMP_BLOCK && threadIdx.x == 0) g_pwg_stamp[63] = nst;
#endif
}

// The three bf16 terms of W (Co x Ci) as two images, k contiguous and zero-padded to whole tiles:
//   N (forward):        rows co < MpN = ceil256(Co), k = ci < KpN = ceil32(Ci)      [3][MpN][KpN] bf16
//   T (data gradient):  rows ci < MpT = ceil256(Ci), k = co < KpT = ceil32(Co)      [3][MpT][KpT] bf16, after N
// One thread per (row, 4 consecutive k).
// frag = 1 (k_pwg3): each plane in MFMA-fragment order instead, [row tile][k-step][lane = 32*(k%16 / 8) + row%32][k%8].
__global__ __launch_bounds__(256) void k_wsplit(const float* __restrict__ w, int Ci, int Co, unsigned short* __restrict__ out,
                                                int MpN, int KpN, int MpT, int KpT, int frag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int t1 = MpN * (KpN >> 2), t2 = MpT * (KpT >> 2);
  float v[4];
  unsigned short* dst;
  int pstride;                                     // elements between the term planes
  if (i < t1) {
    const int r = i / (KpN >> 2), k = 4 * (i - r * (KpN >> 2));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (r < Co && k + e < Ci) ? w[(size_t)r * Ci + k + e] : 0.f;
    dst = out + (frag ? ((size_t)(r >> 5) * (KpN >> 4) + (k >> 4)) * 512 + (((k >> 3) & 1) * 32 + (r & 31)) * 8 + (k & 7)
                      : (size_t)r * KpN + k);
    pstride = MpN * KpN;
  } else if (i - t1 < t2) {
    const int j = i - t1;
    const int r = j / (KpT >> 2), k = 4 * (j - r * (KpT >> 2));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (r < Ci && k + e < Co) ? w[(size_t)(k + e) * Ci + r] : 0.f;
    dst = out + (size_t)3 * MpN * KpN +
          (frag ? ((size_t)(r >> 5) * (KpT >> 4) + (k >> 4)) * 512 + (((k >> 3) & 1) * 32 + (r & 31)) * 8 + (k & 7)
                : (size_t)r * KpT + k);
    pstride = MpT * KpT;
  } else {
    return;
  }
  unsigned p0, p1, p2, q0, q1, q2;
  b3_split(v[0], v[1], p0, p1, p2);
  b3_split(v[2], v[3], q0, q1, q2);
  *reinterpret_cast<u32x2v*>(dst) = u32x2v{p0, q0};
  *reinterpret_cast<u32x2v*>(dst + pstride) = u32x2v{p1, q1};
  *reinterpret_cast<u32x2v*>(dst + 2 * (size_t)pstride) = u32x2v{p2, q2};
}

// The same for up to WS_MAXJOBS convs in ONE launch (the 22 wide convs of a DS-STGCN step: 22 dependent 4 us launches at
// the head of their convs -> one at the head of the step).  The job table rides in the kernel arguments.
constexpr int WS_MAXJOBS = 32;
struct WsJobs {
  const float* w[WS_MAXJOBS];
  unsigned short* out[WS_MAXJOBS];
  int Ci[WS_MAXJOBS], Co[WS_MAXJOBS];
  int blk0[WS_MAXJOBS + 1];                         // first block of job j; blk0[njobs] = grid
  int njobs, frag;
};

__global__ __launch_bounds__(256) void k_wsplit_multi(WsJobs jb) {
  int j = 0;
  while (j + 1 < jb.njobs && (int)blockIdx.x >= jb.blk0[j + 1]) ++j;
  const int Ci = jb.Ci[j], Co = jb.Co[j];
  const float* __restrict__ w = jb.w[j];
  unsigned short* __restrict__ out = jb.out[j];
  const int MpN = (Co + 255) / 256 * 256, KpN = (Ci + 31) / 32 * 32, MpT = (Ci + 255) / 256 * 256, KpT = (Co + 31) / 32 * 32;
  const int i = ((int)blockIdx.x - jb.blk0[j]) * 256 + threadIdx.x;
  const int t1 = MpN * (KpN >> 2), t2 = MpT * (KpT >> 2);
  float v[4];
  unsigned short* dst;
  int pstride;
  if (i < t1) {
    const int r = i / (KpN >> 2), k = 4 * (i - r * (KpN >> 2));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (r < Co && k + e < Ci) ? w[(size_t)r * Ci + k + e] : 0.f;
    dst = out + (jb.frag ? ((size_t)(r >> 5) * (KpN >> 4) + (k >> 4)) * 512 + (((k >> 3) & 1) * 32 + (r & 31)) * 8 + (k & 7)
                         : (size_t)r * KpN + k);
    pstride = MpN * KpN;
  } else if (i - t1 < t2) {
    const int q = i - t1;
    const int r = q / (KpT >> 2), k = 4 * (q - r * (KpT >> 2));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (r < Ci && k + e < Co) ? w[(size_t)(k + e) * Ci + r] : 0.f;
    dst = out + (size_t)3 * MpN * KpN +
          (jb.frag ? ((size_t)(r >> 5) * (KpT >> 4) + (k >> 4)) * 512 + (((k >> 3) & 1) * 32 + (r & 31)) * 8 + (k & 7)
                   : (size_t)r * KpT + k);
    pstride = MpT * KpT;
  } else {
    return;
  }
  unsigned p0, p1, p2, q0, q1, q2;
  b3_split(v[0], v[1], p0, p1, p2);
  b3_split(v[2], v[3], q0, q1, q2);
  *reinterpret_cast<u32x2v*>(dst) = u32x2v{p0, q0};
  *reinterpret_cast<u32x2v*>(dst + pstride) = u32x2v{p1, q1};
  *reinterpret_cast<u32x2v*>(dst + 2 * (size_t)pstride) = u32x2v{p2, q2};
}

struct WsDims { int MpN, KpN, MpT, KpT; size_t bytes; };
WsDims ws_dims(int Ci, int Co) {
  WsDims d;
  d.MpN = (Co + 255) / 256 * 256; d.KpN = (Ci + 31) / 32 * 32;
  d.MpT = (Ci + 255) / 256 * 256; d.KpT = (Co + 31) / 32 * 32;
  d.bytes = ((size_t)3 * d.MpN * d.KpN + (size_t)3 * d.MpT * d.KpT) * 2;
  return d;
}

// K-split form of the tiny-plane launches (lab key 19).  OFF in the product: measured −0.02 ms/step (profiles/r06/
// small_launch_ab.txt) — and it changes the summation order of the projection convs, which moved the frozen whole-gradient
// ratios of the DS-STGCN configurations by 5-20 % (chaotic amplification, every per-kernel test unchanged): not worth a
// re-recording of the marks.
int g_p4_ksp = 0;
int g_p4_fill = 1;      // under-filled launches take one row tile per wave (lab key 20)
int g_p4_mt1 = 0;       // lab key 22: 128-row workgroups also for convs with more than 128 rows (bit 0 forward, bit 1 data gradient)
int g_p4_nq = 0, g_p4_mt = 0, g_p4_pd = 0, g_p4_gemm = 3, g_p4_gmin = 64, g_p4_gminl = 128, g_p4_ws = 2;   // ws: 2 = k_pwg3 on the fragment-order image; 1 = row-major image (lab A/B: no consumer left, k_pwg runs); g_p4_pd unused

struct P4Plan { int MT, NQ, PD, cc, span, WT, ngrp, Kpad, gemm, Lq, ksp; size_t lds; unsigned grid; };

bool p4_plan(int n, int K, int M, int L, P4Plan* p, int epi = 0) {
  const bool ragged = L % 2 != 0 && L >= 4;         // odd planes: runs of 4, the last one moved back to end at the plane's end
  if (L % 2 != 0 && !ragged) return false;
  int NQ = (L % 4 == 0 || ragged) ? 4 : 2;
  const int mtiles = (M + 31) / 32;
  int MT = mtiles >= 2 ? 2 : 1;
  // 33..48 and 65..96 output rows (the K*mid widths of the `pre` conv and of the `post` conv's data gradient): one row
  // tile per wave and one more workgroup per tile column — a 3-tile wave (NQ = 2) left the launch at 200 workgroups,
  // a 2-tile wave wastes a half-empty tile on every wave (tools/kc_bench.py 5=1: 256->96 forward 50 -> 39 us,
  // 96->256 data gradient 70 -> 62 us, 128->48 forward 29 -> 25 us, 48->128 data gradient 45 -> 36 us)
  if (mtiles == 3 || (mtiles == 2 && M <= 48)) MT = 1;
  // tiny planes (the dynamic-adjacency projections: 32 padded joints per sample): the launch is a latency chain of K/2
  // k-steps on few waves, so give every wave the smallest tile (1 x 2 MFMAs per k-step) and the grid the most waves
  const bool tiny = L <= 64 && (long)n * ((L + 127) / 128) < 1024 && !ragged;
  if (tiny) { NQ = 2; MT = 1; }
  // (round 6) a launch that would not even put one workgroup on every CU: one row tile per wave doubles the workgroups
  // (CTR-GCN's conv4 at 64 output channels: 157 position groups x 1 row block on 256 CUs)
  if (g_p4_fill && MT == 2 && ((((long)n * ((L + NQ - 1) / NQ * NQ) + 32 * NQ - 1) / (32 * NQ) + 3) / 4) * ((mtiles + 1) / 2) < 256) MT = 1;
  if ((g_p4_nq == 2 && !ragged) || (g_p4_nq == 4 && L % 4 == 0)) NQ = g_p4_nq;
  if (g_p4_mt) MT = g_p4_mt < mtiles ? g_p4_mt : mtiles;
  if (MT * NQ > 8 && !ragged) NQ = 2;
  if (MT * NQ > 8) MT = ragged ? 2 : 4;
  const int PD = NQ == 4 ? 8 : 16;
  // samples a wave's 32*NQ-position tile can touch; its buffer resources span that many planes (32-bit offsets)
  const int Lq = (L + NQ - 1) / NQ * NQ;
  const int span = (32 * NQ + Lq - 1) / Lq + 1;
  if ((long)K * L * 4 * span >= (1L << 31) - 64 || (long)M * L * 4 * span >= (1L << 31) - 64) return false;
  if ((long)n * L >= (1L << 31) - 256) return false;
  p->MT = MT; p->NQ = NQ; p->PD = PD; p->Lq = Lq;
  p->cc = (mtiles + MT - 1) / MT;
  p->span = span;
  p->WT = (int)(((long)n * Lq + 32 * NQ - 1) / (32 * NQ));
  p->ngrp = (p->WT + 3) / 4;
  p->Kpad = (K + 2 * PD - 1) / (2 * PD) * (2 * PD);
  size_t f = (size_t)((32 * MT * (p->Kpad + 1) + 2 + 3) & ~3) + (size_t)4 * (p->Kpad + 2);   // + the pipeline's read-ahead pad
  const size_t fe = (size_t)4 * 3 * 32 * 36 + (size_t)4 * MT * 32 * 4 + (size_t)MT * 32 * 4;    // epilogue image
  if (f < fe) f = fe;
  p->lds = f * sizeof(float);
  // the K-split form of the tiny-plane launches (k_pw4<.., KSP>): at least one prefetch round per wave; the launcher takes
  // it for plain operands without statistics / input-affine sums (the projections) and then needs room for three waves'
  // accumulators
  p->ksp = (g_p4_ksp && tiny && MT == 1 && NQ == 2 && p->Kpad / (2 * PD) >= 4) ? 1 : 0;
  if (p->ksp && p->lds < (size_t)3 * MT * NQ * 16 * 64 * sizeof(float)) p->lds = (size_t)3 * MT * NQ * 16 * 64 * sizeof(float);
  p->gemm = 0;
  if ((g_p4_gemm & (1 << epi)) && (L % 4 == 0 || ragged) && K % 4 == 0 && M % 4 == 0 && K >= g_p4_gmin && M > 64 &&
      L >= g_p4_gminl) {
    p->gemm = 1; p->MT = 1; p->NQ = 4;
    p->cc = (M + PG_T - 1) / PG_T;
    p->span = (128 + Lq - 1) / Lq + 1;             // (the tiny-plane rule above may have chosen 64-position tiles)
    p->WT = (int)(((long)n * Lq + 127) / 128);
    p->ngrp = p->WT;                               // one position tile per workgroup
    p->Kpad = (K + PG_KC - 1) / PG_KC * PG_KC;
    size_t b = (size_t)3 * ((epi ? PG_TAM : PG_T) + PG_T) * PG_RB + (size_t)p->Kpad * 16;
    const size_t be = ((size_t)4 * 3 * 32 * 36 + 4 * 32 * 4 + 4 * 32 * 4) * sizeof(float);   // epilogue image, per-wave rows
    p->lds = b < be ? be : b;
  }
  if (p->lds > 160 * 1024) return false;
  p->grid = (unsigned)((p->ngrp + 7) / 8 * 8 * p->cc);
  return true;
}

template <int MT, int NQ, int PD>
void p4_launch_cfg(const Pw4Args& a, int mode, int epi, const P4Plan& p, hipStream_t st) {
  const dim3 grid(p.grid, a.ngroup > 1 ? (unsigned)a.ngroup : 1u), blk(P4_NT);
  if (epi == 0) {
    if (mode == 0) hipLaunchKernelGGL((k_pw4<MT, NQ, 0, PD, 0>), grid, blk, p.lds, st, a);
    else if (mode == 1) hipLaunchKernelGGL((k_pw4<MT, NQ, 1, PD, 0>), grid, blk, p.lds, st, a);
    else hipLaunchKernelGGL((k_pw4<MT, NQ, 2, PD, 0>), grid, blk, p.lds, st, a);
  } else {
    if (mode == 0) hipLaunchKernelGGL((k_pw4<MT, NQ, 0, PD, 1>), grid, blk, p.lds, st, a);
    else hipLaunchKernelGGL((k_pw4<MT, NQ, 2, PD, 1>), grid, blk, p.lds, st, a);
  }
}

// the fragment-image form: MT = 2 (256 rows per workgroup) when the conv has more than 128 output rows
template <int MT>
void pwg3_launch(Pw4Args a, int mode, int epi, const P4Plan& p, const unsigned short* wfr, int Mp, hipStream_t st) {
  constexpr int TA = 128 * MT;
  a.cc = (a.M + TA - 1) / TA;
  const size_t main_b = (size_t)2 * G3_BBUF + (size_t)p.Kpad * 16;
  const size_t epi_b = (size_t)4 * (3 * 32 * 36 + MT * 32 * 3 + MT * 32 * 4) * sizeof(float);
  const size_t lds = main_b > epi_b ? main_b : epi_b;
  static bool raised = false;
  if (!raised) {
    const void* fs[5] = {reinterpret_cast<const void*>(&k_pwg3<0, 0, MT>), reinterpret_cast<const void*>(&k_pwg3<1, 0, MT>),
                         reinterpret_cast<const void*>(&k_pwg3<2, 0, MT>), reinterpret_cast<const void*>(&k_pwg3<0, 1, MT>),
                         reinterpret_cast<const void*>(&k_pwg3<2, 1, MT>)};
    for (const void* f : fs) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    raised = true;
  }
  const dim3 grid((unsigned)((p.ngrp + 7) / 8 * 8 * a.cc)), blk(256);
  const int RT = Mp / 32;
  if (epi == 0) {
    if (mode == 0) hipLaunchKernelGGL((k_pwg3<0, 0, MT>), grid, blk, lds, st, a, wfr, RT);
    else if (mode == 1) hipLaunchKernelGGL((k_pwg3<1, 0, MT>), grid, blk, lds, st, a, wfr, RT);
    else hipLaunchKernelGGL((k_pwg3<2, 0, MT>), grid, blk, lds, st, a, wfr, RT);
  } else {
    if (mode == 0) hipLaunchKernelGGL((k_pwg3<0, 1, MT>), grid, blk, lds, st, a, wfr, RT);
    else hipLaunchKernelGGL((k_pwg3<2, 1, MT>), grid, blk, lds, st, a, wfr, RT);
  }
}

template <int PD>
bool p4_launch_pd(const Pw4Args& a, int mode, int epi, const P4Plan& p, hipStream_t st, const unsigned short* wsp = nullptr,
                  int Mp = 0) {
  const int key = p.MT * 10 + p.NQ;
  if (p.gemm && wsp && g_p4_ws == 2) {
    if (a.M > 128 && !(g_p4_mt1 & (1 << epi))) pwg3_launch<2>(a, mode, epi, p, wsp, Mp, st);
    else pwg3_launch<1>(a, mode, epi, p, wsp, Mp, st);
    return true;
  }
  if (p.gemm) {
    static bool raised = false;                    // 64 KB+ of dynamic LDS
    if (!raised) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pwg<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pwg<1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pwg<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pwg<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pwg<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      raised = true;
    }
    const dim3 grid(p.grid), blk(P4_NT);
    if (epi == 0) {
      if (mode == 0) hipLaunchKernelGGL((k_pwg<0, 0>), grid, blk, p.lds, st, a);
      else if (mode == 1) hipLaunchKernelGGL((k_pwg<1, 0>), grid, blk, p.lds, st, a);
      else hipLaunchKernelGGL((k_pwg<2, 0>), grid, blk, p.lds, st, a);
    } else {
      if (mode == 0) hipLaunchKernelGGL((k_pwg<0, 1>), grid, blk, p.lds, st, a);
      else hipLaunchKernelGGL((k_pwg<2, 1>), grid, blk, p.lds, st, a);
    }
    return true;
  }
  // only the prefetch depth the plan chooses for each quad width is instantiated (4-position quads: 8 k-steps, 2-position
  // quads: 16): the other combinations were never launched and compiled to 140-VGPR-spill code objects
  if constexpr (PD == 8) {
    switch (key) {
      case 14: p4_launch_cfg<1, 4, PD>(a, mode, epi, p, st); return true;
      case 24: p4_launch_cfg<2, 4, PD>(a, mode, epi, p, st); return true;
    }
  } else {
    switch (key) {
      case 12: p4_launch_cfg<1, 2, PD>(a, mode, epi, p, st); return true;
      case 22: p4_launch_cfg<2, 2, PD>(a, mode, epi, p, st); return true;
      case 32: p4_launch_cfg<3, 2, PD>(a, mode, epi, p, st); return true;
      case 42: p4_launch_cfg<4, 2, PD>(a, mode, epi, p, st); return true;
    }
  }
  return false;
}

}  // namespace

// Internal (not part of the C ABI): called by dsgcn_pwconv_fwd / dsgcn_pwconv_dgrad when the shape qualifies
// (stride 1, even plane size).  Return 1 = launched, 0 = not eligible, <0 / >0 = error codes as everywhere.
__attribute__((visibility("hidden"))) int dsgcn_p4_tuning(int key, int value) {
  if (key == 0) g_p4_nq = value;
  else if (key == 1) g_p4_mt = value;
  else if (key == 2) g_p4_pd = value;
  else if (key == 3) g_p4_gemm = value;
  else if (key == 4) g_p4_gmin = value;
  else if (key == 5) g_p4_gminl = value;
  else if (key == 6) g_p4_ws = value;
  else if (key == 7) g_p4_ksp = value;
  else if (key == 8) g_p4_fill = value;
  else if (key == 9) g_p4_mt1 = value;
  else return DSGCN_EINVAL;
  return 0;
}

__attribute__((visibility("hidden"))) int dsgcn_p4_groups(int n, int K, int M, int L, int epi) {
  P4Plan p;
  return p4_plan(n, K, M, L, &p, epi) ? p.ngrp : 0;
}

__attribute__((visibility("hidden"))) int dsgcn_p4_fwd(const float* x1, const float* s1, const float* h1,
                                                        const float* x2, const float* s2, const float* h2, int relu,
                                                        const float* w, const float* bias, float* z, float* partial,
                                                        int n, int Ci, int Co, int L, hipStream_t st, const void* ws) {
  P4Plan p;
  if (!p4_plan(n, Ci, Co, L, &p)) return 0;
  const WsDims wd = ws_dims(Ci, Co);
  const unsigned short* wsp = static_cast<const unsigned short*>(ws);
  Pw4Args a = {};
  a.b1 = x1; a.b2 = x2; a.ps1 = s1; a.ph1 = h1; a.ps2 = s2; a.ph2 = h2; a.relu = relu;
  a.w = w; a.w_ldm = Ci; a.w_ldk = 1; a.bias = bias; a.out = z; a.partial = partial;
  a.n = n; a.K = Ci; a.M = Co; a.L = L; a.span = p.span; a.WT = p.WT; a.cc = p.cc; a.Kpad = p.Kpad; a.Lq = p.Lq;
  const int mode = x2 ? 2 : ((s1 || relu) ? 1 : 0);
  if (p.ksp && !p.gemm && mode == 0 && !partial) {
    hipLaunchKernelGGL((k_pw4<1, 2, 0, 16, 0, true>), dim3((unsigned)((p.WT + 7) / 8 * 8 * p.cc)), dim3(P4_NT), p.lds, st, a);
    DSGCN_LAUNCH_CHECK();
    return 1;
  }
  const bool ok = p.PD == 8 ? p4_launch_pd<8>(a, mode, 0, p, st, wsp, wd.MpN) : p4_launch_pd<16>(a, mode, 0, p, st, wsp, wd.MpN);
  if (!ok) return 0;
  DSGCN_LAUNCH_CHECK();
  return 1;
}

// dz_eff = gz + A0 + B0*z (either part may be absent);  x1/x2/s*/h*/relu describe the forward's virtual input.
__attribute__((visibility("hidden"))) int dsgcn_p4_dgrad(const float* x1, const float* s1, const float* h1,
                                                          const float* x2, const float* s2, const float* h2, int relu,
                                                          const float* w, const float* z, const float* gz,
                                                          const float* A0, const float* B0, float* dx1, float* dx2,
                                                          float* ipart, int n, int Ci, int Co, int L, hipStream_t st,
                                                          const void* ws) {
  P4Plan p;
  if (!gz) return 0;                               // (a conv whose output has no direct gradient: not on the fast path)
  if (!p4_plan(n, Co, Ci, L, &p, 1)) return 0;
  Pw4Args a = {};
  a.b1 = gz; a.b2 = A0 ? z : nullptr;
  a.ps1 = nullptr; a.ph1 = A0; a.ps2 = B0; a.ph2 = nullptr; a.relu = 0;
  a.w = w; a.w_ldm = 1; a.w_ldk = Ci; a.bias = nullptr; a.out = dx1; a.partial = nullptr;
  a.ex1 = x1; a.ex2 = x2; a.es1 = s1; a.eh1 = h1; a.es2 = s2; a.eh2 = h2; a.erelu = relu;
  a.out2 = dx2; a.ipart = ipart;
  a.n = n; a.K = Co; a.M = Ci; a.L = L; a.span = p.span; a.WT = p.WT; a.cc = p.cc; a.Kpad = p.Kpad; a.Lq = p.Lq;
  const int mode = A0 ? 2 : 0;
  if (p.ksp && !p.gemm && mode == 0 && !ipart) {
    hipLaunchKernelGGL((k_pw4<1, 2, 0, 16, 1, true>), dim3((unsigned)((p.WT + 7) / 8 * 8 * p.cc)), dim3(P4_NT), p.lds, st, a);
    DSGCN_LAUNCH_CHECK();
    return 1;
  }
  const WsDims wd = ws_dims(Ci, Co);
  const unsigned short* wsp = ws ? static_cast<const unsigned short*>(ws) + (size_t)3 * wd.MpN * wd.KpN : nullptr;   // the T image
  const bool ok = p.PD == 8 ? p4_launch_pd<8>(a, mode, 1, p, st, wsp, wd.MpT) : p4_launch_pd<16>(a, mode, 1, p, st, wsp, wd.MpT);
  if (!ok) return 0;
  DSGCN_LAUNCH_CHECK();
  return 1;
}

// Up to three convs of one shape in ONE launch (k_pw4 only: plain / affine operand, no second stream, no statistics).
// Return 1 = launched, 0 = not eligible (the caller launches them one by one).
__attribute__((visibility("hidden"))) int dsgcn_p4_group_ok(int n, int Ci, int Co, int L) {
  P4Plan pf, pb;
  return (p4_plan(n, Ci, Co, L, &pf, 0) && !pf.gemm && p4_plan(n, Co, Ci, L, &pb, 1) && !pb.gemm) ? 1 : 0;
}

__attribute__((visibility("hidden"))) int dsgcn_p4_fwd_group(const float* const* x1, const float* const* s1,
                                                              const float* const* h1, int relu, const float* const* w,
                                                              float* const* z, int ng, int n, int Ci, int Co, int L,
                                                              hipStream_t st) {
  P4Plan p;
  if (ng < 1 || ng > 3 || !p4_plan(n, Ci, Co, L, &p) || p.gemm) return 0;
  Pw4Args a = {};
  a.b1 = x1[0]; a.ps1 = s1[0]; a.ph1 = h1[0]; a.relu = relu;
  a.w = w[0]; a.w_ldm = Ci; a.w_ldk = 1; a.out = z[0];
  a.n = n; a.K = Ci; a.M = Co; a.L = L; a.span = p.span; a.WT = p.WT; a.cc = p.cc; a.Kpad = p.Kpad; a.Lq = p.Lq;
  a.ngroup = ng;
  for (int g = 0; g < ng; ++g) {
    if ((s1[g] == nullptr) != (s1[0] == nullptr)) return DSGCN_EINVAL;
    a.g[g] = Pw4Args::Grp{x1[g], s1[g], h1[g], w[g], z[g], nullptr, nullptr, nullptr, nullptr};
  }
  const int mode = (s1[0] || relu) ? 1 : 0;
  const bool ok = p.PD == 8 ? p4_launch_pd<8>(a, mode, 0, p, st) : p4_launch_pd<16>(a, mode, 0, p, st);
  if (!ok) return 0;
  DSGCN_LAUNCH_CHECK();
  return 1;
}

__attribute__((visibility("hidden"))) int dsgcn_p4_dgrad_group(const float* const* x1, const float* const* s1,
                                                                const float* const* h1, int relu, const float* const* w,
                                                                const float* const* gz, float* const* dx1, float* const* ipart,
                                                                int ng, int n, int Ci, int Co, int L, hipStream_t st) {
  P4Plan p;
  if (ng < 1 || ng > 3 || !p4_plan(n, Co, Ci, L, &p, 1) || p.gemm) return 0;
  Pw4Args a = {};
  a.b1 = gz[0]; a.relu = 0;
  a.w = w[0]; a.w_ldm = 1; a.w_ldk = Ci; a.out = dx1[0];
  a.ex1 = x1[0]; a.es1 = s1[0]; a.eh1 = h1[0]; a.erelu = relu; a.ipart = ipart[0];
  a.n = n; a.K = Co; a.M = Ci; a.L = L; a.span = p.span; a.WT = p.WT; a.cc = p.cc; a.Kpad = p.Kpad; a.Lq = p.Lq;
  a.ngroup = ng;
  for (int g = 0; g < ng; ++g) {
    if ((s1[g] == nullptr) != (s1[0] == nullptr) || (ipart[g] == nullptr) != (ipart[0] == nullptr)) return DSGCN_EINVAL;
    a.g[g] = Pw4Args::Grp{gz[g], nullptr, nullptr, w[g], dx1[g], x1[g], s1[g], h1[g], ipart[g]};
  }
  const bool ok = p.PD == 8 ? p4_launch_pd<8>(a, 0, 1, p, st) : p4_launch_pd<16>(a, 0, 1, p, st);
  if (!ok) return 0;
  DSGCN_LAUNCH_CHECK();
  return 1;
}

// Bytes of the pre-split weight image a (Ci -> Co) conv over planes of L positions wants (0: neither its forward nor its
// data gradient takes the GEMM form), and the launch that fills it.
__attribute__((visibility("hidden"))) size_t dsgcn_p4_ws_bytes(int n, int Ci, int Co, int L) {
  P4Plan pf, pb;
  const bool f = p4_plan(n, Ci, Co, L, &pf, 0) && pf.gemm, b = p4_plan(n, Co, Ci, L, &pb, 1) && pb.gemm;
  if (!g_p4_ws || !(f || b)) return 0;
  return ws_dims(Ci, Co).bytes;
}

__attribute__((visibility("hidden"))) int dsgcn_p4_wsplit(const float* w, int Ci, int Co, void* out, hipStream_t st) {
  const WsDims d = ws_dims(Ci, Co);
  const long total = (long)d.MpN * (d.KpN >> 2) + (long)d.MpT * (d.KpT >> 2);
  hipLaunchKernelGGL(k_wsplit, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, Ci, Co,
                     static_cast<unsigned short*>(out), d.MpN, d.KpN, d.MpT, d.KpT, g_p4_ws == 2 ? 1 : 0);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

__attribute__((visibility("hidden"))) int dsgcn_p4_wsplit_multi(const float* const* w, void* const* out, const int* Ci,
                                                                const int* Co, int njobs, hipStream_t st) {
  for (int j0 = 0; j0 < njobs; j0 += WS_MAXJOBS) {
    WsJobs jb;
    jb.njobs = njobs - j0 < WS_MAXJOBS ? njobs - j0 : WS_MAXJOBS;
    jb.frag = g_p4_ws == 2 ? 1 : 0;
    int blk = 0;
    for (int j = 0; j < jb.njobs; ++j) {
      const WsDims d = ws_dims(Ci[j0 + j], Co[j0 + j]);
      const long total = (long)d.MpN * (d.KpN >> 2) + (long)d.MpT * (d.KpT >> 2);
      jb.w[j] = w[j0 + j]; jb.out[j] = static_cast<unsigned short*>(out[j0 + j]);
      jb.Ci[j] = Ci[j0 + j]; jb.Co[j] = Co[j0 + j];
      jb.blk0[j] = blk;
      blk += (int)((total + 255) / 256);
    }
    for (int j = jb.njobs; j <= WS_MAXJOBS; ++j) jb.blk0[j] = blk;
    hipLaunchKernelGGL(k_wsplit_multi, dim3((unsigned)blk), dim3(256), 0, st, jb);
    DSGCN_LAUNCH_CHECK();
  }
  return 0;
}

#ifdef DSGCN_LAB
extern "C" int dsgcn_pwg2_phases(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pwg_stamp), sizeof(long long) * 64);
}
extern "C" int dsgcn_pwg2_phases_block(int block) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pwg_stamp_block), &block, sizeof(int));
}
#endif
