// Dense temporal conv ((KT,1) kernel, stride 1, dilation 1, "same" zero padding) as a GEMM on three-term bf16 products.
//
// Replaces `unit_tcn`'s Conv2d((9,1)) + BatchNorm2d statistics (pyskl/models/gcns/utils/tcn.py:21-28, used by ST-GCN
// stgcn.py:40-46 and AAGCN aagcn.py) with the consumer-side BN / ReLU / residual of the spatial unit in front of it
// applied while loading, and the autograd of all of that:
//   forward        z[n,co,t,v]  = b[co] + sum_{tap,ci} W[co,ci,tap] * x'[n,ci,t+tap-pad,v]      x' = relu?(x1*s1+h1 (+ x2*s2+h2))
//   data gradient  dx'[n,ci,t,v] = sum_{tap,co} W[co,ci,tap] * dz[n,co,t-(tap-pad),v]            dz = gz + A0[co] + B0[co]*z
//   weight grad.   dW[co,ci,tap] = sum_{n,t,v} dz[n,co,t,v] * x'[n,ci,t+tap-pad,v],   db[co] = sum dz
// The first-generation kernels (tapconv.hip, k_tapconv_*<9>) ran these at ~50 TF of fp32 MFMA and were 66 % of ST-GCN's
// step (profiles/r03/stgcn_kernel_stats.csv); they also needed the input materialised and the statistics in a pass of
// their own.
//
// k_tcg (forward and data gradient, one template): a workgroup (8 waves) owns 128 output rows x one tile of R = 128/V whole
// frames of ONE sample (R*V <= 128 position slots in natural order).  Per 32-channel chunk the input tile WITH its halo
// ((R + KT-1) frames) is loaded, activated, zeroed outside the sample's frames, split into its three bf16 terms ONCE and
// kept in LDS as [position row][32 channels]; a tap is then nothing but a row offset (tap*V rows) of the B fragment
// address, so the nine taps' products reuse one staging — the split costs ~1/9 of what it costs the 1x1 GEMM form per
// MFMA.  The weights arrive pre-split per tap (k_tsplit: one launch per conv and step, the W image for the forward and
// the tap-flipped W^T image for the data gradient) and are copied into a double-buffered LDS tile one tap ahead: one
// barrier per (chunk, tap) step, 24 MFMAs (v_mfma_f32_32x32x16_bf16) per wave and step.  Waves: 4 row tiles x 2 halves of
// the position slots.  Results leave with 4-byte stores (a lane's slots are 32 apart in natural order; 32 lanes of one
// store are 128 contiguous bytes).  Statistics / input-affine sums per workgroup tile -> partial rows (ordered sums, no
// atomics).
//
// k_tcw (weight gradient): workgroup = 128 co x 64 ci x all taps, wave = 32 x 32 x KT accumulators; K = positions in
// chunks of 32 of one sample; dz is staged once per chunk, x' once per tap with the tap's frame shift folded into the
// load address (elements outside the sample's frames are zero).  K is split over workgroups; every split writes its
// partial row (dsgcn_colsum / the deferred parameter sums reduce them in order).
#include "common.h"
#include "dsgcn_jobs.h"

namespace {

constexpr int TG_NT = 256;                       // 4 waves (k_tcg); the weight gradient brings 8
constexpr int TG_KC = 32;                        // channels per chunk
constexpr int TG_RB = TG_KC * 2 + 16;            // LDS row: 32 bf16 + 16 B pad (conflict-free 16-byte fragment reads)
constexpr int TG_OOB = 0x7ffffff0;
constexpr int TG_MAXIT = 12;                     // staging items per thread and chunk (rows <= 384, 8 channel groups each)
constexpr int TG_PASS = 6;                       // staging items loaded together
constexpr int TW_NT = 512;

struct TcgArgs {
  const float* b1; const float* b2;                                        // B streams (n, K, T, V); b2 NULL unless MODE 2
  const float* ps1; const float* ph1; const float* ps2; const float* ph2;  // per-k affine (NULL = 1 / 0)
  int relu;
  const unsigned short* wsp;                                               // [KT][3][Mp x Kp in fragment order] bf16, rows = output rows
  const float* bias;                                                       // per output row, NULL ok (forward)
  float* out;                                                              // (n, M, T, V)
  float* partial;                                                          // EPI 0: [tiles][M][2] or NULL
  const float* ex1; const float* ex2;                                      // EPI 1: the forward's operands at (n, M, T, V)
  const float* es1; const float* eh1; const float* es2; const float* eh2;
  int erelu;
  float* out2; float* ipart;                                               // EPI 1: d x2 or NULL; [tiles][M][3] or NULL
  int n, K, M, V, KT, R, tps, cc, Mp, Kp;                                  // R output frames per tile, tps tiles per sample
  int Ts, To, st, up;          // frames of the B source / of the output; forward stride (lane row stride); source upsampling (data gradient of a strided conv)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tg_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, bytes, 0x00020000);
}
__device__ __forceinline__ float tg_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ int tg_row32(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// sum over the 32 lanes of a half-wave (lanes l and l^32 keep their own)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// MODE 0: B' = b1;  1: relu?(b1*s1+h1);  2: relu?(b1*s1+h1 + b2*s2+h2).   EPI 0: forward (bias, statistics);  1: data gradient.
// WM: row tiles (of 32 output rows) per workgroup = 4 or 2; the four waves are WM row tiles x 4/WM groups of the position
// slots, a wave holds 32 rows x 32*WM slots.
// QS: stride-1 staging with 16-byte loads (launcher: st == 1, up == 1, source planes a multiple of 4 floats long)
template <int MODE, int EPI, int WM, bool QS = false>
__global__ __launch_bounds__(TG_NT, 2) void k_tcg(TcgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PW = 4 / WM, NQW = WM;             // position groups per workgroup, 32-slot tiles per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int rt = wave % WM, pg = wave / WM;        // row tile, position group (slots 32*NQW*pg ..)
  const int K = a.K, M = a.M, V = a.V, KT = a.KT, R = a.R, Ls = a.Ts * V, L = a.To * V;   // source / output plane lengths
  const int pad = (KT - 1) >> 1;
  const int NR = (a.st * (R - 1) + KT) * V;        // image rows (staged; fragment reads of dead slots are clamped into them)
  const int NRA = NR;
  char* Bb = reinterpret_cast<char*>(lds);                               // [3][NRA rows][RB]
  // workgroup -> (row tile column cz, position tile g): the cc workgroups that stage the same input tile are neighbours
  const int id = blockIdx.x, xcd = id & 7, slot_ = id >> 3;
  const int cz = slot_ % a.cc;
  const int g = (slot_ / a.cc) * 8 + xcd;
  if (g >= a.n * a.tps) return;
  const int ns = g / a.tps, ti = g - ns * a.tps;
  const int f0 = ti * R;                           // first output frame of the tile
  const int mBase0 = cz * 32 * WM;
  const float invV = 1.f / (float)V;
  const __amdgpu_buffer_rsrc_t r1 = tg_rsrc(a.b1 + (size_t)ns * K * Ls, K * Ls * 4);
  const __amdgpu_buffer_rsrc_t r2 = tg_rsrc((MODE == 2 ? a.b2 : a.b1) + (size_t)ns * K * Ls, MODE == 2 ? K * Ls * 4 : 0);
  // staging items: (image row j, channel group cg = 4 channels); item = tid + TG_NT*i
  const int nit = (NR * 8 + TG_NT - 1) / TG_NT;
  int soff[QS ? 1 : TG_MAXIT], sdst[QS ? 1 : TG_MAXIT];
#pragma unroll
  for (int i = 0; i < (QS ? 0 : TG_MAXIT); ++i) {
    const int item = tid + TG_NT * i, cg = item & 7, j = item >> 3;
    int fr, v;
    divmod_small(j, V, invV, fr, v);
    const int fv = a.st * f0 - pad + fr;           // frame on the (upsampled) source grid
    const int f = a.up == 1 ? fv : fv / a.up;
    const bool ok = i < nit && j < NR && fv >= 0 && f * a.up == fv && f < a.Ts;
    soff[i] = ok ? (f * V + v + 4 * cg * Ls) * 4 : TG_OOB;              // + ch0*Ls*4 per chunk
    sdst[i] = (i < nit && j < NR) ? j * TG_RB + cg * 8 : -1;
  }
  // Staging of a chunk: loads, activation, split and LDS stores in passes of TG_PASS items (the raw values are not held
  // across the products: at two workgroups per CU the register file caps a wave at 256 VGPRs, and the partner
  // workgroup's products cover this one's load latency).
  const float lo = a.relu ? 0.f : -__builtin_inff();
  auto stageB = [&](int ch0) {
    if constexpr (QS) return;
    f32x4 pr[4];
    if (MODE != 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = ch0 + 4 * (tid & 7) + e;
        f32x4 p = {0.f, 0.f, 0.f, 0.f};
        if (c < K) {
          p.x = a.ps1 ? a.ps1[c] : 1.f;
          p.y = a.ph1 ? a.ph1[c] : 0.f;
          p.z = a.ps2 ? a.ps2[c] : 1.f;
          p.w = a.ph2 ? a.ph2[c] : 0.f;
        }
        pr[e] = p;
      }
    }
    // (two streams on the 128-row tile: four items per pass — with six the 256-register cap spilled 9-14 values)
    constexpr int PASS = (MODE == 2 && WM == 4) ? 4 : TG_PASS;
#pragma unroll
    for (int i0 = 0; i0 < TG_MAXIT; i0 += PASS) {
      if (i0 < nit) {
        float bw[PASS][4], bw2[MODE == 2 ? PASS : 1][4];
#pragma unroll
        for (int ii = 0; ii < PASS; ++ii) {
          const int i = i0 + ii;
          if (i < nit) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int c = ch0 + 4 * (tid & 7) + e;                    // (item & 7 == tid & 7: TG_NT is a multiple of 8)
              const int vo = (c < K && soff[i] != TG_OOB) ? soff[i] + e * Ls * 4 : TG_OOB;
              bw[ii][e] = tg_load(r1, vo, ch0 * Ls * 4);
              if constexpr (MODE == 2) bw2[ii][e] = tg_load(r2, vo, ch0 * Ls * 4);
            }
          }
        }
#pragma unroll
        for (int ii = 0; ii < PASS; ++ii) {
          const int i = i0 + ii;
          if (i < nit && sdst[i] >= 0) {
            float v[4];
            const bool ok = soff[i] != TG_OOB;      // rows outside the sample's frames: zero AFTER the activation
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = bw[ii][e];
              if (MODE != 0) {
                x = fmaf(x, pr[e].x, pr[e].y);
                if constexpr (MODE == 2) x += fmaf(bw2[ii][e], pr[e].z, pr[e].w);
                x = fmaxf(x, lo);
              }
              v[e] = ok ? x : 0.f;
            }
            unsigned p0, p1, p2, q0, q1, q2;
            b3_split(v[0], v[1], p0, p1, p2);
            b3_split(v[2], v[3], q0, q1, q2);
            char* base = Bb + sdst[i];
            *reinterpret_cast<u32x2v*>(base) = u32x2v{p0, q0};
            *reinterpret_cast<u32x2v*>(base + NRA * TG_RB) = u32x2v{p1, q1};
            *reinterpret_cast<u32x2v*>(base + 2 * NRA * TG_RB) = u32x2v{p2, q2};
          }
        }
      }
    }
  };
  // Stride-1 staging with 16-byte loads (round 4): the image rows are CONSECUTIVE source positions starting at s0 =
  // (f0 - pad)*V, so the tile is covered by the aligned quads [4*q0, 4*q0 + 4*nqd) of every channel's plane.  An item is
  // (quad, group of 4 channels): four 16-byte loads give the thread a 4 x 4 block (channel x position) whose columns are
  // whole 8-byte pieces of the image rows — 12 loads per thread and chunk where the scalar form above issues 48 (96 with
  // two streams), each wave instruction covering 512 contiguous bytes per channel instead of eight 32-byte pieces.  Plane
  // lengths are multiples of 4 here, so a quad is entirely inside or outside its plane (outside: zero AFTER the activation).
  const int s0 = (f0 - pad) * V;
  const int q0 = s0 >> 2;                          // floor (s0 may be negative)
  const int nqd = ((s0 + NR - 1) >> 2) - q0 + 1;
  constexpr int TG_MAXQ = 4, TG_QPASS = MODE == 2 ? 1 : 2;   // <= 98 quads x 8 channel groups = 784 items over 256 threads
  auto stageBq = [&](int ch0) {
    f32x4 pr[4];
    if (MODE != 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = ch0 + 4 * (tid & 7) + e;
        f32x4 p = {0.f, 0.f, 0.f, 0.f};
        if (c < K) {
          p.x = a.ps1 ? a.ps1[c] : 1.f;
          p.y = a.ph1 ? a.ph1[c] : 0.f;
          p.z = a.ps2 ? a.ps2[c] : 1.f;
          p.w = a.ph2 ? a.ph2[c] : 0.f;
        }
        pr[e] = p;
      }
    }
    const int nitq = (nqd * 8 + TG_NT - 1) / TG_NT;
#pragma unroll
    for (int i0 = 0; i0 < TG_MAXQ; i0 += TG_QPASS) {
      if (i0 < nitq) {
        f32x4 bw[TG_QPASS][4], bw2[MODE == 2 ? TG_QPASS : 1][4];
#pragma unroll
        for (int ii = 0; ii < TG_QPASS; ++ii) {
          const int item = tid + TG_NT * (i0 + ii), qi = item >> 3;
          const int p4 = 4 * (q0 + qi);
          const bool qok = (qi < nqd) & (p4 >= 0) & (p4 < Ls);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = ch0 + 4 * (tid & 7) + e;
            const int vo = ((c < K) & qok) ? ((4 * (tid & 7) + e) * Ls + p4) * 4 : TG_OOB;
            bw[ii][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, vo, ch0 * Ls * 4, 0));
            if constexpr (MODE == 2)
              bw2[ii][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, vo, ch0 * Ls * 4, 0));
          }
        }
#pragma unroll
        for (int ii = 0; ii < TG_QPASS; ++ii) {
          const int item = tid + TG_NT * (i0 + ii), qi = item >> 3;
          const int p4 = 4 * (q0 + qi);
          const bool qin = (p4 >= 0) & (p4 < Ls);                       // inside the sample's frames
          if (qi < nqd) {
#pragma unroll
            for (int pe = 0; pe < 4; ++pe) {
              const int j = p4 + pe - s0;                                // image row of this position
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float x = bw[ii][e][pe];
                if (MODE != 0) {
                  x = fmaf(x, pr[e].x, pr[e].y);
                  if constexpr (MODE == 2) x += fmaf(bw2[ii][e][pe], pr[e].z, pr[e].w);
                  x = fmaxf(x, lo);
                }
                v[e] = qin ? x : 0.f;
              }
              if (j >= 0 && j < NR) {
                unsigned p0, p1, p2, q0_, q1, q2;
                b3_split(v[0], v[1], p0, p1, p2);
                b3_split(v[2], v[3], q0_, q1, q2);
                char* base = Bb + j * TG_RB + (tid & 7) * 8;
                *reinterpret_cast<u32x2v*>(base) = u32x2v{p0, q0_};
                *reinterpret_cast<u32x2v*>(base + NRA * TG_RB) = u32x2v{p1, q1};
                *reinterpret_cast<u32x2v*>(base + 2 * NRA * TG_RB) = u32x2v{p2, q2};
              }
            }
          }
        }
      }
    }
  };
  // A fragments straight from the pre-split image (L2): lane (l31, half) of row tile rt wants, per term and k-step, the 8
  // consecutive k [16*ks + 8*half, +8) of row mBase0 + 32*rt + l31 — 16 contiguous bytes.  No LDS copy, no barrier per tap.
  // (round 5) The image is in MFMA-FRAGMENT order inside every (tap, term) plane — [row tile][k-step][lane = 32*(k%16 / 8) +
  // row%32][k%8] — so a wave's fragment is ONE contiguous 1 KB.  On the row-major image a fragment load touched 32 rows x 32 B:
  // the weight fragments were 10-11 GB of L2 requests per ST-GCN step (86-92 % hits, profiles/r05/l2_requests_stgcn.csv), with
  // one tap of products (0.3-0.6 us) to hide each of them behind.
  const unsigned short* arow = a.wsp + ((size_t)((mBase0 >> 5) + rt) * (a.Kp >> 4)) * 512 + (half * 32 + l31) * 8;
  const size_t tstride = (size_t)a.Mp * a.Kp;      // elements between the (tap, term) planes
  u32x4v ac[2][3], an[2][3];
  auto loadA = [&](int tap, int ch0, u32x4v (&af)[2][3]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        af[ks][t] = *reinterpret_cast<const u32x4v*>(arow + (size_t)(tap * 3 + t) * tstride + ((ch0 >> 4) + ks) * 512);
  };
  f32x16 acc[NQW];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = mBase0 + 32 * rt + tg_row32(i, half);
    const float b0 = (EPI == 0 && a.bias && row < M) ? a.bias[row] : 0.f;
#pragma unroll
    for (int q = 0; q < NQW; ++q) acc[q][i] = b0;
  }
  // weight fragments in flight: the next tap's, and for the 64-row workgroups (WM = 2: 24 MFMAs per tap — 0.32 us of
  // products, less than an L2 round trip under load) also the one after it (round 5; the 128-row kernels have no registers
  // left for a third set)
  constexpr bool PF2 = WM == 2;
  u32x4v an2[PF2 ? 2 : 1][3];
  loadA(0, 0, ac);
  if constexpr (PF2) {
    if (KT > 1) loadA(1, 0, an);
    else if (TG_KC < a.Kp) loadA(0, TG_KC, an);
  }
  // B fragment rows: slot s = (frame offset s/V, joint s%V) sits at image row st*(s - s%V) + s%V (+ tap*V); dead slots read row 0
  int bfo[NQW];
#pragma unroll
  for (int q = 0; q < NQW; ++q) {
    const int s_ = 32 * (NQW * pg + q) + l31;
    int fr, v;
    divmod_small(s_, V, invV, fr, v);
    bfo[q] = (s_ < R * V ? a.st * fr * V + v : 0) * TG_RB + 16 * half;
  }
  for (int ch0 = 0; ch0 < a.Kp; ch0 += TG_KC) {
    if (ch0 > 0) {                                 // every wave is done with the previous chunk's image
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if constexpr (QS) stageBq(ch0);
    else stageB(ch0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int tap = 0; tap < KT; ++tap) {
      if constexpr (PF2) {                         // the weights of the step after next
        const int t2 = tap + 2;
        const int ntap = t2 < KT ? t2 : t2 - KT, nch = t2 < KT ? ch0 : ch0 + TG_KC;      // (KT >= 2 here; KT = 1 wraps once too)
        if (nch < a.Kp) loadA(ntap < KT ? ntap : 0, nch, an2);
      } else {                                     // next step's weights
        const int ntap = tap + 1 < KT ? tap + 1 : 0, nch = tap + 1 < KT ? ch0 : ch0 + TG_KC;
        if (nch < a.Kp) loadA(ntap, nch, an);
      }
      const char* Bft = Bb + tap * V * TG_RB;
#pragma unroll
      for (int ks = 0; ks < TG_KC / 16; ++ks) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, ac[ks][0]), a1 = __builtin_bit_cast(bf16x8, ac[ks][1]),
                     a2 = __builtin_bit_cast(bf16x8, ac[ks][2]);
#pragma unroll
        for (int q = 0; q < NQW; ++q) {
          bf16x8 bf[3];
#pragma unroll
          for (int t = 0; t < 3; ++t)
            bf[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(Bft + bfo[q] + t * NRA * TG_RB + 32 * ks));
          f32x16 c = acc[q];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bf[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bf[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bf[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf[0], c, 0, 0, 0);
          acc[q] = c;
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          ac[ks][t] = an[ks][t];
          if constexpr (PF2) an[ks][t] = an2[ks][t];
        }
    }
  }
  __syncthreads();                                 // LDS is reused below

  // ---- epilogue: slot = 32*(NQW*pg + q) + l31 is position f0*V + slot of sample ns (valid below R*V and inside the sample)
  float* Ss = lds;                                 // [4 waves][32 rows][3]
  int opos[NQW];
  bool ovalid[NQW];
#pragma unroll
  for (int q = 0; q < NQW; ++q) {
    const int s_ = 32 * (NQW * pg + q) + l31;
    opos[q] = f0 * V + s_;
    ovalid[q] = s_ < R * V && opos[q] < L;
  }
  const size_t obase = (size_t)ns * M * L;
  if (EPI == 0) {
    const bool stats = a.partial != nullptr;
    const __amdgpu_buffer_rsrc_t ro = tg_rsrc(a.out + obase, M * L * 4);      // dead slots / rows: out-of-range offsets, no branches
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row32 = tg_row32(r, half), row = mBase0 + 32 * rt + row32;
      float s = 0.f, qq = 0.f;
#pragma unroll
      for (int q = 0; q < NQW; ++q) {
        const bool ok = ovalid[q] && row < M;
        const float v = ok ? acc[q][r] : 0.f;
        const int vo = (int)((unsigned)(ok ? opos[q] * 4 : TG_OOB) + (unsigned)(row * L * 4));
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, vo, 0, 0);
        s += v;
        qq = fmaf(v, v, qq);
      }
      if (stats) {
        s = half_sum(s);
        qq = half_sum(qq);
        if (l31 == 0) { Ss[(wave * 32 + row32) * 3 + 0] = s; Ss[(wave * 32 + row32) * 3 + 1] = qq; }
      }
    }
    if (stats) {
      __syncthreads();
      if (tid < 32 * WM) {
        const int row = mBase0 + tid, w0 = tid >> 5, r32 = tid & 31;
        if (row < M) {
          float s = 0.f, qq = 0.f;
#pragma unroll
          for (int p = 0; p < PW; ++p) { const float* x = Ss + ((w0 + WM * p) * 32 + r32) * 3; s += x[0]; qq += x[1]; }
          a.partial[((size_t)g * M + row) * 2 + 0] = s;
          a.partial[((size_t)g * M + row) * 2 + 1] = qq;
        }
      }
    }
  } else {
    const bool has2 = a.ex2 != nullptr, sums = a.ipart != nullptr;
    const bool need_x = a.erelu || a.es1 != nullptr || has2;
    // The forward's operands (ReLU mask, affine sums) come through buffer loads with out-of-range offsets for dead slots /
    // rows — no branches — and are fetched one group of four rows AHEAD of the group being finished: as a per-element
    // `if (valid) { load; use; store }` the epilogue was sixteen dependent round trips per wave (the data gradient ran at
    // 1.5x the forward's time; round 4).
    const __amdgpu_buffer_rsrc_t rx1 = tg_rsrc(a.ex1 + obase, need_x ? M * L * 4 : 0);
    const __amdgpu_buffer_rsrc_t rx2 = tg_rsrc((has2 ? a.ex2 : a.ex1) + obase, has2 ? M * L * 4 : 0);
    const __amdgpu_buffer_rsrc_t ro1 = tg_rsrc(a.out + obase, M * L * 4);
    const __amdgpu_buffer_rsrc_t ro2 = tg_rsrc((a.out2 ? a.out2 : a.out) + obase, a.out2 ? M * L * 4 : 0);
    int ooff[NQW];
#pragma unroll
    for (int q = 0; q < NQW; ++q) ooff[q] = ovalid[q] ? opos[q] * 4 : TG_OOB;
    float xa[2][4][NQW], xb[2][4][NQW];
    auto fetch = [&](int g, int slot) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = mBase0 + 32 * rt + tg_row32(4 * g + rr, half);
#pragma unroll
        for (int q = 0; q < NQW; ++q) {
          const int vo = (int)((unsigned)(row < M ? ooff[q] : TG_OOB) + (unsigned)(row * L * 4));
          xa[slot][rr][q] = tg_load(rx1, vo, 0);
          xb[slot][rr][q] = tg_load(rx2, vo, 0);
        }
      }
    };
    fetch(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int slot = g & 1;
      if (g + 1 < 4) fetch(g + 1, slot ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = 4 * g + rr;
        const int row32 = tg_row32(r, half), row = mBase0 + 32 * rt + row32;
        const bool rok = row < M;
        const float e1s = (rok && a.es1) ? a.es1[row] : 1.f, e1h = (rok && a.es1) ? a.eh1[row] : 0.f;
        const float e2s = (rok && a.es2) ? a.es2[row] : 1.f, e2h = (rok && a.es2) ? a.eh2[row] : 0.f;
        float u0 = 0.f, u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int q = 0; q < NQW; ++q) {
          const bool ok = ovalid[q] && rok;
          const float xav = xa[slot][rr][q], xbv = xb[slot][rr][q];
          float pre = fmaf(xav, e1s, e1h);
          if (has2) pre += fmaf(xbv, e2s, e2h);
          const float dv = (ok && (!a.erelu || pre > 0.f)) ? acc[q][r] : 0.f;
          const int vo = (int)((unsigned)(rok ? ooff[q] : TG_OOB) + (unsigned)(row * L * 4));
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dv * e1s), ro1, vo, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dv * e2s), ro2, vo, 0, 0);   // zero-sized resource: no dx2
          u0 = fmaf(dv, xav, u0);
          u1 += dv;
          u2 = fmaf(dv, xbv, u2);
        }
        if (sums) {
          u0 = half_sum(u0); u1 = half_sum(u1); u2 = half_sum(u2);
          if (l31 == 0) { float* d = Ss + (wave * 32 + row32) * 3; d[0] = u0; d[1] = u1; d[2] = u2; }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (sums) {
      __syncthreads();
      if (tid < 32 * WM) {
        const int row = mBase0 + tid, w0 = tid >> 5, r32 = tid & 31;
        if (row < M) {
          float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
          for (int p = 0; p < PW; ++p) { const float* x = Ss + ((w0 + WM * p) * 32 + r32) * 3; v0 += x[0]; v1 += x[1]; v2 += x[2]; }
          float* o = a.ipart + ((size_t)g * M + row) * 3;
          o[0] = v0; o[1] = v1; o[2] = v2;
        }
      }
    }
  }
}

// The three bf16 terms of W (Co, Ci, KT) per tap, zero-padded to whole tiles, every (tap, term) plane in MFMA-fragment
// order [row tile of 32][k-step of 16][lane = 32*(k%16 / 8) + row%32][k%8] (a wave's A fragment = one contiguous 1 KB):
//   N image (forward):        [KT][3][MpN = ceil128(Co) x KpN = ceil32(Ci)]   value W[co][ci][tap]
//   T image (data gradient):  [KT][3][MpT = ceil128(Ci) x KpT = ceil32(Co)]   value W[co][ci][KT-1-tap]     (after N)
// One thread per (image, tap, row, 4 consecutive k).
// (jobs: one record per conv, blockIdx.y = record — every dense temporal conv of a model in one launch at the head of the
// step, dsgcn_tconv_wsplit_multi.)
struct TsJob { const float* w; unsigned short* out; int Ci, Co, KT, MpN, KpN, MpT, KpT, pad; };
struct TsTable { TsJob j[DSGCN_TSPLIT_JOBS_MAX]; };

__global__ __launch_bounds__(256) void k_tsplit(TsTable tab) {
  const TsJob& jb = tab.j[blockIdx.y];
  const float* __restrict__ w = jb.w;
  unsigned short* __restrict__ out = jb.out;
  const int Ci = jb.Ci, Co = jb.Co, KT = jb.KT, MpN = jb.MpN, KpN = jb.KpN, MpT = jb.MpT, KpT = jb.KpT;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long t1 = (long)KT * MpN * (KpN >> 2), t2 = (long)KT * MpT * (KpT >> 2);
  float v[4];
  unsigned short* dst;
  long pstride;
  if (i < t1) {
    const int per = MpN * (KpN >> 2);
    const int tap = (int)(i / per), rem = (int)(i - (long)tap * per);
    const int r = rem / (KpN >> 2), k = 4 * (rem - r * (KpN >> 2));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (r < Co && k + e < Ci) ? w[((size_t)r * Ci + k + e) * KT + tap] : 0.f;
    pstride = (long)MpN * KpN;
    dst = out + (size_t)tap * 3 * pstride + ((size_t)(r >> 5) * (KpN >> 4) + (k >> 4)) * 512 + (((k >> 3) & 1) * 32 + (r & 31)) * 8 + (k & 7);
  } else if (i - t1 < t2) {
    const long j = i - t1;
    const int per = MpT * (KpT >> 2);
    const int tap = (int)(j / per), rem = (int)(j - (long)tap * per);
    const int r = rem / (KpT >> 2), k = 4 * (rem - r * (KpT >> 2));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (r < Ci && k + e < Co) ? w[((size_t)(k + e) * Ci + r) * KT + (KT - 1 - tap)] : 0.f;
    pstride = (long)MpT * KpT;
    dst = out + (size_t)KT * 3 * MpN * KpN + (size_t)tap * 3 * pstride + ((size_t)(r >> 5) * (KpT >> 4) + (k >> 4)) * 512 +
          (((k >> 3) & 1) * 32 + (r & 31)) * 8 + (k & 7);
  } else {
    return;
  }
  unsigned p0, p1, p2, q0, q1, q2;
  b3_split(v[0], v[1], p0, p1, p2);
  b3_split(v[2], v[3], q0, q1, q2);
  *reinterpret_cast<u32x2v*>(dst) = u32x2v{p0, q0};
  *reinterpret_cast<u32x2v*>(dst + pstride) = u32x2v{p1, q1};
  *reinterpret_cast<u32x2v*>(dst + 2 * pstride) = u32x2v{p2, q2};
}

struct TsDims { int MpN, KpN, MpT, KpT; size_t bytes; };
TsDims ts_dims(int Ci, int Co, int KT) {
  TsDims d;
  d.MpN = (Co + 127) / 128 * 128; d.KpN = (Ci + 31) / 32 * 32;
  d.MpT = (Ci + 127) / 128 * 128; d.KpT = (Co + 31) / 32 * 32;
  d.bytes = (size_t)KT * 3 * ((size_t)d.MpN * d.KpN + (size_t)d.MpT * d.KpT) * 2;
  return d;
}


// ---- weight gradient ---------------------------------------------------------------------------------------------------
constexpr int TW_TM = 128, TW_TN = 64;             // co rows, ci columns per workgroup
constexpr int TW_RING = 256;                       // stride 1: positions per row of the x' ring (>= 32 + 2*pad*V + 6)
__host__ __device__ constexpr int tw_group(int KT) { return KT > 3 ? 3 : KT; }   // taps staged together (register budget: 16*KT accumulators)

struct TcwArgs {
  const float* x1; const float* x2; const float* s1; const float* h1; const float* s2; const float* h2; int relu;
  const float* gz; const float* z; const float* A0; const float* B0;
  float* dwp; float* dbp; int pstride;
  int n, Ci, Co, T, To, st, V, splits, units, cpl, ccM, ccN;          // T input frames, To output frames, st stride
  int narrow;                  // 1: a co tile with <= 64 live rows gives its dead row tiles' waves half of the taps (below)
};

// dW[co,ci,tap] = sum_{n,p} dz[n,co,p] * x'[n,ci,p + (tap-pad)*V] (zero outside the sample's frames), db[co] = sum dz.
// Work unit = (sample, 32 consecutive positions); a workgroup walks the units of its K split: dz (128 x 32) staged once
// per unit, x' (64 x 32) once per tap with the tap's shift in the load address, taps in groups of tw_group(KT) per LDS image.
// Wave (rt, ct) accumulates the 32 x 32 tile (co 32*rt.., ci 32*ct..) of every tap.  The loads of the next group / unit
// are issued before the products of the current one.
#ifdef DSGCN_LAB
// wall-clock stamps (10 ns) of workgroup 0, thread 0 of the weight gradient: per unit and tap group (issue done, barrier
// passed, products done, commit done); [63] = count.  dsgcn_tcw_phases reads them
__device__ long long g_tcw_stamp[64];
#define TCW_STAMP() do { if (blockIdx.x == 0 && threadIdx.x == 0 && nst < 62) g_tcw_stamp[nst++] = wall_clock64(); } while (0)
#else
#define TCW_STAMP() do {} while (0)
#endif

// ST: 1 = stride 1 (the x' ring), 2 = stride 2
template <int KT, int ST>
__global__ __launch_bounds__(TW_NT, 2) void k_tcw(TcwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef DSGCN_LAB
  int nst = 0;
#endif
  constexpr int TW_G = tw_group(KT);
  constexpr int NG = (KT + TW_G - 1) / TW_G;       // tap groups
  constexpr int pad = (KT - 1) >> 1;
  // both images double-buffered: a step's products read one set while the next step's operands are written into the
  // other — one barrier per tap group instead of two
  constexpr int ASZ = 3 * TW_TM * TG_RB, BSZ = TW_G * 3 * TW_TN * TG_RB;
  char* Ab0 = reinterpret_cast<char*>(lds);                              // [2][3][128 co][RB]   k = positions (stride 1: one set)
  char* Bb0 = Ab0 + (ST == 1 ? 1 : 2) * ASZ;                             // [2][TW_G][3][64 ci][RB]  (stride 1: one set)
  // stride 1: the activated, zero-padded x' values of the unit's WHOLE tap span live in LDS as fp32, [64 ci][ring of
  // TW_RING positions]: a unit needs positions [p0 - pad*V, p0 + 32 + pad*V) and the next unit of the sample the same span
  // moved by 32, so a unit loads, activates and masks only its 32 NEW positions per row (one 16-byte load per thread and
  // stream) and the per-tap operand images are cut out of the ring.  (Round 3 loaded a fresh window per tap group: 3 x 88
  // positions per row and unit = 8.25x every x' value through L2, and the lab stamps of round 5 showed the load / activate
  // step as the longest of a tap group's three: profiles/r05.)
  constexpr int XP = TW_RING + 28;                                       // row stride (floats): 28 mod 64 spreads a wave's 8 rows over the banks
  float* XR = reinterpret_cast<float*>(Bb0 + BSZ);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int Ci = a.Ci, Co = a.Co, V = a.V, L = a.To * V, Lx = a.T * V;   // dz / x plane lengths
  const float invV = 1.f / (float)V;
  const int id = blockIdx.x;
  const int cm = id % a.ccM, cn = (id / a.ccM) % a.ccN, sp = id / (a.ccM * a.ccN);
  const int co0 = cm * TW_TM, ci0 = cn * TW_TN;
  // Wave -> (row tile rt, column tile ct) of the 128 x 64 tile, every tap.  A co tile with <= 64 live rows (the 64-channel
  // layers) would leave the waves of row tiles 2 and 3 — SIMDs 2 and 3 — multiplying zeros while SIMDs 0 and 1 carry two
  // live waves each: there the eight waves are (rt in {0,1}) x (ct) x (tap set ts), waves w and w + 4 (one SIMD) share a
  // tile and split every tap group (ts 0: its first two taps, ts 1: the third), so every SIMD carries one tile's products.
  // Every (co, ci, tap) still has ONE wave walking the same units and k-steps in the same order: bit-identical.
  const bool narrow = a.narrow && Co - co0 <= 64;
  const int rt = narrow ? (wave & 1) : (wave & 3), ct = narrow ? ((wave >> 1) & 1) : (wave >> 2);
  const int ts = narrow ? (wave >> 2) : -1;        // -1: every tap
  auto owns = [&](int tl) { return ts < 0 || (ts == 0) == (tl < 2); };
  const int u0 = (int)((long)a.units * sp / a.splits), u1 = (int)((long)a.units * (sp + 1) / a.splits);
  // A loader: row tid/4 (co), positions 8*(tid&3) ..+7;  B loader: row tid/8 (ci), positions 4*(tid&7) ..+3
  const int arow = tid >> 2, apc = tid & 3, brow = tid >> 3, bpc = tid & 7;
  const int aco = co0 + arow, bci = ci0 + brow;
  const bool dz2 = a.A0 != nullptr, x2on = a.x2 != nullptr, xaff = a.s1 != nullptr;
  const float A0v = (dz2 && aco < Co) ? a.A0[aco] : 0.f, B0v = (dz2 && aco < Co) ? a.B0[aco] : 0.f;
  const float s1v = (xaff && bci < Ci) ? a.s1[bci] : 1.f, h1v = (xaff && bci < Ci) ? a.h1[bci] : 0.f;
  const float s2v = (a.s2 && bci < Ci) ? a.s2[bci] : 1.f, h2v = (a.s2 && bci < Ci) ? a.h2[bci] : 0.f;
  const float lo = a.relu ? 0.f : -__builtin_inff();
  f32x4 ag[2], az[2];                              // raw dz operands of a unit
  float bx[TW_G][4], bx2[TW_G][4];                 // raw x' operands of a tap group (aligned 16-byte quads + uniform selects
                                                   // for stride 1 were tried: fewer address-path cycles, but the second
                                                   // quad's registers spill next to 144 accumulators: no faster)
  auto unit_base = [&](int u, int& ns, int& p0) { ns = u / a.cpl; p0 = (u - ns * a.cpl) * 32; };
  // ring loader (stride 1).  Incremental: thread -> (row tid / 8, quad tid % 8) of the 32 new positions of a unit.
  f32x4 xn = {0.f, 0.f, 0.f, 0.f}, xn2 = {0.f, 0.f, 0.f, 0.f};
  f32x4* XPs = reinterpret_cast<f32x4*>(XR + TW_TN * XP);       // [64] (s1, h1, s2, h2) of the window's channels
  const int padV = pad * V;
  if constexpr (ST == 1) {
    if (tid < TW_TN) {
      const int c_ = ci0 + tid;
      const bool ok = c_ < Ci;
      XPs[tid] = f32x4{(ok && xaff) ? a.s1[c_] : 1.f, (ok && xaff) ? a.h1[c_] : 0.f, (ok && a.s2) ? a.s2[c_] : 1.f,
                       (ok && a.s2) ? a.h2[c_] : 0.f};
    }
  }
  const bool xact = xaff || a.relu || x2on;
  // activated, masked quad at plane position xp (a multiple of 4) of row r -> its ring slot
  auto ring_put = [&](int r, int xp, const f32x4& v1, const f32x4& v2) {
    const f32x4 par = XPs[r];
    const float r1[4] = {v1.x, v1.y, v1.z, v1.w}, r2[4] = {v2.x, v2.y, v2.z, v2.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = r1[e];
      if (xact) {
        x = fmaf(x, par.x, par.y);
        if (x2on) x += fmaf(r2[e], par.z, par.w);
        x = fmaxf(x, lo);
      }
      o[e] = (xp + e >= 0 && xp + e < Lx && ci0 + r < Ci) ? x : 0.f;     // zero padding applies to the activated value
    }
    *reinterpret_cast<f32x4*>(XR + r * XP + ((xp + 4 * TW_RING) & (TW_RING - 1))) = f32x4{o[0], o[1], o[2], o[3]};
  };
  auto x_rsrc = [&](int ns, __amdgpu_buffer_rsrc_t& rx1, __amdgpu_buffer_rsrc_t& rx2) {
    rx1 = tg_rsrc(a.x1 + (size_t)ns * Ci * Lx, Ci * Lx * 4);
    rx2 = tg_rsrc((x2on ? a.x2 : a.x1) + (size_t)ns * Ci * Lx, x2on ? Ci * Lx * 4 : 0);
  };
  // aligned end of the span of the unit at p0
  auto span_hi = [&](int p0) { return (p0 + 32 + padV + 3) & ~3; };
  auto issueNew = [&](int u) {                     // the 32 new positions of unit u (a continuation of its sample's previous unit)
    int ns, p0;
    unit_base(u, ns, p0);
    __amdgpu_buffer_rsrc_t rx1, rx2;
    x_rsrc(ns, rx1, rx2);
    const int r = tid >> 3, xp = span_hi(p0) - 32 + 4 * (tid & 7);
    // (quads are 4-aligned and the plane length is a multiple of 4: a quad is entirely inside or outside the plane)
    const int vo = (ci0 + r < Ci && xp >= 0 && xp < Lx) ? ((ci0 + r) * Lx + xp) * 4 : TG_OOB;
    xn = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx1, vo, 0, 0));
    xn2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx2, vo, 0, 0));
  };
  auto stageNew = [&](int u) {
    int ns, p0;
    unit_base(u, ns, p0);
    ring_put(tid >> 3, span_hi(p0) - 32 + 4 * (tid & 7), xn, xn2);
  };
  auto reload = [&](int u) {                       // the whole span of unit u (first unit of a workgroup / of a sample)
    int ns, p0;
    unit_base(u, ns, p0);
    __amdgpu_buffer_rsrc_t rx1, rx2;
    x_rsrc(ns, rx1, rx2);
    const int lo4 = (p0 - padV) & ~3, nq = (span_hi(p0) - lo4) >> 2;     // (floor: also for negative starts)
    for (int q0 = tid; q0 < TW_TN * nq; q0 += 4 * TW_NT) {
      f32x4 w1[4], w2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q_ = q0 + TW_NT * i, r = q_ / nq, xp = lo4 + 4 * (q_ - r * nq);
        const int vo = (q_ < TW_TN * nq && ci0 + r < Ci && xp >= 0 && xp < Lx) ? ((ci0 + r) * Lx + xp) * 4 : TG_OOB;
        w1[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx1, vo, 0, 0));
        w2[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx2, vo, 0, 0));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q_ = q0 + TW_NT * i, r = q_ / nq;
        if (q_ < TW_TN * nq) ring_put(r, lo4 + 4 * (q_ - r * nq), w1[i], w2[i]);
      }
    }
  };
  auto buildB = [&](int u, int grp) {
    int ns, p0;
    unit_base(u, ns, p0);
    const int ws = p0 + (grp * TW_G - pad) * V + 4 * bpc + 4 * TW_RING;   // (+ a multiple of the ring: non-negative)
    const float* xr = XR + brow * XP;
#pragma unroll
    for (int tl = 0; tl < TW_G; ++tl) {
      const int tap = grp * TW_G + tl;
      if (tap < KT) {
        const int q_ = ws + tl * V;
        const float v0 = xr[q_ & (TW_RING - 1)], v1 = xr[(q_ + 1) & (TW_RING - 1)], v2 = xr[(q_ + 2) & (TW_RING - 1)],
                    v3 = xr[(q_ + 3) & (TW_RING - 1)];
        unsigned p0_, p1_, p2_, q0_, q1_, q2_;
        b3_split(v0, v1, p0_, p1_, p2_);
        b3_split(v2, v3, q0_, q1_, q2_);
        char* base = Bb0 + (tl * 3 * TW_TN + brow) * TG_RB + bpc * 8;
        *reinterpret_cast<u32x2v*>(base) = u32x2v{p0_, q0_};
        *reinterpret_cast<u32x2v*>(base + TW_TN * TG_RB) = u32x2v{p1_, q1_};
        *reinterpret_cast<u32x2v*>(base + 2 * TW_TN * TG_RB) = u32x2v{p2_, q2_};
      }
    }
  };
  auto issueA = [&](int u) {
    int ns, p0;
    unit_base(u, ns, p0);
    const int p = p0 + 8 * apc;
    // buffer loads over the sample's planes: rows / quads outside get an out-of-range offset (zeros) — a conditional
    // load is a branch and a wait of its own (measured: 2 us to ISSUE a tap group's loads that way)
    const __amdgpu_buffer_rsrc_t rg = tg_rsrc(a.gz + (size_t)ns * Co * L, Co * L * 4);
    const __amdgpu_buffer_rsrc_t rz = tg_rsrc((dz2 ? a.z : a.gz) + (size_t)ns * Co * L, dz2 ? Co * L * 4 : 0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool ok = aco < Co && p + 4 * h < L;          // (L % 4 == 0: a quad is inside or outside the plane)
      const int vo = ok ? (aco * L + p + 4 * h) * 4 : TG_OOB;
      ag[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, vo, 0, 0));
      az[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, vo, 0, 0));
    }
  };
  // x' position of dz position q for tap 0: (st*t' - pad)*V + v; a tap adds V.  -1: q outside the plane
  int xq[4];
  auto xpos = [&](int p0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = p0 + 4 * bpc + e;
      int t_, v_;
      divmod_small(q, V, invV, t_, v_);
      xq[e] = q < L ? (a.st * t_ - pad) * V + v_ : -(1 << 28);
    }
  };
  auto issueB = [&](int u, int grp) {
    int ns, p0;
    unit_base(u, ns, p0);
    if (grp == 0) xpos(p0);
    const __amdgpu_buffer_rsrc_t rx1 = tg_rsrc(a.x1 + (size_t)ns * Ci * Lx, Ci * Lx * 4);
    const __amdgpu_buffer_rsrc_t rx2 = tg_rsrc((x2on ? a.x2 : a.x1) + (size_t)ns * Ci * Lx, x2on ? Ci * Lx * 4 : 0);
#pragma unroll
    for (int tl = 0; tl < TW_G; ++tl) {
      const int tap = grp * TW_G + tl;
      if (tap < KT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = xq[e] + tap * V;
          const bool ok = bci < Ci && q >= 0 && q < Lx;
          const int vo = ok ? (bci * Lx + q) * 4 : TG_OOB;
          bx[tl][e] = tg_load(rx1, vo, 0);
          bx2[tl][e] = tg_load(rx2, vo, 0);
        }
      }
    }
  };
  float dbacc = 0.f;
  auto commitA = [&](int u, int buf) {
    char* Ab = Ab0 + buf * ASZ;
    int ns, p0;
    unit_base(u, ns, p0);
    const int p = p0 + 8 * apc;
    float v[8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = aco < Co && p + 4 * h < L;
        float x = ag[h][e];
        if (dz2) x += fmaf(B0v, az[h][e], A0v);
        v[4 * h + e] = ok ? x : 0.f;
      }
    float sv = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    sv += __shfl_xor(sv, 1, 64);
    sv += __shfl_xor(sv, 2, 64);
    dbacc += sv;
    unsigned t0[4], t1[4], t2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b3_split(v[2 * j], v[2 * j + 1], t0[j], t1[j], t2[j]);
    char* base = Ab + arow * TG_RB + apc * 16;
    *reinterpret_cast<u32x4v*>(base) = u32x4v{t0[0], t0[1], t0[2], t0[3]};
    *reinterpret_cast<u32x4v*>(base + TW_TM * TG_RB) = u32x4v{t1[0], t1[1], t1[2], t1[3]};
    *reinterpret_cast<u32x4v*>(base + 2 * TW_TM * TG_RB) = u32x4v{t2[0], t2[1], t2[2], t2[3]};
  };
  auto commitB = [&](int u, int grp, int buf) {
    char* Bb = Bb0 + buf * BSZ;
    int ns, p0;
    unit_base(u, ns, p0);
#pragma unroll
    for (int tl = 0; tl < TW_G; ++tl) {
      const int tap = grp * TW_G + tl;
      if (tap < KT) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = xq[e] + tap * V;      // (xq belongs to this unit: the next unit's is computed after its last commit)
          const bool ok = bci < Ci && q >= 0 && q < Lx;
          float x = bx[tl][e];
          if (xaff || a.relu || x2on) {
            x = fmaf(x, s1v, h1v);
            if (x2on) x += fmaf(bx2[tl][e], s2v, h2v);
            x = fmaxf(x, lo);
          }
          v[e] = ok ? x : 0.f;                     // zero padding applies to the activated value
        }
        unsigned p0_, p1_, p2_, q0_, q1_, q2_;
        b3_split(v[0], v[1], p0_, p1_, p2_);
        b3_split(v[2], v[3], q0_, q1_, q2_);
        char* base = Bb + (tl * 3 * TW_TN + brow) * TG_RB + bpc * 8;
        *reinterpret_cast<u32x2v*>(base) = u32x2v{p0_, q0_};
        *reinterpret_cast<u32x2v*>(base + TW_TN * TG_RB) = u32x2v{p1_, q1_};
        *reinterpret_cast<u32x2v*>(base + 2 * TW_TN * TG_RB) = u32x2v{p2_, q2_};
      }
    }
  };
  f32x16 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const char* Af0 = Ab0 + (32 * rt + l31) * TG_RB + 16 * half;
  const char* Bf0 = Bb0 + (32 * ct + l31) * TG_RB + 16 * half;
  auto products = [&](int grp, int abuf, int bbuf) {
    const char* Af = Af0 + abuf * ASZ;
    const char* Bf = Bf0 + bbuf * BSZ;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[3];
#pragma unroll
      for (int t = 0; t < 3; ++t)
        af[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(Af + t * TW_TM * TG_RB + 32 * ks));
#pragma unroll
      for (int tl = 0; tl < TW_G; ++tl) {
        const int tap = grp * TW_G + tl;
        if (tap < KT && owns(tl)) {
          bf16x8 bf[3];
#pragma unroll
          for (int t = 0; t < 3; ++t)
            bf[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(Bf + ((tl * 3 + t) * TW_TN) * TG_RB + 32 * ks));
          f32x16 c = acc[tap];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], c, 0, 0, 0);
          acc[tap] = c;
        }
      }
    }
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  if constexpr (ST == 1) {
    // Units in order; `fresh` = the first unit of this workgroup or of a sample (the ring holds nothing of it yet).
    auto sample_of = [&](int u_) { return u_ / a.cpl; };
    lds_barrier();                                 // the affine table of the window's channels is visible
    if (u0 < u1) issueA(u0);
    for (int u = u0; u < u1; ++u) {
      const bool fresh = u == u0 || sample_of(u) != sample_of(u - 1);
      const bool cont1 = u + 1 < u1 && sample_of(u + 1) == sample_of(u);          // u + 1 continues this sample's ring
      const bool cont2 = cont1 && u + 2 < u1 && sample_of(u + 2) == sample_of(u);
      if (fresh) {
        // (every wave is past the second barrier of the previous unit's last tap group: nobody reads the ring any more)
        reload(u);
        if (cont1) issueNew(u + 1);
      }
#pragma unroll
      for (int grp = 0; grp < NG; ++grp) {           // (unrolled: the tap indices of the accumulators are compile-time)
        TCW_STAMP();
        lds_barrier();                               // every wave is done with the previous step's products: images and dz image are free
        if (grp == 0) commitA(u, 0);
        buildB(u, grp);
        TCW_STAMP();
        lds_barrier();                               // images (and the dz image) complete
        products(grp, 0, 0);
        TCW_STAMP();
        if (grp == 0) {
          // the next unit's 32 new positions go into ring slots no tap group of THIS unit reads after its first one
          // (new: [p0 + 32 + pad*V, + 32); still to be read: [p0 + (TW_G - pad)*V, p0 + 32 + pad*V))
          if (cont1) stageNew(u + 1);
          if (cont2) issueNew(u + 2);
          if (u + 1 < u1) issueA(u + 1);
        }
        TCW_STAMP();
      }
    }
  } else {
  if (u0 < u1) {
    issueA(u0);
    issueB(u0, 0);
  }
  int gs = 0;                                      // running tap-group step: parity = B buffer
  for (int u = u0; u < u1; ++u) {
    const int abuf = (u - u0) & 1;
    commitA(u, abuf);
    commitB(u, 0, gs & 1);
#pragma unroll
    for (int grp = 0; grp < NG; ++grp, ++gs) {
      // loads that land while this group's products run: the next group of this unit, or the next unit's first group
      TCW_STAMP();
      if (grp + 1 < NG) issueB(u, grp + 1);
      else if (u + 1 < u1) { issueA(u + 1); issueB(u + 1, 0); }
      TCW_STAMP();
      lds_barrier();                               // this step's images complete; the other set is free (its readers passed the previous barrier)
      TCW_STAMP();
      products(grp, abuf, gs & 1);
      TCW_STAMP();
      if (grp + 1 < NG) commitB(u, grp + 1, (gs + 1) & 1);
    }
  }
  }
#ifdef DSGCN_LAB
  if (blockIdx.x == 0 && threadIdx.x == 0) g_tcw_stamp[63] = nst;
#endif
  // results: row co = 32*rt + row32(r, half), column ci = 32*ct + l31 of the workgroup's tile
  float* dw = a.dwp + (size_t)sp * a.pstride;
  const int ci = ci0 + 32 * ct + l31;
#pragma unroll
  for (int t = 0; t < KT; ++t)
    if (owns(t % TW_G)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * rt + tg_row32(r, half);
        if (co < Co && ci < Ci) dw[((size_t)co * Ci + ci) * KT + t] = acc[t][r];
      }
    }
  int tide = tid;
  asm volatile("" : "+v"(tide));                 // (same: aco's sign-extended copy is not kept across the loop)
  const int acoe = co0 + (tide >> 2);
  if (cn == 0 && (tide & 3) == 0 && acoe < Co) a.dbp[(size_t)sp * a.pstride + acoe] = dbacc;
}

struct TwPlan { int cpl, units, ccM, ccN, splits, nqi; size_t lds; };
bool tw_plan(int n, int Ci, int Co, int To, int V, int KT, TwPlan* p, int st = 1) {
  const long L = (long)To * V;                     // positions of dz per plane
  if (L % 4 != 0 || (KT != 3 && KT != 5 && KT != 9)) return false;
  if ((long)Co * L * 4 >= (1L << 31) - 64 || (long)Ci * L * 2 * 4 >= (1L << 31) - 64) return false;   // (buffer resources per sample: 32-bit offsets)
  p->cpl = (int)((L + 31) / 32);
  p->units = n * p->cpl;
  p->ccM = (Co + TW_TM - 1) / TW_TM;
  p->ccN = (Ci + TW_TN - 1) / TW_TN;
  const int tiles = p->ccM * p->ccN;
  const long pstride = (long)Co * Ci * KT + Co;
  int splits = (512 + tiles - 1) / tiles;          // ~2 workgroups' worth of work per CU
  const long cap = (96L << 20) / (pstride * 4);    // partial buffer <= 96 MB
  if (splits > cap) splits = (int)cap;
  if (splits > p->units) splits = p->units;
  if (splits < 1) splits = 1;
  // One workgroup per CU (150 KB of LDS): a grid that is not a whole number of 256-workgroup rounds leaves most of the chip
  // idle for its last round.  The 96 MB cap used to produce exactly that — 42 x 8, 170 x 2, 340 x 1, 85 x 4 = 336-340
  // workgroups, 1.3 rounds for the time of two (round 5) — so the split count is rounded down to whole rounds.
  if ((long)splits * tiles > 256) {
    const int rounds = (int)((long)splits * tiles / 256);
    splits = rounds * 256 / tiles;
    if (splits < 1) splits = 1;
  }
  p->splits = splits;
  {
    const size_t asz = (size_t)3 * TW_TM * TG_RB, bsz = (size_t)tw_group(KT) * 3 * TW_TN * TG_RB;
    if (st == 1 && 32 + (KT - 1) * V + 6 > TW_RING) return false;  // (the unit's tap span must fit the x' ring)
    p->nqi = 3;
    p->lds = st == 1 ? asz + bsz + (size_t)TW_TN * (TW_RING + 28) * sizeof(float) + TW_TN * 16 : 2 * (asz + bsz);
  }
  return true;
}

struct TgPlan { int R, tps, cc, tiles, wm; size_t lds; unsigned grid; };
// K input channels, M output rows, Ts / To frames of the source / output, st = lane row stride (forward stride)
bool tg_plan(int n, int K, int M, int Ts, int To, int V, int KT, int st, TgPlan* p) {
  if (V < 1 || V > 32 || KT < 1 || KT > 9 || !(KT & 1) || Ts < 1 || To < 1 || st < 1 || st > 2) return false;
  const int T = To;
  int R = 128 / V < T ? 128 / V : T;
  // (strided: the input tile spans st*(R-1) + KT frames — fewer output frames per tile until its staging fits)
  while (R > 1 && (((st * (R - 1) + KT) * V * 8 + TG_NT - 1) / TG_NT > TG_MAXIT || (size_t)3 * (st * (R - 1) + KT) * V * TG_RB > 160 * 1024)) --R;
  const int NR = (st * (R - 1) + KT) * V;
  if ((NR * 8 + TG_NT - 1) / TG_NT > TG_MAXIT) return false;
  if ((long)K * Ts * V * 4 >= (1L << 31) - 64 || (long)M * To * V * 4 >= (1L << 31) - 64) return false;
  const int Kp = (K + 31) / 32 * 32;
  p->R = R;
  p->tps = (T + R - 1) / R;
  p->wm = M > 64 ? 4 : 2;                          // 128 or 64 output rows per workgroup
  p->cc = (M + 32 * p->wm - 1) / (32 * p->wm);
  p->tiles = n * p->tps;
  (void)Kp;
  p->lds = (size_t)3 * NR * TG_RB;
  if (p->lds > 160 * 1024) return false;
  p->grid = (unsigned)((p->tiles + 7) / 8 * 8 * p->cc);
  return true;
}

int g_tg_quad = 1;                                 // lab knob (dsgcn_tconv_tuning): stride-1 staging with 16-byte loads on / off
int g_tw_narrow = 1;                               // lab knob 1: k_tcw's wave mapping for co tiles of <= 64 rows on / off

template <int EPI, int WM, bool QS>
int tg_launch_qs(const TcgArgs& a, int mode, const TgPlan& p, hipStream_t st) {
  static bool raised = false;
  if (!raised) {
    const void* fs[3] = {reinterpret_cast<const void*>(&k_tcg<0, EPI, WM, QS>), reinterpret_cast<const void*>(&k_tcg<1, EPI, WM, QS>),
                         reinterpret_cast<const void*>(&k_tcg<2, EPI, WM, QS>)};
    for (const void* f : fs) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      if (e != hipSuccess) return (int)e;
    }
    raised = true;
  }
  const dim3 grid(p.grid), blk(TG_NT);
  if (mode == 0) hipLaunchKernelGGL((k_tcg<0, EPI, WM, QS>), grid, blk, p.lds, st, a);
  else if (mode == 1) hipLaunchKernelGGL((k_tcg<1, EPI, WM, QS>), grid, blk, p.lds, st, a);
  else hipLaunchKernelGGL((k_tcg<2, EPI, WM, QS>), grid, blk, p.lds, st, a);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

template <int EPI, int WM>
int tg_launch_wm(const TcgArgs& a, int mode, const TgPlan& p, hipStream_t st) {
  // stride 1 on both sides and source planes a multiple of 4 floats long: the 16-byte staging
  const bool qs = g_tg_quad && a.st == 1 && a.up == 1 && ((a.Ts * a.V) & 3) == 0;
  return qs ? tg_launch_qs<EPI, WM, true>(a, mode, p, st) : tg_launch_qs<EPI, WM, false>(a, mode, p, st);
}

template <int EPI>
int tg_launch(const TcgArgs& a, int mode, const TgPlan& p, hipStream_t st) {
  return p.wm == 4 ? tg_launch_wm<EPI, 4>(a, mode, p, st) : tg_launch_wm<EPI, 2>(a, mode, p, st);
}

}  // namespace

extern "C" {

// Bytes of the pre-split weight image of a dense (KT,1) temporal conv; 0 = the GEMM form does not take this shape
// (then: dsgcn_tapconv_*).
size_t dsgcn_tconv_ws_bytes(int n, int Ci, int Co, int T, int V, int KT, int stride) {
  TgPlan pf, pb;
  if (n <= 0 || Ci <= 0 || Co <= 0 || T <= 0 || stride < 1 || stride > 2) return 0;
  const int To = (T + stride - 1) / stride;
  TwPlan pw;
  if (!tg_plan(n, Ci, Co, T, To, V, KT, stride, &pf) || !tg_plan(n, Co, Ci, To, T, V, KT, 1, &pb) ||
      !tw_plan(n, Ci, Co, To, V, KT, &pw, stride))
    return 0;
  return ts_dims(Ci, Co, KT).bytes;
}

// The weight images of njobs convs in one launch per DSGCN_TSPLIT_JOBS_MAX records (include/dsgcn_jobs.h).
int dsgcn_tconv_wsplit_multi(const dsgcn_tsplit_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs <= 0) return DSGCN_EINVAL;
  for (int i = 0; i < njobs; ++i)
    if (!jobs[i].w || !jobs[i].ws || jobs[i].Ci <= 0 || jobs[i].Co <= 0 || jobs[i].KT <= 0) return DSGCN_EINVAL;
  for (int i0 = 0; i0 < njobs; i0 += DSGCN_TSPLIT_JOBS_MAX) {
    const int m = njobs - i0 < DSGCN_TSPLIT_JOBS_MAX ? njobs - i0 : DSGCN_TSPLIT_JOBS_MAX;
    TsTable t = {};
    long most = 0;
    for (int i = 0; i < m; ++i) {
      const dsgcn_tsplit_job& a = jobs[i0 + i];
      const TsDims d = ts_dims(a.Ci, a.Co, a.KT);
      t.j[i] = TsJob{a.w, static_cast<unsigned short*>(a.ws), a.Ci, a.Co, a.KT, d.MpN, d.KpN, d.MpT, d.KpT, 0};
      const long total = (long)a.KT * ((long)d.MpN * (d.KpN >> 2) + (long)d.MpT * (d.KpT >> 2));
      most = total > most ? total : most;
    }
    hipLaunchKernelGGL(k_tsplit, dim3((unsigned)((most + 255) / 256), (unsigned)m), dim3(256), 0, (hipStream_t)stream, t);
    DSGCN_LAUNCH_CHECK();
  }
  return 0;
}

int dsgcn_tconv_wsplit(const float* w, int Ci, int Co, int KT, void* ws, void* stream) {
  const dsgcn_tsplit_job j = {w, ws, Ci, Co, KT, 0};
  return dsgcn_tconv_wsplit_multi(&j, 1, stream);
}

// rows of the forward's statistics partials (rows, Co, 2) and of the data gradient's input-affine partials (rows, Ci, 3)
// Partial rows: forward (rows, Co, 2) over the OUTPUT frames, data gradient (rows, Ci, 3) over the INPUT frames.
// which: 0 = forward, 1 = data gradient.
int dsgcn_tconv_rows(int which, int n, int Ci, int Co, int T, int V, int KT, int stride) {
  if (n <= 0 || T <= 0 || stride < 1 || stride > 2) return 0;
  const int To = (T + stride - 1) / stride;
  TgPlan p;
  const bool ok = which == 0 ? tg_plan(n, Ci, Co, T, To, V, KT, stride, &p) : tg_plan(n, Co, Ci, To, T, V, KT, 1, &p);
  return ok ? p.tiles : 0;
}

int dsgcn_tconv_fwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2, const float* h2,
                    int relu, const void* ws, const float* bias, float* z, float* partial, int n, int Ci, int Co, int T,
                    int V, int KT, int stride, void* stream) {
  if (!x1 || !ws || !z || (s1 && !h1) || (s2 && !h2) || (x2 && !s1 && false)) return DSGCN_EINVAL;
  TgPlan p;
  if (stride < 1 || stride > 2 || T <= 0) return DSGCN_EUNSUPPORTED;
  const int To = (T + stride - 1) / stride;
  if (!tg_plan(n, Ci, Co, T, To, V, KT, stride, &p)) return DSGCN_EUNSUPPORTED;
  const TsDims d = ts_dims(Ci, Co, KT);
  TcgArgs a = {};
  a.Ts = T; a.To = To; a.st = stride; a.up = 1;
  a.b1 = x1; a.b2 = x2; a.ps1 = s1; a.ph1 = h1; a.ps2 = s2; a.ph2 = h2; a.relu = relu;
  a.wsp = static_cast<const unsigned short*>(ws); a.bias = bias; a.out = z; a.partial = partial;
  a.n = n; a.K = Ci; a.M = Co; a.V = V; a.KT = KT; a.R = p.R; a.tps = p.tps; a.cc = p.cc; a.Mp = d.MpN; a.Kp = d.KpN;
  const int mode = x2 ? 2 : ((s1 || relu) ? 1 : 0);
  return tg_launch<0>(a, mode, p, (hipStream_t)stream);
}

// dz = gz + A0 + B0*z (A0 / B0 / z may be NULL together);  x1/x2/s*/h*/relu describe the forward's virtual input (mask /
// affine epilogue); dx1 (, dx2) fully written; ipart (dsgcn_tconv_rows, Ci, 3) = [sum dv*x1, sum dv, sum dv*x2] or NULL.
int dsgcn_tconv_dgrad(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                      const float* h2, int relu, const void* ws, const float* z, const float* gz, const float* A0,
                      const float* B0, float* dx1, float* dx2, float* ipart, int n, int Ci, int Co, int T, int V, int KT,
                      int stride, void* stream) {
  if (!x1 || !ws || !gz || !dx1 || (A0 && (!B0 || !z)) || (x2 && !dx2)) return DSGCN_EINVAL;
  TgPlan p;
  if (stride < 1 || stride > 2 || T <= 0) return DSGCN_EUNSUPPORTED;
  const int To = (T + stride - 1) / stride;        // frames of z / gz
  if (!tg_plan(n, Co, Ci, To, T, V, KT, 1, &p)) return DSGCN_EUNSUPPORTED;
  const TsDims d = ts_dims(Ci, Co, KT);
  TcgArgs a = {};
  a.Ts = To; a.To = T; a.st = 1; a.up = stride;
  a.b1 = gz; a.b2 = A0 ? z : nullptr;
  a.ps1 = nullptr; a.ph1 = A0; a.ps2 = B0; a.ph2 = nullptr; a.relu = 0;
  a.wsp = static_cast<const unsigned short*>(ws) + (size_t)KT * 3 * d.MpN * d.KpN;
  a.out = dx1; a.out2 = dx2; a.ipart = ipart;
  a.ex1 = x1; a.ex2 = x2; a.es1 = s1; a.eh1 = h1; a.es2 = s2; a.eh2 = h2; a.erelu = relu;
  a.n = n; a.K = Co; a.M = Ci; a.V = V; a.KT = KT; a.R = p.R; a.tps = p.tps; a.cc = p.cc; a.Mp = d.MpT; a.Kp = d.KpT;
  return tg_launch<1>(a, A0 ? 2 : 0, p, (hipStream_t)stream);
}

// K splits of the weight gradient for this shape (0 = shape not taken); split s writes dwp + s*pstride (Co*Ci*KT floats,
// the layout of the weight) and dbp + s*pstride (Co floats).
int dsgcn_tconv_wgrad_splits(int n, int Ci, int Co, int T, int V, int KT, int stride) {
  TwPlan p;
  if (n <= 0 || Ci <= 0 || Co <= 0 || T <= 0 || stride < 1 || stride > 2) return 0;
  if (!tw_plan(n, Ci, Co, (T + stride - 1) / stride, V, KT, &p, stride)) return 0;
  return p.splits;
}

int dsgcn_tconv_wgrad(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                      const float* h2, int relu, const float* z, const float* gz, const float* A0, const float* B0,
                      float* dwp, float* dbp, int pstride, int n, int Ci, int Co, int T, int V, int KT, int stride,
                      void* stream) {
  if (!x1 || !gz || !dwp || !dbp || (s1 && !h1) || (s2 && !h2) || (A0 && (!B0 || !z))) return DSGCN_EINVAL;
  TwPlan p;
  if (stride < 1 || stride > 2 || T <= 0) return DSGCN_EUNSUPPORTED;
  const int To = (T + stride - 1) / stride;
  if (!tw_plan(n, Ci, Co, To, V, KT, &p, stride)) return DSGCN_EUNSUPPORTED;
  if (pstride < Co * Ci * KT) return DSGCN_EINVAL;
  TcwArgs a = {};
  a.x1 = x1; a.x2 = x2; a.s1 = s1; a.h1 = h1; a.s2 = s2; a.h2 = h2; a.relu = relu;
  a.gz = gz; a.z = z; a.A0 = A0; a.B0 = B0; a.dwp = dwp; a.dbp = dbp; a.pstride = pstride;
  a.n = n; a.Ci = Ci; a.Co = Co; a.T = T; a.To = To; a.st = stride; a.V = V; a.splits = p.splits; a.units = p.units; a.cpl = p.cpl;
  a.ccM = p.ccM; a.ccN = p.ccN; a.narrow = g_tw_narrow;
  const dim3 grid((unsigned)(p.splits * p.ccM * p.ccN)), blk(TW_NT);
  static bool raised = false;
  if (!raised) {
    const void* fs[6] = {reinterpret_cast<const void*>(&k_tcw<9, 1>), reinterpret_cast<const void*>(&k_tcw<5, 1>),
                         reinterpret_cast<const void*>(&k_tcw<3, 1>), reinterpret_cast<const void*>(&k_tcw<9, 2>),
                         reinterpret_cast<const void*>(&k_tcw<5, 2>), reinterpret_cast<const void*>(&k_tcw<3, 2>)};
    for (const void* f : fs) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    raised = true;
  }
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1) {
    if (KT == 9) hipLaunchKernelGGL((k_tcw<9, 1>), grid, blk, p.lds, st, a);
    else if (KT == 5) hipLaunchKernelGGL((k_tcw<5, 1>), grid, blk, p.lds, st, a);
    else hipLaunchKernelGGL((k_tcw<3, 1>), grid, blk, p.lds, st, a);
  } else {
    if (KT == 9) hipLaunchKernelGGL((k_tcw<9, 2>), grid, blk, p.lds, st, a);
    else if (KT == 5) hipLaunchKernelGGL((k_tcw<5, 2>), grid, blk, p.lds, st, a);
    else hipLaunchKernelGGL((k_tcw<3, 2>), grid, blk, p.lds, st, a);
  }
  DSGCN_LAUNCH_CHECK();
  return 0;
}

#ifdef DSGCN_LAB
int dsgcn_tcw_phases(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tcw_stamp), sizeof(long long) * 64);
}
int dsgcn_tconv_tuning(int key, int value) {
  if (key == 0) { g_tg_quad = value; return 0; }
  if (key == 1) { g_tw_narrow = value; return 0; }
  return DSGCN_EINVAL;
}
#endif

}  // extern "C"
