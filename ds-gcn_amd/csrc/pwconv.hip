// K-C: 1x1 channel mix ("pointwise conv") on the f32 matrix core, with the surrounding train-mode BatchNorm /
// ReLU / residual work folded into its load prologue and store epilogue.
// Replaces, per call, the reference's Conv2d(1x1) + BatchNorm2d + ReLU (+ residual add) ATen chains:
//   pyskl/models/gcns/utils/gcn.py:2165-2169,2209-2215,2236,2363-2365 (pre / post / down / bn of dgphgcn1)
//   pyskl/models/gcns/utils/tcn.py:379-404,409,422,427 (dgmstcn branch 1x1 convs, global joint, transform, bn)
//   pyskl/models/gcns/utils/tcn.py:21-28 with kernel_size=1 (block residual, dgstgcn.py:59)
//
//   v[n,ci,t,v]  = relu?( x1*s1[ci]+h1[ci] (+ x2*s2[ci]+h2[ci] | + x2) )        "virtual input": deferred BN of the
//   z[n,co,t',v] = sum_ci W[co,ci] * v[n,ci,t'*stride,v] + b[co]                 producer applied while loading
//   zaug[n,co,t'] = mean_v z  (= W . mean_v v + b: the dgmstcn "global joint" column, by linearity)
//   partial[blk,co,0:2] = sum / sum of squares of this block's outputs (incl. zaug)  -> batch statistics
//
// Work decomposition (k_pwconv_fwd2 / _dgrad2, below): wave-independent — each wave owns one 32-position tile of one
// sample and <= 64 output channels; the B operand comes straight from HBM in MFMA fragment shape (raw buffer loads), only
// the weights pass through LDS; per-channel sums leave through an LDS transpose, one partial row per block
// (deterministic two-stage BN statistics, finalised in fp64 by k_bn_finalize).  wgrad stages both operands in LDS.
// Bound: HBM for Ci,Co <= 64 (16 FLOP/B), f32 MFMA above (32-64 FLOP/B vs ridge ~20).
#include "common.h"
#include "bn_jobs.h"

namespace {

constexpr int KCH = 16;            // input channels per LDS chunk
constexpr int WSTR = KCH + 1;      // W tile row stride (odd)
constexpr int NTW = 2;             // N-tiles per wave (NPpad <= 256)
constexpr int PW_NT = 256;
constexpr int XCH = KCH / 4;       // channels staged per wave per chunk

struct PwArgs {
  const float* x1; const float* s1; const float* h1;
  const float* x2; const float* s2; const float* h2;
  int relu;
  const float* w; const float* bias;
  float* z; float* zaug; float* partial;
  int n, Ci, Co, T, V, Tout, stride, aug, TR, NPpad, vec, stats, roll, nbx, cc;
};

__device__ __forceinline__ int mfma_row32(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ float virt1(float a, float b, bool has2, float s1, float h1, float s2, float h2,
                                       int relu) {
  float v = fmaf(a, s1, h1);
  if (has2) v += fmaf(b, s2, h2);
  return relu ? fmaxf(v, 0.f) : v;
}

// ------------------------------------------------------------------------------------------------------------
// Wave-independent formulation (product path).  Each wave owns NW consecutive 32-position tiles of one sample and ALL
// output channels of its block (MT 32-channel tiles): the B operand (virtual input) comes straight from HBM in MFMA
// fragment shape — lane (pos, k-half) loads x[ci = k+half][pos]: two 128-B coalesced segments per instruction — with
// UNR k-steps in flight per wave, the deferred BN/ReLU/residual applied in registers; only the weights go through LDS
// (64-channel K chunks, shared by the 4 waves).  No barrier inside a K chunk, every input element is read once.
// ------------------------------------------------------------------------------------------------------------
constexpr int KW = 64;             // K (input-channel) chunk of the LDS weight tile
constexpr int KWS = KW + 1;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
  // raw buffer, stride 0: reads past num_records return 0 (used instead of per-lane predication)
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
constexpr int OOB_OFF = 0x7ffffff0;

// MODE 0: v = x1 ; 1: v = relu?(x1*s1+h1) ; 2: v = relu?(x1*s1+h1 + x2*s2+h2)
// K is walked in groups of UNR k-steps (2 channels each) with two register sets in ping-pong: the loads of group
// g+1 are issued before the MFMAs of group g.  No per-step branches (they made hipcc sink every load next to its
// use: one full HBM round trip per k-step): the chunk is padded to a multiple of 4*UNR channels — padded channels
// have zero weights in Ws and read either valid memory or past the buffer (raw-buffer OOB -> 0).
constexpr int F2_UNR = 4;          // k-steps per register set for narrow inputs (Ci < 48)
constexpr int F2_UNRW = 4;         // ... for wide inputs: 16 k-steps x MFMAs ~ 4096 cycles of cover per set

template <int NW, int MODE, int UNR>
__device__ __forceinline__ void f2_load(float (&xb)[UNR][NW], float (&yb)[UNR][NW], __amdgpu_buffer_rsrc_t r1,
                                        __amdgpu_buffer_rsrc_t r2, const int (&voff)[NW], int cbase, int cstride4) {
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int soff = (cbase + 2 * u) * cstride4;              // wave-uniform: SGPR operand of the buffer load
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      xb[u][j] = buf_load(r1, voff[j], soff);
      if (MODE == 2) yb[u][j] = buf_load(r2, voff[j], soff);
    }
  }
}

template <int MT, int NW, int MODE, int UNR>
__device__ __forceinline__ void f2_compute(const PwArgs& a, f32x16 (&acc)[MT][NW], const float* Ws, const f32x4* Ps4,
                                           const float (&xb)[UNR][NW], const float (&yb)[UNR][NW], int c0,
                                           int kbase, int kc, int half, int l31) {
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int kl = kbase + 2 * u + half;
    const bool cok = kl < kc;
    float av[MT], bv[NW];
#pragma unroll
    for (int m = 0; m < MT; ++m) av[m] = Ws[(32 * m + l31) * KWS + kl];
    f32x4 p = {1.f, 0.f, 1.f, 0.f};
    if (MODE != 0) p = Ps4[cok ? c0 + kl : 0];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      float v = xb[u][j];
      if (MODE != 0) {
        v = fmaf(v, p.x, p.y);
        if (MODE == 2) v += fmaf(yb[u][j], p.z, p.w);
        if (a.relu) v = fmaxf(v, 0.f);
      }
      bv[j] = cok ? v : 0.f;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < NW; ++j)
        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[j], acc[m][j], 0, 0, 0);
  }
}

template <int MT, int NW, int MODE, int UNR>
__device__ __forceinline__ void fwd2_chunk(const PwArgs& a, f32x16 (&acc)[MT][NW], const float* Ws, const f32x4* Ps4,
                                           __amdgpu_buffer_rsrc_t r1, __amdgpu_buffer_rsrc_t r2, const int (&voff)[NW],
                                           int c0, int kc, int cstride4, int half, int l31) {
  constexpr int G = 2 * UNR;                                  // channels per register set
  float xa[UNR][NW], ya[UNR][NW], xb[UNR][NW], yb[UNR][NW];
  f2_load<NW, MODE, UNR>(xa, ya, r1, r2, voff, c0, cstride4);
  for (int k0 = 0; k0 < kc; k0 += 2 * G) {
    f2_load<NW, MODE, UNR>(xb, yb, r1, r2, voff, c0 + k0 + G, cstride4);
    f2_compute<MT, NW, MODE, UNR>(a, acc, Ws, Ps4, xa, ya, c0, k0, kc, half, l31);
    f2_load<NW, MODE, UNR>(xa, ya, r1, r2, voff, c0 + k0 + 2 * G, cstride4);
    if (k0 + G < kc) f2_compute<MT, NW, MODE, UNR>(a, acc, Ws, Ps4, xb, yb, c0, k0 + G, kc, half, l31);
  }
}

// XCD-aware block decode for the wave-independent kernels: the CC blocks that read the same input tile (one per 64-channel
// output chunk) get consecutive slots on the SAME XCD (blockIdx % 8 labels the XCD group, MI355X_MICROARCH.md), so the
// tile's lines are still in that XCD's L2 when the second..CC-th reader arrives; with the plain (x,y,z) order the readers
// are a whole tensor pass apart and every one of them goes to HBM / Infinity Cache.
struct PwBlk { int bx, n, cz; bool live; };
__device__ __forceinline__ PwBlk pw_decode(int nbx, int ntiles, int cc) {
  const int id = blockIdx.x;
  const int xcd = id & 7, slot = id >> 3;
  const int cz = slot % cc;
  const int tile = (slot / cc) * 8 + xcd;
  PwBlk b;
  b.live = tile < ntiles;
  b.cz = cz;
  b.n = tile / nbx;
  b.bx = tile - b.n * nbx;
  return b;
}

// Whole-chunk rolling prefetch (Ci a multiple of 64, one position tile per wave): the 32 B-operand registers of a
// 64-channel chunk are all in flight at once; each is refilled with the same k-step of the NEXT chunk right after its
// MFMAs have consumed it, and the next chunk's weight rows are fetched into registers during the MFMA phase, so a
// chunk boundary costs two raw s_barriers and an LDS write, not a memory round trip (the 4-k-step ping-pong above
// covers only ~512 cycles per set and re-exposes the latency at every chunk: MFMA pipe 40 % busy at 256 channels).
// The barriers are raw s_barrier + lgkmcnt(0): __syncthreads() would drain vmcnt(0) and with it the prefetch.
template <int MT, int MODE>
__device__ __forceinline__ void fwd2_roll(const PwArgs& a, f32x16 (&acc)[MT][1], float* Ws, const f32x4* Ps4,
                                          __amdgpu_buffer_rsrc_t r1, __amdgpu_buffer_rsrc_t r2, int voff, int coBase,
                                          int cstride4, int tid, int half, int l31) {
  const int Ci = a.Ci, Co = a.Co;
  const int nch = Ci / KW;
  float xr[32], yr[MODE == 2 ? 32 : 1];
  f32x4 wr[2 * MT];
  auto loadW = [&](int c0) {
#pragma unroll
    for (int q = 0; q < 2 * MT; ++q) {
      const int f = tid + PW_NT * q;
      const int rowi = f >> 4, kq = (f & 15) * 4;
      const int co = coBase + rowi;
      wr[q] = co < Co ? *reinterpret_cast<const f32x4*>(a.w + (size_t)co * Ci + c0 + kq) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    xr[u] = buf_load(r1, voff, (2 * u) * cstride4);
    if (MODE == 2) yr[u] = buf_load(r2, voff, (2 * u) * cstride4);
  }
  loadW(0);
  for (int c = 0; c < nch; ++c) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // the previous chunk's weight tile is consumed (first: Ps4 written)
#pragma unroll
    for (int q = 0; q < 2 * MT; ++q) {
      const int f = tid + PW_NT * q;
      const int rowi = f >> 4, kq = (f & 15) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) Ws[rowi * KWS + kq + e] = wr[q][e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = c + 1 < nch;
    const int cn = (c + 1) * KW;
    if (more) loadW(cn);
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int kl = 2 * u + half;
      float av[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) av[m] = Ws[(32 * m + l31) * KWS + kl];
      float v = xr[u];
      if (MODE != 0) {
        const f32x4 p = Ps4[c * KW + kl];
        v = fmaf(v, p.x, p.y);
        if (MODE == 2) v += fmaf(yr[u], p.z, p.w);
        if (a.relu) v = fmaxf(v, 0.f);
      }
      if (more) {
        xr[u] = buf_load(r1, voff, (cn + 2 * u) * cstride4);
        if (MODE == 2) yr[u] = buf_load(r2, voff, (cn + 2 * u) * cstride4);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], v, acc[m][0], 0, 0, 0);
    }
  }
}

int g_pw_roll = 13;        // bit 0: rolling prefetch in the forward (on), bit 1: in the data gradient (off: 64 more VGPRs halve
                           // the occupancy, measured -4 % on the step: round-1 A/B), bit 2: dgrad epilogue operands
                           // fetched before the K loop (on: +0.5 %), bit 3: XCD-aware block order in wgrad (on: +0.5 %)

template <int MT, int NW, bool ROLL>
__global__ __launch_bounds__(PW_NT) void k_pwconv_fwd2(PwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ws = lds;                               // [32*MT][KWS]
  f32x4* Ps4 = reinterpret_cast<f32x4*>(lds + ((32 * MT * KWS + 3) & ~3));   // [Ci] : (s1,h1,s2,h2)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const PwBlk bk = pw_decode(a.nbx, a.nbx * a.n, a.cc);
  if (!bk.live) return;
  const int n = bk.n;
  const int coBase = bk.cz * 32 * MT;
  const int V = a.V, Ci = a.Ci, Co = a.Co;
  const int L = a.Tout * V;                      // output positions per (sample, channel)
  const int tile0 = (bk.bx * 4 + wave) * NW;
  const bool has2 = a.x2 != nullptr;
  const bool aff1 = a.s1 != nullptr, aff2 = a.s2 != nullptr;
  const bool wvec = (Ci & 3) == 0;
  const int cstride = a.T * V;

  // per-lane byte offsets of the NW position tiles inside channel (n, half); out-of-range lanes read as 0 (OOB)
  int voff[NW];
  bool pok[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int pos = (tile0 + j) * 32 + l31;
    pok[j] = pos < L;
    const int pc = pok[j] ? pos : 0;
    const int row = pc / V;
    const int goff = (row * a.stride) * V + (pc - row * V);
    voff[j] = pok[j] ? (int)(((size_t)n * Ci + half) * cstride + goff) * 4 : OOB_OFF;
  }
  for (int i = tid; i < Ci; i += PW_NT) {
    f32x4 p;
    p.x = aff1 ? a.s1[i] : 1.f;
    p.y = aff1 ? a.h1[i] : 0.f;
    p.z = aff2 ? a.s2[i] : 1.f;
    p.w = aff2 ? a.h2[i] : 0.f;
    Ps4[i] = p;
  }
  const size_t xbytes = (size_t)a.n * Ci * cstride * 4;
  const __amdgpu_buffer_rsrc_t r1 = make_rsrc(a.x1, xbytes);
  const __amdgpu_buffer_rsrc_t r2 = make_rsrc(has2 ? a.x2 : a.x1, xbytes);

  f32x16 acc[MT][NW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][j][i] = 0.f;

  const int mode = has2 ? 2 : ((aff1 || a.relu) ? 1 : 0);
  if (NW == 1 && ROLL) {
    if (mode == 0) fwd2_roll<MT, 0>(a, reinterpret_cast<f32x16 (&)[MT][1]>(acc), Ws, Ps4, r1, r2, voff[0], coBase, cstride * 4, tid, half, l31);
    else if (mode == 1) fwd2_roll<MT, 1>(a, reinterpret_cast<f32x16 (&)[MT][1]>(acc), Ws, Ps4, r1, r2, voff[0], coBase, cstride * 4, tid, half, l31);
    else fwd2_roll<MT, 2>(a, reinterpret_cast<f32x16 (&)[MT][1]>(acc), Ws, Ps4, r1, r2, voff[0], coBase, cstride * 4, tid, half, l31);
  } else
  for (int c0 = 0; c0 < Ci; c0 += KW) {
    const int kc = min(KW, Ci - c0);
    __syncthreads();                             // previous chunk's W fully consumed (and Ps written, first time)
    {
      // all 2*MT float4 loads of this thread are issued before the first LDS write (a plain load/store loop gets one
      // HBM round trip per iteration from hipcc: measured 16 us per chunk)
      f32x4 wr[2 * MT];
#pragma unroll
      for (int q = 0; q < 2 * MT; ++q) {
        const int f = tid + PW_NT * q;
        const int rowi = f >> 4, kq = (f & 15) * 4;
        const int co = coBase + rowi, ci = c0 + kq;
        if (wvec) {
          wr[q] = (co < Co && ci < Ci) ? *reinterpret_cast<const f32x4*>(a.w + (size_t)co * Ci + ci)
                                       : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) wr[q][e] = (co < Co && ci + e < Ci) ? a.w[(size_t)co * Ci + ci + e] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 2 * MT; ++q) {
        const int f = tid + PW_NT * q;
        const int rowi = f >> 4, kq = (f & 15) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) Ws[rowi * KWS + kq + e] = wr[q][e];
      }
    }
    __syncthreads();
    if (Ci >= 48) {
      if (mode == 0) fwd2_chunk<MT, NW, 0, F2_UNRW>(a, acc, Ws, Ps4, r1, r2, voff, c0, kc, cstride * 4, half, l31);
      else if (mode == 1) fwd2_chunk<MT, NW, 1, F2_UNRW>(a, acc, Ws, Ps4, r1, r2, voff, c0, kc, cstride * 4, half, l31);
      else fwd2_chunk<MT, NW, 2, F2_UNRW>(a, acc, Ws, Ps4, r1, r2, voff, c0, kc, cstride * 4, half, l31);
    } else {
      if (mode == 0) fwd2_chunk<MT, NW, 0, F2_UNR>(a, acc, Ws, Ps4, r1, r2, voff, c0, kc, cstride * 4, half, l31);
      else if (mode == 1) fwd2_chunk<MT, NW, 1, F2_UNR>(a, acc, Ws, Ps4, r1, r2, voff, c0, kc, cstride * 4, half, l31);
      else fwd2_chunk<MT, NW, 2, F2_UNR>(a, acc, Ws, Ps4, r1, r2, voff, c0, kc, cstride * 4, half, l31);
    }
  }
  __syncthreads();
  // ---- epilogue: bias, coalesced stores; per-channel sum / sum of squares via an LDS transpose of each tile
  //      (a shuffle tree costs 160 ds_bpermute per tile and made the LDS pipe the bottleneck: measured) ----
  float* Tw = lds + wave * (32 * 36);            // per-wave 32x32 tile, row stride 36 (conflict-free b128 row reads)
  double* Ss = reinterpret_cast<double*>(lds + 4 * 32 * 36);   // [4 waves][32][2]; fp64: var = E[z^2]-mean^2 downstream
  const size_t blk = (size_t)bk.n * a.nbx + bk.bx;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    double sv = 0.0, qv = 0.0;                   // fp64 (full rate on CDNA): the variance formula amplifies sum errors
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int pos = (tile0 + j) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mfma_row32(r, half);
        const int co = coBase + 32 * m + row;
        float val = 0.f;
        if (co < Co && pok[j]) {
          val = acc[m][j][r] + (a.bias ? a.bias[co] : 0.f);
          a.z[(size_t)(n * Co + co) * L + pos] = val;
        }
        if (a.stats) Tw[row * 36 + l31] = val;
      }
      if (a.stats) {
        wave_lds_sync();
        const f32x4* rowp = reinterpret_cast<const f32x4*>(Tw + l31 * 36 + half * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = rowp[q];
          sv += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
          qv = fma((double)v.x, (double)v.x, qv); qv = fma((double)v.y, (double)v.y, qv);
          qv = fma((double)v.z, (double)v.z, qv); qv = fma((double)v.w, (double)v.w, qv);
        }
        wave_lds_sync();
      }
    }
    if (a.stats) {
      sv += __shfl_xor(sv, 32, 64);
      qv += __shfl_xor(qv, 32, 64);              // (double overload: two dword shuffles)
      if (half == 0) {
        Ss[(wave * 32 + l31) * 2 + 0] = sv;
        Ss[(wave * 32 + l31) * 2 + 1] = qv;
      }
      __syncthreads();
      if (tid < 32) {
        const int co = coBase + 32 * m + tid;
        if (co < Co) {
          double s4 = 0.0, q4 = 0.0;
#pragma unroll
          for (int w = 0; w < 4; ++w) { s4 += Ss[(w * 32 + tid) * 2]; q4 += Ss[(w * 32 + tid) * 2 + 1]; }
          a.partial[(blk * Co + co) * 2 + 0] = (float)s4;
          a.partial[(blk * Co + co) * 2 + 1] = (float)q4;
        }
      }
      __syncthreads();
    }
  }
}

// zaug[n,c,t] = mean_v z[n,c,t,:] plus the global-joint column's share of the batch statistics.
// One wave per (n,c) plane: coalesced plane read into LDS, lanes = frames for the row means; partial row `n` of the
// extra block holds per-channel sum / sum of squares of zaug.
__global__ __launch_bounds__(64) void k_rowmean_stats(const float* __restrict__ z, float* __restrict__ zaug,
                                                      float* __restrict__ partial, int C, int T, int V) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int L = T * V;
  const float* __restrict__ pz = z + (size_t)plane * L;
  if ((L & 3) == 0) {
    plane_to_lds(pz, lds, L >> 2, lane);
  } else {
#pragma unroll 4
    for (int i = lane; i < L; i += 64) lds[i] = pz[i];
  }
  wave_lds_sync();
  const float invV = 1.f / (float)V;
  double sv = 0.0, qv = 0.0;
  for (int t = lane; t < T; t += 64) {
    float m = 0.f;
    for (int v = 0; v < V; ++v) m += lds[t * V + v];
    m *= invV;
    zaug[(size_t)plane * T + t] = m;
    sv += (double)m;
    qv = fma((double)m, (double)m, qv);
  }
  if (partial) {
    sv = wave_sum_d(sv);
    qv = wave_sum_d(qv);
    if (lane == 0) {
      partial[(size_t)plane * 2 + 0] = (float)sv;       // rows [n][C][2]
      partial[(size_t)plane * 2 + 1] = (float)qv;
    }
  }
}

// dz_eff (n,Co,T,V) materialised for convs with the global-joint column (the dgmstcn branch conv): folds the BN
// statistics terms and the zaug gradient once so that dgrad / wgrad run in their plain (single-stream) mode.
//   dz_eff = gz + A0 + B0*z + (gzaug + A0 + B0*zaug)/V
__global__ __launch_bounds__(64) void k_dz_eff_aug(const float* __restrict__ gz, const float* __restrict__ z,
                                                   const float* __restrict__ gzaug, const float* __restrict__ zaug,
                                                   const float* __restrict__ A0, const float* __restrict__ B0,
                                                   float* __restrict__ out, int C, int T, int V) {
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int c = (int)(plane % C);
  const int L = T * V;
  const float a0 = A0 ? A0[c] : 0.f, b0 = A0 ? B0[c] : 0.f;
  const float invV = 1.f / (float)V;
  const float* __restrict__ pg = gz ? gz + (size_t)plane * L : nullptr;
  const float* __restrict__ pz = z + (size_t)plane * L;
  const float* __restrict__ pga = gzaug ? gzaug + (size_t)plane * T : nullptr;
  const float* __restrict__ pza = zaug + (size_t)plane * T;
  float* __restrict__ po = out + (size_t)plane * L;
#pragma unroll 4
  for (int i = lane; i < L; i += 64) {
    const int t = i / V;
    const float e = (pga ? pga[t] : 0.f) + fmaf(b0, pza[t], a0);
    po[i] = (pg ? pg[i] : 0.f) + fmaf(b0, pz[i], a0) + e * invV;
  }
}

// 16-byte form of k_dz_eff_aug (T*V % 4 == 0): per-frame term e[t] staged in LDS, planes streamed as float4 with
// all loads of a 256-float4 chunk issued before the arithmetic
__global__ __launch_bounds__(64) void k_dz_eff_aug4(const float* __restrict__ gz, const float* __restrict__ z,
                                                    const float* __restrict__ gzaug, const float* __restrict__ zaug,
                                                    const float* __restrict__ A0, const float* __restrict__ B0,
                                                    float* __restrict__ out, int C, int T, int V) {
  extern __shared__ __attribute__((aligned(16))) float lds[];       // [T] e
  const int lane = threadIdx.x;
  const long plane = blockIdx.x;
  const int c = (int)(plane % C);
  const int L = T * V, L4 = L >> 2;
  const float a0 = A0 ? A0[c] : 0.f, b0 = A0 ? B0[c] : 0.f;
  const float invV = 1.f / (float)V;
  for (int t = lane; t < T; t += 64)
    lds[t] = ((gzaug ? gzaug[(size_t)plane * T + t] : 0.f) + fmaf(b0, zaug[(size_t)plane * T + t], a0)) * invV;
  wave_lds_sync();
  const f32x4* __restrict__ pg = gz ? reinterpret_cast<const f32x4*>(gz + (size_t)plane * L) : nullptr;
  const f32x4* __restrict__ pz = reinterpret_cast<const f32x4*>(z + (size_t)plane * L);
  f32x4* __restrict__ po = reinterpret_cast<f32x4*>(out + (size_t)plane * L);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int base = 0; base < L4; base += 256) {
    f32x4 g4[4], z4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = base + q * 64 + lane;
      const bool ok = i < L4;
      g4[q] = (ok && pg) ? pg[i] : zero4;
      z4[q] = ok ? pz[i] : zero4;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = base + q * 64 + lane;
      if (i >= L4) break;
      const float gg[4] = {g4[q].x, g4[q].y, g4[q].z, g4[q].w};
      const float zz[4] = {z4[q].x, z4[q].y, z4[q].z, z4[q].w};
      int t, v;
      divmod_small(4 * i, V, invV, t, v);
      float r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        r[k] = gg[k] + fmaf(b0, zz[k], a0) + lds[t];
        if (++v == V) { v = 0; ++t; }
      }
      po[i] = f32x4{r[0], r[1], r[2], r[3]};
    }
  }
}

// Batch statistics -> BN affine.  partial [nblk][C][2] (fp32 block sums) reduced in fp64: one 256-thread workgroup
// per 32 channels (8 row-slices x 32 channels, coalesced 256-B rows), then a cross-slice LDS reduction.
//   mean, var (biased) saved for backward; scale = gamma*rsqrt(var+eps); shift = beta - mean*scale.
// gamma/beta NULL -> identity affine parameters; channels >= c_affine get the identity affine (1, 0).
__global__ __launch_bounds__(1024) void k_bn_finalize(const float* __restrict__ partial, int nblk, int C, double count,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, float* __restrict__ mean_out,
                                                     float* __restrict__ var_out, float* __restrict__ scale_out,
                                                     float* __restrict__ shift_out, int c_affine) {
  // 8 channels x 128 row slices per block: the kernel is a latency chain over the partial rows (up to ~1800 of them),
  // so the rows are spread over many threads and C/8 blocks rather than walked by 32 slices in C/32 blocks (bn_jobs.h)
  __shared__ double red[16][8][2];
  const BnFinJob J = {partial, gamma, beta, mean_out, var_out, scale_out, shift_out, count, eps, nblk, C, c_affine};
  bn_finalize_block(J, blockIdx.x, red);
}

// several finalize / coefficient jobs in one launch (bn_jobs.h): blocks = the jobs' blocks back to back
__global__ __launch_bounds__(1024) void k_bn_finalize_multi(BnFinTable t) {
  __shared__ double red[16][8][2];
  bnj_dispatch(t, blockIdx.x, [&](const BnFinJob& J, int b) { bn_finalize_block(J, b, red); });
}

__global__ __launch_bounds__(1024) void k_bn_coef_rows_multi(BnCoefTable t) {
  __shared__ double red[16][8][2];
  bnj_dispatch(t, blockIdx.x, [&](const BnCoefJob& J, int b) { bn_coef_rows_block(J, b, red); });
}

// Column sums of a row-major (R, C) fp32 matrix -> out (C), accumulated in fp64 (partial-buffer reductions).
struct ColJob { const float* src; float* out; int R, C, inner, nblk; };

// 16-byte form: a thread owns 4 consecutive columns, a block 128 columns x 32 row slices, so every row visit of a wave is
// two 512-B segments instead of two 128-B ones (the big inputs are the weight-gradient partials: 10-30 MB each, ~0.8 GB
// per DS-STGCN step — these sums are bandwidth-bound, not launch-bound).  C % 4 == 0, rows 16-byte aligned, inner == 1.
__host__ __device__ inline bool colsum_wide(int C, int inner, const float* src) {
  return (C & 3) == 0 && inner == 1 && C >= 128 && ((size_t)src & 15) == 0;
}
__host__ __device__ inline int colsum_blocks(int C, int inner, const float* src) {
  return colsum_wide(C, inner, src) ? (C + 127) / 128 : (C + 31) / 32;
}

__device__ __forceinline__ void colsum_body4(const ColJob& j, int bid, double (*red)[32]) {
  const int cl = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int c = bid * 128 + 4 * cl;
  const int R = j.R, C = j.C;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (c < C) {
    for (int r0 = slice; r0 < R; r0 += 32 * 4) {
      f32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r0 + 32 * q;
        v[q] = r < R ? *reinterpret_cast<const f32x4*>(j.src + (size_t)r * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { s0 += (double)v[q].x; s1 += (double)v[q].y; s2 += (double)v[q].z; s3 += (double)v[q].w; }
    }
  }
  // four passes through the [32][32] staging array, one per column of the quad
  double acc[4] = {s0, s1, s2, s3};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    __syncthreads();
    red[slice][cl] = acc[e];
    __syncthreads();
    if (slice == 0 && c < C) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < 32; ++i) t += red[i][cl];
      j.out[c + e] = (float)t;
    }
  }
}

__device__ __forceinline__ void colsum_body(const ColJob& j, int bid) {
  __shared__ double red[32][32];
  if (colsum_wide(j.C, j.inner, j.src)) { colsum_body4(j, bid, red); return; }
  const int cl = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int c = bid * 32 + cl;
  const int R = j.R, C = j.C;
  double s = 0.0;
  if (c < C) {
    for (int r0 = slice; r0 < R; r0 += 32 * 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = r0 + 32 * q;
        v[q] = r < R ? j.src[(size_t)r * C + c] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) s += (double)v[q];
    }
  }
  red[slice][cl] = s;
  __syncthreads();
  if (slice == 0 && c < C) {
#pragma unroll
    for (int i = 1; i < 32; ++i) s += red[i][cl];
    // inner > 1: columns are (C/inner, inner) pairs and leave transposed, out (inner, C/inner), so that each of the
    // `inner` reductions is a contiguous vector for its consumer
    j.out[j.inner > 1 ? (c % j.inner) * (C / j.inner) + c / j.inner : c] = (float)s;
  }
}

// up to two independent reductions in one launch (the weight-gradient and the input-affine partials of one conv)
__global__ __launch_bounds__(1024) void k_colsum(ColJob a, ColJob b) {
  if ((int)blockIdx.x < a.nblk) colsum_body(a, blockIdx.x);
  else colsum_body(b, blockIdx.x - a.nblk);
}

// any number of independent reductions in one launch: the partial rows of parameter gradients (weights, biases, A,
// alpha / beta, add_coeff) feed nothing but the optimizer, so their ~55 column sums per DS-STGCN step are queued during
// the backward and finished by ONE launch before the gradients are packed.  table: njobs x {src, out, (R, C) packed,
// first block} as 64-bit words (device memory), blocks of a job are contiguous in the grid.
__global__ __launch_bounds__(1024) void k_colsum_multi(const long* __restrict__ table, int njobs) {
  // binary search of the job that owns this block (first-block column is ascending)
  int lo = 0, hi = njobs - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)table[4 * mid + 3] <= b) lo = mid; else hi = mid - 1;
  }
  ColJob j;
  j.src = reinterpret_cast<const float*>(table[4 * lo]);
  j.out = reinterpret_cast<float*>(table[4 * lo + 1]);
  j.R = (int)(table[4 * lo + 2] >> 32);
  j.C = (int)(table[4 * lo + 2] & 0xffffffffL);
  j.inner = 1;
  j.nblk = 0;
  colsum_body(j, b - (int)table[4 * lo + 3]);
}

// the same with the table in the kernel arguments (<= CM_MAXJOBS jobs per launch): no table upload in front of the launch,
// the job lookup runs on scalar loads
constexpr int CM_MAXJOBS = 112;
struct ColTable {
  long w[4 * CM_MAXJOBS];
  int njobs;
};

__global__ __launch_bounds__(1024) void k_colsum_multi_arg(ColTable t) {
  int lo = 0, hi = t.njobs - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)t.w[4 * mid + 3] <= b) lo = mid; else hi = mid - 1;
  }
  ColJob j;
  j.src = reinterpret_cast<const float*>(t.w[4 * lo]);
  j.out = reinterpret_cast<float*>(t.w[4 * lo + 1]);
  j.R = (int)(t.w[4 * lo + 2] >> 32);
  j.C = (int)(t.w[4 * lo + 2] & 0xffffffffL);
  j.inner = 1;
  j.nblk = 0;
  colsum_body(j, b - (int)t.w[4 * lo + 3]);
}

// ------------------------------------------------------------------------------------------------------------
// Backward.
//   dz_eff[co,pos] = g_z + A0[co] + B0[co]*z                     (the BN-statistics terms: d mean / d var of this conv's
//                    (+ (g_zaug[row] + A0 + B0*zaug[row]) / V     own output, folded into the load prologue)
//   dgrad:  dv[ci,pos] = sum_co W[co,ci] dz_eff[co,pos];  through the ReLU mask / affine of the virtual input:
//           d_x1 = dv*m*s1, d_x2 = dv*m*s2, partial sums of dv*m*x1, dv*m, dv*m*x2 (-> d s1, d h1=d h2, d s2)
//   wgrad:  dW[co,ci] = sum_pos dz_eff[co,pos] v[ci,pos];  db[co] = sum_pos dz_eff[co,pos]
// ------------------------------------------------------------------------------------------------------------

struct PwBwdArgs {
  // forward operands
  const float* x1; const float* s1; const float* h1;
  const float* x2; const float* s2; const float* h2;
  int relu;
  const float* w;
  // saved output + incoming gradients
  const float* z; const float* zaug;
  const float* gz; const float* gzaug;
  const float* A0; const float* B0;        // per Co, may be NULL (no batch-stat terms)
  // outputs
  float* dx1; float* dx2; float* ipart;    // ipart [nblk][Ci][3]
  float* dwp; float* dbp;                  // [ksplit][Co][Ci], [ksplit][Co]
  int n, Ci, Co, T, V, Tout, stride, aug, TR, NPpad, vec;
  int chunks_per_split, total_chunks, nb_per_sample, roll, nbx, cc;
  int pstride;                             // floats between consecutive k-splits in dwp / dbp
};

__device__ __forceinline__ float dz_eff_at(const PwBwdArgs& a, size_t gi, size_t gaug, int co, float invV) {
  float d = a.gz ? a.gz[gi] : 0.f;
  if (a.A0) d += fmaf(a.B0[co], a.z[gi], a.A0[co]);
  if (a.aug) {
    float e = a.gzaug ? a.gzaug[gaug] : 0.f;
    if (a.A0) e += fmaf(a.B0[co], a.zaug[gaug], a.A0[co]);
    d = fmaf(e, invV, d);
  }
  return d;
}

// Wave-independent dgrad (product path): same structure as k_pwconv_fwd2 with B = dz_eff straight from HBM
// (gz, z [, gzaug, zaug] via raw buffer loads, coefficients A0/B0 from LDS) and A = W^T from LDS.
template <int NW, bool AUG>
__device__ __forceinline__ void d2_load(float (&g)[F2_UNR][NW], float (&zz)[F2_UNR][NW], float (&ga)[F2_UNR][NW],
                                        float (&za)[F2_UNR][NW], __amdgpu_buffer_rsrc_t rg, __amdgpu_buffer_rsrc_t rz,
                                        __amdgpu_buffer_rsrc_t rga, __amdgpu_buffer_rsrc_t rza, const int (&voff)[NW],
                                        const int (&voffa)[NW], int cbase, int cstride4, int astride4, bool has_g,
                                        bool has_c) {
#pragma unroll
  for (int u = 0; u < F2_UNR; ++u) {
    const int soff = (cbase + 2 * u) * cstride4;
    const int soffa = (cbase + 2 * u) * astride4;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      g[u][j] = has_g ? buf_load(rg, voff[j], soff) : 0.f;
      zz[u][j] = has_c ? buf_load(rz, voff[j], soff) : 0.f;
      if (AUG) {
        ga[u][j] = buf_load(rga, voffa[j], soffa);
        za[u][j] = has_c ? buf_load(rza, voffa[j], soffa) : 0.f;
      }
    }
  }
}

template <int MT, int NW, bool AUG>
__device__ __forceinline__ void d2_compute(f32x16 (&acc)[MT][NW], const float* Ws, const float2* Cs,
                                           const float (&g)[F2_UNR][NW], const float (&zz)[F2_UNR][NW],
                                           const float (&ga)[F2_UNR][NW], const float (&za)[F2_UNR][NW], int kbase,
                                           int kc, int c0, float invV, int half, int l31) {
  constexpr int WS2 = 32 * MT + 1;
#pragma unroll
  for (int u = 0; u < F2_UNR; ++u) {
    const int kl = kbase + 2 * u + half;
    const bool cok = kl < kc;
    float av[MT], bv[NW];
#pragma unroll
    for (int m = 0; m < MT; ++m) av[m] = Ws[kl * WS2 + 32 * m + l31];
    const float2 cf = Cs[cok ? c0 + kl : 0];                  // (A0, B0)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      float d = g[u][j] + fmaf(cf.y, zz[u][j], cf.x);
      if (AUG) d = fmaf(ga[u][j] + fmaf(cf.y, za[u][j], cf.x), invV, d);
      bv[j] = cok ? d : 0.f;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < NW; ++j)
        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[j], acc[m][j], 0, 0, 0);
  }
}

// Rolling whole-chunk prefetch for the data gradient (Co a multiple of 64, no global-joint column): see fwd2_roll.
template <int MT>
__device__ __forceinline__ void dgrad2_roll(const PwBwdArgs& a, f32x16 (&acc)[MT][1], float* Ws, const float2* Cs,
                                            __amdgpu_buffer_rsrc_t rg, __amdgpu_buffer_rsrc_t rz, int voff, int ciBase,
                                            int cstride4, bool has_g, bool has_c, int tid, int half, int l31) {
  constexpr int WS2 = 32 * MT + 1;
  const int Ci = a.Ci, Co = a.Co;
  const int nch = Co / KW;
  float gr[32], zr[32];
  f32x4 wr[2 * MT];
  auto loadW = [&](int c0) {
#pragma unroll
    for (int q = 0; q < 2 * MT; ++q) {
      const int f = tid + PW_NT * q;
      const int rowi = f / (8 * MT), cq = (f - rowi * (8 * MT)) * 4;
      const int ci = ciBase + cq;
      wr[q] = ci < Ci ? *reinterpret_cast<const f32x4*>(a.w + (size_t)(c0 + rowi) * Ci + ci) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    gr[u] = has_g ? buf_load(rg, voff, (2 * u) * cstride4) : 0.f;
    zr[u] = has_c ? buf_load(rz, voff, (2 * u) * cstride4) : 0.f;
  }
  loadW(0);
  for (int c = 0; c < nch; ++c) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 2 * MT; ++q) {
      const int f = tid + PW_NT * q;
      const int rowi = f / (8 * MT), cq = (f - rowi * (8 * MT)) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) Ws[rowi * WS2 + cq + e] = wr[q][e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = c + 1 < nch;
    const int cn = (c + 1) * KW;
    if (more) loadW(cn);
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int kl = 2 * u + half;
      float av[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) av[m] = Ws[kl * WS2 + 32 * m + l31];
      const float2 cf = Cs[c * KW + kl];                      // (A0, B0)
      const float d = gr[u] + fmaf(cf.y, zr[u], cf.x);
      if (more) {
        gr[u] = has_g ? buf_load(rg, voff, (cn + 2 * u) * cstride4) : 0.f;
        zr[u] = has_c ? buf_load(rz, voff, (cn + 2 * u) * cstride4) : 0.f;
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], d, acc[m][0], 0, 0, 0);
    }
  }
}

template <int MT, int NW, bool AUG, bool ROLL, bool PREX>
__global__ __launch_bounds__(PW_NT) void k_pwconv_dgrad2(PwBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WS2 = 32 * MT + 1;
  float* Ws = lds;                               // [KW][WS2]  W[co chunk][ci tile]
  float2* Cs = reinterpret_cast<float2*>(lds + ((KW * WS2 + 3) & ~3));     // [Co] (A0, B0)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const PwBlk bk = pw_decode(a.nbx, a.nbx * a.n, a.cc);
  if (!bk.live) return;
  const int n = bk.n;
  const int ciBase = bk.cz * 32 * MT;
  const int V = a.V, Ci = a.Ci, Co = a.Co;
  const int L = a.Tout * V;
  const int tile0 = (bk.bx * 4 + wave) * NW;
  const bool has_g = a.gz != nullptr, has_c = a.A0 != nullptr;
  const bool wvec = (Ci & 3) == 0;
  const float invV = 1.f / (float)V;

  int voff[NW], voffa[NW], goff[NW];
  bool pok[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int pos = (tile0 + j) * 32 + l31;
    pok[j] = pos < L;
    const int pc = pok[j] ? pos : 0;
    const int row = pc / V;
    goff[j] = (row * a.stride) * V + (pc - row * V);
    voff[j] = pok[j] ? (int)(((size_t)n * Co + half) * L + pc) * 4 : OOB_OFF;
    voffa[j] = pok[j] ? (int)(((size_t)n * Co + half) * a.Tout + row) * 4 : OOB_OFF;
  }
  for (int i = tid; i < Co; i += PW_NT) Cs[i] = has_c ? float2{a.A0[i], a.B0[i]} : float2{0.f, 0.f};
  const size_t zbytes = (size_t)a.n * Co * L * 4, abytes = (size_t)a.n * Co * a.Tout * 4;
  const __amdgpu_buffer_rsrc_t rg = make_rsrc(has_g ? a.gz : a.x1, has_g ? zbytes : 0);
  const __amdgpu_buffer_rsrc_t rz = make_rsrc(has_c ? a.z : a.x1, has_c ? zbytes : 0);
  const __amdgpu_buffer_rsrc_t rga = make_rsrc((AUG && a.gzaug) ? a.gzaug : a.x1, (AUG && a.gzaug) ? abytes : 0);
  const __amdgpu_buffer_rsrc_t rza = make_rsrc((AUG && has_c) ? a.zaug : a.x1, (AUG && has_c) ? abytes : 0);

  f32x16 acc[MT][NW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][j][i] = 0.f;

  // PREX: the epilogue's operands (the forward input at this tile, for the ReLU mask / affine / partial sums) are
  // fetched before the K loop instead of after it, where their latency is fully exposed
  // a plain input (no affine, no ReLU, no second operand — `pre`, `down`, the projections) is not needed at all here
  const bool need_x = a.relu || a.s1 != nullptr || a.x2 != nullptr;
  float xpre[PREX ? MT : 1][PREX ? 16 : 1], ypre[PREX ? MT : 1][PREX ? 16 : 1];
  if (PREX && need_x) {
    const size_t cst = (size_t)a.T * V;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ciBase + 32 * m + mfma_row32(r, half);
        const bool ok = ci < Ci && pok[0];
        const size_t g = ((size_t)n * Ci + (ok ? ci : 0)) * cst + (ok ? goff[0] : 0);
        xpre[m][r] = ok ? a.x1[g] : 0.f;
        ypre[m][r] = (ok && a.x2) ? a.x2[g] : 0.f;
      }
  }
  constexpr int G = 2 * F2_UNR;
  if (NW == 1 && !AUG && ROLL) {
    dgrad2_roll<MT>(a, reinterpret_cast<f32x16 (&)[MT][1]>(acc), Ws, Cs, rg, rz, voff[0], ciBase, L * 4, has_g, has_c, tid,
                    half, l31);
  } else
  for (int c0 = 0; c0 < Co; c0 += KW) {
    const int kc = min(KW, Co - c0);
    __syncthreads();
    {
      f32x4 wr[2 * MT];
#pragma unroll
      for (int q = 0; q < 2 * MT; ++q) {
        const int f = tid + PW_NT * q;
        const int rowi = f / (8 * MT), cq = (f - rowi * (8 * MT)) * 4;
        const int co = c0 + rowi, ci = ciBase + cq;
        if (wvec) {
          wr[q] = (co < Co && ci < Ci) ? *reinterpret_cast<const f32x4*>(a.w + (size_t)co * Ci + ci)
                                       : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) wr[q][e] = (co < Co && ci + e < Ci) ? a.w[(size_t)co * Ci + ci + e] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 2 * MT; ++q) {
        const int f = tid + PW_NT * q;
        const int rowi = f / (8 * MT), cq = (f - rowi * (8 * MT)) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) Ws[rowi * WS2 + cq + e] = wr[q][e];
      }
    }
    __syncthreads();
    const int kpad = (kc + 2 * G - 1) / (2 * G) * (2 * G);
    float ga_[F2_UNR][NW], za_[F2_UNR][NW], gaa[F2_UNR][NW], zaa[F2_UNR][NW];
    float gb_[F2_UNR][NW], zb_[F2_UNR][NW], gab[F2_UNR][NW], zab[F2_UNR][NW];
    d2_load<NW, AUG>(ga_, za_, gaa, zaa, rg, rz, rga, rza, voff, voffa, c0, L * 4, a.Tout * 4, has_g, has_c);
    for (int k0 = 0; k0 < kpad; k0 += 2 * G) {
      d2_load<NW, AUG>(gb_, zb_, gab, zab, rg, rz, rga, rza, voff, voffa, c0 + k0 + G, L * 4, a.Tout * 4, has_g, has_c);
      d2_compute<MT, NW, AUG>(acc, Ws, Cs, ga_, za_, gaa, zaa, k0, kc, c0, invV, half, l31);
      d2_load<NW, AUG>(ga_, za_, gaa, zaa, rg, rz, rga, rza, voff, voffa, c0 + k0 + 2 * G, L * 4, a.Tout * 4, has_g,
                       has_c);
      d2_compute<MT, NW, AUG>(acc, Ws, Cs, gb_, zb_, gab, zab, k0 + G, kc, c0, invV, half, l31);
    }
  }
  __syncthreads();
  // ---- epilogue: ReLU mask / affine of the virtual input, coalesced stores, 3 per-channel partial sums ----
  float* Tw = lds + wave * (32 * 36);
  float* Ss = lds + 4 * 32 * 36;                 // [4 waves][32][3]
  const size_t blk = (size_t)bk.n * a.nbx + bk.bx;
  const size_t cstride = (size_t)a.T * V;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float u0 = 0.f, u1 = 0.f, u2 = 0.f;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      float dvv[16], xav[16], xbv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ciBase + 32 * m + mfma_row32(r, half);
        float dv = 0.f, xa = 0.f, xb = 0.f;
        if (ci < Ci && pok[j]) {
          const size_t g = ((size_t)n * Ci + ci) * cstride + goff[j];
          xa = !need_x ? 0.f : (PREX ? xpre[m][r] : a.x1[g]);
          const float sa = a.s1 ? a.s1[ci] : 1.f;
          float pre = a.s1 ? fmaf(xa, sa, a.h1[ci]) : xa;
          float sb = 1.f;
          if (a.x2) {
            xb = PREX ? ypre[m][r] : a.x2[g];
            if (a.s2) { sb = a.s2[ci]; pre += fmaf(xb, sb, a.h2[ci]); } else pre += xb;
          }
          dv = (!a.relu || pre > 0.f) ? acc[m][j][r] : 0.f;
          a.dx1[g] = dv * sa;
          if (a.dx2) a.dx2[g] = dv * sb;
        }
        dvv[r] = dv; xav[r] = xa; xbv[r] = xb;
      }
      if (a.ipart) {
        // three LDS-transposed row sums: dv*x1, dv, dv*x2
#pragma unroll
        for (int which = 0; which < 3; ++which) {
          if (which == 2 && !a.x2) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float val = which == 0 ? dvv[r] * xav[r] : (which == 1 ? dvv[r] : dvv[r] * xbv[r]);
            Tw[mfma_row32(r, half) * 36 + l31] = val;
          }
          wave_lds_sync();
          const f32x4* rowp = reinterpret_cast<const f32x4*>(Tw + l31 * 36 + half * 16);
          float sacc = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = rowp[q];
            sacc += (v.x + v.y) + (v.z + v.w);
          }
          if (which == 0) u0 += sacc; else if (which == 1) u1 += sacc; else u2 += sacc;
          wave_lds_sync();
        }
      }
    }
    if (a.ipart) {
      u0 += __shfl_xor(u0, 32, 64);
      u1 += __shfl_xor(u1, 32, 64);
      u2 += __shfl_xor(u2, 32, 64);
      if (half == 0) {
        float* q = Ss + (wave * 32 + l31) * 3;
        q[0] = u0; q[1] = u1; q[2] = u2;
      }
      __syncthreads();
      if (tid < 32) {
        const int ci = ciBase + 32 * m + tid;
        if (ci < Ci) {
          float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const float* q = Ss + (w * 32 + tid) * 3;
            v0 += q[0]; v1 += q[1]; v2 += q[2];
          }
          float* o = a.ipart + (blk * Ci + ci) * 3;
          o[0] = v0; o[1] = v1; o[2] = v2;
        }
      }
      __syncthreads();
    }
  }
}

// wgrad: block = (64 co x 64 ci output tile, k-split); K = positions of `chunks_per_split` (sample, TR-row) chunks.
// Thread (row = tid/4, quarter = tid%4) stages positions 16*j + 4*quarter + {0..3}, j < 7, of tile row `row` for both
// operands (dz_eff of channel coBase+row and the virtual input of channel ciBase+row): all loads of a chunk are issued
// before the MFMAs of the previous chunk.  LDS rows have an odd stride so the per-lane column reads
// (A[i=co][k=pos], B[k=pos][j=ci]) are conflict-free.
constexpr int WG_J = 7;            // float4 slots per thread per operand: covers 4*4*7 = 112 >= TR*V positions
constexpr int WG_TR = 4;           // frames per wgrad chunk (multiple of 4 keeps every chunk 16-B aligned)
template <bool VEC, bool HAS2>
__global__ __launch_bounds__(PW_NT) void k_pwconv_wgrad(PwBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int V = a.V, TR = a.TR, Co = a.Co, Ci = a.Ci;
  const int TRV = TR * V;
  const int KP = (TRV + 1) & ~1;                 // positions per chunk, even
  const int LS = KP | 1;                         // odd row stride
  float* Ds = lds;                               // [64][LS] dz_eff
  float* Xs = lds + 64 * LS;                     // [64][LS] virtual input
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  // XCD-aware decode of a 1-D grid when a layer has several (co, ci) tiles: the tiles of one K-split read the same dz /
  // input chunks, so they get consecutive slots on one XCD (blockIdx % 8 labels the XCD group) and share its L2.
  int bxx = blockIdx.x, byy = blockIdx.y, bzz = blockIdx.z;
  if (a.cc > 0) {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int tiles = a.cc * a.nbx;                    // nbx = co tiles, cc = ci tiles
    const int tile = slot % tiles;
    bzz = (slot / tiles) * 8 + xcd;
    bxx = tile % a.nbx;
    byy = tile / a.nbx;
  }
  const int coBase = bxx * 64, ciBase = byy * 64;
  const int split = bzz;
  const int mt = wave >> 1, ntile = wave & 1;
  const float invV = 1.f / (float)V;
  const int row = tid >> 2, quarter = tid & 3;
  const int co = coBase + row, ci = ciBase + row;
  const bool has_g = a.gz != nullptr, has_c = a.A0 != nullptr;
  constexpr bool has2 = HAS2;
  const float A0c = (has_c && co < Co) ? a.A0[co] : 0.f, B0c = (has_c && co < Co) ? a.B0[co] : 0.f;
  const float s1 = (a.s1 && ci < Ci) ? a.s1[ci] : 1.f, h1 = (a.s1 && ci < Ci) ? a.h1[ci] : 0.f;
  const float s2 = (a.s2 && ci < Ci) ? a.s2[ci] : 1.f, h2 = (a.s2 && ci < Ci) ? a.h2[ci] : 0.f;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float dbacc = 0.f;
  f32x4 gr[WG_J], zr[WG_J], xr[WG_J], yr[HAS2 ? WG_J : 1];
  const int ch0 = split * a.chunks_per_split;
  const int ch1 = min(a.total_chunks, ch0 + a.chunks_per_split);

  auto issue = [&](int ch) {
    const int n = ch / a.nb_per_sample;
    const int r0 = (ch - n * a.nb_per_sample) * TR;
    const int nvalid = min(TR, a.Tout - r0) * V;
    const size_t gzb = ((size_t)(n * Co + co) * a.Tout + r0) * V;
    const size_t gxb = ((size_t)(n * Ci + ci) * a.T + (size_t)r0 * a.stride) * V;
#pragma unroll
    for (int j = 0; j < WG_J; ++j) {
      const int p = 16 * j + 4 * quarter;
      if (VEC) {
        if (p < nvalid) {
          if (co < Co) {
            if (has_g) gr[j] = *reinterpret_cast<const f32x4*>(a.gz + gzb + p);
            if (has_c) zr[j] = *reinterpret_cast<const f32x4*>(a.z + gzb + p);
          }
          if (ci < Ci) {
            xr[j] = *reinterpret_cast<const f32x4*>(a.x1 + gxb + p);
            if (HAS2) yr[HAS2 ? j : 0] = *reinterpret_cast<const f32x4*>(a.x2 + gxb + p);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pe = p + e;
          if (pe < nvalid) {
            if (co < Co) {
              if (has_g) gr[j][e] = a.gz[gzb + pe];
              if (has_c) zr[j][e] = a.z[gzb + pe];
            }
            if (ci < Ci) {
              const int rl = (int)(((float)pe + 0.5f) * invV);
              const size_t g = gxb + (size_t)rl * a.stride * V + (pe - rl * V);
              xr[j][e] = a.x1[g];
              if (HAS2) yr[HAS2 ? j : 0][e] = a.x2[g];
            }
          }
        }
      }
    }
  };

  auto commit = [&](int ch) {
    const int n = ch / a.nb_per_sample;
    const int r0 = (ch - n * a.nb_per_sample) * TR;
    const int nvalid = min(TR, a.Tout - r0) * V;
    float dsum = 0.f;
#pragma unroll
    for (int j = 0; j < WG_J; ++j) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int pe = 16 * j + 4 * quarter + e;
        if (pe < KP) {
          float d = 0.f, x = 0.f;
          if (pe < nvalid) {
            if (co < Co) {
              d = has_g ? gr[j][e] : 0.f;
              if (has_c) d += fmaf(B0c, zr[j][e], A0c);
              if (a.aug) {
                const int rl = (int)(((float)pe + 0.5f) * invV);
                const size_t ga = (size_t)(n * Co + co) * a.Tout + r0 + rl;
                float ea = a.gzaug ? a.gzaug[ga] : 0.f;
                if (has_c) ea += fmaf(B0c, a.zaug[ga], A0c);
                d = fmaf(ea, invV, d);
              }
            }
            if (ci < Ci) x = virt1(xr[j][e], HAS2 ? yr[HAS2 ? j : 0][e] : 0.f, has2, s1, h1, s2, h2, a.relu);
          }
          Ds[row * LS + pe] = d;
          Xs[row * LS + pe] = x;
          dsum += d;
        }
      }
    }
    dsum += __shfl_xor(dsum, 1, 64);
    dsum += __shfl_xor(dsum, 2, 64);
    dbacc += dsum;
  };

  if (ch0 < ch1) issue(ch0);
  for (int ch = ch0; ch < ch1; ++ch) {
    commit(ch);
    __syncthreads();
    if (ch + 1 < ch1) issue(ch + 1);
    for (int kk = 0; kk < KP; kk += 2) {
      const float av = Ds[(32 * mt + l31) * LS + kk + half];
      const float bv = Xs[(32 * ntile + l31) * LS + kk + half];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // D[i=co][j=ci]
  const int cio = ciBase + 32 * ntile + l31;
  if (cio < Ci) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int coo = coBase + 32 * mt + mfma_row32(r, half);
      if (coo < Co) a.dwp[(size_t)split * a.pstride + (size_t)coo * Ci + cio] = acc[r];
    }
  }
  if (byy == 0 && quarter == 0 && co < Co) a.dbp[(size_t)split * a.pstride + co] = dbacc;
}

// BN backward coefficients of a conv whose batch statistics feed a deferred affine (scale, shift):
//   d gamma = r (g_scale - mean g_shift), d beta = g_shift, A0 = dmean/M - 2 dvar mean/M, B0 = 2 dvar/M
//   with dmean = -g_shift*gamma*r, dvar = -0.5 r^3 gamma (g_scale - mean g_shift).
__global__ __launch_bounds__(64) void k_bn_bwd_coef(const float* __restrict__ g_scale,
                                                    const float* __restrict__ g_shift,
                                                    const float* __restrict__ mean, const float* __restrict__ var,
                                                    const float* __restrict__ gamma, float eps, double count, int C,
                                                    int c_affine, float* __restrict__ dgamma,
                                                    float* __restrict__ dbeta, float* __restrict__ A0,
                                                    float* __restrict__ B0) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  if (c >= c_affine) { dgamma[c] = 0.f; dbeta[c] = 0.f; A0[c] = 0.f; B0[c] = 0.f; return; }
  const double gs = g_scale ? (double)g_scale[c] : 0.0, gh = g_shift ? (double)g_shift[c] : 0.0;
  const double mu = mean[c], r = 1.0 / sqrt((double)var[c] + (double)eps);
  const double g = gamma ? (double)gamma[c] : 1.0;
  const double t = gs - mu * gh;
  dgamma[c] = (float)(r * t);
  dbeta[c] = (float)gh;
  const double dmean = -gh * g * r;
  const double dvar = -0.5 * r * r * r * g * t;
  A0[c] = (float)((dmean - 2.0 * dvar * mu) / count);
  B0[c] = (float)(2.0 * dvar / count);
}

// The same coefficients straight from a consumer's partial rows: g_scale[c] = sum_r part[r][c][i_ds], g_shift[c] = sum_r
// part[r][c][i_dh] (fp64, rows in order), optionally ADDED to what another consumer of the same BatchNorm already wrote —
// the coefficients are linear in (g_scale, g_shift).  One launch instead of a column sum + k_bn_bwd_coef per consumer.
// Block = 8 channels x 128 row slices (as k_bn_finalize: the launch is a latency chain over up to ~1600 partial rows, so the
// rows are spread over many threads and C / 8 blocks: 6.6 -> 4.x us per launch, 55 launches per DS-STGCN step).
__global__ __launch_bounds__(1024) void k_bn_coef_rows(const float* __restrict__ part, int R, int C, int k, int ids, int idh,
                                                       const float* __restrict__ mean, const float* __restrict__ var,
                                                       const float* __restrict__ gamma, float eps, double count,
                                                       int c_affine, float* __restrict__ coef, int accumulate) {
  __shared__ double red[16][8][2];
  const BnCoefJob J = {part, mean, var, gamma, coef, count, eps, R, C, k, ids, idh, c_affine, accumulate};
  bn_coef_rows_block(J, blockIdx.x, red);
}

}  // namespace

int g_pw_maxmt = 2;      // measured (round-1 ablation): small per-wave tiles + more resident waves win
int g_pw4 = 15;           // bit 0: wide-load forward (pw4.hip), bit 1: wide-load data gradient, bit 2: blocked wgrad (wgrad.hip)

// pw4.hip (internal linkage across the library's objects, not exported)
__attribute__((visibility("hidden"))) int dsgcn_p4_tuning(int key, int value);
__attribute__((visibility("hidden"))) int dsgcn_p4_groups(int n, int K, int M, int L, int epi);
__attribute__((visibility("hidden"))) int dsgcn_wg2_tuning(int key, int value);
__attribute__((visibility("hidden"))) int dsgcn_p4_fwd(const float* x1, const float* s1, const float* h1,
                                                        const float* x2, const float* s2, const float* h2, int relu,
                                                        const float* w, const float* bias, float* z, float* partial,
                                                        int n, int Ci, int Co, int L, hipStream_t st, const void* ws);
__attribute__((visibility("hidden"))) size_t dsgcn_p4_ws_bytes(int n, int Ci, int Co, int L);
__attribute__((visibility("hidden"))) int dsgcn_bwd64_tuning(int value);
__attribute__((visibility("hidden"))) int dsgcn_p4_wsplit(const float* w, int Ci, int Co, void* out, hipStream_t st);
__attribute__((visibility("hidden"))) int dsgcn_p4_wsplit_multi(const float* const* w, void* const* out, const int* Ci,
                                                                const int* Co, int njobs, hipStream_t st);
__attribute__((visibility("hidden"))) int dsgcn_p4_dgrad(const float* x1, const float* s1, const float* h1,
                                                          const float* x2, const float* s2, const float* h2, int relu,
                                                          const float* w, const float* z, const float* gz,
                                                          const float* A0, const float* B0, float* dx1, float* dx2,
                                                          float* ipart, int n, int Ci, int Co, int L, hipStream_t st,
                                                          const void* ws);

__attribute__((visibility("hidden"))) int dsgcn_p4_group_ok(int n, int Ci, int Co, int L);
__attribute__((visibility("hidden"))) int dsgcn_p4_fwd_group(const float* const* x1, const float* const* s1,
                                                              const float* const* h1, int relu, const float* const* w,
                                                              float* const* z, int ng, int n, int Ci, int Co, int L,
                                                              hipStream_t st);
__attribute__((visibility("hidden"))) int dsgcn_p4_dgrad_group(const float* const* x1, const float* const* s1,
                                                                const float* const* h1, int relu, const float* const* w,
                                                                const float* const* gz, float* const* dx1, float* const* ipart,
                                                                int ng, int n, int Ci, int Co, int L, hipStream_t st);

// bwd64.hip
__attribute__((visibility("hidden"))) int dsgcn_bwd64_splits(int n, int Ci, int Co, int L);
__attribute__((visibility("hidden"))) int dsgcn_bwd64(const float* x1, const float* s1, const float* h1, const float* x2,
                                                       const float* s2, const float* h2, int relu, const float* w,
                                                       const float* z, const float* gz, const float* A0, const float* B0,
                                                       float* dx, float* dx2, float* dwp, float* dbp, int pstride,
                                                       float* ipart, int n, int Ci, int Co, int L, hipStream_t st);

// wgrad.hip
__attribute__((visibility("hidden"))) int dsgcn_wg2_splits(int n, int Ci, int Co, int L);
__attribute__((visibility("hidden"))) int dsgcn_wg2(const float* x1, const float* s1, const float* h1, const float* x2,
                                                     const float* s2, const float* h2, int relu, const float* z,
                                                     const float* gz, const float* A0, const float* B0, float* dwp,
                                                     float* dbp, int pstride, int n, int Ci, int Co, int L,
                                                     hipStream_t st, const BnCoefTable* jobs, int ngroup = 0,
                                                     const float* const* gx1 = nullptr, const float* const* gs1 = nullptr,
                                                     const float* const* gh1 = nullptr, const float* const* ggz = nullptr,
                                                     float* const* gdwp = nullptr, float* const* gdbp = nullptr);

// workgroup rows of the statistics / input-affine partial buffers for a (K -> M) mix over n planes of L positions
static int pw_conv_rows(int n, int K, int M, int T, int V, int stride, int which) {
  const int Tout = (T + stride - 1) / stride;
  const int L = Tout * V;
  if (stride == 1 && (g_pw4 & which)) {
    const int g = dsgcn_p4_groups(n, K, M, L, which == 2 ? 1 : 0);
    if (g > 0) return g;
  }
  return n * ((L + 4 * 32 - 1) / (4 * 32));
}

extern "C" {

#ifdef DSGCN_LAB
int dsgcn_pwconv_tuning(int key, int value) {
  if (key == 1) { g_pw_maxmt = value; return 0; }
  if (key == 2) { g_pw_roll = value; return 0; }
  if (key == 3) { g_pw4 = value; return 0; }
  if (key >= 4 && key <= 6) return dsgcn_p4_tuning(key - 4, value);
  if (key >= 7 && key <= 9) return dsgcn_wg2_tuning(key - 7, value);
  if (key >= 10 && key <= 12) return dsgcn_p4_tuning(key - 7, value);      // GEMM form: bits, min K, min plane
  if (key == 13) return dsgcn_wg2_tuning(3, value);                        // weight gradient: bf16 terms on / off
  if (key == 14) return dsgcn_p4_tuning(6, value);                         // pre-split weight image: 0 off, 1 k_pwg2, 2 k_pwg3
  if (key >= 15 && key <= 17) return dsgcn_wg2_tuning(key - 11, value);    // wide weight gradient (k_wg3) on / off, split target, co tile
  if (key == 18) return dsgcn_bwd64_tuning(value);                         // one-pass narrow backward: 1 bf16 terms, 0 fp32 MFMA
  if (key == 19) return dsgcn_p4_tuning(7, value);                         // tiny-plane launches: K split over the four waves on / off
  if (key == 22) return dsgcn_p4_tuning(9, value);                         // wide convs on 128-row workgroups (lab)
  if (key == 20) return dsgcn_p4_tuning(8, value);                         // under-filled launches: one row tile per wave on / off
  return DSGCN_EINVAL;
}
#endif

int dsgcn_pwconv_plan(int Tout, int V, int aug, int* TR, int* NPpad, int* nblk_per_sample) {
  int tr = Tout >= 32 ? 8 : 4;
  if (tr > Tout) tr = Tout;
  int np = tr * (V + (aug ? 1 : 0));
  while (np > 256 && tr > 1) { tr >>= 1; np = tr * (V + (aug ? 1 : 0)); }
  if (np > 256) return DSGCN_EUNSUPPORTED;
  *TR = tr;
  *NPpad = (np + 31) & ~31;
  *nblk_per_sample = (Tout + tr - 1) / tr;
  return 0;
}

// Forward.  x2/s1/h1/s2/h2/bias/zaug/partial may be NULL as allowed by the flags.  partial: (n*nblk_per_sample, Co, 2).
// Rows of the `partial` buffer the forward writes for a given shape (conv blocks [+ global-joint rows when aug]).
int dsgcn_pwconv_partial_rows(int n, int Ci, int Co, int T, int V, int stride, int aug) {
  int rows = pw_conv_rows(n, Ci, Co, T, V, stride, 1);
  if (aug) rows += n;
  return rows;
}

// Forward.  x2/s1/h1/s2/h2/bias/zaug/partial may be NULL as allowed by the flags.
// partial: (dsgcn_pwconv_partial_rows(...), Co, 2).
// ws: NULL, or the pre-split weight image of dsgcn_pwconv_wsplit (same w): the GEMM-form launches then skip the split of
// the weight tile (csrc/pw4.hip, k_pwg2).
int dsgcn_pwconv_fwd_ws(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                        const float* h2, int relu, const float* w, const float* bias, float* z, float* zaug,
                        float* partial, int n, int Ci, int Co, int T, int V, int stride, int aug, int stats,
                        const void* ws, void* stream) {
  if (!x1 || !w || !z || n <= 0 || Ci <= 0 || Co <= 0 || T <= 0 || V <= 0 || stride <= 0) return DSGCN_EINVAL;
  if ((aug && !zaug) || (stats && !partial) || (s1 && !h1) || (s2 && !h2)) return DSGCN_EINVAL;
  const int Tout = (T + stride - 1) / stride;
  PwArgs a = {};
  a.x1 = x1; a.s1 = s1; a.h1 = h1; a.x2 = x2; a.s2 = s2; a.h2 = h2; a.relu = relu;
  a.w = w; a.bias = bias; a.z = z; a.zaug = zaug; a.partial = partial;
  a.n = n; a.Ci = Ci; a.Co = Co; a.T = T; a.V = V; a.Tout = Tout; a.stride = stride; a.aug = aug;
  a.stats = stats;
  a.roll = ((g_pw_roll & 1) && Ci % KW == 0) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  const int L = Tout * V;
  const int conv_rows = pw_conv_rows(n, Ci, Co, T, V, stride, 1);
  int fast = 0;
  if (stride == 1 && (g_pw4 & 1)) {
    fast = dsgcn_p4_fwd(x1, s1, h1, x2, s2, h2, relu, w, bias, z, stats ? partial : nullptr, n, Ci, Co, L, st, ws);
    if (fast != 0 && fast != 1) return fast;
  }
  if (!fast) {
  // the first-generation kernels address the whole tensor with 32-bit byte offsets (the wide-load kernels above carry
  // the sample base in the buffer resource): refuse what they cannot address instead of wrapping silently
  if ((long)n * Ci * T * V * 4 >= (1L << 31) - 64 || (long)n * Co * T * V * 4 >= (1L << 31) - 64) return DSGCN_EUNSUPPORTED;
  const int mtiles = (Co + 31) / 32;
  const int MT = mtiles >= g_pw_maxmt ? g_pw_maxmt : mtiles;
  const int NW = 1;
  const int nbx = (L + 4 * NW * 32 - 1) / (4 * NW * 32);
  a.nbx = nbx;
  a.cc = (mtiles + MT - 1) / MT;
  dim3 grid((unsigned)(((long)nbx * n + 7) / 8 * 8 * a.cc));
  size_t ldsf = (size_t)32 * MT * KWS + 4 + (size_t)4 * Ci;
  if (ldsf < (size_t)4 * 32 * 36 + 512) ldsf = (size_t)4 * 32 * 36 + 512;
  const size_t lds = ldsf * sizeof(float);
#define PW_FWD2(MTV)                                                                                  \
  if (a.roll) hipLaunchKernelGGL((k_pwconv_fwd2<MTV, 1, true>), grid, dim3(PW_NT), lds, st, a);       \
  else hipLaunchKernelGGL((k_pwconv_fwd2<MTV, 1, false>), grid, dim3(PW_NT), lds, st, a)
  switch (MT) {
    case 1: PW_FWD2(1); break;
    case 2: PW_FWD2(2); break;
    case 3: PW_FWD2(3); break;
    default: PW_FWD2(4); break;
  }
#undef PW_FWD2
  DSGCN_LAUNCH_CHECK();
  }
  if (aug) {
    const size_t l2 = (size_t)Tout * V * sizeof(float);
    if (l2 > 64 * 1024) return DSGCN_EUNSUPPORTED;
    hipLaunchKernelGGL(k_rowmean_stats, dim3((unsigned)((long)n * Co)), dim3(64), l2, st, z, zaug,
                       stats ? partial + (size_t)conv_rows * Co * 2 : (float*)nullptr, Co, Tout, V);
    DSGCN_LAUNCH_CHECK();
  }
  return 0;
}

int dsgcn_pwconv_fwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                     const float* h2, int relu, const float* w, const float* bias, float* z, float* zaug,
                     float* partial, int n, int Ci, int Co, int T, int V, int stride, int aug, int stats,
                     void* stream) {
  return dsgcn_pwconv_fwd_ws(x1, s1, h1, x2, s2, h2, relu, w, bias, z, zaug, partial, n, Ci, Co, T, V, stride, aug, stats,
                             nullptr, stream);
}

// Bytes of the pre-split weight image (three bf16 terms of W and of W^T, zero-padded to whole tiles) that the forward /
// data gradient of this conv can use; 0 = neither takes the GEMM form (narrow convs, strided convs, tiny planes).
size_t dsgcn_pwconv_wsplit_bytes(int n, int Ci, int Co, int T, int V, int stride) {
  if (n <= 0 || Ci <= 0 || Co <= 0 || T <= 0 || V <= 0 || stride != 1 || (g_pw4 & 3) != 3) return 0;
  return dsgcn_p4_ws_bytes(n, Ci, Co, T * V);
}

int dsgcn_pwconv_wsplit(const float* w, int Ci, int Co, void* ws, void* stream) {
  if (!w || !ws || Ci <= 0 || Co <= 0) return DSGCN_EINVAL;
  return dsgcn_p4_wsplit(w, Ci, Co, ws, (hipStream_t)stream);
}

int dsgcn_pwconv_wsplit_multi(const float* const* w, void* const* ws, const int* Ci, const int* Co, int njobs,
                              void* stream) {
  if (!w || !ws || !Ci || !Co || njobs <= 0) return DSGCN_EINVAL;
  for (int j = 0; j < njobs; ++j)
    if (!w[j] || !ws[j] || Ci[j] <= 0 || Co[j] <= 0) return DSGCN_EINVAL;
  return dsgcn_p4_wsplit_multi(w, ws, Ci, Co, njobs, (hipStream_t)stream);
}

int dsgcn_bn_finalize(const float* partial, int nblk, int C, double count, const float* gamma, const float* beta,
                      float eps, float* mean_out, float* var_out, float* scale_out, float* shift_out, int c_affine,
                      void* stream) {
  if (!partial || !mean_out || !var_out || !scale_out || !shift_out || nblk <= 0 || C <= 0) return DSGCN_EINVAL;
  hipLaunchKernelGGL(k_bn_finalize, dim3((unsigned)((C + 7) / 8)), dim3(1024), 0, (hipStream_t)stream, partial, nblk,
                     C, count, gamma, beta, eps, mean_out, var_out, scale_out, shift_out, c_affine);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// Up to three 1x1 convs of ONE shape in one launch each way (host arrays of ngroup device pointers): z_g = W_g . (x1_g * s1_g +
// h1_g), no bias / second stream / statistics, stride 1 — CTR-GCN's three conv4's per unit (gcn.py:655-657, one per subset),
// each of which alone leaves the chip under-filled.  dsgcn_pwconv_group_ok: 1 when the shape takes the grouped form (else
// the caller launches the convs one by one); the data gradient writes dx1_g and the input-scale rows ipart_g
// (dsgcn_pwconv_ipart_rows, or NULL for all).  The weight gradients stay per conv (dsgcn_pwconv_wgrad).
int dsgcn_pwconv_group_ok(int n, int Ci, int Co, int T, int V) {
  if (n <= 0 || Ci <= 0 || Co <= 0 || T <= 0 || V <= 0) return 0;
  return ((g_pw4 & 3) == 3) ? dsgcn_p4_group_ok(n, Ci, Co, T * V) : 0;
}

int dsgcn_pwconv_fwd_group(const float* const* x1, const float* const* s1, const float* const* h1, int relu,
                           const float* const* w, float* const* z, int ngroup, int n, int Ci, int Co, int T, int V,
                           void* stream) {
  if (!x1 || !s1 || !h1 || !w || !z || ngroup < 1 || ngroup > 3 || n <= 0 || Ci <= 0 || Co <= 0 || T <= 0 || V <= 0) return DSGCN_EINVAL;
  for (int g = 0; g < ngroup; ++g)
    if (!x1[g] || !w[g] || !z[g] || ((s1[g] == nullptr) != (h1[g] == nullptr))) return DSGCN_EINVAL;
  if (!dsgcn_pwconv_group_ok(n, Ci, Co, T, V)) return DSGCN_EUNSUPPORTED;
  const int rc = dsgcn_p4_fwd_group(x1, s1, h1, relu, w, z, ngroup, n, Ci, Co, T * V, (hipStream_t)stream);
  return rc == 1 ? 0 : (rc == 0 ? DSGCN_EUNSUPPORTED : rc);
}

int dsgcn_pwconv_dgrad_group(const float* const* x1, const float* const* s1, const float* const* h1, int relu,
                             const float* const* w, const float* const* gz, float* const* dx1, float* const* ipart,
                             int ngroup, int n, int Ci, int Co, int T, int V, void* stream) {
  if (!x1 || !s1 || !h1 || !w || !gz || !dx1 || !ipart || ngroup < 1 || ngroup > 3 || n <= 0 || Ci <= 0 || Co <= 0 || T <= 0 ||
      V <= 0)
    return DSGCN_EINVAL;
  for (int g = 0; g < ngroup; ++g)
    if (!x1[g] || !w[g] || !gz[g] || !dx1[g] || ((s1[g] == nullptr) != (h1[g] == nullptr))) return DSGCN_EINVAL;
  if (!dsgcn_pwconv_group_ok(n, Ci, Co, T, V)) return DSGCN_EUNSUPPORTED;
  const int rc = dsgcn_p4_dgrad_group(x1, s1, h1, relu, w, gz, dx1, ipart, ngroup, n, Ci, Co, T * V, (hipStream_t)stream);
  return rc == 1 ? 0 : (rc == 0 ? DSGCN_EUNSUPPORTED : rc);
}

// the weight gradients of the same group (dwp_g / dbp_g: partial rows as dsgcn_pwconv_wgrad, dsgcn_pwconv_wgrad_splits rows each)
int dsgcn_pwconv_wgrad_group(const float* const* x1, const float* const* s1, const float* const* h1, int relu,
                             const float* const* gz, float* const* dwp, float* const* dbp, int pstride, int ngroup, int n,
                             int Ci, int Co, int T, int V, void* stream) {
  if (!x1 || !s1 || !h1 || !gz || !dwp || !dbp || ngroup < 1 || ngroup > 3 || n <= 0 || Ci <= 0 || Co <= 0 || T <= 0 || V <= 0)
    return DSGCN_EINVAL;
  for (int g = 0; g < ngroup; ++g)
    if (!x1[g] || !gz[g] || !dwp[g] || !dbp[g] || ((s1[g] == nullptr) != (h1[g] == nullptr)) ||
        ((s1[g] == nullptr) != (s1[0] == nullptr)))
      return DSGCN_EINVAL;
  if (pstride < Co * Ci || !(g_pw4 & 4)) return pstride < Co * Ci ? DSGCN_EINVAL : DSGCN_EUNSUPPORTED;
  BnCoefTable none = {};
  const int fast = dsgcn_wg2(x1[0], s1[0], h1[0], nullptr, nullptr, nullptr, relu, nullptr, gz[0], nullptr, nullptr, dwp[0],
                             dbp[0], pstride, n, Ci, Co, T * V, (hipStream_t)stream, &none, ngroup, x1, s1, h1, gz, dwp, dbp);
  return fast == 1 ? 0 : (fast == 0 ? DSGCN_EUNSUPPORTED : fast);
}

// njobs <= 4 finalize jobs (include/dsgcn.h: dsgcn_bn_fin_job, the arguments of dsgcn_bn_finalize as a struct) in ONE
// launch — BatchNorms whose producers are independent of one another and both done (gcn.py:2165-2169 `post` + `down`).
int dsgcn_bn_finalize_multi(const dsgcn_bn_fin_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs <= 0 || njobs > BNJ_MAX) return DSGCN_EINVAL;
  BnFinTable t = {};
  t.n = njobs;
  for (int i = 0; i < njobs; ++i) {
    const dsgcn_bn_fin_job& q = jobs[i];
    t.j[i] = BnFinJob{q.partial, q.gamma, q.beta, q.mean, q.var, q.scale, q.shift, q.count, q.eps, q.nblk, q.C, q.c_affine};
  }
  if (!bnj_fin_ok(t)) return DSGCN_EINVAL;
  hipLaunchKernelGGL(k_bn_finalize_multi, dim3((unsigned)bnj_total_blocks(t)), dim3(BNJ_NT), 0, (hipStream_t)stream, t);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// njobs <= 4 coefficient jobs (dsgcn_bn_coef_job = the arguments of dsgcn_bn_coef_rows) in one launch.
int dsgcn_bn_coef_rows_multi(const dsgcn_bn_coef_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs <= 0 || njobs > BNJ_MAX) return DSGCN_EINVAL;
  BnCoefTable t = {};
  t.n = njobs;
  for (int i = 0; i < njobs; ++i) {
    const dsgcn_bn_coef_job& q = jobs[i];
    t.j[i] = BnCoefJob{q.part, q.mean, q.var, q.gamma, q.coef, q.count, q.eps, q.R, q.C, q.k, q.i_ds, q.i_dh, q.c_affine,
                       q.accumulate};
  }
  if (!bnj_coef_ok(t)) return DSGCN_EINVAL;
  hipLaunchKernelGGL(k_bn_coef_rows_multi, dim3((unsigned)bnj_total_blocks(t)), dim3(BNJ_NT), 0, (hipStream_t)stream, t);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// out[c] = sum_r src[r, c]  (fp64 accumulation).
int dsgcn_colsum(const float* src, int R, int C, float* out, void* stream) {
  if (!src || !out || R <= 0 || C <= 0) return DSGCN_EINVAL;
  ColJob a{src, out, R, C, 1, colsum_blocks(C, 1, src)}, b{nullptr, nullptr, 0, 0, 1, 0};
  hipLaunchKernelGGL(k_colsum, dim3((unsigned)a.nblk), dim3(1024), 0, (hipStream_t)stream, a, b);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// Same, with the result transposed: src (R, C/inner, inner) -> out (inner, C/inner).
int dsgcn_colsum_t(const float* src, int R, int C, int inner, float* out, void* stream) {
  if (!src || !out || R <= 0 || C <= 0 || inner <= 0 || C % inner) return DSGCN_EINVAL;
  ColJob a{src, out, R, C, inner, (C + 31) / 32}, b{nullptr, nullptr, 0, 0, 1, 0};
  hipLaunchKernelGGL(k_colsum, dim3((unsigned)a.nblk), dim3(1024), 0, (hipStream_t)stream, a, b);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// Two reductions (as dsgcn_colsum_t each; inner = 1 for the plain form) in one launch.
int dsgcn_colsum2(const float* src_a, int Ra, int Ca, int inner_a, float* out_a, const float* src_b, int Rb, int Cb,
                  int inner_b, float* out_b, void* stream) {
  if (!src_a || !out_a || !src_b || !out_b || Ra <= 0 || Ca <= 0 || Rb <= 0 || Cb <= 0 || inner_a <= 0 ||
      inner_b <= 0 || Ca % inner_a || Cb % inner_b)
    return DSGCN_EINVAL;
  ColJob a{src_a, out_a, Ra, Ca, inner_a, colsum_blocks(Ca, inner_a, src_a)}, b{src_b, out_b, Rb, Cb, inner_b, colsum_blocks(Cb, inner_b, src_b)};
  hipLaunchKernelGGL(k_colsum, dim3((unsigned)(a.nblk + b.nblk)), dim3(1024), 0, (hipStream_t)stream, a, b);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// blocks a job of C columns takes in dsgcn_colsum_multi's grid (the caller lays the jobs' first blocks out with it)
int dsgcn_colsum_blocks(const float* src, int C) { return colsum_blocks(C, 1, src); }

// table (device, njobs x 4 int64): {src pointer, out pointer, (R << 32) | C, first block}; nblocks = sum of dsgcn_colsum_blocks.
int dsgcn_colsum_multi(const long* table, int njobs, int nblocks, void* stream) {
  if (!table || njobs <= 0 || nblocks <= 0) return DSGCN_EINVAL;
  hipLaunchKernelGGL(k_colsum_multi, dim3((unsigned)nblocks), dim3(1024), 0, (hipStream_t)stream, table, njobs);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// The same from a HOST table (njobs x 4 int64, layout as above): the jobs ride in the kernel arguments, CM_MAXJOBS per
// launch (first-block columns are rebased per launch).
int dsgcn_colsum_multi_host(const long* table, int njobs, void* stream) {
  if (!table || njobs <= 0) return DSGCN_EINVAL;
  for (int j0 = 0; j0 < njobs; j0 += CM_MAXJOBS) {
    ColTable t;
    t.njobs = njobs - j0 < CM_MAXJOBS ? njobs - j0 : CM_MAXJOBS;
    const long base = table[4 * j0 + 3];
    for (int j = 0; j < t.njobs; ++j) {
      for (int q = 0; q < 3; ++q) t.w[4 * j + q] = table[4 * (j0 + j) + q];
      t.w[4 * j + 3] = table[4 * (j0 + j) + 3] - base;
    }
    const int jl = j0 + t.njobs - 1;
    const float* lsrc = reinterpret_cast<const float*>(table[4 * jl]);
    const long end = table[4 * jl + 3] + colsum_blocks((int)(table[4 * jl + 2] & 0xffffffffL), 1, lsrc);
    hipLaunchKernelGGL(k_colsum_multi_arg, dim3((unsigned)(end - base)), dim3(1024), 0, (hipStream_t)stream, t);
    DSGCN_LAUNCH_CHECK();
  }
  return 0;
}

// Backward, data path.  gz/gzaug/A0/B0/x2/s*/h*/dx2/ipart may be NULL as in the forward.  dx1 (and dx2) are fully
// written (zero rows where stride skips frames).  ipart: (n*nblk_per_sample, Ci, 3) = [sum dv*x1, sum dv, sum dv*x2].
int dsgcn_pwconv_ipart_rows(int n, int Ci, int Co, int T, int V, int stride) {
  return pw_conv_rows(n, Co, Ci, T, V, stride, 2);
}

int dsgcn_pwconv_dgrad_ws(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                          const float* h2, int relu, const float* w, const float* z, const float* zaug,
                          const float* gz, const float* gzaug, const float* A0, const float* B0, float* dx1, float* dx2,
                          float* ipart, int n, int Ci, int Co, int T, int V, int stride, int aug, const void* ws,
                          void* stream);

int dsgcn_pwconv_dgrad(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                       const float* h2, int relu, const float* w, const float* z, const float* zaug,
                       const float* gz, const float* gzaug, const float* A0, const float* B0, float* dx1, float* dx2,
                       float* ipart, int n, int Ci, int Co, int T, int V, int stride, int aug, void* stream) {
  return dsgcn_pwconv_dgrad_ws(x1, s1, h1, x2, s2, h2, relu, w, z, zaug, gz, gzaug, A0, B0, dx1, dx2, ipart, n, Ci, Co, T,
                               V, stride, aug, nullptr, stream);
}

int dsgcn_pwconv_dgrad_ws(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                          const float* h2, int relu, const float* w, const float* z, const float* zaug,
                          const float* gz, const float* gzaug, const float* A0, const float* B0, float* dx1, float* dx2,
                          float* ipart, int n, int Ci, int Co, int T, int V, int stride, int aug, const void* ws,
                          void* stream) {
  if (!x1 || !w || !dx1 || n <= 0 || Ci <= 0 || Co <= 0 || T <= 0 || V <= 0 || stride <= 0) return DSGCN_EINVAL;
  if ((A0 && (!B0 || !z)) || (aug && A0 && !zaug) || (x2 && !dx2)) return DSGCN_EINVAL;
  // dsgcn_pwconv_ipart_rows sizes `ipart` for the wide-load plan, which needs gz: a NULL gz would drop to the scalar
  // kernels and their (larger) row count — refuse instead of writing past the caller's buffer
  if (!gz && ipart) return DSGCN_EINVAL;
  const int Tout = (T + stride - 1) / stride;
  hipStream_t st = (hipStream_t)stream;
  if (stride > 1) {
    hipError_t e = hipMemsetAsync(dx1, 0, (size_t)n * Ci * T * V * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if (dx2) {
      e = hipMemsetAsync(dx2, 0, (size_t)n * Ci * T * V * sizeof(float), st);
      if (e != hipSuccess) return (int)e;
    }
  }
  PwBwdArgs a = {};
  a.x1 = x1; a.s1 = s1; a.h1 = h1; a.x2 = x2; a.s2 = s2; a.h2 = h2; a.relu = relu; a.w = w;
  a.z = z; a.zaug = zaug; a.gz = gz; a.gzaug = gzaug; a.A0 = A0; a.B0 = B0;
  a.dx1 = dx1; a.dx2 = dx2; a.ipart = ipart;
  a.n = n; a.Ci = Ci; a.Co = Co; a.T = T; a.V = V; a.Tout = Tout; a.stride = stride; a.aug = aug;
  a.roll = ((g_pw_roll & 2) && Co % KW == 0 && (Ci & 3) == 0) ? 1 : 0;
  const int L = Tout * V;
  if (stride == 1 && !aug && (g_pw4 & 2)) {
    const int fast = dsgcn_p4_dgrad(x1, s1, h1, x2, s2, h2, relu, w, z, gz, A0, B0, dx1, dx2, ipart, n, Ci, Co, L, st, ws);
    if (fast == 1) return 0;
    if (fast != 0) return fast;
  }
  // the first-generation kernels address the whole tensor with 32-bit byte offsets (the wide-load kernels above carry
  // the sample base in the buffer resource): refuse what they cannot address instead of wrapping silently
  if ((long)n * Ci * T * V * 4 >= (1L << 31) - 64 || (long)n * Co * T * V * 4 >= (1L << 31) - 64) return DSGCN_EUNSUPPORTED;
  const int mtiles = (Ci + 31) / 32;
  const int MT = mtiles >= g_pw_maxmt ? g_pw_maxmt : mtiles;
  const int NW = 1;
  const int nbx = (L + 4 * NW * 32 - 1) / (4 * NW * 32);
  a.nbx = nbx;
  a.cc = (mtiles + MT - 1) / MT;
  dim3 grid((unsigned)(((long)nbx * n + 7) / 8 * 8 * a.cc));
  size_t ldsf = (size_t)KW * (32 * MT + 1) + 4 + (size_t)2 * Co;
  if (ldsf < (size_t)4 * 32 * 36 + 4 * 32 * 3) ldsf = (size_t)4 * 32 * 36 + 4 * 32 * 3;
  const size_t lds = ldsf * sizeof(float);
#define DSGCN_DGRAD(MTv, NWv)                                                                        \
  do {                                                                                               \
    if (aug) hipLaunchKernelGGL((k_pwconv_dgrad2<MTv, NWv, true, false, false>), grid, dim3(PW_NT), lds, st, a);   \
    else if (a.roll) hipLaunchKernelGGL((k_pwconv_dgrad2<MTv, NWv, false, true, false>), grid, dim3(PW_NT), lds, st, a);   \
    else if (g_pw_roll & 4) hipLaunchKernelGGL((k_pwconv_dgrad2<MTv, NWv, false, false, (NWv == 1)>), grid, dim3(PW_NT), lds, st, a);      \
    else hipLaunchKernelGGL((k_pwconv_dgrad2<MTv, NWv, false, false, false>), grid, dim3(PW_NT), lds, st, a);      \
  } while (0)
  switch (MT) {
    case 1: DSGCN_DGRAD(1, 1); break;
    case 2: DSGCN_DGRAD(2, 1); break;
    case 3: DSGCN_DGRAD(3, 1); break;
    default: DSGCN_DGRAD(4, 1); break;
  }
#undef DSGCN_DGRAD
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// k-split plan of the weight gradient: returns the number of splits (size of dim 0 of dwp / dbp).
int dsgcn_pwconv_wgrad_splits(int n, int Ci, int Co, int T, int V, int stride) {
  const int Tout = (T + stride - 1) / stride;
  if (stride == 1 && (g_pw4 & 4)) {
    const int s = dsgcn_wg2_splits(n, Ci, Co, T * V);
    if (s > 0) return s;
  }
  const int TR = Tout >= WG_TR ? WG_TR : Tout;
  const int chunks = n * ((Tout + TR - 1) / TR);
  const int tiles = ((Co + 63) / 64) * ((Ci + 63) / 64);
  int splits = 512 / tiles;
  if (splits < 1) splits = 1;
  if (splits > chunks) splits = chunks;
  const int per = (chunks + splits - 1) / splits;
  return (chunks + per - 1) / per;
}

// Backward, weight path.  Partial sums per k-split: split s writes dW at dwp + s*pstride (Co*Ci floats) and db at
// dbp + s*pstride (Co floats); one dsgcn_colsum over (splits, pstride) rows finishes both when they share a buffer.
// The weight gradient carrying BatchNorm coefficient jobs (dsgcn_jobs.h) of the conv's INPUT BatchNorms: their partial rows
// were written by the data gradient launched before this call, nothing on the critical chain waits for the weight gradient,
// so the jobs ride in its launch as extra workgroups (the blocked kernels) or go out as one launch ahead of it (the
// first-generation kernels).  njobs = 0: dsgcn_pwconv_wgrad.
int dsgcn_pwconv_wgrad_jobs(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                            const float* h2, int relu, const float* z, const float* zaug, const float* gz,
                            const float* gzaug, const float* A0, const float* B0, float* dwp, float* dbp, int pstride, int n,
                            int Ci, int Co, int T, int V, int stride, int aug, const dsgcn_bn_coef_job* jobs, int njobs,
                            void* stream) {
  if (!x1 || !dwp || !dbp || n <= 0 || Ci <= 0 || Co <= 0 || T <= 0 || V <= 0 || stride <= 0) return DSGCN_EINVAL;
  if ((A0 && (!B0 || !z)) || (aug && A0 && !zaug)) return DSGCN_EINVAL;
  if (njobs < 0 || njobs > BNJ_MAX || (njobs > 0 && !jobs)) return DSGCN_EINVAL;
  BnCoefTable jt = {};
  jt.n = njobs;
  for (int i = 0; i < njobs; ++i) {
    const dsgcn_bn_coef_job& q = jobs[i];
    jt.j[i] = BnCoefJob{q.part, q.mean, q.var, q.gamma, q.coef, q.count, q.eps, q.R, q.C, q.k, q.i_ds, q.i_dh, q.c_affine,
                        q.accumulate};
  }
  if (!bnj_coef_ok(jt)) return DSGCN_EINVAL;
  if (stride == 1 && !aug && (g_pw4 & 4)) {
    if (pstride < Co * Ci) return DSGCN_EINVAL;
    const int fast = dsgcn_wg2(x1, s1, h1, x2, s2, h2, relu, z, gz, A0, B0, dwp, dbp, pstride, n, Ci, Co, T * V,
                               (hipStream_t)stream, &jt);
    if (fast == 1) return 0;
    if (fast != 0) return fast;
  }
  if (njobs) {                                     // not hosted: one launch for the jobs, then the weight gradient
    hipLaunchKernelGGL(k_bn_coef_rows_multi, dim3((unsigned)bnj_total_blocks(jt)), dim3(BNJ_NT), 0, (hipStream_t)stream, jt);
    DSGCN_LAUNCH_CHECK();
  }
  // the first-generation kernels address the whole tensor with 32-bit byte offsets (the wide-load kernels above carry
  // the sample base in the buffer resource): refuse what they cannot address instead of wrapping silently
  if ((long)n * Ci * T * V * 4 >= (1L << 31) - 64 || (long)n * Co * T * V * 4 >= (1L << 31) - 64) return DSGCN_EUNSUPPORTED;
  const int Tout = (T + stride - 1) / stride;
  const int TR = Tout >= WG_TR ? WG_TR : Tout;
  if (TR * V > 16 * WG_J) return DSGCN_EUNSUPPORTED;
  PwBwdArgs a = {};
  a.x1 = x1; a.s1 = s1; a.h1 = h1; a.x2 = x2; a.s2 = s2; a.h2 = h2; a.relu = relu;
  a.z = z; a.zaug = zaug; a.gz = gz; a.gzaug = gzaug; a.A0 = A0; a.B0 = B0; a.dwp = dwp; a.dbp = dbp;
  a.pstride = pstride;
  if (pstride < Co * Ci) return DSGCN_EINVAL;
  a.n = n; a.Ci = Ci; a.Co = Co; a.T = T; a.V = V; a.Tout = Tout; a.stride = stride; a.aug = aug; a.TR = TR;
  a.nb_per_sample = (Tout + TR - 1) / TR;
  a.total_chunks = n * a.nb_per_sample;
  a.vec = (stride == 1 && (T * V) % 4 == 0 && (Tout * V) % 4 == 0 && (TR * V) % 4 == 0 &&
           ((Tout % TR) * V) % 4 == 0) ? 1 : 0;
  const int splits = dsgcn_pwconv_wgrad_splits(n, Ci, Co, T, V, stride);
  a.chunks_per_split = (a.total_chunks + splits - 1) / splits;
  const int KP = (TR * V + 1) & ~1;
  const int LS = KP | 1;
  const size_t lds = (size_t)2 * 64 * LS * sizeof(float);
  dim3 grid((unsigned)((Co + 63) / 64), (unsigned)((Ci + 63) / 64), (unsigned)splits);
  a.nbx = a.cc = 0;
  if (grid.x * grid.y > 1 && splits % 8 == 0 && (g_pw_roll & 8)) {
    a.nbx = (int)grid.x; a.cc = (int)grid.y;
    grid = dim3(grid.x * grid.y * (unsigned)splits);
  }
  hipStream_t st = (hipStream_t)stream;
  if (a.vec && x2) hipLaunchKernelGGL((k_pwconv_wgrad<true, true>), grid, dim3(PW_NT), lds, st, a);
  else if (a.vec) hipLaunchKernelGGL((k_pwconv_wgrad<true, false>), grid, dim3(PW_NT), lds, st, a);
  else if (x2) hipLaunchKernelGGL((k_pwconv_wgrad<false, true>), grid, dim3(PW_NT), lds, st, a);
  else hipLaunchKernelGGL((k_pwconv_wgrad<false, false>), grid, dim3(PW_NT), lds, st, a);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_pwconv_wgrad(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                       const float* h2, int relu, const float* z, const float* zaug, const float* gz,
                       const float* gzaug, const float* A0, const float* B0, float* dwp, float* dbp, int pstride, int n,
                       int Ci, int Co, int T, int V, int stride, int aug, void* stream) {
  return dsgcn_pwconv_wgrad_jobs(x1, s1, h1, x2, s2, h2, relu, z, zaug, gz, gzaug, A0, B0, dwp, dbp, pstride, n, Ci, Co, T, V,
                                 stride, aug, nullptr, 0, stream);
}

// out (n,Co,T,V) = gz + A0 + B0*z + (gzaug + A0 + B0*zaug)/V   (gz, gzaug, A0/B0 may be NULL = zero)
int dsgcn_dz_eff_aug(const float* gz, const float* z, const float* gzaug, const float* zaug, const float* A0,
                     const float* B0, float* out, int n, int C, int T, int V, void* stream) {
  if (!z || !zaug || !out || n <= 0 || C <= 0 || T <= 0 || V <= 0) return DSGCN_EINVAL;
  if ((T * V) % 4 == 0 && T <= 8192) {
    hipLaunchKernelGGL(k_dz_eff_aug4, dim3((unsigned)((long)n * C)), dim3(64), (size_t)T * 4, (hipStream_t)stream, gz, z,
                       gzaug, zaug, A0, B0, out, C, T, V);
    DSGCN_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_dz_eff_aug, dim3((unsigned)((long)n * C)), dim3(64), 0, (hipStream_t)stream, gz, z, gzaug, zaug,
                     A0, B0, out, C, T, V);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_bn_bwd_coef(const float* g_scale, const float* g_shift, const float* mean, const float* var,
                      const float* gamma, float eps, double count, int C, int c_affine, float* dgamma, float* dbeta,
                      float* A0, float* B0, void* stream) {
  if (!mean || !var || !dgamma || !dbeta || !A0 || !B0 || C <= 0) return DSGCN_EINVAL;
  hipLaunchKernelGGL(k_bn_bwd_coef, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, (hipStream_t)stream, g_scale,
                     g_shift, mean, var, gamma, eps, count, C, c_affine, dgamma, dbeta, A0, B0);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// coef (4, C) = [d gamma | d beta | A0 | B0] from partial rows part (R, C, k): column i_ds holds a consumer's partial
// sums of d scale, column i_dh of d shift.  accumulate != 0: added to the coefficients already in coef (a second consumer
// of the same deferred BatchNorm).
int dsgcn_bn_coef_rows(const float* part, int R, int C, int k, int i_ds, int i_dh, const float* mean, const float* var,
                       const float* gamma, float eps, double count, int c_affine, float* coef, int accumulate,
                       void* stream) {
  if (!part || !mean || !var || !coef || R <= 0 || C <= 0 || k <= 0 || i_ds < 0 || i_ds >= k || i_dh < 0 || i_dh >= k)
    return DSGCN_EINVAL;
  hipLaunchKernelGGL(k_bn_coef_rows, dim3((unsigned)((C + 7) / 8)), dim3(1024), 0, (hipStream_t)stream, part, R, C, k, i_ds,
                     i_dh, mean, var, gamma, eps, count, c_affine, coef, accumulate);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"


extern "C" {

// Fused backward (data gradient + weight gradient + input-affine partial sums in one pass over gz, z and the input) for
// narrow convs: stride 1, no global-joint column, Ci and Co <= 64, T*V % 4 == 0 (one or two input streams).
// dsgcn_pwconv_bwd_rows: rows of the partial buffers (dwp/dbp as in dsgcn_pwconv_wgrad with this many splits, ipart
// (rows, Ci, 3)); 0 = shape not covered (use dsgcn_pwconv_dgrad + dsgcn_pwconv_wgrad).
int dsgcn_pwconv_bwd_rows(int n, int Ci, int Co, int T, int V, int stride) {
  if (stride != 1 || !(g_pw4 & 8) || n <= 0 || T <= 0 || V <= 0) return 0;
  return dsgcn_bwd64_splits(n, Ci, Co, T * V);
}

int dsgcn_pwconv_bwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2, const float* h2,
                     int relu, const float* w, const float* z, const float* gz, const float* A0, const float* B0,
                     float* dx1, float* dx2, float* ipart, float* dwp, float* dbp, int pstride, int n, int Ci, int Co,
                     int T, int V, void* stream) {
  if (!x1 || !w || !gz || !dx1 || !dwp || !dbp || n <= 0 || Ci <= 0 || Co <= 0 || T <= 0 || V <= 0) return DSGCN_EINVAL;
  if ((A0 && (!B0 || !z)) || (s1 && !h1) || (s2 && !h2) || (x2 && !dx2) || pstride < Co * Ci) return DSGCN_EINVAL;
  const int rc = dsgcn_bwd64(x1, s1, h1, x2, s2, h2, relu, w, z, gz, A0, B0, dx1, dx2, dwp, dbp, pstride, ipart, n, Ci, Co,
                             T * V, (hipStream_t)stream);
  if (rc == 1) return 0;
  return rc == 0 ? DSGCN_EUNSUPPORTED : rc;
}

}  // extern "C"
