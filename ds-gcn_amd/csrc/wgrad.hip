// K-C weight gradient, register-blocked form:  dW[co,ci] = sum_{n,pos} dz_eff[n,co,pos] * v[n,ci,pos],  db[co] = sum dz_eff
// (backward of the 1x1 convs cited in pwconv.hip: gcn.py:2165-2169,2209-2215,2363-2365, tcn.py:379-404,422,427), with
//   dz_eff = gz + A0[co] + B0[co]*z          (BatchNorm-statistics terms of the conv's own output, folded while loading)
//   v      = relu?(x1*s1+h1 (+ x2*s2+h2))    (the forward's virtual input, rebuilt while loading).
// The reduction runs over positions, so both MFMA operands need "lane = channel" fragments: the tiles are staged through
// LDS — coalesced 16-B global loads (8..32 lanes per 128..512-B row segment), the prologue arithmetic in registers, rows
// (any plane length: a row that is not a multiple of 4 long is loaded with dword-aligned 16-B loads and its tail group is
// masked per element) stored with stride KC+2 (= 2*odd: conflict-free ds_read_b64, each read feeds two k-steps) — and every wave owns a
// (TM/2)x(TN/2) block of the workgroup's TMxTN output tile (up to 2x2 MFMA tiles: one LDS read per MFMA instead of the
// two of the first version, and a 128x128 tile reads each operand row once for 128 output columns instead of 64).
// K (positions) is split over workgroups; each split writes its partial dW / db (deterministic: summed by dsgcn_colsum).
#include "common.h"
#include "bn_jobs.h"

namespace {

constexpr int WG_NT = 256;
constexpr int WG_OOB = 0x7ffffff0;

struct Wg2Args {
  const float* x1; const float* s1; const float* h1;
  const float* x2; const float* s2; const float* h2;
  int relu;
  const float* z; const float* gz; const float* A0; const float* B0;
  float* dwp; float* dbp; int pstride;
  int n, Ci, Co, L, cpn, total_chunks, cps, tm_tiles, tn_tiles;
  int main_blocks;                 // workgroups of the weight gradient; the ones past them run the hosted BatchNorm jobs
  BnCoefTable jobs;                // (bn_jobs.h) coefficient jobs of BatchNorms whose rows the DATA gradient before this launch wrote
  // (round 6) up to three weight gradients of ONE shape in a launch (blockIdx.y = which): CTR-GCN's conv4 per subset
  int ngroup;
  struct Grp { const float* x1; const float* s1; const float* h1; const float* gz; float* dwp; float* dbp; } g[3];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 wg_load(__amdgpu_buffer_rsrc_t r, int voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ int wg_row32(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

typedef float f32x2 __attribute__((ext_vector_type(2)));

// B3: the tiles are staged as the three bf16 terms of every value (common.h: exact split, six bf16 MFMAs per 16 positions
// in place of eight fp32 ones at twice the cycles each); rows of KC bf16 + 16 B of pad per term (conflict-free 16-byte
// fragment reads: a lane reads 8 consecutive positions of its channel row).
template <int TM, int TN, int KC, bool HAS2, bool HASC, bool B3 = false>
__global__ __launch_bounds__(WG_NT, 2) void k_wg2(Wg2Args a_) {
  Wg2Args a = a_;
  if (a_.ngroup > 1) {                             // grouped launch: this workgroup's conv
    const Wg2Args::Grp& q = a_.g[blockIdx.y];
    a.x1 = q.x1; a.s1 = q.s1; a.h1 = q.h1; a.gz = q.gz; a.dwp = q.dwp; a.dbp = q.dbp;
  }
  constexpr int LS = KC + 2;
  constexpr int RB = KC * 2 + 16;                 // B3: bytes per row per term
  constexpr int Q = KC / 4;                       // float4 slots per row
  constexpr int JD = TM * Q / WG_NT, JX = TN * Q / WG_NT;
  constexpr int MI = TM / 64, NI = TN / 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ds = lds;                                // [TM][LS] dz_eff
  float* Xs = lds + TM * LS;                      // [TN][LS] virtual input
  constexpr int TF = B3 ? 3 * (TM + TN) * RB / 4 : (TM + TN) * LS;   // floats of tile storage
  char* Db = reinterpret_cast<char*>(lds);        // B3: [3][TM][RB] dz_eff terms
  char* Xb = Db + 3 * TM * RB;                    // B3: [3][TN][RB] virtual-input terms
  f32x2* Cs = reinterpret_cast<f32x2*>(lds + TF);          // [TM] (A0, B0)
  f32x4* Ps = reinterpret_cast<f32x4*>(lds + TF + 2 * TM);  // [TN] (s1, h1, s2, h2)
  if ((int)blockIdx.x >= a.main_blocks) {          // hosted BatchNorm coefficient jobs: the weight gradient is off the critical chain
    bnj_dispatch(a.jobs, (int)blockIdx.x - a.main_blocks,
                 [&](const BnCoefJob& J, int b) { bn_coef_rows_block<WG_NT>(J, b, reinterpret_cast<double (*)[8][2]>(lds)); });
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  // XCD-aware decode: the tiles of one K-split read the same chunks -> consecutive slots of one XCD (blockIdx % 8)
  const int tiles = a.tm_tiles * a.tn_tiles;
  int split, tile;
  if (tiles > 1) {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    tile = slot % tiles;
    split = (slot / tiles) * 8 + xcd;
  } else {
    tile = 0;
    split = blockIdx.x;
  }
  const int ch0 = split * a.cps;
  const int ch1 = min(a.total_chunks, ch0 + a.cps);
  if (ch0 >= ch1) return;
  const int coBase = (tile % a.tm_tiles) * TM, ciBase = (tile / a.tm_tiles) * TN;
  const int Co = a.Co, Ci = a.Ci, L = a.L;
  const int L4 = L * 4;

  for (int i = tid; i < TM; i += WG_NT) {
    const int co = coBase + i;
    Cs[i] = (HASC && co < Co) ? f32x2{a.A0[co], a.B0[co]} : f32x2{0.f, 0.f};
  }
  for (int i = tid; i < TN; i += WG_NT) {
    const int ci = ciBase + i;
    f32x4 p = {1.f, 0.f, 1.f, 0.f};
    if (ci < Ci) {
      if (a.s1) { p.x = a.s1[ci]; p.y = a.h1[ci]; }
      if (a.s2) { p.z = a.s2[ci]; p.w = a.h2[ci]; }
    }
    Ps[i] = p;
  }

  // this thread's staging slots: slot f = tid + 256*j -> (row f / Q, positions 4*(f % Q) ..+3); rows are fixed per thread
  const int col = (tid % Q) * 4;
  const int rowD0 = tid / Q, rowX0 = tid / Q;     // + (256/Q)*j
  constexpr int RSTEP = WG_NT / Q;
  const float lo = a.relu ? 0.f : -__builtin_inff();

  f32x4 gr[JD], zr[HASC ? JD : 1], xr[JX], yr[HAS2 ? JX : 1];
  auto issue = [&](int ch) {
    const int n = ch / a.cpn;
    const int c0 = (ch - n * a.cpn) * KC;
    const bool pv = c0 + col < L;
    const __amdgpu_buffer_rsrc_t rg = wg_rsrc((a.gz ? a.gz : a.x1) + (size_t)(a.gz ? n : 0) * Co * L, a.gz ? Co * L4 : 0);
    const __amdgpu_buffer_rsrc_t rz = wg_rsrc((HASC ? a.z : a.x1) + (size_t)(HASC ? n : 0) * Co * L, HASC ? Co * L4 : 0);
    const __amdgpu_buffer_rsrc_t rx = wg_rsrc(a.x1 + (size_t)n * Ci * L, Ci * L4);
    const __amdgpu_buffer_rsrc_t ry = wg_rsrc((HAS2 ? a.x2 : a.x1) + (size_t)n * Ci * L, Ci * L4);
#pragma unroll
    for (int j = 0; j < JD; ++j) {
      const int co = coBase + rowD0 + RSTEP * j;
      const int voff = (pv && co < Co) ? (co * L + c0 + col) * 4 : WG_OOB;
      gr[j] = wg_load(rg, voff);
      if constexpr (HASC) zr[j] = wg_load(rz, voff);
    }
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const int ci = ciBase + rowX0 + RSTEP * j;
      const int voff = (pv && ci < Ci) ? (ci * L + c0 + col) * 4 : WG_OOB;
      xr[j] = wg_load(rx, voff);
      if constexpr (HAS2) yr[j] = wg_load(ry, voff);
    }
  };

  float dsum[JD];
#pragma unroll
  for (int j = 0; j < JD; ++j) dsum[j] = 0.f;

  auto commit = [&](int ch) {
    const int n = ch / a.cpn;
    const int c0 = (ch - n * a.cpn) * KC;
    const bool pv = c0 + col < L;
    const int nv = L - (c0 + col);                 // valid elements of this thread's float4 (planes need not be x4 long:
#pragma unroll                                     // the tail group then also holds the next row's first elements)
    for (int j = 0; j < JD; ++j) {
      const int row = rowD0 + RSTEP * j;
      const bool ok = pv && coBase + row < Co;
      f32x4 d = gr[j];
      if constexpr (HASC) {
        const f32x2 c = Cs[row];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] += fmaf(c.y, zr[j][e], c.x);
      }
      if (!ok) d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 1; e < 4; ++e) d[e] = e < nv ? d[e] : 0.f;
      dsum[j] += (d.x + d.y) + (d.z + d.w);
      if constexpr (B3) {
        unsigned p0, p1, p2, q0, q1, q2;
        b3_split(d.x, d.y, p0, p1, p2);
        b3_split(d.z, d.w, q0, q1, q2);
        char* dst = Db + row * RB + col * 2;
        *reinterpret_cast<u32x2v*>(dst) = u32x2v{p0, q0};
        *reinterpret_cast<u32x2v*>(dst + TM * RB) = u32x2v{p1, q1};
        *reinterpret_cast<u32x2v*>(dst + 2 * TM * RB) = u32x2v{p2, q2};
      } else {
        f32x2* dst = reinterpret_cast<f32x2*>(Ds + row * LS + col);
        dst[0] = f32x2{d.x, d.y};
        dst[1] = f32x2{d.z, d.w};
      }
    }
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const int row = rowX0 + RSTEP * j;
      const bool ok = pv && ciBase + row < Ci;
      const f32x4 p = Ps[row];
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = fmaf(xr[j][e], p.x, p.y);
        if constexpr (HAS2) t += fmaf(yr[j][e], p.z, p.w);
        v[e] = (ok && e < nv) ? fmaxf(t, lo) : 0.f;
      }
      if constexpr (B3) {
        unsigned p0, p1, p2, q0, q1, q2;
        b3_split(v.x, v.y, p0, p1, p2);
        b3_split(v.z, v.w, q0, q1, q2);
        char* dst = Xb + row * RB + col * 2;
        *reinterpret_cast<u32x2v*>(dst) = u32x2v{p0, q0};
        *reinterpret_cast<u32x2v*>(dst + TN * RB) = u32x2v{p1, q1};
        *reinterpret_cast<u32x2v*>(dst + 2 * TN * RB) = u32x2v{p2, q2};
      } else {
        f32x2* dst = reinterpret_cast<f32x2*>(Xs + row * LS + col);
        dst[0] = f32x2{v.x, v.y};
        dst[1] = f32x2{v.z, v.w};
      }
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

  const int m0 = (wave >> 1) * (TM / 2), n0 = (wave & 1) * (TN / 2);
  const float* Ap = Ds + (m0 + l31) * LS + 2 * half;
  const float* Bp = Xs + (n0 + l31) * LS + 2 * half;

  __syncthreads();                                // Cs / Ps visible
  issue(ch0);
  for (int ch = ch0; ch < ch1; ++ch) {
    commit(ch);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ch + 1 < ch1) issue(ch + 1);
    if constexpr (B3) {
      const char* Ab = Db + (m0 + l31) * RB + 16 * half;
      const char* Bb = Xb + (n0 + l31) * RB + 16 * half;
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        bf16x8 af[MI][3], bf[NI][3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            af[mi][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(Ab + (t * TM + 32 * mi) * RB + 32 * ks));
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            bf[ni][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(Bb + (t * TN + 32 * ni) * RB + 32 * ks));
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            f32x16 c = acc[mi][ni];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][2], bf[ni][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][1], bf[ni][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][0], bf[ni][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][1], bf[ni][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][0], bf[ni][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][0], bf[ni][0], c, 0, 0, 0);
            acc[mi][ni] = c;
          }
      }
    } else {
#pragma unroll
    for (int w = 0; w < Q; ++w) {
      f32x2 av[MI], bv[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) av[mi] = *reinterpret_cast<const f32x2*>(Ap + 32 * mi * LS + 4 * w);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bv[ni] = *reinterpret_cast<const f32x2*>(Bp + 32 * ni * LS + 4 * w);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi].x, bv[ni].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi].y, bv[ni].y, acc[mi][ni], 0, 0, 0);
        }
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // raw barrier: the next chunk's loads stay in flight
  }

  // D[i = co][j = ci] -> partial dW of this split
  float* dw = a.dwp + (size_t)split * a.pstride;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int ci = ciBase + n0 + 32 * ni + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = coBase + m0 + 32 * mi + wg_row32(r, half);
        if (co < Co && ci < Ci) dw[(size_t)co * Ci + ci] = acc[mi][ni][r];
      }
    }
  // db: the Q threads that stage one row hold its pieces
  if (tile / a.tm_tiles == 0) {
#pragma unroll
    for (int j = 0; j < JD; ++j) {
      float s = dsum[j];
#pragma unroll
      for (int off = 1; off < Q; off <<= 1) s += __shfl_xor(s, off, 64);
      const int co = coBase + rowD0 + RSTEP * j;
      if ((tid % Q) == 0 && co < Co) a.dbp[(size_t)split * a.pstride + co] = s;
    }
  }
}

// ---- wide convs (more than 128 channels on either side): the whole dW tile in one workgroup ("wg3") ------------------
// k_wg2<128,128,32,B3> at 256 -> 256 runs 2 x 2 output tiles x 128 K-splits: every dz_eff / input value is split into its
// three bf16 terms TWICE (42 values per MFMA), in lockstep phases (commit, barrier, 48 products, barrier) on one LDS
// buffer — 75 us against a 20 us HBM / matrix roof (profiles/r03).  Here a workgroup of EIGHT waves owns a 256 x 256 (or
// 256 x 128 / 128 x 256) tile: every value is split once (21 per MFMA), the chunk is 16 positions (one MFMA k-step) in a
// DOUBLE-buffered LDS image (rows of 16 bf16 + 16 B pad per term: conflict-free 16-byte fragment reads), chunk c+1 is
// split and written while chunk c is multiplied, ONE barrier per chunk, two chunks of operand loads in flight in ping-pong
// register sets.  One workgroup per CU (150 KB of LDS), ~256 K-splits: one round.
constexpr int W3_NT = 512, W3_KC = 16, W3_RB = 48;

template <int TM, int TN, bool HAS2, bool HASC>
__global__ __launch_bounds__(W3_NT, 2) void k_wg3(Wg2Args a_) {
  Wg2Args a = a_;
  if (a_.ngroup > 1) {                             // grouped launch: this workgroup's conv
    const Wg2Args::Grp& q = a_.g[blockIdx.y];
    a.x1 = q.x1; a.s1 = q.s1; a.h1 = q.h1; a.gz = q.gz; a.dwp = q.dwp; a.dbp = q.dbp;
  }
  constexpr int JD = TM * 4 / W3_NT, JX = TN * 4 / W3_NT;          // float4 slots per thread and chunk
  constexpr int MI = TM / 128, NI = TN / 64;                       // wave tile: (TM/4) x (TN/2) = MI x NI MFMA tiles
  constexpr int BUF = 3 * (TM + TN) * W3_RB;
  // operand chunks in flight ahead of the one being split: 2 (ping-pong register sets) except where 256 accumulator +
  // fragment registers leave room for one set only (two input streams AND a BatchNorm term on the full tile)
  constexpr int DEPTH = (HAS2 && HASC && TM == 256 && TN == 256) ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* B0p = reinterpret_cast<char*>(lds);                        // buffer: [3][TM][RB] dz_eff terms, [3][TN][RB] input terms
  char* B1p = B0p + BUF;
  f32x2* Cs = reinterpret_cast<f32x2*>(B1p + BUF);                 // [TM] (A0, B0)
  f32x4* Ps = reinterpret_cast<f32x4*>(Cs + TM);                   // [TN] (s1, h1, s2, h2)
  if ((int)blockIdx.x >= a.main_blocks) {          // hosted BatchNorm coefficient jobs (see k_wg2)
    bnj_dispatch(a.jobs, (int)blockIdx.x - a.main_blocks,
                 [&](const BnCoefJob& J, int b) { bn_coef_rows_block<W3_NT>(J, b, reinterpret_cast<double (*)[8][2]>(lds)); });
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles = a.tm_tiles * a.tn_tiles;
  int split, tile;
  if (tiles > 1) {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    tile = slot % tiles;
    split = (slot / tiles) * 8 + xcd;
  } else {
    tile = 0;
    split = blockIdx.x;
  }
  const int ch0 = split * a.cps;
  const int ch1 = min(a.total_chunks, ch0 + a.cps);
  if (ch0 >= ch1) return;
  const int coBase = (tile % a.tm_tiles) * TM, ciBase = (tile / a.tm_tiles) * TN;
  const int Co = a.Co, Ci = a.Ci, L = a.L;
  const int L4 = L * 4;
  for (int i = tid; i < TM; i += W3_NT) {
    const int co = coBase + i;
    Cs[i] = (HASC && co < Co) ? f32x2{a.A0[co], a.B0[co]} : f32x2{0.f, 0.f};
  }
  for (int i = tid; i < TN; i += W3_NT) {
    const int ci = ciBase + i;
    f32x4 p = {1.f, 0.f, 1.f, 0.f};
    if (ci < Ci) {
      if (a.s1) { p.x = a.s1[ci]; p.y = a.h1[ci]; }
      if (a.s2) { p.z = a.s2[ci]; p.w = a.h2[ci]; }
    }
    Ps[i] = p;
  }
  // staging slots: slot f = tid + 512*j -> row f / 4, positions 4*(f % 4) ..+3 of the chunk
  const int col = (tid & 3) * 4;
  const int row0 = tid >> 2;                       // + 128*j
  const float lo = a.relu ? 0.f : -__builtin_inff();
  struct Regs { f32x4 g[JD], z[HASC ? JD : 1], x[JX], y[HAS2 ? JX : 1]; };
  Regs RA, RB;
  auto issue = [&](int ch, Regs& r) {
    const int n = ch / a.cpn;
    const int c0 = (ch - n * a.cpn) * W3_KC;
    const bool pv = (ch < ch1) & (c0 + col < L);
    const __amdgpu_buffer_rsrc_t rg = wg_rsrc((a.gz ? a.gz : a.x1) + (size_t)(a.gz ? n : 0) * Co * L, a.gz ? Co * L4 : 0);
    const __amdgpu_buffer_rsrc_t rz = wg_rsrc((HASC ? a.z : a.x1) + (size_t)(HASC ? n : 0) * Co * L, HASC ? Co * L4 : 0);
    const __amdgpu_buffer_rsrc_t rx = wg_rsrc(a.x1 + (size_t)n * Ci * L, Ci * L4);
    const __amdgpu_buffer_rsrc_t ry = wg_rsrc((HAS2 ? a.x2 : a.x1) + (size_t)n * Ci * L, Ci * L4);
#pragma unroll
    for (int j = 0; j < JD; ++j) {
      const int co = coBase + row0 + 128 * j;
      const bool ok = pv & (co < Co);              // (bitwise: a short-circuit turns the load into a branch + vmcnt(0))
      const int voff = ok ? (co * L + c0 + col) * 4 : WG_OOB;
      r.g[j] = wg_load(rg, voff);
      if constexpr (HASC) r.z[j] = wg_load(rz, voff);
    }
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const int ci = ciBase + row0 + 128 * j;
      const bool ok = pv & (ci < Ci);
      const int voff = ok ? (ci * L + c0 + col) * 4 : WG_OOB;
      r.x[j] = wg_load(rx, voff);
      if constexpr (HAS2) r.y[j] = wg_load(ry, voff);
    }
  };
  float dsum[JD];
#pragma unroll
  for (int j = 0; j < JD; ++j) dsum[j] = 0.f;
  // piece j < JD: dz_eff rows; piece JD + j: input rows
  auto commit_piece = [&](int ch, const Regs& r, char* dst, int piece) {
    const int n = ch / a.cpn;
    const int c0 = (ch - n * a.cpn) * W3_KC;
    const bool pv = (ch < ch1) & (c0 + col < L);
    const int nv = L - (c0 + col);                 // valid elements of this thread's float4 (planes need not be x4 long)
    unsigned p0, p1, p2, q0, q1, q2;
    char* base;
    int tstride;
    if (piece < JD) {
      const int j = piece;
      const int row = row0 + 128 * j;
      const bool ok = pv & (coBase + row < Co);
      f32x4 d = r.g[j];
      if constexpr (HASC) {
        const f32x2 c = Cs[row];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] += fmaf(c.y, r.z[j][e], c.x);
      }
      if (!ok) d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 1; e < 4; ++e) d[e] = e < nv ? d[e] : 0.f;
      dsum[j] += (d.x + d.y) + (d.z + d.w);
      b3_split(d.x, d.y, p0, p1, p2);
      b3_split(d.z, d.w, q0, q1, q2);
      base = dst + row * W3_RB + col * 2;
      tstride = TM * W3_RB;
    } else {
      const int j = piece - JD;
      const int row = row0 + 128 * j;
      const bool ok = pv & (ciBase + row < Ci);
      const f32x4 p = Ps[row];
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = fmaf(r.x[j][e], p.x, p.y);
        if constexpr (HAS2) t += fmaf(r.y[j][e], p.z, p.w);
        v[e] = (ok & (e < nv)) ? fmaxf(t, lo) : 0.f;
      }
      b3_split(v.x, v.y, p0, p1, p2);
      b3_split(v.z, v.w, q0, q1, q2);
      base = dst + 3 * TM * W3_RB + row * W3_RB + col * 2;
      tstride = TN * W3_RB;
    }
    *reinterpret_cast<u32x2v*>(base) = u32x2v{p0, q0};
    *reinterpret_cast<u32x2v*>(base + tstride) = u32x2v{p1, q1};
    *reinterpret_cast<u32x2v*>(base + 2 * tstride) = u32x2v{p2, q2};
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;
  const int m0 = (wave >> 1) * (TM / 4), n0 = (wave & 1) * (TN / 2);
  const int aoff = (m0 + l31) * W3_RB + 16 * half;
  const int boff = 3 * TM * W3_RB + (n0 + l31) * W3_RB + 16 * half;

  __builtin_amdgcn_sched_barrier(0);
  issue(ch0, RA);
  if constexpr (DEPTH == 2) issue(ch0 + 1, RB);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();                                 // Cs / Ps visible (and the first loads landed)
#pragma unroll
  for (int pc = 0; pc < JD + JX; ++pc) commit_piece(ch0, RA, B0p, pc);
  __builtin_amdgcn_sched_barrier(0);
  issue(ch0 + DEPTH, RA);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // one chunk: its 16 positions' products on `cur` with chunk ch+1's split (register set r) written to `nxt` in between
  auto step = [&](int ch, const char* cur, char* nxt, Regs& r) {
    bf16x8 af[MI][3];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        af[mi][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(cur + aoff + (t * TM + 32 * mi) * W3_RB));
    // input fragments one position sub-block ahead, in two register sets pinned by scheduling barriers (left alone the
    // compiler hoists all NI x 3 fragment reads to the top: 48 registers it does not have)
    bf16x8 bfs[2][3];
    auto loadB = [&](int ni, bf16x8 (&bf)[3]) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
        bf[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(cur + boff + (t * TN + 32 * ni) * W3_RB));
    };
    loadB(0, bfs[0]);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      bf16x8 (&bf)[3] = bfs[ni & 1];
      if (ni + 1 < NI) loadB(ni + 1, bfs[(ni + 1) & 1]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        f32x16 c = acc[mi][ni];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][2], bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][1], bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][0], bf[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][1], bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][0], bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][0], bf[0], c, 0, 0, 0);
        acc[mi][ni] = c;
      }
      // the next chunk's split, a piece per position sub-block of products (VALU + LDS stores in the MFMA shadow)
#pragma unroll
      for (int pc = ni; pc < JD + JX; pc += NI) commit_piece(ch + 1, r, nxt, pc);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    issue(ch + 1 + DEPTH, r);                      // this set is free again
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // raw barrier: the loads in flight stay in flight
  };
  if constexpr (DEPTH == 2) {
    // (pairs in the loop, an odd last chunk outside it: a conditional second call would make the compiler allow a
    // first -> first path and drain the set it has just refilled)
    int ch = ch0;
    for (; ch + 1 < ch1; ch += 2) {
      step(ch, B0p, B1p, RB);
      step(ch + 1, B1p, B0p, RA);
    }
    if (ch < ch1) step(ch, B0p, B1p, RB);
  } else {
    char* cur = B0p;
    char* nxt = B1p;
    for (int ch = ch0; ch < ch1; ++ch) {
      step(ch, cur, nxt, RA);
      char* t = cur; cur = nxt; nxt = t;
    }
  }

  // D[i = co][j = ci] -> partial dW of this split
  float* dw = a.dwp + (size_t)split * a.pstride;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int ci = ciBase + n0 + 32 * ni + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = coBase + m0 + 32 * mi + wg_row32(r, half);
        if (co < Co && ci < Ci) dw[(size_t)co * Ci + ci] = acc[mi][ni][r];
      }
    }
  // db: the 4 threads that stage one row hold its pieces
  if (tile / a.tm_tiles == 0) {
#pragma unroll
    for (int j = 0; j < JD; ++j) {
      float s = dsum[j];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      const int co = coBase + row0 + 128 * j;
      if ((tid & 3) == 0 && co < Co) a.dbp[(size_t)split * a.pstride + co] = s;
    }
  }
}

int g_wg2_target4 = 512, g_wg2_kc128 = 32, g_wg2_target1 = 512, g_wg2_b3 = 1, g_wg3 = 3, g_wg3_target = 256, g_wg3_tm = 128;    // lab knobs (dsgcn_pwconv_tuning keys 7, 8); the values are the product's

struct Wg2Plan { int TM, TN, KC, cpn, chunks, tm_tiles, tn_tiles, splits, cps, b3, v3; size_t lds; };

bool wg2_plan(int n, int Ci, int Co, int L, Wg2Plan* p) {
  if ((long)Ci * L * 4 >= (1L << 31) - 64 || (long)Co * L * 4 >= (1L << 31) - 64) return false;
  p->v3 = 0;
  if (g_wg3 && g_wg2_b3 && (Co > 128 || Ci > 128) && ((Co > 64 && Ci > 64) || (g_wg3 & 2))) {     // wide: one workgroup per CU owns the whole tile
    // (g_wg3 bit 1, lab: also the lop-sided convs — CTR-GCN's conv4, R <= 32 -> 128..256 channels — measured below)
    p->v3 = 1; p->b3 = 1;
    p->TM = (Co > 128 && (g_wg3_tm == 256 || Ci <= 128)) ? 256 : 128;     // (g_wg3_tm = 128: two 128 x 256 tiles, half the partial rows)
    p->TN = Ci > 128 ? 256 : 128;
    p->KC = W3_KC;
    p->cpn = (L + p->KC - 1) / p->KC;
    p->chunks = n * p->cpn;
    p->tm_tiles = (Co + p->TM - 1) / p->TM;
    p->tn_tiles = (Ci + p->TN - 1) / p->TN;
    const int tiles = p->tm_tiles * p->tn_tiles;
    int target = g_wg3_target / tiles;
    if (target < 1) target = 1;
    if (tiles > 1) target = (target + 7) / 8 * 8;
    if (target > p->chunks) target = p->chunks;
    p->cps = (p->chunks + target - 1) / target;
    p->splits = (p->chunks + p->cps - 1) / p->cps;
    p->lds = (size_t)2 * 3 * (p->TM + p->TN) * W3_RB + (size_t)p->TM * 8 + (size_t)p->TN * 16;
    return true;
  }
  p->TM = Co > 64 ? 128 : 64;
  p->TN = Ci > 64 ? 128 : 64;
  p->KC = (p->TM == 128 && p->TN == 128) ? g_wg2_kc128 : ((p->TM == 64 && p->TN == 64) ? 128 : 64);
  p->cpn = (L + p->KC - 1) / p->KC;
  p->chunks = n * p->cpn;
  p->tm_tiles = (Co + p->TM - 1) / p->TM;
  p->tn_tiles = (Ci + p->TN - 1) / p->TN;
  const int tiles = p->tm_tiles * p->tn_tiles;
  int target = (tiles >= 4 ? g_wg2_target4 : g_wg2_target1) / tiles;
  if (target < 1) target = 1;
  if (tiles > 1) target = (target + 7) / 8 * 8;       // the XCD decode deals splits in groups of 8
  if (target > p->chunks) target = p->chunks;
  p->cps = (p->chunks + target - 1) / target;
  p->splits = (p->chunks + p->cps - 1) / p->cps;
  p->lds = ((size_t)(p->TM + p->TN) * (p->KC + 2) + 2 * p->TM + 4 * p->TN) * sizeof(float);
  // both sides wide: the fp32 form is bound by the matrix pipe -> three-term bf16 products
  p->b3 = (g_wg2_b3 && p->TM == 128 && p->TN == 128 && p->KC == 32) ? 1 : 0;
  if (p->b3) p->lds = (size_t)3 * (p->TM + p->TN) * (p->KC * 2 + 16) + (2 * p->TM + 4 * p->TN) * sizeof(float);
  return true;
}

template <int TM, int TN, int KC, bool B3 = false>
void wg2_launch(const Wg2Args& a, bool has2, bool hasc, dim3 grid, size_t lds, hipStream_t st) {
  if (has2) {
    if (hasc) hipLaunchKernelGGL((k_wg2<TM, TN, KC, true, true, B3>), grid, dim3(WG_NT), lds, st, a);
    else hipLaunchKernelGGL((k_wg2<TM, TN, KC, true, false, B3>), grid, dim3(WG_NT), lds, st, a);
  } else {
    if (hasc) hipLaunchKernelGGL((k_wg2<TM, TN, KC, false, true, B3>), grid, dim3(WG_NT), lds, st, a);
    else hipLaunchKernelGGL((k_wg2<TM, TN, KC, false, false, B3>), grid, dim3(WG_NT), lds, st, a);
  }
}

template <int TM, int TN>
void wg3_launch(const Wg2Args& a, bool has2, bool hasc, dim3 grid, size_t lds, hipStream_t st) {
  static bool raised = false;
  if (!raised) {
    const void* fs[4] = {reinterpret_cast<const void*>(&k_wg3<TM, TN, false, false>), reinterpret_cast<const void*>(&k_wg3<TM, TN, false, true>),
                         reinterpret_cast<const void*>(&k_wg3<TM, TN, true, false>), reinterpret_cast<const void*>(&k_wg3<TM, TN, true, true>)};
    for (const void* f : fs) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    raised = true;
  }
  if (has2) {
    if (hasc) hipLaunchKernelGGL((k_wg3<TM, TN, true, true>), grid, dim3(W3_NT), lds, st, a);
    else hipLaunchKernelGGL((k_wg3<TM, TN, true, false>), grid, dim3(W3_NT), lds, st, a);
  } else {
    if (hasc) hipLaunchKernelGGL((k_wg3<TM, TN, false, true>), grid, dim3(W3_NT), lds, st, a);
    else hipLaunchKernelGGL((k_wg3<TM, TN, false, false>), grid, dim3(W3_NT), lds, st, a);
  }
}

}  // namespace

__attribute__((visibility("hidden"))) int dsgcn_wg2_tuning(int key, int value) {
  if (key == 0) g_wg2_target4 = value;
  else if (key == 1) g_wg2_kc128 = value;
  else if (key == 2) g_wg2_target1 = value;
  else if (key == 3) g_wg2_b3 = value;
  else if (key == 4) g_wg3 = value;
  else if (key == 5) g_wg3_target = value;
  else if (key == 6) g_wg3_tm = value;
  else return DSGCN_EINVAL;
  return 0;
}

// Internal (not exported): K-split count of the fast weight gradient for this shape, 0 = not eligible.
__attribute__((visibility("hidden"))) int dsgcn_wg2_splits(int n, int Ci, int Co, int L) {
  Wg2Plan p;
  return wg2_plan(n, Ci, Co, L, &p) ? p.splits : 0;
}

// Returns 1 = launched, 0 = not eligible (caller falls back), other = error.
__attribute__((visibility("hidden"))) int dsgcn_wg2(const float* x1, const float* s1, const float* h1, const float* x2,
                                                     const float* s2, const float* h2, int relu, const float* z,
                                                     const float* gz, const float* A0, const float* B0, float* dwp,
                                                     float* dbp, int pstride, int n, int Ci, int Co, int L,
                                                     hipStream_t st, const BnCoefTable* jobs, int ngroup,
                                                     const float* const* gx1, const float* const* gs1,
                                                     const float* const* gh1, const float* const* ggz, float* const* gdwp,
                                                     float* const* gdbp) {
  Wg2Plan p;
  if (!wg2_plan(n, Ci, Co, L, &p)) return 0;
  Wg2Args a = {};
  if (jobs) a.jobs = *jobs;
  if (ngroup > 1) {
    if (ngroup > 3 || x2 || A0) return 0;
    a.ngroup = ngroup;
    for (int g = 0; g < ngroup; ++g) a.g[g] = Wg2Args::Grp{gx1[g], gs1[g], gh1[g], ggz[g], gdwp[g], gdbp[g]};
  }
  a.x1 = x1; a.s1 = s1; a.h1 = h1; a.x2 = x2; a.s2 = s2; a.h2 = h2; a.relu = relu;
  a.z = z; a.gz = gz; a.A0 = A0; a.B0 = B0; a.dwp = dwp; a.dbp = dbp; a.pstride = pstride;
  a.n = n; a.Ci = Ci; a.Co = Co; a.L = L; a.cpn = p.cpn; a.total_chunks = p.chunks; a.cps = p.cps;
  a.tm_tiles = p.tm_tiles; a.tn_tiles = p.tn_tiles;
  const int tiles = p.tm_tiles * p.tn_tiles;
  a.main_blocks = tiles > 1 ? (p.splits + 7) / 8 * 8 * tiles : p.splits;
  const dim3 grid((unsigned)(a.main_blocks + bnj_total_blocks(a.jobs)), ngroup > 1 ? (unsigned)ngroup : 1u);
  const bool has2 = x2 != nullptr, hasc = A0 != nullptr;
  if (p.v3) {
#ifdef DSGCN_LAB                                   // the full 256 x 256 tile (key 17 = 256) measured slower in the step: lab builds only
    if (p.TM == 256 && p.TN == 256) wg3_launch<256, 256>(a, has2, hasc, grid, p.lds, st);
    else
#endif
    if (p.TM == 256) wg3_launch<256, 128>(a, has2, hasc, grid, p.lds, st);
    else wg3_launch<128, 256>(a, has2, hasc, grid, p.lds, st);
    DSGCN_LAUNCH_CHECK();
    return 1;
  }
#ifdef DSGCN_LAB                                   // KC = 64 staging of the 128 x 128 tile (key 8 = 64) spills: lab builds only
  if (p.TM == 128 && p.TN == 128 && p.KC == 64) wg2_launch<128, 128, 64>(a, has2, hasc, grid, p.lds, st);
  else
#endif
  if (p.b3) wg2_launch<128, 128, 32, true>(a, has2, hasc, grid, p.lds, st);
  else if (p.TM == 128 && p.TN == 128) wg2_launch<128, 128, 32>(a, has2, hasc, grid, p.lds, st);
  else if (p.TM == 128) wg2_launch<128, 64, 64>(a, has2, hasc, grid, p.lds, st);
  else if (p.TN == 128) wg2_launch<64, 128, 64>(a, has2, hasc, grid, p.lds, st);
  else wg2_launch<64, 64, 128>(a, has2, hasc, grid, p.lds, st);
  DSGCN_LAUNCH_CHECK();
  return 1;
}
