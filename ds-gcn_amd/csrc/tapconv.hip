// K-D (core): temporal (KT x 1) convolutions, max-pool and strided copy over channel windows of one tensor, one launch
// per direction.  Serves
//   * the temporal branches of dgmstcn between branch_act and combine (reference: pyskl/models/gcns/utils/tcn.py:379-398,
//     410-415): four dilated (3,1) convs, the (3,1) max-pool and the strided '1x1' pass-through;
//   * the branches of CTR-GCN's MSTCN (pyskl/models/gcns/utils/msg3d_utils.py:84-117): two dilated (5,1) convs, max-pool,
//     strided pass-through;
//   * the dense (9,1) temporal conv of ST-GCN's unit_tcn (tcn.py:21-28, stgcn.py:45-46).
//   conv  window : o[n,co0+co,t',x] = b[co] + sum_tap sum_ci W[co,ci,tap] * h[n,ci0+ci, t'*s + (tap-KT/2)*d, x]   (zero pad)
//   max   window : o = max over the valid taps of a (3,1) window, stride s, pad 1
//   copy  window : o = h[..., ::s, :]
// h is (n, Cin, T, V1), o is (n, Cout, T', V1); every window writes its own channel range, so there is no torch.cat.
//
// Convs run on the f32 matrix core, wave-independent like K-C: lane = output position, B operand = h read straight from
// HBM/L2 with raw buffer loads at the tap-shifted row (out-of-range rows -> buffer OOB -> 0), A operand = a 64x64xKT
// weight tile staged in LDS as [co][tap*CP + ci]; wider convs loop over 64-channel source chunks (grid.z covers the
// destination chunks).  Backward: dgrad = the transposed gather (same kernel shape over input positions), wgrad =
// per-tap GEMM over positions with LDS-staged do / shifted-h tiles, K-split partials reduced by dsgcn_colsum.
#include <algorithm>

#include "common.h"

namespace {

constexpr int TC_NT = 256;
constexpr int TC_MAXBR = 8;
constexpr int TC_OOB = 0x7ffffff0;

struct TBranch {
  int type;            // 0 conv, 1 max, 2 copy
  int ci0, co0;        // first channel of the window in h / in o
  int cin, cout;       // window widths (equal for max / copy)
  int dil;
  const float* w;      // (cout, cin, KT, 1)
  const float* b;      // (cout)
  float* dwp;          // (splits, cout*cin*KT) partials
  float* dbp;          // (splits, cout)
};

struct TArgs {
  const float* h;      // (n, Cin, T, V1)
  float* o;            // (n, Cout, Tout, V1)
  const float* go;     // grad of o
  float* dh;           // grad of h
  int n, Cin, Cout, T, Tout, V1, stride, nbr, splits, pstride, dch, narrow;
  TBranch br[TC_MAXBR];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tc_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float tc_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ int tc_row32(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ int tc_cp(const TBranch& br) { return (min(64, max(br.cin, 1)) + 7) & ~7; }

// weight tile (co0l.., ci0l..) of W (cout, cin, KT) -> LDS [co_l][tap*CP + ci_l], zero padded to 64 rows
template <int KT>
__device__ __forceinline__ void tc_stage_w(const TBranch& br, float* Ws, int co0l, int nco, int ci0l, int nci, int CP) {
  const int S = KT * CP + 1;
  if (nco < 64 || nci < CP) {
    for (int i = threadIdx.x; i < 64 * S + 64; i += blockDim.x) Ws[i] = 0.f;
    __syncthreads();
  }
  const int run = nci * KT;
  const int total = nco * run;
  const int ntb = blockDim.x;
  for (int i0 = threadIdx.x; i0 < total; i0 += ntb * 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * ntb;
      if (i < total) {
        const int co = i / run, r = i - co * run;
        v[q] = br.w[((size_t)(co0l + co) * br.cin + ci0l) * KT + r];
      } else {
        v[q] = 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * ntb;
      if (i < total) {
        const int co = i / run, r = i - co * run, ci = r / KT, tap = r - ci * KT;
        Ws[co * S + tap * CP + ci] = v[q];
      }
    }
  }
  __syncthreads();
}

// Accumulate one source chunk: acc[m] += A(Ws) * B(loaded values).
// FWD: A[i=co][k=ci] = Ws[co*S + tap*CP + ci];  BWD (transposed): A[i=ci][k=co] = Ws[co*S + tap*CP + ci]
template <int KT, bool FWD, int NG>
__device__ __forceinline__ void tc_accum(f32x16 (&acc)[2], const float* Ws, __amdgpu_buffer_rsrc_t rs, const int (&voff)[KT],
                                         int cstride4, int sbase, int ns, int mtiles, int CP, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const int S = KT * CP + 1;
  const int nsp = (ns + 7) & ~7;
  if (NG > 0) {
    // every conv window <= 8*NG channels wide: all loads of the tile (KT*4*NG dwords per lane) are issued before the
    // first MFMA — the load -> MFMA chain per tap below exposes one memory round trip per tap, which dominates such
    // short tiles (tools/tc_bench.py: 127 -> 50 us on the 64-channel layers)
    float x[KT][NG * 4];
#pragma unroll
    for (int tap = 0; tap < KT; ++tap)
#pragma unroll
      for (int u = 0; u < NG * 4; ++u) x[tap][u] = tc_load(rs, voff[tap], (sbase + 2 * u) * cstride4);
#pragma unroll
    for (int tap = 0; tap < KT; ++tap)
#pragma unroll
      for (int u = 0; u < NG * 4; ++u) {
        const int kl = 2 * u + half;
        const float bv = kl < ns ? x[tap][u] : 0.f;
#pragma unroll
        for (int m = 0; m < (NG > 4 ? 2 : 1); ++m) {
          const float av = FWD ? Ws[(32 * m + l31) * S + tap * CP + kl] : Ws[kl * S + tap * CP + 32 * m + l31];
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
        }
      }
    return;
  }
#pragma unroll
  for (int tap = 0; tap < KT; ++tap) {
    float xa[4], xb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xa[u] = tc_load(rs, voff[tap], (sbase + 2 * u) * cstride4);
    for (int k0 = 0; k0 < nsp; k0 += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) xb[u] = tc_load(rs, voff[tap], (sbase + k0 + 8 + 2 * u) * cstride4);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kl = k0 + 2 * u + half;
        const float bv = kl < ns ? xa[u] : 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          if (m < mtiles) {
            const float av = FWD ? Ws[(32 * m + l31) * S + tap * CP + kl] : Ws[kl * S + tap * CP + 32 * m + l31];
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
          }
        }
      }
      if (k0 + 8 < nsp) {
#pragma unroll
        for (int u = 0; u < 4; ++u) xa[u] = tc_load(rs, voff[tap], (sbase + k0 + 16 + 2 * u) * cstride4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kl = k0 + 8 + 2 * u + half;
          const float bv = kl < ns ? xb[u] : 0.f;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            if (m < mtiles) {
              const float av = FWD ? Ws[(32 * m + l31) * S + tap * CP + kl] : Ws[kl * S + tap * CP + 32 * m + l31];
              acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
            }
          }
        }
      }
    }
  }
}

// One block = (4 position tiles of 32) x (one destination chunk of <= 64 channels); loops over the source chunks.
template <int KT, bool FWD, int NG>
__device__ __forceinline__ void tc_conv(const TArgs& a, const TBranch& br, float* Ws, int n, int tile, int chunk,
                                        int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const int V1 = a.V1;
  const int CP = tc_cp(br);
  const int Lout = a.Tout * V1, Lin = a.T * V1;
  const int L = FWD ? Lout : Lin;
  const int pos = tile * 32 + l31;
  const bool pok = pos < L;
  const int pc = pok ? pos : 0;
  const int row = pc / V1, col = pc - row * V1;
  // source tensor: forward reads h (T rows, Cin channels), backward reads go (Tout rows, Cout channels)
  const int Tsrc = FWD ? a.T : a.Tout;
  const int Csrc = FWD ? a.Cin : a.Cout;
  const int sch0 = FWD ? br.ci0 : br.co0;          // first source channel of the window
  const int nsrc = FWD ? br.cin : br.cout;
  const int ndst_all = FWD ? br.cout : br.cin;
  const int d0 = chunk * 64, nd = min(64, ndst_all - d0);
  const size_t src_bytes = (size_t)a.n * Csrc * Tsrc * V1 * 4;
  const __amdgpu_buffer_rsrc_t rs = tc_rsrc(FWD ? a.h : a.go, src_bytes);
  const int cstride4 = Tsrc * V1 * 4;
  int voff[KT];
#pragma unroll
  for (int tap = 0; tap < KT; ++tap) {
    int rs_row;
    bool ok = pok;
    if (FWD) {
      rs_row = row * a.stride + (tap - KT / 2) * br.dil;
      ok = ok && rs_row >= 0 && rs_row < a.T;
    } else {
      const int num = row - (tap - KT / 2) * br.dil;
      rs_row = num / a.stride;
      ok = ok && num >= 0 && (num - rs_row * a.stride) == 0 && rs_row < a.Tout;
    }
    voff[tap] = ok ? (int)((((size_t)n * Csrc + sch0 + half) * Tsrc + rs_row) * V1 + col) * 4 : TC_OOB;
  }
  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
  const int mtiles = (nd + 31) / 32;
  for (int s0 = 0; s0 < nsrc; s0 += 64) {
    const int ns = min(64, nsrc - s0);
    if (s0 > 0) __syncthreads();                   // every wave is done with the previous tile
    if (FWD) tc_stage_w<KT>(br, Ws, d0, nd, s0, ns, CP);
    else tc_stage_w<KT>(br, Ws, s0, ns, d0, nd, CP);
    tc_accum<KT, FWD, NG>(acc, Ws, rs, voff, cstride4, s0, ns, mtiles, CP, lane);
  }
  // D[i=channel][j=position]
  float* dst = FWD ? a.o : a.dh;
  const int Tdst = FWD ? a.Tout : a.T;
  const int Cdst = FWD ? a.Cout : a.Cin;
  const int dch0 = (FWD ? br.co0 : br.ci0) + d0;
  if (pok) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = 32 * m + tc_row32(r, half);
        if (ch < nd) {
          const float bias = (FWD && br.b) ? br.b[d0 + ch] : 0.f;
          dst[((size_t)(n * Cdst + dch0 + ch) * Tdst) * V1 + pos] = acc[m][r] + bias;
        }
      }
    }
  }
}

template <int KT, int NG>
__global__ __launch_bounds__(512) void k_tapconv_fwd(TArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int bi = blockIdx.z / a.dch, chunk = blockIdx.z - bi * a.dch;
  const TBranch& br = a.br[bi];
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int V1 = a.V1, Lout = a.Tout * V1;
  if (br.type == 0) {
    if (chunk * 64 >= br.cout) return;
    tc_conv<KT, true, NG>(a, br, lds, n, blockIdx.x * (blockDim.x >> 6) + wave, chunk, lane);
    return;
  }
  if (chunk > 0) return;
  // elementwise windows: thread = output position, loop over the window's channels
  const int pb = blockDim.x >> 1;                  // positions per block (32 per wave)
  const int pos = blockIdx.x * pb + (threadIdx.x % pb);
  if (pos >= Lout) return;
  const int tp = pos / V1, col = pos - tp * V1;
  for (int c = threadIdx.x / pb; c < br.cout; c += 2) {
    const float* hp = a.h + ((size_t)(n * a.Cin + br.ci0 + c) * a.T) * V1 + col;
    float v;
    if (br.type == 2) {
      v = hp[(size_t)(tp * a.stride) * V1];
    } else {
      v = -INFINITY;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int t = tp * a.stride + k - 1;
        if (t >= 0 && t < a.T) v = fmaxf(v, hp[(size_t)t * V1]);
      }
    }
    a.o[((size_t)(n * a.Cout + br.co0 + c) * a.Tout) * V1 + pos] = v;
  }
}

template <int KT, int NG>
__global__ __launch_bounds__(512) void k_tapconv_dgrad(TArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int bi = blockIdx.z / a.dch, chunk = blockIdx.z - bi * a.dch;
  const TBranch& br = a.br[bi];
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int V1 = a.V1, Lin = a.T * V1;
  if (br.type == 0) {
    if (chunk * 64 >= br.cin) return;
    tc_conv<KT, false, NG>(a, br, lds, n, blockIdx.x * (blockDim.x >> 6) + wave, chunk, lane);
    return;
  }
  if (chunk > 0) return;
  const int pb = blockDim.x >> 1;
  const int pos = blockIdx.x * pb + (threadIdx.x % pb);
  if (pos >= Lin) return;
  const int t = pos / V1, col = pos - t * V1;
  for (int c = threadIdx.x / pb; c < br.cin; c += 2) {
    const float* hp = a.h + ((size_t)(n * a.Cin + br.ci0 + c) * a.T) * V1 + col;
    const float* gp = a.go + ((size_t)(n * a.Cout + br.co0 + c) * a.Tout) * V1 + col;
    float g = 0.f;
    if (br.type == 2) {
      if (t % a.stride == 0 && t / a.stride < a.Tout) g = gp[(size_t)(t / a.stride) * V1];
    } else {
      // max-pool: the gradient of window t' goes to its FIRST maximal valid tap (ATen max_pool2d_with_indices order)
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int num = t - (kk - 1);
        if (num < 0 || num % a.stride != 0) continue;
        const int tp = num / a.stride;
        if (tp >= a.Tout) continue;
        float best = -INFINITY;
        int arg = -1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int tt = tp * a.stride + k - 1;
          if (tt >= 0 && tt < a.T) {
            const float v = hp[(size_t)tt * V1];
            if (v > best || arg < 0) { best = v; arg = tt; }
          }
        }
        if (arg == t) g += gp[(size_t)tp * V1];
      }
    }
    a.dh[((size_t)(n * a.Cin + br.ci0 + c) * a.T) * V1 + pos] = g;
  }
}

// wgrad: grid = (splits, nbr * dch * dch).  Block: a 64 (co) x 64 (ci) tile of one conv window, KT accumulators per wave
// tile.  Chunk of work = (sample, 2 output frames): Ds[64][KP] = do, Xs[tap][64][KP] = h at the tap-shifted frames.
// W8: 512 threads — the (co, ci) tile is still 2x2 wave tiles, the two wave groups split the TAPS (each keeps (KT+1)/2
// accumulators); used where the LDS image allows only one workgroup per CU.
template <int KT, bool W8>
__global__ __launch_bounds__(W8 ? 512 : TC_NT) void k_tapconv_wgrad(TArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int bi = blockIdx.y / (a.dch * a.dch);
  const int rem = blockIdx.y - bi * a.dch * a.dch;
  const int coch = rem / a.dch, cich = rem - coch * a.dch;
  const TBranch& br = a.br[bi];
  if (br.type != 0 || coch * 64 >= br.cout || cich * 64 >= br.cin) return;
  const int V1 = a.V1;
  const int nco = min(64, br.cout - coch * 64), nci = min(64, br.cin - cich * 64);
  const int chd = br.co0 + coch * 64, chx = br.ci0 + cich * 64;
  constexpr int TRW = 2;
  const int KP = (TRW * V1 + 1) & ~1, LS = KP | 1;
  float* Ds = lds;                                // [64][LS]
  float* Xs = lds + 64 * LS;                      // [KT][64][LS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  constexpr int KTG = W8 ? (KT + 1) / 2 : KT;        // taps per wave group
  const int tap0 = W8 ? (wave >> 2) * KTG : 0;
  const int mt = (wave >> 1) & 1, nt = wave & 1;
  const int nb = (a.Tout + TRW - 1) / TRW;
  const int total = a.n * nb;
  const int per = (total + a.splits - 1) / a.splits;
  const int ch0 = blockIdx.x * per, ch1 = min(total, ch0 + per);
  f32x16 acc[KTG];
#pragma unroll
  for (int k = 0; k < KTG; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  float dbacc = 0.f;
  constexpr int TPR = W8 ? 8 : 4;                 // staging threads per channel row
  const int row = tid / TPR, quarter = tid % TPR;
  constexpr int NPT = 56 / TPR;                   // positions per thread: TPR*NPT = 56 >= KP for V1 <= 26
  float dv[NPT], xv[KT][NPT];
  auto issue = [&](int ch) {                      // all global loads of a chunk, issued before the previous chunk's MFMAs
    const int n = ch / nb, r0 = (ch - n * nb) * TRW;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int p = quarter + TPR * j;
      const int rl = p / V1, col = p - rl * V1;
      const int tp = r0 + rl;
      const bool live = p < TRW * V1 && tp < a.Tout;
      dv[j] = (live && row < nco) ? a.go[((size_t)(n * a.Cout + chd + row) * a.Tout + tp) * V1 + col] : 0.f;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const int t = tp * a.stride + (k - KT / 2) * br.dil;
        xv[k][j] = (live && row < nci && t >= 0 && t < a.T)
                       ? a.h[((size_t)(n * a.Cin + chx + row) * a.T + t) * V1 + col] : 0.f;
      }
    }
  };
  if (ch0 < ch1) issue(ch0);
  for (int ch = ch0; ch < ch1; ++ch) {
    __syncthreads();
    float dsum = 0.f;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int p = quarter + TPR * j;
      if (p < KP) {
        Ds[row * LS + p] = dv[j];
#pragma unroll
        for (int k = 0; k < KT; ++k) Xs[(k * 64 + row) * LS + p] = xv[k][j];
        dsum += dv[j];
      }
    }
    dsum += __shfl_xor(dsum, 1, 64);
    dsum += __shfl_xor(dsum, 2, 64);
    if (W8) dsum += __shfl_xor(dsum, 4, 64);
    dbacc += dsum;
    __syncthreads();
    if (ch + 1 < ch1) issue(ch + 1);
    for (int kk = 0; kk < KP; kk += 2) {
      const float av = Ds[(32 * mt + l31) * LS + kk + half];
#pragma unroll
      for (int k = 0; k < KTG; ++k) {
        if (tap0 + k < KT) {
          const float bv = Xs[((tap0 + k) * 64 + 32 * nt + l31) * LS + kk + half];
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[k], 0, 0, 0);
        }
      }
    }
  }
  // D[i=co][j=ci] per tap -> dwp[split][(co*cin + ci)*KT + tap]
  const int ci = 32 * nt + l31;
  float* dwp = br.dwp + (size_t)blockIdx.x * a.pstride;
  if (ci < nci) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = 32 * mt + tc_row32(r, half);
      if (co < nco) {
#pragma unroll
        for (int k = 0; k < KTG; ++k)
          if (tap0 + k < KT) dwp[((size_t)(coch * 64 + co) * br.cin + cich * 64 + ci) * KT + tap0 + k] = acc[k][r];
      }
    }
  }
  if (cich == 0 && quarter == 0 && row < nco) br.dbp[(size_t)blockIdx.x * a.pstride + coch * 64 + row] = dbacc;
}

// wgrad for narrow windows (every conv window <= 32 channels wide: the 64-/128-channel dgmstcn layers, MSTCN up to
// 128 channels): the (co x ci) tile is ONE 32x32 MFMA tile, so the four waves of a block would have nothing to split
// but zeros.  Here every wave is its own K-split instead: it walks its own (sample, output frame) chunks with a
// wave-private LDS image (Ds[32][LS] = do, Xs[tap][32][LS] = shifted h), register prefetch of the next chunk, no
// block barrier, and writes its own partial row (row = blockIdx.x*4 + wave).
template <int KT>
__global__ __launch_bounds__(TC_NT) void k_tapconv_wgrad_narrow(TArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const TBranch& br = a.br[blockIdx.y];
  if (br.type != 0) return;
  const int V1 = a.V1;
  const int nco = br.cout, nci = br.cin;
  const int KP = (V1 + 1) & ~1, LS = KP | 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Ds = lds + (size_t)wave * (1 + KT) * 32 * LS;      // [32][LS]
  float* Xs = Ds + 32 * LS;                                 // [KT][32][LS]
  const int half = lane >> 5, l31 = lane & 31;
  // staging: lane = (row parity, position): one instruction moves two whole channel rows (two 104-B segments)
  const int total = a.n * a.Tout;
  const int nsplit = a.splits;                              // rows of the partial buffer = waves in the grid
  const int per = (total + nsplit - 1) / nsplit;
  const int me = blockIdx.x * 4 + wave;
  const int ch0 = me * per, ch1 = min(total, ch0 + per);
  f32x16 acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  constexpr int NRP = 16;                                   // row pairs
  float dbl[NRP];
#pragma unroll
  for (int j = 0; j < NRP; ++j) dbl[j] = 0.f;
  float dv[NRP], xv[KT][NRP];
  const bool plive = l31 < V1;
  auto issue = [&](int ch) {
    const int n = ch / a.Tout, tp = ch - n * a.Tout;
#pragma unroll
    for (int j = 0; j < NRP; ++j) {
      const int row = 2 * j + half;
      dv[j] = (plive && row < nco) ? a.go[((size_t)(n * a.Cout + br.co0 + row) * a.Tout + tp) * V1 + l31] : 0.f;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const int t = tp * a.stride + (k - KT / 2) * br.dil;
        xv[k][j] = (plive && row < nci && t >= 0 && t < a.T)
                       ? a.h[((size_t)(n * a.Cin + br.ci0 + row) * a.T + t) * V1 + l31] : 0.f;
      }
    }
  };
  if (ch0 < ch1) issue(ch0);
  for (int ch = ch0; ch < ch1; ++ch) {
    wave_lds_sync();                                        // the previous chunk's operand reads are done
    if (l31 < KP) {
#pragma unroll
      for (int j = 0; j < NRP; ++j) {
        const int row = 2 * j + half;
        Ds[row * LS + l31] = dv[j];
#pragma unroll
        for (int k = 0; k < KT; ++k) Xs[(k * 32 + row) * LS + l31] = xv[k][j];
        dbl[j] += dv[j];
      }
    }
    if (ch + 1 < ch1) issue(ch + 1);
    wave_lds_sync();
    for (int kk = 0; kk < KP; kk += 2) {
      const float av = Ds[l31 * LS + kk + half];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const float bv = Xs[(k * 32 + l31) * LS + kk + half];
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[k], 0, 0, 0);
      }
    }
  }
  // D[i=co][j=ci] per tap -> dwp[row me][(co*cin + ci)*KT + tap]
  float* dwp = br.dwp + (size_t)me * a.pstride;
  if (l31 < nci) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = tc_row32(r, half);
      if (co < nco) {
#pragma unroll
        for (int k = 0; k < KT; ++k) dwp[((size_t)co * br.cin + l31) * KT + k] = acc[k][r];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NRP; ++j) {
    float v = dbl[j];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);      // over the 32 positions of this half
    const int row = 2 * j + half;
    if (l31 == 0 && row < nco) br.dbp[(size_t)me * a.pstride + row] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Wide-load form of the forward / data gradient ("tap4") for the multi-scale units' narrow windows: stride 1, KT = 3,
// windows <= 64 channels, T*V1 % 4 == 0 and dil*V1 even (DS-STGCN: V1 = 26; K400: 18).  Same idea as K-C's pw4 form: a
// lane owns FOUR consecutive positions of one channel row, register q of the loaded vector is the B fragment of
// position sub-tile q, and the four results of a lane are again 16 consecutive bytes.  The tap shift is a whole number of
// rows (+-dil*V1 positions, even): the centre tap is one aligned 16-byte load, a side tap two 8-byte loads, each pair
// entirely inside or entirely outside the plane (zero padding = the buffer bounds check, no per-element masks).  K runs
// channel-pair major / tap minor through a pinned software pipeline (3*PDK operand slots in flight, A fragments read
// from LDS one step ahead).  Position tiles run over all samples back to back.  The max-pool / pass-through windows
// ride in the same launch as whole-plane passes through LDS (one wave per (n,c) plane) — the first version walked
// their channels in a per-thread loop of dependent loads and recomputed the pooling argmax from global memory
// (tools/tc_bench.py, 128 samples: forward 53..93 us, data gradient 71..134 us per layer before).
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x2t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 t4_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ f32x2t t4_load2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x2t, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

template <bool FWD, int MT, bool S2>
__device__ __forceinline__ void t4_conv(const TArgs& a, const TBranch& br, float* Ws, int grp, int lane, int wave) {
  constexpr int KT = 3, PDK = 2, NS = KT * PDK;
  const int half = lane >> 5, l31 = lane & 31;
  const int V1 = a.V1;
  const int Ls = (FWD ? a.T : a.Tout) * V1, L = (FWD ? a.Tout : a.T) * V1;   // source / destination plane lengths
  const int L4 = Ls * 4;
  const int Csrc = FWD ? a.Cin : a.Cout, Cdst = FWD ? a.Cout : a.Cin;
  const int sch0 = FWD ? br.ci0 : br.co0, nsrc = FWD ? br.cin : br.cout;
  const int dch0 = FWD ? br.co0 : br.ci0, ndst = FWD ? br.cout : br.cin;
  const int CP = tc_cp(br), S = KT * CP + 1;
  tc_stage_w<KT>(br, Ws, 0, br.cout, 0, br.cin, CP);          // [co][tap*CP + ci], zero padded; ends with a barrier

  const int wt = grp * 4 + wave;
  const long total = (long)a.n * L;
  const bool wlive = (long)wt * 128 < total;
  const int g0 = wlive ? wt * 128 : 0;
  const int n0 = g0 / L;
  int p = g0 - n0 * L + 4 * l31, ds = 0;
  while (p >= L) { p -= L; ++ds; }
  const bool pok = wlive && n0 + ds < a.n;
  const __amdgpu_buffer_rsrc_t rs = tc_rsrc(FWD ? a.h : a.go, (size_t)a.n * Csrc * L4);
  const int rowbase = ((n0 + ds) * Csrc + sch0 + half) * Ls;  // (the launch checks the tensor stays below 2^31 bytes)
  // operand offsets of the lane's two position pairs, per tap.  Stride 1: a tap shifts by whole rows, the centre tap is
  // one 16-byte load.  Stride 2 (forward: output row t' reads input rows 2t' + (tap-1)*dil; data gradient: input row t
  // collects from the output rows (t - (tap-1)*dil)/2 that exist): every pair has its own row, all loads are 8 bytes.
  int vP[KT][2];
  if constexpr (!S2) {
    const int sh = (FWD ? br.dil : -br.dil) * V1;             // shift of tap 2 (tap 0: -sh)
#pragma unroll
    for (int tap = 0; tap < KT; ++tap)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = p + (tap - 1) * sh + 2 * j;
        vP[tap][j] = (pok && q >= 0 && q <= Ls - 2) ? (rowbase + q) * 4 : TC_OOB;
      }
  } else {
    const float invV1 = 1.f / (float)V1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int t, x;
      divmod_small(p + 2 * j, V1, invV1, t, x);
#pragma unroll
      for (int tap = 0; tap < KT; ++tap) {
        int row;
        bool ok = pok;
        if (FWD) {
          row = 2 * t + (tap - 1) * br.dil;
          ok = ok && row >= 0 && row < a.T;
        } else {
          const int num = t - (tap - 1) * br.dil;
          row = num >> 1;
          ok = ok && num >= 0 && !(num & 1) && row < a.Tout;
        }
        vP[tap][j] = ok ? (rowbase + row * V1 + x) * 4 : TC_OOB;
      }
    }
  }
  auto load = [&](int tap, int ks) -> f32x4 {
    const int soff = 2 * ks * L4;
    if (!S2 && tap == 1) return t4_load4(rs, vP[1][0], soff);
    const f32x2t lo = t4_load2(rs, vP[tap][0], soff), hi = t4_load2(rs, vP[tap][1], soff);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
  };

  const int KS2 = ((nsrc + 3) >> 2) << 1;                    // k-steps of two channels, even (weights are zero past nsrc)
  f32x4 buf[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    buf[u] = load(u % KT, u / KT);
    __builtin_amdgcn_sched_barrier(0);             // issue order = slot order, as in the loop (exact vmcnt waits)
  }
  f32x16 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;
  // A fragment of step (ks, tap): FWD Ws[(32m + l31)*S + tap*CP + 2ks + half];  BWD Ws[(2ks + half)*S + tap*CP + 32m + l31]
  auto afrag = [&](int tap, int ks, int m) -> float {
    const int kl = 2 * ks + half;
    return FWD ? Ws[(32 * m + l31) * S + tap * CP + kl] : Ws[kl * S + tap * CP + 32 * m + l31];
  };
  float avb[2][MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) avb[0][m] = afrag(0, 0, m);
  for (int base = 0; base < KS2; base += PDK) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int tap = u % KT, ks = base + u / KT;
      const int cur = u & 1, nxt = cur ^ 1;                   // NS is even: the parity survives the back edge
      const int tn = (u + 1) % KT, kn = base + (u + 1) / KT;  // next step (past the end: reads pad rows, never used)
#pragma unroll
      for (int m = 0; m < MT; ++m) avb[nxt][m] = afrag(tn, kn < 32 ? kn : 31, m);
      const f32x4 b = buf[u];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(avb[cur][m], b[q], acc[m][q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      buf[u] = load(tap, min(ks + PDK, KS2 - 1));           // branch-free (the tail re-reads the last step: L2 hits), so the
                                                            // compiler's vmcnt bookkeeping stays exact
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float* dst = FWD ? a.o : a.dh;
  const __amdgpu_buffer_rsrc_t ro = tc_rsrc(dst, (size_t)a.n * Cdst * L * 4);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ch = 32 * m + tc_row32(r, half);
      const float bias = (FWD && br.b && ch < ndst) ? br.b[ch] : 0.f;
      const f32x4 v = {acc[m][0][r] + bias, acc[m][1][r] + bias, acc[m][2][r] + bias, acc[m][3][r] + bias};
      const int voff = (pok && ch < ndst) ? (((n0 + ds) * Cdst + dch0 + ch) * L + p) * 4 : TC_OOB;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, voff, 0, 0);
    }
  }
}

// one wave per (n, c) plane of a max-pool / pass-through window (stride 1 or 2)
template <bool FWD, int ST>
__device__ __forceinline__ void t4_elem(const TArgs& a, const TBranch& br, float* lw, int n, int c, int lane) {
  constexpr int st = ST;
  const int V1 = a.V1;
  const int L = a.T * V1, Lo = a.Tout * V1;
  const float invV1 = 1.f / (float)V1;
  const float* hp = a.h + ((size_t)n * a.Cin + br.ci0 + c) * L;
  if (FWD) {
    f32x4* op = reinterpret_cast<f32x4*>(a.o + ((size_t)n * a.Cout + br.co0 + c) * Lo);
    if (br.type == 2 && st == 1) {
      const f32x4* h4 = reinterpret_cast<const f32x4*>(hp);
      for (int i = lane; i < (L >> 2); i += 64) op[i] = h4[i];
      return;
    }
    plane_to_lds(hp, lw, L >> 2, lane);
    wave_lds_sync();
    for (int i = lane; i < (Lo >> 2); i += 64) {
      int tp, x;
      divmod_small(4 * i, V1, invV1, tp, x);
      float r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = tp * st * V1 + x;               // centre of the window / the strided source element
        float v = lw[e];
        if (br.type == 1) {
          if (tp * st - 1 >= 0) v = fmaxf(v, lw[e - V1]);
          if (tp * st + 1 < a.T) v = fmaxf(v, lw[e + V1]);
        }
        r[k] = v;
        if (++x == V1) { x = 0; ++tp; }
      }
      op[i] = f32x4{r[0], r[1], r[2], r[3]};
    }
    return;
  }
  const float* gp = a.go + ((size_t)n * a.Cout + br.co0 + c) * Lo;
  f32x4* dp = reinterpret_cast<f32x4*>(a.dh + ((size_t)n * a.Cin + br.ci0 + c) * L);
  if (br.type == 2 && st == 1) {
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gp);
    for (int i = lane; i < (L >> 2); i += 64) dp[i] = g4[i];
    return;
  }
  float* lg = lw + L;
  if (br.type == 1) plane_to_lds(hp, lw, L >> 2, lane);
  plane_to_lds(gp, lg, Lo >> 2, lane);
  wave_lds_sync();
  for (int i = lane; i < (L >> 2); i += 64) {
    int t, x;
    divmod_small(4 * i, V1, invV1, t, x);
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float g = 0.f;
      if (br.type == 2) {
        if (t % st == 0 && t / st < a.Tout) g = lg[(t / st) * V1 + x];
      } else {
        // max-pool: the gradient of window t' goes to its FIRST maximal valid row (ATen max_pool2d_with_indices order:
        // ties go to the earlier row).  Row t can be the last row of the window centred at t-1, the centre of its own
        // window, or the first row of the window centred at t+1 (centres are the multiples of the stride); it owns a
        // window iff it beats every earlier row strictly and every later row weakly.  Rows outside the plane: -inf.
        const float v = lw[t * V1 + x];
        auto row = [&](int r) -> float { return (r >= 0 && r < a.T) ? lw[r * V1 + x] : -INFINITY; };
        const float m2 = row(t - 2), m1 = row(t - 1), p1 = row(t + 1), p2 = row(t + 2);
        if (t >= 1 && (t - 1) % st == 0 && (t - 1) / st < a.Tout && v > m2 && v > m1) g += lg[((t - 1) / st) * V1 + x];
        if (t % st == 0 && t / st < a.Tout && v > m1 && v >= p1) g += lg[(t / st) * V1 + x];
        if ((t + 1) % st == 0 && (t + 1) / st < a.Tout && v >= p1 && v >= p2) g += lg[((t + 1) / st) * V1 + x];
      }
      r[k] = g;
      if (++x == V1) { x = 0; ++t; }
    }
    dp[i] = f32x4{r[0], r[1], r[2], r[3]};
  }
}

// grid.x = [conv blocks: (position group, conv window)] ++ [elementwise blocks: 4 planes each]
template <bool FWD, int MT, bool S2>
__global__ __launch_bounds__(TC_NT, MT == 1 ? 3 : 2) void k_tap4(TArgs a, int nconv, int ngrp, int eplanes) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // conv blocks first, elementwise blocks after them (dealing them evenly among the conv blocks measured slower)
  const int cb = nconv * ngrp, b = blockIdx.x;
  if (b < cb) {
    const int w = b % nconv, grp = b / nconv;
    int bi = 0;
    for (int i = 0, k = 0; i < a.nbr; ++i)
      if (a.br[i].type == 0) { if (k == w) bi = i; ++k; }
    t4_conv<FWD, MT, S2>(a, a.br[bi], lds, grp, lane, wave);
    return;
  }
  int pl = (b - cb) * 4 + wave;                    // plane index over (n, elementwise channels)
  if (pl >= a.n * eplanes) return;
  const int n = pl / eplanes;
  int c = pl - n * eplanes;
  for (int i = 0; i < a.nbr; ++i) {
    if (a.br[i].type == 0) continue;
    if (c < a.br[i].cin) {
      t4_elem<FWD, S2 ? 2 : 1>(a, a.br[i], lds + (size_t)wave * 2 * a.T * a.V1, n, c, lane);
      return;
    }
    c -= a.br[i].cin;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Wide-load weight gradient ("tapw") for the same shapes as tap4 (stride 1, KT = 3, V1 even, windows <= 64 channels)
// with T % 4 == 0:   dW[co,ci,tap] = sum_{n,p} do[n,co,p] * h[n,ci,p + (tap-1)*dil*V1],   db[co] = sum do.
// Work unit = (sample, 4 frames).  The workgroup stages do (4 frames) and h (the 4 frames plus a 4-frame halo on either
// side: every dilation <= 4 reads its shifted operand from the SAME tile) with 16-byte loads — the first version loaded
// each tap's shifted copy separately with 4-byte loads in 16-byte row pieces and was bound by the address path (TA), not
// by memory or the matrix core (92..121 us per layer at 128 samples) — into LDS rows of stride = 2 mod 4 floats
// (conflict-free ds_read_b64, one read feeds two MFMA k-steps as in K-C's weight gradient).  CH = 32: the (co x ci) tile
// is one MFMA tile, the four waves split the POSITIONS of a unit and their accumulators are summed through LDS at the
// end; CH = 64: 2x2 wave tiles.  grid = (K-splits, conv windows); every split writes its partial row (dsgcn_colsum).
// ---------------------------------------------------------------------------------------------------------------
constexpr int TW_R = 4, TW_H = 4;                 // frames per unit, halo frames per side (max dilation)

__host__ __device__ inline int tw_ls(int w) { return ((w + 1) & ~3) + 2; }      // >= w, = 2 mod 4

template <int CH>
__global__ __launch_bounds__(TC_NT, CH == 32 ? 2 : 1) void k_tapw(TArgs a, int nconv) {
  constexpr int KT = 3;
  constexpr int JD = CH == 32 ? 4 : 7, JX = CH == 32 ? 10 : 20;     // float4 staging slots per thread (V1 <= 26)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  int bi = 0;
  for (int i = 0, k = 0; i < a.nbr; ++i)
    if (a.br[i].type == 0) { if (k == (int)blockIdx.y) bi = i; ++k; }
  const TBranch& br = a.br[bi];
  const int V1 = a.V1, T = a.T, L = T * V1;
  const int nco = br.cout, nci = br.cin;
  const int GW = TW_R * V1, XW = (TW_R + 2 * TW_H) * V1;            // tile widths (floats); multiples of 4
  const int GW4 = GW >> 2, XW4 = XW >> 2, SEG4 = (TW_H * V1) >> 2;  // float4 per row; float4 per 4-frame segment
  const int LSd = tw_ls(GW), LSx = tw_ls(XW);
  float* Ds = lds;                                                   // [CH][LSd]
  float* Xs = lds + CH * LSd;                                        // [CH][LSx]
  for (int i = tid; i < CH * (LSd + LSx); i += TC_NT) lds[i] = 0.f;  // rows >= nco / nci and the pad columns stay zero

  const int units = a.n * (T / TW_R);
  const int per = (units + a.splits - 1) / a.splits;
  const int u0 = blockIdx.x * per, u1 = min(units, u0 + per);

  const __amdgpu_buffer_rsrc_t rg = tc_rsrc(a.go, (size_t)a.n * a.Cout * L * 4);
  const __amdgpu_buffer_rsrc_t rh = tc_rsrc(a.h, (size_t)a.n * a.Cin * L * 4);
  // staging slots: f = tid + 256*j -> (row, float4 column); fixed per thread
  int vD[JD], lD[JD], vX[JX], lX[JX], segX[JX];
#pragma unroll
  for (int j = 0; j < JD; ++j) {
    const int f = tid + TC_NT * j, row = f / GW4, c4 = f - row * GW4;
    const bool ok = row < nco;
    vD[j] = ok ? (row * L + 4 * c4) * 4 : TC_OOB;
    lD[j] = ok ? row * LSd + 4 * c4 : -1;
  }
#pragma unroll
  for (int j = 0; j < JX; ++j) {
    const int f = tid + TC_NT * j, row = f / XW4, c4 = f - row * XW4;
    const bool ok = row < nci;
    vX[j] = ok ? (row * L + 4 * c4) * 4 : TC_OOB;
    lX[j] = ok ? row * LSx + 4 * c4 : -1;
    segX[j] = c4 / SEG4;                                             // 0 = halo before, 1 = the unit, 2 = halo after
  }
  f32x4 gr[JD], xr[JX];
  float dsum[JD];
#pragma unroll
  for (int j = 0; j < JD; ++j) dsum[j] = 0.f;
  auto issue = [&](int u) {
    const int n = u / (T / TW_R), t0 = (u - n * (T / TW_R)) * TW_R;
    const int sg = ((n * a.Cout + br.co0) * L + t0 * V1) * 4;
    const int sx = ((n * a.Cin + br.ci0) * L + (t0 - TW_H) * V1) * 4;    // may be negative: only used with valid slots
    const bool before = t0 >= TW_H, after = t0 + TW_R < T;
#pragma unroll
    for (int j = 0; j < JD; ++j) gr[j] = t4_load4(rg, vD[j], sg);
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const bool ok = segX[j] == 1 || (segX[j] == 0 ? before : after);
      xr[j] = t4_load4(rh, ok ? vX[j] + sx : TC_OOB, 0);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < JD; ++j) {
      if (lD[j] >= 0) {
        f32x2t* d = reinterpret_cast<f32x2t*>(Ds + lD[j]);
        d[0] = f32x2t{gr[j].x, gr[j].y};
        d[1] = f32x2t{gr[j].z, gr[j].w};
        dsum[j] += (gr[j].x + gr[j].y) + (gr[j].z + gr[j].w);
      }
    }
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      if (lX[j] >= 0) {
        f32x2t* d = reinterpret_cast<f32x2t*>(Xs + lX[j]);
        d[0] = f32x2t{xr[j].x, xr[j].y};
        d[1] = f32x2t{xr[j].z, xr[j].w};
      }
    }
  };

  constexpr int NA = CH == 32 ? 1 : 1;
  f32x16 acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  (void)NA;
  // CH = 32: wave w takes the position groups [g0, g1) of the unit's GW4 groups of 4; CH = 64: wave = (mt, nt), all groups
  const int mt = CH == 64 ? (wave >> 1) : 0, nt = CH == 64 ? (wave & 1) : 0;
  const int g0 = CH == 32 ? (GW4 * wave) / 4 : 0, g1 = CH == 32 ? (GW4 * (wave + 1)) / 4 : GW4;
  const int sh = br.dil * V1;
  const float* Ap = Ds + (32 * mt + l31) * LSd + 2 * half;
  const float* Bp = Xs + (32 * nt + l31) * LSx + TW_H * V1 + 2 * half;

  __syncthreads();                                  // zero fill done
  if (u0 < u1) issue(u0);
  for (int u = u0; u < u1; ++u) {
    commit();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (u + 1 < u1) issue(u + 1);
    for (int g = g0; g < g1; ++g) {
      const f32x2t av = *reinterpret_cast<const f32x2t*>(Ap + 4 * g);
      const f32x2t b0 = *reinterpret_cast<const f32x2t*>(Bp + 4 * g - sh);
      const f32x2t b1 = *reinterpret_cast<const f32x2t*>(Bp + 4 * g);
      const f32x2t b2 = *reinterpret_cast<const f32x2t*>(Bp + 4 * g + sh);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b2.x, acc[2], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b2.y, acc[2], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // raw barrier: the next unit's loads stay in flight
  }

  float* dw = br.dwp + (size_t)blockIdx.x * a.pstride;
  float* db = br.dbp + (size_t)blockIdx.x * a.pstride;
  if (CH == 32) {
    // sum the four position shares: Rs[wave][tap][co][ci] in the tile's LDS (48 KB <= the staging image)
    float* Rs = lds;
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) Rs[((wave * KT + k) * 32 + tc_row32(r, half)) * 33 + l31] = acc[k][r];
    __syncthreads();
    for (int o = tid; o < KT * nco * nci; o += TC_NT) {
      const int co = o / (nci * KT), r2 = o - co * nci * KT, ci = r2 / KT, k = r2 - ci * KT;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += Rs[((w * KT + k) * 32 + co) * 33 + ci];
      dw[o] = v;                                     // (co*cin + ci)*KT + tap
    }
    __syncthreads();
  } else {
    const int ci = 32 * nt + l31;
    if (ci < nci) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * mt + tc_row32(r, half);
        if (co < nco) {
#pragma unroll
          for (int k = 0; k < KT; ++k) dw[((size_t)co * nci + ci) * KT + k] = acc[k][r];
        }
      }
    }
    __syncthreads();
  }
  // db: per-thread row pieces -> LDS -> one thread per row
  float* Bs = lds;                                   // [CH][GW4 + 1]... laid out [row][c4]
#pragma unroll
  for (int j = 0; j < JD; ++j) {
    const int f = tid + TC_NT * j, row = f / GW4, c4 = f - row * GW4;
    if (row < nco) Bs[row * (GW4 + 1) + c4] = dsum[j];
  }
  __syncthreads();
  if (tid < nco) {
    float v = 0.f;
    for (int c = 0; c < GW4; ++c) v += Bs[tid * (GW4 + 1) + c];
    db[tid] = v;
  }
}

// Stride-2 form of tapw:  dW[co,ci,tap] = sum do[n,co,t',x] * h[n,ci, 2t' + (tap-1)*dil, x].  The input rows a unit of 4
// output frames needs are every second row, so the haloed tile does not apply: the three taps get their own decimated
// tiles Xs[tap][ci][4 frames] (8-byte loads: a frame row is V1/2 pairs), the MFMA loop reads do and h at the SAME tile
// position.  Same grid, splits and epilogue as k_tapw.
template <int CH>
__global__ __launch_bounds__(TC_NT, CH == 32 ? 2 : 1) void k_tapw2(TArgs a, int nconv) {
  constexpr int KT = 3;
  constexpr int JD = CH == 32 ? 4 : 7, JX = CH == 32 ? 20 : 40;     // float4 / float2 staging slots per thread (V1 <= 26)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  int bi = 0;
  for (int i = 0, k = 0; i < a.nbr; ++i)
    if (a.br[i].type == 0) { if (k == (int)blockIdx.y) bi = i; ++k; }
  const TBranch& br = a.br[bi];
  const int V1 = a.V1, T = a.T, To = a.Tout, L = T * V1, Lo = To * V1;
  const int nco = br.cout, nci = br.cin;
  const int GW = TW_R * V1, GW4 = GW >> 2, PV = V1 >> 1;
  const int LSd = tw_ls(GW);
  float* Ds = lds;                                                   // [CH][LSd]
  float* Xs = lds + CH * LSd;                                        // [KT][CH][LSd]
  for (int i = tid; i < (1 + KT) * CH * LSd; i += TC_NT) lds[i] = 0.f;

  const int units = a.n * (To / TW_R);
  const int per = (units + a.splits - 1) / a.splits;
  const int u0 = blockIdx.x * per, u1 = min(units, u0 + per);
  const __amdgpu_buffer_rsrc_t rg = tc_rsrc(a.go, (size_t)a.n * a.Cout * Lo * 4);
  const __amdgpu_buffer_rsrc_t rh = tc_rsrc(a.h, (size_t)a.n * a.Cin * L * 4);
  int vD[JD], lD[JD], vX[JX], lX[JX], rX[JX];
#pragma unroll
  for (int j = 0; j < JD; ++j) {
    const int f = tid + TC_NT * j, row = f / GW4, c4 = f - row * GW4;
    const bool ok = row < nco;
    vD[j] = ok ? (row * Lo + 4 * c4) * 4 : TC_OOB;
    lD[j] = ok ? row * LSd + 4 * c4 : -1;
  }
  const int xslots = KT * TW_R * nci * PV;                           // (tap, frame r, channel, pair)
#pragma unroll
  for (int j = 0; j < JX; ++j) {
    const int f = tid + TC_NT * j;
    const int pp = f % PV, q = f / PV, ch = q % nci, tr = q / nci;   // tr = tap*4 + r
    const bool ok = f < xslots;
    const int tap = tr >> 2, r = tr & 3;
    vX[j] = ok ? (ch * L + 2 * pp) * 4 : TC_OOB;                     // + input row offset per unit
    lX[j] = ok ? (tap * CH + ch) * LSd + r * V1 + 2 * pp : -1;
    rX[j] = 2 * r + (tap - 1) * br.dil;                              // input row relative to 2*t0
  }
  f32x4 gr[JD];
  f32x2t xr[JX];
  float dsum[JD];
#pragma unroll
  for (int j = 0; j < JD; ++j) dsum[j] = 0.f;
  auto issue = [&](int u) {
    const int n = u / (To / TW_R), t0 = (u - n * (To / TW_R)) * TW_R;
    const int sg = ((n * a.Cout + br.co0) * Lo + t0 * V1) * 4;
    const int sx = (n * a.Cin + br.ci0) * L * 4;
#pragma unroll
    for (int j = 0; j < JD; ++j) gr[j] = t4_load4(rg, vD[j], sg);
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const int row = 2 * t0 + rX[j];
      const bool ok = row >= 0 && row < T;
      xr[j] = t4_load2(rh, ok ? vX[j] + row * V1 * 4 : TC_OOB, sx);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < JD; ++j) {
      if (lD[j] >= 0) {
        f32x2t* d = reinterpret_cast<f32x2t*>(Ds + lD[j]);
        d[0] = f32x2t{gr[j].x, gr[j].y};
        d[1] = f32x2t{gr[j].z, gr[j].w};
        dsum[j] += (gr[j].x + gr[j].y) + (gr[j].z + gr[j].w);
      }
    }
#pragma unroll
    for (int j = 0; j < JX; ++j)
      if (lX[j] >= 0) *reinterpret_cast<f32x2t*>(Xs + lX[j]) = xr[j];
  };
  f32x16 acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  const int mt = CH == 64 ? (wave >> 1) : 0, nt = CH == 64 ? (wave & 1) : 0;
  const int g0 = CH == 32 ? (GW4 * wave) / 4 : 0, g1 = CH == 32 ? (GW4 * (wave + 1)) / 4 : GW4;
  const float* Ap = Ds + (32 * mt + l31) * LSd + 2 * half;
  const float* Bp = Xs + (32 * nt + l31) * LSd + 2 * half;

  __syncthreads();
  if (u0 < u1) issue(u0);
  for (int u = u0; u < u1; ++u) {
    commit();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (u + 1 < u1) issue(u + 1);
    for (int g = g0; g < g1; ++g) {
      const f32x2t av = *reinterpret_cast<const f32x2t*>(Ap + 4 * g);
      const f32x2t b0 = *reinterpret_cast<const f32x2t*>(Bp + 4 * g);
      const f32x2t b1 = *reinterpret_cast<const f32x2t*>(Bp + CH * LSd + 4 * g);
      const f32x2t b2 = *reinterpret_cast<const f32x2t*>(Bp + 2 * CH * LSd + 4 * g);
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

  float* dw = br.dwp + (size_t)blockIdx.x * a.pstride;
  float* db = br.dbp + (size_t)blockIdx.x * a.pstride;
  if (CH == 32) {
    float* Rs = lds;
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) Rs[((wave * KT + k) * 32 + tc_row32(r, half)) * 33 + l31] = acc[k][r];
    __syncthreads();
    for (int o = tid; o < KT * nco * nci; o += TC_NT) {
      const int co = o / (nci * KT), r2 = o - co * nci * KT, ci = r2 / KT, k = r2 - ci * KT;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += Rs[((w * KT + k) * 32 + co) * 33 + ci];
      dw[o] = v;
    }
    __syncthreads();
  } else {
    const int ci = 32 * nt + l31;
    if (ci < nci) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * mt + tc_row32(r, half);
        if (co < nco) {
#pragma unroll
          for (int k = 0; k < KT; ++k) dw[((size_t)co * nci + ci) * KT + k] = acc[k][r];
        }
      }
    }
    __syncthreads();
  }
  float* Bs = lds;
#pragma unroll
  for (int j = 0; j < JD; ++j) {
    const int f = tid + TC_NT * j, row = f / GW4, c4 = f - row * GW4;
    if (row < nco) Bs[row * (GW4 + 1) + c4] = dsum[j];
  }
  __syncthreads();
  if (tid < nco) {
    float v = 0.f;
    for (int c = 0; c < GW4; ++c) v += Bs[tid * (GW4 + 1) + c];
    db[tid] = v;
  }
}

size_t tc_lds_conv(int CP, int KT) { return (size_t)(64 * (KT * CP + 1) + 64) * sizeof(float); }

constexpr size_t TC_LDS_MAX = 156 * 1024;

template <typename F>
int tc_raise_lds(F* kernel, size_t lds, size_t* have) {
  if (lds > TC_LDS_MAX) return DSGCN_EUNSUPPORTED;
  if (lds > *have) {       // not a stream op: done once per kernel, outside any graph capture (first eager call)
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TC_LDS_MAX);
    if (e != hipSuccess) return (int)e;
    *have = TC_LDS_MAX;
  }
  return 0;
}

// fills a.br / a.dch, returns the widest padded source chunk (CP) over the conv windows, or <0 on error
int tc_fill(TArgs& a, int nbr, const int* type, const int* ci0, const int* co0, const int* cin, const int* cout,
            const int* dil) {
  int cpmax = 8, dch = 1, wmax = 0;
  for (int i = 0; i < nbr; ++i) {
    TBranch& b = a.br[i];
    b.type = type[i]; b.ci0 = ci0[i]; b.co0 = co0[i]; b.cin = cin[i]; b.cout = cout[i]; b.dil = dil[i];
    if (b.cin <= 0 || b.cout <= 0 || b.ci0 < 0 || b.co0 < 0 || b.ci0 + b.cin > a.Cin || b.co0 + b.cout > a.Cout)
      return DSGCN_EINVAL;
    if (b.type != 0 && b.cin != b.cout) return DSGCN_EINVAL;
    if (b.type == 0) {
      const int cp = (std::min(64, b.cin) + 7) & ~7;
      if (cp > cpmax) cpmax = cp;
      const int d = (std::max(b.cin, b.cout) + 63) / 64;
      if (d > dch) dch = d;
      wmax = std::max(wmax, std::max(b.cin, b.cout));
    }
  }
  a.dch = dch;
  a.narrow = wmax;                                 // widest conv window (channels)
  return cpmax;
}


// tap4 eligibility and launch (forward / data gradient); returns 1 = launched, 0 = not eligible, else an error code
template <bool FWD>
int t4_try(const TArgs& a, int KT, hipStream_t st) {
  if (KT != 3 || (a.stride != 1 && a.stride != 2) || a.narrow < 0 || a.narrow > 64) return 0;
  const long L = (long)a.T * a.V1, Lo = (long)a.Tout * a.V1;
  int nconv = 0, eplanes = 0, cpmax = 8;
  for (int i = 0; i < a.nbr; ++i) {
    const TBranch& b = a.br[i];
    if (b.type == 0) {
      ++nconv;
      cpmax = std::max(cpmax, (std::min(64, b.cin) + 7) & ~7);
    } else {
      eplanes += b.cin;
    }
  }
  // (a launch of pooling / pass-through windows only — the strided frame copy in front of a block's residual conv — has no
  // tap shifts to align: any joint count with 16-byte planes; it used to fall to the first-generation kernel, 33-47 us
  // for a copy of 26 MB)
  if (L % 4 || Lo % 4 || (nconv && (a.V1 & 1)) || (a.stride == 2 && (a.T & 1))) return 0;
  if (nconv && a.narrow <= 0) return 0;
  const int cmax = a.Cin > a.Cout ? a.Cin : a.Cout;
  if ((long)a.n * cmax * L * 4 >= (1L << 31) - 4096 || (long)a.n * L >= (1L << 31) - 256) return 0;
  const size_t lds = std::max(tc_lds_conv(cpmax, 3), (size_t)4 * 2 * L * sizeof(float));
  if (lds > 64 * 1024) return 0;
  const long Ld = FWD ? Lo : L;                    // destination plane
  const int WT = (int)(((long)a.n * Ld + 127) / 128), ngrp = (WT + 3) / 4;
  const long blocks = (long)nconv * ngrp + ((long)a.n * eplanes + 3) / 4;
  if (blocks <= 0 || blocks >= (1L << 31)) return 0;
  const dim3 grid((unsigned)blocks), blk(TC_NT);
  const bool m1 = a.narrow <= 32;
  if (a.stride == 1) {
    if (m1) hipLaunchKernelGGL((k_tap4<FWD, 1, false>), grid, blk, lds, st, a, nconv, ngrp, eplanes);
    else hipLaunchKernelGGL((k_tap4<FWD, 2, false>), grid, blk, lds, st, a, nconv, ngrp, eplanes);
  } else {
    if (m1) hipLaunchKernelGGL((k_tap4<FWD, 1, true>), grid, blk, lds, st, a, nconv, ngrp, eplanes);
    else hipLaunchKernelGGL((k_tap4<FWD, 2, true>), grid, blk, lds, st, a, nconv, ngrp, eplanes);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 1 : (int)e;
}

// tapw eligibility; fills the launch geometry.  Returns 1 = eligible.
int tw_plan(const TArgs& a, int KT, int* nconv, int* ch, size_t* lds) {
  if (KT != 3 || (a.stride != 1 && a.stride != 2) || a.narrow <= 0 || a.narrow > 64) return 0;
  if ((a.V1 & 1) || a.V1 > 26 || a.Tout % TW_R || (a.stride == 2 && (a.T & 1))) return 0;
  const long L = (long)a.T * a.V1;
  const int cmax = a.Cin > a.Cout ? a.Cin : a.Cout;
  if ((long)a.n * cmax * L * 4 >= (1L << 31) - 4096) return 0;
  int nc = 0;
  for (int i = 0; i < a.nbr; ++i)
    if (a.br[i].type == 0) {
      ++nc;
      if (a.br[i].dil < 1 || (a.stride == 1 && a.br[i].dil > TW_H)) return 0;
    }
  if (nc == 0) return 0;
  *nconv = nc;
  *ch = a.narrow <= 32 ? 32 : 64;
  const size_t gw = tw_ls(TW_R * a.V1);
  const size_t tile = a.stride == 1 ? (size_t)*ch * (gw + tw_ls((TW_R + 2 * TW_H) * a.V1)) * sizeof(float)
                                    : (size_t)*ch * 4 * gw * sizeof(float);
  const size_t red = (size_t)4 * 3 * 32 * 33 * sizeof(float);
  *lds = std::max(tile, *ch == 32 ? red : (size_t)0);
  return 1;
}

}  // namespace

#define TC_DISPATCH_KT(KTV, CALL) \
  switch (KTV) {                  \
    case 3: { constexpr int KTC = 3; CALL; } break; \
    case 5: { constexpr int KTC = 5; CALL; } break; \
    case 9: { constexpr int KTC = 9; CALL; } break; \
    default: return DSGCN_EUNSUPPORTED;             \
  }

// preload depth by widest window: 8*NG channels; KT*4*NG registers must stay modest (<= 80)
#define TC_LAUNCH_NG(KERNEL)                                                                                         \
  {                                                                                                                   \
    const int w_ = a.narrow;                                                                                          \
    constexpr int cap_ = 80 / (KTC * 4);                                                                              \
    if (w_ > 0 && w_ <= 16 && cap_ >= 2)                                                                              \
      hipLaunchKernelGGL((KERNEL<KTC, (cap_ >= 2 ? 2 : 0)>), grid, dim3(ntb), lds, (hipStream_t)stream, a);          \
    else if (w_ > 0 && w_ <= 24 && cap_ >= 3)                                                                         \
      hipLaunchKernelGGL((KERNEL<KTC, (cap_ >= 3 ? 3 : 0)>), grid, dim3(ntb), lds, (hipStream_t)stream, a);          \
    else if (w_ > 0 && w_ <= 32 && cap_ >= 4)                                                                         \
      hipLaunchKernelGGL((KERNEL<KTC, (cap_ >= 4 ? 4 : 0)>), grid, dim3(ntb), lds, (hipStream_t)stream, a);          \
    else if (w_ > 0 && w_ <= 48 && cap_ >= 6)                                                                         \
      hipLaunchKernelGGL((KERNEL<KTC, (cap_ >= 6 ? 6 : 0)>), grid, dim3(ntb), lds, (hipStream_t)stream, a);          \
    else                                                                                                              \
      hipLaunchKernelGGL((KERNEL<KTC, 0>), grid, dim3(ntb), lds, (hipStream_t)stream, a);                            \
  }

extern "C" {

// Window tables are passed as parallel arrays (nbr <= 8): type (0 conv / 1 max3 / 2 copy), first channel in h / in o,
// widths, dilation, weight and bias pointers (conv only).  All conv windows share the kernel size KT (3, 5 or 9).
int dsgcn_tapconv_fwd(const float* h, float* o, int n, int Cin, int Cout, int T, int V1, int stride, int KT, int nbr,
                      const int* type, const int* ci0, const int* co0, const int* cin, const int* cout, const int* dil,
                      const float* const* w, const float* const* b, void* stream) {
  if (!h || !o || n <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V1 <= 0 || stride <= 0 || nbr <= 0 || nbr > TC_MAXBR)
    return DSGCN_EINVAL;
  if ((size_t)n * Cin * T * V1 * 4 >= (size_t)TC_OOB) return DSGCN_EUNSUPPORTED;
  TArgs a = {};
  a.h = h; a.o = o; a.n = n; a.Cin = Cin; a.Cout = Cout; a.T = T; a.V1 = V1; a.stride = stride; a.nbr = nbr;
  a.Tout = (T + stride - 1) / stride;
  const int cp = tc_fill(a, nbr, type, ci0, co0, cin, cout, dil);
  if (cp < 0) return cp;
  for (int i = 0; i < nbr; ++i) {
    a.br[i].w = w ? w[i] : nullptr; a.br[i].b = b ? b[i] : nullptr;
    if (type[i] == 0 && !a.br[i].w) return DSGCN_EINVAL;
  }
  {
    const int rc4 = t4_try<true>(a, KT, (hipStream_t)stream);
    if (rc4 == 1) return 0;
    if (rc4 != 0) return rc4;
  }
  const size_t lds = tc_lds_conv(cp, KT);
  const int ntb = lds > 80 * 1024 ? 512 : TC_NT;   // one workgroup per CU by LDS -> give it 8 waves
  const int pbk = ntb / 2;
  dim3 grid((unsigned)((a.Tout * V1 + pbk - 1) / pbk), (unsigned)n, (unsigned)(nbr * a.dch));
  TC_DISPATCH_KT(KT, {
    static size_t have = 64 * 1024;
    const int rc = tc_raise_lds(k_tapconv_fwd<KTC, 0>, lds, &have);
    if (rc) return rc;
    TC_LAUNCH_NG(k_tapconv_fwd)
  })
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_tapconv_dgrad(const float* h, const float* go, float* dh, int n, int Cin, int Cout, int T, int V1, int stride,
                        int KT, int nbr, const int* type, const int* ci0, const int* co0, const int* cin,
                        const int* cout, const int* dil, const float* const* w, void* stream) {
  if (!h || !go || !dh || n <= 0 || Cin <= 0 || Cout <= 0 || nbr <= 0 || nbr > TC_MAXBR) return DSGCN_EINVAL;
  TArgs a = {};
  a.h = h; a.go = go; a.dh = dh; a.n = n; a.Cin = Cin; a.Cout = Cout; a.T = T; a.V1 = V1; a.stride = stride; a.nbr = nbr;
  a.Tout = (T + stride - 1) / stride;
  if ((size_t)n * Cout * a.Tout * V1 * 4 >= (size_t)TC_OOB) return DSGCN_EUNSUPPORTED;
  const int cp = tc_fill(a, nbr, type, ci0, co0, cin, cout, dil);
  if (cp < 0) return cp;
  for (int i = 0; i < nbr; ++i) {
    a.br[i].w = w ? w[i] : nullptr;
    if (type[i] == 0 && !a.br[i].w) return DSGCN_EINVAL;
  }
  {
    const int rc4 = t4_try<false>(a, KT, (hipStream_t)stream);
    if (rc4 == 1) return 0;
    if (rc4 != 0) return rc4;
  }
  const size_t lds = tc_lds_conv(cp, KT);
  const int ntb = lds > 80 * 1024 ? 512 : TC_NT;
  const int pbk = ntb / 2;
  dim3 grid((unsigned)((T * V1 + pbk - 1) / pbk), (unsigned)n, (unsigned)(nbr * a.dch));
  TC_DISPATCH_KT(KT, {
    static size_t have = 64 * 1024;
    const int rc = tc_raise_lds(k_tapconv_dgrad<KTC, 0>, lds, &have);
    if (rc) return rc;
    TC_LAUNCH_NG(k_tapconv_dgrad)
  })
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// K-split count the weight gradient prefers for this shape (rows of the partial buffer), 0 = no preference (the caller's
// own heuristic applies).  The wide-load kernel wants ~512 workgroups over the conv windows, several units each.
int dsgcn_tapconv_wgrad_splits(int n, int Cin, int Cout, int T, int V1, int stride, int KT, int nbr, const int* type,
                               const int* cin, const int* cout, const int* dil) {
  if (n <= 0 || nbr <= 0 || nbr > TC_MAXBR || !type || !cin || !cout || !dil) return 0;
  TArgs a = {};
  a.n = n; a.Cin = Cin; a.Cout = Cout; a.T = T; a.V1 = V1; a.stride = stride; a.nbr = nbr;
  a.Tout = stride > 0 ? (T + stride - 1) / stride : T;
  int wmax = 0;
  for (int i = 0; i < nbr; ++i) {
    a.br[i].type = type[i]; a.br[i].cin = cin[i]; a.br[i].cout = cout[i]; a.br[i].dil = dil[i];
    if (type[i] == 0) wmax = std::max(wmax, std::max(cin[i], cout[i]));
  }
  a.narrow = wmax;
  int nconv = 0, ch = 0;
  size_t lds = 0;
  if (!tw_plan(a, KT, &nconv, &ch, &lds)) return 0;
  const int units = n * (a.Tout / TW_R);
  int splits = (ch == 32 ? 512 : 256) / nconv;
  if (splits < 1) splits = 1;
  if (splits > units) splits = units;
  return splits;
}

// Conv window i writes split s of its weight / bias partials at dwp[i] + s*pstride / dbp[i] + s*pstride (all windows
// may share one (splits, pstride) buffer -> one dsgcn_colsum); NULL entries for the other window types.
int dsgcn_tapconv_wgrad(const float* h, const float* go, int n, int Cin, int Cout, int T, int V1, int stride, int KT,
                        int nbr, const int* type, const int* ci0, const int* co0, const int* cin, const int* cout,
                        const int* dil, float* const* dwp, float* const* dbp, int splits, int pstride, void* stream) {
  if (!h || !go || n <= 0 || Cin <= 0 || Cout <= 0 || nbr <= 0 || nbr > TC_MAXBR || splits <= 0) return DSGCN_EINVAL;
  if (V1 > 26) return DSGCN_EUNSUPPORTED;
  TArgs a = {};
  a.h = h; a.go = go; a.n = n; a.Cin = Cin; a.Cout = Cout; a.T = T; a.V1 = V1; a.stride = stride; a.nbr = nbr;
  a.splits = splits; a.pstride = pstride;
  a.Tout = (T + stride - 1) / stride;
  const int cp = tc_fill(a, nbr, type, ci0, co0, cin, cout, dil);
  if (cp < 0) return cp;
  for (int i = 0; i < nbr; ++i) {
    a.br[i].dwp = dwp ? dwp[i] : nullptr; a.br[i].dbp = dbp ? dbp[i] : nullptr;
    if (type[i] == 0 && (!a.br[i].dwp || !a.br[i].dbp)) return DSGCN_EINVAL;
  }
  {
    int nconv = 0, ch = 0;
    size_t ldsw = 0;
    if (tw_plan(a, KT, &nconv, &ch, &ldsw)) {
      const dim3 gridw((unsigned)splits, (unsigned)nconv);
      if (stride == 1 && ch == 32) {
        static size_t have = 64 * 1024;
        const int rc = tc_raise_lds(k_tapw<32>, ldsw, &have);
        if (rc) return rc;
        hipLaunchKernelGGL(k_tapw<32>, gridw, dim3(TC_NT), ldsw, (hipStream_t)stream, a, nconv);
      } else if (stride == 1) {
        static size_t have = 64 * 1024;
        const int rc = tc_raise_lds(k_tapw<64>, ldsw, &have);
        if (rc) return rc;
        hipLaunchKernelGGL(k_tapw<64>, gridw, dim3(TC_NT), ldsw, (hipStream_t)stream, a, nconv);
      } else if (ch == 32) {
        static size_t have = 64 * 1024;
        const int rc = tc_raise_lds(k_tapw2<32>, ldsw, &have);
        if (rc) return rc;
        hipLaunchKernelGGL(k_tapw2<32>, gridw, dim3(TC_NT), ldsw, (hipStream_t)stream, a, nconv);
      } else {
        static size_t have = 64 * 1024;
        const int rc = tc_raise_lds(k_tapw2<64>, ldsw, &have);
        if (rc) return rc;
        hipLaunchKernelGGL(k_tapw2<64>, gridw, dim3(TC_NT), ldsw, (hipStream_t)stream, a, nconv);
      }
      DSGCN_LAUNCH_CHECK();
      return 0;
    }
  }
  int wmax = 0;
  for (int i = 0; i < nbr; ++i)
    if (type[i] == 0) wmax = std::max(wmax, std::max(cin[i], cout[i]));
  if (wmax <= 32 && splits % 4 == 0 && KT <= 5) {
    // narrow windows: one K-split per wave
    const int KPn = (V1 + 1) & ~1, LSn = KPn | 1;
    const size_t ldsn = (size_t)4 * (1 + KT) * 32 * LSn * sizeof(float);
    dim3 gridn((unsigned)(splits / 4), (unsigned)nbr);
    switch (KT) {
      case 3: {
        static size_t have = 64 * 1024;
        const int rc = tc_raise_lds(k_tapconv_wgrad_narrow<3>, ldsn, &have);
        if (rc) return rc;
        hipLaunchKernelGGL(k_tapconv_wgrad_narrow<3>, gridn, dim3(TC_NT), ldsn, (hipStream_t)stream, a);
      } break;
      case 5: {
        static size_t have = 64 * 1024;
        const int rc = tc_raise_lds(k_tapconv_wgrad_narrow<5>, ldsn, &have);
        if (rc) return rc;
        hipLaunchKernelGGL(k_tapconv_wgrad_narrow<5>, gridn, dim3(TC_NT), ldsn, (hipStream_t)stream, a);
      } break;
      default: return DSGCN_EUNSUPPORTED;
    }
    DSGCN_LAUNCH_CHECK();
    return 0;
  }
  const int KP = (2 * V1 + 1) & ~1, LS = KP | 1;
  const size_t lds = (size_t)(1 + KT) * 64 * LS * sizeof(float);
  const bool w8 = lds > 80 * 1024;                          // one workgroup per CU by LDS -> 8 waves
  dim3 grid((unsigned)splits, (unsigned)(nbr * a.dch * a.dch));
  TC_DISPATCH_KT(KT, {
    static size_t have = 64 * 1024;
    static size_t have8 = 64 * 1024;
    if (w8) {
      const int rc = tc_raise_lds(k_tapconv_wgrad<KTC, true>, lds, &have8);
      if (rc) return rc;
      hipLaunchKernelGGL((k_tapconv_wgrad<KTC, true>), grid, dim3(512), lds, (hipStream_t)stream, a);
    } else {
      const int rc = tc_raise_lds(k_tapconv_wgrad<KTC, false>, lds, &have);
      if (rc) return rc;
      hipLaunchKernelGGL((k_tapconv_wgrad<KTC, false>), grid, dim3(TC_NT), lds, (hipStream_t)stream, a);
    }
  })
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
