// K-D (core): the temporal branches of dgmstcn between branch_act and combine, all in one launch per direction
// (reference: pyskl/models/gcns/utils/tcn.py:379-398, 410-415):
//   conv  branch : o[n,c0+co,t',x] = b[co] + sum_tap sum_ci W[co,ci,tap] * h[n,c0+ci, t'*s + (tap-KT/2)*d, x]   (zero pad)
//                  = unit_tcn(kernel (KT,1), dilation d, stride s, norm=None)   tcn.py:21-28,393-396
//   max   branch : o = max over the valid taps of a (3,1) window, stride s, pad 1                                   tcn.py:387-390
//   copy  branch : o = h[..., ::s, :]   (the '1x1' branch: its conv is fused upstream in K-C)                         tcn.py:383
// h is the activated tensor (n, C, T, V1) written by k_branch_act (V1 = V+1: the global-joint column rides along),
// o is (n, C, T', V1): every branch writes its own channel window, so there is no torch.cat.
//
// conv branches run on the f32 matrix core, wave-independent like K-C: lane = output position, B operand = h read
// straight from HBM/L2 with raw buffer loads at the tap-shifted row (out-of-range rows -> buffer OOB -> 0),
// A operand = the branch's whole weight tensor staged once in LDS as [co][tap*bcp + ci].
// Backward: dgrad = the transposed gather (same kernel shape over input positions), wgrad = per-tap GEMM over positions
// with LDS-staged do / shifted-h tiles, K-split partials reduced by dsgcn_colsum.
#include "common.h"

namespace {

constexpr int TC_NT = 256;
constexpr int TC_MAXBR = 8;
constexpr int TC_OOB = 0x7ffffff0;

struct TBranch {
  int type;            // 0 conv, 1 max, 2 copy
  int c0, bc, dil;
  const float* w;      // (bc, bc, KT, 1)
  const float* b;      // (bc)
  float* dwp;          // (splits, bc*bc*KT) partials
  float* dbp;          // (splits, bc)
};

struct TArgs {
  const float* h;      // (n, C, T, V1)
  float* o;            // (n, C, Tout, V1)
  const float* go;     // grad of o
  float* dh;           // grad of h
  int n, C, T, Tout, V1, stride, nbr, splits, pstride;
  TBranch br[TC_MAXBR];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tc_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float tc_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ int tc_row32(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// One MFMA pass: acc[m] += A(Ws rows 32m.., column kbase+2u+half) * B(loaded values), 4 k-steps.
// FWD: A[i=co][k] = Ws[co*S + kcol];  BWD (transposed): A[i=ci][k=co] = Ws[k*S + tap*bcp + i]
template <int KT, bool FWD>
__device__ __forceinline__ void tc_conv(const TArgs& a, const TBranch& br, float* Ws, int n, int tile, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const int V1 = a.V1;
  const int bc = br.bc;
  const int bcp = (bc + 7) & ~7;                 // channels per tap padded to a multiple of 8 (two 4-step groups)
  const int S = KT * bcp + 1;                    // LDS row stride (odd)
  const int Lout = a.Tout * V1, Lin = a.T * V1;
  const int L = FWD ? Lout : Lin;
  const int pos = tile * 32 + l31;
  const bool pok = pos < L;
  const int pc = pok ? pos : 0;
  const int row = pc / V1, col = pc - row * V1;
  // source tensor: forward reads h (T rows), backward reads go (Tout rows)
  const int Tsrc = FWD ? a.T : a.Tout;
  const size_t src_bytes = (size_t)a.n * a.C * Tsrc * V1 * 4;
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
    voff[tap] = ok ? (int)((((size_t)n * a.C + br.c0 + half) * Tsrc + rs_row) * V1 + col) * 4 : TC_OOB;
  }
  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
  const int mtiles = (bc + 31) / 32;
#pragma unroll
  for (int tap = 0; tap < KT; ++tap) {
    float xa[4], xb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xa[u] = tc_load(rs, voff[tap], (2 * u) * cstride4);
    for (int k0 = 0; k0 < bcp; k0 += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) xb[u] = tc_load(rs, voff[tap], (k0 + 8 + 2 * u) * cstride4);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kl = k0 + 2 * u + half;
        const float bv = kl < bc ? xa[u] : 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          if (m < mtiles) {
            const float av = FWD ? Ws[(32 * m + l31) * S + tap * bcp + kl] : Ws[kl * S + tap * bcp + 32 * m + l31];
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
          }
        }
      }
      if (k0 + 8 < bcp) {
#pragma unroll
        for (int u = 0; u < 4; ++u) xa[u] = tc_load(rs, voff[tap], (k0 + 16 + 2 * u) * cstride4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kl = k0 + 8 + 2 * u + half;
          const float bv = kl < bc ? xb[u] : 0.f;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            if (m < mtiles) {
              const float av = FWD ? Ws[(32 * m + l31) * S + tap * bcp + kl] : Ws[kl * S + tap * bcp + 32 * m + l31];
              acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  // D[i=channel][j=position]
  float* dst = FWD ? a.o : a.dh;
  const int Tdst = FWD ? a.Tout : a.T;
  if (pok) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = 32 * m + tc_row32(r, half);
        if (ch < bc) {
          const float bias = (FWD && br.b) ? br.b[ch] : 0.f;
          dst[((size_t)(n * a.C + br.c0 + ch) * Tdst) * V1 + pos] = acc[m][r] + bias;
        }
      }
    }
  }
}

// weights (bc, bc, KT) -> LDS [co][tap*bcp + ci] (zero padded); all loads of a thread are issued in batches of 8
template <int KT>
__device__ __forceinline__ void tc_stage_w(const TBranch& br, float* Ws) {
  const int bc = br.bc, bcp = (bc + 7) & ~7, S = KT * bcp + 1;
  const int rows = (bc + 31) / 32 * 32;
  // zero the padded tile first (cheap: <= 64 x 145 floats), then scatter the real weights
  for (int i = threadIdx.x; i < (rows > bcp ? rows : bcp) * S; i += TC_NT) Ws[i] = 0.f;
  __syncthreads();
  const int total = bc * bc * KT;
  for (int i0 = threadIdx.x; i0 < total; i0 += TC_NT * 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * TC_NT;
      v[q] = i < total ? br.w[i] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * TC_NT;
      if (i < total) {
        const int co = i / (bc * KT), r = i - co * bc * KT, ci = r / KT, tap = r - ci * KT;
        Ws[co * S + tap * bcp + ci] = v[q];
      }
    }
  }
  __syncthreads();
}

template <int KT>
__global__ __launch_bounds__(TC_NT) void k_tapconv_fwd(TArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const TBranch& br = a.br[blockIdx.z];
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int V1 = a.V1, Lout = a.Tout * V1;
  if (br.type == 0) {
    tc_stage_w<KT>(br, lds);
    tc_conv<KT, true>(a, br, lds, n, blockIdx.x * 4 + wave, lane);
    return;
  }
  // elementwise branches: thread = output position, loop over the branch's channels
  const int pos = blockIdx.x * 128 + (threadIdx.x & 127);
  if (pos >= Lout) return;
  const int tp = pos / V1, col = pos - tp * V1;
  for (int c = threadIdx.x >> 7; c < br.bc; c += 2) {
    const float* hp = a.h + ((size_t)(n * a.C + br.c0 + c) * a.T) * V1 + col;
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
    a.o[((size_t)(n * a.C + br.c0 + c) * a.Tout) * V1 + pos] = v;
  }
}

template <int KT>
__global__ __launch_bounds__(TC_NT) void k_tapconv_dgrad(TArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const TBranch& br = a.br[blockIdx.z];
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int V1 = a.V1, Lin = a.T * V1;
  if (br.type == 0) {
    tc_stage_w<KT>(br, lds);
    tc_conv<KT, false>(a, br, lds, n, blockIdx.x * 4 + wave, lane);
    return;
  }
  const int pos = blockIdx.x * 128 + (threadIdx.x & 127);
  if (pos >= Lin) return;
  const int t = pos / V1, col = pos - t * V1;
  for (int c = threadIdx.x >> 7; c < br.bc; c += 2) {
    const float* hp = a.h + ((size_t)(n * a.C + br.c0 + c) * a.T) * V1 + col;
    const float* gp = a.go + ((size_t)(n * a.C + br.c0 + c) * a.Tout) * V1 + col;
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
    a.dh[((size_t)(n * a.C + br.c0 + c) * a.T) * V1 + pos] = g;
  }
}

// wgrad: grid = (splits, nbr_conv).  Block: all (co, ci) of one branch (<= 64 x 64), KT accumulators per wave tile.
// Chunk = (sample, 2 output frames): Ds[64][KP] = do, Xs[tap][64][KP] = h at the tap-shifted frames.
template <int KT>
__global__ __launch_bounds__(TC_NT) void k_tapconv_wgrad(TArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const TBranch& br = a.br[blockIdx.y];
  if (br.type != 0) return;
  const int V1 = a.V1, bc = br.bc;
  constexpr int TRW = 2;
  const int KP = (TRW * V1 + 1) & ~1, LS = KP | 1;
  float* Ds = lds;                                // [64][LS]
  float* Xs = lds + 64 * LS;                      // [KT][64][LS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int mt = wave >> 1, nt = wave & 1;
  const int nb = (a.Tout + TRW - 1) / TRW;
  const int total = a.n * nb;
  const int per = (total + a.splits - 1) / a.splits;
  const int ch0 = blockIdx.x * per, ch1 = min(total, ch0 + per);
  f32x16 acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  float dbacc = 0.f;
  const int row = tid >> 2, quarter = tid & 3;    // staging: 4 threads per channel row
  constexpr int NPT = 14;                         // positions per thread: 4*14 = 56 >= KP for V1 <= 26
  float dv[NPT], xv[KT][NPT];
  auto issue = [&](int ch) {                      // all global loads of a chunk, issued before the previous chunk's MFMAs
    const int n = ch / nb, r0 = (ch - n * nb) * TRW;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int p = quarter + 4 * j;
      const int rl = p / V1, col = p - rl * V1;
      const int tp = r0 + rl;
      const bool live = p < TRW * V1 && tp < a.Tout && row < bc;
      dv[j] = live ? a.go[((size_t)(n * a.C + br.c0 + row) * a.Tout + tp) * V1 + col] : 0.f;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const int t = tp * a.stride + (k - KT / 2) * br.dil;
        xv[k][j] = (live && t >= 0 && t < a.T) ? a.h[((size_t)(n * a.C + br.c0 + row) * a.T + t) * V1 + col] : 0.f;
      }
    }
  };
  if (ch0 < ch1) issue(ch0);
  for (int ch = ch0; ch < ch1; ++ch) {
    __syncthreads();
    float dsum = 0.f;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int p = quarter + 4 * j;
      if (p < KP) {
        Ds[row * LS + p] = dv[j];
#pragma unroll
        for (int k = 0; k < KT; ++k) Xs[(k * 64 + row) * LS + p] = xv[k][j];
        dsum += dv[j];
      }
    }
    dsum += __shfl_xor(dsum, 1, 64);
    dsum += __shfl_xor(dsum, 2, 64);
    dbacc += dsum;
    __syncthreads();
    if (ch + 1 < ch1) issue(ch + 1);
    for (int kk = 0; kk < KP; kk += 2) {
      const float av = Ds[(32 * mt + l31) * LS + kk + half];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const float bv = Xs[(k * 64 + 32 * nt + l31) * LS + kk + half];
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[k], 0, 0, 0);
      }
    }
  }
  // D[i=co][j=ci] per tap -> dwp[split][(co*bc + ci)*KT + tap]
  const int ci = 32 * nt + l31;
  float* dwp = br.dwp + (size_t)blockIdx.x * a.pstride;
  if (ci < bc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = 32 * mt + tc_row32(r, half);
      if (co < bc) {
#pragma unroll
        for (int k = 0; k < KT; ++k) dwp[((size_t)co * bc + ci) * KT + k] = acc[k][r];
      }
    }
  }
  if (quarter == 0 && row < bc) br.dbp[(size_t)blockIdx.x * a.pstride + row] = dbacc;
}

size_t tc_lds_conv(int bcmax, int KT) {
  const int bcp = (bcmax + 7) & ~7, S = KT * bcp + 1;
  const int rows = (bcmax + 31) / 32 * 32;
  return (size_t)((rows > bcp ? rows : bcp) + 1) * S * sizeof(float);
}

}  // namespace

extern "C" {

// Branch tables are passed as parallel arrays (nbr <= 8): type (0 conv / 1 max3 / 2 copy), c0, bc, dil, weight and
// bias pointers (conv only).  All conv branches share the kernel size KT (3 for dgmstcn).
int dsgcn_tapconv_fwd(const float* h, float* o, int n, int C, int T, int V1, int stride, int KT, int nbr,
                      const int* type, const int* c0, const int* bc, const int* dil, const float* const* w,
                      const float* const* b, void* stream) {
  if (!h || !o || n <= 0 || C <= 0 || T <= 0 || V1 <= 0 || stride <= 0 || nbr <= 0 || nbr > TC_MAXBR) return DSGCN_EINVAL;
  if (KT != 3) return DSGCN_EUNSUPPORTED;
  TArgs a = {};
  a.h = h; a.o = o; a.n = n; a.C = C; a.T = T; a.V1 = V1; a.stride = stride; a.nbr = nbr;
  a.Tout = (T + stride - 1) / stride;
  int bcmax = 1;
  for (int i = 0; i < nbr; ++i) {
    a.br[i].type = type[i]; a.br[i].c0 = c0[i]; a.br[i].bc = bc[i]; a.br[i].dil = dil[i];
    a.br[i].w = w ? w[i] : nullptr; a.br[i].b = b ? b[i] : nullptr;
    if (type[i] == 0) { if (bc[i] > 64) return DSGCN_EUNSUPPORTED; if (bc[i] > bcmax) bcmax = bc[i]; }
  }
  const size_t lds = tc_lds_conv(bcmax, KT);
  dim3 grid((unsigned)((a.Tout * V1 + 127) / 128), (unsigned)n, (unsigned)nbr);
  hipLaunchKernelGGL(k_tapconv_fwd<3>, grid, dim3(TC_NT), lds, (hipStream_t)stream, a);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

int dsgcn_tapconv_dgrad(const float* h, const float* go, float* dh, int n, int C, int T, int V1, int stride, int KT,
                        int nbr, const int* type, const int* c0, const int* bc, const int* dil, const float* const* w,
                        void* stream) {
  if (!h || !go || !dh || n <= 0 || nbr <= 0 || nbr > TC_MAXBR) return DSGCN_EINVAL;
  if (KT != 3) return DSGCN_EUNSUPPORTED;
  TArgs a = {};
  a.h = h; a.go = go; a.dh = dh; a.n = n; a.C = C; a.T = T; a.V1 = V1; a.stride = stride; a.nbr = nbr;
  a.Tout = (T + stride - 1) / stride;
  int bcmax = 1;
  for (int i = 0; i < nbr; ++i) {
    a.br[i].type = type[i]; a.br[i].c0 = c0[i]; a.br[i].bc = bc[i]; a.br[i].dil = dil[i];
    a.br[i].w = w ? w[i] : nullptr;
    if (type[i] == 0) { if (bc[i] > 64) return DSGCN_EUNSUPPORTED; if (bc[i] > bcmax) bcmax = bc[i]; }
  }
  const size_t lds = tc_lds_conv(bcmax, KT);
  dim3 grid((unsigned)((T * V1 + 127) / 128), (unsigned)n, (unsigned)nbr);
  hipLaunchKernelGGL(k_tapconv_dgrad<3>, grid, dim3(TC_NT), lds, (hipStream_t)stream, a);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

// Conv branch i writes split s of its weight / bias partials at dwp[i] + s*pstride / dbp[i] + s*pstride (all branches
// may share one (splits, pstride) buffer -> one dsgcn_colsum); NULL entries for the other branch types.
int dsgcn_tapconv_wgrad(const float* h, const float* go, int n, int C, int T, int V1, int stride, int KT, int nbr,
                        const int* type, const int* c0, const int* bc, const int* dil, float* const* dwp,
                        float* const* dbp, int splits, int pstride, void* stream) {
  if (!h || !go || n <= 0 || nbr <= 0 || nbr > TC_MAXBR || splits <= 0) return DSGCN_EINVAL;
  if (KT != 3 || V1 > 26) return DSGCN_EUNSUPPORTED;
  TArgs a = {};
  a.h = h; a.go = go; a.n = n; a.C = C; a.T = T; a.V1 = V1; a.stride = stride; a.nbr = nbr; a.splits = splits;
  a.pstride = pstride;
  a.Tout = (T + stride - 1) / stride;
  for (int i = 0; i < nbr; ++i) {
    a.br[i].type = type[i]; a.br[i].c0 = c0[i]; a.br[i].bc = bc[i]; a.br[i].dil = dil[i];
    a.br[i].dwp = dwp ? dwp[i] : nullptr; a.br[i].dbp = dbp ? dbp[i] : nullptr;
    if (type[i] == 0 && (bc[i] > 64 || !a.br[i].dwp || !a.br[i].dbp)) return DSGCN_EUNSUPPORTED;
  }
  const int KP = (2 * V1 + 1) & ~1, LS = KP | 1;
  const size_t lds = (size_t)(1 + KT) * 64 * LS * sizeof(float);
  dim3 grid((unsigned)splits, (unsigned)nbr);
  hipLaunchKernelGGL(k_tapconv_wgrad<3>, grid, dim3(TC_NT), lds, (hipStream_t)stream, a);
  DSGCN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
