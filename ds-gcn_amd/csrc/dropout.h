// Counter-based dropout masks for the fused block output (fuseout.hip): Philox4x32-10 keyed by the run's seed, counted by
// (element group, training step, call) — no mask tensor exists, the backward regenerates what the forward used.
// Reference semantics: nn.Dropout(p, inplace=True) behind the temporal unit's BatchNorm (tcn.py:30,33; MSTCN
// msg3d_utils.py:141-146): y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) per element.  (torch's own Philox stream is a
// different counter layout: masks are not bit-identical to torch's, their distribution is.)
#pragma once
#include "common.h"

struct DropArgs {
  const long long* step;           // device counter of training steps (NULL = 0): the same value in a step's forward and backward
  unsigned long long seed;
  unsigned call;                   // which fuse_out call of the step (host counter, frozen into a captured graph)
  unsigned thresh;                 // keep iff random word >= thresh = p * 2^32;  0 = dropout off
  float inv;                       // 1 / (1 - p)
};

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// the four random words of element group g (elements 4g .. 4g+3 of the flat tensor)
__device__ __forceinline__ void drop_words(const DropArgs& d, unsigned long long step, unsigned long long g, unsigned (&r)[4]) {
  philox4x32_10((unsigned)g, (unsigned)(g >> 32), (unsigned)step, d.call ^ ((unsigned)(step >> 32) * 0x9E3779B9u),
                (unsigned)d.seed, (unsigned)(d.seed >> 32), r);
}

// multiplier of flat element e: 1/(1-p) if kept, else 0
__device__ __forceinline__ float drop_mult(const DropArgs& d, unsigned long long step, unsigned long long e) {
  unsigned r[4];
  drop_words(d, step, e >> 2, r);
  return r[e & 3] >= d.thresh ? d.inv : 0.f;
}
