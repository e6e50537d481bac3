/* libdsgcn_lab — measurement-only entry points (tools/): A/B variants of superseded kernels, launch-geometry knobs and
 * a raw MFMA issue-rate probe.  NOT part of the product ABI: ``libdsgcn.so`` does not export these; they exist only in
 * ``libdsgcn_lab.so``, the same sources compiled with -DDSGCN_LAB (``dsgcn_amd.native.build(lab=True)``). */
#ifndef DSGCN_LAB_H_
#define DSGCN_LAB_H_
#include "dsgcn.h"
#ifdef __cplusplus
extern "C" {
#endif

/* A/B measurement only: the first (scalar-cache + VALU) formulation of the same product. */
int dsgcn_aggregate_fwd_valu(const float* zp, const float* scale, const float* shift, int relu, const float* ahat,
                             float* y, int n, int KC, int T, int V, void* stream);

/* A/B measurement only: variant 0 = product path, 1 = scalar-cache/VALU, 2 = one-shot MFMA. */
int dsgcn_aggregate_fwd_variant(const float* zp, const float* scale, const float* shift, int relu,
                                const float* ahat, float* y, int n, int KC, int T, int V, int variant, void* stream);

/* Tuning knobs (key 0: persistent waves of K-A forward; 0 = default). */
int dsgcn_set_tuning(int key, int value);

int dsgcn_diag_mfma_probe(float* out, int blocks, int iters, int nacc, void* stream);

/* tuning / ablation knobs used by tools/ (key 0: ablation mask, key 1: max 32-channel tiles per block) */
int dsgcn_pwconv_tuning(int key, int value);

/* launch-geometry knobs of K-A' (key 0: pipelined kernels on/off, 1: forward waves, 2: backward workgroups) */
int dsgcn_aggsum_tuning(int key, int value);

/* fused temporal stage (key 0: ablation mask 1 no staging / 2 no MFMA / 4 no epilogue, 1: workgroups per window, 2: frames per tile) */
int dsgcn_tms_tuning(int key, int value);

/* K-B backward: wall-clock stamps (10 ns units) at the phase boundaries of sample 0's three workgroups, out[3][8]
 * (subset-major; phases: start, prepare, pass 1, masked sums, Gram backward, GEMMs, end). */
int dsgcn_dynadj_phases(long long* out);

/* K-C pre-split GEMM form (k_pwg2): wall-clock stamps (10 ns) of workgroup 0 of the last launch, out[64]: start, operands
 * issued, then (commit done, barrier passed, products done) per 32-channel chunk, main loop drained, epilogue done;
 * out[63] = number of stamps. */
int dsgcn_pwg2_phases(long long* out);
/* k_pwg3 stamps through the same buffer: start, affine table visible, chunk 0 committed, then (products done, barrier passed)
 * per chunk, loop drained, epilogue done.  dsgcn_pwg2_phases_block selects the stamping workgroup (default 0). */
int dsgcn_pwg2_phases_block(int block);

/* one-pass narrow backward (k_bwd64): stamps of workgroup 0: start, tables ready, then per 64-position unit (committed, products
 * done), loop done, partial rows written; out[63] = count. */
int dsgcn_bwd64_phases(long long* out);

/* temporal-conv weight gradient (k_tcw): stamps of workgroup 0, per tap group (before issue, after issue, after the
 * barrier, after the products); out[63] = count. */
int dsgcn_tcw_phases(long long* out);
/* dense temporal conv (k_tcg): key 0 = stride-1 staging with 16-byte loads on (1, default) / off (0: the 4-byte form) */
int dsgcn_tconv_tuning(int key, int value);
/* fuse_out backward: key 0 = the 16-byte form on (1, default) / off (0: the 4-byte form everywhere) */
int dsgcn_fuse_out_tuning(int key, int value);
/* split-layout temporal stage (k_tsp): key 0 = bit mask of parts to SKIP for timing (1 plane blocks, 2 epilogue statistics,
 * 4 the affine + ReLU of the forward operand, 8 conv blocks, ...; results are wrong with any bit set); key 1 = the conv
 * block whose phases are stamped */
int dsgcn_tms_split_tuning(int key, int value);
/* which 0: weight gradient (k_tspw), stamps of workgroup (0, 0): start, then per unit (committed + barrier, next unit
 * requested, products + barrier), partial rows written; out[63] = count (64 values).
 * which 1: one V-column conv block of the LAST k_tsp launch: start, requests issued, staging barrier, staged, main loop done,
 * epilogue done; out[15] = count (16 values). */
int dsgcn_tms_split_phases(int which, long long* out);

#ifdef __cplusplus
}
#endif
#endif
