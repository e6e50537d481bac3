/* Job records of the batched BatchNorm micro-reductions (included by dsgcn.h; plain C, no dependencies).
 * A record is the argument list of dsgcn_bn_finalize / dsgcn_bn_coef_rows as a struct, so that several of them can ride
 * in one launch (dsgcn_bn_finalize_multi, dsgcn_bn_coef_rows_multi) or as extra workgroups of a launch that sits between
 * the BatchNorm's producer and its consumer anyway (dsgcn_dynadj_fwd_jobs / _bwd_jobs). */
#ifndef DSGCN_JOBS_H
#define DSGCN_JOBS_H

#define DSGCN_BN_JOBS_MAX 4

typedef struct {
  const float* partial;            /* (nblk, C, 2) per-block sums of z and z^2 */
  const float* gamma;              /* (C) or NULL = 1 */
  const float* beta;               /* (C) or NULL = 0 */
  float* mean; float* var;         /* (C) out: batch mean, biased variance */
  float* scale; float* shift;      /* (C) out: gamma*rsqrt(var+eps), beta - mean*scale (identity for c >= c_affine) */
  double count;                    /* elements per channel */
  float eps;
  int nblk, C, c_affine;
} dsgcn_bn_fin_job;

typedef struct {
  const float* part;               /* (R, C, k) a consumer's partial rows */
  const float* mean; const float* var;
  const float* gamma;              /* (C) or NULL = 1 */
  float* coef;                     /* (4, C) out / in-out: [d gamma | d beta | A0 | B0] */
  double count;
  float eps;
  int R, C, k, i_ds, i_dh;         /* columns of part holding the partial sums of d scale / d shift */
  int c_affine, accumulate;        /* accumulate != 0: add to coef (a second consumer of the same BatchNorm) */
} dsgcn_bn_coef_job;

/* Dropout of the fused block output's first term (dsgcn_fuse_out_fwd_drop / _bwd_drop): counter-based masks, no mask
 * tensor.  step: DEVICE counter of training steps (NULL = 0; the same value must be in it for a step's forward and its
 * backward); seed: the run's seed; call: which fuse_out call of the step; p in [0, 1): drop probability (0 = off). */
typedef struct {
  const long long* step;
  unsigned long long seed;
  unsigned int call;
  float p;
} dsgcn_dropout;

/* The parameter-only launches of the CTR-GCN units (dsgcn_ctr_wprep / dsgcn_ctr_wfin, dsgcn.h) as records, so that all
 * units of a model share one launch at the head of the step (dsgcn_ctr_wprep_multi) and one at the end of the backward
 * (dsgcn_ctr_wfin_multi).  Fields as the arguments of the single-unit entry points; K <= 4. */
#define DSGCN_CTR_JOBS_MAX 16

typedef struct {
  const float* w[4];               /* K weights (Co, R) */
  const float* b[4];               /* K biases (Co) or NULL */
  const float* alpha;              /* (1) */
  float* wout;                     /* (K, Co, R + 2) out */
  float* sh;                       /* (K, 2, R + 2) out */
  int K, Co, R, reserved;
} dsgcn_ctr_prep_job;

typedef struct {
  const float* dwp[4];             /* K gradients of W'_k (Co, R + 2) or NULL */
  const float* ds[4];              /* K gradients of the input scales (R + 2 elements, ds_stride apart) or NULL */
  float* out[4];                   /* K buffers (Co*R + Co) = [dW_k | db_k] */
  float* dalpha;                   /* (1) out */
  int K, Co, R, ds_stride;
} dsgcn_ctr_fin_job;

/* The per-tap weight images of the dense temporal conv (dsgcn_tconv_wsplit, dsgcn.h) as records: all convs of a model in
 * one launch at the head of the step (dsgcn_tconv_wsplit_multi). */
#define DSGCN_TSPLIT_JOBS_MAX 16

typedef struct {
  const float* w;                  /* (Co, Ci, KT, 1) */
  void* ws;                        /* dsgcn_tconv_ws_bytes(...) bytes out */
  int Ci, Co, KT, reserved;
} dsgcn_tsplit_job;

#endif
