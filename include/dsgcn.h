/* libdsgcn — C ABI of the MI355X-native DS-GCN hot path.
 *
 * The reference (davelailai/DS-GCN, a PYSKL fork) has no FFI: its hot path is Python calling ATen
 * (SURVEY.md §2.2).  The drop-in boundary is therefore the PYSKL registry/config API (mirrored in
 * Python by ds-gcn_amd/), and THIS header is the native boundary underneath it: each entry point
 * replaces the group of ATen calls cited next to it.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - return int: 0 = ok; >0 = hipError_t of a failed launch; DSGCN_EINVAL (-1) = bad argument;
 *     DSGCN_EUNSUPPORTED (-2) = shape outside the compiled variants.  Never throws.
 *   - pointers are DEVICE pointers to contiguous fp32 (or int32 where noted) buffers owned by the
 *     caller; nothing is allocated or freed; no host/device synchronisation happens inside.
 *   - `stream` is a hipStream_t (NULL = default stream); launches are asynchronous on it; calls on
 *     distinct streams are thread-safe.
 *   - layouts: activations (n, C, T, V) with V fastest ("NCHW"), adjacency (n, C, V, V).
 */
#ifndef DSGCN_H_
#define DSGCN_H_

#include <stddef.h>
#include "dsgcn_jobs.h"
#ifdef __cplusplus
extern "C" {
#endif

#define DSGCN_EINVAL (-1)
#define DSGCN_EUNSUPPORTED (-2)

/* ABI version (major*100+minor). */
int dsgcn_version(void);

/* K-A gather-aggregate.  Replaces torch.einsum('nkctv,nkcvw->nkctw', pre_x, A)
 * (reference pyskl/models/gcns/utils/gcn.py:2350-2352) fused with the BN+ReLU of `pre`
 * (gcn.py:2165-2167,2236):   y[b,t,w] = sum_u relu?(zp[b,t,u]*scale[c]+shift[c]) * ahat[b,u,w],
 * b = (sample, channel c) over n*KC units.  scale/shift may be NULL (identity).  V in {25,17,18}. */
int dsgcn_aggregate_fwd(const float* zp, const float* scale, const float* shift, int relu, const float* ahat,
                        float* y, int n, int KC, int T, int V, void* stream);

/* Backward of the above: dzp (n,KC,T,V), dahat (n,KC,V,V) and partial (n*KC,2) =
 * per-unit [sum dP*mask*zp, sum dP*mask] (the d scale / d shift reductions before the sum over n). */
/* rows of dsgcn_aggregate_bwd's `partial` buffer (rows, KC, 2): n, or 2n when a unit is split over two waves */
int dsgcn_aggregate_bwd_partial_rows(int n, int T, int V);
int dsgcn_aggregate_bwd(const float* zp, const float* scale, const float* shift, int relu, const float* ahat,
                        const float* dy, float* dzp, float* dahat, float* partial, int n, int KC, int T, int V,
                        void* stream);

/* K-B dynamic-semantic adjacency.  Replaces gcn.py:2240-2337 (node-typed select of conv1_se 2256-2259, the edge-typed
 * linear with its 625-iteration index loop 2279-2288, tanh 2298, Gram + Softmax(-2) 2314-2326, alpha/beta scale-add
 * 2304-2337).  K = 3 subsets.
 *   proj (n,(4+P)*mid,ld): rows [conv1 (2mid) | conv2 (2mid) | conv1_se (mid*P, row c*P+p)] applied to the time-mean
 *        of the unit input (computed by dsgcn_pwconv_fwd on xbar viewed as (n,Ci,1,ld)); ld >= V is the joint stride of a
 *        row (32: rows zero-padded so the K-C kernels take their 16-byte-per-lane path);
 *   A (3,V,V); alpha,beta (3);  we (E*mid,mid) row e*mid+c, be (E*mid)
 *   node_type (V) int32 in [0,P);  edge_type (V*V) int32 in [0,E);  ahat out (n,3*mid,V,V).  V <= 32, mid <= 32.
 * Backward: dd_ws workspace (n,3*mid,V,V); outputs dproj (n,(4+P)*mid,ld) (padding columns zero) and ppar (n, pstride >=
 * dsgcn_dynadj_partial_stride): per-sample partials [sum_c dAhat (3*V*V) | dalpha (3) | dbeta (3) | dwe (E*mid*mid) |
 * dbe (E*mid)] — their sum over samples (dsgcn_colsum: ordered, no float atomics) gives dA, dalpha, dbeta, dwe, dbe. */
int dsgcn_dynadj_partial_stride(int mid, int V, int E);
int dsgcn_dynadj_fwd(const float* proj, const float* A, const float* alpha, const float* beta, const float* we,
                     const float* be, const int* node_type, const int* edge_type, float* ahat, int n, int mid, int V,
                     int ld, int P, int E, void* stream);
int dsgcn_dynadj_bwd(const float* proj, const float* alpha, const float* beta, const float* we, const float* be,
                     const int* node_type, const int* edge_type, const float* dahat, float* dd_ws, float* dproj,
                     float* ppar, int pstride, int n, int mid, int V, int ld, int P, int E, void* stream);
/* The same launches carrying BatchNorm jobs as extra workgroups (njobs <= DSGCN_BN_JOBS_MAX, host records): K-B runs between
 * the `pre` conv and K-A (gcn.py:2236-2352), so the `pre` BatchNorm's finalize rides in dsgcn_dynadj_fwd_jobs and its
 * backward coefficients (from K-A's partial rows) in dsgcn_dynadj_bwd_jobs — no launch of their own.  njobs = 0: the plain call. */
int dsgcn_dynadj_fwd_jobs(const float* proj, const float* A, const float* alpha, const float* beta, const float* we,
                          const float* be, const int* node_type, const int* edge_type, float* ahat, int n, int mid, int V,
                          int ld, int P, int E, const dsgcn_bn_fin_job* jobs, int njobs, void* stream);
int dsgcn_dynadj_bwd_jobs(const float* proj, const float* alpha, const float* beta, const float* we, const float* be,
                          const int* node_type, const int* edge_type, const float* dahat, float* dd_ws, float* dproj,
                          float* ppar, int pstride, int n, int mid, int V, int ld, int P, int E,
                          const dsgcn_bn_coef_job* jobs, int njobs, void* stream);

/* Block output (materialise once): out = relu?(x1*s1+h1 (+ x2*s2+h2 | + x2)), xbar = mean_t out (optional).
 * Replaces BN + residual add + ReLU of dgstgcn.py:63-65 / tcn.py:427 and x.mean(-2) of gcn.py:2246.
 * relu: bit 0 = the outer ReLU, bit 1 = a ReLU on the first term before the add (CTR-GCN: msg3d_utils.py:139-141
 * followed by ctrgcn.py:60).  bwd part (n*C,4) = per-plane [sum dv1*x1, sum dv, sum dv*x2, sum dv1].
 * xbar / dxbar are (n, C, xbar_ld) with xbar_ld >= V: the joint row zero-padded (32: the layout the dynamic-adjacency
 * projections run on, so no pad / slice launches sit between the blocks). */
int dsgcn_fuse_out_fwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                       const float* h2, int relu, float* out, float* xbar, int n, int C, int T, int V, int xbar_ld,
                       void* stream);
int dsgcn_fuse_out_bwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                       const float* h2, int relu, const float* dout, const float* dxbar, float* dx1, float* dx2,
                       float* part, int n, int C, int T, int V, int xbar_ld, void* stream);
/* the same with up to three gradients of `out`, (dout + dout2) + dout3 formed while loading: the next block reads its
 * input three times (dgstgcn.py:63-65: gcn(x), the unit's residual operand, the block residual) and autograd would
 * otherwise materialise the sum first (dsgcn_add3).  dout2 / dout3 may be NULL. */
int dsgcn_fuse_out_bwd3(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                        const float* h2, int relu, const float* dout, const float* dout2, const float* dout3,
                        const float* dxbar, float* dx1, float* dx2, float* part, int n, int C, int T, int V, int xbar_ld,
                        void* stream);
/* The even frames as a second output / as the shape of the third gradient stream: a stride-2 block's 1x1 residual conv
 * (reference: the block residual unit_tcn(kernel_size=1, stride=2), dgstgcn.py:35-40 + tcn.py:21-28) reads out[:, :, ::2];
 * dsgcn_fuse_out_fwd2 writes that tensor, out_s2 (n, C, ceil(T/2), V) or NULL, beside `out`, and dsgcn_fuse_out_bwd3s with
 * stride3 = 2 takes dout3 in that shape (added on the even frames) — no strided-copy launch, no zero-filled scatter. */
int dsgcn_fuse_out_fwd2(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                        const float* h2, int relu, float* out, float* out_s2, float* xbar, int n, int C, int T, int V,
                        int xbar_ld, void* stream);
int dsgcn_fuse_out_bwd3s(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                         const float* h2, int relu, const float* dout, const float* dout2, const float* dout3, int stride3,
                         const float* dxbar, float* dx1, float* dx2, float* part, int n, int C, int T, int V, int xbar_ld,
                         void* stream);

/* The last block's output as its (T, V) plane means only (what the head's pooling reads, simple_head.py:88-93): pmean
 * (n, C) = mean_tv relu?(x1*s1+h1 (+ x2*s2+h2 | + x2)) — the activation itself is never written.  Backward: dpmean
 * (n, C), dx1 / dx2 / part as dsgcn_fuse_out_bwd. */
int dsgcn_fuse_out_pool_fwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                            const float* h2, int relu, float* pmean, int n, int C, int T, int V, void* stream);
int dsgcn_fuse_out_pool_bwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                            const float* h2, int relu, const float* dpmean, float* dx1, float* dx2, float* part, int n,
                            int C, int T, int V, void* stream);
/* The block output with DROPOUT on its first term (the temporal unit's output: nn.Dropout behind its BatchNorm, tcn.py:30,33;
 * MSTCN msg3d_utils.py:141-146), masks from a counter-based generator (dsgcn_jobs.h: dsgcn_dropout) — no mask tensor, the
 * backward regenerates it:  out = relu?( D * relu2?(x1*s1+h1) + (x2*s2+h2 | x2) ), D = keep/(1-p).  One entry point per
 * direction covers every output form: out / out_s2 / xbar / pmean may be NULL (pmean (n*C): the plane means of
 * dsgcn_fuse_out_pool_fwd); backward stride3 = 1 | 2 as dsgcn_fuse_out_bwd3s, 3 = dout is the (n*C) gradient of the plane
 * means.  d NULL or p = 0: exactly the plain calls.  dsgcn_dropout_mask writes the multipliers a tensor of numel elements
 * gets under d (tests only: the product never materialises them). */
int dsgcn_fuse_out_fwd_drop(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                            const float* h2, int relu, float* out, float* out_s2, float* xbar, float* pmean, int n, int C,
                            int T, int V, int xbar_ld, const dsgcn_dropout* d, void* stream);
int dsgcn_fuse_out_bwd_drop(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                            const float* h2, int relu, const float* dout, const float* dout2, const float* dout3,
                            int stride3, const float* dxbar, float* dx1, float* dx2, float* part, int n, int C, int T, int V,
                            int xbar_ld, const dsgcn_dropout* d, void* stream);
int dsgcn_dropout_mask(float* mask, long numel, const dsgcn_dropout* d, void* stream);

/* Dense (KT,1) temporal conv as a GEMM on bf16 terms (csrc/tcg.hip): unit_tcn's Conv2d((9,1), padding 4) + the statistics
 * of its BatchNorm (tcn.py:21-28), stride 1 or 2 (T = input frames, z has ceil(T/stride)), dilation 1, KT odd <= 9,
 * V <= 32; the virtual input relu?(x1*s1+h1 (+ x2*s2+h2))
 * is formed while loading and zero-padded in time AFTER the activation.
 *   dsgcn_tconv_ws_bytes : bytes of the pre-split weight image for this shape, 0 = shape not taken (use dsgcn_tapconv_*)
 *   dsgcn_tconv_wsplit   : w (Co, Ci, KT) -> image (per tap the three bf16 terms of W and of the tap-flipped W^T);
 *                          dsgcn_tconv_wsplit_multi: the images of njobs convs in one launch (records: dsgcn_jobs.h)
 *   dsgcn_tconv_rows     : partial rows: which = 0 forward (rows, Co, 2) [sum z, sum z^2], 1 data gradient (rows, Ci, 3)
 *   dsgcn_tconv_fwd      : z (n, Co, T, V) = bias + conv;  dsgcn_tconv_dgrad: dz = gz + A0 + B0*z -> dx1 (, dx2), ipart
 *                          (mask / affine of the forward's virtual input as in dsgcn_pwconv_dgrad). */
size_t dsgcn_tconv_ws_bytes(int n, int Ci, int Co, int T, int V, int KT, int stride);
int dsgcn_tconv_wsplit(const float* w, int Ci, int Co, int KT, void* ws, void* stream);
int dsgcn_tconv_wsplit_multi(const dsgcn_tsplit_job* jobs, int njobs, void* stream);
int dsgcn_tconv_rows(int which, int n, int Ci, int Co, int T, int V, int KT, int stride);
int dsgcn_tconv_fwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2, const float* h2,
                    int relu, const void* ws, const float* bias, float* z, float* partial, int n, int Ci, int Co, int T,
                    int V, int KT, int stride, void* stream);
int dsgcn_tconv_dgrad(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                      const float* h2, int relu, const void* ws, const float* z, const float* gz, const float* A0,
                      const float* B0, float* dx1, float* dx2, float* ipart, int n, int Ci, int Co, int T, int V, int KT,
                      int stride, void* stream);

/* weight gradient of the same conv: K (positions) split over dsgcn_tconv_wgrad_splits workgroup groups; split s writes
 * dwp + s*pstride (Co*Ci*KT floats in the weight's layout) and dbp + s*pstride (Co floats); T*V % 4 == 0. */
int dsgcn_tconv_wgrad_splits(int n, int Ci, int Co, int T, int V, int KT, int stride);
int dsgcn_tconv_wgrad(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                      const float* h2, int relu, const float* z, const float* gz, const float* A0, const float* B0,
                      float* dwp, float* dbp, int pstride, int n, int Ci, int Co, int T, int V, int KT, int stride,
                      void* stream);

/* K-D: dgmstcn temporal stages (tcn.py:379-428).
 * branch_act: h (n,C,T,V+1) = act_c(z*scale+shift) with the global-joint column zaug appended (ReLU for c < n_act).
 * tapconv   : temporal windows over h (n,Cin,T,V1) -> o (n,Cout,T',V1), T' = ceil(T/stride).  Window i: type 0 =
 *             (KT,1) conv with dilation dil[i] (weights (cout,cin,KT,1), bias (cout) or NULL), 1 = (3,1) max-pool,
 *             2 = strided copy; it reads channels [ci0,ci0+cin) of h and writes [co0,co0+cout) of o.  KT in {3,5,9},
 *             shared by the conv windows; window tables are host arrays of length nbr <= 8.  dgmstcn (tcn.py:383-396),
 *             MSTCN (msg3d_utils.py:84-117) and the dense unit_tcn conv of ST-GCN (tcn.py:21-28; one window, any
 *             width) all go through it.  wgrad writes K-split partials at dwp[i]/dbp[i] + split*pstride, laid out
 *             (cout,cin,KT) / (cout).  branch_act with zaug == NULL has no global-joint column (h is (n,C,T,V)).
 * combine   : f = o[..,:V] + o[..,V]*coeff, per-plane sum / sum of squares of f (n*C, 2). */
int dsgcn_branch_act_fwd(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                         float* h, int n, int C, int T, int V, void* stream);
int dsgcn_branch_act_bwd(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                         const float* dh, float* dz, float* dzaug, float* part, int n, int C, int T, int V,
                         void* stream);
int dsgcn_tapconv_fwd(const float* h, float* o, int n, int Cin, int Cout, int T, int V1, int stride, int KT, int nbr,
                      const int* type, const int* ci0, const int* co0, const int* cin, const int* cout, const int* dil,
                      const float* const* w, const float* const* b, void* stream);
int dsgcn_tapconv_dgrad(const float* h, const float* go, float* dh, int n, int Cin, int Cout, int T, int V1, int stride,
                        int KT, int nbr, const int* type, const int* ci0, const int* co0, const int* cin,
                        const int* cout, const int* dil, const float* const* w, void* stream);
/* rows of the partial buffer (K-splits) dsgcn_tapconv_wgrad prefers for this shape; 0 = the caller chooses */
int dsgcn_tapconv_wgrad_splits(int n, int Cin, int Cout, int T, int V1, int stride, int KT, int nbr, const int* type,
                               const int* cin, const int* cout, const int* dil);
int dsgcn_tapconv_wgrad(const float* h, const float* go, int n, int Cin, int Cout, int T, int V1, int stride, int KT,
                        int nbr, const int* type, const int* ci0, const int* co0, const int* cin, const int* cout,
                        const int* dil, float* const* dwp, float* const* dbp, int splits, int pstride, void* stream);
int dsgcn_tms_combine_fwd(const float* o, const float* coeff, float* f, float* partial, int n, int C, int T, int V,
                          void* stream);
int dsgcn_tms_combine_bwd(const float* o, const float* coeff, const float* gf, const float* A0, const float* B0,
                          float* dout, float* pcoef, int n, int C, int T, int V, void* stream);
int dsgcn_pwconv_ipart_rows(int n, int Ci, int Co, int T, int V, int stride);

/* ---- K-C: 1x1 channel mix with fused train-mode BatchNorm / ReLU / residual --------------------------------
 * Replaces Conv2d(1x1)+BatchNorm2d+ReLU(+add) chains of gcn.py:2165-2169,2209-2215,2236,2363-2365 and
 * tcn.py:379-404,409,422,427 and tcn.py:21-28 (kernel_size 1).
 *   v = relu?(x1*s1[ci]+h1[ci] (+ x2*s2[ci]+h2[ci] | + x2));  z = W v[:, :, ::stride] + bias;  zaug = mean_v z (aug)
 *   partial (n*nblk, Co, 2): per-block sum / sum of squares of z (+zaug) -> dsgcn_bn_finalize.
 * x1,x2 (n,Ci,T,V); w (Co,Ci); z (n,Co,Tout,V); zaug (n,Co,Tout); s,h (Ci) or NULL. */
int dsgcn_pwconv_plan(int Tout, int V, int aug, int* TR, int* NPpad, int* nblk_per_sample);
/* rows of the forward's `partial` buffer: (rows, Co, 2) */
int dsgcn_pwconv_partial_rows(int n, int Ci, int Co, int T, int V, int stride, int aug);
int dsgcn_pwconv_fwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                     const float* h2, int relu, const float* w, const float* bias, float* z, float* zaug,
                     float* partial, int n, int Ci, int Co, int T, int V, int stride, int aug, int stats,
                     void* stream);
/* partial (nblk,C,2) -> mean,var (biased), scale = gamma*rsqrt(var+eps), shift = beta-mean*scale for c < c_affine,
 * identity affine above.  count = elements per channel. */
int dsgcn_bn_finalize(const float* partial, int nblk, int C, double count, const float* gamma, const float* beta,
                      float eps, float* mean_out, float* var_out, float* scale_out, float* shift_out, int c_affine,
                      void* stream);
/* out[c] = sum_r src[r,c] (fp64 accumulation). */
int dsgcn_colsum(const float* src, int R, int C, float* out, void* stream);
/* same, result transposed: src (R, C/inner, inner) -> out (inner, C/inner): every one of the `inner` sums contiguous */
int dsgcn_colsum_t(const float* src, int R, int C, int inner, float* out, void* stream);
/* two such reductions in one launch (inner = 1: plain) */
int dsgcn_colsum2(const float* src_a, int Ra, int Ca, int inner_a, float* out_a, const float* src_b, int Rb, int Cb,
                  int inner_b, float* out_b, void* stream);
/* njobs independent column sums in ONE launch: the partial rows of parameter gradients, queued during the backward and
 * finished together before the gradients are packed (they feed only the optimizer).  table: DEVICE array of njobs x 4
 * int64 words {src (R, C) fp32, out (C) fp32, (R << 32) | C, first block of the job}, nblocks = sum over the jobs of
 * dsgcn_colsum_blocks(src, C) (128 columns per block when C % 4 == 0 and src is 16-byte aligned, else 32). */
int dsgcn_colsum_blocks(const float* src, int C);
int dsgcn_colsum_multi(const long* table, int njobs, int nblocks, void* stream);
/* The same with the table in HOST memory: the jobs travel in the kernel arguments (no upload, no device table). */
int dsgcn_colsum_multi_host(const long* table, int njobs, void* stream);
/* BN-statistics backward coefficients: dz_eff = gz + A0[c] + B0[c]*z; also d gamma / d beta. */
int dsgcn_bn_bwd_coef(const float* g_scale, const float* g_shift, const float* mean, const float* var,
                      const float* gamma, float eps, double count, int C, int c_affine, float* dgamma, float* dbeta,
                      float* A0, float* B0, void* stream);
/* The same coefficients from a consumer's partial rows part (R, C, k) (column i_ds: partial sums of d scale, i_dh: of
 * d shift), coef (4, C) = [dgamma | dbeta | A0 | B0]; accumulate: add to coef (second consumer of the same BatchNorm).
 * One launch instead of dsgcn_colsum + dsgcn_bn_bwd_coef. */
int dsgcn_bn_coef_rows(const float* part, int R, int C, int k, int i_ds, int i_dh, const float* mean, const float* var,
                       const float* gamma, float eps, double count, int c_affine, float* coef, int accumulate,
                       void* stream);
/* Up to DSGCN_BN_JOBS_MAX finalize / coefficient jobs (dsgcn_jobs.h: the argument lists above as records, in HOST memory —
 * they travel in the kernel arguments) in ONE launch: BatchNorms whose producers are independent of one another (`post` and
 * `down`, gcn.py:2165-2169,2215; transform and the block's residual conv, tcn.py:401-404 / dgstgcn.py:55-61).  Bit-identical
 * to the single launches (same blocks, same summation order). */
int dsgcn_bn_finalize_multi(const dsgcn_bn_fin_job* jobs, int njobs, void* stream);
int dsgcn_bn_coef_rows_multi(const dsgcn_bn_coef_job* jobs, int njobs, void* stream);
/* dz_eff of a conv with the global-joint column, materialised once: gz + A0 + B0*z + (gzaug + A0 + B0*zaug)/V. */
int dsgcn_dz_eff_aug(const float* gz, const float* z, const float* gzaug, const float* zaug, const float* A0,
                     const float* B0, float* out, int n, int C, int T, int V, void* stream);
/* Data gradient: dx1 (, dx2) fully written; ipart (n*nblk, Ci, 3) = [sum dv*x1, sum dv, sum dv*x2] partials. */
int dsgcn_pwconv_dgrad(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                       const float* h2, int relu, const float* w, const float* z, const float* zaug,
                       const float* gz, const float* gzaug, const float* A0, const float* B0, float* dx1, float* dx2,
                       float* ipart, int n, int Ci, int Co, int T, int V, int stride, int aug, void* stream);
/* Wide convs (GEMM form: both widths > 64): the weights split into their three bf16 terms ONCE per conv and step instead
 * of per workgroup and K chunk.  dsgcn_pwconv_wsplit_bytes: size of that image for this shape, 0 = the shape does not
 * use it.  dsgcn_pwconv_wsplit fills it from w (Co, Ci) (both the W and the W^T image); dsgcn_pwconv_fwd_ws /
 * dsgcn_pwconv_dgrad_ws are dsgcn_pwconv_fwd / dsgcn_pwconv_dgrad with that image (ws = NULL: identical to them).
 * Same results as the plain entry points (the split is the same function of w, only evaluated earlier). */
size_t dsgcn_pwconv_wsplit_bytes(int n, int Ci, int Co, int T, int V, int stride);
int dsgcn_pwconv_wsplit(const float* w, int Ci, int Co, void* ws, void* stream);
/* The same for `njobs` convs in one launch per 32 jobs: w / ws / Ci / Co are HOST arrays of length njobs (device pointers
 * and sizes; read during the call, the job table rides in the kernel arguments). */
int dsgcn_pwconv_wsplit_multi(const float* const* w, void* const* ws, const int* Ci, const int* Co, int njobs,
                              void* stream);
int dsgcn_pwconv_fwd_ws(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                        const float* h2, int relu, const float* w, const float* bias, float* z, float* zaug,
                        float* partial, int n, int Ci, int Co, int T, int V, int stride, int aug, int stats,
                        const void* ws, void* stream);
int dsgcn_pwconv_dgrad_ws(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                          const float* h2, int relu, const float* w, const float* z, const float* zaug,
                          const float* gz, const float* gzaug, const float* A0, const float* B0, float* dx1, float* dx2,
                          float* ipart, int n, int Ci, int Co, int T, int V, int stride, int aug, const void* ws,
                          void* stream);
/* Weight gradient: k-split partials, split s at dwp + s*pstride (Co*Ci floats) and dbp + s*pstride (Co floats);
 * splits from dsgcn_pwconv_wgrad_splits. */
int dsgcn_pwconv_wgrad_splits(int n, int Ci, int Co, int T, int V, int stride);
/* Both gradients in ONE pass over gz, z and the input(s), for narrow convs (stride 1, no global-joint column, Ci and
 * Co <= 64, T*V % 4 == 0): dx1 / dx2 = data gradient through the ReLU mask / affines of the virtual input, dwp/dbp =
 * per-split partials (rows = dsgcn_pwconv_bwd_rows, layout as dsgcn_pwconv_wgrad), ipart (rows, Ci, 3) =
 * [sum dv*x1, sum dv, sum dv*x2] (NULL when the input has no affine).  dsgcn_pwconv_bwd_rows returns 0 for shapes it
 * does not cover (then: dsgcn_pwconv_dgrad + dsgcn_pwconv_wgrad).  Replaces the same reference ops as those two. */
int dsgcn_pwconv_bwd_rows(int n, int Ci, int Co, int T, int V, int stride);
int dsgcn_pwconv_bwd(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2, const float* h2,
                     int relu, const float* w, const float* z, const float* gz, const float* A0, const float* B0,
                     float* dx1, float* dx2, float* ipart, float* dwp, float* dbp, int pstride, int n, int Ci, int Co,
                     int T, int V, void* stream);
int dsgcn_pwconv_wgrad(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                       const float* h2, int relu, const float* z, const float* zaug, const float* gz,
                       const float* gzaug, const float* A0, const float* B0, float* dwp, float* dbp, int pstride, int n,
                       int Ci, int Co, int T, int V, int stride, int aug, void* stream);
/* The same carrying BatchNorm coefficient jobs of the conv's INPUT BatchNorms (their partial rows come from the data gradient
 * launched before): extra workgroups of the blocked kernels, one launch ahead of the first-generation ones. */
int dsgcn_pwconv_wgrad_jobs(const float* x1, const float* s1, const float* h1, const float* x2, const float* s2,
                            const float* h2, int relu, const float* z, const float* zaug, const float* gz,
                            const float* gzaug, const float* A0, const float* B0, float* dwp, float* dbp, int pstride, int n,
                            int Ci, int Co, int T, int V, int stride, int aug, const dsgcn_bn_coef_job* jobs, int njobs,
                            void* stream);
/* Up to three 1x1 convs of ONE shape in one launch each way (host arrays of ngroup <= 3 device pointers):
 * z_g = W_g . (x1_g * s1_g[ci] + h1_g[ci]) (s1_g / h1_g NULL for all or none), no bias, second stream or statistics, stride 1
 * — the three conv4's of a CTR-GCN unit (gcn.py:655-657 per subset), each too small to fill the chip alone.
 * dsgcn_pwconv_group_ok = 1 when the shape takes the grouped form (otherwise DSGCN_EUNSUPPORTED: launch them one by one).
 * The data gradient writes dx1_g and, when ipart_g is given, the input-scale rows (dsgcn_pwconv_ipart_rows each). */
int dsgcn_pwconv_group_ok(int n, int Ci, int Co, int T, int V);
int dsgcn_pwconv_fwd_group(const float* const* x1, const float* const* s1, const float* const* h1, int relu,
                           const float* const* w, float* const* z, int ngroup, int n, int Ci, int Co, int T, int V,
                           void* stream);
int dsgcn_pwconv_dgrad_group(const float* const* x1, const float* const* s1, const float* const* h1, int relu,
                             const float* const* w, const float* const* gz, float* const* dx1, float* const* ipart,
                             int ngroup, int n, int Ci, int Co, int T, int V, void* stream);
/* The weight gradients of the same group in one launch: dwp_g / dbp_g take the partial rows of conv g
 * (dsgcn_pwconv_wgrad_splits rows of pstride floats each, as dsgcn_pwconv_wgrad); DSGCN_EUNSUPPORTED when the shape is not on
 * the blocked weight-gradient kernels (launch them one by one then). */
int dsgcn_pwconv_wgrad_group(const float* const* x1, const float* const* s1, const float* const* h1, int relu,
                             const float* const* gz, float* const* dwp, float* const* dbp, int pstride, int ngroup, int n,
                             int Ci, int Co, int T, int V, void* stream);

/* ---- K-A': subset-summed aggregate (ST-GCN unit_gcn gcn.py:81-86, CTR-GCN unit_ctrgcn gcn.py:658,917-921) ----
 * y[n,c,t,w] = sum_k sum_u p[n,k*Co+c,t,u] * adj_k[u,w], adj_k at ahat + n*a_ns + k*a_ks + c*a_cs (element strides):
 * shared A (K,V,V): (0, V*V, 0); per sample and channel (n,K*Co,V,V): (K*Co*V*V, Co*V*V, V*V); per sample, shared by
 * the channels (n,K,V,V) — AAGCN's adaptive topology, gcn.py:431-437: (K*V*V, V*V, 0).
 * partial (n*Co, 2) or NULL: per-plane sum / sum of squares of y -> dsgcn_bn_finalize.
 * bwd: G = gy + A0[c] + B0[c]*y (y, A0, B0 may be NULL); dp (n,K*Co,T,V) fully written; dAhat_k of plane (n,c) is
 * written at dahat + n*d_ns + k*d_ks + c*d_cs (shared A: write per-plane pieces, reduce with dsgcn_colsum). */
/* rows of the forward's `partial` buffer (rows, Co, 2) */
int dsgcn_aggsum_partial_rows(int n, int T, int V);
/* shared adjacency only: R > 0 -> the backward writes dahat as (R, K, V, V) per-wave pieces (d_* strides ignored),
 * 0 -> per-(n,c) pieces through the d_* strides; either way dA = column sum over the pieces */
int dsgcn_aggsum_bwd_piece_rows(int n, int K, int Co, int T, int V);
/* tuning / A-B knobs used by tools/ (0: pipelined kernels on/off, 1: forward waves, 2: backward workgroups) */
int dsgcn_aggsum_fwd(const float* p, const float* ahat, long a_ns, long a_ks, long a_cs, float* y, float* partial,
                     int n, int K, int Co, int T, int V, void* stream);
int dsgcn_aggsum_bwd(const float* p, const float* ahat, long a_ns, long a_ks, long a_cs, const float* gy,
                     const float* y, const float* A0, const float* B0, float* dp, float* dahat, long d_ns, long d_ks,
                     long d_cs, int n, int K, int Co, int T, int V, void* stream);

/* ---- CTR-GCN channel-wise topology (gcn.py:634-666), the parts that are not 1x1 convs ----
 * tanhdiff  : proj (n, 2*K*R, V) rows [k*R+r] = conv1_k(xbar), [K*R+k*R+r] = conv2_k(xbar)
 *             -> d (K, n, R, V, V) = tanh(x1[u] - x2[v])  (gcn.py:653-655); bwd: dproj from dd.
 * ctr_affine: ahat (n, K*Co, V, V) = alpha[0] * s[k] (n,Co,V,V) + A (K,V,V)  (gcn.py:657); s / ds are host arrays
 *             of K <= 4 device pointers.  bwd: ds[k] = alpha*dahat; prow (4*n, K*V*V + K): per (sample, channel
 *             slice) [sum_c dahat | sum dahat*s] (sum over the rows = [dA | per-subset dalpha]).
 * plane_stats: partial (planes, 2) = per-plane [sum, sum of squares] of x (planes, L). */
int dsgcn_tanhdiff_fwd(const float* proj, float* d, int n, int K, int R, int V, void* stream);
int dsgcn_tanhdiff_bwd(const float* d, const float* dd, float* dproj, int n, int K, int R, int V, void* stream);
/* the same with the gradient as K separate tensors (n, R, V, V) (host array of K device pointers, NULL = zero) */
int dsgcn_tanhdiff_bwd_k(const float* d, const float* const* dd, float* dproj, int n, int K, int R, int V, void* stream);
/* Classic CTR-GCN refinement as ONE conv per subset (round 5).  Ahat_k = alpha * conv4_k(d_k) + A[k] (ctrgcn's unit:
 * pyskl/models/gcns/utils/gcn.py unit_ctrgcn, `x1 = conv4(tanh(x1 - x2)) * alpha + A[i]`) equals a 1x1 conv over R + 2
 * channels [d_k | A[k] | 1] with weights [W_k | 1 | b_k] and the per-channel input scale [alpha .. alpha, 1, alpha]: the
 * separate affine pass over the (n, K*Co, V, V) tensor, its backward and their intermediate tensors are gone, and the
 * gradients of alpha (from the input-scale sums), A (the A channel's data gradient), W and b (columns of the weight
 * gradient) fall out of the conv's own backward.
 *   dsgcn_tanhdiff_aug_fwd: d (K, n, R + 2, V, V) = [tanh(x1 - x2) | A[k] | 1].
 *   dsgcn_tanhdiff_aug_bwd: dd = K gradients (n, R + 2, V, V) (host array, NULL = zero) -> dproj as dsgcn_tanhdiff_bwd and
 *     dAp (n, K, V, V) = the A-channel rows (their column sum over n is dA).
 *   dsgcn_ctr_wprep: w / b = K device pointers (Co, R) / (Co) (b[k] may be NULL) -> wout (K, Co, R + 2), sh (K, 2, R + 2) =
 *     per subset [input scale; input shift = 0].
 *   dsgcn_ctr_wfin: dwp = K gradients of wout[k] (NULL = zero), ds = K gradients of the input scales (NULL = zero; their
 *     R + 2 elements ds_stride floats apart) ->
 *     out[k] (Co*R + Co) = [dW_k | db_k], dalpha (1).
 *   dsgcn_ctr_wprep_multi / dsgcn_ctr_wfin_multi: the same for njobs units in one launch per 16 records (dsgcn_jobs.h) —
 *     both depend on parameters / finished parameter-gradient sums only, so a training step runs them once for the
 *     whole model. */
int dsgcn_tanhdiff_aug_fwd(const float* proj, const float* A, float* d, int n, int K, int R, int V, void* stream);
int dsgcn_tanhdiff_aug_bwd(const float* d, const float* const* dd, float* dproj, float* dAp, int n, int K, int R, int V,
                           void* stream);
int dsgcn_ctr_wprep(const float* const* w, const float* const* b, const float* alpha, float* wout, float* sh, int K,
                    int Co, int R, void* stream);
int dsgcn_ctr_wfin(const float* const* dwp, const float* const* ds, int ds_stride, float* const* out, float* dalpha, int K,
                   int Co, int R, void* stream);
int dsgcn_ctr_wprep_multi(const dsgcn_ctr_prep_job* jobs, int njobs, void* stream);
int dsgcn_ctr_wfin_multi(const dsgcn_ctr_fin_job* jobs, int njobs, void* stream);
int dsgcn_ctr_affine_fwd(const float* const* s, const float* alpha, int alpha_stride, const float* A, const float* beta,
                         const float* G, float* ahat, int n, int K, int Co, int V, void* stream);
int dsgcn_ctr_affine_bwd(const float* const* s, const float* alpha, int alpha_stride, const float* dahat,
                         float* const* ds, float* prow, int n, int K, int Co, int V, void* stream);
/* CTRHGC (gcn.py:668-771) extras of the same step: alpha_stride 1 = one alpha per subset (unit_ctrhgcn.alpha, gcn.py:862),
 * 0 = the classic shared scalar; G (n,K,V,V) with beta (K): the "ada" Gram term x1^T x2 added as beta_k * G (gcn.py:755-760),
 * NULL = none.  edge_select: the edge-typed attention conv (gcn.py:737) yields E*R channels; joint pair (u,v) keeps channel
 * eps(u,v)*R + r (gcn.py:738-745, the index_select over a 625-iteration host loop).  in (n,E*R,V,V) -> out (n,R,V,V);
 * backward writes din completely (zeros at the classes not selected). */
int dsgcn_edge_select_fwd(const float* in, const int* edge_type, float* out, int n, int R, int E, int V, void* stream);
int dsgcn_edge_select_bwd(const float* dout, const int* edge_type, float* din, int n, int R, int E, int V, void* stream);
int dsgcn_plane_stats(const float* x, float* partial, long planes, int L, void* stream);

/* out = a + b (+ c), n elements (c may be NULL; 16-B aligned buffers): the gradients of one block input from its
 * consumers (pre conv, residual operand of the temporal unit, block residual: dgstgcn.py:63-65) summed in one pass —
 * replaces autograd's pairwise accumulation. */
int dsgcn_add3(const float* a, const float* b, const float* c, float* out, long n, void* stream);

/* Replaces the per-parameter gradient copies of torch DDP's bucketing (the reference wraps the model in
 * MMDistributedDataParallel, pyskl/apis/train.py:94-102).  Pack `count` gradient tensors into the flat data-parallel
 * buffer in one launch: src_table / dst_offsets / numels
 * are DEVICE arrays (pointers to the tensors, element offsets into dst, element counts). */
int dsgcn_pack(const float* const* src_table, const long* dst_offsets, const int* numels, int count, float* dst,
               void* stream);
/* dsgcn_pack with 64-bit lengths and src_table[i] == NULL meaning "fill that range with zeros": given every parameter of
 * the flat buffer it writes the whole buffer in one launch. */
int dsgcn_pack_fill(const float* const* src_table, const long* dst_offsets, const long* numels, int count, float* dst,
                    void* stream);

/* Depthwise causal temporal taps of unitmlp (tcn.py:525-614: left zero pad + grouped Conv1d, groups = channels):
 *   y[n,c,t',v] = b[c] + sum_{j<KM} w[c,j] * h[n,c, t'*stride - (KM-1-j)*dil[c], v]   (frames < 0 read as zero), KM <= 5;
 * dil (C) int32, 0 = channel outside the mlp windows (y = 0, dh = 0).  h (n,C,T,V); y (n,C,ceil(T/stride),V); w (C,KM).
 * Backward: dh (n,C,T,V), part (n*C, 6) = per-plane [dw_0..dw_4, db] (dsgcn_colsum over n finishes them). */
int dsgcn_dwcausal_fwd(const float* h, const float* w, const float* b, const int* dil, float* y, int n, int C, int T,
                       int V, int stride, int KM, void* stream);
int dsgcn_dwcausal_bwd(const float* h, const float* w, const int* dil, const float* dy, float* dh, float* part, int n,
                       int C, int T, int V, int stride, int KM, void* stream);

/* K-D, fused (csrc/tms.hip): the whole multi-scale temporal stage between the two 1x1 convs of dgmstcn / mstcn / MSTCN
 * (reference: pyskl/models/gcns/utils/tcn.py:383-396,409-420 and msg3d_utils.py:84-117) as ONE launch per direction on
 * the (n, C, T, V) layout, any V:
 *   h = act(z*scale + shift) [ReLU on channels < n_act] with the global-joint column act(zaug*scale + shift) appended
 *       (zaug (n,C,T) or NULL: no global joint);  scale/shift NULL: h = z;
 *   o = per channel window i (type 0: (KT,1) conv of dilation dil, weights w[i] (bc,bc,KT,1), bias b[i]; 1: (3,1)
 *       max-pool, pad 1; 2: strided copy), windows [c0, c0+bc) of input and output coincide, stride 1 or 2 over frames;
 *   f = o[..., :V] + o[..., V] * coeff  (zaug given; else f = o);  oaug (n,C,Tout) = o[..., V] kept for the backward;
 *   stats (rows(0), C, 2): per-workgroup partial sums of f and f^2 (dsgcn_bn_finalize finishes them), or NULL.
 * Replaces dsgcn_branch_act_* + dsgcn_tapconv_* + dsgcn_tms_combine_* and their two (V+1)-column intermediates.
 * Backward: gf = gradient of f; A0/B0 (C) the statistics terms of the BatchNorm that consumes f (dfe = gf + A0 + B0*f),
 *   or NULL; dz (n,C,T,V), dzaug (n,C,T);  paff (rows(1), C, 2) partials of d scale / d shift;  pcoeff (rows(1)*nbr, V)
 *   partials of d coeff;  dsgcn_tms_wgrad: window i writes row r < rows(2) of its weight / bias partials at
 *   dwp[i] + r*pstride / dbp[i] + r*pstride (dsgcn_colsum finishes them).
 * dsgcn_tms_rows(which, ...) -> the partial-row counts above, or 0 when the shape is not eligible (KT 3 or 5, stride 1
 * or 2, T*V and Tout*V multiples of 4, windows <= 64 channels, dil*(KT/2) <= 4): the staged kernels then apply. */
int dsgcn_tms_rows(int which, int n, int C, int T, int V, int stride, int KT, int nbr, const int* type, const int* bc,
                   const int* dil, int aug);
int dsgcn_tms_fwd(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                  const float* coeff, float* f, float* oaug, float* stats, int n, int C, int T, int V, int stride, int KT,
                  int nbr, const int* type, const int* c0, const int* bc, const int* dil, const float* const* w,
                  const float* const* b, void* stream);
int dsgcn_tms_dgrad(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                    const float* coeff, const float* gf, const float* f, const float* A0, const float* B0,
                    const float* oaug, float* dz, float* dzaug, float* paff, float* pcoeff, int n, int C, int T, int V,
                    int stride, int KT, int nbr, const int* type, const int* c0, const int* bc, const int* dil,
                    const float* const* w, void* stream);
int dsgcn_tms_wgrad(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                    const float* coeff, const float* gf, const float* f, const float* A0, const float* B0, int n, int C,
                    int T, int V, int stride, int KT, int nbr, const int* type, const int* c0, const int* bc,
                    const int* dil, float* const* dwp, float* const* dbp, int pstride, void* stream);

/* K-D, split layout (csrc/tmsplit.hip): the same stage as dsgcn_tms_* for dgmstcn units with the global joint
 * (reference: pyskl/models/gcns/utils/tcn.py:379-428), kernel 3, stride 1, V odd (5..25), T % 4 == 0, T*V <= 2048, windows
 * <= 64 channels tiling [0, C) in order, dilation <= 4.  The (V+1)-column intermediates of tcn.py:409-420 are never formed:
 * the V columns run through the matrix-core window kernels on the (n,C,T,V) layout (BatchNorm affine + ReLU + the
 * global-joint term haug*coeff while loading z — the windows are linear, so o + oaug*coeff = conv(h + haug*coeff) +
 * bias*(1 + coeff) —, the batch statistics of f, resp. dz = relu'(.)*dh*scale and the BatchNorm backward sums, in the
 * epilogue), the global-joint column as extra blocks of the same launches on (n,C,T) tensors.
 *   dsgcn_tms_split_rows(which): -1 -> 1 / 0 eligible;  0 -> rows of stats;  1 -> rows of part;  2 -> K-splits of the
 *     weight gradient;  0 whenever the shape is not eligible (the staged kernels then apply).
 *   fwd  : f (n,C,T,V);  oaug (n,C,T) = the windows' output on the global-joint column (kept for prep);  stats
 *          (rows(0), C, 2) partial sums of f, f^2 (dsgcn_bn_finalize, count n*T*V) or NULL.  One launch.
 *   prep : ge = gf + A0 + B0*f (gf NULL = 0; A0/B0 NULL = no BatchNorm terms), doaug (n,C,T) = sum_v ge*coeff,
 *          pcoef (n*C, V) = sum_t ge[t,v]*oaug[t] (dsgcn_colsum finishes d coeff).
 *   dgrad: dz (n,C,T,V), dzaug (n,C,T), part (rows(1), C, 2) = [sum dpre*x, sum dpre] over z AND zaug (the sums
 *          dsgcn_bn_coef_rows takes).  One launch.
 *   wgrad: window i writes split s < rows(2) of its weight / bias partials at dwp[i] + s*pstride / dbp[i] + s*pstride. */
int dsgcn_tms_split_rows(int which, int n, int C, int T, int V, int stride, int KT, int nbr, const int* type,
                         const int* c0, const int* bc, const int* dil);
int dsgcn_tms_split_fwd(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                        const float* coeff, float* f, float* oaug, float* stats, int n, int C, int T, int V, int stride,
                        int nbr, const int* type, const int* c0, const int* bc, const int* dil, const float* const* w,
                        const float* const* b, void* stream);
int dsgcn_tms_split_prep(const float* gf, const float* f, const float* oaug, const float* coeff, const float* A0,
                         const float* B0, float* ge, float* doaug, float* pcoef, int n, int C, int T, int V,
                         void* stream);
int dsgcn_tms_split_dgrad(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                          const float* ge, const float* doaug, float* dz, float* dzaug, float* part, int n, int C, int T,
                          int V, int stride, int nbr, const int* type, const int* c0, const int* bc, const int* dil,
                          const float* const* w, void* stream);
int dsgcn_tms_split_wgrad(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                          const float* ge, const float* doaug, int n, int C, int T, int V, int stride, int nbr,
                          const int* type, const int* c0, const int* bc, const int* dil, float* const* dwp, float* const* dbp,
                          int splits, int pstride, void* stream);
/* The weight gradient carrying BatchNorm coefficient jobs (njobs <= DSGCN_BN_JOBS_MAX, host records) as extra workgroups:
 * the branch BatchNorm's rows come from dsgcn_tms_split_dgrad, nothing waits for the weight gradient. */
int dsgcn_tms_split_wgrad_jobs(const float* z, const float* zaug, const float* scale, const float* shift, int n_act,
                               const float* ge, const float* doaug, int n, int C, int T, int V, int stride, int nbr,
                               const int* type, const int* c0, const int* bc, const int* dil, float* const* dwp,
                               float* const* dbp, int splits, int pstride, const dsgcn_bn_coef_job* jobs, int njobs,
                               void* stream);

/* AAGCN attention gates (csrc/aagcn.hip; reference gcn.py:447-459: y <- y * sigmoid(.) + y, three times).
 * out = y * (1 + g), g broadcast by mode: 0 g (n, V) per joint, 1 g (n, T) per frame, 2 g (n, C) per channel;
 * rmode 1: rout (n, C, T) = mean over joints of out, rmode 2: rout (n, C) = mean over frames and joints of out (the input
 * of the NEXT gate), 0: none.  Backward: G = gout + d(rout) spread back; dy = G * (1 + g); dgp = sum of G * y inside each
 * (n, c) plane over the axes g is broadcast along — (n, C, V) / (n, C, T) / (n, C); the sum over C is the caller's. */
int dsgcn_gate_fwd(const float* y, const float* g, int mode, float* out, float* rout, int rmode, int n, int C, int T,
                   int V, void* stream);
int dsgcn_gate_bwd(const float* y, const float* g, int mode, const float* gout, const float* drout, int rmode, float* dy,
                   float* dgp, int n, int C, int T, int V, void* stream);

/* Skeleton input pipeline, per-element half (csrc/skeleton.hip).  Replaces the numpy transforms of the reference's
 * loader workers — PreNormalize3D (pose_related.py:250-336), RandomRot (144-178), JointToBone / ToMotion / GenSkeFeat
 * (340-442), PoseDecode (19-54), FormatGCNInput (468-518) — for a whole batch in one launch; the per-clip decisions
 * (kept frames, person swap, body centre, linear map, UniformSample's frame indices: sampling.py:10-192) come from the host.
 *   raw: packed clips, clip n = (M[n], T[n], V, C) fp32 at raw + offset[n] (offset in floats, int64);
 *   flags[n]: bit 0 = persons swapped, bit 1 = subtract center[n] from non-zero joints and zero the others (else plain
 *   subtraction);  center (N,3);  matrix (N,9) row-major, applied after centring;  f0 / f1 (N, clips*clip_len): original
 *   frame of each output frame and of its successor in the kept sequence (-1 = none: motion features are 0);
 *   parent (V): bone parent of each joint;  fmask: 2 bits per feature (0 j, 1 b, 2 jm, 3 bm), nfeat features;
 *   scored: third channel is a confidence (2-D layouts: bone / motion entries average it);  loop: with fewer persons
 *   than slots every slot after the first repeats person 0 (FormatGCNInput mode='loop', pose_related.py:492-497);
 *   flags[n] bit 2: the clip came from an fp16 pickle and is not rotated — its bone / motion differences are rounded to
 *   fp16 like the reference's in-place numpy arithmetic on the pickle's dtype (GenSkeFeat runs before PoseDecode's cast
 *   to fp32, pose_related.py:340-412).   flags[n] bit 3: PoseCompact (augmentations.py:21-116) — center[n] is the integer
 *   origin of the clip's compact box and is subtracted from every NON-ZERO coordinate, x and y tested separately
 *   (`kp_x[kp_x != 0] -= min_x`); f1 is then the next SAMPLED frame (the Kinetics config builds its features after
 *   sampling).   out (N, clips, Mout, clip_len, V, C*nfeat). */
int dsgcn_skeleton_prep(const float* raw, const long* offset, const int* M, const int* T, const int* flags,
                        const int* f0, const int* f1, const float* center, const float* matrix, const int* parent,
                        float* out, int N, int clips, int Mout, int clip_len, int V, int C, int nfeat, int fmask,
                        int scored, int loop, void* stream);

/* Classification head of the training step (csrc/head.hip).  Replaces, for the 'GCN' pooling mode, the person mean +
 * nn.Linear of SimpleHead.forward (pyskl/models/heads/simple_head.py:88-98 after the plane mean), BaseHead.loss
 * (heads/base.py:50-84: top-1 / top-5 accuracy, computed there on the host with numpy, + the loss) and CrossEntropyLoss
 * (losses/cross_entropy_loss.py:75-82, scaled by loss_weight: losses/base.py:38-44) — ~25 framework launches per step.
 *   feat (N*M, C): per-person plane means;  w (K, C), b (K) or NULL;  label (N) int64.
 *   pooled (N, C) = mean over the M persons;  score (N, K) = pooled w^T + b;  prob (N, K) = softmax(score);
 *   loss (1) = loss_weight * mean_n -log prob[n, label[n]] (NaN if a label is outside [0, K));
 *   acc (2) fp64 = share of clips whose label is among the 1 / 5 best scores, ties as a stable ascending argsort orders
 *   them (core/evaluation.py:63-88);  clip (N, 3): scratch (per-clip loss and hits).   Two launches.
 * dsgcn_head_loss_bwd: gloss (1) = gradient of the loss scalar; dfeat (N*M, C), dw (K, C), db (K).  One launch. */
int dsgcn_head_loss_fwd(const float* feat, const float* w, const float* b, const long long* label, int N, int M, int C,
                        int K, float loss_weight, float* pooled, float* score, float* prob, float* clip, float* loss,
                        double* acc, void* stream);
int dsgcn_head_loss_bwd(const float* prob, const float* pooled, const float* w, const long long* label, const float* gloss,
                        int N, int M, int C, int K, float loss_weight, float* dfeat, float* dw, float* db, void* stream);

/* Training-mode buffer update of njobs BatchNorm layers in one launch (what F.batch_norm(training=True) does to its
 * buffers, momentum form): running_mean = (1 - m) running_mean + m mean; running_var = (1 - m) running_var + m var *
 * unbias (unbias = count / (count - 1)); num_batches_tracked += 1 (entries may be NULL).  The arrays are HOST arrays of
 * njobs device pointers / values. */
int dsgcn_bn_running_multi(float* const* running_mean, float* const* running_var, const float* const* mean,
                           const float* const* var, long long* const* num_batches_tracked, const int* C,
                           const float* unbias, const float* momentum, int njobs, void* stream);

/* The input BatchNorm of the skeleton backbones (pyskl/models/gcns/dgstgcn.py:158-164; stgcn.py, ctrgcn.py, aagcn.py
 * alike): x (N, M, T, V, C) is normalised per (v, c) ['VC': mvc = 0, statistics over the N*M person-samples and T] or per
 * (m, v, c) ['MVC': mvc = 1, over N and T] channel — nn.BatchNorm1d on the permuted clip — and written as (N*M, C, T, V).
 * training != 0: batch statistics (biased variance; saved in save_mean / save_invstd (channels)), running_mean /
 * running_var (unbiased) / num_batches_tracked updated with `momentum` when the pointers are given; training == 0: the
 * running statistics.  gamma / beta may be NULL.  scratch: 4 * N*M * V*C floats (two fp64 tables; training).  V*C <= 256.
 * dsgcn_data_bn_bwd: dy (N*M, C, T, V) -> pg / pb (N*M, V*C), per-sample partial sums of dgamma / dbeta (summed over the
 * samples of a channel by the caller's column sum); the clip gets no gradient. */
int dsgcn_data_bn_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                      long long* num_batches_tracked, float* y, float* save_mean, float* save_invstd, float* scratch,
                      int N, int M, int T, int V, int C, int mvc, int training, float eps, float momentum,
                      void* stream);
int dsgcn_data_bn_bwd(const float* x, const float* dy, const float* save_mean, const float* save_invstd, float* pg,
                      float* pb, int N, int M, int T, int V, int C, int mvc, void* stream);

/* One SGD step over flat fp32 buffers of n elements (16-byte aligned), torch.optim.SGD's update with dampening 0 — the
 * reference's optimizer (configs/_init_/lr_schedual.py:11-15: momentum 0.9, weight decay 5e-4, nesterov):
 *   g' = g + weight_decay p;  buf = momentum buf + g';  p -= lr[0] * (nesterov ? g' + momentum buf : buf);
 * lr: ONE device float (a captured launch follows the schedule on replay); buf may be NULL when momentum == 0. */
int dsgcn_sgd_step(float* p, const float* g, float* buf, const float* lr, float momentum, float weight_decay,
                   int nesterov, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSGCN_H_ */
