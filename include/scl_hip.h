/*
 * scl_hip.h — C-ABI of the MI355X (gfx950) soft-contrastive hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI of its
 * own: its boundary is two Python modules of free functions,
 *   model/nets.py   (vgg16Netvlad :7-69, vgg16 :72-131)
 *   model/losses.py (wms_loss :5-60, ms_loss :76-122, logratio_loss :125-135,
 *                    evil_* :63-73,197-222, _pairwise_squared_distances :656-661)
 * plus the third-party netvlad_tf.layers.netVLAD (call site model/nets.py:67) and
 * pointnetvlad_cls.{triplet,quadruplet,lazy_*}_loss (call sites
 * train/train.py:700-712), and the KDTree query of evaluation/top-n.py:103-108.
 * Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (HBM); the
 *     library allocates nothing and the compute entry points keep no state between calls,
 *     so they are re-entrant (the reference drives one session from three threads,
 *     train/train.py:967-975).  Two PROCESS-WIDE settings exist in the product library
 *     (libscl_hip.so): the timing sink of scl_prof_begin / scl_prof_end (changes no result) and
 *     the number of CUs the persistent grids leave free, scl_set_reserve_cus (deterministic for a
 *     fixed value; weight gradients differ in rounding between values).  The ablation selector
 *     scl_debug_set_variant is compiled into the DIAGNOSTIC build only (libscl_hip_diag.so,
 *     -DSCL_DIAG; see "Diagnostics" at the end): the product library contains no variant code
 *     and rejects every value but 0;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     no entry point synchronises;
 *   - every function returns 0 on success, a negative SCL_E_* code for a rejected
 *     call (nothing enqueued) or a positive hipError_t from the launch;
 *   - tensors are row-major, contiguous unless a stride argument says otherwise;
 *   - scratch comes from the caller: ask scl_*_workspace_bytes first.  Workspaces
 *     must be 256-byte aligned.
 */
#ifndef SCL_HIP_H
#define SCL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCL_ABI_VERSION 12

/* error codes (negative = rejected before any launch) */
#define SCL_OK 0
#define SCL_E_SHAPE -1       /* unsupported or inconsistent shape            */
#define SCL_E_KIND -2        /* unknown loss / mask / dtype selector         */
#define SCL_E_NULL -3        /* required pointer is NULL                     */
#define SCL_E_WORKSPACE -4   /* workspace too small or misaligned            */

/* element types of the conv5_3 feature map handed to NetVLAD */
#define SCL_DT_F32 0
#define SCL_DT_BF16 1

/* NetVLAD is specialised to the reference's only configuration:
 * D = 512 VGG16 conv5_3 channels, K = 64 clusters (model/nets.py:67). */
#define SCL_VLAD_D 512
#define SCL_VLAD_K 64
/* rows of 64 floats per image in save_vlad: 512 of the pre-norm VLAD, row 512 = sum_n a[n,:],
 * row 513 = sync words of the backward pass (zeroed by the forward, zero again when the
 * backward returns; see scl_netvlad_bwd) */
#define SCL_VLAD_SAVE_ROWS 514

/* pair-mask kinds of the Gram-matrix loss family */
#define SCL_MASK_WMS_EXP 0   /* wms_loss wfunction='exp'  (model/losses.py:17-19) */
#define SCL_MASK_WMS_LIN 1   /* wms_loss wfunction='lin'  (:11-13)                */
#define SCL_MASK_WMS_TANH 2  /* wms_loss wfunction='tanh' (:14-16)                */
#define SCL_MASK_LABELS 3    /* ms_loss / ms_det label adjacency (:88-92)         */

#define SCL_SUM_MS 0         /* sumfunction='ms'    (model/losses.py:48-58)       */
#define SCL_SUM_PLAIN 1      /* sumfunction='plain' (:39-46)                      */

/* tuple-loss kinds: reduction over positives (best=min / worst=max) and over
 * negatives (sum / lazy=max), with or without the second ("other negative") term */
#define SCL_TUPLE_TRIPLET 0          /* pointnetvlad_cls.triplet_loss             */
#define SCL_TUPLE_LAZY_TRIPLET 1     /* pointnetvlad_cls.lazy_triplet_loss        */
#define SCL_TUPLE_EVIL_TRIPLET 2     /* model/losses.py:63-73                     */
#define SCL_TUPLE_QUADRUPLET 3       /* pointnetvlad_cls.quadruplet_loss          */
#define SCL_TUPLE_LAZY_QUADRUPLET 4  /* pointnetvlad_cls.lazy_quadruplet_loss     */
#define SCL_TUPLE_EVIL_QUADRUPLET 5  /* model/losses.py:197-214                   */

int scl_abi_version(void);
const char* scl_error_string(int code);

/* ------------------------------------------------------------------------- *
 * NetVLAD head — replaces tf.nn.l2_normalize(x, axis=-1) + layers.netVLAD(x, 64)
 * (model/nets.py:66-67; upstream netvlad_tf/layers.py).
 *
 *   x        [B, N, 512]  conv5_3 map, channels last, N = H'*W' (f32 or bf16)
 *   assign_w [512, 64]    = assignment/kernel[0,0]
 *   centers  [512, 64]    = cluster_centers[0,0,0]
 *   out      [B, 32768]   index d*64 + k, unit L2 norm
 * For training the forward also leaves, in caller buffers, what the backward
 * re-uses (pass NULL for all four when only inferring):
 *   save_assign [B,N,64] soft-assignment a;  save_logit [B,N,64] logits (float32 feature maps
 *   only: the bf16 kernels neither write nor read it since round 6 — their backward pass takes
 *   log a, which differs from the logit by a per-location constant that cancels — pass NULL);
 *   save_rnorm  [B,N] per-location 1/||x||;  save_vlad  [B,SCL_VLAD_SAVE_ROWS,64].
 * save_assign and save_rnorm double as forward scratch: when NULL they are carved
 * from the workspace.
 *
 * w_planes (the _p entries; NULL = build them inside the call, one more launch): the bf16
 * plane images of assign_w that the fused kernels keep in registers, scl_netvlad_planes_bytes()
 * bytes, 256-byte aligned, written by scl_netvlad_planes() or — in the same launch as the
 * packed convolution weights, once per optimizer step — by a SclPackJob with flags =
 * SCL_PACK_VLAD_W (w = assign_w, cin = 512, kout = 64, strides ignored).  They must be rebuilt
 * whenever assign_w changes (train/train.py:877-879: once per step).
 *
 * Launches for a bf16 map with w_planes given: forward 2 (fused assignment + aggregation; finish),
 * backward 3 (prologue; fused softmax-backward + dW slabs; grad_x with the parameter-gradient sums in
 * its tail).  The 8 workgroups of an image inside the finish / prologue kernels exchange one or
 * two scalars through 8-byte tagged words with a bounded wait and a self-computing fallback: no
 * result depends on dispatch order or co-residency.
 * ------------------------------------------------------------------------- */
size_t scl_netvlad_planes_bytes(void);
int scl_netvlad_planes(const float* assign_w, void* planes, void* stream);
size_t scl_netvlad_fwd_workspace_bytes(int B, int N);
int scl_netvlad_fwd(const void* x, int x_dtype, const float* assign_w, const float* centers,
                    int B, int N, int pre_l2, float* out, float* save_assign,
                    float* save_logit, float* save_rnorm, float* save_vlad, void* workspace,
                    size_t workspace_bytes, void* stream);
int scl_netvlad_fwd_p(const void* x, int x_dtype, const float* assign_w, const float* centers,
                      const void* w_planes, int B, int N, int pre_l2, float* out,
                      float* save_assign, float* save_logit, float* save_rnorm, float* save_vlad,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the above (TF autodiff of the same graph, train/train.py:877-878).
 *   grad_out [B,32768];  grad_x [B,N,512] in x's dtype;  grad_w, grad_c [512,64]
 *   (sums over the batch, overwritten).
 * save_vlad is not const: row 513 of every image carries the exchange words of the prologue
 * kernel; they are zero on entry (the forward left them so) and zero again on return, so the
 * same saved tensors can be back-propagated through more than once — but not by two calls
 * at the same time. */
size_t scl_netvlad_bwd_workspace_bytes(int B, int N);
int scl_netvlad_bwd(const void* x, int x_dtype, const float* assign_w, const float* centers,
                    const float* grad_out, const float* save_assign, const float* save_logit,
                    const float* save_rnorm, float* save_vlad, int B, int N, int pre_l2,
                    void* grad_x, float* grad_w, float* grad_c, void* workspace,
                    size_t workspace_bytes, void* stream);
int scl_netvlad_bwd_p(const void* x, int x_dtype, const float* assign_w, const float* centers,
                      const void* w_planes, const float* grad_out, const float* save_assign,
                      const float* save_logit, const float* save_rnorm, float* save_vlad, int B,
                      int N, int pre_l2, void* grad_x, float* grad_w, float* grad_c,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Gram-matrix losses — replace wms_loss (model/losses.py:5-60) and
 * ms_loss / ms_det (:76-122, :139-185).
 *
 *   emb        [B, E] rows at stride ld_emb floats (not required unit norm: the
 *              op L2-normalises rows like losses.py:7,82)
 *   distances  [B,B] f32 geographic distances (wms kinds) or NULL
 *   dist_rank3 1 when the reference caller fed the rank-3 [1,B,B] placeholder
 *              (train/train.py:684-686): axis=1 then reduces over rows
 *   labels     [B] int64 class ids (SCL_MASK_LABELS) or NULL
 *   loss_out   device scalar
 *   coef       [B,B] f32 or NULL: matrix M with d loss / d emb = M @ emb,
 *              consumed by scl_gram_loss_bwd
 * ------------------------------------------------------------------------- */
size_t scl_gram_loss_workspace_bytes(int B, int E);
int scl_gram_loss_fwd(const float* emb, int64_t ld_emb, int B, int E, int mask_kind,
                      const float* distances, int dist_rank3, float d_alpha, float d_beta,
                      const int64_t* labels, float alpha, float beta, float lamb, float eps,
                      int ms_mining, int sum_kind, float* loss_out, float* coef,
                      void* workspace, size_t workspace_bytes, void* stream);
/* The same with a SYNC BLOCK (B <= 32: the forward is then ONE launch — the Gram kernel, whose last
 * workgroup to arrive runs the finish — instead of two).  sync_words: at least 12 bytes of device
 * memory that are ZERO when the call is enqueued and that no other call in flight uses; the
 * library leaves them zero, so one zero-initialised block per (device, stream) serves every
 * call.  A block that was not zero on entry cannot be repaired: the library then sets word 2 for
 * good and every call on the block returns a NaN loss until the caller zeroes it.  NULL, or B > 32: exactly scl_gram_loss_fwd.  The results are bit-identical either way
 * (same sums in the same order).  (Diagnostic build only: with 8 bytes of 8-byte-aligned sync block
 * scl_debug_set_variant(41) runs 32 < B <= 208 as ONE persistent kernel with grid barriers —
 * bit-identical, deadlock-free by a bounded spin + repair path, and slower than the four launches:
 * profiles/r06/loss_one_launch_persistent.txt.) */
int scl_gram_loss_fwd_s(const float* emb, int64_t ld_emb, int B, int E, int mask_kind,
                        const float* distances, int dist_rank3, float d_alpha, float d_beta,
                        const int64_t* labels, float alpha, float beta, float lamb, float eps,
                        int ms_mining, int sum_kind, float* loss_out, float* coef, void* workspace,
                        size_t workspace_bytes, void* sync_words, void* stream);

/* grad_emb[r, :] = (*grad_loss) * sum_j coef[row_begin + r, j] * emb[j, :]
 * for r in [0, row_count): a data-parallel rank asks only for its own rows.
 * grad_loss is a device scalar (NULL means 1). */
int scl_gram_loss_bwd(const float* emb, int64_t ld_emb, int B, int E, const float* coef,
                      const float* grad_loss, int row_begin, int row_count, float* grad_emb,
                      int64_t ld_grad, void* stream);
/* (ABI 9) For 64 < B <= 256 with B % 4 == 0, at least 96 rows and E % 128 == 0 (16-byte aligned
 * operands) scl_gram_loss_bwd runs both operands as three bf16 planes on the bf16 matrix cores
 * (six products: float32-equivalent, like the Gram of the forward) — the many-row case of a
 * single-process run at B = 192.  scl_gram_loss_bwd_w is the same call with a workspace argument
 * that an earlier form of that path needed; the workspace may be NULL. */
size_t scl_gram_loss_bwd_workspace_bytes(int B, int row_count);
int scl_gram_loss_bwd_w(const float* emb, int64_t ld_emb, int B, int E, const float* coef,
                        const float* grad_loss, int row_begin, int row_count, float* grad_emb,
                        int64_t ld_grad, void* workspace, size_t workspace_bytes, void* stream);

/* _pairwise_squared_distances (model/losses.py:656-661): [T,S,E] -> [T,S,S]. */
size_t scl_pairwise_sqdist_workspace_bytes(int T, int S, int E);
int scl_pairwise_sqdist(const float* feats, int T, int S, int E, float* out, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Tuple losses — replace pointnetvlad_cls.{triplet,lazy_triplet,quadruplet,
 * lazy_quadruplet}_loss (train/train.py:700-712) and the in-tree evil_* twins.
 *
 *   q [T,1,E], pos [T,P,E], neg [T,N,E], other [T,1,E] (quadruplet kinds) are
 *   views into the trainer's [T,S,E] output (train/train.py:654): each has a
 *   tuple stride (floats between tuples) and rows E floats apart.
 *   sqd   [T, P + 2N] out: squared distances anchor-pos | anchor-neg | other-neg
 *   coef  [T, P + 2N] out: d loss / d sqd (consumed by the backward)
 * ------------------------------------------------------------------------- */
int scl_tuple_loss_fwd(int kind, const float* q, int64_t q_tstride, const float* pos,
                       int64_t pos_tstride, const float* neg, int64_t neg_tstride,
                       const float* other, int64_t other_tstride, int T, int P, int N, int E,
                       float m1, float m2, float* loss_out, float* sqd, float* coef,
                       void* stream);
/* distance_triplet_loss / distance_quadruplet_loss (model/losses.py:239-307; trainer losses
 * [huber_]distance_[lazy_]{triplet,quadruplet}, train/train.py:719-763):
 *   kind  SCL_TUPLE_TRIPLET or SCL_TUPLE_LAZY_TRIPLET  (triplet_loss_name)
 *   quad  0: loss = triplet(m1) + lam * dist_term
 *         1: ... + mean_t max_n max(m2 + min_p term_p - |neg_n - other|^2 / f_max, 0)
 *   huber 1: tf.losses.huber_loss (delta 1) on the scaled distances, 0: squared difference
 *   sq_d_dists [T,P]: squared geographic anchor-positive distances (ops['distances'])
 * sqd / coef as in scl_tuple_loss_fwd; the backward is scl_tuple_loss_bwd. */
int scl_distance_tuple_loss_fwd(int kind, int quad, int huber, const float* q, int64_t q_tstride,
                                const float* pos, int64_t pos_tstride, const float* neg,
                                int64_t neg_tstride, const float* other, int64_t other_tstride,
                                int T, int P, int N, int E, float m1, float m2, float lam,
                                const float* sq_d_dists, float d_max_squared,
                                float f_max_squared, float* loss_out, float* sqd, float* coef,
                                void* stream);
/* grads are written with the same strides as their inputs; grad_other may be NULL. */
int scl_tuple_loss_bwd(const float* q, int64_t q_tstride, const float* pos, int64_t pos_tstride,
                       const float* neg, int64_t neg_tstride, const float* other,
                       int64_t other_tstride, int T, int P, int N, int E, const float* coef,
                       const float* grad_loss, float* grad_q, float* grad_pos, float* grad_neg,
                       float* grad_other, void* stream);

/* logratio_loss (model/losses.py:125-135) for the only shape its literal
 * broadcasting admits: T == 1 and P == N.  sq_pos_d / sq_neg_d are the [P] / [N]
 * squared geographic distances; sqd / coef are [P + N] like above (no third block). */
int scl_logratio_fwd(const float* a, const float* pos, const float* neg, int P, int N, int E,
                     const float* sq_pos_d, const float* sq_neg_d, float* loss_out, float* sqd,
                     float* coef, void* stream);

/* ------------------------------------------------------------------------- *
 * Retrieval — replaces KDTree(ref).query(query, k=n, sort_results=True)
 * (evaluation/top-n.py:103-108; train/train.py:1181-1182): exact Euclidean top-n,
 * ascending.  ref [R,d], query [Q,d] f32, d a multiple of 8, n <= 25.
 *   idx_out  [Q,n] int64 (ref row + idx_offset, for sharded reference sets)
 *   dist_out [Q,n] float64 Euclidean distances
 * ------------------------------------------------------------------------- */
size_t scl_topn_l2_workspace_bytes(int R, int Q, int d, int n);
int scl_topn_l2(const float* ref, int R, const float* query, int Q, int d, int n,
                int64_t idx_offset, int64_t* idx_out, double* dist_out, void* workspace,
                size_t workspace_bytes, void* stream);

/* Same call with a scoring-mode flag.  SCL_TOPN_SCORE_BF16X3 nominates the KEEP = 32
 * candidates per (query, reference split) with q.r = q_hi.r_hi + q_hi.r_lo + q_lo.r_hi on the
 * bf16 matrix cores (x = hi + lo, two bf16 each): |score error| <= 1.2e-5 |q||r| instead of
 * the float32 pass's ~1e-7; emitted order and distances are the same float64-exact re-rank.
 * Needs 4 * R * d more workspace bytes (the two reference planes). */
#define SCL_TOPN_SCORE_F32 0
#define SCL_TOPN_SCORE_BF16X3 1
size_t scl_topn_l2_ex_workspace_bytes(int R, int Q, int d, int n, int flags);
int scl_topn_l2_ex(const float* ref, int R, const float* query, int Q, int d, int n,
                   int64_t idx_offset, int64_t* idx_out, double* dist_out, void* workspace,
                   size_t workspace_bytes, int flags, void* stream);

/* The same call with an EXACTNESS CERTIFICATE per query (workspace: the _ex query).  The scan
 * nominates 32 candidates per query by approximate score and the re-rank orders them exactly;
 * the lists equal KDTree.query's (evaluation/top-n.py:103-108) whenever no reference outside
 * the nominated set can reach the n-th exact distance.  That is decided here, not assumed:
 * every outsider scores >= the nomination threshold tau, so its exact squared distance is
 * >= tau + |q|^2 - eps, with eps a rigorous bound of |approximate - exact score| for the mode
 * (float32 or bf16x3 rounding) and the largest reference norm.
 *   uncertified [Q] bytes: 0 = the emitted list is provably the exact top-n; 1 = cannot be
 *               proven (near-duplicate references around the n-th neighbour) — resolve those
 *               queries with scl_topn_exact_filter;
 *   bound_sq    [Q] float64: n-th smallest exact SQUARED distance among the nominated set,
 *               an upper bound of the true n-th squared distance.
 * Both NULL: no certificate (the _ex call). */
int scl_topn_l2_cert(const float* ref, int R, const float* query, int Q, int d, int n,
                     int64_t idx_offset, int64_t* idx_out, double* dist_out,
                     unsigned char* uncertified, double* bound_sq, void* workspace,
                     size_t workspace_bytes, int flags, void* stream);

/* Exact resolution of uncertified queries: for each listed query, float64 sum (q - r)^2 against
 * EVERY reference row (d <= 256, d % 4 == 0); rows with squared distance <= bound_sq[query] are
 * appended (unordered) to that query's candidate list.  The true top-n are the n smallest
 * (distance, row) pairs of the list.
 *   qlist [nq] int32 query rows; bound_sq [Q] float64 indexed by query row;
 *   count [nq] int32 out: candidates found (may exceed cap: then only cap were stored);
 *   cand_d [nq,cap] float64 squared distances; cand_i [nq,cap] int32 reference rows. */
int scl_topn_exact_filter(const float* ref, int R, const float* query, int d, const int* qlist,
                          int nq, const double* bound_sq, int cap, int* count, double* cand_d,
                          int* cand_i, void* stream);

/* Inner products of a block of queries with a block of references for descriptors of ANY width
 * (the in-training localisation check on the raw 32768-wide descriptors, train/train.py:1181-1182;
 * evaluation/top-n.py's sweep over d): out [splits,Q,R] float64; summed over the splits (by the
 * caller, in float64), out[q,r] = sum over chunks of 256 features of the chunk's float32-FMA
 * chain, the chunks added in float64, so that |sum - q.r| <= 1.53e-5 |q||r| + 2^-53 (d / 256 + 1)
 * |q.r| for every d (the bound the caller's exactness certificate uses;
 * evaluation/retrieval._topn_wide).  splits >= 1 cuts the feature axis into runs of whole chunks
 * (one grid layer each: small query / reference blocks then still fill the chip); no split may be
 * empty: splits <= ceil(d / 256) and (splits - 1) * ceil(ceil(d / 256) / splits) < ceil(d / 256). */
int scl_topn_dots(const float* ref, int R, const float* query, int Q, int d, int splits, double* out,
                  void* stream);

/* ------------------------------------------------------------------------- *
 * VGG16 backbone glue — the elementwise ops between the convolutions of
 * model/nets.py:27-63 (tf.layers.conv2d's bias add, tf.nn.relu,
 * tf.layers.max_pooling2d(2, 2)) and their TF-autodiff backward ops, fused so each
 * activation map is streamed once.  They serve the layers whose convolution runs in the
 * library (maps below 30 x 40 in bf16, everything in float32 mode) and the few places where a
 * tail cannot ride on a convolution (the ReLU' under the NetVLAD head, the un-pooling); the
 * hand-written convolutions further down fuse these tails into their epilogues.
 * Activations are channels-last [M = B*H*W, C] (f32 or bf16), C % 8 == 0 and
 * (C / 8) | 256; bias and bias gradients are f32 [C].
 *   scl_vgg_bias_act   y = [relu](y + bias), in place
 *   scl_vgg_act_bwd    gz = g * [a > 0] (a == NULL: no activation, gz untouched) and
 *                      bias_grad[c] = sum_m gz[m, c]
 *   scl_vgg_pool_fwd   a[B,H/2,W/2,C] = relu(maxpool2x2/2 'valid'(z) + bias)
 *                      (== ReLU(pool(z + bias)), nets.py:40-42: max and ReLU commute)
 *   scl_vgg_pool_bwd   gz[B,H,W,C] = gradient of the above w.r.t. z (first maximum of
 *                      each window, recomputed from z) and bias_grad
 * ------------------------------------------------------------------------- */
size_t scl_vgg_workspace_bytes(int C);
int scl_vgg_bias_act(void* y, int dtype, const float* bias, int64_t M, int C, int relu,
                     void* stream);
int scl_vgg_act_bwd(const void* g, const void* a, int dtype, int64_t M, int C, void* gz,
                    float* bias_grad, void* workspace, size_t workspace_bytes, void* stream);
int scl_vgg_pool_fwd(const void* z, int dtype, const float* bias, int B, int H, int W, int C,
                     void* a, void* stream);
int scl_vgg_pool_bwd(const void* g, const void* a, const void* z, int dtype, int B, int H, int W,
                     int C, void* gz, float* bias_grad, void* workspace, size_t workspace_bytes,
                     void* stream);

/* 3x3 / stride 1 / same-padding convolution, 64 -> 64 channels, bf16 channels-last
 * activations, float32 accumulation — VGG16's conv1_2 (model/nets.py:41-42), forward and
 * backward-data, in place of the library kernel.
 *   x, out   [B,H,W,64] bf16
 *   w        bf16 weights, logical [k_out][c_in][3][3] addressed through its element strides
 *   transposed 0: out = conv(x, w)                       (forward)
 *              1: out = conv_transpose(x, w) = d loss / d input for x = d loss / d output
 *   workspace scl_conv64_workspace_bytes() bytes (the packed weight image). */
/* The `transposed` argument of every convolution entry point is a flag word: */
#define SCL_CONV_TRANSPOSED 1 /* backward-data: out = conv_transpose(x, w)                      */
#define SCL_W_PACKED 4        /* w is a packed image written by scl_conv_pack_batch for this     *
                               * (cin, kout, direction); the strides are ignored                 */
#define SCL_W_F32 2           /* w is the float32 master weight (rounded to bf16 while it is    */
                              /* packed: no separate cast pass); default bf16                   */
size_t scl_conv64_workspace_bytes(void);
int scl_conv64(const void* x, const void* w, int64_t w_stride_k, int64_t w_stride_c,
               int64_t w_stride_h, int64_t w_stride_w, int transposed, int B, int H, int W,
               void* out, void* workspace, size_t workspace_bytes, void* stream);

/* The same kernel for the other shapes whose weight slice fits the register file:
 * (cin, kout) in {(64,64), (64,128), (128,64), (128,128)} — conv1_2 and conv2_x, forward and
 * backward-data (for transposed = 1, cin is the weight's k dimension and kout its c dimension).
 * x [B,H,W,cin], out [B,H,W,kout] bf16. */
size_t scl_conv3x3_workspace_bytes(void);
int scl_conv3x3(const void* x, const void* w, int64_t w_stride_k, int64_t w_stride_c,
                int64_t w_stride_h, int64_t w_stride_w, int transposed, int B, int H, int W,
                int cin, int kout, void* out, void* workspace, size_t workspace_bytes,
                void* stream);

/* ... with the layer's elementwise tail fused into the epilogue (model/nets.py:32-63):
 *   pooled == NULL: out = conv + bias (bias may be NULL), ReLU if relu
 *   pooled != NULL: out = raw conv (kept for the backward pass),
 *                   pooled [B,H/2,W/2,kout] = relu(maxpool2x2(out) + bias)
 * bias is float32 [kout]. */
int scl_conv3x3_fused(const void* x, const void* w, int64_t w_stride_k, int64_t w_stride_c,
                      int64_t w_stride_h, int64_t w_stride_w, int transposed, int B, int H,
                      int W, int cin, int kout, void* out, const float* bias, int relu,
                      void* pooled, void* workspace, size_t workspace_bytes, void* stream);

/* The deeper layers (conv3_x .. conv5_x: cin % 32 == 0, kout % 128 == 0, up to 1024): weights
 * streamed through LDS, [12 / 8 / 6 x 40 pixel] x 128-channel workgroup tiles — csrc/convh.hip
 * (v_mfma_f32_16x16x32_bf16; cin % 64 == 0) or csrc/convg.hip (32x32x16; any cin % 32 == 0),
 * same results bit for bit.  Same arguments as scl_conv3x3_fused without the pooled output. */
size_t scl_convg_workspace_bytes(int cin, int kout);
int scl_convg(const void* x, const void* w, int64_t w_stride_k, int64_t w_stride_c,
              int64_t w_stride_h, int64_t w_stride_w, int transposed, int B, int H, int W,
              int cin, int kout, void* out, const float* bias, int relu, void* workspace,
              size_t workspace_bytes, void* stream);

/* conv -> +bias -> max-pool 2x2/2 -> ReLU (model/nets.py:40-42) WITHOUT writing the full-size
 * convolution output: pooled [B,H/2,W/2,kout] bf16 = relu(maxpool2x2(conv) + bias) and
 * pool_idx [B,H/2,W/2,kout] uint8 = window position 2*dy + dx of the first maximum in raster
 * order (on the float32 accumulators), which is all the backward pass needs
 * (scl_vgg_pool_bwd_idx).  Shapes and workspace of scl_conv3x3; forward only. */
int scl_conv3x3_pool_idx(const void* x, const void* w, int64_t w_stride_k, int64_t w_stride_c,
                         int64_t w_stride_h, int64_t w_stride_w, int flags /* SCL_W_F32 or 0 */,
                         int B, int H, int W, int cin, int kout, const float* bias, void* pooled,
                         void* pool_idx, void* workspace, size_t workspace_bytes, void* stream);
/* The same for the LDS-weights shapes of scl_convg (conv3_3 / conv4_3); workspace as scl_convg. */
int scl_convg_pool_idx(const void* x, const void* w, int64_t w_stride_k, int64_t w_stride_c,
                       int64_t w_stride_h, int64_t w_stride_w, int flags /* SCL_W_F32 or 0 */,
                       int B, int H, int W, int cin, int kout, const float* bias, void* pooled,
                       void* pool_idx, void* workspace, size_t workspace_bytes, void* stream);
/* Pool + ReLU backward from that index map: gz [B,H,W,C] = g * [a > 0] at the stored window
 * position, zero elsewhere (and on rows / columns no window covers); bias_grad[c] = sum of
 * g * [a > 0].  Arguments as scl_vgg_pool_bwd with idx in place of z; a == NULL: g is already
 * masked (the backward-data kernel of the layer above applied [a > 0] in its epilogue, a being
 * its own input) and a is not read. */
int scl_vgg_pool_bwd_idx(const void* g, const void* a, const void* idx, int dtype, int B, int H,
                         int W, int C, void* gz, float* bias_grad, void* workspace,
                         size_t workspace_bytes, void* stream);

/* Backward-data with the ReLU' of the layer below in the epilogue (the backward of
 * model/nets.py:39's relu chained to the conv beneath it): out = conv(x, w) * [mask > 0],
 * mask [B,H,W,kout] bf16 the post-activation map the gradient flows into — saves the separate
 * read-modify-write pass over the gradient map.  scl_conv3x3_masked takes the shapes of
 * scl_conv3x3, scl_convg_masked those of scl_convg; workspaces as for those. */
int scl_conv3x3_masked(const void* x, const void* w, int64_t w_stride_k, int64_t w_stride_c,
                       int64_t w_stride_h, int64_t w_stride_w, int transposed, int B, int H,
                       int W, int cin, int kout, void* out, const void* mask, void* workspace,
                       size_t workspace_bytes, void* stream);
/* ... for the layer under a 2x2 max-pooling (cin == kout: 64 or 128): g_pooled [B,H/2,W/2,cin] is
 * the gradient at the pooled map (ReLU' applied), pool_idx the uint8 window positions of
 * scl_conv3x3_pool_idx; the kernel un-pools into its LDS window while staging — bit-identical
 * to scl_vgg_pool_bwd_idx followed by scl_conv3x3_masked, without the full-size gradient in
 * memory.  flags: SCL_CONV_TRANSPOSED | SCL_W_F32 | SCL_W_PACKED as for scl_conv3x3.  H, W even. */
int scl_conv3x3_masked_pooled(const void* g_pooled, const void* pool_idx, const void* w,
                              int64_t w_stride_k, int64_t w_stride_c, int64_t w_stride_h,
                              int64_t w_stride_w, int flags, int B, int H, int W, int cin,
                              int kout, void* out, const void* mask, void* workspace,
                              size_t workspace_bytes, void* stream);
int scl_convg_masked(const void* x, const void* w, int64_t w_stride_k, int64_t w_stride_c,
                     int64_t w_stride_h, int64_t w_stride_w, int transposed, int B, int H, int W,
                     int cin, int kout, void* out, const void* mask, void* workspace,
                     size_t workspace_bytes, void* stream);

/* The first layer in one pass (model/nets.py:22-24, 39): x0 = bf16(img - average_rgb),
 * y = relu(conv3x3(x0, w) + bias).  img [B,H,W,3] float32 (raw 0..255), avg [3], w bf16 logical
 * [64][3][3][3] at the given element strides, bias float32 [64]; x0 [B,H,W,3] bf16 (kept for the
 * weight gradient), y [B,H,W,64] bf16. */
int scl_conv_first(const float* img, const float* avg, const void* w, int64_t w_stride_k,
                   int64_t w_stride_c, int64_t w_stride_h, int64_t w_stride_w,
                   int w_f32 /* w is float32 instead of bf16 */, const float* bias, int B, int H,
                   int W, void* x0, void* y, void* stream);

/* The first TWO layers of the forward pass in one kernel (model/nets.py:22-24, 39-42; round 4):
 * what scl_conv_first followed by scl_conv3x3_pool_idx(64 -> 64) compute, value for value —
 * x0 and y1 = relu(conv1_1 + bias1) as above (both are written: the backward pass reads them),
 * pooled [B,H/2,W/2,64] bf16 = relu(maxpool2x2(conv1_2(y1)) + bias2) and pool_idx (uint8, same
 * shape: the window position 2 dy + dx of each maximum) — but a tile's halo window of y1 is
 * computed into LDS from the image instead of read back from memory.  w1 as in scl_conv_first;
 * w2 logical [64][64][3][3] at its element strides with w2_flags = SCL_W_F32 or 0, or the image
 * scl_conv_pack_batch wrote (SCL_W_PACKED); workspace >= scl_conv3x3_workspace_bytes(). */
int scl_conv_first_pool_idx(const float* img, const float* avg, const void* w1,
                            int64_t w1_stride_k, int64_t w1_stride_c, int64_t w1_stride_h,
                            int64_t w1_stride_w, int w1_f32, const float* bias1, const void* w2,
                            int64_t w2_stride_k, int64_t w2_stride_c, int64_t w2_stride_h,
                            int64_t w2_stride_w, int w2_flags, const float* bias2, int B, int H,
                            int W, void* x0, void* y1, void* pooled, void* pool_idx,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Weight AND bias gradient of the first layer in one pass over its gradient map (the backward
 * of model/nets.py:39's conv1_1 + bias): gw[k][c][kh][kw] = sum gz[b,y,x,k] * x0[b,y+kh-1,
 * x+kw-1,c] (bf16 at the given element strides), gb[k] = sum gz[b,y,x,k] (float32 [64]).
 * x0 [B,H,W,3] bf16 (as written by scl_conv_first), gz [B,H,W,64] bf16.  Deterministic.
 * davg != NULL also returns the gradient of the trainable mean (model/nets.py:22-24) in closed
 * form, davg[c] = - sum w[k][c][kh][kw] * (sum of gz where that tap stays inside the image),
 * from the same pass (w: the layer's bf16 weight at the strides given for gw) — the first
 * layer's backward-data pass is never needed. */
size_t scl_conv_first_wrw_workspace_bytes(void);
int scl_conv_first_wrw(const void* x0, const void* gz, int B, int H, int W, void* gw,
                       int64_t w_stride_k, int64_t w_stride_c, int64_t w_stride_h,
                       int64_t w_stride_w, int w_f32 /* gw and w are float32 instead of bf16 */,
                       float* gb, const void* w, float* davg, void* workspace,
                       size_t workspace_bytes, void* stream);

/* conv1_2's backward-data pass and the first layer's parameter gradients in ONE launch (round 5,
 * ABI 11; train/train.py:877-878's gradient of model/nets.py:39-42): the gradient at conv1_1's
 * pre-activation, gz1 = conv_transpose(unpool(g_pooled, pool_idx), w2) * [mask > 0] — 944 MB at
 * 24 x 640x480 — has ONE consumer, scl_conv_first_wrw (conv1_1's input is the image: no
 * backward-data pass below it), so it is never written: every [8 x 32]-pixel tile of it is
 * multiplied with the im2col of x0 while it sits in LDS, and gw1 / gb1 / davg come out exactly as
 * scl_conv_first_wrw(x0, gz1, ...) defines them (other summation order: float32 rounding differs).
 * g_pooled [B,H/2,W/2,64] bf16 + pool_idx (uint8, same shape: scl_conv3x3_pool_idx's), w2 logical
 * [64][64][3][3] at its element strides (flags: SCL_W_F32, or SCL_W_PACKED for the
 * backward-direction image of scl_conv_pack_batch), mask [B,H,W,64] bf16 = conv1_1's output, x0
 * [B,H,W,3] bf16; gw1 / gb1 / w1 / davg as in scl_conv_first_wrw.  H and W even, B <= 8192.
 * workspace >= scl_conv3x3_workspace_bytes(), fw_workspace >= scl_conv_first_wrw_workspace_bytes().
 * Deterministic for a fixed scl_set_reserve_cus value (one slab per workgroup, fixed-order sums). */
int scl_conv3x3_masked_pooled_first_wrw(
    const void* g_pooled, const void* pool_idx, const void* w2, int64_t w2_stride_k, int64_t w2_stride_c,
    int64_t w2_stride_h, int64_t w2_stride_w, int flags, int B, int H, int W, const void* mask,
    const void* x0, void* gw1, int64_t w1_stride_k, int64_t w1_stride_c, int64_t w1_stride_h,
    int64_t w1_stride_w, int w1_f32, float* gb1, const void* w1, float* davg, void* workspace,
    size_t workspace_bytes, void* fw_workspace, size_t fw_workspace_bytes, void* stream);

/* Weight gradient of the same layer: gw[k][c][kh][kw] = sum_{b,y,x} gz[b,y,x,k] *
 * x[b, y+kh-1, x+kw-1, c]; x, gz [B,H,W,64] bf16, gw bf16 written at the given element
 * strides (logical [64][64][3][3]).  Deterministic (per-CU slabs summed in a fixed order). */
size_t scl_wrw64_workspace_bytes(void);
int scl_wrw64(const void* x, const void* gz, int B, int H, int W, void* gw, int64_t w_stride_k,
              int64_t w_stride_c, int64_t w_stride_h, int64_t w_stride_w, void* workspace,
              size_t workspace_bytes, void* stream);

/* The weight gradient for any cin, kout that are multiples of 64 (up to 1024): the same kernel
 * per [64 c] x [64 k] block, pixels split over workgroups, slabs reduced in a fixed order.
 * x [B,H,W,cin], gz [B,H,W,kout] bf16; gw bf16 logical [kout][cin][3][3] at the given strides. */
size_t scl_wrw3x3_workspace_bytes(int cin, int kout);
int scl_wrw3x3(const void* x, const void* gz, int B, int H, int W, int cin, int kout, void* gw,
               int64_t w_stride_k, int64_t w_stride_c, int64_t w_stride_h, int64_t w_stride_w,
               void* workspace, size_t workspace_bytes, void* stream);
/* ... with the gradient written as float32 (gw_f32 != 0) instead of bf16. */
int scl_wrw3x3_ex(const void* x, const void* gz, int B, int H, int W, int cin, int kout, void* gw,
                  int64_t w_stride_k, int64_t w_stride_c, int64_t w_stride_h, int64_t w_stride_w,
                  int gw_f32, void* workspace, size_t workspace_bytes, void* stream);
/* ... and the bias gradient grad_bias[k] = sum over pixels of gz[.., k] (float32 [kout], NULL =
 * not wanted) from the same pass: gz is in LDS for the weight gradient anyway, so the separate
 * column-sum pass over the gradient map (scl_vgg_act_bwd without a mask) is not needed
 * (tf.nn.bias_add's gradient under model/nets.py:39-63). */
int scl_wrw3x3_bias(const void* x, const void* gz, int B, int H, int W, int cin, int kout, void* gw,
                    int64_t w_stride_k, int64_t w_stride_c, int64_t w_stride_h, int64_t w_stride_w,
                    int gw_f32, float* grad_bias, void* workspace, size_t workspace_bytes,
                    void* stream);
/* The same for a layer that ends in the 2x2 max-pooling (model/nets.py:40-42, conv1_2 .. conv4_3):
 * g_pooled [B,H/2,W/2,kout] bf16 is the gradient at the POOLED map (ReLU' applied) and pool_idx
 * the uint8 window positions scl_conv3x3_pool_idx / scl_convg_pool_idx wrote in the forward
 * pass.  The full-size gradient (one non-zero per window and channel) is built tile by tile in
 * LDS and never exists in memory: bit-identical to scl_vgg_pool_bwd_idx followed by
 * scl_wrw3x3_bias, without that pass's write and this one's read of the full-size map.
 * H and W even. */
int scl_wrw3x3_pooled(const void* x, const void* g_pooled, const void* pool_idx, int B, int H,
                      int W, int cin, int kout, void* gw, int64_t w_stride_k, int64_t w_stride_c,
                      int64_t w_stride_h, int64_t w_stride_w, int gw_f32, float* grad_bias,
                      void* workspace, size_t workspace_bytes, void* stream);

/* All packed weight images of a training step in ONE launch (the convolutions above otherwise
 * each run a 4-10 us packing kernel in front of themselves: 24 per step on the critical
 * path).  The weights change once per step (train/train.py:877-879), so the caller packs every
 * (layer, direction) up front and passes the images with SCL_W_PACKED in the flags argument.
 * jobs is a HOST array; packed buffers are scl_conv_packed_bytes(cin, kout) each (0 = the
 * shape has no packed form: cin % 64, kout % 128 for the LDS-weights kernels), 256-byte
 * aligned.  flags per job: SCL_CONV_TRANSPOSED and / or SCL_W_F32; or SCL_PACK_VLAD_W alone: the
 * job writes the NetVLAD plane images of assign_w (see scl_netvlad_fwd_p) instead. */
#define SCL_PACK_VLAD_W 16
typedef struct SclPackJob {
  const void* w;
  int64_t w_stride_k, w_stride_c, w_stride_h, w_stride_w;
  int flags, cin, kout;      /* cin = contraction channels, kout = output channels of the pass */
  void* packed;
} SclPackJob;
size_t scl_conv_packed_bytes(int cin, int kout);
int scl_conv_pack_batch(const SclPackJob* jobs, int njobs, void* stream);

/* ------------------------------------------------------------------------- *
 * Diagnostics (bench.py's live per-kernel timing; the reference has no counterpart
 * beyond its wall-clock prints, train/train.py:135-161).  Between scl_prof_begin and
 * scl_prof_end every kernel launched through the library BY ANY THREAD of the process
 * (PyTorch runs backward on its own thread) is bracketed by HIP events on its launch stream;
 * one sink at a time.  scl_prof_end waits for them and returns per-launch milliseconds and
 * kernel names (static strings); call it once the launching threads are quiescent.
 * ------------------------------------------------------------------------- */
/* Device calibration (csrc/calibrate.hip): a bare bf16 MFMA loop — `workgroups` workgroups of 512
 * threads, each wave a [128 x 64] accumulator block fed from LDS, `iters` 32-deep k-steps, no global
 * traffic — on `shape` = 32 (v_mfma_f32_32x32x16_bf16) or 16 (v_mfma_f32_16x16x32_bf16).
 * `operands`: 64 KB of bf16 values (16-byte aligned; random values: zeros draw less power and
 * over-state what the device sustains); `sink`: `workgroups` floats (one checksum each).  No
 * counterpart in the reference: bench.py times it beside the convolution kernels so that the
 * datasheet peak its `roofline` is priced against comes with what this device, at its power cap,
 * sustains (`roofline.sustained`).  scl_calibrate_mfma_bf16_flops = the FLOPs one launch executes. */
int scl_calibrate_mfma_bf16(int shape, int workgroups, int iters, const void* operands, float* sink,
                            void* stream);
double scl_calibrate_mfma_bf16_flops(int workgroups, int iters);
/* 1 for libscl_hip_diag.so (-DSCL_DIAG), 0 for the product library. */
int scl_build_is_diag(void);
/* Ablation / tuning switch (scripts/ablate_rowtile.py, scripts/microbench.py) of the DIAGNOSTIC
 * build; 0 restores the production kernels, any other value makes RESULTS MEANINGLESS unless
 * marked CORRECT below.  Returns the old value.  In the product library (no -DSCL_DIAG) the
 * variants do not exist: 0 is accepted (returns 0), anything else returns SCL_E_KIND.
 *   1..7          forward row-tile kernel: bit 0 no epilogue stores, bit 1 no x loads,
 *                 bit 2 no operand staging / barriers
 *   8             bf16 feature maps through the float32-MFMA NetVLAD kernels (CORRECT results:
 *                 the A/B partner of the fused bf16 kernels)
 *   916 / 917 / 918   fused NetVLAD kernels (four-wave forward / grad_x / eight-wave forward):
 *                 wave 0 of every workgroup writes shader-clock stamps into the tail of the
 *                 workspace (scripts/vlad_stamps.py)
 *   920 / 922     NetVLAD head with round 3's launch structure and four-wave kernels / the
 *                 four-wave kernels in round 4's structure (CORRECT results: A/B partners)
 *   921           sibling exchange of the NetVLAD finish / backward prologue: wait limit 0, every
 *                 workgroup takes the self-computing path (CORRECT, bit-identical results)
 *   31            Gram loss, 64 < B <= 256: float32-MFMA Gram instead of bf16x6
 *   32 / 33 / 34  Gram loss: the two-launch forward at B <= 32 and the older guarded backward /
 *                 the 32-column backward of the own rows / the float32-MFMA backward where the
 *                 bf16-plane one would run (CORRECT results: A/B partners)
 *   36            Gram loss: the one-launch forward for 32 < B <= 64 as well (CORRECT, bit-identical
 *                 to the four launches of the product path; measured slower)
 *   2200          weight-gradient kernel: round 4's staging without the buffer-resource path
 *                 (CORRECT results: A/B partner)
 *   2300          weight-gradient kernel: every 32x32x16 product issued as two 16x16x32 MFMAs on the
 *                 same registers (timing only: would the other MFMA shape pay here?  It does not)
 *   100000 * s    Gram loss, B <= 256: force s K-splits
 *   100 + s       top-n: force s reference splits (1..32) instead of the planner's choice
 *   1000 * b (+ 100 + s)   top-n scan: b bit 0 no selection, bit 1 no tile staging,
 *                 bit 2 no MFMAs (values below 8000 only)
 *   2000 + bits / 2100     weight-gradient kernel: timing ablations / one n-block per workgroup
 *   3006 / 3008 / 3012   LDS-weights convolution: pin the block height (6: 16x16x32 kernel only)
 *   3099          LDS-weights convolution: one tile per workgroup instead of persistent ones
 *   3100 + g      LDS-weights convolution: persistent grid of g + 1 groups of 8 * kout / 128
 *   40000 + v / 50000 + v  LDS-weights convolution: pin the v_mfma 32x32x16 kernel
 *                 (csrc/convg.hip) / the 16x16x32 kernel (csrc/convh.hip, the default), with
 *                 v = 0 or one of the 3xxx values above — these two give CORRECT results
 *   60000 + bits  register-weights convolution: bit 0 no window staging after the first
 *                 tile, bit 1 no output stores, bit 2 (and 53040 for the LDS-weights kernel)
 *                 static wave priorities around the MFMA runs (CORRECT results; measured without
 *                 effect, scripts/setprio_ab.sh) */
int scl_debug_set_variant(int variant);
/* CUs the persistent convolution grids leave free for other kernels (RCCL's, with more than one
 * rank per node: DESIGN.md section 4).  Default: the environment variable SCL_RESERVE_CUS, or 0.
 * Process-wide; takes effect at the next launch; returns the previous value; a negative n
 * re-reads the environment.  Results are deterministic for a fixed value (fixed-order reductions);
 * between values the weight and bias gradients differ in ROUNDING: the weight-gradient kernels cut
 * the pixels into one slab per usable CU, so the number of partial sums — and the order in which
 * they are added — follows the setting.  Record the value next to anything that must be
 * reproduced bit for bit (bench.py prints it in `switches`; the trainer's checkpoints carry it as
 * the int32 variable `scl/reserve_cus`).  The fused NetVLAD kernels (csrc/netvlad.hip) do not honour it: their grids are
 * one workgroup per (image, location slice). */
int scl_set_reserve_cus(int n);
/* The value in force (the environment variable is resolved on the way); changes nothing. */
int scl_get_reserve_cus(void);
int scl_prof_begin(int capacity);
int scl_prof_count(void);
int scl_prof_end(float* ms, const char** names, int capacity);
/* Launches an empty kernel (256 x 256 threads) through the same bracket: its event duration minus
 * its device time is what the bracket adds to every kernel (bench.py's correction). */
int scl_prof_null(void* stream);

/* ------------------------------------------------------------------------- *
 * Host utility for the checkpoint bundle reader / writer (tf_bundle.py; the reference
 * restores and saves through tf.train.Saver, train/train.py:882-905, 984, 1079, 1102):
 * CRC-32C (Castagnoli, reflected 0x82F63B78) of n bytes continued from `crc` (0 to start),
 * unmasked.  Runs on the host; touches no device.
 * ------------------------------------------------------------------------- */
unsigned scl_crc32c(unsigned crc, const void* data, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* SCL_HIP_H */
