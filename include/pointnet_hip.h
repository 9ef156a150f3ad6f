/*
 * pointnet_hip.h -- C ABI of libpointnet_hip.so, the MI355X (gfx950) PointNet hot path.
 *
 * The reference (MAPieschl/PointCloudProcessing) is pure Python on TensorFlow/Keras and has no FFI of
 * its own; the boundary it exposes for this path is the Python module API of
 *   point_cloud_analysis/pointnet/PointNet.py      (PointNet, TNet, ConvLayer, DenseLayer, PointCloudNormalization)
 *   point_cloud_analysis/pointcloud/PointCloudSet.py
 *   point_cloud_analysis/pointnet_train.py
 * Each entry point below names the reference call site (file:line, relative to
 * /root/reference/point_cloud_analysis/) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns PN_OK (0) or a negative pn_status; pn_last_error() gives the text
 *     (thread-local);
 *   - the caller owns every buffer: all pointers are DEVICE pointers unless a parameter says "host";
 *     the library never allocates or frees device memory, keeps no global state, never synchronises
 *     the host, and only enqueues work on the `stream` it is given (hipStream_t passed as void*), so
 *     every call is safe to capture into a hipGraph;
 *   - tensors are row-major and contiguous; point tensors are (B clouds) x (N points) x channels,
 *     flattened to M = B*N rows; floating point storage is fp32, except the per-point layer-boundary
 *     tensors where the caller asks for bf16 (PN_STORE_BF16, pn_operand.h16);
 *   - `prec` selects the arithmetic of the per-point contractions with K >= 64, which run on the bf16
 *     MFMA pipe with fp32 accumulation:  PN_PREC_BF16  = operands rounded to bf16 (1 MFMA per product),
 *     PN_PREC_BF16X3 = operands split hi+lo into two bf16 each, 3 MFMAs per product (16 significant
 *     bits per operand, ~1e-5 relative).  Everything else (K = 3 layers, per-cloud dense layers,
 *     statistics, normalisation, losses, optimizer) is fp32 on the vector ALU.
 */
#ifndef POINTNET_HIP_H
#define POINTNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN_ABI_VERSION 6

typedef enum {
  PN_OK = 0,
  PN_ERR_INVALID_ARGUMENT = -1,
  PN_ERR_LAUNCH = -2,
  PN_ERR_UNSUPPORTED = -3,
  PN_ERR_WORKSPACE = -4
} pn_status;

#define PN_PREC_BF16 1
#define PN_PREC_BF16X3 3
/* OR-ed into `prec` of the entry points that WRITE (or read through plain pointers) per-point tensors -- z of pn_conv_fwd; out,
 * addend and zmask of pn_conv_bwd_data; pn_model_desc.prec for every layer-boundary tensor of the model plan (Z, dy, X64, the
 * max-pool backward's addend) -- those tensors are then stored as bf16 (round to nearest even of the fp32 value; 2 bytes per
 * element, same row-major shape) instead of fp32.  The model plan accepts it with PN_PREC_BF16 only.  The layer-boundary tensors of a
 * training step are its HBM traffic, and the contraction that consumes them rounds its operands to bf16 anyway under PN_PREC_BF16.
 * Operands say the same about their sources with pn_operand.h16.  Statistics are always taken from the fp32 values before rounding. */
#define PN_STORE_BF16 0x100
#define PN_IO_KEEP_ACTIVATIONS 1 /* pn_model_io.flags */

typedef void* pn_stream; /* hipStream_t */

int pn_abi_version(void);
const char* pn_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * A "lazy" per-point operand: the value the contraction sees for row m, channel k is
 *     v = max(lo,  ca[k] * s1[m*ld + k]  +  cb[k] * s2[m*ld + k]  +  cc[k])
 * with ca/cb/cc optional (NULL => 1, 0, 0) and s2 optional (NULL => term dropped).
 * This is how training-mode BatchNormalization never costs a pass of its own:
 *   forward :  s1 = previous layer's pre-BN output z, ca = gamma*rsqrt(var+eps), cc = beta - mean*ca,
 *              lo = 0  ==>  v = relu(bn(z))                      (PointNet.py:559-562)
 *   backward:  s1 = dL/dy_hat, s2 = z, (ca,cb,cc) from pn_bn_bwd_finalize  ==>  v = dL/dz through the
 *              batch statistics.
 * ---------------------------------------------------------------------------------------------- */
typedef struct pn_operand {
  const float* s1;
  const float* s2;
  const float* ca;
  const float* cb;
  const float* cc;
  int64_t ld; /* elements between consecutive rows of s1/s2 */
  float lo;   /* lower clamp: 0 for ReLU, -INFINITY for none */
  int32_t h16; /* 0: s1/s2 point to fp32 arrays; 1: to bf16 arrays (ld still in elements, rows 16-byte aligned: ld % 8 == 0) */
} pn_operand;

/* --- PointCloudNormalization.call  (pointnet/PointNet.py:691-706) --------------------------------
 * xyz (B,N,3) -> out (B,N,3) = (xyz - centroid) / max(max_n |xyz - centroid|, 1e-7);
 * centroid (B,3), scale (B) are also returned (the layer's second output). */
int pn_normalize(const float* xyz, int B, int N, float* out, float* centroid, float* scale, pn_stream stream);

/* --- ConvLayer with Cin = 3 (1x1 Conv2D, pointnet/PointNet.py:535-542,556; first layer of
 * input_transform :406 and mlp_1_1 :120).  z[m,:] = x3[m,:] . W[b]  with W (3,C) shared
 * (w_cloud_stride = 0) or one (3,C) matrix per cloud (w_cloud_stride = 3*C: the 3x3 input transform
 * tf.matmul(pc, R) of :207 folded into the kernel: (pc.R).W = pc.(R.W)).
 * Also emits per-row-tile partial sums  part[tile][0][c] = sum z, part[tile][1][c] = sum z^2
 * (tile = 128 rows of one cloud; n_tiles = B*ceil(N/128)) for the BatchNormalization statistics. */
int pn_conv3_fwd(const float* x3, const float* w, int64_t w_cloud_stride, int B, int N, int C, float* z,
                 float* stat_partials, pn_stream stream);

/* weight gradient of the above: slabs[tile][3][C] = sum_rows x3[row,:]^T dz[row,:], dz given lazily */
int pn_conv3_wgrad(const float* x3, const pn_operand* dz, int B, int N, int C, float* slabs, pn_stream stream);

/* --- ConvLayer forward, Cin in {64..1024}: z = relu(bn(x)) . W, stored pre-BN, plus BN partial sums.
 * (pointnet/PointNet.py:554-566 for the layer; the lazy operand folds the PREVIOUS layer's BN+ReLU.)
 *   x        lazy operand over (B*N, K)
 *   w        (K, C) row-major [Keras kernel (1,1,K,C)]; w_cloud_stride != 0 => one matrix per cloud
 *            (tf.matmul(X, R_64), PointNet.py:228, is this call with K = C = 64, w = R_64, stride 4096)
 *   cloud_bias optional (B, C) added per cloud before the statistics (the global-feature half of
 *            seg_l1's kernel applied to the tiled global vector, PointNet.py:268-275)
 *   z        (B*N, C) or NULL (no store)
 *   stat_partials  [n_tiles][2][C] or NULL */
int pn_conv_fwd(const pn_operand* x, const float* w, int64_t w_cloud_stride, int B, int N, int K, int C,
                const float* cloud_bias, float* z, float* stat_partials, int prec, pn_stream stream);

/* --- ConvLayer (128 -> 1024) fused with tf.reduce_max(X, axis=1)  (PointNet.py:242-248, 425-429).
 * Never stores the (B,N,C) tensor.  Because BN (per-channel affine) followed by ReLU is monotone in z
 * with the sign of gamma, max_n relu(bn(z)) = relu(bn(sgn * max_n(sgn*z))).  Emits per tile
 *   pmax[tile][c] = max over the tile's rows of sgn[c]*z,  pidx[tile][c] = its row index inside the cloud
 *   (lowest index on ties), and the same stat_partials as pn_conv_fwd. */
int pn_conv_fwd_max(const pn_operand* x, const float* w, int B, int N, int K, int C, const float* sgn,
                    float* pmax, int32_t* pidx, float* stat_partials, int prec, pn_stream stream);

/* --- the same layer as a KERNEL-STATIONARY ROW-PANEL kernel (the one the model plan uses; pn_panel.hip): every cloud is cut into
 * pn_panel_slots_per_cloud(B, N) contiguous runs of 64-row panels, one workgroup per run; its eight waves keep the kernel columns
 * they own in registers for the whole launch and the run's panels stream through a double-buffered LDS image.  The kernel is read
 * from a fragment-ordered bf16 copy made by pn_weights_prep:
 *     wf_hi[((cb * K/16 + ks) * 64 + lane) * 8 + j] = bf16(s_c * W[k][c]),  c = cb*32 + (lane & 31),  k = ks*16 + (lane >> 5)*8 + j
 *     wf_lo = bf16(s_c * W - hi) (needed for PN_PREC_BF16X3 only);  s_c = -1 where sgn[c] < 0 (sgn may be gamma itself; NULL = +1).
 * K in {64, 128}; C = 256, 512 or a multiple of 1024; slots = B * pn_panel_slots_per_cloud(B, N).  Per slot and channel the kernel emits
 *     pmax   = max over the run's rows of s_c * z,      pblock = index inside the cloud of the 32-row block holding it (lowest on ties),
 *     sumsq  = sum over the rows of z^2,
 * and ADDS to colacc[cloud][NT * K] the column sums of the cloud's staged operand rows (the bf16 hi image, then -- bf16x3, NT = 2 -- the
 * lo image; NT = 1 otherwise) as 64-bit fixed point, unit 2^-24 (integer adds: the result does not depend on the order the cloud's
 * workgroups arrive in; the caller zeroes colacc before the launch -- the model plan's first launch does).  The channel sums of z are
 * not accumulated element by element, nor per slot: the finaliser forms sum z[:, c] = (sum over the clouds of colacc) . W[:, c] once
 * per launch      (sumsq and colacc: both or neither; NULL for inference). */
int pn_weights_prep(const float* w, const float* sgn, int K, int C, void* wf_hi, void* wf_lo, pn_stream stream);
int pn_panel_slots_per_cloud(int B, int N);
int pn_conv_fwd_max_panel(const pn_operand* x, const void* wf_hi, const void* wf_lo, int B, int N, int K, int C, float* pmax,
                          int32_t* pblock, float* sumsq, int64_t* colacc, int prec, pn_stream stream);
/* finaliser of the panel kernel: BatchNormalization coefficients of the layer (as pn_bn_finalize; batch statistics from the slots'
 * colacc / sumsq and the SAME kernel copies wf_hi / wf_lo and prec the panel launch was given, or the moving statistics -- then colacc,
 * sumsq and the copies may be NULL) AND tf.reduce_max over each cloud's slots:  zstar[b][c] = s_c * max,
 * g[b][c] = relu(scale*zstar + shift), arg_block[b][c] = the 32-row block of cloud b holding the row of the maximum. */
int pn_panel_finalize(const float* pmax, const int32_t* pblock, const float* sumsq, const int64_t* colacc, const void* wf_hi, const void* wf_lo,
                      int prec, int B, int N, int K, int C, const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                      float momentum, float eps, int use_batch_stats, int update_moving, float* mean, float* invstd, float* scale, float* shift,
                      float* g, float* zstar, int32_t* arg_block, pn_stream stream);
/* --- inference: a whole max-pooled chain in ONE launch -- ConvLayer(3 | 64 -> 64) -> ConvLayer(64 -> 128) -> ConvLayer(128 -> 1024) ->
 * tf.reduce_max (PointNet.py:236-248 mlp_2; :421-429 the two T-Nets), every BatchNormalization on its moving statistics (scale / shift
 * given per layer).  The two narrow layers' outputs stay in LDS; pmax / pblock are what the three launches pn_conv_fwd (or pn_conv3_fwd),
 * pn_conv_fwd, pn_conv_fwd_max_panel leave in the bf16-storage mode (PN_PREC_BF16 | PN_STORE_BF16), bit for bit: feed them to
 * pn_panel_finalize with use_batch_stats = 0.  Exactly one input: x = a 64-channel lazy operand stored as bf16 with w1t = the first
 * layer's kernel as pn_weights_copy16 transposes it, or xyz = the normalised cloud (B*N, 3) with w1 = the first layer's (3, 64) kernel.
 * w2t: the second layer's (64, 128) kernel through pn_weights_copy16 (transposed copy); wf_hi: the third layer's through
 * pn_weights_prep (with the sign of its gamma). */
int pn_chain_fwd_max(const pn_operand* x, const float* xyz, const float* w1, const void* w1t, const float* scale1, const float* shift1,
                     const void* w2t, const float* scale2, const float* shift2, const void* wf_hi, int B, int N, float* pmax,
                     int32_t* pblock, pn_stream stream);
/* bf16 copies (round to nearest even) of a Keras kernel (K, C), K and C multiples of 8, as the model plan's first launch makes them for
 * its row GEMMs: w16[k * C + c] = bf16(w[k][c]) (the kernel as it is) and wt16[c * K + k] = bf16(w[k][c]) (transposed, k contiguous:
 * what the forward row GEMMs and pn_chain_fwd_max stage); both required */
int pn_weights_copy16(const float* w, int K, int C, void* w16, void* wt16, pn_stream stream);

/* the row of the maximum itself (needed by the backward pass only, where the model plan resolves it inside its scatter kernel):
 * arg[b][c] = the row of cloud b, inside block arg_block[b][c], with the largest s_c * z -- the 32 candidates re-evaluated in fp32 from
 * the same bf16-rounded operands, lowest row on ties (exact ties = duplicated points, as the reference's padding produces). */
int pn_max_resolve(const pn_operand* x, const void* wf_hi, const void* wf_lo, const int32_t* arg_block, int B, int N, int K, int C,
                   int32_t* arg, int prec, pn_stream stream);

/* the sparse term of the backward pass of ConvLayer + BatchNormalization + tf.reduce_max (pointnet/PointNet.py:242-248, 425-429; the
 * gradient TensorFlow's tape sends to the arg-max point of every (cloud, channel), here with the layer's kernel already applied):
 *     D[b][n][k] = q[k] + sum over the channels c with arg[b][c] == n of hs[b][c] * wt[c][k]
 * arg (B, C) rows of the maxima (pn_max_resolve), hs (B, C) the pooled gradients already scaled by the BatchNormalization scale,
 * wt (C, K) the kernel transposed (fp32), q (K); D (B*N, K) written in full, fp32 or bf16 (store16).  K a multiple of 32, at most 128.
 * The sum of a row runs in ascending channel order on the matrix cores with both operands split into bf16 hi + lo (three products):
 * exact for operands of at most 16 significant bits, bitwise reproducible. */
int pn_maxbwd_scatter(const int32_t* arg, const float* hs, const float* wt, const float* q, int B, int N, int K, int C, float* D, int store16,
                      pn_stream stream);

/* --- data gradient of a ConvLayer: out = [relu-mask] (dz . W^T + addend), plus the two partial sums
 * BatchNormalization's backward needs (sum dy_hat, sum dy_hat*z) per channel.
 *   dz      lazy operand over (B*N, K)   (K = the layer's output width)
 *   w       (C, K) row-major             (C = the layer's input width; i.e. the Keras kernel as stored)
 *   addend  optional (B*N, C) added before masking (second consumer of the same activation)
 *   zmask/msc/msh  optional: previous layer's pre-BN z and its BN scale/shift; mask = (msc*z+msh > 0)
 *   stat_partials  [n_tiles][2][C] or NULL */
int pn_conv_bwd_data(const pn_operand* dz, const float* w, int64_t w_cloud_stride, int B, int N, int K, int C,
                     const float* addend, const float* zmask, const float* msc, const float* msh, float* out,
                     float* stat_partials, int prec, pn_stream stream);

/* --- weight gradient / Gram matrix: slabs[s][i][j] = sum over the slab's rows of a[row,i]*b[row,j].
 * slab_rows must be a multiple of 64; slabs are per cloud: n_slabs = B*ceil(N/slab_rows).  Reduce with
 * pn_slab_reduce (fixed order => bitwise reproducible). */
int pn_conv_wgrad(const pn_operand* a, const pn_operand* b, int B, int N, int Ci, int Cj, int slab_rows,
                  float* slabs, int prec, pn_stream stream);

/* out[g][e] = sum_{s < per_group} slabs[g*per_group + s][e],  e < elems;  groups = n_slabs/per_group */
int pn_slab_reduce(const float* slabs, int n_slabs, int per_group, int64_t elems, float* out, pn_stream stream);

/* --- BatchNormalization statistics -> coefficients (keras BatchNormalization, PointNet.py:528,559;
 * momentum 0.99, eps 1e-3, biased variance).
 * training statistics (use_batch_stats=1): reduces stat_partials [n_tiles][2][C] over `count` rows,
 *   writes mean, invstd = rsqrt(var+eps), scale = gamma*invstd, shift = beta - mean*scale, sgn = sign(scale)
 *   and (update_moving=1) moving <- momentum*moving + (1-momentum)*batch, in place.
 * inference statistics (use_batch_stats=0; training=False or a frozen layer, PointNet.py:585-591): the same
 *   coefficients from the moving statistics. */
int pn_bn_finalize(const float* stat_partials, int n_tiles, int C, int64_t count, const float* gamma,
                   const float* beta, float* moving_mean, float* moving_var, float momentum, float eps,
                   int use_batch_stats, int update_moving, float* mean, float* invstd, float* scale, float* shift,
                   pn_stream stream);

/* backward of the same: from partial sums (sum dy_hat, sum dy_hat*z) builds dgamma, dbeta and the lazy
 * coefficients (ca, cb, cc) with dz = ca*dy_hat + cb*z + cc.  batch_stats=0 (frozen / inference BN):
 * ca = gamma*invstd, cb = cc = 0 and no dgamma/dbeta. */
int pn_bn_bwd_finalize(const float* stat_partials, int n_tiles, int C, int64_t count, const float* gamma,
                       const float* mean, const float* invstd, int batch_stats, float* dgamma, float* dbeta,
                       float* ca, float* cb, float* cc, pn_stream stream);

/* sgn[c] = +1 if gamma[c] >= 0 else -1 */
int pn_sign(const float* gamma, int C, float* sgn, pn_stream stream);

/* --- finish tf.reduce_max: reduce pmax/pidx over each cloud's tiles and apply BN+ReLU.
 *   g[b][c] = relu(scale*zstar + shift), zstar[b][c] = sgn*max, arg[b][c] = row index in the cloud */
int pn_max_finalize(const float* pmax, const int32_t* pidx, int B, int tiles_per_cloud, int C, const float* sgn,
                    const float* scale, const float* shift, float* g, float* zstar, int32_t* arg, pn_stream stream);

/* --- DenseLayer (pointnet/PointNet.py:597-679: Dense [+ BatchNormalization over the batch] [+ ReLU] [+ Dropout]) and the
 * T-Net tail X @ w + b (PointNet.py:436-442), rows = clouds.  ONE launch: split-K blocks meet in-launch and the last
 * arriver of each 32-column block applies bias / BN / ReLU / dropout.
 *   z (R, C) = x (R, K; row stride ldx) . W + bias,  W(k, j) = w[k*ldw + j]  (trans = 0)  or  w[j*ldw + k]  (trans = 1: the
 *   data gradient dx = dz . W^T straight from the layer's kernel);  a = dropout(relu(BN(z))) when a_out != NULL.
 *   bn_mode 0 none | 1 batch statistics, moving_mean/var updated in place (momentum), mean/invstd kept for the backward |
 *   2 moving statistics (frozen layer, PointNet.py:655-662).  act 0 none | 1 relu.  keep: (R, C) uint8 mask or NULL,
 *   survivors scaled by keep_scale = 1/(1-rate).
 *   workspace: pn_dense_workspace_floats(R, K, C) floats + 256 uint32 arrival counters that must be ZERO on entry (the
 *   call leaves them zero). */
size_t pn_dense_workspace_floats(int R, int K, int C);
int pn_dense_layer(const float* x, int ldx, const float* w, int ldw, int trans, int R, int K, int C, float* workspace,
                   uint32_t* counters, const float* bias, const float* gamma, const float* beta, float* moving_mean,
                   float* moving_var, float momentum, float eps, int bn_mode, int act, const uint8_t* keep, float keep_scale,
                   float* z_out, float* a_out, float* mean_out, float* invstd_out, pn_stream stream);

/* --- backward of the same layer's tail and its parameters (R <= 32): da (R, C) -> dz (R, C) through dropout, ReLU and the
 * BatchNormalization backward (batch statistics: dgamma, dbeta; none: dbias), and dw (K, C) = x^T dz.  dw may be NULL. */
int pn_dense_bwd(const float* da, const float* z, const float* x, int ldx, int R, int K, int C, const float* gamma,
                 const float* beta, const float* mean, const float* invstd, int bn_mode, int act, const uint8_t* keep,
                 float keep_scale, float* dz, float* dgamma, float* dbeta, float* dbias, float* dw, pn_stream stream);

/* --- one launch per layer of a backward CHAIN of dense layers -- the gradient of DenseLayer.call (PointNet.py:642-654) and of the
 * T-Net's X @ w + b (:436-442) as the model plan runs it; R <= 32 with a tail:
 * dx (R, C) = dz_above (R, K; row stride lddz) . W_above^T from the (C, K)-shaped kernel (w_above[j*ldw + k]) -- which is d(activation)
 * of the layer below -- and, tail != NULL, in the same launch that layer's dropout -> ReLU -> BatchNormalization backward (per
 * column, done by the workgroup that finishes the column block): tail->dz (R, C), dgamma, dbeta (bn_mode 1) or dbias (bn_mode 0).
 * Same arithmetic as pn_dense_layer(trans = 1) followed by pn_dense_bwd(dw = NULL), up to the order of the column sums.
 * workspace / counters: as pn_dense_layer. */
typedef struct pn_dense_tail {
  const float *z, *gamma, *beta, *mean, *invstd;   /* of the layer below: stored pre-BN output, BN parameters, batch mean / invstd */
  const uint8_t* keep; float keep_scale;            /* its dropout mask (R, C) or NULL */
  int bn_mode, act;                                 /* as pn_dense_layer */
  float *dz, *dgamma, *dbeta, *dbias;               /* outputs; dgamma / dbeta / dbias may be NULL */
} pn_dense_tail;
int pn_dense_bwd_step(const float* dz_above, int lddz, const float* w_above, int ldw, int R, int K, int C, float* workspace,
                      uint32_t* counters, float* dx, const pn_dense_tail* tail, pn_stream stream);
/* the weight gradients dw (K, C) = x^T . dz (x: (R, K), row stride ldx) and, db != NULL, db (C) = column sums of dz, of up to 12
 * dense layers in ONE launch (nothing reads them before the optimizer: a backward pass collects them); fp32 fma chain over the rows */
typedef struct pn_dense_wgrad_job { const float* x; int ldx; const float* dz; int R, K, C; float* dw; float* db; } pn_dense_wgrad_job;
int pn_dense_wgrad_batch(const pn_dense_wgrad_job* jobs, int n, pn_stream stream);

/* --- tf.nn.softmax (PointNet.py:134) + keras SparseCategoricalCrossentropy(from_logits=False) + sparse accuracy for rows = B
 * (pointnet_train.py:334-345): probs (R, C); with labels: loss_sum[0] = sum_r nll_r, correct[0] = #(argmax == label), and, if
 * dlogits != NULL, dlogits = grad_scale * d(sum nll)/d(logits) including keras' clip to [1e-7, 1-1e-7] (zero gradient outside). */
int pn_softmax_xent(const float* logits, int R, int C, const int32_t* labels, float grad_scale, float* probs, float* dlogits,
                    float* loss_sum, float* correct, pn_stream stream);

/* --- predicted class / part index of every row of a (R, C) probability (or logit) matrix: the FIRST maximum, as np.argmax and
 * tf.math.argmax return it (the reference's notebooks: examples/pointnet_train.ipynb:445,499, examples/pointnet_example.ipynb:2244;
 * keras' sparse_categorical_accuracy of pointnet_train.py:340-345 compares the same index with the label). */
int pn_argmax_rows(const float* values, int64_t R, int C, int32_t* index, pn_stream stream);

/* --- seg_l5_output (ConvLayer K -> Cseg <= 16 with bias, no BN; PointNet.py:141,288-290) fused with its softmax and the
 * per-point loss: probs (M, C) = softmax(x . w + bias) over M = B*N rows of a lazy operand; with labels: part[] receives
 * per-block partial (sum nll, #correct) pairs -- one block per pn_seg_out_part_rows() rows -- at stride pn_seg_out_part_stride()
 * floats; dlogits (M, C) the scaled gradient. */
int pn_seg_out_part_stride(void);
int pn_seg_out_part_rows(void);
int pn_seg_out_fwd(const pn_operand* x, const float* w, const float* bias, int64_t M, int K, int C, const int32_t* labels,
                   float grad_scale, float* probs, float* dlogits, float* part, pn_stream stream);

/* --- tf.matmul(X, R) with one K x K matrix per cloud (PointNet.py:207 K = 3, :228 K = 64): out (B*N, K) = x (B*N, K) . R[b].
 * K = 64 runs on the MFMA engine (pn_conv_fwd with a per-cloud weight stride); K = 3 is a three-FMA-per-output kernel. */
int pn_bmm(const float* x, const float* R, int B, int N, int K, float* out, int prec, pn_stream stream);

/* --- tf.debugging.check_numerics (PointNet.py:199,208,218,...,288; enabled by `debugging: true`, pointnet_train.py:112):
 * *count (device int32) += the number of NaN / Inf elements among x[0..n); x is an fp32 array, or a bf16 array when is_bf16 != 0.
 * The Python model calls it once per check site of the reference after a forward pass and raises with the reference's message for
 * the first site whose count is non-zero. */
int pn_count_nonfinite(const void* x, int64_t n, int is_bf16, int32_t* count, pn_stream stream);

/* --- farthest point sampling (no counterpart in the reference; build-defined spec):
 * per cloud, start at `start_idx`, repeatedly take the point with the largest squared distance (fp32,
 * d = dx*dx + dy*dy + dz*dz evaluated left to right without fma contraction) to the selected set, ties ->
 * lowest index.  idx_out (B, M) int32 in selection order; mindist (B, N), optional, receives the final distance of
 * every point to the selected set.  workspace: pn_fps_workspace_bytes(B, N) bytes; its first int32 is an error flag
 * (non-zero if a multi-block cloud timed out waiting for a peer block). */
size_t pn_fps_workspace_bytes(int B, int N);
int pn_fps(const float* xyz, int B, int N, int M, int start_idx, int32_t* idx_out, float* mindist, void* workspace,
           size_t workspace_bytes, pn_stream stream);

/* --- voxel-grid downsample (no counterpart in the reference; build-defined spec): key = floor((p-origin)/leaf)
 * per axis (int32, must lie in [0, 2^21)), voxels ordered by ascending (kz, ky, kx); per voxel the centroid
 * (fp64 accumulation in point-index order, rounded to fp32), the point count and the majority label (ties ->
 * lowest label).  n_out is a device int32.  workspace: pn_voxel_workspace_bytes(N). */
size_t pn_voxel_workspace_bytes(int N);
int pn_voxel_downsample(const float* xyz, const int32_t* labels, int N, const float* leaf3_host,
                        const float* origin3_host, int n_labels, float* centroids, int32_t* counts,
                        int32_t* majority, int32_t* n_out, void* workspace, size_t workspace_bytes, pn_stream stream);


/* ================================================================================================
 * Whole-model entry points: PointNet.call (pointnet/PointNet.py:197-292) forward and its backward,
 * sequenced natively on one stream.  Parameters live in ONE flat fp32 buffer (and gradients in a second
 * buffer of the same layout) so that data-parallel training all-reduces a single contiguous range.
 * ============================================================================================== */
typedef struct pn_model_desc {
  int32_t ccls;     /* classification_output_width  (PointNet.py:86)  */
  int32_t cseg;     /* segmentation_output_width    (PointNet.py:87), <= 16 */
  int32_t vanilla;  /* PointNet.py:91: no T-Nets, R = I */
  int32_t reg_in;   /* regularize_input_transform   (PointNet.py:92)  */
  int32_t reg_feat; /* regularize_feature_transform (PointNet.py:93)  */
  int32_t prec;     /* PN_PREC_* */
  float dropout_rate; /* PointNet.py:88 (0.3 in pointnet_train.py:301) */
  float bn_momentum;  /* 0.99 (PointNet.py:502) */
  float bn_eps;       /* 1e-3 (keras default)   */
  /* Synchronised BatchNormalization (data parallel, numerics-parity mode): the number of ranks whose batches form ONE batch for every
   * training-mode BatchNormalization (0 or 1: off -- each rank normalises with the statistics of its own clouds, standard DDP).
   * The reference computes the statistics over the whole batch on one device (PointNet.py:528,559,623,647); with sync_world = W a step
   * on W ranks of B clouds each is the reference's step on the B*W clouds: see pn_model_io.sync_hook.  Sizes the workspace. */
  int32_t sync_world;
} pn_model_desc;

/* one named range of the flat parameter buffer.  kind: 0 kernel, 1 gamma, 2 beta, 3 moving_mean,
 * 4 moving_var, 5 bias, 6 T-Net w, 7 T-Net b.  block: index into the 15 trainability blocks
 * (input_transform, mlp_1_1, mlp_1_2, feature_transform, mlp_2_1, mlp_2_2, mlp_2_3, mlp_cls_1..3, mlp_seg_1..5). */
typedef struct pn_slot_info {
  char name[64];
  int64_t offset; /* in floats */
  int32_t rows, cols, kind, block;
} pn_slot_info;

#define PN_NUM_BLOCKS 15

typedef struct pn_model_io {
  const float* pc; /* (B, N, 3) */
  int32_t B, N;
  float* params;            /* flat parameters (moving statistics are updated in place when training) */
  float* grads;             /* flat gradients, same layout (NULL for inference) */
  const uint8_t* trainable; /* HOST array of PN_NUM_BLOCKS flags (layer.trainable, PointNet.py:294-342); NULL = all */
  int32_t training;         /* keras `training` argument */
  /* != 0 (training, grads != NULL): pn_model_forward clears the gradient buffer in its first launch and the pn_model_backward
   * that follows (same io) does not -- one launch less per step.  0: pn_model_backward clears it itself. */
  int32_t zero_grads_in_forward;
  const uint8_t* keep1; /* dropout keep masks (B,512) / (B,256), 1 = keep; NULL = no dropout */
  const uint8_t* keep2;
  /* optional fused loss (pointnet_train.py:334-345): labels (B) / (B*N) int32, se3 target (B,3,3) */
  const int32_t* labels_cls;
  const int32_t* labels_seg;
  const float* se3;
  float loss_weights[3]; /* classification, segmentation, rotation */
  float pad2_;
  float* out_cls; /* (B, ccls) softmax  */
  float* out_seg; /* (B*N, cseg) softmax */
  float* out_R;   /* (B, 3, 3) or NULL  */
  /* 16 device floats: [0] sum of classification NLL, [1] #correct classes, [2] sum of per-point NLL,
   * [3] #correct points, [4] sum (R - se3)^2, [5] input-transform regulariser, [6] feature-transform regulariser */
  float* scalars;
  void* workspace;
  size_t workspace_bytes;
  /* optional: 6 hipEvent_t handles (HOST array), recorded on `stream` immediately before / after the three fused
   * ConvLayer(128->1024)+reduce_max launches (input_transform, feature_transform, mlp_2_3), in that order; lets a
   * benchmark time the dominant kernel inside the real step.  NULL = no events.  Do not set while capturing a graph. */
  void** prof_events;
  /* optional: a second hipStream_t.  pn_model_backward then launches the parameter-gradient kernels (nothing on the
   * data-gradient chain reads them) on it, forked from and joined back into `stream` with events -- graph edges when
   * the call is being captured.  NULL (or == stream) = everything on `stream`.  Results are bit-identical either way. */
  void* aux_stream;
  /* pn_model_backward only: 0 = the whole pass; 1 = everything down to and including the feature transform -- afterwards every
   * gradient slot from "feature_transform.conv1.kernel" (pn_model_slot_info) to the end of the buffer is final; 2 = the rest (mlp_1,
   * input transform).  Lets a data-parallel caller all-reduce the large first bucket while phase 2 runs.  1 must precede 2. */
  int32_t bwd_phase;
  /* bit 0 (PN_IO_KEEP_ACTIVATIONS): every layer leaves its stored output in the workspace.  Otherwise a segmentation head whose
   * BatchNormalization layers all use their moving statistics and through which no gradient will flow (inference; a frozen head with
   * loss weight 0 under the fused losses) runs as ONE launch that keeps its 512- / 256- / 128-wide activations on chip -- same
   * outputs bit for bit, but the workspace entries s1..s4 are not written (set the bit to inspect them, e.g. for check_numerics). */
  int32_t flags;
  /* optional (training, keep1 / keep2 given): draw the two keep masks inside pn_model_forward's first launch -- what
   * pn_dropout_masks(keep1, B*512, keep2, B*256, dropout_rate, dropout_seed, dropout_step) would write, counter increment
   * included -- instead of taking them as inputs.  dropout_step: device uint32 counter; NULL = the masks are inputs. */
  uint64_t dropout_seed;
  uint32_t* dropout_step;
  /* Synchronised BatchNormalization (pn_model_desc.sync_world = W > 1, training only).  The plan calls sync_hook wherever a quantity
   * has to be formed over all W ranks, between two of its launches, on `stream`:
   *   op 0 (all-reduce): dst[0..n) = sum over the ranks of src[0..n)        (src may equal dst)
   *   op 1 (all-gather): dst[rank r][0..n) = rank r's src[0..n), r = 0..W-1  (src may be dst + sync_rank * n)
   * dtype 0 = float32, 1 = int64.  The hook must order the collective after everything enqueued on `stream` so far and make its
   * result visible to what is enqueued next (a synchronous collective on the stream; torch.distributed does).  Returns 0 on success.
   * What is exchanged: the per-tile BatchNormalization partial sums of every per-point layer, forward and backward (summed); the
   * pooled features, the T-Nets' output gradients and the classification logits' gradients (gathered: the per-cloud dense layers then
   * run on all B*W rows on every rank, so their batch statistics are the whole batch's by construction).  Conventions the caller keeps:
   * the fused loss weights are divided by W (every rank seeds the gradient of the GLOBAL mean loss); keep1 / keep2 hold B*W rows, the
   * same on every rank; gradients are then SUMMED over the ranks with grad_scale 1, after the slots every rank computed in full (all
   * bn.gamma / bn.beta, the dense layers' kernels and bias, the T-Nets' w / b) have been zeroed on every rank but one. */
  int32_t sync_rank;
  int32_t pad3_;
  int (*sync_hook)(void* ctx, int op, const void* src, void* dst, int64_t n, int dtype, void* stream);
  void* sync_ctx;
} pn_model_io;

int pn_model_num_slots(const pn_model_desc* d);
int64_t pn_model_param_floats(const pn_model_desc* d);
int pn_model_slot_info(const pn_model_desc* d, int i, pn_slot_info* out);
size_t pn_model_workspace_bytes(const pn_model_desc* d, int B, int N, int training);
/* byte offset / size of a named intermediate inside the workspace (introspection for tests) */
int pn_model_ws_lookup(const pn_model_desc* d, int B, int N, int training, const char* name, int64_t* offset, int64_t* bytes);

/* enumerate the workspace directory: returns PN_OK and fills name/offset/bytes, or 1 when index is past the end */
int pn_model_ws_entry(const pn_model_desc* d, int B, int N, int training, int index, char* name_out, int name_cap,
                      int64_t* offset, int64_t* bytes);

int pn_model_forward(const pn_model_desc* d, const pn_model_io* io, pn_stream stream);
/* backward of the last forward on the same workspace.  d_cls / d_seg / d_R are optional upstream gradients w.r.t.
 * the three outputs; when NULL the gradients of the fused loss requested in the forward are used. */
int pn_model_backward(const pn_model_desc* d, const pn_model_io* io, const float* d_cls, const float* d_seg,
                      const float* d_R, pn_stream stream);

/* --- keras.layers.Dropout masks for the two classification-head layers (PointNet.py:252-263 via DenseLayer :652-653)
 * from a counter-based generator: keep[i] = u(seed, *step, i) >= rate, *step advanced by the call (device side, so a
 * captured hipGraph draws fresh masks at every replay).  TF's generator stream cannot be reproduced; the parity tests
 * feed masks in explicitly (pn_model_io.keep1/keep2). */
int pn_dropout_masks(uint8_t* keep1, int64_t n1, uint8_t* keep2, int64_t n2, float rate, uint64_t seed, uint32_t* step,
                     pn_stream stream);

/* keras Adam + ExponentialDecay (pointnet_train.py:310-319) over a flat range; the step counter and the step size
 * stay on the device (iterations: int32; alpha_scratch: 4 floats, initialised by pn_adam_prepare: [0] step size and [1] learning rate
 * for the current *iterations, kept current by every pn_adam_step, [2] an internal ticket counter) so the call can be replayed from a hipGraph.
 * grads are multiplied by grad_scale first (1/world_size after a sum all-reduce). */
/* evaluate the schedule for the CURRENT *iterations into alpha_scratch and clear the ticket: once after allocating the
 * state, and again whenever *iterations is set from outside (checkpoint restore) */
/* hyper-parameters travel as doubles: keras forms `1 - beta` from the Python float and only then casts to the variable's fp32
 * (1 - 0.999 -> 0.001f), whereas 1.f - 0.999f is 1.3e-5 off */
int pn_adam_prepare(const int32_t* iterations, float* alpha_scratch, double lr0, double decay_rate, double decay_steps, double beta1,
                    double beta2, pn_stream stream);
int pn_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, int32_t* iterations,
                 float* alpha_scratch, double lr0, double decay_rate, double decay_steps, double beta1, double beta2,
                 double eps, float grad_scale, pn_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* POINTNET_HIP_H */
