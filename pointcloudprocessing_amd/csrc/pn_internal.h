// Internal launcher prototypes shared by the C-ABI wrappers (pn_api.cpp) and the model plan (pn_model.cpp).
#pragma once
#include "pn_common.h"

namespace pn {

// pn_gemm.hip
// w16 (optional; used with bf16 operands, bf16 activations and a shared kernel): a bf16 copy of the kernel laid out [C][K] with k
// contiguous -- for conv_fwd the TRANSPOSED kernel, for conv_bwd_data the kernel as it is -- staged without conversion (pn_prologue.hip)
int conv_fwd(const pn_operand* x, const float* w, long long wcs, int B, int N, int K, int C, const float* cloud_bias, float* z,
             float* stat_partials, int prec, hipStream_t st, const void* w16 = nullptr);
int conv_fwd_max(const pn_operand* x, const float* w, int B, int N, int K, int C, const float* sgn, float* pmax, int* pidx,
                 float* stat_partials, int prec, hipStream_t st);
int conv_bwd_data(const pn_operand* dz, const float* w, long long wcs, int B, int N, int K, int C, const float* addend,
                  const float* zmask, const float* msc, const float* msh, float* out, float* stat_partials, int prec,
                  hipStream_t st, const void* w16 = nullptr, const float* col_bias = nullptr);      // col_bias (C): added to every row
// weight-gradient jobs whose launches are grouped by tile shape (pn_gemm.hip: conv_wgrad_batch)
struct WgradDesc {
  pn_operand a, b;
  int B, N, Ci, Cj, slab_rows;
  float* slabs;
  int prec, colsum;
  int small_tiles;      // 64x64 output tiles whatever the channel counts: for jobs with so few slabs that 128-wide tiles leave the chip idle
};
struct DenseWgradJob;
// dense_riders: the dense layers' batched weight-gradient jobs, carried behind the first 64 x 64 split-operand batch (*rode tells whether)
int conv_wgrad_batch(const WgradDesc* jobs, int n, hipStream_t st, const DenseWgradJob* dense_riders = nullptr, int n_dense = 0, bool* rode = nullptr);
int conv_wgrad(const pn_operand* a, const pn_operand* b, int B, int N, int Ci, int Cj, int slab_rows, float* slabs, int prec,
               hipStream_t st, int colsum = 0);   // colsum: every slab is followed by Ci floats = sum over its rows of operand a

// pn_panel.hip
int weights_prep(const float* w, const float* sgn, int K, int C, void* hi, void* lo, hipStream_t st);
int panel_slots_per_cloud(int B, int N);
int conv_fwd_max_panel(const pn_operand* x, const void* wf_hi, const void* wf_lo, int B, int N, int K, int C, float* pmax, int* pq,
                       float* sumsq, long long* colacc, int prec, hipStream_t st);
int panel_finalize(const float* pmax, const int* pq, const float* sumsq, const long long* colacc, const void* wf_hi, const void* wf_lo, int prec,
                   int B, int N, int K, int C, const float* gamma, const float* beta, float* mm, float* mv, float momentum, float eps,
                   int use_batch, int update, float* mean, float* invstd, float* scale, float* shift, float* g, float* zstar, int* argq,
                   hipStream_t st, int count_mult = 1,    // count_mult: the statistics were summed over that many ranks (synchronised BN)
                   const struct WgradDesc* gram = nullptr);   // a 128-multiple, bf16, single-source weight-gradient job carried behind the finaliser

// inference: ConvLayer(3 | 64 -> 64) -> ConvLayer(64 -> 128) -> ConvLayer(128 -> 1024) -> reduce_max in one launch (pn_panel.hip: chain_max_kernel);
// exactly one of x (64-channel bf16 lazy operand, with w1t = the first kernel's transposed bf16 copy) and xyz (with w1 = the (3, 64) kernel)
int chain_fwd_max(const pn_operand* x, const float* xyz, const float* w1, const void* w1t, const float* sc1, const float* sh1, const void* w2t,
                  const float* sc2, const float* sh2, const void* wf_hi, int B, int N, float* pmax, int* pq, hipStream_t st);
// bf16 copies of a (K, C) kernel as the row GEMMs stage them (pn_prologue.hip, WCopyDesc: nat (K, C), tr (C, K)) as a launch of their own
int weights_copy16(const float* w, int K, int C, void* nat, void* tr, hipStream_t st);

// pn_prologue.hip
int normalize(const float* xyz, int B, int N, float* out, float* centroid, float* scale, hipStream_t st);
// normalisation + fragment-ordered copies of three kernels (+ zero_u cleared) + optionally the gradient buffer cleared (grads) and the
// dropout masks drawn (step): one launch
// + the BatchNormalization coefficients of up to PN_FROZEN_MAX layers that normalise with their MOVING statistics (inference, frozen
//   layers: PointNet.py:585-591) -- they depend on nothing computed in the step, so they need no launch of their own
constexpr int PN_FROZEN_MAX = 12;
struct FrozenBnDesc {
  const float *gamma, *beta, *mm, *mv;
  float *mean, *invstd, *scale, *shift;
  int C;
};
// + bf16 copies (as it is / transposed) of up to PN_WCOPY_MAX kernels (K, C) for the row GEMMs' CopyStage
constexpr int PN_WCOPY_MAX = 12;
struct WCopyDesc {
  const float* w;
  void *nat, *tr;
  int K, C;
  int frag;       // 0: tr = [C][K];  1: tr = the same values FRAGMENT-MAJOR for 32x32x16 MFMAs (the fused segmentation head): the 16 bytes
                  // of lane (h, r) of fragment (cb = c / 32, ks = k / 16) -- column c = 32 cb + r, k = 16 ks + 8 h .. + 8 -- at element
                  // ((cb * K/16 + ks) * 64 + 32 h + r) * 8;  needs C % 32 == 0, K % 16 == 0
};
int fwd_prologue(const float* xyz, int B, int N, float* out, float* centroid, float* scale, const float* const* w, const float* const* sgn,
                 const int* K, const int* C, void* const* hi, void* const* lo, unsigned* zero_u, int zero_u_n, float* grads, long long n_grads,
                 unsigned char* k1, long long n1, unsigned char* k2, long long n2, float rate, unsigned long long seed, unsigned* step,
                 const WCopyDesc* wcopies, int n_wcopies, const FrozenBnDesc* frozen, int n_frozen, float bn_eps, hipStream_t st);

// pn_pointwise.hip
// Rm: optional per-cloud 3x3 matrices folded into the (shared, wcs = 0) kernel on the fly, w_eff[b] = Rm[b] @ w, also written to
// weff_out (B, 3, C) and Rm copied to r_copy (B, 9) when given
int conv3_fwd(const float* x3, const float* w, long long wcs, int B, int N, int C, float* z, float* part, hipStream_t st, int store16 = 0,
              const float* Rm = nullptr, float* weff_out = nullptr, float* r_copy = nullptr);
int conv3_wgrad(const float* x3, const pn_operand* dz, int B, int N, int C, float* slabs, hipStream_t st);
int slab_reduce(const float* slabs, int n_slabs, int per_group, long long elems, float* out, hipStream_t st);
// several whole-range slab reductions (out_j = sum over the n_slabs_j slabs of job j, same summation order as slab_reduce) in one launch
struct SlabJob {
  const float* slabs;
  float* out;
  long long elems;
  int n_slabs;
};
int slab_reduce_batch(const SlabJob* jobs, int n_jobs, hipStream_t st);
int slab_reduce_q(const float* slabs, int n_slabs, long long elems, float* out, const float* w, const float* f, int K, int C, float* q,
                  hipStream_t st);
int bn_finalize(const float* part, int n_tiles, int C, long long count, const float* gamma, const float* beta, float* mm,
                float* mv, float momentum, float eps, int use_batch, int update, float* mean, float* invstd, float* scale,
                float* shift, hipStream_t st);
int bn_bwd_finalize(const float* part, int n_tiles, int C, long long count, const float* gamma, const float* mean,
                    const float* invstd, int batch_stats, float* dgamma, float* dbeta, float* ca, float* cb, float* cc,
                    hipStream_t st);
int sign_of(const float* gamma, int C, float* sgn, hipStream_t st);
int max_finalize(const float* pmax, const int* pidx, int B, int tpc, int C, int n_rows, const float* sgn, const float* scale,
                 const float* shift, float* g, float* zstar, int* arg, hipStream_t st);

// pn_dense.hip
constexpr int DENSE_MAX_COUNTERS = 256;     // column blocks of 32 -> C <= 8192
size_t dense_partial_floats(int R, int K, int C);
int dense_layer(const float* x, int ldx, const float* w, int ldw, bool trans, int R, int K, int C, float* partial, unsigned* counters,
                const float* bias, const float* gamma, const float* beta, float* mm, float* mv, float momentum, float eps, int bn_mode,
                int act, const unsigned char* keep, float keep_scale, float* z_out, float* a_out, float* mean_o, float* invstd_o,
                hipStream_t st);
int dense_layer_with_plain(const float* x, int ldx, const float* w, int ldw, int R, int K, int C, float* partial, unsigned* counters,
                           const float* bias, const float* gamma, const float* beta, float* mm, float* mv, float momentum, float eps,
                           int bn_mode, int act, const unsigned char* keep, float keep_scale, float* z_out, float* a_out, float* mean_o,
                           float* invstd_o, const float* w2, int ldw2, float* out2, hipStream_t st);
int dense_bwd_fused(const float* da, const float* z, const float* x, int ldx, int R, int K, int C, const float* gamma, const float* beta,
                    const float* mean, const float* invstd, int bn_mode, int act, const unsigned char* keep, float keep_scale, float* dz,
                    float* dgamma, float* dbeta, float* dbias, float* dw, hipStream_t st);
int dense_bwd_pre(const float* da, const float* z, int R, int C, const float* gamma, const float* beta, const float* mean,
                  const float* invstd, int bn_mode, int act, const unsigned char* keep, float keep_scale, float* dz, float* dgamma,
                  float* dbeta, float* dbias, hipStream_t st);
int dense_wgrad(const float* x, int ldx, const float* dz, int R, int K, int C, float* dw, hipStream_t st, float* db = nullptr);   // db: column sums of dz
// the layer below a TRANS product, taken backward in the same launch (pn_dense.hip: DenseArgs::bt_*)
struct DenseTail {                     // = pn_dense_tail of the C ABI, field for field
  const float *z, *gamma, *beta, *mean, *invstd;
  const unsigned char* keep; float keep_scale;
  int mode, act;                       // mode: 0 bias only, 1 batch statistics, 2 moving statistics; act: 1 relu
  float *dz, *dgamma, *dbeta, *dbias;  // dgamma / dbeta / dbias may be NULL (frozen layer)
};
static_assert(sizeof(DenseTail) == sizeof(pn_dense_tail), "DenseTail mirrors pn_dense_tail");
int dense_trans_tail(const float* dz, int lddz, const float* w, int ldw, int R, int K, int C, float* partial, unsigned* counters, float* dx,
                     const DenseTail* tail, hipStream_t st);
// dw (K, C) = x^T . dz, db (C) = column sums of dz (or NULL), for several layers in one launch
constexpr int DENSE_WGRAD_MAX_JOBS = 12;
struct DenseWgradJob { const float* x; int ldx; const float* dz; int R, K, C; float* dw; float* db; };      // = pn_dense_wgrad_job
static_assert(sizeof(DenseWgradJob) == sizeof(pn_dense_wgrad_job), "DenseWgradJob mirrors pn_dense_wgrad_job");
int dense_wgrad_batch(const DenseWgradJob* jobs, int n, hipStream_t st);
int transpose(const float* in, int R, int C, float* out, hipStream_t st);
int transpose2(const float* in, int R, int C, const float* rowscale, float* out, float* out2, hipStream_t st);
int softmax_xent_rows(const float* logits, int R, int C, const int* labels, float grad_scale, float* probs, float* dlogits,
                      float* loss_sum, float* correct, hipStream_t st);
int softmax_bwd_rows(const float* probs, const float* dprobs, long long R, int C, float* dlogits, hipStream_t st);
int argmax_rows(const float* values, long long R, int C, int* index, hipStream_t st);

int bmm3(const float* x, const float* R, int B, int N, float* out, hipStream_t st);

// pn_segout.hip
int seg_out_fwd(const pn_operand* x, const float* w, const float* bias, long long M, int K, int C, const int* labels,
                float grad_scale, float* probs, float* dlogits, float* part, hipStream_t st);
// the whole segmentation head (seg_l1 .. output + softmax + loss) in one launch for a head that normalises with moving statistics;
// part entries are per seg_head_fused_rows() rows (pn_segout.hip)
int seg_head_fused_rows();
int seg_head_fused(const pn_operand* x, const float* gb, const void* w1t, const void* w2t, const void* w3t, const void* w4t, const float* sc1,
                   const float* sh1, const float* sc2, const float* sh2, const float* sc3, const float* sh3, const float* sc4, const float* sh4,
                   const float* w5, const float* b5, int B, int N, int C, int s16, const int* labels, float grad_scale, float* probs,
                   float* dlogits, float* part, hipStream_t st);
int seg_out_part_stride();
int seg_out_part_rows();
int seg_out_bwd(const pn_operand* x, const float* w, const float* dlogits, int B, int N, int K, int C, float* dyhat, float* stat_part,
                float* wslab, hipStream_t st, int store16 = 0);
int sum_partials(const float* part, int n, int stride, int elems, float* out, hipStream_t st);
// softmax_xent_rows + sum_partials (n_sum elements; 0: none) + the forward value of mse (mse_out NULL: none) in one launch
int loss_tail(const float* logits, int R, int C, const int* labels, float grad_scale, float* probs, float* dlogits, float* loss_sum,
              float* correct, const float* part, int n, int stride, int n_sum, float* sum_out, const float* Rm, const float* T, int n_mse,
              float* mse_out, hipStream_t st);

// pn_maxbwd.hip
int maxbwd_prep(const float* dg, const float* dg2, const float* g, const float* zstar, int B, int C, const float* mean, const float* invstd,
                const float* scale, int batch_stats, long long count, float* hs, float* e, float* nege, float* f, float* dgamma,
                float* dbeta, const float* W, int K, float* Wt, float* We, hipStream_t st, float* pm_slabs = nullptr);
int colsum_lazy(const pn_operand* x, int B, int N, int C, float* part, hipStream_t st);
// dW of a max-pooled layer (see pn_maxbwd.hip); the layers of one pass can share a launch
struct DwJob {
  pn_operand x;
  const int* arg;
  const float* hs;
  const float* a1;
  const float* f;
  const float* e;
  const float* GW;
  float* dW;
};
int maxbwd_dw_batch(const DwJob* jobs, int n_jobs, int B, int N, int K, int C, hipStream_t st);
int maxbwd_dw(const pn_operand* x, const int* arg, const float* hs, int B, int N, int K, int C, const float* a1, const float* f,
              const float* e, const float* GW, float* dW, hipStream_t st);
int maxbwd_q(const float* w, const float* f, int K, int C, float* q, hipStream_t st);
int maxbwd_scatter(const int* arg, const float* hs, const float* wt, const float* q, int B, int N, int K, int C, float* D,
                   int store16, hipStream_t st);
int maxbwd_scatter_reduce(const int* arg, const float* hs, const float* wt, int B, int N, int K, int C, float* D, int store16,
                          const float* slabs, int n_slabs, long long elems, float* pm, const float* w, const float* f, float* q, hipStream_t st);
int maxbwd_prep_resolve(const float* dg, const float* dg2, const float* g, const float* zstar, int B, int C, const float* mean,
                        const float* invstd, const float* scale, int batch_stats, long long count, float* hs, float* e, float* nege,
                        float* f, float* dgamma, float* dbeta, const float* W, int K, float* Wt, float* We, const pn_operand* x,
                        const void* wf_hi, const void* wf_lo, int prec, const int* argq, int N, int* arg, hipStream_t st, float* pm_slabs = nullptr);
int max_resolve(const pn_operand* x, const void* wf_hi, const void* wf_lo, int prec, const int* argq, int B, int N, int K, int C, int* arg,
                hipStream_t st);

// pn_sample.hip
size_t fps_workspace_bytes(int B, int N);
int fps(const float* xyz, int B, int N, int M, int start_idx, int* idx_out, float* mindist, void* ws, size_t ws_bytes,
        hipStream_t st);
size_t voxel_workspace_bytes(int N);
int voxel_downsample(const float* xyz, const int* labels, int N, const float* leaf, const float* origin, int n_labels,
                     float* centroids, int* counts, int* majority, int* n_out, void* ws, size_t ws_bytes, hipStream_t st);

// pn_optim.hip
int adam_schedule(int* iterations, float lr0, float decay_rate, float decay_steps, float beta1, float beta2, float* alpha, float* lr,
                  hipStream_t st);
int adam(float* p, const float* g, float* m, float* v, long long n, const float* alpha, float beta1, float beta2, float eps,
         float grad_scale, hipStream_t st);
int mse(const float* R, const float* T, int n, float gscale, float* dR, float* loss_sum, hipStream_t st);
int orth_reg(const float* R, int B, int K, float c, float* dR, float* loss_part, hipStream_t st);
int fold3_fwd(const float* R, const float* W, int B, int C, float* Weff, hipStream_t st, float* R_copy = nullptr);
int fold3_bwd(const float* dWeff, const float* R, const float* W, int B, int C, float* dR, float* dW, hipStream_t st);
// the same from the per-tile slabs of conv3_wgrad (n_slabs = B * tpc, cloud-major): slab reduction and both gradients in one launch
int fold3_bwd_slabs(const float* slabs, int n_slabs, int tpc, const float* R, const float* W, int B, int C, float* dR, float* dW,
                    hipStream_t st);
int fill_eye3(float* out, int B, hipStream_t st);
int axpy(const float* x, float a, float* y, long long n, hipStream_t st);
int count_nonfinite(const float* x, long long n, int* count, hipStream_t st, int h16 = 0);
int zero_fill(float* p, long long n, hipStream_t st);
int zero_fill2(float* p, long long n, float* p2, int n2, hipStream_t st);   // + a second, small region
int add2(const float* a, const float* b, float* out, long long n, hipStream_t st);
int adam_prepare(const int* iterations, double lr0, double decay_rate, double decay_steps, double beta1, double beta2, float* scratch,
                 hipStream_t st);
int adam_fused(float* p, const float* g, float* m, float* v, long long n, int* iterations, double lr0, double decay_rate, double decay_steps,
               double beta1, double beta2, double eps, float grad_scale, float* scratch, hipStream_t st);
int dropout_masks(unsigned char* k1, long long n1, unsigned char* k2, long long n2, float rate, unsigned long long seed, unsigned* step,
                  hipStream_t st);
int cloud_bias_grad(const float* bwd_part, const float* fwd_part, int B, int tpc, int N, int C, const float* ca, const float* cb,
                    const float* cc, float* dgb, hipStream_t st);

}  // namespace pn
