// extern "C" surface of libpointnet_hip.so: thin argument adapters over the launchers (see include/pointnet_hip.h).
#include <stdarg.h>
#include "pn_internal.h"

namespace pn {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }
}  // namespace pn

using namespace pn;
static inline hipStream_t S(pn_stream s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" {

int pn_abi_version(void) { return PN_ABI_VERSION; }
const char* pn_last_error(void) { return get_error(); }

int pn_normalize(const float* xyz, int B, int N, float* out, float* centroid, float* scale, pn_stream stream) {
  return normalize(xyz, B, N, out, centroid, scale, S(stream));
}
int pn_conv3_fwd(const float* x3, const float* w, int64_t wcs, int B, int N, int C, float* z, float* part, pn_stream stream) {
  return conv3_fwd(x3, w, wcs, B, N, C, z, part, S(stream));
}
int pn_conv3_wgrad(const float* x3, const pn_operand* dz, int B, int N, int C, float* slabs, pn_stream stream) {
  return conv3_wgrad(x3, dz, B, N, C, slabs, S(stream));
}
int pn_conv_fwd(const pn_operand* x, const float* w, int64_t wcs, int B, int N, int K, int C, const float* cloud_bias, float* z,
                float* part, int prec, pn_stream stream) {
  return conv_fwd(x, w, wcs, B, N, K, C, cloud_bias, z, part, prec, S(stream));
}
int pn_conv_fwd_max(const pn_operand* x, const float* w, int B, int N, int K, int C, const float* sgn, float* pmax, int32_t* pidx,
                    float* part, int prec, pn_stream stream) {
  return conv_fwd_max(x, w, B, N, K, C, sgn, pmax, pidx, part, prec, S(stream));
}
int pn_weights_prep(const float* w, const float* sgn, int K, int C, void* wf_hi, void* wf_lo, pn_stream stream) {
  return weights_prep(w, sgn, K, C, wf_hi, wf_lo, S(stream));
}
int pn_panel_slots_per_cloud(int B, int N) { return (B > 0 && N > 0) ? panel_slots_per_cloud(B, N) : 0; }
int pn_conv_fwd_max_panel(const pn_operand* x, const void* wf_hi, const void* wf_lo, int B, int N, int K, int C, float* pmax,
                          int32_t* pblock, float* sumsq, int64_t* colacc, int prec, pn_stream stream) {
  return conv_fwd_max_panel(x, wf_hi, wf_lo, B, N, K, C, pmax, pblock, sumsq, reinterpret_cast<long long*>(colacc), prec, S(stream));
}
int pn_panel_finalize(const float* pmax, const int32_t* pblock, const float* sumsq, const int64_t* colacc, const void* wf_hi, const void* wf_lo,
                      int prec, int B, int N, int K, int C, const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                      float momentum, float eps, int use_batch_stats, int update_moving, float* mean, float* invstd, float* scale, float* shift,
                      float* g, float* zstar, int32_t* arg_block, pn_stream stream) {
  return panel_finalize(pmax, pblock, sumsq, reinterpret_cast<const long long*>(colacc), wf_hi, wf_lo, prec, B, N, K, C, gamma, beta, moving_mean, moving_var, momentum, eps,
                        use_batch_stats, update_moving, mean, invstd, scale, shift, g, zstar, arg_block, S(stream));
}
int pn_chain_fwd_max(const pn_operand* x, const float* xyz, const float* w1, const void* w1t, const float* scale1, const float* shift1,
                     const void* w2t, const float* scale2, const float* shift2, const void* wf_hi, int B, int N, float* pmax, int32_t* pblock,
                     pn_stream stream) {
  return chain_fwd_max(x, xyz, w1, w1t, scale1, shift1, w2t, scale2, shift2, wf_hi, B, N, pmax, pblock, S(stream));
}
int pn_weights_copy16(const float* w, int K, int C, void* w16, void* wt16, pn_stream stream) { return weights_copy16(w, K, C, w16, wt16, S(stream)); }
int pn_max_resolve(const pn_operand* x, const void* wf_hi, const void* wf_lo, const int32_t* arg_block, int B, int N, int K, int C,
                   int32_t* arg, int prec, pn_stream stream) {
  return max_resolve(x, wf_hi, wf_lo, prec, arg_block, B, N, K, C, arg, S(stream));
}
int pn_maxbwd_scatter(const int32_t* arg, const float* hs, const float* wt, const float* q, int B, int N, int K, int C, float* D, int store16,
                      pn_stream stream) {
  PN_CHECK_ARG(B > 0 && N > 0 && C > 0, "pn_maxbwd_scatter: bad sizes");
  return maxbwd_scatter(arg, hs, wt, q, B, N, K, C, D, store16, S(stream));
}
int pn_conv_bwd_data(const pn_operand* dz, const float* w, int64_t wcs, int B, int N, int K, int C, const float* addend,
                     const float* zmask, const float* msc, const float* msh, float* out, float* part, int prec, pn_stream stream) {
  return conv_bwd_data(dz, w, wcs, B, N, K, C, addend, zmask, msc, msh, out, part, prec, S(stream));
}
int pn_conv_wgrad(const pn_operand* a, const pn_operand* b, int B, int N, int Ci, int Cj, int slab_rows, float* slabs, int prec,
                  pn_stream stream) {
  return conv_wgrad(a, b, B, N, Ci, Cj, slab_rows, slabs, prec, S(stream));
}
int pn_slab_reduce(const float* slabs, int n_slabs, int per_group, int64_t elems, float* out, pn_stream stream) {
  return slab_reduce(slabs, n_slabs, per_group, elems, out, S(stream));
}
int pn_bn_finalize(const float* part, int n_tiles, int C, int64_t count, const float* gamma, const float* beta, float* mm, float* mv,
                   float momentum, float eps, int use_batch_stats, int update_moving, float* mean, float* invstd, float* scale,
                   float* shift, pn_stream stream) {
  return bn_finalize(part, n_tiles, C, count, gamma, beta, mm, mv, momentum, eps, use_batch_stats, update_moving, mean, invstd, scale,
                     shift, S(stream));
}
int pn_bn_bwd_finalize(const float* part, int n_tiles, int C, int64_t count, const float* gamma, const float* mean,
                       const float* invstd, int batch_stats, float* dgamma, float* dbeta, float* ca, float* cb, float* cc,
                       pn_stream stream) {
  return bn_bwd_finalize(part, n_tiles, C, count, gamma, mean, invstd, batch_stats, dgamma, dbeta, ca, cb, cc, S(stream));
}
int pn_sign(const float* gamma, int C, float* sgn, pn_stream stream) { return sign_of(gamma, C, sgn, S(stream)); }
int pn_max_finalize(const float* pmax, const int32_t* pidx, int B, int tpc, int C, const float* sgn, const float* scale,
                    const float* shift, float* g, float* zstar, int32_t* arg, pn_stream stream) {
  return max_finalize(pmax, pidx, B, tpc, C, 0x7fffffff, sgn, scale, shift, g, zstar, arg, S(stream));
}
size_t pn_dense_workspace_floats(int R, int K, int C) { return dense_partial_floats(R, K, C); }
int pn_dense_layer(const float* x, int ldx, const float* w, int ldw, int trans, int R, int K, int C, float* workspace, uint32_t* counters,
                   const float* bias, const float* gamma, const float* beta, float* moving_mean, float* moving_var, float momentum,
                   float eps, int bn_mode, int act, const uint8_t* keep, float keep_scale, float* z_out, float* a_out, float* mean_out,
                   float* invstd_out, pn_stream stream) {
  return dense_layer(x, ldx, w, ldw, trans != 0, R, K, C, workspace, counters, bias, gamma, beta, moving_mean, moving_var, momentum, eps,
                     bn_mode, act, keep, keep_scale, z_out, a_out, mean_out, invstd_out, S(stream));
}
int pn_dense_bwd(const float* da, const float* z, const float* x, int ldx, int R, int K, int C, const float* gamma, const float* beta,
                 const float* mean, const float* invstd, int bn_mode, int act, const uint8_t* keep, float keep_scale, float* dz,
                 float* dgamma, float* dbeta, float* dbias, float* dw, pn_stream stream) {
  return dense_bwd_fused(da, z, x, ldx, R, K, C, gamma, beta, mean, invstd, bn_mode, act, keep, keep_scale, dz, dgamma, dbeta, dbias, dw,
                         S(stream));
}
int pn_dense_bwd_step(const float* dz_above, int lddz, const float* w_above, int ldw, int R, int K, int C, float* workspace,
                      uint32_t* counters, float* dx, const pn_dense_tail* tail, pn_stream stream) {
  return dense_trans_tail(dz_above, lddz, w_above, ldw, R, K, C, workspace, counters, dx, reinterpret_cast<const DenseTail*>(tail), S(stream));
}
int pn_dense_wgrad_batch(const pn_dense_wgrad_job* jobs, int n, pn_stream stream) {
  return dense_wgrad_batch(reinterpret_cast<const DenseWgradJob*>(jobs), n, S(stream));
}
int pn_softmax_xent(const float* logits, int R, int C, const int32_t* labels, float grad_scale, float* probs, float* dlogits,
                    float* loss_sum, float* correct, pn_stream stream) {
  return softmax_xent_rows(logits, R, C, labels, grad_scale, probs, dlogits, loss_sum, correct, S(stream));
}
int pn_argmax_rows(const float* values, int64_t R, int C, int32_t* index, pn_stream stream) {
  return argmax_rows(values, (long long)R, C, index, S(stream));
}
int pn_seg_out_part_stride(void) { return seg_out_part_stride(); }
int pn_seg_out_part_rows(void) { return seg_out_part_rows(); }
int pn_seg_out_fwd(const pn_operand* x, const float* w, const float* bias, int64_t M, int K, int C, const int32_t* labels, float grad_scale,
                   float* probs, float* dlogits, float* part, pn_stream stream) {
  PN_CHECK_ARG(x && x->s1 && w && probs && M > 0, "pn_seg_out_fwd: null pointer");
  return seg_out_fwd(x, w, bias, M, K, C, labels, grad_scale, probs, dlogits, part, S(stream));
}
int pn_bmm(const float* x, const float* R, int B, int N, int K, float* out, int prec, pn_stream stream) {
  PN_CHECK_ARG(x && R && out && B > 0 && N > 0, "pn_bmm: bad arguments");
  PN_CHECK_ARG(K == 3 || K == 64, "pn_bmm: K must be 3 or 64 (K=%d)", K);
  if (K == 3) return bmm3(x, R, B, N, out, S(stream));
  pn_operand o;
  memset(&o, 0, sizeof(o));
  o.s1 = x; o.ld = 64; o.lo = -INFINITY;
  return conv_fwd(&o, R, 4096, B, N, 64, 64, nullptr, out, nullptr, prec, S(stream));
}
int pn_dropout_masks(uint8_t* keep1, int64_t n1, uint8_t* keep2, int64_t n2, float rate, uint64_t seed, uint32_t* step, pn_stream stream) {
  return dropout_masks(keep1, n1, keep2, n2, rate, seed, step, S(stream));
}
int pn_count_nonfinite(const void* x, int64_t n, int is_bf16, int32_t* count, pn_stream stream) {
  return count_nonfinite(reinterpret_cast<const float*>(x), (long long)n, count, S(stream), is_bf16 ? 1 : 0);
}
size_t pn_fps_workspace_bytes(int B, int N) { return fps_workspace_bytes(B, N); }
int pn_fps(const float* xyz, int B, int N, int M, int start_idx, int32_t* idx_out, float* mindist, void* ws, size_t ws_bytes,
           pn_stream stream) {
  return fps(xyz, B, N, M, start_idx, idx_out, mindist, ws, ws_bytes, S(stream));
}
size_t pn_voxel_workspace_bytes(int N) { return voxel_workspace_bytes(N); }
int pn_voxel_downsample(const float* xyz, const int32_t* labels, int N, const float* leaf3_host, const float* origin3_host,
                        int n_labels, float* centroids, int32_t* counts, int32_t* majority, int32_t* n_out, void* ws, size_t ws_bytes,
                        pn_stream stream) {
  return voxel_downsample(xyz, labels, N, leaf3_host, origin3_host, n_labels, centroids, counts, majority, n_out, ws, ws_bytes,
                          S(stream));
}

}  // extern "C"
