// Per-cloud dense layers (rows = B clouds): DenseLayer (pointnet/PointNet.py:597-679), the T-Net tail
// X @ w + b (PointNet.py:436-442), the classification softmax + loss.  These are weight-streaming, latency
// bound problems (M = batch size): fp32 on the vector ALU, split over K for parallelism, reduced in a
// fixed order so results are bitwise reproducible.
#include "pn_common.h"

namespace pn {

constexpr int DENSE_KS = 32;   // k staged per LDS step
constexpr int DENSE_RC = 32;   // rows per register chunk
constexpr int DENSE_MAX_SPLITS = 32;
constexpr int DENSE_TB = 128;        // columns (threads) per block of the partial kernel

// k per split: a multiple of DENSE_KS, at most DENSE_MAX_SPLITS splits
static inline int dense_split_len(int K) {
  int len = cdiv(cdiv(K, DENSE_MAX_SPLITS), DENSE_KS) * DENSE_KS;
  return len < DENSE_KS ? DENSE_KS : len;
}

// partial[ks][r][j] = sum_{k in split ks} x[r][k] * w[k][j]        x: (R, K) ld = ldx ; w: (K, C)
__global__ __launch_bounds__(DENSE_TB) void dense_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                            int R, int K, int C, int split_len, float* __restrict__ partial) {
  __shared__ float xs[DENSE_KS][DENSE_RC];   // [k][r]: a row chunk's value for one k is read as a broadcast
  const int j = blockIdx.x * DENSE_TB + threadIdx.x;
  const int jc = j < C ? j : C - 1;          // clamped: weight loads are unconditional (no load under a lane-dependent branch)
  const int ks = blockIdx.y;
  const int kbeg = ks * split_len, kend = min(K, kbeg + split_len);
  for (int rc = 0; rc < R; rc += DENSE_RC) {
    const int nr = min(DENSE_RC, R - rc);
    float acc[DENSE_RC];
#pragma unroll
    for (int r = 0; r < DENSE_RC; ++r) acc[r] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += DENSE_KS) {
      const int nk = min(DENSE_KS, kend - k0);
      // this thread's weights for the step, all in flight together
      float wv[DENSE_KS];
#pragma unroll
      for (int k = 0; k < DENSE_KS; ++k) wv[k] = w[(long long)(k0 + (k < nk ? k : nk - 1)) * C + jc];
      __syncthreads();
      for (int t = threadIdx.x; t < DENSE_RC * DENSE_KS; t += DENSE_TB) {
        const int r = t / DENSE_KS, k = t % DENSE_KS;
        xs[k][r] = (r < nr && k < nk) ? x[(long long)(rc + r) * ldx + k0 + k] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < DENSE_KS; ++k) {
        const float wk = (k < nk) ? wv[k] : 0.f;
#pragma unroll
        for (int r = 0; r < DENSE_RC; ++r) acc[r] = fmaf(xs[k][r], wk, acc[r]);
      }
    }
    if (j < C)
      for (int r = 0; r < nr; ++r) partial[((long long)ks * R + rc + r) * C + j] = acc[r];
  }
}

// Finish a dense layer: z = sum_ks partial + bias; optional BatchNormalization over the R rows (batch or
// moving statistics); optional ReLU; optional inverted dropout with a given keep mask.
// block = 32 columns x 8 row partitions; all cross-row reductions in a fixed order.
//   bn_mode: 0 no BN, 1 batch statistics (+ moving update), 2 moving statistics;   act: 0 none, 1 relu
__global__ __launch_bounds__(256) void dense_finalize_kernel(const float* __restrict__ partial, int nks, int R, int C,
                                                             const float* __restrict__ bias, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ mm,
                                                             float* __restrict__ mv, float momentum, float eps, int bn_mode,
                                                             int act, const unsigned char* __restrict__ keep, float keep_scale,
                                                             float* __restrict__ z_out, float* __restrict__ a_out,
                                                             float* __restrict__ mean_o, float* __restrict__ invstd_o) {
  constexpr int RP = 16;   // row partitions
  __shared__ float red[RP][16];
  __shared__ float bc[2][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + tx;
  const bool jv = j < C;
  const float b = (jv && bias) ? bias[j] : 0.f;
  float s1 = 0.f;
  if (jv)
    for (int r = ty; r < R; r += RP) {
      float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
      int ks = 0;
      for (; ks + 3 < nks; ks += 4) {
        z0 += partial[((long long)ks * R + r) * C + j];
        z1 += partial[((long long)(ks + 1) * R + r) * C + j];
        z2 += partial[((long long)(ks + 2) * R + r) * C + j];
        z3 += partial[((long long)(ks + 3) * R + r) * C + j];
      }
      for (; ks < nks; ++ks) z0 += partial[((long long)ks * R + r) * C + j];
      const float z = ((z0 + z1) + (z2 + z3)) + b;
      z_out[(long long)r * C + j] = z;
      s1 += z;
    }
  float sc = 1.f, sh = 0.f;
  if (bn_mode) {      // block-uniform
    float mean = 0.f, var = 1.f;
    if (bn_mode == 1) {
      red[ty][tx] = s1;
      __syncthreads();
      if (ty == 0) {
        float t = 0.f;
        for (int q = 0; q < RP; ++q) t += red[q][tx];
        bc[0][tx] = t / (float)R;
      }
      __syncthreads();
      mean = bc[0][tx];
      float s2 = 0.f;
      if (jv)
        for (int r = ty; r < R; r += RP) {
          const float d = z_out[(long long)r * C + j] - mean;   // written by this thread above
          s2 = fmaf(d, d, s2);
        }
      __syncthreads();
      red[ty][tx] = s2;
      __syncthreads();
      if (ty == 0) {
        float t = 0.f;
        for (int q = 0; q < RP; ++q) t += red[q][tx];
        bc[1][tx] = t / (float)R;
      }
      __syncthreads();
      var = bc[1][tx];
      if (ty == 0 && jv) {
        mm[j] = mm[j] * momentum + mean * (1.f - momentum);
        mv[j] = mv[j] * momentum + var * (1.f - momentum);
      }
    } else if (jv) {
      mean = mm[j];
      var = mv[j];
    }
    const float invstd = 1.0f / sqrtf(var + eps);
    if (jv) {
      sc = gamma[j] * invstd;
      sh = beta[j] - mean * sc;
      if (ty == 0) {
        if (mean_o) mean_o[j] = mean;
        if (invstd_o) invstd_o[j] = invstd;
      }
    }
  }
  if (a_out && jv) {
    for (int r = ty; r < R; r += RP) {
      float y = fmaf(sc, z_out[(long long)r * C + j], sh);
      if (act == 1) y = fmaxf(y, 0.f);
      if (keep) y = keep[(long long)r * C + j] ? y * keep_scale : 0.f;
      a_out[(long long)r * C + j] = y;
    }
  }
}

// Backward through [dropout] -> [relu] -> [BN] of a dense layer:  da (R,C) -> dz (R,C), dgamma, dbeta / dbias.
// block = 32 columns x 8 row partitions.
__global__ __launch_bounds__(256) void dense_bwd_pre_kernel(const float* __restrict__ da, const float* __restrict__ z, int R, int C,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            int bn_mode, int act, const unsigned char* __restrict__ keep,
                                                            float keep_scale, float* __restrict__ dz, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dbias) {
  __shared__ float red[8][2][32];
  __shared__ float bc[2][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + tx;
  const bool jv = j < C;
  float sc = 1.f, sh = 0.f, mu = 0.f, is = 1.f;
  if (bn_mode && jv) {
    mu = mean[j]; is = invstd[j];
    sc = gamma[j] * is;
    sh = beta[j] - mu * sc;
  }
  float S1 = 0.f, S2 = 0.f;
  if (jv)
    for (int r = ty; r < R; r += 8) {
      const long long o = (long long)r * C + j;
      float d = da[o];
      if (keep) d = keep[o] ? d * keep_scale : 0.f;
      const float zz = z[o];
      if (act == 1 && !(fmaf(sc, zz, sh) > 0.f)) d = 0.f;
      dz[o] = d;   // dy_hat for now (re-read by this thread below)
      S1 += d;
      S2 = fmaf(d, (zz - mu) * is, S2);
    }
  red[ty][0][tx] = S1;
  red[ty][1][tx] = S2;
  __syncthreads();
  if (ty == 0) {
    float a = 0.f, b = 0.f;
    for (int q = 0; q < 8; ++q) { a += red[q][0][tx]; b += red[q][1][tx]; }
    bc[0][tx] = a;
    bc[1][tx] = b;
  }
  __syncthreads();
  S1 = bc[0][tx];
  S2 = bc[1][tx];
  if (!jv) return;
  if (bn_mode == 1) {
    if (ty == 0) {
      if (dgamma) dgamma[j] = S2;
      if (dbeta) dbeta[j] = S1;
    }
    const float invR = 1.f / (float)R;
    for (int r = ty; r < R; r += 8) {
      const long long o = (long long)r * C + j;
      const float zh = (z[o] - mu) * is;
      dz[o] = sc * (dz[o] - S1 * invR - zh * S2 * invR);
    }
  } else if (bn_mode == 2) {
    for (int r = ty; r < R; r += 8) dz[(long long)r * C + j] *= sc;
  } else if (ty == 0 && dbias) {
    dbias[j] = S1;
  }
}

// dw[k][j] = sum_r x[r][k] * dz[r][j]     thread <-> column j, block <-> 16 consecutive k; rows in chunks of 32 whose
// dz values are loaded up front (unconditional, clamped) so they are all in flight together
__global__ __launch_bounds__(256) void dense_wgrad_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dz, int R,
                                                          int K, int C, float* __restrict__ dw) {
  constexpr int KT = 16;
  constexpr int WRC = 32;
  __shared__ float xs[KT][WRC];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int jc = j < C ? j : C - 1;
  const int k0 = blockIdx.y * KT;
  float acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) acc[k] = 0.f;
  for (int rc = 0; rc < R; rc += WRC) {
    const int nr = min(WRC, R - rc);
    float d[WRC];
#pragma unroll
    for (int r = 0; r < WRC; ++r) d[r] = dz[(long long)(rc + min(r, nr - 1)) * C + jc];
    __syncthreads();
    for (int t = threadIdx.x; t < KT * WRC; t += 256) {
      const int k = t / WRC, r = t % WRC;
      xs[k][r] = (r < nr && k0 + k < K) ? x[(long long)(rc + r) * ldx + k0 + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < WRC; ++r) {
      const float dv = (r < nr) ? d[r] : 0.f;
#pragma unroll
      for (int k = 0; k < KT; ++k) acc[k] = fmaf(xs[k][r], dv, acc[k]);
    }
  }
  if (j < C) {
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k0 + k < K) dw[(long long)(k0 + k) * C + j] = acc[k];
  }
}

// out (C, R) = in (R, C)^T, 32x32 LDS tiles
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int R, int C, float* __restrict__ out) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (by + i < R && bx + tx < C) t[i][tx] = in[(long long)(by + i) * C + bx + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (bx + i < C && by + tx < R) out[(long long)(bx + i) * R + by + tx] = t[tx][i];
}

// out (C, R) = in (R, C)^T and out2 (C, R) = rowscale[c] * in^T (second copy scaled per OUTPUT row)
__global__ __launch_bounds__(256) void transpose2_kernel(const float* __restrict__ in, int R, int C, const float* __restrict__ rowscale,
                                                         float* __restrict__ out, float* __restrict__ out2) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (by + i < R && bx + tx < C) t[i][tx] = in[(long long)(by + i) * C + bx + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (bx + i < C && by + tx < R) {
      const float v = t[tx][i];
      out[(long long)(bx + i) * R + by + tx] = v;
      out2[(long long)(bx + i) * R + by + tx] = rowscale[bx + i] * v;
    }
}

// Row softmax + keras SparseCategoricalCrossentropy (clip 1e-7, log, sparse_softmax_xent) + its gradient
// w.r.t. the logits, one thread per row.  Used for the classification head (rows = B).
//   loss_sum[0] += sum_r nll_r ; correct[0] += #(argmax == label)      (single block => plain stores)
__global__ __launch_bounds__(256) void softmax_xent_rows_kernel(const float* __restrict__ logits, int R, int C,
                                                                const int* __restrict__ labels, float grad_scale,
                                                                float* __restrict__ probs, float* __restrict__ dlogits,
                                                                float* __restrict__ loss_sum, float* __restrict__ correct) {
  __shared__ float rl[256], rc[256];
  float myloss = 0.f, mycorr = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    const float* l = logits + (long long)r * C;
    float mx = -INFINITY;
    int am = 0;
    for (int c = 0; c < C; ++c)
      if (l[c] > mx) { mx = l[c]; am = c; }
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(l[c] - mx);
    const float inv = 1.f / sum;
    float* p = probs + (long long)r * C;
    for (int c = 0; c < C; ++c) p[c] = expf(l[c] - mx) * inv;
    if (labels) {
      const int y = labels[r];
      // keras: q = log(clip(p)), loss = -log_softmax(q)[y]
      float qs = 0.f;
      for (int c = 0; c < C; ++c) qs += fminf(fmaxf(p[c], 1e-7f), 1.f - 1e-7f);
      const float py = fminf(fmaxf(p[y], 1e-7f), 1.f - 1e-7f);
      myloss += -(logf(py) - logf(qs));
      mycorr += (am == y) ? 1.f : 0.f;
      if (dlogits) {
        // dL/dp_i = (s_i - [i==y]) / p_i inside the clip range, 0 outside; s = clip(p)/sum clip(p)
        float dot = 0.f;
        for (int c = 0; c < C; ++c) {
          const float pc = fminf(fmaxf(p[c], 1e-7f), 1.f - 1e-7f);
          const bool inr = (p[c] > 1e-7f) && (p[c] < 1.f - 1e-7f);
          const float dp = inr ? (pc / qs - (c == y ? 1.f : 0.f)) / p[c] : 0.f;
          dot = fmaf(p[c], dp, dot);
        }
        float* d = dlogits + (long long)r * C;
        for (int c = 0; c < C; ++c) {
          const float pc = fminf(fmaxf(p[c], 1e-7f), 1.f - 1e-7f);
          const bool inr = (p[c] > 1e-7f) && (p[c] < 1.f - 1e-7f);
          const float dp = inr ? (pc / qs - (c == y ? 1.f : 0.f)) / p[c] : 0.f;
          d[c] = grad_scale * p[c] * (dp - dot);
        }
      }
    }
  }
  rl[threadIdx.x] = myloss; rc[threadIdx.x] = mycorr;
  __syncthreads();
  if (threadIdx.x == 0 && labels) {
    float a = 0.f, b = 0.f;
    for (int i = 0; i < 256; ++i) { a += rl[i]; b += rc[i]; }
    if (loss_sum) loss_sum[0] = a;
    if (correct) correct[0] = b;
  }
}

// dlogits from an arbitrary upstream d(probs):  dlogit_j = p_j (dp_j - sum_i p_i dp_i)
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ probs, const float* __restrict__ dprobs,
                                                               long long R, int C, float* __restrict__ dlogits) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float* p = probs + r * C;
  const float* dp = dprobs + r * C;
  float dot = 0.f;
  for (int c = 0; c < C; ++c) dot = fmaf(p[c], dp[c], dot);
  float* d = dlogits + r * C;
  for (int c = 0; c < C; ++c) d[c] = p[c] * (dp[c] - dot);
}

// ---- host wrappers ---------------------------------------------------------------------------------------
int dense_partial(const float* x, int ldx, const float* w, int R, int K, int C, float* partial, hipStream_t st) {
  PN_CHECK_ARG(x && w && partial && R > 0 && K > 0 && C > 0, "dense_partial: bad arguments");
  const int len = dense_split_len(K);
  hipLaunchKernelGGL(dense_partial_kernel, dim3(cdiv(C, DENSE_TB), cdiv(K, len)), dim3(DENSE_TB), 0, st, x, ldx, w, R, K, C, len, partial);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
int dense_nsplit(int K) { return cdiv(K, dense_split_len(K)); }

int dense_finalize(const float* partial, int nks, int R, int C, const float* bias, const float* gamma, const float* beta, float* mm,
                   float* mv, float momentum, float eps, int bn_mode, int act, const unsigned char* keep, float keep_scale,
                   float* z_out, float* a_out, float* mean_o, float* invstd_o, hipStream_t st) {
  PN_CHECK_ARG(partial && z_out, "dense_finalize: null pointer");
  hipLaunchKernelGGL(dense_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, st, partial, nks, R, C, bias, gamma, beta, mm, mv,
                     momentum, eps, bn_mode, act, keep, keep_scale, z_out, a_out, mean_o, invstd_o);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int dense_bwd_pre(const float* da, const float* z, int R, int C, const float* gamma, const float* beta, const float* mean,
                  const float* invstd, int bn_mode, int act, const unsigned char* keep, float keep_scale, float* dz, float* dgamma,
                  float* dbeta, float* dbias, hipStream_t st) {
  PN_CHECK_ARG(da && z && dz, "dense_bwd_pre: null pointer");
  hipLaunchKernelGGL(dense_bwd_pre_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, da, z, R, C, gamma, beta, mean, invstd, bn_mode,
                     act, keep, keep_scale, dz, dgamma, dbeta, dbias);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int dense_wgrad(const float* x, int ldx, const float* dz, int R, int K, int C, float* dw, hipStream_t st) {
  PN_CHECK_ARG(x && dz && dw, "dense_wgrad: null pointer");
  hipLaunchKernelGGL(dense_wgrad_kernel, dim3(cdiv(C, 256), cdiv(K, 16)), dim3(256), 0, st, x, ldx, dz, R, K, C, dw);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int transpose(const float* in, int R, int C, float* out, hipStream_t st) {
  PN_CHECK_ARG(in && out && R > 0 && C > 0, "transpose: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, st, in, R, C, out);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int transpose2(const float* in, int R, int C, const float* rowscale, float* out, float* out2, hipStream_t st) {
  PN_CHECK_ARG(in && rowscale && out && out2 && R > 0 && C > 0, "transpose2: bad arguments");
  hipLaunchKernelGGL(transpose2_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, st, in, R, C, rowscale, out, out2);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int softmax_xent_rows(const float* logits, int R, int C, const int* labels, float grad_scale, float* probs, float* dlogits,
                      float* loss_sum, float* correct, hipStream_t st) {
  PN_CHECK_ARG(logits && probs && R > 0 && C > 0, "softmax_xent_rows: bad arguments");
  hipLaunchKernelGGL(softmax_xent_rows_kernel, dim3(1), dim3(256), 0, st, logits, R, C, labels, grad_scale, probs, dlogits,
                     loss_sum, correct);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int softmax_bwd_rows(const float* probs, const float* dprobs, long long R, int C, float* dlogits, hipStream_t st) {
  PN_CHECK_ARG(probs && dprobs && dlogits, "softmax_bwd_rows: null pointer");
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)cdivll(R, 256)), dim3(256), 0, st, probs, dprobs, R, C, dlogits);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

}  // namespace pn
