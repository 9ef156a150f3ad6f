// Backward of "ConvLayer(K -> C) + BatchNormalization + ReLU + reduce_max over points"
// (pointnet/PointNet.py:242-248 and 425-429) WITHOUT ever forming the (B*N, C) tensors.
//
// Forward kept, per (cloud b, channel c): arg = n*(b,c) (row of the max of sgn*z), zstar = z at that row,
// g = relu(scale*zstar + shift).  With h[b,c] = dL/dg * [g > 0] the gradient w.r.t. the pre-BN z is
//     dz[m,c] = hs[b,c] * [m == n*(b,c)]  +  f_c  -  e_c * z[m,c]
//     hs = scale*h,   S1 = sum_b h,   S2 = sum_b h*zhat(zstar),   e = scale*invstd*S2/M,   f = -scale*S1/M + e*mean
// (the last two terms are BatchNormalization's batch-statistics terms; frozen BN => e = f = 0).
// Since z = A.W (A = the layer's lazy input, M x K):
//     dW = A^T dz = gather(A rows at n*) . hs  +  a1 f^T  -  (A^T A) W diag(e)          a1 = A^T 1
//     dA = dz W^T = scatter(hs W^T)            +  1 q^T   -  A (W diag(e) W^T)          q  = W f
// so only the K x K Gram matrix A^T A and two K-wide contractions are needed: 1/8 of the dense backward
// FLOPs at K = 128, C = 1024.  The Gram matrix, (A^T A) W and W diag(e) W^T run on the MFMA engine
// (pn_conv_wgrad / pn_conv_fwd / pn_conv_bwd_data); this file holds the small glue kernels.
#include "pn_common.h"

namespace pn {

// block = 32 channels x 8 partitions of the clouds: h, S1, S2 -> hs (B,C), e, f, dgamma, dbeta
__global__ __launch_bounds__(256) void maxbwd_prep_kernel(const float* __restrict__ dg, const float* __restrict__ g,
                                                          const float* __restrict__ zstar, int B, int C,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ scale, int batch_stats, double inv_count,
                                                          float* __restrict__ hs, float* __restrict__ e, float* __restrict__ nege,
                                                          float* __restrict__ f, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta) {
  __shared__ double red[8][2][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  float sc = 0.f, mu = 0.f, is = 0.f;
  double S1 = 0.0, S2 = 0.0;
  if (c < C) {
    sc = scale[c]; mu = mean[c]; is = invstd[c];
    for (int b = ty; b < B; b += 8) {
      const long long o = (long long)b * C + c;
      const float h = g[o] > 0.f ? dg[o] : 0.f;
      hs[o] = sc * h;
      S1 += (double)h;
      S2 += (double)h * (double)((zstar[o] - mu) * is);
    }
  }
  red[ty][0][tx] = S1;
  red[ty][1][tx] = S2;
  __syncthreads();
  if (ty != 0 || c >= C) return;
  S1 = 0.0; S2 = 0.0;
  for (int q = 0; q < 8; ++q) { S1 += red[q][0][tx]; S2 += red[q][1][tx]; }
  if (batch_stats) {
    if (dgamma) dgamma[c] = (float)S2;
    if (dbeta) dbeta[c] = (float)S1;
    const double ee = (double)sc * (double)is * S2 * inv_count;
    e[c] = (float)ee;
    nege[c] = (float)(-ee);
    f[c] = (float)(-(double)sc * S1 * inv_count + ee * (double)mu);
  } else {
    e[c] = 0.f; nege[c] = 0.f; f[c] = 0.f;
  }
}

// per-tile column sums of a lazy operand: part[tile][c] = sum_rows v[row][c]   (lanes <-> channel)
__global__ __launch_bounds__(256) void colsum_lazy_kernel(const pn_operand x, int N, int C, int tiles_per_cloud,
                                                          float* __restrict__ part) {
  __shared__ float red[4][64];
  const int bx = blockIdx.x, cloud = bx / tiles_per_cloud, tin = bx - cloud * tiles_per_cloud;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lane;
  const float ca = x.ca ? x.ca[c] : 1.f, cc = x.cc ? x.cc[c] : 0.f;
  const int r0 = tin * 128 + wave * 32, r1 = min(N, r0 + 32);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) {
    const long long row = (long long)cloud * N + r;
    s += fmaxf(fmaf(ca, x.s1[row * x.ld + c], cc), x.lo);
  }
  red[wave][lane] = s;
  __syncthreads();
  if (threadIdx.x < 64) part[(long long)bx * C + c] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// dW[k][c] = sum_b A[row(b,c)][k]*hs[b,c] + a1[k]*f[c] - e[c]*GW[k][c]     block per channel c, thread per k
__global__ __launch_bounds__(128) void maxbwd_dw_kernel(const pn_operand x, const int* __restrict__ arg, const float* __restrict__ hs,
                                                        int B, int N, int K, int C, const float* __restrict__ a1,
                                                        const float* __restrict__ f, const float* __restrict__ e,
                                                        const float* __restrict__ GW, float* __restrict__ dW) {
  const int c = blockIdx.x;
  const float fc = f[c], ec = e[c];
  for (int k = threadIdx.x; k < K; k += 128) {
    const float ca = x.ca ? x.ca[k] : 1.f, cc = x.cc ? x.cc[k] : 0.f;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
      const float w = hs[(long long)b * C + c];
      const long long row = (long long)b * N + arg[(long long)b * C + c];
      const float a = fmaxf(fmaf(ca, x.s1[row * x.ld + k], cc), x.lo);
      acc = fmaf(a, w, acc);
    }
    acc = fmaf(a1[k], fc, acc);
    acc = fmaf(-ec, GW[(long long)k * C + c], acc);
    dW[(long long)k * C + c] = acc;
  }
}

// q[k] = sum_c f[c] * W[k][c]        one wave per k
__global__ __launch_bounds__(64) void maxbwd_q_kernel(const float* __restrict__ w, const float* __restrict__ f, int C,
                                                      float* __restrict__ q) {
  const int k = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 64) s = fmaf(f[c], w[(long long)k * C + c], s);
  s = wave_sum(s);
  if (threadIdx.x == 0) q[k] = s;
}

// D[m][k] = q[k] + sum_{c : arg[b][c] == m} hs[b][c] * Wt[c][k]     block per 128-row tile, processed as two 64-row
// halves.  Per half the (at most C) hits are first compacted into LDS in ascending channel order (per-thread counts +
// block scan), then applied in that order => bitwise reproducible, no atomics.  K <= 128, threads <-> k.
__global__ __launch_bounds__(128) void maxbwd_scatter_kernel(const int* __restrict__ arg, const float* __restrict__ hs,
                                                             const float* __restrict__ wt, const float* __restrict__ q, int N,
                                                             int K, int C, int tiles_per_cloud, float* __restrict__ D) {
  constexpr int CHUNK = 1024;                 // channels examined per compaction round (8 per thread)
  __shared__ float tile[64][128];             // 32 KB
  __shared__ int hit_pk[CHUNK];               // (row << 16) | channel-in-chunk
  __shared__ float hit_h[CHUNK];
  __shared__ int cnt[129];
  const int bx = blockIdx.x, cloud = bx / tiles_per_cloud, tin = bx - cloud * tiles_per_cloud;
  const int t = threadIdx.x, k = t;
  const int r0 = tin * 128, nrows = min(128, N - r0);
  const float qk = (k < K) ? q[k] : 0.f;
  const int* ab = arg + (long long)cloud * C;
  const float* hb = hs + (long long)cloud * C;
  for (int half = 0; half < 2; ++half) {
    const int rbase = r0 + 64 * half;
    const int nr = min(64, nrows - 64 * half);
    if (nr <= 0) break;                        // block-uniform
    for (int r = 0; r < 64; ++r) tile[r][k] = qk;
    for (int c0 = 0; c0 < C; c0 += CHUNK) {
      int mine[8];
      int n = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = c0 + 8 * t + i;
        const int m = (c < C) ? (ab[c] - rbase) : -1;
        mine[i] = (m >= 0 && m < nr) ? m : -1;
        n += (mine[i] >= 0) ? 1 : 0;
      }
      __syncthreads();                         // previous round's hit list fully consumed
      cnt[t + 1] = n;
      if (t == 0) cnt[0] = 0;
      __syncthreads();
      if (t == 0)
        for (int i = 1; i <= 128; ++i) cnt[i] += cnt[i - 1];
      __syncthreads();
      int o = cnt[t];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (mine[i] >= 0) {
          hit_pk[o] = (mine[i] << 16) | (8 * t + i);
          hit_h[o] = hb[c0 + 8 * t + i];
          ++o;
        }
      __syncthreads();
      const int nh = cnt[128];
      if (k < K) {
        int i = 0;
        for (; i + 3 < nh; i += 4) {
          const int p0 = hit_pk[i], p1 = hit_pk[i + 1], p2 = hit_pk[i + 2], p3 = hit_pk[i + 3];
          const float w0 = wt[(long long)(c0 + (p0 & 0xffff)) * K + k], w1 = wt[(long long)(c0 + (p1 & 0xffff)) * K + k];
          const float w2 = wt[(long long)(c0 + (p2 & 0xffff)) * K + k], w3 = wt[(long long)(c0 + (p3 & 0xffff)) * K + k];
          tile[p0 >> 16][k] = fmaf(hit_h[i], w0, tile[p0 >> 16][k]);
          tile[p1 >> 16][k] = fmaf(hit_h[i + 1], w1, tile[p1 >> 16][k]);
          tile[p2 >> 16][k] = fmaf(hit_h[i + 2], w2, tile[p2 >> 16][k]);
          tile[p3 >> 16][k] = fmaf(hit_h[i + 3], w3, tile[p3 >> 16][k]);
        }
        for (; i < nh; ++i) {
          const int p0 = hit_pk[i];
          tile[p0 >> 16][k] = fmaf(hit_h[i], wt[(long long)(c0 + (p0 & 0xffff)) * K + k], tile[p0 >> 16][k]);
        }
      }
    }
    if (k < K)
      for (int r = 0; r < nr; ++r) D[((long long)cloud * N + rbase + r) * K + k] = tile[r][k];
    __syncthreads();
  }
}

int maxbwd_prep(const float* dg, const float* g, const float* zstar, int B, int C, const float* mean, const float* invstd,
                const float* scale, int batch_stats, long long count, float* hs, float* e, float* nege, float* f, float* dgamma,
                float* dbeta, hipStream_t st) {
  PN_CHECK_ARG(dg && g && zstar && mean && invstd && scale && hs && e && nege && f, "maxbwd_prep: null pointer");
  hipLaunchKernelGGL(maxbwd_prep_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, dg, g, zstar, B, C, mean, invstd, scale, batch_stats,
                     1.0 / (double)count, hs, e, nege, f, dgamma, dbeta);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int colsum_lazy(const pn_operand* x, int B, int N, int C, float* part, hipStream_t st) {
  PN_CHECK_ARG(x && x->s1 && part && C % 64 == 0, "colsum_lazy: bad arguments");
  const int tpc = cdiv(N, 128);
  hipLaunchKernelGGL(colsum_lazy_kernel, dim3(B * tpc, C / 64), dim3(256), 0, st, *x, N, C, tpc, part);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int maxbwd_dw(const pn_operand* x, const int* arg, const float* hs, int B, int N, int K, int C, const float* a1, const float* f,
              const float* e, const float* GW, float* dW, hipStream_t st) {
  PN_CHECK_ARG(x && x->s1 && arg && hs && a1 && f && e && GW && dW, "maxbwd_dw: null pointer");
  hipLaunchKernelGGL(maxbwd_dw_kernel, dim3(C), dim3(128), 0, st, *x, arg, hs, B, N, K, C, a1, f, e, GW, dW);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int maxbwd_q(const float* w, const float* f, int K, int C, float* q, hipStream_t st) {
  hipLaunchKernelGGL(maxbwd_q_kernel, dim3(K), dim3(64), 0, st, w, f, C, q);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int maxbwd_scatter(const int* arg, const float* hs, const float* wt, const float* q, int B, int N, int K, int C, float* D,
                   hipStream_t st) {
  PN_CHECK_ARG(arg && hs && wt && q && D, "maxbwd_scatter: null pointer");
  PN_CHECK_ARG(K <= 128, "maxbwd_scatter: K must be <= 128 (K=%d)", K);
  const int tpc = cdiv(N, 128);
  hipLaunchKernelGGL(maxbwd_scatter_kernel, dim3(B * tpc), dim3(128), 0, st, arg, hs, wt, q, N, K, C, tpc, D);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

}  // namespace pn
