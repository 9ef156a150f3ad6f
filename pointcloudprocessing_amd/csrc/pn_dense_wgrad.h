// the batched weight-gradient launch of the per-cloud dense layers: shared by pn_dense.hip (a launch of its own) and pn_gemm.hip (round 3:
// its workgroups ride behind those of the G W products of the max-pooled layers, a 96-workgroup launch that left the chip idle)
#pragma once
#include <cstring>
#include "pn_common.h"
#include "pn_internal.h"
namespace pn {
// The weight gradients dw = x^T . dz (+ db = column sums of dz) of several dense layers in ONE launch: nothing reads them before the
// optimizer, so a backward pass collects them and launches once.  Same arithmetic as dense_wgrad_kernel (fp32 fma chain over the rows).
struct DenseWgradBatch {
  DenseWgradJob job[DENSE_WGRAD_MAX_JOBS];
  int first_block[DENSE_WGRAD_MAX_JOBS + 1];      // 1-D grid: job q owns blocks [first_block[q], first_block[q + 1])
  int n;
};
// `bxi`: this block's index among the batch's blocks; xs: KT x WRC floats of LDS
__device__ __forceinline__ void dense_wgrad_batch_body(const DenseWgradBatch& b, const int bxi, float (*xs)[32]) {
  constexpr int KT = 16;
  constexpr int WRC = 32;
  int q = 0;
  while (q + 1 < b.n && bxi >= b.first_block[q + 1]) ++q;     // block-uniform
  const DenseWgradJob& jb = b.job[q];
  const int lb = bxi - b.first_block[q];
  const int ncb = (jb.C + 255) / 256;
  const int j = (lb % ncb) * 256 + threadIdx.x;
  const int jc = j < jb.C ? j : jb.C - 1;
  const int k0 = (lb / ncb) * KT;
  const int R = jb.R, K = jb.K, C = jb.C;
  float acc[KT];
  float colsum = 0.f;
#pragma unroll
  for (int k = 0; k < KT; ++k) acc[k] = 0.f;
  for (int rc = 0; rc < R; rc += WRC) {
    const int nr = min(WRC, R - rc);
    float d[WRC];
#pragma unroll
    for (int r = 0; r < WRC; ++r) d[r] = jb.dz[(long long)(rc + min(r, nr - 1)) * C + jc];
#pragma unroll
    for (int r = 0; r < WRC; ++r) colsum += (r < nr) ? d[r] : 0.f;
    __syncthreads();
    for (int t = threadIdx.x; t < KT * WRC; t += 256) {
      const int k = t / WRC, r = t % WRC;
      xs[k][r] = (r < nr && k0 + k < K) ? jb.x[(long long)(rc + r) * jb.ldx + k0 + k] : 0.f;
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
      if (k0 + k < K) jb.dw[(long long)(k0 + k) * C + j] = acc[k];
    if (jb.db && k0 == 0) jb.db[j] = colsum;
  }
}
static inline int make_dense_wgrad_batch(const DenseWgradJob* jobs, int n, DenseWgradBatch& b, int& blocks) {
  PN_CHECK_ARG(jobs && n >= 1 && n <= DENSE_WGRAD_MAX_JOBS, "dense_wgrad_batch: 1..%d jobs (n=%d)", DENSE_WGRAD_MAX_JOBS, n);
  memset(&b, 0, sizeof(b));
  b.n = n;
  blocks = 0;
  for (int q = 0; q < n; ++q) {
    PN_CHECK_ARG(jobs[q].x && jobs[q].dz && jobs[q].dw && jobs[q].R > 0 && jobs[q].K > 0 && jobs[q].C > 0, "dense_wgrad_batch: bad job %d", q);
    b.job[q] = jobs[q];
    b.first_block[q] = blocks;
    blocks += cdiv(jobs[q].C, 256) * cdiv(jobs[q].K, 16);
  }
  b.first_block[n] = blocks;
  return PN_OK;
}
}  // namespace pn
