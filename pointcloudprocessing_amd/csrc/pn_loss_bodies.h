// Device bodies of the three loss reductions a fused-loss forward pass ends with; each has a launch of its own (pn_dense.hip:
// softmax_xent_rows, pn_segout.hip: sum_partials, pn_optim.hip: mse) and they share one in the model plan (pn_segout.hip: loss_tail).
// Every body is written for any workgroup size that is a multiple of 64 (softmax_xent_rows: of 1024).
#pragma once
#include "pn_common.h"

namespace pn {

__device__ __forceinline__ float grp_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
  return v;
}
__device__ __forceinline__ void softmax_xent_rows_body(const float* __restrict__ logits, int R, int C, const int* __restrict__ labels,
                                                       float grad_scale, float* __restrict__ probs, float* __restrict__ dlogits,
                                                       float* __restrict__ loss_sum, float* __restrict__ correct, float* rl, float* rc) {
  const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;      // 32 groups of 32 lanes: 32 rows per pass
  float myloss = 0.f, mycorr = 0.f;       // lane 0 of each group accumulates its rows in row order
  for (int r0 = 0; r0 < R; r0 += 32) {
    const int r = r0 + grp;
    if (r >= R) continue;                 // group-uniform (a group is half a wave; shuffles below use width 32)
    const float* l = logits + (long long)r * C;
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int c = lane; c < C; c += 32)
      if (l[c] > mx) { mx = l[c]; am = c; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {    // max with the lowest index on ties, like the serial scan
      const float om = __shfl_xor(mx, o, 32);
      const int oa = __shfl_xor(am, o, 32);
      if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    float sum = 0.f;
    for (int c = lane; c < C; c += 32) sum += expf(l[c] - mx);
    sum = grp_sum(sum);
    const float inv = 1.f / sum;
    float* p = probs + (long long)r * C;
    for (int c = lane; c < C; c += 32) p[c] = expf(l[c] - mx) * inv;
    if (labels) {
      const int y = labels[r];
      // keras: q = log(clip(p)), loss = -log_softmax(q)[y]
      float qs = 0.f;
      for (int c = lane; c < C; c += 32) qs += clip_nan(expf(l[c] - mx) * inv, 1e-7f, 1.f - 1e-7f);
      qs = grp_sum(qs);
      const float pyr = expf(l[y] - mx) * inv;
      const float py = clip_nan(pyr, 1e-7f, 1.f - 1e-7f);
      if (lane == 0) {
        myloss += -(logf(py) - logf(qs));
        mycorr += (am == y) ? 1.f : 0.f;
      }
      if (dlogits) {
        // dL/dp_i = (s_i - [i==y]) / p_i inside the clip range, 0 outside; s = clip(p)/sum clip(p)
        float dot = 0.f;
        for (int c = lane; c < C; c += 32) {
          const float pc0 = expf(l[c] - mx) * inv;
          const float pc = clip_nan(pc0, 1e-7f, 1.f - 1e-7f);
          const bool inr = (pc0 > 1e-7f) && (pc0 < 1.f - 1e-7f);
          const float dp = inr ? (pc / qs - (c == y ? 1.f : 0.f)) / pc0 : 0.f;
          dot = fmaf(pc0, dp, dot);
        }
        dot = grp_sum(dot);
        float* d = dlogits + (long long)r * C;
        for (int c = lane; c < C; c += 32) {
          const float pc0 = expf(l[c] - mx) * inv;
          const float pc = clip_nan(pc0, 1e-7f, 1.f - 1e-7f);
          const bool inr = (pc0 > 1e-7f) && (pc0 < 1.f - 1e-7f);
          const float dp = inr ? (pc / qs - (c == y ? 1.f : 0.f)) / pc0 : 0.f;
          d[c] = grad_scale * pc0 * (dp - dot);
        }
      }
    }
  }
  if (lane == 0) { rl[grp] = myloss; rc[grp] = mycorr; }
  __syncthreads();
  if (threadIdx.x == 0 && labels) {
    float x = 0.f, y = 0.f;
    for (int i = 0; i < 32; ++i) { x += rl[i]; y += rc[i]; }
    if (loss_sum) loss_sum[0] = x;
    if (correct) correct[0] = y;
  }
}


// out[e] = sum_{i<n} part[i*stride + e]      one workgroup per element: fixed assignment of rows to threads, wave shuffles, then the
// wave sums in order -> bitwise reproducible
__device__ __forceinline__ void sum_partials_body(const float* __restrict__ part, int n, int stride, int e, float* __restrict__ out,
                                                  double* wsum) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (double)part[(long long)i * stride + e];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
    out[e] = (float)t;
  }
}

// MeanSquaredError over (B,3,3) and its gradient (weight w folded in): out loss_sum = sum (R-T)^2.  Summation order: element i belongs
// to lane i % 256 of a virtual 256-lane workgroup whatever the real size (so the value does not depend on who launches it): 4 wave
// butterflies, then the four in turn
__device__ __forceinline__ void mse_body(const float* __restrict__ R, const float* __restrict__ T, int n, float gscale,
                                         float* __restrict__ dR, float* __restrict__ loss_sum, float* red) {
  float s = 0.f;
  if (threadIdx.x < 256) {
    for (int i = threadIdx.x; i < n; i += 256) {
      const float d = R[i] - T[i];
      s = fmaf(d, d, s);
      if (dR) dR[i] += gscale * d;
    }
    s = wave_sum(s);                                   // fixed order: butterfly inside the wave, then the four waves in turn
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0 && loss_sum) *loss_sum = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace pn
