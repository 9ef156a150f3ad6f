// slab reductions shared by pn_pointwise.hip (stand-alone launches) and pn_maxbwd.hip (riding in the scatter launch)
#pragma once
#include "pn_common.h"
namespace pn {
// ------------------------------------------------------------------------------------------------------
// fixed-order slab reduction: out[g][e] = sum_s slabs[g*per_group+s][e]
// ------------------------------------------------------------------------------------------------------
// block = 32 consecutive elements x 8 partitions of the slab range; each partition is summed with 4 independent
// accumulators, partitions are combined in a fixed order -> bitwise reproducible
__device__ __forceinline__ void slab_reduce_block(const float* __restrict__ slabs, int per_group, long long elems, float* __restrict__ out,
                                                  long long bx, int grp, float (*red)[32]) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long e = bx * 32 + tx;
  float acc = 0.f;
  if (e < elems) {
    const float* s = slabs + (long long)grp * per_group * elems + e;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = ty;
    // four rounds (16 loads) in flight while that many remain: the reduction is a chain of memory round trips and nothing else; the
    // four partial sums take their slabs in the same order as the one-round loop below
    for (; i + 24 + 96 < per_group; i += 128) {
      float x[4][4];
#pragma unroll
      for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) x[it][u] = s[(long long)(i + 32 * it + 8 * u) * elems];
#pragma unroll
      for (int it = 0; it < 4; ++it) { a0 += x[it][0]; a1 += x[it][1]; a2 += x[it][2]; a3 += x[it][3]; }
    }
    for (; i + 24 < per_group; i += 32) {
      const float x0 = s[(long long)i * elems], x1 = s[(long long)(i + 8) * elems];
      const float x2 = s[(long long)(i + 16) * elems], x3 = s[(long long)(i + 24) * elems];
      a0 += x0; a1 += x1; a2 += x2; a3 += x3;
    }
    for (; i < per_group; i += 8) a0 += s[(long long)i * elems];
    acc = (a0 + a1) + (a2 + a3);
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && e < elems)
    out[(long long)grp * elems + e] = ((red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx])) + ((red[4][tx] + red[5][tx]) + (red[6][tx] + red[7][tx]));
}


// q[k] = sum_c f[c] * W[k][c], one wave per k (blocks of four k): sixteen channel groups (32 loads) in flight per lane -- at C = 1024 ONE
// memory round trip instead of sixteen; fixed order of the sum
__device__ __forceinline__ void slab_q_body(int blk, const float* __restrict__ w, const float* __restrict__ f, int K, int C, float* __restrict__ q) {
  const int k = blk * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c0 = lane; c0 < C; c0 += 64 * 16) {
    float fv[16], wv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int c = min(c0 + 64 * u, C - 1);
      fv[u] = f[c];
      wv[u] = w[(long long)k * C + c];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (c0 + 64 * u < C) s = fmaf(fv[u], wv[u], s);
  }
  s = wave_sum(s);
  if (lane == 0) q[k] = s;
}
}  // namespace pn
