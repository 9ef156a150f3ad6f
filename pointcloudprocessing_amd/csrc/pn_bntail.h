// BatchNormalization statistics finished by the PRODUCING launch (training mode, batch statistics; PointNet.py:554-566).
//
// Round 1 / 2: a producer's workgroups left per-tile partial sums (sum z, sum z^2 per channel) and a finaliser launch -- ~4.8 us,
// the floor of a dependent launch on this stack -- combined them into the coefficients the consumers read.  Here every workgroup
// adds its per-channel sums to a small set of 64-bit FIXED-POINT accumulators (unit 2^-24; integer addition is associative, so
// the totals do not depend on the order the workgroups arrive in: bitwise reproducible at full parallelism), drains those atomics,
// and draws a ticket; the workgroup that draws the last one -- every other workgroup's sums are in by then -- turns the totals into
// mean / invstd / scale / shift and moves the moving statistics, exactly as bn_finalize_kernel does.  No workgroup waits for another:
// the tail is work at the end of one workgroup, not a synchronisation.  Accumulators and ticket are zero on entry (the step's first
// launch clears them, pn_prologue.hip); they are sharded BN_SHARDS ways by workgroup index to bound the contention per address.
#pragma once
#include "pn_common.h"

namespace pn {

constexpr int BN_SHARDS = 16;
constexpr double BN_FIX_FWD = 16777216.0;         // 2^24: sums of z and z^2 (up to 5e11 before the 64-bit range ends)
constexpr double BN_FIX_BWD = 1099511627776.0;    // 2^40: sums of gradients (range 8e6, resolution 1e-12)

struct BnTail {
  long long* acc;          // [BN_SHARDS][2][C]; NULL: no tail (the launch writes per-tile partials instead, if asked)
  unsigned* ticket;
  int n_wg, C;
  int kind;                // 0: forward statistics (sum z, sum z^2) -> mean, invstd, scale, shift, moving statistics
                           // 1: backward sums (sum dy, sum dy*z) -> dgamma, dbeta and the coefficients of dz = ca*dy + cb*z + cc
  double fix;              // fixed-point unit^-1 (BN_FIX_FWD / BN_FIX_BWD)
  double inv_count;
  const float* gamma; const float* beta;
  float* mm; float* mv;
  float momentum, eps;
  int update;
  float *mean, *invstd, *scale, *shift;       // kind 0: outputs; kind 1: mean, invstd are INPUTS (the forward pass's)
  float *dgamma, *dbeta, *ca, *cb, *cc;       // kind 1 (dgamma / dbeta may be NULL: frozen affine parameters)
};
static inline size_t bn_tail_words(int C) { return (size_t)BN_SHARDS * 2 * C * 2 + 64; }   // 32-bit words: accumulators + the ticket's line

// one thread per column of the workgroup's tile: its two sums
__device__ __forceinline__ void bn_tail_add(const BnTail& t, int wg, int col, float s1, float s2) {
  long long* a = t.acc + (long long)(wg & (BN_SHARDS - 1)) * 2 * t.C + col;
  __hip_atomic_fetch_add(a, __double2ll_rn((double)s1 * t.fix), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(a + t.C, __double2ll_rn((double)s2 * t.fix), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// whole workgroup (every thread calls it, nthreads = blockDim.x); flag: one word of LDS
__device__ __forceinline__ void bn_tail_meet(const BnTail& t, int tid, int nthreads, unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's atomics have been performed
  __syncthreads();
  if (tid == 0) {
    const unsigned k = __hip_atomic_fetch_add(t.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = (k == (unsigned)t.n_wg - 1u) ? 1u : 0u;
  }
  __syncthreads();
  if (*flag == 0u) return;
  for (int c = tid; c < t.C; c += nthreads) {
    long long v1[BN_SHARDS], v2[BN_SHARDS];
#pragma unroll
    for (int s = 0; s < BN_SHARDS; ++s) {                      // all in flight together
      v1[s] = __hip_atomic_load(t.acc + (long long)s * 2 * t.C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v2[s] = __hip_atomic_load(t.acc + (long long)s * 2 * t.C + t.C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    long long a1 = 0, a2 = 0;
#pragma unroll
    for (int s = 0; s < BN_SHARDS; ++s) { a1 += v1[s]; a2 += v2[s]; }
    const double ufix = 1.0 / t.fix;
    if (t.kind == 1) {      // as bn_bwd_finalize_kernel: S1 = sum dy_hat, S2 = sum dy_hat * zhat, zhat = (z - mean) * invstd
      const double S1 = (double)a1 * ufix;
      const double mean = (double)t.mean[c], is = (double)t.invstd[c];
      const float a = t.gamma[c] * t.invstd[c];
      const double S2 = ((double)a2 * ufix - mean * S1) * is;
      if (t.dgamma) t.dgamma[c] = (float)S2;
      if (t.dbeta) t.dbeta[c] = (float)S1;
      const double b = -(double)a * is * S2 * t.inv_count;
      t.ca[c] = a;
      t.cb[c] = (float)b;
      t.cc[c] = (float)(-(double)a * S1 * t.inv_count - b * mean);
      continue;
    }
    const double m = (double)a1 * ufix * t.inv_count;
    double v = (double)a2 * ufix * t.inv_count - m * m;
    if (v < 0.0) v = 0.0;
    const float mean = (float)m, var = (float)v;
    if (t.update) {
      t.mm[c] = t.mm[c] * t.momentum + mean * (1.f - t.momentum);
      t.mv[c] = t.mv[c] * t.momentum + var * (1.f - t.momentum);
    }
    const float invstd = 1.0f / sqrtf(var + t.eps);
    const float sc = t.gamma[c] * invstd;
    if (t.mean) t.mean[c] = mean;
    if (t.invstd) t.invstd[c] = invstd;
    t.scale[c] = sc;
    t.shift[c] = t.beta[c] - mean * sc;
  }
}

}  // namespace pn
