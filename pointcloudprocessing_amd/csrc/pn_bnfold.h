// BatchNormalization coefficient finalisers as device bodies, and the protocol that lets the kernel CONSUMING the coefficients run
// the finaliser at its own head instead of waiting for a launch of ~5 us that does ~2 us of work:
//   * every workgroup of the consumer draws a ticket; the first cdiv(C, 16) arrivals each finalise 16 channels (they are running, so
//     they finish whatever the dispatch order or residency of the others: no placement assumption), write the coefficients with
//     write-through (sc1) stores, drain them and count themselves done;
//   * every workgroup issues its first operand loads, THEN waits for the done count (one lane polls, bounded, with s_sleep) -- the
//     wait hides under the loads' latency -- and reads the coefficients with sc1 loads (another step's values may sit in its L1 / L2);
//   * every folded launch of a step has sync words of its own, all zeroed by the step's first launch: nothing is counted or cleared on
//     the way out, and a graph replay finds the words as the first launch did.
// The standalone launches (pn_bn_finalize, pn_bn_bwd_finalize; consumers that are not row GEMMs) run the same bodies.
#pragma once
#include "pn_common.h"

namespace pn {

struct BnFin {
  int kind;                       // 0 none, 1 forward coefficients (bn_finalize), 2 backward coefficients (bn_bwd_finalize)
  int n_tiles, C, use_batch, update;
  float momentum, eps;
  const float* part;              // [n_tiles][2][C] partial sums
  double inv_count;
  const float *gamma, *beta;      // beta: forward only
  float *mm, *mv;                 // forward: moving statistics (updated when `update`)
  float *mean, *invstd;           // forward: outputs (optional);  backward: inputs
  float *scale, *shift;           // forward outputs: the consumer's ca / cc
  float *dgamma, *dbeta;          // backward outputs (optional)
  float *ca, *cb, *cc;            // backward outputs: the consumer's dz coefficients
  unsigned* sync;                 // fold only: this launch's BNFOLD_SET_WORDS sync words (see below), zero on entry
};

__device__ __forceinline__ void st_coef(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_coef(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// block = 16 channels x 16 partitions of the tile range, 16 tiles in flight per thread; fp64 combine in a fixed order
__device__ __forceinline__ void reduce_tiles_2(const float* __restrict__ part, int n_tiles, int C, int c, int ty, double (*red)[2][16],
                                               int tx, double& s1, double& s2) {
  double a = 0.0, b = 0.0;
  if (c < C) {
    const float* p = part + c;
    // sixteen tiles (32 independent loads) in flight per thread: the kernel is one memory round trip per batch, and at the usual
    // 256 tiles a thread has exactly one batch.  The summation order (tiles ascending per thread, then the 16 threads) is fixed.
    for (int t = ty; t < n_tiles; t += 16 * 16) {
      float x[16], y[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int tt = t + 16 * u;
        const long long o = (long long)(tt < n_tiles ? tt : t) * 2 * C;
        x[u] = p[o];
        y[u] = p[o + C];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (t + 16 * u < n_tiles) { a += (double)x[u]; b += (double)y[u]; }
    }
  }
  red[ty][0][tx] = a;
  red[ty][1][tx] = b;
  __syncthreads();
  s1 = 0.0; s2 = 0.0;
  for (int q = 0; q < 16; ++q) { s1 += red[q][0][tx]; s2 += red[q][1][tx]; }
}

// 16 channels [16 blk, 16 blk + 16) of finaliser f; 256 threads, every one of them returns (no early exit: the caller goes on)
__device__ __forceinline__ void bn_fin_body(const BnFin& f, int blk, double (*red)[2][16]) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blk * 16 + tx, C = f.C;
  double s1 = 0.0, s2 = 0.0;
  if (f.use_batch) reduce_tiles_2(f.part, f.n_tiles, C, c, ty, red, tx, s1, s2);
  if (ty != 0 || c >= C) return;
  if (f.kind == 1) {
    float mean, var;
    if (f.use_batch) {
      const double m = s1 * f.inv_count;
      double v = s2 * f.inv_count - m * m;
      if (v < 0.0) v = 0.0;
      mean = (float)m;
      var = (float)v;
      if (f.update) {
        f.mm[c] = f.mm[c] * f.momentum + mean * (1.f - f.momentum);
        f.mv[c] = f.mv[c] * f.momentum + var * (1.f - f.momentum);
      }
    } else {
      mean = f.mm[c];
      var = f.mv[c];
    }
    const float invstd = 1.0f / sqrtf(var + f.eps);
    const float sc = f.gamma[c] * invstd;
    if (f.mean) f.mean[c] = mean;
    if (f.invstd) f.invstd[c] = invstd;
    st_coef(f.scale + c, sc);
    st_coef(f.shift + c, f.beta[c] - mean * sc);
  } else {
    const float a = f.gamma[c] * f.invstd[c];
    if (!f.use_batch) {
      st_coef(f.ca + c, a); st_coef(f.cb + c, 0.f); st_coef(f.cc + c, 0.f);
      return;
    }
    // S1 = sum dy_hat ; S2 = sum dy_hat * zhat, zhat = (z - mean) * invstd
    const double S1 = s1;
    const double S2 = (s2 - (double)f.mean[c] * s1) * (double)f.invstd[c];
    if (f.dgamma) f.dgamma[c] = (float)S2;
    if (f.dbeta) f.dbeta[c] = (float)S1;
    const double b = -(double)a * (double)f.invstd[c] * S2 * f.inv_count;
    st_coef(f.ca + c, a);
    st_coef(f.cb + c, (float)b);
    st_coef(f.cc + c, (float)(-(double)a * S1 * f.inv_count - b * (double)f.mean[c]));
  }
}

// ---- the fold: head and wait of a consumer kernel (256-thread workgroups; scratch: 4 KB of LDS for `red`, one word for `slot`) ----
// Sync words (BNFOLD_SET_WORDS per folded launch): 8 ticket shards, one 128-byte line each -- 256 workgroups drawing from ONE word
// queue on it for ~11 ns apiece, ~3 us in all, which is what the fold is there to save -- then the done count.  Every folded launch
// of a step has a set of its own (BNFOLD_SITES of them: pn_model.hip numbers the launches of a pass) and the step's first launch
// zeroes them all, so nothing is counted or cleared on the way out of a kernel.
constexpr int BNFOLD_SHARDS = 8;
constexpr int BNFOLD_LINE = 32;                                   // words per 128-byte line
constexpr int BNFOLD_SET_WORDS = (BNFOLD_SHARDS + 1) * BNFOLD_LINE;
constexpr int BNFOLD_SITES = 48;                                  // forward 0-15, backward (phase 0 / 1) 16-31, backward phase 2 32-47

// n_blocks: workgroups of the launch.  The 16-channel groups are dealt to min(groups, n_blocks) arrivals
__device__ __forceinline__ unsigned bn_fold_roles(const BnFin& f, unsigned n_blocks) {
  const unsigned nfin = (unsigned)((f.C + 15) / 16);
  return nfin < n_blocks ? nfin : n_blocks;
}
__device__ __forceinline__ void bn_fold_head(const BnFin& f, unsigned block, unsigned n_blocks, double (*red)[2][16], unsigned* slot) {
  if (f.kind == 0) return;
  unsigned* mine = f.sync;
  // shard s serves the workgroups with block % shards == s and hands out the roles s, s + shards, ...: whoever of them arrives first
  // takes them -- no assumption about dispatch order or residency -- and each shard has at least as many workgroups as roles
  const unsigned shards = n_blocks < (unsigned)BNFOLD_SHARDS ? n_blocks : (unsigned)BNFOLD_SHARDS;
  const unsigned sh = block % shards;
  const unsigned nfin = (unsigned)((f.C + 15) / 16), roles = bn_fold_roles(f, n_blocks);
  if (threadIdx.x == 0) *slot = __hip_atomic_fetch_add(mine + sh * BNFOLD_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned t = *slot * shards + sh;
  if (t < roles) {                                           // workgroup-uniform
    for (unsigned part = t; part < nfin; part += roles) {
      bn_fin_body(f, (int)part, red);
      __syncthreads();                                       // `red` is reused by the next group
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the coefficient stores (write-through) have left this CU
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(mine + BNFOLD_SHARDS * BNFOLD_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();                                           // `red` / `slot` may be reused by the caller from here on
}
__device__ __forceinline__ void bn_fold_wait(const BnFin& f, unsigned n_blocks) {
  if (f.kind == 0) return;
  if (threadIdx.x == 0) {
    const unsigned* done = f.sync + BNFOLD_SHARDS * BNFOLD_LINE;
    const unsigned roles = bn_fold_roles(f, n_blocks);
    unsigned spins = 0;
    while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < roles) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 22)) __builtin_trap();            // ~ seconds: the finalising workgroups never depend on this one
    }
  }
  __syncthreads();
}

}  // namespace pn
