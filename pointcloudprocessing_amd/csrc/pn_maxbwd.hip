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
#include <cstring>
#include "pn_common.h"
#include "pn_slab_reduce.h"
#include "pn_internal.h"

namespace pn {

// block = 32 channels x 8 partitions of the clouds: h, S1, S2 -> hs (B,C), e, f, dgamma, dbeta
typedef __attribute__((ext_vector_type(8))) __bf16 mb_bf16x8;
typedef __attribute__((ext_vector_type(16))) float mb_f32x16;
struct PrepArgs {
  float* pm_slabs;        // optional (K = 128): workgroup bx leaves its 32 channels' share of Pm = sum_c (-e_c) W[:,c] W[:,c]^T here
  const float *dg, *dg2, *g, *zstar;
  int B, C;
  const float *mean, *invstd, *scale;
  int batch_stats;
  double inv_count;
  float *hs, *e, *nege, *f, *dgamma, *dbeta;
  const float* W;
  int K;
  float *Wt, *We;
};
__device__ __forceinline__ void maxbwd_prep_body(const PrepArgs& a, int bx) {
  const float* __restrict__ dg = a.dg; const float* __restrict__ dg2 = a.dg2; const float* __restrict__ g = a.g;
  const float* __restrict__ zstar = a.zstar; const int B = a.B, C = a.C;
  const float* __restrict__ mean = a.mean; const float* __restrict__ invstd = a.invstd; const float* __restrict__ scale = a.scale;
  const int batch_stats = a.batch_stats; const double inv_count = a.inv_count;
  float* __restrict__ hs = a.hs; float* __restrict__ e = a.e; float* __restrict__ nege = a.nege; float* __restrict__ f = a.f;
  float* __restrict__ dgamma = a.dgamma; float* __restrict__ dbeta = a.dbeta; const float* __restrict__ W = a.W; const int K = a.K;
  float* __restrict__ Wt = a.Wt; float* __restrict__ We = a.We;
  __shared__ double red[8][2][32];
  __shared__ float neg_s[32];
  __shared__ float tt[128][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = bx * 32 + tx;
  float sc = 0.f, mu = 0.f, is = 0.f;
  double S1 = 0.0, S2 = 0.0;
  // The first 128 kernel rows of the channel-major copies below do not depend on the sums: requested first, so that the launch is
  // two memory round trips (these + the clouds' values, then the rest) instead of one per cloud group and one more for the kernel
  const int c0 = bx * 32;
  float v0[16];
  if (W) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v0[i] = W[(long long)min(ty + 8 * i, K - 1) * C + min(c0 + tx, C - 1)];
  }
  if (c < C) {
    sc = scale[c]; mu = mean[c]; is = invstd[c];
    for (int b0 = ty; b0 < B; b0 += 32) {                 // four clouds per thread in flight (unconditional, clamped loads); same order
      float up[4], gv[4], zv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long o = (long long)min(b0 + 8 * u, B - 1) * C + c;
        up[u] = (dg ? dg[o] : 0.f) + (dg2 ? dg2[o] : 0.f);   // the heads' gradients meet here (no separate add)
        gv[u] = g[o];
        zv[u] = zstar[o];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (b0 + 8 * u < B) {
          const long long o = (long long)(b0 + 8 * u) * C + c;
          const float h = gv[u] > 0.f ? up[u] : 0.f;
          hs[o] = sc * h;
          S1 += (double)h;
          S2 += (double)h * (double)((zv[u] - mu) * is);
        }
      }
    }
  }
  red[ty][0][tx] = S1;
  red[ty][1][tx] = S2;
  __syncthreads();
  if (ty == 0) {
    float ng = 0.f;
    if (c < C) {
      S1 = 0.0; S2 = 0.0;
      for (int q = 0; q < 8; ++q) { S1 += red[q][0][tx]; S2 += red[q][1][tx]; }
      if (batch_stats) {
        if (dgamma) dgamma[c] = (float)S2;
        if (dbeta) dbeta[c] = (float)S1;
        const double ee = (double)sc * (double)is * S2 * inv_count;
        e[c] = (float)ee;
        ng = (float)(-ee);
        nege[c] = ng;
        f[c] = (float)(-(double)sc * S1 * inv_count + ee * (double)mu);
      } else {
        e[c] = 0.f; nege[c] = 0.f; f[c] = 0.f;
      }
    }
    neg_s[tx] = ng;
  }
  if (!W) return;
  // channel-major copies of this block's 32 kernel columns: Wt[c][k] = W[k][c], We[c][k] = -e[c] W[k][c]
  for (int k0 = 0; k0 < K; k0 += 128) {           // 128 kernel rows per pass: 16 loads in flight per thread, one barrier pair
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = k0 + ty + 8 * i;
      v[i] = k0 == 0 ? v0[i] : W[(long long)min(k, K - 1) * C + min(c0 + tx, C - 1)];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) tt[ty + 8 * i][tx] = v[i];
    __syncthreads();
    // thread -> (channel i = tid / 8, 16 consecutive k starting at (tid % 8) * 16): 64-byte runs along k
    const int ci = threadIdx.x >> 3, kk0 = (threadIdx.x & 7) * 16;
    if (c0 + ci < C) {
      const float ng = neg_s[ci];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int k = k0 + kk0 + q;
        if (k < K) {
          const float t = tt[kk0 + q][ci];
          Wt[(long long)(c0 + ci) * K + k] = t;
          We[(long long)(c0 + ci) * K + k] = ng * t;
        }
      }
    }
  }
  // Round 3: this workgroup's share of Pm[k'][k] = sum_c (-e_c) W[k'][c] W[k][c] over its 32 channels, from the kernel block it already
  // holds in LDS: A = (-e_c W[k'][c]) (the values of We), B = W[k][c], contraction over c in two 16-wide steps, both operands split
  // into bf16 hi + lo (three products, as the weight-gradient launch that formed Pm did).  32 slabs of K x K, reduced by the launch
  // that forms q -- the launch in between (wgrad_batch<64,64,3>, 6.6-8 us at the dependent-launch floor, three per step) is gone;
  // these workgroups finished long before the row resolution's did.
  if (a.pm_slabs && K == 128) {
    __syncthreads();                              // tt: the whole block (the copy loop above only read it)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, kh = lane >> 5;
    mb_f32x16 acc[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[kb][q] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      mb_bf16x8 ah, al;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int cc = 16 * ks + 8 * kh + q;
        const float av = neg_s[cc] * tt[32 * wave + r][cc];
        ah[q] = (__bf16)av;
        al[q] = (__bf16)(av - (float)ah[q]);
      }
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        mb_bf16x8 bh, bl;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float bv = tt[32 * kb + r][16 * ks + 8 * kh + q];
          bh[q] = (__bf16)bv;
          bl[q] = (__bf16)(bv - (float)bh[q]);
        }
        acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[kb], 0, 0, 0);
        acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[kb], 0, 0, 0);
        acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[kb], 0, 0, 0);
      }
    }
    float* ps = a.pm_slabs + (long long)bx * 128 * 128;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int q = 0; q < 16; ++q) ps[(32 * wave + (q & 3) + 8 * (q >> 2) + 4 * kh) * 128 + 32 * kb + r] = acc[kb][q];
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
  act_switch(x.h16, [&](auto h) {
    constexpr bool H = decltype(h)::value;
    for (int rb = r0; rb < r1; rb += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = act_ld<H>(x.s1, ((long long)cloud * N + min(rb + u, r1 - 1)) * x.ld + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (rb + u < r1) ? clamp_lo(fmaf(ca, v[u], cc), x.lo) : 0.f;
    }
  });
  red[wave][lane] = s;
  __syncthreads();
  if (threadIdx.x < 64) part[(long long)bx * C + c] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// dW[k][c] = sum_b A[row(b,c)][k]*hs[b,c] + a1[k]*f[c] - e[c]*GW[k][c]     block per channel c, thread per k.
// The B (row, weight) pairs of the channel are staged in LDS first so the gathered rows can be loaded 8 at a time.
struct DwBatch {
  DwJob job[3];
};
__global__ __launch_bounds__(128) void maxbwd_dw_kernel(const DwBatch jb, int B, int N, int K, int C) {
  __shared__ long long srow[128];
  __shared__ float sw[128];
  const DwJob& J = jb.job[blockIdx.y];          // the max-pooled layers of one pass share a launch (grid.y = layer)
  const pn_operand& x = J.x;
  const int* __restrict__ arg = J.arg;
  const float* __restrict__ hs = J.hs;
  const int c = blockIdx.x;
  const float fc = J.f[c], ec = J.e[c];
  float acc[2] = {0.f, 0.f};        // K <= 256
  // what the result needs besides the gathered rows does not depend on them: requested first (behind the loop it was one more memory
  // round trip of a launch that is nothing but round trips)
  float a1v[2], gwv[2], cav[2], ccv[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int k = min((int)threadIdx.x + 128 * kk, K - 1);
    a1v[kk] = J.a1[k];
    gwv[kk] = J.GW[(long long)k * C + c];
    cav[kk] = x.ca ? x.ca[k] : 1.f;
    ccv[kk] = x.cc ? x.cc[k] : 0.f;
  }
  for (int b0 = 0; b0 < B; b0 += 128) {
    const int nb = min(128, B - b0);
    __syncthreads();
    if ((int)threadIdx.x < nb) {
      const int b = b0 + threadIdx.x;
      srow[threadIdx.x] = ((long long)b * N + arg[(long long)b * C + c]) * x.ld;
      sw[threadIdx.x] = hs[(long long)b * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int k = threadIdx.x + 128 * kk;
      if (k < K) {
        const float ca = cav[kk], cc = ccv[kk];
        act_switch(x.h16, [&](auto h) {
          constexpr bool H = decltype(h)::value;
          // 32 gathered rows in flight at a time (a batch of 32 clouds is ONE round trip; eight at a time were four), summed in the same order
          for (int bb = 0; bb < nb; bb += 32) {
            float v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = act_ld<H>(x.s1, srow[min(bb + u, nb - 1)] + k);
#pragma unroll
            for (int u = 0; u < 32; ++u)
              if (bb + u < nb) acc[kk] = fmaf(clamp_lo(fmaf(ca, v[u], cc), x.lo), sw[bb + u], acc[kk]);
          }
        });
      }
    }
  }
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int k = threadIdx.x + 128 * kk;
    if (k < K) {
      float a = fmaf(a1v[kk], fc, acc[kk]);
      a = fmaf(-ec, gwv[kk], a);
      J.dW[(long long)k * C + c] = a;
    }
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

// ---- the row of the maximum, found among the 32 candidates the forward pass left (pn_panel.hip) --------------------------------
// The panel kernel records, per (cloud, channel), only WHICH 32-row block of the cloud held max_n sgn*z (argq).  The row itself is
// needed by the backward pass alone and is found here: the block's 32 rows of the layer input (BN + ReLU applied, rounded to the
// MFMA operand precision exactly as the panel kernel stages them) are put in LDS once per workgroup, and for every channel whose
// maximum lies in this block a wave evaluates the 32 candidate pre-activations sgn*z = a . Wf[c] in fp32 and takes the largest,
// lowest row on ties.  Duplicated points (the reference pads clouds with duplicates, PointCloudSet.py:459-463) give bit-identical
// candidates, so the lowest index wins exactly as in the oracle; two DIFFERENT rows whose values agree to the last fp32 rounding
// may resolve to either, which leaves zstar untouched (it is the panel kernel's exact maximum) and moves the gradient between two
// rows of equal activation.
typedef __attribute__((ext_vector_type(8))) __bf16 mb_bf16x8;
typedef __attribute__((ext_vector_type(16))) float mb_f32x16;
constexpr int RS_KMAX = 128;
constexpr int RS_PITCH = RS_KMAX + 8;         // bf16 row pitch of the staged block: conflict-free 16-byte fragment reads (as pn_panel.hip)

// stage rows [rbase, rbase + nr) of cloud `cloud` (K columns) the way the panel kernel stages its panel: BN + ReLU on load, rounded
// once to bf16 (hi image) and, for bf16x3 operands, the bf16 remainder (lo image); rows outside the cloud are zero rows
template <int NT>
__device__ __forceinline__ void resolve_stage(const pn_operand& x, int cloud, int N, int K, int rbase, int nr, __bf16* __restrict__ Ab_hi,
                                              __bf16* __restrict__ Ab_lo, int tid, int nthreads) {
  for (int i = tid; i < 32 * (K / 8); i += nthreads) {
    const int row = i / (K / 8), k = (i % (K / 8)) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ca[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, cc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < nr && x.h16) {
      bf16x8_unpack(act_load8_raw(x.s1, ((long long)cloud * N + rbase + row) * x.ld + k), v);
    } else if (row < nr) {
      const float* s = x.s1 + ((long long)cloud * N + rbase + row) * x.ld + k;
      const float4 v0 = *reinterpret_cast<const float4*>(s), v1 = *reinterpret_cast<const float4*>(s + 4);
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    }
    if (x.ca) {
      const float4 t0 = *reinterpret_cast<const float4*>(x.ca + k), t1 = *reinterpret_cast<const float4*>(x.ca + k + 4);
      ca[0] = t0.x; ca[1] = t0.y; ca[2] = t0.z; ca[3] = t0.w; ca[4] = t1.x; ca[5] = t1.y; ca[6] = t1.z; ca[7] = t1.w;
    }
    if (x.cc) {
      const float4 t0 = *reinterpret_cast<const float4*>(x.cc + k), t1 = *reinterpret_cast<const float4*>(x.cc + k + 4);
      cc[0] = t0.x; cc[1] = t0.y; cc[2] = t0.z; cc[3] = t0.w; cc[4] = t1.x; cc[5] = t1.y; cc[6] = t1.z; cc[7] = t1.w;
    }
    mb_bf16x8 hv, lv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = row < nr ? clamp_lo(fmaf(ca[e], v[e], cc[e]), x.lo) : 0.f;
      hv[e] = (__bf16)t;
      if (NT == 2) lv[e] = (__bf16)(t - (float)hv[e]);
    }
    *reinterpret_cast<mb_bf16x8*>(Ab_hi + row * RS_PITCH + k) = hv;
    if (NT == 2) *reinterpret_cast<mb_bf16x8*>(Ab_lo + row * RS_PITCH + k) = lv;
  }
}
// One wave, up to 32 channels at once (lane & 31 <-> channel c, both half-waves): the 32 x 32 block of pre-activations
// sgn*z[row][c] on the matrix cores, from the same bf16 operands, in the same instruction order as the panel kernel accumulates them
// (so the values are the panel kernel's own), then per channel the largest over the valid rows, lowest row on ties.  Returns the
// row (0 .. nr-1; 0 if every candidate is NaN).
template <int NT>
__device__ __forceinline__ int resolve_group(const __bf16* __restrict__ Ab_hi, const __bf16* __restrict__ Ab_lo, const __bf16* __restrict__ wf_hi,
                                             const __bf16* __restrict__ wf_lo, int c, int K, int nr, int lane) {
  const int r = lane & 31, h = lane >> 5, KS = K / 16;
  const int cb = c >> 5, cl = c & 31;
  mb_f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // every weight fragment of the group is requested before the first MFMA: the gathered 16-byte loads are one L2 round trip in all,
  // not one per k-step (K <= 128: at most 8 k-steps)
  constexpr int KSM = RS_KMAX / 16;
  mb_bf16x8 bh[KSM], bl[NT == 2 ? KSM : 1];
#pragma unroll
  for (int ks = 0; ks < KSM; ++ks) {
    const long long chunk = ((long long)cb * KS + (ks < KS ? ks : 0)) * 64 + h * 32 + cl;
    bh[ks] = *reinterpret_cast<const mb_bf16x8*>(wf_hi + chunk * 8);
    if (NT == 2) bl[ks] = *reinterpret_cast<const mb_bf16x8*>(wf_lo + chunk * 8);
  }
#pragma unroll
  for (int ks = 0; ks < KSM; ++ks) {
    if (ks < KS) {
      const mb_bf16x8 ah = *reinterpret_cast<const mb_bf16x8*>(Ab_hi + r * RS_PITCH + ks * 16 + h * 8);
      if (NT == 2) {
        const mb_bf16x8 al = *reinterpret_cast<const mb_bf16x8*>(Ab_lo + r * RS_PITCH + ks * 16 + h * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ks], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ks], acc, 0, 0, 0);
    }
  }
  if (NT == 2) {
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks) {
      if (ks < KS) {
        const mb_bf16x8 ah = *reinterpret_cast<const mb_bf16x8*>(Ab_hi + r * RS_PITCH + ks * 16 + h * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ks], acc, 0, 0, 0);
      }
    }
  }
  float best = -INFINITY;
  int bi = 0x7fffffff;
#pragma unroll
  for (int e = 0; e < 16; ++e) {              // rows ascend with e: the first maximum wins
    const int il = (e & 3) + 8 * (e >> 2) + 4 * h;
    const float v = il < nr ? acc[e] : -INFINITY;
    const bool better = v > best;
    best = better ? v : best;
    bi = better ? il : bi;
  }
  const float ob = __shfl_xor(best, 32, 64);
  const int oi = __shfl_xor(bi, 32, 64);
  const bool take = ob > best || (ob == best && oi < bi);
  const int row = take ? oi : bi;
  return (row >= 0 && row < nr) ? row : 0;
}

// one workgroup per 32-row block of a cloud: the rows of every channel whose maximum the forward pass located in this block
template <int NT>
__device__ __forceinline__ void max_resolve_body(const pn_operand& x, const __bf16* __restrict__ wf_hi, const __bf16* __restrict__ wf_lo,
                                                 const int* __restrict__ argq, int N, int K, int C, int quarters_per_cloud,
                                                 int* __restrict__ arg, int bx) {
  __shared__ __attribute__((aligned(16))) __bf16 Ab_hi[32 * RS_PITCH];
  __shared__ __attribute__((aligned(16))) __bf16 Ab_lo[NT == 2 ? 32 * RS_PITCH : 8];
  __shared__ int hit_c[1024];
  __shared__ int nhit;
  const int cloud = bx / quarters_per_cloud, qin = bx - cloud * quarters_per_cloud;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int rbase = qin * 32, nr = min(32, N - rbase);
  bool staged = false;
  for (int c0 = 0; c0 < C; c0 += 1024) {
    if (t == 0) nhit = 0;
    __syncthreads();
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + 4 * t + i;
      if (c < C && argq[(long long)cloud * C + c] == qin) hit_c[atomicAdd(&nhit, 1)] = c;
    }
    __syncthreads();
    const int total = nhit;
    if (total > 0 && !staged) {
      resolve_stage<NT>(x, cloud, N, K, rbase, nr, Ab_hi, Ab_lo, t, 256);
      staged = true;
      __syncthreads();
    }
    for (int g0 = wave * 32; g0 < total; g0 += 4 * 32) {          // wave-uniform
      const int i = g0 + (lane & 31);
      const int c = hit_c[min(i, total - 1)];
      const int row = resolve_group<NT>(Ab_hi, Ab_lo, wf_hi, wf_lo, c, K, nr, lane);
      if (lane < 32 && i < total) arg[(long long)cloud * C + c] = rbase + row;
    }
    __syncthreads();
  }
}
// standalone form: the op-level API pn_max_resolve
template <int NT>
__global__ __launch_bounds__(256) void max_resolve_kernel(const pn_operand x, const __bf16* __restrict__ wf_hi, const __bf16* __restrict__ wf_lo,
                                                          const int* __restrict__ argq, int N, int K, int C, int quarters_per_cloud,
                                                          int* __restrict__ arg) {
  max_resolve_body<NT>(x, wf_hi, wf_lo, argq, N, K, C, quarters_per_cloud, arg, blockIdx.x);
}
// the form the model plan uses: ONE launch for the preparation of the Gram-form backward (C / 32 workgroups: few, latency bound)
// and the row resolution (one workgroup per 32-row block: many, independent of the former) -- the second rides in the first's shadow
template <int NT>
__global__ __launch_bounds__(256) void maxbwd_prep_resolve_kernel(const PrepArgs pa, int n_prep, const pn_operand x,
                                                                  const __bf16* __restrict__ wf_hi, const __bf16* __restrict__ wf_lo,
                                                                  const int* __restrict__ argq, int N, int K, int C,
                                                                  int quarters_per_cloud, int* __restrict__ arg) {
  if ((int)blockIdx.x < n_prep) maxbwd_prep_body(pa, blockIdx.x);
  else max_resolve_body<NT>(x, wf_hi, wf_lo, argq, N, K, C, quarters_per_cloud, arg, (int)blockIdx.x - n_prep);
}
__global__ __launch_bounds__(256) void maxbwd_prep_kernel(const PrepArgs pa) { maxbwd_prep_body(pa, blockIdx.x); }

// D[m][k] = q[k] + sum_{c : arg[b][c] == m} hs[b][c] * Wt[c][k]        one block per 32-row quarter tile.
// Critical points are few: the 1024 arg-max rows of a cloud concentrate on a handful of points (bench.py's clouds: ~290 distinct rows,
// the heaviest row 140 hits, the heaviest 32-row tile 170-280, the median tile 22 -- tools/scatter_probe.py), so the launch lasts as
// long as its heaviest tile.  Round 3 form: the tile's sum is a small matrix product on the matrix cores,
//     D_tile (32 x K) = S (32 x H) . Wg (H x K),   S[m][j] = hs of hit j if its row is m, else 0;   Wg[j] = row c_j of Wt,
// H = the tile's hits, compacted IN CHANNEL ORDER (thread t examines channels t, t + 256, ...: one ballot per pass places every hit, 16
// counters order the passes and waves).  Wave w owns columns 32 w .. + 31: a lane's B fragment of a 16-hit step is eight 4-byte
// loads, each coalesced over the 32 columns (one row of Wt per half-wave), four steps' loads in flight at a time.  Both operands are split bf16 hi + lo (three products: fp32-grade, as everywhere), the k order is fixed:
// bitwise reproducible with no atomics.  The round-1 form accumulated 2^-40 fixed-point int64 with LDS atomics (fp64 arithmetic to
// enter and leave fixed point, ~560 same-address 64-bit atomics per wave in the heaviest tile).
// Measured (C2, us per launch): round 1 form 15.9; one owner thread per element walking the list 34.5 (the heaviest row is a serial
// chain); this form 14.2; with two steps per group and the next group's loads requested ahead 16.9; eight waves splitting the hit
// list (166 registers: one workgroup per CU) 21.8.  Ablations of this form (PN_SCATTER_DBG): launch + workgroup scheduling 4.6, the
// list 0.2, the stores 3-4.6, the hits 6.5-9 -- the heaviest tile's chain of dependent groups, with the other tiles long finished.
// K <= 128, K % 32 == 0.
// RB: 32-row blocks per tile (1: 32-row tiles; 2: 64-row tiles for clouds of 2048 points and more -- a cloud has C hits whatever its size, and a
// workgroup costs a fixed chain of round trips: half the workgroups, the same hits)
template <int RB>
__device__ __forceinline__ void maxbwd_scatter_body(const int bx, const int* __restrict__ arg, const float* __restrict__ hs,
                                                    const float* __restrict__ wt, const float* __restrict__ q, int N,
                                                    int K, int C, int quarters_per_cloud, float* __restrict__ D, int store16, int dbg) {
  constexpr int CHUNK = 1024;                  // channels examined per round (4 per thread)
  constexpr int PAD = 64;                      // the list is padded with hits of row -1 (match nothing) to whole groups
  constexpr int GS = 4;                        // 16-hit steps per group: 32 loads per lane in flight
  __shared__ __attribute__((aligned(16))) int s_out[32 * RB * 128]; // the finished tile (fp32 or bf16), then:
  int* s_row = s_out;                                               // the hit list lives in the same 16 KB while it is needed
  int* s_c = s_out + (CHUNK + PAD);
  float* s_h = reinterpret_cast<float*>(s_out + 2 * (CHUNK + PAD));
  static_assert(3 * (CHUNK + PAD) <= 32 * RB * 128, "the hit list fits the output tile's LDS");
  __shared__ int cnt[16];                      // hits of (pass, wave)
  const int cloud = bx / quarters_per_cloud, qin = bx - cloud * quarters_per_cloud;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m = lane & 31, kh = lane >> 5;
  const int rbase = qin * 32 * RB, nr = min(32 * RB, N - rbase);
  const int* ab = arg + (long long)cloud * C;
  const float* hb = hs + (long long)cloud * C;
  const bool wave_on = 32 * wave < K;          // this wave's 32 columns exist
  const int kcol = wave_on ? 32 * wave + m : m;
  const float qk = q ? q[kcol] : 0.f;          // requested with the first rows of the maxima (null: the consumer adds q itself)
  mb_f32x16 acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[rb][e] = 0.f;
  for (int c0 = 0; c0 < C; c0 += CHUNK) {
    int mrow[4];
    float hv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + 256 * i + t;
      mrow[i] = (c < C && !(dbg & 4)) ? ab[c] - rbase : -1;
      hv[i] = (c < C && !(dbg & 4)) ? hb[c] : 0.f;
    }
    unsigned long long bal[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bal[i] = __ballot(mrow[i] >= 0 && mrow[i] < nr);
      if (lane == 0) cnt[4 * i + wave] = __popcll(bal[i]);
    }
    __syncthreads();
    int cn[16], total = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { cn[j] = cnt[j]; total += cn[j]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int before = 0;                          // a pass-i hit of this wave goes behind every (pass, wave) pair that precedes (i, wave)
#pragma unroll
      for (int j = 0; j < 16; ++j) before += (j < 4 * i + wave) ? cn[j] : 0;
      if (mrow[i] >= 0 && mrow[i] < nr) {
        const int at = before + __popcll(bal[i] & ((1ull << lane) - 1ull));
        s_row[at] = mrow[i]; s_c[at] = 256 * i + t; s_h[at] = hv[i];
      }
    }
    if (t < PAD) { s_row[total + t] = -1; s_c[total + t] = 0; s_h[total + t] = 0.f; }
    __syncthreads();
    if (wave_on && total > 0 && !(dbg & 1)) {
      auto load_group = [&](int j0, float (&bv)[GS][8]) {          // hits j0 + 16 s + 8 kh + e (past the list: padding)
#pragma unroll
        for (int s2 = 0; s2 < GS; ++s2) {
          const int jb = j0 + 16 * s2 + 8 * kh;
          const int4 c0v = *reinterpret_cast<const int4*>(s_c + jb), c1v = *reinterpret_cast<const int4*>(s_c + jb + 4);
          const int cj[8] = {c0v.x, c0v.y, c0v.z, c0v.w, c1v.x, c1v.y, c1v.z, c1v.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) bv[s2][e] = wt[(long long)(c0 + cj[e]) * K + kcol];
        }
      };
      auto compute_group = [&](int j0, const float (&bv)[GS][8]) {
#pragma unroll
        for (int s2 = 0; s2 < GS; ++s2) {
          if (j0 + 16 * s2 < total) {            // wave-uniform
            const int jb = j0 + 16 * s2 + 8 * kh;
            const int4 r0v = *reinterpret_cast<const int4*>(s_row + jb), r1v = *reinterpret_cast<const int4*>(s_row + jb + 4);
            const float4 h0v = *reinterpret_cast<const float4*>(s_h + jb), h1v = *reinterpret_cast<const float4*>(s_h + jb + 4);
            const int rj[8] = {r0v.x, r0v.y, r0v.z, r0v.w, r1v.x, r1v.y, r1v.z, r1v.w};
            const float hj[8] = {h0v.x, h0v.y, h0v.z, h0v.w, h1v.x, h1v.y, h1v.z, h1v.w};
            mb_bf16x8 bh, bl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              bh[e] = (__bf16)bv[s2][e];
              bl[e] = (__bf16)(bv[s2][e] - (float)bh[e]);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
              mb_bf16x8 ah, al;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float av = rj[e] == 32 * rb + m ? hj[e] : 0.f;
                ah[e] = (__bf16)av;
                al[e] = (__bf16)(av - (float)ah[e]);
              }
              acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[rb], 0, 0, 0);
              acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[rb], 0, 0, 0);
              acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[rb], 0, 0, 0);
            }
          }
        }
      };
      constexpr int GH = 16 * GS;                // hits per group
      for (int j0 = 0; j0 < total; j0 += GH) {
        float bv[GS][8];
        load_group(j0, bv);
        compute_group(j0, bv);
      }
    }
    __syncthreads();                           // the list and the counters are rewritten by the next chunk
  }
  // the tile leaves through LDS (over the hit list) so that every lane stores 16 contiguous bytes and a wave whole rows: the
  // accumulator layout (a lane = one column, two rows per instruction) gave 2-byte stores, 4.6 us of the launch at B*N = 32,768
  __syncthreads();
  float* ot = reinterpret_cast<float*>(s_out);
  if (wave_on) {
    act_switch(store16, [&](auto h) {
      constexpr bool H = decltype(h)::value;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = 32 * rb + (e & 3) + 8 * (e >> 2) + 4 * kh;
          act_st<H>(ot, (long long)r * K + kcol, acc[rb][e] + qk);
        }
    });
  }
  __syncthreads();
  if (!(dbg & 2)) {
    const int es = store16 ? 2 : 4;                                   // bytes per element
    const int row_bytes = K * es, n16 = nr * row_bytes / 16;          // K % 32 == 0: whole 16-byte pieces
    unsigned char* dst = reinterpret_cast<unsigned char*>(D) + ((long long)cloud * N + rbase) * row_bytes;
    for (int i = t; i < n16; i += 256) *reinterpret_cast<uint4*>(dst + 16ll * i) = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(ot) + 16 * i);
  }
}

template <int RB>
__global__ __launch_bounds__(256) void maxbwd_scatter_kernel(const int* __restrict__ arg, const float* __restrict__ hs,
                                                             const float* __restrict__ wt, const float* __restrict__ q, int N,
                                                             int K, int C, int quarters_per_cloud, float* __restrict__ D, int store16, int dbg) {
  maxbwd_scatter_body<RB>(blockIdx.x, arg, hs, wt, q, N, K, C, quarters_per_cloud, D, store16, dbg);
}
// Round 3: the scatter (without q: the data-gradient GEMM that consumes D adds it as a per-column constant) no longer depends on the
// launch that reduces the Pm slabs and forms q = W f -- so that launch's workgroups ride BEHIND the scatter's (their block ids follow:
// the heaviest tiles start at once, the reductions fill the CUs the light tiles leave): one dependent launch less per max-pooled layer
template <int RB>
__global__ __launch_bounds__(256) void maxbwd_scatter_reduce_kernel(const int* __restrict__ arg, const float* __restrict__ hs,
                                                                    const float* __restrict__ wt, int N, int K, int C, int quarters_per_cloud,
                                                                    int n_scatter, float* __restrict__ D, int store16, int dbg,
                                                                    const float* __restrict__ slabs, int n_slabs, long long elems,
                                                                    float* __restrict__ pm, int nb_reduce, const float* __restrict__ w,
                                                                    const float* __restrict__ f, float* __restrict__ q) {
  if ((int)blockIdx.x < n_scatter) {
    maxbwd_scatter_body<RB>(blockIdx.x, arg, hs, wt, nullptr, N, K, C, quarters_per_cloud, D, store16, dbg);
    return;
  }
  __shared__ float red[8][32];
  const int bx = (int)blockIdx.x - n_scatter;
  if (bx < nb_reduce) slab_reduce_block(slabs, n_slabs, elems, pm, bx, 0, red);
  else slab_q_body(bx - nb_reduce, w, f, K, C, q);
}

static PrepArgs make_prep(const float* dg, const float* dg2, const float* g, const float* zstar, int B, int C, const float* mean,
                          const float* invstd, const float* scale, int batch_stats, long long count, float* hs, float* e, float* nege,
                          float* f, float* dgamma, float* dbeta, const float* W, int K, float* Wt, float* We, float* pm_slabs = nullptr) {
  PrepArgs a;
  a.pm_slabs = pm_slabs;
  a.dg = dg; a.dg2 = dg2; a.g = g; a.zstar = zstar; a.B = B; a.C = C; a.mean = mean; a.invstd = invstd; a.scale = scale;
  a.batch_stats = batch_stats; a.inv_count = 1.0 / (double)count; a.hs = hs; a.e = e; a.nege = nege; a.f = f; a.dgamma = dgamma;
  a.dbeta = dbeta; a.W = W; a.K = K; a.Wt = Wt; a.We = We;
  return a;
}
int maxbwd_prep(const float* dg, const float* dg2, const float* g, const float* zstar, int B, int C, const float* mean, const float* invstd,
                const float* scale, int batch_stats, long long count, float* hs, float* e, float* nege, float* f, float* dgamma,
                float* dbeta, const float* W, int K, float* Wt, float* We, hipStream_t st, float* pm_slabs) {
  PN_CHECK_ARG((dg || dg2) && g && zstar && mean && invstd && scale && hs && e && nege && f, "maxbwd_prep: null pointer");
  PN_CHECK_ARG(!W || (Wt && We && K > 0), "maxbwd_prep: the transposed copies need Wt and We");
  PN_CHECK_ARG(!pm_slabs || (W && K == 128 && C % 32 == 0), "maxbwd_prep: the Pm slabs need the kernel, K = 128, C %% 32 == 0");
  const PrepArgs a = make_prep(dg, dg2, g, zstar, B, C, mean, invstd, scale, batch_stats, count, hs, e, nege, f, dgamma, dbeta, W, K, Wt, We, pm_slabs);
  hipLaunchKernelGGL(maxbwd_prep_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, a);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
// the same + the rows of the maxima (argq -> arg) in one launch
int maxbwd_prep_resolve(const float* dg, const float* dg2, const float* g, const float* zstar, int B, int C, const float* mean,
                        const float* invstd, const float* scale, int batch_stats, long long count, float* hs, float* e, float* nege,
                        float* f, float* dgamma, float* dbeta, const float* W, int K, float* Wt, float* We, const pn_operand* x,
                        const void* wf_hi, const void* wf_lo, int prec, const int* argq, int N, int* arg, hipStream_t st, float* pm_slabs) {
  PN_CHECK_ARG((dg || dg2) && g && zstar && mean && invstd && scale && hs && e && nege && f, "maxbwd_prep: null pointer");
  PN_CHECK_ARG(!pm_slabs || (W && K == 128 && C % 32 == 0), "maxbwd_prep: the Pm slabs need the kernel, K = 128, C %% 32 == 0");
  PN_CHECK_ARG(!W || (Wt && We && K > 0), "maxbwd_prep: the transposed copies need Wt and We");
  PN_CHECK_ARG(x && x->s1 && !x->s2 && wf_hi && argq && arg, "maxbwd_prep_resolve: null pointer");
  PN_CHECK_ARG(K <= RS_KMAX && K % 16 == 0 && C % 32 == 0, "maxbwd_prep_resolve: K must be a multiple of 16, at most 128, C a multiple of 32");
  PN_CHECK_ARG((reinterpret_cast<uintptr_t>(x->s1) & 15) == 0 && x->ld % 4 == 0, "maxbwd_prep_resolve: operand alignment");
  prec &= ~PN_STORE_BF16;
  PN_CHECK_ARG(prec == PN_PREC_BF16 || (prec == PN_PREC_BF16X3 && wf_lo), "maxbwd_prep_resolve: bad prec / missing lo weights");
  PN_CHECK_ARG(x->h16 == 0 || (x->h16 == 1 && x->ld % 8 == 0), "maxbwd_prep_resolve: 16-bit rows need ld %% 8 == 0");
  const PrepArgs a = make_prep(dg, dg2, g, zstar, B, C, mean, invstd, scale, batch_stats, count, hs, e, nege, f, dgamma, dbeta, W, K, Wt, We, pm_slabs);
  const int n_prep = cdiv(C, 32), qpc = cdiv(N, 32);
  const __bf16* wh = reinterpret_cast<const __bf16*>(wf_hi);
  const __bf16* wl = reinterpret_cast<const __bf16*>(wf_lo);
  if (prec == PN_PREC_BF16X3)
    hipLaunchKernelGGL((maxbwd_prep_resolve_kernel<2>), dim3(n_prep + B * qpc), dim3(256), 0, st, a, n_prep, *x, wh, wl, argq, N, K, C, qpc, arg);
  else
    hipLaunchKernelGGL((maxbwd_prep_resolve_kernel<1>), dim3(n_prep + B * qpc), dim3(256), 0, st, a, n_prep, *x, wh, wl, argq, N, K, C, qpc, arg);
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

int maxbwd_dw_batch(const DwJob* jobs, int n_jobs, int B, int N, int K, int C, hipStream_t st) {
  PN_CHECK_ARG(K <= 256, "maxbwd_dw: K must be <= 256 (K=%d)", K);
  for (int j0 = 0; j0 < n_jobs; j0 += 3) {
    DwBatch jb;
    memset(&jb, 0, sizeof(jb));
    const int n = (n_jobs - j0) < 3 ? (n_jobs - j0) : 3;
    for (int j = 0; j < n; ++j) {
      const DwJob& q = jobs[j0 + j];
      PN_CHECK_ARG(q.x.s1 && q.arg && q.hs && q.a1 && q.f && q.e && q.GW && q.dW, "maxbwd_dw: null pointer");
      jb.job[j] = q;
    }
    hipLaunchKernelGGL(maxbwd_dw_kernel, dim3(C, n), dim3(128), 0, st, jb, B, N, K, C);
    PN_CHECK_LAUNCH();
  }
  return PN_OK;
}

int maxbwd_dw(const pn_operand* x, const int* arg, const float* hs, int B, int N, int K, int C, const float* a1, const float* f,
              const float* e, const float* GW, float* dW, hipStream_t st) {
  PN_CHECK_ARG(x != nullptr, "maxbwd_dw: null pointer");
  const DwJob j{*x, arg, hs, a1, f, e, GW, dW};
  return maxbwd_dw_batch(&j, 1, B, N, K, C, st);
}

int maxbwd_q(const float* w, const float* f, int K, int C, float* q, hipStream_t st) {
  hipLaunchKernelGGL(maxbwd_q_kernel, dim3(K), dim3(64), 0, st, w, f, C, q);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int maxbwd_scatter(const int* arg, const float* hs, const float* wt, const float* q, int B, int N, int K, int C, float* D,
                   int store16, hipStream_t st) {
  PN_CHECK_ARG(arg && hs && wt && q && D, "maxbwd_scatter: null pointer");
  PN_CHECK_ARG(K <= 128 && K % 32 == 0, "maxbwd_scatter: K must be a multiple of 32, at most 128 (K=%d)", K);
  // PN_SCATTER_DBG (timing ablations, WRONG results): 1 = no hit processing, 2 = no stores, 4 = no rows of the maxima read
  static const int dbg = getenv("PN_SCATTER_DBG") ? atoi(getenv("PN_SCATTER_DBG")) : 0;
  if (N >= 2048) {
    const int qpc = cdiv(N, 64);
    hipLaunchKernelGGL(maxbwd_scatter_kernel<2>, dim3(B * qpc), dim3(256), 0, st, arg, hs, wt, q, N, K, C, qpc, D, store16, dbg);
  } else {
    const int qpc = cdiv(N, 32);
    hipLaunchKernelGGL(maxbwd_scatter_kernel<1>, dim3(B * qpc), dim3(256), 0, st, arg, hs, wt, q, N, K, C, qpc, D, store16, dbg);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// the scatter WITHOUT q + the reduction of n_slabs slabs of `elems` floats into pm + q = W f, one launch (the caller hands q to the
// data-gradient GEMM as its per-column constant)
int maxbwd_scatter_reduce(const int* arg, const float* hs, const float* wt, int B, int N, int K, int C, float* D, int store16,
                          const float* slabs, int n_slabs, long long elems, float* pm, const float* w, const float* f, float* q, hipStream_t st) {
  PN_CHECK_ARG(arg && hs && wt && D && slabs && pm && w && f && q, "maxbwd_scatter_reduce: null pointer");
  PN_CHECK_ARG(K <= 128 && K % 32 == 0 && n_slabs > 0 && elems > 0, "maxbwd_scatter_reduce: bad sizes (K=%d)", K);
  const int rb = N >= 2048 ? 2 : 1;
  const int qpc = cdiv(N, 32 * rb), n_scatter = B * qpc, nb = (int)cdivll(elems, 32);
  static const int dbg = getenv("PN_SCATTER_DBG") ? atoi(getenv("PN_SCATTER_DBG")) : 0;
  if (rb == 2)
    hipLaunchKernelGGL(maxbwd_scatter_reduce_kernel<2>, dim3(n_scatter + nb + cdiv(K, 4)), dim3(256), 0, st, arg, hs, wt, N, K, C, qpc, n_scatter, D,
                       store16, dbg, slabs, n_slabs, elems, pm, nb, w, f, q);
  else
    hipLaunchKernelGGL(maxbwd_scatter_reduce_kernel<1>, dim3(n_scatter + nb + cdiv(K, 4)), dim3(256), 0, st, arg, hs, wt, N, K, C, qpc, n_scatter, D,
                       store16, dbg, slabs, n_slabs, elems, pm, nb, w, f, q);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int max_resolve(const pn_operand* x, const void* wf_hi, const void* wf_lo, int prec, const int* argq, int B, int N, int K, int C, int* arg,
                hipStream_t st) {
  PN_CHECK_ARG(x && x->s1 && !x->s2 && wf_hi && argq && arg, "pn_max_resolve: null pointer");
  PN_CHECK_ARG(K <= 128 && K % 16 == 0 && C % 32 == 0, "pn_max_resolve: K must be a multiple of 16, at most 128, C a multiple of 32 (K=%d C=%d)", K, C);
  PN_CHECK_ARG((reinterpret_cast<uintptr_t>(x->s1) & 15) == 0 && x->ld % 4 == 0, "pn_max_resolve: operand alignment");
  prec &= ~PN_STORE_BF16;
  PN_CHECK_ARG(prec == PN_PREC_BF16 || (prec == PN_PREC_BF16X3 && wf_lo), "pn_max_resolve: bad prec / missing lo weights");
  PN_CHECK_ARG(x->h16 == 0 || (x->h16 == 1 && x->ld % 8 == 0), "pn_max_resolve: 16-bit rows need ld %% 8 == 0");
  const int qpc = cdiv(N, 32);
  const __bf16* wh = reinterpret_cast<const __bf16*>(wf_hi);
  const __bf16* wl = reinterpret_cast<const __bf16*>(wf_lo);
  if (prec == PN_PREC_BF16X3) hipLaunchKernelGGL((max_resolve_kernel<2>), dim3(B * qpc), dim3(256), 0, st, *x, wh, wl, argq, N, K, C, qpc, arg);
  else hipLaunchKernelGGL((max_resolve_kernel<1>), dim3(B * qpc), dim3(256), 0, st, *x, wh, wl, argq, N, K, C, qpc, arg);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

}  // namespace pn
