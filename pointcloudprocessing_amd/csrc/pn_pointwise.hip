// Bandwidth-bound per-point kernels and the small finalisers between contractions (gfx950).
#include <cstring>
#include "pn_common.h"
#include "pn_slab_reduce.h"
#include "pn_internal.h"

namespace pn {

// ------------------------------------------------------------------------------------------------------
// ConvLayer with Cin = 3 (PointNet.py:406, 120): lane <-> output channel, a wave walks rows; the 3 inputs
// of a row are wave-uniform loads.  Tile = 128 rows of one cloud, 4 waves x 32 rows.  C = 64 * CG.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3_fwd_kernel(const float* __restrict__ x3, const float* __restrict__ w,
                                                        long long wcs, int N, int C, int tiles_per_cloud,
                                                        float* __restrict__ z, float* __restrict__ part, int store16,
                                                        const float* __restrict__ Rm, float* __restrict__ weff_out,
                                                        float* __restrict__ r_copy) {
  __shared__ float red[4][2][64];
  const int bx = blockIdx.x, cloud = bx / tiles_per_cloud, tin = bx - cloud * tiles_per_cloud;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.y * 64 + lane;
  const float* wb = w + (long long)cloud * wcs;
  float w0 = wb[c], w1 = wb[C + c], w2 = wb[2 * C + c];
  if (Rm) {
    // the input transform folded into the kernel: Weff[b] (3, C) = R[b] (3, 3) @ W (3, C) -- tf.matmul(pc, R) then the layer,
    // (pc . R) . W = pc . (R . W); the cloud's first tile also leaves Weff (and a copy of R: the model's third output) in memory
    const float* r = Rm + (long long)cloud * 9;
    const float e0 = fmaf(r[2], w2, fmaf(r[1], w1, r[0] * w0));
    const float e1 = fmaf(r[5], w2, fmaf(r[4], w1, r[3] * w0));
    const float e2 = fmaf(r[8], w2, fmaf(r[7], w1, r[6] * w0));
    w0 = e0; w1 = e1; w2 = e2;
    if (tin == 0 && wave == 0) {
      if (weff_out) {
        float* o = weff_out + (long long)cloud * 3 * C;
        o[c] = w0; o[C + c] = w1; o[2 * C + c] = w2;
      }
      if (r_copy && blockIdx.y == 0 && lane < 9) r_copy[(long long)cloud * 9 + lane] = r[lane];
    }
  }
  // readfirstlane: the wave index is uniform, but only this tells the compiler -- the three coordinates of a row then arrive through
  // scalar loads instead of three vector loads of one address each
  const int r0 = tin * 128 + __builtin_amdgcn_readfirstlane(wave) * 32;
  const int r1 = min(N, r0 + 32);
  float s1 = 0.f, s2 = 0.f;
  act_switch(store16, [&](auto h) {
    constexpr bool H = decltype(h)::value;
#pragma unroll 4
    for (int r = r0; r < r1; ++r) {
      const long long row = (long long)cloud * N + r;
      const float a0 = x3[row * 3], a1 = x3[row * 3 + 1], a2 = x3[row * 3 + 2];
      const float v = fmaf(a2, w2, fmaf(a1, w1, a0 * w0));
      if (z) act_st<H>(z, row * C + c, v);
      s1 += v;
      s2 = fmaf(v, v, s2);
    }
  });
  if (part) {
    red[wave][0][lane] = s1;
    red[wave][1][lane] = s2;
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, l = tid & 63;
      const float s = red[0][which][l] + red[1][which][l] + red[2][which][l] + red[3][which][l];
      part[(long long)bx * 2 * C + which * C + blockIdx.y * 64 + l] = s;
    }
  }
}

int conv3_fwd(const float* x3, const float* w, long long wcs, int B, int N, int C, float* z, float* part, hipStream_t st, int store16,
              const float* Rm, float* weff_out, float* r_copy) {
  PN_CHECK_ARG(x3 && w, "pn_conv3_fwd: null pointer");
  PN_CHECK_ARG(B > 0 && N > 0, "pn_conv3_fwd: B and N must be positive");
  PN_CHECK_ARG(C >= 64 && C % 64 == 0, "pn_conv3_fwd: C must be a multiple of 64 (C=%d)", C);
  const int tpc = cdiv(N, 128);
  hipLaunchKernelGGL(conv3_fwd_kernel, dim3(B * tpc, C / 64), dim3(256), 0, st, x3, w, wcs, N, C, tpc, z, part, store16, Rm, weff_out,
                     r_copy);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// weight gradient: slabs[tile][kk][c] = sum_rows x3[row][kk] * dz[row][c]
__global__ __launch_bounds__(256) void conv3_wgrad_kernel(const float* __restrict__ x3, const pn_operand dz, int N, int C,
                                                          int tiles_per_cloud, float* __restrict__ slabs) {
  __shared__ float red[4][3][64];
  const int bx = blockIdx.x, cloud = bx / tiles_per_cloud, tin = bx - cloud * tiles_per_cloud;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.y * 64 + lane;
  const float ca = dz.ca ? dz.ca[c] : 1.f, cb = (dz.s2 && dz.cb) ? dz.cb[c] : 0.f, cc = dz.cc ? dz.cc[c] : 0.f;
  const int r0 = tin * 128 + __builtin_amdgcn_readfirstlane(wave) * 32;     // provably wave-uniform: x3 rows through scalar loads
  const int r1 = min(N, r0 + 32);
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  const bool two = dz.s2 != nullptr;
  if (dz.h16) {
    // bf16 sources: a lane takes a PAIR of channels (4-byte loads) and every other row of the wave's 32; the two row halves of a
    // pair meet through a cross-half shuffle and land in LDS as in the fp32 form
    const int pr = lane & 31, half = lane >> 5;
    const int c2 = blockIdx.y * 64 + 2 * pr;
    const float ca0 = dz.ca ? dz.ca[c2] : 1.f, ca1 = dz.ca ? dz.ca[c2 + 1] : 1.f;
    const float cb0 = (two && dz.cb) ? dz.cb[c2] : 0.f, cb1 = (two && dz.cb) ? dz.cb[c2 + 1] : 0.f;
    const float cc0 = dz.cc ? dz.cc[c2] : 0.f, cc1 = dz.cc ? dz.cc[c2 + 1] : 0.f;
    const unsigned* s1 = reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(dz.s1) + c2);
    const unsigned* s2 = two ? reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(dz.s2) + c2) : nullptr;
    const long long ld2 = dz.ld / 2;
    float h0[3] = {0.f, 0.f, 0.f}, h1[3] = {0.f, 0.f, 0.f};
    for (int rb = r0; rb < r1; rb += 16) {
      unsigned y[8], zz[8];
      float xa[8], xb[8], xc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {            // 8 rows per half-wave in flight; rows past the end are clamped and masked below
        const long long row = (long long)cloud * N + min(rb + 2 * u + half, r1 - 1);
        y[u] = s1[row * ld2];
        zz[u] = two ? s2[row * ld2] : 0u;
        xa[u] = x3[row * 3]; xb[u] = x3[row * 3 + 1]; xc[u] = x3[row * 3 + 2];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool ok = rb + 2 * u + half < r1;
        float d0 = fmaf(cb0, bf16_bits_f32(zz[u] & 0xffffu), fmaf(ca0, bf16_bits_f32(y[u] & 0xffffu), cc0));
        float d1 = fmaf(cb1, __builtin_bit_cast(float, zz[u] & 0xffff0000u), fmaf(ca1, __builtin_bit_cast(float, y[u] & 0xffff0000u), cc1));
        d0 = ok ? clamp_lo(d0, dz.lo) : 0.f;
        d1 = ok ? clamp_lo(d1, dz.lo) : 0.f;
        h0[0] = fmaf(xa[u], d0, h0[0]); h0[1] = fmaf(xb[u], d0, h0[1]); h0[2] = fmaf(xc[u], d0, h0[2]);
        h1[0] = fmaf(xa[u], d1, h1[0]); h1[1] = fmaf(xb[u], d1, h1[1]); h1[2] = fmaf(xc[u], d1, h1[2]);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      h0[kk] += __shfl_xor(h0[kk], 32, 64);
      h1[kk] += __shfl_xor(h1[kk], 32, 64);
      if (half == 0) { red[wave][kk][2 * pr] = h0[kk]; red[wave][kk][2 * pr + 1] = h1[kk]; }
    }
  } else {
    for (int rb = r0; rb < r1; rb += 8) {
      float y[8], zz[8], xa[8], xb[8], xc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {            // 8 rows in flight; rows past the end are clamped and masked below
        const long long row = (long long)cloud * N + min(rb + u, r1 - 1);
        y[u] = dz.s1[row * dz.ld + c];
        zz[u] = two ? dz.s2[row * dz.ld + c] : 0.f;
        xa[u] = x3[row * 3]; xb[u] = x3[row * 3 + 1]; xc[u] = x3[row * 3 + 2];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float d = fmaf(ca, y[u], cc);
        d = fmaf(cb, zz[u], d);
        d = (rb + u < r1) ? clamp_lo(d, dz.lo) : 0.f;
        g0 = fmaf(xa[u], d, g0);
        g1 = fmaf(xb[u], d, g1);
        g2 = fmaf(xc[u], d, g2);
      }
    }
    red[wave][0][lane] = g0; red[wave][1][lane] = g1; red[wave][2][lane] = g2;
  }
  __syncthreads();
  if (tid < 192) {
    const int kk = tid >> 6, l = tid & 63;
    slabs[(long long)bx * 3 * C + kk * C + blockIdx.y * 64 + l] = red[0][kk][l] + red[1][kk][l] + red[2][kk][l] + red[3][kk][l];
  }
}

int conv3_wgrad(const float* x3, const pn_operand* dz, int B, int N, int C, float* slabs, hipStream_t st) {
  PN_CHECK_ARG(x3 && dz && dz->s1 && slabs, "pn_conv3_wgrad: null pointer");
  PN_CHECK_ARG(B > 0 && N > 0, "pn_conv3_wgrad: B and N must be positive");
  PN_CHECK_ARG(C >= 64 && C % 64 == 0 && dz->ld >= C, "pn_conv3_wgrad: bad C/ld");
  const int tpc = cdiv(N, 128);
  hipLaunchKernelGGL(conv3_wgrad_kernel, dim3(B * tpc, C / 64), dim3(256), 0, st, x3, *dz, N, C, tpc, slabs);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int per_group, long long elems,
                                                          float* __restrict__ out) {
  __shared__ float red[8][32];
  slab_reduce_block(slabs, per_group, elems, out, blockIdx.x, blockIdx.y, red);
}

// The parameter gradients of a backward pass are read by nobody before the optimizer, so their slab reductions wait and go out
// together: one launch instead of one per layer (every dependent launch costs >= 4.5 us on this stack).
constexpr int SLAB_BATCH_MAX = 12;
struct SlabBatch {
  const float* slabs[SLAB_BATCH_MAX];
  float* out[SLAB_BATCH_MAX];
  long long elems[SLAB_BATCH_MAX];
  int n_slabs[SLAB_BATCH_MAX];
  int blk_end[SLAB_BATCH_MAX];      // exclusive prefix of 32-element blocks
  int n_jobs;
};
__global__ __launch_bounds__(256) void slab_reduce_batch_kernel(const SlabBatch jb) {
  __shared__ float red[8][32];
  const int bx = blockIdx.x;
  int j = 0;
  while (j + 1 < jb.n_jobs && bx >= jb.blk_end[j]) ++j;          // block-uniform
  const int first = j ? jb.blk_end[j - 1] : 0;
  slab_reduce_block(jb.slabs[j], jb.n_slabs[j], jb.elems[j], jb.out[j], bx - first, 0, red);
}

// slab reduction + an independent tiny job of the max-pool backward in one launch: q[k] = sum_c f[c] * W[k][c], one wave per k
// (the blocks behind the reduction's; same lane stride and wave sum as a stand-alone launch would use)
__global__ __launch_bounds__(256) void slab_reduce_q_kernel(const float* __restrict__ slabs, int n_slabs, long long elems, float* __restrict__ out,
                                                            int nb_reduce, const float* __restrict__ w, const float* __restrict__ f, int K, int C,
                                                            float* __restrict__ q) {
  __shared__ float red[8][32];
  if ((int)blockIdx.x < nb_reduce) {
    slab_reduce_block(slabs, n_slabs, elems, out, blockIdx.x, 0, red);
    return;
  }
  slab_q_body((int)blockIdx.x - nb_reduce, w, f, K, C, q);
}

int slab_reduce_q(const float* slabs, int n_slabs, long long elems, float* out, const float* w, const float* f, int K, int C, float* q,
                  hipStream_t st) {
  PN_CHECK_ARG(slabs && out && w && f && q, "slab_reduce_q: null pointer");
  PN_CHECK_ARG(n_slabs > 0 && elems > 0 && K > 0 && C > 0, "slab_reduce_q: bad sizes");
  const int nb = (int)cdivll(elems, 32);
  hipLaunchKernelGGL(slab_reduce_q_kernel, dim3(nb + cdiv(K, 4)), dim3(256), 0, st, slabs, n_slabs, elems, out, nb, w, f, K, C, q);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int slab_reduce(const float* slabs, int n_slabs, int per_group, long long elems, float* out, hipStream_t st) {
  PN_CHECK_ARG(slabs && out, "pn_slab_reduce: null pointer");
  PN_CHECK_ARG(n_slabs > 0 && per_group > 0 && n_slabs % per_group == 0 && elems > 0, "pn_slab_reduce: bad sizes");
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)cdivll(elems, 32), n_slabs / per_group), dim3(256), 0, st, slabs,
                     per_group, elems, out);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int slab_reduce_batch(const SlabJob* jobs, int n_jobs, hipStream_t st) {
  for (int j0 = 0; j0 < n_jobs; j0 += SLAB_BATCH_MAX) {
    SlabBatch jb;
    memset(&jb, 0, sizeof(jb));
    jb.n_jobs = (n_jobs - j0) < SLAB_BATCH_MAX ? (n_jobs - j0) : SLAB_BATCH_MAX;
    long long blocks = 0;
    for (int j = 0; j < jb.n_jobs; ++j) {
      const SlabJob& q = jobs[j0 + j];
      PN_CHECK_ARG(q.slabs && q.out && q.n_slabs > 0 && q.elems > 0, "slab_reduce_batch: bad job %d", j0 + j);
      jb.slabs[j] = q.slabs; jb.out[j] = q.out; jb.elems[j] = q.elems; jb.n_slabs[j] = q.n_slabs;
      blocks += cdivll(q.elems, 32);
      PN_CHECK_ARG(blocks < (1ll << 30), "slab_reduce_batch: too many elements");
      jb.blk_end[j] = (int)blocks;
    }
    hipLaunchKernelGGL(slab_reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, jb);
    PN_CHECK_LAUNCH();
  }
  return PN_OK;
}

// ------------------------------------------------------------------------------------------------------
// BatchNormalization coefficient finalisers (keras BatchNormalization semantics, see pointnet_hip.h)
// ------------------------------------------------------------------------------------------------------
// block = 16 channels x 16 partitions of the tile range, 4 tiles in flight per thread; fp64 combine in a fixed order
__device__ __forceinline__ void reduce_tiles_2(const float* __restrict__ part, int n_tiles, int C, int c, int ty, double (*red)[2][16],
                                               int tx, double& s1, double& s2) {
  double a = 0.0, b = 0.0;
  if (c < C) {
    const float* p = part + c;
    // sixteen tiles (32 independent loads) in flight per thread: the kernel is one memory round trip per batch, and at the usual
    // 256 tiles a thread has exactly one batch.  The summation order (tiles ascending per thread, then the 16 threads) is unchanged.
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

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int n_tiles, int C, double inv_count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ mm, float* __restrict__ mv, float momentum,
                                                          float eps, int use_batch, int update, float* __restrict__ mean_o,
                                                          float* __restrict__ invstd_o, float* __restrict__ scale_o,
                                                          float* __restrict__ shift_o) {
  __shared__ double red[16][2][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + tx;
  // the channel's parameters do not depend on the sums: requested with the partials (behind the reduction's barrier they would be a
  // second memory round trip of a launch that is nothing but round trips)
  const int cq = c < C ? c : C - 1;
  const float gam = gamma[cq], bet = beta[cq], mm0 = mm[cq], mv0 = mv[cq];
  double s1 = 0.0, s2 = 0.0;
  if (use_batch) reduce_tiles_2(part, n_tiles, C, c, ty, red, tx, s1, s2);
  if (ty != 0 || c >= C) return;
  float mean, var;
  if (use_batch) {
    const double m = s1 * inv_count;
    double v = s2 * inv_count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    if (update) {
      mm[c] = mm0 * momentum + mean * (1.f - momentum);
      mv[c] = mv0 * momentum + var * (1.f - momentum);
    }
  } else {
    mean = mm0;
    var = mv0;
  }
  const float invstd = 1.0f / sqrtf(var + eps);
  const float sc = gam * invstd;
  if (mean_o) mean_o[c] = mean;
  if (invstd_o) invstd_o[c] = invstd;
  scale_o[c] = sc;
  shift_o[c] = bet - mean * sc;
}

int bn_finalize(const float* part, int n_tiles, int C, long long count, const float* gamma, const float* beta, float* mm,
                float* mv, float momentum, float eps, int use_batch, int update, float* mean, float* invstd, float* scale,
                float* shift, hipStream_t st) {
  PN_CHECK_ARG(gamma && beta && mm && mv && scale && shift, "pn_bn_finalize: null pointer");
  PN_CHECK_ARG(C > 0, "pn_bn_finalize: C must be positive");
  PN_CHECK_ARG(!use_batch || (part && n_tiles > 0 && count > 0), "pn_bn_finalize: batch statistics need partials");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, st, part, n_tiles, C, 1.0 / (double)(count > 0 ? count : 1),
                     gamma, beta, mm, mv, momentum, eps, use_batch, update, mean, invstd, scale, shift);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int n_tiles, int C,
                                                              double inv_count, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              int batch_stats, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ ca,
                                                              float* __restrict__ cb, float* __restrict__ cc) {
  __shared__ double red[16][2][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + tx;
  const int cq = c < C ? c : C - 1;
  const float gam = gamma[cq], mean_c = mean[cq], invstd_c = invstd[cq];      // requested with the partials (see bn_finalize_kernel)
  double s1 = 0.0, s2 = 0.0;
  if (batch_stats) reduce_tiles_2(part, n_tiles, C, c, ty, red, tx, s1, s2);
  if (ty != 0 || c >= C) return;
  const float a = gam * invstd_c;
  if (!batch_stats) {
    ca[c] = a; cb[c] = 0.f; cc[c] = 0.f;
    return;
  }
  // S1 = sum dy_hat ; S2 = sum dy_hat * zhat, zhat = (z - mean) * invstd
  const double S1 = s1;
  const double S2 = (s2 - (double)mean_c * s1) * (double)invstd_c;
  if (dgamma) dgamma[c] = (float)S2;
  if (dbeta) dbeta[c] = (float)S1;
  const double b = -(double)a * (double)invstd_c * S2 * inv_count;
  ca[c] = a;
  cb[c] = (float)b;
  cc[c] = (float)(-(double)a * S1 * inv_count - b * (double)mean_c);
}

int bn_bwd_finalize(const float* part, int n_tiles, int C, long long count, const float* gamma, const float* mean,
                    const float* invstd, int batch_stats, float* dgamma, float* dbeta, float* ca, float* cb, float* cc,
                    hipStream_t st) {
  PN_CHECK_ARG(gamma && invstd && ca && cb && cc, "pn_bn_bwd_finalize: null pointer");
  PN_CHECK_ARG(!batch_stats || (part && mean && n_tiles > 0 && count > 0), "pn_bn_bwd_finalize: batch statistics need partials");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, st, part, n_tiles, C,
                     1.0 / (double)(count > 0 ? count : 1), gamma, mean, invstd, batch_stats, dgamma, dbeta, ca, cb, cc);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// sgn[c] = +1 if gamma[c] >= 0 else -1 (the sign of the BN scale, known before the statistics are)
__global__ void sign_kernel(const float* __restrict__ gamma, int C, float* __restrict__ sgn) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) sgn[c] = gamma[c] >= 0.f ? 1.f : -1.f;
}
int sign_of(const float* gamma, int C, float* sgn, hipStream_t st) {
  hipLaunchKernelGGL(sign_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, gamma, C, sgn);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// ------------------------------------------------------------------------------------------------------
// finish tf.reduce_max (PointNet.py:248, 429)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void max_finalize_kernel(const float* __restrict__ pmax, const int* __restrict__ pidx,
                                                           int tiles_per_cloud, int C, int n_rows, const float* __restrict__ sgn,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ g, float* __restrict__ zstar,
                                                           int* __restrict__ arg) {
  const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int t0 = 0; t0 < tiles_per_cloud; t0 += 16) {      // 16 tiles per pass: all 32 loads in flight before the compare chain
    float v[16];
    int ix[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int t = min(t0 + q, tiles_per_cloud - 1);
      const long long o = ((long long)b * tiles_per_cloud + t) * C + c;
      v[q] = pmax[o];
      ix[q] = pidx[o];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (t0 + q < tiles_per_cloud && (v[q] > best || (v[q] == best && ix[q] < bi))) { best = v[q]; bi = ix[q]; }
  }
  const float zs = (sgn[c] < 0.f ? -1.f : 1.f) * best;   // sgn may be gamma itself
  const long long o = (long long)b * C + c;
  g[o] = clamp_lo(fmaf(scale[c], zs, shift[c]), 0.f);
  if (zstar) zstar[o] = zs;
  if (arg) arg[o] = (bi >= 0 && bi < n_rows) ? bi : 0;   // NaN inputs leave no winner: keep the index in range
}

int max_finalize(const float* pmax, const int* pidx, int B, int tpc, int C, int n_rows, const float* sgn, const float* scale,
                 const float* shift, float* g, float* zstar, int* arg, hipStream_t st) {
  PN_CHECK_ARG(pmax && pidx && sgn && scale && shift && g, "pn_max_finalize: null pointer");
  PN_CHECK_ARG(B > 0 && tpc > 0 && C > 0, "pn_max_finalize: bad sizes");
  hipLaunchKernelGGL(max_finalize_kernel, dim3(cdiv(C, 256), B), dim3(256), 0, st, pmax, pidx, tpc, C, n_rows, sgn, scale, shift, g,
                     zstar, arg);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// tf.matmul(pc, R) for K = 3 (PointNet.py:207): out[b, n, :] = pc[b, n, :] . R[b]   (the model folds this into mlp_1_1's weights;
// the stand-alone op exists for the module API and the parity tests)
__global__ __launch_bounds__(256) void bmm3_kernel(const float* __restrict__ x, const float* __restrict__ R, int N, long long M,
                                                   float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const float* r = R + (i / N) * 9;
  const float a = x[3 * i], b = x[3 * i + 1], c = x[3 * i + 2];
  out[3 * i + 0] = fmaf(c, r[6], fmaf(b, r[3], a * r[0]));
  out[3 * i + 1] = fmaf(c, r[7], fmaf(b, r[4], a * r[1]));
  out[3 * i + 2] = fmaf(c, r[8], fmaf(b, r[5], a * r[2]));
}
int bmm3(const float* x, const float* R, int B, int N, float* out, hipStream_t st) {
  const long long M = (long long)B * N;
  hipLaunchKernelGGL(bmm3_kernel, dim3((unsigned)cdivll(M, 256)), dim3(256), 0, st, x, R, N, M, out);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

}  // namespace pn
