// Segmentation output layer (ConvLayer seg_l5_output: 128 -> Cseg, bias, softmax; pointnet/PointNet.py:141,287)
// fused with keras SparseCategoricalCrossentropy (pointnet_train.py:338) and its gradient.  Cseg is small
// (12 in the reference configs), so this is vector-ALU work: K*Cseg FMAs per point against 4*K bytes read.
#include "pn_common.h"
#include "pn_loss_bodies.h"

namespace pn {

constexpr int SEG_CM = 16;   // class slots held in registers; Cseg <= 16
constexpr int SEG_RPB = 128; // rows (threads) per block of the forward kernel: 256 blocks at 32,768 points, one per CU

// one thread per point: logits -> softmax -> (loss, accuracy, dlogits).
// MF = 1 (K = 128): the logits of a wave's 64 points come from the matrix cores -- two 32-row tiles, 8 k-steps, operands split
// into bf16 hi + lo (3 products, fp32-grade), weights staged once per block as a bf16 channel-major LDS image, the BN + ReLU of
// the input applied on load -- and are handed to their points through an LDS tile.  MF = 0: the vector-ALU form (any K % 4 == 0),
// which reads one weight from LDS per FMA and is LDS-bound (30 us vs the MFMA form's time at M = 32,768).
typedef __attribute__((ext_vector_type(8))) __bf16 seg_bf16x8;
typedef __attribute__((ext_vector_type(16))) float seg_f32x16;
constexpr int SEG_WP = 128 + 8;   // LDS pitch (bf16) of the weight image: conflict-free 16-byte rows

template <int MF>
__global__ __launch_bounds__(SEG_RPB) void seg_out_fwd_kernel(const pn_operand x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          long long M, int K, int C, const int* __restrict__ labels, float grad_scale,
                                                          float* __restrict__ probs, float* __restrict__ dlogits,
                                                          float* __restrict__ part /* [blocks][2 + SEG_CM] */) {
  extern __shared__ float sm[];
  float* ws = sm;                 // MF 0: [K][SEG_CM] fp32 weights.  MF 1: bf16 image [hi|lo][32][SEG_WP] + logit tiles (see below)
  // MF 1 layout: [hi image 32 x SEG_WP bf16][lo image][logit tile SEG_RPB x (SEG_CM+1) fp32][ca K][cc K]
  float* ca = MF == 1 ? reinterpret_cast<float*>(reinterpret_cast<char*>(sm) + 2 * 32 * SEG_WP * sizeof(__bf16)) + SEG_RPB * (SEG_CM + 1)
                      : ws + K * SEG_CM;    // [K]
  float* cc = ca + K;             // [K]
  if (MF == 0) {
    for (int t = threadIdx.x; t < K * SEG_CM; t += SEG_RPB) {
      const int k = t / SEG_CM, c = t % SEG_CM;
      ws[t] = c < C ? w[(long long)k * C + c] : 0.f;
    }
  } else {
    __bf16* whi = reinterpret_cast<__bf16*>(ws);
    __bf16* wlo = whi + 32 * SEG_WP;
    for (int t = threadIdx.x; t < 32 * 128; t += SEG_RPB) {
      const int c = t >> 7, k = t & 127;            // channel-major image W^T[c][k], channels >= C are zero
      const float v = c < C ? w[(long long)k * C + c] : 0.f;
      const __bf16 h = (__bf16)v;
      whi[c * SEG_WP + k] = h;
      wlo[c * SEG_WP + k] = (__bf16)(v - (float)h);
    }
  }
  for (int t = threadIdx.x; t < K; t += SEG_RPB) {
    ca[t] = x.ca ? x.ca[t] : 1.f;
    cc[t] = x.cc ? x.cc[t] : 0.f;
  }
  __syncthreads();
  const long long row = (long long)blockIdx.x * SEG_RPB + threadIdx.x;
  float loss = 0.f, corr = 0.f;
  float dl[SEG_CM];
#pragma unroll
  for (int c = 0; c < SEG_CM; ++c) dl[c] = 0.f;
  float* lgt = nullptr;           // MF 1: this block's logit tile [SEG_RPB rows][SEG_CM + 1]
  if (MF == 1) {
    const __bf16* whi = reinterpret_cast<const __bf16*>(ws);
    const __bf16* wlo = whi + 32 * SEG_WP;
    lgt = reinterpret_cast<float*>(const_cast<__bf16*>(wlo + 32 * SEG_WP));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, lg = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const long long r0 = (long long)blockIdx.x * SEG_RPB + wave * 64 + 32 * t;
      const long long rr = r0 + lr < M ? r0 + lr : M - 1;          // clamped: loads are unconditional
      const float* src = x.s1 + rr * x.ld + 8 * lg;
      seg_f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      float4 xv[8][2];
      if (x.h16) {                          // wave-uniform: eight bf16 = one 16-byte load, raw bits in the first quad
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xv[ks][0] = __builtin_bit_cast(float4, act_load8_raw(x.s1, rr * x.ld + 8 * lg + ks * 16));
      } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          xv[ks][0] = *reinterpret_cast<const float4*>(src + ks * 16);
          xv[ks][1] = *reinterpret_cast<const float4*>(src + ks * 16 + 4);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int k0 = ks * 16 + 8 * lg;
        float v[8];
        if (x.h16) {
          bf16x8_unpack(__builtin_bit_cast(uint4, xv[ks][0]), v);
        } else {
          v[0] = xv[ks][0].x; v[1] = xv[ks][0].y; v[2] = xv[ks][0].z; v[3] = xv[ks][0].w;
          v[4] = xv[ks][1].x; v[5] = xv[ks][1].y; v[6] = xv[ks][1].z; v[7] = xv[ks][1].w;
        }
        seg_bf16x8 ah, al;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = clamp_lo(fmaf(ca[k0 + e], v[e], cc[k0 + e]), x.lo);
          ah[e] = (__bf16)a;
          al[e] = (__bf16)(a - (float)ah[e]);
        }
        const seg_bf16x8 bh = *reinterpret_cast<const seg_bf16x8*>(whi + lr * SEG_WP + k0);
        const seg_bf16x8 bl = *reinterpret_cast<const seg_bf16x8*>(wlo + lr * SEG_WP + k0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      }
      if (lr < SEG_CM) {          // acc[e]: row (e & 3) + 8 (e >> 2) + 4 g of the tile, class lr
#pragma unroll
        for (int e = 0; e < 16; ++e) lgt[(wave * 64 + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * lg) * (SEG_CM + 1) + lr] = acc[e];
      }
    }
    __syncthreads();
  }
  if (row < M) {
    float acc[SEG_CM];
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c) acc[c] = (c < C && bias) ? bias[c] : 0.f;
    if (MF == 1) {
#pragma unroll
      for (int c = 0; c < SEG_CM; ++c) acc[c] += lgt[threadIdx.x * (SEG_CM + 1) + c];
    } else {
      const float* src = x.s1 + row * x.ld;
#pragma unroll 4
      for (int k = 0; k < K; k += 4) {
        float4 v;
        if (x.h16) {
          const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(x.s1) + row * x.ld + k);
          v = make_float4(bf16_bits_f32(t.x & 0xffffu), bf16_bits_f32(t.x >> 16), bf16_bits_f32(t.y & 0xffffu), bf16_bits_f32(t.y >> 16));
        } else {
          v = *reinterpret_cast<const float4*>(src + k);
        }
        const float a0 = clamp_lo(fmaf(ca[k], v.x, cc[k]), x.lo), a1 = clamp_lo(fmaf(ca[k + 1], v.y, cc[k + 1]), x.lo);
        const float a2 = clamp_lo(fmaf(ca[k + 2], v.z, cc[k + 2]), x.lo), a3 = clamp_lo(fmaf(ca[k + 3], v.w, cc[k + 3]), x.lo);
#pragma unroll
        for (int c = 0; c < SEG_CM; ++c) {
          acc[c] = fmaf(a0, ws[k * SEG_CM + c], acc[c]);
          acc[c] = fmaf(a1, ws[(k + 1) * SEG_CM + c], acc[c]);
          acc[c] = fmaf(a2, ws[(k + 2) * SEG_CM + c], acc[c]);
          acc[c] = fmaf(a3, ws[(k + 3) * SEG_CM + c], acc[c]);
        }
      }
    }
    float mx = -INFINITY;
    int am = 0;
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c)
      if (c < C && acc[c] > mx) { mx = acc[c]; am = c; }
    float p[SEG_CM];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c) {
      p[c] = c < C ? expf(acc[c] - mx) : 0.f;
      sum += p[c];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c) p[c] *= inv;
    if (probs) {
      float* po = probs + row * C;
#pragma unroll
      for (int c = 0; c < SEG_CM; ++c)
        if (c < C) po[c] = p[c];
    }
    if (labels) {
      const int y = labels[row];
      float qs = 0.f, py = 1.f;
#pragma unroll
      for (int c = 0; c < SEG_CM; ++c)
        if (c < C) {
          const float pc = clip_nan(p[c], 1e-7f, 1.f - 1e-7f);
          qs += pc;
          if (c == y) py = pc;
        }
      loss = -(logf(py) - logf(qs));
      corr = (am == y) ? 1.f : 0.f;
      float dp[SEG_CM];
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < SEG_CM; ++c) {
        dp[c] = 0.f;
        if (c < C) {
          const float pc = clip_nan(p[c], 1e-7f, 1.f - 1e-7f);
          const bool inr = (p[c] > 1e-7f) && (p[c] < 1.f - 1e-7f);
          dp[c] = inr ? (pc / qs - (c == y ? 1.f : 0.f)) / p[c] : 0.f;
          dot = fmaf(p[c], dp[c], dot);
        }
      }
#pragma unroll
      for (int c = 0; c < SEG_CM; ++c) dl[c] = c < C ? grad_scale * p[c] * (dp[c] - dot) : 0.f;
      if (dlogits) {
        float* d = dlogits + row * C;
#pragma unroll
        for (int c = 0; c < SEG_CM; ++c)
          if (c < C) d[c] = dl[c];
      }
    }
  }
  if (part) {
    // block partials: loss, correct, sum_rows dlogits[c]
    __shared__ float redw[SEG_RPB / 64][2 + SEG_CM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float l = wave_sum(loss), cr = wave_sum(corr);
    if (lane == 0) { redw[wave][0] = l; redw[wave][1] = cr; }
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c) {
      const float s = wave_sum(dl[c]);
      if (lane == 0) redw[wave][2 + c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 2 + SEG_CM)
      part[(long long)blockIdx.x * (2 + SEG_CM) + threadIdx.x] = redw[0][threadIdx.x] + redw[1][threadIdx.x];
    static_assert(SEG_RPB == 128, "two waves per block");
  }
}

// lanes <-> input channel k.  Per 128-row tile of one cloud:
//   dyhat[row][k] = relu'(.) * sum_c dlogits[row][c] * W[k][c]      (stored)
//   stat partials (sum dyhat, sum dyhat*z)                          [tile][2][K]
//   weight-gradient slab  sum_rows a[row][k] * dlogits[row][c]      [tile][K][C]
// K must be 128 (256 threads = 2 row streams x 128 channels).
__global__ __launch_bounds__(256) void seg_out_bwd_kernel(const pn_operand x, const float* __restrict__ w,
                                                          const float* __restrict__ dlogits, int N, int C, int tiles_per_cloud,
                                                          float* __restrict__ dyhat, float* __restrict__ stat_part,
                                                          float* __restrict__ wslab, int store16) {
  constexpr int K = 128;
  __shared__ float red[128][2 + SEG_CM];
  __shared__ __attribute__((aligned(16))) float out_s[128 * SEG_CM];
  const int bx = blockIdx.x, cloud = bx / tiles_per_cloud, tin = bx - cloud * tiles_per_cloud;
  const int k = threadIdx.x & 127, stream = threadIdx.x >> 7;
  float wk[SEG_CM], gw[SEG_CM];
#pragma unroll
  for (int c = 0; c < SEG_CM; ++c) {
    wk[c] = c < C ? w[(long long)k * C + c] : 0.f;
    gw[c] = 0.f;
  }
  const float ca = x.ca ? x.ca[k] : 1.f, cc = x.cc ? x.cc[k] : 0.f, lo = x.lo;
  const int r0 = tin * 128 + stream * 64, r1 = min(N, r0 + 64);
  float S1 = 0.f, S2 = 0.f;
  // the tile's rows of d(logits) (128 x C floats, contiguous) go through LDS once, coalesced: every lane of a wave needs the same C
  // values per row, and fetching them row by row through scalar loads cost one memory round trip per row
  {
    const long long gbase = ((long long)cloud * N + tin * 128) * C;
    const int nvalid = (min(N, tin * 128 + 128) - tin * 128) * C;
    // LDS rows are SEG_CM floats apart (zero padded): a row is then four 16-byte broadcast reads whatever C is
    for (int i = threadIdx.x; i < 128 * SEG_CM; i += 256) out_s[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < nvalid; i += 256) out_s[(i / C) * SEG_CM + (i % C)] = dlogits[gbase + i];
    __syncthreads();
  }
  act_switch(x.h16, [&](auto hx) { act_switch(store16, [&](auto hs) {
  constexpr bool HX = decltype(hx)::value, HS = decltype(hs)::value;
  // 8 rows of the layer input in flight per thread (a row at a time the loop ran at the latency of one load per row: 63 us at
  // B = 32, N = 2048 for 37 MB of traffic)
  for (int rb = r0; rb < r1; rb += 8) {
    float zz[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) zz[u] = act_ld<HX>(x.s1, ((long long)cloud * N + min(rb + u, r1 - 1)) * x.ld + k);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = rb + u;
      if (r < r1) {                 // wave-uniform
        const long long row = (long long)cloud * N + r;
        const float z = zz[u];
        const float pre = fmaf(ca, z, cc);
        const float a = clamp_lo(pre, lo);
        const float4* dl4 = reinterpret_cast<const float4*>(out_s + (r - tin * 128) * SEG_CM);      // LDS broadcast reads
        float gv[SEG_CM];
#pragma unroll
        for (int q = 0; q < SEG_CM / 4; ++q) {
          const float4 t = dl4[q];
          gv[4 * q] = t.x; gv[4 * q + 1] = t.y; gv[4 * q + 2] = t.z; gv[4 * q + 3] = t.w;
        }
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < SEG_CM; ++c) {       // classes >= C: zero gradient, zero weight
          d = fmaf(gv[c], wk[c], d);
          gw[c] = fmaf(a, gv[c], gw[c]);
        }
        if (!(pre > lo)) d = 0.f;   // relu'(pre) with lo = 0; lo = -inf keeps everything
        act_st<HS>(dyhat, row * K + k, d);
        S1 += d;
        S2 = fmaf(d, z, S2);
      }
    }
  }
  }); });
  __syncthreads();                 // out_s (the d(logits) tile) is reused for the weight-gradient slab below
  if (stream == 1) {
    red[k][0] = S1; red[k][1] = S2;
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c) red[k][2 + c] = gw[c];
  }
  __syncthreads();
  if (stream == 0) {
    if (stat_part) {
      stat_part[(long long)bx * 2 * K + k] = S1 + red[k][0];
      stat_part[(long long)bx * 2 * K + K + k] = S2 + red[k][1];
    }
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c)
      if (c < C) out_s[k * C + c] = gw[c] + red[k][2 + c];
  }
  __syncthreads();
  // the slab [K][C] of this tile is contiguous: written through LDS so that consecutive lanes store consecutive floats (a lane per
  // channel k writing its C values put 48-byte strides between lanes: 12 partial cache lines per store instruction)
  float* s = wslab + (long long)bx * K * C;
  for (int i = threadIdx.x; i < K * C; i += 256) s[i] = out_s[i];
}

// out[e] = sum_{i<n} part[i*stride + e]      one 1024-thread block per element: fixed assignment of rows to threads, wave
// shuffles, then the 16 wave sums in order -> bitwise reproducible
__global__ __launch_bounds__(1024) void sum_partials_kernel(const float* __restrict__ part, int n, int stride, int elems,
                                                            float* __restrict__ out) {
  __shared__ double wsum[16];
  if ((int)blockIdx.x < elems) sum_partials_body(part, n, stride, blockIdx.x, out, wsum);
}

// The tail of a forward pass with fused losses, one launch: the classification softmax + loss (workgroup 0: softmax_xent_rows), the
// segmentation loss / accuracy sums over the output kernel's per-block partials (the next n_sum workgroups) and the rotation loss
// (one more workgroup, when there is a target).
struct LossTailArgs {
  const float* logits; int R, C; const int* labels; float grad_scale; float *probs, *dlogits, *loss_sum, *correct;
  const float* part; int n, stride, n_sum; float* sum_out;
  const float *Rm, *T; int n_mse; float* mse_out;
};
__global__ __launch_bounds__(1024) void loss_tail_kernel(const LossTailArgs a) {
  __shared__ float rl[32], rc[32];
  __shared__ double wsum[16];
  int bx = blockIdx.x;
  if (bx == 0) { softmax_xent_rows_body(a.logits, a.R, a.C, a.labels, a.grad_scale, a.probs, a.dlogits, a.loss_sum, a.correct, rl, rc); return; }
  bx -= 1;
  if (bx < a.n_sum) { sum_partials_body(a.part, a.n, a.stride, bx, a.sum_out, wsum); return; }
  bx -= a.n_sum;
  if (bx == 0 && a.mse_out) mse_body(a.Rm, a.T, a.n_mse, 0.f, nullptr, a.mse_out, rl);
}
int loss_tail(const float* logits, int R, int C, const int* labels, float grad_scale, float* probs, float* dlogits, float* loss_sum,
              float* correct, const float* part, int n, int stride, int n_sum, float* sum_out, const float* Rm, const float* T, int n_mse,
              float* mse_out, hipStream_t st) {
  PN_CHECK_ARG(logits && probs && R > 0 && C > 0, "loss_tail: bad arguments");
  PN_CHECK_ARG(n_sum == 0 || (part && sum_out && n > 0), "loss_tail: bad partial-sum arguments");
  PN_CHECK_ARG(!mse_out || (Rm && T && n_mse > 0), "loss_tail: bad rotation-loss arguments");
  LossTailArgs a{logits, R, C, labels, grad_scale, probs, dlogits, loss_sum, correct, part, n, stride, n_sum, sum_out, Rm, T, n_mse, mse_out};
  hipLaunchKernelGGL(loss_tail_kernel, dim3(1 + n_sum + (mse_out ? 1 : 0)), dim3(1024), 0, st, a);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int seg_out_fwd(const pn_operand* x, const float* w, const float* bias, long long M, int K, int C, const int* labels,
                float grad_scale, float* probs, float* dlogits, float* part, hipStream_t st) {
  PN_CHECK_ARG(x && x->s1 && w, "seg_out_fwd: null pointer");
  PN_CHECK_ARG(C >= 1 && C <= SEG_CM, "seg_out_fwd: segmentation width %d not in [1,%d]", C, SEG_CM);
  PN_CHECK_ARG(K % 4 == 0 && K <= 1024 && x->ld % 4 == 0, "seg_out_fwd: bad K/ld");
  const size_t shm = (size_t)(K * SEG_CM + 2 * K) * sizeof(float);
  if (K == 128) {
    const size_t img = (size_t)2 * 32 * SEG_WP * sizeof(__bf16);
    const size_t tile = (size_t)SEG_RPB * (SEG_CM + 1) * sizeof(float);
    const size_t shm_mf = img + tile + (size_t)2 * K * sizeof(float);
    hipLaunchKernelGGL(seg_out_fwd_kernel<1>, dim3((unsigned)cdivll(M, SEG_RPB)), dim3(SEG_RPB), shm_mf, st, *x, w, bias, M, K, C, labels,
                       grad_scale, probs, dlogits, part);
  } else {
    hipLaunchKernelGGL(seg_out_fwd_kernel<0>, dim3((unsigned)cdivll(M, SEG_RPB)), dim3(SEG_RPB), shm, st, *x, w, bias, M, K, C, labels,
                       grad_scale, probs, dlogits, part);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}
int seg_out_part_stride() { return 2 + SEG_CM; }
int seg_out_part_rows() { return SEG_RPB; }

int seg_out_bwd(const pn_operand* x, const float* w, const float* dlogits, int B, int N, int K, int C, float* dyhat, float* stat_part,
                float* wslab, hipStream_t st, int store16) {
  PN_CHECK_ARG(x && x->s1 && w && dlogits && dyhat && wslab, "seg_out_bwd: null pointer");
  PN_CHECK_ARG(K == 128, "seg_out_bwd: the layer feeding the segmentation output must be 128 wide (K=%d)", K);
  PN_CHECK_ARG(C >= 1 && C <= SEG_CM, "seg_out_bwd: segmentation width %d not in [1,%d]", C, SEG_CM);
  const int tpc = cdiv(N, 128);
  hipLaunchKernelGGL(seg_out_bwd_kernel, dim3(B * tpc), dim3(256), 0, st, *x, w, dlogits, N, C, tpc, dyhat, stat_part, wslab, store16);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int sum_partials(const float* part, int n, int stride, int elems, float* out, hipStream_t st) {
  PN_CHECK_ARG(part && out && n > 0 && elems > 0, "sum_partials: bad arguments");
  hipLaunchKernelGGL(sum_partials_kernel, dim3(elems), dim3(1024), 0, st, part, n, stride, elems, out);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

}  // namespace pn
