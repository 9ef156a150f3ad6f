// Segmentation output layer (ConvLayer seg_l5_output: 128 -> Cseg, bias, softmax; pointnet/PointNet.py:141,287)
// fused with keras SparseCategoricalCrossentropy (pointnet_train.py:338) and its gradient.  Cseg is small
// (12 in the reference configs), so this is vector-ALU work: K*Cseg FMAs per point against 4*K bytes read.
#include "pn_common.h"

namespace pn {

constexpr int SEG_CM = 16;   // class slots held in registers; Cseg <= 16
constexpr int SEG_RPB = 128; // rows (threads) per block of the forward kernel: 256 blocks at 32,768 points, one per CU

// one thread per point: logits -> softmax -> (loss, accuracy, dlogits).
// CT > 0: the segmentation width is a compile-time constant, so the (wave-uniform) weights and BN coefficients come through scalar
// loads as SGPR operands of the FMAs; the LDS copy (one ds_read per FMA) made the generic form LDS-bound.
template <int CT>
__global__ __launch_bounds__(SEG_RPB) void seg_out_fwd_kernel(const pn_operand x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          long long M, int K, int C, const int* __restrict__ labels, float grad_scale,
                                                          float* __restrict__ probs, float* __restrict__ dlogits,
                                                          float* __restrict__ part /* [blocks][2 + SEG_CM] */) {
  extern __shared__ float sm[];
  float* ws = sm;                 // [K][SEG_CM]
  float* ca = ws + K * SEG_CM;    // [K]
  float* cc = ca + K;             // [K]
  if (CT == 0) {
    for (int t = threadIdx.x; t < K * SEG_CM; t += SEG_RPB) {
      const int k = t / SEG_CM, c = t % SEG_CM;
      ws[t] = c < C ? w[(long long)k * C + c] : 0.f;
    }
    for (int t = threadIdx.x; t < K; t += SEG_RPB) {
      ca[t] = x.ca ? x.ca[t] : 1.f;
      cc[t] = x.cc ? x.cc[t] : 0.f;
    }
    __syncthreads();
  }
  const long long row = (long long)blockIdx.x * SEG_RPB + threadIdx.x;
  float loss = 0.f, corr = 0.f;
  float dl[SEG_CM];
#pragma unroll
  for (int c = 0; c < SEG_CM; ++c) dl[c] = 0.f;
  if (row < M) {
    float acc[SEG_CM];
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c) acc[c] = (c < C && bias) ? bias[c] : 0.f;
    const float* src = x.s1 + row * x.ld;
    if (CT > 0) {
      const float* __restrict__ gca = x.ca;
      const float* __restrict__ gcc = x.cc;
#pragma unroll 2
      for (int k = 0; k < K; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(src + k);
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float a = fmaxf(fmaf(gca ? gca[k + q] : 1.f, vv[q], gcc ? gcc[k + q] : 0.f), x.lo);
          const float* __restrict__ wk = w + (long long)(k + q) * CT;      // wave-uniform address: scalar loads
#pragma unroll
          for (int c = 0; c < CT; ++c) acc[c] = fmaf(a, wk[c], acc[c]);
        }
      }
    } else {
#pragma unroll 4
      for (int k = 0; k < K; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(src + k);
        const float a0 = fmaxf(fmaf(ca[k], v.x, cc[k]), x.lo), a1 = fmaxf(fmaf(ca[k + 1], v.y, cc[k + 1]), x.lo);
        const float a2 = fmaxf(fmaf(ca[k + 2], v.z, cc[k + 2]), x.lo), a3 = fmaxf(fmaf(ca[k + 3], v.w, cc[k + 3]), x.lo);
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
          const float pc = fminf(fmaxf(p[c], 1e-7f), 1.f - 1e-7f);
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
          const float pc = fminf(fmaxf(p[c], 1e-7f), 1.f - 1e-7f);
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
                                                          float* __restrict__ wslab) {
  constexpr int K = 128;
  __shared__ float red[128][2 + SEG_CM];
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
  for (int r = r0; r < r1; ++r) {
    const long long row = (long long)cloud * N + r;
    const float z = x.s1[row * x.ld + k];
    const float pre = fmaf(ca, z, cc);
    const float a = fmaxf(pre, lo);
    const float* dl = dlogits + row * C;
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c)
      if (c < C) {
        const float g = dl[c];
        d = fmaf(g, wk[c], d);
        gw[c] = fmaf(a, g, gw[c]);
      }
    if (!(pre > lo)) d = 0.f;   // relu'(pre) with lo = 0; lo = -inf keeps everything
    dyhat[row * K + k] = d;
    S1 += d;
    S2 = fmaf(d, z, S2);
  }
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
    float* s = wslab + ((long long)bx * K + k) * C;
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c)
      if (c < C) s[c] = gw[c] + red[k][2 + c];
  }
}

// out[e] = sum_{i<n} part[i*stride + e]
__global__ __launch_bounds__(64) void sum_partials_kernel(const float* __restrict__ part, int n, int stride, int elems,
                                                          float* __restrict__ out) {
  const int e = blockIdx.x;
  if (e >= elems) return;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) acc += (double)part[(long long)i * stride + e];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (threadIdx.x == 0) out[e] = (float)acc;
}

int seg_out_fwd(const pn_operand* x, const float* w, const float* bias, long long M, int K, int C, const int* labels,
                float grad_scale, float* probs, float* dlogits, float* part, hipStream_t st) {
  PN_CHECK_ARG(x && x->s1 && w, "seg_out_fwd: null pointer");
  PN_CHECK_ARG(C >= 1 && C <= SEG_CM, "seg_out_fwd: segmentation width %d not in [1,%d]", C, SEG_CM);
  PN_CHECK_ARG(K % 4 == 0 && K <= 1024 && x->ld % 4 == 0, "seg_out_fwd: bad K/ld");
  const size_t shm = (size_t)(K * SEG_CM + 2 * K) * sizeof(float);
  // seg_out_fwd_kernel<12> (weights through scalar loads) measured 39.7 us against 30.6 us for the LDS form at M = 32,768: kept
  // only as the template's second instantiation for experiments
  hipLaunchKernelGGL(seg_out_fwd_kernel<0>, dim3((unsigned)cdivll(M, SEG_RPB)), dim3(SEG_RPB), shm, st, *x, w, bias, M, K, C, labels,
                     grad_scale, probs, dlogits, part);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
int seg_out_part_stride() { return 2 + SEG_CM; }
int seg_out_part_rows() { return SEG_RPB; }

int seg_out_bwd(const pn_operand* x, const float* w, const float* dlogits, int B, int N, int K, int C, float* dyhat, float* stat_part,
                float* wslab, hipStream_t st) {
  PN_CHECK_ARG(x && x->s1 && w && dlogits && dyhat && wslab, "seg_out_bwd: null pointer");
  PN_CHECK_ARG(K == 128, "seg_out_bwd: the layer feeding the segmentation output must be 128 wide (K=%d)", K);
  PN_CHECK_ARG(C >= 1 && C <= SEG_CM, "seg_out_bwd: segmentation width %d not in [1,%d]", C, SEG_CM);
  const int tpc = cdiv(N, 128);
  hipLaunchKernelGGL(seg_out_bwd_kernel, dim3(B * tpc), dim3(256), 0, st, *x, w, dlogits, N, C, tpc, dyhat, stat_part, wslab);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int sum_partials(const float* part, int n, int stride, int elems, float* out, hipStream_t st) {
  PN_CHECK_ARG(part && out && n > 0 && elems > 0, "sum_partials: bad arguments");
  hipLaunchKernelGGL(sum_partials_kernel, dim3(elems), dim3(64), 0, st, part, n, stride, elems, out);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

}  // namespace pn
