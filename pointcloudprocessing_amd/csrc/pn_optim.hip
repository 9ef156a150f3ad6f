// Optimizer, scalar losses and the 3x3 transform fold (gfx950).  Reference semantics:
// keras.optimizers.Adam + ExponentialDecay (point_cloud_analysis/pointnet_train.py:310-319),
// keras.losses.MeanSquaredError on the input transform (pointnet_train.py:339),
// the orthogonality regulariser 1e-3 * l2_loss(I - R R^T) (pointnet/PointNet.py:447-451).
#include "pn_common.h"
#include "pn_loss_bodies.h"

namespace pn {

// state[0] = iterations (as float bits of an int), hyper[0] = alpha for this step.  One thread; keeps the
// whole schedule on the device so a captured hipGraph replays with the right learning rate.
__global__ void adam_schedule_kernel(int* __restrict__ iterations, float lr0, float decay_rate, float decay_steps, float beta1,
                                     float beta2, float* __restrict__ alpha_out, float* __restrict__ lr_out) {
  const int it = *iterations;           // optimizer.iterations before this update
  const double lr = (double)lr0 * pow((double)decay_rate, (double)it / (double)decay_steps);
  const double t = (double)(it + 1);
  const double alpha = lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t));
  *alpha_out = (float)alpha;
  if (lr_out) *lr_out = (float)lr;
  *iterations = it + 1;
}

// Schedule + update in one launch.  scratch: [0] step size and [1] learning rate for the CURRENT optimizer.iterations (written by
// adam_prepare at construction / after a restore, then kept current by this kernel), [2] ticket counter (uint32, zero).
// The block that draws the last ticket -- every block has used scratch[0] by then -- advances the counter and evaluates the
// schedule for the next step, so the fp64 pow()s are paid by one thread per step, off everyone else's path.
__device__ __forceinline__ void adam_schedule_eval(int it, double lr0, double decay_rate, double decay_steps, double beta1, double beta2,
                                                   float* __restrict__ scratch) {
  const double lr = lr0 * pow(decay_rate, (double)it / decay_steps);
  const double t = (double)(it + 1);
  scratch[0] = (float)(lr * sqrt(1.0 - pow(beta2, t)) / (1.0 - pow(beta1, t)));
  scratch[1] = (float)lr;
}
__global__ void adam_prepare_kernel(const int* __restrict__ iterations, double lr0, double decay_rate, double decay_steps, double beta1,
                                    double beta2, float* __restrict__ scratch) {
  adam_schedule_eval(*iterations, lr0, decay_rate, decay_steps, beta1, beta2, scratch);
  reinterpret_cast<unsigned*>(scratch)[2] = 0u;
}
__global__ __launch_bounds__(1024) void adam_fused_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long long n, int* __restrict__ iterations, double lr0,
                                                         double decay_rate, double decay_steps, double beta1, double beta2, float eps,
                                                         float grad_scale, float* __restrict__ scratch, int vec) {
  const float alpha = scratch[0];
  const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);      // keras: the Python float 1 - beta, cast to fp32
  auto upd = [&](float& pi, float gi0, float& mi0, float& vi0) {
    const float gi = gi0 * grad_scale;
    const float mi = mi0 + (gi - mi0) * omb1;
    const float vi = vi0 + (gi * gi - vi0) * omb2;
    mi0 = mi;
    vi0 = vi;
    pi -= mi * alpha / (sqrtf(vi) + eps);
  };
  // 28 bytes of traffic per parameter and nothing else: 16-byte accesses, two quads per thread in flight (the four arrays are
  // 16-byte aligned: checked by the launcher), a scalar tail for n % 4
  const long long n4 = vec ? (n >> 2) : 0;
  const long long stride = (long long)gridDim.x * 1024;
  float4* p4 = reinterpret_cast<float4*>(p); const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m); float4* v4 = reinterpret_cast<float4*>(v);
  for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += 2 * stride) {
    const long long j = i + stride;
    const bool two = j < n4;
    const long long jj = two ? j : i;
    float4 pa = p4[i], ga = g4[i], ma = m4[i], va = v4[i];
    float4 pb = p4[jj], gb = g4[jj], mb = m4[jj], vb = v4[jj];
    upd(pa.x, ga.x, ma.x, va.x); upd(pa.y, ga.y, ma.y, va.y); upd(pa.z, ga.z, ma.z, va.z); upd(pa.w, ga.w, ma.w, va.w);
    p4[i] = pa; m4[i] = ma; v4[i] = va;
    if (two) {
      upd(pb.x, gb.x, mb.x, vb.x); upd(pb.y, gb.y, mb.y, vb.y); upd(pb.z, gb.z, mb.z, vb.z); upd(pb.w, gb.w, mb.w, vb.w);
      p4[j] = pb; m4[j] = mb; v4[j] = vb;
    }
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * 1024 + threadIdx.x; i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(pi, g[i], mi, vi);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
  __syncthreads();                        // every thread of this block has read scratch[0]
  if (threadIdx.x == 0) {
    unsigned* ticket = reinterpret_cast<unsigned*>(scratch + 2);
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int it = *iterations + 1;
      *iterations = it;
      adam_schedule_eval(it, lr0, decay_rate, decay_steps, beta1, beta2, scratch);
    }
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, const float* __restrict__ alpha_p, float beta1,
                                                   float beta2, float eps, float grad_scale) {
  const float alpha = *alpha_p;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * grad_scale;
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
    const float vi = v[i] + (gi * gi - v[i]) * (1.f - beta2);
    m[i] = mi;
    v[i] = vi;
    p[i] -= mi * alpha / (sqrtf(vi) + eps);
  }
}

// MeanSquaredError over (B,3,3) and its gradient (weight w folded in): out loss_sum = sum (R-T)^2
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ R, const float* __restrict__ T, int n, float gscale,
                                                  float* __restrict__ dR, float* __restrict__ loss_sum) {
  __shared__ float red[16];
  mse_body(R, T, n, gscale, dR, loss_sum, red);
}

// Orthogonality regulariser per cloud: E = I - R R^T; loss += c/2 * sum E^2; dR += -2c E R.   block per cloud
__global__ __launch_bounds__(256) void orth_reg_kernel(const float* __restrict__ R, int K, float c, float* __restrict__ dR,
                                                       float* __restrict__ loss_part) {
  extern __shared__ float sm[];
  float* Rs = sm;            // K*K
  float* Es = sm + K * K;    // K*K
  __shared__ float red[256];
  const int b = blockIdx.x;
  const float* Rb = R + (long long)b * K * K;
  for (int i = threadIdx.x; i < K * K; i += 256) Rs[i] = Rb[i];
  __syncthreads();
  float s = 0.f;
  for (int ij = threadIdx.x; ij < K * K; ij += 256) {
    const int i = ij / K, j = ij % K;
    float d = 0.f;
    for (int k = 0; k < K; ++k) d = fmaf(Rs[i * K + k], Rs[j * K + k], d);
    const float e = (i == j ? 1.f : 0.f) - d;
    Es[ij] = e;
    s = fmaf(e, e, s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f;
    for (int i = 0; i < 256; ++i) a += red[i];
    if (loss_part) loss_part[b] = 0.5f * c * a;
  }
  if (dR) {
    float* dRb = dR + (long long)b * K * K;
    for (int ij = threadIdx.x; ij < K * K; ij += 256) {
      const int i = ij / K, j = ij % K;
      float d = 0.f;
      for (int k = 0; k < K; ++k) d = fmaf(Es[i * K + k], Rs[k * K + j], d);
      dRb[ij] += -2.f * c * d;
    }
  }
}

// Weff[b] (3,C) = R[b] (3,3) @ W (3,C)           -- tf.matmul(pc, R) folded into the first kernel
__global__ __launch_bounds__(256) void fold3_fwd_kernel(const float* __restrict__ R, const float* __restrict__ W, int C,
                                                        float* __restrict__ Weff, float* __restrict__ R_copy) {
  const int b = blockIdx.x;
  if (R_copy && threadIdx.x < 9) R_copy[(long long)b * 9 + threadIdx.x] = R[(long long)b * 9 + threadIdx.x];   // the model's third output
  for (int t = threadIdx.x; t < 3 * C; t += 256) {
    const int i = t / C, c = t % C;
    const float* r = R + (long long)b * 9 + i * 3;
    Weff[(long long)b * 3 * C + t] = fmaf(r[2], W[2 * C + c], fmaf(r[1], W[C + c], r[0] * W[c]));
  }
}
// One launch for both gradients of Weff[b] = R[b] W:   blocks [0, B): dR[b][i][k] = sum_c dWeff[b][i][c] W[k][c]  (first writer of
// dR: later terms add to it);   blocks [B, ...): dW[k][c] = sum_b sum_i R[b][i][k] dWeff[b][i][c]
__global__ __launch_bounds__(256) void fold3_bwd_kernel(const float* __restrict__ dWeff, const float* __restrict__ R,
                                                        const float* __restrict__ W, int B, int C, float* __restrict__ dR,
                                                        float* __restrict__ dW) {
  if ((int)blockIdx.x < B) {
    if (!dR) return;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int ik = wave; ik < 9; ik += 4) {
      const int i = ik / 3, k = ik % 3;
      float s = 0.f;
      for (int c = lane; c < C; c += 64) s = fmaf(dWeff[((long long)b * 3 + i) * C + c], W[k * C + c], s);
      s = wave_sum(s);
      if (lane == 0) dR[(long long)b * 9 + ik] = s;
    }
    return;
  }
  if (!dW) return;
  const int t = ((int)blockIdx.x - B) * 256 + threadIdx.x;
  if (t >= 3 * C) return;
  const int k = t / C, c = t % C;
  float s = 0.f;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < 3; ++i) s = fmaf(R[(long long)b * 9 + i * 3 + k], dWeff[((long long)b * 3 + i) * C + c], s);
  dW[t] = s;
}

// The same two gradients straight from the per-tile slabs conv3_wgrad leaves (slabs[t][i][c], t = cloud * tpc + tile): one launch
// instead of a slab reduction followed by fold3_bwd.  Blocks [0, B): cloud b's dWeff = sum of its tpc slabs (LDS), then dR[b] as above.
// Blocks [B, B + 3C): one per dW element (k, c): thread <-> slab, dW[k][c] = sum_t sum_i R[cloud(t)][i][k] slabs[t][i][c], combined in a
// fixed order (wave butterflies, then the four waves in turn).
__global__ __launch_bounds__(256) void fold3_bwd_slabs_kernel(const float* __restrict__ slabs, int n_slabs, int tpc, const float* __restrict__ R,
                                                              const float* __restrict__ W, int B, int C, float* __restrict__ dR,
                                                              float* __restrict__ dW) {
  __shared__ float dws[3 * 256];
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x < B) {
    if (!dR) return;
    const int b = blockIdx.x;
    for (int t = threadIdx.x; t < 3 * C; t += 256) {
      float s = 0.f;
      for (int t0 = 0; t0 < tpc; t0 += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = slabs[((long long)b * tpc + min(t0 + u, tpc - 1)) * 3 * C + t];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += (t0 + u < tpc) ? v[u] : 0.f;
      }
      dws[t] = s;
    }
    __syncthreads();
    for (int ik = wave; ik < 9; ik += 4) {
      const int i = ik / 3, k = ik % 3;
      float s = 0.f;
      for (int c = lane; c < C; c += 64) s = fmaf(dws[i * C + c], W[k * C + c], s);
      s = wave_sum(s);
      if (lane == 0) dR[(long long)b * 9 + ik] = s;
    }
    return;
  }
  if (!dW) return;
  const int o = (int)blockIdx.x - B, k = o / C, c = o - k * C;
  float acc = 0.f;
  for (int t = threadIdx.x; t < n_slabs; t += 256) {
    const int b = t / tpc;
    const float* sl = slabs + (long long)t * 3 * C + c;
    const float* r = R + (long long)b * 9 + k;
    acc = fmaf(r[6], sl[2 * C], fmaf(r[3], sl[C], fmaf(r[0], sl[0], acc)));
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) dW[o] = (red[0] + red[1]) + (red[2] + red[3]);
}

// y[i] = a*x[i] + y[i] / plain helpers
__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x, float a, float* __restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = fmaf(a, x[i], y[i]);
}

// seg_l1 global-feature bias gradient: dgb[b][c] = ca[c]*sum_n dyhat + cb[c]*sum_n z + N*cc[c]
__global__ __launch_bounds__(256) void cloud_bias_grad_kernel(const float* __restrict__ bwd_part, const float* __restrict__ fwd_part,
                                                              int tiles_per_cloud, int N, int C, const float* __restrict__ ca,
                                                              const float* __restrict__ cb, const float* __restrict__ cc,
                                                              float* __restrict__ dgb) {
  const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  float sd = 0.f, sz = 0.f;
  for (int t = 0; t < tiles_per_cloud; ++t) {
    const long long o = ((long long)b * tiles_per_cloud + t) * 2 * C + c;
    sd += bwd_part[o];
    sz += fwd_part[o];
  }
  dgb[(long long)b * C + c] = fmaf(ca[c], sd, fmaf(cb[c], sz, (float)N * cc[c]));
}

__global__ void fill_eye3_kernel(float* __restrict__ out, int B) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B * 9) out[i] = ((i % 9) % 4 == 0) ? 1.f : 0.f;
}
int fill_eye3(float* out, int B, hipStream_t st) {
  hipLaunchKernelGGL(fill_eye3_kernel, dim3(cdiv(B * 9, 256)), dim3(256), 0, st, out, B);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int adam_schedule(int* iterations, float lr0, float decay_rate, float decay_steps, float beta1, float beta2, float* alpha, float* lr,
                  hipStream_t st) {
  hipLaunchKernelGGL(adam_schedule_kernel, dim3(1), dim3(1), 0, st, iterations, lr0, decay_rate, decay_steps, beta1, beta2, alpha, lr);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
int adam(float* p, const float* g, float* m, float* v, long long n, const float* alpha, float beta1, float beta2, float eps,
         float grad_scale, hipStream_t st) {
  PN_CHECK_ARG(p && g && m && v && alpha && n > 0, "pn_adam: bad arguments");
  const long long blocks = cdivll(n, 256);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, p, g, m, v, n, alpha, beta1,
                     beta2, eps, grad_scale);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
int adam_prepare(const int* iterations, double lr0, double decay_rate, double decay_steps, double beta1, double beta2, float* scratch,
                 hipStream_t st) {
  PN_CHECK_ARG(iterations && scratch, "pn_adam_prepare: null pointer");
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, st, iterations, lr0, decay_rate, decay_steps, beta1, beta2, scratch);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
int adam_fused(float* p, const float* g, float* m, float* v, long long n, int* iterations, double lr0, double decay_rate, double decay_steps,
               double beta1, double beta2, double eps, float grad_scale, float* scratch, hipStream_t st) {
  PN_CHECK_ARG(p && g && m && v && iterations && scratch && n > 0, "pn_adam_step: bad arguments");
  // 256 blocks of 1024 threads: one per CU at full occupancy, and only 256 tickets on the counter (a same-address atomic costs
  // ~11 ns: 2048 blocks spent 22 us of a 43 us launch queueing on it)
  const long long blocks = cdivll(n, 1024);
  const int vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  hipLaunchKernelGGL(adam_fused_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(1024), 0, st, p, g, m, v, n, iterations, lr0,
                     decay_rate, decay_steps, beta1, beta2, (float)eps, grad_scale, scratch, vec);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// out = a + b (either may be NULL = zeros)
__global__ __launch_bounds__(256) void add2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (a ? a[i] : 0.f) + (b ? b[i] : 0.f);
}
int add2(const float* a, const float* b, float* out, long long n, hipStream_t st) {
  PN_CHECK_ARG(out && n > 0, "add2: bad arguments");
  hipLaunchKernelGGL(add2_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, st, a, b, out, n);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int mse(const float* R, const float* T, int n, float gscale, float* dR, float* loss_sum, hipStream_t st) {
  hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(256), 0, st, R, T, n, gscale, dR, loss_sum);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
int orth_reg(const float* R, int B, int K, float c, float* dR, float* loss_part, hipStream_t st) {
  PN_CHECK_ARG(K <= 64, "orth_reg: K must be <= 64");
  hipLaunchKernelGGL(orth_reg_kernel, dim3(B), dim3(256), (size_t)2 * K * K * sizeof(float), st, R, K, c, dR, loss_part);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
int fold3_fwd(const float* R, const float* W, int B, int C, float* Weff, hipStream_t st, float* R_copy) {
  hipLaunchKernelGGL(fold3_fwd_kernel, dim3(B), dim3(256), 0, st, R, W, C, Weff, R_copy);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
int fold3_bwd(const float* dWeff, const float* R, const float* W, int B, int C, float* dR, float* dW, hipStream_t st) {
  if (!dR && !dW) return PN_OK;
  hipLaunchKernelGGL(fold3_bwd_kernel, dim3(B + cdiv(3 * C, 256)), dim3(256), 0, st, dWeff, R, W, B, C, dR, dW);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
int fold3_bwd_slabs(const float* slabs, int n_slabs, int tpc, const float* R, const float* W, int B, int C, float* dR, float* dW,
                    hipStream_t st) {
  PN_CHECK_ARG(slabs && R && W && n_slabs == B * tpc && C > 0 && 3 * C <= 3 * 256, "fold3_bwd_slabs: bad arguments");
  if (!dR && !dW) return PN_OK;
  hipLaunchKernelGGL(fold3_bwd_slabs_kernel, dim3(B + 3 * C), dim3(256), 0, st, slabs, n_slabs, tpc, R, W, B, C, dR, dW);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
// tf.debugging.check_numerics (PointNet.py:199-288, `debugging: true`): *count += number of NaN / Inf elements of x
__global__ __launch_bounds__(256) void count_nonfinite_kernel(const float* __restrict__ x, long long n, int* __restrict__ count, int h16) {
  int bad = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    bad += (__builtin_isfinite(act_load(x, i, h16)) ? 0 : 1);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);
}
int count_nonfinite(const float* x, long long n, int* count, hipStream_t st, int h16) {
  PN_CHECK_ARG(x && count && n >= 0, "pn_count_nonfinite: bad arguments");
  if (n == 0) return PN_OK;
  const long long blocks = cdivll(n, 256 * 8);
  hipLaunchKernelGGL(count_nonfinite_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks < 2048 ? blocks : 2048))), dim3(256), 0, st, x, n, count, h16);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int axpy(const float* x, float a, float* y, long long n, hipStream_t st) {
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, st, x, a, y, n);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
int cloud_bias_grad(const float* bwd_part, const float* fwd_part, int B, int tpc, int N, int C, const float* ca, const float* cb,
                    const float* cc, float* dgb, hipStream_t st) {
  hipLaunchKernelGGL(cloud_bias_grad_kernel, dim3(cdiv(C, 256), B), dim3(256), 0, st, bwd_part, fwd_part, tpc, N, C, ca, cb, cc, dgb);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

}  // namespace pn
