// What a step does before its first layer, none of it depending on anything computed in the step:
//   * PointCloudNormalization (pointnet/PointNet.py:691-706),
//   * the fragment-ordered bf16 copies of the three 128->1024 kernels for the panel kernel (pn_panel.hip),
//   * the inverted-dropout keep masks of the classification head (PointNet.py:255,260),
//   * the zero fill of the gradient buffer and of the dense layers' arrival counters,
//   * bf16 copies (as they are / transposed) of the other per-point layers' kernels for the row GEMMs (pn_gemm.hip: CopyStage).
// The first four exist as entry points of their own (pn_normalize, pn_weights_prep, pn_dropout_masks, zero_fill) and, for the model
// plan, all five are ONE launch whose workgroups take the roles side by side (fwd_prologue): four dependent launches of ~5 us each,
// most of it launch latency, became one.  Every body below is written for any workgroup size.
#include "pn_common.h"
#include "pn_internal.h"

namespace pn {

typedef __attribute__((ext_vector_type(8))) __bf16 pr_bf16x8;

// ------------------------------------------------------------------------------------------------------
// PointCloudNormalization (reference: pointnet/PointNet.py:691-706).  One workgroup per cloud:
// pass 1 centroid (wave shuffle + LDS), pass 2 max radius, pass 3 write.  12 B/point in, 12 B/point out;
// the cloud (<= a few MB) stays in L2 between passes.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void normalize_body(const float* __restrict__ xyz, int N, float* __restrict__ out, float* __restrict__ centroid,
                                               float* __restrict__ scale, int b, float (*red)[3], float* bc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt = blockDim.x, nw = nt >> 6;
  const float* p = xyz + (long long)b * N * 3;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int i = tid; i < N; i += nt) {
    sx += p[3 * i]; sy += p[3 * i + 1]; sz += p[3 * i + 2];
  }
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
  if (lane == 0) { red[wave][0] = sx; red[wave][1] = sy; red[wave][2] = sz; }
  __syncthreads();
  if (tid < 3) {
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += red[w][tid];
    bc[tid] = s / (float)N;
  }
  __syncthreads();
  const float cx = bc[0], cy = bc[1], cz = bc[2];
  float md = 0.f;
  for (int i = tid; i < N; i += nt) {
    const float dx = p[3 * i] - cx, dy = p[3 * i + 1] - cy, dz = p[3 * i + 2] - cz;
    md = fmaxf(md, sqrtf(dx * dx + dy * dy + dz * dz));
  }
  md = wave_max(md);
  __syncthreads();
  if (lane == 0) red[wave][0] = md;
  __syncthreads();
  if (tid == 0) {
    float m = 0.f;
    for (int w = 0; w < nw; ++w) m = fmaxf(m, red[w][0]);
    bc[3] = fmaxf(m, 1e-7f);
  }
  __syncthreads();
  const float sc = bc[3];
  float* o = out + (long long)b * N * 3;
  for (int i = tid; i < N; i += nt) {
    o[3 * i] = (p[3 * i] - cx) / sc;
    o[3 * i + 1] = (p[3 * i + 1] - cy) / sc;
    o[3 * i + 2] = (p[3 * i + 2] - cz) / sc;
  }
  if (tid == 0) {
    if (centroid) { centroid[3 * b] = cx; centroid[3 * b + 1] = cy; centroid[3 * b + 2] = cz; }
    if (scale) scale[b] = sc;
  }
}
__global__ __launch_bounds__(1024) void normalize_kernel(const float* __restrict__ xyz, int N, float* __restrict__ out,
                                                         float* __restrict__ centroid, float* __restrict__ scale) {
  __shared__ float red[16][3];
  __shared__ float bc[4];
  normalize_body(xyz, N, out, centroid, scale, blockIdx.x, red, bc);
}
int normalize(const float* xyz, int B, int N, float* out, float* centroid, float* scale, hipStream_t st) {
  PN_CHECK_ARG(xyz && out, "pn_normalize: null pointer");
  PN_CHECK_ARG(B > 0 && N > 0, "pn_normalize: B and N must be positive (B=%d N=%d)", B, N);
  hipLaunchKernelGGL(normalize_kernel, dim3(B), dim3(1024), 0, st, xyz, N, out, centroid, scale);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// ---- weight preparation: fragment-ordered bf16 copies ------------------------------------------------------------------------
//   Wf_hi[((cb * K/16 + ks) * 64 + lane) * 8 + j] = bf16(s_c * W[k][c]),   c = cb * 32 + (lane & 31),  k = ks * 16 + (lane >> 5) * 8 + j
//   Wf_lo[...] = bf16(s_c * W[k][c] - hi)        s_c = -1 where sgn[c] < 0 (sgn may be gamma itself), else +1: the accumulators of the
//   panel kernel then hold sgn * z and max(sgn * z) needs no multiply (a sign flip commutes with the rounding, so this is exact).
struct Prep3Args {
  const float* sgn[3];
  const float* w[3];
  __bf16* hi[3];
  __bf16* lo[3];
  int K[3], C[3];
};
// one 16-byte chunk (8 consecutive k of one channel) of layer z
__device__ __forceinline__ void prep3_chunk(const Prep3Args& a, int z, long long chunk) {
  const float* __restrict__ w = a.w[z];
  if (!w) return;
  const int K = a.K[z], C = a.C[z], KS = K / 16;
  if (chunk >= (long long)C * K / 8) return;
  const int lane = (int)(chunk & 63);
  const int ks = (int)((chunk >> 6) % KS), cb = (int)((chunk >> 6) / KS);
  const int c = cb * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
  const float sg = (a.sgn[z] && a.sgn[z][c] < 0.f) ? -1.f : 1.f;
  pr_bf16x8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = sg * w[(long long)(k0 + j) * C + c];
    h[j] = (__bf16)v;
    l[j] = (__bf16)(v - (float)h[j]);
  }
  *reinterpret_cast<pr_bf16x8*>(a.hi[z] + chunk * 8) = h;
  if (a.lo[z]) *reinterpret_cast<pr_bf16x8*>(a.lo[z] + chunk * 8) = l;
}
__global__ __launch_bounds__(256) void weights_prep3_kernel(const Prep3Args a) {
  prep3_chunk(a, blockIdx.y, (long long)blockIdx.x * 256 + threadIdx.x);
}
static int make_prep3(const float* const* w, const float* const* sgn, const int* K, const int* C, void* const* hi, void* const* lo,
                      Prep3Args& a, long long& mx) {
  mx = 1;
  for (int i = 0; i < 3; ++i) {
    a.w[i] = w[i]; a.sgn[i] = sgn ? sgn[i] : nullptr; a.K[i] = K[i]; a.C[i] = C[i];
    a.hi[i] = reinterpret_cast<__bf16*>(hi[i]); a.lo[i] = reinterpret_cast<__bf16*>(lo[i]);
    PN_CHECK_ARG(!w[i] || (hi[i] && K[i] > 0 && K[i] % 16 == 0 && C[i] > 0 && C[i] % 32 == 0), "weights_prep3: bad arguments");
    if (w[i] && (long long)K[i] * C[i] / 8 > mx) mx = (long long)K[i] * C[i] / 8;
  }
  return PN_OK;
}
int weights_prep(const float* w, const float* sgn, int K, int C, void* hi, void* lo, hipStream_t st) {
  PN_CHECK_ARG(w && hi, "pn_weights_prep: null pointer");
  PN_CHECK_ARG(K > 0 && K % 16 == 0 && C > 0 && C % 32 == 0, "pn_weights_prep: K must be a multiple of 16 and C of 32 (K=%d C=%d)", K, C);
  const float* ws[3] = {w, nullptr, nullptr};
  const float* sg[3] = {sgn, nullptr, nullptr};
  const int Ks[3] = {K, 0, 0}, Cs[3] = {C, 0, 0};
  void* his[3] = {hi, nullptr, nullptr};
  void* los[3] = {lo, nullptr, nullptr};
  Prep3Args a;
  long long mx;
  PN_TRY(make_prep3(ws, sg, Ks, Cs, his, los, a, mx));
  hipLaunchKernelGGL(weights_prep3_kernel, dim3((unsigned)cdivll(mx, 256), 1), dim3(256), 0, st, a);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// Inverted-dropout keep masks for the two classification-head layers from a counter-based generator (no state but a step
// counter on the device, so a captured graph draws fresh masks at every replay): keep = u(seed, step, index) >= rate.
// One workgroup; the counter moves once everyone has read it.  (keras draws from TF's stateful generator; there is no stream
// to be bit-compatible with, so masks are an INPUT of the parity tests.)
__device__ __forceinline__ unsigned mix32(unsigned x) {      // murmur3 finaliser
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ void dropout_body(unsigned char* __restrict__ k1, long long n1, unsigned char* __restrict__ k2, long long n2,
                                             float rate, unsigned seed_lo, unsigned seed_hi, unsigned* __restrict__ step) {
  const unsigned st = *step;
  __syncthreads();
  const unsigned thr = (unsigned)fminf(rate * 16777216.f, 16777216.f);     // compare on 24 bits
  const unsigned base = mix32(seed_lo ^ mix32(seed_hi + 0x9e3779b9u * (st + 1u)));
  for (long long i = threadIdx.x; i < n1 + n2; i += blockDim.x) {
    const unsigned h = mix32(base + 0x9e3779b9u * (unsigned)i) ^ mix32(seed_hi ^ (unsigned)(i >> 32) ^ (unsigned)i * 0x7feb352du);
    const unsigned char keep = ((h >> 8) >= thr) ? 1 : 0;
    if (i < n1) k1[i] = keep; else k2[i - n1] = keep;
  }
  if (threadIdx.x == 0) *step = st + 1u;
}
__global__ __launch_bounds__(1024) void dropout_masks_kernel(unsigned char* __restrict__ k1, long long n1, unsigned char* __restrict__ k2,
                                                             long long n2, float rate, unsigned seed_lo, unsigned seed_hi,
                                                             unsigned* __restrict__ step) {
  dropout_body(k1, n1, k2, n2, rate, seed_lo, seed_hi, step);
}
int dropout_masks(unsigned char* k1, long long n1, unsigned char* k2, long long n2, float rate, unsigned long long seed, unsigned* step,
                  hipStream_t st) {
  PN_CHECK_ARG(step && n1 >= 0 && n2 >= 0 && (n1 == 0 || k1) && (n2 == 0 || k2) && rate >= 0.f && rate < 1.f, "pn_dropout_masks: bad arguments");
  hipLaunchKernelGGL(dropout_masks_kernel, dim3(1), dim3(1024), 0, st, k1, n1, k2, n2, rate, (unsigned)seed, (unsigned)(seed >> 32), step);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// Zero fill as an ordinary kernel.  hipMemsetAsync is avoided on purpose: captured into a hipGraph (ROCm 7.2) the
// 16 MiB memset node of the gradient buffer replayed with a garbage fill pattern once another model had launched
// work between two replays (tools/graph_hunt.py); a kernel node has no such state.
// workgroup `blk` of `nblk` clears its share of p[0..n)
__device__ __forceinline__ void zero_body(float* __restrict__ p, long long n, int blk, int nblk) {
  const long long n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  const long long stride = (long long)nblk * blockDim.x;
  for (long long i = (long long)blk * blockDim.x + threadIdx.x; i < n4; i += stride) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (blk == 0 && threadIdx.x < (n & 3)) p[(n4 << 2) + threadIdx.x] = 0.f;
}
__global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ p, long long n, float* __restrict__ p2, int n2) {
  if (blockIdx.x == 0 && p2)
    for (int i = threadIdx.x; i < n2; i += 256) p2[i] = 0.f;
  zero_body(p, n, blockIdx.x, gridDim.x);
}
int zero_fill(float* p, long long n, hipStream_t st) { return zero_fill2(p, n, nullptr, 0, st); }
int zero_fill2(float* p, long long n, float* p2, int n2, hipStream_t st) {
  PN_CHECK_ARG(p && n >= 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0, "zero_fill: null or unaligned buffer");
  if (n == 0 && !p2) return PN_OK;
  const long long blocks = cdivll(cdivll(n, 4), 256);
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks < 2048 ? blocks : 2048))), dim3(256), 0, st, p, n, p2, n2);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// ---- the roles in one launch -------------------------------------------------------------------------------------------------------
// bf16 copies of the per-point layers' kernels for the row GEMMs (pn_gemm.hip: CopyStage): `nat` = the kernel as it is, (K, C) with
// the C outputs contiguous -- what the data-gradient GEMM stages (its contraction runs over the outputs); `tr` = transposed, (C, K)
// with the K inputs contiguous -- what the forward GEMM stages.  Same round-to-nearest-even the GEMMs apply when they convert the
// fp32 kernel themselves, so results do not change; a workgroup no longer converts 16-64 KB of kernel per launch.
struct WCopyJobs {
  const float* w[PN_WCOPY_MAX];
  unsigned short* nat[PN_WCOPY_MAX];
  unsigned short* tr[PN_WCOPY_MAX];
  int K[PN_WCOPY_MAX], C[PN_WCOPY_MAX];
  int end[PN_WCOPY_MAX];          // running element count
  unsigned char frag[PN_WCOPY_MAX];   // layout of the transposed copy (WCopyDesc.frag)
  int n;
};
// work item = eight consecutive outputs of one of the two copies: items [0, E/8) write `nat` (8 consecutive elements of the kernel:
// two 16-byte loads, one 16-byte store), items [E/8, E/4) write `tr` (8 consecutive k of one output column c: eight loads that are
// coalesced across the lanes' columns, one 16-byte store); E = elements of all jobs (every K and C is a multiple of 8)
__device__ __forceinline__ void wcopy_body(const WCopyJobs& j, int blk) {
  const int total8 = (j.n ? j.end[j.n - 1] : 0) / 8;
  const int item = blk * (int)blockDim.x + (int)threadIdx.x;
  if (item >= 2 * total8) return;
  const bool tr = item >= total8;
  const int i8 = tr ? item - total8 : item;
  int q = 0;
  while (q + 1 < j.n && i8 * 8 >= j.end[q]) ++q;
  const int local8 = i8 - (q ? j.end[q - 1] : 0) / 8;
  const int K = j.K[q], C = j.C[q];
  const float* __restrict__ w = j.w[q];
  float v[8];
  if (!tr) {
    const float4 a0 = *reinterpret_cast<const float4*>(w + (long long)local8 * 8), a1 = *reinterpret_cast<const float4*>(w + (long long)local8 * 8 + 4);
    v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
  } else {
    const int c = local8 % C, k0 = (local8 / C) * 8;       // lanes walk the columns: each of the eight loads is coalesced
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w[(long long)(k0 + e) * C + c];
  }
  pr_bf16x8 h;
#pragma unroll
  for (int e = 0; e < 8; ++e) h[e] = (__bf16)v[e];
  if (!tr) {
    *reinterpret_cast<pr_bf16x8*>(j.nat[q] + (long long)local8 * 8) = h;
  } else {
    const int c = local8 % C, k0 = (local8 / C) * 8;
    const long long at = j.frag[q] ? ((((long long)(c >> 5) * (K >> 4) + (k0 >> 4)) * 64 + 32 * ((k0 >> 3) & 1) + (c & 31)) * 8) : ((long long)c * K + k0);
    *reinterpret_cast<pr_bf16x8*>(j.tr[q] + at) = h;
  }
}
// the same copies as a launch of their own (op-level ABI: pn_weights_copy16; the model plan makes them in its first launch)
__global__ __launch_bounds__(1024) void weights_copy16_kernel(const WCopyJobs j) { wcopy_body(j, blockIdx.x); }
int weights_copy16(const float* w, int K, int C, void* nat, void* tr, hipStream_t st) {
  PN_CHECK_ARG(w && nat && tr && K > 0 && C > 0 && K % 8 == 0 && C % 8 == 0, "pn_weights_copy16: w, both copies, K and C multiples of 8");
  WCopyJobs j;
  memset(&j, 0, sizeof(j));
  j.w[0] = w; j.nat[0] = reinterpret_cast<unsigned short*>(nat); j.tr[0] = reinterpret_cast<unsigned short*>(tr);
  j.K[0] = K; j.C[0] = C; j.end[0] = K * C; j.n = 1;
  hipLaunchKernelGGL(weights_copy16_kernel, dim3(cdiv(K * C / 4, 1024)), dim3(1024), 0, st, j);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
// coefficients of a layer that normalises with its moving statistics: the arithmetic of pn_bn_finalize with use_batch = 0
struct FrozenBnJobs {
  FrozenBnDesc j[PN_FROZEN_MAX];
  int n;
  float eps;
};
__device__ __forceinline__ void frozen_bn_body(const FrozenBnJobs& f, int q) {
  const FrozenBnDesc& d = f.j[q];
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    const float mean = d.mm[c], var = d.mv[c];
    const float invstd = 1.0f / sqrtf(var + f.eps);
    const float sc = d.gamma[c] * invstd;
    if (d.mean) d.mean[c] = mean;
    if (d.invstd) d.invstd[c] = invstd;
    d.scale[c] = sc;
    d.shift[c] = d.beta[c] - mean * sc;
  }
}
struct PrologueArgs {
  // normalisation: workgroups [0, B)
  const float* xyz; int B, N; float *out, *centroid, *scale;
  // kernel copies: the next n_prep workgroups (1024 chunks each, the three layers laid end to end)
  Prep3Args prep; int n_prep; long long prep_chunks;       // chunks per layer (the largest)
  // counters: cleared by the first kernel-copy workgroup
  unsigned* zero_u; int zero_u_n;
  // gradient buffer: the next n_zero workgroups (0: not this launch's job)
  float* grads; long long n_grads; int n_zero;
  // dropout masks: one more workgroup (0: not this launch's job)
  int n_drop; unsigned char *k1, *k2; long long n1, n2; float rate; unsigned seed_lo, seed_hi; unsigned* step;
  // bf16 kernel copies for the row GEMMs: the next n_wcopy workgroups (1024 items of eight outputs each)
  int n_wcopy; WCopyJobs wc;
  // frozen BatchNormalization coefficients: one workgroup per layer
  FrozenBnJobs fz;
};
__global__ __launch_bounds__(1024) void fwd_prologue_kernel(const PrologueArgs a) {
  __shared__ float red[16][3];
  __shared__ float bc[4];
  int bx = blockIdx.x;
  if (bx < a.B) { normalize_body(a.xyz, a.N, a.out, a.centroid, a.scale, bx, red, bc); return; }
  bx -= a.B;
  if (bx < a.n_prep) {
    if (bx == 0 && a.zero_u)
      for (int i = threadIdx.x; i < a.zero_u_n; i += 1024) a.zero_u[i] = 0u;
    const long long lin = (long long)bx * 1024 + threadIdx.x;            // layer-major: z = lin / prep_chunks
    const int z = (int)(lin / a.prep_chunks);
    if (z < 3) prep3_chunk(a.prep, z, lin - (long long)z * a.prep_chunks);
    return;
  }
  bx -= a.n_prep;
  if (bx < a.n_zero) { zero_body(a.grads, a.n_grads, bx, a.n_zero); return; }
  bx -= a.n_zero;
  if (bx < a.n_drop) { dropout_body(a.k1, a.n1, a.k2, a.n2, a.rate, a.seed_lo, a.seed_hi, a.step); return; }
  bx -= a.n_drop;
  if (bx < a.n_wcopy) { wcopy_body(a.wc, bx); return; }
  bx -= a.n_wcopy;
  if (bx < a.fz.n) frozen_bn_body(a.fz, bx);
}
int fwd_prologue(const float* xyz, int B, int N, float* out, float* centroid, float* scale, const float* const* w, const float* const* sgn,
                 const int* K, const int* C, void* const* hi, void* const* lo, unsigned* zero_u, int zero_u_n, float* grads, long long n_grads,
                 unsigned char* k1, long long n1, unsigned char* k2, long long n2, float rate, unsigned long long seed, unsigned* step,
                 const WCopyDesc* wcopies, int n_wcopies, const FrozenBnDesc* frozen, int n_frozen, float bn_eps, hipStream_t st) {
  PN_CHECK_ARG(n_frozen >= 0 && n_frozen <= PN_FROZEN_MAX && (n_frozen == 0 || frozen), "fwd_prologue: too many frozen layers");
  PN_CHECK_ARG(xyz && out && B > 0 && N > 0, "fwd_prologue: bad cloud arguments");
  PN_CHECK_ARG(n_wcopies >= 0 && n_wcopies <= PN_WCOPY_MAX && (n_wcopies == 0 || wcopies), "fwd_prologue: too many kernel copies");
  PN_CHECK_ARG(!grads || (n_grads >= 0 && (reinterpret_cast<uintptr_t>(grads) & 15) == 0), "fwd_prologue: unaligned gradient buffer");
  PN_CHECK_ARG(!step || ((n1 == 0 || k1) && (n2 == 0 || k2) && rate >= 0.f && rate < 1.f), "fwd_prologue: bad dropout arguments");
  PrologueArgs a;
  memset(&a, 0, sizeof(a));
  a.xyz = xyz; a.B = B; a.N = N; a.out = out; a.centroid = centroid; a.scale = scale;
  long long mx;
  PN_TRY(make_prep3(w, sgn, K, C, hi, lo, a.prep, mx));
  a.prep_chunks = mx;
  a.n_prep = (int)cdivll(3 * mx, 1024);
  a.zero_u = zero_u; a.zero_u_n = zero_u_n;
  if (grads && n_grads > 0) {
    a.grads = grads; a.n_grads = n_grads;
    const long long blocks = cdivll(cdivll(n_grads, 4), 1024);
    a.n_zero = (int)(blocks < 1 ? 1 : (blocks < 1024 ? blocks : 1024));
  }
  if (step) {
    a.n_drop = 1; a.k1 = k1; a.k2 = k2; a.n1 = n1; a.n2 = n2; a.rate = rate;
    a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32); a.step = step;
  }
  int total = 0;
  for (int i = 0; i < n_wcopies; ++i) {
    const WCopyDesc& q = wcopies[i];
    PN_CHECK_ARG(q.w && q.nat && q.tr && q.K > 0 && q.C > 0 && q.K % 8 == 0 && q.C % 8 == 0, "fwd_prologue: bad kernel-copy job");
    a.wc.w[i] = q.w; a.wc.nat[i] = reinterpret_cast<unsigned short*>(q.nat); a.wc.tr[i] = reinterpret_cast<unsigned short*>(q.tr);
    a.wc.K[i] = q.K; a.wc.C[i] = q.C;
    PN_CHECK_ARG(!q.frag || (q.K % 16 == 0 && q.C % 32 == 0), "fwd_prologue: a fragment-major kernel copy needs K %% 16 == 0, C %% 32 == 0");
    a.wc.frag[i] = q.frag ? 1 : 0;
    total += q.K * q.C;
    a.wc.end[i] = total;
  }
  a.wc.n = n_wcopies;
  a.n_wcopy = cdiv(total / 4, 1024);                    // items of eight outputs, two copies: total / 4 items
  for (int i = 0; i < n_frozen; ++i) {
    PN_CHECK_ARG(frozen[i].gamma && frozen[i].beta && frozen[i].mm && frozen[i].mv && frozen[i].scale && frozen[i].shift && frozen[i].C > 0,
                 "fwd_prologue: bad frozen-layer job");
    a.fz.j[i] = frozen[i];
  }
  a.fz.n = n_frozen; a.fz.eps = bn_eps;
  hipLaunchKernelGGL(fwd_prologue_kernel, dim3(B + a.n_prep + a.n_zero + a.n_drop + a.n_wcopy + n_frozen), dim3(1024), 0, st, a);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

}  // namespace pn
