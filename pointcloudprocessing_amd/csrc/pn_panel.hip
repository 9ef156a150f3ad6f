// Row-panel contraction kernel for the dominant layer: ConvLayer(K -> C) + BN sums + reduce_max over points
// (pointnet/PointNet.py:242-248, 425-429) on bf16 MFMA, and the weight preparation it needs.
//
// The generic engine (pn_gemm.hip) gives every (row tile, column tile) pair its own workgroup, so the fp32 activation
// tile is re-staged C/128 times and the fp32 Keras kernel is gathered with 4-byte loads: at C = 1024 the launch is
// bound by L2 -> CU traffic, not by MFMA.  Here a workgroup owns a PANEL of 128 (or 64) point rows for ALL C channels:
//   * the activation panel (rows x K, BN+ReLU applied on load, rounded once to bf16 hi [+lo]) is staged into LDS once;
//   * the kernel is read from a bf16, channel-major copy Wb[C][K] (pn_weights_prep: one launch per step) with
//     16-byte loads straight into the LDS image, one 128-channel tile at a time, prefetched in registers while the
//     previous tile is in the matrix cores;
//   * per channel tile the epilogue keeps max / arg-max row / sum / sum of squares of the panel's rows and writes one
//     partial per (panel, channel); no (B*N) x C tensor ever exists.
// Every panel streams the whole bf16 kernel (256 KB at 128 x 1024) from L2, so the panel height sets that traffic: 128-row
// panels (512 threads, 78 KB of LDS in bf16 mode -> two per CU) move 66 MB per launch at B*N = 32,768, 64-row panels 131 MB
// (22.5 vs 19.9 us at N = 1024, 81 vs 59 us at N = 4096).  Two workgroups per CU let one's epilogue (vector ALU) overlap the
// other's MFMAs and loads.
#include "pn_common.h"

namespace pn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // native vector: stays in registers (HIP's uint4 struct does not)

// ---- weight preparation: Wb_hi[c][k] = bf16(W[k][c]),  Wb_lo[c][k] = bf16(W[k][c] - hi) --------------------------
__global__ __launch_bounds__(256) void weights_prep_kernel(const float* __restrict__ w, int K, int C, __bf16* __restrict__ hi,
                                                           __bf16* __restrict__ lo) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;      // bx over C, by over K
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (by + i < K && bx + tx < C) t[i][tx] = w[(long long)(by + i) * C + bx + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (bx + i < C && by + tx < K) {
      const float v = t[tx][i];
      const __bf16 h = (__bf16)v;
      hi[(long long)(bx + i) * K + by + tx] = h;
      if (lo) lo[(long long)(bx + i) * K + by + tx] = (__bf16)(v - (float)h);
    }
}

// the three max-pooled layers' kernels in one launch (blockIdx.z picks the matrix); block (0,0,0) also clears `zero_n` words
struct Prep3Args {
  const float* sgn[3];          // optional per-channel sign source (gamma): the copy holds sign(gamma_c) * W[:, c]
  const float* w[3];
  __bf16* hi[3];
  __bf16* lo[3];
  int K[3], C[3];
  unsigned* zero_p;
  int zero_n;
};
__global__ __launch_bounds__(256) void weights_prep3_kernel(const Prep3Args a) {
  __shared__ float t[32][33];
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && a.zero_p)
    for (int i = threadIdx.x; i < a.zero_n; i += 256) a.zero_p[i] = 0u;
  const int z = blockIdx.z;
  const float* w = a.w[z];
  if (!w) return;
  const int K = a.K[z], C = a.C[z];
  __bf16* hi = a.hi[z];
  __bf16* lo = a.lo[z];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;      // bx over C, by over K
  if (bx >= C || by >= K) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (by + i < K && bx + tx < C) t[i][tx] = w[(long long)(by + i) * C + bx + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (bx + i < C && by + tx < K) {
      const float sg = (a.sgn[z] && a.sgn[z][bx + i] < 0.f) ? -1.f : 1.f;      // exact: a sign flip commutes with the rounding
      const float v = sg * t[tx][i];
      const __bf16 h = (__bf16)v;
      hi[(long long)(bx + i) * K + by + tx] = h;
      if (lo) lo[(long long)(bx + i) * K + by + tx] = (__bf16)(v - (float)h);
    }
}
int weights_prep3(const float* const* w, const float* const* sgn, const int* K, const int* C, void* const* hi, void* const* lo,
                  unsigned* zero_p, int zero_n, hipStream_t st) {
  Prep3Args a;
  int mk = 1, mc = 1;
  for (int i = 0; i < 3; ++i) {
    a.w[i] = w[i]; a.sgn[i] = sgn ? sgn[i] : nullptr; a.K[i] = K[i]; a.C[i] = C[i];
    a.hi[i] = reinterpret_cast<__bf16*>(hi[i]); a.lo[i] = reinterpret_cast<__bf16*>(lo[i]);
    PN_CHECK_ARG(!w[i] || (hi[i] && K[i] > 0 && C[i] > 0), "weights_prep3: bad arguments");
    if (w[i]) { mk = K[i] > mk ? K[i] : mk; mc = C[i] > mc ? C[i] : mc; }
  }
  a.zero_p = zero_p; a.zero_n = zero_n;
  hipLaunchKernelGGL(weights_prep3_kernel, dim3(cdiv(mc, 32), cdiv(mk, 32), 3), dim3(256), 0, st, a);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int weights_prep(const float* w, int K, int C, void* hi, void* lo, hipStream_t st) {
  PN_CHECK_ARG(w && hi && K > 0 && C > 0, "pn_weights_prep: bad arguments");
  hipLaunchKernelGGL(weights_prep_kernel, dim3(cdiv(C, 32), cdiv(K, 32)), dim3(256), 0, st, w, K, C, reinterpret_cast<__bf16*>(hi),
                     reinterpret_cast<__bf16*>(lo));
  PN_CHECK_LAUNCH();
  return PN_OK;
}

struct PanelArgs {
  pn_operand a;                 // lazy activation operand over (B*N, K)
  const __bf16* wb_hi;          // [C][K]
  const __bf16* wb_lo;          // [C][K] (NS == 3)
  int B, N, K, C;
  int tiles_per_cloud;          // ceil(N / 64)
  const float* sgn;             // per channel; only the sign is used (may be gamma)
  int presigned;                // the weight copy already carries sign(gamma_c): the accumulators hold sgn*z
  float* pmax;                  // [tiles][C]
  int* pidx;                    // [tiles][C]
  float* stat_partials;         // [tiles][2][C] or NULL
};

// 256 threads = 4 waves as 2 (row halves of 32) x 2 (column halves of 64); wave tile 32 x 64 = 1 x 2 MFMA 32x32 blocks.
// K is a compile-time constant so that every loop over k unrolls and every staging array stays in registers.
template <int NT, int PF, int K, int PA, int THREADS>
__device__ __forceinline__ void panel_issue_b(u32x4 (&pf)[NT][PF], const __bf16* __restrict__ whi, const __bf16* __restrict__ wlo,
                                              int ct, int tid) {
  constexpr int CHB = K / 8;
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    const int c = tid + THREADS * p;                 // 128 * CHB chunks per tile, PF * THREADS == 128 * CHB
    const int j = c / CHB, kc = (c % CHB) * 8;
    const long long o = (long long)(ct * 128 + j) * K + kc;
    pf[0][p] = *reinterpret_cast<const u32x4*>(whi + o);
    if (NT == 2) pf[NT - 1][p] = *reinterpret_cast<const u32x4*>(wlo + o);
  }
}
template <int NT, int PF, int K, int PA, int THREADS>
__device__ __forceinline__ void panel_write_b(const u32x4 (&pf)[NT][PF], __bf16* __restrict__ bhi, __bf16* __restrict__ blo, int tid) {
  constexpr int CHB = K / 8;
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    const int c = tid + THREADS * p;
    const int j = c / CHB, kc = (c % CHB) * 8;
    *reinterpret_cast<u32x4*>(bhi + j * PA + kc) = pf[0][p];
    if (NT == 2) *reinterpret_cast<u32x4*>(blo + j * PA + kc) = pf[NT - 1][p];
  }
}

// RG = row groups of 32 per panel: 2 -> 64-row panels, 256 threads; 4 -> 128-row panels, 512 threads.  Every panel streams the
// whole bf16 kernel (256 KB at 128 x 1024) from L2, so a 128-row panel halves that traffic (131 -> 66 MB per launch at B*N = 32,768).
template <int NS, int K, bool STATS, int RG>
__global__ __launch_bounds__(128 * RG) void panel_max_kernel(const PanelArgs g) {
  constexpr int BM = 32 * RG, BN = 128, THREADS = 128 * RG;
  constexpr int PA = K + 8;                          // LDS row pitch (bf16 elements): conflict-free b128 rows
  constexpr int NT = (NS == 3) ? 2 : 1;
  constexpr int PF = (BN * (K / 8)) / THREADS;       // 16-byte weight chunks per thread per tile (8 at K = 128, 256 threads)
  __shared__ __attribute__((aligned(16))) __bf16 Ap[NT][BM * PA];
  __shared__ __attribute__((aligned(16))) __bf16 Bt[NT][BN * PA];
  __shared__ float red[RG][4][BN];                   // [row group][sum, sumsq, max, idx][channel]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int bx = blockIdx.x;
  const int cloud = bx / g.tiles_per_cloud, tin = bx - cloud * g.tiles_per_cloud;
  const int row_in_cloud0 = tin * BM;
  const int nrows = min(BM, g.N - row_in_cloud0);
  const long long row0 = (long long)cloud * g.N + row_in_cloud0;

  // first weight tile in flight while the activation panel is staged
  u32x4 pf[NT][PF];
  panel_issue_b<NT, PF, K, PA, THREADS>(pf, g.wb_hi, g.wb_lo, 0, tid);

  // ---- stage the activation panel: thread <-> (row, 8 consecutive k); coefficients indexed by k -----------------
  {
    constexpr int CH = K / 8;                        // 16-byte chunks per row
    constexpr int RP = THREADS / CH;                 // rows per pass
    constexpr int P = BM / RP;
    const int ch = tid % CH, rin = tid / CH;
    const int k = ch * 8;
    float4 x[P][2];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int rr = p * RP + rin;
      const long long rsrc = (rr < nrows) ? rr : (nrows - 1);
      const float* s = g.a.s1 + (row0 + rsrc) * g.a.ld + k;
      x[p][0] = *reinterpret_cast<const float4*>(s);
      x[p][1] = *reinterpret_cast<const float4*>(s + 4);
    }
    float ca[8], cc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ca[e] = 1.f; cc[e] = 0.f; }
    if (g.a.ca) {
      const float4 t0 = *reinterpret_cast<const float4*>(g.a.ca + k), t1 = *reinterpret_cast<const float4*>(g.a.ca + k + 4);
      ca[0] = t0.x; ca[1] = t0.y; ca[2] = t0.z; ca[3] = t0.w; ca[4] = t1.x; ca[5] = t1.y; ca[6] = t1.z; ca[7] = t1.w;
    }
    if (g.a.cc) {
      const float4 t0 = *reinterpret_cast<const float4*>(g.a.cc + k), t1 = *reinterpret_cast<const float4*>(g.a.cc + k + 4);
      cc[0] = t0.x; cc[1] = t0.y; cc[2] = t0.z; cc[3] = t0.w; cc[4] = t1.x; cc[5] = t1.y; cc[6] = t1.z; cc[7] = t1.w;
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(x[p][q].x), "+v"(x[p][q].y), "+v"(x[p][q].z), "+v"(x[p][q].w));
    const float lo = g.a.lo;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int rr = p * RP + rin;
      const bool rv = rr < nrows;
      const float v[8] = {x[p][0].x, x[p][0].y, x[p][0].z, x[p][0].w, x[p][1].x, x[p][1].y, x[p][1].z, x[p][1].w};
      bf16x8 hv, lv;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = rv ? clamp_lo(fmaf(ca[e], v[e], cc[e]), lo) : 0.f;
        hv[e] = (__bf16)t;
        if (NS == 3) lv[e] = (__bf16)(t - (float)hv[e]);
      }
      *reinterpret_cast<bf16x8*>(&Ap[0][rr * PA + k]) = hv;
      if (NS == 3) *reinterpret_cast<bf16x8*>(&Ap[NT - 1][rr * PA + k]) = lv;
    }
  }
  panel_write_b<NT, PF, K, PA, THREADS>(pf, Bt[0], Bt[NT - 1], tid);
  __syncthreads();

  // ---- channel tiles -------------------------------------------------------------------------------------------
  const int n_ct = g.C / BN;
  const bool full = nrows == BM;                      // block-uniform
  const int rbase_lane = row_in_cloud0 + wm * 32 + 4 * h;
  for (int ct = 0; ct < n_ct; ++ct) {
    if (ct + 1 < n_ct) panel_issue_b<NT, PF, K, PA, THREADS>(pf, g.wb_hi, g.wb_lo, ct + 1, tid);   // flies under the MFMAs
    // this lane's two channel signs, loaded here so that they arrive under the MFMAs (no LDS copy: the 4 KB it took kept a
    // 128-row panel's workgroup above half of the LDS, i.e. at one workgroup per CU)
    float sgv[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) sgv[n] = g.sgn[ct * BN + wn * 64 + n * 32 + r];
    f32x16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < K / 16; ++ks) {
      const int oa = (wm * 32 + r) * PA + ks * 16 + h * 8;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Ap[0][oa]);
      bf16x8 al;
      if (NS == 3) al = *reinterpret_cast<const bf16x8*>(&Ap[NT - 1][oa]);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int ob = (wn * 64 + n * 32 + r) * PA + ks * 16 + h * 8;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bt[0][ob]);
        if (NS == 3) {
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bt[NT - 1][ob]);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[n], 0, 0, 0);
        }
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[n], 0, 0, 0);
      }
    }
    // ---- epilogue of this channel tile: per column max / arg-max / sums over the wave's 32 rows.  Branch-free:
    //      rows outside the cloud are neutralised with selects (a branch per element costs far more than the MFMAs).
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int jl = wn * 64 + n * 32 + r;
      const float sgc = (sgv[n] < 0.f) ? -1.f : 1.f;
      const float sg = g.presigned ? 1.f : sgc;          // presigned copies: acc = sgn*z already, no per-element multiply
      float a1 = 0.f, a2 = 0.f, best = -INFINITY;
      int besti = 0x7fffffff;
      if (full) {
        if (STATS) {                                     // inference (moving statistics) needs no sums: a quarter of the epilogue
          f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};       // packed pairs: v_pk_add_f32 / v_pk_fma_f32
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            const f32x2 v2 = {acc[n][e], acc[n][e + 1]};
            s1 += v2;
            s2 = __builtin_elementwise_fma(v2, v2, s2);
          }
          a1 = s1.x + s1.y;
          a2 = s2.x + s2.y;
        }
        if (g.presigned) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float t = acc[n][e];
            const bool better = t > best;                 // rows ascend with e: first maximum wins
            best = better ? t : best;
            besti = better ? (rbase_lane + (e & 3) + 8 * (e >> 2)) : besti;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float t = sg * acc[n][e];
            const bool better = t > best;
            best = better ? t : best;
            besti = better ? (rbase_lane + (e & 3) + 8 * (e >> 2)) : besti;
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int il = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const bool ok = il < nrows;
          const float v = ok ? acc[n][e] : 0.f;
          if (STATS) {
            a1 += v;
            a2 = fmaf(v, v, a2);
          }
          const float t = ok ? sg * v : -INFINITY;
          const bool better = t > best;
          best = better ? t : best;
          besti = better ? (row_in_cloud0 + il) : besti;
        }
      }
      if (g.presigned) a1 *= sgc;                         // the column sum of z itself
      if (STATS) {
        a1 += __shfl_xor(a1, 32, 64);
        a2 += __shfl_xor(a2, 32, 64);
      }
      const float ob = __shfl_xor(best, 32, 64);
      const int oi = __shfl_xor(besti, 32, 64);
      const bool take = ob > best || (ob == best && oi < besti);
      best = take ? ob : best;
      besti = take ? oi : besti;
      if (h == 0) {
        if (STATS) {
          red[wm][0][jl] = a1;
          red[wm][1][jl] = a2;
        }
        red[wm][2][jl] = best;
        reinterpret_cast<int*>(red[wm][3])[jl] = besti;
      }
    }
    __syncthreads();                                 // red complete; every wave is done reading Bt
    if (tid < BN) {
      const int j = ct * BN + tid;
      if (STATS && g.stat_partials) {
        float* p = g.stat_partials + (long long)bx * 2 * g.C + j;
        float t1 = red[0][0][tid], t2 = red[0][1][tid];
#pragma unroll
        for (int q = 1; q < RG; ++q) { t1 += red[q][0][tid]; t2 += red[q][1][tid]; }
        p[0] = t1;
        p[g.C] = t2;
      }
      float v0 = red[0][2][tid];
      int i0 = reinterpret_cast<int*>(red[0][3])[tid];
#pragma unroll
      for (int q = 1; q < RG; ++q) {                   // row groups ascend: a later group wins only with a strictly larger value
        const float v1 = red[q][2][tid];
        const int i1 = reinterpret_cast<int*>(red[q][3])[tid];
        if (v1 > v0 || (v1 == v0 && i1 < i0)) { v0 = v1; i0 = i1; }
      }
      g.pmax[(long long)bx * g.C + j] = v0;
      g.pidx[(long long)bx * g.C + j] = i0;
    }
    if (ct + 1 < n_ct) panel_write_b<NT, PF, K, PA, THREADS>(pf, Bt[0], Bt[NT - 1], tid);
    __syncthreads();                                 // next tile visible; red free again
  }
}

template <int NS, int K>
static void launch_panel(const PanelArgs& g, dim3 grid, bool stats, int panel_rows, hipStream_t st) {
  if (panel_rows == 256) {
    if constexpr (!(NS == 3 && K == 128)) {          // the hi + lo images of a 256-row panel at K = 128 exceed the LDS
      if (stats) hipLaunchKernelGGL((panel_max_kernel<NS, K, true, 8>), grid, dim3(1024), 0, st, g);
      else hipLaunchKernelGGL((panel_max_kernel<NS, K, false, 8>), grid, dim3(1024), 0, st, g);
    }
  } else if (panel_rows == 128) {
    if (stats) hipLaunchKernelGGL((panel_max_kernel<NS, K, true, 4>), grid, dim3(512), 0, st, g);
    else hipLaunchKernelGGL((panel_max_kernel<NS, K, false, 4>), grid, dim3(512), 0, st, g);
  } else {
    if (stats) hipLaunchKernelGGL((panel_max_kernel<NS, K, true, 2>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((panel_max_kernel<NS, K, false, 2>), grid, dim3(256), 0, st, g);
  }
}

int conv_fwd_max_panel(const pn_operand* x, const void* wb_hi, const void* wb_lo, int B, int N, int K, int C, const float* sgn,
                       float* pmax, int* pidx, float* stat_partials, int prec, hipStream_t st, int presigned, int panel_rows) {
  PN_CHECK_ARG(x && x->s1 && !x->s2, "pn_conv_fwd_max_panel: bad operand");
  PN_CHECK_ARG((reinterpret_cast<uintptr_t>(x->s1) & 15) == 0 && x->ld % 4 == 0 && x->ld >= K, "pn_conv_fwd_max_panel: operand alignment");
  PN_CHECK_ARG(B > 0 && N > 0, "pn_conv_fwd_max_panel: B and N must be positive");
  PN_CHECK_ARG(K == 64 || K == 128, "pn_conv_fwd_max_panel: K must be 64 or 128 (K=%d)", K);
  PN_CHECK_ARG(C >= 128 && C % 128 == 0 && C <= 1024, "pn_conv_fwd_max_panel: C must be a multiple of 128, at most 1024 (C=%d)", C);
  PN_CHECK_ARG(wb_hi && sgn && pmax && pidx, "pn_conv_fwd_max_panel: null pointer");
  PN_CHECK_ARG(prec == PN_PREC_BF16 || (prec == PN_PREC_BF16X3 && wb_lo), "pn_conv_fwd_max_panel: bad prec / missing lo weights");
  PanelArgs g;
  memset(&g, 0, sizeof(g));
  g.a = *x; g.wb_hi = reinterpret_cast<const __bf16*>(wb_hi); g.wb_lo = reinterpret_cast<const __bf16*>(wb_lo);
  g.B = B; g.N = N; g.K = K; g.C = C;
  PN_CHECK_ARG(panel_rows == 64 || panel_rows == 128 || panel_rows == 256, "pn_conv_fwd_max_panel: panel_rows must be 64, 128 or 256");
  PN_CHECK_ARG(!(panel_rows == 256 && prec == PN_PREC_BF16X3 && K == 128), "pn_conv_fwd_max_panel: 256-row panels do not fit LDS with bf16x3 operands at K = 128");
  g.tiles_per_cloud = cdiv(N, panel_rows);
  g.sgn = sgn; g.pmax = pmax; g.pidx = pidx; g.stat_partials = stat_partials; g.presigned = presigned;
  const dim3 grid(B * g.tiles_per_cloud);
  const bool st_ = stat_partials != nullptr;
  if (prec == PN_PREC_BF16X3) {
    if (K == 128) launch_panel<3, 128>(g, grid, st_, panel_rows, st);
    else launch_panel<3, 64>(g, grid, st_, panel_rows, st);
  } else {
    if (K == 128) launch_panel<1, 128>(g, grid, st_, panel_rows, st);
    else launch_panel<1, 64>(g, grid, st_, panel_rows, st);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}

}  // namespace pn
