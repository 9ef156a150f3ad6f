// Row-panel kernel for the dominant layer: ConvLayer(K -> C) + BatchNormalization statistics + tf.reduce_max over the points
// (pointnet/PointNet.py:242-248, 425-429) on bf16 MFMA, its weight preparation and its finaliser.
//
// A workgroup (4 waves) owns a PANEL of 128 (or 64) point rows for ALL C channels:
//   * the activation panel (rows x K, the previous layer's BN + ReLU applied on load, rounded once to bf16 hi [+ lo]) is staged
//     into LDS once and is the only thing the waves share: one barrier at the start of the kernel, none afterwards;
//   * every wave owns whole COLUMNS: wave w computes column blocks w, w + 4, ... (32 channels each) for all rows of the panel, so a
//     column's maximum over the panel never leaves the wave (no cross-wave reduction, no per-tile barrier, no LDS traffic for the
//     results) and the B operand is private to the wave: it is read from a FRAGMENT-ORDERED bf16 copy of the kernel
//     (pn_weights_prep: the 16 bytes lane l needs for k-step ks of column block cb sit at ((cb * K/16 + ks) * 64 + l) * 16, so a
//     wave's load is one coalesced 1 KB transaction straight into the MFMA register layout, prefetched one column block ahead);
//   * the epilogue holds only what cannot be had elsewhere: max over the rows (v_max3: half an instruction per accumulator
//     element) and sum of squares (packed fma, half an instruction).  The column SUM comes from the panel's column sums of A
//     (a1 = A^T 1, taken from LDS once per panel): sum_m z[m][c] = a1 . W[:, c], finished per channel by the finaliser.  The ROW
//     of the maximum is not tracked element by element either (that costs two to three instructions per element): the kernel
//     records which 32-row block of the panel held the maximum and the backward pass, which is the only consumer of the row, finds
//     it among those 32 candidates (pn_maxbwd.hip: max_resolve).
// Per point: 2 * K * C FLOP against 4 * K bytes of input: MFMA-bound by the roofline.
#include "pn_common.h"
#include "pn_internal.h"

namespace pn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // native vector: stays in registers (HIP's uint4 struct does not)

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
  unsigned* zero_p;
  int zero_n;
};
__global__ __launch_bounds__(256) void weights_prep3_kernel(const Prep3Args a) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && a.zero_p)
    for (int i = threadIdx.x; i < a.zero_n; i += 256) a.zero_p[i] = 0u;
  const int z = blockIdx.y;
  const float* __restrict__ w = a.w[z];
  if (!w) return;
  const int K = a.K[z], C = a.C[z], KS = K / 16;
  const long long chunk = (long long)blockIdx.x * 256 + threadIdx.x;      // one 16-byte chunk (8 consecutive k of one channel)
  if (chunk >= (long long)C * K / 8) return;
  const int lane = (int)(chunk & 63);
  const int ks = (int)((chunk >> 6) % KS), cb = (int)((chunk >> 6) / KS);
  const int c = cb * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
  const float sg = (a.sgn[z] && a.sgn[z][c] < 0.f) ? -1.f : 1.f;
  bf16x8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = sg * w[(long long)(k0 + j) * C + c];
    h[j] = (__bf16)v;
    l[j] = (__bf16)(v - (float)h[j]);
  }
  *reinterpret_cast<bf16x8*>(a.hi[z] + chunk * 8) = h;
  if (a.lo[z]) *reinterpret_cast<bf16x8*>(a.lo[z] + chunk * 8) = l;
}
int weights_prep3(const float* const* w, const float* const* sgn, const int* K, const int* C, void* const* hi, void* const* lo,
                  unsigned* zero_p, int zero_n, hipStream_t st) {
  Prep3Args a;
  long long mx = 1;
  for (int i = 0; i < 3; ++i) {
    a.w[i] = w[i]; a.sgn[i] = sgn ? sgn[i] : nullptr; a.K[i] = K[i]; a.C[i] = C[i];
    a.hi[i] = reinterpret_cast<__bf16*>(hi[i]); a.lo[i] = reinterpret_cast<__bf16*>(lo[i]);
    PN_CHECK_ARG(!w[i] || (hi[i] && K[i] > 0 && K[i] % 16 == 0 && C[i] > 0 && C[i] % 32 == 0), "weights_prep3: bad arguments");
    if (w[i] && (long long)K[i] * C[i] / 8 > mx) mx = (long long)K[i] * C[i] / 8;
  }
  a.zero_p = zero_p; a.zero_n = zero_n;
  hipLaunchKernelGGL(weights_prep3_kernel, dim3((unsigned)cdivll(mx, 256), 3), dim3(256), 0, st, a);
  PN_CHECK_LAUNCH();
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
  return weights_prep3(ws, sg, Ks, Cs, his, los, nullptr, 0, st);
}

// ---- the panel kernel --------------------------------------------------------------------------------------------------------
struct PanelArgs {
  pn_operand a;                 // lazy activation operand over (B*N, K)
  const __bf16* wf_hi;          // fragment-ordered copies (weights_prep)
  const __bf16* wf_lo;          // (NS == 3)
  int B, N, C;
  int tiles_per_cloud;          // ceil(N / panel rows)
  float* pmax;                  // [tiles][C]  max over the panel's rows of the accumulator (= sgn * z with presigned weights)
  int* pq;                      // [tiles][C]  index inside the cloud of the 32-row block that held it (lowest on ties)
  float* sumsq;                 // [tiles][C]  sum over the panel's rows of z^2, or NULL
  float* a1;                    // [tiles][NT * K]  column sums of the staged bf16 panel (hi image, then lo image), or NULL
};

// NS: 1 = bf16 operands, 3 = bf16 hi + lo (three products).  K compile time: every k loop unrolls.  MB = 32-row blocks per panel.
template <int NS, int K, bool STATS, int MB>
__global__ __launch_bounds__(256, 2) void panel_max_kernel(const PanelArgs g) {
  constexpr int BM = 32 * MB, THREADS = 256, KS = K / 16;
  constexpr int PA = K + 8;                          // LDS row pitch (bf16 elements): rows r .. r+15 on 16 distinct 16-byte slots
  constexpr int NT = (NS == 3) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) __bf16 Ap[NT][BM * PA];
  __shared__ float colsum[4][NT * K];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int bx = blockIdx.x;
  const int cloud = bx / g.tiles_per_cloud, tin = bx - cloud * g.tiles_per_cloud;
  const int row_in_cloud0 = tin * BM;
  const int nrows = min(BM, g.N - row_in_cloud0);
  const long long row0 = (long long)cloud * g.N + row_in_cloud0;
  const int n_cb = g.C / 32;

  // this wave's first column block of the kernel is in flight while the activation panel is staged
  const u32x4* __restrict__ wfh = reinterpret_cast<const u32x4*>(g.wf_hi);
  const u32x4* __restrict__ wfl = reinterpret_cast<const u32x4*>(g.wf_lo);
  u32x4 bnext[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) bnext[ks] = wfh[((long long)wave * KS + ks) * 64 + lane];

  // ---- stage the activation panel: thread <-> (row, 8 consecutive k); BN + ReLU coefficients indexed by k ---------------------
  {
    constexpr int CH = K / 8;                        // 16-byte bf16 chunks per row
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
    const float lo = g.a.lo;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int rr = p * RP + rin;
      const bool rv = rr < nrows;
      const float v[8] = {x[p][0].x, x[p][0].y, x[p][0].z, x[p][0].w, x[p][1].x, x[p][1].y, x[p][1].z, x[p][1].w};
      bf16x8 hv, lv;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = rv ? clamp_lo(fmaf(ca[e], v[e], cc[e]), lo) : 0.f;    // rows outside the cloud are zero rows
        hv[e] = (__bf16)t;
        if (NS == 3) lv[e] = (__bf16)(t - (float)hv[e]);
      }
      *reinterpret_cast<bf16x8*>(&Ap[0][rr * PA + k]) = hv;
      if (NS == 3) *reinterpret_cast<bf16x8*>(&Ap[NT - 1][rr * PA + k]) = lv;
    }
  }
  __syncthreads();
  if (STATS && g.a1) {
    // column sums of the staged images: thread <-> (column, quarter of the rows), then four partials per column
    for (int i = tid; i < 4 * NT * K; i += THREADS) {
      const int col = i % (NT * K), qr = i / (NT * K);
      const __bf16* img = Ap[col / K];
      const int kk = col % K;
      float s = 0.f;
#pragma unroll 8
      for (int rr = qr * (BM / 4); rr < (qr + 1) * (BM / 4); ++rr) s += (float)img[rr * PA + kk];
      colsum[qr][col] = s;
    }
    __syncthreads();
    for (int col = tid; col < NT * K; col += THREADS)
      g.a1[(long long)bx * (NT * K) + col] = (colsum[0][col] + colsum[1][col]) + (colsum[2][col] + colsum[3][col]);
  }

  // ---- this wave's column blocks: cb = wave, wave + 4, ... ------------------------------------------------------------------
  // One B buffer (KS x 16 bytes per lane) is consumed while the next one is in flight.  bf16 operands: one buffer per column block and
  // the compiler keeps the wave's A fragments in registers across the column blocks (they do not depend on cb: 128 VGPRs at K = 128,
  // four 32-row blocks), so the loop body is MFMAs only.  bf16x3: two buffers per column block -- first the hi image of the kernel
  // (products a_lo.b_hi and a_hi.b_hi), then its lo image (a_hi.b_lo) -- and the A fragments (twice as many) are re-read from LDS in
  // every phase: the compiler barriers below keep it from hoisting 256 registers' worth of them out of the loop.
  const bool full = nrows == BM;                      // block-uniform
  for (int cb = wave; cb < n_cb; cb += 4) {
    u32x4 bcur[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bcur[ks] = bnext[ks];
    if (NS == 3) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bnext[ks] = wfl[((long long)cb * KS + ks) * 64 + lane];
      asm volatile("" ::: "memory");
    } else if (cb + 4 < n_cb) {                       // next column block: flies under this one's MFMAs
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bnext[ks] = wfh[((long long)(cb + 4) * KS + ks) * 64 + lane];
    }
    f32x16 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 vb = __builtin_bit_cast(bf16x8, bcur[ks]);
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int oa = (m * 32 + r) * PA + ks * 16 + h * 8;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Ap[0][oa]);
        if (NS == 3) {
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(&Ap[NT - 1][oa]);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, vb, acc[m], 0, 0, 0);
        }
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, vb, acc[m], 0, 0, 0);
      }
    }
    if (NS == 3) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bcur[ks] = bnext[ks];
      if (cb + 4 < n_cb) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bnext[ks] = wfh[((long long)(cb + 4) * KS + ks) * 64 + lane];
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 vb = __builtin_bit_cast(bf16x8, bcur[ks]);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Ap[0][(m * 32 + r) * PA + ks * 16 + h * 8]);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, vb, acc[m], 0, 0, 0);
        }
      }
    }
    // ---- epilogue: this lane's column over its 16 rows of every 32-row block (the other half-wave holds the other 16) --------
    float best = -INFINITY, ss = 0.f;
    int bq = 0;
    if (STATS) {
      f32x2 s2 = {0.f, 0.f};                           // packed pairs: v_pk_fma_f32
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const f32x2 v2 = {acc[m][e], acc[m][e + 1]};
          s2 = __builtin_elementwise_fma(v2, v2, s2);
        }
      ss = s2.x + s2.y;
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float mx;
      if (full) {
        mx = fmaxf(fmaxf(acc[m][0], acc[m][1]), acc[m][2]);              // v_max3_f32 chains
#pragma unroll
        for (int e = 3; e < 15; e += 2) mx = fmaxf(fmaxf(mx, acc[m][e]), acc[m][e + 1]);
        mx = fmaxf(mx, acc[m][15]);
      } else {
        mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int il = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          mx = fmaxf(mx, il < nrows ? acc[m][e] : -INFINITY);
        }
      }
      const bool better = mx > best;                   // blocks ascend: the first maximum wins
      best = better ? mx : best;
      bq = better ? m : bq;
    }
    const float ob = __shfl_xor(best, 32, 64);
    const int oq = __shfl_xor(bq, 32, 64);
    const bool take = ob > best || (ob == best && oq < bq);
    best = take ? ob : best;
    bq = take ? oq : bq;
    if (STATS) ss += __shfl_xor(ss, 32, 64);
    if (h == 0) {
      const long long o = (long long)bx * g.C + cb * 32 + r;
      g.pmax[o] = best;
      g.pq[o] = (row_in_cloud0 >> 5) + bq;
      if (STATS && g.sumsq) g.sumsq[o] = ss;
    }
  }
}

template <int NS, int K>
static void launch_panel(const PanelArgs& g, dim3 grid, bool stats, int panel_rows, hipStream_t st) {
  if (panel_rows == 128) {
    if (stats) hipLaunchKernelGGL((panel_max_kernel<NS, K, true, 4>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((panel_max_kernel<NS, K, false, 4>), grid, dim3(256), 0, st, g);
  } else {
    if (stats) hipLaunchKernelGGL((panel_max_kernel<NS, K, true, 2>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((panel_max_kernel<NS, K, false, 2>), grid, dim3(256), 0, st, g);
  }
}

int conv_fwd_max_panel(const pn_operand* x, const void* wf_hi, const void* wf_lo, int B, int N, int K, int C, float* pmax, int* pq,
                       float* sumsq, float* a1, int prec, int panel_rows, hipStream_t st) {
  PN_CHECK_ARG(x && x->s1 && !x->s2, "pn_conv_fwd_max_panel: bad operand");
  PN_CHECK_ARG((reinterpret_cast<uintptr_t>(x->s1) & 15) == 0 && x->ld % 4 == 0 && x->ld >= K, "pn_conv_fwd_max_panel: operand alignment");
  PN_CHECK_ARG(B > 0 && N > 0, "pn_conv_fwd_max_panel: B and N must be positive");
  PN_CHECK_ARG(K == 64 || K == 128, "pn_conv_fwd_max_panel: K must be 64 or 128 (K=%d)", K);
  PN_CHECK_ARG(C >= 128 && C % 128 == 0, "pn_conv_fwd_max_panel: C must be a multiple of 128 (C=%d)", C);
  PN_CHECK_ARG(wf_hi && pmax && pq, "pn_conv_fwd_max_panel: null pointer");
  PN_CHECK_ARG((sumsq == nullptr) == (a1 == nullptr), "pn_conv_fwd_max_panel: sumsq and a1 come together (both or neither)");
  PN_CHECK_ARG(prec == PN_PREC_BF16 || (prec == PN_PREC_BF16X3 && wf_lo), "pn_conv_fwd_max_panel: bad prec / missing lo weights");
  PN_CHECK_ARG(panel_rows == 64 || panel_rows == 128, "pn_conv_fwd_max_panel: panel_rows must be 64 or 128");
  PanelArgs g;
  memset(&g, 0, sizeof(g));
  g.a = *x; g.wf_hi = reinterpret_cast<const __bf16*>(wf_hi); g.wf_lo = reinterpret_cast<const __bf16*>(wf_lo);
  g.B = B; g.N = N; g.C = C;
  g.tiles_per_cloud = cdiv(N, panel_rows);
  g.pmax = pmax; g.pq = pq; g.sumsq = sumsq; g.a1 = a1;
  const dim3 grid(B * g.tiles_per_cloud);
  const bool st_ = sumsq != nullptr;
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

// ---- finaliser: BatchNormalization statistics of the layer + reduce_max over each cloud's panels ---------------------------------
// One workgroup per (column block of 32 channels, slice of the clouds).  Training statistics (use_batch): per channel
//   sum z   = s_c * sum_k a1[k] * Wf[c][k]            a1 = sum over the panels of their column sums (both images for bf16x3)
//   sum z^2 = sum over the panels of sumsq
// combined in fp64 -> mean, invstd, scale, shift, moving statistics (only the workgroups of the first cloud slice write them);
// otherwise the coefficients come from the moving statistics (inference / frozen layer, PointNet.py:585-591).  Then per cloud:
// the largest pmax over its panels (lowest panel on ties: rows ascend with the panel index), zstar = s_c * max,
// g = relu(scale * zstar + shift), and the 32-row block that holds the row.
struct PanelFinArgs {
  const float* pmax; const int* pq; const float* sumsq; const float* a1;
  const __bf16* wf_hi; const __bf16* wf_lo;
  int T, tpc, B, C, K, NT, n_blocks32;
  double inv_count;
  const float* gamma; const float* beta; float* mm; float* mv;
  float momentum, eps;
  int use_batch, update;
  float *mean, *invstd, *scale, *shift, *g, *zstar;
  int* argq;
};
__global__ __launch_bounds__(256) void panel_finalize_kernel(const PanelFinArgs a) {
  __shared__ double a1s[256];
  __shared__ double red[8][2][32];
  __shared__ float sc_s[32], sh_s[32], sg_s[32];
  const int tid = threadIdx.x, cl = tid & 31, part = tid >> 5;
  const int cb = blockIdx.x, c = cb * 32 + cl;
  const int K = a.K, KS = K / 16, NTK = a.NT * K;       // NT * K <= 256
  const float gam = a.gamma[c];
  const float sg = gam < 0.f ? -1.f : 1.f;
  if (a.use_batch) {
    // (1) column sums of A over all panels: thread <-> (column, every pstep-th panel); NT * K is 64, 128 or 256
    const int pstep = 256 / NTK;
    {
      const int col = tid % NTK, p0 = tid / NTK;
      double s = 0.0;
      for (int p = p0; p < a.T; p += pstep) s += (double)a.a1[(long long)p * NTK + col];
      a1s[tid] = s;                                    // slot p0 * NTK + col
    }
    __syncthreads();
    // (2) sum z for this block's 32 channels: thread <-> (channel, k-step)
    double sz = 0.0;
    for (int ks = part; ks < KS; ks += 8) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const long long chunk = ((long long)cb * KS + ks) * 64 + hh * 32 + cl;
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(a.wf_hi + chunk * 8);
        bf16x8 wl;
        if (a.NT == 2) wl = *reinterpret_cast<const bf16x8*>(a.wf_lo + chunk * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = ks * 16 + hh * 8 + j;
          double ah = 0.0, al = 0.0;
          for (int q = 0; q < pstep; ++q) {
            ah += a1s[q * NTK + k];
            if (a.NT == 2) al += a1s[q * NTK + K + k];
          }
          double w = (double)(float)wh[j];
          if (a.NT == 2) {
            sz += al * w;                              // a_lo . b_hi
            w += (double)(float)wl[j];
          }
          sz += ah * w;                                // a_hi . (b_hi + b_lo)
        }
      }
    }
    // (3) sum z^2 over the panels: thread <-> (channel, every 8th panel)
    double sq = 0.0;
    for (int p = part; p < a.T; p += 8) sq += (double)a.sumsq[(long long)p * a.C + c];
    red[part][0][cl] = sz;
    red[part][1][cl] = sq;
    __syncthreads();
    if (part == 0) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) { s1 += red[q][0][cl]; s2 += red[q][1][cl]; }
      const double mean = (double)sg * s1 * a.inv_count;
      double var = s2 * a.inv_count - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      const float is = (float)(1.0 / sqrt(var + (double)a.eps));
      const float scl = gam * is;
      const float sft = a.beta[c] - (float)mean * scl;
      sc_s[cl] = scl; sh_s[cl] = sft; sg_s[cl] = sg;
      if (blockIdx.y == 0) {
        a.mean[c] = (float)mean; a.invstd[c] = is; a.scale[c] = scl; a.shift[c] = sft;
        if (a.update) {
          a.mm[c] = a.momentum * a.mm[c] + (1.f - a.momentum) * (float)mean;
          a.mv[c] = a.momentum * a.mv[c] + (1.f - a.momentum) * (float)var;
        }
      }
    }
  } else if (part == 0) {
    const float mean = a.mm[c];
    const float is = 1.f / sqrtf(a.mv[c] + a.eps);
    const float scl = gam * is;
    const float sft = a.beta[c] - mean * scl;
    sc_s[cl] = scl; sh_s[cl] = sft; sg_s[cl] = sg;
    if (blockIdx.y == 0) { a.mean[c] = mean; a.invstd[c] = is; a.scale[c] = scl; a.shift[c] = sft; }
  }
  __syncthreads();
  // (4) per cloud: thread <-> (channel, cloud); this block's slice of the clouds
  const float scl = sc_s[cl], sft = sh_s[cl], sgc = sg_s[cl];
  for (int b = blockIdx.y * 8 + part; b < a.B; b += 8 * gridDim.y) {
    float best = -INFINITY;
    int bq = 0;
    const long long base = (long long)b * a.tpc * a.C + c;
    for (int t0 = 0; t0 < a.tpc; t0 += 8) {
      float v[8];
      int q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = min(t0 + u, a.tpc - 1);
        v[u] = a.pmax[base + (long long)t * a.C];
        q[u] = a.pq[base + (long long)t * a.C];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (t0 + u < a.tpc && v[u] > best) { best = v[u]; bq = q[u]; }
    }
    const float zs = sgc * best;
    const long long o = (long long)b * a.C + c;
    a.g[o] = clamp_lo(fmaf(scl, zs, sft), 0.f);
    if (a.zstar) a.zstar[o] = zs;
    if (a.argq) a.argq[o] = (bq >= 0 && bq < a.n_blocks32) ? bq : 0;     // NaN inputs leave no winner: keep the index in range
  }
}

int panel_finalize(const float* pmax, const int* pq, const float* sumsq, const float* a1, const void* wf_hi, const void* wf_lo, int B, int N,
                   int K, int C, int panel_rows, int prec, const float* gamma, const float* beta, float* mm, float* mv, float momentum,
                   float eps, int use_batch, int update, float* mean, float* invstd, float* scale, float* shift, float* g, float* zstar,
                   int* argq, hipStream_t st) {
  PN_CHECK_ARG(pmax && pq && gamma && beta && mm && mv && mean && invstd && scale && shift && g, "pn_panel_finalize: null pointer");
  PN_CHECK_ARG(!use_batch || (sumsq && a1 && wf_hi), "pn_panel_finalize: batch statistics need sumsq, a1 and the weight copy");
  PN_CHECK_ARG(B > 0 && N > 0 && C > 0 && C % 32 == 0 && (K == 64 || K == 128), "pn_panel_finalize: bad sizes");
  PN_CHECK_ARG(panel_rows == 64 || panel_rows == 128, "pn_panel_finalize: panel_rows must be 64 or 128");
  PN_CHECK_ARG(prec == PN_PREC_BF16 || (prec == PN_PREC_BF16X3 && (!use_batch || wf_lo)), "pn_panel_finalize: bad prec / missing lo weights");
  PanelFinArgs a;
  memset(&a, 0, sizeof(a));
  a.pmax = pmax; a.pq = pq; a.sumsq = sumsq; a.a1 = a1;
  a.wf_hi = reinterpret_cast<const __bf16*>(wf_hi); a.wf_lo = reinterpret_cast<const __bf16*>(wf_lo);
  a.tpc = cdiv(N, panel_rows); a.T = B * a.tpc; a.B = B; a.C = C; a.K = K; a.NT = prec == PN_PREC_BF16X3 ? 2 : 1;
  a.n_blocks32 = cdiv(N, 32);
  a.inv_count = 1.0 / ((double)B * (double)N);
  a.gamma = gamma; a.beta = beta; a.mm = mm; a.mv = mv; a.momentum = momentum; a.eps = eps; a.use_batch = use_batch; a.update = update;
  a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift; a.g = g; a.zstar = zstar; a.argq = argq;
  const int slices = B >= 32 ? 4 : (B >= 16 ? 2 : 1);      // the statistics are recomputed per slice: a few, for parallelism over the clouds
  hipLaunchKernelGGL(panel_finalize_kernel, dim3(C / 32, slices), dim3(256), 0, st, a);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

}  // namespace pn
