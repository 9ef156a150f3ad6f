// Generic per-point contraction engine for gfx950 (bf16 MFMA 32x32x16, fp32 accumulate).
//
//   OUT[i, j] = sum_k  A(i, k) * B(j, k)
//
// Three problem shapes share one core:
//   FWD       i = point row (tile of 128 rows of one cloud), j = output channel, k = input channel.
//             A = lazy activation operand (row-major, k contiguous);  B = W[k][j]   (Keras kernel, k-major)
//   BWD_DATA  i = point row, j = input channel of the layer, k = output channel.
//             A = lazy dz operand;                                    B = W[j][k]   (same Keras kernel)
//   WGRAD     i = channel of operand a, j = channel of operand b, k = point row inside a slab.
//             A, B = lazy operands, both read "transposed" (k is the slow index in memory)
//
// LDS image of either operand is T[r][kk] bf16 with kk contiguous and a row pitch of BK+8 elements, which
// makes every ds_read_b128 fragment read and every ds_write_b128 staging write conflict-free
// (MI355X_MICROARCH.md, LDS table: b128 reads are served in 16-lane groups over 64 banks; pitch 144 B or
// 80 B maps rows r..r+15 to 16 distinct 16-byte slots).  Operands are staged through registers because the
// BatchNorm/ReLU (or BN-backward) affine is applied on the way in -- that is the fusion that removes the
// separate normalisation passes of the reference's dataflow.
//
// MFMA lane maps used (cdna_hip_programming.md section 3): v_mfma_f32_32x32x16_bf16, lane l: r = l&31, h = l>>5
//   A fragment element e (0..7) = A[row r][k = 8h+e];   B fragment element e = B[k = 8h+e][col r]
//   C/D register g (0..15)      = D[row (g&3) + 8*(g>>2) + 4h][col r]
#include <vector>
#include "pn_common.h"
#include "pn_dense_wgrad.h"
#include "pn_gemm_core.h"
#include "pn_internal.h"

namespace pn {

template <int BM, int BN, int NS, bool B2>
__global__ __launch_bounds__(256) void wgrad_batch_kernel(const WgradBatch wb) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[GemmLds<BM, BN, NS>::BYTES];
  wgrad_batch_tile<BM, BN, NS, B2>(wb, (int)blockIdx.x, lds_raw);
}

// Round 3: the G W products of the max-pooled layers are a 96-workgroup launch (three 128 x 1024 outputs in 64 x 64 tiles) that leaves
// most of the chip idle for its 10 us, and the dense layers' batched weight-gradient launch (13 us, ~600 workgroups) depends on nothing
// in the backward pass's tail: its workgroups ride on the block ids behind the G W launch's own.
template <int BM, int BN, int NS, bool B2>
__global__ __launch_bounds__(256) void wgrad_batch_dense_kernel(const WgradBatch wb, int n_w, const DenseWgradBatch db) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[GemmLds<BM, BN, NS>::BYTES];
  static_assert(GemmLds<BM, BN, NS>::BYTES >= 16 * 32 * 4, "the dense riders' LDS tile fits");
  if ((int)blockIdx.x >= n_w) {                    // block-uniform
    dense_wgrad_batch_body(db, (int)blockIdx.x - n_w, reinterpret_cast<float(*)[32]>(lds_raw));
    return;
  }
  wgrad_batch_tile<BM, BN, NS, B2>(wb, (int)blockIdx.x, lds_raw);
}

// ---- the kernel ----------------------------------------------------------------------------------------
// 256 threads = 4 waves arranged 2 (rows) x 2 (cols); wave tile (BM/2) x (BN/2) = MT x NT MFMA tiles.
template <int BM, int BN, int NS, int MODE, bool A2, int EPI, bool ADD, bool MASK, bool AH, bool S16, bool W16 = false>
__device__ __forceinline__ void rows_tile_t(const GemmArgs& g, const int bx, const int by, unsigned char* lds_raw) {
  constexpr int BK = (NS == 3) ? 32 : 64;
  constexpr int PITCH = Geo<BK>::PITCH;
  constexpr int MT = BM / 64, NT = BN / 64;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TILE_A = BM * PITCH, TILE_B = BN * PITCH;
  __bf16* Ahi = reinterpret_cast<__bf16*>(lds_raw);
  __bf16* Bhi = Ahi + TILE_A;
  __bf16* Alo = (NS == 3) ? (Bhi + TILE_B) : Ahi;
  __bf16* Blo = (NS == 3) ? (Alo + TILE_A) : Bhi;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wrow0 = wm * WTM, wcol0 = wn * WTN;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

  const int cloud = bx / g.tiles_per_cloud, tin = bx - cloud * g.tiles_per_cloud;

  const int row_in_cloud0 = tin * BM;
  const int nrows = min(BM, g.N - row_in_cloud0);
  const long long row0 = (long long)cloud * g.N + row_in_cloud0;
  const int col0 = by * BN;
  const long long wbase = (long long)cloud * g.w_cloud_stride;
  pn_operand wop;
  wop.s1 = g.w; wop.s2 = nullptr; wop.ca = nullptr; wop.cb = nullptr; wop.cc = nullptr;
  wop.lo = -INFINITY; wop.h16 = 0;
  wop.ld = (MODE == MODE_FWD) ? g.C : g.K;

  // same one-chunk-ahead pipeline as the weight-gradient loop: chunk k+1's loads fly under chunk k's MFMAs
  NatStage<BM, BK, A2, AH> sa;
  TrnStage<BN, BK, false, false> sbT;      // FWD: weights W[k][j], k slow
  NatStage<BN, BK, false, false> sbN;      // BWD: weights W[j][k], k fast
  CopyStage<BN, BK> sbC;                   // W16: the prepared bf16 copy, rows j, k fast (either direction)
  auto issue_chunk = [&](int k0) {
    if (!(g.dbg & 4)) sa.issue(g.a, row0 * g.a.ld, nrows, k0, tid);
    if constexpr (W16) {
      sbC.issue(g.w16, (long long)col0 * g.K, g.K, g.C - col0, k0, tid);
    } else if (MODE == MODE_FWD) {
      if (!(g.dbg & 8)) sbT.issue(wop, wbase + (long long)k0 * g.C + col0, BK, g.C - col0, tid);
    } else {
      sbN.issue(wop, wbase + (long long)col0 * g.K, g.C - col0, k0, tid);
    }
  };
  if (g.K > 0) issue_chunk(0);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    sa.pin();
    sa.template finish<NS>(Ahi, Alo, g.a, nrows, k0, tid);
    if constexpr (W16) {
      sbC.pin();
      sbC.finish(Bhi, g.C - col0, tid);
    } else if (MODE == MODE_FWD) {
      sbT.pin();
      sbT.template finish<NS>(Bhi, Blo, wop, BK, g.C - col0, 0, tid);
    } else {
      sbN.pin();
      sbN.template finish<NS>(Bhi, Blo, wop, g.C - col0, k0, tid);
    }
    __syncthreads();
    if (k0 + BK < g.K) issue_chunk(k0 + BK);
    mma_chunk<MT, NT, BK, NS>(acc, Ahi, Alo, Bhi, Blo, wrow0, wcol0, lane);
    __syncthreads();
  }

  const int r = lane & 31, h = lane >> 5;
  const bool full = (nrows == BM) && (col0 + BN <= g.C);   // block-uniform: whole tile valid
  float* red = reinterpret_cast<float*>(lds_raw);  // tiles are dead after the final barrier

  if (EPI == EPI_STORE) {
    float s1[NT], s2[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int jl = wcol0 + n * 32 + r, j = col0 + jl;
      const bool jv = j < g.C;
      float bias = 0.f, msc = 0.f, msh = 0.f;
      if (jv && g.cloud_bias) bias = g.cloud_bias[(long long)cloud * g.cloud_bias_stride + j];
      if (jv && g.zmask) { msc = g.msc[j]; msh = g.msh[j]; }
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int il0 = wrow0 + m * 32 + 4 * h;
        if (full) {
          if (g.out && !(g.dbg & 1)) epi_store_block<true, ADD, MASK, true, S16>(acc[m][n], g, row0, il0, nrows, j, jv, bias, msc, msh, a1, a2);
          else epi_store_block<true, ADD, MASK, false, S16>(acc[m][n], g, row0, il0, nrows, j, jv, bias, msc, msh, a1, a2);
        } else {
          if (g.out) epi_store_block<false, ADD, MASK, true, S16>(acc[m][n], g, row0, il0, nrows, j, jv, bias, msc, msh, a1, a2);
          else epi_store_block<false, ADD, MASK, false, S16>(acc[m][n], g, row0, il0, nrows, j, jv, bias, msc, msh, a1, a2);
        }
      }
      s1[n] = a1 + __shfl_xor(a1, 32, 64);
      s2[n] = a2 + __shfl_xor(a2, 32, 64);
    }
    if (g.stat_partials && !(g.dbg & 2)) {
      if (h == 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int jl = wcol0 + n * 32 + r;
          red[(wm * 2 + 0) * BN + jl] = s1[n];
          red[(wm * 2 + 1) * BN + jl] = s2[n];
        }
      }
      __syncthreads();
      if (tid < BN && col0 + tid < g.C) {
        float* p = g.stat_partials + (long long)bx * 2 * g.C + col0 + tid;
        p[0] = red[0 * BN + tid] + red[2 * BN + tid];
        p[g.C] = red[1 * BN + tid] + red[3 * BN + tid];
      }
    }
  } else if (EPI == EPI_MAX) {
    float s1[NT], s2[NT], bv[NT];
    int bi[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int j = col0 + wcol0 + n * 32 + r;
      const bool jv = j < g.C;
      const float sg = (jv && g.sgn[j] < 0.f) ? -1.f : 1.f;   // g.sgn may be gamma itself: only its sign is used
      float a1 = 0.f, a2 = 0.f, best = -INFINITY;
      int besti = 0x7fffffff;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {       // branch-free: rows outside the cloud are neutralised with selects
          const int il = wrow0 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const bool ok = full || il < nrows;
          const float v = ok ? acc[m][n][e] : 0.f;
          a1 += v;
          a2 = fmaf(v, v, a2);
          const float t = ok ? sg * v : -INFINITY;
          const bool better = t > best;          // rows ascend with (m, e): first maximum wins
          best = better ? t : best;
          besti = better ? (row_in_cloud0 + il) : besti;
        }
      s1[n] = a1 + __shfl_xor(a1, 32, 64);
      s2[n] = a2 + __shfl_xor(a2, 32, 64);
      const float ob = __shfl_xor(best, 32, 64);
      const int oi = __shfl_xor(besti, 32, 64);
      const bool take = ob > best || (ob == best && oi < besti);
      bv[n] = take ? ob : best; bi[n] = take ? oi : besti;
    }
    if (h == 0) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int jl = wcol0 + n * 32 + r;
        red[(wm * 4 + 0) * BN + jl] = s1[n];
        red[(wm * 4 + 1) * BN + jl] = s2[n];
        red[(wm * 4 + 2) * BN + jl] = bv[n];
        reinterpret_cast<int*>(red)[(wm * 4 + 3) * BN + jl] = bi[n];
      }
    }
    __syncthreads();
    if (tid < BN && col0 + tid < g.C) {
      const int j = col0 + tid;
      if (g.stat_partials) {
        float* p = g.stat_partials + (long long)bx * 2 * g.C + j;
        p[0] = red[0 * BN + tid] + red[4 * BN + tid];
        p[g.C] = red[1 * BN + tid] + red[5 * BN + tid];
      }
      float v0 = red[2 * BN + tid], v1 = red[6 * BN + tid];
      int i0 = reinterpret_cast<int*>(red)[3 * BN + tid], i1 = reinterpret_cast<int*>(red)[7 * BN + tid];
      if (v1 > v0 || (v1 == v0 && i1 < i0)) { v0 = v1; i0 = i1; }
      g.pmax[(long long)bx * g.C + j] = v0;
      g.pidx[(long long)bx * g.C + j] = i0;
    }
  }
}

// the forward form is two registers over the 256 that let two workgroups share a CU: held to that budget
template <int BM, int BN, int NS, int MODE, bool A2, bool B2, int EPI, bool ADD = false, bool MASK = false>
__global__ __launch_bounds__(256, (MODE == MODE_FWD && EPI == EPI_STORE) ? 2 : 1) void gemm_kernel(const GemmArgs g) {
  constexpr int BK = (NS == 3) ? 32 : 64;
  constexpr int LDS_TILES_BYTES = (BM + BN) * Geo<BK>::PITCH * ((NS == 3) ? 2 : 1) * 2;
  constexpr int LDS_EPI_BYTES = 2 * BN * 4 * 4;  // [2 waves rows][BN] x (up to 4 floats)
  constexpr int LDS_BYTES = LDS_TILES_BYTES > LDS_EPI_BYTES ? LDS_TILES_BYTES : LDS_EPI_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
  if constexpr (MODE == MODE_WGRAD) {
    wgrad_tile<BM, BN, NS, A2, B2>(g, blockIdx.x, blockIdx.y, blockIdx.z, lds_raw);
  } else {
    // Block -> (row tile, column tile), 1-D grid of n_row * ncol.  The ncol column tiles of a row tile read the same activation rows;
    // workgroups are dealt round-robin over the 8 XCDs (observed, for speed only), so inside a group of 8 row tiles the id walks
    // column-tile-major with the row tile in its low 3 bits: a row tile's column tiles run on one XCD one after the other and the
    // re-reads of its rows hit that XCD's L2 (with grid (row, col) they were n_row dispatches apart, on any XCD)
    int bx = blockIdx.x, by = 0;
    const int ncol = g.ncol;
    if (ncol > 1) {
      const int lin = blockIdx.x, nrow = (int)gridDim.x / ncol, full = (nrow / 8) * 8 * ncol;
      if (lin < full) {
        const int grp = lin / (8 * ncol), r = lin - grp * 8 * ncol;
        by = r >> 3;
        bx = grp * 8 + (r & 7);
      } else {
        const int l2 = lin - full, rem = nrow - (nrow / 8) * 8;
        bx = (nrow / 8) * 8 + l2 % rem;
        by = l2 / rem;
      }
    }
    // the storage types of the operand and of the epilogue's tensors are block-uniform: one switch, outside every loop
    if (g.a.h16) {
      if constexpr (NS == 1 && EPI == EPI_STORE) {
        if (g.w16) {                       // bf16 activations and the prepared bf16 kernel copy: the model plan's 'bf16' mode
          if (g.store16) rows_tile_t<BM, BN, NS, MODE, A2, EPI, ADD, MASK, true, true, true>(g, bx, by, lds_raw);
          else rows_tile_t<BM, BN, NS, MODE, A2, EPI, ADD, MASK, true, false, true>(g, bx, by, lds_raw);
          return;
        }
      }
      if (EPI == EPI_STORE && g.store16) rows_tile_t<BM, BN, NS, MODE, A2, EPI, ADD, MASK, true, EPI == EPI_STORE>(g, bx, by, lds_raw);
      else rows_tile_t<BM, BN, NS, MODE, A2, EPI, ADD, MASK, true, false>(g, bx, by, lds_raw);
    } else {
      if (EPI == EPI_STORE && g.store16) rows_tile_t<BM, BN, NS, MODE, A2, EPI, ADD, MASK, false, EPI == EPI_STORE>(g, bx, by, lds_raw);
      else rows_tile_t<BM, BN, NS, MODE, A2, EPI, ADD, MASK, false, false>(g, bx, by, lds_raw);
    }
  }
}

// ---- host-side dispatch ----------------------------------------------------------------------------------
template <int BM, int BN, int NS, int MODE, bool A2, bool B2, int EPI, bool ADD = false, bool MASK = false>
static int launch(const GemmArgs& g, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL((gemm_kernel<BM, BN, NS, MODE, A2, B2, EPI, ADD, MASK>), grid, dim3(256), 0, st, g);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int check_operand(const pn_operand* o, const char* name) {
  PN_CHECK_ARG(o && o->s1, "%s: null operand", name);
  PN_CHECK_ARG(aligned16(o->s1) && (!o->s2 || aligned16(o->s2)), "%s: s1/s2 must be 16-byte aligned", name);
  PN_CHECK_ARG((!o->ca || aligned16(o->ca)) && (!o->cb || aligned16(o->cb)) && (!o->cc || aligned16(o->cc)),
               "%s: coefficient vectors must be 16-byte aligned", name);
  PN_CHECK_ARG(o->ld > 0 && o->ld % 4 == 0, "%s: ld must be a positive multiple of 4", name);
  PN_CHECK_ARG(o->h16 == 0 || (o->h16 == 1 && o->ld % 8 == 0), "%s: h16 must be 0 or 1, and 16-bit rows need ld %% 8 == 0", name);
  return PN_OK;
}

template <int MODE, bool A2, int EPI, bool ADD = false, bool MASK = false>
static int dispatch_rows(const GemmArgs& g_in, int prec, hipStream_t st) {
  GemmArgs g = g_in;
  static const bool narrow = getenv("PN_GEMM_NARROW") != nullptr;   // experiment: 128x64 tiles everywhere (2x the blocks)
  const bool wide = (g.C % 128 == 0) && !narrow;
  if (EPI == EPI_MAX) {
    g.ncol = cdiv(g.C, 128);
    dim3 grid(g.B * g.tiles_per_cloud * g.ncol);
    if (prec == PN_PREC_BF16X3) return launch<128, 128, 3, MODE, A2, false, EPI>(g, grid, st);
    return launch<128, 128, 1, MODE, A2, false, EPI>(g, grid, st);
  }
  // a handful of row tiles (the Gram-sized products of the max-pool backward: 128 rows): 64-wide column tiles double the workgroups
  const bool few = (long long)g.B * g.tiles_per_cloud * (g.C / 128) < 64;
  if (wide && !few) {
    g.ncol = g.C / 128;
    dim3 grid(g.B * g.tiles_per_cloud * g.ncol);
    if (prec == PN_PREC_BF16X3) return launch<128, 128, 3, MODE, A2, false, EPI, ADD, MASK>(g, grid, st);
    return launch<128, 128, 1, MODE, A2, false, EPI, ADD, MASK>(g, grid, st);
  }
  g.ncol = cdiv(g.C, 64);
  dim3 grid(g.B * g.tiles_per_cloud * g.ncol);
  if (prec == PN_PREC_BF16X3) return launch<128, 64, 3, MODE, A2, false, EPI, ADD, MASK>(g, grid, st);
  return launch<128, 64, 1, MODE, A2, false, EPI, ADD, MASK>(g, grid, st);
}

template <bool A2>
static int dispatch_bwd(const GemmArgs& g, int prec, hipStream_t st) {
  const bool ha = g.addend != nullptr, hm = g.zmask != nullptr;
  if (ha && hm) return dispatch_rows<MODE_BWD, A2, EPI_STORE, true, true>(g, prec, st);
  if (ha) return dispatch_rows<MODE_BWD, A2, EPI_STORE, true, false>(g, prec, st);
  if (hm) return dispatch_rows<MODE_BWD, A2, EPI_STORE, false, true>(g, prec, st);
  return dispatch_rows<MODE_BWD, A2, EPI_STORE, false, false>(g, prec, st);
}

int conv_fwd(const pn_operand* x, const float* w, long long wcs, int B, int N, int K, int C, const float* cloud_bias,
             float* z, float* stat_partials, int prec, hipStream_t st, const void* w16) {
  PN_TRY(check_operand(x, "pn_conv_fwd.x"));
  PN_CHECK_ARG(!x->s2, "pn_conv_fwd: the forward operand has no second source");
  PN_CHECK_ARG(B > 0 && N > 0, "pn_conv_fwd: B and N must be positive (B=%d N=%d)", B, N);
  PN_CHECK_ARG(K >= 64 && K % 64 == 0, "pn_conv_fwd: K must be a multiple of 64 (K=%d)", K);
  PN_CHECK_ARG(C >= 64 && C % 64 == 0, "pn_conv_fwd: C must be a multiple of 64 (C=%d)", C);
  PN_CHECK_ARG(x->ld >= K, "pn_conv_fwd: x.ld < K");
  PN_CHECK_ARG(w && aligned16(w), "pn_conv_fwd: w null or unaligned");
  const int store16 = (prec & PN_STORE_BF16) ? 1 : 0;
  prec &= ~PN_STORE_BF16;
  PN_CHECK_ARG(prec == PN_PREC_BF16 || prec == PN_PREC_BF16X3, "pn_conv_fwd: bad prec %d", prec);
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.store16 = store16;
  g.a = *x; g.w = w; g.w_cloud_stride = wcs; g.B = B; g.N = N; g.K = K; g.C = C;
  g.tiles_per_cloud = cdiv(N, 128);
  g.out = z; g.cloud_bias = cloud_bias; g.cloud_bias_stride = C; g.stat_partials = stat_partials;
  if (w16 && wcs == 0 && prec == PN_PREC_BF16 && x->h16 && (reinterpret_cast<uintptr_t>(w16) & 15) == 0 && K % 8 == 0)
    g.w16 = reinterpret_cast<const unsigned short*>(w16);
  static const int dbg = getenv("PN_GEMM_DBG") ? atoi(getenv("PN_GEMM_DBG")) : 0;
  g.dbg = dbg;
  return dispatch_rows<MODE_FWD, false, EPI_STORE>(g, prec, st);
}

int conv_fwd_max(const pn_operand* x, const float* w, int B, int N, int K, int C, const float* sgn, float* pmax, int* pidx,
                 float* stat_partials, int prec, hipStream_t st) {
  PN_TRY(check_operand(x, "pn_conv_fwd_max.x"));
  PN_CHECK_ARG(!x->s2, "pn_conv_fwd_max: the forward operand has no second source");
  PN_CHECK_ARG(B > 0 && N > 0, "pn_conv_fwd_max: B and N must be positive");
  PN_CHECK_ARG(K >= 64 && K % 64 == 0, "pn_conv_fwd_max: K must be a multiple of 64 (K=%d)", K);
  PN_CHECK_ARG(C >= 128 && C % 128 == 0, "pn_conv_fwd_max: C must be a multiple of 128 (C=%d)", C);
  PN_CHECK_ARG(x->ld >= K, "pn_conv_fwd_max: x.ld < K");
  PN_CHECK_ARG(w && aligned16(w) && sgn && pmax && pidx, "pn_conv_fwd_max: null pointer");
  PN_CHECK_ARG(prec == PN_PREC_BF16 || prec == PN_PREC_BF16X3, "pn_conv_fwd_max: bad prec %d", prec);
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a = *x; g.w = w; g.B = B; g.N = N; g.K = K; g.C = C;
  g.tiles_per_cloud = cdiv(N, 128);
  g.sgn = sgn; g.pmax = pmax; g.pidx = pidx; g.stat_partials = stat_partials;
  return dispatch_rows<MODE_FWD, false, EPI_MAX>(g, prec, st);
}

int conv_bwd_data(const pn_operand* dz, const float* w, long long wcs, int B, int N, int K, int C, const float* addend,
                  const float* zmask, const float* msc, const float* msh, float* out, float* stat_partials, int prec,
                  hipStream_t st, const void* w16, const float* col_bias) {
  PN_TRY(check_operand(dz, "pn_conv_bwd_data.dz"));
  PN_CHECK_ARG(B > 0 && N > 0, "pn_conv_bwd_data: B and N must be positive");
  PN_CHECK_ARG(K >= 64 && K % 64 == 0, "pn_conv_bwd_data: K must be a multiple of 64 (K=%d)", K);
  PN_CHECK_ARG(C >= 64 && C % 64 == 0, "pn_conv_bwd_data: C must be a multiple of 64 (C=%d)", C);
  PN_CHECK_ARG(dz->ld >= K, "pn_conv_bwd_data: dz.ld < K");
  PN_CHECK_ARG(w && aligned16(w), "pn_conv_bwd_data: w null or unaligned");
  PN_CHECK_ARG(!zmask || (msc && msh), "pn_conv_bwd_data: zmask needs msc and msh");
  const int store16 = (prec & PN_STORE_BF16) ? 1 : 0;
  prec &= ~PN_STORE_BF16;
  PN_CHECK_ARG(prec == PN_PREC_BF16 || prec == PN_PREC_BF16X3, "pn_conv_bwd_data: bad prec %d", prec);
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.store16 = store16;
  g.a = *dz; g.w = w; g.w_cloud_stride = wcs; g.B = B; g.N = N; g.K = K; g.C = C;
  g.tiles_per_cloud = cdiv(N, 128);
  g.out = out; g.addend = addend; g.zmask = zmask; g.msc = msc; g.msh = msh; g.stat_partials = stat_partials;
  g.cloud_bias = col_bias; g.cloud_bias_stride = 0;
  if (w16 && wcs == 0 && prec == PN_PREC_BF16 && dz->h16 && (reinterpret_cast<uintptr_t>(w16) & 15) == 0 && K % 8 == 0)
    g.w16 = reinterpret_cast<const unsigned short*>(w16);
  if (dz->s2) return dispatch_bwd<true>(g, prec, st);
  return dispatch_bwd<false>(g, prec, st);
}

template <int BM, int BN>
static int dispatch_wgrad(const GemmArgs& g, bool b2, int prec, dim3 grid, hipStream_t st) {
  if (prec == PN_PREC_BF16X3) {
    if (b2) return launch<BM, BN, 3, MODE_WGRAD, false, true, EPI_SLAB>(g, grid, st);
    return launch<BM, BN, 3, MODE_WGRAD, false, false, EPI_SLAB>(g, grid, st);
  }
  if (b2) return launch<BM, BN, 1, MODE_WGRAD, false, true, EPI_SLAB>(g, grid, st);
  return launch<BM, BN, 1, MODE_WGRAD, false, false, EPI_SLAB>(g, grid, st);
}

static int wgrad_args(const pn_operand* a, const pn_operand* b, int B, int N, int Ci, int Cj, int slab_rows, float* slabs, int prec,
                      int colsum, GemmArgs& g) {
  PN_TRY(check_operand(a, "pn_conv_wgrad.a"));
  PN_TRY(check_operand(b, "pn_conv_wgrad.b"));
  PN_CHECK_ARG(!a->s2, "pn_conv_wgrad: operand a has no second source");
  PN_CHECK_ARG(B > 0 && N > 0, "pn_conv_wgrad: B and N must be positive");
  PN_CHECK_ARG(Ci >= 64 && Ci % 64 == 0 && Cj >= 64 && Cj % 64 == 0, "pn_conv_wgrad: Ci, Cj must be multiples of 64 (%d, %d)",
               Ci, Cj);
  PN_CHECK_ARG(slab_rows >= 64 && slab_rows % 64 == 0, "pn_conv_wgrad: slab_rows must be a multiple of 64");
  PN_CHECK_ARG(a->ld >= Ci && b->ld >= Cj, "pn_conv_wgrad: ld too small");
  PN_CHECK_ARG(slabs != nullptr, "pn_conv_wgrad: null slabs");
  prec &= ~PN_STORE_BF16;
  PN_CHECK_ARG(prec == PN_PREC_BF16 || prec == PN_PREC_BF16X3, "pn_conv_wgrad: bad prec %d", prec);
  memset(&g, 0, sizeof(g));
  g.a = *a; g.b = *b; g.B = B; g.N = N; g.K = slab_rows; g.C = Cj; g.Ci = Ci;
  g.tiles_per_cloud = cdiv(N, slab_rows);
  g.out = slabs; g.colsum = colsum;
  return PN_OK;
}

template <int BM, int BN>
static int launch_wgrad_batch(const WgradBatch& wb, bool b2, int prec, int blocks, hipStream_t st) {
  if (prec == PN_PREC_BF16X3) {
    if (b2) hipLaunchKernelGGL((wgrad_batch_kernel<BM, BN, 3, true>), dim3(blocks), dim3(256), 0, st, wb);
    else hipLaunchKernelGGL((wgrad_batch_kernel<BM, BN, 3, false>), dim3(blocks), dim3(256), 0, st, wb);
  } else {
    if (b2) hipLaunchKernelGGL((wgrad_batch_kernel<BM, BN, 1, true>), dim3(blocks), dim3(256), 0, st, wb);
    else hipLaunchKernelGGL((wgrad_batch_kernel<BM, BN, 1, false>), dim3(blocks), dim3(256), 0, st, wb);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// one job of tile shape bm x bn as a batch of its own (for launches of other files that carry weight-gradient workgroups: pn_gemm_core.h)
int wgrad_batch_one(const WgradDesc& q, int bm, int bn, WgradBatch& wb, int& blocks) {
  memset(&wb, 0, sizeof(wb));
  PN_CHECK_ARG(q.Ci % bm == 0 && q.Cj % bn == 0, "wgrad_batch_one: Ci %% %d, Cj %% %d", bm, bn);
  PN_TRY(wgrad_args(&q.a, &q.b, q.B, q.N, q.Ci, q.Cj, q.slab_rows, q.slabs, q.prec, q.colsum, wb.g[0]));
  const int nslab = q.B * wb.g[0].tiles_per_cloud, ny = q.Ci / bm, nz = q.Cj / bn;
  blocks = nslab * ny * nz;
  wb.nslab[0] = nslab; wb.ny[0] = ny; wb.blk_end[0] = blocks; wb.n = 1;
  return PN_OK;
}

// jobs of the same tile shape / operand form / precision share a launch (up to WGRAD_BATCH_MAX each)
int conv_wgrad_batch(const WgradDesc* jobs, int n, hipStream_t st, const DenseWgradJob* dense_riders, int n_dense, bool* rode) {
  if (rode) *rode = false;
  std::vector<char> done(n > 0 ? n : 0, 0);
  // shape 3 (round 3, an experiment behind PN_WGRAD_WIDE=1): 128 x 256 output tiles for the wide jobs of the bf16 mode (seg_l2: 512 x
  // 256) -- a tile re-stages its rows of both operands, so with 128 x 128 tiles that job reads its operands three times over (402 MB
  // for 134 MB at B=32, N=2048).  Measured at C3: 1.519 ms/step with the wide tiles against 1.503 without -- the re-reads are served
  // by the L2 / Infinity Cache, and half the workgroups with 227 registers each lose more than the saved staging wins.  Off by default.
  static const bool wide_ok = getenv("PN_WGRAD_WIDE") && atoi(getenv("PN_WGRAD_WIDE")) == 1;
  auto key = [&](const WgradDesc& q) {
    const bool x3 = (q.prec & ~PN_STORE_BF16) == PN_PREC_BF16X3;
    const bool wide = wide_ok && !q.small_tiles && !x3 && q.a.h16 && q.b.h16 && q.Ci % 128 == 0 && q.Cj % 256 == 0;
    const int shape = q.small_tiles ? 2 : wide ? 3 : (q.Ci % 128 == 0 && q.Cj % 128 == 0) ? 0 : (q.Cj % 128 == 0 ? 1 : 2);
    return shape * 8 + (q.b.s2 ? 4 : 0) + (x3 ? 1 : 0);
  };
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    const int k = key(jobs[i]);
    WgradBatch wb;
    memset(&wb, 0, sizeof(wb));
    long long blocks = 0;
    for (int j = i; j < n && wb.n < WGRAD_BATCH_MAX; ++j) {
      if (done[j] || key(jobs[j]) != k) continue;
      const WgradDesc& q = jobs[j];
      GemmArgs& g = wb.g[wb.n];
      PN_TRY(wgrad_args(&q.a, &q.b, q.B, q.N, q.Ci, q.Cj, q.slab_rows, q.slabs, q.prec, q.colsum, g));
      const int shape = k / 8;
      const int bm = (shape == 0 || shape == 3) ? 128 : 64, bn = shape == 3 ? 256 : (shape == 2 ? 64 : 128);
      const int nslab = q.B * g.tiles_per_cloud, ny = q.Ci / bm, nz = q.Cj / bn;
      blocks += (long long)nslab * ny * nz;
      PN_CHECK_ARG(blocks < (1ll << 30), "pn_conv_wgrad: batch too large");
      wb.nslab[wb.n] = nslab; wb.ny[wb.n] = ny; wb.blk_end[wb.n] = (int)blocks;
      ++wb.n;
      done[j] = 1;
    }
    const bool b2 = (k & 4) != 0;
    const int prec = (k & 1) ? PN_PREC_BF16X3 : PN_PREC_BF16;
    if (dense_riders && n_dense > 0 && rode && !*rode && k / 8 == 2 && !b2 && prec == PN_PREC_BF16X3) {
      DenseWgradBatch db;
      int dblocks = 0;
      PN_TRY(make_dense_wgrad_batch(dense_riders, n_dense, db, dblocks));
      hipLaunchKernelGGL((wgrad_batch_dense_kernel<64, 64, 3, false>), dim3((int)blocks + dblocks), dim3(256), 0, st, wb, (int)blocks, db);
      PN_CHECK_LAUNCH();
      *rode = true;
      continue;
    }
    switch (k / 8) {
      case 0: PN_TRY((launch_wgrad_batch<128, 128>(wb, b2, prec, (int)blocks, st))); break;
      case 1: PN_TRY((launch_wgrad_batch<64, 128>(wb, b2, prec, (int)blocks, st))); break;
      case 3:
        if (b2) hipLaunchKernelGGL((wgrad_batch_kernel<128, 256, 1, true>), dim3((int)blocks), dim3(256), 0, st, wb);
        else hipLaunchKernelGGL((wgrad_batch_kernel<128, 256, 1, false>), dim3((int)blocks), dim3(256), 0, st, wb);
        PN_CHECK_LAUNCH();
        break;
      default: PN_TRY((launch_wgrad_batch<64, 64>(wb, b2, prec, (int)blocks, st))); break;
    }
  }
  return PN_OK;
}

int conv_wgrad(const pn_operand* a, const pn_operand* b, int B, int N, int Ci, int Cj, int slab_rows, float* slabs, int prec,
               hipStream_t st, int colsum) {
  PN_TRY(check_operand(a, "pn_conv_wgrad.a"));
  PN_TRY(check_operand(b, "pn_conv_wgrad.b"));
  PN_CHECK_ARG(!a->s2, "pn_conv_wgrad: operand a has no second source");
  PN_CHECK_ARG(B > 0 && N > 0, "pn_conv_wgrad: B and N must be positive");
  PN_CHECK_ARG(Ci >= 64 && Ci % 64 == 0 && Cj >= 64 && Cj % 64 == 0, "pn_conv_wgrad: Ci, Cj must be multiples of 64 (%d, %d)",
               Ci, Cj);
  PN_CHECK_ARG(slab_rows >= 64 && slab_rows % 64 == 0, "pn_conv_wgrad: slab_rows must be a multiple of 64");
  PN_CHECK_ARG(a->ld >= Ci && b->ld >= Cj, "pn_conv_wgrad: ld too small");
  PN_CHECK_ARG(slabs != nullptr, "pn_conv_wgrad: null slabs");
  prec &= ~PN_STORE_BF16;
  PN_CHECK_ARG(prec == PN_PREC_BF16 || prec == PN_PREC_BF16X3, "pn_conv_wgrad: bad prec %d", prec);
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a = *a; g.b = *b; g.B = B; g.N = N; g.K = slab_rows; g.C = Cj; g.Ci = Ci;
  g.tiles_per_cloud = cdiv(N, slab_rows);
  g.out = slabs; g.colsum = colsum;
  const bool b2 = b->s2 != nullptr;
  const int nslab = B * g.tiles_per_cloud;
  if (Ci % 128 == 0 && Cj % 128 == 0) return dispatch_wgrad<128, 128>(g, b2, prec, dim3(nslab, Ci / 128, Cj / 128), st);
  if (Cj % 128 == 0) return dispatch_wgrad<64, 128>(g, b2, prec, dim3(nslab, Ci / 64, Cj / 128), st);
  return dispatch_wgrad<64, 64>(g, b2, prec, dim3(nslab, Ci / 64, Cj / 64), st);
}

}  // namespace pn
