// Per-cloud dense layers (rows = B clouds): DenseLayer (pointnet/PointNet.py:597-679), the T-Net tail
// X @ w + b (PointNet.py:436-442), the classification softmax + loss.  These are weight-streaming, latency
// bound problems (M = batch size): fp32 on the vector ALU, split over K for parallelism, reduced in a
// fixed order so results are bitwise reproducible.
#include "pn_common.h"
#include "pn_dense_wgrad.h"
#include "pn_internal.h"
#include "pn_loss_bodies.h"

namespace pn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---- one launch per dense layer ---------------------------------------------------------------------------------------
// out[r][j] = sum_k x[r][k] * W(k, j)  (+ bias, BatchNormalization over the R rows, ReLU, inverted dropout).
// Grid = (column blocks of 32) x (K splits).  The products run on the matrix cores in bf16 hi/lo split form (see the kernel);
// operands are read straight from global memory in the MFMA register layout.  The splits of one column block meet in-launch:
// every split writes its 32x32 partial tile, then draws a ticket; the last arriver sums the tiles in a fixed order and
// finishes the layer for those 32 columns (statistics are per column, so column blocks are independent).  This replaces
// the separate reduce/normalise launch (~6 us in the step's graph) by ~1-2 us of tail work in one block per 32 columns.
// Hand-off protocol: cdna_hip_programming.md, "In-launch split-K reduction", write-through form: the partial tiles are stored
// sc1 (relaxed agent-scope atomic stores), every wave drains them, barrier, one lane draws a relaxed ticket; the last arriver
// reads every tile with sc1 loads (relaxed agent-scope atomic loads) -- no release or acquire fence.  The counter is re-zeroed
// by the last arriver AND by the caller before each pass.
//   TRANS = false: W(k, j) = w[k * ldw + j]   (forward: the layer's kernel)
//   TRANS = true : W(k, j) = w[j * ldw + k]   (backward data: dx = dz . W^T read from the same kernel, no transpose pass)
constexpr int DL_COLS = 32;     // columns per block
constexpr int DL_SLICES = 8;    // k-slices per block
constexpr int DL_KPS = 16;      // k per slice and step: 16 weight loads in flight per thread
constexpr int DL_KSTEP = DL_SLICES * DL_KPS;   // 128 k per LDS step
constexpr int DL_ROWS = 32;     // rows per register chunk
constexpr int DL_MAX_SPLITS = 8;

struct DenseArgs {
  const float* x; int ldx;
  const float* w; int ldw;
  int R, K, C, split_len, nsplit;
  float* partial;             // [nsplit][R][C]
  unsigned* counters;         // one per column block, zero on entry, zero on exit
  const float *bias, *gamma, *beta;
  float *mm, *mv;
  float momentum, eps;
  int bn_mode, act;           // bn_mode: 0 none, 1 batch statistics (+ moving update), 2 moving statistics; act: 0 none, 1 relu
  const unsigned char* keep; float keep_scale;
  float *z_out, *a_out, *mean_o, *invstd_o;
  // Backward tail (TRANS launches of a backward chain, R <= 32; bt_dz == NULL: none).  The launch computes dx = dz_above . W_above^T,
  // which IS d(activation) of the layer below; the workgroup that finishes 32 of its columns goes straight on through that layer's
  // dropout -> ReLU -> BatchNormalization backward (all per column: nothing crosses the column block) and leaves its dz, dgamma,
  // dbeta / dbias -- the next launch of the chain is the next TRANS product, with no launch in between (pn_model.hip: bwd_chain)
  const float *bt_z, *bt_gamma, *bt_beta, *bt_mean, *bt_invstd;
  const unsigned char* bt_keep; float bt_keep_scale;
  int bt_mode, bt_act;        // as bn_mode / act, of the layer below
  float *bt_dz, *bt_dgamma, *bt_dbeta, *bt_dbias;
};

// How the contraction is split over workgroups.  A launch of these layers is a chain of memory round trips; the in-launch meeting of
// the K splits adds three of them (partial tiles out, ticket, partial tiles back in).  Measured (round 3, C2 step, hipGraph):
//   128 k per split (eight splits of K = 1024)                         0.788 ms/step
//   one workgroup per 32 columns walks ALL of K <= 1024, every operand load of a wave requested up front (no partial tiles, no
//   ticket; PN_DENSE_KSPLIT=1024)                                     0.836 ms/step
// -- the meeting costs less than the serial work it saves: one workgroup then converts 8 x the operands to bf16 hi + lo (3 vector
// instructions per element, 768 per SIMD at K = 1024) behind 8 x the bytes through one CU's load path.  So the splits stay; what the
// rewrite kept is the load ring below (16-byte loads, every load of a round in flight before the first conversion) and the mask /
// bias loads that no longer sit in front of the operand loads: 0.823 -> 0.788 ms/step.  PN_DENSE_KSPLIT=<k per split> for A/B runs.
static inline int dl_ksplit_env() {
  static const int v = (getenv("PN_DENSE_KSPLIT") && atoi(getenv("PN_DENSE_KSPLIT")) >= 1) ? atoi(getenv("PN_DENSE_KSPLIT")) : 0;
  return v;
}
static inline int dl_nsplit(int K) {
  const int env = dl_ksplit_env();
  const int per_split = env ? env : DL_KSTEP;
  const int s = cdiv(K, per_split);
  return s < 1 ? 1 : (s > DL_MAX_SPLITS ? DL_MAX_SPLITS : s);
}
static inline int dl_split_len(int K) { return cdiv(cdiv(K, dl_nsplit(K)), DL_KSTEP) * DL_KSTEP; }

// Two layers of one grid shape (same K and C) may share a launch: blockIdx.z picks the job (pn_model.hip: the classification head's
// first layer and the global-feature half of seg_l1 both consume the pooled feature vector).
// DEPTH: 64-k steps a wave keeps in flight (2: split form, one step ahead of the one being multiplied; 4 / 8: the whole-K forms);
// VEC: x rows (and, TRANS, kernel rows) are read as 16-byte loads (ldx, ldw % 4 == 0, 16-byte aligned bases, K % 16 == 0)
template <bool TRANS, int DEPTH, bool VEC>
__global__ __launch_bounds__(256) void dense_layer_kernel(const DenseArgs a0, const DenseArgs a1) {
  const DenseArgs& a = blockIdx.z ? a1 : a0;
  // one LDS object (a second one beside a staging array can cost a full vmcnt drain per step)
  __shared__ __attribute__((aligned(16))) float lds[4 * DL_ROWS * DL_COLS + 16 * 32 + 4];
  float* red = lds;                                        // [wave][row][col]
  float* fin = lds + 4 * DL_ROWS * DL_COLS;                // finalize scratch [16][32]
  unsigned* flag = reinterpret_cast<unsigned*>(fin + 16 * 32);
  const int tid = threadIdx.x, c = tid & 31, s = tid >> 5;
  const int j = blockIdx.x * DL_COLS + c;
  const int jc = j < a.C ? j : a.C - 1;        // clamped: weight loads are unconditional
  const int ks = blockIdx.y;
  const int kbeg = ks * a.split_len, kend = min(a.K, kbeg + a.split_len);
  const bool single = a.nsplit == 1;
  const bool small = a.R <= DL_ROWS;           // block-uniform: one row chunk, the tail keeps z in registers
  float own[DL_ROWS / DL_SLICES];
  // backward tail: what it reads of the layer below does not depend on this launch's products -- requested now by every workgroup
  // (only the last arriver of a column block uses them: four small loads wasted elsewhere, one memory round trip saved in the tail)
  // ... and so are the layer's own BatchNormalization parameters and dropout mask, which the last arriver needs only after its
  // reductions (there they would be one more memory round trip at the tail of the launch)
  float pf_g = 1.f, pf_b = 0.f, pf_mm = 0.f, pf_mv = 1.f;
  const float pf_bias = a.bias ? a.bias[jc] : 0.f;        // (at the tail it would be one more dependent round trip)
  // (the mask bytes stay as loaded until their use at the end of the kernel: combined into a bit mask HERE they would be waited for
  //  here -- a whole memory round trip in front of the operand loads)
  unsigned char pf_kb[DL_ROWS / DL_SLICES] = {1, 1, 1, 1};
  if (a.bn_mode) { pf_g = a.gamma[jc]; pf_b = a.beta[jc]; pf_mm = a.mm[jc]; pf_mv = a.mv[jc]; }
  if (a.keep && a.a_out && a.R <= DL_ROWS) {
#pragma unroll
    for (int i = 0; i < DL_ROWS / DL_SLICES; ++i) pf_kb[i] = a.keep[(long long)min(s + DL_SLICES * i, a.R - 1) * a.C + jc];
  }
  float bt_zz[DL_ROWS / DL_SLICES], bt_mu = 0.f, bt_is = 1.f, bt_g = 1.f, bt_b = 0.f;
  unsigned char bt_kb[DL_ROWS / DL_SLICES] = {1, 1, 1, 1};
  if (TRANS && a.bt_dz) {
#pragma unroll
    for (int i = 0; i < DL_ROWS / DL_SLICES; ++i) bt_zz[i] = a.bt_z[(long long)min(s + DL_SLICES * i, a.R - 1) * a.C + jc];     // unconditional, clamped
    if (a.bt_keep) {
#pragma unroll
      for (int i = 0; i < DL_ROWS / DL_SLICES; ++i) bt_kb[i] = a.bt_keep[(long long)min(s + DL_SLICES * i, a.R - 1) * a.C + jc];
    }
    if (a.bt_mode) { bt_mu = a.bt_mean[jc]; bt_is = a.bt_invstd[jc]; bt_g = a.bt_gamma[jc]; bt_b = a.bt_beta[jc]; }
  }

  // ---- products on the matrix cores -----------------------------------------------------------------------------------------
  // z tile (32 rows x 32 columns) = x (32 x k) . W (k x 32), one v_mfma_f32_32x32x16_bf16 triple per 16 k: both operands are split
  // into bf16 hi + lo on the fly and the three significant products (lo*hi, hi*lo, hi*hi) accumulate in fp32 -- fp32-grade results
  // (the dropped lo*lo term is 2^-16 of a product) without the LDS broadcast of x that bounded the vector-ALU form (~4 us per 128 k).
  // Operands go straight from global memory to the MFMA layout: lane (r = lane % 32, g = lane / 32) supplies row r / column r and
  // the 8 consecutive k at 8g.  Wave w owns the k16-steps w, w+4, ... of the block's k range; the four waves' accumulators are
  // combined through LDS in a fixed order.
  const int lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 31, lg = lane >> 5;
  const int jw = blockIdx.x * DL_COLS + lr;                 // this lane's column as the B operand
  const int jwc = jw < a.C ? jw : a.C - 1;
  for (int rc = 0; rc < a.R; rc += DL_ROWS) {
    const int nr = min(DL_ROWS, a.R - rc);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int xr = rc + (lr < nr ? lr : nr - 1);
    // Operand loads of this wave's 64-k steps: DEPTH steps are in flight at any time (slots of a register ring, statically indexed).
    // Addresses are clamped and unconditional; what lies beyond the k range or the row / column count is zeroed at the conversion.
    float xa[DEPTH][8], wb[DEPTH][8];
    auto issue = [&](float (&xv)[8], float (&wv_)[8], int k0) {      // k0: first k of this wave's step
      const int kk = k0 + 8 * lg;
      if constexpr (VEC) {
        const int kc = min(kk, kend - 8);                           // (K % 16 == 0: a step is inside the range or wholly beyond it)
        const float4 t0 = *reinterpret_cast<const float4*>(a.x + (long long)xr * a.ldx + kc);
        const float4 t1 = *reinterpret_cast<const float4*>(a.x + (long long)xr * a.ldx + kc + 4);
        xv[0] = t0.x; xv[1] = t0.y; xv[2] = t0.z; xv[3] = t0.w; xv[4] = t1.x; xv[5] = t1.y; xv[6] = t1.z; xv[7] = t1.w;
        if constexpr (TRANS) {
          const float4 u0 = *reinterpret_cast<const float4*>(a.w + (long long)jwc * a.ldw + kc);
          const float4 u1 = *reinterpret_cast<const float4*>(a.w + (long long)jwc * a.ldw + kc + 4);
          wv_[0] = u0.x; wv_[1] = u0.y; wv_[2] = u0.z; wv_[3] = u0.w; wv_[4] = u1.x; wv_[5] = u1.y; wv_[6] = u1.z; wv_[7] = u1.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) wv_[e] = a.w[(long long)(kc + e) * a.ldw + jwc];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = kk + e;
          const long long kc = k < kend ? k : kend - 1;
          xv[e] = a.x[(long long)xr * a.ldx + kc];
          wv_[e] = TRANS ? a.w[(long long)jwc * a.ldw + kc] : a.w[kc * a.ldw + jwc];
        }
      }
    };
    const int kw = kbeg + 16 * wv;                          // this wave's steps: kw, kw + 64, ...
    const int nsteps = kw < kend ? (kend - kw + 63) / 64 : 0;
    // Rounds of DEPTH steps: every load of a round is requested before the first conversion (one memory round trip per round: K <= 512
    // is one round, K = 1024 two).  No load is issued between the conversions of a round: with more than 63 loads in flight the wait
    // counter saturates and the compiler's waits inside a loop that also issues become "wait for everything" per step.
    for (int base = 0; base < nsteps; base += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
        if (base + d < nsteps) issue(xa[d], wb[d], kw + 64 * (base + d));     // wave-uniform
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int i = base + d;
        if (i >= nsteps) break;                             // wave-uniform
        const int k0 = kw + 64 * i;
        bf16x8 ah, al, bh, bl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool kv = k0 + 8 * lg + e < kend;
          const float xv = (kv && lr < nr) ? xa[d][e] : 0.f;
          const float wvv = (kv && jw < a.C) ? wb[d][e] : 0.f;
          ah[e] = (__bf16)xv;
          al[e] = (__bf16)(xv - (float)ah[e]);
          bh[e] = (__bf16)wvv;
          bl[e] = (__bf16)(wvv - (float)bh[e]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      }
    }
    // combine the four waves in a fixed order: acc[e] is row (e & 3) + 8 (e >> 2) + 4 g, column r
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wv * DL_ROWS + (e & 3) + 8 * (e >> 2) + 4 * lg) * DL_COLS + lr] = acc[e];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DL_ROWS / DL_SLICES; ++i) {
      const int r = s + DL_SLICES * i;
      const float t = (red[(0 * DL_ROWS + r) * DL_COLS + c] + red[(1 * DL_ROWS + r) * DL_COLS + c]) +
                      (red[(2 * DL_ROWS + r) * DL_COLS + c] + red[(3 * DL_ROWS + r) * DL_COLS + c]);
      own[i] = t;      // rows s, s+8, s+16, s+24 of this chunk: exactly the rows this thread finishes below
      // write-through (sc1) store: visible to every XCD once drained, so the meeting below needs no release fence
      if (!(single && small) && r < nr && j < a.C)
        __hip_atomic_store(&a.partial[((long long)ks * a.R + rc + r) * a.C + j], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // ---- the splits of this column block meet here ----
  if (!single) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its partial-tile stores
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(a.counters + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned last = (t == (unsigned)a.nsplit - 1u) ? 1u : 0u;
      if (last) __hip_atomic_store(a.counters + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      *flag = last;
    }
    __syncthreads();
    if (*flag == 0u) return;
  } else {
    __syncthreads();     // the partial tile written above is re-read by other threads of this block below
  }

  // ---- finish the layer for these 32 columns: 32 columns x 8 row partitions, fixed-order reductions ----
  const int R = a.R, C = a.C, nks = a.nsplit;
  const int tx = c, ty = s;
  constexpr int RP = DL_SLICES;
  const bool jv = j < C;
  const float b = jv ? pf_bias : 0.f;
  float zr[DL_ROWS / RP];        // small: this thread's rows ty, ty+8, ty+16, ty+24 stay in registers
  float s1 = 0.f;
  if (small && single) {
#pragma unroll
    for (int i = 0; i < DL_ROWS / RP; ++i) {       // no split: the sums never left the registers
      const int r = ty + RP * i;
      const float z = own[i] + b;
      zr[i] = z;
      if (jv && r < R) {
        a.z_out[(long long)r * C + j] = z;
        s1 += z;
      }
    }
  } else if (small) {
    float pv[DL_ROWS / RP][DL_MAX_SPLITS];
#pragma unroll
    for (int i = 0; i < DL_ROWS / RP; ++i) {
      const int r = min(ty + RP * i, R - 1);
#pragma unroll
      for (int q = 0; q < DL_MAX_SPLITS; ++q)          // clamped, unconditional: up to 32 loads in flight
        pv[i][q] = __hip_atomic_load(&a.partial[((long long)min(q, nks - 1) * R + r) * C + jc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int i = 0; i < DL_ROWS / RP; ++i) {
      float z0 = 0.f, z1 = 0.f;
#pragma unroll
      for (int q = 0; q < DL_MAX_SPLITS; q += 2) {
        z0 += (q < nks) ? pv[i][q] : 0.f;
        z1 += (q + 1 < nks) ? pv[i][q + 1] : 0.f;
      }
      const int r = ty + RP * i;
      const float z = (z0 + z1) + b;
      zr[i] = z;
      if (jv && r < R) {
        a.z_out[(long long)r * C + j] = z;
        s1 += z;
      }
    }
  } else if (jv) {
    for (int r = ty; r < R; r += RP) {
      float z0 = 0.f, z1 = 0.f;
      int q = 0;
      for (; q + 1 < nks; q += 2) {
        z0 += __hip_atomic_load(&a.partial[((long long)q * R + r) * C + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        z1 += __hip_atomic_load(&a.partial[((long long)(q + 1) * R + r) * C + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (q < nks) z0 += __hip_atomic_load(&a.partial[((long long)q * R + r) * C + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float z = (z0 + z1) + b;
      a.z_out[(long long)r * C + j] = z;
      s1 += z;
    }
  }
  float sc = 1.f, sh = 0.f;
  if (a.bn_mode) {      // block-uniform
    float mean = 0.f, var = 1.f;
    if (a.bn_mode == 1) {
      __syncthreads();
      fin[ty * 32 + tx] = s1;
      __syncthreads();
      float t = 0.f;
      for (int q = 0; q < RP; ++q) t += fin[q * 32 + tx];
      mean = t / (float)R;
      float s2 = 0.f;
      if (small) {
#pragma unroll
        for (int i = 0; i < DL_ROWS / RP; ++i) {
          const float d = zr[i] - mean;
          s2 = (jv && ty + RP * i < R) ? fmaf(d, d, s2) : s2;
        }
      } else if (jv) {
        for (int r = ty; r < R; r += RP) {
          const float d = a.z_out[(long long)r * C + j] - mean;   // written by this thread above
          s2 = fmaf(d, d, s2);
        }
      }
      fin[(8 + ty) * 32 + tx] = s2;
      __syncthreads();
      t = 0.f;
      for (int q = 0; q < RP; ++q) t += fin[(8 + q) * 32 + tx];
      var = t / (float)R;
      if (ty == 0 && jv) {
        a.mm[j] = pf_mm * a.momentum + mean * (1.f - a.momentum);
        a.mv[j] = pf_mv * a.momentum + var * (1.f - a.momentum);
      }
    } else if (jv) {
      mean = pf_mm;
      var = pf_mv;
    }
    const float invstd = 1.0f / sqrtf(var + a.eps);
    if (jv) {
      sc = pf_g * invstd;
      sh = pf_b - mean * sc;
      if (ty == 0) {
        if (a.mean_o) a.mean_o[j] = mean;
        if (a.invstd_o) a.invstd_o[j] = invstd;
      }
    }
  }
  if (TRANS && a.bt_dz) {      // block-uniform; small (host-checked): zr[] holds this thread's four rows of d(activation)
    float bsc = 1.f, bsh = 0.f;
    const float mu = bt_mu, is = bt_is;
    if (a.bt_mode) {
      bsc = bt_g * is;
      bsh = bt_b - mu * bsc;
    }
    const float kscale = a.bt_keep ? a.bt_keep_scale : 1.f;
    float v[DL_ROWS / RP], zh[DL_ROWS / RP];
    const float (&zz)[DL_ROWS / DL_SLICES] = bt_zz;
    float S1 = 0.f, S2 = 0.f;
#pragma unroll
    for (int i = 0; i < DL_ROWS / RP; ++i) {
      const int r = ty + RP * i;
      float d = bt_kb[i] ? zr[i] * kscale : 0.f;
      if (a.bt_act == 1 && !(fmaf(bsc, zz[i], bsh) > 0.f)) d = 0.f;
      if (r >= R || !jv) d = 0.f;
      zh[i] = (zz[i] - mu) * is;
      v[i] = d;
      S1 += d;
      S2 = fmaf(d, zh[i], S2);
    }
    __syncthreads();
    fin[ty * 32 + tx] = S1;
    fin[(8 + ty) * 32 + tx] = S2;
    __syncthreads();
    S1 = 0.f; S2 = 0.f;
#pragma unroll
    for (int q = 0; q < RP; ++q) { S1 += fin[q * 32 + tx]; S2 += fin[(8 + q) * 32 + tx]; }
    const float invR = 1.f / (float)R;
    const float m1 = S1 * invR, m2 = S2 * invR;
#pragma unroll
    for (int i = 0; i < DL_ROWS / RP; ++i) {
      const int r = ty + RP * i;
      float d = v[i];
      if (a.bt_mode == 1) d = bsc * (d - m1 - zh[i] * m2);
      else if (a.bt_mode == 2) d *= bsc;
      if (jv && r < R) a.bt_dz[(long long)r * C + j] = d;
    }
    if (ty == 0 && jv) {
      if (a.bt_mode == 1) {
        if (a.bt_dgamma) a.bt_dgamma[j] = S2;
        if (a.bt_dbeta) a.bt_dbeta[j] = S1;
      } else if (a.bt_mode == 0 && a.bt_dbias) {
        a.bt_dbias[j] = S1;
      }
    }
  }
  if (a.a_out && jv) {
    if (small) {
#pragma unroll
      for (int i = 0; i < DL_ROWS / RP; ++i) {
        const int r = ty + RP * i;
        if (r < R) {
          float y = fmaf(sc, zr[i], sh);
          if (a.act == 1) y = clamp_lo(y, 0.f);
          if (a.keep) y = pf_kb[i] ? y * a.keep_scale : 0.f;
          a.a_out[(long long)r * C + j] = y;
        }
      }
    } else {
      for (int r = ty; r < R; r += RP) {
        float y = fmaf(sc, a.z_out[(long long)r * C + j], sh);
        if (a.act == 1) y = clamp_lo(y, 0.f);
        if (a.keep) y = a.keep[(long long)r * C + j] ? y * a.keep_scale : 0.f;
        a.a_out[(long long)r * C + j] = y;
      }
    }
  }
}

// Backward of a dense layer's tail fused into its weight gradient (R <= 32), on the matrix cores.
// Grid = (32-column tiles of C) x (groups of 256 k); a wave owns two 32 x 32 tiles of dw = x^T . dz.  Lane (j = lane % 32,
// g = lane / 32) rebuilds dz for column j and the 16 rows {8g..8g+7, 16+8g..16+8g+7} from da / z / the dropout mask -- exactly
// the B operand of its two MFMA k-steps; the two halves of a column meet with one xor-32 shuffle for the BN-backward sums.
// The first k-group's wave 0 also writes dz, dgamma, dbeta / dbias.  Operands are split bf16 hi + lo (3 products, fp32-grade).
// dw may be NULL (frozen layer: only dz is produced).
constexpr int DB_TPW = 2;                 // 32-k tiles per wave
constexpr int DB_KG = 4 * DB_TPW * 32;    // k per block

__global__ __launch_bounds__(256) void dense_bwd_fused_kernel(const float* __restrict__ da, const float* __restrict__ z,
                                                              const float* __restrict__ x, int ldx, int R, int K, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              int bn_mode, int act, const unsigned char* __restrict__ keep,
                                                              float keep_scale, float* __restrict__ dz, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ dbias,
                                                              float* __restrict__ dw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lg = lane >> 5;
  const int j = blockIdx.x * 32 + lr;
  const bool jv = j < C;
  const int jc = jv ? j : C - 1;
  float sc = 1.f, sh = 0.f, mu = 0.f, is = 1.f;
  if (bn_mode) {
    mu = mean[jc]; is = invstd[jc];
    sc = gamma[jc] * is;
    sh = beta[jc] - mu * sc;
  }
  // this lane's 16 rows of column j
  float d[16], zh[16];
  unsigned kmask = 0xffffu;
#pragma unroll
  for (int q = 0; q < 16; ++q) {            // unconditional, clamped loads: all in flight together
    const int r = (q >> 3) * 16 + 8 * lg + (q & 7);
    const long long o = (long long)min(r, R - 1) * C + jc;
    d[q] = da[o];
    zh[q] = z[o];
  }
  if (keep) {
    kmask = 0u;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = (q >> 3) * 16 + 8 * lg + (q & 7);
      kmask |= (keep[(long long)min(r, R - 1) * C + jc] ? 1u : 0u) << q;
    }
  }
  // first x tile of this wave in flight while dz is rebuilt
  const int kbase = blockIdx.y * DB_KG + wave * (DB_TPW * 32);
  float xa[16];
  auto issue_x = [&](int k0) {
    const int kk = min(k0 + lr, K - 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = (q >> 3) * 16 + 8 * lg + (q & 7);
      xa[q] = x[(long long)min(r, R - 1) * ldx + kk];
    }
  };
  if (dw && kbase < K) issue_x(kbase);
  const float kscale = keep ? keep_scale : 1.f;
  float S1 = 0.f, S2 = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = (q >> 3) * 16 + 8 * lg + (q & 7);
    float v = ((kmask >> q) & 1u) ? d[q] * kscale : 0.f;
    if (act == 1 && !(fmaf(sc, zh[q], sh) > 0.f)) v = 0.f;
    if (r >= R || !jv) v = 0.f;
    zh[q] = (zh[q] - mu) * is;
    d[q] = v;
    S1 += v;
    S2 = fmaf(v, zh[q], S2);
  }
  S1 += __shfl_xor(S1, 32, 64);             // the other 16 rows of the column live in lane ^ 32
  S2 += __shfl_xor(S2, 32, 64);
  if (bn_mode == 1) {
    const float invR = 1.f / (float)R;
    const float m1 = S1 * invR, m2 = S2 * invR;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = (q >> 3) * 16 + 8 * lg + (q & 7);
      d[q] = (r < R && jv) ? sc * (d[q] - m1 - zh[q] * m2) : 0.f;
    }
  } else if (bn_mode == 2) {
#pragma unroll
    for (int q = 0; q < 16; ++q) d[q] *= sc;
  }
  if (blockIdx.y == 0 && wave == 0 && jv) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = (q >> 3) * 16 + 8 * lg + (q & 7);
      if (r < R) dz[(long long)r * C + j] = d[q];
    }
    if (lg == 0) {
      if (bn_mode == 1) {
        if (dgamma) dgamma[j] = S2;
        if (dbeta) dbeta[j] = S1;
      } else if (bn_mode == 0 && dbias) {
        dbias[j] = S1;
      }
    }
  }
  if (!dw) return;
  // B operand (dz) of the two k16-steps, bf16 hi + lo
  bf16x8 bh[2], bl[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = d[8 * t + e];
      bh[t][e] = (__bf16)v;
      bl[t][e] = (__bf16)(v - (float)bh[t][e]);
    }
#pragma unroll
  for (int tile = 0; tile < DB_TPW; ++tile) {
    const int k0 = kbase + tile * 32;
    if (k0 >= K) break;                      // wave-uniform
    bf16x8 ah[2], al[2];
    const bool kv = k0 + lr < K;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = 16 * t + 8 * lg + e;
        const float v = (kv && r < R) ? xa[8 * t + e] : 0.f;
        ah[t][e] = (__bf16)v;
        al[t][e] = (__bf16)(v - (float)ah[t][e]);
      }
    if (tile + 1 < DB_TPW && k0 + 32 < K) issue_x(k0 + 32);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], bh[t], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bl[t], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh[t], acc, 0, 0, 0);
    }
    if (jv) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = k0 + (e & 3) + 8 * (e >> 2) + 4 * lg;
        if (k < K) dw[(long long)k * C + j] = acc[e];
      }
    }
  }
}

// Backward through [dropout] -> [relu] -> [BN] of a dense layer:  da (R,C) -> dz (R,C), dgamma, dbeta / dbias.
// block = 32 columns x 8 row partitions.
__global__ __launch_bounds__(256) void dense_bwd_pre_kernel(const float* __restrict__ da, const float* __restrict__ z, int R, int C,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            int bn_mode, int act, const unsigned char* __restrict__ keep,
                                                            float keep_scale, float* __restrict__ dz, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dbias) {
  __shared__ float red[8][2][32];
  __shared__ float bc[2][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + tx;
  const bool jv = j < C;
  float sc = 1.f, sh = 0.f, mu = 0.f, is = 1.f;
  if (bn_mode && jv) {
    mu = mean[j]; is = invstd[j];
    sc = gamma[j] * is;
    sh = beta[j] - mu * sc;
  }
  float S1 = 0.f, S2 = 0.f;
  if (jv)
    for (int r = ty; r < R; r += 8) {
      const long long o = (long long)r * C + j;
      float d = da[o];
      if (keep) d = keep[o] ? d * keep_scale : 0.f;
      const float zz = z[o];
      if (act == 1 && !(fmaf(sc, zz, sh) > 0.f)) d = 0.f;
      dz[o] = d;   // dy_hat for now (re-read by this thread below)
      S1 += d;
      S2 = fmaf(d, (zz - mu) * is, S2);
    }
  red[ty][0][tx] = S1;
  red[ty][1][tx] = S2;
  __syncthreads();
  if (ty == 0) {
    float a = 0.f, b = 0.f;
    for (int q = 0; q < 8; ++q) { a += red[q][0][tx]; b += red[q][1][tx]; }
    bc[0][tx] = a;
    bc[1][tx] = b;
  }
  __syncthreads();
  S1 = bc[0][tx];
  S2 = bc[1][tx];
  if (!jv) return;
  if (bn_mode == 1) {
    if (ty == 0) {
      if (dgamma) dgamma[j] = S2;
      if (dbeta) dbeta[j] = S1;
    }
    const float invR = 1.f / (float)R;
    for (int r = ty; r < R; r += 8) {
      const long long o = (long long)r * C + j;
      const float zh = (z[o] - mu) * is;
      dz[o] = sc * (dz[o] - S1 * invR - zh * S2 * invR);
    }
  } else if (bn_mode == 2) {
    for (int r = ty; r < R; r += 8) dz[(long long)r * C + j] *= sc;
  } else if (ty == 0 && dbias) {
    dbias[j] = S1;
  }
}

// dw[k][j] = sum_r x[r][k] * dz[r][j]     thread <-> column j, block <-> 16 consecutive k; rows in chunks of 32 whose
// dz values are loaded up front (unconditional, clamped) so they are all in flight together
__global__ __launch_bounds__(256) void dense_wgrad_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dz, int R,
                                                          int K, int C, float* __restrict__ dw, float* __restrict__ db) {
  constexpr int KT = 16;
  constexpr int WRC = 32;
  __shared__ float xs[KT][WRC];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int jc = j < C ? j : C - 1;
  const int k0 = blockIdx.y * KT;
  float acc[KT];
  float colsum = 0.f;                      // db[j] = sum_r dz[r][j] (optional, written by the first k tile's blocks)
#pragma unroll
  for (int k = 0; k < KT; ++k) acc[k] = 0.f;
  for (int rc = 0; rc < R; rc += WRC) {
    const int nr = min(WRC, R - rc);
    float d[WRC];
#pragma unroll
    for (int r = 0; r < WRC; ++r) d[r] = dz[(long long)(rc + min(r, nr - 1)) * C + jc];
#pragma unroll
    for (int r = 0; r < WRC; ++r) colsum += (r < nr) ? d[r] : 0.f;
    __syncthreads();
    for (int t = threadIdx.x; t < KT * WRC; t += 256) {
      const int k = t / WRC, r = t % WRC;
      xs[k][r] = (r < nr && k0 + k < K) ? x[(long long)(rc + r) * ldx + k0 + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < WRC; ++r) {
      const float dv = (r < nr) ? d[r] : 0.f;
#pragma unroll
      for (int k = 0; k < KT; ++k) acc[k] = fmaf(xs[k][r], dv, acc[k]);
    }
  }
  if (j < C) {
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k0 + k < K) dw[(long long)(k0 + k) * C + j] = acc[k];
    if (db && blockIdx.y == 0) db[j] = colsum;
  }
}

// out (C, R) = in (R, C)^T, 32x32 LDS tiles
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int R, int C, float* __restrict__ out) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (by + i < R && bx + tx < C) t[i][tx] = in[(long long)(by + i) * C + bx + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (bx + i < C && by + tx < R) out[(long long)(bx + i) * R + by + tx] = t[tx][i];
}

// out (C, R) = in (R, C)^T and out2 (C, R) = rowscale[c] * in^T (second copy scaled per OUTPUT row)
__global__ __launch_bounds__(256) void transpose2_kernel(const float* __restrict__ in, int R, int C, const float* __restrict__ rowscale,
                                                         float* __restrict__ out, float* __restrict__ out2) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (by + i < R && bx + tx < C) t[i][tx] = in[(long long)(by + i) * C + bx + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (bx + i < C && by + tx < R) {
      const float v = t[tx][i];
      out[(long long)(bx + i) * R + by + tx] = v;
      out2[(long long)(bx + i) * R + by + tx] = rowscale[bx + i] * v;
    }
}

// Row softmax + keras SparseCategoricalCrossentropy (clip 1e-7, log, sparse_softmax_xent) + its gradient
// w.r.t. the logits.  Used for the classification head (rows = B).  32 lanes per row (32 rows per pass of the single 1024-thread block);
// every reduction is a fixed xor-shuffle tree, so the result does not depend on anything but the inputs.
//   loss_sum[0] = sum_r nll_r ; correct[0] = #(argmax == label)
__global__ __launch_bounds__(1024) void softmax_xent_rows_kernel(const float* __restrict__ logits, int R, int C,
                                                                const int* __restrict__ labels, float grad_scale,
                                                                float* __restrict__ probs, float* __restrict__ dlogits,
                                                                float* __restrict__ loss_sum, float* __restrict__ correct) {
  __shared__ float rl[32], rc[32];
  softmax_xent_rows_body(logits, R, C, labels, grad_scale, probs, dlogits, loss_sum, correct, rl, rc);
}

// dlogits from an arbitrary upstream d(probs):  dlogit_j = p_j (dp_j - sum_i p_i dp_i)
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ probs, const float* __restrict__ dprobs,
                                                               long long R, int C, float* __restrict__ dlogits) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float* p = probs + r * C;
  const float* dp = dprobs + r * C;
  float dot = 0.f;
  for (int c = 0; c < C; ++c) dot = fmaf(p[c], dp[c], dot);
  float* d = dlogits + r * C;
  for (int c = 0; c < C; ++c) d[c] = p[c] * (dp[c] - dot);
}

// class / part index of every row: the first maximum (np.argmax / tf.math.argmax order).  Rows are 48..92 bytes, so a
// thread per row reads whole cache lines between neighbours; the launch is bandwidth-trivial (3 MB at B=32, N=2048).
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ v, long long R, int C, int* __restrict__ out) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float* p = v + r * C;
  float best = p[0];
  int bi = 0;
  for (int c = 1; c < C; ++c) {
    const float x = p[c];
    if (x > best) { best = x; bi = c; }
  }
  out[r] = bi;
}

// ---- host wrappers ---------------------------------------------------------------------------------------
size_t dense_partial_floats(int R, int K, int C) { return (size_t)dl_nsplit(K) * R * C; }

static int dense_args(DenseArgs& a, const float* x, int ldx, const float* w, int ldw, int R, int K, int C, float* partial, unsigned* counters,
                      const float* bias, const float* gamma, const float* beta, float* mm, float* mv, float momentum, float eps, int bn_mode,
                      int act, const unsigned char* keep, float keep_scale, float* z_out, float* a_out, float* mean_o, float* invstd_o) {
  PN_CHECK_ARG(x && w && partial && counters && z_out && R > 0 && K > 0 && C > 0, "dense_layer: bad arguments");
  PN_CHECK_ARG(cdiv(C, DL_COLS) <= DENSE_MAX_COUNTERS, "dense_layer: C=%d needs more than %d counters", C, DENSE_MAX_COUNTERS);
  PN_CHECK_ARG(!bn_mode || (gamma && beta && mm && mv), "dense_layer: BatchNormalization needs gamma/beta/moving statistics");
  a.x = x; a.ldx = ldx; a.w = w; a.ldw = ldw; a.R = R; a.K = K; a.C = C;
  a.nsplit = dl_nsplit(K); a.split_len = dl_split_len(K);
  // one column block and a short contraction (the logits layer, the 3 x 3 transform's output): one workgroup walks all of K -- two
  // more 64-k steps per wave cost less than the in-launch meeting of two splits (partial tiles, ticket, re-read)
  if (cdiv(C, DL_COLS) == 1 && K <= 512) { a.nsplit = 1; a.split_len = cdiv(K, DL_KSTEP) * DL_KSTEP; }
  a.partial = partial; a.counters = counters;
  a.bias = bias; a.gamma = gamma; a.beta = beta; a.mm = mm; a.mv = mv; a.momentum = momentum; a.eps = eps;
  a.bn_mode = bn_mode; a.act = act; a.keep = keep; a.keep_scale = keep_scale;
  a.z_out = z_out; a.a_out = a_out; a.mean_o = mean_o; a.invstd_o = invstd_o;
  a.bt_z = a.bt_gamma = a.bt_beta = a.bt_mean = a.bt_invstd = nullptr;
  a.bt_keep = nullptr; a.bt_keep_scale = 1.f; a.bt_mode = 0; a.bt_act = 0;
  a.bt_dz = a.bt_dgamma = a.bt_dbeta = a.bt_dbias = nullptr;
  return PN_OK;
}
// the kernel variant of a launch: DEPTH from the 64-k steps a wave walks (K per split / 4 waves / 64), VEC from the operands' alignment
template <bool TRANS>
static void launch_dense(const DenseArgs& a, const DenseArgs& b, dim3 grid, hipStream_t st) {
  const int steps = cdiv(a.split_len, 64);                  // 64-k steps per wave (wave w walks the k16-steps w, w + 4, ...)
  const int depth = steps <= 2 ? 2 : (steps <= 4 ? 4 : 8);
  auto aligned = [](const DenseArgs& q) {
    return q.K % 16 == 0 && q.ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(q.x) & 15) == 0 &&
           (!TRANS || (q.ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(q.w) & 15) == 0));
  };
  const bool vec = aligned(a) && aligned(b);
  // (unaligned operands -- odd shapes of the op-level tests only -- take the scalar-load form, which keeps one step ahead whatever K)
#define PN_DENSE_CASE(D, V) hipLaunchKernelGGL((dense_layer_kernel<TRANS, D, V>), grid, dim3(256), 0, st, a, b)
  if (vec) { if (depth == 2) PN_DENSE_CASE(2, true); else if (depth == 4) PN_DENSE_CASE(4, true); else PN_DENSE_CASE(8, true); }
  else PN_DENSE_CASE(2, false);
#undef PN_DENSE_CASE
}

int dense_layer(const float* x, int ldx, const float* w, int ldw, bool trans, int R, int K, int C, float* partial, unsigned* counters,
                const float* bias, const float* gamma, const float* beta, float* mm, float* mv, float momentum, float eps, int bn_mode,
                int act, const unsigned char* keep, float keep_scale, float* z_out, float* a_out, float* mean_o, float* invstd_o,
                hipStream_t st) {
  DenseArgs a;
  PN_TRY(dense_args(a, x, ldx, w, ldw, R, K, C, partial, counters, bias, gamma, beta, mm, mv, momentum, eps, bn_mode, act, keep, keep_scale,
                    z_out, a_out, mean_o, invstd_o));
  const dim3 grid(cdiv(C, DL_COLS), a.nsplit);
  if (trans) launch_dense<true>(a, a, grid, st);
  else launch_dense<false>(a, a, grid, st);
  PN_CHECK_LAUNCH();
  return PN_OK;
}
// out2 (R, C) = x . w2 (plain product, no bias) in the launch of a dense layer of the same K and C on the same input: the second job
// uses the upper halves of `partial` (2 * dense_partial_floats) and of the counters
int dense_layer_with_plain(const float* x, int ldx, const float* w, int ldw, int R, int K, int C, float* partial, unsigned* counters,
                           const float* bias, const float* gamma, const float* beta, float* mm, float* mv, float momentum, float eps,
                           int bn_mode, int act, const unsigned char* keep, float keep_scale, float* z_out, float* a_out, float* mean_o,
                           float* invstd_o, const float* w2, int ldw2, float* out2, hipStream_t st) {
  PN_CHECK_ARG(2 * cdiv(C, DL_COLS) <= DENSE_MAX_COUNTERS, "dense_layer_with_plain: C=%d needs more than %d counters", C, DENSE_MAX_COUNTERS);
  DenseArgs a, b;
  PN_TRY(dense_args(a, x, ldx, w, ldw, R, K, C, partial, counters, bias, gamma, beta, mm, mv, momentum, eps, bn_mode, act, keep, keep_scale,
                    z_out, a_out, mean_o, invstd_o));
  PN_TRY(dense_args(b, x, ldx, w2, ldw2, R, K, C, partial + dense_partial_floats(R, K, C), counters + cdiv(C, DL_COLS), nullptr, nullptr, nullptr,
                    nullptr, nullptr, 0.f, 0.f, 0, 0, nullptr, 1.f, out2, nullptr, nullptr, nullptr));
  launch_dense<false>(a, b, dim3(cdiv(C, DL_COLS), a.nsplit, 2), st);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// dx (R, C) = dz (R, K) . W^T from the (C, K)-shaped kernel W (ldw = K), and -- tail != NULL -- the backward of the layer below
// through its dropout / ReLU / BatchNormalization in the same launch (DenseArgs::bt_*)
int dense_trans_tail(const float* dz, int lddz, const float* w, int ldw, int R, int K, int C, float* partial, unsigned* counters, float* dx,
                     const DenseTail* tail, hipStream_t st) {
  DenseArgs a;
  PN_TRY(dense_args(a, dz, lddz, w, ldw, R, K, C, partial, counters, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0, 0, nullptr, 1.f,
                    dx, nullptr, nullptr, nullptr));
  if (tail) {
    PN_CHECK_ARG(R <= DL_ROWS, "dense_trans_tail: the backward tail needs R <= %d (R=%d)", DL_ROWS, R);
    PN_CHECK_ARG(tail->z && tail->dz, "dense_trans_tail: null pointer in the tail");
    PN_CHECK_ARG(!tail->mode || (tail->gamma && tail->beta && tail->mean && tail->invstd), "dense_trans_tail: BatchNormalization needs gamma/beta/mean/invstd");
    a.bt_z = tail->z; a.bt_gamma = tail->gamma; a.bt_beta = tail->beta; a.bt_mean = tail->mean; a.bt_invstd = tail->invstd;
    a.bt_keep = tail->keep; a.bt_keep_scale = tail->keep_scale; a.bt_mode = tail->mode; a.bt_act = tail->act;
    a.bt_dz = tail->dz; a.bt_dgamma = tail->dgamma; a.bt_dbeta = tail->dbeta; a.bt_dbias = tail->dbias;
  }
  launch_dense<true>(a, a, dim3(cdiv(C, DL_COLS), a.nsplit), st);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

__global__ __launch_bounds__(256) void dense_wgrad_batch_kernel(const DenseWgradBatch b) {
  __shared__ float xs[16][32];
  dense_wgrad_batch_body(b, (int)blockIdx.x, xs);
}
int dense_wgrad_batch(const DenseWgradJob* jobs, int n, hipStream_t st) {
  DenseWgradBatch b;
  int blocks = 0;
  PN_TRY(make_dense_wgrad_batch(jobs, n, b, blocks));
  hipLaunchKernelGGL(dense_wgrad_batch_kernel, dim3(blocks), dim3(256), 0, st, b);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int dense_bwd_fused(const float* da, const float* z, const float* x, int ldx, int R, int K, int C, const float* gamma, const float* beta,
                    const float* mean, const float* invstd, int bn_mode, int act, const unsigned char* keep, float keep_scale, float* dz,
                    float* dgamma, float* dbeta, float* dbias, float* dw, hipStream_t st) {
  PN_CHECK_ARG(da && z && dz && R > 0 && R <= 32 && C > 0, "dense_bwd_fused: bad arguments (R must be <= 32)");
  PN_CHECK_ARG(!dw || (x && K > 0), "dense_bwd_fused: the weight gradient needs the layer input");
  PN_CHECK_ARG(!bn_mode || (gamma && beta && mean && invstd), "dense_bwd_fused: BatchNormalization needs gamma/beta/mean/invstd");
  hipLaunchKernelGGL(dense_bwd_fused_kernel, dim3(cdiv(C, 32), dw ? cdiv(K, DB_KG) : 1), dim3(256), 0, st, da, z, x, ldx, R, K, C, gamma,
                     beta, mean, invstd, bn_mode, act, keep, keep_scale, dz, dgamma, dbeta, dbias, dw);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int dense_bwd_pre(const float* da, const float* z, int R, int C, const float* gamma, const float* beta, const float* mean,
                  const float* invstd, int bn_mode, int act, const unsigned char* keep, float keep_scale, float* dz, float* dgamma,
                  float* dbeta, float* dbias, hipStream_t st) {
  PN_CHECK_ARG(da && z && dz, "dense_bwd_pre: null pointer");
  hipLaunchKernelGGL(dense_bwd_pre_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, da, z, R, C, gamma, beta, mean, invstd, bn_mode,
                     act, keep, keep_scale, dz, dgamma, dbeta, dbias);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int dense_wgrad(const float* x, int ldx, const float* dz, int R, int K, int C, float* dw, hipStream_t st, float* db) {
  PN_CHECK_ARG(x && dz && dw, "dense_wgrad: null pointer");
  hipLaunchKernelGGL(dense_wgrad_kernel, dim3(cdiv(C, 256), cdiv(K, 16)), dim3(256), 0, st, x, ldx, dz, R, K, C, dw, db);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int transpose(const float* in, int R, int C, float* out, hipStream_t st) {
  PN_CHECK_ARG(in && out && R > 0 && C > 0, "transpose: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, st, in, R, C, out);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int transpose2(const float* in, int R, int C, const float* rowscale, float* out, float* out2, hipStream_t st) {
  PN_CHECK_ARG(in && rowscale && out && out2 && R > 0 && C > 0, "transpose2: bad arguments");
  hipLaunchKernelGGL(transpose2_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, st, in, R, C, rowscale, out, out2);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int softmax_xent_rows(const float* logits, int R, int C, const int* labels, float grad_scale, float* probs, float* dlogits,
                      float* loss_sum, float* correct, hipStream_t st) {
  PN_CHECK_ARG(logits && probs && R > 0 && C > 0, "softmax_xent_rows: bad arguments");
  hipLaunchKernelGGL(softmax_xent_rows_kernel, dim3(1), dim3(1024), 0, st, logits, R, C, labels, grad_scale, probs, dlogits,
                     loss_sum, correct);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int softmax_bwd_rows(const float* probs, const float* dprobs, long long R, int C, float* dlogits, hipStream_t st) {
  PN_CHECK_ARG(probs && dprobs && dlogits, "softmax_bwd_rows: null pointer");
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)cdivll(R, 256)), dim3(256), 0, st, probs, dprobs, R, C, dlogits);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int argmax_rows(const float* values, long long R, int C, int* index, hipStream_t st) {
  PN_CHECK_ARG(values && index && R > 0 && C > 0, "argmax_rows: bad arguments (R=%lld C=%d)", R, C);
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)cdivll(R, 256)), dim3(256), 0, st, values, R, C, index);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

}  // namespace pn
