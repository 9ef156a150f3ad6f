// Row-panel kernel for the dominant layer: ConvLayer(K -> C) + BatchNormalization statistics + tf.reduce_max over the points
// (pointnet/PointNet.py:242-248, 425-429) on bf16 MFMA, its weight preparation and its finaliser.
//
// A workgroup (8 waves) owns a RUN of 64-row panels of one cloud for ALL C channels (kernel-stationary, see panel_max_kernel):
//   * every wave owns whole COLUMNS: wave w keeps the MFMA B fragments of its column blocks (32 channels each) in registers for the
//     whole launch, read once from a FRAGMENT-ORDERED bf16 copy of the kernel (pn_weights_prep: the 16 bytes lane l needs for k-step
//     ks of column block cb sit at ((cb * K/16 + ks) * 64 + l) * 16: one coalesced 1 KB transaction straight into the MFMA register
//     layout); a column's maximum over the run never leaves the wave (no cross-wave reduction, no LDS traffic for the results);
//   * the activation panels (64 rows x K, the previous layer's BN + ReLU applied on load, rounded once to bf16 hi [+ lo]) stream through
//     a double-buffered LDS image and are the only thing the waves share: one barrier per panel;
//   * the epilogue holds only what cannot be had elsewhere: max over the rows (v_max3: half an instruction per accumulator
//     element) and sum of squares (one fma per element).  The column SUM comes from the run's column sums of A (a1 = A^T 1):
//     sum_m z[m][c] = a1 . W[:, c], formed once per slot.  The ROW of the maximum is not tracked element by element either (that
//     costs two to three instructions per element): the kernel records which 32-row block held the maximum and the backward pass,
//     the only consumer of the row, finds it among those 32 candidates (pn_maxbwd.hip: max_resolve).
// Per point: 2 * K * C FLOP against 4 * K bytes of input: MFMA-bound by the roofline.
#include <cstdlib>
#include "pn_common.h"
#include "pn_internal.h"
#include "pn_gemm_core.h"

namespace pn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // native vector: stays in registers (HIP's uint4 struct does not)

// Maximum of the 16 accumulator values a lane holds, as v_max3_f32 written by hand: through fmaxf the compiler first quiets every
// input (a v_max_f32 x, x per value -- 128 more vector instructions per panel and wave); the instruction itself already returns the
// other operand for a NaN, which is all fmaxf promises.
// HAZARD: nothing pads inline assembly.  An MFMA's result may be read by a vector instruction only 12 wait states after the MFMA
// issued (8-pass XDL), and the compiler's hazard recognizer supplies them only for instructions it knows.  Two safe forms:
//   max16_guarded  -- the first two values go through fmaxf (compiler-known reads: they carry the wait states), the rest of the tree
//                     is ONE statement that depends on that value, so it cannot be scheduled ahead of it;
//   max16_a/_b     -- two bare halves for the software-pipelined loop, which places them (between scheduling fences) behind
//                     compiler-known reads of the same accumulator and several MFMAs of the next chain.
__device__ __forceinline__ float max16_guarded(const f32x16& a) {
  const float m01 = fmaxf(a[0], a[1]);
  float r, t;
  asm("v_max3_f32 %0, %2, %3, %4\n\t"
      "v_max3_f32 %1, %5, %6, %7\n\t"
      "v_max3_f32 %0, %0, %1, %8\n\t"
      "v_max3_f32 %1, %9, %10, %11\n\t"
      "v_max3_f32 %0, %0, %1, %12\n\t"
      "v_max3_f32 %1, %13, %14, %15\n\t"
      "v_max3_f32 %0, %0, %1, %16"
      : "=&v"(r), "=&v"(t)
      : "v"(m01), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]),
        "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]));
  return r;
}
__device__ __forceinline__ void max16_a(const f32x16& a, float& r, float& t) {      // values 0..9
  asm volatile("v_max3_f32 %0, %2, %3, %4\n\t"
               "v_max3_f32 %1, %5, %6, %7\n\t"
               "v_max3_f32 %0, %0, %1, %8\n\t"
               "v_max3_f32 %1, %9, %10, %11"
               : "=&v"(r), "=&v"(t)
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]));
}
__device__ __forceinline__ void max16_b(const f32x16& a, float& r, float& t) {      // values 10..15 joined to (r, t)
  asm volatile("v_max3_f32 %0, %0, %1, %2\n\t"
               "v_max3_f32 %1, %3, %4, %5\n\t"
               "v_max3_f32 %0, %0, %1, %6\n\t"
               "v_max_f32 %0, %0, %7"
               : "+v"(r), "+v"(t)
               : "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]));
}

// ---- the panel kernel --------------------------------------------------------------------------------------------------------
// Work split: every cloud is cut into `spc` (slots per cloud, panel_slots_per_cloud) contiguous runs of 64-row panels, one workgroup
// per run, about one workgroup per CU in all.  A workgroup = 8 waves (two per SIMD, up to 256 VGPRs each).
struct PanelArgs {
  pn_operand a;                 // lazy activation operand over (B*N, K)
  const __bf16* wf_hi;          // fragment-ordered copies (weights_prep)
  const __bf16* wf_lo;          // (NS == 3)
  int B, N, C;
  int spc;                      // slots (workgroups) per cloud
  float* pmax;                  // [B * spc][C]  max over the slot's rows of the accumulator (= sgn * z with presigned weights)
  int* pq;                      // [B * spc][C]  index inside the cloud of the 32-row block that held it (lowest on ties)
  float* sumsq;                 // [B * spc][C]  sum over the slot's rows of z^2, or NULL
  long long* colacc;            // [B][NT * K]  per cloud: column sums of the staged operand rows (hi image, then lo) in 2^-24 fixed point,
                                //   ADDED to what is there (zero on entry), or NULL (comes with sumsq)
};
// PN_PANEL_DBG (timing ablations of the bf16, K = 128, statistics variant; WRONG results; tools/panel_probe.py): template bit mask DBG:
// 1 no epilogue, 4 no activation loads, 8 no MFMAs; 16 (bf16 source only) the product kernel + shader-clock stamps in pq

// PN_PANEL_SPLIT (experiment, bf16 operands, C >= 1024; read once): 1 = a workgroup owns HALF the columns (two column blocks per wave: 64
// fragment registers instead of 128, half the kernel bytes per workgroup) and twice the rows (half the slots per cloud, same number of
// workgroups); 2 = half the columns, the same rows (twice the workgroups)
static int panel_split_mode() {
  static const int m = getenv("PN_PANEL_SPLIT") ? atoi(getenv("PN_PANEL_SPLIT")) : 0;
  return m;
}
int panel_slots_per_cloud(int B, int N) {
  const int tpc = cdiv(N, 64);
  int spc = 256 / (B > 0 ? B : 1);
  if (panel_split_mode() == 1) spc /= 2;
  if (spc < 1) spc = 1;
  return spc < tpc ? spc : tpc;
}

// KERNEL-STATIONARY: the bf16 kernel of the layer (K x C = 256 KB at 128 x 1024) is exactly what the eight waves of a workgroup can
// hold in registers -- wave w keeps the B fragments of its CBW column blocks (CBW * K/16 * 4 VGPRs = 128 at CBW = 4, K = 128) for the
// whole launch -- so the kernel is read from L2 once per workgroup and the main loop streams only activations:
//   * 64-row panels of the workgroup's run go through a double-buffered bf16 LDS image (BN + ReLU applied on load, rounded once);
//     the global loads of panel p+1 are issued before the MFMAs of panel p and converted after them: one barrier per panel;
//   * per panel and owned column block: K/16 x 2 MFMAs 32x32x16 with A fragments from LDS, then the epilogue on the 2 x 16 accumulator
//     values per lane: v_max3 chains (half an instruction per element) into a running (max, 32-row block) and packed fma into a
//     running sum of squares -- both kept in registers across the run's panels and written once per (slot, channel);
//   * bf16x3 operands need hi + lo fragments (twice the registers): a workgroup then owns half the columns (CBW = 2 per wave) and the
//     grid's second dimension walks the column halves.
template <int NS, int K, bool STATS, int CBW, int DBG = 0, bool H16 = false>
__global__ __launch_bounds__(512) void panel_max_kernel(const PanelArgs g) {
  constexpr int BM = 64, THREADS = 512, KS = K / 16;
  constexpr int PA = K + 8;                          // LDS row pitch (bf16 elements): rows r .. r+15 on 16 distinct 16-byte slots
  constexpr int NT = (NS == 3) ? 2 : 1;
  constexpr int CH = K / 8;                          // 16-byte bf16 chunks per row
  constexpr int RP = THREADS / CH;                   // rows per staging pass
  constexpr int P = BM / RP;                         // passes (2 at K = 128, 1 at K = 64)
  constexpr int NBLK = CBW * NT;                     // B blocks (column block x image) of KS fragments each owned by a wave
  constexpr int RBMAX = (NS == 3 || (STATS && (DBG & ~16) != 0)) ? 3 : 4;
  constexpr int RB = NBLK < RBMAX ? NBLK : RBMAX;    // ... of which this many live in registers (RB * KS * 4 VGPRs) and the rest in LDS:
  constexpr int LB = NBLK - RB;                      // 4 x 32 VGPRs + accumulators + staging do not fit 256 registers without spills
  __shared__ __attribute__((aligned(16))) __bf16 Ap[2][NT][BM * PA];
  __shared__ float red[8][NT * K];
  __shared__ __attribute__((aligned(16))) float coef[2 * K];   // BN + ReLU coefficients of the operand's K columns: ca, then cc
  __shared__ u32x4 Bl[LB > 0 ? 8 : 1][LB > 0 ? LB * KS : 1][LB > 0 ? 64 : 1];   // [wave][block, k-step][lane]: lane-linear, conflict-free

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int slot = blockIdx.x, cg = blockIdx.y;
  // DBG & 16: shader-clock stamps of wave 0 at the kernel's phase boundaries, left in the slot's pq row (tools/panel_probe.py)
  __shared__ unsigned stamps[(DBG & 16) ? 2 : 1][(DBG & 16) ? 64 : 1];      // [wave 0, wave 4][stamp]
  int n_stamp = 0;
  auto stamp = [&]() {
    if constexpr ((DBG & 16) != 0) {
      if ((tid == 0 || tid == 256) && n_stamp < 64) stamps[tid >> 8][n_stamp] = (unsigned)__builtin_amdgcn_s_memtime();
      ++n_stamp;
    }
  };
  stamp();                                           // 0: entry
  const int cloud = slot / g.spc, j = slot - cloud * g.spc;
  const int tpc = (g.N + BM - 1) / BM;
  const int p_begin = (int)((long long)j * tpc / g.spc), p_end = (int)((long long)(j + 1) * tpc / g.spc);
  const long long cloud_row0 = (long long)cloud * g.N;

  // ---- staging: thread <-> (row rin + 32 p, 8 consecutive k); BN + ReLU coefficients of its 8 columns --------------------------
  const int ch = tid % CH, rin = tid / CH;
  const int k = ch * 8;
  const float lo = g.a.lo;
  float a1h[8], a1l[NS == 3 ? 8 : 1];                // this thread's share of the column sums of the staged images (its 8 columns)
#pragma unroll
  for (int e = 0; e < 8; ++e) { a1h[e] = 0.f; if (NS == 3) a1l[e] = 0.f; }
  float4 x[P][H16 ? 1 : 2];                          // H16: the activations are stored as bf16 -- eight of them are one 16-byte load
  auto issue = [&](int panel) {
    const int rbase = panel * BM, nrows = min(BM, g.N - rbase);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int rr = p * RP + rin;
      const long long rsrc = cloud_row0 + rbase + (rr < nrows ? rr : nrows - 1);
      if constexpr (H16) {
        x[p][0] = __builtin_bit_cast(float4, act_load8_raw(g.a.s1, rsrc * g.a.ld + k));
      } else {
        const float* s = g.a.s1 + rsrc * g.a.ld + k;
        if (!(DBG & 4)) {
          x[p][0] = *reinterpret_cast<const float4*>(s);
          x[p][1] = *reinterpret_cast<const float4*>(s + 4);
        } else {
          x[p][0] = x[p][1] = make_float4(1.f, 2.f, 3.f, 4.f);
        }
      }
    }
  };
  auto convert = [&](int panel, int buf) {
    const int nrows = min(BM, g.N - panel * BM);
    // the BN + ReLU coefficients of this thread's 8 columns are re-read here rather than kept in 16 registers across the MFMA
    // sections: the kernel's fragments already take 128 of the 256
    asm volatile("" ::: "memory");
    // ... and the staged values are made opaque HERE: their unpacking is plain arithmetic, which the compiler otherwise hoists to
    // the top of the panel (in front of a wait for loads that were issued a moment earlier)
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int q = 0; q < (H16 ? 1 : 2); ++q) {
        u32x4 t = __builtin_bit_cast(u32x4, x[p][q]);
        asm volatile("" : "+v"(t));
        x[p][q] = __builtin_bit_cast(float4, t);
      }
    float ca[8], cc[8];
    {   // from the LDS table (four reads in flight together: one short wait; the global form cost two L1 round trips per panel)
      const float4 t0 = *reinterpret_cast<const float4*>(&coef[k]), t1 = *reinterpret_cast<const float4*>(&coef[k + 4]);
      const float4 u0 = *reinterpret_cast<const float4*>(&coef[K + k]), u1 = *reinterpret_cast<const float4*>(&coef[K + k + 4]);
      ca[0] = t0.x; ca[1] = t0.y; ca[2] = t0.z; ca[3] = t0.w; ca[4] = t1.x; ca[5] = t1.y; ca[6] = t1.z; ca[7] = t1.w;
      cc[0] = u0.x; cc[1] = u0.y; cc[2] = u0.z; cc[3] = u0.w; cc[4] = u1.x; cc[5] = u1.y; cc[6] = u1.z; cc[7] = u1.w;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int rr = p * RP + rin;
      const bool rv = rr < nrows;
      float v[8];
      if constexpr (H16) {
        bf16x8_unpack(__builtin_bit_cast(uint4, x[p][0]), v);
      } else {
        v[0] = x[p][0].x; v[1] = x[p][0].y; v[2] = x[p][0].z; v[3] = x[p][0].w;
        v[4] = x[p][H16 ? 0 : 1].x; v[5] = x[p][H16 ? 0 : 1].y; v[6] = x[p][H16 ? 0 : 1].z; v[7] = x[p][H16 ? 0 : 1].w;
      }
      if constexpr (NS == 1) {
        // one image: pairs of values are rounded by ONE v_cvt_pk_bf16_f32 straight into the dword the LDS store takes; the row mask
        // and the column sums work on that dword (its halves ARE the rounded values)
        u32x4 pk;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const f32x2 tt = {clamp_lo(fmaf(ca[e], v[e], cc[e]), lo), clamp_lo(fmaf(ca[e + 1], v[e + 1], cc[e + 1]), lo)};
          const unsigned w = rv ? __builtin_bit_cast(unsigned, __builtin_convertvector(tt, bf16x2)) : 0u;   // rows outside the cloud are zero rows
          pk[e / 2] = w;
          if (STATS) {
            a1h[e] += __builtin_bit_cast(float, w << 16);
            a1h[e + 1] += __builtin_bit_cast(float, w & 0xffff0000u);
          }
        }
        *reinterpret_cast<u32x4*>(&Ap[buf][0][rr * PA + k]) = pk;
        continue;
      }
      bf16x8 hv, lv;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = rv ? clamp_lo(fmaf(ca[e], v[e], cc[e]), lo) : 0.f;    // rows outside the cloud are zero rows
        hv[e] = (__bf16)t;
        if (NS == 3) lv[e] = (__bf16)(t - (float)hv[e]);
      }
      *reinterpret_cast<bf16x8*>(&Ap[buf][0][rr * PA + k]) = hv;
      if (NS == 3) *reinterpret_cast<bf16x8*>(&Ap[buf][NT - 1][rr * PA + k]) = lv;
      if (STATS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          a1h[e] += (float)hv[e];
          if (NS == 3) a1l[e] += (float)lv[e];
        }
      }
    }
  };
  // ---- this wave's columns of the kernel: loaded once, resident for the whole run ----------------------------------------------
  // Order of the prologue's requests = order of their use (loads return in issue order, and a wait for one drains everything
  // requested before it): the coefficient table, the first panel's rows, THEN the 256 KB of kernel fragments -- the first panel is
  // converted while the fragments are still arriving, and its chains start as their own fragments land (a CU takes in 64 bytes a
  // cycle: the fragments alone are ~2 us)
  float cav = 1.f, ccv = 0.f;
  if (tid < K) {
    if (g.a.ca) cav = g.a.ca[tid];
    if (g.a.cc) ccv = g.a.cc[tid];
  }
  issue(p_begin);
  asm volatile("" ::: "memory");
  const u32x4* __restrict__ wfh = reinterpret_cast<const u32x4*>(g.wf_hi);
  const u32x4* __restrict__ wfl = reinterpret_cast<const u32x4*>(g.wf_lo);
  u32x4 bw[RB][KS];                                  // block q = i * NT + image (0 hi, 1 lo) of owned column block i
  int cbs[CBW];
#pragma unroll
  for (int i = 0; i < CBW; ++i) cbs[i] = (cg * CBW + i) * 8 + wave;
  // peeled first panel (do_panel): bf16 mode, four register-resident blocks, a full first panel
  constexpr bool CAN_PEEL = NS == 1 && (DBG & ~16) == 0 && CBW == 4 && RB == 4;
  const bool peel = CAN_PEEL && g.N - p_begin * BM >= BM && p_begin < p_end;
#pragma unroll
  for (int q = 0; q < NBLK; ++q) {
    const u32x4* __restrict__ src = ((q % NT) ? wfl : wfh) + (long long)cbs[q / NT] * KS * 64 + lane;
    if (CAN_PEEL && q >= 2 && peel) continue;        // block-uniform
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (q < RB) bw[q][ks] = src[ks * 64];
      else Bl[wave][(q - RB) * KS + ks][lane] = src[ks * 64];      // read back by this wave only
    }
  }

  asm volatile("" ::: "memory");
  if (tid < K) {
    coef[tid] = cav;
    coef[K + tid] = ccv;
  }
  stamp();                                           // 1: every request of the prologue issued
  __syncthreads();                                   // the coefficient table
  stamp();                                           // 2
  convert(p_begin, 0);
  __syncthreads();
  stamp();                                           // 3: first panel staged

  // ---- the run's panels ------------------------------------------------------------------------------------------------------
  float best[CBW], ss[CBW];
  int bq[CBW];
#pragma unroll
  for (int i = 0; i < CBW; ++i) { best[i] = -INFINITY; ss[i] = 0.f; bq[i] = 0; }
  // The body of one panel.  FIRST (compile-time) = the run's first panel in the PEELED form: the prologue requested only the kernel
  // fragments of column blocks 0 and 1; blocks 2 and 3 are requested from the gaps of chains 0 and 1, so the 256 KB of fragments
  // (a CU takes in 64 bytes a cycle: ~4,400 cycles) arrive while the first panel is already being multiplied.
  auto do_panel = [&](const int pnl, auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const int buf = (pnl - p_begin) & 1;
    const int nrows = min(BM, g.N - pnl * BM);
    const bool full = nrows == BM;                   // block-uniform
    // flies under this panel's MFMAs.  Unconditional (the run's last panel requests its own rows again and drops them): behind a
    // branch the staging registers are loop-carried values, and the compiler waits for everything in flight at the loop head
    issue(pnl + 1 < p_end ? pnl + 1 : pnl);
    if constexpr (NS == 1 && (DBG & ~16) == 0) {
      if (full) {
        // ---- software-pipelined form (full panels, one bf16 image) ---------------------------------------------------------------
        // The panel's 2 x CBW (row block, column block) accumulators are walked as ONE chain of chains: while chain t's MFMAs issue,
        // the vector instructions of chain t-1's epilogue sit in the gaps between them (an MFMA holds the SIMD's vector issue for 8 of
        // its 32 cycles; a wave issues in order, so work that is to hide under an MFMA must stand between two MFMAs in program order).
        // Scheduling fences pin that interleave.  The epilogue's first unit is compiler-known arithmetic on the accumulator (it
        // carries the MFMA -> VALU wait states), the hand-written max tree comes behind it.  Sums and maxima are taken in the same
        // association as the plain form below: bit-identical results.
        constexpr int NB = 2 * CBW;
        constexpr int T_A = NB >= 4 ? NB / 2 : NB - 1, T_B = (3 * NB) / 4;     // where the two wave groups convert the next panel
        f32x16 acc2[2];
        bf16x8 af[KS];
        u32x4 bl_next;                                 // fragment of the LDS-resident column block, read one k-step ahead
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(&Ap[buf][0][r * PA + ks * 16 + h * 8]);
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, mr = 0.f, mt = 0.f;
        // unit u (0..6) of the epilogue of block tb, on accumulator a: four sum-of-squares groups, two halves of the max tree, the join
        // (arithmetic has no place of its own in the compiler's eyes -- it floats to wherever its operands allow -- so every unit is
        // fenced by empty volatile statements on what it reads and what it leaves)
        auto unit = [&](int u, f32x16& a, int ib, int blk32) {
          if (u < 4) {
            if (STATS) {
              const int e = 4 * u;
              if (u == 0) asm volatile("" : "+v"(a));  // not before this gap
              q0 = fmaf(a[e], a[e], u ? q0 : 0.f);
              q1 = fmaf(a[e + 1], a[e + 1], u ? q1 : 0.f);
              q2 = fmaf(a[e + 2], a[e + 2], u ? q2 : 0.f);
              q3 = fmaf(a[e + 3], a[e + 3], u ? q3 : 0.f);
              asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));   // not after it
            } else if (u == 0) {
              asm volatile("" : "+v"(a));
              mr = fmaxf(a[0], a[1]);                  // compiler-known read of the accumulator: carries the wait states
              asm volatile("" : "+v"(mr));
            }
          } else if (u == 4) {
            max16_a(a, mr, mt);
          } else if (u == 5) {
            max16_b(a, mr, mt);
          } else {
            if (STATS) ss[ib] += (q0 + q1) + (q2 + q3);
            const bool better = mr > best[ib];         // panels and blocks ascend: the first maximum wins
            best[ib] = better ? mr : best[ib];
            bq[ib] = better ? blk32 : bq[ib];
            if (STATS) asm volatile("" : "+v"(ss[ib]));
            asm volatile("" : "+v"(best[ib]), "+v"(bq[ib]));
          }
        };
        auto gap_of = [](int u) { return 1 + (u * (KS - 1)) / 7; };
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          const int i = t % CBW;
          f32x16& acc = acc2[t & 1];
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[e] = 0.f;
          const bool from_lds = i * NT >= RB;          // this chain's B fragments come from LDS, one k-step ahead of their MFMA
          // The next panel's conversion is vector-ALU work only.  Waves w and w + 4 share a SIMD: the first four waves convert in the
          // middle of the panel, the other four three quarters through -- while one wave of a SIMD converts, its partner has the
          // matrix core to itself (in lockstep both would leave it idle for two conversions per panel).
          // (Tried and dropped: the panel's barrier in front of its last chain, whose gaps then request the next panel's first A
          // fragments -- the fragments become live across the panel boundary, 32 registers the kernel does not have: spills in the loop.)
          if (t == T_A) {
            stamp();                                   // (panel) chains 0 .. T_A - 1 issued
            if ((NB < 4 || wave < 4) && pnl + 1 < p_end) convert(pnl + 1, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            stamp();                                   // first group's conversion done
          }
          if (NB >= 4 && t == T_B) {
            stamp();                                   // chains up to T_B - 1 issued
            if (wave >= 4 && pnl + 1 < p_end) convert(pnl + 1, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            stamp();                                   // second group's conversion done
          }
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            u32x4 vbw;
            if (!from_lds) vbw = bw[i < RB ? i : 0][ks];
            else vbw = bl_next;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], __builtin_bit_cast(bf16x8, vbw), acc, 0, 0, 0);
            if constexpr (FIRST && CAN_PEEL) {            // this k-step's fragment of the block two chains ahead
              if (t < 2) bw[(t + 2) < RB ? t + 2 : 0][ks] = (wfh + (long long)cbs[(t + 2) < CBW ? t + 2 : 0] * KS * 64 + lane)[ks * 64];
            }
            // the other row block's A fragments replace this one's as the last chain that needs them passes
            if (t == CBW - 1) af[ks] = *reinterpret_cast<const bf16x8*>(&Ap[buf][0][(32 + r) * PA + ks * 16 + h * 8]);
            // next LDS-resident fragment: the next k-step of this chain, or the first of the next chain
            {
              const int tn = ks + 1 < KS ? t : t + 1, kn = ks + 1 < KS ? ks + 1 : 0;
              const int in = tn % CBW;
              if (tn < NB && in * NT >= RB) bl_next = Bl[wave][(in * NT - RB) * KS + kn][lane];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0) {
#pragma unroll
              for (int u = 0; u < 7; ++u)
                if (gap_of(u) == ks) unit(u, acc2[(t - 1) & 1], (t - 1) % CBW, pnl * 2 + (t - 1) / CBW);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 7; ++u) unit(u, acc2[(NB - 1) & 1], (NB - 1) % CBW, pnl * 2 + 1);
        stamp();                                       // last chain and its epilogue done
        __syncthreads();
        stamp();                                       // 4..: a panel done
        return;
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {                    // the panel's two 32-row blocks in turn
      // this wave's A fragments of the block: read from LDS once and used for all its column blocks (KS x 4 VGPRs per image)
      asm volatile("" ::: "memory");
      bf16x8 af[KS], afl[NS == 3 ? KS : 1];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int oa = (m * 32 + r) * PA + ks * 16 + h * 8;
        af[ks] = *reinterpret_cast<const bf16x8*>(&Ap[buf][0][oa]);
        if (NS == 3) afl[ks] = *reinterpret_cast<const bf16x8*>(&Ap[buf][NT - 1][oa]);
      }
#pragma unroll
      for (int i = 0; i < CBW; ++i) {                // one 32 x 32 accumulator (16 registers) at a time
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 vb = __builtin_bit_cast(bf16x8, (i * NT < RB) ? bw[i * NT < RB ? i * NT : 0][ks] : Bl[wave][(i * NT - RB) * KS + ks][lane]);
          if (NS == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afl[ks], vb, acc, 0, 0, 0);
          if (!(DBG & 8)) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], vb, acc, 0, 0, 0);
          else acc[0] += (float)af[ks][0] * (float)vb[0];
        }
        if (NS == 3) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 vb = __builtin_bit_cast(bf16x8, (i * NT + 1 < RB) ? bw[i * NT + 1 < RB ? i * NT + 1 : 0][ks] : Bl[wave][(i * NT + 1 - RB) * KS + ks][lane]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], vb, acc, 0, 0, 0);
          }
        }
        // ---- epilogue: this lane's column over its 16 rows of the 32-row block (the other half-wave holds the other 16) ----------
        if (DBG & 1) {                                 // ablation: keep the accumulator alive, skip the epilogue
          ss[i] += acc[0] + acc[15];
          continue;
        }
        if (STATS) {
          f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};     // two independent packed chains (v_pk_fma_f32)
#pragma unroll
          for (int e = 0; e < 16; e += 4) {
            const f32x2 v0 = {acc[e], acc[e + 1]}, v1 = {acc[e + 2], acc[e + 3]};
            s0 = __builtin_elementwise_fma(v0, v0, s0);
            s1 = __builtin_elementwise_fma(v1, v1, s1);
          }
          ss[i] += (s0.x + s0.y) + (s1.x + s1.y);
        }
        float mx;
        if (full) {
          mx = max16_guarded(acc);                 // (the software-pipelined form above takes the full panels of the bf16 mode)
        } else {
          mx = -INFINITY;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int il = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            mx = fmaxf(mx, il < nrows ? acc[e] : -INFINITY);
          }
        }
        const bool better = mx > best[i];              // panels and blocks ascend: the first maximum wins
        best[i] = better ? mx : best[i];
        bq[i] = better ? (pnl * 2 + m) : bq[i];
        // pin the epilogue here: left alone, the scheduler sinks the sum-of-squares chains of all eight blocks of a panel behind
        // the last MFMA and keeps 8 x 16 accumulator registers alive for them (spills); the other wave of the SIMD covers the gap
        if (STATS) asm volatile("" : "+v"(ss[i]));
        asm volatile("" : "+v"(best[i]));
        // bf16 mode: only the ragged last panel of a cloud comes here -- one accumulator at a time, nothing carried across blocks
        // (registers, not speed: the pipelined form above sets this kernel's register budget)
        if constexpr (NS == 1 && (DBG & ~16) == 0) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (pnl + 1 < p_end) convert(pnl + 1, buf ^ 1);    // the other image was last read before the previous barrier
    __syncthreads();
  };
  {
    int pnl = p_begin;
    if (peel) {
      do_panel(pnl, BoolTag<true>{});
      ++pnl;
    }
    for (; pnl < p_end; ++pnl) do_panel(pnl, BoolTag<false>{});
  }

  // ---- flush the run: one value per (slot, channel) --------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < CBW; ++i) {
    const float ob = __shfl_xor(best[i], 32, 64);
    const int oq = __shfl_xor(bq[i], 32, 64);
    const bool take = ob > best[i] || (ob == best[i] && oq < bq[i]);
    const float bv = take ? ob : best[i];
    const int bqv = take ? oq : bq[i];
    float sv = ss[i];
    if (STATS || (DBG & 1)) sv += __shfl_xor(sv, 32, 64);
    if (h == 0) {
      const long long o = (long long)slot * g.C + cbs[i] * 32 + r;
      g.pmax[o] = bv;
      g.pq[o] = bqv;
      if (STATS && g.sumsq) g.sumsq[o] = sv;
    }
  }
  stamp();                                           // maxima and sums of squares written
  if (STATS && g.colacc) {
    // Column sums a1 = A^T 1 of the slot's staged rows, gathered from the eight waves (lanes l, l + CH, ... of a wave hold the same 8
    // columns -> butterfly, one slot per (wave, column) in LDS, 8-way sum) and added to the CLOUD's accumulators as 64-bit fixed
    // point (integer addition is associative: the sum does not depend on the order the cloud's workgroups arrive in).  The
    // finaliser turns them into the channel sums of z: sum over ALL rows of z[:, c] = (sum over the clouds of a1) . W[:, c] -- K*C
    // multiply-adds per LAUNCH.  (Round 2 formed a1 . W here, per slot: K*C multiply-adds per workgroup, 5,200 cycles of an
    // 11,000-cycle run at N = 1024.)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      for (int o = CH; o < 64; o <<= 1) {
        a1h[e] += __shfl_xor(a1h[e], o, 64);
        if (NS == 3) a1l[e] += __shfl_xor(a1l[e], o, 64);
      }
    }
    if (lane < CH) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[wave][lane * 8 + e] = a1h[e];
        if (NS == 3) red[wave][K + lane * 8 + e] = a1l[e];
      }
    }
    __syncthreads();
    for (int col = tid; col < NT * K; col += THREADS) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[w][col];
      if (cg == 0)       // (bf16x3: the workgroups of the second column half staged the same rows)
        __hip_atomic_fetch_add(&g.colacc[(long long)cloud * (NT * K) + col], __float2ll_rn(t * 16777216.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if constexpr ((DBG & 16) != 0) {
    stamp();                                         // end
    __syncthreads();
    if (tid < 128) g.pq[(long long)slot * g.C + 1 + tid] = (tid & 63) < n_stamp ? (int)stamps[tid >> 6][tid & 63] : 0;
    if (tid == 0) g.pq[(long long)slot * g.C] = n_stamp;
  }
}

template <int NS, int K, int CBW>
static void launch_panel(const PanelArgs& g, dim3 grid, bool stats, hipStream_t st) {
  const int dbg = getenv("PN_PANEL_DBG") ? atoi(getenv("PN_PANEL_DBG")) : 0;   // read at every call: the probe flips it between runs
  if constexpr (NS == 1 && K == 128 && CBW == 4) {
    if (dbg && stats) {
#define PN_PANEL_DBG_CASE(D)                                                                            \
  if (dbg == D) {                                                                                       \
    hipLaunchKernelGGL((panel_max_kernel<1, 128, true, 4, D>), grid, dim3(512), 0, st, g);             \
    return;                                                                                             \
  }
      PN_PANEL_DBG_CASE(1) PN_PANEL_DBG_CASE(4) PN_PANEL_DBG_CASE(8) PN_PANEL_DBG_CASE(9) PN_PANEL_DBG_CASE(13)
#undef PN_PANEL_DBG_CASE
      if (dbg == 16 && g.a.h16) {
        hipLaunchKernelGGL((panel_max_kernel<1, 128, true, 4, 16, true>), grid, dim3(512), 0, st, g);
        return;
      }
    }
  }
  if constexpr (NS == 1) {
    if (g.a.h16) {
      if (stats) hipLaunchKernelGGL((panel_max_kernel<NS, K, true, CBW, 0, true>), grid, dim3(512), 0, st, g);
      else hipLaunchKernelGGL((panel_max_kernel<NS, K, false, CBW, 0, true>), grid, dim3(512), 0, st, g);
      return;
    }
  }
  if (stats) hipLaunchKernelGGL((panel_max_kernel<NS, K, true, CBW>), grid, dim3(512), 0, st, g);
  else hipLaunchKernelGGL((panel_max_kernel<NS, K, false, CBW>), grid, dim3(512), 0, st, g);
}
template <int NS, int K>
static void launch_panel_cbw(const PanelArgs& g, int C, bool stats, hipStream_t st) {
  constexpr int CBW_MAX = (NS == 3) ? 2 : 4;
  int cbw = C / 256;                                   // column blocks per wave if one workgroup owned every column
  if (cbw > CBW_MAX) cbw = CBW_MAX;
  if (NS == 1 && cbw == 4 && panel_split_mode() != 0) cbw = 2;
  const dim3 grid(g.B * g.spc, C / (256 * cbw));
  if (cbw == 4) { if constexpr (CBW_MAX >= 4) launch_panel<NS, K, 4>(g, grid, stats, st); }
  else if (cbw == 2) launch_panel<NS, K, 2>(g, grid, stats, st);
  else launch_panel<NS, K, 1>(g, grid, stats, st);
}

int conv_fwd_max_panel(const pn_operand* x, const void* wf_hi, const void* wf_lo, int B, int N, int K, int C, float* pmax, int* pq,
                       float* sumsq, long long* colacc, int prec, hipStream_t st) {
  PN_CHECK_ARG(x && x->s1 && !x->s2, "pn_conv_fwd_max_panel: bad operand");
  PN_CHECK_ARG((reinterpret_cast<uintptr_t>(x->s1) & 15) == 0 && x->ld % 4 == 0 && x->ld >= K, "pn_conv_fwd_max_panel: operand alignment");
  PN_CHECK_ARG(B > 0 && N > 0, "pn_conv_fwd_max_panel: B and N must be positive");
  PN_CHECK_ARG(K == 64 || K == 128, "pn_conv_fwd_max_panel: K must be 64 or 128 (K=%d)", K);
  PN_CHECK_ARG(C >= 256 && C % 256 == 0 && (C / 256 == 1 || C / 256 == 2 || C % 1024 == 0), "pn_conv_fwd_max_panel: C must be 256, 512 or a multiple of 1024 (C=%d)", C);
  PN_CHECK_ARG(wf_hi && pmax && pq, "pn_conv_fwd_max_panel: null pointer");
  PN_CHECK_ARG((sumsq == nullptr) == (colacc == nullptr), "pn_conv_fwd_max_panel: sumsq and colacc come together (both or neither)");
  prec &= ~PN_STORE_BF16;
  PN_CHECK_ARG(prec == PN_PREC_BF16 || (prec == PN_PREC_BF16X3 && wf_lo), "pn_conv_fwd_max_panel: bad prec / missing lo weights");
  PN_CHECK_ARG(x->h16 == 0 || (x->h16 == 1 && prec == PN_PREC_BF16 && x->ld % 8 == 0),
               "pn_conv_fwd_max_panel: a 16-bit operand needs PN_PREC_BF16 and ld %% 8 == 0");
  PanelArgs g;
  memset(&g, 0, sizeof(g));
  g.a = *x; g.wf_hi = reinterpret_cast<const __bf16*>(wf_hi); g.wf_lo = reinterpret_cast<const __bf16*>(wf_lo);
  g.B = B; g.N = N; g.C = C;
  g.spc = panel_slots_per_cloud(B, N);
  g.pmax = pmax; g.pq = pq; g.sumsq = sumsq; g.colacc = colacc;
  const bool st_ = sumsq != nullptr;
  if (prec == PN_PREC_BF16X3) {
    if (K == 128) launch_panel_cbw<3, 128>(g, C, st_, st);
    else launch_panel_cbw<3, 64>(g, C, st_, st);
  } else {
    if (K == 128) launch_panel_cbw<1, 128>(g, C, st_, st);
    else launch_panel_cbw<1, 64>(g, C, st_, st);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// ---- inference: a whole max-pooled chain in ONE launch ---------------------------------------------------------------------------
// north_star's named kernel is the per-point MLP + global max: ConvLayer(. -> 64) -> ConvLayer(64 -> 128) -> ConvLayer(128 -> 1024) ->
// reduce_max (PointNet.py:236-248 for mlp_2, :421-429 for the two T-Nets).  With moving statistics (inference) no layer waits for a
// batch statistic, so a workgroup takes its run of 64-row panels through all three layers: the two narrow layers' outputs live in LDS
// only (no (B*N, 64) / (B*N, 128) tensor is written or read back), the wide layer is the kernel-stationary panel product above.
//   FRONT 0: the chain starts from the normalised cloud (B*N, 3): first layer on the vector ALU with conv3_fwd's own fma chain;
//   FRONT 1: it starts from a 64-channel lazy operand stored as bf16 (X_64, or relu(bn(mlp_1_2))): first layer on the matrix cores.
// Every value is rounded where the layer-by-layer plan rounds it (the stored pre-BN output to bf16, the BN + ReLU result to the bf16
// MFMA operand) and every contraction runs in the same k order on the same instruction: pmax / pq are BIT-IDENTICAL to the three
// launches they replace (tests/test_gpu_ops.py::test_chain_kernel_equals_the_layered_launches).
struct ChainArgs {
  pn_operand x;                 // FRONT 1: (B*N, 64) lazy operand, bf16 storage
  const float* xyz;             // FRONT 0: (B*N, 3)
  const float* w1f;             // FRONT 0: the first layer's (3, 64) kernel
  const unsigned short* w1t;    // FRONT 1: bf16 copy [64][64] of the first layer's kernel, TRANSPOSED ([Cout][K], pn_prologue.hip)
  const float *sc1, *sh1;       // BatchNormalization scale / shift (moving statistics) of the first layer
  const unsigned short* w2t;    // bf16 copy [128][64] of the second layer's kernel, transposed
  const float *sc2, *sh2;
  const __bf16* wf_hi;          // third layer: fragment-ordered, presigned copy (weights_prep)
  int B, N, spc;
  float* pmax; int* pq;         // as PanelArgs
};
// one 32 x 32 accumulator block (lane = column col, 16 rows) -> bf16 LDS image, pairs of adjacent columns as one dword (the DPP swap
// of pn_segout.hip: sub-dword LDS stores from 64 lanes serialise)
__device__ __forceinline__ void chain_store_block(__bf16* img, int pitch, int m, int col, int h, int lane, const float (&y)[16]) {
  const bool odd = lane & 1;
#pragma unroll
  for (int e = 0; e < 16; e += 2) {
    const float give = odd ? y[e] : y[e + 1];
    const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));
    const int ee = odd ? e + 1 : e;
    const int row = m * 32 + (ee & 3) + 8 * (ee >> 2) + 4 * h;
    const float lo = odd ? got : y[e], hi = odd ? y[e + 1] : got;
    const unsigned w = (unsigned)f32_bf16_bits(lo) | ((unsigned)f32_bf16_bits(hi) << 16);
    *reinterpret_cast<unsigned*>(img + row * pitch + (col & ~1)) = w;
  }
}
// the stored-then-reloaded form of a layer output: pre-BN value rounded to bf16 (the plan's 16-bit store), BN + ReLU in fp32
__device__ __forceinline__ float chain_bnrelu(float z, float sc, float sh) {
  return clamp_lo(fmaf(sc, bf16_bits_f32(f32_bf16_bits(z)), sh), 0.f);
}
template <int FRONT>
__global__ __launch_bounds__(512) void chain_max_kernel(const ChainArgs g) {
  constexpr int BM = 64, K = 128, KS = K / 16, PA = K + 8, P64 = 64 + 8, CBW = 4, C = 1024;
  __shared__ __attribute__((aligned(16))) __bf16 Ap[BM * PA];      // the wide layer's operand panel
  __shared__ __attribute__((aligned(16))) __bf16 In0[BM * P64];    // FRONT 1: the chain's input panel in operand precision
  __shared__ __attribute__((aligned(16))) __bf16 A1[BM * P64];     // relu(bn(first layer)) in operand precision
  __shared__ u32x4 Bl[8][KS][64];                                  // [wave][k-step][lane]: the fourth column block's fragments (registers: three)
  __shared__ __attribute__((aligned(16))) float ctab[5 * 64];      // FRONT 0: w0, w1, w2, scale, shift per channel; FRONT 1: the operand's ca, cc
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int slot = blockIdx.x;
  const int cloud = slot / g.spc, j = slot - cloud * g.spc;
  const int tpc = (g.N + BM - 1) / BM;
  const int p_begin = (int)((long long)j * tpc / g.spc), p_end = (int)((long long)(j + 1) * tpc / g.spc);
  const long long cloud_row0 = (long long)cloud * g.N;

  // ---- what does not depend on the panel: requested once ----
  // second layer: wave w owns (row block w & 1, column block w >> 1) of the 64 x 128 output
  const int rb2 = wave & 1, cb2 = wave >> 1;
  const float sc2 = g.sc2[32 * cb2 + r], sh2 = g.sh2[32 * cb2 + r];
  u32x4 l2b[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) l2b[ks] = *reinterpret_cast<const u32x4*>(g.w2t + (long long)(32 * cb2 + r) * 64 + ks * 16 + 8 * h);
  // first layer
  const int rb1 = wave & 1, cb1 = (wave >> 1) & 1;                  // FRONT 1: waves 0..3 own the four 32 x 32 blocks of the 64 x 64 output
  float sc1 = 1.f, sh1 = 0.f;
  u32x4 l1b[FRONT == 1 ? 4 : 1];
  const int frow = tid >> 3, fch = (tid & 7) * 8;                   // FRONT 0 / staging: thread <-> (row of the panel, 8 channels)
  if constexpr (FRONT == 1) {
    sc1 = g.sc1[32 * cb1 + r]; sh1 = g.sh1[32 * cb1 + r];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) l1b[ks] = *reinterpret_cast<const u32x4*>(g.w1t + (long long)(32 * cb1 + r) * 64 + ks * 16 + 8 * h);
    if (tid < 64) {
      ctab[tid] = g.x.ca ? g.x.ca[tid] : 1.f;
      ctab[64 + tid] = g.x.cc ? g.x.cc[tid] : 0.f;
    }
  } else {
    if (tid < 64) {
      ctab[tid] = g.w1f[tid]; ctab[64 + tid] = g.w1f[64 + tid]; ctab[128 + tid] = g.w1f[128 + tid];
      ctab[192 + tid] = g.sc1[tid]; ctab[256 + tid] = g.sh1[tid];
    }
  }
  // third layer: this wave's four column blocks, resident for the whole run (as panel_max_kernel)
  const u32x4* __restrict__ wfh = reinterpret_cast<const u32x4*>(g.wf_hi);
  constexpr int RB = 3;
  u32x4 bw[RB][KS];
  int cbs[CBW];
#pragma unroll
  for (int i = 0; i < CBW; ++i) {
    cbs[i] = i * 8 + wave;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 f = (wfh + (long long)cbs[i] * KS * 64 + lane)[ks * 64];
      if (i < RB) bw[i < RB ? i : 0][ks] = f;
      else Bl[wave][ks][lane] = f;                     // read back by this wave only
    }
  }
  float best[CBW];
  int bq[CBW];
#pragma unroll
  for (int i = 0; i < CBW; ++i) { best[i] = -INFINITY; bq[i] = 0; }
  __syncthreads();                                   // the coefficient table

  for (int pnl = p_begin; pnl < p_end; ++pnl) {
    const int rbase = pnl * BM, nrows = min(BM, g.N - rbase);
    const bool full = nrows == BM;
    const long long rsrc = cloud_row0 + rbase + (frow < nrows ? frow : nrows - 1);
    // ---- first layer -> A1 ----
    if constexpr (FRONT == 1) {
      float v[8], xca[8], xcc[8];
      bf16x8_unpack(act_load8_raw(g.x.s1, rsrc * g.x.ld + fch), v);
      *reinterpret_cast<float4*>(&xca[0]) = *reinterpret_cast<const float4*>(&ctab[fch]);
      *reinterpret_cast<float4*>(&xca[4]) = *reinterpret_cast<const float4*>(&ctab[fch + 4]);
      *reinterpret_cast<float4*>(&xcc[0]) = *reinterpret_cast<const float4*>(&ctab[64 + fch]);
      *reinterpret_cast<float4*>(&xcc[4]) = *reinterpret_cast<const float4*>(&ctab[64 + fch + 4]);
      u32x4 pk;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const f32x2 tt = {clamp_lo(fmaf(xca[e], v[e], xcc[e]), g.x.lo), clamp_lo(fmaf(xca[e + 1], v[e + 1], xcc[e + 1]), g.x.lo)};
        pk[e / 2] = frow < nrows ? __builtin_bit_cast(unsigned, __builtin_convertvector(tt, bf16x2)) : 0u;
      }
      *reinterpret_cast<u32x4*>(&In0[frow * P64 + fch]) = pk;
      __syncthreads();
      if (wave < 4) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(&In0[(rb1 * 32 + r) * P64 + ks * 16 + 8 * h]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, l1b[ks]), acc, 0, 0, 0);
        }
        float y[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) y[e] = chain_bnrelu(acc[e], sc1, sh1);
        chain_store_block(A1, P64, rb1, 32 * cb1 + r, h, lane, y);
      }
    } else {
      const float a0 = g.xyz[rsrc * 3], a1 = g.xyz[rsrc * 3 + 1], a2 = g.xyz[rsrc * 3 + 2];
      float w0[8], w1[8], w2[8], s1v[8], h1v[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<float4*>(&w0[4 * q]) = *reinterpret_cast<const float4*>(&ctab[fch + 4 * q]);
        *reinterpret_cast<float4*>(&w1[4 * q]) = *reinterpret_cast<const float4*>(&ctab[64 + fch + 4 * q]);
        *reinterpret_cast<float4*>(&w2[4 * q]) = *reinterpret_cast<const float4*>(&ctab[128 + fch + 4 * q]);
        *reinterpret_cast<float4*>(&s1v[4 * q]) = *reinterpret_cast<const float4*>(&ctab[192 + fch + 4 * q]);
        *reinterpret_cast<float4*>(&h1v[4 * q]) = *reinterpret_cast<const float4*>(&ctab[256 + fch + 4 * q]);
      }
      u32x4 pk;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const float v0 = fmaf(a2, w2[e], fmaf(a1, w1[e], a0 * w0[e]));               // conv3_fwd's chain
        const float v1 = fmaf(a2, w2[e + 1], fmaf(a1, w1[e + 1], a0 * w0[e + 1]));
        const f32x2 tt = {chain_bnrelu(v0, s1v[e], h1v[e]), chain_bnrelu(v1, s1v[e + 1], h1v[e + 1])};
        pk[e / 2] = __builtin_bit_cast(unsigned, __builtin_convertvector(tt, bf16x2));
      }
      *reinterpret_cast<u32x4*>(&A1[frow * P64 + fch]) = pk;
    }
    __syncthreads();
    // ---- second layer -> the wide layer's operand panel ----
    {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(&A1[(rb2 * 32 + r) * P64 + ks * 16 + 8 * h]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, l2b[ks]), acc, 0, 0, 0);
      }
      float y[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) y[e] = chain_bnrelu(acc[e], sc2, sh2);
      chain_store_block(Ap, PA, rb2, 32 * cb2 + r, h, lane, y);
    }
    __syncthreads();
    // ---- wide layer + running maximum (the plain loop of panel_max_kernel: same chains, same association) ----
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      asm volatile("" ::: "memory");
      bf16x8 af[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(&Ap[(m * 32 + r) * PA + ks * 16 + h * 8]);
#pragma unroll
      for (int i = 0; i < CBW; ++i) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const u32x4 vb = i < RB ? bw[i < RB ? i : 0][ks] : Bl[wave][ks][lane];
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], __builtin_bit_cast(bf16x8, vb), acc, 0, 0, 0);
        }
        float mx;
        if (full) {
          mx = max16_guarded(acc);
        } else {
          mx = -INFINITY;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int il = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            mx = fmaxf(mx, il < nrows ? acc[e] : -INFINITY);
          }
        }
        const bool better = mx > best[i];
        best[i] = better ? mx : best[i];
        bq[i] = better ? (pnl * 2 + m) : bq[i];
        asm volatile("" : "+v"(best[i]));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                                 // the three images are rewritten by the next panel
  }
#pragma unroll
  for (int i = 0; i < CBW; ++i) {
    const float ob = __shfl_xor(best[i], 32, 64);
    const int oq = __shfl_xor(bq[i], 32, 64);
    const bool take = ob > best[i] || (ob == best[i] && oq < bq[i]);
    if (h == 0) {
      const long long o = (long long)slot * C + cbs[i] * 32 + r;
      g.pmax[o] = take ? ob : best[i];
      g.pq[o] = take ? oq : bq[i];
    }
  }
}

int chain_fwd_max(const pn_operand* x, const float* xyz, const float* w1, const void* w1t, const float* sc1, const float* sh1, const void* w2t,
                  const float* sc2, const float* sh2, const void* wf_hi, int B, int N, float* pmax, int* pq, hipStream_t st) {
  PN_CHECK_ARG((x != nullptr) != (xyz != nullptr), "pn_chain_fwd_max: exactly one of the 64-channel operand and the xyz cloud");
  PN_CHECK_ARG(sc1 && sh1 && w2t && sc2 && sh2 && wf_hi && pmax && pq && B > 0 && N > 0, "pn_chain_fwd_max: bad arguments");
  PN_CHECK_ARG((reinterpret_cast<uintptr_t>(w2t) & 15) == 0 && (reinterpret_cast<uintptr_t>(wf_hi) & 15) == 0, "pn_chain_fwd_max: kernel copies must be 16-byte aligned");
  ChainArgs g;
  memset(&g, 0, sizeof(g));
  g.sc1 = sc1; g.sh1 = sh1; g.w2t = reinterpret_cast<const unsigned short*>(w2t); g.sc2 = sc2; g.sh2 = sh2;
  g.wf_hi = reinterpret_cast<const __bf16*>(wf_hi);
  g.B = B; g.N = N; g.spc = panel_slots_per_cloud(B, N); g.pmax = pmax; g.pq = pq;
  if (x) {
    PN_CHECK_ARG(x->s1 && !x->s2 && x->h16 == 1 && x->ld >= 64 && x->ld % 8 == 0 && (reinterpret_cast<uintptr_t>(x->s1) & 15) == 0 && w1t &&
                     (reinterpret_cast<uintptr_t>(w1t) & 15) == 0,
                 "pn_chain_fwd_max: the 64-channel operand must be a 16-byte aligned bf16 array with the first kernel's bf16 copy");
    g.x = *x; g.w1t = reinterpret_cast<const unsigned short*>(w1t);
    hipLaunchKernelGGL(chain_max_kernel<1>, dim3(B * g.spc), dim3(512), 0, st, g);
  } else {
    PN_CHECK_ARG(w1 != nullptr, "pn_chain_fwd_max: the first layer's (3, 64) kernel is required");
    g.xyz = xyz; g.w1f = w1;
    hipLaunchKernelGGL(chain_max_kernel<0>, dim3(B * g.spc), dim3(512), 0, st, g);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
}

// ---- finaliser: BatchNormalization statistics of the layer + reduce_max over each cloud's slots ------------------------------------
// One workgroup per (32 channels, slice of the clouds).  Training statistics (use_batch): the slots' column sums of the staged operand
// are added up (fp64) into A1 = A^T 1 over ALL rows and sum z[:, c] = A1 . W[:, c] is formed from the same fragment-ordered bf16 copy
// of the kernel the panel launch multiplied with (hi, and for bf16x3 the three cross terms the matrix cores summed); the slots' sums of
// squares are added per channel; combined in fp64 -> mean (times the channel's sign: the panel kernel works on sgn * z), invstd,
// scale, shift, moving statistics (only the workgroups of the first cloud slice write them); otherwise the coefficients come from the
// moving statistics (inference / frozen layer, PointNet.py:585-591).  Then per cloud: the largest pmax over its slots (lowest slot on
// ties: rows ascend with the slot index), zstar = s_c * max, g = relu(scale * zstar + shift), and the 32-row block that holds the row.
struct PanelFinArgs {
  const float* pmax; const int* pq; const float* sumsq; const long long* colacc;
  const unsigned short *wf_hi, *wf_lo;      // fragment-ordered kernel copies (pn_weights_prep); wf_lo for bf16x3
  int T, tpc, B, C, K, NT, n_blocks32;
  double inv_count;
  const float* gamma; const float* beta; float* mm; float* mv;
  float momentum, eps;
  int use_batch, update;
  float *mean, *invstd, *scale, *shift, *g, *zstar;
  int* argq;
};
constexpr int PANEL_FIN_MAXK = 128;
__device__ __forceinline__ void panel_finalize_body(const PanelFinArgs& a, const int bxi, const int byi) {
  __shared__ double red[8][2][32];
  __shared__ double A1[2 * PANEL_FIN_MAXK];
  __shared__ long long A1part[2 * PANEL_FIN_MAXK];   // [group][column]: 256 entries whatever the split
  __shared__ float sc_s[32], sh_s[32], sg_s[32];
  const int tid = threadIdx.x, cl = tid & 31, part = tid >> 5;
  const int c = bxi * 32 + cl;
  const float gam = a.gamma[c];
  const float sg = gam < 0.f ? -1.f : 1.f;
  // the first cloud's slot maxima of this thread: requested now, used after the statistics
  const int b0 = byi * 8 + part;
  float pre_v[8];
  int pre_q[8];
  {
    const long long base0 = (long long)(b0 < a.B ? b0 : 0) * a.tpc * a.C + c;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = min(u, a.tpc - 1);
      pre_v[u] = a.pmax[base0 + (long long)t * a.C];
      pre_q[u] = a.pq[base0 + (long long)t * a.C];
    }
  }
  if (a.use_batch) {
    double s1 = 0.0, s2 = 0.0;
    // Everything the launch reads is requested up front -- the kernel fragments of this thread's k-step, the clouds' column-sum
    // accumulators (thread <-> column, every `groups`-th cloud) and the slots' sums of squares (thread <-> channel, every 8th slot, 32
    // deep): ONE round trip to memory for T <= 256 slots (the per-cloud maxima further down are requested above: pre_v / pre_q)
    const int ncol = a.NT * a.K, groups = 256 / ncol;
    const int KS = a.K / 16;
    const long long cb = c >> 5;
    const int ksw = part < KS ? part : KS - 1;            // (k-steps beyond the first eight: loaded in the loop below)
    uint4 wraw[2][2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const long long fo = ((cb * KS + ksw) * 64 + hh * 32 + cl) * 8;
      wraw[hh][0] = *reinterpret_cast<const uint4*>(a.wf_hi + fo);
      wraw[hh][1] = a.NT == 2 ? *reinterpret_cast<const uint4*>(a.wf_lo + fo) : make_uint4(0, 0, 0, 0);
    }
    {
      // A1[col] = sum over the clouds of colacc[cloud][col]: 64-bit integers (exact, order-free); the groups' partial sums meet in LDS
      const int col = tid % ncol, gi = tid / ncol;
      long long ta = 0;
      double s2a = 0.0;
      for (int pc = gi, ps = part; pc < a.B || ps < a.T; pc += groups * 16, ps += 8 * 32) {
        long long v[16];
        float q[32];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int pp = pc + u * groups;
          v[u] = a.colacc[(long long)(pp < a.B ? pp : 0) * ncol + col];
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) {
          const int pp = ps + u * 8;
          q[u] = a.sumsq[(long long)(pp < a.T ? pp : 0) * a.C + c];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (pc + u * groups < a.B) ta += v[u];
#pragma unroll
        for (int u = 0; u < 32; ++u)
          if (ps + u * 8 < a.T) s2a += (double)q[u];
      }
      A1part[gi * ncol + col] = ta;
      s2 = s2a;
    }
    __syncthreads();
    if (tid < ncol) {
      long long t = 0;
      for (int q = 0; q < groups; ++q) t += A1part[q * ncol + tid];
      A1[tid] = (double)t * (1.0 / 16777216.0);
    }
    __syncthreads();
    // sum z = A1 . W[:, c]: thread (channel, part) takes the k-steps part, part + 8, ... of its column: the sixteen k of a step are the
    // two 16-byte fragments of lanes (c % 32) and (c % 32) + 32
    for (int ks = part; ks < KS; ks += 8) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        uint4 wh = wraw[hh][0], wl = wraw[hh][1];
        if (ks != ksw) {
          const long long fo = ((cb * KS + ks) * 64 + hh * 32 + cl) * 8;
          wh = *reinterpret_cast<const uint4*>(a.wf_hi + fo);
          if (a.NT == 2) wl = *reinterpret_cast<const uint4*>(a.wf_lo + fo);
        }
        float whf[8], wlf[8];
        bf16x8_unpack(wh, whf);
        bf16x8_unpack(wl, wlf);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = ks * 16 + hh * 8 + e;
          s1 += A1[k] * (double)whf[e];
          if (a.NT == 2) s1 += A1[a.K + k] * (double)whf[e] + A1[k] * (double)wlf[e];
        }
      }
    }
    red[part][0][cl] = s1;
    red[part][1][cl] = s2;
    __syncthreads();
    if (part == 0) {
      s1 = 0.0; s2 = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) { s1 += red[q][0][cl]; s2 += red[q][1][cl]; }
      const double mean = (double)sg * s1 * a.inv_count;
      double var = s2 * a.inv_count - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      const float is = 1.0f / sqrtf((float)var + a.eps);
      const float scl = gam * is;
      const float sft = a.beta[c] - (float)mean * scl;
      sc_s[cl] = scl; sh_s[cl] = sft; sg_s[cl] = sg;
      if (byi == 0) {
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
    if (byi == 0) { a.mean[c] = mean; a.invstd[c] = is; a.scale[c] = scl; a.shift[c] = sft; }
  }
  __syncthreads();
  // per cloud: thread <-> (channel, cloud); this block's slice of the clouds
  const float scl = sc_s[cl], sft = sh_s[cl], sgc = sg_s[cl];
  for (int b = byi * 8 + part; b < a.B; b += 8 * gridDim.y) {
    float best = -INFINITY;
    int bq = 0;
    const long long base = (long long)b * a.tpc * a.C + c;
    for (int t0 = 0; t0 < a.tpc; t0 += 8) {
      float v[8];
      int q[8];
      if (b == b0 && t0 == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { v[u] = pre_v[u]; q[u] = pre_q[u]; }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = min(t0 + u, a.tpc - 1);
          v[u] = a.pmax[base + (long long)t * a.C];
          q[u] = a.pq[base + (long long)t * a.C];
        }
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
__global__ __launch_bounds__(256) void panel_finalize_kernel(const PanelFinArgs a) { panel_finalize_body(a, blockIdx.x, blockIdx.y); }
// Round 3: the finaliser is 128 workgroups on a 256-CU chip, and the Gram matrix A^T A (+ column sums) of the layer's INPUT -- what
// the Gram-form backward of this layer needs, formed from forward activations only -- was a weight-gradient launch of the backward
// pass's tail (three layers: 17.8 us).  Its workgroups ride on the block ids behind the finaliser's (training, layer trainable).
__global__ __launch_bounds__(256) void panel_finalize_gram_kernel(const PanelFinArgs a, int nx, int n_fin, const WgradBatch wb) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[GemmLds<128, 128, 1>::BYTES];
  if ((int)blockIdx.x < n_fin) {                  // block-uniform
    panel_finalize_body(a, (int)blockIdx.x % nx, (int)blockIdx.x / nx);
    return;
  }
  wgrad_batch_tile<128, 128, 1, false>(wb, (int)blockIdx.x - n_fin, lds_raw);
}

int panel_finalize(const float* pmax, const int* pq, const float* sumsq, const long long* colacc, const void* wf_hi, const void* wf_lo, int prec,
                   int B, int N, int K, int C, const float* gamma, const float* beta, float* mm, float* mv, float momentum, float eps,
                   int use_batch, int update, float* mean, float* invstd, float* scale, float* shift, float* g, float* zstar, int* argq,
                   hipStream_t st, int count_mult, const WgradDesc* gram) {
  PN_CHECK_ARG(pmax && pq && gamma && beta && mm && mv && mean && invstd && scale && shift && g, "pn_panel_finalize: null pointer");
  prec &= ~PN_STORE_BF16;
  PN_CHECK_ARG(!use_batch || (sumsq && colacc && wf_hi && (prec != PN_PREC_BF16X3 || wf_lo)),
               "pn_panel_finalize: batch statistics need sumsq, colacc and the kernel copies");
  PN_CHECK_ARG(B > 0 && N > 0 && C > 0 && C % 32 == 0, "pn_panel_finalize: bad sizes");
  PN_CHECK_ARG(!use_batch || ((K == 64 || K == 128) && (prec == PN_PREC_BF16 || prec == PN_PREC_BF16X3)), "pn_panel_finalize: K must be 64 or 128 (K=%d), prec bf16 or bf16x3", K);
  PanelFinArgs a;
  memset(&a, 0, sizeof(a));
  a.pmax = pmax; a.pq = pq; a.sumsq = sumsq; a.colacc = colacc;
  a.wf_hi = reinterpret_cast<const unsigned short*>(wf_hi); a.wf_lo = reinterpret_cast<const unsigned short*>(wf_lo);
  a.K = K; a.NT = prec == PN_PREC_BF16X3 ? 2 : 1;
  a.tpc = panel_slots_per_cloud(B, N); a.T = B * a.tpc; a.B = B; a.C = C;
  a.n_blocks32 = cdiv(N, 32);
  a.inv_count = 1.0 / ((double)B * (double)N * (double)(count_mult > 0 ? count_mult : 1));
  a.gamma = gamma; a.beta = beta; a.mm = mm; a.mv = mv; a.momentum = momentum; a.eps = eps; a.use_batch = use_batch; a.update = update;
  a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift; a.g = g; a.zstar = zstar; a.argq = argq;
  const int slices = B >= 32 ? 4 : (B >= 16 ? 2 : 1);      // the statistics are recomputed per slice: a few, for parallelism over the clouds
  if (gram) {          // the Gram matrix of the layer's input rides behind the finaliser's workgroups (128 x 128 tiles, bf16 operands)
    WgradBatch wb;
    int wblocks = 0;
    PN_CHECK_ARG((gram->prec & ~PN_STORE_BF16) == PN_PREC_BF16 && !gram->b.s2, "pn_panel_finalize: the riding Gram job is a bf16, single-source job");
    PN_TRY(wgrad_batch_one(*gram, 128, 128, wb, wblocks));
    const int n_fin = (C / 32) * slices;
    hipLaunchKernelGGL(panel_finalize_gram_kernel, dim3(n_fin + wblocks), dim3(256), 0, st, a, C / 32, n_fin, wb);
    PN_CHECK_LAUNCH();
    return PN_OK;
  }
  hipLaunchKernelGGL(panel_finalize_kernel, dim3(C / 32, slices), dim3(256), 0, st, a);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

}  // namespace pn
