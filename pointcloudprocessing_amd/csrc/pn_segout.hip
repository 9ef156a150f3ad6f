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

// logits of one point -> softmax, keras SparseCategoricalCrossentropy (probabilities clipped to [1e-7, 1 - 1e-7]), accuracy and
// d(loss)/d(logits); shared by the stand-alone output kernel and the fused frozen head below
__device__ __forceinline__ void seg_row_probs(const float (&acc)[SEG_CM], int C, float (&p)[SEG_CM], int& am) {
    float mx = -INFINITY;
    am = 0;
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c)
      if (c < C && acc[c] > mx) { mx = acc[c]; am = c; }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c) {
      p[c] = c < C ? expf(acc[c] - mx) : 0.f;
      sum += p[c];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c) p[c] *= inv;
}
// loss and accuracy of one point from its probabilities; qs = sum of the clipped probabilities (the gradient needs it)
__device__ __forceinline__ void seg_row_loss(const float (&p)[SEG_CM], int C, int y, int am, float& loss, float& corr, float& qs) {
    float py = 1.f;
    qs = 0.f;
#pragma unroll
    for (int c = 0; c < SEG_CM; ++c)
      if (c < C) {
        const float pc = clip_nan(p[c], 1e-7f, 1.f - 1e-7f);
        qs += pc;
        if (c == y) py = pc;
      }
    loss = -(logf(py) - logf(qs));
    corr = (am == y) ? 1.f : 0.f;
}
__device__ __forceinline__ void seg_row_tail(const float (&acc)[SEG_CM], int C, long long row, const int* __restrict__ labels, float grad_scale,
                                             float* __restrict__ probs, float* __restrict__ dlogits, float& loss, float& corr,
                                             float (&dl)[SEG_CM]) {
    float p[SEG_CM];
    int am;
    seg_row_probs(acc, C, p, am);
    if (probs) {
      float* po = probs + row * C;
#pragma unroll
      for (int c = 0; c < SEG_CM; ++c)
        if (c < C) po[c] = p[c];
    }
    if (labels) {
      const int y = labels[row];
      float qs;
      seg_row_loss(p, C, y, am, loss, corr, qs);
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
    seg_row_tail(acc, C, row, labels, grad_scale, probs, dlogits, loss, corr, dl);
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

// ---- the whole segmentation head in ONE launch, for a head whose BatchNormalization layers use their moving statistics -------------
// (inference, and the reference's `classification_pretrain` profile where the head is frozen and only its loss value and accuracy
// are reported, f15_lidar_config.json:43-69).  Without batch statistics nothing couples the points between layers, so a workgroup
// takes a 64-row tile through seg_l1 (64 -> 512, + the per-cloud global-feature bias), seg_l2 (512 -> 256), seg_l3 (256 -> 128),
// seg_l4 (128 -> 128) and the output layer + softmax + loss without the 512- / 256- / 128-wide tensors ever leaving the CU:
//   * activations live in LDS as bf16 MFMA operands (seg_l1's output in four 128-channel chunks, each consumed at once as a K-chunk of
//     seg_l2, whose 64 x 256 accumulators stay in registers); the kernels come straight from global memory as B fragments out of the
//     transposed bf16 copies the step's first launch makes (one 16-byte load per lane and k-step, each wave its own columns);
//   * every value is formed exactly as the layer-by-layer plan forms it (same k order per accumulator, bias before the store rounding,
//     BatchNormalization + ReLU applied to the stored value, hi + lo operands for the output layer), so the two plans agree bit for
//     bit -- tests/test_gpu_model.py compares them.
// Five launches and ~67 MB of stored activations (C2) become one launch and none.
constexpr int SH_ROWS = 64;                                               // rows per loss / accuracy partial (and per tile of the 64-row form)
constexpr int SH_P64 = 64 + 8, SH_P128 = 128 + 8, SH_P256 = 256 + 8;     // LDS pitches (bf16 elements): conflict-free 16-byte rows
constexpr int SH_PF = 128 + 4;                                            // fp32 pitch of the output layer's input
// A workgroup takes MB 32-row blocks: 64 rows (two workgroups per CU; the default), or 128 rows (one per CU; PN_SEGHEAD_MB=4) -- every
// workgroup streams ALL of the head's kernels (424 KB of bf16 fragments) from L2 whatever its height (seg_head_fused has the timings).
constexpr int sh_region_a(int rows) { return (rows * SH_P64 + rows * SH_P128) * 2; }   // input tile + seg_l1 chunk; later seg_l3's output, then the
                                                                          // output layer's kernel image and logit tile
constexpr int sh_region_b(int rows) { return rows * SH_P256 * 2; }                      // seg_l2's output; later the output layer's fp32 input
static_assert(64 * SH_PF * 4 <= sh_region_b(64) && 128 * SH_PF * 4 <= sh_region_b(128), "region B holds the fp32 input of the output layer");
static_assert(64 * (SEG_CM + 1) * 4 <= sh_region_a(64) && 128 * (SEG_CM + 1) * 4 <= sh_region_a(128), "region A holds the logit tile");
struct SegHeadArgs {
  pn_operand x;                                     // (B*N, 64) lazy operand: X_64 (or relu(bn(mlp_1_2)) for the vanilla model)
  const float* gb;                                  // (B, 512) global-feature half of seg_l1, per cloud
  const unsigned short *w1t, *w2t, *w3t, *w4t;      // bf16 kernels as MFMA fragments (WCopyDesc.frag = 1)
  const float *sc1, *sh1, *sc2, *sh2, *sc3, *sh3, *sc4, *sh4;   // BatchNormalization scale / shift (moving statistics)
  const float *w5, *b5;                             // output layer (128, C) fp32, bias (C)
  int N, C, tiles_per_cloud, s16;                   // s16: the layer-by-layer plan stores z as bf16 -- round the same way
  const int* labels; float grad_scale; float *probs, *dlogits, *part;
};
// One 32 x 32 accumulator block (lane = column c0 + r, 16 rows) -> bf16 LDS image.  Neighbouring lanes swap half of their values
// (one DPP move per row pair) so that every lane stores PAIRS of adjacent columns as one dword: 8 ds_write_b32 per lane instead of
// 16 ds_write_b16 -- sub-dword LDS stores from 64 lanes serialise, and the epilogues of this kernel store 256 values per lane.
__device__ __forceinline__ void sh_store_block(__bf16* img, int pitch, int m, int col, int h, int lane, const float (&y)[16]) {
  const bool odd = lane & 1;
#pragma unroll
  for (int e = 0; e < 16; e += 2) {
    const float give = odd ? y[e] : y[e + 1];
    // the neighbour's value: v_mov_b32_dpp quad_perm:[1,0,3,2] (no LDS crossbar traffic)
    const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));
    const int ee = odd ? e + 1 : e;                           // the row this lane writes of the pair
    const int row = m * 32 + (ee & 3) + 8 * (ee >> 2) + 4 * h;
    const float lo = odd ? got : y[e], hi = odd ? y[e + 1] : got;     // columns (col & ~1, col | 1)
    const unsigned w = (unsigned)f32_bf16_bits(lo) | ((unsigned)f32_bf16_bits(hi) << 16);
    *reinterpret_cast<unsigned*>(img + row * pitch + (col & ~1)) = w;
  }
}
__device__ __forceinline__ float sh_bnrelu(float z, float sc, float sh, int s16) {
  const float zq = s16 ? bf16_bits_f32(f32_bf16_bits(z)) : z;
  return clamp_lo(fmaf(sc, zq, sh), 0.f);
}
// Round 3: the accumulators are formed TRANSPOSED -- the kernel fragment is the MFMA's A operand, the activation fragment its B operand
// (same data, same k order: the same sums) -- so a lane holds ONE POINT's 16 channels c0 + 4h + 8g + {0..3}, g = 0..3, instead of one
// channel's 16 points.  Adjacent channels are adjacent registers: a group of four is rounded by two v_cvt_pk_bf16_f32 and stored with one
// 8-byte LDS write, with no cross-lane exchange.  The epilogues were 17 vector instructions per accumulator element (selects on the
// run-time storage flag, one DPP swap and three selects per pair, two scalar conversions) against 416 MFMAs per wave: the launch was
// bound by them (47 us at C2 for 6 us of matrix-core time).  Per-channel constants are now per REGISTER: loaded as float4 groups.
struct sh_coef { float4 v[4]; };                                       // one constant for each of the lane's 16 channels
__device__ __forceinline__ sh_coef sh_load_coef(const float* __restrict__ p, int c0, int h) {
  sh_coef c;
#pragma unroll
  for (int g = 0; g < 4; ++g) c.v[g] = *reinterpret_cast<const float4*>(p + c0 + 8 * g + 4 * h);
  return c;
}
__device__ __forceinline__ float sh_c(const sh_coef& c, int g, int q) { return q == 0 ? c.v[g].x : (q == 1 ? c.v[g].y : (q == 2 ? c.v[g].z : c.v[g].w)); }
typedef __attribute__((ext_vector_type(2))) float seg_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 seg_bf16x2;
// BN + ReLU of the lane's 16 channels (the stored-then-reloaded form: S16 rounds the pre-BN value to bf16 first, as the layer-by-layer
// plan's 16-bit store does), fp32 results
template <bool S16, bool BIAS>
__device__ __forceinline__ void sh_bnrelu16(const seg_f32x16& acc, const sh_coef& bias, const sh_coef& sc, const sh_coef& sh, float (&y)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
      float z0 = acc[4 * g + q], z1 = acc[4 * g + q + 1];
      if (BIAS) { z0 += sh_c(bias, g, q); z1 += sh_c(bias, g, q + 1); }
      if (S16) {
        const seg_f32x2 zz = {z0, z1};
        const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(zz, seg_bf16x2));
        z0 = __builtin_bit_cast(float, u << 16);
        z1 = __builtin_bit_cast(float, u & 0xffff0000u);
      }
      y[4 * g + q] = clamp_lo(fmaf(sh_c(sc, g, q), z0, sh_c(sh, g, q)), 0.f);
      y[4 * g + q + 1] = clamp_lo(fmaf(sh_c(sc, g, q + 1), z1, sh_c(sh, g, q + 1)), 0.f);
    }
}
// ... -> bf16 -> the point's row of an LDS image (operand of the next layer)
__device__ __forceinline__ void sh_store16(__bf16* img, int pitch, int row, int c0, int h, const float (&y)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const seg_f32x2 a = {y[4 * g], y[4 * g + 1]}, b = {y[4 * g + 2], y[4 * g + 3]};
    uint2 w;
    w.x = __builtin_bit_cast(unsigned, __builtin_convertvector(a, seg_bf16x2));
    w.y = __builtin_bit_cast(unsigned, __builtin_convertvector(b, seg_bf16x2));
    *reinterpret_cast<uint2*>(img + row * pitch + c0 + 8 * g + 4 * h) = w;
  }
}
// Kernel fragments are REQUESTED a phase ahead of their use; a compiler-level memory fence (no instruction) after each group of
// requests keeps the compiler from sinking the loads back to their first use -- a workgroup walks ~80 dependent k-steps, and with
// the fragment of each step requested at the step the kernel ran at one global-memory round trip per step.  (An empty asm that
// NAMES the loaded registers would do the opposite: it is a use, so the wait for the load lands right there.)
#define SH_KEEP_ABOVE() asm volatile("" ::: "memory")
// Workgroup barrier for LDS hand-offs only: this wave's LDS traffic is drained, the global loads in flight are NOT (__syncthreads
// waits for vmcnt(0) too, which would land every prefetched fragment at the next barrier: one memory round trip per barrier)
#define SH_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// DBG (PN_SEGHEAD_DBG=16, tools/seghead_stamps.py): the product kernel + s_memtime stamps of wave 0 at its phase boundaries, left over
// the tile's first 24 output probabilities
#define SH_STAMP(i) do { if (DBG) stamp[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
// Per-channel constants of the whole head, staged once per workgroup (10 KB of LDS): the epilogues read them at LDS latency instead of
// waiting a memory round trip behind the fragment prefetches (measured with the stamps: seg_l2's epilogue 8.9 k cycles for 8 blocks
// against seg_l1's 2.2 k for 4 -- its 64 coefficients were requested where they were used)
constexpr int SH_CO_GB = 0, SH_CO_SC1 = 512, SH_CO_SH1 = 1024, SH_CO_SC2 = 1536, SH_CO_SH2 = 1792, SH_CO_SC3 = 2048, SH_CO_SH3 = 2176, SH_CO_SC4 = 2304,
              SH_CO_SH4 = 2432, SH_CO_B5 = 2560, SH_CO_FLOATS = 2560 + SEG_CM;
// the output layer's kernel image (hi, lo: SEG_CM class rows each), built in the prologue.  The logit MFMAs read 32 "class" rows: rows
// SEG_CM..31 of the lo image lie in the coefficient table behind it -- finite or not, those columns of the product are never stored
constexpr int sh_region_c(int) { return 2 * SEG_CM * SEG_WP * 2; }
static_assert((32 - SEG_CM) * SEG_WP * 2 <= SH_CO_FLOATS * 4, "the unused class rows of the lo image stay inside the allocation");
constexpr int sh_region_d() { return SH_CO_FLOATS * 4; }
constexpr int sh_lds_bytes(int rows) { return sh_region_a(rows) + sh_region_b(rows) + sh_region_c(rows) + sh_region_d(); }
static_assert(sh_lds_bytes(128) <= 160 * 1024 && 2 * sh_lds_bytes(64) <= 160 * 1024, "one 128-row workgroup or two 64-row workgroups per CU");
template <int MB, bool S16, bool DBG = false>
__global__ __launch_bounds__(256, MB == 2 ? 2 : 1) void seg_head_fused_kernel(const SegHeadArgs a) {
  constexpr int ROWS = 32 * MB, SH_REGION_A = sh_region_a(ROWS), SH_REGION_B = sh_region_b(ROWS), SH_REGION_C = sh_region_c(ROWS);
  unsigned stamp[DBG ? 24 : 1];
  SH_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) unsigned char sh_sm[];
  __bf16* Ain = reinterpret_cast<__bf16*>(sh_sm);
  __bf16* S1c = Ain + ROWS * SH_P64;
  __bf16* S3 = reinterpret_cast<__bf16*>(sh_sm);
  __bf16* S2 = reinterpret_cast<__bf16*>(sh_sm + SH_REGION_A);
  float* S4f = reinterpret_cast<float*>(sh_sm + SH_REGION_A);
  float* prob_s = reinterpret_cast<float*>(sh_sm + SH_REGION_A);                                     // after the logits: region B is dead
  __bf16* W5hi = reinterpret_cast<__bf16*>(sh_sm + SH_REGION_A + SH_REGION_B);
  __bf16* W5lo = W5hi + SEG_CM * SEG_WP;
  float* lgt = reinterpret_cast<float*>(sh_sm);                                                       // region A once seg_l3's image is dead
  float* coef = reinterpret_cast<float*>(sh_sm + SH_REGION_A + SH_REGION_B + SH_REGION_C);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int cloud = blockIdx.x / a.tiles_per_cloud, tin = blockIdx.x - cloud * a.tiles_per_cloud;
  const int r0 = tin * ROWS, nrows = min(ROWS, a.N - r0);
  const long long row0 = (long long)cloud * a.N + r0;

  auto afrag = [&](const __bf16* img, int pitch, int m, int k0) {
    return *reinterpret_cast<const seg_bf16x8*>(img + (m * 32 + r) * pitch + k0 + 8 * h);
  };
  // kernel fragments come FRAGMENT-MAJOR (pn_prologue.hip, WCopyDesc.frag): the 64 lanes of fragment (32-column block cb, k-step ks)
  // read 1 KB of consecutive bytes.  From the plain transposed copy [Cout][K] the same load touched 32 cache lines for 32 bytes each
  // (rows K * 2 bytes apart): 2048 line requests per CU and K-chunk of seg_l2, and the stamps showed that phase at 62 cycles per MFMA
  auto bfrag = [&](const unsigned short* wt, int K, int cb, int ks) {
    return __builtin_bit_cast(seg_bf16x8, *reinterpret_cast<const uint4*>(wt + (((long long)cb * (K >> 4) + ks) * 64 + lane) * 8));
  };

  // ---- everything that depends on nothing is requested first: one memory round trip for all of it ----
  // (a) the first kernel fragments of seg_l1 / seg_l2
  seg_bf16x8 b1[4], b2[8][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { b1[ks] = bfrag(a.w1t, 64, wave, ks); }
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int n = 0; n < 2; ++n) b2[ks][n] = bfrag(a.w2t, 512, 2 * wave + n, ks);
  // (b) the per-channel constants (640 float4 groups + the output bias)
  float4 cf[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int i = tid + 256 * q;                       // float4 index in the table
    const float* src = a.gb + (long long)cloud * 512;
    int off = i;
    if (i >= 128) { src = a.sc1; off = i - 128; }
    if (i >= 256) { src = a.sh1; off = i - 256; }
    if (i >= 384) { src = a.sc2; off = i - 384; }
    if (i >= 448) { src = a.sh2; off = i - 448; }
    if (i >= 512) { src = a.sc3; off = i - 512; }
    if (i >= 544) { src = a.sh3; off = i - 544; }
    if (i >= 576) { src = a.sc4; off = i - 576; }
    if (i >= 608) { src = a.sh4; off = i - 608; }
    cf[q] = i < 640 ? *reinterpret_cast<const float4*>(src + 4 * off) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float b5v = (tid < SEG_CM && tid < a.C && a.b5) ? a.b5[tid] : 0.f;
  // (c) this thread's label (the tail: waves 0 (, 1), one point per lane)
  const int prow = 64 * wave + lane;
  const int ylab = (a.labels && wave < MB / 2 && prow < nrows) ? a.labels[row0 + prow] : 0;
  // (d) the output layer's kernel: two adjacent k of one class per item (one 4-byte LDS store for each image)
  float w5v[SEG_CM * 128 / 256];
#pragma unroll
  for (int q = 0; q < SEG_CM * 64 / 256; ++q) {
    const int t = tid + 256 * q, c = t >> 6, k = 2 * (t & 63);
    w5v[2 * q] = c < a.C ? a.w5[(long long)k * a.C + c] : 0.f;
    w5v[2 * q + 1] = c < a.C ? a.w5[(long long)(k + 1) * a.C + c] : 0.f;
  }
  SH_KEEP_ABOVE();
  SH_STAMP(22);

  // ---- the tile's 64 input channels -> LDS (bf16 operand precision; rows past the cloud are zero rows) ----
#pragma unroll
  for (int pass = 0; pass < ROWS / 64; ++pass) {
    const int row = pass * 64 + (tid >> 2), c0 = (tid & 3) * 16;
    float v[16];
    const long long src = (row0 + min(row, nrows - 1)) * a.x.ld + c0;
    if (a.x.h16) {
      bf16x8_unpack(act_load8_raw(a.x.s1, src), *reinterpret_cast<float(*)[8]>(v));
      bf16x8_unpack(act_load8_raw(a.x.s1, src + 8), *reinterpret_cast<float(*)[8]>(v + 8));
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(a.x.s1 + src + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      }
    }
    seg_bf16x8 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float ca = a.x.ca ? a.x.ca[c0 + e] : 1.f, cc = a.x.cc ? a.x.cc[c0 + e] : 0.f;
      const float t = row < nrows ? clamp_lo(fmaf(ca, v[e], cc), a.x.lo) : 0.f;
      if (e < 8) o0[e] = (__bf16)t; else o1[e - 8] = (__bf16)t;
    }
    *reinterpret_cast<seg_bf16x8*>(Ain + row * SH_P64 + c0) = o0;
    *reinterpret_cast<seg_bf16x8*>(Ain + row * SH_P64 + c0 + 8) = o1;
  }
  SH_STAMP(23);
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (tid + 256 * q < 640) *reinterpret_cast<float4*>(coef + 4 * (tid + 256 * q)) = cf[q];
  if (tid < SEG_CM) coef[SH_CO_B5 + tid] = b5v;
#pragma unroll
  for (int q = 0; q < SEG_CM * 64 / 256; ++q) {
    const int t = tid + 256 * q, c = t >> 6, k = 2 * (t & 63);
    const __bf16 h0 = (__bf16)w5v[2 * q], h1 = (__bf16)w5v[2 * q + 1];
    const seg_bf16x2 hi = {h0, h1}, lo = {(__bf16)(w5v[2 * q] - (float)h0), (__bf16)(w5v[2 * q + 1] - (float)h1)};
    *reinterpret_cast<seg_bf16x2*>(W5hi + c * SEG_WP + k) = hi;
    *reinterpret_cast<seg_bf16x2*>(W5lo + c * SEG_WP + k) = lo;
  }
  SH_STAMP(1);
  SH_BARRIER();
  SH_STAMP(2);

  // ---- seg_l1 in four 128-channel chunks, each at once a K-chunk of seg_l2 ----
  seg_f32x16 acc2[MB][2];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[m][n][e] = 0.f;
#pragma unroll 1
  for (int j = 0; j < 4; ++j) {
    if (j == 0) SH_STAMP(3);
    if (j == 1) SH_STAMP(4);
    seg_f32x16 acc1[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[m][e] = 0.f;
    const int c1 = 128 * j + 32 * wave;                     // this wave's 32 seg_l1 channels of the chunk
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int m = 0; m < MB; ++m) acc1[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[ks], afrag(Ain, SH_P64, m, ks * 16), acc1[m], 0, 0, 0);
    }
    // next chunk's seg_l1 fragments: under this chunk's epilogue and seg_l2 steps.  Unconditional (the last chunk re-requests its
    // own): behind a branch the compiler must assume at the join that the requests were NOT made and waits for everything
    const int jn = j < 3 ? j + 1 : 3;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b1[ks] = bfrag(a.w1t, 64, 4 * jn + wave, ks);
    SH_KEEP_ABOVE();
    const sh_coef bias = sh_load_coef(coef + SH_CO_GB, c1, h), sc = sh_load_coef(coef + SH_CO_SC1, c1, h), sh = sh_load_coef(coef + SH_CO_SH1, c1, h);
    if (j == 1) SH_STAMP(5);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float y[16];
      sh_bnrelu16<S16, true>(acc1[m], bias, sc, sh, y);
      sh_store16(S1c, SH_P128, m * 32 + r, 32 * wave, h, y);
    }
    if (j == 1) SH_STAMP(6);
    SH_BARRIER();
    if (j == 1) SH_STAMP(7);
    // seg_l2 over this K-chunk: the activation fragments of step ks + 1 are read while the matrix cores run step ks
    seg_bf16x8 af[2][MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) af[0][m] = afrag(S1c, SH_P128, m, 0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks < 7) {
#pragma unroll
        for (int m = 0; m < MB; ++m) af[(ks + 1) & 1][m] = afrag(S1c, SH_P128, m, (ks + 1) * 16);
      }
#pragma unroll
      for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int n = 0; n < 2; ++n) acc2[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2[ks][n], af[ks & 1][m], acc2[m][n], 0, 0, 0);
      }
      // the registers of this step go straight to the same step of the next chunk (unconditional, as above)
#pragma unroll
      for (int n = 0; n < 2; ++n) b2[ks][n] = bfrag(a.w2t, 512, 2 * wave + n, 8 * jn + ks);
      SH_KEEP_ABOVE();
    }
    if (j == 1) SH_STAMP(8);
    SH_BARRIER();                                        // the chunk image is overwritten by the next chunk
    if (j == 1) SH_STAMP(9);
  }
  SH_STAMP(10);
  // seg_l3's first eight fragments and all of seg_l4's: requested now, land under seg_l2's epilogue
  seg_bf16x8 b3[8], b4[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) { b3[ks] = bfrag(a.w3t, 256, wave, ks); b4[ks] = bfrag(a.w4t, 128, wave, ks); }
  SH_KEEP_ABOVE();
  // ---- seg_l2 -> LDS ----
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const sh_coef sc2v = sh_load_coef(coef + SH_CO_SC2, 64 * wave + 32 * n, h), sh2v = sh_load_coef(coef + SH_CO_SH2, 64 * wave + 32 * n, h);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float y[16];
      sh_bnrelu16<S16, false>(acc2[m][n], sc2v, sc2v, sh2v, y);
      sh_store16(S2, SH_P256, m * 32 + r, 64 * wave + 32 * n, h, y);
    }
  }
  SH_STAMP(11);
  SH_BARRIER();
  SH_STAMP(12);
  // ---- seg_l3 (256 -> 128): 32 columns per wave ----
  {
    const sh_coef sc3v = sh_load_coef(coef + SH_CO_SC3, 32 * wave, h), sh3v = sh_load_coef(coef + SH_CO_SH3, 32 * wave, h);
    seg_f32x16 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b3[ks & 7], afrag(S2, SH_P256, m, ks * 16), acc[m], 0, 0, 0);
      if (ks < 8) {                                         // steps 8..15 take over the registers of steps 0..7
        b3[ks] = bfrag(a.w3t, 256, wave, ks + 8);
        SH_KEEP_ABOVE();
      }
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {                             // region A: the input tile and chunk image are dead
      float y[16];
      sh_bnrelu16<S16, false>(acc[m], sc3v, sc3v, sh3v, y);
      sh_store16(S3, SH_P128, m * 32 + r, 32 * wave, h, y);
    }
  }
  SH_STAMP(13);
  SH_BARRIER();
  SH_STAMP(14);
  // ---- seg_l4 (128 -> 128) -> the output layer's input, fp32 (it is split hi + lo there, as in seg_out_fwd) ----
  {
    const sh_coef sc4v = sh_load_coef(coef + SH_CO_SC4, 32 * wave, h), sh4v = sh_load_coef(coef + SH_CO_SH4, 32 * wave, h);
    seg_f32x16 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b4[ks], afrag(S3, SH_P128, m, ks * 16), acc[m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float y[16];
      sh_bnrelu16<S16, false>(acc[m], sc4v, sc4v, sh4v, y);
#pragma unroll
      for (int g = 0; g < 4; ++g)                                                    // region B: seg_l2's image is dead
        *reinterpret_cast<float4*>(S4f + (m * 32 + r) * SH_PF + 32 * wave + 8 * g + 4 * h) = make_float4(y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3]);
    }
  }
  SH_STAMP(15);
  SH_BARRIER();                                          // seg_l3's image (region A) is dead from here
  SH_STAMP(16);
  // ---- output layer: logits of a 32-row block per wave (waves 0 .. MB-1) from the kernel image the prologue built ----
  SH_STAMP(17);
  SH_STAMP(18);
  if (wave < MB) {
    seg_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k0 = ks * 16 + 8 * h;
      const float4 v0 = *reinterpret_cast<const float4*>(S4f + (wave * 32 + r) * SH_PF + k0);
      const float4 v1 = *reinterpret_cast<const float4*>(S4f + (wave * 32 + r) * SH_PF + k0 + 4);
      const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      seg_bf16x8 ah, al;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ah[e] = (__bf16)v[e];
        al[e] = (__bf16)(v[e] - (float)ah[e]);
      }
      const seg_bf16x8 bh = *reinterpret_cast<const seg_bf16x8*>(W5hi + r * SEG_WP + k0);
      const seg_bf16x8 bl = *reinterpret_cast<const seg_bf16x8*>(W5lo + r * SEG_WP + k0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
    if (r < SEG_CM) {
#pragma unroll
      for (int e = 0; e < 16; ++e) lgt[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * (SEG_CM + 1) + r] = acc[e];
    }
  }
  SH_STAMP(19);
  SH_BARRIER();
  SH_STAMP(20);
  // ---- softmax, loss, accuracy: one thread per point (waves 0 (, 1)); no gradient leaves this kernel (frozen head), so none is formed.
  // The probabilities go to LDS in the layout of the output and leave as whole 16-byte stores from all threads (one thread's twelve
  // 4-byte stores 48 bytes apart were, with the gradient arithmetic and its 12 wave reductions, 27 % of the launch: 25.8 k cycles) ----
  if (wave < MB / 2) {                                   // rows 64 * wave + lane
    float loss = 0.f, corr = 0.f;
    if (prow < nrows) {
      float acc[SEG_CM], p[SEG_CM];
#pragma unroll
      for (int c = 0; c < SEG_CM; ++c) {
        acc[c] = (c < a.C && a.b5) ? coef[SH_CO_B5 + c] : 0.f;
        acc[c] += lgt[prow * (SEG_CM + 1) + c];
      }
      int am;
      seg_row_probs(acc, a.C, p, am);
#pragma unroll
      for (int c = 0; c < SEG_CM; ++c)
        if (c < a.C) prob_s[prow * a.C + c] = p[c];
      if (a.labels) {
        float qs;
        seg_row_loss(p, a.C, ylab, am, loss, corr, qs);
      }
    }
    if (a.part && r0 + 64 * wave < a.N) {             // one partial per 64 rows of a cloud, whatever the tile height
      float* pp = a.part + ((long long)cloud * ((a.N + 63) / 64) + (r0 >> 6) + wave) * (2 + SEG_CM);
      const float l = wave_sum(loss), cr = wave_sum(corr);
      if (lane == 0) { pp[0] = l; pp[1] = cr; }
    }
  }
  if (a.probs) {
    SH_BARRIER();
    const int n = nrows * a.C;
    float* dst = a.probs + row0 * a.C;                  // row0 is a multiple of 64: 16-byte aligned
    for (int i = 4 * tid; i + 3 < n; i += 1024) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(prob_s + i);
    if (tid < (n & 3)) dst[(n & ~3) + tid] = prob_s[(n & ~3) + tid];
  }
  if (DBG) {
    SH_STAMP(21);
    __syncthreads();
    if (tid == 0 && a.probs && nrows * a.C >= 24) {                     // over the tile's first probabilities
      unsigned* p = reinterpret_cast<unsigned*>(a.probs + row0 * a.C);
#pragma unroll
      for (int i = 0; i < 24; ++i) p[i] = stamp[i];
    }
  }
}
#undef SH_STAMP
#undef SH_KEEP_ABOVE
#undef SH_BARRIER
int seg_head_fused_rows() { return SH_ROWS; }
int seg_head_fused(const pn_operand* x, const float* gb, const void* w1t, const void* w2t, const void* w3t, const void* w4t, const float* sc1,
                   const float* sh1, const float* sc2, const float* sh2, const float* sc3, const float* sh3, const float* sc4, const float* sh4,
                   const float* w5, const float* b5, int B, int N, int C, int s16, const int* labels, float grad_scale, float* probs,
                   float* dlogits, float* part, hipStream_t st) {
  PN_CHECK_ARG(x && x->s1 && !x->s2 && gb && w1t && w2t && w3t && w4t && sc1 && sh1 && sc2 && sh2 && sc3 && sh3 && sc4 && sh4 && w5,
               "seg_head_fused: null pointer");
  PN_CHECK_ARG(B > 0 && N > 0 && C >= 1 && C <= SEG_CM, "seg_head_fused: bad sizes (B=%d N=%d C=%d)", B, N, C);
  PN_CHECK_ARG(x->ld >= 64 && x->ld % 8 == 0 && (reinterpret_cast<uintptr_t>(x->s1) & 15) == 0, "seg_head_fused: operand alignment");
  SegHeadArgs a;
  memset(&a, 0, sizeof(a));
  a.x = *x; a.gb = gb;
  a.w1t = reinterpret_cast<const unsigned short*>(w1t); a.w2t = reinterpret_cast<const unsigned short*>(w2t);
  a.w3t = reinterpret_cast<const unsigned short*>(w3t); a.w4t = reinterpret_cast<const unsigned short*>(w4t);
  a.sc1 = sc1; a.sh1 = sh1; a.sc2 = sc2; a.sh2 = sh2; a.sc3 = sc3; a.sh3 = sh3; a.sc4 = sc4; a.sh4 = sh4;
  a.w5 = w5; a.b5 = b5; a.N = N; a.C = C; a.s16 = s16;
  a.labels = labels; a.grad_scale = grad_scale; a.probs = probs; a.dlogits = dlogits; a.part = part;
  static const int force_mb = getenv("PN_SEGHEAD_MB") ? atoi(getenv("PN_SEGHEAD_MB")) : 0;     // experiment switch: 2 or 4
  static const bool dbg = getenv("PN_SEGHEAD_DBG") && atoi(getenv("PN_SEGHEAD_DBG")) == 16;
  // Two 64-row workgroups per CU (8 waves, out of phase with each other) beat one 128-row workgroup since the kernels arrive as whole
  // fragments: 27.4 vs 30.9 us at B*N = 32,768, 93 vs 106 us at 131,072 (round 2, [C][K] copies: 52 vs 48 us -- the 64-row form streams
  // the 424 KB of kernels twice as often and was bound by the 32-line gathers of every fragment load)
  const bool tall = force_mb == 4;
  if (tall) {
    static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_head_fused_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        sh_lds_bytes(128));
    static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_head_fused_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        sh_lds_bytes(128));
    PN_CHECK_ARG(attr1 == hipSuccess && attr0 == hipSuccess, "seg_head_fused: dynamic LDS attribute");
    a.tiles_per_cloud = cdiv(N, 128);
    if (dbg && s16 && N % 128 == 0) {
      static const hipError_t attrd = hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_head_fused_kernel<4, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                          sh_lds_bytes(128));
      PN_CHECK_ARG(attrd == hipSuccess, "seg_head_fused: dynamic LDS attribute");
      hipLaunchKernelGGL((seg_head_fused_kernel<4, true, true>), dim3(B * a.tiles_per_cloud), dim3(256), sh_lds_bytes(128), st, a);
    } else
    if (s16) hipLaunchKernelGGL((seg_head_fused_kernel<4, true>), dim3(B * a.tiles_per_cloud), dim3(256), sh_lds_bytes(128), st, a);
    else hipLaunchKernelGGL((seg_head_fused_kernel<4, false>), dim3(B * a.tiles_per_cloud), dim3(256), sh_lds_bytes(128), st, a);
  } else {
    a.tiles_per_cloud = cdiv(N, 64);
    static const hipError_t attr3 = hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_head_fused_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        sh_lds_bytes(64));
    static const hipError_t attr2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_head_fused_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        sh_lds_bytes(64));
    PN_CHECK_ARG(attr3 == hipSuccess && attr2 == hipSuccess, "seg_head_fused: dynamic LDS attribute");
    if (dbg && s16 && N % 64 == 0) {
      static const hipError_t attrd = hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_head_fused_kernel<2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                          sh_lds_bytes(64));
      PN_CHECK_ARG(attrd == hipSuccess, "seg_head_fused: dynamic LDS attribute");
      hipLaunchKernelGGL((seg_head_fused_kernel<2, true, true>), dim3(B * a.tiles_per_cloud), dim3(256), sh_lds_bytes(64), st, a);
    } else
    if (s16) hipLaunchKernelGGL((seg_head_fused_kernel<2, true>), dim3(B * a.tiles_per_cloud), dim3(256), sh_lds_bytes(64), st, a);
    else hipLaunchKernelGGL((seg_head_fused_kernel<2, false>), dim3(B * a.tiles_per_cloud), dim3(256), sh_lds_bytes(64), st, a);
  }
  PN_CHECK_LAUNCH();
  return PN_OK;
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
