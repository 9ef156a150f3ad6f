// The device core of the contraction engine (pn_gemm.hip has the description): operand stagers, the MFMA chunk, the weight-gradient
// tile and its batch walker -- in a header since round 3 so that OTHER launches can carry weight-gradient workgroups behind their own
// (pn_panel.hip: the Gram matrices of the max-pooled layers ride behind the panel finaliser's 128 workgroups).
#pragma once
#include "pn_common.h"
#include "pn_internal.h"

namespace pn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { MODE_FWD = 0, MODE_BWD = 1, MODE_WGRAD = 2 };
enum { EPI_STORE = 0, EPI_MAX = 1, EPI_SLAB = 2 };

struct GemmArgs {
  pn_operand a;       // FWD/BWD: activation-side operand.  WGRAD: operand a (channels Ci)
  pn_operand b;       // WGRAD only: operand b (channels Cj)
  const float* w;     // FWD: W[K][C];  BWD: W[C][K]
  const unsigned short* w16;   // optional (bf16 operands, shared kernel): a bf16 copy laid out [C][K], k contiguous -- FWD: the transposed
                               // kernel, BWD: the kernel as it is -- staged without conversion instead of `w`
  long long w_cloud_stride;
  int B, N;           // clouds, points per cloud
  int K;              // contraction length (FWD/BWD);   WGRAD: slab_rows
  int C;              // output channels (FWD/BWD);      WGRAD: Cj
  int Ci;             // WGRAD: Ci
  int tiles_per_cloud;  // FWD/BWD: ceil(N/BM);  WGRAD: slabs per cloud
  // epilogue
  float* out;               // STORE: (B*N, C);  SLAB: slabs
  const float* cloud_bias;  // STORE (optional) (B, C)
  long long cloud_bias_stride;   // elements between two clouds' rows of cloud_bias (C; 0: one row for every cloud)
  const float* addend;      // STORE (optional) (B*N, C)
  const float* zmask;       // STORE (optional) relu mask source (B*N, C)
  const float* msc;         // mask scale/shift per channel
  const float* msh;
  float* stat_partials;     // [tiles][2][C] (optional)
  const float* sgn;         // MAX
  float* pmax;              // MAX [tiles][C]
  int* pidx;                // MAX [tiles][C]
  int store16;              // STORE: out, addend and zmask are bf16 arrays (PN_STORE_BF16)
  int ncol;                 // FWD/BWD: column tiles per row tile (the grid is 1-D: row tiles x ncol, see gemm_kernel)
  int colsum;               // WGRAD: also emit sum_rows a[row][i] as an extra row after each slab (slab stride Ci*C + Ci)
  int dbg;                  // PN_GEMM_DBG ablations (tools/gemm_probe.py): 1 no output stores, 2 no statistics, 4 no A loads, 8 no W loads
};

template <int BK>
struct Geo {
  static constexpr int PITCH = BK + 8;  // bf16 elements
};

__device__ __forceinline__ void cvt_store8(__bf16* hi, __bf16* lo, const float (&v)[8], bool split) {
  bf16x8 h;
#pragma unroll
  for (int e = 0; e < 8; ++e) h[e] = (__bf16)v[e];
  *reinterpret_cast<bf16x8*>(hi) = h;
  if (split) {
    bf16x8 l;
#pragma unroll
    for (int e = 0; e < 8; ++e) l[e] = (__bf16)(v[e] - (float)h[e]);
    *reinterpret_cast<bf16x8*>(lo) = l;
  }
}

// Staging is split into an ISSUE phase (all global loads of both operands, unconditional, from clamped addresses)
// and a FINISH phase (affine + ReLU/mask select, bf16 split, ds_write_b128).  Between them every loaded register is
// pinned with an empty asm: without that LLVM sinks a load whose value is only used under the validity select back
// into a branch and waits vmcnt(0) per load, which serialises the tile (cdna_hip_programming.md section 5, trap 4c).

// ---- "natural" stager: source[(row)*ld + k], k contiguous; coefficients indexed by k ------------------
// H16: the sources are bf16 arrays -- eight consecutive k are ONE 16-byte load (compile-time: a run-time storage flag inside the
// unrolled loops costs registers and, worse, a branch per load)
template <int TR, int BK, bool HAS2, bool H16>
struct NatStage {
  static constexpr int CH = BK / 8;     // 16-byte bf16 chunks per LDS row
  static constexpr int RP = 256 / CH;   // rows per pass
  static constexpr int P = TR / RP;
  static constexpr int Q = H16 ? 1 : 2; // register quads per 8 elements
  float4 x[P][Q];
  float4 y[HAS2 ? P : 1][Q];

  __device__ __forceinline__ void issue(const pn_operand& op, long long base, int nvalid_rows, int k0, int tid) {
    const int ch = tid % CH, rin = tid / CH;
    const int k = k0 + ch * 8;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int r = p * RP + rin;
      const long long rr = (r < nvalid_rows) ? r : (nvalid_rows - 1);
      if constexpr (H16) {
        x[p][0] = __builtin_bit_cast(float4, act_load8_raw(op.s1, base + rr * op.ld + k));
        if (HAS2) y[p][0] = __builtin_bit_cast(float4, act_load8_raw(op.s2, base + rr * op.ld + k));
      } else {
        const float* s = op.s1 + base + rr * op.ld + k;
        x[p][0] = *reinterpret_cast<const float4*>(s);
        x[p][1] = *reinterpret_cast<const float4*>(s + 4);
        if (HAS2) {
          const float* s2 = op.s2 + base + rr * op.ld + k;
          y[p][0] = *reinterpret_cast<const float4*>(s2);
          y[p][1] = *reinterpret_cast<const float4*>(s2 + 4);
        }
      }
    }
  }
  __device__ __forceinline__ void pin() {
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        asm volatile("" : "+v"(x[p][q].x), "+v"(x[p][q].y), "+v"(x[p][q].z), "+v"(x[p][q].w));
        if (HAS2) asm volatile("" : "+v"(y[p][q].x), "+v"(y[p][q].y), "+v"(y[p][q].z), "+v"(y[p][q].w));
      }
  }
  template <int NS>
  __device__ __forceinline__ void finish(__bf16* __restrict__ Thi, __bf16* __restrict__ Tlo, const pn_operand& op,
                                         int nvalid_rows, int k0, int tid) {
    constexpr int PITCH = Geo<BK>::PITCH;
    const int ch = tid % CH, rin = tid / CH;
    const int k = k0 + ch * 8;
    float ca[8], cb[8], cc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ca[e] = 1.f; cb[e] = 0.f; cc[e] = 0.f; }
    if (op.ca) {
      const float4 t0 = *reinterpret_cast<const float4*>(op.ca + k), t1 = *reinterpret_cast<const float4*>(op.ca + k + 4);
      ca[0] = t0.x; ca[1] = t0.y; ca[2] = t0.z; ca[3] = t0.w; ca[4] = t1.x; ca[5] = t1.y; ca[6] = t1.z; ca[7] = t1.w;
    }
    if (HAS2 && op.cb) {
      const float4 t0 = *reinterpret_cast<const float4*>(op.cb + k), t1 = *reinterpret_cast<const float4*>(op.cb + k + 4);
      cb[0] = t0.x; cb[1] = t0.y; cb[2] = t0.z; cb[3] = t0.w; cb[4] = t1.x; cb[5] = t1.y; cb[6] = t1.z; cb[7] = t1.w;
    }
    if (op.cc) {
      const float4 t0 = *reinterpret_cast<const float4*>(op.cc + k), t1 = *reinterpret_cast<const float4*>(op.cc + k + 4);
      cc[0] = t0.x; cc[1] = t0.y; cc[2] = t0.z; cc[3] = t0.w; cc[4] = t1.x; cc[5] = t1.y; cc[6] = t1.z; cc[7] = t1.w;
    }
    const float lo = op.lo;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int r = p * RP + rin;
      const bool rv = r < nvalid_rows;
      float v[8], w[8];
      if constexpr (H16) {
        bf16x8_unpack(__builtin_bit_cast(uint4, x[p][0]), v);
        if (HAS2) bf16x8_unpack(__builtin_bit_cast(uint4, y[p][0]), w);
      } else {
        v[0] = x[p][0].x; v[1] = x[p][0].y; v[2] = x[p][0].z; v[3] = x[p][0].w;
        v[4] = x[p][Q - 1].x; v[5] = x[p][Q - 1].y; v[6] = x[p][Q - 1].z; v[7] = x[p][Q - 1].w;
        if (HAS2) {
          w[0] = y[p][0].x; w[1] = y[p][0].y; w[2] = y[p][0].z; w[3] = y[p][0].w;
          w[4] = y[p][Q - 1].x; w[5] = y[p][Q - 1].y; w[6] = y[p][Q - 1].z; w[7] = y[p][Q - 1].w;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = fmaf(ca[e], v[e], cc[e]);
        if (HAS2) t = fmaf(cb[e], w[e], t);
        v[e] = rv ? clamp_lo(t, lo) : 0.f;
      }
      cvt_store8(Thi + r * PITCH + ch * 8, Tlo + r * PITCH + ch * 8, v, NS == 3);
    }
  }
};

// ---- "transposed" stager: source[(k)*ld + r], r (= the LDS row / channel) contiguous in memory;
//      coefficients indexed by r.  Each lane owns one channel and gathers 8 consecutive k for it.
//      H16 (bf16 sources): a lane takes a PAIR of channels -- one 4-byte load per k -- so a pass covers twice the k-groups and half
//      the passes (and load instructions) are needed. -------
template <int TR, int BK, bool HAS2, bool H16>
struct TrnStage {
  static constexpr int KG = BK / 8;                  // k-groups per chunk
  static constexpr int TRL = H16 ? TR / 2 : TR;      // lanes across the channels
  static constexpr int TPG = 256 / TRL;              // k-groups covered per pass
  static constexpr int P = (KG + TPG - 1) / TPG;     // (KG < TPG: the upper lanes of the single pass idle)
  static_assert(KG % TPG == 0 || TPG % KG == 0, "tile geometry");
  float x[P][8];                                     // H16: raw pairs (even channel in the low half)
  float y[HAS2 ? P : 1][8];

  __device__ __forceinline__ void issue(const pn_operand& op, long long base, int nvalid_k, int nvalid_r, int tid) {
    const int rl = tid % TRL, kgin = tid / TRL;
    if constexpr (H16) {
      const int rc = (2 * rl < nvalid_r) ? 2 * rl : ((nvalid_r - 1) & ~1);     // nvalid_r is even (channel counts are multiples of 64)
      const unsigned* p1 = reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(op.s1) + base + rc);
      const unsigned* p2 = HAS2 ? reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(op.s2) + base + rc) : nullptr;
      const long long ld2 = op.ld / 2;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int kg = p * TPG + kgin;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = kg * 8 + e;
          const long long kk = (k < nvalid_k) ? k : (nvalid_k - 1);
          x[p][e] = __builtin_bit_cast(float, p1[kk * ld2]);
          if (HAS2) y[p][e] = __builtin_bit_cast(float, p2[kk * ld2]);
        }
      }
    } else {
      const int rc = (rl < nvalid_r) ? rl : (nvalid_r - 1);
      const float* p1 = op.s1 + base + rc;
      const float* p2 = HAS2 ? (op.s2 + base + rc) : nullptr;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int kg = p * TPG + kgin;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = kg * 8 + e;
          const long long kk = (k < nvalid_k) ? k : (nvalid_k - 1);
          x[p][e] = p1[kk * op.ld];
          if (HAS2) y[p][e] = p2[kk * op.ld];
        }
      }
    }
  }
  __device__ __forceinline__ void pin() {
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        asm volatile("" : "+v"(x[p][e]));
        if (HAS2) asm volatile("" : "+v"(y[p][e]));
      }
  }
  // returns (when want_sum, block-uniform) the sum of the values this lane staged: .x its channel (H16: the even channel of its pair,
  // .y the odd one), the rows it gathered
  template <int NS>
  __device__ __forceinline__ float2 finish(__bf16* __restrict__ Thi, __bf16* __restrict__ Tlo, const pn_operand& op,
                                           int nvalid_k, int nvalid_r, int coef0, int tid, bool want_sum = false) {
    constexpr int PITCH = Geo<BK>::PITCH;
    constexpr int NH = H16 ? 2 : 1;
    float csum[2] = {0.f, 0.f};
    const int rl = tid % TRL, kgin = tid / TRL;
    const float lo = op.lo;
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {                        // H16: the even, then the odd channel of this lane's pair
      const int r = H16 ? 2 * rl + hf : rl;
      const bool rv = r < nvalid_r;
      const int rc = rv ? r : (nvalid_r - 1);
      float ca = 1.f, cb = 0.f, cc = 0.f;
      if (op.ca) ca = op.ca[coef0 + rc];
      if (HAS2 && op.cb) cb = op.cb[coef0 + rc];
      if (op.cc) cc = op.cc[coef0 + rc];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int kg = p * TPG + kgin;
        if (KG < TPG && kg >= KG) continue;                  // lanes beyond the chunk (single pass wider than the chunk)
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = kg * 8 + e;
          float xs, ys = 0.f;
          if constexpr (H16) {
            const unsigned ux = __builtin_bit_cast(unsigned, x[p][e]);
            xs = hf ? __builtin_bit_cast(float, ux & 0xffff0000u) : __builtin_bit_cast(float, ux << 16);
            if (HAS2) {
              const unsigned uy = __builtin_bit_cast(unsigned, y[p][e]);
              ys = hf ? __builtin_bit_cast(float, uy & 0xffff0000u) : __builtin_bit_cast(float, uy << 16);
            }
          } else {
            xs = x[p][e];
            if (HAS2) ys = y[p][e];
          }
          float t = fmaf(ca, xs, cc);
          if (HAS2) t = fmaf(cb, ys, t);
          v[e] = (rv && k < nvalid_k) ? clamp_lo(t, lo) : 0.f;
        }
        if (want_sum) csum[hf] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        cvt_store8(Thi + r * PITCH + kg * 8, Tlo + r * PITCH + kg * 8, v, NS == 3);
      }
    }
    return make_float2(csum[0], csum[1]);
  }
};

// ---- copy stager: rows of a bf16 matrix with k contiguous that is ALREADY in operand precision (the bf16 copies of a layer's kernel
//      the step's first launch makes, pn_prologue.hip): global -> LDS as is, no conversion -----------------------------------------
template <int TR, int BK>
struct CopyStage {
  static constexpr int CH = BK / 8;     // 16-byte chunks per LDS row
  static constexpr int RP = 256 / CH;   // rows per pass
  static constexpr int P = TR / RP;
  uint4 x[P];
  __device__ __forceinline__ void issue(const unsigned short* __restrict__ w, long long base, long long ld, int nvalid_rows, int k0, int tid) {
    const int ch = tid % CH, rin = tid / CH;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int r = p * RP + rin;
      const long long rr = (r < nvalid_rows) ? r : (nvalid_rows - 1);
      x[p] = *reinterpret_cast<const uint4*>(w + base + rr * ld + k0 + ch * 8);
    }
  }
  __device__ __forceinline__ void pin() {
#pragma unroll
    for (int p = 0; p < P; ++p) asm volatile("" : "+v"(x[p].x), "+v"(x[p].y), "+v"(x[p].z), "+v"(x[p].w));
  }
  __device__ __forceinline__ void finish(__bf16* __restrict__ Thi, int nvalid_rows, int tid) {
    constexpr int PITCH = Geo<BK>::PITCH;
    const int ch = tid % CH, rin = tid / CH;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int r = p * RP + rin;
      const uint4 v = (r < nvalid_rows) ? x[p] : make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(Thi + r * PITCH + ch * 8) = v;
    }
  }
};

// ---- MFMA over one staged chunk ------------------------------------------------------------------------
template <int MT, int NT, int BK, int NS>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[MT][NT], const __bf16* Ahi, const __bf16* Alo, const __bf16* Bhi,
                                          const __bf16* Blo, int wrow0, int wcol0, int lane) {
  constexpr int PITCH = Geo<BK>::PITCH;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) {
    bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int o = (wrow0 + m * 32 + r) * PITCH + ks * 16 + h * 8;
      ah[m] = *reinterpret_cast<const bf16x8*>(Ahi + o);
      if (NS == 3) al[m] = *reinterpret_cast<const bf16x8*>(Alo + o);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int o = (wcol0 + n * 32 + r) * PITCH + ks * 16 + h * 8;
      bh[n] = *reinterpret_cast<const bf16x8*>(Bhi + o);
      if (NS == 3) bl[n] = *reinterpret_cast<const bf16x8*>(Blo + o);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (NS == 3) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
        }
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
      }
  }
}

// ---- STORE epilogue for one (m-block, n-block) pair of accumulator registers; all uniform options are template
//      parameters so that the unrolled body has no branches at all (a scalar branch per element serialises the wave)
template <bool FULL, bool HAS_ADD, bool HAS_MASK, bool HAS_OUT, bool S16>
__device__ __forceinline__ void epi_store_block(const f32x16& acc, const GemmArgs& g, long long row0, int il0, int nrows, int j, bool jv,
                                                float bias, float msc, float msh, float& a1, float& a2) {
  float ad[16], zm[16];
  const int jc = jv ? j : (g.C - 1);
  if (HAS_ADD || HAS_MASK) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {          // unconditional, clamped loads first so they are all in flight together
      const int il = il0 + (e & 3) + 8 * (e >> 2);
      const long long oc = (row0 + (FULL || il < nrows ? il : nrows - 1)) * g.C + jc;
      if (HAS_ADD) ad[e] = act_ld<S16>(g.addend, oc);
      if (HAS_MASK) zm[e] = act_ld<S16>(g.zmask, oc);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (HAS_ADD) asm volatile("" : "+v"(ad[e]));
      if (HAS_MASK) asm volatile("" : "+v"(zm[e]));
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int il = il0 + (e & 3) + 8 * (e >> 2);
    const bool ok = FULL || (il < nrows && jv);
    float v = acc[e] + bias;
    if (HAS_ADD) v += ad[e];
    float w2 = v;
    if (HAS_MASK) {
      v = (fmaf(msc, zm[e], msh) > 0.f) ? v : 0.f;
      w2 = zm[e];
    }
    if (!FULL) v = ok ? v : 0.f;
    if (HAS_OUT) {
      if (FULL || ok) act_st<S16>(g.out, (row0 + il) * g.C + j, v);
    }
    a1 += v;
    a2 = fmaf(v, w2, a2);
  }
}

// ---- one weight-gradient tile: slab bx of output tile (by, bz) ----------------------------------------------
template <int BM, int BN, int NS, bool A2, bool B2, bool AH, bool BH>
__device__ __forceinline__ void wgrad_tile_t(const GemmArgs& g, const int bx, const int by, const int bz, unsigned char* lds_raw) {
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
  const int i0 = by * BM, j0 = bz * BN;
  const bool want_cs = g.colsum != 0 && bz == 0;     // block-uniform
  float cs = 0.f, cs2 = 0.f;
  const int rbeg = tin * g.K;
  const int rend = min(g.N, rbeg + g.K);
  // the loads of chunk i+1 are issued before the MFMAs of chunk i and converted after them: one register set, global latency
  // hidden behind the matrix cores
  TrnStage<BM, BK, A2, AH> sa;
  TrnStage<BN, BK, B2, BH> sb;
  // a diagonal tile of a Gram matrix (both operands the same tensor through the same coefficients, i0 == j0): the B image IS the A
  // image -- one operand is loaded, converted and written to LDS instead of two
  const bool same = (BM == BN) && !A2 && !B2 && (AH == BH) && i0 == j0 && g.Ci == g.C && g.a.s1 == g.b.s1 && g.a.ld == g.b.ld &&
                    g.a.ca == g.b.ca && g.a.cc == g.b.cc && g.a.lo == g.b.lo;      // block-uniform
  const __bf16* Bh = same ? Ahi : Bhi;
  const __bf16* Bl = same ? Alo : Blo;
  if (rbeg < rend) {
    const long long rowbase = (long long)cloud * g.N + rbeg;
    const int nk = min(BK, rend - rbeg);
    sa.issue(g.a, rowbase * g.a.ld + i0, nk, g.Ci - i0, tid);
    if (!same) sb.issue(g.b, rowbase * g.b.ld + j0, nk, g.C - j0, tid);
  }
  for (int r0 = rbeg; r0 < rend; r0 += BK) {
    const int nk = min(BK, rend - r0);
    sa.pin();
    const float2 csp = sa.template finish<NS>(Ahi, Alo, g.a, nk, g.Ci - i0, i0, tid, want_cs);
    cs += csp.x; cs2 += csp.y;
    if (!same) {
      sb.pin();
      sb.template finish<NS>(Bhi, Blo, g.b, nk, g.C - j0, j0, tid);
    }
    __syncthreads();
    if (r0 + BK < rend) {
      const long long rowbase = (long long)cloud * g.N + r0 + BK;
      const int nk2 = min(BK, rend - (r0 + BK));
      sa.issue(g.a, rowbase * g.a.ld + i0, nk2, g.Ci - i0, tid);
      if (!same) sb.issue(g.b, rowbase * g.b.ld + j0, nk2, g.C - j0, tid);
    }
    mma_chunk<MT, NT, BK, NS>(acc, Ahi, Alo, Bh, Bl, wrow0, wcol0, lane);
    __syncthreads();
  }
  // slab store
  const long long slab_stride = (long long)g.Ci * g.C + (g.colsum ? g.Ci : 0);
  float* slab = g.out + (long long)bx * slab_stride;
  if (want_cs) {
    // the 256 / BM threads that share a channel combine through LDS (the tiles are dead after the last barrier), fixed order
    constexpr int TPGA = 256 / BM;
    float* red = reinterpret_cast<float*>(lds_raw);
    if constexpr (AH) {                  // a lane staged a channel pair: twice the partitions per channel
      constexpr int BM2 = BM / 2;
      red[(tid / BM2) * BM + 2 * (tid % BM2)] = cs;
      red[(tid / BM2) * BM + 2 * (tid % BM2) + 1] = cs2;
    } else {
      red[(tid / BM) * BM + (tid % BM)] = cs;
    }
    __syncthreads();
    if (tid < BM && i0 + tid < g.Ci) {
      float t = red[tid];
      constexpr int PARTS = AH ? 2 * TPGA : TPGA;
#pragma unroll
      for (int q = 1; q < PARTS; ++q) t += red[q * BM + tid];
      slab[(long long)g.Ci * g.C + i0 + tid] = t;
    }
  }
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int j = j0 + wcol0 + n * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = i0 + wrow0 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (i < g.Ci && j < g.C) slab[(long long)i * g.C + j] = acc[m][n][e];
      }
    }
}

// the storage type of either operand is block-uniform: switch once, outside the loops (pn_common.h: act_switch)
template <int BM, int BN, int NS, bool A2, bool B2>
__device__ __forceinline__ void wgrad_tile(const GemmArgs& g, const int bx, const int by, const int bz, unsigned char* lds_raw) {
  if (g.a.h16) {
    if (g.b.h16) wgrad_tile_t<BM, BN, NS, A2, B2, true, true>(g, bx, by, bz, lds_raw);
    else wgrad_tile_t<BM, BN, NS, A2, B2, true, false>(g, bx, by, bz, lds_raw);
  } else {
    if (g.b.h16) wgrad_tile_t<BM, BN, NS, A2, B2, false, true>(g, bx, by, bz, lds_raw);
    else wgrad_tile_t<BM, BN, NS, A2, B2, false, false>(g, bx, by, bz, lds_raw);
  }
}

template <int BM, int BN, int NS>
struct GemmLds {
  static constexpr int BK = (NS == 3) ? 32 : 64;
  static constexpr int TILES = (BM + BN) * Geo<BK>::PITCH * ((NS == 3) ? 2 : 1) * 2;
  static constexpr int EPI = 2 * BN * 4 * 4;
  static constexpr int BYTES = TILES > EPI ? TILES : EPI;
};

// Several weight-gradient jobs of one tile shape in one launch (the parameter gradients of a backward pass wait for the end
// of the pass, pn_model.hip): the linear block index walks job -> (bz, by, slab).
constexpr int WGRAD_BATCH_MAX = 4;
struct WgradBatch {
  GemmArgs g[WGRAD_BATCH_MAX];
  int blk_end[WGRAD_BATCH_MAX];
  int nslab[WGRAD_BATCH_MAX];
  int ny[WGRAD_BATCH_MAX];
  int n;
};
template <int BM, int BN, int NS, bool B2>
__device__ __forceinline__ void wgrad_batch_tile(const WgradBatch& wb, const int bxi, unsigned char* lds_raw) {
  int j = 0;
  while (j + 1 < wb.n && bxi >= wb.blk_end[j]) ++j;      // block-uniform
  const int first = j ? wb.blk_end[j - 1] : 0;
  const int local = bxi - first;
  const int nslab = wb.nslab[j], ny = wb.ny[j];
  const int n_out = (wb.blk_end[j] - first) / nslab;          // output tiles per slab
  // Block -> (slab, output tile).  The n_out tiles of a slab read the same rows of both operands; workgroups are dealt round-robin
  // over the 8 XCDs (observed, never relied on for correctness: MI355X_MICROARCH.md, Workgroup dispatch), so inside a group of 8
  // slabs the block id walks tile-major with the slab in its low 3 bits: one slab's tiles land on one XCD, next to each other in
  // time, and its re-reads are served by that XCD's L2 instead of the fabric.  Slabs beyond the last full group: slab-fastest.
  int bx, tile;
  const int full = (nslab / 8) * 8 * n_out;
  if (local < full) {
    const int grp = local / (8 * n_out), r = local - grp * 8 * n_out;
    tile = r >> 3;
    bx = grp * 8 + (r & 7);
  } else {
    const int l2 = local - full, rem = nslab - (nslab / 8) * 8;
    bx = (nslab / 8) * 8 + l2 % rem;
    tile = l2 / rem;
  }
  wgrad_tile<BM, BN, NS, false, B2>(wb.g[j], bx, tile % ny, tile / ny, lds_raw);
}

// host side (pn_gemm.hip): one job of tile shape bm x bn as a batch of its own
int wgrad_batch_one(const WgradDesc& q, int bm, int bn, WgradBatch& wb, int& blocks);
}  // namespace pn
