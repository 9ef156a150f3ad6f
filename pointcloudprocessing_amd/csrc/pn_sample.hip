// Point-cloud downsampling for dense scans: farthest point sampling (gfx950); the voxel grid is in pn_voxel.hip.
// The reference has neither (its only resize is truncate / random duplicate,
// pointcloud/PointCloudSet.py:443-470); the specification is build-defined and stated in pointnet_hip.h,
// with NumPy oracles in oracle/sampling_oracle.py.
#include <cstring>
#include "pn_common.h"

namespace pn {
int zero_fill(float* p, long long n, hipStream_t st);   // pn_optim.hip

// ------------------------------------------------------------------------------------------------------
// FPS.  M sequential rounds, each a full argmax over the cloud: latency bound, not bandwidth bound.  A
// block keeps its points (xyz + running min-distance, 4..28 per thread) in registers, so a round touches no
// memory at all: the winner's coordinates travel through LDS.  Measured per round (MI355X): ~0.75 us of
// reductions / barriers / LDS round trips + the distance update (5.75 VALU instructions per point):
// 0.76 us at N <= 1024 (4 waves), 0.93 us at 4096, 1.56 us at 16384 (16 waves), 1.77 us at 21504 (12 waves x 28
// points).  512-thread variants with 32 / 42 points per thread measured the same or slower, and so did a fast path
// for "one wave holds the maximum" (winner published in a single 16-byte LDS slot: +0.07 us from counting candidates).  Clouds above 21504
// points are split over `bpc` co-resident blocks that exchange one 8-byte {distance, index, round-tag} granule
// per round through device-scope relaxed atomics (a single naturally aligned 8-byte sc1 store/load needs no
// other ordering: MI355X_MICROARCH.md, "R2's granule").  Every spin is bounded.
// ------------------------------------------------------------------------------------------------------
constexpr int FPS_T_MULTI = 1024;
constexpr int FPS_PPT_MULTI = 16;     // points per thread when a cloud is split over blocks (and for clouds <= 16384 points)
constexpr int FPS_PPT_WIDE = 28;      // single block of 768 threads (12 waves -> 170 VGPRs each) x 28 points = 21504 points, no spill
constexpr int FPS_T_WIDE = 768;
constexpr int FPS_PER_BLOCK = FPS_T_MULTI * FPS_PPT_MULTI;

__device__ __forceinline__ void fps_better(float& best, int& bi, float ob, int oi) {
  if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
}

// Wave-wide max / min in six DPP steps (quad swaps, half-row and row mirrors, then the two row broadcasts of gfx9): the
// result is complete in lane 63 and handed back as a wave-uniform value.  A shuffle-based tree costs six LDS-crossbar
// round trips instead, and this sits on the critical path of every sampling round.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_move(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, dpp_move<0xB1, 0xf>(v));    // quad_perm [1,0,3,2]
  v = max(v, dpp_move<0x4E, 0xf>(v));    // quad_perm [2,3,0,1]
  v = max(v, dpp_move<0x141, 0xf>(v));   // row_half_mirror
  v = max(v, dpp_move<0x140, 0xf>(v));   // row_mirror
  v = max(v, dpp_move<0x142, 0xa>(v));   // row_bcast:15 into rows 1, 3
  v = max(v, dpp_move<0x143, 0xc>(v));   // row_bcast:31 into rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, dpp_move<0xB1, 0xf>(v));
  v = min(v, dpp_move<0x4E, 0xf>(v));
  v = min(v, dpp_move<0x141, 0xf>(v));
  v = min(v, dpp_move<0x140, 0xf>(v));
  v = min(v, dpp_move<0x142, 0xa>(v));
  v = min(v, dpp_move<0x143, 0xc>(v));
  return __builtin_amdgcn_readlane(v, 63);
}

typedef float fps_v2f __attribute__((ext_vector_type(2)));

// A round is (1) distance update + running max of the VALUE only: packed fp32 math, 5.5 VALU instructions per point;
// (2) wave max (DPP) -> LDS -> barrier -> block max; (3) only in the wave(s) that hold the block max: lowest point index
// with that distance, and the winner lane's coordinates picked out of its registers through a wave-uniform switch -> LDS
// -> barrier.  Tracking the arg-max index inside the update loop costs twice the instructions of step (1).
template <int FPS_PPT, int FPS_T>
__global__ __launch_bounds__(FPS_T) void fps_kernel(const float* __restrict__ xyz, int N, int M, int start_idx, int bpc,
                                                    int* __restrict__ idx_out, float* __restrict__ mindist,
                                                    unsigned long long* __restrict__ xchg, int* __restrict__ err) {
#pragma clang fp contract(off)   // the distance is specified without fused multiply-add (bit-exact vs the oracle)
  static_assert(FPS_PPT % 2 == 0 && FPS_PPT <= 32 && FPS_T % 256 == 0, "fps_kernel: shape");
  constexpr int NW = FPS_T / 64;
  __shared__ __attribute__((aligned(16))) int s_best[16];
  __shared__ __attribute__((aligned(16))) int s_widx[16];
  __shared__ __attribute__((aligned(16))) float s_xyz[16][4];   // coordinates of each wave's candidate: the next round starts without a global load
  __shared__ int s_cur;
  const int cloud = blockIdx.x / bpc, blk = blockIdx.x - cloud * bpc;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* p = xyz + (long long)cloud * N * 3;
  const int base = blk * (FPS_T * FPS_PPT);
  fps_v2f px[FPS_PPT / 2], py[FPS_PPT / 2], pz[FPS_PPT / 2];   // point j of this thread = cloud index base + j * FPS_T + tid
  // Running min-distance, kept as the BIT PATTERN of the fp32 value: distances are sums of squares (>= +0, never -0 or NaN
  // for finite input) and padding is -1.0f, and on that set signed-integer order equals float order -- so min / max are
  // single integer instructions (a float min would first canonicalise its operand: one more instruction per point).
  int md[FPS_PPT];
  constexpr int MD_PAD = (int)0xbf800000u, MD_INF = 0x7f800000;   // -1.0f, +inf
#pragma unroll
  for (int j = 0; j < FPS_PPT; ++j) {
    const int i = base + j * FPS_T + tid;
    float x = 0.f, y = 0.f, z = 0.f;
    md[j] = MD_PAD;                                    // padding: min(-1, d) stays -1 and never wins a round
    if (i < N) { x = p[3 * i]; y = p[3 * i + 1]; z = p[3 * i + 2]; md[j] = MD_INF; }
    if (j & 1) { px[j / 2].y = x; py[j / 2].y = y; pz[j / 2].y = z; }
    else       { px[j / 2].x = x; py[j / 2].x = y; pz[j / 2].x = z; }
  }
  int cur = start_idx;
  float cx = p[3 * cur], cy = p[3 * cur + 1], cz = p[3 * cur + 2];
  unsigned long long* xc = xchg + (long long)cloud * 2 * bpc;
  for (int it = 0; it < M; ++it) {
    if (blk == 0 && tid == 0) idx_out[(long long)cloud * M + it] = cur;
    if (it == M - 1) break;
    // (1) update
    // the centre is wave-uniform: keep it in scalar registers (the 28-point variant has no vector register to spare)
    const float ux = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cx)));
    const float uy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cy)));
    const float uz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cz)));
    const fps_v2f c_x = {ux, ux}, c_y = {uy, uy}, c_z = {uz, uz};
    int best = MD_PAD;
#pragma unroll
    for (int q = 0; q < FPS_PPT / 2; ++q) {
      const fps_v2f dx = px[q] - c_x, dy = py[q] - c_y, dz = pz[q] - c_z;
      const fps_v2f d = (dx * dx + dy * dy) + dz * dz;   // no contraction: see the pragma above
      md[2 * q] = min(md[2 * q], __float_as_int(d.x));
      md[2 * q + 1] = min(md[2 * q + 1], __float_as_int(d.y));
      best = max(best, max(md[2 * q], md[2 * q + 1]));
    }
    // (2) block max of the value
    const int wmax = wave_max_i32(best);
    if (lane == 0) s_best[wave] = wmax;
    __syncthreads();
    int gmax = s_best[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) gmax = max(gmax, s_best[w]);
    // (3) lowest index holding it
    int widx = 0x7fffffff;
    if (wmax == gmax) {                                 // wave-uniform
      int jw = FPS_PPT;
#pragma unroll
      for (int j = FPS_PPT - 1; j >= 0; --j) jw = (md[j] == gmax) ? j : jw;
      const int cand = jw < FPS_PPT ? base + jw * FPS_T + tid : 0x7fffffff;
      widx = wave_min_i32(cand);
      if (cand == widx && jw < FPS_PPT) {               // exactly one lane: publish its point's coordinates
        const int ju = __builtin_amdgcn_readfirstlane(jw);
        float bx = 0.f, by = 0.f, bz = 0.f;
#define PN_FPS_CASE(J)                                                                                     \
  case J:                                                                                                  \
    if constexpr (J < FPS_PPT) {                                                                           \
      bx = (J & 1) ? px[J / 2].y : px[J / 2].x;                                                            \
      by = (J & 1) ? py[J / 2].y : py[J / 2].x;                                                            \
      bz = (J & 1) ? pz[J / 2].y : pz[J / 2].x;                                                            \
    }                                                                                                      \
    break;
        switch (ju) {
          PN_FPS_CASE(0) PN_FPS_CASE(1) PN_FPS_CASE(2) PN_FPS_CASE(3) PN_FPS_CASE(4) PN_FPS_CASE(5) PN_FPS_CASE(6) PN_FPS_CASE(7)
          PN_FPS_CASE(8) PN_FPS_CASE(9) PN_FPS_CASE(10) PN_FPS_CASE(11) PN_FPS_CASE(12) PN_FPS_CASE(13) PN_FPS_CASE(14) PN_FPS_CASE(15)
          PN_FPS_CASE(16) PN_FPS_CASE(17) PN_FPS_CASE(18) PN_FPS_CASE(19) PN_FPS_CASE(20) PN_FPS_CASE(21) PN_FPS_CASE(22) PN_FPS_CASE(23)
          PN_FPS_CASE(24) PN_FPS_CASE(25) PN_FPS_CASE(26) PN_FPS_CASE(27) PN_FPS_CASE(28) PN_FPS_CASE(29) PN_FPS_CASE(30) PN_FPS_CASE(31)
          default: break;
        }
#undef PN_FPS_CASE
        s_xyz[wave][0] = bx; s_xyz[wave][1] = by; s_xyz[wave][2] = bz;
      }
    }
    if (lane == 0) s_widx[wave] = widx;
    __syncthreads();
    int bi = s_widx[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) bi = min(bi, s_widx[w]);
    const float bestv = __int_as_float(gmax);
    if (bpc == 1) {
      const int bw = ((bi - base) % FPS_T) >> 6;        // the wave that owns point bi
      cx = s_xyz[bw][0]; cy = s_xyz[bw][1]; cz = s_xyz[bw][2];
    }
    if (bpc > 1) {
      const int par = it & 1;
      const unsigned tag = (unsigned)(it + 1) & 0xfffu;
      if (tid == 0) {
        const unsigned long long gr = ((unsigned long long)__float_as_uint(bestv) << 32) |
                                      ((unsigned long long)((unsigned)bi & 0xfffffu) << 12) | tag;
        __hip_atomic_store(&xc[par * bpc + blk], gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (wave == 0) {
        float gb = -1.f;
        int gi = 0x7fffffff;
        int timed_out = 0;
        if (lane < bpc) {
          unsigned long long gr = 0;
          int spins = 0;
          for (;;) {
            gr = __hip_atomic_load(&xc[par * bpc + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(gr & 0xfffu) == tag) break;
            if (++spins > (1 << 22)) { timed_out = 1; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          gb = __uint_as_float((unsigned)(gr >> 32));
          gi = (int)((gr >> 12) & 0xfffffu);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float ob = __shfl_xor(gb, o, 64);
          const int oi = __shfl_xor(gi, o, 64);
          fps_better(gb, gi, ob, oi);
        }
        const int any_to = __any(timed_out);
        if (lane == 0) {
          s_cur = any_to ? -1 : gi;
          if (any_to) atomicExch(err, 1);
        }
      }
      __syncthreads();
      bi = s_cur;
      if (bi < 0) break;   // block-uniform: a peer never published (error flag set)
      cx = p[3 * bi]; cy = p[3 * bi + 1]; cz = p[3 * bi + 2];   // the winner may live in another block
    }
    cur = bi;
  }
  if (mindist) {
    int t2 = tid;
    asm volatile("" : "+v"(t2));   // or the 28 point indices of the prologue stay live in registers across the whole loop
#pragma unroll
    for (int j = 0; j < FPS_PPT; ++j) {
      const int i = base + j * FPS_T + t2;
      if (i < N) mindist[(long long)cloud * N + i] = __int_as_float(md[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// FPS with EXACT spatial pruning (round 3): clouds of 4097 .. 20480 points (BASELINE config 5: the voxel grid's ~20 k centroids).
// A sampling round changes the running minimum distance of a point only if the new sample is closer to it than its nearest earlier
// sample -- late in a run a small neighbourhood.  The block keeps, per GROUP of 64 consecutive points (lane j of a wave holds group j of
// that wave): the group's bounding box and the maximum of its points' minimum distances (gm).  Round: every lane j computes the squared
// distance from the new sample to ITS group's box -- with the very operations a point uses, in the same order, each of them monotone
// in fp32, so the box value is <= the value every point of the group would get, exactly -- and a group whose box value is >= gm cannot
// change: it is skipped, result identical.  A wave owns 40 CONSECUTIVE groups (2560 consecutive points: a compact region when the
// input is spatially ordered, as the voxel grid's (kz, ky, kx) order is), so most waves skip a round entirely and republish their
// cached candidate.  One barrier per round: every wave publishes {max min-distance, lowest index holding it, that point's
// coordinates} into a parity-double-buffered LDS slot; after the barrier every wave reduces the eight slots itself.
// Inputs in no spatial order (every box spans the cloud) make every test fail: a wave that found nothing to skip for 64 rounds in a
// row stops testing (plain update of all its points from then on; the group maxima are no longer maintained).
// Bit-exact against oracle/sampling_oracle.py like the plain kernel: same distances, ties -> lowest index.
// ------------------------------------------------------------------------------------------------------
constexpr int FPS_PR_T = 512, FPS_PR_PPT = 40;
__device__ __forceinline__ float wave_min_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#define PN_FPS40(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) \
  X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39)
__global__ __launch_bounds__(FPS_PR_T) void fps_pruned_kernel(const float* __restrict__ xyz, int N, int M, int start_idx,
                                                              int* __restrict__ idx_out, float* __restrict__ mindist, int plain_from_start) {
#pragma clang fp contract(off)   // the distance is specified without fused multiply-add (bit-exact vs the oracle)
  constexpr int PPT = FPS_PR_PPT, T = FPS_PR_T, NW = T / 64;
  __shared__ __attribute__((aligned(16))) int s_cand[2][NW][2];      // [round parity][wave] {max min-distance (bit pattern), lowest index}
  __shared__ __attribute__((aligned(16))) float s_cxyz[2][NW][4];    // that point's coordinates
  const int cloud = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* p = xyz + (long long)cloud * N * 3;
  const int wbase = wave * (64 * PPT);                   // point j of this thread = cloud index wbase + 64 * j + lane
  fps_v2f px[PPT / 2], py[PPT / 2], pz[PPT / 2];
  int md[PPT];
  constexpr int MD_PAD = (int)0xbf800000u, MD_INF = 0x7f800000;   // -1.0f, +inf
  float blx = INFINITY, bly = INFINITY, blz = INFINITY, bhx = -INFINITY, bhy = -INFINITY, bhz = -INFINITY;
  int gm = MD_PAD;                                       // lane j: max of the running minimum distances of group j (bit pattern)
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int i = wbase + 64 * j + lane;
    float x = 0.f, y = 0.f, z = 0.f;
    md[j] = MD_PAD;
    const bool v = i < N;
    if (v) { x = p[3 * i]; y = p[3 * i + 1]; z = p[3 * i + 2]; md[j] = MD_INF; }
    if (j & 1) { px[j / 2].y = x; py[j / 2].y = y; pz[j / 2].y = z; }
    else       { px[j / 2].x = x; py[j / 2].x = y; pz[j / 2].x = z; }
    const float lx = wave_min_f32(v ? x : INFINITY), ly = wave_min_f32(v ? y : INFINITY), lz = wave_min_f32(v ? z : INFINITY);
    const float hx = wave_max_f32(v ? x : -INFINITY), hy = wave_max_f32(v ? y : -INFINITY), hz = wave_max_f32(v ? z : -INFINITY);
    const int g0 = wave_max_i32(md[j]);
    if (lane == j) { blx = lx; bly = ly; blz = lz; bhx = hx; bhy = hy; bhz = hz; gm = g0; }
  }
  int cur = start_idx;
  float cx = p[3 * cur], cy = p[3 * cur + 1], cz = p[3 * cur + 2];
  // this wave's candidate (wave-uniform): recomputed only in rounds that changed one of its points
  int c_val = MD_PAD, c_idx = 0x7fffffff;
  float c_x = 0.f, c_y = 0.f, c_z = 0.f;
  bool have_cand = false;
  bool plain = plain_from_start != 0;                    // wave-uniform: no longer testing (nothing was ever skipped)
  int full_run = 0;
  const unsigned long long all_groups = (1ull << PPT) - 1ull;
  for (int it = 0; it < M; ++it) {
    if (tid == 0) idx_out[(long long)cloud * M + it] = cur;
    if (it == M - 1) break;
    const float ux = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cx)));
    const float uy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cy)));
    const float uz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cz)));
    // (0) the groups the new sample can change
    unsigned long long mask = all_groups;
    if (!plain) {
      const float tlx = blx - ux, thx = bhx - ux, tly = bly - uy, thy = bhy - uy, tlz = blz - uz, thz = bhz - uz;
      const float ax = tlx > 0.f ? tlx : (thx < 0.f ? thx : 0.f);
      const float ay = tly > 0.f ? tly : (thy < 0.f ? thy : 0.f);
      const float az = tlz > 0.f ? tlz : (thz < 0.f ? thz : 0.f);
      const float bound = (ax * ax + ay * ay) + az * az;              // <= the distance of every point of the group, exactly
      mask = __ballot(__float_as_int(bound) < gm) & all_groups;       // (an empty group has gm = -1.0f: never)
      full_run = mask == all_groups ? full_run + 1 : 0;
      if (full_run >= 64) plain = true;
    }
    if (mask != 0ull || !have_cand) {
      const fps_v2f c_xv = {ux, ux}, c_yv = {uy, uy}, c_zv = {uz, uz};
      int best = MD_PAD;
#pragma unroll
      for (int q = 0; q < PPT / 2; ++q) {
        if ((mask >> (2 * q)) & 3ull) {                  // wave-uniform
          const fps_v2f dx = px[q] - c_xv, dy = py[q] - c_yv, dz = pz[q] - c_zv;
          const fps_v2f d = (dx * dx + dy * dy) + dz * dz;
          md[2 * q] = min(md[2 * q], __float_as_int(d.x));
          md[2 * q + 1] = min(md[2 * q + 1], __float_as_int(d.y));
          if (!plain) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int j = 2 * q + u;
              const int old = __builtin_amdgcn_readlane(gm, j);
              if (__ballot(md[j] == old) == 0ull) {      // the point that held the group's maximum moved: take the maximum again
                const int nm = wave_max_i32(md[j]);
                gm = lane == j ? nm : gm;
              }
            }
          }
        }
        if (plain) best = max(best, max(md[2 * q], md[2 * q + 1]));
      }
      // (1) this wave's candidate: its largest minimum distance and the lowest index holding it
      const int wmax = plain ? wave_max_i32(best) : wave_max_i32(lane < PPT ? gm : MD_PAD);
      int jw = PPT;
      if (plain) {
#pragma unroll
        for (int j = PPT - 1; j >= 0; --j) jw = (md[j] == wmax) ? j : jw;
      }
      int js, ls;
      if (plain) {
        const int cand = jw < PPT ? 64 * jw + lane : 0x7fffffff;
        const int wi = wave_min_i32(cand);
        js = wi >> 6; ls = wi & 63;
        if (wi == 0x7fffffff) { js = 0; ls = 0; }
      } else {
        const unsigned long long gj = __ballot(lane < PPT && gm == wmax);
        js = __builtin_amdgcn_readfirstlane(gj ? __ffsll((long long)gj) - 1 : 0);
        int vsel = MD_PAD;
        switch (js) {
#define PN_FPS_SEL(J) case J: vsel = md[J]; break;
          PN_FPS40(PN_FPS_SEL)
#undef PN_FPS_SEL
          default: break;
        }
        const unsigned long long gl = __ballot(vsel == wmax);
        ls = __builtin_amdgcn_readfirstlane(gl ? __ffsll((long long)gl) - 1 : 0);
      }
      js = __builtin_amdgcn_readfirstlane(js);
      ls = __builtin_amdgcn_readfirstlane(ls);
      float bx = 0.f, by = 0.f, bz = 0.f;
      switch (js) {
#define PN_FPS_XYZ(J) case J: bx = (J & 1) ? px[J / 2].y : px[J / 2].x; by = (J & 1) ? py[J / 2].y : py[J / 2].x; \
                              bz = (J & 1) ? pz[J / 2].y : pz[J / 2].x; break;
        PN_FPS40(PN_FPS_XYZ)
#undef PN_FPS_XYZ
        default: break;
      }
      c_val = wmax;
      c_idx = wmax == MD_PAD ? 0x7fffffff : wbase + 64 * js + ls;     // a wave without points never wins
      c_x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx), ls));
      c_y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(by), ls));
      c_z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bz), ls));
      have_cand = true;
    }
    // (2) publish, one barrier, every wave reduces the NW slots itself
    const int par = it & 1;
    if (lane == 0) {
      s_cand[par][wave][0] = c_val; s_cand[par][wave][1] = c_idx;
      s_cxyz[par][wave][0] = c_x; s_cxyz[par][wave][1] = c_y; s_cxyz[par][wave][2] = c_z;
    }
    __syncthreads();
    const int wl = lane < NW ? lane : 0;
    int v = s_cand[par][wl][0], ix = s_cand[par][wl][1];
    if (lane >= NW) { v = MD_PAD; ix = 0x7fffffff; }
    const int gmax = wave_max_i32(v);
    const int bi = wave_min_i32(v == gmax ? ix : 0x7fffffff);
    const unsigned long long who = __ballot(v == gmax && ix == bi);
    const int bw = who ? __ffsll((long long)who) - 1 : 0;
    cx = s_cxyz[par][bw][0]; cy = s_cxyz[par][bw][1]; cz = s_cxyz[par][bw][2];
    cur = bi;
  }
  if (mindist) {
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int i = wbase + 64 * j + lane;
      if (i < N) mindist[(long long)cloud * N + i] = __int_as_float(md[j]);
    }
  }
}
#undef PN_FPS40

static inline int fps_blocks_per_cloud(int N) { return N <= FPS_T_WIDE * FPS_PPT_WIDE ? 1 : cdiv(N, FPS_PER_BLOCK); }

size_t fps_workspace_bytes(int B, int N) {
  const int bpc = fps_blocks_per_cloud(N);
  return 16 + (size_t)B * 2 * bpc * sizeof(unsigned long long);
}

template <int PPT, int T>
static void fps_launch(int blocks, hipStream_t st, const float* xyz, int N, int M, int start_idx, int bpc, int* idx_out, float* mindist,
                       unsigned long long* xchg, int* err) {
  hipLaunchKernelGGL((fps_kernel<PPT, T>), dim3(blocks), dim3(T), 0, st, xyz, N, M, start_idx, bpc, idx_out, mindist, xchg, err);
}

int fps(const float* xyz, int B, int N, int M, int start_idx, int* idx_out, float* mindist, void* ws, size_t ws_bytes,
        hipStream_t st) {
  PN_CHECK_ARG(xyz && idx_out, "pn_fps: null pointer");
  PN_CHECK_ARG(B > 0 && N > 0 && M > 0, "pn_fps: B, N, M must be positive (B=%d N=%d M=%d)", B, N, M);
  PN_CHECK_ARG(start_idx >= 0 && start_idx < N, "pn_fps: start_idx %d outside [0,%d)", start_idx, N);
  const int bpc = fps_blocks_per_cloud(N);
  PN_CHECK_ARG(bpc <= 64, "pn_fps: N=%d exceeds %d points per cloud", N, 64 * FPS_PER_BLOCK);
  PN_CHECK_ARG(ws && ws_bytes >= fps_workspace_bytes(B, N), "pn_fps: workspace too small");
  int* err = reinterpret_cast<int*>(ws);
  unsigned long long* xchg = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + 16);
  PN_TRY(zero_fill(reinterpret_cast<float*>(ws), (long long)(fps_workspace_bytes(B, N) / 4), st));
  // blocks of one cloud must be co-resident (they wait for each other): at most 128 blocks per launch
  const int clouds_per_launch = bpc > 1 ? (128 / bpc > 0 ? 128 / bpc : 1) : B;
  for (int b0 = 0; b0 < B; b0 += clouds_per_launch) {
    const int nb = (B - b0) < clouds_per_launch ? (B - b0) : clouds_per_launch;
    const float* x0 = xyz + (long long)b0 * N * 3;
    int* i0 = idx_out + (long long)b0 * M;
    float* m0 = mindist ? mindist + (long long)b0 * N : nullptr;
    unsigned long long* c0 = xchg + (long long)b0 * 2 * bpc;
    // a round costs a fixed ~1 us of reductions and barriers plus the distance update: small clouds take fewer waves
    // (cheaper barriers, one wave per SIMD) and fewer points per thread
    // PN_FPS_PRUNE: 0 (default) the plain kernels; 1 the pruned kernel; 2 the pruned kernel's one-barrier round without the test
    const int prune = getenv("PN_FPS_PRUNE") ? atoi(getenv("PN_FPS_PRUNE")) : 0;      // (read at every call: the tests flip it)
    if (bpc == 1 && prune && N > 256 * 16 && N <= FPS_PR_T * FPS_PR_PPT) {
      hipLaunchKernelGGL(fps_pruned_kernel, dim3(nb), dim3(FPS_PR_T), 0, st, x0, N, M, start_idx, i0, m0, prune == 2 ? 1 : 0);
      PN_CHECK_LAUNCH();
      continue;
    }
    if (bpc > 1) fps_launch<FPS_PPT_MULTI, FPS_T_MULTI>(nb * bpc, st, x0, N, M, start_idx, bpc, i0, m0, c0, err);
    else if (N <= 256 * 4) fps_launch<4, 256>(nb, st, x0, N, M, start_idx, 1, i0, m0, c0, err);
    else if (N <= 256 * 16) fps_launch<16, 256>(nb, st, x0, N, M, start_idx, 1, i0, m0, c0, err);
    else if (N <= FPS_PER_BLOCK) fps_launch<FPS_PPT_MULTI, FPS_T_MULTI>(nb, st, x0, N, M, start_idx, 1, i0, m0, c0, err);
    else fps_launch<FPS_PPT_WIDE, FPS_T_WIDE>(nb, st, x0, N, M, start_idx, 1, i0, m0, c0, err);
    PN_CHECK_LAUNCH();
  }
  return PN_OK;
}

}  // namespace pn
