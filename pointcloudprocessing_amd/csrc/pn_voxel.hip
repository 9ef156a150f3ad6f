// Voxel-grid downsample for dense scans (gfx950).  The reference has none; the specification is build-defined
// and stated in pointnet_hip.h, with the NumPy oracle in oracle/sampling_oracle.py.
//
//   keys      key = (kz << 42) | (ky << 21) | kx, k = floor((p - origin) / leaf) per axis, 21 bits each
//   sort      stable LSD radix sort of (key, point index), hand-written: nine digit positions (three per axis: bits [0,8), [8,16),
//             [16,21) of kx, ky, kz).  The keys kernel also builds the global histogram of every digit position; a position whose
//             histogram has one bin (all keys share the digit: everything above an axis' occupied bits) is a no-op of a stable sort
//             and its launch exits at once -- kc-46-sized extents at a 0.25 m leaf occupy 8 + 8 + 6 bits: three live passes.
//             A live pass is ONE kernel: a workgroup takes a tile in ticket order, ranks its keys (per-wave digit counters in LDS,
//             ranks inside a wave from eight ballots), publishes the tile's digit histogram as 4-byte {status, count} granules
//             (one sc1 store each: MI355X_MICROARCH.md "R2's granule"), sums its predecessors' granules (decoupled look-back:
//             batches of eight rows, nearest first, stopping at the first inclusive prefix) and scatters.  Tickets make every
//             predecessor a running workgroup, so the waits cannot deadlock; every spin is bounded and reported through `err`.
//   heads     segment starts of the sorted keys: the same ticket + look-back scheme on one counter per tile
//   reduce    per voxel: fp64 centroid in point-index order (the sort is stable), count, majority label
#include <cstring>
#include "pn_common.h"

namespace pn {
int zero_fill(float* p, long long n, hipStream_t st);   // pn_optim.hip

constexpr int VX_POS = 9;                  // digit positions
constexpr int VX_T = 256;                  // threads per workgroup
constexpr int VX_HEAD_PER_THREAD = 8;
constexpr unsigned VX_SPIN_LIMIT = 1u << 22;

__host__ __device__ __forceinline__ int vx_shift(int p) { return (p / 3) * 21 + (p % 3) * 8; }
__host__ __device__ __forceinline__ int vx_bits(int p) { return (p % 3) == 2 ? 5 : 8; }
__device__ __forceinline__ int vx_digit(unsigned long long key, int p) { return (int)((key >> vx_shift(p)) & ((1u << vx_bits(p)) - 1u)); }

// control block at the start of the workspace, cleared before every call
struct VxCtrl {
  int err;                       // 1: a key outside [0, 2^21); 2: a look-back spin ran out
  int n_live;
  int final_sel;                 // which of the two (key, index) buffers holds the sorted pairs
  int pad;
  int ticket[VX_POS + 1];        // tile tickets of the nine passes and of the heads kernel
  int live[VX_POS];
  int src_sel[VX_POS];
  int hist[VX_POS][256];         // global digit histograms
  int base[VX_POS][256];         // their exclusive scans
};

__device__ __forceinline__ unsigned vx_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void vx_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(VX_T) void voxel_keys_kernel(const float* __restrict__ xyz, int N, float lx, float ly, float lz, float ox,
                                                          float oy, float oz, unsigned long long* __restrict__ keys,
                                                          int* __restrict__ vals, VxCtrl* __restrict__ ctrl) {
#pragma clang fp contract(off)
  __shared__ int h[VX_POS][256];
  for (int i = threadIdx.x; i < VX_POS * 256; i += VX_T) (&h[0][0])[i] = 0;
  __syncthreads();
  for (int i = blockIdx.x * VX_T + threadIdx.x; i < N; i += gridDim.x * VX_T) {
    const float fx = floorf((xyz[3 * i] - ox) / lx);
    const float fy = floorf((xyz[3 * i + 1] - oy) / ly);
    const float fz = floorf((xyz[3 * i + 2] - oz) / lz);
    const float lim = 2097152.f;  // 2^21
    if (!(fx >= 0.f && fx < lim && fy >= 0.f && fy < lim && fz >= 0.f && fz < lim)) atomicExch(&ctrl->err, 1);
    const unsigned long long kx = (unsigned long long)fminf(fmaxf(fx, 0.f), lim - 1.f);
    const unsigned long long ky = (unsigned long long)fminf(fmaxf(fy, 0.f), lim - 1.f);
    const unsigned long long kz = (unsigned long long)fminf(fmaxf(fz, 0.f), lim - 1.f);
    const unsigned long long key = (kz << 42) | (ky << 21) | kx;
    keys[i] = key;
    vals[i] = i;
#pragma unroll
    for (int p = 0; p < VX_POS; ++p) atomicAdd(&h[p][vx_digit(key, p)], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < VX_POS * 256; i += VX_T) {
    const int c = (&h[0][0])[i];
    if (c) atomicAdd(&ctrl->hist[0][0] + i, c);
  }
}

// one workgroup: which digit positions are live, which buffer each live pass reads, exclusive scans of the histograms
__global__ __launch_bounds__(256) void voxel_plan_kernel(VxCtrl* __restrict__ ctrl, int N) {
  __shared__ int wsum[4];
  __shared__ int degenerate[VX_POS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < VX_POS) degenerate[tid] = 0;
  __syncthreads();
  for (int p = 0; p < VX_POS; ++p) {
    const int c = ctrl->hist[p][tid];
    if (c == N) degenerate[p] = 1;
    int v = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(v, o, 64);
      if (lane >= o) v += t;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    ctrl->base[p][tid] = off + v - c;
    __syncthreads();
  }
  if (tid == 0) {
    int sel = 0, n = 0;
    for (int p = 0; p < VX_POS; ++p) {
      const int lv = degenerate[p] ? 0 : 1;
      ctrl->live[p] = lv;
      ctrl->src_sel[p] = sel;
      sel ^= lv;
      n += lv;
    }
    ctrl->final_sel = sel;
    ctrl->n_live = n;
  }
}

// decoupled look-back over the 4-byte granules {status (2 bits: 1 aggregate, 2 inclusive prefix), count (30 bits)} of the tiles in
// front of `tile`, for the caller's own column `col` of a row of `stride` granules: returns the exclusive prefix
__device__ __forceinline__ unsigned vx_lookback(const unsigned* __restrict__ status, int tile, int stride, int col, int* __restrict__ err) {
  unsigned excl = 0;
  int t = tile - 1;
  bool done = false;
  while (t >= 0 && !done) {
    unsigned v[8];
    const int nb = t + 1 < 8 ? t + 1 : 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = q < nb ? vx_ld(status + (long long)(t - q) * stride + col) : (2u << 30);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      unsigned spins = 0;
      while ((v[q] >> 30) == 0u) {                       // the predecessor holds an earlier ticket: it is running
        __builtin_amdgcn_s_sleep(2);
        v[q] = vx_ld(status + (long long)(t - q) * stride + col);
        if (++spins > VX_SPIN_LIMIT) {
          atomicExch(err, 2);
          v[q] = 2u << 30;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q < nb && !done) {
        excl += v[q] & 0x3fffffffu;
        if ((v[q] >> 30) == 2u) done = true;
      }
    }
    t -= nb;
  }
  return excl;
}

// one pass of the stable LSD radix sort on digit position P_: KPT keys per thread, tile = 256 * KPT keys.
// Order inside a tile: wave w owns elements [w * 64 * KPT, (w + 1) * 64 * KPT), visited in KPT rounds of 64 consecutive elements.
template <int KPT>
__global__ __launch_bounds__(VX_T) void voxel_sort_pass_kernel(VxCtrl* __restrict__ ctrl, int pos, int N, unsigned long long* __restrict__ kA,
                                                               unsigned long long* __restrict__ kB, int* __restrict__ iA,
                                                               int* __restrict__ iB, unsigned* __restrict__ status) {
  if (!ctrl->live[pos]) return;
  constexpr int TILE = VX_T * KPT;
  __shared__ int cnt[4][256];                // per-wave running digit counts, then per-wave exclusive offsets
  __shared__ unsigned goff[256];             // global offset of the tile's first key of every digit
  __shared__ int tile_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) tile_s = atomicAdd(&ctrl->ticket[pos], 1);
  for (int i = tid; i < 4 * 256; i += VX_T) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const int tile = tile_s;
  const int sel = ctrl->src_sel[pos];
  const unsigned long long* __restrict__ ksrc = sel ? kB : kA;
  const int* __restrict__ isrc = sel ? iB : iA;
  unsigned long long* __restrict__ kdst = sel ? kA : kB;
  int* __restrict__ idst = sel ? iA : iB;
  const long long e0 = (long long)tile * TILE + wave * 64 * KPT + lane;

  unsigned long long key[KPT];
  int idx[KPT], dig[KPT], rank[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const long long e = e0 + j * 64;
    const bool ok = e < N;
    key[j] = ok ? ksrc[e] : 0ull;
    idx[j] = ok ? isrc[e] : 0;
    dig[j] = vx_digit(key[j], pos);
  }
  const unsigned long long lt = (1ull << lane) - 1ull;
  volatile int* wc = cnt[wave];
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const bool ok = (e0 + j * 64) < N;
    unsigned long long peers = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (dig[j] >> b) & 1;
      const unsigned long long m = __ballot(bit && ok);
      peers &= bit ? m : ~m;
    }
    const int before = wc[dig[j]];                              // every lane reads the running count ...
    rank[j] = before + __popcll(peers & lt);
    __builtin_amdgcn_wave_barrier();
    if (ok && (peers & lt) == 0ull) wc[dig[j]] = before + __popcll(peers);   // ... then the lowest lane of each digit group advances it
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {   // thread d: the tile's count of digit d, per-wave exclusive offsets in place, publish, look back
    const int d = tid;
    const int c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
    cnt[0][d] = 0; cnt[1][d] = c0; cnt[2][d] = c0 + c1; cnt[3][d] = c0 + c1 + c2;
    const unsigned total = (unsigned)(c0 + c1 + c2 + c3);
    unsigned* mine = status + (long long)tile * 256 + d;
    if (tile == 0) {
      vx_st(mine, (2u << 30) | total);
      goff[d] = (unsigned)ctrl->base[pos][d];
    } else {
      vx_st(mine, (1u << 30) | total);
      const unsigned excl = vx_lookback(status, tile, 256, d, &ctrl->err);
      vx_st(mine, (2u << 30) | (excl + total));
      goff[d] = (unsigned)ctrl->base[pos][d] + excl;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    if ((e0 + j * 64) < N) {
      const long long dst = (long long)goff[dig[j]] + cnt[wave][dig[j]] + rank[j];
      kdst[dst] = key[j];
      idst[dst] = idx[j];
    }
  }
}

// segment heads of the sorted keys -> seg_start[], n_out.  Tile = 256 threads x 8 consecutive keys.
__global__ __launch_bounds__(VX_T) void voxel_heads_kernel(VxCtrl* __restrict__ ctrl, const unsigned long long* __restrict__ kA,
                                                           const unsigned long long* __restrict__ kB, int N, int n_tiles,
                                                           unsigned* __restrict__ status, int* __restrict__ seg_start,
                                                           int* __restrict__ n_out) {
  constexpr int PT = VX_HEAD_PER_THREAD, TILE = VX_T * PT;
  __shared__ int wsum[4];
  __shared__ int tile_s;
  __shared__ unsigned excl_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) tile_s = atomicAdd(&ctrl->ticket[VX_POS], 1);
  __syncthreads();
  const int tile = tile_s;
  const unsigned long long* __restrict__ keys = ctrl->final_sel ? kB : kA;
  const long long i0 = (long long)tile * TILE + (long long)tid * PT;
  unsigned long long k[PT + 1];
  k[0] = (i0 > 0 && i0 - 1 < N) ? keys[i0 - 1] : 0ull;
#pragma unroll
  for (int j = 0; j < PT; ++j) k[j + 1] = (i0 + j < N) ? keys[i0 + j] : 0ull;
  int flag[PT], c = 0;
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    flag[j] = (i0 + j < N) && (i0 + j == 0 || k[j + 1] != k[j]);
    c += flag[j];
  }
  int v = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  int off = v - c;
  for (int w = 0; w < wave; ++w) off += wsum[w];
  const unsigned total = (unsigned)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
  if (tid == 0) {
    unsigned excl = 0;
    if (tile == 0) {
      vx_st(status, (2u << 30) | total);
    } else {
      vx_st(status + tile, (1u << 30) | total);
      excl = vx_lookback(status, tile, 1, 0, &ctrl->err);
      vx_st(status + tile, (2u << 30) | (excl + total));
    }
    excl_s = excl;
    if (tile == n_tiles - 1) {
      seg_start[excl + total] = N;
      *n_out = (int)(excl + total);
    }
  }
  __syncthreads();
  int o = (int)excl_s + off;
#pragma unroll
  for (int j = 0; j < PT; ++j)
    if (flag[j]) seg_start[o++] = (int)(i0 + j);
}

__global__ __launch_bounds__(256) void voxel_reduce_kernel(const float* __restrict__ xyz, const int* __restrict__ labels,
                                                           const VxCtrl* __restrict__ ctrl, const int* __restrict__ iA,
                                                           const int* __restrict__ iB, const int* __restrict__ seg_start,
                                                           const int* __restrict__ n_out, int n_labels, float* __restrict__ centroids,
                                                           int* __restrict__ counts, int* __restrict__ majority) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= *n_out) return;
  const int* __restrict__ sorted_idx = ctrl->final_sel ? iB : iA;
  const int s = seg_start[v], e = seg_start[v + 1];
  double sx = 0.0, sy = 0.0, sz = 0.0;
  int hist[32];
#pragma unroll
  for (int l = 0; l < 32; ++l) hist[l] = 0;
  for (int t = s; t < e; ++t) {
    const int i = sorted_idx[t];
    sx += (double)xyz[3 * i]; sy += (double)xyz[3 * i + 1]; sz += (double)xyz[3 * i + 2];
    if (labels) {
      const int l = labels[i];
      if (l >= 0 && l < n_labels) hist[l]++;
    }
  }
  const double inv = 1.0 / (double)(e - s);
  centroids[3 * v] = (float)(sx * inv);
  centroids[3 * v + 1] = (float)(sy * inv);
  centroids[3 * v + 2] = (float)(sz * inv);
  if (counts) counts[v] = e - s;
  if (majority) {
    int best = 0, bl = labels ? 0 : -1;
    if (labels)
      for (int l = 0; l < n_labels; ++l)
        if (hist[l] > best) { best = hist[l]; bl = l; }
    majority[v] = bl;
  }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct VoxelLayout {
  int kpt, tiles, head_tiles;
  size_t ctrl, status, status_bytes, head_status, clear_bytes, keys_a, keys_b, idx_a, idx_b, seg, total;
};

static VoxelLayout voxel_layout(int N) {
  VoxelLayout L;
  L.kpt = N <= (1 << 18) ? 4 : 16;                 // small scans: more tiles than keys per thread (the passes are latency bound)
  L.tiles = cdiv(N, VX_T * L.kpt);
  L.head_tiles = cdiv(N, VX_T * VX_HEAD_PER_THREAD);
  size_t off = 0;
  L.ctrl = off; off += align256(sizeof(VxCtrl));
  L.status = off; L.status_bytes = (size_t)VX_POS * L.tiles * 256 * 4; off += align256(L.status_bytes);
  L.head_status = off; off += align256((size_t)L.head_tiles * 4);
  L.clear_bytes = off;                              // everything up to here is cleared by every call
  L.keys_a = off; off += align256((size_t)N * 8);
  L.keys_b = off; off += align256((size_t)N * 8);
  L.idx_a = off; off += align256((size_t)N * 4);
  L.idx_b = off; off += align256((size_t)N * 4);
  L.seg = off; off += align256((size_t)(N + 1) * 4);
  L.total = off;
  return L;
}

size_t voxel_workspace_bytes(int N) { return N > 0 ? voxel_layout(N).total : 0; }

int voxel_downsample(const float* xyz, const int* labels, int N, const float* leaf, const float* origin, int n_labels,
                     float* centroids, int* counts, int* majority, int* n_out, void* ws, size_t ws_bytes, hipStream_t st) {
  PN_CHECK_ARG(xyz && leaf && origin && centroids && n_out, "pn_voxel_downsample: null pointer");
  PN_CHECK_ARG(N > 0 && N <= (1 << 30), "pn_voxel_downsample: N must be in [1, 2^30] (N=%d)", N);
  PN_CHECK_ARG(leaf[0] > 0.f && leaf[1] > 0.f && leaf[2] > 0.f, "pn_voxel_downsample: leaf sizes must be positive");
  PN_CHECK_ARG(n_labels >= 0 && n_labels <= 32, "pn_voxel_downsample: n_labels %d outside [0,32]", n_labels);
  const VoxelLayout L = voxel_layout(N);
  PN_CHECK_ARG(ws && ws_bytes >= L.total, "pn_voxel_downsample: workspace too small (%zu < %zu)", ws_bytes, L.total);
  PN_CHECK_ARG((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "pn_voxel_downsample: workspace must be 16-byte aligned");
  char* w = reinterpret_cast<char*>(ws);
  VxCtrl* ctrl = reinterpret_cast<VxCtrl*>(w + L.ctrl);
  unsigned long long* kA = reinterpret_cast<unsigned long long*>(w + L.keys_a);
  unsigned long long* kB = reinterpret_cast<unsigned long long*>(w + L.keys_b);
  int* iA = reinterpret_cast<int*>(w + L.idx_a);
  int* iB = reinterpret_cast<int*>(w + L.idx_b);
  int* seg = reinterpret_cast<int*>(w + L.seg);
  unsigned* status = reinterpret_cast<unsigned*>(w + L.status);
  unsigned* hstatus = reinterpret_cast<unsigned*>(w + L.head_status);
  PN_TRY(zero_fill(reinterpret_cast<float*>(w), (long long)(L.clear_bytes / 4), st));
  const int kb = cdiv(N, VX_T * 4);
  hipLaunchKernelGGL(voxel_keys_kernel, dim3(kb < 1024 ? kb : 1024), dim3(VX_T), 0, st, xyz, N, leaf[0], leaf[1], leaf[2], origin[0],
                     origin[1], origin[2], kA, iA, ctrl);
  PN_CHECK_LAUNCH();
  hipLaunchKernelGGL(voxel_plan_kernel, dim3(1), dim3(256), 0, st, ctrl, N);
  PN_CHECK_LAUNCH();
  for (int p = 0; p < VX_POS; ++p) {
    unsigned* sp = status + (size_t)p * L.tiles * 256;
    if (L.kpt == 4) hipLaunchKernelGGL((voxel_sort_pass_kernel<4>), dim3(L.tiles), dim3(VX_T), 0, st, ctrl, p, N, kA, kB, iA, iB, sp);
    else hipLaunchKernelGGL((voxel_sort_pass_kernel<16>), dim3(L.tiles), dim3(VX_T), 0, st, ctrl, p, N, kA, kB, iA, iB, sp);
    PN_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(voxel_heads_kernel, dim3(L.head_tiles), dim3(VX_T), 0, st, ctrl, kA, kB, N, L.head_tiles, hstatus, seg, n_out);
  PN_CHECK_LAUNCH();
  hipLaunchKernelGGL(voxel_reduce_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, xyz, labels, ctrl, iA, iB, seg, n_out, n_labels,
                     centroids, counts, majority);
  PN_CHECK_LAUNCH();
  return PN_OK;
}

int voxel_error_flag_offset() { return 0; }

}  // namespace pn
