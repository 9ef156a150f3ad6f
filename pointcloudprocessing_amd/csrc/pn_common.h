// Shared declarations for libpointnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/pointnet_hip.h"

namespace pn {

// thread-local last-error text, read through pn_last_error()
void set_error(const char* fmt, ...);
const char* get_error();

#define PN_CHECK_ARG(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) {                                                \
      ::pn::set_error(__VA_ARGS__);                               \
      return PN_ERR_INVALID_ARGUMENT;                             \
    }                                                             \
  } while (0)

#define PN_CHECK_LAUNCH()                                                        \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      ::pn::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,             \
                      hipGetErrorString(e__));                                   \
      return PN_ERR_LAUNCH;                                                      \
    }                                                                            \
  } while (0)

#define PN_TRY(expr)                  \
  do {                                \
    int rc__ = (expr);                \
    if (rc__ != PN_OK) return rc__;   \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }

// ---- device helpers --------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Lower clamp (ReLU when lo = 0) and two-sided clip that keep a NaN a NaN, as tf.nn.relu / tf.clip_by_value do: v_max_f32 /
// fmaxf return the OTHER operand for a NaN, which would turn a diverged (NaN) activation into a silent zero and leave the loss finite.
__device__ __forceinline__ float clamp_lo(float v, float lo) { return v < lo ? lo : v; }
__device__ __forceinline__ float clip_nan(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
// ---- 16-bit storage of the per-point tensors (pn_operand.h16 / PN_STORE_BF16): the pointer is typed float* in every signature and
// reinterpreted here; indices are ELEMENT indices either way ----
__device__ __forceinline__ float bf16_bits_f32(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ unsigned short f32_bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }   // RNE
__device__ __forceinline__ float act_load(const float* p, long long i, int h16) {
  return h16 ? bf16_bits_f32(reinterpret_cast<const unsigned short*>(p)[i]) : p[i];
}
__device__ __forceinline__ void act_store(float* p, long long i, int h16, float v) {
  if (h16) reinterpret_cast<unsigned short*>(p)[i] = f32_bf16_bits(v);
  else p[i] = v;
}
// compile-time forms: a run of loads behind a RUN-TIME storage flag compiles to a branch per load (the loads no longer fly together),
// so kernels switch once, outside their loops:  act_switch(flag, [&](auto h) { ... act_ld<h.value>(p, i) ... });
template <bool H16>
__device__ __forceinline__ float act_ld(const float* p, long long i) {
  if constexpr (H16) return bf16_bits_f32(reinterpret_cast<const unsigned short*>(p)[i]);
  else return p[i];
}
template <bool H16>
__device__ __forceinline__ void act_st(float* p, long long i, float v) {
  if constexpr (H16) reinterpret_cast<unsigned short*>(p)[i] = f32_bf16_bits(v);
  else p[i] = v;
}
template <bool V>
struct BoolTag { static constexpr bool value = V; };
template <class F>
__device__ __forceinline__ void act_switch(int h16, F&& f) {
  if (h16) f(BoolTag<true>{});
  else f(BoolTag<false>{});
}
// eight consecutive elements from a 16-byte aligned position of a bf16 array
__device__ __forceinline__ uint4 act_load8_raw(const float* p, long long i) {
  return *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p) + i);
}
__device__ __forceinline__ void bf16x8_unpack(const uint4& t, float (&v)[8]) {
  v[0] = __builtin_bit_cast(float, t.x << 16); v[1] = __builtin_bit_cast(float, t.x & 0xffff0000u);
  v[2] = __builtin_bit_cast(float, t.y << 16); v[3] = __builtin_bit_cast(float, t.y & 0xffff0000u);
  v[4] = __builtin_bit_cast(float, t.z << 16); v[5] = __builtin_bit_cast(float, t.z & 0xffff0000u);
  v[6] = __builtin_bit_cast(float, t.w << 16); v[7] = __builtin_bit_cast(float, t.w & 0xffff0000u);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace pn
