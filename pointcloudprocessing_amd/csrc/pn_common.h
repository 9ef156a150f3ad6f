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
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace pn
