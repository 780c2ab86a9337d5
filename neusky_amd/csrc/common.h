// Shared device/host helpers for the neusky_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NSKY_OK 0
#define NSKY_ERR_ARG -1
#define NSKY_ERR_LAUNCH -2

void nsky_set_error(const char* fmt, ...);

#define NSKY_CHECK_ARG(cond, ...)                \
  do {                                           \
    if (!(cond)) {                               \
      nsky_set_error(__VA_ARGS__);               \
      return NSKY_ERR_ARG;                       \
    }                                            \
  } while (0)

#define NSKY_CHECK_LAUNCH(name)                                             \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      nsky_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return NSKY_ERR_LAUNCH;                                               \
    }                                                                       \
  } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
