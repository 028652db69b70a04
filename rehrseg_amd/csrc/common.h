// Shared device/host helpers for the gfx950 kernels of librehrseg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rehrseg_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Launch bookkeeping.  hipGetLastError() is sticky per thread and may still hold
// an unrelated, harmless error from the host framework's own start-up; every
// launch therefore clears it first and records only its own outcome.
static thread_local int rehr_launch_failed = 0;
extern thread_local int rehr_last_hip_error_code;  // defined in elementwise.hip
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)           \
  do {                                                                                              \
    (void)hipGetLastError();                                                                        \
    kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);              \
    hipError_t e__ = hipGetLastError();                                                             \
    if (e__ != hipSuccess) { rehr_launch_failed = 1; rehr_last_hip_error_code = (int)e__; }         \
  } while (0)
#define REHR_LAUNCH_CHECK()                                   \
  do {                                                        \
    if (rehr_launch_failed) {                                 \
      rehr_launch_failed = 0;                                 \
      return REHR_EHIP;                                       \
    }                                                         \
  } while (0)

// Bijective XCD-aware remap of a 1-D grid: blocks that the dispatcher places on
// the same XCD (b % 8 equal) become consecutive logical ids, so neighbouring
// tiles (shared halo / shared weight panel) hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
  const int q = nb >> 3, r = nb & 7, xcd = b & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == REHR_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == REHR_ACT_LRELU) return v > 0.f ? v : v * slope;
  return v;
}
// derivative factor given the OUTPUT of the activation (slope > 0 keeps sign)
__device__ __forceinline__ float act_grad(float y, int act, float slope) {
  if (act == REHR_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == REHR_ACT_LRELU) return y > 0.f ? 1.f : slope;
  return 1.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// 4 consecutive channels of one voxel in the activations' dtype (fp32, or bf16 on the mixed-precision path)
__device__ __forceinline__ void store4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(__bf16* p, const f32x4& v) {
  typedef __bf16 bf4_t __attribute__((ext_vector_type(4)));
  bf4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
  *reinterpret_cast<bf4_t*>(p) = o;
}
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const __bf16* p) {
  typedef __bf16 bf4_t __attribute__((ext_vector_type(4)));
  const bf4_t o = *reinterpret_cast<const bf4_t*>(p);
  return f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
}
