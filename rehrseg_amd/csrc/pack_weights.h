// Weight packing shared by the fp32 and bf16 entry points (rehr_pack_weights_f32 / _bf16):
//   in[a][b][t] (transpose_ab: in[b][a][t]) fp32, the torch parameter layout  ->  out[t][Apad][B] in TO, rows a >= A zero.
// Runs for every matrix-core layer twice per step (forward panel, input-gradient panel).  A thread per OUTPUT element
// read the source at a stride of T floats (108 bytes for 3x3x3 taps: one useful dword per 128-byte line): 0.61 ms of a
// cfg-2 step for 0.76 GB.  Here a block moves a (p, q, t) tile through LDS: the source is read in runs that are
// contiguous in memory (t fastest, then the inner index q), the destination is written in runs contiguous in b.
#pragma once
#include "common.h"

namespace packw {

__device__ __forceinline__ void put(float* p, float v) { *p = v; }
__device__ __forceinline__ void put(__bf16* p, float v) { *p = (__bf16)v; }

// TR = false: in[a][b][t], tile = 4 a x 64 b;  TR = true: in[b][a][t], tile = 32 b x 8 a.  TT = 32 taps per block.
template <typename TO, bool TR>
__global__ __launch_bounds__(256) void pack_tiled_kernel(const float* __restrict__ in, TO* __restrict__ out, int A, int Apad,
                                                         int B, int T) {
  constexpr int PT = TR ? 32 : 4, QT = TR ? 8 : 64, TT = 32, LQ = TT + 1, LP = QT * LQ + (TR ? 1 : 0);
  __shared__ float tile[PT * LP];
  const int P = TR ? B : A, Q = TR ? A : B;
  const int p0 = blockIdx.y * PT, q0 = blockIdx.x * QT, t0 = blockIdx.z * TT;
  const int tn = min(TT, T - t0), qn = min(QT, Q - q0), pn = min(PT, P - p0);
  const int tid = threadIdx.x;
  // load: t fastest, then q (contiguous in memory when the block holds all taps), then p
  for (int i = tid; i < PT * QT * TT; i += 256) {
    const int t = i % TT, q = (i / TT) % QT, p = i / (TT * QT);
    if (p < pn && q < qn && t < tn) tile[p * LP + q * LQ + t] = in[((int64_t)(p0 + p) * Q + q0 + q) * T + t0 + t];
  }
  __syncthreads();
  if (!TR) {   // out[t][a = p][b = q]: q fastest
    for (int i = tid; i < PT * TT * QT; i += 256) {
      const int q = i % QT, t = (i / QT) % TT, p = i / (QT * TT);
      if (p < pn && q < qn && t < tn) put(out + ((int64_t)(t0 + t) * Apad + p0 + p) * B + q0 + q, tile[p * LP + q * LQ + t]);
    }
  } else {     // out[t][a = q][b = p]: p fastest
    for (int i = tid; i < QT * TT * PT; i += 256) {
      const int p = i % PT, t = (i / PT) % TT, q = i / (PT * TT);
      if (p < pn && q < qn && t < tn) put(out + ((int64_t)(t0 + t) * Apad + q0 + q) * B + p0 + p, tile[p * LP + q * LQ + t]);
    }
  }
  // rows a in [A, Apad): zeros, written by the first tile of every tap chunk
  if (blockIdx.x == 0 && blockIdx.y == 0 && Apad > A) {
    const int64_t n = (int64_t)(Apad - A) * B;
    for (int t = 0; t < tn; ++t)
      for (int64_t i = tid; i < n; i += 256) put(out + ((int64_t)(t0 + t) * Apad + A) * B + i, 0.f);
  }
}

template <typename TO>
int launch(const float* in, TO* out, int A, int Apad, int B, int T, int transpose_ab, hipStream_t stream) {
  const int P = transpose_ab ? B : A, Q = transpose_ab ? A : B;
  const int PT = transpose_ab ? 32 : 4, QT = transpose_ab ? 8 : 64;
  const int64_t gx = (Q + QT - 1) / QT, gy = (P + PT - 1) / PT, gz = (T + 31) / 32;
  if (gy > 65535 || gz > 65535) return REHR_ENOSUP;
  dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)gz);
  if (transpose_ab) hipLaunchKernelGGL((pack_tiled_kernel<TO, true>), grid, dim3(256), 0, stream, in, out, A, Apad, B, T);
  else hipLaunchKernelGGL((pack_tiled_kernel<TO, false>), grid, dim3(256), 0, stream, in, out, A, Apad, B, T);
  return REHR_OK;
}

}  // namespace packw
