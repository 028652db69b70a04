// Streaming per-(sample, channel) affine kernels of the hot path -- the HBM-bound half of the fused blocks:
//
//   InstanceNorm3d(affine) + LeakyReLU apply / gradient   nnU-Net ConvDropoutNormReLU (dynamic_network_architectures
//                                                         ==0.3.1, built at train_all.py:474-493)
//   SEGating scale (+ residual) + ReLU / LeakyReLU, gradient   models/FLAVR/resnet_3D.py:112-116,144-149,
//                                                              models/FLAVR/FLAVR_arch.py:188-200
//
// Layout: NDHWC, fp32 or (mixed-precision path, *_bf16 entry points) bf16 activations with fp32 arithmetic and fp32 /
// fp64 statistics; a row = one voxel's C channels, a thread moves 16 bytes (4 fp32 / 8 bf16 channels) per access.  A block owns a contiguous run of rows of ONE sample
// (grid.y = sample), a thread owns ONE channel quad for the whole run: the per-(sample, channel) constants
// (mean, rstd, gamma, beta, gate, ...) are folded into 2-3 registers per channel before the loop, the loop
// itself is 16-byte loads / stores with no index arithmetic beyond a pointer bump -- UNROLL independent rows in
// flight per thread -- and 2-8 VALU operations per element.  (The first version of these kernels spent its
// time in 64-bit div/mod per element and ran at ~3 TB/s)
// Column sums are kept in fp32 over a short run of rows and folded into fp64 per thread, reduced over the block
// in LDS, and finished with one double atomic per (quantity, channel) and block: they cancel heavily.
#include "common.h"

namespace {

constexpr int SW_THREADS = 256;
constexpr int SW_UNROLL = 4;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 16-byte vector of activations, widened to fp32 in registers
template <typename T> struct V16;
template <> struct V16<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void ld(const float* p, float (&v)[4]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t[e];
  }
  __device__ static __forceinline__ void st(float* p, const float (&v)[4]) {
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = v[e];
    *reinterpret_cast<f32x4*>(p) = t;
  }
};
template <> struct V16<__bf16> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void ld(const __bf16* p, float (&v)[8]) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
  }
  __device__ static __forceinline__ void st(__bf16* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x8*>(p) = t;
  }
};

struct Span {
  int64_t r0, r1;   // rows [r0, r1) of the whole tensor (sample offset included)
  int c;            // first channel of this thread's group
  int rl, rpp;      // row lane, rows per pass
  int cgn;          // channel groups per row
  bool active;
};

template <int CPT>
__device__ __forceinline__ Span make_span(int64_t S, int C, int64_t rows_per_block) {
  Span s;
  s.cgn = C / CPT;
  s.rpp = SW_THREADS / s.cgn;
  const int tid = threadIdx.x;
  s.active = tid < s.rpp * s.cgn;
  s.c = (tid % s.cgn) * CPT;
  s.rl = tid / s.cgn;
  const int64_t b0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t b1 = b0 + rows_per_block;
  if (b1 > S) b1 = S;
  const int64_t base = (int64_t)blockIdx.y * S;
  s.r0 = base + b0;
  s.r1 = base + b1;
  return s;
}

// act(v) for act in {none, relu, lrelu} with one multiply + select (slope_eff = 0 for relu, 1 for none)
__device__ __forceinline__ float act_sel(float v, float slope_eff) { return v > 0.f ? v : v * slope_eff; }
inline float slope_eff_of(int act, float slope) {
  return act == REHR_ACT_RELU ? 0.f : (act == REHR_ACT_LRELU ? slope : 1.f);
}

// block-level reduction of NQ x CPT per-thread double partials -> one double atomic per (quantity, channel)
template <int NQ, int CPT>
__device__ __forceinline__ void block_column_atomics(const Span& s, double (&acc)[NQ][CPT], double* out,
                                                     int out_stride) {
  __shared__ double red[SW_THREADS * NQ * CPT];
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int e = 0; e < CPT; ++e) red[(q * CPT + e) * SW_THREADS + tid] = s.active ? acc[q][e] : 0.0;
  __syncthreads();
  for (int o = tid; o < NQ * CPT * s.cgn; o += SW_THREADS) {
    const int cg = o % s.cgn, qe = o / s.cgn;
    double t = 0.0;
    for (int k = 0; k < s.rpp; ++k) t += red[qe * SW_THREADS + k * s.cgn + cg];
    atomicAdd(out + (int64_t)(cg * CPT + (qe % CPT)) * out_stride + (qe / CPT), t);
  }
}

// Row loop shared by every kernel: UNROLL rows' loads are issued before any of them is used.
//   LOAD(r, u)  fills the u-th register set from row r;  USE(r, u) consumes it.
#define SW_ROW_LOOP(LOAD, USE)                                                   \
  {                                                                              \
    const int64_t step = s.rpp;                                                  \
    int64_t r = s.r0 + s.rl;                                                     \
    for (; r + (SW_UNROLL - 1) * step < s.r1; r += SW_UNROLL * step) {           \
      _Pragma("unroll") for (int u = 0; u < SW_UNROLL; ++u) { LOAD(r + u * step, u); } \
      _Pragma("unroll") for (int u = 0; u < SW_UNROLL; ++u) { USE(r + u * step, u); }  \
    }                                                                            \
    for (; r < s.r1; r += step) {                                                \
      LOAD(r, 0);                                                                \
      USE(r, 0);                                                                 \
    }                                                                            \
  }

// ---------------------------------------------------------------- InstanceNorm + activation, forward
// y = act((x - mean) * rstd * gamma + beta) = act(x * sc + sh); mean / rstd from the conv epilogue's {sum, sum^2}.
template <typename T>
__global__ __launch_bounds__(SW_THREADS) void instnorm_act_fwd_kernel(
    const T* __restrict__ x, int ldx, const double* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, T* __restrict__ y, int ldy, float* __restrict__ mr, int64_t S, int C,
    int64_t rows_per_block, double invS, float eps, float se) {
  constexpr int CPT = V16<T>::N;
  const Span s = make_span<CPT>(S, C, rows_per_block);
  if (!s.active) return;
  const int n = blockIdx.y;
  float sc[CPT], sh[CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) {
    const double* st = stats + ((int64_t)n * C + s.c + e) * 2;
    const double m = st[0] * invS;
    double var = st[1] * invS - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0 && s.rl == 0) {   // saved for the backward pass
      mr[((int64_t)n * C + s.c + e) * 2] = mean;
      mr[((int64_t)n * C + s.c + e) * 2 + 1] = rstd;
    }
    sc[e] = rstd * gamma[s.c + e];
    sh[e] = beta[s.c + e] - mean * sc[e];
  }
  float v[SW_UNROLL][CPT];
#define LD_(r, u) V16<T>::ld(x + (r) * ldx + s.c, v[u])
#define US_(r, u)                                                                         \
  {                                                                                       \
    _Pragma("unroll") for (int e = 0; e < CPT; ++e) v[u][e] = act_sel(fmaf(v[u][e], sc[e], sh[e]), se); \
    V16<T>::st(y + (r) * ldy + s.c, v[u]);                                                \
  }
  SW_ROW_LOOP(LD_, US_)
#undef LD_
#undef US_
}

// ---------------------------------------------------------------- InstanceNorm + activation, backward
// pass 1: red[n][c] = { sum dz, sum dz * xhat },  dz = dy * act'(xhat * gamma + beta)
template <typename T>
__global__ __launch_bounds__(SW_THREADS) void instnorm_bwd_reduce_kernel(
    const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx, const float* __restrict__ mr,
    const float* __restrict__ gamma, const float* __restrict__ beta, double* __restrict__ red, int64_t S, int C,
    int64_t rows_per_block, float ga) {   // ga = act'(negative side): slope, 0 (relu) or 1 (none)
  constexpr int CPT = V16<T>::N;
  const Span s = make_span<CPT>(S, C, rows_per_block);
  const int n = blockIdx.y;
  double acc[2][CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) acc[0][e] = acc[1][e] = 0.0;
  if (s.active) {
    float rs[CPT], ms[CPT], g[CPT], b[CPT];
#pragma unroll
    for (int e = 0; e < CPT; ++e) {
      const float* m = mr + ((int64_t)n * C + s.c + e) * 2;
      rs[e] = m[1];
      ms[e] = m[0] * m[1];
      g[e] = gamma[s.c + e];
      b[e] = beta[s.c + e];
    }
    float dv[SW_UNROLL][CPT], xv[SW_UNROLL][CPT];
    float a0[CPT], a1[CPT];   // fp32 over one unrolled pass, then folded into fp64
#define LD_(r, u)                           \
  {                                         \
    V16<T>::ld(dy + (r) * lddy + s.c, dv[u]); \
    V16<T>::ld(x + (r) * ldx + s.c, xv[u]);   \
  }
#define US_(r, u)                                                              \
  {                                                                            \
    if (u == 0) { _Pragma("unroll") for (int e = 0; e < CPT; ++e) a0[e] = a1[e] = 0.f; } \
    _Pragma("unroll") for (int e = 0; e < CPT; ++e) {                          \
      const float xh = fmaf(xv[u][e], rs[e], -ms[e]);                          \
      const float dz = fmaf(xh, g[e], b[e]) > 0.f ? dv[u][e] : dv[u][e] * ga;  \
      a0[e] += dz;                                                             \
      a1[e] = fmaf(dz, xh, a1[e]);                                             \
    }                                                                          \
  }
    // (USE runs for u = 0..UNROLL-1 of one pass back to back, so a0/a1 hold that pass; the tail loop has u = 0 only)
    const int64_t step = s.rpp;
    int64_t r = s.r0 + s.rl;
    for (; r + (SW_UNROLL - 1) * step < s.r1; r += SW_UNROLL * step) {
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) LD_(r + u * step, u)
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) US_(r + u * step, u)
#pragma unroll
      for (int e = 0; e < CPT; ++e) {
        acc[0][e] += (double)a0[e];
        acc[1][e] += (double)a1[e];
      }
    }
    for (; r < s.r1; r += step) {
      LD_(r, 0)
      US_(r, 0)
#pragma unroll
      for (int e = 0; e < CPT; ++e) {
        acc[0][e] += (double)a0[e];
        acc[1][e] += (double)a1[e];
      }
    }
#undef LD_
#undef US_
  }
  block_column_atomics<2, CPT>(s, acc, red + (int64_t)n * C * 2, 2);
}

// dgamma[c] = sum_n red[n][c][1]; dbeta[c] = sum_n red[n][c][0]
__global__ void instnorm_bwd_params_kernel(const double* __restrict__ red, float* __restrict__ dgamma,
                                           float* __restrict__ dbeta, int N, int C) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
    double a = 0.0, b = 0.0;
    for (int n = 0; n < N; ++n) {
      b += red[((int64_t)n * C + c) * 2];
      a += red[((int64_t)n * C + c) * 2 + 1];
    }
    dgamma[c] = (float)a;
    dbeta[c] = (float)b;
  }
}

// pass 2: dx = rstd * gamma * (dz - m1 - xhat * m2),  m1 = sum dz / S, m2 = sum dz*xhat / S
// DSUM: also dsum[c] += sum over the block's rows of the dx written -- the gradient of the convolution's bias in front
// of the normalisation (analytically zero; the reference computes the rounding residue, so does this).  On the
// mixed-precision path the weight-gradient kernels have no fused bias column, and a separate column-sum pass per layer
// was 0.74 ms of a cfg-5 step.
template <typename T, bool DSUM>
__global__ __launch_bounds__(SW_THREADS) void instnorm_bwd_apply_kernel(
    const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx, const float* __restrict__ mr,
    const float* __restrict__ gamma, const float* __restrict__ beta, const double* __restrict__ red,
    T* __restrict__ dx, int lddx, int64_t S, int C, int64_t rows_per_block, double invS, float ga,
    double* __restrict__ dsum) {
  constexpr int CPT = V16<T>::N;
  const Span s = make_span<CPT>(S, C, rows_per_block);
  double dacc[1][CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) dacc[0][e] = 0.0;
  if (s.active) {
  const int n = blockIdx.y;
  float rs[CPT], ms[CPT], g[CPT], b[CPT], k0[CPT], m1[CPT], m2[CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) {
    const float* m = mr + ((int64_t)n * C + s.c + e) * 2;
    const double* rd = red + ((int64_t)n * C + s.c + e) * 2;
    rs[e] = m[1];
    ms[e] = m[0] * m[1];
    g[e] = gamma[s.c + e];
    b[e] = beta[s.c + e];
    k0[e] = m[1] * g[e];
    m1[e] = (float)(rd[0] * invS);
    m2[e] = (float)(rd[1] * invS);
  }
  float dv[SW_UNROLL][CPT], xv[SW_UNROLL][CPT];
#define LD_(r, u)                           \
  {                                         \
    V16<T>::ld(dy + (r) * lddy + s.c, dv[u]); \
    V16<T>::ld(x + (r) * ldx + s.c, xv[u]);   \
  }
#define US_(r, u)                                                              \
  {                                                                            \
    _Pragma("unroll") for (int e = 0; e < CPT; ++e) {                          \
      const float xh = fmaf(xv[u][e], rs[e], -ms[e]);                          \
      const float dz = fmaf(xh, g[e], b[e]) > 0.f ? dv[u][e] : dv[u][e] * ga;  \
      dv[u][e] = k0[e] * (dz - m1[e] - xh * m2[e]);                            \
      if (DSUM) fs[e] += dv[u][e];                                             \
    }                                                                          \
    V16<T>::st(dx + (r) * lddx + s.c, dv[u]);                                  \
  }
  float fs[CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) fs[e] = 0.f;
  {
    // (the row loop of SW_ROW_LOOP, with the fp32 partials folded into fp64 after every unrolled pass)
    const int64_t step = s.rpp;
    int64_t r = s.r0 + s.rl;
    for (; r + (SW_UNROLL - 1) * step < s.r1; r += SW_UNROLL * step) {
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) LD_(r + u * step, u)
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) US_(r + u * step, u)
      if (DSUM) {
#pragma unroll
        for (int e = 0; e < CPT; ++e) { dacc[0][e] += (double)fs[e]; fs[e] = 0.f; }
      }
    }
    for (; r < s.r1; r += step) {
      LD_(r, 0)
      US_(r, 0)
    }
    if (DSUM) {
#pragma unroll
      for (int e = 0; e < CPT; ++e) dacc[0][e] += (double)fs[e];
    }
  }
#undef LD_
#undef US_
  }
  if (DSUM) block_column_atomics<1, CPT>(s, dacc, dsum, 1);
}

// dconv_bias[c] = dsum[c]
__global__ void dsum_finish_kernel(const double* __restrict__ dsum, float* __restrict__ out, int C) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) out[c] = (float)dsum[c];
}

// ---------------------------------------------------------------- y = act(x * gate + res)
template <typename T, bool RES>
__global__ __launch_bounds__(SW_THREADS) void scale_res_act_fwd_kernel(
    const T* __restrict__ x, int ldx, const float* __restrict__ gate, const T* __restrict__ res, int ldr,
    T* __restrict__ y, int ldy, int64_t S, int C, int64_t rows_per_block, float se) {
  constexpr int CPT = V16<T>::N;
  const Span s = make_span<CPT>(S, C, rows_per_block);
  if (!s.active) return;
  float gv[CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) gv[e] = gate[(int64_t)blockIdx.y * C + s.c + e];
  float v[SW_UNROLL][CPT], q[SW_UNROLL][CPT];
#define LD_(r, u)                                      \
  {                                                    \
    V16<T>::ld(x + (r) * ldx + s.c, v[u]);             \
    if (RES) V16<T>::ld(res + (r) * ldr + s.c, q[u]);  \
  }
#define US_(r, u)                                                                             \
  {                                                                                           \
    _Pragma("unroll") for (int e = 0; e < CPT; ++e)                                           \
        v[u][e] = act_sel(RES ? fmaf(v[u][e], gv[e], q[u][e]) : v[u][e] * gv[e], se);         \
    V16<T>::st(y + (r) * ldy + s.c, v[u]);                                                    \
  }
  SW_ROW_LOOP(LD_, US_)
#undef LD_
#undef US_
}

// dz = dy * act'(y); dres = dz; dx = dz * gate; dgate_acc[n][c] += sum dz * x
template <typename T, bool DRES>
__global__ __launch_bounds__(SW_THREADS) void scale_res_act_bwd_kernel(
    const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy, const T* __restrict__ x, int ldx,
    const float* __restrict__ gate, T* __restrict__ dx, int lddx, T* __restrict__ dres, int lddr,
    double* __restrict__ dgate_acc, int64_t S, int C, int64_t rows_per_block, float ga) {
  constexpr int CPT = V16<T>::N;
  const Span s = make_span<CPT>(S, C, rows_per_block);
  const int n = blockIdx.y;
  double acc[1][CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) acc[0][e] = 0.0;
  if (s.active) {
    float gv[CPT];
#pragma unroll
    for (int e = 0; e < CPT; ++e) gv[e] = gate[(int64_t)n * C + s.c + e];
    float dv[SW_UNROLL][CPT], yv[SW_UNROLL][CPT], xv[SW_UNROLL][CPT], a[CPT];
#define LD_(r, u)                             \
  {                                           \
    V16<T>::ld(dy + (r) * lddy + s.c, dv[u]); \
    V16<T>::ld(y + (r) * ldy + s.c, yv[u]);   \
    V16<T>::ld(x + (r) * ldx + s.c, xv[u]);   \
  }
#define US_(r, u)                                                            \
  {                                                                          \
    if (u == 0) { _Pragma("unroll") for (int e = 0; e < CPT; ++e) a[e] = 0.f; } \
    _Pragma("unroll") for (int e = 0; e < CPT; ++e) {                        \
      dv[u][e] = yv[u][e] > 0.f ? dv[u][e] : dv[u][e] * ga;                  \
      a[e] = fmaf(dv[u][e], xv[u][e], a[e]);                                 \
    }                                                                        \
    if (DRES) V16<T>::st(dres + (r) * lddr + s.c, dv[u]);                    \
    _Pragma("unroll") for (int e = 0; e < CPT; ++e) dv[u][e] *= gv[e];       \
    V16<T>::st(dx + (r) * lddx + s.c, dv[u]);                                \
  }
    const int64_t step = s.rpp;
    int64_t r = s.r0 + s.rl;
    for (; r + (SW_UNROLL - 1) * step < s.r1; r += SW_UNROLL * step) {
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) LD_(r + u * step, u)
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) US_(r + u * step, u)
#pragma unroll
      for (int e = 0; e < CPT; ++e) acc[0][e] += (double)a[e];
    }
    for (; r < s.r1; r += step) {
      LD_(r, 0)
      US_(r, 0)
#pragma unroll
      for (int e = 0; e < CPT; ++e) acc[0][e] += (double)a[e];
    }
#undef LD_
#undef US_
  }
  block_column_atomics<1, CPT>(s, acc, dgate_acc + (int64_t)n * C, 1);
}

// x += k[n][c] in place (the mean-pool branch of the SEGating gradient, known only after the block sums)
template <typename T>
__global__ __launch_bounds__(SW_THREADS) void add_channel_const_kernel(T* __restrict__ x, int ldx,
                                                                      const float* __restrict__ k, int64_t S, int C,
                                                                      int64_t rows_per_block) {
  constexpr int CPT = V16<T>::N;
  const Span s = make_span<CPT>(S, C, rows_per_block);
  if (!s.active) return;
  float kv[CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) kv[e] = k[(int64_t)blockIdx.y * C + s.c + e];
  float v[SW_UNROLL][CPT];
#define LD_(r, u) V16<T>::ld(x + (r) * ldx + s.c, v[u])
#define US_(r, u)                                                          \
  {                                                                        \
    _Pragma("unroll") for (int e = 0; e < CPT; ++e) v[u][e] += kv[e];      \
    V16<T>::st(x + (r) * ldx + s.c, v[u]);                                 \
  }
  SW_ROW_LOOP(LD_, US_)
#undef LD_
#undef US_
}

// column sums: out64[c] += sum over rows of x (bias gradients)
template <typename T>
__global__ __launch_bounds__(SW_THREADS) void channel_sum_kernel(const T* __restrict__ x, int ldx, int64_t S, int C,
                                                                int64_t rows_per_block, double* __restrict__ out64) {
  constexpr int CPT = V16<T>::N;
  const Span s = make_span<CPT>(S, C, rows_per_block);
  double acc[1][CPT];
#pragma unroll
  for (int e = 0; e < CPT; ++e) acc[0][e] = 0.0;
  if (s.active) {
    float v[SW_UNROLL][CPT], a[CPT];
    const int64_t step = s.rpp;
    int64_t r = s.r0 + s.rl;
    for (; r + (SW_UNROLL - 1) * step < s.r1; r += SW_UNROLL * step) {
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) V16<T>::ld(x + (r + u * step) * ldx + s.c, v[u]);
#pragma unroll
      for (int e = 0; e < CPT; ++e) a[e] = (v[0][e] + v[1][e]) + (v[2][e] + v[3][e]);
#pragma unroll
      for (int e = 0; e < CPT; ++e) acc[0][e] += (double)a[e];
    }
    for (; r < s.r1; r += step) {
      V16<T>::ld(x + r * ldx + s.c, v[0]);
#pragma unroll
      for (int e = 0; e < CPT; ++e) acc[0][e] += (double)v[0][e];
    }
  }
  block_column_atomics<1, CPT>(s, acc, out64, 1);
}
__global__ void channel_sum_finish_kernel(const double* __restrict__ s64, float* __restrict__ out, int C, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) out[c] = accumulate ? out[c] + (float)s64[c] : (float)s64[c];
}

// dx = dy * act'(y) over a flat tensor of n16 16-byte groups
template <typename T>
__global__ __launch_bounds__(SW_THREADS) void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                            T* __restrict__ dx, int64_t n16, float ga) {
  constexpr int CPT = V16<T>::N;
  const int64_t stride = (int64_t)gridDim.x * SW_THREADS;
  int64_t i = (int64_t)blockIdx.x * SW_THREADS + threadIdx.x;
  float dv[SW_UNROLL][CPT], yv[SW_UNROLL][CPT];
  for (; i + (SW_UNROLL - 1) * stride < n16; i += SW_UNROLL * stride) {
#pragma unroll
    for (int u = 0; u < SW_UNROLL; ++u) {
      V16<T>::ld(dy + (i + u * stride) * CPT, dv[u]);
      V16<T>::ld(y + (i + u * stride) * CPT, yv[u]);
    }
#pragma unroll
    for (int u = 0; u < SW_UNROLL; ++u) {
#pragma unroll
      for (int e = 0; e < CPT; ++e) dv[u][e] = yv[u][e] > 0.f ? dv[u][e] : dv[u][e] * ga;
      V16<T>::st(dx + (i + u * stride) * CPT, dv[u]);
    }
  }
  for (; i < n16; i += stride) {
    V16<T>::ld(dy + i * CPT, dv[0]);
    V16<T>::ld(y + i * CPT, yv[0]);
#pragma unroll
    for (int e = 0; e < CPT; ++e) dv[0][e] = yv[0][e] > 0.f ? dv[0][e] : dv[0][e] * ga;
    V16<T>::st(dx + i * CPT, dv[0]);
  }
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// rows per block: ~2048 blocks over the whole tensor (8 per CU), at least 8 unrolled passes per block
inline int64_t rows_per_block_for(int64_t S, int C, int N, int cpt) {
  const int rpp = SW_THREADS / (C / cpt);
  int64_t target = 2048 / (N > 0 ? N : 1);
  if (target < 1) target = 1;
  int64_t rpb = (S + target - 1) / target;
  const int64_t min_rows = (int64_t)rpp * SW_UNROLL * 8;
  if (rpb < min_rows) rpb = min_rows;
  const int64_t q = (int64_t)rpp * SW_UNROLL;   // whole unrolled passes: only the sample's last block has a tail
  return (rpb + q - 1) / q * q;
}

inline bool shape_ok(int N, int64_t S, int C, int cpt) {
  return N >= 1 && N <= 65535 && S >= 1 && C >= cpt && C % cpt == 0 && C / cpt <= SW_THREADS;
}

#define ST ((hipStream_t)stream)

template <typename T>
int scale_res_act_fwd_t(const T* x, int ldx, const float* gate, const T* res, int ldr, T* y, int ldy, int N, int64_t S,
                        int C, int act, float slope, void* stream) {
  constexpr int CPT = V16<T>::N;
  if (!x || !gate || !y || !shape_ok(N, S, C, CPT) || ldx % CPT || ldy % CPT || (res && ldr % CPT)) return REHR_EINVAL;
  if (!aligned16(x) || !aligned16(y) || (res && !aligned16(res))) return REHR_EINVAL;
  const int64_t rpb = rows_per_block_for(S, C, N, CPT);
  const dim3 grid((unsigned)((S + rpb - 1) / rpb), N);
  const float se = slope_eff_of(act, slope);
  if (res)
    hipLaunchKernelGGL((scale_res_act_fwd_kernel<T, true>), grid, dim3(SW_THREADS), 0, ST, x, ldx, gate, res, ldr, y, ldy,
                       S, C, rpb, se);
  else
    hipLaunchKernelGGL((scale_res_act_fwd_kernel<T, false>), grid, dim3(SW_THREADS), 0, ST, x, ldx, gate, res, ldr, y, ldy,
                       S, C, rpb, se);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

template <typename T>
int scale_res_act_bwd_t(const T* dy, int lddy, const T* y, int ldy, const T* x, int ldx, const float* gate, T* dx,
                        int lddx, T* dres, int lddr, double* dgate_acc, int N, int64_t S, int C, int act, float slope,
                        void* stream) {
  constexpr int CPT = V16<T>::N;
  if (!dy || !y || !x || !gate || !dx || !dgate_acc || !shape_ok(N, S, C, CPT)) return REHR_EINVAL;
  if (lddy % CPT || ldy % CPT || ldx % CPT || lddx % CPT || (dres && lddr % CPT)) return REHR_EINVAL;
  if (!aligned16(dy) || !aligned16(y) || !aligned16(x) || !aligned16(dx) || (dres && !aligned16(dres))) return REHR_EINVAL;
  const int64_t rpb = rows_per_block_for(S, C, N, CPT);
  const dim3 grid((unsigned)((S + rpb - 1) / rpb), N);
  const float ga = slope_eff_of(act, slope);
  if (dres)
    hipLaunchKernelGGL((scale_res_act_bwd_kernel<T, true>), grid, dim3(SW_THREADS), 0, ST, dy, lddy, y, ldy, x, ldx, gate,
                       dx, lddx, dres, lddr, dgate_acc, S, C, rpb, ga);
  else
    hipLaunchKernelGGL((scale_res_act_bwd_kernel<T, false>), grid, dim3(SW_THREADS), 0, ST, dy, lddy, y, ldy, x, ldx, gate,
                       dx, lddx, dres, lddr, dgate_acc, S, C, rpb, ga);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

template <typename T>
int add_channel_const_t(T* x, int ldx, const float* k, int N, int64_t S, int C, void* stream) {
  constexpr int CPT = V16<T>::N;
  if (!x || !k || !shape_ok(N, S, C, CPT) || ldx % CPT || !aligned16(x)) return REHR_EINVAL;
  const int64_t rpb = rows_per_block_for(S, C, N, CPT);
  hipLaunchKernelGGL(add_channel_const_kernel<T>, dim3((unsigned)((S + rpb - 1) / rpb), N), dim3(SW_THREADS), 0, ST, x,
                     ldx, k, S, C, rpb);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

template <typename T>
int instnorm_act_fwd_t(const T* x, int ldx, const double* stats, const float* gamma, const float* beta, T* y, int ldy,
                       float* mean_rstd, int N, int64_t S, int C, float eps, int act, float slope, void* stream) {
  constexpr int CPT = V16<T>::N;
  if (!x || !stats || !gamma || !beta || !y || !mean_rstd || !shape_ok(N, S, C, CPT)) return REHR_EINVAL;
  if (ldx % CPT || ldy % CPT || !aligned16(x) || !aligned16(y)) return REHR_EINVAL;
  const int64_t rpb = rows_per_block_for(S, C, N, CPT);
  hipLaunchKernelGGL(instnorm_act_fwd_kernel<T>, dim3((unsigned)((S + rpb - 1) / rpb), N), dim3(SW_THREADS), 0, ST, x,
                     ldx, stats, gamma, beta, y, ldy, mean_rstd, S, C, rpb, 1.0 / (double)S, eps,
                     slope_eff_of(act, slope));
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

template <typename T>
int instnorm_act_bwd_t(const T* dy, int lddy, const T* x, int ldx, const float* mean_rstd, const float* gamma,
                       const float* beta, T* dx, int lddx, float* dgamma, float* dbeta, double* red, int N, int64_t S,
                       int C, int act, float slope, void* stream, double* dsum = nullptr, float* dconv_bias = nullptr) {
  constexpr int CPT = V16<T>::N;
  if (!dy || !x || !mean_rstd || !gamma || !beta || !dx || !dgamma || !dbeta || !red) return REHR_EINVAL;
  if (!shape_ok(N, S, C, CPT) || lddy % CPT || ldx % CPT || lddx % CPT) return REHR_EINVAL;
  if (!aligned16(dy) || !aligned16(x) || !aligned16(dx)) return REHR_EINVAL;
  const int64_t rpb = rows_per_block_for(S, C, N, CPT);
  const dim3 grid((unsigned)((S + rpb - 1) / rpb), N);
  const float ga = slope_eff_of(act, slope);
  hipLaunchKernelGGL(instnorm_bwd_reduce_kernel<T>, grid, dim3(SW_THREADS), 0, ST, dy, lddy, x, ldx, mean_rstd, gamma,
                     beta, red, S, C, rpb, ga);
  hipLaunchKernelGGL(instnorm_bwd_params_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, red, dgamma, dbeta, N, C);
  if (dsum != nullptr && dconv_bias != nullptr) {
    if (hipMemsetAsync(dsum, 0, sizeof(double) * C, ST) != hipSuccess) return REHR_EHIP;
    hipLaunchKernelGGL((instnorm_bwd_apply_kernel<T, true>), grid, dim3(SW_THREADS), 0, ST, dy, lddy, x, ldx, mean_rstd,
                       gamma, beta, red, dx, lddx, S, C, rpb, 1.0 / (double)S, ga, dsum);
    hipLaunchKernelGGL(dsum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, dsum, dconv_bias, C);
  } else {
    hipLaunchKernelGGL((instnorm_bwd_apply_kernel<T, false>), grid, dim3(SW_THREADS), 0, ST, dy, lddy, x, ldx, mean_rstd,
                       gamma, beta, red, dx, lddx, S, C, rpb, 1.0 / (double)S, ga, (double*)nullptr);
  }
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

template <typename T>
int channel_sum_t(const T* x, int ldx, int64_t rows, int C, float* out, int accumulate, double* scratch, void* stream) {
  constexpr int CPT = V16<T>::N;
  if (!x || !out || !scratch || !shape_ok(1, rows, C, CPT) || ldx % CPT || !aligned16(x)) return REHR_EINVAL;
  if (hipMemsetAsync(scratch, 0, sizeof(double) * C, ST) != hipSuccess) return REHR_EHIP;
  const int64_t rpb = rows_per_block_for(rows, C, 1, CPT);
  hipLaunchKernelGGL(channel_sum_kernel<T>, dim3((unsigned)((rows + rpb - 1) / rpb), 1), dim3(SW_THREADS), 0, ST, x, ldx,
                     rows, C, rpb, scratch);
  hipLaunchKernelGGL(channel_sum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, scratch, out, C, accumulate);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

template <typename T>
int act_bwd_t(const T* dy, const T* y, T* dx, int64_t n, int act, float slope, void* stream) {
  constexpr int CPT = V16<T>::N;
  if (!dy || !y || !dx || n < CPT || n % CPT || !aligned16(dy) || !aligned16(y) || !aligned16(dx)) return REHR_EINVAL;
  const int64_t n16 = n / CPT;
  int64_t blocks = (n16 + SW_THREADS * SW_UNROLL - 1) / (SW_THREADS * SW_UNROLL);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(act_bwd_kernel<T>, dim3((unsigned)blocks), dim3(SW_THREADS), 0, ST, dy, y, dx, n16,
                     slope_eff_of(act, slope));
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

}  // namespace

#define BF(p) reinterpret_cast<const __bf16*>(p)
#define BFM(p) reinterpret_cast<__bf16*>(p)

extern "C" int rehr_scale_res_act_fwd_f32(const float* x, int32_t ldx, const float* gate, const float* res,
                                          int32_t ldr, float* y, int32_t ldy, int32_t N, int64_t S, int32_t C,
                                          int32_t act, float slope, void* stream) {
  return scale_res_act_fwd_t<float>(x, ldx, gate, res, ldr, y, ldy, N, S, C, act, slope, stream);
}
extern "C" int rehr_scale_res_act_fwd_bf16(const void* x, int32_t ldx, const float* gate, const void* res, int32_t ldr,
                                           void* y, int32_t ldy, int32_t N, int64_t S, int32_t C, int32_t act,
                                           float slope, void* stream) {
  return scale_res_act_fwd_t<__bf16>(BF(x), ldx, gate, BF(res), ldr, BFM(y), ldy, N, S, C, act, slope, stream);
}
extern "C" int rehr_scale_res_act_bwd_f32(const float* dy, int32_t lddy, const float* y, int32_t ldy, const float* x,
                                          int32_t ldx, const float* gate, float* dx, int32_t lddx, float* dres,
                                          int32_t lddr, double* dgate_acc, int32_t N, int64_t S, int32_t C,
                                          int32_t act, float slope, void* stream) {
  return scale_res_act_bwd_t<float>(dy, lddy, y, ldy, x, ldx, gate, dx, lddx, dres, lddr, dgate_acc, N, S, C, act, slope,
                                    stream);
}
extern "C" int rehr_scale_res_act_bwd_bf16(const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x,
                                           int32_t ldx, const float* gate, void* dx, int32_t lddx, void* dres,
                                           int32_t lddr, double* dgate_acc, int32_t N, int64_t S, int32_t C,
                                           int32_t act, float slope, void* stream) {
  return scale_res_act_bwd_t<__bf16>(BF(dy), lddy, BF(y), ldy, BF(x), ldx, gate, BFM(dx), lddx, BFM(dres), lddr,
                                     dgate_acc, N, S, C, act, slope, stream);
}
extern "C" int rehr_add_channel_const_f32(float* x, int32_t ldx, const float* k, int32_t N, int64_t S, int32_t C,
                                          void* stream) {
  return add_channel_const_t<float>(x, ldx, k, N, S, C, stream);
}
extern "C" int rehr_add_channel_const_bf16(void* x, int32_t ldx, const float* k, int32_t N, int64_t S, int32_t C,
                                           void* stream) {
  return add_channel_const_t<__bf16>(BFM(x), ldx, k, N, S, C, stream);
}
extern "C" int rehr_instnorm_act_fwd_f32(const float* x, int32_t ldx, const double* stats, const float* gamma,
                                         const float* beta, float* y, int32_t ldy, float* mean_rstd, int32_t N,
                                         int64_t S, int32_t C, float eps, int32_t act, float slope, void* stream) {
  return instnorm_act_fwd_t<float>(x, ldx, stats, gamma, beta, y, ldy, mean_rstd, N, S, C, eps, act, slope, stream);
}
extern "C" int rehr_instnorm_act_fwd_bf16(const void* x, int32_t ldx, const double* stats, const float* gamma,
                                          const float* beta, void* y, int32_t ldy, float* mean_rstd, int32_t N,
                                          int64_t S, int32_t C, float eps, int32_t act, float slope, void* stream) {
  return instnorm_act_fwd_t<__bf16>(BF(x), ldx, stats, gamma, beta, BFM(y), ldy, mean_rstd, N, S, C, eps, act, slope,
                                    stream);
}
extern "C" int rehr_instnorm_act_bwd_f32(const float* dy, int32_t lddy, const float* x, int32_t ldx,
                                         const float* mean_rstd, const float* gamma, const float* beta, float* dx,
                                         int32_t lddx, float* dgamma, float* dbeta, double* red, int32_t N, int64_t S,
                                         int32_t C, int32_t act, float slope, void* stream) {
  return instnorm_act_bwd_t<float>(dy, lddy, x, ldx, mean_rstd, gamma, beta, dx, lddx, dgamma, dbeta, red, N, S, C, act,
                                   slope, stream);
}
extern "C" int rehr_instnorm_act_bwd_bf16(const void* dy, int32_t lddy, const void* x, int32_t ldx,
                                          const float* mean_rstd, const float* gamma, const float* beta, void* dx,
                                          int32_t lddx, float* dgamma, float* dbeta, double* red, int32_t N, int64_t S,
                                          int32_t C, int32_t act, float slope, void* stream) {
  return instnorm_act_bwd_t<__bf16>(BF(dy), lddy, BF(x), ldx, mean_rstd, gamma, beta, BFM(dx), lddx, dgamma, dbeta, red,
                                    N, S, C, act, slope, stream);
}
extern "C" int rehr_instnorm_act_bwd_dbias_bf16(const void* dy, int32_t lddy, const void* x, int32_t ldx,
                                                const float* mean_rstd, const float* gamma, const float* beta, void* dx,
                                                int32_t lddx, float* dgamma, float* dbeta, double* red, int32_t N,
                                                int64_t S, int32_t C, int32_t act, float slope, double* dsum,
                                                float* dconv_bias, void* stream) {
  if (!dsum || !dconv_bias) return REHR_EINVAL;
  return instnorm_act_bwd_t<__bf16>(BF(dy), lddy, BF(x), ldx, mean_rstd, gamma, beta, BFM(dx), lddx, dgamma, dbeta, red,
                                    N, S, C, act, slope, stream, dsum, dconv_bias);
}
extern "C" int rehr_channel_sum_f32(const float* x, int32_t ldx, int64_t rows, int32_t C, float* out,
                                    int32_t accumulate, double* scratch, void* stream) {
  return channel_sum_t<float>(x, ldx, rows, C, out, accumulate, scratch, stream);
}
extern "C" int rehr_channel_sum_bf16(const void* x, int32_t ldx, int64_t rows, int32_t C, float* out,
                                     int32_t accumulate, double* scratch, void* stream) {
  return channel_sum_t<__bf16>(BF(x), ldx, rows, C, out, accumulate, scratch, stream);
}
extern "C" int rehr_act_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, int32_t act, float slope,
                                void* stream) {
  return act_bwd_t<float>(dy, y, dx, n, act, slope, stream);
}
extern "C" int rehr_act_bwd_bf16(const void* dy, const void* y, void* dx, int64_t n, int32_t act, float slope,
                                 void* stream) {
  return act_bwd_t<__bf16>(BF(dy), BF(y), BFM(dx), n, act, slope, stream);
}
