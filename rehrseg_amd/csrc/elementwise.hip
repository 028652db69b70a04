// HBM-bound kernels of the hot path: SEGating (gate, scale+residual+activation
// and their gradients), InstanceNorm3d+LeakyReLU (apply and gradient), depth-only
// linear upsample, bias-gradient column sums, weight packing and layout edges.
//
// All tensors are NDHWC fp32; every streaming access is a 16-byte load/store per
// lane with the channel index innermost, so a wave touches 1 KiB contiguous.
// Per-(sample, channel) reductions keep the channel quad fixed per thread, reduce
// over a block's rows in registers + LDS, and finish with one double atomic per
// channel per block.
#include "common.h"

namespace {

constexpr int EW_THREADS = 256;
constexpr int EW_MAX_BLOCKS = 2048;  // 256 CUs x 8

inline int ew_blocks(int64_t work_items) {
  int64_t b = (work_items + EW_THREADS - 1) / EW_THREADS;
  if (b > EW_MAX_BLOCKS) b = EW_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

// ---------------------------------------------------------------- pack weights
__global__ void pack_weights_kernel(const float* __restrict__ in, float* __restrict__ out, int A,
                                    int Apad, int B, int T, int transpose_ab) {
  const int64_t total = (int64_t)T * Apad * B;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % B);
    const int64_t r = i / B;
    const int a = (int)(r % Apad);
    const int t = (int)(r / Apad);
    float v = 0.f;
    if (a < A) v = transpose_ab ? in[((int64_t)b * A + a) * T + t] : in[((int64_t)a * B + b) * T + t];
    out[i] = v;
  }
}

// ---------------------------------------------------------------- SE gate
// one wave per (n, c): gate = sigmoid(b[c] + W[c,:] . mean[n,:])
__global__ void se_gate_fwd_kernel(const double* __restrict__ stats, const float* __restrict__ w,
                                   const float* __restrict__ b, float* __restrict__ gate,
                                   float* __restrict__ mean, int N, int C, double invS) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 4 + wave;
  const int n = blockIdx.y;
  if (c >= C) return;
  float acc = 0.f;
  for (int k = lane; k < C; k += 64) {
    const float m = (float)(stats[((int64_t)n * C + k) * 2] * invS);
    if (c == 0 || (blockIdx.x == 0 && wave == 0)) mean[(int64_t)n * C + k] = m;
    acc += w[(int64_t)c * C + k] * m;
  }
  acc = wave_sum(acc);
  if (lane == 0) gate[(int64_t)n * C + c] = 1.f / (1.f + __expf(-(acc + b[c])));
}

// dW[c][k] = sum_n ds[n][c] * mean[n][k]; db[c] = sum_n ds[n][c]; ds = dg*g*(1-g)
__global__ void se_gate_bwd_w_kernel(const double* __restrict__ dgate, const float* __restrict__ gate,
                                     const float* __restrict__ mean, float* __restrict__ dw,
                                     float* __restrict__ db, int N, int C) {
  const int64_t total = (int64_t)C * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % C), c = (int)(i / C);
    float s = 0.f, sb = 0.f;
    for (int n = 0; n < N; ++n) {
      const float g = gate[(int64_t)n * C + c];
      const float ds = (float)dgate[(int64_t)n * C + c] * g * (1.f - g);
      s += ds * mean[(int64_t)n * C + k];
      sb += ds;
    }
    dw[i] = s;
    if (k == 0) db[c] = sb;
  }
}
// kconst[n][k] = (sum_c W[c][k] * ds[n][c]) / S
// block = 64 columns k x 4 slices of c (a thread per (n,k) walking all C rows serially is one
// dependent-latency chain per output: 79 us for C = 512)
__global__ __launch_bounds__(256) void se_gate_bwd_k_kernel(const double* __restrict__ dgate,
                                                            const float* __restrict__ gate,
                                                            const float* __restrict__ w, float* __restrict__ kconst,
                                                            int N, int C, float invS) {
  __shared__ float part[4][64];
  const int n = blockIdx.y;
  const int kl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + kl;
  const int per = (C + 3) / 4, c0 = sl * per, c1 = min(C, c0 + per);
  float s0 = 0.f, s1 = 0.f;
  if (k < C) {
    int c = c0;
    for (; c + 1 < c1; c += 2) {
      const float g0 = gate[(int64_t)n * C + c], g1 = gate[(int64_t)n * C + c + 1];
      s0 += w[(int64_t)c * C + k] * ((float)dgate[(int64_t)n * C + c] * g0 * (1.f - g0));
      s1 += w[(int64_t)(c + 1) * C + k] * ((float)dgate[(int64_t)n * C + c + 1] * g1 * (1.f - g1));
    }
    if (c < c1) {
      const float g0 = gate[(int64_t)n * C + c];
      s0 += w[(int64_t)c * C + k] * ((float)dgate[(int64_t)n * C + c] * g0 * (1.f - g0));
    }
  }
  part[sl][kl] = s0 + s1;
  __syncthreads();
  if (sl == 0 && k < C) kconst[(int64_t)n * C + k] = (part[0][kl] + part[1][kl] + part[2][kl] + part[3][kl]) * invS;
}

// Column-reduction skeleton: a block owns `rows_per_block` consecutive rows of
// ONE sample; thread t owns channel quad (t % c4n) and walks rows t / c4n,
// + rpp, ...  NQ quantities x 4 channels are reduced over the block in LDS and
// finished with one double atomic per (quantity, channel).
template <int NQ, typename F>
__device__ __forceinline__ void column_reduce(int64_t row_begin, int64_t row_end, int C, double* out,
                                              int out_stride, F body) {
  __shared__ double red[EW_THREADS * NQ * 4];
  const int c4n = C >> 2;
  const int tid = threadIdx.x;
  const int rpp = EW_THREADS / c4n;  // rows per pass (c4n <= 256)
  const int cq = tid % c4n, rl = tid / c4n;
  double acc[NQ][4];  // fp64: these sums cancel heavily behind InstanceNorm / SE pooling
#pragma unroll
  for (int qn = 0; qn < NQ; ++qn)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[qn][e] = 0.0;
  if (rl < rpp) {
    for (int64_t r = row_begin + rl; r < row_end; r += rpp) body(r, cq * 4, acc);
  }
#pragma unroll
  for (int qn = 0; qn < NQ; ++qn)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(qn * 4 + e) * EW_THREADS + tid] = (rl < rpp) ? acc[qn][e] : 0.f;
  __syncthreads();
  // thread (qn, e, cq) sums the rl copies
  for (int o = tid; o < NQ * 4 * c4n; o += EW_THREADS) {
    const int cqq = o % c4n, qe = o / c4n;
    double s = 0.0;
    for (int k = 0; k < rpp; ++k) s += red[qe * EW_THREADS + k * c4n + cqq];
    const int qn = qe >> 2, e = qe & 3;
    atomicAdd(out + (int64_t)(cqq * 4 + e) * out_stride + qn, s);
  }
}

// ---------------------------------------------------------------- depth upsample
__device__ __forceinline__ void depth_src(int od, int Di, int Do, int& i0, int& i1, float& w1) {
  // align_corners=True: src = od * (Di-1)/(Do-1)
  const float scale = (Do > 1) ? (float)(Di - 1) / (float)(Do - 1) : 0.f;
  const float src = scale * (float)od;
  i0 = (int)src;
  if (i0 > Di - 1) i0 = Di - 1;
  i1 = (i0 < Di - 1) ? i0 + 1 : i0;
  w1 = src - (float)i0;
}
__global__ void upsample_depth_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                          int Di, int Do, int64_t HW, int C) {
  const int64_t plane = HW * C / 4;  // float4 per depth slice
  const int64_t total = (int64_t)N * Do * plane;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i % plane;
    const int64_t r = i / plane;
    const int od = (int)(r % Do), n = (int)(r / Do);
    int i0, i1;
    float w1;
    depth_src(od, Di, Do, i0, i1, w1);
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[((int64_t)n * Di + i0) * plane + p];
    const f32x4 b = reinterpret_cast<const f32x4*>(x)[((int64_t)n * Di + i1) * plane + p];
    reinterpret_cast<f32x4*>(y)[i] = a * (1.f - w1) + b * w1;
  }
}
__global__ void upsample_depth_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N,
                                          int Di, int Do, int64_t HW, int C) {
  const int64_t plane = HW * C / 4;
  const int64_t total = (int64_t)N * Di * plane;
  // outputs that can touch input slice id lie in [lo, hi]
  const float inv = (Di > 1) ? (float)(Do - 1) / (float)(Di - 1) : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i % plane;
    const int64_t r = i / plane;
    const int id = (int)(r % Di), n = (int)(r / Di);
    int lo = (int)floorf((float)(id - 1) * inv) - 1, hi = (int)ceilf((float)(id + 1) * inv) + 1;
    if (lo < 0 || Di == 1) lo = 0;
    if (hi > Do - 1 || Di == 1) hi = Do - 1;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int od = lo; od <= hi; ++od) {
      int i0, i1;
      float w1;
      depth_src(od, Di, Do, i0, i1, w1);
      float wgt = 0.f;
      if (i0 == id) wgt += 1.f - w1;
      if (i1 == id) wgt += w1;
      if (wgt != 0.f) s += reinterpret_cast<const f32x4*>(dy)[((int64_t)n * Do + od) * plane + p] * wgt;
    }
    reinterpret_cast<f32x4*>(dx)[i] = s;
  }
}

// ---------------------------------------------------------------- depth upsample fused with a depth-tap sum
// y = act(b + conv3x3x3(upsample_depth(f)))  (models/seg_model.py:204-205, sr_head.0 on the 4x upsampled features)
// is linear in f, so the 3x3 part of every depth tap is taken on the LOW-resolution slices first,
//   G[n][j][hw][kd*C + co] = sum_{ci,kh,kw} f[n][j][..][ci] w[co][ci][kd][kh][kw]      (a (1,3,3) conv, KD*C outputs)
// and this kernel interpolates and sums the taps:
//   y[n][d][hw][co] = act(b[co] + sum_kd [0 <= d+kd-p < Do] ((1-t) G[n][i0][hw][kd*C+co] + t G[n][i1][hw][kd*C+co]))
// with (i0, i1, t) the align_corners source of upsampled slice d+kd-p.  upscale x fewer multiplications, and the
// upscale x larger feature tensor never exists.
// A thread walks UPMIX_RUN consecutive output slices of one (sample, voxel, channel quad) column and keeps the two
// source slices of every depth tap in registers: they change once per `upscale` steps, so g is read ~once.
constexpr int UPMIX_RUN = 16, UPMIX_MAXKD = 5;

// four consecutive elements as fp32 (activations may be fp32 or, on the mixed-precision path, bf16)
template <typename T> struct Quad;
template <> struct Quad<float> {
  __device__ static __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  __device__ static __forceinline__ void st(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Quad<__bf16> {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  __device__ static __forceinline__ f32x4 ld(const __bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
  __device__ static __forceinline__ void st(__bf16* p, const f32x4 v) {
    const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
  }
};
template <typename TG, typename TY>
__global__ void upmix_depth_fwd_kernel(const TG* __restrict__ g, const float* __restrict__ bias,
                                       TY* __restrict__ y, int N, int Di, int Do, int64_t HW, int C, int KD,
                                       int pd, int act, float slope) {
  const int cq = C / 4;
  const int64_t plane = HW * cq;  // float4 of y per depth slice
  const int runs = (Do + UPMIX_RUN - 1) / UPMIX_RUN;
  const int64_t total = (int64_t)N * runs * plane;
  const int64_t gplane = HW * KD * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i % plane;
    const int64_t r = i / plane;
    const int run = (int)(r % runs), n = (int)(r / runs);
    const int q = (int)(p % cq);
    const int64_t hw = p / cq;
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    const TG* gn = g + ((int64_t)n * Di * gplane + hw * KD * cq + q) * 4;   // (gplane / cq count channel quads)
    int c0[UPMIX_MAXKD], c1[UPMIX_MAXKD];  // cached source slices per depth tap
    f32x4 v0[UPMIX_MAXKD], v1[UPMIX_MAXKD];
#pragma unroll
    for (int kd = 0; kd < UPMIX_MAXKD; ++kd) { c0[kd] = -1; c1[kd] = -1; v0[kd] = bv; v1[kd] = bv; }
    const int od_end = min(Do, (run + 1) * UPMIX_RUN);
    for (int od = run * UPMIX_RUN; od < od_end; ++od) {
      f32x4 s = bv;
#pragma unroll
      for (int kd = 0; kd < UPMIX_MAXKD; ++kd) {
        if (kd < KD) {
          const int ud = od + kd - pd;  // slice of the (virtual) upsampled tensor
          if ((unsigned)ud < (unsigned)Do) {
            int i0, i1;
            float w1;
            depth_src(ud, Di, Do, i0, i1, w1);
            if (i0 != c0[kd]) {
              if (i0 == c1[kd]) v0[kd] = v1[kd];  // the interval moved up by one slice
              else v0[kd] = Quad<TG>::ld(gn + ((int64_t)i0 * gplane + kd * cq) * 4);
              c0[kd] = i0;
            }
            if (i1 != c1[kd]) {
              v1[kd] = (i1 == i0) ? v0[kd] : Quad<TG>::ld(gn + ((int64_t)i1 * gplane + kd * cq) * 4);
              c1[kd] = i1;
            }
            s += v0[kd] * (1.f - w1) + v1[kd] * w1;
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = apply_act(s[e], act, slope);
      Quad<TY>::st(y + (((int64_t)n * Do + od) * plane + p) * 4, s);
    }
  }
}
// dG[n][j][hw][kd*C+co] = sum over upsampled slices ud with source j of coef(ud, j) * dz[n][ud - kd + p][hw][co]
//
// A thread owns one (sample, voxel column, channel quad) and walks the Do output slices ONCE: dz (and y, for the
// activation gradient formed on the fly) are read exactly once, every dG element is written exactly once.  Per depth
// tap the two source slices an upsampled slice interpolates between move monotonically with od, so two running
// accumulators per tap (for slices b and b+1) suffice; an accumulator is stored the moment the window leaves its
// slice.  (The gather form -- a thread per dG element looping over the ~2*scale+2 slices that touch it -- fetched
// every dz / y element 3.3 times: 7.98 GB per launch against 2.95 GB algorithmic, PMC, profiles/r02_pmc_hbm_stream.json.)
template <typename TG, typename TY>
__global__ __launch_bounds__(256) void upmix_depth_bwd_kernel(const TY* __restrict__ dz, const TY* __restrict__ yact,
                                                             TG* __restrict__ dg, int N, int Di, int Do, int64_t HW, int C,
                                                             int KD, int pd, int act, float slope) {
  const int cq = C / 4;
  const int64_t plane = HW * cq;            // float4 per output slice
  const int64_t gplane = HW * KD * cq;      // float4 per source slice of g
  const int64_t total = (int64_t)N * plane;
  const float ga = act == REHR_ACT_RELU ? 0.f : (act == REHR_ACT_LRELU ? slope : 1.f);
  const float scale = (Do > 1) ? (float)(Di - 1) / (float)(Do - 1) : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i % plane;
    const int n = (int)(i / plane);
    const int q = (int)(p % cq);
    const int64_t hw = p / cq;
    const TY* dzp = dz + ((int64_t)n * Do * plane + p) * 4;
    const TY* yp = yact ? yact + ((int64_t)n * Do * plane + p) * 4 : nullptr;
    TG* gout = dg + ((int64_t)n * Di * gplane + hw * KD * cq + q) * 4;
    f32x4 a0[UPMIX_MAXKD], a1[UPMIX_MAXKD];
    int base[UPMIX_MAXKD];
#pragma unroll
    for (int kd = 0; kd < UPMIX_MAXKD; ++kd) {
      a0[kd] = f32x4{0.f, 0.f, 0.f, 0.f};
      a1[kd] = f32x4{0.f, 0.f, 0.f, 0.f};
      base[kd] = 0;
    }
    for (int od = 0; od < Do; ++od) {
      f32x4 d = Quad<TY>::ld(dzp + (int64_t)od * plane * 4);
      if (yp != nullptr) {
        const f32x4 yv = Quad<TY>::ld(yp + (int64_t)od * plane * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = yv[e] > 0.f ? d[e] : d[e] * ga;
      }
#pragma unroll
      for (int kd = 0; kd < UPMIX_MAXKD; ++kd) {
        if (kd < KD) {
          const int ud = od + kd - pd;
          if ((unsigned)ud < (unsigned)Do) {
            const float src = scale * (float)ud;     // depth_src(): align_corners=True
            int i0 = (int)src;
            if (i0 > Di - 1) i0 = Di - 1;
            const float w1 = src - (float)i0;
            while (base[kd] < i0) {                  // the window leaves slice base: its accumulator is final
              Quad<TG>::st(gout + ((int64_t)base[kd] * gplane + kd * cq) * 4, a0[kd]);
              a0[kd] = a1[kd];
              a1[kd] = f32x4{0.f, 0.f, 0.f, 0.f};
              ++base[kd];
            }
            a0[kd] += d * (1.f - w1);
            if (i0 < Di - 1) a1[kd] += d * w1;
            else a0[kd] += d * w1;                    // i1 == i0 at the last slice
          }
        }
      }
    }
#pragma unroll
    for (int kd = 0; kd < UPMIX_MAXKD; ++kd) {
      if (kd < KD) {                                  // flush the window and any slice no upsampled slice touched
        for (int id = base[kd]; id < Di; ++id) {
          Quad<TG>::st(gout + ((int64_t)id * gplane + kd * cq) * 4,
                       (id == base[kd]) ? a0[kd] : ((id == base[kd] + 1) ? a1[kd] : f32x4{0.f, 0.f, 0.f, 0.f}));
        }
      }
    }
  }
}

// ---------------------------------------------------------------- stem of overlapping depth windows
// The distillation teacher runs UNet_3D_3D's encoder on every 4-slice window of a volume (train_all.py:85-112):
// neighbouring windows share 3 of their 4 slices, and the stem conv (3 depth taps, zero padding at the WINDOW's
// borders, per-window mean subtracted from channel 0 first, FLAVR_arch.py:181) is linear, so its (kH,kW) part is
// taken once per volume slice and depth tap -- g[kd][slice][hw][c], the last "slice" being the response r[kd] to a
// constant 1 in channel 0 -- and this kernel assembles window w, slice k:
//   y[w][k][hw][c] = relu(bias[c] + sum_{kd: 0 <= k+kd-1 <= 3} (g[kd][w+k+kd-1][hw][c] - mean[w] * r[kd][hw][c]))
// 4x fewer multiplications than convolving every window.  nslices = slices per sample incl. the padding (= nwin + 3).
template <typename TO>   // float, or __bf16 when the teacher runs in mixed precision (its next layer takes bf16)
__global__ void window_stem_assemble_kernel(const float* __restrict__ g0, const float* __restrict__ g1,
                                            const float* __restrict__ g2, const float* __restrict__ mean,
                                            const float* __restrict__ bias, TO* __restrict__ y, int B, int nwin,
                                            int nslices, int64_t HW, int C, int act, float slope) {
  const int cq = C / 4;
  const int64_t plane = HW * cq;
  const int64_t total = (int64_t)B * nwin * 4 * plane;
  const f32x4* gs[3] = {reinterpret_cast<const f32x4*>(g0), reinterpret_cast<const f32x4*>(g1),
                        reinterpret_cast<const f32x4*>(g2)};
  const int64_t rslice = (int64_t)B * nslices;  // index of the constant-1 response
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i % plane;
    int64_t r = i / plane;
    const int k = (int)(r % 4); r /= 4;
    const int w = (int)(r % nwin);
    const int b = (int)(r / nwin);
    const int q = (int)(p % cq);
    const float m = mean[(int64_t)b * nwin + w];
    f32x4 s = bias ? *reinterpret_cast<const f32x4*>(bias + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int ks = k + kd - 1;  // window slice this tap reads
      if (ks < 0 || ks > 3) continue;
      const f32x4 gv = gs[kd][((int64_t)b * nslices + w + ks) * plane + p];
      const f32x4 rv = gs[kd][rslice * plane + p];
      s += gv - rv * m;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = apply_act(s[e], act, slope);
    store4(y + 4 * i, s);
  }
}

// ---------------------------------------------------------------- quadrant max-pool (Distiller's structure loss)
// CriterionPairWiseforWholeFeatAfterPool (models/seg_model.py:95-113) max-pools every (sample, depth) slice of a
// 64-channel feature map with a (H/2, W/2) window: 4 outputs per slice and channel.  ATen's NHWC pooling kernel
// walks such a window serially per output (0.55 ms for 128 slices of 64x64); here a block owns one (slice,
// quadrant): lanes = channels (coalesced voxel records), 4 row groups, one LDS step.  idx = row-major position
// of the FIRST maximum inside the slice (what MaxPool2d's backward uses).
__global__ __launch_bounds__(256) void quad_maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int* __restrict__ idx, int H, int W, int C) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const int slice = blockIdx.x >> 2, qd = blockIdx.x & 3;
  const int ph = H / 2, pw = W / 2;
  const int h0 = (qd >> 1) * ph, w0 = (qd & 1) * pw;
  const int lanes = C < 256 ? C : 256, groups = 256 / lanes;
  const int c_l = threadIdx.x % lanes, g = threadIdx.x / lanes;
  for (int c = c_l; c < C; c += lanes) {
    float best = -INFINITY;
    int bi = (h0 * W + w0);
    if (g < groups) {
      for (int r = g; r < ph; r += groups) {
        const float* row = x + (((int64_t)slice * H + h0 + r) * W + w0) * C + c;
        for (int q = 0; q < pw; ++q) {
          const float v = row[(int64_t)q * C];
          if (v > best || v != v) { best = v; bi = (h0 + r) * W + w0 + q; }  // (NaN propagates like ATen)
        }
      }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    if (g == 0) {
      for (int k = 1; k < groups; ++k) {
        const float v = sv[k * lanes + c_l];
        const int vi = si[k * lanes + c_l];
        if (v > best || (v == best && vi < bi)) { best = v; bi = vi; }
      }
      y[((int64_t)slice * 4 + qd) * C + c] = best;
      idx[((int64_t)slice * 4 + qd) * C + c] = bi;
    }
    __syncthreads();
  }
}
// dx (zero-filled by the caller) [slice][idx][c] = dy[slice][quadrant][c]
__global__ void quad_maxpool_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                        float* __restrict__ dx, int64_t total, int HW, int C) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t slice = i / ((int64_t)4 * C);
    dx[(slice * HW + idx[i]) * C + c] = dy[i];
  }
}

// ---------------------------------------------------------------- channel-normalised cosine distance (Distiller)
// cosine_distance_loss (models/seg_model.py:60-78): t = x / max(|x[:, v]|_2, 1e-12) per voxel (F.normalize over
// channels), then per (sample, channel) the cosine similarity of t1[c, :] and t2[c, :] over the flattened voxels,
// loss = mean(1 - cos).  ATen runs it as ~10 passes over the two 64-channel tensors; here lane = channel (C == 64),
// a wave = one voxel at a time: pass 1 accumulates S12, S11, S22 per (sample, channel) (fp64 atomics per block),
// pass 2 (gradient w.r.t. x1 only: x2 is the frozen teacher) recomputes the voxel norms and applies
//   g = -(gl / (N C)) (t2 / (a b) - S12 t1 / (a^3 b)),  a = max(sqrt(S11), eps), b = max(sqrt(S22), eps)
//   dx1 = (g - t1 * sum_c(g t1)) / n1.
__global__ __launch_bounds__(256) void cosdist_stats_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                            double* __restrict__ stats, int64_t S,
                                                            int64_t vox_per_block) {
  __shared__ float red[4][3][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y;
  const int64_t v0 = (int64_t)blockIdx.x * vox_per_block;
  const int64_t v1 = min(S, v0 + vox_per_block);
  float s12 = 0.f, s11 = 0.f, s22 = 0.f;
  for (int64_t v = v0 + wave; v < v1; v += 4) {
    const float a = x1[((int64_t)n * S + v) * 64 + lane], b = x2[((int64_t)n * S + v) * 64 + lane];
    const float na = fmaxf(sqrtf(wave_sum(a * a)), 1e-12f), nb = fmaxf(sqrtf(wave_sum(b * b)), 1e-12f);
    const float ta = a / na, tb = b / nb;
    s12 += ta * tb;
    s11 += ta * ta;
    s22 += tb * tb;
  }
  red[wave][0][lane] = s12;
  red[wave][1][lane] = s11;
  red[wave][2][lane] = s22;
  __syncthreads();
  if (threadIdx.x < 192) {
    const int k = threadIdx.x >> 6;
    const float t = red[0][k][lane] + red[1][k][lane] + red[2][k][lane] + red[3][k][lane];
    atomicAdd(stats + ((int64_t)n * 64 + lane) * 3 + k, (double)t);
  }
}
__global__ __launch_bounds__(256) void cosdist_bwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                          const double* __restrict__ stats, float* __restrict__ dx1,
                                                          int64_t S, int64_t total_vox, float scale) {
  const int lane = threadIdx.x & 63;
  for (int64_t gv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); gv < total_vox; gv += (int64_t)gridDim.x * 4) {
    const int n = (int)(gv / S);
    const double* st = stats + ((int64_t)n * 64 + lane) * 3;
    const float S12 = (float)st[0];
    const float ca = fmaxf(sqrtf((float)st[1]), 1e-8f), cb = fmaxf(sqrtf((float)st[2]), 1e-8f);
    const float a = x1[gv * 64 + lane], b = x2[gv * 64 + lane];
    const float na = fmaxf(sqrtf(wave_sum(a * a)), 1e-12f), nb = fmaxf(sqrtf(wave_sum(b * b)), 1e-12f);
    const float ta = a / na, tb = b / nb;
    const float g = scale * (tb / (ca * cb) - S12 * ta / (ca * ca * ca * cb));
    const float dot = wave_sum(g * ta);
    dx1[gv * 64 + lane] = (g - ta * dot) / na;
  }
}

// ---------------------------------------------------------------- misc
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4,
                               int act, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act, slope);
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}

// column sums of dy * act'(y): the bias gradient behind a fused activation without materialising dz
template <typename T>
__global__ void channel_sum_actgrad_kernel(const T* __restrict__ dy, const T* __restrict__ y, int ld,
                                           int64_t rows, int C, int64_t rows_per_block, int act, float slope,
                                           double* __restrict__ scratch) {
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > rows) r_end = rows;
  column_reduce<1>(r_begin, r_end, C, scratch, 1, [&](int64_t row, int c, double(&acc)[1][4]) {
    const f32x4 v = Quad<T>::ld(dy + row * ld + c);
    const f32x4 yv = Quad<T>::ld(y + row * ld + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[0][e] += v[e] * act_grad(yv[e], act, slope);
  });
}
__global__ void channel_sum_finish_kernel(const double* __restrict__ scratch, float* __restrict__ out,
                                          int C, int accumulate) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x)
    out[c] = accumulate ? out[c] + (float)scratch[c] : (float)scratch[c];
}

__global__ void copy_channels_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                     int ldy, int64_t rows, int C) {
  const int c4n = C >> 2;
  const int64_t total = rows * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const int64_t row = i / c4n;
    *reinterpret_cast<f32x4*>(y + row * ldy + c) = *reinterpret_cast<const f32x4*>(x + row * ldx + c);
  }
}

// NCDHW <-> NDHWC through a 32x32 LDS tile (coalesced on both sides)
__global__ void transpose_cs_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                    int64_t S, int to_nhwc) {
  __shared__ float t[32][33];
  const int n = blockIdx.z;
  const int64_t s0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* xn = x + (int64_t)n * C * S;
  float* yn = y + (int64_t)n * C * S;
  if (to_nhwc) {
    for (int k = ty; k < 32; k += 8) {
      const int c = c0 + k;
      const int64_t s = s0 + tx;
      t[k][tx] = (c < C && s < S) ? xn[(int64_t)c * S + s] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      const int64_t s = s0 + k;
      const int c = c0 + tx;
      if (c < C && s < S) yn[s * C + c] = t[tx][k];
    }
  } else {
    for (int k = ty; k < 32; k += 8) {
      const int64_t s = s0 + k;
      const int c = c0 + tx;
      t[k][tx] = (c < C && s < S) ? xn[s * C + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      const int c = c0 + k;
      const int64_t s = s0 + tx;
      if (c < C && s < S) yn[(int64_t)c * S + s] = t[tx][k];
    }
  }
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
// blocks for a per-sample column reduction: cover S rows with <= ~1024 blocks/sample
inline int64_t rows_per_block_for(int64_t S, int C, int N) {
  const int rpp = EW_THREADS / (C / 4);
  int64_t target_blocks = 2048 / (N > 0 ? N : 1);
  if (target_blocks < 1) target_blocks = 1;
  int64_t rpb = (S + target_blocks - 1) / target_blocks;
  const int64_t min_rows = (int64_t)rpp * 8;
  if (rpb < min_rows) rpb = min_rows;
  return rpb;
}

}  // namespace

#define ST ((hipStream_t)stream)

thread_local int rehr_last_hip_error_code = 0;
extern "C" int rehr_abi_version(void) { return 4; }
extern "C" const char* rehr_last_hip_error(void) {
  return hipGetErrorString((hipError_t)rehr_last_hip_error_code);
}

extern "C" int rehr_pack_weights_f32(const float* in, float* out, int32_t A, int32_t Apad, int32_t B,
                                     int32_t T, int32_t transpose_ab, void* stream) {
  if (!in || !out || A < 1 || Apad < A || B < 1 || T < 1) return REHR_EINVAL;
  // (an LDS-tiled version with coalesced reads was tried in round 3 and measured slower: 19-20 us against 8-11 us per
  // call -- the strided reads of this kernel are served by the 256 MB Infinity Cache, the tiled kernel had too few blocks
  // on the small panels: gpurun_out r3f / DESIGN.md section 7)
  const int64_t total = (int64_t)T * Apad * B;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, in, out, A,
                     Apad, B, T, transpose_ab);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_se_gate_fwd_f32(const double* stats, const float* w, const float* b, float* gate,
                                    float* mean, int32_t N, int32_t C, int64_t S, void* stream) {
  if (!stats || !w || !b || !gate || !mean || N < 1 || C < 1 || S < 1 || N > 65535) return REHR_EINVAL;
  hipLaunchKernelGGL(se_gate_fwd_kernel, dim3((C + 3) / 4, N), dim3(256), 0, ST, stats, w, b, gate,
                     mean, N, C, 1.0 / (double)S);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_se_gate_bwd_f32(const double* dgate_acc, const float* gate, const float* mean,
                                    const float* w, float* dw, float* db, float* kconst, int32_t N,
                                    int32_t C, int64_t S, void* stream) {
  if (!dgate_acc || !gate || !mean || !w || !dw || !db || !kconst || N < 1 || C < 1 || S < 1)
    return REHR_EINVAL;
  hipLaunchKernelGGL(se_gate_bwd_w_kernel, dim3(ew_blocks((int64_t)C * C)), dim3(EW_THREADS), 0, ST,
                     dgate_acc, gate, mean, dw, db, N, C);
  hipLaunchKernelGGL(se_gate_bwd_k_kernel, dim3((C + 63) / 64, N), dim3(256), 0, ST, dgate_acc, gate, w, kconst, N, C,
                     (float)(1.0 / (double)S));
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_upsample_depth_fwd_f32(const float* x, float* y, int32_t N, int32_t Di, int32_t Do,
                                           int64_t HW, int32_t C, void* stream) {
  if (!x || !y || N < 1 || Di < 1 || Do < 1 || HW < 1 || C < 4 || C % 4 || !aligned16(x) || !aligned16(y))
    return REHR_EINVAL;
  const int64_t total = (int64_t)N * Do * HW * C / 4;
  hipLaunchKernelGGL(upsample_depth_fwd_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, x, y, N,
                     Di, Do, HW, C);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
extern "C" int rehr_upsample_depth_bwd_f32(const float* dy, float* dx, int32_t N, int32_t Di, int32_t Do,
                                           int64_t HW, int32_t C, void* stream) {
  if (!dy || !dx || N < 1 || Di < 1 || Do < 1 || HW < 1 || C < 4 || C % 4 || !aligned16(dy) ||
      !aligned16(dx))
    return REHR_EINVAL;
  const int64_t total = (int64_t)N * Di * HW * C / 4;
  hipLaunchKernelGGL(upsample_depth_bwd_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, dy, dx,
                     N, Di, Do, HW, C);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

namespace {
template <typename TG, typename TY>
int upmix_depth_fwd_t(const TG* g, const float* bias, TY* y, int32_t N, int32_t Di, int32_t Do, int64_t HW, int32_t C,
                      int32_t KD, int32_t pd, int32_t act, float slope, void* stream) {
  if (!g || !y || N < 1 || Di < 1 || Do < 1 || HW < 1 || C < 4 || C % 4 || KD < 1 || pd < 0 || pd >= KD ||
      !aligned16(g) || !aligned16(y) || (bias && !aligned16(bias)) || KD > UPMIX_MAXKD)
    return REHR_EINVAL;
  const int64_t total = (int64_t)N * ((Do + UPMIX_RUN - 1) / UPMIX_RUN) * HW * C / 4;
  hipLaunchKernelGGL((upmix_depth_fwd_kernel<TG, TY>), dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, g, bias, y, N, Di,
                     Do, HW, C, KD, pd, act, slope);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
template <typename TG, typename TY>
int upmix_depth_bwd_t(const TY* dz, const TY* y, TG* dg, int32_t N, int32_t Di, int32_t Do, int64_t HW, int32_t C,
                      int32_t KD, int32_t pd, int32_t act, float slope, void* stream) {
  if (!dz || !dg || N < 1 || Di < 1 || Do < 1 || HW < 1 || C < 4 || C % 4 || KD < 1 || pd < 0 || pd >= KD ||
      !aligned16(dz) || !aligned16(dg) || (y && !aligned16(y)))
    return REHR_EINVAL;
  const int64_t total = (int64_t)N * HW * C / 4;   // one thread per (sample, voxel column, channel quad)
  hipLaunchKernelGGL((upmix_depth_bwd_kernel<TG, TY>), dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, dz,
                     act == REHR_ACT_NONE ? nullptr : y, dg, N, Di, Do, HW, C, KD, pd, act, slope);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
template <typename T>
int channel_sum_actgrad_t(const T* dy, const T* y, int32_t ld, int64_t rows, int32_t C, int32_t act, float slope,
                          float* out, double* scratch, void* stream) {
  if (!dy || !y || !out || !scratch || rows < 1 || C < 4 || C % 4 || C > 1024 || ld % 4 || !aligned16(dy) ||
      !aligned16(y))
    return REHR_EINVAL;
  if (hipMemsetAsync(scratch, 0, sizeof(double) * C, ST) != hipSuccess) return REHR_EHIP;
  const int64_t rpb = rows_per_block_for(rows, C, 1);
  const int blocks = (int)((rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(channel_sum_actgrad_kernel<T>, dim3(blocks), dim3(EW_THREADS), 0, ST, dy, y, ld, rows, C, rpb, act,
                     slope, scratch);
  hipLaunchKernelGGL(channel_sum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, scratch, out, C, 0);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
}  // namespace

extern "C" int rehr_channel_sum_actgrad_f32(const float* dy, const float* y, int32_t ld, int64_t rows, int32_t C,
                                            int32_t act, float slope, float* out, double* scratch, void* stream) {
  return channel_sum_actgrad_t<float>(dy, y, ld, rows, C, act, slope, out, scratch, stream);
}
extern "C" int rehr_channel_sum_actgrad_bf16(const void* dy, const void* y, int32_t ld, int64_t rows, int32_t C,
                                             int32_t act, float slope, float* out, double* scratch, void* stream) {
  return channel_sum_actgrad_t<__bf16>(reinterpret_cast<const __bf16*>(dy), reinterpret_cast<const __bf16*>(y), ld, rows,
                                       C, act, slope, out, scratch, stream);
}
extern "C" int rehr_upmix_depth_fwd_f32(const float* g, const float* bias, float* y, int32_t N, int32_t Di, int32_t Do,
                                        int64_t HW, int32_t C, int32_t KD, int32_t pd, int32_t act, float slope,
                                        void* stream) {
  return upmix_depth_fwd_t<float, float>(g, bias, y, N, Di, Do, HW, C, KD, pd, act, slope, stream);
}
extern "C" int rehr_upmix_depth_bwd_f32(const float* dz, const float* y, float* dg, int32_t N, int32_t Di, int32_t Do,
                                        int64_t HW, int32_t C, int32_t KD, int32_t pd, int32_t act, float slope,
                                        void* stream) {
  return upmix_depth_bwd_t<float, float>(dz, y, dg, N, Di, Do, HW, C, KD, pd, act, slope, stream);
}
// mixed precision: g / dg and y / dz are bf16 (8-byte quads), bias fp32, arithmetic fp32
extern "C" int rehr_upmix_depth_fwd_bf16(const void* g, const float* bias, void* y, int32_t N, int32_t Di, int32_t Do,
                                         int64_t HW, int32_t C, int32_t KD, int32_t pd, int32_t act, float slope,
                                         void* stream) {
  return upmix_depth_fwd_t<__bf16, __bf16>(reinterpret_cast<const __bf16*>(g), bias, reinterpret_cast<__bf16*>(y), N, Di,
                                           Do, HW, C, KD, pd, act, slope, stream);
}
extern "C" int rehr_upmix_depth_bwd_bf16(const void* dz, const void* y, void* dg, int32_t N, int32_t Di, int32_t Do,
                                         int64_t HW, int32_t C, int32_t KD, int32_t pd, int32_t act, float slope,
                                         void* stream) {
  return upmix_depth_bwd_t<__bf16, __bf16>(reinterpret_cast<const __bf16*>(dz), reinterpret_cast<const __bf16*>(y),
                                           reinterpret_cast<__bf16*>(dg), N, Di, Do, HW, C, KD, pd, act, slope, stream);
}

extern "C" int rehr_window_stem_assemble_f32(const float* g0, const float* g1, const float* g2, const float* mean,
                                             const float* bias, float* y, int32_t B, int32_t nwin, int32_t nslices,
                                             int64_t HW, int32_t C, int32_t act, float slope, void* stream) {
  if (!g0 || !g1 || !g2 || !mean || !y || B < 1 || nwin < 1 || nslices != nwin + 3 || HW < 1 || C < 4 || C % 4 ||
      !aligned16(g0) || !aligned16(g1) || !aligned16(g2) || !aligned16(y) || (bias && !aligned16(bias)))
    return REHR_EINVAL;
  const int64_t total = (int64_t)B * nwin * 4 * HW * C / 4;
  hipLaunchKernelGGL(window_stem_assemble_kernel<float>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, g0, g1, g2, mean,
                     bias, y, B, nwin, nslices, HW, C, act, slope);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_window_stem_assemble_bf16(const float* g0, const float* g1, const float* g2, const float* mean,
                                              const float* bias, void* y, int32_t B, int32_t nwin, int32_t nslices,
                                              int64_t HW, int32_t C, int32_t act, float slope, void* stream) {
  if (!g0 || !g1 || !g2 || !mean || !y || B < 1 || nwin < 1 || nslices != nwin + 3 || HW < 1 || C < 4 || C % 4 ||
      !aligned16(g0) || !aligned16(g1) || !aligned16(g2) || (((uintptr_t)y) & 7) || (bias && !aligned16(bias)))
    return REHR_EINVAL;
  const int64_t total = (int64_t)B * nwin * 4 * HW * C / 4;
  hipLaunchKernelGGL(window_stem_assemble_kernel<__bf16>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, g0, g1, g2, mean,
                     bias, (__bf16*)y, B, nwin, nslices, HW, C, act, slope);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_quad_maxpool_fwd_f32(const float* x, float* y, int32_t* idx, int64_t slices, int32_t H, int32_t W,
                                         int32_t C, void* stream) {
  if (!x || !y || !idx || slices < 1 || slices * 4 >= (1ll << 31) || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 1)
    return REHR_EINVAL;
  hipLaunchKernelGGL(quad_maxpool_fwd_kernel, dim3((unsigned)(slices * 4)), dim3(256), 0, ST, x, y, idx, H, W, C);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
extern "C" int rehr_quad_maxpool_bwd_f32(const float* dy, const int32_t* idx, float* dx, int64_t slices, int32_t H,
                                         int32_t W, int32_t C, void* stream) {
  if (!dy || !idx || !dx || slices < 1 || H < 2 || W < 2 || C < 1) return REHR_EINVAL;
  if (hipMemsetAsync(dx, 0, (size_t)slices * H * W * C * sizeof(float), ST) != hipSuccess) return REHR_EHIP;
  const int64_t total = slices * 4 * C;
  hipLaunchKernelGGL(quad_maxpool_bwd_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, dy, idx, dx, total, H * W, C);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_cosdist_stats_f32(const float* x1, const float* x2, double* stats, int32_t N, int64_t S, int32_t C,
                                      void* stream) {
  if (!x1 || !x2 || !stats || N < 1 || N > 65535 || S < 1 || C != 64) return REHR_EINVAL;
  if (hipMemsetAsync(stats, 0, sizeof(double) * N * 64 * 3, ST) != hipSuccess) return REHR_EHIP;
  int64_t blocks = 1024 / N > 0 ? 1024 / N : 1;
  int64_t vpb = (S + blocks - 1) / blocks;
  if (vpb < 64) vpb = 64;
  blocks = (S + vpb - 1) / vpb;
  hipLaunchKernelGGL(cosdist_stats_kernel, dim3((unsigned)blocks, N), dim3(256), 0, ST, x1, x2, stats, S, vpb);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
extern "C" int rehr_cosdist_bwd_f32(const float* x1, const float* x2, const double* stats, float* dx1, int32_t N,
                                    int64_t S, int32_t C, float scale, void* stream) {
  if (!x1 || !x2 || !stats || !dx1 || N < 1 || S < 1 || C != 64) return REHR_EINVAL;
  const int64_t total = (int64_t)N * S;
  int64_t blocks = (total + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(cosdist_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, ST, x1, x2, stats, dx1, S, total, scale);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_act_fwd_f32(const float* x, float* y, int64_t n, int32_t act, float slope,
                                void* stream) {
  if (!x || !y || n < 4 || n % 4 || !aligned16(x) || !aligned16(y)) return REHR_EINVAL;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_blocks(n / 4)), dim3(EW_THREADS), 0, ST, x, y, n / 4, act,
                     slope);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
extern "C" int rehr_channel_sum_actgrad_f32(const float* dy, const float* y, int32_t ld, int64_t rows, int32_t C,
                                            int32_t act, float slope, float* out, double* scratch, void* stream);
extern "C" int rehr_channel_sum_actgrad_bf16(const void* dy, const void* y, int32_t ld, int64_t rows, int32_t C,
                                             int32_t act, float slope, float* out, double* scratch, void* stream);

extern "C" int rehr_copy_channels_f32(const float* x, int32_t ldx, float* y, int32_t ldy, int64_t rows,
                                      int32_t C, void* stream) {
  if (!x || !y || rows < 1 || C < 4 || C % 4 || ldx % 4 || ldy % 4 || !aligned16(x) || !aligned16(y))
    return REHR_EINVAL;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(ew_blocks(rows * (C / 4))), dim3(EW_THREADS), 0, ST, x,
                     ldx, y, ldy, rows, C);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_nchw_to_nhwc_f32(const float* x, float* y, int32_t N, int32_t C, int64_t S,
                                     void* stream) {
  if (!x || !y || N < 1 || N > 65535 || C < 1 || S < 1 || (C + 31) / 32 > 65535) return REHR_EINVAL;
  hipLaunchKernelGGL(transpose_cs_kernel, dim3((unsigned)((S + 31) / 32), (C + 31) / 32, N), dim3(256), 0,
                     ST, x, y, C, S, 1);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
extern "C" int rehr_nhwc_to_nchw_f32(const float* x, float* y, int32_t N, int32_t C, int64_t S,
                                     void* stream) {
  if (!x || !y || N < 1 || N > 65535 || C < 1 || S < 1 || (C + 31) / 32 > 65535) return REHR_EINVAL;
  hipLaunchKernelGGL(transpose_cs_kernel, dim3((unsigned)((S + 31) / 32), (C + 31) / 32, N), dim3(256), 0,
                     ST, x, y, C, S, 0);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

// ---------------------------------------------------------------- split-K combine
// y[row][c] = act(bias[c] + sum_s slabs[s][row][c])   (few-tile, many-tap contractions
// such as feature_fuse run as S partial launches over tap ranges; fixed summation order)
namespace {
// TO = float or __bf16: the combined value is stored in the activations' dtype (mixed precision: fp32 slabs from the
// bf16 gather-GEMM with REHR_GG_Y_F32, bf16 activations out; store4 in common.h)

template <typename TO>
__global__ void sum_slabs_bias_act_kernel(const float* __restrict__ slabs, int S, int64_t slab_stride,
                                          const float* __restrict__ bias, TO* __restrict__ y, int64_t rows,
                                          int C, int act, float slope) {
  const int c4n = C >> 2;
  const int64_t total = rows * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) v = *reinterpret_cast<const f32x4*>(bias + c);
    for (int s = 0; s < S; ++s) v += reinterpret_cast<const f32x4*>(slabs + s * slab_stride)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act, slope);
    store4(y + 4 * i, v);
  }
}
}  // namespace

namespace {
// the same combine for one sample per blockIdx.y, with the per-(n,c) sum / sum of squares of the
// stored value (SE pool / InstanceNorm statistics of a split-K layer)
template <typename TO>
__global__ void sum_slabs_stats_kernel(const float* __restrict__ slabs, int S, int64_t slab_stride,
                                       const float* __restrict__ bias, TO* __restrict__ y, int64_t SV, int C,
                                       int64_t rows_per_block, int act, float slope, double* __restrict__ stats) {
  const int n = blockIdx.y;
  const int64_t s_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t s_end = s_begin + rows_per_block;
  if (s_end > SV) s_end = SV;
  const int64_t base = (int64_t)n * SV;
  column_reduce<2>(base + s_begin, base + s_end, C, stats + (int64_t)n * C * 2, 2,
                   [&](int64_t row, int c, double(&acc)[2][4]) {
                     f32x4 v = {0.f, 0.f, 0.f, 0.f};
                     if (bias != nullptr) v = *reinterpret_cast<const f32x4*>(bias + c);
                     for (int s = 0; s < S; ++s)
                       v += *reinterpret_cast<const f32x4*>(slabs + s * slab_stride + row * C + c);
#pragma unroll
                     for (int e = 0; e < 4; ++e) {
                       v[e] = apply_act(v[e], act, slope);
                       acc[0][e] += v[e];
                       acc[1][e] += (double)v[e] * v[e];
                     }
                     store4(y + row * C + c, v);   // (statistics: of the fp32 sums, as the conv epilogues form them)
                   });
}
}  // namespace

extern "C" int rehr_sum_slabs_stats_f32(const float* slabs, int32_t S, int64_t slab_stride, const float* bias,
                                        float* y, int32_t N, int64_t SV, int32_t C, int32_t act, float slope,
                                        double* stats, void* stream) {
  if (!slabs || !y || !stats || S < 1 || N < 1 || N > 65535 || SV < 1 || C < 4 || C % 4 || C > 1024 ||
      slab_stride < (int64_t)N * SV * C || slab_stride % 4)
    return REHR_EINVAL;
  if ((((uintptr_t)slabs) | ((uintptr_t)y)) & 15) return REHR_EINVAL;
  const int64_t rpb = rows_per_block_for(SV, C, N);
  const int blocks = (int)((SV + rpb - 1) / rpb);
  hipLaunchKernelGGL(sum_slabs_stats_kernel<float>, dim3(blocks, N), dim3(EW_THREADS), 0, (hipStream_t)stream, slabs, S,
                     slab_stride, bias, y, SV, C, rpb, act, slope, stats);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_sum_slabs_stats_bf16(const float* slabs, int32_t S, int64_t slab_stride, const float* bias,
                                         void* y, int32_t N, int64_t SV, int32_t C, int32_t act, float slope,
                                         double* stats, void* stream) {
  if (!slabs || !y || !stats || S < 1 || N < 1 || N > 65535 || SV < 1 || C < 4 || C % 4 || C > 1024 ||
      slab_stride < (int64_t)N * SV * C || slab_stride % 4)
    return REHR_EINVAL;
  if ((((uintptr_t)slabs) & 15) || (((uintptr_t)y) & 7)) return REHR_EINVAL;
  const int64_t rpb = rows_per_block_for(SV, C, N);
  const int blocks = (int)((SV + rpb - 1) / rpb);
  hipLaunchKernelGGL(sum_slabs_stats_kernel<__bf16>, dim3(blocks, N), dim3(EW_THREADS), 0, (hipStream_t)stream, slabs, S,
                     slab_stride, bias, (__bf16*)y, SV, C, rpb, act, slope, stats);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_sum_slabs_bias_act_bf16(const float* slabs, int32_t S, int64_t slab_stride, const float* bias,
                                            void* y, int64_t rows, int32_t C, int32_t act, float slope, void* stream) {
  if (!slabs || !y || S < 1 || rows < 1 || C < 4 || C % 4 || slab_stride < rows * C || slab_stride % 4) return REHR_EINVAL;
  if ((((uintptr_t)slabs) & 15) || (((uintptr_t)y) & 7)) return REHR_EINVAL;
  hipLaunchKernelGGL(sum_slabs_bias_act_kernel<__bf16>, dim3(ew_blocks(rows * (C / 4))), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, slabs, S, slab_stride, bias, (__bf16*)y, rows, C, act, slope);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_sum_slabs_bias_act_f32(const float* slabs, int32_t S, int64_t slab_stride, const float* bias,
                                           float* y, int64_t rows, int32_t C, int32_t act, float slope,
                                           void* stream) {
  if (!slabs || !y || S < 1 || rows < 1 || C < 4 || C % 4 || slab_stride < rows * C || slab_stride % 4) return REHR_EINVAL;
  if ((((uintptr_t)slabs) | ((uintptr_t)y)) & 15) return REHR_EINVAL;
  hipLaunchKernelGGL(sum_slabs_bias_act_kernel<float>, dim3(ew_blocks(rows * (C / 4))), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, slabs, S, slab_stride, bias, y, rows, C, act, slope);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
