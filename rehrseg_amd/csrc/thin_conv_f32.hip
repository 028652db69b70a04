// sr_head.2 -- Conv3d(16 -> 2, 5x5x5, stride 1, pad 2) on the depth-upsampled features (models/seg_model.py:199, :205)
// -- on the fp32 matrix cores (v_mfma_f32_16x16x4_f32): the fp32 counterpart of thin_conv_bf16.hip, same three
// formulations, same block organisation, fragments of one float per lane.
//
//   forward   P[v][kw][co] = sum_{kd,kh,ci} x[v + (kd-2, kh-2, 0)][ci] w[co][ci][kd][kh][kw],  y[v][co] = b + sum_kw P[v + kw-2][kw][co]
//             M = 16 voxels along w, N = (kw, co) = 10 -> 16, K = 4 channels per instruction; input stationary along
//             depth: every A fragment (one LDS dword per lane) feeds the five output planes it touches, the weight
//             fragments stream from L1 one group ahead, a plane is staged in two 8-channel halves (32 B per voxel in LDS).
//   dgrad     dx[v][ci] = sum dy[v + (kd-2, kh-2, kw-2)][co] w[co][ci][4-kd][4-kh][4-kw]
//             M = ci (75 weight fragments in registers), N = 16 voxels, K = (2 kw, 2 co) per instruction.
//   wgrad     acc[(kd, kw)][ci][(kh, co)] += sum_w x[dx][hx][w + kw-2][ci] dy[dx - kd+2][hx - kh+2][w][co]
//             M = ci, N = (kh, co) = 10 -> 16, K = 4 voxels per instruction; x is [w][ci] in LDS, so the A fragment is
//             a plain dword read (no transpose as for bf16); partial sums per block, fixed-order reduction.
// On the VALU kernels of direct_conv.hip the layer costs 7.0 ms of a 52.6 ms cfg-3 step (profiles/r02_seg_kernel_stats.csv:
// 2.33 + 1.94 + 2.77 ms); the fp32 matrix pipe has the vector pipe's peak, the gain comes from the 5-fold fragment reuse
// and from weights that never leave the registers.
#include "common.h"
#include <type_traits>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TF_CIN = 16, TF_K = 5, TF_BH = 4;
constexpr int TF_ROWS = TF_BH + TF_K - 1;
constexpr int TF_PP = 11;               // floats per voxel in the P row (10 used, odd pitch)
constexpr int TF_NW_FWD = 100;          // weight fragments of the forward: [kd][kh][channel quad]
constexpr int TF_NW_DG = 75;            // input gradient: [tap (kd,kh)][kw pair]
constexpr int TF_RING = 6;
constexpr int TF_PADW = 8;              // dY row of the input gradient: 2 zero voxels in front, 6 behind
constexpr int TF_SLAB = 25 * 16 * 16;   // weight-gradient partial sums per block: [kd*5 + kw][ci][kh*2 + co]
constexpr int TF_SLABF = TF_SLAB + 16;  // + column sums of dY

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ------------------------------------------------------------------------------------------------------------ forward
struct ThinF32Params {
  const float* x;      // [N][D][H][W][ldx]
  int ldx, N, D, H, W;
  const float* wfrag;  // [TF_NW_FWD][64]
  const float* bias;
  float* y;            // [N][D][H][W][ldy]
  int ldy;
  int dseg, nseg, nstrip;
};

// fragment f = (kd*5 + kh)*4 + cq, lane (n = l & 15, kq = l >> 4): B[k = kq][n] = w[co][4 cq + kq][kd][kh][kw], n = kw*2 + co
__global__ void thinf_pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= TF_NW_FWD * 64) return;
  const int l = i & 63, f = i >> 6;
  const int cq = f & 3, kh = (f >> 2) % TF_K, kd = (f >> 2) / TF_K;
  const int n = l & 15, kq = l >> 4, kw = n >> 1, co = n & 1;
  out[i] = (n < 2 * TF_K) ? w[(((co * TF_CIN + 4 * cq + kq) * TF_K + kd) * TF_K + kh) * TF_K + kw] : 0.f;
}

template <int NTW>
__global__ __launch_bounds__(512) void thinf_fwd_kernel(const ThinF32Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int W = NTW * 32;
  constexpr int plane_bytes = TF_ROWS * W * 32;                     // 8 channels x 4 B per voxel
  constexpr int PROW = (W + 4) * TF_PP;
  unsigned char* xs = smem;                                         // [2][TF_ROWS][W][8] fp32
  float* prow = reinterpret_cast<float*>(smem + 2 * plane_bytes);   // [2][4 rows][W + 4][TF_PP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = wave & 3, hf = wave >> 2;
  int b = blockIdx.x;
  const int seg = b % p.nseg; b /= p.nseg;
  const int strip = b % p.nstrip;
  const int n_img = b / p.nstrip;
  const int h0 = strip * TF_BH;
  const int d0 = seg * p.dseg, d1 = min(p.D, d0 + p.dseg);

  // staging: 16-byte pieces of (plane, channel half): piece = (row, w, quad); NTW pieces per thread
  const uint32_t img_bytes = (uint32_t)p.D * p.H * p.W * p.ldx * 4u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x) + (int64_t)n_img * p.D * p.H * p.W * p.ldx, 0, img_bytes, 0x00020000);
  uint32_t poff[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int piece = tid + 512 * i;
    const int q2 = piece & 1, w = (piece >> 1) % W, row = (piece >> 1) / W;
    const int ih = h0 - 2 + row;
    poff[i] = ((unsigned)ih < (unsigned)p.H) ? ((uint32_t)(ih * p.W + w) * p.ldx + q2 * 4) * 4u : img_bytes;
  }
  const uint32_t dplane = (uint32_t)p.H * p.W * p.ldx * 4u;
  u32x4 rx[NTW];
  auto fetch = [&](int dp, int hc) {   // channel half hc of plane dp
    const bool dok = (unsigned)dp < (unsigned)p.D;
#pragma unroll
    for (int i = 0; i < NTW; ++i)
      rx[i] = __builtin_amdgcn_raw_buffer_load_b128(
          rs, (dok && poff[i] != img_bytes) ? (uint32_t)dp * dplane + poff[i] + (uint32_t)hc * 32u : img_bytes, 0, 0);
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NTW; ++i) *reinterpret_cast<u32x4*>(xs + buf * plane_bytes + (tid + 512 * i) * 16) = rx[i];
  };

  const int m = lane & 15, kq = lane >> 4;
  const int abase = (r * W + hf * (W / 2) + m) * 32 + kq * 4;   // + kh * W * 32 + cql * 16 + t * 16 * 32
  if (hf == 0) {  // the two zero voxels on either side of both copies of this row's P line
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float* q = prow + (c * 4 + r) * PROW;
      if (lane < 2 * TF_PP) q[lane] = 0.f;
      if (lane < 2 * TF_PP) q[(W + 2) * TF_PP + lane] = 0.f;
    }
  }
  const float bias_v = p.bias ? p.bias[lane & 1] : 0.f;
  const int oh = h0 + r;

  f32x4 acc[5][NTW];
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  fetch(d0 - 2, 0);
  stage(0);
  __syncthreads();

  int buf = 0, pbuf = 0;
  const __amdgpu_buffer_rsrc_t rsw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wfrag), 0, TF_NW_FWD * 64 * 4, 0x00020000);
  const uint32_t wlane = (uint32_t)lane * 4u;
  auto half_step = [&](auto S, auto HC) {
    constexpr int S0 = decltype(S)::value, hc = decltype(HC)::value;
    const unsigned char* xb = xs + buf * plane_bytes;
    // group g = (kh, channel quad of this half): 5 weight fragments (kd, from L1 two groups ahead) and one A dword per
    // tile (from LDS one group ahead); 5 * NTW MFMAs per group
    float bw[3][TF_K], a[2][NTW];
    auto ldw = [&](float (&dst)[TF_K], const int g_) {
#pragma unroll
      for (int kd = 0; kd < TF_K; ++kd)
        dst[kd] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                      rsw, wlane, ((kd * TF_K + (g_ >> 1)) * 4 + hc * 2 + (g_ & 1)) * 256, 0));
    };
    auto lda = [&](float (&dst)[NTW], const int g_) {
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        dst[t] = *reinterpret_cast<const float*>(xb + abase + (g_ >> 1) * W * 32 + (g_ & 1) * 16 + t * 16 * 32);
    };
    ldw(bw[0], 0);
    ldw(bw[1], 1);
    lda(a[0], 0);
#pragma unroll
    for (int g = 0; g < 2 * TF_K; ++g) {
      if (g + 2 < 2 * TF_K) ldw(bw[(g + 2) % 3], g + 2);
      if (g + 1 < 2 * TF_K) lda(a[(g + 1) & 1], g + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int kd = 0; kd < TF_K; ++kd) {
          const int s = (S0 - kd + 5) % 5;   // output plane dp - kd + 2
          acc[s][t] = mfma4(a[g & 1][t], bw[g % 3][kd], acc[s][t]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto plane_step = [&](const int dp, auto S) {
    constexpr int S0 = decltype(S)::value;
    if (dp > d1 + 1) return;   // (block-uniform)
    fetch(dp, 1);
    half_step(S, std::integral_constant<int, 0>{});
    stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
    fetch(dp + 1, 0);
    half_step(S, std::integral_constant<int, 1>{});
    // output plane dp - 2 (kd = 4) is complete: slot (S0 + 1) mod 5
    constexpr int SD = (S0 + 1) % 5;
    const int dout = dp - 2;
    const bool live = (dout >= d0) & (dout < d1) & (oh < p.H);
    float* pw_ = prow + ((pbuf * 4) + r) * PROW;
    if (live && m < 2 * TF_K) {
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) pw_[(2 + hf * (W / 2) + t * 16 + kq * 4 + i) * TF_PP + m] = acc[SD][t][i];
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[SD][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
    if (live) {  // shift-add over kw
      float* yrow = p.y + ((((int64_t)n_img * p.D + dout) * p.H + oh) * p.W) * p.ldy;
      const int co = lane & 1;
#pragma unroll
      for (int it = 0; it < (NTW + 1) / 2; ++it) {
        const int wl = it * 32 + (lane >> 1);
        const int w = hf * (W / 2) + wl;
        if (wl < W / 2) {
          float s = bias_v;
#pragma unroll
          for (int kw = 0; kw < TF_K; ++kw) s += pw_[(w + kw) * TF_PP + kw * 2 + co];
          yrow[(int64_t)w * p.ldy + co] = s;
        }
      }
    }
    pbuf ^= 1;
  };

  for (int dp = d0 - 2; dp <= d1 + 1; dp += 5) {
    plane_step(dp, std::integral_constant<int, 0>{});
    plane_step(dp + 1, std::integral_constant<int, 1>{});
    plane_step(dp + 2, std::integral_constant<int, 2>{});
    plane_step(dp + 3, std::integral_constant<int, 3>{});
    plane_step(dp + 4, std::integral_constant<int, 4>{});
  }
}

// ------------------------------------------------------------------------------------------------------- input gradient
struct ThinF32DgradParams {
  const float* dy;     // [N][D][H][W][ldy]
  int ldy, N, D, H, W;
  const float* wfrag;  // [TF_NW_DG][64]
  float* dx;           // [N][D][H][W][lddx]
  int lddx;
  int dseg, nseg, nstrip;
};

// fragment f = tap*3 + pr, lane (m = ci = l & 15, kq = l >> 4): A[m][k = kq], kw = 2 pr + (kq >> 1), co = kq & 1,
// tap = kd*5 + kh (all flipped)
__global__ void thinf_pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= TF_NW_DG * 64) return;
  const int l = i & 63, f = i >> 6;
  const int pr = f % 3, tap = f / 3, kd = tap / TF_K, kh = tap % TF_K;
  const int ci = l & 15, kq = l >> 4, kw = 2 * pr + (kq >> 1), co = kq & 1;
  out[i] = (kw < TF_K) ? w[(((co * TF_CIN + ci) * TF_K + (4 - kd)) * TF_K + (4 - kh)) * TF_K + (4 - kw)] : 0.f;
}

template <int NTW>
__global__ __launch_bounds__(512) void thinf_dgrad_kernel(const ThinF32DgradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int W = NTW * 32;
  constexpr int ROWB = (W + TF_PADW) * 8;            // bytes per dY row (2 floats per voxel)
  constexpr int PLANEB = TF_ROWS * ROWB;
  constexpr int NPIECE = (TF_ROWS * W + 511) / 512;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = wave & 3, hf = wave >> 2;
  int b = blockIdx.x;
  const int seg = b % p.nseg; b /= p.nseg;
  const int strip = b % p.nstrip;
  const int n_img = b / p.nstrip;
  const int h0 = strip * TF_BH;
  const int d0 = seg * p.dseg, d1 = min(p.D, d0 + p.dseg);

  float wf[TF_NW_DG];
#pragma unroll
  for (int f = 0; f < TF_NW_DG; ++f) wf[f] = p.wfrag[f * 64 + lane];

  for (int i = tid; i < TF_RING * TF_ROWS * TF_PADW; i += 512) {   // row pads of every ring slot, once
    const int pv = i % TF_PADW, row = i / TF_PADW;
    const int vox = pv < 2 ? pv : W + pv;
    *reinterpret_cast<float2*>(smem + row * ROWB + vox * 8) = make_float2(0.f, 0.f);
  }

  const float* dyn = p.dy + (int64_t)n_img * p.D * p.H * p.W * p.ldy;
  int pvox[NPIECE], plds[NPIECE];
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) {
    const int piece = tid + 512 * i;
    const int w = piece % W, row = piece / W;
    const int ih = h0 - 2 + row;
    const bool ok = (piece < TF_ROWS * W) & ((unsigned)ih < (unsigned)p.H);
    pvox[i] = ok ? ih * p.W + w : -1;
    plds[i] = (piece < TF_ROWS * W) ? row * ROWB + (w + 2) * 8 : -1;
  }
  float2 rx[NPIECE];
  auto fetch = [&](int dp) {
    const bool dok = (unsigned)dp < (unsigned)p.D;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      rx[i] = make_float2(0.f, 0.f);
      if (dok && pvox[i] >= 0) {
        const float* q = dyn + ((int64_t)dp * p.H * p.W + pvox[i]) * p.ldy;
        rx[i] = make_float2(q[0], q[1]);
      }
    }
  };
  auto stage = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i)
      if (plds[i] >= 0) *reinterpret_cast<float2*>(smem + slot * PLANEB + plds[i]) = rx[i];
  };

  // B fragment of this lane: voxel hf*W/2 + 16 t + n + kw (stored index: + 2 pad - 2 shift), kw = 2 pr + (kq >> 1), channel kq & 1
  const int n = lane & 15, kq = lane >> 4;
  const int lbase = (r * (W + TF_PADW) + hf * (W / 2) + n + (kq >> 1)) * 8 + (kq & 1) * 4;   // + kh * ROWB + pr * 16 + t * 128

  for (int s = 0; s < 5; ++s) {
    fetch(d0 - 2 + s);
    stage(s);
  }
  __syncthreads();

  float* dxn = p.dx + (int64_t)n_img * p.D * p.H * p.W * p.lddx;
  const int oh = h0 + r;

  auto plane_step = [&](const int d, auto PH) {
    constexpr int P = decltype(PH)::value;
    if (d >= d1) return;   // (block-uniform)
    fetch(d + 3);
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < TF_K * TF_K; ++tap) {
        const int off = ((P + tap / TF_K) % TF_RING) * PLANEB + (tap % TF_K) * ROWB;
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
          const float bv = *reinterpret_cast<const float*>(smem + off + lbase + pr * 16 + t * 128);
          acc[t] = mfma4(wf[tap * 3 + pr], bv, acc[t]);
        }
      }
    }
    if (oh < p.H) {
      float* row = dxn + (((int64_t)d * p.H + oh) * p.W + hf * (W / 2) + n) * p.lddx + 4 * kq;
#pragma unroll
      for (int t = 0; t < NTW; ++t) *reinterpret_cast<f32x4*>(row + (int64_t)t * 16 * p.lddx) = acc[t];
    }
    stage((P + 5) % TF_RING);
    __syncthreads();
  };

  for (int d = d0; d < d1; d += 6) {
    plane_step(d, std::integral_constant<int, 0>{});
    plane_step(d + 1, std::integral_constant<int, 1>{});
    plane_step(d + 2, std::integral_constant<int, 2>{});
    plane_step(d + 3, std::integral_constant<int, 3>{});
    plane_step(d + 4, std::integral_constant<int, 4>{});
    plane_step(d + 5, std::integral_constant<int, 5>{});
  }
}

// ------------------------------------------------------------------------------------------------------ weight gradient
struct ThinF32WgradParams {
  const float* x;      // [N][D][H][W][ldx]
  const float* dy;     // [N][D][H][W][ldy]
  int ldx, ldy, N, D, H, W;
  float* slabs;        // [blocks][TF_SLABF]
  int dseg, nseg, nstrip;
};

// 2 NW waves: wave = (32-voxel chunk c of the row, row pair rp); the four x rows of a plane step are independent in this
// loop (every row reads its own A and B fragments), so the split costs nothing and gives every SIMD a second wave
template <int NW>
__global__ __launch_bounds__(128 * NW) void thinf_wgrad_kernel(const ThinF32WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int W = NW * 32, NTHR = 128 * NW;
  constexpr int XROWB = (W + 4) * 64;                 // x row: 16 channels x 4 B per voxel, 2 zero voxels either side
  constexpr int XBUF = TF_BH * XROWB;
  constexpr int DROWB = W * 4 + 16;                   // one (row, co) line of dY
  constexpr int DPLANE = TF_ROWS * 2 * DROWB;
  unsigned char* xs = smem;                           // [2][4 rows][W + 4][16]
  unsigned char* ds = smem + 2 * XBUF;                // [6][8 rows][2 co][W (+4)]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv2 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = wv2 % NW, rp = wv2 / NW;                    // 32-voxel chunk of the row, row pair
  int b = blockIdx.x;
  const int seg = b % p.nseg; b /= p.nseg;
  const int strip = b % p.nstrip;
  const int n_img = b / p.nstrip;
  const int h0 = strip * TF_BH;
  const int d0 = seg * p.dseg, d1 = min(p.D, d0 + p.dseg);

  for (int i = tid; i < 2 * TF_BH * 4 * 4; i += NTHR) {   // zero pads of the x rows (both buffers), once
    const int q4 = i & 3, pv = (i >> 2) & 3, row = i >> 4;
    const int vox = pv < 2 ? pv : W + pv;
    *reinterpret_cast<u32x4*>(xs + row * XROWB + vox * 64 + q4 * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  // x staging: 4 rows x W voxels x 4 quads = 4 pieces of 16 bytes per thread
  const uint32_t img_bytes = (uint32_t)p.D * p.H * p.W * p.ldx * 4u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x) + (int64_t)n_img * p.D * p.H * p.W * p.ldx, 0, img_bytes, 0x00020000);
  const uint32_t dplane = (uint32_t)p.H * p.W * p.ldx * 4u;
  uint32_t xoff[4];
  int xlds[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = tid + NTHR * i;
    const int q4 = piece & 3, w = (piece >> 2) % W, row = (piece >> 2) / W;
    const int ih = h0 + row;
    xoff[i] = (ih < p.H) ? ((uint32_t)(ih * p.W + w) * p.ldx + q4 * 4) * 4u : img_bytes;
    xlds[i] = row * XROWB + (w + 2) * 64 + q4 * 16;
  }
  u32x4 rxx[4];
  auto fetch_x = [&](int dp) {
    const bool dok = (unsigned)dp < (unsigned)p.D;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      rxx[i] = __builtin_amdgcn_raw_buffer_load_b128(
          rs, (dok && xoff[i] != img_bytes) ? (uint32_t)dp * dplane + xoff[i] : img_bytes, 0, 0);
  };
  auto stage_x = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(xs + buf * XBUF + xlds[i]) = rxx[i];
  };
  // dY staging: 8 rows x W voxels, 2 voxels per thread
  const float* dyn = p.dy + (int64_t)n_img * p.D * p.H * p.W * p.ldy;
  int yvox[2], ylds[2];
  bool yown[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int v = tid + NTHR * i;
    const int w = v % W, row = v / W;
    const int ih = h0 - 2 + row;
    yvox[i] = ((unsigned)ih < (unsigned)p.H) ? ih * p.W + w : -1;
    ylds[i] = row * 2 * DROWB + w * 4;
    yown[i] = (row >= 2) & (row < 2 + TF_BH);
  }
  float2 ry[2];
  float db0 = 0.f, db1 = 0.f;
  auto fetch_y = [&](int dp) {
    const bool dok = (unsigned)dp < (unsigned)p.D;
    const bool down = (dp >= d0) & (dp < d1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ry[i] = make_float2(0.f, 0.f);
      if (dok && yvox[i] >= 0) {
        const float* q = dyn + ((int64_t)dp * p.H * p.W + yvox[i]) * p.ldy;
        ry[i] = make_float2(q[0], q[1]);
        if (down && yown[i]) { db0 += ry[i].x; db1 += ry[i].y; }
      }
    }
  };
  auto stage_y = [&](int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<float*>(ds + slot * DPLANE + ylds[i]) = ry[i].x;
      *reinterpret_cast<float*>(ds + slot * DPLANE + ylds[i] + DROWB) = ry[i].y;
    }
  };

  // A: lane (m = ci, kq) reads x[row][32c + 4j + kq + kw][ci] (stored voxel index = w + kw: the +2 pad and the -2 shift cancel)
  const int mm = lane & 15, kq = lane >> 4;
  const int abase = (32 * c + kq) * 64 + mm * 4;       // + row * XROWB + (4 j + kw) * 64
  // B: lane (n = kh*2 + co, kq) reads dy[row r + 4 - kh][co][32c + 4j + kq]
  const int khl = (mm >> 1) < TF_K ? (mm >> 1) : TF_K - 1, col = mm & 1;
  const int bbase = ((4 - khl) * 2 + col) * DROWB + (32 * c + kq) * 4;   // + r * 2 * DROWB + j * 16

  f32x4 acc[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int s = 0; s < 5; ++s) {
    fetch_y(d0 - 2 + s);
    stage_y(s);
  }
  fetch_x(d0);
  stage_x(0);
  __syncthreads();

  int buf = 0;
  auto plane_step = [&](const int dx, auto PH) {
    constexpr int P = decltype(PH)::value;
    if (dx >= d1) return;   // (block-uniform)
    fetch_x(dx + 1);
    fetch_y(dx + 3);
    const unsigned char* xb = xs + buf * XBUF;
#pragma unroll
    for (int r2 = 0; r2 < TF_BH / 2; ++r2) {
      const int r = rp * (TF_BH / 2) + r2;
#pragma unroll 2
      for (int j = 0; j < 8; ++j) {
        float a[TF_K];
#pragma unroll
        for (int kw = 0; kw < TF_K; ++kw) a[kw] = *reinterpret_cast<const float*>(xb + r * XROWB + (4 * j + kw) * 64 + abase);
#pragma unroll
        for (int kd = 0; kd < TF_K; ++kd) {
          const int slot = (P + 4 - kd) % TF_RING;
          const float bv = *reinterpret_cast<const float*>(ds + slot * DPLANE + r * 2 * DROWB + j * 16 + bbase);
#pragma unroll
          for (int kw = 0; kw < TF_K; ++kw) acc[kd * 5 + kw] = mfma4(a[kw], bv, acc[kd * 5 + kw]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    stage_x(buf ^ 1);
    stage_y((P + 5) % TF_RING);
    __syncthreads();
    buf ^= 1;
  };

  for (int dx = d0; dx < d1; dx += 6) {
    plane_step(dx, std::integral_constant<int, 0>{});
    plane_step(dx + 1, std::integral_constant<int, 1>{});
    plane_step(dx + 2, std::integral_constant<int, 2>{});
    plane_step(dx + 3, std::integral_constant<int, 3>{});
    plane_step(dx + 4, std::integral_constant<int, 4>{});
    plane_step(dx + 5, std::integral_constant<int, 5>{});
  }

  // block sum over the waves (LDS), then the slab; register i of lane (n = l & 15, q = l >> 4) is C[ci = 4q + i][n]
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
  float* slab = p.slabs + (int64_t)blockIdx.x * TF_SLABF;
  {
    db0 = wave_sum(db0);
    db1 = wave_sum(db1);
    if (lane == 0) { red[wv2 * 2] = db0; red[wv2 * 2 + 1] = db1; }
    __syncthreads();
    if (tid < 2) {
      float sdb = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 2 * NW; ++w2) sdb += red[w2 * 2 + tid];
      slab[TF_SLAB + tid] = sdb;
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 25; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) red[(wv2 * 16 + 4 * kq + i) * 16 + mm] = acc[t][i];
    __syncthreads();
    for (int e = tid; e < 256; e += NTHR) {
      float s = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 2 * NW; ++w2) s += red[w2 * 256 + e];
      slab[t * 256 + e] = s;
    }
    __syncthreads();
  }
}

// 8 threads share an output element (thread (e, g) sums the blocks g, g + 8, ...; fixed-order combine through LDS): one
// thread per element walking every block's slab was a 0.29 ms latency chain at 160^3
__global__ __launch_bounds__(256) void thinf_wgrad_reduce_kernel(const float* __restrict__ slabs, int nblocks, float* __restrict__ dw,
                                                             float* __restrict__ dbias) {
  constexpr int G = 8, EPB = 256 / G, NDW = 2 * TF_CIN * 125;
  __shared__ float part[256];
  const int el = threadIdx.x % EPB, g = threadIdx.x / EPB;
  const int i = blockIdx.x * EPB + el;   // index into dw (2,16,5,5,5), then the two bias gradients
  int si = -1;
  if (i < NDW) {
    const int kw = i % 5, kh = (i / 5) % 5, kd = (i / 25) % 5, ci = (i / 125) % TF_CIN, co = i / (125 * TF_CIN);
    si = ((kd * 5 + kw) * 16 + ci) * 16 + kh * 2 + co;
  } else if (i < NDW + 2) {
    si = TF_SLAB + (i - NDW);
  }
  float s0 = 0.f, s1 = 0.f;
  if (si >= 0) {
    int bq = g;
    for (; bq + G < nblocks; bq += 2 * G) {
      s0 += slabs[(int64_t)bq * TF_SLABF + si];
      s1 += slabs[(int64_t)(bq + G) * TF_SLABF + si];
    }
    for (; bq < nblocks; bq += G) s0 += slabs[(int64_t)bq * TF_SLABF + si];
  }
  part[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (g == 0 && si >= 0) {
    float s = part[el];
#pragma unroll
    for (int q = 1; q < G; ++q) s += part[q * EPB + el];
    if (i < NDW) dw[i] = s;
    else if (dbias != nullptr) dbias[i - NDW] = s;
  }
}

bool thinf_shape_ok(const rehr_direct_conv_desc& d) {
  return d.Cin == TF_CIN && d.Cout == 2 && d.KD == TF_K && d.KH == TF_K && d.KW == TF_K && d.sd == 1 && d.sh == 1 &&
         d.sw == 1 && d.pd == 2 && d.ph == 2 && d.pw == 2 && d.Do == d.Di && d.Ho == d.Hi && d.Wo == d.Wi &&
         d.Wi % 32 == 0 && d.Wi >= 32 && d.Wi <= 128 && d.ldx % 4 == 0 && d.ldx >= TF_CIN && d.ldy >= 2 &&
         (int64_t)d.Di * d.Hi * d.Wi * d.ldx * 4 < ((int64_t)1 << 32);   // (W = 160 would spill in the forward kernel)
}

void thinf_segments(const rehr_direct_conv_desc& d, int& nstrip, int& dseg, int& nseg) {
  nstrip = (d.Hi + TF_BH - 1) / TF_BH;
  int ns = 1;
  while ((int64_t)d.N * nstrip * ns < 512 && d.Di / (ns * 2) >= 16) ns *= 2;
  dseg = (d.Di + ns - 1) / ns;
  nseg = (d.Di + dseg - 1) / dseg;
}

}  // namespace

extern "C" int64_t rehr_conv5_thin_f32_workspace_bytes(const rehr_direct_conv_desc* dp) {
  if (dp == nullptr) return REHR_EINVAL;
  if (!thinf_shape_ok(*dp)) return REHR_ENOSUP;
  int nstrip, dseg, nseg;
  thinf_segments(*dp, nstrip, dseg, nseg);
  const int64_t slabs = (int64_t)dp->N * nstrip * nseg * TF_SLABF * 4;
  const int64_t pack = (int64_t)TF_NW_FWD * 64 * 4;
  return slabs > pack ? slabs : pack;
}

extern "C" int rehr_conv5_thin_f32_supported(const rehr_direct_conv_desc* d) {
  return d != nullptr && thinf_shape_ok(*d) ? 1 : 0;
}

#define TF_SWITCH(KERNEL, THREADS_OF)                                                                               \
  switch (d.Wi / 32) {                                                                                              \
    case 1: TF_LAUNCH(KERNEL, 1, THREADS_OF(1)); break;                                                             \
    case 2: TF_LAUNCH(KERNEL, 2, THREADS_OF(2)); break;                                                             \
    case 3: TF_LAUNCH(KERNEL, 3, THREADS_OF(3)); break;                                                             \
    case 4: TF_LAUNCH(KERNEL, 4, THREADS_OF(4)); break;                                                             \
    default: return REHR_ENOSUP;                                                                                    \
  }
#define TF_LAUNCH(KERNEL, NT_, THREADS)                                                                             \
  do {                                                                                                              \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL<NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)smem);                                                                           \
    hipLaunchKernelGGL(KERNEL<NT_>, dim3((unsigned)blocks), dim3(THREADS), smem, st, p);                            \
  } while (0)
#define TF_T512(n) 512
#define TF_T64N(n) (128 * (n))

extern "C" int rehr_conv5_thin_fwd_f32(const rehr_direct_conv_desc* dp, void* workspace, int64_t workspace_bytes,
                                       void* stream) {
  if (dp == nullptr || dp->x == nullptr || dp->w == nullptr || dp->y == nullptr || workspace == nullptr) return REHR_EINVAL;
  const rehr_direct_conv_desc& d = *dp;
  if (!thinf_shape_ok(d)) return REHR_ENOSUP;
  if (workspace_bytes < rehr_conv5_thin_f32_workspace_bytes(dp) || d.act != REHR_ACT_NONE || d.stats_mode != 0) return REHR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(thinf_pack_fwd_kernel, dim3((TF_NW_FWD * 64 + 255) / 256), dim3(256), 0, st, d.w,
                     reinterpret_cast<float*>(workspace));
  ThinF32Params p;
  p.x = d.x; p.ldx = d.ldx; p.N = d.N; p.D = d.Di; p.H = d.Hi; p.W = d.Wi;
  p.wfrag = reinterpret_cast<const float*>(workspace);
  p.bias = d.bias; p.y = d.y; p.ldy = d.ldy;
  thinf_segments(d, p.nstrip, p.dseg, p.nseg);
  const int64_t blocks = (int64_t)d.N * p.nstrip * p.nseg;
  const size_t smem = (size_t)2 * TF_ROWS * d.Wi * 32 + (size_t)2 * TF_BH * (d.Wi + 4) * TF_PP * 4;
  TF_SWITCH(thinf_fwd_kernel, TF_T512)
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_conv5_thin_dgrad_f32(const rehr_direct_conv_desc* dp, float* dx, int32_t lddx, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  if (dp == nullptr || dp->w == nullptr || dp->y == nullptr || dx == nullptr || workspace == nullptr) return REHR_EINVAL;
  const rehr_direct_conv_desc& d = *dp;
  if (!thinf_shape_ok(d)) return REHR_ENOSUP;
  if (workspace_bytes < rehr_conv5_thin_f32_workspace_bytes(dp) || lddx < TF_CIN || lddx % 4) return REHR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(thinf_pack_dgrad_kernel, dim3((TF_NW_DG * 64 + 255) / 256), dim3(256), 0, st, d.w,
                     reinterpret_cast<float*>(workspace));
  ThinF32DgradParams p;
  p.dy = d.y; p.ldy = d.ldy; p.N = d.N; p.D = d.Di; p.H = d.Hi; p.W = d.Wi;
  p.wfrag = reinterpret_cast<const float*>(workspace);
  p.dx = dx; p.lddx = lddx;
  thinf_segments(d, p.nstrip, p.dseg, p.nseg);
  const int64_t blocks = (int64_t)d.N * p.nstrip * p.nseg;
  const size_t smem = (size_t)TF_RING * TF_ROWS * (d.Wi + TF_PADW) * 8;
  TF_SWITCH(thinf_dgrad_kernel, TF_T512)
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_conv5_thin_wgrad_f32(const rehr_direct_conv_desc* dp, float* dw, float* dbias, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  if (dp == nullptr || dp->x == nullptr || dp->y == nullptr || dw == nullptr || workspace == nullptr) return REHR_EINVAL;
  const rehr_direct_conv_desc& d = *dp;
  if (!thinf_shape_ok(d)) return REHR_ENOSUP;
  if (workspace_bytes < rehr_conv5_thin_f32_workspace_bytes(dp)) return REHR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  ThinF32WgradParams p;
  p.x = d.x; p.dy = d.y; p.ldx = d.ldx; p.ldy = d.ldy; p.N = d.N; p.D = d.Di; p.H = d.Hi; p.W = d.Wi;
  p.slabs = reinterpret_cast<float*>(workspace);
  thinf_segments(d, p.nstrip, p.dseg, p.nseg);
  const int64_t blocks = (int64_t)d.N * p.nstrip * p.nseg;
  const size_t smem = (size_t)2 * TF_BH * (d.Wi + 4) * 64 + (size_t)TF_RING * TF_ROWS * 2 * (d.Wi * 4 + 16);
  TF_SWITCH(thinf_wgrad_kernel, TF_T64N)
  REHR_LAUNCH_CHECK();
  hipLaunchKernelGGL(thinf_wgrad_reduce_kernel, dim3((2 * TF_CIN * 125 + 2 + 31) / 32), dim3(256), 0, st, p.slabs,
                     (int)blocks, dw, dbias);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
