// sr_head.2 -- Conv3d(16 -> 2, 5x5x5, stride 1, pad 2) on the 4x depth-upsampled features (models/seg_model.py:199,
// :205) -- on the bf16 matrix cores for the mixed-precision path (BASELINE.json configs[4]).  On the VALU kernels
// of direct_conv.hip this one layer is 7 ms of a 31 ms mixed-precision step (profiles/r02_seg_bf16_kernel_stats.csv).
//
// Two output channels are far too thin for an MFMA N dimension, so the kw taps are moved there:
//   P[v][kw][co] = sum_{kd,kh,ci} x[v + (kd-2, kh-2, 0)][ci] * w[co][ci][kd][kh][kw]     GEMM: M = voxels along w,
//   y[v][co]     = b[co] + sum_kw P[v + (0, 0, kw-2)][kw][co]                             N = (kw, co) = 10 -> 16,
// K = (kd, kh, ci) = 400, v_mfma_f32_16x16x32_bf16 with two kh taps x 16 channels per instruction (kh padded to 6).
// P outside the row is zero (x is), so P is only formed for the W voxels of a row: no halo along w.
//
// Forward kernel (input stationary along depth): a block owns 4 output rows x W of one depth segment (wave = row) and
// marches over the INPUT planes; the plane (8 rows, 32 B per voxel) is staged once in LDS and every A fragment -- one
// aligned 16-byte LDS read per lane, conflict-free in the plain [row][w][16] layout -- feeds the five output planes
// it touches (kd = 0..4), whose accumulators (5 planes x W/16 tiles x 4 registers) rotate through a five-phase
// unrolled loop.  All 15 weight fragments stay in registers.  A finished plane's P goes through a per-wave LDS row
// for the shift-add over kw and leaves as 8 bytes per voxel.
#include "common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int TC_CIN = 16, TC_K = 5, TC_BH = 4;   // channels, kernel extent, output rows per block (= waves)
constexpr int TC_ROWS = TC_BH + TC_K - 1;          // staged input rows per plane
constexpr int TC_NFRAG = 15;                       // weight fragments: kd x kh-pair
constexpr int TC_PP = 11;                          // floats per voxel in the P row (10 used + 1: odd pitch, conflict-free reads)

struct ThinParams {
  const void* x;       // bf16 [N][D][H][W][ldx]
  int ldx, N, D, H, W;
  const u32x4* wfrag;  // [TC_NFRAG][64 lanes] 8 bf16 each
  const float* bias;
  float* y;            // fp32 [N][D][H][W][ldy]
  int ldy;
  int dseg, nseg, nstrip;
};

// weight fragment f = kd * 3 + khp, lane (n = l & 15, kb = l >> 4): B[k = kb*8 + j][n] with kh = 2*khp + (kb >> 1),
// ci = (kb & 1) * 8 + j, n = kw * 2 + co; zero for kh > 4 or n >= 10
__global__ void thin_pack_fwd_kernel(const float* __restrict__ w, __bf16* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= TC_NFRAG * 64 * 8) return;
  const int j = i & 7, l = (i >> 3) & 63, f = i >> 9;
  const int kd = f / 3, khp = f % 3, n = l & 15, kb = l >> 4;
  const int kh = 2 * khp + (kb >> 1), ci = (kb & 1) * 8 + j, kw = n >> 1, co = n & 1;
  float v = 0.f;
  if (kh < TC_K && n < 2 * TC_K) v = w[(((co * TC_CIN + ci) * TC_K + kd) * TC_K + kh) * TC_K + kw];
  out[i] = (__bf16)v;
}

// NTW = 16-voxel tiles per wave; 8 waves = 4 rows x 2 halves of the row, W = 32 * NTW
template <int NTW>
__global__ __launch_bounds__(512) void thin_fwd_kernel(const ThinParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int W = NTW * 32;
  constexpr int plane_bytes = TC_ROWS * W * 32;
  constexpr int PROW = (W + 4) * TC_PP;                             // floats per P row
  unsigned char* xs = smem;                                         // [2][TC_ROWS][W][16] bf16
  float* prow = reinterpret_cast<float*>(smem + 2 * plane_bytes);   // [2][4 rows][W + 4][TC_PP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = wave & 3, hf = wave >> 2;
  int b = blockIdx.x;
  const int seg = b % p.nseg; b /= p.nseg;
  const int strip = b % p.nstrip;
  const int n_img = b / p.nstrip;
  const int h0 = strip * TC_BH;
  const int d0 = seg * p.dseg, d1 = min(p.D, d0 + p.dseg);

  // weights: 15 fragments in registers for the whole kernel
  bf16x8 wf[TC_NFRAG];
#pragma unroll
  for (int f = 0; f < TC_NFRAG; ++f) wf[f] = __builtin_bit_cast(bf16x8, p.wfrag[f * 64 + lane]);

  // staging: 16-byte pieces of the plane, piece = (row, w, half); pieces per thread = TC_ROWS * W * 2 / 512 = NTW
  const uint32_t img_bytes = (uint32_t)p.D * p.H * p.W * p.ldx * 2u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.x)) + (int64_t)n_img * img_bytes, 0, img_bytes,
      0x00020000);
  uint32_t poff[NTW];   // byte offset inside a depth plane, or out of bounds (reads as zero)
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int piece = tid + 512 * i;
    const int half = piece & 1, w = (piece >> 1) % W, row = (piece >> 1) / W;
    const int ih = h0 - 2 + row;
    poff[i] = ((unsigned)ih < (unsigned)p.H) ? ((uint32_t)(ih * p.W + w) * p.ldx + half * 8) * 2u : img_bytes;
  }
  const uint32_t dplane = (uint32_t)p.H * p.W * p.ldx * 2u;
  u32x4 rx[NTW];
  auto fetch = [&](int dp) {
    const bool dok = (unsigned)dp < (unsigned)p.D;
#pragma unroll
    for (int i = 0; i < NTW; ++i)
      rx[i] = __builtin_amdgcn_raw_buffer_load_b128(
          rs, (dok && poff[i] != img_bytes) ? (uint32_t)dp * dplane + poff[i] : img_bytes, 0, 0);
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NTW; ++i) *reinterpret_cast<u32x4*>(xs + buf * plane_bytes + (tid + 512 * i) * 16) = rx[i];
  };

  // A fragment address of this lane inside a staged plane: row r + 2*khp + (kb >> 1), voxel 16 t + m, half kb & 1
  const int m = lane & 15, kb = lane >> 4;
  int aoff[3];
#pragma unroll
  for (int khp = 0; khp < 3; ++khp) {
    const int row = r + 2 * khp + ((khp == 2) ? 0 : (kb >> 1));  // kh = 5 has zero weights: re-read the kh = 4 row
    aoff[khp] = (row * W + hf * (W / 2) + m) * 32 + (kb & 1) * 16;
  }
  // the two zero voxels on either side of both copies of this wave's P row
  if (hf == 0) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float* q = prow + (c * 4 + r) * PROW;
      if (lane < 2 * TC_PP) q[lane] = 0.f;
      if (lane < 2 * TC_PP) q[(W + 2) * TC_PP + lane] = 0.f;
    }
  }
  const float bias_v = p.bias ? p.bias[lane & 1] : 0.f;
  const int oh = h0 + r;

  f32x4 acc[5][NTW];
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  fetch(d0 - 2);
  stage(0);
  __syncthreads();

  // one input plane: S = slot of the output plane this input plane is the kd = 0 tap of (dout = dp + 2)
  int buf = 0;
  auto plane_step = [&](const int dp, auto S) {
    constexpr int S0 = decltype(S)::value;
    if (dp > d1 + 1) return;   // (block-uniform)
    fetch(dp + 1);
    const unsigned char* xb = xs + buf * plane_bytes;
    // the A fragments of tile t+1 are read while the 15 MFMAs of tile t run (one scheduling region per tile keeps
    // the compiler from hoisting every read of the plane to the top: registers)
    bf16x8 a[2][3];
#pragma unroll
    for (int khp = 0; khp < 3; ++khp) a[0][khp] = *reinterpret_cast<const bf16x8*>(xb + aoff[khp]);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      if (t + 1 < NTW) {
#pragma unroll
        for (int khp = 0; khp < 3; ++khp)
          a[(t + 1) & 1][khp] = *reinterpret_cast<const bf16x8*>(xb + aoff[khp] + (t + 1) * 16 * 32);
      }
#pragma unroll
      for (int khp = 0; khp < 3; ++khp) {
#pragma unroll
        for (int kd = 0; kd < 5; ++kd) {
          const int s = (S0 - kd + 5) % 5;  // output plane dp - kd + 2
          acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 1][khp], wf[kd * 3 + khp], acc[s][t], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // output plane dp - 2 (kd = 4) is complete: slot (S0 - 4) mod 5 = (S0 + 1) mod 5
    constexpr int SD = (S0 + 1) % 5;
    const int dout = dp - 2;
    const bool live = (dout >= d0) & (dout < d1) & (oh < p.H);
    float* pw_ = prow + ((buf * 4) + r) * PROW;   // the P copy alternates with the plane buffer
    if (live && m < 2 * TC_K) {
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) pw_[(2 + hf * (W / 2) + t * 16 + kb * 4 + i) * TC_PP + m] = acc[SD][t][i];
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[SD][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    stage(buf ^ 1);
    __syncthreads();
    if (live) {  // shift-add over kw: this wave's half of the row, 32 voxels x 2 channels per pass
      float* yrow = p.y + ((((int64_t)n_img * p.D + dout) * p.H + oh) * p.W) * p.ldy;
      const int co = lane & 1;
#pragma unroll
      for (int it = 0; it < (NTW + 1) / 2; ++it) {
        const int wl = it * 32 + (lane >> 1);          // voxel inside this wave's half row (16 * NTW voxels)
        const int w = hf * (W / 2) + wl;
        if (wl < W / 2) {
          float s = bias_v;
#pragma unroll
          for (int kw = 0; kw < TC_K; ++kw) s += pw_[(w + kw) * TC_PP + kw * 2 + co];
          yrow[(int64_t)w * p.ldy + co] = s;
        }
      }
    }
    buf ^= 1;
  };

  // input planes d0-2 .. d1+1; phase k handles slot k (one loop exit: the accumulators keep their registers)
  for (int dp = d0 - 2; dp <= d1 + 1; dp += 5) {
    plane_step(dp, std::integral_constant<int, 0>{});
    plane_step(dp + 1, std::integral_constant<int, 1>{});
    plane_step(dp + 2, std::integral_constant<int, 2>{});
    plane_step(dp + 3, std::integral_constant<int, 3>{});
    plane_step(dp + 4, std::integral_constant<int, 4>{});
  }
}

size_t thin_fwd_smem(int W) { return (size_t)2 * TC_ROWS * W * 32 + (size_t)2 * TC_BH * (W + 4) * TC_PP * 4; }

bool thin_shape_ok(const rehr_direct_conv_desc& d) {
  return d.Cin == TC_CIN && d.Cout == 2 && d.KD == TC_K && d.KH == TC_K && d.KW == TC_K && d.sd == 1 && d.sh == 1 &&
         d.sw == 1 && d.pd == 2 && d.ph == 2 && d.pw == 2 && d.Do == d.Di && d.Ho == d.Hi && d.Wo == d.Wi &&
         d.Wi % 32 == 0 && d.Wi >= 32 && d.Wi <= 160 && d.ldx % 8 == 0 && d.ldx >= TC_CIN && d.ldy >= 2 &&
         (int64_t)d.Di * d.Hi * d.Wi * d.ldx * 2 < ((int64_t)1 << 32);
}


// ------------------------------------------------------------------------------------------------------------------
// Input gradient: dx[v][ci] = sum_{kd,kh,kw,co} dy[v + (kd-2, kh-2, kw-2)][co] * w[co][ci][4-kd][4-kh][4-kw]
// GEMM with M = ci (weights, 13 fragments in registers), N = 16 voxels along w, K = (two (kd,kh) taps) x (8 kw slots,
// 5 real) x (2 co) = 32 per MFMA: the B fragment of a lane is 4 consecutive voxels x 2 channels of the bf16 dY row
// held in LDS as [row][w + pad][2] -- four consecutive dwords at a voxel (4-byte) granular offset.  A block owns
// 4 output rows x W of a depth segment (wave = row x half row) and keeps a ring of six dY planes (8 rows each) in LDS.
// The accumulator quad of a lane is 4 consecutive ci of one voxel: 8-byte bf16 stores, 512 contiguous bytes per wave.
constexpr int TD_NFRAG = 13;           // 25 (kd,kh) taps in pairs
constexpr int TD_RING = 6;
constexpr int TD_PADW = 8;             // dY row: 2 zero voxels in front, 6 behind

struct ThinDgradParams {
  const float* dy;     // fp32 [N][D][H][W][ldy]
  int ldy, N, D, H, W;
  const u32x4* wfrag;  // [TD_NFRAG][64]
  void* dx;            // bf16 [N][D][H][W][lddx]
  int lddx;
  int dseg, nseg, nstrip;
};

// fragment j, lane (m = ci = l & 15, kb = l >> 4): A[m][k = kb*8 + e], tap = 2j + (kb >> 1) = kd*5 + kh (flipped),
// kw = (kb & 1)*4 + (e >> 1), co = e & 1
__global__ void thin_pack_dgrad_kernel(const float* __restrict__ w, __bf16* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= TD_NFRAG * 64 * 8) return;
  const int e = i & 7, l = (i >> 3) & 63, j = i >> 9;
  const int ci = l & 15, kb = l >> 4;
  const int tap = 2 * j + (kb >> 1), kw = (kb & 1) * 4 + (e >> 1), co = e & 1;
  float v = 0.f;
  if (tap < TC_K * TC_K && kw < TC_K) {
    const int kd = tap / TC_K, kh = tap % TC_K;
    v = w[(((co * TC_CIN + ci) * TC_K + (4 - kd)) * TC_K + (4 - kh)) * TC_K + (4 - kw)];
  }
  out[i] = (__bf16)v;
}

template <int NTW>
__global__ __launch_bounds__(512) void thin_dgrad_kernel(const ThinDgradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int W = NTW * 32;
  constexpr int ROWB = (W + TD_PADW) * 4;          // bytes per dY row
  constexpr int PLANEB = TC_ROWS * ROWB;
  constexpr int NPIECE = (TC_ROWS * W + 511) / 512;  // voxels staged per thread and plane
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = wave & 3, hf = wave >> 2;
  int b = blockIdx.x;
  const int seg = b % p.nseg; b /= p.nseg;
  const int strip = b % p.nstrip;
  const int n_img = b / p.nstrip;
  const int h0 = strip * TC_BH;
  const int d0 = seg * p.dseg, d1 = min(p.D, d0 + p.dseg);

  bf16x8 wf[TD_NFRAG];
#pragma unroll
  for (int f = 0; f < TD_NFRAG; ++f) wf[f] = __builtin_bit_cast(bf16x8, p.wfrag[f * 64 + lane]);

  // zero the row pads of every ring slot once (staging only ever writes voxels 0 .. W-1)
  for (int i = tid; i < TD_RING * TC_ROWS * TD_PADW; i += 512) {
    const int pv = i % TD_PADW, row = i / TD_PADW;
    const int vox = pv < 2 ? pv : W + pv;
    *reinterpret_cast<uint32_t*>(smem + row * ROWB + vox * 4) = 0u;
  }

  const float* dyn = p.dy + (int64_t)n_img * p.D * p.H * p.W * p.ldy;
  int pvox[NPIECE];   // voxel offset inside a depth plane or -1
  int plds[NPIECE];
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) {
    const int piece = tid + 512 * i;
    const int w = piece % W, row = piece / W;
    const int ih = h0 - 2 + row;
    const bool ok = (piece < TC_ROWS * W) & ((unsigned)ih < (unsigned)p.H);
    pvox[i] = ok ? ih * p.W + w : -1;
    plds[i] = (piece < TC_ROWS * W) ? row * ROWB + (w + 2) * 4 : -1;
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
    for (int i = 0; i < NPIECE; ++i) {
      if (plds[i] >= 0) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 v = {(__bf16)rx[i].x, (__bf16)rx[i].y};
        *reinterpret_cast<bf16x2*>(smem + slot * PLANEB + plds[i]) = v;
      }
    }
  };

  // B fragment address of this lane: dY row (r + kh), stored voxel hf*W/2 + 16 t + n + (kb & 1) * 4
  const int n = lane & 15, kb = lane >> 4, hi = kb >> 1;
  const int lbase = (hf * (W / 2) + n + (kb & 1) * 4) * 4;

  // planes d0-2 .. d0+2 into slots 0 .. 4
  for (int s = 0; s < 5; ++s) {
    fetch(d0 - 2 + s);
    stage(s);
  }
  __syncthreads();

  __bf16* dxn = reinterpret_cast<__bf16*>(p.dx) + (int64_t)n_img * p.D * p.H * p.W * p.lddx;
  const int oh = h0 + r;

  // output plane d, P = (d - d0) mod 6: dY plane d + kd - 2 sits in slot (P + kd) mod 6
  auto plane_step = [&](const int d, auto PH) {
    constexpr int P = decltype(PH)::value;
    if (d >= d1) return;   // (block-uniform)
    fetch(d + 3);
    int off[TD_NFRAG];
#pragma unroll
    for (int j = 0; j < TD_NFRAG; ++j) {
      const int t0 = 2 * j, t1 = (2 * j + 1 < TC_K * TC_K) ? 2 * j + 1 : 2 * j;   // the 26th tap has zero weights
      const int o0 = ((P + t0 / TC_K) % TD_RING) * PLANEB + (t0 % TC_K) * ROWB;
      const int o1 = ((P + t1 / TC_K) % TD_RING) * PLANEB + (t1 % TC_K) * ROWB;
      off[j] = (hi ? o1 : o0) + r * ROWB + lbase;
    }
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < TD_NFRAG; ++j) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(smem + off[j] + t * 64);
        const u32x4 v = {q[0], q[1], q[2], q[3]};
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], __builtin_bit_cast(bf16x8, v), acc[t], 0, 0, 0);
      }
    }
    if (oh < p.H) {
      __bf16* row = dxn + (((int64_t)d * p.H + oh) * p.W + hf * (W / 2) + n) * p.lddx + 4 * kb;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        const bf16x4 o = {(__bf16)acc[t][0], (__bf16)acc[t][1], (__bf16)acc[t][2], (__bf16)acc[t][3]};
        *reinterpret_cast<bf16x4*>(row + (int64_t)t * 16 * p.lddx) = o;
      }
    }
    stage((P + 5) % TD_RING);   // plane d + 3 takes the slot plane d - 3 left
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

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient: dW[co][ci][kd][kh][kw] = sum_v dy[v][co] * x[v + (kd-2, kh-2, kw-2)][ci]
// The reduction index of the MFMA is the voxel (32 along w per instruction).  The kw shift is put on the x operand --
// an x voxel is 32 bytes, so a shifted row is still aligned for the transposing LDS read -- and the kh shift selects
// one of five dY rows, which makes N = (kh, co) = 10 -> 16:
//   acc[(kd, kw)][ci][(kh, co)] += sum_w x[dx][hx][w + kw - 2][ci] * dy[dx - kd + 2][hx - kh + 2][w][co]
// A block owns 4 x rows of a depth segment and marches over the x planes (wave = 32-voxel chunk of the row, all 25
// accumulator tiles = 100 registers); per x row a wave reads 5 shifted A fragments and, for each of the five dY planes
// in the ring, one B fragment: 25 MFMAs per 10 fragment reads.  Partial sums go to one slab per block, summed in a
// fixed order by thin_wgrad_reduce_kernel (bitwise reproducible), which also writes the torch weight layout.
constexpr int TW_ROWS = TC_BH + 4;     // dY rows per plane
constexpr int TW_SLAB = 25 * 16 * 16;  // floats per block: [kd*5 + kw][ci][n = kh*2 + co]
constexpr int TW_SLABF = TW_SLAB + 16; // + the block's column sums of dY (bias gradient) in the first two extra slots

struct ThinWgradParams {
  const void* x;       // bf16 [N][D][H][W][ldx]
  const float* dy;     // fp32 [N][D][H][W][ldy]
  int ldx, ldy, N, D, H, W;
  float* slabs;        // [blocks][TW_SLAB]
  int dseg, nseg, nstrip;
};

template <int NW>
__global__ __launch_bounds__(64 * NW) void thin_wgrad_kernel(const ThinWgradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int W = NW * 32, NTHR = 64 * NW;
  constexpr int XROWB = (W + 4) * 32;                 // x row: 2 zero voxels either side
  constexpr int XBUF = TC_BH * XROWB;
  constexpr int DROWB = W * 2 + 16;                   // one (row, co) line of dY, bf16
  constexpr int DPLANE = TW_ROWS * 2 * DROWB;
  unsigned char* xs = smem;                           // [2][4 rows][W + 4][16]
  unsigned char* ds = smem + 2 * XBUF;                // [6][8 rows][2 co][W (+8)]
  const int tid = threadIdx.x, lane = tid & 63;
  const int c = __builtin_amdgcn_readfirstlane(tid >> 6);   // chunk of the row
  int b = blockIdx.x;
  const int seg = b % p.nseg; b /= p.nseg;
  const int strip = b % p.nstrip;
  const int n_img = b / p.nstrip;
  const int h0 = strip * TC_BH;
  const int d0 = seg * p.dseg, d1 = min(p.D, d0 + p.dseg);

  // zero pads of the x rows (both buffers), once
  for (int i = tid; i < 2 * TC_BH * 4 * 2; i += NTHR) {
    const int half = i & 1, pv = (i >> 1) & 3, row = i >> 3;
    const int vox = pv < 2 ? pv : W + pv;
    *reinterpret_cast<u32x4*>(xs + row * XROWB + vox * 32 + half * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  // x staging: 4 rows x W voxels x 2 halves = 4 pieces of 16 bytes per thread
  const uint32_t img_bytes = (uint32_t)p.D * p.H * p.W * p.ldx * 2u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.x)) + (int64_t)n_img * img_bytes, 0, img_bytes,
      0x00020000);
  const uint32_t dplane = (uint32_t)p.H * p.W * p.ldx * 2u;
  uint32_t xoff[4];
  int xlds[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = tid + NTHR * i;
    const int half = piece & 1, w = (piece >> 1) % W, row = (piece >> 1) / W;
    const int ih = h0 + row;
    xoff[i] = (ih < p.H) ? ((uint32_t)(ih * p.W + w) * p.ldx + half * 8) * 2u : img_bytes;
    xlds[i] = row * XROWB + (w + 2) * 32 + half * 16;
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
  // dY staging: 8 rows x W voxels, 2 voxels per thread and pass -> 2 passes
  const float* dyn = p.dy + (int64_t)n_img * p.D * p.H * p.W * p.ldy;
  int yvox[2], ylds[2];
  bool yown[2];   // rows h0 .. h0+3 are this block's share of the bias gradient
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pair = tid + NTHR * i;                 // (row, w pair)
    const int w = (pair % (W / 2)) * 2, row = pair / (W / 2);
    const int ih = h0 - 2 + row;
    yvox[i] = ((unsigned)ih < (unsigned)p.H) ? ih * p.W + w : -1;
    ylds[i] = row * 2 * DROWB + w * 2;
    yown[i] = (row >= 2) & (row < 2 + TC_BH);
  }
  float ry[2][4];
  float db0 = 0.f, db1 = 0.f;
  auto fetch_y = [&](int dp) {
    const bool dok = (unsigned)dp < (unsigned)p.D;
    const bool down = (dp >= d0) & (dp < d1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ry[i][0] = ry[i][1] = ry[i][2] = ry[i][3] = 0.f;
      if (dok && yvox[i] >= 0) {
        const float* q = dyn + ((int64_t)dp * p.H * p.W + yvox[i]) * p.ldy;
        ry[i][0] = q[0]; ry[i][1] = q[1]; ry[i][2] = q[p.ldy]; ry[i][3] = q[p.ldy + 1];
        if (down && yown[i]) { db0 += ry[i][0] + ry[i][2]; db1 += ry[i][1] + ry[i][3]; }
      }
    }
  };
  auto stage_y = [&](int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      const bf16x2 c0 = {(__bf16)ry[i][0], (__bf16)ry[i][2]}, c1 = {(__bf16)ry[i][1], (__bf16)ry[i][3]};
      *reinterpret_cast<bf16x2*>(ds + slot * DPLANE + ylds[i]) = c0;
      *reinterpret_cast<bf16x2*>(ds + slot * DPLANE + ylds[i] + DROWB) = c1;
    }
  };

  // A (x, transposing read): 16-lane group g covers voxels 8g .. 8g+7 of the chunk; lane 4q + pp of a group
  // supplies voxel 8g + q (+4 for the second read), channels 4pp .. 4pp+3
  const int grp = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int abase = (32 * c + 8 * grp + qq) * 32 + pp * 8;   // + row * XROWB + kw * 32 (the +2 pad and the -2 shift cancel)
  // B (dY): lane (n = kh*2 + co, kb): row (r + 4 - kh), line co, voxels 32c + 8kb .. +7
  const int nn = lane & 15, kb = lane >> 4;
  const int khl = (nn >> 1) < TC_K ? (nn >> 1) : TC_K - 1, col = nn & 1;
  const int bbase = ((4 - khl) * 2 + col) * DROWB + (32 * c + 8 * kb) * 2;   // + r * 2 * DROWB

  f32x4 acc[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // dY planes d0-2 .. d0+2 -> slots 0 .. 4, x plane d0 -> buffer 0
  for (int s = 0; s < 5; ++s) {
    fetch_y(d0 - 2 + s);
    stage_y(s);
  }
  fetch_x(d0);
  stage_x(0);
  __syncthreads();

  auto frag_a = [&](const unsigned char* base) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * 32));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  // x plane dx, P = (dx - d0) mod 6: dY plane dx - kd + 2 sits in slot (P + 4 - kd) mod 6
  int buf = 0;
  auto plane_step = [&](const int dx, auto PH) {
    constexpr int P = decltype(PH)::value;
    if (dx >= d1) return;   // (block-uniform)
    fetch_x(dx + 1);
    fetch_y(dx + 3);
    const unsigned char* xb = xs + buf * XBUF;
#pragma unroll
    for (int r = 0; r < TC_BH; ++r) {
      bf16x8 a[TC_K];
#pragma unroll
      for (int kw = 0; kw < TC_K; ++kw) a[kw] = frag_a(xb + r * XROWB + kw * 32 + abase);
#pragma unroll
      for (int kd = 0; kd < TC_K; ++kd) {
        const int slot = (P + 4 - kd) % TD_RING;
        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(ds + slot * DPLANE + r * 2 * DROWB + bbase);
#pragma unroll
        for (int kw = 0; kw < TC_K; ++kw)
          acc[kd * 5 + kw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kw], bfr, acc[kd * 5 + kw], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);   // one x row per scheduling region: bounds the fragment registers in flight
    }
    stage_x(buf ^ 1);
    stage_y((P + 5) % TD_RING);
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

  // block sum over the waves (LDS), then the slab; accumulator register i of lane (n = l & 15, q = l >> 4) is
  // C[m = ci = 4q + i][n]
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);       // [NW][TW_SLAB] would not fit: reduce tile by tile
  float* slab = p.slabs + (int64_t)blockIdx.x * TW_SLABF;
  {  // bias gradient: wave sums, then the block
    db0 = wave_sum(db0);
    db1 = wave_sum(db1);
    if (lane == 0) { red[c * 2] = db0; red[c * 2 + 1] = db1; }
    __syncthreads();
    if (tid < 2) {
      float sdb = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) sdb += red[w2 * 2 + tid];
      slab[TW_SLAB + tid] = sdb;
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 25; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) red[(c * 16 + 4 * kb + i) * 16 + nn] = acc[t][i];
    __syncthreads();
    for (int e = tid; e < 256; e += NTHR) {
      float s = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) s += red[w2 * 256 + e];
      slab[t * 256 + e] = s;
    }
    __syncthreads();
  }
}

// dw (2,16,5,5,5) = sum over block slabs, fixed order; slab index [(kd*5 + kw)][ci][kh*2 + co]
// 8 threads share an output element (thread (e, g) sums the blocks g, g + 8, ...; fixed-order combine through LDS): one
// thread per element walking every block's slab was a 0.29 ms latency chain at 160^3
__global__ __launch_bounds__(256) void thin_wgrad_reduce_kernel(const float* __restrict__ slabs, int nblocks, float* __restrict__ dw,
                                                             float* __restrict__ dbias) {
  constexpr int G = 8, EPB = 256 / G, NDW = 2 * TC_CIN * 125;
  __shared__ float part[256];
  const int el = threadIdx.x % EPB, g = threadIdx.x / EPB;
  const int i = blockIdx.x * EPB + el;   // index into dw (2,16,5,5,5), then the two bias gradients
  int si = -1;
  if (i < NDW) {
    const int kw = i % 5, kh = (i / 5) % 5, kd = (i / 25) % 5, ci = (i / 125) % TC_CIN, co = i / (125 * TC_CIN);
    si = ((kd * 5 + kw) * 16 + ci) * 16 + kh * 2 + co;
  } else if (i < NDW + 2) {
    si = TW_SLAB + (i - NDW);
  }
  float s0 = 0.f, s1 = 0.f;
  if (si >= 0) {
    int bq = g;
    for (; bq + G < nblocks; bq += 2 * G) {
      s0 += slabs[(int64_t)bq * TW_SLABF + si];
      s1 += slabs[(int64_t)(bq + G) * TW_SLABF + si];
    }
    for (; bq < nblocks; bq += G) s0 += slabs[(int64_t)bq * TW_SLABF + si];
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

}  // namespace

static void thin_segments(const rehr_direct_conv_desc& d, int& nstrip, int& dseg, int& nseg) {
  nstrip = (d.Hi + TC_BH - 1) / TC_BH;
  // depth segments: enough blocks for the chip, each long enough to amortise the halo planes
  int ns = 1;
  while ((int64_t)d.N * nstrip * ns < 512 && d.Di / (ns * 2) >= 16) ns *= 2;
  dseg = (d.Di + ns - 1) / ns;
  nseg = (d.Di + dseg - 1) / dseg;
}

extern "C" int64_t rehr_conv5_thin_workspace_bytes(const rehr_direct_conv_desc* dp) {
  if (dp == nullptr) return REHR_EINVAL;
  if (!thin_shape_ok(*dp)) return REHR_ENOSUP;
  int nstrip, dseg, nseg;
  thin_segments(*dp, nstrip, dseg, nseg);
  const int64_t slabs = (int64_t)dp->N * nstrip * nseg * TW_SLABF * 4;
  const int64_t pack = (int64_t)TC_NFRAG * 64 * 16;
  return slabs > pack ? slabs : pack;
}

extern "C" int rehr_conv5_thin_supported(const rehr_direct_conv_desc* d) { return d != nullptr && thin_shape_ok(*d) ? 1 : 0; }

#define TC_SWITCH(KERNEL, THREADS_OF)                                                                               \
  switch (d.Wi / 32) {                                                                                              \
    case 1: TC_LAUNCH(KERNEL, 1, THREADS_OF(1)); break;                                                             \
    case 2: TC_LAUNCH(KERNEL, 2, THREADS_OF(2)); break;                                                             \
    case 3: TC_LAUNCH(KERNEL, 3, THREADS_OF(3)); break;                                                             \
    case 4: TC_LAUNCH(KERNEL, 4, THREADS_OF(4)); break;                                                             \
    case 5: TC_LAUNCH(KERNEL, 5, THREADS_OF(5)); break;                                                             \
    default: return REHR_ENOSUP;                                                                                    \
  }
#define TC_LAUNCH(KERNEL, NT_, THREADS)                                                                             \
  do {                                                                                                              \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL<NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)smem);                                                                           \
    hipLaunchKernelGGL(KERNEL<NT_>, dim3((unsigned)blocks), dim3(THREADS), smem, st, p);                            \
  } while (0)
#define TC_T512(n) 512
#define TC_T64N(n) (64 * (n))

extern "C" int rehr_conv5_thin_fwd_bf16(const rehr_direct_conv_desc* dp, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
  if (dp == nullptr || dp->x == nullptr || dp->w == nullptr || dp->y == nullptr || workspace == nullptr) return REHR_EINVAL;
  const rehr_direct_conv_desc& d = *dp;
  if (!thin_shape_ok(d)) return REHR_ENOSUP;
  if (workspace_bytes < rehr_conv5_thin_workspace_bytes(dp) || d.act != REHR_ACT_NONE || d.stats_mode != 0) return REHR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(thin_pack_fwd_kernel, dim3((TC_NFRAG * 512 + 255) / 256), dim3(256), 0, st, d.w,
                     reinterpret_cast<__bf16*>(workspace));
  ThinParams p;
  p.x = d.x; p.ldx = d.ldx; p.N = d.N; p.D = d.Di; p.H = d.Hi; p.W = d.Wi;
  p.wfrag = reinterpret_cast<const u32x4*>(workspace);
  p.bias = d.bias; p.y = d.y; p.ldy = d.ldy;
  thin_segments(d, p.nstrip, p.dseg, p.nseg);
  const int64_t blocks = (int64_t)d.N * p.nstrip * p.nseg;
  const size_t smem = thin_fwd_smem(d.Wi);
  TC_SWITCH(thin_fwd_kernel, TC_T512)
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_conv5_thin_dgrad_bf16(const rehr_direct_conv_desc* dp, void* dx, int32_t lddx, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
  if (dp == nullptr || dp->w == nullptr || dp->y == nullptr || dx == nullptr || workspace == nullptr) return REHR_EINVAL;
  const rehr_direct_conv_desc& d = *dp;
  if (!thin_shape_ok(d)) return REHR_ENOSUP;
  if (workspace_bytes < rehr_conv5_thin_workspace_bytes(dp) || lddx < TC_CIN || lddx % 4) return REHR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(thin_pack_dgrad_kernel, dim3((TD_NFRAG * 512 + 255) / 256), dim3(256), 0, st, d.w,
                     reinterpret_cast<__bf16*>(workspace));
  ThinDgradParams p;
  p.dy = d.y; p.ldy = d.ldy; p.N = d.N; p.D = d.Di; p.H = d.Hi; p.W = d.Wi;
  p.wfrag = reinterpret_cast<const u32x4*>(workspace);
  p.dx = dx; p.lddx = lddx;
  thin_segments(d, p.nstrip, p.dseg, p.nseg);
  const int64_t blocks = (int64_t)d.N * p.nstrip * p.nseg;
  const size_t smem = (size_t)TD_RING * TC_ROWS * (d.Wi + TD_PADW) * 4;
  TC_SWITCH(thin_dgrad_kernel, TC_T512)
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_conv5_thin_wgrad_bf16(const rehr_direct_conv_desc* dp, float* dw, float* dbias, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
  if (dp == nullptr || dp->x == nullptr || dp->y == nullptr || dw == nullptr || workspace == nullptr) return REHR_EINVAL;
  const rehr_direct_conv_desc& d = *dp;
  if (!thin_shape_ok(d)) return REHR_ENOSUP;
  if (workspace_bytes < rehr_conv5_thin_workspace_bytes(dp)) return REHR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  ThinWgradParams p;
  p.x = d.x; p.dy = d.y; p.ldx = d.ldx; p.ldy = d.ldy; p.N = d.N; p.D = d.Di; p.H = d.Hi; p.W = d.Wi;
  p.slabs = reinterpret_cast<float*>(workspace);
  thin_segments(d, p.nstrip, p.dseg, p.nseg);
  const int64_t blocks = (int64_t)d.N * p.nstrip * p.nseg;
  const size_t smem = (size_t)2 * TC_BH * (d.Wi + 4) * 32 + (size_t)TD_RING * TW_ROWS * 2 * (d.Wi * 2 + 16);
  TC_SWITCH(thin_wgrad_kernel, TC_T64N)
  REHR_LAUNCH_CHECK();
  hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((2 * TC_CIN * 125 + 2 + 31) / 32), dim3(256), 0, st, p.slabs,
                     (int)blocks, dw, dbias);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
