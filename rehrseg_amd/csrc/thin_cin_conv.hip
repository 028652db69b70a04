// Thin-INPUT convolutions (C_in = 1 or 2: the network input) on the fp32 matrix cores: the first nnU-Net conv
// (Conv3d(1, 32, 3x3x3 or 1x3x3), dynamic_network_architectures' first StackedConvBlocks, built at train_all.py:474-493)
// and the FLAVR stem Conv3d(img_channels, 64, (3,7,7), stride (1,2,2)) (models/FLAVR/resnet_3D.py:47-48), incl. the
// per-slice (1,7,7) responses of the teacher's overlapping windows (train_all.py:85-112).  The vector-pipe kernel of
// direct_conv.hip runs them at 14-26 TFLOP/s (one x load + 4 LDS reads per 16 FMAs: issue bound): 0.64 ms of a cfg-3
// step, 2.0 ms of cfg-5.
//
// GEMM per 16 output voxels along w:  C[m = c_out][n = voxel] += A[m][k] * B[k][n] on v_mfma_f32_16x16x4_f32 with
// k = (row r = (ci, kd, kh), group g of 4 kw taps, kq = kw - 4g): the kq index of the instruction is the kw tap, so a
// lane's B value is the input patch at (row, x = voxel * sw + 4g + kq) -- one LDS dword at (lane base + immediate),
// no index arithmetic in the loop; kw is padded to a multiple of 4 with zero weights (3 -> 4, 7 -> 8).
// A block stages the input patch of its output tile (4 rows x 64 or 32 columns, wave = row) and ALL weights in LDS
// ([k step][kq][c_out], loaded once per block; blocks walk several tiles), every A dword feeds NV voxel tiles and every
// B dword all c_out tiles.  M = c_out makes the accumulator quad of a lane 4 consecutive channels of one voxel: 16-byte
// NDHWC stores.  Epilogue: bias, activation, fp64 statistics of the stored values (one atomic per channel and block).
#include "common.h"

namespace {

struct ThinCinParams {
  rehr_direct_conv_desc d;
  int G;            // kw groups of 4
  int ksteps;       // Cin * KD * KH * G
  int PH, PW;       // patch rows / padded columns (floats)
  int tiles_h, tiles_w, tiles_per_img, tiles_per_block;
};

template <int NTN, int NV, typename TO>   // c_out tiles (Cout = 16 NTN), voxel tiles per wave, stored dtype (float / __bf16)
__global__ __launch_bounds__(256) void thin_cin_fwd_kernel(const ThinCinParams p) {
  const rehr_direct_conv_desc& d = p.d;
  constexpr int COUT = 16 * NTN, COLS = 16 * NV, ROWS = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;                                   // [ksteps][4][COUT]
  float* Ps = smem + (size_t)p.ksteps * 4 * COUT;     // [Cin*KD][PH][PW]
  float* red = Ps;                                    // statistics scratch (the patch is idle then)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_img = blockIdx.y;

  // weights -> LDS, zero for the padded kw taps
  for (int i = tid; i < p.ksteps * 4 * COUT; i += 256) {
    const int co = i % COUT, kq = (i / COUT) & 3, ks = i / (4 * COUT);
    const int g = ks % p.G, r = ks / p.G;
    const int kh = r % d.KH, kd = (r / d.KH) % d.KD, ci = r / (d.KH * d.KD);
    const int kw = 4 * g + kq;
    Wl[i] = kw < d.KW ? d.w[(((int64_t)(co * d.Cin + ci) * d.KD + kd) * d.KH + kh) * d.KW + kw] : 0.f;
  }

  const int nn = lane & 15, kq = lane >> 4;
  float bv[NTN][4];
#pragma unroll
  for (int mt = 0; mt < NTN; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[mt][i] = d.bias ? d.bias[mt * 16 + 4 * kq + i] : 0.f;
  float s1[NTN][4], s2[NTN][4];
#pragma unroll
  for (int mt = 0; mt < NTN; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) { s1[mt][i] = 0.f; s2[mt][i] = 0.f; }

  const float* xn = d.x + (int64_t)n_img * d.Di * d.Hi * d.Wi * d.ldx;
  TO* yn = reinterpret_cast<TO*>(d.y) + (int64_t)n_img * d.Do * d.Ho * d.Wo * d.ldy;
  const int planes = d.Cin * d.KD;
  const int patch = planes * p.PH * p.PW;
  const int t_begin = blockIdx.x * p.tiles_per_block;
  const int t_end = min(p.tiles_per_img, t_begin + p.tiles_per_block);

  for (int tile = t_begin; tile < t_end; ++tile) {
    const int tw = tile % p.tiles_w;
    const int th = (tile / p.tiles_w) % p.tiles_h;
    const int od = tile / (p.tiles_w * p.tiles_h);
    const int oh0 = th * ROWS, ow0 = tw * COLS;
    const int id0 = od * d.sd - d.pd, ih0 = oh0 * d.sh - d.ph, iw0 = ow0 * d.sw - d.pw;
    __syncthreads();   // the previous tile's patch reads (and the weight stores of the first round) are done
    for (int i = tid; i < patch; i += 256) {
      const int px = i % p.PW, py = (i / p.PW) % p.PH, pl = i / (p.PW * p.PH);
      const int kd = pl % d.KD, ci = pl / d.KD;
      const int id = id0 + kd, ih = ih0 + py, iw = iw0 + px;
      const bool ok = ((unsigned)id < (unsigned)d.Di) & ((unsigned)ih < (unsigned)d.Hi) & ((unsigned)iw < (unsigned)d.Wi);
      Ps[i] = ok ? xn[(((int64_t)id * d.Hi + ih) * d.Wi + iw) * d.ldx + ci] : 0.f;
    }
    __syncthreads();

    f32x4 acc[NV][NTN];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int mt = 0; mt < NTN; ++mt) acc[v][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane base inside the patch: output row `wave`, voxel nn of tile 0, tap kq
    const float* pb = Ps + (wave * d.sh) * p.PW + nn * d.sw + kq;
    const float* wb = Wl + kq * COUT + nn;
    int ks = 0;
    for (int pl = 0; pl < planes; ++pl) {
      for (int kh = 0; kh < d.KH; ++kh) {
        const float* prow = pb + (pl * p.PH + kh) * p.PW;
        for (int g = 0; g < p.G; ++g, ++ks) {
          float a[NTN], b[NV];
#pragma unroll
          for (int mt = 0; mt < NTN; ++mt) a[mt] = wb[ks * 4 * COUT + mt * 16];
#pragma unroll
          for (int v = 0; v < NV; ++v) b[v] = prow[v * 16 * d.sw + 4 * g];
#pragma unroll
          for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int mt = 0; mt < NTN; ++mt)
              acc[v][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[v], acc[v][mt], 0, 0, 0);
        }
      }
    }

    const int oh = oh0 + wave;
    if (oh < d.Ho) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int ow = ow0 + v * 16 + nn;
        if (ow < d.Wo) {
          TO* yo = yn + (((int64_t)od * d.Ho + oh) * d.Wo + ow) * d.ldy + 4 * kq;
#pragma unroll
          for (int mt = 0; mt < NTN; ++mt) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float r = apply_act(acc[v][mt][i] + bv[mt][i], d.act, d.slope);
              o[i] = r;
              s1[mt][i] += r;
              s2[mt][i] += r * r;
            }
            store4(yo + mt * 16, o);   // (statistics above: of the fp32 values, also when the store rounds to bf16)
          }
        }
      }
    }
  }

  if (d.stats_mode != 0) {   // lanes of one kq hold the same channels: sum over the 16 voxel lanes, then the 4 waves
#pragma unroll
    for (int mt = 0; mt < NTN; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          s1[mt][i] += __shfl_xor(s1[mt][i], o, 64);
          s2[mt][i] += __shfl_xor(s2[mt][i], o, 64);
        }
      }
    __syncthreads();
    if (nn == 0) {
#pragma unroll
      for (int mt = 0; mt < NTN; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = mt * 16 + 4 * kq + i;
          red[(wave * 2 + 0) * COUT + c] = s1[mt][i];
          red[(wave * 2 + 1) * COUT + c] = s2[mt][i];
        }
    }
    __syncthreads();
    for (int c = tid; c < 2 * COUT; c += 256) {
      const int st = c / COUT, ch = c % COUT;
      float s = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) s += red[(w4 * 2 + st) * COUT + ch];
      if (st == 0 || d.stats_mode == 2) atomicAdd(d.stats + ((int64_t)n_img * d.Cout + ch) * 2 + st, (double)s);
    }
  }
}

}  // namespace

// rehr_conv_small_cin_fwd_f32 tries this first; REHR_ENOSUP = not a shape for it
// y_bf16: d.y points at bf16 elements (mixed precision: the layer behind takes bf16 activations)
int thin_cin_fwd_try(const rehr_direct_conv_desc& d, hipStream_t stream, bool y_bf16) {
  if (d.Cin < 1 || d.Cin > 2 || (d.Cout != 32 && d.Cout != 64) || d.KW > 8 || d.sw < 1 || d.sw > 2 || d.ldy % 4 ||
      (reinterpret_cast<uintptr_t>(d.y) & (y_bf16 ? 7 : 15)))
    return REHR_ENOSUP;
  ThinCinParams p;
  p.d = d;
  p.G = (d.KW + 3) / 4;
  p.ksteps = d.Cin * d.KD * d.KH * p.G;
  const int NTN = d.Cout / 16, NV = d.Cout == 32 ? 4 : 2, COLS = 16 * NV;
  p.PH = 3 * d.sh + d.KH;
  p.PW = (COLS - 1) * d.sw + 4 * p.G;
  p.PW += (p.PW & 1) ? 0 : 1;                       // odd row pitch
  const size_t wbytes = (size_t)p.ksteps * 4 * d.Cout * sizeof(float);
  size_t pbytes = (size_t)d.Cin * d.KD * p.PH * p.PW * sizeof(float);
  const size_t rbytes = (size_t)4 * 2 * d.Cout * sizeof(float);
  if (pbytes < rbytes) pbytes = rbytes;
  const size_t smem = wbytes + pbytes;
  if (smem > 150 * 1024) return REHR_ENOSUP;
  p.tiles_h = (d.Ho + 3) / 4;
  p.tiles_w = (d.Wo + COLS - 1) / COLS;
  const int64_t tpi = (int64_t)d.Do * p.tiles_h * p.tiles_w;
  if (tpi >= ((int64_t)1 << 31)) return REHR_ENOSUP;
  p.tiles_per_img = (int)tpi;
  // blocks walk runs of tiles: the weights are loaded once per block, statistics cost one atomic per channel and block
  int64_t bx = tpi;
  const int64_t cap = 1024 / d.N > 0 ? 1024 / d.N : 1;
  if (bx > cap) bx = cap;
  p.tiles_per_block = (int)((tpi + bx - 1) / bx);
  bx = (tpi + p.tiles_per_block - 1) / p.tiles_per_block;
  dim3 grid((unsigned)bx, d.N);
#define TCI_LAUNCH(NTN_, NV_, TO_)                                                                                   \
  do {                                                                                                               \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(thin_cin_fwd_kernel<NTN_, NV_, TO_>),                      \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)                   \
      return REHR_EHIP;                                                                                              \
    hipLaunchKernelGGL((thin_cin_fwd_kernel<NTN_, NV_, TO_>), grid, dim3(256), smem, stream, p);                     \
  } while (0)
  if (NTN == 2 && y_bf16) TCI_LAUNCH(2, 4, __bf16);
  else if (NTN == 2) TCI_LAUNCH(2, 4, float);
  else if (y_bf16) TCI_LAUNCH(4, 2, __bf16);
  else TCI_LAUNCH(4, 2, float);
#undef TCI_LAUNCH
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Weight (+ bias) gradient of the same layers (the last convolution without its own kernel; it ran
// as im2col -> [voxels x 32..160] columns in HBM -> a 1x1x1 weight-gradient GEMM: 0.53 ms of a cfg-3 step at 2 TB/s,
// 0.44 ms of cfg-2).
//
//   dW[co][ci][tap] = sum_{n, voxel} dY[n][voxel][co] * x[n][voxel * s - p + tap][ci]        db[co] = sum dY
//
// GEMM per 4 output voxels along w on v_mfma_f32_16x16x4_f32:  C[m = co][n = tap] += A[m][k = voxel] * B[k][n], the
// reduction index of the instruction is the VOXEL.  A = dY from an LDS tile [voxel][co] (pitch = C_out + 16 floats:
// the two voxel groups of a 32-lane half hit disjoint banks), B = the staged input patch at (voxel * sw + tap offset):
// one LDS dword at (voxel base + per-lane tap offset), the tap offsets of a lane's columns live in NTN registers.  One
// spare column carries the bias gradient: its B value is the constant 1.  grid.z = ci (a block stages the planes of one
// input channel); a block walks a run of (4 rows x 64 voxels) tiles with its accumulators resident, then the 4 waves
// are summed through LDS and the block writes ONE slab; a second kernel adds the slabs in a fixed order (bitwise
// reproducible, no atomics).  HBM traffic = dY once + x once (+ the slabs).
namespace {

struct ThinCinWgParams {
  rehr_direct_conv_desc d;      // d.y = dY
  int T;                        // taps per input channel: KD*KH*KW
  int NTP;                      // padded tap columns = 16 * NTN  (T real + 1 bias column + zero columns)
  int PH, PW;
  int tiles_h, tiles_w, tiles_per_img, tiles_per_block;
  int ZP;                       // dY tile pitch per voxel (floats)
  int want_bias;
  float* slabs;                 // [Cin][gridDim.y * gridDim.x][COUT][NTP]
};

template <int NTM, int NTN, typename TZ>   // c_out tiles (Cout = 16 NTM), tap tiles, dtype of dY (float / __bf16)
__global__ __launch_bounds__(256, (NTM * NTN <= 8) ? 2 : 1) void thin_cin_wgrad_kernel(const ThinCinWgParams p) {
  const rehr_direct_conv_desc& d = p.d;
  constexpr int COUT = 16 * NTM, COLS = 64, ROWS = 4, NTPc = 16 * NTN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zs = smem;                                    // [ROWS * COLS][ZP]
  float* Ps = smem + (size_t)ROWS * COLS * p.ZP;       // [KD][PH][PW]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_img = blockIdx.y, ci = blockIdx.z;
  const int nn = lane & 15, kq = lane >> 4;

  // per-lane tap offsets into the patch for its column of every tap tile
  int toff[NTN];
  bool is_bias[NTN];
#pragma unroll
  for (int t = 0; t < NTN; ++t) {
    const int n = t * 16 + nn;
    int off = 0;
    if (n < p.T) {
      const int kw = n % d.KW, kh = (n / d.KW) % d.KH, kd = n / (d.KW * d.KH);
      off = (kd * p.PH + kh) * p.PW + kw;
    }
    toff[t] = off;
    is_bias[t] = (n == p.T);
  }

  f32x4 acc[NTM][NTN];
#pragma unroll
  for (int m = 0; m < NTM; ++m)
#pragma unroll
    for (int t = 0; t < NTN; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* xn = d.x + (int64_t)n_img * d.Di * d.Hi * d.Wi * d.ldx + ci;
  const TZ* zn = reinterpret_cast<const TZ*>(d.y) + (int64_t)n_img * d.Do * d.Ho * d.Wo * d.ldy;
  const int patch = d.KD * p.PH * p.PW;
  const int t_begin = blockIdx.x * p.tiles_per_block;
  const int t_end = min(p.tiles_per_img, t_begin + p.tiles_per_block);
  constexpr int ZV = COUT / 4;                         // float4 pieces per voxel
  constexpr int NZ = ROWS * COLS * ZV / 256;           // dY pieces per thread (8 or 16)
  constexpr int NPMAX = NTN >= 10 ? 24 : 8;            // patch floats per thread ((3,7,7) stem: 3 x 13 x 135 / 256 = 21; 3x3x3: 5)

  // software pipeline: the next tile's dY and patch are fetched into registers while the current tile is multiplied
  f32x4 zr[NZ];
  float pr[NPMAX];
  auto fetch = [&](int tile) {
    const int tw = tile % p.tiles_w;
    const int th = (tile / p.tiles_w) % p.tiles_h;
    const int od = tile / (p.tiles_w * p.tiles_h);
    const int oh0 = th * ROWS, ow0 = tw * COLS;
    const int id0 = od * d.sd - d.pd, ih0 = oh0 * d.sh - d.ph, iw0 = ow0 * d.sw - d.pw;
#pragma unroll
    for (int k = 0; k < NPMAX; ++k) {
      const int i = tid + 256 * k;
      float v = 0.f;
      if (i < patch) {
        const int px = i % p.PW, py = (i / p.PW) % p.PH, kd = i / (p.PW * p.PH);
        const int id = id0 + kd, ih = ih0 + py, iw = iw0 + px;
        const bool ok = ((unsigned)id < (unsigned)d.Di) & ((unsigned)ih < (unsigned)d.Hi) & ((unsigned)iw < (unsigned)d.Wi);
        if (ok) v = xn[(((int64_t)id * d.Hi + ih) * d.Wi + iw) * d.ldx];
      }
      pr[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const int i = tid + 256 * k;
      const int j = i % ZV, v = i / ZV;
      const int col = v % COLS, row = v / COLS;
      const int oh = oh0 + row, ow = ow0 + col;
      f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
      if (oh < d.Ho && ow < d.Wo) z = load4(zn + (((int64_t)od * d.Ho + oh) * d.Wo + ow) * d.ldy + 4 * j);
      zr[k] = z;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int k = 0; k < NPMAX; ++k) {
      const int i = tid + 256 * k;
      if (i < patch) Ps[i] = pr[k];
    }
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const int i = tid + 256 * k;
      const int j = i % ZV, v = i / ZV;
      *reinterpret_cast<f32x4*>(Zs + (size_t)v * p.ZP + 4 * j) = zr[k];
    }
  };

  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();   // the previous tile's LDS reads are done
    stage();
    __syncthreads();
    if (tile + 1 < t_end) fetch(tile + 1);   // in flight under the MFMAs below

    const float* zb = Zs + (size_t)(wave * COLS + kq) * p.ZP + nn;
    const float* pb = Ps + (wave * d.sh) * p.PW + kq * d.sw;
    constexpr int GUNR = NTN >= 10 ? 1 : 4;
#pragma unroll GUNR
    for (int g = 0; g < COLS / 4; ++g) {
      float a[NTM], b[NTN];
#pragma unroll
      for (int m = 0; m < NTM; ++m) a[m] = zb[(size_t)(4 * g) * p.ZP + m * 16];
#pragma unroll
      for (int t = 0; t < NTN; ++t) {
        const float v = pb[4 * g * d.sw + toff[t]];
        b[t] = is_bias[t] ? 1.f : v;
      }
#pragma unroll
      for (int m = 0; m < NTM; ++m)
#pragma unroll
        for (int t = 0; t < NTN; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[t], acc[m][t], 0, 0, 0);
    }
  }

  // sum the 4 waves through LDS, one c_out tile at a time (16 x NTP floats per wave), and write the block's slab
  const int64_t blk = ((int64_t)ci * gridDim.y + n_img) * gridDim.x + blockIdx.x;
  float* slab = p.slabs + blk * COUT * NTPc;
#pragma unroll
  for (int m = 0; m < NTM; ++m) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NTN; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) smem[(wave * 16 + 4 * kq + i) * NTPc + t * 16 + nn] = acc[m][t][i];
    __syncthreads();
    for (int e = tid; e < 16 * NTPc; e += 256) {
      const float s = (smem[e] + smem[16 * NTPc + e]) + (smem[2 * 16 * NTPc + e] + smem[3 * 16 * NTPc + e]);
      slab[m * 16 * NTPc + e] = s;
    }
  }
}

// dW[co][ci][n] = sum_b slab[ci][b][co][n];  db[co] = sum_b slab[0][b][co][T]
__global__ void thin_cin_wgrad_reduce_kernel(const float* __restrict__ slabs, int nb, int Cin, int Cout, int T, int NTP,
                                             float* __restrict__ dw, float* __restrict__ dbias) {
  const int total = Cout * Cin * (T + 1);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int n = i % (T + 1), ci = (i / (T + 1)) % Cin, co = i / ((T + 1) * Cin);
    if (n == T && (ci != 0 || dbias == nullptr)) continue;
    const float* s = slabs + ((int64_t)ci * nb * Cout + co) * NTP + n;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = 0;
    for (; b + 4 <= nb; b += 4) {
      a0 += s[(int64_t)(b + 0) * Cout * NTP];
      a1 += s[(int64_t)(b + 1) * Cout * NTP];
      a2 += s[(int64_t)(b + 2) * Cout * NTP];
      a3 += s[(int64_t)(b + 3) * Cout * NTP];
    }
    for (; b < nb; ++b) a0 += s[(int64_t)b * Cout * NTP];
    const float r = (a0 + a1) + (a2 + a3);
    if (n == T) dbias[co] = r;
    else dw[((int64_t)co * Cin + ci) * T + n] = r;
  }
}

bool thin_cin_wgrad_plan(const rehr_direct_conv_desc& d, ThinCinWgParams& p, dim3& grid, size_t& smem) {
  if (d.Cin < 1 || d.Cin > 2 || (d.Cout != 32 && d.Cout != 64) || d.sw < 1 || d.sw > 2 || d.ldy % 4 ||
      (reinterpret_cast<uintptr_t>(d.y) & 7))
    return false;
  p.d = d;
  p.T = d.KD * d.KH * d.KW;
  const int ntn = (p.T + 1 + 15) / 16;
  if (ntn != 1 && ntn != 2 && ntn != 10) return false;   // 1x3x3, 3x3x3, (3,7,7): the instantiated shapes
  p.NTP = 16 * ntn;
  p.PH = 3 * d.sh + d.KH;
  p.PW = 63 * d.sw + d.KW;
  p.PW += (p.PW & 1) ? 0 : 1;
  p.ZP = d.Cout + 16;
  smem = ((size_t)4 * 64 * p.ZP + (size_t)d.KD * p.PH * p.PW) * sizeof(float);
  const size_t red = (size_t)4 * 16 * p.NTP * sizeof(float);
  if (smem < red) smem = red;
  if (smem > 150 * 1024) return false;
  if (d.KD * p.PH * p.PW > (ntn >= 10 ? 24 : 8) * 256) return false;   // the register prefetch holds 24 / 8 patch floats per thread
  p.tiles_h = (d.Ho + 3) / 4;
  p.tiles_w = (d.Wo + 63) / 64;
  const int64_t tpi = (int64_t)d.Do * p.tiles_h * p.tiles_w;
  if (tpi >= ((int64_t)1 << 31) || d.N > 65535) return false;
  p.tiles_per_img = (int)tpi;
  int64_t bx = tpi;
  const int64_t cap = 512 / ((int64_t)d.N * d.Cin) > 0 ? 512 / ((int64_t)d.N * d.Cin) : 1;   // ~2 blocks per CU in total
  if (bx > cap) bx = cap;
  p.tiles_per_block = (int)((tpi + bx - 1) / bx);
  bx = (tpi + p.tiles_per_block - 1) / p.tiles_per_block;
  grid = dim3((unsigned)bx, d.N, d.Cin);
  return true;
}

}  // namespace

int64_t thin_cin_wgrad_workspace_bytes(const rehr_direct_conv_desc& d) {
  ThinCinWgParams p;
  dim3 grid;
  size_t smem;
  if (!thin_cin_wgrad_plan(d, p, grid, smem)) return 0;
  return (int64_t)grid.x * grid.y * grid.z * d.Cout * p.NTP * (int64_t)sizeof(float);
}

// rehr_conv_small_cin_wgrad_f32 tries this first; REHR_ENOSUP = not a shape for it
int thin_cin_wgrad_try(const rehr_direct_conv_desc& d, float* dw, float* dbias, float* workspace, int64_t workspace_bytes,
                       hipStream_t stream, bool dy_bf16) {
  ThinCinWgParams p;
  dim3 grid;
  size_t smem;
  if (!thin_cin_wgrad_plan(d, p, grid, smem)) return REHR_ENOSUP;
  if (!dy_bf16 && (reinterpret_cast<uintptr_t>(d.y) & 15)) return REHR_ENOSUP;
  const int64_t need = (int64_t)grid.x * grid.y * grid.z * d.Cout * p.NTP * (int64_t)sizeof(float);
  if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15)) return REHR_EINVAL;
  p.slabs = workspace;
  p.want_bias = dbias != nullptr;
  const int ntm = d.Cout / 16, ntn = p.NTP / 16;
#define TCW_LAUNCH(NTM_, NTN_)                                                                                       \
  do {                                                                                                               \
    if (dy_bf16) {                                                                                                   \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(thin_cin_wgrad_kernel<NTM_, NTN_, __bf16>),              \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)                 \
        return REHR_EHIP;                                                                                            \
      hipLaunchKernelGGL((thin_cin_wgrad_kernel<NTM_, NTN_, __bf16>), grid, dim3(256), smem, stream, p);             \
    } else {                                                                                                         \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(thin_cin_wgrad_kernel<NTM_, NTN_, float>),               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)                 \
        return REHR_EHIP;                                                                                            \
      hipLaunchKernelGGL((thin_cin_wgrad_kernel<NTM_, NTN_, float>), grid, dim3(256), smem, stream, p);              \
    }                                                                                                                \
  } while (0)
  if (ntm == 2 && ntn == 1) TCW_LAUNCH(2, 1);
  else if (ntm == 2 && ntn == 2) TCW_LAUNCH(2, 2);
  else if (ntm == 2 && ntn == 10) TCW_LAUNCH(2, 10);
  else if (ntm == 4 && ntn == 1) TCW_LAUNCH(4, 1);
  else if (ntm == 4 && ntn == 2) TCW_LAUNCH(4, 2);
  else TCW_LAUNCH(4, 10);
#undef TCW_LAUNCH
  const int total = d.Cout * d.Cin * (p.T + 1);
  hipLaunchKernelGGL(thin_cin_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, workspace,
                     (int)(grid.x * grid.y), d.Cin, d.Cout, p.T, p.NTP, dw, dbias);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
