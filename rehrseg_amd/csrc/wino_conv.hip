// Winograd F(2x2, 3x3) over (H, W) x direct taps over depth, on the fp32 matrix cores.
//
// The stride-1 3x3x3 (and 1x3x3) convolutions -- forward and input gradient -- are ~80 % of the
// FLOPs of both networks and the fp32 MFMA pipe is the bound, so the lever left is doing fewer
// multiplications: the 2-D minimal filtering algorithm needs 16 products per 2x2 output tile
// instead of 36 (2.25x fewer MFMA k-steps); depth taps stay a plain sum.  Same descriptor and
// results as gather-GEMM (fp32, differences ~1e-6 from the transform arithmetic).
//
//   block   = 32 Winograd tiles (4 x 8 tiles = 8 x 16 outputs of one depth slice) x 32 channels
//   wave r  = Winograd row r: holds M[r][0..3] (4 accumulator tiles of 32 tiles x 32 channels)
//   K item  = (32-channel chunk, depth tap): the 10 x 18 input patch of that slice is staged in
//             LDS (fetched one item ahead into registers); each lane builds its A fragments
//             V[r][c] = (B^T d B)[r][c] on the fly from 8 LDS reads per k-group (VALU adds hide
//             under the MFMAs); B fragments = transformed weights U straight from L1/L2, one
//             column ahead.
//   output  = Y = A^T M A: column combine in registers, row combine across the 4 waves via LDS.
#include "common.h"
#include "wino_conv.h"

namespace {

constexpr int TH = 4, TW = 8;                     // Winograd tiles per block
constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;   // staged input patch: 10 x 18
constexpr int PVOX = PH * PW;                     // 180
constexpr int LDX = 36;
constexpr int NX = (PVOX * 8 + 255) / 256;        // 16-byte pieces staged per thread (6)

struct WinoParams {
  rehr_gather_gemm_desc d;
  int nb_h, nb_w;       // 8 x 16 output regions per depth slice
  int kchunks;
  int dh0, dw0;         // source offset of patch row/col 0 relative to the region origin (= -1 here)
  const float* up;      // U[jd][16][Npad][Cin]
  uint32_t up_bytes;
};

// U[jd][xi = r*4 + c][n][ci] = sum_{a,b} G[r][a] G[c][b] g'[a][b],  g'[dh+1][dw+1] = wp[tap with offsets (dh,dw)]
__global__ void wino_weights_kernel(const rehr_gather_gemm_desc d, float* __restrict__ up) {
  const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  const int64_t per = (int64_t)d.Npad * d.Cin;
  const int64_t total = (int64_t)d.td.count * per;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int jd = (int)(i / per);
    const int64_t nc = i - (int64_t)jd * per;
    float g[3][3];
#pragma unroll
    for (int jh = 0; jh < 3; ++jh)
#pragma unroll
      for (int jw = 0; jw < 3; ++jw) {
        const int a = d.bh + d.th.off0 + d.th.offs * jh + 1;   // source offset + 1 in {0,1,2}
        const int b = d.bw + d.tw.off0 + d.tw.offs * jw + 1;
        const int wt = ((d.td.k0 + d.td.ks * jd) * d.KH + (d.th.k0 + d.th.ks * jh)) * d.KW + (d.tw.k0 + d.tw.ks * jw);
        const float v = d.wp[(int64_t)wt * per + nc];
#pragma unroll
        for (int aa = 0; aa < 3; ++aa)
#pragma unroll
          for (int bb = 0; bb < 3; ++bb)
            if (aa == a && bb == b) g[aa][bb] = v;
      }
    float t[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int b = 0; b < 3; ++b) t[r][b] = G[r][0] * g[0][b] + G[r][1] * g[1][b] + G[r][2] * g[2][b];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        up[((int64_t)jd * 16 + r * 4 + c) * per + nc] = t[r][0] * G[c][0] + t[r][1] * G[c][1] + t[r][2] * G[c][2];
  }
}

__global__ __launch_bounds__(256, 2) void wino_conv_kernel(const WinoParams p) {
  const rehr_gather_gemm_desc& d = p.d;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;  // [PVOX][LDX]; reused as the row-combine exchange buffer at the end

  const int tid = threadIdx.x, lane = tid & 63;
  const int r = __builtin_amdgcn_readfirstlane(tid >> 6);  // Winograd row of this wave
  const int half = lane >> 5, col = lane & 31;
  const int n_img = blockIdx.z;
  const int n0 = blockIdx.y * 32;
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int bw_ = b % p.nb_w; b /= p.nb_w;
  const int bh_ = b % p.nb_h;
  const int od = b / p.nb_h;
  const int oh0 = bh_ * 2 * TH, ow0 = bw_ * 2 * TW;

  // B^T rows: V[r] = s1 * d[i1] + s2 * d[i2]
  const int i1 = (r == 0) ? 0 : 1, i2 = (r == 3) ? 3 : 2;
  const float s1 = (r == 2) ? -1.f : 1.f, s2 = (r == 0 || r == 3) ? -1.f : 1.f;

  // lane's tile -> patch origin
  const int t_ = col;  // tile index = MFMA row
  const int th_ = t_ / TW, tw_ = t_ % TW;
  const float* xa = Xs + ((2 * th_ + i1) * PW + 2 * tw_) * LDX + 4 * half;
  const float* xb = Xs + ((2 * th_ + i2) * PW + 2 * tw_) * LDX + 4 * half;

  // staging pieces of this thread
  const uint32_t img_elems = (uint32_t)d.Di * d.Hi * d.Wi;
  int pv[NX], pq[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int piece = tid + 256 * i;
    pv[i] = piece < PVOX * 8 ? piece >> 3 : -1;
    pq[i] = piece & 7;
  }
  f32x4 rx[NX];
  const int items = p.kchunks * d.td.count;
  auto fetch = [&](int it) {
    const bool live = it < items;
    const int ii = live ? it : 0;
    const int jd = ii % d.td.count;
    const int cc = (ii / d.td.count) * 32;
    const int id = od + d.bd + d.td.off0 + d.td.offs * jd;
    const bool first = cc < d.c1;
    const float* src = first ? d.x1 : d.x2;
    const uint32_t ld = (uint32_t)(first ? d.ldx1 : d.ldx2);
    const int coff = first ? cc : cc - d.c1;
    const uint32_t nrec = img_elems * ld * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(src) + (int64_t)n_img * img_elems * ld, 0, nrec, 0x00020000);
    const bool dok = live & ((unsigned)id < (unsigned)d.Di);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int v = pv[i] < 0 ? 0 : pv[i];
      const int ph = v / PW, pw_ = v - ph * PW;
      const int ih = oh0 + p.dh0 + ph, iw = ow0 + p.dw0 + pw_;
      const bool ok = dok & (pv[i] >= 0) & ((unsigned)ih < (unsigned)d.Hi) & ((unsigned)iw < (unsigned)d.Wi) &
                      ((cc + pq[i] * 4) < d.Cin);
      const uint32_t off = (uint32_t)((id * d.Hi + ih) * d.Wi + iw) * ld * 4u + (uint32_t)(coff + pq[i] * 4) * 4u;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : nrec, 0, 0));
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NX; ++i)
      if (pv[i] >= 0) *reinterpret_cast<f32x4*>(Xs + pv[i] * LDX + pq[i] * 4) = rx[i];
  };

  // transformed weights: lane's B fragment of (jd, xi = r*4 + c), k-group kk
  const __amdgpu_buffer_rsrc_t rsu =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.up), 0, p.up_bytes, 0x00020000);
  const uint32_t per_b = (uint32_t)d.Npad * d.Cin * 4u;
  const uint32_t ulane = ((uint32_t)(n0 + col) * d.Cin + 4u * half) * 4u;
  auto load_u = [&](int it, int c, f32x4 (&ub)[4]) {
    const int jd = it % d.td.count;
    const int cc = (it / d.td.count) * 32;
    const uint32_t base = (uint32_t)(jd * 16 + r * 4 + c) * per_b + (uint32_t)cc * 4u + ulane;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      ub[kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, base + kk * 32u, 0, 0));
  };

  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;

  fetch(0);
  stage();
  __syncthreads();

  f32x4 ub0[4], ub1[4];
  for (int it = 0; it < items; ++it) {
    fetch(it + 1);
    load_u(it, 0, ub0);
    // row combine of the lane's 4x4 patch rows: R[j][kk] (j = patch column), all 4 k-groups
    f32x4 R[4][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xa + j * LDX + kk * 8);
        const f32x4 bq = *reinterpret_cast<const f32x4*>(xb + j * LDX + kk * 8);
        R[j][kk] = a * s1 + bq * s2;
      }
    // column combine + MFMAs, one Winograd column at a time, next column's weights in flight
    auto column = [&](const int c, const f32x4 (&ub)[4]) {
      const int ja = (c == 0) ? 0 : 1, jb = (c == 3) ? 3 : 2;
      const float sa = (c == 2) ? -1.f : 1.f, sb = (c == 0 || c == 3) ? -1.f : 1.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const f32x4 v = R[ja][kk] * sa + R[jb][kk] * sb;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[e], ub[kk][e], acc[c], 0, 0, 0);
      }
    };
    load_u(it, 1, ub1);
    column(0, ub0);
    load_u(it, 2, ub0);
    column(1, ub1);
    load_u(it, 3, ub1);
    column(2, ub0);
    column(3, ub1);
    __syncthreads();  // every wave is done with this slice
    stage();
    __syncthreads();
  }

  // ---- output transform.  Columns (registers): T[c'] for c' = 0, 1
  f32x16 T0 = acc[0] + acc[1] + acc[2];
  f32x16 T1 = acc[1] - acc[2] - acc[3];
  // rows across waves through LDS: ex[r][c'][reg][lane]
  float* ex = smem;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    ex[((r * 2 + 0) * 16 + q) * 64 + lane] = T0[q];
    ex[((r * 2 + 1) * 16 + q) * 64 + lane] = T1[q];
  }
  __syncthreads();
  // wave w -> output position (r' = w >> 1, c' = w & 1) of every tile
  const int ro = r >> 1, co = r & 1;
  const int col_n = n0 + col;
  const bool colok = col_n < d.Cout;
  const float bv = (d.bias != nullptr && colok) ? d.bias[col_n] : 0.f;
  float s1_ = 0.f, s2_ = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float t0 = ex[((0 * 2 + co) * 16 + q) * 64 + lane], t1 = ex[((1 * 2 + co) * 16 + q) * 64 + lane],
                t2 = ex[((2 * 2 + co) * 16 + q) * 64 + lane], t3 = ex[((3 * 2 + co) * 16 + q) * 64 + lane];
    const float yv = ro == 0 ? (t0 + t1 + t2) : (t1 - t2 - t3);
    const int tile = (q & 3) + 8 * (q >> 2) + 4 * half;  // MFMA C row = tile index
    const int oh = oh0 + 2 * (tile / TW) + ro, ow = ow0 + 2 * (tile % TW) + co;
    const float v = apply_act(yv + bv, d.act, d.slope);
    if (colok && oh < d.Lh && ow < d.Lw) {
      d.y[((((int64_t)n_img * d.Dy + od) * d.Hy + oh) * d.Wy + ow) * d.ldy + col_n] = v;
      s1_ += v;
      s2_ += v * v;
    }
  }
  if (d.stats_mode != 0) {
    s1_ += __shfl_xor(s1_, 32, 64);
    s2_ += __shfl_xor(s2_, 32, 64);
    if (half == 0 && colok) {
      double* st = d.stats + ((int64_t)n_img * d.Cout + col_n) * 2;
      atomicAdd(st, (double)s1_);
      if (d.stats_mode == 2) atomicAdd(st + 1, (double)s2_);
    }
  }
}

bool three_taps(const rehr_axis_taps& t, int b) {
  if (t.count != 3) return false;
  const int o0 = b + t.off0, o1 = b + t.off0 + t.offs, o2 = b + t.off0 + 2 * t.offs;
  return (o1 == 0) && ((o0 == -1 && o2 == 1) || (o0 == 1 && o2 == -1));
}

}  // namespace

// scratch bytes when the descriptor suits the kernel, else 0
int64_t wino_workspace_bytes(const rehr_gather_gemm_desc& d) {
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return 0;
  if (d.osd != 1 || d.osh != 1 || d.osw != 1 || d.obd || d.obh || d.obw) return 0;
  if (d.Ld != d.Dy || d.Lh != d.Hy || d.Lw != d.Wy) return 0;
  if (!three_taps(d.th, d.bh) || !three_taps(d.tw, d.bw)) return 0;
  if (d.td.count < 1 || d.td.count > 3) return 0;
  if (d.Lh < 8 || d.Lw < 8) return 0;
  const int64_t nb_h = (d.Lh + 2 * TH - 1) / (2 * TH), nb_w = (d.Lw + 2 * TW - 1) / (2 * TW);
  if (nb_h * 2 * TH * nb_w * 2 * TW * 10 > (int64_t)d.Lh * d.Lw * 13) return 0;
  const int64_t need = (int64_t)d.td.count * 16 * d.Npad * d.Cin * (int64_t)sizeof(float);
  if (need >= (1ll << 32) - 64) return 0;
  const int64_t img = (int64_t)d.Di * d.Hi * d.Wi * 4;
  if (img * d.ldx1 >= (1ll << 32) - 64 || (d.x2 && img * d.ldx2 >= (1ll << 32) - 64)) return 0;
  if (nb_h * nb_w * d.Ld >= (1ll << 31) || d.Npad / 32 > 65535 || d.N > 65535) return 0;
  return need;
}

// REHR_OK launched; REHR_ENOSUP not applicable.
int wino_conv_try(const rehr_gather_gemm_desc& d, hipStream_t stream) {
  if (!d.wino_ws) return REHR_ENOSUP;
  const int64_t need = wino_workspace_bytes(d);
  if (need == 0 || d.wino_ws_bytes < need || ((uintptr_t)d.wino_ws & 15)) return REHR_ENOSUP;
  const int64_t nb_h = (d.Lh + 2 * TH - 1) / (2 * TH), nb_w = (d.Lw + 2 * TW - 1) / (2 * TW);

  // weight transform (reads the packed panel, writes the workspace)
  {
    const int64_t total = (int64_t)d.td.count * d.Npad * d.Cin;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d, d.wino_ws);
  }
  WinoParams p;
  p.d = d;
  p.nb_h = (int)nb_h;
  p.nb_w = (int)nb_w;
  p.kchunks = (d.Cin + 31) / 32;
  p.dh0 = -1;
  p.dw0 = -1;
  p.up = d.wino_ws;
  p.up_bytes = (uint32_t)need;
  const size_t smem_x = (size_t)PVOX * LDX * sizeof(float), smem_e = (size_t)4 * 2 * 16 * 64 * sizeof(float);
  const size_t smem = smem_x > smem_e ? smem_x : smem_e;
  dim3 grid((unsigned)(nb_h * nb_w * d.Ld), d.Npad / 32, d.N);
  hipLaunchKernelGGL(wino_conv_kernel, grid, dim3(256), smem, stream, p);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
