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
#include <cstdlib>

namespace {

constexpr int TH = 4, TW = 8;                     // Winograd tiles per block
constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;   // staged input patch: 10 x 18
constexpr int PVOX = PH * PW;                     // 180
constexpr int LDX = 36;
constexpr int NX = (PVOX * 8 + 255) / 256;        // 16-byte pieces staged per thread (6)

struct WinoParams {
  rehr_gather_gemm_desc d;
  int nb_h, nb_w;       // 8 x 16 output regions per depth slice
  int band_major;       // tile order (bw, od, bh) instead of (bw, bh, od): see the kernels
  int kchunks;
  int dh0, dw0;         // source offset of patch row/col 0 relative to the region origin (= -1 here)
  const float* up;      // U[jd][16][Npad][Cin]
  uint32_t up_bytes;
  // split-K over depth-tap ranges (big-tile kernel only; feature_fuse: 128 depth taps on one output slice):
  // grid.z = sample * nsplit + part, every part with its own taps, transformed weights and output slab
  int nsplit;
  rehr_axis_taps s_td[8];
  const float* s_up[8];
  uint32_t s_up_bytes[8];
  float* s_y[8];
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
  float* Xs = smem;  // [2][PVOX][LDX]; reused as the row-combine exchange buffer at the end

  const int tid = threadIdx.x, lane = tid & 63;
  const int r = __builtin_amdgcn_readfirstlane(tid >> 6);  // Winograd row of this wave
  const int half = lane >> 5, col = lane & 31;
  const int n_img = blockIdx.z;
  const int n0 = blockIdx.y * 32;
  // Tile order: xcd_remap gives every XCD a contiguous range of logical tiles.  band_major (default): that range walks
  // DEPTH inside one band of output rows (bw fastest, then od, then bh), so the 64 blocks an XCD has in flight cover
  // ~8 consecutive slices of one band and the three source slices of a tile are L2 hits left by its depth neighbours
  // (18 rows x W x 32 channels = 0.3 MB per slice and band against 4 MB of L2).  Slice-major order (bw, bh, od) makes
  // the in-flight set one whole slice: every source slice is fetched for each of its three depth taps.
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int bw_ = b % p.nb_w; b /= p.nb_w;
  const int bh_ = p.band_major ? b / p.d.Ld : b % p.nb_h;
  const int od = p.band_major ? b % p.d.Ld : b / p.nb_h;
  const int oh0 = bh_ * 2 * TH, ow0 = bw_ * 2 * TW;

  // B^T rows: (d0 - d2, d1 + d2, d2 - d1, d1 - d3).  The lane computes R = d[i1] + s2 * d[i2]
  // (one fma); row 2 comes out negated, which the output transform takes back (rsign).
  const int i1 = (r == 0) ? 0 : 1, i2 = (r == 3) ? 3 : 2;
  const float s2 = (r == 1) ? 1.f : -1.f;
  const float rsign = (r == 2) ? -1.f : 1.f;

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
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i)
      if (pv[i] >= 0) *reinterpret_cast<f32x4*>(Xs + buf + pv[i] * LDX + pq[i] * 4) = rx[i];
  };

  // transformed weights: lane's B fragment of (jd, xi = r*4 + c), k-group kk
  const __amdgpu_buffer_rsrc_t rsu =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.up), 0, p.up_bytes, 0x00020000);
  const uint32_t per_b = (uint32_t)d.Npad * d.Cin * 4u;
  const uint32_t ulane = ((uint32_t)(n0 + col) * d.Cin + 4u * half) * 4u;
  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;

  // two LDS slices: item it is read from slice it&1 while item it+1 (fetched one item earlier
  // into rx) is written to the other one and item it+2 is fetched; one barrier per item.
  // Inside an item the 4 k-groups are software-pipelined by hand: while k-group kk feeds the
  // matrix cores, the LDS reads and the weight fragments of kk+1 are in flight.
  auto read_r = [&](int cur, int kk, f32x4 (&R)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xa + cur + j * LDX + kk * 8);
      const f32x4 bq = *reinterpret_cast<const f32x4*>(xb + cur + j * LDX + kk * 8);
      R[j] = a + bq * s2;
    }
  };
  auto load_u = [&](int it, int kk, f32x4 (&ub)[4]) {
    const int jd = it % d.td.count;
    const int cc = (it / d.td.count) * 32;
    const uint32_t base = (uint32_t)(jd * 16 + r * 4) * per_b + (uint32_t)(cc + kk * 8) * 4u + ulane;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      ub[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, base + c * per_b, 0, 0));
  };
  // columns (d0 - d2, d1 + d2, d2 - d1 [negated here, undone at the output], d1 - d3), then 16 MFMAs
  auto compute = [&](const f32x4 (&R)[4], const f32x4 (&ub)[4]) {
    f32x4 v[4];
    v[0] = R[0] - R[2];
    v[1] = R[1] + R[2];
    v[2] = R[1] - R[2];
    v[3] = R[1] - R[3];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][e], ub[c][e], acc[c], 0, 0, 0);
  };

  fetch(0);
  stage(0);
  fetch(1);
  f32x4 Ra[4], Rb[4], ua[4], ubb[4];
  load_u(0, 0, ua);
  __syncthreads();

  for (int it = 0; it < items; ++it) {
    const int cur = (it & 1) * (PVOX * LDX);
    read_r(cur, 0, Ra);
    read_r(cur, 1, Rb);
    load_u(it, 1, ubb);
    stage(cur ^ (PVOX * LDX));  // item it+1 (zeros past the end)
    __builtin_amdgcn_sched_barrier(0);
    compute(Ra, ua);
    __builtin_amdgcn_sched_barrier(0);
    read_r(cur, 2, Ra);
    load_u(it, 2, ua);
    fetch(it + 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(Rb, ubb);
    __builtin_amdgcn_sched_barrier(0);
    read_r(cur, 3, Rb);
    load_u(it, 3, ubb);
    __builtin_amdgcn_sched_barrier(0);
    compute(Ra, ua);
    __builtin_amdgcn_sched_barrier(0);
    load_u(it + 1 < items ? it + 1 : it, 0, ua);
    __builtin_amdgcn_sched_barrier(0);
    compute(Rb, ubb);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }

  // ---- output transform.  Columns (registers): T[c'] for c' = 0, 1
  // (column 2 and row 2 were accumulated negated)
  f32x16 T0 = (acc[0] + acc[1] - acc[2]) * rsign;
  f32x16 T1 = (acc[1] + acc[2] - acc[3]) * rsign;
  // rows across waves through LDS: ex[r][c'][reg][lane]
  float* ex = smem;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    ex[((r * 2 + 0) * 16 + q) * 64 + lane] = T0[q];
    ex[((r * 2 + 1) * 16 + q) * 64 + lane] = T1[q];
  }
  __syncthreads();
  // wave w -> output position (r' = w >> 1, c' = w & 1) of every tile; branch-free so the LDS
  // reads go out back to back
  const int ro = r >> 1, co = r & 1;
  const float k0 = ro == 0 ? 1.f : 0.f, k2 = ro == 0 ? 1.f : -1.f, k3 = ro == 0 ? 0.f : -1.f;
  const float neg_slope = d.act == REHR_ACT_NONE ? 1.f : (d.act == REHR_ACT_RELU ? 0.f : d.slope);
  const int col_n = n0 + col;
  const bool colok = col_n < d.Cout;
  const float bv = (d.bias != nullptr && colok) ? d.bias[col_n] : 0.f;
  float s1_ = 0.f, s2_ = 0.f;
  float t[4][16];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int q = 0; q < 16; ++q) t[rr][q] = ex[((rr * 2 + co) * 16 + q) * 64 + lane];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float yv = k0 * t[0][q] + t[1][q] + k2 * t[2][q] + k3 * t[3][q] + bv;
    const float v = fmaxf(yv, 0.f) + neg_slope * fminf(yv, 0.f);
    const int tile = (q & 3) + 8 * (q >> 2) + 4 * half;  // MFMA C row = tile index
    const int oh = oh0 + 2 * (tile / TW) + ro, ow = ow0 + 2 * (tile % TW) + co;
    const bool ok = colok & (oh < d.Lh) & (ow < d.Lw);
    if (ok) d.y[((((int64_t)n_img * d.Dy + od) * d.Hy + oh) * d.Wy + ow) * d.ldy + col_n] = v;
    s1_ += ok ? v : 0.f;
    s2_ += ok ? v * v : 0.f;
  }
  if (d.stats_mode != 0) {  // block-level sums first: one atomic per column and block
    s1_ += __shfl_xor(s1_, 32, 64);
    s2_ += __shfl_xor(s2_, 32, 64);
    __syncthreads();  // everybody is done reading ex
    float* red = smem;
    if (half == 0) {
      red[(r * 2 + 0) * 32 + col] = s1_;
      red[(r * 2 + 1) * 32 + col] = s2_;
    }
    __syncthreads();
    if (r == 0 && half == 0 && colok) {
      const float a1 = red[col] + red[64 + col] + red[128 + col] + red[192 + col];
      const float a2 = red[32 + col] + red[96 + col] + red[160 + col] + red[224 + col];
      double* st = d.stats + ((int64_t)n_img * d.Cout + col_n) * 2;
      atomicAdd(st, (double)a1);
      if (d.stats_mode == 2) atomicAdd(st + 1, (double)a2);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Big-tile variant: 64 Winograd tiles (16 x 16 outputs) x 64 channels per block, ONE block per CU
// with the full register file.  Each weight fragment feeds 2 tile groups and each input fragment 2
// channel groups, which halves the L2->CU operand traffic per MFMA -- the limiter of the small-tile
// kernel above (measured: 2.16 ms with, 1.45 ms without operand loads on 512->512 @ 128x16x16).
// Latency hiding is explicit: the LDS reads / weight loads of the next k-group step in flight under
// the MFMAs of the current one; one barrier per item, placed where nothing is pending.
// (Rounds 1-2 ran this tile as 4 waves, one per SIMD with 16 accumulator tiles each; the 8-wave
// organisation below replaced it: -1.5 ms per cfg-2 step, profiles/r03_ab_superseded.txt.)
constexpr int PW2 = 18, PVOX2 = 18 * 18;
// LDS patch layout of the big-tile kernel: within a row the 9 even columns come first, then the 9 odd
// ones, and rows are padded by 8 floats.  A fragment read touches, per 16-lane phase, tiles (th 0..1,
// tw 0..7) at ONE column parity: neighbouring tiles are then 36 floats (9 bank groups, odd) apart and
// the two tile rows 32 floats (8 groups) apart -> 16 distinct 4-bank groups.  Voxel-major order had
// both strides even: SQ_LDS_BANK_CONFLICT was 68 % of SQ_LDS_IDX_ACTIVE.
constexpr int RP2 = PW2 * LDX + 8;     // row pitch in floats
constexpr int BUF2 = PW2 * RP2;        // floats per slice buffer

// U2 in MFMA fragment order: [jd][xi][Npad/32][kchunks][kk][lane 64][4]; lane = half*32 + col
// holds n = nt*32 + col, ci = chunk*32 + kk*8 + half*4 + e (zero beyond Cin)
__global__ void wino_weights_frag_kernel(const rehr_gather_gemm_desc d, float* __restrict__ up, int kchunks) {
  const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  const int cpad = kchunks * 32;
  const int64_t per = (int64_t)d.Npad * cpad;
  const int64_t per_src = (int64_t)d.Npad * d.Cin;
  const int64_t total = (int64_t)d.td.count * per;
  const int NT = d.Npad / 32;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int jd = (int)(i / per);
    const int64_t rem = i - (int64_t)jd * per;
    const int n = (int)(rem / cpad), ci = (int)(rem - (int64_t)n * cpad);
    float g[3][3];
#pragma unroll
    for (int jh = 0; jh < 3; ++jh)
#pragma unroll
      for (int jw = 0; jw < 3; ++jw) {
        const int a = d.bh + d.th.off0 + d.th.offs * jh + 1;
        const int b = d.bw + d.tw.off0 + d.tw.offs * jw + 1;
        const int wt = ((d.td.k0 + d.td.ks * jd) * d.KH + (d.th.k0 + d.th.ks * jh)) * d.KW + (d.tw.k0 + d.tw.ks * jw);
        const float v = ci < d.Cin ? d.wp[(int64_t)wt * per_src + (int64_t)n * d.Cin + ci] : 0.f;
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
    const int nt = n >> 5, col = n & 31, chunk = ci >> 5, kk = (ci >> 3) & 3, half = (ci >> 2) & 1, e = ci & 3;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int64_t o = (((((int64_t)jd * 16 + r * 4 + c) * NT + nt) * kchunks + chunk) * 4 + kk) * 256 +
                          (half * 32 + col) * 4 + e;
        up[o] = t[r][0] * G[c][0] + t[r][1] * G[c][1] + t[r][2] * G[c][2];
      }
  }
}

// ------------------------------------------------------------------------------------------
// Eight-wave organisation of the big-tile kernel: wave =
// (Winograd row r, tile group fm), two waves per SIMD with 128 accumulator registers each.  A four-wave kernel
// gives one wave a whole SIMD: whatever stalls its in-order instruction stream (an LDS fragment not yet back, the
// barrier, a weight load) idles the matrix pipe -- measured on 512->512 @ 128x16x16: 1.71 ms against 1.43 ms of MFMA
// time + prologue / epilogue, 1.63 ms with the LDS reads removed, 1.67 ms without the barrier.  With a second resident
// wave the pipe has somebody else to serve; the price is that both tile-group waves of a row fetch the same weight
// fragments (L1 hits) and that the software pipeline is shallower (128 VGPRs): weights of (k-group, channel group)
// arrive one half k-group ahead, A fragments one k-group ahead.
constexpr int NX8 = (PVOX2 * 8 + 511) / 512;   // 16-byte pieces staged per thread (6)

__global__ __launch_bounds__(512) void wino_conv_big8_kernel(const WinoParams p) {
  const rehr_gather_gemm_desc& d = p.d;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;  // [2][BUF2]; reused as the row-combine exchange buffer at the end
  constexpr int BUF = BUF2;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = wv & 3, fm = wv >> 2;
  const int half = lane >> 5, col = lane & 31;
  const int part = (p.nsplit > 1) ? (int)(blockIdx.z % p.nsplit) : 0;
  const int n_img = (p.nsplit > 1) ? (int)(blockIdx.z / p.nsplit) : (int)blockIdx.z;
  const rehr_axis_taps td = (p.nsplit > 1) ? p.s_td[part] : p.d.td;
  const float* const up = (p.nsplit > 1) ? p.s_up[part] : p.up;
  const uint32_t up_bytes = (p.nsplit > 1) ? p.s_up_bytes[part] : p.up_bytes;
  float* const yout = (p.nsplit > 1) ? p.s_y[part] : p.d.y;
  const int nt0 = blockIdx.y * 2, n0 = blockIdx.y * 64;
  // Tile order: xcd_remap gives every XCD a contiguous range of logical tiles.  band_major (default): that range walks
  // DEPTH inside one band of output rows (bw fastest, then od, then bh), so the 64 blocks an XCD has in flight cover
  // ~8 consecutive slices of one band and the three source slices of a tile are L2 hits left by its depth neighbours
  // (18 rows x W x 32 channels = 0.3 MB per slice and band against 4 MB of L2).  Slice-major order (bw, bh, od) makes
  // the in-flight set one whole slice: every source slice is fetched for each of its three depth taps.
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int bw_ = b % p.nb_w; b /= p.nb_w;
  const int bh_ = p.band_major ? b / p.d.Ld : b % p.nb_h;
  const int od = p.band_major ? b % p.d.Ld : b / p.nb_h;
  const int oh0 = bh_ * 16, ow0 = bw_ * 16;

  const int i1 = (r == 0) ? 0 : 1, i2 = (r == 3) ? 3 : 2;
  const float s2 = (r == 1) ? 1.f : -1.f;
  const float rsign = (r == 2) ? -1.f : 1.f;
  const int th_ = fm * 4 + (col >> 3), tw_ = col & 7;
  const float* xa = Xs + (2 * th_ + i1) * RP2 + tw_ * LDX + 4 * half;
  const float* xb = Xs + (2 * th_ + i2) * RP2 + tw_ * LDX + 4 * half;

  int pvx[NX8];
  uint32_t pok = 0;
#pragma unroll
  for (int i = 0; i < NX8; ++i) {
    const int piece = tid + 512 * i;
    const int v = piece >> 3;
    const int ph = v / PW2, slot = v - ph * PW2;
    const int pw_ = slot < 9 ? 2 * slot : 2 * (slot - 9) + 1;
    const int ih = oh0 + p.dh0 + ph, iw = ow0 + p.dw0 + pw_;
    const bool ok = (piece < PVOX2 * 8) & ((unsigned)ih < (unsigned)d.Hi) & ((unsigned)iw < (unsigned)d.Wi);
    pvx[i] = ok ? ih * d.Wi + iw : 0;
    pok |= (ok ? 1u : 0u) << i;
  }
  const int pq = tid & 7;
  const uint32_t img_elems = (uint32_t)d.Di * d.Hi * d.Wi;
  constexpr int NXH = NX8 / 2;   // pieces per staging half
  f32x4 rx[NXH];
  int jd_lo = td.count, jd_hi = -1;
  for (int j = 0; j < td.count; ++j) {
    const int id = od + d.bd + td.off0 + td.offs * j;
    if ((unsigned)id < (unsigned)d.Di) { jd_lo = min(jd_lo, j); jd_hi = max(jd_hi, j); }
  }
  const int items = p.kchunks * max(0, jd_hi - jd_lo + 1);
  struct Item { int chunk, jd; };
  auto advance = [&](Item& t) {
    if (++t.jd > jd_hi) { t.jd = jd_lo; ++t.chunk; }
  };
  auto fetch_to = [&](f32x4 (&rx)[NXH], const Item& t, const int lo) {   // pieces lo .. lo + NXH - 1
    const bool live = (t.chunk < p.kchunks) & (items > 0);
    const int jd = t.jd;
    const int cc = (live ? t.chunk : 0) * 32;
    const int id = od + d.bd + td.off0 + td.offs * jd;
    const bool first = cc < d.c1;
    const float* src = first ? d.x1 : d.x2;
    const uint32_t ld = (uint32_t)(first ? d.ldx1 : d.ldx2);
    const int coff = first ? cc : cc - d.c1;
    const uint32_t nrec = img_elems * ld * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(src) + (int64_t)n_img * img_elems * ld, 0, nrec, 0x00020000);
    const bool dok = live & ((unsigned)id < (unsigned)d.Di) & ((cc + pq * 4) < d.Cin);
    const uint32_t base = (uint32_t)(id * d.Hi * d.Wi) * ld * 4u + (uint32_t)(coff + pq * 4) * 4u;
#pragma unroll
    for (int i = 0; i < NXH; ++i) {
      const bool ok = dok & ((pok >> (lo + i)) & 1u);
      const uint32_t off = base + (uint32_t)pvx[lo + i] * ld * 4u;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : nrec, 0, 0));
    }
  };
  auto stage_from = [&](const f32x4 (&rx)[NXH], int buf, const int lo) {
#pragma unroll
    for (int i = 0; i < NXH; ++i) {
      const int piece = tid + 512 * (lo + i);
      const int v = piece >> 3, row = (v * 3641) >> 16;  // v / 18 for v < 1024
      if (piece < PVOX2 * 8) *reinterpret_cast<f32x4*>(Xs + buf + v * LDX + row * 8 + pq * 4) = rx[i];
    }
  };
  auto fetch = [&](const Item& t, const int lo) { fetch_to(rx, t, lo); };
  auto stage = [&](int buf, const int lo) { stage_from(rx, buf, lo); };

  const __amdgpu_buffer_rsrc_t rsu =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(up), 0, up_bytes, 0x00020000);
  const int NT = d.Npad / 32;
  const uint32_t xi_stride = (uint32_t)NT * p.kchunks * 4096u, nt_stride = (uint32_t)p.kchunks * 4096u;
  const uint32_t ulane = (uint32_t)lane * 16u;
  const uint32_t ubase = (uint32_t)(r * 4) * xi_stride + (uint32_t)nt0 * nt_stride;
  auto load_u = [&](const Item& t, const int kk, const int fn, f32x4 (&ub)[4]) {
    const int chunk = t.chunk < p.kchunks ? t.chunk : 0;  // (one item past the end is requested, never used)
    const uint32_t base = ubase + (uint32_t)(t.jd * 16) * xi_stride + (uint32_t)(chunk * 4 + kk) * 1024u + fn * nt_stride;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      ub[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, ulane, base + c * xi_stride, 0));
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int fn = 0; fn < 2; ++fn)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[fn][c][q] = 0.f;

  f32x4 ra[4], rb[4];
  auto issue_reads = [&](int buf, const int kk) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ra[j] = *reinterpret_cast<const f32x4*>(xa + buf + ((j & 1) * 9 + (j >> 1)) * LDX + kk * 8);
      rb[j] = *reinterpret_cast<const f32x4*>(xb + buf + ((j & 1) * 9 + (j >> 1)) * LDX + kk * 8);
    }
  };
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  struct VFrag { f32x2 p[4][2]; };
  const f32x2 s2v = {s2, s2};
  auto combine = [&](VFrag& v) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x2 R[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 a = h ? ra[j].hi : ra[j].lo, bq = h ? rb[j].hi : rb[j].lo;
        R[j] = __builtin_elementwise_fma(bq, s2v, a);
      }
      v.p[0][h] = R[0] - R[2];
      v.p[1][h] = R[1] + R[2];
      v.p[2][h] = R[1] - R[2];  // negated column, undone at the output
      v.p[3][h] = R[1] - R[3];
    }
  };
  auto mfmas = [&](const int fn, const VFrag& v, const f32x4 (&ub)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[fn][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.p[c][e >> 1][e & 1], ub[c][e], acc[fn][c], 0, 0, 0);
  };

  VFrag VA, VB;
  f32x4 u0[4], u1[4];   // weights of channel group 0 / 1 of the current k-group
  Item ci = {0, min(jd_lo, td.count - 1)}, ni = ci;
  {  // item 0: both halves of the patch in flight at once (u1's registers are free here): one memory round trip
    static_assert(NXH <= 4, "the second half borrows u1");
    f32x4 (&rx2)[NXH] = reinterpret_cast<f32x4 (&)[NXH]>(u1);
    fetch(ci, 0);
    fetch_to(rx2, ci, NXH);
    load_u(ci, 0, 0, u0);
    stage(0, 0);
    stage_from(rx2, 0, NXH);
  }
  load_u(ci, 0, 1, u1);
  __syncthreads();
  issue_reads(0, 0);
  combine(VA);

  // half step = (k-group kk, channel group fn) = 16 MFMAs; behind them: the weights this half step has just released
  // are re-loaded for the next k-group, the A fragments of the next k-group are read and combined, the next item's
  // patch is fetched (kk 0, 2) and staged (kk 1, 3)
#ifndef BIG8_FENCE
#define BIG8_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifdef BIG8_NO_PRIO   // (A/B builds)
#define BIG8_PRIO(n) (void)0
#else
#define BIG8_PRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
#define BIG8_KGROUP(kk, vcur, vnext, NEXT_T, NEXT_KK, READS, EXTRA0, EXTRA1)  \
  BIG8_FENCE();                                                               \
  READS;                                                                      \
  EXTRA0;                                                                     \
  mfmas(0, vcur, u0);                                                         \
  BIG8_FENCE();                                                               \
  load_u(NEXT_T, NEXT_KK, 0, u0);                                             \
  EXTRA1;                                                                     \
  mfmas(1, vcur, u1);                                                         \
  combine(vnext);                                                             \
  BIG8_FENCE();                                                               \
  load_u(NEXT_T, NEXT_KK, 1, u1);

#ifdef BIG8_STAMPS
  long long b8_bar = 0;
  const long long b8_t0 = __builtin_readcyclecounter();
#endif
  for (int it = 0; it < items; ++it) {
    const int cur = (it & 1) * BUF, nxt = cur ^ BUF;
    advance(ni);
    // (s_setprio: a wave's issue priority falls with its progress since the item barrier, so that the arbiter -- oldest
    // wave first by default -- serves the wave of a SIMD that is behind; see wino22_conv.hip)
    BIG8_PRIO(2);
    BIG8_KGROUP(0, VA, VB, ci, 1, issue_reads(cur, 1), fetch(ni, 0), (void)0)
    BIG8_PRIO(1);
    BIG8_KGROUP(1, VB, VA, ci, 2, issue_reads(cur, 2), (void)0, stage(nxt, 0))
    BIG8_KGROUP(2, VA, VB, ci, 3, issue_reads(cur, 3), fetch(ni, NXH), (void)0)
    // last k-group of the item: its A fragments are in VB and nobody reads the current patch any more, so the barrier
    // sits between its two half steps and the next item's first fragments are read and combined behind 16 MFMAs
    __builtin_amdgcn_sched_barrier(0);
    BIG8_PRIO(0);
    stage(nxt, NXH);
    mfmas(0, VB, u0);
    __builtin_amdgcn_sched_barrier(0);
    load_u(ni, 0, 0, u0);
#ifdef BIG8_STAMPS
    const long long sa_ = __builtin_readcyclecounter();
#endif
    __syncthreads();
#ifdef BIG8_STAMPS
    b8_bar += __builtin_readcyclecounter() - sa_;
#endif
    __builtin_amdgcn_sched_barrier(0);
    BIG8_PRIO(3);
    issue_reads(nxt, 0);
    mfmas(1, VB, u1);
    combine(VA);
    __builtin_amdgcn_sched_barrier(0);
    load_u(ni, 0, 1, u1);
    ci = ni;
  }
#undef BIG8_KGROUP
  BIG8_PRIO(0);
#ifdef BIG8_STAMPS
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 777 && blockIdx.y == 0 && blockIdx.z == 0)
    printf("big8 wave %d items %d: loop %lld cycles, barrier waits %lld\n", (int)(threadIdx.x >> 6), items,
           (long long)(__builtin_readcyclecounter() - b8_t0), b8_bar);
#endif
  __syncthreads();

  // ---- output transform: columns in registers, rows across the 4 row-waves of a tile group through LDS
  float* ex = smem;  // [fm*2+fn][r][c'][q][lane]
#pragma unroll
  for (int fn = 0; fn < 2; ++fn) {
    const f32x16 T0 = (acc[fn][0] + acc[fn][1] - acc[fn][2]) * rsign;
    const f32x16 T1 = (acc[fn][1] + acc[fn][2] - acc[fn][3]) * rsign;
    float* e0 = ex + (((fm * 2 + fn) * 4 + r) * 2) * 16 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      e0[q * 64] = T0[q];
      e0[(16 + q) * 64] = T1[q];
    }
  }
  __syncthreads();
  const int ro = r >> 1, co = r & 1;
  const float k0 = ro == 0 ? 1.f : 0.f, k2 = ro == 0 ? 1.f : -1.f, k3 = ro == 0 ? 0.f : -1.f;
  const float neg_slope = d.act == REHR_ACT_NONE ? 1.f : (d.act == REHR_ACT_RELU ? 0.f : d.slope);
  float ssum[2][2];
  float* ybase = yout + ((((int64_t)n_img * d.Dy + od) * d.Hy + (oh0 + ro)) * d.Wy + (ow0 + co + 8 * half)) * d.ldy +
                 n0 + col;
  const int64_t rowstep = 2 * (int64_t)d.Wy * d.ldy, colstep = 2 * (int64_t)d.ldy;
  const bool interior = (oh0 + 16 <= d.Lh) & (ow0 + 16 <= d.Lw) & (n0 + 64 <= d.Cout);
#pragma unroll
  for (int fn = 0; fn < 2; ++fn) {
    const int col_n = n0 + fn * 32 + col;
    const bool colok = col_n < d.Cout;
    const float bv = (d.bias != nullptr && colok) ? d.bias[col_n] : 0.f;
    float s1_ = 0.f, s2_ = 0.f;
    const float* e0 = ex + ((fm * 2 + fn) * 4 * 2 + co) * 16 * 64 + lane;
    float t[4][16];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int q = 0; q < 16; ++q) t[rr][q] = e0[(rr * 32 + q) * 64];
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float yv = k0 * t[0][q] + t[1][q] + k2 * t[2][q] + k3 * t[3][q] + bv;
      v[q] = fmaxf(yv, 0.f) + neg_slope * fminf(yv, 0.f);
    }
    float* yb = ybase + fn * 32 + (fm * 4) * rowstep;
    if (interior) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        yb[(q >> 2) * rowstep + (q & 3) * colstep] = v[q];
        s1_ += v[q];
        s2_ += v[q] * v[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int oh = oh0 + 2 * (fm * 4 + (q >> 2)) + ro, ow = ow0 + 2 * ((q & 3) + 4 * half) + co;
        const bool ok = colok & (oh < d.Lh) & (ow < d.Lw);
        if (ok) yb[(q >> 2) * rowstep + (q & 3) * colstep] = v[q];
        s1_ += ok ? v[q] : 0.f;
        s2_ += ok ? v[q] * v[q] : 0.f;
      }
    }
    ssum[fn][0] = s1_ + __shfl_xor(s1_, 32, 64);
    ssum[fn][1] = s2_ + __shfl_xor(s2_, 32, 64);
  }
  if (d.stats_mode != 0) {  // block-level sums first: one atomic per column and block
    __syncthreads();  // everybody is done reading ex
    float* red = smem;   // [wave 8][fn 2][2][32]
    if (half == 0) {
#pragma unroll
      for (int fn = 0; fn < 2; ++fn) {
        red[((wv * 2 + fn) * 2 + 0) * 32 + col] = ssum[fn][0];
        red[((wv * 2 + fn) * 2 + 1) * 32 + col] = ssum[fn][1];
      }
    }
    __syncthreads();
    if (wv < 2 && half == 0) {  // wave fn sums the eight waves' partials of its 32 columns
      const int fn = wv, col_n = n0 + fn * 32 + col;
      if (col_n < d.Cout) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          a1 += red[((w * 2 + fn) * 2 + 0) * 32 + col];
          a2 += red[((w * 2 + fn) * 2 + 1) * 32 + col];
        }
        double* st = d.stats + ((int64_t)n_img * d.Cout + col_n) * 2;
        atomicAdd(st, (double)a1);
        if (d.stats_mode == 2) atomicAdd(st + 1, (double)a2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Wide-tile variant for 32-wide output-channel tiles (nnU-Net stage-0 / last decoder stage, SR
// head): up to 128 Winograd tiles (32 x 16 outputs of one depth slice) x 32 channels per block.  The tile
// groups share every weight fragment, so the small-tile kernel's weight traffic per MFMA from L2 (its
// limiter: 2 blocks/CU on different tiles) drops.  K items are 16-channel half chunks so that the 34 x 18
// patch double-buffers in LDS (row pitch 368 floats, even/odd column split: conflict-free fragment reads).
// (Rounds 1-2 ran it as 16 waves of thread-level parallelism, wino_conv_w32_kernel, 0.58 of the matrix pipe;
// the software-pipelined 8-wave kernel below replaced it: profiles/r03_ab_superseded.txt.)
constexpr int W3_LD = 20, W3_RP = PW2 * W3_LD + 8;
// NFM = tile groups of 32 tiles (4 tile rows) of the staged patch: 4 (32 x 16 outputs) or 2 (16 x 16 outputs)
template <int NFM> struct W3 {
  static constexpr int ROWS = 8 * NFM + 2, BUF = ROWS * W3_RP, VOX = ROWS * PW2, NTHR = 256 * NFM;
  static constexpr int NXT = (VOX * 4 + NTHR - 1) / NTHR;
};

// ------------------------------------------------------------------------------------------
// The same 128 tiles x 32 channels block with the big-tile kernel's hand-built pipeline instead of 16 waves of
// thread-level parallelism: 8 waves = (Winograd row r, tile-group pair g), two per SIMD; a wave owns TWO tile groups
// (2 x 4 accumulator tiles) that share every weight fragment, so a k-group is 2 x 16 MFMAs per wave with the roles of
// the big-tile kernel swapped: the A fragments alternate between two registers sets (V0 = group 2g, V1 = group 2g+1),
// the weights ping-pong between k-groups.  Behind every 16 MFMAs: the other group's fragments are read and combined,
// the next k-group's weights loaded, the next 16-channel item's patch fetched (step 0) and staged (step 2).
// G = 2: 512 threads, 32 x 16 outputs, one block per CU.  G = 1: 256 threads, 16 x 16 outputs, two independent blocks
// per CU -- with few input channels a block lives only a few K items, and one block's prologue / output transform then
// runs behind the other block's MFMAs.
template <int G>
__global__ __launch_bounds__(256 * G, G == 1 ? 2 : 1) void wino_conv_w32p_kernel(const WinoParams p) {
  constexpr int NTHR = 256 * G, VOX = W3<2 * G>::VOX, BUF = W3<2 * G>::BUF + 32;   // + a spare voxel slot
  constexpr int W3P_NX = (VOX * 4 + NTHR - 1) / NTHR;                               // 16-byte pieces per thread
  const rehr_gather_gemm_desc& d = p.d;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;  // [2][BUF]; reused as the exchange buffer at the end

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = wv & 3, g = wv >> 2;
  const int half = lane >> 5, col = lane & 31;
  const int n_img = blockIdx.z;
  const int nt0 = blockIdx.y, n0 = blockIdx.y * 32;
  // Tile order: xcd_remap gives every XCD a contiguous range of logical tiles.  band_major (default): that range walks
  // DEPTH inside one band of output rows (bw fastest, then od, then bh), so the 64 blocks an XCD has in flight cover
  // ~8 consecutive slices of one band and the three source slices of a tile are L2 hits left by its depth neighbours
  // (18 rows x W x 32 channels = 0.3 MB per slice and band against 4 MB of L2).  Slice-major order (bw, bh, od) makes
  // the in-flight set one whole slice: every source slice is fetched for each of its three depth taps.
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int bw_ = b % p.nb_w; b /= p.nb_w;
  const int bh_ = p.band_major ? b / p.d.Ld : b % p.nb_h;
  const int od = p.band_major ? b % p.d.Ld : b / p.nb_h;
  const int oh0 = bh_ * (16 * G), ow0 = bw_ * 16;

  const int i1 = (r == 0) ? 0 : 1, i2 = (r == 3) ? 3 : 2;
  const float s2 = (r == 1) ? 1.f : -1.f;
  const float rsign = (r == 2) ? -1.f : 1.f;
  // tile group 2g (+ W3P_GOFF floats: group 2g + 1 = tile rows +4 = patch rows +8)
  const int th_ = g * 8 + (col >> 3), tw_ = col & 7;
  const float* xa = Xs + (2 * th_ + i1) * W3_RP + tw_ * W3_LD + 4 * half;
  const float* xb = Xs + (2 * th_ + i2) * W3_RP + tw_ * W3_LD + 4 * half;
  constexpr int GOFF = 8 * W3_RP;

  int pvx[W3P_NX];
  uint32_t pok = 0;
#pragma unroll
  for (int i = 0; i < W3P_NX; ++i) {
    const int piece = tid + NTHR * i;
    const int v = piece >> 2;
    const int ph = v / PW2, slot = v - ph * PW2;
    const int pw_ = slot < 9 ? 2 * slot : 2 * (slot - 9) + 1;
    const int ih = oh0 + p.dh0 + ph, iw = ow0 + p.dw0 + pw_;
    const bool ok = (piece < VOX * 4) & ((unsigned)ih < (unsigned)d.Hi) & ((unsigned)iw < (unsigned)d.Wi);
    pvx[i] = ok ? ih * d.Wi + iw : 0;
    pok |= (ok ? 1u : 0u) << i;
  }
  const int pq = tid & 3;
  const uint32_t img_elems = (uint32_t)d.Di * d.Hi * d.Wi;
  const int nhalf = (d.Cin + 15) / 16;
  int jd_lo = d.td.count, jd_hi = -1;
  for (int j = 0; j < d.td.count; ++j) {
    const int id = od + d.bd + d.td.off0 + d.td.offs * j;
    if ((unsigned)id < (unsigned)d.Di) { jd_lo = min(jd_lo, j); jd_hi = max(jd_hi, j); }
  }
  const int items = nhalf * max(0, jd_hi - jd_lo + 1);
  struct Item { int h16, jd; };
  auto advance = [&](Item& t) {
    if (++t.jd > jd_hi) { t.jd = jd_lo; ++t.h16; }
  };
  f32x4 rx[W3P_NX];
  auto fetch = [&](const Item& t) {
    const bool live = (t.h16 < nhalf) & (items > 0);
    const int jd = t.jd;
    const int cc = (live ? t.h16 : 0) * 16;
    const int id = od + d.bd + d.td.off0 + d.td.offs * jd;
    const bool first = cc < d.c1;
    const float* src = first ? d.x1 : d.x2;
    const uint32_t ld = (uint32_t)(first ? d.ldx1 : d.ldx2);
    const int coff = first ? cc : cc - d.c1;
    const uint32_t nrec = img_elems * ld * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(src) + (int64_t)n_img * img_elems * ld, 0, nrec, 0x00020000);
    const bool dok = live & ((unsigned)id < (unsigned)d.Di) & ((cc + pq * 4) < d.Cin);
    const uint32_t base = (uint32_t)(id * d.Hi * d.Wi) * ld * 4u + (uint32_t)(coff + pq * 4) * 4u;
#pragma unroll
    for (int i = 0; i < W3P_NX; ++i) {
      const bool ok = dok & ((pok >> i) & 1u);
      const uint32_t off = base + (uint32_t)pvx[i] * ld * 4u;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : nrec, 0, 0));
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < W3P_NX; ++i) {
      // (pieces past the patch -- zeros -- land in one spare slot behind the buffer: no divergent branch in the loop)
      const int v = min((tid + NTHR * i) >> 2, VOX), row = (v * 3641) >> 16;  // v / 18 for v < 1024
      *reinterpret_cast<f32x4*>(Xs + buf + v * W3_LD + row * 8 + pq * 4) = rx[i];
    }
  };

  const __amdgpu_buffer_rsrc_t rsu =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.up), 0, p.up_bytes, 0x00020000);
  const int NT = d.Npad / 32;
  const uint32_t xi_stride = (uint32_t)NT * p.kchunks * 4096u, nt_stride = (uint32_t)p.kchunks * 4096u;
  const uint32_t ulane = (uint32_t)lane * 16u, ubase = (uint32_t)(r * 4) * xi_stride + (uint32_t)nt0 * nt_stride;
  auto load_u = [&](const Item& t, const int kkl, f32x4 (&ub)[4]) {
    const int h16 = t.h16 < nhalf ? t.h16 : 0;  // (one item past the end is requested, never used)
    const uint32_t base = ubase + (uint32_t)(t.jd * 16) * xi_stride +
                          (uint32_t)((h16 >> 1) * 4 + (h16 & 1) * 2 + kkl) * 1024u;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      ub[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, ulane, base + c * xi_stride, 0));
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[m][c][q] = 0.f;

  f32x4 ra[4], rb[4];
  auto issue_reads = [&](int buf, const int m, const int kkl) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#ifdef W3P_DBG_NOLDS    // (timing diagnostics only: wrong results)
      ra[j] = f32x4{(float)lane, 1.f, 2.f, (float)j}; rb[j] = f32x4{2.f, (float)lane, 1.f, (float)m}; (void)buf; (void)kkl;
#else
      const int o = buf + m * GOFF + ((j & 1) * 9 + (j >> 1)) * W3_LD + kkl * 8;
      ra[j] = *reinterpret_cast<const f32x4*>(xa + o);
      rb[j] = *reinterpret_cast<const f32x4*>(xb + o);
#endif
    }
  };
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  struct VFrag { f32x2 p[4][2]; };
  const f32x2 s2v = {s2, s2};
  auto combine = [&](VFrag& v) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x2 R[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 a = h ? ra[j].hi : ra[j].lo, bq = h ? rb[j].hi : rb[j].lo;
        R[j] = __builtin_elementwise_fma(bq, s2v, a);
      }
      v.p[0][h] = R[0] - R[2];
      v.p[1][h] = R[1] + R[2];
      v.p[2][h] = R[1] - R[2];  // negated column, undone at the output
      v.p[3][h] = R[1] - R[3];
    }
  };
  auto mfmas = [&](const int m, const VFrag& v, const f32x4 (&ub)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.p[c][e >> 1][e & 1], ub[c][e], acc[m][c], 0, 0, 0);
  };

  VFrag V0, V1;
  f32x4 u0[4], u1[4];
  Item ci = {0, min(jd_lo, d.td.count - 1)}, ni = ci;
  fetch(ci);
  load_u(ci, 0, u0);
  load_u(ci, 1, u1);
  stage(0);
  __syncthreads();
  issue_reads(0, 0, 0);
  combine(V0);

#define W3P_FENCE() __builtin_amdgcn_sched_barrier(0)
  for (int it = 0; it < items; ++it) {
    const int cur = (it & 1) * BUF, nxt = cur ^ BUF;
    advance(ni);
    // step 0: (group 0, k-group 0)
    W3P_FENCE();
    issue_reads(cur, 1, 0);
    fetch(ni);
    mfmas(0, V0, u0);
    combine(V1);
    // step 1: (group 1, k-group 0); u0 is released behind it
    W3P_FENCE();
    issue_reads(cur, 0, 1);
    mfmas(1, V1, u0);
    combine(V0);
    W3P_FENCE();
    load_u(ni, 0, u0);
    // step 2: (group 0, k-group 1)
    W3P_FENCE();
    issue_reads(cur, 1, 1);
    stage(nxt);
    mfmas(0, V0, u1);
    combine(V1);
    // the next item's patch is complete and nobody reads the current one any more (step 3's fragments are in V1):
    // the barrier goes HERE, so that the next item's first fragments are read and combined behind step 3's MFMAs
    W3P_FENCE();
#ifndef W3P_DBG_NOBAR   // (timing diagnostics only: wrong results)
    __syncthreads();
#endif
    // step 3: (group 1, k-group 1); u1 is released behind it
    W3P_FENCE();
    issue_reads(nxt, 0, 0);
    mfmas(1, V1, u1);
    combine(V0);
    W3P_FENCE();
    load_u(ni, 1, u1);
    ci = ni;
  }
#undef W3P_FENCE
  __syncthreads();

  // ---- output transform: columns in registers, rows across the 4 row-waves of a tile group through LDS
  float* ex = smem;  // [fm][r][c'][q][lane], fm = 2 g + m
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const f32x16 T0 = (acc[m][0] + acc[m][1] - acc[m][2]) * rsign;
    const f32x16 T1 = (acc[m][1] + acc[m][2] - acc[m][3]) * rsign;
    float* e0 = ex + (((2 * g + m) * 4 + r) * 2) * 16 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      e0[q * 64] = T0[q];
      e0[(16 + q) * 64] = T1[q];
    }
  }
  __syncthreads();
  const int ro = r >> 1, co = r & 1;
  const float k0 = ro == 0 ? 1.f : 0.f, k2 = ro == 0 ? 1.f : -1.f, k3 = ro == 0 ? 0.f : -1.f;
  const float neg_slope = d.act == REHR_ACT_NONE ? 1.f : (d.act == REHR_ACT_RELU ? 0.f : d.slope);
  float* ybase = d.y + ((((int64_t)n_img * d.Dy + od) * d.Hy + (oh0 + ro)) * d.Wy + (ow0 + co + 8 * half)) * d.ldy +
                 n0 + col;
  const int64_t rowstep = 2 * (int64_t)d.Wy * d.ldy, colstep = 2 * (int64_t)d.ldy;
  const bool interior = (oh0 + 16 * G <= d.Lh) & (ow0 + 16 <= d.Lw) & (n0 + 32 <= d.Cout);
  const int col_n = n0 + col;
  const bool colok = col_n < d.Cout;
  const float bv = (d.bias != nullptr && colok) ? d.bias[col_n] : 0.f;
  // statistics in double from the first addition on: a thread sums 32 values here, and InstanceNorm's variance
  // (E[y^2] - mean^2) amplifies every systematic rounding of the two sums
  double s1_ = 0.0, s2_ = 0.0;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int fm = 2 * g + m;
    const float* e0 = ex + (fm * 4 * 2 + co) * 16 * 64 + lane;
    float t[4][16];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int q = 0; q < 16; ++q) t[rr][q] = e0[(rr * 32 + q) * 64];
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float yv = k0 * t[0][q] + t[1][q] + k2 * t[2][q] + k3 * t[3][q] + bv;
      v[q] = fmaxf(yv, 0.f) + neg_slope * fminf(yv, 0.f);
    }
    float* yb = ybase + (fm * 4) * rowstep;
    if (interior) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        yb[(q >> 2) * rowstep + (q & 3) * colstep] = v[q];
        s1_ += (double)v[q];
        s2_ += (double)v[q] * (double)v[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int oh = oh0 + 2 * (fm * 4 + (q >> 2)) + ro, ow = ow0 + 2 * ((q & 3) + 4 * half) + co;
        const bool ok = colok & (oh < d.Lh) & (ow < d.Lw);
        if (ok) yb[(q >> 2) * rowstep + (q & 3) * colstep] = v[q];
        s1_ += ok ? (double)v[q] : 0.0;
        s2_ += ok ? (double)v[q] * (double)v[q] : 0.0;
      }
    }
  }
  if (d.stats_mode != 0) {  // block-level sums first: one atomic per column and block
    s1_ += __shfl_xor(s1_, 32, 64);
    s2_ += __shfl_xor(s2_, 32, 64);
    __syncthreads();  // everybody is done reading ex
    double* red = reinterpret_cast<double*>(smem);  // [wave][2][32]
    if (half == 0) {
      red[(wv * 2 + 0) * 32 + col] = s1_;
      red[(wv * 2 + 1) * 32 + col] = s2_;
    }
    __syncthreads();
    if (wv == 0 && half == 0 && colok) {
      double a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int w = 0; w < 4 * G; ++w) {
        a1 += red[(w * 2 + 0) * 32 + col];
        a2 += red[(w * 2 + 1) * 32 + col];
      }
      double* st = d.stats + ((int64_t)n_img * d.Cout + col_n) * 2;
      atomicAdd(st, a1);
      if (d.stats_mode == 2) atomicAdd(st + 1, a2);
    }
  }
}

bool three_taps(const rehr_axis_taps& t, int b) {
  if (t.count != 3) return false;
  const int o0 = b + t.off0, o1 = b + t.off0 + t.offs, o2 = b + t.off0 + 2 * t.offs;
  return (o1 == 0) && ((o0 == -1 && o2 == 1) || (o0 == 1 && o2 == -1));
}

bool w32_ok(const rehr_gather_gemm_desc& d) {
  if (d.Npad % 64 == 0 || d.Lh < 16 || d.Lw < 16) return false;  // 64-multiples: the big-tile kernel is faster
  const int64_t nb_h = (d.Lh + 15) / 16, nb_w = (d.Lw + 15) / 16;
  return nb_h * 16 * nb_w * 16 * 10 <= (int64_t)d.Lh * d.Lw * 13;
}
// four tile groups (32 x 16 outputs) when 32-row regions tile the plane within 1.3x, else two (16 x 16 outputs)
int w32_groups(const rehr_gather_gemm_desc& d) {
  const int64_t nb_h = (d.Lh + 31) / 32, nb_w = (d.Lw + 15) / 16;
  if (d.Lh >= 32 && nb_h * 32 * nb_w * 16 * 10 <= (int64_t)d.Lh * d.Lw * 13) return 4;
  return 2;
}

template <int G>
int launch_w32p(const WinoParams& p, const rehr_gather_gemm_desc& d, hipStream_t stream) {
  const size_t smem_x = (size_t)2 * (W3<2 * G>::BUF + 32) * sizeof(float);
  const size_t smem_e = (size_t)2 * G * 4 * 2 * 16 * 64 * sizeof(float);
  const size_t smem = smem_x > smem_e ? smem_x : smem_e;
  static bool attr_set32p = false;
  if (!attr_set32p) {
    if (hipFuncSetAttribute((const void*)wino_conv_w32p_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess)
      return REHR_EHIP;
    attr_set32p = true;
  }
  dim3 grid((unsigned)((int64_t)p.nb_h * p.nb_w * d.Ld), d.Npad / 32, d.N);
  hipLaunchKernelGGL(wino_conv_w32p_kernel<G>, grid, dim3(256 * G), smem, stream, p);
  return REHR_OK;
}

bool big_ok(const rehr_gather_gemm_desc& d) {
  if (d.Npad % 64 || d.Lh < 16 || d.Lw < 16) return false;
  const int64_t nb_h = (d.Lh + 15) / 16, nb_w = (d.Lw + 15) / 16;
  return nb_h * 16 * nb_w * 16 * 10 <= (int64_t)d.Lh * d.Lw * 13;
}

}  // namespace

// fragment-ordered weight transform, shared with wino_flat8_conv.hip
int wino_weights_frag_launch(const rehr_gather_gemm_desc& d, int kchunks, hipStream_t stream) {
  if (d.flags & REHR_GG_WS_READY) return REHR_OK;   // the caller kept the transformed weights of an earlier call
  const int64_t total = (int64_t)d.td.count * d.Npad * kchunks * 32;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(wino_weights_frag_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d, d.wino_ws, kchunks);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

// scratch bytes when the descriptor suits one of the kernels, else 0
int64_t wino_workspace_bytes(const rehr_gather_gemm_desc& d) {
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return 0;
  if (d.osd != 1 || d.osh != 1 || d.osw != 1 || d.obd || d.obh || d.obw) return 0;
  if (d.Ld != d.Dy || d.Lh != d.Hy || d.Lw != d.Wy) return 0;
  if (!three_taps(d.th, d.bh) || !three_taps(d.tw, d.bw)) return 0;
  if (d.td.count < 1 || d.td.count > 256) return 0;  // depth taps are plain K items (feature_fuse's gradient: 128)
  if (d.Lh < 8 || d.Lw < 8) return 0;
  const bool big = big_ok(d) || w32_ok(d);
  const int64_t nb_h = (d.Lh + 2 * TH - 1) / (2 * TH), nb_w = (d.Lw + 2 * TW - 1) / (2 * TW);
  if (!big && nb_h * 2 * TH * nb_w * 2 * TW * 100 > (int64_t)d.Lh * d.Lw * 134) return 0;  // 24 x 24 (1.333) is in
  const int64_t cpad = (int64_t)((d.Cin + 31) / 32) * 32;  // the fragment-order layout pads Cin to 32
  const int64_t need = (int64_t)d.td.count * 16 * d.Npad * cpad * (int64_t)sizeof(float);
  if (need >= (1ll << 32) - 64) return 0;
  const int64_t img = (int64_t)d.Di * d.Hi * d.Wi * 4;
  if (img * d.ldx1 >= (1ll << 32) - 64 || (d.x2 && img * d.ldx2 >= (1ll << 32) - 64)) return 0;
  if (nb_h * nb_w * d.Ld >= (1ll << 31) || d.Npad / 32 > 65535 || d.N > 65535) return 0;
  return need;
}

// Split-K parts of one layer (same operands; consecutive depth-tap ranges; one output slab each) in ONE launch of
// the big-tile kernel.  REHR_OK launched; REHR_ENOSUP not applicable.
int wino_conv_split_try(const rehr_gather_gemm_desc* ds, int count, hipStream_t stream) {
  if (count < 2 || count > 8) return REHR_ENOSUP;
  const rehr_gather_gemm_desc& d0 = ds[0];
  if (!big_ok(d0) || (int64_t)d0.N * count > 65535) return REHR_ENOSUP;
  WinoParams p;
  p.d = d0;
  p.kchunks = (d0.Cin + 31) / 32;
  p.dh0 = -1;
  p.dw0 = -1;
  p.up = nullptr;
  p.up_bytes = 0;
  p.nsplit = count;
  p.band_major = (d0.debug_flags & REHR_DBG_GG_SLICE_MAJOR) ? 0 : 1;
  for (int i = 0; i < count; ++i) {
    const rehr_gather_gemm_desc& d = ds[i];
    if (!d.wino_ws || ((uintptr_t)d.wino_ws & 15)) return REHR_ENOSUP;
    const int64_t need = wino_workspace_bytes(d);
    if (need == 0 || d.wino_ws_bytes < need) return REHR_ENOSUP;
    // the parts differ in the depth taps and the destination only; no epilogue work (the combine has it)
    if (d.bias || d.stats_mode || d.act != REHR_ACT_NONE || d.x2 != d0.x2 || d.Cin != d0.Cin || d.c1 != d0.c1 ||
        d.ldx1 != d0.ldx1 || d.ldx2 != d0.ldx2 || d.Cout != d0.Cout || d.ldy != d0.ldy || d.Ld != d0.Ld ||
        d.Lh != d0.Lh || d.Lw != d0.Lw || d.Di != d0.Di || d.Hi != d0.Hi || d.Wi != d0.Wi || d.bd != d0.bd ||
        d.bh != d0.bh || d.bw != d0.bw || d.KH != d0.KH || d.KW != d0.KW || d.th.count != d0.th.count ||
        d.th.off0 != d0.th.off0 || d.th.offs != d0.th.offs || d.th.k0 != d0.th.k0 || d.th.ks != d0.th.ks ||
        d.tw.count != d0.tw.count || d.tw.off0 != d0.tw.off0 || d.tw.offs != d0.tw.offs || d.tw.k0 != d0.tw.k0 ||
        d.tw.ks != d0.tw.ks)
      return REHR_ENOSUP;
    p.s_td[i] = d.td;
    p.s_up[i] = d.wino_ws;
    p.s_up_bytes[i] = (uint32_t)need;
    p.s_y[i] = d.y;
  }
  for (int i = 0; i < count; ++i) {
    const int rc = wino_weights_frag_launch(ds[i], p.kchunks, stream);
    if (rc != REHR_OK) return rc;
  }
  if (d0.flags & REHR_GG_WS_ONLY) return REHR_OK;
  p.nb_h = (d0.Lh + 15) / 16;
  p.nb_w = (d0.Lw + 15) / 16;
  const size_t smem_x = (size_t)2 * BUF2 * sizeof(float), smem_e = (size_t)4 * 4 * 2 * 16 * 64 * sizeof(float);
  const size_t smem = smem_x > smem_e ? smem_x : smem_e;
  if (hipFuncSetAttribute((const void*)wino_conv_big8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
      hipSuccess)
    return REHR_EHIP;
  dim3 grid((unsigned)((int64_t)p.nb_h * p.nb_w * d0.Ld), d0.Npad / 64, d0.N * count);
  hipLaunchKernelGGL(wino_conv_big8_kernel, grid, dim3(512), smem, stream, p);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

// REHR_OK launched; REHR_ENOSUP not applicable.
int wino_conv_try(const rehr_gather_gemm_desc& d, hipStream_t stream) {
  if (!d.wino_ws) return REHR_ENOSUP;
  const int64_t need = wino_workspace_bytes(d);
  if (need == 0 || d.wino_ws_bytes < need || ((uintptr_t)d.wino_ws & 15)) return REHR_ENOSUP;
  WinoParams p;
  p.d = d;
  p.kchunks = (d.Cin + 31) / 32;
  p.dh0 = -1;
  p.dw0 = -1;
  p.up = d.wino_ws;
  p.up_bytes = (uint32_t)need;
  p.nsplit = 0;
  p.band_major = (d.debug_flags & REHR_DBG_GG_SLICE_MAJOR) ? 0 : 1;
  if (w32_ok(d)) {
    const int wrc = wino_weights_frag_launch(d, p.kchunks, stream);
    if (wrc != REHR_OK || (d.flags & REHR_GG_WS_ONLY)) return wrc;
    const int nfm = w32_groups(d);
    p.nb_h = (d.Lh + 8 * nfm - 1) / (8 * nfm);
    p.nb_w = (d.Lw + 15) / 16;
    // few K items per block (<= 32 input channels): two half-size blocks per CU hide each other's turnover (2 %)
    const bool two = (d.debug_flags & REHR_DBG_GG_W32P_TWO_PER_CU) ? true
                     : (d.debug_flags & REHR_DBG_GG_W32P_ONE_PER_CU) ? false : d.Cin * d.td.count <= 96;
    const int g = (nfm == 4 && !two) ? 2 : 1;
    p.nb_h = (d.Lh + 16 * g - 1) / (16 * g);
    const int rc = g == 2 ? launch_w32p<2>(p, d, stream) : launch_w32p<1>(p, d, stream);
    if (rc != REHR_OK) return rc;
    REHR_LAUNCH_CHECK();
    return REHR_OK;
  }
  if (big_ok(d)) {
    const int wrc = wino_weights_frag_launch(d, p.kchunks, stream);
    if (wrc != REHR_OK || (d.flags & REHR_GG_WS_ONLY)) return wrc;
    p.nb_h = (d.Lh + 15) / 16;
    p.nb_w = (d.Lw + 15) / 16;
    const size_t smem_x = (size_t)2 * BUF2 * sizeof(float), smem_e = (size_t)4 * 4 * 2 * 16 * 64 * sizeof(float);
    const size_t smem = smem_x > smem_e ? smem_x : smem_e;
    dim3 grid((unsigned)((int64_t)p.nb_h * p.nb_w * d.Ld), d.Npad / 64, d.N);
    static bool attr_set8 = false;
    if (!attr_set8) {
      if (hipFuncSetAttribute((const void*)wino_conv_big8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem) != hipSuccess)
        return REHR_EHIP;
      attr_set8 = true;
    }
    hipLaunchKernelGGL(wino_conv_big8_kernel, grid, dim3(512), smem, stream, p);
    REHR_LAUNCH_CHECK();
    return REHR_OK;
  }
  const int64_t nb_h = (d.Lh + 2 * TH - 1) / (2 * TH), nb_w = (d.Lw + 2 * TW - 1) / (2 * TW);

  // weight transform (reads the packed panel, writes the workspace)
  if (!(d.flags & REHR_GG_WS_READY)) {
    const int64_t total = (int64_t)d.td.count * d.Npad * d.Cin;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d, d.wino_ws);
    REHR_LAUNCH_CHECK();
  }
  if (d.flags & REHR_GG_WS_ONLY) return REHR_OK;
  p.nb_h = (int)nb_h;
  p.nb_w = (int)nb_w;
  const size_t smem_x = (size_t)2 * PVOX * LDX * sizeof(float), smem_e = (size_t)4 * 2 * 16 * 64 * sizeof(float);
  const size_t smem = smem_x > smem_e ? smem_x : smem_e;
  dim3 grid((unsigned)(nb_h * nb_w * d.Ld), d.Npad / 32, d.N);
  hipLaunchKernelGGL(wino_conv_kernel, grid, dim3(256), smem, stream, p);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
