// Halo-tile convolution on the fp32 matrix cores: the narrow-N (Cout <= 64), stride-1 member
// of the gather-GEMM family (same descriptor, same results).
//
// gather_gemm.hip re-gathers the A tile from L2 for every tap.  With only 32 or 64 output
// channels each gathered byte feeds few MFMAs, and the 128^3 layers of both networks (FLAVR
// layer1 / decoder.3, every 32/64-channel nnU-Net stage, sr_head.0) end up bound by L2 -> LDS
// traffic and per-tap barriers instead of the matrix pipe.  Here a block stages the 3-D input
// brick WITH ITS HALO in LDS once per 32-channel chunk and all taps read their A fragments
// from it at (voxel row + tap offset): no per-tap gather, no per-tap barrier.  B fragments
// (weights) go straight from L1/L2 into registers one tap ahead.  Blocks are persistent over a
// range of tiles and fetch the next (tile, chunk) halo into registers during the current
// sweep, so HBM/L2 latency hides behind ~28k-55k cycles of MFMA work.
#include "common.h"
#include "halo_conv.h"

namespace {

constexpr int BD = 2, BH = 8, BW = 8, BVOX = BD * BH * BW;  // lattice brick = 128 rows
constexpr int LDX = 36;                                     // LDS row stride (32 + 4)
constexpr int MAXX = 13;                                    // halo rows staged per thread (32 rows per pass)

struct HaloParams {
  rehr_gather_gemm_desc d;
  int HD, HH, HW, hvox;
  int mind, minh, minw;
  int nb_d, nb_h, nb_w, tiles_per_img;
  int64_t ntiles;       // N * tiles_per_img
  int tiles_per_block;
  int kchunks;
  uint32_t wp_bytes;
};

// WGM x WGN waves; wave tile (BVOX / WGM) x (BN / WGN)
template <int BN, int WGM, int WGN>
__global__ __launch_bounds__(256, 2) void halo_conv_kernel(const HaloParams p) {
  constexpr int WTM = BVOX / WGM, WTN = BN / WGN;
  constexpr int FM = WTM / 32, FN = WTN / 32;
  static_assert(WGM * WGN == 4 && FM >= 1 && FN >= 1, "4 waves");
  const rehr_gather_gemm_desc& d = p.d;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                           // [hvox][LDX]
  int* row_out = (int*)(smem + p.hvox * LDX);  // [BVOX]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int n0 = blockIdx.y * BN;
  const int q = tid & 7, r0 = tid >> 3;

  // halo rows owned by this thread (brick independent)
  int hcoord[MAXX];
#pragma unroll
  for (int i = 0; i < MAXX; ++i) {
    const int hv = r0 + 32 * i;
    const int hw_ = hv % p.HW;
    const int t2 = hv / p.HW;
    hcoord[i] = hv < p.hvox ? (((t2 / p.HH) << 20) | ((t2 % p.HH) << 10) | hw_) : -1;
  }
  const uint32_t img_elems = (uint32_t)d.Di * d.Hi * d.Wi;

  // A-fragment rows of this lane (brick independent): voxel r -> halo row at tap origin
  int arow[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int r = wm * WTM + i * 32 + (lane & 31);
    arow[i] = (((r >> 6) * p.HH + ((r >> 3) & 7)) * p.HW + (r & 7)) * LDX + 4 * (lane >> 5);
  }

  f32x16 acc[FM][FN];
  f32x4 rx[MAXX];

  const int64_t t_begin = (int64_t)blockIdx.x * p.tiles_per_block;
  int64_t t_end = t_begin + p.tiles_per_block;
  if (t_end > p.ntiles) t_end = p.ntiles;
  const int64_t items = (t_end > t_begin ? t_end - t_begin : 0) * p.kchunks;

  // fetch the halo of work item `it` = (tile, chunk) into registers (branch-free)
  auto fetch = [&](int64_t it) {
    const bool live = it < items;
    const int64_t ii = live ? it : 0;
    const int64_t tile = t_begin + ii / p.kchunks;
    const int cc = (int)(ii % p.kchunks) * 32;
    const int n = (int)(tile / p.tiles_per_img);
    int tr = (int)(tile - (int64_t)n * p.tiles_per_img);
    const int bw_ = tr % p.nb_w; tr /= p.nb_w;
    const int bh_ = tr % p.nb_h;
    const int bd_ = tr / p.nb_h;
    const int gd0 = bd_ * BD + p.mind, gh0 = bh_ * BH + p.minh, gw0 = bw_ * BW + p.minw;
    const bool first = cc < d.c1;
    const float* src = first ? d.x1 : d.x2;
    const uint32_t ld = (uint32_t)(first ? d.ldx1 : d.ldx2);
    const int coff = (first ? cc : cc - d.c1) + q * 4;
    const uint32_t nrec = img_elems * ld * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(src) + (int64_t)n * img_elems * ld, 0, nrec, 0x00020000);
    const bool kok = (cc + q * 4) < d.Cin;
#pragma unroll
    for (int i = 0; i < MAXX; ++i) {
      const int hc = hcoord[i];
      const int id = gd0 + (hc >> 20), ih = gh0 + ((hc >> 10) & 1023), iw = gw0 + (hc & 1023);
      const bool ok = live & kok & (hc >= 0) & ((unsigned)id < (unsigned)d.Di) & ((unsigned)ih < (unsigned)d.Hi) &
                      ((unsigned)iw < (unsigned)d.Wi);
      const uint32_t off = (uint32_t)((id * d.Hi + ih) * d.Wi + iw) * ld * 4u + (uint32_t)coff * 4u;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : nrec, 0, 0));
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < MAXX; ++i)
      if (hcoord[i] >= 0) *reinterpret_cast<f32x4*>(Xs + (r0 + 32 * i) * LDX + q * 4) = rx[i];
  };

  // weights: lane's B fragment of tap wt, chunk cc, k-group kk: wp[wt][n][cc + kk*8 + 4*half .. +3]
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.wp), 0, p.wp_bytes, 0x00020000);
  const uint32_t wlane = ((uint32_t)(n0 + wn * WTN + (lane & 31)) * d.Cin + 4u * (lane >> 5)) * 4u;
  auto load_b = [&](int wt, int cc, f32x4 (&rb)[FN][4]) {
    const uint32_t base = ((uint32_t)wt * d.Npad * d.Cin + cc) * 4u + wlane;
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        rb[j][kk] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, base + (uint32_t)(j * 32) * d.Cin * 4u + kk * 32u, 0, 0));
  };

  const int ntaps = d.td.count * d.th.count * d.tw.count;
  // tap iterator: (jd, jh, jw) counters advanced without divisions (all wave-uniform scalars)
  int jd = 0, jh = 0, jw = 0;
  auto tap_geom = [&](int& tapoff, int& wt) {
    const int od_ = d.bd + d.td.off0 + d.td.offs * jd - p.mind;
    const int oh_ = d.bh + d.th.off0 + d.th.offs * jh - p.minh;
    const int ow_ = d.bw + d.tw.off0 + d.tw.offs * jw - p.minw;
    tapoff = ((od_ * p.HH + oh_) * p.HW + ow_) * LDX;
    wt = ((d.td.k0 + d.td.ks * jd) * d.KH + (d.th.k0 + d.th.ks * jh)) * d.KW + (d.tw.k0 + d.tw.ks * jw);
    // advance; after the last tap the counters wrap to tap 0 (that prefetch is never used)
    ++jw;
    const bool cw = jw >= d.tw.count;
    jw = cw ? 0 : jw;
    jh += cw ? 1 : 0;
    const bool ch = jh >= d.th.count;
    jh = ch ? 0 : jh;
    jd += ch ? 1 : 0;
    jd = jd >= d.td.count ? 0 : jd;
  };
  auto mfma_tap = [&](int tapoff, const f32x4 (&rb)[FN][4], int kgroups) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk >= kgroups) break;  // half-filled last chunk (Cin % 32 == 16): skip the zero k-groups
      f32x4 fa[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Xs + arow[i] + tapoff + kk * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], rb[j][kk][e], acc[i][j], 0, 0, 0);
    }
  };

  if (items > 0) {
    fetch(0);
    stage();
  }
  __syncthreads();

  for (int64_t it = 0; it < items; ++it) {
    const int64_t tile = t_begin + it / p.kchunks;
    const int chunk = (int)(it % p.kchunks);
    const int cc = chunk * 32;
    if (chunk == 0) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    fetch(it + 1);  // next halo: in flight during this sweep
    const int kgroups = (d.Cin - cc) >= 32 ? 4 : (d.Cin - cc) / 8;

    // taps, weights one tap ahead (two named register sets)
    f32x4 rb0[FN][4], rb1[FN][4];
    int off0, wt0, off1, wt1;
    jd = jh = jw = 0;
    tap_geom(off0, wt0);
    load_b(wt0, cc, rb0);
    for (int t = 0; t < ntaps; t += 2) {
      tap_geom(off1, wt1);          // tap t+1 (wraps harmlessly past the end)
      load_b(wt1, cc, rb1);
      mfma_tap(off0, rb0, kgroups);
      if (t + 1 >= ntaps) break;
      tap_geom(off0, wt0);          // tap t+2
      load_b(wt0, cc, rb0);
      mfma_tap(off1, rb1, kgroups);
    }

    if (chunk == p.kchunks - 1) {
      // ---- epilogue of this tile
      const int n_img = (int)(tile / p.tiles_per_img);
      int tr = (int)(tile - (int64_t)n_img * p.tiles_per_img);
      const int bw_ = tr % p.nb_w; tr /= p.nb_w;
      const int bh_ = tr % p.nb_h;
      const int bd_ = tr / p.nb_h;
      if (tid < BVOX) {
        const int od = bd_ * BD + (tid >> 6), oh = bh_ * BH + ((tid >> 3) & 7), ow = bw_ * BW + (tid & 7);
        int off = -1;
        if (od < d.Ld && oh < d.Lh && ow < d.Lw)
          off = ((n_img * d.Dy + od * d.osd + d.obd) * d.Hy + oh * d.osh + d.obh) * d.Wy + ow * d.osw + d.obw;
        row_out[tid] = off;
      }
      __syncthreads();
      const int chalf = lane >> 5;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = n0 + wn * WTN + j * 32 + (lane & 31);
        const bool colok = col < d.Cout;
        const float bv = (d.bias != nullptr && colok) ? d.bias[col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
            const int off = row_out[row];
            const float v = apply_act(acc[i][j][r] + bv, d.act, d.slope);
            if (off >= 0 && colok) {
              d.y[(int64_t)off * d.ldy + col] = v;
              s1 += v;
              s2 += v * v;
            }
          }
        }
        if (d.stats_mode != 0) {
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 32, 64);
          if (chalf == 0 && colok) {
            double* st = d.stats + ((int64_t)n_img * d.Cout + col) * 2;
            atomicAdd(st, (double)s1);
            if (d.stats_mode == 2) atomicAdd(st + 1, (double)s2);
          }
        }
      }
    }
    __syncthreads();  // every wave is done reading this halo (and row_out)
    stage();          // next halo: registers -> LDS
    __syncthreads();
  }
}

void span(const rehr_axis_taps& t, int b, int* mn, int* mx) {
  int lo = b + t.off0, hi = lo;
  for (int j = 1; j < t.count; ++j) {
    const int o = b + t.off0 + t.offs * j;
    if (o < lo) lo = o;
    if (o > hi) hi = o;
  }
  *mn = lo;
  *mx = hi;
}

template <int BN, int WGM, int WGN>
int launch(const HaloParams& p, size_t smem, hipStream_t stream) {
  auto kern = halo_conv_kernel<BN, WGM, WGN>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            96 * 1024) != hipSuccess)
      return REHR_EHIP;
    attr_set = true;
  }
  const int64_t blocks_x = (p.ntiles + p.tiles_per_block - 1) / p.tiles_per_block;
  dim3 grid((unsigned)blocks_x, p.d.Npad / BN, 1);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, p);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

}  // namespace

// REHR_ENOSUP = "not this kernel's case" (the caller then uses the generic gather-GEMM).
int halo_conv_try(const rehr_gather_gemm_desc& d, hipStream_t stream) {
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return REHR_ENOSUP;
  if (d.Npad > 64) return REHR_ENOSUP;
  const int T = d.td.count * d.th.count * d.tw.count;
  if (T < 9 || T > 64) return REHR_ENOSUP;   // few taps: the gather kernel is already cheap per byte
  if (d.Ld < BD || d.Lh < BH || d.Lw < BW) return REHR_ENOSUP;
  const int64_t nb_d = (d.Ld + BD - 1) / BD, nb_h = (d.Lh + BH - 1) / BH, nb_w = (d.Lw + BW - 1) / BW;
  if (nb_d * BD * nb_h * BH * nb_w * BW * 10 > (int64_t)d.Ld * d.Lh * d.Lw * 13) return REHR_ENOSUP;
  int mn[3], mx[3];
  span(d.td, d.bd, &mn[0], &mx[0]);
  span(d.th, d.bh, &mn[1], &mx[1]);
  span(d.tw, d.bw, &mn[2], &mx[2]);
  HaloParams p;
  p.d = d;
  p.HD = BD + mx[0] - mn[0]; p.HH = BH + mx[1] - mn[1]; p.HW = BW + mx[2] - mn[2];
  p.hvox = p.HD * p.HH * p.HW;
  if (p.hvox > MAXX * 32 || p.HH > 1023 || p.HW > 1023) return REHR_ENOSUP;
  const size_t smem = (size_t)p.hvox * LDX * sizeof(float) + BVOX * sizeof(int);
  if (smem > 80 * 1024) return REHR_ENOSUP;
  const int64_t img = (int64_t)d.Di * d.Hi * d.Wi * 4;
  if (img * d.ldx1 >= (1ll << 32) - 64 || (d.x2 && img * d.ldx2 >= (1ll << 32) - 64)) return REHR_ENOSUP;
  if ((int64_t)d.N * d.Dy * d.Hy * d.Wy >= (1ll << 31)) return REHR_ENOSUP;
  p.mind = mn[0]; p.minh = mn[1]; p.minw = mn[2];
  p.nb_d = (int)nb_d; p.nb_h = (int)nb_h; p.nb_w = (int)nb_w;
  p.tiles_per_img = (int)(nb_d * nb_h * nb_w);
  p.ntiles = (int64_t)d.N * p.tiles_per_img;
  p.kchunks = (d.Cin + 31) / 32;
  {
    const int64_t kd_max = d.td.k0 + (int64_t)d.td.ks * (d.td.count - 1);
    const int64_t kh_max = d.th.k0 + (int64_t)d.th.ks * (d.th.count - 1);
    const int64_t kw_max = d.tw.k0 + (int64_t)d.tw.ks * (d.tw.count - 1);
    const int64_t wb = (((kd_max * d.KH) + kh_max) * d.KW + kw_max + 1) * d.Npad * d.Cin * 4;
    if (wb >= (1ll << 32) - 64) return REHR_ENOSUP;
    p.wp_bytes = (uint32_t)wb;
  }
  // persistent blocks: 512 resident (256 CUs x 2); full rounds, >= 2 tiles each when possible
  const int n_tiles = d.Npad / (d.Npad % 64 == 0 ? 64 : 32);
  int64_t want = 1024 / n_tiles;
  if (want > p.ntiles) want = p.ntiles;
  if (want < 1) want = 1;
  p.tiles_per_block = (int)((p.ntiles + want - 1) / want);
  if (d.Npad % 64 == 0) return launch<64, 2, 2>(p, smem, stream);
  return launch<32, 4, 1>(p, smem, stream);
}
