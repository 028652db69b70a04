// Halo-brick convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the unit-stride
// 3x3x3 / 1x3x3 member of the mixed-precision gather-GEMM family (same descriptor, same results as
// gather_gemm_bf16.hip) -- forward AND input gradient (flipped taps) of every nnU-Net stage conv and FLAVR block.
//
// At 16x the fp32 rate the matrix pipe outruns everything that re-gathers operands per tap: gather_gemm_bf16
// reads each input voxel 27 times through L1/L2 and reaches ~0.2 PFLOP/s on the 32/64-channel layers at 128^3-160^3,
// which hold 60 % of the SegModel's FLOPs.  Here ("LDS-staged 3-D input tiles with halo", BASELINE north_star):
//   * a block owns a brick of 128-512 output voxels x BN output channels; the input brick WITH ITS HALO for one
//     32-channel chunk is staged in LDS once (64-byte rows + 16 pad) and ALL taps read their A fragments from it at
//     (voxel row + tap offset) -- no per-tap gather, no per-tap barrier;
//   * a wave owns 128 voxels x 32 output channels (4 x 1 accumulator tiles): every weight fragment (one 16-byte
//     L1/L2 load per lane, one tap ahead, two register sets) feeds four MFMAs, so the weight stream stays at
//     ~32 B/clk/CU, and the LDS sees one ds_read_b128 per MFMA;
//   * blocks are persistent over a range of (brick, chunk) items and fetch the next halo into registers during
//     the current sweep of 27 taps x 2 k-steps x 4 tiles = 216 MFMAs per wave.
// Epilogue as in the gather kernel: bias, ReLU / LeakyReLU, bf16 (or fp32) store, fp64 statistics from the fp32
// accumulators.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;            // channels per chunk
constexpr int ROWB = BK * 2 + 16; // LDS row stride in bytes
constexpr int TPR = BK / 8;       // threads per row (16 bytes each)
constexpr int RPP = 256 / TPR;    // rows per staging pass
constexpr int FM = 4;             // accumulator tiles per wave along the voxels

struct HBParams {
  rehr_gather_gemm_desc d;
  int HD, HH, HW, hvox;
  int mind, minh, minw;
  int nb_d, nb_h, nb_w, tiles_per_img;
  int64_t ntiles;
  int tiles_per_block;
  int kchunks;
  uint32_t wp_bytes;
};

// brick BD x BH x BW (powers of two, BD*BH*BW = 128 * WGM), WGM x WGN waves, BN = 32 * WGN
template <int WGM, int WGN, int BD, int BH, int BW>
__global__ __launch_bounds__(256, (WGM == 4 ? 1 : 2)) void halo_conv_bf16_kernel(const HBParams p) {
  constexpr int BVOX = BD * BH * BW;
  constexpr int BN = 32 * WGN;
  constexpr int MAXX = ((BD + 2) * (BH + 2) * (BW + 2) + RPP - 1) / RPP;
  static_assert(WGM * WGN == 4 && BVOX == 128 * WGM, "wave = 128 voxels x 32 channels");
  const rehr_gather_gemm_desc& d = p.d;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* Xs = smem_b;                           // [hvox][ROWB]
  int* row_out = (int*)(smem_b + p.hvox * ROWB);        // [BVOX]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int n0 = blockIdx.y * BN;
  const int q = tid % TPR, r0 = tid / TPR;

  int hcoord[MAXX];
#pragma unroll
  for (int i = 0; i < MAXX; ++i) {
    const int hv = r0 + RPP * i;
    const int hw_ = hv % p.HW;
    const int t2 = hv / p.HW;
    hcoord[i] = hv < p.hvox ? (((t2 / p.HH) << 20) | ((t2 % p.HH) << 10) | hw_) : -1;
  }
  const uint32_t img_elems = (uint32_t)d.Di * d.Hi * d.Wi;

  // A-fragment byte offsets of this lane (brick independent): voxel r -> its halo row at tap origin
  int arow[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int r = wm * 128 + i * 32 + (lane & 31);
    const int rd = r / (BH * BW), rh = (r / BW) % BH, rw = r % BW;
    arow[i] = ((rd * p.HH + rh) * p.HW + rw) * ROWB + 16 * (lane >> 5);
  }

  f32x16 acc[FM];
  u32x4 rx[MAXX];

  const int64_t t_begin = (int64_t)blockIdx.x * p.tiles_per_block;
  int64_t t_end = t_begin + p.tiles_per_block;
  if (t_end > p.ntiles) t_end = p.ntiles;
  const int64_t items = (t_end > t_begin ? t_end - t_begin : 0) * p.kchunks;

  auto fetch = [&](int64_t it) {
    const bool live = it < items;
    const int64_t ii = live ? it : 0;
    const int64_t tile = t_begin + ii / p.kchunks;
    const int cc = (int)(ii % p.kchunks) * BK;
    const int n = (int)(tile / p.tiles_per_img);
    int tr = (int)(tile - (int64_t)n * p.tiles_per_img);
    const int bw_ = tr % p.nb_w; tr /= p.nb_w;
    const int bh_ = tr % p.nb_h;
    const int bd_ = tr / p.nb_h;
    const int gd0 = bd_ * BD + p.mind, gh0 = bh_ * BH + p.minh, gw0 = bw_ * BW + p.minw;
    const bool first = cc < d.c1;
    const __bf16* src = reinterpret_cast<const __bf16*>(first ? d.x1 : d.x2);
    const uint32_t ld = (uint32_t)(first ? d.ldx1 : d.ldx2);
    const int coff = (first ? cc : cc - d.c1) + q * 8;
    const uint32_t nrec = img_elems * ld * 2u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(src) + (int64_t)n * img_elems * ld, 0, nrec, 0x00020000);
    const bool kok = (cc + q * 8) < d.Cin;
#pragma unroll
    for (int i = 0; i < MAXX; ++i) {
      const int hc = hcoord[i];
      const int id = gd0 + (hc >> 20), ih = gh0 + ((hc >> 10) & 1023), iw = gw0 + (hc & 1023);
      const bool ok = live & kok & (hc >= 0) & ((unsigned)id < (unsigned)d.Di) & ((unsigned)ih < (unsigned)d.Hi) &
                      ((unsigned)iw < (unsigned)d.Wi);
      const uint32_t off = (uint32_t)((id * d.Hi + ih) * d.Wi + iw) * ld * 2u + (uint32_t)coff * 2u;
      rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : nrec, 0, 0);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < MAXX; ++i)
      if (hcoord[i] >= 0) *reinterpret_cast<u32x4*>(Xs + (r0 + RPP * i) * ROWB + q * 16) = rx[i];
  };

  // weights: lane's B fragments of tap wt, chunk cc: wp[wt][n][cc + 16*kk + 8*half .. +7], kk = 0, 1
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(reinterpret_cast<const __bf16*>(d.wp)), 0, p.wp_bytes, 0x00020000);
  const uint32_t wlane = ((uint32_t)(n0 + wn * 32 + (lane & 31)) * d.Cin + 8u * (lane >> 5)) * 2u;
  auto load_b = [&](int wt, int cc, u32x4 (&rb)[2], bool k1) {
    const uint32_t base = ((uint32_t)wt * d.Npad * d.Cin + cc) * 2u + wlane;
    rb[0] = __builtin_amdgcn_raw_buffer_load_b128(rsw, base, 0, 0);
    rb[1] = __builtin_amdgcn_raw_buffer_load_b128(rsw, k1 ? base + 32u : p.wp_bytes, 0, 0);
  };

  const int ntaps = d.td.count * d.th.count * d.tw.count;
  int jd = 0, jh = 0, jw = 0;
  auto tap_geom = [&](int& tapoff, int& wt) {
    const int od_ = d.bd + d.td.off0 + d.td.offs * jd - p.mind;
    const int oh_ = d.bh + d.th.off0 + d.th.offs * jh - p.minh;
    const int ow_ = d.bw + d.tw.off0 + d.tw.offs * jw - p.minw;
    tapoff = ((od_ * p.HH + oh_) * p.HW + ow_) * ROWB;
    wt = ((d.td.k0 + d.td.ks * jd) * d.KH + (d.th.k0 + d.th.ks * jh)) * d.KW + (d.tw.k0 + d.tw.ks * jw);
    ++jw;
    const bool cw = jw >= d.tw.count;
    jw = cw ? 0 : jw;
    jh += cw ? 1 : 0;
    const bool ch = jh >= d.th.count;
    jh = ch ? 0 : jh;
    jd += ch ? 1 : 0;
    jd = jd >= d.td.count ? 0 : jd;
  };
  auto mfma_tap = [&](int tapoff, const u32x4 (&rb)[2], bool k1) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk == 1 && !k1) break;   // half-filled last chunk (Cin % 32 == 16)
      bf16x8 fa[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(Xs + arow[i] + tapoff + kk * 32);
      const bf16x8 fb = __builtin_bit_cast(bf16x8, rb[kk]);
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb, acc[i], 0, 0, 0);
    }
  };

  if (items > 0) {
    fetch(0);
    stage();
  }
  __syncthreads();

  const bool y32 = (d.flags & REHR_GG_Y_F32) != 0;
  __bf16* yb = reinterpret_cast<__bf16*>(d.y);
  for (int64_t it = 0; it < items; ++it) {
    const int64_t tile = t_begin + it / p.kchunks;
    const int chunk = (int)(it % p.kchunks);
    const int cc = chunk * BK;
    if (chunk == 0) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }
    fetch(it + 1);
    const bool k1 = (d.Cin - cc) >= 32;

    u32x4 rb0[2], rb1[2];
    int off0, wt0, off1, wt1;
    jd = jh = jw = 0;
    tap_geom(off0, wt0);
    load_b(wt0, cc, rb0, k1);
    for (int t = 0; t < ntaps; t += 2) {
      tap_geom(off1, wt1);
      load_b(wt1, cc, rb1, k1);
      mfma_tap(off0, rb0, k1);
      if (t + 1 >= ntaps) break;
      tap_geom(off0, wt0);
      load_b(wt0, cc, rb0, k1);
      mfma_tap(off1, rb1, k1);
    }

    if (chunk == p.kchunks - 1) {
      const int n_img = (int)(tile / p.tiles_per_img);
      int tr = (int)(tile - (int64_t)n_img * p.tiles_per_img);
      const int bw_ = tr % p.nb_w; tr /= p.nb_w;
      const int bh_ = tr % p.nb_h;
      const int bd_ = tr / p.nb_h;
      for (int v = tid; v < BVOX; v += 256) {
        const int od = bd_ * BD + v / (BH * BW), oh = bh_ * BH + (v / BW) % BH, ow = bw_ * BW + v % BW;
        int off = -1;
        if (od < d.Ld && oh < d.Lh && ow < d.Lw)
          off = ((n_img * d.Dy + od * d.osd + d.obd) * d.Hy + oh * d.osh + d.obh) * d.Wy + ow * d.osw + d.obw;
        row_out[v] = off;
      }
      __syncthreads();
      const int chalf = lane >> 5;
      const int col = n0 + wn * 32 + (lane & 31);
      const bool colok = col < d.Cout;
      const float bv = (d.bias != nullptr && colok) ? d.bias[col] : 0.f;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
          const int off = row_out[row];
          const float v = apply_act(acc[i][r] + bv, d.act, d.slope);
          if (off >= 0 && colok) {
            if (y32) d.y[(int64_t)off * d.ldy + col] = v;
            else yb[(int64_t)off * d.ldy + col] = (__bf16)v;
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
    __syncthreads();
    stage();
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

template <int WGM, int WGN, int BD, int BH, int BW>
int launch(HBParams p, hipStream_t stream) {
  constexpr int BN = 32 * WGN;
  const rehr_gather_gemm_desc& d = p.d;
  const int64_t nb_d = (d.Ld + BD - 1) / BD, nb_h = (d.Lh + BH - 1) / BH, nb_w = (d.Lw + BW - 1) / BW;
  if (nb_d * BD * nb_h * BH * nb_w * BW * 10 > (int64_t)d.Ld * d.Lh * d.Lw * 13) return REHR_ENOSUP;  // <= 1.3x padding
  p.HD = BD + p.HD; p.HH = BH + p.HH; p.HW = BW + p.HW;      // (span extents were left in HD/HH/HW)
  p.hvox = p.HD * p.HH * p.HW;
  if (p.HD > BD + 2 || p.HH > BH + 2 || p.HW > BW + 2) return REHR_ENOSUP;
  const size_t smem = (size_t)p.hvox * ROWB + (size_t)BD * BH * BW * sizeof(int);
  p.nb_d = (int)nb_d; p.nb_h = (int)nb_h; p.nb_w = (int)nb_w;
  p.tiles_per_img = (int)(nb_d * nb_h * nb_w);
  p.ntiles = (int64_t)d.N * p.tiles_per_img;
  // persistent blocks: one (BN = 32) or two resident per CU; whole rounds of them
  const int n_tiles = d.Npad / BN;
  const int resident = 256 * (WGM == 4 ? 1 : 2);
  int64_t want = (2 * resident) / n_tiles;
  if (want > p.ntiles) want = p.ntiles;
  if (want < 1) want = 1;
  p.tiles_per_block = (int)((p.ntiles + want - 1) / want);
  auto kern = halo_conv_bf16_kernel<WGM, WGN, BD, BH, BW>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((size_t)(BD + 2) * (BH + 2) * (BW + 2) * ROWB + BD * BH * BW * sizeof(int))) != hipSuccess)
      return REHR_EHIP;
    attr_set = true;
  }
  const int64_t blocks_x = (p.ntiles + p.tiles_per_block - 1) / p.tiles_per_block;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks_x, n_tiles, 1), dim3(256), smem, stream, p);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

}  // namespace

// REHR_OK = launched; REHR_ENOSUP = not this kernel's case (the caller uses gather_gemm_bf16); other = error.
int halo_conv_bf16_try(const rehr_gather_gemm_desc& d, hipStream_t stream) {
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return REHR_ENOSUP;
  const int T = d.td.count * d.th.count * d.tw.count;
  if (T < 9 || T > 27) return REHR_ENOSUP;        // few taps: the gather kernel is already cheap per byte
  if (d.td.count > 3 || d.th.count > 3 || d.tw.count > 3) return REHR_ENOSUP;
  if (d.c1 < d.Cin && d.c1 % BK) return REHR_ENOSUP;
  int mn[3], mx[3];
  span(d.td, d.bd, &mn[0], &mx[0]);
  span(d.th, d.bh, &mn[1], &mx[1]);
  span(d.tw, d.bw, &mn[2], &mx[2]);
  HBParams p;
  p.d = d;
  p.HD = mx[0] - mn[0]; p.HH = mx[1] - mn[1]; p.HW = mx[2] - mn[2];   // halo extents, completed per brick in launch()
  p.mind = mn[0]; p.minh = mn[1]; p.minw = mn[2];
  const int64_t img = (int64_t)d.Di * d.Hi * d.Wi * 2;
  if (img * d.ldx1 >= (1ll << 32) - 64 || (d.x2 && img * d.ldx2 >= (1ll << 32) - 64)) return REHR_ENOSUP;
  if ((int64_t)d.N * d.Dy * d.Hy * d.Wy >= (1ll << 31)) return REHR_ENOSUP;
  p.kchunks = (d.Cin + BK - 1) / BK;
  {
    const int64_t kd_max = d.td.k0 + (int64_t)d.td.ks * (d.td.count - 1);
    const int64_t kh_max = d.th.k0 + (int64_t)d.th.ks * (d.th.count - 1);
    const int64_t kw_max = d.tw.k0 + (int64_t)d.tw.ks * (d.tw.count - 1);
    const int64_t wb = (((kd_max * d.KH) + kh_max) * d.KW + kw_max + 1) * d.Npad * d.Cin * 2;
    if (wb >= (1ll << 32) - 64) return REHR_ENOSUP;
    p.wp_bytes = (uint32_t)wb;
  }
  // wave = 128 voxels x 32 channels; the block's 4 waves tile (voxels x channels) by the layer's width.
  // Candidates in order of preference; a brick shape that pads the lattice by more than 1.3x declines.
  int rc = REHR_ENOSUP;
  if (d.Npad % 128 == 0) {
    if (d.Ld >= 2 && d.Lh >= 8 && d.Lw >= 8) rc = launch<1, 4, 2, 8, 8>(p, stream);
    return rc;
  }
  if (d.Npad % 64 == 0) {
    if (d.Ld >= 2 && d.Lh >= 8 && d.Lw >= 16) rc = launch<2, 2, 2, 8, 16>(p, stream);
    if (rc == REHR_ENOSUP && d.Ld >= 4 && d.Lh >= 8 && d.Lw >= 8) rc = launch<2, 2, 4, 8, 8>(p, stream);
    return rc;
  }
  if (d.Ld >= 4 && d.Lh >= 8 && d.Lw >= 16) rc = launch<4, 1, 4, 8, 16>(p, stream);
  if (rc == REHR_ENOSUP && d.Ld >= 2 && d.Lh >= 16 && d.Lw >= 16) rc = launch<4, 1, 2, 16, 16>(p, stream);
  return rc;
}
