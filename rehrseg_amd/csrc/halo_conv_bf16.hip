// Halo-brick convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the unit-stride
// 3x3x3 / 1x3x3 member of the mixed-precision gather-GEMM family (same descriptor, same results as
// gather_gemm_bf16.hip) -- forward AND input gradient (flipped taps) of every nnU-Net stage conv and FLAVR block,
// and the stride phases of the (3,4,4)/(1,2,2) transposed convolutions.
//
// At 16x the fp32 rate the matrix pipe outruns everything that re-gathers operands per tap: gather_gemm_bf16
// reads each input voxel 27 times through L1/L2 and reaches ~0.2 PFLOP/s on the 32/64-channel layers at
// 128^3-160^3, which hold 60 % of the SegModel's FLOPs.  Here ("LDS-staged 3-D input tiles with halo", BASELINE
// north_star) BOTH operands of a 32-channel chunk live in LDS:
//   * a block (4 waves, one per SIMD, the whole register file and 144 KB of LDS: one block per CU) owns a brick of
//     512 output voxels x 32 output channels; the input brick WITH ITS HALO (<= 6x10x18 voxels x 64 bytes, rows
//     padded to 80 bytes) is staged once per chunk and every tap reads its fragments at (voxel row + tap offset);
//   * the weights of ALL taps of the chunk (27 x 32 x 32 bf16 = 54 KB, 16-byte pieces XOR-swizzled by the row) are
//     staged once per chunk -- once per BLOCK for 32-channel layers, whose persistent blocks keep them for every
//     brick.  (The first version fetched weight fragments from L1/L2 one tap ahead: with one wave per SIMD every
//     tap paid the L2 latency, 60k cycles per brick against 6.9k cycles of MFMA.)
//   * a wave owns 128 voxels x 32 channels: per 16-channel k step 4 activation reads + 1 weight read
//     (ds_read_b128, conflict-free) feed 4 MFMAs; reads run one MFMA group ahead in two register sets;
//   * the MFMA takes the WEIGHTS as its A operand: the accumulator then holds a voxel per lane and 4 consecutive
//     channels per register quad, so the epilogue stores 8 bytes per lane (16 for fp32 output) instead of single
//     bf16 values; the per-(sample, channel) statistics stay in registers across the bricks of a sample and are
//     reduced across lanes once per sample.
// Epilogue semantics as in the gather kernel: bias, ReLU / LeakyReLU, bf16 (or fp32) store, fp64 statistics formed
// from the fp32 values.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;              // channels per chunk
constexpr int XROW = BK * 2 + 16;   // halo row stride in bytes (padded)
constexpr int WROW = BK * 2;        // weight row (one output channel of one tap) in bytes, pieces swizzled
constexpr int TPR = 4;              // threads per halo row (16 bytes each)
constexpr int RPP = 256 / TPR;      // halo rows per staging pass
constexpr int FM = 4;               // 32-voxel tiles per wave
constexpr int BVOX = 512;
constexpr int BN = 32;
constexpr int MAXTAPS = 27;

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

template <int BD, int BH, int BW>
__global__ __launch_bounds__(256, 1) void halo_conv_bf16_kernel(const HBParams p) {
  static_assert(BD * BH * BW == BVOX, "512 voxels");
  constexpr int MAXX = ((BD + 2) * (BH + 2) * (BW + 2) + RPP - 1) / RPP;
  const rehr_gather_gemm_desc& d = p.d;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* Xs = smem_b;                                   // [hvox][XROW]
  unsigned char* Ws = smem_b + p.hvox * XROW;                   // [ntaps][32][WROW]
  const int ntaps = d.td.count * d.th.count * d.tw.count;
  int* row_out = (int*)(Ws + ntaps * BN * WROW);                // [BVOX]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
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

  // activation-fragment byte offsets of this lane (brick independent): voxel r -> its halo row at the tap origin
  int arow[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int r = wave * 128 + i * 32 + (lane & 31);
    const int rd = r / (BH * BW), rh = (r / BW) % BH, rw = r % BW;
    arow[i] = ((rd * p.HH + rh) * p.HW + rw) * XROW + 16 * half;
  }
  // weight-fragment byte offsets: row = output channel (lane & 31), 16-byte piece c = 2*kk + half, swizzled by the row
  const int wco = lane & 31;
  int wrow[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) wrow[kk] = wco * WROW + (((2 * kk + half) ^ ((wco >> 2) & 3)) << 4);

  f32x16 acc[FM];
  u32x4 rx[MAXX];
  float s1[16], s2[16];   // per-lane partial statistics of channels n0 + 8g + 4*half + e (index 4g + e)
#pragma unroll
  for (int k = 0; k < 16; ++k) s1[k] = s2[k] = 0.f;
  int stats_n = -1;

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
      if (hcoord[i] >= 0) *reinterpret_cast<u32x4*>(Xs + (r0 + RPP * i) * XROW + q * 16) = rx[i];
  };

  // all taps' weights of chunk cc -> LDS: piece (t, co, c) = wp[wt(t)][n0 + co][cc + 8c .. +7]
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(reinterpret_cast<const __bf16*>(d.wp)), 0, p.wp_bytes, 0x00020000);
  auto stage_weights = [&](int cc) {
    const int npieces = ntaps * BN * 4;
    const int thw = d.th.count * d.tw.count;
    for (int base = 0; base < npieces; base += 256 * 4) {
      u32x4 v[4];
      int dst[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pc = base + u * 256 + tid;
        const int t = pc >> 7, co = (pc >> 2) & 31, c = pc & 3;
        const int jd = t / thw, jr = t - jd * thw, jh = jr / d.tw.count, jw = jr - jh * d.tw.count;
        const int wt = ((d.td.k0 + d.td.ks * jd) * d.KH + (d.th.k0 + d.th.ks * jh)) * d.KW + (d.tw.k0 + d.tw.ks * jw);
        const bool ok = (pc < npieces) & ((cc + 8 * c) < d.Cin);
        const uint32_t off = (((uint32_t)wt * d.Npad + n0 + co) * d.Cin + cc + 8 * c) * 2u;
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsw, ok ? off : p.wp_bytes, 0, 0);
        dst[u] = pc < npieces ? (t * BN + co) * WROW + ((c ^ ((co >> 2) & 3)) << 4) : -1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (dst[u] >= 0) *reinterpret_cast<u32x4*>(Ws + dst[u]) = v[u];
    }
  };

  // tap iterator (wave-uniform scalars): halo byte offset of the tap
  int jd = 0, jh = 0, jw = 0;
  auto tap_off = [&]() -> int {
    const int od_ = d.bd + d.td.off0 + d.td.offs * jd - p.mind;
    const int oh_ = d.bh + d.th.off0 + d.th.offs * jh - p.minh;
    const int ow_ = d.bw + d.tw.off0 + d.tw.offs * jw - p.minw;
    const int off = ((od_ * p.HH + oh_) * p.HW + ow_) * XROW;
    ++jw;
    const bool cw = jw >= d.tw.count;
    jw = cw ? 0 : jw;
    jh += cw ? 1 : 0;
    const bool ch = jh >= d.th.count;
    jh = ch ? 0 : jh;
    jd += ch ? 1 : 0;
    jd = jd >= d.td.count ? 0 : jd;
    return off;
  };
  auto read_frags = [&](int xoff, int woff, int kk, bf16x8 (&fx)[FM], bf16x8& fw) {
#pragma unroll
    for (int i = 0; i < FM; ++i) fx[i] = *reinterpret_cast<const bf16x8*>(Xs + arow[i] + xoff + kk * 32);
    fw = *reinterpret_cast<const bf16x8*>(Ws + woff + wrow[kk]);
  };
  auto mfma4 = [&](const bf16x8 (&fx)[FM], const bf16x8& fw) {
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fx[i], acc[i], 0, 0, 0);
  };

  const bool y32 = (d.flags & REHR_GG_Y_F32) != 0;
  __bf16* yb = reinterpret_cast<__bf16*>(d.y);
  auto flush_stats = [&]() {
    if (d.stats_mode == 0 || stats_n < 0) return;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a = s1[k], b = s2[k];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {   // over the 32 lanes of this half (voxels)
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
      }
      const int col = n0 + 8 * (k >> 2) + 4 * half + (k & 3);
      if ((lane & 31) == 0 && col < d.Cout) {
        double* st = d.stats + ((int64_t)stats_n * d.Cout + col) * 2;
        atomicAdd(st, (double)a);
        if (d.stats_mode == 2) atomicAdd(st + 1, (double)b);
      }
      s1[k] = s2[k] = 0.f;
    }
  };

  if (items > 0) {
    stage_weights(0);
    fetch(0);
    stage();
  }
  __syncthreads();

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
    fetch(it + 1);   // next halo: in flight during this sweep
    const bool k1 = (d.Cin - cc) >= 32;

    jd = jh = jw = 0;
    if (k1) {
      // two register sets: the reads of a k step are issued one MFMA group (4 MFMAs) before their use
      bf16x8 fx0[FM], fx1[FM], fw0, fw1;
      int xoff = tap_off();
      read_frags(xoff, 0, 0, fx0, fw0);
      for (int t = 0; t < ntaps; ++t) {
        read_frags(xoff, t * (BN * WROW), 1, fx1, fw1);
        mfma4(fx0, fw0);
        xoff = tap_off();                                   // tap t + 1 (wraps harmlessly past the end)
        const int tn = (t + 1 < ntaps) ? t + 1 : 0;
        read_frags(xoff, tn * (BN * WROW), 0, fx0, fw0);
        mfma4(fx1, fw1);
      }
    } else {   // half-filled last chunk (Cin % 32 == 16): one k step per tap
      for (int t = 0; t < ntaps; ++t) {
        bf16x8 fx[FM], fw;
        read_frags(tap_off(), t * (BN * WROW), 0, fx, fw);
        mfma4(fx, fw);
      }
    }

    if (chunk == p.kchunks - 1) {
      // ---- epilogue of this brick
      const int n_img = (int)(tile / p.tiles_per_img);
      if (n_img != stats_n) {
        flush_stats();
        stats_n = n_img;
      }
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
      // accumulator register r of tile i: channel n0 + 8*(r>>2) + 4*half + (r&3), voxel wave*128 + 32*i + (lane&31)
      float bv[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int col = n0 + 8 * (k >> 2) + 4 * half + (k & 3);
        bv[k] = (d.bias != nullptr && col < d.Cout) ? d.bias[col] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int off = row_out[wave * 128 + i * 32 + (lane & 31)];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = n0 + 8 * g + 4 * half;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[i][4 * g + e] + bv[4 * g + e], d.act, d.slope);
          if (off >= 0 && col + 3 < d.Cout) {        // Cout % 4 == 0: a quad is in or out as a whole
            if (y32) {
              f32x4 o = {v[0], v[1], v[2], v[3]};
              *reinterpret_cast<f32x4*>(d.y + (int64_t)off * d.ldy + col) = o;
            } else {
              bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
              *reinterpret_cast<bf16x4*>(yb + (int64_t)off * d.ldy + col) = o;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s1[4 * g + e] += v[e];
              s2[4 * g + e] = fmaf(v[e], v[e], s2[4 * g + e]);
            }
          }
        }
      }
    }
    __syncthreads();   // every wave is done reading this halo, these weights and row_out
    stage();           // next halo: registers -> LDS
    if (p.kchunks > 1 && it + 1 < items) stage_weights((int)((it + 1) % p.kchunks) * BK);
    __syncthreads();
  }
  flush_stats();
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

template <int BD, int BH, int BW>
int launch(HBParams p, hipStream_t stream) {
  const rehr_gather_gemm_desc& d = p.d;
  if (d.Ld < (BD + 1) / 2 || d.Lh < BH || d.Lw < BW) return REHR_ENOSUP;
  const int64_t nb_d = (d.Ld + BD - 1) / BD, nb_h = (d.Lh + BH - 1) / BH, nb_w = (d.Lw + BW - 1) / BW;
  if (nb_d * BD * nb_h * BH * nb_w * BW * 10 > (int64_t)d.Ld * d.Lh * d.Lw * 13) return REHR_ENOSUP;  // <= 1.3x padding
  p.HD = BD + p.HD; p.HH = BH + p.HH; p.HW = BW + p.HW;      // (the tap spans were left in HD/HH/HW)
  p.hvox = p.HD * p.HH * p.HW;
  if (p.HD > BD + 2 || p.HH > BH + 2 || p.HW > BW + 2) return REHR_ENOSUP;
  const int ntaps = d.td.count * d.th.count * d.tw.count;
  const size_t smem = (size_t)p.hvox * XROW + (size_t)ntaps * BN * WROW + (size_t)BVOX * sizeof(int);
  if (smem > 158 * 1024) return REHR_ENOSUP;
  p.nb_d = (int)nb_d; p.nb_h = (int)nb_h; p.nb_w = (int)nb_w;
  p.tiles_per_img = (int)(nb_d * nb_h * nb_w);
  p.ntiles = (int64_t)d.N * p.tiles_per_img;
  // persistent blocks, one per CU: whole rounds of 256
  const int n_tiles = d.Npad / BN;
  int64_t want = 256 / n_tiles;
  if (want < 1) want = 1;
  if (want > p.ntiles) want = p.ntiles;
  p.tiles_per_block = (int)((p.ntiles + want - 1) / want);
  auto kern = halo_conv_bf16_kernel<BD, BH, BW>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            158 * 1024) != hipSuccess)
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
  if (T < 9 || T > MAXTAPS) return REHR_ENOSUP;   // few taps: the gather kernel is already cheap per byte
  if (d.td.count > 3 || d.th.count > 3 || d.tw.count > 3) return REHR_ENOSUP;
  if (d.c1 < d.Cin && d.c1 % BK) return REHR_ENOSUP;
  if (d.Cout % 4 || d.ldy % 4) return REHR_ENOSUP;
  if (((uintptr_t)d.y & 15) || (d.ldy * ((d.flags & REHR_GG_Y_F32) ? 4 : 2)) % 8) return REHR_ENOSUP;
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
  // brick shapes in order of preference; one that pads the lattice by more than 1.3x (or does not fit it) declines
  int rc = launch<4, 8, 16>(p, stream);
  if (rc == REHR_ENOSUP) rc = launch<8, 8, 8>(p, stream);
  if (rc == REHR_ENOSUP) rc = launch<2, 16, 16>(p, stream);
  return rc;
}
