// Weight gradient of unit-stride 3x3x3 / 1x3x3 convolutions on the bf16 matrix cores with BOTH operands staged as
// bricks in LDS (mixed-precision path; fp32 accumulate, fp32 slabs and result):
//
//   dW[tap][a][c] = sum over lattice voxels v of  L[v][a] * G[v + off(tap)][c]      (L = dY, G = x for Conv3d)
//
// wgrad_bf16_kernel (wgrad.hip) treats every tap as its own GEMM and re-reads both operands from L2 27 times; on the
// 32/64-channel layers at 128^3-160^3 it reaches 0.1-0.2 PFLOP/s and was the largest entry of a mixed-precision step.
// Here a block owns one 32 x 32 channel tile pair and a run of 512-voxel bricks (4 x 8 x 16):
//   * per brick the L brick (512 rows x 64 B) and the G brick WITH ITS HALO (<= 6 x 10 x 18 rows x 64 B) are staged in
//     LDS once and serve all 27 taps;
//   * the MFMA k index is the voxel, 16 consecutive voxels along w per step; both operands are [voxel][channel]
//     images (64-byte rows, no padding needed) read with the transposing ds_read_b64_tr_b16: conflict-free, two reads
//     per fragment;
//   * the 4 waves split the TAPS (wave w takes taps w, w+4, ...: 7 accumulator tiles at most = 112 registers); the
//     L fragment of a k step is read once per wave and reused by its taps;
//   * the 32 k steps of a brick are fully unrolled: every LDS address is a per-lane base (one per tap, set up once
//     per block) plus an immediate, the loop body is 16 LDS reads + 7 MFMAs + 1 global load of the next brick;
//   * accumulators live across all bricks of the block; one fp32 slab per block, summed in a fixed order by
//     wgrad_reduce_kernel (bitwise reproducible).
#include "common.h"
#include "wgrad_shared.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int BD = 4, BH = 8, BW = 16, BVOX = BD * BH * BW;   // brick
constexpr int ROW = 64;                                       // bytes per voxel row of an LDS image (32 bf16)
// NWV = waves per block: 4 (one per SIMD, <= 7 taps each) or 8 (two per SIMD, <= 4 taps each: the second resident wave
// fills the issue gaps of the first; the L fragment of a k step then serves 3-4 MFMAs instead of 6-7)
template <int NWV> struct WBW {
  static constexpr int RPP = 16 * NWV;                        // rows staged per pass (4 threads x 16 B per row)
  static constexpr int LPASS = BVOX / RPP;                    // passes for the L brick
  static constexpr int MAXW = (27 + NWV - 1) / NWV;           // taps per wave
};

// KD = taps along depth (3 or 1); HD/HH/HW = halo extents
template <int KD, int NWV>
__global__ __launch_bounds__(64 * NWV, 1) void wgrad_brick_bf16_kernel(const WGParams p, const BrickBf16 g) {
  constexpr int RPP = WBW<NWV>::RPP, LPASS = WBW<NWV>::LPASS, MAXW = WBW<NWV>::MAXW;
  constexpr int HD = BD + KD - 1, HH = BH + 2, HW = BW + 2, HVOX = HD * HH * HW;
  constexpr int GPASS = (HVOX + RPP - 1) / RPP;                // 17 (KD 3) / 12 (KD 1)
  constexpr int NT = KD * 9;
  const rehr_wgrad_desc& d = p.d;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* Gs = smem_b;                       // [GPASS * RPP][ROW]
  unsigned char* Ls = smem_b + GPASS * RPP * ROW;   // [BVOX][ROW]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q4 = tid & 3, r0 = tid >> 2;
  const int at = blockIdx.y / p.c_tiles, ct = blockIdx.y - at * p.c_tiles;
  const int a0 = at * 32, c0 = ct * 32;

  // halo rows owned by this thread
  int hcoord[GPASS], hrel[GPASS];
#pragma unroll
  for (int i = 0; i < GPASS; ++i) {
    const int hv = r0 + RPP * i;
    const int hw_ = hv % HW, t2 = hv / HW, hd_ = t2 / HH, hh_ = t2 % HH;
    hcoord[i] = hv < HVOX ? ((hd_ << 20) | (hh_ << 10) | hw_) : (1023 << 20);
    hrel[i] = (hd_ * d.Hg + hh_) * d.Wg + hw_;
  }
  // brick rows owned by this thread (row = (d * BH + h) * BW + w)
  int lrel[LPASS], lcoord[LPASS];
#pragma unroll
  for (int i = 0; i < LPASS; ++i) {
    const int v = r0 + RPP * i;
    const int vw = v % BW, vh = (v / BW) % BH, vd = v / (BW * BH);
    lcoord[i] = (vd << 20) | (vh << 10) | vw;
    lrel[i] = (vd * d.Lh + vh) * d.Lw + vw;
  }

  const uint32_t g_img = (uint32_t)d.Dg * d.Hg * d.Wg, l_img = (uint32_t)d.Ld * d.Lh * d.Lw;
  const __bf16* gp = reinterpret_cast<const __bf16*>(d.g);
  const __bf16* lp = reinterpret_cast<const __bf16*>(d.l);
  const bool g_ok = (c0 + q4 * 8) < d.Cg, l_ok = (a0 + q4 * 8) < d.Ca;
  const uint32_t g_cb = (uint32_t)(c0 + q4 * 8) * 2u, l_cb = (uint32_t)(a0 + q4 * 8) * 2u;
  const uint32_t ldgb = (uint32_t)d.ldg * 2u, ldlb = (uint32_t)d.ldl * 2u;
  const uint32_t g_nrec = g_img * ldgb, l_nrec = l_img * ldlb;

  // transposed fragment reads: 16-lane group -> (k half h, channel half cg); lane 4q+pp of a group supplies voxel row q,
  // channels 4pp .. 4pp+3 of the block
  const int grp = lane >> 4, hk = grp >> 1, cg = grp & 1, qq = (lane & 15) >> 2, pp = lane & 3;
  const int lane_row = 8 * hk + qq;                          // voxel row inside a 16-voxel k step
  const int lane_col = (16 * cg + 4 * pp) * 2;               // byte offset of the lane's 4 channels

  // this wave's taps: t = wave + NWV j; per-lane LDS base of each (the halo offset of the tap + the lane's row / column)
  int gbase[MAXW];
  int ntw = 0;
#pragma unroll
  for (int j = 0; j < MAXW; ++j) {
    const int t = wave + NWV * j;
    const bool on = t < NT;
    const int tt = on ? t : 0;
    const int jd = tt / 9, jh = (tt / 3) % 3, jw = tt % 3;
    // g position = lattice position + b + off(tap); halo origin = lattice brick origin + min offset
    const int od_ = (KD == 3) ? (d.bd + d.td.off0 + d.td.offs * jd - g.mind) : 0;
    const int oh_ = d.bh + d.th.off0 + d.th.offs * jh - g.minh;
    const int ow_ = d.bw + d.tw.off0 + d.tw.offs * jw - g.minw;
    gbase[j] = ((od_ * HH + oh_) * HW + ow_ + lane_row) * ROW + lane_col;
    ntw += on ? 1 : 0;
  }
  const int lbase = lane_row * ROW + lane_col;

  f32x16 acc[MAXW];
#pragma unroll
  for (int j = 0; j < MAXW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int t_begin = blockIdx.x * g.tiles_per_block;
  int t_end = t_begin + g.tiles_per_block;
  if (t_end > g.ntiles) t_end = g.ntiles;
  const int ntile = t_end > t_begin ? t_end - t_begin : 0;

  u32x4 rg[GPASS], rl[LPASS];
  uint32_t goff[GPASS], loff[LPASS];
  __amdgpu_buffer_rsrc_t grs, lrs;
  auto prep = [&](int ti) {   // load offsets of brick t_begin + ti (zero fill outside the tensors / past the last brick)
    const bool live = ti < ntile;
    const int tile = t_begin + (live ? ti : 0);
    const int n = tile / g.tiles_per_img;
    int tr = tile - n * g.tiles_per_img;
    const int bw_ = tr % g.nb_w; tr /= g.nb_w;
    const int bh_ = tr % g.nb_h;
    const int bd_ = tr / g.nb_h;
    const int ld0 = bd_ * BD, lh0 = bh_ * BH, lw0 = bw_ * BW;
    const int gd0 = ld0 + g.mind, gh0 = lh0 + g.minh, gw0 = lw0 + g.minw;
    grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(gp) + (int64_t)n * g_img * d.ldg, 0, g_nrec, 0x00020000);
    lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(lp) + (int64_t)n * l_img * d.ldl, 0, l_nrec, 0x00020000);
    const int gv0 = (gd0 * d.Hg + gh0) * d.Wg + gw0, lv0 = (ld0 * d.Lh + lh0) * d.Lw + lw0;
#pragma unroll
    for (int i = 0; i < GPASS; ++i) {
      const int hc = hcoord[i];
      const bool ok = live & g_ok & ((unsigned)(gd0 + (hc >> 20)) < (unsigned)d.Dg) &
                      ((unsigned)(gh0 + ((hc >> 10) & 1023)) < (unsigned)d.Hg) & ((unsigned)(gw0 + (hc & 1023)) < (unsigned)d.Wg);
      goff[i] = ok ? (uint32_t)(gv0 + hrel[i]) * ldgb + g_cb : g_nrec;
    }
#pragma unroll
    for (int i = 0; i < LPASS; ++i) {
      const int lc = lcoord[i];
      const bool ok = live & l_ok & ((ld0 + (lc >> 20)) < d.Ld) & ((lh0 + ((lc >> 10) & 1023)) < d.Lh) &
                      ((lw0 + (lc & 1023)) < d.Lw);
      loff[i] = ok ? (uint32_t)(lv0 + lrel[i]) * ldlb + l_cb : l_nrec;
    }
  };
#define WB_ISSUE(s)                                                                   \
  if ((s) < GPASS) rg[(s) < GPASS ? (s) : 0] = __builtin_amdgcn_raw_buffer_load_b128(grs, goff[(s) < GPASS ? (s) : 0], 0, 0); \
  else if ((s) - GPASS < LPASS)                                                       \
    rl[(s) - GPASS < LPASS && (s) >= GPASS ? (s) - GPASS : 0] =                       \
        __builtin_amdgcn_raw_buffer_load_b128(lrs, loff[(s) - GPASS < LPASS && (s) >= GPASS ? (s) - GPASS : 0], 0, 0)
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < GPASS; ++i) *reinterpret_cast<u32x4*>(Gs + (r0 + RPP * i) * ROW + q4 * 16) = rg[i];
#pragma unroll
    for (int i = 0; i < LPASS; ++i) *reinterpret_cast<u32x4*>(Ls + (r0 + RPP * i) * ROW + q4 * 16) = rl[i];
  };
  auto frag = [&](const unsigned char* img, int off) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off + 4 * ROW));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  if (ntile > 0) {
    prep(0);
#pragma unroll
    for (int s = 0; s < GPASS + LPASS; ++s) { WB_ISSUE(s); }
    stage();
  }
  __syncthreads();

  for (int ti = 0; ti < ntile; ++ti) {
    prep(ti + 1);
    // 32 k steps of 16 voxels: step s = (d, h) row of the brick, voxels w = 0..15
#pragma unroll
    for (int s = 0; s < BD * BH; ++s) {
      const int sd = s / BH, sh = s % BH;
      const int goffs = ((sd * HH + sh) * HW) * ROW;      // compile-time: immediate offsets after unrolling
      const int loffs = ((sd * BH + sh) * BW) * ROW;
      if (s < GPASS + LPASS) { WB_ISSUE(s); }             // next brick's loads, one per k step
      const bf16x8 fl = frag(Ls, lbase + loffs);
#pragma unroll
      for (int j = 0; j < MAXW; ++j) {
        if (j < ntw) {                                    // (wave-uniform)
          const bf16x8 fg = frag(Gs, gbase[j] + goffs);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, fg, acc[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();   // every wave is done with these bricks
    stage();
    __syncthreads();
  }
#undef WB_ISSUE

  // slab[blockIdx.x][tap][a][c] (fp32); accumulator register r: row a = (r&3) + 8*(r>>2) + 4*(lane>>5), column c = lane&31
  float* slab = d.workspace + ((int64_t)blockIdx.x * p.T) * p.Capad * p.Cgpad;
#pragma unroll
  for (int j = 0; j < MAXW; ++j) {
    const int t = wave + NWV * j;
    if (t < NT) {
      float* o = slab + ((int64_t)t * p.Capad + a0) * p.Cgpad + c0 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * p.Cgpad] = acc[j][r];
    }
  }
}

}  // namespace

// Fills the slab geometry of `w` (splits = blocks along the bricks, padded channel counts) and the brick plan.
bool wgrad_brick_bf16_plan(const rehr_wgrad_desc& d, WGParams& w, BrickBf16& o) {
  if (d.sd != 1 || d.sh != 1 || d.sw != 1 || d.dbias != nullptr) return false;
  const int KD = d.td.count;
  if ((KD != 3 && KD != 1) || d.th.count != 3 || d.tw.count != 3) return false;
  // taps must be the contiguous {-1, 0, +1} window (any order) on the axes that have three
  auto win = [](const rehr_axis_taps& t, int b, int* mn) {
    int lo = b + t.off0, hi = lo;
    for (int j = 1; j < t.count; ++j) {
      const int v = b + t.off0 + t.offs * j;
      lo = v < lo ? v : lo;
      hi = v > hi ? v : hi;
    }
    *mn = lo;
    return hi - lo == t.count - 1 && (t.offs == 1 || t.offs == -1 || t.count == 1);
  };
  if (!win(d.td, d.bd, &o.mind) || !win(d.th, d.bh, &o.minh) || !win(d.tw, d.bw, &o.minw)) return false;
  if (d.Ca % 8 || d.Cg % 8 || d.ldl % 8 || d.ldg % 8) return false;
  if (d.Ld < 2 || d.Lh < 8 || d.Lw < 16) return false;
  const int64_t nb_d = (d.Ld + BD - 1) / BD, nb_h = (d.Lh + BH - 1) / BH, nb_w = (d.Lw + BW - 1) / BW;
  if (nb_d * BD * nb_h * BH * nb_w * BW * 10 > (int64_t)d.Ld * d.Lh * d.Lw * 13) return false;
  const int64_t gimg = (int64_t)d.Dg * d.Hg * d.Wg * d.ldg * 2, limg = (int64_t)d.Ld * d.Lh * d.Lw * d.ldl * 2;
  if (gimg >= (1ll << 32) - 64 || limg >= (1ll << 32) - 64) return false;
  o.nb_d = (int)nb_d; o.nb_h = (int)nb_h; o.nb_w = (int)nb_w;
  o.tiles_per_img = (int)(nb_d * nb_h * nb_w);
  const int64_t ntiles = (int64_t)d.N * o.tiles_per_img;
  if (ntiles >= (1ll << 31) - 4096) return false;
  o.ntiles = (int)ntiles;
  w.d = d;
  w.T = KD * 9;
  w.a_tiles = (d.Ca + 31) / 32;
  w.c_tiles = (d.Cg + 31) / 32;
  w.Capad = w.a_tiles * 32;
  w.Cgpad = w.c_tiles * 32;
  const int pairs = w.a_tiles * w.c_tiles;
  int64_t want = 256 / pairs;            // one block per CU
  if (want < 1) want = 1;
  if (want > ntiles) want = ntiles;
  o.tiles_per_block = (int)((ntiles + want - 1) / want);
  w.splits = (int)((ntiles + o.tiles_per_block - 1) / o.tiles_per_block);
  w.slab_bias = nullptr;
  return true;
}

template <int KD, int NWV>
static int wgrad_brick_bf16_launch_t(const WGParams& w, const BrickBf16& o, hipStream_t stream) {
  constexpr int RPP = WBW<NWV>::RPP;
  const int hvox = (BD + KD - 1) * (BH + 2) * (BW + 2);
  const size_t smem = (size_t)((hvox + RPP - 1) / RPP) * RPP * ROW + (size_t)BVOX * ROW;
  const dim3 grid(w.splits, w.a_tiles * w.c_tiles, 1);
  auto kern = wgrad_brick_bf16_kernel<KD, NWV>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024) !=
        hipSuccess)
      return REHR_EHIP;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * NWV), smem, stream, w, o);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

int wgrad_brick_bf16_launch(const WGParams& w, const BrickBf16& o, hipStream_t stream) {
  // eight waves per block (<= 4 taps per wave); four waves measured 0.85 ms per bf16 cfg-3 step slower
  // (profiles/r03_ab_superseded.txt)
  if (w.d.td.count == 3) return wgrad_brick_bf16_launch_t<3, 8>(w, o, stream);
  return wgrad_brick_bf16_launch_t<1, 8>(w, o, stream);
}
