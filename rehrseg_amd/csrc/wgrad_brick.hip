// Weight gradient through LDS halo bricks (stride-1 lattice -> source maps, <= 28 taps).
//
// The slab kernel in wgrad.hip re-reads both operands once per tap.  Here a block owns one
// 32 x 32 (lattice channel x gathered channel) tile for ALL taps: it stages a 2x8x8 brick of
// the lattice tensor (dY) and the halo of the gathered tensor (x) around it in LDS once, and
// its four waves split the taps (7 accumulator tiles each for 3x3x3).  For a k-step (two
// adjacent voxels) the dY fragment is read once and reused by the wave's 7 MFMAs; every x
// fragment is one conflict-free ds_read_b32 at (brick row + tap offset).  Global traffic per
// MFMA drops ~25x, which is what the low-channel layers (32/64 channels at 128^3) need: they
// were L2-bandwidth bound, not MFMA bound.  Partial tiles go to the same slab layout as the
// slab kernel and are reduced by wgrad_reduce_kernel (deterministic).
#include "common.h"
#include "wgrad_shared.h"

namespace {

constexpr int BD = 2, BH = 8, BW = 8, BVOX = BD * BH * BW;  // lattice brick
constexpr int LDY = 36, LDX = 36;                            // LDS row strides (32 + 4)
constexpr int MAXTPW = 7;                                    // taps per wave (ceil(28 / 4))
constexpr int MAXX = 13;                                     // halo rows staged per thread (32 rows per pass)

struct BrickParams {
  WGParams w;
  int HD, HH, HW;        // halo extents
  int mind, minh, minw;  // smallest tap offset per axis (incl. the lattice->source offset b)
  int nb_d, nb_h, nb_w;  // bricks per sample along each axis
  int64_t nbricks;
  int bricks_per_split;
};

// TPW = taps per wave is a compile-time constant so the k-step body is branch-free: a wave
// whose share runs past the last tap recomputes tap T-1 and simply does not store it.
template <int TPW>
__global__ __launch_bounds__(256, 2) void wgrad_brick_kernel(const BrickParams p) {
  const rehr_wgrad_desc& d = p.w.d;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ys = smem;                 // [BVOX][LDY]
  float* Xs = smem + BVOX * LDY;    // [hvox][LDX]
  const int hvox = p.HD * p.HH * p.HW;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = b % p.w.c_tiles;
  const int at = b / p.w.c_tiles;
  const int a0 = at * 32, c0 = ct * 32;
  const int split = blockIdx.y;
  const int T = p.w.T;
  const int t_begin = wave * TPW;

  // per-wave tap table (wave-uniform scalars)
  int tapbase[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int t = t_begin + j;
    const int tt = t < T ? t : T - 1;
    const int jw = tt % d.tw.count;
    const int jh = (tt / d.tw.count) % d.th.count;
    const int jd = tt / (d.tw.count * d.th.count);
    const int od_ = d.bd + d.td.off0 + d.td.offs * jd - p.mind;
    const int oh_ = d.bh + d.th.off0 + d.th.offs * jh - p.minh;
    const int ow_ = d.bw + d.tw.off0 + d.tw.offs * jw - p.minw;
    tapbase[j] = (od_ * p.HH + oh_) * p.HW + ow_;
  }

  f32x16 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float bsum = 0.f;
  const float bias_w = (p.w.slab_bias != nullptr && ct == 0 && wave == 0) ? 1.f : 0.f;

  const int64_t lvox = (int64_t)d.Ld * d.Lh * d.Lw;
  const int64_t gvox = (int64_t)d.Dg * d.Hg * d.Wg;
  const int col = lane & 31, half = lane >> 5;
  const int q = tid & 7, r0 = tid >> 3;  // staging: 8 x 16-byte pieces per 32-channel row

  const int64_t br_begin = (int64_t)split * p.bricks_per_split;
  int64_t br_end = br_begin + p.bricks_per_split;
  if (br_end > p.nbricks) br_end = p.nbricks;

  // Halo pieces owned by this thread: the (hd,hh,hw) split of a halo row does not depend on
  // the brick, so it is done once here (no divisions inside the brick loop).
  int hcoord[MAXX];  // hd<<20 | hh<<10 | hw, or -1 past the halo
#pragma unroll
  for (int i = 0; i < MAXX; ++i) {
    const int hv = r0 + 32 * i;
    const int hw_ = hv % p.HW;
    const int t2 = hv / p.HW;
    hcoord[i] = hv < hvox ? (((t2 / p.HH) << 20) | ((t2 % p.HH) << 10) | hw_) : -1;
  }
  const __amdgpu_buffer_rsrc_t rl_ = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.l), 0, (uint32_t)((int64_t)d.N * lvox * d.ldl * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rg_ = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.g), 0, (uint32_t)((int64_t)d.N * gvox * d.ldg * 4), 0x00020000);
  const uint32_t l_oob = (uint32_t)((int64_t)d.N * lvox * d.ldl * 4);
  const uint32_t g_oob = (uint32_t)((int64_t)d.N * gvox * d.ldg * 4);
  const bool a_ok = (a0 + q * 4) < d.Ca, c_ok = (c0 + q * 4) < d.Cg;

  // Branch-free fetch of one brick into registers (buffer loads: out of range -> 0); it is
  // issued for brick br+1 before the MFMA sweep of brick br so HBM/L2 latency and the address
  // arithmetic sit in the shadow of ~29k cycles of matrix work.
  f32x4 ry[BVOX / 32], rx[MAXX];
  auto fetch = [&](int64_t br) {
    const bool live = br < br_end;
    const int64_t bb = live ? br : br_begin;
    const int bw_ = (int)(bb % p.nb_w);
    int64_t r = bb / p.nb_w;
    const int bh_ = (int)(r % p.nb_h); r /= p.nb_h;
    const int bd_ = (int)(r % p.nb_d);
    const int n = (int)(r / p.nb_d);
    const int od0 = bd_ * BD, oh0 = bh_ * BH, ow0 = bw_ * BW;
    const uint32_t lbase = (uint32_t)n * (uint32_t)lvox;
#pragma unroll
    for (int i = 0; i < BVOX / 32; ++i) {
      const int v = r0 + 32 * i;
      const int od = od0 + (v >> 6), oh = oh0 + ((v >> 3) & 7), ow = ow0 + (v & 7);
      const bool ok = live & (od < d.Ld) & (oh < d.Lh) & (ow < d.Lw) & a_ok;
      const uint32_t off = (lbase + (uint32_t)((od * d.Lh + oh) * d.Lw + ow)) * (uint32_t)d.ldl * 4u +
                           (uint32_t)(a0 + q * 4) * 4u;
      ry[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl_, ok ? off : l_oob, 0, 0));
    }
    const int gd0 = od0 + p.mind, gh0 = oh0 + p.minh, gw0 = ow0 + p.minw;
    const uint32_t gbase = (uint32_t)n * (uint32_t)gvox;
#pragma unroll
    for (int i = 0; i < MAXX; ++i) {
      const int hc = hcoord[i];
      const int id = gd0 + (hc >> 20), ih = gh0 + ((hc >> 10) & 1023), iw = gw0 + (hc & 1023);
      const bool ok = live & (hc >= 0) & ((unsigned)id < (unsigned)d.Dg) & ((unsigned)ih < (unsigned)d.Hg) &
                      ((unsigned)iw < (unsigned)d.Wg) & c_ok;
      const uint32_t off = (gbase + (uint32_t)((id * d.Hg + ih) * d.Wg + iw)) * (uint32_t)d.ldg * 4u +
                           (uint32_t)(c0 + q * 4) * 4u;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg_, ok ? off : g_oob, 0, 0));
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < BVOX / 32; ++i) *reinterpret_cast<f32x4*>(Ys + (r0 + 32 * i) * LDY + q * 4) = ry[i];
#pragma unroll
    for (int i = 0; i < MAXX; ++i)
      if (hcoord[i] >= 0) *reinterpret_cast<f32x4*>(Xs + (r0 + 32 * i) * LDX + q * 4) = rx[i];
  };

  const float* yb = Ys + half * LDY + col;
  const float* xb = Xs + half * LDX + col;
  if (br_begin < br_end) {
    fetch(br_begin);
    stage();
  }
  __syncthreads();
  for (int64_t br = br_begin; br < br_end; ++br) {
    fetch(br + 1);  // past the end: every offset is out of range
    // ---- 64 k-steps (voxel pairs along w), all of this wave's taps per step
    // LDS fragments are read one k-step ahead (two named register sets): a 32x32x2 MFMA
    // issues every 64 cycles, about one LDS round trip, so reading "just in time" stalls it.
    float a0_, x0_[TPW], a1_, x1_[TPW];
    auto lds_read = [&](int ks, float& a, float (&x)[TPW]) {
      const int v = 2 * ks;
      const int row0 = ((v >> 6) * p.HH + ((v >> 3) & 7)) * p.HW + (v & 7);
      a = yb[v * LDY];
#pragma unroll
      for (int j = 0; j < TPW; ++j) x[j] = xb[(row0 + tapbase[j]) * LDX];
    };
    auto mfma_step = [&](const float a, const float (&x)[TPW]) {
      bsum += a * bias_w;
#pragma unroll
      for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x[j], acc[j], 0, 0, 0);
    };
    lds_read(0, a0_, x0_);
    for (int ks = 0; ks < BVOX / 2; ks += 2) {
      lds_read(ks + 1, a1_, x1_);
      mfma_step(a0_, x0_);
      lds_read((ks + 2) & (BVOX / 2 - 1), a0_, x0_);  // wraps to step 0 after the last pair (unused)
      mfma_step(a1_, x1_);
    }
    __syncthreads();  // every wave is done with this brick
    stage();          // next brick: registers -> LDS
    __syncthreads();
  }

  // ---- partial tiles -> slab[split][tap][Capad][Cgpad]
  const int chalf = lane >> 5;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int t = t_begin + j;
    if (t < T) {
      float* slab = d.workspace + (((int64_t)split * T + t) * p.w.Capad) * p.w.Cgpad;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = a0 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
        slab[(int64_t)row * p.w.Cgpad + c0 + col] = acc[j][r];
      }
    }
  }
  if (bias_w != 0.f) {
    const float v = bsum + __shfl_xor(bsum, 32, 64);
    const int row = a0 + col;
    if (chalf == 0 && row < d.Ca) p.w.slab_bias[(int64_t)split * d.Ca + row] = v;
  }
}

void axis_span(const rehr_axis_taps& t, int b, int* mn, int* mx) {
  int lo = b + t.off0, hi = b + t.off0;
  for (int j = 1; j < t.count; ++j) {
    const int o = b + t.off0 + t.offs * j;
    if (o < lo) lo = o;
    if (o > hi) hi = o;
  }
  *mn = lo;
  *mx = hi;
}

}  // namespace

// Returns true when the brick kernel applies; fills the plan (splits, slab geometry).
bool wgrad_brick_plan(const rehr_wgrad_desc& d, WGParams& w, BrickPlanOut& out) {
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return false;
  const int T = d.td.count * d.th.count * d.tw.count;
  if (T < 3 || T > 4 * MAXTPW) return false;
  if (d.Ca % 4 || d.Cg % 4) return false;  // partial 32-tiles are masked (sr_head.0: 16 x 32)
  // Measured on MI355X: the brick kernel wins for thin tensors (32/64 channels: 52 -> 95 TF,
  // 90 -> 99 TF), the slab kernel's 128x128 tiles win from 128 channels up (110 vs 114 TF).
  if ((int64_t)((d.Ca + 31) / 32) * ((d.Cg + 31) / 32) > 8) return false;
  if (d.Ld < BD || d.Lh < BH || d.Lw < BW) return false;
  // padding waste of the 2x8x8 brick must stay small
  const int64_t nb_d = (d.Ld + BD - 1) / BD, nb_h = (d.Lh + BH - 1) / BH, nb_w = (d.Lw + BW - 1) / BW;
  const int64_t padded = nb_d * BD * nb_h * BH * nb_w * BW;
  if (padded * 10 > (int64_t)d.Ld * d.Lh * d.Lw * 13) return false;
  int mn[3], mx[3];
  axis_span(d.td, d.bd, &mn[0], &mx[0]);
  axis_span(d.th, d.bh, &mn[1], &mx[1]);
  axis_span(d.tw, d.bw, &mn[2], &mx[2]);
  const int HD = BD + mx[0] - mn[0], HH = BH + mx[1] - mn[1], HW = BW + mx[2] - mn[2];
  const int64_t hvox = (int64_t)HD * HH * HW;
  const size_t smem = (size_t)(BVOX * LDY + hvox * LDX) * sizeof(float);
  if (smem > 80 * 1024 || hvox > MAXX * 32 || HH > 1023 || HW > 1023) return false;
  // buffer-addressed operands: 32-bit byte offsets over the whole tensors
  if ((int64_t)d.N * d.Ld * d.Lh * d.Lw * d.ldl * 4 >= (1ll << 32) - 64 ||
      (int64_t)d.N * d.Dg * d.Hg * d.Wg * d.ldg * 4 >= (1ll << 32) - 64)
    return false;
  w.d = d;
  w.T = T;
  w.a_tiles = (d.Ca + 31) / 32;
  w.c_tiles = (d.Cg + 31) / 32;
  w.Capad = w.a_tiles * 32;
  w.Cgpad = w.c_tiles * 32;
  w.kv_total = (int64_t)d.N * d.Ld * d.Lh * d.Lw;
  out.HD = HD; out.HH = HH; out.HW = HW;
  out.mind = mn[0]; out.minh = mn[1]; out.minw = mn[2];
  out.nb_d = (int)nb_d; out.nb_h = (int)nb_h; out.nb_w = (int)nb_w;
  out.nbricks = (int64_t)d.N * nb_d * nb_h * nb_w;
  const int64_t tiles = (int64_t)w.a_tiles * w.c_tiles;
  // 512 blocks are resident at once; aim for full rounds, >= 4 bricks per block
  int64_t want = (1024 + tiles - 1) / tiles;
  const int64_t max_by_k = (out.nbricks + 3) / 4;
  if (want > max_by_k) want = max_by_k;
  if (want < 1) want = 1;
  {
    int64_t best = want;
    double best_eff = 0.0;
    const int64_t lo = want > 2 ? want - want / 3 : 1;
    int64_t hi = want + want / 2 + 1;
    if (hi > max_by_k) hi = max_by_k > want ? max_by_k : want;
    for (int64_t s = lo; s <= hi; ++s) {
      const double rounds = (double)(tiles * s) / 512.0;
      const double eff = rounds / (double)(int64_t)(rounds + 0.999999);
      if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
    }
    want = best;
  }
  if (want > 65535) want = 65535;
  out.bricks_per_split = (int)((out.nbricks + want - 1) / want);
  w.splits = (int)((out.nbricks + out.bricks_per_split - 1) / out.bricks_per_split);
  w.kv_per_split = 0;
  out.smem = smem;
  return true;
}

int wgrad_brick_launch(const WGParams& w, const BrickPlanOut& o, hipStream_t stream) {
  BrickParams p;
  p.w = w;
  p.HD = o.HD; p.HH = o.HH; p.HW = o.HW;
  p.mind = o.mind; p.minh = o.minh; p.minw = o.minw;
  p.nb_d = o.nb_d; p.nb_h = o.nb_h; p.nb_w = o.nb_w;
  p.nbricks = o.nbricks;
  p.bricks_per_split = o.bricks_per_split;
  dim3 grid(w.a_tiles * w.c_tiles, w.splits, 1);
  const int tpw = (w.T + 3) / 4;
#define BRICK_CASE(N_)                                                                                   \
  case N_: {                                                                                             \
    static bool attr_set = false;                                                                        \
    if (!attr_set) {                                                                                     \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_brick_kernel<N_>),                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)      \
        return REHR_EHIP;                                                                                \
      attr_set = true;                                                                                   \
    }                                                                                                    \
    hipLaunchKernelGGL(wgrad_brick_kernel<N_>, grid, dim3(256), o.smem, stream, p);                      \
  } break;
  switch (tpw) {
    BRICK_CASE(1) BRICK_CASE(2) BRICK_CASE(3) BRICK_CASE(4) BRICK_CASE(5) BRICK_CASE(6) BRICK_CASE(7)
    default: return REHR_ENOSUP;
  }
#undef BRICK_CASE
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
