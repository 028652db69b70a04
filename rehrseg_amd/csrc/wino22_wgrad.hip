// Winograd F(2x2, 2x2) weight gradient for the stride-2, 4-tap transposed convolutions of the FLAVR decoder
// (models/FLAVR/FLAVR_arch.py:40-70): per source parity (p_h, p_w) of dY the 2 x 2 taps k = p + 2a' are a
// unit-stride 2-tap problem over that parity's sub-lattice,
//
//   dU[p][jd][xi][ca][cg] = sum_{n, od, 2x2 lattice tiles} (A x A^T)[xi][tile][ca] * (B^T dY_p B)[xi][tile][cg]
//   dW[ca][cg][kd][k_h(p,a')][k_w(p,b')] = sum_{r in {a',a'+1}} sum_{c in {b',b'+1}} dU[p][jd][(r,c)]
//
// with A = [[1,0],[1,1],[0,1]], B^T = [[1,-1,0],[0,1,0],[0,-1,1]]: 9 products per tile instead of 16.
//
//   block  = 64 x-channels x 64 dY-channels of one (parity, depth tap), 12 waves = (Winograd row, 32-channel
//            group of x, 32-channel group of dY) with the row's 3 accumulator tiles; three waves per SIMD
//   stage  = 4 x 16 lattice outputs: x tile and the 5 x 17 sub-lattice patch of dY in LDS, TRANSPOSED to
//            [row][channel][column] while staging (conflict-free scalar writes: 16 columns x 4 channel quads
//            per wave instruction), so a lane (= channel) reads its 8-9 consecutive columns as 128-bit words:
//            10 LDS reads per 12 MFMAs instead of the 34 scalar reads of the voxel-major layout
//   split-K over (sample, depth, region) ranges; slabs reduced in a fixed order with the G^T . G fold.
#include "common.h"
#include "wgrad_shared.h"
#include "wino22_shared.h"
#include <cstdlib>

namespace {

constexpr int RH = 4, RW = 16;      // lattice outputs per stage (2 x 8 tiles)
constexpr int XH = 5, XW = 17;      // sub-lattice patch
constexpr int WP = 20;              // column pitch (floats) of a [row][channel] line
constexpr int YT = RH * 64 * WP, XT = XH * 64 * WP, STG = YT + XT;  // floats per stage
constexpr int NTH = 768;

struct Phase22 {
  int ph, pw, dh0, dw0, kh[2], kw[2];
};

struct WW22Params {
  rehr_wgrad_desc d;
  int nb_h, nb_w, items, items_per_split, splits, a_tiles, c_tiles, Capad, Cgpad, nphase, sh, sw;
  Phase22 phase[4];
  float* slabs;  // [splits][nphase*KD][9][Capad][Cgpad]
};

__global__ __launch_bounds__(NTH) void wino22_wgrad_kernel(const WW22Params p) {
  const rehr_wgrad_desc& d = p.d;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = wv % 3, fa = (wv / 3) & 1, fb = wv / 6;
  const int half = lane >> 5, col = lane & 31;
  const int split = blockIdx.x;
  const int at = blockIdx.y / p.c_tiles, ct = blockIdx.y - at * p.c_tiles;
  const int z = blockIdx.z;
  const int ph_i = z / d.td.count, jd = z - ph_i * d.td.count;
  const Phase22& P = p.phase[ph_i];
  const int ca0 = at * 64, cg0 = ct * 64;
  // a depth tap walks only the output slices whose source slice exists (unless REHR_DBG_WGRAD_NO_TAP_SKIP)
  const bool skip = !(d.debug_flags & REHR_DBG_WGRAD_NO_TAP_SKIP);
  const int doff_ = d.bd + d.td.off0 + d.td.offs * jd;
  const int od_lo = skip ? max(0, -doff_) : 0, od_hi = skip ? min(d.Ld, d.Dg - doff_) : d.Ld;
  const int nod = max(0, od_hi - od_lo);
  const int items_tap = d.N * nod * p.nb_h * p.nb_w;
  const int ips = skip ? (items_tap + (int)gridDim.x - 1) / (int)gridDim.x : p.items_per_split;
  const int it0 = split * ips;
  const int it1 = min(it0 + ips, skip ? items_tap : p.items);
  const int nstages = max(0, it1 - it0);

  // Z rows (A x): (x0, x0+x1, x1); V rows (B^T d): (d0-d1, d1, d2-d1)
  const float zka = (r == 2) ? 0.f : 1.f, zkb = (r == 0) ? 0.f : 1.f;
  const float vkb = (r == 1) ? 0.f : 1.f;

  // ---- staging: piece = 4 channels of one voxel; a wave instruction covers 16 columns x 4 quads of one
  // (row, quad group), written as 4 scalar stores into [row][channel][column] (64 distinct banks)
  const int w16 = lane & 15, ql = lane >> 4;
  const int64_t l_img = (int64_t)d.Ld * d.Lh * d.Lw * d.ldl, g_img = (int64_t)d.Dg * d.Hg * d.Wg * d.ldg;
  const uint32_t l_bytes = (uint32_t)(l_img * 4), g_bytes = (uint32_t)(g_img * 4);
  f32x4 ry[2], rx[2], rt;
  // the stage walk keeps (region column, region row, depth, sample) counters: one division chain per block
  int s_bw, s_bh, s_od, s_n;
  {
    int it = it0;
    s_bw = it % p.nb_w; it /= p.nb_w;
    s_bh = it % p.nb_h; it /= p.nb_h;
    const int nd = max(nod, 1);
    s_od = od_lo + it % nd;
    s_n = it / nd;
  }
  auto fetch = [&](int st) {
    const bool live = st < nstages;
    const int bw_ = s_bw, bh_ = s_bh, od = s_od, n = live ? s_n : 0;
    if (++s_bw == p.nb_w) {
      s_bw = 0;
      if (++s_bh == p.nb_h) {
        s_bh = 0;
        if (++s_od == od_hi) { s_od = od_lo; ++s_n; }
      }
    }
    const int oh0 = bh_ * RH, ow0 = bw_ * RW;
    const int id = od + d.bd + d.td.off0 + d.td.offs * jd;
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.l) + (int64_t)n * l_img, 0, l_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.g) + (int64_t)n * g_img, 0, g_bytes, 0x00020000);
    const bool dok = live & ((unsigned)id < (unsigned)d.Dg);
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // x tile: combos (row, quad group) = wv + 12 i < 16
      const int combo = wv + 12 * i;
      const int row = combo >> 2, q = (combo & 3) * 4 + ql;
      const int gh = oh0 + row, gw = ow0 + w16;
      const bool ok = live & (combo < 16) & (gh < d.Lh) & (gw < d.Lw) & ((ca0 + 4 * q) < d.Ca);
      const uint32_t off = (uint32_t)(((od * d.Lh + gh) * d.Lw + gw) * d.ldl + ca0 + 4 * q) * 4u;
      ry[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, ok ? off : l_bytes, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // dY patch, columns 0..15: combos wv + 12 i < 20
      const int combo = wv + 12 * i;
      const int row = combo >> 2, q = (combo & 3) * 4 + ql;
      const int ih = (oh0 + P.dh0 + row) * p.sh + P.ph, iw = (ow0 + P.dw0 + w16) * p.sw + P.pw;
      const bool ok = dok & (combo < 20) & ((unsigned)ih < (unsigned)d.Hg) & ((unsigned)iw < (unsigned)d.Wg) &
                      ((cg0 + 4 * q) < d.Cg);
      const uint32_t off = (uint32_t)(((id * d.Hg + ih) * d.Wg + iw) * d.ldg + cg0 + 4 * q) * 4u;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? off : g_bytes, 0, 0));
    }
    {  // dY patch, column 16: 5 rows x 16 quads on the first 80 threads
      const int row = tid >> 4, q = tid & 15;
      const int ih = (oh0 + P.dh0 + row) * p.sh + P.ph, iw = (ow0 + P.dw0 + 16) * p.sw + P.pw;
      const bool ok = dok & (tid < 80) & ((unsigned)ih < (unsigned)d.Hg) & ((unsigned)iw < (unsigned)d.Wg) &
                      ((cg0 + 4 * q) < d.Cg);
      const uint32_t off = (uint32_t)(((id * d.Hg + ih) * d.Wg + iw) * d.ldg + cg0 + 4 * q) * 4u;
      rt = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? off : g_bytes, 0, 0));
    }
  };
  auto stage = [&](int buf) {
    float* Y = smem + buf;
    float* X = smem + buf + YT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int combo = wv + 12 * i;
      const int row = combo >> 2, q = (combo & 3) * 4 + ql;
      if (combo < 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) Y[(row * 64 + 4 * q + e) * WP + w16] = ry[i][e];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int combo = wv + 12 * i;
      const int row = combo >> 2, q = (combo & 3) * 4 + ql;
      if (combo < 20) {
#pragma unroll
        for (int e = 0; e < 4; ++e) X[(row * 64 + 4 * q + e) * WP + w16] = rx[i][e];
      }
    }
    if (tid < 80) {
      const int row = tid >> 4, q = tid & 15;
#pragma unroll
      for (int e = 0; e < 4; ++e) X[(row * 64 + 4 * q + e) * WP + 16] = rt[e];
    }
  };

  f32x16 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[c][k] = 0.f;

  // one k-group = tile row g (8 tiles; this lane's 4 tiles are columns 8*half .. 8*half + 7 / + 8)
  const int zoff = (fa * 32 + col) * WP + 8 * half, voff = (fb * 32 + col) * WP + 8 * half;
  auto kgroup = [&](int buf, const int g) {
    const float* Y = smem + buf + zoff;
    const float* X = smem + buf + YT + voff;
    float y0[8], y1[8], xa[9], xb[9];
    {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(Y + (2 * g) * 64 * WP), a1 = *reinterpret_cast<const f32x4*>(Y + (2 * g) * 64 * WP + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(Y + (2 * g + 1) * 64 * WP), b1 = *reinterpret_cast<const f32x4*>(Y + (2 * g + 1) * 64 * WP + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { y0[k] = a0[k]; y0[4 + k] = a1[k]; y1[k] = b0[k]; y1[4 + k] = b1[k]; }
      const float* pa = X + (2 * g + r) * 64 * WP;
      const float* pb = X + (2 * g + 1) * 64 * WP;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(pa), c1 = *reinterpret_cast<const f32x4*>(pa + 4);
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(pb), d1 = *reinterpret_cast<const f32x4*>(pb + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { xa[k] = c0[k]; xa[4 + k] = c1[k]; xb[k] = d0[k]; xb[4 + k] = d1[k]; }
      xa[8] = pa[8];
      xb[8] = pb[8];
    }
    float zz[8], R[9];
#pragma unroll
    for (int k = 0; k < 8; ++k) zz[k] = zka * y0[k] + zkb * y1[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = xa[k] - vkb * xb[k];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(zz[2 * e], R[2 * e] - R[2 * e + 1], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(zz[2 * e] + zz[2 * e + 1], R[2 * e + 1], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(zz[2 * e + 1], R[2 * e + 2] - R[2 * e + 1], acc[2], 0, 0, 0);
    }
  };

  if (nstages > 0) {
    fetch(0);
    stage(0);
    __syncthreads();
    for (int st = 0; st < nstages; ++st) {
      const int cur = (st & 1) * STG, nxt = cur ^ STG;
      fetch(st + 1);
      kgroup(cur, 0);
      kgroup(cur, 1);
      stage(nxt);
      __syncthreads();
    }
  }

  // ---- store: slab[split][z][r*3 + c][ca][cg]
  float* slab = p.slabs + (((int64_t)split * gridDim.z + z) * 9 + r * 3) * (int64_t)p.Capad * p.Cgpad;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float* sc = slab + (int64_t)c * p.Capad * p.Cgpad;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int row = (k & 3) + 8 * (k >> 2) + 4 * half;
      sc[(int64_t)(ca0 + fa * 32 + row) * p.Cgpad + cg0 + fb * 32 + col] = acc[c][k];
    }
  }
}

// dst[ca*sa + cg*sc + tap*st] (+)= sum_{r in {a',a'+1}, c in {b',b'+1}} sum_splits slab[s][z][r*3+c][ca][cg]
__global__ void wino22_wgrad_reduce_kernel(const WW22Params p) {
  const rehr_wgrad_desc& d = p.d;
  const int KD = d.td.count, Z = p.nphase * KD;
  const int64_t total = (int64_t)Z * d.Ca * d.Cg;
  const int64_t plane = (int64_t)p.Capad * p.Cgpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % d.Cg);
    const int64_t t = i / d.Cg;
    const int ca = (int)(t % d.Ca);
    const int z = (int)(t / d.Ca);
    const int ph_i = z / KD, jd = z - ph_i * KD;
    const Phase22& P = p.phase[ph_i];
    float u[9];
#pragma unroll
    for (int x = 0; x < 9; ++x) u[x] = 0.f;
    for (int s = 0; s < p.splits; ++s) {
      const float* sp = p.slabs + (((int64_t)s * Z + z) * 9) * plane + (int64_t)ca * p.Cgpad + cg;
#pragma unroll
      for (int x = 0; x < 9; ++x) u[x] += sp[x * plane];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float v = u[a * 3 + b] + u[a * 3 + b + 1] + u[(a + 1) * 3 + b] + u[(a + 1) * 3 + b + 1];
        const int wt = ((d.td.k0 + d.td.ks * jd) * d.KH + P.kh[a]) * d.KW + P.kw[b];
        float* o = d.dst + ca * d.dst_sa + cg * d.dst_sc + wt * d.dst_st;
        *o = d.accumulate ? (*o + v) : v;
      }
  }
}

bool plan(const rehr_wgrad_desc& d, WW22Params& p) {
  if ((d.debug_flags & REHR_DBG_WGRAD_DIRECT) || d.dbias != nullptr) return false;
  if (d.sd != 1) return false;
  AxisPlan ah, aw;
  if (!plan_axis(d.th, d.sh, d.bh, ah) || !plan_axis(d.tw, d.sw, d.bw, aw)) return false;
  if (ah.nph != aw.nph) return false;
  if (d.td.count < 1 || d.td.count > 3) return false;
  if (d.Ca < 64 || d.Cg < 64 || d.Ca % 4 || d.Cg % 4) return false;
  if (d.Lh < 4 || d.Lw < 16) return false;
  p.d = d;
  p.sh = ah.stride;
  p.sw = aw.stride;
  p.nb_h = (d.Lh + RH - 1) / RH;
  p.nb_w = (d.Lw + RW - 1) / RW;
  if ((int64_t)p.nb_h * RH * p.nb_w * RW * 100 > (int64_t)d.Lh * d.Lw * 134) return false;
  const int64_t items = (int64_t)d.N * d.Ld * p.nb_h * p.nb_w;
  if (items >= (1ll << 30) || items < 4) return false;
  p.items = (int)items;
  p.a_tiles = (d.Ca + 63) / 64;
  p.c_tiles = (d.Cg + 63) / 64;
  p.Capad = p.a_tiles * 64;
  p.Cgpad = p.c_tiles * 64;
  if ((int64_t)p.a_tiles * p.c_tiles > 65535) return false;
  if ((int64_t)p.Capad * p.Cgpad * 10 > (int64_t)d.Ca * d.Cg * 14) return false;
  if ((int64_t)d.Ld * d.Lh * d.Lw * d.ldl * 4 >= (1ll << 32) - 64 ||
      (int64_t)d.Dg * d.Hg * d.Wg * d.ldg * 4 >= (1ll << 32) - 64)
    return false;
  p.nphase = ah.nph * aw.nph;
  for (int i = 0; i < ah.nph; ++i)
    for (int j = 0; j < aw.nph; ++j) {
      Phase22& P = p.phase[i * aw.nph + j];
      P.ph = ah.par[i]; P.pw = aw.par[j];
      P.dh0 = ah.dmin[i]; P.dw0 = aw.dmin[j];
      P.kh[0] = ah.kidx[i][0]; P.kh[1] = ah.kidx[i][1];
      P.kw[0] = aw.kidx[j][0]; P.kw[1] = aw.kidx[j][1];
    }
  // split count: whole rounds of 256 single-block CUs, >= 16 stages per block
  const int tiles = p.a_tiles * p.c_tiles * p.nphase * d.td.count;
  int best_s = 1;
  double best_eff = 0.0;
  for (int k = 1; k <= 4; ++k) {
    int s = (256 * k) / tiles;
    if (s < 1) s = 1;
    if ((int64_t)s * 16 > items) s = (int)(items / 16);
    if (s < 1) s = 1;
    const int64_t blocks = (int64_t)s * tiles;
    const int64_t rounds = (blocks + 255) / 256;
    const double eff = (double)blocks / (double)(rounds * 256);
    if (eff > best_eff + 0.03) { best_eff = eff; best_s = s; }
  }
  p.splits = best_s;
  p.items_per_split = (p.items + p.splits - 1) / p.splits;
  p.splits = (p.items + p.items_per_split - 1) / p.items_per_split;
  if (p.splits > 65535) return false;
  return true;
}

int64_t slab_floats(const WW22Params& p) {
  return (int64_t)p.splits * p.nphase * p.d.td.count * 9 * p.Capad * p.Cgpad;
}

}  // namespace

int64_t wino22_wgrad_workspace_bytes(const rehr_wgrad_desc& d) {
  WW22Params p;
  if (!plan(d, p)) return 0;
  return slab_floats(p) * 4 + 64;
}

int wino22_wgrad_try(const rehr_wgrad_desc& d, hipStream_t stream) {
  WW22Params p;
  if (!plan(d, p)) return REHR_ENOSUP;
  const int64_t need = slab_floats(p) * 4 + 64;
  if (!d.workspace || d.workspace_bytes < need || ((uintptr_t)d.workspace & 15)) return REHR_EINVAL;
  p.slabs = d.workspace;
  const size_t smem = (size_t)2 * STG * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wino22_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
        hipSuccess)
      return REHR_EHIP;
    attr_set = true;
  }
  dim3 grid(p.splits, p.a_tiles * p.c_tiles, p.nphase * d.td.count);
  hipLaunchKernelGGL(wino22_wgrad_kernel, grid, dim3(NTH), smem, stream, p);
  const int64_t total = (int64_t)p.nphase * d.td.count * d.Ca * d.Cg;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(wino22_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
