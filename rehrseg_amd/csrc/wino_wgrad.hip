// Winograd weight gradient for unit-stride convolutions with 3x3 taps over (H, W):
//
//   dU[jd][xi][co][ci] = sum_{n, od, 2x2 output tiles}  (A dY A^T)[xi][tile][co] * (B^T x B)[xi][tile][ci]
//   dW[kd][a][b]       = sum_{r,c} G[r][a] G[c][b] dU[jd][(r,c)]            (reduce kernel, fixed order)
//
// 16 products per tile and depth tap instead of 36: 2.25x fewer MFMA k-steps than the direct
// weight gradient, which (fp32 matrix pipe bound, ~105 TF) was the largest item of a step.
//
//   block   = 64 co x 64 ci of ONE depth tap, a contiguous range of (sample, depth, 4x16-output
//             region) items (split-K, slabs reduced deterministically afterwards)
//   wave r  = Winograd row r: 4 columns x (2 x 2) 32x32 tiles = 16 accumulator tiles in AGPRs
//             (one wave per SIMD, full register file -- all latency hiding is explicit)
//   stage   = one region: dY (4 x 16 voxels) and the x patch (6 x 18) in LDS, TRANSPOSED while
//             staging to [row][channel][column] (scalar stores, 16 columns x 4 channel quads per
//             wave instruction = 64 distinct banks).  The MFMA k index is the TILE, so a lane
//             (= channel) needs 8 / 10 consecutive columns per row: two or three 128/64-bit reads
//             instead of 8 / 10 scalar ones (20 LDS reads per k-group instead of 72).
//   k-group = 8 tiles (one tile row) = 4 quarters (fa, fb) of 16 MFMAs = 16 steps of 4 MFMAs.  The loop
//             body is ONE basic block (counter-walked items, predicated loads, a spare LDS pad for the
//             threads without a share of the last staging piece): fetch and staging run in the shadow
//             of the MFMAs instead of between them, laid out per step (see kgroup).
#include "common.h"
#include "wgrad_shared.h"
#include <cstdlib>
#include <type_traits>

namespace {

#ifndef WW_W8_REGION
#define WW_W8_REGION 1   // 8-wave mode: keep the per-quarter scheduling regions (0: one region per k-group; measured below)
#endif

constexpr int RH = 4, RW = 16;           // outputs per staged region = 2 x 8 tiles
constexpr int XH = RH + 2, XW = RW + 2;  // input patch
constexpr int YV = RH * RW, XV = XH * XW;

struct WWParams {
  rehr_wgrad_desc d;
  int nb_h, nb_w;
  int items, items_per_split, splits;
  int a_tiles, c_tiles, Capad, Cgpad;
  int fa, fb;
  bool w8;       // 64 x 64 block of 8 waves (two wave sets)
  int colocate;  // 1-D (split x tap) grid with the taps of a split on one XCD
  // narrow planes (Lw < 16, e.g. the 12x12 layers of the reference's 96x96 crops): G consecutive (sample, depth)
  // slices are laid side by side, each with a 1-column zero gutter on either side, into a virtual lattice of
  // width G*(Lw+2) -- a pure index map in the fetch; dY is zero in the gutters, so whatever the transform
  // multiplies it with drops out.  G = 0: plain lattice.
  int G, Lw2, nslices, rcp;
  uint32_t rcp_ld;   // ceil(2^32 / Ld), or ceil(2^32 / N) when the slices are numbered depth-major (skip)
  // skip: a depth tap walks only the output slices whose source slice exists (od + offset inside the volume) instead of
  // multiplying zeros for the others -- 1/6 of the products at depth 4 with 3 taps.  The plain lattice walks od in
  // [od_lo, od_hi); the side-by-side mode numbers its slices depth-major (od * N + n, N % G == 0) so that whole groups
  // drop out.  bias_jd = the tap whose walk sees every slice (the bias gradient is a by-product of ONE tap's walk).
  int skip, bias_jd;
  float* slabs;      // [splits][KD][16][Capad][Cgpad]
  float* slab_bias;  // [splits][Capad] or null: per-split column sums of dY (bias gradient)
};

// FA / FB = 32-channel groups of dY / x per block (2 x 2: one block per CU with the full register
// file; smaller shapes for 32-channel layers run several blocks per CU).
// WS = wave sets per block: 1 (4 waves), or 2 (8 waves, two per SIMD: set ws owns the dY groups ws*FA .. ws*FA+FA-1 and
// all FB x groups; both sets read the same staged region, so nothing is staged twice -- unlike two co-resident blocks)
template <int FA, int FB, int WS = 1>
struct WWCfg {
  // LDS holds both operands TRANSPOSED, [row][channel][column] with a 20-float column pitch: a lane
  // (= channel) reads its 8 / 10 consecutive columns as 128-bit words (5 groups of 4 banks between
  // neighbouring channels: conflict-free), 20 reads per k-group instead of 72 scalar ones
  static constexpr int WP = 20;
  static constexpr int CA = FA * WS * 32, CG = FB * 32;
  static constexpr int YBUF = RH * CA * WP, XBUF = XH * CG * WP, BUF = YBUF + XBUF;
  // staging pieces per thread: a wave instruction covers 16 columns x 4 channel quads of one (row, quad
  // group) -> 64 distinct banks for each of the 4 scalar stores of a piece
  static constexpr int NPY = 2 * FA, NPXM = 3 * FB / WS, NP = NPY + NPXM + 1;
  static_assert((3 * FB) % WS == 0, "x patch pieces must divide over the waves");
  static constexpr int NPA = (NP + 1) / 2;
  static constexpr int NQ = FA * FB, NPREP = FA + FB;
  static constexpr int MINB = (WS == 2) ? 1 : ((FA * FB == 4) ? 1 : ((FA * FB == 2) ? 1 : 2));
};

// MINB_ = resident blocks per CU the register budget is cut for: the 64 x 32 shape also runs two blocks per CU (two
// waves per SIMD, 256 registers each) for the 64 x 64-and-wider layers -- see plan()
template <int FA, int FB, bool VIRT, int MINB_ = WWCfg<FA, FB>::MINB, int WS = 1>
__global__ __launch_bounds__(256 * WS, MINB_) void wino_wgrad_kernel(const WWParams p) {
  using C = WWCfg<FA, FB, WS>;
  constexpr int WP = C::WP, CA = C::CA, CG = C::CG, YBUF = C::YBUF, BUF = C::BUF;
  constexpr int NPY = C::NPY, NPXM = C::NPXM, NP = C::NP, NPA = C::NPA;
  const rehr_wgrad_desc& d = p.d;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = wv8 & 3, ws = wv8 >> 2;   // Winograd row, wave set
  const int half = lane >> 5, col = lane & 31;
  // colocate (default): the depth taps of one split are consecutive blocks of ONE XCD (1-D grid over split x tap):
  // they walk the same (sample, slice, region) items at the same time, so the dY regions -- and the x slices, which a
  // tap reads one slice apart from its neighbour -- come from that XCD's L2 for two of the three taps.  With
  // blockIdx.z = tap the taps of a split sat on different XCDs and every one fetched its own copy (32 x 32 channels at
  // 2 x 128^3: 3.99 GB per launch for 1.07 GB of operands, 3.7 TB/s at the memory side).
  int split = blockIdx.x, jd = blockIdx.z;
  if (p.colocate) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    jd = __builtin_amdgcn_readfirstlane(j % d.td.count);
    split = __builtin_amdgcn_readfirstlane((j / d.td.count) * 8 + xcd);
    if (split >= p.splits) return;   // (block-uniform padding block)
  }
  const int at = blockIdx.y / p.c_tiles, ct = blockIdx.y - at * p.c_tiles;
  const int ca0 = at * (FA * WS * 32), cg0 = ct * (FB * 32);
  const int doff = d.bd + d.td.off0 + d.td.offs * jd;
  // output slices this tap reaches, and its share of the split-K walk
  const int od_lo = p.skip ? max(0, -doff) : 0, od_hi = p.skip ? min(d.Ld, d.Dg - doff) : d.Ld;
  const int nod = max(0, od_hi - od_lo);
  int items_tap, g_lo = 0;
  if constexpr (VIRT) {
    g_lo = p.skip ? (od_lo * d.N) / p.G : 0;
    const int g_hi = p.skip ? (od_hi * d.N) / p.G : (p.nslices + p.G - 1) / p.G;
    items_tap = max(0, g_hi - g_lo) * p.nb_h * p.nb_w;
  } else {
    items_tap = d.N * nod * p.nb_h * p.nb_w;
  }
  const int ips = p.skip ? (items_tap + p.splits - 1) / p.splits : p.items_per_split;
  const int it0 = split * ips;
  const int it1 = min(it0 + ips, p.skip ? items_tap : p.items);
  const int nstages = max(0, it1 - it0);

  // Z rows (A dY): (y0, y0+y1, y0-y1, -y1) -> ka*y0 + kb*y1, row 3 folded negated
  const float zka = (r == 3) ? 0.f : 1.f, zkb = (r == 0) ? 0.f : ((r == 2) ? -1.f : 1.f);
  // V rows (B^T x): (x0-x2, x1+x2, x2-x1 [negated], x1-x3) -> x[i1] + s2 * x[i2]
  const int i1 = (r == 0) ? 0 : 1, i2 = (r == 3) ? 3 : 2;
  const float s2 = (r == 1) ? 1.f : -1.f;

  // ---- staging pieces: lane = (column w16, quad ql) of the wave's (row, quad group) combo
  const int wv = wv8;  // the wave index is the staging wave id
  constexpr int QY = 2 * FA * WS, QX = 2 * FB, WSTEP = 4 * WS;   // 16-channel groups per dY / x row; waves per block
  const int w16 = lane & 15, ql = lane >> 4;
  const int64_t l_img = (int64_t)d.Ld * d.Lh * d.Lw * d.ldl, g_img = (int64_t)d.Dg * d.Hg * d.Wg * d.ldg;
  const uint32_t l_bytes = (uint32_t)(l_img * 4), g_bytes = (uint32_t)(g_img * 4);
  f32x4 rx[NPA];
  // piece i -> (kind, row, column, quad): 0 = dY tile, 1 = x patch
  auto piece = [&](const int i, int& kind, int& row, int& cw, int& q) {
    if (i < NPY) {
      const int combo = wv + WSTEP * i;
      kind = 0; row = combo / QY; q = (combo % QY) * 4 + ql; cw = w16;
    } else if (i < NPY + NPXM) {
      const int combo = wv + WSTEP * (i - NPY);
      kind = 1; row = combo / QX; q = (combo % QX) * 4 + ql; cw = w16;
    } else {  // patch columns 16, 17: 6 rows x 2 columns x 8*FB quads on the first 96*FB threads
      const int rest = tid / (8 * FB);
      kind = (tid < 96 * FB) ? 1 : 2; row = rest >> 1; cw = 16 + (rest & 1); q = tid % (8 * FB);
    }
  };
  // The staged item (region bw_, bh_ of depth slice od of sample n -- or of slice group grp in the
  // side-by-side mode) is walked with counters: no integer division and no branch inside the loop, so
  // that the address arithmetic and the loads are ordinary instructions of the MFMA region they sit in.
  int f_bw, f_bh, f_od, f_n, f_left = nstages;
  {
    int it = it0;
    f_bw = it % p.nb_w; it /= p.nb_w;
    f_bh = it % p.nb_h; it /= p.nb_h;
    if constexpr (VIRT) { f_od = 0; f_n = g_lo + it; } else { const int nd = max(nod, 1); f_od = od_lo + it % nd; f_n = it / nd; }
  }
  auto advance = [&]() {
    --f_left;
    const bool w_end = (f_bw + 1 == p.nb_w);
    f_bw = w_end ? 0 : f_bw + 1;
    const bool h_end = w_end & (f_bh + 1 == p.nb_h);
    f_bh = w_end ? (h_end ? 0 : f_bh + 1) : f_bh;
    if constexpr (VIRT) {
      f_n += h_end ? 1 : 0;
    } else {
      const bool d_end = h_end & (f_od + 1 == od_hi);
      f_od = h_end ? (d_end ? od_lo : f_od + 1) : f_od;
      f_n += d_end ? 1 : 0;
    }
  };
  auto fetch = [&](const int lo, const int hi, const int base) {  // pieces [lo, hi) of the cursor's item -> rx[i - base]
    const bool live = f_left > 0;
    const int oh0 = f_bh * RH, ow0 = f_bw * RW;
    if constexpr (!VIRT) {
      const int od = f_od, n = live ? f_n : 0;
      const int id = od + doff;
      const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(d.l) + (int64_t)n * l_img, 0, l_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(d.g) + (int64_t)n * g_img, 0, g_bytes, 0x00020000);
      const bool dok = live & ((unsigned)id < (unsigned)d.Dg);
#pragma unroll
      for (int i = lo; i < hi; ++i) {
        int kind, row, cw, q;
        piece(i, kind, row, cw, q);
        if (i < NPY) {
          const int gh = oh0 + row, gw = ow0 + cw;
          const bool ok = live & ((ca0 + 4 * q) < d.Ca) & (gh < d.Lh) & (gw < d.Lw);
          const uint32_t off = (uint32_t)(((od * d.Lh + gh) * d.Lw + gw) * d.ldl + ca0 + 4 * q) * 4u;
          rx[i - base] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, max(off, ok ? 0u : l_bytes), 0, 0));
        } else {
          const int ih = oh0 - 1 + row, iw = ow0 - 1 + cw;
          const bool ok = dok & (kind == 1) & ((cg0 + 4 * q) < d.Cg) & ((unsigned)ih < (unsigned)d.Hg) &
                          ((unsigned)iw < (unsigned)d.Wg);
          const uint32_t off = (uint32_t)(((id * d.Hg + ih) * d.Wg + iw) * d.ldg + cg0 + 4 * q) * 4u;
          rx[i - base] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, max(off, ok ? 0u : g_bytes), 0, 0));
        }
      }
    } else {
      // virtual wide lattice: column cv -> slice grp*G + cv / (Lw+2), plane column cv % (Lw+2) - 1
      const int sg = f_n * p.G;
      const uint32_t lt = (uint32_t)d.N * l_bytes, gt = (uint32_t)d.N * g_bytes;  // whole tensors (< 4 GiB, planner)
      const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.l), 0, lt, 0x00020000);
      const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.g), 0, gt, 0x00020000);
#pragma unroll
      for (int i = lo; i < hi; ++i) {
        int kind, row, cw, q;
        piece(i, kind, row, cw, q);
        const bool isy = i < NPY;
        const int cv = isy ? ow0 + cw : ow0 - 1 + cw;
        const int cvc = cv < 0 ? 0 : cv;
        const int s_ = (cvc * p.rcp) >> 16;                 // cv / (Lw+2) for cv < 4096
        const int c = cvc - s_ * p.Lw2 - 1;
        const int slice = sg + s_;
        // slice / Ld: multiply-high by ceil(2^32 / Ld), exact below 65536 slices (planner); Ld = 1 has no 32-bit reciprocal
        int n, od;
        if (p.skip) {   // depth-major numbering: od * N + n
          od = (d.N == 1) ? slice : (int)__umulhi((uint32_t)slice, p.rcp_ld);
          n = slice - od * d.N;
        } else {
          n = (d.Ld == 1) ? slice : (int)__umulhi((uint32_t)slice, p.rcp_ld);
          od = slice - n * d.Ld;
        }
        const bool sok = live & (cv >= 0) & (s_ < p.G) & (slice < p.nslices) & ((unsigned)c < (unsigned)d.Lw);
        if (isy) {
          const int gh = oh0 + row;
          const bool ok = sok & ((ca0 + 4 * q) < d.Ca) & (gh < d.Lh);
          const uint32_t off = (uint32_t)((((n * d.Ld + od) * d.Lh + gh) * d.Lw + c) * d.ldl + ca0 + 4 * q) * 4u;
          rx[i - base] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, max(off, ok ? 0u : lt), 0, 0));
        } else {
          const int ih = oh0 - 1 + row, id = od + doff;
          const bool ok = sok & (kind == 1) & ((cg0 + 4 * q) < d.Cg) & ((unsigned)ih < (unsigned)d.Hg) &
                          ((unsigned)id < (unsigned)d.Dg);
          const uint32_t off = (uint32_t)((((n * d.Dg + id) * d.Hg + ih) * d.Wg + c) * d.ldg + cg0 + 4 * q) * 4u;
          rx[i - base] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, max(off, ok ? 0u : gt), 0, 0));
        }
      }
    }
  };
  // threads without a share of the last piece store to a spare pad behind the two buffers: no branch
  float* const spare = smem + 2 * BUF + lane;
  auto stage = [&](int buf, const int lo, const int hi, const int base) {
#pragma unroll
    for (int i = lo; i < hi; ++i) {
      int kind, row, cw, q;
      piece(i, kind, row, cw, q);
      float* dst;
      if (i < NPY) dst = smem + buf + (row * CA + 4 * q) * WP + cw;
      else dst = (kind == 1) ? smem + buf + YBUF + (row * CG + 4 * q) * WP + cw : spare;
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[e * WP] = rx[i - base][e];
    }
  };

  float bsum[FA];
#pragma unroll
  for (int fa = 0; fa < FA; ++fa) bsum[fa] = 0.f;
  // ---- operand preparation for one k-group (tile row g of the region)
  // Z[c][e] for 32 co: tiles 4*half + e, dY rows 2g, 2g+1, columns 8*half .. 8*half + 7
  const float* ybase = smem + col * WP + 8 * half;
  const float* xbase = smem + YBUF + col * WP + 8 * half;
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  struct Raw { f32x4 a0, a1, b0, b1; f32x2_ a2, b2; };  // the LDS words of one operand set, untransformed
  auto read_z = [&](int buf, const int g, const int fa, Raw& w) {
    const float* y0 = ybase + buf + ((2 * g) * CA + (ws * FA + fa) * 32) * WP;
    const float* y1 = y0 + CA * WP;
    w.a0 = *reinterpret_cast<const f32x4*>(y0); w.a1 = *reinterpret_cast<const f32x4*>(y0 + 4);
    w.b0 = *reinterpret_cast<const f32x4*>(y1); w.b1 = *reinterpret_cast<const f32x4*>(y1 + 4);
  };
  auto xform_z = [&](const int fa, const Raw& w, f32x4 (&Z)[4]) {
    float z[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      z[k] = zka * w.a0[k] + zkb * w.b0[k];
      z[4 + k] = zka * w.a1[k] + zkb * w.b1[k];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      Z[0][e] = z[2 * e];
      Z[1][e] = z[2 * e] + z[2 * e + 1];
      Z[2][e] = z[2 * e] - z[2 * e + 1];
      Z[3][e] = z[2 * e + 1];  // negated column, undone at the store
    }
    // (row 1, column 1) of A dY A^T is the plain sum of the 2x2 tile: the bias gradient for free
    bsum[fa] += (Z[1][0] + Z[1][1]) + (Z[1][2] + Z[1][3]);
  };
  // V[c][e] for 32 ci: patch rows 2g + i1, 2g + i2, columns 8*half .. 8*half + 9
  auto read_v = [&](int buf, const int g, const int fb, Raw& w) {
    const float* xa = xbase + buf + ((2 * g + i1) * CG + fb * 32) * WP;
    const float* xb = xbase + buf + ((2 * g + i2) * CG + fb * 32) * WP;
    w.a0 = *reinterpret_cast<const f32x4*>(xa); w.a1 = *reinterpret_cast<const f32x4*>(xa + 4);
    w.a2 = *reinterpret_cast<const f32x2_*>(xa + 8);
    w.b0 = *reinterpret_cast<const f32x4*>(xb); w.b1 = *reinterpret_cast<const f32x4*>(xb + 4);
    w.b2 = *reinterpret_cast<const f32x2_*>(xb + 8);
  };
  auto xform_v = [&](const Raw& w, f32x4 (&V)[4]) {
    float R[10];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      R[k] = w.a0[k] + s2 * w.b0[k];
      R[4 + k] = w.a1[k] + s2 * w.b1[k];
    }
    R[8] = w.a2[0] + s2 * w.b2[0];
    R[9] = w.a2[1] + s2 * w.b2[1];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      V[0][e] = R[2 * e] - R[2 * e + 2];
      V[1][e] = R[2 * e + 1] + R[2 * e + 2];
      V[2][e] = R[2 * e + 1] - R[2 * e + 2];  // negated column, undone at the store
      V[3][e] = R[2 * e + 1] - R[2 * e + 3];
    }
  };
  // the FA + FB operand sets of a k-group, in the order (Z0, V0, Z1, V1) of their first use
  auto set_is_z = [](const int j) {
    if constexpr (FA == 2 && FB == 2) return (j & 1) == 0;
    else if constexpr (FA == 1) return j == 0;
    else return j != 1;
  };
  auto set_idx = [](const int j) {
    if constexpr (FA == 1 && FB == 2) return j == 0 ? 0 : j - 1;
    else return j >> 1;
  };
  auto prep_read = [&](const int j, int buf, const int g, Raw& w) {
    if (set_is_z(j)) read_z(buf, g, set_idx(j), w); else read_v(buf, g, set_idx(j), w);
  };
  auto prep_xform = [&](const int j, const Raw& w, f32x4 (&ZS)[FA][4], f32x4 (&VS)[FB][4]) {
    if (set_is_z(j)) xform_z(set_idx(j), w, ZS[set_idx(j)]); else xform_v(w, VS[set_idx(j)]);
  };

  f32x16 acc[FA][FB][4];
#pragma unroll
  for (int fa = 0; fa < FA; ++fa)
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[fa][fb][c][k] = 0.f;
  // A k-group = NS steps of 4 MFMAs (one tile pair e of one quarter); a quarter (4 steps) is one scheduling region
  // and the accumulator anchors below keep every step's MFMAs in place inside it.  With one
  // wave per SIMD nothing hides a wait but the wave's own MFMAs in flight, so the work between them is laid out
  // by hand: the LDS reads of an operand set are issued one step before its transform (measured: 1 step / quarter
  // regions 4-5 % faster than 2 steps / per-step regions; whole-k-group regions slower again), the LDS stores of the
  // pieces fetched during the previous k-group go into the first half of the steps, their refill (address
  // arithmetic + global loads) into the second half, the cursor moves at the end.  Everything is branch-free.
  constexpr int NS = 4 * C::NQ, SPC = 1, NPREP = C::NPREP;
  auto kgroup = [&](f32x4 (&ZS)[FA][4], f32x4 (&VS)[FB][4], int pbuf, const int pg, f32x4 (&ZN)[FA][4],
                    f32x4 (&VN)[FB][4], int sbuf, auto second_) {
    // first k-group of a stage: stores the second half of the pieces, refills the first half; second: vice versa
    constexpr bool second = decltype(second_)::value;
    constexpr int slo = second ? 0 : NPA, ns = second ? NPA : NP - NPA;
    constexpr int flo = second ? NPA : 0, nf = second ? NP - NPA : NPA;
    Raw raw[2];
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      if ((s_ & 3) == 0 && (WS == 1 || WW_W8_REGION)) __builtin_amdgcn_sched_barrier(0);  // one scheduling region per quarter (16 MFMAs)
#pragma unroll
      for (int j = 0; j < NPREP; ++j) {
        if (s_ == (j + 1) * SPC) prep_xform(j, raw[j & 1], ZN, VN);
        if (s_ == j * SPC) prep_read(j, pbuf, pg, raw[j & 1]);
      }
#pragma unroll
      for (int k = 0; k < ns; ++k)
        if (s_ == k * (NS / 2) / ns) stage(sbuf, slo + k, slo + k + 1, slo);
#pragma unroll
      for (int k = 0; k < nf; ++k)
        if (s_ == NS / 2 + k * (NS / 2) / nf) fetch(flo + k, flo + k + 1, flo);
      if (second && s_ == NS - 1) advance();
      const int qi = s_ >> 2, e = s_ & 3;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[qi / FB][qi % FB][c] =
            __builtin_amdgcn_mfma_f32_32x32x2f32(ZS[qi / FB][c][e], VS[qi % FB][c][e], acc[qi / FB][qi % FB][c], 0, 0, 0);
      // empty anchor on the four accumulators: instruction selection may not let the MFMAs drift out of their
      // step (measured: -2 %; anchoring the transforms or the address arithmetic the same way measured worse).
      // The two-blocks-per-CU shape has a second wave to fill gaps and runs better without it.
      if constexpr (C::MINB == 1 && WS == 1)
      asm volatile("" : "+a"(acc[qi / FB][qi % FB][0]), "+a"(acc[qi / FB][qi % FB][1]), "+a"(acc[qi / FB][qi % FB][2]),
                        "+a"(acc[qi / FB][qi % FB][3]));
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using std::integral_constant;

  f32x4 ZP[FA][4], VP[FB][4], ZQ[FA][4], VQ[FB][4];
  if (nstages > 0) {
    // prologue: stage 0 complete in buffer 0, first half of stage 1 in buffer 1, second half in flight
    fetch(0, NPA, 0);
    stage(0, 0, NPA, 0);
    fetch(NPA, NP, NPA);
    stage(0, NPA, NP, NPA);
    advance();
    fetch(0, NPA, 0);
    stage(BUF, 0, NPA, 0);
    fetch(NPA, NP, NPA);
    advance();
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NPREP; ++j) {
      Raw w;
      prep_read(j, 0, 0, w);
      prep_xform(j, w, ZP, VP);
    }

#ifdef WW_NO_PRIO   // (A/B builds)
#define WW_PRIO(n) (void)0
#else
#define WW_PRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
    for (int st = 0; st < nstages; ++st) {
      const int cur = (st & 1) * BUF, nxt = cur ^ BUF;
      // tile row 0 with set P; set Q <- tile row 1 of this stage.  Staging: second half of stage
      // st+1 lands in nxt (free since the previous midpoint), first half of st+2 is fetched.
      WW_PRIO(1);   // (issue priority falls with progress since the barrier: the arbiter serves the wave that is behind)
      kgroup(ZP, VP, cur, 1, ZQ, VQ, nxt, integral_constant<bool, false>{});
      // midpoint: nobody reads `cur` any more, stage st+1 is complete in `nxt`
      __syncthreads();
      WW_PRIO(2);
      kgroup(ZQ, VQ, nxt, 0, ZP, VP, cur, integral_constant<bool, true>{});
    }
    WW_PRIO(0);
  }

  if (p.slab_bias != nullptr && r == 1 && jd == p.bias_jd && ct == 0) {
#pragma unroll
    for (int fa = 0; fa < FA; ++fa) {
      const float t = bsum[fa] + __shfl_xor(bsum[fa], 32, 64);
      if (half == 0) p.slab_bias[(int64_t)split * p.Capad + ca0 + (ws * FA + fa) * 32 + col] = t;
    }
  }
  // ---- store the tiles of this wave: slab[split][jd][r*4+c][co][ci], signs of the folded
  // negations taken back (Z row 3, V row 2, Z column 3, V column 2)
  const float rs = ((r == 3) ? -1.f : 1.f) * ((r == 2) ? -1.f : 1.f);
  float* slab = p.slabs + (((int64_t)split * d.td.count + jd) * 16 + r * 4) * (int64_t)p.Capad * p.Cgpad;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float sgn = (c >= 2) ? -rs : rs;
    float* sc = slab + (int64_t)c * p.Capad * p.Cgpad;
#pragma unroll
    for (int fa = 0; fa < FA; ++fa)
#pragma unroll
      for (int fb = 0; fb < FB; ++fb)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int row = (k & 3) + 8 * (k >> 2) + 4 * half;
          sc[(int64_t)(ca0 + (ws * FA + fa) * 32 + row) * p.Cgpad + cg0 + fb * 32 + col] = sgn * acc[fa][fb][c][k];
        }
  }
}

// dst[co*sa + ci*sc + tap*st] (+)= sum_{r,c} G[r][a] G[c][b] sum_splits slab[s][jd][r*4+c][co][ci]
// NG threads share an output element on the layers with few channels and many slabs (thread (e, g) sums the slabs
// g, g + NG, ...; the partial sums are combined in a fixed order through LDS): one thread per element walking all slabs
// was a latency-bound chain there (49 us per launch on average in cfg-2).
template <int NG>
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const WWParams p) {
  constexpr int EPB = 256 / NG;
  __shared__ float part[NG > 1 ? 16 * 256 : 1];
  const rehr_wgrad_desc& d = p.d;
  const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  const int KD = d.td.count;
  const int64_t total = (int64_t)KD * d.Ca * d.Cg;
  const int64_t plane = (int64_t)p.Capad * p.Cgpad;
  const int el = threadIdx.x % EPB, g = threadIdx.x / EPB;
  for (int64_t i0 = (int64_t)blockIdx.x * EPB; i0 < total; i0 += (int64_t)gridDim.x * EPB) {
    const int64_t i = i0 + el;
    const bool ok = i < total;
    const int ci = ok ? (int)(i % d.Cg) : 0;
    const int64_t t = ok ? i / d.Cg : 0;
    const int co = (int)(t % d.Ca);
    const int jd = (int)(t / d.Ca);
    float u[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) u[x] = 0.f;
    if (ok) {
      for (int s = g; s < p.splits; s += NG) {
        const float* sp = p.slabs + (((int64_t)s * KD + jd) * 16) * plane + (int64_t)co * p.Cgpad + ci;
#pragma unroll
        for (int x = 0; x < 16; ++x) u[x] += sp[x * plane];
      }
    }
    if (NG > 1) {
      __syncthreads();
#pragma unroll
      for (int x = 0; x < 16; ++x) part[x * 256 + threadIdx.x] = u[x];
      __syncthreads();
      if (g == 0) {
#pragma unroll
        for (int q = 1; q < NG; ++q)
#pragma unroll
          for (int x = 0; x < 16; ++x) u[x] += part[x * 256 + q * EPB + el];
      }
    }
    if (!ok || g != 0) continue;
    float tmp[3][4];  // G^T u : [a][c]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        tmp[a][c] = G[0][a] * u[c] + G[1][a] * u[4 + c] + G[2][a] * u[8 + c] + G[3][a] * u[12 + c];
#pragma unroll
    for (int jh = 0; jh < 3; ++jh)
#pragma unroll
      for (int jw = 0; jw < 3; ++jw) {
        const int a = d.bh + d.th.off0 + d.th.offs * jh + 1, b = d.bw + d.tw.off0 + d.tw.offs * jw + 1;
        float v = 0.f;
#pragma unroll
        for (int aa = 0; aa < 3; ++aa)
#pragma unroll
          for (int bb = 0; bb < 3; ++bb)
            if (aa == a && bb == b)
              v = tmp[aa][0] * G[0][bb] + tmp[aa][1] * G[1][bb] + tmp[aa][2] * G[2][bb] + tmp[aa][3] * G[3][bb];
        const int wt = ((d.td.k0 + d.td.ks * jd) * d.KH + (d.th.k0 + d.th.ks * jh)) * d.KW + (d.tw.k0 + d.tw.ks * jw);
        float* o = d.dst + co * d.dst_sa + ci * d.dst_sc + wt * d.dst_st;
        *o = d.accumulate ? (*o + v) : v;
      }
  }
  if (p.slab_bias != nullptr && d.dbias != nullptr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.Ca; i += (int64_t)gridDim.x * blockDim.x) {
      float sum = 0.f;
      for (int k = 0; k < p.splits; ++k) sum += p.slab_bias[(int64_t)k * p.Capad + i];
      d.dbias[i] = d.accumulate ? (d.dbias[i] + sum) : sum;
    }
  }
}

template <int FA, int FB, bool VIRT, int MINB_ = WWCfg<FA, FB>::MINB, int WS = 1>
int launch_ww(const WWParams& p, dim3 grid, hipStream_t stream) {
  const size_t smem = ((size_t)2 * WWCfg<FA, FB, WS>::BUF + 128) * sizeof(float);  // + the spare pad of stage()
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wino_wgrad_kernel<FA, FB, VIRT, MINB_, WS>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return REHR_EHIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((wino_wgrad_kernel<FA, FB, VIRT, MINB_, WS>), grid, dim3(256 * WS), smem, stream, p);
  return REHR_OK;
}

bool three_taps_w(const rehr_axis_taps& t, int b) {
  if (t.count != 3) return false;
  const int o0 = b + t.off0, o1 = b + t.off0 + t.offs, o2 = b + t.off0 + 2 * t.offs;
  return (o1 == 0) && ((o0 == -1 && o2 == 1) || (o0 == 1 && o2 == -1));
}

bool plan(const rehr_wgrad_desc& d, WWParams& p) {
  if (d.debug_flags & REHR_DBG_WGRAD_DIRECT) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return false;
  if (!three_taps_w(d.th, d.bh) || !three_taps_w(d.tw, d.bw)) return false;
  if (d.td.count < 1 || d.td.count > 256) return false;  // any number of depth taps (feature_fuse: 128): one grid.z each
  if (d.Hg != d.Lh || d.Wg != d.Lw) return false;
  if (d.Ca < 16 || d.Cg < 16 || d.Ca % 4 || d.Cg % 4) return false;
  if (d.Lh < 4 || d.Lw < 6) return false;
  p.d = d;
  p.nb_h = (d.Lh + RH - 1) / RH;
  p.G = 0;
  int64_t items;
  if (d.Lw >= 16) {
    p.nb_w = (d.Lw + RW - 1) / RW;
    if ((int64_t)p.nb_h * RH * p.nb_w * RW * 100 > (int64_t)d.Lh * d.Lw * 134) return false;  // 24 x 24 (1.333) is in
    items = (int64_t)d.N * d.Ld * p.nb_h * p.nb_w;
  } else {
    // narrow planes: 8 slices side by side (see WWParams::G); whole-tensor buffer offsets must fit 32 bits
    p.G = 8;
    p.Lw2 = d.Lw + 2;
    p.rcp = (65536 + p.Lw2 - 1) / p.Lw2;
    p.nslices = d.N * d.Ld;
    if (p.nslices >= 65536) return false;
    p.rcp_ld = d.Ld > 1 ? (uint32_t)((0x100000000ull + d.Ld - 1) / d.Ld) : 0u;
    const int64_t groups = ((int64_t)p.nslices + p.G - 1) / p.G;
    p.nb_w = (p.G * p.Lw2 + RW - 1) / RW;
    if ((int64_t)groups * p.nb_h * RH * p.nb_w * RW * 100 > (int64_t)p.nslices * d.Lh * d.Lw * 140) return false;
    if ((int64_t)d.N * d.Ld * d.Lh * d.Lw * d.ldl * 4 >= (1ll << 32) - 64 ||
        (int64_t)d.N * d.Dg * d.Hg * d.Wg * d.ldg * 4 >= (1ll << 32) - 64)
      return false;
    items = groups * p.nb_h * p.nb_w;
    if (items < 128) return false;  // too few stages to split over the chip: the slab kernel's finer split-K wins
  }
  if (items >= (1ll << 30) || items < 4) return false;
  p.items = (int)items;
  // depth-tap skipping needs one tap that reaches every output slice (it carries the bias gradient)
  p.skip = 0;
  p.bias_jd = 0;
  for (int j = 0; j < d.td.count && !(d.debug_flags & REHR_DBG_WGRAD_NO_TAP_SKIP); ++j) {
    const int dd = d.bd + d.td.off0 + d.td.offs * j;
    if (dd >= 0 && d.Dg - dd >= d.Ld) { p.skip = 1; p.bias_jd = j; break; }   // od_lo == 0 and od_hi == Ld
  }
  if (p.G != 0) {
    if (p.skip && d.N % p.G == 0) p.rcp_ld = d.N > 1 ? (uint32_t)((0x100000000ull + d.N - 1) / d.N) : 0u;
    else { p.skip = 0; p.bias_jd = 0; }
  }
  p.fa = (d.Ca <= 32) ? 1 : 2;  // 32-channel sides take a single 32-wide group
  p.fb = (d.Cg <= 32) ? 1 : 2;
  // the 64 x 64 block runs as two wave sets of 32 x 64 (8 waves, two per SIMD, one staged region).  Tried and removed
  // in round 3 (profiles/r03_ab_superseded.txt): two independent 64 x 32 blocks per CU (+1.9 ms per cfg-2 step: dY staged
  // twice, 20 spilled registers) and the 4-wave 64 x 64 block (equal within noise).
  p.w8 = p.fa == 2 && p.fb == 2;
  p.a_tiles = (d.Ca + p.fa * 32 - 1) / (p.fa * 32);
  p.c_tiles = (d.Cg + p.fb * 32 - 1) / (p.fb * 32);
  p.Capad = p.a_tiles * p.fa * 32;
  p.Cgpad = p.c_tiles * p.fb * 32;
  if ((int64_t)p.a_tiles * p.c_tiles > 65535) return false;
  // channel padding waste (e.g. 96 -> 128) must not eat the gain; a 16-channel side padded to one
  // 32-wide group still beats the direct kernels by a wide margin (SR head: 8.2 ms -> see DESIGN)
  const int64_t limit = (p.fa * p.fb == 1) ? 21 : 14;
  if ((int64_t)p.Capad * p.Cgpad * 10 > (int64_t)d.Ca * d.Cg * limit) return false;
  if ((int64_t)d.Ld * d.Lh * d.Lw * d.ldl * 4 >= (1ll << 32) - 64 ||
      (int64_t)d.Dg * d.Hg * d.Wg * d.ldg * 4 >= (1ll << 32) - 64)
    return false;
  // split count: fill whole rounds of 256 single-block CUs, >= 16 stages per block
  const int tiles = p.a_tiles * p.c_tiles * d.td.count;
  const int slots = 256 * ((p.fa * p.fb == 1) ? 2 : 1);  // resident blocks
  int best_s = 1;
  double best_eff = 0.0;
  for (int k = 1; k <= 4; ++k) {
    int s = (slots * k) / tiles;
    if (s < 1) s = 1;
    if ((int64_t)s * 16 > items) s = (int)(items / 16);
    if (s < 1) s = 1;
    const int64_t blocks = (int64_t)s * tiles;
    const int64_t rounds = (blocks + slots - 1) / slots;
    const double eff = (double)blocks / (double)(rounds * slots);
    if (eff > best_eff + 0.03) { best_eff = eff; best_s = s; }
  }
  // many splits (few channel tiles, e.g. 32 x 32 channels: 170): a multiple of 8, so that the tap-colocated grid (the
  // taps of a split on one XCD, see the kernel) gives every XCD the same number of blocks
  if (best_s >= 64 && d.td.count > 1) best_s &= ~7;
  p.splits = best_s;
  p.items_per_split = (p.items + p.splits - 1) / p.splits;
  p.splits = (p.items + p.items_per_split - 1) / p.items_per_split;
  return true;
}

int64_t slab_floats(const WWParams& p) {
  return (int64_t)p.splits * p.d.td.count * 16 * p.Capad * p.Cgpad;
}

}  // namespace

// workspace bytes when the Winograd path takes this descriptor, else 0
int64_t wino_wgrad_workspace_bytes(const rehr_wgrad_desc& d) {
  WWParams p;
  if (!plan(d, p)) return 0;
  return (slab_floats(p) + (int64_t)p.splits * p.Capad) * 4 + 64;
}

// REHR_OK launched, REHR_ENOSUP not applicable
int wino_wgrad_try(const rehr_wgrad_desc& d, hipStream_t stream) {
  WWParams p;
  if (!plan(d, p)) return REHR_ENOSUP;
  const int64_t need = (slab_floats(p) + (int64_t)p.splits * p.Capad) * 4 + 64;
  if (!d.workspace || d.workspace_bytes < need || ((uintptr_t)d.workspace & 15)) return REHR_EINVAL;
  p.slabs = d.workspace;
  p.slab_bias = d.dbias ? d.workspace + slab_floats(p) : nullptr;
  // (only with many splits in whole groups of 8: with a handful of splits the valid blocks would pile up on a few XCDs)
  p.colocate = (d.td.count > 1 && p.splits >= 64 && p.splits % 8 == 0 && !(d.debug_flags & REHR_DBG_WGRAD_NO_TAP_COLOCATE)) ? 1 : 0;
  dim3 grid(p.splits, p.a_tiles * p.c_tiles, d.td.count);
  if (p.colocate) grid = dim3((unsigned)(((p.splits + 7) / 8) * 8 * d.td.count), p.a_tiles * p.c_tiles, 1);
  int rc;
  const bool virt = p.G != 0;
  if (p.w8) rc = virt ? launch_ww<1, 2, true, 1, 2>(p, grid, stream) : launch_ww<1, 2, false, 1, 2>(p, grid, stream);
  else if (p.fa == 1 && p.fb == 1) rc = virt ? launch_ww<1, 1, true>(p, grid, stream) : launch_ww<1, 1, false>(p, grid, stream);
  else if (p.fa == 1) rc = virt ? launch_ww<1, 2, true>(p, grid, stream) : launch_ww<1, 2, false>(p, grid, stream);
  else rc = virt ? launch_ww<2, 1, true>(p, grid, stream) : launch_ww<2, 1, false>(p, grid, stream);
  if (rc != REHR_OK) return rc;
  const int64_t total = (int64_t)d.td.count * d.Ca * d.Cg;
  if (total < ((int64_t)1 << 17) && p.splits >= 16) {
    int blocks = (int)((total + 31) / 32);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel<8>, dim3(blocks), dim3(256), 0, stream, p);
  } else {
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel<1>, dim3(blocks), dim3(256), 0, stream, p);
  }
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
