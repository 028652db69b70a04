// Winograd F(2x2,3x3) x depth-direct convolution for planes that the 16x16-output regions of wino_conv.hip do not
// tile well: the reference's own training shapes (96x96 crops -> 24x24 and 12x12 planes after layer2 / layer3,
// configs/brain.yaml, train_all.py:316-330) and the low nnU-Net stages.
//
// The 2x2 output tiles of ALL slices are numbered consecutively (slice = depth-major: od * N + n, so that the slices
// a block touches share their reachable depth taps); a block takes 32*NFM consecutive tiles x 64 output channels,
// whatever rows and slices they fall into, and stages the CONTIGUOUS RANGE OF PADDED ROWS those tiles read
// (row index = slice * (H + 2) + padded row) in LDS -- rows between two slices are the zero borders.
//
// Schedule: the big-tile kernel's (wino_conv_big8_kernel).  Wave = (Winograd row r, tile group fm), 2 x 4 accumulator
// tiles; 32-channel K items double-buffered in LDS; behind the 16 MFMAs of every half step the next k-group's
// weights are re-loaded, its A fragments read and combined, and the next item's rows fetched and staged.
// NFM = 2: 512 threads, one block per CU (two waves per SIMD).  NFM = 1: 256 threads, two independent blocks per CU
// -- half the tiles per block, for layers whose 64-tile block count quantises badly on 256 CUs.
#include "common.h"
#include "wino_conv.h"

namespace {

constexpr int F8_LDX = 36;   // floats per voxel slot (32 channels + 4)
constexpr int F8_NX = 8;     // 16-byte pieces staged per thread (two halves of 4)
constexpr int F8_NXH = F8_NX / 2;
constexpr int F8_MAXSLOT = 8;  // slices a block may touch (statistics slots)

struct Flat8Params {
  rehr_gather_gemm_desc d;
  int nth, ntw, tps;   // tiles per slice
  int ntiles;          // N * Ld * tps
  int PH;              // padded rows per slice (2 * nth + 2)
  int PWs, nev;        // column slots per row; even columns first (nev of them)
  int RP, rowpad;      // row pitch in floats = PWs * F8_LDX + rowpad
  int rows;            // rows a block stages at most
  int rowmagic;        // v / PWs == (v * rowmagic) >> 16 for every staged v
  int kchunks;
  int tab_off;         // float offset of the per-tile tables behind the exchange buffer
  const float* up;
  uint32_t up_bytes;
};

template <int NFM>
__global__ __launch_bounds__(256 * NFM, NFM == 1 ? 2 : 1) void wino_flat8_conv_kernel(const Flat8Params p) {
  constexpr int NT_ = 256 * NFM, TB = 32 * NFM;
  const rehr_gather_gemm_desc& d = p.d;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;
  const int BUF = p.rows * p.RP + F8_LDX + 4;   // + the spare slot
  const int vtrash = p.rows * p.PWs;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = wv & 3, fm = wv >> 2;
  const int half = lane >> 5, col = lane & 31;
  const int nt0 = blockIdx.y * 2, n0 = blockIdx.y * 64;
  const int T0 = blockIdx.x * TB;
  const int Tl = min(T0 + TB, p.ntiles) - 1;             // last live tile of the block
  const int s0 = T0 / p.tps, sl_ = Tl / p.tps;
  const int G0 = s0 * p.PH + 2 * ((T0 - s0 * p.tps) / p.ntw);
  const int rowsB = sl_ * p.PH + 2 * ((Tl - sl_ * p.tps) / p.ntw) + 4 - G0;   // <= p.rows (planner)
  const int HW = d.Hi * d.Wi;

  // ---- per-tile tables for the epilogue: output voxel of the tile's first pixel, slice slot, (th, tw)
  int* tab = reinterpret_cast<int*>(smem + p.tab_off);
  if (tid < TB) {
    const int T = T0 + tid;
    const int s = T / p.tps, tt = T - s * p.tps;
    const int th = tt / p.ntw, tw = tt - th * p.ntw;
    const int od = s / d.N, n = s - od * d.N;
    tab[tid] = T < p.ntiles ? ((n * d.Dy + od) * d.Hy + 2 * th) * d.Wy + 2 * tw : -1;
    tab[TB + tid] = min(s - s0, F8_MAXSLOT - 1);
    tab[2 * TB + tid] = (th << 16) | tw;
  }

  const int i1 = (r == 0) ? 0 : 1, i2 = (r == 3) ? 3 : 2;
  const float s2 = (r == 1) ? 1.f : -1.f;
  const float rsign = (r == 2) ? -1.f : 1.f;
  const float *xa, *xb;
  {
    const int T = min(T0 + fm * 32 + col, p.ntiles - 1);
    const int s = T / p.tps, tt = T - s * p.tps;
    const int th = tt / p.ntw, tw = tt - th * p.ntw;
    const int grow = s * p.PH + 2 * th - G0;
    xa = Xs + (grow + i1) * p.RP + tw * F8_LDX + 4 * half;
    xb = Xs + (grow + i2) * p.RP + tw * F8_LDX + 4 * half;
  }
  const int off_odd = p.nev * F8_LDX;   // patch column 2*tw + j -> slot (j & 1) * nev + tw + (j >> 1)

  // ---- staging pieces: (row, slot, channel quad); source voxel incl. the output depth, and 3 tap-validity bits each
  int pvx[F8_NX];
  uint32_t pmask = 0;
  const int v0 = tid >> 3, pq = tid & 7;
#pragma unroll
  for (int i = 0; i < F8_NX; ++i) {
    const int v = v0 + (NT_ / 8) * i;
    const int row = (v * p.rowmagic) >> 16, slot = v - row * p.PWs;
    const int Gr = G0 + row;
    const int s = Gr / p.PH, prow = Gr - s * p.PH;
    const int od = s / d.N, n = s - od * d.N;
    const int pc = slot < p.nev ? 2 * slot : 2 * (slot - p.nev) + 1;
    const int ih = prow - 1, iw = pc - 1;
    const bool ok = (row < rowsB) & (od < d.Ld) & ((unsigned)ih < (unsigned)d.Hi) & ((unsigned)iw < (unsigned)d.Wi);
    pvx[i] = ok ? ((n * d.Di + od) * d.Hi + ih) * d.Wi + iw : 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int id = od + d.bd + d.td.off0 + d.td.offs * j;
      const bool okj = ok & (j < d.td.count) & ((unsigned)id < (unsigned)d.Di);
      pmask |= (okj ? 1u : 0u) << (3 * i + j);
    }
  }
  // depth taps any staged slice reaches
  int jd_lo = d.td.count, jd_hi = -1;
  {
    const int od_a = s0 / d.N, od_b = min(sl_ / d.N, d.Ld - 1);
    for (int j = 0; j < d.td.count; ++j) {
      const int dd = d.bd + d.td.off0 + d.td.offs * j;
      if (od_b + dd >= 0 && od_a + dd < d.Di) { jd_lo = min(jd_lo, j); jd_hi = max(jd_hi, j); }
    }
  }
  const int items = p.kchunks * max(0, jd_hi - jd_lo + 1);
  struct Item { int chunk, jd; };
  auto advance = [&](Item& t) {
    if (++t.jd > jd_hi) { t.jd = jd_lo; ++t.chunk; }
  };
  const uint32_t tot1 = (uint32_t)d.N * d.Di * HW;
  auto fetch_to = [&](f32x4 (&rx)[F8_NXH], const Item& t, const int lo) {   // pieces lo .. lo + NXH - 1
    const bool live = (t.chunk < p.kchunks) & (items > 0);
    const int jd = min(t.jd, 2);
    const int cc = (live ? t.chunk : 0) * 32;
    const int dd = d.bd + d.td.off0 + d.td.offs * jd;
    const bool first = cc < d.c1;
    const float* src = first ? d.x1 : d.x2;
    const uint32_t ld = (uint32_t)(first ? d.ldx1 : d.ldx2);
    const int coff = first ? cc : cc - d.c1;
    const uint32_t nrec = tot1 * ld * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nrec, 0x00020000);
    const bool cok = live & ((cc + pq * 4) < d.Cin);
    const uint32_t m = cok ? (pmask >> jd) : 0u;
    const int dvox = dd * HW;
    const uint32_t cb = (uint32_t)(coff + pq * 4) * 4u;
#pragma unroll
    for (int i = 0; i < F8_NXH; ++i) {
      const bool ok = (m >> (3 * (lo + i))) & 1u;
      const uint32_t off = (uint32_t)(pvx[lo + i] + dvox) * ld * 4u + cb;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : nrec, 0, 0));
    }
  };
  auto stage_from = [&](const f32x4 (&rx)[F8_NXH], int buf, const int lo) {
#pragma unroll
    for (int i = 0; i < F8_NXH; ++i) {
      // (voxels past the staged rows -- zeros -- all land in one spare slot behind the buffer: no branch in the loop)
      const int v = min(v0 + (NT_ / 8) * (lo + i), vtrash);
      const int row = (v * p.rowmagic) >> 16;
      *reinterpret_cast<f32x4*>(Xs + buf + v * F8_LDX + row * p.rowpad + pq * 4) = rx[i];
    }
  };
  f32x4 rx[F8_NXH];
  auto fetch = [&](const Item& t, const int lo) { fetch_to(rx, t, lo); };
  auto stage = [&](int buf, const int lo) { stage_from(rx, buf, lo); };

  const __amdgpu_buffer_rsrc_t rsu =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.up), 0, p.up_bytes, 0x00020000);
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
      const int o = buf + ((j & 1) ? off_odd : 0) + (j >> 1) * F8_LDX + kk * 8;
      ra[j] = *reinterpret_cast<const f32x4*>(xa + o);
      rb[j] = *reinterpret_cast<const f32x4*>(xb + o);
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
  f32x4 u0[4], u1[4];
  Item ci = {0, min(jd_lo, d.td.count - 1)}, ni = ci;
  {  // item 0: both halves of the rows in flight at once (u1's registers are free here): one memory round trip
    f32x4 (&rx2)[F8_NXH] = reinterpret_cast<f32x4 (&)[F8_NXH]>(u1);
    fetch(ci, 0);
    fetch_to(rx2, ci, F8_NXH);
    load_u(ci, 0, 0, u0);
    stage(0, 0);
    stage_from(rx2, 0, F8_NXH);
  }
  load_u(ci, 0, 1, u1);
  __syncthreads();
  issue_reads(0, 0);
  combine(VA);

#define F8_FENCE() __builtin_amdgcn_sched_barrier(0)
#define F8_KGROUP(vcur, vnext, NEXT_T, NEXT_KK, READS, EXTRA0, EXTRA1) \
  F8_FENCE();                                                          \
  READS;                                                               \
  EXTRA0;                                                              \
  mfmas(0, vcur, u0);                                                  \
  F8_FENCE();                                                          \
  load_u(NEXT_T, NEXT_KK, 0, u0);                                      \
  EXTRA1;                                                              \
  mfmas(1, vcur, u1);                                                  \
  combine(vnext);                                                      \
  F8_FENCE();                                                          \
  load_u(NEXT_T, NEXT_KK, 1, u1);

  for (int it = 0; it < items; ++it) {
    const int cur = (it & 1) * BUF, nxt = cur ^ BUF;
    advance(ni);
    F8_KGROUP(VA, VB, ci, 1, issue_reads(cur, 1), fetch(ni, 0), (void)0)
    F8_KGROUP(VB, VA, ci, 2, issue_reads(cur, 2), (void)0, stage(nxt, 0))
    F8_KGROUP(VA, VB, ci, 3, issue_reads(cur, 3), fetch(ni, F8_NXH), (void)0)
    // last k-group: nobody reads the current rows any more (its fragments are in VB), so the barrier sits between its
    // two half steps and the next item's first fragments are read and combined behind 16 MFMAs
    F8_FENCE();
    stage(nxt, F8_NXH);
    mfmas(0, VB, u0);
    F8_FENCE();
    load_u(ni, 0, 0, u0);
    __syncthreads();
    F8_FENCE();
    issue_reads(nxt, 0);
    mfmas(1, VB, u1);
    combine(VA);
    F8_FENCE();
    load_u(ni, 0, 1, u1);
    ci = ni;
  }
#undef F8_KGROUP
#undef F8_FENCE
  __syncthreads();

  // ---- output transform: columns in registers, rows across the 4 row-waves of a tile group through LDS
  float* ex = smem;  // [fm*2+fn][r][c'][q][lane]
#pragma unroll
  for (int fn = 0; fn < 2; ++fn) {
    const f32x16 T0v = (acc[fn][0] + acc[fn][1] - acc[fn][2]) * rsign;
    const f32x16 T1v = (acc[fn][1] + acc[fn][2] - acc[fn][3]) * rsign;
    float* e0 = ex + (((fm * 2 + fn) * 4 + r) * 2) * 16 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      e0[q * 64] = T0v[q];
      e0[(16 + q) * 64] = T1v[q];
    }
  }
  __syncthreads();
  const int ro = r >> 1, co = r & 1;
  const float k0 = ro == 0 ? 1.f : 0.f, k2 = ro == 0 ? 1.f : -1.f, k3 = ro == 0 ? 0.f : -1.f;
  const float neg_slope = d.act == REHR_ACT_NONE ? 1.f : (d.act == REHR_ACT_RELU ? 0.f : d.slope);
  // per-q tile facts of this lane (same for both channel groups)
  int yv[16];
  uint32_t okbits = 0;                  // 16 x 1 bit: the tile exists and the pixel lies inside the lattice
  uint32_t slots = 0, slots_hi = 0;     // 16 x 4 bits: statistics slot (slice relative to the block's first) per value
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int m = fm * 32 + 8 * (q >> 2) + 4 * half + (q & 3);   // MFMA C row = tile
    const int tv = tab[m], hwp = tab[2 * TB + m], sl = tab[TB + m];
    const int oh = 2 * (hwp >> 16) + ro, ow = 2 * (hwp & 0xffff) + co;
    const bool ok = (tv >= 0) & (oh < d.Lh) & (ow < d.Lw);
    yv[q] = tv + ro * d.Wy + co;
    okbits |= (ok ? 1u : 0u) << q;
    if (q < 8) slots |= (uint32_t)sl << (4 * q);
    else slots_hi |= (uint32_t)sl << (4 * (q - 8));
  }
  float ssum[2][F8_MAXSLOT][2];
#pragma unroll
  for (int fn = 0; fn < 2; ++fn) {
    const int col_n = n0 + fn * 32 + col;
    const bool colok = col_n < d.Cout;
    const float bv = (d.bias != nullptr && colok) ? d.bias[col_n] : 0.f;
    const float* e0 = ex + ((fm * 2 + fn) * 4 * 2 + co) * 16 * 64 + lane;
    float t[4][16];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int q = 0; q < 16; ++q) t[rr][q] = e0[(rr * 32 + q) * 64];
#pragma unroll
    for (int s = 0; s < F8_MAXSLOT; ++s) ssum[fn][s][0] = ssum[fn][s][1] = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float y = k0 * t[0][q] + t[1][q] + k2 * t[2][q] + k3 * t[3][q] + bv;
      const float v = fmaxf(y, 0.f) + neg_slope * fminf(y, 0.f);
      const bool ok = colok & ((okbits >> q) & 1u);
      if (ok) d.y[(int64_t)yv[q] * d.ldy + col_n] = v;
      if (d.stats_mode != 0) {
        const int sl = (int)(((q < 8 ? slots : slots_hi) >> (4 * (q & 7))) & 15u);
        const float vs = ok ? v : 0.f;
#pragma unroll
        for (int s = 0; s < F8_MAXSLOT; ++s) {
          ssum[fn][s][0] += (s == sl) ? vs : 0.f;
          ssum[fn][s][1] += (s == sl) ? vs * vs : 0.f;
        }
      }
    }
  }
  if (d.stats_mode != 0) {  // block-level sums per (slice slot, column): one atomic each
    __syncthreads();         // everybody is done reading ex
    float* red = smem;       // [slot][wave][half][fn][2][32]
    const int nslot = min(sl_ - s0 + 1, F8_MAXSLOT);
#pragma unroll
    for (int fn = 0; fn < 2; ++fn)
#pragma unroll
      for (int s = 0; s < F8_MAXSLOT; ++s)
        if (s < nslot) {
          float* rp = red + ((((s * (4 * NFM) + wv) * 2 + half) * 2 + fn) * 2) * 32 + col;
          rp[0] = ssum[fn][s][0];
          rp[32] = ssum[fn][s][1];
        }
    __syncthreads();
    for (int task = tid; task < nslot * 64; task += NT_) {
      const int s = task >> 6, fn = (task >> 5) & 1, c = task & 31;
      const int col_n = n0 + fn * 32 + c;
      const int slice = s0 + s, od = slice / d.N, n = slice - od * d.N;
      if (col_n < d.Cout && od < d.Ld) {
        float a1 = 0.f, a2 = 0.f;
        for (int w = 0; w < 4 * NFM * 2; ++w) {
          const float* rp = red + (((s * (4 * NFM) * 2 + w) * 2 + fn) * 2) * 32 + c;
          a1 += rp[0];
          a2 += rp[32];
        }
        double* st = d.stats + ((int64_t)n * d.Cout + col_n) * 2;
        atomicAdd(st, (double)a1);
        if (d.stats_mode == 2) atomicAdd(st + 1, (double)a2);
      }
    }
  }
}

bool three_taps_f8(const rehr_axis_taps& t, int b) {
  if (t.count != 3) return false;
  const int o0 = b + t.off0, o1 = b + t.off0 + t.offs, o2 = b + t.off0 + 2 * t.offs;
  return (o1 == 0) && ((o0 == -1 && o2 == 1) || (o0 == 1 && o2 == -1));
}

// LDS bank-group spread of one 16-lane phase of a fragment read: lanes = 16 consecutive tiles at one column parity,
// 16-byte units (tw * 9 + (row pitch / 2) * tile row) mod 16; returns the worst multiplicity over the alignments
int f8_conflicts(int ntw, int RP) {
  int worst = 0;
  for (int a = 0; a < ntw; ++a) {
    int cnt[16] = {0};
    for (int l = 0; l < 16; ++l) {
      const int t = a + l, th = t / ntw, tw = t - th * ntw;
      const int unit = ((tw * F8_LDX + th * 2 * RP) / 4) & 15;
      if (++cnt[unit] > worst) worst = cnt[unit];
    }
  }
  return worst;
}

bool plan_flat8(const rehr_gather_gemm_desc& d, Flat8Params& p, int nfm) {
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return false;
  if (d.osd != 1 || d.osh != 1 || d.osw != 1 || d.obd || d.obh || d.obw) return false;
  if (d.Ld != d.Dy || d.Lh != d.Hy || d.Lw != d.Wy) return false;
  if (d.Ld != d.Di || d.Lh != d.Hi || d.Lw != d.Wi) return false;  // "same" convolution: source plane = output plane
  if (!three_taps_f8(d.th, d.bh) || !three_taps_f8(d.tw, d.bw)) return false;
  if (d.td.count < 1 || d.td.count > 3) return false;
  if (d.Npad % 64 || d.Lh < 6 || d.Lw < 6 || d.Lw > 64) return false;
  if (d.Lh % 16 == 0 && d.Lw % 16 == 0) return false;      // whole 16 x 16 regions: the region kernel has less halo
  const int TB = 32 * nfm, NT_ = 256 * nfm;
  p.d = d;
  p.nth = (d.Lh + 1) / 2;
  p.ntw = (d.Lw + 1) / 2;
  p.tps = p.nth * p.ntw;
  if ((int64_t)p.nth * 2 * p.ntw * 2 * 10 > (int64_t)d.Lh * d.Lw * 13) return false;  // odd extents pad a half tile
  const int64_t ntiles = (int64_t)d.N * d.Ld * p.tps;
  if (ntiles >= (1ll << 30) || ntiles < TB) return false;
  p.ntiles = (int)ntiles;
  // below ~half a chip of 64 x 64 units the split-K direct path wins
  if ((ntiles + 63) / 64 * (d.Npad / 64) < 128) return false;
  if ((TB - 2) / p.tps + 2 > F8_MAXSLOT) return false;
  p.PH = 2 * p.nth + 2;
  p.nev = p.ntw + 1;
  p.PWs = 2 * p.nev;
  const int trr = (TB - 2) / p.ntw + 2, sdiff = (TB - 2) / p.tps + 1;
  p.rows = 2 * (trr - 1) + 2 * sdiff + 4;
  if ((int64_t)p.rows * p.PWs * 8 > (int64_t)F8_NX * NT_) return false;
  p.rowmagic = 65536 / p.PWs + 1;
  for (int v = 0; v <= p.rows * p.PWs; ++v)
    if (((v * p.rowmagic) >> 16) != v / p.PWs) return false;
  int best = 1 << 30;
  p.rowpad = 8;
  for (int pad = 4; pad <= 64; pad += 4) {
    const int c = f8_conflicts(p.ntw, p.PWs * F8_LDX + pad);
    if (c < best) { best = c; p.rowpad = pad; }
  }
  p.RP = p.PWs * F8_LDX + p.rowpad;
  p.kchunks = (d.Cin + 31) / 32;
  const int64_t need = (int64_t)d.td.count * 16 * d.Npad * p.kchunks * 32 * 4;
  if (need >= (1ll << 32) - 64) return false;
  p.up_bytes = (uint32_t)need;
  const int64_t tot = (int64_t)d.N * d.Di * d.Hi * d.Wi * 4;
  if (tot * d.ldx1 >= (1ll << 32) - 64 || (d.x2 && tot * d.ldx2 >= (1ll << 32) - 64)) return false;
  if ((int64_t)d.N * d.Dy * d.Hy * d.Wy >= (1ll << 31)) return false;
  if (d.Npad / 64 > 65535) return false;
  const int64_t xfl = (int64_t)2 * (p.rows * p.RP + F8_LDX + 4), efl = (int64_t)nfm * 2 * 4 * 2 * 16 * 64;
  const int64_t rfl = (int64_t)F8_MAXSLOT * 4 * nfm * 2 * 2 * 2 * 32;
  int64_t fl = xfl > efl ? xfl : efl;
  if (rfl > fl) fl = rfl;
  p.tab_off = (int)fl;
  const int64_t bytes = (fl + 3 * TB) * 4;
  if (bytes > (nfm == 1 ? 80 : 160) * 1024) return false;
  return true;
}

template <int NFM>
int launch_flat8(const Flat8Params& p, hipStream_t stream) {
  constexpr int TB = 32 * NFM;
  const size_t smem = (size_t)(p.tab_off + 3 * TB) * sizeof(float);
  static size_t attr_smem = 0;
  if (smem > attr_smem) {
    if (hipFuncSetAttribute((const void*)wino_flat8_conv_kernel<NFM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess)
      return REHR_EHIP;
    attr_smem = smem;
  }
  dim3 grid((unsigned)((p.ntiles + TB - 1) / TB), p.d.Npad / 64, 1);
  hipLaunchKernelGGL(wino_flat8_conv_kernel<NFM>, grid, dim3(256 * NFM), smem, stream, p);
  return REHR_OK;
}

// tiles per block: 64 unless the 64-tile grid leaves a badly filled last round on 256 CUs and the 32-tile one does not
int f8_pick(const rehr_gather_gemm_desc& d, const Flat8Params& p64) {
  if (d.debug_flags & REHR_DBG_GG_FLAT8_HALF) return 1;
  if (d.debug_flags & REHR_DBG_GG_FLAT8_FULL) return 2;
  const int64_t units = (int64_t)((p64.ntiles + 63) / 64) * (d.Npad / 64);
  const double c64 = (double)((units + 255) / 256);             // rounds of one 64-tile block per CU
  const int64_t u2 = (int64_t)((p64.ntiles + 31) / 32) * (d.Npad / 64), rem = u2 % 512;
  const double c32 = (double)(u2 / 512) + (rem == 0 ? 0.0 : rem <= 256 ? 0.5 : 1.0);   // two 32-tile blocks per CU
  return (c32 * 1.08 < c64) ? 1 : 2;
}

}  // namespace

int64_t wino_flat8_workspace_bytes(const rehr_gather_gemm_desc& d) {
  if (d.debug_flags & REHR_DBG_GG_NO_FLAT8) return 0;
  Flat8Params p;
  return plan_flat8(d, p, 2) ? (int64_t)p.up_bytes : 0;
}

int wino_flat8_conv_try(const rehr_gather_gemm_desc& d, hipStream_t stream) {
  if ((d.debug_flags & REHR_DBG_GG_NO_FLAT8) || !d.wino_ws) return REHR_ENOSUP;
  Flat8Params p;
  if (!plan_flat8(d, p, 2)) return REHR_ENOSUP;
  if (d.wino_ws_bytes < (int64_t)p.up_bytes || ((uintptr_t)d.wino_ws & 15)) return REHR_ENOSUP;
  int nfm = f8_pick(d, p);
  Flat8Params p1;
  if (nfm == 1 && !plan_flat8(d, p1, 1)) nfm = 2;
  Flat8Params& q = nfm == 1 ? p1 : p;
  q.up = d.wino_ws;
  int rc = wino_weights_frag_launch(d, q.kchunks, stream);
  if (rc != REHR_OK || (d.flags & REHR_GG_WS_ONLY)) return rc;
  rc = nfm == 1 ? launch_flat8<1>(q, stream) : launch_flat8<2>(q, stream);
  if (rc != REHR_OK) return rc;
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
