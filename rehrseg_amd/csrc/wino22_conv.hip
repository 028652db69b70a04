// Winograd F(2x2, 2x2) over (H, W) for the (., 4, 4) / (., 2, 2) transposed convolutions of the FLAVR
// decoder (models/FLAVR/FLAVR_arch.py:40-70) -- 27 % of a cfg-2 step ran them as direct GEMMs.
//
// A stride-2 transposed convolution with a 4-wide kernel is, per output parity (ph, pw), a unit-stride
// convolution with 2 x 2 taps on the input lattice; its input gradient is the sum over the 4 source
// parities of 2 x 2-tap convolutions over the parity sub-lattices of dY.  Both are 2-tap problems:
//
//   m0 = (d0 - d1) g0,  m1 = d1 (g0 + g1),  m2 = (d2 - d1) g1;   y0 = m0 + m1,  y1 = m1 + m2
//
// i.e. B^T = [[1,-1,0],[0,1,0],[0,-1,1]], G = [[1,0],[1,1],[0,1]], A^T = [[1,1,0],[0,1,1]] (all +-1): 9 products
// per 2 x 2 outputs instead of 16, exact-fp32 MFMA as everywhere else.
//
//   block    = 64 tiles (16 x 16 lattice outputs of one depth slice) x 64 channels, 12 waves: wave =
//              (Winograd row r, tile group, channel group) with 3 accumulator tiles (the row's 3 columns)
//   K item   = (source phase, 32-channel chunk, depth tap): the 17 x 17 patch of that phase's sub-lattice
//              in LDS (double-buffered, even/odd column split + row pad: conflict-free fragment reads)
//   schedule = three waves per SIMD hide each other's LDS latency; weights one k-group ahead in
//              registers, next patch fetched at the top of an item, one barrier per item
#include "common.h"
#include "wino_conv.h"
#include "wino22_shared.h"
#include <cstdlib>

namespace {

constexpr int LD = 36;                 // floats per voxel slot (32 channels + 4)
constexpr int PW = 17;                 // patch is 17 x 17
constexpr int RP = PW * LD + 12;       // row pitch: two rows = 32 floats mod 64
constexpr int BUF = PW * RP;           // floats per slice buffer
constexpr int PVOX = PW * PW;          // 289
constexpr int NT_ = 768;               // threads: 12 waves
constexpr int NX = (PVOX * 8 + NT_ - 1) / NT_;  // 4 pieces per thread
constexpr int MAXPH = 4;

struct Phase {
  int sh, sw;      // source stride of the sub-lattice (1 or 2)
  int ph, pw;      // parity offset inside the source
  int dh0, dw0;    // sub-lattice position of patch row/col 0 relative to the region origin
};

struct W22Params {
  rehr_gather_gemm_desc d;
  int nb_h, nb_w, kchunks, nphase;
  Phase phase[MAXPH];
  const float* up;   // U[phase][jd][9][Npad/32][kchunks][4][64][4]
  uint32_t up_bytes;
};

// U[phase][jd][xi = r*3+c] = G g G^T of the phase's 2x2 taps, in MFMA fragment order (zero beyond Cin)
__global__ void w22_weights_kernel(const rehr_gather_gemm_desc d, float* __restrict__ up, int kchunks) {
  AxisPlan ah, aw;
  plan_axis(d.th, d.sh, d.bh, ah);
  plan_axis(d.tw, d.sw, d.bw, aw);
  const int nphase = ah.nph * aw.nph;
  const int cpad = kchunks * 32;
  const int64_t per = (int64_t)d.Npad * cpad;
  const int64_t per_src = (int64_t)d.Npad * d.Cin;
  const int64_t total = (int64_t)nphase * d.td.count * per;
  const int NT = d.Npad / 32;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pj = (int)(i / per);            // phase * KD + jd
    const int p = pj / d.td.count, jd = pj - p * d.td.count;
    const int p_h = p / aw.nph, p_w = p - p_h * aw.nph;
    const int64_t rem = i - (int64_t)pj * per;
    const int n = (int)(rem / cpad), ci = (int)(rem - (int64_t)n * cpad);
    float g[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int wt = ((d.td.k0 + d.td.ks * jd) * d.KH + ah.kidx[p_h][a]) * d.KW + aw.kidx[p_w][b];
        g[a][b] = ci < d.Cin ? d.wp[(int64_t)wt * per_src + (int64_t)n * d.Cin + ci] : 0.f;
      }
    float u[9];
    u[0] = g[0][0];              u[1] = g[0][0] + g[0][1];                     u[2] = g[0][1];
    u[3] = g[0][0] + g[1][0];    u[4] = g[0][0] + g[0][1] + g[1][0] + g[1][1]; u[5] = g[0][1] + g[1][1];
    u[6] = g[1][0];              u[7] = g[1][0] + g[1][1];                     u[8] = g[1][1];
    const int nt = n >> 5, col = n & 31, chunk = ci >> 5, kk = (ci >> 3) & 3, half = (ci >> 2) & 1, e = ci & 3;
#pragma unroll
    for (int x = 0; x < 9; ++x) {
      const int64_t o = (((((int64_t)pj * 9 + x) * NT + nt) * kchunks + chunk) * 4 + kk) * 256 + (half * 32 + col) * 4 + e;
      up[o] = u[x];
    }
  }
}

__global__ __launch_bounds__(NT_) void wino22_conv_kernel(const W22Params p) {
  const rehr_gather_gemm_desc& d = p.d;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;

  // 12 waves = 3 Winograd rows x 2 tile groups x 2 channel groups: three waves per SIMD (9 points do not
  // divide over 4 SIMDs any other way -- the 3-wave version left one SIMD idle), 3 accumulator tiles each
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = wv % 3, fm = (wv / 3) & 1, fn = wv / 6;
  const int half = lane >> 5, col = lane & 31;
  const int n_img = blockIdx.z;
  const int nt0 = blockIdx.y * 2 + fn, n0 = blockIdx.y * 64;
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int bw_ = b % p.nb_w; b /= p.nb_w;
  const int bh_ = b % p.nb_h;
  const int od = b / p.nb_h;
  const int oh0 = bh_ * 16, ow0 = bw_ * 16;

  // B^T rows (d0 - d1, d1, d2 - d1): R = x[ia] (- x[1] unless ia == 1)
  const int ia = r;
  const int th_ = fm * 4 + (col >> 3), tw_ = col & 7;
  // patch column 2*tw_ + j -> slot (j&1)*9 + tw_ + (j>>1)
  const float* xa = Xs + (2 * th_ + ia) * RP + tw_ * LD + 4 * half;
  const float* xb = Xs + (2 * th_ + 1) * RP + tw_ * LD + 4 * half;

  // staging pieces (LDS order)
  int prow[NX], pcol[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int piece = tid + NT_ * i;
    const int v = piece >> 3;
    const int ph = (v * 3856) >> 16;          // v / 17 for v < 512
    const int slot = v - ph * PW;
    prow[i] = ph;
    pcol[i] = slot < 9 ? 2 * slot : 2 * (slot - 9) + 1;
  }
  const int pq = tid & 7;
  const uint32_t img_elems = (uint32_t)d.Di * d.Hi * d.Wi;
  int jd_lo = d.td.count, jd_hi = -1;  // depth taps with a source slice inside the volume (block-uniform)
  for (int j = 0; j < d.td.count; ++j) {
    const int id = od + d.bd + d.td.off0 + d.td.offs * j;
    if ((unsigned)id < (unsigned)d.Di) { jd_lo = min(jd_lo, j); jd_hi = max(jd_hi, j); }
  }
  const int items = p.nphase * p.kchunks * max(0, jd_hi - jd_lo + 1);
  f32x4 rx[NX];
  // items are walked with counters (phase, chunk, depth tap): no integer division in the loop
  struct Item { int ph, chunk, jd; };
  auto advance = [&](Item& t) {
    if (++t.jd > jd_hi) {
      t.jd = jd_lo;
      if (++t.chunk == p.kchunks) { t.chunk = 0; ++t.ph; }
    }
  };
  auto fetch = [&](const Item& t) {
    const bool live = t.ph < p.nphase;
    const int ph_i = live ? t.ph : 0;
    const int jd = t.jd;
    const int cc = t.chunk * 32;
    const Phase& P = p.phase[ph_i];
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
    for (int i = 0; i < NX; ++i) {
      const int piece = tid + NT_ * i;
      const int ih = (oh0 + P.dh0 + prow[i]) * P.sh + P.ph, iw = (ow0 + P.dw0 + pcol[i]) * P.sw + P.pw;
      const bool ok = dok & (piece < PVOX * 8) & ((unsigned)ih < (unsigned)d.Hi) & ((unsigned)iw < (unsigned)d.Wi);
      const uint32_t off = base + (uint32_t)(ih * d.Wi + iw) * ld * 4u;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : nrec, 0, 0));
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int piece = tid + NT_ * i;
      const int v = piece >> 3;
      if (piece < PVOX * 8) *reinterpret_cast<f32x4*>(Xs + buf + v * LD + prow[i] * 12 + pq * 4) = rx[i];
    }
  };

  const __amdgpu_buffer_rsrc_t rsu =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.up), 0, p.up_bytes, 0x00020000);
  const int NT = d.Npad / 32;
  const uint32_t xi_stride = (uint32_t)NT * p.kchunks * 4096u, nt_stride = (uint32_t)p.kchunks * 4096u;
  const uint32_t ulane = (uint32_t)(r * 3) * xi_stride + (uint32_t)nt0 * nt_stride + (uint32_t)lane * 16u;
  auto load_u = [&](const Item& t, int kk, f32x4 (&ub)[3]) {
    const int ph_i = t.ph < p.nphase ? t.ph : 0;  // (one item past the end is requested and never used)
#ifdef W22_DBG_SAMEW   // (timing diagnostics only: every k-group reads the same, cache-hot fragments)
    const uint32_t base = ulane + 0u * (uint32_t)(ph_i + kk);
#else
    const uint32_t base = (uint32_t)((ph_i * d.td.count + t.jd) * 9) * xi_stride + (uint32_t)(t.chunk * 4 + kk) * 1024u + ulane;
#endif
#pragma unroll
    for (int c = 0; c < 3; ++c)
      ub[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, base + c * xi_stride, 0, 0));
  };

  f32x16 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;

  // A fragments of k-group kk of slice buffer `buf`: v = B^T d B restricted to this wave's Winograd row
  auto read_a = [&](int buf, const int kk, f32x4 (&v)[3]) {
    f32x4 R[3];
#ifdef W22_DBG_NOLDS   // (timing diagnostics only: wrong results)
    R[0] = R[1] = R[2] = f32x4{1.f, 2.f, 3.f, 4.f};
#else
#pragma unroll
    for (int j = 0; j < 3; ++j) R[j] = *reinterpret_cast<const f32x4*>(xa + buf + ((j & 1) * 9 + (j >> 1)) * LD + kk * 8);
    if (r != 1) {  // wave-uniform: the middle Winograd row is the patch row itself
#pragma unroll
      for (int j = 0; j < 3; ++j) R[j] -= *reinterpret_cast<const f32x4*>(xb + buf + ((j & 1) * 9 + (j >> 1)) * LD + kk * 8);
    }
#endif
    v[0] = R[0] - R[1];
    v[1] = R[1];
    v[2] = R[2] - R[1];
  };
  auto mma = [&](const f32x4 (&v)[3], const f32x4 (&ub)[3]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][e], ub[c][e], acc[c], 0, 0, 0);
  };

  // Item loop.  Weights AND A fragments run one k-group ahead in registers, and the item barrier sits between the third
  // and the fourth k-group: behind it every wave still holds a k-group's operands (12 MFMAs) and has already asked for
  // the next item's first fragments.  (Round 3, s_memtime stamps: with the barrier at the end of an item all 12 waves
  // came out of it with nothing but address arithmetic and LDS latency in front of them -- 12.4k cycles per item for
  // 9.2k cycles of matrix-pipe work.)  s_setprio: the issue arbiter serves the oldest wave of a SIMD first, so the first
  // of its three waves ran its four k-groups back to back and then sat at the barrier for two thirds of an item while
  // the last one worked alone (stamps: 45-64k of 143k cycles waiting); a wave's priority now falls with every k-group
  // since the barrier, which makes the arbiter prefer whoever is behind.
#ifdef W22_STAMPS   // (timing diagnostics only)
  const long long st0 = __builtin_readcyclecounter();
#endif
  f32x4 u0[3], u1[3], va[3], vb[3];
  Item cur_i = {0, 0, min(jd_lo, d.td.count - 1)}, nxt_i = cur_i, fet_i;
  fetch(cur_i);
  load_u(cur_i, 0, u0);
  stage(0);
  advance(nxt_i);
  fetch(nxt_i);                      // patch 1 travels while item 0 computes (an item past the end loads zeros)
  fet_i = nxt_i;
  __syncthreads();
  read_a(0, 0, va);
#ifdef W22_STAMPS
  const long long st1 = __builtin_readcyclecounter();
  long long st_stage = 0, st_bar = 0;
#endif
  for (int it = 0; it < items; ++it) {
    const int cur = (it & 1) * BUF, nxt = cur ^ BUF;
    load_u(cur_i, 1, u1);
    read_a(cur, 1, vb);
    __builtin_amdgcn_s_setprio(2);
    mma(va, u0);
    load_u(cur_i, 2, u0);
    read_a(cur, 2, va);
    __builtin_amdgcn_s_setprio(1);
    mma(vb, u1);
    load_u(cur_i, 3, u1);
    read_a(cur, 3, vb);
    __builtin_amdgcn_s_setprio(0);
    mma(va, u0);
#ifdef W22_STAMPS
    const long long sa = __builtin_readcyclecounter();
#endif
    __builtin_amdgcn_s_setprio(3);   // everybody waits for the slowest stage: it goes first
    stage(nxt);                      // patch it + 1, asked for a whole item ago
#ifdef W22_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    const long long sb = __builtin_readcyclecounter();
#endif
#ifndef W22_DBG_NOBAR
    __syncthreads();
#endif
#ifdef W22_STAMPS
    const long long sc = __builtin_readcyclecounter();
    st_stage += sb - sa;
    st_bar += sc - sb;
#endif
    load_u(nxt_i, 0, u0);
    read_a(nxt, 0, va);
    __builtin_amdgcn_s_setprio(3);
    mma(vb, u1);
    advance(fet_i);
    fetch(fet_i);                    // patch it + 2 (address arithmetic under the MFMAs just issued)
    cur_i = nxt_i;
    nxt_i = fet_i;
  }
  __builtin_amdgcn_s_setprio(0);

#ifdef W22_STAMPS
  const long long st2 = __builtin_readcyclecounter();
#endif
  // ---- output transform Y = A^T M A, A^T = [[1,1,0],[0,1,1]]: columns in registers, rows through LDS
  float* ex = smem;  // [fm*2+fn][r 3][c' 2][q][lane]
  {
    const f32x16 T0 = acc[0] + acc[1];
    const f32x16 T1 = acc[1] + acc[2];
    float* e0 = ex + (((fm * 2 + fn) * 3 + r) * 2) * 16 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      e0[q * 64] = T0[q];
      e0[(16 + q) * 64] = T1[q];
    }
  }
  __syncthreads();
  const float neg_slope = d.act == REHR_ACT_NONE ? 1.f : (d.act == REHR_ACT_RELU ? 0.f : d.slope);
  const int col_n = n0 + fn * 32 + col;
  const bool colok = col_n < d.Cout;
  const float bv = (d.bias != nullptr && colok) ? d.bias[col_n] : 0.f;
  float s1_ = 0.f, s2_ = 0.f;
  // the wave's (tile group, channel group): output positions (ro, co) = r, and r + 3 for wave row 0
  for (int pos = r; pos < 4; pos += 3) {
    const int ro = pos >> 1, co = pos & 1;
    const float k0 = ro == 0 ? 1.f : 0.f, k2 = ro == 0 ? 0.f : 1.f;   // rows: (1,1,0) / (0,1,1)
    const float* e0 = ex + ((fm * 2 + fn) * 3 * 2 + co) * 16 * 64 + lane;
    float t[3][16];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int q = 0; q < 16; ++q) t[rr][q] = e0[(rr * 32 + q) * 64];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float yv = k0 * t[0][q] + t[1][q] + k2 * t[2][q] + bv;
      const float v = fmaxf(yv, 0.f) + neg_slope * fminf(yv, 0.f);
      const int oh = oh0 + 2 * (fm * 4 + (q >> 2)) + ro, ow = ow0 + 2 * ((q & 3) + 4 * half) + co;
      const bool ok = colok & (oh < d.Lh) & (ow < d.Lw);
      if (ok)
        d.y[((((int64_t)n_img * d.Dy + (od * d.osd + d.obd)) * d.Hy + (oh * d.osh + d.obh)) * d.Wy +
             (ow * d.osw + d.obw)) * d.ldy + col_n] = v;
      s1_ += ok ? v : 0.f;
      s2_ += ok ? v * v : 0.f;
    }
  }
  if (d.stats_mode != 0) {  // block-level sums first: one atomic per column and block
    s1_ += __shfl_xor(s1_, 32, 64);
    s2_ += __shfl_xor(s2_, 32, 64);
    __syncthreads();
    float* red = smem;  // [wave 12][2][32]
    if (half == 0) {
      red[(wv * 2 + 0) * 32 + col] = s1_;
      red[(wv * 2 + 1) * 32 + col] = s2_;
    }
    __syncthreads();
    if (wv < 2 && half == 0) {  // wave fn' sums the six (row, tile group) partials of its 32 columns
      const int fq = wv, cn = n0 + fq * 32 + col;
      if (cn < d.Cout) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w6 = 0; w6 < 6; ++w6) {
          a1 += red[((fq * 6 + w6) * 2 + 0) * 32 + col];
          a2 += red[((fq * 6 + w6) * 2 + 1) * 32 + col];
        }
        double* st = d.stats + ((int64_t)n_img * d.Cout + cn) * 2;
        atomicAdd(st, (double)a1);
        if (d.stats_mode == 2) atomicAdd(st + 1, (double)a2);
      }
    }
  }
#ifdef W22_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  const long long st3 = __builtin_readcyclecounter();
  if (lane == 0 && (blockIdx.x % 509) == 0 && blockIdx.z == 0)
    printf("w22 block %d wave %d items %d: prologue %lld  kloop %lld (stage waits %lld, barrier waits %lld)  epilogue %lld\n",
           (int)blockIdx.x, wv, items, st1 - st0, st2 - st1, st_stage, st_bar, st3 - st2);
#endif
}

// ------------------------------------------------------------------------------------------
// Flattened-tile variant for planes that whole 16 x 16 regions pad badly (12 x 12, 24 x 24: the transposed
// convolutions of the reference's own 96 x 96 crops): the 2 x 2 tiles of all slices are numbered consecutively (slice =
// depth-major, od * N + n), a block takes 64 consecutive tiles x 64 channels and stages the contiguous range of rows they
// read in the index space slice * (2 nth + 1) + lattice row -- the organisation of wino_flat8_conv.hip; compute loop,
// weight panel and summation order are this file's.
constexpr int F22_NX = 5;        // 16-byte pieces per thread
constexpr int F22_MAXSLOT = 8;   // slices a block may touch

struct F22Params {
  rehr_gather_gemm_desc d;
  int kchunks, nphase;
  Phase phase[MAXPH];
  const float* up;
  uint32_t up_bytes;
  int nth, ntw, tps, ntiles, PH, PWs, nev, RP, rowpad, rows, rowmagic, tab_off;
  // parts > 1: the output phases of one transposed convolution in ONE grid (blockIdx.z = part; each part a single
  // source phase = phase[part], its own weight panel and output offsets) -- 4 x 288 blocks as one 1152-block launch
  // instead of four launches of 1.125 rounds each
  int parts;
  const float* s_up[MAXPH];
  int s_obh[MAXPH], s_obw[MAXPH];
};

__global__ __launch_bounds__(NT_) void wino22_flat_conv_kernel(const F22Params p) {
  constexpr int TB = 64;
  const rehr_gather_gemm_desc& d = p.d;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;
  const int FBUF = p.rows * p.RP + LD + 4;   // + a spare voxel slot
  const int vtrash = p.rows * p.PWs;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = wv % 3, fm = (wv / 3) & 1, fn = wv / 6;
  const int half = lane >> 5, col = lane & 31;
  const int nt0 = blockIdx.y * 2 + fn, n0 = blockIdx.y * 64;
  const int T0 = blockIdx.x * TB;
  const int part = p.parts > 1 ? (int)blockIdx.z : 0;
  const int obh = p.parts > 1 ? p.s_obh[part] : d.obh, obw = p.parts > 1 ? p.s_obw[part] : d.obw;
  const float* const up = p.parts > 1 ? p.s_up[part] : p.up;
  const int nphase = p.parts > 1 ? 1 : p.nphase;
  const int Tl = min(T0 + TB, p.ntiles) - 1;
  const int s0 = T0 / p.tps, sl_ = Tl / p.tps;
  const int G0 = s0 * p.PH + 2 * ((T0 - s0 * p.tps) / p.ntw);
  const int rowsB = sl_ * p.PH + 2 * ((Tl - sl_ * p.tps) / p.ntw) + 3 - G0;   // <= p.rows (planner)

  int* tab = reinterpret_cast<int*>(smem + p.tab_off);
  if (tid < TB) {
    const int T = T0 + tid;
    const int s = T / p.tps, tt = T - s * p.tps;
    const int th = tt / p.ntw, tw = tt - th * p.ntw;
    const int od = s / d.N, n = s - od * d.N;
    tab[tid] = T < p.ntiles ? ((n * d.Dy + od * d.osd + d.obd) * d.Hy + 2 * th * d.osh + obh) * d.Wy + 2 * tw * d.osw + obw
                            : -1;
    tab[TB + tid] = min(s - s0, F22_MAXSLOT - 1);
    tab[2 * TB + tid] = (th << 16) | tw;
  }

  const int ia = r;
  const float *xa, *xb;
  {
    const int T = min(T0 + fm * 32 + col, p.ntiles - 1);
    const int s = T / p.tps, tt = T - s * p.tps;
    const int th = tt / p.ntw, tw = tt - th * p.ntw;
    const int grow = s * p.PH + 2 * th - G0;
    xa = Xs + (grow + ia) * p.RP + tw * LD + 4 * half;
    xb = Xs + (grow + 1) * p.RP + tw * LD + 4 * half;
  }
  const int off_odd = p.nev * LD;   // patch column 2*tw + j -> slot (j & 1) * nev + tw + (j >> 1)

  // staging pieces: (row, slot, channel quad) -> sample/depth base, lattice (row, column), live bit
  int pnd[F22_NX], prc[F22_NX];
  uint32_t pok = 0;
  const int v0 = tid >> 3, pq = tid & 7;
#pragma unroll
  for (int i = 0; i < F22_NX; ++i) {
    const int v = v0 + (NT_ / 8) * i;
    const int row = (v * p.rowmagic) >> 16, slot = v - row * p.PWs;
    const int Gr = G0 + row;
    const int s = Gr / p.PH, lr = Gr - s * p.PH;
    const int od = s / d.N, n = s - od * d.N;
    const int lc = slot < p.nev ? 2 * slot : 2 * (slot - p.nev) + 1;
    const bool ok = (row < rowsB) & (od < d.Ld);
    pnd[i] = ok ? n * d.Di + od : 0;
    prc[i] = (od << 24) | (lr << 12) | lc;     // od < 128, lr / lc < 4096 (planner)
    pok |= (ok ? 1u : 0u) << i;
  }
  int jd_lo = d.td.count, jd_hi = -1;
  {
    const int od_a = s0 / d.N, od_b = min(sl_ / d.N, d.Ld - 1);
    for (int j = 0; j < d.td.count; ++j) {
      const int dd = d.bd + d.td.off0 + d.td.offs * j;
      if (od_b + dd >= 0 && od_a + dd < d.Di) { jd_lo = min(jd_lo, j); jd_hi = max(jd_hi, j); }
    }
  }
  const int items = nphase * p.kchunks * max(0, jd_hi - jd_lo + 1);
  f32x4 rx[F22_NX];
  struct Item { int ph, chunk, jd; };
  auto advance = [&](Item& t) {
    if (++t.jd > jd_hi) {
      t.jd = jd_lo;
      if (++t.chunk == p.kchunks) { t.chunk = 0; ++t.ph; }
    }
  };
  const uint32_t tot1 = (uint32_t)d.N * d.Di * d.Hi * d.Wi;
  auto fetch = [&](const Item& t) {
    const bool live = (t.ph < nphase) & (items > 0);
    const int ph_i = live ? t.ph : 0;
    const int cc = t.chunk * 32;
    const Phase& P = p.phase[part + ph_i];
    const int dd = d.bd + d.td.off0 + d.td.offs * t.jd;
    const bool first = cc < d.c1;
    const float* src = first ? d.x1 : d.x2;
    const uint32_t ld = (uint32_t)(first ? d.ldx1 : d.ldx2);
    const int coff = first ? cc : cc - d.c1;
    const uint32_t nrec = tot1 * ld * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nrec, 0x00020000);
    const bool cok = live & ((cc + pq * 4) < d.Cin);
    const uint32_t cb = (uint32_t)(coff + pq * 4) * 4u;
#pragma unroll
    for (int i = 0; i < F22_NX; ++i) {
      const int lr = (prc[i] >> 12) & 0xfff, lc = prc[i] & 0xfff;
      const int ih = (lr + P.dh0) * P.sh + P.ph, iw = (lc + P.dw0) * P.sw + P.pw;
      const int nd = pnd[i] + dd;                      // n * Di + id
      const int id = (prc[i] >> 24) + dd;
      const bool ok = cok & ((pok >> i) & 1u) & ((unsigned)id < (unsigned)d.Di) & ((unsigned)ih < (unsigned)d.Hi) &
                      ((unsigned)iw < (unsigned)d.Wi);
      const uint32_t off = (uint32_t)((nd * d.Hi + ih) * d.Wi + iw) * ld * 4u + cb;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : nrec, 0, 0));
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < F22_NX; ++i) {
      const int v = min(v0 + (NT_ / 8) * i, vtrash);
      const int row = (v * p.rowmagic) >> 16;
      *reinterpret_cast<f32x4*>(Xs + buf + v * LD + row * p.rowpad + pq * 4) = rx[i];
    }
  };

  const __amdgpu_buffer_rsrc_t rsu =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(up), 0, p.up_bytes, 0x00020000);
  const int NT = d.Npad / 32;
  const uint32_t xi_stride = (uint32_t)NT * p.kchunks * 4096u, nt_stride = (uint32_t)p.kchunks * 4096u;
  const uint32_t ulane = (uint32_t)(r * 3) * xi_stride + (uint32_t)nt0 * nt_stride + (uint32_t)lane * 16u;
  auto load_u = [&](const Item& t, int kk, f32x4 (&ub)[3]) {
    const int ph_i = t.ph < nphase ? t.ph : 0;  // (one item past the end is requested and never used)
    const uint32_t base = (uint32_t)((ph_i * d.td.count + t.jd) * 9) * xi_stride + (uint32_t)(t.chunk * 4 + kk) * 1024u + ulane;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      ub[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, base + c * xi_stride, 0, 0));
  };

  f32x16 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;

  auto kstep = [&](int buf, const int kk, const f32x4 (&ub)[3]) {
    f32x4 R[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
      R[j] = *reinterpret_cast<const f32x4*>(xa + buf + ((j & 1) ? off_odd : 0) + (j >> 1) * LD + kk * 8);
    if (r != 1) {  // wave-uniform: the middle Winograd row is the patch row itself
#pragma unroll
      for (int j = 0; j < 3; ++j)
        R[j] -= *reinterpret_cast<const f32x4*>(xb + buf + ((j & 1) ? off_odd : 0) + (j >> 1) * LD + kk * 8);
    }
    f32x4 v[3];
    v[0] = R[0] - R[1];
    v[1] = R[1];
    v[2] = R[2] - R[1];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][e], ub[c][e], acc[c], 0, 0, 0);
  };

  f32x4 u0[3], u1[3];
  Item cur_i = {0, 0, min(jd_lo, d.td.count - 1)}, nxt_i = cur_i;
  fetch(cur_i);
  load_u(cur_i, 0, u0);
  stage(0);
  __syncthreads();
  for (int it = 0; it < items; ++it) {
    const int cur = (it & 1) * FBUF, nxt = cur ^ FBUF;
    advance(nxt_i);
    fetch(nxt_i);
    load_u(cur_i, 1, u1);
    kstep(cur, 0, u0);
    load_u(cur_i, 2, u0);
    kstep(cur, 1, u1);
    load_u(cur_i, 3, u1);
    kstep(cur, 2, u0);
    load_u(nxt_i, 0, u0);
    kstep(cur, 3, u1);
    stage(nxt);
    cur_i = nxt_i;
    __syncthreads();
  }

  // ---- output transform Y = A^T M A, A^T = [[1,1,0],[0,1,1]]: columns in registers, rows through LDS
  float* ex = smem;  // [fm*2+fn][r 3][c' 2][q][lane]
  {
    const f32x16 T0v = acc[0] + acc[1];
    const f32x16 T1v = acc[1] + acc[2];
    float* e0 = ex + (((fm * 2 + fn) * 3 + r) * 2) * 16 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      e0[q * 64] = T0v[q];
      e0[(16 + q) * 64] = T1v[q];
    }
  }
  __syncthreads();
  const float neg_slope = d.act == REHR_ACT_NONE ? 1.f : (d.act == REHR_ACT_RELU ? 0.f : d.slope);
  const int col_n = n0 + fn * 32 + col;
  const bool colok = col_n < d.Cout;
  const float bv = (d.bias != nullptr && colok) ? d.bias[col_n] : 0.f;
  float ssum[F22_MAXSLOT][2];
#pragma unroll
  for (int s = 0; s < F22_MAXSLOT; ++s) ssum[s][0] = ssum[s][1] = 0.f;
  // the wave's (tile group, channel group): output positions (ro, co) = r, and r + 3 for wave row 0
  for (int pos = r; pos < 4; pos += 3) {
    const int ro = pos >> 1, co = pos & 1;
    const float k0 = ro == 0 ? 1.f : 0.f, k2 = ro == 0 ? 0.f : 1.f;   // rows: (1,1,0) / (0,1,1)
    const float* e0 = ex + ((fm * 2 + fn) * 3 * 2 + co) * 16 * 64 + lane;
    float t[3][16];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int q = 0; q < 16; ++q) t[rr][q] = e0[(rr * 32 + q) * 64];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float yv = k0 * t[0][q] + t[1][q] + k2 * t[2][q] + bv;
      const float v = fmaxf(yv, 0.f) + neg_slope * fminf(yv, 0.f);
      const int m = fm * 32 + 8 * (q >> 2) + 4 * half + (q & 3);   // MFMA C row = tile
      const int tv = tab[m], hwp = tab[2 * TB + m], sl = tab[TB + m];
      const int oh = 2 * (hwp >> 16) + ro, ow = 2 * (hwp & 0xffff) + co;
      const bool ok = colok & (tv >= 0) & (oh < d.Lh) & (ow < d.Lw);
      if (ok) d.y[((int64_t)tv + (int64_t)ro * d.osh * d.Wy + co * d.osw) * d.ldy + col_n] = v;
      if (d.stats_mode != 0) {
        const float vs = ok ? v : 0.f;
#pragma unroll
        for (int s = 0; s < F22_MAXSLOT; ++s) {
          ssum[s][0] += (s == sl) ? vs : 0.f;
          ssum[s][1] += (s == sl) ? vs * vs : 0.f;
        }
      }
    }
  }
  if (d.stats_mode != 0) {  // block-level sums per (slice slot, column): one atomic each
    __syncthreads();
    float* red = smem;       // [slot][wave 12][half][2][32]
    const int nslot = min(sl_ - s0 + 1, F22_MAXSLOT);
#pragma unroll
    for (int s = 0; s < F22_MAXSLOT; ++s)
      if (s < nslot) {
        float* rp = red + (((s * 12 + wv) * 2 + half) * 2) * 32 + col;
        rp[0] = ssum[s][0];
        rp[32] = ssum[s][1];
      }
    __syncthreads();
    for (int task = tid; task < nslot * 64; task += NT_) {
      const int s = task >> 6, fq = (task >> 5) & 1, c = task & 31;
      const int cn = n0 + fq * 32 + c;
      const int slice = s0 + s, od = slice / d.N, n = slice - od * d.N;
      if (cn < d.Cout && od < d.Ld) {
        float a1 = 0.f, a2 = 0.f;
        for (int w6 = 0; w6 < 6; ++w6)        // the six (row, tile group) waves of channel group fq, both halves
          for (int h = 0; h < 2; ++h) {
            const float* rp = red + (((s * 12 + fq * 6 + w6) * 2 + h) * 2) * 32 + c;
            a1 += rp[0];
            a2 += rp[32];
          }
        double* st = d.stats + ((int64_t)n * d.Cout + cn) * 2;
        atomicAdd(st, (double)a1);
        if (d.stats_mode == 2) atomicAdd(st + 1, (double)a2);
      }
    }
  }
}

int f22_conflicts(int ntw, int RPf) {
  int worst = 0;
  for (int a = 0; a < ntw; ++a) {
    int cnt[16] = {0};
    for (int l = 0; l < 16; ++l) {
      const int t = a + l, th = t / ntw, tw = t - th * ntw;
      const int unit = ((tw * LD + th * 2 * RPf) / 4) & 15;
      if (++cnt[unit] > worst) worst = cnt[unit];
    }
  }
  return worst;
}

bool plan22_flat(const rehr_gather_gemm_desc& d, F22Params& p, int parts = 1) {
  if ((d.debug_flags & REHR_DBG_GG_NO_FLAT8) || d.sd != 1) return false;
  AxisPlan ah, aw;
  if (!plan_axis(d.th, d.sh, d.bh, ah) || !plan_axis(d.tw, d.sw, d.bw, aw)) return false;
  if (ah.nph != aw.nph) return false;
  if (ah.nph == 2 && (d.osh != 1 || d.osw != 1 || d.obh || d.obw)) return false;
  if (d.td.count < 1 || d.td.count > 3) return false;
  if (d.Npad % 64 || d.Lh < 6 || d.Lw < 6 || d.Lw > 64 || d.Lh > 2048 || d.Ld > 127) return false;
  if (d.Lh % 16 == 0 && d.Lw % 16 == 0) return false;      // whole regions: the region kernel stages less
  constexpr int TB = 64;
  p.d = d;
  p.nth = (d.Lh + 1) / 2;
  p.ntw = (d.Lw + 1) / 2;
  p.tps = p.nth * p.ntw;
  if ((int64_t)p.nth * 2 * p.ntw * 2 * 10 > (int64_t)d.Lh * d.Lw * 13) return false;
  const int64_t ntiles = (int64_t)d.N * d.Ld * p.tps;
  if (ntiles >= (1ll << 30) || ntiles < TB) return false;
  p.ntiles = (int)ntiles;
  // one 768-thread block per CU: below one full round of 64-tile blocks the padded region kernel (more, equally long
  // blocks) fills the chip better (512->128 @12x12: 144 blocks here against 256 there, 1.21 vs 1.12 ms)
  if ((ntiles + 63) / 64 * (d.Npad / 64) * parts < 256) return false;
  if ((TB - 2) / p.tps + 2 > F22_MAXSLOT) return false;
  p.PH = 2 * p.nth + 1;
  p.nev = p.ntw + 1;
  p.PWs = 2 * p.ntw + 1;
  const int trr = (TB - 2) / p.ntw + 2, sdiff = (TB - 2) / p.tps + 1;
  p.rows = 2 * (trr - 1) + sdiff + 3;
  if ((int64_t)p.rows * p.PWs * 8 > (int64_t)F22_NX * NT_) return false;
  p.rowmagic = 65536 / p.PWs + 1;
  for (int v = 0; v <= p.rows * p.PWs; ++v)
    if (((v * p.rowmagic) >> 16) != v / p.PWs) return false;
  int best = 1 << 30;
  p.rowpad = 12;
  for (int pad = 4; pad <= 64; pad += 4) {
    const int c = f22_conflicts(p.ntw, p.PWs * LD + pad);
    if (c < best) { best = c; p.rowpad = pad; }
  }
  p.RP = p.PWs * LD + p.rowpad;
  p.kchunks = (d.Cin + 31) / 32;
  p.parts = 1;
  p.nphase = ah.nph * aw.nph;
  for (int i = 0; i < ah.nph; ++i)
    for (int j = 0; j < aw.nph; ++j) {
      Phase& P = p.phase[i * aw.nph + j];
      P.sh = ah.stride; P.sw = aw.stride;
      P.ph = ah.par[i]; P.pw = aw.par[j];
      P.dh0 = ah.dmin[i]; P.dw0 = aw.dmin[j];
    }
  const int64_t need = (int64_t)p.nphase * d.td.count * 9 * d.Npad * p.kchunks * 32 * 4;
  if (need >= (1ll << 32) - 64) return false;
  p.up_bytes = (uint32_t)need;
  const int64_t tot = (int64_t)d.N * d.Di * d.Hi * d.Wi * 4;
  if (tot * d.ldx1 >= (1ll << 32) - 64 || (d.x2 && tot * d.ldx2 >= (1ll << 32) - 64)) return false;
  if ((int64_t)d.N * d.Dy * d.Hy * d.Wy >= (1ll << 31) || d.Npad / 64 > 65535) return false;
  const int64_t xfl = (int64_t)2 * (p.rows * p.RP + LD + 4), efl = (int64_t)4 * 3 * 2 * 16 * 64;
  const int64_t rfl = (int64_t)F22_MAXSLOT * 12 * 2 * 2 * 32;
  int64_t fl = xfl > efl ? xfl : efl;
  if (rfl > fl) fl = rfl;
  p.tab_off = (int)fl;
  if ((fl + 3 * TB) * 4 > 160 * 1024) return false;
  return true;
}

bool plan22(const rehr_gather_gemm_desc& d, W22Params& p) {
  if (d.sd != 1) return false;
  AxisPlan ah, aw;
  if (!plan_axis(d.th, d.sh, d.bh, ah) || !plan_axis(d.tw, d.sw, d.bw, aw)) return false;
  if (ah.nph != aw.nph) return false;                    // (2-tap, 2-tap) phases or (4-tap, 4-tap) stride-2 gathers
  if (ah.nph == 2 && (d.osh != 1 || d.osw != 1 || d.obh || d.obw)) return false;
  if (d.td.count < 1 || d.td.count > 3) return false;
  if (d.Npad % 64 || d.Lh < 12 || d.Lw < 12) return false;
  p.nb_h = (d.Lh + 15) / 16;
  p.nb_w = (d.Lw + 15) / 16;
  // padding to whole 16 x 16 regions: up to 1.3x always; up to 1.78x (12 x 12, 24 x 24 planes: the reference's own crops)
  // when the grid still fills the chip -- the padded transform-domain kernel then equals the direct kernel's MFMA count
  // at a better utilisation
  const int64_t padded = (int64_t)p.nb_h * 16 * p.nb_w * 16, exact = (int64_t)d.Lh * d.Lw;
  if (padded * 10 > exact * 13) {
    if (padded * 100 > exact * 178 || (int64_t)p.nb_h * p.nb_w * d.Ld * d.N * (d.Npad / 64) < 256) return false;
  }
  p.d = d;
  p.kchunks = (d.Cin + 31) / 32;
  p.nphase = ah.nph * aw.nph;
  for (int i = 0; i < ah.nph; ++i)
    for (int j = 0; j < aw.nph; ++j) {
      Phase& P = p.phase[i * aw.nph + j];
      P.sh = ah.stride; P.sw = aw.stride;
      P.ph = ah.par[i]; P.pw = aw.par[j];
      P.dh0 = ah.dmin[i]; P.dw0 = aw.dmin[j];
    }
  const int64_t need = (int64_t)p.nphase * d.td.count * 9 * d.Npad * p.kchunks * 32 * 4;
  if (need >= (1ll << 32) - 64) return false;
  p.up_bytes = (uint32_t)need;
  const int64_t img = (int64_t)d.Di * d.Hi * d.Wi * 4;
  if (img * d.ldx1 >= (1ll << 32) - 64 || (d.x2 && img * d.ldx2 >= (1ll << 32) - 64)) return false;
  if ((int64_t)p.nb_h * p.nb_w * d.Ld >= (1ll << 31) || d.Npad / 64 > 65535 || d.N > 65535) return false;
  return true;
}

}  // namespace

int64_t wino22_workspace_bytes(const rehr_gather_gemm_desc& d) {
  F22Params f;
  if (plan22_flat(d, f)) return (int64_t)f.up_bytes;
  W22Params p;
  return plan22(d, p) ? (int64_t)p.up_bytes : 0;
}

static bool f22_allow_smem(size_t smem) {   // the attribute only ever grows
  static size_t attr_smem = 0;
  if (smem > attr_smem) {
    if (hipFuncSetAttribute((const void*)wino22_flat_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
        hipSuccess)
      return false;
    attr_smem = smem;
  }
  return true;
}

static int wino22_flat_try(const rehr_gather_gemm_desc& d, hipStream_t stream) {
  F22Params p;
  if (!plan22_flat(d, p)) return REHR_ENOSUP;
  if (d.wino_ws_bytes < (int64_t)p.up_bytes || ((uintptr_t)d.wino_ws & 15)) return REHR_ENOSUP;
  p.up = d.wino_ws;
  const int64_t total = (int64_t)p.nphase * d.td.count * d.Npad * p.kchunks * 32;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (!(d.flags & REHR_GG_WS_READY)) {
    hipLaunchKernelGGL(w22_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d, d.wino_ws, p.kchunks);
    REHR_LAUNCH_CHECK();
  }
  if (d.flags & REHR_GG_WS_ONLY) return REHR_OK;
  const size_t smem = (size_t)(p.tab_off + 3 * 64) * sizeof(float);
  if (!f22_allow_smem(smem)) return REHR_EHIP;
  dim3 grid((unsigned)((p.ntiles + 63) / 64), d.Npad / 64, 1);
  hipLaunchKernelGGL(wino22_flat_conv_kernel, grid, dim3(NT_), smem, stream, p);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

// The output phases of one transposed convolution (same operands and lattice, different taps / output offsets / weight
// scratch) in one grid of the flattened-tile kernel.  REHR_OK launched; REHR_ENOSUP not applicable.
int wino22_flat_multi_try(const rehr_gather_gemm_desc* ds, int count, hipStream_t stream) {
  if (count < 2 || count > MAXPH) return REHR_ENOSUP;
  F22Params p;
  if (!ds[0].wino_ws || !plan22_flat(ds[0], p, count) || p.nphase != 1) return REHR_ENOSUP;
  const Phase ph0 = p.phase[0];
  const rehr_gather_gemm_desc& d0 = ds[0];
  for (int i = 0; i < count; ++i) {
    const rehr_gather_gemm_desc& d = ds[i];
    F22Params q;
    if (!d.wino_ws || ((uintptr_t)d.wino_ws & 15) || !plan22_flat(d, q, count) || q.nphase != 1 ||
        d.wino_ws_bytes < (int64_t)q.up_bytes || q.up_bytes != p.up_bytes)
      return REHR_ENOSUP;
    if (d.x1 != d0.x1 || d.x2 != d0.x2 || d.Cin != d0.Cin || d.c1 != d0.c1 || d.ldx1 != d0.ldx1 || d.ldx2 != d0.ldx2 ||
        d.Cout != d0.Cout || d.Npad != d0.Npad || d.y != d0.y || d.ldy != d0.ldy || d.N != d0.N || d.Ld != d0.Ld ||
        d.Lh != d0.Lh || d.Lw != d0.Lw || d.Di != d0.Di || d.Hi != d0.Hi || d.Wi != d0.Wi || d.Dy != d0.Dy ||
        d.Hy != d0.Hy || d.Wy != d0.Wy || d.osd != d0.osd || d.osh != d0.osh || d.osw != d0.osw || d.obd != d0.obd ||
        d.bd != d0.bd || d.td.count != d0.td.count || d.td.off0 != d0.td.off0 || d.td.offs != d0.td.offs ||
        d.bias != d0.bias || d.act != d0.act || d.slope != d0.slope || d.stats != d0.stats ||
        d.stats_mode != d0.stats_mode || q.phase[0].sh != ph0.sh || q.phase[0].sw != ph0.sw)
      return REHR_ENOSUP;
    p.phase[i] = q.phase[0];
    p.s_up[i] = d.wino_ws;
    p.s_obh[i] = d.obh;
    p.s_obw[i] = d.obw;
  }
  p.parts = count;
  p.up = nullptr;
  for (int i = 0; i < count; ++i) {
    const int64_t total = (int64_t)ds[i].td.count * ds[i].Npad * p.kchunks * 32;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (!(ds[i].flags & REHR_GG_WS_READY))
      hipLaunchKernelGGL(w22_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ds[i], ds[i].wino_ws, p.kchunks);
  }
  if (d0.flags & REHR_GG_WS_ONLY) {
    REHR_LAUNCH_CHECK();
    return REHR_OK;
  }
  const size_t smem = (size_t)(p.tab_off + 3 * 64) * sizeof(float);
  if (!f22_allow_smem(smem)) return REHR_EHIP;
  dim3 grid((unsigned)((p.ntiles + 63) / 64), d0.Npad / 64, count);
  hipLaunchKernelGGL(wino22_flat_conv_kernel, grid, dim3(NT_), smem, stream, p);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

int wino22_conv_try(const rehr_gather_gemm_desc& d, hipStream_t stream) {
  if (d.wino_ws) {
    const int frc = wino22_flat_try(d, stream);
    if (frc != REHR_ENOSUP) return frc;
  }
  W22Params p;
  if (!d.wino_ws || !plan22(d, p)) return REHR_ENOSUP;
  if (d.wino_ws_bytes < (int64_t)p.up_bytes || ((uintptr_t)d.wino_ws & 15)) return REHR_ENOSUP;
  p.up = d.wino_ws;
  const int64_t total = (int64_t)p.nphase * d.td.count * d.Npad * p.kchunks * 32;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (!(d.flags & REHR_GG_WS_READY)) {
    hipLaunchKernelGGL(w22_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d, d.wino_ws, p.kchunks);
    REHR_LAUNCH_CHECK();
  }
  if (d.flags & REHR_GG_WS_ONLY) return REHR_OK;
  const size_t smem_x = (size_t)2 * BUF * sizeof(float), smem_e = (size_t)4 * 3 * 2 * 16 * 64 * sizeof(float);
  const size_t smem = smem_x > smem_e ? smem_x : smem_e;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wino22_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
        hipSuccess)
      return REHR_EHIP;
    attr_set = true;
  }
  dim3 grid((unsigned)((int64_t)p.nb_h * p.nb_w * d.Ld), d.Npad / 64, d.N);
  hipLaunchKernelGGL(wino22_conv_kernel, grid, dim3(NT_), smem, stream, p);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
