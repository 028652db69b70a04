// Winograd F(2x2,3x3) for SMALL lattices (fewer than 64 tiles per depth slice: 8x8 ... 14x14 outputs), where
// the region kernels of wino_conv.hip would pad a slice to their 16x16 / 8x16 regions (12x12 -> 1.78x).
// These are the reference's own training shapes: 96x96 crops reach 24x24 / 12x12 after layer2 / layer3
// (configs/brain.yaml, train_all.py:316-330), and the nnU-Net bottom stages.
//
// The 2x2 tiles of ALL slices (sample x depth) are numbered consecutively; a block takes 64 consecutive tiles
// x 64 output channels, whatever slices they fall into (at most 4), and stages those slices' whole padded
// planes in LDS.  16 waves = (Winograd row, tile group, channel group), four per SIMD, 4 accumulator tiles
// each, 16-channel K items -- the thread-level-parallel organisation of wino_conv_w32_kernel.
#include "common.h"
#include "wino_conv.h"

namespace {

constexpr int LDF = 20;       // floats per voxel slot (16 channels + 4)
constexpr int NSMAX = 4;      // slices a block can touch
constexpr int NTF = 1024;

struct FlatParams {
  rehr_gather_gemm_desc d;
  int nth, ntw, tps;          // tiles per slice
  int ntiles;                 // N * Ld * tps
  int ns;                     // slices staged per block
  int PH, PWs, nev, RP, SL;   // padded plane: rows, column slots, even-column count, row pitch, slice pitch (floats)
  int kchunks;
  const float* up;
  uint32_t up_bytes;
};

__global__ __launch_bounds__(NTF) void wino_flat_conv_kernel(const FlatParams p) {
  const rehr_gather_gemm_desc& d = p.d;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;
  const int BUF = p.ns * p.SL;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = wv & 3, fm = (wv >> 2) & 1, fn = wv >> 3;
  const int half = lane >> 5, col = lane & 31;
  const int T0 = blockIdx.x * 64;
  const int sl0 = T0 / p.tps;                 // first slice (= n * Ld + od) of this block
  const int nt0 = blockIdx.y * 2 + fn, n0 = blockIdx.y * 64;

  const int i1 = (r == 0) ? 0 : 1, i2 = (r == 3) ? 3 : 2;
  const float s2 = (r == 1) ? 1.f : -1.f;
  const float rsign = (r == 2) ? -1.f : 1.f;
  // this lane's tile
  const int T = min(T0 + fm * 32 + col, p.ntiles - 1);
  const int sl = T / p.tps, tt = T - sl * p.tps;
  const int th_ = tt / p.ntw, tw_ = tt - th_ * p.ntw;
  const float* xa = Xs + (sl - sl0) * p.SL + (2 * th_ + i1) * p.RP + tw_ * LDF + 4 * half;
  const float* xb = Xs + (sl - sl0) * p.SL + (2 * th_ + i2) * p.RP + tw_ * LDF + 4 * half;
  const int off_j1 = p.nev * LDF;  // patch column 2*tw + j -> slot (j&1)*nev + tw + (j>>1)

  // ---- staging pieces: (slice, row, slot, quad) in LDS order; the source voxel offset without the depth part
  const int per_slice = p.PH * p.PWs * 4, total_pieces = p.ns * per_slice;
  constexpr int NXF = 4;  // ns * PH * PWs * 4 <= 4 * 1024 (checked by the planner)
  int pbase[NXF], pdep[NXF], plds[NXF];
  const int HW = d.Hi * d.Wi;
#pragma unroll
  for (int i = 0; i < NXF; ++i) {
    const int piece = tid + NTF * i;
    const int s = piece / per_slice, rem = piece - s * per_slice;
    const int v = rem >> 2, row = v / p.PWs, slot = v - row * p.PWs;
    const int pc = slot < p.nev ? 2 * slot : 2 * (slot - p.nev) + 1;
    const int slice = sl0 + s;                          // n * Ld + od  (output lattice = source lattice here)
    const int n = slice / d.Ld, od = slice - n * d.Ld;
    const int ih = row - 1, iw = pc - 1;
    const bool ok = (piece < total_pieces) & (n < d.N) & ((unsigned)ih < (unsigned)d.Hi) & ((unsigned)iw < (unsigned)d.Wi);
    pbase[i] = ok ? (n * d.Di) * HW + ih * d.Wi + iw : -1;
    pdep[i] = od;
    plds[i] = s * p.SL + row * p.RP + slot * LDF + (piece & 3) * 4;
  }
  const int pq = tid & 3;
  int jd_lo = 0, jd_hi = d.td.count - 1;
  const int nhalf = (d.Cin + 15) / 16;
  const int items = nhalf * d.td.count;
  struct Item { int h16, jd; };
  auto advance = [&](Item& t) {
    if (++t.jd > jd_hi) { t.jd = jd_lo; ++t.h16; }
  };
  const uint32_t tot1 = (uint32_t)d.N * d.Di * HW;  // voxels of a source tensor
  f32x4 rx[NXF];
  auto fetch = [&](const Item& t) {
    const bool live = t.h16 < nhalf;
    const int cc = (live ? t.h16 : 0) * 16;
    const int doff = d.bd + d.td.off0 + d.td.offs * t.jd;
    const bool first = cc < d.c1;
    const float* src = first ? d.x1 : d.x2;
    const uint32_t ld = (uint32_t)(first ? d.ldx1 : d.ldx2);
    const int coff = first ? cc : cc - d.c1;
    const uint32_t nrec = tot1 * ld * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nrec, 0x00020000);
    const bool cok = live & ((cc + pq * 4) < d.Cin);
#pragma unroll
    for (int i = 0; i < NXF; ++i) {
      const int id = pdep[i] + doff;
      const bool ok = cok & (pbase[i] >= 0) & ((unsigned)id < (unsigned)d.Di);
      const uint32_t off = ((uint32_t)(pbase[i] + id * HW) * ld + (uint32_t)(coff + pq * 4)) * 4u;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : nrec, 0, 0));
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NXF; ++i)
      if (tid + NTF * i < total_pieces) *reinterpret_cast<f32x4*>(Xs + buf + plds[i]) = rx[i];
  };

  const __amdgpu_buffer_rsrc_t rsu =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.up), 0, p.up_bytes, 0x00020000);
  const int NT = d.Npad / 32;
  const uint32_t xi_stride = (uint32_t)NT * p.kchunks * 4096u, nt_stride = (uint32_t)p.kchunks * 4096u;
  const uint32_t ulane = (uint32_t)lane * 16u, ubase = (uint32_t)(r * 4) * xi_stride + (uint32_t)nt0 * nt_stride;
  auto load_u = [&](const Item& t, const int kkl, f32x4 (&ub)[4]) {
    const int h16 = t.h16 < nhalf ? t.h16 : 0;
    const uint32_t base = ubase + (uint32_t)(t.jd * 16) * xi_stride +
                          (uint32_t)((h16 >> 1) * 4 + (h16 & 1) * 2 + kkl) * 1024u;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      ub[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsu, ulane, base + c * xi_stride, 0));
  };

  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
  auto kstep = [&](int buf, const int kkl, const f32x4 (&ub)[4]) {
    f32x4 R[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = ((j & 1) ? off_j1 : 0) + (j >> 1) * LDF + kkl * 8;
      const f32x4 a = *reinterpret_cast<const f32x4*>(xa + buf + o);
      const f32x4 bq = *reinterpret_cast<const f32x4*>(xb + buf + o);
      R[j] = a + bq * s2;
    }
    f32x4 v[4];
    v[0] = R[0] - R[2];
    v[1] = R[1] + R[2];
    v[2] = R[1] - R[2];  // negated column, undone at the output
    v[3] = R[1] - R[3];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][e], ub[c][e], acc[c], 0, 0, 0);
  };

  f32x4 u0[4], u1[4];
  Item ci = {0, 0}, ni = {0, 0};
  fetch(ci);
  load_u(ci, 0, u0);
  stage(0);
  __syncthreads();
  for (int it = 0; it < items; ++it) {
    const int cur = (it & 1) * BUF, nxt = cur ^ BUF;
    advance(ni);
    fetch(ni);
    load_u(ci, 1, u1);
    kstep(cur, 0, u0);
    load_u(ni, 0, u0);
    kstep(cur, 1, u1);
    stage(nxt);
    ci = ni;
    __syncthreads();
  }

  // ---- output transform (columns in registers, rows across the four row-waves through LDS)
  float* ex = smem;  // [fm*2+fn][r][c'][q][lane]
  {
    const f32x16 T0v = (acc[0] + acc[1] - acc[2]) * rsign;
    const f32x16 T1v = (acc[1] + acc[2] - acc[3]) * rsign;
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
  const int col_n = n0 + fn * 32 + col;
  const bool colok = col_n < d.Cout;
  const float bv = (d.bias != nullptr && colok) ? d.bias[col_n] : 0.f;
  const float* e0 = ex + ((fm * 2 + fn) * 4 * 2 + co) * 16 * 64 + lane;
  float t[4][16];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int q = 0; q < 16; ++q) t[rr][q] = e0[(rr * 32 + q) * 64];
  // the rows of a lane belong to up to 4 different samples: statistics are flushed whenever the sample changes
  float s1_ = 0.f, s2_ = 0.f;
  int n_cur = -1;
  auto flush = [&]() {
    if (d.stats_mode != 0 && n_cur >= 0 && colok) {
      double* st = d.stats + ((int64_t)n_cur * d.Cout + col_n) * 2;
      atomicAdd(st, (double)s1_);
      if (d.stats_mode == 2) atomicAdd(st + 1, (double)s2_);
    }
    s1_ = 0.f;
    s2_ = 0.f;
  };
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float yv = k0 * t[0][q] + t[1][q] + k2 * t[2][q] + k3 * t[3][q] + bv;
    const float v = fmaxf(yv, 0.f) + neg_slope * fminf(yv, 0.f);
    const int Tq = T0 + fm * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;  // MFMA C row = tile
    const int slq = Tq / p.tps, ttq = Tq - slq * p.tps;
    const int thq = ttq / p.ntw, twq = ttq - thq * p.ntw;
    const int n = slq / d.Ld, od = slq - n * d.Ld;
    const int oh = 2 * thq + ro, ow = 2 * twq + co;
    const bool ok = colok & (Tq < p.ntiles) & (oh < d.Lh) & (ow < d.Lw);
    if (n != n_cur) {
      flush();
      n_cur = n;
    }
    if (ok) {
      d.y[((((int64_t)n * d.Dy + od) * d.Hy + oh) * d.Wy + ow) * d.ldy + col_n] = v;
      s1_ += v;
      s2_ += v * v;
    }
  }
  flush();
}

bool three_taps_f(const rehr_axis_taps& t, int b) {
  if (t.count != 3) return false;
  const int o0 = b + t.off0, o1 = b + t.off0 + t.offs, o2 = b + t.off0 + 2 * t.offs;
  return (o1 == 0) && ((o0 == -1 && o2 == 1) || (o0 == 1 && o2 == -1));
}

bool plan_flat(const rehr_gather_gemm_desc& d, FlatParams& p) {
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return false;
  if (d.osd != 1 || d.osh != 1 || d.osw != 1 || d.obd || d.obh || d.obw) return false;
  if (d.Ld != d.Dy || d.Lh != d.Hy || d.Lw != d.Wy) return false;
  if (d.Ld != d.Di || d.Lh != d.Hi || d.Lw != d.Wi) return false;  // "same" convolution: source plane = output plane
  if (!three_taps_f(d.th, d.bh) || !three_taps_f(d.tw, d.bw)) return false;
  if (d.td.count < 1 || d.td.count > 3) return false;
  if (d.Npad % 64 || d.Lh < 6 || d.Lw < 6 || d.Lh > 16 || d.Lw > 16) return false;
  p.d = d;
  p.nth = (d.Lh + 1) / 2;
  p.ntw = (d.Lw + 1) / 2;
  p.tps = p.nth * p.ntw;
  if (p.tps >= 64 || p.tps < 16) return false;            // >= 64: the region kernels; < 16: too many slices per block
  if ((int64_t)p.nth * 2 * p.ntw * 2 * 10 > (int64_t)d.Lh * d.Lw * 13) return false;  // odd extents pad a half tile
  const int64_t ntiles = (int64_t)d.N * d.Ld * p.tps;
  if (ntiles >= (1ll << 30) || ntiles < 64) return false;
  p.ntiles = (int)ntiles;
  // one block per 64 tiles x 64 channels: below ~half a chip of blocks the split-K direct path wins
  if ((ntiles + 63) / 64 * (d.Npad / 64) < 128) return false;
  p.ns = (64 % p.tps == 0) ? 64 / p.tps : 63 / p.tps + 2;
  if (p.ns > NSMAX) return false;
  p.PH = 2 * p.nth + 2;           // the last tile reads patch rows 2*nth-2 .. 2*nth+1 (odd extents: one zero row more)
  const int PW = 2 * p.ntw + 2;
  p.PWs = PW;
  p.nev = (PW + 1) / 2;
  p.RP = PW * LDF + 4;
  p.SL = p.PH * p.RP;
  if (p.ns * p.PH * p.PWs * 4 > 4 * NTF) return false;
  p.kchunks = (d.Cin + 31) / 32;
  const int64_t need = (int64_t)d.td.count * 16 * d.Npad * p.kchunks * 32 * 4;
  if (need >= (1ll << 32) - 64) return false;
  p.up_bytes = (uint32_t)need;
  const int64_t tot = (int64_t)d.N * d.Di * d.Hi * d.Wi * 4;
  if (tot * d.ldx1 >= (1ll << 32) - 64 || (d.x2 && tot * d.ldx2 >= (1ll << 32) - 64)) return false;
  if (d.Npad / 64 > 65535) return false;
  return true;
}

}  // namespace

int64_t wino_flat_workspace_bytes(const rehr_gather_gemm_desc& d) {
  FlatParams p;
  return plan_flat(d, p) ? (int64_t)p.up_bytes : 0;
}

int wino_flat_conv_try(const rehr_gather_gemm_desc& d, hipStream_t stream) {
  FlatParams p;
  if (!d.wino_ws || !plan_flat(d, p)) return REHR_ENOSUP;
  if (d.wino_ws_bytes < (int64_t)p.up_bytes || ((uintptr_t)d.wino_ws & 15)) return REHR_ENOSUP;
  p.up = d.wino_ws;
  const int rc = wino_weights_frag_launch(d, p.kchunks, stream);
  if (rc != REHR_OK) return rc;
  const size_t smem_x = (size_t)2 * p.ns * p.SL * sizeof(float), smem_e = (size_t)4 * 4 * 2 * 16 * 64 * sizeof(float);
  const size_t smem = smem_x > smem_e ? smem_x : smem_e;
  if (smem > 160 * 1024) return REHR_ENOSUP;
  static size_t attr_smem = 0;
  if (smem > attr_smem) {
    if (hipFuncSetAttribute((const void*)wino_flat_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
        hipSuccess)
      return REHR_EHIP;
    attr_smem = smem;
  }
  dim3 grid((unsigned)((p.ntiles + 63) / 64), d.Npad / 64, 1);
  hipLaunchKernelGGL(wino_flat_conv_kernel, grid, dim3(NTF), smem, stream, p);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
