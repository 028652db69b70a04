// Weight gradient on the fp32 matrix cores (see rehr_wgrad_desc).
//
//   dst[a][c][tap] = sum over (sample, lattice voxel v)  L[v][a] * G[src(v,tap)][c]
//
// GEMM view per tap: M = lattice channels a, N = gathered channels c, K = all
// lattice voxels of the batch.  K is huge (up to 2M voxels) and M x N is small,
// so the grid is (tap, a-tile, c-tile) x K-splits; every split writes its
// partial tile to a workspace slab with plain stores and a second kernel sums
// the slabs in a fixed order (bitwise reproducible, unlike float atomics) while
// transposing to the torch parameter layout.
//
// Both operands arrive voxel-major ([voxel][channel], channel contiguous), which
// is exactly the MFMA A/B register order for k = voxel: lane l reads
// tile[k = 2*step + (l>>5)][i = l&31] with a conflict-free ds_read_b32, no
// transposition anywhere.
//
// The K-step body is one basic block: raw buffer loads return 0 for an offset
// past num_records (tail of a split, taps that leave the source), and the lattice
// coordinates of a row come from multiply-high "magic" divisions, so the address
// arithmetic of the next tile schedules into the shadow of the current MFMAs.
#include "common.h"
#include "wgrad_shared.h"

namespace {

constexpr int NTHREADS = 256;

inline Magic make_magic(uint32_t d) {
  Magic g;
  if (d <= 1) { g.m = 0; g.sh = 255; return g; }
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  if (s < 1) s = 1;
  g.m = (uint32_t)(((1ull << (31 + s)) / d) + 1);
  g.sh = s - 1;
  return g;
}
__device__ __forceinline__ uint32_t mdiv(uint32_t n, Magic g) {
  const uint32_t q = __umulhi(n, g.m) >> (g.sh & 31);
  return g.sh == 255 ? n : q;
}

// BA x BG output tile, 4 waves as WGA x WGG x WGK (WGK waves split the K tile)
template <int BA, int BG, int WGA, int WGG, int BKV>
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_kernel(const WGParams p) {
  constexpr int WTA = BA / WGA, WTG = BG / WGG;
  constexpr int FA = WTA / 32, FG = WTG / 32;
  constexpr int LDA = BA + 4, LDG = BG + 4;
  constexpr int A_TPR = BA / 4, G_TPR = BG / 4;          // threads per row
  constexpr int A_RPP = NTHREADS / A_TPR, G_RPP = NTHREADS / G_TPR;  // rows per pass
  constexpr int A_PASS = BKV / A_RPP, G_PASS = BKV / G_RPP;
  constexpr int WGK = 4 / (WGA * WGG);
  static_assert(WGA * WGG * WGK == 4, "4 waves");
  static_assert(A_PASS >= 1 && G_PASS >= 1, "tile too narrow for BKV");
  constexpr int KSTEPS = (BKV / 2) / WGK;  // MFMA k-steps per wave per K tile
  const rehr_wgrad_desc& d = p.d;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ls = smem;                    // [2][BKV][LDA]
  float* Gs = smem + 2 * BKV * LDA;    // [2][BKV][LDG]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / (WGA * WGG);
  const int wa = (wave % (WGA * WGG)) / WGG, wg = wave % WGG;

  // block -> (tap, a tile, c tile); c fastest so blocks sharing L rows are neighbours
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = b % p.c_tiles; b /= p.c_tiles;
  const int at = b % p.a_tiles; b /= p.a_tiles;
  const int tap = b;
  const int jw = tap % d.tw.count;
  const int jh = (tap / d.tw.count) % d.th.count;
  const int jd = tap / (d.tw.count * d.th.count);
  const int dd = d.bd + d.td.off0 + d.td.offs * jd, dh = d.bh + d.th.off0 + d.th.offs * jh,
            dw = d.bw + d.tw.off0 + d.tw.offs * jw;
  const int a0 = at * BA, c0 = ct * BG;
  const int split = blockIdx.y;
  const int64_t v_begin = (int64_t)split * p.kv_per_split;
  int64_t v_end = v_begin + p.kv_per_split;
  if (v_end > p.kv_total) v_end = p.kv_total;
  const uint32_t lhw = (uint32_t)d.Lh * d.Lw;
  const uint32_t lvox = (uint32_t)d.Ld * lhw;

  f32x16 acc[FA][FG];
#pragma unroll
  for (int i = 0; i < FA; ++i)
#pragma unroll
    for (int j = 0; j < FG; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[FA];
#pragma unroll
  for (int i = 0; i < FA; ++i) bsum[i] = 0.f;
  const bool do_bias = (p.slab_bias != nullptr) && tap == 0 && ct == 0 && wg == 0;  // every wk
  const float bias_w = do_bias ? 1.f : 0.f;

  // L rows of this split are addressed relative to v_begin: rows past v_end are
  // past num_records and read as zero.
  const int64_t nrows = v_end > v_begin ? v_end - v_begin : 0;
  const uint32_t l_bytes = (uint32_t)(nrows * d.ldl * 4);
  const __amdgpu_buffer_rsrc_t rl_ = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.l) + v_begin * d.ldl, 0, l_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg_ =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.g), 0, p.g_bytes, 0x00020000);

  // two register sets: tiles are fetched two K-steps ahead (HBM-latency tolerance)
  f32x4 rl0[A_PASS], rg0[G_PASS], rl1[A_PASS], rg1[G_PASS];
  const int aq = tid % A_TPR, ar = tid / A_TPR;
  const int gq = tid % G_TPR, gr = tid / G_TPR;
  const bool a_col_ok = (a0 + aq * 4) < d.Ca;   // Ca % 32 == 0 and tiles are 32-multiples
  const bool g_col_ok = (c0 + gq * 4) < d.Cg;
  const uint32_t a_cb = (uint32_t)(a0 + aq * 4) * 4u, g_cb = (uint32_t)(c0 + gq * 4) * 4u;
  const uint32_t ldlb = (uint32_t)d.ldl * 4u, ldgb = (uint32_t)d.ldg * 4u;

  // rbase = row offset of the tile inside the split
  auto issue_loads = [&](uint32_t rbase, f32x4 (&rl)[A_PASS], f32x4 (&rg)[G_PASS]) {
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const uint32_t row = rbase + ar + i * A_RPP;
      const uint32_t off = a_col_ok ? row * ldlb + a_cb : l_bytes;
      rl[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl_, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < G_PASS; ++i) {
      const uint32_t row = rbase + gr + i * G_RPP;
      const uint32_t v = (uint32_t)v_begin + row;      // kv_total < 2^31 (checked on the host)
      const uint32_t n = mdiv(v, p.mg_vox);
      uint32_t rem = v - n * lvox;
      const uint32_t od = mdiv(rem, p.mg_hw);
      rem -= od * lhw;
      const uint32_t oh = mdiv(rem, p.mg_w);
      const uint32_t ow = rem - oh * d.Lw;
      const int id = (int)od * d.sd + dd, ih = (int)oh * d.sh + dh, iw = (int)ow * d.sw + dw;
      const bool inb = ((unsigned)id < (unsigned)d.Dg) & ((unsigned)ih < (unsigned)d.Hg) &
                       ((unsigned)iw < (unsigned)d.Wg) & (row < (uint32_t)nrows) & g_col_ok;
      const uint32_t gv = ((n * d.Dg + id) * d.Hg + ih) * d.Wg + iw;
      const uint32_t off = inb ? gv * ldgb + g_cb : p.g_bytes;
      rg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg_, off, 0, 0));
    }
  };
  auto commit_loads = [&](int buf, const f32x4 (&rl)[A_PASS], const f32x4 (&rg)[G_PASS]) {
    float* l = Ls + buf * BKV * LDA;
    float* g = Gs + buf * BKV * LDG;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i)
      *reinterpret_cast<f32x4*>(l + (ar + i * A_RPP) * LDA + aq * 4) = rl[i];
#pragma unroll
    for (int i = 0; i < G_PASS; ++i)
      *reinterpret_cast<f32x4*>(g + (gr + i * G_RPP) * LDG + gq * 4) = rg[i];
  };

  const int nsteps = (int)((nrows + BKV - 1) / BKV);
  const int acol = wa * WTA + (lane & 31);
  const int gcol = wg * WTG + (lane & 31);
  const int krow = lane >> 5;
  auto compute = [&](int buf) {
    const float* l = Ls + buf * BKV * LDA;
    const float* g = Gs + buf * BKV * LDG;
#pragma unroll
    for (int kk = wk * KSTEPS; kk < (wk + 1) * KSTEPS; ++kk) {
      float fa[FA], fg[FG];
#pragma unroll
      for (int i = 0; i < FA; ++i) fa[i] = l[(kk * 2 + krow) * LDA + acol + 32 * i];
#pragma unroll
      for (int j = 0; j < FG; ++j) fg[j] = g[(kk * 2 + krow) * LDG + gcol + 32 * j];
#pragma unroll
      for (int i = 0; i < FA; ++i) bsum[i] += fa[i] * bias_w;  // branch-free (bias_w is 0 or 1)
#pragma unroll
      for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FG; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fg[j], acc[i][j], 0, 0, 0);
      if (kk == wk * KSTEPS + 1) __builtin_amdgcn_sched_barrier(0);  // fetch leaves early, not after the tile
    }
  };

  if (nsteps > 0) {
    issue_loads(0, rl0, rg0);
    issue_loads(BKV, rl1, rg1);  // past the split: zeros
    commit_loads(0, rl0, rg0);
  }
  __syncthreads();

  // top of loop: tile s staged in LDS[0], tile s+1 in flight in set 1
  for (int s = 0; s < nsteps; s += 2) {
    issue_loads((uint32_t)(s + 2) * BKV, rl0, rg0);
    compute(0);
    commit_loads(1, rl1, rg1);
    __syncthreads();
    if (s + 1 >= nsteps) break;
    issue_loads((uint32_t)(s + 3) * BKV, rl1, rg1);
    compute(1);
    commit_loads(0, rl0, rg0);
    __syncthreads();
  }

  if (WGK > 1) {
    // waves that split K combine through LDS (the staging buffers are free now)
    float* red = smem;  // [WGK-1][FA*FG*16 + FA][64]
    constexpr int PER = FA * FG * 16 + FA;
    if (wk > 0) {
      float* o = red + (int64_t)(wk - 1) * PER * 64;
#pragma unroll
      for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FG; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[((i * FG + j) * 16 + r) * 64 + lane] = acc[i][j][r];
#pragma unroll
      for (int i = 0; i < FA; ++i) o[(FA * FG * 16 + i) * 64 + lane] = bsum[i];
    }
    __syncthreads();
    if (wk > 0) return;
    for (int k = 0; k < WGK - 1; ++k) {
      const float* o = red + (int64_t)k * PER * 64;
#pragma unroll
      for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FG; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += o[((i * FG + j) * 16 + r) * 64 + lane];
#pragma unroll
      for (int i = 0; i < FA; ++i) bsum[i] += o[(FA * FG * 16 + i) * 64 + lane];
    }
  }
  // partial tile -> slab[split][tap][Capad][Cgpad]
  float* slab = d.workspace + (((int64_t)split * p.T + tap) * p.Capad) * p.Cgpad;
  const int chalf = lane >> 5;
#pragma unroll
  for (int i = 0; i < FA; ++i)
#pragma unroll
    for (int j = 0; j < FG; ++j) {
      const int col = c0 + wg * WTG + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = a0 + wa * WTA + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
        slab[(int64_t)row * p.Cgpad + col] = acc[i][j][r];
      }
    }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < FA; ++i) {
      float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);
      const int row = a0 + wa * WTA + i * 32 + (lane & 31);
      if (chalf == 0 && row < d.Ca) p.slab_bias[(int64_t)split * d.Ca + row] = v;
    }
  }
}

// dst[a*sa + c*sc + wt*st] (+)= sum_split slab[split][tap][a][c]
// G threads share an output element: thread (e, g) sums the slabs k = g, g + G, ... (four independent loads in flight),
// the G partial sums are combined in a fixed order through LDS -- the result does not depend on the launch geometry of
// anything but this kernel, i.e. it stays bitwise reproducible.  (One thread per element walking all slabs serially was a
// latency-bound chain: 36-47 us per launch on the 32/64-channel layers, 1.2-1.6 ms of a mixed-precision step.)
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WGParams p) {
  constexpr int EPB = 256 / G;   // elements per block
  __shared__ float part[G > 1 ? 256 : 1];
  const rehr_wgrad_desc& d = p.d;
  const int64_t total = (int64_t)p.T * d.Ca * d.Cg;
  const int64_t slab_sz = (int64_t)p.T * p.Capad * p.Cgpad;
  const int el = threadIdx.x % EPB, g = threadIdx.x / EPB;
  for (int64_t i0 = (int64_t)blockIdx.x * EPB; i0 < total; i0 += (int64_t)gridDim.x * EPB) {
    const int64_t i = i0 + el;
    const bool ok = i < total;
    const int c = ok ? (int)(i % d.Cg) : 0;
    const int64_t r = ok ? i / d.Cg : 0;
    const int a = (int)(r % d.Ca);
    const int tap = (int)(r / d.Ca);
    const float* s = d.workspace + ((int64_t)tap * p.Capad + a) * p.Cgpad + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (ok) {
      int k = g;
      for (; k + 3 * G < p.splits; k += 4 * G) {
        s0 += s[(int64_t)k * slab_sz];
        s1 += s[(int64_t)(k + G) * slab_sz];
        s2 += s[(int64_t)(k + 2 * G) * slab_sz];
        s3 += s[(int64_t)(k + 3 * G) * slab_sz];
      }
      for (; k < p.splits; k += G) s0 += s[(int64_t)k * slab_sz];
    }
    float sum = (s0 + s1) + (s2 + s3);
    if (G > 1) {
      __syncthreads();
      part[threadIdx.x] = sum;
      __syncthreads();
      if (g == 0) {
#pragma unroll
        for (int q = 1; q < G; ++q) sum += part[q * EPB + el];
      }
    }
    if (ok && g == 0) {
      const int jw = tap % d.tw.count;
      const int jh = (tap / d.tw.count) % d.th.count;
      const int jd = tap / (d.tw.count * d.th.count);
      const int wt = ((d.td.k0 + d.td.ks * jd) * d.KH + (d.th.k0 + d.th.ks * jh)) * d.KW +
                     (d.tw.k0 + d.tw.ks * jw);
      float* o = d.dst + a * d.dst_sa + c * d.dst_sc + wt * d.dst_st;
      *o = d.accumulate ? (*o + sum) : sum;
    }
  }
  if (p.slab_bias != nullptr && d.dbias != nullptr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.Ca;
         i += (int64_t)gridDim.x * blockDim.x) {
      float sum = 0.f;
      for (int k = 0; k < p.splits; ++k) sum += p.slab_bias[(int64_t)k * d.Ca + i];
      d.dbias[i] = d.accumulate ? (d.dbias[i] + sum) : sum;
    }
  }
}

int launch_wgrad_reduce(const WGParams& p, hipStream_t st) {
  const int64_t total = (int64_t)p.T * p.d.Ca * p.d.Cg;
  // few elements and many slabs: spread every element over 8 threads
  if (total < ((int64_t)1 << 19) && p.splits >= 16) {
    int64_t blocks = (total + 31) / 32;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, p);
  } else {
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, p);
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Mixed precision: bf16 operands, fp32 accumulate on v_mfma_f32_32x32x16_bf16, fp32 slabs / result.
// The MFMA k index is the voxel and a lane's fragment is 8 CONSECUTIVE voxels of its channel, while both
// operands arrive voxel-major ([voxel][channel]): the LDS image keeps that layout (16-byte coalesced stores)
// and the fragments are read with gfx950's transposing ds_read_b64_tr_b16 -- a 16-lane group fetches a block of
// 4 voxels x 16 channels and every lane receives its channel's 4 voxels; two reads make the 8-voxel fragment.
// Row stride = B*2 bytes (+ 64 for B > 32): the 4 voxel rows of a 32-lane half fall into disjoint bank ranges.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int B, int WGA, int WGG, int BKV>
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_bf16_kernel(const WGParams p) {
  constexpr int WTA = B / WGA, WTG = B / WGG;
  constexpr int FA = WTA / 32, FG = WTG / 32;
  constexpr int ROW = B * 2 + (B == 32 ? 0 : 64);              // bytes per voxel row of an LDS image
  constexpr int TPR = B / 8, RPP = NTHREADS / TPR, PASS = BKV / RPP;
  constexpr int WGK = 4 / (WGA * WGG);
  constexpr int KSTEPS = (BKV / 16) / WGK;
  static_assert(WGA * WGG * WGK == 4 && PASS >= 1 && KSTEPS >= 1, "tile shape");
  const rehr_wgrad_desc& d = p.d;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* Ls = smem_b;                       // [2][BKV][ROW]
  unsigned char* Gs = smem_b + 2 * BKV * ROW;       // [2][BKV][ROW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / (WGA * WGG);
  const int wa = (wave % (WGA * WGG)) / WGG, wg = wave % WGG;

  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = b % p.c_tiles; b /= p.c_tiles;
  const int at = b % p.a_tiles; b /= p.a_tiles;
  const int tap = b;
  const int jw = tap % d.tw.count;
  const int jh = (tap / d.tw.count) % d.th.count;
  const int jd = tap / (d.tw.count * d.th.count);
  const int dd = d.bd + d.td.off0 + d.td.offs * jd, dh = d.bh + d.th.off0 + d.th.offs * jh,
            dw = d.bw + d.tw.off0 + d.tw.offs * jw;
  const int a0 = at * B, c0 = ct * B;
  const int split = blockIdx.y;
  const int64_t v_begin = (int64_t)split * p.kv_per_split;
  int64_t v_end = v_begin + p.kv_per_split;
  if (v_end > p.kv_total) v_end = p.kv_total;
  const uint32_t lhw = (uint32_t)d.Lh * d.Lw;
  const uint32_t lvox = (uint32_t)d.Ld * lhw;

  f32x16 acc[FA][FG];
#pragma unroll
  for (int i = 0; i < FA; ++i)
#pragma unroll
    for (int j = 0; j < FG; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int64_t nrows = v_end > v_begin ? v_end - v_begin : 0;
  const uint32_t l_bytes = (uint32_t)(nrows * d.ldl * 2);
  const __bf16* lp = reinterpret_cast<const __bf16*>(d.l);
  const __amdgpu_buffer_rsrc_t rl_ = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(lp) + v_begin * d.ldl, 0, l_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg_ = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(reinterpret_cast<const __bf16*>(d.g)), 0, p.g_bytes, 0x00020000);

  u32x4 rl0[PASS], rg0[PASS], rl1[PASS], rg1[PASS];
  const int tq = tid % TPR, tr = tid / TPR;
  const bool a_col_ok = (a0 + tq * 8) < d.Ca;     // Ca % 8 == 0
  const bool g_col_ok = (c0 + tq * 8) < d.Cg;
  const uint32_t a_cb = (uint32_t)(a0 + tq * 8) * 2u, g_cb = (uint32_t)(c0 + tq * 8) * 2u;
  const uint32_t ldlb = (uint32_t)d.ldl * 2u, ldgb = (uint32_t)d.ldg * 2u;

  auto issue_loads = [&](uint32_t rbase, u32x4 (&rl)[PASS], u32x4 (&rg)[PASS]) {
#pragma unroll
    for (int i = 0; i < PASS; ++i) {
      const uint32_t row = rbase + tr + i * RPP;
      rl[i] = __builtin_amdgcn_raw_buffer_load_b128(rl_, a_col_ok ? row * ldlb + a_cb : l_bytes, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PASS; ++i) {
      const uint32_t row = rbase + tr + i * RPP;
      const uint32_t v = (uint32_t)v_begin + row;
      const uint32_t n = mdiv(v, p.mg_vox);
      uint32_t rem = v - n * lvox;
      const uint32_t od = mdiv(rem, p.mg_hw);
      rem -= od * lhw;
      const uint32_t oh = mdiv(rem, p.mg_w);
      const uint32_t ow = rem - oh * d.Lw;
      const int id = (int)od * d.sd + dd, ih = (int)oh * d.sh + dh, iw = (int)ow * d.sw + dw;
      const bool inb = ((unsigned)id < (unsigned)d.Dg) & ((unsigned)ih < (unsigned)d.Hg) &
                       ((unsigned)iw < (unsigned)d.Wg) & (row < (uint32_t)nrows) & g_col_ok;
      const uint32_t gv = ((n * d.Dg + id) * d.Hg + ih) * d.Wg + iw;
      rg[i] = __builtin_amdgcn_raw_buffer_load_b128(rg_, inb ? gv * ldgb + g_cb : p.g_bytes, 0, 0);
    }
  };
  auto commit_loads = [&](int buf, const u32x4 (&rl)[PASS], const u32x4 (&rg)[PASS]) {
    unsigned char* l = Ls + buf * BKV * ROW;
    unsigned char* g = Gs + buf * BKV * ROW;
#pragma unroll
    for (int i = 0; i < PASS; ++i) *reinterpret_cast<u32x4*>(l + (tr + i * RPP) * ROW + tq * 16) = rl[i];
#pragma unroll
    for (int i = 0; i < PASS; ++i) *reinterpret_cast<u32x4*>(g + (tr + i * RPP) * ROW + tq * 16) = rg[i];
  };

  const int nsteps = (int)((nrows + BKV - 1) / BKV);
  // transposed fragment reads: 16-lane group -> (k half h, channel half cg); lane 4q+pp supplies voxel row q, channels 4pp..
  const int grp = lane >> 4, h = grp >> 1, cg = grp & 1, q = (lane & 15) >> 2, pp = lane & 3;
  const int lrow = 8 * h + q;
  const int lcolA = (wa * WTA + 16 * cg + 4 * pp) * 2, lcolG = (wg * WTG + 16 * cg + 4 * pp) * 2;
  auto frag = [&](const unsigned char* img, int kk, int colb) -> bf16x8 {
    const unsigned char* a = img + (kk * 16 + lrow) * ROW + colb;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * ROW));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto compute = [&](int buf) {
    const unsigned char* l = Ls + buf * BKV * ROW;
    const unsigned char* g = Gs + buf * BKV * ROW;
#pragma unroll
    for (int kk = wk * KSTEPS; kk < (wk + 1) * KSTEPS; ++kk) {
      bf16x8 fa[FA], fg[FG];
#pragma unroll
      for (int i = 0; i < FA; ++i) fa[i] = frag(l, kk, lcolA + 64 * i);
#pragma unroll
      for (int j = 0; j < FG; ++j) fg[j] = frag(g, kk, lcolG + 64 * j);
#pragma unroll
      for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FG; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fg[j], acc[i][j], 0, 0, 0);
    }
  };

  if (nsteps > 0) {
    issue_loads(0, rl0, rg0);
    issue_loads(BKV, rl1, rg1);
    commit_loads(0, rl0, rg0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; s += 2) {
    issue_loads((uint32_t)(s + 2) * BKV, rl0, rg0);
    compute(0);
    commit_loads(1, rl1, rg1);
    __syncthreads();
    if (s + 1 >= nsteps) break;
    issue_loads((uint32_t)(s + 3) * BKV, rl1, rg1);
    compute(1);
    commit_loads(0, rl0, rg0);
    __syncthreads();
  }

  if (WGK > 1) {
    float* red = reinterpret_cast<float*>(smem_b);  // [WGK-1][FA*FG*16][64]
    constexpr int PER = FA * FG * 16;
    __syncthreads();
    if (wk > 0) {
      float* o = red + (int64_t)(wk - 1) * PER * 64;
#pragma unroll
      for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FG; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[((i * FG + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (wk > 0) return;
    for (int k = 0; k < WGK - 1; ++k) {
      const float* o = red + (int64_t)k * PER * 64;
#pragma unroll
      for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FG; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += o[((i * FG + j) * 16 + r) * 64 + lane];
    }
  }
  float* slab = d.workspace + (((int64_t)split * p.T + tap) * p.Capad) * p.Cgpad;
  const int chalf = lane >> 5;
#pragma unroll
  for (int i = 0; i < FA; ++i)
#pragma unroll
    for (int j = 0; j < FG; ++j) {
      const int col = c0 + wg * WTG + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = a0 + wa * WTA + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
        slab[(int64_t)row * p.Cgpad + col] = acc[i][j][r];
      }
    }
}

template <int B, int WGA, int WGG, int BKV>
int launch_wg_bf16(const WGParams& p, hipStream_t stream) {
  constexpr int ROW = B * 2 + (B == 32 ? 0 : 64);
  size_t smem = (size_t)4 * BKV * ROW;
  constexpr int WGK = 4 / (WGA * WGG);
  constexpr size_t red = (size_t)(WGK - 1) * ((B / WGA / 32) * (B / WGG / 32) * 16) * 64 * 4;
  if (red > smem) smem = red;
  auto kern = wgrad_bf16_kernel<B, WGA, WGG, BKV>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess)
      return REHR_EHIP;
    attr_set = true;
  }
  dim3 grid(p.T * p.a_tiles * p.c_tiles, p.splits, 1);
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), smem, stream, p);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}


int tile_for(int c) { return (c % 128 == 0) ? 128 : ((c % 64 == 0) ? 64 : 32); }  // c % 32 != 0 -> masked 32-tiles
int bkv_for(int t) { return t == 128 ? 32 : 64; }

// 0 ok, else REHR_* code
int plan(const rehr_wgrad_desc& d, WGParams& p, int es = 4) {
  if (!d.l || !d.g || !d.dst) return REHR_EINVAL;
  if (d.Ca < 4 || d.Ca % 4 || d.Cg < 4 || d.Cg % 4) return REHR_EINVAL;  // tiles are padded, columns masked
  if (d.ldl % 4 || d.ldg % 4) return REHR_EINVAL;
  if (((uintptr_t)d.l | (uintptr_t)d.g) & 15) return REHR_EINVAL;
  if (d.N < 1 || d.Ld < 1 || d.Lh < 1 || d.Lw < 1) return REHR_EINVAL;
  if (d.Dg < 1 || d.Hg < 1 || d.Wg < 1) return REHR_EINVAL;
  if (d.td.count < 1 || d.th.count < 1 || d.tw.count < 1) return REHR_EINVAL;
  p.d = d;
  p.T = d.td.count * d.th.count * d.tw.count;
  // one tile size for both operands keeps the instantiation count small
  int t = tile_for(d.Ca);
  const int tg = tile_for(d.Cg);
  if (tg < t) t = tg;
  p.a_tiles = (d.Ca + t - 1) / t;
  p.c_tiles = (d.Cg + t - 1) / t;
  p.Capad = p.a_tiles * t;
  p.Cgpad = p.c_tiles * t;
  p.kv_total = (int64_t)d.N * d.Ld * d.Lh * d.Lw;
  const int64_t gbytes = (int64_t)d.N * d.Dg * d.Hg * d.Wg * d.ldg * es;
  if (p.kv_total >= (1ll << 31) - 4096 || gbytes >= (1ll << 32) - 64) return REHR_ENOSUP;
  p.g_bytes = (uint32_t)gbytes;
  const int bkv = es == 2 ? 64 : bkv_for(t);
  const int64_t tiles = (int64_t)p.T * p.a_tiles * p.c_tiles;
  // aim for ~4 blocks per CU-slot (256 CUs x 2 resident) with >= 16 K steps each
  int64_t want = (2048 + tiles - 1) / tiles;
  const int64_t max_by_k = (p.kv_total + 16 * bkv - 1) / (16 * bkv);
  if (want > max_by_k) want = max_by_k;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  {
    // All blocks take the same time and 512 are resident at once (256 CUs x 2): pick the
    // split count near `want` whose last round of blocks is fullest (tail effect).
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
  int64_t per = (p.kv_total + want - 1) / want;
  per = (per + bkv - 1) / bkv * bkv;
  if ((int64_t)per * d.ldl * es >= (1ll << 32) - 64) return REHR_ENOSUP;
  p.kv_per_split = per;
  p.splits = (int)((p.kv_total + per - 1) / per);
  p.mg_vox = make_magic((uint32_t)(d.Ld * d.Lh * d.Lw));
  p.mg_hw = make_magic((uint32_t)(d.Lh * d.Lw));
  p.mg_w = make_magic((uint32_t)d.Lw);
  return REHR_OK;
}

int64_t ws_bytes(const WGParams& p) {
  int64_t f = (int64_t)p.splits * p.T * p.Capad * p.Cgpad;
  f += (int64_t)p.splits * p.d.Ca;  // bias slab
  return f * (int64_t)sizeof(float);
}

template <int B, int WGA, int WGG, int BKV>
int launch_wg(const WGParams& p, hipStream_t stream) {
  size_t smem = (size_t)2 * BKV * ((B + 4) + (B + 4)) * sizeof(float);
  constexpr int WGK = 4 / (WGA * WGG);
  constexpr size_t red = (size_t)(WGK - 1) * ((B / WGA / 32) * (B / WGG / 32) * 16 + (B / WGA / 32)) * 64 * 4;
  if (red > smem) smem = red;
  auto kern = wgrad_kernel<B, B, WGA, WGG, BKV>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return REHR_EHIP;
    attr_set = true;
  }
  dim3 grid(p.T * p.a_tiles * p.c_tiles, p.splits, 1);
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), smem, stream, p);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

int plan_bf16(const rehr_wgrad_desc& d, WGParams& p) {
  if (d.Ca % 8 || d.Cg % 8 || d.ldl % 8 || d.ldg % 8 || d.dbias != nullptr) return REHR_EINVAL;
  return plan(d, p, 2);
}

}  // namespace

extern "C" int64_t rehr_wgrad_bf16_workspace_bytes(const rehr_wgrad_desc* dp) {
  if (!dp) return REHR_EINVAL;
  WGParams p;
  rehr_wgrad_desc d = *dp;
  if (!d.dst) d.dst = reinterpret_cast<float*>(16);
  const int rc = plan_bf16(d, p);
  if (rc != REHR_OK) return rc;
  BrickBf16 bo;
  WGParams pb = p;
  if (!(d.debug_flags & REHR_DBG_WGRAD_DIRECT) && wgrad_brick_bf16_plan(d, pb, bo))
    return (int64_t)pb.splits * pb.T * pb.Capad * pb.Cgpad * (int64_t)sizeof(float);
  return (int64_t)p.splits * p.T * p.Capad * p.Cgpad * (int64_t)sizeof(float);
}

extern "C" int rehr_wgrad_bf16(const rehr_wgrad_desc* dp, void* stream) {
  if (!dp) return REHR_EINVAL;
  WGParams p;
  int rc = plan_bf16(*dp, p);
  if (rc != REHR_OK) return rc;
  BrickBf16 bo;
  // unit-stride 3x3(x3) taps: both operands as LDS bricks (REHR_DBG_WGRAD_DIRECT keeps the per-tap slab kernel)
  const bool brick = !(dp->debug_flags & REHR_DBG_WGRAD_DIRECT) && wgrad_brick_bf16_plan(*dp, p, bo);
  const rehr_wgrad_desc& d = p.d;
  if (!d.workspace || d.workspace_bytes < (int64_t)p.splits * p.T * p.Capad * p.Cgpad * (int64_t)sizeof(float))
    return REHR_EINVAL;
  if (p.splits > 65535) return REHR_EINVAL;
  p.slab_bias = nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (brick) {
    rc = wgrad_brick_bf16_launch(p, bo, st);
  } else {
    const int t = p.Capad / p.a_tiles;
    if (t == 128) rc = launch_wg_bf16<128, 2, 2, 64>(p, st);
    else if (t == 64) rc = launch_wg_bf16<64, 1, 1, 64>(p, st);
    else rc = launch_wg_bf16<32, 1, 1, 64>(p, st);
  }
  if (rc != REHR_OK) return rc;
  launch_wgrad_reduce(p, st);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int64_t rehr_wgrad_workspace_bytes(const rehr_wgrad_desc* dp) {
  if (!dp) return REHR_EINVAL;
  WGParams p;
  rehr_wgrad_desc d = *dp;
  if (!d.dst) d.dst = reinterpret_cast<float*>(16);  // size query may come before allocation
  const int rc = plan(d, p);
  if (rc != REHR_OK) return rc;
  const int64_t wino = wino_wgrad_workspace_bytes(d);
  if (wino > 0) return wino;
  const int64_t wino22 = wino22_wgrad_workspace_bytes(d);
  if (wino22 > 0) return wino22;
  BrickPlanOut bo;
  WGParams pb = p;
  if (wgrad_brick_plan(d, pb, bo)) return ws_bytes(pb);
  return ws_bytes(p);
}

extern "C" int rehr_wgrad_uses_winograd(const rehr_wgrad_desc* dp) {
  if (!dp) return 0;
  WGParams p;
  rehr_wgrad_desc d = *dp;
  if (!d.dst) d.dst = reinterpret_cast<float*>(16);
  if (plan(d, p) != REHR_OK) return 0;
  return (wino_wgrad_workspace_bytes(d) > 0 || wino22_wgrad_workspace_bytes(d) > 0) ? 1 : 0;
}

extern "C" int rehr_wgrad_f32(const rehr_wgrad_desc* dp, void* stream) {
  if (!dp) return REHR_EINVAL;
  WGParams p;
  int rc = plan(*dp, p);
  if (rc != REHR_OK) return rc;
  rc = wino_wgrad_try(*dp, (hipStream_t)stream);  // 2.25x fewer multiplications where it applies
  if (rc != REHR_ENOSUP) return rc;
  rc = wino22_wgrad_try(*dp, (hipStream_t)stream);  // stride-2 4-tap transposed convs: 1.78x fewer
  if (rc != REHR_ENOSUP) return rc;
  BrickPlanOut bo;
  const bool brick = wgrad_brick_plan(*dp, p, bo);  // overrides tiles / splits / slab geometry when it applies
  const rehr_wgrad_desc& d = p.d;
  if (!d.workspace || d.workspace_bytes < ws_bytes(p)) return REHR_EINVAL;
  if (p.splits > 65535) return REHR_EINVAL;
  p.slab_bias = d.dbias ? d.workspace + (int64_t)p.splits * p.T * p.Capad * p.Cgpad : nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (brick) {
    rc = wgrad_brick_launch(p, bo, st);
  } else {
    const int t = p.Capad / p.a_tiles;
    if (t == 128) rc = launch_wg<128, 2, 2, 32>(p, st);
    else if (t == 64) rc = launch_wg<64, 1, 1, 64>(p, st);
    else rc = launch_wg<32, 1, 1, 64>(p, st);
  }
  if (rc != REHR_OK) return rc;
  launch_wgrad_reduce(p, st);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
