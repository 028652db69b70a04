// Gather-GEMM on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// One kernel family computes every wide-channel convolution of the hot path:
// Conv3d forward, Conv3d input gradient (per stride phase), ConvTranspose3d
// forward (per stride phase) and its input gradient, Conv2d (D == 1) -- see
// rehr_gather_gemm_desc in include/rehrseg_hip.h for the contraction.
//
// Design (MI355X first):
//  * implicit GEMM, M = 128 lattice voxels of one sample (a td x th x tw brick
//    so the 27-tap halo of a tile stays in the XCD's L2), N = 32/64/128 output
//    channels, K = taps x 32-channel chunks.  NDHWC makes every A row a
//    contiguous 128-byte line; taps of one chunk are visited back to back so
//    the shifted re-reads hit L1/L2, not HBM.
//  * fp32 MFMA is exact fp32 (fmaf chain) at the vector-peak rate, 64 cycles per
//    32x32x2 instruction; one accumulator chain per wave already saturates the
//    pipe, so a plain register-staged double buffer (global -> VGPR -> LDS, one
//    barrier per K step) is enough to keep it fed.
//  * both operands sit in LDS as [row][32 k] with a 36-float row stride:
//    ds_read_b128 of 4 consecutive k per lane is bank-conflict free, and one
//    128-bit read feeds four MFMA k-steps (the k order inside a step is the
//    same permutation for A and B, which a dot product does not care about).
//  * epilogue fuses bias, ReLU/LeakyReLU and the per-(sample, channel) sum /
//    sum-of-squares that SEGating's pool and InstanceNorm3d need (double
//    atomics, one per channel per wave).
#include "common.h"
#include "halo_conv.h"
#include "wino_conv.h"

namespace {

constexpr int BK = 32;       // channels per K step
constexpr int LDS_LD = 36;   // floats per LDS row (32 + 4 pad)
constexpr int NTHREADS = 256;

struct GGParams {
  rehr_gather_gemm_desc d;
  int tiles_d, tiles_h, tiles_w, m_tiles, n_tiles;
  int kchunks;  // Cin / 32
  int64_t wp_bytes;
};

// LDSBUF = 2: double-buffered LDS tiles, one barrier per K step (register-heavy 128x128 tile,
//             2 blocks per CU anyway).
// LDSBUF = 1: one LDS tile + the two register sets as the pipeline, two barriers per K step;
//             halves LDS so the narrow-N tiles run 3 blocks per CU, whose MFMA phases fill each
//             other's barrier bubbles.
template <int BM, int BN, int WGM, int WGN, int LDSBUF>
__device__ __forceinline__ void gather_gemm_body(const GGParams& p, const int nblocks, const int logical_in = -1) {
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int FM = WTM / 32, FN = WTN / 32;
  constexpr int AROWS = BM / 32, BROWS = BN / 32;  // rows per thread per tile
  static_assert(WGM * WGN == 4, "4 waves");
  const rehr_gather_gemm_desc& d = p.d;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                    // [LDSBUF][BM][LDS_LD]
  float* Bs = smem + LDSBUF * BM * LDS_LD;             // [LDSBUF][BN][LDS_LD]
  int* row_out = (int*)(Bs + LDSBUF * BN * LDS_LD);    // [BM] destination voxel or -1

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int n_img = blockIdx.y;

  const int logical = logical_in >= 0 ? logical_in : xcd_remap(blockIdx.x, nblocks);
  const int mt = logical / p.n_tiles;
  const int nt = logical - mt * p.n_tiles;
  const int n0 = nt * BN;

  // tile origin on the lattice (brick mode), or flattened run (tile_d == 0)
  const bool linear = d.tile_d == 0;
  const int tx = mt % p.tiles_w;
  const int ty = (mt / p.tiles_w) % p.tiles_h;
  const int tz = mt / (p.tiles_w * p.tiles_h);
  const int thw = linear ? 1 : d.tile_h * d.tile_w;
  const int lhw = d.Lh * d.Lw;
  auto row_coords = [&](int r, int& od, int& oh, int& ow) -> bool {
    if (linear) {
      const int flat = mt * BM + r;
      od = flat / lhw;
      const int rem = flat - od * lhw;
      oh = rem / d.Lw;
      ow = rem - oh * d.Lw;
      return od < d.Ld;
    }
    const int ld_ = r / thw, rem = r - ld_ * thw;
    const int lh_ = rem / d.tile_w, lw_ = rem - lh_ * d.tile_w;
    od = tz * d.tile_d + ld_;
    oh = ty * d.tile_h + lh_;
    ow = tx * d.tile_w + lw_;
    return od < d.Ld && oh < d.Lh && ow < d.Lw;
  };

  // destination offsets of the tile rows (used by the epilogue)
  if (tid < BM) {
    int od, oh, ow;
    int off = -1;
    if (row_coords(tid, od, oh, ow)) {
      const int yd = od * d.osd + d.obd, yh = oh * d.osh + d.obh, yw = ow * d.osw + d.obw;
      off = ((n_img * d.Dy + yd) * d.Hy + yh) * d.Wy + yw;
    }
    row_out[tid] = off;
  }

  // per-thread gather state: rows r0 + 32*i, 16-byte chunk q of the 128-byte row
  const int q = tid & 7, r0 = tid >> 3;
  int sd0[AROWS], sh0[AROWS], sw0[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    int od, oh, ow;
    const bool ok = row_coords(r0 + 32 * i, od, oh, ow);
    // an invalid row gets a base far outside the source so every tap misses
    sd0[i] = ok ? od * d.sd + d.bd : -(1 << 28);
    sh0[i] = oh * d.sh + d.bh;
    sw0[i] = ow * d.sw + d.bw;
  }
  const int64_t img_vox = (int64_t)n_img * d.Di * d.Hi * d.Wi;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Taps that no row of this tile can reach are skipped outright (block-uniform):
  // boundary tiles drop their padding taps, and a contraction whose source is thin
  // along an axis (feature_fuse's input gradient: depth-1 dY against 128 depth taps)
  // does 1/128 of the nominal work.
  int olo[3], ohi[3];
  if (linear) {
    const int f0 = mt * BM;
    int f1 = f0 + BM - 1;
    if (f1 > d.Ld * lhw - 1) f1 = d.Ld * lhw - 1;
    const int d0 = f0 / lhw, d1 = f1 / lhw;
    olo[0] = d0; ohi[0] = d1;
    olo[1] = 0; ohi[1] = d.Lh - 1; olo[2] = 0; ohi[2] = d.Lw - 1;
    if (d0 == d1) {
      const int h0 = (f0 - d0 * lhw) / d.Lw, h1 = (f1 - d0 * lhw) / d.Lw;
      olo[1] = h0; ohi[1] = h1;
      if (h0 == h1) { olo[2] = f0 - d0 * lhw - h0 * d.Lw; ohi[2] = f1 - d0 * lhw - h0 * d.Lw; }
    }
  } else {
    olo[0] = tz * d.tile_d; ohi[0] = min(olo[0] + d.tile_d, d.Ld) - 1;
    olo[1] = ty * d.tile_h; ohi[1] = min(olo[1] + d.tile_h, d.Lh) - 1;
    olo[2] = tx * d.tile_w; ohi[2] = min(olo[2] + d.tile_w, d.Lw) - 1;
  }
  auto clip = [](const rehr_axis_taps& t, int s, int b, int lo, int hi, int size, int& j0, int& j1) {
    j0 = t.count; j1 = -1;
    const int plo = lo * s + b + t.off0, phi = hi * s + b + t.off0;
    for (int j = 0; j < t.count; ++j) {
      const int a = plo + t.offs * j, c = phi + t.offs * j;  // source interval of the tile for tap j
      if (c >= 0 && a <= size - 1) { if (j < j0) j0 = j; j1 = j; }
    }
  };
  int jd0, jd1, jh0, jh1, jw0, jw1;
  clip(d.td, d.sd, d.bd, olo[0], ohi[0], d.Di, jd0, jd1);
  clip(d.th, d.sh, d.bh, olo[1], ohi[1], d.Hi, jh0, jh1);
  clip(d.tw, d.sw, d.bw, olo[2], ohi[2], d.Wi, jw0, jw1);
  const bool any_tap = jd1 >= jd0 && jh1 >= jh0 && jw1 >= jw0;

  // K-step iterator: channel chunk outermost, then jd, jh, jw (innermost)
  int cc = 0, jd = jd0, jh = jh0, jw = jw0;
  const int nsteps = any_tap ? p.kchunks * (jd1 - jd0 + 1) * (jh1 - jh0 + 1) * (jw1 - jw0 + 1) : 0;

  // two register sets: tiles are fetched TWO K-steps ahead, so a load that misses L2
  // (HBM latency, ~2x a K step under load) does not stall the block at the barrier
  f32x4 ra0[AROWS], rb0[BROWS], ra1[AROWS], rb1[BROWS];

  // Branch-free gather: raw buffer loads return 0 for an offset >= num_records, so a
  // tap that leaves the source (zero padding) is just an out-of-range offset.  The
  // whole K-step body is then one basic block and the address arithmetic of the next
  // tile schedules into the shadow of the current tile's MFMAs.
  const uint32_t img_elems = (uint32_t)d.Di * d.Hi * d.Wi;
  const uint32_t nrec1 = img_elems * (uint32_t)d.ldx1 * 4u;
  const uint32_t nrec2 = d.x2 ? img_elems * (uint32_t)d.ldx2 * 4u : nrec1;
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.x1) + img_vox * d.ldx1, 0, nrec1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = d.x2 ? __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.x2) + img_vox * d.ldx2, 0, nrec2, 0x00020000) : rs1;
  const uint32_t nrecw = (uint32_t)p.wp_bytes;
  const __amdgpu_buffer_rsrc_t rsw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.wp), 0, nrecw, 0x00020000);
  int rowvox[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i)
    rowvox[i] = (sd0[i] < 0 && sd0[i] < -(1 << 20)) ? 0 : (sd0[i] * d.Hi + sh0[i]) * d.Wi + sw0[i];
  const uint32_t brow_off = (uint32_t)(n0 + r0) * d.Cin * 4u + q * 16u;

  auto issue_loads = [&](f32x4 (&ra)[AROWS], f32x4 (&rb)[BROWS]) {
    const int dd = d.td.off0 + d.td.offs * jd;
    const int dh = d.th.off0 + d.th.offs * jh;
    const int dw = d.tw.off0 + d.tw.offs * jw;
    const int tapvox = (dd * d.Hi + dh) * d.Wi + dw;
    const bool first = cc < d.c1;
    const __amdgpu_buffer_rsrc_t rs = first ? rs1 : rs2;
    const uint32_t ldb = (uint32_t)(first ? d.ldx1 : d.ldx2) * 4u;
    const uint32_t cb = (uint32_t)((first ? cc : cc - d.c1) + q * 4) * 4u;
    const uint32_t oob = first ? nrec1 : nrec2;
    const bool kok = (cc + q * 4) < d.Cin;  // Cin % 16 == 0: the last 32-chunk may be half empty
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int id = sd0[i] + dd, ih = sh0[i] + dh, iw = sw0[i] + dw;
      const bool inb = ((unsigned)id < (unsigned)d.Di) & ((unsigned)ih < (unsigned)d.Hi) &
                       ((unsigned)iw < (unsigned)d.Wi);  // bitwise: no short-circuit branches
      const uint32_t lin = (uint32_t)(rowvox[i] + tapvox) * ldb + cb;
      const uint32_t off = (inb & kok) ? lin : oob;
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
    const int wt = ((d.td.k0 + d.td.ks * jd) * d.KH + (d.th.k0 + d.th.ks * jh)) * d.KW +
                   (d.tw.k0 + d.tw.ks * jw);
    const uint32_t woff = ((uint32_t)wt * d.Npad * d.Cin + cc) * 4u + brow_off;
#pragma unroll
    for (int i = 0; i < BROWS; ++i)
      rb[i] = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                     rsw, kok ? woff + (uint32_t)(32 * i) * d.Cin * 4u : nrecw, 0, 0));
    // advance iterator (scalar)
    ++jw;
    const bool cw = jw > jw1;
    jw = cw ? jw0 : jw;
    jh += cw ? 1 : 0;
    const bool ch = jh > jh1;
    jh = ch ? jh0 : jh;
    jd += ch ? 1 : 0;
    const bool cd = jd > jd1;
    jd = cd ? jd0 : jd;
    cc += cd ? BK : 0;
  };
  auto commit_loads = [&](int buf, const f32x4 (&ra)[AROWS], const f32x4 (&rb)[BROWS]) {
    float* a = As + buf * BM * LDS_LD;
    float* b = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int i = 0; i < AROWS; ++i)
      *reinterpret_cast<f32x4*>(a + (r0 + 32 * i) * LDS_LD + q * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i)
      *reinterpret_cast<f32x4*>(b + (r0 + 32 * i) * LDS_LD + q * 4) = rb[i];
  };

  const int arow = wm * WTM + (lane & 31);
  const int brow = wn * WTN + (lane & 31);
  const int koff = 4 * (lane >> 5);
  auto compute = [&](int buf) {
    const float* a = As + buf * BM * LDS_LD;
    const float* b = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 fa[FM], fb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        fa[i] = *reinterpret_cast<const f32x4*>(a + (arow + 32 * i) * LDS_LD + kk * 8 + koff);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        fb[j] = *reinterpret_cast<const f32x4*>(b + (brow + 32 * j) * LDS_LD + kk * 8 + koff);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
      // the fetch issued above must leave in the shadow of the first MFMA group,
      // not be sunk behind the whole tile
      if (kk == 0) __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (nsteps > 0) {
    issue_loads(ra0, rb0);  // tile 0
    issue_loads(ra1, rb1);  // tile 1 (past the end: out-of-range / never used)
    commit_loads(0, ra0, rb0);
  }
  __syncthreads();

  // Invariant at the top: tile s is staged in LDS[0], tile s+1 is in flight in set 1.
  // Fetches run past the last tile (the iterator is then beyond Cin: such loads are
  // out of range or harmless and never consumed), which keeps each half branch-free.
  if (LDSBUF == 2) {
    for (int s = 0; s < nsteps; s += 2) {
      issue_loads(ra0, rb0);            // tile s+2
      compute(0);
      commit_loads(1, ra1, rb1);        // tile s+1
      __syncthreads();
      if (s + 1 >= nsteps) break;
      issue_loads(ra1, rb1);            // tile s+3
      compute(1);
      commit_loads(0, ra0, rb0);        // tile s+2
      __syncthreads();
    }
  } else {
    for (int s = 0; s < nsteps; s += 2) {
      issue_loads(ra0, rb0);            // tile s+2
      compute(0);
      __syncthreads();                  // every wave is done reading tile s
      commit_loads(0, ra1, rb1);        // tile s+1
      __syncthreads();
      if (s + 1 >= nsteps) break;
      issue_loads(ra1, rb1);            // tile s+3
      compute(0);
      __syncthreads();
      commit_loads(0, ra0, rb0);        // tile s+2
      __syncthreads();
    }
  }

  // ---- epilogue: bias + activation + store (+ statistics) ----
  const int chalf = lane >> 5;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int col = n0 + wn * WTN + j * 32 + (lane & 31);
    const bool colok = col < d.Cout;
    const float bv = (d.bias != nullptr && colok) ? d.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
        const int off = row_out[row];
        float v = apply_act(acc[i][j][r] + bv, d.act, d.slope);
        if (off >= 0 && colok) {
          d.y[(int64_t)off * d.ldy + col] = v;
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
}

constexpr int MAX_PHASES = 8;
// Several launches that differ only in lattice / taps / destination offset (the stride
// phases of one transposed conv or strided input gradient) share ONE grid: blockIdx.z picks
// the phase, so four quarter-size launches fill the chip like one full-size launch.
// interleave != 0 (REHR_DBG_GG_INTERLEAVE; all phases have the same tile counts -- kernel = stride transposed
// convolutions, input gradients of strided convolutions on even extents): a 1-D grid in which the `count` phases of one
// lattice tile are CONSECUTIVE blocks of ONE XCD, so that the tile's source rows come from HBM once and from that XCD's L2
// for the other phases (with blockIdx.z = phase the source tensor is streamed once per phase).  Tried in round 3 and
// measured SLOWER in the step (cfg-3 +1.4 ms, cfg-5 +0.25 ms, profiles/r03_ab_phase_interleave.txt): these launches are
// bound by block turnover (K = C_in only: two k-steps per block), not by the source re-reads, and eight blocks storing
// into the same 2x2x2 output neighbourhood at once serialise at the memory side.  Off by default; tests keep it alive.
struct GGMulti {
  GGParams ph[MAX_PHASES];
  int interleave, count, no_interleave;
};

// block b of the interleaved grid -> (phase, logical tile); false: padding block
__device__ __forceinline__ bool interleaved_block(int b, int m_tiles, int n_tiles, int count, int& phase, int& logical) {
  const int xcd = b & 7, j = b >> 3, per = count * n_tiles;
  const int mt = (j / per) * 8 + xcd, rem = j % per;
  phase = rem / n_tiles;
  logical = mt * n_tiles + (rem - phase * n_tiles);
  return mt < m_tiles;
}

template <int BM, int BN, int WGM, int WGN, int LDSBUF>
__global__ __launch_bounds__(NTHREADS, (LDSBUF == 1 ? 3 : 2)) void gather_gemm_multi_kernel(const GGMulti pm) {
  int phase, logical;
  if (pm.interleave) {
    if (!interleaved_block((int)blockIdx.x, pm.ph[0].m_tiles, pm.ph[0].n_tiles, pm.count, phase, logical)) return;
  } else {
    phase = blockIdx.z;
    const int nb = pm.ph[phase].m_tiles * pm.ph[phase].n_tiles;
    if ((int)blockIdx.x >= nb) return;  // block-uniform: phases have different tile counts
    logical = xcd_remap(blockIdx.x, nb);
  }
  // (block-uniform, but computed with VALU divisions: pinned to SGPRs so that pm.ph[phase] stays a scalar kernarg load;
  // ONE call site: the body is register-tight and must not be inlined twice)
  phase = __builtin_amdgcn_readfirstlane(phase);
  logical = __builtin_amdgcn_readfirstlane(logical);
  gather_gemm_body<BM, BN, WGM, WGN, LDSBUF>(pm.ph[phase], 0, logical);
}
// single launch: parameters at fixed kernarg offsets (the dynamically indexed form above costs
// a few VGPR spills in the register-tight 128x128 tile)
template <int BM, int BN, int WGM, int WGN, int LDSBUF>
__global__ __launch_bounds__(NTHREADS, (LDSBUF == 1 ? 3 : 2)) void gather_gemm_kernel(const GGParams p) {
  gather_gemm_body<BM, BN, WGM, WGN, LDSBUF>(p, (int)gridDim.x);
}

template <int BM, int BN, int WGM, int WGN, int LDSBUF>
int launch_gg(const GGMulti& pm, int count, hipStream_t stream) {
  const size_t smem = (size_t)LDSBUF * (BM + BN) * LDS_LD * sizeof(float) + BM * sizeof(int);
  static bool attr_set = false;
  auto kern1 = gather_gemm_kernel<BM, BN, WGM, WGN, LDSBUF>;
  auto kernm = gather_gemm_multi_kernel<BM, BN, WGM, WGN, LDSBUF>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern1), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernm), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess)
      return REHR_EHIP;
    attr_set = true;
  }
  if (count == 1) {
    const GGParams& p = pm.ph[0];
    hipLaunchKernelGGL(kern1, dim3(p.m_tiles * p.n_tiles, p.d.N, 1), dim3(NTHREADS), smem, stream, p);
  } else {
    int nb = 0;
    bool uniform = true;
    for (int i = 0; i < count; ++i) {
      const int n = pm.ph[i].m_tiles * pm.ph[i].n_tiles;
      nb = n > nb ? n : nb;
      uniform = uniform && pm.ph[i].m_tiles == pm.ph[0].m_tiles && pm.ph[i].n_tiles == pm.ph[0].n_tiles;
    }
    GGMulti pmi = pm;
    pmi.count = count;
    const int64_t gx = (int64_t)((pm.ph[0].m_tiles + 7) / 8) * 8 * count * pm.ph[0].n_tiles;
    pmi.interleave = (uniform && !pm.no_interleave && gx < (1ll << 31)) ? 1 : 0;
    if (pmi.interleave) hipLaunchKernelGGL(kernm, dim3((unsigned)gx, pm.ph[0].d.N, 1), dim3(NTHREADS), smem, stream, pmi);
    else hipLaunchKernelGGL(kernm, dim3(nb, pm.ph[0].d.N, count), dim3(NTHREADS), smem, stream, pmi);
  }
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

bool taps_ok(const rehr_axis_taps& t) { return t.count >= 1; }

int validate(const rehr_gather_gemm_desc& d) {
  if (!d.x1 || !d.wp || !d.y) return REHR_EINVAL;
  if (d.N < 1 || d.Cin < 16 || d.Cin % 16 || d.c1 < 0 || d.c1 > d.Cin) return REHR_EINVAL;
  if (d.c1 < d.Cin && d.c1 % 32) return REHR_EINVAL;  // a virtual concat splits on a chunk boundary
  if (d.c1 < d.Cin && !d.x2) return REHR_EINVAL;
  if (d.c1 == 0) return REHR_EINVAL;
  if (d.ldx1 % 4 || (d.x2 && d.ldx2 % 4)) return REHR_EINVAL;
  if (((uintptr_t)d.x1 | (uintptr_t)d.wp | (uintptr_t)(d.x2 ? d.x2 : d.x1)) & 15) return REHR_EINVAL;
  if (d.Npad % 32 || d.Npad < d.Cout || d.Cout < 1) return REHR_EINVAL;
  if (d.Ld < 1 || d.Lh < 1 || d.Lw < 1) return REHR_EINVAL;
  if (!taps_ok(d.td) || !taps_ok(d.th) || !taps_ok(d.tw)) return REHR_EINVAL;
  if (d.tile_d != 0 &&
      (d.tile_d < 1 || d.tile_h < 1 || d.tile_w < 1 || d.tile_d * d.tile_h * d.tile_w != 128))
    return REHR_EINVAL;
  if (d.stats_mode != 0 && !d.stats) return REHR_EINVAL;
  if (d.N > 65535) return REHR_EINVAL;
  // destination extent check: the last lattice point must land inside y
  const int64_t yd = (int64_t)(d.Ld - 1) * d.osd + d.obd, yh = (int64_t)(d.Lh - 1) * d.osh + d.obh,
                yw = (int64_t)(d.Lw - 1) * d.osw + d.obw;
  if (d.obd < 0 || d.obh < 0 || d.obw < 0 || yd >= d.Dy || yh >= d.Hy || yw >= d.Wy) return REHR_EINVAL;
  if (d.ldy < d.Cout) return REHR_EINVAL;
  if ((int64_t)d.N * d.Dy * d.Hy * d.Wy >= (1ll << 31)) return REHR_EINVAL;
  return REHR_OK;
}

int plan(const rehr_gather_gemm_desc& d, GGParams& p) {
  p.d = d;
  if (d.tile_d == 0) {
    p.tiles_d = p.tiles_h = 1;
    p.tiles_w = (int)(((int64_t)d.Ld * d.Lh * d.Lw + 127) / 128);
    p.m_tiles = p.tiles_w;
  } else {
    p.tiles_d = (d.Ld + d.tile_d - 1) / d.tile_d;
    p.tiles_h = (d.Lh + d.tile_h - 1) / d.tile_h;
    p.tiles_w = (d.Lw + d.tile_w - 1) / d.tile_w;
    p.m_tiles = p.tiles_d * p.tiles_h * p.tiles_w;
  }
  p.kchunks = (d.Cin + 31) / 32;
  // buffer-addressed operands: 32-bit byte offsets per sample / per weight panel
  const int64_t kd_max = d.td.k0 + (int64_t)d.td.ks * (d.td.count - 1);
  const int64_t kh_max = d.th.k0 + (int64_t)d.th.ks * (d.th.count - 1);
  const int64_t kw_max = d.tw.k0 + (int64_t)d.tw.ks * (d.tw.count - 1);
  const int64_t taps_all = ((kd_max * d.KH) + kh_max) * d.KW + kw_max + 1;
  p.wp_bytes = taps_all * d.Npad * d.Cin * 4;
  const int64_t img = (int64_t)d.Di * d.Hi * d.Wi * 4;
  if (p.wp_bytes >= (1ll << 32) - 64 || img * d.ldx1 >= (1ll << 32) - 64 ||
      (d.x2 && img * d.ldx2 >= (1ll << 32) - 64))
    return REHR_ENOSUP;
  p.n_tiles = d.Npad / (d.Npad % 128 == 0 ? 128 : (d.Npad % 64 == 0 ? 64 : 32));
  return REHR_OK;
}

int launch_generic(const GGMulti& pm, int count, hipStream_t st) {
  const int npad = pm.ph[0].d.Npad;
  if (npad % 128 == 0) return launch_gg<128, 128, 2, 2, 2>(pm, count, st);
  if (npad % 64 == 0) return launch_gg<128, 64, 2, 2, 1>(pm, count, st);
  return launch_gg<128, 32, 4, 1, 1>(pm, count, st);
}

}  // namespace

extern "C" int rehr_gather_gemm_multi_f32(const rehr_gather_gemm_desc* descs, int32_t count, void* stream) {
  if (descs == nullptr || count < 1 || count > MAX_PHASES) return REHR_EINVAL;
  for (int i = 0; i < count; ++i) {
    const int rc = validate(descs[i]);
    if (rc != REHR_OK) return rc;
    // phases of one layer: same operands and channel geometry
    if (descs[i].Npad != descs[0].Npad || descs[i].N != descs[0].N || descs[i].wp != descs[0].wp ||
        descs[i].x1 != descs[0].x1)
      return REHR_EINVAL;  // (y may differ: split-K partials go to separate slabs)
  }
  hipStream_t st = (hipStream_t)stream;
  // REHR_GG_WS_ONLY: the weight transforms of this call and nothing else (every Winograd try-function below returns
  // right behind its transform launch; whatever no Winograd kernel takes has nothing to prepare)
  const bool ws_only = (descs[0].flags & REHR_GG_WS_ONLY) != 0;
  for (int i = 1; i < count; ++i)
    if ((descs[i].flags ^ descs[0].flags) & (REHR_GG_WS_ONLY | REHR_GG_WS_READY)) return REHR_EINVAL;
  if (count > 1 && !ws_only) {   // the stride phases of a kernel == stride transposed convolution: one fused launch
    const int trc = tconv_ks_try(descs, count, false, st);
    if (trc != REHR_ENOSUP) return trc;
  }
  if (count > 1 && descs[0].wino_ws != nullptr && (descs[0].td.count > 3 || descs[0].td.count == 1)) {
    // tap-range parts of a split-K launch (many depth taps; or the single depth taps of a layer whose plain Winograd
    // grid would leave half the chip idle): all parts in one Winograd grid, or none
    const int src = wino_conv_split_try(descs, count, st);
    if (src != REHR_ENOSUP) return src;
  }
  if (count > 1 && descs[0].wino_ws != nullptr && descs[0].td.count <= 3) {
    // output phases of a transposed convolution on small planes: one grid of the flattened-tile F(2x2,2x2) kernel
    const int frc = wino22_flat_multi_try(descs, count, st);
    if (frc != REHR_ENOSUP) return frc;
  }
  GGMulti pm;
  pm.interleave = 0;
  pm.count = 0;
  pm.no_interleave = (descs[0].debug_flags & REHR_DBG_GG_INTERLEAVE) ? 0 : 1;
  int n = 0;
  for (int i = 0; i < count; ++i) {
    // fewer multiplications beat better tiling: Winograd first -- except for the tap-range parts of a split-K
    // launch (many depth taps, few tiles): one Winograd launch per part would run them one after the other on a
    // quarter of the chip, the generic kernel runs all parts in one grid
    if (descs[i].wino_ws != nullptr && !(count > 1 && descs[i].td.count > 3)) {
      int wrc = wino_flat8_conv_try(descs[i], st);
      if (wrc == REHR_ENOSUP) wrc = wino_conv_try(descs[i], st);
      if (wrc == REHR_ENOSUP) wrc = wino22_conv_try(descs[i], st);
      if (wrc == REHR_OK) continue;
      if (wrc != REHR_ENOSUP) return wrc;
    }
    if (ws_only) continue;
    if (descs[i].tile_d >= 0) {  // the halo-tile kernel takes what it is good at, one launch each
      const int hrc = halo_conv_try(descs[i], st);
      if (hrc == REHR_OK) continue;
      if (hrc != REHR_ENOSUP) return hrc;
    }
    const int rc = plan(descs[i], pm.ph[n]);
    if (rc != REHR_OK) return rc;
    ++n;
  }
  if (n == 0) return REHR_OK;
  return launch_generic(pm, n, st);
}

extern "C" int64_t rehr_gather_gemm_wino_bytes(const rehr_gather_gemm_desc* dp) {
  if (dp == nullptr || validate(*dp) != REHR_OK) return 0;
  int64_t b = wino_flat8_workspace_bytes(*dp);
  if (b == 0) b = wino_workspace_bytes(*dp);
  if (b == 0) b = wino22_workspace_bytes(*dp);
  return b;
}

extern "C" int rehr_gather_gemm_f32(const rehr_gather_gemm_desc* dp, void* stream) {
  if (dp == nullptr) return REHR_EINVAL;
  return rehr_gather_gemm_multi_f32(dp, 1, stream);
}
