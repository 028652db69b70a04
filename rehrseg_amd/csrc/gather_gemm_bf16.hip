// Gather-GEMM on the bf16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the mixed-precision
// path of BASELINE.json configs[4] (bf16 conv inputs / weights, fp32 accumulation, fp32 statistics).
//
// Same contraction, descriptor and tile geometry as gather_gemm.hip (rehr_gather_gemm_desc: Conv3d forward, Conv3d
// input gradient per stride phase, ConvTranspose3d forward per phase and its input gradient, virtual channel
// concat, fused bias + ReLU/LeakyReLU + per-(sample, channel) sum / sum of squares) with bf16 operands:
//   x1, x2, wp   bf16 (NDHWC activations; packed panel wp[tap][Npad][Cin] from rehr_pack_weights_bf16)
//   y            bf16, or fp32 with REHR_GG_Y_F32 in flags
//   bias fp32, stats fp64 -- the statistics are formed from the fp32 accumulators, before rounding to bf16.
// The bf16 pipe runs 16x the fp32 pipe, so there is no Winograd variant here: the direct contraction is already
// operand-bandwidth bound (a 128x128x64 K step is 16 MFMAs = 512 cycles per wave against 32 KiB of tile loads).
//
//  * M = 128 lattice voxels (td x th x tw brick or 128 flattened voxels), N = 32/64/128 channels,
//    K step = 64 channels of one tap (128-byte rows, the same byte geometry as the fp32 kernel's 32 floats) or
//    32 channels (64-byte rows) when Cin or the concat split is not a multiple of 64.
//  * LDS rows are padded by 16 bytes: the ds_read_b128 of a lane's 8 consecutive k (its 32x32x16 fragment) is
//    conflict-free; one read feeds one MFMA.
//  * register-staged pipeline two K steps ahead, branch-free raw buffer loads (out-of-range = zero padding), as
//    in the fp32 kernel.
#include "common.h"
#include "wino_conv.h"

int halo_conv_bf16_try(const rehr_gather_gemm_desc& d, hipStream_t stream);  // halo_conv_bf16.hip

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NTHREADS = 256;
constexpr int ES = 2;  // bytes per element

struct GBParams {
  rehr_gather_gemm_desc d;
  int tiles_d, tiles_h, tiles_w, m_tiles, n_tiles;
  int kchunks;
  int64_t wp_bytes;
};

template <int BM, int BN, int WGM, int WGN, int LDSBUF, int BK>
__device__ __forceinline__ void gg_bf16_body(const GBParams& p, const int nblocks, const int logical_in = -1) {
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int FM = WTM / 32, FN = WTN / 32;
  constexpr int TPR = BK / 8;                 // threads per row (16 bytes = 8 channels each)
  constexpr int RPP = NTHREADS / TPR;         // rows per pass
  constexpr int AROWS = BM / RPP, BROWS = (BN + RPP - 1) / RPP;
  constexpr int ROWB = BK * ES + 16;          // LDS row stride in bytes
  static_assert(WGM * WGN == 4, "4 waves");
  const rehr_gather_gemm_desc& d = p.d;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* As = smem_b;                                    // [LDSBUF][BM][ROWB]
  unsigned char* Bs = smem_b + LDSBUF * BM * ROWB;               // [LDSBUF][BN][ROWB]
  int* row_out = (int*)(Bs + LDSBUF * BN * ROWB);                // [BM] destination voxel or -1

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int n_img = blockIdx.y;

  const int logical = logical_in >= 0 ? logical_in : xcd_remap(blockIdx.x, nblocks);
  const int mt = logical / p.n_tiles;
  const int nt = logical - mt * p.n_tiles;
  const int n0 = nt * BN;

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

  if (tid < BM) {
    int od, oh, ow;
    int off = -1;
    if (row_coords(tid, od, oh, ow)) {
      const int yd = od * d.osd + d.obd, yh = oh * d.osh + d.obh, yw = ow * d.osw + d.obw;
      off = ((n_img * d.Dy + yd) * d.Hy + yh) * d.Wy + yw;
    }
    row_out[tid] = off;
  }

  const int q = tid % TPR, r0 = tid / TPR;
  int sd0[AROWS], sh0[AROWS], sw0[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    int od, oh, ow;
    const bool ok = row_coords(r0 + RPP * i, od, oh, ow);
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

  // taps no row of this tile can reach are skipped (block-uniform)
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
      const int a = plo + t.offs * j, c = phi + t.offs * j;
      if (c >= 0 && a <= size - 1) { if (j < j0) j0 = j; j1 = j; }
    }
  };
  int jd0, jd1, jh0, jh1, jw0, jw1;
  clip(d.td, d.sd, d.bd, olo[0], ohi[0], d.Di, jd0, jd1);
  clip(d.th, d.sh, d.bh, olo[1], ohi[1], d.Hi, jh0, jh1);
  clip(d.tw, d.sw, d.bw, olo[2], ohi[2], d.Wi, jw0, jw1);
  const bool any_tap = jd1 >= jd0 && jh1 >= jh0 && jw1 >= jw0;

  int cc = 0, jd = jd0, jh = jh0, jw = jw0;
  const int nsteps = any_tap ? p.kchunks * (jd1 - jd0 + 1) * (jh1 - jh0 + 1) * (jw1 - jw0 + 1) : 0;

  u32x4 ra0[AROWS], rb0[BROWS], ra1[AROWS], rb1[BROWS];

  const uint32_t img_elems = (uint32_t)d.Di * d.Hi * d.Wi;
  const uint32_t nrec1 = img_elems * (uint32_t)d.ldx1 * ES;
  const uint32_t nrec2 = d.x2 ? img_elems * (uint32_t)d.ldx2 * ES : nrec1;
  const __bf16* x1 = reinterpret_cast<const __bf16*>(d.x1);
  const __bf16* x2 = reinterpret_cast<const __bf16*>(d.x2);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(x1) + img_vox * d.ldx1, 0, nrec1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = d.x2 ? __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(x2) + img_vox * d.ldx2, 0, nrec2, 0x00020000) : rs1;
  const uint32_t nrecw = (uint32_t)p.wp_bytes;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(reinterpret_cast<const __bf16*>(d.wp)), 0, nrecw, 0x00020000);
  int rowvox[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i)
    rowvox[i] = (sd0[i] < -(1 << 20)) ? 0 : (sd0[i] * d.Hi + sh0[i]) * d.Wi + sw0[i];
  const uint32_t brow_off = (uint32_t)(n0 + r0) * d.Cin * ES + q * 16u;

  auto issue_loads = [&](u32x4 (&ra)[AROWS], u32x4 (&rb)[BROWS]) {
    const int dd = d.td.off0 + d.td.offs * jd;
    const int dh = d.th.off0 + d.th.offs * jh;
    const int dw = d.tw.off0 + d.tw.offs * jw;
    const int tapvox = (dd * d.Hi + dh) * d.Wi + dw;
    const bool first = cc < d.c1;
    const __amdgpu_buffer_rsrc_t rs = first ? rs1 : rs2;
    const uint32_t ldb = (uint32_t)(first ? d.ldx1 : d.ldx2) * ES;
    const uint32_t cb = (uint32_t)((first ? cc : cc - d.c1) + q * 8) * ES;
    const uint32_t oob = first ? nrec1 : nrec2;
    const bool kok = (cc + q * 8) < d.Cin;   // Cin % 8 == 0: the last chunk may be partly empty
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int id = sd0[i] + dd, ih = sh0[i] + dh, iw = sw0[i] + dw;
      const bool inb = ((unsigned)id < (unsigned)d.Di) & ((unsigned)ih < (unsigned)d.Hi) &
                       ((unsigned)iw < (unsigned)d.Wi);
      const uint32_t lin = (uint32_t)(rowvox[i] + tapvox) * ldb + cb;
      const uint32_t off = (inb & kok) ? lin : oob;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    }
    const int wt = ((d.td.k0 + d.td.ks * jd) * d.KH + (d.th.k0 + d.th.ks * jh)) * d.KW +
                   (d.tw.k0 + d.tw.ks * jw);
    const uint32_t woff = ((uint32_t)wt * d.Npad * d.Cin + cc) * ES + brow_off;
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      const bool rok = (BN % RPP == 0) || (r0 + RPP * i < BN);
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(
          rsw, (kok && rok) ? woff + (uint32_t)(RPP * i) * d.Cin * ES : nrecw, 0, 0);
    }
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
  auto commit_loads = [&](int buf, const u32x4 (&ra)[AROWS], const u32x4 (&rb)[BROWS]) {
    unsigned char* a = As + buf * BM * ROWB;
    unsigned char* b = Bs + buf * BN * ROWB;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) *reinterpret_cast<u32x4*>(a + (r0 + RPP * i) * ROWB + q * 16) = ra[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i)
      if ((BN % RPP == 0) || (r0 + RPP * i < BN))
        *reinterpret_cast<u32x4*>(b + (r0 + RPP * i) * ROWB + q * 16) = rb[i];
  };

  const int arow = wm * WTM + (lane & 31);
  const int brow = wn * WTN + (lane & 31);
  const int koff = 16 * (lane >> 5);   // bytes: lane half h holds k = 8h .. 8h+7 of a 16-wide MFMA k step
  auto compute = [&](int buf) {
    const unsigned char* a = As + buf * BM * ROWB;
    const unsigned char* b = Bs + buf * BN * ROWB;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 fa[FM], fb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        fa[i] = *reinterpret_cast<const bf16x8*>(a + (arow + 32 * i) * ROWB + kk * 32 + koff);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        fb[j] = *reinterpret_cast<const bf16x8*>(b + (brow + 32 * j) * ROWB + kk * 32 + koff);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      if (kk == 0) __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (nsteps > 0) {
    issue_loads(ra0, rb0);
    issue_loads(ra1, rb1);
    commit_loads(0, ra0, rb0);
  }
  __syncthreads();

  if (LDSBUF == 2) {
    for (int s = 0; s < nsteps; s += 2) {
      issue_loads(ra0, rb0);
      compute(0);
      commit_loads(1, ra1, rb1);
      __syncthreads();
      if (s + 1 >= nsteps) break;
      issue_loads(ra1, rb1);
      compute(1);
      commit_loads(0, ra0, rb0);
      __syncthreads();
    }
  } else {
    for (int s = 0; s < nsteps; s += 2) {
      issue_loads(ra0, rb0);
      compute(0);
      __syncthreads();
      commit_loads(0, ra1, rb1);
      __syncthreads();
      if (s + 1 >= nsteps) break;
      issue_loads(ra1, rb1);
      compute(0);
      __syncthreads();
      commit_loads(0, ra0, rb0);
      __syncthreads();
    }
  }

  // ---- epilogue: bias + activation + store (+ statistics from the fp32 values) ----
  const int chalf = lane >> 5;
  const bool y32 = (d.flags & REHR_GG_Y_F32) != 0;
  __bf16* yb = reinterpret_cast<__bf16*>(d.y);
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
        const float v = apply_act(acc[i][j][r] + bv, d.act, d.slope);
        if (off >= 0 && colok) {
          if (y32) d.y[(int64_t)off * d.ldy + col] = v;
          else yb[(int64_t)off * d.ldy + col] = (__bf16)v;
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
// interleave: see GGMulti in gather_gemm.hip -- the phases of one lattice tile as consecutive blocks of one XCD
struct GBMulti {
  GBParams ph[MAX_PHASES];
  int interleave, count, no_interleave;
};

__device__ __forceinline__ bool interleaved_block_b(int b, int m_tiles, int n_tiles, int count, int& phase, int& logical) {
  const int xcd = b & 7, j = b >> 3, per = count * n_tiles;
  const int mt = (j / per) * 8 + xcd, rem = j % per;
  phase = rem / n_tiles;
  logical = mt * n_tiles + (rem - phase * n_tiles);
  return mt < m_tiles;
}

template <int BM, int BN, int WGM, int WGN, int LDSBUF, int BK>
__global__ __launch_bounds__(NTHREADS, 2) void gather_gemm_bf16_multi_kernel(const GBMulti pm) {
  int phase, logical;
  if (pm.interleave) {
    if (!interleaved_block_b((int)blockIdx.x, pm.ph[0].m_tiles, pm.ph[0].n_tiles, pm.count, phase, logical)) return;
  } else {
    phase = blockIdx.z;
    const int nb = pm.ph[phase].m_tiles * pm.ph[phase].n_tiles;
    if ((int)blockIdx.x >= nb) return;
    logical = xcd_remap(blockIdx.x, nb);
  }
  phase = __builtin_amdgcn_readfirstlane(phase);      // SGPR index: pm.ph[phase] stays a scalar kernarg load
  logical = __builtin_amdgcn_readfirstlane(logical);
  gg_bf16_body<BM, BN, WGM, WGN, LDSBUF, BK>(pm.ph[phase], 0, logical);
}
template <int BM, int BN, int WGM, int WGN, int LDSBUF, int BK>
__global__ __launch_bounds__(NTHREADS, 2) void gather_gemm_bf16_kernel(const GBParams p) {
  gg_bf16_body<BM, BN, WGM, WGN, LDSBUF, BK>(p, (int)gridDim.x);
}

template <int BM, int BN, int WGM, int WGN, int LDSBUF, int BK>
int launch_gb(const GBMulti& pm, int count, hipStream_t stream) {
  const size_t smem = (size_t)LDSBUF * (BM + BN) * (BK * ES + 16) + BM * sizeof(int);
  static bool attr_set = false;
  auto kern1 = gather_gemm_bf16_kernel<BM, BN, WGM, WGN, LDSBUF, BK>;
  auto kernm = gather_gemm_bf16_multi_kernel<BM, BN, WGM, WGN, LDSBUF, BK>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern1), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernm), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess)
      return REHR_EHIP;
    attr_set = true;
  }
  if (count == 1) {
    const GBParams& p = pm.ph[0];
    hipLaunchKernelGGL(kern1, dim3(p.m_tiles * p.n_tiles, p.d.N, 1), dim3(NTHREADS), smem, stream, p);
  } else {
    int nb = 0;
    bool uniform = true;
    for (int i = 0; i < count; ++i) {
      const int n = pm.ph[i].m_tiles * pm.ph[i].n_tiles;
      nb = n > nb ? n : nb;
      uniform = uniform && pm.ph[i].m_tiles == pm.ph[0].m_tiles && pm.ph[i].n_tiles == pm.ph[0].n_tiles;
    }
    GBMulti pmi = pm;
    pmi.count = count;
    const int64_t gx = (int64_t)((pm.ph[0].m_tiles + 7) / 8) * 8 * count * pm.ph[0].n_tiles;
    pmi.interleave = (uniform && !pm.no_interleave && gx < (1ll << 31)) ? 1 : 0;
    if (pmi.interleave) hipLaunchKernelGGL(kernm, dim3((unsigned)gx, pm.ph[0].d.N, 1), dim3(NTHREADS), smem, stream, pmi);
    else hipLaunchKernelGGL(kernm, dim3(nb, pm.ph[0].d.N, count), dim3(NTHREADS), smem, stream, pmi);
  }
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

int validate(const rehr_gather_gemm_desc& d) {
  if (!d.x1 || !d.wp || !d.y) return REHR_EINVAL;
  if (d.N < 1 || d.Cin < 16 || d.Cin % 16 || d.c1 < 1 || d.c1 > d.Cin) return REHR_EINVAL;
  if (d.c1 < d.Cin && (d.c1 % 32 || !d.x2)) return REHR_EINVAL;  // a virtual concat splits on a chunk boundary
  if (d.ldx1 % 8 || (d.x2 && d.ldx2 % 8)) return REHR_EINVAL;    // 16-byte rows
  if (((uintptr_t)d.x1 | (uintptr_t)d.wp | (uintptr_t)(d.x2 ? d.x2 : d.x1)) & 15) return REHR_EINVAL;
  if (d.Npad % 32 || d.Npad < d.Cout || d.Cout < 1) return REHR_EINVAL;
  if (d.Ld < 1 || d.Lh < 1 || d.Lw < 1) return REHR_EINVAL;
  if (d.td.count < 1 || d.th.count < 1 || d.tw.count < 1) return REHR_EINVAL;
  if (d.tile_d != 0 && (d.tile_d < 1 || d.tile_h < 1 || d.tile_w < 1 || d.tile_d * d.tile_h * d.tile_w != 128))
    return REHR_EINVAL;
  if (d.stats_mode != 0 && !d.stats) return REHR_EINVAL;
  if (d.N > 65535) return REHR_EINVAL;
  const int64_t yd = (int64_t)(d.Ld - 1) * d.osd + d.obd, yh = (int64_t)(d.Lh - 1) * d.osh + d.obh,
                yw = (int64_t)(d.Lw - 1) * d.osw + d.obw;
  if (d.obd < 0 || d.obh < 0 || d.obw < 0 || yd >= d.Dy || yh >= d.Hy || yw >= d.Wy) return REHR_EINVAL;
  if (d.ldy < d.Cout) return REHR_EINVAL;
  if ((int64_t)d.N * d.Dy * d.Hy * d.Wy >= (1ll << 31)) return REHR_EINVAL;
  return REHR_OK;
}

bool chunk64(const rehr_gather_gemm_desc& d) { return d.Cin % 64 == 0 && (d.c1 == d.Cin || d.c1 % 64 == 0); }

int plan(const rehr_gather_gemm_desc& d, GBParams& p) {
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
  const int bk = chunk64(d) ? 64 : 32;
  p.kchunks = (d.Cin + bk - 1) / bk;
  const int64_t kd_max = d.td.k0 + (int64_t)d.td.ks * (d.td.count - 1);
  const int64_t kh_max = d.th.k0 + (int64_t)d.th.ks * (d.th.count - 1);
  const int64_t kw_max = d.tw.k0 + (int64_t)d.tw.ks * (d.tw.count - 1);
  const int64_t taps_all = ((kd_max * d.KH) + kh_max) * d.KW + kw_max + 1;
  p.wp_bytes = taps_all * d.Npad * d.Cin * ES;
  const int64_t img = (int64_t)d.Di * d.Hi * d.Wi * ES;
  if (p.wp_bytes >= (1ll << 32) - 64 || img * d.ldx1 >= (1ll << 32) - 64 ||
      (d.x2 && img * d.ldx2 >= (1ll << 32) - 64))
    return REHR_ENOSUP;
  p.n_tiles = d.Npad / (d.Npad % 128 == 0 ? 128 : (d.Npad % 64 == 0 ? 64 : 32));
  return REHR_OK;
}

int launch_generic(const GBMulti& pm, int count, hipStream_t st) {
  const int npad = pm.ph[0].d.Npad;
  const bool k64 = chunk64(pm.ph[0].d);
  if (npad % 128 == 0)
    return k64 ? launch_gb<128, 128, 2, 2, 2, 64>(pm, count, st) : launch_gb<128, 128, 2, 2, 2, 32>(pm, count, st);
  if (npad % 64 == 0)
    return k64 ? launch_gb<128, 64, 2, 2, 2, 64>(pm, count, st) : launch_gb<128, 64, 2, 2, 2, 32>(pm, count, st);
  return k64 ? launch_gb<128, 32, 4, 1, 2, 64>(pm, count, st) : launch_gb<128, 32, 4, 1, 2, 32>(pm, count, st);
}

// in[a][b][t] (or [b][a][t]) fp32 -> out[t][Apad][B] bf16, zero rows beyond A
__global__ void pack_weights_bf16_kernel(const float* __restrict__ in, __bf16* __restrict__ out, int A, int Apad,
                                         int B, int T, int transpose_ab) {
  const int64_t total = (int64_t)T * Apad * B;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % B);
    const int64_t r = i / B;
    const int a = (int)(r % Apad);
    const int t = (int)(r / Apad);
    float v = 0.f;
    if (a < A) v = transpose_ab ? in[((int64_t)b * A + a) * T + t] : in[((int64_t)a * B + b) * T + t];
    out[i] = (__bf16)v;
  }
}

}  // namespace

extern "C" int rehr_gather_gemm_multi_bf16(const rehr_gather_gemm_desc* descs, int32_t count, void* stream) {
  if (descs == nullptr || count < 1 || count > MAX_PHASES) return REHR_EINVAL;
  if (count > 1) {   // the stride phases of a kernel == stride transposed convolution: one fused launch (tconv_ks.hip)
    bool ok = true;
    for (int i = 0; i < count && ok; ++i) ok = validate(descs[i]) == REHR_OK;
    if (ok) {
      const int trc = tconv_ks_try(descs, count, true, (hipStream_t)stream);
      if (trc != REHR_ENOSUP) return trc;
    }
  }
  GBMulti pm;
  pm.interleave = 0;
  pm.count = 0;
  pm.no_interleave = (descs[0].debug_flags & REHR_DBG_GG_INTERLEAVE) ? 0 : 1;
  int n = 0;
  for (int i = 0; i < count; ++i) {
    int rc = validate(descs[i]);
    if (rc != REHR_OK) return rc;
    if (descs[i].Npad != descs[0].Npad || descs[i].N != descs[0].N || descs[i].wp != descs[0].wp ||
        descs[i].x1 != descs[0].x1 || descs[i].Cin != descs[0].Cin || descs[i].c1 != descs[0].c1)
      return REHR_EINVAL;
    // (tap-range parts of a split-K launch -- they differ in y -- stay together in ONE generic grid: a halo launch per
    // part would run them one after the other on a few CUs each)
    const bool split_part = count > 1 && descs[i].y != descs[(i + 1) % count].y;
    if (!(descs[i].debug_flags & REHR_DBG_GG_NO_HALO) && !split_part) {   // unit-stride 3x3(x3) taps: input brick + halo staged in LDS
      rc = halo_conv_bf16_try(descs[i], (hipStream_t)stream);
      if (rc == REHR_OK) continue;
      if (rc != REHR_ENOSUP) return rc;
    }
    rc = plan(descs[i], pm.ph[n]);
    if (rc != REHR_OK) return rc;
    ++n;
  }
  if (n == 0) return REHR_OK;
  return launch_generic(pm, n, (hipStream_t)stream);
}

extern "C" int rehr_gather_gemm_bf16(const rehr_gather_gemm_desc* dp, void* stream) {
  if (dp == nullptr) return REHR_EINVAL;
  return rehr_gather_gemm_multi_bf16(dp, 1, stream);
}

extern "C" int rehr_pack_weights_bf16(const float* in, void* out, int32_t A, int32_t Apad, int32_t B, int32_t T,
                                      int32_t transpose_ab, void* stream) {
  if (!in || !out || A < 1 || Apad < A || B < 1 || T < 1) return REHR_EINVAL;
  const int64_t total = (int64_t)T * Apad * B;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in,
                     reinterpret_cast<__bf16*>(out), A, Apad, B, T, transpose_ab);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
