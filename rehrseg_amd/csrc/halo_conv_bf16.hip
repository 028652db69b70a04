// Halo-brick convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the unit-stride
// 3x3x3 / 1x3x3 member of the mixed-precision gather-GEMM family (same descriptor, same results as
// gather_gemm_bf16.hip) -- forward AND input gradient (flipped taps) of every nnU-Net stage conv and FLAVR block,
// and the stride phases of the (3,4,4)/(1,2,2) transposed convolutions.
//
// At 16x the fp32 rate the matrix pipe outruns everything that re-gathers operands per tap: gather_gemm_bf16
// reads each input voxel 27 times through L1/L2 and reaches ~0.2 PFLOP/s on the 32/64-channel layers at
// 128^3-160^3, which hold 60 % of the SegModel's FLOPs.  Here ("LDS-staged 3-D input tiles with halo", BASELINE
// north_star) BOTH operands of a 32-channel chunk live in LDS:
//   * a block (4 waves, one per SIMD, the whole register file and 144 KB of LDS: one block per CU) owns a brick of
//     512 output voxels x 32 output channels; the input brick WITH ITS HALO (<= 6x10x18 voxels x 64 bytes, rows
//     padded to 80 bytes) is staged once per chunk and every tap reads its fragments at (voxel row + tap offset);
//   * the weights of ALL taps of the chunk (27 x 32 x 32 bf16 = 54 KB, 16-byte pieces XOR-swizzled by the row) are
//     staged once per chunk -- once per BLOCK for 32-channel layers, whose persistent blocks keep them for every
//     brick.  (The first version fetched weight fragments from L1/L2 one tap ahead: with one wave per SIMD every
//     tap paid the L2 latency, 60k cycles per brick against 6.9k cycles of MFMA.)
//   * a wave owns 128 voxels x 32 channels: per 16-channel k step 4 activation reads + 1 weight read
//     (ds_read_b128, conflict-free) feed 4 MFMAs; reads run one MFMA group ahead in two register sets;
//   * the MFMA takes the WEIGHTS as its A operand: the accumulator then holds a voxel per lane and 4 consecutive
//     channels per register quad, so the epilogue stores 8 bytes per lane (16 for fp32 output) instead of single
//     bf16 values; the per-(sample, channel) statistics stay in registers across the bricks of a sample and are
//     reduced across lanes once per sample.
// Epilogue semantics as in the gather kernel: bias, ReLU / LeakyReLU, bf16 (or fp32) store, fp64 statistics formed
// from the fp32 values.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;              // channels per chunk
constexpr int XROW = BK * 2 + 16;   // halo row stride in bytes (padded)
constexpr int WROW = BK * 2;        // weight row (one output channel of one tap) in bytes, pieces swizzled
constexpr int TPR = 4;              // threads per halo row (16 bytes each)
// NWV = waves per block: 4 (one per SIMD, 128 voxels = 4 tiles each) or 8 (two per SIMD, 64 voxels = 2 tiles each:
// a second resident wave fills the issue gaps of the first, at the price of one weight-fragment read per 2 MFMAs)
template <int NWV> struct HBW {
  static constexpr int NTHR = 64 * NWV, RPP = NTHR / TPR, FM = 16 / NWV;
};
constexpr int BVOX = 512;
constexpr int BN = 32;

struct HBParams {
  rehr_gather_gemm_desc d;
  int HD, HH, HW, hvox;
  int mind, minh, minw;
  int nb_d, nb_h, nb_w, tiles_per_img;
  int64_t ntiles;
  int tiles_per_block;
  int kchunks;
  uint32_t wp_bytes;
};

// NT3 = taps / 3 (9: 3x3x3, 3: 1x3x3): the sweep is fully unrolled so that the next halo's loads can be spread over it
template <int BD, int BH, int BW, int NT3, int NWV>
__global__ __launch_bounds__(64 * NWV, 1) void halo_conv_bf16_kernel(const HBParams p) {
  static_assert(BD * BH * BW == BVOX, "512 voxels");
  constexpr int NTHR = HBW<NWV>::NTHR, RPP = HBW<NWV>::RPP, FM = HBW<NWV>::FM;
  constexpr int MAXX = ((BD + 2) * (BH + 2) * (BW + 2) + RPP - 1) / RPP;
  const rehr_gather_gemm_desc& d = p.d;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* Xs = smem_b;                                   // [MAXX * RPP >= hvox][XROW]
  unsigned char* Ws = smem_b + MAXX * RPP * XROW;               // [ntaps][32][WROW]
  constexpr int ntaps = 3 * NT3;
  int* row_out = (int*)(Ws + ntaps * BN * WROW);                // [BVOX]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int n0 = blockIdx.y * BN;
  const int q = tid % TPR, r0 = tid / TPR;

  // halo rows owned by this thread (brick independent): packed (hd, hh, hw) and the voxel index relative to the
  // halo origin; rows past the halo get coordinates that fail every bounds test
  int hcoord[MAXX], hrel[MAXX];
#pragma unroll
  for (int i = 0; i < MAXX; ++i) {
    const int hv = r0 + RPP * i;
    const int hw_ = hv % p.HW;
    const int t2 = hv / p.HW;
    const int hd_ = t2 / p.HH, hh_ = t2 % p.HH;
    hcoord[i] = hv < p.hvox ? ((hd_ << 20) | (hh_ << 10) | hw_) : (1023 << 20);
    hrel[i] = (hd_ * d.Hi + hh_) * d.Wi + hw_;
  }
  const uint32_t img_elems = (uint32_t)d.Di * d.Hi * d.Wi;

  // Voxel of a 32-voxel tile held by this lane.  ds_read_b128 is served in 16-lane groups {0-3,12-15,20-27} and
  // {4-11,16-19,28-31} (per 32-lane half); giving each hardware group 16 CONSECUTIVE voxels of one halo row makes its
  // 16 addresses 80 bytes apart = 16 distinct 16-byte bank slots (5 is coprime to 16): conflict-free.  With the
  // natural order lane = voxel the two halo rows of a group collide on 2 of 16 slots and every read took 2x
  // (SQ_LDS_BANK_CONFLICT = 94 % of the LDS cycles).
  const int l5 = lane & 31;
  const bool g0 = (l5 < 4) | ((l5 >= 12) & (l5 < 16)) | ((l5 >= 20) & (l5 < 28));
  const int vtile = g0 ? (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12)) : 16 + (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16));
  // activation-fragment byte offsets of this lane (brick independent): voxel r -> its halo row at the tap origin
  int arow[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int r = wave * (FM * 32) + i * 32 + vtile;
    const int rd = r / (BH * BW), rh = (r / BW) % BH, rw = r % BW;
    arow[i] = ((rd * p.HH + rh) * p.HW + rw) * XROW + 16 * half;
  }
  // weight-fragment byte offsets: row = output channel (lane & 31), 16-byte piece c = 2*kk + half, swizzled by the row
  const int wco = lane & 31;
  int wrow[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) wrow[kk] = wco * WROW + (((2 * kk + half) ^ ((wco >> 2) & 3)) << 4);

  f32x16 acc[FM];
  u32x4 rx[MAXX];
  float s1[16], s2[16];   // per-lane partial statistics of channels n0 + 8g + 4*half + e (index 4g + e)
#pragma unroll
  for (int k = 0; k < 16; ++k) s1[k] = s2[k] = 0.f;
  int stats_n = -1;

  const int64_t t_begin = (int64_t)blockIdx.x * p.tiles_per_block;
  int64_t t_end = t_begin + p.tiles_per_block;
  if (t_end > p.ntiles) t_end = p.ntiles;
  const int64_t items = (t_end > t_begin ? t_end - t_begin : 0) * p.kchunks;

  // Next halo: fetch_prep() forms the 17 load offsets of the next (brick, chunk) -- pure ALU work -- and the loads
  // themselves are issued a few at a time INSIDE the sweep (fetch_issue).  Issued in one burst at the top of a
  // brick (first version) every CU of the chip asks for its 69 KB at the same moment, the requests queue at ~10 B/clk
  // per CU and the issuing wave -- the only one on its SIMD -- sits in the queue for 6.6k cycles instead of
  // multiplying (s_memtime stamps).
  int f_tile = 0, f_chunk = 0;   // (brick, chunk) the next fetch_prep() addresses; advanced by it
  uint32_t foff[MAXX];
  __amdgpu_buffer_rsrc_t frs;
  auto fetch_prep = [&](int64_t it) {
    const bool live = it < items;
    const int tile32 = (int)t_begin + (live ? f_tile : 0);     // ntiles < 2^31 (checked on the host)
    const int cc = (live ? f_chunk : 0) * BK;
    ++f_chunk;
    if (f_chunk == p.kchunks) { f_chunk = 0; ++f_tile; }
    const int n = tile32 / p.tiles_per_img;
    int tr = tile32 - n * p.tiles_per_img;
    const int bw_ = tr % p.nb_w; tr /= p.nb_w;
    const int bh_ = tr % p.nb_h;
    const int bd_ = tr / p.nb_h;
    const int gd0 = bd_ * BD + p.mind, gh0 = bh_ * BH + p.minh, gw0 = bw_ * BW + p.minw;
    const bool first = cc < d.c1;
    const __bf16* src = reinterpret_cast<const __bf16*>(first ? d.x1 : d.x2);
    const uint32_t ld = (uint32_t)(first ? d.ldx1 : d.ldx2);
    const int coff = (first ? cc : cc - d.c1) + q * 8;
    const uint32_t nrec = img_elems * ld * 2u;
    frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(src) + (int64_t)n * img_elems * ld, 0, nrec, 0x00020000);
    const bool kok = live & ((cc + q * 8) < d.Cin);
    const int vbase = (gd0 * d.Hi + gh0) * d.Wi + gw0;            // voxel index of the halo origin (may be negative)
    const uint32_t ldb = ld * 2u, cob = (uint32_t)coff * 2u;
#pragma unroll
    for (int i = 0; i < MAXX; ++i) {
      const int hc = hcoord[i];
      const bool ok = kok & ((unsigned)(gd0 + (hc >> 20)) < (unsigned)d.Di) &
                      ((unsigned)(gh0 + ((hc >> 10) & 1023)) < (unsigned)d.Hi) &
                      ((unsigned)(gw0 + (hc & 1023)) < (unsigned)d.Wi);
      foff[i] = ok ? (uint32_t)(vbase + hrel[i]) * ldb + cob : nrec;   // out of range = zero padding
    }
  };
#define HB_FETCH_ISSUE(i) rx[i] = __builtin_amdgcn_raw_buffer_load_b128(frs, foff[i], 0, 0)
  auto stage = [&]() {   // (rows past hvox exist in LDS and receive zeros: no conditions here)
#pragma unroll
    for (int i = 0; i < MAXX; ++i) *reinterpret_cast<u32x4*>(Xs + (r0 + RPP * i) * XROW + q * 16) = rx[i];
  };

  // all taps' weights of chunk cc -> LDS: piece (t, co, c) = wp[wt(t)][n0 + co][cc + 8c .. +7]
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(reinterpret_cast<const __bf16*>(d.wp)), 0, p.wp_bytes, 0x00020000);
  auto stage_weights = [&](int cc) {
    constexpr int NPC = ntaps * BN * 4;            // 16-byte pieces
    constexpr int PPT = (NPC + NTHR - 1) / NTHR;   // per thread (4 waves): 14 (27 taps) / 5 (9 taps)
    const int thw = d.th.count * d.tw.count;
    u32x4 v[PPT];
    int dst[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {                // every load in flight before the first LDS write
      const int pc = u * NTHR + tid;
      const int t = pc >> 7, co = (pc >> 2) & 31, c = pc & 3;
      const int jd = t / thw, jr = t - jd * thw, jh = jr / d.tw.count, jw = jr - jh * d.tw.count;
      const int wt = ((d.td.k0 + d.td.ks * jd) * d.KH + (d.th.k0 + d.th.ks * jh)) * d.KW + (d.tw.k0 + d.tw.ks * jw);
      const bool ok = (pc < NPC) & ((cc + 8 * c) < d.Cin);
      const uint32_t off = (((uint32_t)wt * d.Npad + n0 + co) * d.Cin + cc + 8 * c) * 2u;
      v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsw, ok ? off : p.wp_bytes, 0, 0);
      dst[u] = pc < NPC ? (t * BN + co) * WROW + ((c ^ ((co >> 2) & 3)) << 4) : -1;
    }
#pragma unroll
    for (int u = 0; u < PPT; ++u)
      if (dst[u] >= 0) *reinterpret_cast<u32x4*>(Ws + dst[u]) = v[u];
  };

  // halo byte offset of tap t, held by lane t of a VGPR (one v_readlane per tap in the sweep: with a single wave
  // per SIMD every instruction of the loop costs an issue slot the MFMAs cannot hide)
  int my_tapoff;
  {
    const int t = lane < ntaps ? lane : 0;
    const int thw = d.th.count * d.tw.count;
    const int jd = t / thw, jr = t - jd * thw, jh = jr / d.tw.count, jw = jr - jh * d.tw.count;
    const int od_ = d.bd + d.td.off0 + d.td.offs * jd - p.mind;
    const int oh_ = d.bh + d.th.off0 + d.th.offs * jh - p.minh;
    const int ow_ = d.bw + d.tw.off0 + d.tw.offs * jw - p.minw;
    my_tapoff = ((od_ * p.HH + oh_) * p.HW + ow_) * XROW;
  }

  const bool y32 = (d.flags & REHR_GG_Y_F32) != 0;
  __bf16* yb = reinterpret_cast<__bf16*>(d.y);
  // epilogue constants, once per block: bias of this lane's 16 channels; act(v) = max(v, v * se) with se = 1 (none),
  // 0 (ReLU) or the LeakyReLU slope (0 <= slope <= 1)
  float bv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int col = n0 + 8 * (k >> 2) + 4 * half + (k & 3);
    bv[k] = (d.bias != nullptr && col < d.Cout) ? d.bias[col] : 0.f;
  }
  const float se = d.act == REHR_ACT_RELU ? 0.f : (d.act == REHR_ACT_LRELU ? d.slope : 1.f);
  const bool want_stats = d.stats_mode != 0;
  const uint32_t y_bytes = (uint32_t)((int64_t)d.N * d.Dy * d.Hy * d.Wy * d.ldy * 2);   // < 2^32: checked on the host
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(yb, 0, y_bytes, 0x00020000);
  // Statistics of sample `sn` out of the registers: summed over the block in LDS (the halo buffer is idle at both call
  // sites), then ONE atomic per (channel, statistic) and block, the 64 of them contiguous in one wave instruction.
  // (Per-wave atomics from single lanes -- 256 per block onto the same few cache lines -- serialised at the memory
  // side: 5 ms for a 1x32x160^3 layer.)  Block-uniform: every thread calls it.
  auto flush_stats = [&](int sn) {
    float* red = reinterpret_cast<float*>(Xs);   // [32 values][NTHR threads]
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      red[k * NTHR + tid] = s1[k];
      red[(16 + k) * NTHR + tid] = s2[k];
      s1[k] = s2[k] = 0.f;
    }
    __syncthreads();
    if (wave == 0) {
      const int c = lane >> 1, st = lane & 1;                     // channel n0 + c, statistic st
      const int k = 4 * (c >> 3) + (c & 3) + 16 * st, h = (c >> 2) & 1;
      float sum = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < NWV; ++w4)
        for (int l = 0; l < 32; ++l) sum += red[k * NTHR + w4 * 64 + h * 32 + l];
      if (n0 + c < d.Cout && (st == 0 || d.stats_mode == 2))
        atomicAdd(d.stats + ((int64_t)sn * d.Cout + n0 + c) * 2 + st, (double)sum);
    }
    __syncthreads();
  };

  if (items > 0) {
    stage_weights(0);
    fetch_prep(0);
#pragma unroll
    for (int i = 0; i < MAXX; ++i) HB_FETCH_ISSUE(i);
    stage();
  }
  __syncthreads();

#ifdef HB_STAMPS
  long long st_sweep = 0, st_epi = 0, st_stage = 0, st_fetch = 0;
#define HB_T() __builtin_amdgcn_s_memtime()
#endif
  int tile_i = 0, chunk = 0;   // (brick, chunk) of the current item, advanced without divisions
  for (int64_t it = 0; it < items; ++it) {
#ifdef HB_STAMPS
    const long long t0 = HB_T();
#endif
    const int64_t tile = t_begin + tile_i;
    const int cc = chunk * BK;
    if (chunk == 0) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }
    fetch_prep(it + 1);   // offsets of the next halo; its loads are issued inside the sweep
    const bool k1 = (d.Cin - cc) >= 32;
#ifdef HB_STAMPS
    const long long t1 = HB_T();
#endif

    if (k1) {
      // 54 k steps (27 taps x 2) of 4 MFMAs.  Three register sets: the 5 reads of step s + 2 are issued behind the
      // first MFMA of step s, so an LDS read has two MFMA groups (256 cycles) to land; the loop is unrolled over 3
      // taps = 6 steps so that the set indices are compile-time, and its instruction stream is pinned
      // (sched_group_barrier) and kept to ~15 instructions per step.
      bf16x8 fx[3][FM], fw[3];
      int xb[2][FM];           // halo row + tap offset of the two taps in flight
      auto tap_base = [&](int slot, int t) {
        const int xo = __builtin_amdgcn_readlane(my_tapoff, t);
#pragma unroll
        for (int i = 0; i < FM; ++i) xb[slot][i] = arow[i] + xo;
      };
      auto rd = [&](int set, int slot, int t, int kk) {
        fw[set] = *reinterpret_cast<const bf16x8*>(Ws + t * (BN * WROW) + wrow[kk]);
#pragma unroll
        for (int i = 0; i < FM; ++i) fx[set][i] = *reinterpret_cast<const bf16x8*>(Xs + xb[slot][i] + kk * 32);
      };
      auto mm = [&](int set) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[set], fx[set][i], acc[i], 0, 0, 0);
      };
#define HB_PIN()                                                                    \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      /* 1 MFMA */                \
  __builtin_amdgcn_sched_group_barrier(0x100, FM + 1, 0); /* the step's DS reads */   \
  __builtin_amdgcn_sched_group_barrier(0x008, FM - 1, 0); /* the other MFMAs */       \
  __builtin_amdgcn_sched_barrier(0)
      tap_base(0, 0);
      rd(0, 0, 0, 0);          // step 0 = (tap 0, kk 0)
      rd(1, 0, 0, 1);          // step 1 = (tap 0, kk 1)
      __builtin_amdgcn_sched_barrier(0);
      constexpr int LPI = (MAXX + NT3 - 1) / NT3;   // halo loads issued per 3-tap iteration: 2 (27 taps) / 6 (9 taps)
#pragma unroll
      for (int j = 0; j < NT3; ++j) {
        const int tb = 3 * j;
        // steps 6j .. 6j+5 = taps tb, tb+1, tb+2; reads run two steps (= one tap) ahead; past the last tap they wrap
        // to tap 0 (in range, never consumed)
        const int t1_ = tb + 1, t2_ = tb + 2, t3_ = (tb + 3 < ntaps) ? tb + 3 : 0;
#pragma unroll
        for (int u = 0; u < LPI; ++u)
          if (j * LPI + u < MAXX) HB_FETCH_ISSUE(j * LPI + u);
        tap_base(1, t1_);
        rd(2, 1, t1_, 0); mm(0); HB_PIN();     // step (tb, 0)   | reads (tb+1, 0)
        rd(0, 1, t1_, 1); mm(1); HB_PIN();     // step (tb, 1)   | reads (tb+1, 1)
        tap_base(0, t2_);
        rd(1, 0, t2_, 0); mm(2); HB_PIN();     // step (tb+1, 0) | reads (tb+2, 0)
        rd(2, 0, t2_, 1); mm(0); HB_PIN();     // step (tb+1, 1) | reads (tb+2, 1)
        tap_base(1, t3_);
        rd(0, 1, t3_, 0); mm(1); HB_PIN();     // step (tb+2, 0) | reads (tb+3, 0)
        rd(1, 1, t3_, 1); mm(2); HB_PIN();     // step (tb+2, 1) | reads (tb+3, 1)
      }
#undef HB_PIN
    } else {   // half-filled last chunk (Cin % 32 == 16): one k step per tap, no pipelining (rare)
#pragma unroll
      for (int i = 0; i < MAXX; ++i) HB_FETCH_ISSUE(i);
      for (int t = 0; t < ntaps; ++t) {
        const int xo = __builtin_amdgcn_readlane(my_tapoff, t);
        const bf16x8 fwv = *reinterpret_cast<const bf16x8*>(Ws + t * (BN * WROW) + wrow[0]);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const bf16x8 fxv = *reinterpret_cast<const bf16x8*>(Xs + arow[i] + xo);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwv, fxv, acc[i], 0, 0, 0);
        }
      }
    }

#ifdef HB_STAMPS
    const long long t2 = HB_T();
#endif
    if (chunk == p.kchunks - 1) {
      // ---- epilogue of this brick
      const int n_img = (int)(tile / p.tiles_per_img);
      int tr = (int)(tile - (int64_t)n_img * p.tiles_per_img);
      const int bw_ = tr % p.nb_w; tr /= p.nb_w;
      const int bh_ = tr % p.nb_h;
      const int bd_ = tr / p.nb_h;
      for (int v = tid; v < BVOX; v += NTHR) {
        const int od = bd_ * BD + v / (BH * BW), oh = bh_ * BH + (v / BW) % BH, ow = bw_ * BW + v % BW;
        int off = -1;
        if (od < d.Ld && oh < d.Lh && ow < d.Lw)
          off = ((n_img * d.Dy + od * d.osd + d.obd) * d.Hy + oh * d.osh + d.obh) * d.Wy + ow * d.osw + d.obw;
        row_out[v] = off;
      }
      __syncthreads();   // row_out complete; every wave has finished its sweep: the halo buffer is free
      if (want_stats && n_img != stats_n) {   // (block-uniform)
        if (stats_n >= 0) flush_stats(stats_n);
        stats_n = n_img;
      }
      // accumulator register r of tile i: channel n0 + 8*(r>>2) + 4*half + (r&3), voxel wave*128 + 32*i + vtile
      if (!y32) {
        // bf16 output through a wave-private LDS image [128 voxels][32 channels] (80-byte rows): the accumulator
        // layout (a voxel per lane, 4 channels per register quad) would store 8-byte pieces of 32 different rows
        // per instruction; from the image every lane stores 16 bytes and 4 lanes complete a voxel's 64-byte segment.
        unsigned char* img = Xs + wave * (FM * 32 * XROW);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          float lv = 0.f;   // 1 for a voxel inside the lattice: the statistics skip padding voxels
          if (want_stats) lv = row_out[wave * (FM * 32) + i * 32 + vtile] >= 0 ? 1.f : 0.f;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t = acc[i][4 * g + e] + bv[4 * g + e];
              v[e] = fmaxf(t, t * se);
            }
            const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *reinterpret_cast<bf16x4*>(img + (i * 32 + vtile) * XROW + (8 * g + 4 * half) * 2) = o;
            if (want_stats) {   // (channels past Cout accumulate too; flush_stats drops them)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float m = v[e] * lv;
                s1[4 * g + e] += m;
                s2[4 * g + e] = fmaf(m, v[e], s2[4 * g + e]);
              }
            }
          }
        }
        // (wave-private image: the wave's own LDS writes are visible to its reads once they have completed)
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        const int piece = lane & 3, vv = lane >> 2;
        const bool pok = (n0 + 8 * piece + 7) < d.Cout;
        const uint32_t cby = (uint32_t)(n0 + 8 * piece) * 2u, ldyb = (uint32_t)d.ldy * 2u;
#pragma unroll
        for (int j = 0; j < 2 * FM; ++j) {   // stores of 16 voxels x 64 bytes
          const int row = j * 16 + vv;
          const int off = row_out[wave * (FM * 32) + row];
          const u32x4 val = *reinterpret_cast<const u32x4*>(img + row * XROW + piece * 16);
          __builtin_amdgcn_raw_buffer_store_b128(val, rsy, (off >= 0 && pok) ? (uint32_t)off * ldyb + cby : y_bytes, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int off = row_out[wave * (FM * 32) + i * 32 + vtile];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int col = n0 + 8 * g + 4 * half;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[i][4 * g + e] + bv[4 * g + e], d.act, d.slope);
            if (off >= 0 && col + 3 < d.Cout) {        // Cout % 4 == 0: a quad is in or out as a whole
              const f32x4 o = {v[0], v[1], v[2], v[3]};
              *reinterpret_cast<f32x4*>(d.y + (int64_t)off * d.ldy + col) = o;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                s1[4 * g + e] += v[e];
                s2[4 * g + e] = fmaf(v[e], v[e], s2[4 * g + e]);
              }
            }
          }
        }
      }
    }
#ifdef HB_STAMPS
    const long long t3 = HB_T();
#endif
    __syncthreads();   // every wave is done with this halo, these weights, row_out and its store image
    stage();           // next halo: registers -> LDS
    ++chunk;
    if (chunk == p.kchunks) { chunk = 0; ++tile_i; }
    if (p.kchunks > 1 && it + 1 < items) stage_weights(chunk * BK);
    __syncthreads();
#ifdef HB_STAMPS
    const long long t4 = HB_T();
    st_fetch += t1 - t0; st_sweep += t2 - t1; st_epi += t3 - t2; st_stage += t4 - t3;
#endif
  }
#ifdef HB_STAMPS
  if (d.wino_ws != nullptr && blockIdx.x == 3 && blockIdx.y == 0 && tid == 64) {
    long long* o = reinterpret_cast<long long*>(d.wino_ws);
    o[0] = st_fetch; o[1] = st_sweep; o[2] = st_epi; o[3] = st_stage; o[4] = items;
  }
#endif
  if (want_stats && stats_n >= 0) flush_stats(stats_n);
}

void span(const rehr_axis_taps& t, int b, int* mn, int* mx) {
  int lo = b + t.off0, hi = lo;
  for (int j = 1; j < t.count; ++j) {
    const int o = b + t.off0 + t.offs * j;
    if (o < lo) lo = o;
    if (o > hi) hi = o;
  }
  *mn = lo;
  *mx = hi;
}

template <int BD, int BH, int BW, int NT3, int NWV>
int launch(HBParams p, hipStream_t stream) {
  constexpr int RPP = HBW<NWV>::RPP;
  const rehr_gather_gemm_desc& d = p.d;
  if (d.Ld < (BD + 1) / 2 || d.Lh < BH || d.Lw < BW) return REHR_ENOSUP;
  const int64_t nb_d = (d.Ld + BD - 1) / BD, nb_h = (d.Lh + BH - 1) / BH, nb_w = (d.Lw + BW - 1) / BW;
  if (nb_d * BD * nb_h * BH * nb_w * BW * 10 > (int64_t)d.Ld * d.Lh * d.Lw * 13) return REHR_ENOSUP;  // <= 1.3x padding
  p.HD = BD + p.HD; p.HH = BH + p.HH; p.HW = BW + p.HW;      // (the tap spans were left in HD/HH/HW)
  p.hvox = p.HD * p.HH * p.HW;
  if (p.HD > BD + 2 || p.HH > BH + 2 || p.HW > BW + 2) return REHR_ENOSUP;
  constexpr int ntaps = 3 * NT3;
  constexpr int XROWS = (((BD + 2) * (BH + 2) * (BW + 2) + RPP - 1) / RPP) * RPP;   // = MAXX * RPP in the kernel
  const size_t smem = (size_t)XROWS * XROW + (size_t)ntaps * BN * WROW + (size_t)BVOX * sizeof(int);
  if (smem > 158 * 1024) return REHR_ENOSUP;
  if ((int64_t)d.N * d.Dy * d.Hy * d.Wy * d.ldy * 2 >= (1ll << 32) - 64) return REHR_ENOSUP;   // buffer-addressed stores
  p.nb_d = (int)nb_d; p.nb_h = (int)nb_h; p.nb_w = (int)nb_w;
  p.tiles_per_img = (int)(nb_d * nb_h * nb_w);
  p.ntiles = (int64_t)d.N * p.tiles_per_img;
  if (p.ntiles >= (1ll << 31) - 4096) return REHR_ENOSUP;
  // persistent blocks, one per CU: whole rounds of 256
  const int n_tiles = d.Npad / BN;
  int64_t want = 256 / n_tiles;
  if (want < 1) want = 1;
  if (want > p.ntiles) want = p.ntiles;
  p.tiles_per_block = (int)((p.ntiles + want - 1) / want);
  auto kern = halo_conv_bf16_kernel<BD, BH, BW, NT3, NWV>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            158 * 1024) != hipSuccess)
      return REHR_EHIP;
    attr_set = true;
  }
  const int64_t blocks_x = (p.ntiles + p.tiles_per_block - 1) / p.tiles_per_block;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks_x, n_tiles, 1), dim3(64 * NWV), smem, stream, p);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

}  // namespace

// REHR_OK = launched; REHR_ENOSUP = not this kernel's case (the caller uses gather_gemm_bf16); other = error.
int halo_conv_bf16_try(const rehr_gather_gemm_desc& d, hipStream_t stream) {
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return REHR_ENOSUP;
  const int T = d.td.count * d.th.count * d.tw.count;
  if (T != 27 && T != 9) return REHR_ENOSUP;   // 3x3x3 and 1x3x3 / 3x3x1 taps (the sweep is unrolled per tap count)
  if (d.td.count > 3 || d.th.count > 3 || d.tw.count > 3) return REHR_ENOSUP;
  if (d.c1 < d.Cin && d.c1 % BK) return REHR_ENOSUP;
  if (d.Cout % 8 || d.ldy % 8 || ((uintptr_t)d.y & 15)) return REHR_ENOSUP;   // 16-byte pieces of a voxel's row
  int mn[3], mx[3];
  span(d.td, d.bd, &mn[0], &mx[0]);
  span(d.th, d.bh, &mn[1], &mx[1]);
  span(d.tw, d.bw, &mn[2], &mx[2]);
  HBParams p;
  p.d = d;
  p.HD = mx[0] - mn[0]; p.HH = mx[1] - mn[1]; p.HW = mx[2] - mn[2];   // halo extents, completed per brick in launch()
  p.mind = mn[0]; p.minh = mn[1]; p.minw = mn[2];
  const int64_t img = (int64_t)d.Di * d.Hi * d.Wi * 2;
  if (img * d.ldx1 >= (1ll << 32) - 64 || (d.x2 && img * d.ldx2 >= (1ll << 32) - 64)) return REHR_ENOSUP;
  if ((int64_t)d.N * d.Dy * d.Hy * d.Wy >= (1ll << 31)) return REHR_ENOSUP;
  p.kchunks = (d.Cin + BK - 1) / BK;
  {
    const int64_t kd_max = d.td.k0 + (int64_t)d.td.ks * (d.td.count - 1);
    const int64_t kh_max = d.th.k0 + (int64_t)d.th.ks * (d.th.count - 1);
    const int64_t kw_max = d.tw.k0 + (int64_t)d.tw.ks * (d.tw.count - 1);
    const int64_t wb = (((kd_max * d.KH) + kh_max) * d.KW + kw_max + 1) * d.Npad * d.Cin * 2;
    if (wb >= (1ll << 32) - 64) return REHR_ENOSUP;
    p.wp_bytes = (uint32_t)wb;
  }
  // brick shapes in order of preference; one that pads the lattice by more than 1.3x (or does not fit it) declines
  // (eight waves per block, two per SIMD; the four-wave organisation of round 2's first half was 1.0 ms per step slower
  // on the bf16 cfg-3 step: profiles/r03_ab_superseded.txt)
  int rc;
  if (T == 27) {
    rc = launch<4, 8, 16, 9, 8>(p, stream);
    if (rc == REHR_ENOSUP) rc = launch<8, 8, 8, 9, 8>(p, stream);
  } else {
    rc = launch<4, 8, 16, 3, 8>(p, stream);
    if (rc == REHR_ENOSUP) rc = launch<8, 8, 8, 3, 8>(p, stream);
    if (rc == REHR_ENOSUP) rc = launch<2, 16, 16, 3, 8>(p, stream);
  }
  return rc;
}
