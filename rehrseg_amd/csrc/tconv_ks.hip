// Transposed convolution with kernel == stride (nnU-Net's UNetDecoder.transpconvs, called at the reference's
// models/seg_model.py:35; built from train_all.py:474-493's strides) -- forward, fp32 and bf16 operands.
//
//   y[n, o*s + p, co] = bias[co] + sum_ci x[n, o, ci] * W[ci][co][p]            p = stride phase = kernel tap
//
// Every output voxel has exactly ONE tap: the layer is a GEMM [voxels x C_in] x [C_in x (phases * C_out)] whose result
// rows are scattered over the s_d*s_h*s_w output positions of the input voxel.  As `phases` launches of the generic
// gather-GEMM (one grid, blockIdx.z = phase) each phase streamed the whole input again and a block lived for two
// K-steps: the 64 -> 32 layer of cfg-3 (2 x 64^3 -> 128^3) fetched 1.1 GB for 0.13 GB of input and ran at 2.5 TB/s of
// memory-side traffic in 0.64 ms (profiles/r02_pmc_hbm_seg.json), bound by block turnover rather than by HBM.
// Here a block stages its 128 input voxels (all C_in channels: <= 128 x 132 floats) in LDS ONCE and walks the phases:
// per phase the weight panel streams through a double-buffered LDS tile (register prefetch one K-chunk ahead) and
// the 128 x C_out result is stored as full 128-byte lines (lanes = consecutive output channels).
// HBM traffic = x once + y once (+ the weights from L2); a block does `phases` times the work per launch overhead.
#include "common.h"

namespace {

constexpr int TK_BM = 128, TK_BK = 32, TK_LD = 36, TK_THREADS = 256, TK_MAXPH = 8;

struct TconvKsParams {
  const void* x;
  const void* wp;       // [tap][Npad][Cin]
  const float* bias;
  void* y;
  int ldx, ldy, Cin, Cout, Npad;
  int N, Di, Hi, Wi, Do, Ho, Wo;
  int sd, sh, sw;
  int nph;
  int tap[TK_MAXPH];        // weight tap index of the phase
  int yoff[TK_MAXPH];       // output voxel offset of the phase: (pd * Ho + ph) * Wo + pw
  int64_t vox_total;        // N * Di * Hi * Wi
  int lda;                  // LDS row pitch of the staged input tile (elements)
};

// ---------------------------------------------------------------------------------------------------------- fp32
// NT = C_out tiles of 32 per block (all of them: the input tile is staged once for every output channel)
template <int NT>
__global__ __launch_bounds__(TK_THREADS, 2) void tconv_ks_f32_kernel(const TconvKsParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                         // [128][lda]
  float* Bs = smem + TK_BM * p.lda;                         // [2][NT*32][TK_LD]
  int* row_out = (int*)(Bs + 2 * NT * 32 * TK_LD);          // [128] output voxel of phase 0, or -1
  const float* x = (const float*)p.x;
  const float* wp = (const float*)p.wp;
  float* y = (float*)p.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t v0 = (int64_t)blockIdx.x * TK_BM;

  if (tid < TK_BM) {
    const int64_t v = v0 + tid;
    int off = -1;
    if (v < p.vox_total) {
      const int w = (int)(v % p.Wi);
      int64_t t = v / p.Wi;
      const int h = (int)(t % p.Hi); t /= p.Hi;
      const int dd = (int)(t % p.Di);
      const int n = (int)(t / p.Di);
      off = ((n * p.Do + dd * p.sd) * p.Ho + h * p.sh) * p.Wo + w * p.sw;
    }
    row_out[tid] = off;
  }
  // stage the input tile: rows r0 + 32 i, 16-byte piece q of every 32-channel chunk
  const int q = tid & 7, r0 = tid >> 3;
  const int kchunks = p.Cin / TK_BK;
  for (int c = 0; c < kchunks; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 32 * i;
      const int64_t v = v0 + row;
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
      if (v < p.vox_total) a = *reinterpret_cast<const f32x4*>(x + v * p.ldx + c * TK_BK + q * 4);
      *reinterpret_cast<f32x4*>(As + row * p.lda + c * TK_BK + q * 4) = a;
    }

  // weight chunk (phase ph, K-chunk c) -> registers: NT*32 rows x 128 bytes, row = r0 + 32 j
  f32x4 rb[NT];
  auto fetch_b = [&](int ph, int c) {
    const float* w = wp + ((int64_t)p.tap[ph] * p.Npad) * p.Cin + c * TK_BK + q * 4;
#pragma unroll
    for (int j = 0; j < NT; ++j) rb[j] = *reinterpret_cast<const f32x4*>(w + (int64_t)(r0 + 32 * j) * p.Cin);
  };
  auto commit_b = [&](int buf) {
    float* b = Bs + buf * NT * 32 * TK_LD;
#pragma unroll
    for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(b + (r0 + 32 * j) * TK_LD + q * 4) = rb[j];
  };

  const int arow = wave * 32 + (lane & 31), koff = 4 * (lane >> 5), chalf = lane >> 5;
  const int steps = p.nph * kchunks;
  fetch_b(0, 0);
  commit_b(0);
  __syncthreads();

  f32x16 acc[NT];
  int ph = 0, c = 0;
  for (int s = 0; s < steps; ++s) {
    if (c == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
    // next step's weights in flight under this step's MFMAs
    int nph_ = ph, nc = c + 1;
    if (nc == kchunks) { nc = 0; ++nph_; }
    const bool more = s + 1 < steps;
    if (more) fetch_b(nph_, nc);
    const float* a = As + arow * p.lda + c * TK_BK + koff;
    const float* b = Bs + (s & 1) * NT * 32 * TK_LD + (lane & 31) * TK_LD + koff;
#pragma unroll
    for (int kk = 0; kk < TK_BK / 8; ++kk) {
      const f32x4 fa = *reinterpret_cast<const f32x4*>(a + kk * 8);
      f32x4 fb[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * TK_LD + kk * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[j][e], acc[j], 0, 0, 0);
    }
    if (more) commit_b((s + 1) & 1);      // the other buffer: its last readers finished before the previous barrier
    if (c == kchunks - 1) {               // phase complete: bias + store, 128-byte lines (lanes = channels)
      const int yo = p.yoff[ph];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = j * 32 + (lane & 31);
        const bool colok = col < p.Cout;
        const float bv = (p.bias != nullptr && colok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
          const int off = row_out[row];
          if (off >= 0 && colok) y[(int64_t)(off + yo) * p.ldy + col] = acc[j][r] + bv;
        }
      }
    }
    __syncthreads();
    ph = nph_;
    c = nc;
  }
}

// ---------------------------------------------------------------------------------------------------------- bf16
typedef __bf16 tk_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int TK_LDH = 40;   // bf16 elements per weight-tile row (32 + 8: 80 bytes, conflict-free b128 reads)

template <int NT>
__global__ __launch_bounds__(TK_THREADS, 2) void tconv_ks_bf16_kernel(const TconvKsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  __bf16* As = (__bf16*)smem_h;                             // [128][lda]
  __bf16* Bs = As + TK_BM * p.lda;                          // [2][NT*32][TK_LDH]
  int* row_out = (int*)(Bs + 2 * NT * 32 * TK_LDH);
  const __bf16* x = (const __bf16*)p.x;
  const __bf16* wp = (const __bf16*)p.wp;
  __bf16* y = (__bf16*)p.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t v0 = (int64_t)blockIdx.x * TK_BM;

  if (tid < TK_BM) {
    const int64_t v = v0 + tid;
    int off = -1;
    if (v < p.vox_total) {
      const int w = (int)(v % p.Wi);
      int64_t t = v / p.Wi;
      const int h = (int)(t % p.Hi); t /= p.Hi;
      const int dd = (int)(t % p.Di);
      const int n = (int)(t / p.Di);
      off = ((n * p.Do + dd * p.sd) * p.Ho + h * p.sh) * p.Wo + w * p.sw;
    }
    row_out[tid] = off;
  }
  // input tile: a 32-channel chunk of a row is 64 bytes = 4 pieces of 8 bf16; 64 rows per pass
  const int q = tid & 3, r0 = tid >> 2;
  const int kchunks = p.Cin / TK_BK;
  for (int c = 0; c < kchunks; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + 64 * i;
      const int64_t v = v0 + row;
      tk_bf16x8 a;
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = (__bf16)0.f;
      if (v < p.vox_total) a = *reinterpret_cast<const tk_bf16x8*>(x + v * p.ldx + c * TK_BK + q * 8);
      *reinterpret_cast<tk_bf16x8*>(As + row * p.lda + c * TK_BK + q * 8) = a;
    }

  constexpr int NBR = (NT * 32 + 63) / 64;     // weight rows per thread
  tk_bf16x8 rb[NBR];
  auto fetch_b = [&](int ph, int c) {
    const __bf16* w = wp + ((int64_t)p.tap[ph] * p.Npad) * p.Cin + c * TK_BK + q * 8;
#pragma unroll
    for (int j = 0; j < NBR; ++j) {
      const int row = r0 + 64 * j;
      if (row < NT * 32) rb[j] = *reinterpret_cast<const tk_bf16x8*>(w + (int64_t)row * p.Cin);
    }
  };
  auto commit_b = [&](int buf) {
    __bf16* b = Bs + buf * NT * 32 * TK_LDH;
#pragma unroll
    for (int j = 0; j < NBR; ++j) {
      const int row = r0 + 64 * j;
      if (row < NT * 32) *reinterpret_cast<tk_bf16x8*>(b + row * TK_LDH + q * 8) = rb[j];
    }
  };

  const int arow = wave * 32 + (lane & 31), koff = 8 * (lane >> 5), chalf = lane >> 5;
  const int steps = p.nph * kchunks;
  fetch_b(0, 0);
  commit_b(0);
  __syncthreads();

  f32x16 acc[NT];
  int ph = 0, c = 0;
  for (int s = 0; s < steps; ++s) {
    if (c == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
    int nph_ = ph, nc = c + 1;
    if (nc == kchunks) { nc = 0; ++nph_; }
    const bool more = s + 1 < steps;
    if (more) fetch_b(nph_, nc);
    const __bf16* a = As + arow * p.lda + c * TK_BK + koff;
    const __bf16* b = Bs + (s & 1) * NT * 32 * TK_LDH + (lane & 31) * TK_LDH + koff;
#pragma unroll
    for (int kk = 0; kk < TK_BK / 16; ++kk) {      // v_mfma_f32_32x32x16_bf16: 8 k per lane half
      const tk_bf16x8 fa = *reinterpret_cast<const tk_bf16x8*>(a + kk * 16);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const tk_bf16x8 fb = *reinterpret_cast<const tk_bf16x8*>(b + j * 32 * TK_LDH + kk * 16);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
      }
    }
    if (more) commit_b((s + 1) & 1);
    if (c == kchunks - 1) {
      const int yo = p.yoff[ph];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = j * 32 + (lane & 31);
        const bool colok = col < p.Cout;
        const float bv = (p.bias != nullptr && colok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
          const int off = row_out[row];
          if (off >= 0 && colok) y[(int64_t)(off + yo) * p.ldy + col] = (__bf16)(acc[j][r] + bv);
        }
      }
    }
    __syncthreads();
    ph = nph_;
    c = nc;
  }
}

// (A persistent variant with ALL weights of the layer resident in LDS and the next input tile prefetched into registers
// -- no barrier inside a tile -- was measured in round 3 and lost: one block of 4 waves per CU cannot hide the store and
// LDS latencies that two or three co-resident streaming blocks hide; 64 -> 32 at 2 x 64^3: 432 us fp32 / 302 us bf16
// against 345 / 119 us for the kernels above and 450 / 292 us for the generic grid, tools/bench_tconv_ks.py.)

// The `count` descriptors are the stride phases of ONE kernel == stride transposed convolution
bool tconv_ks_match(const rehr_gather_gemm_desc* ds, int count, bool bf16, TconvKsParams& p) {
  if (count < 2 || count > TK_MAXPH) return false;
  const rehr_gather_gemm_desc& d0 = ds[0];
  if (d0.x2 != nullptr || d0.c1 != d0.Cin || d0.Cin % TK_BK || d0.Cin > 128 || d0.Npad % 32 || d0.Npad > 128) return false;
  if (d0.act != REHR_ACT_NONE || d0.stats_mode != 0 || (d0.flags & REHR_GG_Y_F32)) return false;
  if (d0.osd * d0.osh * d0.osw != count) return false;
  if (d0.Dy != d0.Di * d0.osd || d0.Hy != d0.Hi * d0.osh || d0.Wy != d0.Wi * d0.osw) return false;
  if ((int64_t)d0.N * d0.Dy * d0.Hy * d0.Wy >= (1ll << 31)) return false;
  if (d0.KH != d0.osh || d0.KW != d0.osw) return false;
  unsigned seen = 0;
  for (int i = 0; i < count; ++i) {
    const rehr_gather_gemm_desc& d = ds[i];
    if (d.x1 != d0.x1 || d.wp != d0.wp || d.y != d0.y || d.bias != d0.bias || d.x2 != nullptr || d.Cin != d0.Cin ||
        d.ldx1 != d0.ldx1 || d.ldy != d0.ldy || d.Cout != d0.Cout || d.Npad != d0.Npad || d.N != d0.N)
      return false;
    if (d.td.count != 1 || d.th.count != 1 || d.tw.count != 1 || d.td.off0 || d.th.off0 || d.tw.off0) return false;
    if (d.sd != 1 || d.sh != 1 || d.sw != 1 || d.bd || d.bh || d.bw) return false;
    if (d.Ld != d0.Di || d.Lh != d0.Hi || d.Lw != d0.Wi || d.Di != d0.Di || d.Hi != d0.Hi || d.Wi != d0.Wi) return false;
    if (d.osd != d0.osd || d.osh != d0.osh || d.osw != d0.osw || d.Dy != d0.Dy || d.Hy != d0.Hy || d.Wy != d0.Wy) return false;
    if (d.obd != d.td.k0 || d.obh != d.th.k0 || d.obw != d.tw.k0) return false;
    if (d.obd < 0 || d.obd >= d.osd || d.obh < 0 || d.obh >= d.osh || d.obw < 0 || d.obw >= d.osw) return false;
    if (d.act != REHR_ACT_NONE || d.stats_mode != 0 || d.flags != d0.flags) return false;
    const int id = (d.obd * d.osh + d.obh) * d.osw + d.obw;
    if (seen & (1u << id)) return false;
    seen |= 1u << id;
    p.tap[i] = (d.td.k0 * d.KH + d.th.k0) * d.KW + d.tw.k0;
    p.yoff[i] = (d.obd * d.Hy + d.obh) * d.Wy + d.obw;
  }
  const int align = bf16 ? 15 : 15;
  if (((uintptr_t)d0.x1 | (uintptr_t)d0.wp) & align) return false;
  if (bf16 ? (d0.ldx1 % 8 || d0.ldy % 2) : (d0.ldx1 % 4)) return false;
  p.x = d0.x1; p.wp = d0.wp; p.bias = d0.bias; p.y = d0.y;
  p.ldx = d0.ldx1; p.ldy = d0.ldy; p.Cin = d0.Cin; p.Cout = d0.Cout; p.Npad = d0.Npad;
  p.N = d0.N; p.Di = d0.Di; p.Hi = d0.Hi; p.Wi = d0.Wi; p.Do = d0.Dy; p.Ho = d0.Hy; p.Wo = d0.Wy;
  p.sd = d0.osd; p.sh = d0.osh; p.sw = d0.osw;
  p.nph = count;
  p.vox_total = (int64_t)d0.N * d0.Di * d0.Hi * d0.Wi;
  p.lda = bf16 ? d0.Cin + 8 : d0.Cin + 4;
  return true;
}

template <int NT>
int tconv_launch(const TconvKsParams& p, bool bf16, hipStream_t stream) {
  const int64_t blocks = (p.vox_total + TK_BM - 1) / TK_BM;
  if (blocks >= (1ll << 31)) return REHR_ENOSUP;
  if (!bf16 && p.Cin > 64) return REHR_ENOSUP;   // fp32, 128 input channels: one block per CU (LDS) -- the generic grid is faster
  if (bf16) {
    const size_t smem = ((size_t)TK_BM * p.lda + 2 * NT * 32 * TK_LDH) * 2 + TK_BM * sizeof(int);
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)tconv_ks_bf16_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) !=
          hipSuccess)
        return REHR_EHIP;
      attr = true;
    }
    hipLaunchKernelGGL(tconv_ks_bf16_kernel<NT>, dim3((unsigned)blocks), dim3(TK_THREADS), smem, stream, p);
  } else {
    const size_t smem = ((size_t)TK_BM * p.lda + 2 * NT * 32 * TK_LD) * sizeof(float) + TK_BM * sizeof(int);
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)tconv_ks_f32_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024) !=
          hipSuccess)
        return REHR_EHIP;
      attr = true;
    }
    hipLaunchKernelGGL(tconv_ks_f32_kernel<NT>, dim3((unsigned)blocks), dim3(TK_THREADS), smem, stream, p);
  }
  return REHR_OK;
}

}  // namespace

// REHR_OK launched; REHR_ENOSUP: not the phases of a kernel == stride transposed convolution this kernel takes
int tconv_ks_try(const rehr_gather_gemm_desc* ds, int count, bool bf16, hipStream_t stream) {
  if (ds[0].debug_flags & REHR_DBG_GG_NO_TCONV_KS) return REHR_ENOSUP;
  TconvKsParams p;
  if (!tconv_ks_match(ds, count, bf16, p)) return REHR_ENOSUP;
  int rc;
  switch (p.Npad / 32) {
    case 1: rc = tconv_launch<1>(p, bf16, stream); break;
    case 2: rc = tconv_launch<2>(p, bf16, stream); break;
    case 3: rc = tconv_launch<3>(p, bf16, stream); break;
    default: rc = tconv_launch<4>(p, bf16, stream); break;
  }
  if (rc != REHR_OK) return rc;
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
