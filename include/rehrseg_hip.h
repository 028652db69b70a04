/*
 * rehrseg_hip.h -- C-ABI of librehrseg_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the 3D-convolutional hot path of zhiyuns/REHRSeg.  The
 * reference has no FFI of its own: its operator layer is torch.nn
 * (Conv3d / ConvTranspose3d / InstanceNorm3d / LeakyReLU / F.interpolate, see
 * the citations on every entry point).  These entry points are what a
 * maintainer would bind with ctypes in place of those torch.nn calls
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (no allocation
 *     happens behind the ABI; scratch comes from a caller-provided workspace);
 *   - activations are fp32, channel-innermost ("NDHWC"): element (n,d,h,w,c) of
 *     a tensor with voxel stride `ld` lives at ((n*D+d)*H+h)*W+w)*ld + c, so a
 *     channel slice of a wider buffer is addressed by offsetting the pointer
 *     and keeping `ld` (this is how skip concatenation costs nothing);
 *   - `stream` is a hipStream_t passed as void*; launches are stream ordered,
 *     re-entrant, and never synchronise;
 *   - return value 0 = launched; <0 = rejected argument (REHR_E*), nothing was
 *     launched.  No exceptions, no global state.
 */
#ifndef REHRSEG_HIP_H
#define REHRSEG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REHR_OK 0
#define REHR_EINVAL -1   /* malformed descriptor (sizes, alignment, null) */
#define REHR_ENOSUP -2   /* valid but not supported by this build */
#define REHR_EHIP -3     /* hip launch error */

/* activation codes used by the fused epilogues */
#define REHR_ACT_NONE 0
#define REHR_ACT_RELU 1
#define REHR_ACT_LRELU 2 /* slope given separately */

/* Taps of one spatial axis, described arithmetically (no device tables):
 * tap j (0 <= j < count) reads the source at  base + off0 + offs*j  and uses
 * weight index  k0 + ks*j  along this axis.                                   */
typedef struct {
  int32_t count, off0, offs, k0, ks;
} rehr_axis_taps;

/* ------------------------------------------------------------------------- *
 * Gather-GEMM: the one contraction behind Conv3d forward, Conv3d input
 * gradient (one launch per stride phase), ConvTranspose3d forward (one launch
 * per stride phase), ConvTranspose3d input gradient, and Conv2d (D = 1).
 *
 *   for every lattice point o=(od,oh,ow), sample n, output channel co:
 *     y[n, o*os+ob, co] = act( bias[co] + sum_{taps j} sum_{ci}
 *           x[n, o*s+b+off(j), ci] * wp[tap(j)][co][ci] )
 *   (source positions outside [0,Di)x[0,Hi)x[0,Wi) read as zero)
 *
 * Replaces: torch.nn.Conv3d.forward as used at models/FLAVR/resnet_3D.py:19-33,
 * 42-50, 196-200; models/FLAVR/FLAVR_arch.py:50,77,145,153-156;
 * models/seg_model.py:197-199 and the PlainConvUNet blocks behind
 * models/seg_model.py:174-191, plus the autograd input-gradient of each.
 * ------------------------------------------------------------------------- */
typedef struct {
  /* source tensor(s): channels [0,c1) come from x1, [c1,Cin) from x2 (virtual
   * concat, models/FLAVR/FLAVR_arch.py:14-21); x2 may be NULL when c1 == Cin. */
  const float* x1;
  const float* x2;
  int32_t c1, ldx1, ldx2;
  int32_t N, Di, Hi, Wi, Cin;      /* Cin % 16 == 0; c1 % 32 == 0 when x2 is used */
  /* lattice and its map into the source */
  int32_t Ld, Lh, Lw;
  int32_t sd, sh, sw, bd, bh, bw;
  rehr_axis_taps td, th, tw;
  int32_t KH, KW;                  /* weight tap index = (kd*KH+kh)*KW+kw */
  /* packed weights wp[tap][Npad][Cin] (ci innermost), Npad % 32 == 0 */
  const float* wp;
  int32_t Npad;
  /* destination tensor and the lattice's map into it */
  float* y;
  int32_t Dy, Hy, Wy, Cout, ldy;
  int32_t osd, osh, osw, obd, obh, obw;
  /* fused epilogue */
  const float* bias;               /* NULL or [Cout] */
  int32_t act;
  float slope;
  /* optional per-(n,co) statistics of the stored value, accumulated with
   * double atomics into stats[n][co][2] = {sum, sum of squares}; 0 = off,
   * 1 = sum only (SEGating pool), 2 = sum and squares (InstanceNorm3d).       */
  double* stats;
  int32_t stats_mode;
  /* lattice tile: a td x th x tw brick with td*th*tw == 128, or tile_d == 0
   * for runs of 128 consecutive flattened lattice points (odd extents)         */
  int32_t tile_d, tile_h, tile_w;
  /* optional scratch for the Winograd F(2x2,3x3)-over-(H,W) kernel (NULL = never use
   * it).  When given, >= rehr_gather_gemm_wino_bytes(d) bytes, 16-byte aligned, and
   * the contraction is a unit-stride one with taps {-1,0,+1} along H and W, the
   * library transforms wp into it and runs the 2.25x-fewer-multiplications path
   * (same fp32 results up to the transform's rounding, ~1e-6 relative).          */
  float* wino_ws;
  int64_t wino_ws_bytes;
  int32_t flags;                   /* REHR_GG_* bits; 0 = the library picks the measured-best kernel */
  int32_t debug_flags;             /* REHR_DBG_GG_* bits (below): kernel-selection overrides for tests and A/B
                                    * runs.  NOT part of the stable interface -- integrators leave it 0.        */
} rehr_gather_gemm_desc;

/* bf16 entry points: store y as fp32 instead of bf16 (logits, features handed to fp32 losses) */
#define REHR_GG_Y_F32 1
/* fp32 entry points, wino_ws given.  The Winograd-domain weights in wino_ws depend on wp and on the descriptor's
 * geometry only, so a caller whose weights have not changed since an earlier call with the same geometry may keep the
 * scratch and skip the transform launch:
 *   REHR_GG_WS_ONLY   run ONLY the weight transform(s) this descriptor (or this multi-call) would run: wino_ws is
 *                     filled, nothing else is launched, x1 / x2 / y / bias / stats are not dereferenced (REHR_OK also
 *                     when no Winograd kernel takes the descriptor: nothing to prepare);
 *   REHR_GG_WS_READY  wino_ws already holds what REHR_GG_WS_ONLY (or a full call) with the same geometry, the same wp
 *                     contents and the same debug_flags wrote: skip the transform.
 * In a multi-call every descriptor carries the same two bits. */
#define REHR_GG_WS_READY 2
#define REHR_GG_WS_ONLY 4

/* debug_flags (unstable; the parity tests compare kernel organisations with each other through them):
 * bf16: never take the LDS halo-brick kernel (per-tap gather kernel instead) */
#define REHR_DBG_GG_NO_HALO 1
/* fp32: the flattened-tile Winograd kernels (wino_flat8_conv.hip, wino22_flat) off / forced to 32 or 64 tiles per block */
#define REHR_DBG_GG_NO_FLAT8 2
#define REHR_DBG_GG_FLAT8_HALF 4
#define REHR_DBG_GG_FLAT8_FULL 8
/* fp32: the 32-channel-tile Winograd kernel forced to two 256-thread blocks per CU / one 512-thread block */
#define REHR_DBG_GG_W32P_TWO_PER_CU 16
#define REHR_DBG_GG_W32P_ONE_PER_CU 32
/* multi-phase launches with equal tile counts: the phases of a lattice tile as consecutive blocks of one XCD instead of
 * blockIdx.z = phase.  Measured SLOWER (profiles/r03_ab_phase_interleave.txt: +1.4 ms per cfg-3 step, +0.25 ms cfg-5),
 * kept as a test-selectable organisation only. */
#define REHR_DBG_GG_INTERLEAVE 64
/* kernel == stride transposed convolutions: the generic one-block-per-(tile, phase) grid instead of the fused kernel
 * that stages a tile once for all phases (tconv_ks.hip) */
#define REHR_DBG_GG_NO_TCONV_KS 128
/* fp32 Winograd kernels: tiles in slice-major order (bw, bh, od) instead of the band-major order (bw, od, bh) that keeps
 * the depth neighbours of a tile on one XCD */
#define REHR_DBG_GG_SLICE_MAJOR 256

/* scratch bytes the Winograd path needs for this descriptor; 0 = not applicable */
int64_t rehr_gather_gemm_wino_bytes(const rehr_gather_gemm_desc* d);
int rehr_gather_gemm_f32(const rehr_gather_gemm_desc* d, void* stream);
/* The stride phases of ONE layer (same x1/x2, wp, y, N, Npad; count <= 8), differing
 * only in lattice, taps and destination offset, in a single grid.                   */
int rehr_gather_gemm_multi_f32(const rehr_gather_gemm_desc* descs, int32_t count,
                               void* stream);
/* Mixed-precision variants (BASELINE.json configs[4]: bf16 conv inputs / weights, fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16, fp32 bias / fp64 statistics formed from the fp32 accumulators): the SAME descriptor
 * with x1, x2, wp and y pointing at bf16 elements (ld* and channel counts stay in ELEMENTS; rows must be
 * 16-byte aligned: ldx % 8 == 0; c1 % 32 == 0 when x2 is used), y fp32 with REHR_GG_Y_F32; wino_ws is
 * ignored (the bf16 pipe is 16x the fp32 one: the direct contraction is operand-bandwidth bound already).
 * rehr_pack_weights_bf16: the torch parameter layout in[a][b][t] (transpose_ab: in[b][a][t]) fp32 ->
 * out[t][Apad][B] bf16 -- master weights stay fp32.
 * Replaces the same torch.nn.Conv3d / ConvTranspose3d call sites under bf16 autocast.                */
int rehr_gather_gemm_bf16(const rehr_gather_gemm_desc* d, void* stream);
int rehr_gather_gemm_multi_bf16(const rehr_gather_gemm_desc* descs, int32_t count, void* stream);
int rehr_pack_weights_bf16(const float* in, void* out, int32_t A, int32_t Apad, int32_t B, int32_t T,
                           int32_t transpose_ab, void* stream);

/* Split-K combine: y[row][c] = act(bias[c] + sum_s slabs[s*slab_stride + row*C + c]).
 * A contraction with few lattice tiles and very many taps (feature_fuse:
 * models/FLAVR/FLAVR_arch.py:145, 16384 voxels x 1152 taps) is launched as S
 * tap ranges writing S dense slabs through rehr_gather_gemm_multi_f32 (descs that
 * differ in y), then combined here in a fixed order.                            */
int rehr_sum_slabs_bias_act_f32(const float* slabs, int32_t S, int64_t slab_stride,
                                const float* bias, float* y, int64_t rows, int32_t C,
                                int32_t act, float slope, void* stream);

/* The same combine with the statistics epilogue of the gather-GEMM: stats[N][C][2] += {sum, sum of
 * squares} of the stored value per sample and channel (y is dense [N][SV][C]; stats pre-zeroed by
 * the caller).  Lets the low-resolution nnU-Net stages (<= 8^3 voxels, 320 channels: 40 tiles on
 * 256 CUs) run split over their taps.                                                       */
int rehr_sum_slabs_stats_f32(const float* slabs, int32_t S, int64_t slab_stride, const float* bias,
                             float* y, int32_t N, int64_t SV, int32_t C, int32_t act, float slope,
                             double* stats, void* stream);

/* Mixed precision: the same two combines with fp32 slabs (the bf16 gather-GEMM writes them with REHR_GG_Y_F32) and a
 * bf16 result; statistics are formed from the fp32 sums, like the conv epilogues.                  */
int rehr_sum_slabs_bias_act_bf16(const float* slabs, int32_t S, int64_t slab_stride, const float* bias, void* y,
                                 int64_t rows, int32_t C, int32_t act, float slope, void* stream);
int rehr_sum_slabs_stats_bf16(const float* slabs, int32_t S, int64_t slab_stride, const float* bias, void* y,
                              int32_t N, int64_t SV, int32_t C, int32_t act, float slope, double* stats,
                              void* stream);

/* ------------------------------------------------------------------------- *
 * Weight gradient (Conv3d.weight.grad / ConvTranspose3d.weight.grad):
 *
 *   dst[a*dst_sa + c*dst_sc + tap*dst_st] = sum_{n, lattice o}
 *         l[n, o, a] * g[n, o*s+b+off(tap), c]
 *
 * l = the tensor living ON the lattice (dY for Conv3d, the input for
 * ConvTranspose3d), g = the gathered tensor (X for Conv3d, dY for
 * ConvTranspose3d).  Deterministic: partial sums go to `workspace` slabs
 * and are reduced in a fixed order.
 * Replaces autograd's conv weight gradient for the call sites listed above.
 * ------------------------------------------------------------------------- */
typedef struct {
  const float* l;
  int32_t ldl, Ca;                 /* Ca % 4 == 0 */
  const float* g;
  int32_t ldg, Cg;                 /* Cg % 4 == 0 */
  int32_t N, Ld, Lh, Lw;           /* lattice (= spatial dims of l) */
  int32_t Dg, Hg, Wg;              /* spatial dims of g */
  int32_t sd, sh, sw, bd, bh, bw;
  rehr_axis_taps td, th, tw;
  int32_t KH, KW;
  float* dst;
  int64_t dst_sa, dst_sc, dst_st;
  int32_t accumulate;              /* 0: dst = sum, 1: dst += sum */
  float* workspace;                /* >= rehr_wgrad_workspace_bytes() */
  int64_t workspace_bytes;
  float* dbias;                    /* NULL or [Ca]: dbias[a] (+)= sum l[...,a] */
  int32_t flags;                   /* reserved, 0 */
  int32_t debug_flags;             /* REHR_DBG_WGRAD_* bits: kernel-selection overrides for tests and A/B runs;
                                    * NOT part of the stable interface -- integrators leave it 0.  Everything that
                                    * selects a kernel travels in the descriptors: the library reads no environment
                                    * variables and keeps no mutable state between calls.                        */
} rehr_wgrad_desc;

/* debug_flags: force the direct (slab / brick) kernels even where a transform-domain (Winograd) kernel applies */
#define REHR_DBG_WGRAD_DIRECT 1
/* Winograd weight gradient: walk every output slice for every depth tap (default: a tap skips the slices whose source
 * slice lies outside the volume; tests compare both) */
#define REHR_DBG_WGRAD_NO_TAP_SKIP 2
/* Winograd weight gradient: blockIdx.z = depth tap instead of the taps of a split as consecutive blocks of one XCD */
#define REHR_DBG_WGRAD_NO_TAP_COLOCATE 4

/* Mixed-precision weight gradient: l and g point at bf16 elements (ld* in elements, % 8 == 0; Ca, Cg % 8 == 0),
 * fp32 accumulation on v_mfma_f32_32x32x16_bf16, fp32 slabs and fp32 dst (the master-weight gradient).  dbias must
 * be NULL (the bias gradient is a column sum of dY: rehr_channel_sum).  Own workspace query.                  */
int64_t rehr_wgrad_bf16_workspace_bytes(const rehr_wgrad_desc* d);
int rehr_wgrad_bf16(const rehr_wgrad_desc* d, void* stream);
int64_t rehr_wgrad_workspace_bytes(const rehr_wgrad_desc* d);
/* 1 when rehr_wgrad_f32 will take the Winograd path for this descriptor (unit stride,
 * taps {-1,0,+1} over H and W, >= 64 channels on both sides): the products are formed
 * in the transform domain, 16 per 2x2 output tile and depth tap instead of 36.       */
int rehr_wgrad_uses_winograd(const rehr_wgrad_desc* d);
int rehr_wgrad_f32(const rehr_wgrad_desc* d, void* stream);

/* out[t][a][b] = in[a][b][t] (transpose_ab = 0) or in[b][a][t] (1); rows
 * a >= A are written as zero up to Apad.  Turns the (Cout,Cin,kD,kH,kW)
 * parameter layout of torch.nn.Conv3d / the (Cin,Cout,...) layout of
 * ConvTranspose3d into wp[tap][Npad][Cin].                                   */
int rehr_pack_weights_f32(const float* in, float* out, int32_t A, int32_t Apad,
                          int32_t B, int32_t T, int32_t transpose_ab,
                          void* stream);

/* ------------------------------------------------------------------------- *
 * Direct convolution for Cin in {1,2} -> Cout in {16,32,64} (HBM bound, no
 * MFMA): the stem Conv3d (3,7,7)/(1,2,2) of models/FLAVR/resnet_3D.py:47-48
 * and the first nnU-Net conv.  Weights stay in the torch parameter layout
 * (Cout,Cin,kD,kH,kW); x/y are NDHWC.  The input of these layers is the
 * network input, so no input gradient exists.
 * ------------------------------------------------------------------------- */
typedef struct {
  const float* x;
  int32_t ldx, N, Di, Hi, Wi, Cin;
  const float* w;                  /* (Cout,Cin,KD,KH,KW) */
  const float* bias;               /* NULL or [Cout] */
  float* y;                        /* forward: output; wgrad: dY (read) */
  int32_t ldy, Do, Ho, Wo, Cout;
  int32_t KD, KH, KW, sd, sh, sw, pd, ph, pw;
  int32_t act;
  float slope;
  double* stats;
  int32_t stats_mode;
} rehr_direct_conv_desc;

int rehr_conv_small_cin_fwd_f32(const rehr_direct_conv_desc* d, void* stream);
/* out[n,o][k], k = ci*T + tap (order of the (Cin,kD,kH,kW) weight), zero for
 * k >= Cin*T and in the padding; Kpad % 32 == 0.  The weight gradient of the thin
 * conv is then rehr_wgrad_f32 with l = dY, g = out (1x1x1 taps) on the MFMA path. */
int rehr_im2col_f32(const rehr_direct_conv_desc* d, float* out, int32_t Kpad,
                    void* stream);
int64_t rehr_conv_small_cin_wgrad_workspace_bytes(const rehr_direct_conv_desc* d);
/* 1 when rehr_conv_small_cin_wgrad_f32 takes this shape on the matrix cores (thin_cin_conv.hip: C_out 32 / 64,
 * 1x3x3 / 3x3x3 / (3,7,7) taps, stride_w <= 2) -- dY and x are then each read once, no im2col columns */
int rehr_conv_small_cin_wgrad_on_mfma(const rehr_direct_conv_desc* d);
/* Mixed precision: the same layers with y (forward) / dY (weight gradient) pointing at bf16 elements -- the thin-input
 * layers compute in fp32 on the fp32 image, the layer behind takes bf16.  Matrix-core shapes only
 * (rehr_conv_small_cin_wgrad_on_mfma; C_out 32 / 64, kW <= 8, stride_w <= 2 for the forward), else REHR_ENOSUP.
 * Same workspace query as the fp32 entry. */
int rehr_conv_small_cin_fwd_ybf16(const rehr_direct_conv_desc* d, void* stream);
int rehr_conv_small_cin_wgrad_dybf16(const rehr_direct_conv_desc* d, float* dw, float* dbias, float* workspace,
                                     int64_t workspace_bytes, void* stream);
/* dw (Cout,Cin,KD,KH,KW) = sum_{n,o} dY[n,o,co] * x[n,o*s-p+k,ci]; dbias optional */
int rehr_conv_small_cin_wgrad_f32(const rehr_direct_conv_desc* d, float* dw,
                                  float* dbias, float* workspace,
                                  int64_t workspace_bytes, void* stream);

/* Direct convolution for Cout <= 4, Cin % 16 == 0, stride 1, kW <= 7 (HBM/VALU
 * bound): the 1x1x1 segmentation layer (models/seg_model.py:42-44) and sr_head's
 * 5x5x5 16->num_classes conv (models/seg_model.py:199), with both gradients.
 * Same descriptor; `y` is the output (forward) or dY (gradients).              */
int rehr_conv_small_cout_fwd_f32(const rehr_direct_conv_desc* d, void* stream);
int rehr_conv_small_cout_dgrad_f32(const rehr_direct_conv_desc* d, float* dx,
                                   void* stream);
int64_t rehr_conv_small_cout_wgrad_workspace_bytes(const rehr_direct_conv_desc* d);
int rehr_conv_small_cout_wgrad_f32(const rehr_direct_conv_desc* d, float* dw,
                                   float* dbias, float* workspace,
                                   int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * SEGating (models/FLAVR/resnet_3D.py:100-116) and its fused neighbours.
 *   gate[n][c] = sigmoid( b[c] + sum_k W[c][k] * stats[n][k][0] / S )
 *   y = act( x * gate[n][c] + res )      (res optional: BasicBlock :144-149;
 *                                         act lrelu(0.2): FLAVR_arch.py:188-200)
 * ------------------------------------------------------------------------- */
int rehr_se_gate_fwd_f32(const double* stats, const float* w, const float* b,
                         float* gate, float* mean, int32_t N, int32_t C,
                         int64_t S, void* stream);
int rehr_scale_res_act_fwd_f32(const float* x, int32_t ldx, const float* gate,
                               const float* res, int32_t ldr, float* y,
                               int32_t ldy, int32_t N, int64_t S, int32_t C,
                               int32_t act, float slope, void* stream);
/* dz = dy * act'(y); dres = dz (optional); dx = dz * gate;
 * dgate_acc[n][c] += sum_s dz * x   (double)                                 */
int rehr_scale_res_act_bwd_f32(const float* dy, int32_t lddy, const float* y,
                               int32_t ldy, const float* x, int32_t ldx,
                               const float* gate, float* dx, int32_t lddx,
                               float* dres, int32_t lddr, double* dgate_acc,
                               int32_t N, int64_t S, int32_t C, int32_t act,
                               float slope, void* stream);
/* From dgate_acc: ds = dgate*g*(1-g); dW += ds (x) mean; db += ds;
 * kconst[n][c] = (W^T ds)[c] / S.                                             */
int rehr_se_gate_bwd_f32(const double* dgate_acc, const float* gate,
                         const float* mean, const float* w, float* dw,
                         float* db, float* kconst, int32_t N, int32_t C,
                         int64_t S, void* stream);
/* x[n,s,c] += k[n][c] in place */
int rehr_add_channel_const_f32(float* x, int32_t ldx, const float* k, int32_t N,
                               int64_t S, int32_t C, void* stream);

/* ------------------------------------------------------------------------- *
 * InstanceNorm3d(affine, eps) + LeakyReLU applied to a conv output whose
 * {sum, sum of squares} were accumulated by the conv epilogue
 * (nnU-Net ConvDropoutNormReLU: dynamic_network_architectures==0.3.1, call
 * sites models/seg_model.py:174-191).
 *   mean = s1/S; var = s2/S - mean^2 (biased); rstd = 1/sqrt(var+eps)
 *   y = lrelu( (x-mean)*rstd*gamma + beta )
 * ------------------------------------------------------------------------- */
int rehr_instnorm_act_fwd_f32(const float* x, int32_t ldx, const double* stats,
                              const float* gamma, const float* beta, float* y,
                              int32_t ldy, float* mean_rstd /* [N][C][2] */,
                              int32_t N, int64_t S, int32_t C, float eps,
                              int32_t act, float slope, void* stream);
/* dz = dy*act'(xhat*gamma+beta); dgamma = sum dz*xhat; dbeta = sum dz;
 * dx = rstd*gamma*(dz - mean_s(dz) - xhat*mean_s(dz*xhat)).  `red` is a
 * caller-zeroed [N][C][2] double scratch.                                     */
int rehr_instnorm_act_bwd_f32(const float* dy, int32_t lddy, const float* x,
                              int32_t ldx, const float* mean_rstd,
                              const float* gamma, const float* beta, float* dx,
                              int32_t lddx, float* dgamma, float* dbeta,
                              double* red, int32_t N, int64_t S, int32_t C,
                              int32_t act, float slope, void* stream);

/* ------------------------------------------------------------------------- *
 * Depth-only linear upsample, align_corners=True
 * (F.interpolate(scale_factor=(upscale,1,1), mode='trilinear'),
 * models/seg_model.py:204).  Do = floor(Di*upscale).
 * ------------------------------------------------------------------------- */
int rehr_upsample_depth_fwd_f32(const float* x, float* y, int32_t N, int32_t Di,
                                int32_t Do, int64_t HW, int32_t C, void* stream);
int rehr_upsample_depth_bwd_f32(const float* dy, float* dx, int32_t N,
                                int32_t Di, int32_t Do, int64_t HW, int32_t C,
                                void* stream);

/* Depth upsample fused with the depth-tap sum of the convolution that follows it
 * (models/seg_model.py:204-205: F.interpolate(features, (upscale,1,1), trilinear, align_corners) ->
 * sr_head[0] = Conv3d(32, 16, 3, padding=1) -> ReLU).  Both are linear, so the (kH,kW) part of every
 * depth tap kd is taken on the low-resolution slices first -- g[n][j][hw][kd*C + co], a (1,kH,kW)
 * convolution with KD*C output channels on the ordinary conv path -- and
 *   y[n][d][hw][co] = act(bias[co] + sum_kd [0 <= u < Do] ((1-t) g[n][i0][hw][kd*C+co] + t g[n][i1][..])),
 *   u = d + kd - pd, (i0, i1, t) = align_corners source of upsampled slice u.
 * bwd: dg from dz = dy * act'(y), formed on the fly from dy and the saved output y.  C % 4 == 0.    */
int rehr_upmix_depth_fwd_f32(const float* g, const float* bias, float* y, int32_t N,
                             int32_t Di, int32_t Do, int64_t HW, int32_t C, int32_t KD,
                             int32_t pd, int32_t act, float slope, void* stream);
int rehr_upmix_depth_bwd_f32(const float* dy, const float* y, float* dg, int32_t N,
                             int32_t Di, int32_t Do, int64_t HW, int32_t C, int32_t KD,
                             int32_t pd, int32_t act, float slope, void* stream);
/* out[c] = sum over rows of dy[row][c] * act'(y[row][c]): the bias gradient behind a fused activation */
int rehr_channel_sum_actgrad_f32(const float* dy, const float* y, int32_t ld, int64_t rows,
                                 int32_t C, int32_t act, float slope, float* out,
                                 double* scratch, void* stream);

/* cosine_distance_loss (models/seg_model.py:60-78) on two (N, 64, D, H, W) NDHWC tensors, S = D*H*W:
 * stats[n][c] = (S12, S11, S22) of the per-voxel channel-normalised tensors (fp64, zeroed by the call);
 * loss = mean_{n,c}(1 - S12 / (max(sqrt(S11), 1e-8) * max(sqrt(S22), 1e-8))) is three lines of host code.
 * bwd: gradient w.r.t. x1 only (x2 = the frozen teacher), scale = -dloss / (N*C).                   */
int rehr_cosdist_stats_f32(const float* x1, const float* x2, double* stats, int32_t N,
                           int64_t S, int32_t C, void* stream);
int rehr_cosdist_bwd_f32(const float* x1, const float* x2, const double* stats, float* dx1,
                         int32_t N, int64_t S, int32_t C, float scale, void* stream);

/* UASR head of UNet_3D_3D(use_uncertainty=True) (reference models/FLAVR/FLAVR_arch.py:203-246): per output voxel the
 * K candidate (image, segmentation) pairs are blended with softmax weights, and the weights give the uncertainty:
 *   s = softmax_i ue_i;  out0 = sum_i s_i (tanh(om_2i) + 1)/2;  out1 = sum_i s_i om_{2i+1};
 *   unc = sigmoid(bu + sum_i s_i wu_i)                      (uncertainty_out = Conv3d(K, 1, 1), :151,245)
 * om [n][hw][d*2K + c], ue [n][hw][d*K + i]: the NDHWC outputs of feature_fuse1 / uncertainty_early on the fused
 * slice (the split(dim=1) + stack(dim=2) of :205-211 is this indexing); out (N,2,D,H,W) and unc (N,1,D,H,W) NCDHW.
 * K in {4, 8, 16, 32}; om, ue (dom, due) 16-byte aligned.
 * bwd: dom, due from (gout, gunc) with the softmax recomputed; partial[blocks][K+1] (double) holds per-block sums of
 * (dwu[0..K), dbu) -- blocks = rehr_uasr_mix_blocks(N, D, HW); the host adds them.                       */
int32_t rehr_uasr_mix_blocks(int32_t N, int32_t D, int64_t HW);
int rehr_uasr_mix_fwd_f32(const float* om, const float* ue, const float* wu, const float* bu, float* out,
                          float* unc, int32_t N, int32_t K, int32_t D, int64_t HW, void* stream);
int rehr_uasr_mix_bwd_f32(const float* om, const float* ue, const float* wu, const float* bu,
                          const float* gout, const float* gunc, float* dom, float* due, double* partial,
                          int32_t N, int32_t K, int32_t D, int64_t HW, void* stream);

/* Max-pool of every (sample, depth) slice with a (H/2, W/2) window and stride: the 2x2 summary per slice and
 * channel that Distiller's structure loss compares (CriterionPairWiseforWholeFeatAfterPool,
 * models/seg_model.py:95-113; MaxPool2d(ceil_mode=True) with even H, W).  x [slices][H][W][C] (NDHWC slices),
 * y [slices][2][2][C], idx = row-major position of the first maximum inside its slice (for the backward).
 * bwd: dx is zero-filled, then dx[slice][idx][c] = dy[slice][q][c].                                          */
int rehr_quad_maxpool_fwd_f32(const float* x, float* y, int32_t* idx, int64_t slices,
                              int32_t H, int32_t W, int32_t C, void* stream);
int rehr_quad_maxpool_bwd_f32(const float* dy, const int32_t* idx, float* dx, int64_t slices,
                              int32_t H, int32_t W, int32_t C, void* stream);

/* Stem of the distillation teacher's overlapping 4-slice windows (get_intermediate_features,
 * train_all.py:85-112 -> UNet_3D_3D.forward's mean subtraction FLAVR_arch.py:181 -> BasicStem
 * resnet_3D.py:42-50).  g0..g2 = the (1,kH,kW) part of the stem's three depth taps on every slice of the
 * depth-padded volume (B*nslices slices, NHWC) followed by one more "slice": the response to a
 * constant 1 in channel 0.  y[b*nwin + w][k][hw][c] = act(bias[c] + sum over the taps kd that stay inside
 * the window (0 <= k+kd-1 <= 3) of g[kd][b*nslices + w + k+kd-1] - mean[b*nwin + w] * g[kd][B*nslices]).
 * nslices = nwin + 3.  No gradient (the teacher is frozen).                                           */
int rehr_window_stem_assemble_f32(const float* g0, const float* g1, const float* g2,
                                  const float* mean, const float* bias, float* y, int32_t B,
                                  int32_t nwin, int32_t nslices, int64_t HW, int32_t C,
                                  int32_t act, float slope, void* stream);
/* the same with a bf16 result (mixed precision: the teacher's first block takes bf16); fp32 responses in */
int rehr_window_stem_assemble_bf16(const float* g0, const float* g1, const float* g2,
                                   const float* mean, const float* bias, void* y, int32_t B,
                                   int32_t nwin, int32_t nslices, int64_t HW, int32_t C,
                                   int32_t act, float slope, void* stream);

/* ------------------------------------------------------------------------- *
 * Fused segmentation loss: softmax + cross-entropy, optionally
 * weighted by the uncertainty map with the reference's (B,D,H,W)*(B,1,D,H,W)
 * broadcast, + nnU-Net soft Dice, one pass over the logits each way.
 * Replaces utils/seg_utils.py:289-304 (RobustCrossEntropyLoss), :306-351
 * (DC_and_weighted_CE_loss) and nnunetv2 MemoryEfficientSoftDiceLoss behind them.
 *   logits [N][S][ld] (NDHWC, C <= ld, 2 <= C <= 4, N <= 4), target [N][S] float class
 *   ids, unc NULL or [N][S].  fwd fills stats[N*C*3 + 1] = {I,P,G per (n,c); sum of
 *   (sum_b ce)*(sum_a unc) or sum of ce}; the caller forms the scalar
 *     w_ce * stats[last] / (unc ? N*N*S : N*S)
 *       - w_dice * mean_{n, c >= !do_bg} (2I+smooth)/max(G+P+smooth, 1e-8).
 *   bwd writes dlogits[N][S][ldd] = *grad_out * d(loss)/d(logits).
 * ------------------------------------------------------------------------- */
int rehr_seg_loss_fwd_f32(const float* logits, int32_t ld, const float* target, const float* unc,
                          int32_t N, int32_t C, int64_t S, double* stats, void* stream);
int rehr_seg_loss_bwd_f32(const float* logits, int32_t ld, const float* target, const float* unc,
                          int32_t N, int32_t C, int64_t S, const double* stats, float w_ce,
                          float w_dice, float smooth, int32_t do_bg, const float* grad_out,
                          float* dlogits, int32_t ldd, void* stream);

/* Elementwise helpers used by the blocks above. */
int rehr_act_fwd_f32(const float* x, float* y, int64_t n, int32_t act,
                     float slope, void* stream);
int rehr_act_bwd_f32(const float* dy, const float* y, float* dx, int64_t n,
                     int32_t act, float slope, void* stream);
/* out[c] (+)= sum over rows of x[row*ld + c], rows = N*S (bias gradient).     */
int rehr_channel_sum_f32(const float* x, int32_t ldx, int64_t rows, int32_t C,
                         float* out, int32_t accumulate, double* scratch,
                         void* stream);
/* strided channel-slice copy: y[row*ldy + c] = x[row*ldx + c], c < C          */
int rehr_copy_channels_f32(const float* x, int32_t ldx, float* y, int32_t ldy,
                           int64_t rows, int32_t C, void* stream);
/* NCDHW <-> NDHWC for API-edge tensors                                        */
int rehr_nchw_to_nhwc_f32(const float* x, float* y, int32_t N, int32_t C,
                          int64_t S, void* stream);
int rehr_nhwc_to_nchw_f32(const float* x, float* y, int32_t N, int32_t C,
                          int64_t S, void* stream);

/* BCE-with-logits + Dice of the SR stage's segmentation channel in one pass each way: utils/seg_utils.py:786-885
 * (BCEDiceLoss: alpha * BCEWithLogitsLoss + beta * (1 - mean_c 2 sum(p t) / clamp(sum p^2 + sum t^2, 1e-6)), p = sigmoid(x),
 * sums per channel over batch and space; called from train_sr, train_all.py:134).  x, t dense [N][C][S] fp32;
 * stats[C][4] double = {sum bce, sum p t, sum p^2, sum t^2} (zeroed inside); loss = alpha * sum_c bce / (N C S) +
 * beta * (1 - mean_c dice_c) is formed by the caller; grad_out = device scalar. */
int rehr_bce_dice_fwd_f32(const float* x, const float* t, int32_t N, int32_t C, int64_t S, double* stats, void* stream);
int rehr_bce_dice_bwd_f32(const float* x, const float* t, int32_t N, int32_t C, int64_t S, const double* stats,
                          float alpha, float beta, const float* grad_out, float* dx, void* stream);

/* ------------------------------------------------------------------------- *
 * sr_head.2 of the segmentation model -- Conv3d(16 -> 2, 5x5x5, stride 1, pad 2) on the depth-upsampled features,
 * models/seg_model.py:199, :205 -- on the bf16 matrix cores (mixed-precision path, BASELINE.json configs[4]).
 * Same descriptor as the rehr_conv_small_cout_* entry points with `x` addressing bf16 elements (ldx % 8 == 0),
 * weights / bias fp32 in the torch layout, y (forward output, or dY for the gradients) fp32 NDHWC.
 * Supported: Cin 16, Cout 2, 5x5x5, stride 1, pad 2, Wi % 32 == 0, 32 <= Wi <= 160 (rehr_conv5_thin_supported);
 * anything else returns REHR_ENOSUP.  `workspace`: rehr_conv5_thin_workspace_bytes(d) bytes of device scratch
 * (repacked weights; the weight-gradient partial sums).
 * ------------------------------------------------------------------------- */
int64_t rehr_conv5_thin_workspace_bytes(const rehr_direct_conv_desc* d);  /* < 0: REHR_E* */
int rehr_conv5_thin_supported(const rehr_direct_conv_desc* d);
int rehr_conv5_thin_fwd_bf16(const rehr_direct_conv_desc* d, void* workspace, int64_t workspace_bytes, void* stream);
/* dx (bf16 NDHWC, lddx % 4 == 0) = input gradient of the layer from dY = d->y (fp32) */
int rehr_conv5_thin_dgrad_bf16(const rehr_direct_conv_desc* d, void* dx, int32_t lddx, void* workspace,
                               int64_t workspace_bytes, void* stream);
/* dw (2,16,5,5,5) fp32 = weight gradient from x = d->x (bf16) and dY = d->y (fp32), dbias (NULL or [2]) = column sums
 * of dY; partial sums per block in the workspace, combined in a fixed order (bitwise reproducible). */
int rehr_conv5_thin_wgrad_bf16(const rehr_direct_conv_desc* d, float* dw, float* dbias, void* workspace,
                               int64_t workspace_bytes, void* stream);

/* The same layer on the fp32 matrix cores (v_mfma_f32_16x16x4_f32) for the fp32 path: x, dx fp32 (ldx, lddx % 4 == 0),
 * Wi % 32 == 0, 32 <= Wi <= 128; same descriptor, same workspace protocol (rehr_conv5_thin_f32_workspace_bytes).
 * Replaces rehr_conv_small_cout_{fwd,dgrad,wgrad}_f32 for sr_head.2 where the shape qualifies. */
int64_t rehr_conv5_thin_f32_workspace_bytes(const rehr_direct_conv_desc* d);
int rehr_conv5_thin_f32_supported(const rehr_direct_conv_desc* d);
int rehr_conv5_thin_fwd_f32(const rehr_direct_conv_desc* d, void* workspace, int64_t workspace_bytes, void* stream);
int rehr_conv5_thin_dgrad_f32(const rehr_direct_conv_desc* d, float* dx, int32_t lddx, void* workspace,
                              int64_t workspace_bytes, void* stream);
int rehr_conv5_thin_wgrad_f32(const rehr_direct_conv_desc* d, float* dw, float* dbias, void* workspace,
                              int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Training-patch feed from HBM-resident volumes.
 * Replaces the host-side numpy chain of utils/train_set.py:100-160 (TrainSetMultipleSegSREfficient.__getitem__),
 * :205-224 (TrainSetMultipleSegSR.__getitem__) and :330-434 (TrainSetMultiple.__getitem__): transposition, crop,
 * constant pad (utils/pad.py:14-21), flips, every-k-th slice, astype(float32), transpose / permute.
 *   out[item][o0][o1][o2][o3] = ((lo[k] <= o_k < hi[k] for every k) ? src[base + sum_k o_k * stride[k]] : 0) * scale + bias
 * `items` is a HOST array of n_items descriptors (copied into the kernel arguments, REHR_PATCH_MAX_ITEMS per launch);
 * src / dst are device pointers; strides and base count source elements and may be negative (flips).
 * ------------------------------------------------------------------------- */
#define REHR_PATCH_F32 0
#define REHR_PATCH_U8 1
#define REHR_PATCH_MAX_ITEMS 16
typedef struct {
  const void* src;      /* volume (device), fp32 or uint8 elements */
  int64_t base;         /* element offset of output index (0,0,0,0) */
  int64_t stride[4];    /* source elements per step of each output axis */
  int32_t lo[4], hi[4]; /* output indices outside [lo, hi) read as 0 (the constant pad) */
} rehr_patch_item;
typedef struct {
  int32_t n_items;
  int32_t dims[4];         /* output extent of one item */
  int32_t src_dtype;       /* REHR_PATCH_F32 / REHR_PATCH_U8 */
  float scale, bias;       /* applied after the pad (uncertainty maps: utils/train_set.py:147) */
  float* dst;              /* [n_items][dst_item_stride], fp32 */
  int64_t dst_item_stride; /* >= prod(dims) */
} rehr_patch_gather_desc;
int rehr_patch_gather(const rehr_patch_gather_desc* d, const rehr_patch_item* items, void* stream);

/* 1-D weighted gather along one axis: dst[o][j][i] = sum_t w[j][t] * src[o][idx[j][t]][i] (idx < 0: term dropped).
 * Serves the slice-profile blur (utils/train_set.py:306-318: F.conv2d with the (L,1) kernel of
 * utils/blur_kernel_ops.py:7-18, padding="same") and the 1-D down-sampling of the LR simulation (:403-404).
 * src [outer][n_in][inner], dst [outer][n_out][inner] dense fp32; idx / w [n_out][taps] on the device. */
int rehr_axis_resample_f32(const float* src, float* dst, const int32_t* idx, const float* w, int64_t outer,
                           int32_t n_in, int32_t n_out, int64_t inner, int32_t taps, void* stream);

/* ------------------------------------------------------------------------- *
 * Mixed-precision (*_bf16) variants of the HBM-bound fused-block kernels: the SAME arguments as the *_f32 entry
 * points above with every ACTIVATION pointer (x, y, res, dy, dx, dres) addressing bf16 elements (ld* in elements,
 * % 8 == 0, C % 8 == 0); gates, gamma / beta, mean_rstd stay fp32, statistics and reduction buffers fp64, the
 * arithmetic is fp32 in registers.  BASELINE.json configs[4] (bf16 autocast of the same torch.nn call sites).
 * ------------------------------------------------------------------------- */
int rehr_scale_res_act_fwd_bf16(const void* x, int32_t ldx, const float* gate, const void* res, int32_t ldr, void* y,
                                int32_t ldy, int32_t N, int64_t S, int32_t C, int32_t act, float slope, void* stream);
int rehr_scale_res_act_bwd_bf16(const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x, int32_t ldx,
                                const float* gate, void* dx, int32_t lddx, void* dres, int32_t lddr,
                                double* dgate_acc, int32_t N, int64_t S, int32_t C, int32_t act, float slope,
                                void* stream);
int rehr_upmix_depth_fwd_bf16(const void* g, const float* bias, void* y, int32_t N, int32_t Di, int32_t Do, int64_t HW,
                              int32_t C, int32_t KD, int32_t pd, int32_t act, float slope, void* stream);
int rehr_upmix_depth_bwd_bf16(const void* dz, const void* y, void* dg, int32_t N, int32_t Di, int32_t Do, int64_t HW,
                              int32_t C, int32_t KD, int32_t pd, int32_t act, float slope, void* stream);
int rehr_channel_sum_actgrad_bf16(const void* dy, const void* y, int32_t ld, int64_t rows, int32_t C, int32_t act,
                                  float slope, float* out, double* scratch, void* stream);
int rehr_add_channel_const_bf16(void* x, int32_t ldx, const float* k, int32_t N, int64_t S, int32_t C, void* stream);
int rehr_instnorm_act_fwd_bf16(const void* x, int32_t ldx, const double* stats, const float* gamma, const float* beta,
                               void* y, int32_t ldy, float* mean_rstd, int32_t N, int64_t S, int32_t C, float eps,
                               int32_t act, float slope, void* stream);
int rehr_instnorm_act_bwd_bf16(const void* dy, int32_t lddy, const void* x, int32_t ldx, const float* mean_rstd,
                               const float* gamma, const float* beta, void* dx, int32_t lddx, float* dgamma,
                               float* dbeta, double* red, int32_t N, int64_t S, int32_t C, int32_t act, float slope,
                               void* stream);
/* the same, and dconv_bias[c] = sum over all samples and voxels of the dx it writes: the gradient of the bias of the
 * convolution in front of the normalisation (the mixed-precision weight-gradient kernels carry no bias column; this
 * replaces a separate rehr_channel_sum_bf16 pass over dx).  dsum: [C] doubles of scratch. */
int rehr_instnorm_act_bwd_dbias_bf16(const void* dy, int32_t lddy, const void* x, int32_t ldx, const float* mean_rstd,
                                     const float* gamma, const float* beta, void* dx, int32_t lddx, float* dgamma,
                                     float* dbeta, double* red, int32_t N, int64_t S, int32_t C, int32_t act,
                                     float slope, double* dsum, float* dconv_bias, void* stream);
int rehr_channel_sum_bf16(const void* x, int32_t ldx, int64_t rows, int32_t C, float* out, int32_t accumulate,
                          double* scratch, void* stream);
int rehr_act_bwd_bf16(const void* dy, const void* y, void* dx, int64_t n, int32_t act, float slope, void* stream);

/* ABI version, bumped on any signature change. */
int rehr_abi_version(void);
/* Text of the last HIP launch error seen on the calling thread (diagnostics). */
const char* rehr_last_hip_error(void);

#ifdef __cplusplus
}
#endif
#endif /* REHRSEG_HIP_H */
