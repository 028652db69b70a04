// Training-patch feed with the volumes resident in HBM.
//
// The reference cuts its training patches on the host with numpy (utils/train_set.py:100-160 and :330-434:
// transposition, crop, constant pad, flips, every-k-th slice, astype(float32), permute) and blurs / resamples
// them with torch on the CPU (:306-318 F.conv2d with the slice-profile kernel, :403-404 `resize`).  With 288 GB
// of HBM a whole data set of volumes stays on the card, so one patch is one strided gather:
//
//   rehr_patch_gather     out[item][o0][o1][o2][o3] = (lo <= o < hi on every axis) ? src[base + sum_k o_k * stride_k] : 0,
//                         then * scale + bias; every crop / transposition / flip / sub-sampling / pad of the
//                         reference's __getitem__ is a choice of (base, stride, lo, hi).  HBM-bound byte
//                         moving: 4 (or 1) bytes read + 4 written per output element.
//   rehr_axis_resample    dst[outer][j][inner] = sum_t w[j][t] * src[outer][idx[j][t]][inner] (idx < 0: no term):
//                         the 1-D slice-profile blur with zero "same" padding and the 1-D down-sampling
//                         (nearest / linear / cubic) are tap tables built on the host.
#include "common.h"

namespace {

struct GatherParams {
  rehr_patch_gather_desc d;
  rehr_patch_item it[REHR_PATCH_MAX_ITEMS];
};

template <typename T>
__device__ __forceinline__ float load_as_float(const void* p, int64_t i) {
  return (float)reinterpret_cast<const T*>(p)[i];
}

__device__ __forceinline__ float fetch(const rehr_patch_item& it, int dtype, int64_t off) {
  return dtype == REHR_PATCH_U8 ? load_as_float<uint8_t>(it.src, off) : load_as_float<float>(it.src, off);
}

// one thread per output element, o3 fastest: coalesced writes; reads coalesced when |stride[3]| == 1
__global__ __launch_bounds__(256) void patch_gather_kernel(const GatherParams p) {
  const rehr_patch_item& it = p.it[blockIdx.y];
  const int D1 = p.d.dims[1], D2 = p.d.dims[2], D3 = p.d.dims[3];
  const int64_t total = (int64_t)p.d.dims[0] * D1 * D2 * D3;
  float* dst = p.d.dst + (int64_t)blockIdx.y * p.d.dst_item_stride;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int o3 = (int)(i % D3);
    int64_t r = i / D3;
    const int o2 = (int)(r % D2); r /= D2;
    const int o1 = (int)(r % D1);
    const int o0 = (int)(r / D1);
    const bool ok = (o0 >= it.lo[0]) & (o0 < it.hi[0]) & (o1 >= it.lo[1]) & (o1 < it.hi[1]) & (o2 >= it.lo[2]) &
                    (o2 < it.hi[2]) & (o3 >= it.lo[3]) & (o3 < it.hi[3]);
    float v = 0.f;
    if (ok) v = fetch(it, p.d.src_dtype, it.base + o0 * it.stride[0] + o1 * it.stride[1] + o2 * it.stride[2] + o3 * it.stride[3]);
    dst[i] = v * p.d.scale + p.d.bias;
  }
}

// the source runs fastest along output axis `a` != 3 (a transposing patch): 32 x 32 tiles over (a, 3) go
// through LDS so that both the reads (along a) and the writes (along o3) are contiguous
template <int A>
__global__ __launch_bounds__(256) void patch_gather_tr_kernel(const GatherParams p, const int tiles_a, const int tiles_3) {
  __shared__ float tile[32][33];
  const rehr_patch_item& it = p.it[blockIdx.y];
  constexpr int B0 = A == 0 ? 1 : 0, B1 = A == 2 ? 1 : 2;  // the two remaining axes
  const int DA = p.d.dims[A], D3 = p.d.dims[3], DB1 = p.d.dims[B1];
  int b = blockIdx.x;
  const int t3 = b % tiles_3; b /= tiles_3;
  const int ta = b % tiles_a; b /= tiles_a;
  const int ob1 = b % DB1, ob0 = b / DB1;
  const bool okb = (ob0 >= it.lo[B0]) & (ob0 < it.hi[B0]) & (ob1 >= it.lo[B1]) & (ob1 < it.hi[B1]);
  const int64_t baseb = it.base + ob0 * it.stride[B0] + ob1 * it.stride[B1];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int oa = ta * 32 + tx, o3 = t3 * 32 + ty + 8 * k;
    const bool ok = okb & (oa >= it.lo[A]) & (oa < it.hi[A]) & (o3 >= it.lo[3]) & (o3 < it.hi[3]);
    float v = 0.f;
    if (ok) v = fetch(it, p.d.src_dtype, baseb + oa * it.stride[A] + o3 * it.stride[3]);
    tile[ty + 8 * k][tx] = v;
  }
  __syncthreads();
  float* dst = p.d.dst + (int64_t)blockIdx.y * p.d.dst_item_stride;
  int64_t ostr[4];
  ostr[3] = 1; ostr[2] = D3; ostr[1] = (int64_t)p.d.dims[2] * D3; ostr[0] = (int64_t)p.d.dims[1] * ostr[1];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int oa = ta * 32 + ty + 8 * k, o3 = t3 * 32 + tx;
    if (oa < DA && o3 < D3)
      dst[ob0 * ostr[B0] + ob1 * ostr[B1] + oa * ostr[A] + o3] = tile[tx][ty + 8 * k] * p.d.scale + p.d.bias;
  }
}

__global__ __launch_bounds__(256) void axis_resample_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            const int32_t* __restrict__ idx, const float* __restrict__ w,
                                                            const int64_t outer, const int n_in, const int n_out,
                                                            const int64_t inner, const int taps) {
  const int64_t total = outer * n_out * inner;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t in_ = i % inner;
    const int64_t r = i / inner;
    const int j = (int)(r % n_out);
    const int64_t o = r / n_out;
    const float* s = src + o * n_in * inner + in_;
    float acc = 0.f;
    for (int t = 0; t < taps; ++t) {
      const int k = idx[j * taps + t];
      if (k >= 0) acc = fmaf(w[j * taps + t], s[(int64_t)k * inner], acc);
    }
    dst[i] = acc;
  }
}

}  // namespace

extern "C" int rehr_patch_gather(const rehr_patch_gather_desc* dp, const rehr_patch_item* items, void* stream) {
  if (dp == nullptr || items == nullptr || dp->dst == nullptr) return REHR_EINVAL;
  const rehr_patch_gather_desc& d = *dp;
  if (d.n_items < 1 || (d.src_dtype != REHR_PATCH_F32 && d.src_dtype != REHR_PATCH_U8)) return REHR_EINVAL;
  int64_t total = 1;
  for (int k = 0; k < 4; ++k) {
    if (d.dims[k] < 1) return REHR_EINVAL;
    total *= d.dims[k];
  }
  if (total >= ((int64_t)1 << 40) || d.dst_item_stride < total) return REHR_EINVAL;
  for (int i = 0; i < d.n_items; ++i) {
    const rehr_patch_item& it = items[i];
    if (it.src == nullptr) return REHR_EINVAL;
    for (int k = 0; k < 4; ++k)
      if (it.lo[k] < 0 || it.hi[k] > d.dims[k]) return REHR_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < d.n_items; i0 += REHR_PATCH_MAX_ITEMS) {
    GatherParams p;
    p.d = d;
    const int n = d.n_items - i0 < REHR_PATCH_MAX_ITEMS ? d.n_items - i0 : REHR_PATCH_MAX_ITEMS;
    p.d.dst = d.dst + (int64_t)i0 * d.dst_item_stride;
    // the source-contiguous output axis, judged on the first item (all items of a launch share the kernel; a
    // mixed batch stays correct either way, the choice only decides which side is coalesced)
    int a = 3;
    for (int k = 0; k < 4; ++k) {
      const int64_t s = items[i0].stride[k] < 0 ? -items[i0].stride[k] : items[i0].stride[k];
      if (s == 1 && d.dims[k] > 1) a = k;
    }
    for (int i = 0; i < n; ++i) p.it[i] = items[i0 + i];
    if (a == 3 || d.dims[3] < 8) {
      const int64_t blocks = (total + 255) / 256;
      dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192), n);
      hipLaunchKernelGGL(patch_gather_kernel, grid, dim3(256), 0, st, p);
    } else {
      const int tiles_a = (d.dims[a] + 31) / 32, tiles_3 = (d.dims[3] + 31) / 32;
      const int b0 = a == 0 ? 1 : 0, b1 = a == 2 ? 1 : 2;
      const int64_t blocks = (int64_t)tiles_a * tiles_3 * d.dims[b0] * d.dims[b1];
      if (blocks >= ((int64_t)1 << 31)) return REHR_EINVAL;
      dim3 grid((unsigned)blocks, n);
      if (a == 0) hipLaunchKernelGGL(patch_gather_tr_kernel<0>, grid, dim3(256), 0, st, p, tiles_a, tiles_3);
      else if (a == 1) hipLaunchKernelGGL(patch_gather_tr_kernel<1>, grid, dim3(256), 0, st, p, tiles_a, tiles_3);
      else hipLaunchKernelGGL(patch_gather_tr_kernel<2>, grid, dim3(256), 0, st, p, tiles_a, tiles_3);
    }
    REHR_LAUNCH_CHECK();
  }
  return REHR_OK;
}

extern "C" int rehr_axis_resample_f32(const float* src, float* dst, const int32_t* idx, const float* w, int64_t outer,
                                      int32_t n_in, int32_t n_out, int64_t inner, int32_t taps, void* stream) {
  if (src == nullptr || dst == nullptr || idx == nullptr || w == nullptr) return REHR_EINVAL;
  if (outer < 1 || n_in < 1 || n_out < 1 || inner < 1 || taps < 1 || taps > 4096) return REHR_EINVAL;
  const int64_t total = outer * n_out * inner;
  const int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(axis_resample_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, idx, w, outer, n_in, n_out, inner, taps);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
