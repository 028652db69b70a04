// Microbenchmark: fp32 MFMA 32x32x2 throughput vs number of independent accumulator chains
// per wave and waves per SIMD (GPU box only; informs the wave-tile shapes of the kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NCHAIN>
__global__ void k(float* out, int iters) {
  f32x16 acc[NCHAIN];
  for (int c = 0; c < NCHAIN; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < NCHAIN; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < NCHAIN; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCHAIN>
void run(int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, threads = 256;  // 4 waves per block -> 1 per SIMD per block
  float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
  const int iters = 20000 / NCHAIN;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NCHAIN><<<blocks, threads>>>(out, iters); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NCHAIN><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * NCHAIN * 4096.0;
  printf("chains=%d waves/SIMD=%d: %.1f TFLOP/s (%.2f ms)\n", NCHAIN, waves_per_simd, flops / ms / 1e9, ms);
  hipFree(out);
}
int main() {
  for (int w = 1; w <= 3; ++w) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
  return 0;
}
