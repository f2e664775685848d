// Raw MFMA issue-rate microbenchmark (gfx950): register-only loops, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float seed) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed + threadIdx.x * 0.001f + e); b[e] = (_Float16)(seed * 0.5f - e); }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float seed) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed + threadIdx.x * 0.001f + e); b[e] = (_Float16)(seed * 0.5f - e); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class K>
void run(const char* name, K kern, int blocks, int iters, int nacc, double flop_per_mfma) {
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, 256>>>(out, 10, 1.0f);
  hipEventRecord(e0);
  kern<<<blocks, 256>>>(out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mfmas = (double)blocks * 4 * iters * nacc;
  double tf = mfmas * flop_per_mfma / (ms * 1e-3) / 1e12;
  // cycles per MFMA per SIMD at 2.4 GHz nominal, assuming blocks spread over 256 CUs x 4 SIMDs
  double waves_per_simd = blocks * 4 / 1024.0;
  double cyc = ms * 1e-3 * 2.4e9 / (iters * nacc * (waves_per_simd < 1 ? 1 : waves_per_simd));
  printf("%-28s blocks %4d: %8.3f ms  %7.1f TFLOP/s  ~%.1f cyc/MFMA/SIMD @2.4GHz\n", name, blocks, ms, tf, cyc);
  hipFree(out);
}
int main() {
  const int it = 20000;
  run("32x32x16 f16, 4 acc", k32<4>, 256, it, 4, 32768.0);
  run("32x32x16 f16, 4 acc", k32<4>, 512, it, 4, 32768.0);
  run("32x32x16 f16, 1 acc", k32<1>, 256, it, 1, 32768.0);
  run("32x32x16 f16, 4 acc (192 WG)", k32<4>, 192, it, 4, 32768.0);
  run("16x16x32 f16, 4 acc", k16<4>, 256, it, 4, 16384.0);
  run("16x16x32 f16, 8 acc", k16<8>, 512, it, 8, 16384.0);
  return 0;
}
