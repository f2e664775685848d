// Microbenchmark (diagnostic, not part of the product): how much VALU / transcendental / LDS-read issue hides under
// v_mfma_f32_32x32x16_f16 on one SIMD, with 1 / 2 / 4 waves per SIMD, and what the shader clock is under that load.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int M, int E, int P, int L>
__global__ void probe(float* out, unsigned long long* stamps, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((float*)lds)[i] = (float)i;
  __syncthreads();
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
  f32x16 acc0 = {}, acc1 = {};
  float x[8];
  for (int e = 0; e < 8; ++e) x[e] = -0.5f - 0.01f * (lane + e);
  float y[8];
  for (int e = 0; e < 8; ++e) y[e] = 0.25f * (lane + e);
  f32x4 ld[2] = {};
  const unsigned lp = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds + lane * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();        // constant 100 MHz
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();            // shader clock
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (M) {
        if (m == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int l = 0; l < L; ++l) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[l & 1]) : "v"(lp), "n"(1024 * l));
#pragma unroll
      for (int e = 0; e < E; ++e) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(m * E + e) & 7]));
#pragma unroll
      for (int e = 0; e < P; ++e) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(y[(m * P + e) & 7]) : "v"(y[(e + 3) & 7]), "v"(y[(e + 5) & 7]));
      if (L) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
  for (int e = 0; e < 8; ++e) s += x[e] + y[e];
  s += ld[0][0] + ld[1][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = c1 - c0; }
}

template <int M, int E, int P, int L>
void run(const char* name) {
  float* out; unsigned long long* st;
  hipMalloc(&out, 256 * 1024 * 4 * sizeof(float)); hipMalloc(&st, 16);
  const int iters = 2000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    hipLaunchKernelGGL((probe<M, E, P, L>), dim3(256), dim3(256 * wps), 0, 0, out, st, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<M, E, P, L>), dim3(256), dim3(256 * wps), 0, 0, out, st, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    const double ns = h[0] * 10.0, cyc = (double)h[1];
    // per SIMD and per (MFMA + its fillers) slot: wps waves share the SIMD
    printf("%-28s waves/SIMD %d: %7.1f shader cyc/slot/wave  %6.1f cyc/slot/SIMD  clock %.2f GHz  kernel %.1f us\n", name, wps,
           cyc / (2.0 * iters), cyc / (2.0 * iters) / wps, cyc / ns, ms * 1e3);
  }
  hipFree(out); hipFree(st);
}

int main() {
  run<1, 0, 0, 0>("mfma only");
  run<0, 2, 0, 0>("2 exp only");
  run<0, 0, 5, 0>("5 max3 only");
  run<1, 1, 0, 0>("mfma + 1 exp");
  run<1, 2, 0, 0>("mfma + 2 exp");
  run<1, 3, 0, 0>("mfma + 3 exp");
  run<1, 4, 0, 0>("mfma + 4 exp");
  run<1, 0, 3, 0>("mfma + 3 max3");
  run<1, 0, 5, 0>("mfma + 5 max3");
  run<1, 0, 8, 0>("mfma + 8 max3");
  run<1, 2, 3, 0>("mfma + 2 exp + 3 max3");
  run<1, 2, 3, 1>("mfma + 2 exp + 3 max3 + 1 lds");
  run<1, 0, 0, 1>("mfma + 1 lds");
  run<1, 0, 0, 2>("mfma + 2 lds");
  run<0, 0, 0, 1>("1 lds only");
  return 0;
}
