// Microbenchmark (diagnostic, not part of the product): the floor of a one-pass elementwise kernel in the UNet's chain -- read an fp32
// tensor that ANOTHER kernel has just written, write its fp16 copy -- with gn_apply_kernel's launch shape (512 threads, two 8-channel
// items per thread, every load of a workgroup in flight at once).  What GroupNorm-apply could reach if statistics and arithmetic were free.
//   hipcc --offload-arch=gfx950 -O3 -o stream_floor stream_floor.hip && ./stream_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void produce(float* x, size_t n4, float v) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) ((f32x4*)x)[i] = f32x4{v, v + 1.f, v + 2.f, v + 3.f};
}

__global__ __launch_bounds__(512, 4) void consume(const float* x, _Float16* y, size_t items) {
  f32x4 a[2], b[2];
  size_t it[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    it[k] = (size_t)blockIdx.x * 1024 + threadIdx.x + k * 512;
    if (it[k] < items) { a[k] = *(const f32x4*)(x + it[k] * 8); b[k] = *(const f32x4*)(x + it[k] * 8 + 4); }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (it[k] < items) {
      f16x8 o;
      for (int e = 0; e < 4; ++e) { o[e] = (_Float16)(a[k][e] * 1.5f + 0.25f); o[4 + e] = (_Float16)(b[k][e] * 1.5f + 0.25f); }
      *(f16x8*)(y + it[k] * 8) = o;
    }
}

int main() {
  const size_t sizes[] = {2 * 1024 * 320, 2 * 4096 * 320, 2 * 1024 * 640, 2 * 4096 * 640, 2 * 4096 * 960};
  const char* names[] = {"C=320 P=1024 B=2", "C=320 P=4096 B=2", "C=640 P=1024 B=2", "C=640 P=4096 B=2", "C=960 P=4096 B=2"};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int s = 0; s < 5; ++s) {
    const size_t n = sizes[s], items = n / 8;
    float* x; _Float16* y;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 2);
    const int cblocks = (int)((items + 1023) / 1024);
    float t_pair, t_prod, t_cons;
    const int reps = 50;
    for (int w = 0; w < 3; ++w) { hipLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, 0, x, n / 4, 1.f); hipLaunchKernelGGL(consume, dim3(cblocks), dim3(512), 0, 0, x, y, items); }
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, 0, x, n / 4, (float)r); hipLaunchKernelGGL(consume, dim3(cblocks), dim3(512), 0, 0, x, y, items); }
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&t_pair, e0, e1);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, 0, x, n / 4, (float)r);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&t_prod, e0, e1);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(consume, dim3(cblocks), dim3(512), 0, 0, x, y, items);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&t_cons, e0, e1);
    const double mb = n * 6.0 / 1e6;
    const double us_cold = (t_pair - t_prod) * 1e3 / reps, us_warm = t_cons * 1e3 / reps;
    printf("%-18s %6.1f MB moved, %4d workgroups: behind a producer %6.2f us (%.2f TB/s)   back to back on the same input %6.2f us (%.2f TB/s)   [producer alone %6.2f us]\n",
           names[s], mb, cblocks, us_cold, mb / us_cold, us_warm, mb / us_warm, t_prod * 1e3 / reps);
    hipFree(x); hipFree(y);
  }
  return 0;
}
