// Operand-path ceiling of a GEMM workgroup (diagnostic): how fast can the CUs pull L2-resident operand tiles, by path --
//   mode 0: LDS-DMA (global_load_lds_dwordx4)            mode 1: global_load_dwordx4 -> VGPR
//   mode 2: BOTH at once, half of the waves each          mode 3: both at once, every wave alternating the two
// The access pattern is the GEMM kernels' B staging: a piece = 8 rows x 128 B (one 1-KiB wave instruction), per-lane pointers
// set up ONCE and advanced by one 128-byte K-slice per piece (no division in the loop: the r03 load_paths probe recomputed
// piece -> (row group, K-slice) with two integer divisions per piece and so measured its own address arithmetic,
// ~200 cycles per piece per wave, not the memory path).
//   shared = 1: every workgroup streams the SAME matrix (weights), each starting at its own K-slice (no lockstep hot spot)
//   shared = 0: every workgroup streams its own 64 KiB matrix (activation tiles)
//   hipcc --offload-arch=gfx950 -O3 -o operand_paths operand_paths.hip && ./operand_paths
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int W, int G, int MODE>
__global__ __launch_bounds__(64 * W) void k(const char* src, int nrows, int row_bytes, int shared, int iters, unsigned long long* out,
                                            float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ksl = row_bytes / 128;
  const char* base = shared ? src : src + (size_t)blockIdx.x * nrows * row_bytes;
  // this wave's G row groups (8 rows each), all at K-slice k0; the slice advances by one per iteration and wraps
  const char* ptr[G];
  const int k0 = shared ? (int)((blockIdx.x * 7u) % (unsigned)ksl) : 0;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int rg = (wave * G + g) % (nrows / 8);
    ptr[g] = base + (size_t)(rg * 8 + (lane >> 3)) * row_bytes + (size_t)k0 * 128 + (lane & 7) * 16;
  }
  int kk = k0;
  float acc = 0.f;
  const bool dma_wave = MODE == 0 || (MODE == 2 && (wave & 1) == 0) || MODE == 3;
  const bool reg_wave = MODE == 1 || (MODE == 2 && (wave & 1) == 1) || MODE == 3;
  f32x4 ra[G], rb[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { ra[g] = f32x4{0.f, 0.f, 0.f, 0.f}; rb[g] = ra[g]; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
  auto advance = [&]() {
    ++kk;
    const int wrap = kk == ksl;
    const long step = wrap ? -(long)(ksl - 1) * 128 : 128;
    kk = wrap ? 0 : kk;
#pragma unroll
    for (int g = 0; g < G; ++g) ptr[g] += step;
  };
  for (int it = 0; it < iters; it += 2) {
    // two intervals per iteration: DMA pieces go to alternating halves of this wave's LDS slots, register pieces to ra / rb
    if (dma_wave) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        __builtin_amdgcn_global_load_lds((gptr_t)ptr[g], (lptr_t)(smem + ((wave * 2 * G + g) % 128) * 1024), 16, 0, 0);
    }
    if (reg_wave && MODE != 3) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        asm volatile("" ::"v"(ra[g]));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[g]) : "v"(ptr[g]) : "memory");
      }
    }
    advance();
    if (MODE == 3) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        asm volatile("" ::"v"(ra[g]));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[g]) : "v"(ptr[g]) : "memory");
      }
      advance();
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");            // the interval before this one has landed
    if (dma_wave && MODE != 3) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        __builtin_amdgcn_global_load_lds((gptr_t)ptr[g], (lptr_t)(smem + ((wave * 2 * G + G + g) % 128) * 1024), 16, 0, 0);
    }
    if (reg_wave && MODE != 3) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        asm volatile("" ::"v"(rb[g]));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[g]) : "v"(ptr[g]) : "memory");
      }
    }
    if (MODE != 3) {
      advance();
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int g = 0; g < G; ++g) acc += ra[g][0] + rb[g][1];
  if (lane == 0 && wave == 0) { out[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0; out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - w0; }
  if (acc == 123.456f) sink[0] = acc;
}

template <int W, int G, int MODE>
void run(const char* src, int nrows, int row_bytes, int shared, int blocks) {
  unsigned long long* out; hipMalloc(&out, blocks * 16);
  float* sink; hipMalloc(&sink, 4);
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k<W, G, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  k<W, G, MODE><<<blocks, 64 * W, 128 * 1024>>>(src, nrows, row_bytes, shared, 20, out, sink);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<W, G, MODE><<<blocks, 64 * W, 128 * 1024>>>(src, nrows, row_bytes, shared, iters, out, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), out, blocks * 16, hipMemcpyDeviceToHost);
  // pieces per wave per iteration pair: modes 0/1: 2G; mode 2: 2G (either kind); mode 3: 2G (G DMA + G register)
  const double bytes = (double)W * G * 1024 * iters;
  std::vector<double> gbs(blocks);
  double clk = 0;
  for (int b = 0; b < blocks; ++b) { gbs[b] = bytes / ((double)h[2 * b + 1] * 10.0); clk += (double)h[2 * b] / ((double)h[2 * b + 1] * 10.0); }
  std::sort(gbs.begin(), gbs.end());
  static const char* names[4] = {"LDS-DMA      ", "-> VGPR      ", "DMA|VGPR wave", "DMA+VGPR each"};
  printf("%s %2d waves x %d pieces, %3d WGs, %s %5d KB: per-CU GB/s min %6.1f med %6.1f max %6.1f  (%.2f GHz)  chip %5.2f TB/s\n", names[MODE], W, G,
         blocks, shared ? "shared" : "own   ", nrows * row_bytes / 1024, gbs.front(), gbs[blocks / 2], gbs.back(), clk / blocks,
         bytes * blocks / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(sink);
}

template <int MODE>
void sweep(const char* src, int nrows, int row_bytes, int shared, int blocks) {
  run<4, 4, MODE>(src, nrows, row_bytes, shared, blocks);
  run<4, 8, MODE>(src, nrows, row_bytes, shared, blocks);
  run<8, 4, MODE>(src, nrows, row_bytes, shared, blocks);
  run<8, 8, MODE>(src, nrows, row_bytes, shared, blocks);
  run<16, 4, MODE>(src, nrows, row_bytes, shared, blocks);
  run<16, 8, MODE>(src, nrows, row_bytes, shared, blocks);
}

int main() {
  char* src; size_t sz = (size_t)64 << 20;
  hipMalloc(&src, sz); hipMemset(src, 1, sz);
  printf("--- shared 320 x 2560 B weight matrix (0.8 MB: in every XCD's L2), 256 workgroups\n");
  sweep<0>(src, 320, 2560, 1, 256); sweep<1>(src, 320, 2560, 1, 256); sweep<2>(src, 320, 2560, 1, 256); sweep<3>(src, 320, 2560, 1, 256);
  printf("--- own 128 x 512 B tile per workgroup (64 KiB each, 16 MiB in all), 256 workgroups\n");
  sweep<0>(src, 128, 512, 0, 256); sweep<1>(src, 128, 512, 0, 256); sweep<2>(src, 128, 512, 0, 256);
  printf("--- shared 1280 x 2560 B (3.3 MB), 256 workgroups\n");
  run<8, 8, 0>(src, 1280, 2560, 1, 256); run<8, 8, 1>(src, 1280, 2560, 1, 256); run<8, 8, 2>(src, 1280, 2560, 1, 256);
  run<16, 8, 0>(src, 1280, 2560, 1, 256); run<16, 8, 1>(src, 1280, 2560, 1, 256); run<16, 8, 2>(src, 1280, 2560, 1, 256);
  return 0;
}
