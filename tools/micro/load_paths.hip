// Per-CU streaming rate of an L2-resident weight matrix by load path (diagnostic): LDS-DMA (global_load_lds_dwordx4) vs
// global_load_dwordx4 into registers vs global_load_dwordx4 + ds_write_b128.  Every workgroup streams the SAME matrix
// (rows x 128-byte K-slices, like the weight operand of the GEMM kernels), W waves per workgroup, G 1-KiB pieces per
// wave and interval with up to 2 G pieces in flight.
//   hipcc --offload-arch=gfx950 -O3 -o load_paths load_paths.hip && ./load_paths
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: LDS-DMA   1: global_load -> VGPR   2: global_load -> VGPR -> ds_write_b128
template <int W, int G, int MODE>
__global__ __launch_bounds__(64 * W) void k(const char* src, int nrows, int row_bytes, int iters, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ksl = row_bytes / 128;                         // 128-byte K-slices per row
  const int row_groups = nrows / 8;                        // one piece = 8 rows x 128 B
  float acc = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
  int piece = wave;                                        // pieces are dealt round-robin to the waves
  auto src_of = [&]() {
    const int rg = piece % row_groups, ks = (piece / row_groups) % ksl;
    piece += W;
    return src + (size_t)(rg * 8 + (lane >> 3)) * row_bytes + ks * 128 + (lane & 7) * 16;
  };
  if constexpr (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        __builtin_amdgcn_global_load_lds((gptr_t)src_of(), (lptr_t)(smem + ((wave * 2 * G + (it & 1) * G + g) % 64) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");     // the previous interval's pieces have landed
    }
  } else {
    // compiler-visible loads (it counts vmcnt itself): two register sets, one in flight while the other is consumed
    f32x4 ra[G], rb[G];
    auto issue = [&](f32x4 (&r)[G]) {
#pragma unroll
      for (int g = 0; g < G; ++g) r[g] = __builtin_nontemporal_load((const f32x4*)src_of());
    };
    auto consume = [&](f32x4 (&r)[G]) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (MODE == 2) *(f32x4*)(smem + ((wave * G + g) % 64) * 1024 + lane * 16) = r[g];
        else asm volatile("" ::"v"(r[g]));
      }
    };
    issue(ra);
    for (int it = 0; it < iters; it += 2) {
      issue(rb); consume(ra);
      issue(ra); consume(rb);
    }
    consume(ra);
    acc += ra[0][0];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0; out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - w0; }
  if (acc == 123.456f) sink[0] = acc;
}

template <int W, int G, int MODE>
void run(const char* src, int nrows, int row_bytes, int blocks) {
  unsigned long long* out; hipMalloc(&out, blocks * 16);
  float* sink; hipMalloc(&sink, 4);
  const int iters = 1500;
  hipFuncSetAttribute((const void*)k<W, G, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  k<W, G, MODE><<<blocks, 64 * W, 64 * 1024>>>(src, nrows, row_bytes, 10, out, sink);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<W, G, MODE><<<blocks, 64 * W, 64 * 1024>>>(src, nrows, row_bytes, iters, out, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
  const double bytes = (double)W * G * 1024 * iters;
  static const char* names[3] = {"LDS-DMA            ", "global_load -> VGPR", "global_load+ds_write"};
  printf("%s waves %2d x %d pieces, %3d WGs, matrix %4d KB: %5.1f B/clk/CU  %6.1f GB/s/CU (clock %.2f GHz)  chip %5.2f TB/s\n", names[MODE], W, G, blocks,
         nrows * row_bytes / 1024, bytes / (double)h[0], bytes / ((double)h[1] * 10.0), (double)h[0] / ((double)h[1] * 10.0), bytes * blocks / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(sink);
}

template <int MODE>
void sweep(const char* src, int nrows, int row_bytes, int blocks) {
  run<1, 8, MODE>(src, nrows, row_bytes, blocks);
  run<2, 8, MODE>(src, nrows, row_bytes, blocks);
  run<4, 4, MODE>(src, nrows, row_bytes, blocks);
  run<4, 8, MODE>(src, nrows, row_bytes, blocks);
  run<8, 4, MODE>(src, nrows, row_bytes, blocks);
  run<8, 8, MODE>(src, nrows, row_bytes, blocks);
}

int main() {
  char* src; size_t sz = (size_t)64 << 20;
  hipMalloc(&src, sz); hipMemset(src, 1, sz);
  // the back-to-back kernel's weights: 320 rows x 1280 fp16 (819 KB), streamed by 256 workgroups
  printf("--- 320 x 2560 B (0.8 MB, L2-resident), 256 workgroups\n");
  sweep<0>(src, 320, 2560, 256); sweep<1>(src, 320, 2560, 256);   // (mode 2's stores are dead code to the compiler: not swept)
  printf("--- 1280 x 23040 B (29.5 MB: beyond L2, inside the MALL), 256 workgroups\n");
  sweep<0>(src, 1280, 23040, 256); sweep<1>(src, 1280, 23040, 256);
  printf("--- 320 x 2560 B, 128 workgroups\n");
  run<4, 8, 0>(src, 320, 2560, 128); run<8, 8, 0>(src, 320, 2560, 128); run<4, 8, 1>(src, 320, 2560, 128); run<8, 8, 1>(src, 320, 2560, 128);
  return 0;
}
