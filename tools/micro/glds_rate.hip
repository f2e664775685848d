// LDS-DMA (global_load_lds_dwordx4) issue-rate microbenchmark on gfx950: W waves per workgroup each issue G
// 1-KiB pieces per interval (row-gather pattern: 8 rows x 128 B per instruction, or one contiguous 1 KiB), then
// wait + barrier.  Reports cycles per interval and B/clk/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int W, int G, int GATHER, int BAR>
__global__ __launch_bounds__(64 * W) void k(const char* src, size_t row_stride, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + ((size_t)(blockIdx.x % 8) * W + wave) * (1 << 16);   // 2-8 MB footprint: L2-resident
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const char* p = GATHER ? base + (size_t)((it * G + g) % 6 * 8 + (lane >> 3)) * row_stride + (lane & 7) * 16
                             : base + (size_t)((it * G + g) % 60) * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(smem + ((wave * G + g) % 32) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

template <int W, int G, int GATHER, int BAR>
void run(const char* src, int blocks) {
  unsigned long long* out; hipMalloc(&out, blocks * 8);
  const int iters = 2000;
  k<W, G, GATHER, BAR><<<blocks, 64 * W, 64 * 1024>>>(src, 1280, 10, out);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<W, G, GATHER, BAR><<<blocks, 64 * W, 64 * 1024>>>(src, 1280, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1024]; hipMemcpy(h, out, blocks * 8, hipMemcpyDeviceToHost);
  double cyc = (double)h[0] / iters;
  double bytes = (double)W * G * 1024;
  printf("waves %2d  glds/wave %d  %s %s: %7.1f cyc/interval  %5.1f cyc/glds/CU  %5.1f B/clk/CU   chip %.1f TB/s\n", W, G,
         GATHER ? "gather" : "contig", BAR ? "barrier" : "nobar  ", cyc, cyc / (W * G), bytes / cyc,
         bytes * blocks * iters / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  char* src; size_t sz = (size_t)256 * 16 * (1 << 20);
  hipMalloc(&src, sz); hipMemset(src, 1, sz);
  const int B = 256;
  run<4, 8, 1, 1>(src, B); run<4, 4, 1, 1>(src, B); run<4, 2, 1, 1>(src, B);
  run<8, 4, 1, 1>(src, B); run<8, 2, 1, 1>(src, B); run<16, 2, 1, 1>(src, B);
  run<4, 8, 0, 1>(src, B); run<8, 4, 0, 1>(src, B);
  run<4, 8, 1, 0>(src, B); run<1, 8, 1, 0>(src, B); run<2, 8, 1, 0>(src, B);
  return 0;
}
