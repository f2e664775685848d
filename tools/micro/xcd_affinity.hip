// Cross-kernel L2 affinity on MI355X (8 XCDs, private L2s): kernel A writes a buffer, kernel B reads it.
//  (1) Is "blockIdx % 8 -> physical XCD" stable from launch to launch (also behind launches whose grid is not a multiple of 8)?
//  (2) How much faster does B read a slice that the SAME physical XCD wrote in A than a slice another XCD wrote?
// Slices are assigned by the PHYSICAL XCC id (s_getreg HW_REG_XCC_ID) through a per-XCD ticket counter, so every slice is
// processed exactly once whatever the placement.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7;
}

__global__ void who(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

// rotate: the slice group read by physical XCD x is the one written by XCD (x + rotate) % 8
__global__ __launch_bounds__(256) void writer(float4* buf, size_t slice_elems, int slices_per_xcd, unsigned* heads, unsigned* owner) {
  __shared__ unsigned s_slice;
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id();
    unsigned got = 0xffffffffu;
    for (int k = 0; k < 8 && got == 0xffffffffu; ++k) {          // own XCD's range first, then steal
      const unsigned xx = (x + k) & 7;
      const unsigned t = atomicAdd(&heads[xx], 1u);
      if (t < (unsigned)slices_per_xcd) got = xx * slices_per_xcd + t;
    }
    s_slice = got;
    if (got != 0xffffffffu) owner[got] = x;
  }
  __syncthreads();
  if (s_slice == 0xffffffffu) return;
  float4* p = buf + (size_t)s_slice * slice_elems;
  for (size_t i = threadIdx.x; i < slice_elems; i += 256) p[i] = float4{1.f, 2.f, 3.f, (float)s_slice};
}

__global__ __launch_bounds__(256) void reader(const float4* buf, size_t slice_elems, int slices_per_xcd, unsigned* heads, int rotate, float* sink,
                                              unsigned long long* cyc) {
  __shared__ unsigned s_slice;
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id();
    unsigned got = 0xffffffffu;
    for (int k = 0; k < 8 && got == 0xffffffffu; ++k) {
      const unsigned xx = (x + k) & 7;
      const unsigned t = atomicAdd(&heads[xx], 1u);
      if (t < (unsigned)slices_per_xcd) got = ((xx + rotate) & 7) * slices_per_xcd + t;
    }
    s_slice = got;
  }
  __syncthreads();
  if (s_slice == 0xffffffffu) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const float4* p = buf + (size_t)s_slice * slice_elems;
  float acc = 0.f;
  for (size_t i = threadIdx.x; i < slice_elems; i += 256) { const float4 v = p[i]; acc += v.x + v.w; }
  if (acc == 12345.f) sink[0] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

int main() {
  // (1) label stability
  unsigned* d_who; hipMalloc(&d_who, 4096 * 4);
  std::vector<unsigned> h(4096);
  const int grids[] = {256, 256, 257, 256, 13, 256, 1021, 256, 64, 256};
  for (int g : grids) {
    who<<<g, 64>>>(d_who);
    hipMemcpy(h.data(), d_who, g * 4, hipMemcpyDeviceToHost);
    int consistent = 1;
    for (int b = 8; b < g; ++b) consistent &= h[b] == h[b - 8];
    printf("grid %4d: block 0..7 on XCC %u %u %u %u %u %u %u %u   b%%8 classes share an XCC: %s\n", g, h[0], h[1], h[2], h[3], h[4], h[5],
           h[6], h[7], consistent ? "yes" : "NO");
  }
  // (2) affinity: total MB in {8, 32, 64}, 256 slices
  for (size_t mb : {8, 24, 64}) {
    const int slices = 256, spx = slices / 8;
    const size_t slice_elems = mb * (1 << 20) / slices / 16;
    float4* buf; hipMalloc(&buf, slices * slice_elems * 16);
    unsigned *heads, *owner; hipMalloc(&heads, 64); hipMalloc(&owner, slices * 4);
    float* sink; hipMalloc(&sink, 4);
    unsigned long long* cyc; hipMalloc(&cyc, 512 * 8);
    char* thrash; hipMalloc(&thrash, 64 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rotate : {0, 1, 4, 0, 1}) {
      float best = 1e9f, sum = 0.f;
      const int reps = 8;
      for (int r = 0; r < reps; ++r) {
        hipMemset(thrash, r, 64 << 20);                       // evict the L2s
        hipMemset(heads, 0, 64);
        writer<<<256, 256>>>(buf, slice_elems, spx, heads, owner);
        hipMemset(heads, 0, 64);
        hipEventRecord(e0);
        reader<<<256, 256>>>(buf, slice_elems, spx, heads, rotate, sink, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
      }
      std::vector<unsigned> ow(slices);
      hipMemcpy(ow.data(), owner, slices * 4, hipMemcpyDeviceToHost);
      int stolen = 0;
      for (int s2 = 0; s2 < slices; ++s2) stolen += ow[s2] != (unsigned)(s2 / spx);
      std::vector<unsigned long long> hc(256);
      hipMemcpy(hc.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
      unsigned long long med[256]; for (int i = 0; i < 256; ++i) med[i] = hc[i];
      std::sort(med, med + 256);
      printf("%3zu MB, reader takes the slices written by XCD+%d: %7.2f us best, %7.2f us mean (%.2f TB/s), in-kernel read median %llu cycles; %d slices stolen by another XCD\n",
             mb, rotate, best * 1e3f, sum / reps * 1e3f, mb * 1048576.0 / (best * 1e-3) / 1e12, med[128], stolen);
    }
    hipFree(buf); hipFree(heads); hipFree(owner); hipFree(sink); hipFree(cyc); hipFree(thrash);
  }
  return 0;
}
