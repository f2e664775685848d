// Gate of the round-5 plan: what does ONE grid-wide barrier inside a persistent launch cost on MI355X,
// against the 1.1 - 1.9 us of the kernel boundary it would replace?
//
//   grid  = one workgroup per CU (256) -- checked on the host against hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs; when it
//           does not hold the program returns EINVAL (22) WITHOUT launching.  Every spin is bounded: a workgroup that waits longer
//           than kSpinMax polls sets the error word and leaves, so a grid that is not co-resident ends with err != 0, never hangs.
//   forms   A  flat: one monotonic counter; every workgroup: stores drained (vmcnt 0) -> workgroup barrier -> lane 0 agent-scope
//              release (buffer_wbl2 sc1) -> asm vmcnt(0) -> relaxed agent atomic add -> relaxed sc1 poll with s_sleep -> agent-scope
//              acquire (buffer_inv sc1) -> vmcnt(0) -> workgroup barrier
//           B  hierarchical by the PHYSICAL XCC id (s_getreg HW_REG_XCC_ID; group sizes counted in the kernel's first, flat
//              barrier): arrive at the XCD's counter; the XCD's last arriver does the ONE release for that L2, then the top
//              counter; the last of the 8 writes one generation word per XCD; everybody polls its XCD's word, then acquires
//           C  B without any fence (counter traffic only): the floor if every handed-off byte travelled sc1 / write-through --
//              NOT a valid barrier for plain loads and stores; its stale-read count is reported to show exactly that
//   payload  every phase each workgroup writes a 128-byte record (plain stores) and, behind the barrier, checks the records of
//            three other workgroups (same XCD label, next label, far away) that it PRE-READ with plain loads in front of the
//            barrier (an L1-warm consumer: the case that exposes a missing acquire); odd workgroups optionally spin ~2 us first
//            (uneven arrival)
//   timing   in-kernel: s_memrealtime (100 MHz) of workgroup 0 around the NB barrier phases; host: HIP events around the launch,
//            NB phases against 0 phases
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/grid_barrier tools/micro/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr unsigned kSpinMax = 1u << 21;     // x (s_sleep 1 + one L2-served load) ~ 0.3 s: a lost workgroup costs a run, not the box
constexpr int kLine = 32;                   // words between two polled / added words: one 128-byte line each
// state words (all zeroed by a hipMemsetAsync in front of every launch):
//   [0]            flat counter
//   [kLine*(1+x)]  XCD x arrival counter          x = 0..7
//   [kLine*9]      top counter
//   [kLine*(10+x)] XCD x generation word
//   [kLine*18]     error word (a spin timed out: the barrier phase it happened in, + 1)
//   [kLine*(19+x)] XCD x population (census, first barrier)
//   [kLine*27]     stale-read count of the payload check
constexpr int kStateWords = kLine * 28;

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7;
}
__device__ __forceinline__ unsigned ld(unsigned* p) { return __hip_atomic_load((gu32*)p, RLX_AGENT); }
__device__ __forceinline__ void st(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, RLX_AGENT); }
__device__ __forceinline__ unsigned add(unsigned* p, unsigned v) { return __hip_atomic_fetch_add((gu32*)p, v, RLX_AGENT); }

// one lane waits until *p >= want; false (and the error word set) when the bound is hit
__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned want, unsigned* err, unsigned code) {
  for (unsigned spins = 0; spins < kSpinMax; ++spins) {
    if (ld(p) >= want) return true;
    if (ld(err) != 0) return false;               // somebody else gave up: leave too
    __builtin_amdgcn_s_sleep(1);
  }
  st(err, code);
  return false;
}

template <int FORM>   // 0 = A flat, 1 = B hierarchical with fences, 2 = C hierarchical without fences
__device__ __forceinline__ bool grid_barrier(unsigned* s, unsigned k /* 1-based phase */, unsigned nwg, unsigned xcc, unsigned xcc_pop,
                                             unsigned n_xcc) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains its stores ...
  __syncthreads();                                          // ... before the one lane that signals for them
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    bool ok = true;
    unsigned* err = s + kLine * 18;
    if (FORM == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (ROCm 7.2 may drop the fence's own wait: always an asm one behind it)
      add(s, 1u);
      ok = wait_ge(s, k * nwg, err, k);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else {
      const unsigned old = add(s + kLine * (1 + xcc), 1u);
      if (old + 1 == k * xcc_pop) {                          // this XCD's last arriver: every store of the XCD has reached its L2
        if (FORM == 1) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // ONE write-back of that L2
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned oldt = add(s + kLine * 9, 1u);
        if (oldt + 1 == k * n_xcc)
          for (unsigned x = 0; x < 8; ++x) st(s + kLine * (10 + x), k);
      }
      ok = wait_ge(s + kLine * (10 + xcc), k, err, k);
      if (FORM == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // holds the workgroup barrier until the invalidate has completed
    ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}

// FORM 3 = no barrier at all (the payload traffic alone)
template <int FORM>
__global__ __launch_bounds__(256) void persist(unsigned* s, unsigned* rec /* [2][nwg][32] */, int nb, int uneven, unsigned long long* ticks,
                                               unsigned* where /* [nwg]: physical XCC of every workgroup */) {
  const unsigned nwg = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
  const unsigned xcc = xcc_id();
  // census (and the co-residency proof): a FLAT barrier on word 0; afterwards every XCD's population is known
  __shared__ unsigned pop_s[8];
  __shared__ int ok0;
  if (tid == 0) {
    where[b] = xcc;
    add(s + kLine * (19 + xcc), 1u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    add(s, 1u);
    ok0 = wait_ge(s, nwg, s + kLine * 18, 0x80000000u);
    for (int x = 0; x < 8; ++x) pop_s[x] = ld(s + kLine * (19 + x));
  }
  __syncthreads();
  if (!ok0) return;
  unsigned n_xcc = 0;
  for (int x = 0; x < 8; ++x) n_xcc += pop_s[x] != 0;
  const unsigned pop = pop_s[xcc];
  const unsigned peers[3] = {(b + 8) % nwg, (b + 1) % nwg, (b + nwg / 2 + 3) % nwg};
  unsigned stale = 0;
  unsigned long long t0 = 0;
  if (b == 0 && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 1; it <= nb; ++it) {
    unsigned* mine = rec + ((size_t)(it & 1) * nwg + b) * 32;
    if (tid < 32) mine[tid] = (unsigned)it * 1000003u + b * 31u + tid;            // plain stores
    // L1-warm consumer: plain pre-reads of the lines this workgroup will check behind the barrier (they hold phase it-2's values)
    unsigned pre = 0;
    if (tid < 96) pre = rec[((size_t)(it & 1) * nwg + peers[tid >> 5]) * 32 + (tid & 31)];
    if (pre == 0xdeadbeefu) stale += 1u << 20;                                   // (keeps the pre-read alive)
    if (uneven && (b & 1)) {                                                     // odd workgroups arrive ~2 us late
      const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - w0 < 200) __builtin_amdgcn_s_sleep(2);
    }
    if (FORM == 0) { if (!grid_barrier<0>(s, (unsigned)it + 1u, nwg, xcc, pop, n_xcc)) return; }   // (+1: the census used phase 1 of word 0)
    if (FORM == 1) { if (!grid_barrier<1>(s, (unsigned)it, nwg, xcc, pop, n_xcc)) return; }
    if (FORM == 2) { if (!grid_barrier<2>(s, (unsigned)it, nwg, xcc, pop, n_xcc)) return; }
    if (tid < 96) {
      const unsigned p = peers[tid >> 5];
      const unsigned got = rec[((size_t)(it & 1) * nwg + p) * 32 + (tid & 31)];   // plain loads
      stale += got != (unsigned)it * 1000003u + p * 31u + (tid & 31);
    }
    if (FORM == 3) __syncthreads();
  }
  if (b == 0 && tid == 0) { ticks[0] = __builtin_amdgcn_s_memrealtime() - t0; }
  if (stale & 0xfffff) atomicAdd(s + kLine * 27, stale & 0xfffff);
}

template <int FORM>
static int run(const char* name, int nwg, int nb, int uneven, unsigned* d_state, unsigned* d_rec, unsigned long long* d_ticks, unsigned* d_where,
               int cus, bool print_census) {
  int occ = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persist<FORM>, 256, 0);
  if (e != hipSuccess || occ < 1 || nwg > occ * cus) {
    fprintf(stderr, "grid_barrier: %d workgroups do not fit %d CUs x %d resident blocks (%s): NOT launched\n", nwg, cus, occ, hipGetErrorString(e));
    return -EINVAL;
  }
  // margin for the occupancy API's known optimism (SGPR-heavy kernels admit one block per CU fewer than it says): the grid
  // must also fit with ONE block per CU fewer whenever it needs more than one
  if (nwg > cus && nwg > (occ - 1) * cus) {
    fprintf(stderr, "grid_barrier: %d workgroups need all %d blocks per CU the API admits: no margin, NOT launched\n", nwg, occ);
    return -EINVAL;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best[2] = {1e9f, 1e9f};
  unsigned long long tick_best = ~0ull;
  unsigned err = 0, stale_tot = 0;
  for (int rep = 0; rep < 6; ++rep) {
    for (int which = 0; which < 2; ++which) {                 // 0 phases, then nb phases: the difference is the phases' cost
      const int n = which ? nb : 0;
      hipMemsetAsync(d_state, 0, kStateWords * 4, 0);
      hipMemsetAsync(d_rec, 0, (size_t)2 * nwg * 32 * 4, 0);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(persist<FORM>, dim3(nwg), dim3(256), 0, 0, d_state, d_rec, n, uneven, d_ticks, d_where);
      hipEventRecord(e1, 0);
      if (hipEventSynchronize(e1) != hipSuccess) { fprintf(stderr, "launch failed\n"); return -EIO; }
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best[which]) best[which] = ms;
      std::vector<unsigned> st(kStateWords);
      hipMemcpy(st.data(), d_state, kStateWords * 4, hipMemcpyDeviceToHost);
      err |= st[kLine * 18];
      if (which) {
        stale_tot += st[kLine * 27];
        unsigned long long t = 0;
        hipMemcpy(&t, d_ticks, 8, hipMemcpyDeviceToHost);
        if (rep > 0 && t < tick_best) tick_best = t;
        if (print_census && rep == 0) {
          printf("  census: workgroups per physical XCC:");
          for (int x = 0; x < 8; ++x) printf(" %u", st[kLine * (19 + x)]);
          std::vector<unsigned> wh(nwg);
          hipMemcpy(wh.data(), d_where, nwg * 4, hipMemcpyDeviceToHost);
          int same = 1;
          for (int i = 8; i < nwg; ++i) same &= wh[i] == wh[i - 8];
          printf("   (blockIdx %% 8 classes share an XCC: %s)\n", same ? "yes" : "NO");
        }
      }
    }
  }
  printf("  %-44s %4d WG  uneven=%d  in-kernel %6.2f us/phase   host-paired %6.2f us/phase   timeouts=%s  stale reads=%u of %d\n", name, nwg, uneven,
         tick_best * 0.01 / nb, (best[1] - best[0]) * 1e3 / nb, err ? "YES" : "none", stale_tot, 6 * nb * nwg * 96);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return err ? -ETIMEDOUT : 0;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no GPU\n"); return 1; }
  const int cus = prop.multiProcessorCount;
  const int nb = argc > 1 ? atoi(argv[1]) : 200;
  printf("%s, %d CUs; %d barrier phases per launch, best of 5 launches; payload = one 128-byte record per workgroup and phase\n", prop.gcnArchName, cus, nb);
  unsigned *d_state, *d_rec, *d_where;
  unsigned long long* d_ticks;
  hipMalloc(&d_state, kStateWords * 4);
  hipMalloc(&d_rec, (size_t)2 * 1024 * 32 * 4);
  hipMalloc(&d_where, 1024 * 4);
  hipMalloc(&d_ticks, 64);
  int rc = 0;
  for (int nwg : {cus, 2 * cus}) {
    for (int uneven : {0, 1}) {
      rc |= run<3>("no barrier (payload traffic + __syncthreads)", nwg, nb, uneven, d_state, d_rec, d_ticks, d_where, cus, uneven == 0);
      rc |= run<0>("A flat counter, release + acquire per WG", nwg, nb, uneven, d_state, d_rec, d_ticks, d_where, cus, false);
      rc |= run<1>("B per-XCD counters, one release per XCD", nwg, nb, uneven, d_state, d_rec, d_ticks, d_where, cus, false);
      rc |= run<2>("C per-XCD counters, NO fences (invalid)", nwg, nb, uneven, d_state, d_rec, d_ticks, d_where, cus, false);
    }
  }
  // the refusal path: a grid that cannot be co-resident is not launched
  const int too_many = 64 * cus;
  const int r = run<1>("B oversize grid (must be refused)", too_many, 4, 0, d_state, d_rec, d_ticks, d_where, cus, false);
  printf("  oversize grid of %d workgroups: %s\n", too_many, r == -EINVAL ? "refused with EINVAL, nothing launched" : "NOT refused");
  return (rc == 0 && r == -EINVAL) ? 0 : 2;
}
