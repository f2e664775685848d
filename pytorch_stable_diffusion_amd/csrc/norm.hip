// GroupNorm(32) and LayerNorm over NHWC (token-major) activations for gfx950.
//
// GroupNorm (reference call sites sd/diffusion.py:173,199,294,733): two launches,
//   gn_stats : per (image, pixel-chunk) partial {sum, sum of squares} per group, fp32, from one or
//              two concat sources (the skip concat of sd/diffusion.py:671 is never materialised);
//   gn_apply : reduces the partials, y = (x - mean) * rstd * gamma + beta, optional SiLU, fp16 out
//              (this is the A operand of the following conv / 1x1 conv).
// LayerNorm (sd/diffusion.py:317,334,351): one wave per token row, exact two-pass statistics held
// in registers, fp16 out.  Statistics are fp32; biased variance like torch.
#include "common.h"

#ifdef SDMI_GNA_PROBE
// diagnostic build (tools/build_variant.sh probe norm -DSDMI_GNA_PROBE): shader-clock stamps of thread 0 of the first 2048 workgroups of
// gn_apply_kernel: {loads requested, records summed, mean / rstd ready, table ready, stores issued} since the workgroup's start
__device__ unsigned long long g_gna_clk[2048][6];
extern "C" int sdmi_dbg_read_gna(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gna_clk), (size_t)n * 48) == hipSuccess ? 0 : -5;
}
#define GNA_STAMP(i) do { if (tid == 0) gna_t[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GNA_STAMP(i) do { } while (0)
#endif

namespace {

__device__ __forceinline__ void load8(const void* base, int is_f32, size_t elem_off, float (&v)[8]) {
  if (is_f32) {
    const f32x4 a = *(const f32x4*)((const float*)base + elem_off);
    const f32x4 b = *(const f32x4*)((const float*)base + elem_off + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
  } else {
    const f16x8 a = *(const f16x8*)((const f16*)base + elem_off);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
  }
}

// grid: (nchunk, B); block: (C/8) * PY threads (<= 320), thread = (pixel lane py, channel chunk c8).
// Deterministic: per-thread partials go to LDS and each group is summed in a fixed order.
__global__ __launch_bounds__(320) void gn_stats_kernel(GnArgs p, int PY, int pix_per_chunk) {
  sdmi_kernarg_warm<sizeof(GnArgs) + 8>();     // one miss latency for the argument block instead of one per line the compiler reaches for (common.h)
  __shared__ float s_part[320 * 4];     // per thread: {sum0, sq0, sum1, sq1} (first / second group of its chunk)
  const int C = p.C0 + p.C1, C8 = C / 8, cpg = C / 32;
  const int tid = threadIdx.x;
  const int c8 = tid % C8, py = tid / C8;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int c = c8 * 8;
  const bool second = c >= p.C0;
  const void* base = second ? p.x1 : p.x0;
  const int cs = second ? p.C1 : p.C0;
  const int cc = second ? c - p.C0 : c;
  float sum[8], sq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sum[e] = 0.f; sq[e] = 0.f; }
  const int p0 = chunk * pix_per_chunk;
  const int p1 = min(p0 + pix_per_chunk, p.P);
#pragma unroll 4
  for (int px = p0 + py; px < p1; px += PY) {
    float v[8];
    load8(base, p.in_f32, ((size_t)n * p.P + px) * cs + cc, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { sum[e] += v[e]; sq[e] += v[e] * v[e]; }
  }
  const int g0 = c / cpg;
  float a0 = 0.f, q0 = 0.f, a1 = 0.f, q1 = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (c - g0 * cpg + e < cpg) { a0 += sum[e]; q0 += sq[e]; } else { a1 += sum[e]; q1 += sq[e]; }
  }
  s_part[tid * 4 + 0] = a0; s_part[tid * 4 + 1] = q0; s_part[tid * 4 + 2] = a1; s_part[tid * 4 + 3] = q1;
  __syncthreads();
  if (tid < 32) {
    const int g = tid;
    const int lo = (g * cpg) / 8, hi = ((g + 1) * cpg - 1) / 8;
    double s = 0.0, q = 0.0;       // fp64 above the per-thread level: the variance is E[x^2] - E[x]^2 (see gn_apply_kernel)
    for (int k = lo; k <= hi; ++k) {
      const int kg0 = (k * 8) / cpg;
      const int sel = (kg0 == g) ? 0 : 2;       // chunk k contributes its first or its second part to g
      for (int y = 0; y < PY; ++y) {
        const float* e = s_part + (y * C8 + k) * 4 + sel;
        s += e[0];
        q += e[1];
      }
    }
    float* o = p.partial + (((size_t)n * p.nchunk + chunk) * 32 + g) * 2;
    o[0] = (float)s;
    o[1] = (float)q;
  }
}

// grid: (pixel blocks, B); block 512.  Phase A: every thread requests its activations (and gamma / beta); statistics: all threads
// add up the chunk partials of gn_stats_kernel -- or the records the producers' epilogues left behind (GnRec) -- in a fixed
// order to mean / rstd per group; phase B: normalise (+SiLU) 8-channel chunks.
// 512 threads x 2 items, not 256 x 4 (round 4): with the records and the gamma / beta prefetch the 256-thread form needed 189
// registers per lane -- two workgroups per CU, two rounds of workgroups on the 64x64 maps; this one stays under 128 at 16
// statistics loads in flight per thread.
// Items (8-channel chunks) per thread: 2, or 4 when two would need more workgroups than the chip holds at once (this kernel runs two
// 512-thread workgroups per CU: 512 slots): the concat GroupNorms of the 64x64 level were 684 workgroups = two rounds, each with
// its own statistics chain (20 us where the one-source form of half the size took 9; round 5).
constexpr int GNA_NT = 512, GNA_SH = GNA_NT / 32;      // threads, shares per group
constexpr int GNA_MAXC = 2560, GNA_NCH = GNA_MAXC / GNA_NT;        // channels (launcher: C / 8 <= 320); table channels per thread
template <int GNA_IT>
__global__ __launch_bounds__(GNA_NT, 4) void gn_apply_kernel(GnArgs p, int pix_per_block) {
  sdmi_kernarg_warm<sizeof(GnArgs) + 4>();     // one miss latency for the argument block instead of one per line the compiler reaches for (common.h)
  __shared__ double s_red[GNA_SH][32][2];
  __shared__ float s_mean[32], s_rstd[32];
  // per-channel {mean of the group, rstd * gamma, beta}: built once per workgroup behind the statistics and read as 16-byte
  // vectors in phase B (round 5).  Before, every thread fetched its items' gamma / beta itself -- 64 bytes per item through the
  // CU's one texture path beside the 32 bytes of the item, and 32 registers held across the statistics -- and looked the group
  // of every element up in s_mean / s_rstd with scalar LDS reads.  Same arithmetic per element: (x - mean) * (rstd gamma) + beta.
  extern __shared__ __attribute__((aligned(16))) float gna_tab[];      // 3 C floats (launcher)
  const int C = p.C0 + p.C1, C8 = C / 8, cpg = C / 32;
  float* const t_mean = gna_tab;
  float* const t_a = gna_tab + C;
  float* const t_b = gna_tab + 2 * C;
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
#ifdef SDMI_GNA_PROBE
  unsigned long long gna_t[6] = {0, 0, 0, 0, 0, 0};
  GNA_STAMP(0);
#endif
  // phase A FIRST: the activation loads (cold, from the producer kernel's XCDs) fly while the statistics are reduced
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, p.P);
  const int items = (p1 - p0) * C8;           // <= GNA_IT * 512 by construction (launcher)
  constexpr int IT = GNA_IT;
  float v[IT][8];
  int px_[IT], c_[IT];
  bool ok[IT];
  const int q_first = tid / C8, r_first = tid - q_first * C8;     // item = tid + NT k -> (pixel, chunk) incrementally
  const int dq = GNA_NT / C8, dr = GNA_NT - dq * C8;
  int qq = q_first, rr = r_first;
#pragma unroll
  for (int k = 0; k < IT; ++k) {               // phase A: every load of this thread in flight at once
    const int it = tid + k * GNA_NT;
    ok[k] = it < items;
    px_[k] = p0 + (ok[k] ? qq : 0);
    c_[k] = (ok[k] ? rr : 0) * 8;
    qq += dq; rr += dr;
    if (rr >= C8) { rr -= C8; ++qq; }
    const bool second = c_[k] >= p.C0;
    const void* base = second ? p.x1 : p.x0;
    const int cs = second ? p.C1 : p.C0;
    const int cc = second ? c_[k] - p.C0 : c_[k];
    if (ok[k]) load8(base, p.in_f32, ((size_t)n * p.P + px_[k]) * cs + cc, v[k]);
  }
  // the layer's gamma / beta come from HBM (every weight is read once per step): requested here, with the activations, instead of
  // behind the statistics (one more exposed memory latency per launch) -- one channel per thread and round, for the table
  float gm[GNA_NCH], bt[GNA_NCH];
#pragma unroll
  for (int k = 0; k < GNA_NCH; ++k) {
    const int c = tid + k * GNA_NT;
    gm[k] = c < C ? p.gamma[c] : 0.f;
    bt[k] = c < C ? p.beta[c] : 0.f;
  }
  GNA_STAMP(1);
  const int g = tid & 31, sl = tid >> 5;
  if (p.acc0 != nullptr) {
    // statistics the producers left behind (GnRec, common.h; sd/diffusion.py:173,199,294,733): thread (group g, share sl of 16)
    // adds up its share of the group's records -- apg atoms x T record rows (x parts), per concat source -- all loads in flight
    // at once, then the shares meet in LDS exactly as the chunk partials below do.  One pass, no statistics launch.
    const int apg = cpg / p.atom, na0 = p.C0 / p.atom;
    const int Tmax = max(p.accT0, p.accT1);
    const int npair = apg * Tmax;                       // (atom of the group, record row) pairs; this thread: sl, sl + 16, ...
    constexpr int MAXR = 12;                            // in flight at once: 4096 records / 32 groups / 16 shares = 8 per source; a concat of a 128-row-block source and a 64-row-block one has 12; the loop below takes what is beyond
    f32x4 rv[MAXR];
    // pair f = sl + 16 k -> (record row t, atom r of the group) incrementally: ONE division per thread instead of one per record
    // (twelve run-time divisions sat in front of the record loads: 1.5 us of the launch by the in-kernel stamps)
    const int dt = GNA_SH / apg, dr = GNA_SH - dt * apg;
    int t_i = sl / apg, r_i = sl - t_i * apg;
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
      const int f = sl + GNA_SH * k;
      rv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int t = t_i, a = g * apg + r_i;               // atom index in concat channel space
      t_i += dt; r_i += dr;
      if (r_i >= apg) { r_i -= apg; ++t_i; }
      if (f < npair) {
        const bool second = a >= na0;
        const int T = second ? p.accT1 : p.accT0, parts = second ? p.accP1 : p.accP0;
        if (t < T) {
          const float* r = (second ? p.acc1 : p.acc0) + ((size_t)(n * T + t) * (second ? p.C1 / p.atom : na0) + (second ? a - na0 : a)) * parts * 2;
          if (parts == 2) rv[k] = *(const f32x4*)r;
          else { const f32x2 u = *(const f32x2*)r; rv[k][0] = u[0]; rv[k][1] = u[1]; }
        }
      }
    }
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int k = 0; k < MAXR; ++k) { s += (double)rv[k][0] + (double)rv[k][2]; q += (double)rv[k][1] + (double)rv[k][3]; }
    for (int f = sl + GNA_SH * MAXR; f < npair; f += GNA_SH) {      // (more records than the producers' bound allows: still correct)
      const int t = f / apg, a = g * apg + (f - t * apg);
      const bool second = a >= na0;
      const int T = second ? p.accT1 : p.accT0, parts = second ? p.accP1 : p.accP0;
      if (t < T) {
        const float* r = (second ? p.acc1 : p.acc0) + ((size_t)(n * T + t) * (second ? p.C1 / p.atom : na0) + (second ? a - na0 : a)) * parts * 2;
        s += (double)r[0]; q += (double)r[1];
        if (parts == 2) { s += (double)r[2]; q += (double)r[3]; }
      }
    }
    s_red[sl][g][0] = s;
    s_red[sl][g][1] = q;
  } else {
    // (statistics: the chunk partials of gn_stats_kernel)
    double s = 0.0, q = 0.0;
    // nchunk <= 128 -> at most 8 chunks per share: all loads issued at once (one latency round)
    f32x2 pv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ch = sl + GNA_SH * k;
      pv[k] = ch < p.nchunk ? *(const f32x2*)(p.partial + (((size_t)n * p.nchunk + ch) * 32 + g) * 2) : f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { s += (double)pv[k][0]; q += (double)pv[k][1]; }
    s_red[sl][g][0] = s;
    s_red[sl][g][1] = q;
  }
  GNA_STAMP(2);
  __syncthreads();
  if (tid < 32) {
    // Variance as E[x^2] - E[x]^2 from fixed-order partial sums.  The cancellation costs eps * (mean/sigma)^2 relative
    // accuracy, so everything above the per-thread fp32 partials (<= 32 values each) is combined in fp64: measured
    // (tests/test_gpu_kernels.py::test_groupnorm_large_mean) the output stays within an fp16 ulp up to |mean| = 100 sigma.
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int k = 0; k < GNA_SH; ++k) { s += s_red[k][tid][0]; q += s_red[k][tid][1]; }
    const double cnt = (double)cpg * (double)p.P;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    s_mean[tid] = (float)mean;
    s_rstd[tid] = rsqrtf((float)var + p.eps);
  }
  __syncthreads();
  GNA_STAMP(3);
  {
    const unsigned magic = ((1u << 20) + cpg - 1) / cpg;      // c / cpg == (c * magic) >> 20 for c < 2560, cpg <= 80
#pragma unroll
    for (int k = 0; k < GNA_NCH; ++k) {
      const int c = tid + k * GNA_NT;
      if (c < C) {
        const int gg = (int)(((unsigned)c * magic) >> 20);
        t_mean[c] = s_mean[gg];
        t_a[c] = s_rstd[gg] * gm[k];
        t_b[c] = bt[k];
      }
    }
  }
  __syncthreads();
  GNA_STAMP(4);
#pragma unroll
  for (int k = 0; k < IT; ++k) {               // phase B: normalise (+SiLU), convert, store
    if (!ok[k]) continue;
    const int c = c_[k];
    const f32x4 m0 = *(const f32x4*)(t_mean + c), m1 = *(const f32x4*)(t_mean + c + 4);
    const f32x4 a0 = *(const f32x4*)(t_a + c), a1 = *(const f32x4*)(t_a + c + 4);
    const f32x4 b0 = *(const f32x4*)(t_b + c), b1 = *(const f32x4*)(t_b + c + 4);
    f16x8 o;
    float yf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float mu = e < 4 ? m0[e] : m1[e - 4], aa = e < 4 ? a0[e] : a1[e - 4], bb = e < 4 ? b0[e] : b1[e - 4];
      float y = (v[k][e] - mu) * aa + bb;
      if (p.silu) y = y * __builtin_amdgcn_rcpf(1.f + __expf(-y));     // v_rcp_f32 (1 ulp), not the ~10-instruction IEEE division
      o[e] = (f16)y;
      yf[e] = y;
    }
    *(f16x8*)(p.y + ((size_t)n * p.P + px_[k]) * C + c) = o;
    if (p.y32) {                                 // accurate mode: the conv that follows reads the fp32 values
      float* yp = p.y32 + ((size_t)n * p.P + px_[k]) * C + c;
      *(f32x4*)yp = f32x4{yf[0], yf[1], yf[2], yf[3]};
      *(f32x4*)(yp + 4) = f32x4{yf[4], yf[5], yf[6], yf[7]};
    }
  }
#ifdef SDMI_GNA_PROBE
  GNA_STAMP(5);
  const int bid = blockIdx.y * gridDim.x + blockIdx.x;
  if (tid == 0 && bid < 2048)
    for (int i = 0; i < 6; ++i) g_gna_clk[bid][i] = i ? gna_t[i] - gna_t[0] : gna_t[0];
#endif
}

// Single-launch GroupNorm for small feature maps: one 512-thread block per (group, image) keeps its whole
// P x cpg slab in registers (<= MAXQ 4-channel quads per thread), exact two-pass statistics, fixed-order
// block reductions (deterministic).  Replaces stats+apply where the slab fits: one kernel boundary less.
// Accesses are 4 channels wide (16-B fp32 / 8-B fp16 loads, 8-B stores): the first form moved 2 channels per access
// (4-byte stores) and spent most of its 6-11 us issuing them.
constexpr int GNF_NT = 512;
// SLAB: the slab of the (group, image) is not read from a tensor but summed from the split-K partial sums of the producing
// conv (GnArgs::slab), + bias (+ residual), in slab order like splitk_finalize: that launch and its 0.5 - 1 GB/step of
// round trip through the finished tensor are gone, and the statistics are taken from the fp32 sums instead of their fp16 copy.
template <int MAXQ, bool SLAB>
__global__ __launch_bounds__(GNF_NT) void gn_fused_kernel(GnArgs p) {
  sdmi_kernarg_warm<sizeof(GnArgs)>();     // one miss latency for the argument block instead of one per line the compiler reaches for (common.h)
  __shared__ float s_w[2][GNF_NT / 64];
  __shared__ __attribute__((aligned(16))) float s_gb[2][128];        // this group's gamma / beta (cpg <= 128)
  const int C = p.C0 + p.C1, cpg = C / 32, q4 = cpg / 4;
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int total = p.P * q4;
  // the layer's own gamma / beta come from HBM: requested here, with the slab, instead of next to their use behind the two
  // reductions (that was one more exposed memory latency per launch); visible after the reductions' barriers
  if (tid < cpg) { s_gb[0][tid] = p.gamma[g * cpg + tid]; s_gb[1][tid] = p.beta[g * cpg + tid]; }
  f32x4 v[MAXQ];
  float s = 0.f;
  // slot = tid + i*NT walks (pixel, channel quad) incrementally: one integer division per thread instead of one per
  // slot and phase (the compiler's 32-bit division is ~30 VALU instructions)
  const int px_first = tid / q4, j_first = tid - px_first * q4;
  const int dpx = GNF_NT / q4, dj = GNF_NT - dpx * q4;
  int px = px_first, j = j_first;
  if constexpr (SLAB) {
    // 64 workgroups pull ksplit x (their share of the map): what matters is bytes in flight.  Slab-major order -- for each
    // slab (or group of ZB slabs) the loads of ALL of this thread's slots are issued before anything is added -- keeps
    // MAXQ (x ZB) 16-byte loads per thread outstanding instead of one slot's four; the additions per element stay in slab
    // order, so the sum is bit-identical to splitk_finalize's.
    const size_t MN = (size_t)p.B * p.P * C;
    constexpr int ZB = MAXQ <= 4 ? 4 : 1;
    size_t off[MAXQ];
    bool ok[MAXQ];
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
      ok[i] = tid + i * GNF_NT < total;
      off[i] = ok[i] ? ((size_t)n * p.P + px) * C + g * cpg + 4 * j : 0;
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      px += dpx; j += dj;
      if (j >= q4) { j -= q4; ++px; }
    }
    f32x4 r[MAXQ];
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {                 // residual first: the coldest load
      r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok[i] && p.sres) {
        if (p.sres_f32) r[i] = *(const f32x4*)((const float*)p.sres + off[i]);
        else { const f16x4 hv = *(const f16x4*)((const f16*)p.sres + off[i]); r[i] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]}; }
      }
    }
    for (int z0 = 0; z0 < p.ksplit; z0 += ZB) {
      f32x4 t[ZB][MAXQ];
#pragma unroll
      for (int zz = 0; zz < ZB; ++zz)
#pragma unroll
        for (int i = 0; i < MAXQ; ++i)
          t[zz][i] = (ok[i] && z0 + zz < p.ksplit) ? *(const f32x4*)(p.slab + (size_t)(z0 + zz) * MN + off[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int zz = 0; zz < ZB; ++zz)
#pragma unroll
        for (int i = 0; i < MAXQ; ++i)
          if (z0 + zz < p.ksplit) v[i] += t[zz][i];
    }
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
      if (ok[i]) {
        if (p.sbias) v[i] += *(const f32x4*)(p.sbias + (off[i] % C));
        v[i] += r[i];
        if (p.sout) *(f32x4*)(p.sout + off[i]) = v[i];
        if (p.sout16) *(f16x4*)(p.sout16 + off[i]) = f16x4{(f16)v[i][0], (f16)v[i][1], (f16)v[i][2], (f16)v[i][3]};
      }
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
      const int slot = tid + i * GNF_NT;
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (slot < total) {
        const int c = g * cpg + 4 * j;
        const bool second = c >= p.C0;
        const void* base = second ? p.x1 : p.x0;
        const int cs = second ? p.C1 : p.C0;
        const size_t off = ((size_t)n * p.P + px) * cs + (second ? c - p.C0 : c);
        if (p.in_f32) v[i] = *(const f32x4*)((const float*)base + off);
        else { const f16x4 hv = *(const f16x4*)((const f16*)base + off); v[i] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]}; }
      }
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      px += dpx; j += dj;
      if (j >= q4) { j -= q4; ++px; }
    }
  }
  auto block_sum = [&](float x, int which) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if ((tid & 63) == 0) s_w[which][tid >> 6] = x;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < GNF_NT / 64; ++w) t += s_w[which][w];
    return t;
  };
  const float cnt = (float)cpg * (float)p.P;
  const float mean = block_sum(s, 0) / cnt;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    if (tid + i * GNF_NT < total) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(block_sum(q, 1) / cnt + p.eps);
  px = px_first; j = j_first;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int slot = tid + i * GNF_NT;
    if (slot < total) {
      const int c = g * cpg + 4 * j;
      const f32x4 ga = *(const f32x4*)(&s_gb[0][4 * j]), be = *(const f32x4*)(&s_gb[1][4 * j]);
      f16x4 o;
      f32x4 yf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = (v[i][e] - mean) * rstd * ga[e] + be[e];
        if (p.silu) y = y * __builtin_amdgcn_rcpf(1.f + __expf(-y));
        o[e] = (f16)y;
        yf[e] = y;
      }
      *(f16x4*)(p.y + ((size_t)n * p.P + px) * C + c) = o;
      if (p.y32) *(f32x4*)(p.y32 + ((size_t)n * p.P + px) * C + c) = yf;
    }
    px += dpx; j += dj;
    if (j >= q4) { j -= q4; ++px; }
  }
}

// LayerNorm: each wave normalises LN_ROWS rows (all of their loads are issued before the first reduction);
// C in {320, 640, 768, 1280} -> 40..160 chunks of 8 -> up to 3 chunks per lane.  Exact two-pass statistics.
constexpr int LN_ROWS = 4;
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs p) {
  sdmi_kernarg_warm<sizeof(LnArgs)>();     // one miss latency for the argument block instead of one per line the compiler reaches for (common.h)
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_ROWS;
  if (row0 >= p.M) return;
  const int C8 = p.C / 8;
  float v[LN_ROWS][NCH][8];
#pragma unroll
  for (int rr = 0; rr < LN_ROWS; ++rr) {
    const int row = min(row0 + rr, p.M - 1);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c8 = lane + i * 64;
      if (c8 < C8) load8(p.x, p.in_f32, (size_t)row * p.C + c8 * 8, v[rr][i]);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[rr][i][e] = 0.f;
      }
    }
  }
#pragma unroll
  for (int rr = 0; rr < LN_ROWS; ++rr) {
    const int row = row0 + rr;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[rr][i][e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)p.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c8 = lane + i * 64;
      if (c8 < C8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[rr][i][e] - mean; q += d * d; }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)p.C + p.eps);
    if (row >= p.M) continue;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c8 = lane + i * 64;
      if (c8 < C8) {
        const int c = c8 * 8;
        const f32x4 ga = *(const f32x4*)(p.gamma + c), gb = *(const f32x4*)(p.gamma + c + 4);
        const f32x4 ba = *(const f32x4*)(p.beta + c), bb = *(const f32x4*)(p.beta + c + 4);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float yv = (v[rr][i][e] - mean) * rstd * (e < 4 ? ga[e] : gb[e - 4]) + (e < 4 ? ba[e] : bb[e - 4]);
          o[e] = (f16)yv;
          if (p.y32) p.y32[(size_t)row * p.C + c + e] = yv;
        }
        if (p.y) *(f16x8*)(p.y + (size_t)row * p.C + c) = o;
      }
    }
  }
}

}  // namespace

int sdmi_gn_nchunk(int P) {
  // >= 8 pixels per chunk, at most 128 chunks per image
  int ppc = (P + 127) / 128;
  if (ppc < 8) ppc = 8;
  return (P + ppc - 1) / ppc;
}

// 1 when the single-launch kernel takes this shape (slab of one group fits the block's registers), else 2
int sdmi_gn_launches(const GnArgs& a) {
  if (a.acc0) return 1;
  static const int max_px = getenv("SDMI_GN_FUSED_MAXPX") ? atoi(getenv("SDMI_GN_FUSED_MAXPX")) : 1024;   // up to 32x32 the one launch is as fast as stats + apply (same-box 4.277 vs 4.281 ms/step, 11 launches fewer); beyond, 64 blocks cannot pull the map fast enough
  const int C = a.C0 + a.C1, cpg = C / 32;
  const long quads = ((long)a.P * (cpg / 4) + GNF_NT - 1) / GNF_NT;
  return (cpg % 4 == 0 && cpg <= 128 && a.C0 % 4 == 0 && a.P <= max_px && quads <= 12) ? 1 : 2;
}
// ~1024 items (8-channel chunks) per block, 2048 when that keeps the launch inside one round of workgroups (two per CU)
static void launch_gn_apply(const GnArgs& a, int C8, hipStream_t st) {
  const int C = C8 * 8;
  static const int force_it = getenv("SDMI_GNA_IT") ? atoi(getenv("SDMI_GNA_IT")) : 0;      // A/B only: 2 or 4
  int ppb2 = (GNA_NT * 2) / C8;
  if (ppb2 < 1) ppb2 = 1;
  const long blocks2 = (long)((a.P + ppb2 - 1) / ppb2) * a.B;
  const bool four = force_it ? force_it == 4 : (blocks2 > 512 && 4 * GNA_NT / C8 >= 2);
  if (four) {
    const int ppb = (GNA_NT * 4) / C8;
    hipLaunchKernelGGL(gn_apply_kernel<4>, dim3((a.P + ppb - 1) / ppb, a.B), dim3(GNA_NT), 3 * C * sizeof(float), st, a, ppb);
  } else {
    hipLaunchKernelGGL(gn_apply_kernel<2>, dim3((a.P + ppb2 - 1) / ppb2, a.B), dim3(GNA_NT), 3 * C * sizeof(float), st, a, ppb2);
  }
}

int sdmi_launch_groupnorm(const GnArgs& a, hipStream_t st) {
  const int C = a.C0 + a.C1;
  SDMI_REQUIRE(C % 32 == 0 && C % 8 == 0 && a.C0 % 8 == 0, "groupnorm: C=%d (C0=%d) must be multiples of 32/8", C, a.C0);
  SDMI_REQUIRE(C / 8 <= 320 && C >= 128, "groupnorm: C=%d out of range (128..2560)", C);
  SDMI_REQUIRE(a.partial && a.y && (a.x0 || a.slab) && a.gamma && a.beta, "groupnorm: null pointer");
  SDMI_REQUIRE(a.nchunk == sdmi_gn_nchunk(a.P), "groupnorm: nchunk mismatch");
  SDMI_REQUIRE(!a.slab || (a.C1 == 0 && a.ksplit >= 1 && sdmi_gn_launches(a) == 1),
               "groupnorm: the split-K slab input needs one source and a map the single-launch kernel takes");
  if (a.acc0) {
    const int cpg = C / 32;
    SDMI_REQUIRE(!a.slab && a.atom >= 4 && cpg % a.atom == 0 && a.C0 % a.atom == 0 && (a.C1 == 0 || a.acc1) && cpg / a.atom <= 8 && C % 32 == 0 &&
                     a.accT0 > 0 && (a.accP0 == 1 || a.accP0 == 2) && (a.C1 == 0 || (a.accT1 > 0 && (a.accP1 == 1 || a.accP1 == 2))),
                 "groupnorm: producer statistics need atoms that tile every group and both concat sources, T > 0 and parts 1 or 2");
    const int C8a = C / 8;
    launch_gn_apply(a, C8a, st);
    SDMI_CHECK_HIP(hipGetLastError());
    return SDMI_OK;
  }
  if (sdmi_gn_launches(a) == 1) {
    {
      const long quads = ((long)a.P * (C / 128) + GNF_NT - 1) / GNF_NT;
      const dim3 grid(32, a.B), block(GNF_NT);
      if (a.slab) {
        if (quads <= 4) hipLaunchKernelGGL((gn_fused_kernel<4, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((gn_fused_kernel<12, true>), grid, block, 0, st, a);
      } else {
        if (quads <= 4) hipLaunchKernelGGL((gn_fused_kernel<4, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((gn_fused_kernel<12, false>), grid, block, 0, st, a);
      }
      SDMI_CHECK_HIP(hipGetLastError());
      return SDMI_OK;
    }
  }
  const int C8 = C / 8;
  int PY = 256 / C8;
  if (PY < 1) PY = 1;
  const int ppc = (a.P + a.nchunk - 1) / a.nchunk;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(a.nchunk, a.B), dim3(C8 * PY), 0, st, a, PY, ppc);
  SDMI_CHECK_HIP(hipGetLastError());
  // apply: ~1024 items (8-channel chunks) per block = 4 per thread
  launch_gn_apply(a, C8, st);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

// statistics only: partial {sum, sum of squares} per (image, pixel chunk, group) (consumer: the back-to-back kernel, SDMI_B2B_GN)
int sdmi_launch_gn_stats(const GnArgs& a, hipStream_t st) {
  const int C = a.C0 + a.C1;
  SDMI_REQUIRE(C % 32 == 0 && C % 8 == 0 && a.C0 % 8 == 0, "gn_stats: C=%d (C0=%d) must be multiples of 32/8", C, a.C0);
  SDMI_REQUIRE(C / 8 <= 320 && C >= 128, "gn_stats: C=%d out of range (128..2560)", C);
  SDMI_REQUIRE(a.partial && a.x0, "gn_stats: null pointer");
  SDMI_REQUIRE(a.nchunk == sdmi_gn_nchunk(a.P), "gn_stats: nchunk mismatch");
  const int C8 = C / 8;
  int PY = 256 / C8;
  if (PY < 1) PY = 1;
  const int ppc = (a.P + a.nchunk - 1) / a.nchunk;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(a.nchunk, a.B), dim3(C8 * PY), 0, st, a, PY, ppc);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_layernorm(const LnArgs& a, hipStream_t st) {
  SDMI_REQUIRE(a.C % 8 == 0 && a.C <= 8 * 64 * 3, "layernorm: C=%d unsupported", a.C);
  SDMI_REQUIRE(a.x && (a.y || a.y32) && a.gamma && a.beta && a.M > 0, "layernorm: bad args");
  const int nch = (a.C / 8 + 63) / 64;
  dim3 grid((a.M + 4 * LN_ROWS - 1) / (4 * LN_ROWS)), block(256);
  if (nch == 1) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, st, a);
  else if (nch == 2) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, st, a);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}
