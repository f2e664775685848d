// Back-to-back GEMM for the 320-channel attention blocks (sd/diffusion.py:312-363 at the 64x64 level):
//
//     S = A1 W1^T + b1 (+ R1)                    (conv_input, or an attention out_proj + residual: sd/diffusion.py:312,325,343)
//     Y = f(LayerNorm(S)) W2^T ... + b2 (+ R2)   (the next Linear, with the LayerNorm folded as in gemm.hip)
//
// in ONE launch: a workgroup owns 32 or 64 full rows (all 320 columns), so the row statistics of S are available in-tile and
// the fp16 copy of S stays in LDS as the A operand of the second product.  Forms of the second product:
//   full fold   (partial = 0):  Y = cscale * (LN0(S) W2'^T + h)                       -- q_proj after out_proj 1
//   partial fold(partial = 1):  Y = LN0(S) W2a'^T + S W2b^T + h + R2,  W2 = [W2a' | W2b] -- the composed feed-forward
//   three passes (npass2 = 3):  [q | k | v] = LN0(S) W2'^T + h, 320 columns per pass    -- in_proj after conv_input; q scaled,
//                               v written transposed in the attention kernel's key order
// with LN0(S) = (S - mean) rstd, W2' = gamma (.) W2 and h = b + W2 beta (sdmi_launch_ln_fold_prep).  Optionally (gx) the
// first A operand is the block's GroupNorm of the raw stream, applied by the DMA waves on the way into LDS.
// The K = C GEMMs of these blocks are bound by launch latency plus the fp32 stream's bytes (gemm.hip: ~5 us + bytes / 5 TB/s
// for 1.7 GFLOP); fusing a pair removes one launch, the re-read of S and -- for the feed-forward, whose input nobody
// else reads -- the 15.7 MB write of S.
//
// Workgroup = 2 or 4 MFMA waves (32 rows x 160 columns each: 5 accumulator blocks of 32x32) + 8 DMA waves that stream the
// 320x64 weight tile (+ the A1 tile) of each K-step into a THREE-stage LDS ring with global_load_lds (counted vmcnt: two
// tiles in flight while one is multiplied -- with two stages every K-step waited a full L2 latency); one s_barrier per
// K-step.  LDS: weight ring 3 x 40 KiB (aliased by the fp32 tile of the epilogues) + 20 / 40 KiB that hold the A1 ring during
// the first product and then S as fp16 in the A-operand layout (five 64-column panels, 16-byte chunks XOR-swizzled on the
// row).  The LayerNorm is applied when the panels are written (full fold) or to the A fragments in registers on their way to
// the MFMA (partial fold: the plain half needs S itself); gamma / beta are in the folded weights and bias.  The DMA waves
// run both epilogues (the MFMA waves keep their registers for accumulators) over rows whose residuals were fetched into
// registers before the product -- a load-use chain per row cost six cold latencies per epilogue.
#include "common.h"

namespace {

constexpr int kC = 320, kDW = 8;
constexpr int kNS = 3;
constexpr int kWStage = kC * 128;                              // 40 KiB
constexpr int kRing = kNS * kWStage;                           // 120 KiB
constexpr int kCsLd = kC + 4;                                  // fp32 tile row stride: +4 floats so the two lane halves of an
                                                               // accumulator store (rows 4 apart) fall on different LDS banks
// BM rows per workgroup: 64 (4 MFMA waves) or 32 (2 MFMA waves; twice the workgroups -- one per CU at M = 8192 -- for
// twice the weight traffic out of L2, and half the epilogue per workgroup)
template <int BM>
struct B2bCfg {
  static constexpr int kBM = BM, kMW = BM / 16, kNT = 64 * (kMW + kDW), kNR = BM / kDW;   // kNR: epilogue rows per DMA wave
  static constexpr int kAStage = BM * 128, kS16 = 5 * kAStage;
  static constexpr int kLds = kRing + kS16;
  static constexpr int kCsBytes = BM * kCsLd * 4;
  static_assert(BM == 32 || BM == 64, "tile height");
  static_assert(kLds <= 160 * 1024, "LDS budget");
  static_assert(kCsBytes + 2 * BM * 4 <= kRing, "the fp32 epilogue tile and the row statistics alias the weight ring");
  static_assert(kNS * kAStage <= kS16, "the A1 ring aliases the S panels");
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

#ifdef SDMI_B2B_PROBE
__device__ unsigned long long g_b2b_clk[2][8];   // diagnostic build: shader-clock stamps of workgroup 0 {MFMA wave 0, DMA wave 0}
#define B2B_STAMP(role, i) do { if (blockIdx.x == 0 && lane == 0 && wave_id == (role ? Cf::kMW : 0)) g_b2b_clk[role][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define B2B_STAMP(role, i) do { } while (0)
#endif

// GACC: the launch also accumulates the GroupNorm statistics of its output (B2bArgs::gacc; the feed-forward form, whose output
// is the block's result).  A template parameter, not a run-time flag: the kernel sits at the 168-register limit of three waves
// per SIMD and the other forms must not pay for the eight extra accumulators.
template <class Cf, bool GACC>
__global__ __launch_bounds__(Cf::kNT) void b2b_kernel(B2bArgs p) {
  sdmi_kernarg_warm<sizeof(B2bArgs)>();
  constexpr int kBM = Cf::kBM, kMW = Cf::kMW, kNR = Cf::kNR, kAStage = Cf::kAStage, kCsBytes = Cf::kCsBytes;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * kBM;
  char* ring = smem;                       // weight ring; Cs + s_ln alias it between the products
  char* S16 = smem + kRing;                // A1 ring (first product), then S as fp16
  float* Cs = (float*)smem;
  float* s_ln = (float*)(smem + kCsBytes);
  const int n2 = p.K2 / 64;

  // The two roles are separate code paths with their own variables, so the register allocator overlays the MFMA waves'
  // accumulators with the DMA waves' prefetched residual rows.  Both paths execute the same sequence of barriers.
  if (wave_id >= kMW) {
    // =============================== DMA + epilogue waves ===============================
    const int dw = wave_id - kMW;
    // the lane index from the execution mask (v_mbcnt), not from threadIdx: otherwise the thread-id register stays live -- and,
    // at the 168-register limit, is spilled -- across the whole MFMA path (the two paths join behind their last barrier)
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int a_row = (dw * 64 + lane) >> 3, a_pc = lane & 7;
    const f16* a_src = p.a1 + (size_t)(m0 + (a_row < kBM ? a_row : 0)) * p.lda1 + ((a_pc ^ ((a_row >> 1) & 7)) * 8);
    // one K-step = (first product only) the 64x64 A1 tile + the 320x64 weight tile, 16 B per lane per instruction
    auto issue = [&](const f16* w, int ldw, int kofs, bool with_a, int stage) {
      char* sa = S16 + stage * kAStage;
      char* sb = ring + stage * kWStage;
      if (with_a && dw * 8 < kBM) glds16(a_src + kofs, sa + dw * 1024);      // 8 rows per wave instruction
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int q = (i * kDW + dw) * 64 + lane;
        const int row = q >> 3, pc = q & 7;
        glds16(w + (size_t)row * ldw + kofs + ((pc ^ ((row >> 1) & 7)) * 8), sb + (i * kDW + dw) * 1024);
      }
    };
    // `young`: ordinary loads issued AFTER the first two tiles (the residual rows of the first product, fetched behind the
    // tiles so the cold first tile is requested as early as possible): they are younger than tiles 0 and 1, so the waits for
    // those two tiles tolerate them in flight; tile 2 is younger than they are, so from then on the count is the ring's own.
    auto stream = [&](const f16* w, int ldw, int n, bool a_global, int young) {
      for (int t = 0; t < n; ++t) {
        // tile t has landed when at most the newest group (tile t+1) [+ the young loads] is still in flight
        if (t + 1 < n) {
          if (a_global) {
            // `young` is a LOWER bound of the young loads (a stricter wait is always safe, a laxer one never)
            // (waves beyond the A tile's rows issue 5 loads per tile, the others 6: the count is per wave)
            if (dw * 8 < kBM) {
              if (t < 2 && young >= 16) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
              else if (t < 2 && young >= 8) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
              else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
              if (t < 2 && young >= 16) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
              else if (t < 2 && young >= 8) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
              else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            }
          } else {
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
          }
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (t + 2 < n) issue(w, ldw, (t + 2) * 64, a_global, (t + 2) % kNS);   // its stage was multiplied in step t-1
      }
    };
    // Epilogue ownership: wave dw takes rows dw + 8 i; lane l < 40 takes two column quads, 4l..4l+3 and 160+4l..163+4l, so
    // that a 16-byte LDS / global access has a 16-byte lane stride (8 contiguous columns per lane made every ds_read_b128 of
    // the fp32 tile a 4-way bank conflict: 4 400 cycles for this pass).
    // Residual rows are fetched into registers BEFORE the product they follow so the epilogue never waits on memory (a
    // load-use chain per row cost six cold latencies per epilogue: 34 us per launch).  The loads are older than every ring
    // load of the product, so the counted vmcnt waits of the ring retire them first.
    const bool own = lane < kC / 8;
    const int c0 = lane * 4, c1 = kC / 2 + lane * 4;
    f32x4 rv[kNR][2], bv[2];
    auto fetch = [&](const void* res, int is_f32, const float* bias) {
#pragma unroll
      for (int i = 0; i < kNR; ++i) {
        rv[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        rv[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (res && own) {
        if (is_f32) {
#pragma unroll
          for (int i = 0; i < kNR; ++i) {
            const float* rp = (const float*)res + (size_t)(m0 + dw + 8 * i) * kC;
            rv[i][0] = *(const f32x4*)(rp + c0);
            rv[i][1] = *(const f32x4*)(rp + c1);
          }
        } else {
          f16x4 t[kNR][2];
#pragma unroll
          for (int i = 0; i < kNR; ++i) {
            const f16* rp = (const f16*)res + (size_t)(m0 + dw + 8 * i) * kC;
            t[i][0] = *(const f16x4*)(rp + c0);
            t[i][1] = *(const f16x4*)(rp + c1);
          }
#pragma unroll
          for (int i = 0; i < kNR; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { rv[i][0][e] = (float)t[i][0][e]; rv[i][1][e] = (float)t[i][1][e]; }
        }
      }
      bv[0] = own ? *(const f32x4*)(bias + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
      bv[1] = own ? *(const f32x4*)(bias + c1) : f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- product 1, epilogue 1: S = A1 W1^T + b1 + R1, its fp16 copy into the A-operand panels, row statistics ----
    B2B_STAMP(1, 0);
    const bool gn_in = p.gx != nullptr;                  // A1 = GroupNorm(gx) computed here (no A1 tensor, no A1 ring)
    issue(p.w1, kC, 0, !gn_in, 0);
    issue(p.w1, kC, 64, !gn_in, 1);
    if (gn_in) {
      // GroupNorm(32 groups of 10 channels) of this tile's rows, as gn_apply_kernel (norm.hip) does it: group statistics
      // from the chunk partials of gn_stats_kernel summed in fp64 (lane g of every wave owns group g), then
      // y = (x - mean) rstd gamma + beta in fp32, rounded to fp16 into the A-operand panels of the first product.
      const int img = m0 / p.S;
      // the rows first: their (cold) loads fly while the statistics are reduced
      f32x4 xv[kNR][2];
#pragma unroll
      for (int i = 0; i < kNR; ++i) {
        const size_t off = (size_t)(m0 + dw + 8 * i) * kC;
        if (p.gx_f32) {
          xv[i][0] = own ? *(const f32x4*)((const float*)p.gx + off + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
          xv[i][1] = own ? *(const f32x4*)((const float*)p.gx + off + c1) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
          const f16x4 t0 = own ? *(const f16x4*)((const f16*)p.gx + off + c0) : f16x4{};
          const f16x4 t1 = own ? *(const f16x4*)((const f16*)p.gx + off + c1) : f16x4{};
#pragma unroll
          for (int e = 0; e < 4; ++e) { xv[i][0][e] = (float)t0[e]; xv[i][1][e] = (float)t1[e]; }
        }
      }
      // lane (half, g): chunks j = half, half + 2, ... of group g, two interleaved fp64 chains (a single 128-long chain of
      // dependent fp64 adds cost ~2 us at the start of every workgroup); the halves meet through one shuffle
      double gs0 = 0.0, gq0 = 0.0, gs1 = 0.0, gq1 = 0.0;
      if (p.gn_parts == 2) {
        // the producers' statistics records (GnRec, parts = 2: an entry is two {sum, sum of squares} pairs): same order of additions
        // per entry as gn_apply_kernel's (pair 0 + pair 1), entries chunk by chunk
        const float* pp = p.gn_partial + ((size_t)img * p.gn_nchunk * 32 + (lane & 31)) * 4;
#pragma unroll 4
        for (int j = lane >> 5; j < p.gn_nchunk; j += 4) {
          const f32x4 t = *(const f32x4*)(pp + (size_t)j * 128);
          gs0 += (double)t[0] + (double)t[2]; gq0 += (double)t[1] + (double)t[3];
          if (j + 2 < p.gn_nchunk) {
            const f32x4 u = *(const f32x4*)(pp + (size_t)(j + 2) * 128);
            gs1 += (double)u[0] + (double)u[2]; gq1 += (double)u[1] + (double)u[3];
          }
        }
      } else {
      const float* pp = p.gn_partial + ((size_t)img * p.gn_nchunk * 32 + (lane & 31)) * 2;
#pragma unroll 4
      for (int j = lane >> 5; j < p.gn_nchunk; j += 4) {
        const f32x2 t = *(const f32x2*)(pp + (size_t)j * 64);
        gs0 += (double)t[0]; gq0 += (double)t[1];
        if (j + 2 < p.gn_nchunk) {
          const f32x2 u = *(const f32x2*)(pp + (size_t)(j + 2) * 64);
          gs1 += (double)u[0]; gq1 += (double)u[1];
        }
      }
      }
      double gs = gs0 + gs1, gq = gq0 + gq1;
      gs += __shfl_xor(gs, 32);
      gq += __shfl_xor(gq, 32);
      const double cnt = (double)(kC / 32) * (double)p.S;
      const double gmean = gs / cnt;
      double gvar = gq / cnt - gmean * gmean;
      gvar = gvar < 0.0 ? 0.0 : gvar;
      const float mean_g = (float)gmean, rstd_g = rsqrtf((float)gvar + p.gn_eps);
      float cm[2][4], cg[2][4], cb[2][4];                // per owned channel: group mean, rstd * gamma, beta
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = (own ? (q ? c1 : c0) : 0) + e;
          const int g = c / (kC / 32);
          cm[q][e] = __shfl(mean_g, g);
          cg[q][e] = __shfl(rstd_g, g) * p.gn_gamma[c];
          cb[q][e] = p.gn_beta[c];
        }
      if (own) {
#pragma unroll
        for (int i = 0; i < kNR; ++i) {
          const int row = dw + 8 * i;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            f16x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = (f16)((xv[i][q][e] - cm[q][e]) * cg[q][e] + cb[q][e]);
            const int n = q ? c1 : c0;
            *(f16x4*)(S16 + (n >> 6) * kAStage + row * 128 + (((((n >> 3) & 7) ^ ((row >> 1) & 7))) << 4) + (n & 7) * 2) = a;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // panels written before the first barrier of the product
    }
    fetch(p.r1, p.r1_f32, p.b1);                         // >= two loads per row when there is a residual
    B2B_STAMP(1, 1);
    stream(p.w1, kC, kC / 64, !gn_in, (!gn_in && p.r1) ? 2 * kNR : 0);
    B2B_STAMP(1, 2);
    __syncthreads();                                     // B1: every ring read retired
    __syncthreads();                                     // B2: accumulators are in Cs
    B2B_STAMP(1, 3);
    // The fp32 tile is read four rows at a time, unconditionally (lanes >= 40 read column 0 and discard it), so the LDS
    // reads of a batch are in flight together instead of one latency per quad behind a divergent branch.
    f16x4 o16[kNR][2];
    float su[kNR], sq[kNR];
    const int cs0 = own ? c0 : 0, cs1 = own ? c1 : 0;
#pragma unroll
    for (int hb = 0; hb < kNR; hb += 4) {
      f32x4 cv[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        cv[i][0] = *(const f32x4*)(Cs + (dw + 8 * (hb + i)) * kCsLd + cs0);
        cv[i][1] = *(const f32x4*)(Cs + (dw + 8 * (hb + i)) * kCsLd + cs1);
      }
      // The epilogue is vector-ALU bound (64 x 320 elements on one CU), so the statistics of the fp16 values use the packed
      // dot product: sum += x0*1 + x1*1 and sumsq += x0*x0 + x1*x1 are one v_dot2_f32_f16 per element pair each.
      const f16x2 ones = f16x2{(f16)1.f, (f16)1.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        su[hb + i] = 0.f; sq[hb + i] = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = own ? cv[i][q][e] + bv[q][e] + rv[hb + i][q][e] : 0.f;
            cv[i][q][e] = v;
            o16[hb + i][q][e] = (f16)v;
          }
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const f16x2 x2 = f16x2{o16[hb + i][q][e], o16[hb + i][q][e + 1]};
            su[hb + i] = __builtin_amdgcn_fdot2(x2, ones, su[hb + i], false);
            sq[hb + i] = __builtin_amdgcn_fdot2(x2, x2, sq[hb + i], false);
          }
        }
      }
      if (own) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const size_t mo = (size_t)(m0 + dw + 8 * (hb + i)) * kC;
          if (p.s32) { *(f32x4*)(p.s32 + mo + c0) = cv[i][0]; *(f32x4*)(p.s32 + mo + c1) = cv[i][1]; }
          if (p.s16) { *(f16x4*)(p.s16 + mo + c0) = o16[hb + i][0]; *(f16x4*)(p.s16 + mo + c1) = o16[hb + i][1]; }
        }
      }
    }
#ifdef SDMI_B2B_PROBE2
    B2B_STAMP(1, 4);
#endif
    // Wave reduction of 8 rows x 2 statistics as ONE butterfly that halves the number of live values per step (4 + 2 + 1
    // exchanges, then 3 plain steps: 10 ds_bpermute per statistic instead of 48 -- the LDS pipe is shared by the 8 waves).
    // Afterwards the 8 lanes with equal bits 5..3 hold the totals of row (bit5, bit4, bit3).
    {
      const bool hi5 = lane & 32, hi4 = lane & 16, hi3 = lane & 8;
      float a1, b1;
      if constexpr (kNR == 8) {
        float a4[4], b4[4], a2[2], b2[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float sa = hi5 ? su[i] : su[i + 4], ka = hi5 ? su[i + 4] : su[i];
          const float sb = hi5 ? sq[i] : sq[i + 4], kb = hi5 ? sq[i + 4] : sq[i];
          a4[i] = ka + __shfl_xor(sa, 32);
          b4[i] = kb + __shfl_xor(sb, 32);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float sa = hi4 ? a4[i] : a4[i + 2], ka = hi4 ? a4[i + 2] : a4[i];
          const float sb = hi4 ? b4[i] : b4[i + 2], kb = hi4 ? b4[i + 2] : b4[i];
          a2[i] = ka + __shfl_xor(sa, 16);
          b2[i] = kb + __shfl_xor(sb, 16);
        }
        const float sa = hi3 ? a2[0] : a2[1], ka = hi3 ? a2[1] : a2[0];
        const float sb = hi3 ? b2[0] : b2[1], kb = hi3 ? b2[1] : b2[0];
        a1 = ka + __shfl_xor(sa, 8);
        b1 = kb + __shfl_xor(sb, 8);
      } else {                                             // 4 rows: bits 5, 4 select the row, bit 3 is a plain step
        float a2[2], b2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float sa = hi5 ? su[i] : su[i + 2], ka = hi5 ? su[i + 2] : su[i];
          const float sb = hi5 ? sq[i] : sq[i + 2], kb = hi5 ? sq[i + 2] : sq[i];
          a2[i] = ka + __shfl_xor(sa, 32);
          b2[i] = kb + __shfl_xor(sb, 32);
        }
        const float sa = hi4 ? a2[0] : a2[1], ka = hi4 ? a2[1] : a2[0];
        const float sb = hi4 ? b2[0] : b2[1], kb = hi4 ? b2[1] : b2[0];
        a1 = ka + __shfl_xor(sa, 16);
        b1 = kb + __shfl_xor(sb, 16);
        a1 += __shfl_xor(a1, 8);
        b1 += __shfl_xor(b1, 8);
        (void)hi3;
      }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o); b1 += __shfl_xor(b1, o); }
      // statistics of the fp16 values the second product multiplies (as the row-statistics epilogue of gemm.hip)
      // the variance E[x^2] - E[x]^2 in fp64: one cancellation per row, (mean/sigma)^2 ulps of fp32 otherwise
      const double mean_d = (double)a1 * (1.0 / kC);
      double var_d = (double)b1 * (1.0 / kC) - mean_d * mean_d;
      var_d = var_d < 0.0 ? 0.0 : var_d;
      const float mean = (float)mean_d;
      const float rstd = rsqrtf((float)var_d + p.eps);
      // (every lane of an 8-lane group holds the same row: one of them reports)
      if (p.ln_guard != nullptr && (lane & 7) == 0 && mean_d * mean_d > (double)p.ln_guard_thr2 * var_d) atomicAdd(p.ln_guard, 1);
#pragma unroll
      for (int i = 0; i < kNR; ++i) {                    // the lanes holding row i of this wave: its index in lane bits 5.. down
        const int src = kNR == 8 ? (((i >> 2) << 5) | (((i >> 1) & 1) << 4) | ((i & 1) << 3)) : (((i >> 1) << 5) | ((i & 1) << 4));
        su[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mean), src));
        sq[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rstd), src));
      }
    }
#ifdef SDMI_B2B_PROBE2
    B2B_STAMP(1, 5);
#endif
#pragma unroll
    for (int i = 0; i < kNR; ++i) {
      const int row = dw + 8 * i;
      const float mean = su[i], rstd = sq[i];
      if (lane == 0) { s_ln[2 * row] = mean; s_ln[2 * row + 1] = rstd; }
      if (own) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          // Full fold: the second product only ever multiplies LN0(S), so the panels hold it (the value the separate
          // LayerNorm kernel would have written).  Partial fold: they hold S, which the plain half needs; the MFMA waves
          // normalise the fragments of the folded half on the fly.
          f16x4 a = o16[i][q];
          if (!p.partial) {
            const float nmr = -mean * rstd;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = (f16)fmaf((float)a[e], rstd, nmr);      // v_fma_mix: (s - mean) rstd
          }
          // panel = 64-column group, 16-byte chunk position = chunk ^ key(row): the layout the MFMA waves read A fragments from
          const int n = q ? c1 : c0;
          *(f16x4*)(S16 + (n >> 6) * kAStage + row * 128 + (((((n >> 3) & 7) ^ ((row >> 1) & 7))) << 4) + (n & 7) * 2) = a;
        }
      }
    }
#ifdef SDMI_B2B_PROBE2
    B2B_STAMP(1, 6);
#else
    B2B_STAMP(1, 4);
#endif
    fetch(p.r2, p.r2_f32, p.h2);
    __syncthreads();                                     // B3: S16 and the statistics are written
    __syncthreads();                                     // B4: the MFMA waves hold the statistics; the ring may be refilled
#ifdef SDMI_B2B_PROBE2
    B2B_STAMP(1, 7);
#else
    B2B_STAMP(1, 5);
#endif

    // ---- product 2, epilogue 2: Y = (folded S) W2^T + h2 (+ R2); npass2 = 3: q | k | v of in_proj, 320 columns per pass ----
    for (int pass = 0; pass < p.npass2; ++pass) {
      const f16* w2 = p.w2 + (size_t)pass * kC * p.K2;
      if (pass > 0) {                                    // (the first pass's bias came with fetch())
        bv[0] = own ? *(const f32x4*)(p.h2 + pass * kC + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
        bv[1] = own ? *(const f32x4*)(p.h2 + pass * kC + c1) : f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // keep the ring's counted waits exact
      }
      issue(w2, p.K2, 0, false, 0);
      issue(w2, p.K2, 64, false, 1);
      stream(w2, p.K2, n2, false, 0);
#ifndef SDMI_B2B_PROBE2
      B2B_STAMP(1, 6);
#endif
      __syncthreads();                                   // B5
      __syncthreads();                                   // B6
      if (p.vt && pass == 2) {
        // V pass: transposed store, 8 tokens of one column per 16-byte store along the key axis in the attention kernel's
        // quad-permuted order (gemm.hip vt_pos: quads of a 16-key group stored as q0,q2,q1,q3 -> quads {hf, hf+2} adjacent)
        const int b = m0 / p.S, s0 = m0 - b * p.S;        // the tile lies inside one image (S % BM == 0)
        for (int idx = (wave_id - kMW) * 64 + lane; idx < kC * (kBM / 8); idx += kDW * 64) {
          const int col = idx % kC, piece = idx / kC;
          const int g16 = piece >> 1, hf = piece & 1;
          const float bias = p.h2[2 * kC + col];
          f16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (f16)(Cs[(g16 * 16 + hf * 4 + e) * kCsLd + col] + bias);
            o[4 + e] = (f16)(Cs[(g16 * 16 + 8 + hf * 4 + e) * kCsLd + col] + bias);
          }
          *(f16x8*)(p.vt + ((size_t)b * kC + col) * p.ldt + s0 + g16 * 16 + hf * 8) = o;
        }
      } else {
        const float cs = (p.npass2 == 1 || pass == 0) ? p.cscale : 0.f;
        const size_t cofs = (size_t)pass * kC;
        // GroupNorm statistics of the output (GnRec, common.h): this lane's two column quads stay the same over its rows.  Taken
        // from the fp16 values with the packed dot product (one v_dot2_f32_f16 per pair and moment, as the row statistics
        // above); a quad starts at a multiple of 4 and an atom is even, so a PAIR never straddles two atoms: the first pair of
        // a quad always belongs to the quad's first atom, the second pair to it or to the next (decided once, at the end)
        constexpr bool gstat = GACC;
        constexpr int HB2 = GACC ? 2 : 4;          // rows per batch of this epilogue (GACC: two, the statistics need the registers)
        float gq[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // per quad {sum pair 0, sumsq pair 0, sum pair 1, sumsq pair 1}
#pragma unroll
        for (int hb = 0; hb < kNR; hb += HB2) {
          f32x4 cv[HB2][2];
#pragma unroll
          for (int i = 0; i < HB2; ++i) {
            cv[i][0] = *(const f32x4*)(Cs + (dw + 8 * (hb + i)) * kCsLd + cs0);
            cv[i][1] = *(const f32x4*)(Cs + (dw + 8 * (hb + i)) * kCsLd + cs1);
          }
          f16x4 o[HB2][2];
#pragma unroll
          for (int i = 0; i < HB2; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float v = cv[i][q][e] + bv[q][e];
                if (cs != 0.f) v *= cs;
                v += rv[hb + i][q][e];
                cv[i][q][e] = v;
                o[i][q][e] = (f16)v;
              }
              if constexpr (gstat) {
                const f16x2 ones = f16x2{(f16)1.f, (f16)1.f};
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                  // the moments of WHAT THE GROUPNORM WILL READ (as store_tile, splitk_finalize and the stem take them): the fp32
                  // values when the stream is fp32 (gn_apply reads x.f), the fp16 ones otherwise
                  if (p.out_f32) {
                    const float xa = cv[i][q][2 * h2], xb = cv[i][q][2 * h2 + 1];
                    gq[q][2 * h2] += xa + xb;
                    gq[q][2 * h2 + 1] = fmaf(xa, xa, fmaf(xb, xb, gq[q][2 * h2 + 1]));
                  } else {
                    const f16x2 x2 = f16x2{o[i][q][2 * h2], o[i][q][2 * h2 + 1]};     // (lanes beyond the 40 owners accumulate junk they never publish)
                    gq[q][2 * h2] = __builtin_amdgcn_fdot2(x2, ones, gq[q][2 * h2], false);
                    gq[q][2 * h2 + 1] = __builtin_amdgcn_fdot2(x2, x2, gq[q][2 * h2 + 1], false);
                  }
                }
              }
            }
          if (own) {
#pragma unroll
            for (int i = 0; i < HB2; ++i) {
              const size_t mo = (size_t)(m0 + dw + 8 * (hb + i)) * p.ldo + cofs;
              if (p.out_f32) {
                *(f32x4*)((float*)p.out + mo + c0) = cv[i][0];
                *(f32x4*)((float*)p.out + mo + c1) = cv[i][1];
                if (p.out16) { *(f16x4*)(p.out16 + mo + c0) = o[i][0]; *(f16x4*)(p.out16 + mo + c1) = o[i][1]; }
              } else {
                *(f16x4*)((f16*)p.out + mo + c0) = o[i][0];
                *(f16x4*)((f16*)p.out + mo + c1) = o[i][1];
              }
            }
          }
        }
        if constexpr (gstat) {
          // one record per (DMA wave, quad) behind the fp32 tile (the ring is idle: every tile of the last product has been
          // multiplied), then lanes 0..31 of the first DMA wave add up the quads of one 10-channel atom each over the 8 waves
          // in a fixed order (fp64) and store them as this tile's record
          float* s_q = (float*)(smem + kCsBytes + 2 * kBM * 4 + 256);
          if (own) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int n = q ? c1 : c0;
              const bool both = (gnrec_div_atom(p.gacc, n) + 1) * p.gacc.atom - n >= 4;      // the quad lies inside one atom
              // record = {first atom: sum, sumsq; second atom: sum, sumsq}
              *(f32x4*)(s_q + (dw * 80 + q * 40 + lane) * 4) =
                  both ? f32x4{gq[q][0] + gq[q][2], gq[q][1] + gq[q][3], 0.f, 0.f} : f32x4{gq[q][0], gq[q][1], gq[q][2], gq[q][3]};
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (LDS only: __syncthreads() would wait for the output stores to retire)
          __builtin_amdgcn_s_barrier();                    // B8 (statistics only)
          const int atom = p.gacc.atom;
          if (dw == 0 && lane < p.gacc.natoms) {
            const int at = lane;
            const int q_lo = (at * atom) >> 2, q_hi = ((at + 1) * atom - 1) >> 2;     // the (at most 6) quads of the atom
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int qq = 0; qq < 6; ++qq) {
              const int qd = min(q_lo + qq, q_hi);
              const bool use = q_lo + qq <= q_hi;
              const int sel = (gnrec_div_atom(p.gacc, 4 * qd) == at) ? 0 : 2;
              f32x2 t[kDW];
#pragma unroll
              for (int w = 0; w < kDW; ++w) t[w] = *(const f32x2*)(s_q + (w * 80 + qd) * 4 + sel);
#pragma unroll
              for (int w = 0; w < kDW; ++w) { s1 += use ? t[w][0] : 0.f; s2 += use ? t[w][1] : 0.f; }
            }
            const int img = gnrec_div_rows(p.gacc, m0);
            const int t_row = (m0 - img * p.gacc.rows_img) / kBM;       // this workgroup's record row (parts = 1)
            ((f32x2*)p.gacc.rec)[(size_t)(img * p.gacc.T + t_row) * p.gacc.natoms + at] = f32x2{s1, s2};
          }
        }
      }
      if (pass + 1 < p.npass2) __syncthreads();          // B7: the tile is consumed; the next pass refills the ring under it
    }
#ifndef SDMI_B2B_PROBE2
    B2B_STAMP(1, 7);
#endif
  } else {
    // =============================== MFMA waves ===============================
    const int r = lane & 31, h = lane >> 5;
    const int key = (r >> 1) & 7;
    const int rb = wave_id % (kBM / 32), cg = wave_id / (kBM / 32);   // row block, column group (5 blocks of 32 columns)
    f32x16 acc[5];
    float rsd = 1.f, nmr = 0.f;                          // rstd and -mean*rstd of the row whose A fragments this lane reads
    auto zero_acc = [&]() {
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    };
    // norm: the A fragments are LayerNorm-ed on the fly, a = fp16((s - mean) rstd) in fp32 -- the value the separate
    // LayerNorm kernel would have written; gamma / beta live in the weights / bias (sdmi_launch_ln_fold_prep)
    auto compute = [&](const char* As, const char* Bs, bool norm) {
      const char* ap = As + (rb * 32 + r) * 128;
      const char* bp = Bs + (cg * 160 + r) * 128;
      f16x8 af[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) af[s] = *(const f16x8*)(ap + (((2 * s + h) ^ key) << 4));
      if (norm) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) af[s][e] = (f16)fmaf((float)af[s][e], rsd, nmr);     // v_fma_mix: (s - mean) rstd
      }
      // The 20 W fragments of a K-step (5 column blocks x 4 k16 sub-steps) go through a 6-deep register ring, each read
      // four MFMAs ahead of its use: the LDS latency hides under the matrix pipe, at 24 registers instead of the 32 of a
      // whole double-buffered block (the kernel sits at the 168-register limit of three waves per SIMD).
      f16x8 bf[6];
#pragma unroll
      for (int q = 0; q < 4; ++q) bf[q] = *(const f16x8*)(bp + (q >> 2) * 32 * 128 + (((2 * (q & 3) + h) ^ key) << 4));
#pragma unroll
      for (int q = 0; q < 20; ++q) {
        if (q + 4 < 20) {
          const int qq = q + 4;
          bf[qq % 6] = *(const f16x8*)(bp + (qq >> 2) * 32 * 128 + (((2 * (qq & 3) + h) ^ key) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[q >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[q & 3], bf[q % 6], acc[q >> 2], 0, 0, 0);
      }
    };
    auto acc_to_lds = [&]() {
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          Cs[row * kCsLd + (cg * 5 + j) * 32 + r] = acc[j][e];
        }
    };

    zero_acc();
    B2B_STAMP(0, 0);
    for (int t = 0; t < kC / 64; ++t) {
      __builtin_amdgcn_s_barrier();
      if (t == 0) B2B_STAMP(0, 1);
      const int st = t % kNS;
      compute(p.gx ? S16 + t * kAStage : S16 + st * kAStage, ring + st * kWStage, false);   // panels (GroupNorm input) or A1 ring
    }
    B2B_STAMP(0, 2);
    __syncthreads();                                     // B1
    acc_to_lds();
    __syncthreads();                                     // B2
    zero_acc();
    __syncthreads();                                     // B3
    rsd = s_ln[2 * (rb * 32 + r) + 1];
    nmr = -s_ln[2 * (rb * 32 + r)] * rsd;
    __syncthreads();                                     // B4
    B2B_STAMP(0, 3);
    const int n_norm = p.partial ? kC / 64 : 0;          // K-steps whose A fragments are normalised here (partial fold only)
    for (int pass = 0; pass < p.npass2; ++pass) {
      if (pass > 0) zero_acc();
      for (int t = 0; t < n2; ++t) {
        __builtin_amdgcn_s_barrier();
        if (t == 0) B2B_STAMP(0, 4);
        compute(S16 + (t % 5) * kAStage, ring + (t % kNS) * kWStage, t < n_norm);
      }
      B2B_STAMP(0, 5);
      __syncthreads();                                   // B5
      acc_to_lds();
      __syncthreads();                                   // B6
      if (pass + 1 < p.npass2) __syncthreads();          // B7
    }
    if constexpr (GACC) __builtin_amdgcn_s_barrier();    // B8 (statistics only)
    B2B_STAMP(0, 6);
  }
}

template <int BM>
int launch_b2b(const B2bArgs& a, hipStream_t st) {
  typedef B2bCfg<BM> Cf;
  static bool attr_done[16] = {};            // the dynamic-LDS attribute is per device
  int dev = 0;
  SDMI_CHECK_HIP(hipGetDevice(&dev));
  SDMI_REQUIRE(dev >= 0 && dev < 16, "b2b: device index %d out of range", dev);
  if (!attr_done[dev]) {
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)b2b_kernel<Cf, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::kLds));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)b2b_kernel<Cf, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::kLds));
    attr_done[dev] = true;
  }
  if (a.gacc.rec) {
    hipLaunchKernelGGL((b2b_kernel<Cf, true>), dim3(a.M / BM), dim3(Cf::kNT), Cf::kLds, st, a);
    SDMI_CHECK_HIP(hipGetLastError());
    return SDMI_OK;
  }
  hipLaunchKernelGGL((b2b_kernel<Cf, false>), dim3(a.M / BM), dim3(Cf::kNT), Cf::kLds, st, a);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

}  // namespace

#ifdef SDMI_B2B_PROBE
extern "C" int sdmi_dbg_read_b2b(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_b2b_clk), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -5;
}
#endif

// bm: 32 or 64 rows per workgroup; 0 = 32 while that is at most one workgroup per CU (the 64x64 level: 256 of them; measured
// 15.6 / 19.1 us against 18.9 / 22.1 us with 64 rows at M = 8192, but 39.5 / 49.7 against 36.1 / 44.3 us at M = 18432)
int sdmi_launch_b2b(const B2bArgs& a_in, hipStream_t st, int bm) {
  B2bArgs a = a_in;
  gnrec_magic(a.gacc);
  SDMI_REQUIRE(a.M > 0 && a.M % 32 == 0, "b2b: M=%d must be a positive multiple of 32", a.M);
  SDMI_REQUIRE(a.K2 == kC || (a.K2 == 2 * kC && a.partial), "b2b: K2=%d (C = %d: K2 = C, or 2C with the partial fold)", a.K2, kC);
  SDMI_REQUIRE(!a.partial || a.K2 == 2 * kC, "b2b: the partial fold needs K2 = 2C");
  SDMI_REQUIRE((a.a1 || a.gx) && a.w1 && a.b1 && a.w2 && a.h2 && a.out && (a.gx || (a.lda1 >= kC && a.lda1 % 8 == 0)), "b2b: null pointer / bad lda");
  SDMI_REQUIRE(!a.gx || (a.gn_partial && a.gn_gamma && a.gn_beta && a.gn_nchunk > 0 && a.S > 0 && a.S % 32 == 0 && a.M % a.S == 0),
               "b2b: the GroupNorm input form needs the statistics partials, gamma / beta and whole 32-row tiles per image");
  SDMI_REQUIRE(a.npass2 == 1 || (a.npass2 == 3 && !a.partial && !a.r2 && !a.out_f32 && a.vt && a.S > 0 && a.S % 32 == 0 && a.M % a.S == 0 &&
                                 a.ldt >= a.S && a.ldt % 8 == 0),
               "b2b: the three-pass form (q | k | v) needs fp16 outputs, no residual, a V^T target and S %% 32 == 0");
  SDMI_REQUIRE(a.ldo >= a.npass2 * kC - (a.npass2 == 3 ? kC : 0) && a.ldo % 8 == 0, "b2b: output row stride %d", a.ldo);
  if (bm == 0) bm = sdmi_b2b_tile_rows(a);
  if (a.gx && a.S % 64 != 0) bm = 32;
  SDMI_REQUIRE((a.npass2 == 1 && !a.gx) || a.S % bm == 0, "b2b: a %d-row tile would straddle images of %d tokens", bm, a.S);
  SDMI_REQUIRE((bm == 32 || bm == 64) && a.M % bm == 0, "b2b: tile height %d does not divide M=%d", bm, a.M);
  // (round 5: the 64-row form takes the statistics too -- 165 registers, no spill -- so the batched multi-prompt mode, whose M
  // runs in whole rounds of 64-row workgroups, keeps the statistics chain)
  SDMI_REQUIRE(!a.gacc.rec || (a.npass2 == 1 && a.gacc.atom >= 4 && a.gacc.atom <= 16 && a.gacc.atom % 2 == 0 && a.gacc.natoms * a.gacc.atom == kC && a.gacc.natoms <= 32 &&
                               a.gacc.rows_img % bm == 0 && a.M % a.gacc.rows_img == 0 && a.gacc.parts == 1 && a.gacc.T == a.gacc.rows_img / bm),
               "b2b: GroupNorm statistics need the one-pass form, even atoms that tile the 320 columns, whole %d-row tiles per image and T = rows / %d, parts = 1", bm, bm);
  return bm == 32 ? launch_b2b<32>(a, st) : launch_b2b<64>(a, st);
}

// rows per workgroup sdmi_launch_b2b picks for bm = 0 (the statistics' record rows follow it: GnRec::T = rows per image / this)
int sdmi_b2b_tile_rows(const B2bArgs& a) {
  return (a.M % 64 != 0 || a.M / 32 <= 256 || (a.npass2 == 3 && a.S % 64 != 0)) ? 32 : 64;
}
