// Flash-style multi-head attention for gfx950 (MI355X): softmax(Q K^T / sqrt(d)) V without
// materialising the S x S scores.  Replaces the materialised-score attention of the reference
// (sd/attention.py:55-86 self-attention, :219-244 cross-attention) for the UNet's head sizes
// d in {40, 80, 160} (8 heads; sd/diffusion.py:543-626).
//
// Design (one workgroup = 4 waves = 128 queries of one (batch, head); 64-key K/V tiles):
//   * S^T = K Q^T with v_mfma_f32_32x32x16_f16: the query index lands on the LANE and the key index
//     in the accumulator registers, so the online-softmax row max/sum is register-local plus one
//     lane^32 exchange, and the exponentiated tile P^T is ALREADY laid out as the B operand of the
//     next product (accumulator-as-operand; no LDS round trip for P).
//   * O^T += V^T P^T: V arrives pre-transposed ([b][h*d+dd][key], written by the projection GEMM's
//     transposed epilogue), so both K and V^T tiles are plain row gathers staged by LDS-DMA.
//   * K tile rows are d*2 bytes (80-byte stride is bank-conflict-free for d=40, the S=4096 case that
//     carries 88% of attention FLOPs); V^T tile rows are 128 B with the GEMM's XOR swizzle.
//   * the head-dim contraction is zero-padded on the Q fragment only (d=40 -> 48).
//   * keys >= Skv are masked (cross-attention: 77 keys in a 128-key double tile).
//   * SPLIT (short sequences, Sq <= 1024): a workgroup takes 64 queries and its wave pairs split the KEYS in two halves,
//     each wave running the online softmax over its half; the pair merges (max, sum, O) through LDS at the end.  The
//     tile loop is a dependent chain (~1 us per 64-key tile), so at S = 1024 / 256 the kernel was latency-bound with
//     128 / 32 workgroups on 256 CUs: halving the chain and doubling the workgroups is worth more than the merge costs.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

// Diagnostic builds (tools/build_variant.sh ... -DSDMI_ATTN_ABLATE=<bits>): drop one component of the tile loop to read its cost
// off the kernel time (results are then wrong): 1 exp2 -> move, 2 three quarters of the PV MFMAs + V reads, 4 two thirds of the
// QK^T MFMAs + K reads, 8 no K/V staging after the first tile, 16 no per-tile wait + barrier (use with 8)
#ifndef SDMI_ATTN_ABLATE
#define SDMI_ATTN_ABLATE 0
#endif

#ifdef SDMI_ATTN_PROBE
// diagnostic build (tools/build_variant.sh probe attention -DSDMI_ATTN_PROBE): shader-clock cycles every wave of the first 512 workgroups
// spends in the parts of its tile loop, summed over the tiles: {stage issue, tile body, -, DMA wait, barrier, tiles, whole life, life in 100 MHz ticks}
__device__ unsigned long long g_attn_clk[4096][8];   // [6], [7]: shader clocks and 100 MHz ticks of the wave's whole life
extern "C" int sdmi_dbg_read_attn(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_clk), (size_t)n * 64) == hipSuccess ? 0 : -5;
}
#define ATTN_T() __builtin_amdgcn_s_memtime()
#endif

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

template <int D>
struct ACfg {
  static constexpr int NS = (D + 15) / 16;          // k16 steps of the QK^T contraction
  static constexpr int DP = ((D + 31) / 32) * 32;   // padded head dim (rows of V^T tile)
  static constexpr int DB = DP / 32;
  static constexpr int KCH = D / 8;                 // 16-B chunks per K row
  static constexpr int K_BYTES = 64 * D * 2;
  static constexpr int K_INST = (64 * KCH) / 64;    // = KCH wave-instructions per tile
  static constexpr int V_BYTES = DP * 128;
  static constexpr int V_INST = DP / 8;
  static constexpr int STAGE = K_BYTES + V_BYTES;
  static constexpr bool LROW = DP > D;             // a spare V^T row exists (d = 40, 80): row sums come from the MFMA
  // d = 40: the last k16 step of QK^T covers head-dim columns 32..47, of which 40..47 are padding (zeros on the Q side).
  // The running-max bias (K side 1, Q side -m_run) rides in column 40 instead of a k16 step of its own: 6 instead of 8
  // MFMAs per 64-key tile in the S = 4096 self-attention.
  static constexpr bool SPARE = (D % 16) != 0;
  static constexpr int LDS = 2 * STAGE;
  static constexpr int EXTRA = SPARE ? 16 : 0;     // the K-side bias fragment behind the stages
};

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate must be a constant): n LDS-DMA instructions may stay in flight
__device__ __forceinline__ void wait_vm_upto(int n) {
#define SDMI_VM_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    SDMI_VM_CASE(1) SDMI_VM_CASE(2) SDMI_VM_CASE(3) SDMI_VM_CASE(4) SDMI_VM_CASE(5) SDMI_VM_CASE(6) SDMI_VM_CASE(7) SDMI_VM_CASE(8)
    SDMI_VM_CASE(9) SDMI_VM_CASE(10) SDMI_VM_CASE(11) SDMI_VM_CASE(12) SDMI_VM_CASE(13) SDMI_VM_CASE(14) SDMI_VM_CASE(15) SDMI_VM_CASE(16)
    SDMI_VM_CASE(17) SDMI_VM_CASE(18) SDMI_VM_CASE(19) SDMI_VM_CASE(20) SDMI_VM_CASE(21) SDMI_VM_CASE(22) SDMI_VM_CASE(23) SDMI_VM_CASE(24)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef SDMI_VM_CASE
}

// The end of every form: (SPLIT) the wave pair's two key halves meet through LDS, then O / l is written (fp16, and fp32 for the
// accurate mode).  `smem` is the start of the dynamic LDS: the K/V stages are dead by now and hold the exchange.
template <int D, bool SPLIT>
__device__ __forceinline__ void attn_finish(const AttnArgs& p, char* smem, f32x16 (&oacc)[ACfg<D>::DB], float m_run, float l_run,
                                            int wave, int lane, int hk, int q0, int b, int head) {
  typedef ACfg<D> C;
  const int r = lane & 31, h = lane >> 5;
  if constexpr (SPLIT) {
    // merge the pair's halves: wave hk = 1 hands (m, l, O^T) over through LDS (aliasing the K/V stages: every tile has been
    // consumed), wave hk = 0 combines: m = max, O = O_a 2^(m_a - m) + O_b 2^(m_b - m) (the row of ones in V^T carries l
    // along inside O for d = 40 / 80; d = 160 merges l_run explicitly), then normalises and stores as usual.
    constexpr int NV = C::DB * 16 + 2;
    __syncthreads();
    float* mx = (float*)smem + ((size_t)(wave >> 1) * 64 + lane) * NV;
    if (hk == 1) {
#pragma unroll
      for (int d = 0; d < C::DB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx[d * 16 + e] = oacc[d][e];
      mx[C::DB * 16] = m_run;
      mx[C::DB * 16 + 1] = l_run;
    }
    __syncthreads();
    if (hk == 1) return;
    const float m_b = mx[C::DB * 16], l_b = mx[C::DB * 16 + 1];
    const float m = fmaxf(m_run, m_b);
    const float sa = __builtin_amdgcn_exp2f(m_run - m), sb = __builtin_amdgcn_exp2f(m_b - m);
#pragma unroll
    for (int d = 0; d < C::DB; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[d][e] = oacc[d][e] * sa + mx[d * 16 + e] * sb;
    l_run = l_run * sa + l_b * sb;
  }
  // ---- normalise and store: lane = query row, registers = head-dim ----
  float l_tot;
  if constexpr (C::LROW) l_tot = __shfl(oacc[D / 32][4 * ((D % 32) / 8)], r);   // accumulator row D (lane r of the low half)
  else l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;
  const int qi = q0 + r;
  if (qi < p.Sq) {
    f16* op = p.o + ((size_t)b * p.Sq + qi) * p.ldo + head * D;
#pragma unroll
    for (int d = 0; d < C::DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = d * 32 + 8 * g + 4 * h;
        if (dd < D) {
          f16x4 o4;
          f32x4 of;
#pragma unroll
          for (int e = 0; e < 4; ++e) { of[e] = oacc[d][4 * g + e] * inv; o4[e] = (f16)of[e]; }
          *(f16x4*)(op + dd) = o4;
          if (p.o32) *(f32x4*)(p.o32 + ((size_t)b * p.Sq + qi) * p.ldo + head * D + dd) = of;
        }
      }
  }
}

// NW = waves per workgroup: 4, or 8 for the key-split form over 128 queries (long sequences: four waves per SIMD at the
// same K/V bytes staged per query as the unsplit form).
// NB = depth of the K/V ring in LDS.  2: the tiles of step t+1 are requested when step t starts and waited for (vmcnt(0)) when it
// ends.  3: the tiles of step t+2 are requested when step t starts, and the wait at the end of step t is a COUNTED one for step
// t+1's tiles only (requested a whole step earlier) -- the newest stage stays in flight across the barrier.
// Grid: ONE dimension, remapped so that every XCD owns a contiguous range of (batch*head, query tile) ids: workgroups are dealt
// round-robin to the 8 XCDs, so with the plain (query tile, batch*head) grid the 32 query tiles of one head landed on all
// eight L2s and every L2 streamed every head's K and V (measured: 89 MB fetched per S = 4096 launch for 15.7 MB of Q/K/V).
// With two heads per XCD their K/V (1.3 MB) stay in that XCD's L2.  Speed only: any placement gives the same result.
template <int D, bool SPLIT, int NW = 4, int NB = 2>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 1) void attn_kernel(AttnArgs p) {
  sdmi_kernarg_warm<sizeof(AttnArgs) + 16>();     // + the hidden grid size (gridDim.x)
  typedef ACfg<D> C;
  static_assert(NW == 4 || (SPLIT && NW == 8), "8-wave workgroups only in the key-split form");
  static_assert(NB == 2 || NB == 3, "ring depth");
  constexpr int NSUB = SPLIT ? 2 : 1;                 // key streams staged per step (SPLIT: one tile of each half)
  constexpr int STAGE2 = NSUB * C::STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef SDMI_ATTN_PROBE
  const unsigned long long life_c0 = ATTN_T(), life_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  int qtile, bh;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qq = nwg >> 3, rr = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int L = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;      // bijective (cdna guide T1)
    const int ntq = SPLIT ? p.Sq / (NW * 16) : (p.Sq + 127) >> 7;
    bh = L / ntq;
    qtile = L - bh * ntq;
  }
  const int b = bh / p.H, head = bh % p.H;
  const int hk = SPLIT ? (wave & 1) : 0;              // which half of the keys this wave reduces
  const int q0 = SPLIT ? qtile * (NW * 16) + (wave >> 1) * 32 : qtile * 128 + wave * 32;
  const int ntiles = SPLIT ? (p.Skv >> 7) : (p.Skv + 63) >> 6;   // tiles per stream (SPLIT: Skv % 128 == 0, launcher)

  const f16* kbase = p.k + (size_t)b * p.k_batch_stride * p.ldk + head * D;
  const f16* vbase = p.vt + ((size_t)(b * p.H + head) * D) * p.ldvt;

  // per-lane LDS-DMA source pointers, advanced by one 64-key tile per stage() call
  constexpr int KI = (C::K_INST + NW - 1) / NW, VI = (C::V_INST + NW - 1) / NW;
  const f16* kptr[NSUB][KI];
  int krow[KI];
  const f16* vptr[NSUB][VI];
  int vinc[VI];
  const int half_keys = ntiles * 64;                   // SPLIT: first key of the second stream
#pragma unroll
  for (int i = 0; i < KI; ++i) {
    const int q = (i * NW + wave) * 64 + lane;
    const int row = q / C::KCH, c = q - row * C::KCH;
    krow[i] = row;
#pragma unroll
    for (int u = 0; u < NSUB; ++u) kptr[u][i] = kbase + (size_t)(u * half_keys + row) * p.ldk + c * 8;
  }
#pragma unroll
  for (int i = 0; i < VI; ++i) {
    const int q = (i * NW + wave) * 64 + lane;
    const int row = q >> 3, pc = q & 7;
    const int gch = pc ^ ((row >> 1) & 7);
    // row D of the padded V^T tile is all ones: the PV MFMA then also produces the row sums of P (the softmax
    // denominator) in accumulator row D, and the VALU never adds the probabilities up
#pragma unroll
    for (int u = 0; u < NSUB; ++u)
      vptr[u][i] = row < D ? vbase + (size_t)row * p.ldvt + u * half_keys + gch * 8 : ((C::LROW && row == D) ? p.ones : p.zero) + gch * 8;
    vinc[i] = row < D ? 64 : 0;
  }
  const size_t kstep = (size_t)64 * p.ldk;
  int stage_key0 = 0;
  // LDS-DMA instructions THIS wave issues per stage() call (wave-uniform; the waves differ where K_INST / V_INST % NW != 0)
  int per_stage = 0;
#pragma unroll
  for (int i = 0; i < KI; ++i) per_stage += (i * NW + wave < C::K_INST) ? NSUB : 0;
#pragma unroll
  for (int i = 0; i < VI; ++i) per_stage += (C::V_INST % NW == 0 || i * NW + wave < C::V_INST) ? NSUB : 0;

  auto stage = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
      char* sk = smem + buf * STAGE2 + u * C::STAGE;
      char* sv = sk + C::K_BYTES;
#pragma unroll
      for (int i = 0; i < KI; ++i) {
        const int ii = i * NW + wave;
        if (ii < C::K_INST) {
          const f16* g = (SPLIT || stage_key0 + krow[i] < p.Skv) ? kptr[u][i] : p.zero;
          glds16(g, sk + ii * 1024);
          kptr[u][i] += kstep;
        }
      }
#pragma unroll
      for (int i = 0; i < VI; ++i) {
        if (C::V_INST % NW == 0 || i * NW + wave < C::V_INST) {
          glds16(vptr[u][i], sv + (i * NW + wave) * 1024);
          vptr[u][i] += vinc[i];
        }
      }
    }
    stage_key0 += 64;
  };

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q = r, half h) holds dk = 16 s + 8 h + j ----
  f16x8 qf[C::NS];
  {
    int qi = q0 + r;
    if (qi >= p.Sq) qi = p.Sq - 1;
    const f16* qp = p.q + ((size_t)b * p.Sq + qi) * p.ldq + head * D;
#pragma unroll
    for (int s = 0; s < C::NS; ++s) {
      const int dk = 16 * s + 8 * h;
      if (dk < D) {
        qf[s] = *(const f16x8*)(qp + dk);
        if (!p.prescaled) {          // op-level callers hand over plain q: scale here (one extra fp16 rounding)
#pragma unroll
          for (int e = 0; e < 8; ++e) qf[s][e] = (f16)((float)qf[s][e] * (p.scale * 1.4426950408889634f));
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = (f16)0.f;
      }
    }
  }

  // K fragment byte offsets within a K tile row (clamped to the row start where Q is zero-padded)
  int koff[C::NS];
#pragma unroll
  for (int s = 0; s < C::NS; ++s) {
    const int dk = 16 * s + 8 * h;
    koff[s] = dk < D ? dk * 2 : 0;
  }
  const int vkey = (r >> 1) & 7;

  f32x16 oacc[C::DB];
#pragma unroll
  for (int d = 0; d < C::DB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
  // Scores arrive in the log2 domain (Q carries scale*log2 e).  The running max m_run is subtracted INSIDE the QK^T
  // chain by one extra k16 step: K side = 1 at k-index 0, Q side = -m_run (kept exactly representable in fp16), so the
  // softmax is p = exp2(acc) with no per-score multiply or subtract on the VALU.
  float m_run = 0.f, l_run = 0.f;
  f16x8 kbias, qbias;
#pragma unroll
  for (int e = 0; e < 8; ++e) { kbias[e] = (f16)0.f; qbias[e] = (f16)0.f; }
  // SPARE: k-index 40 is element 0 of the half-wave h = 1 in the last k16 step; otherwise k-index 0 of the extra step (h = 0)
  if (h == (C::SPARE ? 1 : 0)) kbias[0] = (f16)1.f;
  // SPARE: the h = 1 lanes of the last k16 step read their K fragment {1, 0, ..., 0} from 16 bytes behind the K/V stages (one
  // select on the LDS address instead of four on the loaded registers); visible after the first __syncthreads()
  const char* const kbias_lds = smem + NB * STAGE2;
  if (C::SPARE && tid == 0) {
    f16x8 one;
#pragma unroll
    for (int e = 0; e < 8; ++e) one[e] = (f16)0.f;
    one[0] = (f16)1.f;
    *(f16x8*)(smem + NB * STAGE2) = one;
  }

  // one 64-key tile: S'^T = K Q'^T - m_run, online softmax, O^T += V^T P^T.  FIRST: tile 0 (m_run not set yet).
  auto tile_body = [&](auto masked_tag, auto first_tag, int t, int cur) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    const char* Ks = smem + cur * STAGE2 + hk * C::STAGE;
    const char* Vs = Ks + C::K_BYTES;
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.f;
#pragma unroll
      for (int s = 0; s < ((SDMI_ATTN_ABLATE & 4) ? 1 : C::NS); ++s) {
        const char* ka = Ks + (kb * 32 + r) * (D * 2) + koff[s];
        if (C::SPARE && s == C::NS - 1) ka = h == 1 ? kbias_lds : ka;       // padding columns of K: {1, 0, ...} against Q's {-m_run, 0, ...}
        const f16x8 kf = *(const f16x8*)ka;
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc[kb], 0, 0, 0);
      }
      if constexpr (!FIRST && !C::SPARE) sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kbias, qbias, sacc[kb], 0, 0, 0);
    }
    if constexpr (MASKED) {   // keys beyond Skv (ragged last tile) and, for causal attention, keys after the query
      const int kbase_i = t * 64 + 4 * h;
      const int qlim = p.causal ? q0 + r : 0x7fffffff;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kbase_i + kb * 32 + (e & 3) + 8 * (e >> 2);
          if (key >= p.Skv || key > qlim) sacc[kb][e] = -INFINITY;
        }
    }
    // online softmax (per query = per lane; the two half-waves hold disjoint keys).  Lazy rescale: the running max is
    // raised (and O, l rescaled) only when some query's tile max exceeds it by more than 2^8; otherwise P <= 256
    // (fine for fp16 P, fp32 l/O).  Wave-uniform branch.
    float mx0 = fmaxf(fmaxf(sacc[0][0], sacc[0][1]), sacc[0][2]);
    float mx1 = fmaxf(fmaxf(sacc[1][0], sacc[1][1]), sacc[1][2]);
#pragma unroll
    for (int e = 3; e < 15; e += 2) {
      mx0 = fmaxf(fmaxf(mx0, sacc[0][e]), sacc[0][e + 1]);
      mx1 = fmaxf(fmaxf(mx1, sacc[1][e]), sacc[1][e + 1]);
    }
    float mx = fmaxf(fmaxf(mx0, sacc[0][15]), fmaxf(mx1, sacc[1][15]));
    {   // lane ^ 32 exchange on the VALU (v_permlane32_swap) instead of an LDS round trip (ds_bpermute + wait, ~120 cycles of
        // the tile's dependent chain): swapping a register with a copy of itself leaves {lower half, lower half} in one and
        // {upper, upper} in the other
      const unsigned mb = __float_as_uint(mx);
      const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (FIRST || __builtin_expect(__any(mx > 8.f), 0)) {
      float d = FIRST ? mx : fmaxf(mx, 0.f);
      if (!(d > -60000.f)) d = 0.f;                       // a query with no valid key in this tile keeps its max
      const float m_new = (float)(f16)(m_run + d);          // exactly representable: it re-enters through qbias
      const float de = m_new - m_run;
      if constexpr (!FIRST) {
        const float alpha = __builtin_amdgcn_exp2f(-de);
        l_run *= alpha;
#pragma unroll
        for (int dd = 0; dd < C::DB; ++dd)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[dd][e] *= alpha;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[kb][e] -= de;
      m_run = m_new;
      if constexpr (C::SPARE) { if (h == 1) qf[C::NS - 1][0] = (f16)(-m_run); }
      else qbias[0] = h == 0 ? (f16)(-m_run) : (f16)0.f;
    }
    float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p0 = (SDMI_ATTN_ABLATE & 1) ? sacc[0][e] : __builtin_amdgcn_exp2f(sacc[0][e]);
      const float p1 = (SDMI_ATTN_ABLATE & 1) ? sacc[1][e] : __builtin_amdgcn_exp2f(sacc[1][e]);
      sacc[0][e] = p0;
      sacc[1][e] = p1;
      if constexpr (!C::LROW) { ps0 += p0; ps1 += p1; }
    }
    if constexpr (!C::LROW) l_run += ps0 + ps1;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (f16)sacc[kb][8 * s2 + e];
        // keys of this k16 step held by half h: quads h and 2+h of 16-key group 2*kb + s2 -- adjacent in the
        // permuted V^T storage (gemm.hip vt_pos): one 16-byte read at chunk 2*group + h
        const int o0 = (((4 * kb + 2 * s2 + h) ^ vkey) << 4);
        if ((SDMI_ATTN_ABLATE & 2) && (kb | s2)) { asm volatile("" ::"v"(pf)); continue; }
#pragma unroll
        for (int d = 0; d < C::DB; ++d) {
          const f16x8 vf = *(const f16x8*)(Vs + (d * 32 + r) * 128 + o0);
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[d], 0, 0, 0);
        }
      }
  };

  stage(0);
  if (NB == 3 && ntiles > 1) {
    stage(1);
    wait_vm_upto(per_stage);              // tile 0 has landed, tile 1 may still fly
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the K-side bias fragment written above
  __builtin_amdgcn_s_barrier();                           // (a raw barrier: __syncthreads() would drain the second stage)
  const bool ragged = !SPLIT && (p.Skv & 63) != 0;
  const int nfull = (!SPLIT && p.causal) ? 0 : (ragged ? ntiles - 1 : ntiles);   // tiles that need no masking
  // Three runs of tiles, each with ONE body in its loop (separate loops keep the accumulators in place: with both
  // bodies in one loop the register allocator copied all of O^T every tile): tile 0, unmasked tiles, masked tail.
  int cur = 0, nxt = NB - 1;
#ifdef SDMI_ATTN_PROBE
  unsigned long long pa[6] = {0, 0, 0, 0, 0, 0};
#endif
  auto step = [&](int t, auto masked_tag, auto first_tag) {
    // the buffer written here was last read in step t-1 (NB = 3: it holds tile t+2, NB = 2: tile t+1); the barrier that ended
    // step t-1 closed that WAR window
#ifdef SDMI_ATTN_PROBE
    const unsigned long long c0 = ATTN_T();
#endif
    if (!(SDMI_ATTN_ABLATE & 8) && t + NB - 1 < ntiles) stage(nxt);
#ifdef SDMI_ATTN_PROBE
    const unsigned long long c1 = ATTN_T();
#endif
    tile_body(masked_tag, first_tag, t, cur);
#ifdef SDMI_ATTN_PROBE
    const unsigned long long c2 = ATTN_T();
    unsigned long long c3 = c2, c4 = c2;
#endif
    if (!(SDMI_ATTN_ABLATE & 16) && t + 1 < ntiles) {
      // RAW for step t+1: every wave waits for ITS pieces of tile t+1, then the barrier publishes them.  NB = 3: the stage
      // issued in this step (tile t+2) stays in flight -- a plain s_barrier, not __syncthreads(), whose fence would drain it
      if (NB == 3 && t + 2 < ntiles) wait_vm_upto(per_stage);
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SDMI_ATTN_PROBE
      c3 = ATTN_T();
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#ifdef SDMI_ATTN_PROBE
      c4 = ATTN_T();
#endif
    }
#ifdef SDMI_ATTN_PROBE
    pa[0] += c1 - c0; pa[1] += c2 - c1; pa[3] += c3 - c2; pa[4] += c4 - c3; pa[5] += 1;
#endif
    cur = cur + 1 == NB ? 0 : cur + 1;
    nxt = nxt + 1 == NB ? 0 : nxt + 1;
  };
  if (nfull > 0) step(0, std::false_type{}, std::true_type{});
  else step(0, std::true_type{}, std::true_type{});
  int t = 1;
  for (; t < nfull; ++t) step(t, std::false_type{}, std::false_type{});
  for (; t < ntiles; ++t) step(t, std::true_type{}, std::false_type{});
#ifdef SDMI_ATTN_PROBE
  if (lane == 0 && blockIdx.x < 512 && NW == 8) {
    unsigned long long* o = g_attn_clk[blockIdx.x * 8 + wave];
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = pa[i];
  }
#endif

#ifdef SDMI_ATTN_PROBE
  if (lane == 0 && blockIdx.x < 512 && NW == 8) { g_attn_clk[blockIdx.x * 8 + wave][6] = ATTN_T() - life_c0; g_attn_clk[blockIdx.x * 8 + wave][7] = __builtin_amdgcn_s_memrealtime() - life_r0; }
#endif
  attn_finish<D, SPLIT>(p, smem, oacc, m_run, l_run, wave, lane, hk, q0, b, head);
}

template <int D>
int launch(const AttnArgs& a, hipStream_t st) {
  typedef ACfg<D> C;
  static bool attr_done[16] = {};            // the dynamic-LDS attribute is per device
  int dev = 0;
  SDMI_CHECK_HIP(hipGetDevice(&dev));
  SDMI_REQUIRE(dev >= 0 && dev < 16, "attention: device index %d out of range", dev);
  // ring depth 3 fits where the LDS allows it at the occupancy the form is built for: d = 40 (two 8-wave workgroups per CU:
  // 78 KB each) and d = 80 split (one workgroup per CU either way); d = 160 split fills the LDS with two stages.  Measured
  // (round 4, one MI355X, back-to-back launches): S = 4096 d = 40 78.0 us against 77.1 with two stages, S = 1024 d = 80 16.1
  // against 15.0, whole step 3.992 vs 3.994 ms -- the K/V tiles are L2-resident and already land under the previous tile's
  // MFMAs, a third stage only costs LDS.  So two stages stay the default; SDMI_ATTN_NB=3 selects the deeper ring.
  static const bool nb3 = getenv("SDMI_ATTN_NB") && atoi(getenv("SDMI_ATTN_NB")) == 3;
  constexpr bool CAN3 = 3 * 2 * C::STAGE + C::EXTRA <= (D == 40 ? 80 * 1024 : 160 * 1024);
  if (!attr_done[dev]) {
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_kernel<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS + C::EXTRA));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_kernel<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C::LDS + C::EXTRA));
    if constexpr (CAN3)
      SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_kernel<D, true, 4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * C::LDS + C::EXTRA));
    if constexpr (D == 40) {
      SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_kernel<D, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C::LDS + C::EXTRA));
      SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_kernel<D, true, 8, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * C::LDS + C::EXTRA));
    }
    attr_done[dev] = true;
  }
  // key-split form for short self-attention (see the header): both halves whole 64-key tiles, whole 64-query workgroups
  static const bool split_on = !(getenv("SDMI_ATTN_SPLIT") && atoi(getenv("SDMI_ATTN_SPLIT")) == 0);
  // (at S = 4096 the split form stages four times the K/V bytes per query and loses: attention 0.646 vs 0.617 ms/step)
  static const int split_max = getenv("SDMI_ATTN_SPLIT_MAXS") ? atoi(getenv("SDMI_ATTN_SPLIT_MAXS")) : 1024;
  // long sequences (d = 40, S = 4096): 128-query workgroups of EIGHT waves, the wave pairs again splitting the keys: the K/V
  // bytes staged per query equal the unsplit form's, and four waves per SIMD hide each other's dependent tile chains
  static const int split8_min = getenv("SDMI_ATTN_SPLIT8_MINS") ? atoi(getenv("SDMI_ATTN_SPLIT8_MINS")) : 2048;
  // all grids are one-dimensional: (query tiles) x (batch * heads) ids, decoded XCD-aware inside the kernel
  const int BH = a.B * a.H;
  if (D == 40 && split_on && !a.causal && a.Skv % 128 == 0 && a.Sq % 128 == 0 && a.Sq >= split8_min) {
    if constexpr (D == 40) {
      dim3 grid(a.Sq / 128 * BH);
      if (nb3) hipLaunchKernelGGL((attn_kernel<D, true, 8, 3>), grid, dim3(512), 3 * C::LDS + C::EXTRA, st, a);
      else hipLaunchKernelGGL((attn_kernel<D, true, 8>), grid, dim3(512), 2 * C::LDS + C::EXTRA, st, a);
    }
  } else if (split_on && !a.causal && a.Skv % 128 == 0 && a.Sq % 64 == 0 && a.Sq <= split_max && 2 * C::LDS + C::EXTRA <= 160 * 1024) {
    dim3 grid(a.Sq / 64 * BH);
    if constexpr (CAN3) {
      if (nb3) hipLaunchKernelGGL((attn_kernel<D, true, 4, 3>), grid, dim3(256), 3 * C::LDS + C::EXTRA, st, a);
      else hipLaunchKernelGGL((attn_kernel<D, true>), grid, dim3(256), 2 * C::LDS + C::EXTRA, st, a);
    } else {
      hipLaunchKernelGGL((attn_kernel<D, true>), grid, dim3(256), 2 * C::LDS + C::EXTRA, st, a);
    }
  } else {
    dim3 grid((a.Sq + 127) / 128 * BH);
    // diagnostic knob (A/B only): extra dynamic LDS per workgroup caps the workgroups per CU, i.e. the waves per SIMD
    static const int lds_pad = getenv("SDMI_ATTN_LDS_PAD") ? atoi(getenv("SDMI_ATTN_LDS_PAD")) : 0;
    if (lds_pad > 0) {
      static bool pad_attr = false;
      if (!pad_attr) { SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_kernel<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); pad_attr = true; }
    }
    hipLaunchKernelGGL((attn_kernel<D, false>), grid, dim3(256), C::LDS + C::EXTRA + lds_pad, st, a);
  }
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

}  // namespace

int sdmi_launch_attention(const AttnArgs& a, hipStream_t st) {
  SDMI_REQUIRE(a.q && a.k && a.vt && a.o && a.zero && a.ones, "attention: null pointer");
  SDMI_REQUIRE(a.Sq > 0 && a.Skv > 0 && a.B > 0 && a.H > 0, "attention: bad sizes");
  SDMI_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldvt % 8 == 0 && a.ldo % 4 == 0, "attention: strides must be 16-B aligned");
  SDMI_REQUIRE(a.ldvt >= ((a.Skv + 63) / 64) * 64, "attention: V^T rows must be padded to a multiple of 64 keys (ldvt=%d, Skv=%d)", a.ldvt, a.Skv);
  switch (a.d) {
    case 40: return launch<40>(a, st);
    case 80: return launch<80>(a, st);
    case 160: return launch<160>(a, st);
    case 64: return launch<64>(a, st);
    default: sdmi_set_error("attention: unsupported head dim %d", a.d); return SDMI_EINVAL;
  }
}
