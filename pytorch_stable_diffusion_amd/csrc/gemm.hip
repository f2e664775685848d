// Implicit-GEMM convolution / linear kernel for gfx950 (MI355X).
//
//   D[m][n] = sum_k A(m,k) * W[n][k]      fp16 operands, fp32 MFMA accumulation
//
// Replaces every nn.Conv2d (3x3 s1/s2 and 1x1) and nn.Linear on the reference's UNet hot path
// (reference: sd/diffusion.py:179,205,298,359,363,381,435,545-569; sd/attention.py:42,91,190-198,249)
// with one LDS-tiled MFMA kernel:
//   * activations are NHWC (= token-major) fp16, K is ordered (kh, kw, ci), so every 64-wide
//     K-chunk of an im2col row is ONE contiguous 128-byte run of a source pixel (or zeros);
//   * A and W tiles are staged global->LDS with LDS-DMA (global_load_lds, 16 B/lane).  The LDS
//     image is lane-linear; bank conflicts are removed by XOR-swizzling the 16-B chunk index on the
//     per-lane SOURCE address and on the ds_read side with the same involution
//     (chunk' = chunk ^ ((row>>1)&7): conflict-free for ds_read_b128 of 32x32x16 fragments);
//   * nearest-x2 upsample, stride-2 and the skip-connection channel concat are folded into the A
//     address generator (no materialised copies: sd/diffusion.py:430,671);
//   * epilogue via LDS (fp32) for 16-B coalesced stores: + bias (+ time vector), + residual,
//     fp16 and/or fp32 output, optional transposed tail (V^T for the attention kernel);
//   * split-K over gridDim.y writes fp32 slabs combined by splitk_finalize (small-M layers).
#include "common.h"
#include <type_traits>

#ifndef SDMI_GACC_ABLATE
#define SDMI_GACC_ABLATE 0      // diagnostic builds: 1 = no flush at all, 2 = flush without the record stores, 4 = no per-item accumulation
#endif

namespace {

template <int BM_, int BN_, int WM_, int WN_, int NS_, int STG_ = 0, int KPI_ = 1, int PW_ = 1>
struct Cfg {
  static constexpr int KPI = KPI_;   // K-steps (of 64) per barrier interval (wave-specialised ring only)
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, NS = NS_;   // NS = LDS ring depth
  static constexpr int STG = STG_;   // 0: every wave stages and computes; 2: producer/consumer wave specialisation
  // NW = waves that own MFMA sub-tiles (and, for STG 0/1, also stage).  STG 2 adds NW producer waves that
  // only issue LDS-DMA, so each SIMD holds one MFMA wave and one DMA wave.
  // PW (STG 2 only): producer waves per MFMA wave.  One wave issues ~4.6 B/clk of LDS-DMA whatever its queue depth
  // (profiles/r03_load_paths.txt), so the K-loop of a small tile with NW producers runs at their issue rate, not at the
  // matrix pipe's; SW = staging waves.
  static constexpr int NW = WM * WN, SW = (STG_ == 2 ? NW * PW_ : NW), NT = 64 * (NW + (STG_ == 2 ? SW : 0));
  static_assert(STG_ == 2 || PW_ == 1, "producer multiplier needs the wave-specialised ring");
  static constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int RA = BM * 8 / (64 * SW), RB = BN * 8 / (64 * SW);
  static constexpr int CS_BYTES = BM * BN * 4;
  static constexpr int RING = NS * KPI * STAGE;
  static constexpr int LDS = (RING > CS_BYTES) ? RING : CS_BYTES;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(NS >= 2 && NS <= 8, "ring depth");
  static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile must be a multiple of 32x32");
  static_assert((BM * 8) % (64 * SW) == 0 && (BN * 8) % (64 * SW) == 0, "staging must divide evenly");
  static_assert(NT <= 1024, "workgroup size");
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  // 16 B per lane, LDS destination = wave-uniform base + lane*16
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// Counted wait of an NS-deep LDS-DMA ring: `rem` groups of G loads were issued after the one needed now; up to
// D = NS-2 of them may stay in flight (the immediate must be a constant, hence the unrolled chain).
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int D, int G>
__device__ __forceinline__ void wait_ring(int rem) {
  static_assert(D * G <= 63, "vmcnt is a 6-bit counter");
  if constexpr (D <= 0) wait_vmcnt<0>();
  else {
    if (rem >= D) wait_vmcnt<D * G>();
    else wait_ring<D - 1, G>(rem);
  }
}

// Position of key s inside a V^T row.  Within every group of 16 keys the four 4-key quads are stored in the order
// (q0, q2, q1, q3): the PV MFMA of the attention kernel wants, per lane half h, the quads h and 2+h of a group as ONE
// 16-byte LDS read (quads 0,2 | 1,3 become adjacent).  Purely a storage permutation of the key axis.
__device__ __forceinline__ int vt_pos(int s, int perm) {
  if (!perm) return s;
  const int u = s >> 2, g = u & 3;
  const int gp = (g == 1) ? 2 : (g == 2) ? 1 : g;
  return (((u & ~3) | gp) << 2) | (s & 3);
}

// phase2 GEMMs (GemmArgs::phase2): row m = phase*(M/4) + (b, y, x) on the source grid -> output pixel row of the 2x map
__device__ __forceinline__ int out_row(const GemmArgs& p, int m) {
  if (!p.phase2) return m;
  const int rp = p.M >> 2;
  const int ph = m / rp, rem = m - ph * rp;
  const int x = rem % p.Ws, t = rem / p.Ws;
  const int y = t % p.Hs, b = t / p.Hs;
  return (b * 2 * p.Hs + 2 * y + (ph >> 1)) * (2 * p.Ws) + 2 * x + (ph & 1);
}

// Shared epilogue: fp32 tile Cs[BM][BN] in LDS -> global (bias, residual, fp32/fp16 outputs, transposed
// V^T tail, or split-K slab).  All NT threads of the workgroup call it after a barrier.
template <int BM, int BN, int NT>
__device__ __forceinline__ void store_tile(const GemmArgs& p, const float* Cs, int m0, int n0, int kz, int tid,
                                           const float* s_ln = nullptr, int tn = 0, int tiles_n = 1, int vec_off = 0) {
  if (p.ksplit > 1) {
    float* slab = p.slab + (size_t)kz * p.M * p.N;
    for (int idx = tid; idx < BM * (BN / 4); idx += NT) {
      const int row = idx / (BN / 4), c4 = idx % (BN / 4);
      const int m = m0 + row, n = n0 + c4 * 4;
      if (m < p.M && n < p.N) *(f32x4*)(slab + (size_t)m * p.N + n) = *(const f32x4*)(Cs + row * BN + c4 * 4);
    }
    return;
  }

  const bool transposed = p.outT != nullptr && n0 >= p.nt0;
  // GroupNorm statistics of the stored values (GnRec, common.h): with NT a multiple of the BN/8 column chunks of a row, a thread
  // keeps the SAME 8 columns in every item, so it accumulates their moments in four registers (the two atoms they fall into)
  constexpr bool GACC = (NT % (BN / 8) == 0) && (BN % 64 == 0);
  // 160-wide tiles (round 5): NT is no multiple of the 20 column chunks of a row, so a thread's items wander over the columns and
  // cannot keep per-column registers.  They write the values they store back into the fp32 tile instead; afterwards one thread per
  // COLUMN adds its 128 (64) rows up in row order and one thread per 10-channel atom its columns: a tile holds whole atoms
  // (160 = 16 x 10, n0 a multiple of 160), so every record is {moments, 0}.  ~1 us per launch where taken; it keeps the
  // statistics chain (and with it the one-pass GroupNorm / the GroupNorm inside the halo conv) alive behind the 160-wide
  // plans the batched multi-prompt mode is tuned to.
  constexpr bool GACC160 = !GACC && BN == 160;
  const bool gstat = GACC && p.gacc.rec != nullptr;      // workgroup-uniform
  const bool gstat160 = GACC160 && p.gacc.rec != nullptr;
  GaccThread<(NT < 1024)> gth;
  int g_split = 8;
  if (gstat) {
    const int n = n0 + (tid % (BN / 8)) * 8;
    g_split = min(8, (gnrec_div_atom(p.gacc, n) + 1) * p.gacc.atom - n);
  }
  if (!transposed) {
    // Phase 1 issues EVERY residual / bias load of this thread's items before anything consumes them, so the tile
    // pays one memory latency instead of one per item (a one-K-step workgroup used to live 14 K cycles, most of
    // it in this loop).  Phase 2 reads the tile from LDS, adds, converts and stores.
    // Items are taken CH at a time so the epilogue's register footprint stays bounded on the big tiles.
    constexpr int ITEMS = (BM * (BN / 8) + NT - 1) / NT;
    constexpr int CH0 = ITEMS < 4 ? ITEMS : (NT >= 1024 ? 2 : 4);   // 1024-thread workgroups: 128 VGPRs per lane
    constexpr int CH = ITEMS % CH0 == 0 ? CH0 : (ITEMS % 5 == 0 ? 5 : 1);   // BN = 160 tiles: 5 or 10 items per thread
    static_assert(ITEMS % CH == 0, "epilogue chunking");
    const bool fold = p.ln_stat != nullptr && p.ln_ksteps == 0;
#pragma unroll
    for (int ch = 0; ch < ITEMS; ch += CH) {
    f32x4 r0[CH], r1[CH], b0[CH], b1[CH];
    f32x4 g0[CH], g1[CH];                  // LayerNorm fold: column sums of gamma (.) W
    f16x8 rh[CH];
    bool ok[CH];
#pragma unroll
    for (int it = 0; it < CH; ++it) {
      const int idx = (ch + it) * NT + tid;
      const int row = idx / (BN / 8), c8 = idx % (BN / 8);
      const int m = m0 + row, n = n0 + c8 * 8;
      ok[it] = idx < BM * (BN / 8) && m < p.M && n < p.N;
      const int mo = ok[it] ? out_row(p, m) : 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) { r0[it][e] = 0.f; r1[it][e] = 0.f; b0[it][e] = 0.f; b1[it][e] = 0.f; }
#pragma unroll
      for (int e = 0; e < 8; ++e) rh[it][e] = (f16)0.f;
      if (ok[it]) {
        if (p.bias) { b0[it] = *(const f32x4*)(p.bias + vec_off + n); b1[it] = *(const f32x4*)(p.bias + vec_off + n + 4); }
        if (fold) { g0[it] = *(const f32x4*)(p.ln_g + vec_off + n); g1[it] = *(const f32x4*)(p.ln_g + vec_off + n + 4); }
        if (p.res) {
          if (p.res_f32) {
            const float* rp = (const float*)p.res + (size_t)mo * p.ldr + n;
            r0[it] = *(const f32x4*)rp;
            r1[it] = *(const f32x4*)(rp + 4);
          } else {
            rh[it] = *(const f16x8*)((const f16*)p.res + (size_t)mo * p.ldr + n);
          }
        }
      }
    }
#pragma unroll
    for (int it = 0; it < CH; ++it) {
      const int idx = (ch + it) * NT + tid;
      const int row = idx / (BN / 8), c8 = idx % (BN / 8);
      const int m = m0 + row, n = n0 + c8 * 8;
      float rs = 0.f, rq = 0.f;
      if (ok[it]) {
        const f32x4 v0 = *(const f32x4*)(Cs + row * BN + c8 * 8);
        const f32x4 v1 = *(const f32x4*)(Cs + row * BN + c8 * 8 + 4);
        float v[8];
        if (fold) {
          const float mean = s_ln[2 * row], rstd = s_ln[2 * row + 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = rstd * (v0[e] - mean * g0[it][e]) + b0[it][e];
            v[4 + e] = rstd * (v1[e] - mean * g1[it][e]) + b1[it][e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = v0[e] + b0[it][e];
            v[4 + e] = v1[e] + b1[it][e];
          }
        }
        if (n < p.cs_hi) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= p.cscale;
        }
        if (p.act == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[e]));
        }
        if constexpr (BN == 128) {
          if (p.act == 2) {
            // softmax over this row of the tile (one attention head: the 16 lanes of a row are consecutive and aligned; the
            // launcher guarantees full tiles, so every lane of the wave is here).  Scores are already in the log2 domain.
            float mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              if (c8 * 8 + e >= p.sm_valid) v[e] = -INFINITY;
              mx = fmaxf(mx, v[e]);
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            float su = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] = __builtin_amdgcn_exp2f(v[e] - mx); su += v[e]; }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) su += __shfl_xor(su, o);
            const float inv = __builtin_amdgcn_rcpf(su);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= inv;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] += r0[it][e] + (float)rh[it][e];
          v[4 + e] += r1[it][e] + (float)rh[it][4 + e];
        }
        f16x8 o16;
#pragma unroll
        for (int e = 0; e < 8; ++e) o16[e] = (f16)v[e];
        const int mo = out_row(p, m);
        if (p.out_f32) {
          float* op = (float*)p.out + (size_t)mo * p.ldc + n;
          f32x4 o0, o1;
#pragma unroll
          for (int e = 0; e < 4; ++e) { o0[e] = v[e]; o1[e] = v[4 + e]; }
          *(f32x4*)op = o0;
          *(f32x4*)(op + 4) = o1;
          if (p.out16) *(f16x8*)(p.out16 + (size_t)mo * p.ldc + n) = o16;
        } else {
          *(f16x8*)((f16*)p.out + (size_t)mo * p.ldc + n) = o16;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float x = (float)o16[e]; rs += x; rq += x * x; }
        if (gstat) {       // moments of what the GroupNorm will read: the fp32 stream where there is one, else the fp16 tensor
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = p.out_f32 ? v[e] : (float)o16[e];
          if (!(SDMI_GACC_ABLATE & 4)) gth.add(x, g_split);
        }
        if constexpr (GACC160) {
          if (gstat160) {  // the same values back into the tile (this thread's own item: nobody else reads or writes these 32 bytes)
            float* cw = const_cast<float*>(Cs) + row * BN + c8 * 8;
            f32x4 w0, w1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { w0[e] = p.out_f32 ? v[e] : (float)o16[e]; w1[e] = p.out_f32 ? v[4 + e] : (float)o16[4 + e]; }
            *(f32x4*)cw = w0;
            *(f32x4*)(cw + 4) = w1;
          }
        }
      }
      if (p.rowstat && ((BN / 8) & (BN / 8 - 1)) == 0) {   // wave-uniform branch; the BN/8 lanes of one row are consecutive and aligned (power-of-two tiles only)
#pragma unroll
        for (int o = 1; o < BN / 8; o <<= 1) { rs += __shfl_xor(rs, o); rq += __shfl_xor(rq, o); }
        if (c8 == 0 && idx < BM * (BN / 8) && m < p.M) *(f32x2*)(p.rowstat + ((size_t)m * tiles_n + tn) * 2) = f32x2{rs, rq};
      }
    }
    }   // chunk
    if constexpr (GACC160) {
      if (gstat160) {
        float* s_g = const_cast<float*>(Cs);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // every item of the tile has been rewritten
        float cs_ = 0.f, cq_ = 0.f;
        if (tid < BN) {
#pragma unroll 8
          for (int rr = 0; rr < BM; ++rr) { const float x = s_g[rr * BN + tid]; cs_ += x; cq_ = fmaf(x, x, cq_); }
        }
        __builtin_amdgcn_s_barrier();                      // the column sums are in registers: the tile may be overwritten
        if (tid < BN) *(f32x2*)(s_g + 2 * tid) = f32x2{cs_, cq_};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int atom = p.gacc.atom, na = BN / atom;      // launcher: BN % atom == 0, n0 % atom == 0
        if (tid < na && n0 + tid * atom < p.N) {
          float s1 = 0.f, s2 = 0.f;
          for (int k = 0; k < atom; ++k) { const f32x2 t = *(const f32x2*)(s_g + 2 * (tid * atom + k)); s1 += t[0]; s2 += t[1]; }
          const bool phased = p.gacc.mod != p.M;
          const int ph = phased ? m0 / p.gacc.mod : 0;
          const int mm = m0 - ph * p.gacc.mod;
          const int img = gnrec_div_rows(p.gacc, mm);
          const int t_row = ph * (p.gacc.rows_img / BM) + (mm - img * p.gacc.rows_img) / BM;
          const int at = gnrec_div_atom(p.gacc, n0) + tid;
          f32x2* r = (f32x2*)p.gacc.rec + ((size_t)(img * p.gacc.T + t_row) * p.gacc.natoms + at) * 2;
          r[0] = f32x2{s1, s2};
          r[1] = f32x2{0.f, 0.f};
        }
      }
    }
    if constexpr (GACC) {
      if (gstat && !(SDMI_GACC_ABLATE & 1)) {
        // fixed-order reduction inside the workgroup: lanes of a wave that hold the same column chunk (butterfly), one record
        // per (wave, chunk) in LDS, then one thread per atom of the tile adds its chunks' records over the waves in fp64 and
        // stores the atom's moments into this tile's record (common.h GnRec)
        constexpr int CH8 = BN / 8, NWV = NT / 64;
        static_assert(NWV * CH8 * 16 <= BM * BN * 4, "the records fit the fp32 tile they replace");
        // the records go where the fp32 tile was (every thread has read its items: barrier first); no LDS of their own -- the
        // halo kernels run with all 160 KiB of dynamic LDS
        // (raw barriers with an LDS-only wait: __syncthreads() also waits for vmcnt(0), i.e. for every output store of the
        // epilogue to RETIRE -- measured 1.2 us per launch, all of what the statistics cost)
        float* s_g = const_cast<float*>(Cs);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const f32x4 mine = gth.parts(g_split);
        float qa = mine[0], qb = mine[1], qc = mine[2], qd = mine[3];
        if constexpr (CH8 <= 8) { qa = wave_xor_sum<8>(qa); qb = wave_xor_sum<8>(qb); qc = wave_xor_sum<8>(qc); qd = wave_xor_sum<8>(qd); }
        if constexpr (CH8 <= 16) { qa = wave_xor_sum<16>(qa); qb = wave_xor_sum<16>(qb); qc = wave_xor_sum<16>(qc); qd = wave_xor_sum<16>(qd); }
        qa = wave_xor_sum<32>(qa); qb = wave_xor_sum<32>(qb); qc = wave_xor_sum<32>(qc); qd = wave_xor_sum<32>(qd);
        const int ln = tid & 63, wv = tid >> 6;
        if (ln < CH8) *(f32x4*)(s_g + (wv * CH8 + ln) * 4) = f32x4{qa, qb, qc, qd};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // one lane per atom of the tile adds the (at most three) column chunks of the atom over the waves' records in a fixed
        // order; fp32 like the per-thread sums below them (the GroupNorm adds the tiles' records up in fp64)
        const int atom = p.gacc.atom;
        const int a_first = gnrec_div_atom(p.gacc, n0), a_last = gnrec_div_atom(p.gacc, min(p.N, n0 + BN) - 1);
        if (tid <= a_last - a_first) {
          const int at = a_first + tid;
          const int lo_col = max(at * atom, n0) - n0, hi_col = min((at + 1) * atom, n0 + BN) - 1 - n0;   // tile-local columns
          const int c_lo = lo_col >> 3, c_hi = hi_col >> 3;
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) {                            // at most three chunks (atom <= 16 ... 2 x 8 + 2)
            const int c = min(c_lo + cc, c_hi);
            const bool use = c_lo + cc <= c_hi;
            const int sel = (gnrec_div_atom(p.gacc, n0 + 8 * c) == at) ? 0 : 2;      // the chunk's first or second part belongs to this atom
            f32x2 t[NWV];
#pragma unroll
            for (int w = 0; w < NWV; ++w) t[w] = *(const f32x2*)(s_g + (w * CH8 + c) * 4 + sel);
#pragma unroll
            for (int w = 0; w < NWV; ++w) { s1 += use ? t[w][0] : 0.f; s2 += use ? t[w][1] : 0.f; }
          }
          // record row of this tile: (phase,) image, row block inside the image
          const bool phased = p.gacc.mod != p.M;
          const int ph = phased ? m0 / p.gacc.mod : 0;
          const int mm = m0 - ph * p.gacc.mod;
          const int img = gnrec_div_rows(p.gacc, mm);
          const int t_row = ph * (p.gacc.rows_img / BM) + (mm - img * p.gacc.rows_img) / BM;
          f32x2* r = (f32x2*)p.gacc.rec + ((size_t)(img * p.gacc.T + t_row) * p.gacc.natoms + at) * 2;
          // the tile that holds the atom's first channel writes part 0, the one that holds its last channel part 1 (zero
          // when both are this tile): every slot is written exactly once per launch
          const bool has_first = at * atom >= n0, has_last = (at + 1) * atom <= n0 + BN;
          if (SDMI_GACC_ABLATE & 2) { asm volatile("" ::"v"(s1), "v"(s2), "v"(r)); }
          else {
            if (has_first) r[0] = f32x2{s1, s2};
            if (has_last) r[1] = has_first ? f32x2{0.f, 0.f} : f32x2{s1, s2};
          }
        }
      }
    }
  } else {
    // transposed tail: 8 rows (tokens) of one column -> ONE 16-byte store along the key axis.  With the attention
    // kernel's quad-permuted key order (vt_pos: quads of a 16-key group stored as q0,q2,q1,q3) the 8 tokens a thread
    // takes are quads {hf, hf+2} of a group -- adjacent in storage -- instead of two neighbouring quads that land 16 B
    // apart (two 8-byte stores per item before; the V^T tail of in_proj is store-transaction bound).
    const int Ct = p.N - p.nt0;
    const bool wide = (p.S & 15) == 0 && (BM & 15) == 0;
    for (int idx = tid; idx < (BM / 8) * BN; idx += NT) {
      const int col = idx % BN, r8 = idx / BN;
      const int n = n0 + col;
      int rows[8];
      if (wide && p.tperm) {
        const int g16 = r8 >> 1, hf = r8 & 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { rows[e] = g16 * 16 + hf * 4 + e; rows[4 + e] = g16 * 16 + 8 + hf * 4 + e; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) rows[e] = r8 * 8 + e;
      }
      const int m = m0 + rows[0];
      if (m >= p.M || n >= p.N) continue;
      const float bv = p.bias ? p.bias[vec_off + n] : 0.f;
      f16x8 o16;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = Cs[rows[e] * BN + col];
        if (p.ln_stat && p.ln_ksteps == 0) v = s_ln[2 * rows[e] + 1] * (v - s_ln[2 * rows[e]] * p.ln_g[vec_off + n]);
        o16[e] = (f16)(v + bv);
      }
      if (wide && p.tperm) {
        // rows of a 16-token group never straddle images (S % 16 == 0, m0 % 16 == 0); a group beyond M is skipped whole
        const int b = m / p.S, s = m - b * p.S;               // s = first token of quad hf of its group
        f16* row = p.outT + ((size_t)b * Ct + (n - p.nt0)) * p.ldt;
        if (m0 + rows[7] < p.M) *(f16x8*)(row + vt_pos(s, 1)) = o16;
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (m0 + rows[e] < p.M) row[vt_pos(s - (rows[0] - rows[e]), 1)] = o16[e];
        }
      } else if ((p.S & 7) == 0) {
        const int b = m / p.S, s = m - b * p.S;
        f16* row = p.outT + ((size_t)b * Ct + (n - p.nt0)) * p.ldt;
        *(f16x4*)(row + vt_pos(s, p.tperm)) = f16x4{o16[0], o16[1], o16[2], o16[3]};
        *(f16x4*)(row + vt_pos(s + 4, p.tperm)) = f16x4{o16[4], o16[5], o16[6], o16[7]};
      } else {   // tiny maps (S not a multiple of 8): element-wise, rows may straddle images
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int me = m + e;
          if (me < p.M) {
            const int b = me / p.S, s = me - b * p.S;
            p.outT[((size_t)b * Ct + (n - p.nt0)) * p.ldt + vt_pos(s, p.tperm)] = o16[e];
          }
        }
      }
    }
  }
}

#ifdef SDMI_CLK_PROBE
__device__ unsigned long long g_clk_probe[2048][2];   // diagnostic build only: {shader cycles, 100 MHz ticks} per workgroup
__device__ unsigned long long g_clk_pre[2048][10];     // igemm STG 2: producer {setup done, NS-1 stages issued, first stage landed}, consumer {first barrier passed}
__device__ unsigned long long g_clk_phase[2048][6];   // {realtime start, end, cycles after setup / K loop / tile in LDS / end}
#endif

constexpr int kGnaMaxC = 1280;       // widest K = C of a GroupNorm-on-fragments launch (GemmArgs::gna_rec)
// ACC (the ACCURATE mode, SDMI_FLAG_ACCURATE; the plain STG 0 ring only): the A operand comes from the fp32 tensors (GemmArgs::a0f
// ...) and is split on its way into LDS into a hi + lo fp16 pair, a = hi + lo with hi = fp16(a), lo = fp16(a - hi); every
// fragment is multiplied twice against the same W fragment, acc += hi W + lo W.  The activation then enters the product with
// ~22 significant bits instead of 11 (what is left is the fp16 rounding of the WEIGHTS, the measured 4.7e-4 floor of
// tests/golden/stress_floor.json against 1.4e-3 for fp16 activations).  Twice the MFMA work and the A tile staged through
// registers (global_load fp32 -> split -> ds_write, in the lane-linear image the LDS-DMA would have written): a validation /
// high-accuracy path whose cost bench.py reports as `accurate_mode`, not the default.
template <class C, bool GNA = false, bool ACC = false>
__global__ __launch_bounds__(C::NT) void igemm_kernel(GemmArgs p) {
  static_assert(!ACC || (C::STG == 0 && !GNA && C::NT < 1024), "the wide-operand variant is built on the plain ring");
#ifdef SDMI_CLK_PROBE
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // every argument the prologue needs, requested in ONE batch: left to itself the compiler loads each next to its first use behind
  // its own wait -- a dozen serial scalar-cache misses (the launch's kernarg slot is new memory), and even hits cost ~100 cycles
  // apiece in series.  The batch touches every 64-byte line of the struct (p.bias: the epilogue's line), so the later loads hit.
  if constexpr (C::NT < 1024)
    asm volatile("" ::"s"(p.a0), "s"(p.w), "s"(p.zero), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.C0), "s"(p.C1), "s"(p.lda0), "s"(p.ldw),
                 "s"(p.ks), "s"(p.stride), "s"(p.ups), "s"(p.phase2), "s"(p.ksplit), "s"(p.ksteps_per), "s"(p.n_major), "s"(p.img_rows),
                 "s"(p.w_img_stride), "s"(p.vec_img_stride), "s"(p.ln_stat), "s"(p.ln_ksteps), "s"(p.tiles), "s"(p.tdiv), "s"(p.plain),
                 "s"(p.tiles_magic), "s"(p.tdiv_magic), "s"(p.bias), "s"(p.ln_guard), "s"(p.gacc.rec), "s"(p.gacc.rows_magic));
  else      // the 16-wave kernels have 128 VGPRs per lane and no room for the SGPR pressure of the batch: warm the lines only
    sdmi_kernarg_warm<sizeof(GemmArgs) + 16>();     // + the hidden grid size this kernel reads (gridDim.x)
#ifdef SDMI_CLK_PROBE
  if (threadIdx.x == 0 && blockIdx.x < 2048) g_clk_pre[blockIdx.x][4] = __builtin_amdgcn_s_memtime() - clk_t0;
#endif
  constexpr int BM = C::BM, BN = C::BN, NW = C::NW, NT = C::NT, SW = C::SW;
  constexpr int FM = C::FM, FN = C::FN, RA = C::RA, RB = C::RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave-uniform roles (STG 2): waves [0,NW) run ds_read + MFMA, waves [NW,2NW) only issue LDS-DMA
  const bool producer = C::STG == 2 && wave_id >= NW;
  const int wave = producer ? 0 : wave_id;                    // MFMA wave index (producers own no sub-tile)
  const int sw = C::STG == 2 ? (producer ? wave_id - NW : 0) : wave_id;   // index among the staging waves
  const int wm = wave / C::WN, wn = wave % C::WN;
  // XCD-aware block -> tile map.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and
  // b+8 share an L2), so a plain map makes every XCD stream the whole activation tensor through its
  // 4 MiB L2.  Remap (bijectively) so each XCD owns a CONTIGUOUS range of (k-split, m-tile, n-tile)
  // ids: the 9 taps x n-tiles re-reads of an activation band then hit that XCD's L2.  Speed only.
  const int tiles_n = (p.N + BN - 1) / BN;
  int kz, tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    kz = p.ksplit == 1 ? 0 : (int)(((unsigned long long)L * p.tiles_magic) >> 36);      // L / tiles
    tile = L - kz * p.tiles;
  }
  // tile order inside a K-slice (speed only): the 8 XCDs own CONTIGUOUS tile ranges.  m-major (n fastest) makes every XCD
  // stream all of W and 1/8 of A; n-major the reverse.  The launcher picks the order that moves fewer bytes through the
  // eight L2s (n-major when the weights are the larger operand: the 16x16 / 8x8 levels).
  const int tq = (int)(((unsigned long long)tile * p.tdiv_magic) >> 36), tr = tile - tq * p.tdiv;   // tile / tdiv, no division
  const int tm = p.n_major ? tr : tq;
  const int tn = p.n_major ? tq : tr;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nkt = p.K >> 6;
  const int kt0 = kz * p.ksteps_per;
  const int kt1 = min(kt0 + p.ksteps_per, nkt);

#ifdef SDMI_CLK_PROBE
  if (C::STG == 2 && threadIdx.x == 64 * C::NW && blockIdx.x < 2048) g_clk_pre[blockIdx.x][5] = __builtin_amdgcn_s_memtime() - clk_t0;
#endif
  // LayerNorm fold: mean / rstd of this tile's rows from the producer's per-n-tile partial sums.  Issued first so
  // the loads overlap the main loop; read after the epilogue barrier.
  __shared__ float s_ln[2 * BM + BN];       // {mean, rstd} per tile row, then ln_g of the tile's columns (ln_ksteps > 0)
  // per-image operands (folded cross-attention): a tile never straddles images (BM divides img_rows)
  const int img = p.img_rows ? m0 / p.img_rows : 0;
  const int vec_off = img * p.vec_img_stride;
  const f16* w_img = p.w + (size_t)img * p.w_img_stride;
  if (p.ln_stat != nullptr && p.ln_ksteps > 0 && tid < BN) s_ln[2 * BM + tid] = n0 + tid < p.N ? p.ln_g[vec_off + n0 + tid] : 0.f;
  if (p.ln_stat != nullptr && tid < BM) {
    const int m = min(m0 + tid, p.M - 1);
    const f32x2* sp = (const f32x2*)p.ln_stat + (size_t)m * p.ln_ntn;
    // per-n-tile partial sums are fp32; they are combined and the variance E[x^2] - E[x]^2 is taken in fp64 (as the GroupNorm
    // reductions of norm.hip): in fp32 the cancellation costs ~(mean/sigma)^2 ulps of the variance on rows with a large mean
    double su = 0.0, sq = 0.0;
    for (int j = 0; j < p.ln_ntn; ++j) { const f32x2 t = sp[j]; su += (double)t[0]; sq += (double)t[1]; }
    const double inv = 1.0 / (double)p.ln_C;
    const double mean_d = su * inv;
    double var = sq * inv - mean_d * mean_d;
    var = var < 0.0 ? 0.0 : var;
    const float mean = (float)mean_d;
    s_ln[2 * tid] = mean;
    s_ln[2 * tid + 1] = rsqrtf((float)var + p.ln_eps);
    if (p.ln_guard != nullptr && m0 + tid < p.M && tn == 0 && kz == 0 && mean_d * mean_d > (double)p.ln_guard_thr2 * var) atomicAdd(p.ln_guard, 1);
    if (p.ln_out != nullptr && kz == 0 && tn == 0 && m0 + tid < p.M) {      // for splitk_finalize (partial fold + split-K)
      p.ln_out[2 * (m0 + tid)] = mean;
      p.ln_out[2 * (m0 + tid) + 1] = s_ln[2 * tid + 1];
    }
  }

#ifdef SDMI_CLK_PROBE
  if (C::STG == 2 && threadIdx.x == 64 * C::NW && blockIdx.x < 2048) g_clk_pre[blockIdx.x][6] = __builtin_amdgcn_s_memtime() - clk_t0;
#endif
  const int Cin = p.C0 + p.C1;
  const int Hi = p.Hs << p.ups, Wi = p.Ws << p.ups;

  // ---- per-lane staging state --------------------------------------------------------------
  // K is walked in SEGMENTS = (tap, concat source): inside a segment consecutive K-steps advance the
  // per-row source pointer by 64 channels, so the im2col address is recomputed only at segment
  // boundaries (every C_src/64 steps) and a K-step costs two 64-bit adds per staged row.
  int a_ihb[RA], a_iwb[RA], a_pix0[RA], a_gch[RA];
  bool a_ok[RA];
  typedef typename std::conditional<ACC, float, f16>::type AT;      // element type of the A sources
  // (the segment walk below selects between MEMBERS of the argument struct, never between local copies of them: a lambda captures
  // locals by address, the compiler turns select(load a, load b) into load(select(&a, &b)) inside it, and locals whose addresses
  // feed a select stay in scratch memory -- seen as 900 B of scratch per lane in a first form of the wide-operand variant; a
  // run-time offset into the kernarg segment is just a scalar load)
  const AT* const src_zero = (const AT*)(const void*)p.zero;       // >= 2 KB of zero bytes: 8 fp32 at chunk offsets up to 56 fit
  const AT* a_ptr[RA];
  int a_inc[RA];
  const bool plain = C::NT < 1024 && p.plain;      // (the 16-wave kernels have no registers to spare for a second path)
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int q = (i * SW + sw) * 64 + lane;
    const int row = q >> 3, pc = q & 7;
    a_gch[i] = (pc ^ ((row >> 1) & 7)) * 8;          // element offset of the global chunk to fetch
    const int m = m0 + row;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    if (plain) {
      // plain GEMM (every linear / 1x1 conv over one source: 122 of the launches of a step): output row m IS source row m and K is
      // ONE segment, so neither the im2col decomposition (four integer divisions per staged row) nor the segment walk exists --
      // the prologue of these launches was ~2 k cycles of address arithmetic before the first tile was requested
      a_ptr[i] = a_ok[i] ? (ACC ? (const AT*)(const void*)p.a0f : (const AT*)(const void*)p.a0) + ((size_t)mm * p.lda0 + (size_t)kt0 * 64 + a_gch[i]) : src_zero + a_gch[i];
      a_inc[i] = a_ok[i] ? 64 : 0;
      a_ihb[i] = 0; a_iwb[i] = 0; a_pix0[i] = mm;
    } else if (p.ks == 1 && p.stride == 1 && p.ups == 0) {
      a_ihb[i] = 0; a_iwb[i] = 0; a_pix0[i] = mm;      // 1x1 over a virtual concat: row m is source pixel m of both sources
    } else if (p.phase2) {
      const int rp = p.M >> 2;
      const int ph = mm / rp, rem = mm - ph * rp;
      const int x = rem % p.Ws, t = rem / p.Ws;
      const int y = t % p.Hs, b = t / p.Hs;
      a_ihb[i] = y + (ph >> 1) - 1;
      a_iwb[i] = x + (ph & 1) - 1;
      a_pix0[i] = b * p.Hs * p.Ws;
    } else {
      const int ow = mm % p.Wo, t = mm / p.Wo;
      const int oh = t % p.Ho, b = t / p.Ho;
      a_ihb[i] = oh * p.stride - p.pad;
      a_iwb[i] = ow * p.stride - p.pad;
      a_pix0[i] = b * p.Hs * p.Ws;
    }
  }
#ifdef SDMI_CLK_PROBE
  if (C::STG == 2 && threadIdx.x == 64 * C::NW && blockIdx.x < 2048) g_clk_pre[blockIdx.x][7] = __builtin_amdgcn_s_memtime() - clk_t0;
#endif
  const f16* b_ptr[RB];
  int b_inc[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int q = (i * SW + sw) * 64 + lane;
    const int row = q >> 3, pc = q & 7;
    const int gch = (pc ^ ((row >> 1) & 7)) * 8;
    const int n = n0 + row;
    const bool ok = n < p.N;
    b_ptr[i] = ok ? w_img + (size_t)n * p.ldw + (size_t)kt0 * 64 + gch : p.zero + gch;
    b_inc[i] = ok ? 64 : 0;
  }

#ifdef SDMI_CLK_PROBE
  if (C::STG == 2 && threadIdx.x == 64 * C::NW && blockIdx.x < 2048) g_clk_pre[blockIdx.x][8] = __builtin_amdgcn_s_memtime() - clk_t0;
#endif
  // segment state (wave-uniform)
  int tap = 0, seg_c = 0;
  int seg_left = 0;                       // K-steps left in the current segment
  if (plain) seg_left = 1 << 30;          // one segment, opened above
  else {
    tap = (kt0 * 64) / Cin;
    seg_c = kt0 * 64 - tap * Cin;         // channel offset inside the concatenated Cin
    if (tap >= p.ks * p.ks) { tap = p.ks * p.ks; seg_c = kt0 * 64 - tap * Cin; }   // inside the extra 1x1 segment
  }

  const int ntaps = p.ks * p.ks;
  auto open_segment = [&]() {
    const bool extra = tap >= ntaps;                 // fused 1x1 skip segment: centre tap of x0 | x1
    const int kh = extra ? p.pad : ((p.ks == 3) ? tap / 3 : (p.ks == 2 ? tap >> 1 : 0));
    const int kw = extra ? p.pad : ((p.ks == 3) ? tap - (tap / 3) * 3 : (p.ks == 2 ? tap & 1 : 0));
    const int Ca = extra ? p.X0 : p.C0, Cb = extra ? p.X1 : p.C1;
    const bool second = seg_c >= Ca;
    const AT* base;
    if constexpr (ACC) base = (const AT*)(const void*)(extra ? (second ? p.x1f : p.x0f) : (second ? p.a1f : p.a0f));
    else base = (const AT*)(const void*)(extra ? (second ? p.x1 : p.x0) : (second ? p.a1 : p.a0));
    const int cs = second ? Cb : Ca;
    const int ld = extra ? (second ? p.ldx1 : p.ldx0) : (second ? p.lda1 : p.lda0);
    const int cc = second ? seg_c - Ca : seg_c;
    seg_left = (cs - cc) >> 6;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int ih = a_ihb[i] + kh, iw = a_iwb[i] + kw;
      const bool v = a_ok[i] && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi;
      const int pix = a_pix0[i] + (extra ? ih * p.Ws + iw : (ih >> p.ups) * p.Ws + (iw >> p.ups));
      const AT* gr = base + ((size_t)pix * ld + cc + a_gch[i]);
      const AT* gz = src_zero + a_gch[i];
      a_ptr[i] = v ? gr : gz;
      a_inc[i] = v ? 64 : 0;
    }
    seg_c += seg_left << 6;
    if (seg_c >= Ca + Cb) { seg_c = 0; ++tap; }
  };

  constexpr int STAGE = C::STAGE + (ACC ? C::A_BYTES : 0);      // ACC: [A hi | A lo | W] per stage
  constexpr int B_OFF = C::A_BYTES * (ACC ? 2 : 1);
  auto stage = [&](int buf) {
    char* sa = smem + buf * STAGE;
    char* sb = sa + B_OFF;
    if (seg_left == 0) open_segment();
    if constexpr (ACC) {
      // through registers: 8 fp32 per lane and row chunk -> hi | lo fp16 -> the two A images, at the LDS position the 16-byte
      // LDS-DMA of the same lane would have written (wave base + lane * 16), so the fragment reads below do not change
      f32x4 v0[RA], v1[RA];
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        typedef const __attribute__((address_space(1))) f32x4* gvec_t;      // global, not flat: the tensors live in device memory
        v0[i] = *(gvec_t)(const void*)a_ptr[i];
        v1[i] = *(gvec_t)(const void*)(a_ptr[i] + 4);
        a_ptr[i] += a_inc[i];
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        glds16(b_ptr[i], sb + (i * SW + sw) * 1024);
        b_ptr[i] += b_inc[i];
      }
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = e < 4 ? v0[i][e] : v1[i][e - 4];
          hi[e] = (f16)x;
          lo[e] = (f16)(x - (float)hi[e]);
        }
        *(f16x8*)(sa + (i * SW + sw) * 1024 + lane * 16) = hi;
        *(f16x8*)(sa + C::A_BYTES + (i * SW + sw) * 1024 + lane * 16) = lo;
      }
    } else {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      glds16(a_ptr[i], sa + (i * SW + sw) * 1024);
      a_ptr[i] += a_inc[i];
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      glds16(b_ptr[i], sb + (i * SW + sw) * 1024);
      b_ptr[i] += b_inc[i];
    }
    }
    --seg_left;
  };

  // ---- fragment read offsets ---------------------------------------------------------------
  const int r = lane & 31, h = lane >> 5;
  const int key = (r >> 1) & 7;
  int coff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) coff[s] = ((2 * s + h) ^ key) * 16;
  const int a_row_off = (wm * C::TM + r) * 128;
  const int b_row_off = (wn * C::TN + r) * 128;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // MFMA over one staged K-step.  Fragments are double-buffered in registers: the ds_read_b128 of k16
  // sub-step s+1 are issued BEFORE the MFMAs of sub-step s, so the compiler's counted lgkmcnt waits only for
  // the older reads and the LDS latency hides under the matrix pipe (it was exposed 4x per K-step before).
  // GroupNorm on the A fragments (GNA kernels, GemmArgs::gna_rec): per channel {mean_hi, mean_lo, rstd gamma, beta} in fp16, written
  // by gna_prologue below; the lane's eight K columns of sub-step s of K-step kt are kt*64 + 16 s + 8 h .. + 7 for every row
  __shared__ __attribute__((aligned(16))) f16 s_gna[GNA ? 4 * kGnaMaxC : 8];
  __shared__ double s_gred[GNA ? 8 : 1][32][2];
  __shared__ float s_gms[GNA ? 64 : 1];
  auto compute = [&](int buf, int kt) {
    const char* As = smem + buf * STAGE + a_row_off;
    const char* Bs = smem + buf * STAGE + B_OFF + b_row_off;
    f16x8 af[2][FM], bf[2][FN];
    f16x8 al[2][ACC ? FM : 1];            // ACC: the lo halves of the A fragments
    f16x8 gv[2][4];                       // GNA: the four vectors of the sub-step, read one sub-step ahead like the fragments
    const f16* gsrc = s_gna + (GNA ? kt * 64 + 8 * h : 0);
#pragma unroll
    for (int i = 0; i < FM; ++i) af[0][i] = *(const f16x8*)(As + i * 32 * 128 + coff[0]);
    if constexpr (ACC) {
#pragma unroll
      for (int i = 0; i < FM; ++i) al[0][i] = *(const f16x8*)(As + C::A_BYTES + i * 32 * 128 + coff[0]);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) bf[0][j] = *(const f16x8*)(Bs + j * 32 * 128 + coff[0]);
    if constexpr (GNA) {
#pragma unroll
      for (int v = 0; v < 4; ++v) gv[0][v] = *(const f16x8*)(gsrc + v * kGnaMaxC);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) {
#pragma unroll
        for (int i = 0; i < FM; ++i) af[(s + 1) & 1][i] = *(const f16x8*)(As + i * 32 * 128 + coff[s + 1]);
        if constexpr (ACC) {
#pragma unroll
          for (int i = 0; i < FM; ++i) al[(s + 1) & 1][i] = *(const f16x8*)(As + C::A_BYTES + i * 32 * 128 + coff[s + 1]);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[(s + 1) & 1][j] = *(const f16x8*)(Bs + j * 32 * 128 + coff[s + 1]);
        if constexpr (GNA) {
#pragma unroll
          for (int v = 0; v < 4; ++v) gv[(s + 1) & 1][v] = *(const f16x8*)(gsrc + v * kGnaMaxC + 16 * (s + 1));
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the next sub-step's reads ahead of this sub-step's MFMAs
      if constexpr (GNA) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
          af[s & 1][i] = __builtin_elementwise_fma((af[s & 1][i] - gv[s & 1][0]) - gv[s & 1][1], gv[s & 1][2], gv[s & 1][3]);
      }
      if constexpr (ACC) {                 // the small terms first: lo W, then hi W on top
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s & 1][i], bf[s & 1][j], acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s & 1][i], bf[s & 1][j], acc[i][j], 0, 0, 0);
    }
  };

  // partial LayerNorm fold (ln_ksteps): rescale the accumulators in registers once the folded K segment has been summed
  auto rescale_ln = [&]() {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const float gcol = s_ln[2 * BM + wn * C::TN + j * 32 + r];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm * C::TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          acc[i][j][e] = s_ln[2 * row + 1] * (acc[i][j][e] - s_ln[2 * row] * gcol);
        }
      }
  };
  // GNA prologue: the statistics records of this tile's image -> mean / rstd per group (fp64, as gn_apply_kernel of norm.hip adds
  // them) -> the four per-channel fp16 vectors.  Three workgroup barriers; every wave of the workgroup calls it exactly once
  // (the producers after they have primed the ring, so the records' latency hides under the first tiles').
  auto gna_prologue = [&]() {
    if constexpr (GNA) {
      const int Cn = p.C0, cpg = Cn >> 5, apg = cpg / p.gna_atom, natoms = Cn / p.gna_atom;
      const int img = m0 / p.gna_rows;
      const int T = p.gna_T, parts = p.gna_parts;
      // gamma / beta of this thread's channels requested first (they come from HBM: every weight is read once per step)
      constexpr int NCH = (kGnaMaxC + NT - 1) / NT;
      float gm[NCH], bt[NCH];
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int c = tid + k * NT;
        gm[k] = c < Cn ? p.gna_gamma[c] : 0.f;
        bt[k] = c < Cn ? p.gna_beta[c] : 0.f;
      }
      if (tid < 256) {
        const int g = tid & 31, sl = tid >> 5;          // group, share (8 shares of the group's (atom, record row) pairs)
        const int npair = apg * T;
        constexpr int MAXR = 16;
        f32x4 rv[MAXR];
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
          const int f = sl + 8 * k;
          rv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (f < npair) {
            const int t = f / apg, a = g * apg + (f - t * apg);
            const float* rr = p.gna_rec + ((size_t)(img * T + t) * natoms + a) * parts * 2;
            if (parts == 2) rv[k] = *(const f32x4*)rr;
            else { const f32x2 u = *(const f32x2*)rr; rv[k][0] = u[0]; rv[k][1] = u[1]; }
          }
        }
        double su = 0.0, sq = 0.0;
#pragma unroll
        for (int k = 0; k < MAXR; ++k) { su += (double)rv[k][0] + (double)rv[k][2]; sq += (double)rv[k][1] + (double)rv[k][3]; }
        for (int f = sl + 8 * MAXR; f < npair; f += 8) {
          const int t = f / apg, a = g * apg + (f - t * apg);
          const float* rr = p.gna_rec + ((size_t)(img * T + t) * natoms + a) * parts * 2;
          su += (double)rr[0]; sq += (double)rr[1];
          if (parts == 2) { su += (double)rr[2]; sq += (double)rr[3]; }
        }
        s_gred[sl][g][0] = su;
        s_gred[sl][g][1] = sq;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid < 32) {
        double su = 0.0, sq = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { su += s_gred[k][tid][0]; sq += s_gred[k][tid][1]; }
        const double cnt = (double)cpg * (double)p.gna_rows;
        const double mean = su / cnt;
        double var = sq / cnt - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        s_gms[tid] = (float)mean;
        s_gms[32 + tid] = rsqrtf((float)var + p.gna_eps);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int c = tid + k * NT;
        if (c < Cn) {
          const int g = c / cpg;
          const float mean = s_gms[g], rstd = s_gms[32 + g];
          const f16 mh = (f16)mean;
          s_gna[c] = mh;
          s_gna[kGnaMaxC + c] = (f16)(mean - (float)mh);
          s_gna[2 * kGnaMaxC + c] = (f16)(rstd * gm[k]);
          s_gna[3 * kGnaMaxC + c] = (f16)bt[k];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  };
  const int ln_at = (p.ln_stat != nullptr && p.ksplit == 1) ? p.ln_ksteps : 0;   // K-step count after which to rescale (one-pass path; split-K rescales in splitk_finalize)
#ifdef SDMI_CLK_PROBE
  const unsigned long long clk_setup = __builtin_amdgcn_s_memtime() - clk_t0;
#endif
  const int nk = kt1 - kt0;
  if constexpr (C::STG == 2) {
    // ---- wave-specialised ring: waves [NW, 2NW) issue LDS-DMA NS-1 intervals ahead, waves [0, NW) run
    //      ds_read + MFMA.  An interval = KPI K-steps; one s_barrier per interval for both roles: the
    //      producer passes it only after the loads of the NEXT interval have landed (counted vmcnt), the
    //      consumer after it has finished reading the current buffers -> the barrier closes both the RAW
    //      and the WAR window.  Ring slot = interval % NS, sub-buffer j of a slot = (slot*KPI + j).
    constexpr int NS = C::NS, KPI = C::KPI;
    constexpr int G = (RA + RB) * KPI;      // LDS-DMA instructions per wave per full interval
    const int ni = (nk + KPI - 1) / KPI;    // intervals (the last may be partial)
    if (producer) {
      auto stage_interval = [&](int it, int slot) {
#pragma unroll
        for (int j = 0; j < KPI; ++j)
          if (it * KPI + j < nk) stage(slot * KPI + j);
          else if (KPI > 1) {            // keep the per-interval DMA count constant for counted vmcnt
#pragma unroll
            for (int i = 0; i < RA + RB; ++i) glds16(p.zero, smem + ((slot * KPI + j) * C::STAGE) + (i * SW + sw) * 1024);
          }
      };
#ifdef SDMI_CLK_PROBE
      const unsigned long long pre0 = __builtin_amdgcn_s_memtime() - clk_t0;
#endif
#pragma unroll
      for (int s = 0; s < NS - 1; ++s)
        if (s < ni) stage_interval(s, s);
#ifdef SDMI_CLK_PROBE
      const unsigned long long pre1 = __builtin_amdgcn_s_memtime() - clk_t0;
#endif
      gna_prologue();
      {
        wait_ring<NS - 2, G>(ni - 1);
      }
#ifdef SDMI_CLK_PROBE
      if (lane == 0 && sw == 0 && blockIdx.x < 2048) {
        g_clk_pre[blockIdx.x][0] = pre0; g_clk_pre[blockIdx.x][1] = pre1; g_clk_pre[blockIdx.x][2] = __builtin_amdgcn_s_memtime() - clk_t0;
      }
#endif
      __builtin_amdgcn_s_barrier();
      int nxt = NS - 1;
#ifdef SDMI_CLK_PROBE
      unsigned long long acc_issue = 0, acc_vm = 0, acc_bar = 0;
#endif
      for (int t = 0; t < ni; ++t) {
#ifdef SDMI_CLK_PROBE
        const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#endif
        if (t + NS - 1 < ni) stage_interval(t + NS - 1, nxt);
#ifdef SDMI_CLK_PROBE
        const unsigned long long s1 = __builtin_amdgcn_s_memtime();
#endif
        const int rem = ni - 2 - t;           // intervals issued after interval t+1
        wait_ring<NS - 2, G>(rem);
#ifdef SDMI_CLK_PROBE
        const unsigned long long s2 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef SDMI_CLK_PROBE
        const unsigned long long s3 = __builtin_amdgcn_s_memtime();
        acc_issue += s1 - s0; acc_vm += s2 - s1; acc_bar += s3 - s2;
#endif
        nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
      }
#ifdef SDMI_CLK_PROBE
      if (lane == 0 && wave == 0 && blockIdx.x < 512) {
        g_clk_probe[512 + blockIdx.x][0] = acc_issue; g_clk_probe[512 + blockIdx.x][1] = acc_vm;
        g_clk_probe[1024 + blockIdx.x][0] = acc_bar;
      }
#endif
    } else {
      gna_prologue();
      __builtin_amdgcn_s_barrier();
      int cur = 0;
#ifdef SDMI_CLK_PROBE
      if (tid == 0 && blockIdx.x < 2048) g_clk_pre[blockIdx.x][3] = __builtin_amdgcn_s_memtime() - clk_t0;
      unsigned long long acc_cmp = 0, acc_cbar = 0;
#endif
      for (int t = 0; t < ni; ++t) {
#ifdef SDMI_CLK_PROBE
        const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int j = 0; j < KPI; ++j)
          if (t * KPI + j < nk) {
            compute(cur * KPI + j, kt0 + t * KPI + j);
            if (ln_at && t * KPI + j + 1 == ln_at) rescale_ln();
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef SDMI_CLK_PROBE
        const unsigned long long s1 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef SDMI_CLK_PROBE
        acc_cmp += s1 - s0; acc_cbar += __builtin_amdgcn_s_memtime() - s1;
#endif
        cur = (cur + 1 == NS) ? 0 : cur + 1;
      }
#ifdef SDMI_CLK_PROBE
      if (lane == 0 && wave == 0 && blockIdx.x < 512) {
        g_clk_probe[1536 + blockIdx.x][0] = acc_cmp; g_clk_probe[1536 + blockIdx.x][1] = acc_cbar;
      }
#endif
    }
  } else {
  // ---- LDS-DMA ring: NS-deep, loads stay in flight across barriers (counted vmcnt) ---------------
  //   iteration t: wait until the loads of step t have landed (later groups may stay in flight),
  //   barrier (also closes the WAR window on the buffer read in iteration t-1), issue step t+NS-1
  //   into that buffer, then MFMA over buffer t % NS.
  constexpr int NS = C::NS;
  constexpr int G = ACC ? RB : RA + RB;   // LDS-DMA instructions per wave per stage (ACC: the A rows come through registers, waited for in stage())
  {
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < nk) stage(s);
    int cur = 0, nxt = NS - 1;
    for (int t = 0; t < nk; ++t) {
      const int rem = nk - 1 - t;           // groups issued after step t
      wait_ring<NS - 2, G>(rem);
      if constexpr (ACC) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's ds_writes of the A images (a raw barrier waits for no counter)
      __builtin_amdgcn_s_barrier();
      if (t + NS - 1 < nk) stage(nxt);
      compute(cur, kt0 + t);
      if (ln_at && t + 1 == ln_at) rescale_ln();
      cur = (cur + 1 == NS) ? 0 : cur + 1;
      nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  }

#ifdef SDMI_CLK_PROBE
  const unsigned long long clk_loop = __builtin_amdgcn_s_memtime() - clk_t0;
#endif
  // ---- epilogue: accumulators -> LDS (fp32, row-major [BM][BN]) -> coalesced global ----------
  float* Cs = (float*)smem;
  if (!producer)
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * C::TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int col = wn * C::TN + j * 32 + r;
        Cs[row * BN + col] = acc[i][j][e];
      }
  __syncthreads();
#ifdef SDMI_CLK_PROBE
  const unsigned long long clk_cs = __builtin_amdgcn_s_memtime() - clk_t0;
#endif

  store_tile<BM, BN, NT>(p, Cs, m0, n0, kz, tid, s_ln, tn, tiles_n, vec_off);
#ifdef SDMI_CLK_PROBE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0 && blockIdx.x < 2048) {
    g_clk_phase[blockIdx.x][0] = clk_r0; g_clk_phase[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
    g_clk_phase[blockIdx.x][2] = clk_setup; g_clk_phase[blockIdx.x][3] = clk_loop; g_clk_phase[blockIdx.x][4] = clk_cs;
    g_clk_phase[blockIdx.x][5] = __builtin_amdgcn_s_memtime() - clk_t0;
    g_clk_probe[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - clk_t0;
    g_clk_probe[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  }
#endif
}

#ifdef SDMI_CLK_PROBE
extern "C" int sdmi_dbg_read_phase(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_clk_phase), (size_t)n * 48) == hipSuccess ? 0 : -5;
}
extern "C" int sdmi_dbg_read_pre(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_clk_pre), (size_t)n * 80) == hipSuccess ? 0 : -5;
}
extern "C" int sdmi_dbg_read_clk(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_clk_probe), (size_t)n * 16) == hipSuccess ? 0 : -5;
}
#endif

// =============================================================================================
// 3x3 / stride-1 convolution with HALO REUSE (wave-specialised).
//
// The generic kernel above re-stages the activation tile for each of the 9 taps.  Profiling shows all of its
// variants pinned at ~75 GB/s per CU of LDS-DMA (bytes staged per flop), so for 3x3 convs (64 % of the UNet's GEMM
// FLOPs) this kernel walks K chunk-major -- for ci-chunk c: for tap (kh,kw) -- and stages the (TH+2) x W input rows
// that a BM = TH*W output tile needs ONCE per chunk; the 9 taps read shifted fragments of the same LDS image.
// A-side DMA drops 9x -> ~2.3x, total staged bytes per flop ~0.6x (128x128) ... 0.45x (256x128).
//   * tile = TH whole image rows (BM % W == 0, H*W % BM == 0; otherwise the launcher falls back to igemm);
//   * halo image: pixel hp = hy*(W+2) + hx, 128 B per pixel (64 channels), chunk swizzle keyed on hp; the two
//     side columns hx = 0, W+1 are image padding for every tile and are zeroed once;
//   * halo(c+1) is issued one LDS-DMA per producer wave per tap while chunk c is being multiplied;
//   * weights: one BN x 64 tile per (chunk, tap) through an NS-deep ring, exactly as in the generic kernel.
template <int BM_, int BN_, int WM_, int WN_, int NS_>
struct HCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, NS = NS_;
  static constexpr int NW = WM * WN, NT = 128 * NW;
  static constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  static constexpr int B_BYTES = BN * 128, RB = BN * 8 / (64 * NW);
  static constexpr int CS_BYTES = BM * BN * 4;
  static constexpr int NTAPH = 9 - (NS - 2);      // taps of chunk c during which halo(c+1) is issued
  static constexpr int G = RB + 1;                // LDS-DMA per producer wave per interval (weights + 1 halo piece)
  static_assert(NS == 3 || NS == 4, "ring depth");
};

// GN (round 5; GemmArgs::hgn): the conv's A operand is GroupNorm(32)(+SiLU) of the raw tensor(s) hgn.x0 | hgn.x1, applied by the
// PRODUCER waves on the way into the halo image -- "GroupNorm -> SiLU -> conv3x3" of sd/diffusion.py:173-179,199-205 as one launch,
// no normalised intermediate in memory:
//   * prologue (all waves): the statistics records the producers of x0 / x1 left (GnRec) are summed to mean / rstd per group in
//     fp64, as gn_apply_kernel of norm.hip sums them, and folded with gamma / beta into a per-channel {a, b} table in LDS
//     (y = a x + b), all C0 + C1 channels of this tile's image;
//   * a THIRD group of NW waves ("normalisers", GN only: 192 NW threads per workgroup) owns the halo image: the first chunk's
//     pieces go through registers; in the steady state the RAW piece of chunk c+1 goes to a two-slot staging area per wave by
//     LDS-DMA (fp32: two 1 KiB instructions, fp16: one) and is normalised LDS -> LDS one interval later, while the next piece's
//     DMA flies -- each lane rewrites the 16 bytes it fetched itself, so no extra barrier; border pixels stay zero (the conv pads
//     the NORMALISED tensor).  The weight-ring producer waves keep only their weight tiles (one DMA fewer per interval than in
//     the plain kernel, whose K loop is paced by their DMA issue: two forms that gave the normalisation to THEM -- behind the
//     ring's counted wait, or one interval later in front of it -- cost the 64x64 convs 19 - 25 us each, more than the
//     GroupNorm launch they replaced).
// The halo image, the fragment reads and the MFMA loop are the plain kernel's: same sums in the same order on the same fp16
// values gn_apply_kernel would have written.
// F32: the raw tensors are fp32 (the residual stream: two DMA instructions per raw piece) or fp16 (conv_feature's output: one).
template <class C, bool GN = false, bool F32 = true>
__global__ __launch_bounds__(GN ? C::NT + 64 * C::NW : C::NT) void conv3_halo_kernel(GemmArgs p, int halo_bytes) {
  sdmi_kernarg_warm<sizeof(GemmArgs) + 24>();     // + halo_bytes + the hidden grid size (gridDim.x)
  constexpr int BM = C::BM, BN = C::BN, NW = C::NW, NT = GN ? C::NT + 64 * C::NW : C::NT, NS = C::NS;
  constexpr int FM = C::FM, FN = C::FN, RB = C::RB, G = GN ? RB : C::G, NTAPH = C::NTAPH;      // GN: the weight producers issue no halo DMA
  constexpr int NR = F32 ? 2 : 1;                  // GN: raw-piece DMA instructions per interval and normaliser wave
  const int gd = GN ? p.hgn.depth : 0, gslots = gd + 1;   // GN: request-ahead distance of the raw pieces (intervals), staging slots per wave
#ifdef SDMI_CLK_PROBE
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long clk_setup = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave_id >= NW;                 // weight-ring producers AND (GN) normalisers: everybody but the MFMA waves
  const bool normaliser = GN && wave_id >= 2 * NW;
  const int wave = wave_id % NW;
  const int wm = wave / C::WN, wn = wave % C::WN;

  const int tiles_n = (p.N + BN - 1) / BN;
  int kz, tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, rr = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int L = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
    kz = p.ksplit == 1 ? 0 : (int)(((unsigned long long)L * p.tiles_magic) >> 36);      // L / tiles (launcher-computed reciprocal)
    tile = L - kz * p.tiles;
  }
  // tile order inside a K-slice (speed only): the 8 XCDs own CONTIGUOUS tile ranges.  m-major (n fastest) makes every XCD
  // stream all of W and 1/8 of A; n-major the reverse.  The launcher picks the order that moves fewer bytes through the
  // eight L2s (n-major when the weights are the larger operand: the 16x16 / 8x8 levels).
  const int tq = (int)(((unsigned long long)tile * p.tdiv_magic) >> 36), tr = tile - tq * p.tdiv;
  const int tm = p.n_major ? tr : tq;
  const int tn = p.n_major ? tq : tr;
  const int m0 = tm * BM, n0 = tn * BN;
  const int Cin = p.C0 + p.C1, nchunk = Cin >> 6;
  const int nkt = 9 * nchunk;
  const int kt0 = kz * p.ksteps_per;                 // multiple of 9 (launcher)
  const int kt1 = min(kt0 + p.ksteps_per, nkt);
  const int nk = kt1 - kt0;
  const int c_first = kt0 / 9;

  const int W = p.Wo, Hh = p.Ho, W2 = W + 2;
  const int TH = BM / W;
  const int img = m0 / (Hh * W), y0 = (m0 - img * Hh * W) / W;
  const int Hi = p.Hs << p.ups, Wi = p.Ws << p.ups;      // == Hh, W for stride 1 / pad 1

  char* hb0 = smem;
  char* hb1 = smem + halo_bytes;
  char* bring = smem + 2 * halo_bytes;
  char* dump = bring + NS * C::B_BYTES;
  // GN: [dump 1 KiB][staging: NW waves x GSLOTS x 2 KiB][table: {a, b} per channel][mean | rstd: 64 floats]
  char* const gstage = dump + 1024;
  float* const gtab = (float*)(gstage + NW * gslots * 2048);
  if constexpr (GN) {
    const HaloGn& g = p.hgn;
    const int cpg = Cin >> 5, apg = cpg / g.atom, na0 = g.C0 / g.atom, na1 = g.C1 / g.atom;
    float* const s_ms = gtab + 2 * Cin;
    double* const s_red = (double*)hb1;                 // (the second halo buffer is idle until the first chunk is being multiplied)
    constexpr int SH = NT / 32;                         // shares per group
    constexpr int NCH = (2560 + NT - 1) / NT;           // channels per thread of the table
    float gm[NCH], bt[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {                     // gamma / beta come from HBM (every weight is read once per step): requested first
      const int c = tid + k * NT;
      gm[k] = c < Cin ? g.gamma[c] : 0.f;
      bt[k] = c < Cin ? g.beta[c] : 0.f;
    }
    {
      const int gi = tid & 31, sl = tid >> 5;
      const int npair = apg * max(g.T0, g.T1);          // (atom of the group, record row) pairs; this thread: sl, sl + SH, ...
      auto rec_of = [&](int f, f32x4& v) {
        v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (f >= npair) return;
        const int t = f / apg, a = gi * apg + (f - t * apg);           // atom index in concat channel space
        const bool second = a >= na0;
        const int T = second ? g.T1 : g.T0, parts = second ? g.P1 : g.P0;
        if (t >= T) return;
        const float* rr = (second ? g.rec1 : g.rec0) + ((size_t)(img * T + t) * (second ? na1 : na0) + (second ? a - na0 : a)) * parts * 2;
        if (parts == 2) v = *(const f32x4*)rr;
        else { const f32x2 u = *(const f32x2*)rr; v[0] = u[0]; v[1] = u[1]; }
      };
      constexpr int MAXR = 6;                           // in flight at once (a concat of two 32-row-block sources at C = 640: 4 per share)
      f32x4 rv[MAXR];
#pragma unroll
      for (int k = 0; k < MAXR; ++k) rec_of(sl + SH * k, rv[k]);
      double su = 0.0, sq = 0.0;
#pragma unroll
      for (int k = 0; k < MAXR; ++k) { su += (double)rv[k][0] + (double)rv[k][2]; sq += (double)rv[k][1] + (double)rv[k][3]; }
      for (int f = sl + SH * MAXR; f < npair; f += SH) { f32x4 v; rec_of(f, v); su += (double)v[0] + (double)v[2]; sq += (double)v[1] + (double)v[3]; }
      s_red[(sl * 32 + gi) * 2] = su;
      s_red[(sl * 32 + gi) * 2 + 1] = sq;
    }
    __syncthreads();
    if (tid < 32) {
      double su = 0.0, sq = 0.0;
#pragma unroll
      for (int k = 0; k < SH; ++k) { su += s_red[(k * 32 + tid) * 2]; sq += s_red[(k * 32 + tid) * 2 + 1]; }
      const double cnt = (double)cpg * (double)(Hh * W);
      const double mean = su / cnt;
      double var = sq / cnt - mean * mean;               // E[x^2] - E[x]^2 in fp64 above the fp32 records (norm.hip gn_apply_kernel)
      var = var < 0.0 ? 0.0 : var;
      s_ms[tid] = (float)mean;
      s_ms[32 + tid] = rsqrtf((float)var + g.eps);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int c = tid + k * NT;
      if (c < Cin) {
        const int gg = c / cpg;
        const float a = s_ms[32 + gg] * gm[k];
        gtab[2 * c] = a;
        gtab[2 * c + 1] = bt[k] - s_ms[gg] * a;
      }
    }
    __syncthreads();                                    // table complete; hb1 free again
  }

  const int r = lane & 31, h = lane >> 5;
  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (producer) {
    // ---- halo pieces owned by this wave: piece q = tau*NW + wave covers 8 consecutive interior pixels ----
    const int rowp = W >> 3;                          // pieces per image row
    const int NHI = (TH + 2) * rowp;                  // pieces per chunk (<= NTAPH * NW, checked by the launcher)
    // A piece's descriptor is COMPUTED when the piece is requested (~20 integer instructions), not kept in per-tap register arrays
    // indexed through an unrolled `if (tt == tap)` chain (round 5): that chain was eight copies of the request code and an
    // eight-way branch in front of every request -- the in-kernel stamps put the producers' DMA-issue phase at 720 cycles per
    // interval for FIVE instructions (the four weight pieces alone: 210), on the role that paces the K loop.
    const int rp_magic = (65536 + rowp - 1) / rowp;   // q / rowp == (q * rp_magic) >> 16 for q < 64, rowp <= 8
    struct Piece { int pix, gch, lds; bool in; };
    auto piece_of = [&](int t) {
      const int q = t * NW + wave;
      const int hy = (q * rp_magic) >> 16, seg = q - hy * rowp;
      const int hx = 1 + seg * 8 + (lane >> 3);       // interior columns 1..W
      const int hp = hy * W2 + hx;
      const int y = y0 - 1 + hy, x = hx - 1;
      Piece d;
      d.in = q < NHI && (unsigned)y < (unsigned)Hi && (unsigned)x < (unsigned)Wi;
      d.pix = img * p.Hs * p.Ws + (y >> p.ups) * p.Ws + (x >> p.ups);
      d.gch = ((lane & 7) ^ ((hp >> 1) & 7)) * 8;
      d.lds = q < NHI ? (hy * W2 + 1 + seg * 8) * 128 : -1;      // wave-uniform LDS byte offset of the piece
      return d;
    };
    auto halo_piece = [&](int t, int chunk, char* hb) {
      const Piece d = piece_of(t);
      const int cabs = chunk << 6;
      const bool second = cabs >= p.C0;
      const f16* base = second ? p.a1 : p.a0;
      const int ld = second ? p.lda1 : p.lda0;
      const int cc = second ? cabs - p.C0 : cabs;
      const int lds_off = __builtin_amdgcn_readfirstlane(d.lds);
      const f16* gz = p.zero + d.gch;
      const f16* g = d.in ? base + ((size_t)d.pix * ld + cc + d.gch) : gz;
      glds16(lds_off >= 0 ? g : gz, lds_off >= 0 ? hb + lds_off : dump);
    };
    // ---- GN: raw piece -> normalised fp16 piece ----
    // byte address of this lane's 8 raw channels of piece t in chunk `chunk` (nullptr: border pixel / no such piece)
    auto raw_src = [&](const Piece& d, int chunk) -> const char* {
      const int cabs = chunk << 6;
      const bool second = cabs >= p.hgn.C0;
      const char* base = (const char*)(second ? p.hgn.x1 : p.hgn.x0);
      const int ld = second ? p.hgn.C1 : p.hgn.C0;
      const int cc = second ? cabs - p.hgn.C0 : cabs;
      const size_t el = (size_t)d.pix * ld + cc + d.gch;
      return (d.in && d.lds >= 0) ? base + el * (F32 ? 4 : 2) : nullptr;
    };
    // y = SiLU(a x + b) of the lane's 8 channels (table entries of concat channels chunk*64 + gch ..), zero outside the image,
    // written where the LDS-DMA of the plain kernel would have put the lane's 16 bytes
    auto norm_store = [&](const Piece& d, int chunk, char* hb, const float (&x)[8]) {
      const int lds_off = __builtin_amdgcn_readfirstlane(d.lds);
      if (lds_off < 0) return;
      const float* tp = gtab + 2 * ((chunk << 6) + d.gch);
      f32x4 tv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) tv[k] = *(const f32x4*)(tp + 4 * k);          // {a0 b0 a1 b1} ...
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = fmaf(x[e], tv[e >> 1][2 * (e & 1)], tv[e >> 1][2 * (e & 1) + 1]);
        if (p.hgn.silu) y = y * __builtin_amdgcn_rcpf(1.f + __expf(-y));
        o[e] = d.in ? (f16)y : (f16)0.f;
      }
      *(f16x8*)(hb + lds_off + lane * 16) = o;
    };
    char* const my_stage = gstage + wave * (gslots * 2048);
    auto raw_dma = [&](int t, int chunk, int slot) {      // NR LDS-DMA instructions, always (constant group size for the counted waits)
      const char* src = raw_src(piece_of(t), chunk);
      char* sl = my_stage + slot * 2048;
      const char* z = (const char*)p.zero + (lane & 7) * 16;
      glds16(src ? src : z, sl);
      if constexpr (F32) glds16(src ? src + 16 : z, sl + 1024);
    };
    auto norm_from_stage = [&](int t, int chunk, char* hb, int slot) {
      const char* sl = my_stage + slot * 2048 + lane * 16;
      float x[8];
      if constexpr (F32) {
        const f32x4 u0 = *(const f32x4*)sl, u1 = *(const f32x4*)(sl + 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = u0[e]; x[4 + e] = u1[e]; }
      } else {
        const f16x8 u = *(const f16x8*)sl;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (float)u[e];
      }
      norm_store(piece_of(t), chunk, hb, x);
    };
    // ---- weight tile pointers: w[n][tap*Cin + chunk*64 + ...] ----
    const f16* b_ptr[RB];
    int b_ok[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int q = (i * NW + wave) * 64 + lane;
      const int row = q >> 3, pc = q & 7;
      const int gch = (pc ^ ((row >> 1) & 7)) * 8;
      const int n = n0 + row;
      b_ok[i] = n < p.N;
      b_ptr[i] = b_ok[i] ? p.w + (size_t)n * p.ldw + (size_t)c_first * 64 + gch : p.zero + gch;
    }
    int tap = 0, chunk = c_first;
    auto stage_b = [&](int slot) {
      char* sb = bring + slot * C::B_BYTES;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        glds16(b_ptr[i], sb + (i * NW + wave) * 1024);
        if (b_ok[i]) b_ptr[i] += (tap == 8) ? (64 - 8 * Cin) : Cin;
      }
      if (++tap == 9) { tap = 0; ++chunk; }
    };
    // GN: the first chunk's pieces go through registers (all of a wave's loads in flight at once, then converted); the normaliser
    // wave q takes the even pieces and the weight producer wave q the odd ones, so the set-up is half as long as with one role.
    auto first_chunk = [&](auto PAR) {
      constexpr int par = decltype(PAR)::value, NH = (NTAPH + 1 - par) / 2;
      f32x4 r0[NH > 0 ? NH : 1], r1[NH > 0 ? NH : 1];
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        const char* src = raw_src(piece_of(2 * k + par), c_first);
        r0[k] = f32x4{0.f, 0.f, 0.f, 0.f}; r1[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (src) {
          r0[k] = *(const f32x4*)src;
          if constexpr (F32) r1[k] = *(const f32x4*)(src + 16);
        }
      }
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        float x[8];
        if constexpr (F32) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { x[e] = r0[k][e]; x[4 + e] = r1[k][e]; }
        } else {
          const f16x8 u = __builtin_bit_cast(f16x8, r0[k]);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = (float)u[e];
        }
        norm_store(piece_of(2 * k + par), c_first, hb0, x);
      }
    };
    if (normaliser) {
      // ================= GN: the normaliser waves own the halo image =================
      if constexpr (GN) {
        first_chunk(std::integral_constant<int, 0>{});    // the even pieces of the first chunk (the weight producers take the odd ones)
        // The raw piece that is normalised in interval u (tap k = u % 9 < NTAPH of chunk c: piece k of chunk c+1) is requested gd
        // intervals earlier -- the tensors were just written by another kernel, a piece takes 1 - 2 us to arrive from the other
        // XCDs' L2 / the Infinity Cache, two to four intervals of this loop (a first form that gave it ONE interval waited for it
        // every time: +17 us per 64x64 conv).  EVERY interval requests exactly NR instructions (a dummy when there is no piece),
        // so "piece of interval u has landed" is the constant count vmcnt(gd x NR).
        int itap = 0, ichunk = c_first;                 // (chunk, tap) of the interval whose piece is REQUESTED next
        auto request = [&](int u) {
          const bool real = itap < NTAPH && (ichunk + 1) * 9 < kt1 && u < nk;
          if (real) {
            raw_dma(itap, ichunk + 1, u % gslots);
          } else {
            glds16(p.zero, dump);
            if constexpr (F32) glds16(p.zero, dump);
          }
          if (++itap == 9) { itap = 0; ++ichunk; }
        };
        for (int u = 0; u < gd; ++u) request(u);        // the pieces of the first gd intervals, behind the first chunk's own loads
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int ctap = 0, cchunk = c_first;                 // (chunk, tap) being CONSUMED in interval t
#ifdef SDMI_CLK_PROBE_FINE
        unsigned long long n_req = 0, n_vm = 0, n_cv = 0, n_bar = 0;
#endif
        for (int t = 0; t < nk; ++t) {
#ifdef SDMI_CLK_PROBE_FINE
          const unsigned long long q0 = __builtin_amdgcn_s_memtime();
#endif
          request(t + gd);
#ifdef SDMI_CLK_PROBE_FINE
          const unsigned long long q1 = __builtin_amdgcn_s_memtime();
#endif
          switch (gd * NR) {                            // all but the newest gd x NR instructions have landed: interval t's piece is there
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
          }
#ifdef SDMI_CLK_PROBE_FINE
          const unsigned long long q2 = __builtin_amdgcn_s_memtime();
#endif
          if (ctap < NTAPH && (cchunk + 1) * 9 < kt1) {
            char* const hbn = ((cchunk + 1 - c_first) & 1) ? hb1 : hb0;
            norm_from_stage(ctap, cchunk + 1, hbn, t % gslots);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef SDMI_CLK_PROBE_FINE
          const unsigned long long q3 = __builtin_amdgcn_s_memtime();
#endif
          __builtin_amdgcn_s_barrier();
#ifdef SDMI_CLK_PROBE_FINE
          n_req += q1 - q0; n_vm += q2 - q1; n_cv += q3 - q2; n_bar += __builtin_amdgcn_s_memtime() - q3;
#endif
          if (++ctap == 9) { ctap = 0; ++cchunk; }
        }
#ifdef SDMI_CLK_PROBE_FINE
        if (lane == 0 && wave == 0 && blockIdx.x < 512) {
          g_clk_pre[blockIdx.x][0] = n_req; g_clk_pre[blockIdx.x][1] = n_vm; g_clk_pre[blockIdx.x][2] = n_cv; g_clk_pre[blockIdx.x][3] = n_bar;
        }
#endif
      }
    } else {
    // ================= weight-ring producers (plain kernel: they also request the halo pieces) =================
    // prologue: whole halo of the first chunk + first NS-1 weight tiles
    if constexpr (!GN) {
#pragma unroll
      for (int t = 0; t < NTAPH; ++t) halo_piece(t, c_first, hb0);
    }
#pragma unroll
    for (int s2 = 0; s2 < NS - 1; ++s2)
      if (s2 < nk) stage_b(s2);
    if constexpr (GN) first_chunk(std::integral_constant<int, 1>{});      // behind the weight tiles' requests
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int nxt = NS - 1;
    int ctap = 0, cchunk = c_first;                   // (chunk, tap) being CONSUMED in interval t
#ifdef SDMI_CLK_PROBE_FINE
    unsigned long long acc_issue = 0, acc_vm = 0, acc_bar = 0;
#endif
    for (int t = 0; t < nk; ++t) {
#ifdef SDMI_CLK_PROBE_FINE
      const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#endif
      if (t + NS - 1 < nk) {
        stage_b(nxt);
        if constexpr (!GN) {
          // one halo piece of the NEXT chunk (dummy DMA keeps the per-interval count constant)
          const bool have_next = (cchunk + 1) * 9 < kt1;
          if (ctap < NTAPH && have_next) {
            halo_piece(ctap, cchunk + 1, ((cchunk + 1 - c_first) & 1) ? hb1 : hb0);
          } else {
            glds16(p.zero, dump);
          }
        }
      }
#ifdef SDMI_CLK_PROBE_FINE
      const unsigned long long s1 = __builtin_amdgcn_s_memtime();
#endif
      const int rem = nk - 2 - t;
      if (NS >= 4 && rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
      else if (rem >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SDMI_CLK_PROBE_FINE
      const unsigned long long s2 = __builtin_amdgcn_s_memtime();
#endif
      __builtin_amdgcn_s_barrier();
#ifdef SDMI_CLK_PROBE_FINE
      acc_issue += s1 - s0; acc_vm += s2 - s1; acc_bar += __builtin_amdgcn_s_memtime() - s2;
#endif
      nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
      if (++ctap == 9) { ctap = 0; ++cchunk; }
    }
#ifdef SDMI_CLK_PROBE_FINE
    if (lane == 0 && wave == 0 && blockIdx.x < 512) {
      g_clk_probe[512 + blockIdx.x][0] = acc_issue; g_clk_probe[512 + blockIdx.x][1] = acc_vm;
      g_clk_probe[1024 + blockIdx.x][0] = acc_bar;
    }
#endif
    }   // weight-ring producers
  } else {
    // ---- consumers: zero the padding columns of both halo buffers once, then MFMA ----
    for (int i = tid; i < (TH + 2) * 2 * 8 * 2; i += NW * 64) {       // (row, side, 16-B chunk, buffer)
      const int buf = i & 1, ch = (i >> 1) & 7, side = (i >> 4) & 1, hy = i >> 5;
      const int hp = hy * W2 + (side ? W + 1 : 0);
      f16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (f16)0.f;
      *(f16x8*)((buf ? hb1 : hb0) + hp * 128 + ch * 16) = z;
    }
    int base_hp[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int rr = wm * C::TM + i * 32 + r;
      const int ty = rr / W, tx = rr - ty * W;
      base_hp[i] = ty * W2 + tx;
    }
    const int bkey = (r >> 1) & 7;
    const int b_row_off = (wn * C::TN + r) * 128;
    const int bco = (h ^ bkey) << 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef SDMI_CLK_PROBE
    if (tid == 0) clk_setup = __builtin_amdgcn_s_memtime() - clk_t0;
#endif
    int cur = 0, ctap = 0, cchunk = 0;
#ifdef SDMI_CLK_PROBE_FINE
    unsigned long long acc_cmp = 0, acc_cbar = 0;
#endif
    for (int t = 0; t < nk; ++t) {
#ifdef SDMI_CLK_PROBE_FINE
      const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#endif
      const int kh = ctap / 3, kw = ctap - kh * 3;
      const char* hb = (cchunk & 1) ? hb1 : hb0;
      const char* Bs = bring + cur * C::B_BYTES + b_row_off;
      int a_off[FM], a_co[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int hp = base_hp[i] + kh * W2 + kw;
        a_off[i] = hp * 128;
        a_co[i] = (h ^ ((hp >> 1) & 7)) << 4;
      }
      f16x8 af[2][FM], bf[2][FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[0][i] = *(const f16x8*)(hb + a_off[i] + a_co[i]);
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[0][j] = *(const f16x8*)(Bs + j * 32 * 128 + bco);
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        if (s2 < 3) {
#pragma unroll
          for (int i = 0; i < FM; ++i) af[(s2 + 1) & 1][i] = *(const f16x8*)(hb + a_off[i] + (a_co[i] ^ ((s2 + 1) << 5)));
#pragma unroll
          for (int j = 0; j < FN; ++j) bf[(s2 + 1) & 1][j] = *(const f16x8*)(Bs + j * 32 * 128 + (bco ^ ((s2 + 1) << 5)));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s2 & 1][i], bf[s2 & 1][j], acc[i][j], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef SDMI_CLK_PROBE_FINE
      const unsigned long long s1 = __builtin_amdgcn_s_memtime();
#endif
      __builtin_amdgcn_s_barrier();
#ifdef SDMI_CLK_PROBE_FINE
      acc_cmp += s1 - s0; acc_cbar += __builtin_amdgcn_s_memtime() - s1;
#endif
      cur = (cur + 1 == NS) ? 0 : cur + 1;
      if (++ctap == 9) { ctap = 0; ++cchunk; }
    }
#ifdef SDMI_CLK_PROBE_FINE
    if (lane == 0 && wave == 0 && blockIdx.x < 512) {
      g_clk_probe[1536 + blockIdx.x][0] = acc_cmp; g_clk_probe[1536 + blockIdx.x][1] = acc_cbar;
    }
#endif
  }

#ifdef SDMI_CLK_PROBE
  const unsigned long long clk_loop = __builtin_amdgcn_s_memtime() - clk_t0;
#endif
  float* Cs = (float*)smem;
  if (!producer)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm * C::TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const int col = wn * C::TN + j * 32 + r;
          Cs[row * BN + col] = acc[i][j][e];
        }
  __syncthreads();
#ifdef SDMI_CLK_PROBE
  const unsigned long long clk_cs = __builtin_amdgcn_s_memtime() - clk_t0;
#endif
  store_tile<BM, BN, NT>(p, Cs, m0, n0, kz, tid);
#ifdef SDMI_CLK_PROBE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0 && blockIdx.x < 2048) {
    g_clk_phase[blockIdx.x][0] = clk_r0; g_clk_phase[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
    g_clk_phase[blockIdx.x][2] = clk_setup; g_clk_phase[blockIdx.x][3] = clk_loop; g_clk_phase[blockIdx.x][4] = clk_cs;
    g_clk_phase[blockIdx.x][5] = __builtin_amdgcn_s_memtime() - clk_t0;
  }
#endif
}

// out = sum_z slab[z] + bias + res  (same epilogue semantics as the fused path): one item = 8 columns n .. n+7 of slab row ms.
// Returns true with the values the GroupNorm of this tensor will read in x (the non-transposed path only).
__device__ __forceinline__ bool finalize_item(const GemmArgs& p, int ms, int n, size_t MN, float (&x)[8]) {
  {
    const int m = out_row(p, ms);                               // output row (phase2 scatter; identity otherwise)
    float v[8], va[8];                 // va: slabs of the LayerNorm-folded K range (partial fold + split-K), v: the rest
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = 0.f; va[e] = 0.f; }
    const int n_fold = (p.ln_stat != nullptr && p.ln_ksteps > 0) ? p.ln_ksteps / p.ksteps_per : 0;
    const bool transposed = p.outT != nullptr && n >= p.nt0;
    // residual issued first: it is the coldest load of the item
    f32x4 rf0 = {0.f, 0.f, 0.f, 0.f}, rf1 = {0.f, 0.f, 0.f, 0.f};
    f16x8 rh = {};
    if (p.res && !transposed) {
      if (p.res_f32) {
        const float* rp = (const float*)p.res + (size_t)m * p.ldr + n;
        rf0 = *(const f32x4*)rp;
        rf1 = *(const f32x4*)(rp + 4);
      } else {
        rh = *(const f16x8*)((const f16*)p.res + (size_t)m * p.ldr + n);
      }
    }
    // bias with the first slab loads (it was read element by element after the last slab: one more exposed latency)
    f32x4 bv0 = {0.f, 0.f, 0.f, 0.f}, bv1 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) { bv0 = *(const f32x4*)(p.bias + n); bv1 = *(const f32x4*)(p.bias + n + 4); }
    // four slabs' loads in flight at a time (a load-add-load-add chain paid one memory latency per slab); the
    // additions stay in slab order, so the result is bit-identical
    for (int z0 = 0; z0 < p.ksplit; z0 += 4) {
      f32x4 s0[4], s1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (z0 + j < p.ksplit) {
          const float* sp = p.slab + (size_t)(z0 + j) * MN + (size_t)ms * p.N + n;
          s0[j] = *(const f32x4*)sp;
          s1[j] = *(const f32x4*)(sp + 4);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (z0 + j < p.ksplit) {
          if (z0 + j < n_fold) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] += s0[j][e]; va[4 + e] += s1[j][e]; }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += s0[j][e]; v[4 + e] += s1[j][e]; }
          }
        }
      }
    }
    if (n_fold > 0) {
      const float mean = p.ln_out[2 * ms], rstd = p.ln_out[2 * ms + 1];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rstd * (va[e] - mean * p.ln_g[n + e]);
    }
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += bv0[e]; v[4 + e] += bv1[e]; }
    }
    if (n < p.cs_hi) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= p.cscale;
    }
    if (p.act == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[e]));
    }
    if (transposed) {
      const int Ct = p.N - p.nt0;
      const int b = m / p.S, s = m - b * p.S;
#pragma unroll
      for (int e = 0; e < 8; ++e) p.outT[((size_t)b * Ct + (n + e - p.nt0)) * p.ldt + vt_pos(s, p.tperm)] = (f16)v[e];
      return false;
    }
    if (p.res) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += rf0[e] + (float)rh[e];
        v[4 + e] += rf1[e] + (float)rh[4 + e];
      }
    }
    f16x8 o16;
#pragma unroll
    for (int e = 0; e < 8; ++e) o16[e] = (f16)v[e];
    if (p.out_f32) {
      float* op = (float*)p.out + (size_t)m * p.ldc + n;
#pragma unroll
      for (int e = 0; e < 8; ++e) op[e] = v[e];
      if (p.out16) *(f16x8*)(p.out16 + (size_t)m * p.ldc + n) = o16;
    } else {
      *(f16x8*)((f16*)p.out + (size_t)m * p.ldc + n) = o16;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = p.out_f32 ? v[e] : (float)o16[e];
    return true;
  }
}

__global__ __launch_bounds__(256) void splitk_finalize_kernel(GemmArgs p) {
  sdmi_kernarg_warm<sizeof(GemmArgs)>();
  const unsigned total8 = (unsigned)p.M * (unsigned)(p.N / 8);      // < 2^31 (launcher): 32-bit index math, no 64-bit division
  const unsigned n8 = (unsigned)(p.N / 8);
  const size_t MN = (size_t)p.M * p.N;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total8; idx += gridDim.x * 256u) {
    const unsigned mq = idx / n8;
    float x[8];
    (void)finalize_item(p, (int)mq, (int)(idx - mq * n8) * 8, MN, x);
  }
}

// The same combine with the GroupNorm statistics of the result (GnRec, common.h) taken on the way: a workgroup of 640 threads
// takes `rows_wg` consecutive rows of one image and one 320-column slice of them, thread t always column chunk t % 40, so the
// moments of its 8 columns stay in registers; one record per thread in LDS, then one thread per atom of the slice adds
// its chunks' records in a fixed order (fp64) and stores them as this row block's record.
constexpr int kFinNT = 640, kFinCols = 320;     // 40 column chunks x 16 rows per pass (320 threads and twice the items per thread: 1 - 2 us slower per launch)
template <int NI>          // rows per workgroup = 16 NI: the NI items of a thread are unrolled, so their slab loads fly together
__global__ __launch_bounds__(kFinNT) void splitk_finalize_gacc_kernel(GemmArgs p, int nsl) {
  sdmi_kernarg_warm<sizeof(GemmArgs) + 8>();
  __shared__ float s_g[kFinNT * 4];
  const int tid = threadIdx.x;
  constexpr int n8 = kFinCols / 8, rpp = kFinNT / n8;   // 40 column chunks, 16 rows per pass
  constexpr int rows_wg = NI * rpp;
  const int rb = blockIdx.x / nsl, cs = blockIdx.x - rb * nsl;
  const int c8 = tid % n8, r0 = tid / n8;
  const int n = cs * kFinCols + c8 * 8;
  const size_t MN = (size_t)p.M * p.N;
  const int row_base = rb * rows_wg;
  const int atom = p.gacc.atom;
  const int split = min(8, (gnrec_div_atom(p.gacc, n) + 1) * atom - n);
  GaccThread<> gth;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int ms = row_base + r0 + i * rpp;      // < M: rows_wg divides the rows of an image (launcher)
    float x[8];
    if (finalize_item(p, ms, n, MN, x)) gth.add(x, split);
  }
  *(f32x4*)(s_g + tid * 4) = gth.parts(split);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (not __syncthreads(): its vmcnt(0) would wait for the output stores to retire)
  __builtin_amdgcn_s_barrier();
  const int aps = gnrec_div_atom(p.gacc, kFinCols);     // atoms per slice (the slice starts on an atom boundary)
  if (tid < aps) {
    const int lo_col = tid * atom, hi_col = (tid + 1) * atom - 1;      // slice-local columns
    const int c_lo = lo_col >> 3, c_hi = hi_col >> 3;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      const int c = min(c_lo + cc, c_hi);
      const bool use = c_lo + cc <= c_hi;
      const int sel = (gnrec_div_atom(p.gacc, 8 * c) == tid) ? 0 : 2;
      f32x2 t[rpp];
#pragma unroll
      for (int k = 0; k < rpp; ++k) t[k] = *(const f32x2*)(s_g + (k * n8 + c) * 4 + sel);
#pragma unroll
      for (int k = 0; k < rpp; ++k) { s1 += use ? t[k][0] : 0.f; s2 += use ? t[k][1] : 0.f; }
    }
    const bool phased = p.gacc.mod != p.M;
    const int ph = phased ? row_base / p.gacc.mod : 0;
    const int mm = row_base - ph * p.gacc.mod;
    const int img = gnrec_div_rows(p.gacc, mm);
    const int t_row = ph * (p.gacc.rows_img / rows_wg) + (mm - img * p.gacc.rows_img) / rows_wg;
    ((f32x2*)p.gacc.rec)[(size_t)(img * p.gacc.T + t_row) * p.gacc.natoms + cs * aps + tid] = f32x2{s1, s2};
  }
}

struct CfgInfo {
  const char* name;
  int BM, BN, NS, NT, LDS;
  void (*kern)(GemmArgs);
  void (*hkern)(GemmArgs, int);   // halo-reuse 3x3 kernel (kern == nullptr)
  int ntaph, nw;
  void (*kern_gna)(GemmArgs);     // the same tile with GroupNorm on the A fragments (GemmArgs::gna_rec), or nullptr
  void (*kern_acc)(GemmArgs);     // the same tile with the fp32 A operand as a hi + lo fp16 pair (GemmArgs::accurate), or nullptr
  int LDS_acc;
  void (*hkern_gn)(GemmArgs, int); // halo kernel that normalises its own A operand (GemmArgs::hgn) from fp32 tensors, or nullptr
  void (*hkern_gn16)(GemmArgs, int); // ... from fp16 tensors
};

#define CFG_ENTRY(BM, BN, WM, WN, NS) \
  {"t" #BM "x" #BN "s" #NS, BM, BN, NS, Cfg<BM, BN, WM, WN, NS>::NT, Cfg<BM, BN, WM, WN, NS>::LDS, igemm_kernel<Cfg<BM, BN, WM, WN, NS>>, nullptr, 0, 0}
#define ACC_LDS(BM, BN, WM, WN, NS) \
  ((NS * (Cfg<BM, BN, WM, WN, NS>::STAGE + Cfg<BM, BN, WM, WN, NS>::A_BYTES)) > Cfg<BM, BN, WM, WN, NS>::CS_BYTES ? \
   (NS * (Cfg<BM, BN, WM, WN, NS>::STAGE + Cfg<BM, BN, WM, WN, NS>::A_BYTES)) : Cfg<BM, BN, WM, WN, NS>::CS_BYTES)
#define CFG_ENTRY_A(BM, BN, WM, WN, NS) \
  {"t" #BM "x" #BN "s" #NS, BM, BN, NS, Cfg<BM, BN, WM, WN, NS>::NT, Cfg<BM, BN, WM, WN, NS>::LDS, igemm_kernel<Cfg<BM, BN, WM, WN, NS>>, nullptr, 0, 0, \
   nullptr, igemm_kernel<Cfg<BM, BN, WM, WN, NS>, false, true>, ACC_LDS(BM, BN, WM, WN, NS)}
#define CFG_ENTRY_W(BM, BN, WM, WN, NS, TAG) \
  {"t" #BM "x" #BN "s" #NS TAG, BM, BN, NS, Cfg<BM, BN, WM, WN, NS>::NT, Cfg<BM, BN, WM, WN, NS>::LDS, igemm_kernel<Cfg<BM, BN, WM, WN, NS>>, nullptr, 0, 0}
#define CFG_ENTRY_P(BM, BN, WM, WN, NS) \
  {"t" #BM "x" #BN "s" #NS "p", BM, BN, NS, Cfg<BM, BN, WM, WN, NS, 2>::NT, Cfg<BM, BN, WM, WN, NS, 2>::LDS, igemm_kernel<Cfg<BM, BN, WM, WN, NS, 2>>, nullptr, 0, 0}
#define CFG_ENTRY_W2(BM, BN, WM, WN, NS, TAG) \
  {"t" #BM "x" #BN "s" #NS "p" TAG, BM, BN, NS, Cfg<BM, BN, WM, WN, NS, 2>::NT, Cfg<BM, BN, WM, WN, NS, 2>::LDS, igemm_kernel<Cfg<BM, BN, WM, WN, NS, 2>>, nullptr, 0, 0}
#define CFG_ENTRY_Q(BM, BN, WM, WN, NS, PW) \
  {"t" #BM "x" #BN "s" #NS "q" #PW, BM, BN, NS, Cfg<BM, BN, WM, WN, NS, 2, 1, PW>::NT, Cfg<BM, BN, WM, WN, NS, 2, 1, PW>::LDS, igemm_kernel<Cfg<BM, BN, WM, WN, NS, 2, 1, PW>>, nullptr, 0, 0}
#define CFG_ENTRY_P2(BM, BN, WM, WN, NS, KPI) \
  {"t" #BM "x" #BN "s" #NS "p" #KPI, BM, BN, NS, Cfg<BM, BN, WM, WN, NS, 2, KPI>::NT, Cfg<BM, BN, WM, WN, NS, 2, KPI>::LDS, igemm_kernel<Cfg<BM, BN, WM, WN, NS, 2, KPI>>, nullptr, 0, 0}
#define CFG_ENTRY_PG(BM, BN, WM, WN, NS) \
  {"t" #BM "x" #BN "s" #NS "p", BM, BN, NS, Cfg<BM, BN, WM, WN, NS, 2>::NT, Cfg<BM, BN, WM, WN, NS, 2>::LDS, igemm_kernel<Cfg<BM, BN, WM, WN, NS, 2>>, nullptr, 0, 0, \
   igemm_kernel<Cfg<BM, BN, WM, WN, NS, 2>, true>}
#define CFG_ENTRY_QG(BM, BN, WM, WN, NS, PW) \
  {"t" #BM "x" #BN "s" #NS "q" #PW, BM, BN, NS, Cfg<BM, BN, WM, WN, NS, 2, 1, PW>::NT, Cfg<BM, BN, WM, WN, NS, 2, 1, PW>::LDS, igemm_kernel<Cfg<BM, BN, WM, WN, NS, 2, 1, PW>>, nullptr, 0, 0, \
   igemm_kernel<Cfg<BM, BN, WM, WN, NS, 2, 1, PW>, true>}
const CfgInfo kCfgs[] = {
    // ("A": also built with the wide A operand of the accurate mode, GemmArgs::accurate)
    CFG_ENTRY(128, 128, 2, 2, 2), CFG_ENTRY_A(128, 128, 2, 2, 3), CFG_ENTRY(128, 128, 2, 2, 4),
    CFG_ENTRY(128, 64, 2, 2, 2),  CFG_ENTRY(128, 64, 2, 2, 4),
    CFG_ENTRY(64, 128, 2, 2, 2),  CFG_ENTRY_A(64, 128, 2, 2, 4),
    CFG_ENTRY(64, 64, 2, 2, 2),   CFG_ENTRY(64, 64, 2, 2, 3),  CFG_ENTRY_A(64, 64, 2, 2, 4),
    CFG_ENTRY(256, 128, 4, 2, 2), CFG_ENTRY(256, 128, 4, 2, 3),
    CFG_ENTRY(128, 256, 2, 4, 2), CFG_ENTRY(128, 256, 2, 4, 3),
    // 8/16-wave variants: two waves per SIMD so DMA issue / LDS latency of one hides under the other's MFMAs
    CFG_ENTRY_W(128, 128, 2, 4, 2, "w8"), CFG_ENTRY_W(128, 128, 2, 4, 3, "w8"), CFG_ENTRY_W(128, 128, 4, 2, 3, "w8m"),
    CFG_ENTRY_W(128, 64, 4, 2, 2, "w8"),  CFG_ENTRY_W(64, 128, 2, 4, 2, "w8"),
    CFG_ENTRY_W(256, 128, 4, 4, 2, "w16"), CFG_ENTRY_W(256, 128, 4, 4, 3, "w16"),
    CFG_ENTRY_W(128, 256, 4, 4, 2, "w16"),
    // wave-specialised (producer/consumer) rings
    // ("G": also built with GroupNorm on the A fragments -- the tiles the attention blocks' conv_input GEMMs are planned with)
    CFG_ENTRY_P(128, 128, 2, 2, 3), CFG_ENTRY_P(128, 128, 2, 2, 4), CFG_ENTRY_PG(64, 64, 2, 2, 4),
    CFG_ENTRY_P(128, 64, 2, 2, 4),  CFG_ENTRY_PG(64, 128, 2, 2, 4),
    CFG_ENTRY_P(256, 128, 4, 2, 3), CFG_ENTRY_P(128, 256, 2, 4, 3),
    // specialised, 8 consumer waves (two MFMA waves per SIMD) + 8 producer waves
    CFG_ENTRY_W2(128, 128, 2, 4, 3, "c8"), CFG_ENTRY_W2(128, 128, 4, 2, 3, "c8m"), CFG_ENTRY_W2(128, 128, 2, 4, 4, "c8"),
    CFG_ENTRY_W2(128, 64, 4, 2, 4, "c8"),  CFG_ENTRY_W2(64, 128, 2, 4, 4, "c8"),
    // (deeper rings, NS = 6..8, were measured on the weight-streaming M = 128 shapes: no gain over NS = 4)
    // deeper rings for the 1x1 GEMMs whose A operand streams from beyond L2 (21 MB GeGLU tensor at 64x64): with two
    // stages in flight a 128x128 step waits ~1.5 us of loaded Infinity-Cache latency per 32 KiB (measured 2050 cycles
    // per K-step on M=8192,N=320,K=1600 against ~1040 on the L2-resident 3x3 convs)
    CFG_ENTRY_P(64, 64, 2, 2, 6), CFG_ENTRY_P(128, 64, 2, 2, 6), CFG_ENTRY_P(64, 128, 2, 2, 6), CFG_ENTRY_P(64, 64, 2, 2, 8),
    CFG_ENTRY_W2(128, 64, 4, 2, 6, "c8"),  CFG_ENTRY_W2(64, 128, 2, 4, 6, "c8"),
    // 160-wide tiles: N = 320 / 640 / 1280 divide by 160, not by 128 (17 % of a 128-wide tiling of N = 320 multiplies
    // padding), and 64 m-tiles x 2 n-tiles x split-K 2 is exactly one workgroup per CU at 64x64.  3x3 convs only.
    CFG_ENTRY_P(128, 160, 4, 1, 3), CFG_ENTRY_P(128, 160, 4, 1, 4), CFG_ENTRY_P(64, 160, 2, 1, 4),
    // two K-steps per barrier interval
    CFG_ENTRY_P2(128, 128, 2, 2, 2, 2), CFG_ENTRY_P2(64, 64, 2, 2, 3, 2), CFG_ENTRY_P2(64, 64, 2, 2, 2, 4),
    CFG_ENTRY_P2(128, 64, 2, 2, 3, 2),  CFG_ENTRY_P2(64, 128, 2, 2, 3, 2),
    // 32-row tiles, one head (128 columns) wide: the softmax GEMM of the folded cross-attention has only N / 128 = 8
    // n-tiles and no split-K, so at 16x16 / 8x8 (M = 512 / 128) the m-tile count is all the parallelism there is
    CFG_ENTRY_P(32, 128, 1, 4, 4), CFG_ENTRY_P(32, 128, 1, 4, 6), CFG_ENTRY(32, 128, 1, 4, 4),
    // two producer waves per MFMA wave ("q2"): the K = C GEMMs' loop is paced by DMA issue per wave
    CFG_ENTRY_QG(64, 64, 2, 2, 4, 2), CFG_ENTRY_QG(64, 64, 2, 2, 6, 2), CFG_ENTRY_QG(64, 128, 2, 2, 4, 2), CFG_ENTRY_Q(128, 64, 2, 2, 4, 2),
    CFG_ENTRY_Q(128, 128, 2, 2, 3, 2), CFG_ENTRY_Q(128, 128, 2, 2, 4, 2),
};
#define CFG_ENTRY_H(BM, BN, WM, WN, NS) \
  {"h" #BM "x" #BN "s" #NS, BM, BN, NS, HCfg<BM, BN, WM, WN, NS>::NT, 0, nullptr, conv3_halo_kernel<HCfg<BM, BN, WM, WN, NS>>, \
   HCfg<BM, BN, WM, WN, NS>::NTAPH, HCfg<BM, BN, WM, WN, NS>::NW, nullptr, nullptr, 0, nullptr, nullptr}
// ("N": also built with GroupNorm(+SiLU) applied to its own A operand, GemmArgs::hgn -- the tiles the 64x64 / 32x32 / 16x16
// convs of a step are planned with)
#define CFG_ENTRY_HN(BM, BN, WM, WN, NS) \
  {"h" #BM "x" #BN "s" #NS, BM, BN, NS, HCfg<BM, BN, WM, WN, NS>::NT, 0, nullptr, conv3_halo_kernel<HCfg<BM, BN, WM, WN, NS>>, \
   HCfg<BM, BN, WM, WN, NS>::NTAPH, HCfg<BM, BN, WM, WN, NS>::NW, nullptr, nullptr, 0, conv3_halo_kernel<HCfg<BM, BN, WM, WN, NS>, true, true>, \
   conv3_halo_kernel<HCfg<BM, BN, WM, WN, NS>, true, false>}
const CfgInfo kHaloCfgs[] = {
    CFG_ENTRY_HN(128, 128, 2, 2, 3), CFG_ENTRY_H(128, 64, 2, 2, 3), CFG_ENTRY_H(64, 64, 2, 2, 3), CFG_ENTRY_H(64, 128, 2, 2, 3),
    CFG_ENTRY_H(256, 128, 4, 2, 3), CFG_ENTRY_HN(128, 128, 2, 2, 4), CFG_ENTRY_H(256, 64, 4, 2, 3),
    CFG_ENTRY_HN(128, 160, 4, 1, 3),
};
constexpr int kNumHalo = sizeof(kHaloCfgs) / sizeof(kHaloCfgs[0]);
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);
constexpr int kMaxDev = 16;
bool g_attr_done[kMaxDev][kNumCfgs + kNumHalo] = {};   // hipFuncSetAttribute is per device
bool g_attr_done_gna[kMaxDev][kNumCfgs] = {};
bool g_attr_done_acc[kMaxDev][kNumCfgs] = {};
bool g_attr_done_hgn[kMaxDev][kNumHalo][2] = {};

}  // namespace

static const CfgInfo& cfg_info(int cfg) {
  if (cfg < kNumCfgs) return kCfgs[cfg];
  return kHaloCfgs[cfg - kNumCfgs];
}
int sdmi_gemm_num_cfgs() { return kNumCfgs + kNumHalo; }
const char* sdmi_gemm_cfg_name(int cfg) {
  if (cfg >= 0 && cfg < sdmi_gemm_num_cfgs()) return cfg_info(cfg).name;
  return "?";
}

int sdmi_gemm_num_plain_cfgs(void) { return kNumCfgs; }
void sdmi_gemm_cfg_dims(int cfg, int* bm, int* bn) {
  const CfgInfo& c = cfg_info(cfg);
  *bm = c.BM;
  *bn = c.BN;
}

// halo-reuse kernel applicability: 3x3 stride-1 pad-1, tile = whole image rows inside one image
static bool halo_ok(const GemmArgs& a, const CfgInfo& c) {
  if (a.ks != 3 || a.stride != 1 || a.pad != 1 || a.X0 != 0 || a.rowstat || a.ln_stat) return false;
  if ((a.Hs << a.ups) != a.Ho || (a.Ws << a.ups) != a.Wo) return false;
  if (a.Wo % 8 != 0 || c.BM % a.Wo != 0 || (a.Ho * a.Wo) % c.BM != 0 || a.M % c.BM != 0) return false;
  const int TH = c.BM / a.Wo;
  if ((TH + 2) * (a.Wo / 8) > c.ntaph * c.nw) return false;      // one halo piece per producer wave per tap
  return true;
}

// dynamic LDS of a halo launch; gn: with the raw-piece staging ring and the {a, b} table of the GN variant
static int halo_lds_bytes(const GemmArgs& a, const CfgInfo& c, bool gn, int* halo_bytes_out) {
  const int TH = c.BM / a.Wo;
  const int halo_bytes = (((TH + 2) * (a.Wo + 2) * 128) + 1023) / 1024 * 1024;
  int lds = 2 * halo_bytes + c.NS * c.BN * 128 + 1024;
  if (gn) lds += c.nw * (a.hgn.depth + 1) * 2048 + (2 * (a.C0 + a.C1) + 64) * 4;   // depth + 1 raw-piece staging slots per normaliser wave + the {a, b} table
  lds = (lds + 15) / 16 * 16;
  if (lds < c.BM * c.BN * 4) lds = c.BM * c.BN * 4;
  if (halo_bytes_out) *halo_bytes_out = halo_bytes;
  return lds;
}

bool sdmi_gemm_cfg_applicable(const GemmArgs& a, int cfg) {
  if (cfg < 0 || cfg >= sdmi_gemm_num_cfgs()) return false;
  const CfgInfo& c = cfg_info(cfg);
  if (c.BN % 64 != 0 && (a.ks != 3 || a.rowstat || a.ln_stat || a.outT)) return false;   // 160-wide tiles: 3x3 convs only
  if (a.outT && (a.nt0 % c.BN) != 0) return false;
  if (a.act == 2 && (c.BN != 128 || cfg >= kNumCfgs)) return false;                      // softmax epilogue: one head per n-tile
  if (a.img_rows && (a.img_rows % c.BM != 0 || cfg >= kNumCfgs)) return false;
  if (cfg >= kNumCfgs) {
    if (!halo_ok(a, c)) return false;
    if (halo_lds_bytes(a, c, false, nullptr) > 160 * 1024) return false;
  }
  return true;
}

// GroupNorm on the A fragments (GemmArgs::gna_rec): a plain 1x1 GEMM over one source, K = C0 <= kGnaMaxC, whole tiles inside one
// image, a config that was built with the variant
bool sdmi_gemm_gna_ok(const GemmArgs& a, int cfg) {
  if (cfg < 0 || cfg >= kNumCfgs) return false;
  const CfgInfo& c = kCfgs[cfg];
  if (!c.kern_gna) return false;
  if (a.ks != 1 || a.stride != 1 || a.ups != 0 || a.C1 != 0 || a.X0 != 0 || a.X1 != 0 || a.K != a.C0 || a.phase2 || a.img_rows || a.ln_stat) return false;
  if (a.C0 > kGnaMaxC || a.C0 % 320 != 0 || a.gna_atom <= 0 || (a.C0 / 32) % a.gna_atom != 0) return false;
  if (a.gna_rows <= 0 || a.gna_rows % c.BM != 0 || a.M % a.gna_rows != 0 || a.gna_T <= 0 || (a.gna_parts != 1 && a.gna_parts != 2)) return false;
  return true;
}

// Every GroupNorm workgroup adds up all T x natoms x parts records of its image (as it did the chunk partials of gn_stats): 32 KiB at most
constexpr int kGaccMaxRec = 4096;          // T * natoms * parts: 16 loads per thread of the GroupNorm that adds them up (norm.hip gn_apply_kernel)
static bool gacc_atom_ok(int atom) { return atom == 4 || (atom >= 8 && atom <= 16 && atom % 2 == 0); }   // an 8-column chunk spans <= 2 atoms, an atom <= 3 chunks, pairs never straddle

// one-pass epilogue (store_tile) statistics: whole tiles inside one image, a tile shape whose threads keep their columns
bool sdmi_gemm_gacc_ok(const GemmArgs& a, int cfg) {
  if (cfg < 0 || cfg >= sdmi_gemm_num_cfgs()) return false;
  const CfgInfo& c = cfg_info(cfg);
  if (a.outT || !gacc_atom_ok(a.gacc.atom) || a.gacc.natoms * a.gacc.atom != a.N || a.gacc.rows_img <= 0) return false;
  if (c.BN == 160) {                                        // (store_tile GACC160: whole atoms per tile, column sums from the tile)
    if (160 % a.gacc.atom != 0 || a.N % 160 != 0 || a.M % c.BM != 0) return false;
  } else if (c.BN % 64 != 0 || c.NT % (c.BN / 8) != 0 || c.BN / a.gacc.atom + 2 > c.NT) return false;   // one lane per atom of a tile
  if (a.gacc.rows_img % c.BM != 0 || a.gacc.mod % a.gacc.rows_img != 0 || a.M % a.gacc.mod != 0) return false;
  return (long)sdmi_gemm_gacc_T(a, cfg) * a.gacc.natoms * 2 <= kGaccMaxRec;
}
int sdmi_gemm_gacc_T(const GemmArgs& a, int cfg) {
  const CfgInfo& c = cfg_info(cfg);
  return (a.M / a.gacc.mod) * (a.gacc.rows_img / c.BM);
}

bool sdmi_gemm_hgn_ok(const GemmArgs& a, int cfg) {
  if (cfg < kNumCfgs || cfg >= kNumCfgs + kNumHalo) return false;
  const CfgInfo& c = kHaloCfgs[cfg - kNumCfgs];
  const HaloGn& g = a.hgn;
  if (!c.hkern_gn || !halo_ok(a, c) || a.accurate || a.ups != 0 || c.NT + 64 * c.nw > 1024) return false;
  const int Cin = a.C0 + a.C1;
  if (!g.x0 || !g.gamma || !g.beta || !g.rec0 || g.C0 <= 0 || g.C0 % 64 != 0 || g.C1 % 64 != 0 || g.C0 + g.C1 != Cin || (g.C1 > 0 && (!g.x1 || !g.rec1))) return false;
  if (Cin > 2560 || Cin % 32 != 0 || g.atom <= 0 || (Cin / 32) % g.atom != 0 || g.C0 % g.atom != 0) return false;
  if (g.T0 <= 0 || (g.P0 != 1 && g.P0 != 2) || (g.C1 > 0 && (g.T1 <= 0 || (g.P1 != 1 && g.P1 != 2)))) return false;
  return sdmi_gemm_hgn_depth(a, cfg) >= 2;
}

// request-ahead distance of the raw pieces: the deepest of 4 .. 1 intervals whose staging slots still fit the LDS (0: none does).
// Below 2 the pieces arrive too late (measured: one interval costs the conv more than the GroupNorm launch it saves).
int sdmi_gemm_hgn_depth(const GemmArgs& a, int cfg) {
  if (cfg < kNumCfgs || cfg >= kNumCfgs + kNumHalo) return 0;
  static const int max_depth = getenv("SDMI_HALO_GN_DEPTH") ? atoi(getenv("SDMI_HALO_GN_DEPTH")) : 4;
  GemmArgs t = a;
  for (int d = max_depth < 4 ? max_depth : 4; d >= 1; --d) {
    t.hgn.depth = d;
    if (halo_lds_bytes(t, kHaloCfgs[cfg - kNumCfgs], true, nullptr) <= 160 * 1024) return d;
  }
  return 0;
}

bool sdmi_gemm_acc_ok(int cfg) { return cfg >= 0 && cfg < kNumCfgs && kCfgs[cfg].kern_acc != nullptr; }

// Tile and split-K factor of an accurate-mode launch.  No tuner here (the mode is for validation, its plans must not depend on
// timings): the largest of the three wide-operand tiles that still gives the chip a round of workgroups, K split to fill the
// rest where the epilogue allows it.
int sdmi_gemm_pick_acc_cfg(const GemmArgs& a, int* ksplit) {
  int best = -1;
  auto find = [&](int bm, int bn) { for (int c = 0; c < kNumCfgs; ++c) if (kCfgs[c].kern_acc && kCfgs[c].BM == bm && kCfgs[c].BN == bn) return c; return -1; };
  const int big = find(128, 128), wide = find(64, 128), small = find(64, 64);
  auto tiles_of = [&](int c) { return ((a.M + kCfgs[c].BM - 1) / kCfgs[c].BM) * ((a.N + kCfgs[c].BN - 1) / kCfgs[c].BN); };
  auto fits = [&](int c) {
    if (c < 0) return false;
    if (a.act == 2 && kCfgs[c].BN != 128) return false;
    if (a.img_rows && a.img_rows % kCfgs[c].BM != 0) return false;
    if (a.outT && a.nt0 % kCfgs[c].BN != 0) return false;
    return true;
  };
  if (fits(big) && tiles_of(big) >= 192) best = big;
  else if (fits(wide) && a.N % 128 == 0 && tiles_of(wide) >= 128) best = wide;
  else if (fits(small)) best = small;
  else if (fits(wide)) best = wide;
  else best = big;
  int ks = 1;
  const int nkt = a.K / 64;
  const bool can_split = !a.outT && !a.act && !a.rowstat && !a.ln_stat && !(a.img_rows && !a.phase2) && a.slab != nullptr;
  if (can_split && best >= 0) {
    const int t = tiles_of(best);
    while (ks < 16 && t * ks * 2 <= 320 && nkt / (ks * 2) >= 4) ks *= 2;
  }
  if (ksplit) *ksplit = ks;
  return best;
}

static int pick_cfg(const GemmArgs& a) {
  // heuristic default (the UNet plan autotunes over all cfgs x split-K instead)
  if (a.act == 2) return a.img_rows % 128 == 0 ? 1 : 6;            // softmax epilogue: 128-wide tiles (t128x128s3 / t64x128s4)
  if (a.img_rows % 128 != 0) return 9;                             // per-image weights on a 64-pixel map
  if (a.M <= 64) return 9;                                         // t64x64s4
  if (a.N % 128 != 0 && a.N % 64 == 0 && a.N < 512) return 4;      // t128x64s4
  return 1;                                                        // t128x128s3
}

int sdmi_gemm_pick_cfg(const GemmArgs& a) { return pick_cfg(a); }

size_t sdmi_gemm_slab_bytes(const GemmArgs& a, int /*cfg*/, int ksplit) {
  return ksplit > 1 ? (size_t)ksplit * a.M * a.N * sizeof(float) : 0;
}

// the split-K factor a launch of `a` with tile config `cfg` really runs with: the requested one clamped to the K-steps there are,
// slices rounded up to whole channel chunks (9 taps) for the halo kernels, no empty slices
int sdmi_gemm_effective_ksplit(const GemmArgs& a, int cfg, int* ksteps_per_out) {
  const int nkt = a.K / 64;
  int ks = a.ksplit < 1 ? 1 : a.ksplit;
  if (ks > nkt) ks = nkt;
  if (ks < 1) ks = 1;
  int per = (nkt + ks - 1) / ks;
  if (cfg >= kNumCfgs) per = (per + 8) / 9 * 9;
  if (per < 1) per = 1;
  if (ksteps_per_out) *ksteps_per_out = per;
  return (nkt + per - 1) / per;
}

int sdmi_launch_gemm(const GemmArgs& a, int cfg, hipStream_t st, int* ksplit_out, int* ksteps_per_out) {
  SDMI_REQUIRE(a.K % 64 == 0 && a.K > 0, "gemm: K=%d must be a positive multiple of 64", a.K);
  SDMI_REQUIRE(a.N % 8 == 0 && a.N > 0, "gemm: N=%d must be a positive multiple of 8", a.N);
  SDMI_REQUIRE(a.M > 0, "gemm: M=%d", a.M);
  SDMI_REQUIRE(a.C0 % 64 == 0 && a.C1 % 64 == 0 && a.C0 > 0, "gemm: C0=%d C1=%d must be multiples of 64", a.C0, a.C1);
  SDMI_REQUIRE(a.K == a.ks * a.ks * (a.C0 + a.C1) + a.X0 + a.X1, "gemm: K=%d != ks^2*(C0+C1) + X0+X1", a.K);
  SDMI_REQUIRE(a.cs_hi % 8 == 0 && (a.cs_hi == 0 || !a.outT || a.cs_hi <= a.nt0), "gemm: scaled column range must be a multiple of 8 outside the transposed tail");
  SDMI_REQUIRE(!a.rowstat || (a.ksplit <= 1 && !a.outT), "gemm: row statistics need ksplit == 1 and no transposed tail");
  SDMI_REQUIRE(!a.ln_stat || (a.ln_g && a.ln_ntn > 0 && a.ln_C > 0 && (!a.res || a.ln_ksteps > 0)), "gemm: bad LayerNorm-fold arguments");
  SDMI_REQUIRE(!a.ln_stat || a.ksplit <= 1 ||
                   (a.ln_ksteps > 0 && a.ln_out && (a.K / 64) % a.ksplit == 0 && a.ln_ksteps % ((a.K / 64) / a.ksplit) == 0 && a.ks == 1 && !a.img_rows),
               "gemm: split-K with the LayerNorm fold needs the partial fold, ln_out, and K-slices that end on the fold boundary");
  SDMI_REQUIRE(a.ln_ksteps >= 0 && (a.ln_ksteps == 0 || (a.ln_stat && a.ks == 1 && !a.outT && a.ln_ksteps * 64 < a.K && a.ln_ksteps * 64 == a.ln_C)),
               "gemm: partial LayerNorm fold needs ln_stat, a 1x1 GEMM and ln_ksteps*64 == ln_C < K");
  SDMI_REQUIRE(a.X0 % 64 == 0 && a.X1 % 64 == 0 && (a.X0 == 0 || (a.x0 && a.ups == 0 && a.stride == 1)), "gemm: bad extra segment");
  SDMI_REQUIRE(a.ks == 1 || a.ks == 3 || (a.ks == 2 && a.phase2), "gemm: ks=%d", a.ks);
  SDMI_REQUIRE(!a.phase2 || (a.ks == 2 && a.M % 4 == 0 && a.img_rows == a.M / 4 && a.M == 4 * (a.M / 4 / (a.Hs * a.Ws)) * a.Hs * a.Ws && a.ups == 0 &&
                             a.stride == 1 && a.X0 == 0 && !a.outT && !a.rowstat && !a.ln_stat),
               "gemm: a phase-decomposed x2-upsample conv needs ks = 2, M = 4*B*Hs*Ws, img_rows = M/4 and a plain epilogue");
  SDMI_REQUIRE(a.zero && a.a0 && a.w && a.out, "gemm: null pointer");
  SDMI_REQUIRE(a.ldc % 8 == 0 && (!a.res || a.ldr % 8 == 0), "gemm: ldc/ldr must be multiples of 8");
  if (cfg < 0) cfg = pick_cfg(a);
  SDMI_REQUIRE(cfg < sdmi_gemm_num_cfgs(), "gemm: bad cfg %d", cfg);
  const bool halo = cfg >= kNumCfgs;
  const CfgInfo& c = cfg_info(cfg);
  if (halo) SDMI_REQUIRE(halo_ok(a, c) && !a.accurate, "gemm: halo config %s not applicable to this conv", c.name);
  SDMI_REQUIRE(c.BN % 64 == 0 || (a.ks == 3 && !a.rowstat && !a.ln_stat && !a.outT), "gemm: config %s (160-wide tile) is not applicable to this GEMM: 3x3 convs only", c.name);
  SDMI_REQUIRE(a.act != 2 || (c.BN == 128 && !halo && a.ksplit <= 1 && !a.outT && !a.out_f32 && !a.res && a.N % 128 == 0 && a.M % c.BM == 0 &&
                              a.sm_valid > 0 && a.sm_valid <= 128),
               "gemm: the softmax epilogue needs a BN=128 plain tile, full tiles (M %% BM == 0, N %% 128 == 0), fp16 output, no split-K / residual");
  SDMI_REQUIRE(a.img_rows == 0 || (a.img_rows % c.BM == 0 && a.M % a.img_rows == 0 && !halo && (a.ks == 1 || a.phase2) && (a.ksplit <= 1 || a.phase2)),
               "gemm: per-image weights need BM | img_rows | M, a 1x1 GEMM and no split-K");
  if (a.outT) {
    SDMI_REQUIRE(a.nt0 % c.BN == 0, "gemm: transposed tail start %d not a multiple of BN=%d", a.nt0, c.BN);
    SDMI_REQUIRE(a.S > 0 && a.ldt % (a.tperm ? 16 : 8) == 0, "gemm: transposed tail needs S > 0 and ldt a multiple of 8 (16 with the quad-permuted key order)");
  }
  const int nkt = a.K / 64;
  GemmArgs p = a;
  gnrec_magic(p.gacc);
  if (p.gacc.rec && p.ksplit <= 1)
    SDMI_REQUIRE(sdmi_gemm_gacc_ok(a, cfg) && a.gacc.parts == 2 && a.gacc.T == sdmi_gemm_gacc_T(a, cfg),
                 "gemm: config %s cannot accumulate GroupNorm statistics for this shape (rows per image %d, T %d, parts %d)", c.name, a.gacc.rows_img, a.gacc.T, a.gacc.parts);
  if (p.lda0 <= 0) p.lda0 = p.C0;
  if (p.lda1 <= 0) p.lda1 = p.C1;
  if (p.ldw <= 0) p.ldw = p.K;
  if (p.ldx0 <= 0) p.ldx0 = p.X0;
  if (p.ldx1 <= 0) p.ldx1 = p.X1;
  SDMI_REQUIRE(p.lda0 % 8 == 0 && p.lda1 % 8 == 0 && p.ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8");
  {
    static const int force = getenv("SDMI_TILE_ORDER") ? atoi(getenv("SDMI_TILE_ORDER")) : -1;   // A/B knob: 0 m-major, 1 n-major
    const double w_bytes = 2.0 * a.N * a.K;
    const double a_bytes = 2.0 * ((double)a.M * a.stride * a.stride / (a.ups ? 4 : 1)) * (a.C0 + a.C1) + 2.0 * a.M * (a.X0 + a.X1);
    p.n_major = force >= 0 ? force : (w_bytes > a_bytes);
  }
  p.ksplit = sdmi_gemm_effective_ksplit(a, cfg, &p.ksteps_per);
  if (p.ksplit > 1) SDMI_REQUIRE(p.slab != nullptr, "gemm: split-K needs a slab");
  // The statistics layout (T, parts) was validated above against the REQUESTED split-K factor.  When the clamps lower it to 1
  // the one-pass epilogue would write its parts = 2 records into a table laid out (and sized) for the combine's parts = 1:
  // take no statistics instead -- the caller sees the effective factor in *ksplit_out and treats the records as absent
  // (Engine::gemm: gok = false; sdmi_op_gemm lays its records out for the effective factor up front: unet.hip gacc_layout).
  if (p.gacc.rec && a.ksplit > 1 && p.ksplit <= 1) p.gacc = GnRec{};
  const int tiles_m = (a.M + c.BM - 1) / c.BM, tiles_n = (a.N + c.BN - 1) / c.BN;
  const int tiles = tiles_m * tiles_n;
  SDMI_REQUIRE((long long)tiles * p.ksplit < (1ll << 22) && (long long)tiles * tiles * p.ksplit < (1ll << 36),
               "gemm: %d tiles x %d K-slices exceed the tile map's range", tiles, p.ksplit);
  auto magic = [](int d) -> unsigned long long { return (1ull << 36) / (unsigned long long)d + 1ull; };
  p.tiles = tiles;
  p.tiles_magic = magic(tiles);
  p.tdiv = p.n_major ? tiles_m : tiles_n;
  p.tdiv_magic = magic(p.tdiv);
  p.plain = !halo && p.ks == 1 && p.stride == 1 && p.ups == 0 && p.C1 == 0 && p.X0 == 0 && p.X1 == 0 && p.phase2 == 0;
  int dev = 0;
  SDMI_CHECK_HIP(hipGetDevice(&dev));
  SDMI_REQUIRE(dev >= 0 && dev < kMaxDev, "gemm: device index %d out of range", dev);
  // > 64 KiB of dynamic LDS needs the function attribute once per (kernel, device)
  auto set_attr = [&](const void* fn, int bytes) -> int {
    if (!g_attr_done[dev][cfg]) {
      SDMI_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      g_attr_done[dev][cfg] = true;
    }
    return SDMI_OK;
  };
  SDMI_REQUIRE(!a.hgn.x0 || halo, "gemm: GroupNorm inside the conv (hgn) needs a halo-reuse config");
  if (halo) {
    int halo_bytes = 0;
    const bool gn = a.hgn.x0 != nullptr;
    if (gn) {
      SDMI_REQUIRE(sdmi_gemm_hgn_ok(a, cfg), "gemm: halo config %s cannot apply GroupNorm to its A operand for this conv", c.name);
      p.hgn.depth = sdmi_gemm_hgn_depth(a, cfg);
    }
    const int lds = halo_lds_bytes(p, c, gn, &halo_bytes);
    SDMI_REQUIRE(lds <= 160 * 1024, "gemm: halo config %s needs %d B of LDS", c.name, lds);
    if (gn) {
      const int f32 = a.hgn.in_f32 ? 1 : 0;
      void (*kern)(GemmArgs, int) = f32 ? c.hkern_gn : c.hkern_gn16;
      if (!g_attr_done_hgn[dev][cfg - kNumCfgs][f32]) {
        SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        g_attr_done_hgn[dev][cfg - kNumCfgs][f32] = true;
      }
      hipLaunchKernelGGL(kern, dim3(tiles * p.ksplit), dim3(c.NT + 64 * c.nw), lds, st, p, halo_bytes);      // + the normaliser waves
    } else {
      if (set_attr((const void*)c.hkern, 160 * 1024) != SDMI_OK) return SDMI_EHIP;
      hipLaunchKernelGGL(c.hkern, dim3(tiles * p.ksplit), dim3(c.NT), lds, st, p, halo_bytes);
    }
    SDMI_CHECK_HIP(hipGetLastError());
  } else if (a.accurate) {
    SDMI_REQUIRE(c.kern_acc != nullptr, "gemm: config %s was not built with the wide A operand (accurate mode)", c.name);
    SDMI_REQUIRE(a.a0f && (a.C1 == 0 || a.a1f) && (a.X0 == 0 || a.x0f) && (a.X1 == 0 || a.x1f) && !a.gna_rec,
                 "gemm: the accurate mode needs the fp32 copy of every A source");
    if (!g_attr_done_acc[dev][cfg]) {
      SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)c.kern_acc, hipFuncAttributeMaxDynamicSharedMemorySize, c.LDS_acc));
      g_attr_done_acc[dev][cfg] = true;
    }
    hipLaunchKernelGGL(c.kern_acc, dim3(tiles * p.ksplit), dim3(c.NT), c.LDS_acc, st, p);
    SDMI_CHECK_HIP(hipGetLastError());
  } else {
    if (a.gna_rec) {
      SDMI_REQUIRE(sdmi_gemm_gna_ok(a, cfg) && a.gna_gamma && a.gna_beta && p.plain, "gemm: config %s cannot apply GroupNorm to its A fragments for this launch", c.name);
      if (!g_attr_done_gna[dev][cfg]) {
        SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)c.kern_gna, hipFuncAttributeMaxDynamicSharedMemorySize, c.LDS));
        g_attr_done_gna[dev][cfg] = true;
      }
      hipLaunchKernelGGL(c.kern_gna, dim3(tiles * p.ksplit), dim3(c.NT), c.LDS, st, p);
      SDMI_CHECK_HIP(hipGetLastError());
    } else {
    if (set_attr((const void*)c.kern, c.LDS) != SDMI_OK) return SDMI_EHIP;
    hipLaunchKernelGGL(c.kern, dim3(tiles * p.ksplit), dim3(c.NT), c.LDS, st, p);
    SDMI_CHECK_HIP(hipGetLastError());
    }
  }
  if (ksplit_out) *ksplit_out = p.ksplit;
  if (ksteps_per_out) *ksteps_per_out = p.ksteps_per;
  if (p.ksplit > 1 && !p.no_finalize) return sdmi_launch_splitk_finalize(p, st);
  return SDMI_OK;
}

// rows per workgroup of the statistics-taking combine (0: this shape cannot take it): a multiple of the 16 rows per pass that
// divides the rows of an image, few enough record rows, aiming at >= 256 workgroups
static int finalize_gacc_rows(const GemmArgs& a) {
  if (!a.gacc.rec || a.outT || a.N % kFinCols != 0 || !gacc_atom_ok(a.gacc.atom) || kFinCols % a.gacc.atom != 0 ||
      a.gacc.natoms * a.gacc.atom != a.N || a.gacc.rows_img <= 0 || a.gacc.mod % a.gacc.rows_img != 0 || a.M % a.gacc.mod != 0)
    return 0;
  const int nsl = a.N / kFinCols, phases = a.M / a.gacc.mod;
  int rows = kFinNT / (kFinCols / 8);
  if (a.gacc.rows_img % rows != 0) return 0;
  auto recs = [&](int r) { return (long)phases * (a.gacc.rows_img / r) * a.gacc.natoms; };
  while (recs(rows) > kGaccMaxRec) {
    if (a.gacc.rows_img % (rows * 2) != 0) return 0;
    rows *= 2;
  }
  while (rows < 128 && a.gacc.rows_img % (rows * 2) == 0 && (long)(a.M / (rows * 2)) * nsl >= 256) rows *= 2;
  return rows <= 128 ? rows : 0;          // (8 items per thread at most: kernel instantiations below)
}
bool sdmi_finalize_gacc_ok(const GemmArgs& a, int* T_out) {
  const int rows = finalize_gacc_rows(a);
  if (rows > 0 && T_out) *T_out = (a.M / a.gacc.mod) * (a.gacc.rows_img / rows);
  return rows > 0;
}

int sdmi_launch_splitk_finalize(const GemmArgs& a, hipStream_t st) {
  SDMI_REQUIRE((size_t)a.M * (a.N / 8) < ((size_t)1 << 31), "splitk_finalize: M*N too large");
  if (a.gacc.rec) {
    GemmArgs ag = a;
    gnrec_magic(ag.gacc);
    const GemmArgs& a = ag;
    const int rows = finalize_gacc_rows(a);
    SDMI_REQUIRE(rows > 0, "splitk_finalize: GroupNorm statistics were asked of a shape the combine cannot take (N=%d)", a.N);
    SDMI_REQUIRE(a.gacc.parts == 1 && a.gacc.T == (a.M / a.gacc.mod) * (a.gacc.rows_img / rows), "splitk_finalize: statistics record rows T=%d parts=%d do not match the launch", a.gacc.T, a.gacc.parts);
    const int nsl = a.N / kFinCols;
    const dim3 grid((a.M / rows) * nsl), block(kFinNT);
    switch (rows / (kFinNT / (kFinCols / 8))) {
      case 1: hipLaunchKernelGGL(splitk_finalize_gacc_kernel<1>, grid, block, 0, st, a, nsl); break;
      case 2: hipLaunchKernelGGL(splitk_finalize_gacc_kernel<2>, grid, block, 0, st, a, nsl); break;
      case 4: hipLaunchKernelGGL(splitk_finalize_gacc_kernel<4>, grid, block, 0, st, a, nsl); break;
      default: hipLaunchKernelGGL(splitk_finalize_gacc_kernel<8>, grid, block, 0, st, a, nsl); break;
    }
    SDMI_CHECK_HIP(hipGetLastError());
    return SDMI_OK;
  }
  const size_t total8 = (size_t)a.M * (a.N / 8);
  int blocks = (int)((total8 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, st, a);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}
