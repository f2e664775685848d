// Common device/host declarations for the sdmi (Stable Diffusion on MI355X) native library.
// gfx950 / CDNA4 only: 64-wide wavefronts, MFMA 32x32x16 f16, LDS-DMA (global_load_lds).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef __HIPCC__
// Kernel arguments are read with s_load next to their first use, each behind its own s_waitcnt: with a 100-360 byte argument
// struct that is a dozen SERIAL scalar-cache misses (~300 cycles each: the launch's kernarg slot is new memory) before the
// first tile is requested -- 1.6 us of the 3.8 k-cycle set-up measured on the K = C GEMMs (tools/phase_probe.py).  Touching
// every 64-byte line of the struct at the top of the kernel turns them into ONE miss latency; the later loads hit.
template <int BYTES>
__device__ __forceinline__ void sdmi_kernarg_warm() {
#ifdef SDMI_NO_KERNARG_WARM      // A/B builds
  return;
#endif
  typedef const __attribute__((address_space(4))) int* kptr_t;
  kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int L = (BYTES + 63) / 64;
  int t[L + 1];
#pragma unroll
  for (int i = 0; i < L; ++i) t[i] = ka[i * 16];
  t[L] = ka[(BYTES - 4) / 4];            // the segment is only guaranteed 16-byte aligned: the last argument may sit in one more line
                                         // (never read past BYTES: a kernel without hidden arguments ends there)
#pragma unroll
  for (int i = 0; i <= L; ++i) asm volatile("" ::"s"(t[i]));
}
#endif

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics taken from the PRODUCER's epilogue (sd/diffusion.py:173,199,294,733: every GroupNorm input is the
// output of a conv / linear of this library).  The kernel that writes a tensor also leaves, per image, per block of rows it
// owns ("record row" t) and per ATOM of `atom` consecutive channels (10 in the UNet: every group of every GroupNorm that reads
// the tensor -- alone or as one half of a skip concat, 10 / 20 / 30 / 40 / 60 / 80 channels per group -- is a whole number of
// atoms), the moments {sum x, sum x^2} of what it stored; the GroupNorm is then ONE normalising pass whose workgroups add up the
// records of their own groups in a fixed order (fp64), and the statistics launch (a full extra read of the tensor) is gone.
// Plain stores, no atomics: every (record row, atom, part) slot is written by exactly one workgroup, so nothing has to be
// zeroed and the result does not depend on the order of the workgroups.  (A first version accumulated 64-bit fixed-point
// integers with atomics: 4 x atoms x workgroups of them on a few KB cost 2 - 10 us per producer launch: 235 vs 252 steps/s.)
// Layout: rec[((image * T + t) * natoms + atom) * parts + part] = {sum, sum of squares} (float2).  parts = 2 for tiled
// producers: an atom that straddles two column tiles gets its first channels' moments from the left tile (part 0) and the
// rest from the right one (part 1); a tile that holds the whole atom writes {moments, 0}.
struct GnRec {
  float* rec;          // nullptr: off
  int atom;            // channels per atom; the tensor has natoms = C / atom of them
  int natoms;
  int rows_img;        // rows (pixels) of one image inside a block of `mod` rows; a producer's row block never straddles images
  int mod;             // M, or M / 4 for the phase-decomposed upsample conv (rows are phase-major there)
  int T;               // record rows per image
  int parts;           // 1 or 2
  // filled by the launchers (gnrec_magic): exact fixed-point reciprocals, so that no integer division by a run-time value is
  // left in the epilogues (each is ~40 instructions of one wave; a dozen of them were most of the 1.2 us a first version of the
  // statistics added to every producer launch):  x / atom == (x * atom_magic) >> 20 for x < 40000,
  // m / rows_img == (m * rows_magic) >> 36 for m * rows_img < 2^36
  unsigned atom_magic;
  unsigned long long rows_magic;
};
inline void gnrec_magic(GnRec& g) {
  if (!g.rec || g.atom <= 0 || g.rows_img <= 0) return;
  g.atom_magic = (1u << 20) / (unsigned)g.atom + 1u;
  g.rows_magic = (1ull << 36) / (unsigned long long)g.rows_img + 1ull;
}
#ifdef __HIPCC__
__device__ __forceinline__ int gnrec_div_atom(const GnRec& g, int x) { return (int)(((unsigned)x * g.atom_magic) >> 20); }
__device__ __forceinline__ int gnrec_div_rows(const GnRec& g, int m) { return (int)(((unsigned long long)m * g.rows_magic) >> 36); }
// sum over the lanes l ^ O of a wave, O in {8, 16, 32}, on the VALU (DPP row rotate / v_permlane16_swap / v_permlane32_swap)
// instead of ds_bpermute round trips: swapping a register with a copy of itself leaves {even rows, even rows} in one result
// and {odd, odd} in the other (rows of 16 lanes; halves of 32 for permlane32), so their sum is the xor-partner sum in every lane
template <int O>
__device__ __forceinline__ float wave_xor_sum(float x) {
  static_assert(O == 8 || O == 16 || O == 32, "lane distance");
  const unsigned u = __float_as_uint(x);
  if constexpr (O == 8) {
    return x + __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x128 /* row_ror:8 */, 0xf, 0xf, false));
  } else if constexpr (O == 16) {
    const auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  } else {
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
}
// Per-thread side: a thread stores 8 consecutive columns n .. n+7 of several rows and keeps their moments per COLUMN (one add
// and one fma per element; a first version split them into the two atoms per element -- four selects and four adds more -- and
// cost the large-map GEMMs ~1.5 us each); the columns are dealt to the (at most two) atoms they fall into once, at the end:
// the first `split` columns to atom n / atom, the rest to the next one.
template <bool PERCOL = true>
struct GaccThread {
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  __device__ __forceinline__ void add(const float (&x)[8], int) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] += x[e]; q[e] = fmaf(x[e], x[e], q[e]); }
  }
  // {first atom: sum, sumsq; second atom: sum, sumsq}
  __device__ __forceinline__ f32x4 parts(int split) const {
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool a = e < split;
      r[0] += a ? s[e] : 0.f; r[1] += a ? q[e] : 0.f;
      r[2] += a ? 0.f : s[e]; r[3] += a ? 0.f : q[e];
    }
    return r;
  }
};
// the 1024-thread GEMM kernels have 128 registers per lane and one to four items per thread: four accumulators, the columns
// dealt to the two atoms as they are added
template <>
struct GaccThread<false> {
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
  __device__ __forceinline__ void add(const float (&x)[8], int split) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool a = e < split;
      r[0] += a ? x[e] : 0.f; r[1] += a ? x[e] * x[e] : 0.f;
      r[2] += a ? 0.f : x[e]; r[3] += a ? 0.f : x[e] * x[e];
    }
  }
  __device__ __forceinline__ f32x4 parts(int) const { return r; }
};
#endif

#define SDMI_OK 0
#define SDMI_EINVAL (-22)
#define SDMI_ENOMEM (-12)
#define SDMI_EHIP (-5)
#define SDMI_ENOENT (-2)

// Thread-local error message (sdmi_last_error).
void sdmi_set_error(const char* fmt, ...);

#define SDMI_CHECK_HIP(expr)                                                         \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      sdmi_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return SDMI_EHIP;                                                              \
    }                                                                                \
  } while (0)

#define SDMI_REQUIRE(cond, ...)                   \
  do {                                            \
    if (!(cond)) {                                \
      sdmi_set_error(__VA_ARGS__);                \
      return SDMI_EINVAL;                         \
    }                                             \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM descriptor:  D[m][n] = sum_k A(m,k) * W[n][k]
//   A(m,k) is gathered on the fly from one or two NHWC fp16 tensors (virtual channel concat),
//   optional nearest x2 upsample, ks x ks taps with stride/pad (ks = 1 -> plain GEMM).
//   k is ordered (kh, kw, ci); W is packed [N][K] fp16 with K contiguous.
// ---------------------------------------------------------------------------------------------
// GroupNorm(32) (+SiLU) of the A operand applied INSIDE the halo-reuse 3x3 conv (gemm.hip conv3_halo_kernel<.., GN>): the
// reference's "GroupNorm -> SiLU -> conv3x3" (sd/diffusion.py:173-179,199-205) as ONE launch.  The raw sources x0 | x1 (NHWC, dense,
// C0 | C1 channels, fp32 or fp16) replace a0 / a1; the statistics are the records their producers left (GnRec layout: T record
// rows per image, parts, atoms of `atom` channels, per source); gamma / beta over the C0 + C1 concat channels.
struct HaloGn {
  const void* x0; const void* x1; int in_f32; int C0, C1;
  const float* gamma; const float* beta; float eps; int silu;
  const float* rec0; const float* rec1; int T0, T1, P0, P1, atom;
  int depth;           // set by the launcher: intervals a raw piece is requested ahead of its normalisation (1 .. 4, by the LDS left)
};

struct GemmArgs {
  const f16* a0;
  const f16* a1;       // second concat source or nullptr
  int C0, C1;          // channels per source (C1 = 0 when a1 == nullptr); multiples of 64
  int lda0, lda1;      // pixel (row) stride of each source in elements; 0 = dense (= C0 / C1)
  int ldw;             // row stride of w in elements; 0 = dense (= K)
  // optional EXTRA 1x1 segment appended to K after the ks*ks taps (ResBlock skip conv fused into conv2):
  // sources x0 | x1 (virtual concat) read at the output pixel, K += X0 + X1
  const f16* x0; const f16* x1; int X0, X1, ldx0, ldx1;
  int Hs, Ws;          // stored source spatial dims
  int Ho, Wo;          // output spatial dims; M = B*Ho*Wo
  int ups;             // 1: nearest x2 upsample applied to the source on read
  int stride, pad, ks;
  int M, N, K;         // K = ks*ks*(C0+C1), multiple of 64; N multiple of 8
  const f16* w;        // [N][K]
  const f16* zero;     // >= 256 B of zeros (source for padded taps / out-of-range rows)
  // epilogue:  out = D + bias[n] + res[m][n]
  const float* bias;   // [N] or nullptr
  const void* res;     // [M][ldr] fp16 or fp32, or nullptr
  int res_f32;
  int ldr;
  void* out;           // [M][ldc] fp16 or fp32
  int out_f32;
  int ldc;
  f16* out16;          // optional fp16 shadow copy [M][ldc] (used when out_f32)
  // columns n >= nt0 are written transposed: outT[(b*(N-nt0) + n-nt0)*ldt + s],  m = b*S + s
  f16* outT;
  int nt0;
  int S;
  int ldt;
  int tperm;           // 1: store the key axis of outT in the attention kernel's quad-permuted order (gemm.hip vt_pos)
  // split-K: partial sums to slab[z][M][N] (fp32), combined by splitk_finalize
  float* slab;
  int ksplit;
  int ksteps_per;
  int n_major;         // tile order inside a K-slice, set by the launcher (gemm.hip): 1 = consecutive tiles walk m first
  // launcher-computed tile map (gemm.hip): tile = L % tiles, (tm, tn) from tile / tdiv with exact fixed-point reciprocals
  // magic = floor(2^36 / d) + 1:  (n * magic) >> 36 == n / d  for n * d < 2^36, n < 2^22 (checked by the launcher) -- a scalar
  // division is ~30 instructions of one wave, and there were three in front of the first tile request
  int tiles, tdiv, plain;
  unsigned long long tiles_magic, tdiv_magic;
  int act;             // epilogue after bias: 0 none, 1 quick-GELU x*sigmoid(1.702x) (sd/clip.py:170), 2 row softmax in the
                       //   log2 domain over the tile's 128 columns, columns >= sm_valid masked (one head of the folded
                       //   cross-attention per n-tile: BN = 128 configs only, fp16 output)
  int sm_valid;
  // per-image weights (the folded cross-attention, engine.h xattn_*): rows [i*img_rows, (i+1)*img_rows) use
  // w + i*w_img_stride and bias / ln_g + i*vec_img_stride.  img_rows = 0: one weight matrix.  BM must divide img_rows.
  int img_rows, w_img_stride, vec_img_stride;
  // phase2 = 1: nearest x2 upsample + 3x3 conv (sd/diffusion.py:430-435) as FOUR 2x2 convs on the source grid, one per
  // output parity (py, px): output pixel (2y+py, 2x+px) only sees source rows {y+py-1, y+py} and columns {x+px-1, x+px},
  // with the 3x3 taps that land on the same source pixel pre-summed (misc.hip pack_ups_phase).  Rows are
  // m = phase*(M/4) + (b, y, x) over the SOURCE grid (Hs x Ws = Ho x Wo here), ks = 2, K = 4*(C0+C1), img_rows = M/4
  // selects the phase's weights, and the epilogue scatters row m to output pixel (b, 2y+py, 2x+px) of the 2Hs x 2Ws map.
  // 4/9 of the multiplies of the 9-tap form.
  int phase2;
  // LayerNorm folded around the GEMM (sd/diffusion.py:317,334,351 feeding 321/339/356):
  //  producer side: rowstat != null -> the epilogue also writes per-row {sum, sum of squares} of the fp16 output
  //    over this n-tile to rowstat[(m*tiles_n + tn)*2] (ksplit == 1, no transposed tail);
  //  consumer side: ln_stat != null -> A is the RAW stream, w = gamma (.) W, and the epilogue applies
  //    out = rstd[m]*(acc - mean[m]*ln_g[n]) + bias[n], mean/rstd from ln_stat[m][0..ln_ntn) over ln_C columns.
  // columns n < cs_hi are multiplied by cscale after bias (the attention kernel takes Q pre-multiplied by
  // log2(e)/sqrt(d): folded into the projection in fp32, before the fp16 rounding); cs_hi = 0: off, multiple of 8
  float cscale; int cs_hi;
  float* rowstat;
  const float* ln_stat; int ln_ntn; const float* ln_g; int ln_C; float ln_eps;
  //  ln_ksteps > 0: only the FIRST ln_ksteps K-steps (of 64) are the LayerNorm-folded segment; after them the accumulator is
  //    rescaled in registers, acc = rstd[m]*(acc - mean[m]*ln_g[n]), and the remaining K-steps accumulate a plain product
  //    on top (the composed feed-forward of the attention block: y = LN(s) Wf^T + s Wo^T + b + x in one GEMM); the
  //    epilogue is then the plain bias (+ residual) one.  0: the whole K range is folded (epilogue form above).
  int ln_ksteps;
  // split-K together with the partial fold (ksplit > 1, ln_ksteps > 0): K-slices end on the fold boundary, the slabs hold raw
  // partial sums, workgroups (kz = 0, tn = 0) also write {mean, rstd} of their rows to ln_out[M][2], and splitk_finalize
  // applies  rstd (sum of folded slabs - mean g) + sum of plain slabs + bias (+ res)
  float* ln_out;
  // GroupNorm (no SiLU) applied to the A FRAGMENTS in registers (round 4: the attention block's groupnorm -> conv_input pair,
  // sd/diffusion.py:294-298, without the GroupNorm launch): gna_rec != nullptr -> A is the RAW tensor (fp16), a plain 1x1 GEMM
  // over one source with K = C0; the workgroup sums the producer's statistics records of its image (GnRec layout: gna_T record
  // rows, gna_parts, atoms of gna_atom channels) to mean / rstd per group and normalises every fragment between ds_read and MFMA,
  // y = ((x - mean_hi) - mean_lo) * (rstd gamma) + beta in packed fp16 (the mean as a two-term fp16 sum: x - mean_hi is exact
  // where it matters, so a large |mean| / sigma costs nothing).  gna_rows: rows per image (a multiple of the tile's BM).
  const float* gna_rec; const float* gna_gamma; const float* gna_beta; float gna_eps; int gna_T, gna_parts, gna_atom, gna_rows;
  // Guard of the fold: it multiplies the RAW stream's fp16 shadow, so its error grows like |row mean| / sigma x 2^-12 (measured
  // 1.5e-3 at 10 sigma, 3.8e-3 at 30 against 2e-4 unfused).  ln_guard != nullptr: a workgroup that meets a row with
  // mean^2 > ln_guard_thr2 * variance adds 1 to *ln_guard; the caller reads the counter when the loop is over and repeats it
  // with the fold off (Engine::ln_fold_on) -- no host synchronisation inside the loop.
  int* ln_guard; float ln_guard_thr2;
  int no_finalize;     // split-K: leave the partial sums in the slabs, launch no splitk_finalize
  // GroupNorm statistics of the output from this launch's epilogue (GnRec above): the one-pass path accumulates them in
  // store_tile, a split-K launch in splitk_finalize.  Needs rows_img % BM == 0 (a tile inside one image), no transposed tail.
  GnRec gacc;
  // ACCURATE mode (SDMI_FLAG_ACCURATE): accurate = 1 -> the A operand is read from the fp32 tensors a0f / a1f (and x0f / x1f for the
  // extra 1x1 segment) with the SAME shapes, strides (in elements) and address generator as a0 / a1 / x0 / x1, and enters the
  // product as a hi + lo fp16 pair (two MFMAs per fragment against the same fp16 weights: gemm.hip igemm_kernel<.., ACC>).  Only
  // the tile configs built with the variant take it (sdmi_gemm_acc_ok); no halo kernel, no GroupNorm on the fragments.
  const float* a0f; const float* a1f; const float* x0f; const float* x1f;
  int accurate;
  HaloGn hgn;          // hgn.x0 != nullptr: see HaloGn (halo configs built with the variant only: sdmi_gemm_hgn_ok)
};
bool sdmi_gemm_hgn_ok(const GemmArgs& a, int cfg);         // this halo config can normalise its own A operand for this conv (shape, LDS)
int sdmi_gemm_hgn_depth(const GemmArgs& a, int cfg);       // the request-ahead distance it would run with (HaloGn::depth; 0: no room)
bool sdmi_gemm_acc_ok(int cfg);                              // this tile config was built with the wide-operand variant
int sdmi_gemm_pick_acc_cfg(const GemmArgs& a, int* ksplit);  // the accurate mode's tile and split-K factor for a shape (heuristic; slab_bytes = room for slabs)
// can the one-pass epilogue of tile config `cfg` accumulate the GroupNorm statistics of this GEMM? (gemm.hip)
bool sdmi_gemm_gacc_ok(const GemmArgs& a, int cfg);

// cfg < 0: heuristic.  ksplit_out / ksteps_per_out: effective split-K factor and K-steps per slice of the launch.
// a.no_finalize = 1: a split-K launch only writes its slabs; the caller combines them (sdmi_launch_splitk_finalize with
// those two values filled in, or sdmi_launch_groupnorm with GnArgs::slab)
int sdmi_launch_gemm(const GemmArgs& a, int cfg, hipStream_t st, int* ksplit_out = nullptr, int* ksteps_per_out = nullptr);
int sdmi_gemm_effective_ksplit(const GemmArgs& a, int cfg, int* ksteps_per_out = nullptr);   // what the launcher's clamps make of a.ksplit

// Back-to-back GEMM of the 320-channel attention blocks (b2b.hip): S = A1 W1^T + b1 + R1, then the next Linear with the
// LayerNorm of S folded in, in one launch.  All matrices have 320 columns / output rows.
struct B2bArgs {
  const f16* a1; int lda1;        // [M][lda1], K1 = 320
  const f16* w1; const float* b1; // [320][320], [320]
  const void* r1; int r1_f32;     // residual of S, [M][320] fp32 or fp16 (or nullptr)
  float* s32; f16* s16;           // optional copies of S to memory (nullptr: S never leaves the workgroup)
  const f16* w2; int K2;          // [320][K2]: K2 = 320 (full fold) or 640 = [folded half | plain half] (partial = 1)
  const float* h2;                // bias of the folded w2: b + W2 beta (sdmi_launch_ln_fold_prep); w2's folded part is gamma (.) W2
  int partial;
  float cscale;                   // 0: none
  const void* r2; int r2_f32;     // residual of Y or nullptr
  void* out; int out_f32; f16* out16;
  int M;
  float eps;
  // npass2 = 3: the second product is in_proj (sd/attention.py:42): w2 = [960][320] folded, three passes of 320 output
  // columns -- q (x cscale) and k into out[M][ldo] at columns 0 / 320, v transposed into vt[(b*320 + n)*ldt + pos(s)] in the
  // attention kernel's key order (gemm.hip vt_pos), m = b*S + s.  npass2 = 1: one pass, out[M][ldo].
  int npass2, ldo;
  f16* vt; int S, ldt;
  // gx != nullptr: the first product's A operand is GroupNorm(gx) (32 groups, no SiLU: sd/diffusion.py:294,312), computed
  // in the kernel from the raw stream gx [M][320] (fp32 or fp16) and gn_stats_kernel's partials ([B][gn_nchunk][32][2],
  // images of S rows); a1 is unused.
  const void* gx; int gx_f32;
  const float* gn_partial; int gn_nchunk; const float* gn_gamma; const float* gn_beta; float gn_eps;
  int gn_parts;            // 0 / 1: gn_partial holds {sum, sum of squares} per (chunk, group); 2: two such pairs per entry (GnRec records with parts = 2)
  int* ln_guard; float ln_guard_thr2;     // as GemmArgs::ln_guard, for the rows of S
  GnRec gacc;                     // GroupNorm statistics of `out` (npass2 == 1 only; rows_img = pixels per image, mod = M, T = rows_img / 32, parts = 1)
};
int sdmi_launch_b2b(const B2bArgs& a, hipStream_t st, int bm = 0);   // bm: 32 / 64 rows per workgroup, 0 = by M
int sdmi_b2b_tile_rows(const B2bArgs& a);                            // what bm = 0 resolves to
int sdmi_gemm_num_cfgs();
const char* sdmi_gemm_cfg_name(int cfg);
int sdmi_gemm_num_plain_cfgs(void);   // configs [0, n) are igemm_kernel tiles; the rest are halo-reuse conv kernels
void sdmi_gemm_cfg_dims(int cfg, int* bm, int* bn);
bool sdmi_gemm_cfg_applicable(const GemmArgs& a, int cfg);
size_t sdmi_gemm_slab_bytes(const GemmArgs& a, int cfg, int ksplit);

// ---------------------------------------------------------------------------------------------
struct AttnArgs {
  const f16* q; int ldq;     // [B*Sq][ldq], head h at columns h*d
  const f16* k; int ldk;     // [B*Skv_stored][ldk]
  const f16* vt; int ldvt;   // V transposed: [(b*H*d + h*d + dd)][ldvt] (keys contiguous)
  f16* o; int ldo;           // [B*Sq][ldo]
  float* o32;                // optional fp32 copy of o (same layout; accurate mode: out_proj's wide A operand)
  int B, H, d;               // d in {40, 80, 160} (multiple of 8, <= 160)
  int Sq, Skv;               // Skv = number of valid keys (masking beyond)
  int k_batch_stride;        // rows of k per batch (>= Skv)
  const f16* zero;
  const f16* ones;           // >= 256 B of fp16 1.0 (source of the row-sum row of the V^T tile)
  float scale;               // 1/sqrt(d)
  int prescaled;             // 1: q already carries scale*log2(e) (folded into its projection); 0: the kernel scales Q
  int causal;                // 1: key index > query index is masked (CLIP, sd/attention.py:58-62)
};
int sdmi_launch_attention(const AttnArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// Norms over NHWC activations (stream dtype fp16 or fp32), fp16 normalised output.
struct GnArgs {
  const void* x0; const void* x1;   // concat sources (x1 may be null)
  int in_f32;
  int C0, C1;
  int B, P;                          // P = H*W pixels
  const float* gamma; const float* beta;   // [C]
  float eps;
  int silu;
  f16* y;                            // [B*P][C0+C1]
  float* y32;                        // optional fp32 copy of the output (accurate mode: the next GEMM's wide A operand)
  float* partial;                    // scratch: [B][nchunk][32][2]
  int nchunk;
  // slab != nullptr (single-launch kernel only, C1 = 0): the input is NOT a tensor but the split-K partial sums of the
  // producing GEMM, x = sum_z slab[z][m][c] + sbias[c] (+ sres[m][c]) added in slab order exactly like splitk_finalize --
  // the finalize launch of a split-K conv and the GroupNorm that follows it (sd/diffusion.py:179 -> 199) as ONE kernel.
  // sout / sout16 (optional) receive x itself (fp32 / fp16 [B*P][C]) when another consumer needs the conv's output.
  const float* slab; int ksplit; const float* sbias; const void* sres; int sres_f32; float* sout; f16* sout16;
  // acc0 != nullptr: the statistics come from the producers' epilogues (GnRec: acc0 for x0's C0 channels, acc1 for x1's, with
  // their own record rows T and parts), atoms of `atom` channels; one normalising pass, no statistics launch
  const float* acc0; const float* acc1; int atom; int accT0, accT1, accP0, accP1;
};
int sdmi_gn_nchunk(int P);
int sdmi_gn_launches(const GnArgs& a);
int sdmi_launch_groupnorm(const GnArgs& a, hipStream_t st);
int sdmi_launch_gn_stats(const GnArgs& a, hipStream_t st);   // statistics only (consumer: b2b_kernel via B2bArgs::gn_partial)

struct LnArgs {
  const void* x; int in_f32;  // [M][C]
  int M, C;
  const float* gamma; const float* beta;
  float eps;
  f16* y;
  float* y32;                 // optional fp32 copy of the output (CLIP's final LayerNorm)
};
int sdmi_launch_layernorm(const LnArgs& a, hipStream_t st);

// misc kernels (misc.hip)
int sdmi_launch_cast_f32_f16(const float* x, f16* y, size_t n, hipStream_t st);
int sdmi_launch_pack_conv(const void* w, int w_f32, f16* out, int O, int I, int ks, int o_keep, hipStream_t st);
int sdmi_launch_pack_stem(const void* w, int w_f32, float* w36, int Cout, int Cin, hipStream_t st);
int sdmi_launch_cast_any_f32(const void* x, int in_f32, float* y, size_t n, hipStream_t st);
// y[m][n] = sum_k act(x[m][k]) * W[n][k] + b[n]   (fp32 x/y, fp16 W), act = SiLU if silu
int sdmi_launch_small_linear(const float* x, const f16* w, const float* b, float* y, int M, int N, int K,
                             int silu, int ldy, hipStream_t st);
// stem conv 4->Cout from NCHW fp32 latents (image b reads latent b mod lat_batch); gn_rec: GroupNorm statistics records of the
// output (GnRec layout, atom 10, parts 1, gn_rec_T = H*W/64 record rows per image), or nullptr
int sdmi_launch_stem_conv(const float* lat, int lat_batch, const float* w36, const float* bias,
                          void* out, int out_f32, f16* out16, int B, int H, int W, int Cout, int Cin, hipStream_t st,
                          float* gn_rec = nullptr, int gn_rec_T = 0);
// final conv Cin->4 from NHWC fp16 (already GN+SiLU) to NCHW fp32
int sdmi_launch_final_conv(const f16* x, const f16* w, const float* bias, float* out, int B, int H, int W,
                           int Cin, int Cout, hipStream_t st, const float* x32 = nullptr);   // x32: read the input in fp32 (accurate mode)
// the same conv for prompt i's conditional (b = i) and unconditional (b = P + i) image + CFG combine + DDPM update of the latents
// (P,4,H,W) in one launch (the eps tensor is never written); coef as sdmi_launch_cfg_ddpm
int sdmi_launch_final_conv_step(const f16* x, const f16* w, const float* bias, int P, int H, int W, int Cin, int do_cfg, float cfg_scale,
                                float* latents, const float* noise, const float* coef, hipStream_t st);
int sdmi_launch_row_softmax(const f16* s, f16* p, int rows, int L, float scale, hipStream_t st);
int sdmi_launch_q4_reinterpret_add(const float* o, const void* x, int x_f32, void* y, int y_f32, f16* y16, int B, int P,
                                   int C, hipStream_t st);
int sdmi_launch_clip_embed(const int64_t* tokens, const float* tok_emb, const float* pos_emb, float* out, f16* out16,
                           int rows, int T, int C, int vocab, hipStream_t st);
int sdmi_launch_vae_sample(const float* mom, const float* noise, float* out, int B, size_t HW, hipStream_t st);
int sdmi_launch_conv1x1_nchw_small(const float* x, const float* w, const float* b, float* y, int B, int Cin, int Cout,
                                   size_t HW, float in_scale, hipStream_t st);
int sdmi_launch_cfg_ddpm(const float* eps, int do_cfg, float cfg_scale, float* latents, const float* noise,
                         const float* coef, size_t n, float* eps_out, hipStream_t st);
int sdmi_launch_add_vec(const float* a, const float* b, float* y, size_t n, hipStream_t st);
int sdmi_launch_ln_fold_prep(const void* w_src, int is_f32, const float* gamma, const float* beta, const float* bias,
                            f16* w_out, float* g_out, float* h_out, int N, int C, hipStream_t st);
int sdmi_launch_compose_linear(const void* A, int a_f32, const void* B, int b_f32, void* out, int out_f32, int N, int K, int J,
                               int ldo, hipStream_t st);
int sdmi_launch_compose_bias(const void* A, int a_f32, const float* b_in, const float* b_out, float* out, int N, int K,
                             hipStream_t st);
int sdmi_launch_cast_rows(const void* src, int is_f32, f16* dst, int rows, int cols, int ld, hipStream_t st);
int sdmi_launch_transpose_scale(const void* src, int is_f32, f16* dst, int R, int Cc, float scale, hipStream_t st);
int sdmi_launch_xattn_mask(const f16* k, const f16* v, f16* dk, f16* dv, int B, int H, int d, int kv_rows, int n_valid,
                           hipStream_t st);
int sdmi_launch_pack_ups_phase(const void* w, int w_f32, f16* out, int O, int I, hipStream_t st);
int sdmi_launch_splitk_finalize(const GemmArgs& a, hipStream_t st);
// can the split-K combine take GroupNorm statistics for this shape?  T_out: record rows per image it will write (parts = 1)
bool sdmi_finalize_gacc_ok(const GemmArgs& a, int* T_out = nullptr);
// record rows per image (GnRec::T) / parts the one-pass epilogue of tile config `cfg` writes
int sdmi_gemm_gacc_T(const GemmArgs& a, int cfg);
bool sdmi_gemm_gna_ok(const GemmArgs& a, int cfg);     // this config can normalise its A fragments for this launch (GemmArgs::gna_rec)
int sdmi_gemm_pick_cfg(const GemmArgs& a);       // the heuristic tile (cfg < 0)
