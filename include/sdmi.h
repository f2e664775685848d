/*
 * sdmi.h -- C ABI of the MI355X-native Stable Diffusion denoising path (libsdmi.so).
 *
 * Plain C: pointers + sizes only, no torch/C++ types.  All *_dev pointers are HIP device pointers
 * owned by the caller; `stream` is a hipStream_t passed as void* (NULL = default stream).  Every call
 * is asynchronous with respect to the host (stream-ordered) unless stated.  Return value: 0 on
 * success, negative errno-style code otherwise (-22 EINVAL, -12 ENOMEM, -5 HIP failure, -2 missing
 * tensor); sdmi_last_error() returns the thread-local message.  One handle per device; a handle is
 * not thread-safe.
 *
 * The reference (dawmro/pytorch_stable_diffusion) has no FFI; the seam this ABI sits behind is its
 * Python call surface:
 *   models["diffusion"](latent, context, time)      sd/pipeline.py:225  -> sd/diffusion.py:797-837
 *   the CFG combine + sampler.step of the loop      sd/pipeline.py:230-237 -> sd/ddpm.py:102-139
 * and its weight ABI is the state-dict key set of sd/model_converter.py:13-650,1009-1024.
 */
#ifndef SDMI_H
#define SDMI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDMI_F32 0
#define SDMI_F16 1

/* sdmi_unet_create flags */
#define SDMI_FLAG_STREAM_F32 1 /* keep the residual stream in fp32 (fp16 shadow feeds the MFMA operands) */
#define SDMI_FLAG_PARTIAL 2    /* allow a subset of the 654 tensors (block-level tests) */
#define SDMI_FLAG_NO_TUNE 4    /* skip the per-shape tile/split-K autotune (heuristic configs) */
#define SDMI_FLAG_ACCURATE 8   /* ACCURATE mode: every GEMM / conv reads its activation operand in fp32 and multiplies it as a hi + lo
                                  fp16 pair (two MFMAs per fragment, fp32 accumulate) against the fp16 weights; fp32 tensors between
                                  the kernels; no LayerNorm fold / back-to-back / halo / folded-cross-attention forms.  What remains
                                  of the fp16 path's error is the weights' own fp16 rounding: for validation and for weight laws on
                                  which fp16 activations miss the 1e-3 pixel tolerance.  Several times slower (bench.py accurate_mode) */

typedef struct sdmi_unet sdmi_unet;

/* One tensor of the reference's Diffusion state dict, in PyTorch layout on the device:
 * conv weights OIHW, linear weights [out][in], fused self-attention in_proj rows ordered q|k|v
 * (sd/model_converter.py:1009). */
typedef struct sdmi_tensor_desc {
  const char* name;
  const void* data_dev;
  int dtype; /* SDMI_F32 or SDMI_F16 */
  int ndim;
  int64_t shape[4];
} sdmi_tensor_desc;

const char* sdmi_last_error(void);
int sdmi_version(void);

/* Replaces Diffusion.__init__ + load_state_dict (sd/diffusion.py:797-812, sd/model_loader.py:38-42):
 * packs the weights into fp16 kernel layouts ([N][kh][kw][Cin]); the library owns the packed copies
 * and its activation arena.  Synchronous. */
int sdmi_unet_create(const sdmi_tensor_desc* tensors, int n_tensors, int flags, sdmi_unet** out);
void sdmi_unet_destroy(sdmi_unet* u);
/* A second lane over the SAME packed weights (borrowed: `src` must outlive the clone and is destroyed after it): own
 * activation arena, split-K slabs, context / schedule buffers.  Two lanes driven on two streams run two independent
 * denoising loops (sd/pipeline.py:146: every generate() is independent) concurrently on one GPU. */
int sdmi_unet_clone(const sdmi_unet* src, sdmi_unet** out);

/* Hoists the 32 loop-invariant cross-attention K/V projections of the context
 * (sd/attention.py:221-222 via sd/diffusion.py:338).  ctx_dev: (batch, n_tokens, 768) fp32. */
int sdmi_unet_set_context(sdmi_unet* u, const float* ctx_dev, int batch, int n_tokens, void* stream);

/* Precomputes TimeEmbedding (sd/diffusion.py:64-76) and the 22 per-ResBlock SiLU->linear_time
 * vectors (sd/diffusion.py:184-187) for every step of a schedule.
 * temb_dev: (n_steps, 320) fp32 = get_time_embedding(t) rows (sd/pipeline.py:310-349). */
int sdmi_unet_set_schedule(sdmi_unet* u, const float* temb_dev, int n_steps, void* stream);

/* Replaces Diffusion.forward (sd/diffusion.py:797-837) on NCHW fp32 I/O like the reference.
 * latents_dev: (latent_batch, 4, h, w) fp32, latent_batch divides batch: image b reads latent b mod latent_batch, i.e. the CFG
 * `repeat(2,1,1,1)` (sd/pipeline.py:221) without a copy (latent_batch == 1 with batch == 2 in generate()).  Time input: temb_dev != NULL -> (1,320)
 * fp32 embedding used directly; else row step_idx of the schedule.  eps_out_dev: (batch,4,h,w) fp32. */
int sdmi_unet_forward(sdmi_unet* u, const float* latents_dev, int latent_batch, const float* temb_dev,
                      int step_idx, float* eps_out_dev, int batch, int h, int w, void* stream);

/* Replaces the CFG combine (sd/pipeline.py:230-233) + DDPMSampler.step (sd/ddpm.py:102-139), fused.
 * eps_dev: (2,4,h,w) if do_cfg else (1,4,h,w); latents_dev updated in place; noise_dev: N(0,1) draw
 * or NULL when t == 0; coef = {sqrt(1-abar_t), sqrt(abar_t), pred_original_sample_coeff,
 * current_sample_coeff, sqrt(variance)} evaluated by the host in the reference's fp32 order;
 * n = 4*h*w elements.  eps_out_dev (optional) receives the guided eps. */
int sdmi_cfg_ddpm_step(const float* eps_dev, int do_cfg, float cfg_scale, float* latents_dev,
                       const float* noise_dev, const float* coef, int64_t n, float* eps_out_dev, void* stream);

/* One whole denoising step = sdmi_unet_forward(step_idx) + sdmi_cfg_ddpm_step on the handle's own
 * eps buffer (the loop body of sd/pipeline.py:208-237). */
int sdmi_unet_denoise_step(sdmi_unet* u, float* latents_dev, int step_idx, int do_cfg, float cfg_scale,
                           const float* noise_dev, const float* coef, int h, int w, void* stream);
/* The same for n_prompts independent prompts through ONE chain (throughput mode; sd/pipeline.py:146: generate() itself is
 * batch 1 per call): latents_dev / noise_dev are (n_prompts,4,h,w), the UNet runs batch 2*n_prompts (n_prompts without
 * guidance) in the order cat([cond_0..cond_P-1, uncond_0..uncond_P-1]) -- the context given to sdmi_unet_set_context must
 * have that batch and order -- and every prompt is at step step_idx of the one schedule (coef as above).  2*n_prompts <= 16. */
int sdmi_unet_denoise_step_batch(sdmi_unet* u, float* latents_dev, int n_prompts, int step_idx, int do_cfg, float cfg_scale,
                                 const float* noise_dev, const float* coef, int h, int w, void* stream);

/* Block-level entry points (parity tests): run ONE reference sub-module on NHWC fp32 device tensors.
 * kind: 0 = UNET_ResidualBlock (sd/diffusion.py:145-209; time_dev = (1,1280) TimeEmbedding output),
 *       1 = UNET_AttentionBlock (sd/diffusion.py:271-381; uses the context set by set_context),
 *       2 = Upsample (sd/diffusion.py:412-435), 3 = 3x3 conv stride `arg`, 4 = UNET_OutputLayer
 *       (out is NCHW fp32 (B,4,H,W)).  x1_dev/c1: optional second concat source for kind 0. */
int sdmi_unet_run_block(sdmi_unet* u, const char* prefix, int kind, int arg, const float* x0_dev, int c0,
                        const float* x1_dev, int c1, int batch, int h, int w, const float* time_dev,
                        float* out_dev, void* stream);

/* Per-launch HIP-event profiling of the forward (bench.py roofline): enable, run forwards, then read
 * summed milliseconds / algorithmic FLOPs / launch counts into arrays of FOUR classes: 0 = MFMA GEMM kernels
 * (implicit-GEMM conv + linear, halo conv, back-to-back GEMM), 1 = flash attention, 2 = norms, 3 = splitk_finalize (the
 * combine launches of split-K GEMMs).  Events are recorded on the forward's own stream. */
int sdmi_unet_profile(sdmi_unet* u, int enable);
int sdmi_unet_profile_read(sdmi_unet* u, double* ms_by_class, double* flops_by_class, int* launches_by_class);

/* Guard of the LayerNorm fold (sd/diffusion.py:317-321,334-339,351-356 run as GEMMs on the raw stream: their error grows with
 * |row mean| / sigma): *hits_out = rows with |mean| > 8 sigma (SDMI_LN_GUARD_SIGMA) that folded GEMMs of this handle have met since
 * the last reset; reset != 0 clears the counter; fold_on = 0 / 1 switches the handle to the separate LayerNorm kernel / back
 * (-1: unchanged).  Synchronises the stream: read it when a loop is over and repeat the loop unfused when it is not zero
 * (Diffusion.denoise_native does). */
int sdmi_unet_ln_guard(sdmi_unet* u, int* hits_out, int reset, int fold_on, void* stream);

/* Number of kernel launches enqueued by the last sdmi_unet_forward, and bytes of packed weights. */
int sdmi_unet_last_launch_count(const sdmi_unet* u);
int64_t sdmi_unet_weight_bytes(const sdmi_unet* u);
/* GEMM shapes this handle had to time itself so far (0 when every shape came from the plan tables: the table
 * shipped as <package>/plans/gfx950.txt or the per-library cache ~/.cache/sdmi/plans-<hash>.txt), and the HIP
 * device the handle lives on.  Every entry point that launches work returns -22 unless that device is current. */
int sdmi_unet_tuned_shapes(const sdmi_unet* u);
int sdmi_unet_device(const sdmi_unet* u);
/* FNV-1a 64 of the loaded libsdmi.so: the key of the per-library plan cache above and of the rocprofv3 counter profiles
 * bench.py quotes (profiles/<round>_*.json carry the hash of the binary they were taken with). */
uint64_t sdmi_library_hash(void);
/* Activation arena of the handle: bytes allocated and the high-water mark of any forward so far.  The arena is sized by the
 * forward that is about to run (256 KiB per latent pixel and image; SDMI_ARENA_GB, read at create, is a minimum), so a handle
 * made for single prompts grows by itself when a batched chain (sdmi_unet_denoise_step_batch at P > 4) first comes through. */
int sdmi_unet_arena(const sdmi_unet* u, int64_t* capacity_out, int64_t* peak_out);

/* ---- VAE decoder (next row after the hot path; reference sd/decoder.py:342-374) ------------------
 * tensors: the 136-entry state dict of VAE_Decoder (keys "0.weight" ... "25.bias",
 * sd/model_converter.py:750-880,883-884,1028-1030).  latents_dev: (B,4,h,w) NCHW fp32 that the caller has
 * already divided by 0.18215 (the reference does so in place, sd/decoder.py:364); image_dev: (B,3,8h,8w)
 * NCHW fp32.  Reproduces the reference's VAE_AttentionBlock quirks (no groupnorm; reinterpreting view). */
typedef struct sdmi_vae sdmi_vae;
int sdmi_vae_decoder_create(const sdmi_tensor_desc* tensors, int n_tensors, int flags, sdmi_vae** out);
void sdmi_vae_destroy(sdmi_vae* v);
int sdmi_vae_decode(sdmi_vae* v, const float* latents_dev, float* image_dev, int batch, int h, int w, void* stream);
int sdmi_vae_last_launch_count(const sdmi_vae* v);
/* VAE encoder (reference sd/encoder.py:95-155; 104-entry state dict, keys "0.weight" ... "18.bias").
 * image_dev: (B,3,H,W) NCHW fp32 in [-1,1]; noise_dev, latents_dev: (B,4,H/8,W/8) fp32.  Includes the asymmetric
 * (0,1,0,1) pad before each stride-2 conv, the logvar clamp, the reparameterisation and the 0.18215 scale. */
int sdmi_vae_encoder_create(const sdmi_tensor_desc* tensors, int n_tensors, int flags, sdmi_vae** out);
int sdmi_vae_encode(sdmi_vae* v, const float* image_dev, const float* noise_dev, float* latents_dev, int batch, int H, int W,
                    void* stream);

/* ---- CLIP text encoder (reference sd/clip.py:227-261) ------------------------------------------------
 * tensors: the 148-entry state dict of CLIP (sd/model_converter.py:885-1008,1031-1054).
 * tokens_dev: (batch, 77) int64 token ids; out_dev: (batch, 77, 768) fp32. */
typedef struct sdmi_clip sdmi_clip;
int sdmi_clip_create(const sdmi_tensor_desc* tensors, int n_tensors, int flags, sdmi_clip** out);
void sdmi_clip_destroy(sdmi_clip* c);
int sdmi_clip_encode(sdmi_clip* c, const int64_t* tokens_dev, float* out_dev, int batch, void* stream);
int sdmi_clip_last_launch_count(const sdmi_clip* c);

/* ---- kernel-level entry points (parity tests / micro-benchmarks) ------------------------------ */

/* Implicit-GEMM conv / linear:  out[m][n] = sum_k A(m,k) w[n][k] + bias[n] + res[m][n].
 * a0/a1: NHWC fp16 sources (virtual channel concat), w: packed [N][ks*ks*(c0+c1)] fp16 with k ordered
 * (kh,kw,ci).  cfg < 0: heuristic tile; ksplit >= 1.  out_t (optional): columns >= nt0 written
 * transposed as out_t[(b*(N-nt0)+n-nt0)*ldt + pos(s)] with m = b*S + s; pos(s) = s, or with out_t_perm the four 4-key
 * quads of every 16-key group in the order (q0, q2, q1, q3) -- the V^T layout sdmi_op_attention reads (ldt % 16 == 0). */
typedef struct sdmi_gemm_desc {
  const void* a0; const void* a1;
  int c0, c1, hs, ws, ho, wo, ups, stride, pad, ks;
  int M, N, K;
  const void* w;
  const float* bias;
  const void* res; int res_f32; int ldr;
  void* out; int out_f32; int ldc;
  void* out16;
  void* out_t; int nt0; int S; int ldt;
  int cfg; int ksplit;
  /* optional extra 1x1 K-range after the ks*ks taps: x0 | x1 ([M][cx0], [M][cx1]) read at the output pixel;
   * K = ks*ks*(c0+c1) + cx0 + cx1 (the ResBlock skip conv fused into conv_merged, sd/diffusion.py:143,209) */
  const void* x0; const void* x1; int cx0, cx1;
  /* LayerNorm folded around the GEMM (sd/diffusion.py:317-321,334-339,351-356).  Producer: rowstat != NULL ->
   * the epilogue also writes per-row {sum, sum of squares} of the fp16 output over each n-tile to
   * rowstat[(m*tiles_n + tn)*2] (tiles_n = ceil(N / tile width of cfg); ksplit 1, no out_t).  Consumer:
   * ln_stat != NULL -> a0 is the raw (un-normalised) tensor, w/ln_g/bias come from sdmi_op_ln_fold_prep and
   * out = rstd[m]*(acc - mean[m]*ln_g[n]) + bias[n], statistics summed from ln_stat[m][0..ln_ntn) over ln_c columns. */
  float* rowstat;
  const float* ln_stat; int ln_ntn; const float* ln_g; int ln_c; float ln_eps;
  int out_t_perm;   /* 1: out_t's key axis in the quad-permuted order sdmi_op_attention reads; 0: natural order */
  /* act = 2: the epilogue is a row softmax in the log2 domain, out = 2^(v - max) / sum over each 128-column n-tile with
   * columns >= sm_valid (per tile) masked: one attention head per tile (128-wide configs, fp16 out, full tiles).
   * img_rows > 0: per-image weights, rows [i*img_rows, (i+1)*img_rows) use w + i*w_img_stride (elements, row stride ldw
   * or K when 0) and bias / ln_g + i*vec_img_stride.  Together: the cross-attention of sd/attention.py:219-256 with
   * q_proj / out_proj folded into the per-prompt K / V (two GEMMs). */
  int act, sm_valid, img_rows, w_img_stride, vec_img_stride, ldw;
  /* phase2 = 1: nearest x2 upsample + 3x3 conv (sd/diffusion.py:430-435) as four 2x2 convs on the SOURCE grid, one per
   * output parity: ks = 2, hs x ws = ho x wo = source size, M = 4*B*hs*ws (rows ordered phase, b, y, x), K = 4*(c0+c1),
   * w = sdmi_op_pack_ups_phase's [4][N][2][2][C], img_rows = M/4, w_img_stride = N*K; out is the (B, 2hs, 2ws, N) map. */
  int phase2;
  /* ln_ksteps > 0: only the first ln_ksteps K-steps (of 64) are the LayerNorm-folded range, a plain product follows in the
   * same accumulator (the composed feed-forward).  With ksplit > 1 the K-slices must end on that boundary and
   * ln_out ([M][2] fp32 scratch) must be given. */
  int ln_ksteps; float* ln_out;
  /* gacc != NULL: the launch also leaves the GroupNorm statistics of what it writes (sd/diffusion.py:173,199,294: every
   * GroupNorm input is a conv / linear output): per image, per block of rows one workgroup owns ("record row") and per atom of
   * gacc_atom consecutive channels the moments {sum, sum of squares} as float2,
   *   gacc[((image * T + t) * (N / gacc_atom) + atom) * parts + part]
   * with T and parts as sdmi_op_gemm_stat_layout reports them for this descriptor (parts = 2: an atom that straddles two column
   * tiles has its moments split over the two parts; they add up).  Plain stores, every slot written exactly once: nothing to
   * zero, no dependence on workgroup order.  gacc_rows_img = rows (pixels) per image; the row blocks never straddle images. */
  float* gacc; int gacc_atom; int gacc_rows_img;
  /* ln_guard != NULL (with ln_stat): every row whose |mean| exceeds ln_guard_sigma standard deviations adds 1 to *ln_guard
   * (the guard of the LayerNorm fold: sdmi_unet_ln_guard) */
  int* ln_guard; float ln_guard_sigma;
  /* gna_rec != NULL: GroupNorm (32 groups, no SiLU; sd/diffusion.py:294-298) of the A operand applied inside the launch -- a0 is the RAW
   * fp16 tensor, a plain 1x1 GEMM with K = c0 -- from the statistics records its producer left (layout as gacc above: gna_t record
   * rows per image, gna_parts, atoms of gna_atom channels, gna_rows rows per image), gamma / beta [c0] fp32.  Configs built with the
   * variant only (the "p" / "q2" rings with 64-row tiles); others fail with "cannot apply GroupNorm". */
  const float* gna_rec; const float* gna_gamma; const float* gna_beta; float gna_eps; int gna_t, gna_parts, gna_atom, gna_rows;
  /* accurate != 0 (the kernels of SDMI_FLAG_ACCURATE): the A operand is read from the fp32 tensors a0f / a1f / x0f / x1f (same
   * shapes and strides as a0 / a1 / x0 / x1, which are then unused) and multiplied as a hi + lo fp16 pair against the fp16
   * weights.  cfg < 0 picks the mode's own tile and split-K factor; an explicit cfg must be one built with the variant. */
  const float* a0f; const float* a1f; const float* x0f; const float* x1f; int accurate;
  /* hgn_x0 != NULL (3x3 stride-1 convs on a halo-reuse config built with the variant): the conv's input is
   * GroupNorm(32)(+SiLU if hgn_silu) of the RAW NHWC tensor(s) hgn_x0 | hgn_x1 (hgn_c0 | hgn_c1 channels, fp32 if hgn_in_f32 else
   * fp16, dense), normalised inside the launch by the kernel's producer waves -- sd/diffusion.py:173-179,199-205 as one launch.
   * c0 must be hgn_c0 + hgn_c1 and c1 = 0 (a0 is not read).  Statistics: the records the tensors' producers left (layout as gacc
   * above: hgn_t0 / hgn_t1 record rows per image, hgn_p0 / hgn_p1 parts, atoms of hgn_atom channels); gamma / beta [c0] fp32. */
  const void* hgn_x0; const void* hgn_x1; int hgn_in_f32, hgn_c0, hgn_c1;
  const float* hgn_gamma; const float* hgn_beta; float hgn_eps; int hgn_silu;
  const float* hgn_rec0; const float* hgn_rec1; int hgn_t0, hgn_t1, hgn_p0, hgn_p1, hgn_atom;
} sdmi_gemm_desc;
int sdmi_op_gemm(const sdmi_gemm_desc* d, void* stream);
/* record rows per image (T) and parts of the statistics the launch described by d writes to d->gacc (d->gacc != NULL);
 * the buffer needs images * T * (N / gacc_atom) * parts float2.  Fails when the tile config / split-K of d cannot take them. */
int sdmi_op_gemm_stat_layout(const sdmi_gemm_desc* d, int* T, int* parts);
/* iters back-to-back launches of the same GEMM between two HIP events -> microseconds per launch; iters < 0: -iters launches
 * timed one by one with the L2s evicted before each (64 MiB fill), minimum returned */
int sdmi_bench_gemm(const sdmi_gemm_desc* d, int iters, float* us_per_iter, void* stream);
/* Back-to-back GEMM of the 320-channel attention blocks (csrc/b2b.hip; sd/diffusion.py:321-363): S = a1 w1^T + b1 + r1, then
 * Y = cscale * (LN0(S) w2'^T + h2) [partial = 0, K2 = 320] or Y = LN0(S) w2a'^T + S w2b^T + h2 + r2 [partial = 1, K2 = 640,
 * w2 = [w2a' | w2b]] with LN0 = LayerNorm without affine (w2' / h2 from sdmi_op_ln_fold_prep); every matrix
 * has 320 columns (M % 32 == 0).  s32 / s16: optional copies of S.  iters >= 1 launches; us_per_iter (optional) = time per launch. */
typedef struct sdmi_b2b_desc {
  const void* a1; int lda1;
  const void* w1; const float* b1;
  const void* r1; int r1_f32;
  float* s32; void* s16;
  const void* w2; int K2;
  const float* h2;
  int partial;
  float cscale;
  const void* r2; int r2_f32;
  void* out; int out_f32; void* out16;
  int M;
  float eps;
  int bm;          /* rows per workgroup: 32, 64, or 0 = chosen from M */
  /* npass2 = 3 (0 / 1: one pass): the second product is the attention in_proj, w2 = [960][320]: q (x cscale) and k go to
   * out[M][ldo] (fp16) at columns 0 and 320, v is written transposed to vt[(b*320 + n)*ldt + pos(s)] (m = b*S + s, S % 32 == 0)
   * in the key order sdmi_op_attention reads.  ldo = 0: 320. */
  int npass2, ldo;
  void* vt; int S, ldt;
  /* gx != NULL: the first product's A operand is GroupNorm(gx) (32 groups, eps gn_eps, no SiLU: the attention block's
   * groupnorm, sd/diffusion.py:294,312) computed in the kernel from the raw [M][320] tensor gx (fp32 if gx_f32) and the
   * partial statistics sdmi_op_gn_stats wrote ([B][gn_nchunk][32][2]); images of S rows, S % 32 == 0; a1 is ignored. */
  const void* gx; int gx_f32;
  const float* gn_partial; int gn_nchunk; const float* gn_gamma; const float* gn_beta; float gn_eps;
  /* GroupNorm statistics of `out` as in sdmi_gemm_desc (one-pass form, 32-row tiles: M <= 8192): T = gacc_rows_img / 32, parts = 1 */
  float* gacc; int gacc_atom; int gacc_rows_img;
} sdmi_b2b_desc;
int sdmi_op_b2b(const sdmi_b2b_desc* d, int iters, float* us_per_iter, void* stream);
int sdmi_gemm_num_configs(void);
const char* sdmi_gemm_config_name(int cfg);

/* tile (rows, columns) of GEMM config cfg */
void sdmi_gemm_config_dims(int cfg, int* bm, int* bn);
/* Linear [N][C] (first N rows of w_dev) with a LayerNorm(gamma, beta) folded in: w_out = fp16(gamma (.) W),
 * g_out[n] = sum_c w_out[n][c], h_out[n] = bias[n] + sum_c beta[c] W[n][c]  (bias may be NULL). */
int sdmi_op_ln_fold_prep(const void* w_dev, int w_dtype, const float* gamma, const float* beta, const float* bias,
                         void* w_out, float* g_out, float* h_out, int N, int C, void* stream);

/* PyTorch OIHW (fp32/fp16) -> packed [o < o_keep][kh][kw][I] fp16. */
int sdmi_op_pack_conv(const void* w_dev, int w_dtype, void* out_dev, int O, int I, int ks, int o_keep, void* stream);
/* OIHW 3x3 weights -> the four phase matrices of sdmi_gemm_desc::phase2: out[4][O][2][2][I] fp16 (taps that land on the same
 * source pixel after the nearest upsample summed in fp32) */
int sdmi_op_pack_ups_phase(const void* w_dev, int w_dtype, void* out_dev, int O, int I, void* stream);

/* Flash attention: q [B*Sq][ldq], k [B*k_batch_stride][ldk], vt [(b*H+h)*d+dd][ldvt] (keys along the row in the
 * quad-permuted order sdmi_op_gemm's out_t writes, zero-padded to a multiple of 64), o [B*Sq][ldo]; all fp16.
 * q is plain: the kernel applies 1/sqrt(d) itself on this entry point. */
int sdmi_op_attention(const void* q, int ldq, const void* k, int ldk, int k_batch_stride, const void* vt, int ldvt,
                      void* o, int ldo, int B, int H, int d, int Sq, int Skv, void* stream);

/* GroupNorm(32) over NHWC (optionally two concat sources), fp16 out; LayerNorm over rows. */
int sdmi_op_groupnorm(const void* x0, const void* x1, int in_f32, int c0, int c1, int B, int P,
                      const float* gamma, const float* beta, float eps, int silu, void* y_f16, void* stream);
/* GroupNorm(32)(+SiLU) of x = sum_z slab[z][B*P][C] + bias[C] (+ res[B*P][C]), slabs added in order: the split-K combine of a conv
 * (sd/diffusion.py:179, 205) and the GroupNorm that follows it (:199, :294) in one launch.  P <= 1024, (C/32) % 4 == 0.
 * out32 / out16 (optional): copies of x for other readers. */
int sdmi_op_groupnorm_slab(const float* slab, int ksplit, const float* bias, const void* res, int res_f32, int C, int B, int P,
                           const float* gamma, const float* beta, float eps, int silu, void* y_f16, float* out32, void* out16,
                           void* stream);
/* GroupNorm(32)(+SiLU) whose statistics were left behind by the producers of x0 / x1 (sdmi_gemm_desc::gacc): one normalising
 * pass, no statistics launch.  rec0 / rec1 with their T and parts as the producers wrote them; (c0 + c1) / 32 and c0 are
 * multiples of atom, at most 8 atoms per group. */
int sdmi_op_groupnorm_acc(const void* x0, const void* x1, int in_f32, int c0, int c1, int B, int P, const float* rec0, int T0,
                          int parts0, const float* rec1, int T1, int parts1, int atom, const float* gamma, const float* beta,
                          float eps, int silu, void* y_f16, void* stream);
int sdmi_op_layernorm(const void* x, int in_f32, int M, int C, const float* gamma, const float* beta, float eps,
                      void* y_f16, void* stream);
/* GroupNorm statistics only: per (image, pixel chunk, group) partial {sum, sum of squares} into partial_out
 * ([B][nchunk][32][2] fp32, nchunk = sdmi_gn_num_chunks(P)); consumed by sdmi_op_b2b's gn_partial. */
int sdmi_gn_num_chunks(int P);
int sdmi_op_gn_stats(const void* x0, const void* x1, int in_f32, int c0, int c1, int B, int P, float* partial_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDMI_H */
