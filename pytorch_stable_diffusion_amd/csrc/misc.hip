// Small kernels around the GEMM/attention core (gfx950):
//   weight packing, casts, the timestep-embedding GEMV chain, the 4->320 stem conv that reads NCHW
//   fp32 latents directly, the 320->4 output conv that writes NCHW fp32, and the fused
//   classifier-free-guidance combine + DDPM ancestral step.
#include "common.h"
#include <type_traits>

namespace {

__global__ void cast_f32_f16_kernel(const float* x, f16* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = (f16)x[i];
}

__global__ void cast_any_f32_kernel(const void* x, int in_f32, float* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = in_f32 ? ((const float*)x)[i] : (float)((const f16*)x)[i];
}

// PyTorch conv weight [O][I][ks][ks] -> packed [o < o_keep][kh][kw][I] fp16 (K ordered (kh,kw,ci)).
// Linear weights are the ks = 1 case.  (reference weight ABI: sd/model_converter.py:13-650)
__global__ void pack_conv_kernel(const void* w, int w_f32, f16* out, int O, int I, int ks, int o_keep) {
  const size_t total = (size_t)o_keep * ks * ks * I;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(idx % I);
    size_t t = idx / I;
    const int kw = (int)(t % ks);
    t /= ks;
    const int kh = (int)(t % ks);
    const int o = (int)(t / ks);
    const size_t src = (((size_t)o * I + ci) * ks + kh) * ks + kw;
    out[idx] = w_f32 ? (f16)((const float*)w)[src] : ((const f16*)w)[src];
  }
}

// y[m][n] = sum_k act(x[m][k]) W[n][k] + b[n];  one wave per n, W row kept in registers, loop m.
// Timestep path: sd/diffusion.py:64-76 (TimeEmbedding) and :184-187 (SiLU -> linear_time).
template <int NCH>
__global__ __launch_bounds__(256) void small_linear_kernel(const float* x, const f16* w, const float* b, float* y,
                                                           int M, int N, int K, int silu, int ldy) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int K8 = K / 8;
  float wv[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c8 = lane + i * 64;
    if (c8 < K8) {
      const f16x8 t = *(const f16x8*)(w + (size_t)n * K + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[i][e] = (float)t[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[i][e] = 0.f;
    }
  }
  const float bias = b ? b[n] : 0.f;
  for (int m = 0; m < M; ++m) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c8 = lane + i * 64;
      if (c8 < K8) {
        const float* xp = x + (size_t)m * K + c8 * 8;
        const f32x4 a0 = *(const f32x4*)xp, a1 = *(const f32x4*)(xp + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = e < 4 ? a0[e] : a1[e - 4];
          if (silu) v = v / (1.f + expf(-v));
          acc += v * wv[i][e];
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) y[(size_t)m * ldy + n] = acc + bias;
  }
}

// Stem conv 4 -> Cout, 3x3 s1 p1 (sd/diffusion.py:545), input NCHW fp32 latents (batch-broadcast
// implements latents.repeat(2,1,1,1) of sd/pipeline.py:221 without a copy), output NHWC.
// w36: [36][Cout] fp32 with k = (kh*3+kw)*4 + ci.
// A lane is a pixel, a wave takes `cpw` chunks of 8 output channels for its 64 pixels: the pixel's 3x3 x Cin neighbourhood
// is loaded once (<= 36 loads in flight, zeros outside the map) and the weights are wave-uniform: staged once per wave in LDS and
// read as broadcasts (the form with a thread per (pixel, chunk) read 36 x 32 B per thread through the texture path, the form with
// scalar loads waited for a chain of scalar-cache misses: 18 us a step either way).  Products are added in (kh, kw, ci) order onto
// the bias, as before.
constexpr int STEM_CPW_MAX = 8;      // chunks of 8 output channels per wave (36 KiB of weights in LDS per workgroup at Cin = 4)
// GACC: the wave also leaves the GroupNorm statistics of what it stores (GnRec, common.h): it then takes exactly 5 chunks = 40
// channels = four 10-channel atoms for its 64 pixels, so a record (64-pixel row block, atom) is one wave's business.
template <int CIN, bool GACC>
__global__ __launch_bounds__(256, 4) void stem_conv_kernel(const float* __restrict__ lat, int lat_batch, const float* __restrict__ w36,
                                                        const float* __restrict__ bias, void* out, int out_f32, f16* out16,
                                                        int B, int H, int W, int Cout, int cpw, int nchunk, float* rec, int rec_T) {
  constexpr int Cin = CIN;
  const int lane = threadIdx.x & 63;
  const unsigned gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
  const unsigned grp = gw / (unsigned)nchunk;
  const int chunk = (int)(gw - grp * nchunk);
  const unsigned npix = (unsigned)B * H * W;                           // < 2^31: 32-bit index math
  const unsigned pix = grp * 64u + lane;
  if (grp * 64u >= npix) return;
  const bool live = pix < npix;
  const unsigned pc = live ? pix : npix - 1;
  const unsigned prow = pc / (unsigned)W;
  const int ow = (int)(pc - prow * W);
  const int b = (int)(prow / (unsigned)H);
  const int oh = (int)(prow - (unsigned)b * H);
  const int lb = b % lat_batch;                 // latents.repeat(batch / lat_batch, 1, 1, 1): image b reads latent b mod lat_batch
  f32x2 xv[9][CIN];                              // (x, x): the packed FMA's multiplicand, made once per value
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
    const bool in = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const float v = lat[(((unsigned)lb * Cin + ci) * H + (in ? ih : 0)) * W + (in ? iw : 0)];   // always a valid address
      xv[t][ci] = in ? f32x2{v, v} : f32x2{0.f, 0.f};
      asm volatile("" : "+v"(xv[t][ci]));      // one register pair per value (the compiler otherwise keeps a copy per use and spills)
    }
  }
  const int C8 = Cout / 8;
  const int c_begin = chunk * cpw, c_end = min(C8, (chunk + 1) * cpw);
  // this wave's weights -> its own LDS region as [chunk of 8 channels][k][8] (coalesced 32-byte runs of w36); the products below
  // then read them as wave-wide broadcasts.  (Straight from memory the 36 x 32 B per chunk are a chain of scalar-cache misses.)
  __shared__ __attribute__((aligned(16))) float s_w[4][STEM_CPW_MAX * 36 * CIN / 4 * 8];
  float* sw = s_w[threadIdx.x >> 6];
  {
    const int nc = (c_end - c_begin) * 8, nk = 9 * CIN;
    for (int idx = lane; idx < nk * nc; idx += 64) {
      const int k = idx / nc, c = idx - k * nc;
      sw[((c >> 3) * nk + k) * 8 + (c & 7)] = w36[(unsigned)(k * Cout + c_begin * 8 + c)];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // same wave reads what its lanes wrote: LDS ops of a wave are in order
  }
  float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};     // GACC: moments of the wave's four atoms, this lane's pixel
  auto do_chunk = [&](int c8, auto JJ) {
    constexpr int jj = decltype(JJ)::value;        // chunk index inside the wave's 40 channels (GACC: compile-time atom of every column)
    f32x2 a2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a2[j] = *(const f32x2*)(bias + c8 * 8 + 2 * j);
    const float* wc = sw + (c8 - c_begin) * 9 * CIN * 8;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const f32x4 w0 = *(const f32x4*)(wc + (t * CIN + ci) * 8), w1 = *(const f32x4*)(wc + (t * CIN + ci) * 8 + 4);
        a2[0] = __builtin_elementwise_fma(xv[t][ci], f32x2{w0[0], w0[1]}, a2[0]);
        a2[1] = __builtin_elementwise_fma(xv[t][ci], f32x2{w0[2], w0[3]}, a2[1]);
        a2[2] = __builtin_elementwise_fma(xv[t][ci], f32x2{w1[0], w1[1]}, a2[2]);
        a2[3] = __builtin_elementwise_fma(xv[t][ci], f32x2{w1[2], w1[3]}, a2[3]);
      }
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[2 * j] = a2[j][0]; acc[2 * j + 1] = a2[j][1]; }
    if (!live) return;
    if (GACC) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { gs[(jj * 8 + e) / 10] += acc[e]; gq[(jj * 8 + e) / 10] += acc[e] * acc[e]; }
    }
    f16x8 o16;
#pragma unroll
    for (int e = 0; e < 8; ++e) o16[e] = (f16)acc[e];
    const size_t off = (size_t)pix * Cout + c8 * 8;
    if (out_f32) {
      float* op = (float*)out + off;
      *(f32x4*)op = f32x4{acc[0], acc[1], acc[2], acc[3]};
      *(f32x4*)(op + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
      if (out16) *(f16x8*)(out16 + off) = o16;
    } else {
      *(f16x8*)((f16*)out + off) = o16;
    }
  };
  if (GACC) {
    // cpw == 5 (launcher): chunks jj = 0 .. 4 of the wave's 40 channels, unrolled so every column's atom is a constant
    do_chunk(c_begin + 0, std::integral_constant<int, 0>{});
    do_chunk(c_begin + 1, std::integral_constant<int, 1>{});
    do_chunk(c_begin + 2, std::integral_constant<int, 2>{});
    do_chunk(c_begin + 3, std::integral_constant<int, 3>{});
    do_chunk(c_begin + 4, std::integral_constant<int, 4>{});
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { gs[a] += __shfl_xor(gs[a], o); gq[a] += __shfl_xor(gq[a], o); }
    if (lane == 0) {
      const unsigned hw = (unsigned)H * W, p0 = grp * 64u;            // (H * W a multiple of 64: the block lies in one image)
      const unsigned img = p0 / hw, t_row = (p0 - img * hw) >> 6;
      f32x2* r = (f32x2*)rec + ((size_t)img * rec_T + t_row) * (Cout / 10) + chunk * 4;
#pragma unroll
      for (int a = 0; a < 4; ++a) r[a] = f32x2{gs[a], gq[a]};
    }
  } else {
    for (int c8 = c_begin; c8 < c_end; ++c8) do_chunk(c8, std::integral_constant<int, 0>{});
  }
}

__global__ void pack_stem_kernel(const void* w, int w_f32, float* w36, int Cout, int Cin) {
  const int total = 9 * Cin * Cout;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int co = idx % Cout, k = idx / Cout;
    const int ci = k % Cin, tap = k / Cin, kh = tap / 3, kw = tap % 3;
    const size_t src = (((size_t)co * Cin + ci) * 3 + kh) * 3 + kw;
    w36[idx] = w_f32 ? ((const float*)w)[src] : (float)((const f16*)w)[src];
  }
}

// Output conv Cin -> 4, 3x3 s1 p1 (sd/diffusion.py:744) on the GN+SiLU'd NHWC fp16 tensor,
// result NCHW fp32 (B,4,H,W).  One wave per FOUR consecutive output pixels of a row (round 4: one per pixel re-read the
// 23 KB of weights for every pixel -- 190 MB per step out of L1/L2, 16 us): a lane takes (tap, 8-channel chunk) items, loads
// the item's weights once and multiplies them with the four pixels' inputs; w packed [4][3][3][Cin] fp16.
constexpr int FC_PX = 4;
// XF32 (accurate mode): the input is read from its fp32 copy x32 and multiplied in fp32 (v_fma) against the fp16 weights
template <int NCO, bool XF32 = false>        // output channels the kernel holds accumulators for: 4 (UNet eps, VAE image) or 8 (VAE encoder moments)
__global__ __launch_bounds__(256) void final_conv_kernel(const f16* x, const f16* w, const float* bias, float* out,
                                                         int B, int H, int W, int Cin, int Cout, const float* x32 = nullptr) {
  const int lane = threadIdx.x & 63;
  const unsigned grp = blockIdx.x * 4u + (threadIdx.x >> 6);          // group of FC_PX pixels (W % FC_PX == 0: launcher)
  const unsigned ngrp = (unsigned)B * H * W / FC_PX;
  if (grp >= ngrp) return;
  const unsigned pix0 = grp * FC_PX;
  const unsigned prow = pix0 / (unsigned)W;
  const int ow0 = (int)(pix0 - prow * W);
  const int b = (int)(prow / (unsigned)H);
  const int oh = (int)(prow - (unsigned)b * H);
  const int C8 = Cin / 8;
  const int items = 9 * C8;
  float acc[FC_PX][NCO];
#pragma unroll
  for (int px = 0; px < FC_PX; ++px)
#pragma unroll
    for (int co = 0; co < NCO; ++co) acc[px][co] = 0.f;
  for (int it = lane; it < items; it += 64) {
    const int tap = it / C8, c8 = it - tap * C8;
    const int kh = tap / 3, kw = tap - kh * 3;
    const int ih = oh + kh - 1;
    if ((unsigned)ih >= (unsigned)H) continue;
    f16x8 wv[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) wv[co] = co < Cout ? *(const f16x8*)(w + ((size_t)co * 9 + tap) * Cin + c8 * 8) : f16x8{};
    if constexpr (XF32) {
      float xf[FC_PX][8];
#pragma unroll
      for (int px = 0; px < FC_PX; ++px) {
        const int iw = ow0 + px + kw - 1;
        const bool in = (unsigned)iw < (unsigned)W;
        const float* xp = x32 + (((size_t)b * H + ih) * W + (in ? iw : 0)) * Cin + c8 * 8;
        const f32x4 a = in ? *(const f32x4*)xp : f32x4{0.f, 0.f, 0.f, 0.f}, c = in ? *(const f32x4*)(xp + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) { xf[px][e] = a[e]; xf[px][4 + e] = c[e]; }
      }
#pragma unroll
      for (int px = 0; px < FC_PX; ++px)
#pragma unroll
        for (int co = 0; co < NCO; ++co)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[px][co] = fmaf(xf[px][e], (float)wv[co][e], acc[px][co]);
    } else {
    f16x8 xv[FC_PX];
#pragma unroll
    for (int px = 0; px < FC_PX; ++px) {
      const int iw = ow0 + px + kw - 1;
      xv[px] = (unsigned)iw < (unsigned)W ? *(const f16x8*)(x + (((size_t)b * H + ih) * W + iw) * Cin + c8 * 8) : f16x8{};
    }
#pragma unroll
    for (int px = 0; px < FC_PX; ++px)
#pragma unroll
      for (int co = 0; co < NCO; ++co)
#pragma unroll
        for (int e = 0; e < 8; e += 2)      // v_dot2_f32_f16: two fp16 products + fp32 accumulate per instruction
          acc[px][co] = __builtin_amdgcn_fdot2(f16x2{xv[px][e], xv[px][e + 1]}, f16x2{wv[co][e], wv[co][e + 1]}, acc[px][co], false);
    }
  }
#pragma unroll
  for (int px = 0; px < FC_PX; ++px)
#pragma unroll
    for (int co = 0; co < NCO; ++co)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[px][co] += __shfl_xor(acc[px][co], o);
  if (lane == 0) {
#pragma unroll
    for (int co = 0; co < NCO; ++co)
      if (co < Cout) {
        float* op = out + (((size_t)b * Cout + co) * H + oh) * W + ow0;
#pragma unroll
        for (int px = 0; px < FC_PX; ++px) op[px] = acc[px][co] + bias[co];
      }
  }
}

struct DdpmCoef { float sqrt_beta_prod, sqrt_alpha_prod, c0, ct, sigma; int has_noise; };

// CFG combine (sd/pipeline.py:230-233, cond first) + ancestral DDPM step (sd/ddpm.py:116-137).
// Operation order and rounding follow the reference's fp32 tensor expression exactly (no FMA
// contraction), so with identical eps the result is bit-identical to the reference step.
__device__ __forceinline__ float cfg_combine(float e, float eu, float cfg_scale) {
#pragma clang fp contract(off)
  const float d = e - eu;
  const float sd = cfg_scale * d;
  return sd + eu;
}
__device__ __forceinline__ float ddpm_prev(float e, float x, float nz, const DdpmCoef& k) {
#pragma clang fp contract(off)
  const float t1 = k.sqrt_beta_prod * e;
  const float t2 = x - t1;
  const float x0 = t2 / k.sqrt_alpha_prod;
  const float a = k.c0 * x0;
  const float bb = k.ct * x;
  float prev = a + bb;
  if (k.has_noise) {
    const float v = k.sigma * nz;
    prev = prev + v;
  }
  return prev;
}
__global__ __launch_bounds__(256) void cfg_ddpm_kernel(const float* eps, int do_cfg, float cfg_scale, float* latents,
                                                       const float* noise, DdpmCoef k, size_t n, float* eps_out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float e = eps[i];
    if (do_cfg) e = cfg_combine(e, eps[n + i], cfg_scale);
    if (eps_out) eps_out[i] = e;
    latents[i] = ddpm_prev(e, latents[i], k.has_noise ? noise[i] : 0.f, k);
  }
}

// The step's last two launches as one (round 4): the output conv (Cin -> 4, above) of prompt i's conditional image b = i and
// unconditional image b = P + i for four pixels per wave, then the guidance combine and the DDPM update of those 16 latents by
// the wave itself -- the eps tensor never reaches memory.  Per image the sums are taken exactly as final_conv_kernel takes them
// (same items per lane, same reduction), the arithmetic behind them is cfg_combine / ddpm_prev: bit-identical to the two launches.
__global__ __launch_bounds__(256) void final_conv_step_kernel(const f16* x, const f16* w, const float* bias, int P, int H, int W, int Cin,
                                                              int do_cfg, float cfg_scale, float* latents, const float* noise, DdpmCoef k) {
  constexpr int NCO = 4;
  const int lane = threadIdx.x & 63;
  const unsigned grp = blockIdx.x * 4u + (threadIdx.x >> 6);
  const unsigned ngrp = (unsigned)P * H * W / FC_PX;
  if (grp >= ngrp) return;
  const unsigned pix0 = grp * FC_PX;
  const unsigned prow = pix0 / (unsigned)W;
  const int ow0 = (int)(pix0 - prow * W);
  const int b = (int)(prow / (unsigned)H);                   // prompt index
  const int oh = (int)(prow - (unsigned)b * H);
  const int C8 = Cin / 8;
  const int items = 9 * C8;
  // both images in ONE item loop (their loads fly together); per image the items, their order and the reduction are
  // final_conv_kernel's, so each sum has that kernel's bits
  float acc[2][FC_PX][NCO];
#pragma unroll
  for (int im = 0; im < 2; ++im)
#pragma unroll
    for (int px = 0; px < FC_PX; ++px)
#pragma unroll
      for (int co = 0; co < NCO; ++co) acc[im][px][co] = 0.f;
  const int nim = do_cfg ? 2 : 1;
  for (int it = lane; it < items; it += 64) {
    const int tap = it / C8, c8 = it - tap * C8;
    const int kh = tap / 3, kw = tap - kh * 3;
    const int ih = oh + kh - 1;
    if ((unsigned)ih >= (unsigned)H) continue;
    f16x8 wv[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) wv[co] = *(const f16x8*)(w + ((size_t)co * 9 + tap) * Cin + c8 * 8);
    f16x8 xv[2][FC_PX];
#pragma unroll
    for (int im = 0; im < 2; ++im)
#pragma unroll
      for (int px = 0; px < FC_PX; ++px) {
        const int iw = ow0 + px + kw - 1;
        xv[im][px] = (im < nim && (unsigned)iw < (unsigned)W) ? *(const f16x8*)(x + (((size_t)(b + im * P) * H + ih) * W + iw) * Cin + c8 * 8) : f16x8{};
      }
#pragma unroll
    for (int im = 0; im < 2; ++im)
#pragma unroll
      for (int px = 0; px < FC_PX; ++px)
#pragma unroll
        for (int co = 0; co < NCO; ++co)
#pragma unroll
          for (int e = 0; e < 8; e += 2)
            acc[im][px][co] = __builtin_amdgcn_fdot2(f16x2{xv[im][px][e], xv[im][px][e + 1]}, f16x2{wv[co][e], wv[co][e + 1]}, acc[im][px][co], false);
  }
  float eps[2][FC_PX][NCO];
#pragma unroll
  for (int im = 0; im < 2; ++im)
#pragma unroll
    for (int px = 0; px < FC_PX; ++px)
#pragma unroll
      for (int co = 0; co < NCO; ++co) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[im][px][co] += __shfl_xor(acc[im][px][co], o);
        eps[im][px][co] = acc[im][px][co] + bias[co];
      }
  if (lane == 0) {
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
      const size_t o = (((size_t)b * NCO + co) * H + oh) * W + ow0;      // latents / noise: NCHW (P, 4, H, W)
#pragma unroll
      for (int px = 0; px < FC_PX; ++px) {
        float e = eps[0][px][co];
        if (do_cfg) e = cfg_combine(e, eps[1][px][co], cfg_scale);
        latents[o + px] = ddpm_prev(e, latents[o + px], k.has_noise ? noise[o + px] : 0.f, k);
      }
    }
  }
}

__global__ void add_vec_kernel(const float* a, const float* b, float* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = a[i] + b[i];
}


// Row softmax over materialised scores (VAE single-head d=512 attention, sd/attention.py:55-76 via
// sd/decoder.py:57): p = softmax(scale * s) per row, fp16 in / fp16 out, fp32 statistics.
// One wave per row, row length L (multiple of 8).
__global__ __launch_bounds__(256) void row_softmax_kernel(const f16* s, f16* p, int rows, int L, float scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f16* sp = s + (size_t)row * L;
  f16* pp = p + (size_t)row * L;
  const int L8 = L / 8;
  float mx = -INFINITY;
  for (int c8 = lane; c8 < L8; c8 += 64) {
    const f16x8 v = *(const f16x8*)(sp + c8 * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)v[e]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  const float c = scale * 1.4426950408889634f, mc = mx * c;
  float sum = 0.f;
  for (int c8 = lane; c8 < L8; c8 += 64) {
    const f16x8 v = *(const f16x8*)(sp + c8 * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += __builtin_amdgcn_exp2f((float)v[e] * c - mc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.f / sum;
  for (int c8 = lane; c8 < L8; c8 += 64) {
    const f16x8 v = *(const f16x8*)(sp + c8 * 8);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)(__builtin_amdgcn_exp2f((float)v[e] * c - mc) * inv);
    *(f16x8*)(pp + c8 * 8) = o;
  }
}

// Quirk Q4 of the reference's VAE_AttentionBlock (sd/decoder.py:62,67): the (n, h*w, c) attention
// output is REINTERPRETED as (n, c, h, w) by view() and added to the NCHW residual.  In NHWC storage:
//   y[n][p][c] = o_flat[n][c*P + p] + x[n][p][c]        (P = h*w pixels, o fp32 [n][P*C])
__global__ __launch_bounds__(256) void q4_reinterpret_add_kernel(const float* o, const void* x, int x_f32, void* y,
                                                                 int y_f32, f16* y16, int B, int P, int C) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const float* on = o + (size_t)n * P * C;
  for (int j = ty; j < 32; j += 8) {                             // read o as [C][P]: rows c, contiguous p
    const int c = c0 + j, pp = p0 + tx;
    tile[j][tx] = (c < C && pp < P) ? on[(size_t)c * P + pp] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {                             // write y as [P][C]: rows p, contiguous c
    const int pp = p0 + j, c = c0 + tx;
    if (pp < P && c < C) {
      const size_t idx = ((size_t)n * P + pp) * C + c;
      const float xv = x_f32 ? ((const float*)x)[idx] : (float)((const f16*)x)[idx];
      const float v = tile[tx][j] + xv;
      if (y_f32) { ((float*)y)[idx] = v; if (y16) y16[idx] = (f16)v; }
      else ((f16*)y)[idx] = (f16)v;
    }
  }
}

// Tiny pointwise conv on NCHW fp32 (C <= 8 in and out): VAE decoder's first 1x1 conv 4->4
// (sd/decoder.py:201) and the encoder's last 8->8 (sd/encoder.py:93).  in_scale multiplies the input.
__global__ __launch_bounds__(256) void conv1x1_nchw_small_kernel(const float* x, const float* w, const float* b, float* y,
                                                                 int B, int Cin, int Cout, size_t HW, float in_scale) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)B * HW; i += (size_t)gridDim.x * 256) {
    const size_t n = i / HW, px = i - n * HW;
    float v[8];
    for (int ci = 0; ci < Cin; ++ci) v[ci] = x[(n * Cin + ci) * HW + px] * in_scale;
    for (int co = 0; co < Cout; ++co) {
      float acc = b[co];
      for (int ci = 0; ci < Cin; ++ci) acc += w[co * Cin + ci] * v[ci];
      y[(n * Cout + co) * HW + px] = acc;
    }
  }
}

// CLIP embedding (sd/clip.py:34-63): out[row] = token_embedding[tokens[row]] + position_embedding[row % T]
__global__ __launch_bounds__(256) void clip_embed_kernel(const int64_t* tokens, const float* tok_emb, const float* pos_emb,
                                                         float* out, f16* out16, int rows, int T, int C, int vocab) {
  const size_t total = (size_t)rows * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / C), c = (int)(i - (size_t)row * C);
    long long tk = tokens[row];
    tk = tk < 0 ? 0 : (tk >= vocab ? vocab - 1 : tk);
    const float v = tok_emb[(size_t)tk * C + c] + pos_emb[(size_t)(row % T) * C + c];
    out[i] = v;
    out16[i] = (f16)v;
  }
}

// VAE encoder tail (sd/encoder.py:127-152): moments (B,8,h,w) NCHW -> mean, clamp(logvar,-30,20), z = mean +
// sqrt(exp(logvar)) * noise, z *= 0.18215.   out (B,4,h,w) NCHW fp32.
__global__ __launch_bounds__(256) void vae_sample_kernel(const float* mom, const float* noise, float* out, int B, size_t HW) {
  const size_t total = (size_t)B * 4 * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t n = i / (4 * HW), rem = i - n * 4 * HW;
    const float mean = mom[n * 8 * HW + rem];
    float lv = mom[n * 8 * HW + 4 * HW + rem];
    lv = fminf(fmaxf(lv, -30.f), 20.f);
    const float sd = sqrtf(expf(lv));
    out[i] = (mean + sd * noise[i]) * 0.18215f;
  }
}

inline int nblocks(size_t n, int per = 256, int cap = 4096) {
  size_t b = (n + per - 1) / per;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// LayerNorm folded into the following Linear (y = LN(x) W^T + b):  W' = gamma (.) W (fp16), g[n] = sum_c W'[n][c]
// (of the ROUNDED values, so that mean * g cancels exactly what the MFMA accumulates), h[n] = b[n] + sum_c beta[c] W[n][c].
__global__ __launch_bounds__(256) void ln_fold_prep_kernel(const void* w_src, int is_f32, const float* gamma, const float* beta,
                                                            const float* bias, f16* w_out, float* g_out, float* h_out, int C) {
  __shared__ float s_red[2][4];
  const int n = blockIdx.x, tid = threadIdx.x;
  float gs = 0.f, hs = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float w = is_f32 ? ((const float*)w_src)[(size_t)n * C + c] : (float)((const f16*)w_src)[(size_t)n * C + c];
    const f16 wf = (f16)(gamma[c] * w);
    w_out[(size_t)n * C + c] = wf;
    gs += (float)wf;
    hs += beta[c] * w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { gs += __shfl_xor(gs, o); hs += __shfl_xor(hs, o); }
  if ((tid & 63) == 0) { s_red[0][tid >> 6] = gs; s_red[1][tid >> 6] = hs; }
  __syncthreads();
  if (tid == 0) {
    g_out[n] = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    h_out[n] = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3] + (bias ? bias[n] : 0.f);
  }
}

// Weights of the phase-decomposed x2-upsample conv (GemmArgs::phase2): OIHW 3x3 -> [4 phases][O][ty][tx][I] fp16.
// Output parity py sees source rows {y+py-1, y+py}: for py = 0 tap row ty = 0 is kernel row 0 and ty = 1 is rows 1+2 (both
// land on source row y after the nearest upsample); for py = 1, ty = 0 is rows 0+1 and ty = 1 is row 2.  Same along x.
// The sums are taken in fp32 and rounded to fp16 once.
__global__ __launch_bounds__(256) void pack_ups_phase_kernel(const void* w, int w_f32, f16* out, int O, int I) {
  const size_t total = (size_t)16 * O * I;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int i = (int)(idx % I);
    size_t t = idx / I;
    const int tap = (int)(t & 3); t >>= 2;
    const int o = (int)(t % O);
    const int ph = (int)(t / O);
    const int py = ph >> 1, px = ph & 1, ty = tap >> 1, tx = tap & 1;
    const int y0 = py == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), y1 = py == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
    const int x0 = px == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), x1 = px == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
    float acc = 0.f;
    for (int dy = y0; dy <= y1; ++dy)
      for (int dx = x0; dx <= x1; ++dx) {
        const size_t src = (((size_t)o * I + i) * 3 + dy) * 3 + dx;
        acc += w_f32 ? ((const float*)w)[src] : (float)((const f16*)w)[src];
      }
    out[idx] = (f16)acc;
  }
}

// dst[c][r] = fp16(scale * src[r][c]): a Linear weight [R][Cc] stored transposed (contraction over its OUTPUT index; the
// folded cross-attention multiplies K by Wq from the left, engine.h xattn_fold)
__global__ __launch_bounds__(256) void transpose_scale_kernel(const void* src, int is_f32, f16* dst, int R, int Cc, float scale) {
  __shared__ float t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < Cc) v = is_f32 ? ((const float*)src)[(size_t)r * Cc + c] : (float)((const f16*)src)[(size_t)r * Cc + c];
    t[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < R) dst[(size_t)c * R + r] = (f16)(scale * t[tx][i]);
  }
}

// Per-head masked copies of the hoisted cross-attention K and V: row (b*H + h)*128 + j of dk / dv is row b*kv_rows + j of
// k / v with every column outside head h zeroed; keys j >= n_valid (and the padding rows up to 128) are zero rows.
__global__ __launch_bounds__(256) void xattn_mask_kernel(const f16* k, const f16* v, f16* dk, f16* dv, int H, int d, int kv_rows,
                                                         int n_valid) {
  const int row = blockIdx.x;                 // (b*H + h)*128 + j
  const int j = row & 127, bh = row >> 7, h = bh % H, b = bh / H;
  const int C = H * d;
  const bool live = j < n_valid;
  for (int c8 = threadIdx.x; c8 < C / 8; c8 += 256) {
    f16x8 zk, zv;
#pragma unroll
    for (int e = 0; e < 8; ++e) { zk[e] = (f16)0.f; zv[e] = (f16)0.f; }
    const int c = c8 * 8;                       // d is a multiple of 8: a chunk never straddles heads
    if (live && c / d == h) {
      zk = *(const f16x8*)(k + ((size_t)b * kv_rows + j) * C + c);
      zv = *(const f16x8*)(v + ((size_t)b * kv_rows + j) * C + c);
    }
    *(f16x8*)(dk + (size_t)row * C + c) = zk;
    *(f16x8*)(dv + (size_t)row * C + c) = zv;
  }
}

// Two consecutive linear maps composed at load time (conv_output o linear_geglu_2, sd/diffusion.py:363,381):
//   out[n][j] = sum_c A[n][c] * B[c][j]   A: [N][K], B: [K][J], fp32 accumulate, fp16 out with row stride ldo.
__global__ __launch_bounds__(256) void compose_linear_kernel(const void* A, int a_f32, const void* B, int b_f32, void* out,
                                                              int out_f32, int N, int K, int J, int ldo) {
  __shared__ float sa[16][17], sb[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int n = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const int ka = k0 + tx, kb = k0 + ty;
    float va = 0.f, vb = 0.f;
    if (n < N && ka < K) va = a_f32 ? ((const float*)A)[(size_t)n * K + ka] : (float)((const f16*)A)[(size_t)n * K + ka];
    if (kb < K && j < J) vb = b_f32 ? ((const float*)B)[(size_t)kb * J + j] : (float)((const f16*)B)[(size_t)kb * J + j];
    sa[ty][tx] = va;
    sb[ty][tx] = vb;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += sa[ty][k] * sb[k][tx];
    __syncthreads();
  }
  if (n < N && j < J) {
    if (out_f32) ((float*)out)[(size_t)n * ldo + j] = acc;
    else ((f16*)out)[(size_t)n * ldo + j] = (f16)acc;
  }
}
// bias'[n] = sum_c A[n][c] * b_in[c] + b_out[n]
__global__ __launch_bounds__(64) void compose_bias_kernel(const void* A, int a_f32, const float* b_in, const float* b_out,
                                                           float* out, int K) {
  const int n = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int c = lane; c < K; c += 64)
    s += (a_f32 ? ((const float*)A)[(size_t)n * K + c] : (float)((const f16*)A)[(size_t)n * K + c]) * b_in[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[n] = s + b_out[n];
}
// dst[r][0..cols) (row stride ld) = fp16(src[r][0..cols))
__global__ __launch_bounds__(256) void cast_rows_kernel(const void* src, int is_f32, f16* dst, int rows, int cols, int ld) {
  const size_t total = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[(size_t)r * ld + c] = is_f32 ? (f16)((const float*)src)[i] : ((const f16*)src)[i];
  }
}

}  // namespace

int sdmi_launch_cast_f32_f16(const float* x, f16* y, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(nblocks(n)), dim3(256), 0, st, x, y, n);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_cast_any_f32(const void* x, int in_f32, float* y, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(cast_any_f32_kernel, dim3(nblocks(n)), dim3(256), 0, st, x, in_f32, y, n);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_pack_conv(const void* w, int w_f32, f16* out, int O, int I, int ks, int o_keep, hipStream_t st) {
  const size_t total = (size_t)o_keep * ks * ks * I;
  hipLaunchKernelGGL(pack_conv_kernel, dim3(nblocks(total)), dim3(256), 0, st, w, w_f32, out, O, I, ks, o_keep);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_pack_stem(const void* w, int w_f32, float* w36, int Cout, int Cin, hipStream_t st) {
  hipLaunchKernelGGL(pack_stem_kernel, dim3(nblocks(9 * (size_t)Cin * Cout)), dim3(256), 0, st, w, w_f32, w36, Cout, Cin);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_small_linear(const float* x, const f16* w, const float* b, float* y, int M, int N, int K,
                             int silu, int ldy, hipStream_t st) {
  SDMI_REQUIRE(K % 8 == 0 && K <= 8 * 64 * 3, "small_linear: K=%d unsupported", K);
  const int nch = (K / 8 + 63) / 64;
  dim3 grid((N + 3) / 4), block(256);
  if (nch == 1) hipLaunchKernelGGL(small_linear_kernel<1>, grid, block, 0, st, x, w, b, y, M, N, K, silu, ldy);
  else if (nch == 2) hipLaunchKernelGGL(small_linear_kernel<2>, grid, block, 0, st, x, w, b, y, M, N, K, silu, ldy);
  else hipLaunchKernelGGL(small_linear_kernel<3>, grid, block, 0, st, x, w, b, y, M, N, K, silu, ldy);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_stem_conv(const float* lat, int lat_batch, const float* w36, const float* bias, void* out,
                          int out_f32, f16* out16, int B, int H, int W, int Cout, int Cin, hipStream_t st, float* gn_rec, int gn_rec_T) {
  SDMI_REQUIRE(Cout % 8 == 0 && Cin >= 1 && Cin <= 4, "stem conv: Cout=%d Cin=%d", Cout, Cin);
  const size_t npix = (size_t)B * H * W;
  SDMI_REQUIRE(npix * Cout < ((size_t)1 << 31), "stem conv: too many outputs");
  const int C8 = Cout / 8;
  const size_t groups = (npix + 63) / 64;
  // chunks of 8 channels per wave: as many as keeps >= ~2048 waves in the launch (the neighbourhood loads are per wave)
  int cpw = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)C8, STEM_CPW_MAX), (size_t)C8 * groups / 2048));
  if (gn_rec) {
    // GroupNorm statistics records (one per 64-pixel row block and 10-channel atom, parts = 1): a wave owns 40 channels
    SDMI_REQUIRE(Cin == 4 && Cout % 40 == 0 && (H * W) % 64 == 0 && gn_rec_T == H * W / 64 && out_f32,
                 "stem conv: statistics records need Cin = 4, Cout %% 40 == 0, H*W %% 64 == 0, T = H*W/64 (Cout=%d H=%d W=%d T=%d)", Cout, H, W, gn_rec_T);
    cpw = 5;
  }
  const int nchunk = (C8 + cpw - 1) / cpw;
  const size_t waves = groups * nchunk;
  const dim3 grid((unsigned)((waves + 3) / 4));
#define SDMI_STEM(CI, GA) hipLaunchKernelGGL((stem_conv_kernel<CI, GA>), grid, dim3(256), 0, st, lat, lat_batch, w36, bias, out, out_f32, out16, B, H, W, Cout, cpw, nchunk, gn_rec, gn_rec_T)
  if (gn_rec) SDMI_STEM(4, true);
  else switch (Cin) {
    case 1: SDMI_STEM(1, false); break;
    case 2: SDMI_STEM(2, false); break;
    case 3: SDMI_STEM(3, false); break;
    default: SDMI_STEM(4, false); break;
  }
#undef SDMI_STEM
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_final_conv(const f16* x, const f16* w, const float* bias, float* out, int B, int H, int W, int Cin,
                           int Cout, hipStream_t st, const float* x32) {
  SDMI_REQUIRE(Cin % 8 == 0 && Cout >= 1 && Cout <= 8 && W % FC_PX == 0, "final conv: Cin=%d Cout=%d W=%d (Cout <= 8, W a multiple of %d)", Cin, Cout, W, FC_PX);
  const size_t npix = (size_t)B * H * W;
  SDMI_REQUIRE(npix < ((size_t)1 << 31), "final conv: too many pixels");
  const size_t ngrp = npix / FC_PX;
  const float* const nof = nullptr;
  if (x32) {
    SDMI_REQUIRE(Cout <= 4, "final conv: the fp32-input form holds 4 output channels");
    hipLaunchKernelGGL((final_conv_kernel<4, true>), dim3((unsigned)((ngrp + 3) / 4)), dim3(256), 0, st, x, w, bias, out, B, H, W, Cin, Cout, x32);
  } else if (Cout <= 4) hipLaunchKernelGGL((final_conv_kernel<4, false>), dim3((unsigned)((ngrp + 3) / 4)), dim3(256), 0, st, x, w, bias, out, B, H, W, Cin, Cout, nof);
  else hipLaunchKernelGGL((final_conv_kernel<8, false>), dim3((unsigned)((ngrp + 3) / 4)), dim3(256), 0, st, x, w, bias, out, B, H, W, Cin, Cout, nof);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_cfg_ddpm(const float* eps, int do_cfg, float cfg_scale, float* latents, const float* noise,
                         const float* coef, size_t n, float* eps_out, hipStream_t st) {
  DdpmCoef k;
  k.sqrt_beta_prod = coef[0];
  k.sqrt_alpha_prod = coef[1];
  k.c0 = coef[2];
  k.ct = coef[3];
  k.sigma = coef[4];
  k.has_noise = noise != nullptr;
  hipLaunchKernelGGL(cfg_ddpm_kernel, dim3(nblocks(n, 256, 1024)), dim3(256), 0, st, eps, do_cfg, cfg_scale, latents,
                     noise, k, n, eps_out);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_final_conv_step(const f16* x, const f16* w, const float* bias, int P, int H, int W, int Cin, int do_cfg, float cfg_scale,
                                float* latents, const float* noise, const float* coef, hipStream_t st) {
  SDMI_REQUIRE(Cin % 8 == 0 && W % FC_PX == 0 && P >= 1, "final conv + step: Cin=%d W=%d P=%d", Cin, W, P);
  const size_t npix = (size_t)P * H * W;
  SDMI_REQUIRE(npix * 2 < ((size_t)1 << 31), "final conv + step: too many pixels");
  DdpmCoef k;
  k.sqrt_beta_prod = coef[0];
  k.sqrt_alpha_prod = coef[1];
  k.c0 = coef[2];
  k.ct = coef[3];
  k.sigma = coef[4];
  k.has_noise = noise != nullptr;
  const size_t ngrp = npix / FC_PX;
  hipLaunchKernelGGL(final_conv_step_kernel, dim3((unsigned)((ngrp + 3) / 4)), dim3(256), 0, st, x, w, bias, P, H, W, Cin, do_cfg, cfg_scale,
                     latents, noise, k);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_add_vec(const float* a, const float* b, float* y, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(add_vec_kernel, dim3(nblocks(n)), dim3(256), 0, st, a, b, y, n);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_row_softmax(const f16* s, f16* p, int rows, int L, float scale, hipStream_t st) {
  SDMI_REQUIRE(L % 8 == 0 && rows > 0, "row_softmax: rows=%d L=%d", rows, L);
  hipLaunchKernelGGL(row_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, s, p, rows, L, scale);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_q4_reinterpret_add(const float* o, const void* x, int x_f32, void* y, int y_f32, f16* y16, int B, int P,
                                   int C, hipStream_t st) {
  dim3 grid((P + 31) / 32, (C + 31) / 32, B);
  hipLaunchKernelGGL(q4_reinterpret_add_kernel, grid, dim3(256), 0, st, o, x, x_f32, y, y_f32, y16, B, P, C);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_conv1x1_nchw_small(const float* x, const float* w, const float* b, float* y, int B, int Cin, int Cout,
                                   size_t HW, float in_scale, hipStream_t st) {
  SDMI_REQUIRE(Cin <= 8 && Cout <= 8, "conv1x1_nchw_small: Cin=%d Cout=%d", Cin, Cout);
  hipLaunchKernelGGL(conv1x1_nchw_small_kernel, dim3(nblocks((size_t)B * HW)), dim3(256), 0, st, x, w, b, y, B, Cin, Cout,
                     HW, in_scale);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_clip_embed(const int64_t* tokens, const float* tok_emb, const float* pos_emb, float* out, f16* out16,
                           int rows, int T, int C, int vocab, hipStream_t st) {
  hipLaunchKernelGGL(clip_embed_kernel, dim3(nblocks((size_t)rows * C)), dim3(256), 0, st, tokens, tok_emb, pos_emb, out,
                     out16, rows, T, C, vocab);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_vae_sample(const float* mom, const float* noise, float* out, int B, size_t HW, hipStream_t st) {
  hipLaunchKernelGGL(vae_sample_kernel, dim3(nblocks((size_t)B * 4 * HW)), dim3(256), 0, st, mom, noise, out, B, HW);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_ln_fold_prep(const void* w_src, int is_f32, const float* gamma, const float* beta, const float* bias,
                            f16* w_out, float* g_out, float* h_out, int N, int C, hipStream_t st) {
  SDMI_REQUIRE(w_src && gamma && beta && w_out && g_out && h_out && N > 0 && C > 0, "ln_fold_prep: bad arguments");
  hipLaunchKernelGGL(ln_fold_prep_kernel, dim3(N), dim3(256), 0, st, w_src, is_f32, gamma, beta, bias, w_out, g_out, h_out, C);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_compose_linear(const void* A, int a_f32, const void* B, int b_f32, void* out, int out_f32, int N, int K, int J,
                               int ldo, hipStream_t st) {
  SDMI_REQUIRE(A && B && out && N > 0 && K > 0 && J > 0 && ldo >= J, "compose_linear: bad arguments");
  hipLaunchKernelGGL(compose_linear_kernel, dim3((J + 15) / 16, (N + 15) / 16), dim3(256), 0, st, A, a_f32, B, b_f32, out, out_f32, N, K, J, ldo);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}
int sdmi_launch_compose_bias(const void* A, int a_f32, const float* b_in, const float* b_out, float* out, int N, int K,
                             hipStream_t st) {
  SDMI_REQUIRE(A && b_in && b_out && out && N > 0 && K > 0, "compose_bias: bad arguments");
  hipLaunchKernelGGL(compose_bias_kernel, dim3(N), dim3(64), 0, st, A, a_f32, b_in, b_out, out, K);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}
int sdmi_launch_cast_rows(const void* src, int is_f32, f16* dst, int rows, int cols, int ld, hipStream_t st) {
  SDMI_REQUIRE(src && dst && rows > 0 && cols > 0 && ld >= cols, "cast_rows: bad arguments");
  size_t total = (size_t)rows * cols;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cast_rows_kernel, dim3(blocks), dim3(256), 0, st, src, is_f32, dst, rows, cols, ld);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

int sdmi_launch_transpose_scale(const void* src, int is_f32, f16* dst, int R, int Cc, float scale, hipStream_t st) {
  SDMI_REQUIRE(src && dst && R > 0 && Cc > 0, "transpose_scale: bad arguments");
  hipLaunchKernelGGL(transpose_scale_kernel, dim3((Cc + 31) / 32, (R + 31) / 32), dim3(256), 0, st, src, is_f32, dst, R, Cc, scale);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}
int sdmi_launch_xattn_mask(const f16* k, const f16* v, f16* dk, f16* dv, int B, int H, int d, int kv_rows, int n_valid,
                           hipStream_t st) {
  SDMI_REQUIRE(k && v && dk && dv && B > 0 && H > 0 && d % 8 == 0 && n_valid > 0 && n_valid <= 128 && n_valid <= kv_rows,
               "xattn_mask: bad arguments");
  hipLaunchKernelGGL(xattn_mask_kernel, dim3(B * H * 128), dim3(256), 0, st, k, v, dk, dv, H, d, kv_rows, n_valid);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}
int sdmi_launch_pack_ups_phase(const void* w, int w_f32, f16* out, int O, int I, hipStream_t st) {
  SDMI_REQUIRE(w && out && O > 0 && I > 0, "pack_ups_phase: bad arguments");
  size_t total = (size_t)16 * O * I;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_ups_phase_kernel, dim3(blocks), dim3(256), 0, st, w, w_f32, out, O, I);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}
