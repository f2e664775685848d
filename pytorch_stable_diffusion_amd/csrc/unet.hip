// Host side of libsdmi: weight packing, static UNet schedule, per-shape GEMM autotune, C ABI.
//
// The reference runs ~391 eager torch ops per UNet call through nn.Module dispatch
// (sd/diffusion.py:458-496, 628-676).  Here the network is a static launch schedule over NHWC fp16
// activations (fp32 accumulation, optional fp32 residual stream): norms, implicit-GEMM convs/linears
// with fused bias/time-vector/residual epilogues, flash attention.  Cross-attention K/V are hoisted
// per prompt (set_context) and all timestep vectors per schedule (set_schedule).
#include "engine.h"

using namespace sdmi;

// ---------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void sdmi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct sdmi_unet : Engine {};

// =============================================================================================
namespace {

struct StageOp { int kind; int a, b, c; };   // kind: 0 conv(cin,cout,stride) 1 res(cin,cout) 2 attn(heads,dh) 3 up(c)
typedef std::vector<StageOp> Stage;

const std::vector<Stage>& encoders() {
  static const std::vector<Stage> v = {
      {{0, 4, 320, 1}},
      {{1, 320, 320, 0}, {2, 8, 40, 0}},
      {{1, 320, 320, 0}, {2, 8, 40, 0}},
      {{0, 320, 320, 2}},
      {{1, 320, 640, 0}, {2, 8, 80, 0}},
      {{1, 640, 640, 0}, {2, 8, 80, 0}},
      {{0, 640, 640, 2}},
      {{1, 640, 1280, 0}, {2, 8, 160, 0}},
      {{1, 1280, 1280, 0}, {2, 8, 160, 0}},
      {{0, 1280, 1280, 2}},
      {{1, 1280, 1280, 0}},
      {{1, 1280, 1280, 0}},
  };
  return v;
}
const Stage& bottleneck() {
  static const Stage v = {{1, 1280, 1280, 0}, {2, 8, 160, 0}, {1, 1280, 1280, 0}};
  return v;
}
const std::vector<Stage>& decoders() {
  static const std::vector<Stage> v = {
      {{1, 2560, 1280, 0}},
      {{1, 2560, 1280, 0}},
      {{1, 2560, 1280, 0}, {3, 1280, 0, 0}},
      {{1, 2560, 1280, 0}, {2, 8, 160, 0}},
      {{1, 2560, 1280, 0}, {2, 8, 160, 0}},
      {{1, 1920, 1280, 0}, {2, 8, 160, 0}, {3, 1280, 0, 0}},
      {{1, 1920, 640, 0}, {2, 8, 80, 0}},
      {{1, 1280, 640, 0}, {2, 8, 80, 0}},
      {{1, 960, 640, 0}, {2, 8, 80, 0}, {3, 640, 0, 0}},
      {{1, 960, 320, 0}, {2, 8, 40, 0}},
      {{1, 640, 320, 0}, {2, 8, 40, 0}},
      {{1, 640, 320, 0}, {2, 8, 40, 0}},
  };
  return v;
}

std::string key(const char* group, int i, int j) {
  char buf[96];
  if (i >= 0) snprintf(buf, sizeof(buf), "unet.%s.%d.%d", group, i, j);
  else snprintf(buf, sizeof(buf), "unet.%s.%d", group, j);
  return buf;
}

int load_op(sdmi_unet* u, const std::string& p, const StageOp& op, bool lenient) {
  // in partial mode a block whose first tensor is absent is skipped entirely
  auto present = [&](const char* leaf) { return u->has(p + leaf); };
  switch (op.kind) {
    case 0:
      if (lenient && !present(".weight")) return SDMI_OK;
      if (op.a == 4) {
        const sdmi_tensor_desc* t;
        TRY(u->need(p + ".weight", &t, 4, {op.b, 4, 3, 3}));
        TRY(u->dmalloc(&u->stem_w36, (size_t)36 * op.b * 4));
        TRY(sdmi_launch_pack_stem(t->data_dev, t->dtype == SDMI_F32, u->stem_w36, op.b, 4, u->st));
        TRY(u->load_vec(p + ".bias", op.b, &u->stem_bias));
        u->stem_cout = op.b;
        u->has_stem = true;
        u->weight_bytes += 36 * op.b * 4;
        return SDMI_OK;
      }
      return u->load_plain_conv(p, op.a, op.b);
    case 1:
      if (lenient && !present(".groupnorm_feature.weight")) return SDMI_OK;
      return u->load_res(p, op.a, op.b);
    case 2:
      if (lenient && !present(".groupnorm.weight")) return SDMI_OK;
      return u->load_attn(p, op.a, op.b);
    case 3:
      if (lenient && !present(".conv.weight")) return SDMI_OK;
      return u->load_ups_conv(p + ".conv", op.a);
  }
  return SDMI_EINVAL;
}

int run_stage(sdmi_unet* u, const std::string& grp_prefix, const Stage& stage, Act x, const Act* skip,
              const float* tv, Act* out) {
  Act cur = x;
  bool first = true;
  for (size_t j = 0; j < stage.size(); ++j) {
    const StageOp& op = stage[j];
    const std::string p = grp_prefix + "." + std::to_string(j);
    Act y;
    switch (op.kind) {
      case 0: {
        auto it = u->convs.find(p);
        if (it == u->convs.end()) { sdmi_set_error("conv '%s' not loaded", p.c_str()); return SDMI_ENOENT; }
        TRY(u->conv3(it->second, cur, op.c, 0, &y));
        break;
      }
      case 1: {
        auto it = u->res.find(p);
        if (it == u->res.end()) { sdmi_set_error("block '%s' not loaded", p.c_str()); return SDMI_ENOENT; }
        // an attention block next: its GroupNorm is the only launch between conv_merged and everything else that reads y
        const bool to_gn = j + 1 < stage.size() && stage[j + 1].kind == 2;
        TRY(u->res_block(it->second, cur, (first ? skip : nullptr), tv + it->second.time_off, &y, to_gn));
        break;
      }
      case 2: {
        auto it = u->attn.find(p);
        if (it == u->attn.end()) { sdmi_set_error("block '%s' not loaded", p.c_str()); return SDMI_ENOENT; }
        TRY(u->attn_block(it->second, cur, &y));
        break;
      }
      case 3: {
        auto it = u->convs.find(p + ".conv");
        if (it == u->convs.end()) { sdmi_set_error("conv '%s.conv' not loaded", p.c_str()); return SDMI_ENOENT; }
        TRY(u->conv3(it->second, cur, 1, 1, &y));
        break;
      }
    }
    cur = y;
    first = false;
  }
  TRY(u->flush_pending());
  *out = cur;
  return SDMI_OK;
}

// the sampler step fused behind the output conv (sdmi_unet_denoise_step_batch): latents (P,4,h,w) updated in place
struct StepFuse { float* latents; const float* noise; const float* coef; int n_prompts, do_cfg; float cfg_scale; };

int final_layer(sdmi_unet* u, const Act& x, float* eps_out, const StepFuse* sf = nullptr) {
  if (!u->has_final) { sdmi_set_error("final layer not loaded"); return SDMI_ENOENT; }
  Act t;
  TRY(u->groupnorm(x, nullptr, u->final_gn, 1e-5f, 1, &t));
  if (sf) {
    TRY(sdmi_launch_final_conv_step(t.h, u->final_conv.w, u->final_conv.bias, sf->n_prompts, x.H, x.W, x.C, sf->do_cfg, sf->cfg_scale,
                                    sf->latents, sf->noise, sf->coef, u->st));
    u->launches += 1;
    u->log_launch("final_conv");
    return SDMI_OK;
  }
  TRY(sdmi_launch_final_conv(t.h, u->final_conv.w, u->final_conv.bias, eps_out, x.B, x.H, x.W, x.C, 4, u->st, u->accurate ? t.f : nullptr));
  u->launches += 1;
  u->log_launch("final_conv");
  return SDMI_OK;
}

// NHWC fp32 device tensor -> Act (fp16 + optional fp32 stream copy)
int import_act(sdmi_unet* u, const float* x, int B, int H, int W, int C, Act* a) {
  TRY(u->new_act(B, H, W, C, true, a));
  const size_t n = (size_t)B * H * W * C;
  TRY(sdmi_launch_cast_f32_f16(x, a->h, n, u->st));
  if (a->f) SDMI_CHECK_HIP(hipMemcpyAsync(a->f, x, n * 4, hipMemcpyDeviceToDevice, u->st));
  return SDMI_OK;
}
int export_act(sdmi_unet* u, const Act& a, float* out) {
  const size_t n = (size_t)a.M() * a.C;
  if (a.f) SDMI_CHECK_HIP(hipMemcpyAsync(out, a.f, n * 4, hipMemcpyDeviceToDevice, u->st));
  else TRY(sdmi_launch_cast_any_f32(a.h, 0, out, n, u->st));
  return SDMI_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

const char* sdmi_last_error(void) { return g_err; }
int sdmi_version(void) { return 200; }

int sdmi_unet_create(const sdmi_tensor_desc* tensors, int n_tensors, int flags, sdmi_unet** out) {
  if (!tensors || !out || n_tensors <= 0) { sdmi_set_error("sdmi_unet_create: bad arguments"); return SDMI_EINVAL; }
  sdmi_unet* u = new sdmi_unet();
  u->flags = flags;
  u->stream_f32 = (flags & SDMI_FLAG_STREAM_F32) != 0;
  u->partial = (flags & SDMI_FLAG_PARTIAL) != 0;
  u->tune = (flags & SDMI_FLAG_NO_TUNE) == 0;
  u->accurate = (flags & SDMI_FLAG_ACCURATE) != 0;
  if (u->accurate) { u->ln_fold_on = false; u->tune = false; }      // (engine.h Engine::accurate)
  for (int i = 0; i < n_tensors; ++i) {
    if (!tensors[i].name || !tensors[i].data_dev) { delete u; sdmi_set_error("tensor %d: null name/data", i); return SDMI_EINVAL; }
    u->src[tensors[i].name] = tensors[i];
  }
  auto fail = [&](int rc) { delete u; return rc; };
  int rc;
  if ((rc = u->dmalloc(&u->zero, 4096)) != SDMI_OK) return fail(rc);
  if (hipMemset(u->zero, 0, 4096) != hipSuccess) return fail(SDMI_EHIP);
  if (hipMemsetD16((hipDeviceptr_t)(u->zero + 1024), 0x3C00, 1024) != hipSuccess) return fail(SDMI_EHIP);   // fp16 ones (Engine::ones)
  const bool len = u->partial;
  // time embedding
  if (!len || u->has("time_embedding.linear_1.weight")) {
    if ((rc = u->load_conv("time_embedding.linear_1", kTime, 320, 1, true, &u->te1)) != SDMI_OK) return fail(rc);
    if ((rc = u->load_conv("time_embedding.linear_2", kTime, kTime, 1, true, &u->te2)) != SDMI_OK) return fail(rc);
    u->has_time = true;
  }
  for (size_t i = 0; i < encoders().size(); ++i)
    for (size_t j = 0; j < encoders()[i].size(); ++j)
      if ((rc = load_op(u, key("encoders", (int)i, (int)j), encoders()[i][j], len)) != SDMI_OK) return fail(rc);
  for (size_t j = 0; j < bottleneck().size(); ++j)
    if ((rc = load_op(u, key("bottleneck", -1, (int)j), bottleneck()[j], len)) != SDMI_OK) return fail(rc);
  for (size_t i = 0; i < decoders().size(); ++i)
    for (size_t j = 0; j < decoders()[i].size(); ++j)
      if ((rc = load_op(u, key("decoders", (int)i, (int)j), decoders()[i][j], len)) != SDMI_OK) return fail(rc);
  if (!len || u->has("final.groupnorm.weight")) {
    if ((rc = u->load_norm("final.groupnorm", 320, &u->final_gn)) != SDMI_OK) return fail(rc);
    if ((rc = u->load_conv("final.conv", 4, 320, 3, true, &u->final_conv)) != SDMI_OK) return fail(rc);
    u->has_final = true;
  }
  // scratch
  u->max_steps = 1000;
  if ((rc = u->dmalloc(&u->time_scratch, (size_t)2 * u->max_steps * kTime * 4)) != SDMI_OK) return fail(rc);
  if ((rc = u->dmalloc(&u->timevec, (size_t)u->max_steps * (u->time_total ? u->time_total : 1) * 4)) != SDMI_OK) return fail(rc);
  if ((rc = u->dmalloc(&u->timevec_adhoc, (size_t)(u->time_total ? u->time_total : 1) * 4)) != SDMI_OK) return fail(rc);
  u->slab_bytes = (size_t)96 << 20;
  if ((rc = u->dmalloc(&u->slab, u->slab_bytes)) != SDMI_OK) return fail(rc);
  if ((rc = u->dmalloc(&u->gn_partial, (size_t)16 * 128 * 32 * 2 * 4)) != SDMI_OK) return fail(rc);
  u->gacc_enabled = true;                   // GroupNorm statistics from the producers' epilogues (engine.h attach_gacc)
  if ((rc = u->dmalloc(&u->ln_guard, 256)) != SDMI_OK) return fail(rc);
  if (hipMemset(u->ln_guard, 0, 256) != hipSuccess) return fail(SDMI_EHIP);
  // activation arena: one bump allocation per forward (deterministic addresses); 6 GiB carry UNet batch <= 8 at 64x64 latents
  // (four prompts of the batched throughput mode) and batch 2 at 96x96; SDMI_ARENA_GB sizes it for more (batch 16: 24)
  // (SDMI_ARENA_GB, read per handle: a minimum; a forward that needs more grows it -- Engine::ensure_arena, arena_need below)
  const size_t arena_gb = getenv("SDMI_ARENA_GB") ? (size_t)atoi(getenv("SDMI_ARENA_GB")) : 6;
  u->arena.cap = u->partial ? ((size_t)1 << 30) : ((arena_gb ? arena_gb : 6) << 30);
  if ((rc = u->dmalloc(&u->arena.base, u->arena.cap)) != SDMI_OK) return fail(rc);
  if (hipDeviceSynchronize() != hipSuccess) { sdmi_set_error("weight packing failed: %s", hipGetErrorString(hipGetLastError())); return fail(SDMI_EHIP); }
  u->src.clear();   // caller's tensors are no longer referenced
  *out = u;
  return SDMI_OK;
}

// A second LANE over the same packed weights: its own activation arena, split-K slabs, context buffers, time vectors and
// plans, so two denoising loops (two prompts) can run concurrently on two streams of one GPU and fill each other's
// per-launch latency (a single chain leaves ~20 % of the chip idle: profiles/r02_bench_2rank_rehearsal.json).  The clone
// borrows every weight pointer of `src`; `src` must outlive it.
int sdmi_unet_clone(const sdmi_unet* src, sdmi_unet** out) {
  if (!src || !out) { sdmi_set_error("sdmi_unet_clone: null argument"); return SDMI_EINVAL; }
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != src->device) {
    sdmi_set_error("sdmi_unet_clone: the source handle lives on HIP device %d but device %d is current", src->device, cur);
    return SDMI_EINVAL;
  }
  sdmi_unet* u = new sdmi_unet();
  u->flags = src->flags; u->stream_f32 = src->stream_f32; u->partial = src->partial; u->tune = src->tune;
  u->is_lane = true;
  u->accurate = src->accurate;
  u->weight_bytes = 0;                                   // borrowed
  u->res = src->res; u->attn = src->attn; u->convs = src->convs;
  u->te1 = src->te1; u->te2 = src->te2; u->has_time = src->has_time; u->time_total = src->time_total;
  u->res_order = src->res_order; u->attn_order = src->attn_order;
  u->stem_w36 = src->stem_w36; u->stem_bias = src->stem_bias; u->stem_cout = src->stem_cout; u->has_stem = src->has_stem;
  u->final_gn = src->final_gn; u->final_conv = src->final_conv; u->has_final = src->has_final;
  u->plans = src->plans;
  for (auto& kv : u->plans) { kv.second.calls = 0; }
  auto fail = [&](int rc) { delete u; return rc; };
  int rc;
  if ((rc = u->dmalloc(&u->zero, 4096)) != SDMI_OK) return fail(rc);
  if (hipMemset(u->zero, 0, 4096) != hipSuccess) return fail(SDMI_EHIP);
  if (hipMemsetD16((hipDeviceptr_t)(u->zero + 1024), 0x3C00, 1024) != hipSuccess) return fail(SDMI_EHIP);
  u->max_steps = src->max_steps;
  if ((rc = u->dmalloc(&u->time_scratch, (size_t)2 * u->max_steps * kTime * 4)) != SDMI_OK) return fail(rc);
  if ((rc = u->dmalloc(&u->timevec, (size_t)u->max_steps * (u->time_total ? u->time_total : 1) * 4)) != SDMI_OK) return fail(rc);
  if ((rc = u->dmalloc(&u->timevec_adhoc, (size_t)(u->time_total ? u->time_total : 1) * 4)) != SDMI_OK) return fail(rc);
  u->slab_bytes = src->slab_bytes;
  if ((rc = u->dmalloc(&u->slab, u->slab_bytes)) != SDMI_OK) return fail(rc);
  if ((rc = u->dmalloc(&u->gn_partial, (size_t)16 * 128 * 32 * 2 * 4)) != SDMI_OK) return fail(rc);
  u->gacc_enabled = src->gacc_enabled;
  u->ln_fold_on = src->ln_fold_on;
  if ((rc = u->dmalloc(&u->ln_guard, 256)) != SDMI_OK) return fail(rc);
  if (hipMemset(u->ln_guard, 0, 256) != hipSuccess) return fail(SDMI_EHIP);
  u->arena.cap = src->arena.cap;
  if ((rc = u->dmalloc(&u->arena.base, u->arena.cap)) != SDMI_OK) return fail(rc);
  if (hipDeviceSynchronize() != hipSuccess) return fail(SDMI_EHIP);
  *out = u;
  return SDMI_OK;
}

void sdmi_unet_destroy(sdmi_unet* u) {
  if (u) { (void)hipDeviceSynchronize(); u->plan_report("unet"); delete u; }
}

int sdmi_unet_set_context(sdmi_unet* u, const float* ctx_dev, int batch, int n_tokens, void* stream) {
  if (!u || !ctx_dev) { sdmi_set_error("set_context: null argument"); return SDMI_EINVAL; }
  SDMI_REQUIRE(batch >= 1 && batch <= 16 && n_tokens >= 1 && n_tokens <= kCtxPad, "set_context: batch=%d tokens=%d unsupported (tokens <= %d)", batch, n_tokens, kCtxPad);
  TRY(u->enter(stream));
  if (u->ctx_batch != batch || !u->ctx16) {
    // per-batch buffers: the previous set is released (hipFree waits for work in flight), and ctx_batch stays 0 until the
    // new set is complete, so a failure half way cannot be mistaken for a usable context by the next call
    u->ctx_batch = 0;
    u->free_ctx();
    TRY(u->dmalloc_ctx(&u->ctx16, (size_t)batch * kCtxPad * kCtx * 2));
    if (u->accurate) TRY(u->dmalloc_ctx(&u->ctx32, (size_t)batch * kCtxPad * kCtx * 4));
    int cmax = 0;
    for (const std::string& p : u->attn_order) {
      const AttnW& w = u->attn[p];
      f16 *k, *v;
      TRY(u->dmalloc_ctx(&k, (size_t)batch * kCtxPad * w.C * 2));
      TRY(u->dmalloc_ctx(&v, (size_t)batch * w.C * kCtxVtLd * 2));
      SDMI_CHECK_HIP(hipMemsetAsync(v, 0, (size_t)batch * w.C * kCtxVtLd * 2, u->st));
      u->ctxK.push_back(k);
      u->ctxVt.push_back(v);
      f16 *w1 = nullptr, *w2 = nullptr;
      float *g = nullptr, *h = nullptr;
      if (w.wqT && !u->accurate) {
        const size_t rows = (size_t)batch * Engine::kXfCols;
        TRY(u->dmalloc_ctx(&w1, rows * w.C * 2));
        TRY(u->dmalloc_ctx(&w2, rows * w.C * 2));
        TRY(u->dmalloc_ctx(&g, rows * 4));
        TRY(u->dmalloc_ctx(&h, rows * 4));
        cmax = w.C > cmax ? w.C : cmax;
      }
      u->xfW1.push_back(w1); u->xfW2.push_back(w2); u->xfG.push_back(g); u->xfH.push_back(h);
    }
    if (cmax) {
      const size_t rows = (size_t)batch * Engine::kXfCols;
      TRY(u->dmalloc_ctx(&u->xf_km, rows * cmax * 2));
      TRY(u->dmalloc_ctx(&u->xf_vm, rows * cmax * 2));
      TRY(u->dmalloc_ctx(&u->xf_vp, (size_t)batch * kCtxPad * cmax * 2));
      TRY(u->dmalloc_ctx(&u->xf_kq, rows * cmax * 4));
    }
  }
  SDMI_CHECK_HIP(hipMemsetAsync(u->ctx16, 0, (size_t)batch * kCtxPad * kCtx * 2, u->st));
  for (int b = 0; b < batch; ++b)
    TRY(sdmi_launch_cast_f32_f16(ctx_dev + (size_t)b * n_tokens * kCtx, u->ctx16 + (size_t)b * kCtxPad * kCtx, (size_t)n_tokens * kCtx, u->st));
  u->ctx_batch = batch;
  u->ctx_tokens = n_tokens;
  Act c;
  c.h = u->ctx16; c.B = batch; c.H = kCtxPad; c.W = 1; c.C = kCtx;
  if (u->accurate) {       // the k / v projections read the context itself, zero-padded to 80 rows like its fp16 copy
    SDMI_CHECK_HIP(hipMemsetAsync(u->ctx32, 0, (size_t)batch * kCtxPad * kCtx * 4, u->st));
    SDMI_CHECK_HIP(hipMemcpy2DAsync(u->ctx32, (size_t)kCtxPad * kCtx * 4, ctx_dev, (size_t)n_tokens * kCtx * 4, (size_t)n_tokens * kCtx * 4, batch,
                                    hipMemcpyDeviceToDevice, u->st));
    c.f = u->ctx32;
  }
  for (size_t i = 0; i < u->attn_order.size(); ++i) {
    const AttnW& w = u->attn[u->attn_order[i]];
    { GemmArgs a = Engine::base_args(c, nullptr, w.k, kCtxPad, 1, 1, 0); a.out = u->ctxK[i]; a.ldc = w.C; TRY(u->gemm(a)); }
    {
      GemmArgs a = Engine::base_args(c, nullptr, w.v, kCtxPad, 1, 1, 0);
      a.out = u->ctxK[i];   // unused (all columns go to the transposed tail)
      a.ldc = w.C;
      a.outT = u->ctxVt[i]; a.nt0 = 0; a.S = kCtxPad; a.ldt = kCtxVtLd; a.tperm = 1;
      TRY(u->gemm(a));
    }
    if (u->xfW1[i]) TRY(u->xattn_fold((int)i, w, c));
  }
  return SDMI_OK;
}

int sdmi_unet_set_schedule(sdmi_unet* u, const float* temb_dev, int n_steps, void* stream) {
  if (!u || !temb_dev || n_steps < 1) { sdmi_set_error("set_schedule: bad arguments"); return SDMI_EINVAL; }
  TRY(u->enter(stream));
  TRY(u->compute_timevecs(temb_dev, n_steps, u->timevec));
  u->n_steps = n_steps;
  return SDMI_OK;
}

// Arena bytes one forward of `batch` images of h x w latents may take: the bump allocations of a step add up to ~150 KiB per
// latent pixel and image with the fp32 residual stream (measured: sdmi_unet_arena after a 512x512 step), 256 KiB leaves room for
// the odd-size paths (V^T padding, row statistics of every plan)
static size_t arena_need(int batch, int h, int w) { return (size_t)batch * h * w * ((size_t)256 << 10); }

static int unet_forward_impl(sdmi_unet* u, const float* latents_dev, int latent_batch, const float* temb_dev, int step_idx,
                             float* eps_out_dev, int batch, int h, int w, void* stream, const StepFuse* sf) {
  if (!u || !latents_dev || (!eps_out_dev && !sf)) { sdmi_set_error("forward: null argument"); return SDMI_EINVAL; }
  SDMI_REQUIRE(latent_batch >= 1 && batch % latent_batch == 0, "forward: latent_batch=%d must divide batch=%d", latent_batch, batch);
  SDMI_REQUIRE(h >= 8 && w >= 8 && h % 8 == 0 && w % 8 == 0, "forward: latent size %dx%d must be multiples of 8", h, w);
  SDMI_REQUIRE(u->has_stem && u->has_final && u->has_time, "forward: incomplete weights (handle created with SDMI_FLAG_PARTIAL?)");
  TRY(u->enter(stream));
  u->launches = 0;
  if (!u->partial) TRY(u->ensure_arena(arena_need(batch, h, w)));
  u->arena.off = 0;
  u->launch_log.clear();
  const float* tv;
  if (temb_dev) {
    TRY(u->compute_timevecs(temb_dev, 1, u->timevec_adhoc));
    tv = u->timevec_adhoc;
    u->launches += 2 + (int)u->res_order.size();
  } else {
    SDMI_REQUIRE(step_idx >= 0 && step_idx < u->n_steps, "forward: step_idx %d outside schedule of %d", step_idx, u->n_steps);
    tv = u->timevec + (size_t)step_idx * u->time_total;
  }
  Act x;
  TRY(u->new_act(batch, h, w, u->stem_cout, true, &x));
  // the stem leaves the GroupNorm statistics of its output too (encoders.1.0's groupnorm_feature and, through the skip, the last
  // decoder stage's concat GroupNorm read them: sd/diffusion.py:173,671)
  const bool stem_rec = x.grec && x.f && u->stem_cout % 40 == 0 && (h * w) % 64 == 0 && (h * w / 64) * (u->stem_cout / kGnAtom) <= kGnRecMax;
  TRY(sdmi_launch_stem_conv(latents_dev, latent_batch, u->stem_w36, u->stem_bias, x.f ? (void*)x.f : (void*)x.h,
                            x.f != nullptr, x.f ? x.h : nullptr, batch, h, w, u->stem_cout, 4, u->st, stem_rec ? x.grec : nullptr, h * w / 64));
  if (stem_rec) { x.gok = true; x.gT = h * w / 64; x.gparts = 1; }
  u->launches += 1;
  u->log_launch("stem");
  std::vector<Act> skips;
  skips.push_back(x);
  for (size_t i = 1; i < encoders().size(); ++i) {
    Act y;
    TRY(run_stage(u, "unet.encoders." + std::to_string(i), encoders()[i], x, nullptr, tv, &y));
    x = y;
    skips.push_back(x);
  }
  { Act y; TRY(run_stage(u, "unet.bottleneck", bottleneck(), x, nullptr, tv, &y)); x = y; }
  for (size_t i = 0; i < decoders().size(); ++i) {
    Act sk = skips.back();
    skips.pop_back();
    Act y;
    TRY(run_stage(u, "unet.decoders." + std::to_string(i), decoders()[i], x, &sk, tv, &y));
    x = y;
  }
  TRY(final_layer(u, x, eps_out_dev, sf));
  u->write_launch_log();
  return SDMI_OK;
}

int sdmi_unet_forward(sdmi_unet* u, const float* latents_dev, int latent_batch, const float* temb_dev, int step_idx,
                      float* eps_out_dev, int batch, int h, int w, void* stream) {
  return unet_forward_impl(u, latents_dev, latent_batch, temb_dev, step_idx, eps_out_dev, batch, h, w, stream, nullptr);
}

int sdmi_cfg_ddpm_step(const float* eps_dev, int do_cfg, float cfg_scale, float* latents_dev, const float* noise_dev,
                       const float* coef, int64_t n, float* eps_out_dev, void* stream) {
  if (!eps_dev || !latents_dev || !coef || n <= 0) { sdmi_set_error("cfg_ddpm_step: bad arguments"); return SDMI_EINVAL; }
  return sdmi_launch_cfg_ddpm(eps_dev, do_cfg, cfg_scale, latents_dev, noise_dev, coef, (size_t)n, eps_out_dev, (hipStream_t)stream);
}

int sdmi_unet_denoise_step(sdmi_unet* u, float* latents_dev, int step_idx, int do_cfg, float cfg_scale,
                           const float* noise_dev, const float* coef, int h, int w, void* stream) {
  return sdmi_unet_denoise_step_batch(u, latents_dev, 1, step_idx, do_cfg, cfg_scale, noise_dev, coef, h, w, stream);
}

// P prompts through ONE chain (throughput mode): the UNet runs batch 2P (P with guidance off) -- the P conditional images first,
// then the P unconditional ones, exactly latents.repeat(2, 1, 1, 1) / cat([cond, uncond]) of sd/pipeline.py:118-122,221 at
// latent batch P -- so every weight is read once for P prompts and the launch-bound low-resolution levels do P times the work
// per launch.  All prompts are at the same step of the same schedule (same coef); the context set by sdmi_unet_set_context
// has batch 2P in that order.
int sdmi_unet_denoise_step_batch(sdmi_unet* u, float* latents_dev, int n_prompts, int step_idx, int do_cfg, float cfg_scale,
                                 const float* noise_dev, const float* coef, int h, int w, void* stream) {
  if (!u) { sdmi_set_error("denoise_step: null handle"); return SDMI_EINVAL; }
  SDMI_REQUIRE(n_prompts >= 1 && n_prompts * (do_cfg ? 2 : 1) <= 16, "denoise_step: %d prompts (batch <= 16)", n_prompts);
  TRY(u->enter(stream));             // before any allocation: eps_buf must live on the handle's device
  const int batch = (do_cfg ? 2 : 1) * n_prompts;
  const size_t need = (size_t)batch * 4 * h * w;
  if (u->eps_elems < need) {
    TRY(u->dmalloc(&u->eps_buf, need * 4));
    u->eps_elems = need;
  }
  // SDMI_STEP_FUSE=0: the output conv writes eps and cfg_ddpm_kernel makes the step (two launches; A/B and the bit-identity test)
  static const bool fuse_on = !(getenv("SDMI_STEP_FUSE") && atoi(getenv("SDMI_STEP_FUSE")) == 0);
  if (fuse_on && w % 4 == 0 && !u->accurate) {       // (accurate mode: the output conv reads fp32 and the step stays its own launch)
    if (!coef) { sdmi_set_error("denoise_step: null coefficients"); return SDMI_EINVAL; }
    StepFuse sf{latents_dev, noise_dev, coef, n_prompts, do_cfg, cfg_scale};
    return unet_forward_impl(u, latents_dev, n_prompts, nullptr, step_idx, nullptr, batch, h, w, stream, &sf);
  }
  TRY(sdmi_unet_forward(u, latents_dev, n_prompts, nullptr, step_idx, u->eps_buf, batch, h, w, stream));
  TRY(sdmi_launch_cfg_ddpm(u->eps_buf, do_cfg, cfg_scale, latents_dev, noise_dev, coef, (size_t)n_prompts * 4 * h * w, nullptr, (hipStream_t)stream));
  u->launches += 1;
  return SDMI_OK;
}

int sdmi_unet_run_block(sdmi_unet* u, const char* prefix, int kind, int arg, const float* x0_dev, int c0,
                        const float* x1_dev, int c1, int batch, int h, int w, const float* time_dev, float* out_dev,
                        void* stream) {
  if (!u || !prefix || !x0_dev || !out_dev) { sdmi_set_error("run_block: null argument"); return SDMI_EINVAL; }
  TRY(u->enter(stream));
  u->arena.off = 0;
  u->launches = 0;
  Act x0, x1, y;
  TRY(import_act(u, x0_dev, batch, h, w, c0, &x0));
  if (x1_dev) TRY(import_act(u, x1_dev, batch, h, w, c1, &x1));
  const std::string p(prefix);
  if (kind == 0) {
    auto it = u->res.find(p);
    if (it == u->res.end()) { sdmi_set_error("block '%s' not loaded", prefix); return SDMI_ENOENT; }
    SDMI_REQUIRE(time_dev != nullptr, "run_block: residual block needs time_dev");
    float* tv = (float*)u->arena.alloc((size_t)it->second.cout * 4);
    TRY(u->compute_timevec_from_time(time_dev, it->second, tv));
    TRY(u->res_block(it->second, x0, x1_dev ? &x1 : nullptr, tv, &y));
  } else if (kind == 1) {
    auto it = u->attn.find(p);
    if (it == u->attn.end()) { sdmi_set_error("block '%s' not loaded", prefix); return SDMI_ENOENT; }
    TRY(u->attn_block(it->second, x0, &y));
  } else if (kind == 2 || kind == 3) {
    auto it = u->convs.find(kind == 2 ? p + ".conv" : p);
    if (it == u->convs.end()) { sdmi_set_error("conv '%s' not loaded", prefix); return SDMI_ENOENT; }
    TRY(u->conv3(it->second, x0, kind == 3 ? arg : 1, kind == 2 ? 1 : 0, &y));
  } else if (kind == 4) {
    return final_layer(u, x0, out_dev);
  } else {
    sdmi_set_error("run_block: unknown kind %d", kind);
    return SDMI_EINVAL;
  }
  return export_act(u, y, out_dev);
}

int sdmi_unet_profile(sdmi_unet* u, int enable) {
  if (!u) { sdmi_set_error("profile: null handle"); return SDMI_EINVAL; }
  for (auto& r : u->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  u->prof.clear();
  u->profiling = enable != 0;
  return SDMI_OK;
}

int sdmi_unet_profile_read(sdmi_unet* u, double* ms_by_class, double* flops_by_class, int* launches_by_class) {
  if (!u || !ms_by_class || !flops_by_class || !launches_by_class) { sdmi_set_error("profile_read: null argument"); return SDMI_EINVAL; }
  for (int c = 0; c < 4; ++c) { ms_by_class[c] = 0; flops_by_class[c] = 0; launches_by_class[c] = 0; }
  for (auto& r : u->prof) {
    SDMI_CHECK_HIP(hipEventSynchronize(r.e1));
    float ms = 0.f;
    SDMI_CHECK_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
    ms_by_class[r.cls] += ms;
    flops_by_class[r.cls] += r.flops;
    launches_by_class[r.cls] += 1;
  }
  return SDMI_OK;
}

// LayerNorm-fold guard: rows beyond the threshold (|mean| > 8 sigma, SDMI_LN_GUARD_SIGMA) met by folded GEMMs since the last
// reset; synchronises the stream (call it when a loop is over, not inside it).  fold_on >= 0 also switches the fold on / off.
int sdmi_unet_ln_guard(sdmi_unet* u, int* hits_out, int reset, int fold_on, void* stream) {
  if (!u) { sdmi_set_error("ln_guard: null handle"); return SDMI_EINVAL; }
  TRY(u->enter(stream));
  if (hits_out) {
    SDMI_CHECK_HIP(hipMemcpyAsync(hits_out, u->ln_guard, sizeof(int), hipMemcpyDeviceToHost, u->st));
    SDMI_CHECK_HIP(hipStreamSynchronize(u->st));
  }
  if (reset) SDMI_CHECK_HIP(hipMemsetAsync(u->ln_guard, 0, sizeof(int), u->st));
  if (fold_on >= 0) u->ln_fold_on = fold_on != 0;
  return SDMI_OK;
}

int sdmi_unet_last_launch_count(const sdmi_unet* u) { return u ? u->launches : 0; }
int sdmi_unet_tuned_shapes(const sdmi_unet* u) { return u ? u->tuned_shapes : 0; }
int sdmi_unet_device(const sdmi_unet* u) { return u ? u->device : -1; }
// FNV-1a 64 of the loaded library file: the key of the per-library plan cache (engine.h PlanStore) and of the committed counter
// profiles bench.py quotes (a profile is only valid for the binary it was taken with)
uint64_t sdmi_library_hash(void) {
  static const uint64_t h = PlanStore::fnv_file(PlanStore::lib_path());
  return h;
}
int sdmi_unet_arena(const sdmi_unet* u, int64_t* capacity_out, int64_t* peak_out) {
  if (!u) { sdmi_set_error("sdmi_unet_arena: null handle"); return SDMI_EINVAL; }
  if (capacity_out) *capacity_out = (int64_t)u->arena.cap;
  if (peak_out) *peak_out = (int64_t)u->arena.peak;
  return SDMI_OK;
}
int64_t sdmi_unet_weight_bytes(const sdmi_unet* u) { return u ? u->weight_bytes : 0; }

// ---- kernel-level entry points --------------------------------------------------------------
static f16* g_zero = nullptr;
static float* g_slab = nullptr;
static size_t g_slab_bytes = 0;
static int ensure_globals(size_t slab_need) {
  if (!g_zero) {
    SDMI_CHECK_HIP(hipMalloc((void**)&g_zero, 4096));
    SDMI_CHECK_HIP(hipMemset(g_zero, 0, 4096));
    SDMI_CHECK_HIP(hipMemsetD16((hipDeviceptr_t)(g_zero + 1024), 0x3C00, 1024));   // second half: fp16 ones
  }
  if (slab_need > g_slab_bytes) {
    if (g_slab) (void)hipFree(g_slab);
    SDMI_CHECK_HIP(hipMalloc((void**)&g_slab, slab_need));
    g_slab_bytes = slab_need;
  }
  return SDMI_OK;
}

// statistics records of an op-level GEMM: fills a.gacc from the descriptor; T / parts as the launch will write them
static int gacc_layout(const sdmi_gemm_desc* d, GemmArgs& a, int* T, int* parts) {
  *T = 0; *parts = 0;
  if (!d->gacc) return SDMI_OK;
  SDMI_REQUIRE(d->gacc_atom >= 4 && d->N % d->gacc_atom == 0 && d->gacc_rows_img > 0, "op_gemm: bad statistics arguments");
  a.gacc.rec = d->gacc; a.gacc.atom = d->gacc_atom; a.gacc.natoms = d->N / d->gacc_atom; a.gacc.rows_img = d->gacc_rows_img;
  a.gacc.mod = d->phase2 ? d->M / 4 : d->M;
  const int cfg = d->cfg < 0 ? sdmi_gemm_pick_cfg(a) : d->cfg;
  SDMI_REQUIRE(cfg >= 0 && cfg < sdmi_gemm_num_cfgs(), "op_gemm: bad cfg %d", cfg);
  // the layout follows the split-K factor the launcher will really use (it clamps the requested one to the K-steps there are): a
  // request of 2 on a K = 64 GEMM runs the one-pass epilogue, whose records are parts = 2
  if (sdmi_gemm_effective_ksplit(a, cfg) > 1) {
    a.gacc.parts = 1;
    SDMI_REQUIRE(sdmi_finalize_gacc_ok(a, &a.gacc.T), "op_gemm: the split-K combine cannot take GroupNorm statistics for this shape");
  } else {
    a.ksplit = 1;
    a.gacc.parts = 2;
    SDMI_REQUIRE(sdmi_gemm_gacc_ok(a, cfg), "op_gemm: config %s cannot accumulate GroupNorm statistics for this shape", sdmi_gemm_cfg_name(cfg));
    a.gacc.T = sdmi_gemm_gacc_T(a, cfg);
  }
  *T = a.gacc.T; *parts = a.gacc.parts;
  return SDMI_OK;
}

int sdmi_op_gemm(const sdmi_gemm_desc* d, void* stream) {
  if (!d) { sdmi_set_error("op_gemm: null desc"); return SDMI_EINVAL; }
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.a0 = (const f16*)d->a0; a.a1 = (const f16*)d->a1; a.C0 = d->c0; a.C1 = d->c1;
  a.Hs = d->hs; a.Ws = d->ws; a.Ho = d->ho; a.Wo = d->wo; a.ups = d->ups; a.stride = d->stride; a.pad = d->pad; a.ks = d->ks;
  a.M = d->M; a.N = d->N; a.K = d->K; a.w = (const f16*)d->w; a.bias = d->bias;
  a.res = d->res; a.res_f32 = d->res_f32; a.ldr = d->ldr;
  a.out = d->out; a.out_f32 = d->out_f32; a.ldc = d->ldc; a.out16 = (f16*)d->out16;
  a.outT = (f16*)d->out_t; a.nt0 = d->nt0; a.S = d->S; a.ldt = d->ldt; a.tperm = d->out_t_perm;
  a.x0 = (const f16*)d->x0; a.x1 = (const f16*)d->x1; a.X0 = d->cx0; a.X1 = d->cx1;
  a.rowstat = d->rowstat; a.ln_stat = d->ln_stat; a.ln_ntn = d->ln_ntn; a.ln_g = d->ln_g; a.ln_C = d->ln_c; a.ln_eps = d->ln_eps;
  a.act = d->act; a.sm_valid = d->sm_valid; a.img_rows = d->img_rows; a.w_img_stride = d->w_img_stride;
  a.vec_img_stride = d->vec_img_stride; a.ldw = d->ldw; a.phase2 = d->phase2;
  a.ln_ksteps = d->ln_ksteps; a.ln_out = d->ln_out;
  a.ln_guard = d->ln_guard; a.ln_guard_thr2 = d->ln_guard_sigma * d->ln_guard_sigma;
  a.gna_rec = d->gna_rec; a.gna_gamma = d->gna_gamma; a.gna_beta = d->gna_beta; a.gna_eps = d->gna_eps;
  a.gna_T = d->gna_t; a.gna_parts = d->gna_parts; a.gna_atom = d->gna_atom; a.gna_rows = d->gna_rows;
  a.ksplit = d->ksplit < 1 ? 1 : d->ksplit;
  a.a0f = d->a0f; a.a1f = d->a1f; a.x0f = d->x0f; a.x1f = d->x1f; a.accurate = d->accurate;
  if (d->hgn_x0) {
    HaloGn& g = a.hgn;
    g.x0 = d->hgn_x0; g.x1 = d->hgn_x1; g.in_f32 = d->hgn_in_f32; g.C0 = d->hgn_c0; g.C1 = d->hgn_c1;
    g.gamma = d->hgn_gamma; g.beta = d->hgn_beta; g.eps = d->hgn_eps; g.silu = d->hgn_silu;
    g.rec0 = d->hgn_rec0; g.rec1 = d->hgn_rec1; g.T0 = d->hgn_t0; g.T1 = d->hgn_t1; g.P0 = d->hgn_p0; g.P1 = d->hgn_p1; g.atom = d->hgn_atom;
    if (!a.a0) a.a0 = (const f16*)d->hgn_x0;                 // (the launcher's null check; never read)
  }
  if (a.accurate) a.a0 = a.a0 ? a.a0 : (const f16*)d->a0f;          // (the launcher's null check; never read)
  int cfg = d->cfg;
  if (a.accurate && cfg < 0) {
    TRY(ensure_globals((size_t)16 * a.M * a.N * 4));
    a.slab = g_slab;
    int ks = 1;
    cfg = sdmi_gemm_pick_acc_cfg(a, &ks);
    if (d->ksplit <= 1) a.ksplit = ks;
  }
  TRY(ensure_globals(a.ksplit > 1 ? (size_t)a.ksplit * a.M * a.N * 4 : 0));
  a.zero = g_zero; a.slab = g_slab;
  int T = 0, parts = 0;
  TRY(gacc_layout(d, a, &T, &parts));
  return sdmi_launch_gemm(a, cfg < 0 ? sdmi_gemm_pick_cfg(a) : cfg, (hipStream_t)stream);
}
int sdmi_op_gemm_stat_layout(const sdmi_gemm_desc* d, int* T, int* parts) {
  if (!d || !T || !parts) { sdmi_set_error("op_gemm_stat_layout: null argument"); return SDMI_EINVAL; }
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.M = d->M; a.N = d->N; a.K = d->K; a.ks = d->ks; a.phase2 = d->phase2; a.img_rows = d->img_rows; a.act = d->act;
  a.outT = (f16*)d->out_t; a.ksplit = d->ksplit < 1 ? 1 : d->ksplit;
  SDMI_REQUIRE(d->gacc != nullptr, "op_gemm_stat_layout: the descriptor asks for no statistics");
  return gacc_layout(d, a, T, parts);
}
// Micro-benchmark: `iters` back-to-back launches bracketed by two HIP events on `stream`; iters < 0: -iters launches timed one
// by one, each behind a 64 MiB fill that evicts the eight L2s (the state a GEMM finds inside the denoising step: csrc/engine.h
// tune_gemm), minimum returned.
int sdmi_bench_gemm(const sdmi_gemm_desc* d, int iters, float* us_per_iter, void* stream) {
  if (!d || !us_per_iter || iters == 0) { sdmi_set_error("bench_gemm: bad arguments"); return SDMI_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  SDMI_CHECK_HIP(hipEventCreate(&e0));
  SDMI_CHECK_HIP(hipEventCreate(&e1));
  TRY(sdmi_op_gemm(d, stream));
  if (iters < 0) {
    static char* thrash = nullptr;
    const size_t kThrash = (size_t)64 << 20;
    if (!thrash) SDMI_CHECK_HIP(hipMalloc((void**)&thrash, kThrash));
    float best = 1e30f;
    for (int i = 0; i < -iters; ++i) {
      SDMI_CHECK_HIP(hipMemsetAsync(thrash, i, kThrash, st));
      SDMI_CHECK_HIP(hipEventRecord(e0, st));
      TRY(sdmi_op_gemm(d, stream));
      SDMI_CHECK_HIP(hipEventRecord(e1, st));
      SDMI_CHECK_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      SDMI_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
      if (ms * 1e3f < best) best = ms * 1e3f;
    }
    *us_per_iter = best;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return SDMI_OK;
  }
  SDMI_CHECK_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) TRY(sdmi_op_gemm(d, stream));
  SDMI_CHECK_HIP(hipEventRecord(e1, st));
  SDMI_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  SDMI_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  *us_per_iter = ms * 1e3f / iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return SDMI_OK;
}
int sdmi_op_b2b(const sdmi_b2b_desc* d, int iters, float* us_per_iter, void* stream) {
  if (!d || iters < 1) { sdmi_set_error("op_b2b: bad arguments"); return SDMI_EINVAL; }
  B2bArgs a;
  memset(&a, 0, sizeof(a));
  a.a1 = (const f16*)d->a1; a.lda1 = d->lda1; a.w1 = (const f16*)d->w1; a.b1 = d->b1; a.r1 = d->r1; a.r1_f32 = d->r1_f32;
  a.s32 = d->s32; a.s16 = (f16*)d->s16; a.w2 = (const f16*)d->w2; a.K2 = d->K2; a.h2 = d->h2; a.partial = d->partial;
  a.cscale = d->cscale; a.r2 = d->r2; a.r2_f32 = d->r2_f32; a.out = d->out; a.out_f32 = d->out_f32; a.out16 = (f16*)d->out16;
  a.M = d->M; a.eps = d->eps;
  a.npass2 = d->npass2 > 0 ? d->npass2 : 1; a.ldo = d->ldo > 0 ? d->ldo : 320; a.vt = (f16*)d->vt; a.S = d->S; a.ldt = d->ldt;
  a.gx = d->gx; a.gx_f32 = d->gx_f32; a.gn_partial = d->gn_partial; a.gn_nchunk = d->gn_nchunk; a.gn_gamma = d->gn_gamma;
  a.gn_beta = d->gn_beta; a.gn_eps = d->gn_eps;
  if (d->gacc) {
    SDMI_REQUIRE(d->gacc_atom >= 4 && 320 % d->gacc_atom == 0 && d->gacc_rows_img > 0, "op_b2b: bad statistics arguments");
    a.gacc.rec = d->gacc; a.gacc.atom = d->gacc_atom; a.gacc.natoms = 320 / d->gacc_atom; a.gacc.rows_img = d->gacc_rows_img; a.gacc.mod = d->M;
    const int bm = d->bm ? d->bm : sdmi_b2b_tile_rows(a);
    a.gacc.T = d->gacc_rows_img / bm; a.gacc.parts = 1;          // one record row per tile of bm rows
  }
  hipStream_t st = (hipStream_t)stream;
  if (!us_per_iter) {
    for (int i = 0; i < iters; ++i) TRY(sdmi_launch_b2b(a, st, d->bm));
    return SDMI_OK;
  }
  hipEvent_t e0, e1;
  SDMI_CHECK_HIP(hipEventCreate(&e0));
  SDMI_CHECK_HIP(hipEventCreate(&e1));
  TRY(sdmi_launch_b2b(a, st, d->bm));
  SDMI_CHECK_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) TRY(sdmi_launch_b2b(a, st, d->bm));
  SDMI_CHECK_HIP(hipEventRecord(e1, st));
  SDMI_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  SDMI_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  *us_per_iter = ms * 1e3f / iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return SDMI_OK;
}
int sdmi_gemm_num_configs(void) { return sdmi_gemm_num_cfgs(); }
const char* sdmi_gemm_config_name(int cfg) { return sdmi_gemm_cfg_name(cfg); }

void sdmi_gemm_config_dims(int cfg, int* bm, int* bn) { sdmi_gemm_cfg_dims(cfg, bm, bn); }
int sdmi_op_ln_fold_prep(const void* w_dev, int w_dtype, const float* gamma, const float* beta, const float* bias,
                         void* w_out, float* g_out, float* h_out, int N, int C, void* stream) {
  return sdmi_launch_ln_fold_prep(w_dev, w_dtype == SDMI_F32, gamma, beta, bias, (f16*)w_out, g_out, h_out, N, C, (hipStream_t)stream);
}

int sdmi_op_pack_ups_phase(const void* w_dev, int w_dtype, void* out_dev, int O, int I, void* stream) {
  return sdmi_launch_pack_ups_phase(w_dev, w_dtype == SDMI_F32, (f16*)out_dev, O, I, (hipStream_t)stream);
}
int sdmi_op_pack_conv(const void* w_dev, int w_dtype, void* out_dev, int O, int I, int ks, int o_keep, void* stream) {
  return sdmi_launch_pack_conv(w_dev, w_dtype == SDMI_F32, (f16*)out_dev, O, I, ks, o_keep, (hipStream_t)stream);
}

int sdmi_op_attention(const void* q, int ldq, const void* k, int ldk, int k_batch_stride, const void* vt, int ldvt, void* o,
                      int ldo, int B, int H, int d, int Sq, int Skv, void* stream) {
  TRY(ensure_globals(0));
  AttnArgs t;
  memset(&t, 0, sizeof(t));
  t.q = (const f16*)q; t.ldq = ldq; t.k = (const f16*)k; t.ldk = ldk; t.k_batch_stride = k_batch_stride;
  t.vt = (const f16*)vt; t.ldvt = ldvt; t.o = (f16*)o; t.ldo = ldo; t.B = B; t.H = H; t.d = d; t.Sq = Sq; t.Skv = Skv;
  t.zero = g_zero; t.ones = g_zero + 1024; t.scale = 1.f / sqrtf((float)d); t.prescaled = 0;
  return sdmi_launch_attention(t, (hipStream_t)stream);
}

int sdmi_op_groupnorm(const void* x0, const void* x1, int in_f32, int c0, int c1, int B, int P, const float* gamma,
                      const float* beta, float eps, int silu, void* y_f16, void* stream) {
  static float* partial = nullptr;
  if (!partial) SDMI_CHECK_HIP(hipMalloc((void**)&partial, (size_t)64 * 128 * 32 * 2 * 4));
  SDMI_REQUIRE(B <= 64, "op_groupnorm: batch %d > 64", B);
  GnArgs g;
  memset(&g, 0, sizeof(g));
  g.x0 = x0; g.x1 = x1; g.in_f32 = in_f32; g.C0 = c0; g.C1 = c1; g.B = B; g.P = P; g.gamma = gamma; g.beta = beta;
  g.eps = eps; g.silu = silu; g.y = (f16*)y_f16; g.partial = partial; g.nchunk = sdmi_gn_nchunk(P);
  return sdmi_launch_groupnorm(g, (hipStream_t)stream);
}

int sdmi_op_groupnorm_acc(const void* x0, const void* x1, int in_f32, int c0, int c1, int B, int P, const float* rec0, int T0,
                          int parts0, const float* rec1, int T1, int parts1, int atom, const float* gamma, const float* beta,
                          float eps, int silu, void* y_f16, void* stream) {
  static float* partial = nullptr;
  if (!partial) SDMI_CHECK_HIP(hipMalloc((void**)&partial, 256));
  SDMI_REQUIRE(rec0 && atom >= 4, "op_groupnorm_acc: bad arguments");
  GnArgs g;
  memset(&g, 0, sizeof(g));
  g.x0 = x0; g.x1 = x1; g.in_f32 = in_f32; g.C0 = c0; g.C1 = c1; g.B = B; g.P = P; g.gamma = gamma; g.beta = beta;
  g.eps = eps; g.silu = silu; g.y = (f16*)y_f16; g.partial = partial; g.nchunk = sdmi_gn_nchunk(P);
  g.acc0 = rec0; g.accT0 = T0; g.accP0 = parts0; g.acc1 = rec1; g.accT1 = T1; g.accP1 = parts1; g.atom = atom;
  return sdmi_launch_groupnorm(g, (hipStream_t)stream);
}

// GroupNorm(32)(+SiLU) of x = sum_z slab[z] + bias (+ res): the split-K combine of a conv and the norm that follows it in one
// launch (maps the single-launch kernel takes: P <= 1024, (C/32) % 4 == 0).  out32 / out16: optional copies of x.
int sdmi_op_groupnorm_slab(const float* slab, int ksplit, const float* bias, const void* res, int res_f32, int C, int B, int P,
                           const float* gamma, const float* beta, float eps, int silu, void* y_f16, float* out32, void* out16,
                           void* stream) {
  static float* partial = nullptr;
  if (!partial) SDMI_CHECK_HIP(hipMalloc((void**)&partial, (size_t)64 * 128 * 32 * 2 * 4));
  SDMI_REQUIRE(slab && ksplit >= 1 && B <= 64, "op_groupnorm_slab: bad arguments");
  GnArgs g;
  memset(&g, 0, sizeof(g));
  g.C0 = C; g.B = B; g.P = P; g.gamma = gamma; g.beta = beta; g.eps = eps; g.silu = silu; g.y = (f16*)y_f16;
  g.partial = partial; g.nchunk = sdmi_gn_nchunk(P);
  g.slab = slab; g.ksplit = ksplit; g.sbias = bias; g.sres = res; g.sres_f32 = res_f32; g.sout = out32; g.sout16 = (f16*)out16;
  return sdmi_launch_groupnorm(g, (hipStream_t)stream);
}

int sdmi_gn_num_chunks(int P) { return sdmi_gn_nchunk(P); }
int sdmi_op_gn_stats(const void* x0, const void* x1, int in_f32, int c0, int c1, int B, int P, float* partial_out, void* stream) {
  GnArgs g;
  memset(&g, 0, sizeof(g));
  g.x0 = x0; g.x1 = x1; g.in_f32 = in_f32; g.C0 = c0; g.C1 = c1; g.B = B; g.P = P;
  g.partial = partial_out; g.nchunk = sdmi_gn_nchunk(P);
  return sdmi_launch_gn_stats(g, (hipStream_t)stream);
}

int sdmi_op_layernorm(const void* x, int in_f32, int M, int C, const float* gamma, const float* beta, float eps,
                      void* y_f16, void* stream) {
  LnArgs l;
  memset(&l, 0, sizeof(l));
  l.x = x; l.in_f32 = in_f32; l.M = M; l.C = C; l.gamma = gamma; l.beta = beta; l.eps = eps; l.y = (f16*)y_f16;
  return sdmi_launch_layernorm(l, (hipStream_t)stream);
}

}  // extern "C"
