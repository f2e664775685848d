// Native CLIP text encoder (reference sd/clip.py:7-261) on the UNet's kernels:
// embedding gather, 12 x [LayerNorm -> fused q|k|v GEMM (+bias, V^T tail) -> causal flash attention (12 heads,
// d = 64, S = 77) -> out_proj GEMM (+bias, +residual) -> LayerNorm -> fc1 GEMM (+bias, quick-GELU epilogue) ->
// fc2 GEMM (+bias, +residual)], final LayerNorm written in fp32.  Residual stream fp32, GEMM operands fp16.
#include "engine.h"
#include "../../include/sdmi.h"

using namespace sdmi;

namespace {
constexpr int kVocab = 49408, kDim = 768, kTok = 77, kLayers = 12, kClipHeads = 12, kDh = 64, kVtLd = 128;
struct ClipLayerW { NormW ln1, ln2; ConvW in_proj, out_proj, fc1, fc2; };
}  // namespace

struct sdmi_clip : Engine {
  float* tok_emb = nullptr;
  float* pos_emb = nullptr;
  ClipLayerW layer[kLayers];
  NormW final_ln;
};

extern "C" {

int sdmi_clip_create(const sdmi_tensor_desc* tensors, int n_tensors, int flags, sdmi_clip** out) {
  if (!tensors || !out || n_tensors <= 0) { sdmi_set_error("sdmi_clip_create: bad arguments"); return SDMI_EINVAL; }
  sdmi_clip* c = new sdmi_clip();
  c->flags = flags;
  c->stream_f32 = true;
  c->tune = (flags & SDMI_FLAG_NO_TUNE) == 0;
  for (int i = 0; i < n_tensors; ++i) {
    if (!tensors[i].name || !tensors[i].data_dev) { delete c; sdmi_set_error("tensor %d: null name/data", i); return SDMI_EINVAL; }
    c->src[tensors[i].name] = tensors[i];
  }
  auto fail = [&](int rc) { delete c; return rc; };
  int rc;
  if ((rc = c->dmalloc(&c->zero, 4096)) != SDMI_OK) return fail(rc);
  if (hipMemset(c->zero, 0, 4096) != hipSuccess) return fail(SDMI_EHIP);
  if (hipMemsetD16((hipDeviceptr_t)(c->zero + 1024), 0x3C00, 1024) != hipSuccess) return fail(SDMI_EHIP);   // fp16 ones
  {
    const sdmi_tensor_desc* t;
    if ((rc = c->need("embedding.token_embedding.weight", &t, 2, {kVocab, kDim})) != SDMI_OK) return fail(rc);
    if ((rc = c->dmalloc(&c->tok_emb, (size_t)kVocab * kDim * 4)) != SDMI_OK) return fail(rc);
    if ((rc = sdmi_launch_cast_any_f32(t->data_dev, t->dtype == SDMI_F32, c->tok_emb, (size_t)kVocab * kDim, c->st)) != SDMI_OK) return fail(rc);
    if ((rc = c->need("embedding.position_embedding", &t, 2, {kTok, kDim})) != SDMI_OK) return fail(rc);
    if ((rc = c->dmalloc(&c->pos_emb, (size_t)kTok * kDim * 4)) != SDMI_OK) return fail(rc);
    if ((rc = sdmi_launch_cast_any_f32(t->data_dev, t->dtype == SDMI_F32, c->pos_emb, (size_t)kTok * kDim, c->st)) != SDMI_OK) return fail(rc);
  }
  for (int i = 0; i < kLayers; ++i) {
    const std::string p = "layers." + std::to_string(i);
    ClipLayerW& w = c->layer[i];
    if ((rc = c->load_norm(p + ".layernorm_1", kDim, &w.ln1)) != SDMI_OK) return fail(rc);
    if ((rc = c->load_conv(p + ".attention.in_proj", 3 * kDim, kDim, 1, true, &w.in_proj)) != SDMI_OK) return fail(rc);
    if ((rc = c->load_conv(p + ".attention.out_proj", kDim, kDim, 1, true, &w.out_proj)) != SDMI_OK) return fail(rc);
    if ((rc = c->load_norm(p + ".layernorm_2", kDim, &w.ln2)) != SDMI_OK) return fail(rc);
    if ((rc = c->load_conv(p + ".linear_1", 4 * kDim, kDim, 1, true, &w.fc1)) != SDMI_OK) return fail(rc);
    if ((rc = c->load_conv(p + ".linear_2", kDim, 4 * kDim, 1, true, &w.fc2)) != SDMI_OK) return fail(rc);
  }
  if ((rc = c->load_norm("layernorm", kDim, &c->final_ln)) != SDMI_OK) return fail(rc);
  c->slab_bytes = (size_t)32 << 20;
  if ((rc = c->dmalloc(&c->slab, c->slab_bytes)) != SDMI_OK) return fail(rc);
  c->arena.cap = (size_t)256 << 20;
  if ((rc = c->dmalloc(&c->arena.base, c->arena.cap)) != SDMI_OK) return fail(rc);
  if (hipDeviceSynchronize() != hipSuccess) { sdmi_set_error("clip weight packing failed"); return fail(SDMI_EHIP); }
  c->src.clear();
  *out = c;
  return SDMI_OK;
}

void sdmi_clip_destroy(sdmi_clip* c) {
  if (c) { (void)hipDeviceSynchronize(); delete c; }
}

// tokens_dev: (batch, 77) int64; out_dev: (batch, 77, 768) fp32   (reference: sd/clip.py:227-261)
int sdmi_clip_encode(sdmi_clip* c, const int64_t* tokens_dev, float* out_dev, int batch, void* stream) {
  if (!c || !tokens_dev || !out_dev) { sdmi_set_error("clip_encode: null argument"); return SDMI_EINVAL; }
  SDMI_REQUIRE(batch >= 1 && batch <= 64, "clip_encode: batch=%d unsupported", batch);
  TRY(c->enter(stream));
  c->arena.off = 0;
  c->launches = 0;
  const int M = batch * kTok;
  Act x;                       // (batch, 77, 1, 768) token-major
  TRY(c->new_act(batch, kTok, 1, kDim, true, &x));
  TRY(sdmi_launch_clip_embed(tokens_dev, c->tok_emb, c->pos_emb, x.f, x.h, M, kTok, kDim, kVocab, c->st));
  c->launches += 1;
  f16* vt = (f16*)c->arena.alloc((size_t)batch * kDim * kVtLd * 2);
  if (!vt) { sdmi_set_error("activation arena exhausted"); return SDMI_ENOMEM; }
  SDMI_CHECK_HIP(hipMemsetAsync(vt, 0, (size_t)batch * kDim * kVtLd * 2, c->st));   // keys 77..127 stay zero
  for (int i = 0; i < kLayers; ++i) {
    const ClipLayerW& w = c->layer[i];
    Act u, qk, ao, s1, g, s2;
    TRY(c->layernorm(x, w.ln1, &u));
    TRY(c->new_act(batch, kTok, 1, 2 * kDim, false, &qk));
    {
      GemmArgs a = Engine::base_args(u, nullptr, w.in_proj, kTok, 1, 1, 0);
      a.out = qk.h; a.ldc = 2 * kDim;
      a.outT = vt; a.nt0 = 2 * kDim; a.S = kTok; a.ldt = kVtLd; a.tperm = 1;
      a.cscale = 1.4426950408889634f / sqrtf((float)kDh); a.cs_hi = kDim;     // Q columns carry scale * log2(e)
      TRY(c->gemm(a));
    }
    TRY(c->new_act(batch, kTok, 1, kDim, false, &ao));
    {
      AttnArgs t;
      memset(&t, 0, sizeof(t));
      t.q = qk.h; t.ldq = 2 * kDim; t.k = qk.h + kDim; t.ldk = 2 * kDim; t.k_batch_stride = kTok;
      t.vt = vt; t.ldvt = kVtLd; t.o = ao.h; t.ldo = kDim; t.B = batch; t.H = kClipHeads; t.d = kDh;
      t.Sq = kTok; t.Skv = kTok; t.zero = c->zero; t.ones = c->zero + 1024; t.scale = 1.f / sqrtf((float)kDh); t.prescaled = 1; t.causal = 1;
      TRY(sdmi_launch_attention(t, c->st));
      c->launches += 1;
    }
    TRY(c->new_act(batch, kTok, 1, kDim, true, &s1));
    { GemmArgs a = Engine::base_args(ao, nullptr, w.out_proj, kTok, 1, 1, 0); Engine::set_res(a, x); c->set_out(a, s1); TRY(c->gemm(a)); }
    TRY(c->layernorm(s1, w.ln2, &u));
    TRY(c->new_act(batch, kTok, 1, 4 * kDim, false, &g));
    { GemmArgs a = Engine::base_args(u, nullptr, w.fc1, kTok, 1, 1, 0); a.out = g.h; a.ldc = 4 * kDim; a.act = 1; TRY(c->gemm(a)); }
    TRY(c->new_act(batch, kTok, 1, kDim, true, &s2));
    { GemmArgs a = Engine::base_args(g, nullptr, w.fc2, kTok, 1, 1, 0); Engine::set_res(a, s1); c->set_out(a, s2); TRY(c->gemm(a)); }
    x = s2;
  }
  LnArgs l;
  memset(&l, 0, sizeof(l));
  l.x = x.f; l.in_f32 = 1; l.M = M; l.C = kDim; l.gamma = c->final_ln.gamma; l.beta = c->final_ln.beta; l.eps = 1e-5f;
  l.y = nullptr; l.y32 = out_dev;
  TRY(sdmi_launch_layernorm(l, c->st));
  c->launches += 1;
  return SDMI_OK;
}

int sdmi_clip_last_launch_count(const sdmi_clip* c) { return c ? c->launches : 0; }

}  // extern "C"
