// Native VAE decoder (reference sd/decoder.py:7-374) on the same HIP kernels as the
// UNet: implicit-GEMM convs (incl. fused nearest x2 upsample), two-pass GroupNorm(+SiLU), and -- for
// the single-head d=512 attention block -- three GEMMs around a row-softmax kernel.
//
// Reference quirks reproduced on purpose (parity target = the reference's behaviour):
//   Q3  VAE_AttentionBlock.groupnorm is loaded but never applied            (sd/decoder.py:31,34-73)
//   Q4  the (n, h*w, c) attention output is reinterpreted as (n, c, h, w)   (sd/decoder.py:62,67)
#include "engine.h"
#include "../../include/sdmi.h"

using namespace sdmi;

namespace {

struct VaeAttnW { ConvW in_proj, out_proj; int C = 0; };
struct VaeOp { int kind; int a, b; };   // 0 conv3x3(cin,cout) 1 res(cin,cout) 2 attn(c) 3 up 4 final-gn(c) 5 silu

const std::vector<VaeOp>& decoder_ops() {
  // nn.Sequential positions of sd/decoder.py:196-340 (index = state-dict key prefix)
  static const std::vector<VaeOp> v = {
      {6, 4, 4},      // 0: conv1x1 4->4 (pointwise, NCHW)
      {7, 4, 512},    // 1: conv3x3 4->512 (stem kernel)
      {1, 512, 512}, {2, 512, 0}, {1, 512, 512}, {1, 512, 512}, {1, 512, 512}, {1, 512, 512},
      {3, 0, 0}, {0, 512, 512}, {1, 512, 512}, {1, 512, 512}, {1, 512, 512},
      {3, 0, 0}, {0, 512, 512}, {1, 512, 256}, {1, 256, 256}, {1, 256, 256},
      {3, 0, 0}, {0, 256, 256}, {1, 256, 128}, {1, 128, 128}, {1, 128, 128},
      {4, 128, 0}, {5, 0, 0}, {8, 128, 3},   // 23: GroupNorm, 24: SiLU, 25: conv3x3 128->3 (NCHW out)
  };
  return v;
}

const std::vector<VaeOp>& encoder_ops() {
  // nn.Sequential positions of sd/encoder.py:8-94
  static const std::vector<VaeOp> v = {
      {7, 3, 128},                                   // 0: conv3x3 3->128 (stem kernel, NCHW fp32 image in)
      {1, 128, 128}, {1, 128, 128}, {9, 128, 128},   // 3: stride-2 conv, asymmetric pad
      {1, 128, 256}, {1, 256, 256}, {9, 256, 256},
      {1, 256, 512}, {1, 512, 512}, {9, 512, 512},
      {1, 512, 512}, {1, 512, 512}, {1, 512, 512}, {2, 512, 0}, {1, 512, 512},
      {4, 512, 0}, {5, 0, 0}, {8, 512, 8},           // 15 GroupNorm, 16 SiLU, 17 conv3x3 512->8 (NCHW out)
      {6, 8, 8},                                     // 18: conv1x1 8->8 (pointwise, NCHW)
  };
  return v;
}

}  // namespace

struct sdmi_vae : Engine {
  bool is_encoder = false;
  int stem_cin = 4, stem_cout = 512;
  float* mom = nullptr; float* mom2 = nullptr; size_t mom_elems = 0;
  float* w0 = nullptr; float* b0 = nullptr;          // 0: 1x1 conv 4->4 (fp32)
  float* stem_w = nullptr; float* stem_b = nullptr;  // 1: conv 4->512
  std::map<int, ResW> vres;
  std::map<int, VaeAttnW> vattn;
  std::map<int, ConvW> vconv;
  NormW out_gn; ConvW out_conv;
  float* tmp_lat = nullptr; size_t tmp_lat_elems = 0;

  int load_vres(int idx, int cin, int cout) {
    const std::string p = std::to_string(idx);
    ResW r;
    r.cin = cin; r.cout = cout;
    TRY(load_norm(p + ".groupnorm_1", cin, &r.gn1));
    TRY(load_conv(p + ".conv_1", cout, cin, 3, true, &r.conv1));
    TRY(load_norm(p + ".groupnorm_2", cout, &r.gn2));
    TRY(load_conv(p + ".conv_2", cout, cout, 3, true, &r.conv2));
    r.has_skip = cin != cout;
    if (r.has_skip) TRY(load_conv(p + ".residual_layer", cout, cin, 1, true, &r.skip));
    r.bias1 = r.conv1.bias;
    vres[idx] = r;
    return SDMI_OK;
  }

  // VAE_AttentionBlock (sd/decoder.py:34-73): single head, d = C, no groupnorm (Q3), reinterpreting add (Q4)
  int vae_attn_block(const VaeAttnW& w, const Act& x, Act* y) {
    const int B = x.B, P = x.H * x.W, C = w.C;
    if (x.C != C) { sdmi_set_error("vae attn: C %d vs %d", x.C, C); return SDMI_EINVAL; }
    if (P % 64 != 0) { sdmi_set_error("vae attn: h*w=%d must be a multiple of 64", P); return SDMI_EINVAL; }
    Act qk, oa;
    TRY(new_act(B, x.H, x.W, 2 * C, false, &qk));
    f16* vt = (f16*)arena.alloc((size_t)B * C * P * 2);
    f16* sc = (f16*)arena.alloc((size_t)P * P * 2);
    float* of = (float*)arena.alloc((size_t)B * P * C * 4);
    if (!vt || !sc || !of) { sdmi_set_error("activation arena exhausted"); return SDMI_ENOMEM; }
    {
      GemmArgs a = base_args(x, nullptr, w.in_proj, x.H, x.W, 1, 0);
      a.out = qk.h; a.ldc = 2 * C;
      a.outT = vt; a.nt0 = 2 * C; a.S = P; a.ldt = P;
      TRY(gemm(a));
    }
    TRY(new_act(B, x.H, x.W, C, false, &oa));
    for (int n = 0; n < B; ++n) {
      const f16* qn = qk.h + (size_t)n * P * 2 * C;
      GemmArgs s;                        // scores[P][P] = q k^T
      memset(&s, 0, sizeof(s));
      s.a0 = qn; s.C0 = C; s.lda0 = 2 * C; s.Hs = P; s.Ws = 1; s.Ho = P; s.Wo = 1; s.stride = 1; s.ks = 1;
      s.M = P; s.N = P; s.K = C; s.w = qn + C; s.ldw = 2 * C; s.out = sc; s.ldc = P; s.ksplit = 1;
      TRY(gemm(s));
      TRY(sdmi_launch_row_softmax(sc, sc, P, P, 1.f / sqrtf((float)C), st));
      launches += 1;
      GemmArgs o;                        // o[P][C] = softmax(scores) v
      memset(&o, 0, sizeof(o));
      o.a0 = sc; o.C0 = P; o.Hs = P; o.Ws = 1; o.Ho = P; o.Wo = 1; o.stride = 1; o.ks = 1;
      o.M = P; o.N = C; o.K = P; o.w = vt + (size_t)n * C * P; o.ldw = P; o.out = oa.h + (size_t)n * P * C; o.ldc = C;
      o.ksplit = 1;
      TRY(gemm(o));
    }
    {
      GemmArgs a = base_args(oa, nullptr, w.out_proj, x.H, x.W, 1, 0);
      a.out = of; a.out_f32 = 1; a.ldc = C;
      TRY(gemm(a));
    }
    TRY(new_act(B, x.H, x.W, C, true, y));
    TRY(sdmi_launch_q4_reinterpret_add(of, x.f ? (const void*)x.f : (const void*)x.h, x.f != nullptr,
                                       y->f ? (void*)y->f : (void*)y->h, y->f != nullptr, y->f ? y->h : nullptr, B, P, C, st));
    launches += 1;
    return SDMI_OK;
  }
};

extern "C" {

static int vae_create(const sdmi_tensor_desc* tensors, int n_tensors, int flags, bool encoder, sdmi_vae** out) {
  if (!tensors || !out || n_tensors <= 0) { sdmi_set_error("sdmi_vae_create: bad arguments"); return SDMI_EINVAL; }
  sdmi_vae* v = new sdmi_vae();
  v->is_encoder = encoder;
  v->flags = flags;
  v->stream_f32 = (flags & SDMI_FLAG_STREAM_F32) != 0;
  v->tune = (flags & SDMI_FLAG_NO_TUNE) == 0;
  for (int i = 0; i < n_tensors; ++i) {
    if (!tensors[i].name || !tensors[i].data_dev) { delete v; sdmi_set_error("tensor %d: null name/data", i); return SDMI_EINVAL; }
    v->src[tensors[i].name] = tensors[i];
  }
  auto fail = [&](int rc) { delete v; return rc; };
  int rc;
  if ((rc = v->dmalloc(&v->zero, 4096)) != SDMI_OK) return fail(rc);
  if (hipMemset(v->zero, 0, 4096) != hipSuccess) return fail(SDMI_EHIP);
  const auto& ops = encoder ? encoder_ops() : decoder_ops();
  for (size_t i = 0; i < ops.size(); ++i) {
    const VaeOp& op = ops[i];
    const std::string p = std::to_string(i);
    switch (op.kind) {
      case 6: {   // pointwise conv (4->4 / 8->8), kept fp32
        const sdmi_tensor_desc* t;
        if ((rc = v->need(p + ".weight", &t, 4, {op.b, op.a, 1, 1})) != SDMI_OK) return fail(rc);
        if ((rc = v->dmalloc(&v->w0, (size_t)op.a * op.b * 4)) != SDMI_OK) return fail(rc);
        if ((rc = sdmi_launch_cast_any_f32(t->data_dev, t->dtype == SDMI_F32, v->w0, (size_t)op.a * op.b, v->st)) != SDMI_OK) return fail(rc);
        if ((rc = v->load_vec(p + ".bias", op.b, &v->b0)) != SDMI_OK) return fail(rc);
        break;
      }
      case 7: {
        const sdmi_tensor_desc* t;
        if ((rc = v->need(p + ".weight", &t, 4, {op.b, op.a, 3, 3})) != SDMI_OK) return fail(rc);
        if ((rc = v->dmalloc(&v->stem_w, (size_t)9 * op.a * op.b * 4)) != SDMI_OK) return fail(rc);
        if ((rc = sdmi_launch_pack_stem(t->data_dev, t->dtype == SDMI_F32, v->stem_w, op.b, op.a, v->st)) != SDMI_OK) return fail(rc);
        if ((rc = v->load_vec(p + ".bias", op.b, &v->stem_b)) != SDMI_OK) return fail(rc);
        v->stem_cin = op.a; v->stem_cout = op.b;
        break;
      }
      case 0:
      case 9: {
        ConvW c;
        if ((rc = v->load_conv(p, op.b, op.a, 3, true, &c)) != SDMI_OK) return fail(rc);
        // the conv after an Upsample (sd/decoder.py:258,283,308): also as four 2x2 phase convs (Engine::conv3)
        if (i > 0 && ops[i - 1].kind == 3 && (rc = v->pack_ups_phase(p, &c)) != SDMI_OK) return fail(rc);
        v->vconv[(int)i] = c;
        break;
      }
      case 1:
        if ((rc = v->load_vres((int)i, op.a, op.b)) != SDMI_OK) return fail(rc);
        break;
      case 2: {
        VaeAttnW a;
        a.C = op.a;
        if ((rc = v->load_conv(p + ".attention.in_proj", 3 * op.a, op.a, 1, true, &a.in_proj)) != SDMI_OK) return fail(rc);
        if ((rc = v->load_conv(p + ".attention.out_proj", op.a, op.a, 1, true, &a.out_proj)) != SDMI_OK) return fail(rc);
        // Q3: groupnorm weights exist in the state dict but are never used; require them for strictness
        const sdmi_tensor_desc* t;
        if ((rc = v->need(p + ".groupnorm.weight", &t, 1, {op.a})) != SDMI_OK) return fail(rc);
        v->vattn[(int)i] = a;
        break;
      }
      case 4:
        if ((rc = v->load_norm(p, op.a, &v->out_gn)) != SDMI_OK) return fail(rc);
        break;
      case 8:
        if ((rc = v->load_conv(p, op.b, op.a, 3, true, &v->out_conv)) != SDMI_OK) return fail(rc);
        break;
      default: break;
    }
  }
  v->slab_bytes = (size_t)96 << 20;
  if ((rc = v->dmalloc(&v->slab, v->slab_bytes)) != SDMI_OK) return fail(rc);
  if ((rc = v->dmalloc(&v->gn_partial, (size_t)16 * 128 * 32 * 2 * 4)) != SDMI_OK) return fail(rc);
  v->arena.cap = (size_t)10 << 30;
  if ((rc = v->dmalloc(&v->arena.base, v->arena.cap)) != SDMI_OK) return fail(rc);
  if (hipDeviceSynchronize() != hipSuccess) { sdmi_set_error("vae weight packing failed"); return fail(SDMI_EHIP); }
  v->src.clear();
  *out = v;
  return SDMI_OK;
}

int sdmi_vae_decoder_create(const sdmi_tensor_desc* tensors, int n_tensors, int flags, sdmi_vae** out) {
  return vae_create(tensors, n_tensors, flags, false, out);
}
int sdmi_vae_encoder_create(const sdmi_tensor_desc* tensors, int n_tensors, int flags, sdmi_vae** out) {
  return vae_create(tensors, n_tensors, flags, true, out);
}

// image_dev: (B,3,H,W) NCHW fp32 in [-1,1]; noise_dev: (B,4,H/8,W/8); latents_dev: (B,4,H/8,W/8) fp32
// (reference: sd/encoder.py:95-155 incl. the asymmetric pad before each stride-2 conv and the x0.18215 scale).
int sdmi_vae_encode(sdmi_vae* v, const float* image_dev, const float* noise_dev, float* latents_dev, int batch, int H, int W,
                    void* stream) {
  if (!v || !image_dev || !noise_dev || !latents_dev) { sdmi_set_error("vae_encode: null argument"); return SDMI_EINVAL; }
  SDMI_REQUIRE(v->is_encoder, "vae_encode: handle is a decoder");
  SDMI_REQUIRE(batch >= 1 && batch <= 8 && H % 64 == 0 && W % 64 == 0 && H >= 64 && W >= 64, "vae_encode: batch=%d H=%d W=%d unsupported", batch, H, W);
  TRY(v->enter(stream));
  v->arena.off = 0;
  v->launches = 0;
  Act x;
  TRY(v->new_act(batch, H, W, v->stem_cout, true, &x));
  TRY(sdmi_launch_stem_conv(image_dev, batch, v->stem_w, v->stem_b, x.f ? (void*)x.f : (void*)x.h, x.f != nullptr,
                            x.f ? x.h : nullptr, batch, H, W, v->stem_cout, v->stem_cin, v->st));
  v->launches += 1;
  const auto& ops = encoder_ops();
  for (size_t i = 1; i < ops.size(); ++i) {
    const VaeOp& op = ops[i];
    Act y;
    switch (op.kind) {
      case 9:
        TRY(v->conv3(v->vconv[(int)i], x, 2, 0, &y, 0));
        x = y;
        break;
      case 1: {
        const ResW& r = v->vres[(int)i];
        TRY(v->res_block(r, x, nullptr, r.bias1, &y));
        x = y;
        break;
      }
      case 2:
        TRY(v->vae_attn_block(v->vattn[(int)i], x, &y));
        x = y;
        break;
      case 4: {
        Act t;
        TRY(v->groupnorm(x, nullptr, v->out_gn, 1e-5f, 1, &t));
        const size_t n = (size_t)batch * 8 * x.H * x.W;
        if (v->mom_elems < n) { TRY(v->dmalloc(&v->mom, n * 4)); TRY(v->dmalloc(&v->mom2, n * 4)); v->mom_elems = n; }
        TRY(sdmi_launch_final_conv(t.h, v->out_conv.w, v->out_conv.bias, v->mom, x.B, x.H, x.W, x.C, 8, v->st));
        TRY(sdmi_launch_conv1x1_nchw_small(v->mom, v->w0, v->b0, v->mom2, batch, 8, 8, (size_t)x.H * x.W, 1.0f, v->st));
        TRY(sdmi_launch_vae_sample(v->mom2, noise_dev, latents_dev, batch, (size_t)x.H * x.W, v->st));
        v->launches += 3;
        return SDMI_OK;
      }
      default: break;
    }
  }
  sdmi_set_error("vae_encode: malformed op list");
  return SDMI_EINVAL;
}

void sdmi_vae_destroy(sdmi_vae* v) {
  if (v) { (void)hipDeviceSynchronize(); delete v; }
}

// latents_dev: (B,4,h,w) NCHW fp32, ALREADY divided by 0.18215 by the caller (the reference does that in
// place on the caller's tensor, sd/decoder.py:364); image_dev: (B,3,8h,8w) NCHW fp32.
int sdmi_vae_decode(sdmi_vae* v, const float* latents_dev, float* image_dev, int batch, int h, int w, void* stream) {
  if (!v || !latents_dev || !image_dev) { sdmi_set_error("vae_decode: null argument"); return SDMI_EINVAL; }
  SDMI_REQUIRE(!v->is_encoder, "vae_decode: handle is an encoder");
  SDMI_REQUIRE(batch >= 1 && batch <= 8 && h >= 8 && w >= 8 && (h * w) % 64 == 0, "vae_decode: batch=%d h=%d w=%d unsupported", batch, h, w);
  TRY(v->enter(stream));
  v->arena.off = 0;
  v->launches = 0;
  const size_t nlat = (size_t)batch * 4 * h * w;
  if (v->tmp_lat_elems < nlat) { TRY(v->dmalloc(&v->tmp_lat, nlat * 4)); v->tmp_lat_elems = nlat; }
  TRY(sdmi_launch_conv1x1_nchw_small(latents_dev, v->w0, v->b0, v->tmp_lat, batch, 4, 4, (size_t)h * w, 1.0f, v->st));
  Act x;
  TRY(v->new_act(batch, h, w, 512, true, &x));
  TRY(sdmi_launch_stem_conv(v->tmp_lat, batch, v->stem_w, v->stem_b, x.f ? (void*)x.f : (void*)x.h, x.f != nullptr,
                            x.f ? x.h : nullptr, batch, h, w, 512, 4, v->st));
  v->launches += 2;
  const auto& ops = decoder_ops();
  int pending_up = 0;
  for (size_t i = 2; i < ops.size(); ++i) {
    const VaeOp& op = ops[i];
    Act y;
    switch (op.kind) {
      case 0:
        TRY(v->conv3(v->vconv[(int)i], x, 1, pending_up, &y));
        pending_up = 0;
        x = y;
        break;
      case 1: {
        const ResW& r = v->vres[(int)i];
        TRY(v->res_block(r, x, nullptr, r.bias1, &y));
        x = y;
        break;
      }
      case 2:
        TRY(v->vae_attn_block(v->vattn[(int)i], x, &y));
        x = y;
        break;
      case 3:
        pending_up = 1;          // nearest x2 folded into the next conv's address generator
        break;
      case 4: {
        Act t;
        TRY(v->groupnorm(x, nullptr, v->out_gn, 1e-5f, 1, &t));   // GroupNorm + the following SiLU
        TRY(sdmi_launch_final_conv(t.h, v->out_conv.w, v->out_conv.bias, image_dev, x.B, x.H, x.W, x.C, 3, v->st));
        v->launches += 1;
        return SDMI_OK;
      }
      default: break;
    }
  }
  sdmi_set_error("vae_decode: malformed op list");
  return SDMI_EINVAL;
}

int sdmi_vae_last_launch_count(const sdmi_vae* v) { return v ? v->launches : 0; }

}  // extern "C"
