// Shared host-side engine of libsdmi: packed-weight loaders, activation arena, per-shape GEMM plans
// (autotune), and launch helpers for the kernels in gemm/attention/norm/misc.  Used by the UNet
// (unet.hip) and the VAE decoder (vae.hip).
#pragma once
#include "common.h"
#include "../../include/sdmi.h"

#include <dlfcn.h>
#include <sys/stat.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>


#define TRY(expr)                  \
  do {                             \
    int _rc = (expr);              \
    if (_rc != SDMI_OK) return _rc; \
  } while (0)

namespace sdmi {

constexpr int kTime = 1280, kCtx = 768, kHeads = 8, kCtxPad = 80, kCtxVtLd = 128;

struct ConvW {
  f16* w = nullptr; float* bias = nullptr; int O = 0, I = 0, ks = 0;
  f16* w4 = nullptr;   // Upsample convs only: [4 phases][O][2][2][I] pre-summed taps (GemmArgs::phase2)
};
struct NormW { float* gamma = nullptr; float* beta = nullptr; int C = 0; };
struct ResW {
  NormW gn1, gn2;
  ConvW conv1, conv2, skip;
  f16* w2s = nullptr;      // conv2 | skip weights concatenated along K: [cout][9*cout + cin] (fused skip conv)
  float* bias2s = nullptr; // conv2.bias + skip.bias
  ConvW time;             // linear_time [cout][1280]
  float* bias1 = nullptr; // conv_feature.bias + linear_time.bias (time-independent part)
  bool has_skip = false;
  int cin = 0, cout = 0, time_off = 0;
};
// Linear with the preceding LayerNorm folded in: w = gamma (.) W (fp16), g[n] = sum_c w[n][c], h[n] = b[n] + sum_c beta[c] W[n][c]
struct FoldW { f16* w = nullptr; float* g = nullptr; float* h = nullptr; };
// per-row {sum, sum of squares} partials a GEMM epilogue left behind for the LayerNorm of its output
struct RowStat { const float* ptr = nullptr; int ntn = 0; };

struct AttnW {
  NormW gn, ln1, ln2, ln3;
  ConvW conv_in, in_proj, out1, q, k, v, out2;
  FoldW in_proj_f, q_f;            // layernorm_1/2 folded into in_proj / q_proj
  // The block's feed-forward and output conv are ONE linear map of (LN3(s2), s2): the reference discards the GeGLU gate
  // (quirk Q2, sd/diffusion.py:359), so nothing non-linear sits between linear_geglu_1, linear_geglu_2 and conv_output:
  //   y = Wo (W2 (W1a LN3(s2) + b1a) + b2 + s2) + bo + x = Wf LN3(s2) + Wo s2 + bf + x,  Wf = Wo W2 W1a (C x C).
  // ffn: [C][2C] fp16 = [Wf | Wo] over the virtual concat [LN3(s2) | s2], bias bf;  ffn_f: the same with layernorm_3
  // folded into the first C columns ([gamma (.) Wf | Wo], g, h) for the partial LayerNorm fold (GemmArgs::ln_ksteps).
  ConvW ffn;
  FoldW ffn_f;
  // Folded cross-attention (Engine::xattn_fold, C >= 640): q_proj transposed and pre-multiplied by log2(e)/sqrt(d), the
  // left factor of  scores = LN2(s1) (Wq^T K_h^T)
  f16* wqT = nullptr;
  int C = 0, dh = 0, ctx_idx = 0;
};
struct Act {
  f16* h = nullptr;
  float* f = nullptr;
  int B = 0, H = 0, W = 0, C = 0;
  // GroupNorm statistics of this tensor, left behind by the kernel that wrote it (GnRec, common.h): grec = its records
  // (room for kGnRecMax float2 per image), gT / gparts = record rows per image and parts as that producer wrote them,
  // gok = the producer really did
  float* grec = nullptr;
  int gT = 0, gparts = 0;
  bool gok = false;
  int M() const { return B * H * W; }
};
constexpr int kGnAtom = 10;     // every GroupNorm group of the UNet (10 .. 80 channels) is a whole number of 10-channel atoms
constexpr int kGnRecMax = 4096; // records (T x atoms x parts) per image a producer may write (gemm.hip kGaccMaxRec)

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0, peak = 0;
  void* alloc(size_t bytes) {
    const size_t a = (off + 255) & ~(size_t)255;
    if (a + bytes > cap) return nullptr;
    off = a + bytes;
    if (off > peak) peak = off;
    return base + a;
  }
};

typedef std::tuple<int, int, int, int, int, int, int, int, int, int> ShapeKey;
struct Plan { int cfg = -1; int ksplit = 1; float us = 0.f; long calls = 0; double flops = 0.0; bool provisional = false; };

// ---- tuner plan store ---------------------------------------------------------------------------
// Plans are (cfg NAME, split-K) per GEMM shape key.  Two text tables are read when an engine is created:
//   1. the table shipped next to the library (<package>/plans/gfx950.txt, tuned once on an MI355X and committed):
//      every process then runs the SAME plans (same summation order) and pays no tuning for the known shapes;
//   2. the user's cache $SDMI_PLAN_CACHE_DIR (default ~/.cache/sdmi)/plans-<fnv64 of libsdmi.so>.txt, to which
//      every newly tuned shape is appended -- keyed by the library's content hash, so a rebuilt library re-tunes.
// SDMI_RETUNE=1 ignores both (and still appends to the cache file); a line is "k0 .. k9 cfgname ksplit us".
struct PlanStore {
  std::map<ShapeKey, std::pair<std::string, std::pair<int, float>>> table;   // key -> (cfg name, (ksplit, us))
  std::string cache_path;
  bool loaded = false;
  std::mutex mu;      // the store is process-wide; handles on several devices / threads read and append concurrently (ctypes drops the GIL)

  static std::string lib_path() {
    Dl_info info;
    if (dladdr((void*)&sdmi_last_error, &info) && info.dli_fname) return info.dli_fname;
    return "";
  }
  static unsigned long long fnv_file(const std::string& path) {
    unsigned long long h = 1469598103934665603ull;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return 0;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0)
      for (size_t i = 0; i < n; ++i) { h ^= buf[i]; h *= 1099511628211ull; }
    fclose(f);
    return h;
  }
  void read_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return;
    char line[512];
    while (fgets(line, sizeof(line), f)) {
      if (line[0] == '#') continue;
      int k[10], ks;
      char name[64];
      float us;
      if (sscanf(line, "%d %d %d %d %d %d %d %d %d %d %63s %d %f", &k[0], &k[1], &k[2], &k[3], &k[4], &k[5], &k[6], &k[7],
                 &k[8], &k[9], name, &ks, &us) != 13) continue;
      table[ShapeKey(k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[8], k[9])] = {name, {ks, us}};
    }
    fclose(f);
  }
  void load() {
    std::lock_guard<std::mutex> lk(mu);
    if (loaded) return;
    loaded = true;
    const std::string lib = lib_path();
    const char* dir = getenv("SDMI_PLAN_CACHE_DIR");
    std::string d = dir ? dir : (getenv("HOME") ? std::string(getenv("HOME")) + "/.cache/sdmi" : "/tmp/sdmi-cache");
    char hash[32];
    snprintf(hash, sizeof(hash), "%016llx", fnv_file(lib));
    (void)mkdir(d.substr(0, d.rfind('/')).c_str(), 0755);
    (void)mkdir(d.c_str(), 0755);
    cache_path = d + "/plans-" + hash + ".txt";
    if (getenv("SDMI_RETUNE")) return;
    const char* shipped = getenv("SDMI_PLAN_FILE");
    if (shipped) read_file(shipped);
    else if (!lib.empty()) {
      const size_t s1 = lib.rfind('/');
      const size_t s2 = s1 == std::string::npos ? s1 : lib.rfind('/', s1 - 1);
      if (s2 != std::string::npos) read_file(lib.substr(0, s2) + "/plans/gfx950.txt");
    }
    read_file(cache_path);            // the user's own tuning wins over the shipped table
  }
  // copy of the stored plan for `k` (false: none)
  bool lookup(const ShapeKey& k, std::string* name, int* ksplit, float* us) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = table.find(k);
    if (it == table.end()) return false;
    *name = it->second.first; *ksplit = it->second.second.first; *us = it->second.second.second;
    return true;
  }
  void append(const ShapeKey& k, const char* name, int ksplit, float us) {
    std::lock_guard<std::mutex> lk(mu);
    table[k] = {name, {ksplit, us}};
    if (cache_path.empty()) return;
    FILE* f = fopen(cache_path.c_str(), "a");
    if (!f) return;
    fprintf(f, "%d %d %d %d %d %d %d %d %d %d %s %d %.2f\n", std::get<0>(k), std::get<1>(k), std::get<2>(k), std::get<3>(k),
            std::get<4>(k), std::get<5>(k), std::get<6>(k), std::get<7>(k), std::get<8>(k), std::get<9>(k), name, ksplit, us);
    fclose(f);
  }
};
inline PlanStore& plan_store() { static PlanStore s; return s; }

struct Engine {
  int flags = 0;
  int device = -1;            // HIP device the handle (weights, arena, plans) lives on
  bool stream_f32 = false, partial = false, tune = true;
  // ACCURATE mode (SDMI_FLAG_ACCURATE): every tensor that feeds a GEMM also exists in fp32 (Act::f) and enters the product as a
  // hi + lo fp16 pair (gemm.hip igemm_kernel<.., ACC>); no LayerNorm fold, no back-to-back / halo / folded-cross-attention /
  // phase-conv forms (each of them multiplies an fp16 activation), heuristic plans (no tuner).  What is left of the fp16 path's
  // error is the fp16 rounding of the weights and the attention kernel's fp16 q / k / v / p.
  bool accurate = false;
  bool is_lane = false;       // made by sdmi_unet_clone: borrows its parent's weights, never tunes (plan_of)
  std::vector<void*> owned;   // hipMalloc'd blocks
  int64_t weight_bytes = 0;
  std::map<std::string, sdmi_tensor_desc> src;
  std::map<std::string, ResW> res;
  std::map<std::string, AttnW> attn;
  std::map<std::string, ConvW> convs;   // plain convs + upsample convs (3x3)
  // time path
  ConvW te1, te2;
  bool has_time = false;
  int time_total = 0;                  // sum of cout over residual blocks
  std::vector<std::string> res_order;  // residual block prefixes in schedule order
  float* timevec = nullptr;            // [n_steps][time_total]
  int n_steps = 0;
  float* timevec_adhoc = nullptr;      // [1][time_total]
  float* time_scratch = nullptr;       // [max_steps][1280] x2
  int max_steps = 0;
  // stem / final
  float* stem_w36 = nullptr; float* stem_bias = nullptr; int stem_cout = 0; bool has_stem = false;
  NormW final_gn; ConvW final_conv; bool has_final = false;
  // context
  std::vector<std::string> attn_order;
  f16* ctx16 = nullptr;                // [B][80][768]
  float* ctx32 = nullptr;              // the same in fp32 (accurate mode)
  std::vector<f16*> ctxK, ctxVt;       // per attention block: [B*80][C], [B][C][128]
  // Folded cross-attention.  The context is constant over the denoising loop, so q_proj and out_proj are multiplied INTO
  // the hoisted K and V once per prompt (sd/attention.py:219-256 regrouped):
  //   scores_h = LN2(s1) Wq_h^T K_h^T / sqrt(d) = LN2(s1) W1_h^T,   W1_h = K_h Wq_h / sqrt(d)   (128 x C per head and image)
  //   out      = sum_h P_h V_h Wo_h^T + bo      = P W2^T + bo,      W2[:, h*128 + j] = Wo_h V_h[j]  (C x 1024 per image)
  // so the block's cross-attention is TWO GEMMs with per-image weights -- [M x C] x [C x 1024] with the softmax over each
  // head's 77 keys in the epilogue (LayerNorm folded as usual), then [M x 1024] x [1024 x C] + residual -- instead of
  // q_proj, the attention kernel and out_proj.  Used for the C >= 640 blocks on maps of 64..1024 pixels, a multiple of 64
  // (SDMI_XATTN_FOLD=0: never).  Measured on one MI355X, same box: 4.39 -> 4.31 ms per step (316 -> 305 launches); the
  // 320-channel blocks keep the three-kernel form (P is 3.2x wider than q there: 4.32 -> 4.36 ms when folded too), and so
  // do maps beyond 1024 pixels (48x48 at 768x768: 42 us against 36 us).
  std::vector<f16*> xfW1, xfW2;        // per attention block (nullptr: not folded): [B][1024][C], [C][B*1024]
  std::vector<float*> xfG, xfH;        // LayerNorm-fold vectors of W1: [B][1024]
  f16 *xf_km = nullptr, *xf_vm = nullptr, *xf_vp = nullptr;   // set_context scratch: masked K / V [B*1024][Cmax], plain V
  float* xf_kq = nullptr;                                      //   and K Wq in fp32 [B*1024][Cmax]
  static constexpr int kXfCols = kHeads * 128;
  int ctx_batch = 0, ctx_tokens = 0;
  // scratch
  f16* zero = nullptr;
  Arena arena;
  float* slab = nullptr; size_t slab_bytes = 0;
  float* gn_partial = nullptr;
  // LayerNorm fold and its guard (GemmArgs::ln_guard): ln_fold_on = false takes the separate LayerNorm kernel everywhere (what
  // SDMI_NO_LNFOLD=1 does for a whole process); ln_guard counts rows whose |mean| exceeds ln_guard_thr sigma in any folded GEMM
  bool ln_fold_on = getenv("SDMI_NO_LNFOLD") == nullptr;
  int* ln_guard = nullptr;
  static float ln_guard_thr() { static const float v = getenv("SDMI_LN_GUARD_SIGMA") ? (float)atof(getenv("SDMI_LN_GUARD_SIGMA")) : 8.f; return v; }
  // producer-side GroupNorm statistics (GnRec): the records live in the activation arena next to the tensor they describe
  bool gacc_enabled = false;           // the UNet engine turns it on (the VAE / CLIP engines take every GroupNorm's own statistics)
  static bool gacc_on() { static const bool on = !(getenv("SDMI_GN_ACC") && atoi(getenv("SDMI_GN_ACC")) == 0); return on; }
  static int gacc_min_px() { static const int v = getenv("SDMI_GN_ACC_MINPX") ? atoi(getenv("SDMI_GN_ACC_MINPX")) : 1024; return v; }
  void attach_gacc(Act* a) {
    a->grec = nullptr; a->gok = false; a->gT = a->gparts = 0;
    if (!gacc_enabled || !gacc_on() || a->C % (32 * kGnAtom) != 0 || a->H * a->W < gacc_min_px()) return;
    a->grec = (float*)arena.alloc((size_t)a->B * kGnRecMax * 8);       // (none left: the GroupNorm takes its own statistics)
  }
  static void set_gacc(GemmArgs& a, const Act& y) {
    a.gacc.rec = y.grec; a.gacc.atom = kGnAtom; a.gacc.natoms = y.C / kGnAtom;
    a.gacc.rows_img = a.phase2 ? a.Hs * a.Ws : a.Ho * a.Wo;
    a.gacc.mod = a.phase2 ? a.M / 4 : a.M;
  }
  float* eps_buf = nullptr; size_t eps_elems = 0;
  std::map<ShapeKey, Plan> plans;
  int tuned_shapes = 0;       // shapes this handle had to time itself (not found in the plan store)
  int launches = 0;
  hipStream_t st = nullptr;
  // Deferred split-K combine.  A split-K conv whose only consumer is a GroupNorm the single-launch kernel takes (maps up to
  // 32x32: conv_feature -> groupnorm_merged inside every ResBlock, conv_merged -> the following attention block's GroupNorm;
  // sd/diffusion.py:179 -> 199, 205 -> 294) leaves its partial sums in the slabs; that GroupNorm adds them up itself
  // (GnArgs::slab) and writes the conv's output on the side when somebody else needs it.  One launch and one round trip of
  // the tensor less per pair.  `pend` is checked by EVERY launch helper: whoever comes next and is not that GroupNorm gets
  // the ordinary splitk_finalize first (flush_pending), so the deferral can never be observed.
  bool pend = false;
  bool pend_keep16 = false;            // the deferred conv's fp16-only output has a reader besides the GroupNorm
  GemmArgs pend_args;
  const void* pend_key = nullptr;      // output pointer of the deferred conv
  static bool defer_on() { static const bool on = !(getenv("SDMI_FIN_GN") && atoi(getenv("SDMI_FIN_GN")) == 0); return on; }
  // Largest map (pixels) on which the combine is deferred.  Measured on one MI355X (rocprofv3 per shape, same box,
  // gpurun_out/r3i_A vs r3i_B): the 64-workgroup GroupNorm that also adds the slabs costs 8.3 us at 8x8 (splitk_finalize 4.7 +
  // GroupNorm 4.6 before), 11 - 17 us at 16x16 (12 - 14 before) and 20 - 29 us at 32x32 (16 - 20 before: 80-byte pieces of
  // 3 - 6 slabs through 64 CUs); with every eligible pair deferred the step has 27 launches fewer and is 1.4 % SLOWER
  // (234.0 vs 237.3 steps/s, two interleaved runs each).  Default: 8x8 maps only.
  static int defer_max_px() { static const int v = getenv("SDMI_FIN_GN_MAXPX") ? atoi(getenv("SDMI_FIN_GN_MAXPX")) : 64; return v; }
  int flush_pending() {
    if (!pend) return SDMI_OK;
    pend = false;
    prof_begin(3, 0.0);
    TRY(sdmi_launch_splitk_finalize(pend_args, st));
    prof_end();
    launches += 1;
    log_launch("finalize M=%d N=%d K=%d split=%d", pend_args.M, pend_args.N, pend_args.K, pend_args.ksplit);
    return SDMI_OK;
  }
  // optional per-launch HIP-event profiling (bench roofline): class 0 = MFMA GEMM kernels (igemm / halo conv / back-to-back),
  // 1 = attention, 2 = norms, 3 = splitk_finalize
  bool profiling = false;
  struct ProfRec { hipEvent_t e0, e1; int cls; double flops; };
  std::vector<ProfRec> prof;
  // optional launch log (SDMI_LAUNCH_LOG=<file>): one line per kernel launch of the last forward, in launch order,
  // so a rocprofv3 kernel trace / PMC pass can be joined with the shape each kernel ran (tools/join_trace.py)
  struct LaunchRec { char text[160]; };
  std::vector<LaunchRec> launch_log;
  bool logging = getenv("SDMI_LAUNCH_LOG") != nullptr;
  void log_launch(const char* fmt, ...) {
    if (!logging) return;
    LaunchRec r;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(r.text, sizeof(r.text), fmt, ap);
    va_end(ap);
    launch_log.push_back(r);
  }
  void write_launch_log() {
    if (!logging) return;
    FILE* f = fopen(getenv("SDMI_LAUNCH_LOG"), "w");
    if (!f) return;
    for (const LaunchRec& r : launch_log) fprintf(f, "%s\n", r.text);
    fclose(f);
  }
  void prof_begin(int cls, double flops) {
    if (!profiling) return;
    ProfRec r; r.cls = cls; r.flops = flops;
    (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e0, st);
    prof.push_back(r);
  }
  void prof_end() {
    if (!profiling) return;
    (void)hipEventRecord(prof.back().e1, st);
  }

  Engine() { (void)hipGetDevice(&device); }
  ~Engine() {
    for (void* p : owned) (void)hipFree(p);
    for (void* p : ctx_owned) (void)hipFree(p);
  }
  // buffers sized by the context batch (sdmi_unet_set_context): released and re-made when the batch changes
  std::vector<void*> ctx_owned;
  void free_ctx() {
    for (void* p : ctx_owned) (void)hipFree(p);
    ctx_owned.clear();
    ctx16 = nullptr; ctx32 = nullptr;
    ctxK.clear(); ctxVt.clear();
    xfW1.clear(); xfW2.clear(); xfG.clear(); xfH.clear();
    xf_km = xf_vm = xf_vp = nullptr; xf_kq = nullptr;
  }
  template <class T>
  int dmalloc_ctx(T** out, size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
      sdmi_set_error("hipMalloc(%zu) failed", bytes);
      return SDMI_ENOMEM;
    }
    ctx_owned.push_back(p);
    *out = (T*)p;
    return SDMI_OK;
  }

  // Every entry point that launches work calls this first: the handle's memory lives on `device`, and a launch
  // issued while another device is current would run there on pointers it cannot reach (a GPU fault, not an
  // error code).  The caller must make the handle's device current (hipSetDevice / torch.cuda.device).
  int enter(void* stream) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != device) {
      sdmi_set_error("handle lives on HIP device %d but device %d is current", device, cur);
      return SDMI_EINVAL;
    }
    st = (hipStream_t)stream;
    // a deferred split-K combine never outlives the call that made it (run_stage flushes); if that call FAILED between the
    // conv and its GroupNorm the flag would still be up, and the next entry's first launch would run splitk_finalize on
    // slabs and arena addresses of the failed forward
    pend = false;
    return SDMI_OK;
  }

  template <class T>
  int dmalloc(T** out, size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
      sdmi_set_error("hipMalloc(%zu) failed", bytes);
      return SDMI_ENOMEM;
    }
    owned.push_back(p);
    *out = (T*)p;
    return SDMI_OK;
  }

  // ---- weight lookup / packing --------------------------------------------------------------
  const sdmi_tensor_desc* find(const std::string& name) const {
    auto it = src.find(name);
    return it == src.end() ? nullptr : &it->second;
  }
  bool has(const std::string& name) const { return src.count(name) != 0; }

  int need(const std::string& name, const sdmi_tensor_desc** out, int ndim, std::initializer_list<int64_t> shape) {
    const sdmi_tensor_desc* t = find(name);
    if (!t) { sdmi_set_error("missing tensor '%s'", name.c_str()); return SDMI_ENOENT; }
    if (t->ndim != ndim) { sdmi_set_error("tensor '%s': ndim %d, expected %d", name.c_str(), t->ndim, ndim); return SDMI_EINVAL; }
    int i = 0;
    for (int64_t s : shape) {
      if (s >= 0 && t->shape[i] != s) {
        sdmi_set_error("tensor '%s': dim %d is %lld, expected %lld", name.c_str(), i, (long long)t->shape[i], (long long)s);
        return SDMI_EINVAL;
      }
      ++i;
    }
    *out = t;
    return SDMI_OK;
  }

  int load_vec(const std::string& name, int n, float** out) {
    const sdmi_tensor_desc* t;
    TRY(need(name, &t, 1, {n}));
    TRY(dmalloc(out, (size_t)n * 4));
    TRY(sdmi_launch_cast_any_f32(t->data_dev, t->dtype == SDMI_F32, *out, n, st));
    weight_bytes += (int64_t)n * 4;
    return SDMI_OK;
  }
  int load_norm(const std::string& p, int C, NormW* w) {
    w->C = C;
    TRY(load_vec(p + ".weight", C, &w->gamma));
    TRY(load_vec(p + ".bias", C, &w->beta));
    return SDMI_OK;
  }
  // conv (ks=3/1, 4-D weight) or linear (2-D weight); o_keep < O keeps only the first rows
  int load_conv(const std::string& p, int O, int I, int ks, bool bias, ConvW* w, int o_keep = -1) {
    if (o_keep < 0) o_keep = O;
    const sdmi_tensor_desc* t;
    if (find(p + ".weight") && find(p + ".weight")->ndim == 2) TRY(need(p + ".weight", &t, 2, {O, (int64_t)I * ks * ks}));
    else TRY(need(p + ".weight", &t, 4, {O, I, ks, ks}));
    w->O = o_keep; w->I = I; w->ks = ks;
    const size_t n = (size_t)o_keep * ks * ks * I;
    TRY(dmalloc(&w->w, n * 2));
    TRY(sdmi_launch_pack_conv(t->data_dev, t->dtype == SDMI_F32, w->w, O, I, ks, o_keep, st));
    weight_bytes += (int64_t)n * 2;
    if (bias) {
      const sdmi_tensor_desc* b;
      TRY(need(p + ".bias", &b, 1, {O}));
      TRY(dmalloc(&w->bias, (size_t)o_keep * 4));
      TRY(sdmi_launch_cast_any_f32(b->data_dev, b->dtype == SDMI_F32, w->bias, o_keep, st));
      weight_bytes += (int64_t)o_keep * 4;
    }
    return SDMI_OK;
  }

  // conv_merged and the 1x1 residual_layer share one accumulator: weights side by side along K, biases summed
  int fuse_skip(ResW* r) {
    const size_t k2 = (size_t)9 * r->cout, ks = (size_t)r->cin, kk = k2 + ks;
    TRY(dmalloc(&r->w2s, (size_t)r->cout * kk * 2));
    SDMI_CHECK_HIP(hipMemcpy2DAsync(r->w2s, kk * 2, r->conv2.w, k2 * 2, k2 * 2, r->cout, hipMemcpyDeviceToDevice, st));
    SDMI_CHECK_HIP(hipMemcpy2DAsync(r->w2s + k2, kk * 2, r->skip.w, ks * 2, ks * 2, r->cout, hipMemcpyDeviceToDevice, st));
    TRY(dmalloc(&r->bias2s, (size_t)r->cout * 4));
    TRY(sdmi_launch_add_vec(r->conv2.bias, r->skip.bias, r->bias2s, r->cout, st));
    weight_bytes += (int64_t)r->cout * kk * 2;
    return SDMI_OK;
  }
  int load_res(const std::string& p, int cin, int cout) {
    ResW r;
    r.cin = cin; r.cout = cout;
    TRY(load_norm(p + ".groupnorm_feature", cin, &r.gn1));
    TRY(load_conv(p + ".conv_feature", cout, cin, 3, true, &r.conv1));
    TRY(load_conv(p + ".linear_time", cout, kTime, 1, true, &r.time));
    TRY(load_norm(p + ".groupnorm_merged", cout, &r.gn2));
    TRY(load_conv(p + ".conv_merged", cout, cout, 3, true, &r.conv2));
    r.has_skip = cin != cout;
    if (r.has_skip) TRY(load_conv(p + ".residual_layer", cout, cin, 1, true, &r.skip));
    if (r.has_skip && cin % 64 == 0) TRY(fuse_skip(&r));
    TRY(dmalloc(&r.bias1, (size_t)cout * 4));
    TRY(sdmi_launch_add_vec(r.conv1.bias, r.time.bias, r.bias1, cout, st));
    r.time_off = time_total;
    time_total += cout;
    res[p] = r;
    res_order.push_back(p);
    return SDMI_OK;
  }
  // Composes the attention block's feed-forward + conv_output (sd/diffusion.py:351-381) into one [C][2C] GEMM at load
  // (AttnW::ffn / ffn_f).  All products in fp32 on the device, rounded to fp16 once; temporaries are freed again.
  int load_ffn(const std::string& p, int C, AttnW* a) {
    const sdmi_tensor_desc *w1, *b1, *w2, *b2, *wo, *bo;
    TRY(need(p + ".linear_geglu_1.weight", &w1, 2, {8 * C, C}));
    TRY(need(p + ".linear_geglu_1.bias", &b1, 1, {8 * C}));
    TRY(need(p + ".linear_geglu_2.weight", &w2, 2, {C, 4 * C}));
    TRY(need(p + ".linear_geglu_2.bias", &b2, 1, {C}));
    if (find(p + ".conv_output.weight") && find(p + ".conv_output.weight")->ndim == 2) TRY(need(p + ".conv_output.weight", &wo, 2, {C, C}));
    else TRY(need(p + ".conv_output.weight", &wo, 4, {C, C, 1, 1}));
    TRY(need(p + ".conv_output.bias", &bo, 1, {C}));
    const bool w1f = w1->dtype == SDMI_F32, w2f = w2->dtype == SDMI_F32, wof = wo->dtype == SDMI_F32;
    float *t1 = nullptr, *wf = nullptr, *vec = nullptr;     // T1 = Wo W2 [C][4C], Wf = T1 W1a [C][C], 4 vectors
    SDMI_CHECK_HIP(hipMalloc((void**)&t1, (size_t)C * 4 * C * 4));
    SDMI_CHECK_HIP(hipMalloc((void**)&wf, (size_t)C * C * 4));
    SDMI_CHECK_HIP(hipMalloc((void**)&vec, (size_t)(4 * C + 3 * C) * 4));
    float *b1a = vec, *b2f = vec + 4 * C, *bof = b2f + C, *bt = bof + C;
    auto done = [&](int rc) { (void)hipStreamSynchronize(st); (void)hipFree(t1); (void)hipFree(wf); (void)hipFree(vec); return rc; };
    int rc;
    // quirk Q2: only rows [0, 4C) of linear_geglu_1 (the un-gated half) reach the output
    if ((rc = sdmi_launch_cast_any_f32(b1->data_dev, b1->dtype == SDMI_F32, b1a, 4 * C, st)) != SDMI_OK) return done(rc);
    if ((rc = sdmi_launch_cast_any_f32(b2->data_dev, b2->dtype == SDMI_F32, b2f, C, st)) != SDMI_OK) return done(rc);
    if ((rc = sdmi_launch_cast_any_f32(bo->data_dev, bo->dtype == SDMI_F32, bof, C, st)) != SDMI_OK) return done(rc);
    if ((rc = sdmi_launch_compose_linear(wo->data_dev, wof, w2->data_dev, w2f, t1, 1, C, C, 4 * C, 4 * C, st)) != SDMI_OK) return done(rc);
    if ((rc = sdmi_launch_compose_linear(t1, 1, w1->data_dev, w1f, wf, 1, C, 4 * C, C, C, st)) != SDMI_OK) return done(rc);
    ConvW& f = a->ffn;
    f.O = C; f.I = 2 * C; f.ks = 1;
    if ((rc = dmalloc(&f.w, (size_t)C * 2 * C * 2)) != SDMI_OK) return done(rc);
    if ((rc = dmalloc(&f.bias, (size_t)C * 4)) != SDMI_OK) return done(rc);
    // bf = T1 b1a + (Wo b2 + bo)
    if ((rc = sdmi_launch_compose_bias(wo->data_dev, wof, b2f, bof, bt, C, C, st)) != SDMI_OK) return done(rc);
    if ((rc = sdmi_launch_compose_bias(t1, 1, b1a, bt, f.bias, C, 4 * C, st)) != SDMI_OK) return done(rc);
    if ((rc = sdmi_launch_cast_rows(wf, 1, f.w, C, C, 2 * C, st)) != SDMI_OK) return done(rc);
    if ((rc = sdmi_launch_cast_rows(wo->data_dev, wof, f.w + C, C, C, 2 * C, st)) != SDMI_OK) return done(rc);
    // layernorm_3 folded into the Wf half
    FoldW& ff = a->ffn_f;
    f16* wfold = nullptr;
    if ((rc = dmalloc(&ff.w, (size_t)C * 2 * C * 2)) != SDMI_OK) return done(rc);
    if ((rc = dmalloc(&ff.g, (size_t)C * 4)) != SDMI_OK) return done(rc);
    if ((rc = dmalloc(&ff.h, (size_t)C * 4)) != SDMI_OK) return done(rc);
    if (hipMalloc((void**)&wfold, (size_t)C * C * 2) != hipSuccess) { sdmi_set_error("hipMalloc failed"); return done(SDMI_ENOMEM); }
    rc = sdmi_launch_ln_fold_prep(wf, 1, a->ln3.gamma, a->ln3.beta, f.bias, wfold, ff.g, ff.h, C, C, st);
    if (rc == SDMI_OK && hipMemcpy2DAsync(ff.w, (size_t)2 * C * 2, wfold, (size_t)C * 2, (size_t)C * 2, C, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = SDMI_EHIP;
    if (rc == SDMI_OK) rc = sdmi_launch_cast_rows(wo->data_dev, wof, ff.w + C, C, C, 2 * C, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(wfold);
    weight_bytes += (int64_t)2 * C * 2 * C * 2 + (int64_t)3 * C * 4;
    return done(rc);
  }
  // first N rows of Linear `p` ([O][C], O >= N) with LayerNorm `ln` folded in
  int load_fold(const std::string& p, int N, int C, const NormW& ln, const float* bias, FoldW* f) {
    const sdmi_tensor_desc* t;
    TRY(need(p + ".weight", &t, 2, {-1, C}));
    if (t->shape[0] < N) { sdmi_set_error("tensor '%s.weight': %lld rows, need %d", p.c_str(), (long long)t->shape[0], N); return SDMI_EINVAL; }
    TRY(dmalloc(&f->w, (size_t)N * C * 2));
    TRY(dmalloc(&f->g, (size_t)N * 4));
    TRY(dmalloc(&f->h, (size_t)N * 4));
    TRY(sdmi_launch_ln_fold_prep(t->data_dev, t->dtype == SDMI_F32, ln.gamma, ln.beta, bias, f->w, f->g, f->h, N, C, st));
    weight_bytes += (int64_t)N * C * 2 + (int64_t)N * 8;
    return SDMI_OK;
  }
  int load_attn(const std::string& p, int heads, int dh) {
    AttnW a;
    const int C = heads * dh;
    a.C = C; a.dh = dh;
    TRY(load_norm(p + ".groupnorm", C, &a.gn));
    TRY(load_conv(p + ".conv_input", C, C, 1, true, &a.conv_in));
    TRY(load_norm(p + ".layernorm_1", C, &a.ln1));
    TRY(load_conv(p + ".attention_1.in_proj", 3 * C, C, 1, false, &a.in_proj));
    TRY(load_conv(p + ".attention_1.out_proj", C, C, 1, true, &a.out1));
    TRY(load_norm(p + ".layernorm_2", C, &a.ln2));
    TRY(load_conv(p + ".attention_2.q_proj", C, C, 1, false, &a.q));
    TRY(load_conv(p + ".attention_2.k_proj", C, kCtx, 1, false, &a.k));
    TRY(load_conv(p + ".attention_2.v_proj", C, kCtx, 1, false, &a.v));
    TRY(load_conv(p + ".attention_2.out_proj", C, C, 1, true, &a.out2));
    TRY(load_norm(p + ".layernorm_3", C, &a.ln3));
    TRY(load_ffn(p, C, &a));
    TRY(load_fold(p + ".attention_1.in_proj", 3 * C, C, a.ln1, nullptr, &a.in_proj_f));
    TRY(load_fold(p + ".attention_2.q_proj", C, C, a.ln2, nullptr, &a.q_f));
    static const int xf_min_c = getenv("SDMI_XATTN_MINC") ? atoi(getenv("SDMI_XATTN_MINC")) : 640;
    if (C >= xf_min_c) {
      const sdmi_tensor_desc* t;
      TRY(need(p + ".attention_2.q_proj.weight", &t, 2, {C, C}));
      TRY(dmalloc(&a.wqT, (size_t)C * C * 2));
      TRY(sdmi_launch_transpose_scale(t->data_dev, t->dtype == SDMI_F32, a.wqT, C, C, q_scale(dh), st));
      weight_bytes += (int64_t)C * C * 2;
    }
    a.ctx_idx = (int)attn_order.size();
    attn[p] = a;
    attn_order.push_back(p);
    return SDMI_OK;
  }
  int load_plain_conv(const std::string& p, int cin, int cout) {
    ConvW c;
    TRY(load_conv(p, cout, cin, 3, true, &c));
    convs[p] = c;
    return SDMI_OK;
  }
  // the conv of an Upsample layer (sd/diffusion.py:426-435), additionally as four 2x2 phase convs on the source grid
  int load_ups_conv(const std::string& p, int C) {
    TRY(load_plain_conv(p, C, C));
    return pack_ups_phase(p, &convs[p]);
  }
  int pack_ups_phase(const std::string& p, ConvW* c) {
    const sdmi_tensor_desc* t;
    TRY(need(p + ".weight", &t, 4, {c->O, c->I, 3, 3}));
    TRY(dmalloc(&c->w4, (size_t)16 * c->O * c->I * 2));
    TRY(sdmi_launch_pack_ups_phase(t->data_dev, t->dtype == SDMI_F32, c->w4, c->O, c->I, st));
    weight_bytes += (int64_t)16 * c->O * c->I * 2;
    return SDMI_OK;
  }

  // ---- activations ---------------------------------------------------------------------------
  // The arena is sized by the forward that is about to run (unet.hip unet_forward_impl), not by a process-wide knob read once:
  // a handle made for single prompts grows when a batched chain (pipeline.generate_batch, UNet batch 2P) first comes through
  // it.  Growing releases the old block first (hipFree waits for the work in flight that still reads it); addresses stay
  // deterministic per (batch, size) because the forward restarts its bump pointer at 0 either way.
  int ensure_arena(size_t need) {
    if (need <= arena.cap) return SDMI_OK;
    if (arena.base) {
      for (size_t i = 0; i < owned.size(); ++i)
        if (owned[i] == (void*)arena.base) { owned.erase(owned.begin() + i); break; }
      (void)hipFree(arena.base);
      arena.base = nullptr; arena.cap = 0;
    }
    const size_t cap = (need + ((size_t)1 << 30) - 1) >> 30 << 30;          // whole GiB
    TRY(dmalloc(&arena.base, cap));
    arena.cap = cap; arena.off = 0;
    return SDMI_OK;
  }
  int new_act(int B, int H, int W, int C, bool is_stream, Act* a) {
    a->B = B; a->H = H; a->W = W; a->C = C;
    const size_t n = (size_t)B * H * W * C;
    a->h = (f16*)arena.alloc(n * 2);
    a->f = nullptr;
    const bool want_f = (is_stream && stream_f32) || accurate;
    if (want_f) a->f = (float*)arena.alloc(n * 4);
    if (!a->h || (want_f && !a->f)) { sdmi_set_error("activation arena exhausted (%zu MiB; SDMI_ARENA_GB sets a larger minimum)", arena.cap >> 20); return SDMI_ENOMEM; }
    a->grec = nullptr; a->gok = false; a->gT = a->gparts = 0;
    if (is_stream) attach_gacc(a);
    return SDMI_OK;
  }

  // ---- GEMM with per-shape plan -----------------------------------------------------------------
  // rs != null: ask the epilogue for LayerNorm row statistics of the output (granted when the plan has ksplit == 1)
  // plan of a GEMM shape: this handle's table, else the plan store (shipped table / per-library cache), else tuned now
  int plan_of(GemmArgs& a, std::map<ShapeKey, Plan>::iterator* out) {
    a.zero = zero;
    a.slab = slab;
    ShapeKey key(a.M, a.N, a.K, a.ks + 16 * a.pad + 64 * (a.X0 != 0) + 128 * (a.ln_stat != nullptr) + 256 * (a.out_f32 != 0) + 512 * (a.res != nullptr) + 2048 * (a.img_rows != 0),
                 a.stride, a.ups, a.C0, a.C1 + 4096 * (a.lda0 != 0 || a.ldw != 0) + 8192 * a.act, a.Wo, a.outT ? a.nt0 + 1 : 0);
    auto it = plans.find(key);
    // a lane's heuristic stand-in for a shape its parent had not tuned yet: look the store up again, the parent may have by now
    if (it != plans.end() && it->second.provisional) { plans.erase(it); it = plans.end(); }
    if (it == plans.end()) {
      Plan pl;
      if (tune) {
        PlanStore& ps = plan_store();
        ps.load();
        bool have = false;
        std::string st_name;
        int st_ks = 1;
        float st_us = 0.f;
        if (ps.lookup(key, &st_name, &st_ks, &st_us)) {          // stored plan: valid only if this build still has the config ...
          for (int c = 0; c < sdmi_gemm_num_cfgs() && !have; ++c)
            if (st_name == sdmi_gemm_cfg_name(c) && sdmi_gemm_cfg_applicable(a, c)) {
              pl.cfg = c; pl.ksplit = st_ks; pl.us = st_us; have = true;
            }
          // ... and its split-K is one the tuner itself could have chosen for this engine: the slabs must fit (the table is
          // shared by engines with different slab sizes, and a hand-edited line must not write past the slab)
          if (have && !ksplit_ok(a, pl.cfg, pl.ksplit)) have = false;
        }
        // a LANE (sdmi_unet_clone) never times anything: its kernels share the GPU with the other lanes' streams, the cold-L2
        // timings would be distorted and then persisted for every later process.  It runs what its parent tuned (copied at
        // clone time, or found in the store above once the parent has appended it), else the heuristic tile -- unrecorded and
        // PROVISIONAL: the next call for this shape asks the store again, so the lane converges on its parent's plan
        if (!have && !is_lane) {
          TRY(tune_gemm(a, &pl));
          ++tuned_shapes;
          if (pl.cfg >= 0) ps.append(key, sdmi_gemm_cfg_name(pl.cfg), pl.ksplit, pl.us);
        } else if (!have) {
          pl.provisional = true;
        }
      }
      it = plans.emplace(key, pl).first;
    }
    *out = it;
    return SDMI_OK;
  }

  // defer: a split-K launch leaves its partial sums in the slabs for the GroupNorm that follows (see `pend`)
  // yact: the output tensor; when it carries an accumulator (Act::gacc) the launch also takes the GroupNorm statistics of
  // what it writes -- in the one-pass epilogue or in the split-K combine -- if the plan's tile allows it (yact->gok says so)
  int gemm(GemmArgs a, RowStat* rs = nullptr, bool defer = false, Act* yact = nullptr) {
    TRY(flush_pending());
    std::map<ShapeKey, Plan>::iterator it = plans.end();
    int plan_cfg = -1;
    if (accurate) {
      a.zero = zero; a.slab = slab; a.accurate = 1;
      if (!a.a0f || (a.C1 && !a.a1f) || (a.X0 && !a.x0f) || (a.X1 && !a.x1f)) { sdmi_set_error("accurate mode: a GEMM operand has no fp32 copy (M=%d N=%d K=%d)", a.M, a.N, a.K); return SDMI_EINVAL; }
      int ks = 1;
      plan_cfg = sdmi_gemm_pick_acc_cfg(a, &ks);
      while (ks > 1 && (size_t)ks * a.M * a.N * 4 > slab_bytes) ks /= 2;
      a.ksplit = ks;
    } else {
      TRY(plan_of(a, &it));
      a.ksplit = it->second.ksplit;
      plan_cfg = it->second.cfg;
    }
    defer = defer && defer_on() && a.ksplit > 1 && !a.outT && !a.act && !a.phase2 && a.cs_hi == 0 && !a.ln_stat && !a.rowstat &&
            a.ldc == a.N && (!a.res || a.ldr == a.N) && a.Ho * a.Wo <= defer_max_px();
    a.no_finalize = 1;                     // the combine is this function's own launch (timed as its own class) or deferred
    if (a.ln_stat && ln_guard) { a.ln_guard = ln_guard; a.ln_guard_thr2 = ln_guard_thr() * ln_guard_thr(); }
    memset(&a.gacc, 0, sizeof(a.gacc));
    if (yact && yact->grec && !defer && plan_cfg >= 0) {
      set_gacc(a, *yact);
      bool ok;
      if (a.ksplit > 1) { a.gacc.parts = 1; ok = sdmi_finalize_gacc_ok(a, &a.gacc.T); }
      else { a.gacc.parts = 2; ok = sdmi_gemm_gacc_ok(a, plan_cfg); if (ok) a.gacc.T = sdmi_gemm_gacc_T(a, plan_cfg); }
      if (!ok) memset(&a.gacc, 0, sizeof(a.gacc));
    }
    if (rs) {
      const bool no_fold = !ln_fold_on;
      rs->ptr = nullptr;
      if (!no_fold && a.ksplit == 1 && !a.outT && plan_cfg >= 0 && plan_cfg < sdmi_gemm_num_plain_cfgs()) {
        int bm, bn;
        sdmi_gemm_cfg_dims(plan_cfg, &bm, &bn);
        const int ntn = (a.N + bn - 1) / bn;
        float* buf = (float*)arena.alloc((size_t)a.M * ntn * 8);
        if (!buf) { sdmi_set_error("activation arena exhausted"); return SDMI_ENOMEM; }
        a.rowstat = buf;
        rs->ptr = buf; rs->ntn = ntn;
      }
    }
    if (it != plans.end()) {
      it->second.calls += 1;
      it->second.flops = 2.0 * a.M * a.N * a.K;
    }
    int ks_eff = 1, kper = 0;
    prof_begin(0, 2.0 * a.M * a.N * a.K);
    TRY(sdmi_launch_gemm(a, plan_cfg, st, &ks_eff, &kper));
    prof_end();
    const bool deferred = defer && ks_eff > 1;
    // (the launcher may lower the split-K factor: statistics taken by the combine need ks_eff > 1, by the epilogue == 1)
    if (yact) {
      yact->gok = a.gacc.rec != nullptr && (a.ksplit > 1) == (ks_eff > 1) && (ks_eff <= 1 || ks_eff == a.ksplit);
      yact->gT = a.gacc.T; yact->gparts = a.gacc.parts;
    }
    if (ks_eff > 1) {
      GemmArgs f = a;
      f.ksplit = ks_eff; f.no_finalize = 0;
      f.ksteps_per = kper;                 // as the launcher split K (splitk_finalize needs it for the partial LayerNorm fold)
      if (deferred) { pend = true; pend_args = f; pend_key = a.out; }
      else {
        prof_begin(3, 0.0);
        TRY(sdmi_launch_splitk_finalize(f, st));
        prof_end();
      }
    }
    const int nl = (ks_eff > 1 && !deferred) ? 2 : 1;
    launches += nl;
    if (logging) {
      const int cfg = plan_cfg;
      const bool halo = cfg >= sdmi_gemm_num_plain_cfgs();
      log_launch("%s M=%d N=%d K=%d ks=%d s=%d up=%d cfg=%s%s split=%d flops=%.0f wbytes=%.0f out32=%d res=%d", halo ? "halo" : "igemm", a.M, a.N,
                 a.K, a.ks, a.stride, a.ups, cfg >= 0 ? sdmi_gemm_cfg_name(cfg) : "heur", a.hgn.x0 ? "+gn" : "", a.ksplit, 2.0 * a.M * a.N * a.K,
                 2.0 * a.N * a.K, a.out_f32, a.res != nullptr);
      if (ks_eff > 1 && !deferred) log_launch("finalize M=%d N=%d K=%d split=%d", a.M, a.N, a.K, a.ksplit);
    }
    return SDMI_OK;
  }

  // SDMI_TUNE_LOG: per-shape plan table (isolated tuned time x calls) on stderr, e.g. at destroy
  void plan_report(const char* who) const {
    if (!getenv("SDMI_TUNE_LOG")) return;
    double tot = 0.0;
    for (const auto& kv : plans) tot += (double)kv.second.us * kv.second.calls;
    fprintf(stderr, "[sdmi plans] %s: %zu shapes, sum(tuned us x calls) = %.1f us\n", who, plans.size(), tot);
    for (const auto& kv : plans) {
      const Plan& pl = kv.second;
      if (pl.calls == 0 || pl.cfg < 0) continue;
      fprintf(stderr, "[sdmi plans]   M=%-5d N=%-5d K=%-5d ks=%d s=%d up=%d  %-18s split %2d  %7.1f us x %4ld = %8.1f us  %6.1f TF/s  (%.2f GFLOP)\n",
              std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first) & 15,
              std::get<4>(kv.first), std::get<5>(kv.first), sdmi_gemm_cfg_name(pl.cfg), pl.ksplit, pl.us, pl.calls, pl.us * pl.calls, pl.flops / pl.us * 1e-6, pl.flops * 1e-9);
    }
  }

  // the tuner's admission rule for a split-K factor (also applied to plans read from a table)
  bool ksplit_ok(const GemmArgs& a0, int cfg, int ks) const {
    if (ks < 1 || ks > 16) return false;
    if (ks == 1) return true;
    int bm, bn;
    sdmi_gemm_cfg_dims(cfg, &bm, &bn);
    const int nkt = a0.K / 64;
    const int tiles = ((a0.M + bm - 1) / bm) * ((a0.N + bn - 1) / bn);
    if (tiles * ks > 1024 || nkt / ks < 4) return false;
    if ((size_t)ks * a0.M * a0.N * 4 > slab_bytes) return false;
    if (a0.ln_stat && !(a0.ln_ksteps > 0 && a0.ln_out && nkt % ks == 0 && a0.ln_ksteps % (nkt / ks) == 0 && cfg < sdmi_gemm_num_plain_cfgs())) return false;
    if ((a0.img_rows && !a0.phase2) || a0.act == 2) return false;       // these epilogues live in the one-pass path only
    return true;
  }

  static constexpr size_t kThrashBytes = (size_t)64 << 20;
  char* thrash = nullptr;
  int tune_gemm(const GemmArgs& a0, Plan* best) {
    TRY(flush_pending());           // the candidates' split-K launches write the slabs a deferred conv may still own
    if (!thrash && !getenv("SDMI_TUNE_WARM")) TRY(dmalloc(&thrash, kThrashBytes));
    hipEvent_t e0, e1;
    SDMI_CHECK_HIP(hipEventCreate(&e0));
    SDMI_CHECK_HIP(hipEventCreate(&e1));
    float best_us = 1e30f;
    for (int cfg = 0; cfg < sdmi_gemm_num_cfgs(); ++cfg) {
      if (!sdmi_gemm_cfg_applicable(a0, cfg)) continue;
      for (int ks : {1, 2, 3, 4, 5, 6, 8, 10, 12, 16}) {
        if (!ksplit_ok(a0, cfg, ks)) continue;
        GemmArgs a = a0;
        a.ksplit = ks;
        float us = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
          // Time every candidate with COLD L2s: in the step a GEMM's activations were just written by another kernel
          // (other XCDs' L2 / Infinity Cache) and its weights come from HBM, while back-to-back repetitions of one
          // launch keep both L2-warm and mis-rank the configs (measured: a plan that won warm at 44 us ran 57 us in
          // the step where the warm-loser ran 45 us).  A 64 MiB fill between repetitions evicts the 8 x 4 MiB L2s.
          if (thrash) SDMI_CHECK_HIP(hipMemsetAsync(thrash, rep, kThrashBytes, st));
          SDMI_CHECK_HIP(hipEventRecord(e0, st));
          int rc = sdmi_launch_gemm(a, cfg, st);
          if (rc != SDMI_OK) return rc;
          SDMI_CHECK_HIP(hipEventRecord(e1, st));
          SDMI_CHECK_HIP(hipEventSynchronize(e1));
          float ms = 0.f;
          SDMI_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
          if (rep > 0 && ms * 1e3f < us) us = ms * 1e3f;
        }
        if (us < best_us) { best_us = us; best->cfg = cfg; best->ksplit = ks; best->us = us; }
      }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (getenv("SDMI_TUNE_LOG"))
      fprintf(stderr, "[sdmi tune] M=%d N=%d K=%d ks=%d s=%d up=%d -> %s split %d  %.1f us  (%.1f TF/s)\n", a0.M, a0.N,
              a0.K, a0.ks, a0.stride, a0.ups, sdmi_gemm_cfg_name(best->cfg), best->ksplit, best->us,
              2.0 * a0.M * a0.N * a0.K / best->us * 1e-6);
    return SDMI_OK;
  }

  static GemmArgs base_args(const Act& x, const Act* x1, const ConvW& w, int Ho, int Wo, int stride, int ups) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.a0 = x.h; a.a0f = x.f; a.C0 = x.C;
    if (x1) { a.a1 = x1->h; a.a1f = x1->f; a.C1 = x1->C; }
    a.Hs = x.H; a.Ws = x.W; a.Ho = Ho; a.Wo = Wo;
    a.ups = ups; a.stride = stride; a.ks = w.ks; a.pad = w.ks == 3 ? 1 : 0;
    a.M = x.B * Ho * Wo; a.N = w.O; a.K = w.ks * w.ks * (a.C0 + a.C1);
    a.w = w.w; a.bias = w.bias;
    a.ksplit = 1;
    return a;
  }
  void set_out(GemmArgs& a, const Act& y) const {
    if (y.f) { a.out = y.f; a.out_f32 = 1; a.out16 = y.h; }
    else { a.out = y.h; a.out_f32 = 0; a.out16 = nullptr; }
    a.ldc = y.C;
  }
  static void set_res(GemmArgs& a, const Act& r) {
    if (r.f) { a.res = r.f; a.res_f32 = 1; } else { a.res = r.h; a.res_f32 = 0; }
    a.ldr = r.C;
  }

  int groupnorm(const Act& x, const Act* x1, const NormW& w, float eps, int silu, Act* y) {
    const int C = x.C + (x1 ? x1->C : 0);
    if (w.C != C) { sdmi_set_error("groupnorm: weight C=%d vs input C=%d", w.C, C); return SDMI_EINVAL; }
    TRY(new_act(x.B, x.H, x.W, C, false, y));
    GnArgs g;
    memset(&g, 0, sizeof(g));
    const bool f32 = x.f != nullptr;
    g.x0 = f32 ? (const void*)x.f : (const void*)x.h;
    if (x1) g.x1 = f32 ? (const void*)x1->f : (const void*)x1->h;
    g.in_f32 = f32; g.C0 = x.C; g.C1 = x1 ? x1->C : 0;
    g.B = x.B; g.P = x.H * x.W;
    g.gamma = w.gamma; g.beta = w.beta; g.eps = eps; g.silu = silu;
    g.y = y->h; g.y32 = accurate ? y->f : nullptr; g.partial = gn_partial; g.nchunk = sdmi_gn_nchunk(g.P);
    // x is the output of a split-K conv that has not been combined yet: combine + normalise in one launch
    // statistics already taken by the producers' epilogues: one normalising pass (maps the single-launch kernel handles in
    // 64 workgroups keep it below gacc_min_px: attach_gacc)
    if (x.gok && x.grec && (!x1 || (x1->gok && x1->grec)) && !pend) {
      g.acc0 = x.grec; g.accT0 = x.gT; g.accP0 = x.gparts; g.atom = kGnAtom;
      if (x1) { g.acc1 = x1->grec; g.accT1 = x1->gT; g.accP1 = x1->gparts; }
    }
    const bool from_slab = pend && !x1 && pend_key == (f32 ? (const void*)x.f : (const void*)x.h) && sdmi_gn_launches(g) == 1 &&
                           pend_args.M == x.B * g.P && pend_args.N == C;
    if (from_slab) {
      pend = false;
      g.slab = pend_args.slab; g.ksplit = pend_args.ksplit; g.sbias = pend_args.bias;
      g.sres = pend_args.res; g.sres_f32 = pend_args.res_f32;
      if (pend_args.out_f32) { g.sout = (float*)pend_args.out; g.sout16 = pend_args.out16; }
      else if (pend_keep16) g.sout16 = (f16*)pend_args.out;        // fp16-only output somebody else still reads
    } else {
      TRY(flush_pending());
    }
    prof_begin(2, 0.0);
    TRY(sdmi_launch_groupnorm(g, st));
    prof_end();
    launches += sdmi_gn_launches(g);
    if (g.acc0) log_launch("gn_apply_acc C=%d P=%d B=%d silu=%d", C, g.P, g.B, silu);
    else if (from_slab) log_launch("gn_fused_slab C=%d P=%d B=%d silu=%d split=%d", C, g.P, g.B, silu, g.ksplit);
    else if (sdmi_gn_launches(g) == 1) log_launch("gn_fused C=%d P=%d B=%d silu=%d", C, g.P, g.B, silu);
    else { log_launch("gn_stats C=%d P=%d B=%d", C, g.P, g.B); log_launch("gn_apply C=%d P=%d B=%d silu=%d", C, g.P, g.B, silu); }
    return SDMI_OK;
  }
  int gn_stats(const Act& x, const Act* x1) {
    TRY(flush_pending());
    GnArgs g;
    memset(&g, 0, sizeof(g));
    const bool f32 = x.f != nullptr;
    g.x0 = f32 ? (const void*)x.f : (const void*)x.h;
    if (x1) g.x1 = f32 ? (const void*)x1->f : (const void*)x1->h;
    g.in_f32 = f32; g.C0 = x.C; g.C1 = x1 ? x1->C : 0;
    g.B = x.B; g.P = x.H * x.W;
    g.partial = gn_partial; g.nchunk = sdmi_gn_nchunk(g.P);
    prof_begin(2, 0.0);
    TRY(sdmi_launch_gn_stats(g, st));
    prof_end();
    launches += 1;
    log_launch("gn_stats C=%d P=%d B=%d", g.C0 + g.C1, g.P, g.B);
    return SDMI_OK;
  }

  int layernorm(const Act& x, const NormW& w, Act* y) {
    TRY(flush_pending());
    TRY(new_act(x.B, x.H, x.W, x.C, false, y));
    LnArgs l;
    memset(&l, 0, sizeof(l));
    l.x = x.f ? (const void*)x.f : (const void*)x.h;
    l.in_f32 = x.f != nullptr;
    l.M = x.M(); l.C = x.C; l.gamma = w.gamma; l.beta = w.beta; l.eps = 1e-5f; l.y = y->h; l.y32 = accurate ? y->f : nullptr;
    prof_begin(2, 0.0);
    TRY(sdmi_launch_layernorm(l, st));
    prof_end();
    launches += 1;
    log_launch("layernorm M=%d C=%d", l.M, l.C);
    return SDMI_OK;
  }

  // GroupNorm -> SiLU -> conv3x3 as ONE launch (gemm.hip conv3_halo_kernel<.., GN>, GemmArgs::hgn): taken when the conv's plan is a
  // halo-reuse tile built with the variant and the raw tensor(s) arrive with their producers' statistics records -- the
  // normalising launch (gn_apply: one read of the fp32 map + one write of the fp16 one, 7 - 22 us at 32x32 / 64x64) and its
  // intermediate are gone; the conv's producer waves normalise the pieces of the halo image they stage anyway.
  // `a` describes the conv on the MATERIALISED normalised tensor (C0 = all channels, C1 = 0: that is also its plan key); on
  // success a.hgn is filled and the caller must not launch the GroupNorm.  SDMI_HALO_GN=0: never.
  // opt-in (SDMI_HALO_GN=1): measured 0.7 - 1.5 % SLOWER per step than the separate GroupNorm launch (DESIGN.md section 8.2:
  // each of the N / BN column tiles re-normalises the same input and SiLU's two transcendentals are quarter rate)
  static bool halo_gn_on() { static const bool on = getenv("SDMI_HALO_GN") && atoi(getenv("SDMI_HALO_GN")) != 0; return on; }
  bool try_halo_gn(GemmArgs& a, const Act& x, const Act* x1, const NormW& w, float eps, int silu) {
    if (!halo_gn_on() || accurate || pend) return false;
    if (!x.gok || !x.grec || (x1 && (!x1->gok || !x1->grec))) return false;
    if (x1 && ((x.f != nullptr) != (x1->f != nullptr))) return false;
    std::map<ShapeKey, Plan>::iterator it;
    GemmArgs probe = a;
    if (plan_of(probe, &it) != SDMI_OK || it->second.cfg < 0) return false;
    HaloGn g;
    memset(&g, 0, sizeof(g));
    g.in_f32 = x.f != nullptr;
    g.x0 = g.in_f32 ? (const void*)x.f : (const void*)x.h; g.C0 = x.C;
    if (x1) { g.x1 = g.in_f32 ? (const void*)x1->f : (const void*)x1->h; g.C1 = x1->C; }
    g.gamma = w.gamma; g.beta = w.beta; g.eps = eps; g.silu = silu;
    g.rec0 = x.grec; g.T0 = x.gT; g.P0 = x.gparts; g.atom = kGnAtom;
    if (x1) { g.rec1 = x1->grec; g.T1 = x1->gT; g.P1 = x1->gparts; }
    probe.hgn = g;
    probe.zero = zero;
    if (!sdmi_gemm_hgn_ok(probe, it->second.cfg)) return false;
    a.hgn = g;
    return true;
  }

  // ---- blocks -------------------------------------------------------------------------------
  // UNET_ResidualBlock (sd/diffusion.py:145-209).  bias1 = conv_feature.bias + linear_time(silu(time))
  // y_to_gn: the caller guarantees that the next launch on y is a single-source GroupNorm (an attention block follows)
  int res_block(const ResW& r, const Act& x, const Act* x1, const float* bias1, Act* y, bool y_to_gn = false) {
    const int cin = x.C + (x1 ? x1->C : 0);
    if (cin != r.cin) { sdmi_set_error("res_block: cin %d vs %d", cin, r.cin); return SDMI_EINVAL; }
    Act t0, h, t1, sk;
    TRY(new_act(x.B, x.H, x.W, r.cout, false, &h));
    {
      // GroupNorm -> SiLU -> conv_feature (sd/diffusion.py:173-179): the norm writes the normalised fp16 tensor (the virtual
      // concat of x | x1 materialises here), the conv reads it.  (A conv that normalised its own A operand in LDS,
      // conv3_gn_kernel, was built in round 2 and lost to this pair on every 512x512 shape: removed in round 3.)
      // (round 5: where the plan is a halo tile and x | x1 carry statistics records, the conv normalises its own operand: try_halo_gn)
      Act shape0 = x;                          // the conv as planned: on the materialised concat, cin channels
      shape0.C = cin;
      GemmArgs a = base_args(shape0, nullptr, r.conv1, x.H, x.W, 1, 0);
      if (!try_halo_gn(a, x, x1, r.gn1, 1e-5f, 1)) {
        TRY(groupnorm(x, x1, r.gn1, 1e-5f, 1, &t0));
        a = base_args(t0, nullptr, r.conv1, x.H, x.W, 1, 0);
      }
      a.bias = bias1; a.out = h.h; a.ldc = h.C;
      if (accurate) set_out(a, h);             // groupnorm_merged reads the fp32 values
      pend_keep16 = false;                     // h has one reader: groupnorm_merged
      attach_gacc(&h);
      TRY(gemm(a, nullptr, /*defer=*/true, &h));
    }
    TRY(new_act(x.B, x.H, x.W, r.cout, true, y));
    if (r.has_skip && !r.w2s) {
      TRY(new_act(x.B, x.H, x.W, r.cout, true, &sk));
      GemmArgs s = base_args(x, x1, r.skip, x.H, x.W, 1, 0);
      if (sk.f) { s.out = sk.f; s.out_f32 = 1; } else { s.out = sk.h; }
      s.ldc = sk.C;
      TRY(gemm(s));
    }
    GemmArgs a = base_args(h, nullptr, r.conv2, x.H, x.W, 1, 0);
    // (blocks with a fused skip segment run the generic kernel, which re-stages its A tile per tap: they keep the GroupNorm launch)
    if ((r.has_skip && r.w2s) || !try_halo_gn(a, h, nullptr, r.gn2, 1e-5f, 1)) {
      TRY(groupnorm(h, nullptr, r.gn2, 1e-5f, 1, &t1));
      a = base_args(t1, nullptr, r.conv2, x.H, x.W, 1, 0);
    }
    if (r.has_skip && r.w2s) {                 // the 1x1 skip conv as an extra K-range of conv_merged (sd/diffusion.py:143,209)
      a.x0 = x.h; a.x0f = x.f; a.X0 = x.C;
      if (x1) { a.x1 = x1->h; a.x1f = x1->f; a.X1 = x1->C; }
      a.K += a.X0 + a.X1;
      a.w = r.w2s; a.bias = r.bias2s;
    } else if (r.has_skip) {
      set_res(a, sk);
    } else {
      set_res(a, x);
    }
    set_out(a, *y);
    pend_keep16 = true;
    TRY(gemm(a, nullptr, /*defer=*/y_to_gn, y));
    return SDMI_OK;
  }

  int attention(const f16* q, int ldq, const f16* k, int ldk, int kbs, const f16* vt, int ldvt, f16* o, int ldo, int B,
                int d, int Sq, int Skv, float* o32 = nullptr) {
    TRY(flush_pending());
    AttnArgs t;
    memset(&t, 0, sizeof(t));
    t.q = q; t.ldq = ldq; t.k = k; t.ldk = ldk; t.k_batch_stride = kbs; t.vt = vt; t.ldvt = ldvt;
    t.o = o; t.o32 = o32; t.ldo = ldo; t.B = B; t.H = kHeads; t.d = d; t.Sq = Sq; t.Skv = Skv; t.zero = zero; t.ones = zero + 1024;
    t.scale = 1.f / sqrtf((float)d); t.prescaled = 1;       // callers fold scale*log2(e) into the Q projection (q_scale)
    prof_begin(1, 4.0 * B * kHeads * (double)Sq * Skv * d);
    TRY(sdmi_launch_attention(t, st));
    prof_end();
    launches += 1;
    log_launch("attn B=%d H=%d d=%d Sq=%d Skv=%d flops=%.0f", B, kHeads, d, Sq, Skv, 4.0 * B * kHeads * (double)Sq * Skv * d);
    return SDMI_OK;
  }

  // LayerNorm -> Linear with the norm folded into the GEMM: A = raw stream (fp16 shadow), row statistics from the
  // producer's epilogue, out = rstd*(x W'^T - mean*g) + h.  No LayerNorm launch, no normalised intermediate.
  static void fold_ln(GemmArgs& a, const FoldW& f, const RowStat& rs, int C) {
    a.w = f.w; a.bias = f.h;
    a.ln_stat = rs.ptr; a.ln_ntn = rs.ntn; a.ln_g = f.g; a.ln_C = C; a.ln_eps = 1e-5f;
  }

  // factor folded into every Q projection feeding `attention()`: softmax(q k^T / sqrt(d)) = 2^(q' k^T) normalised
  static float q_scale(int d) { return 1.4426950408889634f / sqrtf((float)d); }

  // out_proj + residual, then the next Linear with its LayerNorm, as one launch (b2b.hip): C = 320 blocks only
  int b2b(const Act& a1, const ConvW& w1, const Act& r1, const Act* s_out, const FoldW& f2, int K2, int partial, float cscale,
          const Act* r2, const Act& y, Act* y_gacc = nullptr) {
    TRY(flush_pending());
    B2bArgs t;
    memset(&t, 0, sizeof(t));
    t.a1 = a1.h; t.lda1 = a1.C; t.w1 = w1.w; t.b1 = w1.bias;
    if (r1.f) { t.r1 = r1.f; t.r1_f32 = 1; } else { t.r1 = r1.h; }
    if (s_out) { t.s32 = s_out->f; t.s16 = s_out->h; }
    t.w2 = f2.w; t.K2 = K2; t.h2 = f2.h; t.partial = partial; t.cscale = cscale;
    if (r2) { if (r2->f) { t.r2 = r2->f; t.r2_f32 = 1; } else { t.r2 = r2->h; } }
    if (y.f) { t.out = y.f; t.out_f32 = 1; t.out16 = y.h; } else { t.out = y.h; }
    t.M = a1.M(); t.eps = 1e-5f; t.npass2 = 1; t.ldo = 320;
    t.ln_guard = ln_guard; t.ln_guard_thr2 = ln_guard_thr() * ln_guard_thr();
    const int bm = sdmi_b2b_tile_rows(t);                  // 32 rows per workgroup up to one round of workgroups, else 64
    if (y_gacc && y_gacc->grec && (a1.H * a1.W) % 64 == 0 && (a1.H * a1.W) / bm * (320 / kGnAtom) <= kGnRecMax) {
      t.gacc.rec = y_gacc->grec; t.gacc.atom = kGnAtom; t.gacc.natoms = 320 / kGnAtom; t.gacc.rows_img = a1.H * a1.W; t.gacc.mod = t.M;
      t.gacc.T = t.gacc.rows_img / bm; t.gacc.parts = 1;
      y_gacc->gok = true; y_gacc->gT = t.gacc.T; y_gacc->gparts = 1;
    }
    const double flops = 2.0 * t.M * 320.0 * (320.0 + K2);
    prof_begin(0, flops);
    TRY(sdmi_launch_b2b(t, st));
    prof_end();
    launches += 1;
    log_launch("b2b M=%d K2=%d partial=%d flops=%.0f", t.M, K2, partial, flops);
    return SDMI_OK;
  }

  // conv_input, then layernorm_1 + in_proj (q | k | v, V transposed for the attention kernel) as one launch (b2b.hip)
  // gn != null: a1 is the RAW stream and the block's GroupNorm (statistics already in gn_partial) is applied inside the kernel
  int b2b_qkv(const Act& a1, const NormW* gn, const ConvW& w1, const Act& s_out, const FoldW& f2, float cscale, const Act& qk,
              f16* vt, int S, int ldt, bool from_records = false) {
    TRY(flush_pending());
    B2bArgs t;
    memset(&t, 0, sizeof(t));
    t.a1 = a1.h; t.lda1 = a1.C; t.w1 = w1.w; t.b1 = w1.bias;
    if (gn) {
      t.gx = a1.f ? (const void*)a1.f : (const void*)a1.h; t.gx_f32 = a1.f != nullptr;
      t.gn_partial = gn_partial; t.gn_nchunk = sdmi_gn_nchunk(S); t.gn_gamma = gn->gamma; t.gn_beta = gn->beta; t.gn_eps = 1e-6f;
      if (from_records) {      // the statistics its producer left with the tensor (one 10-channel atom = one group at C = 320)
        t.gn_partial = a1.grec; t.gn_nchunk = a1.gT; t.gn_parts = a1.gparts;
      }
    }
    t.s32 = s_out.f; t.s16 = s_out.h;
    t.w2 = f2.w; t.K2 = 320; t.h2 = f2.h; t.cscale = cscale;
    t.out = qk.h; t.ldo = qk.C; t.npass2 = 3; t.vt = vt; t.S = S; t.ldt = ldt;
    t.M = a1.M(); t.eps = 1e-5f;
    t.ln_guard = ln_guard; t.ln_guard_thr2 = ln_guard_thr() * ln_guard_thr();
    const double flops = 2.0 * t.M * 320.0 * (320.0 + 960.0);
    prof_begin(0, flops);
    TRY(sdmi_launch_b2b(t, st));
    prof_end();
    launches += 1;
    log_launch("b2b M=%d K2=960 partial=0 flops=%.0f", t.M, flops);
    return SDMI_OK;
  }

  // set_context side of the folded cross-attention of block i (see xfW1): W1 = (K masked per head) Wq^T-form with
  // layernorm_2 folded, W2 = Wo (V masked per head)^T.  Both products run on the MFMA GEMM over the full C (the per-head
  // masks make the off-head terms exact zeros), ~25 us per block once per prompt.
  int xattn_fold(int i, const AttnW& w, const Act& ctx) {
    const int B = ctx.B, C = w.C, R = B * kXfCols;
    { GemmArgs a = base_args(ctx, nullptr, w.v, kCtxPad, 1, 1, 0); a.out = xf_vp; a.ldc = C; TRY(gemm(a)); }
    TRY(sdmi_launch_xattn_mask(ctxK[i], xf_vp, xf_km, xf_vm, B, kHeads, w.dh, kCtxPad, ctx_tokens, st));
    {
      Act km; km.h = xf_km; km.B = 1; km.H = R; km.W = 1; km.C = C;
      ConvW wq; wq.w = w.wqT; wq.O = C; wq.I = C; wq.ks = 1;
      GemmArgs a = base_args(km, nullptr, wq, R, 1, 1, 0);
      a.out = xf_kq; a.out_f32 = 1; a.ldc = C;
      TRY(gemm(a));
    }
    TRY(sdmi_launch_ln_fold_prep(xf_kq, 1, w.ln2.gamma, w.ln2.beta, nullptr, xfW1[i], xfG[i], xfH[i], R, C, st));
    {
      Act wo; wo.h = w.out2.w; wo.B = 1; wo.H = C; wo.W = 1; wo.C = C;
      ConvW vm; vm.w = xf_vm; vm.O = R; vm.I = C; vm.ks = 1;
      GemmArgs a = base_args(wo, nullptr, vm, C, 1, 1, 0);
      a.out = xfW2[i]; a.ldc = R;
      TRY(gemm(a));
    }
    return SDMI_OK;
  }

  // UNET_AttentionBlock (sd/diffusion.py:271-381)
  int attn_block(const AttnW& w, const Act& x, Act* y) {
    if (x.C != w.C) { sdmi_set_error("attn_block: C %d vs %d", x.C, w.C); return SDMI_EINVAL; }
    if (ctx_batch != x.B || (int)ctxK.size() <= w.ctx_idx) {
      sdmi_set_error("attn_block: context not set for batch %d (sdmi_unet_set_context)", x.B);
      return SDMI_EINVAL;
    }
    const int B = x.B, S = x.H * x.W, C = w.C;
    const int Spad = ((S + 63) / 64) * 64;
    Act t0, s0, u, qk, ao, s1, q2, s2;
    // The block's INNER stream (s0 -> s1 -> s2: three additions, sd/diffusion.py:321-363) can be kept in fp16 only
    // (SDMI_ATTN_INNER_F16=1): each of these tensors is a 10.5 MB fp32 write + a 10.5 MB fp32 read per GEMM at 64x64, half
    // the traffic of the four K = C GEMMs that are bound by it.  Measured on one MI355X (same box, same plans): 205.6 vs
    // 202.3 steps/s, attention blocks rel-L2 1.9e-4 vs 1.6e-4, 50-step txt2img pixel MAE 6.97e-4 vs 6.30e-4.  The default
    // keeps the fp32 copies: 1.6 % of speed is not worth 10 % of the parity budget.
    static const bool inner_f32 = getenv("SDMI_ATTN_INNER_F16") == nullptr;
    TRY(new_act(B, x.H, x.W, C, inner_f32, &s0));
    RowStat rs;
    // C = 320 (the 64x64 level): conv_input + in_proj, out_proj 1 + q_proj, and out_proj 2 + feed-forward, each as ONE
    // back-to-back launch
    static const bool b2b_on = !(getenv("SDMI_B2B") && atoi(getenv("SDMI_B2B")) == 0);
    static const bool b2b_qkv_on = !(getenv("SDMI_B2B_QKV") && atoi(getenv("SDMI_B2B_QKV")) == 0);
    const bool no_fold = !ln_fold_on;
    // one round of workgroups only (<= 256 of them, 32 or 64 rows each): at 768x768 (M = 18432: 288 workgroups) the second,
    // nearly empty round makes the fused form slower than the GEMM pairs (same box: 8.17 vs 8.11 ms/step)
    // (beyond 16384 rows -- the batched multi-prompt mode -- only whole rounds of 64-row workgroups: SDMI_B2B_FULLROUNDS=0 to A/B)
    static const bool b2b_rounds = !(getenv("SDMI_B2B_FULLROUNDS") && atoi(getenv("SDMI_B2B_FULLROUNDS")) == 0);
    const bool use_b2b = b2b_on && !no_fold && !accurate && C == 320 && (B * S) % 32 == 0 &&
                         (B * S <= 16384 || (b2b_rounds && (B * S) % (64 * 256) == 0 && S % 64 == 0));
    TRY(new_act(B, x.H, x.W, 2 * C, false, &qk));
    f16* vt = (f16*)arena.alloc((size_t)B * C * Spad * 2);
    if (!vt) { sdmi_set_error("activation arena exhausted"); return SDMI_ENOMEM; }
    if (Spad != S) { SDMI_CHECK_HIP(hipMemsetAsync(vt, 0, (size_t)B * C * Spad * 2, st)); launches += 1; }
    // SDMI_B2B_GN=1: the block's GroupNorm is applied inside the back-to-back kernel as well (statistics launch only).
    // Parity-green and measured time-neutral on one MI355X (same box: 4.173 vs 4.171 ms/step, 5 launches and 0.05 GB of
    // traffic fewer): the kernel's start-up takes what the 8.8 us gn_apply gave back, so the default keeps gn_apply.
    static const bool b2b_gn_on = getenv("SDMI_B2B_GN") && atoi(getenv("SDMI_B2B_GN")) != 0;
    // With the producer's statistics records at hand (x.gok) neither the statistics launch nor gn_apply is needed: the kernel
    // sums the records of its image (T x 32 entries) in its prologue and normalises its rows into the A panels.  Same box,
    // interleaved: 253.5 / 253.3 -> 254.9 / 255.3 steps/s, 251 -> 246 launches (the three-pass launch grows 32.5 -> 35.5 us,
    // five gn_apply launches of 8.4 us go).  SDMI_B2B_GN_REC=0 keeps gn_apply.
    static const bool b2b_gn_rec_on = !(getenv("SDMI_B2B_GN_REC") && atoi(getenv("SDMI_B2B_GN_REC")) == 0);
    const bool gn_from_rec = b2b_gn_rec_on && x.gok && x.grec && kGnAtom == C / 32 && !pend;
    if (use_b2b && b2b_qkv_on && S % 32 == 0 && gn_from_rec) {
      TRY(b2b_qkv(x, &w.gn, w.conv_in, s0, w.in_proj_f, q_scale(w.dh), qk, vt, S, Spad, true));
    } else if (use_b2b && b2b_qkv_on && S % 32 == 0 && b2b_gn_on) {
      // groupnorm + conv_input + layernorm_1 + in_proj: statistics launch, then everything else in the back-to-back kernel
      TRY(gn_stats(x, nullptr));
      TRY(b2b_qkv(x, &w.gn, w.conv_in, s0, w.in_proj_f, q_scale(w.dh), qk, vt, S, Spad));
    } else if (use_b2b && b2b_qkv_on && S % 32 == 0) {
      TRY(groupnorm(x, nullptr, w.gn, 1e-6f, 0, &t0));
      TRY(b2b_qkv(t0, nullptr, w.conv_in, s0, w.in_proj_f, q_scale(w.dh), qk, vt, S, Spad));
    } else {
      // GroupNorm (no SiLU) -> 1x1 conv_input (sd/diffusion.py:294-298).  SDMI_GN_FRAG=1: with the producer's statistics records at
      // hand and a plan whose tile was built with the variant, the GEMM normalises its own A fragments (GemmArgs::gna_rec): no
      // GroupNorm launch, no normalised intermediate.  Parity-green and OFF by default: measured on one MI355X (same box, interleaved)
      // 259.2 / 259.1 steps/s without against 259.1 / 259.3 with it at 32x32 (5 launches fewer), 258.0 / 255.7 when the 16x16 / 8x8
      // maps take records too -- the records prologue, the extra LDS reads and the packed-fp16 arithmetic per fragment cost what the
      // 7 us gn_apply launch gave back, and three fp16 roundings instead of one leave 1.7 x its error on that GEMM
      // (tests/test_gpu_kernels.py::test_gemm_groupnorm_on_a_fragments).
      static const bool gna_on = getenv("SDMI_GN_FRAG") && atoi(getenv("SDMI_GN_FRAG")) != 0;
      bool gna_done = false;
      if (gna_on && x.gok && x.grec && x.h && !pend && C % (32 * kGnAtom) == 0) {
        GemmArgs a = base_args(x, nullptr, w.conv_in, x.H, x.W, 1, 0);
        set_out(a, s0);
        std::map<ShapeKey, Plan>::iterator it;
        TRY(plan_of(a, &it));               // (the plan of the plain launch: a shape that has to be timed is timed without the variant)
        a.gna_rec = x.grec; a.gna_gamma = w.gn.gamma; a.gna_beta = w.gn.beta; a.gna_eps = 1e-6f;
        a.gna_T = x.gT; a.gna_parts = x.gparts; a.gna_atom = kGnAtom; a.gna_rows = x.H * x.W;
        if (sdmi_gemm_gna_ok(a, it->second.cfg)) {
          TRY(gemm(a, &rs));
          gna_done = true;
        }
      }
      if (!gna_done) {
        TRY(groupnorm(x, nullptr, w.gn, 1e-6f, 0, &t0));
        GemmArgs a = base_args(t0, nullptr, w.conv_in, x.H, x.W, 1, 0); set_out(a, s0); TRY(gemm(a, &rs));
      }
      // self-attention
      if (!rs.ptr) TRY(layernorm(s0, w.ln1, &u));
      GemmArgs a = base_args(rs.ptr ? s0 : u, nullptr, w.in_proj, x.H, x.W, 1, 0);
      if (rs.ptr) fold_ln(a, w.in_proj_f, rs, C);
      a.out = qk.h; a.ldc = 2 * C;
      a.outT = vt; a.nt0 = 2 * C; a.S = S; a.ldt = Spad; a.tperm = 1;
      a.cscale = q_scale(w.dh); a.cs_hi = C;
      TRY(gemm(a));
    }
    TRY(new_act(B, x.H, x.W, C, false, &ao));
    TRY(attention(qk.h, 2 * C, qk.h + C, 2 * C, S, vt, Spad, ao.h, C, B, w.dh, S, S, accurate ? ao.f : nullptr));
    TRY(new_act(B, x.H, x.W, C, inner_f32, &s1));
    bool q_done = false;
    if (use_b2b) {
      TRY(new_act(B, x.H, x.W, C, false, &q2));
      TRY(b2b(ao, w.out1, s0, &s1, w.q_f, C, 0, q_scale(w.dh), nullptr, q2));
      q_done = true;
    } else {
      GemmArgs a = base_args(ao, nullptr, w.out1, x.H, x.W, 1, 0); set_res(a, s0); set_out(a, s1); TRY(gemm(a, &rs));
    }
    // cross-attention (K/V hoisted in set_context)
    static const bool xfold_on = !(getenv("SDMI_XATTN_FOLD") && atoi(getenv("SDMI_XATTN_FOLD")) == 0);
    const bool xfold = !q_done && xfold_on && !accurate && rs.ptr && S % 64 == 0 && S <= 1024 && (int)xfW1.size() > w.ctx_idx && xfW1[w.ctx_idx] != nullptr;
    TRY(new_act(B, x.H, x.W, C, inner_f32, &s2));
    if (xfold) {
      // two GEMMs with per-image weights (see xfW1 / xfW2): probabilities, then values x out_proj + residual
      Act pr;
      TRY(new_act(B, x.H, x.W, kXfCols, false, &pr));
      {
        ConvW w1; w1.w = xfW1[w.ctx_idx]; w1.bias = xfH[w.ctx_idx]; w1.O = kXfCols; w1.I = C; w1.ks = 1;
        GemmArgs a = base_args(s1, nullptr, w1, x.H, x.W, 1, 0);
        a.ln_stat = rs.ptr; a.ln_ntn = rs.ntn; a.ln_g = xfG[w.ctx_idx]; a.ln_C = C; a.ln_eps = 1e-5f;
        a.img_rows = S; a.w_img_stride = kXfCols * C; a.vec_img_stride = kXfCols;
        a.act = 2; a.sm_valid = ctx_tokens;
        a.out = pr.h; a.ldc = kXfCols;
        TRY(gemm(a));
      }
      {
        ConvW w2; w2.w = xfW2[w.ctx_idx]; w2.bias = w.out2.bias; w2.O = C; w2.I = kXfCols; w2.ks = 1;
        GemmArgs a = base_args(pr, nullptr, w2, x.H, x.W, 1, 0);
        a.ldw = B * kXfCols; a.img_rows = S; a.w_img_stride = kXfCols; a.vec_img_stride = 0;
        set_res(a, s1);
        set_out(a, s2);
        TRY(gemm(a, &rs));
      }
    } else {
      if (!q_done) {
        if (!rs.ptr) TRY(layernorm(s1, w.ln2, &u));
        TRY(new_act(B, x.H, x.W, C, false, &q2));
        GemmArgs a = base_args(rs.ptr ? s1 : u, nullptr, w.q, x.H, x.W, 1, 0);
        if (rs.ptr) fold_ln(a, w.q_f, rs, C);
        a.out = q2.h; a.ldc = C;
        a.cscale = q_scale(w.dh); a.cs_hi = C;
        TRY(gemm(a));
      }
      TRY(attention(q2.h, C, ctxK[w.ctx_idx], C, kCtxPad, ctxVt[w.ctx_idx], kCtxVtLd, ao.h, C, B, w.dh, S, ctx_tokens, accurate ? ao.f : nullptr));
      if (use_b2b) {
        // s2 = out_proj 2 + s1 never leaves the workgroup: its only reader is the feed-forward
        TRY(new_act(B, x.H, x.W, C, true, y));
        TRY(b2b(ao, w.out2, s1, nullptr, w.ffn_f, 2 * C, 1, 0.f, &x, *y, y));
        return SDMI_OK;
      }
      { GemmArgs a = base_args(ao, nullptr, w.out2, x.H, x.W, 1, 0); set_res(a, s1); set_out(a, s2); TRY(gemm(a, &rs)); }
    }
    // feed-forward + conv_output + the block's long residual: ONE GEMM over [LN3(s2) | s2] (AttnW::ffn)
    if (!rs.ptr) TRY(layernorm(s2, w.ln3, &u));
    TRY(new_act(B, x.H, x.W, C, true, y));
    {
      GemmArgs a = base_args(rs.ptr ? s2 : u, &s2, w.ffn, x.H, x.W, 1, 0);
      if (rs.ptr) {
        fold_ln(a, w.ffn_f, rs, C);
        a.ln_ksteps = C / 64;
        a.ln_out = (float*)arena.alloc((size_t)a.M * 8);        // lets the plan split K at the fold boundary (small maps)
        if (!a.ln_out) { sdmi_set_error("activation arena exhausted"); return SDMI_ENOMEM; }
      }
      set_res(a, x);
      set_out(a, *y);
      TRY(gemm(a, nullptr, false, y));
    }
    return SDMI_OK;
  }

  // pad = 1: the UNet's / decoder's convs.  pad = 0 with stride 2 is the VAE encoder's asymmetric
  // F.pad(x,(0,1,0,1)) + padding-0 conv (sd/encoder.py:120-122): taps past the right/bottom edge read zeros.
  int conv3(const ConvW& w, const Act& x, int stride, int ups, Act* y, int pad = 1) {
    const int Hi = x.H << ups, Wi = x.W << ups;
    const int Ho = pad ? (Hi - 1) / stride + 1 : (Hi + 1 - 3) / stride + 1;
    const int Wo = pad ? (Wi - 1) / stride + 1 : (Wi + 1 - 3) / stride + 1;
    TRY(new_act(x.B, Ho, Wo, w.O, true, y));
    // x2 upsample + 3x3 conv as four 2x2 phase convs over the source grid: 4/9 of the multiplies, but 16/9 of the weight
    // bytes -- used from 512 source pixels up (the 8x8 -> 16x16 conv streams its 29 MB of weights at M = 128 and is
    // weight-bound).  SDMI_UPS_PHASE=0: never; SDMI_UPS_PHASE_MINROWS moves the threshold.
    static const bool phase_on = !(getenv("SDMI_UPS_PHASE") && atoi(getenv("SDMI_UPS_PHASE")) == 0);
    static const int phase_min = getenv("SDMI_UPS_PHASE_MINROWS") ? atoi(getenv("SDMI_UPS_PHASE_MINROWS")) : 512;
    if (phase_on && !accurate && ups == 1 && stride == 1 && pad == 1 && w.w4 && x.M() >= phase_min && x.M() % 64 == 0) {
      ConvW wp = w;
      wp.w = w.w4; wp.ks = 2;
      GemmArgs a = base_args(x, nullptr, wp, x.H, x.W, 1, 0);
      a.img_rows = a.M; a.w_img_stride = w.O * 4 * x.C; a.M *= 4; a.phase2 = 1;
      set_out(a, *y);
      TRY(gemm(a, nullptr, false, y));
      return SDMI_OK;
    }
    GemmArgs a = base_args(x, nullptr, w, Ho, Wo, stride, ups);
    a.pad = pad;
    set_out(a, *y);
    TRY(gemm(a, nullptr, false, y));
    return SDMI_OK;
  }

  // ---- time path ----------------------------------------------------------------------------
  int compute_timevecs(const float* temb, int n, float* out_rows) {
    if (!has_time) { sdmi_set_error("time embedding weights not loaded"); return SDMI_ENOENT; }
    if (n > max_steps) { sdmi_set_error("schedule of %d steps exceeds max %d", n, max_steps); return SDMI_EINVAL; }
    float* t1 = time_scratch;
    float* t2 = time_scratch + (size_t)max_steps * kTime;
    TRY(sdmi_launch_small_linear(temb, te1.w, te1.bias, t1, n, kTime, 320, 0, kTime, st));
    TRY(sdmi_launch_small_linear(t1, te2.w, te2.bias, t2, n, kTime, kTime, 1, kTime, st));
    for (const std::string& p : res_order) {
      const ResW& r = res[p];
      TRY(sdmi_launch_small_linear(t2, r.time.w, r.bias1, out_rows + r.time_off, n, r.cout, kTime, 1, time_total, st));
    }
    return SDMI_OK;
  }
  // time_dev: (1,1280) TimeEmbedding output given directly (block tests)
  int compute_timevec_from_time(const float* time_dev, const ResW& r, float* out) {
    return sdmi_launch_small_linear(time_dev, r.time.w, r.bias1, out, 1, r.cout, kTime, 1, r.cout, st);
  }
};

}  // namespace sdmi
