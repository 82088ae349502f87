// sd_plan.hip -- the launch list and the hipGraph of a network live in the library (sd_unet_forward,
// sd_vae_decode, sd_vae_encode as C entry points).
//
// A MODEL is (i) a registry of the device buffers its launches touch, (ii) named bindings (inputs / outputs) and (iii) one or more
// named PLANS: flat lists of recorded sd_* launches.  A plan is recorded once -- between sd_model_record_begin / _end every sd_*
// launch entry point called on the thread appends its arguments instead of launching -- and then executed natively: eagerly
// (sd_model_run) or as a hipGraph captured by the library on a private stream (sd_model_replay).  sd_model_save writes the
// registry, the bindings, the plans (pointers rewritten as buffer + offset) and the contents of the persistent buffers (weights,
// constants) to ONE file; sd_model_load rebuilds all of it in library-owned device memory, so that a caller without Python runs
//     sd_model_load("unet.sdm", &m); sd_unet_set_context(m, ctx, s); sd_unet_forward(m, x_in, t, eps, s);
// replaces: the nn.Module objects the reference pipeline calls -- self.unet(...) (utils/adaptive_mask_inpainting.py:1001-1007),
// self.vae.decode (:1086, :1112), self.vae.encode (:677-680).  Which layer follows which is decided by whoever records the plan
// (coma_amd/sd/unet.py, vae.py); the library owns execution.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "sd_plan.h"
#include "../../include/sd_hip.h"

namespace sd {

using coma::fail;

struct Buffer { char* ptr; size_t bytes; int flags; bool owned; };
struct Binding { char* ptr; size_t bytes; };
struct Plan {
  std::string name;
  std::vector<PlanRec> recs;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};
struct Model {
  std::vector<Buffer> bufs;
  std::map<std::string, Binding> binds;
  std::vector<Plan> plans;
  hipStream_t cap_stream = nullptr;
  Plan* find(const char* name) {
    for (auto& p : plans) if (p.name == name) return &p;
    return nullptr;
  }
  int locate(const void* q) const {              // index of the registered buffer that contains q, or -1
    const char* c = static_cast<const char*>(q);
    for (size_t k = 0; k < bufs.size(); ++k)
      if (c >= bufs[k].ptr && c < bufs[k].ptr + bufs[k].bytes) return (int)k;
    return -1;
  }
};

static thread_local Model* t_model = nullptr;
static thread_local Plan* t_plan = nullptr;

bool plan_recording() { return t_plan != nullptr; }
int plan_record(const PlanRec& r) {
  t_plan->recs.push_back(r);
  return COMA_OK;
}

static void drop_graph(Plan& p) {
  if (p.exec) (void)hipGraphExecDestroy(p.exec);
  if (p.graph) (void)hipGraphDestroy(p.graph);
  p.exec = nullptr;
  p.graph = nullptr;
}

// ---- replay of one record: the public entry point with its arguments unpacked (this thread is not recording here)
static int launch(const PlanRec& r, void* st) {
  void* const* p = r.p;
  const int64_t* i = r.i;
  const double* f = r.f;
  switch (r.kind) {
    case PK_CONV: {
      sd_conv_gemm_desc d;
      memset(&d, 0, sizeof d);
      d.a0 = p[0]; d.a1 = p[1]; d.w = p[2]; d.bias = p[3]; d.bias_bn = p[4]; d.res = p[5]; d.out = p[6]; d.workspace = p[7];
      d.colstats = (float*)p[8];
      d.c0 = (int)i[0]; d.c1 = (int)i[1]; d.batch = (int)i[2]; d.in_h = (int)i[3]; d.in_w = (int)i[4]; d.out_h = (int)i[5]; d.out_w = (int)i[6];
      d.taps = (int)i[7]; d.stride = (int)i[8]; d.upsample = (int)i[9]; d.pad = (int)i[10]; d.n = (int)i[11]; d.ldbb = (int)i[12];
      d.ldr = (int)i[13]; d.ldo = (int)i[14]; d.epi = (int)i[15]; d.nbatch_z = (int)i[16]; d.stride_a = i[17]; d.stride_w = i[18];
      d.stride_out = i[19]; d.stride_res = i[20]; d.workspace_bytes = (size_t)i[21];
      d.out_t = p[9]; d.n_split = (int)(i[22] & 0xfffff); d.ldo_t = (int)((i[22] >> 20) & 0xfffff); d.rows_per_sample = (int)((i[22] >> 40) & 0xfffff);
      d.phase = (int)((i[22] >> 60) & 7);
      return sd_conv_gemm_f16(&d, st);
    }
    case PK_GN:
      return sd_groupnorm_f16(p[0], p[1], (int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], (float)f[0], p[2], p[3], (int)i[5], p[4],
                              (float*)p[5], st);
    case PK_GN_COLSTATS:
      return sd_groupnorm_colstats_f16(p[0], p[1], (int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], (float)f[0], p[2], p[3], (int)i[5],
                                       p[4], (float*)p[5], (const float*)p[6], (const float*)p[7], st);
    case PK_LN:
      return sd_layernorm_f16(p[0], i[0], (int)i[1], (float)f[0], p[1], p[2], p[3], st);
    case PK_ATTN:
      return sd_attention_f16(p[0], p[1], p[2], p[3], (int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], (int)i[5], (int)i[6], (int)i[7],
                              (int)i[8], (float)f[0], (int)i[9], st);
    case PK_SOFTMAX:
      return sd_softmax_f16(p[0], i[0], (int)i[1], (int)i[2], (float)f[0], st);
    case PK_TEMB:
      return sd_timestep_embedding_f16((const float*)p[0], (int)i[0], (int)i[1], p[1], st);
    case PK_ATTN_WIDE:
      return sd_attention_wide_f16(p[0], p[1], p[2], p[3], (int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], (int)i[5], (int)i[6], (int)i[7],
                                   (int)i[8], (float)f[0], st);
    case PK_XCHAIN:
      return sd_xattn_chain_f16(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], p[13], p[14], i[0], (int)i[1],
                                (int)i[2], (int)i[3], (float)f[0], nullptr, 0, st);
    case PK_XFRONT:
      return sd_xfront_f16(p[0], (const float*)p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], i[0], (int)i[1], (int)i[2], (float)f[0], st);
    case PK_GN_TABLE:
      return sd_groupnorm_table_f16(p[0], (int)i[0], (int)i[1], (int)i[2], (int)i[3], (float)f[0], p[1], p[2], (float*)p[3], (const float*)p[4], (int)i[4], st);
    case PK_XTAIL:
      return sd_xtail_f16(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], (float*)p[10], i[0], st);
    case PK_CONV_SMALL_N:
      return sd_conv3x3_small_n_f16(p[0], (const float*)p[1], (int)i[0], p[2], p[3], (int)i[1], (int)i[2], (int)i[3], (int)i[4], (int)i[5], p[4],
                                    (int)i[6], st);
    case PK_WINO_IN:
      return sd_winograd_input_f16(p[0], p[1], (int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], (int)i[5], (const float*)p[3], (int)i[6], (float)f[0], p[2], st);
    case PK_WINO_OUT:
      return sd_winograd_output_f16(p[0], (int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], p[1], p[2], (int)i[5], p[3], (int)i[6], p[4],
                                    (int)i[7], (int)i[8], (float)f[0], (float*)p[5], st);
    case PK_GN_WINO_IN:
      return sd_gn_winograd_input_f16(p[0], p[1], (int)i[0], (int)i[1], p[2], (int)i[2], p[3], p[4], (int)i[3], (int)i[4], (int)i[5], (int)i[6],
                                      (int)i[7], (float)f[0], p[5], p[6], (int)i[8], (float)f[1], p[7], st);
    case PK_IM2COL_C3:
      return sd_im2col3x3_c3_f16(p[0], (int)i[0], (int)i[1], (int)i[2], (int)i[3], p[1], st);
    case PK_GN_TABLE_CAT:
      return sd_groupnorm_table_cat_f16((int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], (float)f[0], p[0], p[1], (float*)p[2], (const float*)p[3],
                                        (const float*)p[4], st);
    case PK_CONV_HALO:
      return sd_conv3x3_halo_f16(p[0], (int)i[0], (const float*)p[1], (int)i[1], p[2], p[3], p[4], (int)i[2], (int)i[3], (int)i[4], (int)i[5],
                                 (int)i[6], p[5], (int)i[7], (float*)p[6], st);
    case PK_CONV_C3:
      return sd_conv3x3_c3_f16(p[0], (int)i[0], p[1], p[2], (int)i[1], (int)i[2], (int)i[3], (int)i[4], p[3], (int)i[5], (float*)p[4], st);
    case PK_COPY:
      return sd_copy_d2d(p[0], p[1], (size_t)i[0], st);
    case PK_SEG:
      return seg_replay(r, st);
    default:
      return fail(COMA_E_INVALID, "sd plan: unknown launch kind %d", r.kind);
  }
}

static int run_plan(Plan& pl, void* st) {
  if (plan_recording()) return fail(COMA_E_INVALID, "sd_model_run: this thread is recording a plan");
  for (const PlanRec& r : pl.recs) {
    const int rc = launch(r, st);
    if (rc != COMA_OK) return rc;
  }
  return COMA_OK;
}

static Model* as_model(void* m) { return static_cast<Model*>(m); }

}  // namespace sd

using namespace sd;

extern "C" int sd_copy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
  if (plan_recording()) {
    PlanRec r{};
    r.kind = PK_COPY; r.p[0] = dst; r.p[1] = const_cast<void*>(src); r.i[0] = (int64_t)bytes;
    return plan_record(r);
  }
  if (!dst || !src || bytes == 0) return fail(COMA_E_INVALID, "sd_copy_d2d: bad args");
  if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
    return fail(COMA_E_LAUNCH, "sd_copy_d2d: hipMemcpyAsync failed");
  return COMA_OK;
}

extern "C" int sd_model_create(void** model) {
  if (!model) return fail(COMA_E_INVALID, "sd_model_create: null pointer");
  *model = new Model();
  return COMA_OK;
}

extern "C" int sd_model_destroy(void* model) {
  if (!model) return COMA_OK;
  Model* m = as_model(model);
  if (t_model == m) { t_model = nullptr; t_plan = nullptr; }
  for (auto& p : m->plans) drop_graph(p);
  if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
  for (auto& b : m->bufs) if (b.owned) (void)hipFree(b.ptr);
  delete m;
  return COMA_OK;
}

extern "C" int sd_model_register_buffer(void* model, void* ptr, size_t bytes, int flags) {
  if (!model || !ptr || bytes == 0) return fail(COMA_E_INVALID, "sd_model_register_buffer: bad args");
  Model* m = as_model(model);
  char* lo = static_cast<char*>(ptr);
  char* hi = lo + bytes;
  // SD_BUF_IF_NEW (what the recording hook passes for every tensor a launch touches): a range some entry already covers keeps that
  // entry's flags -- scratch the owner registered with 0 / SD_BUF_ZEROED must not become part of a saved model just because a
  // recorded launch points into it (a UNet at batch 16 would save GBs of activations) -- and anything else is a constant
  // seen for the first time: registered SD_BUF_PERSISTENT.
  if (flags & SD_BUF_IF_NEW) {
    for (const Buffer& b : m->bufs)
      if (lo >= b.ptr && hi <= b.ptr + b.bytes) return COMA_OK;
    flags = (flags & ~SD_BUF_IF_NEW) | SD_BUF_PERSISTENT;
  }
  // validate before mutating: a range that straddles a buffer the model owns is refused with the registry untouched
  for (const Buffer& b : m->bufs)
    if (b.owned && lo < b.ptr + b.bytes && b.ptr < hi && !(lo >= b.ptr && hi <= b.ptr + b.bytes))
      return fail(COMA_E_INVALID, "sd_model_register_buffer: range straddles a buffer the model owns");
  // Fold EVERY entry the range touches into one (a range bridging two entries must not leave overlapping entries behind) and OR
  // the flags also when the range was already covered: a buffer first seen as scratch and later EXPLICITLY registered
  // SD_BUF_PERSISTENT (constants Python filled outside a plan) must be saved with the model.  Anything filled outside a plan has to
  // be registered PERSISTENT by its owner; a persistent sub-range makes the whole entry persistent (larger file, never a missing constant).
  for (size_t k = 0; k < m->bufs.size();) {
    Buffer& b = m->bufs[k];
    if (lo < b.ptr + b.bytes && b.ptr < hi) {
      if (b.owned) { b.flags |= flags; return COMA_OK; }     // library-allocated (loaded model), range inside it (checked above)
      if (b.ptr < lo) lo = b.ptr;
      if (b.ptr + b.bytes > hi) hi = b.ptr + b.bytes;
      flags |= b.flags;
      m->bufs.erase(m->bufs.begin() + (long)k);
      continue;                               // the union may now reach entries already passed: the sweep below catches them
    }
    ++k;
  }
  // one more sweep for entries that only the grown union reaches
  for (bool again = true; again;) {
    again = false;
    for (size_t k = 0; k < m->bufs.size(); ++k) {
      Buffer& b = m->bufs[k];
      if (!b.owned && lo < b.ptr + b.bytes && b.ptr < hi) {
        if (b.ptr < lo) lo = b.ptr;
        if (b.ptr + b.bytes > hi) hi = b.ptr + b.bytes;
        flags |= b.flags;
        m->bufs.erase(m->bufs.begin() + (long)k);
        again = true;
        break;
      }
    }
  }
  m->bufs.push_back(Buffer{lo, (size_t)(hi - lo), flags, false});
  return COMA_OK;
}

extern "C" int sd_model_bind(void* model, const char* name, void* ptr, size_t bytes) {
  if (!model || !name || !ptr || bytes == 0 || strlen(name) > 31) return fail(COMA_E_INVALID, "sd_model_bind: bad args");
  as_model(model)->binds[name] = Binding{static_cast<char*>(ptr), bytes};
  return COMA_OK;
}

extern "C" int sd_model_binding(const void* model, const char* name, void** ptr, size_t* bytes) {
  if (!model || !name) return fail(COMA_E_INVALID, "sd_model_binding: bad args");
  const Model* m = static_cast<const Model*>(model);
  auto it = m->binds.find(name);
  if (it == m->binds.end()) return fail(COMA_E_INVALID, "sd_model_binding: no binding named '%s'", name);
  if (ptr) *ptr = it->second.ptr;
  if (bytes) *bytes = it->second.bytes;
  return COMA_OK;
}

extern "C" int sd_model_record_begin(void* model, const char* plan_name) {
  if (!model || !plan_name || strlen(plan_name) > 31) return fail(COMA_E_INVALID, "sd_model_record_begin: bad args");
  if (t_plan) return fail(COMA_E_INVALID, "sd_model_record_begin: this thread is already recording");
  Model* m = as_model(model);
  Plan* p = m->find(plan_name);
  if (!p) { m->plans.push_back(Plan()); p = &m->plans.back(); p->name = plan_name; }
  drop_graph(*p);
  p->recs.clear();
  t_model = m;
  t_plan = p;
  return COMA_OK;
}

extern "C" int sd_model_record_end(void* model) {
  if (!t_plan || t_model != as_model(model)) return fail(COMA_E_INVALID, "sd_model_record_end: not recording into this model");
  t_plan = nullptr;
  t_model = nullptr;
  return COMA_OK;
}

extern "C" int sd_model_num_launches(const void* model, const char* plan_name) {
  if (!model || !plan_name) return -1;
  Plan* p = const_cast<Model*>(static_cast<const Model*>(model))->find(plan_name);
  return p ? (int)p->recs.size() : -1;
}

extern "C" int sd_model_run(void* model, const char* plan_name, void* stream) {
  if (!model || !plan_name) return fail(COMA_E_INVALID, "sd_model_run: bad args");
  Plan* p = as_model(model)->find(plan_name);
  if (!p) return fail(COMA_E_INVALID, "sd_model_run: no plan named '%s'", plan_name);
  return run_plan(*p, stream);
}

namespace sd {
// capture the eager replay of a plan on a private stream (nothing executes) and instantiate it, once
static int ensure_exec(Model* m, Plan* p, const char* who) {
  if (p->exec) return COMA_OK;
  if (!m->cap_stream && hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking) != hipSuccess)
    return fail(COMA_E_LAUNCH, "%s: cannot create the capture stream", who);
  if (hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess)
    return fail(COMA_E_LAUNCH, "%s: hipStreamBeginCapture failed", who);
  const int rc = run_plan(*p, m->cap_stream);
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(m->cap_stream, &g);
  if (rc != COMA_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess || !g) return fail(COMA_E_LAUNCH, "%s: capture of plan '%s' failed: %s", who, p->name.c_str(), hipGetErrorString(e));
  if (hipGraphInstantiate(&p->exec, g, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGraphDestroy(g);
    p->exec = nullptr;
    return fail(COMA_E_LAUNCH, "%s: hipGraphInstantiate failed for plan '%s'", who, p->name.c_str());
  }
  p->graph = g;
  return COMA_OK;
}
}  // namespace sd

extern "C" int sd_model_prepare(void* model, const char* plan_name) {
  if (!model || !plan_name) return fail(COMA_E_INVALID, "sd_model_prepare: bad args");
  Model* m = as_model(model);
  Plan* p = m->find(plan_name);
  if (!p) return fail(COMA_E_INVALID, "sd_model_prepare: no plan named '%s'", plan_name);
  return sd::ensure_exec(m, p, "sd_model_prepare");
}

extern "C" int sd_model_replay(void* model, const char* plan_name, void* stream) {
  if (!model || !plan_name) return fail(COMA_E_INVALID, "sd_model_replay: bad args");
  Model* m = as_model(model);
  Plan* p = m->find(plan_name);
  if (!p) return fail(COMA_E_INVALID, "sd_model_replay: no plan named '%s'", plan_name);
  if (int rc = sd::ensure_exec(m, p, "sd_model_replay")) return rc;
  if (hipGraphLaunch(p->exec, (hipStream_t)stream) != hipSuccess) return fail(COMA_E_LAUNCH, "sd_model_replay: hipGraphLaunch failed");
  return COMA_OK;
}

// ---- file format (little endian, version 3: the launch-record kinds / conv descriptor of r5 -- version 2 files carried LayerNorm-fold fields):
//   "SDMODEL3" | u64 file_bytes | u64 checksum (FNV-1a over 64-bit words of everything after this 24-byte header)
//   | u32 nbuf | { u64 bytes, u32 flags } x nbuf | u32 nbind | { char name[32], u32 buf, u64 offset, u64 bytes } x nbind
//   | u32 nplans | { char name[32], u32 nrec, PlanRec x nrec with every non-null pointer rewritten as ((buf + 1) << 48) | offset }
//   | the bytes of every SD_BUF_PERSISTENT buffer, in registry order
// A launch record only carries the START of every operand; what a launch touches beyond it follows from its integer arguments.  A
// file is therefore accepted only if its size and checksum say it is exactly what sd_model_save wrote (from a model whose plans had
// run): a truncated or corrupted file is refused before any device memory is allocated, not discovered as an out-of-bounds access.
namespace sd {
struct Hasher {                      // FNV-1a on little-endian 64-bit words, tail zero-padded; streaming
  uint64_t h = 0xcbf29ce484222325ULL, carry = 0, total = 0;
  int ncarry = 0;
  void word(uint64_t w) { h = (h ^ w) * 0x100000001b3ULL; }
  void update(const void* p, size_t n) {
    const unsigned char* c = static_cast<const unsigned char*>(p);
    total += n;
    while (n && ncarry) { carry |= (uint64_t)*c++ << (8 * ncarry); --n; if (++ncarry == 8) { word(carry); carry = 0; ncarry = 0; } }
    for (; n >= 8; n -= 8, c += 8) { uint64_t w; memcpy(&w, c, 8); word(w); }
    for (; n; --n) { carry |= (uint64_t)*c++ << (8 * ncarry); ++ncarry; }
  }
  uint64_t digest() const { uint64_t r = h; if (ncarry) r = (r ^ carry) * 0x100000001b3ULL; return r; }
};
struct Writer {
  FILE* f;
  Hasher hash;
  bool ok = true;
  bool put(const void* p, size_t n) { hash.update(p, n); ok = ok && fwrite(p, 1, n, f) == n; return ok; }
};
static bool rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }
constexpr uint64_t kMaxBufIndex = 0xfffe, kMaxOffset = (1ULL << 48) - 1;
}  // namespace sd

extern "C" int sd_model_save(const void* model, const char* path) {
  if (!model || !path) return fail(COMA_E_INVALID, "sd_model_save: bad args");
  const Model* m = static_cast<const Model*>(model);
  if (m->bufs.size() > kMaxBufIndex) return fail(COMA_E_INVALID, "sd_model_save: %zu buffers do not fit the 16-bit buffer field", m->bufs.size());
  for (const auto& b : m->bufs)
    if (b.bytes > kMaxOffset) return fail(COMA_E_INVALID, "sd_model_save: a buffer of %zu bytes does not fit the 48-bit offset field", b.bytes);
  FILE* f = fopen(path, "wb");
  if (!f) return fail(COMA_E_INVALID, "sd_model_save: cannot open %s", path);
  const uint64_t zero2[2] = {0, 0};
  bool ok = fwrite("SDMODEL3", 1, 8, f) == 8 && fwrite(zero2, 1, 16, f) == 16;      // size + checksum are patched in at the end
  Writer w{f};
  const uint32_t nbuf = (uint32_t)m->bufs.size();
  w.put(&nbuf, 4);
  for (const auto& b : m->bufs) { const uint64_t by = b.bytes; const uint32_t fl = (uint32_t)b.flags; w.put(&by, 8); w.put(&fl, 4); }
  const uint32_t nbind = (uint32_t)m->binds.size();
  w.put(&nbind, 4);
  for (const auto& kv : m->binds) {
    char name[32] = {0};
    strncpy(name, kv.first.c_str(), 31);
    const int k = m->locate(kv.second.ptr);
    if (k < 0) { fclose(f); return fail(COMA_E_INVALID, "sd_model_save: binding '%s' is not inside a registered buffer", name); }
    const uint32_t kb = (uint32_t)k;
    const uint64_t off = (uint64_t)(kv.second.ptr - m->bufs[k].ptr), by = kv.second.bytes;
    w.put(name, 32); w.put(&kb, 4); w.put(&off, 8); w.put(&by, 8);
  }
  const uint32_t nplans = (uint32_t)m->plans.size();
  w.put(&nplans, 4);
  for (const auto& pl : m->plans) {
    char name[32] = {0};
    strncpy(name, pl.name.c_str(), 31);
    const uint32_t nrec = (uint32_t)pl.recs.size();
    w.put(name, 32); w.put(&nrec, 4);
    for (PlanRec r : pl.recs) {
      for (auto& q : r.p) {
        if (!q) continue;
        const int k = m->locate(q);
        if (k < 0) { fclose(f); return fail(COMA_E_INVALID, "sd_model_save: plan '%s' uses a pointer outside every registered buffer", name); }
        q = reinterpret_cast<void*>(((uint64_t)(k + 1) << 48) | (uint64_t)(static_cast<char*>(q) - m->bufs[k].ptr));
      }
      w.put(&r, sizeof r);
    }
  }
  std::vector<char> host;
  for (const auto& b : m->bufs) {
    if (!(b.flags & SD_BUF_PERSISTENT)) continue;
    host.resize(b.bytes);
    if (hipMemcpy(host.data(), b.ptr, b.bytes, hipMemcpyDeviceToHost) != hipSuccess) { fclose(f); return fail(COMA_E_LAUNCH, "sd_model_save: device read failed"); }
    w.put(host.data(), b.bytes);
  }
  const uint64_t trailer[2] = {24 + w.hash.total, w.hash.digest()};
  ok = ok && w.ok && fseek(f, 8, SEEK_SET) == 0 && fwrite(trailer, 1, 16, f) == 16;
  ok = (fclose(f) == 0) && ok;
  return ok ? COMA_OK : fail(COMA_E_INVALID, "sd_model_save: short write to %s", path);
}

extern "C" int sd_model_load(const char* path, void** model) {
  if (!path || !model) return fail(COMA_E_INVALID, "sd_model_load: bad args");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(COMA_E_INVALID, "sd_model_load: cannot open %s", path);
  Model* m = new Model();
  auto bail = [&](const char* what) { fclose(f); sd_model_destroy(m); return fail(COMA_E_INVALID, "sd_model_load: %s (%s)", what, path); };
  char magic[8];
  uint64_t header[2];
  if (!rd(f, magic, 8) || memcmp(magic, "SDMODEL3", 8) || !rd(f, header, 16)) return bail("not a model file (or one of an older format version)");
  {
    // first pass: the file must be exactly what sd_model_save wrote -- size and checksum -- before anything is allocated or trusted
    Hasher hs;
    std::vector<char> chunk(1 << 22);
    for (size_t n; (n = fread(chunk.data(), 1, chunk.size(), f)) > 0;) hs.update(chunk.data(), n);
    if (24 + hs.total != header[0]) return bail("truncated or over-long file");
    if (hs.digest() != header[1]) return bail("checksum mismatch: the file is corrupt");
    if (fseek(f, 24, SEEK_SET) != 0) return bail("seek failed");
  }
  uint32_t nbuf = 0;
  if (!rd(f, &nbuf, 4) || nbuf > kMaxBufIndex) return bail("bad buffer count");
  for (uint32_t k = 0; k < nbuf; ++k) {
    uint64_t by; uint32_t fl;
    if (!rd(f, &by, 8) || !rd(f, &fl, 4) || by == 0 || by > kMaxOffset) return bail("bad buffer table");
    void* d = nullptr;
    if (hipMalloc(&d, by) != hipSuccess) return bail("out of device memory");
    m->bufs.push_back(Buffer{static_cast<char*>(d), (size_t)by, (int)fl, true});
    if (hipMemset(d, 0, by) != hipSuccess) return bail("memset failed");
  }
  uint32_t nbind = 0;
  if (!rd(f, &nbind, 4)) return bail("truncated");
  for (uint32_t k = 0; k < nbind; ++k) {
    char name[32]; uint32_t kb; uint64_t off, by;
    if (!rd(f, name, 32) || !rd(f, &kb, 4) || !rd(f, &off, 8) || !rd(f, &by, 8) || kb >= nbuf || off + by > m->bufs[kb].bytes) return bail("bad binding");
    name[31] = 0;
    m->binds[name] = Binding{m->bufs[kb].ptr + off, (size_t)by};
  }
  uint32_t nplans = 0;
  if (!rd(f, &nplans, 4) || nplans > 64) return bail("bad plan count");
  m->plans.reserve(nplans);
  for (uint32_t k = 0; k < nplans; ++k) {
    char name[32]; uint32_t nrec;
    if (!rd(f, name, 32) || !rd(f, &nrec, 4) || nrec > 10000000) return bail("bad plan header");
    name[31] = 0;
    m->plans.push_back(Plan());
    Plan& pl = m->plans.back();
    pl.name = name;
    pl.recs.resize(nrec);
    for (auto& r : pl.recs) {
      if (!rd(f, &r, sizeof r) || r.kind <= 0 || r.kind >= PK_COUNT_) return bail("bad launch record");
      for (auto& q : r.p) {
        const uint64_t v = reinterpret_cast<uint64_t>(q);
        if (!v) continue;
        const uint64_t kb = (v >> 48) - 1, off = v & 0xffffffffffffULL;
        if (kb >= nbuf || off >= m->bufs[kb].bytes) return bail("bad pointer in a launch record");
        q = m->bufs[kb].ptr + off;
      }
    }
  }
  std::vector<char> host;
  for (auto& b : m->bufs) {
    if (!(b.flags & SD_BUF_PERSISTENT)) continue;
    host.resize(b.bytes);
    if (!rd(f, host.data(), b.bytes)) return bail("truncated buffer data");
    if (hipMemcpy(b.ptr, host.data(), b.bytes, hipMemcpyHostToDevice) != hipSuccess) return bail("device write failed");
  }
  fclose(f);
  *model = m;
  return COMA_OK;
}

// ---- network-level entry points: stage the inputs into the model's bound buffers, replay the plan's hipGraph, copy the result
namespace sd {
static int io_copy(Model* m, const char* name, const void* src, void* dst, void* stream) {
  auto it = m->binds.find(name);
  if (it == m->binds.end()) return fail(COMA_E_INVALID, "model has no binding named '%s'", name);
  const hipError_t e = src ? hipMemcpyAsync(it->second.ptr, src, it->second.bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream)
                           : hipMemcpyAsync(dst, it->second.ptr, it->second.bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  return e == hipSuccess ? COMA_OK : fail(COMA_E_LAUNCH, "copy of binding '%s' failed: %s", name, hipGetErrorString(e));
}
static int forward(void* model, const char* plan, const char* in0, const void* src0, const char* in1, const void* src1, const char* out,
                   void* dst, void* stream) {
  if (!model) return fail(COMA_E_INVALID, "sd forward: null model");
  Model* m = as_model(model);
  int rc = COMA_OK;
  if (src0 && (rc = io_copy(m, in0, src0, nullptr, stream)) != COMA_OK) return rc;
  if (src1 && (rc = io_copy(m, in1, src1, nullptr, stream)) != COMA_OK) return rc;
  if ((rc = sd_model_replay(model, plan, stream)) != COMA_OK) return rc;
  if (dst && (rc = io_copy(m, out, nullptr, dst, stream)) != COMA_OK) return rc;
  return COMA_OK;
}
}  // namespace sd

extern "C" int sd_unet_set_context(void* model, const void* ctx, void* stream) {
  return sd::forward(model, "context", "ctx", ctx, nullptr, nullptr, nullptr, nullptr, stream);
}
extern "C" int sd_unet_forward(void* model, const void* x_in, const float* timesteps, void* eps_out, void* stream) {
  return sd::forward(model, "step", "x_in", x_in, "timesteps", timesteps, "eps", eps_out, stream);
}
extern "C" int sd_vae_decode(void* model, const void* z, void* image_out, void* stream) {
  return sd::forward(model, "decode", "z", z, nullptr, nullptr, "image", image_out, stream);
}
extern "C" int sd_vae_encode(void* model, const void* image, void* moments_out, void* stream) {
  return sd::forward(model, "encode", "x", image, nullptr, nullptr, "moments", moments_out, stream);
}
